#!/usr/bin/env python3
"""profiles/r03_traffic.json from the pmc_summary.txt that tools/gpu_profile.sh writes (per-launch FETCH_SIZE / WRITE_SIZE of the headline
command, packed and pitched coefficients).  python tools/headline_traffic_json.py gpurun_out/prof_<tag> <tag> > profiles/r03_traffic.json"""
import ast
import json
import os
import re
import sys

d, tag = sys.argv[1], sys.argv[2]
vals = {}
for line in open(os.path.join(d, "pmc_summary.txt")):
    m = re.match(r"\S*/pmc_(pitched_)?(FETCH_SIZE|WRITE_SIZE) (FWD|INV) (\{[^}]*\})", line)
    if m:
        vals[(bool(m.group(1)), m.group(3), m.group(2))] = ast.literal_eval(m.group(4))[m.group(2)]
names = {}
for line in open(os.path.join(d, "kernel_stats.csv")):
    m = re.search(r"fused3_kernel<ndwt::((Fwd3|Inv3Y)<[^>]*>)", line)
    if m:
        names.setdefault(m.group(2), m.group(1).replace(" ", ""))
ALG = 36 * 512 ** 3


def entry(pitched, kind):
    f, w = vals[(pitched, kind, "FETCH_SIZE")], vals[(pitched, kind, "WRITE_SIZE")]
    t = (2 * f + w) * 1024
    return {"kernel": names["Fwd3" if kind == "FWD" else "Inv3Y"], "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "traffic_bytes": t,
            "algorithmic_bytes": ALG, "ratio": round(t / ALG, 3)}


out = {
    "workload": "cfg3: 3D fp32 512x512x512 db4, one level launch = 134217728 voxels",
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --steps 2 --warmup 1 --no-cpu-baseline "
              "--packed-only --no-others --no-live-traffic` and, for `pitched`, `... --band-pitch auto` (tools/gpu_profile.sh " + tag + "); per-launch "
              "averages by tools/pmc_summary.py; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM: FETCH_SIZE counts 64-B "
              "requests that gfx950 issues for 128-B lines)",
    "fused_synthesis": entry(False, "INV"), "fused_analysis": entry(False, "FWD"),
    "pitched": {"band_pitch": "prod(dims) + 64 elements (ndwt_band_pitch)", "fused_synthesis": entry(True, "INV"), "fused_analysis": entry(True, "FWD")},
    "note": "round " + tag[1:3].lstrip("0") + "; rocprofv3 --kernel-trace --stats of the packed command: profiles/" + tag[:3] + "_kernel_stats.csv.  bench.py measures the same two "
            "counters live (two rocprofv3 child passes) and falls back to this file only when rocprofv3 is missing.",
}
print(json.dumps(out, indent=1))
