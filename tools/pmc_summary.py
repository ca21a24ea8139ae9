#!/usr/bin/env python3
"""Average per-launch counter values of the fused kernels from a rocprofv3 --pmc csv directory."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in rows:
            k = r["Kernel_Name"]
            if "ndwt" not in k:
                continue
            kind = ("INV" if "Inv" in k else "FWD" if "Fwd" in k else "AXIS") + ("2" if "2<" in k or "Fwd2" in k or "Inv2" in k else "")
            agg[kind][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[kind] = {x: r.get(x) for x in ("VGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size")}
        for kind in sorted(agg):
            print(d, kind, {c: round(sum(v) / len(v)) for c, v in sorted(agg[kind].items())}, meta[kind])
