#!/usr/bin/env python3
"""Kernel timeline of one rank's share (world size 1) from a rocprofv3 kernel trace: per kernel start, duration and the gap to the one before.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/slab_timeline.py run <overlap 0|1>
    python tools/slab_timeline.py show <dir>
"""
import csv
import glob
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    import torch
    sh = importlib.import_module("non-decimated_wavelets_amd.sharded")
    dev = torch.device("cuda", 0)
    x = torch.randn(64, 512, 512, device=dev)
    if len(sys.argv) > 3 and sys.argv[3] == "after_one_piece":     # allocation history: an engine of the other schedule ran (and died) before
        e0 = sh.ShardedNdDwt(["db4"] * 3, [512, 512, 64], pres_l2_norm=True, precision="single", device=dev, overlap=False)
        for _ in range(10):
            r = e0.rec(e0.dec(x, 3))
        torch.cuda.synchronize()
    eng = sh.ShardedNdDwt(["db4"] * 3, [512, 512, 64], pres_l2_norm=True, precision="single", device=dev, overlap=sys.argv[2] == "1",
                          two_streams=False)
    for _ in range(10):
        r = eng.rec(eng.dec(x, 3))
    torch.cuda.synchronize()
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-(len(rows) // 10) * 2:]                      # the last two steps
    prev_end = None
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {r['Kernel_Name'][:90]}")
        prev_end = e
    print("span of two steps: %.1f us" % ((int(rows[-1]["End_Timestamp"]) - t0) / 1e3))
