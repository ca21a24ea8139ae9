import importlib, os, sys, time
sys.path.insert(0, "/root/repo")
import torch
sh = importlib.import_module("non-decimated_wavelets_amd.sharded")
dev = torch.device("cuda", 0)
for wname, level in (("db4", 3),):
    x = torch.randn(64, 512, 512, device=dev)
    for overlap in (False, True):
        eng = sh.ShardedNdDwt([wname] * 3, [512, 512, 64], pres_l2_norm=True, precision="single", device=dev, overlap=overlap)
        for _ in range(5):
            r = eng.rec(eng.dec(x, level))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            r = eng.rec(eng.dec(x, level))
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{wname} overlap={overlap}: enqueue {(t1-t0)/200*1e3:.4f} ms/step, total {(t2-t0)/200*1e3:.4f} ms/step")
