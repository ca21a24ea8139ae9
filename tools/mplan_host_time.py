"""Host time of the single-process multi-device plan (ndwt_mplan_*: ONE host thread queues the work of every slab): how long ndwt_mdec / ndwt_mrec
take to QUEUE a call for G slabs (all on device 0 here), next to what one device of a G-GPU run would compute.  If the queueing takes longer
than one device's share of the compute, a real G-GPU run is bound by the host thread.  python tools/mplan_host_time.py [G] [wname] [level]"""
import importlib
import sys
import time

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
L = importlib.import_module("non-decimated_wavelets_amd._lib")
as_json = "--json" in sys.argv
if as_json:
    sys.argv.remove("--json")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wname = sys.argv[2] if len(sys.argv) > 2 else "db4"
level = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = 512
res = {}
for overlap, threads in ((True, True), (True, False), (False, True), (False, False)):
    mp = api.MultiPlan([n, n, n], [wname] * 3, torch.float32, [0] * G, pres_l2_norm=True, max_level=level).set_overlap(overlap).set_threads(threads)
    xs = [torch.randn(nz, n, n, device="cuda") for (_, _, nz) in mp.slabs()]
    ys = mp.dec_device(xs, level)
    rs = mp.rec_device(ys)
    reps, qd, qr, wall = 20, 0.0, 0.0, 0.0
    for i in range(reps):
        t0 = time.perf_counter()
        ys = mp.dec_device(xs, level)
        qd += L.lib().ndwt_mplan_last_enqueue_us(mp._h)
        rs = mp.rec_device(ys)
        qr += L.lib().ndwt_mplan_last_enqueue_us(mp._h)
        wall += time.perf_counter() - t0
    err = max(float((r - x).abs().max()) for r, x in zip(rs, xs))
    if overlap:
        res["host_thread_per_slab" if threads else "one_host_thread"] = {"queueing_ms_per_dec_rec": round((qd + qr) / reps / 1e3, 3),
                                                                          "wall_ms_all_slabs_on_one_gpu": round(wall / reps * 1e3, 3)}
    if not as_json:
        print(f"{G} slabs on one GPU, {wname} {level} levels, overlap={overlap}, one host thread per slab={threads}: queueing dec {qd / reps:7.1f} us + rec {qr / reps:7.1f} us per call; "
          f"wall {wall / reps * 1e3:6.2f} ms per dec+rec (all slabs on ONE device: /{G} = {wall / reps * 1e3 / G:5.2f} ms per device); max |rec - x| {err:.1e}")
    del mp, xs, ys, rs
if as_json:
    import json
    res["slabs"] = G
    res["one_device_share_ms"] = round(res["host_thread_per_slab"]["wall_ms_all_slabs_on_one_gpu"] / G, 3)
    print(json.dumps(res))
