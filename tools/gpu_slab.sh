#!/bin/bash
# one rank's share of the 8-GPU run on one GPU (world size 1: the exchange is local copies): host overhead + small-slab kernels
set -o pipefail
mkdir -p gpurun_out
for ov in 1 0; do
  NDWT_BENCH_FORCE_SHARDED=1 NDWT_BENCH_OVERLAP=$ov python bench.py --size 512 512 64 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/slab_ov$ov.log 2>&1
  tail -1 gpurun_out/slab_ov$ov.log | cut -c1-400
done
python bench.py --size 512 512 64 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/slab_plain.log 2>&1; tail -1 gpurun_out/slab_plain.log | cut -c1-400
NDWT_BENCH_FORCE_SHARDED=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/slab_full_sh.log 2>&1; tail -1 gpurun_out/slab_full_sh.log | cut -c1-400
