#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the fp64 fused kernels at 512^3 (tools/pmc_summary.py prints per-launch averages)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/f64_$c -- python tools/bench_fp64.py 512 > /dev/null 2>&1
  python tools/pmc_summary.py gpurun_out/f64_$c
done
