#!/bin/bash
# L2 hit / miss / request counters of the synthesis kernels: float (cfg3 bench) and double (512^3)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d gpurun_out/l2_f32 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/l2_f32 | cut -c1-200
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d gpurun_out/l2_f64 -- python tools/bench_fp64.py 512 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/l2_f64 | cut -c1-200
