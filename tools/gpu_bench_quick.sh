#!/bin/bash
# quick GPU iteration: parity tests (fast subset) + bench + LDS/occupancy counters
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or random or full_size or uneven" > gpurun_out/q_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/q_tests.log
tail -3 gpurun_out/q_tests.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/q_bench.log 2>&1; tail -1 gpurun_out/q_bench.log | cut -c1-900
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d gpurun_out/q_pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/q_pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/q_pmc
