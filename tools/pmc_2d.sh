#!/bin/bash
# PMC evidence for the cascaded 2-D kernels of cfg2 (Fwd2C / Inv2C, 4096^2 fp32 db4, 3 levels in one launch) on the current build.
# One rocprofv3 --pmc pass per counter group (nothing traced alongside).   tools/pmc_2d.sh  -> gpurun_out/pmc_2d/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
top=gpurun_out/pmc_2d
rm -rf $top; mkdir -p $top
run() {   # name, counters...
  local name=$1; shift
  timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d $top/$name -- python tools/ab_2d.py 0 > $top/$name.log 2>&1
}
run sq  SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU &&
run sq2 GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT &&
run ta  TA_BUSY_avr TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE
echo "== cfg2 ($(tail -2 $top/sq.log | tr '\n' ' '))" | tee -a $top/summary.txt
python tools/pmc_summary.py $top/sq $top/sq2 $top/ta $top/fetch $top/write | cut -c1-700 | tee -a $top/summary.txt
