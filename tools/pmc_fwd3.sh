#!/bin/bash
# PMC picture of the float analysis kernel Fwd3 with 8 taps (cfg3) and 12 taps (cfg4, the pinned-tap form) on the current build: the counterpart of
# tools/pmc_inv3y.sh.   tools/pmc_fwd3.sh -> gpurun_out/pmc_fwd3/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
top=gpurun_out/pmc_fwd3
rm -rf $top; mkdir -p $top
for w in db4 db6; do
  out=$top/$w
  mkdir -p $out
  cmd="python tools/bench_wavelets.py 512 ${w#db} f32fused"
  timeout -k 10 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/sq -- $cmd > $out/sq.log 2>&1
  timeout -k 10 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/sq2 -- $cmd > $out/sq2.log 2>&1
  timeout -k 10 150 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $out/ta -- $cmd > $out/ta.log 2>&1
  timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $cmd > $out/fetch.log 2>&1
  timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $cmd > $out/write.log 2>&1
  echo "== $w" | tee -a $top/summary.txt
  python tools/pmc_summary.py $out/sq $out/sq2 $out/ta $out/fetch $out/write | grep " FWD " | cut -c1-600 | tee -a $top/summary.txt
done
