#!/bin/bash
# PMC evidence for the float synthesis kernel Inv3Y with 8 taps (cfg3) and 12 taps (cfg4) on the current build: instruction mix, wave-cycle
# shares, texture-addresser / L1 busy, fabric traffic.  One rocprofv3 --pmc pass per counter group (nothing traced alongside).
#   tools/pmc_inv3y.sh   -> gpurun_out/pmc_inv3y/<wname>/..., summary in gpurun_out/pmc_inv3y/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
top=gpurun_out/pmc_inv3y
rm -rf $top; mkdir -p $top
for w in db4 db6; do
  out=$top/$w
  mkdir -p $out
  timeout -k 10 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/sq -- python tools/ab_inv.py 0 $w > $out/sq.log 2>&1
  timeout -k 10 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/sq2 -- python tools/ab_inv.py 0 $w > $out/sq2.log 2>&1
  timeout -k 10 150 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC --output-format csv -d $out/sq3 -- python tools/ab_inv.py 0 $w > $out/sq3.log 2>&1
  timeout -k 10 150 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $out/ta -- python tools/ab_inv.py 0 $w > $out/ta.log 2>&1
  timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python tools/ab_inv.py 0 $w > $out/fetch.log 2>&1
  timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python tools/ab_inv.py 0 $w > $out/write.log 2>&1
  echo "== $w  ($(tail -1 $out/sq.log))" | tee -a $top/summary.txt
  python tools/pmc_summary.py $out/sq $out/sq2 $out/sq3 $out/ta $out/fetch $out/write | grep " INV " | cut -c1-600 | tee -a $top/summary.txt
done
