#!/bin/bash
# PMC picture of the float synthesis kernel variants (NDWT_VARIANT_INV values given as arguments): wave-cycle shares, VALU
# activity, clock, fabric traffic.  One rocprofv3 --pmc pass per counter group (kernel-trace / stats only alongside).
#   tools/pmc_inv.sh 0 5 6      -> gpurun_out/pmc_inv_<variant>/..., summary on stdout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  out=gpurun_out/pmc_inv_$v
  rm -rf $out; mkdir -p $out
  export NDWT_VARIANT_INV=$v
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/sq -- python tools/ab_inv.py $v > $out/sq.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/sq2 -- python tools/ab_inv.py $v > $out/sq2.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python tools/ab_inv.py $v > $out/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python tools/ab_inv.py $v > $out/write.log 2>&1
  echo "== variant $v"; tail -1 $out/sq.log
  python tools/pmc_summary.py $out/sq $out/sq2 $out/fetch $out/write | grep " INV " | cut -c1-420
done
