# fabric traffic of the fused kernels with the band streams skewed (diagnostic library, see tools/exp_bandpad.py)
# tools/pmc_pad.sh pad/div/mask ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export NDWT_LIB_VARIANT=exp
for c in "$@"; do
  out=gpurun_out/pmc_pad_${c//\//_}; rm -rf $out; mkdir -p $out
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python tools/exp_bandpad.py $c $c > $out/fetch.log 2>&1
  echo "== pad/div/mask $c"; grep "^pad" $out/fetch.log | tail -1; python tools/pmc_summary.py $out/fetch | grep INV | cut -c1-80
done
