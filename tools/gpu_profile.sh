#!/bin/bash
# Profile artifacts for profiles/: kernel-trace stats of the default bench (--packed-only: without the secondary pass on
# pitched coefficients, which launches the same kernels), then FETCH_SIZE / WRITE_SIZE in separate --pmc passes.  Usage (on the GPU box): tools/gpu_profile.sh <tag>   -> gpurun_out/prof_<tag>/...
set -o pipefail
tag=${1:-x}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --packed-only --no-others --no-live-traffic > $out/bench_under_profiler.json 2> $out/stats.log
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
head -3 $out/kernel_stats.csv | cut -c1-200
tail -1 $out/bench_under_profiler.json | cut -c1-400
rm -f $out/pmc_summary.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --packed-only --no-others --no-live-traffic > /dev/null 2> $out/pmc_$c.log
  python tools/pmc_summary.py $out/pmc_$c | tee -a $out/pmc_summary.txt
  # the same with pitched coefficients (ndwt_band_pitch)
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_pitched_$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --band-pitch auto --no-others --no-live-traffic > /dev/null 2> $out/pmc_pitched_$c.log
  python tools/pmc_summary.py $out/pmc_pitched_$c | tee -a $out/pmc_summary.txt
done
python bench.py --steps 20 --warmup 5 > $out/bench.json 2>/dev/null; tail -1 $out/bench.json
