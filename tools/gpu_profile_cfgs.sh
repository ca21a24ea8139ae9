#!/bin/bash
# rocprofv3 evidence for the non-headline single-GPU BASELINE configurations (cfg2 4096^2 db4 L3, cfg4's transform 512^3 db6 L4,
# cfg5 256^3x32 db4 L3): kernel-trace stats of `python bench.py <cfg args>` and separate --pmc FETCH_SIZE / WRITE_SIZE passes.
#   tools/gpu_profile_cfgs.sh <tag> [cfg2 cfg4 cfg5 cfg3]   -> gpurun_out/prof_<tag>/{<cfg>_kernel_stats.csv, <cfg>_bench.json, traffic_<cfg>.json}
set -o pipefail
tag=${1:-x}; shift
cfgs=${@:-cfg2 cfg4 cfg5}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in $cfgs; do
  case $c in
    cfg2) args="--ndim 2 --size 4096 4096 --wname db4 --level 3 --steps 50";;
    cfg3) args="--ndim 3 --size 512 512 512 --wname db4 --level 3 --steps 20";;
    cfg4) args="--ndim 3 --size 512 512 512 --wname db6 --level 4 --steps 20";;
    cfg5) args="--ndim 4 --size 256 256 256 32 --wname db4 --level 3 --steps 5";;
  esac
  common="--warmup 3 --no-cpu-baseline --packed-only --no-others --no-live-traffic"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$c -- python bench.py $args $common > $out/${c}_bench.json 2> $out/stats_$c.log || { echo "stats pass of $c failed"; tail -5 $out/stats_$c.log; exit 1; }
  f=$(ls $out/stats_$c/*/*kernel_stats.csv | head -1)
  grep -E '^"Name"|ndwt::' $f > $out/${c}_kernel_stats.csv
  for k in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $k --output-format csv -d $out/pmc_${c}_$k -- python bench.py $args --steps 2 --warmup 1 --no-cpu-baseline --packed-only --no-others --no-live-traffic > /dev/null 2> $out/pmc_${c}_$k.log || { echo "pmc pass $k of $c failed"; exit 1; }
  done
  python tools/traffic_json.py $c "$args" $out/${c}_kernel_stats.csv $out/pmc_${c}_FETCH_SIZE $out/pmc_${c}_WRITE_SIZE > $out/traffic_$c.json
  python - $out/${c}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"  {r['Name'][:100]:100s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
  tail -1 $out/${c}_bench.json | cut -c1-300
  rm -rf $out/stats_$c $out/pmc_${c}_FETCH_SIZE $out/pmc_${c}_WRITE_SIZE
done
