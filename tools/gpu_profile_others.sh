#!/bin/bash
# rocprofv3 kernel stats of the non-headline configurations -> gpurun_out/prof_others/<name>_kernel_stats.csv
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/prof_others
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- "$@" > $out/$name.log 2>&1
  f=$(ls $out/$name/*/*kernel_stats.csv | head -1)
  grep -E '^"Name"|ndwt::' $f > $out/${name}_kernel_stats.csv
  python - "$out/${name}_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"  {r['Name'][:110]:110s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
}
echo cfg2; run cfg2_2d python tools/bench2d.py
echo fp64_512; run fp64_512 python tools/bench_fp64.py 512
echo atrous; run atrous python tools/bench_atrous.py
echo cfg5; run cfg5_4d python tools/bench4d.py 256 256 256 32
