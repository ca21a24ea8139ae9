#!/bin/bash
# time (+ optionally FETCH_SIZE) of the synthesis kernel for several variants / z-chunks
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export NDWT_ZCHUNK_FWD=64
for cfg in "0 64" "1 128" "1 256" "1 512" "3 64" "3 128" "3 256" "4 128"; do set -- $cfg
  export NDWT_VARIANT_INV=$1 NDWT_ZCHUNK_INV=$2
  t=$(python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['other_kernel']['avg_launch_ms'], 'rt', d['roundtrip_rel_l2'])")
  echo "inv_variant=$1 zchunk_inv=$2 : $t"
done
