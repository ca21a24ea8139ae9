#!/bin/bash
# parity + bench of every fused-kernel variant in one GPU call
mkdir -p gpurun_out
for vi in 1 2; do
  NDWT_VARIANT_INV=$vi python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or uneven or random or full_size" 2>&1 | tail -2
done
for v in "0 0" "0 1" "0 2" "1 1"; do set -- $v
  echo "== variant fwd=$1 inv=$2"
  NDWT_VARIANT_FWD=$1 NDWT_VARIANT_INV=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['ms_per_step'], 'ms/step', d['value'], 'Mvox/s', r['kernel'], r['avg_launch_ms'], r['other_kernel'], 'rt', d.get('roundtrip_rel_l2'))"
done
