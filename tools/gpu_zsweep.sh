#!/bin/bash
# z-chunk sweep of the analysis kernel (workgroups = 256 tiles x chunks; 768 fit on the chip at once)
run() {
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$*', d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['other_kernel'])"
}
run NDWT_ZCHUNK_FWD=64
run NDWT_ZCHUNK_FWD=256
run NDWT_ZCHUNK_FWD=512
run NDWT_ZCHUNK_FWD=200
run NDWT_ZCHUNK_FWD=256 NDWT_VARIANT_FWD=2
run NDWT_ZCHUNK_FWD=512 NDWT_VARIANT_FWD=2
run NDWT_ZCHUNK_FWD=256 NDWT_VARIANT_FWD=1
run NDWT_ZCHUNK_FWD=256
run NDWT_ZCHUNK_FWD=64
