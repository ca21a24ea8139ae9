#!/bin/bash
# Recompile only the given translation units of libndwt_hip.so and relink with the other objects as they are (kernel
# iteration: `make` rebuilds everything that includes ndwt_device.h, 4 minutes).  tools/quick_relink.sh ndwt_fused3_f32_invy [...]
set -e
cd "$(dirname "$0")/../non-decimated_wavelets_amd/csrc"
V=${VARIANT:+_$VARIANT}       # VARIANT=stamps NDWT_DEFS=-DNDWT_STAMPS tools/quick_relink.sh ...  ->  build_stamps/, libndwt_hip_stamps.so
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed -mllvm -simplifycfg-sink-common=false"
for tu in "$@"; do
  extra=""
  case $tu in ndwt_fused3_f32_den|ndwt_fused3_f32_inv|ndwt_fused3_f32_inve|ndwt_fused3_f32_invy|ndwt_fused3_f32_invys|ndwt_fused3_f32_invyc|ndwt_fused3_f32_long|ndwt_fused3_f32_longb|ndwt_fused3_f32_longi|ndwt_fused3_f64_inv|ndwt_fused3_f64_long|ndwt_fused2_f32|ndwt_fused2_f32_fwdb|ndwt_fused2_f32_inva|ndwt_fused2_f32_invb|ndwt_fused2_f32_invp|ndwt_fused2_f32_fwdl|ndwt_fused2_f32_invl|ndwt_fused2_f32_invm|ndwt_fused2_f32_fwdc|ndwt_fused2_f32_invc|ndwt_fused2_f64_long|ndwt_fused2_f64|ndwt_fused2_f64_inv) extra="-fno-slp-vectorize";; esac
  hipcc $FLAGS $extra $NDWT_DEFS -c $tu.hip -o build$V/$tu.o &
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 build$V/*.o -o ../libndwt_hip$V.so
touch build$V/*.o ../libndwt_hip$V.so
ls -la ../libndwt_hip$V.so
