#!/bin/bash
# Runs the host-emulation tests of the HIP kernels under AddressSanitizer (CPU only).
set -e
cd "$(dirname "$0")/../.."
make -j8 -C tests/emu libndwt_emu.so
ASAN_RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
[ -f "$ASAN_RT" ] || ASAN_RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
LD_PRELOAD="$ASAN_RT" ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_emulated_kernels.py -q -x "$@"
