#!/bin/bash
# Sanitizer build of the device code for the host (see tools/hostemu/hip/hip_runtime.h).  Usage:
#   bash tools/hostemu/build.sh            -> tools/hostemu/libvicgpu_hostemu.so  (ASan + UBSan)
#   SAN=none bash tools/hostemu/build.sh   -> same without sanitizers (for HOSTEMU_POISON runs)
# Run:  LD_PRELOAD=$(bash tools/hostemu/build.sh --asan-runtime) ASAN_OPTIONS=detect_leaks=0 \
#       VICGPU_LIB=$PWD/tools/hostemu/libvicgpu_hostemu.so python tools/hostemu/run_case.py quickflux_winter
set -e
CXX=/opt/rocm/lib/llvm/bin/clang++
if [ "$1" = "--asan-runtime" ]; then $CXX -print-file-name=libclang_rt.asan-x86_64.so; exit 0; fi
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SANFLAGS="-fsanitize=address,undefined -shared-libsan -fno-omit-frame-pointer"
OUT=$ROOT/tools/hostemu/libvicgpu_hostemu.so
if [ "${SAN:-asan}" = "none" ]; then SANFLAGS=""; OUT=$ROOT/tools/hostemu/libvicgpu_hostemu_plain.so; fi
$CXX ${EXTRA:-} -x c++ -std=c++17 -O1 -g -ffp-contract=off -ftrivial-auto-var-init=zero -fPIC -shared $SANFLAGS -Wno-unknown-attributes -Wno-ignored-attributes \
  -I$ROOT/tools/hostemu -I$ROOT/include -I$ROOT/vic_amd/csrc $ROOT/vic_amd/csrc/vicgpu_api.hip -o $OUT -lpthread
echo $OUT
