#!/bin/bash
# development: tools/build_variant.sh name [-Dflags...]  ->  variants/librtc_<name>.so (select with RTC_LIB=...)
# F64_FLAGS (environment) replaces the extra flags of the fp64 translation unit (default: the Makefile's)
cd "$(dirname "$0")/.." && mkdir -p variants
name=$1; shift
F64=${F64_FLAGS--mllvm -disable-machine-licm}
C="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-unused-function"
S=ray_tracer_challenge_amd/csrc
($C "$@" -c -o variants/$name.api.o $S/rtc_api.hip 2>&1 | grep -E "error") &
($C "$@" $F64 -c -o variants/$name.f64.o $S/rtc_trace_f64.hip 2>&1 | grep -E "error") &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o variants/librtc_$name.so variants/$name.api.o variants/$name.f64.o 2>&1 | grep -E "error"
rm -f variants/$name.api.o variants/$name.f64.o
echo "built $name"
