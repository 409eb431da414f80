#!/bin/bash
# development: tools/build_variant.sh name [-Dflags...]  ->  variants/librtc_<name>.so (select with RTC_LIB=...)
cd "$(dirname "$0")/.." && mkdir -p variants
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=on -Wno-unused-function "$@" -o variants/librtc_$name.so ray_tracer_challenge_amd/csrc/rtc_api.hip 2>&1 | grep -E "error" 
echo "built $name"
