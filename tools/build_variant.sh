#!/bin/bash
# A/B builds: tools/build_variant.sh <name> <source.hip> <extra hipcc flags...>
# compiles ONE source of dtcwt_amd/csrc with the extra flags and links it with the other (already built) objects
# into dtcwt_amd/libdtcwt_hip_<name>.so; select it at run time with DTCWT_HIP_LIBRARY (tools/ab_*.sh).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; shift 2
cd $R/dtcwt_amd/csrc
make -s -j8 >/dev/null
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -Wall -Wno-unused-function"
[ "$src" = fused2d_inv.hip ] && flags="$flags -fno-slp-vectorize"
[ "$src" = fused3d.hip ] && [ -z "$SLP" ] && flags="$flags -fno-slp-vectorize"
/opt/rocm/bin/hipcc $flags "$@" -c $src -o /tmp/variant_$name.o
objs=""
for o in *.o; do [ "$o" = "${src%.hip}.o" ] && objs="$objs /tmp/variant_$name.o" || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdtcwt_hip_$name.so $objs -ldl -lpthread
echo built dtcwt_amd/libdtcwt_hip_$name.so
