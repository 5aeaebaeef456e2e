#!/bin/bash
# development helper: tools/build_variant.sh <name> [hipcc flags...]  ->  <package>/variants/libprt_<name>.so
# (same sources as libprt.so with extra -D flags; picked up by tools/ab_builds.sh / ab_lib_env.sh through PRT_LIB)
set -e
name=$1; shift
P="$(cd "$(dirname "$0")/.." && pwd)/photorealistic-rendering-using-opencl_amd"
O=/tmp/prt_variant_$name; mkdir -p $O $P/variants
# the variant says what it is (prt_build_id): its name, the hash of the sources it was built from and of its extra flags
ID="variant-$name-$(cd $P/.. && python3 -c "
import importlib.util, sys
spec = importlib.util.spec_from_file_location('b', '$P/build.py'); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
print(m.source_build_id(extra=sys.argv[1:]))" "$@")"
COMMON="-std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -DPT_UNITY -I$P/../include -I$P/csrc/host -I$P/csrc/hip"
for s in prt_api.cpp pt_pack.cpp pt_kernels.hip prt_build_id.cpp; do
  /opt/rocm/bin/hipcc $COMMON -x hip --offload-arch=gfx950 "$@" -DPRT_BUILD_ID="\"$ID\"" -c $P/csrc/hip/$s -o $O/$s.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $P/variants/libprt_$name.so $O/*.o $P/build/host_*.o
echo $P/variants/libprt_$name.so $ID
