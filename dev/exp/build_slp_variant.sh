#!/bin/bash
# the library with naic.hip (the bounding tail) compiled WITH the SLP vectoriser, every other object as shipped:
# the build in which round 2 saw v_pk_fma_f32 chains give wrong sums beside other kernels' MFMAs
set -e
cd "$(dirname "$0")/../.."
python -m boficap_amd.build > /dev/null
mkdir -p build/obj_slp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude -Iboficap_amd/csrc $EXTRA -c boficap_amd/csrc/naic.hip -o build/obj_slp/naic.o
objs=$(ls build/obj/*.o | grep -v '/naic.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o boficap_amd/libboficap_hip_slp.so $objs build/obj_slp/naic.o
/opt/rocm/lib/llvm/bin/llvm-objdump -d boficap_amd/libboficap_hip_slp.so > /dev/null 2>&1 || true      # (no --offloading here: it writes the code objects next to the library)
echo built boficap_amd/libboficap_hip_slp.so
