#!/bin/bash
# Diagnostic: build libibvh with -DIBVH_PHASE_STAMPS (s_memtime stamps at the phase boundaries of the sort kernels,
# written to a buffer nothing else reads) into variants/libibvh_stamps.so (HERE, where hipcc is; variants/ travels to the
# GPU box, gpurun_out/ does not), then on the GPU box:
#   IBVH_LIB=variants/libibvh_stamps.so python tools/phase_stamps.py <n>
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/implicitbvh.jl_amd/csrc; O=/tmp/ibvh_stamps_obj
mkdir -p "$R/variants" "$O"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -DIBVH_PHASE_STAMPS -w"
ls $C/*.hip | xargs -P 8 -I{} sh -c "f={}; /opt/rocm/bin/hipcc $FLAGS -c \$f -o $O/\$(basename \$f .hip).o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o "$R/variants/libibvh_stamps.so" $O/*.o -ldl
echo built
