#!/bin/bash
# Diagnostic: build libibvh with -DIBVH_PHASE_STAMPS (s_memtime stamps at the phase boundaries of the sort kernels,
# written to a buffer nothing else reads) into variants/libibvh_stamps.so (HERE, where hipcc is; variants/ travels to the
# GPU box, gpurun_out/ does not), then on the GPU box:
#   IBVH_LIB=variants/libibvh_stamps.so python tools/phase_stamps.py <n>
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/implicitbvh.jl_amd/csrc
mkdir -p "$R/variants"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -DIBVH_PHASE_STAMPS -shared -Wl,--no-undefined -w -o "$R/variants/libibvh_stamps.so" $C/*.hip -ldl
echo built
