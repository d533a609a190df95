#!/bin/bash
# Rebuild the HIP library with resource remarks and dump a compact ISA summary of the named kernels.
set -e
cd "$(dirname "$0")/../spatial_vae_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Rpass-analysis=kernel-resource-usage api.hip -o ../libsvae_hip.so 2> /tmp/build.log || { grep -E "error" -A3 /tmp/build.log | head -30; exit 1; }
grep -A8 -E "Name: _ZN4svae12(dense_kernelILi(8)ELb[01]ELb0|wgrad)" /tmp/build.log | grep -E "Name|VGPRs:|AGPRs|Scratch|Occupancy" | sed -E 's/.*remark: //; s/ \[-Rpass.*//' | paste - - - - -
hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only api.hip -o /tmp/api.s 2>/dev/null
python3 ../../tools/isa_summary.py /tmp/api.s "$@"
