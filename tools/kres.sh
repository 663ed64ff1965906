#!/bin/bash
# Register / LDS / occupancy of the kernels of one source file whose mangled name matches a pattern:
#   bash tools/kres.sh nr_attn_mfma.hip 'bwd_kernel'
cd "$(dirname "$0")/../newsrecommendation_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 \
  | grep -E "Function Name|VGPRs:|AGPRs|Spill|Occupancy|LDS Size|ScratchSize" | sed 's/.*remark: //; s/ \[-Rpass.*//' | paste - - - - - - - - | grep -E "$2" | sed 's/Function Name: //' | cut -c1-330
