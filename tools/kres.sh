#!/bin/bash
# Register / spill / LDS report of one HIP source of the library (no GPU needed):
#   tools/kres.sh mlp_chain.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off \
  -munsafe-fp-atomics -fvisibility=hidden -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage "$@" \
  -I3dioumatch_amd/csrc -c 3dioumatch_amd/csrc/$src -o /tmp/kres.o 2>&1 | grep -E "error|warning:|Function Name|VGPRs:|AGPRs:|Spill|Occupancy|ScratchSize" | grep -v -A7 zero_words
