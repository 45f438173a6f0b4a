#!/bin/bash
# Development aid: register / scratch usage of the pipeline kernel instantiations, with the Makefile's flags.
# Usage: tools/kres.sh [file.hip] [name filter]
cd "$(dirname "$0")/../supersonic_amd/csrc"
F=${1:-pipeline_kernels.hip}; PAT=${2:-pipeline_kernelILi}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -munsafe-fp-atomics -mllvm -structurizecfg-skip-uniform-regions \
  -mllvm -amdgpu-use-divergent-register-indexing -c "$F" -o /tmp/kres_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|VGPRs Spill|SGPRs Spill" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - - - | grep "$PAT"
rm -f /tmp/kres_$$.o
