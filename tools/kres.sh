#!/bin/bash
# usage: tools/kres.sh <translation unit in pffft_amd/csrc> [extra hipcc flags]: registers / scratch / occupancy of every kernel
# in it (device-only compile, no GPU needed)
TU=$1; shift
cd "$(dirname "$0")/../pffft_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wno-pass-failed \
  -mllvm -amdgpu-atomic-optimizer-strategy=None --cuda-device-only -c -o /tmp/kres_$$.o "$TU" "$@" \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|ScratchSize|Occupancy" \
  | sed -e 's/.*remark: //' -e 's/.*error:/error:/' -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | paste - - - - \
  | sed -e 's/Function Name: //' | while read -r name rest; do echo "$(echo $name | c++filt | cut -c1-150) | $rest"; done
rm -f /tmp/kres_$$.o
