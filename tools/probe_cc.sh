#!/bin/bash
# compile one probe translation unit (a file that includes the csrc headers) for gfx950 with the library's flags and print the
# compiler's resource remarks (VGPRs, scratch, occupancy) per kernel:   tools/probe_cc.sh file.hip [-Dflags...]
set -e
src=$1; shift
cd /root/repo/pffft_amd/csrc
cp "$src" ./_probe_tu.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wno-pass-failed \
  -mllvm -amdgpu-atomic-optimizer-strategy=None -Rpass-analysis=kernel-resource-usage "$@" -c _probe_tu.hip -o /tmp/_probe_tu.o 2>&1 \
  | grep -i "Function Name\|VGPRs:\|Scratch\|Occupancy \[" | sed -e 's/.*remark: *//' -e 's/\[-Rpass.*//'
rm -f _probe_tu.hip
