#!/bin/bash
# build tools/_bin/fir32_timeline (phase timeline of fft_fir32.h) here; run it on the GPU box with: gpurun -- tools/_bin/fir32_timeline
mkdir -p tools/_bin && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-pass-failed -DPF_FIR32_DEBUG "$@" -I pffft_amd/csrc -I include tools/fir32_timeline.hip -o tools/_bin/fir32_timeline 2>&1 | grep -v "warning\|^ *[0-9]* *|\|^ *|\|generated" | head
