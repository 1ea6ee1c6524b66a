#!/bin/bash
# Regenerate the Stockham compile-time plans and re-tune the per-kernel occupancy caps until nothing spills.
#   tools/gen_stock_plans.hip -> pffft_amd/csrc/stock_ct_*_gen.hip ; tools/tune_stock_wpe.py -> pffft_amd/csrc/stock_wpe_gen.h
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
/opt/rocm/bin/hipcc -std=c++17 -O1 --offload-arch=gfx950 tools/gen_stock_plans.hip -o /tmp/gen_stock_plans
/tmp/gen_stock_plans pffft_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None --cuda-device-only -S -Rpass-analysis=kernel-resource-usage"
for pass in 1 2 3 4; do
  : > /tmp/stock_res.txt
  for t in f32c_a f32c_b f32r_a f32r_b f64c_a f64c_b f64r_a f64r_b; do
    ( cd /tmp && /opt/rocm/bin/hipcc $FLAGS -o /tmp/stock_$t.s $ROOT/pffft_amd/csrc/stock_ct_${t}_gen.hip 2> /tmp/stock_res_$t.txt ) &
  done
  wait
  cat /tmp/stock_res_*.txt > /tmp/stock_res.txt
  if grep -q "error" /tmp/stock_res.txt; then grep -A5 error /tmp/stock_res.txt | head -30; exit 1; fi
  python tools/tune_stock_wpe.py /tmp/stock_res.txt | tee /tmp/stock_tune.txt
  grep -q "(0 new)" /tmp/stock_tune.txt && break
done
