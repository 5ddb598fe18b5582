#!/bin/bash
# 8-bit prefill GEMM, weights as load-time image: activations in place (MI355X_F8_ROWMAJOR=1) against packed (0),
# and both operands in place (3, no image) — per projection, small and large M.
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp13.txt
{
for rm in 0 1 0 1; do
  echo "== prepacked weights, MI355X_F8_ROWMAJOR=$rm"
  MI355X_F8_ROWMAJOR=$rm timeout -k 10 300 python scripts/bench_scaled_mm.py --prepacked 576 1024 4096 8192 2>&1 | grep "^fp8"
done
echo "== weights in place, MI355X_F8_ROWMAJOR=3"
MI355X_F8_ROWMAJOR=3 timeout -k 10 300 python scripts/bench_scaled_mm.py 576 1024 4096 8192 2>&1 | grep "^fp8"
for args in "--quant fp8" "--quant fp8 --chunk-tokens 512"; do
  MI355X_PREPACK_WEIGHTS=0 MI355X_F8_ROWMAJOR=3 timeout -k 10 400 python bench.py $args --skip-cpu 2> gpurun_out/r03c.err | tail -n 1 > gpurun_out/r03c_tmp.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r03c_tmp.json"))
print("no image, rowmajor=3 bench $args:", d["value"], d["ms_per_step"], d["ttft_p50_ms"], d.get("hbm_weights_gb"))
PY
done
} > $O 2>&1
tail -n 5 $O
