#!/bin/bash
# prefill kernel K split for shapes with few tiles: tests, micro (MI355X_F8_PACKED_SK=1 = no split), chunked jobs
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp22.txt
{
timeout -k 10 900 python -m pytest tests/test_gpu_fp8_gemm.py tests/test_gpu_int8.py -x -q 2>&1 | tail -n 3
for sk in 1 0; do
  echo "== MI355X_F8_PACKED_SK=$sk (1 = unsplit, 0 = planned)"
  MI355X_F8_PACKED_SK=$sk timeout -k 10 300 python scripts/bench_scaled_mm.py --fp8 576 1024 2048 2>&1 | grep "^fp8"
  MI355X_F8_PACKED_SK=$sk timeout -k 10 300 python scripts/bench_scaled_mm.py --fp8 --70b-rank 576 2048 2>&1 | grep "^fp8"
done
for sk in 1 0 1 0; do
  for args in "--quant fp8 --chunk-tokens 512" "--model llama-3-70b --tp-rank-of 8 --chunk-tokens 2048"; do
    MI355X_F8_PACKED_SK=$sk timeout -k 10 400 python bench.py $args --skip-cpu --no-plugin-surface 2> gpurun_out/r03c.err | tail -n 1 > gpurun_out/r03c_tmp.json
    python - <<PY
import json
d=json.load(open("gpurun_out/r03c_tmp.json"))
t={e["kernel"]:e for e in [d["roofline"]]+d["roofline_other"]}
print("sk=$sk bench $args:", d["value"], d["ms_per_step"], d["ttft_p50_ms"], {k:round(e["avg_launch_us"],2) for k,e in t.items() if "gemm_large" in k})
PY
  done
done
} > $O 2>&1
tail -n 12 $O
