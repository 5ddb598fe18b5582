#!/bin/bash
# 8-bit decode GEMM: activation copies as whole lines (8 rows x 128 bytes) against the 16 rows x 16-byte-per-lane form
# (variants/libf8old.so = the previous commit's fp8_gemm.hip): tests, micro (L2-hot activations), rank / 8B jobs
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp17.txt
{
timeout -k 10 600 python -m pytest tests/test_gpu_fp8_gemm.py tests/test_gpu_int8.py -x -q 2>&1 | tail -n 2
for lib in variants/libf8old.so "" variants/libf8old.so ""; do
  echo "== lib=${lib:-new}"
  MI355X_HOTPATH_LIB=$lib timeout -k 10 200 python scripts/bench_scaled_mm_decode.py fp8 70b 2>&1 | tail -n 6
done
for lib in variants/libf8old.so "" variants/libf8old.so ""; do
  for args in "--model llama-3-70b --tp-rank-of 8" "--quant fp8"; do
    MI355X_HOTPATH_LIB=$lib timeout -k 10 400 python bench.py $args --skip-cpu --no-plugin-surface 2> gpurun_out/r03c.err | tail -n 1 > gpurun_out/r03c_tmp.json
    python - <<PY
import json
d=json.load(open("gpurun_out/r03c_tmp.json"))
t={e["kernel"]:e for e in [d["roofline"]]+d["roofline_other"]}
print("lib=${lib:-new} bench $args:", d["value"], d["ms_per_step"], d.get("decode_ms_per_token"), {k:round(e["avg_launch_us"],2) for k,e in t.items() if "gemm_small" in k})
PY
  done
done
} > $O 2>&1
tail -n 45 $O
