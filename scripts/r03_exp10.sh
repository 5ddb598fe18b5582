#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp10.txt
{
timeout -k 10 900 python -m pytest tests/test_gpu_cache_norm_rotary.py tests/test_gpu_ref_fixtures.py tests/test_gpu_golden_backend.py tests/test_gpu_tp.py -x -q 2>&1 | tail -n 5
for w in 1 0 1 0; do
for args in "--model llama-3-70b --tp-rank-of 8" "--model qwen2-72b --tp-rank-of 8"; do
  MI355X_NORM_WAVE_ROWS=$w timeout -k 10 400 python bench.py $args --skip-cpu --no-plugin-surface 2> gpurun_out/r03c.err | tail -n 1 > gpurun_out/r03c_tmp.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r03c_tmp.json"))
t={e["kernel"]:e for e in [d["roofline"]]+d["roofline_other"]}
names=[k for k in t if "norm" in k and "decode" not in k]
print("MI355X_NORM_WAVE_ROWS=$w $args:", d["value"], d["ms_per_step"], d["ttft_p50_ms"], {k:(round(t[k]["avg_launch_us"],1), round(t[k]["frac"],3)) for k in names})
PY
done
done
} > $O 2>&1
tail -n 20 $O
