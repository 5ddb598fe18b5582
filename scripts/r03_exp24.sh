#!/bin/bash
# w4a16 image GEMM from 384 rows + K split for shapes with few tiles: tests, micro, chunked and default jobs
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp24.txt
{
timeout -k 10 900 python -m pytest tests/test_gpu_w4a16.py tests/test_gpu_patch_fused_layers.py tests/test_gpu_golden_backend.py -x -q 2>&1 | tail -n 4
echo "== per-call awq_gemm (stripe passes below 1024 rows)"
timeout -k 10 300 python scripts/bench_gemm.py 576 1024 2048 2>&1 | grep "total"
echo "== image GEMM, unsplit (MI355X_W4_PACKED_SK=1)"
MI355X_W4_PACKED_SK=1 timeout -k 10 300 python scripts/bench_gemm.py 576 1024 2048 --image 2>&1 | grep "total"
echo "== image GEMM, planned split"
timeout -k 10 300 python scripts/bench_gemm.py 576 1024 2048 --image 2>&1 | grep -v amdgpu
for args in "--chunk-tokens 512" "--chunk-tokens 2048" ""; do
    timeout -k 10 400 python bench.py $args --skip-cpu 2> gpurun_out/r03c.err | tail -n 1 > gpurun_out/r03c_tmp.json
    python - <<PY
import json
d=json.load(open("gpurun_out/r03c_tmp.json"))
t={e["kernel"]:e for e in [d["roofline"]]+d["roofline_other"]}
ps=d.get("plugin_surface") or {}
print("bench $args:", d["value"], d["ms_per_step"], d["ttft_p50_ms"], {k:(round(e["avg_launch_us"],2), round(e["frac"],3)) for k,e in t.items() if "gemm_large" in k}, ps.get("value"), (ps.get("with_register_patch") or {}).get("value"))
PY
done
} > $O 2>&1
tail -n 30 $O
