#!/bin/bash
# chunked-prefill job (512-token budget, fp8): activations in place against packed, per-kernel times from the bench's own table
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp14.txt
{
for rm in 0 1 0 1; do
  MI355X_F8_ROWMAJOR=$rm timeout -k 10 400 python bench.py --quant fp8 --chunk-tokens 512 --skip-cpu 2> gpurun_out/r03c.err | tail -n 1 > gpurun_out/r03c_tmp.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r03c_tmp.json"))
t={e["kernel"]:e for e in [d["roofline"]]+d["roofline_other"]}
print("rowmajor=$rm chunk512:", d["value"], d["ms_per_step"], d["ttft_p50_ms"])
for k,e in sorted(t.items(), key=lambda kv:-kv[1].get("share",0) if isinstance(kv[1].get("share",0),(int,float)) else 0)[:14]:
    print("   ", k, {x:(round(v,3) if isinstance(v,float) else v) for x,v in e.items() if x in ("avg_launch_us","launches","share","frac","ms_per_step")})
PY
done
} > $O 2>&1
tail -n 70 $O
