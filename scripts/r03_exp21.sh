#!/bin/bash
# decode attention with the fused qkv prologue: TIMING-ONLY build without the release fence (vmcnt(0)) in front of the barrier
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp21.txt
{
for lib in "" variants/libpanofence.so "" variants/libpanofence.so; do
    MI355X_HOTPATH_LIB=$lib timeout -k 10 400 python bench.py --skip-cpu --no-plugin-surface 2> gpurun_out/r03c.err | tail -n 1 > gpurun_out/r03c_tmp.json
    python - <<PY
import json
d=json.load(open("gpurun_out/r03c_tmp.json"))
t={e["kernel"]:e for e in [d["roofline"]]+d["roofline_other"]}
print("lib=${lib:-default}:", d["value"], d["ms_per_step"], {k:round(e["avg_launch_us"],2) for k,e in t.items() if "attention_v" in k})
PY
done
} > $O 2>&1
cat $O
