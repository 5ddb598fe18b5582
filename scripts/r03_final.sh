#!/bin/bash
# Round-3 final measurement on the GPU box (one call): GPU suite, bench records (default line with cpu_baseline +
# plugin_surface, fp8, fp8 KV, the two TP = 8 rank rehearsals, 512-token chunked prefill), rocprofv3 kernel traces of the
# default job and the 70B rank.  Outputs under gpurun_out/r03f_*; copy what is kept into profiles/.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
what="${*:-tests bench prof}"
if [[ $what == *tests* ]]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03f_gputests.log 2>&1 || { tail -n 30 gpurun_out/r03f_gputests.log; exit 1; }
  tail -n 3 gpurun_out/r03f_gputests.log
fi
run() {  # name, args...
  local name=$1; shift
  timeout -k 10 400 python bench.py "$@" 2> gpurun_out/r03f_$name.err | tail -n 1 > gpurun_out/r03f_$name.json || { echo "bench $name failed"; tail -n 5 gpurun_out/r03f_$name.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03f_$name.json"))
print("$name:", d["value"], d["ms_per_step"], d.get("ttft_p50_ms"), d["roofline"]["kernel"], round(d["roofline"]["frac"],3), (d.get("plugin_surface") or {}).get("value"))
PY
}
if [[ $what == *bench* ]]; then
  run bench_default || exit 2
  run bench_quant_fp8 --quant fp8 --skip-cpu || exit 2
  run bench_fp8kv --kv-cache-dtype fp8 --skip-cpu --no-plugin-surface || exit 2
  run rank_of_8_70b_fp8 --model llama-3-70b --tp-rank-of 8 --skip-cpu || exit 2
  run rank_of_8_72b_gptq --model qwen2-72b --tp-rank-of 8 --skip-cpu || exit 2
  run bench_chunk512_awq --chunk-tokens 512 --skip-cpu --no-plugin-surface --steps 2 || exit 2
  run bench_chunk512_fp8 --quant fp8 --chunk-tokens 512 --skip-cpu --no-plugin-surface --steps 2 || exit 2
  run bench_chunk2048_awq --chunk-tokens 2048 --skip-cpu --no-plugin-surface --steps 2 || exit 2
  run rank_of_8_70b_fp8_chunk2048 --model llama-3-70b --tp-rank-of 8 --chunk-tokens 2048 --skip-cpu --steps 2 || exit 2
  run rank_of_8_72b_gptq_chunk2048 --model qwen2-72b --tp-rank-of 8 --chunk-tokens 2048 --skip-cpu --steps 2 || exit 2
fi
if [[ $what == *prof* ]]; then
  O=gpurun_out/prof_r03f
  rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 bench.py --steps 1 --warmup 1 --skip-cpu --no-plugin-surface > $O/bench_under_rocprof.json 2> $O/bench.err
  python3 scripts/prof_summary.py $O/bench > gpurun_out/r03f_bench_kernel_trace_summary.txt
  find $O/bench -name '*kernel_stats.csv' -exec cp {} gpurun_out/r03f_bench_kernel_stats.csv \;
  find $O/bench -name '*.csv' -size +1M -delete
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/rank8 -- python3 bench.py --model llama-3-70b --tp-rank-of 8 --steps 1 --warmup 1 --skip-cpu > $O/rank8_under_rocprof.json 2> $O/rank8.err
  python3 scripts/prof_summary.py $O/rank8 > gpurun_out/r03f_rank_of_8_70b_fp8_kernel_trace_summary.txt
  find $O/rank8 -name '*kernel_stats.csv' -exec cp {} gpurun_out/r03f_rank_of_8_70b_fp8_kernel_stats.csv \;
  find $O/rank8 -name '*.csv' -size +1M -delete
  tail -n 1 $O/bench_under_rocprof.json > gpurun_out/r03f_bench_under_rocprof.json
  head -n 16 gpurun_out/r03f_bench_kernel_trace_summary.txt | cut -c1-60,111-200
  head -n 22 gpurun_out/r03f_rank_of_8_70b_fp8_kernel_trace_summary.txt | cut -c1-60,111-200
fi
