#!/bin/bash
# Round-3 measurement call on the GPU box: GPU test suite, the bench records of the round (default line, --quant fp8,
# one TP=8 rank of the 70B fp8 / 72B GPTQ models), then scripts/prof_r03.sh (rocprofv3 kernel trace + PMC passes).
# Everything lands under gpurun_out/; the summaries to keep are copied into profiles/ afterwards.
#   gpurun --timeout 1200 -- 'bash scripts/r03_measure.sh [tests] [bench] [prof]'
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
what="${*:-tests bench prof}"
set -o pipefail
if [[ $what == *tests* ]]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputests.log 2>&1 || { tail -n 30 gpurun_out/r03_gputests.log; exit 1; }
  tail -n 3 gpurun_out/r03_gputests.log
fi
if [[ $what == *bench* ]]; then
  timeout -k 10 300 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err || exit 2
  timeout -k 10 300 python bench.py --quant fp8 --skip-cpu > gpurun_out/r03_bench_quant_fp8.json 2> gpurun_out/r03_bench_quant_fp8.err || exit 3
  timeout -k 10 300 python bench.py --model llama-3-70b --tp-rank-of 8 --skip-cpu > gpurun_out/r03_rank_of_8_70b_fp8.json 2> gpurun_out/r03_rank_of_8_70b_fp8.err || exit 4
  timeout -k 10 300 python bench.py --model qwen2-72b --tp-rank-of 8 --skip-cpu > gpurun_out/r03_rank_of_8_72b_gptq.json 2> gpurun_out/r03_rank_of_8_72b_gptq.err || exit 5
  for f in gpurun_out/r03_bench_default.json gpurun_out/r03_bench_quant_fp8.json gpurun_out/r03_rank_of_8_*.json; do echo "== $f"; cut -c1-400 $f; done
fi
if [[ $what == *prof* ]]; then
  bash scripts/prof_r03.sh > gpurun_out/r03_prof.log 2>&1
  tail -n 12 gpurun_out/r03_prof.log
fi
