#!/bin/bash
# late-round-3 kernel traces: the fp8 8B job and the 512-token chunked-prefill jobs (AWQ, fp8)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/prof_r03d
rm -rf $O; mkdir -p $O
run() {  # tag, bench args...
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python3 bench.py "$@" --steps 1 --warmup 1 --skip-cpu --no-plugin-surface > $O/$tag.json 2> $O/$tag.err
  python3 scripts/prof_summary.py $O/$tag > gpurun_out/r03d_${tag}_kernel_trace_summary.txt
  find $O/$tag -name '*.csv' -size +1M -delete
  head -n 12 gpurun_out/r03d_${tag}_kernel_trace_summary.txt | cut -c1-70,111-200
}
run quant_fp8 --quant fp8
run chunk512_awq --chunk-tokens 512
run chunk512_fp8 --quant fp8 --chunk-tokens 512
