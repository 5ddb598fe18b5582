#!/bin/bash
# Round-3 profile passes on the GPU box (writes under gpurun_out/prof_r03/, summaries to gpurun_out/r03_*.txt).
# kernel trace + stats of the whole bench job; then one PMC group per pass over scripts/prof_r03.py.
set -x
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/prof_r03
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 bench.py --steps 1 --warmup 1 --skip-cpu --no-plugin-surface > $O/bench_under_rocprof.json 2> $O/bench.err
python3 scripts/prof_summary.py $O/bench > gpurun_out/r03_bench_kernel_trace_summary.txt
find $O/bench -name '*kernel_stats.csv' -exec cp {} gpurun_out/r03_bench_kernel_stats.csv \;
find $O/bench -name '*.csv' -size +1M -delete      # (gpurun copies back at most 64 MiB)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rank8 -- python3 bench.py --model llama-3-70b --tp-rank-of 8 --steps 1 --warmup 1 > $O/rank8_under_rocprof.json 2> $O/rank8.err
python3 scripts/prof_summary.py $O/rank8 > gpurun_out/r03_rank_of_8_70b_fp8_kernel_trace_summary.txt
find $O/rank8 -name '*kernel_stats.csv' -exec cp {} gpurun_out/r03_rank_of_8_70b_fp8_kernel_stats.csv \;
find $O/rank8 -name '*.csv' -size +1M -delete
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_$tag -- python3 scripts/prof_r03.py > $O/pmc_$tag.log 2>&1
done
python3 scripts/prof_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_WAVE_CYCLES > gpurun_out/r03_pmc_summary.txt
find $O -name '*.csv' -size +4M -delete
for f in $O/*.log; do echo "== $f"; tail -n 3 $f; done
du -sh gpurun_out
