#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/prof_r03b
rm -rf $O; mkdir -p $O
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_$grp -- python3 scripts/prof_r03b.py > $O/pmc_$grp.log 2>&1
done
python3 scripts/prof_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > gpurun_out/r03b_pmc_summary.txt
find $O -name '*.csv' -size +4M -delete
grep -E "^==|mi355x" gpurun_out/r03b_pmc_summary.txt | cut -c1-70,111-260 | head -40
