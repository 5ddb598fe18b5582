#!/bin/bash
# late-round-3 PMC passes (one counter group per pass) over scripts/prof_r03c.py -> gpurun_out/r03c_pmc_summary.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/prof_r03c
rm -rf $O; mkdir -p $O
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_$grp -- python3 scripts/prof_r03c.py > $O/pmc_$grp.log 2>&1
done
python3 scripts/prof_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > gpurun_out/r03c_pmc_summary.txt
find $O -name '*.csv' -size +4M -delete
for f in $O/*.log; do echo "== $f"; tail -n 2 $f; done
head -n 60 gpurun_out/r03c_pmc_summary.txt | cut -c1-200
