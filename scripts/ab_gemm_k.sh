#!/bin/bash
# like ab_gemm.sh, but prints only the per-layer totals and the gate_up line: scripts/ab_gemm_k.sh M lib...
M=$1; shift
for r in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then
      python scripts/bench_gemm.py $M 2>/dev/null | grep "gate_up\|total" | sed "s|^|[default r$r] |"
    else
      MI355X_HOTPATH_LIB=$PWD/$lib python scripts/bench_gemm.py $M 2>/dev/null | grep "gate_up\|total" | sed "s|^|[$lib r$r] |"
    fi
  done
done
