#!/bin/bash
# A/B of libmi355x_hotpath.so builds on the decode GEMM shapes: scripts/ab_gemm.sh M lib1.so lib2.so ...
# ("default" = the in-tree build); 2 alternating rounds, one child process per run.
M=$1; shift
for r in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then
      python scripts/bench_gemm.py $M | sed "s|^|[default r$r] |"
    else
      MI355X_HOTPATH_LIB=$PWD/$lib python scripts/bench_gemm.py $M | sed "s|^|[$lib r$r] |"
    fi
  done
done
