#!/bin/bash
# A/B of library builds on any bench script: scripts/ab_lib.sh "python scripts/X.py args" lib1.so lib2.so ...  ("default" = in-tree)
cmd=$1; shift
for r in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then $cmd 2>/dev/null | sed "s|^|[default r$r] |"
    else MI355X_HOTPATH_LIB=$PWD/$lib $cmd 2>/dev/null | sed "s|^|[$lib r$r] |"; fi
  done
done
