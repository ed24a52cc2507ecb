#!/bin/bash
# same-box A/B of library builds on the engine stage 1: tools/debug/ab_stage1.sh "L S N" lib1.so lib2.so ...
shape=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    echo "$(basename $lib): $(TPHIP_LIB=$lib timeout -k 10 200 python tools/stage1_timing.py $shape cuda 2>/dev/null | tail -1 | cut -c1-110)"
  done
done
