#!/bin/bash
# value-kernel variants on the three stage-1 shapes (run on the GPU box from the repo root)
for shape in "8 20000 64" "1000 500 16" "2000 1000 64"; do
  for v in ${@:-1 2}; do
    export TPHIP_VALUE_COLS=$v
    echo "== $shape cols $v"
    timeout -k 10 200 python tools/stage1_timing.py $shape 2>&1 | grep -E "general model:|202 models" || exit 1
  done
done
