#!/bin/bash
# kernel stats of stage 1 at 2000 x 1000 x 64 for value-kernel variants (GPU box, repo root)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/valprof
mkdir -p $OUT
cd /tmp
for v in ${@:-1 2}; do
  export TPHIP_VALUE_COLS=$v
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c$v -- python3 $R/tools/stage1_timing.py 2000 1000 64 > $OUT/c$v.log 2>&1 || { echo "c$v failed"; exit 1; }
  cp $(ls $OUT/c$v/*/*kernel_stats.csv | head -1) $OUT/c${v}_kernel_stats.csv
  rm -rf $OUT/c$v
  head -8 $OUT/c${v}_kernel_stats.csv | cut -c1-150
done
