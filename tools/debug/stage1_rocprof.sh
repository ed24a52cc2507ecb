#!/bin/bash
# rocprofv3 kernel stats of stage 1 on three shapes (run on the GPU box from the repo root); summaries -> gpurun_out/s1prof/
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/s1prof
mkdir -p $OUT
cd /tmp
for shape in "8 20000 64" "1000 500 16" "2000 1000 64"; do
  tag=$(echo $shape | tr ' ' x)
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $R/tools/stage1_timing.py $shape > $OUT/$tag.log 2>&1 || { echo "$tag failed"; exit 1; }
  cp $(ls $OUT/$tag/*/*kernel_stats.csv | head -1) $OUT/${tag}_kernel_stats.csv
  tail -9 $OUT/$tag.log
  rm -rf $OUT/$tag
done
