#!/bin/bash
# rocprofv3 kernel stats of the ENGINE stage 1 (tphip_stage1_fit) on the target shapes; summaries -> gpurun_out/<tag>/
# usage: tools/debug/s1_engine_rocprof.sh TAG "L C N" ["L C N" ...]   (run on the GPU box from the repo root)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
for shape in "$@"; do
  tag=$(echo $shape | tr ' ' x)
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $R/tools/stage1_timing.py $shape reps=2 > $OUT/$tag.log 2>&1 || { echo "$tag failed"; tail -5 $OUT/$tag.log; exit 1; }
  cp $(ls $OUT/$tag/*/*kernel_stats.csv | head -1) $OUT/${tag}_kernel_stats.csv
  tail -2 $OUT/$tag.log
  rm -rf $OUT/$tag
done
