#!/bin/bash
# the measurement pass behind profiles/r03_b_*: run on the GPU box from the repo root, results under gpurun_out/r03_final/
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_final
mkdir -p $OUT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo "default bench failed"; tail -3 $OUT/bench_default.err; }
python bench.py --workload C2 --steps 20 --warmup 5 --cpu-seconds 0 --stage1-loci 0 > $OUT/c2.json 2>/dev/null
python bench.py --workload C5 --steps 3 --warmup 1 --cpu-seconds 0 --stage1-loci 0 > $OUT/c5_full_1gpu.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --workload C4 --steps 5 --warmup 2 --cpu-seconds 0 --stage1-loci 0 2>/dev/null | grep "^{" > $OUT/c4_full_1rank_torchrun.json
python bench.py --workload R1 > $OUT/r1.json 2>/dev/null
( cd /tmp && timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --stage1-loci 0 > $OUT/c3_bench_under_rocprof.json 2> $OUT/prof.err )
cp $(ls $OUT/prof/*/*kernel_stats.csv | head -1) $OUT/c3_kernel_stats.csv; rm -rf $OUT/prof
for shape in "1000 500 16" "2000 1000 64" "16 50000 64" "200 2000 256"; do
  tag=$(echo $shape | tr ' ' x)
  python tools/stage1_timing.py $shape > $OUT/stage1_$tag.txt 2>&1; tail -1 $OUT/stage1_$tag.txt | cut -c1-200
done
for f in bench_default c2 c5_full_1gpu c4_full_1rank_torchrun r1; do python - <<PY
import json
try:
    d = json.loads([l for l in open("$OUT/$f.json") if l.startswith("{")][0])
    print("$f", round(d["ms_per_step"], 4), "%.4g" % d["value"], d.get("fp64", {}).get("frac"), d.get("stage1", {}).get("columns_per_s") if isinstance(d.get("stage1"), dict) else None)
except Exception as e:
    print("$f FAILED", e)
PY
done
