#!/bin/bash
# final measurement pass of a round (run on the GPU box from the repo root): bench lines, rocprofv3 kernel stats, PMC,
# stage-1 kernel stats, command line with model averaging.  Everything lands in gpurun_out/final_<tag>/
set -u
TAG=${1:-x}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final_$TAG
mkdir -p $OUT
cd $R
echo "== default bench"; timeout -k 10 500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo "default bench failed"; tail -5 $OUT/bench_default.err; exit 1; }
tail -c 2500 $OUT/bench_default.json
for w in C2 C5; do
  echo "== $w"; timeout -k 10 300 python bench.py --workload $w --steps 10 --warmup 3 --cpu-seconds 0 --stage1-loci 0 > $OUT/bench_$w.json 2> $OUT/bench_$w.err || { echo "$w failed"; exit 1; }
done
echo "== C4 whole, one rank under torchrun"
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload C4 --steps 5 --warmup 2 --cpu-seconds 0 --stage1-loci 0 > $OUT/bench_C4_1rank.json 2> $OUT/bench_C4.err || { echo "C4 failed"; exit 1; }
echo "== rocprofv3 kernel stats, C3"
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --workload C3 --steps 5 --warmup 2 --cpu-seconds 0 --stage1-loci 0 > $OUT/c3_bench_under_rocprof.json 2> $OUT/prof.err ) || { echo "rocprof failed"; exit 1; }
cp $(ls $OUT/prof/*/*kernel_stats.csv | head -1) $OUT/c3_kernel_stats.csv && rm -rf $OUT/prof
head -6 $OUT/c3_kernel_stats.csv | cut -c1-160
echo "== PMC"
bash tools/pmc.sh final_$TAG --workload C3 --stage1-loci 0 > $OUT/pmc.log 2>&1; cp gpurun_out/pmc_final_$TAG/summary.json $OUT/c3_pmc.json
echo "== stage 1 kernel stats"
bash tools/debug/stage1_rocprof.sh > $OUT/stage1_rocprof.log 2>&1; cp gpurun_out/s1prof/*_kernel_stats.csv gpurun_out/s1prof/*.log $OUT/ 2>/dev/null
grep -E "general model:|202 models" $OUT/stage1_rocprof.log
echo "== stage 1 timings without the profiler"
for shape in "8 20000 64" "1000 500 16" "2000 1000 64"; do timeout -k 10 200 python tools/stage1_timing.py $shape 2>&1 | grep -E "^loci|general model:|202 models"; done | tee $OUT/stage1_timing.log
echo "== command line with model averaging, 4000 loci"
timeout -k 10 400 python tools/e2e_cli_timing.py 4000 1000 64 --model-averaging --multiprocessing 2>&1 | grep -E "CLI end|stages" | tee $OUT/e2e_model_averaging.log
echo "== host-pointer path"
timeout -k 10 200 python tools/pcie_inclusive.py 2>&1 | grep -v amdgpu | tail -6 | tee $OUT/pcie.log
