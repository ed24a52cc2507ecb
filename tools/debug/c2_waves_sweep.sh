#!/bin/bash
# C2 step time of the mixed-loci mode against the number of waves (= equal shares of the work list), one box.
mkdir -p gpurun_out/c2sweep
for w in ${WAVES:-512 768 1024 1280 1536 2048 3072 4096}; do
  TPHIP_SITE_WAVES=$w timeout -k 10 200 python bench.py --workload ${WORKLOAD:-C2} --steps 30 --warmup 5 --cpu-seconds 0 --stage1-loci 0 > gpurun_out/c2sweep/w$w.json 2> gpurun_out/c2sweep/w$w.err || exit 1
  python - $w <<'PY'
import json, sys
d = json.loads(open("gpurun_out/c2sweep/w%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("waves", sys.argv[1], "ms/step %.4f site %.4f evals %d" % (d["ms_per_step"], d["stages_ms"]["site_rate_kernel"], d["fp64"]["evals_per_launch"]), flush=True)
PY
done
