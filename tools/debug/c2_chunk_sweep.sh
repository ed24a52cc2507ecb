#!/bin/bash
# C2 step time against the slice length of the small-batch mode (and the slow-first order off), one box.
mkdir -p gpurun_out/c2sweep
for c in 512 256 192 128 96; do
  TPHIP_SITE_CHUNK=$c timeout -k 10 200 python bench.py --workload C2 --steps 30 --warmup 5 --cpu-seconds 0 --stage1-loci 0 > gpurun_out/c2sweep/c$c.json 2> gpurun_out/c2sweep/c$c.err || exit 1
  python - $c <<'PY'
import json, sys
d = json.loads(open("gpurun_out/c2sweep/c%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("chunk", sys.argv[1], "ms/step %.4f site %.4f evals %d" % (d["ms_per_step"], d["stages_ms"]["site_rate_kernel"], d["fp64"]["evals_per_launch"]), flush=True)
PY
done
TPHIP_SITE_NO_REORDER=1 TPHIP_SITE_CHUNK=256 timeout -k 10 200 python bench.py --workload C2 --steps 30 --warmup 5 --cpu-seconds 0 --stage1-loci 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no reorder chunk 256: ms/step %.4f site %.4f' % (d['ms_per_step'], d['stages_ms']['site_rate_kernel']))"
