#!/bin/bash
# small batches of 64-taxon trees (mixed-loci mode by default) against the slice mode
mkdir -p gpurun_out/mixm
for wl in "C4 --loci 300" "C4 --loci 1000" "C4 --loci 1900" "C2 --loci 3000"; do
  for m in 0 1; do
    tag=$(echo $wl | tr -d ' -')_m$m
    TPHIP_SITE_MIXED=$m timeout -k 10 280 python bench.py --workload $wl --steps 20 --warmup 3 --cpu-seconds 0 --stage1-loci 0 --no-single-gpu-check > gpurun_out/mixm/$tag.json 2> gpurun_out/mixm/$tag.err || { echo "$tag failed"; tail -3 gpurun_out/mixm/$tag.err; exit 1; }
    python - $tag <<'PY'
import json, sys
d = json.loads(open("gpurun_out/mixm/%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "ms/step %.3f site %.3f evals %d sha %s" % (d["ms_per_step"], d["stages_ms"]["site_rate_kernel"], d["fp64"]["evals_per_launch"], d["table_sha256"][:12]), flush=True)
PY
  done
done
