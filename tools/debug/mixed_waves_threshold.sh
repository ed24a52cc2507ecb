#!/bin/bash
# one wave per SIMD or two in the mixed-loci mode, on batches around the switch
mkdir -p gpurun_out/mixw
for wl in "C2 --loci 2000" "C2 --loci 3000" "C2 --loci 3900" "C4 --loci 300"; do
  for w in 1024 2048; do
    tag=$(echo $wl | tr -d ' -')_w$w
    TPHIP_SITE_MIXED=1 TPHIP_SITE_WAVES=$w timeout -k 10 280 python bench.py --workload $wl --steps 20 --warmup 3 --cpu-seconds 0 --stage1-loci 0 --no-single-gpu-check > gpurun_out/mixw/$tag.json 2> gpurun_out/mixw/$tag.err || { echo "$tag failed"; tail -3 gpurun_out/mixw/$tag.err; exit 1; }
    python - $tag <<'PY'
import json, sys
d = json.loads(open("gpurun_out/mixw/%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "ms/step %.3f site %.3f evals %d" % (d["ms_per_step"], d["stages_ms"]["site_rate_kernel"], d["fp64"]["evals_per_launch"]), flush=True)
PY
  done
done
for w in 1024 2048; do TPHIP_SITE_MIXED=1 TPHIP_SITE_WAVES=$w timeout -k 10 280 python bench.py --workload R1 --cpu-seconds 0 2>/dev/null | grep -o "\"dedup\": {[^}]*}" | sed "s/^/R1 waves $w /"; done
