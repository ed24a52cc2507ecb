#!/bin/bash
# share-size sweep of the persistent site-rate grid on the whole C5 (run on the GPU box)
for cfg in "3 0.8" "3 0.9" "2 0.8" "2 0.9" "4 0.85" "1 0"; do
  set -- $cfg
  if [ "$1" = "1" ]; then export TPHIP_SITE_GRID_MULT=1; unset TPHIP_SITE_FIRST_FRACTION; else export TPHIP_SITE_GRID_MULT=$1 TPHIP_SITE_FIRST_FRACTION=$2; fi
  timeout -k 10 200 python bench.py --workload C5 --steps 3 --warmup 1 --cpu-seconds 0 --stage1-loci 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('mult $1 first $2: site %.2f ms, step %.2f ms' % (d['stages_ms']['site_rate_kernel'], d['ms_per_step']))"
done
