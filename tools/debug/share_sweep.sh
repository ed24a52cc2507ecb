#!/bin/bash
# share-size sweep of the persistent site-rate grid: tools/debug/share_sweep.sh WORKLOAD "mult frac" ...
W=$1; shift
for cfg in "$@"; do
  set -- $cfg
  TPHIP_SITE_GRID_MULT=$1 TPHIP_SITE_FIRST_FRACTION=$2 timeout -k 10 200 python bench.py --workload $W --steps 10 --warmup 3 --cpu-seconds 0 --stage1-loci 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$W mult $1 first $2: site %.3f ms, step %.3f ms' % (d['stages_ms']['site_rate_kernel'], d['ms_per_step']))"
done
