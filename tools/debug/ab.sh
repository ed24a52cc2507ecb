#!/bin/bash
# same-box A/B of two library builds: tools/debug/ab.sh <libA.so> <libB.so> [workloads...]
A=$1; B=$2; shift 2
for w in "${@:-C3}"; do
  for rep in 1 2; do
    for lib in $A $B; do
      TPHIP_LIB=$lib timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 3 --cpu-seconds 0 --stage1-loci 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$w $(basename $lib): site %.4f ms step %.4f ms' % (d['stages_ms']['site_rate_kernel'], d['ms_per_step']))"
    done
  done
done
