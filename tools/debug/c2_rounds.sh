#!/bin/bash
# needs a build with TPHIP_HIPCC_FLAGS=-DTPHIP_SITE_TRACE_ROUNDS: rounds and time per wave of site_rate_kernel on C2
export TPHIP_SITE_TRACE_ROUNDS=1
for cfg in "TPHIP_SITE_MIXED=0" "TPHIP_SITE_MIXED=1" "TPHIP_SITE_MIXED=1 TPHIP_SITE_WAVES=1536" "TPHIP_SITE_MIXED=1 TPHIP_SITE_WAVES=1024" "TPHIP_SITE_MIXED=1 TPHIP_SITE_WAVES=1024 TPHIP_SITE_NO_REORDER=1"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python bench.py --workload ${WORKLOAD:-C2} --steps 5 --warmup 2 --cpu-seconds 0 --stage1-loci 0 2>&1 | grep -o "site_rate_kernel: .*\|\"site_rate_kernel\": [0-9.]*" | tail -2
done
