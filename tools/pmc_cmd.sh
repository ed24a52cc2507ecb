#!/bin/bash
# usage: tools/pmc_cmd.sh <tag> <script.py> [args...]   (run on the GPU box from the repo root)
# PMC passes (one counter group per run, as the microarchitecture guide prescribes) over any python script;
# per-kernel averages of every tphip kernel go to gpurun_out/pmc_<tag>/summary.json
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
SCRIPT=$R/$1; shift
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
         "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_VALU_TRANS_F64" \
         "GRBM_GUI_ACTIVE SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH SQ_IFETCH SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- python3 $SCRIPT "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections, json
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "tphip::" in kn:
            k = kn.split("tphip::")[1].split("(")[0]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
out = {k: {c: tot[k][c] / max(1, n[k][c]) for c in sorted(tot[k])} for k in tot}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for k in out:
    print(k, json.dumps(out[k]))
PY
rm -rf $OUT/p[0-9]*
