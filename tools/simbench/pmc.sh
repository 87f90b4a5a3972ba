#!/bin/bash
# counter passes over ./sim_bench<variant> (args after the variant are passed on)
cd "$(dirname "$0")"
V=$1; shift
OUT=../../gpurun_out/sbpmc$V
mkdir -p $OUT; export TMPDIR=/tmp
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_IFETCH SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/p$i -- ./sim_bench$V 3 "$@" > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'filip_sim' in k: agg[(k,r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(k[0],k[1],"%.4g"%(sum(v)/len(v)))
PY
