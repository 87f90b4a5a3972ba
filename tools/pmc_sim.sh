#!/bin/bash
# Counter passes over the FILIP score kernel alone (tools/run_filip_fwd.py), printed per kernel.
set -o pipefail
OUT=gpurun_out/${1:-pmcsim}
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/p$i -- python3 tools/run_filip_fwd.py 3 unit > $OUT/p$i.log 2>&1 || { tail -20 $OUT/p$i.log; exit 1; }
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
