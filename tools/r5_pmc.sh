#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CN_OVERLAP_WGRAD=0
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
   --kernel-trace --output-format csv -d $O/m -o m -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 $R/tools/pmc_mfma.py $O/m/m_counter_collection.csv $O/pmc_mfma_f32.json | head -8
rocprofv3 -L 2>/dev/null | grep -i -o 'SQ_[A-Z_]*LDS[A-Z_]*' | sort -u | head -20
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
   --kernel-trace --output-format csv -d $O/l -o l -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - <<PY
import csv,re
from collections import defaultdict
agg=defaultdict(lambda: defaultdict(float))
try:
    for r in csv.DictReader(open('$O/l/l_counter_collection.csv')):
        k=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]).split("(")[0].replace("void ","").strip()
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k in ("cn_wgrad_vec3_kernel<1>","cn_conv_igemm_vec_kernel<4, 1, 5, 1, 1>","cn_wgrad_vec3_kernel<2>"):
        v=agg.get(k)
        if v: print(k, {n:round(x) for n,x in v.items()})
except Exception as e: print('lds pass failed', e)
PY
rm -rf $O/m $O/l
