#!/bin/bash
# usage: tools/pmc.sh <kernel-name-substring> <python script + args...>   (run on the GPU box via gpurun)
export TMPDIR=/tmp; R=$PWD; pat=$1; shift
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1)); (cd /tmp && rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $R/gpurun_out/pmc_$i -- python3 $R/"$@" > /dev/null 2>&1)
  f=$(ls $R/gpurun_out/pmc_$i/*/*counter_collection.csv | head -1)
  python3 - "$f" "$pat" <<PY
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
agg=collections.defaultdict(list)
for r in rows: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(f"{k:28s} {sum(v[-3:])/min(3,len(v)):16.0f}")
PY
done
