#!/bin/bash
# extra SQ counters of k_compress (one pass per set; --kernel-trace only, never with other trace domains)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_extra; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-single --no-verify"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_SMEM" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/p$i -o p -- $CMD > $OUT/p$i.line 2> $OUT/p$i.err || echo "pass $i failed"
done
cd $ROOT
python3 - <<'PY'
import csv, glob, json
res={}
for f in glob.glob("gpurun_out/pmc_extra/p*/**/*counter_collection.csv", recursive=True):
    acc={}
    for r in csv.DictReader(open(f)):
        if "k_compress" not in r["Kernel_Name"]: continue
        acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
    for k,v in acc.items(): res[k]=sum(v)/len(v)
json.dump(res, open("gpurun_out/pmc_extra.json","w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/p*/
