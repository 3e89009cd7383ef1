#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_ic; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-single --no-verify"
rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $OUT/p1 -o p -- $CMD > $OUT/p1.line 2> $OUT/p1.err || echo fail1
rocprofv3 --kernel-trace --output-format csv --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM -d $OUT/p2 -o p -- $CMD > $OUT/p2.line 2> $OUT/p2.err || echo fail2
cd $ROOT
python3 - <<'PY'
import csv, glob, json
res={}
for f in glob.glob("gpurun_out/pmc_ic/p*/**/*counter_collection.csv", recursive=True):
    acc={}
    for r in csv.DictReader(open(f)):
        if "k_compress" not in r["Kernel_Name"]: continue
        acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
    for k,v in acc.items(): res[k]=sum(v)/len(v)
print(json.dumps(res, indent=1))
PY
tail -3 $OUT/p1.err
rm -rf $OUT/p*/
