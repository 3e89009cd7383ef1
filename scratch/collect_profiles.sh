#!/bin/bash
# Collects the rocprofv3 evidence for one round tag:  bash scratch/collect_profiles.sh r01i   (run on the GPU box from the repo root)
set -e
TAG=${1:-r01x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-single"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $CMD > $OUT/line.json 2> $OUT/stats.err
for C in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/pmc_$N -o p -- $CMD > $OUT/pmc_$N.line 2> $OUT/pmc_$N.err
done
cd $ROOT
find $OUT -type f | head -30
tail -3 $OUT/stats.err
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, os
out, tag = sys.argv[1], sys.argv[2]
res = {"k_compress": {}}
st = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)
if st:
    rows = list(csv.DictReader(open(st[0])))
    open(os.path.join(os.path.dirname(out), f"{tag}_bench_b1024_kernel_stats.csv"), "w").write(open(st[0]).read())
    for r in rows:
        if "k_compress" in r.get("Name", ""):
            res["k_compress"]["avg_ns"] = float(r["AverageNs"]); res["k_compress"]["calls"] = int(r["Calls"])
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    acc = {}
    for r in csv.DictReader(open(f)):
        if "k_compress" not in r["Kernel_Name"]: continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res["k_compress"][k + "_per_launch"] = sum(v) / len(v)
        res["k_compress"]["launches"] = len(v)
json.dump(res, open(os.path.join(os.path.dirname(out), f"{tag}_pmc_raw.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
print(open(out + "/line.json").read()[:400])
PY
rm -rf $OUT/stats $OUT/pmc_*/   # raw traces are large; the summaries above are what is kept
