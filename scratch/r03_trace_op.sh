#!/bin/bash
# per-dispatch kernel durations of one --op line:  bash scratch/r03_trace_op.sh <op> <batch>
OP=$1; B=$2
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/trace_${OP}_$B
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o s -- python3 $ROOT/bench.py --op $OP --steps 3 --warmup 1 --batch $B > $OUT/line.json 2> $OUT/err.txt
cd $ROOT
F=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-12:]:
    print(r["Kernel_Name"][:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "ms", "grid", r.get("Grid_Size_X", r.get("Grid_Size", "")), "wg", r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
PY
rm -rf $OUT/t
