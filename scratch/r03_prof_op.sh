#!/bin/bash
# rocprofv3 kernel stats of one --op line:  bash scratch/r03_prof_op.sh <op> <batch> <tag>
OP=$1; B=$2; TAG=$3
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_${OP}_$B
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $ROOT/bench.py --op $OP --steps 9 --warmup 3 --batch $B > $OUT/line.json 2> $OUT/err.txt
cd $ROOT
ST=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp $ST gpurun_out/${TAG}_op_${OP}_b${B}_kernel_stats.csv
head -3 $ST | cut -c1-200
tail -1 $OUT/line.json | cut -c1-300
rm -rf $OUT/stats
