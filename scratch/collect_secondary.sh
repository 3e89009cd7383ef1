#!/bin/bash
# rocprofv3 kernel statistics of the secondary ("next"-row) kernels, from their diagnostic drivers.  Run on the GPU box from the repo root.
TAG=${1:-r01x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/sec_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$ROOT
run() {  # name, script + args
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o s -- python3 "$@" > $OUT/$name.log 2> $OUT/$name.err
  cp $OUT/$name/s_kernel_stats.csv $ROOT/gpurun_out/${TAG}_${name}_kernel_stats.csv 2>/dev/null
  rm -rf $OUT/$name
}
run swap $ROOT/tools/diag_swap.py 20 4 4 256
run hsvd $ROOT/tools/diag_hsvd.py
run als $ROOT/tools/diag_als.py 8 16 64 2
cd $ROOT
head -4 gpurun_out/${TAG}_*_kernel_stats.csv | cut -c1-160
