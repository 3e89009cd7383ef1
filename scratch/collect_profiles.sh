#!/bin/bash
# Collects the rocprofv3 evidence for one round tag:  bash scratch/collect_profiles.sh r02a   (run on the GPU box from the repo root)
#  1. rocprofv3 --kernel-trace --stats around the default bench command        -> gpurun_out/<tag>_bench_b1024_kernel_stats.csv
#  2. one --pmc pass per counter set (never combined with the trace domains)   -> gpurun_out/<tag>_pmc.json
#  3. gpurun_out/k_compress_traffic.json = the UNTAGGED file bench.py reads `roofline.traffic` from (copy it, and the tagged
#     summaries, to profiles/ and commit them; nothing here is ever edited by hand)
#  OPS=1: also a --kernel-trace --stats pass + FETCH/WRITE passes for each `bench.py --op X` line -> gpurun_out/<tag>_op_<X>.json
set -e
TAG=${1:-r02x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-single --no-verify --no-c2 --no-ops"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $CMD > $OUT/line.json 2> $OUT/stats.err
for C in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/pmc_$N -o p -- $CMD > $OUT/pmc_$N.line 2> $OUT/pmc_$N.err
done
if [ -n "$OPS" ]; then
  for OP in apply hadamard add scale dot orthogonalize; do
    OCMD="python3 $ROOT/bench.py --op $OP --steps 5 --warmup 2 --batch ${OPBATCH:-256}"
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/op_$OP/stats -o s -- $OCMD > $OUT/op_$OP.line 2> $OUT/op_$OP.err
    for C in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/op_$OP/pmc_$C -o p -- $OCMD > /dev/null 2>> $OUT/op_$OP.err
    done
  done
fi
cd $ROOT
tail -3 $OUT/stats.err
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, os, subprocess
out, tag = sys.argv[1], sys.argv[2]
dst = os.path.dirname(out)


def kernel_stats(d, name):
    st = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    if not st:
        return None, None
    rows = list(csv.DictReader(open(st[0])))
    for r in rows:
        if name in r.get("Name", ""):
            return st[0], {"avg_ns": float(r["AverageNs"]), "calls": int(r["Calls"])}
    return st[0], None


def counters(pattern, name):
    res = {}
    for f in glob.glob(pattern, recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if name not in r["Kernel_Name"]:
                continue
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            res[k + "_per_launch"] = sum(v) / len(v)
            res["launches"] = len(v)
    return res


res = {"tag": tag, "k_compress": {}}
path, ks = kernel_stats(out + "/stats", "k_compress")
if path:
    open(os.path.join(dst, f"{tag}_bench_b1024_kernel_stats.csv"), "w").write(open(path).read())
if ks:
    res["k_compress"].update(ks)
res["k_compress"].update(counters(out + "/pmc_*/**/*counter_collection.csv", "k_compress"))
line = {}
try:
    line = json.loads([ln for ln in open(out + "/line.json") if ln.startswith("{")][-1])
except Exception:
    pass
res["bench_line_under_profiler"] = line
kc = res["k_compress"]
if "SQ_BUSY_CYCLES_per_launch" in kc and "avg_ns" in kc:
    # SIMD-cycles available = 4 SIMDs x 256 CUs x kernel cycles (2.4 GHz); the SQ counters are summed over SEs/XCDs
    simd_cycles = 4 * 256 * kc["avg_ns"] * 2.4
    res["derived"] = {"kernel_ms": kc["avg_ns"] / 1e6, "simd_cycles_available": simd_cycles,
                      "mfma_pipe_busy_frac": kc.get("SQ_VALU_MFMA_BUSY_CYCLES_per_launch", 0) / simd_cycles,
                      "valu_busy_frac": 4 * kc.get("SQ_ACTIVE_INST_VALU_per_launch", 0) / simd_cycles}
json.dump(res, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
# the untagged traffic file bench.py reads.  FETCH_SIZE / WRITE_SIZE are in KB (rocprofv3 derived metrics).  gfx950: FETCH_SIZE
# reports half the bytes of wide coalesced reads and is uncalibrated for the 8-byte-per-lane strided reads that dominate
# k_compress (MI355X_MICROARCH.md, HBM): traffic = WRITE + FETCH_raw is the lower figure, WRITE + 2 FETCH_raw the upper one.
if "FETCH_SIZE_per_launch" in kc and "WRITE_SIZE_per_launch" in kc:
    cfg = line.get("config", {})
    w, f = kc["WRITE_SIZE_per_launch"] * 1024.0, kc["FETCH_SIZE_per_launch"] * 1024.0
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    json.dump({"kernel": "k_compress", "tag": tag, "commit": commit or None, "d": cfg.get("d"), "rank": cfg.get("rank"), "batch": cfg.get("batch_per_gpu"),
               "command": "bench.py --steps 3 --warmup 1 --no-cpu --no-single --no-verify --no-c2 --no-ops under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)",
               "write_bytes_per_launch": w, "fetch_bytes_per_launch_raw": f,
               "traffic_bytes_per_launch": w + f, "traffic_bytes_per_launch_upper": w + 2 * f,
               "note": "raw FETCH_SIZE under-counts wide coalesced reads by 1/2 on gfx950 and is uncalibrated for 8-byte strided reads: "
                       "traffic_bytes_per_launch = WRITE + FETCH_raw (lower), _upper = WRITE + 2 FETCH_raw"},
              open(os.path.join(dst, "k_compress_traffic.json"), "w"), indent=1)
for opline in sorted(glob.glob(out + "/op_*.line")):
    op = os.path.basename(opline)[3:-5]
    kname = {"apply": "k_apply", "hadamard": "k_hadamard", "add": "k_add", "scale": "k_scale", "dot": "k_dot_fused", "orthogonalize": "k_orthogonalize"}[op]
    rec = {"tag": tag, "op": op, "kernel": kname}
    try:
        rec["bench_line_under_profiler"] = json.loads([ln for ln in open(opline) if ln.startswith("{")][-1])
    except Exception:
        pass
    _, ks = kernel_stats(out + f"/op_{op}/stats", kname)
    if ks:
        rec.update(ks)
    if op == "orthogonalize":
        # three kernels since round 3 (ramp sites by one wave per train, tall sites by the 512-thread kernel, the 1024-thread kernel for the
        # left sweep / refused trains): per-kernel averages, and their sum as the figure to hold against bench.py's avg_launch_ms
        rec["kernels"] = {}
        for kn in ("k_ortho_ramp", "k_ortho512", "k_orthogonalize"):
            _, k2 = kernel_stats(out + f"/op_{op}/stats", kn)
            if k2:
                rec["kernels"][kn] = k2
        rec["kernel"] = " + ".join(rec["kernels"]) or kname
        rec["avg_ns"] = sum(v["avg_ns"] for v in rec["kernels"].values())
    if op == "orthogonalize":
        tot = {}
        for kn in rec.get("kernels", {}):
            for k, v in counters(out + f"/op_{op}/pmc_*/**/*counter_collection.csv", kn).items():
                if k.endswith("_per_launch"):
                    tot[k] = tot.get(k, 0.0) + v
        rec.update(tot)
    else:
        rec.update(counters(out + f"/op_{op}/pmc_*/**/*counter_collection.csv", kname))
    json.dump(rec, open(os.path.join(dst, f"{tag}_op_{op}.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
rm -rf $OUT/stats $OUT/pmc_*/ $OUT/op_*/   # raw traces are large; the summaries above are what is kept
