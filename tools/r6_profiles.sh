#!/bin/bash
# Round-6 evidence in one GPU-box call: the bench line, rocprofv3 kernel stats of the headline workload and of C3a / C3b /
# C4, HBM traffic of the headline kernel (separate --pmc passes) and the MFMA counters of the C4 kernels.
# Everything lands under gpurun_out/r6f/ (copied to profiles/r6_* afterwards).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_N1.json 2> $OUT/bench_N1.err < /dev/null; echo "bench rc=$?"
python $R/bench.py --steps 20 --warmup 5 > $OUT/bench_N1_driver_args.json 2>> $OUT/bench_N1.err < /dev/null; echo "bench (driver args) rc=$?"
prof() { # tag, command...
  tag=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o run -- "$@" > $OUT/prof_$tag.log 2>&1 < /dev/null
  echo "prof $tag rc=$?"
  for f in $(find $OUT/prof_$tag -name "*kernel_stats.csv"); do cp $f $OUT/${tag}_kernel_stats.csv; done
  rm -rf $OUT/prof_$tag
}
BA="--no-cpu-baseline --no-minimize-wall --no-configs --min-timed-seconds 0.1"
prof de_M python $R/bench.py $BA
prof cmaes_c4 python $R/tools/run_c4.py 60
prof pso_c3 python $R/tools/run_pso_c3.py
prof cpso_c3b python $R/tools/run_cpso_c3b.py
# BASELINE config 5's shard shape (DE n=1024, P=16384) and the whole config on one GPU (P=131072) on the CURRENT binary
prof de_n1024_p16384 python $R/bench.py $BA --workload de_rosenbrock_n1024_p16384 --steps 200 --warmup 20
prof de_n1024_p131072 python $R/bench.py $BA --workload de_rosenbrock_n1024_p131072 --steps 40 --warmup 5
python $R/bench.py $BA --workload de_rosenbrock_n1024_p16384 --steps 200 --warmup 20 > $OUT/bench_c5_shard.json 2>> $OUT/bench_N1.err < /dev/null
python $R/bench.py $BA --workload de_rosenbrock_n1024_p131072 --steps 40 --warmup 5 > $OUT/bench_c5_full_1gpu.json 2>> $OUT/bench_N1.err < /dev/null
# the objective kernel alone (one-batch rows: eight lanes per row), off-grid shapes, wide rows
prof eval_r8 python $R/tools/eval_stream_ab.py arm
python $R/tools/bench_eval.py 2>&1 | grep -v amdgpu.ids > $OUT/eval_kernel.txt
python $R/tools/bench_shapes.py 2>&1 | grep -v amdgpu.ids > $OUT/shapes.txt
python $R/tools/bench_wide.py 2>&1 | grep -v amdgpu.ids > $OUT/wide_rows.txt
python $R/tools/de_occupancy_ab.py 2>&1 | grep -v amdgpu.ids > $OUT/de_population_sizes.txt
(hipcc --offload-arch=gfx950 -O3 $R/tools/probes/de_gather_probe.cpp -o /tmp/dgp && /tmp/dgp) > $OUT/de_gather_probe.txt 2>&1
pmc() { # tag counters -- command
  tag=$1; shift; ctr=$1; shift
  timeout 400 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$tag -o run -- "$@" > $OUT/pmc_$tag.log 2>&1 < /dev/null
  echo "pmc $tag rc=$?"
}
pmc de_M_fetch FETCH_SIZE python $R/bench.py $BA --kernel-timing-launches 50
pmc de_M_write WRITE_SIZE python $R/bench.py $BA --kernel-timing-launches 50
pmc c5_fetch FETCH_SIZE python $R/bench.py $BA --workload de_rosenbrock_n1024_p16384 --steps 100 --warmup 10 --kernel-timing-launches 50
pmc c5_write WRITE_SIZE python $R/bench.py $BA --workload de_rosenbrock_n1024_p16384 --steps 100 --warmup 10 --kernel-timing-launches 50
pmc de_M_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" python $R/bench.py $BA --kernel-timing-launches 50
pmc c4_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES SQ_INSTS_VALU" python $R/tools/run_c4.py 12
# the objective kernel on one-batch rows: is it instruction issue (round 6's reading) or memory?  the same counters for the
# one-visit kernel with 16 / 32 / 64 lanes per row (SX_EVAL_R8=0) and for eight lanes per row, + the bytes fetched
SQC="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
SX_EVAL_R8=0 pmc eval_old_sq "$SQC" python $R/tools/eval_stream_ab.py arm rosenbrock:128:1048576
pmc eval_r8_sq "$SQC" python $R/tools/eval_stream_ab.py arm rosenbrock:128:1048576
pmc eval_r8_fetch FETCH_SIZE python $R/tools/eval_stream_ab.py arm rosenbrock:128:1048576
python - <<PY
import collections, csv, glob, json
out = {}
for tag in ("de_M_fetch", "de_M_write", "de_M_sq", "c5_fetch", "c5_write", "c4_mfma", "eval_old_sq", "eval_r8_sq", "eval_r8_fetch"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in agg.items():
        if not any(s in k for s in ("de_generation", "eigh_round", "eigh_gemm", "cma_gemm", "eval_kernel", "eval_r8_kernel")):
            continue
        pre = "c5:" if tag.startswith("c5") else ("eval_old:" if tag.startswith("eval_old") else "eval_r8:" if tag.startswith("eval_r8") else "")
        for c, v in d.items():
            out.setdefault(pre + k, {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
def mean(sub, ctr, pre=""):
    for k, d in out.items():
        if k.startswith("c5:") != (pre == "c5:") or k.startswith("eval_"):
            continue
        if sub in k and ctr in d:
            return d[ctr]["mean"], d[ctr]["n"]
    return None, 0
lat = {}
f, nf = mean("de_generation", "FETCH_SIZE"); w, nw = mean("de_generation", "WRITE_SIZE")
if f is not None and w is not None:
    lat["de_rosenbrock_n128_p4096"] = {
        "kernel": "de_generation_kernel<rosenbrock, philox, chained, LPR=32, FULL, NFIX=128, STRAT=best1bin>",
        "source": "rocprofv3 --pmc (separate passes: FETCH_SIZE, WRITE_SIZE, SQ counters), tools/r6_profiles.sh, per-dispatch means over %d launches (profiles/r6_pmc_summary.json); FETCH_SIZE x2 per the gfx950 calibration (tools/calib.sh, MI355X_MICROARCH.md section HBM)" % nf,
        "fetch_size_kb": f, "write_size_kb": w, "fetch_correction": 2.0,
        "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0, "algorithmic_bytes_per_launch": 16842752}
    sq = {c: mean("de_generation", c)[0] for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY")}
    if sq["SQ_WAVES"]:
        lat["de_rosenbrock_n128_p4096"]["per_wave"] = {k: v / sq["SQ_WAVES"] for k, v in sq.items() if v is not None and k != "SQ_WAVES"}
        lat["de_rosenbrock_n128_p4096"]["SQ_WAVES"] = sq["SQ_WAVES"]
f, nf = mean("de_generation", "FETCH_SIZE", "c5:"); w, nw = mean("de_generation", "WRITE_SIZE", "c5:")
if f is not None and w is not None:
    lat["de_rosenbrock_n1024_p16384"] = {"fetch_size_kb": f, "write_size_kb": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                                         "source": "as above (c5_fetch / c5_write passes), %d launches" % nf}
import subprocess
lat["_commit"] = subprocess.run(["git", "-C", "$R", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "see profiles/r6_commit.txt"
lat["_round"] = 5
json.dump(lat, open("$OUT/pmc_latest.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3500])
PY
rm -rf $OUT/pmc_*/
tail -c 400 $OUT/bench_N1.json; echo
for t in de_M cmaes_c4 pso_c3 cpso_c3b de_n1024_p16384 de_n1024_p131072 eval_r8; do echo "== $t"; head -8 $OUT/${t}_kernel_stats.csv | cut -c1-230; done
