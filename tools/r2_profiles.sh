#!/bin/bash
# Round-2 evidence in one GPU-box call: the bench line, rocprofv3 kernel stats of the headline workload, its HBM
# traffic (separate --pmc passes), and kernel stats / MFMA counters of the eigensolver, the CMA-ES device loop and the
# other claimed paths.  Everything lands under gpurun_out/r2/ (copied to profiles/r2_* by hand afterwards).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_N1.json 2> $OUT/bench_N1.err < /dev/null; echo "bench rc=$?"
prof() { # tag, command...
  tag=$1; shift
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o run -- "$@" > $OUT/prof_$tag.log 2>&1 < /dev/null
  echo "prof $tag rc=$?"
  for f in $(find $OUT/prof_$tag -name "*kernel_stats.csv"); do cp $f $OUT/${tag}_kernel_stats.csv; done
  rm -rf $OUT/prof_$tag/*kernel_trace.csv  # (large)
}
prof de_M python $R/bench.py --no-cpu-baseline --no-minimize-wall
prof eigh python $R/tools/bench_eigh_jacobi.py 128 256 512 1024
prof cmaes_c4 python $R/tools/bench_c4.py 10 60
prof misc python $R/tools/bench_misc.py
pmc() { # tag counters -- command
  tag=$1; shift; ctr=$1; shift
  timeout 400 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$tag -o run -- "$@" > $OUT/pmc_$tag.log 2>&1 < /dev/null
  echo "pmc $tag rc=$?"
}
pmc de_M_fetch FETCH_SIZE python $R/bench.py --no-cpu-baseline --no-minimize-wall --kernel-timing-launches 50
pmc de_M_write WRITE_SIZE python $R/bench.py --no-cpu-baseline --no-minimize-wall --kernel-timing-launches 50
pmc eigh_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES SQ_INSTS_VALU" python $R/tools/bench_eigh_jacobi.py 512
python - <<PY
import collections, csv, glob, json
out = {}
for tag in ("de_M_fetch", "de_M_write", "eigh_mfma"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in agg.items():
        if not any(s in k for s in ("de_generation", "eigh_round", "eigh_gemm")):
            continue
        for c, v in d.items():
            out.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
tail -c 600 $OUT/bench_N1.json; echo
for t in de_M eigh cmaes_c4 misc; do echo "== $t"; head -8 $OUT/${t}_kernel_stats.csv | cut -c1-230; done
grep -v "^W20\|^I20\|^E20" $OUT/prof_eigh.log | tail -9; grep -v "^W20\|^I20\|^E20" $OUT/prof_cmaes_c4.log | tail -5; grep -v "^W20\|^I20\|^E20" $OUT/prof_misc.log | tail -6
