"""Fill the @@TOKEN@@ placeholders of DESIGN.md from the tracked evidence under profiles/ (round 4): the numbers in the
document are then the numbers of the CSVs / bench lines, not a transcription.  usage: python tools/fill_design.py"""
import csv, json, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda f: os.path.join(R, "profiles", f)

def stats(f):
    return {r["Name"]: r for r in csv.DictReader(open(P(f)))}

def pick(d, sub):
    rows = [r for k, r in d.items() if sub in k]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    return rows[0]

def sig(x, n=3):
    return ("%." + str(n) + "g") % x

def sci(x):
    m, e = ("%.2e" % x).split("e")
    return "%se%d" % (m.rstrip("0").rstrip(".") if "." in m else m, int(e))

t = {}
b = json.loads(open(P("r4_bench_N1.json")).read().strip().splitlines()[-1])
bd = json.loads(open(P("r4_bench_N1_driver_args.json")).read().strip().splitlines()[-1])
cf = b["configs"]
m = pick(stats("r4_de_M_kernel_stats.csv"), "de_generation_kernel")
mk = float(m["AverageNs"]) / 1e3
t.update(MK="%.2f" % mk, MKNS="%d" % round(float(m["AverageNs"])), MN=f"{int(m['Calls']):,}".replace(",", " "), MTB="%.2f" % (16.842752 / mk), MF="%.3f" % (16.842752 / mk / 8.0),
         MBK="%.2f" % b["roofline"]["kernel_us"], MBF="%.3f" % b["roofline"]["frac"], BFRAC="%.3f" % b["roofline"]["frac"],
         VAL=sci(b["value"]), USSTEP="%.2f" % (b["ms_per_step"] * 1e3), DRVUS="%.2f" % (bd["ms_per_step"] * 1e3), DRVV=sci(bd["value"]),
         WALL=sci(b["minimize_wall"]["value"]), WALLF="%.2f" % (b["minimize_wall"]["value"] / b["value"]),
         CPU=sci(b["cpu_baseline"]["value"]), XCPU=f"{round(b['gpu_over_cpu'], -2):,.0f}".replace(",", " "), LOKY=sci(b["cpu_baseline_loky"]["value"]))
c2 = cf["C2_de_rastrigin_n128_p4096"]
t.update(C2="%.2f" % c2["us_per_generation"], C2E=sci(c2["evals_per_s"]), C2F="%.3f" % c2["frac"])
k5 = pick(stats("r4_de_n1024_p16384_kernel_stats.csv"), "de_generation_kernel"); k5us = float(k5["AverageNs"]) / 1e3
c5 = cf["C5_shard_de_n1024_p16384"]
t.update(C5K="%.1f" % k5us, C5N=f"{int(k5['Calls']):,}".replace(",", " "), C5TB="%.2f" % (537.133056 / k5us), C5KF="%.2f" % (537.133056 / k5us / 8.0),
         C5G="%.1f" % c5["us_per_generation"], C5E=sci(c5["evals_per_s"]), C5F="%.2f" % c5["frac"])
k5f = pick(stats("r4_de_n1024_p131072_kernel_stats.csv"), "de_generation_kernel"); k5fus = float(k5f["AverageNs"]) / 1e3
c5f = cf["C5_full_de_n1024_p131072_1gpu"]
t.update(C5FK="%.0f" % k5fus, C5FTB="%.2f" % (4297.064448 / k5fus), C5FKF="%.2f" % (4297.064448 / k5fus / 8.0), C5FG="%.0f" % c5f["us_per_generation"],
         C5FE=sci(c5f["evals_per_s"]), C5FF="%.2f" % c5f["frac"])
ps = stats("r4_pso_c3_kernel_stats.csv"); pk = pick(ps, "pso_generation_kernel"); pus = float(pk["AverageNs"]) / 1e3
c3 = cf["C3a_pso_ackley_n256_p16384"]
t.update(C3K="%.1f" % pus, C3N=pk["Calls"], C3TB="%.1f" % (201.719808 / pus), C3KF="%.2f" % (201.719808 / pus / 8.0),
         C3FIN="%.1f" % (float(pick(ps, "select_finalize")["AverageNs"]) / 1e3), C3G="%.2f" % c3["us_per_generation"], C3E=sci(c3["evals_per_s"]), C3F="%.2f" % c3["frac"])
cs = stats("r4_cpso_c3b_kernel_stats.csv"); c3b = cf["C3b_cpso_ackley_n256_p16384"]
t.update(C3BG="%.1f" % (float(pick(cs, "pso_generation_kernel")["AverageNs"]) / 1e3), C3BS="%.1f" % (float(pick(cs, "pso_restart_select")["AverageNs"]) / 1e3),
         C3BR="%.1f" % (float(pick(cs, "pso_radius")["AverageNs"]) / 1e3), C3BF="%.1f" % (float(pick(cs, "select_finalize")["AverageNs"]) / 1e3),
         C3BT="%.1f" % c3b["us_per_generation"], C3BE=sci(c3b["evals_per_s"]), C3BFR="%.2f of PSO's bytes" % c3b["frac"])
c4s = stats("r4_cmaes_c4_kernel_stats.csv"); c4 = cf["C4_cmaes_rosenbrock_n512_p1024"]
sam = float(pick(c4s, "cma_gemm_kernel<0")["AverageNs"]) / 1e3; rm = float(pick(c4s, "cma_gemm_kernel<1")["AverageNs"]) / 1e3
er = pick(c4s, "eigh_round_kernel"); eg = float(pick(c4s, "eigh_gemm_kernel<false>")["AverageNs"]) / 1e3
gens = int(pick(c4s, "cma_gemm_kernel<0")["Calls"])
t.update(SAMK="%.2f" % sam, SAMTF="%.1f" % (536.870912 / sam / 1e3 * 1e3 / 1e3 if False else 0.536870912 / sam * 1e3), SAMF="%.2f" % (0.536870912 / sam * 1e3 / 78.6),
         RMK="%.2f" % rm, RMTF="%.1f" % (0.268435456 / rm * 1e3), RMF="%.2f" % (0.268435456 / rm * 1e3 / 78.6),
         EIGK="%.1f" % (float(er["AverageNs"]) / 1e3), EIGN=f"{int(er['Calls']):,}".replace(",", " "), EIGL="%d" % round(int(er["Calls"]) / gens),
         EGK="%.1f" % eg, EGTF="%.1f" % (0.268435456 / eg * 1e3), EGF="%.2f" % (0.268435456 / eg * 1e3 / 78.6),
         C4MS="%.2f" % c4["ms_per_generation"], C4E=sci(c4["evals_per_s"]))
leg = cf["M_numpy_legacy_de_rosenbrock_n128_p4096"]
t.update(LEG=sci(leg["evals_per_s"]), XLEG="%.0f" % (leg["evals_per_s"] / b["cpu_baseline"]["value"]))
ev = {}
for line in open(P("r4_eval_kernel.txt")):
    mm = re.match(r"sx_eval (\w+)\s+n=\s*(\d+) P=\s*(\d+):\s+([\d.]+) us .*\(([\d.]+) of 8 TB/s\)", line)
    if mm:
        ev[(mm.group(1), int(mm.group(2)), int(mm.group(3)))] = (mm.group(4), mm.group(5))
t.update(EV128=ev[("rosenbrock", 128, 1 << 20)][0], EV128F=ev[("rosenbrock", 128, 1 << 20)][1], EV64F=ev[("sphere", 64, 1 << 21)][1],
         EV1024F=ev[("rosenbrock", 1024, 1 << 17)][1], EVACK=ev[("ackley", 256, 1 << 19)][1])
s = open(os.path.join(R, "DESIGN.md")).read()
missing = set(re.findall(r"@@(\w+)@@", s)) - set(t)
assert not missing, missing
s = re.sub(r"@@(\w+)@@", lambda mm: t[mm.group(1)], s)
open(os.path.join(R, "DESIGN.md"), "w").write(s)
print({k: t[k] for k in sorted(t)})
