"""profiles/pmc_latest.json <- gpurun_out/r2/pmc_summary.json (tools/r2_profiles.sh) + the per-wave SQ counters of
tools/pmc_de_m_sq.sh given on the command line, stamped with HEAD:
    python tools/stamp_pmc.py VALU SALU SMEM LDS WAVE_CYCLES WAIT_ANY WAIT_INST_ANY"""
import json, subprocess, sys
s = json.load(open("gpurun_out/r2/pmc_summary.json"))
key, k = [(key, v) for key, v in s.items() if "de_generation_kernel" in key][0]
p = json.load(open("profiles/pmc_latest.json"))
d = p["de_rosenbrock_n128_p4096"]
d["fetch_size_kb"] = k["FETCH_SIZE"]["mean"]
d["write_size_kb"] = k["WRITE_SIZE"]["mean"]
d["hbm_bytes_per_launch"] = (2.0 * d["fetch_size_kb"] + d["write_size_kb"]) * 1024
d["source"] = ("rocprofv3 --pmc (separate passes: FETCH_SIZE, WRITE_SIZE), tools/r2_profiles.sh, per-dispatch means over %d "
               "launches; SQ counters per wave from tools/pmc_de_m_sq.sh" % k["FETCH_SIZE"]["n"])
if len(sys.argv) > 7:
    v = [int(x) for x in sys.argv[1:8]]
    d["per_wave"] = dict(zip(["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES_quad",
                              "SQ_WAIT_ANY_quad", "SQ_WAIT_INST_ANY_quad"], v))
p["_commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip()
json.dump(p, open("profiles/pmc_latest.json", "w"), indent=1)
print(key[:80], d["hbm_bytes_per_launch"], p["_commit"])
