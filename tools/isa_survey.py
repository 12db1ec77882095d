#!/usr/bin/env python3
"""Register and scratch use of every kernel of the library, from the compiler's own assembly (no GPU needed):

    python tools/isa_survey.py [> profiles/rN_isa_survey.txt]

Compiles each csrc/*.hip with `-S --cuda-device-only` into build_ab/ (git-ignored) and prints, per translation unit, the
kernels that use scratch memory (spills) or sit at the VGPR cap of their workgroup size, then the headline kernels.
This is how the spilling `updating="immediate"` sweeps and the 130-VGPR variant of the PSO kernel were found."""
import collections, glob, os, re, shutil, subprocess, sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, OUT = os.path.join(ROOT, "stochopy_amd", "csrc"), os.path.join(ROOT, "build_ab")
os.makedirs(OUT, exist_ok=True)
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-S",
         "--cuda-device-only"]


def asm(path):
    out = os.path.join(OUT, os.path.basename(path)[:-4] + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [path, "-o", out], check=True, stderr=subprocess.DEVNULL)
    return out


with ThreadPoolExecutor(8) as ex:
    files = list(ex.map(asm, sorted(glob.glob(os.path.join(SRC, "*.hip")))))
rows = []
for f in files:
    s = open(f).read()
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        g = lambda k: int((re.search(r"\.amdhsa_" + k + r"\s+(\d+)", m.group(2)) or [0, 0])[1])
        rows.append([os.path.basename(f), m.group(1), g("next_free_vgpr"), g("next_free_sgpr"), g("private_segment_fixed_size"),
                     g("group_segment_fixed_size")])
filt = shutil.which("c++filt")
names = (subprocess.run([filt], input="\n".join(r[1] for r in rows), capture_output=True, text=True).stdout.split("\n")
         if filt else [r[1] for r in rows])
for r, d in zip(rows, names):
    r[1] = d.replace("(anonymous namespace)::", "")
print(f"{len(rows)} kernels in {len(files)} translation units (hipcc -O3, gfx950)\n")
print("kernels that use scratch memory (bytes per lane), or more than 128 VGPRs:")
per = collections.defaultdict(list)
for f, name, vgpr, sgpr, scratch, lds in rows:
    if scratch or vgpr > 128:
        per[f].append((name, vgpr, scratch))
for f in sorted(per):
    print(f"  {f}: {len(per[f])}")
    for name, vgpr, scratch in per[f]:
        print(f"      vgpr {vgpr:3d}  scratch {scratch:4d}  {name[:110]}")
print("\nheadline kernels:")
for f, name, vgpr, sgpr, scratch, lds in rows:
    if any(k in name for k in ("de_generation_kernel<4, 1, 1, 32, true, 128, 2>", "de_generation_kernel<3, 1, 1, 32, true, 128, 2>",
                               "de_generation_kernel<4, 1, 0, 64, false, 0, -1>", "de_generation_kernel<4, 1, 2, 32, true, 128, 2>",
                               "pso_generation_kernel<0, 1, 64", "eigh_round_kernel", "cma_gemm_kernel<0, 32, 32", "vd_update_kernel",
                               "de_async_kernel<4, 1, 32, 128>", "de_async_kernel<0, 1, 32, 128>")):
        print(f"  vgpr {vgpr:3d}  sgpr {sgpr:3d}  scratch {scratch:4d}  static LDS {lds:6d}  {name[:100]}")
