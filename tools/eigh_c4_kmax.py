"""Per-sweep off-diagonal mass and max|K| of the decompositions inside a device-resident CMA-ES run at BASELINE config 4's
size (EighInfo.offm / kmax2 after every sweep): what the refinement steps' entry rules see.  SX_EIGH_DEEP / SX_EIGH_REFINE
from the environment.  usage: eigh_c4_kmax.py [n P gens]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stochopy_amd as sa
from stochopy_amd.optimize import _cmaes

n, P, gens = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 1024, 30)
lo, up = np.full(n, -5.12), np.full(n, 5.12)
run = _cmaes._CmaDeviceRun(sa.factory.rosenbrock.sx_id, lo, up, None, gens + 1, P, 0.1, 0.5, 0.0, -1.0, 0, run=False)
run.args.eig_sweeps = 24
eigeneval = 0
for gen in range(1, gens + 1):
    due = gen * P - eigeneval > run.eig_every
    if due:
        due = 2 if eigeneval else 1
        eigeneval = gen * P
    with torch.cuda.stream(run.ctx.stream):
        run.step(gen, int(due))
    run.ctx.sync()
    if due:
        raw = run.eig.ws[:200].cpu().numpy()
        hdr = raw[:2].view(np.int32)
        sweeps, conv = int(hdr[1]), int(hdr[3])
        norm2 = raw[2]
        offm = np.sqrt(raw[64:64 + sweeps] / norm2)
        kmax = np.sqrt(raw[125:125 + sweeps].view(np.uint64).view(np.float64))
        deep_steps = int(raw[193:194].view(np.int32)[0])
        deep_off = np.sqrt(np.abs(raw[185:189]) / norm2)
        deep_k = np.sqrt(raw[189:193].view(np.uint64).view(np.float64))
        print(f"gen {gen:3d} sweeps {sweeps} conv {conv} deep steps {deep_steps}\n   off/|C| after sweep: " +
              " ".join("%.1e" % v for v in offm) + "\n   max|K|   after sweep: " + " ".join("%.1e" % v for v in kmax) +
              ("\n   after deep steps: off " + " ".join("%.1e" % v for v in deep_off[:deep_steps]) + "  max|K| " + " ".join("%.1e" % v for v in deep_k[:deep_steps]) if deep_steps else ""),
              flush=True)
