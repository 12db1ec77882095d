"""Host enqueue cost vs GPU time of the hinted chain launches (sx_de_chain_run) at the metric shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stochopy_amd import _lib
from stochopy_amd.optimize import _de
hint = os.environ.get("SX_DE_HINT", "1")
n, P = 128, 4096
run = _de._DeRun(_lib.FUN_IDS["rosenbrock"], np.full(n, -5.12), np.full(n, 5.12), None, 2**31 - 2, P, 0.5, 0.9, "best1bin",
                 None, 0.0, -1.0, False, 1.0, None, "philox", 1234, 1, autorun=False)
with torch.cuda.stream(run.ctx.stream):
    run._setup(); run.prepare_graphs(); run.enqueue(200); run.ctx.sync()
    for ngen in (2000, 2000):
        t0 = time.perf_counter(); run.enqueue(ngen); t1 = time.perf_counter(); run.ctx.sync(); t2 = time.perf_counter()
        print(f"SX_DE_HINT={hint}: enqueue of {ngen} generations took {(t1-t0)/ngen*1e6:.2f} us/gen on the host; until done {(t2-t0)/ngen*1e6:.2f} us/gen", flush=True)
run.close()
