"""What torch's NCCL flight recorder says about outstanding collectives (parallel.World._watchdog_idle relies on the
`retired` field): one rank, one all-gather, then the record every 20 ms."""
import os, pickle, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29631")
import torch, torch.distributed as dist
from torch._C._distributed_c10d import _dump_nccl_trace
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(8, device="cuda"); y = torch.empty(8, device="cuda")
dist.all_gather_into_tensor(y, x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(12):
    e = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=False))["entries"]
    print(f"{(time.perf_counter()-t0)*1e3:6.1f} ms: {len(e)} entries", [(d.get("profiling_name"), d.get("state"), d.get("retired")) for d in e], flush=True)
    if k == 0 and e: print("keys:", sorted(e[0].keys()))
    time.sleep(0.02)
from stochopy_amd.parallel import World
print("World._watchdog_idle():", World._watchdog_idle())
dist.destroy_process_group()
