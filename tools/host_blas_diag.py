import os, time, numpy as np
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu.stat"):
    try: print(f, open(f).read().strip().replace("\n", " | "))
    except Exception as e: print(f, "-", e)
from threadpoolctl import threadpool_info, threadpool_limits
print([(d["internal_api"], d["num_threads"]) for d in threadpool_info()])
v = np.random.rand(16384)
for k in range(6):
    t0 = time.perf_counter(); s = float(np.dot(v, v)); print(f"np.dot 16384 #{k}: {(time.perf_counter()-t0)*1e3:.3f} ms", s.hex()); time.sleep(0.3 * k)
with threadpool_limits(limits=1, user_api="blas"):
    for k in range(3):
        t0 = time.perf_counter(); s = float(np.dot(v, v)); print(f"1 thread  #{k}: {(time.perf_counter()-t0)*1e3:.3f} ms", s.hex())
print("cpu.stat after:", open("/sys/fs/cgroup/cpu.stat").read().strip().replace("\n", " | ") if os.path.exists("/sys/fs/cgroup/cpu.stat") else "-")
