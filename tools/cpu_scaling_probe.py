import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from oracle import oracle
from uav_motion_planning_amd import workloads as W
oracle.build()
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
b = W.uniform_batch(2, 512, 8, 4, time_mode="distance")
M = 8
t0 = time.perf_counter(); oracle.osqp_solve_batch(4, b["seg_offsets"], b["waypoints"], b["times"], b["bc"], threads=1); d1 = time.perf_counter() - t0
print("1 thread: %.1f traj/s" % (512 / d1))
for th in (2, 4, 8, 16, 32, 64, 128):
    rep = th * 512 // 512
    n = 512 * rep
    so = (np.arange(n + 1) * M).astype(np.int32)
    a = (4, so, np.tile(b["waypoints"], (rep, 1, 1)), np.tile(b["times"], (rep, 1)), np.tile(b["bc"], (rep, 1, 1, 1)))
    oracle.osqp_solve_batch(*a, threads=th)
    t0 = time.perf_counter(); oracle.osqp_solve_batch(*a, threads=th); d = time.perf_counter() - t0
    print("%3d threads: %.0f traj/s, efficiency %.2f" % (th, n / d, (n / d) / (th * 512 / d1)))
