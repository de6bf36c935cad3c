"""Cycles per section of the ragged pair kernel (wave 0 = the longest trajectories of the first window), from a -DG2_TIMING build:
  hipcc ... -DG2_TIMING -shared -o tools/ubench/libuavqp_g2_timing.so uavqp.hip ;  UAVQP_LIB_PATH=tools/ubench/libuavqp_g2_timing.so python tools/generic2_sections.py"""
import ctypes, os, sys, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W
dev = torch.device("cuda", 0); s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
ctx = U.Context(0); ctx.set_stream(s.cuda_stream)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
r, n = 4, 32768
b = W.ragged_batch(4, n, r); so = b["seg_offsets"]; d_so = torch.from_numpy(so).to(dev); tot = int(so[-1])
d = {k: up(b[k]) for k in ("waypoints", "times", "bc")}
st = torch.zeros(n, dtype=torch.int32, device=dev)
out = torch.zeros(tot * 6 * r, dtype=torch.float64, device=dev)
ctx.set_settings(generic_lanes_per_traj=2)
lib = U.lib()
for rep in range(3):
    ctx.solve_batch_device(r, n, 0, 24, d_so, d["waypoints"], d["times"], d["bc"], out, st)
    buf = (ctypes.c_longlong * 15)()
    assert lib.uavqp_debug_generic2_stamps(ctx._h, buf) == 0
    t = list(buf)
    names = ["offsets", "dma issue", "bc + dma wait", "forward", "meeting", "backward"]
    print(json.dumps({"own_segments_lane0": t[8], **{names[k]: t[k + 1] - t[k] for k in range(6)}, "total": t[6] - t[0],
                      "backward parts": {"y": t[9], "reload+emission": t[10], "touch other": t[11], "lds reads": t[12], "stores+touch own": t[13]}}))
