"""A/B of the ragged / runtime-M equality kernels: one lane per trajectory (1), a lane pair per trajectory (2), one lane per
(trajectory, axis) (3) -- results compared bit for bit / to 1e-9 with mode 1, times per launch.  GPU box: python tools/generic2_probe.py"""
import os, sys, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W
from tools.bench_configs import timeit
dev = torch.device("cuda", 0); s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
ctx = U.Context(0); ctx.set_stream(s.cuda_stream)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)

def case(r, n, M=None, tag=""):
    if M is None:
        b = W.ragged_batch(4, n, r); so = b["seg_offsets"]; d_so = torch.from_numpy(so).to(dev); tot = int(so[-1]); uni = 0; mx = int(np.diff(so).max())
    else:
        b = W.uniform_batch(4, n, M, r, time_mode="distance"); d_so = None; tot = n * M; uni = M; mx = M
    d = {k: up(b[k]) for k in ("waypoints", "times", "bc")}
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    res = {}
    for mode in (1, 2, 3):
        for wpc in ((0,) if mode != 2 else (0, 2, 8)):
            ctx.set_settings(generic_lanes_per_traj=mode, generic_waves_per_cu=wpc)
            out = torch.zeros(tot * 6 * r, dtype=torch.float64, device=dev)
            ms = timeit(lambda: ctx.solve_batch_device(r, n, uni, mx, d_so, d["waypoints"], d["times"], d["bc"], out, st), s)
            s.synchronize()
            res[(mode, wpc)] = out.cpu().numpy()
            ref = res[(1, 0)]
            err = float(np.max(np.abs(res[(mode, wpc)] - ref) / (1 + np.abs(ref))))
            print(json.dumps({"case": tag, "r": r, "n": n, "M": M, "mode": mode, "wpc": wpc, "us": ms * 1e3, "rel_err_vs_mode1": err,
                              "status_ok": int((st.cpu().numpy() == 1).sum())}), flush=True)
    ctx.set_settings(generic_lanes_per_traj=0, generic_waves_per_cu=0)

ctx.set_variant(1)
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    for (mode, wpc) in ((1, 0), (2, 0)):
        pass
    case(4, 32768, None, "config4"); case(3, 32768, None, "ragged r3"); case(4, 4096, None, "ragged 4096"); case(4, 32768, 1, "uniform 1"); case(4, 262144, 14, "big 14")
    sys.exit(0)
case(4, 32768, None, "config4")
case(3, 32768, None, "ragged r3")
case(4, 4096, None, "ragged 4096")
case(4, 1000, None, "ragged 1000")
case(4, 32768, 14, "uniform 14")
case(4, 32768, 24, "uniform 24")
case(4, 32768, 1, "uniform 1")
case(4, 32768, 2, "uniform 2")
case(4, 32768, 3, "uniform 3")
case(3, 65536, 20, "uniform 20 r3")
case(4, 262144, 14, "big 14")
