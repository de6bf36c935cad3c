"""A/B of the two general-rows kernels (uavqp_settings.rows_lanes_per_problem: 1 = one lane per problem, HBM workspace -- qp_rows.h;
2 = lane pair per problem, state in LDS -- qp_rows2.h): statuses, iteration counts, working sets and coefficients compared problem by
problem, times per dispatch.  GPU box: python tools/rows_ab.py [n_traj] [small]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W
from tools.bench_configs import timeit

dev = torch.device("cuda", 0); s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
ctx = U.Context(0); ctx.set_stream(s.cuda_stream)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def case(tag, r, n, M=None, K=2, boxes=True, timing=True):
    if M is None:
        b = W.ragged_batch(5, n, r, m_lo=1, m_hi=20); so = b["seg_offsets"]; d_so = up(so); tot = int(so[-1]); uni = 0; mx = int(np.diff(so).max())
        wp = np.asarray(b["waypoints"]).reshape(-1, 3)
        seg_first = (so[:-1] + np.arange(n))
        mid = np.concatenate([0.5 * (wp[so[k] + k:so[k + 1] + k] + wp[so[k] + k + 1:so[k + 1] + k + 1]) for k in range(n)])
    else:
        b = W.uniform_batch(3, n, M, r, time_mode="distance"); d_so = None; tot = n * M; uni = M; mx = M
        wpn = b["waypoints"]; mid = 0.5 * (wpn[:, :-1] + wpn[:, 1:]).reshape(n * M, 3)
    lo, hi = W.corridor_boxes(b, config_index=3)
    tau = np.full((tot, K), 0.5); drv = np.tile(np.array([0, 1], dtype=np.int32)[:K], (tot, 1))
    rlo, rhi = np.zeros((tot, K, 3)), np.zeros((tot, K, 3))
    rlo[:, 0], rhi[:, 0] = mid - 0.25, mid + 0.25
    if K > 1:
        rlo[:, 1], rhi[:, 1] = -3.5, 3.5
    d = {k: up(b[k]) for k in ("waypoints", "times", "bc")}
    d_lo, d_hi = (up(lo), up(hi)) if boxes else (None, None)
    d_tau, d_drv, d_rlo, d_rhi = up(tau), up(drv), up(rlo), up(rhi)
    res = {}
    for mode in (1, 2):
        ctx.set_settings(rows_lanes_per_problem=mode, corridor_initial_guess=1)   # (A/B of the two kernels from the same start: the box set)
        out = torch.zeros(tot * 6 * r, dtype=torch.float64, device=dev)
        st = torch.zeros(n, dtype=torch.int32, device=dev); it = torch.zeros(n, dtype=torch.int32, device=dev)
        act = torch.zeros(n * 3 * (2 + 2 * K), dtype=torch.int64, device=dev)
        run = lambda: ctx.solve_rows_device(r, n, uni, mx, d_so, d["waypoints"], d["times"], d["bc"], d_lo, d_hi, K, d_tau, d_drv, d_rlo, d_rhi, out, st, it, act)
        run(); s.synchronize()
        ms = timeit(run, s, n=3, warm=1) if timing else float("nan")
        res[mode] = dict(out=out.cpu().numpy(), st=st.cpu().numpy(), it=it.cpu().numpy(), act=act.cpu().numpy().reshape(n, 3, -1), ms=ms)
    a, c = res[1], res[2]
    solved = (a["st"] == 1) & (c["st"] == 1)
    scale = np.maximum(1.0, np.abs(a["out"]).max())
    so64 = (np.arange(n + 1) * M if M is not None else so).astype(np.int64)
    err = np.array([np.max(np.abs(a["out"][6 * r * so64[k]:6 * r * so64[k + 1]] - c["out"][6 * r * so64[k]:6 * r * so64[k + 1]]), initial=0.0) for k in range(n)])
    print(json.dumps({"case": tag, "r": r, "n": n, "M": M, "K": K, "ms_one_lane": a["ms"], "ms_pair": c["ms"], "status_equal": bool(np.array_equal(a["st"], c["st"])),
                      "solved": int(solved.sum()), "capped_one_lane": int((a["st"] == -2).sum()), "capped_pair": int((c["st"] == -2).sum()),
                      "iters_mean": [float(a["it"].mean()), float(c["it"].mean())], "iters_max": [int(a["it"].max()), int(c["it"].max())],
                      "iters_equal_frac": float((a["it"] == c["it"]).mean()), "working_sets_equal_frac(solved)": float(np.all(a["act"] == c["act"], axis=(1, 2))[solved].mean()) if solved.any() else None,
                      "max_coef_diff_rel(solved)": float(err[solved].max() / scale) if solved.any() else None}), flush=True)
    ctx.set_settings(rows_lanes_per_problem=0, corridor_initial_guess=2)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
if len(sys.argv) > 2:
    case("tiny M=2", 3, 64, 2); case("tiny M=3", 3, 64, 3); case("tiny M=5 r4", 4, 64, 5); case("M=16 small", 3, 512, 16); case("ragged", 4, 1000, None); case("no boxes K=1", 3, 300, 7, K=1, boxes=False)
else:
    case("config 3 + K=2 rows", 3, n, 16)
