"""Can a closed-form guess of the working set save iterations of the corridor solver?  Guess: the min-jerk (min-snap) polynomial with
only the end states fixed -- one quintic (septic) over the total time -- evaluated at the knot times; a knot whose box it misses
is guessed active on that side.  Compared with the cold start on BASELINE config 3.  GPU box: python tools/corridor_warm_guess_probe.py"""
import json, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W
from tools.bench_configs import timeit

dev = torch.device("cuda", 0); s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
ctx = U.Context(0); ctx.set_stream(s.cuda_stream)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
r, M, n = 3, 16, 65536
b = W.uniform_batch(3, n, M, r, time_mode="distance")
lo, hi = W.corridor_boxes(b)
wp, T, bc = b["waypoints"], b["times"], b["bc"]              # [n, M+1, 3], [n, M], [n, 2, r-1, 3]


def end_state_polynomial(wp, T, bc):
    """positions at the interior knots of the degree-(2r-1) polynomial matching position + r-1 derivatives at both ends"""
    Ttot = T.sum(axis=1)
    tk = np.cumsum(T, axis=1)[:, :-1]                          # [n, M-1]
    nc = 2 * r
    out = np.zeros((wp.shape[0], M - 1, 3))
    fact = [1.0, 1.0, 2.0, 6.0]
    for ax in range(3):
        # unknown coefficients c_r..c_{2r-1}; c_d = start derivative d / d!
        c_lo = np.stack([wp[:, 0, ax]] + [bc[:, 0, d, ax] / fact[d + 1] for d in range(r - 1)], axis=1)        # [n, r]
        A = np.zeros((wp.shape[0], r, r)); rhs = np.zeros((wp.shape[0], r))
        endv = np.stack([wp[:, -1, ax]] + [bc[:, 1, d, ax] for d in range(r - 1)], axis=1)                     # end derivatives 0..r-1
        for d in range(r):
            for j in range(r):
                k = r + j
                A[:, d, j] = np.prod(np.arange(k - d + 1, k + 1)) * Ttot ** (k - d)
            known = sum(np.prod(np.arange(k - d + 1, k + 1)) * c_lo[:, k] * Ttot ** (k - d) for k in range(d, r))
            rhs[:, d] = endv[:, d] - known
        c_hi = np.linalg.solve(A, rhs[:, :, None])[:, :, 0]
        c = np.concatenate([c_lo, c_hi], axis=1)
        out[:, :, ax] = sum(c[:, k, None] * tk ** k for k in range(nc))
    return out


q = end_state_polynomial(wp, T, bc)
act = np.zeros((n, 3, 2), dtype=np.int64)
for k in range(1, M):
    below = q[:, k - 1] < lo[:, k]
    above = q[:, k - 1] > hi[:, k]
    act[:, :, 0] |= ((below | above).astype(np.int64) << k)
    act[:, :, 1] |= (above.astype(np.int64) << k)
print("guess: %.1f of %d knots active per (trajectory, axis)" % (np.mean([bin(int(v)).count("1") for v in act[:2000, :, 0].ravel()]), M - 1))

d = {k: up(v) for k, v in (("wp", wp.reshape(-1, 3)), ("T", T.reshape(-1)), ("bc", bc), ("lo", lo.reshape(-1, 3)), ("hi", hi.reshape(-1, 3)))}
out = torch.zeros(n * 3 * M * 2 * r, dtype=torch.float64, device=dev)
st = torch.zeros(n, dtype=torch.int32, device=dev); it = torch.zeros(n, dtype=torch.int32, device=dev)
a_cold = torch.zeros((n, 3, 2), dtype=torch.int64, device=dev)
ctx.solve_corridor_device(r, n, M, M, None, d["wp"], d["T"], d["bc"], d["lo"], d["hi"], out, st, it, a_cold, False); s.synchronize()
ref = out.clone(); it_cold = it.float().mean().item()
true_act = a_cold.cpu().numpy()
print("solution: %.1f knots active per (trajectory, axis); guess agrees on %.1f %% of the knot decisions"
      % (np.mean([bin(int(v)).count("1") for v in true_act[:2000, :, 0].ravel()]),
         100.0 * np.mean([(bin(int((a ^ g) & 0xFFFE)).count("1")) for a, g in zip(true_act[:4000, :, 0].ravel(), act[:4000, :, 0].ravel())]) / (M - 1)))
def local_chord_guess(alpha):
    """knot k is guessed active on the side of the midpoint of its neighbours if that midpoint is further than alpha box half-widths
    from the waypoint (the smooth path cuts the corner)"""
    g = np.zeros((n, 3, 2), dtype=np.int64)
    mid = 0.5 * (wp[:, :-2] + wp[:, 2:])                      # [n, M-1, 3] for knots 1..M-1
    h = 0.5 * (hi[:, 1:M] - lo[:, 1:M])
    c = 0.5 * (hi[:, 1:M] + lo[:, 1:M])
    up_ = mid > c + alpha * h
    dn_ = mid < c - alpha * h
    for k in range(1, M):
        g[:, :, 0] |= ((up_[:, k - 1] | dn_[:, k - 1]).astype(np.int64) << k)
        g[:, :, 1] |= (up_[:, k - 1].astype(np.int64) << k)
    return g


def disagreement(g):
    return 100.0 * np.mean([bin(int((a ^ b_) & 0xFFFE)).count("1") for a, b_ in zip(true_act[:4000, :, 0].ravel(), g[:4000, :, 0].ravel())]) / (M - 1)


def end_state_guess(beta):
    """active where the end-state polynomial leaves the box scaled by beta about its centre (beta = 0: every knot active)"""
    g = np.zeros((n, 3, 2), dtype=np.int64)
    h = 0.5 * (hi[:, 1:M] - lo[:, 1:M]); c = 0.5 * (hi[:, 1:M] + lo[:, 1:M])
    up_ = q > c + beta * h
    dn_ = q < c - beta * h
    if beta == 0.0:
        dn_ = ~up_
    for k in range(1, M):
        g[:, :, 0] |= ((up_[:, k - 1] | dn_[:, k - 1]).astype(np.int64) << k)
        g[:, :, 1] |= (up_[:, k - 1].astype(np.int64) << k)
    return g


guesses = {"end-state polynomial": act}
for beta in (0.0, 0.5, 1.5, 2.5):
    guesses["end-state beta %.1f" % beta] = end_state_guess(beta)
for alpha in (1.0,):
    guesses["local chord %.1f" % alpha] = local_chord_guess(alpha)
res = {}
for name, g in [("cold", None)] + list(guesses.items()):
    warm = g is not None
    a_guess0 = up(g) if warm else None
    if warm:
        print("%s: %.1f active guessed, %.1f %% of the knot decisions differ from the solution" % (name, np.mean([bin(int(v)).count("1") for v in g[:2000, :, 0].ravel()]), disagreement(g)))
    def run():
        a = a_guess0.clone() if warm else a_cold
        ctx.solve_corridor_device(r, n, M, M, None, d["wp"], d["T"], d["bc"], d["lo"], d["hi"], out, st, it, a, warm)
    ms = timeit(run, s, n=10, warm=2)
    s.synchronize()
    res[name] = dict(ms=ms, iters_mean=it.float().mean().item(), iters_max=int(it.max().item()), solved=int((st == 1).sum().item()),
                     max_diff=float((out - ref).abs().max().item()))
print(json.dumps(res))
