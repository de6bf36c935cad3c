#!/usr/bin/env python3
"""Randomised soak of the specialised (register-resident) kernels against the generic one at random batch sizes -- partial
last tiles, every tile shape, both orders: python tools/soak_tiles.py [n_draws] [seed].  Agreement 1e-9 relative per
trajectory (1e-7 for the stress time allocation) (both are checked against the oracle elsewhere; this covers sizes the oracle is too slow for)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uav_motion_planning_amd as U  # noqa: E402
from uav_motion_planning_amd import workloads as W  # noqa: E402


def main():
    n_draws = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    ctx = U.Context(0)
    shapes = [(4, m) for m in (2, 3, 4, 5, 6, 7, 8, 9, 10, 12)] + [(3, m) for m in (2, 3, 4, 5, 6, 7, 8, 10, 12, 16)]
    worst = 0.0
    for draw in range(n_draws):
        r, M = shapes[int(rng.integers(0, len(shapes)))]
        n = int(rng.choice([rng.integers(1, 300), rng.integers(300, 9000), rng.integers(9000, 90000)]))
        mode = str(rng.choice(["reference", "distance", "wide"]))
        b = W.uniform_batch(draw, n, M, r, time_mode=mode, seed=seed * 100000 + draw)
        b["bc"] = rng.uniform(-2.0, 2.0, size=b["bc"].shape)
        d = {k: torch.from_numpy(np.ascontiguousarray(b[k])).to(dev) for k in ("waypoints", "times", "bc")}
        outs = []
        variant = int(rng.choice([0, 2, 4, 8, 16, 32]))
        for v in (variant, 1):
            ctx.set_variant(v)
            out = torch.full((n * 3 * M * 2 * r,), float("nan"), dtype=torch.float64, device=dev)
            st = torch.zeros(n, dtype=torch.int32, device=dev)
            ctx.solve_batch_device(r, n, M, M, None, d["waypoints"], d["times"], d["bc"], out, st)
            ctx.synchronize()
            if not bool((st == U.UAVQP_SOLVED).all()):
                print("STATUS FAILURE", draw, r, M, n, v)
                return 1
            outs.append(out.reshape(n, -1))
        ctx.set_variant(0)
        err = ((outs[0] - outs[1]).abs().amax(dim=1) / outs[1].abs().amax(dim=1)).max().item()
        worst = max(worst, err)
        if not err < (1e-7 if mode == "wide" else 1e-9):   # the stress allocation T in [0.2, 5] costs both kernels two digits
            print("AGREEMENT FAILURE", draw, dict(r=r, M=M, n=n, variant=variant, mode=mode, err=err))
            return 1
    print("tile soak ok: %d draws, seed %d, worst relative difference specialised vs generic %.2e" % (n_draws, seed, worst))
    return 0


if __name__ == "__main__":
    sys.exit(main())
