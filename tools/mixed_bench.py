"""Side figure for config C5 (mixed horizons N in {30, 50, 100}, equiprobable): closed-loop RTI
steps/s of a cfnmpc_fleet with device-resident I/O (same plant, kicks and targets as bench.py's
hover workload).  Not the headline metric (that is bench.py on C3); prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--opt", action="append", default=[], help="cfnmpc_opts field = integer value, e.g. --opt as_passes=-1")
    args = ap.parse_args()
    import torch
    from crazyflie_nmpc_amd import sim
    from crazyflie_nmpc_amd.fleet import MixedHorizonFleet
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    from crazyflie_nmpc_amd.synthetic import sample_hover_x0
    dev = torch.device("cuda", 0)
    B, P = args.batch, 20
    rng = np.random.default_rng(5)
    horizons = rng.choice([30, 50, 100], size=B)
    fleet = MixedHorizonFleet(horizons, **{k: int(v) for k, v in (a.split('=') for a in args.opt)})
    fleet.set_regulation(np.tile([0.0, 0.0, 0.4], (B, 1)), 15.7777)
    x = torch.from_numpy(sample_hover_x0(rng, B)).to(dev)
    xn = torch.empty_like(x)
    u0 = torch.empty((B, 4), dtype=torch.float64, device=dev)
    cohort = (B + P - 1) // P
    kicks = torch.from_numpy(sample_hover_x0(rng, cohort * P).reshape(P, cohort, 13)).to(dev)
    fleet.set_x0(x); fleet.init_iterate(INIT_HOVER)
    t = 0

    def step():
        nonlocal x, xn, t
        c0 = (t % P) * cohort
        c1 = min(c0 + cohort, B)
        if c1 > c0:
            x[c0:c1].copy_(kicks[t % P, : c1 - c0])
        fleet.set_x0(x); fleet.solve(1); fleet.get_u(0, u0)
        sim(x, u0, T=0.015, steps=1, out=xn)
        x, xn = xn, x
        t += 1

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    st, it, _ = fleet.stats()
    print(json.dumps({"workload": "C5 mixed horizons {30,50,100}", "batch": B, "rti_steps_per_s": B * args.steps / el,
                      "stage_steps_per_s": float(horizons.sum()) * args.steps / el, "ms_per_step": el / args.steps * 1e3,
                      "status_ok_frac": float((st == 0).mean()), "mean_qp_solves": float(it.mean()),
                      "workspace_GB": fleet.workspace_bytes / 1e9}))


if __name__ == "__main__":
    main()
