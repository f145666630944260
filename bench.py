#!/usr/bin/env python
"""bench.py -- NMPC RTI steps/s of the MI355X-native Crazyflie SQP-RTI engine.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it
under torch.distributed.run (one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json metric: "NMPC RTI steps/sec (batch=65536, N=50, nx=13, nu=4)";
SURVEY.md section 8d configs C2/C3): per GPU a synthetic fleet of 65536 Crazyflie hover-regulation
problems, horizon N = 50, run CLOSED LOOP through the RK4 plant (device-resident):
    step = { x0 <- plant state ; acados_solve() equivalent: linearise + Riccati-IPM QP + update ;
             u0 -> plant RK4 step ; 1/20 of the fleet is kicked to a fresh random perturbed state }
The staggered kicks make every step statistically identical to the time average of SURVEY's
"20 RTI steps closed loop from a perturbed hover" (so the timed region never degenerates into
the converged, bound-free regime).  Instances are independent: the batch shards across GPUs
with no data-path collective ("scaling": "weak" -- 65536 instances per GPU); RCCL is used only
to aggregate the report (max time, statistics).

Also reported on the same line:
  roofline     -- dominant kernel (k_qp): algorithmic bytes per launch / HIP-event duration vs
                  8 TB/s;
  cpu_baseline -- the plain-C CPU restatement of the same algorithm (oracle/cfnmpc_ref.c,
                  "port": acados itself cannot be built here) timed on this host's cores on a
                  bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12         # B/s, MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
FP64_VEC_PEAK = 78.6e12   # FLOP/s, FP64 vector (= FP64 matrix) peak of MI355X
N_HORIZON = 50
KICK_PERIOD = 20


def alg_bytes_step(N):
    """algorithmic bytes per RTI step per instance, 8*(553 N + 80)."""
    return 8 * (553 * N + 80)


def alg_bytes_qp(N):
    """Share of the QP kernel: stage blocks read once (251 N + 13 words) +
    x0, yref, iterate read, iterate write, status (51 N + 54 words)."""
    return 8 * (302 * N + 67)


def cpu_baseline(seed, n_inst=2048, n_steps=4, reps=3):
    """Times the CPU restatement on a bounded sample of the same workload (closed loop).
    Best of `reps` repetitions for both figures: GPU boxes are shared hosts and single timings
    of the host cores vary by up to 10x between runs."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cfnmpc_oracle as o
    import cref
    cref.build()
    yr, ye = o.regulation_yref(N_HORIZON, (0.0, 0.0, 0.4))
    yref = np.repeat(yr[None], n_inst, 0).copy()
    yref_e = np.repeat(ye[None], n_inst, 0).copy()
    opts = cref.default_opts(active_set=1)   # same QP method as the engine's default
    cores = os.cpu_count() or 1
    best_all, best_one, used, iters = 0.0, 0.0, 1, []
    for _ in range(reps):
        rng = np.random.default_rng(seed)
        x = o.sample_hover_x0(rng, n_inst)
        xit = np.repeat(x[:, None, :], N_HORIZON + 1, 1).copy()
        uit = np.full((n_inst, N_HORIZON, 4), o.HOV_W)
        t_solve = 0.0
        iters = []
        for _s in range(n_steps):
            t0 = time.perf_counter()
            st, it, rs, used = cref.rti_step(opts, xit, uit, x.copy(), yref, yref_e, nthreads=0)
            t_solve += time.perf_counter() - t0
            iters.append(float(it.mean()))
            x = cref.sim(x, uit[:, 0, :].copy(), 0.015, 1)
        best_all = max(best_all, n_inst * n_steps / t_solve)
        # single-thread rate on a small slice of the same states
        m = min(64, n_inst)
        xs = np.repeat(x[:m, None, :], N_HORIZON + 1, 1).copy(); us = np.full((m, N_HORIZON, 4), o.HOV_W)
        t0 = time.perf_counter()
        cref.rti_step(opts, xs, us, x[:m].copy(), yref[:m].copy(), yref_e[:m].copy(), nthreads=1)
        best_one = max(best_one, m / (time.perf_counter() - t0))
    return {
        "value": best_all, "unit": "RTI steps/s", "cores": int(used),
        "host_cores": int(cores), "kind": "port",
        "single_thread_steps_per_s": best_one,
        "sample": f"best of {reps} x ({n_inst} instances x {n_steps} closed-loop RTI steps of the same hover workload), "
                  f"oracle/cfnmpc_ref.c (CPU restatement, not acados; same QP method as the engine: active-set solves, "
                  f"interior point as fall-back), OpenMP over {used} threads; mean QP solves {np.mean(iters):.2f}",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--active-horizon", type=int, default=1)
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl = RCCL over xGMI (default); gloo only for functional checks of the N > 1 path on "
                         "a box with fewer GPUs than ranks (ranks then share devices)")
    ap.add_argument("--workload", choices=["hover", "figure8"], default="hover",
                    help="hover = config C3 (the metric's configuration); figure8 = config C4 tracking with "
                         "device-side reference windows")
    ap.add_argument("--overlap", type=int, default=None, help="cfnmpc_opts.overlap_linearise (default: library default)")
    ap.add_argument("--ah-margin", type=float, default=None)
    ap.add_argument("--ah-extra", type=int, default=None)
    ap.add_argument("--streams", type=int, default=1,
                    help="sub-batches per GPU, each with its own solver object and HIP stream (overlaps the "
                         "latency-bound interior-point tail of one shard with the streaming kernels of the others)")
    args = ap.parse_args()

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the engine has no CPU path)")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    if args.dist_backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible")
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    dist = None
    red_dev = dev  # device of the tiny reduction tensors
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            red_dev = torch.device("cpu")

    from crazyflie_nmpc_amd import BatchSolver, default_opts, sim
    from crazyflie_nmpc_amd.synthetic import regulation_row, sample_hover_x0 as sample_x0
    from crazyflie_nmpc_amd.solver import INIT_HOVER

    B, N = args.batch, N_HORIZON
    from crazyflie_nmpc_amd import parallel
    seed = parallel.shard_seed(rank)
    rng = np.random.default_rng(seed)
    row = regulation_row((0.0, 0.0, 0.4))
    S = max(1, min(args.streams, B // 1024 if B >= 1024 else 1))

    class Shard:
        """One sub-batch: own solver object, own HIP stream, device-resident plant state."""

        def __init__(self, lo, hi):
            n = hi - lo
            self.n = n
            self.stream = torch.cuda.Stream(dev) if S > 1 else torch.cuda.current_stream(dev)
            self.x = torch.from_numpy(sample_x0(rng, n)).to(dev)
            self.xn = torch.empty_like(self.x)
            self.u0 = torch.empty((n, 4), dtype=torch.float64, device=dev)
            self.cohort = (n + KICK_PERIOD - 1) // KICK_PERIOD
            self.kicks = torch.from_numpy(sample_x0(rng, self.cohort * KICK_PERIOD).reshape(KICK_PERIOD, self.cohort, 13)).to(dev)
            kw = dict(active_horizon=args.active_horizon)
            if args.overlap is not None:
                kw["overlap_linearise"] = args.overlap
            if args.ah_margin is not None:
                kw["ah_margin"] = args.ah_margin
            if args.ah_extra is not None:
                kw["ah_extra"] = args.ah_extra
            self.solver = BatchSolver(n, default_opts(**kw))
            self.track = args.workload == "figure8"
            if self.track:
                # config C4: figure-8 reference synthesised from the reference's
                # crazyflie_demo/scripts/figure8.csv (three laps + N+1 hold rows), per-instance phase
                # offsets, z offset 0.5 m; windows are generated on the device every step
                from crazyflie_nmpc_amd.trajectories import Figure8, figure8_reference
                lap = figure8_reference(Figure8(np.load(os.path.join(ROOT, "crazyflie_nmpc_amd", "data", "figure8_coeffs.npy"))), z0=0.5, N=N)
                lap1 = lap[:-(N + 1)]
                self.traj = torch.from_numpy(np.concatenate([lap1, lap1, lap1, lap[-(N + 1):]])).to(dev)
                self.it = torch.from_numpy(rng.integers(0, 436, n).astype(np.int32)).to(dev)
                self.mode = torch.ones(n, dtype=torch.int32, device=dev)
                self.des = torch.zeros((n, 3), dtype=torch.float64, device=dev)
                hover = torch.tensor([0, 0, 0.4, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0.0], dtype=torch.float64, device=dev)
                self.pert = 0.3 * (self.kicks - hover)            # perturbations around the reference row
                self.x = self.traj[self.it.long(), :13] + 0.3 * (self.x - hover)
                self.x[:, 3:7] /= torch.linalg.norm(self.x[:, 3:7], dim=1, keepdim=True)
                self.solver.set_yref_windows(self.traj, self.mode, self.it.clone(), self.des, 15.7777)
            else:
                yref = torch.from_numpy(np.tile(row, (n, N, 1))).to(dev)
                yref_e = torch.from_numpy(np.tile(row[:13], (n, 1))).to(dev)
                self.solver.set_yref(yref, yref_e)
            self.solver.set_x0(self.x)
            self.solver.init_iterate(INIT_HOVER)
            self.t = 0

        def step(self):
            with torch.cuda.stream(self.stream):
                t, xc = self.t, self.x
                c0 = (t % KICK_PERIOD) * self.cohort
                c1 = min(c0 + self.cohort, self.n)
                if c1 > c0 and not self.track:
                    xc[c0:c1].copy_(self.kicks[t % KICK_PERIOD, : c1 - c0])  # disturbance of one cohort
                if self.track:
                    if c1 > c0:   # disturbance relative to the vehicle's current reference row
                        ref = self.traj[self.it[c0:c1].long(), :13]
                        kick = ref + self.pert[t % KICK_PERIOD, : c1 - c0]
                        kick[:, 3:7] /= torch.linalg.norm(kick[:, 3:7], dim=1, keepdim=True)
                        xc[c0:c1].copy_(kick)
                    # NMPC::iteration window logic on the device (acados_mpc.cpp:460-485)
                    self.solver.set_yref_windows(self.traj, self.mode, self.it, self.des, 15.7777)
                self.solver.set_x0(xc)                               # lbx = ubx = x0 (acados_mpc.cpp:581)
                self.solver.solve(1, self.stream.cuda_stream)        # acados_solve()  (acados_mpc.cpp:611)
                self.solver.get_u(0, out=self.u0)                    # ocp_nlp_out_get(.., 0, "u")  (:619)
                sim(xc, self.u0, T=0.015, steps=1, out=self.xn)      # plant: one RK4 step of the ODE
                self.x, self.xn = self.xn, xc
                self.t = t + 1

    shards = [Shard(*parallel.shard_range(B, i, S)) for i in range(S)]
    torch.cuda.synchronize(dev)

    def step():
        for sh in shards:
            sh.step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # statistics of the last step + per-kernel durations from a short profiled continuation
    st = np.concatenate([sh.solver.stats()[0] for sh in shards])
    it = np.concatenate([sh.solver.stats()[1] for sh in shards])
    heads = np.concatenate([sh.solver.heads() for sh in shards])
    # kernel durations: ONE shard at a time on an otherwise idle GPU, so that the HIP events bracket
    # the kernels alone (with overlapping streams an event pair would also span foreign kernels)
    prof_steps = min(args.steps, 10)
    ms_lin = ms_qp = 0.0
    for sh in shards:
        torch.cuda.synchronize(dev)
        sh.solver.set_profiling(True)
        for _ in range(prof_steps):
            sh.step()
        torch.cuda.synchronize(dev)
        a_, b_, _n = sh.solver.get_profile()
        sh.solver.set_profiling(False)
        ms_lin += a_
        ms_qp += b_

    stats = torch.tensor([float((st == 0).sum()), float((st != 0).sum()), float(it.sum()), float((it > 0).sum()),
                          float(heads.sum()), ms_lin, ms_qp], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)   # RCCL: aggregate reporting only
    stats = stats.cpu().numpy()
    total_inst = B * world
    ms_lin_avg, ms_qp_avg = stats[5] / world, stats[6] / world

    if rank == 0:
        value = total_inst * args.steps / elapsed
        # BASELINE.md section 4 / SURVEY.md section 8d: algorithmic bytes of ONE RTI STEP (B_alg per instance x the
        # instances of one launch) over the duration of the step's kernels, measured with HIP events on
        # the launch stream (linearisation phase + QP phase); the QP phase is reported beside it
        ms_step = ms_lin_avg + ms_qp_avg
        ach = alg_bytes_step(N) * B / (ms_step * 1e-3)
        ach_qp = alg_bytes_qp(N) * B / (ms_qp_avg * 1e-3)
        traffic = traffic_qp = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if int(tj.get("batch", -1)) == B:
                    traffic_qp = tj.get("hbm_bytes_per_launch_k_qp")
                    traffic = tj.get("hbm_bytes_per_step")
                    if traffic is None and traffic_qp is not None:
                        traffic = traffic_qp + tj["kernels"]["cfn::k_linearise"]["total_bytes"]
            except Exception:
                traffic = traffic_qp = None
        out = {
            "metric": "NMPC RTI steps/sec (batch=65536, N=50, nx=13, nu=4)",
            "value": value, "unit": "RTI steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("C3 hover regulation" if args.workload == "hover" else
                                    "C4 figure-8 tracking (device-side reference windows)") +
                                   ", closed loop through the RK4 plant, staggered kicks "
                                   f"(1/{KICK_PERIOD} of the fleet per step)", "batch_per_gpu": B, "horizon_N": N,
                       "streams_per_gpu": S,
                       "nx": 13, "nu": 4, "sharding": f"independent instances, {world} shard(s), no data-path collective",
                       "qp": ("primal-dual active-set solves on the Riccati factorisation (exact, KKT-verified; "
                              "Mehrotra interior point, tol 1e-8, as fall-back), ") +
                             ("active-horizon sweeps" if args.active_horizon else "full-horizon sweeps")},
            "roofline": {"bound": "hbm", "kernel": "one RTI step = k_linearise + k_factor + k_forward + k_compact + k_scatter + "
                                                     "k_as + k_ipm_rest (HIP events on the launch stream around the two "
                                                     "phases, summed over the sub-batch launches)",
                         "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": ach / HBM_PEAK, "traffic": traffic,
                         "alg_bytes_per_launch": alg_bytes_step(N) * B, "kernel_ms": ms_step,
                         "linearise_ms": ms_lin_avg, "qp_ms": ms_qp_avg,
                         "qp_phase": {"kernels": "k_factor + k_forward + k_compact + k_scatter + k_as + k_ipm_rest",
                                      "alg_bytes_per_launch": alg_bytes_qp(N) * B, "achieved": ach_qp / 1e9,
                                      "frac": ach_qp / HBM_PEAK, "traffic": traffic_qp},
                         # the same model against the wall clock of the whole closed loop (plant, I/O kernels included)
                         "step_frac_hbm": alg_bytes_step(N) * value / (world * HBM_PEAK)},
            "qp_stats": {"status_ok_frac": stats[0] / total_inst, "mean_qp_solves": stats[2] / total_inst,
                         "frac_constrained": stats[3] / total_inst, "mean_head_stages": stats[4] / total_inst},
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU restatement is timed at N = 1 only
            try:
                out["cpu_baseline"] = cpu_baseline(seed)
            except Exception as e:  # the baseline is a report, never a dependency of the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "RTI steps/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
