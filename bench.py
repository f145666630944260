#!/usr/bin/env python
"""bench.py -- NMPC RTI steps/s of the MI355X-native Crazyflie SQP-RTI engine.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it
under torch.distributed.run (one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json metric: "NMPC RTI steps/sec (batch=65536, N=50, nx=13, nu=4) at 1/2/4/8
MI355X"): a synthetic fleet of 65 536 Crazyflie hover-regulation
problems, horizon N = 50, run CLOSED LOOP through the RK4 plant (device-resident):
    step = { x0 <- plant state ; acados_solve() equivalent: linearise + QP + full step ;
             u0 -> plant RK4 step ; 1/20 of the fleet is kicked to a fresh random perturbed state }
The staggered kicks make every step statistically identical to the time average of SURVEY's
"20 RTI steps closed loop from a perturbed hover" (the timed region never degenerates into the
converged, bound-free regime).  Instances are independent: the fleet shards across GPUs with no
data-path collective; RCCL only aggregates the report (max time, statistics).

Scaling: the metric's batch is 65 536 IN TOTAL, so for N > 1 the default is `--scaling strong`
(65 536 instances split into contiguous shards, 65 536 / N per GPU; `value` = that run) and the
weak-scaling figure (65 536 per GPU) is measured right after it and reported beside it under
`weak_scaling`.  `--scaling weak` makes the weak run the timed one.

Also on the same line (rank 0, N = 1):
  roofline     -- one RTI step's algorithmic bytes / the duration of its kernels, measured with HIP
                  events on the launch stream over the SAME K timed steps;
  cpu_baseline -- BASELINE.md section 3: the plain-C CPU restatement of the same algorithm
                  (oracle/cfnmpc_ref.c, "port": acados itself cannot be built here) on this host's
                  cores: B-lat (1 instance, 1000 steps, median / p99), B-thr (4096 instances x 20
                  steps, all usable cores), B-mix (N in {30, 50, 100});
  sensitivity  -- what the headline depends on: interior point only (active_set = 0), harder
                  disturbances (kick scale x2, x3 -> larger constrained fraction), config C2
                  (batch 4096), config C4 (figure-8 tracking), config C5 (mixed horizons, delay-
                  compensated x0).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12         # B/s, MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
FP64_VECTOR_PEAK = 78.6e12   # half the 157.3 TFLOP/s FP32 vector peak of MI355X_MICROARCH.md (256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz)
FP64_VEC_PEAK = 78.6e12   # FLOP/s, FP64 vector (= FP64 matrix) peak of MI355X
N_HORIZON = 50
KICK_PERIOD = 20
TOTAL_BATCH = 65536       # BASELINE.json metric
DEADLINE_MS = 15.0        # the reference's control period (acados_estimator.cpp:642: 66.6 Hz)


def alg_bytes_step(N):
    """algorithmic bytes per RTI step per instance, 8*(553 N + 80)."""
    return 8 * (553 * N + 80)


def alg_bytes_qp(N):
    """Share of the QP kernels: stage blocks read once (251 N + 13 words) +
    x0, yref, iterate read, iterate write, status (51 N + 54 words)."""
    return 8 * (302 * N + 67)


# ---------------------------------------------------------------------------------------------
# CPU baseline -- runs in a child process without torch so that the OpenMP
# runtime of the restatement starts with pinned threads (OMP_PROC_BIND / OMP_PLACES)
# ---------------------------------------------------------------------------------------------
def usable_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (GPU boxes
    expose all 256 hardware threads of the host but grant a fraction of them as CPU time; running one
    OpenMP thread per visible CPU then spends the run being throttled)."""
    try:   # the parent (HIP / torch runtimes loaded) may hand down a narrowed mask: widen it as far as the cpuset allows
        os.sched_setaffinity(0, range(os.cpu_count() or 1))
    except Exception:
        pass
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())       # cgroup v1
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def _cpu_baseline_child(seed):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cfnmpc_oracle as o
    import cref
    cref.build()          # rebuilds when the library was compiled on another host (-march=native is the BUILD host's)
    march = cref.march_native()
    N = N_HORIZON
    yr, ye = o.regulation_yref(N, (0.0, 0.0, 0.4))
    opts = cref.default_opts(active_set=1)          # same QP method as the engine's default
    cores = os.cpu_count() or 1
    nthr, quota = usable_cpus()
    rng = np.random.default_rng(seed)
    # B-lat: one instance, 1000 consecutive closed-loop RTI steps on one core; every 20th step the
    # vehicle is kicked like the fleet is (otherwise 980 of the steps would be converged ones)
    lat, bad = [], 0
    for c in range(50):
        x = o.sample_hover_x0(rng, 1)
        r = cref.closed_loop(opts, x, yr[None].copy(), ye[None].copy(), KICK_PERIOD, nthreads=1, latencies=True)
        lat.append(r["lat_us"]); bad += r["bad"]
    lat = np.sort(np.concatenate(lat))
    # B-thr: first 4096 instances of config C2, 20 closed-loop steps each, all host cores
    B = 4096
    x2 = o.sample_hover_x0(np.random.default_rng(seed), B)
    yref = np.repeat(yr[None], B, 0).copy(); yref_e = np.repeat(ye[None], B, 0).copy()
    best = None
    for _ in range(3):       # shared hosts: best of three
        r = cref.closed_loop(opts, x2.copy(), yref, yref_e, KICK_PERIOD, nthreads=nthr)
        if best is None or r["seconds"] < best["seconds"]:
            best = r
    thr = B * KICK_PERIOD / best["seconds"]
    one = None
    for _ in range(3):
        r1 = cref.closed_loop(opts, x2[:64].copy(), yref[:64], yref_e[:64], KICK_PERIOD, nthreads=1)
        one = r1 if one is None or r1["seconds"] < one["seconds"] else one
    thr1 = 64 * KICK_PERIOD / one["seconds"]
    # B-mix: 4096 instances of config C5 (N in {30, 50, 100}, delay-compensated x0)
    hz = np.random.default_rng(seed + 2).choice([30, 50, 100], size=B)
    xm = cref.sim(x2, np.full((B, 4), o.HOV_W), 0.06, 4)
    t_mix, stage_steps = 0.0, 0
    for n in (30, 50, 100):
        idx = np.nonzero(hz == n)[0]
        yrn, yen = o.regulation_yref(int(n), (0.0, 0.0, 0.4))
        rr = cref.closed_loop(cref.default_opts(N=int(n), active_set=1), xm[idx].copy(), np.repeat(yrn[None], len(idx), 0).copy(),
                              np.repeat(yen[None], len(idx), 0).copy(), KICK_PERIOD, nthreads=nthr)
        t_mix += rr["seconds"]; stage_steps += len(idx) * int(n) * KICK_PERIOD
    out = {
        "value": thr, "unit": "RTI steps/s", "cores": int(best["threads"]), "host_cores": int(cores), "cpu_quota": quota, "kind": "port",
        "per_core_steps_per_s": thr / best["threads"], "single_thread_steps_per_s": thr1,
        "parallel_efficiency": thr / (best["threads"] * thr1),
        "latency_us": {"median": float(lat[len(lat) // 2]), "p99": float(lat[int(len(lat) * 0.99)]), "max": float(lat[-1]),
                       "steps": int(len(lat)), "budget_us": 15000.0, "bad_status": int(bad)},
        "mixed_horizon_steps_per_s": B * KICK_PERIOD / t_mix, "mixed_horizon_stage_steps_per_s": stage_steps / t_mix,
        "mean_qp_solves": best["iters"] / (B * KICK_PERIOD),
        "march_native": march,
        "sample": (f"oracle/cfnmpc_ref.c (CPU restatement, NOT acados; gcc -O3 -march=native [= {march} on this host, compiled here] -fopenmp, FP64, same QP method as the "
                   f"engine: active-set solves, interior point as fall-back). B-thr: first {B} instances of config C2 x {KICK_PERIOD} "
                   f"closed-loop RTI steps, one instance at a time per thread, {best['threads']} threads = the CPUs this process may use "
                   f"(affinity mask capped by the cgroup CPU quota; host has {cores}), best of 3. B-lat: 1 instance, {len(lat)} closed-loop steps on one core "
                   f"(kicked every {KICK_PERIOD} steps). B-mix: {B} instances N in {{30,50,100}}, delay-compensated x0."),
    }
    print(json.dumps(out))


def cpu_baseline(seed):
    env = dict(os.environ, OMP_PROC_BIND="false", OMP_DYNAMIC="false")   # measured on the GPU box: unbound threads under the CPU quota are fastest
    for k in ("OMP_NUM_THREADS", "OMP_PLACES", "GOMP_CPU_AFFINITY"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", str(seed)], env=env,
                       capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-500:])
    return json.loads(r.stdout.strip().splitlines()[-1])


# ---------------------------------------------------------------------------------------------
# one closed-loop fleet on this rank's GPU
# ---------------------------------------------------------------------------------------------
class Fleet:
    """`n` vehicles: own solver object, device-resident plant state, staggered kicks."""

    def __init__(self, n, dev, rng, workload="hover", kick_scale=1.0, stream=None, **opt_kw):
        import torch
        from crazyflie_nmpc_amd import BatchSolver, default_opts
        from crazyflie_nmpc_amd.solver import INIT_HOVER
        from crazyflie_nmpc_amd.synthetic import regulation_row, sample_hover_x0 as sample_x0
        N = N_HORIZON
        self.n, self.dev = n, dev
        self.x = torch.from_numpy(sample_x0(rng, n, scale=kick_scale)).to(dev)
        self.xn = torch.empty_like(self.x)
        self.u0 = torch.empty((n, 4), dtype=torch.float64, device=dev)
        self.cohort = (n + KICK_PERIOD - 1) // KICK_PERIOD
        self.kicks = torch.from_numpy(sample_x0(rng, self.cohort * KICK_PERIOD, scale=kick_scale)
                                      .reshape(KICK_PERIOD, self.cohort, 13)).to(dev)
        self.solver = BatchSolver(n, default_opts(**opt_kw))
        self.track = workload == "figure8"
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
            row = regulation_row((0.0, 0.0, 0.4))
            yref = torch.from_numpy(np.tile(row, (n, N, 1))).to(dev)
            yref_e = torch.from_numpy(np.tile(row[:13], (n, 1))).to(dev)
            self.solver.set_yref(yref, yref_e)
            del yref
        self.solver.set_x0(self.x)
        self.solver.init_iterate(INIT_HOVER)
        self.t = 0
        self.stream = (stream or torch.cuda.current_stream(dev)).cuda_stream

    def step(self):
        import torch
        from crazyflie_nmpc_amd import sim
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
        self.solver.solve(1, self.stream)                    # acados_solve()  (acados_mpc.cpp:611)
        self.solver.get_u(0, out=self.u0)                    # ocp_nlp_out_get(.., 0, "u")  (:619)
        sim(xc, self.u0, T=0.015, steps=1, out=self.xn)      # plant: one RK4 step of the ODE
        self.x, self.xn = self.xn, xc
        self.t = t + 1

    def close(self):
        self.solver.close()


class MixedFleetLoop:
    """Config C5 closed loop on this rank's GPU: vehicles with N in {30, 50, 100} (one cfnmpc_fleet = one solver per horizon
    bucket behind one handle), regulation targets U(-1,1)^2 x U(0.2,1); the plant applies every input 60 ms
    = 4 sampling periods after it was computed (the communication delay the reference compensates,
    acados_mpc.cpp:624, launch/acados_predictor.launch:62) and x0 is the prediction of the state over that
    delay: the predictor kernel (crazyflie_acados_sim_solve's batch form, acados_estimator.cpp:573-593)
    applied to the four inputs in flight, oldest first.  (The reference's estimator holds the LATEST input over
    the whole delay; with raw motor speeds as plant inputs -- no onboard attitude loop in between -- that
    approximation destabilises the closed loop, measured: the fleet diverges within 60 steps.)  Closed loop,
    staggered kicks (a kicked vehicle restarts with hover inputs in flight; keeping stale inputs in flight
    across the jump makes a handful of vehicles per step fall back to 30 - 50 interior-point iterations,
    which then set the duration of the whole fleet's step: 15 - 24 ms instead of 5.8 ms, measured).
    predictor = "latest": the REFERENCE's protocol instead -- x0 = one crazyflie_acados_sim_solve over the whole 60 ms with the
    latest input held (acados_estimator.cpp:573-593: sim_in_set "T" = delay, "x", "u" = the last published motor speeds), same
    plant.  `horizons` [n] are this rank's vehicles (for N > 1 GPUs: its part of parallel.shard_by_horizon)."""

    def __init__(self, horizons, dev, rng, predictor="queued", **opt_kw):
        import torch
        from crazyflie_nmpc_amd.fleet import MixedHorizonFleet
        from crazyflie_nmpc_amd.solver import INIT_HOVER
        from crazyflie_nmpc_amd.synthetic import HOV_W, sample_hover_x0
        self.horizons = np.ascontiguousarray(horizons, dtype=np.int32)
        self.n = n = len(self.horizons)
        self.dev, self.predictor, self.hov = dev, predictor, HOV_W
        self.fleet = MixedHorizonFleet(self.horizons, **opt_kw)
        tgt = np.concatenate([rng.uniform(-1, 1, (n, 2)), rng.uniform(0.2, 1.0, (n, 1))], axis=1)
        self.fleet.set_regulation(tgt, HOV_W)
        off = np.concatenate([tgt - [0.0, 0.0, 0.4], np.zeros((n, 10))], axis=1)
        self.x = torch.from_numpy(sample_hover_x0(rng, n) + off).to(dev)
        self.xn, self.xp = torch.empty_like(self.x), torch.empty_like(self.x)
        self.u0 = torch.full((n, 4), HOV_W, dtype=torch.float64, device=dev)
        self.uq = [self.u0.clone() for _ in range(4)]           # inputs in flight: computed at t - 4 .. t - 1
        self.cohort = (n + KICK_PERIOD - 1) // KICK_PERIOD
        self.kicks = torch.from_numpy(sample_hover_x0(rng, self.cohort * KICK_PERIOD).reshape(KICK_PERIOD, self.cohort, 13)).to(dev)
        self.offd = torch.from_numpy(off).to(dev)
        self.fleet.set_x0(self.x); self.fleet.init_iterate(INIT_HOVER)
        self.t = 0

    def step(self):
        from crazyflie_nmpc_amd import sim
        t, x, uq = self.t, self.x, self.uq
        c0 = (t % KICK_PERIOD) * self.cohort
        c1 = min(c0 + self.cohort, self.n)
        if c1 > c0:   # a kicked vehicle is a fresh one: new state, hover inputs in flight
            x[c0:c1].copy_(self.kicks[t % KICK_PERIOD, : c1 - c0] + self.offd[c0:c1])
            for q in uq:
                q[c0:c1] = self.hov
        if self.predictor == "none":     # no delay at all (isolates what the horizon buckets cost: same loop as config C3's)
            self.fleet.set_x0(x); self.fleet.solve(1); self.fleet.get_u(0, self.u0)
            sim(x, self.u0, T=0.015, steps=1, out=self.xn)
            self.x, self.xn = self.xn, x
            self.t = t + 1
            return
        if self.predictor == "latest":   # the reference's predictor: the latest input held over the delay, RK4 with 4 sub-steps (App. D-8)
            sim(x, uq[(t + 3) % 4], T=0.06, steps=4, out=self.xp)
        else:
            sim(x, uq[t % 4], T=0.015, steps=1, out=self.xp)   # delay compensation: through the four inputs in flight
            for j in (1, 2, 3):
                sim(self.xp, uq[(t + j) % 4], T=0.015, steps=1, out=self.xn)
                self.xp, self.xn = self.xn, self.xp
        self.fleet.set_x0(self.xp); self.fleet.solve(1); self.fleet.get_u(0, self.u0)
        sim(x, uq[t % 4], T=0.015, steps=1, out=self.xn)   # the plant sees the inputs computed 4 periods ago
        uq[t % 4].copy_(self.u0)
        self.x, self.xn = self.xn, x
        self.t = t + 1

    def stats(self):
        return self.fleet.stats()

    def close(self):
        self.fleet.close()


def mixed_horizon_run(batch, dev, rng, steps, warmup, predictor="queued"):
    """Config C5 on one GPU for the `sensitivity` block (see MixedFleetLoop).  predictor = "latest" (the reference's protocol)
    is reported with how long that loop stays healthy (`steps_until_ok_frac_below_0.99`)."""
    import torch
    horizons = rng.choice([30, 50, 100], size=batch)
    loop = MixedFleetLoop(horizons, dev, rng, predictor)
    step = loop.step
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    st, it, _ = loop.stats()
    out = {"value": batch * steps / el, "stage_steps_per_s": float(horizons.sum()) * steps / el, "ms_per_step": el / steps * 1e3,
           "frac_constrained": float((it > 0).mean()), "mean_qp_solves": float(it.mean()), "status_ok_frac": float((st == 0).mean()),
           "buckets": {int(n): int((horizons == n).sum()) for n in (30, 50, 100)},
           "deadline_15ms_ok": bool(el / steps * 1e3 <= DEADLINE_MS)}   # (mean closed-loop step; the buckets run concurrently, no per-step events)
    if predictor == "latest":
        # how long the reference's protocol keeps this plant (raw motor speeds, no onboard attitude loop) healthy: the loop goes
        # on untimed, checked every 5 steps, until fewer than 99 % of the vehicles end their step with status 0 or a state
        # leaves every plausible range (|p| > 50 m)
        survived = warmup + steps
        while survived < 400:
            for _ in range(5):
                step()
            survived += 5
            st2 = loop.stats()[0]
            if float((st2 == 0).mean()) < 0.99 or not bool(torch.isfinite(loop.x).all()) or float(loop.x[:, :3].abs().max()) > 50.0:
                break
        out["steps_until_ok_frac_below_0.99"] = survived if survived < 400 else ">= 400"
        out["status_ok_frac_at_the_end"] = float((loop.stats()[0] == 0).mean())
    loop.close()
    del loop
    torch.cuda.empty_cache()
    return out


def timed_run(fleet, steps, warmup, barrier, profile=True):
    """W untimed + exactly K timed steps between barrier + synchronize; the step's kernels are
    bracketed by HIP events on the launch stream during the SAME K steps (cfnmpc_set_profiling).
    profile = False: no events in the timed region (the per-kernel split is then all zeros).
    -> (elapsed seconds, ms linearise, ms qp, stats of the last step)"""
    for _ in range(warmup):
        fleet.step()
    fleet.solver.set_profiling(bool(profile))
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fleet.step()
    barrier()
    elapsed = time.perf_counter() - t0
    per = fleet.solver.get_profile_steps()   # [timed step][linearise | factor | forward | compaction | active set | interior point]
    kms = [float(v) for v in per.mean(axis=0)] if len(per) else [0.0] * 6
    ms_lin, ms_qp = kms[0], sum(kms[1:])
    # the active-set kernels per STEP, not only on average: a step in which a tail check fails (re-solve over a longer head
    # inside the wave) takes a multiple of the mean, and a deadline sees the worst step
    as_steps = np.sort(per[:, 4] + per[:, 5]) if len(per) else np.zeros(1)
    step_kernels = np.sort(per.sum(axis=1)) if len(per) else np.zeros(1)
    pct = lambda a: {"p50": float(a[len(a) // 2]), "p99": float(a[min(len(a) - 1, int(np.ceil(0.99 * len(a))) - 1)]), "max": float(a[-1]), "steps": int(len(a))}
    if getattr(fleet.solver.opts, "cond_N2", 0):
        # condensed steps report their two phases in kms[0] / kms[5] only (include/cfnmpc.h: cfnmpc_get_profile_kernels): no
        # per-kernel split for them
        ms_lin, ms_qp = kms[0], sum(kms[1:])
        kms = [0.0] * 6
    fleet.solver.set_profiling(False)
    # (n_prof == steps unless K exceeds the library's cap of timed steps: the average then covers the first 4096)
    st, it, _rs = fleet.solver.stats()
    heads = fleet.solver.heads()
    return elapsed, ms_lin, ms_qp, dict(ok=float((st == 0).sum()), bad=float((st != 0).sum()), solves=float(it.sum()),
                                        constrained=float((it > 0).sum()), heads=float(heads.sum()), kms=kms,
                                        as_pct=pct(as_steps), step_pct=pct(step_kernels))


def _relaunch_under_torchrun(n, backend, ndev):
    """exec torch.distributed.run with the same argv: --nnodes=1, one rank per GPU, rendezvous on 127.0.0.1 (the container's
    hostname may not resolve).  Refuses (non-zero exit) when RCCL would need more devices than are visible."""
    if backend == "nccl" and ndev < n:
        raise SystemExit(f"bench.py: --gpus {n} with the nccl (RCCL) backend needs {n} visible HIP devices, found {ndev} "
                         "(--dist-backend gloo lets ranks share devices for functional checks only)")
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this host driver
    os.execv(sys.executable, cmd)


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--cpu-baseline-child":
        return _cpu_baseline_child(int(sys.argv[2]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=TOTAL_BATCH,
                    help="instances: in total with --scaling strong, per GPU with --scaling weak")
    ap.add_argument("--scaling", choices=["strong", "weak"], default=None,
                    help="strong (default for N > 1): --batch instances split over the GPUs; weak: --batch per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="no HIP events around the kernels of the timed steps (roofline fields from kernel time are then void)")
    ap.add_argument("--no-extras", action="store_true", help="skip the sensitivity runs (and the weak run beside a strong one)")
    ap.add_argument("--active-horizon", type=int, default=1)
    ap.add_argument("--active-set", type=int, default=1)
    ap.add_argument("--cond-n2", type=int, default=None, help="cfnmpc_opts.cond_N2 (partial condensing; default: library default)")
    ap.add_argument("--kick-scale", type=float, default=1.0)
    ap.add_argument("--step-graph", type=int, default=None, help="cfnmpc_opts.step_graph (captured hipGraph per RTI step)")
    ap.add_argument("--forward-sweep", type=int, default=None, help="cfnmpc_opts.forward_sweep (0 auto, 1 matrix-free, 2 row groups)")
    ap.add_argument("--as-passes", type=int, default=None, help="cfnmpc_opts.as_passes (scheduling of the active-set solves: 0 auto, -1 monolithic, "
                                                                "-3 solves + commit kernel)")
    ap.add_argument("--as-warm", type=int, default=None, help="cfnmpc_opts.as_warm (warm start of the active set from the previous RTI step)")
    ap.add_argument("--as-dense", type=int, default=None, help="cfnmpc_opts.as_dense (head-condensed dense active-set solves: 1 on, -1 off, 0 auto)")
    ap.add_argument("--forward-split", type=int, default=None, help="cfnmpc_opts.forward_split (1 on, -1 off, 0 auto)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl = RCCL over xGMI (default); gloo only for functional checks of the N > 1 path on "
                         "a box with fewer GPUs than ranks (ranks then share devices)")
    ap.add_argument("--workload", choices=["hover", "figure8", "mixed"], default="hover",
                    help="hover = config C3 (the metric's configuration); figure8 = config C4 tracking with "
                         "device-side reference windows; mixed = config C5: horizons N in {30, 50, 100}, delay-compensated x0, "
                         "vehicles dealt out over the GPUs by horizon bucket so that sum N is balanced (parallel.shard_by_horizon)")
    ap.add_argument("--predictor", choices=["queued", "latest"], default="queued",
                    help="--workload mixed: delay compensation through the four queued inputs (default) or the reference's "
                         "latest-input-held predictor (acados_estimator.cpp:573-593)")
    ap.add_argument("--ah-margin", type=float, default=None)
    ap.add_argument("--ah-extra", type=int, default=None)
    ap.add_argument("--start-solve", type=int, default=None, help="cfnmpc_opts.start_solve (0 auto, 1 k_linearise + k_factor on stored blocks, "
                    "2 fused k_linfactor, 3 fused factorisation + stored blocks)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the engine has no CPU path)")
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` WITHOUT a launcher: become `python -m torch.distributed.run --nproc-per-node N bench.py <same
        # argv>` (one rank per GPU; rank 0's JSON line goes to the same stdout) -- `--gpus N` means N GPUs whoever starts it
        _relaunch_under_torchrun(args.gpus, args.dist_backend, torch.cuda.device_count())
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:   # never print a line whose n_gpus differs from what was asked for
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch one rank per GPU, or plain `python bench.py --gpus N`)")
    ndev = torch.cuda.device_count()
    if args.dist_backend == "nccl" and ndev < world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} with the nccl (RCCL) backend needs {world} visible HIP devices, found {ndev} "
                         "(--dist-backend gloo lets ranks share devices for functional checks only)")
    if args.dist_backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible")
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    dist = None
    red_dev = dev  # device of the tiny reduction tensors
    # (under torchrun with ONE rank the process group is brought up as well: the report's two all-reduces then run on RCCL
    #  with device tensors -- the 8-GPU run is not the collective's first contact; tests/test_gpu_bench_distributed.py)
    launched = world > 1 or ("RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ)
    if launched:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            red_dev = torch.device("cpu")

    from crazyflie_nmpc_amd import parallel
    N = N_HORIZON
    scaling = args.scaling or "strong"
    seed = parallel.shard_seed(rank)
    opt_kw = dict(active_horizon=args.active_horizon, active_set=args.active_set)
    for k, v in (("ah_margin", args.ah_margin), ("ah_extra", args.ah_extra), ("cond_N2", args.cond_n2),
                 ("step_graph", args.step_graph), ("forward_sweep", args.forward_sweep), ("as_passes", args.as_passes),
                 ("start_solve", args.start_solve), ("as_warm", args.as_warm), ("as_dense", args.as_dense), ("forward_split", args.forward_split)):
        if v is not None:
            opt_kw[k] = v

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def measure(batch_rank, steps, warmup, workload=args.workload, kick_scale=args.kick_scale, seed_off=0, profile=True, **kw):
        """One closed-loop run of `batch_rank` vehicles on this rank; aggregated over ranks."""
        okw = dict(opt_kw); okw.update(kw)
        fleet = Fleet(batch_rank, dev, np.random.default_rng(seed + 1000 * seed_off), workload, kick_scale, **okw)
        torch.cuda.synchronize(dev)
        elapsed, ms_lin, ms_qp, st = timed_run(fleet, steps, warmup, barrier, profile and not args.no_profile)
        if not profile and not args.no_profile:
            # (sensitivity runs: the K timed steps carry NO events -- seven event records per step cost ~30 us, 4 - 8 % of a step
            #  at 2048 - 8192 instances, 0.5 % at 65 536 --; the per-kernel split comes from 10 more steps with them)
            _el, ms_lin, ms_qp, st2 = timed_run(fleet, 10, 0, barrier, True)
            st["kms"] = st2["kms"]; st["as_pct"] = st2["as_pct"]; st["step_pct"] = st2["step_pct"]
        fleet.close()
        del fleet
        torch.cuda.empty_cache()
        sums = [st["ok"], st["bad"], st["solves"], st["constrained"], st["heads"], ms_lin, ms_qp, float(batch_rank)] + list(st["kms"])
        elapsed, sums = parallel.aggregate_report(elapsed, sums, dist, red_dev)
        tot = sums[7]
        return dict(elapsed=elapsed, total=tot, value=tot * steps / elapsed, ms_per_step=elapsed / steps * 1e3,
                    ms_lin=sums[5] / world, ms_qp=sums[6] / world, ok_frac=sums[0] / tot, mean_qp_solves=sums[2] / tot,
                    frac_constrained=sums[3] / tot, mean_head=sums[4] / tot, batch_rank=batch_rank,
                    kms=[float(v) / world for v in sums[8:14]], as_pct=st["as_pct"], step_pct=st["step_pct"])   # (percentiles: this rank's)

    if args.workload == "mixed":
        # config C5 across the GPUs (SURVEY.md section 8e "Partitioning": bucket by N, then balance the buckets over the GPUs
        # by sum N_i): ONE fleet-wide horizon draw (same on every rank), dealt out by parallel.shard_by_horizon; every rank runs
        # its vehicles as one cfnmpc_fleet (a solver per horizon bucket).  No data-path collective; the report travels as usual.
        total = args.batch if scaling == "strong" else args.batch * world
        hz_all = np.random.default_rng(parallel.BASE_SEED + 5005).choice([30, 50, 100], size=total)
        mine = parallel.shard_by_horizon(hz_all, world)[rank]
        hz = hz_all[mine]
        loop = MixedFleetLoop(hz, dev, np.random.default_rng(seed + 9000), args.predictor, **opt_kw)
        torch.cuda.synchronize(dev)
        for _ in range(args.warmup):
            loop.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loop.step()
        barrier()
        elapsed = time.perf_counter() - t0
        st, it, _rs = loop.stats()
        loop.close()
        slots = np.zeros((world, 4))          # per-rank: sum N, bucket sizes -- one slot per rank, summed over ranks = gathered
        slots[rank] = [float(hz.sum())] + [float((hz == n).sum()) for n in (30, 50, 100)]
        sums = [float((st == 0).sum()), float(it.sum()), float((it > 0).sum()), float(len(hz)), float(hz.sum())] + slots.ravel().tolist()
        elapsed, sums = parallel.aggregate_report(elapsed, sums, dist, red_dev)
        if rank == 0:
            tot, sumN = sums[3], sums[4]
            per_rank = sums[5:].reshape(world, 4)
            alg = float(sum(alg_bytes_step(int(n)) * per_rank[:, 1 + j].sum() for j, n in enumerate((30, 50, 100))))
            ms = elapsed / args.steps * 1e3
            out = {
                "metric": "NMPC RTI steps/sec (batch=65536, N=50, nx=13, nu=4)", "value": tot * args.steps / elapsed, "unit": "RTI steps/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
                "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "stage_steps_per_s": sumN * args.steps / elapsed,
                "config": {"workload": "C5 mixed horizons N in {30, 50, 100}, delay-compensated x0 (60 ms, predictor: "
                                       + ("the four queued inputs, oldest first" if args.predictor == "queued" else
                                          "the reference's: latest input held, acados_estimator.cpp:573-593")
                                       + f"), closed loop through the RK4 plant, staggered kicks (1/{KICK_PERIOD} of the fleet per step)",
                           "total_batch": int(tot), "horizons": {str(n): int(per_rank[:, 1 + j].sum()) for j, n in enumerate((30, 50, 100))},
                           "nx": 13, "nu": 4,
                           "sharding": (f"{scaling} scaling: {int(tot)} vehicles bucketed by horizon and dealt out over {world} GPU(s) so that sum N "
                                        "is balanced (parallel.shard_by_horizon), one solver per bucket and rank, no data-path collective"),
                           "per_rank": [{"rank": r, "sum_N": int(per_rank[r, 0]), "vehicles": int(per_rank[r, 1:].sum()),
                                         "buckets": {str(n): int(per_rank[r, 1 + j]) for j, n in enumerate((30, 50, 100))}} for r in range(world)],
                           "sum_N_imbalance": float((per_rank[:, 0].max() - per_rank[:, 0].min()) / per_rank[:, 0].mean())},
                # no per-kernel events here: the buckets' steps run concurrently on forked streams, so the model is set against the
                # wall clock of the whole closed-loop step (plant, predictor, row gathers included)
                "roofline": {"bound": "hbm", "kernel": "whole closed-loop step (wall clock; buckets run concurrently on forked streams)",
                             "achieved": alg / world / (ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                             "frac": alg / world / (ms * 1e-3) / HBM_PEAK, "traffic": None, "alg_bytes_per_step_all_gpus": alg},
                "qp_stats": {"status_ok_frac": sums[0] / tot, "mean_qp_solves": sums[1] / tot, "frac_constrained": sums[2] / tot},
            }
            if dist is not None:
                out["report_collective"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "device": red_dev.type}
            print(json.dumps(out))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if scaling == "strong":
        lo, hi = parallel.shard_range(args.batch, rank, world)
        B_rank = hi - lo
    else:
        B_rank = args.batch
    # N = 1: HIP events around the kernels of the K timed steps themselves (the roofline's kernel time is measured over the timed
    # region).  N > 1: the timed steps carry no events -- seven records per step cost ~30 us, 4 % of a step at the 8192 instances
    # per GPU of the 8-GPU split -- and the per-kernel split comes from 10 more steps with them (roofline.kernel_ms_source)
    main_run = measure(B_rank, args.steps, args.warmup, profile=(world == 1))
    weak = None
    if world > 1 and scaling == "strong" and not args.no_extras:
        weak = measure(args.batch, args.steps, args.warmup, seed_off=1, profile=False)

    extras = {}
    if world == 1 and not args.no_extras and args.batch == TOTAL_BATCH and args.workload == "hover" and args.kick_scale == 1.0 \
            and args.active_set == 1 and args.cond_n2 is None and args.as_passes is None:
        ws = min(args.warmup, 20)

        def brief(r):
            return {"value": r["value"], "ms_per_step": r["ms_per_step"], "kernel_ms": r["ms_lin"] + r["ms_qp"],   # (kernel_ms: 10 steps WITH events after the timed ones)
                    "frac_constrained": r["frac_constrained"], "mean_qp_solves": r["mean_qp_solves"], "status_ok_frac": r["ok_frac"],
                    # the reference's control period: acados_estimator.cpp:642 runs the loop at 66.6 Hz -- a step of the whole
                    # fleet (mean, and the slowest of the 10 event-timed steps) against those 15 ms
                    "deadline_15ms_ok": bool(r["ms_per_step"] <= DEADLINE_MS and r["step_pct"]["max"] <= DEADLINE_MS),
                    "step_kernels_ms_max": r["step_pct"]["max"], "active_set_group_ms_per_step": r["as_pct"]}
        extras["interior_point_only (active_set=0)"] = brief(measure(B_rank, 20, ws, active_set=0, seed_off=2, profile=False))
        extras["kick_scale_x2"] = brief(measure(B_rank, 20, ws, kick_scale=2.0, seed_off=3, profile=False))
        extras["kick_scale_x3"] = brief(measure(B_rank, 20, ws, kick_scale=3.0, seed_off=4, profile=False))
        # cfnmpc_opts.reinit_failed: vehicles whose QP failed (status 4) restart from their current state instead of keeping
        # the iterate that failed -- NOT the reference's behaviour (its node ignores the status), reported beside it
        extras["kick_scale_x2 + reinit_failed"] = brief(measure(B_rank, 20, ws, kick_scale=2.0, seed_off=3, reinit_failed=1, profile=False))
        extras["kick_scale_x3 + reinit_failed"] = brief(measure(B_rank, 20, ws, kick_scale=3.0, seed_off=4, reinit_failed=1, profile=False))
        extras["full_horizon_sweeps (active_horizon=0)"] = brief(measure(B_rank, 20, ws, active_horizon=0, seed_off=5, profile=False))
        extras["config_C2_batch_4096"] = brief(measure(4096, 40, 40, seed_off=6, profile=False))
        extras["batch_8192 (one GPU's share of 65536 at 8 GPUs)"] = brief(measure(8192, 40, 40, seed_off=7, profile=False))
        extras["batch_16384 (one GPU's share of 65536 at 4 GPUs)"] = brief(measure(16384, 40, 40, seed_off=9, profile=False))
        extras["batch_32768 (one GPU's share of 65536 at 2 GPUs)"] = brief(measure(32768, 40, 40, seed_off=10, profile=False))
        # The metric is quoted "at 1/2/4/8 MI355X" for 65 536 instances IN TOTAL (strong scaling: contiguous shards, no data-path
        # collective).  No multi-GPU node has been available to any run of any round, so this is the line's only
        # driver-timed evidence for N > 1: N GPUs x the measured single-GPU rate at 65 536 / N instances -- a PREDICTION (it
        # leaves out nothing but the two tiny report all-reduces, which are outside the timed steps' data path).
        shard = {2: "batch_32768 (one GPU's share of 65536 at 2 GPUs)", 4: "batch_16384 (one GPU's share of 65536 at 4 GPUs)",
                 8: "batch_8192 (one GPU's share of 65536 at 8 GPUs)"}
        extras["predicted_strong_scaling"] = {
            "what": "N x (single-GPU closed-loop rate measured in THIS run at 65536 / N instances); predicted, not measured on N GPUs",
            "n_gpus": {str(n): {"instances_per_gpu": TOTAL_BATCH // n, "value": n * extras[k]["value"], "unit": "RTI steps/s",
                                "ms_per_step": extras[k]["ms_per_step"],
                                "efficiency_vs_1gpu": n * extras[k]["value"] / (n * main_run["value"])} for n, k in shard.items()}}
        extras["config_C4_figure8_tracking"] = brief(measure(B_rank, 20, ws, workload="figure8", seed_off=8, profile=False))
        extras["config_C5_mixed_horizons_30_50_100_delay_compensated"] = mixed_horizon_run(B_rank, dev, np.random.default_rng(seed + 9000), 20, ws)
        extras["config_C5_mixed_horizons_30_50_100_delay_compensated"]["predictor"] = (
            "x0 = RK4 prediction over the 60 ms delay THROUGH THE FOUR QUEUED INPUTS (oldest first); the reference's estimator holds the "
            "LATEST input over the delay (acados_estimator.cpp:573-593) -- with raw motor speeds as plant inputs that closed loop diverges "
            "(no onboard attitude loop in this plant), hence the departure")
        # what of C5's distance to the uniform fleet is the BUCKETS and what the workload: the same mixed-horizon fleet in config
        # C3's loop (no delay, targets as drawn) -- stage-steps/s against the headline's value x 50
        c5p = mixed_horizon_run(B_rank, dev, np.random.default_rng(seed + 9000), 20, ws, predictor="none")
        c5p["predictor"] = "none: no delay, x0 = the plant state (config C3's loop on the mixed-horizon fleet)"
        c5p["stage_steps_vs_uniform_fleet"] = c5p["stage_steps_per_s"] / (main_run["value"] * N)
        extras["config_C5_mixed_horizons_30_50_100_no_delay (cost of the horizon buckets alone)"] = c5p
        # ... and the reference's own protocol beside it, for as long as it holds on this plant
        c5r = mixed_horizon_run(B_rank, dev, np.random.default_rng(seed + 9000), 20, ws, predictor="latest")
        c5r["predictor"] = ("the reference's: x0 = ONE sim solve over the 60 ms delay with the LATEST input held (acados_estimator.cpp:573-593, "
                            "launch/acados_predictor.launch:62); timed over the same 20 steps after the same warm-up, then run on until "
                            "the loop degrades (steps_until_ok_frac_below_0.99)")
        extras["config_C5_mixed_horizons_30_50_100_reference_predictor (latest input held)"] = c5r

    if rank == 0:
        r = main_run
        B_launch = r["batch_rank"]          # instances one launch of this rank's kernels covers
        ms_step = r["ms_lin"] + r["ms_qp"]
        # same K steps: the kernels are part of the step, so kernel_ms <= ms_per_step (ranks sharing one GPU in a
        # functional run can interleave and break this: reported, never fatal)
        kernel_time_consistent = bool(ms_step <= r["ms_per_step"] * 1.001)
        ach = alg_bytes_step(N) * B_launch / (ms_step * 1e-3)
        ach_qp = alg_bytes_qp(N) * B_launch / (r["ms_qp"] * 1e-3)
        traffic = traffic_qp = None
        tsrc = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if int(tj.get("batch", -1)) == B_launch:
                    traffic_qp = tj.get("hbm_bytes_per_launch_k_qp")
                    traffic = tj.get("hbm_bytes_per_step")
                    if traffic is None and traffic_qp is not None:
                        traffic = traffic_qp + tj["kernels"]["cfn::k_linearise"]["total_bytes"]
                    tsrc = ("profiles/pmc_traffic_latest.json (STATIC: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                            "tools/profile_round.sh on the builder's box, same command line; not measured in this run)")
            except Exception:
                traffic = traffic_qp = None
        out = {
            "metric": "NMPC RTI steps/sec (batch=65536, N=50, nx=13, nu=4)",
            "value": r["value"], "unit": "RTI steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("C3 hover regulation" if args.workload == "hover" else
                                    "C4 figure-8 tracking (device-side reference windows)") +
                                   ", closed loop through the RK4 plant, staggered kicks "
                                   f"(1/{KICK_PERIOD} of the fleet per step, scale {args.kick_scale:g})",
                       "total_batch": int(r["total"]), "batch_per_gpu": B_launch, "horizon_N": N, "nx": 13, "nu": 4,
                       "sharding": (f"{scaling} scaling: " + (f"{args.batch} instances split into {world} contiguous shard(s)" if scaling == "strong"
                                                             else f"{args.batch} instances per GPU") + ", no data-path collective"),
                       "qp": (("primal-dual active-set solves on the Riccati factorisation (exact, KKT-verified; "
                               "Mehrotra interior point, tol 1e-8, as fall-back), ") if args.active_set else
                              "Mehrotra predictor-corrector interior point on stage-wise Riccati sweeps, tol 1e-8, ") +
                             ("active-horizon sweeps" if args.active_horizon else "full-horizon sweeps") +
                             (f", partial condensing N2 = {args.cond_n2}" if args.cond_n2 else "")},
            "roofline": {"bound": "hbm", "kernel": "one RTI step = k_linearise + k_factor + k_forward + k_compact + k_scatter + "
                                                     "k_as + k_ipm_rest (HIP events on the launch stream around the two "
                                                     "phases, averaged over the K timed steps themselves)",
                         "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": ach / HBM_PEAK, "traffic": traffic, "traffic_source": tsrc,
                         "alg_bytes_per_launch": alg_bytes_step(N) * B_launch, "kernel_ms": ms_step,
                         "kernel_ms_source": ("HIP events on the launch stream over the K timed steps" if world == 1 and not args.no_profile else
                                              "10 steps with HIP events AFTER the K timed steps (which carry none)"),
                         "kernel_ms_within_step": kernel_time_consistent,
                         "linearise_ms": r["ms_lin"], "qp_ms": r["ms_qp"],
                         # the same K timed steps per kernel group (seven HIP events per step on the launch stream)
                         "kernels_ms": dict(zip(("k_linearise", "k_factor", "k_forward", "k_compact+k_scatter",
                                                 "k_as (active-set solves, roll-out)", "k_ipm_rest (interior-point fall-back)"), r["kms"])),
                         "traffic_over_alg": (traffic / (alg_bytes_step(N) * B_launch)) if traffic else None,
                         # active-set group (k_as + k_ipm_rest, or solves + commit + retry + fall-back) and the whole step's kernels
                         # per event-timed step: median, 99th percentile, slowest
                         "active_set_group_ms_per_step": r["as_pct"], "step_kernels_ms_per_step": r["step_pct"],
                         # the survey's LOWER bound for reference (section 8d): an engine that never materialised the stage
                         # blocks would still move x0, yref, the iterate in and out and the status = 8 (51 N + 54) bytes
                         "compulsory_io": {"bytes_per_instance": 8 * (51 * N + 54),
                                           "achieved": 8 * (51 * N + 54) * B_launch / (ms_step * 1e-3) / 1e9, "unit": "GB/s",
                                           "frac": 8 * (51 * N + 54) * B_launch / (ms_step * 1e-3) / HBM_PEAK,
                                           "traffic_over_compulsory": (traffic / (8 * (51 * N + 54) * B_launch)) if traffic else None},
                         "qp_phase": {"kernels": "k_factor + k_forward + k_compact + k_scatter + k_as + k_ipm_rest",
                                      "alg_bytes_per_launch": alg_bytes_qp(N) * B_launch, "achieved": ach_qp / 1e9,
                                      "frac": ach_qp / HBM_PEAK, "traffic": traffic_qp},
                         # the same model against the wall clock of the whole closed loop (plant, I/O kernels included)
                         "step_frac_hbm": alg_bytes_step(N) * r["value"] / (world * HBM_PEAK),
                         # secondary ceiling: FP64 vector rate against the survey's flop model --
                         # linearise N*4*(150 + 2*60*17), one Riccati solve = N * 10 kflop, (1 + mean extra solves) of them
                         "fp64_vector": (lambda fl: {"alg_flops_per_step": fl, "achieved_tflops": fl * r["value"] / world / 1e12,
                                                     "peak_tflops": FP64_VECTOR_PEAK / 1e12,
                                                     "frac": fl * r["value"] / world / FP64_VECTOR_PEAK})(
                             N * 4 * (150 + 2 * 60 * 17) + (1.0 + r["mean_qp_solves"]) * N * 10e3)},
            "qp_stats": {"status_ok_frac": r["ok_frac"], "mean_qp_solves": r["mean_qp_solves"],
                         "frac_constrained": r["frac_constrained"], "mean_head_stages": r["mean_head"]},
        }
        if dist is not None:   # how the aggregate report travelled (torch.distributed under torchrun)
            out["report_collective"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "device": red_dev.type}
        if weak is not None:
            out["weak_scaling"] = {"value": weak["value"], "unit": "RTI steps/s", "batch_per_gpu": weak["batch_rank"],
                                   "total_batch": int(weak["total"]), "ms_per_step": weak["ms_per_step"],
                                   "kernel_ms": weak["ms_lin"] + weak["ms_qp"], "steps": args.steps, "warmup": args.warmup}
        if extras:
            out["sensitivity"] = extras
        if traffic is None:
            out["roofline"]["traffic_source"] = ("n/a: the PMC passes (tools/profile_round.sh) were taken at 65536 instances per launch, this run's launches cover "
                                                 f"{B_launch}")
        if world > 1:
            out["cpu_baseline"] = {"omitted": "timed on rank 0 at N = 1 only (one bounded sample of host-core work per report, bench.py --gpus 1)"}
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
