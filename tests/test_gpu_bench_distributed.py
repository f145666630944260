"""What the driver's SCALE run launches: `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N ...` -- driven here with two ranks sharing the one GPU of the test box (gloo for the report's
two all-reduces, the data path has no collective), asserting on the JSON line."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
           "--batch", "8192", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_two_ranks_strong_scaling_line():
    d = _run([])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 2
    assert d["scaling"] == "strong" and d["unit"] == "RTI steps/s" and d["dtype"] == "f64"
    assert d["config"]["total_batch"] == 8192 and d["config"]["batch_per_gpu"] == 4096
    assert d["weak_scaling"]["total_batch"] == 16384 and d["weak_scaling"]["batch_per_gpu"] == 8192
    assert d["qp_stats"]["status_ok_frac"] == 1.0
    assert d["value"] > 0 and abs(d["value"] - 8192 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    # N > 1 lines SAY what they leave out instead of dropping the keys
    assert set(d["cpu_baseline"]) == {"omitted"} and d["roofline"]["bound"] == "hbm"
    assert d["roofline"]["traffic"] is None and d["roofline"]["traffic_source"].startswith("n/a")
    assert d["roofline"]["active_set_group_ms_per_step"]["max"] >= d["roofline"]["active_set_group_ms_per_step"]["p50"] > 0


def test_bench_two_ranks_weak_scaling_line():
    d = _run(["--scaling", "weak"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["total_batch"] == 16384 and d["config"]["batch_per_gpu"] == 8192
    assert "weak_scaling" not in d
    assert d["qp_stats"]["status_ok_frac"] == 1.0


def test_bench_one_rank_rccl_smoke():
    """One rank under torchrun with the nccl backend (= RCCL on ROCm): the process group is initialised with the device id
    and the report's MAX / SUM all-reduces run on device tensors -- so that the driver's 8-GPU launch is not RCCL's first
    contact with this code (one all-reduce per report, no data-path collective)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dist-backend", "nccl",
           "--batch", "4096", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-extras"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["total_batch"] == 4096 and d["qp_stats"]["status_ok_frac"] == 1.0
    assert d["report_collective"] == {"backend": "nccl", "world_size": 1, "device": "cuda"}
    assert d["value"] > 0


def _plain(extra, expect_ok=True):
    """`python bench.py --gpus 2 ...` with NO launcher in the command and no RANK / WORLD_SIZE in the environment"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    if not expect_ok:
        return r
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_plain_python_bench_gpus_2_means_two_ranks():
    """BASELINE.json's metric is quoted "at 1/2/4/8 MI355X": `python bench.py --gpus N` started WITHOUT torch.distributed.run
    re-launches itself under it (one rank per GPU) -- it never benches one GPU and prints n_gpus 1."""
    d = _plain(["--dist-backend", "gloo", "--batch", "8192"])
    assert d["n_gpus"] == 2 and d["report_collective"]["world_size"] == 2 and d["report_collective"]["backend"] == "gloo"
    assert d["config"]["total_batch"] == 8192 and d["config"]["batch_per_gpu"] == 4096
    assert d["qp_stats"]["status_ok_frac"] == 1.0 and d["value"] > 0


def test_plain_python_bench_gpus_2_with_rccl_fails_loudly_on_one_gpu():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices visible: the nccl launch is legitimate here")
    r = _plain(["--dist-backend", "nccl", "--batch", "8192"], expect_ok=False)
    assert r.returncode != 0
    assert "needs 2 visible HIP devices" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    # ... and a launcher that starts fewer ranks than --gpus asks for is refused as well (no line with n_gpus != --gpus)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
           "--batch", "4096", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_config_c5_mixed_horizons_over_two_ranks():
    """Config C5 ("mixed horizons N in {30, 50, 100} ... 8 x MI355X") as a multi-rank workload: the fleet-wide horizon draw is
    dealt out by parallel.shard_by_horizon, every rank runs its vehicles as one cfnmpc_fleet; per-rank sum N within 1 %."""
    d = _plain(["--dist-backend", "gloo", "--batch", "6000", "--workload", "mixed"])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["workload"].startswith("C5 mixed horizons") and c["total_batch"] == 6000
    assert sum(c["horizons"].values()) == 6000 and all(v > 1500 for v in c["horizons"].values())
    pr = c["per_rank"]
    assert len(pr) == 2 and sum(p["vehicles"] for p in pr) == 6000
    assert c["sum_N_imbalance"] < 0.01 and abs(pr[0]["sum_N"] - pr[1]["sum_N"]) <= 100
    for n in ("30", "50", "100"):
        assert abs(pr[0]["buckets"][n] - pr[1]["buckets"][n]) <= 2 and pr[0]["buckets"][n] + pr[1]["buckets"][n] == c["horizons"][n]
    assert d["qp_stats"]["status_ok_frac"] == 1.0 and d["qp_stats"]["frac_constrained"] > 0
    assert abs(d["value"] - 6000 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"] and d["stage_steps_per_s"] > d["value"] * 30


def test_bench_config_c4_figure8_over_two_ranks():
    d = _plain(["--dist-backend", "gloo", "--batch", "8192", "--workload", "figure8", "--no-extras"])
    assert d["n_gpus"] == 2 and d["config"]["workload"].startswith("C4 figure-8 tracking")
    assert d["config"]["total_batch"] == 8192 and d["config"]["batch_per_gpu"] == 4096
    assert d["qp_stats"]["status_ok_frac"] == 1.0 and d["qp_stats"]["frac_constrained"] > 0
