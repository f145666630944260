"""examples/: the batch C-ABI used from plain C99 (no Python, no HIP headers) -- compiles against
include/cfnmpc.h alone (CPU), runs a closed loop on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "crazyflie_nmpc_amd")


def _build(tmp_path):
    exe = str(tmp_path / "batch_hover")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "batch_hover.c"), "-L" + LIBDIR, "-lcfnmpc",
                           "-Wl,-rpath," + LIBDIR, "-lm", "-o", exe])
    return exe


def test_c_example_compiles_and_links_against_the_header_alone(tmp_path):
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_c_example_closed_loop_converges(tmp_path):
    out = subprocess.run([_build(tmp_path), "300", "300"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK converged after 300 steps" in out.stdout
