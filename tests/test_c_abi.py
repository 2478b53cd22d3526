"""A C program (tests/c/abi_smoke.c) drives libemx.so through include/emx.h alone."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "abi_smoke.c")
LIB = os.path.join(ROOT, "emcee_amd", "libemx.so")


def build(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    subprocess.run([cc, "-std=gnu11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe, "-ldl", "-lm"],
                   check=True)
    return exe


def test_header_compiles_as_c_and_host_entry_points_work(tmp_path):
    exe = build(tmp_path)
    r = subprocess.run([exe, LIB], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host entry points ok" in r.stdout


@pytest.mark.gpu
def test_c_program_runs_the_sampler_on_the_gpu(tmp_path):
    exe = build(tmp_path)
    env = dict(os.environ)
    # share the process-wide HIP runtime choice with the Python tests: none is preloaded here
    r = subprocess.run([exe, LIB, "gpu"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gpu run ok" in r.stdout
