"""plan_log (csrc/emx_planlog.hpp) -- the logarithm every plan entry takes on the device (log u of red_blue.py:100, log zz of
stretch.py:33) -- against the 80-bit logl, and the rewritten integer helpers of the native plans against their defining forms.
The same source compiles for the device; the GPU side is covered by the plan parity tests (tests/test_gpu_full_size.py,
tests/test_gpu_parity.py), which replay the device's plans through the oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "emcee_amd", "csrc")


def _build(tmp_path, name):
    exe = str(tmp_path / name)
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-I", CSRC,
                    os.path.join(ROOT, "tools", "ubench", name + ".cpp"), "-o", exe], check=True)
    return exe


def test_plan_log_is_within_0_55_ulp_of_logl(tmp_path):
    r = subprocess.run([_build(tmp_path, "plan_log_check"), "2000000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "log(1) = 0x0p+0" in r.stdout


def test_table_header_is_what_the_generator_writes(tmp_path):
    """emx_logtab.hpp is generated (tools/gen_logtab.py, 60-digit decimal arithmetic): the committed file is its output"""
    out = str(tmp_path / "emx_logtab.hpp")
    subprocess.run(["python", os.path.join(ROOT, "tools", "gen_logtab.py"), out], check=True, capture_output=True)
    assert open(out).read() == open(os.path.join(CSRC, "emx_logtab.hpp")).read()


def test_integer_helpers_of_the_native_plans(tmp_path):
    r = subprocess.run([_build(tmp_path, "rng_helpers_check")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bad 0" in r.stdout
