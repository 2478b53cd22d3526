"""exchange='logprob' with a Python log_prob_fn between PROCESSES (tests/workers/logprob_host_worker.py): every rank runs the
whole sampler on its own device context and calls the user's function on its share of the proposals only; the chains,
log-probs, blobs and acceptance fractions equal the single-process sampler's.  One GPU on the box: both ranks use cuda:0."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_python_log_prob_calls_are_shared_out_over_the_ranks():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29683", os.path.join(ROOT, "tests", "workers", "logprob_host_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert out.count("OK") >= 2 and "MISMATCH" not in out, out[-4000:]
