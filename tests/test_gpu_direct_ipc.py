"""Direct exchange between PROCESSES: hipIpc-mapped coordinate arrays and barrier flags (tests/workers/direct_ipc_worker.py).
The box has one GPU, so both processes use cuda:0; what is exercised is exactly what differs from the in-process
logical-rank tests of test_gpu_sharded.py: emx_direct_export / emx_direct_import and the device-side barrier across
process boundaries."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_processes_map_each_other_and_reproduce_the_single_rank_chain():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29671", os.path.join(ROOT, "tests", "workers", "direct_ipc_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert out.count("OK") >= 2 and "MISMATCH" not in out, out[-4000:]
