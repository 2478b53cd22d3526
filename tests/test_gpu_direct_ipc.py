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


def test_two_processes_store_their_decisions_into_each_other_and_replay():
    """device-side replay exchange between processes (tests/workers/replay_ipc_worker.py): emx_direct_export / _import of the
    receive buffers, remote stores, the barrier, the replay -- every rank's full replica equals the single-rank chain"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29673", os.path.join(ROOT, "tests", "workers", "replay_ipc_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert out.count("OK") >= 2 and "MISMATCH" not in out, out[-4000:]


def test_the_sampler_api_under_torchrun_with_the_device_side_replay_exchange():
    """EnsembleSampler(..., distributed=True, exchange="replay_push") in two processes (tests/workers/sampler_replay_worker.py):
    the drop-in API over the sharded emx_run, no collective library involved; every rank ends with the single-process chain, for
    a closed-form device target and for a torch callable"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29677", os.path.join(ROOT, "tests", "workers", "sampler_replay_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert out.count("OK") >= 4 and "MISMATCH" not in out, out[-4000:]
