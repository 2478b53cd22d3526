"""bench.py --gpus N started without a launcher: it starts its own N rank processes and passes rank 0's line through."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tools.benchkit.out import _claim_stdout, log

BENCH_PY = os.path.join(ROOT, "bench.py")



# ------------------------------------------------------------------------------------------------ self-launch (N > 1)
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def error_line(args, msg, extra=None):
    line = {"metric": "walker-updates/sec (whole node), 64-dim correlated Gaussian, StretchMove a=2", "value": None,
            "unit": "walker-updates/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
            "higher_is_better": True, "dtype": "f64", "data": "synthetic", "error": msg}
    if extra:
        line.update(extra)
    return line


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks HERE, one process per GPU, with the
    environment torch.distributed.run would have given them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), and pass
    rank 0's single JSON line through.  This process never touches a GPU.  Under torchrun (WORLD_SIZE already set) main() takes
    the rank path directly, so both ways of starting an N-GPU run execute the same code."""
    import subprocess
    out = _claim_stdout()
    N = args.gpus
    ndev = None
    if args.all_on_device is None and not os.environ.get("EMX_BENCH_STUB"):
        try:
            from emcee_amd import _lib
            ndev = _lib.device_count()
        except Exception as e:  # noqa: BLE001
            log("device count unavailable:", e)
        if ndev is not None and ndev < N:
            emit_line(error_line(args, "--gpus %d but only %d HIP device(s) are visible" % (N, ndev), {"devices_visible": ndev}))
            return 2
    port = _free_port()
    env0 = dict(os.environ)
    env0.update({"WORLD_SIZE": str(N), "LOCAL_WORLD_SIZE": str(N), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                 "EMX_BENCH_SELF_LAUNCHED": "1"})
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL and hipIpc between processes need it
    procs = []
    for r in range(N):
        env = dict(env0)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r)})
        cmd = [sys.executable, BENCH_PY] + list(argv)
        # rank 0's stdout carries the line; the other ranks' goes to stderr (they print nothing there by contract)
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=None, text=(r == 0),
                                      start_new_session=True))
    log("self-launch: %d ranks (pids %s), rendezvous 127.0.0.1:%d" % (N, [p.pid for p in procs], port))
    deadline = time.time() + args.launch_timeout
    line0 = None
    try:
        try:
            line0, _ = procs[0].communicate(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            line0 = None
        for p in procs[1:]:
            try:
                p.wait(timeout=max(1.0, min(60.0, deadline - time.time())))
            except subprocess.TimeoutExpired:
                pass
    finally:
        for p in procs:                       # exactly the process groups started above
            if p.poll() is None:
                try:
                    os.killpg(p.pid, 9)
                except Exception:  # noqa: BLE001
                    pass
    rcs = [p.returncode for p in procs]
    text = [ln for ln in (line0 or "").splitlines() if ln.strip().startswith("{")]
    if not text:
        emit_line(error_line(args, "the ranks produced no result line (exit codes %s%s)"
                             % (rcs, "; timed out after %.0f s" % args.launch_timeout if line0 is None else "")))
        return 1
    try:
        line = json.loads(text[-1])
        line["launcher"] = "bench.py self-launch: %d rank processes, one per GPU (no torchrun around it)" % N
        out.write(json.dumps(line) + "\n")
    except Exception:  # noqa: BLE001
        out.write(text[-1] + "\n")
    out.flush()
    return 0 if all(rc == 0 for rc in rcs) else 1


# ------------------------------------------------------------------------------------------------ main
def emit_line(line):
    """every line goes out through tools/benchkit/emit.py: the compact summary on stdout, the full record beside it"""
    from tools.benchkit.emit import emit_record
    emit_record(line)
