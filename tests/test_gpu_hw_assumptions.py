"""The hardware behaviour the one-XCD persistent kernels rely on, checked on the box the suite runs on ("keep the
probe runnable as a test so a new ROCm shows up as a failure, not as a wrong chain").

k_persist<..., LOCAL> / k_persist_valu (csrc/emx_kernels.hpp: persist_barrier_local, persist_handshake<true>) synchronise the
workgroups of ONE XCD through that XCD's L2: plain stores, `sc1` loads, one flag word per workgroup.  HIP's memory model promises
none of this; tools/exp/xcd_local_probe.hip measures it (profiles/r04/xcd_local_probe.txt):
  * workgroup i of a launch lands on XCD i mod 8 (every eighth workgroup of an eight times larger grid shares an XCD);
  * a plain store is in the XCD's L2 when acknowledged, and an sc1 load of a line this XCD wrote is answered by that L2 -- a flag
    barrier in that flavour completes, every round, for 8 and for 32 workgroups;
  * it is the fast flavour (well under the device-wide one's cost).
The kernels additionally check the XCC_ID themselves at every launch and give up untouched when it is not uniform."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe():
    exe = os.path.join(ROOT, "tools", "ubench", "bin", "xcd_local_probe")
    if not os.path.exists(exe):                        # (normally built by __graft_entry__.build(); hipcc is in the image)
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "tools", "exp", "xcd_local_probe.hip"),
                        "-o", exe], check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = {}
    for ln in r.stdout.splitlines():
        m = re.match(r"(.+?)\s+groups\s+(\d+): rounds completed\s+(\d+) of (\d+), ([0-9.]+) us per round, XCC mask (0x[0-9a-f]+)", ln)
        if m:
            rows[(m.group(1).strip(), int(m.group(2)))] = (int(m.group(3)), int(m.group(4)), float(m.group(5)), int(m.group(6), 16))
    assert rows, r.stdout
    return rows


def test_the_one_xcd_flag_barrier_works_as_the_local_kernels_assume():
    rows = _probe()
    for ng in (8, 32):
        done, total, us, mask = rows[("plain store, sc1 load", ng)]
        assert done == total, "plain store / sc1 load: the flag barrier of %d workgroups on one XCD stalled after %d rounds" % (ng, done)
        assert mask != 0 and (mask & (mask - 1)) == 0, "every eighth workgroup did not land on one XCD (XCC mask 0x%x)" % mask
        wide = rows[("sc1 store, sc1 load (the device-wide flavour)", ng)]
        assert wide[0] == wide[1]
        assert us < 1.0 and us <= wide[2] * 1.05, "the one-XCD flavour (%.3f us per round) is no faster than the device-wide one (%.3f)" % (us, wide[2])
