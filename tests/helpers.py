"""Shared helpers for the parity tests (may import oracle/: test infrastructure)."""
import hashlib
import json
import os

import numpy as np

from oracle import cases
from oracle import sampler_oracle as so

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def load_digests():
    return json.load(open(os.path.join(GOLDEN, "digests.json")))


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def rng_from_fixture(g, which="0"):
    rs = np.random.RandomState()
    rs.set_state(("MT19937", g["rng_key" + which], int(g["rng_pos" + which]),
                  int(g["rng_has_gauss" + which]), float(g["rng_cached" + which])))
    return rs


def rng_for_case(spec):
    rs = np.random.RandomState()
    rs.seed(spec["rng_seed"])
    return rs


def run_oracle(spec, p0, rs, trace=None):
    fn = cases.make_target(spec["desc"])
    return so.run(p0, spec["nsteps"], fn, rs, moves=spec["moves"], weights=spec["weights"],
                  thin_by=spec["thin_by"], trace=trace)
