"""Generate tests/golden/*.npz by running REFERENCE emcee (build container only).

TEST INFRASTRUCTURE.  Usage:  python -m oracle.gen_golden [case ...]

For every case in oracle/cases.py the live reference (imported from
/root/reference/src through oracle/ref_shim.py) is run from a recorded
MT19937 state; the fixture stores the inputs (p0, initial RNG state) and the
reference outputs (chain, log_prob, per-walker accepted counts, final RNG
state, and -- through a recording proxy around ``sampler._random`` -- the
split labels after each shuffle and every randint/choice draw).  Larger
DIGEST cases store only SHA-256 digests of the outputs.
"""
import hashlib
import json
import os
import sys

import numpy as np

from . import cases, ref_shim

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class RecordingRandom:
    """Proxy around a RandomState that logs what the reference draws."""

    def __init__(self, rs):
        self._rs = rs
        self.labels = []
        self.ints = []

    def shuffle(self, x):
        self._rs.shuffle(x)
        if isinstance(x, np.ndarray) and x.ndim == 1 and x.dtype.kind == "i":
            self.labels.append(x.copy())

    def randint(self, *a, **k):
        r = self._rs.randint(*a, **k)
        self.ints.append(np.atleast_1d(np.asarray(r, dtype=np.int64)).copy())
        return r

    def choice(self, a, *args, **k):
        r = self._rs.choice(a, *args, **k)
        if isinstance(a, (int, np.integer)):
            self.ints.append(np.atleast_1d(np.asarray(r, dtype=np.int64)).copy())
        return r

    def __getattr__(self, name):
        return getattr(self._rs, name)


def _ref_moves(emcee, specs, weights):
    out = []
    for m in specs:
        kw = dict(nsplits=m.nsplits, randomize_split=m.randomize_split, live_dangerously=m.live_dangerously)
        if m.kind == "stretch":
            out.append(emcee.moves.StretchMove(a=m.a, **kw))
        elif m.kind == "de":
            out.append(emcee.moves.DEMove(sigma=m.sigma, gamma0=m.gamma0, **kw))
        elif m.kind == "snooker":
            kw.pop("nsplits")
            out.append(emcee.moves.DESnookerMove(gammas=m.gammas, **kw))
        elif m.kind == "gaussian":
            out.append(emcee.moves.GaussianMove(m.cov, mode=m.mode, factor=m.factor))
        elif m.kind == "walk":
            out.append(emcee.moves.WalkMove(s=m.s, **kw))
        elif m.kind == "kde":
            out.append(emcee.moves.KDEMove(bw_method=m.bw_method, **kw))
    if weights is not None:
        return list(zip(out, weights))
    return out


def run_reference(name):
    emcee = ref_shim.import_reference()
    spec = cases.build(name)
    fn = cases.make_target(spec["desc"])
    if spec["per_walker"]:
        lp = lambda p: float(fn(p[None, :])[0])  # noqa: E731
        vec = False
    else:
        lp, vec = fn, True
    N, D = spec["N"], spec["D"]
    sampler = emcee.EnsembleSampler(N, D, lp, moves=_ref_moves(emcee, spec["moves"], spec["weights"]), vectorize=vec)
    sampler._random.seed(spec["rng_seed"])
    state0 = sampler._random.get_state()
    if any(m.kind == "kde" for m in spec["moves"]):
        rec = sampler._random          # scipy's gaussian_kde.resample insists on a real RandomState
    else:
        rec = RecordingRandom(sampler._random)
        sampler._random = rec
    sampler.run_mcmc(spec["p0"], spec["nsteps"], thin_by=spec["thin_by"], skip_initial_state_check=True)
    st1 = rec.get_state()
    out = dict(
        p0=spec["p0"], rng_key0=state0[1], rng_pos0=state0[2], rng_has_gauss0=state0[3], rng_cached0=state0[4],
        chain=sampler.get_chain(), log_prob=sampler.get_log_prob(), accepted_count=sampler.backend.accepted,
        rng_key1=st1[1], rng_pos1=st1[2], rng_has_gauss1=st1[3], rng_cached1=st1[4],
    )
    if getattr(rec, "labels", None):
        out["labels"] = np.stack(rec.labels)
    if getattr(rec, "ints", None):
        out["ints"] = np.concatenate(rec.ints)
    return out


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main(argv):
    os.makedirs(GOLDEN, exist_ok=True)
    names = argv or (list(cases.CASES) + list(cases.HOST_MOVE_CASES) + list(cases.DIGEST_CASES))
    digests = {}
    dpath = os.path.join(GOLDEN, "digests.json")
    if os.path.exists(dpath):
        digests = json.load(open(dpath))
    for name in names:
        out = run_reference(name)
        if name in cases.CASES or name in cases.HOST_MOVE_CASES:
            np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **out)
            print("wrote", name, out["chain"].shape)
        else:
            digests[name] = dict(
                chain=digest(out["chain"]), log_prob=digest(out["log_prob"]),
                accepted_count=digest(out["accepted_count"]), rng_key1=digest(out["rng_key1"]),
                rng_pos1=int(out["rng_pos1"]), accepted_total=float(out["accepted_count"].sum()),
                p0=digest(cases.build(name)["p0"]),
                labels=digest(out["labels"]) if "labels" in out else None,
                ints=digest(out["ints"]) if "ints" in out else None,
            )
            print("digest", name, digests[name]["accepted_total"])
    json.dump(digests, open(dpath, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:])
