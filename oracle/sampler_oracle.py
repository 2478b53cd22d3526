"""NumPy oracle of the red/blue split-ensemble update (TEST INFRASTRUCTURE).

Every function cites the reference lines it restates (paths relative to
``/root/reference/src/emcee``).  Random draws are taken from a NumPy legacy
``RandomState`` in exactly the order the reference takes them, so a run of
this oracle with the same MT19937 state reproduces reference emcee bit for
bit (pinned by ``tests/golden`` -- see ``oracle/__init__.py``).

Nothing here is imported by the product (``emcee_amd``).
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "iso_gauss", "diag_gauss", "dense_gauss", "rosenbrock",
    "MoveSpec", "propose", "run", "de_pair_decode", "de_pair_table",
    "integrated_time",
]


# --------------------------------------------------------------------------
# Targets (the "batched log-prob" half of the path, ensemble.py:458-553 with
# vectorize=True; formulas from SURVEY.md 8d).
# --------------------------------------------------------------------------
def iso_gauss(x):
    """-0.5*sum(x^2): tests/integration/test_proposal.py:21-22, vectorised."""
    x = np.atleast_2d(x)
    return -0.5 * np.sum(x ** 2, axis=1)


def diag_gauss(x, mu, ivar):
    """-0.5*sum(ivar*(x-mu)^2): docs/index.rst:41-45."""
    d = np.atleast_2d(x) - mu
    return -0.5 * np.sum(ivar * d * d, axis=1)


def dense_gauss(x, mu, icov):
    """-0.5*(x-mu)^T icov (x-mu): docs/tutorials/quickstart.ipynb:76."""
    d = np.atleast_2d(x) - mu
    return -0.5 * np.einsum("ij,ij->i", d @ icov, d)


def rosenbrock(x):
    """Chained Rosenbrock / 20 (not in the reference)."""
    x = np.atleast_2d(x)
    a = x[:, 1:] - x[:, :-1] ** 2
    b = 1.0 - x[:, :-1]
    return -np.sum(100.0 * a * a + b * b, axis=1) / 20.0


# --------------------------------------------------------------------------
# Moves
# --------------------------------------------------------------------------
class MoveSpec:
    """Plain description of one of the three hot-path moves.

    kind: 'stretch' (moves/stretch.py:22), 'de' (moves/de.py:28) or
    'snooker' (moves/de_snooker.py:26-29, nsplits forced to 4).
    """

    def __init__(self, kind="stretch", a=2.0, sigma=1.0e-5, gamma0=None,
                 gammas=1.7, nsplits=2, randomize_split=True,
                 live_dangerously=False, cov=None, mode="vector", factor=None,
                 s=None, bw_method=None):
        self.kind = kind
        # 'gaussian' (moves/gaussian.py + moves/mh.py), 'walk' (moves/walk.py), 'kde' (moves/kde.py):
        # the SURVEY.md 8(f) widening; host-side proposals in the reference and in the product
        self.cov = cov
        self.mode = mode
        self.factor = factor
        self.index = 0          # gaussian.py:66,97: the sequential mode's cursor lives in the move
        self.s = s
        self.bw_method = bw_method
        self.a = a
        self.sigma = sigma
        self.gamma0 = gamma0
        self.gammas = gammas
        self.nsplits = 4 if kind == "snooker" else int(nsplits)
        self.randomize_split = randomize_split
        self.live_dangerously = live_dangerously


def de_pair_table(n):
    """moves/de.py:67-77 verbatim semantics (small n only: n(n-1) rows)."""
    rows, cols = np.tril_indices(n, -1)
    return np.column_stack([np.concatenate([rows, cols]),
                            np.concatenate([cols, rows])])


def de_pair_decode(k, n):
    """Row ``k`` of de_pair_table(n) in closed form.

    tril_indices(n, -1) enumerates (1,0),(2,0),(2,1),(3,0)...; the table is
    [(row, col) for k < T] + [(col, row) for k >= T], T = n(n-1)/2.
    Returns (first, second) so that diff = c[second] - c[first] (de.py:53).
    """
    k = np.asarray(k, dtype=np.int64)
    T = n * (n - 1) // 2
    kk = np.where(k < T, k, k - T)
    i = ((1 + np.sqrt(1.0 + 8.0 * kk.astype(np.float64))) // 2).astype(np.int64)
    # guard the float sqrt at triangular-number boundaries
    i = np.where(i * (i - 1) // 2 > kk, i - 1, i)
    i = np.where((i + 1) * i // 2 <= kk, i + 1, i)
    j = kk - i * (i - 1) // 2
    first = np.where(k < T, i, j)
    second = np.where(k < T, j, i)
    return first, second


def _stretch(s, c, random, a, tr):
    """moves/stretch.py:26-33."""
    c = np.concatenate(c, axis=0)
    Ns, Nc = len(s), len(c)
    ndim = s.shape[1]
    u = random.rand(Ns)
    zz = ((a - 1.0) * u + 1) ** 2.0 / a
    factors = (ndim - 1.0) * np.log(zz)
    rint = random.randint(Nc, size=(Ns,))
    if tr is not None:
        tr.update(u_z=u, zz=zz, rint=rint)
    return c[rint] - (c[rint] - s) * zz[:, None], factors


def _de(s, c, random, sigma, g0, tr):
    """moves/de.py:40-64 with the pair table replaced by its closed form."""
    c = np.concatenate(c, axis=0)
    ns, ndim = s.shape
    nc = c.shape[0]
    indices = random.choice(nc * (nc - 1), size=ns, replace=True)
    first, second = de_pair_decode(indices, nc)
    diffs = c[second] - c[first]
    g = random.randn(ns, 1)
    gamma = g0 * (1 + sigma * g)
    if tr is not None:
        tr.update(pair_index=indices, first=first, second=second, gauss=g[:, 0])
    return s + gamma * diffs, np.zeros(ns, dtype=np.float64)


def _snooker(s, c, random, gammas, tr):
    """moves/de_snooker.py:31-46 (per-walker loop: the draw order matters)."""
    Ns = len(s)
    Nc = list(map(len, c))
    ndim = s.shape[1]
    q = np.empty_like(s)
    metropolis = np.empty(Ns, dtype=np.float64)
    picks = np.empty((Ns, 3), dtype=np.int64)
    perm = np.empty((Ns, 3), dtype=np.int64)
    for i in range(Ns):
        r = [random.randint(Nc[j]) for j in range(3)]
        w = np.array([c[j][r[j]] for j in range(3)])
        order = np.arange(3)
        # random.shuffle(w) on a (3, D) array draws random_interval(2) then
        # random_interval(1); shuffling a companion index array with a copy of
        # the generator state would double-consume, so replay the swaps:
        st = random.get_state()
        random.shuffle(w)
        st2 = random.get_state()
        random.set_state(st)
        random.shuffle(order)
        assert _same_state(random.get_state(), st2)
        picks[i] = r
        perm[i] = order
        z, z1, z2 = w
        delta = s[i] - z
        norm = np.linalg.norm(delta)
        u = delta / norm
        q[i] = s[i] + u * gammas * (np.dot(u, z1) - np.dot(u, z2))
        metropolis[i] = np.log(np.linalg.norm(q[i] - z)) - np.log(norm)
    if tr is not None:
        tr.update(picks=picks, perm=perm)
    return q, (ndim - 1.0) * metropolis


def _same_state(a, b):
    return (a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2] == b[2]
            and a[3] == b[3] and a[4] == b[4])


def _gaussian_proposal(x0, random, move):
    """moves/gaussian.py:78-103 (+ :110-117 for a matrix ``cov``)."""
    nw, nd = x0.shape
    try:
        float(move.cov)
        scale, full = np.sqrt(move.cov), False                     # gaussian.py:54-56
    except TypeError:
        cov = np.atleast_1d(move.cov)
        full = cov.ndim == 2
        scale = cov if full else np.sqrt(cov)                        # gaussian.py:42,47
    f = 1.0
    if move.factor is not None:                                      # gaussian.py:81-84, drawn first
        lf = np.log(move.factor)
        f = np.exp(random.uniform(-lf, lf))
    if full:
        xnew = x0 + f * random.multivariate_normal(np.zeros(len(scale)), scale)
    else:
        xnew = x0 + f * scale * random.randn(*(x0.shape))            # gaussian.py:87
    if move.mode == "vector":
        return xnew, np.zeros(nw)
    if move.mode == "random":
        cols = random.randint(nd, size=nw)                           # gaussian.py:93
    else:
        cols = move.index % nd + np.zeros(nw, dtype=int)             # gaussian.py:95-96
        move.index = (move.index + 1) % nd
    x = np.array(x0)
    x[np.arange(nw), cols] = xnew[np.arange(nw), cols]
    return x, np.zeros(nw)


def _propose_mh(coords, log_prob, lp_fn, random, move):
    """moves/mh.py:35-65: all walkers at once; ``log(rand) < lnpdiff`` (strict, uniforms drawn last)."""
    q, factors = _gaussian_proposal(coords, random, move)
    new_lp = np.asarray(lp_fn(q), dtype=np.float64)
    if np.any(np.isnan(new_lp)):
        raise ValueError("Probability function returned NaN")
    lnpdiff = new_lp - log_prob + factors
    with np.errstate(divide="ignore"):
        accepted = np.log(random.rand(len(coords))) < lnpdiff
    coords[accepted] = q[accepted]
    log_prob[accepted] = new_lp[accepted]
    return accepted


def _walk(s, c, random, nhelp):
    """moves/walk.py:28-42."""
    c = np.concatenate(c, axis=0)
    Ns, Nc = len(s), len(c)
    q = np.empty_like(s)
    s0 = Nc if nhelp is None else nhelp
    for i in range(Ns):
        inds = random.choice(Nc, s0, replace=False)
        cov = np.atleast_2d(np.cov(c[inds], rowvar=0))
        q[i] = random.multivariate_normal(s[i], cov)
    return q, np.zeros(Ns, dtype=np.float64)


def _kde(s, c, random, bw_method):
    """moves/kde.py:40-45 (scipy.stats.gaussian_kde is the reference's own dependency)."""
    from scipy.stats import gaussian_kde
    c = np.concatenate(c, axis=0)
    kde = gaussian_kde(c.T, bw_method=bw_method)
    q = kde.resample(len(s), random)
    factor = kde.logpdf(s.T) - kde.logpdf(q)
    return q.T, factor


def propose(coords, log_prob, lp_fn, random, move, trace=None):
    """moves/red_blue.py:52-106 + moves/move.py:29-45, in place.

    ``lp_fn`` is a vectorised log-prob ((n, D) -> (n,)).  Returns the
    accepted mask (bool[N]).  ``trace`` (a list) receives one dict per split.
    """
    nwalkers, ndim = coords.shape
    if move.kind == "gaussian":
        return _propose_mh(coords, log_prob, lp_fn, random, move)
    if nwalkers < 2 * ndim and not move.live_dangerously:
        raise RuntimeError("It is unadvisable to use a red-blue move with fewer "
                           "walkers than twice the number of dimensions.")
    g0 = None
    if move.kind == "de":                       # de.py:33-38
        g0 = move.gamma0
        if g0 is None:
            g0 = 2.38 / np.sqrt(2 * ndim)
    accepted = np.zeros(nwalkers, dtype=bool)
    all_inds = np.arange(nwalkers)
    inds = all_inds % move.nsplits              # red_blue.py:78
    if move.randomize_split:
        random.shuffle(inds)                    # red_blue.py:80
    for split in range(move.nsplits):
        S1 = inds == split
        sets = [coords[inds == j] for j in range(move.nsplits)]
        s = sets[split]
        c = sets[:split] + sets[split + 1:]
        tr = None if trace is None else {"split": split, "labels": inds.copy()}
        if move.kind == "stretch":
            q, factors = _stretch(s, c, random, move.a, tr)
        elif move.kind == "de":
            q, factors = _de(s, c, random, move.sigma, g0, tr)
        elif move.kind == "snooker":
            q, factors = _snooker(s, c, random, move.gammas, tr)
        elif move.kind == "walk":
            q, factors = _walk(s, c, random, move.s)
        elif move.kind == "kde":
            q, factors = _kde(s, c, random, move.bw_method)
        else:
            raise ValueError(move.kind)
        if np.any(np.isinf(q)):                 # ensemble.py:476-479
            raise ValueError("At least one parameter value was infinite")
        if np.any(np.isnan(q)):
            raise ValueError("At least one parameter value was NaN")
        new_lp = np.asarray(lp_fn(q), dtype=np.float64)
        if np.any(np.isnan(new_lp)):            # ensemble.py:550-551
            raise ValueError("Probability function returned NaN")
        # red_blue.py:96-101 -- one scalar rand() per walker, ascending order,
        # is the same MT stream as one rand(Ns).
        u_acc = random.rand(len(s))
        with np.errstate(divide="ignore", invalid="ignore"):
            lnpdiff = factors + new_lp - log_prob[S1]
            acc = lnpdiff > np.log(u_acc)
        accepted[S1] = acc
        # move.py:31-34
        m1 = S1 & accepted
        coords[m1] = q[acc]
        log_prob[m1] = new_lp[acc]
        if tr is not None:
            tr.update(q=q, factors=factors, new_lp=new_lp, u_acc=u_acc, acc=acc)
            trace.append(tr)
    return accepted


def run(p0, nsteps, lp_fn, random, moves=None, weights=None, thin_by=1,
        store=True, trace=None, log_prob0=None):
    """ensemble.py:258-424 (sample loop) for the in-scope feature set.

    Returns dict(chain, log_prob, accepted_count, coords, lp, move_choices).
    ``nsteps`` counts stored steps; ``nsteps*thin_by`` proposals are made.
    """
    if moves is None:
        moves = [MoveSpec("stretch")]
    if weights is None:
        weights = np.ones(len(moves))
    weights = np.atleast_1d(weights).astype(float)
    weights /= np.sum(weights)                  # ensemble.py:128-129
    coords = np.array(p0, dtype=np.float64, copy=True)
    N, D = coords.shape
    lp = (np.asarray(lp_fn(coords), dtype=np.float64) if log_prob0 is None
          else np.array(log_prob0, dtype=np.float64, copy=True))
    if np.any(np.isnan(lp)):
        raise ValueError("The initial log_prob was NaN")
    chain = np.empty((nsteps if store else 0, N, D))
    lps = np.empty((nsteps if store else 0, N))
    acc_count = np.zeros(N)
    choices = []
    i = 0
    for it in range(nsteps):
        for _ in range(thin_by):
            k = int(random.choice(len(moves), p=weights))   # ensemble.py:406
            choices.append(k)
            step_trace = None if trace is None else []
            acc = propose(coords, lp, lp_fn, random, moves[k], step_trace)
            if trace is not None:
                trace.append(step_trace)
            if store and (i + 1) % thin_by == 0:            # ensemble.py:416
                chain[it] = coords
                lps[it] = lp
                acc_count += acc                            # backend.py:229
            i += 1
    return dict(chain=chain, log_prob=lps, accepted_count=acc_count,
                coords=coords, lp=lp, move_choices=np.array(choices))


# --------------------------------------------------------------------------
# Integrated autocorrelation time (autocorr.py:20-123), used for the tau leg
# of the metric.
# --------------------------------------------------------------------------
def _next_pow_two(n):
    i = 1
    while i < n:
        i = i << 1
    return i


def _function_1d(x):
    """autocorr.py:20-39."""
    n = _next_pow_two(len(x))
    f = np.fft.fft(x - np.mean(x), n=2 * n)
    acf = np.fft.ifft(f * np.conjugate(f))[: len(x)].real
    acf /= acf[0]
    return acf


def integrated_time(x, c=5):
    """autocorr.py:49-106 for x of shape (n_step, n_walker, n_dim); no tol check."""
    n_t, n_w, n_d = x.shape
    tau = np.empty(n_d)
    for d in range(n_d):
        f = np.zeros(n_t)
        for k in range(n_w):
            f += _function_1d(x[:, k, d])
        f /= n_w
        taus = 2.0 * np.cumsum(f) - 1.0
        m = np.arange(len(taus)) < c * taus
        win = np.argmin(m) if np.any(m) else len(taus) - 1
        tau[d] = taus[win]
    return tau


# --------------------------------------------------------------------------
# "Inputs mode": replay a fully resolved step plan (the arrays the device
# consumes: plan order, partner walkers, zz/gamma, accept uniforms) with the
# reference's arithmetic.  Used to check the native (Philox) mode, whose draws
# come from the device, against NumPy math, and for teacher-forced parity.
# --------------------------------------------------------------------------
def propose_planned(coords, log_prob, lp_fn, plan, move):
    """In-place red/blue step from a resolved plan; returns the accepted mask.

    plan: dict(off, order, p0, p1, p2, s0, uacc) in plan order (split 0's
    members, then split 1's, ...).  Arithmetic per stretch.py:33, de.py:53-62,
    de_snooker.py:41-46, red_blue.py:96-104.
    """
    N, ndim = coords.shape
    off, order = plan["off"], plan["order"]
    accepted = np.zeros(N, dtype=bool)
    for split in range(len(off) - 1):
        sl = slice(off[split], off[split + 1])
        idx = order[sl]
        s = coords[idx]
        if move.kind == "stretch":
            zz = plan["s0"][sl]
            cj = coords[plan["p0"][sl]]
            q = cj - (cj - s) * zz[:, None]
            factors = (ndim - 1.0) * np.log(zz)
        elif move.kind == "de":
            diffs = coords[plan["p1"][sl]] - coords[plan["p0"][sl]]
            q = s + plan["s0"][sl][:, None] * diffs
            factors = np.zeros(len(idx))
        else:
            z, z1, z2 = coords[plan["p0"][sl]], coords[plan["p1"][sl]], coords[plan["p2"][sl]]
            q = np.empty_like(s)
            met = np.empty(len(idx))
            for i in range(len(idx)):
                delta = s[i] - z[i]
                norm = np.linalg.norm(delta)
                u = delta / norm
                q[i] = s[i] + u * move.gammas * (np.dot(u, z1[i]) - np.dot(u, z2[i]))
                met[i] = np.log(np.linalg.norm(q[i] - z[i])) - np.log(norm)
            factors = (ndim - 1.0) * met
        new_lp = np.asarray(lp_fn(q), dtype=np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            acc = factors + new_lp - log_prob[idx] > np.log(plan["uacc"][sl])
        accepted[idx] = acc
        coords[idx[acc]] = q[acc]
        log_prob[idx[acc]] = new_lp[acc]
    return accepted
