"""Randomised check of the exact-mode plan pipeline against the serial host twin (no GPU): usage pipeline_fuzz.py [seed] [cases]"""
import sys, zlib, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import test_mt_pipeline_cpu as T
rs0 = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    N = int(rs0.choice([rs0.randint(8, 300), rs0.randint(300, 5000), rs0.randint(5000, 40000), 2 ** rs0.randint(5, 16)]))
    kind = rs0.choice(["de", "snooker", "stretch", "mix"])
    S = int(rs0.choice([2, 3, 4, 5]))
    if kind == "snooker" and S < 4: S = 4
    if N < 4 * S + 8: N = 4 * S + 8
    moves = {"de": [T.md("de", S=S)], "snooker": [T.md("snooker", S=S)], "stretch": [T.md("stretch", S=S)],
             "mix": [T.md("stretch", S=2), T.md("de", S=S), T.md("snooker", S=max(4, S))]}[kind]
    w = np.ones(len(moves)); cdf = np.cumsum(w / w.sum()); cdf /= cdf[-1]
    rs = np.random.RandomState(rs0.randint(1 << 30))
    if rs0.rand() < 0.5: rs.randn(1)
    st = list(rs.get_state()); st[2] = int(rs0.randint(0, 625)); st = tuple(st)
    nsteps = int(rs0.randint(3, 9))
    workers, nsinks = int(rs0.choice([1, 2, 3, 6])), int(rs0.choice([2, 4, 16]))
    want, ws = T.serial(st, N, 4, moves, cdf, nsteps)
    got, gs, _ = T.stream(st, N, 4, moves, cdf, nsteps, workers, nsinks)
    ok = T.same_state(ws, gs)
    for n, ((ka, pa), (kb, pb)) in enumerate(zip(want, got)):
        ok = ok and ka == kb
        k = moves[ka].kind
        for key in ["order", "p0", "uacc", "s0"] + (["p1", "p2"] if k != 0 else []):
            ok = ok and np.array_equal(pa[key], pb[key])
    print(it, kind, "N", N, "S", S, "steps", nsteps, "workers", workers, "sinks", nsinks, "OK" if ok else "MISMATCH", flush=True)
    bad += not ok
print("bad", bad)
