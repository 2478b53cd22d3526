"""Stress of the exact-mode plan pipeline's life cycle: a random sequence of run_mcmc chunks, sample() generators abandoned early,
random_state reads / writes, resets and sampler teardown at several sizes, executed TWICE from the same seed (argument: samplers per pass) -- the two
passes must end in identical states (every retire point of the persistent pipeline has to leave the generator exactly
after the last step taken), and nothing may hang.   python tools/stress_exact.py [samplers]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import emcee_amd  # noqa: E402
from emcee_amd import moves, targets  # noqa: E402


def one_pass(seed, nsamplers):
    rs = np.random.RandomState(seed)
    log = []
    nops = 0
    for _ in range(nsamplers):
        N = int(rs.choice([160, 1024, 8192, 20000, 65536]))
        D = int(rs.choice([3, 16, 64]))
        mv = [moves.StretchMove(), [(moves.DEMove(), 0.7), (moves.DESnookerMove(), 0.3)]][rs.randint(2)]
        np.random.seed(int(rs.randint(1 << 30)))
        s = emcee_amd.EnsembleSampler(N, D, targets.IsoGaussian(), moves=mv, rng="mt19937")
        state = np.random.RandomState(int(rs.randint(1 << 30))).randn(N, D)
        for _ in range(int(rs.randint(2, 7))):
            op = rs.randint(6)
            nops += 1
            if op == 0:
                state = s.run_mcmc(state, int(rs.randint(1, 40)), store=False, skip_initial_state_check=True)
            elif op == 1:                               # generator abandoned early: the pipeline has produced plans ahead
                k = int(rs.randint(1, 12))
                for i, st in enumerate(s.sample(state, iterations=50, store=False, skip_initial_state_check=True)):
                    state = st
                    if i + 1 == k:
                        break
            elif op == 2:
                key = s.random_state                    # retires the pipeline, reads the generator
                s.random_state = key
            elif op == 3:
                state = s.run_mcmc(state, int(rs.randint(1, 6)), thin_by=int(rs.randint(1, 4)), skip_initial_state_check=True)
            elif op == 4:
                s.reset()
            else:
                state = s.run_mcmc(state, 1, store=False, skip_initial_state_check=True)
        fin = s.run_mcmc(state, 2, store=False, skip_initial_state_check=True)
        log.append((N, D, float(np.sum(fin.coords)), float(np.sum(fin.log_prob)), int(s.random_state[2]),
                    int(s.random_state[1][0])))
        del s
    return log


if __name__ == "__main__":
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    t0 = time.perf_counter()
    a = one_pass(2024, count)
    b = one_pass(2024, count)
    n = min(len(a), len(b))
    ok = n > 0 and a[:n] == b[:n]
    print("stress_exact: %d samplers per pass, %.1f s, %s" % (n, time.perf_counter() - t0, "IDENTICAL" if ok else "MISMATCH"))
    if not ok:
        for x, y in zip(a[:n], b[:n]):
            if x != y:
                print("first difference:", x, y)
                break
    sys.exit(0 if ok else 1)
