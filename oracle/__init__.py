"""CPU oracle for the red/blue split-ensemble hot path -- TEST INFRASTRUCTURE ONLY.

This package is a NumPy restatement of the algorithm reference emcee runs on
its hot path.  It exists so that the HIP kernels in
``emcee_amd/csrc`` can be checked on a box where ``/root/reference`` is absent.

Rules (enforced by tests/test_layout.py):
  * Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import anything from here.
  * Nothing under ``emcee_amd/`` imports it; the product path has no CPU
    fallback and raises when the HIP library is missing.

Pinning: the reference holds no golden vectors for this path, so
the oracle is pinned against outputs of the reference itself, run in the build
container by ``oracle/gen_golden.py`` (which imports ``/root/reference/src``
through a version shim) and committed under ``tests/golden/``.
``tests/test_oracle_golden.py`` replays every fixture through this oracle and
requires bit-identical coords, log-probs, accept masks, split labels, partner
indices and final MT19937 state.

Third-party arithmetic on the path: NumPy's legacy ``RandomState`` (MT19937).
It is not vendored in the reference (``setup.py:25`` pins no version); the
oracle uses NumPy's own ``RandomState`` (numpy 2.2.6 in this image; the legacy
stream is frozen by NumPy's compatibility policy), while the product carries an
independent C++ re-implementation (``emcee_amd/csrc/mt19937_legacy.hpp``) that
the CPU tests compare word-for-word against NumPy.
"""
