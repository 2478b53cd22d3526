"""Import reference emcee: from /root/reference/src (build container) or from oracle/_ref (anywhere).

TEST INFRASTRUCTURE.  The reference tree lacks the setuptools_scm generated
``emcee_version`` module (``setup.py:59-64``), so a stub is pre-registered; no
bytecode is written into the read-only tree (SURVEY.md 8c "Pitfall").
``/root/reference`` does not exist on the GPU box; ``tools/make_ref.sh`` (run by
``__graft_entry__.build()`` in the build container) materialises the reference's
package, unmodified, in the git-ignored ``oracle/_ref/emcee``, which travels to
the GPU box like the built ``libemx.so`` does -- so ``bench.py``'s ``cpu_baseline``
leg can time the reference itself there (``kind: "reference"``).  Callers gate on
:func:`available`; nothing under ``emcee_amd/`` may import this module
(``tests/test_layout.py``).
"""
import os
import sys
import types

REF_SRC = "/root/reference/src"
LOCAL_SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def source():
    """Directory holding the reference's ``emcee`` package, or None."""
    for d in (REF_SRC, LOCAL_SRC):
        if os.path.isfile(os.path.join(d, "emcee", "ensemble.py")):
            return d
    return None


def available():
    return source() is not None


def import_reference():
    src = source()
    if src is None:
        raise ImportError("reference emcee is present neither at %s nor at %s (tools/make_ref.sh)" % (REF_SRC, LOCAL_SRC))
    sys.dont_write_bytecode = True
    mod = sys.modules.get("emcee")
    if mod is not None and (getattr(mod, "__file__", "") or "").startswith(src):
        return mod
    if src not in sys.path:
        sys.path.insert(0, src)
    if not os.path.isfile(os.path.join(src, "emcee", "emcee_version.py")):
        stub = types.ModuleType("emcee.emcee_version")
        stub.__version__ = "0+reference"
        sys.modules["emcee.emcee_version"] = stub
    import emcee  # noqa: E402

    return emcee
