"""Import reference emcee from /root/reference/src (build container only).

TEST INFRASTRUCTURE.  The reference tree lacks the setuptools_scm generated
``emcee_version`` module (``setup.py:59-64``), so a stub is pre-registered; no
bytecode is written into the read-only tree (SURVEY.md 8c "Pitfall").
``/root/reference`` does not exist on the GPU box: callers must gate on
:func:`available`.
"""
import os
import sys
import types

REF_SRC = "/root/reference/src"


def available():
    return os.path.isdir(os.path.join(REF_SRC, "emcee"))


def import_reference():
    if not available():
        raise ImportError("reference emcee is not present at %s" % REF_SRC)
    sys.dont_write_bytecode = True
    if "emcee" in sys.modules and getattr(sys.modules["emcee"], "__file__", "").startswith(REF_SRC):
        return sys.modules["emcee"]
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    stub = types.ModuleType("emcee.emcee_version")
    stub.__version__ = "0+reference"
    sys.modules["emcee.emcee_version"] = stub
    import emcee  # noqa: E402

    return emcee
