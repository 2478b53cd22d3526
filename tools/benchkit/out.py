"""bench.py: stdout carries exactly one JSON line -- everything else goes to stderr."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)



# ------------------------------------------------------------------------------------------------ stdout
# The contract is ONE JSON line on stdout.  Libraries in the process write there too (gloo announces its mesh, RCCL its
# version ...), so file descriptor 1 is pointed at stderr for the whole run and the line goes to a private copy of the
# original stdout.
_REAL_STDOUT = None


def _claim_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _REAL_STDOUT


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)
