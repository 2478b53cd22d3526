"""Step time vs ensemble size (native mode), to check the launch heuristics across N."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tools.ablate import run

if __name__ == "__main__":
    for target in ("dense", "iso"):
        for N in (256, 1024, 4096, 16384, 65536, 262144, 1048576):
            res = []
            for wpb in ((8, 4, 2, 1) if target == "dense" else (8,)):
                try:
                    res.append("wpb%d=%.2f" % (wpb, run(0, wpb=wpb, bpc=2, N=N, D=64, steps=100, target=target)))
                except Exception as e:  # noqa: BLE001
                    res.append("wpb%d=ERR(%s)" % (wpb, str(e)[:40]))
            print("%-5s N=%-8d us/step: %s" % (target, N, "  ".join(res)), flush=True)
