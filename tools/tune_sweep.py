import sys
sys.path.insert(0, ".")
from tools.ablate import run
for rep in range(2):
    for wpb in (2, 4, 8):
        for bpc in (1, 2, 4):
            print("wpb=%d bpc=%d  %.2f us/step" % (wpb, bpc, run(0, wpb, bpc, steps=400)), flush=True)
