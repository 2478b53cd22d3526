import sys
sys.path.insert(0, ".")
from tools.ablate import run
if __name__ == "__main__":
    for target in ("iso", "dense"):
        for wpb in (8, 4):
            print(target, "wpb", wpb, "empty-launch floor us/step: %.2f" % run(64, wpb=wpb, target=target), " full: %.2f" % run(0, wpb=wpb, target=target), flush=True)
