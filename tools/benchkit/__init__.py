"""The pieces of bench.py (repo root): model (workloads, byte models), out (the one JSON line), cpu (CPU baseline), single (one GPU),
sharded (N > 1), launcher (self-launch).  bench.py is the command line and the line's assembly."""
