"""Exact mode at C4's size: the host pipeline's stage times (and emx_run's rate) for DE-only, snooker-only and the 0.8 / 0.2
mixture, 65 536 x 64 -- which move costs what, before building the device finish of DE / snooker steps (round 6)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tools.benchkit.model import Workload
from tools.benchkit.single import measure_single
bench._claim_stdout()
os.environ["EMX_PIPE_STATS"] = "1"
tunes = [json.loads(a) for a in sys.argv[1:]] or [{}]
for name, w in (("de_only", [1.0, 0.0]), ("snooker_only", [0.0, 1.0]), ("mix_0.8_0.2", [0.8, 0.2])):
    for tune in tunes:
        wl = Workload("c4", 65536)
        wl.weights = w
        res = measure_single(wl, 100, 20, device=0, rng="mt19937", spin_s=0.05, want_kernel=False, tuning=tune)
        p = res.get("pipeline") or {}
        print("%-14s %-40s %.2f us/step | gen %.1f tok %.1f fin(sum) %.1f [%s] waits words %.1f consumer %.1f" % (
            name, json.dumps(tune), res["wall_s"] * 1e4, p.get("generator_us", 0), p.get("tokenizer_us", 0), p.get("finishers_us_summed", 0),
            p.get("finisher_threads"), p.get("tokenizer_waited_for_words_us", 0), p.get("tokenizer_waited_for_consumer_us", 0)), file=sys.stderr, flush=True)
