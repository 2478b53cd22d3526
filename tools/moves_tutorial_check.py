"""The reference's moves tutorial (docs/tutorials/moves.ipynb): 1-d bimodal target, Python log_prob_fn, 32 walkers, 5000
steps.  Published outputs: autocorrelation time 40.03 steps with the StretchMove, 6.49 steps with
[(DEMove, 0.8), (DESnookerMove, 0.2)]."""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
import emcee_amd  # noqa: E402


def logprob(x):
    return np.sum(np.logaddexp(-0.5 * (x - 2) ** 2, -0.5 * (x + 2) ** 2) - 0.5 * np.log(2 * np.pi) - np.log(2))


np.random.seed(589403)
init = np.random.randn(32, 1)
sampler0 = emcee_amd.EnsembleSampler(32, 1, logprob)
sampler0.run_mcmc(init, 5000)
t0 = float(sampler0.get_autocorr_time()[0])
print("StretchMove: autocorrelation time %.2f steps (tutorial 40.03)" % t0, flush=True)

np.random.seed(93284)
sampler = emcee_amd.EnsembleSampler(32, 1, logprob, moves=[(emcee_amd.moves.DEMove(), 0.8), (emcee_amd.moves.DESnookerMove(), 0.2)])
sampler.run_mcmc(init, 5000)
t1 = float(sampler.get_autocorr_time()[0])
print("DE + snooker: autocorrelation time %.2f steps (tutorial 6.49)" % t1, flush=True)
json.dump({"stretch_tau": t0, "de_snooker_tau": t1}, open("gpurun_out/moves_tutorial_check.json", "w"), indent=1)
