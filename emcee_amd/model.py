"""Model: what a Move sees of the sampler (reference ``model.py:8-10``)."""
from collections import namedtuple

__all__ = ["Model"]

Model = namedtuple("Model", ("log_prob_fn", "compute_log_prob_fn", "map_fn", "random"))
