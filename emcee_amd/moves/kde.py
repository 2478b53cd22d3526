"""KDEMove: proposals resampled from a Gaussian KDE of the complement (reference
``moves/kde.py:16-45``).  Host proposal (scipy), device accept/commit like every custom
split-ensemble move."""
import numpy as np

from .red_blue import RedBlueMove

__all__ = ["KDEMove"]


class KDEMove(RedBlueMove):
    """:param bw_method: bandwidth rule passed to ``scipy.stats.gaussian_kde``."""

    def __init__(self, bw_method=None, **kwargs):
        try:
            from scipy.stats import gaussian_kde  # noqa: F401
        except ImportError:
            raise ImportError("you need scipy.stats.gaussian_kde to use the KDEMove")
        self.bw_method = bw_method
        super(KDEMove, self).__init__(**kwargs)

    def get_proposal(self, s, c, random):
        from scipy.stats import gaussian_kde
        density = gaussian_kde(np.concatenate(c, axis=0).T, bw_method=self.bw_method)
        q = density.resample(len(s), random)
        log_ratio = density.logpdf(s.T) - density.logpdf(q)
        return q.T, log_ratio
