"""Progress bar shim (reference ``pbar.py``): tqdm when available, otherwise a no-op."""
import logging

__all__ = ["get_progress_bar"]

logger = logging.getLogger(__name__)

try:
    import tqdm
except ImportError:  # pragma: no cover
    tqdm = None


class _NoOpPBar(object):
    def __enter__(self, *a, **k):
        return self

    def __exit__(self, *a, **k):
        pass

    def update(self, count):
        pass


def get_progress_bar(display, total, **kwargs):
    """``display``: False -> no bar; True -> tqdm; a string selects ``tqdm.<name>.tqdm``."""
    if display:
        if tqdm is None:
            logger.warning("You must install the tqdm library to use progress indicators with emcee")
            return _NoOpPBar()
        if display is True:
            return tqdm.tqdm(total=total, **kwargs)
        return getattr(tqdm, "tqdm_" + display)(total=total, **kwargs)
    return _NoOpPBar()
