"""Progress reporting for ``EnsembleSampler.sample(progress=...)``.

Same contract as the reference's ``pbar.get_progress_bar`` (a context manager with an
``update(n)`` method; ``progress`` may be ``False``, ``True`` or the name of a tqdm flavour such as
``"notebook"``), implemented as one small adapter class."""
import importlib
import logging

__all__ = ["get_progress_bar"]

logger = logging.getLogger(__name__)


class _Progress(object):
    """Context manager wrapping an optional tqdm instance."""

    def __init__(self, bar=None):
        self._bar = bar

    def __enter__(self):
        if self._bar is not None:
            self._bar.__enter__()
        return self

    def __exit__(self, *exc):
        if self._bar is not None:
            return self._bar.__exit__(*exc)
        return False

    def update(self, count):
        if self._bar is not None:
            self._bar.update(count)


def _tqdm_factory(flavour):
    """The ``tqdm`` class of the submodule ``tqdm.<flavour>`` (``tqdm.auto`` for ``True``), as the reference picks it
    (``pbar.py:54-58``); None when tqdm is not installed."""
    try:
        module = importlib.import_module("tqdm." + ("auto" if flavour is True else str(flavour)))
    except ImportError as exc:
        if exc.name == "tqdm":
            return None
        raise
    return module.tqdm


def get_progress_bar(display, total, **kwargs):
    """Return a progress context for ``total`` steps (``None`` = unknown length)."""
    if not display:
        return _Progress()
    factory = _tqdm_factory(display)
    if factory is None:
        logger.warning("You must install the tqdm library to use progress indicators with emcee")
        return _Progress()
    return _Progress(factory(total=total, **kwargs))
