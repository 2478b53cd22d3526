"""Chain storage.  ``Backend`` keeps the chain in HBM next to the kernels that append to it.
``HDFBackend`` (reference ``backends/hdf.py``) is host file I/O outside this path: pass a
reference ``emcee.backends.HDFBackend`` instance as ``backend=`` -- the sampler feeds any object
with the reference ``Backend`` interface through ``save_step``."""
from .backend import Backend

__all__ = ["Backend"]
