"""TEST-ONLY numpy shim of the `autoray` surface that cotengra touches.

This is build-authored test tooling (NOT reference code and NOT product code).
autoray is cotengra's single hard dependency (reference `pyproject.toml:33`) and
is not installed in this image; this shim lets the real reference be imported
*in the build container only* so that `tests/golden/make_golden.py` can pin the
oracle and generate golden vectors.  Nothing under `cotengra_amd/` imports it and
it never runs on the GPU box (the reference does not travel there).

Call sites served: reference `cotengra/contract.py:8,339-409,744-746`,
`cotengra/core.py:11,157-159,3874`, `cotengra/interface.py:5,507,612-631,862`,
`cotengra/utils.py:14,1578,1594`.
"""
import contextlib

import numpy as _np


def infer_backend(x):
    return "numpy"


def infer_backend_multi(*xs):
    return "numpy"


def get_namespace(backend=None):
    return _np


def shape(x):
    try:
        return tuple(int(d) for d in x.shape)
    except AttributeError:
        return tuple(int(d) for d in _np.shape(x))


def do(name, *args, like=None, **kwargs):
    if name == "astype":
        x, dtype = args
        return _np.asarray(x).astype(dtype)
    return getattr(_np, name)(*args, **kwargs)


def to_numpy(x):
    return _np.asarray(x)


@contextlib.contextmanager
def backend_like(backend):
    yield


def autojit(fn=None, **kwargs):
    if fn is None:
        return lambda f: f
    return fn


class _Lazy:
    class Variable:
        def __init__(self, *a, **k):
            raise NotImplementedError("autoray.lazy is not part of the shim")


lazy = _Lazy()
