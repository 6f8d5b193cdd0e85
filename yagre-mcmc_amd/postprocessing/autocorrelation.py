"""Autocorrelation post-processing on the device (reference:
yagremcmc/postprocessing/autocorrelation.py).  Same function names and
arguments for host arrays; for an ensemble sampler use
`sampler.integrated_autocorrelation(...)`, which works on the record in HBM
and returns one value per chain."""
import numpy as np

from .. import _ffi

_METHODS = {"mean": 0, "max": 1}


def estimate_autocorrelation_function_1d(sequence, device=0):
    """autocorrelation.py:5-29: the normalised autocorrelation of a 1-D sequence, all lags."""
    seq = np.asarray(sequence, dtype=np.float64).reshape(1, -1)
    _, acf = _ffi.default_context(device).iat_series(seq, acf_lags=seq.shape[1])
    return acf[0]


def integrated_autocorrelation(seq, method='mean', sokalConst=5., device=0, allowUndefined=False):
    """autocorrelation.py:92-140.  `seq`: (n, d) states of one chain -> int, or
    (n, K, d) states of K chains -> array of K ints.  A constant series has no autocorrelation:
    ValueError (the reference fails on 0 / 0 too), or IAT_UNDEFINED (INT32_MIN) in its place with allowUndefined."""
    if method not in _METHODS:
        raise ValueError(f"Invalid IAT - Type: {method}. Options are 'mean' and 'max'.")
    arr = np.asarray(seq, dtype=np.float64)
    if arr.ndim == 1:
        arr = arr.reshape(-1, 1)
    single = arr.ndim == 2
    if single:
        arr = arr[:, None, :]
    out = _ffi.default_context(device).iat_states(arr, _METHODS[method], sokalConst, allowUndefined)
    return int(out[0]) if single else out
