"""lws_amd -- MI355X-native drop-in for the hot path of Jonathan-LeRoux/lws.

``import lws_amd as lws`` gives the reference module's surface (python/lws.pyx): the helper
functions, ``batch_lws`` / ``nofuture_lws`` / ``online_lws`` and ``class lws``; the per-bin update
loops run as HIP kernels on gfx950 through the C ABI of include/lws_hip.h.
"""
from .lws import (  # noqa: F401
    __version__, hann, synthwin, stft, istft, get_consistency, extspec, create_weights,
    build_asymmetric_windows, get_thresholds, batch_lws, nofuture_lws, online_lws, lws, clear_plan_cache,
    stft_dev, istft_dev,
)
from . import _capi  # noqa: F401
from ._capi import Plan, MultiPlan, LwsHipError  # noqa: F401
