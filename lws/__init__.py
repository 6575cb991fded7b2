"""``import lws`` -- the reference's module name (python/lws.pyx, setup.py:71-75), served by the MI355X engine.

The reference ships one extension module called ``lws``; user code is ``import lws; lws.lws(512, 128).run_lws(X0)``
(python/README.md:92-102).  This package is that name: it re-exports the whole surface of ``lws_amd`` (same
functions, same ``class lws``), so the README snippet runs unchanged on the GPU.  Nothing is implemented here.
"""
from lws_amd import (  # noqa: F401
    __version__, hann, synthwin, stft, istft, get_consistency, extspec, create_weights,
    build_asymmetric_windows, get_thresholds, batch_lws, nofuture_lws, online_lws, lws,
)
import lws_amd as _engine  # noqa: F401  (lws._engine.Plan, ._capi for device-resident use)
