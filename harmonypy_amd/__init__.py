"""harmonypy_amd -- MI355X-native Harmony iteration engine.

Drop-in for the hot path of slowkow/harmonypy: ``run_harmony`` / ``Harmony`` keep the
reference's signatures (harmonypy/__init__.py:1-4, harmony.py:49-67, 218-229); the
``harmonize()`` loop runs as hand-written HIP kernels for gfx950 behind a C ABI
(include/hmx.h, harmonypy_amd/libhmx.so).  ``compute_lisi`` (lisi.py:24-66) is the
reference's integration metric on the same device.
"""
from .harmony import Harmony, run_harmony, BatchCodes  # noqa: F401
from .dist import Shard  # noqa: F401
from .lisi import compute_lisi  # noqa: F401

__version__ = "0.3.0"


def engine_version() -> str:
    """Identity of the kernel set in the loaded libhmx.so: the build id ``_build.py`` computed from csrc/ and hmx.h
    (``hmx_build_id()``).  profiles/*_pmc_hbm.json name the build their counters were collected on and bench.py only
    quotes counter-derived traffic for the build that is running -- nothing here is maintained by hand."""
    from . import _capi
    return _capi.build_id()


__all__ = ["Harmony", "run_harmony", "BatchCodes", "Shard", "compute_lisi", "__version__", "engine_version"]
