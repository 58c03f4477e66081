"""raytracing_amd — MI355X-native ray-casting hot path of mbillingr/raytracing.

Layers:
  * ``librtmi.so``  — hand-written HIP kernels for gfx950 behind the C ABI of ``include/rtmi.h``
  * ``_front``      — the C++ host front end (``include/rtmi.hpp``: World / Shape / Phong / Camera ...
                      with the reference's names), bound with pybind11
  * ``abi``         — ctypes view of the C ABI for device-resident rendering (torch tensors in, no copies)
  * ``scenes``      — the BASELINE scenes restated with the front end
  * ``tiles``       — row-tile sharding + RCCL gather for one-process-per-GPU runs

There is no CPU fallback anywhere in this package: rendering needs ``librtmi.so`` and a GPU.
"""
import os as _os

# torch wheels bundle their own libamdhip64; a process must not end up with two HIP runtimes (the second one
# sees no GPU).  Importing torch first makes librtmi.so bind to the runtime torch already loaded.
if not _os.environ.get("RTMI_NO_TORCH"):
    try:
        import torch as _torch  # noqa: F401
    except ImportError:
        _torch = None

_HERE = _os.path.dirname(_os.path.abspath(__file__))

try:
    from ._front import *  # noqa: F401,F403  (the reference's API names)
    from . import _front
except ImportError as e:  # fail loudly: the native extension IS the product
    raise ImportError(
        "raytracing_amd: native extension missing (%s). Build it with "
        "`make -C raytracing_amd/csrc` (needs hipcc + pybind11)." % (e,)
    ) from e

LIB_PATH = _os.environ.get("RTMI_LIB") or _os.path.join(_HERE, "librtmi.so")  # RTMI_LIB: A/B builds of the same ABI
PRECISION_F32 = 0
PRECISION_F64 = 1
