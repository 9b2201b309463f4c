"""gpu_raytracer_amd — MI355X-native ray-casting hot path of kije/gpu_raytracer.

Only what the hot path needs lives here: csrc/ (HIP kernels + the C ABI of
include/rt_hip.h, built to librt_hip.so), the ctypes binding (api.py) and the
numpy harness that produces reference-layout inputs (types.py, hostpack.py,
scenes.py).  The CPU oracle is NOT part of this package (see oracle/).
"""
__version__ = "0.1.0"
