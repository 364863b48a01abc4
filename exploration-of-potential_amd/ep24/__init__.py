"""ep24 - MI355X-native training path of the YOLOX-24p detector (HIP kernels behind include/ep24.h).

Submodules: ``_lib`` (ctypes binding), ``nn`` (parameter tree with the reference's names), ``engine``
(static launch plan), ``loss`` (SimOTA + 24-circle loss), ``train`` (captured step, fused SGD), ``dp``
(RCCL gradient reduction), ``sector`` (fisheye sector warp), ``synth`` (synthetic inputs).
"""
__all__ = ["nn", "engine", "loss", "train", "dp", "sector", "synth"]
