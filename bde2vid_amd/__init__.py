"""MI355X-native BDE2VID inference path (hand-written gfx950 HIP kernels behind a C ABI).

Only the hot path of SURVEY.md §8 lives here.  Importing the package is cheap and
GPU-free; anything that computes goes through `bde2vid_amd._lib` (ctypes over
`libbde2vid.so`) and raises if the library is missing -- there is no CPU fallback.
"""
from .config import GeneratorConfig, canonical  # noqa: F401

__all__ = ['GeneratorConfig', 'canonical']
