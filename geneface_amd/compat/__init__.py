"""The four operator modules GeneFace's wrappers import by name.

`install()` registers them in sys.modules under the reference's extension names so that an
*unmodified* GeneFace checkout picks up the MI355X kernels through its own
`try: import _raymarching_face as _backend` seam (raymarching/raymarching.py:9-12,
gridencoder/grid.py:9-12, shencoder/sphere_harmonics.py:9-12, freqencoder/freq.py:9-12).
"""
import sys

from . import _freqencoder, _gridencoder, _raymarching_face, _shencoder

NAMES = ("_raymarching_face", "_gridencoder", "_shencoder", "_freqencoder")


def install(force: bool = False):
    for name, mod in zip(NAMES, (_raymarching_face, _gridencoder, _shencoder, _freqencoder)):
        if force or name not in sys.modules:
            sys.modules[name] = mod
