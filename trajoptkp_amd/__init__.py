"""trajoptkp_amd -- MI355X-native engine for the numerical hot path of keypoint-interpolated iLQR
(drop-in for the iLQR path of DMackRus/TrajOptKP).  The product is libkpilqr.so (HIP, gfx950) behind
the C ABI of include/kpilqr.h; this package is the thin Python host side used by tests and bench.py.
"""
from ._lib import Dims, build, load, LIB_PATH, SYMBOLS  # noqa: F401
from .engine import Engine, KpilqrError  # noqa: F401

__version__ = "0.1.0"
