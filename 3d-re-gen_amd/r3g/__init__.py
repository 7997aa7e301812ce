"""r3g -- host side of the MI355X-native Hunyuan_2d_to_3d hot path (ctypes over libr3g.so)."""
from .ffi import R3GError, LevelRangeError, NoSurfaceError, lib, context  # noqa: F401
