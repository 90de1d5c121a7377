"""ctypes front of oracle/cv_oracle.c: cv2.resize(INTER_LINEAR, f32) and cv2.remap(u8) restated.

Test infrastructure only; see oracle/__init__.py.  PARITY UNPINNED (no cv2 here, no fixtures in
the reference).
"""

import ctypes
from ctypes import c_int, c_ssize_t, c_void_p

import numpy as np

from .build_oracle import LIB

INTER_NEAREST = 0
INTER_LINEAR = 1

_lib = None


def _load():
	global _lib
	if _lib is None:
		if not LIB.exists():
			raise RuntimeError("%s missing: run `python -m oracle.build_oracle`" % LIB)
		lib = ctypes.CDLL(str(LIB))
		lib.orc_resize_linear_f32.restype = c_int
		lib.orc_resize_linear_f32.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_int]
		lib.orc_remap_u8.restype = c_int
		lib.orc_remap_u8.argtypes = [
			c_void_p, c_int, c_int, c_int, c_ssize_t, c_void_p, c_int, c_int, c_void_p, c_ssize_t,
			c_int, c_int]
		lib.orc_max_threads.restype = c_int
		_lib = lib
	return _lib


def max_threads():
	return _load().orc_max_threads()


def resize_linear_f32(plane, s, threads=1):
	"""cv2.resize(plane, (w*s, h*s), interpolation=cv2.INTER_LINEAR) for one float32 plane.
	ref: origami/core/dewarp.py:522-523."""
	plane = np.ascontiguousarray(plane, dtype=np.float32)
	h, w = plane.shape
	out = np.empty((h * s, w * s), dtype=np.float32)
	rc = _load().orc_resize_linear_f32(plane.ctypes.data, h, w, int(s), out.ctypes.data, int(threads))
	if rc != 0:
		raise RuntimeError("orc_resize_linear_f32 failed (%d)" % rc)
	return out


def dense_grid(nodes, s, threads=1):
	"""Grid.points("full"): dstack of the two resized planes.  ref: origami/core/dewarp.py:518-524."""
	nodes = np.asarray(nodes, dtype=np.float32)
	xs = resize_linear_f32(nodes[:, :, 0], s, threads)
	ys = resize_linear_f32(nodes[:, :, 1], s, threads)
	return np.dstack([xs, ys])


def remap_u8(src, map_xy, interpolation=INTER_LINEAR, threads=1):
	"""cv2.remap(src, map_xy, None, interpolation) for uint8 sources (BORDER_CONSTANT 0).
	ref: origami/core/dewarp.py:612-614, 625-627."""
	src = np.ascontiguousarray(src, dtype=np.uint8)
	squeeze = src.ndim == 2
	if squeeze:
		src = src[:, :, np.newaxis]
	map_xy = np.ascontiguousarray(map_xy, dtype=np.float32)
	H, W, C = src.shape
	Hd, Wd = map_xy.shape[:2]
	dst = np.empty((Hd, Wd, C), dtype=np.uint8)
	rc = _load().orc_remap_u8(
		src.ctypes.data, H, W, C, src.strides[0], map_xy.ctypes.data, Hd, Wd, dst.ctypes.data,
		dst.strides[0], int(interpolation), int(threads))
	if rc != 0:
		raise RuntimeError("orc_remap_u8 failed (%d)" % rc)
	return dst[:, :, 0] if squeeze else dst
