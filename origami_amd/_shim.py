"""ctypes binding of libogdewarp.so (include/ogdewarp.h).

The library is the only compute path: there is no CPU fallback.  If the shared object is missing
or no gfx950 device is visible, calls raise `OgdError` loudly.
"""

import ctypes
import os
from ctypes import (
	POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_ssize_t, c_uint8,
	c_void_p)
from pathlib import Path

import numpy as np

INTER_NEAREST = 0  # == cv2.INTER_NEAREST
INTER_LINEAR = 1  # == cv2.INTER_LINEAR

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libogdewarp.so"


class OgdError(RuntimeError):
	def __init__(self, status, message):
		super().__init__("libogdewarp: %s (status %d)" % (message, status))
		self.status = status


_f32p = POINTER(c_float)
_f64p = POINTER(c_double)
_u8p = POINTER(c_uint8)
_i32p = POINTER(c_int32)
_i64p = POINTER(c_int64)

# name -> (restype, argtypes); mirrors include/ogdewarp.h one to one
SIGNATURES = {
	"ogd_last_error": (c_char_p, []),
	"ogd_version": (c_int, []),
	"ogd_device_count": (c_int, [POINTER(c_int)]),
	"ogd_init": (c_int, [c_int]),
	"ogd_device_pci_bus_id": (c_int, [c_int, c_char_p, c_int]),
	"ogd_trim": (c_int, [POINTER(c_size_t)]),
	"ogd_pool_busy": (c_int, [POINTER(c_int)]),
	"ogd_dev_alloc": (c_int, [POINTER(c_void_p), c_size_t]),
	"ogd_dev_free": (c_int, [c_void_p]),
	"ogd_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
	"ogd_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
	"ogd_stream_sync": (c_int, [c_void_p]),
	"ogd_memcpy2d_h2d": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),
	"ogd_memcpy2d_d2h": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),
	"ogd_mem_info": (c_int, [POINTER(c_size_t), POINTER(c_size_t)]),
	"ogd_host_alloc": (c_int, [POINTER(c_void_p), c_size_t]),
	"ogd_host_free": (c_int, [c_void_p]),
	"ogd_stream_create": (c_int, [POINTER(c_void_p)]),
	"ogd_stream_destroy": (c_int, [c_void_p]),
	"ogd_stream_wait_event": (c_int, [c_void_p, c_void_p]),
	"ogd_event_create": (c_int, [POINTER(c_void_p), c_int]),
	"ogd_event_destroy": (c_int, [c_void_p]),
	"ogd_event_record": (c_int, [c_void_p, c_void_p]),
	"ogd_event_sync": (c_int, [c_void_p]),
	"ogd_event_query": (c_int, [c_void_p, POINTER(c_int)]),
	"ogd_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
	"ogd_grid_fill": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p]),
	"ogd_grid_fill_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p]),
	"ogd_grid_fill_rows_dev": (c_int, [
		c_void_p, c_int, c_int, c_int, c_float, c_float, c_int, c_int, c_void_p, c_void_p]),
	"ogd_grid_points_strided": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
	"ogd_remap_u8": (c_int, [
		c_void_p, c_int, c_int, c_int, c_ssize_t, c_void_p, c_int, c_int, c_void_p, c_ssize_t, c_int]),
	"ogd_remap_u8_dev": (c_int, [
		c_void_p, c_int, c_int, c_int, c_ssize_t, c_void_p, c_int, c_int, c_void_p, c_ssize_t, c_int,
		c_void_p]),
	"ogd_dewarp_u8": (c_int, [
		c_void_p, c_int, c_int, c_int, c_ssize_t, c_void_p, c_int, c_int, c_int, c_float, c_float,
		c_void_p, c_ssize_t, c_int]),
	"ogd_dewarp_u8_dev": (c_int, [
		c_void_p, c_int, c_int, c_int, c_ssize_t, c_void_p, c_int, c_int, c_int, c_float, c_float,
		c_void_p, c_ssize_t, c_int, c_void_p]),
	"ogd_tri_create": (c_int, [
		c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, POINTER(c_void_p)]),
	"ogd_tri_create_scipy": (c_int, [
		c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, POINTER(c_void_p)]),
	"ogd_tri_destroy": (c_int, [c_void_p]),
	"ogd_tri_eval": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
	"ogd_tri_eval_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
	"ogd_tri_eval_ragged": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
	"ogd_tri_eval_ragged_dev": (c_int, [
		c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
	"ogd_rings_self_cross": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
	"ogd_tri_rasterize": (c_int, [
		c_void_p, c_void_p, c_void_p, c_double, c_double, c_double, c_int, c_int, c_void_p, c_int64,
		c_int64, c_int]),
	"ogd_tri_rasterize_dev": (c_int, [
		c_void_p, c_void_p, c_void_p, c_double, c_double, c_double, c_int, c_int, c_void_p, c_int64,
		c_int64, c_int, c_void_p]),
	"ogd_grid_sample_bilinear": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
	"ogd_page_create": (c_int, [c_void_p, c_int, c_int, c_int, c_ssize_t, POINTER(c_void_p)]),
	"ogd_page_destroy": (c_int, [c_void_p]),
	"ogd_page_dewarp_u8": (c_int, [
		c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_ssize_t, c_int]),
	"ogd_page_dewarp_jpeg_gray": (c_int, [
		c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, POINTER(c_size_t)]),
	"ogd_line_images_u8": (c_int, [
		c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
		c_void_p]),
	"ogd_line_images_u8_dev": (c_int, [
		c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
		c_void_p, c_void_p]),
	"ogd_lattice_delaunay": (c_int, [
		c_void_p, c_int, c_int, c_void_p, c_int64, POINTER(c_int64), c_void_p, POINTER(c_int), c_void_p]),
	"ogd_delaunay_points": (c_int, [
		c_void_p, c_int64, c_void_p, c_int64, POINTER(c_int64), c_void_p, POINTER(c_int64), c_void_p]),
	"ogd_wkt_ring_text": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p]),
	"ogd_png_info": (c_int, [c_void_p, c_int64, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
	"ogd_png_decode_u8": (c_int, [c_void_p, c_int64, c_void_p, c_int64]),
	"ogd_inflate_zlib": (c_int, [c_void_p, c_int64, c_void_p, c_int64, POINTER(c_int64)]),
	"ogd_deflate_raw": (c_int, [c_void_p, c_int64, c_void_p, c_int64, POINTER(c_int64)]),
	"ogd_png_bound": (c_int64, [c_int, c_int, c_int]),
	"ogd_png_encode_many": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
	"ogd_deflate_many": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
	"ogd_inflate_many": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
	"ogd_wkt_parse": (c_int, [
		c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
		POINTER(c_int64)]),
	"ogd_rings_area_signed": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
	"ogd_march_shape": (c_int, [
		c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_double, c_void_p, c_int, c_int, c_int,
		POINTER(c_int), POINTER(c_int)]),
	"ogd_march_grid": (c_int, [
		c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_double, c_void_p, c_int, c_int, c_int, c_int,
		c_void_p, c_void_p]),
	"ogd_march_grid_noalloc": (c_int, [
		c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_double, c_void_p, c_int, c_int, c_int, c_int,
		c_void_p, c_void_p]),
	"ogd_jpeg_encode_gray": (c_int, [c_void_p, c_int, c_int, c_ssize_t, c_int, c_void_p, c_size_t, POINTER(c_size_t)]),
	"ogd_jpeg_encode_gray_dev": (c_int, [
		c_void_p, c_int, c_int, c_ssize_t, c_int, c_void_p, c_size_t, POINTER(c_size_t), c_void_p]),
	"ogd_dewarp_jpeg_gray": (c_int, [
		c_void_p, c_int, c_int, c_ssize_t, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
		POINTER(c_size_t)]),
	"ogd_jpeg_info": (c_int, [c_void_p, c_size_t, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
	"ogd_jpeg_header": (c_int, [c_void_p, c_size_t, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
	"ogd_jpeg_dewarp_jpeg_gray": (c_int, [
		c_void_p, c_size_t, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, POINTER(c_size_t)]),
	"ogd_page_create_jpeg": (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
	"ogd_jpeg_decode_gray": (c_int, [c_void_p, c_size_t, c_void_p, c_int, c_int, c_ssize_t]),
	"ogd_jpeg_decode_gray_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_int, c_int, c_ssize_t, c_void_p]),
	"ogd_line_label_counts": (c_int, [
		c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_int,
		c_void_p]),
}

_lib = None


def load():
	"""dlopen libogdewarp.so and type its exports.  Makes no HIP call (safe before fork)."""
	global _lib
	if _lib is not None:
		return _lib
	path = Path(os.environ.get("OGDEWARP_LIB", LIB_PATH))
	if not path.exists():
		raise OgdError(-4, "%s is missing: build it with `python -m origami_amd.build`" % path)
	lib = ctypes.CDLL(str(path))
	for name, (restype, argtypes) in SIGNATURES.items():
		fn = getattr(lib, name)  # AttributeError if the header and the library disagree
		fn.restype = restype
		fn.argtypes = argtypes
	_lib = lib
	return lib


ERR_ASSERT = -7


def check(status):
	if status != 0:
		msg = load().ogd_last_error()
		msg = msg.decode("utf8", "replace") if msg else "unknown error"
		if status == ERR_ASSERT:  # where the reference has an `assert`
			raise AssertionError(msg)
		raise OgdError(status, msg)


def device_count():
	n = c_int(0)
	try:
		check(load().ogd_device_count(ctypes.byref(n)))
	except OgdError:
		return 0
	return n.value


def init(device=-1):
	check(load().ogd_init(int(device)))


def device_pci_bus_id(device):
	"""'0000:c1:00.0' of GPU `device`, or None (no such device, no driver)."""
	buf = ctypes.create_string_buffer(64)
	if load().ogd_device_pci_bus_id(int(device), buf, 64) != 0:
		return None
	return buf.value.decode("ascii", "replace").lower() or None


def _ptr(a):
	return c_void_p(a.ctypes.data)


def _c(a, dtype):
	return np.ascontiguousarray(a, dtype=dtype)


def _pixels_out(shape):
	"""Host array for page pixels coming back from the device: pinned memory (origami_amd.pinned — the copy runs at
	the link's rate instead of a quarter of it, and the array can go straight back up), pageable without a GPU."""
	from . import pinned
	return pinned.empty_if_available(shape, np.uint8)


# ---- K1a ------------------------------------------------------------------------------------------

def grid_fill(nodes, cell, sx=1.0, sy=1.0):
	"""(h, w, 2) float32 node grid -> (h*cell, w*cell, 2) float32 dense sampling grid."""
	nodes = _c(nodes, np.float32)
	if nodes.ndim != 3 or nodes.shape[2] != 2:
		raise ValueError("nodes must have shape (h, w, 2)")
	h, w = nodes.shape[:2]
	out = np.empty((h * cell, w * cell, 2), dtype=np.float32)
	check(load().ogd_grid_fill(_ptr(nodes), h, w, int(cell), float(sx), float(sy), _ptr(out)))
	return out


def grid_points_strided(nodes, cell, step):
	nodes = _c(nodes, np.float32)
	h, w = nodes.shape[:2]
	oh, ow = -(-h * cell // step), -(-w * cell // step)
	out = np.empty((oh, ow, 2), dtype=np.float32)
	check(load().ogd_grid_points_strided(_ptr(nodes), h, w, int(cell), int(step), _ptr(out)))
	return out


def grid_sample_bilinear(nodes, cell, q_yx):
	nodes = _c(nodes, np.float32)
	h, w = nodes.shape[:2]
	q = _c(q_yx, np.float64).reshape(-1, 2)
	out = np.empty_like(q)
	check(load().ogd_grid_sample_bilinear(_ptr(nodes), h, w, int(cell), _ptr(q), len(q), _ptr(out)))
	return out


# ---- K2 -------------------------------------------------------------------------------------------

def _image(a):
	a = np.asarray(a)
	if a.dtype != np.uint8:
		raise TypeError("image must be uint8, got %s" % a.dtype)
	if a.ndim == 2:
		a = a[:, :, np.newaxis]
		squeeze = True
	elif a.ndim == 3:
		squeeze = False
	else:
		raise ValueError("image must be (H, W) or (H, W, C)")
	if a.strides[2] != 1 or a.strides[1] != a.shape[2] or a.strides[0] < a.shape[1] * a.shape[2]:
		a = np.ascontiguousarray(a)
	return a, squeeze


def remap_u8(src, map_xy, interpolation=INTER_LINEAR):
	"""cv2.remap(src, map_xy, None, interpolation) for uint8 sources, BORDER_CONSTANT 0."""
	src, squeeze = _image(src)
	map_xy = _c(map_xy, np.float32)
	if map_xy.ndim != 3 or map_xy.shape[2] != 2:
		raise ValueError("map must have shape (Hd, Wd, 2)")
	H, W, C = src.shape
	Hd, Wd = map_xy.shape[:2]
	dst = _pixels_out((Hd, Wd, C))
	check(load().ogd_remap_u8(
		_ptr(src), H, W, C, src.strides[0], _ptr(map_xy), Hd, Wd, _ptr(dst), dst.strides[0],
		int(interpolation)))
	return dst[:, :, 0] if squeeze else dst


def dewarp_u8(src, nodes, cell, sx=1.0, sy=1.0, interpolation=INTER_LINEAR):
	"""Fused grid_fill + remap_u8: the dense map never exists in memory."""
	src, squeeze = _image(src)
	nodes = _c(nodes, np.float32)
	H, W, C = src.shape
	h, w = nodes.shape[:2]
	dst = _pixels_out((h * cell, w * cell, C))
	check(load().ogd_dewarp_u8(
		_ptr(src), H, W, C, src.strides[0], _ptr(nodes), h, w, int(cell), float(sx), float(sy),
		_ptr(dst), dst.strides[0], int(interpolation)))
	return dst[:, :, 0] if squeeze else dst


# ---- JPEG files (N4) ---------------------------------------------------------------------------------

def _jpeg_call(call, pixels_hint):
	"""Runs an encoder entry point into a buffer sized from the pixel count (JPEG at quality 75 stays
	far below half a byte per pixel); retried once with the exact size if a page is denser."""
	cap = max(4096, pixels_hint // 2 + 1024)
	for _ in range(2):
		out = np.empty(cap, dtype=np.uint8)
		size = c_size_t(0)
		status = call(_ptr(out), cap, ctypes.byref(size))
		if status == 0:
			return out[:size.value].tobytes()
		if size.value <= cap:
			check(status)
		cap = size.value
	check(status)


def jpeg_encode_gray(pixels, quality=75):
	"""The bytes PIL writes for Image.fromarray(pixels).save(f, "JPEG", quality=quality) ("L" image)."""
	a = np.asarray(pixels)
	if a.dtype != np.uint8 or a.ndim != 2:
		raise TypeError("jpeg_encode_gray takes a (H, W) uint8 array")
	if a.strides[1] != 1 or a.strides[0] < a.shape[1]:
		a = np.ascontiguousarray(a)
	H, W = a.shape
	lib = load()
	return _jpeg_call(
		lambda out, cap, size: lib.ogd_jpeg_encode_gray(_ptr(a), H, W, a.strides[0], int(quality), out, cap, size),
		H * W)


def dewarp_jpeg_gray(src, nodes, cell, quality=75):
	"""dewarp_u8 of a gray page followed by jpeg_encode_gray, the dewarped pixels staying on the device."""
	a = np.asarray(src)
	if a.dtype != np.uint8 or a.ndim != 2:
		raise TypeError("dewarp_jpeg_gray takes a (H, W) uint8 page")
	if a.strides[1] != 1 or a.strides[0] < a.shape[1]:
		a = np.ascontiguousarray(a)
	nodes = _c(nodes, np.float32)
	H, W = a.shape
	h, w = nodes.shape[:2]
	lib = load()
	return _jpeg_call(
		lambda out, cap, size: lib.ogd_dewarp_jpeg_gray(
			_ptr(a), H, W, a.strides[0], _ptr(nodes), h, w, int(cell), int(quality), out, cap, size),
		h * cell * w * cell)


def jpeg_dewarp_jpeg_gray(data, nodes, cell, quality=75):
	"""Scan file -> dewarped page's JPEG file, all on the device; None if the scan is outside the decoder's domain."""
	buf = np.frombuffer(data, dtype=np.uint8)
	nodes = _c(nodes, np.float32)
	h, w = nodes.shape[:2]
	lib = load()
	try:  # the library parses the file once and answers ERR_UNSUPPORTED for what ogd_jpeg_info calls unsupported
		return _jpeg_call(
			lambda out, cap, size: lib.ogd_jpeg_dewarp_jpeg_gray(
				_ptr(buf), buf.size, _ptr(nodes), h, w, int(cell), int(quality), out, cap, size),
			h * cell * w * cell)
	except OgdError as e:
		if e.status in (ERR_UNSUPPORTED, ERR_RUNTIME):
			return None
		raise


def jpeg_info(data):
	"""(H, W, components, supported) of a JPEG file held in `data` (bytes)."""
	buf = np.frombuffer(data, dtype=np.uint8)
	H, W, nc, ok = c_int(0), c_int(0), c_int(0), c_int(0)
	check(load().ogd_jpeg_info(_ptr(buf), buf.size, ctypes.byref(H), ctypes.byref(W), ctypes.byref(nc), ctypes.byref(ok)))
	return H.value, W.value, nc.value, bool(ok.value)


def jpeg_header(data):
	"""(H, W, components, plausible) from the marker segments alone (the scan is not walked: that is the decode call's
	single pass over the file)."""
	buf = np.frombuffer(data, dtype=np.uint8)
	H, W, nc, ok = c_int(0), c_int(0), c_int(0), c_int(0)
	check(load().ogd_jpeg_header(_ptr(buf), buf.size, ctypes.byref(H), ctypes.byref(W), ctypes.byref(nc), ctypes.byref(ok)))
	return H.value, W.value, nc.value, bool(ok.value)


def jpeg_decode_gray(data):
	"""np.array(PIL.Image.open(file)) for a baseline JPEG (grayscale, or YCbCr converted to "L" as PIL does), decoded on
	the device; None if the file is outside the decoder's domain (progressive, restart intervals, CMYK ...: the
	caller keeps PIL)."""
	buf = np.frombuffer(data, dtype=np.uint8)
	H, W, nc, ok = jpeg_header(data)
	if not ok:
		return None
	out = _pixels_out((H, W))
	status = load().ogd_jpeg_decode_gray(_ptr(buf), buf.size, _ptr(out), H, W, out.strides[0])
	if status in (ERR_UNSUPPORTED, ERR_RUNTIME):  # ERR_RUNTIME: truncated scan — PIL decides what to make of it
		return None
	check(status)
	return out


# ---- resident pages, per-line images (N1) ---------------------------------------------------------

class DevicePage:
	"""A uint8 page kept in device memory (ogd_page): upload once, dewarp / cut lines many times."""

	def __init__(self, pixels):
		src, self._squeeze = _image(pixels)
		self.shape = src.shape
		h = c_void_p()
		check(load().ogd_page_create(
			_ptr(src), src.shape[0], src.shape[1], src.shape[2], src.strides[0], ctypes.byref(h)))
		self._h = h

	@classmethod
	def from_jpeg(cls, data):
		"""The page of a grayscale baseline JPEG scan, decoded straight into device memory; None for any other
		file (decode it on the host and use DevicePage(pixels))."""
		H, W, nc, ok = jpeg_header(data)
		if not ok:
			return None
		buf = np.frombuffer(data, dtype=np.uint8)
		h = c_void_p()
		status = load().ogd_page_create_jpeg(_ptr(buf), buf.size, ctypes.byref(h))
		if status in (ERR_UNSUPPORTED, ERR_RUNTIME):
			return None
		check(status)
		self = cls.__new__(cls)
		self._squeeze = True
		self.shape = (H, W, 1)
		self._h = h
		return self

	def close(self):
		if getattr(self, "_h", None) is not None and _lib is not None:
			_lib.ogd_page_destroy(self._h)
		self._h = None

	__del__ = close

	def __enter__(self):
		return self

	def __exit__(self, *exc):
		self.close()

	@property
	def handle(self):
		if self._h is None:
			raise ValueError("page is closed")
		return self._h

	def dewarp(self, nodes, cell, sx=1.0, sy=1.0, interpolation=INTER_LINEAR):
		nodes = _c(nodes, np.float32)
		h, w = nodes.shape[:2]
		dst = _pixels_out((h * cell, w * cell, self.shape[2]))
		check(load().ogd_page_dewarp_u8(
			self.handle, _ptr(nodes), h, w, int(cell), float(sx), float(sy), _ptr(dst), dst.strides[0],
			int(interpolation)))
		return dst[:, :, 0] if self._squeeze else dst

	def dewarp_jpeg(self, nodes, cell, quality=75):
		"""The JPEG file of the dewarped page (what dewarp_jpeg_gray gives for the same pixels), gray pages only."""
		nodes = _c(nodes, np.float32)
		h, w = nodes.shape[:2]
		lib, handle = load(), self.handle
		return _jpeg_call(
			lambda out, cap, size: lib.ogd_page_dewarp_jpeg_gray(
				handle, _ptr(nodes), h, w, int(cell), int(quality), out, cap, size),
			h * cell * w * cell)

	@property
	def nbytes(self):
		return int(self.shape[0]) * int(self.shape[1]) * int(self.shape[2])

	def line_images(self, nodes, cell, lines, xsteps, ysteps, interpolation=INTER_LINEAR):
		"""lines: (n, 6) float64 rows p0.x p0.y right.x right.y up.x up.y -> list of n uint8 images."""
		nodes = _c(nodes, np.float32)
		h, w = nodes.shape[:2]
		lines = _c(lines, np.float64).reshape(-1, 6)
		n = len(lines)
		xs = _c(np.broadcast_to(xsteps, (n,)), np.int32)
		ys = _c(np.broadcast_to(ysteps, (n,)), np.int32)
		C = self.shape[2]
		offsets = np.zeros(n + 1, dtype=np.int64)
		np.cumsum(xs.astype(np.int64) * ys * C, out=offsets[1:])
		out = np.empty(int(offsets[-1]), dtype=np.uint8)
		check(load().ogd_line_images_u8(
			self.handle, _ptr(nodes), h, w, int(cell), _ptr(lines), _ptr(xs), _ptr(ys), n,
			int(interpolation), _ptr(out), _ptr(offsets)))
		images = []
		for k in range(n):
			im = out[offsets[k]:offsets[k + 1]].reshape(ys[k], xs[k], C)
			images.append(im[:, :, 0] if self._squeeze else im)
		return images

	def line_images_dev(self, nodes, cell, lines, xsteps, ysteps, d_out, interpolation=INTER_LINEAR, stream=None):
		"""line_images() with the result left on the device: d_out = device address of a buffer of at least the
		returned offsets[-1] bytes; returns the byte offsets (image l: ysteps[l] x xsteps[l] x C at offsets[l])."""
		nodes = _c(nodes, np.float32)
		h, w = nodes.shape[:2]
		lines = _c(lines, np.float64).reshape(-1, 6)
		n = len(lines)
		xs = _c(np.broadcast_to(xsteps, (n,)), np.int32)
		ys = _c(np.broadcast_to(ysteps, (n,)), np.int32)
		offsets = np.zeros(n + 1, dtype=np.int64)
		np.cumsum(xs.astype(np.int64) * ys * self.shape[2], out=offsets[1:])
		check(load().ogd_line_images_u8_dev(
			self.handle, _ptr(nodes), h, w, int(cell), _ptr(lines), _ptr(xs), _ptr(ys), n,
			int(interpolation), c_void_p(int(d_out)), _ptr(offsets), c_void_p(stream) if stream else None))
		return offsets

	def line_label_counts(self, nodes, cell, lines, xsteps, ysteps, sx, sy):
		"""This page being a label map: (n, 256) int32 histograms of the labels under each line lattice."""
		nodes = _c(nodes, np.float32)
		h, w = nodes.shape[:2]
		lines = _c(lines, np.float64).reshape(-1, 6)
		n = len(lines)
		xs = _c(np.broadcast_to(xsteps, (n,)), np.int32)
		ys = _c(np.broadcast_to(ysteps, (n,)), np.int32)
		counts = np.zeros((n, 256), dtype=np.int32)
		check(load().ogd_line_label_counts(
			self.handle, _ptr(nodes), h, w, int(cell), float(sx), float(sy), _ptr(lines), _ptr(xs), _ptr(ys),
			n, _ptr(counts)))
		return counts


# ---- N3: lattice Delaunay (host) --------------------------------------------------------------------

ERR_INVALID = -1
ERR_UNSUPPORTED = -5
ERR_RUNTIME = -6


def lattice_delaunay(points_hw2, transform=True):
	"""(h, w, 2) lattice positions -> (simplices (nt, 3) int32, hull vertex indices, transform (nt, 3, 2)),
	as scipy.spatial.Delaunay / ConvexHull give them; raises OgdError(ERR_UNSUPPORTED) outside its domain.
	transform=False: None in its place (Triangulation(..., scipy_transform=True) computes it on the device)."""
	pts = _c(points_hw2, np.float64)
	if pts.ndim != 3 or pts.shape[2] != 2:
		raise ValueError("lattice must have shape (h, w, 2)")
	h, w = pts.shape[:2]
	cap = 2 * h * w
	simplices = np.empty((cap, 3), dtype=np.int32)
	transform = np.empty((cap, 3, 2), dtype=np.float64) if transform else None
	hull = np.empty(2 * (h + w), dtype=np.int32)
	nt, nh = c_int64(0), c_int(0)
	check(load().ogd_lattice_delaunay(
		_ptr(pts), h, w, _ptr(simplices), cap, ctypes.byref(nt), _ptr(hull), ctypes.byref(nh),
		_ptr(transform) if transform is not None else None))
	return simplices[:nt.value], hull[:nh.value], transform[:nt.value] if transform is not None else None


def delaunay_points(points, twins=False, transform=True):
	"""(n, 2) scattered points -> (simplices (nt, 3) int32, transform (nt, 3, 2), n_ties) as scipy.spatial.Delaunay
	gives them when n_ties == 0; raises OgdError(ERR_UNSUPPORTED) for collinear / too few points and, unless
	twins=True, for coincident points.  twins=True appends a fourth result: twin (n,) int32, -1 for triangulated
	points, else the index of the coincident point that stands for this one.  transform=False: None in its place."""
	pts = _c(points, np.float64)
	if pts.ndim != 2 or pts.shape[1] != 2:
		raise ValueError("points must have shape (n, 2)")
	cap = 2 * len(pts) + 4
	simplices = np.empty((cap, 3), dtype=np.int32)
	transform = np.empty((cap, 3, 2), dtype=np.float64) if transform else None
	twin = np.empty(len(pts), dtype=np.int32) if twins else None
	nt, ties = c_int64(0), c_int64(0)
	check(load().ogd_delaunay_points(
		_ptr(pts), len(pts), _ptr(simplices), cap, ctypes.byref(nt),
		_ptr(transform) if transform is not None else None, ctypes.byref(ties), _ptr(twin) if twins else None))
	out = (simplices[:nt.value], transform[:nt.value] if transform is not None else None, ties.value)
	return out + (twin,) if twins else out


# ---- K3 / K1b -------------------------------------------------------------------------------------

class Triangulation:
	"""Device-resident piecewise-linear interpolant over a host (Qhull) triangulation.  Without `transform`:
	scipy_transform=True computes scipy.spatial.Delaunay's transform on the device (bit for bit the one
	lattice_delaunay / delaunay_points return), False the plain fp64 inverse of the edge matrix."""

	def __init__(self, points, simplices, values, transform=None, scipy_transform=False):
		points = _c(points, np.float64)
		simplices = _c(simplices, np.int32)
		values = _c(values, np.float64)
		if values.ndim == 1:
			values = values[:, np.newaxis]
		if transform is not None:
			transform = _c(transform, np.float64)
			if transform.shape != (len(simplices), 3, 2):
				raise ValueError("transform must have shape (nt, 3, 2)")
		if len(values) != len(points):
			raise ValueError("one value row per point expected")
		self.nv = values.shape[1]
		self._h = c_void_p()
		if transform is None and scipy_transform:
			check(load().ogd_tri_create_scipy(
				_ptr(points), len(points), _ptr(simplices), len(simplices), _ptr(values), self.nv,
				ctypes.byref(self._h)))
		else:
			check(load().ogd_tri_create(
				_ptr(points), len(points), _ptr(simplices), len(simplices),
				_ptr(transform) if transform is not None else None, _ptr(values), self.nv,
				ctypes.byref(self._h)))

	@property
	def handle(self):
		return self._h

	def close(self):
		if self._h:
			load().ogd_tri_destroy(self._h)
			self._h = c_void_p()

	def __del__(self):
		try:
			self.close()
		except Exception:
			pass


def _bounds(bounds):
	if bounds is None:
		return None, None
	b = np.array(bounds, dtype=np.float64).reshape(4)
	return b, _ptr(b)


def tri_eval(inner, extra, bounds, q):
	q = _c(q, np.float64).reshape(-1, 2)
	out = np.empty((len(q), extra.nv), dtype=np.float64)
	b, bp = _bounds(bounds)
	check(load().ogd_tri_eval(
		inner.handle if inner is not None else None, extra.handle, bp, _ptr(q), len(q), _ptr(out)))
	return out


def tri_eval_ragged(inner, extra, bounds, q, offsets):
	q = _c(q, np.float64).reshape(-1, 2)
	offsets = _c(offsets, np.int64)
	if len(offsets) < 1 or offsets[-1] != len(q):
		raise ValueError("offsets must end at the number of points")
	out = np.empty((len(q), extra.nv), dtype=np.float64)
	b, bp = _bounds(bounds)
	check(load().ogd_tri_eval_ragged(
		inner.handle if inner is not None else None, extra.handle, bp, _ptr(q), _ptr(offsets),
		len(offsets) - 1, _ptr(out)))
	return out


def rings_self_cross_table(flat, offsets):
	"""One flag per closed ring flat[offsets[r]:offsets[r + 1]]: do two of its non-adjacent segments touch or cross
	(wavefront per ring)."""
	flat = _c(flat, np.float64).reshape(-1, 2)
	offsets = _c(offsets, np.int64)
	flags = np.zeros(len(offsets) - 1, dtype=np.uint8)
	if len(flags):
		check(load().ogd_rings_self_cross(_ptr(flat), _ptr(offsets), len(flags), _ptr(flags)))
	return flags.astype(bool)


def rings_self_cross(rings):
	if not rings:
		return np.zeros(0, dtype=bool)
	offsets = np.zeros(len(rings) + 1, dtype=np.int64)
	np.cumsum([len(r) for r in rings], out=offsets[1:])
	flat = np.concatenate([np.asarray(r, dtype=np.float64).reshape(-1, 2) for r in rings])
	return rings_self_cross_table(flat, offsets)


def tri_rasterize(inner, extra, bounds, x0, y0, step, nx, ny, order="xy", dtype=np.float64):
	"""Values on the lattice (x0 + i*step, y0 + j*step); order "xy" -> (nx, ny, nv) like
	lingrid's mgrid layout, "yx" -> (ny, nx, nv) image layout."""
	nv = extra.nv
	dtype = np.dtype(dtype)
	if dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
		raise TypeError("dtype must be float32 or float64")
	if order == "xy":
		out = np.empty((nx, ny, nv), dtype=dtype)
		sx, sy = ny * nv, nv
	elif order == "yx":
		out = np.empty((ny, nx, nv), dtype=dtype)
		sx, sy = nv, nx * nv
	else:
		raise ValueError(order)
	b, bp = _bounds(bounds)
	check(load().ogd_tri_rasterize(
		inner.handle if inner is not None else None, extra.handle, bp, float(x0), float(y0),
		float(step), int(nx), int(ny), _ptr(out), sx, sy, int(dtype == np.float64)))
	return out


# ---- N2: the node-grid march ------------------------------------------------------------------------

def _field_args(field):
	"""field: (inner Triangulation or None, extra Triangulation) or a constant angle (float)."""
	if isinstance(field, tuple):
		inner, extra = field
		return (inner.handle if inner is not None else None, extra.handle, 0.0)
	return (None, None, float(field))


def march_shape(field_h, field_v, bounds, width, height, cell):
	"""(est_height, est_width) of GridFactory.grid_shape: both extent walks in one launch."""
	b, bp = _bounds(bounds)
	n_rows, n_cols = c_int(0), c_int(0)
	check(load().ogd_march_shape(
		*_field_args(field_h), *_field_args(field_v), bp, int(width), int(height), int(cell),
		ctypes.byref(n_rows), ctypes.byref(n_cols)))
	return n_rows.value, n_cols.value


def march_grid(field_h, field_v, bounds, cell, n_rows, n_cols, num_slices, noalloc=False):
	"""grid_h (with the reference's in-place row-end extension) and grid_hv before border extension.
	noalloc: through ogd_march_grid_noalloc (fixed device scratch, no allocation; at most 262144 nodes)."""
	b, bp = _bounds(bounds)
	grid_h = np.empty((n_rows, n_cols, 2), dtype=np.float32)
	grid_hv = np.empty((n_rows, n_cols, 2), dtype=np.float32)
	lib = load()
	check((lib.ogd_march_grid_noalloc if noalloc else lib.ogd_march_grid)(
		*_field_args(field_h), *_field_args(field_v), bp, int(cell), int(n_rows), int(n_cols),
		int(num_slices), _ptr(grid_h), _ptr(grid_hv)))
	return grid_h, grid_hv


# ---- WKT coordinate text (host code in the library) -------------------------------------------------

def wkt_ring_texts(rings):
	"""["(x y, x y, ...)", ...] for a list of (n, 2) arrays, formatted like GEOS' WKTWriter ("%.16g")."""
	if not rings:
		return []
	offsets = np.zeros(len(rings) + 1, dtype=np.int64)
	np.cumsum([len(r) for r in rings], out=offsets[1:])
	flat = np.concatenate([np.asarray(r, dtype=np.float64).reshape(-1, 2) for r in rings])
	return wkt_ring_texts_table(flat, offsets)


def wkt_ring_texts_table(flat, offsets):
	flat = _c(flat, np.float64).reshape(-1, 2)
	offsets = _c(offsets, np.int64)
	nrings = len(offsets) - 1
	if nrings <= 0:
		return []
	cap = 52 * int(offsets[-1]) + 2 * nrings
	buf = ctypes.create_string_buffer(cap)
	text_offsets = np.zeros(nrings + 1, dtype=np.int64)
	check(load().ogd_wkt_ring_text(
		_ptr(flat), _ptr(offsets), nrings, ctypes.cast(buf, c_void_p), cap, _ptr(text_offsets)))
	raw = buf.raw[:int(text_offsets[-1])].decode("ascii")
	bounds = text_offsets.tolist()
	return [raw[a:b] for a, b in zip(bounds[:-1], bounds[1:])]


def wkt_parse(texts):
	"""Many WKT texts (bytes) in one call -> (kinds (n,) int8 [0 polygon, 1 linestring, 2 multipolygon],
	pts (N, 2) float64, ring_offsets (R + 1,) int64, ring_geom (R,) int32, ring_part (R,) int32);
	raises OgdError(ERR_UNSUPPORTED) if any text is not plain POLYGON / LINESTRING / MULTIPOLYGON."""
	n = len(texts)
	blob = b"\n".join(texts)
	text_offsets = np.zeros(n + 1, dtype=np.int64)
	if n:
		np.cumsum([len(t) + 1 for t in texts], out=text_offsets[1:])
		text_offsets[-1] -= 1  # no separator after the last text; bytes objects end in a NUL
	cap_pts = blob.count(b",") + n + 1
	cap_rings = blob.count(b"(") + 1
	kinds = np.zeros(n, dtype=np.int8)
	pts = np.empty((cap_pts, 2), dtype=np.float64)
	ring_offsets = np.zeros(cap_rings + 1, dtype=np.int64)
	ring_geom = np.empty(cap_rings, dtype=np.int32)
	ring_part = np.empty(cap_rings, dtype=np.int32)
	nr = c_int64(0)
	# text g ends where its separator starts: offsets[g + 1] - 1 for all but the last
	ends = text_offsets.copy()
	check(load().ogd_wkt_parse(
		ctypes.cast(ctypes.c_char_p(blob), c_void_p), _ptr(ends), n, _ptr(kinds), _ptr(pts), cap_pts,
		_ptr(ring_offsets), _ptr(ring_geom), _ptr(ring_part), cap_rings, ctypes.byref(nr)))
	r = nr.value
	return kinds, pts[:ring_offsets[r]], ring_offsets[:r + 1], ring_geom[:r], ring_part[:r]


def rings_area_signed(flat, offsets):
	"""Signed areas (GEOS summation order) of the rings flat[offsets[r]:offsets[r + 1]]."""
	flat = _c(flat, np.float64).reshape(-1, 2)
	offsets = _c(offsets, np.int64)
	out = np.zeros(len(offsets) - 1, dtype=np.float64)
	check(load().ogd_rings_area_signed(_ptr(flat), _ptr(offsets), len(out), _ptr(out)))
	return out


def inflate_zlib(data, size):
	"""zlib.decompress(data) for a stream whose decompressed size is known to be at most `size` (host code)."""
	data = bytes(data)
	out = np.empty(max(int(size), 1), dtype=np.uint8)
	n = c_int64(0)
	check(load().ogd_inflate_zlib(ctypes.cast(ctypes.c_char_p(data), c_void_p), len(data), _ptr(out), int(size), ctypes.byref(n)))
	return out[:n.value]


def png_info(data):
	"""(width, height, channels) of an 8-bit non-interlaced grey / RGB PNG; OgdError(ERR_UNSUPPORTED) for other kinds."""
	data = bytes(data)
	w, h, c = c_int(0), c_int(0), c_int(0)
	check(load().ogd_png_info(ctypes.cast(ctypes.c_char_p(data), c_void_p), len(data), ctypes.byref(w), ctypes.byref(h), ctypes.byref(c)))
	return w.value, h.value, c.value


def png_decode(data):
	"""(H, W) or (H, W, 3) uint8 pixels of a PNG file's bytes, as np.array(PIL.Image.open(...)) gives them."""
	data = bytes(data)
	w, h, c = png_info(data)
	out = _pixels_out((h, w) if c == 1 else (h, w, c))
	check(load().ogd_png_decode_u8(ctypes.cast(ctypes.c_char_p(data), c_void_p), len(data), _ptr(out), w * c))
	return out


def deflate_raw(data):
	"""A raw deflate stream (zlib.decompress(x, -15) reads it) of `data`: Huffman-coded literals, host code."""
	data = bytes(data)
	cap = len(data) + len(data) // 1000 + 64
	out = ctypes.create_string_buffer(cap)
	n = c_int64(0)
	check(load().ogd_deflate_raw(ctypes.cast(ctypes.c_char_p(data), c_void_p), len(data), ctypes.cast(out, c_void_p), cap, ctypes.byref(n)))
	return out.raw[:n.value]


def inflate_many(buf, members):
	"""The members of a zip archive held in `buf` (bytes), all in one call: members = [(method, crc32, compressed
	size, size, data offset)] -> list of bytes.  OgdError(ERR_INVALID) for a damaged member."""
	n = len(members)
	if n == 0:
		return []
	table = np.array(members, dtype=np.int64).reshape(n, 5)
	methods = np.ascontiguousarray(table[:, 0], dtype=np.int32)
	crcs = np.ascontiguousarray(table[:, 1], dtype=np.uint32)
	in_len = np.ascontiguousarray(table[:, 2])
	in_off = np.ascontiguousarray(table[:, 4])
	out_off = np.zeros(n + 1, dtype=np.int64)
	np.cumsum(table[:, 3], out=out_off[1:])
	out = ctypes.create_string_buffer(max(int(out_off[-1]), 1))
	check(load().ogd_inflate_many(
		ctypes.cast(ctypes.c_char_p(buf), c_void_p), len(buf), _ptr(in_off), _ptr(in_len), _ptr(methods), n,
		ctypes.cast(out, c_void_p), _ptr(out_off), _ptr(crcs)))
	raw = out.raw
	bounds = out_off.tolist()
	return [raw[a:b] for a, b in zip(bounds[:-1], bounds[1:])]


def png_encode_many(images, with_crc=False, views=False):
	"""PNG files (bytes) of a list of uint8 images, all (H, W) grey or all (H, W, 3) RGB, in one call: the same
	pixels for any PNG reader as PIL's `save(format="png")`, written ~25x faster (host code; ogd_png_encode_many).
	with_crc: [(file, CRC-32 of the file)] instead, for a zip archive that stores the files.
	views: the files as memoryviews into one buffer instead of bytes objects (no copy; for a writer that only passes
	them on)."""
	n = len(images)
	if n == 0:
		return []
	arrays = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
	C = 1 if arrays[0].ndim == 2 else arrays[0].shape[2]
	for a in arrays:
		if a.ndim not in (2, 3) or (1 if a.ndim == 2 else a.shape[2]) != C or a.size == 0:
			raise TypeError("png_encode_many takes non-empty images that are all grey or all RGB")
	heights = np.array([a.shape[0] for a in arrays], dtype=np.int32)
	widths = np.array([a.shape[1] for a in arrays], dtype=np.int32)
	pointers = np.array([a.ctypes.data for a in arrays], dtype=np.uintp)  # `arrays` keeps them alive
	lib = load()
	raw = heights.astype(np.int64) * (widths.astype(np.int64) * C + 1)
	starts = np.zeros(n + 1, dtype=np.int64)
	np.cumsum(raw + raw // 1000 + 127, out=starts[1:])  # ogd_png_bound per image
	assert int(starts[1]) == lib.ogd_png_bound(int(heights[0]), int(widths[0]), C)
	out = np.empty(int(starts[-1]), dtype=np.uint8)
	sizes = np.zeros(n, dtype=np.int64)
	crcs = np.zeros(n, dtype=np.uint32)
	check(lib.ogd_png_encode_many(
		_ptr(pointers), _ptr(widths), _ptr(heights), C, n, _ptr(out), _ptr(starts), _ptr(sizes), _ptr(crcs)))
	if views:
		whole = memoryview(out)
		files = [whole[a:a + m] for a, m in zip(starts[:-1].tolist(), sizes.tolist())]
	else:
		files = [out[a:a + m].tobytes() for a, m in zip(starts[:-1].tolist(), sizes.tolist())]
	return list(zip(files, crcs.tolist())) if with_crc else files


def deflate_many(members):
	"""[(deflate stream, crc32)] of a list of bytes objects in one call (the members of a zip archive)."""
	n = len(members)
	if n == 0:
		return []
	blob = b"".join(members)
	offsets = np.zeros(n + 1, dtype=np.int64)
	np.cumsum([len(m) for m in members], out=offsets[1:])
	cap = len(blob) + len(blob) // 1000 + 64 * n + 64
	out = ctypes.create_string_buffer(cap)
	out_offsets = np.zeros(n + 1, dtype=np.int64)
	crcs = np.zeros(n, dtype=np.uint32)
	check(load().ogd_deflate_many(
		ctypes.cast(ctypes.c_char_p(blob), c_void_p), _ptr(offsets), n, ctypes.cast(out, c_void_p), cap,
		_ptr(out_offsets), _ptr(crcs)))
	raw = out.raw[:int(out_offsets[-1])]
	bounds = out_offsets.tolist()
	return [(raw[a:b], int(c)) for a, b, c in zip(bounds[:-1], bounds[1:], crcs.tolist())]
