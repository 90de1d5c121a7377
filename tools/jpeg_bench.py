#!/usr/bin/env python3
"""N4 measurement: the JPEG file of a dewarped broadsheet page — PIL / libjpeg-turbo on one host core (what
the reference's `image.save(path)` costs) against the device-side encoder.
usage: jpeg_bench.py [W H reps]   (default 6000 8000 5: the dewarped page is 6075 x 8075)
Kernel times: run under `rocprofv3 --kernel-trace --stats`."""
import ctypes
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, synth  # noqa: E402
from oracle import jpeg as ojpeg  # noqa: E402

_shim.init(0)
lib = _shim.load()
W, H, reps = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (6000, 8000, 5)
s = 25
rng = np.random.default_rng(11)
page = synth.text_page(rng, (W, H))
nodes = synth.Warp.random(rng, (W, H)).node_grid(s)
h, w = nodes.shape[:2]
Hd, Wd = h * s, w * s
dewarped = _shim.dewarp_u8(page, nodes, s)

t0 = time.perf_counter()
want = ojpeg.save_gray(dewarped)
t_pil = time.perf_counter() - t0
print("dewarped page %dx%d (%.1f MP), JPEG quality 75: %d bytes (%.3f B/px)" % (
	Wd, Hd, Hd * Wd / 1e6, len(want), len(want) / (Hd * Wd)))
print("PIL / libjpeg-turbo, 1 host thread:        %8.1f ms = %7.0f MP/s" % (t_pil * 1e3, Hd * Wd / 1e6 / t_pil))


def timed(fn):
	fn()
	t = time.perf_counter()
	for _ in range(reps):
		out = fn()
	return (time.perf_counter() - t) / reps, out


dt, got = timed(lambda: _shim.jpeg_encode_gray(dewarped))
assert got == want
print("ogd_jpeg_encode_gray (host in, host out):  %8.1f ms = %7.0f MP/s" % (dt * 1e3, Hd * Wd / 1e6 / dt))

# device-resident image -> host file
d_img = ctypes.c_void_p()
_shim.check(lib.ogd_dev_alloc(ctypes.byref(d_img), Hd * Wd))
_shim.check(lib.ogd_memcpy_h2d(d_img, dewarped.ctypes.data, Hd * Wd, None))
_shim.check(lib.ogd_stream_sync(None))
out = np.empty(len(want) + 64, np.uint8)
size = ctypes.c_size_t(0)


def dev_call():
	_shim.check(lib.ogd_jpeg_encode_gray_dev(d_img, Hd, Wd, Wd, 75, out.ctypes.data, out.size, ctypes.byref(size), None))
	return out[:size.value].tobytes()


dt, got = timed(dev_call)
assert got == want
print("ogd_jpeg_encode_gray_dev (device in):      %8.1f ms = %7.0f MP/s" % (dt * 1e3, Hd * Wd / 1e6 / dt))
_shim.check(lib.ogd_dev_free(d_img))

dt, got = timed(lambda: _shim.dewarp_jpeg_gray(page, nodes, s))
assert got == want
print("ogd_dewarp_jpeg_gray (page in, file out):  %8.1f ms" % (dt * 1e3))
dt2, _ = timed(lambda: ojpeg.save_gray(_shim.dewarp_u8(page, nodes, s)))
print("ogd_dewarp_u8 + PIL save (before):         %8.1f ms" % (dt2 * 1e3))
print("all outputs byte-identical to PIL's file")

# ---- decoding the scan (ref: origami/core/page.py:82-84) ----
import io
import PIL.Image
scan = ojpeg.save_gray(page)
t0 = time.perf_counter()
want_px = np.array(PIL.Image.open(io.BytesIO(scan)).convert("L"))
t_pil = time.perf_counter() - t0
print("scan %dx%d as a JPEG file of %d bytes" % (W, H, len(scan)))
print("PIL / libjpeg-turbo decode, 1 host thread:  %8.1f ms = %7.0f MP/s" % (t_pil * 1e3, H * W / 1e6 / t_pil))
dt, got_px = timed(lambda: _shim.jpeg_decode_gray(scan))
assert np.array_equal(got_px, want_px)
print("ogd_jpeg_decode_gray (file in, host pixels): %7.1f ms = %7.0f MP/s" % (dt * 1e3, H * W / 1e6 / dt))
buf = np.frombuffer(scan, np.uint8)
d_px = ctypes.c_void_p()
_shim.check(lib.ogd_dev_alloc(ctypes.byref(d_px), H * W))
dt, _ = timed(lambda: _shim.check(lib.ogd_jpeg_decode_gray_dev(buf.ctypes.data, buf.size, d_px, H, W, W, None)))
print("ogd_jpeg_decode_gray_dev (pixels stay on the device): %5.1f ms = %7.0f MP/s" % (dt * 1e3, H * W / 1e6 / dt))
_shim.check(lib.ogd_dev_free(d_px))
print("decoded pixels identical to PIL's")

# ---- the same scan in colour (JFIF YCbCr; the reference converts it to "L" on opening) ----
rgb = synth.colour_page(rng, (W, H))
for name, params in (("4:2:0", dict(subsampling=2)), ("4:4:4", dict(subsampling=0))):
	b = io.BytesIO()
	PIL.Image.fromarray(rgb).save(b, format="JPEG", quality=85, **params)
	cscan = b.getvalue()
	t0 = time.perf_counter()
	want_px = np.array(PIL.Image.open(io.BytesIO(cscan)).convert("L"))
	t_pil = time.perf_counter() - t0
	print("colour scan %dx%d %s as a JPEG file of %d bytes" % (W, H, name, len(cscan)))
	print("  PIL decode + convert('L'), 1 host thread:  %8.1f ms = %7.0f MP/s" % (t_pil * 1e3, H * W / 1e6 / t_pil))
	dt, got_px = timed(lambda: _shim.jpeg_decode_gray(cscan))
	assert np.array_equal(got_px, want_px)
	print("  ogd_jpeg_decode_gray (file in, host pixels): %7.1f ms = %7.0f MP/s" % (dt * 1e3, H * W / 1e6 / dt))
	cbuf = np.frombuffer(cscan, np.uint8)
	d_px = ctypes.c_void_p()
	_shim.check(lib.ogd_dev_alloc(ctypes.byref(d_px), H * W))
	dt, _ = timed(lambda: _shim.check(lib.ogd_jpeg_decode_gray_dev(cbuf.ctypes.data, cbuf.size, d_px, H, W, W, None)))
	print("  ogd_jpeg_decode_gray_dev (pixels stay on the device): %5.1f ms = %7.0f MP/s" % (dt * 1e3, H * W / 1e6 / dt))
	_shim.check(lib.ogd_dev_free(d_px))
print("colour scans: 'L' pixels identical to PIL's")

# ---- scans with restart intervals (DRI): one lane per interval ----
for name, img, params in (("gray, restart every MCU row", page, dict(restart_marker_rows=1)),
		("4:2:0 colour, restart every MCU row", rgb, dict(restart_marker_rows=1, quality=85, subsampling=2))):
	b = io.BytesIO()
	PIL.Image.fromarray(img).save(b, format="JPEG", **params)
	rscan = b.getvalue()
	t0 = time.perf_counter()
	want_px = np.array(PIL.Image.open(io.BytesIO(rscan)).convert("L"))
	t_pil = time.perf_counter() - t0
	print("scan %dx%d, %s: %d bytes" % (W, H, name, len(rscan)))
	print("  PIL decode, 1 host thread:                 %8.1f ms" % (t_pil * 1e3))
	dt, got_px = timed(lambda: _shim.jpeg_decode_gray(rscan))
	assert np.array_equal(got_px, want_px)
	print("  ogd_jpeg_decode_gray (file in, host pixels): %7.1f ms" % (dt * 1e3))
	rbuf = np.frombuffer(rscan, np.uint8)
	d_px = ctypes.c_void_p()
	_shim.check(lib.ogd_dev_alloc(ctypes.byref(d_px), H * W))
	dt, _ = timed(lambda: _shim.check(lib.ogd_jpeg_decode_gray_dev(rbuf.ctypes.data, rbuf.size, d_px, H, W, W, None)))
	print("  ogd_jpeg_decode_gray_dev (pixels stay on the device): %5.1f ms" % (dt * 1e3))
	_shim.check(lib.ogd_dev_free(d_px))
print("restart-interval scans: pixels identical to PIL's")
