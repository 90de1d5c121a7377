#!/usr/bin/env python3
"""4:2:0 colour broadsheet scan: decode time on the device and settle rounds (experiment)."""
import io, sys, time, ctypes
from pathlib import Path
import numpy as np
import PIL.Image
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, synth
_shim.init(0)
lib = _shim.load()
W, H = 6000, 8000
rng = np.random.default_rng(11)
rgb = synth.colour_page(rng, (W, H))
for name, params in (("4:2:0 q90", dict(subsampling=2, quality=90)), ("4:2:0 q75", dict(subsampling=2, quality=75)), ("4:4:4 q90", dict(subsampling=0, quality=90))):
	f = io.BytesIO()
	PIL.Image.fromarray(rgb).save(f, format="jpeg", **params)
	data = f.getvalue()
	want = np.array(PIL.Image.open(io.BytesIO(data)).convert("L"))
	d_out = ctypes.c_void_p()
	_shim.check(lib.ogd_dev_alloc(ctypes.byref(d_out), H * W))
	buf = np.frombuffer(data, np.uint8)
	def call():
		_shim.check(lib.ogd_jpeg_decode_gray_dev(buf.ctypes.data, len(data), d_out, H, W, W, None))
		_shim.check(lib.ogd_stream_sync(None))
	call()
	got = np.empty((H, W), np.uint8)
	_shim.check(lib.ogd_memcpy_d2h(got.ctypes.data, d_out, H * W, None))
	_shim.check(lib.ogd_stream_sync(None))
	ok = np.array_equal(got, want)
	ts = []
	for _ in range(8):
		t = time.perf_counter(); call(); ts.append(time.perf_counter() - t)
	print("%s (%d bytes): %.2f ms (min %.2f)  %s" % (name, len(data), 1e3 * np.median(ts), 1e3 * min(ts), "same pixels" if ok else "PIXELS DIFFER"), flush=True)
	_shim.check(lib.ogd_dev_free(d_out))
