#!/usr/bin/env python3
"""N4 decoder on periodic streams: blank margins, flat images, checkerboards — the cases in which
chunk walks do not re-synchronise by themselves.  Prints PIL's and the device decoder's time per case; run with
OGD_JPEG_DEBUG=1 to see the round counts.  Usage: jpeg_hard_cases.py [W H [part of a case name]]"""
import io
import sys
import time
from pathlib import Path

import numpy as np
import PIL.Image

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, synth  # noqa: E402

_shim.init(0)
rng = np.random.default_rng(5)
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3000, 4000)
print("%d x %d" % (W, H))


ONLY = sys.argv[3] if len(sys.argv) > 3 else ""


def case(name, img, **params):
	if ONLY not in name:
		return
	b = io.BytesIO()
	PIL.Image.fromarray(img).save(b, format="JPEG", **params)
	data = b.getvalue()
	t0 = time.perf_counter()
	want = np.array(PIL.Image.open(io.BytesIO(data)).convert("L"))
	t_pil = time.perf_counter() - t0
	got = _shim.jpeg_decode_gray(data)
	if got is None:
		t0 = time.perf_counter()
		_shim.jpeg_decode_gray(data)
		dt = time.perf_counter() - t0
		print("%-44s %8d bytes  PIL %6.1f ms  device: handed back to PIL after %6.1f ms" % (name, len(data), t_pil * 1e3, dt * 1e3), flush=True)
		return
	assert np.array_equal(got, want), name
	t0 = time.perf_counter()
	for _ in range(3):
		_shim.jpeg_decode_gray(data)
	dt = (time.perf_counter() - t0) / 3
	print("%-44s %8d bytes  PIL %6.1f ms  device %6.1f ms (host pixels out)" % (name, len(data), t_pil * 1e3, dt * 1e3), flush=True)


page = synth.text_page(rng, (W, H))
margins = np.full((H, W), 255, np.uint8)
margins[H // 8:H - H // 8, W // 8:W - W // 8] = page[H // 8:H - H // 8, W // 8:W - W // 8]
yy, xx = np.mgrid[0:H, 0:W]
case("text page", page)
case("text page with 25 % blank margins", margins)
case("flat white", np.full((H, W), 255, np.uint8))
case("flat gray", np.full((H, W), 128, np.uint8))
case("checkerboard 8 px, quality 100", (((yy // 8 + xx // 8) & 1) * 255).astype(np.uint8), quality=100)
case("checkerboard 1 px, quality 100", (((yy + xx) & 1) * 255).astype(np.uint8), quality=100)
rgb = synth.colour_page(rng, (W, H))
case("colour page 4:2:0", rgb, quality=85)
cm = np.full((H, W, 3), 255, np.uint8)
cm[H // 8:H - H // 8, W // 8:W - W // 8] = rgb[H // 8:H - H // 8, W // 8:W - W // 8]
case("colour page 4:2:0 with blank margins", cm, quality=85)
case("flat colour 4:2:0", np.broadcast_to(np.array([200, 100, 50], np.uint8), (H, W, 3)).copy())
case("flat colour 4:2:0, restart per MCU row", np.broadcast_to(np.array([200, 100, 50], np.uint8), (H, W, 3)).copy(), restart_marker_rows=1)
