#!/usr/bin/env python3
"""N1 measurement: all line images of one page, GPU (one launch) vs the oracle's per-line loop.
usage: lines_bench.py [W H n_lines]   (default: 6000 8000 3000 — a broadsheet page, BASELINE configs[1])"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, synth  # noqa: E402
from oracle import block as oblock, build_oracle, dewarp as odw  # noqa: E402

build_oracle.build()
_shim.init(0)
W, H, n = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (6000, 8000, 3000)
s = 25
rng = np.random.default_rng(3)
h, w = -(-H // s) + 2, -(-W // s) + 2
gy, gx = np.mgrid[0:h, 0:w].astype(np.float64)
nodes = np.stack([
	(gx - 0.5) * s + 9 * np.sin(gy / 40.0) + rng.normal(0, 0.3, size=(h, w)),
	(gy - 0.5) * s + 9 * np.cos(gx / 50.0) + rng.normal(0, 0.3, size=(h, w))], axis=-1).astype(np.float32)
page = synth.text_page(rng, (W, H))
Wd, Hd = w * s, h * s
geo = np.empty((n, 6))
for k in range(n):
	length = rng.uniform(0.05 * Wd, 0.3 * Wd)  # newspaper columns
	height = rng.uniform(24, 48)
	ang = rng.normal(0, 0.01)
	geo[k] = (rng.uniform(5, Wd - length - 5), rng.uniform(height + 5, Hd - 5), length * np.cos(ang), length * np.sin(ang), height * np.sin(ang), -height * np.cos(ang))
xsteps = np.maximum(2, np.ceil(np.hypot(geo[:, 2], geo[:, 3]))).astype(np.int32)
ysteps = np.full(n, 48, np.int32)
px = int((xsteps.astype(np.int64) * ysteps).sum())

t0 = time.perf_counter()
dp = _shim.DevicePage(page)
t1 = time.perf_counter()
dp.line_images(nodes, s, geo[:8], xsteps[:8], ysteps[:8])  # warm the block cache
reps = 5
t2 = time.perf_counter()
for _ in range(reps):
	images = dp.line_images(nodes, s, geo, xsteps, ysteps)
t3 = time.perf_counter()
gpu_s = (t3 - t2) / reps
print("page %dx%d, %d lines, %.1f Mpx of line images" % (W, H, n, px / 1e6))
print("upload page: %.1f ms; ogd_line_images_u8 (host in, host out): %.2f ms/page = %.0f Mpx/s" % (
	(t1 - t0) * 1e3, gpu_s * 1e3, px / gpu_s / 1e6))

# device-output form: the images stay in device memory (no 152 MB device -> host copy)
import ctypes
lib = _shim.load()
d_out = ctypes.c_void_p()
_shim.check(lib.ogd_dev_alloc(ctypes.byref(d_out), px))
dp.line_images_dev(nodes, s, geo, xsteps, ysteps, d_out.value)
t6 = time.perf_counter()
for _ in range(reps):
	dp.line_images_dev(nodes, s, geo, xsteps, ysteps, d_out.value)
t7 = time.perf_counter()
dev_s = (t7 - t6) / reps
print("ogd_line_images_u8_dev (host descriptors in, images stay on the device): %.2f ms/page = %.0f Mpx/s" % (
	dev_s * 1e3, px / dev_s / 1e6))
_shim.check(lib.ogd_dev_free(d_out))

grid = odw.Grid(nodes, s)
grid.points_full()
m = min(n, 40)
t4 = time.perf_counter()
bad = tot = 0
for k in range(m):
	v = geo[k]
	wg = oblock.warped_grid(grid, v[0:2], v[2:4], v[4:6], xsteps=int(xsteps[k]), ysteps=48)
	from oracle import cv as ocv
	want = ocv.remap_u8(page, wg, ocv.INTER_LINEAR)[::-1, :]
	d = np.abs(want.astype(np.int16) - images[k])
	assert d.max() <= 1
	bad += np.count_nonzero(d)
	tot += d.size
t5 = time.perf_counter()
cpu_px = int((xsteps[:m].astype(np.int64) * 48).sum())
cpu_s = t5 - t4
print("oracle (scipy RegularGridInterpolator + remap, 1 core), %d lines: %.1f ms/line, %.2f Mpx/s -> %.1f s/page; pixels differing: %d of %d" % (
	m, cpu_s / m * 1e3, cpu_px / cpu_s / 1e6, px / (cpu_px / cpu_s), bad, tot))
print("speedup vs 1 core: %.0fx" % ((px / (cpu_px / cpu_s)) / gpu_s))
