#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points on one broadsheet page:
pageable numpy buffers in, numpy buffers out, hipMalloc / hipMemcpy / kernel / hipMemcpy per call."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, synth  # noqa: E402

_shim.init(0)
rng = np.random.default_rng(0)
size = (6000, 8000)
page = synth.text_page(rng, size)
nodes = synth.Warp(size, 0.03, 0.02, 1, 2).node_grid(25)
for name, fn in (
		("ogd_dewarp_u8 (fused K1a+K2)", lambda: _shim.dewarp_u8(page, nodes, 25)),
		("ogd_grid_fill + ogd_remap_u8", lambda: _shim.remap_u8(page, _shim.grid_fill(nodes, 25)))):
	out = fn()
	t0 = time.perf_counter()
	n = 3
	for _ in range(n):
		out = fn()
	dt = (time.perf_counter() - t0) / n
	print("%-32s %.1f ms/page  %.0f MP/s (host buffers, PCIe inclusive)" % (name, dt * 1e3, out.shape[0] * out.shape[1] / 1e6 / dt))
