#!/usr/bin/env python3
"""K3 (contour point warp) at BASELINE configs[4] scale: 5000 rings / ~270 k points through the forward
transformer of a heavily warped broadsheet grid; device-resident, HIP events."""
import ctypes
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, synth  # noqa: E402
from origami_amd.dewarp import Grid  # noqa: E402

_shim.init(0)
lib = _shim.load()
dev = torch.device("cuda", 0)
PAGE = (6000, 8000)
rng = np.random.default_rng(5000)
for heavy, n_rings in ((False, 400), (True, 5000)):
	warp = synth.Warp.random(rng, PAGE, heavy=heavy)
	grid = Grid(warp.node_grid(25), 25)
	tf = grid.transformer
	inner, extra = tf._interp.triangulations
	bounds = np.array(tf._interp.bounds, dtype=np.float64)
	rings = synth.polygon_rings(rng, PAGE, n_rings, integer_scale=6000 / 1280)
	off = np.concatenate([[0], np.cumsum([len(r) for r in rings])]).astype(np.int64)
	q = torch.from_numpy(np.concatenate(rings)).to(dev)
	o = torch.from_numpy(off).to(dev)
	out = torch.empty((len(q), 2), dtype=torch.float64, device=dev)
	stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

	# the same points with rings longer than 64 vertices cut into 64-vertex pieces (one wavefront each)
	cut = np.unique(np.concatenate([np.arange(a, b, 64) for a, b in zip(off[:-1], off[1:])] + [off[-1:]]))
	o64 = torch.from_numpy(cut.astype(np.int64)).to(dev)

	def launch_rings():
		_shim.check(lib.ogd_tri_eval_ragged_dev(
			inner.handle, extra.handle, bounds.ctypes.data, q.data_ptr(), o.data_ptr(), n_rings, out.data_ptr(), stream))

	def launch_pieces():
		_shim.check(lib.ogd_tri_eval_ragged_dev(
			inner.handle, extra.handle, bounds.ctypes.data, q.data_ptr(), o64.data_ptr(), len(cut) - 1, out.data_ptr(), stream))

	def launch_flat():
		_shim.check(lib.ogd_tri_eval_dev(
			inner.handle, extra.handle, bounds.ctypes.data, q.data_ptr(), len(q), out.data_ptr(), stream))

	for what, launch in (("wavefront per ring", launch_rings), ("wavefront per <= 64-vertex piece", launch_pieces), ("flat, lane per point", launch_flat)):
		for _ in range(3):
			launch()
		torch.cuda.synchronize()
		e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		e0.record()
		for _ in range(20):
			launch()
		e1.record()
		torch.cuda.synchronize()
		ms = e0.elapsed_time(e1) / 20
		n = len(q)
		print("K3 %s, %s: %d rings (longest %d), %d points: %.1f us  %.2f Gpoints/s  (%.1f GB/s of 32 B/point)" % (
			"heavy" if heavy else "normal", what, n_rings, int(np.diff(off).max()), n, ms * 1e3, n / ms / 1e6, n * 32 / ms / 1e6))
