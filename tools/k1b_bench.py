#!/usr/bin/env python3
"""K1b (barycentric rasteriser, ogd_tri_rasterize_dev) at broadsheet scale: (a) an angle field of 2000 flow
samples on the (W+1) x (H+1) pixel lattice of a 6000 x 8000 page, float32 out (ref: lingrid.py:252-282);
(b) the transformer's two-component interpolant (157 k triangles over the node lattice) on the same lattice.
Device-resident, HIP events.  usage: k1b_bench.py [iters]"""
import ctypes
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, synth  # noqa: E402
from origami_amd.dewarp import Grid  # noqa: E402
from origami_amd.lingrid import lininterp  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.cuda.set_device(0)
_shim.init(0)
lib = _shim.load()
dev = torch.device("cuda", 0)
W, H = 6000, 8000
rng = np.random.default_rng(7)
warp = synth.Warp.random(rng, (W, H))
h_samples, _ = synth.flow_samples(rng, warp, 2000, 2000)
field = lininterp(h_samples.points, h_samples.values, (0, 0, W, H))
tf = Grid(warp.node_grid(25), 25).transformer._interp
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def bench(name, interp, nv, out_f64):
	inner, extra = interp.triangulations
	nx, ny = W + 1, H + 1
	out = torch.empty((ny, nx, nv), dtype=torch.float64 if out_f64 else torch.float32, device=dev)
	bounds = np.array(interp.bounds, dtype=np.float64)

	def launch():
		_shim.check(lib.ogd_tri_rasterize_dev(
			inner.handle if inner is not None else None, extra.handle, bounds.ctypes.data, 0.0, 0.0, 1.0, nx, ny,
			out.data_ptr(), nv, nx * nv, int(out_f64), stream))

	for _ in range(2):
		launch()
	torch.cuda.synchronize()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(iters):
		launch()
	e1.record()
	torch.cuda.synchronize()
	ms = e0.elapsed_time(e1) / iters
	px = nx * ny
	by = px * nv * (8 if out_f64 else 4)
	assert not bool(torch.isnan(out).any())
	print("K1b %-46s %d x %d lattice: %.3f ms  %.1f Gpx/s  %.1f GB/s written (%.1f%% of 8 TB/s)" % (
		name, nx, ny, ms, px / ms / 1e6, by / ms / 1e6, by / ms / 8e7))


bench("angle field, 2000 samples, f32", field, 1, False)
bench("transformer (157 k triangles), 2 x f32", tf, 2, False)
bench("transformer (157 k triangles), 2 x f64", tf, 2, True)
