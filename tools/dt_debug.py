#!/usr/bin/env python3
"""Dumps the point sets the dewarp stage triangulates for one synthetic broadsheet page, with what
ogd_delaunay_points says about each (ties / unsupported).  usage: dt_debug.py OUT.npz"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, lingrid, synth  # noqa: E402
from origami_amd.dewarp import Grid  # noqa: E402

seen = []
real = lingrid.InterpolatorFactory._triangulate


def spy(points, values):
	p = np.asarray(points, dtype=np.float64)
	try:
		_, _, ties = _shim.delaunay_points(p)
		note = "ties %d" % ties
	except _shim.OgdError as e:
		note = str(e)
	print(p.shape, note, flush=True)
	seen.append(p)
	return real(points, values)


lingrid.InterpolatorFactory._triangulate = staticmethod(spy)
_shim.init(0)
rng = np.random.default_rng(1)
size = (6000, 8000)
warp = synth.Warp(size, 0.03, 0.025, 1, 2)
h, v = synth.flow_samples(rng, warp, 1500, 2000)


class _Page:
	class warped:
		size = (6000, 8000)


grid = Grid.create(_Page, h, v, grid_res=25)
grid.transformer.transform_rings([np.array([[10.0, 10.0], [50.0, 10.0], [50.0, 80.0], [10.0, 10.0]])])
np.savez(sys.argv[1], *seen)
