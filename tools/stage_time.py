#!/usr/bin/env python3
"""Where the dewarp stage spends its time on one page (host march vs GPU work); DESIGN.md §5."""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import numpy as np
import PIL.Image

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, synth  # noqa: E402
from origami_amd.dewarp import Grid  # noqa: E402

_shim.init(0)
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2400, 3400)
n_h, n_v = (1500, 2000) if W > 3000 else (300, 500)
rng = np.random.default_rng(1)
warp = synth.Warp.random(rng, (W, H))
h, v = synth.flow_samples(rng, warp, n_h, n_v)


class Page:
	warped = PIL.Image.new("L", (W, H))


def run():
	t0 = time.perf_counter()
	grid = Grid.create(Page, h, v, grid_res=25)
	t1 = time.perf_counter()
	tf = grid.transformer
	t2 = time.perf_counter()
	rings = synth.polygon_rings(rng, (W, H), 400)
	tf.transform_rings(rings)
	t3 = time.perf_counter()
	print("page %dx%d: Grid.create %.2f s (grid %s), transformer build %.2f s, 400 rings %.4f s" % (
		W, H, t1 - t0, grid.points("sample").shape, t2 - t1, t3 - t2))


run()
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
