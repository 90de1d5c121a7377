#!/usr/bin/env python3
"""Do the march kernels of several worker PROCESSES overlap on one GPU?  Each of N processes builds the same
broadsheet node grid K times (Grid.create: two angle fields + ogd_march_shape + ogd_march_grid — a handful of
wavefronts busy for ~4 ms); if the GPU ran one process's kernels at a time, N processes would take N times as long.
usage: march_overlap.py [K]   (runs N = 1, 2, 4, 8)"""
import multiprocessing as mp
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def work(k, barrier, out):
	from origami_amd import _shim, synth
	from origami_amd.dewarp import Grid
	_shim.init(0)
	W, H = 6000, 8000
	rng = np.random.default_rng(1)
	warp = synth.Warp((W, H), 0.03, 0.025, 1, 2)
	sh, sv = synth.flow_samples(rng, warp, 1500, 2000)

	class _Page:
		@staticmethod
		def size(dewarped):
			return (W, H)

	Grid.create(_Page, sh, sv, grid_res=25)
	barrier.wait()
	t0 = time.perf_counter()
	for _ in range(k):
		Grid.create(_Page, sh, sv, grid_res=25)
	out.put(time.perf_counter() - t0)


if __name__ == "__main__":
	K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
	ctx = mp.get_context("spawn")
	for n in (1, 2, 4, 8):
		barrier, out = ctx.Barrier(n), ctx.Queue()
		procs = [ctx.Process(target=work, args=(K, barrier, out)) for _ in range(n)]
		for p in procs:
			p.start()
		times = [out.get() for _ in procs]
		for p in procs:
			p.join()
		print("%d processes x %d grids: slowest %.3f s = %.2f ms per grid and process, %.2f ms per grid overall" % (
			n, K, max(times), max(times) / K * 1e3, max(times) / K / n * 1e3), flush=True)
