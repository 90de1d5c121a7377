#!/usr/bin/env python3
"""K1a and K2 over a batch of resident pages in two orders: page by page (K1a, K2, K1a, K2 ... through ONE dense
grid buffer, as bench.py's step does) and in phases (K1a for G pages into G buffers, then K2 for those pages).
usage: phase_order.py [n_pages [G,G,...]]     (never bench.py's `value`)"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from origami_amd import _shim  # noqa: E402


def main():
	n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 64
	groups = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 8, 64]
	dev = torch.device("cuda:0")
	_shim.init(0)
	lib, check = _shim.load(), _shim.check
	wl = bench.build_workload(torch, _shim, dev, n_pages, 0)
	W, H = bench.PAGE
	Hd, Wd, h, w = wl["Hd"], wl["Wd"], wl["h"], wl["w"]
	stream = torch.cuda.current_stream().cuda_stream
	dense = list(wl["dense"])
	while len(dense) < max(groups):
		dense.append(torch.empty_like(dense[0]))
	print("%d pages, %d dense grid buffers of %.0f MB" % (n_pages, len(dense), dense[0].numel() * 4 / 1e6), flush=True)

	def k1a(p, buf):
		wp = wl["warps"][p % len(wl["warps"])]
		check(lib.ogd_grid_fill_dev(wp["nodes_d"].data_ptr(), h, w, bench.CELL, 1.0, 1.0, buf.data_ptr(), stream))

	def k2(p, buf):
		check(lib.ogd_remap_u8_dev(
			wl["pages"][p].data_ptr(), H, W, 1, W, buf.data_ptr(), Hd, Wd, wl["outs"][p].data_ptr(), Wd, 1, stream))

	def run(G):
		for g0 in range(0, n_pages, G):
			ps = range(g0, min(g0 + G, n_pages))
			for p in ps:
				k1a(p, dense[p - g0])
			for p in ps:
				k2(p, dense[p - g0])

	def timed(fn, reps=4):
		fn()
		torch.cuda.synchronize()
		best = []
		for _ in range(reps):
			t0 = time.perf_counter()
			fn()
			torch.cuda.synchronize()
			best.append(time.perf_counter() - t0)
		return best

	want = None
	for rnd in (1, 2):
		for G in groups:
			ts = timed(lambda: run(G))
			got = wl["outs"][n_pages - 1].clone()
			if want is None:
				want = got
			same = torch.equal(want, got)
			print("pass %d, groups of %2d pages: %s us per page (K1a + K2)%s" % (
				rnd, G, " ".join("%.1f" % (1e6 * t / n_pages) for t in ts), "" if same else "  OUTPUT DIFFERS"), flush=True)
		# the two kernels by themselves, back to back
		ts = timed(lambda: [k1a(p, dense[0]) for p in range(n_pages)])
		print("pass %d, K1a alone: %s us" % (rnd, " ".join("%.1f" % (1e6 * t / n_pages) for t in ts)), flush=True)
		ts = timed(lambda: [k2(p, dense[0]) for p in range(n_pages)])
		print("pass %d, K2 alone : %s us" % (rnd, " ".join("%.1f" % (1e6 * t / n_pages) for t in ts)), flush=True)


if __name__ == "__main__":
	main()
