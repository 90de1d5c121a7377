#!/usr/bin/env python3
"""PCIe-inclusive page rates of the PRODUCT's host-pixel paths on broadsheet pages (6000 x 8000 gray, cell 25):

  sync, pageable in    _shim.dewarp_u8 from an ordinary numpy array (what round 2's product did: 6.8 ms/page)
  sync, pinned in/out  the same entry point fed from origami_amd.pinned memory (what Page / Dewarper do now)
  pipeline             origami_amd.pipeline.PagePipeline over a ring of pinned pages, result pages in pinned memory
  pipeline, pageable   the same with ordinary input arrays (one host memcpy into pinned memory per page)
  pipeline + contours  with 400 contour rings per page through the page's transformer in the same job

usage: host_paths.py [n_pages [slots]]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, pinned, synth  # noqa: E402
from origami_amd.dewarp import Grid  # noqa: E402
from origami_amd.pipeline import PageJob, PagePipeline  # noqa: E402

n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 96
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 3
RING = 8
_shim.init(0)
rng = np.random.default_rng(0)
size = (6000, 8000)
base = synth.text_page(rng, size)
nodes = synth.Warp(size, 0.03, 0.02, 1, 2).node_grid(25)
grid = Grid(nodes, 25)
Hd, Wd = nodes.shape[0] * 25, nodes.shape[1] * 25
pages = [np.roll(base, (37 * k, 101 * k), axis=(0, 1)) for k in range(RING)]
pinned_pages = [pinned.as_pinned(p) for p in pages]
want = _shim.dewarp_u8(pages[1], nodes, 25).copy()


def report(name, dt, n):
	print("%-28s %6.2f ms/page  %8.0f MP/s" % (name, dt / n * 1e3, n * Hd * Wd / 1e6 / dt), flush=True)


def sync_loop(srcs, n):
	out = None
	for k in range(n):
		out = _shim.dewarp_u8(srcs[k % RING], nodes, 25)
	return out


for name, srcs in (("sync, pageable in", pages), ("sync, pinned in/out", pinned_pages)):
	sync_loop(srcs, 4)
	n = max(8, n_pages // 4)
	t0 = time.perf_counter()
	out = sync_loop(srcs, n)
	report(name, time.perf_counter() - t0, n)

rings = synth.polygon_rings(rng, size, 400, integer_scale=size[0] / 1280)
flat = np.concatenate(rings)
offsets = np.zeros(len(rings) + 1, dtype=np.int64)
np.cumsum([len(r) for r in rings], out=offsets[1:])
tf = grid.transformer

for name, srcs, with_rings in (
		("pipeline (pinned ring)", pinned_pages, False), ("pipeline, pageable in", pages, False),
		("pipeline + 400 contours", pinned_pages, True)):
	with PagePipeline(slots=slots) as pipe:
		def jobs(n):
			for k in range(n):
				yield PageJob(srcs[k % RING], nodes, 25, rings=(tf, flat, offsets) if with_rings else None, tag=k % RING)
		for r in pipe.map(jobs(max(2 * RING, 64))):  # warm-up: the copy engines take ~60 pages to reach their rate
			pass
		t0 = time.perf_counter()
		for r in pipe.map(jobs(n_pages)):
			last = r
		dt = time.perf_counter() - t0
	report(name, dt, n_pages)
	if last.tag == 1 or n_pages % RING == 2:
		pass
	check = _shim.dewarp_u8(pages[last.tag], nodes, 25)
	assert np.array_equal(last.page, check), "pipeline output differs"
	if with_rings:
		assert np.array_equal(last.rings, tf.transform_table(flat, offsets)), "pipeline contours differ"
print("outputs identical to ogd_dewarp_u8")
