#!/usr/bin/env python3
"""K2 writing its output pages packed one behind the other in one allocation (page i begins at i * Hd * Wd bytes, any
alignment) against pages in allocations of their own (256-byte aligned, what bench.py uses): what the alignment of the
output buys, per launch ("by how much is not recorded").  usage: k2_packed_pages.py [pages] [iters]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, synth  # noqa: E402

n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.cuda.set_device(0)
_shim.init(0)
lib, check = _shim.load(), _shim.check
dev = torch.device("cuda:0")
W, H, cell = 6000, 8000, 25
rng = np.random.default_rng(3)
nodes = synth.Warp((W, H), 0.03, 0.02, 2, 1).node_grid(cell)
h, w = nodes.shape[:2]
Hd, Wd = h * cell, w * cell
d_src = torch.from_numpy(synth.text_page(rng, (W, H))).to(dev)
d_nodes = torch.from_numpy(nodes).to(dev)
d_map = torch.empty((Hd, Wd, 2), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
check(lib.ogd_grid_fill_dev(d_nodes.data_ptr(), h, w, cell, 1.0, 1.0, d_map.data_ptr(), st))
page_bytes = Hd * Wd
packed = torch.empty(n_pages * page_bytes + 256, dtype=torch.uint8, device=dev)
own = [torch.empty(page_bytes, dtype=torch.uint8, device=dev) for _ in range(n_pages)]
alg = Hd * Wd * 9 + H * W


def run(pointers):
	def once():
		for p in pointers:
			check(lib.ogd_remap_u8_dev(d_src.data_ptr(), H, W, 1, W, d_map.data_ptr(), Hd, Wd, p, Wd, 1, st))
	for _ in range(2):
		once()
	torch.cuda.synchronize()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(iters):
		once()
	e1.record()
	torch.cuda.synchronize()
	return e0.elapsed_time(e1) / (iters * len(pointers))


for rep in range(2):
	t_own = run([t.data_ptr() for t in own])
	t_packed = run([packed.data_ptr() + k * page_bytes for k in range(n_pages)])
	t_odd = run([packed.data_ptr() + 37 + k * page_bytes for k in range(n_pages)])
	print("pass %d: pages in allocations of their own %.2f us = %.1f %% of 8 TB/s | packed (page starts at %s mod 64) %.2f us = %.1f %% | packed + 37 bytes %.2f us = %.1f %%" % (
		rep, 1e3 * t_own, alg / t_own / 8e7, sorted(set((k * page_bytes) % 64 for k in range(n_pages))), 1e3 * t_packed, alg / t_packed / 8e7,
		1e3 * t_odd, alg / t_odd / 8e7), flush=True)
first = torch.empty(page_bytes, dtype=torch.uint8, device=dev)
check(lib.ogd_remap_u8_dev(d_src.data_ptr(), H, W, 1, W, d_map.data_ptr(), Hd, Wd, first.data_ptr(), Wd, 1, st))
torch.cuda.synchronize()
print("same bytes wherever the page lies:", all(torch.equal(first, packed[37 + k * page_bytes:37 + (k + 1) * page_bytes]) for k in range(n_pages)))
