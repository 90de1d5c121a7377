#!/usr/bin/env python3
"""K2 / fused kernel times for the other pixel formats of BASELINE.json configs[3]: RGB page and label map
(nearest, scaled grid) through an 8-px node grid, next to the gray page of configs[1].  Device-resident,
HIP events on the launch stream.  usage: k2_variants_bench.py [iters]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = sys.argv[2] if len(sys.argv) > 2 else ""  # substring of the case names to run
torch.cuda.set_device(0)
_shim.init(0)
lib, check = _shim.load(), _shim.check
dev = torch.device("cuda:0")
W, H = 6000, 8000
rng = np.random.default_rng(3)


def timed(fn):
	for _ in range(3):
		fn()
	torch.cuda.synchronize()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(iters):
		fn()
	e1.record()
	torch.cuda.synchronize()
	return e0.elapsed_time(e1) / iters


def case(name, src, cell, interp, sx=1.0, sy=1.0, amp=(0.03, 0.02)):
	if only and only not in name:
		return
	warp = synth.Warp(( W, H), amp[0], amp[1], 2, 1)
	nodes = warp.node_grid(cell)
	h, w = nodes.shape[:2]
	Hd, Wd = h * cell, w * cell
	C = 1 if src.ndim == 2 else src.shape[2]
	sh, sw = src.shape[:2]
	d_src = torch.from_numpy(np.ascontiguousarray(src)).to(dev)
	d_nodes = torch.from_numpy(nodes).to(dev)
	d_map = torch.empty((Hd, Wd, 2), dtype=torch.float32, device=dev)
	d_dst = torch.empty((Hd, Wd, C), dtype=torch.uint8, device=dev)
	st = torch.cuda.current_stream().cuda_stream
	check(lib.ogd_grid_fill_dev(d_nodes.data_ptr(), h, w, cell, sx, sy, d_map.data_ptr(), st))
	t_k2 = timed(lambda: check(lib.ogd_remap_u8_dev(
		d_src.data_ptr(), sh, sw, C, sw * C, d_map.data_ptr(), Hd, Wd, d_dst.data_ptr(), Wd * C, interp, st)))
	ref = d_dst.clone()
	t_f = timed(lambda: check(lib.ogd_dewarp_u8_dev(
		d_src.data_ptr(), sh, sw, C, sw * C, d_nodes.data_ptr(), h, w, cell, sx, sy, d_dst.data_ptr(), Wd * C, interp, st)))
	same = bool(torch.equal(ref, d_dst))
	alg = Hd * Wd * (8 + C) + sh * sw * C
	alg_f = Hd * Wd * C + sh * sw * C + h * w * 8
	print("%-44s K2 %.4f ms = %6.1f GB/s (%4.1f%% of 8 TB/s, %d B/px) | fused %.4f ms = %6.1f GB/s of its %.0f MB | %s" % (
		name, t_k2, alg / t_k2 / 1e6, alg / t_k2 / 8e7, 8 + 2 * C, t_f, alg_f / t_f / 1e6, alg_f / 1e6,
		"same bytes" if same else "MISMATCH"))
	sys.stdout.flush()


gray = synth.text_page(rng, (W, H))
rgb = synth.text_page(rng, (W, H), channels=3)
labels = synth.label_map(rng, (1280, 2400), 4, blocks=64)
case("gray 6000x8000, cell 25, linear", gray, 25, _shim.INTER_LINEAR)
case("gray 6000x8000, cell 8, linear", gray, 8, _shim.INTER_LINEAR)
case("RGB 6000x8000, cell 8, linear", rgb, 8, _shim.INTER_LINEAR)
case("RGBA 6000x8000, cell 8, linear", np.dstack([rgb, gray]), 8, _shim.INTER_LINEAR)
case("RGB 6000x8000, cell 25, linear", rgb, 25, _shim.INTER_LINEAR)
case("RGB 6000x8000, cell 25, linear, heavy warp 0.12 rad", rgb, 25, _shim.INTER_LINEAR, amp=(0.12, 0.12))
case("labels 1280x2400 -> page grid, cell 8, nearest", labels, 8, _shim.INTER_NEAREST, 1280 / W, 2400 / H)
case("gray 6000x8000, cell 25, nearest", gray, 25, _shim.INTER_NEAREST)
