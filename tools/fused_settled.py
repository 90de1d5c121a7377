#!/usr/bin/env python3
"""The fused kernel (ogd_dewarp_u8_dev) on a device whose clocks have settled: N launches between one pair of HIP
events after a busy warm-up, against the same N launches started on a device that has been idle for a second.
usage: fused_settled.py [launches [pages]]   (pages: that many different source pages in turn, default 1)"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
n_pages = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # > 5: the sources no longer fit the 256 MB Infinity Cache
dev = torch.device("cuda:0")
_shim.init(0)
lib, check = _shim.load(), _shim.check
W, H, CELL = 6000, 8000, 25
rng = np.random.default_rng(3)
page0 = torch.from_numpy(synth.text_page(rng, (W, H))).to(dev)
pages = [page0] + [torch.roll(page0, shifts=(37 * k, 101 * k), dims=(0, 1)) for k in range(1, n_pages)]
count = [0]
nodes = synth.Warp.random(rng, (W, H)).node_grid(CELL)
h, w = nodes.shape[:2]
Hd, Wd = h * CELL, w * CELL
nodes_d = torch.from_numpy(nodes).to(dev)
out = torch.empty((Hd, Wd), dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream


def launch():
	page = pages[count[0] % n_pages]
	count[0] += 1
	check(lib.ogd_dewarp_u8_dev(page.data_ptr(), H, W, 1, W, nodes_d.data_ptr(), h, w, CELL, 1.0, 1.0, out.data_ptr(), Wd, 1, stream))


def timed(k):
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(k):
		launch()
	e1.record()
	torch.cuda.synchronize()
	return 1e3 * e0.elapsed_time(e1) / k


launch()
torch.cuda.synchronize()
for rnd in range(3):
	time.sleep(1.0)
	cold20 = timed(20)
	time.sleep(1.0)
	cold = timed(n)
	warm = timed(n)
	warm2 = timed(n)
	print("pass %d: after 1 s idle, 20 launches %.1f us each; after 1 s idle, %d launches %.1f us each; then %d more %.1f, and %d more %.1f" % (
		rnd, cold20, n, cold, n, warm, n, warm2), flush=True)
