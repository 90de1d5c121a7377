#!/usr/bin/env python3
"""Throughput (ii) of SURVEY §8d: pinned host -> H2D -> fused dewarp (K1a∘K2) -> D2H, overlapped on three
streams with a ring of device slots.  PCIe-inclusive; never bench.py's `value`.
usage: pipeline_bench.py [n_pages slots ring]"""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, synth  # noqa: E402

W, H, CELL = 6000, 8000, 25
n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 64
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 3

torch.cuda.set_device(0)
_shim.init(0)
lib, check = _shim.load(), _shim.check
dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
warp = synth.Warp.random(rng, (W, H))
nodes = warp.node_grid(CELL)
h, w = nodes.shape[:2]
Hd, Wd = h * CELL, w * CELL
nodes_d = torch.from_numpy(nodes).to(dev)

ring = int(sys.argv[3]) if len(sys.argv) > 3 else 8  # distinct pinned host pages cycled over the jobs
base = synth.text_page(rng, (W, H))
host_in = torch.empty((ring, H, W), dtype=torch.uint8).pin_memory()
for k in range(ring):
	host_in[k] = torch.from_numpy(np.roll(base, (37 * k, 101 * k), axis=(0, 1)))
host_out = torch.empty((ring, Hd, Wd), dtype=torch.uint8).pin_memory()
d_in = torch.empty((slots, H, W), dtype=torch.uint8, device=dev)
d_out = torch.empty((slots, Hd, Wd), dtype=torch.uint8, device=dev)

s_in, s_k, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
if os.environ.get("PIPE_KERNELS_ON_DEFAULT_STREAM"):  # experiment: what bench.py's leg did until round 2
	s_k = torch.cuda.current_stream()
ev_in = [torch.cuda.Event() for _ in range(slots)]
ev_k = [torch.cuda.Event() for _ in range(slots)]
ev_out = [torch.cuda.Event() for _ in range(slots)]


import os
MODE = os.environ.get("PIPE_MODE", "")  # experiments: "nokernel", "noevents" (copies only, no dependencies), "deep" (all jobs queued at once)


def run(n):
	if MODE == "noevents":
		for p in range(n):
			sl = p % slots
			with torch.cuda.stream(s_in):
				d_in[sl].copy_(host_in[p % ring], non_blocking=True)
			with torch.cuda.stream(s_out):
				host_out[p % ring].copy_(d_out[sl], non_blocking=True)
		torch.cuda.synchronize()
		return
	for p in range(n):
		sl = p % slots
		if p >= slots and MODE != "deep":
			ev_out[sl].synchronize()  # the host takes delivery of the slot's previous page: at most `slots` jobs in flight
		with torch.cuda.stream(s_in):
			s_in.wait_event(ev_k[sl])  # the slot's previous kernel has read d_in[sl]
			d_in[sl].copy_(host_in[p % ring], non_blocking=True)
			ev_in[sl].record(s_in)
		s_k.wait_event(ev_in[sl])
		s_k.wait_event(ev_out[sl])  # the slot's previous download has read d_out[sl]
		if MODE != "nokernel":
			check(lib.ogd_dewarp_u8_dev(
				d_in[sl].data_ptr(), H, W, 1, W, nodes_d.data_ptr(), h, w, CELL, 1.0, 1.0,
				d_out[sl].data_ptr(), Wd, 1, s_k.cuda_stream))
		ev_k[sl].record(s_k)
		with torch.cuda.stream(s_out):
			s_out.wait_event(ev_k[sl])
			host_out[p % ring].copy_(d_out[sl], non_blocking=True)
			ev_out[sl].record(s_out)
	torch.cuda.synchronize()


run(slots * 2)
t0 = time.perf_counter()
run(n_pages)
dt = time.perf_counter() - t0
mp = n_pages * Hd * Wd / 1e6
print("ring %d:" % ring, end=" ")
print("pipeline (pinned host in/out, %d slots, fused dewarp): %d pages in %.1f ms = %.2f ms/page = %.0f MP/s; "
	"PCIe %.1f GB/s in + %.1f GB/s out" % (
		slots, n_pages, dt * 1e3, dt / n_pages * 1e3, mp / dt, n_pages * H * W / dt / 1e9, n_pages * Hd * Wd / dt / 1e9))

# check the last page against a direct call
if not MODE:
	want = _shim.dewarp_u8(host_in[(n_pages - 1) % ring].numpy(), nodes, CELL)
	assert np.array_equal(host_out[(n_pages - 1) % ring].numpy(), want)
	print("last page identical to ogd_dewarp_u8")
