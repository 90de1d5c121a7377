#!/usr/bin/env python3
"""PCIe copy rates with pinned host memory: one copy per direction against the same bytes split over two streams,
alone and with the other direction running (what bounds the `pipeline` leg of bench.py)."""
import time
import torch

torch.cuda.set_device(0)
dev = torch.device("cuda:0")
N = 48_000_000
h_in = torch.empty(N, dtype=torch.uint8).pin_memory()
h_out = torch.empty(N, dtype=torch.uint8).pin_memory()
d_a = torch.empty(N, dtype=torch.uint8, device=dev)
d_b = torch.empty(N, dtype=torch.uint8, device=dev)
streams = [torch.cuda.Stream() for _ in range(4)]


def run(label, up_parts, down_parts, reps=20):
	def once():
		k = 0
		for parts, (dst, src) in ((up_parts, (d_a, h_in)), (down_parts, (h_out, d_b))):
			if not parts:
				continue
			step = N // parts
			for p in range(parts):
				with torch.cuda.stream(streams[k % 4]):
					dst[p * step:(p + 1) * step].copy_(src[p * step:(p + 1) * step], non_blocking=True)
				k += 1
	once()
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(reps):
		once()
	torch.cuda.synchronize()
	dt = (time.perf_counter() - t0) / reps
	print("%-44s %.2f ms: up %5.1f GB/s, down %5.1f GB/s" % (
		label, dt * 1e3, (N / dt / 1e9) if up_parts else 0, (N / dt / 1e9) if down_parts else 0), flush=True)


run("H2D, one copy", 1, 0)
run("H2D, two halves on two streams", 2, 0)
run("D2H, one copy", 0, 1)
run("D2H, two halves on two streams", 0, 2)
run("both directions, one copy each", 1, 1)
run("both directions, two halves each (4 streams)", 2, 2)

# the same through a ring of distinct pinned pages (what a page queue looks like: every copy reads other host memory)
RING = 32
ring_in = torch.empty((RING, N), dtype=torch.uint8, pin_memory=True)
ring_in.fill_(7)
ring_out = torch.empty((4, N), dtype=torch.uint8, pin_memory=True)


def run_ring(label, up, down, reps=64):
	def once(j):
		if up:
			with torch.cuda.stream(streams[0]):
				d_a.copy_(ring_in[j % RING], non_blocking=True)
		if down:
			with torch.cuda.stream(streams[1]):
				ring_out[j % 4].copy_(d_b, non_blocking=True)
	for j in range(4):
		once(j)
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for j in range(reps):
		once(j)
	torch.cuda.synchronize()
	dt = (time.perf_counter() - t0) / reps
	print("%-44s %.2f ms: up %5.1f GB/s, down %5.1f GB/s" % (
		label, dt * 1e3, (N / dt / 1e9) if up else 0, (N / dt / 1e9) if down else 0), flush=True)


run_ring("ring of 32 pages: H2D", True, False)
run_ring("ring of 32 pages: D2H into 4 buffers", False, True)
run_ring("ring of 32 pages: both directions", True, True)

# which pairs of streams copy in both directions at once?  (a process has GPU_MAX_HW_QUEUES hardware queues)
more = streams + [torch.cuda.Stream() for _ in range(4)]


def run_pair(a, b, reps=32):
	def once(j):
		with torch.cuda.stream(more[a]):
			d_a.copy_(ring_in[j % RING], non_blocking=True)
		with torch.cuda.stream(more[b]):
			ring_out[j % 4].copy_(d_b, non_blocking=True)
	for j in range(4):
		once(j)
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for j in range(reps):
		once(j)
	torch.cuda.synchronize()
	dt = (time.perf_counter() - t0) / reps
	print("up on stream %d, down on stream %d: %.2f ms = %5.1f GB/s each way" % (a, b, dt * 1e3, N / dt / 1e9), flush=True)


for a, b in ((0, 1), (0, 2), (0, 3), (0, 4), (1, 2), (1, 3), (2, 3), (0, 5), (4, 5), (0, 0)):
	run_pair(a, b)

# the same pairs once libogdewarp is initialised in the process and has run a kernel on a stream of its caller
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from origami_amd import _shim
_shim.init(0)
print("libogdewarp initialised")
for a, b in ((0, 1), (0, 2)):
	run_pair(a, b)
nodes = np.zeros((12, 9, 2), np.float32)
nodes_d = torch.from_numpy(nodes).to(dev)
page_d = torch.zeros((200, 150), dtype=torch.uint8, device=dev)
out_d = torch.zeros((300, 225), dtype=torch.uint8, device=dev)
_shim.check(_shim.load().ogd_dewarp_u8_dev(page_d.data_ptr(), 200, 150, 1, 150, nodes_d.data_ptr(), 12, 9, 25, 1.0, 1.0,
	out_d.data_ptr(), 225, 1, more[1].cuda_stream))
torch.cuda.synchronize()
print("one fused dewarp launched on stream 1")
for a, b in ((0, 1), (0, 2), (2, 3)):
	run_pair(a, b)
# 2-D views as the pipeline tools use them
h2 = torch.empty((8, 8000, 6000), dtype=torch.uint8).pin_memory()
o2 = torch.empty((8, 8075, 6075), dtype=torch.uint8).pin_memory()
di = torch.empty((3, 8000, 6000), dtype=torch.uint8, device=dev)
do = torch.empty((3, 8075, 6075), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for j in range(32):
	with torch.cuda.stream(more[0]):
		di[j % 3].copy_(h2[j % 8], non_blocking=True)
	with torch.cuda.stream(more[2]):
		o2[j % 8].copy_(do[j % 3], non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 32
print("page-shaped tensors, streams 0 and 2: %.2f ms per page" % (dt * 1e3))

for reps in (16, 32, 48, 64, 96, 128):
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for j in range(reps):
		with torch.cuda.stream(more[0]):
			di[j % 3].copy_(h2[j % 8], non_blocking=True)
		with torch.cuda.stream(more[2]):
			o2[j % 8].copy_(do[j % 3], non_blocking=True)
	torch.cuda.synchronize()
	dt = (time.perf_counter() - t0) / reps
	print("page-shaped tensors, %3d pages queued without a sync: %.2f ms per page" % (reps, dt * 1e3), flush=True)
