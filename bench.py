#!/usr/bin/env python3
"""Benchmark of the dewarp hot path on MI355X (BASELINE.json metric: dewarped megapixels/s).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the GPU path over one batch of synthetic pages resident in HBM
(BASELINE.json configs[1]: 64 synthetic 6000x8000 grayscale broadsheet pages): per page
  K1a  ogd_grid_fill_dev        node grid -> dense sampling grid (8075 x 6075 x 2 f32)
  K2   ogd_remap_u8_dev         page -> dewarped page through that grid (the roofline kernel)
  K3   ogd_tri_eval_ragged_dev  every contour ring through the forward transformer
All launches go through the C ABI (ctypes): K1a / K2 on torch's current stream, K3 (which depends on
neither) on a second stream; torch only owns device memory, streams, events and (N > 1) the barrier.  Pages are independent: with N > 1 every rank dewarps its own
batch (weak scaling, no collective on the data path).

The host-side grid construction (marching the node grid, Qhull) is set-up, as it is in the reference,
where the dewarp stage writes dewarp.zip once and every later stage only re-runs K1a + K2.

`--gpus N` (N > 1) without a launcher starts N ranks itself (`python -m torch.distributed.run ...` as a child
process, before this process touches a GPU) — the plain command the driver's contract names always measures N
ranks.  `value` is the device-resident rate at every N (inputs in HBM when the timed region starts); the
PCIe-inclusive rate of BASELINE.json configs[2] (page jobs from a pinned host ring -> H2D -> fused dewarp + K3 ->
D2H on three streams per GPU) is measured after the timed steps under the same barrier /
max-over-ranks contract and reported beside it as `pipeline` — it shows the host-side limiter (PCIe, DRAM) that
the resident loop cannot.
"""

import argparse
import ctypes
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PAGE = (6000, 8000)  # (W, H)
CELL = 25
N_WARPS = 8  # distinct warps cycled over the pages of a batch (Qhull set-up is ~2 s each)
N_POLYGONS = 400
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
K2_KERNEL = "ogd::remap_u8_wave_kernel<1, 1, 16, 2, 4096, ogd::MapGlobal>"
TRAFFIC_JSON = ROOT / "profiles" / "r02" / "k2_traffic.json"  # rocprofv3 --pmc passes (tools/pmc.sh)


def measured_traffic(k2_bytes):
	"""HBM bytes per K2 launch from the committed PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs,
	gfx950 correction applied there) — only if they were taken on this same launch shape."""
	try:
		t = json.loads(TRAFFIC_JSON.read_text())
	except (OSError, ValueError):
		return None
	if t.get("kernel") != K2_KERNEL or t.get("algorithmic_bytes_per_launch") != k2_bytes:
		return None
	return t["hbm_bytes_per_launch"]


def parse_args():
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=1)
	ap.add_argument("--steps", type=int, default=3)
	ap.add_argument("--warmup", type=int, default=1)
	ap.add_argument("--pages", type=int, default=64, help="pages per batch and GPU")
	ap.add_argument("--no-cpu-baseline", action="store_true")
	ap.add_argument("--no-pipeline", action="store_true", help="skip the host-ring pipeline leg (configs[2])")
	ap.add_argument("--pipeline-jobs", type=int, default=0, help="page jobs per GPU in the pipeline leg (0 = 2 x pages)")
	ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(16, cores)")
	ap.add_argument("--harness-selftest", action="store_true",
		help="CPU / gloo run of the N-rank timing harness with a sleeping step (tests/test_multi.py)")
	return ap.parse_args()


def build_workload(torch, shim, dev, n_pages, rank):
	from origami_amd import synth
	from origami_amd.dewarp import Grid

	W, H = PAGE
	rng = np.random.default_rng(2000 + rank)
	base = synth.text_page(rng, PAGE)
	base_d = torch.from_numpy(base).to(dev)
	pages = torch.empty((n_pages, H, W), dtype=torch.uint8, device=dev)
	for p in range(n_pages):
		# distinct page content per job: the base page rolled by a page-specific offset
		pages[p] = torch.roll(base_d, shifts=(37 * p % H, 101 * p % W), dims=(0, 1))

	warps = []
	for k in range(min(N_WARPS, n_pages)):
		wrng = np.random.default_rng(2000 * 1000 + k)
		warp = synth.Warp.random(wrng, PAGE)
		nodes = warp.node_grid(CELL)
		grid = Grid(nodes, CELL)
		tf = grid.transformer  # host: Qhull over all nodes + extrapolation set; device: records
		inner, extra = tf._interp.triangulations
		bounds = np.array(tf._interp.bounds, dtype=np.float64)
		warps.append(dict(
			nodes=nodes, nodes_d=torch.from_numpy(nodes).to(dev), tf=tf, inner=inner, extra=extra,
			bounds=bounds))

	polys = []
	for p in range(n_pages):
		prng = np.random.default_rng(2000 * 2000 + p)
		rings = synth.polygon_rings(prng, PAGE, N_POLYGONS, integer_scale=W / 1280)
		ring_off = np.zeros(len(rings) + 1, dtype=np.int64)
		np.cumsum([len(r) for r in rings], out=ring_off[1:])
		# one wavefront per ring piece of <= 64 vertices, as the host entry point does (include/ogdewarp.h)
		offsets = np.unique(np.concatenate(
			[np.arange(a, b, 64) for a, b in zip(ring_off[:-1], ring_off[1:])] + [ring_off[-1:]])).astype(np.int64)
		flat = np.concatenate(rings)
		polys.append(dict(
			rings=rings, n=len(flat), nrings=len(offsets) - 1,
			q=torch.from_numpy(flat).to(dev), off=torch.from_numpy(offsets).to(dev),
			out=torch.empty((len(flat), 2), dtype=torch.float64, device=dev)))

	h, w = warps[0]["nodes"].shape[:2]
	Hd, Wd = h * CELL, w * CELL
	dense = torch.empty((Hd, Wd, 2), dtype=torch.float32, device=dev)
	# one allocation per dewarped page (a page of 6075 x 8075 bytes is an odd number of bytes long: packed pages would
	# start at every alignment; K2 aligns its stores to the output whatever that is, its map loads follow when it can)
	outs = [torch.empty((Hd, Wd), dtype=torch.uint8, device=dev) for _ in range(n_pages)]
	return dict(base=base, pages=pages, warps=warps, polys=polys, dense=dense, outs=outs, Hd=Hd, Wd=Wd, h=h, w=w)


def run_step(torch, lib, check, wl, stream, events=None, side_stream=None):
	W, H = PAGE
	Hd, Wd, h, w = wl["Hd"], wl["Wd"], wl["h"], wl["w"]
	n_pages = wl["pages"].shape[0]
	dense_p = wl["dense"].data_ptr()
	for p in range(n_pages):
		wp = wl["warps"][p % len(wl["warps"])]
		poly = wl["polys"][p]
		check(lib.ogd_grid_fill_dev(wp["nodes_d"].data_ptr(), h, w, CELL, 1.0, 1.0, dense_p, stream))
		if events is not None:
			e0 = torch.cuda.Event(enable_timing=True)
			e1 = torch.cuda.Event(enable_timing=True)
			e0.record()
		check(lib.ogd_remap_u8_dev(
			wl["pages"][p].data_ptr(), H, W, 1, W, dense_p, Hd, Wd, wl["outs"][p].data_ptr(), Wd, 1, stream))
		if events is not None:
			e1.record()
			events.append((e0, e1))
		# K3 depends on nothing K1a / K2 produce: it runs on its own stream and fills the gaps
		check(lib.ogd_tri_eval_ragged_dev(
			wp["inner"].handle, wp["extra"].handle, wp["bounds"].ctypes.data, poly["q"].data_ptr(),
			poly["off"].data_ptr(), poly["nrings"], poly["out"].data_ptr(), side_stream or stream))


def fused_pass(torch, lib, check, wl, stream, side_stream, repeats=2):
	"""Secondary figure, outside the timed steps: the same batch through the fused kernel the product's
	Dewarper uses (ogd_dewarp_u8_dev: the dense map never exists) + K3.  Returns (MP/s, ms per page)."""
	W, H = PAGE
	Hd, Wd, h, w = wl["Hd"], wl["Wd"], wl["h"], wl["w"]
	n_pages = wl["pages"].shape[0]
	# parity of the fused path against the unfused outputs of the timed steps (page 0)
	want = wl["outs"][0].clone()

	def one_pass():
		for p in range(n_pages):
			wp = wl["warps"][p % len(wl["warps"])]
			poly = wl["polys"][p]
			check(lib.ogd_dewarp_u8_dev(
				wl["pages"][p].data_ptr(), H, W, 1, W, wp["nodes_d"].data_ptr(), h, w, CELL, 1.0, 1.0,
				wl["outs"][p].data_ptr(), Wd, 1, stream))
			check(lib.ogd_tri_eval_ragged_dev(
				wp["inner"].handle, wp["extra"].handle, wp["bounds"].ctypes.data, poly["q"].data_ptr(),
				poly["off"].data_ptr(), poly["nrings"], poly["out"].data_ptr(), side_stream))

	one_pass()
	torch.cuda.synchronize()
	if not torch.equal(wl["outs"][0], want):
		raise SystemExit("bench: fused path differs from the unfused path")
	t0 = time.perf_counter()
	for _ in range(repeats):
		one_pass()
	torch.cuda.synchronize()
	dt = (time.perf_counter() - t0) / repeats
	return n_pages * Hd * Wd / 1e6 / dt, dt / n_pages * 1e3


def verify(torch, wl):
	"""Spot parity of one page of the batch against the oracle on a crop (whole pages are covered by
	tests/); keeps a broken kernel from producing a bench line."""
	from oracle import cv as ocv
	nodes = wl["warps"][0]["nodes"][:20, :20]
	page = wl["pages"][0].cpu().numpy()
	# dense pixels < 19 * CELL only use node rows / columns 0..19, i.e. no clamped edge of the crop
	n = 19 * CELL
	want = ocv.remap_u8(page, ocv.dense_grid(nodes, CELL))[:n, :n]
	got = wl["outs"][0][:n, :n].cpu().numpy()
	if not np.array_equal(got, want):
		raise SystemExit("bench: GPU output differs from the oracle on the verification crop")


def cpu_baseline(wl, threads, budget_s=12.0):
	"""The CPU restatement of the same path (oracle: C/OpenMP resize + remap, scipy interpolants)
	timed on a bounded sample of the same batch: pages 0, 1, ... (dense grid + remap + the page's 400
	contour rings each) until about `budget_s` seconds of CPU work are done."""
	from oracle import cv as ocv
	from oracle import dewarp as odw
	grids = {}
	spent = t_pix = t_pts = 0.0
	mp = 0.0
	n = 0
	while spent < budget_s and n < wl["pages"].shape[0]:
		k = n % len(wl["warps"])
		if k not in grids:  # set-up (Qhull), not timed — as on the GPU side
			grids[k] = odw.Grid(wl["warps"][k]["nodes"], CELL)
			grids[k].transformer
		grid = grids[k]
		page = wl["pages"][n].cpu().numpy()
		t0 = time.perf_counter()
		grid._full = None
		out = ocv.remap_u8(page, grid.points_full(threads), ocv.INTER_LINEAR, threads)
		t1 = time.perf_counter()
		odw.transform_rings(grid.transformer, wl["polys"][n]["rings"])
		t2 = time.perf_counter()
		t_pix += t1 - t0
		t_pts += t2 - t1
		spent += t2 - t0
		mp += out.shape[0] * out.shape[1] / 1e6
		n += 1
	return dict(
		value=round(mp / spent, 2), unit="MP/s", cores=threads, kind="port",
		sample="%d pages 6000x8000 gray of the batch in %.1f s: dense grid + remap %.2f s (C/OpenMP oracle, %d threads) + %d contour rings/page %.2f s (scipy, 1 thread)" % (
			n, spent, t_pix, threads, len(wl["polys"][0]["rings"]), t_pts))


def pipeline_leg(torch, wl, n_jobs, barrier, streams, slots=3):
	"""BASELINE.json configs[2] per GPU through the PRODUCT's pipeline class
	(origami_amd/pipeline.py:PagePipeline): `n_jobs` page jobs drawn by cycling the batch's pages from a PINNED HOST
	ring -> H2D -> fused dewarp (ogd_dewarp_u8_dev) + the page's contour rings (K3) -> D2H of the dewarped page and
	the warped contours, on three streams (upload / kernels / download) with `slots` device slots and at most
	`slots` jobs in flight.  The last job's page is checked against the resident path's output.  Returns (elapsed
	seconds of this rank, bytes up, bytes down).
	streams = (upload, kernels, download) raw hipStream_t values: the process must not hold more streams than the
	GPU_MAX_HW_QUEUES = 4 hardware queues it gets — with a fifth stream two of them share a queue, upload and
	download take turns and the leg runs at a third of the PCIe rate (measured: 4.2 instead of 1.6 ms/page)."""
	from origami_amd import pinned
	from origami_amd.pipeline import DeviceArray, PageJob, PagePipeline
	W, H = PAGE
	Hd, Wd = wl["Hd"], wl["Wd"]
	n_pages = wl["pages"].shape[0]
	host_in = pinned.empty((n_pages, H, W), np.uint8)  # the pinned host ring of page scans
	torch.from_numpy(host_in).copy_(wl["pages"])
	jobs = []
	for p in range(n_pages):
		wp, poly = wl["warps"][p % len(wl["warps"])], wl["polys"][p]
		rings = (wp["tf"], DeviceArray(poly["q"].data_ptr(), (poly["n"], 2)),
			DeviceArray(poly["off"].data_ptr(), (poly["nrings"] + 1,)))
		jobs.append(PageJob(
			host_in[p], DeviceArray(wp["nodes_d"].data_ptr(), wp["nodes"].shape), CELL, rings=rings, tag=p))

	with PagePipeline(slots=slots, streams=streams) as pipe:
		def run(n):
			last = None
			for result in pipe.map(jobs[j % n_pages] for j in range(n)):
				last = result  # the consumer takes delivery (and drops the previous page)
			return last

		# Warm-up: one whole cycle over the host ring (the copy engines take some 60 pages to reach their rate:
		# tools/pipeline_segments.py, profiles/r02/q_pipeline_warmup.log); configs[2] is a queue of 10 000 jobs.
		run(max(2 * slots, n_pages))
		barrier()
		t0 = time.perf_counter()
		last = run(n_jobs)
		elapsed = time.perf_counter() - t0
		barrier()
	if not np.array_equal(last.page, wl["outs"][last.tag].cpu().numpy()):
		raise SystemExit("bench: pipeline output differs from the resident path")
	up = n_jobs * H * W
	down = n_jobs * Hd * Wd + sum(wl["polys"][j % n_pages]["n"] for j in range(n_jobs)) * 16
	return elapsed, up, down


def launch_ranks(n):
	"""`python bench.py --gpus N` without a launcher: start N ranks under torch.distributed.run as a CHILD process
	and leave with its exit code.  Called before this process has touched a GPU (nothing is re-exec'ed)."""
	import socket
	import subprocess
	with socket.socket() as sock:
		sock.bind(("127.0.0.1", 0))
		port = sock.getsockname()[1]
	cmd = [
		sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
		"--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
	env = dict(os.environ, MASTER_ADDR="127.0.0.1")
	env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
	env.setdefault("OMP_NUM_THREADS", "2")
	return subprocess.call(cmd, env=env, cwd=str(ROOT))


def timed_steps(step, steps, warmup, sync, dist, world, device, between=None):
	"""The timing contract: `warmup` untimed steps, then exactly `steps` steps bracketed by a barrier +
	device synchronisation on both sides; returns the MAX over ranks of the elapsed seconds."""
	import torch

	def barrier():
		sync()
		if world > 1:
			dist.barrier()
			sync()

	for _ in range(warmup):
		step(False)
	barrier()
	if between is not None:
		between()
		barrier()
	t0 = time.perf_counter()
	for _ in range(steps):
		step(True)
	barrier()
	elapsed = time.perf_counter() - t0
	if world > 1:
		t = torch.tensor([elapsed], dtype=torch.float64, device=device)
		dist.all_reduce(t, op=dist.ReduceOp.MAX)
		elapsed = float(t.item())
	return elapsed


def harness_selftest(args, rank, world):
	"""Ranks sleep (rank + 1) * 10 ms per step: the reported time must be the slowest rank's."""
	import torch
	import torch.distributed as dist
	if world > 1:
		os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
		dist.init_process_group("gloo", rank=rank, world_size=world)
	calls = []

	def step(timed):
		calls.append(timed)
		time.sleep(0.01 * (rank + 1))

	elapsed = timed_steps(step, args.steps, args.warmup, lambda: None, dist, world, torch.device("cpu"))
	assert calls == [False] * args.warmup + [True] * args.steps
	if rank == 0:
		print(json.dumps({"selftest": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
			"ms_per_step": round(elapsed / args.steps * 1e3, 3), "value": world * args.steps / elapsed}), flush=True)
	if world > 1:
		dist.barrier()
		dist.destroy_process_group()


def main():
	args = parse_args()
	rank = int(os.environ.get("RANK", "0"))
	local_rank = int(os.environ.get("LOCAL_RANK", "0"))
	world = int(os.environ.get("WORLD_SIZE", "1"))
	if args.gpus < 1:
		raise SystemExit("--gpus must be >= 1")
	if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
		sys.exit(launch_ranks(args.gpus))  # no launcher: be one (child process; this one never touches a GPU)
	if world != args.gpus:
		raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
	if args.harness_selftest:
		return harness_selftest(args, rank, world)

	import torch
	import torch.distributed as dist
	from origami_amd import _shim

	if not torch.cuda.is_available():
		raise SystemExit("bench.py needs a GPU (no CPU fallback exists on the product path)")
	# ranks of an N > 1 run stage their pinned rings through the DRAM of their own GPU's socket (origami_amd/numa.py:
	# the helper the stage workers use)
	from origami_amd import numa
	numa_node = numa.bind_to_gpu_numa_node(local_rank) if world > 1 else None
	torch.cuda.set_device(local_rank)
	dev = torch.device("cuda", local_rank)
	_shim.init(local_rank)
	lib = _shim.load()
	check = _shim.check
	if world > 1:
		os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
		dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

	wl = build_workload(torch, _shim, dev, args.pages, rank)
	stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
	side = torch.cuda.Stream()  # contour warps (K3) overlap the pixel kernels; both are synchronised by the barrier
	side_stream = ctypes.c_void_p(side.cuda_stream)

	events = []
	elapsed = timed_steps(
		lambda timed: run_step(torch, lib, check, wl, stream, events if timed else None, side_stream),
		args.steps, args.warmup, torch.cuda.synchronize, dist, world, dev, between=lambda: verify(torch, wl))

	fused_mps, fused_ms = fused_pass(torch, lib, check, wl, stream, side_stream)
	if world > 1:
		t = torch.tensor([fused_mps], dtype=torch.float64, device=dev)
		dist.all_reduce(t, op=dist.ReduceOp.SUM)
		fused_mps = float(t.item())

	pipeline = None
	if not args.no_pipeline:
		def barrier():
			torch.cuda.synchronize()
			if world > 1:
				dist.barrier()
				torch.cuda.synchronize()
		n_jobs = args.pipeline_jobs or 2 * args.pages
		# kernels on the stream the timed steps used, uploads on their side stream, downloads on a third
		third = torch.cuda.Stream()
		p_elapsed, up, down = pipeline_leg(
			torch, wl, n_jobs, barrier, (side.cuda_stream, torch.cuda.current_stream().cuda_stream, third.cuda_stream))
		t = torch.tensor([p_elapsed], dtype=torch.float64, device=dev)
		if world > 1:
			dist.all_reduce(t, op=dist.ReduceOp.MAX)
		p_elapsed = float(t.item())
		pipeline = dict(n_jobs=n_jobs, elapsed=p_elapsed, up=up, down=down)

	Hd, Wd = wl["Hd"], wl["Wd"]
	W, H = PAGE
	mp_per_step = args.pages * Hd * Wd / 1e6
	value = world * mp_per_step * args.steps / elapsed
	k2_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
	k2_bytes = Hd * Wd * (8 + 1) + H * W * 1
	achieved = k2_bytes / (k2_ms * 1e-3) / 1e9

	if rank == 0:
		line = {
			"metric": "dewarped megapixels/sec (whole node)",
			"value": round(value, 1),
			"unit": "MP/s",
			"n_gpus": world,
			"steps": args.steps,
			"warmup": args.warmup,
			"ms_per_step": round(elapsed / args.steps * 1e3, 3),
			"higher_is_better": True,
			"scaling": "weak",
			"vs_baseline": None,
			"dtype": "u8",
			"data": "synthetic",
			"config": {
				"workload": "configs[1]: %d synthetic 6000x8000 grayscale pages per GPU, cell %d, dense grid %dx%d, %d contour rings/page; K1a grid fill + K2 remap + K3 contour warp per page" % (
					args.pages, CELL, Wd, Hd, N_POLYGONS),
				"pages_per_gpu": args.pages,
				"distinct_warps": len(wl["warps"]),
				"parallelism": "page-sharded x%d, no collective" % world,
			},
			"fused_path": {
				"what": "same batch through ogd_dewarp_u8_dev (K1a fused into K2, no dense map; what Dewarper.dewarped runs) + K3; measured after the timed steps, not part of value",
				"value": round(fused_mps, 1), "unit": "MP/s", "ms_per_page": round(fused_ms, 4),
			},
			"roofline": {
				"kernel": K2_KERNEL,
				"bound": "hbm",
				"achieved": round(achieved, 1),
				"peak": HBM_PEAK_GBS,
				"unit": "GB/s",
				"frac": round(achieved / HBM_PEAK_GBS, 4),
				"traffic": measured_traffic(k2_bytes),
				"algorithmic_bytes_per_launch": k2_bytes,
				"avg_launch_ms": round(k2_ms, 4),
			},
		}
		if pipeline is not None:
			pe = pipeline["elapsed"]
			line["pipeline"] = {
				"what": "configs[2] (SURVEY 8d ii) through the product's origami_amd.pipeline.PagePipeline: %d page jobs per GPU cycled from a pinned host ring of the batch's %d pages -> H2D -> fused dewarp (ogd_dewarp_u8_dev) + K3 -> D2H of page and contours, 3 streams / 3 device slots per GPU, at most 3 jobs in flight; barrier + max over ranks; PCIe-inclusive, never `value`" % (
					pipeline["n_jobs"], args.pages),
				"value": round(world * pipeline["n_jobs"] * Hd * Wd / 1e6 / pe, 1), "unit": "MP/s",
				"ms_per_page_per_gpu": round(pe / pipeline["n_jobs"] * 1e3, 4),
				"h2d_GBs_per_gpu": round(pipeline["up"] / pe / 1e9, 2),
				"d2h_GBs_per_gpu": round(pipeline["down"] / pe / 1e9, 2),
				"numa_node_rank0": numa_node,
			}
		if not args.no_cpu_baseline and world == 1:
			threads = args.cpu_threads or min(16, os.cpu_count() or 1)
			line["cpu_baseline"] = cpu_baseline(wl, threads)
		print(json.dumps(line), flush=True)
	if world > 1:
		dist.barrier()
		dist.destroy_process_group()


if __name__ == "__main__":
	main()
