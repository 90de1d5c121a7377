#!/usr/bin/env python3
"""tools/pipeline_bench.py's job loop timed in segments of 32 pages: how long the upload / kernels / download pipeline
takes to reach its settled rate (PIPE_MODE=noevents: the two copy directions alone).  usage: pipeline_segments.py"""
import os
import time

src = open(os.path.join(os.path.dirname(__file__), "pipeline_bench.py")).read()
cut = src.index("run(slots * 2)")
exec(compile(src[:cut], "pipeline_bench_head", "exec"))
if os.environ.get("PIPE_BIG_FOOTPRINT"):  # experiment: a resident batch beside the pipeline, as in bench.py
	resident_in = torch.zeros((64, H, W), dtype=torch.uint8, device=dev)
	resident_out = torch.zeros((64, Hd, Wd), dtype=torch.uint8, device=dev)
	big_ring = torch.empty((64, H, W), dtype=torch.uint8, pin_memory=True)
	big_ring.copy_(resident_in)
	torch.cuda.synchronize()
if os.environ.get("PIPE_LIKE_BENCH"):  # experiment: bench.py's workload built (and kept) in this process first
	import importlib.util
	spec = importlib.util.spec_from_file_location("bench", str(Path(__file__).resolve().parent.parent / "bench.py"))
	bench = importlib.util.module_from_spec(spec)
	spec.loader.exec_module(bench)
	keep = bench.build_workload(torch, _shim, dev, int(os.environ["PIPE_LIKE_BENCH"]), 0)
	torch.cuda.synchronize()
	what = os.environ.get("PIPE_LIKE_BENCH_STEPS", "")
	if what:
		import ctypes
		stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
		side = torch.cuda.Stream()
		side_stream = ctypes.c_void_p(side.cuda_stream)
		events = []
		for k in range(4):
			bench.run_step(torch, lib, check, keep, stream, events if "events" in what else None, side_stream if "side" in what else None)
		torch.cuda.synchronize()
		if "verify" in what:
			bench.verify(torch, keep)
		if "fused" in what:
			bench.fused_pass(torch, lib, check, keep, stream, side_stream)
		torch.cuda.synchronize()
late = os.environ.get("PIPE_LATE", "")  # experiment: the pipeline's own objects created AFTER the workload, as in bench.py
if "host" in late:
	host_in = torch.empty((ring, H, W), dtype=torch.uint8, pin_memory=True)
	host_in.fill_(200)
	host_out = torch.empty((ring, Hd, Wd), dtype=torch.uint8, pin_memory=True)
if "device" in late:
	d_in = torch.empty((slots, H, W), dtype=torch.uint8, device=dev)
	d_out = torch.empty((slots, Hd, Wd), dtype=torch.uint8, device=dev)
if "streams" in late:
	s_in, s_k, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
if "events" in late:
	ev_in = [torch.cuda.Event() for _ in range(slots)]
	ev_k = [torch.cuda.Event() for _ in range(slots)]
	ev_out = [torch.cuda.Event() for _ in range(slots)]
for seg in range(8):
	t0 = time.perf_counter()
	run(32)
	print("segment %d: %.2f ms per page" % (seg, (time.perf_counter() - t0) / 32 * 1e3), flush=True)
