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
for seg in range(8):
	t0 = time.perf_counter()
	run(32)
	print("segment %d: %.2f ms per page" % (seg, (time.perf_counter() - t0) / 32 * 1e3), flush=True)
