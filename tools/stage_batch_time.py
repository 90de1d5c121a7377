#!/usr/bin/env python3
"""Throughput (iii) of SURVEY §8d: the `dewarp` batch stage end to end (directory walk, artifact I/O,
grid construction, contour warp, zip output) on synthetic broadsheet page directories, one worker on one GPU.
usage: stage_batch_time.py [n_pages n_polygons [W H [workers[,workers...]]]]   (workers > 0: spawned workers, all on GPU 0 of a one-GPU box; a list times the stage once per count on the same pages)   (never bench.py's `value`)"""
import cProfile
import os
import pstats
import shutil
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
from click.testing import CliRunner

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, synth  # noqa: E402
from origami_amd.batch.detect.dewarp import dewarp  # noqa: E402

def main():
	n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 6
	n_poly = int(sys.argv[2]) if len(sys.argv) > 2 else 400
	W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (6000, 8000)
	worker_counts = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0]
	if max(worker_counts) > 15:  # the GPU boxes allow 16 processes on a GPU at once, and this one is one of them
		raise SystemExit("at most 15 workers")
	workers = worker_counts[0]

	_shim.init(0)
	root = Path(tempfile.mkdtemp(prefix="ogd_stage_"))
	try:
		rng = np.random.default_rng(1)
		t0 = time.perf_counter()
		first = synth.write_page_dir(root, "page0000", rng, size=(W, H), n_h=1500, n_v=2000, n_regions=n_poly)
		for k in range(1, n_pages):  # same pixels (the stage reads only the size), own warp / flow / contours
			page = root / ("page%04d.png" % k)
			try:
				os.link(first, page)  # the stage reads only the size of the page image
			except OSError:
				shutil.copyfile(first, page)
			synth.write_page_inputs(page, rng, size=(W, H), n_h=1500, n_v=2000, n_regions=n_poly)
			if k % 16 == 0:
				print("  page %d written" % k, flush=True)
		print("wrote %d page directories %dx%d, %d polygons each, in %.1f s" % (n_pages, W, H, n_poly, time.perf_counter() - t0))

		def run(overwrite, workers=workers):
			args = [str(root), "--lock-strategy", "NONE", "--plain"] + (["--overwrite"] if overwrite else [])
			if workers:
				args += ["--gpus", "1", "--workers-per-gpu", str(workers)]
			t = time.perf_counter()
			res = CliRunner().invoke(dewarp, args)
			assert res.exit_code == 0, res.output
			return time.perf_counter() - t

		dt = run(False)
		print("first pass : %.3f s = %.1f ms/page (includes context creation and first-use costs)" % (dt, 1e3 * dt / n_pages), flush=True)
		dt = run(True)
		mp = n_pages * W * H / 1e6
		print("second pass: %.3f s = %.1f ms/page = %.0f MP/s (page megapixels per second through the stage)" % (
			dt, 1e3 * dt / n_pages, mp / dt))
		if len(worker_counts) > 1:
			import resource
			for n in worker_counts[1:]:
				r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
				dt = run(True, n)
				r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
				cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
				print("%2d workers : %.3f s = %.1f ms/page = %.0f MP/s; workers' CPU time %.1f s = %.1f core-ms/page (user %.1f s, system %.1f s), %.1f cores busy" % (
					n, dt, 1e3 * dt / n_pages, mp / dt, cpu, 1e3 * cpu / n_pages, r1.ru_utime - r0.ru_utime, r1.ru_stime - r0.ru_stime, cpu / dt), flush=True)
			return
		# the stage that writes annotation.dewarped.jpg from dewarp.zip (ref: origami/batch/utils/dewarp.py)
		from origami_amd.batch.utils.dewarp import dewarp as dewarp_image_stage

		def run_image_stage():
			args = [str(root), "--lock-strategy", "NONE", "--plain", "--overwrite"] + (["--gpus", "1", "--workers-per-gpu", str(workers)] if workers else [])
			t = time.perf_counter()
			res = CliRunner().invoke(dewarp_image_stage, args)
			assert res.exit_code == 0, res.output
			return time.perf_counter() - t

		run_image_stage()
		dt = run_image_stage()
		import PIL.Image
		t = time.perf_counter()
		PIL.Image.open(first).convert("L").load()
		t_dec = time.perf_counter() - t
		print("utils.dewarp stage (page PNG -> annotation.dewarped.jpg): %.3f s = %.1f ms/page, of which PIL decodes the page for %.1f ms" % (
			dt, 1e3 * dt / n_pages, 1e3 * t_dec))
		# the same pages as grayscale JPEG scans: decoded on the device (ogd_jpeg_decode_gray)
		im = PIL.Image.open(first).convert("L")
		for k in range(n_pages):
			png = root / ("page%04d.png" % k)
			im.save(png.with_suffix(".jpg"), quality=90)
			png.unlink()
		run_image_stage()
		dt = run_image_stage()
		t = time.perf_counter()
		PIL.Image.open(root / "page0000.jpg").convert("L").load()
		t_dec = time.perf_counter() - t
		print("utils.dewarp stage (page JPG -> annotation.dewarped.jpg): %.3f s = %.1f ms/page (PIL would decode such a page in %.1f ms)" % (
			dt, 1e3 * dt / n_pages, 1e3 * t_dec))
		if not workers and os.environ.get("STAGE_PROFILE_JPG"):
			pr = cProfile.Profile()
			pr.enable()
			run_image_stage()
			pr.disable()
			pstats.Stats(pr).sort_stats("tottime").print_stats(25)
		for k in range(n_pages):  # back to PNG pages for the profile below
			jpg = root / ("page%04d.jpg" % k)
			im.save(jpg.with_suffix(".png"), compress_level=1)
			jpg.unlink()
		if not workers:
			pr = cProfile.Profile()
			pr.enable()
			run(True)
			pr.disable()
			pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
			pstats.Stats(pr).sort_stats("tottime").print_stats(22)
			if os.environ.get("STAGE_PROFILE_SERIAL"):
				# the same with the side thread's half in the main thread, so that the profile holds a page's whole host work
				from origami_amd.batch.core.processor import Processor
				from origami_amd.batch.detect.dewarp import DewarpProcessor
				DewarpProcessor.prefetch = Processor.prefetch
				run(True)
				pr = cProfile.Profile()
				pr.enable()
				dt = run(True)
				pr.disable()
				print("without prefetch, profiled: %.1f ms/page" % (1e3 * dt / n_pages))
				pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
				pstats.Stats(pr).sort_stats("tottime").print_stats(40)
	finally:
		shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":  # spawned workers import this module: nothing may run at import time
	main()
