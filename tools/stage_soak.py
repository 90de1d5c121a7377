#!/usr/bin/env python3
"""Leak check for long runs: the `dewarp` stage and the `utils.dewarp` stage over the same pages again and again in
ONE process (a worker of a 10 000-page job lives that long), printing the process's resident set and the device
memory in use after every pass.  Both must level off after the first passes (block cache and pools filled).
usage: stage_soak.py [n_pages [passes [W H]]]"""
import os
import shutil
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import PIL.Image
import torch
from click.testing import CliRunner

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, synth  # noqa: E402
from origami_amd.batch.detect.dewarp import dewarp  # noqa: E402
from origami_amd.batch.utils.dewarp import dewarp as dewarp_image_stage  # noqa: E402


def rss_mb():
	with open("/proc/self/statm") as f:
		return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2 ** 20


def device_used_mb():
	free, total = torch.cuda.mem_get_info(0)
	return (total - free) / 2 ** 20


def main():
	n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 8
	passes = int(sys.argv[2]) if len(sys.argv) > 2 else 12
	W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3000, 4000)
	_shim.init(0)
	root = Path(tempfile.mkdtemp(prefix="ogd_soak_"))
	try:
		rng = np.random.default_rng(3)
		first = synth.write_page_dir(root, "page0000", rng, size=(W, H), n_h=600, n_v=800, n_regions=150)
		jpg = root.parent / (root.name + "_source.jpg")  # outside the data folder: not a page of its own
		PIL.Image.open(first).save(jpg, quality=90)
		for k in range(1, n_pages):
			page = root / ("page%04d.%s" % (k, "png" if k % 2 else "jpg"))
			shutil.copyfile(first if k % 2 else jpg, page)
			synth.write_page_inputs(page, rng, size=(W, H), n_h=600, n_v=800, n_regions=150)
		args = [str(root), "--lock-strategy", "NONE", "--plain", "--overwrite"]
		trace = os.environ.get("SOAK_TRACEMALLOC")  # Python-level allocations that survive from pass 3 to the last
		if trace:
			import tracemalloc
			tracemalloc.start(12)
		for k in range(passes):
			if trace and k == 3:
				before = tracemalloc.take_snapshot()
			t = time.perf_counter()
			for stage in (dewarp, dewarp_image_stage):
				res = CliRunner().invoke(stage, args)
				assert res.exit_code == 0, res.output
			print("pass %2d: %.2f s, resident set %7.1f MB, device memory in use %7.1f MB" % (
				k, time.perf_counter() - t, rss_mb(), device_used_mb()), flush=True)
		if trace:
			import gc
			gc.collect()
			after = tracemalloc.take_snapshot()
			for stat in after.compare_to(before, "traceback")[:8]:
				print("%+.1f KB in %+d blocks" % (stat.size_diff / 1024, stat.count_diff))
				for line in stat.traceback.format()[-6:]:
					print("   ", line)
	finally:
		shutil.rmtree(root, ignore_errors=True)
		try:
			os.remove(root.parent / (root.name + "_source.jpg"))
		except OSError:
			pass


if __name__ == "__main__":
	main()
