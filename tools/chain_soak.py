#!/usr/bin/env python3
"""`utils.dewarp` + `export.lines` chained over the same pages again and again in one process, with a resident-page
budget smaller than the pages (evictions every round): device memory in use and the resident set must level off.
usage: chain_soak.py [rounds] [pages]"""
import os
import resource
import shutil
import sys
import tempfile
from pathlib import Path

import numpy as np
import PIL.Image

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, synth  # noqa: E402
from origami_amd import page as page_mod  # noqa: E402
from origami_amd.batch import chain  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n_pages = int(sys.argv[2]) if len(sys.argv) > 2 else 6
W, H, s = 3000, 4000, 25
_shim.init(0)
rng = np.random.default_rng(4)
root = Path(tempfile.mkdtemp(prefix="ogd_chain_soak_"))
try:
	for k in range(n_pages):
		path = root / ("page%04d%s" % (k, ".jpg" if k % 2 else ".png"))
		im = PIL.Image.fromarray(synth.text_page(rng, (W, H)))
		im.save(path, quality=90) if path.suffix == ".jpg" else im.save(path, compress_level=1)
		synth.write_reliable_stage(path, rng, synth.Warp((W, H), 0.03, 0.02, 1, 2).node_grid(s), s, n_lines=60, n_table_lines=2)
	page_mod.resident = cache = page_mod.ResidentPages(int(2.5 * W * H))  # room for two of the pages
	stages = [("utils.dewarp", ["--lock-strategy", "NONE", "--overwrite"]), ("export.lines", ["--lock-strategy", "NONE", "--overwrite"])]
	devnull = open(os.devnull, "w")
	for r in range(rounds):
		out, sys.stdout = sys.stdout, devnull
		try:
			chain.run_chain(str(root), stages, pages_per_round=2)
		finally:
			sys.stdout = out
		if r % 5 == 4 or r == 0:
			import ctypes
			f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
			_shim.check(_shim.load().ogd_mem_info(ctypes.byref(f), ctypes.byref(t)))
			free, total = f.value, t.value
			print("round %3d: device memory in use %6.0f MB, resident set %5.0f MB, cache %d pages / %.0f MB, %d hits %d misses %d evictions" % (
				r + 1, (total - free) / 2**20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024, len(cache), cache.nbytes / 2**20,
				cache.hits, cache.misses, cache.evictions), flush=True)
finally:
	shutil.rmtree(root, ignore_errors=True)
