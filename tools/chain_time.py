#!/usr/bin/env python3
"""`utils.dewarp` + `export.lines` over the same broadsheet pages: each stage alone with nothing kept between them
(the reference's way: every stage opens and remaps the page again, ref: origami/batch/core/io.py:275-280) against
both chained in one process over resident pages (origami_amd.batch.chain, origami_amd.page.resident).
usage: chain_time.py [n_pages] [.png|.jpg] [n_lines]"""
import shutil
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import PIL.Image

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from click.testing import CliRunner  # noqa: E402

from origami_amd import _shim, synth  # noqa: E402
from origami_amd import page as page_mod  # noqa: E402
from origami_amd.batch import chain  # noqa: E402
from origami_amd.batch.export.lines import export_lines  # noqa: E402
from origami_amd.batch.utils.dewarp import dewarp as dewarp_image_stage  # noqa: E402

n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 8
suffix = sys.argv[2] if len(sys.argv) > 2 else ".png"
n_lines = int(sys.argv[3]) if len(sys.argv) > 3 else 300
W, H, s = 6000, 8000, 25
_shim.init(0)
rng = np.random.default_rng(9)
root = Path(tempfile.mkdtemp(prefix="ogd_chain_time_"))
try:
	base = root / "base"
	base.mkdir()
	pixels = synth.text_page(rng, (W, H))
	for k in range(n_pages):
		nodes = synth.Warp((W, H), 0.03, 0.02, 1 + k % 2, 2 - k % 2).node_grid(s)
		path = base / ("page%04d%s" % (k, suffix))
		im = PIL.Image.fromarray(np.roll(pixels, 37 * k, axis=1))
		im.save(path, quality=90) if suffix == ".jpg" else im.save(path, compress_level=1)
		synth.write_reliable_stage(path, rng, nodes, s, n_lines=n_lines, n_table_lines=4)
	mp = n_pages * W * H / 1e6
	for rep in range(2):
		alone, chained = root / ("alone%d" % rep), root / ("chained%d" % rep)
		shutil.copytree(base, alone)
		shutil.copytree(base, chained)
		page_mod.resident = page_mod.ResidentPages(0)
		t0 = time.perf_counter()
		for command in (dewarp_image_stage, export_lines):
			res = CliRunner().invoke(command, [str(alone), "--lock-strategy", "NONE"])
			assert res.exit_code == 0, res.output
		t_alone = time.perf_counter() - t0
		page_mod.resident = cache = page_mod.ResidentPages(2048 << 20)
		t0 = time.perf_counter()
		chain.run_chain(str(chained), [("utils.dewarp", ["--lock-strategy", "NONE"]), ("export.lines", ["--lock-strategy", "NONE"])], pages_per_round=16)
		t_chain = time.perf_counter() - t0
		print("pass %d, %d %s pages 6000x8000, %d lines each: stages alone %.3f s = %.1f ms/page | chained %.3f s = %.1f ms/page (%d misses, %d hits)" % (
			rep, n_pages, suffix, n_lines, t_alone, 1e3 * t_alone / n_pages, t_chain, 1e3 * t_chain / n_pages, cache.misses, cache.hits), flush=True)
		same = all((alone / d.name / "annotation.dewarped.jpg").read_bytes() == (d / "annotation.dewarped.jpg").read_bytes() and
			(alone / d.name / "images.lines.dewarped.zip").stat().st_size == (d / "images.lines.dewarped.zip").stat().st_size
			for d in chained.glob("*.out"))
		print("        same files:", same, flush=True)
finally:
	shutil.rmtree(root, ignore_errors=True)
