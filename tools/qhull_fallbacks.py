#!/usr/bin/env python3
"""How often InterpolatorFactory._triangulate leaves ogd_delaunay_points for Qhull in the `dewarp` stage, and what each
of its steps costs (host time per call).  usage: qhull_fallbacks.py [n_pages]"""
import collections
import os
import sys
import tempfile
import shutil
import time
from pathlib import Path

import numpy as np
import scipy.spatial
from click.testing import CliRunner

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, synth, lingrid  # noqa: E402
from origami_amd.batch.detect.dewarp import dewarp  # noqa: E402

spent = collections.defaultdict(lambda: [0, 0.0])


def timed(name, fn):
	def wrapper(*args, **kwargs):
		t = time.perf_counter()
		try:
			return fn(*args, **kwargs)
		finally:
			rec = spent[name]
			rec[0] += 1
			rec[1] += time.perf_counter() - t
	return wrapper


def main():
	n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 4
	_shim.init(0)
	root = Path(tempfile.mkdtemp(prefix="ogd_qh_"))
	try:
		rng = np.random.default_rng(1)
		first = synth.write_page_dir(root, "page0000", rng, size=(6000, 8000), n_h=1500, n_v=2000, n_regions=400)
		for k in range(1, n_pages):
			page = root / ("page%04d.png" % k)
			os.link(first, page)
			synth.write_page_inputs(page, rng, size=(6000, 8000), n_h=1500, n_v=2000, n_regions=400)
		args = [str(root), "--lock-strategy", "NONE", "--plain", "--overwrite"]
		assert CliRunner().invoke(dewarp, args[:-1]).exit_code == 0
		real = scipy.spatial.Delaunay
		sizes = []

		def counted(points, *a, **kw):
			sizes.append(len(points))
			return real(points, *a, **kw)
		scipy.spatial.Delaunay = timed("scipy.spatial.Delaunay", counted)
		_shim.delaunay_points = timed("delaunay_points", _shim.delaunay_points)
		_shim.lattice_delaunay = timed("lattice_delaunay", _shim.lattice_delaunay)
		_shim.Triangulation.__init__ = timed("Triangulation", _shim.Triangulation.__init__)
		lingrid.InterpolatorFactory._triangulate = staticmethod(timed("_triangulate", lingrid.InterpolatorFactory._triangulate))
		lingrid.InterpolatorFactory.__init__ = timed("InterpolatorFactory.__init__", lingrid.InterpolatorFactory.__init__)
		t = time.perf_counter()
		res = CliRunner().invoke(dewarp, args)
		assert res.exit_code == 0, res.output
		print("stage: %.1f ms/page" % (1e3 * (time.perf_counter() - t) / n_pages))
		for name, (n, dt) in sorted(spent.items()):
			print("%-30s %4d calls %8.2f ms/page %8.3f ms/call" % (name, n, 1e3 * dt / n_pages, 1e3 * dt / n))
		print("Qhull fallbacks on point counts:", sizes)
	finally:
		shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
	main()
