#!/opt/conda/bin/python3.9
"""Qhull's triangulations of the golden inputs under SciPy 1.7.1 (the build container's second interpreter,
/opt/conda/bin/python3.9) -> tests/golden/scipy171.npz.

    /opt/conda/bin/python3.9 tests/golden/make_scipy171.py

The reference pins scipy==1.3.1 (docker/pip_requirements.txt:155); the tests and the product run SciPy 1.15.3.
Nodes of a marched grid are nearly cocircular, so Qhull's choice of diagonal may depend on its version
(SURVEY.md §7 hard part 3, §8c).  This script records what the older SciPy gives for every Delaunay / ConvexHull
input of the golden fixtures — the inner point set and the extrapolation point set (oracle/lingrid.py:Factory)
of the angle fields of field.npz and grid.npz and of the transformer of transformer.npz —
and tests/test_oracle.py::test_qhull_diagonals_scipy_171_vs_current counts the flips against the current SciPy.
Only scipy / numpy code runs here; nothing of the reference is involved.
"""

import sys
from pathlib import Path

import numpy as np
import scipy
import scipy.spatial

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))

from oracle import lingrid as olg  # noqa: E402  (numpy / scipy only)


def point_sets():
	"""name -> (points, values, bounds) for every lininterp() the golden fixtures go through."""
	sets = {}
	z = np.load(HERE / "field.npz")
	W, H = (int(v) for v in z["size"])
	sets["field_n64"] = (z["n64_points"], z["n64_values"], (0, 0, W, H))
	z = np.load(HERE / "grid.npz")
	W, H = (int(v) for v in z["size"])
	sets["grid_h"] = (z["h_points"], z["h_values"], (0, 0, W, H))
	sets["grid_v"] = (z["v_points"], z["v_values"], (0, 0, W, H))
	z = np.load(HERE / "transformer.npz")
	from oracle import cv as ocv  # the C restatement of cv2.resize, through ctypes
	src = ocv.dense_grid(z["nodes"], 25)[::25, ::25].astype(np.float64)  # points("full")[::r, ::r], ref: dewarp.py:557-559
	sets["transformer"] = _transformer_set(src)
	# configs[0] geometry: the transformer of a 2400 x 3400 page at cell 25 (139 x 99 nodes, ~27 k triangles)
	from origami_amd import synth  # input generator only
	nodes = synth.Warp((2400, 3400), 0.03, 0.025, 1, 2).node_grid(25)
	sets["transformer_cfg0"] = _transformer_set(ocv.dense_grid(nodes, 25)[::25, ::25].astype(np.float64))
	return sets


def _transformer_set(src):
	h, w = src.shape[:2]
	src = src.reshape(h * w, 2)
	gy, gx = np.mgrid[0:h, 0:w]
	target = np.stack([gx, gy], axis=-1).reshape(h * w, 2) * 25
	bounds = (src[:, 0].min(), src[:, 1].min(), src[:, 0].max(), src[:, 1].max())
	return src, target.astype(np.float64), bounds


def triangulate(points):
	tri = scipy.spatial.Delaunay(np.asarray(points, dtype=np.float64))
	return np.sort(tri.simplices, axis=1).astype(np.int32)


def main():
	out = dict(scipy_version=np.array(scipy.__version__))
	for name, (pts, vals, bounds) in point_sets().items():
		vals_list = list(vals) if np.ndim(vals) == 1 else [v for v in vals]
		fac = olg.Factory([p for p in np.asarray(pts, dtype=np.float64)], vals_list, bounds)
		out[name + "_hull"] = scipy.spatial.ConvexHull(np.asarray(pts, dtype=np.float64)).vertices.astype(np.int32)
		out[name + "_inner"] = triangulate(pts)
		out[name + "_points"] = np.asarray(pts, dtype=np.float64)
		out[name + "_extra_points"] = np.asarray(fac.extra_pts, dtype=np.float64)
		out[name + "_extra"] = triangulate(fac.extra_pts)
	np.savez_compressed(HERE / "scipy171.npz", **out)
	print("scipy", scipy.__version__, {k: v.shape for k, v in out.items() if k != "scipy_version"})


if __name__ == "__main__":
	main()
