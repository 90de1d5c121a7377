"""CPU tests of the oracle itself (no GPU): hand-derived known answers for the restated cv2 calls,
the scipy-backed interpolation against scipy called directly, and the committed golden fixtures.

The reference holds no tests or fixtures for this path, so the known answers below are
derived by hand from the published OpenCV 4.2 arithmetic — PARITY UNPINNED for
the cv2 part."""

from pathlib import Path

import numpy as np
import pytest
import scipy.interpolate

from oracle import cv as ocv
from oracle import dewarp as odw
from oracle import lingrid as olin

GOLDEN = Path(__file__).resolve().parent / "golden"


def golden(name):
	return np.load(GOLDEN / ("%s.npz" % name), allow_pickle=False)


# ---- cv2.resize(INTER_LINEAR) restatement ------------------------------------------------------------

def test_resize_known_answer_2x2():
	# the classic cv2.resize([[0, 1], [2, 3]], (4, 4)) result: taps at -0.25, 0.25, 0.75, 1.25 (clamped)
	src = np.array([[0, 1], [2, 3]], dtype=np.float32)
	want = np.array([
		[0.0, 0.25, 0.75, 1.0],
		[0.5, 0.75, 1.25, 1.5],
		[1.5, 1.75, 2.25, 2.5],
		[2.0, 2.25, 2.75, 3.0]], dtype=np.float32)
	np.testing.assert_array_equal(ocv.resize_linear_f32(src, 2), want)


@pytest.mark.parametrize("s", [1, 5, 25])
def test_resize_nodes_sit_on_cell_centres(s):
	# odd cell: node g lands exactly on dense pixel g*s + (s-1)/2; edges are flat
	rng = np.random.default_rng(s)
	src = rng.normal(0, 100, size=(6, 7)).astype(np.float32)
	dense = ocv.resize_linear_f32(src, s)
	assert dense.shape == (6 * s, 7 * s)
	c = (s - 1) // 2
	np.testing.assert_array_equal(dense[c::s, c::s], src)
	# columns beyond the last node centre are exactly flat (x taps are clamped with weight 0) ...
	np.testing.assert_array_equal(dense[:, -c - 1:], np.repeat(dense[:, -c - 1:][:, :1], c + 1, axis=1))
	# ... rows above the first node centre only up to fp32 rounding: the y weights keep their fraction
	# and blend the clipped row with itself (v * (1 - f) + v * f)
	np.testing.assert_allclose(dense[:c + 1, :], np.repeat(dense[c:c + 1, :], c + 1, axis=0), rtol=3e-7)


def test_resize_reproduces_affine_planes():
	gy, gx = np.mgrid[0:9, 0:11].astype(np.float32)
	plane = (3 * gx - 2 * gy + 7).astype(np.float32)
	dense = ocv.resize_linear_f32(plane, 8)
	X = (np.arange(11 * 8) + 0.5) / 8 - 0.5
	Y = (np.arange(9 * 8) + 0.5) / 8 - 0.5
	inside = dense[4:-4, 4:-4]
	want = 3 * X[None, 4:-4] - 2 * Y[4:-4, None] + 7
	np.testing.assert_allclose(inside, want, atol=1e-4)


# ---- cv2.remap restatement ---------------------------------------------------------------------------

SRC4 = np.array([[10, 20], [30, 40]], dtype=np.uint8)


def _remap1(x, y, interp=ocv.INTER_LINEAR, src=SRC4):
	m = np.array([[[x, y]]], dtype=np.float32)
	return int(ocv.remap_u8(src, m, interp)[0, 0])


def test_remap_linear_known_answers():
	assert _remap1(0, 0) == 10 and _remap1(1, 0) == 20 and _remap1(0, 1) == 30 and _remap1(1, 1) == 40
	# centre: the mean, (8192 * 100 + 16384) >> 15 = 25
	assert _remap1(0.5, 0.5) == (8192 * (10 + 20 + 30 + 40) + 16384) >> 15
	# 1/32 steps: x = 0.25 -> fx = 8
	assert _remap1(0.25, 0) == (24 * 32 * 32 * 10 + 8 * 32 * 32 * 20 + 16384) >> 15
	# coordinates are rounded to 1/32 px, half-to-even: 1/64 -> 0, 3/64 -> 2/32
	assert _remap1(1 / 64, 0) == 10
	assert _remap1(3 / 64, 0) == (30 * 32 * 32 * 10 + 2 * 32 * 32 * 20 + 16384) >> 15
	# BORDER_CONSTANT 0 blends in at the edges, fully outside is 0
	assert _remap1(-0.5, 0) == (16 * 32 * 32 * 10 + 16384) >> 15
	assert _remap1(1.5, 1) == (16 * 32 * 32 * 40 + 16384) >> 15
	assert _remap1(-1, 0) == 0 and _remap1(2, 0) == 0 and _remap1(0, 2.0) == 0
	assert _remap1(np.nan, 0) == 0 and _remap1(1e9, 0) == 0 and _remap1(-1e9, 0) == 0


def test_remap_nearest_known_answers():
	n = ocv.INTER_NEAREST
	assert _remap1(0.49, 0, n) == 10 and _remap1(0.51, 0, n) == 20
	assert _remap1(0.5, 0, n) == 10  # round-half-even: 0.5 -> 0
	assert _remap1(1.5, 1.5, n) == 0  # 1.5 -> 2 = outside
	assert _remap1(-0.5, 0, n) == 10  # -0.5 -> -0
	assert _remap1(-0.51, 0, n) == 0
	assert _remap1(np.nan, 0, n) == 0 and _remap1(70000, 0, n) == 0


def test_remap_identity_map_is_identity():
	rng = np.random.default_rng(0)
	src = rng.integers(0, 256, size=(17, 23, 3), dtype=np.uint8)
	yy, xx = np.mgrid[0:17, 0:23].astype(np.float32)
	m = np.stack([xx, yy], axis=-1)
	for interp in (ocv.INTER_LINEAR, ocv.INTER_NEAREST):
		np.testing.assert_array_equal(ocv.remap_u8(src, m, interp), src)


def test_remap_linear_is_within_one_lsb_of_float_bilinear_on_quantised_coords():
	rng = np.random.default_rng(1)
	src = rng.integers(0, 256, size=(40, 50), dtype=np.uint8)
	q = rng.integers(0, 32 * 38, size=(30, 30, 2)).astype(np.float32) / 32  # exact 1/32 coordinates
	got = ocv.remap_u8(src, q).astype(np.float64)
	x, y = q[..., 0].astype(np.float64), q[..., 1].astype(np.float64)
	ix, iy = np.floor(x).astype(int), np.floor(y).astype(int)
	fx, fy = x - ix, y - iy
	s = src.astype(np.float64)
	want = (s[iy, ix] * (1 - fx) + s[iy, ix + 1] * fx) * (1 - fy) + (s[iy + 1, ix] * (1 - fx) + s[iy + 1, ix + 1] * fx) * fy
	assert np.abs(got - want).max() <= 0.5 + 1e-9


def test_threads_do_not_change_results():
	rng = np.random.default_rng(2)
	nodes = rng.normal(0, 50, size=(7, 9, 2)).astype(np.float32)
	np.testing.assert_array_equal(ocv.dense_grid(nodes, 11, threads=1), ocv.dense_grid(nodes, 11, threads=4))


# ---- lingrid restatement against scipy called directly -----------------------------------------------

def test_lininterp_inside_hull_is_scipy_linear_nd():
	rng = np.random.default_rng(3)
	pts = rng.uniform([0, 0], [600, 800], size=(200, 2))
	vals = np.sin(pts[:, 0] / 90) + np.cos(pts[:, 1] / 70)
	f = olin.lininterp(list(pts), list(vals), (0, 0, 600, 800))
	direct = scipy.interpolate.LinearNDInterpolator(pts, vals)
	q = rng.uniform([0, 0], [600, 800], size=(2000, 2))
	d = direct(q)
	inside = ~np.isnan(d)
	assert inside.sum() > 1000
	np.testing.assert_array_equal(f(q)[inside], d[inside])
	assert not np.any(np.isnan(f(q)))  # the extrapolation covers the rest of the box


def test_lininterp_extrapolation_set_structure():
	# hull (CCW) + two projections per hull edge + four corners, corners last in TL, TR, BR, BL order
	pts = [np.array(p, dtype=np.float64) for p in [(100, 100), (500, 120), (480, 700), (90, 650), (300, 400)]]
	vals = [1.0, 2.0, 3.0, 4.0, 5.0]
	fac = olin.Factory(pts, vals, (0, 0, 600, 800))
	assert not fac.is_collinear
	assert len(fac.extra_pts) == 4 + 2 * 4 + 4
	np.testing.assert_array_equal(fac.extra_pts[-4:], [[-1, -1], [601, -1], [601, 801], [-1, 801]])
	# every projected point lies on the outset box
	proj = fac.extra_pts[4:-4]
	on_box = (proj[:, 0] == -1) | (proj[:, 0] == 601) | (proj[:, 1] == -1) | (proj[:, 1] == 801)
	assert on_box.all()
	# values of the projections are those of the hull vertices they come from
	assert set(fac.extra_val[4:-4]) <= {1.0, 2.0, 3.0, 4.0}


def test_lininterp_clips_queries_to_bounds():
	rng = np.random.default_rng(4)
	pts = rng.uniform([0, 0], [100, 100], size=(30, 2))
	vals = rng.normal(size=30)
	f = olin.lininterp(list(pts), list(vals), (0, 0, 100, 100))
	np.testing.assert_array_equal(f(np.array([[-50.0, 20.0], [150.0, 200.0]])), f(np.array([[0.0, 20.0], [100.0, 100.0]])))


def test_lininterp_collinear_and_too_few():
	pts = [np.array([100.0, 100.0]), np.array([200.0, 150.0]), np.array([300.0, 200.0])]
	fac = olin.Factory(pts, [0.01, 0.02, 0.04], (0, 0, 600, 800))
	assert fac.is_collinear
	out = fac.interpolator()(np.array([[200.0, 150.0], [0.0, 0.0], [600.0, 800.0]]))
	assert not np.any(np.isnan(out))
	# only the border projections carry values in the collinear case (lingrid.py:228-237), so even a
	# sample point is not reproduced exactly; results stay within the sample range
	assert np.all((out >= 0.01 - 1e-12) & (out <= 0.04 + 1e-12))
	with pytest.raises(ValueError, match="not enough values"):
		olin.lininterp(pts[:1], [0.5], (0, 0, 600, 800))


def test_field_constant_for_two_samples():
	f = odw.Field(odw.Samples((600, 800), [np.zeros(2), np.ones(2)], [0.1, 0.2]), (600, 800), np.pi / 2)
	d = f.get(np.zeros((5, 2), dtype=np.float32))
	assert d.dtype == np.float32 and d.shape == (5, 2)
	np.testing.assert_allclose(d, np.tile([np.cos(np.float32(np.pi / 2)), 1.0], (5, 1)), atol=1e-7)


# ---- golden fixtures ----------------------------------------------------------------------------------

def test_golden_field():
	g = golden("field")
	size = tuple(int(v) for v in g["size"])
	probes = g["probes"]
	f64 = odw.Field(odw.Samples(size, list(g["n64_points"]), list(g["n64_values"])), size, 0)
	np.testing.assert_array_equal(f64.get(probes.copy()), g["n64_field"])
	f3 = odw.Field(odw.Samples(size, list(g["n3_points"]), list(g["n3_values"])), size, 0)
	np.testing.assert_array_equal(f3.get(probes.copy()), g["n3_field"])
	f2 = odw.Field(odw.Samples(size, list(g["n3_points"][:2]), list(g["n3_values"][:2])), size, np.pi / 2)
	np.testing.assert_array_equal(f2.get(probes.copy()), g["n2_field"])


def test_golden_grid_create():
	g = golden("grid")
	size = tuple(int(v) for v in g["size"])
	grid = odw.Grid.create(
		size, odw.Samples(size, list(g["h_points"]), list(g["h_values"])),
		odw.Samples(size, list(g["v_points"]), list(g["v_values"])), grid_res=25)
	assert grid.hv.dtype == np.float32
	np.testing.assert_array_equal(grid.hv, g["grid_hv"])
	hv = grid.hv
	# the grid covers the page: border nodes lie outside on every side (dewarp.py:341-419)
	# (rows / columns appended later by the other axis' extension may poke back in by < 1 cell)
	assert hv[:, 0, 0].max() < 25 and hv[:, -1, 0].min() > size[0] - 25
	assert hv[0, :, 1].max() <= 0 and hv[-1, :, 1].min() >= size[1]
	# neighbouring nodes are about one cell apart
	dx = np.linalg.norm(hv[:, 1:] - hv[:, :-1], axis=-1)
	assert 10 < dx.min() and dx.max() < 40


def test_golden_dense_and_inverse():
	g = golden("dense")
	for s in (8, 25):
		np.testing.assert_array_equal(ocv.dense_grid(g["nodes_s%d" % s], s), g["dense_s%d" % s])
	got = odw.Grid(g["nodes_s8"], 8).inverse(g["inverse_q_xy"])
	np.testing.assert_allclose(got, g["inverse_s8"], rtol=0, atol=1e-9)


def test_golden_remap():
	g = golden("remap")
	src, m = g["src_rgb"], g["map"]
	gray = np.ascontiguousarray(src[:, :, 0])
	np.testing.assert_array_equal(ocv.remap_u8(gray, m, ocv.INTER_LINEAR), g["gray_linear"])
	np.testing.assert_array_equal(ocv.remap_u8(gray, m, ocv.INTER_NEAREST), g["gray_nearest"])
	np.testing.assert_array_equal(ocv.remap_u8(src, m, ocv.INTER_LINEAR), g["rgb_linear"])
	np.testing.assert_array_equal(ocv.remap_u8(src, m, ocv.INTER_NEAREST), g["rgb_nearest"])
	# per-channel arithmetic: channel 0 of the RGB result is the gray result
	np.testing.assert_array_equal(g["rgb_linear"][:, :, 0], g["gray_linear"])


def test_golden_transformer():
	g = golden("transformer")
	grid = odw.Grid(g["nodes"], 25)
	off = g["ring_offsets"]
	rings = [g["ring_points"][a:b] for a, b in zip(off[:-1], off[1:])]
	got = np.concatenate(odw.transform_rings(grid.transformer, rings))
	np.testing.assert_allclose(got, g["ring_warped"], rtol=0, atol=1e-9)
	size = tuple(int(v) for v in g["size"])
	np.testing.assert_array_equal(odw.dewarped(g["page"], grid), g["page_dewarped"])
	np.testing.assert_array_equal(
		odw.dewarp_image(size, g["labels"], grid, ocv.INTER_NEAREST), g["labels_dewarped"])


def test_transformer_inverts_the_grid():
	# forward map (warped -> dewarped) of the dense grid's own samples gives back their pixel position
	g = golden("transformer")
	grid = odw.Grid(g["nodes"], 25)
	full = grid.points_full()
	ys, xs = np.mgrid[30:full.shape[0] - 30:40, 30:full.shape[1] - 30:40]
	src = full[ys, xs].reshape(-1, 2).astype(np.float64)
	x, y = grid.transformer(tuple(src[:, 0]), tuple(src[:, 1]))
	err = np.hypot(x - xs.ravel(), y - ys.ravel())
	# piecewise-linear over Delaunay triangles vs bilinear over cells offset by half a cell
	assert np.median(err) < 0.5 and err.max() < 3


# ---- pins that exist in the build container: sympy (corner projection), SciPy 1.7.1 (Qhull diagonals) ------

def _sympy_corner_value(cx, cy, p1, p2):
	"""The corner value exactly as the reference computes it (origami/core/lingrid.py:115-128), through sympy."""
	import sympy.geometry
	line = sympy.geometry.Line(sympy.geometry.Point(*p1[:2]), sympy.geometry.Point(*p2[:2]))
	p = line.projection(sympy.geometry.Point(cx, cy))
	p = p.evalf()
	p = np.array([p.x, p.y], dtype=np.float64)
	d = np.linalg.norm(p - p1[:2])
	d_total = np.linalg.norm(p2[:2] - p1[:2])
	x = d / d_total
	return (1 - x) * p1[2:] + x * p2[2:]


def test_corner_projection_is_pinned_by_sympy():
	"""Box._add_corner of the product and BorderBox._corner of the oracle restate sympy's
	Line.projection(...).evalf() in closed form (exact rationals of the doubles).  sympy IS importable here
	(1.14; the reference pins 1.5.1), so both are checked against the call the reference makes.  sympy.Point
	turns floats into rationals through their 15-digit decimal form, so its projection is that of points moved
	by <= 5e-16 relative: the bar is 1e-12 relative on the corner value (observed ~1e-15), scalar and vector."""
	pytest.importorskip("sympy")
	from origami_amd import lingrid as plin
	rng = np.random.default_rng(77)
	worst = 0.0
	for k in range(12):
		W, H = rng.uniform(300, 8000, size=2)
		nv = 1 + k % 2
		# a point on the left border and one on the top border, as _nearest_to_corner hands them over
		p1 = np.hstack([[-1.0, rng.uniform(0, H)], rng.normal(0, 1 if nv == 1 else 3000, size=nv)])
		p2 = np.hstack([[rng.uniform(0, W), -1.0], rng.normal(0, 1 if nv == 1 else 3000, size=nv)])
		want = _sympy_corner_value(-1.0, -1.0, p1, p2)
		box = plin.Box(0, 0, W, H)
		box._add_corner(-1.0, -1.0, p1, p2)
		got_product = box._corners[(-1.0, -1.0)]
		obox = olin.BorderBox(0, 0, W, H)
		obox._corner(-1.0, -1.0, p1, p2)
		got_oracle = obox.corners[(-1.0, -1.0)]
		np.testing.assert_array_equal(got_product, got_oracle)
		scale = max(np.abs(p1[2:]).max(), np.abs(p2[2:]).max())
		err = np.abs(got_oracle - want).max() / scale
		worst = max(worst, err)
		assert err < 1e-12
	print("corner projection vs sympy: worst relative difference %.2e" % worst)


def test_qhull_diagonals_scipy_171_vs_current():
	"""tests/golden/scipy171.npz holds Qhull's triangulations of every Delaunay input of the golden fixtures as
	SciPy 1.7.1 builds them (generator: tests/golden/make_scipy171.py, run under /opt/conda/bin/python3.9; the
	reference pins 1.3.1).  The current SciPy must give the same hull, and the same triangles except where four
	nodes are cocircular to rounding — flipped diagonals are counted and printed; the interpolant across a
	flipped quad differs by O(curvature * cell^2), covered by the 1e-3 px bar of the transformer tests."""
	import scipy.spatial
	z = golden("scipy171")
	assert str(z["scipy_version"]) == "1.7.1"
	report = {}
	for name in ("field_n64", "grid_h", "grid_v", "transformer", "transformer_cfg0"):
		pts = z[name + "_points"]
		hull = scipy.spatial.ConvexHull(pts).vertices
		# same cyclic order up to the starting vertex
		old = list(z[name + "_hull"])
		assert sorted(hull) == sorted(old)
		k = old.index(hull[0])
		assert old[k:] + old[:k] == list(hull)
		# the extrapolation set is built from the hull: identical point sets under both versions
		fac = olin.Factory(list(pts), list(np.zeros(len(pts))), (pts[:, 0].min(), pts[:, 1].min(), pts[:, 0].max(), pts[:, 1].max())) \
			if name.startswith("transformer") else None
		if fac is not None:
			np.testing.assert_array_equal(fac.extra_pts, z[name + "_extra_points"])
		for part, p in (("inner", pts), ("extra", z[name + "_extra_points"])):
			new = {tuple(t) for t in np.sort(scipy.spatial.Delaunay(p).simplices, axis=1)}
			was = {tuple(t) for t in z["%s_%s" % (name, part)]}
			assert len(new) == len(was)
			report["%s_%s" % (name, part)] = (len(was - new), len(was))
	print("Qhull triangles that differ between SciPy 1.7.1 and %s: %s" % (scipy.__version__, report))
	# scattered flow samples are in general position: no flips there
	for key in ("field_n64_inner", "grid_h_inner", "grid_v_inner"):
		assert report[key][0] == 0, report
	# a marched lattice is nearly cocircular everywhere: flips may happen, but stay a small minority
	for key in ("transformer_inner", "transformer_cfg0_inner"):
		assert report[key][0] <= 0.02 * report[key][1], report


def test_ring_self_cross_reference_known_answers():
	"""oracle/rings.py (exact, O(n^2)) on hand-derived cases: simple, bow-tie, self-touching at a vertex,
	T-touch of a vertex on an edge, a spike (collinear overlap), the closing-segment adjacency."""
	from oracle import rings as orings
	square = [(0, 0), (4, 0), (4, 4), (0, 4), (0, 0)]
	bowtie = [(0, 0), (2, 2), (2, 0), (0, 2), (0, 0)]  # edges cross at (1, 1)
	touch = [(0, 0), (0, 2), (1, 1), (2, 2), (2, 0), (1, 1), (0, 0)]  # the shapely manual's bow-tie: vertex met twice
	t_touch = [(0, 0), (4, 0), (4, 4), (2, 0), (0, 4), (0, 0)]  # vertex (2, 0) lies on the first edge
	spike = [(0, 0), (4, 0), (4, 4), (4, 6), (4, 4), (0, 4), (0, 0)]  # out and back along the same line
	tri = [(0, 0), (1, 0), (0, 1), (0, 0)]
	assert not orings.ring_self_crosses(square)
	assert not orings.ring_self_crosses(tri)
	assert orings.ring_self_crosses(bowtie)
	assert orings.ring_self_crosses(touch)
	assert orings.ring_self_crosses(t_touch)
	assert not orings.ring_self_crosses(spike[:4] + spike[5:])
	# a spike visits (4, 4) twice: the segment arriving there first and the one returning from the tip are not
	# adjacent and share that point
	assert orings.ring_self_crosses(spike)
	# a crossing 1e-9 away from a vertex is still found (exact arithmetic)
	near = [(0, 0), (2, 2), (2, 0), (1e-9, 2), (0, 0)]
	assert orings.ring_self_crosses(near)


def _resize_exact(nodes, s):
	"""The rule of SURVEY Appendix A.1 evaluated in float64 by plain numpy (a third, independent statement)."""
	n = nodes.astype(np.float64)

	def taps(size):
		u = (np.arange(size * s) + 0.5) / s - 0.5
		i = np.floor(u).astype(int)
		f = u - i
		f = np.where((i < 0) | (i >= size - 1), 0.0, f)
		i = np.clip(i, 0, size - 1)
		return i, np.minimum(i + 1, size - 1), f

	h, w = n.shape[:2]
	ix, ix1, fx = taps(w)
	iy, iy1, fy = taps(h)
	rows = n[:, ix] * (1 - fx)[None, :, None] + n[:, ix1] * fx[None, :, None]
	return rows[iy] * (1 - fy)[:, None, None] + rows[iy1] * fy[:, None, None]


@pytest.mark.parametrize("shape,cell,span", [((12, 9), 8, 6), ((40, 33), 25, 3), ((40, 33), 25, 6), ((7, 131), 9, 6)])
def test_dense_grid_half_pixel_convention_agrees_with_torch_interpolate(shape, cell, span):
	"""Two INDEPENDENT implementations of the same alignment rule (they do not pin cv2: PARITY UNPINNED stays,
	DESIGN.md §2): torch.nn.functional.interpolate(mode="bilinear", align_corners=False) places output pixel X at
	source position (X + 0.5) / s - 0.5 and clamps at the edges — what SURVEY Appendix A.1 says of
	cv2.resize(INTER_LINEAR) and what oracle/cv.py restates —, and the same rule in float64 numpy.
	What the comparison can show is the convention, not the last bits: a two-pass float32 lerp rounds four times at
	the magnitude of the coordinate, and float32 resolves 4.9e-4 px between 4096 and 8192, so ANY two float32
	implementations differ by up to ~3 ulp there (measured: oracle - exact 1.2e-3 px, torch - exact 1.1e-3 px,
	oracle - torch 1.5e-3 px at x ~ 5700; 9e-5 px at x ~ 500).  The north star's 1e-3 px on the float grid is
	therefore two float32 steps on a broadsheet page: it separates conventions (a half-pixel shift is 0.5 / s of a
	cell, tens of pixels of difference between neighbouring nodes), not operation orders."""
	import torch
	import torch.nn.functional as F
	h, w = shape
	rng = np.random.default_rng(h * 100 + w)
	gy, gx = np.mgrid[0:h, 0:w].astype(np.float64)
	x = (gx - 1) * cell * span + 40 * np.sin(gy / 3.0) + rng.normal(0, 2, size=shape)
	y = (gy - 1) * cell * span + 40 * np.cos(gx / 4.0) + rng.normal(0, 2, size=shape)
	nodes = np.stack([x, y], axis=-1).astype(np.float32)
	got = ocv.dense_grid(nodes, cell)
	t = torch.from_numpy(nodes).permute(2, 0, 1).unsqueeze(0)  # (1, 2, h, w)
	tor = F.interpolate(t, size=(h * cell, w * cell), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
	exact = _resize_exact(nodes, cell)
	assert got.shape == tor.shape == exact.shape
	step = float(np.spacing(np.float32(np.abs(exact).max())))  # one float32 step at the largest coordinate
	assert np.abs(got - exact).max() <= 3 * step
	assert np.abs(tor - exact).max() <= 3 * step
	assert np.abs(got - tor).max() <= 4 * step
	if np.abs(exact).max() < 4096:  # pages up to 4096 px: every pair within the north star's 1e-3 px
		assert np.abs(got - tor).max() < 1e-3 and np.abs(got - exact).max() < 1e-3
	# a different convention would be off by whole fractions of a cell: align_corners=True differs by > 1 px here
	other = F.interpolate(t, size=(h * cell, w * cell), mode="bilinear", align_corners=True)[0].permute(1, 2, 0).numpy()
	assert np.abs(got - other).max() > 1.0
	# the structure the Transformer relies on (SURVEY §7 hard part 2): node g sits on pixel g * s + (s - 1) / 2 when
	# s is odd, and the first / last s / 2 pixels are flat
	if cell % 2:
		c = (cell - 1) // 2
		assert np.array_equal(got[c::cell, c::cell], nodes)
	assert np.array_equal(got[0, 0], nodes[0, 0]) and np.array_equal(got[-1, -1], nodes[-1, -1])


def test_remap_convention_against_scipy_map_coordinates():
	"""An INDEPENDENT implementation of what cv2.remap(INTER_LINEAR, BORDER_CONSTANT 0) means — it does not pin cv2
	(PARITY UNPINNED stays): scipy.ndimage.map_coordinates(order=1, mode="grid-constant") samples the image at
	(y, x) in pixel-index coordinates with zeros outside, in float64.  cv2's arithmetic differs in the last bits by
	design — coordinates rounded to 1/32 px, 15-bit weights — so on an image whose neighbouring pixels differ by up to
	g grey levels the two may differ by g / 64 + 1; what the comparison shows is the convention: sample positions at
	pixel centres named by integers, zeros (not replicated edges) outside.  A map shifted by half a pixel, the usual
	way to get this wrong, is an order of magnitude further off."""
	import scipy.ndimage
	rng = np.random.default_rng(5)
	H, W = 240, 320
	yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
	smooth = 128 + 60 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + 40 * np.sin((xx + yy) / 23.0)
	img = np.clip(smooth + rng.normal(0, 2, size=(H, W)), 0, 255).astype(np.uint8)
	Hd, Wd = 260, 350  # the map reaches past every edge of the image
	my, mx = np.mgrid[0:Hd, 0:Wd].astype(np.float64)
	x = (mx - 12) * 0.97 + 6 * np.sin(my / 40.0) + rng.uniform(0, 1, size=(Hd, Wd))
	y = (my - 9) * 1.02 + 5 * np.cos(mx / 55.0) + rng.uniform(0, 1, size=(Hd, Wd))
	dense = np.stack([x, y], axis=-1).astype(np.float32)
	got = ocv.remap_u8(img, dense).astype(np.float64)
	fx, fy = dense[:, :, 0].astype(np.float64), dense[:, :, 1].astype(np.float64)

	def sampled(dx, dy):
		return scipy.ndimage.map_coordinates(img.astype(np.float64), [fy + dy, fx + dx], order=1, mode="grid-constant", cval=0.0)

	ref = sampled(0.0, 0.0)
	# the largest step between neighbouring pixels bounds what 1/64 px of coordinate rounding can do
	g = max(np.abs(np.diff(img.astype(np.int32), axis=0)).max(), np.abs(np.diff(img.astype(np.int32), axis=1)).max())
	inside = (fx >= 0) & (fx <= W - 1) & (fy >= 0) & (fy <= H - 1)
	diff = np.abs(got - ref)
	assert diff[inside].max() <= 2 * g / 64.0 + 1.0  # x and y each contribute up to g / 64
	assert diff[inside].mean() < 0.5
	# outside the image: zeros, fading in over the last pixel (an edge-replicating border would give ~128 there)
	far = (fx < -1) | (fx > W) | (fy < -1) | (fy > H)
	assert far.any() and got[far].max() == 0 and ref[far].max() == 0
	rim = ~inside & ~far
	assert rim.any() and np.abs(got[rim] - ref[rim]).max() <= 255 / 32.0 + 1.0  # the step to the zero border is up to 255
	# the wrong convention is far away: half a pixel off in either axis
	for dx, dy in ((0.5, 0.0), (0.0, 0.5), (-0.5, -0.5)):
		assert np.abs(got - sampled(dx, dy))[inside].mean() > 5 * diff[inside].mean()
