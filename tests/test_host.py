"""Host-side logic of the product (no GPU): artifact I/O, WKT geometry, and the extrapolation point set
of lingrid, which is built on the host and only evaluated on the device."""

import io
import json
import zipfile

import numpy as np
import pytest

from oracle import lingrid as olin
from origami_amd import _shim, geometry, synth
from origami_amd.dewarp import Geometry, Grid, Samples
from origami_amd.lingrid import InterpolatorFactory


# ---- artifacts (ref: origami/core/dewarp.py:43-75, 537-553; docs/formats.md) ---------------------------

def test_samples_round_trip_and_format():
	s = Samples(Geometry(2400, 3400))
	s.append(np.array([10.0, 20.0]), 0.01)
	s.extend([np.array([30.5, 40.25])], [-0.02])
	buf = io.BytesIO()
	with zipfile.ZipFile(buf, "w") as zf:
		s.save(zf, "h")
		Samples(Geometry(2400, 3400)).save(zf, "v")
	with zipfile.ZipFile(buf, "r") as zf:
		assert sorted(zf.namelist()) == ["h.json", "h.npy", "v.json", "v.npy"]
		assert json.loads(zf.read("h.json")) == {"version": 1, "size": [2400, 3400]}
		arr = np.load(io.BytesIO(zf.read("h.npy")))
		assert arr.dtype == np.float64 and arr.shape == (2, 3)
		np.testing.assert_array_equal(arr, [[10, 20, 0.01], [30.5, 40.25, -0.02]])
		assert np.load(io.BytesIO(zf.read("v.npy"))).shape == (3, 0)  # empty case, dewarp.py:65-66
		h = Samples.open(zf, "h")
		v = Samples.open(zf, "v")
	assert len(h) == 2 and h.geometry.size == (2400, 3400)
	np.testing.assert_array_equal(np.array(h.points), [[10, 20], [30.5, 40.25]])
	assert h.values == [0.01, -0.02]
	assert len(v.values) == 0 and np.asarray(v.points).shape == (2, 0)  # dewarp.py:55-57


def test_grid_save_open_round_trip(tmp_path):
	nodes = synth.Warp((600, 800)).node_grid(25)
	grid = Grid(nodes, 25)
	path = tmp_path / "dewarp.zip"
	grid.save(path)
	with zipfile.ZipFile(path) as zf:
		assert zf.namelist() == ["data.npy", "meta.json"]  # data first, dewarp.py:552-553
		meta = json.loads(zf.read("meta.json"))
		assert meta == {"version": 1, "cell": 25, "shape": list(nodes.shape)}
		data = np.load(io.BytesIO(zf.read("data.npy")))
		assert data.dtype == np.float32
		assert all(i.compress_type == zipfile.ZIP_DEFLATED for i in zf.infolist())
	again = Grid.open(path)
	assert again.resolution == 25
	np.testing.assert_array_equal(again.points("sample"), nodes)
	h, w = nodes.shape[:2]
	assert again.geometry.size == (w * 25, h * 25)
	assert Geometry(300, 400).rel_area(0.01) == pytest.approx((500 * 0.01) ** 2)


# ---- lingrid host part against the oracle ---------------------------------------------------------------

@pytest.mark.parametrize("n,vector", [(5, False), (64, False), (300, True)])
def test_extrapolation_set_equals_oracle(n, vector):
	rng = np.random.default_rng(n)
	pts = rng.uniform([0, 0], [600, 800], size=(n, 2))
	if vector:
		vals = np.stack([pts[:, 0] * 0.9 + 3, pts[:, 1] * 1.1 - 2], axis=1)
		values = vals
	else:
		values = list(0.03 * np.sin(pts[:, 0] / 100))
	bounds = (0, 0, 600, 800)
	want = olin.Factory(list(pts), values, bounds)
	got = InterpolatorFactory(list(pts), values, bounds)
	assert got.is_collinear == want.is_collinear
	np.testing.assert_array_equal(got.extra_points, want.extra_pts)
	np.testing.assert_array_equal(got.extra_values, want.extra_val)


def test_extrapolation_set_collinear_and_duplicates():
	pts = [np.array([100.0, 100.0]), np.array([200.0, 150.0]), np.array([300.0, 200.0])]
	vals = [0.01, 0.02, 0.04]
	want = olin.Factory(pts, vals, (0, 0, 600, 800))
	got = InterpolatorFactory(pts, vals, (0, 0, 600, 800))
	assert got.is_collinear and want.is_collinear
	np.testing.assert_array_equal(got.extra_points, want.extra_pts)
	np.testing.assert_array_equal(got.extra_values, want.extra_val)
	# coincident samples: the value of the last one wins (dict build, lingrid.py:217-218)
	rng = np.random.default_rng(0)
	p = rng.uniform(0, 500, size=(20, 2))
	p = np.vstack([p, p[:3]])
	v = list(np.arange(23, dtype=np.float64))
	want = olin.Factory(list(p), v, (0, 0, 500, 500))
	got = InterpolatorFactory(list(p), v, (0, 0, 500, 500))
	np.testing.assert_array_equal(got.extra_values, want.extra_val)


def test_lininterp_argument_errors():
	with pytest.raises(ValueError, match="not enough values given."):
		InterpolatorFactory([np.zeros(2)], [1.0], (0, 0, 10, 10))
	with pytest.raises(ValueError, match="outside given domain"):
		InterpolatorFactory(
			[np.array([0.0, 0.0]), np.array([50.0, 5.0]), np.array([20.0, 40.0])], [1.0, 2.0, 3.0], (10, 10, 60, 60))


# ---- WKT / geometry (ref: origami/batch/detect/dewarp.py:23-39) ----------------------------------------

def test_wkt_round_trip_and_number_format():
	text = "POLYGON ((0 0, 10.5 0, 10.5 7.25, 0 7.25, 0 0), (2 2, 3 2, 3 3, 2 2))"
	g = geometry.parse_wkt(text)
	assert g.geom_type == "Polygon" and len(g.rings) == 2
	assert g.wkt == text
	assert g.area == pytest.approx(10.5 * 7.25 - 0.5)
	ls = geometry.parse_wkt("LINESTRING (1.5 2, 3 4.125, 1e3 -2.5e-3)")
	assert ls.geom_type == "LineString" and ls.area == 0.0
	assert ls.wkt == "LINESTRING (1.5 2, 3 4.125, 1000 -0.0025)"
	assert geometry.format_number(0.1 + 0.2) == "0.3"  # 16 significant digits, trimmed
	assert geometry.format_number(1 / 3) == "0.3333333333333333"
	assert geometry.format_number(4687.500000000001) == "4687.500000000001"
	assert geometry.parse_wkt("POLYGON EMPTY").is_empty


def test_native_wkt_writer_equals_python_formatting():
	"""ogd_wkt_ring_text (host code in libogdewarp, no GPU needed) against "%.16g" % v."""
	rng = np.random.default_rng(0)
	rings = [rng.normal(0, 3000, (n, 2)) for n in (5, 1, 0, 400)]
	rings.append(np.array([[1.0, 2.5], [1e-7, -3e22], [np.nan, np.inf], [-np.inf, 0.1], [123456789.123456789, 1 / 3]]))
	rings.append(np.round(rng.normal(0, 1000, (30, 2))))  # integer coordinates print without a fraction
	rings.append(rng.normal(0, 1, (50, 2)) * 10.0 ** rng.integers(-12, 12, (50, 2)))
	assert _shim.wkt_ring_texts(rings) == [geometry._ring_text(r) for r in rings]
	assert _shim.wkt_ring_texts([]) == []
	geoms = [
		geometry.parse_wkt("POLYGON ((0 0, 10.5 0, 10.5 7.25, 0 7.25, 0 0), (2 2, 3 2, 3 3, 2 2))"),
		geometry.parse_wkt("POLYGON EMPTY"), geometry.parse_wkt("LINESTRING (1.5 2, 3 4.125, 1e3 -2.5e-3)"),
		geometry.Geom("Polygon", [rings[3]])]
	assert geometry.to_wkt_many(geoms) == [g.wkt for g in geoms]
	# parse: numpy's text parser and the regex path agree, also on nan / inf spellings and 3-D input
	g = geometry.parse_wkt("LINESTRING (nan 1, inf -2, -inf 3e4)")
	assert np.isnan(g.rings[0][0, 0]) and g.rings[0][1, 0] == np.inf and g.rings[0][2].tolist() == [-np.inf, 3e4]
	z = geometry.parse_wkt("POLYGON Z ((0 0 1, 1 0 1, 1 1 1, 0 0 1))")
	assert z.rings[0].tolist() == [[0, 0], [1, 0], [1, 1], [0, 0]]
	with pytest.raises(ValueError):
		geometry.parse_wkt("LINESTRING (1 2, 3)")


def test_validity_and_transform():
	square = geometry.parse_wkt("POLYGON ((0 0, 4 0, 4 4, 0 4, 0 0))")
	bowtie = geometry.parse_wkt("POLYGON ((0 0, 4 4, 4 0, 0 4, 0 0))")
	assert square.is_valid and not bowtie.is_valid
	moved = square.transform(lambda x, y: (np.asarray(x) + 1, np.asarray(y) * 2))
	assert moved.wkt == "POLYGON ((1 0, 5 0, 5 8, 1 8, 1 0))"
	assert moved.area == 32
	with pytest.raises(ValueError):
		geometry.parse_wkt("MULTILINESTRING ((0 0, 1 0))")
	# a repaired contour may come back from contours.1.zip as a MultiPolygon (ref: batch/detect/dewarp.py:36-39)
	multi = geometry.parse_wkt(
		"MULTIPOLYGON (((0 0, 0 4, 4 4, 4 0, 0 0), (1 1, 2 1, 2 2, 1 1)), ((10 10, 10 12, 12 12, 10 10)))")
	assert [len(p) for p in multi.parts] == [2, 1] and len(multi.rings) == 3
	assert multi.area == 16 - 0.5 + 2 and multi.is_valid
	assert multi.wkt == "MULTIPOLYGON (((0 0, 0 4, 4 4, 4 0, 0 0), (1 1, 2 1, 2 2, 1 1)), ((10 10, 10 12, 12 12, 10 10)))"
	assert geometry.to_wkt_many([multi, square]) == [multi.wkt, square.wkt]
	moved = multi.transform(lambda x, y: (np.asarray(x) + 1, np.asarray(y)))
	assert [len(p) for p in moved.parts] == [2, 1] and moved.rings[2][0].tolist() == [11, 10]


def test_synthetic_page_dir_layout(tmp_path):
	rng = np.random.default_rng(5)
	page = synth.write_page_dir(tmp_path, "page0001", rng, size=(300, 400), n_h=20, n_v=30, n_regions=5, n_separators=2, label_size=(160, 208))
	out = page.with_suffix(".out")
	assert sorted(p.name for p in out.iterdir()) == ["contours.0.zip", "flow.zip", "segment.zip"]
	with zipfile.ZipFile(out / "contours.0.zip") as zf:
		names = zf.namelist()
		assert "meta.json" in names and "separators/V/meta.json" in names
		assert sum(n.startswith("regions/TEXT/") for n in names) == 5
		g = geometry.parse_wkt(zf.read("regions/TEXT/0.wkt").decode("utf8"))
		assert g.geom_type == "Polygon" and g.area > 0


# ---- N1: line lattices (host side) ------------------------------------------------------------------

def test_oracle_line_lattice_identity_grid():
	"""On an identity grid the warped lattice IS the dewarped lattice, and the image is the flipped crop."""
	from oracle import block as oblock
	from oracle import dewarp as odw
	s, h, w = 10, 8, 12
	gy, gx = np.mgrid[0:h, 0:w].astype(np.float32)
	# cv2.resize pixel-centre alignment: nodes at (k + 0.5) * s - 0.5 give map(x, y) = (x, y) inside
	nodes = np.stack([(gx + 0.5) * s - 0.5, (gy + 0.5) * s - 0.5], axis=-1).astype(np.float32)
	grid = odw.Grid(nodes, s)
	wg = oblock.warped_grid(grid, (20, 50), (40, 0), (0, -16), xsteps=41, ysteps=17)
	assert wg.shape == (17, 41, 2) and wg.dtype == np.float32
	np.testing.assert_allclose(wg[0, :, 0], 20 + np.arange(41), atol=1e-4)
	np.testing.assert_allclose(wg[:, 0, 1], 50 - np.arange(17), atol=1e-4)
	rng = np.random.default_rng(0)
	page = rng.integers(0, 256, size=(h * s, w * s), dtype=np.uint8)
	im = oblock.dewarped_image(page, grid, (20, 50), (40, 0), (0, -16), target_height=17)
	assert im.shape == (17, 41)
	np.testing.assert_array_equal(im, page[34:51, 20:61])
	# steps chosen from the rough lattice: ceil(|dx| = 40) columns (41 if float32 noise lands above 40)
	hh, ww, _ = oblock.warped_grid(grid, (20, 50), (40, 0), (0, -16)).shape
	assert hh in (16, 17) and ww in (40, 41)


def test_line_position_with_column():
	from origami_amd import block as blk
	ln = blk.Line(None, (10.0, 100.0), (200.0, 4.0), (0.0, -30.0))
	p0, right, up, xres = ln._position(1, None)
	assert p0.tolist() == [10.0, 100.0] and right.tolist() == [200.0, 4.0] and xres == 1
	p0, right, up, xres = ln._position(1, (60.0, 110.0))
	np.testing.assert_allclose(p0, [60.0, 101.0])
	np.testing.assert_allclose(right, [50.0, 1.0])
	assert xres == pytest.approx(0.25)
	p0, right, up, xres = ln._position(2, (None, 110.0))
	np.testing.assert_allclose(p0, [10.0, 100.0])
	np.testing.assert_allclose(right, [100.0, 2.0])
	assert xres == pytest.approx(1.0)
	# a column outside the baseline leaves that end alone (intersect_segments default)
	p0, right, up, xres = ln._position(1, (-50.0, 400.0))
	np.testing.assert_allclose(p0, [10.0, 100.0])
	np.testing.assert_allclose(right, [200.0, 4.0])


# ---- N3: lattice Delaunay (host code of the library, no device) ---------------------------------------

def _warped_lattice(rng, h, w, s=25, amp=6.0):
	gy, gx = np.mgrid[0:h, 0:w].astype(np.float64)
	x = (gx - 1) * s + amp * np.sin(gy / 3.0) + rng.normal(0, 0.5, size=(h, w))
	y = (gy - 1) * s + amp * np.cos(gx / 4.0) + rng.normal(0, 0.5, size=(h, w))
	return np.stack([x, y], axis=-1).astype(np.float32)


def _tri_set(simplices):
	return set(map(tuple, np.sort(simplices, axis=1).tolist()))


@pytest.mark.parametrize("shape,mirror", [((12, 9), False), ((2, 2), False), ((2, 40), False), ((60, 45), False), ((20, 15), True)])
def test_lattice_delaunay_equals_qhull(shape, mirror):
	"""ogd_lattice_delaunay (known topology + exact Lawson flips) gives Qhull's triangulation, hull and
	barycentric transforms on warped lattices (ref: dewarp.py:143-161 via lingrid.py:208-221, 286-290)."""
	import scipy.spatial
	from origami_amd import _shim
	rng = np.random.default_rng(shape[0] * 100 + shape[1])
	nodes = _warped_lattice(rng, *shape)
	if mirror:
		nodes = np.ascontiguousarray(nodes[:, ::-1])
	pts = nodes.reshape(-1, 2).astype(np.float64)
	simplices, hull, transform = _shim.lattice_delaunay(nodes)
	ref = scipy.spatial.Delaunay(pts)
	assert _tri_set(simplices) == _tri_set(ref.simplices)
	hv = list(scipy.spatial.ConvexHull(pts).vertices)
	k = hv.index(hull[0])
	assert hv[k:] + hv[:k] == list(hull)  # same vertices, same (counter-clockwise) cyclic order
	# transforms: same barycentric coordinates per vertex for points around each triangle
	lut = {tuple(sorted(s)): i for i, s in enumerate(ref.simplices.tolist())}
	for i, s in enumerate(simplices.tolist()):
		j = lut[tuple(sorted(s))]
		x = pts[s].mean(axis=0) + rng.normal(0, 3, 2)

		def bary(T, verts):
			c = T[:2] @ (x - T[2])
			return dict(zip(verts, [c[0], c[1], 1 - c[0] - c[1]]))
		a, b = bary(transform[i], s), bary(ref.transform[j], ref.simplices[j].tolist())
		assert max(abs(a[v] - b[v]) for v in s) < 1e-9  # slivers in hull pockets are ill-conditioned


def test_lattice_delaunay_concave_outline_and_collinear_edges():
	"""Pockets between a bowed outline and its hull are filled; outline nodes exactly on a hull edge split them."""
	import scipy.spatial
	from origami_amd import _shim
	h, w = 9, 11
	gy, gx = np.mgrid[0:h, 0:w].astype(np.float64)
	x = gx * 16 + 3 * np.sin(np.pi * gy / (h - 1))  # left side bows inwards, right side outwards
	y = gy * 16 - 5 * np.sin(np.pi * gx / (w - 1)) * (gy == 0) + 4 * np.sin(np.pi * gx / (w - 1)) * (gy == h - 1)
	nodes = (np.round(np.stack([x, y], axis=-1) * 64) / 64).astype(np.float32)  # sin(pi) leaves 1e-16 otherwise
	nodes[h - 1, :, 1] = nodes[h - 1, 0, 1]  # bottom row exactly straight: its nodes lie ON the hull edge
	pts = nodes.reshape(-1, 2).astype(np.float64)
	simplices, hull, _ = _shim.lattice_delaunay(nodes)
	ref = scipy.spatial.Delaunay(pts)
	def area(s):
		u, v = pts[s[:, 1]] - pts[s[:, 0]], pts[s[:, 2]] - pts[s[:, 0]]
		return 0.5 * np.abs(u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]).sum()
	assert area(simplices) == pytest.approx(scipy.spatial.ConvexHull(pts).volume, rel=1e-12)
	assert len(simplices) > 2 * (h - 1) * (w - 1)  # pockets were added
	ours, theirs = _tri_set(simplices), _tri_set(ref.simplices)
	# the regular interior is all cocircular quads (either diagonal is Delaunay): compare where unique
	cocircular_ok = all(len(set(t)) == 3 for t in ours ^ theirs)
	assert cocircular_ok and len(ours) == len(theirs)
	assert set(hull) == set(scipy.spatial.ConvexHull(pts).vertices)


def test_lattice_delaunay_rejects_what_it_cannot_do():
	from origami_amd import _shim
	rng = np.random.default_rng(3)
	nodes = _warped_lattice(rng, 8, 8)
	folded = nodes.copy()
	folded[3, 3] = folded[5, 5]  # coincident / folded cell
	with pytest.raises(_shim.OgdError) as e:
		_shim.lattice_delaunay(folded)
	assert e.value.status == _shim.ERR_UNSUPPORTED
	inexact = nodes.astype(np.float64)
	inexact[2, 2, 0] += 2.0 ** -45  # not a multiple of 2^-40
	with pytest.raises(_shim.OgdError) as e:
		_shim.lattice_delaunay(inexact)
	assert e.value.status == _shim.ERR_UNSUPPORTED
	huge = nodes.astype(np.float64)
	huge[0, 0, 0] = -3e6
	with pytest.raises(_shim.OgdError) as e:
		_shim.lattice_delaunay(huge)
	assert e.value.status == _shim.ERR_UNSUPPORTED


# ---- ogd_delaunay_points: scattered points without Qhull (host) ---------------------------------------------------

def _field_like_points(rng, n):
	"""Samples as the flow stage writes them: points strung along text lines and separators."""
	lines = rng.uniform([200, 200], [5800, 7800], size=(n // 10 + 1, 2))
	t = rng.uniform(-300, 300, size=(len(lines), 10, 1))
	d = np.stack([np.cos(rng.normal(0, 0.03, len(lines))), np.sin(rng.normal(0, 0.03, len(lines)))], axis=-1)
	return (lines[:, None, :] + t * d[:, None, :]).reshape(-1, 2)[:n]


def _extrapolation_set(rng, shape):
	nodes = _warped_lattice(rng, *shape).reshape(-1, 2).astype(np.float64)
	vals = list(rng.uniform(0, 1, len(nodes)))
	bounds = (nodes[:, 0].min(), nodes[:, 1].min(), nodes[:, 0].max(), nodes[:, 1].max())
	return np.asarray(olin.Factory(list(nodes), vals, bounds).extra_pts, dtype=np.float64)


@pytest.mark.parametrize("kind,n", [
	("uniform", 3), ("uniform", 4), ("uniform", 7), ("uniform", 500), ("uniform", 6000), ("clustered", 3000),
	("field", 1500), ("extrapolation", (12, 9)), ("extrapolation", (60, 45)), ("extrapolation", (2, 2))])
def test_delaunay_points_equals_qhull(kind, n):
	"""ogd_delaunay_points gives the simplices and barycentric transforms scipy.spatial.Delaunay (Qhull) gives for the
	point sets the stage triangulates (ref: lingrid.py:286-290): samples, and extrapolation sets whose points
	sit exactly on the four box sides."""
	import scipy.spatial
	from origami_amd import _shim
	rng = np.random.default_rng(len(kind) * 1000 + (n if isinstance(n, int) else n[0]))
	if kind == "uniform":
		pts = rng.uniform([0, 0], [6000, 8000], size=(n, 2))
	elif kind == "clustered":
		pts = rng.normal(0, 1, size=(n, 2)) * rng.choice([1.0, 30.0, 900.0], size=(n, 1)) + 3000
	elif kind == "field":
		pts = _field_like_points(rng, n)
	else:
		pts = _extrapolation_set(rng, n)
	simplices, transform, ties = _shim.delaunay_points(pts)
	ref = scipy.spatial.Delaunay(pts)
	assert ties == 0
	ours, theirs = _tri_set(simplices), _tri_set(ref.simplices)
	# Qhull works with a round-off tolerance: where samples strung along one line happen to lie on the hull it
	# fans its own slivers (area ~ 1e-12 px^2) between that line and the hull edge, where the exact triangulation
	# has the Delaunay ones.  Either way their transforms are NaN by scipy's rcond rule (below): no query ever
	# lands in one, the interpolant is the same.
	extra = np.array(sorted(ours ^ theirs), dtype=np.int64).reshape(-1, 3)
	eu, ev = pts[extra[:, 1]] - pts[extra[:, 0]], pts[extra[:, 2]] - pts[extra[:, 0]]
	assert np.all(np.abs(eu[:, 0] * ev[:, 1] - eu[:, 1] * ev[:, 0]) < 1e-8)
	for i, sx in enumerate(ref.simplices.tolist()):
		if tuple(sorted(sx)) not in ours:
			assert np.isnan(ref.transform[i][:2]).all()
	assert len(extra) == 0 or kind == "field"
	u, v = pts[simplices[:, 1]] - pts[simplices[:, 0]], pts[simplices[:, 2]] - pts[simplices[:, 0]]
	assert np.all(u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0] > 0)  # positively oriented
	lut = {tuple(sorted(s)): i for i, s in enumerate(ref.simplices.tolist())}
	for i, s in enumerate(simplices.tolist()):
		if tuple(sorted(s)) not in lut:
			assert np.isnan(transform[i][:2]).all()
			continue
		if i % max(1, len(simplices) // 400):
			continue
		j = lut[tuple(sorted(s))]
		x = pts[s].mean(axis=0)

		def bary(T, verts):
			c = T[:2] @ (x - T[2])
			return dict(zip(verts, [c[0], c[1], 1 - c[0] - c[1]]))
		a, b = bary(transform[i], s), bary(ref.transform[j], ref.simplices[j].tolist())
		if np.isnan(ref.transform[j]).any():
			assert np.isnan(transform[i][:2]).all()  # same rcond rule for slivers
		else:
			assert max(abs(a[k] - b[k]) for k in s) < 1e-9


def _exact_incircle(a, b, c, d):
	from fractions import Fraction as F
	rows = []
	for p in (a, b, c):
		x, y = F(float(p[0])) - F(float(d[0])), F(float(p[1])) - F(float(d[1]))
		rows.append((x, y, x * x + y * y))
	(a0, a1, a2), (b0, b1, b2), (c0, c1, c2) = rows
	return a0 * (b1 * c2 - b2 * c1) - a1 * (b0 * c2 - b2 * c0) + a2 * (b0 * c1 - b1 * c0)


def test_delaunay_points_is_exact_on_nearly_cocircular_points():
	"""Points on a circle moved by single ulps, plus nearly collinear triples: every interior edge of the result is
	locally Delaunay in rational arithmetic, and the reported ties are exactly the cocircular edges."""
	from origami_amd import _shim
	rng = np.random.default_rng(5)
	ang = np.sort(rng.uniform(0, 2 * np.pi, 40))
	ring = 1000 + 300 * np.stack([np.cos(ang), np.sin(ang)], axis=-1)
	ring = np.nextafter(ring, ring + rng.choice([-1.0, 1.0], ring.shape))
	line = np.stack([np.linspace(100, 1900, 30), np.linspace(100, 1900, 30) * 0.5 + 7], axis=-1)
	line[::3] = np.nextafter(line[::3], 0)
	square = np.array([[400.0, 400.0], [400.0, 464.0], [464.0, 400.0], [464.0, 464.0]])  # one exact tie
	pts = np.vstack([ring, line, square, [[1000.0, 1000.0]]])
	simplices, _, ties = _shim.delaunay_points(pts)
	edges = {}
	for t in simplices.tolist():
		for k in range(3):
			edges.setdefault(frozenset((t[k], t[(k + 1) % 3])), []).append(t)
	exact_ties = 0
	for e, ts in edges.items():
		if len(ts) != 2:
			continue
		t, u = ts
		d = [v for v in u if v not in e][0]
		s = _exact_incircle(pts[t[0]], pts[t[1]], pts[t[2]], pts[d])
		assert s <= 0, "edge %s is not locally Delaunay" % sorted(e)
		exact_ties += s == 0
	assert ties == exact_ties >= 1
	import scipy.spatial
	u, v = pts[simplices[:, 1]] - pts[simplices[:, 0]], pts[simplices[:, 2]] - pts[simplices[:, 0]]
	assert 0.5 * np.sum(u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]) == pytest.approx(scipy.spatial.ConvexHull(pts).volume, rel=1e-12)


def test_delaunay_points_reports_ties_and_rejects_what_qhull_must_decide():
	from origami_amd import _shim
	g = np.stack(np.meshgrid(np.arange(12.0), np.arange(9.0)), axis=-1).reshape(-1, 2)
	simplices, _, ties = _shim.delaunay_points(g)
	assert len(simplices) == 2 * 11 * 8 and ties == 11 * 8  # every unit square is a tie
	huge = np.array([[0.0, 0.0], [1e80, 0.0], [0.0, 1.0], [1.0, 1.0]])
	for bad in (g[:2], np.vstack([g, g[5:6]]), np.stack([np.arange(5.0), 2 * np.arange(5.0)], axis=-1), huge, huge * 1e-150):
		with pytest.raises(_shim.OgdError) as e:
			_shim.delaunay_points(bad)
		assert e.value.status == _shim.ERR_UNSUPPORTED
	with pytest.raises(_shim.OgdError) as e:
		_shim.delaunay_points(np.array([[0.0, 0.0], [1.0, np.nan], [2.0, 5.0]]))
	assert e.value.status != _shim.ERR_UNSUPPORTED


def test_delaunay_points_merges_coincident_points_on_request():
	"""The extrapolation set of a lattice holds every hull vertex's two border projections, which can coincide;
	ogd_delaunay_points then triangulates the first of each group and names it as the twin of the others."""
	import scipy.spatial
	from origami_amd import _shim
	rng = np.random.default_rng(11)
	pts = rng.uniform(0, 1000, size=(300, 2))
	pts = np.vstack([pts, pts[[7, 7, 120]]])  # 300, 301 = 7; 302 = 120
	simplices, _, ties, twin = _shim.delaunay_points(pts, twins=True)
	assert ties == 0
	assert twin[300] == 7 and twin[301] == 7 and twin[302] == 120 and np.all(twin[:300] == -1)
	assert not np.isin(simplices, [300, 301, 302]).any()
	assert _tri_set(simplices) == _tri_set(scipy.spatial.Delaunay(pts[:300]).simplices)


# ---- ring tables: the per-page batch forms of parse_wkt / ring_area_signed / _ring_is_valid ------------------------

def test_parse_wkt_many_equals_parse_wkt():
	from origami_amd.geometry import parse_wkt, parse_wkt_many, to_wkt
	rng = np.random.default_rng(8)
	texts = [
		"POLYGON ((0 0, 10 0, 10 10, 0 0), (1 1, 2 1, 2 2, 1 1))", " linestring (1.5 2e3, -3 4)\n", "POLYGON EMPTY",
		"LINESTRING EMPTY", "MULTIPOLYGON (((0 0, 1 0, 1 1, 0 0)), ((5 5, 6 5, 6 6, 5 5), (5.1 5.1, 5.2 5.1, 5.2 5.2, 5.1 5.1)))",
		"POLYGON Z ((0 0 1, 1 0 1, 1 1 1, 0 0 1))", "LINESTRING (nan inf, -inf 1)", "POLYGON((0 0,1 0,1 1,0 0))"]
	for ring in synth.polygon_rings(rng, (6000, 8000), 50, mu=20):
		texts.append(to_wkt(geometry.Geom("Polygon", [ring])))
	got = parse_wkt_many([t.encode("utf8") for t in texts])
	for text, g in zip(texts, got):
		w = parse_wkt(text)
		assert g.geom_type == w.geom_type and len(g.rings) == len(w.rings) and len(g.parts) == len(w.parts)
		for a, b in zip(g.rings, w.rings):
			np.testing.assert_array_equal(a, b)  # both read numbers as strtod does
		assert [len(p) for p in g.parts] == [len(p) for p in w.parts]
	# one text the batch reader does not take sends the whole batch through parse_wkt, which raises as before
	with pytest.raises(ValueError):
		parse_wkt_many([b"POLYGON ((0 0, 1 0, 1 1, 0 0))", b"POINT (1 2)"])
	assert parse_wkt_many([]) == []


def test_ring_tables_equal_the_per_ring_functions():
	from origami_amd import _shim, geometry
	rng = np.random.default_rng(9)
	rings = synth.polygon_rings(rng, (3000, 4000), 60, mu=25, valid=False)
	rings.append(np.array([[0.0, 0.0], [4.0, 0.0], [4.0, 0.0], [4.0, 4.0], [4.0, 4.0], [0.0, 0.0]]))  # repeats
	rings.append(np.array([[0.0, 0.0], [4.0, 4.0], [4.0, 0.0], [0.0, 4.0], [0.0, 0.0]]))  # bow-tie
	rings.append(np.array([[0.0, 0.0], [4.0, 0.0], [4.0, 4.0], [1.0, 1.0]]))  # open
	rings.append(np.array([[0.0, 0.0], [4.0, 0.0], [np.nan, 4.0], [0.0, 0.0]]))
	rings.append(np.array([[0.0, 0.0], [4.0, 0.0], [0.0, 0.0]]))  # too short
	rings.append(np.array([[0.0, 0.0], [2.0, 0.0], [4.0, 0.0], [0.0, 0.0]]))  # no area
	rings.append(np.array([[1.0, 1.0], [1.0, 1.0], [1.0, 1.0], [1.0, 1.0]]))
	rings.append(np.empty((0, 2)))
	flat, offsets = geometry.ring_table(rings)
	np.testing.assert_array_equal(
		_shim.rings_area_signed(flat, offsets), [geometry.ring_area_signed(r) for r in rings])
	d_flat, d_offsets = geometry.without_repeats_table(flat, offsets)
	for k, r in enumerate(rings):
		np.testing.assert_array_equal(d_flat[d_offsets[k]:d_offsets[k + 1]], geometry.without_repeats(r))
	crossing = [len(r) >= 4 and geometry._segments_cross(geometry.without_repeats(r)) for r in rings]
	valid, finite, distinct = geometry.rings_valid_table(flat, offsets, crossing=crossing)
	assert valid.tolist() == [bool(geometry._ring_is_valid(r)) for r in rings]
	assert valid.sum() > 10 and (~valid).sum() >= 8
	assert finite.tolist() == [bool(np.all(np.isfinite(r))) for r in rings]
	assert distinct.tolist() == [len(geometry.without_repeats(r)) for r in rings]


def test_wkt_numbers_equal_printf_16g():
	"""ogd_wkt_ring_text formats coordinates of page magnitude without printf (128-bit integer rounding of the exact
	decimal expansion); every number must read as C's "%.16g" does, which is what GEOS' WKTWriter prints
	(ref: origami/batch/detect/dewarp.py:39).  Python's % formatting is correctly rounded like glibc's."""
	rng = np.random.default_rng(12)
	x = np.arange(1, 50001, dtype=np.float64)
	cases = np.concatenate([
		rng.uniform(0, 8000, 200000), 10.0 ** rng.uniform(-7, 17, 200000) * rng.choice([-1, 1], 200000),
		rng.integers(0, 8000, 50000) / 1024.0, np.round(rng.uniform(0, 8000, 50000), 3),
		np.nextafter(x / 8, np.inf), np.nextafter(x / 8, -np.inf), x * 0.1, x / 3,
		[0.0, -0.0, 1.0, -1.0, 0.1, 0.5, 1e-4, 9.999999999999999e-5, 1e15, 999999999999999.9, 999999999999999.94,
			0.9999999999999999, 9.999999999999999e14, 123456789012345.6, 1e-5, 5e-324, 1.7976931348623157e308, 2.5,
			0.00012345678901234567, 1234.5, 8075.0, 0.30000000000000004, np.inf, -np.inf]])
	flat = cases.reshape(-1, 2)
	text = _shim.wkt_ring_texts_table(flat, np.array([0, len(flat)], dtype=np.int64))[0]
	got = text[1:-1].replace(",", "").split(" ")
	want = ["%.16g" % v for v in cases]
	assert got == want


def test_delaunay_points_on_degenerate_sets_is_a_delaunay_triangulation_of_the_hull():
	"""Small point sets on a coarse integer grid — collinear runs, points on hull edges, cocircular quadruples, twins —
	checked in exact integer arithmetic: positively oriented triangles that tile the convex hull, every interior edge
	locally Delaunay, ties counted exactly, twins named; UNSUPPORTED only for collinear input."""
	from fractions import Fraction
	rng = np.random.default_rng(21)

	def orient(a, b, c):
		return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])

	def hull_area2(points):
		pts = sorted(set(points))
		def chain(seq):
			out = []
			for p in seq:
				while len(out) >= 2 and orient(out[-2], out[-1], p) <= 0:
					out.pop()
				out.append(p)
			return out[:-1]
		ring = chain(pts) + chain(pts[::-1])
		return sum(ring[k][0] * ring[(k + 1) % len(ring)][1] - ring[(k + 1) % len(ring)][0] * ring[k][1] for k in range(len(ring)))

	n_unsupported = 0
	for trial in range(400):
		n = int(rng.integers(3, 40))
		span = int(rng.choice([2, 3, 5, 9, 30]))
		ipts = [tuple(int(v) for v in p) for p in rng.integers(0, span + 1, size=(n, 2))]
		pts = np.array(ipts, dtype=np.float64)
		distinct = sorted(set(ipts))
		collinear = len(distinct) < 3 or all(orient(distinct[0], distinct[1], p) == 0 for p in distinct)
		try:
			simplices, _, ties, twin = _shim.delaunay_points(pts, twins=True)
		except _shim.OgdError as e:
			assert e.status == _shim.ERR_UNSUPPORTED and collinear, (trial, ipts)
			n_unsupported += 1
			continue
		assert not collinear
		# twins: every dropped point names an earlier, kept point at the same place
		for i, t in enumerate(twin.tolist()):
			if t >= 0:
				assert t < i and twin[t] == -1 and ipts[t] == ipts[i]
		assert not np.isin(simplices, np.flatnonzero(twin >= 0)).any()
		area2, edges = 0, {}
		for t in simplices.tolist():
			a, b, c = (ipts[v] for v in t)
			o = orient(a, b, c)
			assert o > 0, (trial, t)
			area2 += o
			for k in range(3):
				edges.setdefault(frozenset((t[k], t[(k + 1) % 3])), []).append(t)
		assert area2 == hull_area2(ipts), (trial, ipts)
		exact_ties = 0
		for e, ts in edges.items():
			assert len(ts) <= 2
			if len(ts) == 2:
				t, u = ts
				d = [v for v in u if v not in e][0]
				s = _exact_incircle(pts[t[0]], pts[t[1]], pts[t[2]], pts[d])
				assert s <= 0, (trial, sorted(e))
				exact_ties += s == 0
		assert ties == exact_ties, (trial, ties, exact_ties)
	assert 0 < n_unsupported < 200
