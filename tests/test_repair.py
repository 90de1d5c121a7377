"""Polygon repair = shapely's `geom.buffer(0)` as the dewarp stage uses it (ref: origami/batch/detect/dewarp.py:
35-38).  GEOS cannot be installed here (PARITY UNPINNED against the library itself); pinned instead by
(1) the one published known answer — the shapely manual's bow-tie example, whose buffer(0) is a MultiPolygon of the
two triangles — and hand-derived cases, and (2) an independent statement of GEOS' rule as a point predicate
(oracle/rings.py:buffer0_depth: depth >= 1, no arrangement needed), against which the product's polygons are checked
on random self-crossing rings, rings with holes that cross the shell, and spikes."""

import numpy as np
import pytest

from oracle import rings as orings
from origami_amd import geometry
from origami_amd.repair import buffer0_rings, is_ccw_geos


def ring(*pts):
	return np.array(pts, dtype=np.float64)


def canon(r):
	"""Closed ring -> tuple of its vertices starting at the smallest one (orientation kept)."""
	pts = [tuple(p) for p in np.asarray(r)[:-1].tolist()]
	k = pts.index(min(pts))
	return tuple(pts[k:] + pts[:k])


def signed_area(r):
	r = np.asarray(r)
	return 0.5 * float(np.sum(r[:-1, 0] * r[1:, 1] - r[1:, 0] * r[:-1, 1]))


def test_shapely_manual_bowtie_gives_both_triangles():
	# shapely manual, "object.buffer": bowtie = Polygon([(0,0),(0,2),(1,1),(2,2),(2,0),(1,1),(0,0)]);
	# bowtie.is_valid is False; clean = bowtie.buffer(0) is a valid MultiPolygon of len 2 whose parts are
	# [(0,0),(0,2),(1,1),(0,0)] and [(1,1),(2,2),(2,0),(1,1)]
	bowtie = geometry.Geom("Polygon", [ring((0, 0), (0, 2), (1, 1), (2, 2), (2, 0), (1, 1), (0, 0))])
	assert not bowtie.is_valid
	clean = bowtie.buffer0()
	assert clean.geom_type == "MultiPolygon" and clean.is_valid and len(clean.parts) == 2
	got = sorted(canon(p[0]) for p in clean.parts)
	assert got == [((0.0, 0.0), (0.0, 2.0), (1.0, 1.0)), ((1.0, 1.0), (2.0, 2.0), (2.0, 0.0))]
	assert clean.wkt == "MULTIPOLYGON (((0 0, 0 2, 1 1, 0 0)), ((1 1, 2 2, 2 0, 1 1)))"


def test_crossing_bowtie_keeps_the_lobes_wound_like_its_top_corner():
	# edges cross at (1, 1): the lobes are wound opposite ways, buffer(0) keeps one ("drops half a bow-tie")
	res = buffer0_rings([ring((0, 0), (2, 2), (2, 0), (0, 2), (0, 0))])
	assert len(res) == 1 and len(res[0]) == 1
	assert canon(res[0][0]) == ((1.0, 1.0), (2.0, 2.0), (2.0, 0.0))
	# mirrored: the top corner now belongs to the other lobe
	res = buffer0_rings([ring((2, 0), (0, 2), (0, 0), (2, 2), (2, 0))])
	assert len(res) == 1 and canon(res[0][0]) == ((0.0, 0.0), (0.0, 2.0), (1.0, 1.0))


def test_valid_input_comes_back_as_itself_clockwise():
	sq = ring((0, 0), (4, 0), (4, 4), (0, 4), (0, 0))
	hole = ring((1, 1), (1, 2), (2, 2), (2, 1), (1, 1))
	for shell in (sq, sq[::-1]):
		res = buffer0_rings([shell, hole])
		assert len(res) == 1 and len(res[0]) == 2
		assert signed_area(res[0][0]) == -16 and signed_area(res[0][1]) == 1  # GEOS: shells CW, holes CCW
		assert set(canon(res[0][0])) == {(0.0, 0.0), (4.0, 0.0), (4.0, 4.0), (0.0, 4.0)}


def test_hole_crossing_the_shell_is_cut_out():
	sq = ring((0, 0), (4, 0), (4, 4), (0, 4), (0, 0))
	hole = ring((3, 1), (6, 1), (6, 3), (3, 3), (3, 1))
	res = buffer0_rings([sq, hole])
	assert len(res) == 1 and len(res[0]) == 1
	assert set(canon(res[0][0])) == {(0.0, 0.0), (0.0, 4.0), (4.0, 4.0), (4.0, 3.0), (3.0, 3.0), (3.0, 1.0), (4.0, 1.0), (4.0, 0.0)}
	assert signed_area(res[0][0]) == -(16 - 2)


def test_hole_touching_the_shell_stays_a_hole():
	sq = ring((0, 0), (4, 0), (4, 4), (0, 4), (0, 0))
	hole = ring((0, 2), (2, 1), (2, 3), (0, 2))
	res = buffer0_rings([sq, hole])
	assert len(res) == 1 and len(res[0]) == 2 and canon(res[0][1])[0] == (0.0, 2.0)


def test_spikes_and_double_loops():
	# a spike away from the top corner vanishes; the ring walked twice is its area once (depth 2 >= 1)
	res = buffer0_rings([ring((0, 0), (4, 0), (6, 0), (4, 0), (4, 4), (0, 4), (0, 0))])
	assert len(res) == 1 and signed_area(res[0][0]) == -16
	twice = ring((0, 0), (4, 0), (4, 4), (0, 4), (0, 0), (4, 0), (4, 4), (0, 4), (0, 0))
	res = buffer0_rings([twice])
	assert len(res) == 1 and signed_area(res[0][0]) == -16
	# a spike AT the top corner: GEOS 3.8's isCCW sees A-B-A there, calls the ring "not CCW" although it is, and
	# labels its inside as outside -> nothing is left.  Reproduced, not improved.
	top_spike = ring((0, 0), (4, 0), (4, 4), (4, 6), (4, 4), (0, 4), (0, 0))
	assert not is_ccw_geos([tuple(p) for p in top_spike.tolist()]) and signed_area(top_spike) > 0
	assert buffer0_rings([top_spike]) == []
	assert geometry.Geom("Polygon", [top_spike]).buffer0().wkt == "POLYGON EMPTY"
	# lines have no area
	assert geometry.Geom("LineString", [ring((1, 1), (1, 1))]).buffer0().wkt == "POLYGON EMPTY"
	# fewer than three distinct corners: no curve, no area
	assert buffer0_rings([ring((0, 0), (1, 1), (0, 0))]) == []


def test_is_ccw_matches_the_independent_statement():
	rng = np.random.default_rng(3)
	for _ in range(300):
		p = rng.integers(0, 8, size=(int(rng.integers(3, 9)), 2)).astype(np.float64)
		r = [tuple(q) for q in np.vstack([p, p[:1]]).tolist()]
		r = [q for i, q in enumerate(r) if i == 0 or q != r[i - 1]]
		if len(r) < 4:
			continue
		assert is_ccw_geos(r) == orings.geos_is_ccw(r), r


def _check_against_depth_rule(rings_in, rng, lo, hi, n_samples=300):
	res = buffer0_rings(rings_in)
	for poly in res:
		assert signed_area(poly[0]) < 0  # shell clockwise
		for k, r in enumerate(poly):
			assert np.array_equal(r[0], r[-1]) and len(r) >= 4
			assert not orings.ring_self_crosses(r), "a repaired ring still meets itself"
			if k > 0:
				assert signed_area(r) > 0
	bad = 0
	for p in rng.uniform(lo, hi, size=(n_samples, 2)):
		want = orings.buffer0_depth(rings_in, p) >= 1
		bad += orings.point_in_polygons(res, p) != want
	return res, bad


def test_random_self_crossing_rings_follow_geos_depth_rule():
	rng = np.random.default_rng(11)
	n_multi = n_empty = 0
	for trial in range(120):
		k = int(rng.integers(4, 13))
		p = rng.integers(0, 20, size=(k, 2)).astype(np.float64)
		shell = np.vstack([p, p[:1]])
		res, bad = _check_against_depth_rule([shell], rng, -1, 21)
		assert bad == 0, (trial, shell.tolist())
		n_multi += len(res) > 1
		n_empty += len(res) == 0
	assert n_multi > 10  # the cases are not trivial


def test_random_shells_with_crossing_holes_follow_geos_depth_rule():
	rng = np.random.default_rng(12)
	for trial in range(60):
		ang = np.sort(rng.uniform(0, 2 * np.pi, size=int(rng.integers(5, 12))))
		shell = np.round(np.stack([10 + 8 * np.cos(ang), 10 + 8 * np.sin(ang)], axis=1) * 4) / 4
		shell = np.vstack([shell, shell[:1]])
		if rng.random() < 0.5:
			shell = shell[::-1]
		holes = []
		for _ in range(int(rng.integers(1, 3))):
			c = rng.uniform(4, 16, size=2)
			a2 = np.sort(rng.uniform(0, 2 * np.pi, size=int(rng.integers(3, 7))))
			hole = np.round(np.stack([c[0] + 5 * np.cos(a2), c[1] + 5 * np.sin(a2)], axis=1) * 4) / 4
			holes.append(np.vstack([hole, hole[:1]]))
		res, bad = _check_against_depth_rule([shell] + holes, rng, 0, 20)
		assert bad == 0, (trial, shell.tolist(), [h.tolist() for h in holes])


def test_fractional_coordinates_as_a_warp_produces_them():
	"""Warped contours have arbitrary doubles, crossings fall on no lattice: the noded points are rounded to double
	and the arrangement must still close (Euler check inside) and follow the depth rule."""
	rng = np.random.default_rng(13)
	for trial in range(60):
		k = int(rng.integers(5, 30))
		p = rng.uniform(0, 3000, size=(k, 2))
		shell = np.vstack([p, p[:1]])
		res, bad = _check_against_depth_rule([shell], rng, 0, 3000, n_samples=200)
		assert bad == 0, trial
