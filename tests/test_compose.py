"""`Document.rewarp` of the compose stage (ref: origami/batch/detect/compose.py:413-432): inverse point map on the
GPU + shapely's validity / convex hull / clip-to-page on the host.  The host part is tested here on CPU with a
stand-in grid; tests/test_gpu_lines.py checks the inverse map itself against scipy's RegularGridInterpolator."""

import numpy as np
import pytest

from origami_amd.batch.detect.compose import Rewarper, page_outline
from origami_amd.geometry import convex_hull_ring


def area(ring):
	r = np.asarray(ring)
	return 0.5 * float(np.sum(r[:-1, 0] * r[1:, 1] - r[1:, 0] * r[:-1, 1]))


def corners(ring):
	return sorted(set(map(tuple, np.asarray(ring).tolist())))


class ShiftGrid:
	"""inverse(): dewarped -> warped by a constant shift (enough to test the host geometry)."""

	def __init__(self, dx, dy):
		self.d = np.array([dx, dy], dtype=np.float64)
		self.calls = 0

	def inverse(self, pts):
		self.calls += 1
		return np.asarray(pts, dtype=np.float64) + self.d


def test_outline_inside_the_page_is_returned_whole():
	sq = [(10, 10), (50, 10), (50, 40), (10, 40)]
	out = page_outline(sq, (100, 80))
	assert corners(out) == [(10.0, 10.0), (10.0, 40.0), (50.0, 10.0), (50.0, 40.0)]
	assert np.array_equal(out[0], out[-1]) and abs(area(out)) == 40 * 30


def test_outline_is_clipped_to_the_page_no_negative_coordinates():
	sq = [(-20, -10), (50, -10), (50, 40), (-20, 40)]
	out = page_outline(sq, (100, 80))
	assert corners(out) == [(0.0, 0.0), (0.0, 40.0), (50.0, 0.0), (50.0, 40.0)]
	assert out.min() >= 0


def test_invalid_outline_becomes_its_convex_hull():
	bowtie = [(10, 10), (50, 50), (50, 10), (10, 50)]
	out = page_outline(bowtie, (100, 80))
	assert corners(out) == [(10.0, 10.0), (10.0, 50.0), (50.0, 10.0), (50.0, 50.0)]


def test_outline_cut_into_several_parts_becomes_their_convex_hull():
	# a U lying on its side, its two arms reach into the page, its back lies left of it
	u = [(-30, 10), (40, 10), (40, 20), (-10, 20), (-10, 50), (40, 50), (40, 60), (-30, 60)]
	out = page_outline(u, (100, 80))
	assert corners(out) == [(0.0, 10.0), (0.0, 60.0), (40.0, 10.0), (40.0, 60.0)]


def test_outline_outside_the_page_raises_like_the_reference():
	with pytest.raises(RuntimeError, match="failed to rewarp coords"):
		page_outline([(-50, -50), (-10, -50), (-10, -10), (-50, -10)], (100, 80))


def test_rewarper_maps_all_outlines_with_one_inverse_call():
	grid = ShiftGrid(-5, 3)
	rw = Rewarper(grid, (100, 80))
	a = [(10, 10), (30, 10), (30, 30), (10, 30)]
	b = [(0, 0), (20, 0), (20, 20), (0, 20)]
	got = rw.many([a, b])
	assert grid.calls == 1
	assert corners(got[0]) == [(5.0, 13.0), (5.0, 33.0), (25.0, 13.0), (25.0, 33.0)]
	assert corners(got[1]) == [(0.0, 3.0), (0.0, 23.0), (15.0, 3.0), (15.0, 23.0)]  # x clipped at 0
	assert corners(rw(a)) == corners(got[0])
	assert rw.many([]) == []


def test_convex_hull_ring():
	pts = [(0, 0), (4, 0), (2, 1), (4, 4), (0, 4), (2, 2), (0, 0)]
	hull = convex_hull_ring(pts)
	assert corners(hull) == [(0.0, 0.0), (0.0, 4.0), (4.0, 0.0), (4.0, 4.0)] and area(hull) == -16  # clockwise
	assert convex_hull_ring([(0, 0), (1, 1), (2, 2)]) is None
