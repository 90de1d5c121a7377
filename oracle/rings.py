"""Independent CPU reference for the ring predicates of the dewarp stage — TEST INFRASTRUCTURE ONLY.

ref: origami/batch/detect/dewarp.py:35 (`geom.is_valid`, shapely -> GEOS IsValidOp).  For a single ring the
quadratic part of GEOS' validity test is "no two segments meet except consecutive ones at their shared vertex"
(IsValidOp::checkNoSelfIntersectingRings / ConsistentAreaTester).  This file states that test in the plainest
way — a double loop over segment pairs with exact rational arithmetic (fractions.Fraction), no numpy
broadcasting, no floating-point orientation signs — so that it shares neither code nor rounding behaviour with
origami_amd/geometry.py or the device kernel it checks (ogd_rings_self_cross).  PARITY UNPINNED against GEOS
itself (not installable here); pinned by the hand-derived cases in tests/test_oracle.py.
"""

from fractions import Fraction


def _orient(p, q, r):
	"""Sign of the cross product (q - p) x (r - p), exact."""
	v = (q[0] - p[0]) * (r[1] - p[1]) - (q[1] - p[1]) * (r[0] - p[0])
	return (v > 0) - (v < 0)


def _in_box(p, q, r):
	return min(p[0], q[0]) <= r[0] <= max(p[0], q[0]) and min(p[1], q[1]) <= r[1] <= max(p[1], q[1])


def segments_meet(a, b, c, d):
	"""Do the closed segments ab and cd share at least one point?  Exact."""
	o1, o2 = _orient(a, b, c), _orient(a, b, d)
	o3, o4 = _orient(c, d, a), _orient(c, d, b)
	if o1 * o2 < 0 and o3 * o4 < 0:
		return True
	return (
		(o1 == 0 and _in_box(a, b, c)) or (o2 == 0 and _in_box(a, b, d)) or
		(o3 == 0 and _in_box(c, d, a)) or (o4 == 0 and _in_box(c, d, b)))


def ring_self_crosses(ring):
	"""ring: sequence of (x, y), closed (first == last).  True if two NON-adjacent segments touch or cross
	(adjacent = consecutive in the ring, the closing pair last/first included)."""
	pts = [(Fraction(float(x)), Fraction(float(y))) for x, y in ring]
	n = len(pts) - 1  # segments
	if n < 3:
		return False
	for i in range(n):
		for j in range(i + 2, n):
			if i == 0 and j == n - 1:
				continue  # the closing segment is adjacent to the first
			if segments_meet(pts[i], pts[i + 1], pts[j], pts[j + 1]):
				return True
	return False


# ---- what GEOS' buffer(0) keeps of an invalid polygon, stated as a point predicate -----------------------------
# ref: origami/batch/detect/dewarp.py:36 (`geom = geom.buffer(0)`).  GEOS labels both sides of every ring as if
# the ring were clockwise (after asking Orientation::isCCW and swapping for counter-clockwise rings), nodes the
# rings, and keeps the faces whose depth is >= 1.  For a point p off the rings that depth is
#     sum over rings of (+1 shell, -1 hole) * w_r(p) * (+1 if isCCW(r) else -1)
# with w_r the winding number of ring r around p.  The product (origami_amd/repair.py) builds the arrangement;
# this predicate needs none, so the two share nothing but the definition.

import math


def geos_is_ccw(ring):
	"""Orientation::isCCW of GEOS 3.8: the turn at the highest vertex (first of equals), taken between its nearest
	distinct neighbours; an A-B-A configuration there counts as "not CCW"; collinear: prev.x > next.x."""
	pts = [(float(x), float(y)) for x, y in ring]
	n = len(pts) - 1
	hi = max(range(n + 1), key=lambda i: (pts[i][1], -i))
	i = hi
	for _ in range(n + 1):
		i = (i - 1) % (n + 1) if i > 0 else n
		if pts[i] != pts[hi]:
			break
	j = hi
	for _ in range(n + 1):
		j = (j + 1) % n
		if pts[j] != pts[hi]:
			break
	a, b, c = pts[i], pts[hi], pts[j]
	if a == b or c == b or a == c:
		return False
	turn = _orient(tuple(map(Fraction, a)), tuple(map(Fraction, b)), tuple(map(Fraction, c)))
	return a[0] > c[0] if turn == 0 else turn > 0


def winding_number(ring, p):
	"""Winding number of the closed ring around p (not on the ring) by summing the signed angles its segments
	subtend — the textbook definition, in floating point, rounded to the nearest integer."""
	total = 0.0
	for (x0, y0), (x1, y1) in zip(ring[:-1], ring[1:]):
		ax, ay, bx, by = x0 - p[0], y0 - p[1], x1 - p[0], y1 - p[1]
		total += math.atan2(ax * by - ay * bx, ax * bx + ay * by)
	return int(round(total / (2 * math.pi)))


def buffer0_depth(rings, p):
	"""rings[0] = shell, rest = holes (closed); GEOS' depth at p.  p belongs to buffer(0)'s result iff depth >= 1."""
	depth = 0
	for k, ring in enumerate(rings):
		ring = [(float(x), float(y)) for x, y in ring]
		if len(ring) < 4:
			continue
		sign = (1 if k == 0 else -1) * (1 if geos_is_ccw(ring) else -1)
		depth += sign * winding_number(ring, p)
	return depth


def point_in_polygons(polygons, p):
	"""Even-odd membership of p in a list of polygons [[shell, hole, ...], ...] with disjoint interiors."""
	inside = False
	for poly in polygons:
		for ring in poly:
			for (x0, y0), (x1, y1) in zip(ring[:-1], ring[1:]):
				if (y0 > p[1]) != (y1 > p[1]) and p[0] < x0 + (p[1] - y0) * (x1 - x0) / (y1 - y0):
					inside = not inside
	return inside
