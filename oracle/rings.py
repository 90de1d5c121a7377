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
