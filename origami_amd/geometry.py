"""Minimal planar geometry for the dewarp stage: WKT in/out, area, validity.

Origami reads and writes contours as WKT through shapely (ref: origami/batch/detect/dewarp.py:23-39,
origami/batch/core/io.py:142-144); shapely/GEOS is not part of the MI355X image, so the few
operations the dewarp stage needs are implemented here.  Text output follows GEOS 3.8's WKTWriter
as driven by shapely 1.7's `geom.wkt` (trim on, 16 significant digits: C++ stream default float
format, i.e. "%.16g"); area follows GEOS Area::ofRingSigned.
"""

import re

import numpy as np

_NUM = r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?|[-+]?(?:nan|inf)"
_RING = re.compile(r"\(([^()]*)\)")
_PLAIN = re.compile(r"^[-+0-9.eE,\s]*$")


class Geom:
	"""A Polygon (rings[0] = shell, rest = holes), a LineString (one ring) or — only as the result of a repair
	(buffer0) or read back from a repaired contour — a MultiPolygon (parts = [[shell, hole, ...], ...];
	`rings` then lists the rings of all parts in order)."""

	__slots__ = ("geom_type", "rings", "parts")

	def __init__(self, geom_type, rings):
		if geom_type not in ("Polygon", "LineString", "MultiPolygon"):
			raise ValueError("unsupported geometry type %s" % geom_type)
		self.geom_type = geom_type
		as_ring = lambda r: np.asarray(r, dtype=np.float64).reshape(-1, 2)
		if geom_type == "MultiPolygon":
			self.parts = [[as_ring(r) for r in part] for part in rings]
			self.rings = [r for part in self.parts for r in part]
		else:
			self.rings = [as_ring(r) for r in rings]
			self.parts = [self.rings] if geom_type == "Polygon" and self.rings else []

	@property
	def is_empty(self):
		return len(self.rings) == 0 or len(self.rings[0]) == 0

	@property
	def area(self):
		if self.geom_type == "LineString" or self.is_empty:
			return 0.0
		a = 0.0
		for part in self.parts:
			a += abs(ring_area_signed(part[0]))
			for hole in part[1:]:
				a -= abs(ring_area_signed(hole))
		return a

	def buffer0(self):
		"""`geom.buffer(0)` as the dewarp stage uses it to repair an invalid contour (ref: origami/batch/detect/
		dewarp.py:36): the area GEOS keeps (origami_amd/repair.py) as a Polygon, a MultiPolygon, or POLYGON EMPTY
		(always for a LineString: a line has no area)."""
		from .repair import buffer0_rings
		polygons = []
		if self.geom_type != "LineString":
			for part in self.parts:
				polygons.extend(buffer0_rings(part))
		if not polygons:
			return Geom("Polygon", [])
		if len(polygons) == 1:
			return Geom("Polygon", polygons[0])
		return Geom("MultiPolygon", polygons)

	@property
	def is_valid(self):
		return self.valid_given(None)

	def valid_given(self, crossing):
		"""is_valid with the quadratic part precomputed: crossing[k] tells whether ring k — with repeated
		consecutive points removed (without_repeats) — touches or crosses itself (the batch stage gets these flags
		for all rings of a page from one launch of ogd_rings_self_cross); None computes them here."""
		if self.is_empty:
			return True
		if self.geom_type == "LineString":
			r = self.rings[0]
			return len(r) >= 2 and bool(np.all(np.isfinite(r))) and not bool(np.all(r == r[0]))
		flags = list(crossing) if crossing is not None else [None] * len(self.rings)
		k = 0
		for part in self.parts:
			n = len(part)
			if not all(_ring_is_valid(r, c) for r, c in zip(part, flags[k:k + n])) or not holes_inside(part):
				return False
			k += n
		return True

	def transform(self, fn):
		"""shapely.ops.transform: fn(xs, ys) -> (xs', ys') applied ring by ring."""
		out = []
		for r in self.rings:
			x, y = fn(tuple(r[:, 0]), tuple(r[:, 1]))
			out.append(np.stack([np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)], axis=1))
		return self.with_rings(out)

	def with_rings(self, rings):
		"""The same geometry with every ring replaced (rings in the order of `self.rings`)."""
		if self.geom_type != "MultiPolygon":
			return Geom(self.geom_type, rings)
		parts, k = [], 0
		for part in self.parts:
			parts.append(rings[k:k + len(part)])
			k += len(part)
		return Geom("MultiPolygon", parts)

	@property
	def wkt(self):
		return to_wkt(self)


def ring_area_signed(ring):
	n = len(ring)
	if n < 3:
		return 0.0
	x = ring[1:n - 1, 0] - ring[0, 0]
	total = 0.0
	for xi, y_prev, y_next in zip(x, ring[0:n - 2, 1], ring[2:n, 1]):
		total += xi * (y_prev - y_next)
	return total / 2.0


def _segments_cross(r):
	"""True if any two non-adjacent segments of the closed ring r touch or cross."""
	a, b = r[:-1], r[1:]
	n = len(a)
	if n < 3:
		return False
	d = b - a

	def orient(p, q, s):  # sign of (q - p) x (s - p), broadcast
		return np.sign((q[..., 0] - p[..., 0]) * (s[..., 1] - p[..., 1]) - (q[..., 1] - p[..., 1]) * (s[..., 0] - p[..., 0]))

	A, B = a[:, None, :], b[:, None, :]
	C, D = a[None, :, :], b[None, :, :]
	o1, o2 = orient(A, B, C), orient(A, B, D)
	o3, o4 = orient(C, D, A), orient(C, D, B)
	proper = (o1 * o2 < 0) & (o3 * o4 < 0)

	def on_seg(p, q, s, o):  # s collinear with pq and inside its bounding box
		return (o == 0) & (np.minimum(p, q) <= s).all(-1) & (s <= np.maximum(p, q)).all(-1)

	touch = on_seg(A, B, C, o1) | on_seg(A, B, D, o2) | on_seg(C, D, A, o3) | on_seg(C, D, B, o4)
	i, j = np.indices((n, n))
	gap = np.abs(i - j)
	non_adjacent = (gap > 1) & (gap < n - 1)
	del d
	return bool(np.any((proper | touch) & non_adjacent))


def without_repeats(r):
	"""The ring without repeated consecutive points.  GEOS' validity test works on the ring with repeats removed:
	a vertex written twice is legal, it is not a ring touching itself."""
	r = np.asarray(r, dtype=np.float64).reshape(-1, 2)
	if len(r) < 2:
		return r
	keep = np.ones(len(r), dtype=bool)
	keep[1:] = np.any(r[1:] != r[:-1], axis=1)
	return r if keep.all() else r[keep]


def _ring_is_valid(r, crosses=None):
	"""crosses: the precomputed answer of the quadratic test for without_repeats(r), or None to compute it here."""
	if len(r) < 4 or not np.all(np.isfinite(r)):
		return False
	if not np.array_equal(r[0], r[-1]):
		return False
	r = without_repeats(r)
	if len(r) < 4:  # fewer than three distinct corners ("too few points")
		return False
	if ring_area_signed(r) == 0:
		return False
	return not (_segments_cross(r) if crosses is None else crosses)


def convex_hull_ring(points):
	"""Convex hull of a point set as a closed ring, clockwise like GEOS' ConvexHull writes it (what
	`geom.convex_hull.exterior.coords` holds, up to the starting vertex); None if the points span no area."""
	pts = sorted(set(map(tuple, np.asarray(points, dtype=np.float64).reshape(-1, 2).tolist())))
	if len(pts) < 3:
		return None

	def chain(seq):
		out = []
		for p in seq:
			while len(out) >= 2 and (
					(out[-1][0] - out[-2][0]) * (p[1] - out[-2][1]) - (out[-1][1] - out[-2][1]) * (p[0] - out[-2][0])) <= 0:
				out.pop()
			out.append(p)
		return out[:-1]

	ccw = chain(pts) + chain(pts[::-1])
	if len(ccw) < 3:
		return None
	cw = ccw[::-1]
	return np.array(cw + cw[:1], dtype=np.float64)


def _point_in_ring(p, r):
	x, y = p
	x0, y0 = r[:-1, 0], r[:-1, 1]
	x1, y1 = r[1:, 0], r[1:, 1]
	cond = (y0 > y) != (y1 > y)
	with np.errstate(divide="ignore", invalid="ignore"):
		xi = x0 + (y - y0) * (x1 - x0) / (y1 - y0)
	return bool(np.count_nonzero(cond & (x < xi)) % 2)


def holes_inside(rings):
	"""rings[0] = shell, the rest holes: does every hole start inside the shell."""
	shell = rings[0]
	return all(_point_in_ring(h[0], shell) for h in rings[1:])


def parse_wkt(text):
	"""POLYGON / LINESTRING (2-D) -> Geom."""
	text = text.strip()
	m = re.match(r"^(MULTIPOLYGON|POLYGON|LINESTRING)\s*(Z\s*)?(EMPTY|\(.*\))\s*$", text, re.IGNORECASE | re.DOTALL)
	if not m:
		raise ValueError("unsupported WKT: %.40s" % text)
	kind = {"POLYGON": "Polygon", "LINESTRING": "LineString", "MULTIPOLYGON": "MultiPolygon"}[m.group(1).upper()]
	body = m.group(3)
	if body.upper() == "EMPTY":
		return Geom(kind, [])
	if kind == "MultiPolygon":
		# "((ring), (ring)), ((ring))": a polygon is a parenthesis group at depth 1 of the body
		polygons, depth, start = [], 0, 0
		for i, ch in enumerate(body[1:-1]):
			if ch == "(":
				if depth == 0:
					start = i
				depth += 1
			elif ch == ")":
				depth -= 1
				if depth == 0:
					polygons.append(_parse_rings(_RING.findall(body[1:-1][start + 1:i])))
		return Geom(kind, polygons)
	if kind == "LineString":
		parts = [body[1:-1]]
	else:
		parts = _RING.findall(body[1:-1])
	return Geom(kind, _parse_rings(parts))


def _parse_rings(parts):
	rings = []
	for part in parts:
		n_coord = part.count(",") + 1 if part.strip() else 0
		if n_coord == 0:
			rings.append(np.empty((0, 2)))
			continue
		if _PLAIN.match(part):  # digits, signs, exponents, blanks, commas: numpy's text parser (strtod)
			nums = np.fromstring(part.replace(",", " "), dtype=np.float64, sep=" ")
		else:  # nan / inf spellings and anything unusual
			nums = np.array(re.findall(_NUM, part, re.IGNORECASE), dtype=np.float64)
		dim = len(nums) // n_coord
		if dim < 2 or dim * n_coord != len(nums):
			raise ValueError("malformed WKT coordinates")
		rings.append(nums.reshape(n_coord, dim)[:, :2])
	return rings


def format_number(v):
	"""GEOS 3.8 WKTWriter::writeNumber with trim: std::setprecision(16) on the default float format."""
	if np.isnan(v):
		return "nan"
	if np.isinf(v):
		return "inf" if v > 0 else "-inf"
	return "%.16g" % v


def _ring_text(r):
	return "(" + ", ".join("%s %s" % (format_number(x), format_number(y)) for x, y in r) + ")"


def to_wkt(geom):
	if geom.geom_type == "MultiPolygon":
		if geom.is_empty:
			return "MULTIPOLYGON EMPTY"
		return "MULTIPOLYGON (" + ", ".join(
			"(" + ", ".join(_ring_text(r) for r in part) + ")" for part in geom.parts) + ")"
	if geom.geom_type == "Polygon":
		if geom.is_empty:
			return "POLYGON EMPTY"
		return "POLYGON (" + ", ".join(_ring_text(r) for r in geom.rings) + ")"
	if geom.is_empty:
		return "LINESTRING EMPTY"
	return "LINESTRING " + _ring_text(geom.rings[0])


def to_wkt_many(geoms):
	"""WKT of many geometries with the coordinate text of ALL their rings formatted by one call into
	libogdewarp (ogd_wkt_ring_text, host code: what GEOS' C++ WKTWriter is to the reference)."""
	from . import _shim
	texts = iter(_shim.wkt_ring_texts([r for g in geoms if not g.is_empty for r in g.rings]))
	out = []
	for g in geoms:
		if g.is_empty:
			out.append(to_wkt(g))
		elif g.geom_type == "MultiPolygon":
			out.append("MULTIPOLYGON (" + ", ".join(
				"(" + ", ".join(next(texts) for _ in part) + ")" for part in g.parts) + ")")
		elif g.geom_type == "Polygon":
			out.append("POLYGON (" + ", ".join(next(texts) for _ in g.rings) + ")")
		else:
			out.append("LINESTRING " + next(texts))
	return out


# ---- many geometries at once: flat ring tables -------------------------------------------------------------------
# A page has hundreds of contours of a dozen vertices each; per-ring numpy calls cost more than the arithmetic.
# The batch stage therefore keeps all rings of a page in one (N, 2) array with ring r at [offsets[r], offsets[r+1])
# and asks the questions above once per page.

_KINDS = ("Polygon", "LineString", "MultiPolygon")


def parse_wkt_many(texts):
	"""[Geom] for many WKT texts (bytes or str) through one call of ogd_wkt_parse (host code in libogdewarp: what
	GEOS' WKTReader is to the reference); texts it does not take go through parse_wkt one by one, which raises
	the errors."""
	from . import _shim
	texts = [t.encode("utf8") if isinstance(t, str) else bytes(t) for t in texts]
	try:
		kinds, pts, ring_offsets, ring_geom, ring_part = _shim.wkt_parse(texts)
	except _shim.OgdError as e:
		if e.status != _shim.ERR_UNSUPPORTED:
			raise
		return [parse_wkt(t.decode("utf8")) for t in texts]
	first = np.searchsorted(ring_geom, np.arange(len(texts) + 1))
	out = []
	for g, kind in enumerate(kinds.tolist()):
		r0, r1 = int(first[g]), int(first[g + 1])
		rings = [pts[ring_offsets[r]:ring_offsets[r + 1]] for r in range(r0, r1)]
		if kind == 2:
			parts = [[] for _ in range(int(ring_part[r1 - 1]) + 1)] if r1 > r0 else []
			for r, ring in zip(range(r0, r1), rings):
				parts[ring_part[r]].append(ring)
			out.append(Geom("MultiPolygon", parts))
		else:
			out.append(Geom(_KINDS[kind], rings))
	return out


def ring_table(rings):
	"""(flat (N, 2) float64, offsets (R + 1,) int64) of a list of rings."""
	offsets = np.zeros(len(rings) + 1, dtype=np.int64)
	if rings:
		np.cumsum([len(r) for r in rings], out=offsets[1:])
		flat = np.concatenate([np.asarray(r, dtype=np.float64).reshape(-1, 2) for r in rings])
	else:
		flat = np.empty((0, 2), dtype=np.float64)
	return np.ascontiguousarray(flat), offsets


def without_repeats_table(flat, offsets):
	"""without_repeats for every ring of a table."""
	keep = np.ones(len(flat), dtype=bool)
	if len(flat) > 1:
		keep[1:] = np.any(flat[1:] != flat[:-1], axis=1)
		starts = offsets[:-1]
		keep[starts[starts < len(flat)]] = True  # the first point of a ring is never a repeat
	kept_before = np.concatenate([[0], np.cumsum(keep)])
	return np.ascontiguousarray(flat[keep]), kept_before[offsets].astype(np.int64)


def rings_valid_table(flat, offsets, crossing=None):
	"""_ring_is_valid for every ring of a table (closed polygon rings) -> (valid (R,) bool, and the by-products
	finite (R,) bool, distinct (R,) number of points left by without_repeats).  crossing: the flags of
	ogd_rings_self_cross for the rings without repeats, or None to ask the device here."""
	from . import _shim
	n = np.diff(offsets)
	d_flat, d_offsets = without_repeats_table(flat, offsets)
	nd = np.diff(d_offsets)
	bad = np.concatenate([[0], np.cumsum(~np.isfinite(flat).all(axis=1))])
	finite = bad[offsets[1:]] == bad[offsets[:-1]]
	some = n > 0
	closed = np.zeros(len(n), dtype=bool)
	closed[some] = np.all(flat[offsets[:-1][some]] == flat[offsets[1:][some] - 1], axis=1)
	if crossing is None:
		crossing = _shim.rings_self_cross_table(d_flat, d_offsets)
	area = _shim.rings_area_signed(d_flat, d_offsets)
	valid = (n >= 4) & finite & closed & (nd >= 4) & (area != 0) & ~np.asarray(crossing, dtype=bool)
	return valid, finite, nd
