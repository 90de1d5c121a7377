"""Robust linear inter-/extrapolation of scattered 2-D samples over a box — the MI355X build of
origami/core/lingrid.py (same names, arguments and errors).

Host side (this file): topology and the small sequential set-up — convex hull and Delaunay by
Qhull (scipy.spatial), the border projections and corner values of the extrapolation point set
(ref: lingrid.py:27-165, 192-250).  Device side (libogdewarp.so): every evaluation — point queries
(K3, `ogd_tri_eval`, ref: lingrid.py:174-189, 286-290) and the dense lattice (K1b,
`ogd_tri_rasterize`, ref: lingrid.py:252-282).  There is no CPU evaluation path.
"""

from enum import Enum
from fractions import Fraction

import numpy as np
import scipy.spatial

try:
	from scipy.spatial import QhullError
except ImportError:  # pragma: no cover  (scipy < 1.8)
	from scipy.spatial.qhull import QhullError

from . import _shim


class Border(Enum):
	LEFT = 1
	TOP = 2
	BOTTOM = 3
	RIGHT = 4


def lerp(a, b, x):
	return (1 - x) * a + x * b


class Box:
	"""Outset bounding box that collects the extrapolation samples (ref: lingrid.py:27-165).

	The samples are kept as array chunks and the projections of a whole hull ring are computed in one go
	(add_projections): a lattice hull has hundreds of edges, and one numpy call per number was 8 ms of a page's
	set-up.  The arithmetic per number is unchanged — IEEE double operations in the reference's order; the edge
	length comes from the same BLAS dot product np.linalg.norm uses on a 2-vector, taken row by row."""

	def __init__(self, minx, miny, maxx, maxy, margin=1):
		self._minx, self._miny = minx - margin, miny - margin
		self._maxx, self._maxy = maxx + margin, maxy + margin
		self._box_size = max(maxx - minx, maxy - miny) + 2 * margin + 1
		self._chunks = []  # (rows [n][2 + nv] float64, tags [n][4] bool in the order of _SIDES), insertion order
		self._corners = {}

	_SIDES = (Border.LEFT, Border.RIGHT, Border.TOP, Border.BOTTOM)

	def _table(self):
		if len(self._chunks) > 1:
			self._chunks = [(np.vstack([c[0] for c in self._chunks]), np.vstack([c[1] for c in self._chunks]))]
		return self._chunks[0] if self._chunks else (np.empty((0, 3)), np.empty((0, 4), dtype=bool))

	@property
	def points(self):
		for row in self._table()[0]:
			yield row
		for (cx, cy), val in self._corners.items():
			yield np.hstack([[cx, cy], val])

	def _intersect(self, p, direction):
		"""First crossing of the rays p + u * direction (0 <= u <= box_size, |direction| = 1) with the box outline,
		replacing shapely's LineString/LinearRing intersection (ref: lingrid.py:62-93); p, direction: [n][2].
		The coordinate of the border that is hit is returned exactly, so the reference's equality test for
		border membership holds by construction.  Returns x, y, tags [n][4]."""
		px, py = p[:, 0], p[:, 1]
		dx, dy = direction[:, 0], direction[:, 1]
		inside = (self._minx < px) & (px < self._maxx) & (self._miny < py) & (py < self._maxy)
		if not inside.all():
			raise ValueError("point %s outside given domain %s" % (
				p[np.argmin(inside)], (self._minx, self._miny, self._maxx, self._maxy)))
		wall_x = np.where(dx > 0, self._maxx, self._minx)
		wall_y = np.where(dy > 0, self._maxy, self._miny)
		with np.errstate(divide="ignore", invalid="ignore"):
			ux = np.where(dx != 0, (wall_x - px) / dx, np.inf)
			uy = np.where(dy != 0, (wall_y - py) / dy, np.inf)
		by_x = ux <= uy
		u = np.where(by_x, ux, uy)
		if not (u <= self._box_size).all():
			k = np.argmin(u <= self._box_size)
			raise ValueError("point %s outside given domain, ray along %s does not hit the box" % (p[k], direction[k]))
		x = np.where(by_x, wall_x, px + u * dx)
		y = np.where(uy <= ux, wall_y, py + u * dy)
		left, top = x == self._minx, y == self._miny
		tags = np.stack([left, ~left & (x == self._maxx), top, ~top & (y == self._maxy)], axis=1)
		return x, y, tags

	def add_projections(self, a, b):
		"""Cast both end points of every edge (a[k], b[k]) along the edge normal onto the box, in edge order:
		a's image, then b's (ref: lingrid.py:95-104).  a, b: [n][2 + nv] rows of [x, y, value...]."""
		a = np.asarray(a, dtype=np.float64).reshape(len(a), -1)
		b = np.asarray(b, dtype=np.float64).reshape(len(b), -1)
		v = a[:, :2] - b[:, :2]
		length = np.sqrt(np.array([e.dot(e) for e in v]).reshape(len(v)))
		keep = ~(length < 1e-2)
		a, b, v = a[keep], b[keep], v[keep]
		if len(v) == 0:
			return
		normal = np.stack([-v[:, 1], v[:, 0]], axis=1)
		normal /= np.sqrt(np.array([e.dot(e) for e in normal]))[:, np.newaxis]
		ends = np.stack([a, b], axis=1).reshape(2 * len(a), -1)  # a0, b0, a1, b1, ...
		x, y, tags = self._intersect(ends[:, :2], np.repeat(normal, 2, axis=0))
		self._chunks.append((np.column_stack([x, y, ends[:, 2:]]), tags))

	def add_projection(self, a, b):
		self.add_projections(np.asarray(a, dtype=np.float64)[np.newaxis], np.asarray(b, dtype=np.float64)[np.newaxis])

	def _nearest_to_corner(self, border, f, axis):
		rows, tags = self._table()
		rows = rows[tags[:, self._SIDES.index(border)]]
		if not len(rows):
			return None
		d = dict(x=0, y=1)[axis]
		# the first of equal keys, as Python's min / max over the rows in insertion order
		return rows[np.argmin(rows[:, d]) if f is min else np.argmax(rows[:, d])]

	def _add_corner(self, cx, cy, p1, p2):
		# ref: lingrid.py:106-130; the sympy projection is evaluated exactly (rationals), then rounded
		if p1 is None and p2 is None:
			return
		if p1 is None:
			val = p2[2:]
		elif p2 is None:
			val = p1[2:]
		else:
			a = (Fraction(float(p1[0])), Fraction(float(p1[1])))
			e = (Fraction(float(p2[0])) - a[0], Fraction(float(p2[1])) - a[1])
			ee = e[0] * e[0] + e[1] * e[1]
			if ee == 0:
				raise ValueError("Line requires two unique Points")
			t = ((Fraction(float(cx)) - a[0]) * e[0] + (Fraction(float(cy)) - a[1]) * e[1]) / ee
			p = np.array([float(a[0] + t * e[0]), float(a[1] + t * e[1])], dtype=np.float64)
			d = np.linalg.norm(p - p1[:2])
			d_total = np.linalg.norm(p2[:2] - p1[:2])
			if d_total == 0:
				return
			val = lerp(p1[2:], p2[2:], d / d_total)
		self._corners[(cx, cy)] = val

	def add_corners(self):
		# ref: lingrid.py:138-165 — top left, top right, bottom right, bottom left
		near = self._nearest_to_corner
		self._add_corner(self._minx, self._miny, near(Border.LEFT, min, "y"), near(Border.TOP, min, "x"))
		self._add_corner(self._maxx, self._miny, near(Border.RIGHT, min, "y"), near(Border.TOP, max, "x"))
		self._add_corner(self._maxx, self._maxy, near(Border.RIGHT, max, "y"), near(Border.BOTTOM, max, "x"))
		self._add_corner(self._minx, self._maxy, near(Border.LEFT, max, "y"), near(Border.BOTTOM, min, "x"))


class Interpolator:
	"""Clip to bounds, evaluate the inner interpolant, fall back to the extrapolation where it is
	NaN (ref: lingrid.py:168-189).  Evaluation runs on the GPU (K3)."""

	def __init__(self, inter, extra, bounds, squeeze):
		self._inter = inter  # _shim.Triangulation or None (collinear input)
		self._extra = extra
		self._bounds = bounds
		self._squeeze = squeeze

	@property
	def bounds(self):
		return self._bounds

	@property
	def triangulations(self):
		return self._inter, self._extra

	def __call__(self, pts):
		pts = np.array(pts)
		if len(pts.shape) == 1:
			pts = pts[np.newaxis, :]
		minx, miny, maxx, maxy = self._bounds
		# clipping happens in the dtype of the query (float32 for Field.get), as in the reference
		pts[:, 0] = np.clip(pts[:, 0], minx, maxx)
		pts[:, 1] = np.clip(pts[:, 1], miny, maxy)
		out = _shim.tri_eval(self._inter, self._extra, None, pts)
		return out[:, 0] if self._squeeze else out

	def ragged(self, pts, offsets):
		"""Batch form for many rings: one wavefront per ring; clipping on the device (fp64)."""
		out = _shim.tri_eval_ragged(self._inter, self._extra, self._bounds, pts, offsets)
		return out[:, 0] if self._squeeze else out


class InterpolatorFactory:
	"""ref: lingrid.py:192-293."""

	def __init__(self, points, values, bounds, lattice_shape=None):
		"""lattice_shape=(h, w): `points` are the row-major nodes of a warped lattice (the Transformer's
		input).  Hull and Delaunay triangulation then come from ogd_lattice_delaunay (known topology,
		exact Lawson flips, identical to Qhull's result wherever that is unique) instead of two Qhull
		runs; inputs outside its domain fall back to Qhull."""
		if len(values) < 2:
			raise ValueError("not enough values given.")

		self._points = np.asarray(points, dtype=np.float64).reshape(-1, 2)
		self._squeeze = not isinstance(values[0], np.ndarray)  # scalar vs vector values
		self._values = np.asarray(values, dtype=np.float64)
		if self._squeeze:
			self._values = self._values.reshape(-1)
		self._bounds = bounds

		box = Box(*bounds)
		pts, vals = self._points, self._values

		self._lattice = None
		if lattice_shape is not None and tuple(lattice_shape) >= (2, 2):
			h, w = lattice_shape
			try:
				self._lattice = _shim.lattice_delaunay(pts.reshape(h, w, 2), transform=False)
			except _shim.OgdError as e:
				if e.status != _shim.ERR_UNSUPPORTED:
					raise
		try:
			if self._lattice is not None:
				hull_vertices = self._lattice[1]
			else:
				hull_vertices = scipy.spatial.ConvexHull(pts).vertices  # counter-clockwise in 2-D
			self._is_collinear = False
		except QhullError:
			self._is_collinear = True  # most probably points are collinear

		if not self._is_collinear:
			# of coincident input points the reference keeps the value of the last one (dict build)
			if self._lattice is not None:
				idx = list(hull_vertices)  # a lattice that passed ogd_lattice_delaunay has no coincident nodes
			else:
				idx = [int(np.flatnonzero((pts[:, 0] == pts[i, 0]) & (pts[:, 1] == pts[i, 1]))[-1]) for i in hull_vertices]
			ring = np.hstack([pts[hull_vertices], vals[idx].reshape(len(idx), -1)])
			box.add_projections(ring, np.roll(ring, -1, axis=0))
			head = ring
		else:
			rows = np.hstack([pts, vals.reshape(len(pts), -1)])
			# (a, b) then (b, a) for every consecutive pair, as the reference's loop interleaves them
			box.add_projections(
				np.stack([rows[:-1], rows[1:]], axis=1).reshape(-1, rows.shape[1]),
				np.stack([rows[1:], rows[:-1]], axis=1).reshape(-1, rows.shape[1]))
			head = np.empty((0, rows.shape[1]))

		box.add_corners()
		rows = np.array(list(box.points), dtype=np.float64).reshape(-1, head.shape[1])
		rows = np.vstack([head, rows])
		self._extra_pts = np.ascontiguousarray(rows[:, :2])
		self._extra_val = np.ascontiguousarray(rows[:, 2:])
		if self._squeeze:
			assert self._extra_val.shape[1] == 1
			self._extra_val = self._extra_val[:, 0]
		self._interpolator = None

	@property
	def extra_points(self):
		return self._extra_pts

	@property
	def extra_values(self):
		return self._extra_val

	@property
	def is_collinear(self):
		return self._is_collinear

	@staticmethod
	def _triangulate(points, values):
		# The triangulation that LinearNDInterpolator builds internally (ref: lingrid.py:286-290).  The Delaunay
		# triangulation is unique unless four points share an empty circle, so ogd_delaunay_points (exact sweep +
		# flips, host) returns Qhull's simplices at a fraction of the time; with cocircular ties, coincident points
		# of different values or collinear points it is Qhull itself, scipy's Delaunay defaults ("Qbb Qc Qz Q12" + "Qt"), that decides.
		try:
			simplices, _, ties, twin = _shim.delaunay_points(points, twins=True, transform=False)
			transform = None  # computed on the device, as scipy computes it (ogd_tri_create_scipy)
			dropped = np.flatnonzero(twin >= 0)
			if len(dropped) and not np.array_equal(np.asarray(values)[dropped], np.asarray(values)[twin[dropped]]):
				ties = -1  # coincident points with different values: the answer depends on which one Qhull keeps
		except _shim.OgdError as e:
			if e.status != _shim.ERR_UNSUPPORTED:
				raise
			ties = -1
		if ties != 0:
			tri = scipy.spatial.Delaunay(points)
			simplices, transform = tri.simplices, tri.transform
		return _shim.Triangulation(points, simplices, values, transform, scipy_transform=True)

	@property
	def interpolator(self):
		if self._interpolator is None:
			extra = self._triangulate(self._extra_pts, self._extra_val)
			inter = None
			if self._lattice is not None:
				simplices, _, _ = self._lattice
				inter = _shim.Triangulation(self._points, simplices, self._values, scipy_transform=True)
				self._lattice = None
			elif not self._is_collinear:
				inter = self._triangulate(self._points, self._values)
			self._interpolator = Interpolator(inter, extra, self._bounds, self._squeeze)
		return self._interpolator

	@property
	def grid(self):
		"""Dense variant on the integer lattice [0..maxx] x [0..maxy], indexed [x, y] as the
		reference's mgrid (ref: lingrid.py:252-282); rasterised on the GPU (K1b)."""
		minx, miny, maxx, maxy = self._bounds
		assert minx == 0 and miny == 0
		inter, extra = self.interpolator.triangulations
		nx, ny = int(maxx) + 1, int(maxy) + 1
		pixels = _shim.tri_rasterize(inter, extra, None, 0.0, 0.0, 1.0, nx, ny, order="xy")
		return pixels[:, :, 0] if self._squeeze else pixels


def lingrid(points, values, width, height):
	return InterpolatorFactory(points, values, (0, 0, width - 1, height - 1)).grid


def lininterp(points, values, bounds, lattice_shape=None):
	return InterpolatorFactory(points, values, bounds, lattice_shape=lattice_shape).interpolator
