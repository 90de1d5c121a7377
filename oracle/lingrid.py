"""CPU restatement of origami/core/lingrid.py — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Robust linear inter-/extrapolation of scattered 2-D samples over a box: a Delaunay interpolant
inside the convex hull plus a second one over hull + border projections + corners.

SciPy (Qhull, LinearNDInterpolator, griddata) is the same third-party library the reference calls, so
it is called here directly.  Shapely/GEOS (ray-box intersection, lingrid.py:62-93) and sympy (corner
projection, lingrid.py:115-120) are absent from the build container: their results are restated in
closed form (exact rational arithmetic for the sympy part).  PARITY UNPINNED for those two.
"""

from fractions import Fraction

import numpy as np
import scipy.interpolate
import scipy.spatial

try:  # scipy >= 1.8
	from scipy.spatial import QhullError
except ImportError:  # pragma: no cover
	from scipy.spatial.qhull import QhullError

LEFT, TOP, BOTTOM, RIGHT = "left", "top", "bottom", "right"


class BorderBox:
	"""lingrid.py:27-165 (class Box).  Collects value-carrying points on the outset box."""

	def __init__(self, minx, miny, maxx, maxy, margin=1):
		# lingrid.py:29-41
		self.minx = minx - margin
		self.miny = miny - margin
		self.maxx = maxx + margin
		self.maxy = maxy + margin
		self.ray_length = max(maxx - minx, maxy - miny) + 2 * margin + 1
		self.points = []  # rows [x, y, val...]
		self.on_border = {LEFT: [], TOP: [], BOTTOM: [], RIGHT: []}
		self.corners = {}  # (cx, cy) -> val, insertion ordered

	def cast(self, p, direction):
		"""lingrid.py:62-93: where the segment p -> p + direction * ray_length crosses the box ring,
		and which borders that point lies on (exact float equality, lingrid.py:79-91)."""
		px, py = float(p[0]), float(p[1])
		dx, dy = float(direction[0]), float(direction[1])
		if not (self.minx < px < self.maxx and self.miny < py < self.maxy):
			raise ValueError("point %s outside given domain" % (p,))
		ux = np.inf
		uy = np.inf
		if dx > 0:
			bx = self.maxx
			ux = (bx - px) / dx
		elif dx < 0:
			bx = self.minx
			ux = (bx - px) / dx
		if dy > 0:
			by = self.maxy
			uy = (by - py) / dy
		elif dy < 0:
			by = self.miny
			uy = (by - py) / dy
		u = min(ux, uy)
		if not u <= self.ray_length:
			raise ValueError("ray from %s along %s does not hit the box" % (p, direction))
		x = bx if ux <= uy else px + u * dx
		y = by if uy <= ux else py + u * dy
		borders = []
		if x == self.minx:
			borders.append(LEFT)
		elif x == self.maxx:
			borders.append(RIGHT)
		if y == self.miny:
			borders.append(TOP)
		elif y == self.maxy:
			borders.append(BOTTOM)
		return borders, x, y

	def add(self, borders, x, y, val):
		# lingrid.py:47-53
		pt = np.hstack([[x, y], val]).astype(np.float64)
		self.points.append(pt)
		for b in borders:
			self.on_border[b].append(pt)

	def add_projection(self, a, b):
		"""lingrid.py:95-104: project both ends of hull edge (a, b) along the outward edge normal."""
		a = np.asarray(a, dtype=np.float64)
		b = np.asarray(b, dtype=np.float64)
		v = a[:2] - b[:2]
		if np.linalg.norm(v) < 1e-2:
			return
		normal = np.array([-v[1], v[0]])
		normal /= np.linalg.norm(normal)
		self.add(*self.cast(a[:2], normal), a[2:])
		self.add(*self.cast(b[:2], normal), b[2:])

	def _corner(self, cx, cy, p1, p2):
		# lingrid.py:106-130
		if p1 is None and p2 is None:
			return
		if p1 is None:
			val = p2[2:]
		elif p2 is None:
			val = p1[2:]
		else:
			# sympy Line(p1, p2).projection(corner).evalf(): exact rational projection rounded to
			# double (lingrid.py:115-120)
			ax, ay = Fraction(float(p1[0])), Fraction(float(p1[1]))
			bx, by = Fraction(float(p2[0])), Fraction(float(p2[1]))
			ex, ey = bx - ax, by - ay
			den = ex * ex + ey * ey
			if den == 0:
				raise ValueError("Line requires two unique Points")
			t = ((Fraction(float(cx)) - ax) * ex + (Fraction(float(cy)) - ay) * ey) / den
			p = np.array([float(ax + t * ex), float(ay + t * ey)], dtype=np.float64)
			d = np.linalg.norm(p - p1[:2])
			d_total = np.linalg.norm(p2[:2] - p1[:2])
			if d_total == 0:
				return
			x = d / d_total
			val = (1 - x) * p1[2:] + x * p2[2:]
		self.corners[(cx, cy)] = val

	def _nearest(self, border, pick, axis):
		# lingrid.py:132-136 (min / max keep the first of equal keys)
		pts = self.on_border[border]
		if not pts:
			return None
		return pick(pts, key=lambda pt: pt[axis])

	def add_corners(self):
		# lingrid.py:138-165: TL, TR, BR, BL
		self._corner(self.minx, self.miny, self._nearest(LEFT, min, 1), self._nearest(TOP, min, 0))
		self._corner(self.maxx, self.miny, self._nearest(RIGHT, min, 1), self._nearest(TOP, max, 0))
		self._corner(self.maxx, self.maxy, self._nearest(RIGHT, max, 1), self._nearest(BOTTOM, max, 0))
		self._corner(self.minx, self.maxy, self._nearest(LEFT, max, 1), self._nearest(BOTTOM, min, 0))

	def all_points(self):
		# lingrid.py:55-60
		for pt in self.points:
			yield pt
		for (cx, cy), val in self.corners.items():
			yield np.hstack([[cx, cy], val])


class Factory:
	"""lingrid.py:192-293 (InterpolatorFactory)."""

	def __init__(self, points, values, bounds):
		if len(values) < 2:
			raise ValueError("not enough values given.")
		self.points = points
		self.values = values
		self.bounds = bounds
		box = BorderBox(*bounds)
		self.squeeze = not isinstance(values[0], np.ndarray)  # lingrid.py:203-206

		try:
			hull = scipy.spatial.ConvexHull(points)
			hull_pts = list(hull.points[hull.vertices])
			self.is_collinear = False
		except QhullError:
			self.is_collinear = True

		if not self.is_collinear:
			# lingrid.py:217-227; a dict keyed by coordinates: of duplicate points the last value wins
			lookup = dict(zip([tuple(p) for p in points], values))
			hull_val = [lookup[tuple(p)] for p in hull_pts]
			extra_pts = hull_pts[:]
			extra_val = hull_val[:]
			n = len(hull_pts)
			for i in range(n):
				j = (i + 1) % n
				box.add_projection(
					np.hstack([hull_pts[i], hull_val[i]]), np.hstack([hull_pts[j], hull_val[j]]))
		else:
			# lingrid.py:229-237
			for i in range(len(points) - 1):
				a = np.hstack([points[i], values[i]])
				b = np.hstack([points[i + 1], values[i + 1]])
				box.add_projection(a, b)
				box.add_projection(b, a)
			extra_pts = []
			extra_val = []

		box.add_corners()
		for pt in box.all_points():
			extra_pts.append(pt[:2])
			if self.squeeze:
				assert len(pt[2:]) == 1
				extra_val.extend(pt[2:])
			else:
				extra_val.append(pt[2:])
		self.extra_pts = np.array(extra_pts)
		self.extra_val = np.array(extra_val)

	def interpolator(self):
		# lingrid.py:284-293
		extra = scipy.interpolate.LinearNDInterpolator(self.extra_pts, self.extra_val, fill_value=np.nan)
		inner = None
		if not self.is_collinear:
			inner = scipy.interpolate.LinearNDInterpolator(self.points, self.values, fill_value=np.nan)
		return Interpolator(inner, extra, self.bounds)

	def grid(self):
		"""lingrid.py:252-282: dense variant on mgrid[0:maxx+1, 0:maxy+1], indexed [x, y].
		(In the reference the collinear case leaves `pixels` unbound; here it returns the
		extrapolation alone.)"""
		minx, miny, maxx, maxy = self.bounds
		assert minx == 0 and miny == 0
		lattice = np.dstack(np.mgrid[0:maxx + 1, 0:maxy + 1])
		extra = scipy.interpolate.griddata(
			self.extra_pts, self.extra_val, lattice, method="linear", fill_value=np.nan)
		if self.is_collinear:
			return extra
		inner = scipy.interpolate.griddata(
			np.asarray(self.points), np.asarray(self.values), lattice, method="linear", fill_value=np.nan)
		return np.where(np.isnan(inner), extra, inner)


class Interpolator:
	"""lingrid.py:168-189."""

	def __init__(self, inner, extra, bounds):
		self.inner = inner
		self.extra = extra
		self.bounds = bounds

	def __call__(self, pts):
		pts = np.array(pts)
		if pts.ndim == 1:
			pts = pts[np.newaxis, :]
		minx, miny, maxx, maxy = self.bounds
		pts[:, 0] = np.clip(pts[:, 0], minx, maxx)
		pts[:, 1] = np.clip(pts[:, 1], miny, maxy)
		if self.inner is None:
			return self.extra(pts)
		ri = self.inner(pts)
		rx = self.extra(pts)
		return np.where(np.isnan(ri), rx, ri)


def lininterp(points, values, bounds):
	"""lingrid.py:300-301."""
	return Factory(points, values, bounds).interpolator()


def lingrid(points, values, width, height):
	"""lingrid.py:296-297."""
	return Factory(points, values, (0, 0, width - 1, height - 1)).grid()
