"""CPU restatement of origami/core/dewarp.py and of the numeric part of
origami/batch/detect/dewarp.py — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Plain numpy / scipy loops that follow the reference function by function.  The shapely calls
(ray x polyline intersection, dewarp.py:220-263) are restated as closed-form segment intersections,
cv2.resize / cv2.remap by oracle/cv_oracle.c.  PARITY UNPINNED (no fixtures in the reference; cv2 and
shapely cannot run here).
"""

import io
import json
import math
import zipfile

import numpy as np
import scipy.interpolate

from . import cv
from .lingrid import lininterp


class Samples:
	"""dewarp.py:34-105: flow samples (x, y, phi) of one direction."""

	def __init__(self, size, points=None, values=None):
		self.size = tuple(size)
		self.points = [] if points is None else points
		self.values = [] if values is None else values

	def __len__(self):
		return len(self.points)

	@staticmethod
	def open(zf, name):
		# dewarp.py:43-59
		info = json.loads(zf.read("%s.json" % name))
		array = np.load(io.BytesIO(zf.read("%s.npy" % name)), allow_pickle=False)
		if array.size > 0:
			return Samples(info["size"], list(array[:, :2]), list(array[:, 2]))
		return Samples(info["size"], np.empty((2, 0)), [])

	def save(self, zf, name):
		# dewarp.py:61-75
		points = np.array(self.points)
		values = np.array(self.values)
		if points.size == 0:
			array = np.empty((3, 0))
		else:
			array = np.hstack([points, values[:, np.newaxis]])
		data = io.BytesIO()
		np.save(data, array.astype(np.float64), allow_pickle=False)
		zf.writestr("%s.npy" % name, data.getvalue())
		zf.writestr("%s.json" % name, json.dumps(dict(version=1, size=self.size)))


class Field:
	"""dewarp.py:108-140: angle field -> unit direction field."""

	def __init__(self, samples, size, phi0):
		self.size = size
		if len(samples.points) <= 2:
			self.interp = lambda pts: [phi0 for _ in pts]
		else:
			self.interp = lininterp(samples.points, samples.values, (0, 0, size[0], size[1]))

	def get(self, pts):
		angles = self.interp(pts)
		dx = np.cos(angles, dtype=np.float32)
		dy = np.sin(angles, dtype=np.float32)
		return np.squeeze(np.dstack([dx, dy]), axis=0)

	def estimate_extent(self, axis, limit, step_size):
		pts = np.array([[0, y] for y in range(0, self.size[1 - axis], step_size)]).astype(np.float32)
		if axis != 0:
			pts = np.flip(pts, axis=-1)
		n_steps = 1
		max_steps = 2 * (1 + self.size[axis] // n_steps)
		while np.any(pts[:, axis] < limit) and n_steps < max_steps:
			pts += self.get(pts) * step_size
			n_steps += 1
		if n_steps >= max_steps:
			raise RuntimeError("n_steps exceeded %d" % max_steps)
		return n_steps


def _extrapolate(a, b, x):
	# dewarp.py:164-167 (float32 in, float32 out)
	v = b - a
	v /= np.linalg.norm(v)
	return b + x * v


def extend_row_ends(grid_h, grid_res, k=3):
	"""The side effect of ShapelyBatchIntersections.make_row_slice (dewarp.py:199-218): numpy basic
	slices are views, so building the per-row segment pieces overwrites the first and last node of
	rows 1.. of grid_h with points pushed `large` pixels outwards — the last node once per piece
	that reaches the row end."""
	grid_w = grid_h.shape[1]
	large = grid_w * grid_res
	for gy in range(1, grid_h.shape[0]):
		for i in range(0, grid_w, k - 1):
			pts = grid_h[gy, min(i, grid_w - k):i + k]
			if i == 0:
				pts[0] = _extrapolate(pts[1], pts[0], large)
			elif i + k >= grid_w:
				pts[-1] = _extrapolate(pts[-2], pts[-1], large)


def _segment_hits(p0, p1, row):
	"""All points where segment p0-p1 meets the polyline `row` (n, 2), float64, closed intervals."""
	hits = []
	x1, y1 = float(p0[0]), float(p0[1])
	x2, y2 = float(p1[0]), float(p1[1])
	rx, ry = x2 - x1, y2 - y1
	for j in range(len(row) - 1):
		x3, y3 = float(row[j, 0]), float(row[j, 1])
		x4, y4 = float(row[j + 1, 0]), float(row[j + 1, 1])
		sx, sy = x4 - x3, y4 - y3
		den = rx * sy - ry * sx
		if den == 0:
			continue  # parallel (a collinear overlap makes the reference raise)
		qx, qy = x3 - x1, y3 - y1
		t = (qx * sy - qy * sx) / den
		u = (qx * ry - qy * rx) / den
		if 0 <= t <= 1 and 0 <= u <= 1:
			hits.append((x1 + t * rx, y1 + t * ry))
	return hits


def intersect_rows(grid_h, pts0, pts1, gy):
	"""dewarp.py:220-263: replace each ray end pts1[i] by the nearest crossing of ray i with row
	gy + 1 of grid_h (falling through to later rows if there is none)."""
	n_rows = grid_h.shape[0] - 1
	for i in range(len(pts0)):
		dy = 0
		while True:
			hits = _segment_hits(pts0[i], pts1[i], grid_h[gy + 1 + dy])
			if hits:
				break
			if gy + dy + 1 < n_rows:
				dy += 1
			else:
				raise RuntimeError(
					"failed to find intersection for i=%d, n=%d, %s to %s." % (
						i, len(pts0), pts0[i], pts1[i]))
		if len(hits) == 1:
			pts1[i] = hits[0]
		else:
			dist = np.linalg.norm(np.array(hits) - pts0[i], axis=1)
			pts1[i] = hits[np.argmin(dist)]


def _make_slices(n, k):
	s = math.ceil(n / k)
	for i in range(0, n, s):
		yield slice(i, i + s)


class GridFactory:
	"""dewarp.py:292-493."""

	def __init__(self, page_size, samples_h, samples_v, grid_res=25, max_grid_size=1000, num_threads=2):
		self.width, self.height = page_size
		self.grid_res = grid_res
		self.max_grid_size = max_grid_size
		self.num_threads = num_threads
		size = (self.width, self.height)
		self.field_h = Field(samples_h, size, phi0=0)
		self.field_v = Field(samples_v, size, phi0=math.pi / 2)

	def grid_shape(self):
		# dewarp.py:328-339
		est_width = self.field_h.estimate_extent(0, self.width, step_size=self.grid_res)
		est_height = self.field_v.estimate_extent(1, self.height, step_size=self.grid_res)
		if max(est_width, est_height) > self.max_grid_size:
			raise RuntimeError("estimated grid too big: (%d, %d)" % (est_height, est_width))
		return est_height, est_width

	def grid_h(self):
		# dewarp.py:421-439
		shape = self.grid_shape()
		grid = np.zeros((shape[0], shape[1], 2), dtype=np.float32)
		res = self.grid_res
		pts = np.array([[0, gy * res] for gy in range(grid.shape[0])], dtype=np.float32)
		for gx in range(grid.shape[1]):
			grid[:, gx, :] = pts
			pts += self.field_h.get(pts) * res
		assert not np.any(np.isnan(grid))
		return grid

	def grid_hv(self):
		# dewarp.py:441-493
		grid_h = self.grid_h()
		field_v = self.field_v.get
		grid_hv = np.zeros(grid_h.shape, dtype=np.float32)
		extend_row_ends(grid_h, self.grid_res)  # constructor of BatchIntersections, dewarp.py:448

		def compute_slice(sel):
			max_dy = 0
			for gy in range(1, grid_h.shape[0]):
				max_dy = max(max_dy, np.max(grid_h[gy, sel, 1]) - np.min(grid_h[gy - 1, sel, 1]))
			max_step_len = max_dy / math.cos(60 * (math.pi / 180))
			pts0 = grid_h[0][sel]
			for gy in range(grid_h.shape[0] - 1):
				grid_hv[gy, sel, :] = pts0
				pts1 = pts0 + field_v(pts0) * max_step_len
				intersect_rows(grid_h, pts0, pts1, gy)
				pts0 = pts1
			grid_hv[-1, sel, :] = pts0

		if self.num_threads < 2:
			compute_slice(slice(0, grid_h.shape[1]))
		else:
			for sel in _make_slices(grid_h.shape[1], self.num_threads):
				compute_slice(sel)

		grid_hv = self._extend_h(grid_hv, "left")
		grid_hv = self._extend_h(grid_hv, "right")
		grid_hv = self._extend_v(grid_hv, "top")
		grid_hv = self._extend_v(grid_hv, "bottom")
		assert not np.any(np.isnan(grid_hv))
		return grid_hv

	def _extend_h(self, grid_hv, side):
		# dewarp.py:341-379
		max_borders = self.max_grid_size // 2
		if side == "left":
			cond = lambda x: np.any(x[:, 0, 0] > 0)
			step = -self.grid_res
			borders = [grid_hv[:, :1]]
		else:
			cond = lambda x: np.any(x[:, -1, 0] < self.width)
			step = self.grid_res
			borders = [grid_hv[:, -1:]]
		while cond(borders[-1]):
			if len(borders) >= max_borders:
				raise RuntimeError("border extension not terminating")
			pts = borders[-1][:, 0]
			new_pts = pts + self.field_h.get(pts) * step
			borders.append(new_pts.reshape((grid_hv.shape[0], 1, 2)))
		if len(borders) == 1:
			return grid_hv
		new = borders[1:]
		return np.hstack(list(reversed(new)) + [grid_hv] if side == "left" else [grid_hv] + new)

	def _extend_v(self, grid_hv, side):
		# dewarp.py:381-419
		max_borders = self.max_grid_size // 2
		if side == "top":
			cond = lambda x: np.any(x[0, :, 1] > 0)
			step = -self.grid_res
			borders = [grid_hv[:1, :]]
		else:
			cond = lambda x: np.any(x[-1, :, 1] < self.height)
			step = self.grid_res
			borders = [grid_hv[-1:, :]]
		while cond(borders[-1]):
			if len(borders) >= max_borders:
				raise RuntimeError("border extension not terminating")
			pts = borders[-1][0, :]
			new_pts = pts + self.field_v.get(pts) * step
			borders.append(new_pts.reshape((1, grid_hv.shape[1], 2)))
		if len(borders) == 1:
			return grid_hv
		new = borders[1:]
		return np.vstack(list(reversed(new)) + [grid_hv] if side == "top" else [grid_hv] + new)


class Transformer:
	"""dewarp.py:143-161: forward point map warped -> dewarped."""

	def __init__(self, grid, grid_res):
		h, w = grid.shape[:2]
		source = grid.reshape((h * w, 2))
		target = np.flip(np.dstack(np.mgrid[0:h, 0:w]), axis=-1).reshape((h * w, 2)) * grid_res
		bounds = (np.min(source[:, 0]), np.min(source[:, 1]), np.max(source[:, 0]), np.max(source[:, 1]))
		self.interp = lininterp(source, target, bounds)

	def __call__(self, x, y):
		pts = self.interp(np.squeeze(np.dstack([x, y]), axis=0))
		assert not np.any(np.isnan(pts))
		return pts[:, 0], pts[:, 1]


class Grid:
	"""dewarp.py:496-576."""

	def __init__(self, hv, res):
		self.hv = hv
		self.res = res
		self._full = None
		self._transformer = None

	@staticmethod
	def create(page_size, samples_h, samples_v, **kwargs):
		factory = GridFactory(page_size, samples_h, samples_v, **kwargs)
		return Grid(factory.grid_hv(), factory.grid_res)

	@property
	def dewarped_size(self):
		h, w = self.hv.shape[:2]
		return w * self.res, h * self.res

	def points_full(self, threads=1):
		# dewarp.py:518-524 (cached like the reference's lru_cache: one array object)
		if self._full is None:
			self._full = cv.dense_grid(self.hv, self.res, threads)
		return self._full

	@property
	def transformer(self):
		# dewarp.py:555-559
		if self._transformer is None:
			r = self.res
			self._transformer = Transformer(self.points_full()[::r, ::r], r)
		return self._transformer

	def inverse_yx(self, pts_yx):
		# dewarp.py:561-572
		grid = self.points_full()
		interp = scipy.interpolate.RegularGridInterpolator(
			(np.arange(grid.shape[0]), np.arange(grid.shape[1])), grid, method="linear",
			bounds_error=False, fill_value=None)
		return interp(pts_yx)

	def inverse(self, pts_xy):
		# dewarp.py:574-576
		return self.inverse_yx(np.flip(pts_xy, axis=-1))

	def save(self, path):
		# dewarp.py:546-553
		data = io.BytesIO()
		np.save(data, self.hv.astype(np.float32), allow_pickle=False)
		info = dict(version=1, cell=self.res, shape=self.hv.shape)
		with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED) as zf:
			zf.writestr("data.npy", data.getvalue())
			zf.writestr("meta.json", json.dumps(info))


def dewarped(pixels, grid, threads=1):
	"""Dewarper.dewarped, dewarp.py:608-614."""
	return cv.remap_u8(pixels, grid.points_full(threads), cv.INTER_LINEAR, threads)


def dewarp_image(page_size, pixels, grid, interpolation=cv.INTER_LINEAR, threads=1):
	"""Dewarper.dewarp_image, dewarp.py:616-627, on a COPY of the dense grid (the reference scales
	its lru-cached array in place; parity is defined on the first call)."""
	w0, h0 = page_size
	h1, w1 = pixels.shape[:2]
	full = grid.points_full(threads).copy()
	full[:, :, 0] *= w1 / w0
	full[:, :, 1] *= h1 / h0
	return cv.remap_u8(pixels, full, interpolation, threads)


def transform_rings(transformer, rings):
	"""shapely.ops.transform(transformer, geom), batch/detect/dewarp.py:26: one transformer call per
	ring (shell incl. closing vertex, then holes) / linestring.  rings: list of (n, 2) float64."""
	out = []
	for ring in rings:
		x, y = transformer(tuple(ring[:, 0]), tuple(ring[:, 1]))
		out.append(np.stack([x, y], axis=1))
	return out
