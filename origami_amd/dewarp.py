"""Dewarping with a vector field interpolated from scattered flow samples — the MI355X build of
origami/core/dewarp.py (same classes, arguments, artifacts and errors).

Host side (this file) keeps what is small: sample I/O, the border extension of the node grid (a few
steps), artifact I/O.  Everything per pixel, per point or per marching step runs in libogdewarp.so
on the GPU:

  Field.get / Transformer.__call__   -> K3  ogd_tri_eval[_ragged]   (ref: dewarp.py:119, 159)
  GridFactory.grid_shape / grid_h / grid_hv -> ogd_march_shape, ogd_march_grid: every walk of the
                                        march is one launch       (ref: dewarp.py:328-339, 421-493)
  Grid.points("full")                -> K1a ogd_grid_fill           (ref: dewarp.py:522-523)
  Dewarper.dewarped / dewarp_image   -> K2  ogd_dewarp_u8 (fused K1a+K2; ref: dewarp.py:612-614, 625-627)
  Grid.inverse                       -> ogd_grid_sample_bilinear    (ref: dewarp.py:565-567)
"""

import io
import json
import math
import os
import zipfile

import numpy as np
import PIL.Image

from . import _shim
from ._shim import INTER_LINEAR, INTER_NEAREST  # noqa: F401  (cv2-compatible constants)
from .lingrid import lininterp


class Geometry:
	"""ref: origami/core/math.py:69-91."""

	def __init__(self, w, h):
		self._size = (w, h)

	@property
	def size(self):
		return self._size

	@property
	def area(self):
		w, h = self.size
		return w * h

	@property
	def diameter(self):
		w, h = self.size
		return np.sqrt(w * w + h * h)

	def rel_length(self, x):
		return self.diameter * x

	def rel_area(self, a):
		return (self.diameter * a) ** 2


class Samples:
	"""Flow samples (x, y, phi) of one direction; `flow.zip` members <name>.npy / <name>.json
	(ref: dewarp.py:34-105, docs/formats.md:145-187: a float64 (n, 3) table, or shape (3, 0) when there are none,
	beside {"version": 1, "size": [W, H]})."""

	def __init__(self, geometry, points=(), values=()):
		self._geometry = geometry
		self._points = list(points)
		self._values = list(values)

	@classmethod
	def open(cls, zf, name):
		size = json.loads(zf.read(name + ".json"))["size"]
		table = np.load(io.BytesIO(zf.read(name + ".npy")), allow_pickle=False)
		if table.size:
			return cls(Geometry(*size), table[:, :2], table[:, 2])
		none = cls(Geometry(*size))
		none._points = np.empty((2, 0))  # what the reference leaves in `points` for an empty member
		return none

	def save(self, zf, name):
		if len(self._values):
			xy = np.asarray(self._points, dtype=np.float64).reshape(-1, 2)
			table = np.column_stack([xy, np.asarray(self._values, dtype=np.float64)])
		else:
			table = np.empty((3, 0), dtype=np.float64)
		member = io.BytesIO()
		np.save(member, table, allow_pickle=False)
		zf.writestr(name + ".npy", member.getvalue())
		zf.writestr(name + ".json", json.dumps(dict(version=1, size=self._geometry.size)))

	def __len__(self):
		return len(self._points)

	geometry = property(lambda self: self._geometry)
	points = property(lambda self: self._points)
	values = property(lambda self: self._values)

	def append(self, point, value):
		self.extend([point], [value])

	def extend(self, points, values):
		self._points.extend(points)
		self._values.extend(values)

	@property
	def std(self):
		return np.std(self._values) if len(self._values) > 3 else 0


class Field:
	"""Angle field -> unit direction field (ref: dewarp.py:108-140)."""

	def __init__(self, samples, size, phi0):
		self._size = size
		if len(samples.points) <= 2:
			self._interp = lambda pts: [phi0 for _ in pts]
			self._device = float(phi0)
		else:
			self._interp = lininterp(samples.points, samples.values, (0, 0, size[0], size[1]))
			self._device = self._interp.triangulations

	@property
	def device(self):
		"""What the march kernels evaluate: the (inner, extra) triangulations, or the constant angle."""
		return self._device

	def get(self, pts):
		angles = self._interp(pts)  # fp64 on the GPU
		dx = np.cos(angles, dtype=np.float32)
		dy = np.sin(angles, dtype=np.float32)
		return np.squeeze(np.dstack([dx, dy]), axis=0)

	def estimate_extent(self, axis, limit, step_size):
		"""Euler steps of `step_size` until every walk started along the other axis has passed `limit`, plus one
		(ref: dewarp.py:124-140).  The stage asks ogd_march_shape for both axes in one launch; this host form is
		the reference's method, kept for its callers."""
		starts = np.arange(0, self._size[1 - axis], step_size)
		pts = np.zeros((len(starts), 2), dtype=np.float32)
		pts[:, 1 - axis] = starts
		bound = 2 * (1 + self._size[axis])
		for n_steps in range(1, bound):
			if not np.any(pts[:, axis] < limit):
				return n_steps
			pts += self.get(pts) * step_size
		raise RuntimeError("n_steps exceeded %d" % bound)


class Transformer:
	"""Forward point map warped page -> dewarped page (ref: dewarp.py:143-161)."""

	def __init__(self, grid, grid_res):
		h, w = grid.shape[:2]
		source = grid.reshape((h * w, 2))
		gy, gx = np.mgrid[0:h, 0:w]
		target = np.stack([gx, gy], axis=-1).reshape((h * w, 2)) * grid_res
		bounds = (
			np.min(source[:, 0]), np.min(source[:, 1]),
			np.max(source[:, 0]), np.max(source[:, 1]))
		# the nodes are a warped lattice: triangulated from that topology, not by Qhull
		self._interp = lininterp(source, target, bounds, lattice_shape=(h, w))

	def __call__(self, x, y):
		pts = self._interp(np.squeeze(np.dstack([x, y]), axis=0))
		assert not np.any(np.isnan(pts))
		return pts[:, 0], pts[:, 1]

	def transform_table(self, flat, offsets):
		"""All rings of a page — ring r is flat[offsets[r]:offsets[r + 1]] — in ONE launch, one wavefront per ring
		(K3 ragged); equivalent to calling the transformer once per ring as shapely.ops.transform does
		(ref: origami/batch/detect/dewarp.py:26)."""
		if len(flat) == 0:
			return np.empty((0, 2), dtype=np.float64)
		out = self._interp.ragged(flat, offsets)
		assert not np.any(np.isnan(out))
		return out

	def transform_rings(self, rings):
		"""transform_table for a list of (n, 2) rings; returns the list of warped rings."""
		if not rings:
			return []
		offsets = np.zeros(len(rings) + 1, dtype=np.int64)
		np.cumsum([len(r) for r in rings], out=offsets[1:])
		flat = np.concatenate([np.asarray(r, dtype=np.float64).reshape(-1, 2) for r in rings])
		out = self.transform_table(flat, offsets)
		return [out[a:b] for a, b in zip(offsets[:-1], offsets[1:])]


class GridFactory:
	"""ref: dewarp.py:292-493."""

	def __init__(self, page, samples_h, samples_v, grid_res=25, max_grid_size=1000, num_threads=2):
		# the page's size without decoding its pixels where the page object can tell (origami_amd.page.Page);
		# any object with a `warped` image works as in the reference (ref: dewarp.py:299)
		size = page.size(False) if callable(getattr(page, "size", None)) else page.warped.size
		self._width = size[0]
		self._height = size[1]
		self._grid_res = grid_res
		self._max_grid_size = max_grid_size
		self._num_threads = num_threads
		self._samples_h = samples_h
		self._samples_v = samples_v
		self._cache = {}

	def _cached(self, key, make):
		if key not in self._cache:
			self._cache[key] = make()
		return self._cache[key]

	@property
	def res(self):
		return self._grid_res

	@property
	def field_h(self):
		return self._cached("field_h", lambda: Field(
			self._samples_h, (self._width, self._height), phi0=0))

	@property
	def field_v(self):
		return self._cached("field_v", lambda: Field(
			self._samples_v, (self._width, self._height), phi0=math.pi / 2))

	@property
	def _bounds(self):
		return (0, 0, self._width, self._height)

	@property
	def grid_shape(self):
		def make():
			# both extent walks (Field.estimate_extent, ref: dewarp.py:124-140) in one launch
			est_height, est_width = _shim.march_shape(
				self.field_h.device, self.field_v.device, self._bounds, self._width, self._height, self._grid_res)
			if max(est_width, est_height) > self._max_grid_size:
				raise RuntimeError("estimated grid too big: (%d, %d)" % (est_height, est_width))
			return est_height, est_width
		return self._cached("grid_shape", make)

	def _extend(self, grid_hv, axis, sign):
		"""Append whole columns (axis 1) / rows (axis 0) by Euler steps of +-grid_res until every
		border node lies outside the page (ref: dewarp.py:341-419)."""
		field = (self.field_h if axis == 1 else self.field_v).get
		limit = self._width if axis == 1 else self._height
		edge = -1 if sign > 0 else 0
		max_borders = self._max_grid_size // 2
		pts = np.take(grid_hv, edge, axis=axis)
		added = []

		def inside(p):
			c = p[:, 1 - axis]
			return np.any(c < limit) if sign > 0 else np.any(c > 0)

		while inside(pts):
			if len(added) + 1 >= max_borders:
				raise RuntimeError("border extension not terminating")
			pts = pts + field(pts) * (self._grid_res * sign)
			added.append(np.expand_dims(pts, axis))
		if not added:
			return grid_hv
		parts = [grid_hv] + added if sign > 0 else added[::-1] + [grid_hv]
		return np.concatenate(parts, axis=axis)

	def extend_border_h(self, grid_hv, side):
		if side not in ("left", "right"):
			raise ValueError(side)
		return self._extend(grid_hv, 1, -1 if side == "left" else 1)

	def extend_border_v(self, grid_hv, side):
		if side not in ("top", "bottom"):
			raise ValueError(side)
		return self._extend(grid_hv, 0, -1 if side == "top" else 1)

	def _march(self):
		"""grid_h and the core of grid_hv on the device: rows of grid_h are independent walks, and so are
		the columns of grid_hv once grid_h exists (ogd_march_grid; ref: dewarp.py:421-483)."""
		def make():
			n_rows, n_cols = self.grid_shape
			return _shim.march_grid(
				self.field_h.device, self.field_v.device, self._bounds, self._grid_res, n_rows, n_cols,
				max(1, self._num_threads))
		return self._cached("march", make)

	@property
	def grid_h(self):
		"""As the reference leaves it once grid_hv has run: the first and the last node of rows 1..
		are pushed outwards by make_row_slice's in-place extension (ref: dewarp.py:205-211)."""
		return self._march()[0]

	@property
	def grid_hv(self):
		return self._cached("grid_hv", self._make_grid_hv)

	def _make_grid_hv(self):
		grid_hv = self._march()[1]
		grid_hv = self.extend_border_h(grid_hv, "left")
		grid_hv = self.extend_border_h(grid_hv, "right")
		grid_hv = self.extend_border_v(grid_hv, "top")
		grid_hv = self.extend_border_v(grid_hv, "bottom")
		assert not np.any(np.isnan(grid_hv))
		return grid_hv


class Grid:
	"""The marched node grid and everything derived from it (ref: dewarp.py:496-576)."""

	def __init__(self, hv, res):
		self._grid_hv = hv
		self._grid_res = res
		self._full = {}
		self._transformer = None

	@property
	def geometry(self):
		h, w = self._grid_hv.shape[:2]
		r = self._grid_res
		return Geometry(w * r, h * r)

	def points(self, resolution="full", interpolation=INTER_LINEAR):
		if resolution == "sample":
			return self._grid_hv
		elif resolution == "full":
			if interpolation != INTER_LINEAR:
				raise ValueError("only INTER_LINEAR dense grids are built on the GPU")
			if "full" not in self._full:  # one array object per grid, like the reference's lru_cache
				self._full["full"] = _shim.grid_fill(self._grid_hv, self._grid_res)
			return self._full["full"]
		else:
			raise ValueError(resolution)

	@property
	def resolution(self):
		return self._grid_res

	@staticmethod
	def create(*args, **kwargs):
		factory = GridFactory(*args, **kwargs)
		return Grid(factory.grid_hv, factory.res)

	@staticmethod
	def open(path):
		"""dewarp.zip: data.npy (float32 node positions) + meta.json {"version", "cell", "shape"}
		(ref: dewarp.py:537-553, docs/formats.md:272-306)."""
		from .batch.core.fastzip import ZipReader
		if hasattr(path, "read"):
			members = dict(ZipReader(path).items())
		else:
			with open(path, "rb") as f:
				members = dict(ZipReader(f).items())
		meta = json.loads(members["meta.json"])
		nodes = np.load(io.BytesIO(members["data.npy"]), allow_pickle=False)
		return Grid(nodes.reshape(meta["shape"]), meta["cell"])

	def save(self, path, compression=zipfile.ZIP_DEFLATED):
		member = io.BytesIO()
		np.save(member, np.asarray(self._grid_hv, dtype=np.float32), allow_pickle=False)
		meta = dict(version=1, cell=self._grid_res, shape=self._grid_hv.shape)
		# method (deflate), member names, order and contents are the reference's; the deflate streams come from
		# libogdewarp's entropy-only writer (batch/core/fastzip.py): float32 node positions hold nothing for a string
		# matcher — zlib spends 14 ms per page finding that out and ends 2 % smaller
		from .batch.core.fastzip import ZipWriter
		if isinstance(path, (str, os.PathLike)):
			with open(path, "wb") as f, ZipWriter(f, compression) as zf:
				zf.writestr("data.npy", member.getvalue())
				zf.writestr("meta.json", json.dumps(meta))
		else:
			with ZipWriter(path, compression) as zf:
				zf.writestr("data.npy", member.getvalue())
				zf.writestr("meta.json", json.dumps(meta))

	@property
	def transformer(self):
		if self._transformer is None:
			r = self._grid_res
			# points("full")[::r, ::r] evaluated directly at those pixels (ref: dewarp.py:557-559)
			source = _shim.grid_points_strided(self._grid_hv, r, r)
			self._transformer = Transformer(source, r)
		return self._transformer

	@property
	def inverse_yx(self):
		def f(pts_yx):
			return _shim.grid_sample_bilinear(self._grid_hv, self._grid_res, pts_yx)
		return f

	@property
	def inverse(self):
		interp = self.inverse_yx

		def f(pts):
			return interp(np.flip(pts, axis=-1))
		return f


class Dewarper:
	"""ref: dewarp.py:579-627."""

	def __init__(self, im, grid, pixels=None, device_page=None, warped=None):
		"""pixels: the (H, W) uint8 array `im` was made from, if the caller still holds it (origami_amd.page.Page
		after decoding a scan itself): spares the 48 MB copy PIL makes on the way back to numpy.
		device_page: a callable giving the page resident in device memory (`_shim.DevicePage`, gray); `dewarped` and
		`dewarped_jpeg` then work from it and `im` may be None, with `warped` a callable that makes the PIL image
		should `dewarp_image` need its size (origami_amd.page: one decode and one upload per page and process)."""
		self._im_ = im
		self._warped = warped
		self._grid = grid
		self._dewarped = None
		self._pixels = pixels
		self._device_page = device_page

	@property
	def _im(self):
		if self._im_ is None and self._warped is not None:
			self._im_ = self._warped()
		return self._im_

	def _array(self):
		return self._pixels if self._pixels is not None else np.asarray(self._im)

	@property
	def grid(self):
		return self._grid

	@property
	def dewarped(self):
		if self._dewarped is None and self._device_page is not None:
			self._dewarped = PIL.Image.fromarray(self._device_page().dewarp(
				self._grid.points("sample"), self._grid.resolution, interpolation=INTER_LINEAR))
		if self._dewarped is None:
			self._dewarped = PIL.Image.fromarray(_shim.dewarp_u8(
				self._array(), self._grid.points("sample"), self._grid.resolution,
				interpolation=INTER_LINEAR))
		return self._dewarped

	def dewarped_jpeg(self, quality=75):
		"""The file `self.dewarped.save(path.jpg)` writes (ref: origami/batch/core/io.py:606-610), as bytes:
		dewarp and JPEG encoding both run on the device (ogd_dewarp_jpeg_gray), only the code stream comes
		back.  Gray pages only (Page converts to "L", ref: origami/core/page.py:84)."""
		if self._device_page is not None:
			return self._device_page().dewarp_jpeg(self._grid.points("sample"), self._grid.resolution, quality)
		if self._im.mode != "L":
			raise ValueError("dewarped_jpeg needs an 'L' page, got mode %s" % self._im.mode)
		return _shim.dewarp_jpeg_gray(
			self._array(), self._grid.points("sample"), self._grid.resolution, quality=quality)

	def dewarp_image(self, im, interpolation=INTER_LINEAR):
		"""Remap an image of any size (label map, RGB page) through the page's grid, scaled by
		the size ratio.  The reference multiplies its cached dense grid in place (dewarp.py:621-623),
		which leaks the scale into later users; here the scale is applied to a private copy inside
		the kernel, i.e. the behaviour of the reference's first call."""
		if self._device_page is not None:
			h0, w0 = self._device_page().shape[:2]
		else:
			w0, h0 = self._im.size
		w1, h1 = im.size
		return PIL.Image.fromarray(_shim.dewarp_u8(
			np.array(im), self._grid.points("sample"), self._grid.resolution,
			sx=np.float32(w1 / w0), sy=np.float32(h1 / h0), interpolation=interpolation))
