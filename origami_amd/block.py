"""Text lines of a dewarped page and their images (ref: origami/core/block.py:35-333).

The reference cuts every line out of the WARPED page with its own small cv2.remap: a 48 x L lattice in
dewarped space, mapped back through Grid.inverse_yx (block.py:223-259), thousands of times per page on
an 8-thread pool (batch/core/lines.py:148).  Here `Line.dewarped_image` keeps that interface, and
`dewarped_line_images` does all lines of a page in two device calls: one inverse-map call for the
2 x 2 rough lattices that size the images, one launch that produces every pixel of every line
(ogd_line_images_u8) from the page resident on the device (Page.device_page).

Only the dewarped path is GPU work and only it is mirrored; deskewed / masked images (cv2.warpAffine,
rasterised polygon masks; block.py:129-193) belong to other stages.
"""

import math

import numpy as np
import PIL.Image

from . import _shim
from ._shim import INTER_LINEAR


def _cut_baseline(p0, p1, x, default):
	"""intersect_segments(bottom, vertical at x) of block.py:27-32,210-216: the point of the segment p0-p1
	at abscissa x, or `default` if there is no single such point."""
	(ax, ay), (bx, by) = p0, p1
	if ax == bx or not (min(ax, bx) <= x <= max(ax, bx)):
		return default
	t = (x - ax) / (bx - ax)
	return np.array([x, ay + t * (by - ay)])


class Line:
	"""ref: block.py:35-333 — geometry (p, right, up), confidence and the dewarped image."""

	def __init__(self, block, p, right, up, tesseract_data=None, wkt=None, text_area=None, confidence=1):
		self._tesseract_data = tesseract_data
		self._block = block
		self._p = np.array(p)
		self._right = np.array(right)
		self._up = np.array(up)
		self._wkt = wkt
		self._confidence = confidence

	@property
	def block(self):
		return self._block

	@property
	def angle(self):
		return math.atan2(self._right[1], self._right[0])

	@property
	def confidence(self):
		if not isinstance(self._confidence, dict):
			return self._confidence
		values = [v for k, v in self._confidence.items() if not k.endswith("/BACKGROUND")]
		return max(values) if values else 0

	@property
	def predicted_path(self):
		if not isinstance(self._confidence, dict):
			return None
		items = [(k, v) for k, v in self._confidence.items() if not k.endswith("/BACKGROUND")]
		if items:
			return tuple(items[int(np.argmax([x[1] for x in items]))][0].split("/"))
		return None

	def predicted_path_error(self, path):
		# block.py:91-102
		if not isinstance(self._confidence, dict):
			return 0
		items = [(k, v) for k, v in self._confidence.items() if not k.endswith("/BACKGROUND")]
		if items:
			i = int(np.argmax([x[1] for x in items]))
			if tuple(items[i][0].split("/")) == path:
				return 0
			return items[i][1] - self._confidence.get("/".join(path), 0)
		return 0

	def update_confidence(self, confidence):
		self._confidence = confidence

	@property
	def length(self):
		return np.linalg.norm(self._right)

	@property
	def height(self):
		return np.linalg.norm(self._up)

	@property
	def info(self):
		return dict(
			p=self._p.tolist(), right=self._right.tolist(), up=self._up.tolist(), wkt=self._wkt,
			confidence=self._confidence, tesseract_data=self._tesseract_data)

	def _position(self, xres, column):
		# block.py:195-221
		p0, right, up = self._p, self._right, self._up
		if column is not None:
			p1 = p0 + right
			px0, px1 = min(p0[0], p1[0]), max(p0[0], p1[0])
			x0, x1 = column
			if x0 is None:
				x0 = px0
			if x1 is None:
				x1 = px1
			q0 = _cut_baseline(p0, p1, x0, default=p0)
			q1 = _cut_baseline(p0, p1, x1, default=p1)
			p0, p1 = q0, q1
			right = p1 - p0
			xres *= (x1 - x0) / (px1 - px0)
		return p0, right, up, xres

	def _steps(self, rough, xsteps, ysteps, xres, yres):
		# block.py:230-236; rough: (2, 2, 2) warped positions of the lattice corners
		if xsteps is None:
			xsteps = np.max(np.abs(rough[:, 0, 0] - rough[:, 1, 0]))
			xsteps = max(2, int(np.ceil(xsteps * xres)))
		if ysteps is None:
			ysteps = np.max(np.abs(rough[0, :, 1] - rough[1, :, 1]))
			ysteps = max(2, int(np.ceil(ysteps * yres)))
		return xsteps, ysteps

	def warped_grid(self, xsteps=None, ysteps=None, xres=1, yres=1, column=None):
		"""(ysteps, xsteps, 2) float32 positions (x, y) on the warped page.  ref: block.py:223-248."""
		p0, right, up, xres = self._position(xres, column)
		if xsteps is None or ysteps is None:
			rough = self.warped_grid(xsteps=2, ysteps=2)
			xsteps, ysteps = self._steps(rough, xsteps, ysteps, xres, yres)
		ys = np.linspace([0, 0], up, ysteps)
		xs = np.linspace([0, 0], right, xsteps)
		dewarped_grid = (ys + p0)[:, np.newaxis] + xs[np.newaxis, :]
		dewarped_grid = np.flip(dewarped_grid, axis=-1)
		inv = self.block.page.dewarper.grid.inverse_yx
		warped_grid = inv(dewarped_grid.reshape((len(ys) * len(xs), 2)))
		return warped_grid.reshape((len(ys), len(xs), 2)).astype(np.float32)

	def dewarped_image(self, target_height=48, column=None, interpolation=INTER_LINEAR):
		"""ref: block.py:250-259."""
		assert self.block.stage.is_dewarped
		return dewarped_line_images(
			self.block.page, [(self, column)], target_height, interpolation)[0]

	def image(self, target_height=48, column=None, dewarped=True, deskewed=True, binarizer=None):
		# block.py:129-143
		if not dewarped:
			raise NotImplementedError("only dewarped line images are produced on the GPU")
		im = self.dewarped_image(target_height, column=column)
		if binarizer:
			im = binarizer(im)
		return im

	def warped_path(self, resolution=1):
		# block.py:261-267
		assert self.block.stage.is_dewarped
		warped_grid = self.warped_grid(ysteps=2, xres=resolution)
		height = np.median(np.linalg.norm(warped_grid[1] - warped_grid[0], axis=-1))
		return np.mean(warped_grid, axis=0), abs(height)

	def dewarped_height(self, dewarper):
		# block.py:326-333
		assert not self._block.stage.is_dewarped
		p0, up = self._p, self._up
		p1 = p0 + up
		tx, ty = dewarper.grid.transformer(*np.array([p0, p1]).transpose())
		q0, q1 = np.array([tx, ty]).transpose()
		return np.linalg.norm(q1 - q0)


class Block:
	"""The part of origami.core.block.Block a Line needs: page, polygon and stage (block.py:377-402).
	`page` may be a zero-argument callable: readers pass one so that listing the regions of a page
	does not decode (and dewarp) its image."""

	def __init__(self, page, polygon, stage):
		self._page = page
		self._polygon = polygon
		self._stage = stage

	@property
	def page(self):
		if callable(self._page):
			self._page = self._page()
		return self._page

	@property
	def image_space_polygon(self):
		return self._polygon

	@property
	def stage(self):
		return self._stage


def line_lattices(grid, items, target_height=None, xres=1, yres=1):
	"""The lattice of every (line, column) item: (n, 6) float64 rows p0, right, up and the step counts,
	with one device call for all rough 2 x 2 lattices (block.py:226-236)."""
	n = len(items)
	geo = np.empty((n, 6), dtype=np.float64)
	xr = np.empty(n, dtype=np.float64)
	for k, (line, column) in enumerate(items):
		p0, right, up, r = line._position(xres, column)
		geo[k, 0:2], geo[k, 2:4], geo[k, 4:6], xr[k] = p0, right, up, r
	# the rough lattice is taken WITHOUT the column (block.py:227 calls warped_grid(xsteps=2, ysteps=2))
	rough = np.empty((n, 2, 2, 2), dtype=np.float64)
	for k, (line, _) in enumerate(items):
		p0, right, up = line._p, line._right, line._up
		ys = np.linspace([0, 0], up, 2)
		xs = np.linspace([0, 0], right, 2)
		rough[k] = (ys + p0)[:, np.newaxis] + xs[np.newaxis, :]
	rough = grid.inverse_yx(np.flip(rough, axis=-1).reshape(-1, 2)).reshape(n, 2, 2, 2).astype(np.float32)
	xsteps = np.empty(n, dtype=np.int32)
	ysteps = np.empty(n, dtype=np.int32)
	for k, (line, _) in enumerate(items):
		xsteps[k], ysteps[k] = line._steps(rough[k], None, target_height, xr[k], yres)
	return geo, xsteps, ysteps


def dewarped_line_images(page, items, target_height=48, interpolation=INTER_LINEAR):
	"""Line.dewarped_image for many lines of one page.  items: [(line, column or None)] -> [PIL.Image]."""
	if not items:
		return []
	grid = page.dewarper.grid
	geo, xsteps, ysteps = line_lattices(grid, items, target_height)
	images = page.device_page.line_images(
		grid.points("sample"), grid.resolution, geo, xsteps, ysteps, interpolation)
	return [PIL.Image.fromarray(im) for im in images]
