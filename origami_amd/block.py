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


def _evidence(confidence):
	"""[(label path string, value)] of a line's per-label confidences without the BACKGROUND entries;
	None if the confidence is a plain number (block.py:63-102 repeat this filter three times)."""
	if not isinstance(confidence, dict):
		return None
	return [(key, value) for key, value in confidence.items() if not key.endswith("/BACKGROUND")]


class Line:
	"""ref: block.py:35-333 — a text line in dewarped coordinates: origin `p`, baseline vector `right`, height
	vector `up`; confidence (a number, or a dict label path -> evidence from ConfidenceSampler); the dewarped
	image."""

	def __init__(self, block, p, right, up, tesseract_data=None, wkt=None, text_area=None, confidence=1):
		self._block = block
		self._p, self._right, self._up = (np.array(v) for v in (p, right, up))
		self._wkt, self._tesseract_data = wkt, tesseract_data
		self._confidence = confidence

	block = property(lambda self: self._block)
	angle = property(lambda self: math.atan2(self._right[1], self._right[0]))
	length = property(lambda self: np.linalg.norm(self._right))
	height = property(lambda self: np.linalg.norm(self._up))

	@property
	def confidence(self):
		found = _evidence(self._confidence)
		if found is None:
			return self._confidence
		return max((value for _, value in found), default=0)

	def _best_label(self):
		"""(label path tuple, evidence) of the strongest non-background label (the first of equals), or None."""
		found = _evidence(self._confidence)
		if not found:
			return None
		key, value = found[int(np.argmax([v for _, v in found]))]
		return tuple(key.split("/")), value

	@property
	def predicted_path(self):
		best = self._best_label()
		return best[0] if best else None

	def predicted_path_error(self, path):
		"""How much more evidence the strongest label has than `path` (block.py:91-102); 0 if they agree."""
		best = self._best_label()
		if best is None or best[0] == path:
			return 0
		return best[1] - self._confidence.get("/".join(path), 0)

	def update_confidence(self, confidence):
		self._confidence = confidence

	@property
	def info(self):
		# key order as the reference writes it (lines are stored as JSON)
		record = {name: getattr(self, "_" + name).tolist() for name in ("p", "right", "up")}
		for name in ("wkt", "confidence", "tesseract_data"):
			record[name] = getattr(self, "_" + name)
		return record

	def _position(self, xres, column):
		"""(origin, right, up, xres) of the line, or of its part between the abscissae column = (x0, x1), either
		of which may be None for "the line's own end" (block.py:195-221)."""
		origin, right = self._p, self._right
		if column is None:
			return origin, right, self._up, xres
		end = origin + right
		lo, hi = sorted((origin[0], end[0]))
		x0, x1 = (own if asked is None else asked for asked, own in zip(column, (lo, hi)))
		cut0 = _cut_baseline(origin, end, x0, default=origin)
		cut1 = _cut_baseline(origin, end, x1, default=end)
		return cut0, cut1 - cut0, self._up, xres * ((x1 - x0) / (hi - lo))

	def _steps(self, rough, xsteps, ysteps, xres, yres):
		"""Lattice size from the warped positions `rough` (2, 2, 2) of the line's corners: the longer of the two
		edges per direction times the resolution, at least 2 (block.py:230-236)."""
		def count(given, extent, res):
			return given if given is not None else max(2, int(np.ceil(extent * res)))
		return (count(xsteps, np.max(np.abs(rough[:, 0, 0] - rough[:, 1, 0])), xres),
			count(ysteps, np.max(np.abs(rough[0, :, 1] - rough[1, :, 1])), yres))

	def warped_grid(self, xsteps=None, ysteps=None, xres=1, yres=1, column=None):
		"""(ysteps, xsteps, 2) float32 positions (x, y) on the warped page.  ref: block.py:223-248."""
		origin, right, up, xres = self._position(xres, column)
		if xsteps is None or ysteps is None:
			xsteps, ysteps = self._steps(self.warped_grid(xsteps=2, ysteps=2), xsteps, ysteps, xres, yres)
		rows = origin + np.linspace([0, 0], up, ysteps)
		columns = np.linspace([0, 0], right, xsteps)
		lattice_yx = (rows[:, None] + columns[None, :])[..., ::-1]
		warped = self.block.page.dewarper.grid.inverse_yx(lattice_yx.reshape(-1, 2))
		return warped.reshape(ysteps, xsteps, 2).astype(np.float32)

	def dewarped_image(self, target_height=48, column=None, interpolation=INTER_LINEAR):
		"""ref: block.py:250-259."""
		assert self.block.stage.is_dewarped
		return dewarped_line_images(self.block.page, [(self, column)], target_height, interpolation)[0]

	def image(self, target_height=48, column=None, dewarped=True, deskewed=True, binarizer=None):
		"""ref: block.py:129-143; only the dewarped variant is built."""
		if not dewarped:
			raise NotImplementedError("only dewarped line images are produced on the GPU")
		im = self.dewarped_image(target_height, column=column)
		return binarizer(im) if binarizer else im

	def warped_path(self, resolution=1):
		"""(centre line on the warped page, median height).  ref: block.py:261-267."""
		assert self.block.stage.is_dewarped
		lower, upper = self.warped_grid(ysteps=2, xres=resolution)
		return (lower + upper) / 2, abs(np.median(np.linalg.norm(upper - lower, axis=-1)))

	def dewarped_height(self, dewarper):
		"""Height of a line given in WARPED coordinates once it is dewarped.  ref: block.py:326-333."""
		assert not self._block.stage.is_dewarped
		ends = np.array([self._p, self._p + self._up])
		tx, ty = dewarper.grid.transformer(ends[:, 0], ends[:, 1])
		return np.linalg.norm(np.array([tx[1] - tx[0], ty[1] - ty[0]]))


class Block:
	"""The part of origami.core.block.Block a Line needs: page, polygon and stage (block.py:377-402).
	`page` may be a zero-argument callable: readers pass one so that listing the regions of a page
	does not decode (and dewarp) its image."""

	def __init__(self, page, polygon, stage):
		self._page = page
		self.image_space_polygon = polygon
		self.stage = stage

	@property
	def page(self):
		if callable(self._page):
			self._page = self._page()
		return self._page


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
	# np.linspace([0, 0], v, 2) is [[0, 0], v] exactly, so the four corners are p0, p0 + right, up + p0 and
	# (up + p0) + right in the reference's order of additions — for all lines at once (two linspace calls per line
	# were 15 us each, thousands of lines per page)
	P0 = np.array([line._p for line, _ in items], dtype=np.float64).reshape(n, 2)
	R = np.array([line._right for line, _ in items], dtype=np.float64).reshape(n, 2)
	U = np.array([line._up for line, _ in items], dtype=np.float64).reshape(n, 2)
	rough = np.empty((n, 2, 2, 2), dtype=np.float64)
	rough[:, 0, 0] = P0
	rough[:, 0, 1] = P0 + R
	rough[:, 1, 0] = U + P0
	rough[:, 1, 1] = (U + P0) + R
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
	out = []
	for a in images:
		im = PIL.Image.fromarray(a)
		im._ogd_pixels = a  # the export stage encodes from the array and spares the copy back out of PIL
		out.append(im)
	return out
