"""ORACLE - test infrastructure only (see oracle/__init__.py).  parity unpinned: the reference holds no
fixtures for line images; the inverse map is scipy's own RegularGridInterpolator, cv2.remap is oracle/cv.

CPU restatement of the per-line image extraction of a dewarped page:
origami/core/block.py:223-259 (Line.warped_grid, Line.dewarped_image) without the `column` clipping
(block.py:195-221, shapely segment intersection on the host, ahead of the lattice and unchanged by it).
"""
import numpy as np

from . import cv


def warped_grid(grid, p0, right, up, xsteps=None, ysteps=None, xres=1, yres=1):
	"""block.py:223-248.  grid: oracle.dewarp.Grid; p0, right, up: (x, y) in dewarped pixels."""
	p0, right, up = (np.asarray(v, dtype=np.float64) for v in (p0, right, up))
	if xsteps is None or ysteps is None:
		rough = warped_grid(grid, p0, right, up, xsteps=2, ysteps=2)
	if xsteps is None:
		xsteps = np.max(np.abs(rough[:, 0, 0] - rough[:, 1, 0]))
		xsteps = max(2, int(np.ceil(xsteps * xres)))
	if ysteps is None:
		ysteps = np.max(np.abs(rough[0, :, 1] - rough[1, :, 1]))
		ysteps = max(2, int(np.ceil(ysteps * yres)))
	ys = np.linspace([0, 0], up, ysteps)
	xs = np.linspace([0, 0], right, xsteps)
	lattice = (ys + p0)[:, np.newaxis] + xs[np.newaxis, :]
	lattice = np.flip(lattice, axis=-1)
	w = grid.inverse_yx(lattice.reshape((len(ys) * len(xs), 2)))
	return w.reshape((len(ys), len(xs), 2)).astype(np.float32)


def dewarped_image(warped_pixels, grid, p0, right, up, target_height=48, interpolation=cv.INTER_LINEAR):
	"""block.py:250-259."""
	wg = warped_grid(grid, p0, right, up, ysteps=target_height)
	pixels = cv.remap_u8(warped_pixels, wg, interpolation)
	return pixels[::-1, :]


def label_evidence_counts(labels, page_shape, grid, p0, right, up, res=0.5, minlength=0):
	"""ConfidenceSampler.__call__, batch/detect/lines.py:34-44: np.bincount of the label map remapped
	(nearest) through the line's half-resolution lattice scaled to the label map's size.
	page_shape: (H, W) of the warped page."""
	wg = warped_grid(grid, p0, right, up, xres=res, yres=res)
	h0, w0 = page_shape
	h1, w1 = labels.shape
	wg[:, :, 0] *= w1 / w0  # scale_grid(), lines.py:18-22 (float32 array, in place)
	wg[:, :, 1] *= h1 / h0
	remapped = cv.remap_u8(labels, wg, cv.INTER_NEAREST)
	return np.bincount(remapped.flatten(), minlength=minlength)
