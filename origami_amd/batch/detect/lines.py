"""Segmentation evidence under detected lines (ref: origami/batch/detect/lines.py:18-54).

Of the line detection stage only ConfidenceSampler runs remaps: per line a half-resolution lattice, scaled to
the label map, remapped with nearest neighbour and counted.  Here the label maps stay resident on the device
and `sample_many` counts every line of a page in one launch per predictor (ogd_line_label_counts); the
remapped images are never materialised.  The Tesseract-driven line detector itself is out of scope.
"""

import numpy as np

from ... import _shim
from ...block import line_lattices


class ConfidenceSampler:
	def __init__(self, blocks, segmentation):
		self._predictions = dict()
		for p in segmentation.predictions:
			self._predictions[p.name] = p
		self._page = list(blocks.values())[0].page
		self._page_shape = tuple(reversed(self._page.warped.size))
		self._device_labels = dict()

	def _labels_on_device(self, name):
		if name not in self._device_labels:
			self._device_labels[name] = _shim.DevicePage(
				np.ascontiguousarray(self._predictions[name].labels, dtype=np.uint8))
		return self._device_labels[name]

	def sample_many(self, items, res=0.5):
		"""items: [(path, line)] -> [evidence dict], one launch per predictor name."""
		out = [None] * len(items)
		by_predictor = dict()
		for k, (path, line) in enumerate(items):
			by_predictor.setdefault(path[0], []).append(k)
		grid = self._page.dewarper.grid
		for name, ks in by_predictor.items():
			predictor = self._predictions[name]
			geo, xsteps, ysteps = line_lattices(
				grid, [(items[k][1], None) for k in ks], None, xres=res, yres=res)
			h0, w0 = self._page_shape
			h1, w1 = predictor.labels.shape
			counts = self._labels_on_device(name).line_label_counts(
				grid.points("sample"), grid.resolution, geo, xsteps, ysteps,
				np.float32(w1 / w0), np.float32(h1 / h0))
			for k, c in zip(ks, counts):
				evidence = dict()
				sum_all = np.sum(c)
				if sum_all > 0:
					for cls in predictor.classes:
						evidence["%s/%s" % (name, cls.name)] = c[cls.value] / sum_all
				out[k] = evidence
		return out

	def __call__(self, path, line, res=0.5):
		return self.sample_many([(path, line)], res=res)[0]

	def close(self):
		for page in self._device_labels.values():
			page.close()
		self._device_labels = dict()
