"""Reading `segment.zip`: label maps and their classes (ref: origami/core/segment.py:17-67,
origami/core/predict.py:136-146, 371-397).  Only what ConfidenceSampler needs: per predictor a palette PNG
`<name>.png` with the labels and `<name>.json` with {"type", "name", "classes"}."""

import collections
import enum
import io
import json
import zipfile

import numpy as np
import PIL.Image

from .batch.core.io import PredictorType

_LABEL_ENUMS = {PredictorType.REGION: "RegionLabel", PredictorType.SEPARATOR: "SeparatorLabel"}


class Prediction(collections.namedtuple("Prediction", "type name labels classes")):
	"""One predictor's label map: `labels` (h, w) uint8, `classes` the Enum of its label values."""
	__slots__ = ()

	@property
	def background_label(self):
		return self.classes["BACKGROUND"]


def _read_prediction(zf, stem):
	meta = json.loads(zf.read(stem + ".json"))
	kind = PredictorType[meta["type"]]
	with PIL.Image.open(io.BytesIO(zf.read(stem + ".png"))) as im:
		labels = np.array(im)
	return Prediction(kind, meta["name"], labels, enum.Enum(_LABEL_ENUMS[kind], meta["classes"]))


class Segmentation:
	def __init__(self, predictions):
		self.predictions = tuple(predictions)

	@property
	def size(self):
		"""(width, height) of the label maps."""
		h, w = self.predictions[0].labels.shape[:2]
		return (w, h)

	@staticmethod
	def open(path):
		with zipfile.ZipFile(path, "r") as zf:
			stems = [n[:-len(".png")] for n in zf.namelist() if n.endswith(".png")]
			return Segmentation(_read_prediction(zf, s) for s in stems)
