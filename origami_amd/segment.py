"""Reading `segment.zip`: label maps and their classes (ref: origami/core/segment.py:17-67,
origami/core/predict.py:136-146, 371-397).  Only what ConfidenceSampler needs."""

import enum
import io
import json
import zipfile

import numpy as np
import PIL.Image

from .batch.core.io import PredictorType


class Prediction:
	def __init__(self, type, name, labels, classes):
		self._type = type
		self._name = name
		self._labels = labels
		self._classes = classes
		self._background = self._classes["BACKGROUND"]

	@property
	def type(self):
		return self._type

	@property
	def name(self):
		return self._name

	@property
	def background_label(self):
		return self._background

	@property
	def labels(self):
		return self._labels

	@property
	def classes(self):
		return self._classes


class Segmentation:
	def __init__(self, predictions):
		self._predictions = tuple(predictions)

	@property
	def predictions(self):
		return self._predictions

	@property
	def size(self):
		shape = self._predictions[0].labels.shape
		return tuple(reversed(list(shape)[:2]))

	@staticmethod
	def open(path):
		predictions = []
		with zipfile.ZipFile(path, "r") as zf:
			tasks = [name.rsplit(".", 1)[0] for name in zf.namelist() if name.endswith(".png")]
			for task in tasks:
				with io.BytesIO(zf.read(task + ".png")) as f:
					im = PIL.Image.open(f)
					im.load()
				meta = json.loads(zf.read(task + ".json"))
				t = PredictorType[meta["type"]]
				label_enum = "RegionLabel" if t == PredictorType.REGION else "SeparatorLabel"
				classes = enum.Enum(label_enum, meta["classes"])
				predictions.append(Prediction(t, meta["name"], np.array(im), classes))
		return Segmentation(predictions)
