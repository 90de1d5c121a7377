"""Artifacts of a page and their readers / writers — the part of origami/batch/core/io.py the dewarp
and line-export stages touch (same names, file names and skip / overwrite rules).

ref: origami/batch/core/io.py:25-62 (find_data_path, Stage, Artifact), :92-103 (DebuggingArtifact, Annotation),
:109-146 (read_contours), :166-224 (Regions, Lines), :227-373 (Reader), :375-387 (Input),
:411-416 (AtomicFileWriter), :507-620 (Writer, Output).
"""

import collections
import enum
import json
import os
import tempfile
import zipfile
from contextlib import contextmanager
from pathlib import Path

from ...geometry import parse_wkt


def find_data_path(page_path):
	return Path(page_path).with_suffix(".out")


class Stage(enum.Enum):
	WARPED = 0
	DEWARPED = 1
	AGGREGATE = 2
	RELIABLE = 3
	ANY = -1

	@property
	def is_dewarped(self):
		return self.value >= Stage.DEWARPED.value


class Artifact(enum.Enum):
	SEGMENTATION = ("segment.zip",)
	FLOW = ("flow.zip",)
	DEWARPING_TRANSFORM = ("dewarp.zip",)
	TABLES = ("tables.json",)
	RUNTIME = ("runtime.json",)
	CONTOURS = ("contours.%s.zip", {
		Stage.WARPED: 0,
		Stage.DEWARPED: 1,
		Stage.AGGREGATE: 2,
		Stage.RELIABLE: 3})
	LINES = ("lines.%s.zip", {
		Stage.WARPED: 0,
		Stage.RELIABLE: 3})

	def __init__(self, filename, stages=None):
		self._filename = filename
		self._stages = stages

	def filename(self, stage=None):
		s = self._filename
		if self._stages is not None:
			if stage is None:
				raise RuntimeError("need to specify stage for loading %s" % self)
			variant = self._stages.get(stage)
			if variant is None:
				raise RuntimeError("%s is not supported for stage %s" % (self, stage))
			s = s % str(variant)
		return s


class DebuggingArtifact:  # ref: io.py:92-97
	def __init__(self, filename):
		self._filename = filename

	def filename(self, stage=None):
		return self._filename


class Annotation(DebuggingArtifact):  # ref: io.py:100-103
	def __init__(self, name):
		super().__init__("annotation.%s.jpg" % name)


class PredictorType(enum.Enum):  # ref: origami/core/predict.py:136-138
	REGION = 1
	SEPARATOR = 2


Contours = collections.namedtuple("Contours", ["items", "meta"])


def read_contours(path, pred_type, open=open):
	"""[(path parts, Geom)] of one predictor type from a contours zip; ref: io.py:109-146."""
	items = []
	pred_meta = dict()
	with open(path, "rb") as f:
		with zipfile.ZipFile(f, "r") as zf:
			meta = json.loads(zf.read("meta.json"))
			if meta["version"] > 1:
				predictions = dict((x["name"], x) for x in meta["predictions"])
			else:
				predictions = meta

			def wanted(parts):
				return PredictorType[predictions[parts[0]]["type"]] == pred_type

			for name in zf.namelist():
				if name.endswith("/meta.json"):
					parts = tuple(name.split("/"))
					if wanted(parts):
						pred_meta[tuple(parts[:-1])] = json.loads(zf.read(name))
				if not name.endswith(".wkt"):
					continue
				parts = tuple(name.rsplit(".", 1)[0].split("/"))
				if wanted(parts):
					items.append((parts, parse_wkt(zf.read(name).decode("utf8"))))
	return Contours(items, pred_meta)


class Regions:
	"""ref: io.py:166-189.  `page` may be a zero-argument callable (see Block)."""

	def __init__(self, path, page, stage, open=open):
		from ...block import Block
		self._blocks = dict(
			(parts, Block(page, polygon, stage))
			for parts, polygon in read_contours(path, PredictorType.REGION, open=open).items)

	@property
	def by_path(self):
		return self._blocks


class Lines:
	"""ref: io.py:192-224 — one JSON per line (Line.info) under the path of its block."""

	def __init__(self, path, regions, stage=Stage.WARPED, open=open):
		from ...block import Line
		blocks = regions.by_path
		assert all(block.stage == stage for block in blocks.values())
		self._meta = None
		lines = dict()
		with open(path, "rb") as lf:
			with zipfile.ZipFile(lf, "r") as zf:
				for name in zf.namelist():
					if name == "meta.json":
						self._meta = json.loads(zf.read(name))
						continue
					if not name.endswith(".json"):
						raise RuntimeError("illegal file %s in %s." % (name, path))
					parts = tuple(name.rsplit(".", 1)[0].split("/"))
					block = blocks[tuple(parts[:3])]
					lines[parts] = Line(block, **json.loads(zf.read(name)))
		self._lines = lines

	@property
	def meta(self):
		return self._meta

	@property
	def min_confidence(self):
		return self.meta.get("min_confidence", 0.5)

	@property
	def by_path(self):
		return self._lines


class Reader:
	def __init__(self, artifacts, stage, page_path, take_any=False, open=open):
		artifacts = set(artifacts)
		if Artifact.LINES in artifacts:
			artifacts.add(Artifact.CONTOURS)  # io.py:231-232
		if Artifact.CONTOURS in artifacts:
			artifacts.add(Artifact.SEGMENTATION)  # io.py:233-234
		if stage and stage.is_dewarped and Artifact.CONTOURS in artifacts:
			artifacts.add(Artifact.DEWARPING_TRANSFORM)
		self._artifacts = artifacts
		self._stage = stage
		self._page_path = Path(page_path)
		self._data_path = find_data_path(page_path)
		self._take_any = take_any
		self._open = open
		self._cache = {}

	@property
	def data_path(self):
		return self._data_path

	@property
	def paths(self):
		return [self.path(a) for a in self._artifacts]

	def path(self, artifact):
		if artifact not in self._artifacts:
			raise RuntimeError("read on undeclared %s" % artifact)
		return self._data_path / artifact.filename(self._stage)

	def fix_inconsistent(self):
		pass

	def is_ready(self):
		return True if self._take_any else all(p.exists() for p in self.paths)

	@property
	def missing(self):
		return [p for p in self.paths if not p.exists()]

	@property
	def page(self):
		if "page" not in self._cache:
			from ...page import Page
			grid = self.dewarping_transform if self._stage.is_dewarped else None
			self._cache["page"] = Page(self._page_path, grid)
		return self._cache["page"]

	@property
	def regions(self):
		if "regions" not in self._cache:
			self._cache["regions"] = Regions(
				self.path(Artifact.CONTOURS), lambda: self.page, self._stage, open=self._open)
		return self._cache["regions"]

	@property
	def lines(self):
		if "lines" not in self._cache:
			self._cache["lines"] = Lines(
				self.path(Artifact.LINES), self.regions, self._stage, open=self._open)
		return self._cache["lines"]

	def load_json(self, artifact):
		with open(self.path(artifact), "r") as f:
			return json.loads(f.read())

	@property
	def tables(self):
		if "tables" not in self._cache:
			self._cache["tables"] = self.load_json(Artifact.TABLES)
		return self._cache["tables"]

	@property
	def dewarping_transform(self):
		if "grid" not in self._cache:
			from ...dewarp import Grid
			self._cache["grid"] = Grid.open(self.path(Artifact.DEWARPING_TRANSFORM))
		return self._cache["grid"]

	@property
	def flow(self):
		return zipfile.ZipFile(self.path(Artifact.FLOW), "r")


class Input:
	def __init__(self, *artifacts, stage=None, take_any=False):
		assert all(isinstance(x, Artifact) for x in artifacts)
		self._artifacts = set(artifacts)
		self._stage = stage
		self._take_any = take_any

	def instantiate(self, processor, file_writer, **kwargs):
		return Reader(
			self._artifacts, self._stage, take_any=self._take_any, open=processor.lock_or_open, **kwargs)


class AtomicFileWriter:
	"""Write to a temporary file next to the target, then rename (atomicwrites' behaviour,
	ref: io.py:411-416)."""

	def __init__(self, overwrite):
		self._overwrite = overwrite

	@property
	def overwrite(self):
		return self._overwrite

	@contextmanager
	def __call__(self, path, mode):
		path = Path(path)
		fd, tmp = tempfile.mkstemp(prefix=".tmp", dir=str(path.parent))
		try:
			with os.fdopen(fd, mode) as f:
				yield f
			if not self._overwrite and path.exists():
				raise FileExistsError(str(path))
			os.replace(tmp, path)
		except BaseException:
			try:
				os.unlink(tmp)
			except OSError:
				pass
			raise


class Writer:
	def __init__(self, artifacts, stage, page_path, processor, file_writer):
		self._artifacts = artifacts
		self._stage = stage
		self._page_path = page_path
		self._data_path = find_data_path(page_path)
		self._processor = processor
		self._write = file_writer

	@property
	def compression(self):
		return zipfile.ZIP_DEFLATED

	@property
	def data_path(self):
		return self._data_path

	@property
	def paths(self):
		return [self.path(a) for a in self._artifacts]

	def path(self, artifact):
		if artifact not in self._artifacts:
			raise RuntimeError("write on undeclared %s" % artifact)
		return self._data_path / artifact.filename(self._stage)

	def fix_inconsistent(self):
		# a half-written output set is removed so that the page is redone (io.py:533-540)
		if self._write.overwrite:
			return
		e = [p.exists() for p in self.paths]
		if any(e) and not all(e):
			for p in self.paths:
				if p.exists():
					os.remove(p)

	def is_ready(self):
		return self._write.overwrite or not any(p.exists() for p in self.paths)

	@property
	def missing(self):
		return []

	@contextmanager
	def write_zip_file(self, artifact):
		with self._write(self.path(artifact), mode="wb") as f:
			with zipfile.ZipFile(f, "w", self.compression) as zf:
				yield zf

	@contextmanager
	def contours(self, copy_meta_from=None):
		with self.write_zip_file(Artifact.CONTOURS) as f:
			if copy_meta_from is not None:
				path = copy_meta_from.path(Artifact.CONTOURS)
				with zipfile.ZipFile(path, "r") as zf:
					f.writestr("meta.json", zf.read("meta.json"))
					for name in zf.namelist():
						if name.endswith("/meta.json"):
							f.writestr(name, zf.read(name))
			yield f

	@contextmanager
	def dewarping_transform(self):
		with self._write(self.path(Artifact.DEWARPING_TRANSFORM), mode="wb") as f:
			yield f

	def annotation(self, image):
		"""ref: io.py:606-610 (image.save to annotation.<name>.jpg); written through the atomic writer here."""
		assert len(self._artifacts) == 1
		annotation = list(self._artifacts)[0]
		assert isinstance(annotation, Annotation)
		with self._write(self.path(annotation), mode="wb") as f:
			if isinstance(image, (bytes, bytearray, memoryview)):
				f.write(image)  # an already encoded file (Page.dewarped_jpeg: the device-side encoder)
			else:
				image.save(f, format="JPEG")


class Output:
	def __init__(self, *artifacts, stage=None):
		assert all(isinstance(x, (Artifact, DebuggingArtifact)) for x in artifacts)
		self._artifacts = set(artifacts)
		self._stage = stage

	def instantiate(self, **kwargs):
		return Writer(self._artifacts, self._stage, **kwargs)
