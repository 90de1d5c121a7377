"""Page artifacts on disk: which files a stage reads and writes, under which names, and when a page counts as
ready, done or half-done.

This is the slice of the reference's artifact layer that the dewarp and line-export stages touch
(ref: origami/batch/core/io.py — names and rules: :25-62 stages and file names, :92-103 annotations, :109-146 the
contour zip, :192-224 the line zip, :231-236 implied inputs, :261-265 / :542-543 readiness, :533-540 half-written
output sets, :411-416 atomic writes, :565-577 meta members copied first, :606-610 annotation images).  What a
drop-in must share with the reference is the schema — enum members, file names, member order, error messages —
and that is all this module shares; the code is organised around one table of file-name patterns and one common
base for the reading and the writing side.
"""

import enum
import json
import os
import tempfile
import zipfile
from contextlib import contextmanager
from pathlib import Path
from typing import NamedTuple

from ...geometry import parse_wkt_many
from . import fastzip


def find_data_path(page_path):
	"""`<page>.out/`, the directory of a page's artifacts."""
	return Path(page_path).with_suffix(".out")


class Stage(enum.Enum):
	WARPED = 0
	DEWARPED = 1
	AGGREGATE = 2
	RELIABLE = 3
	ANY = -1

	@property
	def is_dewarped(self):
		return self.value > Stage.WARPED.value


# file-name pattern of every artifact; a "%s" is filled with the number of the stage the file belongs to
_PATTERNS = dict(
	SEGMENTATION="segment.zip", FLOW="flow.zip", DEWARPING_TRANSFORM="dewarp.zip", TABLES="tables.json",
	RUNTIME="runtime.json", CONTOURS="contours.%s.zip", LINES="lines.%s.zip")
# stages at which a staged artifact exists
_STAGED = dict(
	CONTOURS=(Stage.WARPED, Stage.DEWARPED, Stage.AGGREGATE, Stage.RELIABLE),
	LINES=(Stage.WARPED, Stage.RELIABLE))


class Artifact(enum.Enum):
	SEGMENTATION = enum.auto()
	FLOW = enum.auto()
	DEWARPING_TRANSFORM = enum.auto()
	TABLES = enum.auto()
	RUNTIME = enum.auto()
	CONTOURS = enum.auto()
	LINES = enum.auto()

	def filename(self, stage=None):
		pattern = _PATTERNS[self.name]
		stages = _STAGED.get(self.name)
		if stages is None:
			return pattern
		if stage is None:
			raise RuntimeError("need to specify stage for loading %s" % self)
		if stage not in stages:
			raise RuntimeError("%s is not supported for stage %s" % (self, stage))
		return pattern % stage.value


class DebuggingArtifact:
	"""A file that is a by-product, not an input of any stage."""

	def __init__(self, filename):
		self._name = filename

	def filename(self, stage=None):
		return self._name


class Annotation(DebuggingArtifact):
	def __init__(self, name):
		super().__init__("annotation.%s.jpg" % name)


class PredictorType(enum.Enum):  # ref: origami/core/predict.py:136-138
	REGION = 1
	SEPARATOR = 2


# ---- zip members ---------------------------------------------------------------------------------------

def _member_path(name):
	"""'regions/TEXT/3.wkt' -> ('regions', 'TEXT', '3')"""
	return tuple(name.rpartition(".")[0].split("/"))


class Contours(NamedTuple):
	items: list  # [(member path, Geom)]
	meta: dict  # {(predictor, label): contents of that folder's meta.json}


def read_members(path, open=open):
	"""{member name: bytes} of a zip of small members, in archive order (fastzip: one read, no per-member cost)."""
	with open(path, "rb") as f:
		return dict(fastzip.ZipReader(f).items())


def read_contours(path, pred_type, open=open, members=None, parsed=None):
	"""The geometries of one predictor type (regions or separators) in a contours zip, in member order, with the
	per-label meta.json files of the same predictors.  members: the archive's contents if the caller holds them
	already; parsed: a {member name: Geom} cache shared between callers that read the same archive."""
	if members is None:
		members = read_members(path, open=open)
	if parsed is None:
		parsed = {}
	head = json.loads(members["meta.json"])
	# version 1 keyed the predictors by name at the top level, version 2 lists them
	kinds = head if head["version"] <= 1 else {p["name"]: p for p in head["predictions"]}
	of_type = {name for name, p in kinds.items() if isinstance(p, dict) and PredictorType[p["type"]] == pred_type}
	mine = [name for name in members if name.partition("/")[0] in of_type]
	todo = [name for name in mine if name.endswith(".wkt") and name not in parsed]
	for name, geom in zip(todo, parse_wkt_many([members[name] for name in todo])):  # one call for the page
		parsed[name] = geom
	geoms, meta = [], {}
	for name in mine:
		if name.endswith("/meta.json"):
			meta[tuple(name.split("/")[:-1])] = json.loads(members[name])
		elif name.endswith(".wkt"):
			geoms.append((_member_path(name), parsed[name]))
	return Contours(geoms, meta)


class Regions:
	"""Blocks of a page by member path.  `page` may be a zero-argument callable (see Block)."""

	def __init__(self, path, page, stage, open=open, members=None, parsed=None):
		from ...block import Block
		found = read_contours(path, PredictorType.REGION, open=open, members=members, parsed=parsed)
		self._blocks = {where: Block(page, polygon, stage) for where, polygon in found.items}

	@property
	def by_path(self):
		return self._blocks


class Lines:
	"""The lines zip: `meta.json` plus one JSON per line (Line.info) filed under its block's path."""

	def __init__(self, path, regions, stage=Stage.WARPED, open=open):
		from ...block import Line
		blocks = regions.by_path
		if any(b.stage != stage for b in blocks.values()):
			raise AssertionError("blocks and lines of different stages")
		self._meta, self._lines = None, {}
		with open(path, "rb") as f:
			zf = fastzip.ZipReader(f)
			for name in zf.namelist():
				if not name.endswith(".json"):
					raise RuntimeError("illegal file %s in %s." % (name, path))
				record = json.loads(zf.read(name))
				if name == "meta.json":
					self._meta = record
				else:
					where = _member_path(name)
					self._lines[where] = Line(blocks[where[:3]], **record)

	@property
	def meta(self):
		return self._meta

	@property
	def min_confidence(self):
		return self.meta.get("min_confidence", 0.5)

	@property
	def by_path(self):
		return self._lines


# ---- the files of one page, as a stage declared them -----------------------------------------------------

class _PageFiles:
	"""Common to both sides: the declared artifacts of one page at one stage and where they live."""

	_verb = "access"

	def __init__(self, artifacts, stage, page_path):
		self._artifacts = set(artifacts)
		self._stage = stage
		self._page_path = Path(page_path)
		self._data_path = find_data_path(page_path)

	@property
	def data_path(self):
		return self._data_path

	def path(self, artifact):
		if artifact not in self._artifacts:
			raise RuntimeError("%s on undeclared %s" % (self._verb, artifact))
		return self._data_path / artifact.filename(self._stage)

	@property
	def paths(self):
		return [self.path(a) for a in self._artifacts]

	def _present(self):
		return [p.exists() for p in self.paths]


class Reader(_PageFiles):
	_verb = "read"

	def __init__(self, artifacts, stage, page_path, take_any=False, open=open):
		wanted = set(artifacts)
		# lines hang off blocks, blocks off the segmentation's label names; dewarped contours need their grid
		if Artifact.LINES in wanted:
			wanted.add(Artifact.CONTOURS)
		if Artifact.CONTOURS in wanted:
			wanted.add(Artifact.SEGMENTATION)
			if stage is not None and stage.is_dewarped:
				wanted.add(Artifact.DEWARPING_TRANSFORM)
		super().__init__(wanted, stage, page_path)
		self._take_any = take_any
		self._open = open
		self._loaded = {}

	def _once(self, key, load):
		if key not in self._loaded:
			self._loaded[key] = load()
		return self._loaded[key]

	def fix_inconsistent(self):
		pass  # inputs are never touched

	def is_ready(self):
		return self._take_any or all(self._present())

	@property
	def missing(self):
		return [p for p in self.paths if not p.exists()]

	@property
	def page(self):
		def load():
			from ...page import Page
			return Page(self._page_path, self.dewarping_transform if self._stage.is_dewarped else None)
		return self._once("page", load)

	@property
	def contour_members(self):
		"""{member name: bytes} of this page's contours zip, read once (every stage that takes CONTOURS reads
		the regions, and the dewarp stage then every member again; ref: io.py:275-280)."""
		return self._once("contour_members", lambda: read_members(self.path(Artifact.CONTOURS), open=self._open))

	@property
	def contour_geoms(self):
		"""{member name: Geom} of the .wkt members parsed so far (shared by `regions` and the dewarp stage)."""
		return self._once("contour_geoms", dict)

	@property
	def regions(self):
		return self._once("regions", lambda: Regions(
			self.path(Artifact.CONTOURS), lambda: self.page, self._stage, open=self._open,
			members=self.contour_members, parsed=self.contour_geoms))

	@property
	def lines(self):
		return self._once("lines", lambda: Lines(
			self.path(Artifact.LINES), self.regions, self._stage, open=self._open))

	def load_json(self, artifact):
		return json.loads(self.path(artifact).read_text())

	@property
	def tables(self):
		return self._once("tables", lambda: self.load_json(Artifact.TABLES))

	@property
	def dewarping_transform(self):
		def load():
			from ...dewarp import Grid
			return Grid.open(self.path(Artifact.DEWARPING_TRANSFORM))
		return self._once("grid", load)

	@property
	def flow(self):
		return zipfile.ZipFile(self.path(Artifact.FLOW), "r")

	@property
	def flow_samples(self):
		"""(samples_h, samples_v) of flow.zip, read once (ref: origami/batch/detect/dewarp.py:72-74)."""
		def load():
			from ...dewarp import Samples
			with self.flow as zf:
				return Samples.open(zf, "h"), Samples.open(zf, "v")
		return self._once("flow_samples", load)


class Input:
	"""Declaration of what a stage reads; the processor turns it into a Reader per page."""

	def __init__(self, *artifacts, stage=None, take_any=False):
		if not all(isinstance(a, Artifact) for a in artifacts):
			raise AssertionError("inputs are Artifacts")
		self._spec = dict(artifacts=set(artifacts), stage=stage, take_any=take_any)

	def instantiate(self, processor, file_writer, **kwargs):
		return Reader(open=processor.lock_or_open, **self._spec, **kwargs)


class AtomicFileWriter:
	"""Files appear complete or not at all: written under a temporary name in the target's directory, renamed
	on success (what the reference gets from `atomicwrites`)."""

	def __init__(self, overwrite):
		self._overwrite = overwrite

	@property
	def overwrite(self):
		return self._overwrite

	@contextmanager
	def __call__(self, path, mode):
		path = Path(path)
		handle, scratch = tempfile.mkstemp(prefix=".tmp", dir=str(path.parent))
		done = False
		try:
			with os.fdopen(handle, mode) as f:
				yield f
			if path.exists() and not self._overwrite:
				raise FileExistsError(str(path))
			os.replace(scratch, path)
			done = True
		finally:
			if not done:
				try:
					os.unlink(scratch)
				except OSError:
					pass


class Writer(_PageFiles):
	_verb = "write"

	def __init__(self, artifacts, stage, page_path, processor, file_writer):
		super().__init__(artifacts, stage, page_path)
		self._processor = processor
		self._write = file_writer

	@property
	def compression(self):
		return zipfile.ZIP_DEFLATED

	def fix_inconsistent(self):
		"""Some but not all outputs exist (an interrupted run): remove them, so the page is done again."""
		if self._write.overwrite:
			return
		present = self._present()
		if any(present) and not all(present):
			for p, there in zip(self.paths, present):
				if there:
					os.remove(p)

	def is_ready(self):
		return self._write.overwrite or not any(self._present())

	@property
	def missing(self):
		return []

	@contextmanager
	def write_zip_file(self, artifact):
		# deflate level 1: WKT and JSON text shrinks as far as at the default level (485 vs 479 KB for a
		# broadsheet page's contours) in two thirds of the time; members, order and contents are unchanged
		# fastzip.ZipWriter: the same archive bytes as zipfile would lay out, assembled without its per-member cost
		with self._write(self.path(artifact), mode="wb") as f, \
				fastzip.ZipWriter(f, self.compression, compresslevel=1) as zf:
			yield zf

	@contextmanager
	def contours(self, copy_meta_from=None):
		"""The stage's contours zip, opened for writing; the meta members of `copy_meta_from`'s contours zip
		(meta.json first, then every <predictor>/<label>/meta.json) are carried over before anything else."""
		with self.write_zip_file(Artifact.CONTOURS) as out:
			if copy_meta_from is not None:
				src = copy_meta_from.contour_members
				folders = [n for n in src if n.endswith("/meta.json")]
				for name in ["meta.json"] + folders:
					out.writestr(name, src[name])
			yield out

	@contextmanager
	def dewarping_transform(self):
		with self._write(self.path(Artifact.DEWARPING_TRANSFORM), mode="wb") as f:
			yield f

	def annotation(self, image):
		"""annotation.<name>.jpg: a PIL image (saved as JPEG, as the reference does) or the bytes of an already
		encoded file (Page.dewarped_jpeg: the device-side encoder)."""
		(target,) = self._artifacts
		if not isinstance(target, Annotation):
			raise AssertionError("annotation() needs an Annotation output")
		with self._write(self.path(target), mode="wb") as f:
			if isinstance(image, (bytes, bytearray, memoryview)):
				f.write(image)
			else:
				image.save(f, format="JPEG")


class Output:
	"""Declaration of what a stage writes; the processor turns it into a Writer per page."""

	def __init__(self, *artifacts, stage=None):
		if not all(isinstance(a, (Artifact, DebuggingArtifact)) for a in artifacts):
			raise AssertionError("outputs are Artifacts")
		self._spec = dict(artifacts=set(artifacts), stage=stage)

	def instantiate(self, **kwargs):
		return Writer(**self._spec, **kwargs)
