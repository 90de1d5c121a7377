"""From the lines of a page to their images, for OCR and export.

Schema shared with the reference (origami/batch/core/lines.py:51-149): the classes `LineRewriter` and
`LineExtractor`, the three CLI options, and the naming of table cells — a line `k` of a table block
`<block>.<division>.*.*` that is cut at the block's column separators becomes the parts
`<block>.<division>.<k + 1>.<column>` / `0`, columns counted from 1.  Everything else is this build's: the
reference renders one line at a time on a thread pool (one small cv2.remap each, lines.py:148); here the whole
page is planned first — which lines survive, which column intervals they are cut into — and all images of the
plan come out of one launch (origami_amd.block.dewarped_line_images).
"""

import logging
from typing import NamedTuple, Optional, Tuple

import click

from ...block import dewarped_line_images


class LinePart(NamedTuple):
	path: Tuple[str, ...]  # member path of the image
	line: object  # the Line it is cut from
	column: Optional[Tuple[Optional[float], Optional[float]]]  # x interval in dewarped pixels, None = whole line


def _table_cell_path(path, column):
	"""Path of column `column` (1-based) of the table row stored at `path`."""
	if column < 1:
		raise AssertionError("columns count from 1")
	predictor, label, block_id, row = path
	fields = block_id.split(".")
	if len(fields) != 4:
		raise RuntimeError("%s is not a valid table path" % str(path))
	cell = "%s.%s.%d.%d" % (fields[0], fields[1], int(row) + 1, column)
	return predictor, label, cell, "0"


class LineRewriter:
	"""Cuts the rows of table blocks at the column separators found in tables.json."""

	def __init__(self, tables):
		separators = (tables or {}).get("columns", {})
		self._separators = {tuple(block.split("/")): list(xs) for block, xs in separators.items()}

	def parts_of(self, path, line):
		xs = self._separators.get(tuple(path[:3]))
		if xs is None:
			yield LinePart(path, line, None)
			return
		edges = [None, *xs, None]  # open at both ends of the row
		for k in range(len(edges) - 1):
			yield LinePart(_table_cell_path(path, k + 1), line, (edges[k], edges[k + 1]))

	def __call__(self, lines):
		return [tuple(part) for path, line in lines.items() for part in self.parts_of(path, line)]


_UNSUPPORTED = dict(
	binarize="--binarize is not part of the GPU line path",
	do_not_dewarp="--do-not-dewarp: deskewed / warped line images are not part of the GPU line path")


class LineExtractor:
	def __init__(self, tables, line_height, options, min_confidence=0.5):
		if line_height is None:
			raise AssertionError("line_height is required")
		if str(options.get("binarize") or "").strip():
			raise click.UsageError(_UNSUPPORTED["binarize"])
		if options.get("do_not_dewarp"):
			raise click.UsageError(_UNSUPPORTED["do_not_dewarp"])
		self._line_height = line_height
		self._min_confidence = min_confidence
		self._rewriter = LineRewriter(tables)

	@staticmethod
	def options(f):
		for decorate in (
				click.option('--do-not-deskew', default=False, is_flag=True, help='do not deskew line images'),
				click.option('--do-not-dewarp', default=False, is_flag=True, help='do not dewarp line images'),
				click.option(
					'--binarize', type=str, default="",
					help="binarization algorithm to use (e.g. otsu), or empty if none")):
			f = decorate(f)
		return f

	def plan(self, lines, ignored=None):
		"""The parts to render, in the order of `lines`: lines of ignored (predictor, label) classes and lines
		below the confidence threshold are dropped, table rows are cut into cells."""
		parts = []
		for path, line in lines.items():
			if ignored is not None and ignored(tuple(path[:2])):
				continue
			if line.confidence < self._min_confidence:
				logging.info("skipping line %s with confidence %.1f" % (str(path), line.confidence))
				continue
			parts.extend(self._rewriter.parts_of(path, line))
		return parts

	def __call__(self, lines, ignored=None):
		"""{path: Line} -> [(path, PIL.Image)]"""
		parts = self.plan(lines, ignored)
		if not parts:
			return []
		if not all(part.line.block.stage.is_dewarped for part in parts):
			raise AssertionError("line images are cut from dewarped stages only")
		page = parts[0].line.block.page
		images = dewarped_line_images(page, [(part.line, part.column) for part in parts], self._line_height)
		return [(part.path, image) for part, image in zip(parts, images)]
