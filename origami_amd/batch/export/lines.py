#!/usr/bin/env python3
"""`python -m origami_amd.batch.export.lines DATA_PATH` — the dewarped line images of every page as PNG files in
one zip per page.

Shared with the reference's stage of the same name (origami/batch/export/lines.py:18-105): the command and its
options, the inputs (lines + tables of the RELIABLE stage), the output file name
`images.lines[.dewarped|.deskewed][.binarized].zip`, member names `<path joined by "/" or "-">.png`, PNG written
with optimize=True."""

import io
import zipfile
from pathlib import Path

import click

from ..core.io import Artifact, DebuggingArtifact, Input, Output, Stage
from ..core.lines import LineExtractor
from ..core.processor import Processor


def output_name(options):
	"""Name of the zip for a combination of rendering options."""
	if not options["do_not_dewarp"]:
		geometry = ["dewarped"]
	elif not options["do_not_deskew"]:
		geometry = ["deskewed"]
	else:
		geometry = []
	tone = ["binarized"] if options.get("binarize") else []
	return ".".join(["images", "lines", *geometry, *tone, "zip"])


def png_bytes(image):
	data = io.BytesIO()
	image.save(data, format="png", optimize=True)
	return data.getvalue()


def _pixels_of(image):
	"""The array a line image was made from, if origami_amd.block.dewarped_line_images left it on the image (a PIL
	operation on the image gives a new object without it), else a copy out of PIL."""
	import numpy as np
	a = getattr(image, "_ogd_pixels", None)
	if a is not None and a.dtype == np.uint8 and a.shape[:2] == (image.height, image.width) and a.flags.c_contiguous:
		return a
	return np.asarray(image)


def png_files(items):
	"""(path, PNG file, its CRC-32 or None) for the (path, PIL image) items of a page, in their order.  The reference saves every line
	with PIL (ref: origami/batch/export/lines.py:62-67: `im.save(f, format='png', optimize=True)`, zlib at level 9):
	13 ms per 48 x 3000 line image, 4 s per broadsheet page and 96 % of this stage once the line images come from
	one kernel launch (profiles/r03/v_export_lines_profile_before.log).  8-bit grey and RGB images — what the stage
	produces unless it binarizes — go through the library's PNG writer in one call (ogd_png_encode_many: same pixels
	for any reader, files ~2 % larger than PIL's, ~15x faster); anything else is saved by PIL as before."""
	import numpy as np
	from ... import _shim
	items = list(items)
	files = [None] * len(items)
	for mode in ("L", "RGB"):
		picked = [k for k, (_, im) in enumerate(items) if im.mode == mode and im.width > 0 and im.height > 0]
		if picked:
			for k, item in zip(picked, _shim.png_encode_many([_pixels_of(items[k][1]) for k in picked], with_crc=True, views=True)):
				files[k] = item
	for k, (path, im) in enumerate(items):
		yield (path,) + (files[k] if files[k] is not None else (png_bytes(im), None))


class LineExtractionProcessor(Processor):
	def __init__(self, options):
		super().__init__(options)
		self._options = dict(options)
		self._output = DebuggingArtifact(output_name(self._options))

	@property
	def output(self):
		return self._output

	def artifacts(self):
		return [
			("input", Input(Artifact.LINES, Artifact.TABLES, stage=Stage.RELIABLE)),
			("output", Output(self._output)),
		]

	def process(self, page_path: Path, input, output):
		extract = LineExtractor(input.tables, self._options["line_height"], self._options)
		joiner = "-" if self._options["flat"] else "/"
		with output.write_zip_file(self._output) as zf:
			for path, data, crc in png_files(extract(input.lines.by_path)):
				# a PNG file is a deflate stream already: stored, not deflated again (the reference's ZIP_DEFLATED
				# spends its time finding that out; any zip reader takes either)
				zf.writestr(joiner.join(path) + ".png", data, compress_type=zipfile.ZIP_STORED, crc=crc)


@click.command()
@Processor.options
@LineExtractor.options
@click.option('-h', '--line-height', default=48, type=int, help='height of line images in pixels')
@click.option('-f', '--flat', default=False, is_flag=True, help='generate flat hierarchy in zip file')
@click.argument('data_path', type=click.Path(exists=True), required=True)
def export_lines(data_path, **kwargs):
	""" Export line images from all document images in DATA_PATH. Needs
	information from previous batches. """
	LineExtractionProcessor(kwargs).traverse(data_path)


if __name__ == "__main__":
	export_lines()
