#!/usr/bin/env python3
"""`python -m origami_amd.batch.export.lines DATA_PATH` — the dewarped line images of every page as PNG files in
one zip per page.

Shared with the reference's stage of the same name (origami/batch/export/lines.py:18-105): the command and its
options, the inputs (lines + tables of the RELIABLE stage), the output file name
`images.lines[.dewarped|.deskewed][.binarized].zip`, member names `<path joined by "/" or "-">.png`, PNG written
with optimize=True."""

import io
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
			for path, image in extract(input.lines.by_path):
				zf.writestr(joiner.join(path) + ".png", png_bytes(image))


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
