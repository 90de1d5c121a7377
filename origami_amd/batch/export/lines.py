#!/usr/bin/env python3
"""`python -m origami_amd.batch.export.lines DATA_PATH` — export the dewarped line images of every page
(ref: origami/batch/export/lines.py:18-105; same options, same images.lines.dewarped.zip)."""

import io
from pathlib import Path

import click

from ..core.io import Artifact, DebuggingArtifact, Input, Output, Stage
from ..core.lines import LineExtractor
from ..core.processor import Processor


class LineExtractionProcessor(Processor):
	def __init__(self, options):
		super().__init__(options)
		self._options = dict(options)
		self._output = None

	@property
	def output(self):
		if self._output is None:
			name = ["images", "lines"]
			if not self._options["do_not_dewarp"]:
				name.append("dewarped")
			elif not self._options["do_not_deskew"]:
				name.append("deskewed")
			if self._options.get("binarize", False):
				name.append("binarized")
			name.append("zip")
			self._output = DebuggingArtifact(".".join(name))
		return self._output

	def artifacts(self):
		return [
			("input", Input(Artifact.LINES, Artifact.TABLES, stage=Stage.RELIABLE)),
			("output", Output(self.output)),
		]

	def process(self, page_path: Path, input, output):
		lines = input.lines.by_path
		extractor = LineExtractor(input.tables, self._options["line_height"], self._options)
		images = extractor(lines)
		zip_sep = "-" if self._options["flat"] else "/"
		with output.write_zip_file(self.output) as zf:
			for stem, im in images:
				with io.BytesIO() as f:
					im.save(f, format='png', optimize=True)
					data = f.getvalue()
				zf.writestr("%s.png" % zip_sep.join(stem), data)


@click.command()
@Processor.options
@LineExtractor.options
@click.option('-h', '--line-height', default=48, type=int, help='height of line images in pixels')
@click.option('-f', '--flat', default=False, is_flag=True, help='generate flat hierarchy in zip file')
@click.argument('data_path', type=click.Path(exists=True), required=True)
def export_lines(data_path, **kwargs):
	""" Export line images from all document images in DATA_PATH. Needs
	information from previous batches. """
	processor = LineExtractionProcessor(kwargs)
	processor.traverse(data_path)


if __name__ == "__main__":
	export_lines()
