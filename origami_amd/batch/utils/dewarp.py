#!/usr/bin/env python3
"""`python -m origami_amd.batch.utils.dewarp DATA_PATH` — write the dewarped image of every page as
`annotation.dewarped.jpg` (ref: origami/batch/utils/dewarp.py:11-42; same artifact name and options).

The one stage of the reference that does nothing but the pixel hot loop: Reader.page opens the page with
its dewarping transform (ref: origami/batch/core/io.py:275-280, origami/core/page.py:82-88) and the page
goes through the fused grid-fill + remap kernel (ogd_dewarp_u8) on first use of `page.dewarped`."""

from pathlib import Path

import click

from ..core.io import Annotation, Artifact, Input, Output, Stage
from ..core.processor import Processor


class DewarpProcessor(Processor):
	def __init__(self, options):
		super().__init__(options)
		self._options = dict(options)

	@property
	def processor_name(self):
		return "origami.batch.utils.dewarp"  # the reference's __loader__.name (runtime.json key)

	def artifacts(self):
		return [
			("dewarped", Input(Artifact.DEWARPING_TRANSFORM, stage=Stage.DEWARPED)),
			("output", Output(Annotation("dewarped"))),
		]

	def process(self, page_path: Path, dewarped, output):
		output.annotation(dewarped.page.dewarped)


@click.command()
@click.argument('data_path', type=click.Path(exists=True), required=True)
@Processor.options
def dewarp(data_path, **kwargs):
	""" Create dewarped images for all document images in DATA_PATH. """
	processor = DewarpProcessor(kwargs)
	processor.traverse(data_path)


if __name__ == "__main__":
	dewarp()
