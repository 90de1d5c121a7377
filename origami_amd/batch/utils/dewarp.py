#!/usr/bin/env python3
"""`python -m origami_amd.batch.utils.dewarp DATA_PATH` — write the dewarped image of every page as
`annotation.dewarped.jpg` (ref: origami/batch/utils/dewarp.py:11-42; same artifact name and options).

The one stage of the reference that does nothing but the pixel hot loop: Reader.page opens the page with
its dewarping transform (ref: origami/batch/core/io.py:275-280, origami/core/page.py:82-88) and the page
goes through the fused grid-fill + remap kernel and the device-side JPEG encoder (ogd_dewarp_jpeg_gray): the
file is the one PIL writes for `page.dewarped`, byte for byte, and only its ~0.14 B/px cross PCIe."""

import click

from ..core import io
from ..core.processor import Processor


class DewarpProcessor(Processor):
	# runtime.json key: the reference's __loader__.name for this stage
	processor_name = "origami.batch.utils.dewarp"

	def __init__(self, options):
		super().__init__(options)
		self._options = dict(options)

	def artifacts(self):
		"""Reads dewarp.zip (through the page it opens), writes the annotation image "dewarped"."""
		source = io.Input(io.Artifact.DEWARPING_TRANSFORM, stage=io.Stage.DEWARPED)
		return [("dewarped", source), ("output", io.Output(io.Annotation("dewarped")))]

	def prefetch(self, page_path, dewarped, output):
		"""The next page's grid (dewarp.zip) and scan — file bytes of a JPEG the device decodes, decoded pixels in pinned
		memory for everything else — while this page is on the device."""
		dewarped.page.prefetch()

	def process(self, page_path, dewarped, output):
		# the reference saves the PIL image `dewarped.page.dewarped`; the same file comes encoded from the device
		output.annotation(dewarped.page.dewarped_jpeg())


@click.command(help="Create dewarped images for all document images in DATA_PATH.")
@click.argument("data_path", type=click.Path(exists=True), required=True)
@Processor.options
def dewarp(data_path, **options):
	DewarpProcessor(options).traverse(data_path)


if __name__ == "__main__":
	dewarp()
