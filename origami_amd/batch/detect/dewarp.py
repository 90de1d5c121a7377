#!/usr/bin/env python3
"""The `dewarp` batch stage: flow samples + warped contours -> dewarp.zip + contours.1.zip.

    python -m origami_amd.batch.detect.dewarp DATA_PATH [--grid-cell-size 25] [--region-area 0] [--gpus N]

ref: origami/batch/detect/dewarp.py:16-112 (same options, artifacts and member names).  All contour
rings of a page go through the forward transformer in ONE launch of the wavefront-per-ring kernel
(K3) instead of one scipy call per ring (ref: dewarp.py:26).
"""

import logging
from pathlib import Path

import click

from ... import _shim
from ...dewarp import Grid, Samples
from ...geometry import parse_wkt, to_wkt, to_wkt_many, without_repeats
from ...repair import RepairError
from ..core.io import Artifact, Input, Output, Stage
from ..core.processor import Processor


def dewarped_contours(warped, transformer, min_areas):
	"""Yields (member name, WKT bytes) of every contour that survives the warp; ref: dewarp.py:16-39."""
	names, geoms = [], []
	parsed = warped.contour_geoms  # the regions among them were parsed when the page's blocks were read
	for name, data in warped.contour_members.items():
		if not name.endswith(".wkt"):
			continue
		geom = parsed.get(name)
		if geom is None:
			geom = parsed[name] = parse_wkt(data.decode("utf8"))
		assert not geom.is_empty
		names.append(name)
		geoms.append(geom)

	# one ragged launch for all rings of the page
	rings = [r for g in geoms for r in g.rings]
	warped_rings = transformer.transform_rings(rings)
	# ... and one for the quadratic part of every ring's validity test (ref: dewarp.py:35)
	crossing = _shim.rings_self_cross([without_repeats(r) for r in warped_rings])
	k = 0
	kept_names, kept = [], []
	for name, warped_geom in zip(names, geoms):
		n = len(warped_geom.rings)
		geom = warped_geom.with_rings(warped_rings[k:k + n])
		ring_flags = crossing[k:k + n]
		k += n
		path = tuple(name.rsplit(".", 1)[0].split("/"))
		if geom.is_empty or geom.area < min_areas.get(path[0], 0):
			logging.warning("lost contour %s (A=%.1f) during dewarping." % (path, warped_geom.area))
			continue
		if geom.geom_type not in ("Polygon", "LineString"):
			logging.error("dewarped contour %s is %s" % (name, geom.geom_type))
		if not geom.valid_given(ring_flags):
			try:
				geom = geom.buffer0()  # shapely's geom.buffer(0), ref: dewarp.py:36 (origami_amd/repair.py)
			except RepairError as e:
				logging.error("contour %s is invalid and could not be repaired (%s); written as it is." % (name, e))
			if not geom.is_valid:
				logging.error("invalid geom %s", to_wkt(geom))
		kept_names.append(name)
		kept.append(geom)
	for name, text in zip(kept_names, to_wkt_many(kept)):
		yield name, text.encode("utf8")


class DewarpProcessor(Processor):
	def __init__(self, options):
		super().__init__(options)
		self._options = dict(options)

	@property
	def processor_name(self):
		return "origami.batch.detect.dewarp"  # the reference's __loader__.name (runtime.json key)

	def artifacts(self):
		return [
			("warped", Input(Artifact.CONTOURS, Artifact.FLOW, stage=Stage.WARPED)),
			("output", Output(Artifact.DEWARPING_TRANSFORM, Artifact.CONTOURS, stage=Stage.DEWARPED)),
		]

	def process(self, page_path: Path, warped, output):
		blocks = warped.regions.by_path
		if not blocks:
			return

		page = warped.page
		with warped.flow as zf:
			samples_h = Samples.open(zf, "h")
			samples_v = Samples.open(zf, "v")

		grid = Grid.create(page, samples_h, samples_v, grid_res=self._options["grid_cell_size"])

		min_areas = dict(regions=grid.geometry.rel_area(self._options["region_area"]), separators=0)

		with output.contours(copy_meta_from=warped) as zf:
			for name, data in dewarped_contours(warped, grid.transformer, min_areas=min_areas):
				zf.writestr(name, data)

		with output.dewarping_transform() as f:
			grid.save(f)


@click.command()
@click.argument('data_path', type=click.Path(exists=True), required=True)
@click.option(
	'--grid-cell-size', type=int, default=25,
	help="grid cell size used for dewarping (smaller is better, but takes longer).")
@click.option('--region-area', type=float, default=0, help="Ignore regions below this relative size.")
@Processor.options
def dewarp(data_path, **kwargs):
	""" Dewarp documents in DATA_PATH. """
	processor = DewarpProcessor(kwargs)
	processor.traverse(data_path)


if __name__ == "__main__":
	dewarp()
