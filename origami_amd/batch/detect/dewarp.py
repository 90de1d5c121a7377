#!/usr/bin/env python3
"""The `dewarp` batch stage: flow samples + warped contours -> dewarp.zip + contours.1.zip.

    python -m origami_amd.batch.detect.dewarp DATA_PATH [--grid-cell-size 25] [--region-area 0] [--gpus N]

ref: origami/batch/detect/dewarp.py:16-112 (same options, artifacts and member names).  All contour
rings of a page go through the forward transformer in ONE launch of the wavefront-per-ring kernel
(K3) instead of one scipy call per ring (ref: dewarp.py:26).
"""

import logging
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import click
import numpy as np

from ... import _shim
from ...dewarp import Grid
from ...geometry import holes_inside, parse_wkt_many, ring_table, rings_valid_table, to_wkt
from ...repair import RepairError
from ..core.io import Artifact, Input, Output, Stage
from ..core.processor import Processor


def dewarped_contours(warped, transformer, min_areas):
	"""Yields (member name, WKT bytes) of every contour that survives the warp; ref: dewarp.py:16-39.

	The page's rings live in one table (geometry.ring_table): one launch warps them (K3 ragged), one launch answers
	the quadratic part of every ring's validity test (ogd_rings_self_cross), areas and coordinate text come from
	one host call each; Python touches a contour only to decide its fate, and builds a Geom only for the few
	that need a repair."""
	names, geoms = [], []
	parsed = warped.contour_geoms  # the regions among them were parsed when the page's blocks were read
	members = warped.contour_members
	todo = [name for name in members if name.endswith(".wkt") and name not in parsed]
	for name, geom in zip(todo, parse_wkt_many([members[name] for name in todo])):
		parsed[name] = geom
	for name in members:
		if name.endswith(".wkt"):
			assert not parsed[name].is_empty
			names.append(name)
			geoms.append(parsed[name])

	flat, offsets = ring_table([r for g in geoms for r in g.rings])
	warped_flat = transformer.transform_table(flat, offsets)
	ring_ok, finite, distinct = rings_valid_table(warped_flat, offsets)  # ref: dewarp.py:35, ring by ring
	area = np.abs(_shim.rings_area_signed(warped_flat, offsets)).tolist()
	texts = _shim.wkt_ring_texts_table(warped_flat, offsets)
	n_pts = np.diff(offsets).tolist()
	ring_ok, finite, distinct = ring_ok.tolist(), finite.tolist(), distinct.tolist()

	k = 0
	for name, warped_geom in zip(names, geoms):
		n = len(warped_geom.rings)
		r0, r1 = k, k + n
		k += n
		kind = warped_geom.geom_type
		path = tuple(name.rsplit(".", 1)[0].split("/"))
		if kind == "LineString":
			a = 0.0
			valid = n_pts[r0] >= 2 and finite[r0] and distinct[r0] >= 2
		else:
			a, r, holes = 0.0, r0, False
			for part in warped_geom.parts:  # shell minus holes, part by part (GEOS' Polygon::getArea)
				a += area[r] - sum(area[r + 1:r + len(part)])
				holes = holes or len(part) > 1
				r += len(part)
			valid = all(ring_ok[r0:r1])
		if n_pts[r0] == 0 or a < min_areas.get(path[0], 0):
			logging.warning("lost contour %s (A=%.1f) during dewarping." % (path, warped_geom.area))
			continue
		if kind not in ("Polygon", "LineString"):
			logging.error("dewarped contour %s is %s" % (name, kind))
		if valid and kind != "LineString" and holes:
			r = r0
			for part in warped_geom.parts:
				valid = valid and holes_inside([warped_flat[offsets[q]:offsets[q + 1]] for q in range(r, r + len(part))])
				r += len(part)
		if not valid:
			geom = warped_geom.with_rings([warped_flat[offsets[r]:offsets[r + 1]] for r in range(r0, r1)])
			try:
				geom = geom.buffer0()  # shapely's geom.buffer(0), ref: dewarp.py:36 (origami_amd/repair.py)
			except RepairError as e:
				logging.error("contour %s is invalid and could not be repaired (%s); written as it is." % (name, e))
			if not geom.is_valid:
				logging.error("invalid geom %s", to_wkt(geom))
			text = to_wkt(geom)
		elif kind == "LineString":
			text = "LINESTRING " + texts[r0]
		elif kind == "Polygon":
			text = "POLYGON (" + ", ".join(texts[r0:r1]) + ")"
		else:
			chunks, r = [], r0
			for part in warped_geom.parts:
				chunks.append("(" + ", ".join(texts[r:r + len(part)]) + ")")
				r += len(part)
			text = "MULTIPOLYGON (" + ", ".join(chunks) + ")"
		yield name, text.encode("utf8")


class DewarpProcessor(Processor):
	def __init__(self, options):
		super().__init__(options)
		self._options = dict(options)
		self._saver = None

	@property
	def processor_name(self):
		return "origami.batch.detect.dewarp"  # the reference's __loader__.name (runtime.json key)

	def artifacts(self):
		return [
			("warped", Input(Artifact.CONTOURS, Artifact.FLOW, stage=Stage.WARPED)),
			("output", Output(Artifact.DEWARPING_TRANSFORM, Artifact.CONTOURS, stage=Stage.DEWARPED)),
		]

	def _grid(self, warped):
		"""The page's node grid (ref: dewarp.py:75-76), built once per Reader."""
		def create():
			samples_h, samples_v = warped.flow_samples
			return Grid.create(warped.page, samples_h, samples_v, grid_res=self._options["grid_cell_size"])
		return warped._once("created_grid", create)

	def prefetch(self, page_path, warped, output):
		"""The next page's half of the work that depends on nothing but its own inputs, while this page's contours are
		warped and written: the contour archive read, inflated and its regions parsed, the flow samples, the page's
		header (its size is all this stage asks of it), the two angle fields and the march of the node grid (~12 of a
		page's ~24 ms; the library's host code and its launches release the GIL)."""
		if warped.regions.by_path:
			self._grid(warped)

	def process(self, page_path: Path, warped, output):
		blocks = warped.regions.by_path
		if not blocks:
			return

		grid = self._grid(warped)

		# dewarp.zip is 600 KB of float32 that deflate chews on for 14 ms (zlib releases the GIL): it is written by
		# a second thread while this one builds the transformer and warps the contours (most of that time is spent
		# inside libogdewarp, which releases the GIL too).  Either file appears complete or not at all, and a page
		# with only one of them is redone (Writer.fix_inconsistent), so the order of the two does not matter.
		def save_grid():
			with output.dewarping_transform() as f:
				grid.save(f)

		if self._saver is None:
			self._saver = ThreadPoolExecutor(max_workers=1, thread_name_prefix="dewarp-save")
		saving = self._saver.submit(save_grid)
		try:
			min_areas = dict(regions=grid.geometry.rel_area(self._options["region_area"]), separators=0)
			with output.contours(copy_meta_from=warped) as zf:
				for name, data in dewarped_contours(warped, grid.transformer, min_areas=min_areas):
					zf.writestr(name, data)
		except BaseException:
			# The reference writes the contours first and never writes the grid after a contour failure (ref:
			# dewarp.py:84-90): take the grid back, and let the saver's own failure, if any, not mask this one.
			try:
				saving.result()
			except Exception:
				logging.exception("writing %s failed as well." % output.path(Artifact.DEWARPING_TRANSFORM))
			try:
				output.path(Artifact.DEWARPING_TRANSFORM).unlink()
			except OSError:
				pass
			raise
		saving.result()


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
