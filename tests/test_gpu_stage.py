"""The dewarp batch stage end to end on the GPU (BASELINE.json configs[0] plumbing, scaled down):
synthetic page dir -> `dewarp` CLI -> dewarp.zip + contours.1.zip + runtime.json, checked against the
CPU oracle on the same inputs.  ref: origami/batch/detect/dewarp.py:42-112."""

import io
import json
import zipfile

import numpy as np
import pytest
from click.testing import CliRunner

from oracle import dewarp as odw
from origami_amd import geometry, synth
from origami_amd.batch.detect.dewarp import dewarp
from origami_amd.dewarp import Grid
from origami_amd.page import Page

pytestmark = pytest.mark.gpu


def _assert_warped_like_oracle(g0, g1, tf):
	"""g0: a contour as read from contours.0.zip, g1: the stage's output for it.  Every ring must be the
	oracle's warp of the input ring to 1e-3 px — unless that warped geometry is invalid (a ring that meets itself:
	synthetic star polygons rounded to label-map pixels sometimes do), in which case the stage must have written its
	buffer(0) repair (ref: batch/detect/dewarp.py:35-38): a valid geometry that holds exactly what GEOS' depth
	rule keeps (checked at random points away from the outline)."""
	from oracle import rings as orings
	want = odw.transform_rings(tf, g0.rings)
	trimmed = [geometry.without_repeats(r) for r in want]
	broken = g0.geom_type == "Polygon" and any(len(r) >= 4 and orings.ring_self_crosses(r) for r in trimmed)
	if not broken:
		assert g1.geom_type == g0.geom_type and len(g1.rings) == len(g0.rings)
		for w, got in zip(want, g1.rings):
			assert np.abs(w - got).max() < 1e-3
		return False
	assert g1.geom_type in ("Polygon", "MultiPolygon") and g1.is_valid
	rng = np.random.default_rng(len(want[0]))
	lo, hi = want[0].min(axis=0) - 3, want[0].max(axis=0) + 3
	for p in rng.uniform(lo, hi, size=(150, 2)):
		seg = want[0]
		near = min(np.hypot(*(seg[:-1] + t * (seg[1:] - seg[:-1]) - p).T).min() for t in np.linspace(0, 1, 101))
		if near > 0.5:
			assert orings.point_in_polygons(g1.parts, p) == (orings.buffer0_depth(want, p) >= 1)
	return True


def test_dewarp_stage_matches_oracle(gpu, tmp_path):
	rng = np.random.default_rng(11)
	size = (600, 800)
	page_path = synth.write_page_dir(
		tmp_path, "page0001", rng, size=size, n_h=120, n_v=160, n_regions=12, n_separators=3,
		label_size=(320, 427), warp=synth.Warp(size, 0.03, 0.025, 1, 2))
	out = page_path.with_suffix(".out")
	res = CliRunner().invoke(dewarp, [str(tmp_path), "--lock-strategy", "NONE", "--plain"])
	assert res.exit_code == 0, res.output
	assert sorted(p.name for p in out.iterdir()) == [
		"contours.0.zip", "contours.1.zip", "dewarp.zip", "flow.zip", "runtime.json", "segment.zip"]
	info = json.loads((out / "runtime.json").read_text())["origami.batch.detect.dewarp"]
	assert info["status"] == "COMPLETED"

	# dewarp.zip: format and grid values (float grid within 1e-3 px of the oracle's march)
	with zipfile.ZipFile(out / "dewarp.zip") as zf:
		assert zf.namelist() == ["data.npy", "meta.json"]
		meta = json.loads(zf.read("meta.json"))
		data = np.load(io.BytesIO(zf.read("data.npy")))
	assert meta["version"] == 1 and meta["cell"] == 25 and meta["shape"] == list(data.shape)
	assert data.dtype == np.float32
	with zipfile.ZipFile(out / "flow.zip") as zf:
		oh, ov = odw.Samples.open(zf, "h"), odw.Samples.open(zf, "v")
	ogrid = odw.Grid.create(size, oh, ov, grid_res=25)
	assert data.shape == ogrid.hv.shape
	assert np.abs(data - ogrid.hv).max() < 1e-3

	# contours.1.zip: same members as contours.0.zip, every ring warped like the oracle does
	with zipfile.ZipFile(out / "contours.0.zip") as z0, zipfile.ZipFile(out / "contours.1.zip") as z1:
		assert sorted(z1.namelist()) == sorted(z0.namelist())
		assert z1.namelist()[0] == "meta.json"
		assert z1.read("meta.json") == z0.read("meta.json")
		tf = odw.Grid(data, 25).transformer
		for name in z0.namelist():
			if not name.endswith(".wkt"):
				continue
			g0 = geometry.parse_wkt(z0.read(name).decode("utf8"))
			g1 = geometry.parse_wkt(z1.read(name).decode("utf8"))
			_assert_warped_like_oracle(g0, g1, tf)

	# a second run skips the page; --overwrite redoes it
	res = CliRunner().invoke(dewarp, [str(tmp_path), "--lock-strategy", "NONE", "--plain"])
	assert "0 ready to process" in res.output
	res = CliRunner().invoke(dewarp, [str(tmp_path), "--lock-strategy", "NONE", "--plain", "--overwrite"])
	assert "1 ready to process" in res.output and res.exit_code == 0

	# every later stage opens the page through the saved grid (ref: io.py:275-280, page.py:86-88)
	page = Page(page_path, Grid.open(out / "dewarp.zip"))
	h, w = data.shape[:2]
	assert page.dewarped.size == (w * 25, h * 25)
	want = odw.dewarped(np.array(page.warped), odw.Grid(data, 25))
	np.testing.assert_array_equal(np.array(page.dewarped), want)


def test_dewarp_stage_config0_full_size(gpu, tmp_path):
	"""BASELINE.json configs[0] at its stated workload: the `dewarp` CLI on ONE 2400 x 3400
	gray page with 300 horizontal / 500 vertical flow samples, 60 region polygons, 12 separator line strings and a
	dummy segment.zip (1280 x 2400 label maps), checked against the oracle like the scaled-down test above."""
	rng = np.random.default_rng(1000)
	size = (2400, 3400)
	page_path = synth.write_page_dir(
		tmp_path, "page0001", rng, size=size, n_h=300, n_v=500, n_regions=60, n_separators=12,
		label_size=(1280, 2400), warp=synth.Warp(size, 0.03, 0.025, 1, 2))
	out = page_path.with_suffix(".out")
	res = CliRunner().invoke(dewarp, [str(tmp_path), "--lock-strategy", "NONE", "--plain"])
	assert res.exit_code == 0, res.output
	assert sorted(p.name for p in out.iterdir()) == [
		"contours.0.zip", "contours.1.zip", "dewarp.zip", "flow.zip", "runtime.json", "segment.zip"]
	assert json.loads((out / "runtime.json").read_text())["origami.batch.detect.dewarp"]["status"] == "COMPLETED"

	with zipfile.ZipFile(out / "dewarp.zip") as zf:
		assert zf.namelist() == ["data.npy", "meta.json"]
		meta = json.loads(zf.read("meta.json"))
		data = np.load(io.BytesIO(zf.read("data.npy")))
	assert meta == dict(version=1, cell=25, shape=list(data.shape)) and data.dtype == np.float32
	with zipfile.ZipFile(out / "flow.zip") as zf:
		oh, ov = odw.Samples.open(zf, "h"), odw.Samples.open(zf, "v")
	assert len(oh) == 300 and len(ov) == 500
	ogrid = odw.Grid.create(size, oh, ov, grid_res=25)
	assert data.shape == ogrid.hv.shape and data.shape[0] >= 137 and data.shape[1] >= 97
	assert np.abs(data - ogrid.hv).max() < 1e-3

	with zipfile.ZipFile(out / "contours.0.zip") as z0, zipfile.ZipFile(out / "contours.1.zip") as z1:
		names = z0.namelist()
		assert sum(n.startswith("regions/TEXT/") for n in names) == 60
		assert sum(n.startswith("separators/V/") and n.endswith(".wkt") for n in names) == 12
		assert sorted(z1.namelist()) == sorted(names) and z1.namelist()[0] == "meta.json"
		assert z1.read("separators/V/meta.json") == z0.read("separators/V/meta.json")
		tf = odw.Grid(data, 25).transformer
		repaired = 0
		for name in names:
			if not name.endswith(".wkt"):
				continue
			g0 = geometry.parse_wkt(z0.read(name).decode("utf8"))
			g1 = geometry.parse_wkt(z1.read(name).decode("utf8"))
			if _assert_warped_like_oracle(g0, g1, tf):
				repaired += 1
				continue
			for want, got in zip(odw.transform_rings(tf, g0.rings), g1.rings):
				# "warped integer contours bit-exact": rounded coordinates agree off the .5 boundary
				off_tie = np.abs(want - np.floor(want) - 0.5) > 1e-4
				assert np.array_equal(np.round(want)[off_tie], np.round(got)[off_tie])
		assert repaired <= 6  # a handful of the 60 synthetic regions may meet themselves; most must be plain

	# the page every later stage sees (ref: io.py:275-280, page.py:86-88), whole page bit-exact
	page = Page(page_path, Grid.open(out / "dewarp.zip"))
	want = odw.dewarped(np.array(page.warped), odw.Grid(data, 25), threads=8)
	np.testing.assert_array_equal(np.array(page.dewarped), want)


def test_invalid_warped_contours_are_repaired_like_buffer0(gpu, tmp_path, caplog):
	"""ref: origami/batch/detect/dewarp.py:35-38 — a contour that is invalid after the warp is replaced by its
	buffer(0).  A bow-tie whose edges cross (one lobe survives), the shapely manual's self-touching bow-tie (both
	lobes: MULTIPOLYGON) and a degenerate separator (POLYGON EMPTY, as shapely writes for a line's buffer) go
	through the stage; the written geometry must contain exactly the points GEOS' depth rule keeps of the
	oracle-warped ring (oracle/rings.py:buffer0_depth)."""
	from oracle import rings as orings
	rng = np.random.default_rng(41)
	size = (600, 800)
	page_path = synth.write_page_dir(
		tmp_path, "page0001", rng, size=size, n_h=120, n_v=160, n_regions=4, n_separators=1,
		label_size=(320, 427), warp=synth.Warp(size, 0.03, 0.025, 1, 2))
	out = page_path.with_suffix(".out")
	cross = np.array([[100, 100], [300, 300], [300, 100], [100, 300], [100, 100]], dtype=np.float64)
	touch = np.array([[350, 400], [350, 600], [450, 500], [550, 600], [550, 400], [450, 500], [350, 400]], dtype=np.float64)
	with zipfile.ZipFile(out / "contours.0.zip", "a") as zf:
		zf.writestr("regions/TEXT/900.wkt", geometry.to_wkt(geometry.Geom("Polygon", [cross])))
		zf.writestr("regions/TEXT/901.wkt", geometry.to_wkt(geometry.Geom("Polygon", [touch])))
		zf.writestr("separators/V/900.wkt", "LINESTRING (200 200, 200 200)")
	res = CliRunner().invoke(dewarp, [str(tmp_path), "--lock-strategy", "NONE", "--plain"])
	assert res.exit_code == 0, res.output
	with zipfile.ZipFile(out / "dewarp.zip") as zf:
		data = np.load(io.BytesIO(zf.read("data.npy")))
	tf = odw.Grid(data, 25).transformer
	with zipfile.ZipFile(out / "contours.1.zip") as zf:
		got_cross = geometry.parse_wkt(zf.read("regions/TEXT/900.wkt").decode("utf8"))
		got_touch = geometry.parse_wkt(zf.read("regions/TEXT/901.wkt").decode("utf8"))
		assert zf.read("separators/V/900.wkt") == b"POLYGON EMPTY"
		valid = geometry.parse_wkt(zf.read("regions/TEXT/0.wkt").decode("utf8"))
	assert valid.geom_type == "Polygon" and len(valid.rings) == 1  # valid contours pass through untouched
	assert got_cross.geom_type == "Polygon" and len(got_cross.parts) == 1 and got_cross.is_valid
	assert got_touch.geom_type == "MultiPolygon" and len(got_touch.parts) == 2 and got_touch.is_valid
	for src, got in ((cross, got_cross), (touch, got_touch)):
		warped = odw.transform_rings(tf, [src])[0]
		lo, hi = warped.min(axis=0) - 5, warped.max(axis=0) + 5
		pts = rng.uniform(lo, hi, size=(400, 2))
		# off the outline by more than the 1e-3 px the two transformers may differ
		for p in pts:
			want = orings.buffer0_depth([warped], p) >= 1
			near = min(np.hypot(*(warped[:-1] + t * (warped[1:] - warped[:-1]) - p).T).min() for t in np.linspace(0, 1, 201))
			if near > 0.5:
				assert orings.point_in_polygons(got.parts, p) == want
		assert abs(got.area - sum(abs(0.5 * np.sum(r[:-1, 0] * r[1:, 1] - r[1:, 0] * r[:-1, 1])) for part in got.parts for r in part)) < 1e-6


def test_page_without_regions_is_skipped_silently(gpu, tmp_path):
	rng = np.random.default_rng(12)
	page_path = synth.write_page_dir(tmp_path, "empty", rng, size=(300, 400), n_h=20, n_v=20, n_regions=0, n_separators=1, label_size=(160, 213))
	res = CliRunner().invoke(dewarp, [str(tmp_path), "--lock-strategy", "NONE"])
	assert res.exit_code == 0
	out = page_path.with_suffix(".out")
	assert not (out / "dewarp.zip").exists() and not (out / "contours.1.zip").exists()  # ref: dewarp.py:64-67


def test_export_lines_stage_matches_oracle(gpu, tmp_path):
	"""`export.lines` on a RELIABLE-stage page dir: every PNG equals the oracle's Line.dewarped_image
	(ref: origami/batch/export/lines.py:51-67, batch/core/lines.py:131-149, core/block.py:223-259)."""
	import PIL.Image
	from oracle import block as oblock
	from oracle import cv as ocv
	from origami_amd.batch.export.lines import export_lines

	rng = np.random.default_rng(21)
	s, h, w = 25, 34, 26
	size = (w * s - 20, h * s - 30)
	gy, gx = np.mgrid[0:h, 0:w].astype(np.float64)
	nodes = np.stack([
		(gx - 0.5) * s + 5 * np.sin(gy / 4.0) + rng.normal(0, 0.4, size=(h, w)),
		(gy - 0.5) * s + 5 * np.cos(gx / 5.0) + rng.normal(0, 0.4, size=(h, w))], axis=-1).astype(np.float32)
	page_path = tmp_path / "page0001.png"
	pixels = synth.text_page(rng, size)
	PIL.Image.fromarray(pixels).save(page_path)
	lines = synth.write_reliable_stage(page_path, rng, nodes, s, n_lines=30, n_table_lines=3)
	out = page_path.with_suffix(".out")

	res = CliRunner().invoke(export_lines, [str(tmp_path), "--lock-strategy", "NONE", "--plain"])
	assert res.exit_code == 0, res.output
	info = json.loads((out / "runtime.json").read_text())["LineExtractionProcessor"]
	assert info["status"] == "COMPLETED", info

	ogrid = odw.Grid(nodes, s)
	tables = json.loads((out / "tables.json").read_text())["columns"]
	with zipfile.ZipFile(out / "images.lines.dewarped.zip") as zf:
		names = zf.namelist()
		# 30 text lines minus the low-confidence one, 3 table rows x 3 columns
		assert len(names) == 29 + 9
		assert "regions/TEXT/0/1.png" not in names
		bad = total = 0
		for path, (p, right, up) in lines.items():
			if path[1] == "TEXT":
				if path == ("regions", "TEXT", "0", "1"):
					continue
				got = np.array(PIL.Image.open(io.BytesIO(zf.read("/".join(path) + ".png"))))
				want = oblock.dewarped_image(pixels, ogrid, p, right, up, target_height=48)
				assert got.shape == want.shape
				d = np.abs(got.astype(np.int16) - want)
				assert d.max() <= 1
				bad += np.count_nonzero(d)
				total += d.size
			else:
				block, division, _, _ = path[2].split(".")
				xs = [None] + tables["/".join(path[:3])] + [None]
				widths = []
				for c in range(3):
					name = "regions/TABULAR/%s.%s.%d.%d/0.png" % (block, division, 1 + int(path[3]), 1 + c)
					im = PIL.Image.open(io.BytesIO(zf.read(name)))
					assert im.size[1] == 48
					widths.append(im.size[0])
				full = oblock.warped_grid(ogrid, p, right, up, ysteps=48).shape[1]
				assert abs(sum(widths) - full) <= 3 and abs(widths[1] - 0.4 * full) <= 2
		assert bad <= 1e-3 * total


def _zip_members(path):
	with zipfile.ZipFile(path) as zf:
		return {n: zf.read(n) for n in zf.namelist()}


def test_two_spawned_workers_on_one_gpu_equal_the_in_process_run(gpu, tmp_path):
	"""`--gpus 2` on a one-GPU box: both spawned workers bind GPU 0 (devices[r % 1]), create their own HIP
	context after the spawn and share the page queue; every artifact equals the in-process run byte for byte
	(ref: origami/batch/core/processor.py:470-515, one process per GPU instead of the forked Pool)."""
	import shutil
	rng = np.random.default_rng(21)
	a = tmp_path / "a"
	a.mkdir()
	for k in range(4):
		synth.write_page_dir(
			a, "page%04d" % k, rng, size=(500, 650), n_h=100, n_v=130, n_regions=10, n_separators=2,
			label_size=(250, 325))
	b = tmp_path / "b"
	shutil.copytree(a, b)
	res = CliRunner().invoke(dewarp, [str(a), "--lock-strategy", "NONE", "--plain"])
	assert res.exit_code == 0, res.output
	res = CliRunner().invoke(dewarp, [str(b), "--gpus", "2", "--workers-per-gpu", "1", "--plain"])
	assert res.exit_code == 0, res.output
	for k in range(4):
		oa, ob = a / ("page%04d.out" % k), b / ("page%04d.out" % k)
		assert json.loads((ob / "runtime.json").read_text())["origami.batch.detect.dewarp"]["status"] == "COMPLETED"
		for name in ("dewarp.zip", "contours.1.zip"):
			assert _zip_members(oa / name) == _zip_members(ob / name), (k, name)


def test_entry_points_are_thread_safe(gpu):
	"""Callers are thread pools (ref: origami/core/dewarp.py:481-483, origami/batch/core/lines.py:148): eight
	threads dewarp different pages and query the transformer at once; results equal the serial ones."""
	from multiprocessing.pool import ThreadPool
	rng = np.random.default_rng(22)
	size = (700, 900)
	jobs = []
	for k in range(8):
		nodes = synth.Warp(size, 0.02 + 0.002 * k, 0.02, 1, 2).node_grid(25)
		jobs.append((synth.text_page(rng, size), nodes, rng.uniform([0, 0], size, size=(500, 2))))
	grids = [Grid(nodes, 25) for _, nodes, _ in jobs]

	def work(i):
		page, nodes, q = jobs[i]
		x, y = grids[i].transformer(tuple(q[:, 0]), tuple(q[:, 1]))
		return gpu.dewarp_u8(page, nodes, 25), np.stack([x, y], axis=1)

	serial = [work(i) for i in range(8)]
	with ThreadPool(8) as pool:
		for _ in range(3):
			threaded = pool.map(work, range(8))
			for (p0, t0), (p1, t1) in zip(serial, threaded):
				np.testing.assert_array_equal(p0, p1)
				np.testing.assert_array_equal(t0, t1)


def test_dewarped_image_stage_matches_oracle(gpu, tmp_path):
	"""`batch.utils.dewarp` after `batch.detect.dewarp`: annotation.dewarped.jpg is the JPEG of exactly the
	page the oracle dewarps through the stored grid (ref: origami/batch/utils/dewarp.py:27-28)."""
	import PIL.Image
	from oracle import dewarp as odw2
	from origami_amd.batch.utils.dewarp import dewarp as dewarp_image_stage
	rng = np.random.default_rng(31)
	size = (640, 480)
	page_path = synth.write_page_dir(
		tmp_path, "page0001", rng, size=size, n_h=100, n_v=120, n_regions=8, n_separators=2, label_size=(320, 240))
	out = page_path.with_suffix(".out")
	assert CliRunner().invoke(dewarp, [str(tmp_path), "--lock-strategy", "NONE"]).exit_code == 0
	res = CliRunner().invoke(dewarp_image_stage, [str(tmp_path), "--lock-strategy", "NONE", "--plain"])
	assert res.exit_code == 0, res.output
	info = json.loads((out / "runtime.json").read_text())
	assert info["origami.batch.utils.dewarp"]["status"] == "COMPLETED"
	got = PIL.Image.open(out / "annotation.dewarped.jpg")
	grid = Grid.open(out / "dewarp.zip")
	nodes = grid.points("sample")
	assert got.size == (nodes.shape[1] * 25, nodes.shape[0] * 25) and got.mode == "L"
	pixels = np.array(PIL.Image.open(page_path).convert("L"))
	want = PIL.Image.fromarray(odw2.dewarped(pixels, odw2.Grid(nodes, 25)))
	ref = io.BytesIO()
	want.save(ref, format="JPEG")
	assert (out / "annotation.dewarped.jpg").read_bytes() == ref.getvalue()  # same pixels -> same JPEG bytes
	# done pages are skipped on a second run (output exists), redone with --overwrite
	res = CliRunner().invoke(dewarp_image_stage, [str(tmp_path), "--lock-strategy", "NONE", "--plain"])
	assert res.exit_code == 0


def test_dewarped_image_stage_on_a_jpeg_scan(gpu, tmp_path):
	"""The reference's usual page format: a grayscale .jpg scan.  The page is decoded on the device
	(ogd_jpeg_decode_gray, ref: origami/core/page.py:82-84), dewarped and encoded there; the result is the file
	PIL writes for the oracle's dewarp of the PIL-decoded scan."""
	import PIL.Image
	from oracle import dewarp as odw2
	from origami_amd.batch.utils.dewarp import dewarp as dewarp_image_stage
	from origami_amd.page import Page
	rng = np.random.default_rng(32)
	size = (648, 470)
	page_path = synth.write_page_dir(
		tmp_path, "page0001", rng, size=size, n_h=100, n_v=120, n_regions=8, n_separators=2, label_size=(324, 235),
		suffix=".jpg")
	assert PIL.Image.open(page_path).mode == "L"
	pixels = np.array(PIL.Image.open(page_path).convert("L"))
	assert np.array_equal(np.array(Page(page_path).warped), pixels)  # device decode == PIL decode
	out = page_path.with_suffix(".out")
	assert CliRunner().invoke(dewarp, [str(tmp_path), "--lock-strategy", "NONE"]).exit_code == 0
	res = CliRunner().invoke(dewarp_image_stage, [str(tmp_path), "--lock-strategy", "NONE", "--plain"])
	assert res.exit_code == 0, res.output
	nodes = Grid.open(out / "dewarp.zip").points("sample")
	want = PIL.Image.fromarray(odw2.dewarped(pixels, odw2.Grid(nodes, 25)))
	ref = io.BytesIO()
	want.save(ref, format="JPEG")
	assert (out / "annotation.dewarped.jpg").read_bytes() == ref.getvalue()


def test_dewarped_image_stage_on_a_colour_jpeg_scan_with_restart_markers(gpu, tmp_path, monkeypatch):
	"""A colour scan (JFIF YCbCr 4:2:0, a restart marker per MCU row): the reference opens it with
	`.convert("L")` (origami/core/page.py:82-84); here it is decoded to "L" on the device, and the stage's output is
	the file PIL writes for the oracle's dewarp of PIL's own "L" pixels."""
	import PIL.Image
	from oracle import dewarp as odw2
	from origami_amd.batch.utils.dewarp import dewarp as dewarp_image_stage
	from origami_amd.page import Page
	rng = np.random.default_rng(33)
	size = (648, 470)
	page_path = synth.write_page_dir(
		tmp_path, "page0001", rng, size=size, n_h=100, n_v=120, n_regions=8, n_separators=2, label_size=(324, 235),
		suffix=".jpg")
	PIL.Image.fromarray(synth.colour_page(rng, size)).save(page_path, format="JPEG", quality=88, restart_marker_rows=1)
	assert PIL.Image.open(page_path).mode == "RGB"
	assert gpu.jpeg_info(page_path.read_bytes())[2:] == (3, True)
	pixels = np.array(PIL.Image.open(page_path).convert("L"))
	# the device decoder must be the one that runs (not PIL compared with PIL): count its calls and results
	calls = dict(decode=0, fused=0)
	plain_decode, plain_fused = gpu.jpeg_decode_gray, gpu.jpeg_dewarp_jpeg_gray

	def decode(data):
		out = plain_decode(data)
		calls["decode"] += out is not None
		return out

	def fused(*args, **kwargs):
		out = plain_fused(*args, **kwargs)
		calls["fused"] += out is not None
		return out

	plain_from_jpeg = gpu.DevicePage.from_jpeg.__func__

	def from_jpeg(cls, data):  # the route with resident pages (origami_amd.page.resident): decoded into an ogd_page
		out = plain_from_jpeg(cls, data)
		calls["fused"] += out is not None
		return out

	monkeypatch.setattr(gpu, "jpeg_decode_gray", decode)
	monkeypatch.setattr(gpu, "jpeg_dewarp_jpeg_gray", fused)
	monkeypatch.setattr(gpu.DevicePage, "from_jpeg", classmethod(from_jpeg))
	assert np.array_equal(np.array(Page(page_path).warped), pixels)  # device decode == PIL decode + convert
	assert calls["decode"] == 1
	out = page_path.with_suffix(".out")
	assert CliRunner().invoke(dewarp, [str(tmp_path), "--lock-strategy", "NONE"]).exit_code == 0
	res = CliRunner().invoke(dewarp_image_stage, [str(tmp_path), "--lock-strategy", "NONE", "--plain"])
	assert res.exit_code == 0, res.output
	assert calls["fused"] == 1  # scan file in -> dewarped JPEG out, all on the device (one call, or resident page + one call)
	nodes = Grid.open(out / "dewarp.zip").points("sample")
	want = PIL.Image.fromarray(odw2.dewarped(pixels, odw2.Grid(nodes, 25)))
	ref = io.BytesIO()
	want.save(ref, format="JPEG")
	assert (out / "annotation.dewarped.jpg").read_bytes() == ref.getvalue()


def test_a_contour_failure_leaves_no_grid_behind(gpu, tmp_path, monkeypatch):
	"""The grid is written by a side thread while the contours are warped; if the contours fail, dewarp.zip must not
	stay behind (the reference writes it after the contours, ref: batch/detect/dewarp.py:84-90), the page is FAILED
	with the contour error — not the saver's — and a second run with the fault gone completes it.  The second page of
	the run (its grid was built ahead by the prefetch thread) is unaffected."""
	import origami_amd.batch.detect.dewarp as stage
	rng = np.random.default_rng(12)
	for k in range(2):
		synth.write_page_dir(
			tmp_path, "page%04d" % k, rng, size=(500, 650), n_h=100, n_v=130, n_regions=8, n_separators=2,
			label_size=(250, 325))
	real = stage.dewarped_contours
	calls = []

	def failing(warped, transformer, min_areas):
		calls.append(1)
		if len(calls) == 1:
			raise ValueError("synthetic contour failure")
		return real(warped, transformer, min_areas)

	monkeypatch.setattr(stage, "dewarped_contours", failing)
	res = CliRunner().invoke(dewarp, [str(tmp_path), "--lock-strategy", "NONE", "--plain"])
	assert res.exit_code == 0, res.output
	out0, out1 = tmp_path / "page0000.out", tmp_path / "page0001.out"
	assert not (out0 / "dewarp.zip").exists() and not (out0 / "contours.1.zip").exists()
	info = json.loads((out0 / "runtime.json").read_text())["origami.batch.detect.dewarp"]
	assert info["status"] == "FAILED" and "synthetic contour failure" in info["traceback"]
	assert (out1 / "dewarp.zip").exists() and (out1 / "contours.1.zip").exists()
	res = CliRunner().invoke(dewarp, [str(tmp_path), "--lock-strategy", "NONE", "--plain"])
	assert res.exit_code == 0, res.output
	assert (out0 / "dewarp.zip").exists() and (out0 / "contours.1.zip").exists()
	assert json.loads((out0 / "runtime.json").read_text())["origami.batch.detect.dewarp"]["status"] == "COMPLETED"
