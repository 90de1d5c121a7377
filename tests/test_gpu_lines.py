"""GPU parity of the per-line image path against oracle/block.py, through the C ABI.

Bar: the inverse map is fp64 on both sides and cast to float32 before cv2.remap's fixed-point
arithmetic, so the images are bit-exact except where a 1e-12 difference in fp64 flips the float32
cast across a 1/32-px quantisation boundary: required >= 99.9 % of pixels exact and all within +-1."""

import numpy as np
import pytest

from oracle import block as oblock
from oracle import cv as ocv
from oracle import dewarp as odw
from origami_amd import synth
from origami_amd.batch.core.io import Stage

pytestmark = pytest.mark.gpu


def _nodes(rng, h, w, s, amp=6.0):
	gy, gx = np.mgrid[0:h, 0:w].astype(np.float64)
	x = (gx - 1) * s + amp * np.sin(gy / 3.0) + rng.normal(0, 0.5, size=(h, w))
	y = (gy - 1) * s + amp * np.cos(gx / 4.0) + rng.normal(0, 0.5, size=(h, w))
	return np.stack([x, y], axis=-1).astype(np.float32)


def _lines(rng, n, Wd, Hd):
	"""p0 on the baseline, right along it (slightly skewed), up pointing to smaller y."""
	out = np.empty((n, 6))
	for k in range(n):
		length = rng.uniform(30, Wd * 0.8)
		height = rng.uniform(12, 60)
		ang = rng.normal(0, 0.02)
		x = rng.uniform(-10, Wd - length + 10)  # some lines poke out of the page
		y = rng.uniform(height, Hd + 5)
		out[k] = (x, y, length * np.cos(ang), length * np.sin(ang), height * np.sin(ang), -height * np.cos(ang))
	return out


def _check(got, want):
	assert got.shape == want.shape
	d = np.abs(got.astype(np.int16) - want.astype(np.int16))
	assert d.max(initial=0) <= 1
	return int(np.count_nonzero(d)), d.size


@pytest.mark.parametrize("C", [1, 3])
@pytest.mark.parametrize("interp", [ocv.INTER_LINEAR, ocv.INTER_NEAREST])
def test_line_images_match_oracle(gpu, C, interp):
	rng = np.random.default_rng(100 + C + interp)
	h, w, s = 24, 18, 25
	nodes = _nodes(rng, h, w, s)
	grid = odw.Grid(nodes, s)
	page = synth.text_page(rng, (w * s - 30, h * s - 40), channels=C)
	lines = _lines(rng, 40, w * s, h * s)
	xsteps = rng.integers(2, 300, size=len(lines)).astype(np.int32)
	ysteps = np.where(rng.random(len(lines)) < 0.7, 48, rng.integers(1, 64, size=len(lines))).astype(np.int32)
	xsteps[0] = 1
	with gpu.DevicePage(page) as dp:
		got = dp.line_images(nodes, s, lines, xsteps, ysteps, interp)
	bad = total = 0
	for k, v in enumerate(lines):
		wg = oblock.warped_grid(grid, v[0:2], v[2:4], v[4:6], xsteps=int(xsteps[k]), ysteps=int(ysteps[k]))
		want = ocv.remap_u8(page, wg, interp)[::-1, :]
		b, t = _check(got[k], want)
		bad += b
		total += t
	assert bad <= 1e-3 * total, (bad, total)


def test_line_images_device_output_equals_host_output(gpu):
	"""ogd_line_images_u8_dev leaves the images in device memory (for a GPU-resident consumer): same bytes as
	ogd_line_images_u8, read back through ogd_memcpy_d2h."""
	import ctypes
	rng = np.random.default_rng(321)
	h, w, s = 24, 18, 25
	nodes = _nodes(rng, h, w, s)
	page = synth.text_page(rng, (w * s - 30, h * s - 40), channels=3)
	lines = _lines(rng, 25, w * s, h * s)
	xsteps = rng.integers(2, 300, size=len(lines)).astype(np.int32)
	ysteps = np.full(len(lines), 48, dtype=np.int32)
	lib = gpu.load()
	with gpu.DevicePage(page) as dp:
		want = dp.line_images(nodes, s, lines, xsteps, ysteps)
		total = int(np.sum(xsteps.astype(np.int64) * ysteps) * 3)
		d_out = ctypes.c_void_p()
		gpu.check(lib.ogd_dev_alloc(ctypes.byref(d_out), total))
		try:
			offsets = dp.line_images_dev(nodes, s, lines, xsteps, ysteps, d_out.value)
			assert offsets[-1] == total
			flat = np.empty(total, dtype=np.uint8)
			gpu.check(lib.ogd_memcpy_d2h(flat.ctypes.data, d_out, total, None))
			gpu.check(lib.ogd_stream_sync(None))
		finally:
			gpu.check(lib.ogd_dev_free(d_out))
	for k, im in enumerate(want):
		got = flat[offsets[k]:offsets[k + 1]].reshape(im.shape)
		np.testing.assert_array_equal(got, im)


def test_line_mirror_matches_oracle(gpu):
	"""Line.warped_grid / dewarped_image / dewarped_line_images (sizes chosen by the rough lattice)."""
	from origami_amd import block as blk
	from origami_amd import dewarp as gdw
	import PIL.Image

	rng = np.random.default_rng(5)
	h, w, s = 30, 22, 25
	nodes = _nodes(rng, h, w, s)
	ogrid = odw.Grid(nodes, s)
	pixels = synth.text_page(rng, (w * s - 30, h * s - 40))

	class FakePage:
		warped = PIL.Image.fromarray(pixels)
		dewarper = gdw.Dewarper(warped, gdw.Grid(nodes, s))
		device_page = gpu.DevicePage(pixels)

	block = blk.Block(FakePage, None, Stage.DEWARPED)
	geo = _lines(rng, 25, w * s, h * s)
	lines = [blk.Line(block, v[0:2], v[2:4], v[4:6], confidence={"regions/TEXT": 0.9}) for v in geo]

	wg = lines[3].warped_grid(ysteps=48)
	want = oblock.warped_grid(ogrid, geo[3, 0:2], geo[3, 2:4], geo[3, 4:6], ysteps=48)
	assert wg.shape == want.shape and wg.dtype == np.float32
	np.testing.assert_allclose(wg, want, atol=1e-3)

	images = blk.dewarped_line_images(FakePage, [(ln, None) for ln in lines], target_height=48)
	bad = total = 0
	for im, v in zip(images, geo):
		want = oblock.dewarped_image(pixels, ogrid, v[0:2], v[2:4], v[4:6], target_height=48)
		b, t = _check(np.array(im), want)
		bad += b
		total += t
	assert bad <= 1e-3 * total, (bad, total)

	one = np.array(lines[7].dewarped_image(48))
	np.testing.assert_array_equal(one, np.array(images[7]))

	# a table column clips the baseline (block.py:200-219) and scales the horizontal resolution
	ln = lines[2]
	x_lo, x_hi = sorted((ln._p[0], ln._p[0] + ln._right[0]))
	col = (x_lo + 0.25 * (x_hi - x_lo), x_lo + 0.75 * (x_hi - x_lo))
	part = np.array(ln.dewarped_image(48, column=col))
	full = np.array(images[2])
	assert part.shape[0] == 48 and abs(part.shape[1] - 0.5 * full.shape[1]) <= 2
	p0, right, up, _ = ln._position(1, col)
	wgc = oblock.warped_grid(ogrid, p0, right, up, xsteps=part.shape[1], ysteps=48)
	b, t = _check(part, ocv.remap_u8(pixels, wgc, ocv.INTER_LINEAR)[::-1, :])
	assert b <= 1e-3 * t
	FakePage.device_page.close()


def test_resident_page_dewarp_equals_host_entry(gpu):
	rng = np.random.default_rng(11)
	h, w, s = 16, 12, 25
	nodes = _nodes(rng, h, w, s)
	page = synth.text_page(rng, (w * s - 10, h * s - 10))
	with gpu.DevicePage(page) as dp:
		a = dp.dewarp(nodes, s)
		b = dp.dewarp(nodes, s, interpolation=ocv.INTER_NEAREST)
	np.testing.assert_array_equal(a, gpu.dewarp_u8(page, nodes, s))
	np.testing.assert_array_equal(b, gpu.dewarp_u8(page, nodes, s, interpolation=ocv.INTER_NEAREST))


def test_line_images_reject_bad_arguments(gpu):
	rng = np.random.default_rng(1)
	nodes = _nodes(rng, 4, 4, 8)
	page = synth.text_page(rng, (30, 30))
	with gpu.DevicePage(page) as dp:
		assert dp.line_images(nodes, 8, np.zeros((0, 6)), [], []) == []
		with pytest.raises(gpu.OgdError, match="steps"):
			dp.line_images(nodes, 8, np.zeros((1, 6)), [0], [48])
	with pytest.raises(ValueError, match="closed"):
		dp.line_images(nodes, 8, np.zeros((1, 6)), [4], [4])


def test_line_label_counts_match_oracle(gpu):
	"""ConfidenceSampler (ref: batch/detect/lines.py:25-54): histogram of the nearest-remapped label map
	under each line's half-resolution lattice.  Bar: label work is bit-exact except where the float32 cast
	of the fp64 inverse map flips a round-to-nearest tie: <= 0.1 % of samples may land in another bin."""
	from origami_amd import block as blk
	from origami_amd import dewarp as gdw
	from origami_amd.batch.detect.lines import ConfidenceSampler
	from origami_amd.segment import Prediction, Segmentation
	from origami_amd.batch.core.io import PredictorType
	import enum
	import PIL.Image

	rng = np.random.default_rng(9)
	h, w, s = 30, 22, 25
	nodes = _nodes(rng, h, w, s)
	ogrid = odw.Grid(nodes, s)
	W, H = w * s - 30, h * s - 40
	pixels = synth.text_page(rng, (W, H))
	labels = synth.label_map(rng, (W // 3, H // 3 + 7), n_classes=4, blocks=9)
	assert labels.shape == (H // 3 + 7, W // 3) and labels.dtype == np.uint8
	geo = _lines(rng, 30, w * s, h * s)

	# C ABI level, explicit steps
	xsteps = rng.integers(2, 200, size=len(geo)).astype(np.int32)
	ysteps = rng.integers(2, 30, size=len(geo)).astype(np.int32)
	sx, sy = np.float32(labels.shape[1] / W), np.float32(labels.shape[0] / H)
	with gpu.DevicePage(labels) as dl:
		got = dl.line_label_counts(nodes, s, geo, xsteps, ysteps, sx, sy)
	assert got.shape == (len(geo), 256) and got.dtype == np.int32
	moved = total = 0
	for k, v in enumerate(geo):
		wg = oblock.warped_grid(ogrid, v[0:2], v[2:4], v[4:6], xsteps=int(xsteps[k]), ysteps=int(ysteps[k]))
		wg[:, :, 0] *= labels.shape[1] / W
		wg[:, :, 1] *= labels.shape[0] / H
		want = np.bincount(ocv.remap_u8(labels, wg, ocv.INTER_NEAREST).flatten(), minlength=256)
		assert got[k].sum() == want.sum() == int(xsteps[k]) * int(ysteps[k])
		moved += np.abs(got[k] - want).sum() // 2
		total += want.sum()
	assert moved <= 1e-3 * total, (moved, total)

	# the reference's interface
	class FakePage:
		warped = PIL.Image.fromarray(pixels)
		dewarper = gdw.Dewarper(warped, gdw.Grid(nodes, s))

	classes = enum.Enum("RegionLabel", dict(BACKGROUND=0, TEXT=1, TABULAR=2, ILLUSTRATION=3))
	seg = Segmentation([Prediction(PredictorType.REGION, "regions", labels, classes)])
	block = blk.Block(FakePage, None, Stage.AGGREGATE)
	lines = [blk.Line(block, v[0:2], v[2:4], v[4:6]) for v in geo]
	sampler = ConfidenceSampler({("regions", "TEXT", "0"): block}, seg)
	evidence = sampler.sample_many([(("regions", "TEXT", "0"), ln) for ln in lines])
	for ev, v in zip(evidence, geo):
		want = oblock.label_evidence_counts(labels, (H, W), ogrid, v[0:2], v[2:4], v[4:6], minlength=4)
		assert sorted(ev) == ["regions/BACKGROUND", "regions/ILLUSTRATION", "regions/TABULAR", "regions/TEXT"]
		for c in classes:
			assert ev["regions/%s" % c.name] == pytest.approx(want[c.value] / want.sum(), abs=2e-3)
		assert sum(ev.values()) == pytest.approx(1.0)
	one = sampler(("regions", "TEXT", "0"), lines[4])
	assert one == evidence[4]
	lines[4].update_confidence(one)
	assert lines[4].confidence == max(v for k, v in one.items() if not k.endswith("/BACKGROUND"))
	sampler.close()
