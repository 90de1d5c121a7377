"""N4: the JPEG encoder (ogd_jpeg_encode_gray, ogd_dewarp_jpeg_gray) against PIL / libjpeg-turbo, the library
the reference's `image.save(path)` runs (ref: origami/batch/core/io.py:606-610).  Bar: the same file, byte
for byte."""

import io

import numpy as np
import PIL.Image
import pytest

from oracle import jpeg as ojpeg
from origami_amd import synth

pytestmark = pytest.mark.gpu


def _first_diff(a, b):
	n = min(len(a), len(b))
	d = np.nonzero(np.frombuffer(a[:n], np.uint8) != np.frombuffer(b[:n], np.uint8))[0]
	return "sizes %d / %d, first difference at byte %s" % (len(a), len(b), d[0] if len(d) else "-")


def _check(gpu, pixels, **params):
	want = ojpeg.save_gray(pixels, **params)
	got = gpu.jpeg_encode_gray(pixels, **params)
	assert got == want, _first_diff(got, want)


@pytest.mark.parametrize("shape", [(1, 1), (7, 9), (8, 8), (17, 33), (64, 64), (250, 301), (1000, 777)])
def test_noise_images_of_ragged_sizes(gpu, shape):
	rng = np.random.default_rng(shape[0] * 1000 + shape[1])
	_check(gpu, rng.integers(0, 256, size=shape, dtype=np.uint8))


@pytest.mark.parametrize("quality", [1, 10, 50, 75, 90, 95, 100])
def test_qualities(gpu, quality):
	rng = np.random.default_rng(quality)
	page = synth.text_page(rng, (640, 480))
	_check(gpu, page, quality=quality)
	noise = rng.integers(0, 256, size=(96, 200), dtype=np.uint8)
	_check(gpu, noise, quality=quality)


def test_flat_gradient_and_extreme_images(gpu):
	_check(gpu, np.zeros((40, 40), np.uint8))
	_check(gpu, np.full((40, 40), 255, np.uint8))
	_check(gpu, np.tile(np.arange(256, dtype=np.uint8), (100, 3)))
	# checkerboards of full swing: the largest coefficients, long codes, many 0xFF bytes in the stream
	yy, xx = np.mgrid[0:200, 0:312]
	for period in (1, 2, 3, 8):
		_check(gpu, (((yy // period + xx // period) & 1) * 255).astype(np.uint8), quality=100)
		_check(gpu, (((yy // period + xx // period) & 1) * 255).astype(np.uint8))


def test_row_stride_and_non_contiguous_input(gpu):
	rng = np.random.default_rng(5)
	big = rng.integers(0, 256, size=(300, 500), dtype=np.uint8)
	view = big[10:210, 33:400]  # row stride 500, width 367
	assert gpu.jpeg_encode_gray(view) == ojpeg.save_gray(view)
	assert gpu.jpeg_encode_gray(big[::2, ::3]) == ojpeg.save_gray(big[::2, ::3])


def test_file_decodes_and_reports_size(gpu):
	rng = np.random.default_rng(6)
	page = synth.text_page(rng, (800, 600))
	data = gpu.jpeg_encode_gray(page)
	im = PIL.Image.open(io.BytesIO(data))
	assert im.mode == "L" and im.size == (800, 600)
	assert np.abs(np.asarray(im).astype(int) - page).mean() < 6


def test_rejects_bad_arguments(gpu):
	with pytest.raises(TypeError):
		gpu.jpeg_encode_gray(np.zeros((4, 4, 3), np.uint8))
	with pytest.raises(gpu.OgdError):
		gpu.jpeg_encode_gray(np.zeros((4, 4), np.uint8), quality=0)
	with pytest.raises(gpu.OgdError):
		gpu.jpeg_encode_gray(np.zeros((4, 4), np.uint8), quality=101)


def test_whole_broadsheet_page(gpu):
	"""configs[1] size: the dewarped page of a 6000 x 8000 scan (6075 x 8075, neither a multiple of 8)."""
	rng = np.random.default_rng(7)
	page = synth.text_page(rng, (6075, 8075))
	_check(gpu, page)


def test_dewarp_then_encode_on_the_device(gpu):
	rng = np.random.default_rng(8)
	size = (1200, 1600)
	page = synth.text_page(rng, size)
	nodes = synth.Warp.random(rng, size).node_grid(25)
	dewarped = gpu.dewarp_u8(page, nodes, 25)
	assert gpu.dewarp_jpeg_gray(page, nodes, 25) == ojpeg.save_gray(dewarped)
	assert gpu.dewarp_jpeg_gray(page, nodes, 25, quality=90) == ojpeg.save_gray(dewarped, quality=90)
