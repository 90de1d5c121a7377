"""N4: the device-side baseline JPEG decoder (ogd_jpeg_decode_gray) against PIL / libjpeg-turbo, which is what
the reference's `PIL.Image.open(path).convert("L")` runs (ref: origami/core/page.py:82-84).  Bar: the same
pixels, bit for bit.  The files are made here with PIL's encoder (and with the device encoder)."""

import io

import numpy as np
import PIL.Image
import pytest

from oracle import jpeg as ojpeg
from origami_amd import synth

pytestmark = pytest.mark.gpu


def _pil_decode(data):
	return np.array(PIL.Image.open(io.BytesIO(data)).convert("L"))


def _check(gpu, pixels, **params):
	data = ojpeg.save_gray(pixels, **params)
	got = gpu.jpeg_decode_gray(data)
	assert got is not None, "decoder refused a baseline grayscale file"
	want = _pil_decode(data)
	assert got.shape == want.shape
	assert np.array_equal(got, want), "%d pixels differ, max %d" % (
		(got != want).sum(), np.abs(got.astype(int) - want).max())


@pytest.mark.parametrize("shape", [(1, 1), (7, 9), (8, 8), (17, 33), (64, 64), (250, 301), (1000, 777)])
def test_noise_images_of_ragged_sizes(gpu, shape):
	rng = np.random.default_rng(shape[0] * 1000 + shape[1])
	_check(gpu, rng.integers(0, 256, size=shape, dtype=np.uint8))


@pytest.mark.parametrize("quality", [1, 10, 50, 75, 90, 95, 100])
def test_qualities(gpu, quality):
	rng = np.random.default_rng(quality)
	_check(gpu, synth.text_page(rng, (640, 480)), quality=quality)
	_check(gpu, rng.integers(0, 256, size=(96, 200), dtype=np.uint8), quality=quality)


def test_optimised_huffman_tables(gpu):
	"""optimize=True: tables fitted to the image (codes of up to 16 bits, not the standard ones)."""
	rng = np.random.default_rng(3)
	_check(gpu, synth.text_page(rng, (800, 600)), optimize=True)
	_check(gpu, rng.integers(0, 256, size=(200, 200), dtype=np.uint8), optimize=True, quality=95)


def test_flat_and_extreme_images(gpu):
	_check(gpu, np.zeros((40, 40), np.uint8))
	_check(gpu, np.full((2000, 40), 255, np.uint8))  # long runs of identical blocks: chunks without a block start
	yy, xx = np.mgrid[0:200, 0:312]
	for period in (1, 3, 8):
		_check(gpu, (((yy // period + xx // period) & 1) * 255).astype(np.uint8), quality=100)


def _with_margins(page, fill):
	"""The middle of `page` on blank paper: an eighth of the height and width on every side is `fill`."""
	out = np.empty_like(page)
	out[...] = fill
	h, w = page.shape[:2]
	out[h // 8:h - h // 8, w // 8:w - w // 8] = page[h // 8:h - h // 8, w // 8:w - w // 8]
	return out


@pytest.mark.parametrize("params", [{}, {"optimize": True}, {"quality": 30}], ids=["standard", "fitted-tables", "q30"])
def test_blank_margins_and_blank_pages(gpu, params):
	"""Blank paper: every block the same few bits.  The state pass crosses such stretches by arithmetic on the period
	of the bit string (period_kernel, the lattice steps of decode_chunks_kernel) — same pixels as libjpeg, and the
	whole-page cases must not take the hand-back route (a 6 MP blank page is 1100 chunks of one repeated block)."""
	rng = np.random.default_rng(11)
	page = synth.text_page(rng, (1500, 2000))
	for fill in (255, 128, 0):
		_check(gpu, _with_margins(page, fill), **params)
	_check(gpu, np.full((3000, 2000), 255, np.uint8), **params)
	_check(gpu, np.full((3000, 2000), 77, np.uint8), **params)
	half = np.full((2000, 1496), 255, np.uint8)  # text above, blank paper below: one long periodic tail
	half[:700] = page[:700, :1496]
	_check(gpu, half, **params)


@pytest.mark.parametrize("subsampling", [0, 1, 2])
def test_colour_blank_margins_and_blank_pages(gpu, subsampling):
	rng = np.random.default_rng(12 + subsampling)
	page = synth.colour_page(rng, (1500, 2000))
	for fill in ((255, 255, 255), (200, 100, 50)):
		_check_colour(gpu, _with_margins(page, np.array(fill, np.uint8)), subsampling=subsampling, quality=85)
		_check_colour(gpu, np.broadcast_to(np.array(fill, np.uint8), (2000, 1504, 3)).copy(), subsampling=subsampling)
	_check_colour(gpu, _with_margins(page, np.array((255, 255, 255), np.uint8)), subsampling=subsampling, optimize=True)


@pytest.mark.parametrize("restart", [dict(restart_marker_rows=1), dict(restart_marker_rows=7), dict(restart_marker_blocks=100),
	dict(restart_marker_blocks=64)])
def test_blank_paper_with_restart_intervals(gpu, restart):
	"""The lattice steps stop in front of every interval start (an MCU that ends less than 8 bits before one is
	followed by the interval, not by the next period): intervals of many chunks, of about a chunk, and such that end
	on byte borders (64 blank blocks are 48 bytes: no padding between the intervals)."""
	rng = np.random.default_rng(14)
	page = synth.text_page(rng, (1500, 2000))
	_check_file(gpu, _restart_file(_with_margins(page, 255), **restart))
	_check_file(gpu, _restart_file(np.full((2000, 3000), 255, np.uint8), **restart))
	_check_file(gpu, _restart_file(np.full((1000, 1504), 128, np.uint8), **restart))
	cpage = synth.colour_page(rng, (1104, 1504))
	_check_file(gpu, _restart_file(_with_margins(cpage, np.array((255, 255, 255), np.uint8)), quality=85, **restart))
	_check_file(gpu, _restart_file(np.broadcast_to(np.array((200, 100, 50), np.uint8), (2000, 1504, 3)).copy(), **restart))


def test_random_periodic_images(gpu):
	"""Thirty seeded images that are periodic in parts: a tile of 8 .. 32 pixels repeated over stretches of the page,
	flat bands, noise in between; random quality, Huffman tables, colour sampling and restart intervals.  Periodic
	pixels make periodic bit strings of every period the lattice steps do and do not cover; wherever the decoder
	answers, its pixels are libjpeg's (a file it hands back within its time allowance is none of this test's business)."""
	rng = np.random.default_rng(77)
	for case in range(30):
		h, w = int(rng.integers(200, 900)), int(rng.integers(200, 1200))
		colour = rng.random() < 0.4
		img = rng.integers(0, 256, size=(h, w, 3) if colour else (h, w), dtype=np.uint8)
		for _ in range(int(rng.integers(1, 4))):  # periodic or flat bands over the noise
			y0, y1 = sorted(int(v) for v in rng.integers(0, h, 2))
			t = int(rng.choice([8, 16, 24, 32]))
			tile = rng.integers(0, 256, size=(t, t) + img.shape[2:], dtype=np.uint8)
			if rng.random() < 0.4:
				tile[...] = tile[0, 0]
			reps = (-(-(y1 - y0) // t), -(-w // t)) + (1,) * (img.ndim - 2)
			img[y0:y1] = np.tile(tile, reps)[:y1 - y0, :w]
		params = dict(quality=int(rng.choice([30, 60, 75, 90, 100])), optimize=bool(rng.random() < 0.3))
		if colour:
			params["subsampling"] = int(rng.integers(0, 3))
		if rng.random() < 0.3:
			params["restart_marker_rows"] = int(rng.integers(1, 5))
			params["optimize"] = False  # PIL's writer cannot do both on large noise images ("broken data stream")
		data = _colour_file(img, **params)
		got = gpu.jpeg_decode_gray(data)
		if got is not None:
			assert np.array_equal(got, _pil_decode(data)), "case %d: %r %r" % (case, img.shape, params)


def test_end_of_scan_taken_on_trust_and_checked_on_the_device(gpu):
	"""The decode calls take the end of the scan from a trailing EOI marker and let the first device pass find out
	whether another marker stands inside; then (and for files that do not end in EOI) the host walks the scan as
	ogd_jpeg_info does.  Same pixels or the same refusal either way."""
	rng = np.random.default_rng(31)
	page = synth.text_page(rng, (900, 700))
	data = ojpeg.save_gray(page)
	want = _pil_decode(data)
	assert data[-2:] == b"\xff\xd9"
	assert np.array_equal(gpu.jpeg_decode_gray(data), want)
	assert np.array_equal(gpu.jpeg_decode_gray(data + b"\x00\x00\x00"), want)  # bytes behind EOI: the host walks
	assert np.array_equal(gpu.jpeg_decode_gray(data + b"\xff\xd9"), want)  # a second EOI: the first ends the scan
	body = bytearray(data)
	at = len(body) // 2
	while body[at] == 0xFF or body[at - 1] == 0xFF:
		at += 1
	body[at:at] = b"\xff\xd3"  # an RST marker in a file without restart intervals
	assert gpu.jpeg_info(bytes(body))[3] is False
	assert gpu.jpeg_decode_gray(bytes(body)) is None
	body = bytearray(data)
	body[-3] = 0xFF  # the scan's last byte an 0xFF: FF FF D9 is a fill byte and EOI for libjpeg; refused here
	assert gpu.jpeg_decode_gray(bytes(body)) is None


def test_second_parse_does_not_strand_a_device_block(gpu):
	"""A file that ends in EOI but holds another marker inside the trusted scan area is parsed twice; the second
	pass must not take a second H x W block while the first stays marked busy for ever (48 MB per such broadsheet
	file in a long-running worker).  The calling thread's block cache has no busy block between calls."""
	import ctypes
	lib = gpu.load()
	rng = np.random.default_rng(32)
	page = synth.text_page(rng, (900, 700))
	data = ojpeg.save_gray(page) + b"\xff\xd9"
	want = _pil_decode(data)
	busy = ctypes.c_int(-1)
	for _ in range(3):
		assert np.array_equal(gpu.jpeg_decode_gray(data), want)
		assert lib.ogd_pool_busy(ctypes.byref(busy)) == 0 and busy.value == 0
		with gpu.DevicePage.from_jpeg(data) as dp:
			assert dp is not None
		assert lib.ogd_pool_busy(ctypes.byref(busy)) == 0 and busy.value == 0


def test_large_periodic_stream_is_handed_back_in_bounded_time(gpu):
	"""A 1-pixel checkerboard at quality 100: every block the same ~700 bits, wrong parses are self-consistent and
	single lanes would walk the whole 5 MB stream (seconds).  The decoder gives up after its time allowance and the
	caller keeps PIL."""
	import time
	yy, xx = np.mgrid[0:3000, 0:2000]
	data = ojpeg.save_gray((((yy + xx) & 1) * 255).astype(np.uint8), quality=100)
	t0 = time.perf_counter()
	got = gpu.jpeg_decode_gray(data)
	dt = time.perf_counter() - t0
	assert dt < 1.5, "%.2f s" % dt
	assert got is None or np.array_equal(got, _pil_decode(data))


def test_files_outside_the_domain_are_refused_not_misdecoded(gpu):
	rng = np.random.default_rng(4)
	rgb = rng.integers(0, 256, size=(64, 64, 3), dtype=np.uint8)
	buf = io.BytesIO()
	PIL.Image.fromarray(rgb).convert("CMYK").save(buf, format="JPEG")
	assert gpu.jpeg_info(buf.getvalue())[3] is False  # Adobe marker, four components
	assert gpu.jpeg_decode_gray(buf.getvalue()) is None
	buf = io.BytesIO()
	PIL.Image.fromarray(rgb).save(buf, format="JPEG", progressive=True)
	assert gpu.jpeg_info(buf.getvalue())[2:] == (3, False)
	assert gpu.jpeg_decode_gray(buf.getvalue()) is None
	buf = io.BytesIO()
	PIL.Image.fromarray(rgb[:, :3]).save(buf, format="JPEG", subsampling=2)  # libjpeg upsamples such strips plainly
	assert gpu.jpeg_info(buf.getvalue())[2:] == (3, False)
	buf = io.BytesIO()
	PIL.Image.fromarray(rgb[:, :, 0]).save(buf, format="JPEG", progressive=True)
	assert gpu.jpeg_info(buf.getvalue())[3] is False
	assert gpu.jpeg_decode_gray(buf.getvalue()) is None
	with pytest.raises(gpu.OgdError):
		gpu.jpeg_info(b"not a jpeg file at all")


def _colour_file(rgb, **params):
	buf = io.BytesIO()
	PIL.Image.fromarray(rgb).save(buf, format="JPEG", **params)
	return buf.getvalue()


def _check_colour(gpu, rgb, **params):
	data = _colour_file(rgb, **params)
	assert gpu.jpeg_info(data)[2:] == (3, True)
	got = gpu.jpeg_decode_gray(data)
	assert got is not None, "decoder refused a baseline YCbCr file"
	want = _pil_decode(data)
	assert got.shape == want.shape
	assert np.array_equal(got, want), "%d pixels differ, max %d" % (
		(got != want).sum(), np.abs(got.astype(int) - want).max())


_colour_page = synth.colour_page


@pytest.mark.parametrize("subsampling", [0, 1, 2])
@pytest.mark.parametrize("shape", [(8, 8), (16, 16), (17, 33), (33, 17), (64, 64), (250, 301), (301, 250), (999, 778)])
def test_colour_noise_images_of_ragged_sizes(gpu, shape, subsampling):
	"""Three components, 4:4:4 / 4:2:2 / 4:2:0: libjpeg's fancy upsampling, YCbCr -> RGB, PIL's RGB -> L."""
	rng = np.random.default_rng(shape[0] * 1000 + shape[1] + subsampling)
	_check_colour(gpu, rng.integers(0, 256, size=shape + (3,), dtype=np.uint8), subsampling=subsampling, quality=90)


@pytest.mark.parametrize("quality", [5, 50, 75, 95, 100])
def test_colour_qualities_and_fitted_tables(gpu, quality):
	rng = np.random.default_rng(100 + quality)
	page = _colour_page(rng, (640, 480))
	_check_colour(gpu, page, quality=quality)  # PIL's default: 4:2:0 below quality 90... whatever it picks
	_check_colour(gpu, page, quality=quality, optimize=True, subsampling=1)
	_check_colour(gpu, page, quality=quality, subsampling="4:4:4")


def test_colour_flat_and_saturated_images(gpu):
	for colour in ((0, 0, 0), (255, 255, 255), (255, 0, 0), (0, 255, 0), (0, 0, 255), (12, 200, 99)):
		_check_colour(gpu, np.broadcast_to(np.array(colour, np.uint8), (120, 200, 3)).copy(), quality=100, subsampling=2)
	yy, xx = np.mgrid[0:200, 0:312]
	chk = (((yy // 3 + xx // 3) & 1) * 255).astype(np.uint8)
	_check_colour(gpu, np.dstack([chk, 255 - chk, chk]), quality=100, subsampling=2)
	_check_colour(gpu, np.dstack([chk, 255 - chk, chk]), quality=100, subsampling=0)


def test_whole_colour_broadsheet_scan(gpu):
	"""configs[1] size in colour, 4:2:0 (what scanners and PIL write by default)."""
	rng = np.random.default_rng(8)
	_check_colour(gpu, _colour_page(rng, (6000, 8000)), quality=85)


def test_colour_scan_file_to_dewarped_file_on_the_device(gpu):
	rng = np.random.default_rng(13)
	size = (1100, 1500)
	scan = _colour_file(_colour_page(rng, size), quality=88)
	nodes = synth.Warp.random(rng, size).node_grid(25)
	want = ojpeg.save_gray(gpu.dewarp_u8(_pil_decode(scan), nodes, 25))
	assert gpu.jpeg_dewarp_jpeg_gray(scan, nodes, 25) == want


def _restart_file(pixels, **params):
	buf = io.BytesIO()
	PIL.Image.fromarray(pixels).save(buf, format="JPEG", **params)
	data = buf.getvalue()
	assert b"\xff\xdd" in data  # a DRI segment
	return data


def _check_file(gpu, data):
	got = gpu.jpeg_decode_gray(data)
	assert got is not None, "decoder refused the file: %s" % (gpu.jpeg_info(data),)
	want = _pil_decode(data)
	assert got.shape == want.shape
	assert np.array_equal(got, want), "%d pixels differ, max %d" % (
		(got != want).sum(), np.abs(got.astype(int) - want).max())


@pytest.mark.parametrize("shape", [(8, 8), (17, 33), (250, 301), (640, 999)])
@pytest.mark.parametrize("restart", [dict(restart_marker_rows=1), dict(restart_marker_blocks=1),
	dict(restart_marker_blocks=3), dict(restart_marker_blocks=17), dict(restart_marker_rows=4)])
def test_scans_with_restart_intervals(gpu, shape, restart):
	"""DRI: every interval is byte-aligned and starts its DC predictions anew; more than eight markers (RST0-7 cycle)."""
	rng = np.random.default_rng(shape[0] + shape[1])
	_check_file(gpu, _restart_file(rng.integers(0, 256, size=shape, dtype=np.uint8), quality=90, **restart))
	rgb = rng.integers(0, 256, size=shape + (3,), dtype=np.uint8)
	for sub in (0, 1, 2):
		_check_file(gpu, _restart_file(rgb, quality=80, subsampling=sub, **restart))


def test_restart_intervals_text_pages_and_fitted_tables(gpu):
	rng = np.random.default_rng(21)
	_check_file(gpu, _restart_file(synth.text_page(rng, (1500, 2000)), restart_marker_rows=1))
	_check_file(gpu, _restart_file(synth.text_page(rng, (1500, 2000)), restart_marker_rows=2, optimize=True, quality=95))
	_check_file(gpu, _restart_file(synth.colour_page(rng, (1100, 1500)), restart_marker_rows=1, quality=85))
	_check_file(gpu, _restart_file(np.full((300, 200), 255, np.uint8), restart_marker_blocks=5))  # all-ones bytes: stuffing


def test_restart_intervals_long_colour_streams(gpu):
	"""More than 65536 chunks with three components (a case that once decoded wrongly)."""
	rng = np.random.default_rng(24)
	noise = rng.integers(0, 256, size=(1600, 2000, 3), dtype=np.uint8)
	_check_file(gpu, _restart_file(noise, restart_marker_rows=1, subsampling=0, quality=98))
	_check_file(gpu, _restart_file(noise, restart_marker_blocks=7, subsampling=2, quality=100))
	_check_file(gpu, _restart_file(synth.colour_page(rng, (6000, 8000)), restart_marker_rows=1, quality=85))


def test_restart_intervals_whole_broadsheet_scan(gpu):
	rng = np.random.default_rng(22)
	_check_file(gpu, _restart_file(synth.text_page(rng, (6000, 8000)), restart_marker_rows=1))


def test_restart_intervals_damaged_or_long(gpu):
	rng = np.random.default_rng(23)
	page = synth.text_page(rng, (512, 384))
	data = bytearray(_restart_file(page, restart_marker_rows=1))
	# a marker removed: the count no longer matches -> handed back
	pos = data.index(b"\xff\xd3")
	assert gpu.jpeg_decode_gray(bytes(data[:pos] + data[pos + 2:])) is None
	# markers out of order -> handed back
	swapped = bytearray(data)
	swapped[pos + 1] = 0xD5
	assert gpu.jpeg_decode_gray(bytes(swapped)) is None
	# an interval cut short (bytes removed in front of a marker) -> handed back, no fault
	assert gpu.jpeg_decode_gray(bytes(data[:pos - 40] + data[pos:])) is None
	# random damage inside intervals: an image or None, never a fault
	start = data.index(b"\xff\xda") + 10
	for trial in range(8):
		bad = bytearray(data)
		for q in rng.integers(start, len(bad) - 2, size=4 + trial * 8):
			if bad[q] != 0xFF and bad[q - 1] != 0xFF:
				bad[q] = int(rng.integers(0, 255))
		out = gpu.jpeg_decode_gray(bytes(bad))
		assert out is None or out.shape == (384, 512)
	# a DRI segment but one interval for the whole scan, and two long ones
	_check_file(gpu, _restart_file(synth.text_page(rng, (2400, 3400)), restart_marker_rows=500))
	_check_file(gpu, _restart_file(synth.text_page(rng, (2400, 3400)), restart_marker_rows=213))
	_check_file(gpu, bytes(data))


def test_truncated_scan_is_left_to_pil(gpu):
	rng = np.random.default_rng(9)
	data = ojpeg.save_gray(synth.text_page(rng, (400, 300)))
	cut = data[:len(data) // 2] + b"\xff\xd9"  # half the scan, then EOI
	assert gpu.jpeg_decode_gray(cut) is None


def test_decodes_what_the_device_encoder_wrote(gpu):
	rng = np.random.default_rng(5)
	page = synth.text_page(rng, (900, 700))
	data = gpu.jpeg_encode_gray(page, quality=85)
	assert np.array_equal(gpu.jpeg_decode_gray(data), _pil_decode(data))


def test_whole_broadsheet_scan(gpu):
	"""configs[1] size: a 6000 x 8000 scan."""
	rng = np.random.default_rng(7)
	_check(gpu, synth.text_page(rng, (6000, 8000)))


def test_scan_file_to_dewarped_file_on_the_device(gpu):
	"""ogd_jpeg_dewarp_jpeg_gray = decode, dewarp, encode without leaving the device: the file PIL writes for the
	oracle-checked dewarp of the PIL-decoded scan."""
	rng = np.random.default_rng(10)
	size = (1100, 1500)
	scan = ojpeg.save_gray(synth.text_page(rng, size), quality=88)
	nodes = synth.Warp.random(rng, size).node_grid(25)
	want = ojpeg.save_gray(gpu.dewarp_u8(_pil_decode(scan), nodes, 25))
	assert gpu.jpeg_dewarp_jpeg_gray(scan, nodes, 25) == want
	buf = io.BytesIO()
	PIL.Image.fromarray(rng.integers(0, 256, size=(64, 64), dtype=np.uint8)).save(buf, format="JPEG", progressive=True)
	assert gpu.jpeg_dewarp_jpeg_gray(buf.getvalue(), nodes, 25) is None


def test_resident_page_from_a_scan_file(gpu):
	"""ogd_page_create_jpeg: the page decoded into device memory dewarps like the PIL-decoded one."""
	rng = np.random.default_rng(11)
	size = (900, 1200)
	scan = ojpeg.save_gray(synth.text_page(rng, size), quality=92)
	nodes = synth.Warp.random(rng, size).node_grid(25)
	with gpu.DevicePage.from_jpeg(scan) as page:
		assert page.shape == (1200, 900, 1)
		got = page.dewarp(nodes, 25)
	assert np.array_equal(got, gpu.dewarp_u8(_pil_decode(scan), nodes, 25))
	assert gpu.DevicePage.from_jpeg(ojpeg.save_gray(np.zeros((16, 16), np.uint8), progressive=True)) is None


def test_damaged_scans_never_fault_or_hang(gpu):
	"""Bytes of the entropy-coded segment overwritten at random: the decoder either produces an image (what it
	shows is undefined, as with any decoder) or hands the file back (None) — it must not fault, hang or raise
	anything but OgdError for a structurally broken file."""
	rng = np.random.default_rng(12)
	data = bytearray(ojpeg.save_gray(synth.text_page(rng, (512, 384)), quality=80))
	start = data.index(b"\xff\xda") + 10
	for trial in range(12):
		bad = bytearray(data)
		for pos in rng.integers(start, len(bad) - 2, size=1 + trial * 8):
			bad[pos] = int(rng.integers(0, 255))  # never 0xFF: the marker structure stays intact
		try:
			out = gpu.jpeg_decode_gray(bytes(bad))
		except gpu.OgdError:
			continue
		assert out is None or out.shape == (384, 512)
	# a healthy file still decodes afterwards (nothing was left in a bad state)
	assert np.array_equal(gpu.jpeg_decode_gray(bytes(data)), _pil_decode(bytes(data)))


@pytest.mark.parametrize("params", [dict(subsampling=2), dict(subsampling=0), dict(subsampling=2, restart_marker_rows=1),
	dict(subsampling=1, restart_marker_blocks=5)])
def test_damaged_colour_scans_never_fault_or_hang(gpu, params):
	"""As test_damaged_scans_never_fault_or_hang, for three-component scans with and without restart intervals."""
	rng = np.random.default_rng(25)
	data = bytearray(_colour_file(synth.colour_page(rng, (512, 384)), quality=80, **params))
	start = data.index(b"\xff\xda") + 14
	for trial in range(10):
		bad = bytearray(data)
		for pos in rng.integers(start, len(bad) - 2, size=1 + trial * 8):
			if bad[pos] != 0xFF and bad[pos - 1] != 0xFF:  # markers and stuffing stay intact
				bad[pos] = int(rng.integers(0, 255))
		try:
			out = gpu.jpeg_decode_gray(bytes(bad))
		except gpu.OgdError:
			continue
		assert out is None or out.shape == (384, 512)
	_check_file(gpu, bytes(data))
