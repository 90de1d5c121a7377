"""libogdewarp's PNG page decoder and its inflate (host code, CPU tests) against zlib and PIL."""
import io
import shutil
import struct
import subprocess
import zlib
from pathlib import Path

import numpy as np
import PIL.Image
import pytest

from origami_amd import _shim, synth
from origami_amd.page import Page

ROOT = Path(__file__).resolve().parent.parent


def _deflate(raw, level, strategy):
	c = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strategy)
	return c.compress(raw) + c.flush()


def test_inflate_equals_zlib():
	rng = np.random.default_rng(1)
	samples = [
		b"", b"a", bytes(1000), rng.integers(0, 256, 100000, dtype=np.uint8).tobytes(), b"hello world, " * 5000,
		rng.integers(0, 4, 300000, dtype=np.uint8).tobytes(), rng.integers(228, 240, 500000, dtype=np.uint8).tobytes(),
		bytes(range(256)) * 300]
	for raw in samples:
		for level in (0, 1, 6, 9):
			for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
				z = _deflate(raw, level, strategy)
				assert _shim.inflate_zlib(z, len(raw)).tobytes() == raw
				assert _shim.inflate_zlib(z, len(raw) + 1000).tobytes() == raw


def test_inflate_rejects_damage():
	raw = b"the quick brown fox jumps over the lazy dog " * 2000
	z = _deflate(raw, 6, zlib.Z_DEFAULT_STRATEGY)
	for bad, size in [
			(z[:-1], len(raw)), (z[:len(z) // 2], len(raw)), (z[:5], len(raw)), (b"", 10), (z, len(raw) - 1),
			(b"\x78\x9d" + z[2:], len(raw)), (z[:-4] + b"\0\0\0\0", len(raw)), (z[:100] + bytes(50) + z[150:], len(raw))]:
		with pytest.raises(_shim.OgdError):
			_shim.inflate_zlib(bad, size)


def _chunk(kind, data):
	return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data))


def _png_with_filters(pixels, filters, idat_size=7000, level=6):
	"""A PNG file of `pixels` ((H, W) or (H, W, 3) uint8) whose row y uses filter type filters[y % len(filters)]."""
	h, w = pixels.shape[:2]
	bpp = 1 if pixels.ndim == 2 else 3
	rows = pixels.reshape(h, w * bpp).astype(np.int32)
	raw = bytearray()
	prev = np.zeros(w * bpp, dtype=np.int32)
	for y in range(h):
		cur = rows[y]
		left = np.concatenate([np.zeros(bpp, dtype=np.int32), cur[:-bpp]]) if w * bpp > bpp else np.zeros(w * bpp, dtype=np.int32)
		upleft = np.concatenate([np.zeros(bpp, dtype=np.int32), prev[:-bpp]]) if w * bpp > bpp else np.zeros(w * bpp, dtype=np.int32)
		t = filters[y % len(filters)]
		if t == 0:
			f = cur
		elif t == 1:
			f = cur - left
		elif t == 2:
			f = cur - prev
		elif t == 3:
			f = cur - (left + prev) // 2
		else:
			p = left + prev - upleft
			pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
			f = cur - np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
		raw.append(t)
		raw += (f & 255).astype(np.uint8).tobytes()
		prev = cur
	z = zlib.compress(bytes(raw), level)
	out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0 if bpp == 1 else 2, 0, 0, 0))
	out += _chunk(b"tEXt", b"Comment\0made by the test")
	for k in range(0, len(z), idat_size):
		out += _chunk(b"IDAT", z[k:k + idat_size])
	return out + _chunk(b"IEND", b"")


@pytest.mark.parametrize("shape", [(1, 1), (3, 2), (5, 7), (64, 33), (200, 311)])
@pytest.mark.parametrize("colour", [False, True])
def test_png_every_filter_type_equals_pil(shape, colour):
	rng = np.random.default_rng(shape[0] * 7 + colour)
	h, w = shape
	pixels = rng.integers(0, 256, size=(h, w, 3) if colour else (h, w), dtype=np.uint8)
	pixels[h // 2:] = (pixels[h // 2:] // 32) * 32  # flat runs as well as noise
	for filters in ([0], [1], [2], [3], [4], [4, 3, 2, 1, 0], [1, 4], [2, 3, 4]):
		data = _png_with_filters(pixels, filters)
		want = np.array(PIL.Image.open(io.BytesIO(data)))
		np.testing.assert_array_equal(want, pixels)  # the test's own encoder is right
		assert _shim.png_info(data) == (w, h, 3 if colour else 1)
		np.testing.assert_array_equal(_shim.png_decode(data), pixels)


def test_png_written_by_pil_and_page_open(tmp_path):
	rng = np.random.default_rng(3)
	gray = synth.text_page(rng, (1200, 1700))
	rgb = synth.colour_page(rng, (500, 700))
	for pixels in (gray, rgb):
		for level in (1, 6, 9):
			b = io.BytesIO()
			PIL.Image.fromarray(pixels).save(b, format="PNG", compress_level=level)
			np.testing.assert_array_equal(_shim.png_decode(b.getvalue()), pixels)
	# Page.warped takes this route for a PNG scan and gives PIL's "L" pixels
	for name, pixels in (("gray.png", gray), ("rgb.png", rgb)):
		PIL.Image.fromarray(pixels).save(tmp_path / name)
		want = np.array(PIL.Image.open(tmp_path / name).convert("L"))
		seen = []
		real = _shim.png_decode
		try:
			_shim.png_decode = lambda data: seen.append(1) or real(data)
			got = np.array(Page(tmp_path / name).warped)
		finally:
			_shim.png_decode = real
		assert seen == [1]
		np.testing.assert_array_equal(got, want)


def test_png_kinds_left_to_pil_and_damaged_files(tmp_path):
	rng = np.random.default_rng(4)
	gray = rng.integers(0, 256, size=(40, 50), dtype=np.uint8)

	def saved(im, **kw):
		b = io.BytesIO()
		im.save(b, format="PNG", **kw)
		return b.getvalue()
	others = [
		saved(PIL.Image.fromarray(gray).convert("P")), saved(PIL.Image.fromarray(gray).convert("RGBA")),
		saved(PIL.Image.fromarray(gray).convert("LA")), saved(PIL.Image.fromarray(gray.astype(np.uint16) * 200)),
		saved(PIL.Image.fromarray(gray > 128))]
	for data in others:
		with pytest.raises(_shim.OgdError) as e:
			_shim.png_info(data)
		assert e.value.status == _shim.ERR_UNSUPPORTED
	good = saved(PIL.Image.fromarray(gray))
	interlaced = bytearray(good)
	interlaced[28] = 1  # IHDR interlace method (the CRC is not looked at)
	with pytest.raises(_shim.OgdError) as e:
		_shim.png_decode(bytes(interlaced))
	assert e.value.status == _shim.ERR_UNSUPPORTED
	for bad in (good[:60], good[:-20], good[:50] + bytes(30) + good[80:], b"not a png", good.replace(b"IEND", b"IENX")):
		with pytest.raises(_shim.OgdError):
			_shim.png_decode(bad)
	# a kind the decoder leaves alone still opens through PIL
	PIL.Image.fromarray(gray).convert("P").save(tmp_path / "palette.png")
	np.testing.assert_array_equal(
		np.array(Page(tmp_path / "palette.png").warped), np.array(PIL.Image.open(tmp_path / "palette.png").convert("L")))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_inflate_fuzz_under_address_sanitizer(tmp_path):
	"""tools/inflate_fuzz.cpp: damaged and truncated streams, exact-size heap buffers, ASan + UBSan on the host."""
	exe = tmp_path / "inflate_fuzz"
	build = subprocess.run(
		["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", str(ROOT / "tools" / "inflate_fuzz.cpp"),
			"-lz", "-o", str(exe)], capture_output=True, text=True)
	if build.returncode != 0:
		pytest.skip("sanitizer build not available here: %s" % build.stderr[-300:])
	run = subprocess.run([str(exe), "60"], capture_output=True, text=True, timeout=300)
	if run.returncode != 0 and "runtime does not come first" in run.stderr:
		pytest.skip("another preloaded library keeps AddressSanitizer from starting here")
	assert run.returncode == 0, run.stdout + run.stderr
	assert "valid streams decoded" in run.stdout
