"""libogdewarp's deflate writer (csrc/ogd_deflate.h, host code) read back by zlib and by the library's own inflate."""
import io
import shutil
import subprocess
import zipfile
import zlib
from pathlib import Path

import numpy as np
import pytest

from origami_amd import _shim, synth
from origami_amd.dewarp import Grid

ROOT = Path(__file__).resolve().parent.parent


def _cases():
	rng = np.random.default_rng(2)
	fib = [1, 1]
	while len(fib) < 40:
		fib.append(fib[-1] + fib[-2])
	skewed = b"".join(bytes([k]) * min(f, 200000) for k, f in enumerate(fib[:27]))  # a Huffman tree deeper than 15 bits
	return {
		"empty": b"", "one": b"x", "two": b"xy", "zeros": bytes(100000), "all bytes": bytes(range(256)) * 300,
		"random": rng.integers(0, 256, 200000, dtype=np.uint8).tobytes(), "text": b"12.5 7001.25, " * 4000,
		"skewed": skewed, "65534": bytes(rng.integers(48, 58, 65534, dtype=np.uint8)),
		"65535": bytes(rng.integers(48, 58, 65535, dtype=np.uint8)), "65536": bytes(rng.integers(48, 58, 65536, dtype=np.uint8)),
		"131071": bytes(rng.integers(0, 3, 131071, dtype=np.uint8)),
	}


@pytest.mark.parametrize("name", list(_cases()))
def test_deflate_is_read_back_by_zlib_and_by_our_inflate(name):
	data = _cases()[name]
	z = _shim.deflate_raw(data)
	assert zlib.decompress(z, -15) == data
	framed = b"\x78\x01" + z + zlib.adler32(data).to_bytes(4, "big")
	assert _shim.inflate_zlib(framed, len(data)).tobytes() == data
	assert len(z) <= len(data) + len(data) // 1000 + 64


def test_deflate_size_against_zlib_on_the_stage_outputs():
	rng = np.random.default_rng(3)
	member = io.BytesIO()
	np.save(member, synth.Warp((6000, 8000), 0.03, 0.025, 1, 2).node_grid(25).astype(np.float32), allow_pickle=False)
	npy = member.getvalue()
	ring = synth.polygon_rings(rng, (6000, 8000), 1, mu=20)[0]
	wkt = ("POLYGON ((" + ", ".join("%.16g %.16g" % (x, y) for x, y in ring) + "))").encode()
	for data, slack in ((npy, 1.05), (wkt, 1.10)):
		ours = len(_shim.deflate_raw(data))
		c = zlib.compressobj(1, zlib.DEFLATED, -15)
		theirs = len(c.compress(data) + c.flush())
		assert ours <= slack * theirs, (ours, theirs)
		assert ours < 0.95 * len(data)


def test_deflate_many_and_the_archives_built_from_it(tmp_path):
	members = [b"", b"POLYGON ((0 0, 1 0, 1 1, 0 0))", bytes(70000), b"abc" * 1000]
	packed = _shim.deflate_many(members)
	assert [zlib.decompress(z, -15) for z, _ in packed] == members
	assert [c for _, c in packed] == [zlib.crc32(m) for m in members]
	assert _shim.deflate_many([]) == []
	# Grid.save -> an ordinary deflate zip that zipfile and Grid.open read
	nodes = synth.Warp((1200, 1600), 0.03, 0.025, 1, 2).node_grid(25)
	Grid(nodes, 25).save(tmp_path / "dewarp.zip")
	with zipfile.ZipFile(tmp_path / "dewarp.zip") as zf:
		assert zf.testzip() is None
		assert zf.namelist() == ["data.npy", "meta.json"]
		assert all(i.compress_type == zipfile.ZIP_DEFLATED for i in zf.infolist())
	np.testing.assert_array_equal(Grid.open(tmp_path / "dewarp.zip").points("sample"), nodes.astype(np.float32))
	with open(tmp_path / "again.zip", "wb") as f:
		Grid(nodes, 25).save(f)
	assert (tmp_path / "again.zip").read_bytes()[:4] == b"PK\x03\x04"


def test_png_writer_gives_the_pixels_back_through_pil_and_through_our_own_decoder():
	"""ogd_png_encode_many (the line images of the export stage, ref: origami/batch/export/lines.py:62-67): every file
	is a valid PNG holding exactly the pixels, for grey and RGB, rows of every filter type, and the degenerate sizes."""
	import io
	import PIL.Image
	from origami_amd import _shim, synth
	rng = np.random.default_rng(3)
	page = synth.text_page(rng, (1500, 400))
	grad = (np.add.outer(np.arange(40), np.arange(300)) % 256).astype(np.uint8)  # Paeth / Average rows
	grey = [page[k * 48:(k + 1) * 48, :int(rng.integers(100, 1500))] for k in range(8)] + [
		grad, grad.T.copy(), np.zeros((3, 300), np.uint8), np.full((1, 1), 7, np.uint8),
		rng.integers(0, 256, size=(1, 50), dtype=np.uint8), rng.integers(0, 256, size=(50, 1), dtype=np.uint8),
		rng.integers(0, 256, size=(64, 64), dtype=np.uint8)]
	files = _shim.png_encode_many(grey)
	assert len(files) == len(grey)
	filters = set()
	for im, data in zip(grey, files):
		with PIL.Image.open(io.BytesIO(data)) as f:
			f.verify()  # chunk CRCs
		got = PIL.Image.open(io.BytesIO(data))
		assert got.mode == "L" and np.array_equal(np.array(got), im)
		assert np.array_equal(_shim.png_decode(data), im)
		assert len(data) <= _shim.load().ogd_png_bound(im.shape[0], im.shape[1], 1)
		raw = zlib.decompress(data[41:-16])  # one IDAT between IHDR (33 bytes in) and IEND: Adler-32 verified by zlib
		rows = set(raw[::im.shape[1] + 1])
		assert len(rows - {0, 1}) <= 1 or im.shape[0] == 1  # one filter per image (its first row may say None / Sub for Up / Paeth)
		filters |= rows
	assert len(filters) >= 2  # and the heuristic (one filter per image, from probe rows) does choose between them
	rgb = [rng.integers(0, 256, size=(9, 31, 3), dtype=np.uint8), np.dstack([page[:40, :500], page[40:80, :500], page[80:120, :500]]),
		rng.integers(0, 256, size=(1, 1, 3), dtype=np.uint8)]
	for im, data in zip(rgb, _shim.png_encode_many(rgb)):
		got = PIL.Image.open(io.BytesIO(data))
		assert got.mode == "RGB" and np.array_equal(np.array(got), im)
	assert _shim.png_encode_many([]) == []
	with pytest.raises(TypeError):
		_shim.png_encode_many([grey[0], rgb[0]])
	# text lines: within a few per cent of PIL's optimize=True
	ours = sum(len(d) for d in files[:8])
	ref = 0
	for im in grey[:8]:
		b = io.BytesIO()
		PIL.Image.fromarray(im).save(b, format="png", optimize=True)
		ref += len(b.getvalue())
	assert ours < 1.10 * ref


def test_checksums_equal_zlibs_over_lengths_and_alignments():
	"""csrc/ogd_checksum.h: CRC-32 by carry-less multiplication and Adler-32 by SSSE3 multiply-add where the CPU has
	them (else the portable forms) — through the entry points that use them: the member CRCs of ogd_deflate_many and
	the Adler-32 check of ogd_inflate_zlib, on lengths around every block boundary of the vector code."""
	import ctypes
	rng = np.random.default_rng(8)
	blob = rng.integers(0, 256, size=300000, dtype=np.uint8).tobytes()
	lengths = list(range(0, 200)) + [255, 256, 257, 1023, 4095, 4096, 4097, 5535, 5536, 5537, 5552, 11072, 11073, 65535, 65536, 65537, 200001]
	members = []
	for k, n in enumerate(lengths):
		off = (7 * k) % 61
		members.append(blob[off:off + n])
	members.append(b"\xff" * 100000)  # the sums at their largest
	for data, (stream, crc) in zip(members, _shim.deflate_many(members)):
		assert crc == zlib.crc32(data), len(data)
		assert zlib.decompress(stream, -15) == data
	lib = _shim.load()
	for data in members[::3] + [members[-1]]:
		z = zlib.compress(data, 1)
		out = np.empty(max(len(data), 1), dtype=np.uint8)
		n_out = ctypes.c_int64(0)
		assert lib.ogd_inflate_zlib(z, len(z), out.ctypes.data, out.size, ctypes.byref(n_out)) == 0, len(data)
		assert out[:n_out.value].tobytes() == data
		bad = bytearray(z)
		bad[-1] ^= 1  # a wrong Adler-32 is noticed
		assert lib.ogd_inflate_zlib(bytes(bad), len(bad), out.ctypes.data, out.size, ctypes.byref(n_out)) != 0


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_deflate_fuzz_under_address_sanitizer(tmp_path):
	"""tools/deflate_fuzz.cpp: the entropy coder into exact-size heap buffers of every capacity around the stream's
	size (zlib reads every stream back, short buffers are refused), the SIMD checksums against zlib's; ASan + UBSan."""
	exe = tmp_path / "deflate_fuzz"
	build = subprocess.run(
		["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
			str(ROOT / "tools" / "deflate_fuzz.cpp"), "-lz", "-o", str(exe)], capture_output=True, text=True)
	if build.returncode != 0:
		pytest.skip("sanitizer build not available here: %s" % build.stderr[-300:])
	run = subprocess.run([str(exe), "60"], capture_output=True, text=True, timeout=300)
	if run.returncode != 0 and "runtime does not come first" in run.stderr:
		pytest.skip("another preloaded library keeps AddressSanitizer from starting here")
	assert run.returncode == 0, run.stdout + run.stderr
	assert "streams read back by zlib" in run.stdout
