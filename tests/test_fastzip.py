"""origami_amd/batch/core/fastzip.py against the zipfile module, in both directions (CPU)."""
import io
import zipfile
import zlib

import pytest

from origami_amd.batch.core import fastzip

MEMBERS = [
	("meta.json", b'{"version": 2}'),
	("regions/TEXT/0.wkt", b"POLYGON ((0 0, 10 0, 10 10, 0 0))" * 40),
	("regions/TEXT/empty.wkt", b""),
	("separators/V/über.wkt", bytes(range(256)) * 30),
]


@pytest.mark.parametrize("method", [zipfile.ZIP_DEFLATED, zipfile.ZIP_STORED])
def test_what_zipwriter_writes_zipfile_reads(method):
	f = io.BytesIO()
	with fastzip.ZipWriter(f, method) as zf:
		for name, data in MEMBERS:
			zf.writestr(name, data)
		zf.writestr("text.txt", "as str ✓")
	with zipfile.ZipFile(io.BytesIO(f.getvalue())) as zf:
		assert zf.testzip() is None
		assert zf.namelist() == [n for n, _ in MEMBERS] + ["text.txt"]
		for name, data in MEMBERS:
			assert zf.read(name) == data
			assert zf.getinfo(name).compress_type == method
		assert zf.read("text.txt").decode("utf8") == "as str ✓"


@pytest.mark.parametrize("method", [zipfile.ZIP_DEFLATED, zipfile.ZIP_STORED])
def test_what_zipfile_writes_zipreader_reads(method):
	f = io.BytesIO()
	with zipfile.ZipFile(f, "w", method) as zf:
		for name, data in MEMBERS:
			zf.writestr(name, data)
	for source in (f.getvalue(), io.BytesIO(f.getvalue()), b"junk in front" + f.getvalue()):
		zr = fastzip.ZipReader(source)
		assert zr.namelist() == [n for n, _ in MEMBERS]
		assert dict(zr.items()) == dict(MEMBERS)
		assert "meta.json" in zr and "nope" not in zr
		with pytest.raises(KeyError):
			zr.read("nope")


def test_reader_hands_unusual_archives_to_zipfile_and_rejects_damage():
	f = io.BytesIO()
	with zipfile.ZipFile(f, "w", zipfile.ZIP_BZIP2) as zf:  # a method fastzip does not speak
		zf.writestr("a.txt", b"hello " * 100)
	assert fastzip.ZipReader(f.getvalue()).read("a.txt") == b"hello " * 100
	with pytest.raises(zipfile.BadZipFile):
		fastzip.ZipReader(b"not a zip file at all")
	whole = io.BytesIO()
	with fastzip.ZipWriter(whole) as zf:
		zf.writestr("a.txt", b"hello " * 100)
		zf.writestr("b.txt", b"world " * 100)
	eocd = whole.getvalue()[-22:]
	with pytest.raises(zipfile.BadZipFile):  # the end record alone: its directory lies before the start of the data
		fastzip.ZipReader(eocd)
	g = io.BytesIO()
	with fastzip.ZipWriter(g) as zf:
		zf.writestr("a.txt", b"hello " * 100)
	damaged = bytearray(g.getvalue())
	damaged[40] ^= 0xFF  # inside the deflate stream
	with pytest.raises((zipfile.BadZipFile, Exception)):
		fastzip.ZipReader(bytes(damaged)).read("a.txt")


def test_items_inflates_all_members_in_one_call_and_names_damage():
	"""ZipReader.items() goes through ogd_inflate_many: deflated and stored members, empty ones, archives written by
	zipfile at several levels (string matches, unlike ZipWriter's literal-only streams); a damaged member raises
	BadZipFile (or zlib's error from the one-by-one path that names it) and no member of a damaged archive is handed out."""
	import numpy as np
	rng = np.random.default_rng(9)
	members = {"meta.json": b"{}", "empty.bin": b"", "noise.bin": rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(),
		"text/a.wkt": b"POLYGON ((0 0, 10 0, 10 10, 0 10, 0 0))" * 50, "zeros": bytes(200000)}
	for level in (0, 1, 6, 9):
		f = io.BytesIO()
		with zipfile.ZipFile(f, "w", zipfile.ZIP_DEFLATED if level else zipfile.ZIP_STORED, compresslevel=level or None) as zf:
			for name, data in members.items():
				zf.writestr(name, data)
		assert dict(fastzip.ZipReader(f.getvalue()).items()) == members
	g = io.BytesIO()
	with fastzip.ZipWriter(g) as zf:
		for name, data in members.items():
			zf.writestr(name, data)
	assert dict(fastzip.ZipReader(g.getvalue()).items()) == members
	whole = g.getvalue()
	at = whole.index(b"noise.bin") + len(b"noise.bin") + 1000  # inside the member's data
	for damaged in (whole[:at] + bytes([whole[at] ^ 0x55]) + whole[at + 1:],):
		with pytest.raises(Exception) as info:
			dict(fastzip.ZipReader(damaged).items())
		assert isinstance(info.value, (zipfile.BadZipFile, zlib.error)), info.value


def test_writer_leaves_nothing_behind_when_the_block_raises():
	f = io.BytesIO()
	with pytest.raises(RuntimeError):
		with fastzip.ZipWriter(f) as zf:
			zf.writestr("a", b"1")
			raise RuntimeError("stage failed")
	assert f.getvalue() == b""


def test_members_of_one_archive_may_be_stored_or_deflated():
	"""writestr(..., compress_type=ZIP_STORED) for a member that is compressed already (the PNG files of the line
	export, ref: origami/batch/export/lines.py:62-67): zipfile and fastzip both read the mixed archive."""
	import io
	import os
	blob = os.urandom(5000)
	f = io.BytesIO()
	with fastzip.ZipWriter(f, zipfile.ZIP_DEFLATED) as zf:
		zf.writestr("a/text.wkt", "POLYGON ((0 0, 1 0, 1 1, 0 0))" * 40)
		zf.writestr("a/image.png", blob, compress_type=zipfile.ZIP_STORED)
		zf.writestr("b.json", b"{}", compress_type=zipfile.ZIP_DEFLATED)
		zf.writestr("a/other.png", blob[:100], compress_type=zipfile.ZIP_STORED, crc=zlib.crc32(blob[:100]))  # CRC given by the caller
	with zipfile.ZipFile(io.BytesIO(f.getvalue())) as zf:
		assert zf.testzip() is None
		infos = {i.filename: i for i in zf.infolist()}
		assert infos["a/image.png"].compress_type == zipfile.ZIP_STORED and infos["a/image.png"].compress_size == 5000
		assert infos["a/text.wkt"].compress_type == zipfile.ZIP_DEFLATED and infos["a/text.wkt"].compress_size < 1200 * 0.6  # entropy coding only: no string matching
		assert zf.read("a/image.png") == blob and zf.read("a/other.png") == blob[:100]
	zr = fastzip.ZipReader(io.BytesIO(f.getvalue()))
	assert dict(zr.items())["a/image.png"] == blob and zr.read("b.json") == b"{}"
