"""origami_amd/batch/core/fastzip.py against the zipfile module, in both directions (CPU)."""
import io
import zipfile

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


def test_writer_leaves_nothing_behind_when_the_block_raises():
	f = io.BytesIO()
	with pytest.raises(RuntimeError):
		with fastzip.ZipWriter(f) as zf:
			zf.writestr("a", b"1")
			raise RuntimeError("stage failed")
	assert f.getvalue() == b""
