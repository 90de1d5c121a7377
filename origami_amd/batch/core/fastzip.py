"""Zip archives of many small members, read and written whole.

A contours zip holds one WKT text of a kilobyte or two per contour (ref: origami/batch/core/io.py:142-144 reads
them, origami/batch/detect/dewarp.py:98-100 writes them, both through `zipfile`).  The zipfile module spends
≈ 60 µs of interpreter time per member on either side — 25 ms per broadsheet page and direction, more than the
GPU works on the page.  The archives are a few hundred kilobytes, so they are taken and produced in one piece here:
one read and a walk over the central directory with `struct`, or one buffer assembled and written at close.

The bytes on disk stay an ordinary zip (deflate, method 8, or stored): `zipfile`, `unzip` and the reference read
what ZipWriter writes, and ZipReader reads what they write.  Anything outside the plain format (zip64, encryption,
other methods, multi-disk) goes to the zipfile module, so no archive is refused that the reference would accept.
"""

import io
import struct
import time
import zipfile
import zlib

_EOCD = struct.Struct("<4s4H2LH")       # end of central directory
_CENTRAL = struct.Struct("<4s6H3L5H2L")  # central directory file header
_LOCAL = struct.Struct("<4s5H3L2H")      # local file header
_SUPPORTED_FLAGS = 0x0800 | 0x0006       # UTF-8 names, deflate level hints


class _Fallback(Exception):
	pass


def _scan(buf):
	"""[(name, method, crc, compressed size, size, data offset)] in central directory order."""
	tail = buf[-(1 << 16) - _EOCD.size:]
	at = tail.rfind(b"PK\x05\x06")
	if at < 0:
		raise zipfile.BadZipFile("File is not a zip file")
	_, disk, cd_disk, n_here, n_total, cd_size, cd_offset, _ = _EOCD.unpack_from(tail, at)
	if disk or cd_disk or n_here != n_total or n_total == 0xFFFF or cd_offset == 0xFFFFFFFF or cd_size == 0xFFFFFFFF:
		raise _Fallback()
	shift = (len(buf) - len(tail) + at) - (cd_offset + cd_size)  # bytes in front of the archive
	if shift < 0:
		raise zipfile.BadZipFile("Bad offset for central directory")
	pos = cd_offset + shift
	members = []
	for _ in range(n_total):
		(sig, _, _, flags, method, _, _, crc, csize, size, n_name, n_extra, n_comment, _, _, _,
			header_at) = _CENTRAL.unpack_from(buf, pos)
		if sig != b"PK\x01\x02":
			raise zipfile.BadZipFile("Bad magic number for central directory")
		if flags & ~_SUPPORTED_FLAGS or method not in (0, 8) or 0xFFFFFFFF in (csize, size, header_at):
			raise _Fallback()
		raw = bytes(buf[pos + _CENTRAL.size:pos + _CENTRAL.size + n_name])
		name = raw.decode("utf-8" if flags & 0x0800 else "cp437")
		header_at += shift
		if buf[header_at:header_at + 4] != b"PK\x03\x04":
			raise zipfile.BadZipFile("Bad magic number for file header")
		l_name, l_extra = struct.unpack_from("<2H", buf, header_at + 26)
		members.append((name, method, crc, csize, size, header_at + _LOCAL.size + l_name + l_extra))
		pos += _CENTRAL.size + n_name + n_extra + n_comment
	return members


class ZipReader:
	"""namelist() / read(name) / items() over an archive held in memory."""

	def __init__(self, source):
		"""source: bytes, or a binary file object (read to its end)."""
		self._buf = source if isinstance(source, (bytes, bytearray, memoryview)) else source.read()
		self._zf = None
		try:
			self._members = {m[0]: m for m in _scan(self._buf)}
		except struct.error:  # a directory or header that runs past the end of the file
			raise zipfile.BadZipFile("Truncated zip file")
		except _Fallback:
			self._zf = zipfile.ZipFile(io.BytesIO(self._buf), "r")
			self._members = dict.fromkeys(self._zf.namelist())

	def namelist(self):
		return list(self._members)

	def __contains__(self, name):
		return name in self._members

	def read(self, name):
		if self._zf is not None:
			return self._zf.read(name)
		try:
			_, method, crc, csize, size, at = self._members[name]
		except KeyError:
			raise KeyError("There is no item named %r in the archive" % name)
		data = self._buf[at:at + csize]
		if method == 8:
			data = zlib.decompress(data, -15, size or 1)
		else:
			data = bytes(data)
		if len(data) != size or zlib.crc32(data) != crc:
			raise zipfile.BadZipFile("Bad CRC-32 for file %r" % name)
		return data

	def _small_members(self):
		"""True if the archive's deflated members are a few hundred bytes each (a page's contours: 400 members of
		160 bytes).  The library's inflate builds 10-bit tables with literal pairs per block, which pays from a few
		kilobytes on (a 600 KB data.npy: 1 ms against zlib's 4); on such members zlib's leaner set-up wins
		(tools/zip_read_time.py: 1.1 against 2.4 ms for the archive)."""
		sizes = [m[3] for m in self._members.values() if m[1] == 8]
		return bool(sizes) and sum(sizes) < 2048 * len(sizes)

	def items(self):
		"""(name, data) of every member.  Archives of substantial members are inflated and checked in ONE call into
		libogdewarp (ogd_inflate_many); archives of many tiny members go member by member through zlib."""
		if self._zf is not None or not self._members or self._small_members():
			for name in self._members:
				yield name, self.read(name)
			return
		from ... import _shim
		rows = [m[1:] for m in self._members.values()]
		try:
			blobs = _shim.inflate_many(bytes(self._buf) if not isinstance(self._buf, bytes) else self._buf, rows)
		except _shim.OgdError as e:
			if e.status != _shim.ERR_INVALID:
				raise
			for name in self._members:  # name the damaged member the way the one-by-one path does
				self.read(name)
			raise zipfile.BadZipFile(str(e))
		yield from zip(self._members, blobs)

	def __enter__(self):
		return self

	def __exit__(self, *exc):
		self.close()

	def close(self):
		if self._zf is not None:
			self._zf.close()


class ZipWriter:
	"""writestr(name, data) as zipfile.ZipFile(f, "w", ZIP_DEFLATED) offers it; the archive is assembled in memory
	and written to `fileobj` by close() (compresslevel is accepted for zipfile's signature; the deflate writer has no
	levels).  Members beyond the plain format's limits (65 534 members, 4 GiB) make
	close() hand everything to the zipfile module, which writes zip64."""

	def __init__(self, fileobj, compression=zipfile.ZIP_DEFLATED, compresslevel=1):
		if compression not in (zipfile.ZIP_DEFLATED, zipfile.ZIP_STORED):
			raise NotImplementedError("compression method %r" % (compression,))
		self._f = fileobj
		self._method = compression
		self._level = compresslevel
		self._members = []  # (name, data)
		self._methods = []  # per member
		self._crcs = []  # per member: given by the caller, or None
		self._closed = False

	def writestr(self, name, data, compress_type=None, crc=None):
		"""compress_type (as zipfile's): ZIP_STORED for a member that is compressed already (a PNG file) — it is
		written as it is instead of being run through the deflate coder to find that out.  crc: the CRC-32 of `data`
		if the caller has it (stored members only; computed here otherwise)."""
		if isinstance(data, str):
			data = data.encode("utf-8")
		if compress_type not in (None, zipfile.ZIP_DEFLATED, zipfile.ZIP_STORED):
			raise NotImplementedError("compression method %r" % (compress_type,))
		# a memoryview is kept as it is (the PNG files of the line export are views into the encoder's one buffer)
		self._members.append((name, data if isinstance(data, (bytes, memoryview)) else bytes(data)))
		self._methods.append(self._method if compress_type is None else compress_type)
		self._crcs.append(None if crc is None else int(crc) & 0xFFFFFFFF)

	def namelist(self):
		return [name for name, _ in self._members]

	def __enter__(self):
		return self

	def __exit__(self, kind, *exc):
		if kind is None:
			self.close()

	def close(self):
		if self._closed:
			return
		self._closed = True
		if len(self._members) >= 0xFFFF or sum(len(d) + len(n) + 128 for n, d in self._members) >= (1 << 31):
			with zipfile.ZipFile(self._f, "w", self._method, compresslevel=self._level) as zf:
				for (name, data), method in zip(self._members, self._methods):
					zf.writestr(name, data, compress_type=method)
			return
		t = time.localtime()
		dos_time = t.tm_hour << 11 | t.tm_min << 5 | t.tm_sec // 2
		dos_date = max(t.tm_year - 1980, 0) << 9 | t.tm_mon << 5 | t.tm_mday
		out, central, offset = [], [], 0
		packed = [None] * len(self._members)
		deflated = [k for k, m in enumerate(self._methods) if m == zipfile.ZIP_DEFLATED]
		if deflated:
			# libogdewarp's entropy-only deflate (csrc/ogd_deflate.h), all members in one call: coordinate text and
			# float32 arrays hold next to nothing for a string matcher to find, and zlib's set-up per member alone
			# costs more than coding a kilobyte
			from ... import _shim
			for k, item in zip(deflated, _shim.deflate_many([self._members[k][1] for k in deflated])):
				packed[k] = item
		for k, (_, data) in enumerate(self._members):
			if packed[k] is None:
				packed[k] = (data, self._crcs[k] if self._crcs[k] is not None else zlib.crc32(data))
		for (name, data), (body, crc), method in zip(self._members, packed, self._methods):
			try:
				raw, flags = name.encode("ascii"), 0
			except UnicodeEncodeError:
				raw, flags = name.encode("utf-8"), 0x0800
			fields = (20, flags, method, dos_time, dos_date, crc, len(body), len(data), len(raw), 0)
			out.append(_LOCAL.pack(b"PK\x03\x04", *fields))
			out.append(raw)
			out.append(body)
			# made by: 3.0 on unix (3 << 8 | 20... zipfile writes 20 / system 3); regular file, mode 0600
			central.append(_CENTRAL.pack(b"PK\x01\x02", 3 << 8 | 20, *fields, 0, 0, 0, 0o600 << 16, offset))
			central.append(raw)
			offset += _LOCAL.size + len(raw) + len(body)
		directory = b"".join(central)
		n = len(self._members)
		out.append(directory)
		out.append(_EOCD.pack(b"PK\x05\x06", 0, 0, n, n, len(directory), offset, 0))
		self._f.writelines(out)  # (no joined copy of the whole archive)
