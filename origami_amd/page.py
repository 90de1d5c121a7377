"""A page image and, given a dewarping grid, its dewarped form (ref: origami/core/page.py:65-117).

The reference remaps the full page as soon as a Page is opened with a grid (page.py:86-88); here that
happens on the GPU (Dewarper: K1a + K2 fused) and on first use of `dewarped`, so that stages which only
cut lines out of the warped page (block.py:255) do not pay for a remap they never look at."""

import collections
import os
import threading
from pathlib import Path

import numpy as np
import PIL.Image

from . import _shim
from .dewarp import Dewarper, Geometry


class ResidentPages:
	"""Pages kept in device memory across Page objects: an LRU of `_shim.DevicePage` keyed by (path, mtime, size) of
	the scan's file and bounded in bytes (OGD_PAGE_CACHE_MB, default 1024; 0 switches it off).

	In the reference every stage that looks at pixels opens the page and remaps it again (ref: origami/batch/core/
	io.py:275-280, origami/core/page.py:82-88), and every stage is a process of its own.  Here stages can run in ONE
	process (`python -m origami_amd.batch.chain`), and then a page is read, decoded and uploaded once for all of
	them: `utils.dewarp` leaves it resident, `export.lines` cuts its lines from it.  A page whose file changed has
	another key; an evicted page is freed when the last Page object that uses it lets go of it."""

	def __init__(self, budget_bytes):
		self.budget = int(budget_bytes)
		self._pages = collections.OrderedDict()
		self._bytes = 0
		self._lock = threading.Lock()
		self.hits = self.misses = self.evictions = 0

	@property
	def enabled(self):
		return self.budget > 0

	@staticmethod
	def key(path):
		st = os.stat(path)
		return (os.path.realpath(path), st.st_mtime_ns, st.st_size)

	def __contains__(self, key):
		with self._lock:
			return key in self._pages

	def __len__(self):
		return len(self._pages)

	@property
	def nbytes(self):
		return self._bytes

	def get(self, key):
		with self._lock:
			page = self._pages.get(key)
			if page is None:
				self.misses += 1
				return None
			self._pages.move_to_end(key)
			self.hits += 1
			return page

	def put(self, key, page):
		if not self.enabled or page.nbytes > self.budget:
			return
		with self._lock:
			old = self._pages.pop(key, None)
			if old is not None:
				self._bytes -= old.nbytes
			self._pages[key] = page
			self._bytes += page.nbytes
			while self._bytes > self.budget:
				_, gone = self._pages.popitem(last=False)
				self._bytes -= gone.nbytes
				self.evictions += 1

	def clear(self):
		with self._lock:
			self._pages.clear()
			self._bytes = 0


resident = ResidentPages(int(float(os.environ.get("OGD_PAGE_CACHE_MB", "1024")) * (1 << 20)))


def _is_image(path):
	try:
		with PIL.Image.open(str(path)) as im:
			im.verify()
		return True
	except Exception:
		return False


def _find_image_path(path):
	path = Path(path)
	if path.exists():
		return path
	# do not be picky about the image extension (ref: page.py:70-78)
	candidates = [
		c for c in path.parent.glob(path.stem + ".*")
		if c.name.endswith(".jp2") or (c.is_file() and _is_image(c))]
	if len(candidates) != 1:
		raise FileNotFoundError(path)
	return candidates[0]


class Page:
	def __init__(self, path, dewarping_transform=None):
		path = _find_image_path(path)
		# The reference decodes and converts here (ref: page.py:84).  PIL reads only the header on open(),
		# and the dewarp stage asks a warped page for nothing but its size, so the pixels are decoded on
		# first use: 0.19 s per broadsheet PNG that this stage never spends.
		self._opened = PIL.Image.open(str(path))
		self._warped = None
		self._pixels = None  # the array self._warped was made from, when this class decoded the scan itself
		self._device_page = None
		self._grid = dewarping_transform
		self._dewarper = None
		self._file_bytes = None
		self._key = None
		self._filename = getattr(self._opened, "filename", None)

	def _resident_key(self):
		"""Key of this page in `resident`, or None (cache off, or a page that did not come from a file)."""
		if self._key is None and resident.enabled:
			if self._filename:
				try:
					self._key = ResidentPages.key(self._filename)
				except OSError:
					pass
		return self._key

	def _read_file(self):
		if self._file_bytes is None:
			self._file_bytes = Path(self._opened.filename).read_bytes()
		return self._file_bytes

	def prefetch(self):
		"""What a stage that will look at this page's pixels can start before it gets to the page (the batch runtime
		calls it for page k + 1 on a side thread while page k is on the device): a JPEG scan the device decodes is
		only read from disk — decode, dewarp and encode happen in one device call later —, any other file is decoded
		now (PNG: libogdewarp's inflate, else PIL), into pinned memory that goes up at the link's rate."""
		if self._warped is not None or self._device_page is not None:
			return
		key = self._resident_key()
		if key is not None and key in resident:
			return  # an earlier stage of this process left the page on the device
		if self._on_device_format():
			self._read_file()
		else:
			self.warped

	@property
	def warped(self):
		if self._warped is None:
			self._warped = self._decode_on_device() or self._decode_png() or self._opened.convert("L")
			self._opened = None
			self._file_bytes = None
		return self._warped

	def _decode_on_device(self):
		"""A baseline JPEG scan (grayscale or YCbCr colour) is decoded by ogd_jpeg_decode_gray: the pixels PIL would give, bit for
		bit, in a few ms instead of 0.15 s.  Any other file (PNG, progressive or CMYK JPEG, JPEG 2000 ...)
		returns None and goes through PIL as in the reference (ref: page.py:82-84)."""
		if not self._on_device_format():
			return None
		data = self._read_file()
		pixels = _shim.jpeg_decode_gray(data)
		if pixels is None:
			return None
		self._pixels = pixels
		return PIL.Image.fromarray(pixels)

	def _decode_png(self):
		"""An 8-bit grey or RGB PNG without interlacing goes through libogdewarp's own inflate (ogd_png_decode_u8,
		host code: a deflate stream is serial, what counts is the decoder's inner loop): the pixels PIL gives, in a
		third of the time.  Other kinds of PNG, and files the decoder finds fault with, return None and go through
		PIL as in the reference (ref: page.py:82-84), which then raises its own errors."""
		if not (self._opened.format == "PNG" and self._opened.mode in ("L", "RGB") and getattr(self._opened, "filename", None)):
			return None
		try:
			pixels = _shim.png_decode(Path(self._opened.filename).read_bytes())
		except (_shim.OgdError, MemoryError):  # MemoryError: a header that claims 2^30 x 2^30 pixels
			return None
		im = PIL.Image.fromarray(pixels)
		if im.mode != "L":
			return im.convert("L")
		self._pixels = pixels
		return im

	@property
	def dewarped(self):
		return self.dewarper.dewarped if self._grid is not None else None

	def dewarped_jpeg(self, quality=75):
		"""`self.dewarped` as the JPEG file PIL would save (bytes), encoded on the device.  For a page that is
		still an undecoded grayscale JPEG scan, decode, dewarp and encoding run back to back on the device
		(ogd_jpeg_dewarp_jpeg_gray): only the two files cross PCIe."""
		if self._grid is None:
			return None
		if self._resident_key() is not None:  # through the resident page (decoded / uploaded now if no stage has yet)
			return self.device_page.dewarp_jpeg(self._grid.points("sample"), self._grid.resolution, quality)
		if self._warped is None and self._on_device_format():
			data = self._read_file()
			out = _shim.jpeg_dewarp_jpeg_gray(data, self._grid.points("sample"), self._grid.resolution, quality)
			if out is not None:
				return out
		return self.dewarper.dewarped_jpeg(quality)

	def _on_device_format(self):
		# "L": one-component scans; "RGB": what PIL calls a three-component (JFIF YCbCr) scan.  Whether the device
		# decoder covers the file (baseline, JFIF, interleaved ...) is ogd_jpeg_info's decision; Adobe / CMYK /
		# progressive files come back None from the decode call and go through PIL.
		return (self._opened is not None and self._opened.format == "JPEG" and self._opened.mode in ("L", "RGB")
			and bool(getattr(self._opened, "filename", None)))

	def size(self, dewarped):
		if dewarped:
			return self.dewarped.size
		return self._warped.size if self._warped is not None else self._opened.size

	def geometry(self, dewarped):
		return Geometry(*self.size(dewarped))

	def pixels(self, dewarped):
		return np.array(self.dewarped if dewarped else self.warped)

	@property
	def dewarper(self):
		if self._dewarper is None and self._grid is not None:
			if self._warped is None and self._resident_key() is not None:
				# the pixels stay where they are (on the device, or about to be put there once for every stage of this
				# process); the PIL image of the warped page is made only if somebody asks the Dewarper for it
				self._dewarper = Dewarper(None, self._grid, device_page=lambda: self.device_page, warped=lambda: self.warped)
			else:
				self._dewarper = Dewarper(self.warped, self._grid, pixels=self._pixels)
		return self._dewarper

	@property
	def device_page(self):
		"""The warped page resident in device memory; line images are cut from it (block.py:255)."""
		if self._device_page is None:
			key = self._resident_key()
			if key is not None:
				self._device_page = resident.get(key)
			if self._device_page is None:
				if self._warped is None and self._on_device_format():  # an undecoded JPEG scan never visits the host
					self._device_page = _shim.DevicePage.from_jpeg(self._read_file())
				if self._device_page is None:
					warped = self.warped
					self._device_page = _shim.DevicePage(self._pixels if self._pixels is not None else np.asarray(warped))
				if key is not None:
					resident.put(key, self._device_page)
		return self._device_page
