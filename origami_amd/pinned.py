"""Page-locked (pinned) host arrays for page pixels.

A broadsheet page is 48 MB on its way to the GPU and 49 MB on its way back for 0.07 ms of kernel, so what a
stage pays per page is the copy — and a copy from pageable memory runs at ~14 GB/s where one from pinned memory
runs at ~56 GB/s (tools/pcie_probe.py).  Everything in this package that holds page pixels on the host
(origami_amd.page.Page's decoded scan, the dewarped page, the pipeline's buffers) therefore lives in arrays from
this module: plain numpy arrays whose memory came from ogd_host_alloc (hipHostMalloc).

Pinning memory is slow (milliseconds per page-sized block), so blocks are cached: an array's block goes back to
a process-wide free list when the last numpy view of it dies, and the next request of a similar size takes it
from there.  ref: the reference keeps its pages in pageable numpy / PIL memory (origami/core/page.py:82-88); it
has no device to feed.
"""

import ctypes
import os
import threading

import numpy as np

from . import _shim

_GRANULE = 4 << 20  # blocks come in 4 MiB steps so that pages of similar sizes share them


class _Cache:
	def __init__(self):
		self.lock = threading.Lock()
		self.free = {}  # capacity -> [address]
		self.idle_bytes = 0
		self.limit = int(os.environ.get("OGD_PINNED_CACHE_MB", "512")) << 20
		self.pid = os.getpid()

	def take(self, nbytes):
		cap = max(_GRANULE, -(-nbytes // _GRANULE) * _GRANULE)
		with self.lock:
			self._after_fork()
			# best fit among idle blocks at most a quarter (+ one granule) too large
			fits = [c for c, blocks in self.free.items() if blocks and cap <= c <= cap + cap // 4 + _GRANULE]
			if fits:
				c = min(fits)
				self.idle_bytes -= c
				return self.free[c].pop(), c
		p = ctypes.c_void_p()
		_shim.check(_shim.load().ogd_host_alloc(ctypes.byref(p), cap))
		return p.value, cap

	def give(self, address, cap, pid):
		with self.lock:
			if pid != os.getpid():  # a fork's copy of the parent's block: not ours to hand out or free
				return
			if self.idle_bytes + cap <= self.limit:
				self.free.setdefault(cap, []).append(address)
				self.idle_bytes += cap
				return
		try:
			_shim.load().ogd_host_free(ctypes.c_void_p(address))
		except Exception:
			pass

	def _after_fork(self):
		if self.pid != os.getpid():  # pinned mappings do not survive fork(): forget the parent's free list
			self.free, self.idle_bytes, self.pid = {}, 0, os.getpid()

	def trim(self):
		"""Unpin every idle block; returns the bytes released."""
		with self.lock:
			self._after_fork()
			blocks = [(a, c) for c, lst in self.free.items() for a in lst]
			self.free, self.idle_bytes = {}, 0
		lib = _shim.load()
		for a, _ in blocks:
			lib.ogd_host_free(ctypes.c_void_p(a))
		return sum(c for _, c in blocks)


_cache = _Cache()


class _Block:
	"""Owner of one pinned block; numpy keeps it alive as the base of every view (array interface protocol)."""

	def __init__(self, address, nbytes, cap):
		self.address, self.nbytes, self.cap = address, nbytes, cap
		self.pid = os.getpid()
		self.__array_interface__ = dict(shape=(nbytes,), typestr="|u1", data=(address, False), version=3)

	def __del__(self):
		try:
			_cache.give(self.address, self.cap, self.pid)
		except Exception:  # interpreter shutdown
			pass


def empty(shape, dtype=np.uint8):
	"""np.empty(shape, dtype) in pinned host memory (C order).  Raises OgdError without a GPU."""
	dtype = np.dtype(dtype)
	shape = (int(shape),) if np.isscalar(shape) else tuple(int(n) for n in shape)
	nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
	address, cap = _cache.take(max(nbytes, 1))
	flat = np.asarray(_Block(address, max(nbytes, 1), cap))
	return flat[:nbytes].view(dtype).reshape(shape)


def available():
	"""Can pinned memory be had at all (is there a GPU for it to be pinned to)?"""
	return _shim.device_count() > 0


def empty_if_available(shape, dtype=np.uint8):
	"""empty() on a GPU box; an ordinary pageable array where there is no device (the host-only decoders of this
	package — PNG inflate, zip members — also run and are tested there)."""
	if available():
		try:
			return empty(shape, dtype)
		except _shim.OgdError:
			pass
	return np.empty(shape, dtype=dtype)


def block_of(a):
	"""The pinned block under a numpy array, or None for ordinary memory."""
	while isinstance(a, np.ndarray):
		a = a.base
	return a if isinstance(a, _Block) else None


def is_pinned(a):
	return block_of(a) is not None


def as_pinned(a):
	"""`a` itself if it is C-contiguous in pinned memory, else a pinned copy (one host memcpy)."""
	a = np.asarray(a)
	if a.flags.c_contiguous and is_pinned(a):
		return a
	out = empty(a.shape, a.dtype)
	np.copyto(out, a)
	return out


def trim():
	return _cache.trim()
