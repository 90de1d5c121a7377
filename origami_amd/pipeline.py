"""The overlapped page pipeline: pinned host pixels -> H2D -> fused dewarp (+ the page's contours) -> D2H.

The reference dewarps a page the moment it is opened (ref: origami/batch/core/io.py:275-280 ->
origami/core/page.py:82-88 -> origami/core/dewarp.py:608-614) and keeps many pages going through a process pool
(ref: origami/batch/core/processor.py:476-493).  On MI355X the fused kernel needs 0.07 ms for a broadsheet page
whose pixels need 1.0 ms on PCIe each way, so what keeps pages going is copies that overlap — with each other
(the link is full duplex) and with the kernels:

    upload stream     H2D  page k+1 |  page k+2 | ...
    kernel stream          dewarp k |  dewarp k+1 | ...        (ogd_dewarp_u8_dev, ogd_tri_eval_ragged_dev)
    download stream        D2H page k-1 |  page k | ...

with `slots` device slots (a slot = device buffers for one page in flight + three events) and AT MOST `slots`
jobs in flight: before a slot takes a new job the host takes delivery of its previous one.  That bound matters:
with hundreds of copies queued the engines interleave them badly and the same loop runs 2.5 - 5x slower
(DESIGN.md §5).  Three streams, because a process has four hardware queues (GPU_MAX_HW_QUEUES) and the
host-buffer entry points of the library use the null stream.

    with PagePipeline() as pipe:
        for result in pipe.map(jobs):           # jobs: iterable of PageJob, results come in order
            result.page                          # (Hd, Wd[, C]) uint8 in pinned memory, the consumer's to keep

Pixels handed in from pinned memory (origami_amd.pinned; origami_amd.page.Page decodes into it) go up without a
host copy; ordinary arrays are first copied into a pinned buffer (one memcpy, ~5 ms for a broadsheet page: the
difference between 1 and 6 ms per page, so decode into pinned memory).
"""

import ctypes
from collections import deque
from ctypes import c_void_p

import numpy as np

from . import _shim, pinned
from ._shim import INTER_LINEAR, check


def _lib():
	return _shim.load()


class Stream:
	"""A non-blocking HIP stream owned by the library (ogd_stream_create); `ptr` is what *_dev entry points take."""

	def __init__(self, ptr=None):
		self._own = ptr is None
		if self._own:
			p = c_void_p()
			check(_lib().ogd_stream_create(ctypes.byref(p)))
			ptr = p.value
		self.ptr = c_void_p(ptr)

	def wait(self, event):
		check(_lib().ogd_stream_wait_event(self.ptr, event.ptr))

	def synchronize(self):
		check(_lib().ogd_stream_sync(self.ptr))

	def close(self):
		if self._own and self.ptr is not None and _shim._lib is not None:
			_shim._lib.ogd_stream_destroy(self.ptr)
		self.ptr = None

	__del__ = close


class Event:
	def __init__(self, timing=False):
		p = c_void_p()
		check(_lib().ogd_event_create(ctypes.byref(p), int(bool(timing))))
		self.ptr = c_void_p(p.value)

	def record(self, stream):
		check(_lib().ogd_event_record(self.ptr, stream.ptr))

	def synchronize(self):
		check(_lib().ogd_event_sync(self.ptr))

	def done(self):
		flag = ctypes.c_int(0)
		check(_lib().ogd_event_query(self.ptr, ctypes.byref(flag)))
		return bool(flag.value)

	def elapsed_ms(self, later):
		ms = ctypes.c_float(0)
		check(_lib().ogd_event_elapsed_ms(self.ptr, later.ptr, ctypes.byref(ms)))
		return ms.value

	def close(self):
		if self.ptr is not None and _shim._lib is not None:
			_shim._lib.ogd_event_destroy(self.ptr)
		self.ptr = None

	__del__ = close


class DeviceBuffer:
	"""Grow-only device block (ogd_dev_alloc)."""

	def __init__(self, nbytes=0):
		self.ptr, self.nbytes = None, 0
		if nbytes:
			self.reserve(nbytes)

	def reserve(self, nbytes):
		"""Make room for nbytes; returns True if the block was replaced (the caller made sure it is idle)."""
		if nbytes <= self.nbytes:
			return False
		self.close()
		nbytes = -(-nbytes // (2 << 20)) * (2 << 20)
		p = c_void_p()
		check(_lib().ogd_dev_alloc(ctypes.byref(p), nbytes))
		self.ptr, self.nbytes = p.value, nbytes
		return True

	def close(self):
		if self.ptr is not None and _shim._lib is not None:
			_shim._lib.ogd_dev_free(c_void_p(self.ptr))
		self.ptr, self.nbytes = None, 0

	__del__ = close


class DeviceArray:
	"""Something already resident on the device (pixels of a page, a node grid): pointer + shape.  `keep` holds
	whatever owns the memory (a DeviceBuffer, a torch tensor ...)."""

	def __init__(self, ptr, shape, keep=None):
		self.ptr, self.shape, self.keep = int(ptr), tuple(shape), keep

	@classmethod
	def upload(cls, a):
		a = np.ascontiguousarray(a)
		buf = DeviceBuffer(max(a.nbytes, 16))
		check(_lib().ogd_memcpy_h2d(c_void_p(buf.ptr), c_void_p(a.ctypes.data), a.nbytes, None))
		check(_lib().ogd_stream_sync(None))
		return cls(buf.ptr, a.shape, keep=buf)


class PageJob:
	"""One page for the pipeline.

	pixels   (H, W) or (H, W, C) uint8, C in {1, 3, 4}: numpy (pinned memory goes up as it lies) or DeviceArray
	nodes    (h, w, 2) float32 node grid (Grid.points("sample")): numpy or DeviceArray
	cell     grid cell size; sx, sy: the dewarp_image scale (ref: dewarp.py:621-623); interpolation: INTER_*
	rings    None or (transformer, flat (n, 2) float64, offsets (r + 1,) int64): the page's contour rings through the
	         forward map (K3) in the same job — flat / offsets may be DeviceArrays
	tag      anything; comes back on the result
	"""

	def __init__(self, pixels, nodes, cell, sx=1.0, sy=1.0, interpolation=INTER_LINEAR, rings=None, tag=None):
		self.pixels, self.nodes, self.cell = pixels, nodes, int(cell)
		self.sx, self.sy, self.interpolation = float(sx), float(sy), int(interpolation)
		self.rings, self.tag = rings, tag


class PageResult:
	def __init__(self, job, slot):
		self.tag = job.tag
		self.page = None  # dewarped pixels, pinned host memory
		self.rings = None  # (n, 2) float64 warped ring points, or None
		self._slot = slot
		self._held = []  # host arrays the copies of this job read from / write to
		self._done = False

	@property
	def done(self):
		return self._done


class _Slot:
	def __init__(self):
		self.src, self.dst, self.nodes = DeviceBuffer(), DeviceBuffer(), DeviceBuffer()
		self.q, self.off, self.pts = DeviceBuffer(), DeviceBuffer(), DeviceBuffer()
		self.ev_in, self.ev_k, self.ev_out = Event(), Event(), Event()
		self.result = None  # the job in flight in this slot

	def close(self):
		for b in (self.src, self.dst, self.nodes, self.q, self.off, self.pts):
			b.close()
		for e in (self.ev_in, self.ev_k, self.ev_out):
			e.close()


def _ring_pieces(offsets, piece=64):
	"""Ring offsets -> offsets of pieces of at most `piece` vertices (one wavefront each, include/ogdewarp.h)."""
	offsets = np.asarray(offsets, dtype=np.int64)
	if len(offsets) < 2 or int(np.diff(offsets).max(initial=0)) <= piece:
		return offsets
	starts = [np.arange(a, b, piece) for a, b in zip(offsets[:-1].tolist(), offsets[1:].tolist())]
	return np.concatenate(starts + [offsets[-1:]]).astype(np.int64)


class PagePipeline:
	def __init__(self, slots=3, streams=None):
		"""streams: (upload, kernels, download) as Stream objects or raw hipStream_t values to use instead of three
		new ones — a process that already holds streams must stay within its four hardware queues."""
		if slots < 1:
			raise ValueError("slots must be >= 1")
		_shim.check(_lib().ogd_init(-1))
		self._own_streams = streams is None
		if streams is None:
			streams = (Stream(), Stream(), Stream())
		self._s_in, self._s_k, self._s_out = [s if isinstance(s, Stream) else Stream(s) for s in streams]
		self._slots = [_Slot() for _ in range(slots)]
		self._next = 0
		self._pending = deque()  # results in submission order
		self.jobs_done = 0

	# ---- one job ---------------------------------------------------------------------------------------

	def submit(self, job):
		"""Queue a job on the three streams; returns its PageResult (filled in once wait() returned it)."""
		lib = _lib()
		slot = self._slots[self._next]
		self._next = (self._next + 1) % len(self._slots)
		if slot.result is not None:
			self._deliver(slot.result)  # at most `slots` jobs in flight (and the slot's buffers are idle now)
		res = PageResult(job, slot)

		# geometry of the job
		pix = job.pixels
		shape = pix.shape
		if len(shape) == 2:
			H, W, C = shape[0], shape[1], 1
		elif len(shape) == 3:
			H, W, C = shape
		else:
			raise ValueError("pixels must be (H, W) or (H, W, C)")
		nshape = job.nodes.shape
		if len(nshape) != 3 or nshape[2] != 2:
			raise ValueError("nodes must have shape (h, w, 2)")
		h, w = nshape[:2]
		Hd, Wd = h * job.cell, w * job.cell
		out = pinned.empty((Hd, Wd) if len(shape) == 2 else (Hd, Wd, C), np.uint8)
		res.page = out

		# upload: pixels, node grid, ring points
		if isinstance(pix, DeviceArray):
			d_src, res_keep = pix.ptr, pix
			res._held.append(res_keep)
		else:
			pix = np.asarray(pix)
			if pix.dtype != np.uint8:
				raise TypeError("pixels must be uint8")
			host = pinned.as_pinned(pix)
			res._held.append(host)
			slot.src.reserve(host.nbytes)
			d_src = slot.src.ptr
			check(lib.ogd_memcpy_h2d(c_void_p(d_src), c_void_p(host.ctypes.data), host.nbytes, self._s_in.ptr))
		if isinstance(job.nodes, DeviceArray):
			d_nodes = job.nodes.ptr
			res._held.append(job.nodes)
		else:
			nodes = pinned.as_pinned(np.asarray(job.nodes, dtype=np.float32))
			res._held.append(nodes)
			slot.nodes.reserve(nodes.nbytes)
			d_nodes = slot.nodes.ptr
			check(lib.ogd_memcpy_h2d(c_void_p(d_nodes), c_void_p(nodes.ctypes.data), nodes.nbytes, self._s_in.ptr))
		ring_call = None
		if job.rings is not None:
			tf, flat, offsets = job.rings
			if isinstance(flat, DeviceArray):
				n_pts, d_q, d_off, n_pieces = flat.shape[0], flat.ptr, offsets.ptr, offsets.shape[0] - 1
				res._held += [flat, offsets]
			else:
				flat = pinned.as_pinned(np.asarray(flat, dtype=np.float64).reshape(-1, 2))
				pieces = pinned.as_pinned(_ring_pieces(offsets))
				n_pts, n_pieces = len(flat), len(pieces) - 1
				res._held += [flat, pieces]
				slot.q.reserve(max(flat.nbytes, 16))
				slot.off.reserve(pieces.nbytes)
				d_q, d_off = slot.q.ptr, slot.off.ptr
				if n_pts:
					check(lib.ogd_memcpy_h2d(c_void_p(d_q), c_void_p(flat.ctypes.data), flat.nbytes, self._s_in.ptr))
				check(lib.ogd_memcpy_h2d(c_void_p(d_off), c_void_p(pieces.ctypes.data), pieces.nbytes, self._s_in.ptr))
			if n_pts:
				slot.pts.reserve(n_pts * 16)
				inner, extra = tf._interp.triangulations
				bounds = np.array(tf._interp.bounds, dtype=np.float64)
				res._held += [tf, bounds]
				res.rings = pinned.empty((n_pts, 2), np.float64)
				ring_call = (inner.handle if inner is not None else None, extra.handle, bounds, d_q, d_off, n_pieces)
			else:
				res.rings = np.empty((0, 2), dtype=np.float64)
		slot.ev_in.record(self._s_in)

		# kernels
		self._s_k.wait(slot.ev_in)
		slot.dst.reserve(out.nbytes)
		check(lib.ogd_dewarp_u8_dev(
			c_void_p(d_src), H, W, C, W * C, c_void_p(d_nodes), h, w, job.cell, job.sx, job.sy,
			c_void_p(slot.dst.ptr), Wd * C, job.interpolation, self._s_k.ptr))
		if ring_call is not None:
			inner_h, extra_h, bounds, d_q, d_off, n_pieces = ring_call
			check(lib.ogd_tri_eval_ragged_dev(
				inner_h, extra_h, c_void_p(bounds.ctypes.data), c_void_p(d_q), c_void_p(d_off), n_pieces,
				c_void_p(slot.pts.ptr), self._s_k.ptr))
		slot.ev_k.record(self._s_k)

		# download
		self._s_out.wait(slot.ev_k)
		check(lib.ogd_memcpy_d2h(c_void_p(out.ctypes.data), c_void_p(slot.dst.ptr), out.nbytes, self._s_out.ptr))
		if ring_call is not None:
			check(lib.ogd_memcpy_d2h(
				c_void_p(res.rings.ctypes.data), c_void_p(slot.pts.ptr), res.rings.nbytes, self._s_out.ptr))
		slot.ev_out.record(self._s_out)

		slot.result = res
		self._pending.append(res)
		return res

	def _deliver(self, res):
		if res._done:
			return
		res._slot.ev_out.synchronize()
		res._done = True
		res._held = []
		if res._slot.result is res:
			res._slot.result = None
		res._slot = None
		self.jobs_done += 1

	def wait(self, res=None):
		"""Block until `res` (default: the oldest job still pending) is complete and return it."""
		if res is None:
			if not self._pending:
				return None
			res = self._pending[0]
		self._deliver(res)
		while self._pending and self._pending[0]._done:
			self._pending.popleft()
		return res

	@property
	def in_flight(self):
		return sum(1 for r in self._pending if not r._done)

	def map(self, jobs):
		"""Run an iterable of PageJob through the pipeline; yields PageResults in job order, keeping up to `slots`
		jobs in flight (the next job is drawn from the iterable — i.e. the next page is decoded — while the
		previous ones are on the device)."""
		queue = deque()
		for job in jobs:
			queue.append(self.submit(job))
			if len(queue) >= len(self._slots):
				yield self.wait(queue.popleft())
		while queue:
			yield self.wait(queue.popleft())

	def drain(self):
		while self._pending:
			self.wait()

	def close(self):
		if getattr(self, "_slots", None) is None:
			return
		try:
			self.drain()
		finally:
			for s in self._slots:
				s.close()
			if self._own_streams:
				for s in (self._s_in, self._s_k, self._s_out):
					s.close()
			self._slots = None

	def __enter__(self):
		return self

	def __exit__(self, *exc):
		self.close()

	def __del__(self):
		try:
			self.close()
		except Exception:
			pass


def dewarp_pages(pages, slots=3):
	"""[(pixels, grid)] or [(pixels, grid, rings)] -> generator of PageResult, in order, through one PagePipeline
	(`grid`: origami_amd.dewarp.Grid; rings: (flat, offsets) warped through grid.transformer)."""
	def jobs():
		for k, item in enumerate(pages):
			pixels, grid = item[0], item[1]
			rings = None
			if len(item) > 2 and item[2] is not None:
				rings = (grid.transformer, item[2][0], item[2][1])
			yield PageJob(pixels, grid.points("sample"), grid.resolution, rings=rings, tag=k)

	with PagePipeline(slots) as pipe:
		yield from pipe.map(jobs())
