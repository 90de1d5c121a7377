"""Batch runtime of a stage: directory walk, readiness / skip rules, page locks, runtime.json, and the
fan-out of pages over worker processes — one worker process per GPU.

ref: origami/batch/core/processor.py:230-662 (Processor).  The CLI keeps the reference's option set
(processor.py:276-347); `--gpus` and `--devices` are additive.  Where the reference forks a
multiprocessing.Pool over chunks of 25 pages (processor.py:476-493), this runtime *spawns* its workers
(no HIP state is ever inherited), binds worker r to GPU devices[r % len(devices)] before its first
HIP call, and lets the workers pull chunks from one host-side queue: pages are independent, so there is
no collective and nothing crosses xGMI.
"""

import json
import logging
import multiprocessing
import os
import queue as queue_mod
import re
import sys
import time
import traceback
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import click

from .io import Artifact, AtomicFileWriter, find_data_path
from .mutex import DatabaseMutex, DummyMutex, FileMutex

IMAGE_SUFFIXES = (".jpg", ".png", ".tif", ".tiff")


def chunks(items, size):
	"""Consecutive slices of at most `size` items."""
	return (items[at:at + size] for at in range(0, len(items), size))


def _worker_main(processor_cls, options, data_path, rank, device, task_q, done_q):
	"""Body of one spawned worker: bind to its GPU, then pull chunks of page paths until the queue
	hands out None."""
	try:
		numa_node = None
		if device is not None:
			os.environ["OGD_DEVICE"] = str(device)
			# run on the cores next to the GPU before anything allocates pinned page buffers (first touch)
			from ... import numa
			numa_node = numa.bind_to_gpu_numa_node(device)
		processor = processor_cls(options)
		processor._numa_node = numa_node
		processor._rank = rank
		processor._device = device
		processor._mutex = processor._make_mutex(data_path)
		while True:
			chunk = task_q.get()
			if chunk is None:
				break
			prepared = []  # (readers / writers are rebuilt here: they do not cross the process boundary)
			for i, p in chunk:
				kwargs = processor.prepare_process(Path(p))
				if kwargs is not False:
					prepared.append((i, Path(p), kwargs))
			for i, p in processor._trigger_process(prepared):
				done_q.put((rank, i, str(p)))
			taken = set(j for j, _, _ in prepared)
			for i, p in chunk:  # pages a concurrent worker finished meanwhile still count as progress
				if i not in taken:
					done_q.put((rank, i, str(p)))
	except BaseException:
		logging.exception("worker %d died" % rank)
		done_q.put((rank, -1, traceback.format_exc()))
		raise
	finally:
		done_q.put((rank, None, None))


# The stage CLI every Processor shares: option names, types and defaults are the reference's (processor.py:276-347),
# so that existing command lines keep working; --gpus and --devices are this build's additions.
_CLI = (
	("--processes", dict(type=int, default=1, help="Worker processes (CPU-side parallelism).")),
	("--alive", dict(type=int, default=600, help="Seconds without progress after which the workers are stopped.")),
	("--name", dict(type=str, default="", help="Regular expression a page path must match to be processed.")),
	("--lock-strategy", dict(
		type=click.Choice(["FILE", "DB", "NONE"], case_sensitive=False), default="DB",
		help="Where page locks live: lock files, a SQLite database, or nowhere.")),
	("--lock-level", dict(
		type=click.Choice(["PAGE", "TASK"], case_sensitive=False), default="PAGE",
		help="What a lock covers: the page, or this stage on the page.")),
	("--lock-database", dict(type=click.Path(), required=False, help="Path of the lock database (DB strategy).")),
	("--lock-timeout", dict(type=int, default=60, required=False, help="Seconds to wait for a lock (raise on NFS).")),
	("--max-lock-age", dict(type=int, default=600, required=False, help="Locks older than this many seconds are stale.")),
	("--overwrite", dict(is_flag=True, default=False, help="Redo pages whose results exist.")),
	("--profile", dict(is_flag=True, default=False, help="(accepted for compatibility; not supported)")),
	("--plain", dict(is_flag=True, default=False, help="One line per finished page, for pipes.")),
	("--debug-write", dict(is_flag=True, default=False, help="(accepted for compatibility; not supported)")),
	("--track-changes", dict(type=str, default="", help="(accepted for compatibility; not supported)")),
	("--gpus", dict(type=int, default=0, help="GPUs to spread the pages over with spawned workers (0: work in this process).")),
	("--workers-per-gpu", dict(
		type=int, default=0,
		help="Worker processes per GPU (0: as many as half the usable cores allow, at most 8; the stages are host-bound).")),
	("--devices", dict(type=str, default="", help="Comma separated GPU indices to use (default: all visible).")),
)

_LOCK_ACTORS = dict(PAGE=lambda processor: "page", TASK=lambda processor: processor.processor_name)


class Processor:
	def __init__(self, options):
		self._options = dict(options)
		get = self._options.get
		self._overwrite = bool(get("overwrite", False))
		self._processes = get("processes", 1)
		self._timeout = get("alive", 600)
		self._name = get("name", "")
		self._verbose = False
		self._plain = get("plain")
		self._print_paths = bool(self._plain)

		self._lock_strategy = (get("lock_strategy") or "DB").upper()
		self._lock_level = (get("lock_level") or "PAGE").upper()
		self._lock_timeout = get("lock_timeout", 60)
		self._max_lock_age = get("max_lock_age")
		self._lock_database = get("lock_database")
		self._lock_chunk_size = 25  # pages locked (and handed to a worker) together, ref: processor.py:451-453
		self._mutex = None
		if self._lock_strategy not in ("DB", "FILE", "NONE"):
			raise ValueError(self._lock_strategy)
		if self._lock_level not in _LOCK_ACTORS:
			raise ValueError(self._lock_level)

		self._gpus = get("gpus", 0) or 0
		self._workers_per_gpu = get("workers_per_gpu", 0) or 0
		self._devices = get("devices") or ""
		self._numa_node = None
		self._rank = 0
		self._device = None
		self._prefetcher = None
		if get("profile") or get("track_changes") or get("debug_write"):
			logging.warning("--profile, --track-changes and --debug-write are not supported by this build.")

	@staticmethod
	def options(command):
		"""Decorator that hangs the shared options on a click command."""
		for flag, spec in reversed(_CLI):
			command = click.option(flag, **spec)(command)
		return command

	# ---- what a stage provides -------------------------------------------------------------------------

	@property
	def processor_name(self):
		return type(self).__name__

	def artifacts(self):
		raise NotImplementedError()

	def process(self, page_path, **kwargs):
		raise NotImplementedError()

	def should_process(self, page_path):
		return True

	def prefetch(self, page_path, **kwargs):
		"""Called with the NEXT page's readers / writers on a side thread while process() works on the current page:
		a stage starts here whatever its process() would wait for first — reading and inflating the page's
		archives, decoding the scan into pinned memory — so that the host work of page k + 1 overlaps the device
		work of page k.  Whatever it loads must land in the readers' own caches (Reader._once): process() then finds
		it there.  Errors are dropped; process() meets them again in context.  Default: nothing.
		The reference gets this overlap from its process pool alone (ref: origami/batch/core/processor.py:476-493);
		a GPU worker has a device to keep busy as well."""

	def is_image(self, path):
		if path.suffix.lower() in IMAGE_SUFFIXES:
			return True
		import PIL.Image
		try:
			with PIL.Image.open(str(path)) as im:
				im.verify()
		except Exception:
			return False
		return True

	def lock_or_open(self, path, mode):
		return open(path, mode)

	# ---- one page ---------------------------------------------------------------------------------------

	def prepare_process(self, page_path):
		"""The page's readers and writers by argument name, or False if an input is missing or the outputs exist
		(ref: processor.py:381-393)."""
		writer = AtomicFileWriter(overwrite=self._overwrite)
		files = {}
		for arg, spec in self.artifacts():
			handle = spec.instantiate(page_path=page_path, processor=self, file_writer=writer)
			handle.fix_inconsistent()
			if not handle.is_ready():
				if self._verbose:
					print("skipping %s: missing %s" % (page_path, handle.missing))
				return False
			files[arg] = handle
		return files

	def _run_page(self, page, files, locked):
		"""process() for one page under its lock, with the outcome recorded in the page's runtime.json
		(ref: processor.py:395-423: COMPLETED with the elapsed time, or FAILED with the traceback)."""
		try:
			if not locked:
				logging.warning("failed to obtain lock for %s. ignoring." % page)
				return
			if not all(f.is_ready() for f in files.values()):
				return  # another worker did this page between queueing and locking
			started = time.perf_counter()
			find_data_path(page).mkdir(exist_ok=True)
			record = self.process(page, **files) or {}
			record["status"] = "COMPLETED"
			record["elapsed"] = round(time.perf_counter() - started, 2)
			if self._device is not None:
				record["device"] = self._device
			self._update_runtime_info(page, {self.processor_name: record})
		except KeyboardInterrupt:
			logging.exception("Interrupted at %s." % page)
			raise
		except Exception:
			logging.exception("Failed to process %s." % page)
			self._update_runtime_info(page, {self.processor_name: dict(status="FAILED", traceback=traceback.format_exc())})
		finally:
			files.clear()

	def _trigger_process(self, chunk):
		"""Runs the (index, page, files) items of a chunk under their locks; yields (index, page) as each is over."""
		if not chunk:
			return
		actor = _LOCK_ACTORS[self._lock_level](self)
		# lock files are taken page by page, database locks for the chunk in one transaction
		groups = [[item] for item in chunk] if isinstance(self._mutex, FileMutex) else [chunk]
		for group in groups:
			with self._mutex.lock(actor, [str(page) for _, page, _ in group]) as locked:
				ahead = None
				for k, (index, page, files) in enumerate(group):
					mine, ahead = ahead, None
					if locked and k + 1 < len(group):
						ahead = self._start_prefetch(group[k + 1][1], group[k + 1][2])
					if mine is not None:
						mine.result()  # never raises (_prefetch_quietly)
					self._run_page(page, files, locked)
					yield index, page

	def _start_prefetch(self, page, files):
		if type(self).prefetch is Processor.prefetch or os.environ.get("OGD_PREFETCH", "1") == "0":
			return None
		if self._prefetcher is None:
			self._prefetcher = ThreadPoolExecutor(max_workers=1, thread_name_prefix="prefetch")
		return self._prefetcher.submit(self._prefetch_quietly, page, files)

	def _prefetch_quietly(self, page, files):
		try:
			if all(f.is_ready() for f in files.values()):
				self.prefetch(page, **files)
		except Exception:
			logging.debug("prefetch of %s failed:\n%s" % (page, traceback.format_exc()))

	# ---- queue ---------------------------------------------------------------------------------------

	@staticmethod
	def _listed_pages(listing):
		"""Paths named one per line in a text file."""
		with open(listing, "r") as f:
			for line in f:
				if line.strip():
					page = Path(line.strip())
					if page.exists():
						yield page
					else:
						print("skipping %s: path does not exist." % page)

	@staticmethod
	def _pages_below(root):
		"""Files of a directory tree in sorted order; `<page>.out` folders hold results, not pages."""
		for folder, dirs, names in os.walk(root):
			if folder.endswith(".out"):
				dirs.clear()
				continue
			dirs.sort()
			for name in sorted(names):
				yield Path(folder) / name

	def _build_queue(self, path):
		"""[(index, page path, readers / writers)] of the pages that are ready for this stage
		(ref: processor.py:425-474: same walk order, filters and messages)."""
		root = Path(path)
		if not root.exists():
			raise FileNotFoundError("%s does not exist." % root)
		tree = root.is_dir()
		if not tree and root.suffix != ".txt":
			raise FileNotFoundError("%s is not a valid path or text file of paths." % root)
		if tree:
			print(f"scanning {root}... ", flush=True, end="")
		n_images, queued = 0, []
		for page in (self._pages_below(root) if tree else self._listed_pages(root)):
			if self._name and not re.search(self._name, str(page)):
				continue
			if not self.is_image(page):
				continue
			n_images += 1
			if not self.should_process(page):
				continue
			files = self.prepare_process(page)
			if files is not False:
				queued.append((len(queued), page, files))
		if tree:
			print("done.", flush=True)
			print(f"{n_images} documents found, {len(queued)} ready to process.")
		return queued

	def _make_mutex(self, path):
		if self._lock_strategy == "FILE":
			return FileMutex()
		if self._lock_strategy == "NONE":
			return DummyMutex()
		where = Path(path)
		database = Path(self._lock_database) if self._lock_database else (
			where if where.is_dir() else where.parent) / "origami.lock.db"
		return DatabaseMutex(database, timeout=self._lock_timeout)

	def worker_devices(self, n_workers):
		"""GPU index of every worker: worker r -> devices[r % len(devices)], so with K workers per GPU the workers of a
		GPU are r, r + n, r + 2n ... ([0..7, 0..7] for 16 workers over 8 GPUs)."""
		if self._devices:
			devices = [int(d) for d in str(self._devices).split(",") if d.strip() != ""]
		else:
			from ... import _shim
			devices = list(range(max(1, _shim.device_count())))
		return [devices[r % len(devices)] for r in range(n_workers)]

	def gpu_worker_count(self):
		"""--gpus N x --workers-per-gpu K spawned workers (K = 0: numa.auto_workers_per_gpu)."""
		if self._gpus <= 0:
			return 0
		per_gpu = self._workers_per_gpu
		if per_gpu <= 0:
			from ... import numa
			per_gpu = numa.auto_workers_per_gpu(self._gpus)
		return self._gpus * per_gpu

	def _progress(self, i, p, total):
		if self._print_paths:
			nd = len(str(total))
			print(f"[{str(i + 1).rjust(nd)} / {total}] {p}", flush=True)

	def _process_queue(self, queued, data_path):
		n_workers = self.gpu_worker_count() if self._gpus > 0 else self._processes
		if n_workers <= 1:
			if self._gpus == 1:
				self._device = self.worker_devices(1)[0]
				os.environ["OGD_DEVICE"] = str(self._device)
			for chunk in chunks(queued, self._lock_chunk_size):
				for i, p in self._trigger_process(chunk):
					self._progress(i, p, len(queued))
			return

		devices = self.worker_devices(n_workers)
		if n_workers > len(set(devices)):
			# several workers per GPU share a few cores each: their threads sleep while they wait for the device instead of
			# spinning (HIP's default) on cores the other workers' host work needs (7.5 -> 6.9 ms/page at 8 workers,
			# profiles/r03/m_sync_mode.log); the spawned workers inherit the setting
			# (set for the children only: this process keeps the waiting mode it started with)
			sync_was_unset = "OGD_SYNC" not in os.environ
			os.environ.setdefault("OGD_SYNC", "block")
		# small enough chunks that every worker gets several pulls (dynamic load balance)
		chunk_size = max(1, min(self._lock_chunk_size, -(-len(queued) // (4 * n_workers))))
		ctx = multiprocessing.get_context("spawn")
		task_q, done_q = ctx.Queue(), ctx.Queue()
		for chunk in chunks([(i, str(p)) for i, p, _ in queued], chunk_size):
			task_q.put(chunk)
		for _ in range(n_workers):
			task_q.put(None)
		workers = [
			ctx.Process(
				target=_worker_main,
				args=(self.__class__, self._options, str(data_path), r, devices[r], task_q, done_q),
				daemon=True)
			for r in range(n_workers)]
		for w in workers:
			w.start()
		if n_workers > len(set(devices)) and sync_was_unset:
			del os.environ["OGD_SYNC"]
		alive, done, last = n_workers, 0, time.monotonic()
		failed = False
		while alive > 0:
			try:
				rank, i, p = done_q.get(timeout=1.0)
			except queue_mod.Empty:
				if time.monotonic() - last > self._timeout:  # watchdog, ref: processor.py:175-222
					logging.error("no progress for %d s, terminating workers." % self._timeout)
					failed = True
					break
				if not any(w.is_alive() for w in workers) and done_q.empty():
					break
				continue
			last = time.monotonic()
			if i is None:
				alive -= 1
			elif i < 0:
				failed = True
			else:
				done += 1
				self._progress(i, p, len(queued))
		for w in workers:
			if failed and w.is_alive():
				w.terminate()
			w.join(timeout=30)
		# A worker killed by a signal (GPU fault -> abort, SIGSEGV, the kernel's OOM killer) runs neither its
		# `except` nor its `finally`: it reports nothing, its chunk stays undone and its lock rows stay behind.
		for r, w in enumerate(workers):
			if w.exitcode not in (0, None):
				logging.error("worker %d (pid %s, GPU %s) died with exit code %s." % (r, w.pid, devices[r], w.exitcode))
				failed = True
				if isinstance(self._mutex, DatabaseMutex) and w.pid is not None:
					self._mutex.unlock_pid(w.pid)
		if not failed and (alive > 0 or done < len(queued)):
			logging.error("workers ended with %d of %d pages unaccounted for." % (len(queued) - done, len(queued)))
			failed = True
		if failed:
			sys.exit(1)

	def traverse(self, path):
		print(f"running {self.processor_name}.", flush=True)
		queued = self._build_queue(path)
		self._mutex = self._make_mutex(path)
		if isinstance(self._mutex, DatabaseMutex):
			self._mutex.clear_locks(self._max_lock_age)
		try:
			self._process_queue(queued, path)
		finally:
			self._mutex = None

	# ---- runtime.json --------------------------------------------------------------------------------

	def _update_runtime_info(self, page_path, updates):
		"""Merges `updates` ({stage name: record}, None deletes) into the page's runtime.json; the file is replaced in
		one step so that readers never see half of it (ref: processor.py:349-379).  Failures are logged, not raised:
		bookkeeping must not fail a page that was processed."""
		try:
			target = find_data_path(page_path) / Artifact.RUNTIME.filename()
			merged = json.loads(target.read_text()) if target.exists() else {}
			for key, value in updates.items():
				if value is None:
					merged.pop(key)
				else:
					merged[key] = value
			scratch = target.with_name(target.stem + ".updated" + target.suffix)
			scratch.write_text(json.dumps(merged))
			os.replace(scratch, target)
		except Exception:
			logging.error(traceback.format_exc())
