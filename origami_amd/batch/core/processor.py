"""Batch runtime of a stage: directory walk, readiness / skip rules, page locks, runtime.json, and the
fan-out of pages over worker processes — one worker process per GPU.

ref: origami/batch/core/processor.py:230-662 (Processor).  The CLI keeps the reference's option set
(processor.py:276-347); `--gpus` and `--devices` are additive.  Where the reference forks a
multiprocessing.Pool over chunks of 25 pages (processor.py:476-493), this runtime *spawns* its workers
(no HIP state is ever inherited), binds worker r to GPU devices[r % len(devices)] before its first
HIP call, and lets the workers pull chunks from one host-side queue: pages are independent, so there is
no collective and nothing crosses xGMI.
"""

import functools
import json
import logging
import multiprocessing
import os
import queue as queue_mod
import re
import sys
import time
import traceback
from pathlib import Path

import click

from .io import Artifact, AtomicFileWriter, find_data_path
from .mutex import DatabaseMutex, DummyMutex, FileMutex

IMAGE_SUFFIXES = (".jpg", ".png", ".tif", ".tiff")


def chunks(items, n):
	for i in range(0, len(items), n):
		yield items[i:i + n]


def _worker_main(processor_cls, options, data_path, rank, device, task_q, done_q):
	"""Body of one spawned worker: bind to its GPU, then pull chunks of page paths until the queue
	hands out None."""
	try:
		if device is not None:
			os.environ["OGD_DEVICE"] = str(device)
		processor = processor_cls(options)
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


class Processor:
	def __init__(self, options):
		self._options = dict(options)
		self._overwrite = options.get("overwrite", False)
		self._processes = options.get("processes", 1)
		self._timeout = options.get("alive", 600)
		self._name = options.get("name", "")
		self._verbose = False

		self._lock_strategy = (options.get("lock_strategy") or "DB").upper()
		self._lock_level = (options.get("lock_level") or "PAGE").upper()
		self._lock_timeout = options.get("lock_timeout", 60)
		self._max_lock_age = options.get("max_lock_age")
		self._lock_chunk_size = 25
		self._lock_database = options.get("lock_database")
		self._mutex = None
		if self._lock_strategy not in ("DB", "FILE", "NONE"):
			raise ValueError(self._lock_strategy)

		self._gpus = options.get("gpus", 0) or 0
		self._devices = options.get("devices") or ""
		self._rank = 0
		self._device = None

		self._plain = options.get("plain")
		self._print_paths = bool(self._plain)
		if options.get("profile") or options.get("track_changes") or options.get("debug_write"):
			logging.warning("--profile, --track-changes and --debug-write are not supported by this build.")

	@staticmethod
	def options(f):
		options = [
			click.option('--processes', type=int, default=1, help="Number of parallel processes to employ."),
			click.option('--alive', type=int, default=600, help="Seconds to wait after inactive process is killed."),
			click.option('--name', type=str, default="", help="Only process paths that conform to the given pattern."),
			click.option(
				'--lock-strategy', type=click.Choice(['FILE', 'DB', 'NONE'], case_sensitive=False), default="DB",
				help="How to implement locking for concurrency."),
			click.option(
				'--lock-level', type=click.Choice(['PAGE', 'TASK'], case_sensitive=False), default="PAGE",
				help="Lock granularity."),
			click.option(
				'--lock-database', type=click.Path(), required=False,
				help="Mutex database path used for concurrent processing"),
			click.option(
				'--lock-timeout', type=int, default=60, required=False,
				help="Seconds to wait to acquire locking. NFS volumes might need high values."),
			click.option(
				'--max-lock-age', type=int, default=600, required=False,
				help="Maximum age of a lock in seconds until it is considered invalid."),
			click.option('--overwrite', is_flag=True, default=False, help="Recompute and overwrite existing result files."),
			click.option('--profile', is_flag=True, default=False, help="Enable profiling and show results."),
			click.option('--plain', is_flag=True, default=False, help="Print plain output that is friendly to piping."),
			click.option('--debug-write', is_flag=True, default=False, help="Debug which files are written."),
			click.option('--track-changes', type=str, default="", help="Recompute files and track changes with given tag."),
			# additive (MI355X build)
			click.option('--gpus', type=int, default=0, help="Worker processes, one per GPU (0: work in this process)."),
			click.option('--devices', type=str, default="", help="Comma separated GPU indices to use (default: all visible)."),
		]
		return functools.reduce(lambda x, opt: opt(x), options, f)

	@property
	def processor_name(self):
		return self.__class__.__name__

	def is_image(self, path):
		if path.suffix.lower() in IMAGE_SUFFIXES:
			return True
		try:
			import PIL.Image
			with PIL.Image.open(str(path)) as im:
				im.verify()
			return True
		except Exception:
			return False

	def should_process(self, page_path):
		return True

	def artifacts(self):
		raise NotImplementedError()

	def process(self, page_path, **kwargs):
		raise NotImplementedError()

	def lock_or_open(self, path, mode):
		return open(path, mode)

	def prepare_process(self, page_path):
		file_writer = AtomicFileWriter(overwrite=self._overwrite)
		kwargs = dict()
		for arg, spec in self.artifacts():
			f = spec.instantiate(page_path=page_path, processor=self, file_writer=file_writer)
			f.fix_inconsistent()
			if not f.is_ready():
				if self._verbose:
					print("skipping %s: missing %s" % (page_path, f.missing))
				return False
			kwargs[arg] = f
		return kwargs

	def _trigger_process1(self, p, kwargs, locked):
		work = locked
		if not locked:
			logging.warning("failed to obtain lock for %s. ignoring." % p)
		try:
			if work:
				# a concurrent worker might already have done this.
				work = all(f.is_ready() for f in kwargs.values())
			if work:
				t0 = time.perf_counter()
				find_data_path(p).mkdir(exist_ok=True)
				runtime_info = self.process(p, **kwargs)
				if runtime_info is None:
					runtime_info = dict()
				runtime_info["status"] = "COMPLETED"
				runtime_info["elapsed"] = round(time.perf_counter() - t0, 2)
				if self._device is not None:
					runtime_info["device"] = self._device
				self._update_runtime_info(p, {self.processor_name: runtime_info})
		except KeyboardInterrupt:
			logging.exception("Interrupted at %s." % p)
			raise
		except Exception:
			logging.exception("Failed to process %s." % p)
			self._update_runtime_info(p, {
				self.processor_name: dict(status="FAILED", traceback=traceback.format_exc())})
		finally:
			kwargs.clear()

	def _trigger_process(self, chunk):
		if not chunk:
			return
		if self._lock_level == "PAGE":
			actor = "page"
		elif self._lock_level == "TASK":
			actor = self.processor_name
		else:
			raise ValueError(self._lock_level)
		if isinstance(self._mutex, FileMutex):
			groups = [[item] for item in chunk]
		else:
			groups = [chunk]
		for group in groups:
			with self._mutex.lock(actor, [str(p) for _, p, _ in group]) as locked:
				for i, p, kwargs in group:
					self._trigger_process1(p, kwargs, locked)
					yield i, p

	# ---- queue ---------------------------------------------------------------------------------------

	def _build_queue(self, path):
		path = Path(path)
		if not path.exists():
			raise FileNotFoundError("%s does not exist." % path)
		queued = []
		counts = dict(images=0)

		def add_path(p):
			if not p.exists():
				print("skipping %s: path does not exist." % p)
				return
			if self._name and not re.search(self._name, str(p)):
				return
			if not self.is_image(p):
				return
			counts['images'] += 1
			if not self.should_process(p):
				return
			kwargs = self.prepare_process(p)
			if kwargs is not False:
				queued.append((len(queued), p, kwargs))

		if not path.is_dir():
			if path.suffix == ".txt":
				with open(path, "r") as f:
					for line in f:
						line = line.strip()
						if line:
							add_path(Path(line))
			else:
				raise FileNotFoundError("%s is not a valid path or text file of paths." % path)
		else:
			print(f"scanning {path}... ", flush=True, end="")
			for folder, dirs, filenames in os.walk(path):
				folder = Path(folder)
				if folder.name.endswith(".out"):
					dirs.clear()
					continue
				dirs.sort()
				for filename in sorted(filenames):
					add_path(folder / filename)
			print("done.", flush=True)
			print(f"{counts['images']} documents found, {len(queued)} ready to process.")
		return queued

	def _make_mutex(self, path):
		if self._lock_strategy == "DB":
			if self._lock_database:
				db_path = Path(self._lock_database)
			elif Path(path).is_dir():
				db_path = Path(path) / "origami.lock.db"
			else:
				db_path = Path(path).parent / "origami.lock.db"
			return DatabaseMutex(db_path, timeout=self._lock_timeout)
		elif self._lock_strategy == "FILE":
			return FileMutex()
		return DummyMutex()

	def worker_devices(self, n_workers):
		"""GPU index of every worker: worker r -> devices[r % len(devices)]."""
		if self._devices:
			devices = [int(d) for d in str(self._devices).split(",") if d.strip() != ""]
		else:
			from ... import _shim
			devices = list(range(max(1, _shim.device_count())))
		return [devices[r % len(devices)] for r in range(n_workers)]

	def _progress(self, i, p, total):
		if self._print_paths:
			nd = len(str(total))
			print(f"[{str(i + 1).rjust(nd)} / {total}] {p}", flush=True)

	def _process_queue(self, queued, data_path):
		n_workers = self._gpus if self._gpus > 0 else self._processes
		if n_workers <= 1:
			if self._gpus == 1:
				self._device = self.worker_devices(1)[0]
				os.environ["OGD_DEVICE"] = str(self._device)
			for chunk in chunks(queued, self._lock_chunk_size):
				for i, p in self._trigger_process(chunk):
					self._progress(i, p, len(queued))
			return

		devices = self.worker_devices(n_workers)
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

	def _update_json(self, page_path, artifact, updates):
		try:
			json_path = find_data_path(page_path) / artifact.filename()
			new_json_path = json_path.parent / (json_path.stem + ".updated" + json_path.suffix)
			if new_json_path.exists():
				os.remove(new_json_path)
			data = dict()
			if json_path.exists():
				with open(json_path, "r") as f:
					data = json.loads(f.read())
			for k, v in updates.items():
				if v is None:
					del data[k]
				else:
					data[k] = v
			with open(new_json_path, "w") as f:
				json.dump(data, f)
			os.replace(new_json_path, json_path)
		except Exception:
			logging.error(traceback.format_exc())

	def _update_runtime_info(self, page_path, updates):
		self._update_json(page_path, Artifact.RUNTIME, updates)
