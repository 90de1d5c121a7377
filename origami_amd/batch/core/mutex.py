"""Page / task locks for several workers (or machines) on one data folder.

ref: origami/batch/core/mutex.py:24-175 — the three strategies a stage can be started with (`--lock-strategy DB |
FILE | NONE`) and, for DB, the table `mutex(path, processor, pid, time)` with its primary key (path, processor), so
that this build and the reference can work on one folder side by side.  sqlite3 from the standard library stands
in for sqlalchemy, fcntl for portalocker.
"""

import datetime
import fcntl
import os
import sqlite3
import time
from contextlib import contextmanager
from pathlib import Path

_SCHEMA = (
	"CREATE TABLE IF NOT EXISTS mutex (path TEXT NOT NULL, processor TEXT NOT NULL, "
	"pid BIGINT NOT NULL, time DATETIME NOT NULL, CONSTRAINT mutex_pk PRIMARY KEY (path, processor))")

_BUSY_RETRIES = 9  # waits of 0.1 s, 0.2 s, ... 25.6 s: a lock database on NFS can stay busy for a while


class _Lockable:
	"""lock(processor, paths) as a context manager over try_lock / unlock; yields whether the lock was obtained."""

	@contextmanager
	def lock(self, processor, paths):
		held = self.try_lock(processor, paths)
		try:
			yield held
		finally:
			if held:
				self.unlock(processor, paths)


class DatabaseMutex(_Lockable):
	"""One row per locked (path, processor); taking the locks of a chunk is one exclusive transaction, so either
	all of its pages are locked or none."""

	def __init__(self, path, timeout=1):
		self._path = str(Path(path))
		self._timeout = float(timeout)
		self._exclusive(lambda db: db.execute(_SCHEMA))

	def _exclusive(self, work):
		"""work(connection) inside BEGIN EXCLUSIVE ... COMMIT (ROLLBACK if it raises), on a connection of its own;
		a busy database is retried with exponential back-off."""
		for attempt in range(_BUSY_RETRIES + 1):
			# isolation_level None: no implicit BEGIN, the transaction below is the only one
			db = sqlite3.connect(self._path, timeout=self._timeout, isolation_level=None)
			try:
				db.execute("BEGIN EXCLUSIVE")
				try:
					result = work(db)
				except BaseException:
					db.execute("ROLLBACK")
					raise
				db.execute("COMMIT")
				return result
			except sqlite3.OperationalError:
				if attempt == _BUSY_RETRIES:
					raise
				time.sleep(0.1 * 2 ** attempt)
			finally:
				db.close()

	def clear_locks(self, age=0):
		"""Forget every lock, or with `age` those older than that many seconds (left behind by crashed runs)."""
		if not age:
			self._exclusive(lambda db: db.execute("DELETE FROM mutex"))
			return
		limit = (datetime.datetime.now() - datetime.timedelta(seconds=age)).isoformat(" ")
		self._exclusive(lambda db: db.execute("DELETE FROM mutex WHERE time < ?", (limit,)))

	def try_lock(self, processor, paths):
		rows = [(p, processor, os.getpid(), datetime.datetime.now().isoformat(" ")) for p in paths]
		try:
			self._exclusive(lambda db: db.executemany("INSERT INTO mutex (path, processor, pid, time) VALUES (?, ?, ?, ?)", rows))
		except sqlite3.IntegrityError:  # somebody holds one of them: the transaction was rolled back
			return False
		return True

	def unlock(self, processor, paths):
		rows = [(processor, p, os.getpid()) for p in paths]
		self._exclusive(lambda db: db.executemany("DELETE FROM mutex WHERE processor = ? AND path = ? AND pid = ?", rows))

	def unlock_pid(self, pid):
		"""Drops every lock a (dead) process of THIS machine still holds; the stage's parent calls it for a
		worker that was killed before it could unlock."""
		self._exclusive(lambda db: db.execute("DELETE FROM mutex WHERE pid = ?", (int(pid),)))

	def held(self):
		"""[(path, processor, pid)] of all current locks (diagnostics, tests)."""
		db = sqlite3.connect(self._path, timeout=self._timeout)
		try:
			return list(db.execute("SELECT path, processor, pid FROM mutex ORDER BY path"))
		finally:
			db.close()


class FileMutex:
	"""An advisory lock on the page file itself; one page at a time."""

	@contextmanager
	def lock(self, processor, paths):
		if len(paths) != 1:
			raise RuntimeError("FileMutex does not support chunked locking")
		with open(paths[0], "r") as f:
			try:
				fcntl.flock(f, fcntl.LOCK_EX | fcntl.LOCK_NB)
				held = True
			except OSError:
				held = False
			try:
				yield held
			finally:
				if held:
					fcntl.flock(f, fcntl.LOCK_UN)


class DummyMutex(_Lockable):
	"""--lock-strategy NONE: everybody gets every lock."""

	def try_lock(self, processor, paths):
		return True

	def unlock(self, processor, paths):
		pass
