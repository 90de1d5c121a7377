"""Page / task locks for several workers (or machines) on one data folder.

ref: origami/batch/core/mutex.py:24-175.  Same strategies and the same `mutex(path, processor, pid,
time)` table; sqlite3 from the standard library replaces sqlalchemy, fcntl replaces portalocker.
"""

import datetime
import fcntl
import os
import sqlite3
import time
from contextlib import contextmanager
from pathlib import Path


def run_db_operation(operation):
	backoff = 0
	while True:
		try:
			return operation()
		except sqlite3.OperationalError:
			if backoff > 8:
				raise
			time.sleep(0.1 * (2 ** backoff))
			backoff += 1


class DatabaseMutex:
	def __init__(self, path, timeout=1):
		self._path = str(Path(path))
		self._timeout = float(timeout)

		def create():
			with self._connect() as conn:
				conn.execute("BEGIN EXCLUSIVE")
				conn.execute(
					"CREATE TABLE IF NOT EXISTS mutex (path TEXT NOT NULL, processor TEXT NOT NULL, "
					"pid BIGINT NOT NULL, time DATETIME NOT NULL, CONSTRAINT mutex_pk PRIMARY KEY (path, processor))")
				conn.execute("COMMIT")
		run_db_operation(create)

	@contextmanager
	def _connect(self):
		# isolation_level None: no implicit BEGIN, every operation opens its own BEGIN EXCLUSIVE
		conn = sqlite3.connect(self._path, timeout=self._timeout, isolation_level=None)
		try:
			yield conn
		finally:
			conn.close()

	def clear_locks(self, age=0):
		def perform():
			with self._connect() as conn:
				conn.execute("BEGIN EXCLUSIVE")
				if not age:
					conn.execute("DELETE FROM mutex")
				else:
					limit = datetime.datetime.now() - datetime.timedelta(seconds=age)
					conn.execute("DELETE FROM mutex WHERE time < ?", (limit.isoformat(" "),))
				conn.execute("COMMIT")
		run_db_operation(perform)

	def try_lock(self, processor, paths):
		def perform():
			with self._connect() as conn:
				conn.execute("BEGIN EXCLUSIVE")
				try:
					now = datetime.datetime.now().isoformat(" ")
					conn.executemany(
						"INSERT INTO mutex (path, processor, pid, time) VALUES (?, ?, ?, ?)",
						[(p, processor, os.getpid(), now) for p in paths])
					conn.execute("COMMIT")
					return True
				except sqlite3.IntegrityError:
					conn.execute("ROLLBACK")
					return False
		return run_db_operation(perform)

	def unlock(self, processor, paths):
		def perform():
			with self._connect() as conn:
				conn.execute("BEGIN EXCLUSIVE")
				conn.executemany(
					"DELETE FROM mutex WHERE processor = ? AND path = ? AND pid = ?",
					[(processor, p, os.getpid()) for p in paths])
				conn.execute("COMMIT")
		run_db_operation(perform)

	def unlock_pid(self, pid):
		"""Drops every lock a (dead) process of THIS machine still holds; the stage's parent calls it for a
		worker that was killed before it could unlock."""
		def perform():
			with self._connect() as conn:
				conn.execute("BEGIN EXCLUSIVE")
				conn.execute("DELETE FROM mutex WHERE pid = ?", (int(pid),))
				conn.execute("COMMIT")
		run_db_operation(perform)

	def held(self):
		"""[(path, processor, pid)] of all current locks (diagnostics, tests)."""
		with self._connect() as conn:
			return list(conn.execute("SELECT path, processor, pid FROM mutex ORDER BY path"))

	@contextmanager
	def lock(self, processor, paths):
		success = self.try_lock(processor, paths)
		try:
			yield success
		finally:
			if success:
				self.unlock(processor, paths)


class FileMutex:
	@contextmanager
	def lock(self, processor, paths):
		if len(paths) != 1:
			raise RuntimeError("FileMutex does not support chunked locking")
		with open(paths[0], "r") as f:
			try:
				fcntl.flock(f, fcntl.LOCK_EX | fcntl.LOCK_NB)
			except OSError:
				yield False
				return
			try:
				yield True
			finally:
				fcntl.flock(f, fcntl.LOCK_UN)


class DummyMutex:
	def try_lock(self, processor, paths):
		return True

	def unlock(self, processor, paths):
		pass

	@contextmanager
	def lock(self, processor, paths):
		yield True
