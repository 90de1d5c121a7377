"""Batch runtime on CPU: queue building, readiness / skip / overwrite rules, locks, runtime.json and the
spawned one-worker-per-GPU fan-out (with a stage that does no GPU work).

ref: origami/batch/core/processor.py:368-394, 517-577; io.py:231-236, 261-265, 533-543."""

import json
import zipfile

import numpy as np
import pytest

from origami_amd import synth
from origami_amd.batch.core.io import Artifact, AtomicFileWriter, Stage, find_data_path
from origami_amd.batch.core.mutex import DatabaseMutex, DummyMutex, FileMutex
from helpers.fake_stage import FakeDewarpProcessor


def make_pages(root, n, prefix="page"):
	rng = np.random.default_rng(7)
	paths = []
	for i in range(n):
		paths.append(synth.write_page_dir(
			root, "%s%04d" % (prefix, i), rng, size=(120, 160), n_h=8, n_v=8, n_regions=3, n_separators=1,
			label_size=(60, 80)))
	return paths


def options(**kw):
	base = dict(lock_strategy="NONE", processes=1, alive=60, overwrite=False, plain=True)
	base.update(kw)
	return base


def test_artifact_names():
	assert Artifact.CONTOURS.filename(Stage.WARPED) == "contours.0.zip"
	assert Artifact.CONTOURS.filename(Stage.DEWARPED) == "contours.1.zip"
	assert Artifact.DEWARPING_TRANSFORM.filename() == "dewarp.zip"
	assert Artifact.FLOW.filename() == "flow.zip" and Artifact.RUNTIME.filename() == "runtime.json"
	with pytest.raises(RuntimeError):
		Artifact.CONTOURS.filename()
	assert find_data_path("/a/b/page.png").name == "page.out"
	assert Stage.DEWARPED.is_dewarped and not Stage.WARPED.is_dewarped


def test_queue_readiness_skip_and_overwrite(tmp_path, capsys):
	pages = make_pages(tmp_path, 3)
	(find_data_path(pages[2]) / "segment.zip").unlink()  # CONTOURS implies SEGMENTATION: not ready
	(tmp_path / "notes.txt").write_text("not an image")
	proc = FakeDewarpProcessor(options())
	queued = proc._build_queue(tmp_path)
	assert [p.name for _, p, _ in queued] == ["page0000.png", "page0001.png"]
	proc.traverse(tmp_path)
	for p in pages[:2]:
		out = find_data_path(p)
		assert (out / "dewarp.zip").exists() and (out / "contours.1.zip").exists()
		info = json.loads((out / "runtime.json").read_text())["fake.dewarp"]
		assert info["status"] == "COMPLETED" and "elapsed" in info
		with zipfile.ZipFile(out / "contours.1.zip") as zf:  # meta copied first, io.py:569-575
			assert zf.namelist()[:2] == ["meta.json", "separators/V/meta.json"]
	assert not (find_data_path(pages[2]) / "dewarp.zip").exists()
	# outputs exist: nothing is queued again unless --overwrite
	assert FakeDewarpProcessor(options())._build_queue(tmp_path) == []
	assert len(FakeDewarpProcessor(options(overwrite=True))._build_queue(tmp_path)) == 2
	# a half-written output set is removed and the page redone (io.py:533-540)
	(find_data_path(pages[0]) / "dewarp.zip").unlink()
	again = FakeDewarpProcessor(options())._build_queue(tmp_path)
	assert [p.name for _, p, _ in again] == ["page0000.png"]
	assert not (find_data_path(pages[0]) / "contours.1.zip").exists()
	# --name filters paths, a .txt file lists pages
	assert len(FakeDewarpProcessor(options(name="0001", overwrite=True))._build_queue(tmp_path)) == 1
	listing = tmp_path / "pages.txt"
	listing.write_text("%s\n\n%s\n" % (pages[0], tmp_path / "missing.png"))
	assert len(FakeDewarpProcessor(options(overwrite=True))._build_queue(listing)) == 1


def test_failure_is_recorded_not_fatal(tmp_path):
	make_pages(tmp_path, 1, prefix="bad")
	make_pages(tmp_path, 1, prefix="good")
	FakeDewarpProcessor(options()).traverse(tmp_path)
	bad = json.loads((tmp_path / "bad0000.out" / "runtime.json").read_text())["fake.dewarp"]
	assert bad["status"] == "FAILED" and "synthetic failure" in bad["traceback"]
	assert not (tmp_path / "bad0000.out" / "dewarp.zip").exists()  # atomic writer leaves nothing behind
	assert [p.name for p in (tmp_path / "bad0000.out").glob(".tmp*")] == []
	good = json.loads((tmp_path / "good0000.out" / "runtime.json").read_text())["fake.dewarp"]
	assert good["status"] == "COMPLETED"


def test_atomic_writer(tmp_path):
	target = tmp_path / "x.bin"
	with AtomicFileWriter(overwrite=False)(target, "wb") as f:
		f.write(b"one")
	assert target.read_bytes() == b"one"
	with pytest.raises(FileExistsError):
		with AtomicFileWriter(overwrite=False)(target, "wb") as f:
			f.write(b"two")
	with AtomicFileWriter(overwrite=True)(target, "wb") as f:
		f.write(b"three")
	assert target.read_bytes() == b"three"
	assert sorted(p.name for p in tmp_path.iterdir()) == ["x.bin"]


def test_mutexes(tmp_path):
	db = DatabaseMutex(tmp_path / "origami.lock.db", timeout=5)
	with db.lock("page", ["/a", "/b"]) as locked:
		assert locked
		assert not db.try_lock("page", ["/b", "/c"])  # all or nothing
		assert db.try_lock("other", ["/b"])  # per processor
	assert db.try_lock("page", ["/a"])  # released
	db.clear_locks(0)
	assert db.try_lock("page", ["/a"])
	f = tmp_path / "page.png"
	f.write_bytes(b"x")
	fm = FileMutex()
	with fm.lock("page", [str(f)]) as a:
		assert a
		with fm.lock("page", [str(f)]) as b:
			assert not b
	with pytest.raises(RuntimeError):
		with fm.lock("page", ["/a", "/b"]):
			pass
	with DummyMutex().lock("page", ["/a"]) as ok:
		assert ok


@pytest.mark.parametrize("lock", ["NONE", "DB"])
def test_two_gpu_workers_share_one_queue(tmp_path, lock, monkeypatch):
	"""--gpus 2: two spawned workers, bound to GPUs 0 and 1 before any HIP call, pull chunks from one
	queue; every page is done exactly once (weak scaling, no data-path collective)."""
	pages = make_pages(tmp_path, 9)
	rdv = tmp_path.parent / (tmp_path.name + ".rdv")
	rdv.mkdir()
	monkeypatch.setenv("FAKE_STAGE_RENDEZVOUS", "%s:2" % rdv)  # no worker finishes before both have started
	proc = FakeDewarpProcessor(options(gpus=2, workers_per_gpu=1, devices="0,1", lock_strategy=lock))
	assert proc.worker_devices(4) == [0, 1, 0, 1]
	proc.traverse(tmp_path)
	seen = {}
	for p in pages:
		out = find_data_path(p)
		with zipfile.ZipFile(out / "contours.1.zip") as zf:
			who = json.loads(zf.read("who.json"))
		info = json.loads((out / "runtime.json").read_text())["fake.dewarp"]
		assert info["status"] == "COMPLETED" and info["device"] == int(who["device"])
		assert who["device"] == str(who["rank"])  # worker r -> device r
		seen.setdefault(who["rank"], set()).add(who["pid"])
	assert set(seen) == {0, 1}  # both workers took part (9 pages -> chunks of 2)
	assert all(len(pids) == 1 for pids in seen.values())
	assert FakeDewarpProcessor(options())._build_queue(tmp_path) == []


def test_a_worker_killed_by_a_signal_fails_the_stage_and_frees_its_locks(tmp_path):
	"""A spawned worker that dies without unwinding (SIGKILL here; a GPU fault's abort, SIGSEGV or the OOM killer
	in the field) sends no message: the stage must still end non-zero, and the lock rows the dead worker held
	must be gone so that a re-run can take its pages at once."""
	make_pages(tmp_path, 2, prefix="kill")
	make_pages(tmp_path, 4, prefix="page")
	proc = FakeDewarpProcessor(options(gpus=2, workers_per_gpu=1, devices="0,1", lock_strategy="DB", alive=30))
	with pytest.raises(SystemExit) as e:
		proc.traverse(tmp_path)
	assert e.value.code == 1
	assert not (tmp_path / "kill0000.out" / "dewarp.zip").exists()
	assert DatabaseMutex(tmp_path / "origami.lock.db", timeout=5).held() == []


def test_workers_per_gpu_are_dealt_round_robin_over_the_gpus():
	"""--gpus 8 --workers-per-gpu 2: sixteen workers, worker r on GPU r % 8 ([0..7, 0..7]); without the option the
	count comes from the usable cores (half of them, at most 8 per GPU, at least 1)."""
	from origami_amd import numa
	proc = FakeDewarpProcessor(options(gpus=8, workers_per_gpu=2, devices="0,1,2,3,4,5,6,7"))
	assert proc.gpu_worker_count() == 16
	assert proc.worker_devices(proc.gpu_worker_count()) == list(range(8)) * 2
	assert FakeDewarpProcessor(options(gpus=0)).gpu_worker_count() == 0
	assert numa.auto_workers_per_gpu(8, cpus=256) == 8  # capped
	assert numa.auto_workers_per_gpu(8, cpus=96) == 6
	assert numa.auto_workers_per_gpu(1, cpus=8) == 4
	assert numa.auto_workers_per_gpu(8, cpus=4) == 1
	auto = FakeDewarpProcessor(options(gpus=2, devices="0,1"))
	assert auto.gpu_worker_count() == 2 * numa.auto_workers_per_gpu(2)


def _fake_sysfs(root):
	"""Two NUMA nodes of 8 cores; GPUs c1 on node 0, e1 on node 1, a device whose node is unknown (-1)."""
	for node, cpulist in ((0, "0-3,8-11"), (1, "4-7,12-15")):
		d = root / "devices" / "system" / "node" / ("node%d" % node)
		d.mkdir(parents=True)
		(d / "cpulist").write_text(cpulist + "\n")
	for bus, node in (("0000:c1:00.0", 0), ("0000:e1:00.0", 1), ("0000:05:00.0", -1)):
		d = root / "bus" / "pci" / "devices" / bus
		d.mkdir(parents=True)
		(d / "numa_node").write_text("%d\n" % node)


def test_numa_helper_reads_a_sysfs_tree(tmp_path):
	from origami_amd import numa
	_fake_sysfs(tmp_path)
	assert numa.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
	assert numa.node_of_pci_device("0000:E1:00.0", tmp_path) == 1
	assert numa.node_of_pci_device("0000:05:00.0", tmp_path) is None
	assert numa.node_of_pci_device("0000:99:00.0", tmp_path) is None
	assert numa.cpus_of_node(1, tmp_path) == {4, 5, 6, 7, 12, 13, 14, 15}
	bound = []
	every = set(range(16))
	node = numa.bind_to_gpu_numa_node(
		1, sysfs=tmp_path, bus_id="0000:e1:00.0", setaffinity=bound.append, getaffinity=lambda: every)
	assert node == 1 and bound == [{4, 5, 6, 7, 12, 13, 14, 15}]
	# a container that only grants some of the node's cores: the intersection
	bound.clear()
	assert numa.bind_to_gpu_numa_node(
		0, sysfs=tmp_path, bus_id="0000:c1:00.0", setaffinity=bound.append, getaffinity=lambda: {2, 3, 4}) == 0
	assert bound == [{2, 3}]


def test_a_worker_that_cannot_bind_still_runs(tmp_path):
	"""No NUMA information, none of the node's cores granted, no such GPU, a failing sched_setaffinity: the helper
	answers None and leaves the affinity alone; the spawned workers of a stage (which call it with device indices
	that do not exist in this container) do their pages all the same."""
	from origami_amd import numa
	sysfs = tmp_path / "sys"
	_fake_sysfs(sysfs)
	bound = []
	every = set(range(16))
	assert numa.bind_to_gpu_numa_node(0, sysfs=sysfs, bus_id="0000:05:00.0", setaffinity=bound.append, getaffinity=lambda: every) is None
	assert numa.bind_to_gpu_numa_node(0, sysfs=sysfs, bus_id="0000:c1:00.0", setaffinity=bound.append, getaffinity=lambda: {4, 5}) is None
	assert numa.bind_to_gpu_numa_node(0, sysfs=tmp_path / "none", bus_id="0000:c1:00.0", setaffinity=bound.append, getaffinity=lambda: every) is None

	def refuse(cpus):
		raise OSError("not permitted")
	assert numa.bind_to_gpu_numa_node(0, sysfs=sysfs, bus_id="0000:c1:00.0", setaffinity=refuse, getaffinity=lambda: every) is None
	assert bound == []
	assert numa.bind_to_gpu_numa_node(5) is None  # no GPU 5 here (no GPU at all in the build container)
	pages = make_pages(tmp_path, 4)
	proc = FakeDewarpProcessor(options(gpus=2, workers_per_gpu=2, devices="0,1"))
	proc.traverse(tmp_path)
	for p in pages:
		info = json.loads((find_data_path(p) / "runtime.json").read_text())["fake.dewarp"]
		assert info["status"] == "COMPLETED" and "numa_node" not in info
