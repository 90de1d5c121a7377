"""origami_amd.page.ResidentPages (the LRU of pages kept on the device across stages of one process) and the
plumbing of `origami_amd.batch.chain`, without a GPU (ref: origami/batch/core/io.py:275-280: every stage of the
reference opens and remaps the page again; quickstart/02_cpu.sh:11-16: one process per stage)."""

import os

import click
import PIL.Image
import numpy as np
import pytest

from origami_amd.batch import chain
from origami_amd.page import ResidentPages


class FakePage:
	def __init__(self, nbytes):
		self.nbytes = nbytes


def test_key_follows_the_file(tmp_path):
	f = tmp_path / "p.png"
	f.write_bytes(b"x" * 10)
	k0 = ResidentPages.key(str(f))
	assert k0 == ResidentPages.key(str(tmp_path / "." / "p.png"))  # one file, one key
	f.write_bytes(b"x" * 11)
	k1 = ResidentPages.key(str(f))
	assert k1 != k0  # size
	os.utime(f, ns=(1, 1))
	assert ResidentPages.key(str(f)) != k1  # mtime
	with pytest.raises(OSError):
		ResidentPages.key(str(tmp_path / "gone.png"))


def test_lru_is_bounded_in_bytes_and_counts():
	cache = ResidentPages(100)
	a, b, c = FakePage(40), FakePage(40), FakePage(40)
	cache.put("a", a)
	cache.put("b", b)
	assert cache.get("a") is a and cache.nbytes == 80  # a is now the most recently used
	cache.put("c", c)  # 120 bytes: the least recently used page (b) goes
	assert "b" not in cache and "a" in cache and "c" in cache and cache.nbytes == 80
	assert cache.get("b") is None
	assert (cache.hits, cache.misses, cache.evictions) == (1, 1, 1)
	cache.put("a", FakePage(60))  # replacing a page accounts for its old size
	assert cache.nbytes == 100 and len(cache) == 2
	cache.put("huge", FakePage(101))  # a page larger than the budget is not kept (and evicts nothing)
	assert "huge" not in cache and len(cache) == 2
	cache.clear()
	assert cache.nbytes == 0 and len(cache) == 0


def test_budget_zero_switches_it_off():
	cache = ResidentPages(0)
	assert not cache.enabled
	cache.put("a", FakePage(1))
	assert "a" not in cache and cache.get("a") is None


def test_chain_finds_the_stage_commands_and_the_pages(tmp_path):
	for name in ("detect.dewarp", "utils.dewarp", "export.lines"):
		assert isinstance(chain.stage_command(name), click.Command)
	with pytest.raises(click.UsageError):
		chain.stage_command("detect.compose")  # a module of the path, not a stage with a command line
	with pytest.raises(ImportError):
		chain.stage_command("detect.nothing")
	(tmp_path / "b").mkdir()
	for rel in ("a.png", "b/c.jpg"):
		PIL.Image.fromarray(np.zeros((4, 4), np.uint8)).save(tmp_path / rel)
	(tmp_path / "a.out").mkdir()
	PIL.Image.fromarray(np.zeros((4, 4), np.uint8)).save(tmp_path / "a.out" / "annotation.dewarped.jpg")  # a result, not a page
	(tmp_path / "notes.txt").write_text("no image")
	pages = chain.pages_below(tmp_path)
	assert [p.relative_to(tmp_path).as_posix() for p in pages] == ["a.png", "b/c.jpg"]
	assert [p.name for p in chain.pages_below(tmp_path, name=r"c\.jpg$")] == ["c.jpg"]
