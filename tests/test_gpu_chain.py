"""Stages chained in one process share the page they load (origami_amd.batch.chain, origami_amd.page.resident):
`utils.dewarp` then `export.lines` over the same pages decode and upload every page once, and write the files they
write when each runs alone with the cache switched off (ref: origami/batch/core/io.py:275-280 — in the reference
every stage opens and remaps the page again)."""

import shutil
import zipfile

import numpy as np
import PIL.Image
import pytest

from origami_amd import page as page_mod
from origami_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def gpu():
	from origami_amd import _shim
	_shim.init(0)
	return _shim


def _make_pages(root, suffix, n=3):
	rng = np.random.default_rng(5)
	s, h, w = 25, 30, 24
	size = (w * s - 20, h * s - 30)
	gy, gx = np.mgrid[0:h, 0:w].astype(np.float64)
	for k in range(n):
		nodes = np.stack([
			(gx - 0.5) * s + 4 * np.sin(gy / (4.0 + k)) + rng.normal(0, 0.3, size=(h, w)),
			(gy - 0.5) * s + 4 * np.cos(gx / 5.0) + rng.normal(0, 0.3, size=(h, w))], axis=-1).astype(np.float32)
		path = root / ("page%04d%s" % (k, suffix))
		im = PIL.Image.fromarray(synth.text_page(rng, size))
		im.save(path, quality=90) if suffix == ".jpg" else im.save(path)
		synth.write_reliable_stage(path, rng, nodes, s, n_lines=12, n_table_lines=2)


def _results(root):
	out = {}
	for d in sorted(root.glob("*.out")):
		out[d.name + "/annotation.dewarped.jpg"] = (d / "annotation.dewarped.jpg").read_bytes()
		with zipfile.ZipFile(d / "images.lines.dewarped.zip") as zf:
			for name in zf.namelist():
				out[d.name + "/lines/" + name] = zf.read(name)
	return out


@pytest.mark.parametrize("suffix", [".png", ".jpg"])
def test_chained_stages_load_a_page_once_and_write_the_same_files(gpu, tmp_path, monkeypatch, suffix):
	from click.testing import CliRunner
	from origami_amd.batch import chain
	from origami_amd.batch.export.lines import export_lines
	from origami_amd.batch.utils.dewarp import dewarp as dewarp_image_stage

	alone, chained = tmp_path / "alone", tmp_path / "chained"
	alone.mkdir()
	_make_pages(alone, suffix)
	shutil.copytree(alone, chained)

	# every stage alone, nothing kept between them
	monkeypatch.setattr(page_mod, "resident", page_mod.ResidentPages(0))
	for command in (dewarp_image_stage, export_lines):
		res = CliRunner().invoke(command, [str(alone), "--lock-strategy", "NONE"])
		assert res.exit_code == 0, res.output
	want = _results(alone)
	assert len(want) > 3 * 10

	cache = page_mod.ResidentPages(256 << 20)
	monkeypatch.setattr(page_mod, "resident", cache)
	n = chain.run_chain(str(chained), [("utils.dewarp", ["--lock-strategy", "NONE"]), ("export.lines", ["--lock-strategy", "NONE"])],
		pages_per_round=2, echo=lambda *a: None)
	assert n == 3
	got = _results(chained)
	assert got.keys() == want.keys()
	for name in want:
		assert got[name] == want[name], name
	# three pages went to the device once each (utils.dewarp), export.lines found all three there
	assert (cache.misses, cache.hits) == (3, 3) and len(cache) == 3

	# a page whose file changes is another page
	first = chained / ("page0000" + suffix)
	im = PIL.Image.open(first)
	im = PIL.Image.fromarray(255 - np.array(im))
	im.save(first, quality=90) if suffix == ".jpg" else im.save(first)
	chain.run_chain(str(chained), [("utils.dewarp", ["--lock-strategy", "NONE", "--overwrite"])], pages_per_round=8)
	assert cache.misses == 4 and cache.hits == 5
	assert (chained / "page0000.out" / "annotation.dewarped.jpg").read_bytes() != want["page0000.out/annotation.dewarped.jpg"]
