"""The C-ABI library loads and exports every symbol include/ogdewarp.h declares; the ctypes table
mirrors the header one to one; nothing computes without a GPU (no CPU fallback on the product path)."""

import re
from pathlib import Path

import numpy as np
import pytest

from origami_amd import _shim

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "ogdewarp.h"


def header_functions():
	text = HEADER.read_text()
	text = re.sub(r"/\*.*?\*/", "", text, flags=re.DOTALL)
	return set(re.findall(r"\b(ogd_[a-z0-9_]+)\s*\(", text))


def test_header_and_ctypes_table_agree():
	assert header_functions() == set(_shim.SIGNATURES)


def test_library_exports_every_declared_symbol():
	lib = _shim.load()
	for name in header_functions():
		assert getattr(lib, name) is not None


def test_header_cites_the_reference_for_every_compute_entry_point():
	text = HEADER.read_text()
	for ref in ("dewarp.py:522-523", "dewarp.py:612-614", "lingrid.py:286-290", "dewarp.py:561-576", "batch/detect/dewarp.py:26"):
		assert ref in text, ref


def test_plumbing_calls_work_without_a_gpu():
	lib = _shim.load()
	assert lib.ogd_version() >= 100
	assert isinstance(_shim.device_count(), int)
	assert lib.ogd_last_error() is not None


def test_argument_errors_are_reported_not_fatal():
	lib = _shim.load()
	# argument validation precedes any device work: bad arguments come back as OGD_ERR_INVALID
	rc = lib.ogd_remap_u8_dev(None, 4, 4, 2, 4, None, 4, 4, None, 4, 1, None)
	assert rc == -1
	assert b"channels" in lib.ogd_last_error()
	rc = lib.ogd_grid_fill_dev(None, 0, 3, 25, 1.0, 1.0, None, None)
	assert rc == -1
	with pytest.raises(_shim.OgdError) as e:
		_shim.check(rc)
	assert e.value.status == -1


@pytest.mark.skipif(_shim.device_count() > 0, reason="checks the no-GPU behaviour")
def test_compute_fails_loudly_without_a_gpu():
	nodes = np.zeros((3, 3, 2), dtype=np.float32)
	with pytest.raises(_shim.OgdError):
		_shim.grid_fill(nodes, 5)
	with pytest.raises(_shim.OgdError):
		_shim.remap_u8(np.zeros((4, 4), dtype=np.uint8), np.zeros((4, 4, 2), dtype=np.float32))


def test_product_never_imports_the_oracle():
	for path in (ROOT / "origami_amd").rglob("*.py"):
		text = path.read_text()
		assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.MULTILINE), path
	for path in (ROOT / "origami_amd" / "csrc").glob("*"):
		assert "oracle" not in path.read_text(), path
