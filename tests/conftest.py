import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
	sys.path.insert(0, str(ROOT))


def pytest_configure(config):
	config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
	try:
		from origami_amd import _shim
		return _shim.device_count() > 0
	except Exception:
		return False


@pytest.fixture(scope="session")
def gpu():
	"""Binds the process to GPU 0; GPU tests fail (not skip) when the library cannot run."""
	from origami_amd import _shim
	_shim.init(int(os.environ.get("OGD_DEVICE", "0")))
	return _shim


@pytest.fixture(scope="session", autouse=True)
def _oracle_built():
	from oracle import build_oracle
	build_oracle.build()
