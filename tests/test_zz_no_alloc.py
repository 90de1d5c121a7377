"""ogd_march_grid_noalloc (the entry point that allocates nothing: its device buffers are a fixed array in the
library's device image) against ogd_march_grid, through the C ABI.

Bar: identical kernels on identical inputs -> the two node grids must be bit-identical.  Two calls in a row,
the second with a smaller grid, so that whatever the first call left in the shared scratch would show."""

import numpy as np
import pytest

from origami_amd import synth
from origami_amd.dewarp import Field

pytestmark = pytest.mark.gpu


def _fields(rng, size, n_h, n_v):
	warp = synth.Warp(size, 0.03, 0.025, 1, 2)
	h, v = synth.flow_samples(rng, warp, n_h, n_v)
	return Field(h, size, 0.0), Field(v, size, np.pi / 2)


def _both(gpu, size, cell, slices, n_h, n_v, seed):
	rng = np.random.default_rng(seed)
	fh, fv = _fields(rng, size, n_h, n_v)
	bounds = (0, 0, size[0], size[1])
	n_rows, n_cols = gpu.march_shape(fh.device, fv.device, bounds, size[0], size[1], cell)
	want = gpu.march_grid(fh.device, fv.device, bounds, cell, n_rows, n_cols, slices)
	got = gpu.march_grid(fh.device, fv.device, bounds, cell, n_rows, n_cols, slices, noalloc=True)
	return (n_rows, n_cols), want, got


def test_noalloc_march_equals_march_twice_in_a_row(gpu):
	shape1, want1, got1 = _both(gpu, (1500, 2000), 25, 2, 300, 400, 1)
	shape2, want2, got2 = _both(gpu, (600, 500), 25, 3, 120, 150, 2)  # smaller: the scratch still holds call 1
	assert shape2[0] * shape2[1] < shape1[0] * shape1[1]
	for want, got in ((want1, got1), (want2, got2)):
		for w, g in zip(want, got):
			assert g.shape == w.shape and g.dtype == np.float32
			assert np.array_equal(g, w)
			assert np.isfinite(g).all()


def test_noalloc_march_takes_no_device_memory(gpu):
	"""hipMemGetInfo before and after: the call must not shrink the GPU's free memory (its buffers are part of the
	library's device image, loaded with the first launch), while ogd_march_grid on a grid larger than anything its
	block cache holds does."""
	import ctypes
	lib = gpu.load()

	def free_bytes():
		f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
		gpu.check(lib.ogd_mem_info(ctypes.byref(f), ctypes.byref(t)))
		return f.value

	args = (0.0, np.pi / 2, (0, 0, 500, 500), 1, 500, 500, 2)
	gpu.march_grid(*args, noalloc=True)  # module, scratch address and kernels are loaded now
	gpu.check(lib.ogd_trim(None))
	before = free_bytes()
	for _ in range(3):
		gpu.march_grid(*args, noalloc=True)
		assert free_bytes() >= before
	gpu.march_grid(*args)  # the allocating form takes 2 x 2 MB for the same grids
	assert free_bytes() < before


def test_noalloc_march_rejects_grids_beyond_its_scratch(gpu):
	with pytest.raises(gpu.OgdError, match="exceed the fixed scratch"):
		gpu.march_grid(0.0, np.pi / 2, (0, 0, 100, 100), 1, 600, 600, 2, noalloc=True)
	# and the limit itself is accepted (constant fields, 512 x 512 nodes)
	gh, ghv = gpu.march_grid(0.0, np.pi / 2, (0, 0, 512, 512), 1, 512, 512, 2, noalloc=True)
	want_h, want_hv = gpu.march_grid(0.0, np.pi / 2, (0, 0, 512, 512), 1, 512, 512, 2)
	assert np.array_equal(gh, want_h) and np.array_equal(ghv, want_hv)
