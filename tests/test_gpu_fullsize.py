"""Parity at BASELINE.json's full sizes (configs[1], [3], [4]) on the GPU: whole broadsheet pages against
the CPU oracle (bit-exact), plus size-independent properties (identity map, tiling invariance, linearity of
the transformer on affine grids)."""

import ctypes

import numpy as np
import pytest

from oracle import cv as ocv
from oracle import dewarp as odw
from origami_amd import synth
from origami_amd.dewarp import Grid

pytestmark = pytest.mark.gpu

PAGE = (6000, 8000)


def test_config2_broadsheet_gray_page_bit_exact(gpu):
	rng = np.random.default_rng(2000)
	page = synth.text_page(rng, PAGE)
	nodes = synth.Warp.random(rng, PAGE).node_grid(25)
	dense = gpu.grid_fill(nodes, 25)
	want_dense = ocv.dense_grid(nodes, 25, threads=8)
	assert dense.shape == (8075, 6075, 2)
	np.testing.assert_array_equal(dense, want_dense)  # float grid: bit-exact (bar: 1e-3 px)
	want = ocv.remap_u8(page, want_dense, ocv.INTER_LINEAR, threads=8)
	got = gpu.remap_u8(page, dense)
	np.testing.assert_array_equal(got, want)  # bar: +-1 LSB on 99.9 %; here every pixel
	np.testing.assert_array_equal(gpu.dewarp_u8(page, nodes, 25), want)  # fused path, same bytes
	assert (got > 0).mean() > 0.9
	# the same page into rows of a padded pitch in a buffer off every alignment (K2's sheared tiles drift differently)
	lib = gpu.load()
	Hd, Wd = want.shape
	pitch, lead = 6144 + 27, 37
	out = np.full(lead + Hd * pitch + 64, 0x5A, dtype=np.uint8)
	d_src, d_map, d_out = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
	for d, n in ((d_src, page.nbytes), (d_map, dense.nbytes), (d_out, out.nbytes)):
		gpu.check(lib.ogd_dev_alloc(ctypes.byref(d), n))
	try:
		gpu.check(lib.ogd_memcpy_h2d(d_src, page.ctypes.data, page.nbytes, None))
		gpu.check(lib.ogd_memcpy_h2d(d_map, dense.ctypes.data, dense.nbytes, None))
		gpu.check(lib.ogd_memcpy_h2d(d_out, out.ctypes.data, out.nbytes, None))
		gpu.check(lib.ogd_remap_u8_dev(d_src, PAGE[1], PAGE[0], 1, PAGE[0], d_map, Hd, Wd, d_out.value + lead, pitch, 1, None))
		gpu.check(lib.ogd_memcpy_d2h(out.ctypes.data, d_out, out.nbytes, None))
		gpu.check(lib.ogd_stream_sync(None))
	finally:
		for d in (d_src, d_map, d_out):
			lib.ogd_dev_free(d)
	rows = out[lead:lead + Hd * pitch].reshape(Hd, pitch)
	np.testing.assert_array_equal(rows[:, :Wd], want)
	assert (rows[:-1, Wd:] == 0x5A).all() and (out[:lead] == 0x5A).all() and (out[lead + (Hd - 1) * pitch + Wd:] == 0x5A).all()


def test_config4_rgb_page_and_label_map_fine_grid_bit_exact(gpu):
	"""RGB page + 4-class label map through an 8-px grid of 1001 x 751 nodes (built directly: the
	reference's max_grid_size guard, dewarp.py:336-338, would refuse to march it)."""
	rng = np.random.default_rng(4000)
	warp = synth.Warp(PAGE, 0.03, 0.02, 2, 1)
	nodes = warp.node_grid(8, margin=0)[:1001, :751]
	assert nodes.shape == (1001, 751, 2)
	rgb = synth.text_page(rng, PAGE, channels=3)
	dense = ocv.dense_grid(nodes, 8, threads=8)
	want = ocv.remap_u8(rgb, dense, ocv.INTER_LINEAR, threads=8)
	np.testing.assert_array_equal(gpu.remap_u8(rgb, gpu.grid_fill(nodes, 8)), want)
	np.testing.assert_array_equal(gpu.dewarp_u8(rgb, nodes, 8), want)
	labels = synth.label_map(rng, (1280, 2400), 4, blocks=64)
	sx, sy = np.float32(1280 / 6000), np.float32(2400 / 8000)
	dense[:, :, 0] *= sx
	dense[:, :, 1] *= sy
	want_l = ocv.remap_u8(labels, dense, ocv.INTER_NEAREST, threads=8)
	got_l = gpu.dewarp_u8(labels, nodes, 8, sx, sy, ocv.INTER_NEAREST)
	np.testing.assert_array_equal(got_l, want_l)  # label maps: bit-exact
	np.testing.assert_array_equal(gpu.remap_u8(labels, gpu.grid_fill(nodes, 8, sx, sy), ocv.INTER_NEAREST), want_l)
	assert set(np.unique(got_l)) <= {0, 1, 2, 3}


def test_config5_heavy_warp_5000_polygons(gpu):
	rng = np.random.default_rng(5000)
	warp = synth.Warp.random(rng, PAGE, heavy=True)
	nodes = warp.node_grid(25)
	grid = Grid(nodes, 25)
	rings = synth.polygon_rings(rng, PAGE, 5000, integer_scale=6000 / 1280)
	got = np.concatenate(grid.transformer.transform_rings(rings))
	assert len(got) > 200000
	ogrid = odw.Grid(nodes, 25)
	want = np.concatenate(odw.transform_rings(ogrid.transformer, rings))
	assert np.abs(got - want).max() < 1e-3
	near_half = np.abs((want - np.floor(want)) - 0.5) < 1e-4
	assert np.array_equal(np.round(got)[~near_half], np.round(want)[~near_half])
	# heavy warp page: the LDS window of many tiles is large, some fall back to global gathers
	page = synth.text_page(rng, PAGE)
	dense = gpu.grid_fill(nodes, 25)
	np.testing.assert_array_equal(gpu.remap_u8(page, dense), ocv.remap_u8(page, dense, ocv.INTER_LINEAR, threads=8))


def test_identity_and_tiling_properties(gpu):
	rng = np.random.default_rng(7)
	src = rng.integers(0, 256, size=(3001, 2999), dtype=np.uint8)
	yy, xx = np.mgrid[0:3001, 0:2999].astype(np.float32)
	ident = np.stack([xx, yy], axis=-1)
	np.testing.assert_array_equal(gpu.remap_u8(src, ident), src)
	np.testing.assert_array_equal(gpu.remap_u8(src, ident, gpu.INTER_NEAREST), src)
	# remapping a band of the map equals the band of the remapped image (no tile depends on its position)
	m = ident + rng.normal(0, 3, size=ident.shape).astype(np.float32)
	full = gpu.remap_u8(src, m)
	np.testing.assert_array_equal(gpu.remap_u8(src, m[1003:2011]), full[1003:2011])
	np.testing.assert_array_equal(gpu.remap_u8(src, np.ascontiguousarray(m[:, 517:1900])), full[:, 517:1900])
	# wild maps (every tile's window overflows LDS): the global fallback gives the same bytes as the oracle
	wild = rng.uniform(-50, 3050, size=(257, 300, 2)).astype(np.float32)
	np.testing.assert_array_equal(gpu.remap_u8(src, wild), ocv.remap_u8(src, wild))


def test_transformer_is_exact_on_affine_grids(gpu):
	# a sheared, scaled node grid: the forward map is the inverse affine map (linearity property)
	gy, gx = np.mgrid[0:40, 0:30].astype(np.float64)
	X, Y = gx * 25, gy * 25
	nodes = np.stack([0.9 * X + 0.1 * Y + 5, -0.05 * X + 1.1 * Y - 3], axis=-1).astype(np.float32)
	grid = Grid(nodes, 25)
	rng = np.random.default_rng(1)
	q = rng.uniform([100, 100], [500, 800], size=(1000, 2))
	x, y = grid.transformer(tuple(q[:, 0]), tuple(q[:, 1]))
	A = np.array([[0.9, 0.1], [-0.05, 1.1]])
	# transformer maps dense[gy*25, gx*25] -> (gx*25, gy*25); dense[Y, X] = A (X - 12, Y - 12) + b inside
	inv = np.linalg.solve(A, (q - [5, -3]).T).T + 12
	assert np.abs(np.stack([x, y], axis=1) - inv).max() < 2e-2
