"""GPU parity of the HIP kernels against the CPU oracle, through the C ABI (ctypes).

Bars (BASELINE.json north_star): label maps / integer work bit-exact; uint8 images within +-1 LSB on
>= 99.9 % of pixels (here: bit-exact, both sides implement the same fixed-point arithmetic); float
grid within 1e-3 px (here: bit-exact); fp64 interpolation within 1e-9 (bit-exact away from shared
triangle edges)."""

import ctypes

import numpy as np
import pytest
import scipy.interpolate

from oracle import cv as ocv
from oracle import lingrid as olin
from origami_amd import synth

pytestmark = pytest.mark.gpu


def _nodes(rng, h, w, s, amp=6.0):
	gy, gx = np.mgrid[0:h, 0:w].astype(np.float64)
	x = (gx - 1) * s + amp * np.sin(gy / 3.0) + rng.normal(0, 0.5, size=(h, w))
	y = (gy - 1) * s + amp * np.cos(gx / 4.0) + rng.normal(0, 0.5, size=(h, w))
	return np.stack([x, y], axis=-1).astype(np.float32)


# ---- K1a ----------------------------------------------------------------------------------------

@pytest.mark.parametrize("h,w,s", [(12, 9, 8), (12, 9, 25), (1, 1, 5), (2, 7, 1), (40, 33, 25), (5, 300, 7),
	(1, 1, 8), (3, 1, 9), (1, 40, 9), (7, 131, 9), (100, 41, 8), (9, 62, 8), (4, 63, 16), (30, 3, 31)])
def test_grid_fill_bit_exact(gpu, h, w, s):
	rng = np.random.default_rng(h * 1000 + w * 10 + s)
	nodes = _nodes(rng, h, w, s)
	got = gpu.grid_fill(nodes, s)
	want = ocv.dense_grid(nodes, s)
	assert got.shape == (h * s, w * s, 2)
	np.testing.assert_array_equal(got, want)


def test_grid_fill_random_shapes(gpu):
	"""Forty seeded random (h, w, cell) with scales: both K1a kernels (cells below and from 8 pixels), widths that
	are and are not multiples of the 16-pixel line and of the 496-column tile, heights that end inside a row band."""
	rng = np.random.default_rng(2024)
	for _ in range(40):
		h, w, s = int(rng.integers(1, 40)), int(rng.integers(1, 90)), int(rng.integers(1, 41))
		nodes = _nodes(rng, h, w, s)
		sx, sy = (np.float32(1), np.float32(1)) if rng.random() < 0.5 else np.float32(rng.uniform(0.2, 1.5, 2))
		want = ocv.dense_grid(nodes, s)
		want[:, :, 0] *= sx
		want[:, :, 1] *= sy
		np.testing.assert_array_equal(gpu.grid_fill(nodes, s, sx, sy), want, err_msg="h=%d w=%d cell=%d" % (h, w, s))


@pytest.mark.parametrize("cell", [7, 8, 25])
def test_grid_fill_rows_into_any_8_byte_aligned_buffer(gpu, cell):
	"""ogd_grid_fill_rows_dev: a band of rows written to wherever the caller points (K1a's spans begin on the
	128-byte lines of the actual address: a buffer 8, 24 or 120 bytes past a line must give the same values)."""
	lib = gpu.load()
	rng = np.random.default_rng(cell)
	h, w = 9, 47
	nodes = _nodes(rng, h, w, cell)
	want = ocv.dense_grid(nodes, cell)
	Hd, Wd = want.shape[:2]
	d_nodes, d_out = ctypes.c_void_p(), ctypes.c_void_p()
	gpu.check(lib.ogd_dev_alloc(ctypes.byref(d_nodes), nodes.nbytes))
	gpu.check(lib.ogd_dev_alloc(ctypes.byref(d_out), want.nbytes + 256))
	try:
		gpu.check(lib.ogd_memcpy_h2d(d_nodes, nodes.ctypes.data, nodes.nbytes, None))
		for shift, y0, n in ((8, 0, Hd), (24, 3, Hd - 3), (120, cell * 4 + 1, 11), (0, Hd - 1, 1)):
			got = np.full((n, Wd, 2), np.nan, np.float32)
			gpu.check(lib.ogd_memcpy_h2d(d_out.value + shift, got.ctypes.data, got.nbytes, None))
			gpu.check(lib.ogd_grid_fill_rows_dev(d_nodes, h, w, cell, 1.0, 1.0, y0, n, d_out.value + shift, None))
			gpu.check(lib.ogd_memcpy_d2h(got.ctypes.data, d_out.value + shift, got.nbytes, None))
			gpu.check(lib.ogd_stream_sync(None))
			np.testing.assert_array_equal(got, want[y0:y0 + n])
		with pytest.raises(gpu.OgdError):
			gpu.check(lib.ogd_grid_fill_rows_dev(d_nodes, h, w, cell, 1.0, 1.0, 0, 1, d_out.value + 4, None))
	finally:
		lib.ogd_dev_free(d_nodes)
		lib.ogd_dev_free(d_out)


def test_grid_fill_scaled_copy(gpu):
	rng = np.random.default_rng(7)
	nodes = _nodes(rng, 10, 8, 25)
	sx, sy = np.float32(1280 / 2400), np.float32(2400 / 3400)
	got = gpu.grid_fill(nodes, 25, sx, sy)
	want = ocv.dense_grid(nodes, 25)
	want[:, :, 0] *= sx
	want[:, :, 1] *= sy
	np.testing.assert_array_equal(got, want)


def test_grid_points_strided(gpu):
	rng = np.random.default_rng(8)
	nodes = _nodes(rng, 14, 11, 25)
	got = gpu.grid_points_strided(nodes, 25, 25)
	want = ocv.dense_grid(nodes, 25)[::25, ::25]
	np.testing.assert_array_equal(got, want)


def test_grid_sample_bilinear(gpu):
	rng = np.random.default_rng(9)
	nodes = _nodes(rng, 9, 8, 10)
	dense = ocv.dense_grid(nodes, 10)
	Hd, Wd = dense.shape[:2]
	q = np.concatenate([
		rng.uniform([-5, -5], [Hd + 5, Wd + 5], size=(500, 2)),
		np.array([[0, 0], [Hd - 1, Wd - 1], [3, 4], [3.5, 4.0], [Hd, Wd]], dtype=np.float64)])
	interp = scipy.interpolate.RegularGridInterpolator(
		(np.arange(Hd), np.arange(Wd)), dense, method="linear", bounds_error=False, fill_value=None)
	want = interp(q)
	got = gpu.grid_sample_bilinear(nodes, 10, q)
	np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)


# ---- K2 -----------------------------------------------------------------------------------------

def _map_with_edge_cases(rng, H, W, Hd, Wd):
	yy, xx = np.mgrid[0:Hd, 0:Wd].astype(np.float32)
	m = np.stack([xx * (W / Wd) + 3 * np.sin(yy / 17), yy * (H / Hd) + 3 * np.cos(xx / 13)], axis=-1)
	m += rng.normal(0, 0.3, size=m.shape)
	m = m.astype(np.float32)
	m[0, :8] = [[-0.5, -0.5], [0, 0], [0.5, 0.5], [W - 1, H - 1], [W - 0.5, H - 0.5], [W, H], [-1, -1], [-1.0001, 3]]
	m[1, :6] = [[1.5, 2.5], [2.5, 3.5], [1e9, 1e9], [-1e9, 5], [np.nan, 2], [40000, 2]]
	m[2, :4] = [[3.015625, 4.984375], [3.0156249, 4.9843751], [W - 1.5, 0.5], [0.5, H - 1.5]]
	# exact 1/64 ties of the 1/32 rounding
	m[3, :Wd] = np.stack([np.arange(Wd) * 0.25 + 1 / 64, np.full(Wd, 5 + 3 / 64)], axis=-1)
	return m


@pytest.mark.parametrize("C", [1, 3, 4])
@pytest.mark.parametrize("interp", [ocv.INTER_LINEAR, ocv.INTER_NEAREST])
def test_remap_bit_exact(gpu, C, interp):
	rng = np.random.default_rng(100 + C)
	H, W, Hd, Wd = 256, 256, 203, 277
	src = rng.integers(0, 256, size=(H, W, C), dtype=np.uint8)
	src[100:140, :, :] = np.where(np.arange(W)[None, :, None] % 7 < 3, 0, 255)  # hard edges
	if C == 1:
		src = src[:, :, 0]
	m = _map_with_edge_cases(rng, H, W, Hd, Wd)
	got = gpu.remap_u8(src, m, interp)
	want = ocv.remap_u8(src, m, interp)
	np.testing.assert_array_equal(got, want)


def test_remap_strided_source_and_tiny_images(gpu):
	rng = np.random.default_rng(5)
	big = rng.integers(0, 256, size=(40, 64), dtype=np.uint8)
	src = big[:, :37]  # row stride 64, width 37
	m = rng.uniform(-2, 42, size=(31, 45, 2)).astype(np.float32)
	np.testing.assert_array_equal(gpu.remap_u8(src, m), ocv.remap_u8(src, m))
	one = np.array([[200]], dtype=np.uint8)
	m1 = rng.uniform(-1.5, 1.5, size=(9, 9, 2)).astype(np.float32)
	np.testing.assert_array_equal(gpu.remap_u8(one, m1), ocv.remap_u8(one, m1))
	np.testing.assert_array_equal(gpu.remap_u8(one, m1, 0), ocv.remap_u8(one, m1, 0))


@pytest.mark.parametrize("C", [1, 3])
def test_remap_dev_into_any_stride_and_alignment(gpu, C):
	"""K2 cuts every output row into pieces that begin on 64-byte sectors of the output ADDRESS (sheared tiles,
	ogd_remap_lds.h:rl_shear; one and three channels): every residue of the row stride mod 64 (drifting rows as well
	as rows that wrap), misaligned buffers, outputs narrower than a tile and wider than several; nothing outside the
	rows is written."""
	lib = gpu.load()
	rng = np.random.default_rng(64)
	H, W = 120, 150
	src = rng.integers(0, 256, size=(H, W) if C == 1 else (H, W, C), dtype=np.uint8)
	cap = 64 * 1024 * C
	d_src, d_map, d_out = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
	gpu.check(lib.ogd_dev_alloc(ctypes.byref(d_src), src.nbytes))
	gpu.check(lib.ogd_dev_alloc(ctypes.byref(d_map), 277 * 41 * 8))
	gpu.check(lib.ogd_dev_alloc(ctypes.byref(d_out), cap))
	launches = 0
	try:
		gpu.check(lib.ogd_memcpy_h2d(d_src, src.ctypes.data, src.nbytes, None))
		for Wd, Hd in [(1, 9), (3, 20), (63, 17), (64, 8), (65, 33), (200, 41), (277, 23)]:
			yy, xx = np.mgrid[0:Hd, 0:Wd].astype(np.float32)
			m = np.stack([xx * (W / Wd) + 2 * np.sin(yy / 5), yy * (H / Hd) + 2 * np.cos(xx / 7)], -1)
			m = np.ascontiguousarray((m + rng.normal(0, 0.4, m.shape)).astype(np.float32))
			m[0, 0] = [-3, -3]
			m[-1, -1] = [np.nan, 2]
			gpu.check(lib.ogd_memcpy_h2d(d_map, m.ctypes.data, m.nbytes, None))
			for interp in (ocv.INTER_LINEAR, ocv.INTER_NEAREST):
				row = Wd * C
				want = ocv.remap_u8(src, m, interp).reshape(Hd, row)
				for extra in range(64 if interp == ocv.INTER_LINEAR else 8):
					stride = row + extra
					n = (Hd - 1) * stride + row
					lead = (-d_out.value) % 64 + (0, 1, 37, 63, 4, 32)[extra % 6]
					total = lead + n + 128
					assert total <= cap
					out = np.full(total, 0xEE, dtype=np.uint8)
					gpu.check(lib.ogd_memcpy_h2d(d_out, out.ctypes.data, total, None))
					gpu.check(lib.ogd_remap_u8_dev(d_src, H, W, C, W * C, d_map, Hd, Wd, d_out.value + lead, stride, interp, None))
					gpu.check(lib.ogd_memcpy_d2h(out.ctypes.data, d_out, total, None))
					gpu.check(lib.ogd_stream_sync(None))
					launches += 1
					msg = "Wd %d Hd %d stride %d lead %d interp %d" % (Wd, Hd, stride, lead, interp)
					assert (out[:lead] == 0xEE).all() and (out[lead + n:] == 0xEE).all(), msg
					rows = np.lib.stride_tricks.as_strided(out[lead:], shape=(Hd, row), strides=(stride, 1))
					np.testing.assert_array_equal(rows, want, err_msg=msg)
					if extra:
						gaps = np.lib.stride_tricks.as_strided(out[lead + row:], shape=(Hd - 1, extra), strides=(stride, 1))
						assert (gaps == 0xEE).all(), msg
	finally:
		for d in (d_src, d_map, d_out):
			lib.ogd_dev_free(d)
	assert launches == 7 * 72


@pytest.mark.parametrize("C,cases", [(1, 800), (3, 500), (4, 300)])
def test_remap_dev_random_shapes_strides_and_maps(gpu, C, cases):
	"""Seeded random cases through ogd_remap_u8_dev against the oracle, bit for bit, for one, three (both: sheared
	tiles) and four channels: source and output sizes, output strides and misalignments, both interpolations; maps that
	are smooth, slanted, sheared beyond the LDS window, incoherent or folded with a NaN; guard bytes around and between
	the rows."""
	lib = gpu.load()
	rng = np.random.default_rng(7 + C)
	cap_src, cap_map, cap_out = 4 << 20, 8 << 20, 4 << 20
	d_src, d_map, d_out = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
	for d, n in ((d_src, cap_src), (d_map, cap_map), (d_out, cap_out)):
		gpu.check(lib.ogd_dev_alloc(ctypes.byref(d), n))
	for case in range(cases):
		H, W = int(rng.integers(1, 700)), int(rng.integers(1, 900))
		Hd, Wd = int(rng.integers(1, 120)), int(rng.integers(1, 1100))
		if rng.random() < 0.1:
			Wd = int(rng.integers(1, 70))
		row = Wd * C
		stride = row + int(rng.choice([0, 0, 1, 5, 27, 32, 59, 64, int(rng.integers(0, 200))]))
		lead = int(rng.integers(0, 130))
		interp = int(rng.choice([1, 1, 1, 0]))
		src = rng.integers(0, 256, size=(H, W) if C == 1 else (H, W, C), dtype=np.uint8)
		yy, xx = np.mgrid[0:Hd, 0:Wd].astype(np.float32)
		kind = int(rng.integers(0, 5))
		if kind == 0:
			m = np.stack([xx * (W / Wd), yy * (H / Hd)], -1) + rng.normal(0, 0.5, (Hd, Wd, 2))
		elif kind == 1:
			m = np.stack([xx + 0.04 * yy + rng.uniform(-3, 3), yy - 0.03 * xx + rng.uniform(-3, 3)], -1)
		elif kind == 2:
			m = np.stack([xx * 3.1 - yy * 2.2, yy * 2.7 + xx * 1.1], -1)
		elif kind == 3:
			m = rng.uniform([-W, -H], [2 * W, 2 * H], (Hd, Wd, 2))
		else:
			m = np.stack([np.abs(W / 2 - xx), np.abs(H / 2 - yy * 2)], -1)
			m[rng.integers(0, Hd), rng.integers(0, Wd)] = [np.nan, 1e30]
		m = np.ascontiguousarray(m.astype(np.float32))
		n = (Hd - 1) * stride + row
		total = lead + n + 200
		assert src.nbytes <= cap_src and m.nbytes <= cap_map and total <= cap_out
		want = ocv.remap_u8(src, m, interp).reshape(Hd, row)
		out = np.full(total, 0xA5, dtype=np.uint8)
		gpu.check(lib.ogd_memcpy_h2d(d_src, src.ctypes.data, src.nbytes, None))
		gpu.check(lib.ogd_memcpy_h2d(d_map, m.ctypes.data, m.nbytes, None))
		gpu.check(lib.ogd_memcpy_h2d(d_out, out.ctypes.data, total, None))
		gpu.check(lib.ogd_remap_u8_dev(d_src, H, W, C, W * C, d_map, Hd, Wd, d_out.value + lead, stride, interp, None))
		gpu.check(lib.ogd_memcpy_d2h(out.ctypes.data, d_out, total, None))
		gpu.check(lib.ogd_stream_sync(None))
		msg = "case %d: C %d src %dx%d out %dx%d stride %d lead %d interp %d kind %d" % (case, C, W, H, Wd, Hd, stride, lead, interp, kind)
		rows = np.lib.stride_tricks.as_strided(out[lead:], shape=(Hd, row), strides=(stride, 1))
		np.testing.assert_array_equal(rows, want, err_msg=msg)
		guard = np.ones(total, dtype=bool)
		for y in range(Hd):
			guard[lead + y * stride: lead + y * stride + row] = False
		assert (out[guard] == 0xA5).all(), "wrote outside its rows, " + msg
	for d in (d_src, d_map, d_out):
		lib.ogd_dev_free(d)


def test_remap_label_map_through_scaled_grid(gpu):
	"""layout.py:172-182: nearest remap of a label map through the page grid scaled to label size."""
	rng = np.random.default_rng(11)
	page = (600, 800)
	warp = synth.Warp(page, 0.03, 0.02, 1, 2)
	nodes = warp.node_grid(25)
	labels = synth.label_map(rng, (320, 480), 4)
	sx, sy = np.float32(320 / 600), np.float32(480 / 800)
	want_map = ocv.dense_grid(nodes, 25)
	want_map[:, :, 0] *= sx
	want_map[:, :, 1] *= sy
	want = ocv.remap_u8(labels, want_map, ocv.INTER_NEAREST)
	got_fused = gpu.dewarp_u8(labels, nodes, 25, sx, sy, ocv.INTER_NEAREST)
	got_unfused = gpu.remap_u8(labels, gpu.grid_fill(nodes, 25, sx, sy), ocv.INTER_NEAREST)
	np.testing.assert_array_equal(got_fused, want)
	np.testing.assert_array_equal(got_unfused, want)
	assert set(np.unique(got_fused)) <= {0, 1, 2, 3}


@pytest.mark.parametrize("C", [1, 3])
def test_fused_dewarp_equals_fill_then_remap(gpu, C):
	rng = np.random.default_rng(21 + C)
	page = (700, 500)
	warp = synth.Warp(page, 0.04, 0.03, 2, 1)
	nodes = warp.node_grid(25)
	src = synth.text_page(rng, page, C)
	want = ocv.remap_u8(src, ocv.dense_grid(nodes, 25))
	got = gpu.dewarp_u8(src, nodes, 25)
	np.testing.assert_array_equal(got, want)
	# a page is mostly recovered, not blank
	assert (got > 0).mean() > 0.5


def test_remap_and_fused_dewarp_random_cases(gpu):
	"""Thirty-six seeded random cases through K2 and the fused kernel against the oracle, bit for bit: source and
	output sizes that are no multiples of any tile, 1 / 3 / 4 channels, both interpolations, maps that are smooth,
	strongly sheared (windows beyond the LDS budget: the global-gather path), folded over themselves or leave the
	image; node grids of random cell size with warps of up to a cell, scaled for the label-map case."""
	rng = np.random.default_rng(4242)
	for case in range(36):
		C = int(rng.choice([1, 1, 3, 4]))
		interp = int(rng.choice([ocv.INTER_LINEAR, ocv.INTER_LINEAR, ocv.INTER_NEAREST]))
		H, W = int(rng.integers(1, 400)), int(rng.integers(1, 400))
		src = rng.integers(0, 256, size=(H, W, C), dtype=np.uint8)
		if C == 1:
			src = src[:, :, 0]
		msg = "case %d: C=%d interp=%d src %dx%d" % (case, C, interp, W, H)
		if case % 2 == 0:  # K2 through a dense map
			Hd, Wd = int(rng.integers(1, 300)), int(rng.integers(1, 300))
			yy, xx = np.mgrid[0:Hd, 0:Wd].astype(np.float32)
			kind = case % 8
			if kind == 0:
				m = np.stack([xx * (W / Wd), yy * (H / Hd)], -1) + rng.normal(0, 0.4, (Hd, Wd, 2))
			elif kind == 2:
				m = np.stack([xx * 3.7 - yy * 2.9, yy * 4.1 + xx * 1.3], -1)  # sheared: wide windows
			elif kind == 4:
				m = np.stack([np.abs(W / 2 - xx * 1.5), np.abs(H / 2 - yy * 1.5)], -1)  # folded
			else:
				m = rng.uniform([-W, -H], [2 * W, 2 * H], (Hd, Wd, 2))  # mostly outside, incoherent
			m = m.astype(np.float32)
			np.testing.assert_array_equal(gpu.remap_u8(src, m, interp), ocv.remap_u8(src, m, interp), err_msg=msg)
		else:  # the fused kernel through a node grid
			s = int(rng.integers(1, 40))
			h, w = int(rng.integers(1, 16)), int(rng.integers(1, 16))
			gy, gx = np.mgrid[0:h, 0:w].astype(np.float64)
			nodes = np.stack([(gx - 1) * s * W / max(1, (w - 2) * s), (gy - 1) * s * H / max(1, (h - 2) * s)], -1)
			nodes = (nodes + rng.normal(0, 0.5 * s, nodes.shape)).astype(np.float32)
			sx, sy = (1.0, 1.0) if rng.random() < 0.6 else (float(rng.uniform(0.3, 1.2)), float(rng.uniform(0.3, 1.2)))
			want_map = ocv.dense_grid(nodes, s)
			want_map[:, :, 0] *= np.float32(sx)
			want_map[:, :, 1] *= np.float32(sy)
			want = ocv.remap_u8(src, want_map, interp)
			np.testing.assert_array_equal(gpu.dewarp_u8(src, nodes, s, sx, sy, interp), want, err_msg=msg + " cell %d nodes %dx%d" % (s, w, h))


def test_remap_rejects_bad_arguments(gpu):
	src = np.zeros((4, 4, 2), dtype=np.uint8)
	m = np.zeros((4, 4, 2), dtype=np.float32)
	with pytest.raises(gpu.OgdError):
		gpu.remap_u8(src, m)
	with pytest.raises(gpu.OgdError):
		gpu.remap_u8(np.zeros((4, 4), dtype=np.uint8), m, 7)


# ---- K3 / K1b -------------------------------------------------------------------------------------

def _scattered(rng, n, size, vector=False):
	W, H = size
	pts = rng.uniform([0, 0], [W, H], size=(n, 2))
	if vector:
		vals = np.stack([pts[:, 0] * 0.9 + 3 * np.sin(pts[:, 1] / 50), pts[:, 1] * 1.1 + 2 * np.cos(pts[:, 0] / 40)], axis=1)
	else:
		vals = 0.03 * np.sin(pts[:, 0] / 100) + 0.02 * np.cos(pts[:, 1] / 130)
	return pts, vals


def _close(got, want, atol=1e-9, same_bits=True):
	assert got.shape == want.shape
	assert np.array_equal(np.isnan(got), np.isnan(want))
	np.testing.assert_allclose(got, want, rtol=0, atol=atol)
	if same_bits:
		# identical containing triangle and vertex order => identical bits; only shared-edge hits may differ
		same = (got == want) | np.isnan(want)
		assert same.mean() > 0.99


@pytest.fixture(params=["ogd_delaunay_points", "qhull"])
def triangulator(request, monkeypatch):
	"""lininterp's two triangulators.  ogd_delaunay_points returns Qhull's triangles (tests/test_host.py) but lists
	the three vertices of a triangle in its own order, and scipy's barycentric arithmetic depends on that order in
	the last bits: values agree to 1e-9.  With Qhull itself (the route taken on ties, twins and collinear input,
	forced here) K3 reproduces LinearNDInterpolator bit for bit."""
	if request.param == "qhull":
		from origami_amd import _shim

		def refuse(points, twins=False, transform=True):
			raise _shim.OgdError(_shim.ERR_UNSUPPORTED, "forced by the test")
		monkeypatch.setattr(_shim, "delaunay_points", refuse)
	return request.param


@pytest.mark.parametrize("n,vector", [(64, False), (2000, False), (500, True), (4, False)])
def test_tri_eval_matches_scipy(gpu, n, vector, triangulator):
	from origami_amd.lingrid import InterpolatorFactory
	rng = np.random.default_rng(n + vector)
	size = (600, 800)
	pts, vals = _scattered(rng, n, size, vector)
	bounds = (0, 0, size[0], size[1])
	values = list(vals) if not vector else vals
	want_f = olin.Factory(list(pts), values, bounds).interpolator()
	got_f = InterpolatorFactory(list(pts), values, bounds).interpolator
	q = np.concatenate([
		rng.uniform([-50, -50], [size[0] + 50, size[1] + 50], size=(5000, 2)),
		pts[:20],  # exactly on vertices
		(pts[:-1][:20] + pts[1:][:20]) / 2,
		np.array([[0, 0], [size[0], size[1]], [0, size[1]], [size[0], 0]], dtype=np.float64)])
	bits = triangulator == "qhull"
	_close(got_f(q), want_f(q), same_bits=bits)
	q32 = q.astype(np.float32)  # Field.get queries in float32
	_close(got_f(q32), want_f(q32), same_bits=bits)


def test_tri_eval_collinear_and_errors(gpu, triangulator):
	from origami_amd.lingrid import InterpolatorFactory, lininterp
	pts = [np.array([100.0, 100.0]), np.array([200.0, 150.0]), np.array([300.0, 200.0])]
	vals = [0.01, 0.02, 0.04]
	bounds = (0, 0, 600, 800)
	want_f = olin.Factory(pts, vals, bounds).interpolator()
	got_fac = InterpolatorFactory(pts, vals, bounds)
	assert got_fac.is_collinear
	rng = np.random.default_rng(0)
	q = rng.uniform([0, 0], [600, 800], size=(1000, 2))
	_close(got_fac.interpolator(q), want_f(q), same_bits=triangulator == "qhull")
	with pytest.raises(ValueError, match="not enough values"):
		lininterp(pts[:1], vals[:1], bounds)


def test_transform_computed_on_the_device_is_the_hosts(gpu):
	"""ogd_tri_create_scipy against ogd_tri_create fed the transform ogd_lattice_delaunay / ogd_delaunay_points
	write (scipy's, tests/test_host.py): one source for both, so every interpolated value has the same bits;
	slivers and flat simplices (scipy's NaN rows) included."""
	from origami_amd import _shim
	rng = np.random.default_rng(11)
	# a sheared lattice with a few collinear border triples, and scattered points with near-degenerate slivers
	h, w = 23, 31
	yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
	nodes = np.stack([xx * 40 + yy * 3 + rng.uniform(-6, 6, (h, w)), yy * 40 + rng.uniform(-6, 6, (h, w))], axis=-1)
	nodes = np.round(nodes * 1024) / 1024  # the lattice route wants coordinates it can do exact arithmetic on
	simplices, _, transform = _shim.lattice_delaunay(nodes)
	cases = [(nodes.reshape(-1, 2), simplices, transform)]
	pts = rng.uniform(0, 1000, (3000, 2))
	pts[:200, 1] = pts[:200, 0] * 0.5 + rng.uniform(-1e-9, 1e-9, 200)  # nearly collinear run
	simplices, transform, _ = _shim.delaunay_points(pts)
	cases.append((pts, simplices, transform))
	# flat simplices given by hand: the transform has NaN rows, the interpolant skips them
	flat_pts = np.array([[0, 0], [1, 1], [2, 2], [0, 2], [3, 3]], dtype=np.float64)
	flat_simp = np.array([[0, 1, 2], [0, 2, 3], [0, 1, 4]], dtype=np.int32)
	cases.append((flat_pts, flat_simp, None))
	for points, simp, tr in cases:
		vals = rng.normal(size=(len(points), 2))
		lo, hi = points.min(axis=0), points.max(axis=0)
		q = np.concatenate([rng.uniform(lo - 5, hi + 5, size=(20000, 2)), points[:50]])
		device = _shim.Triangulation(points, simp, vals, scipy_transform=True)
		if tr is None:
			a, b, c = (points[simp[:, k]] for k in range(3))
			assert np.any(np.cross(a - c, b - c) == 0)
			tr = np.empty((len(simp), 3, 2))
			for t in range(len(simp)):
				T = np.array([a[t] - c[t], b[t] - c[t]]).T
				tr[t, :2] = np.linalg.inv(T) if np.linalg.matrix_rank(T) == 2 else np.nan
				tr[t, 2] = c[t]
			got = _shim.tri_eval(None, device, None, q)
			want = _shim.tri_eval(None, _shim.Triangulation(points, simp, vals, tr), None, q)
			assert np.array_equal(np.isnan(got), np.isnan(want))
			np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
			continue
		host = _shim.Triangulation(points, simp, vals, tr)
		np.testing.assert_array_equal(_shim.tri_eval(None, device, None, q), _shim.tri_eval(None, host, None, q))


def test_tri_eval_ragged_equals_flat(gpu):
	from origami_amd.lingrid import InterpolatorFactory
	rng = np.random.default_rng(3)
	size = (600, 800)
	pts, vals = _scattered(rng, 800, size, True)
	f = InterpolatorFactory(pts, vals, (0, 0, size[0], size[1])).interpolator
	rings = synth.polygon_rings(rng, size, 40, mu=30) + [np.empty((0, 2)), rng.uniform(0, 600, size=(1, 2))]
	offsets = np.concatenate([[0], np.cumsum([len(r) for r in rings])]).astype(np.int64)
	flat = np.concatenate(rings)
	np.testing.assert_array_equal(f.ragged(flat, offsets), f(flat))


@pytest.mark.parametrize("vector", [False, True])
def test_tri_rasterize_matches_griddata(gpu, vector, triangulator):
	from origami_amd.lingrid import lingrid
	rng = np.random.default_rng(17 + vector)
	W, H = 150, 110
	pts, vals = _scattered(rng, 300, (W - 1, H - 1), vector)
	values = list(vals) if not vector else vals
	want = olin.lingrid(list(pts), values, W, H)
	got = lingrid(list(pts), values, W, H)
	assert got.shape == want.shape == ((W, H, 2) if vector else (W, H))
	_close(got, want, same_bits=triangulator == "qhull")


def test_tri_rasterize_equals_point_queries(gpu):
	from origami_amd.lingrid import InterpolatorFactory
	rng = np.random.default_rng(23)
	pts, vals = _scattered(rng, 500, (299, 199), False)
	fac = InterpolatorFactory(list(pts), list(vals), (0, 0, 299, 199))
	dense = fac.grid
	xs, ys = np.mgrid[0:300, 0:200]
	q = np.stack([xs.ravel(), ys.ravel()], axis=1).astype(np.float64)
	np.testing.assert_array_equal(dense.ravel(), fac.interpolator(q))


def test_rings_self_cross_matches_the_exact_reference(gpu):
	"""ogd_rings_self_cross (wavefront per ring, fp64 orientation signs) against oracle/rings.py — an independent
	O(n^2) double loop in exact rational arithmetic that shares no code with the product — and, as a second
	opinion, the product's own numpy form: simple rings, bow-ties, rings that touch themselves at a vertex or
	along an edge, tiny rings."""
	from oracle import rings as orings
	from origami_amd import geometry
	rng = np.random.default_rng(42)
	rings = synth.polygon_rings(rng, (3000, 4000), 300, mu=30)
	for _ in range(100):  # random vertex order: most of these cross themselves
		p = rng.integers(0, 50, size=(int(rng.integers(3, 90)), 2)).astype(np.float64)
		rings.append(np.vstack([p, p[:1]]))
	rings.append(np.array([[0, 0], [4, 4], [4, 0], [0, 4], [0, 0]], dtype=np.float64))  # bow-tie
	rings.append(np.array([[0, 0], [4, 0], [4, 4], [2, 0], [0, 4], [0, 0]], dtype=np.float64))  # touches an edge
	rings.append(np.array([[0, 0], [4, 0], [2, 2], [4, 4], [0, 4], [2, 2], [0, 0]], dtype=np.float64))  # vertex twice
	rings.append(np.array([[0, 0], [1, 0], [0, 1], [0, 0]], dtype=np.float64))  # triangle
	rings.append(np.array([[0, 0], [1, 1], [0, 0]], dtype=np.float64))  # degenerate: two segments
	rings.append(np.array([[0, 0], [4, 0], [4, 4], [0, 4], [0, 0]], dtype=np.float64))
	got = gpu.rings_self_cross(rings)
	want = np.array([orings.ring_self_crosses(r) for r in rings])
	assert got.dtype == bool and got.shape == want.shape
	np.testing.assert_array_equal(got, want)
	np.testing.assert_array_equal(np.array([geometry._segments_cross(r) for r in rings]), want)
	assert want[:300].sum() == 0 and want[300:400].sum() > 50
	assert len(gpu.rings_self_cross([])) == 0


@pytest.mark.parametrize("shape", [(3, 32767), (32767, 3)])
def test_remap_at_the_int16_coordinate_limit(gpu, shape):
	"""Largest image remap accepts (H, W < 32768: cv2.remap's int16 coordinates): taps at and beyond the far
	edge, where the packed int16 bounding box of a tile runs up to 32767 and the zero border starts."""
	H, W = shape
	rng = np.random.default_rng(H)
	src = rng.integers(1, 256, size=(H, W), dtype=np.uint8)
	n = 4096
	far = np.array([W, H], dtype=np.float32)
	m = np.empty((8, n, 2), dtype=np.float32)
	m[..., 0] = rng.uniform(W - 70, W + 3, size=(8, n))
	m[..., 1] = rng.uniform(H - 70, H + 3, size=(8, n))
	m[0, :8] = far - np.array([[1, 1], [0.5, 1], [1, 0.5], [1.015625, 1.015625], [0, 0], [2, 2], [1, 40000], [40000, 1]], dtype=np.float32)
	m = np.maximum(m, -5)
	for interp in (ocv.INTER_LINEAR, ocv.INTER_NEAREST):
		np.testing.assert_array_equal(gpu.remap_u8(src, m, interp), ocv.remap_u8(src, m, interp))
	# smooth map along the long axis: the LDS window path, not only the erratic-map fallback
	k = max(H, W)
	t = np.arange(k - 4000, k + 40, dtype=np.float32)
	sm = np.zeros((8, len(t), 2), dtype=np.float32)
	if W > H:
		sm[..., 0] = t[None, :] + 0.3
		sm[..., 1] = np.linspace(-0.5, 2.5, 8, dtype=np.float32)[:, None]
	else:
		sm[..., 1] = t[None, :] + 0.3
		sm[..., 0] = np.linspace(-0.5, 2.5, 8, dtype=np.float32)[:, None]
	np.testing.assert_array_equal(gpu.remap_u8(src, sm), ocv.remap_u8(src, sm))
	with pytest.raises(gpu.OgdError):
		gpu.remap_u8(np.zeros((2, 32768), dtype=np.uint8), m)


def test_empty_and_degenerate_inputs(gpu):
	"""Empty ragged batches, rings of one point, a 1 x 1 node grid, a 1 x 1 map."""
	pts = np.array([[0, 0], [10, 0], [0, 10], [10, 10]], dtype=np.float64)
	vals = np.array([[0.0, 1.0], [1.0, 1.0], [0.0, 2.0], [1.0, 2.0]])
	tri = gpu.Triangulation(pts, np.array([[0, 1, 2], [1, 3, 2]], dtype=np.int32), vals)
	assert gpu.tri_eval(None, tri, None, np.empty((0, 2))).shape == (0, 2)
	assert gpu.tri_eval_ragged(None, tri, None, np.empty((0, 2)), np.zeros(1, dtype=np.int64)).shape == (0, 2)
	out = gpu.tri_eval_ragged(None, tri, None, np.array([[5.0, 5.0], [1.0, 1.0]]), np.array([0, 0, 1, 1, 2], dtype=np.int64))
	np.testing.assert_allclose(out, [[0.5, 1.5], [0.1, 1.1]], atol=1e-12)
	assert np.isnan(gpu.tri_eval(None, tri, None, np.array([[20.0, 20.0]]))).all()  # outside: NaN, like scipy
	one = np.array([[[3.25, 7.5]]], dtype=np.float32)
	np.testing.assert_array_equal(gpu.grid_fill(one, 4), np.broadcast_to(one, (4, 4, 2)))
	src = np.arange(12, dtype=np.uint8).reshape(3, 4)
	np.testing.assert_array_equal(gpu.remap_u8(src, np.array([[[2.0, 1.0]]], dtype=np.float32)), [[6]])
	np.testing.assert_array_equal(gpu.dewarp_u8(src, np.array([[[2.0, 1.0]]], dtype=np.float32), 2), np.full((2, 2), 6, dtype=np.uint8))
