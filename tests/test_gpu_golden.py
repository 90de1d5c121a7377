"""GPU product path (origami_amd, through the C ABI) against the committed golden fixtures
(tests/golden/*.npz, made by tests/golden/make_golden.py from the CPU oracle).

Bars (BASELINE.json north_star): label maps bit-exact; uint8 images within +-1 LSB on >= 99.9 % of
pixels (asserted bit-exact here: both sides implement the same fixed-point arithmetic); float grid
within 1e-3 px (asserted bit-exact for the dense grid, 1e-3 px for the marched node grid); contour
coordinates within 1e-3 px and their rounded (integer) form bit-exact except within 1e-4 px of a .5
boundary."""

from pathlib import Path

import numpy as np
import PIL.Image
import pytest

from origami_amd import synth  # noqa: F401
from origami_amd.dewarp import Dewarper, Field, Geometry, Grid, INTER_NEAREST, Samples

pytestmark = pytest.mark.gpu

GOLDEN = Path(__file__).resolve().parent / "golden"


def golden(name):
	return np.load(GOLDEN / ("%s.npz" % name), allow_pickle=False)


def _samples(size, points, values):
	s = Samples(Geometry(*size))
	s.extend(list(points), list(values))
	return s


def test_field_matches_golden(gpu):
	g = golden("field")
	size = tuple(int(v) for v in g["size"])
	probes = g["probes"]
	f64 = Field(_samples(size, g["n64_points"], g["n64_values"]), size, 0)
	np.testing.assert_allclose(f64.get(probes.copy()), g["n64_field"], rtol=0, atol=1e-6)
	f3 = Field(_samples(size, g["n3_points"], g["n3_values"]), size, 0)
	np.testing.assert_allclose(f3.get(probes.copy()), g["n3_field"], rtol=0, atol=1e-6)
	f2 = Field(_samples(size, g["n3_points"][:2], g["n3_values"][:2]), size, np.pi / 2)
	np.testing.assert_array_equal(f2.get(probes.copy()), g["n2_field"])


def test_grid_create_matches_golden(gpu):
	g = golden("grid")
	size = tuple(int(v) for v in g["size"])

	class Page:
		warped = PIL.Image.new("L", size)

	grid = Grid.create(
		Page, _samples(size, g["h_points"], g["h_values"]), _samples(size, g["v_points"], g["v_values"]),
		grid_res=25)
	hv = grid.points("sample")
	assert hv.shape == g["grid_hv"].shape and hv.dtype == np.float32
	assert np.abs(hv - g["grid_hv"]).max() < 1e-3  # float grid within 1e-3 px


def test_dense_grid_and_inverse_match_golden(gpu):
	g = golden("dense")
	for s in (8, 25):
		grid = Grid(g["nodes_s%d" % s], s)
		np.testing.assert_array_equal(grid.points("full"), g["dense_s%d" % s])
	got = Grid(g["nodes_s8"], 8).inverse(g["inverse_q_xy"])
	np.testing.assert_allclose(got, g["inverse_s8"], rtol=0, atol=1e-9)


def test_remap_matches_golden(gpu):
	g = golden("remap")
	src, m = g["src_rgb"], g["map"]
	gray = np.ascontiguousarray(src[:, :, 0])
	np.testing.assert_array_equal(gpu.remap_u8(gray, m, gpu.INTER_LINEAR), g["gray_linear"])
	np.testing.assert_array_equal(gpu.remap_u8(gray, m, gpu.INTER_NEAREST), g["gray_nearest"])
	np.testing.assert_array_equal(gpu.remap_u8(src, m, gpu.INTER_LINEAR), g["rgb_linear"])
	np.testing.assert_array_equal(gpu.remap_u8(src, m, gpu.INTER_NEAREST), g["rgb_nearest"])


def test_dewarper_and_transformer_match_golden(gpu):
	g = golden("transformer")
	size = tuple(int(v) for v in g["size"])
	grid = Grid(g["nodes"], 25)
	dewarper = Dewarper(PIL.Image.fromarray(g["page"]), grid)
	np.testing.assert_array_equal(np.array(dewarper.dewarped), g["page_dewarped"])
	labels = np.array(dewarper.dewarp_image(PIL.Image.fromarray(g["labels"]), INTER_NEAREST))
	np.testing.assert_array_equal(labels, g["labels_dewarped"])  # label maps: bit-exact
	# a second call sees the same (unscaled) grid: the scale is applied to a private copy
	np.testing.assert_array_equal(np.array(Dewarper(PIL.Image.fromarray(g["page"]), grid).dewarped), g["page_dewarped"])

	off = g["ring_offsets"]
	rings = [g["ring_points"][a:b] for a, b in zip(off[:-1], off[1:])]
	want = g["ring_warped"]
	got = np.concatenate(grid.transformer.transform_rings(rings))
	assert np.abs(got - want).max() < 1e-3  # float contour coordinates within 1e-3 px
	near_half = np.abs((want - np.floor(want)) - 0.5) < 1e-4
	assert np.array_equal(np.round(got)[~near_half], np.round(want)[~near_half])  # integer contours bit-exact
	# per-ring calls (what shapely.ops.transform does) equal the batched launch
	x, y = grid.transformer(tuple(rings[0][:, 0]), tuple(rings[0][:, 1]))
	np.testing.assert_array_equal(np.stack([x, y], axis=1), got[:len(rings[0])])
