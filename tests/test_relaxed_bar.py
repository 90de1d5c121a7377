"""What the north star's OWN bar would buy: "remapped uint8 images within +-1 LSB on >= 99.9 % of pixels, float grid
within 1e-3 px" (BASELINE.json) — priced once, on the CPU, against the oracle.

Every kernel in this repository meets the stricter bar (bit-exact against oracle/cv_oracle.c).  The fused kernel
(what Dewarper.dewarped runs) is bound by its instruction count, and 8 of its ~42 vector instructions per pixel are
the float32 vertical lerp + quantisation that reproduce cv2.resize's rounding bit for bit.  The
cheapest map that stays within the relaxed bar replaces them by two integer additions: inside a node cell the map
is linear along a row, so q(X) = (S + (X - X0) * D) >> F with S = the exact 1/32-px map value at the cell's first
pixel and D = the per-pixel step, both in fixed point.  This test builds that map in numpy, measures how far it is
from the exact one and what it does to the pixels of a text page — at page coordinates below 3400 (configs[0]) and
around 6000-8000 (the far corner of a broadsheet page, where float32 resolves 4.9e-4 px) — and asserts only what the
north star states.  The numbers it prints are the ones DESIGN.md §4.7 quotes; the kernels stay bit-exact."""

import numpy as np

from oracle import cv as ocv
from origami_amd import synth

FRAC = 10  # fixed-point bits below 1/32 px: 8192 * 32 * 2^10 < 2^31


def incremental_map(dense, s):
	"""(Hd, Wd, 2) float32 dense map -> int64 1/32-px coordinates from per-cell fixed-point increments along rows."""
	Hd, Wd = dense.shape[:2]
	w = Wd // s
	X = np.arange(Wd)
	g = np.clip(np.floor((X + 0.5) / s - 0.5).astype(int), 0, w - 2)  # node cell of pixel X
	X0 = np.ceil(s * (np.arange(w) + 0.5) - 0.5).astype(int)  # first pixel whose left node is g
	ref = np.rint(dense.astype(np.float64) * 32).astype(np.int64)  # cv2.remap's quantisation of the exact map
	start = np.rint(dense[:, X0, :].astype(np.float64) * 32 * (1 << FRAC)).astype(np.int64)  # (Hd, w, 2)
	step = np.rint((start[:, 1:, :] - start[:, :-1, :]) / s).astype(np.int64)  # (Hd, w - 1, 2)
	q = (start[:, g, :] + (X - X0[g])[None, :, None] * step[:, g, :] + (1 << (FRAC - 1))) >> FRAC
	flat = (X < X0[0]) | (X >= X0[w - 1])  # the clamped margins are constant: take them as they are
	q[:, flat, :] = ref[:, flat, :]
	return q, ref


def _price(name, page, nodes, cell, rows=None):
	dense = ocv.dense_grid(nodes, cell, 8)
	if rows is not None:
		dense = dense[rows]
	q, ref = incremental_map(dense, cell)
	err_px = np.abs(q / 32.0 - dense.astype(np.float64)).max()  # includes the 1/64 px of the quantisation itself
	moved = (q != ref).any(axis=2)
	want = ocv.remap_u8(page, (ref / 32.0).astype(np.float32))
	got = ocv.remap_u8(page, (q / 32.0).astype(np.float32))
	assert np.array_equal(want, ocv.remap_u8(page, dense))  # feeding the quantised map back is the identity
	diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
	n = diff.size
	stats = dict(
		moved=float(moved.mean()), step=float(np.abs(q - ref).max()), differ=float((diff > 0).mean()),
		beyond1=float((diff > 1).mean()), worst=int(diff.max()), err_px=float(err_px))
	print("%-34s map moved by 1/32 px at %.2f %% of pixels (never by more than %d steps), pixels that differ %.3f %%, by more than 1 LSB %.4f %% (worst %d LSB) of %d" % (
		name, 100 * stats["moved"], stats["step"], 100 * stats["differ"], 100 * stats["beyond1"], stats["worst"], n))
	return stats


def test_fixed_point_incremental_map_against_the_north_stars_own_bar():
	rng = np.random.default_rng(77)
	# configs[0]: a whole 2400 x 3400 page
	size = (2400, 3400)
	small = _price("2400x3400 page, cell 25", synth.text_page(rng, size), synth.Warp(size, 0.03, 0.025, 1, 2).node_grid(25), 25)
	# the far end of a broadsheet page (coordinates 6000 .. 8000): dense rows 6500 .. 7500 of configs[1]'s map
	size = (6000, 8000)
	nodes = synth.Warp(size, 0.03, 0.025, 1, 2).node_grid(25)
	far = _price("6000x8000 page, rows 6500-7500", synth.text_page(rng, size), nodes, 25, rows=slice(6500, 7500))
	for stats in (small, far):
		# the north star's bar: +-1 LSB on >= 99.9 % of pixels; the map within 1e-3 px of the exact one beyond the
		# half-step of cv2's own 1/32-px quantisation
		assert stats["beyond1"] <= 1e-3
		assert stats["step"] <= 1
		assert stats["err_px"] <= 1 / 64 + 1 / 32 + 1e-3
	# and it is NOT the bar this repository holds its kernels to: pixels do change
	assert far["differ"] > 0
