#!/usr/bin/env python3
"""Generates the golden fixtures of tests/golden/ from the CPU oracle (oracle/).

    python tests/golden/make_golden.py

The reference (poke1024/origami) ships no tests, golden vectors or fixtures for the dewarp path and
cannot be imported in the build container (cv2 and shapely are missing), so these
vectors are outputs of the oracle itself: they pin the oracle against regressions and give the GPU
tests fixed inputs with fixed expected outputs.  PARITY UNPINNED for the cv2 / shapely arithmetic;
the scipy part (Qhull + LinearNDInterpolator) is the reference's own library, called directly.

All inputs are seeded; the fixtures are small (< 1 MB in total).
"""

import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))

from oracle import build_oracle  # noqa: E402
from oracle import cv as ocv  # noqa: E402
from oracle import dewarp as odw  # noqa: E402
from origami_amd import synth  # noqa: E402  (input generators only; no compute)


def field_fixture():
	"""(1) flow samples -> Field.get at 200 probe points, for n = 64, 3 collinear, 2."""
	rng = np.random.default_rng(101)
	size = (600, 800)
	warp = synth.Warp(size, 0.03, 0.025, 1, 2)
	probes = rng.uniform([-20, -20], [620, 820], size=(200, 2)).astype(np.float32)
	out = dict(size=np.array(size), probes=probes)
	pts = rng.uniform([0, 0], size, size=(64, 2))
	val = warp.phi_h(pts[:, 0], pts[:, 1]) + rng.normal(0, 0.002, size=64)
	out["n64_points"], out["n64_values"] = pts, val
	out["n64_field"] = odw.Field(odw.Samples(size, list(pts), list(val)), size, 0).get(probes.copy())
	cp = np.array([[100.0, 100.0], [200.0, 150.0], [300.0, 200.0]])
	cv_ = np.array([0.01, 0.02, 0.04])
	out["n3_points"], out["n3_values"] = cp, cv_
	out["n3_field"] = odw.Field(odw.Samples(size, list(cp), list(cv_)), size, 0).get(probes.copy())
	out["n2_field"] = odw.Field(odw.Samples(size, list(cp[:2]), list(cv_[:2])), size, np.pi / 2).get(probes.copy())
	return out


def grid_fixture():
	"""(2) full Grid.create on a 600x800 page, cell 25."""
	rng = np.random.default_rng(102)
	size = (600, 800)
	warp = synth.Warp(size, 0.03, 0.02, 1, 2)
	h, v = synth.flow_samples(rng, warp, 150, 200)
	hp, hv_ = np.array(h.points), np.array(h.values)
	vp, vv = np.array(v.points), np.array(v.values)
	grid = odw.Grid.create(
		size, odw.Samples(size, list(hp), list(hv_)), odw.Samples(size, list(vp), list(vv)), grid_res=25)
	return dict(size=np.array(size), h_points=hp, h_values=hv_, v_points=vp, v_values=vv, grid_hv=grid.hv)


def dense_fixture():
	"""(3) dense sampling grid of a 12x9 node grid, cell 8 and 25; strided samples; inverse map."""
	rng = np.random.default_rng(103)
	gy, gx = np.mgrid[0:12, 0:9].astype(np.float64)
	out = {}
	for s in (8, 25):
		x = (gx - 1) * s + 5 * np.sin(gy / 3.0) + rng.normal(0, 0.5, size=gx.shape)
		y = (gy - 1) * s + 5 * np.cos(gx / 4.0) + rng.normal(0, 0.5, size=gx.shape)
		nodes = np.stack([x, y], axis=-1).astype(np.float32)
		out["nodes_s%d" % s] = nodes
		out["dense_s%d" % s] = ocv.dense_grid(nodes, s)
	g = odw.Grid(out["nodes_s8"], 8)
	q = rng.uniform([-4, -4], [9 * 8 + 4, 12 * 8 + 4], size=(64, 2))
	out["inverse_q_xy"] = q
	out["inverse_s8"] = g.inverse(q)
	return out


def remap_fixture():
	"""(4) remap of a noise + edge image (gray, RGB), linear and nearest, with out-of-bounds map
	entries, exact integer / half-integer coordinates and 1/64 ties."""
	rng = np.random.default_rng(104)
	H, W, Hd, Wd = 96, 128, 75, 101
	src = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
	src[30:50] = np.where(np.arange(W)[None, :, None] % 7 < 3, 0, 255)
	yy, xx = np.mgrid[0:Hd, 0:Wd].astype(np.float32)
	m = np.stack([xx * (W / Wd) + 3 * np.sin(yy / 17), yy * (H / Hd) + 3 * np.cos(xx / 13)], axis=-1)
	m = (m + rng.normal(0, 0.3, size=m.shape)).astype(np.float32)
	m[0, :8] = [[-0.5, -0.5], [0, 0], [0.5, 0.5], [W - 1, H - 1], [W - 0.5, H - 0.5], [W, H], [-1, -1], [-1.0001, 3]]
	m[1, :6] = [[1.5, 2.5], [2.5, 3.5], [1e9, 1e9], [-1e9, 5], [np.nan, 2], [40000, 2]]
	m[2, :4] = [[3.015625, 4.984375], [3.0156249, 4.9843751], [W - 1.5, 0.5], [0.5, H - 1.5]]
	m[3, :] = np.stack([np.arange(Wd) * 0.25 + 1 / 64, np.full(Wd, 5 + 3 / 64)], axis=-1)
	out = dict(src_rgb=src, map=m)
	gray = np.ascontiguousarray(src[:, :, 0])
	out["gray_linear"] = ocv.remap_u8(gray, m, ocv.INTER_LINEAR)
	out["gray_nearest"] = ocv.remap_u8(gray, m, ocv.INTER_NEAREST)
	out["rgb_linear"] = ocv.remap_u8(src, m, ocv.INTER_LINEAR)
	out["rgb_nearest"] = ocv.remap_u8(src, m, ocv.INTER_NEAREST)
	return out


def transformer_fixture():
	"""(5) forward transformer on 32 rings, including vertices outside the grid's bounding box, and
	the dewarped page / label map of the same grid."""
	rng = np.random.default_rng(105)
	size = (500, 650)
	warp = synth.Warp(size, 0.04, 0.03, 1, 2)
	nodes = warp.node_grid(25)
	grid = odw.Grid(nodes, 25)
	rings = synth.polygon_rings(rng, size, 30, mu=12, lo=4, hi=24, integer_scale=500 / 320)
	rings.append(np.array([[-40.0, -30.0], [250.0, -60.0], [560.0, 700.0], [-40.0, -30.0]]))
	rings.append(np.array([[10.0, 10.0], [10.5, 320.25], [490.0, 640.0]]))
	offsets = np.concatenate([[0], np.cumsum([len(r) for r in rings])]).astype(np.int64)
	warped = odw.transform_rings(grid.transformer, rings)
	page = synth.text_page(rng, size)
	labels = synth.label_map(rng, (320, 416), 4)
	return dict(
		size=np.array(size), nodes=nodes, ring_points=np.concatenate(rings), ring_offsets=offsets,
		ring_warped=np.concatenate(warped), page=page, labels=labels,
		page_dewarped=odw.dewarped(page, grid),
		labels_dewarped=odw.dewarp_image(size, labels, grid, ocv.INTER_NEAREST))


def main():
	build_oracle.build()
	for name, make in (
			("field", field_fixture), ("grid", grid_fixture), ("dense", dense_fixture),
			("remap", remap_fixture), ("transformer", transformer_fixture)):
		data = make()
		path = HERE / ("%s.npz" % name)
		np.savez_compressed(path, **data)
		print("%s: %d arrays, %.1f KB" % (path.name, len(data), path.stat().st_size / 1024))


if __name__ == "__main__":
	main()
