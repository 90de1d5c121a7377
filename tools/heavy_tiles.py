#!/usr/bin/env python3
"""Which tiles of a heavily warped page (BASELINE.json configs[4]: A = B = 0.12 rad, marched grid) leave the LDS path of
K2 / the fused kernel?  Recomputes, in numpy, the window test of remap_u8_wave_kernel (ogd_remap_lds.h: wbytes > PF or
nrows > MAXROWS -> remap_tile_global) for every tile of the dense map.  usage: heavy_tiles.py [A]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, synth  # noqa: E402
from origami_amd.dewarp import Grid  # noqa: E402

A = float(sys.argv[1]) if len(sys.argv) > 1 else 0.12
W, H, CELL = 6000, 8000, 25
_shim.init(0)
rng = np.random.default_rng(4000)
warp = synth.Warp((W, H), A, A, 1, 2)
sh, sv = synth.flow_samples(rng, warp, 2000, 3000)


class _Page:
	@staticmethod
	def size(dewarped):
		return (W, H)


grid = Grid.create(_Page, sh, sv, grid_res=CELL)
nodes = grid.points("sample")
dense = _shim.grid_fill(nodes, CELL)
Hd, Wd = dense.shape[:2]
q = np.rint(dense.astype(np.float64) * 32).astype(np.int64)
cx, cy = np.clip(q[:, :, 0] >> 5, -2, W), np.clip(q[:, :, 1] >> 5, -2, H)
print("heavy grid: nodes %s, dense %dx%d; local scale |d map / d pixel| along rows: min %.3f max %.3f" % (
	nodes.shape, Wd, Hd, np.hypot(*np.diff(dense[::50], axis=1).transpose(2, 0, 1)).min(), np.hypot(*np.diff(dense[::50], axis=1).transpose(2, 0, 1)).max()))
for name, TW, TH, PF, MAXROWS in (("K2 gray 64x8 (sheared: + up to 35 columns)", 64 + 35, 8, 168, 24), ("K2 gray 64x8 unsheared", 64, 8, 88, 46), ("fused gray 32x32", 32, 32, 48, 85), ("K2 RGB 64x8", 64, 8, 264 // 3, 23), ("fused RGB 32x32", 32, 32, 144 // 3, 56)):
	nby, nbx = -(-Hd // TH), -(-Wd // TW)
	bad_w = bad_r = 0
	widths, rows = [], []
	for by in range(nby):
		ys = slice(by * TH, min((by + 1) * TH, Hd))
		for bx in range(nbx):
			xs = slice(bx * TW, min((bx + 1) * TW, Wd))
			x0, x1 = int(cx[ys, xs].min()), int(cx[ys, xs].max())
			y0, y1 = int(cy[ys, xs].min()), int(cy[ys, xs].max())
			wpx = x1 + 2 - (x0 & ~3)
			nr = y1 + 2 - y0
			widths.append(wpx)
			rows.append(nr)
			bad_w += wpx > PF
			bad_r += nr > MAXROWS and wpx <= PF
	n = nby * nbx
	print("%-44s tiles %6d  window width px: median %3d max %3d (pitch %3d) -> %5.1f %% too wide; rows: median %2d max %3d (budget %2d) -> %5.1f %% too tall" % (
		name, n, np.median(widths), max(widths), PF, 100.0 * bad_w / n, np.median(rows), max(rows), MAXROWS, 100.0 * bad_r / n))
