#!/usr/bin/env python3
"""Where does the device march differ from the oracle's walk?  usage: march_debug.py seed a f  (GPU box)"""
import sys
from pathlib import Path
import numpy as np
import PIL.Image
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import dewarp as odw  # noqa: E402
from origami_amd import _shim, synth  # noqa: E402
from origami_amd.dewarp import GridFactory  # noqa: E402

seed, a, f = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
size = (1500, 2000)
_shim.init(0)
rng = np.random.default_rng(seed)
warp = synth.Warp(size, a, a, f, f)
h, v = synth.flow_samples(rng, warp, 500, 700)


class Page:
	warped = PIL.Image.new("L", size)


fac = GridFactory(Page, h, v, grid_res=25)
ofac = odw.GridFactory(size, odw.Samples(size, h.points, h.values), odw.Samples(size, v.points, v.values), grid_res=25)
gh = fac.grid_h
ogh = ofac.grid_h()
odw.extend_row_ends(ogh, 25)
print("grid_h shape", gh.shape, ogh.shape, "max diff inner", np.abs(gh[:, 1:-1] - ogh[:, 1:-1]).max())
core = fac._march()[1]
# the oracle's core (before border extension): re-run its loop
ohv_full = ofac.grid_hv()
hv_full = fac.grid_hv
print("full shapes", hv_full.shape, ohv_full.shape)
d = np.abs(hv_full - ohv_full).max(axis=-1)
print("max", d.max(), "n > 1e-3:", int((d > 1e-3).sum()))
idx = np.argwhere(d > 1e-3)
for gy, gx in idx[:40]:
	print(gy, gx, hv_full[gy, gx], ohv_full[gy, gx], d[gy, gx])
np.savez("gpurun_out/march_debug_%d.npz" % seed, gh=gh, ogh=ogh, hv=hv_full, ohv=ohv_full, core=core)
