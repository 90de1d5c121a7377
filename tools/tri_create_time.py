import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, scipy.spatial
from origami_amd import _shim, synth
_shim.init(0)
rng = np.random.default_rng(0)
for n in (2000, 80000):
    pts = rng.uniform(0, 6000, (n, 2)); vals = rng.normal(size=(n, 2))
    tri = scipy.spatial.Delaunay(pts)
    for rep in range(4):
        t0 = time.perf_counter(); T = _shim.Triangulation(pts, tri.simplices, vals, tri.transform); t1 = time.perf_counter(); T.close(); t2 = time.perf_counter()
        print("n=%d nt=%d create %.2f ms destroy %.2f ms" % (n, len(tri.simplices), 1e3*(t1-t0), 1e3*(t2-t1)))
        t0 = time.perf_counter(); T = _shim.Triangulation(pts, tri.simplices, vals, scipy_transform=True); t1 = time.perf_counter(); T.close()
        print("   transform on the device: create %.2f ms" % (1e3*(t1-t0)))
