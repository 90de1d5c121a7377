"""Builds libogdewarp.so (HIP, gfx950 only) in-tree with hipcc.

    python -m origami_amd.build [--force]

The library is compiled here (hipcc cross-compiles without a GPU) and travels to the GPU box as a
built file; nothing is JIT-compiled at run time.
"""

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB_DIR = PKG / "lib"
LIB_PATH = LIB_DIR / "libogdewarp.so"

SOURCES = ["ogd_runtime.hip", "ogd_grid.hip", "ogd_remap.hip", "ogd_tri.hip", "ogd_lines.hip", "ogd_lattice.hip", "ogd_delaunay.hip", "ogd_png.hip", "ogd_deflate.hip", "ogd_march.hip", "ogd_wkt.hip", "ogd_jpeg.hip", "ogd_jpeg_dec.hip"]
HEADERS = ["ogd_common.h", "ogd_lawson.h", "ogd_inflate.h", "ogd_deflate.h", "ogd_checksum.h", "ogd_tri_dev.h", "ogd_resize_math.h", "ogd_grid_lines.h", "ogd_remap_math.h", "ogd_remap_lds.h", "../../include/ogdewarp.h"]

# -ffp-contract=off: the parity-critical fp32/fp64 chains must round exactly like the CPU libraries
# they replace (OpenCV resize, scipy interpnd); the kernels additionally spell those chains with
# __fmul_rn / __fadd_rn / __dmul_rn / __dadd_rn.
HIPCC_FLAGS = [
	"--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
	"-fno-gpu-sanitize", "-Wall", "-Wno-unused-function"]


def hipcc():
	exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
	if not Path(exe).exists():
		raise RuntimeError("hipcc not found; libogdewarp.so cannot be built")
	return exe


def _stale():
	if not LIB_PATH.exists():
		return True
	t = LIB_PATH.stat().st_mtime
	deps = [CSRC / s for s in SOURCES] + [(CSRC / h).resolve() for h in HEADERS] + [Path(__file__)]
	return any(d.stat().st_mtime > t for d in deps)


def build_library(force=False, verbose=False):
	"""Compile every .hip source for gfx950 and link libogdewarp.so; returns its path."""
	if not force and not _stale():
		return LIB_PATH
	LIB_DIR.mkdir(exist_ok=True)
	obj_dir = PKG / "build"
	obj_dir.mkdir(exist_ok=True)
	objs = []
	procs = []
	for src in SOURCES:
		obj = obj_dir / (Path(src).stem + ".o")
		# OGD_EXTRA_FLAGS: extra -D switches for kernel experiments (tools/); the shipped library is built without
		cmd = [hipcc()] + HIPCC_FLAGS + os.environ.get("OGD_EXTRA_FLAGS", "").split() + ["-c", str(CSRC / src), "-o", str(obj)]
		if verbose:
			print(" ".join(cmd), file=sys.stderr)
		procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
		objs.append(str(obj))
	for src, p in procs:
		out, _ = p.communicate()
		if p.returncode != 0:
			raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
		if verbose and out:
			print(out.decode(errors="replace"), file=sys.stderr)
	tmp = LIB_PATH.with_suffix(".so.tmp")
	cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-sanitize", "-o", str(tmp)] + objs
	r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
	if r.returncode != 0:
		raise RuntimeError("link failed:\n%s" % r.stdout.decode(errors="replace"))
	os.replace(tmp, LIB_PATH)
	return LIB_PATH


if __name__ == "__main__":
	path = build_library(force="--force" in sys.argv, verbose=True)
	print(path)
