"""Compiles oracle/cv_oracle.c (the C restatement of cv2.resize / cv2.remap) with gcc.

    python -m oracle.build_oracle

Test infrastructure only; see oracle/__init__.py.
"""

import os
import shutil
import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
SRC = HERE / "cv_oracle.c"
LIB = HERE / "liborigami_cv_oracle.so"


def build(force=False):
	if not force and LIB.exists() and LIB.stat().st_mtime >= SRC.stat().st_mtime:
		return LIB
	gcc = shutil.which("gcc")
	if gcc is None:
		raise RuntimeError("gcc not found; the C oracle cannot be built")
	tmp = LIB.with_suffix(".so.tmp")
	cmd = [
		gcc, "-O2", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-fopenmp",
		"-Wall", str(SRC), "-o", str(tmp), "-lm"]
	r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
	if r.returncode != 0:
		raise RuntimeError("gcc failed:\n%s" % r.stdout.decode(errors="replace"))
	os.replace(tmp, LIB)
	return LIB


if __name__ == "__main__":
	print(build(force=True))
