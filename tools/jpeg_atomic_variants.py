#!/usr/bin/env python3
"""Rebuilds the variant libraries behind profiles/r02/c_jpeg_atomic_probe*.log ("the atomic in
the walk"): the decoder source of commit 465d566 (the round-1 state in which the fault was first seen) with the
restart counter's atomic placed in different ways inside the divergent decode loop, linked with the current
objects of every other file.  Run on the CPU box after `python -m origami_amd.build`; the GPU side is
tools/jpeg_atomic_probe.sh.

  C  atomicAdd(rst_check, 1) in the loop (no return value)            -> wrong pixels, different on every run
  D  as committed: counted in a register, one atomic after the loop   -> correct
  E  C built with -mllvm -amdgpu-atomic-optimizer-strategy=None       -> wrong pixels
  F  the atomic's return value is used (compiler waits for it)        -> correct
  I  a plain store in place of the atomic                             -> correct
  J  C followed by s_waitcnt 0                                        -> correct
"""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "origami_amd" / "csrc"
OUT = ROOT / "origami_amd" / "lib"
TMP = Path("/tmp/ogd_jpeg_variants")

IN_LOOP = '''            if (WRITE) {
#if VARIANT == 'F'
              { int old_ = atomicAdd(rst_check, 1); asm volatile("" ::"v"(old_)); }
#elif VARIANT == 'I'
              rst_check[2] = 1;
#elif VARIANT == 'J'
              atomicAdd(rst_check, 1); __builtin_amdgcn_s_waitcnt(0);
#elif VARIANT == 'C' || VARIANT == 'E'
              atomicAdd(rst_check, 1);
#endif
              ++taken;'''
AFTER_LOOP = '''#if VARIANT == 'D' || VARIANT == 'I'
    if (taken) atomicAdd(rst_check, taken);
#endif'''


def main():
	TMP.mkdir(exist_ok=True)
	src = subprocess.run(["git", "show", "465d566:origami_amd/csrc/ogd_jpeg_dec.hip"], cwd=ROOT, check=True,
		stdout=subprocess.PIPE).stdout.decode()
	a = "            if (WRITE) {\n              ++taken;"
	b = "    if (taken) atomicAdd(rst_check, taken);"
	assert a in src and b in src
	src = src.replace(a, IN_LOOP).replace(b, AFTER_LOOP)
	(TMP / "jd.hip").write_text(src)
	others = [str(p) for p in (ROOT / "origami_amd" / "build").glob("*.o") if p.name != "ogd_jpeg_dec.o"]
	flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-gpu-sanitize", "-I" + str(CSRC)]
	for v in sys.argv[1:] or list("CDEFIJ"):
		extra = ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"] if v == "E" else []
		obj = TMP / ("jd%s.o" % v)
		subprocess.run(["hipcc"] + flags + extra + ["-DVARIANT='%s'" % v, "-c", str(TMP / "jd.hip"), "-o", str(obj)], check=True)
		subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-sanitize", "-o",
			str(OUT / ("libogd_var%s.so" % v))] + others + [str(obj)], check=True)
		print("built", OUT / ("libogd_var%s.so" % v))


if __name__ == "__main__":
	main()
