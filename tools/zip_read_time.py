#!/usr/bin/env python3
"""Reading a page's contours archive (hundreds of members of a few hundred bytes): ZipReader.items() (one call into
ogd_inflate_many) against a zlib.decompress + zlib.crc32 loop over the same members, and zipfile.  Host only."""
import shutil
import sys
import tempfile
import time
import zipfile
import zlib
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import synth  # noqa: E402
from origami_amd.batch.core import fastzip  # noqa: E402

root = Path(tempfile.mkdtemp(prefix="ogd_z_"))
rng = np.random.default_rng(1)
synth.write_page_dir(root, "page0000", rng, size=(6000, 8000), n_h=1500, n_v=2000, n_regions=400)
path = sorted(root.rglob("contours*.zip"))[0]
data = path.read_bytes()
members = fastzip._scan(data)
print("%d members, %d -> %d bytes" % (len(members), sum(m[3] for m in members), sum(m[4] for m in members)))


def best(fn, n=20):
	t = 1e9
	for _ in range(n):
		t0 = time.perf_counter()
		fn()
		t = min(t, time.perf_counter() - t0)
	return 1e3 * t


def loop():
	for name, method, crc, csize, size, at in members:
		d = zlib.decompress(data[at:at + csize], -15, size or 1)
		assert zlib.crc32(d) == crc


print("directory scan               %.3f ms" % best(lambda: fastzip._scan(data)))
print("ZipReader(data).items()      %.3f ms" % best(lambda: list(fastzip.ZipReader(data).items())))
print("zlib + crc32 loop (no scan)  %.3f ms" % best(loop))
zf = zipfile.ZipFile(path)
print("zipfile read of all members  %.3f ms" % best(lambda: [zf.read(i) for i in zf.infolist()], 5))
shutil.rmtree(root, ignore_errors=True)
