#!/usr/bin/env python3
"""Decode time of a broadsheet page stored as PNG: PIL against ogd_png_decode_u8 (host code, no GPU involved),
and zlib against libogdewarp's inflate on the same IDAT stream.  usage: png_bench.py [W H]"""
import io
import struct
import sys
import time
import zlib
from pathlib import Path

import numpy as np
import PIL.Image

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from origami_amd import _shim, synth  # noqa: E402


def best(f, n=3):
	t = []
	for _ in range(n):
		t0 = time.perf_counter()
		f()
		t.append(time.perf_counter() - t0)
	return min(t) * 1e3


def main():
	W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (6000, 8000)
	rng = np.random.default_rng(1)
	for name, pixels in (("text page, gray", synth.text_page(rng, (W, H))), ("colour page, RGB", synth.colour_page(rng, (W // 2, H // 2)))):
		b = io.BytesIO()
		PIL.Image.fromarray(pixels).save(b, format="PNG")
		data = b.getvalue()
		pos, idat = 8, []
		while pos < len(data):
			n, kind = struct.unpack(">I4s", data[pos:pos + 8])
			if kind == b"IDAT":
				idat.append(data[pos + 8:pos + 8 + n])
			pos += 12 + n
		z = b"".join(idat)
		raw = zlib.decompress(z)
		assert np.array_equal(_shim.png_decode(data), pixels)
		print("%s %s: file %.1f MB, %.1f MB of filtered rows" % (name, pixels.shape, len(data) / 1e6, len(raw) / 1e6))
		print("  PIL.Image.open + load      %7.1f ms" % best(lambda: PIL.Image.open(io.BytesIO(data)).load()))
		print("  ogd_png_decode_u8          %7.1f ms" % best(lambda: _shim.png_decode(data)))
		print("  zlib.decompress (IDAT)     %7.1f ms" % best(lambda: zlib.decompress(z)))
		print("  ogd_inflate_zlib (IDAT)    %7.1f ms" % best(lambda: _shim.inflate_zlib(z, len(raw))))


if __name__ == "__main__":
	main()
