"""Oracle for the JPEG file of a dewarped page — TEST INFRASTRUCTURE ONLY (tests/, smoke, tools).

ref: origami/batch/core/io.py:606-610: `image.save(str(path))` of the "L" image `page.dewarped` to
annotation.<name>.jpg.  The arithmetic behind that call is PIL's JPEG plugin on libjpeg(-turbo), a third-party
library the reference depends on and that IS importable here and on the GPU box — so the oracle is that
library itself, called exactly as the reference calls it (no restatement, parity pinned on the real thing)."""

import io

import numpy as np
import PIL.Image


def save_gray(pixels, **params):
	"""The bytes `PIL.Image.fromarray(pixels).save(path.jpg, **params)` writes (default: quality 75)."""
	im = PIL.Image.fromarray(np.ascontiguousarray(pixels))
	assert im.mode == "L"
	buf = io.BytesIO()
	im.save(buf, format="JPEG", **params)
	return buf.getvalue()
