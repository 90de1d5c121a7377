"""Seeded synthetic inputs of the dewarp path: warp model, flow samples, text-like
pages, label maps, contour polygons and whole `<page>.out/` directories.  Used by the tests, the
smoke run and bench.py; there is no dataset on the GPU box."""

import io
import json
import zipfile

import numpy as np
import PIL.Image

from .dewarp import Geometry, Samples
from .geometry import Geom, to_wkt


class Warp:
	"""phi_h(x, y) = A sin(2 pi f1 y / H + x / W);  phi_v = pi/2 + B sin(2 pi f2 x / W + y / H)."""

	def __init__(self, size, A=0.03, B=0.03, f1=1, f2=1):
		self.size = size
		self.A, self.B, self.f1, self.f2 = A, B, f1, f2

	@staticmethod
	def random(rng, size, heavy=False):
		if heavy:
			return Warp(size, 0.12, 0.12, int(rng.integers(1, 3)), int(rng.integers(1, 3)))
		return Warp(
			size, float(rng.uniform(0.01, 0.04)), float(rng.uniform(0.01, 0.04)),
			int(rng.integers(1, 3)), int(rng.integers(1, 3)))

	def phi_h(self, x, y):
		W, H = self.size
		return self.A * np.sin(2 * np.pi * y / H * self.f1 + x / W)

	def phi_v(self, x, y):
		W, H = self.size
		return np.pi / 2 + self.B * np.sin(2 * np.pi * x / W * self.f2 + y / H)

	def node_grid(self, cell, margin=1):
		"""A node grid of the shape GridFactory produces (page + border nodes), built in closed form:
		rows follow phi_h, columns follow phi_v to first order.  For benchmarks, where marching
		64 broadsheet grids on the host would only add set-up time."""
		W, H = self.size
		w = int(np.ceil(W / cell)) + 1 + 2 * margin
		h = int(np.ceil(H / cell)) + 1 + 2 * margin
		gy, gx = np.mgrid[0:h, 0:w].astype(np.float64)
		X = (gx - margin) * cell
		Y = (gy - margin) * cell
		# integrate the small-angle fields: dy/dx = phi_h along rows, dx/dy = -(phi_v - pi/2) along columns
		y = Y + self.A * W * (np.cos(2 * np.pi * Y / H * self.f1) - np.cos(2 * np.pi * Y / H * self.f1 + X / W))
		x = X - self.B * H * (np.cos(2 * np.pi * X / W * self.f2) - np.cos(2 * np.pi * X / W * self.f2 + Y / H))
		return np.stack([x, y], axis=-1).astype(np.float32)


def flow_samples(rng, warp, n_h, n_v, noise=0.002):
	W, H = warp.size
	geometry = Geometry(W, H)
	h, v = Samples(geometry), Samples(geometry)
	for samples, n, phi in ((h, n_h, warp.phi_h), (v, n_v, warp.phi_v)):
		pts = rng.uniform([0, 0], [W, H], size=(n, 2))
		val = phi(pts[:, 0], pts[:, 1]) + rng.normal(0, noise, size=n)
		samples.extend(list(pts), list(val))
	return h, v


def text_page(rng, size, channels=1):
	"""White-ish background (235 +- 5) with dark 1-3 px strokes in line-like runs."""
	W, H = size
	page = rng.normal(235, 5, size=(H, W)).clip(0, 255).astype(np.uint8)
	line_h = 40
	for y0 in range(60, H - 60, line_h):
		x = 80
		while x < W - 80:
			run = int(rng.integers(20, 160))
			t = int(rng.integers(1, 4))
			yy = y0 + int(rng.integers(0, 12))
			page[yy:yy + t, x:min(x + run, W - 80)] = rng.integers(0, 40)
			if rng.random() < 0.3:
				xs = x + int(rng.integers(0, run))
				page[y0:y0 + 18, xs:xs + t] = rng.integers(0, 40)
			x += run + int(rng.integers(8, 30))
	if channels == 1:
		return page
	tint = rng.integers(-12, 13, size=(1, 1, channels))
	return (page[:, :, None].astype(np.int16) + tint).clip(0, 255).astype(np.uint8)


def label_map(rng, size, n_classes=4, blocks=24):
	"""Blocky uint8 label map with values 0 .. n_classes-1."""
	W, H = size
	coarse = rng.integers(0, n_classes, size=(blocks, blocks), dtype=np.uint8)
	ys = np.minimum((np.arange(H) * blocks) // H, blocks - 1)
	xs = np.minimum((np.arange(W) * blocks) // W, blocks - 1)
	return np.ascontiguousarray(coarse[np.ix_(ys, xs)])


def polygon_rings(rng, size, n, mu=48.0, sigma=0.6, lo=4, hi=400, integer_scale=None, valid=True):
	"""n closed star-shaped rings; vertex counts ~ clip(lognormal(ln mu, sigma), lo, hi).  With
	integer_scale the vertices are integer label-map pixels times that scale, as contours that
	come from cv2.findContours are (ref: origami/core/contours.py:59-73, page.py:44-62)."""
	W, H = size
	rings = []
	while len(rings) < n:
		k = int(np.clip(rng.lognormal(np.log(mu), sigma), lo, hi))
		cx, cy = rng.uniform(0.05 * W, 0.95 * W), rng.uniform(0.05 * H, 0.95 * H)
		rad = rng.uniform(0.01, 0.05) * min(W, H)
		ang = np.sort(rng.uniform(0, 2 * np.pi, size=k))
		r = rad * rng.uniform(0.6, 1.0, size=k)
		pts = np.stack([cx + r * np.cos(ang), cy + r * np.sin(ang)], axis=1)
		pts[:, 0] = pts[:, 0].clip(0, W - 1)
		pts[:, 1] = pts[:, 1].clip(0, H - 1)
		if integer_scale is not None:
			pts = np.round(pts / integer_scale) * integer_scale
		ring = np.vstack([pts, pts[:1]])
		# contour tracing yields simple rings; rounding and clipping a star can fold it onto itself: draw again
		if valid and not Geom("Polygon", [ring]).is_valid:
			continue
		rings.append(ring)
	return rings


def write_page_dir(root, name, rng, size=(2400, 3400), n_h=300, n_v=500, n_regions=60,
		n_separators=12, warp=None, label_size=(1280, 2400), suffix=".png"):
	"""Writes `<root>/<name>.png` (or `.jpg`: the reference's usual page format) and the `<name>.out/` inputs
	of the dewarp stage: segment.zip, contours.0.zip, flow.zip."""
	page_path = root / (name + suffix)
	PIL.Image.fromarray(text_page(rng, size)).save(page_path)
	write_page_inputs(page_path, rng, size, n_h, n_v, n_regions, n_separators, warp, label_size)
	return page_path


def write_page_inputs(page_path, rng, size=(2400, 3400), n_h=300, n_v=500, n_regions=60,
		n_separators=12, warp=None, label_size=(1280, 2400)):
	"""The `<name>.out/` inputs of the dewarp stage for an existing page image of `size`."""
	W, H = size
	warp = warp or Warp.random(rng, size)
	out = page_path.with_suffix(".out")
	out.mkdir(exist_ok=True)

	with zipfile.ZipFile(out / "segment.zip", "w", zipfile.ZIP_DEFLATED) as zf:
		for pname, ptype, classes in (("regions", "REGION", ["BACKGROUND", "TEXT"]), ("separators", "SEPARATOR", ["BACKGROUND", "V"])):
			im = PIL.Image.fromarray(label_map(rng, label_size, len(classes)), "P")
			data = io.BytesIO()
			im.save(data, format="PNG")
			zf.writestr("%s.png" % pname, data.getvalue())
			zf.writestr("%s.json" % pname, json.dumps(dict(
				name=pname, type=ptype, classes=dict((c, i) for i, c in enumerate(classes)))))

	scale = W / label_size[0]
	with zipfile.ZipFile(out / "contours.0.zip", "w", zipfile.ZIP_DEFLATED) as zf:
		zf.writestr("meta.json", json.dumps(dict(version=2, predictions=[
			dict(name="regions", type="REGION"), dict(name="separators", type="SEPARATOR")])))
		for i, ring in enumerate(polygon_rings(rng, size, n_regions, mu=20, lo=8, hi=40, integer_scale=scale)):
			zf.writestr("regions/TEXT/%d.wkt" % i, to_wkt(Geom("Polygon", [ring])))
		widths = []
		for i in range(n_separators):
			x = rng.uniform(0.1 * W, 0.9 * W)
			y0, y1 = sorted(rng.uniform(0.05 * H, 0.95 * H, size=2))
			pts = np.stack([np.full(6, x) + rng.normal(0, 2, 6), np.linspace(y0, y1, 6)], axis=1)
			zf.writestr("separators/V/%d.wkt" % i, to_wkt(Geom("LineString", [pts])))
			widths.append(float(rng.uniform(2, 6)))
		zf.writestr("separators/V/meta.json", json.dumps(dict(width=widths)))

	h, v = flow_samples(rng, warp, n_h, n_v)
	with zipfile.ZipFile(out / "flow.zip", "w", zipfile.ZIP_DEFLATED) as zf:
		h.save(zf, "h")
		v.save(zf, "v")
	return page_path


def write_reliable_stage(page_path, rng, nodes, cell, n_lines=30, n_table_lines=4):
	"""Adds the inputs of the line stages (OCR, line export; Stage.RELIABLE) to `<name>.out/`:
	dewarp.zip (the given node grid), contours.3.zip, lines.3.zip, tables.json.
	Returns {line path: (p, right, up)} in dewarped pixel coordinates."""
	out = page_path.with_suffix(".out")
	out.mkdir(exist_ok=True)
	h, w = nodes.shape[:2]
	Wd, Hd = w * cell, h * cell

	data = io.BytesIO()
	np.save(data, nodes.astype(np.float32), allow_pickle=False)
	with zipfile.ZipFile(out / "dewarp.zip", "w", zipfile.ZIP_DEFLATED) as zf:
		zf.writestr("data.npy", data.getvalue())
		zf.writestr("meta.json", json.dumps(dict(version=1, cell=cell, shape=list(nodes.shape))))
	if not (out / "segment.zip").exists():
		with zipfile.ZipFile(out / "segment.zip", "w") as zf:
			zf.writestr("regions.json", json.dumps(dict(
				name="regions", type="REGION", classes=dict(BACKGROUND=0, TEXT=1, TABULAR=2))))

	def geometry_of(k):
		length = rng.uniform(0.15 * Wd, 0.8 * Wd)
		height = rng.uniform(18, 44)
		ang = rng.normal(0, 0.015)
		x = rng.uniform(5, Wd - length - 5)
		y = rng.uniform(height + 5, Hd - 5)
		return ([x, y], [length * np.cos(ang), length * np.sin(ang)], [height * np.sin(ang), -height * np.cos(ang)])

	lines = {}
	blocks = {}
	per_block = 5
	for k in range(n_lines):
		b = ("regions", "TEXT", str(k // per_block))
		lines[b + (str(k % per_block),)] = geometry_of(k)
		blocks[b] = None
	# one table block whose rows are its lines; LineRewriter cuts every row at the block's column xs
	columns = {}
	if n_table_lines:
		b = ("regions", "TABULAR", "7.1.1.1")
		blocks[b] = None
		tx, tlen = 0.1 * Wd, 0.7 * Wd
		for k in range(n_table_lines):
			y = (k + 1) * Hd / (n_table_lines + 2)
			skew = rng.normal(0, 2)
			lines[b + (str(k),)] = ([tx, y], [tlen, skew], [0.0, -rng.uniform(20, 40)])
		columns["/".join(b)] = [tx + 0.3 * tlen, tx + 0.7 * tlen]

	with zipfile.ZipFile(out / "contours.3.zip", "w", zipfile.ZIP_DEFLATED) as zf:
		zf.writestr("meta.json", json.dumps(dict(version=2, predictions=[dict(name="regions", type="REGION")])))
		for b in blocks:
			ring = np.array([[1, 1], [Wd - 2, 1], [Wd - 2, Hd - 2], [1, Hd - 2], [1, 1]], dtype=np.float64)
			zf.writestr("/".join(b) + ".wkt", to_wkt(Geom("Polygon", [ring])))
	with zipfile.ZipFile(out / "lines.3.zip", "w", zipfile.ZIP_DEFLATED) as zf:
		zf.writestr("meta.json", json.dumps(dict(min_confidence=0.5)))
		for i, (path, (p, right, up)) in enumerate(lines.items()):
			conf = 0.3 if i == 1 else 0.9  # one line below min_confidence is skipped by LineExtractor
			zf.writestr("/".join(path) + ".json", json.dumps(dict(
				p=p, right=right, up=up, wkt=None, confidence={"regions/TEXT": conf, "regions/BACKGROUND": 0.95},
				tesseract_data=dict(baseline=[p, [p[0] + right[0], p[1] + right[1]]], descent=4, ascent=12, height=20))))
	(out / "tables.json").write_text(json.dumps(dict(version=1, columns=columns, dividers={})))
	return lines


def colour_page(rng, size):
	"""A scan in colour, (H, W, 3) uint8: yellowed paper, dark text, a coloured stamp, sensor noise."""
	W, H = size
	text = text_page(rng, size).astype(np.float32) * (1.0 / 255.0)
	paper = np.array([236, 222, 190], np.float32)
	ink = np.array([40, 32, 36], np.float32)
	rgb = ink + (paper - ink) * text[:, :, None]
	yy, xx = np.ogrid[0:H, 0:W]
	stamp = ((yy - H // 3) ** 2 + (xx - W // 2) ** 2) < (min(W, H) // 6) ** 2
	rgb[stamp] = 0.5 * rgb[stamp] + 0.5 * np.array([200, 30, 40], np.float32)
	rgb += rng.normal(0, 3, size=rgb.shape).astype(np.float32)
	return np.clip(rgb, 0, 255).astype(np.uint8)
