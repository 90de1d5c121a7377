"""Repair of invalid polygons — what `geom.buffer(0)` does in the dewarp stage.

ref: origami/batch/detect/dewarp.py:35-38
        if not geom.is_valid:
            geom = geom.buffer(0)

shapely hands `buffer(0)` to GEOS (3.8 with the pinned Shapely 1.7.0), whose BufferOp with distance 0 does this
to a polygon (geos/operation/buffer: OffsetCurveSetBuilder, BufferSubgraph, PolygonBuilder):

  1. every ring becomes a curve labelled EXTERIOR | INTERIOR on its two sides — as if the ring were clockwise;
     rings that `Orientation::isCCW` finds counter-clockwise get the labels swapped, holes the opposite labels;
  2. all curves are noded against each other (split at every crossing, touch and overlap);
  3. the depth of every face of the resulting planar graph is found by walking outwards-in from depth 0 and adding
     +1 / -1 across each edge according to those labels — i.e. depth(p) = sum over rings of
     (+1 shell | -1 hole) * (ring's winding number around p, counted in the ring's OWN orientation as isCCW sees it);
  4. the result is the set of faces with depth >= 1, dissolved; rings that touch themselves at a node are split into
     minimal rings; shells come out clockwise, holes counter-clockwise.

Consequences worth knowing (and tested): a ring that touches itself at a vertex with both lobes wound the same way
gives a MultiPolygon of both lobes (the shapely manual's bow-tie example); a ring whose edges cross (lobes wound
opposite ways) keeps only the lobes wound like the ring's topmost corner — the well-known "buffer(0) drops half a
bow-tie".  A LineString has no area: its buffer(0) is POLYGON EMPTY.

The same steps are carried out here: exact rational predicates and intersection points (fractions.Fraction; the
points are then rounded to float64, as GEOS stores them), a half-edge face walk, depths by breadth-first search.
Contours needing repair are rare and small (tens to hundreds of vertices), so this is host code in plain Python.
PARITY UNPINNED against GEOS itself (not installable in the build image): vertex coordinates of crossings may
differ from GEOS' in the last bits, and the starting vertex of each output ring / the order of the parts of a
MultiPolygon are this implementation's own (both are irrelevant to the point set).
"""

import math
from fractions import Fraction

import numpy as np


class RepairError(RuntimeError):
	"""The noded arrangement turned out inconsistent (coordinates rounded onto each other); nothing is repaired."""


def _frac(p):
	return (Fraction(float(p[0])), Fraction(float(p[1])))


def _orient(p, q, r):
	"""Sign of (q - p) x (r - p), exact: a floating-point evaluation decides when its error bound allows
	(Shewchuk's orient2d filter), rational arithmetic otherwise.  Points are pairs of Fractions of doubles."""
	ax, ay = float(q[0]) - float(p[0]), float(q[1]) - float(p[1])
	bx, by = float(r[0]) - float(p[0]), float(r[1]) - float(p[1])
	l, m = ax * by, ay * bx
	det = l - m
	# the differences above are rounded too: 8 eps covers them and the products (coordinates are doubles)
	bound = 1.8e-15 * (abs(l) + abs(m))
	if det > bound:
		return 1
	if det < -bound:
		return -1
	v = (q[0] - p[0]) * (r[1] - p[1]) - (q[1] - p[1]) * (r[0] - p[0])
	return (v > 0) - (v < 0)


def _in_box(p, q, r):
	return min(p[0], q[0]) <= r[0] <= max(p[0], q[0]) and min(p[1], q[1]) <= r[1] <= max(p[1], q[1])


def is_ccw_geos(ring):
	"""geos::algorithm::Orientation::isCCW (GEOS 3.8 / JTS 1.16): the turn at the topmost vertex decides.
	ring: (n, 2) closed.  Degenerate rings (flat, fewer than 3 distinct points) are "not CCW"."""
	n = len(ring) - 1
	if n < 3:
		return False
	same = lambda a, b: a[0] == b[0] and a[1] == b[1]
	hi = 0
	for i in range(1, n + 1):
		if ring[i][1] > ring[hi][1]:
			hi = i
	ip = hi  # the distinct point before the highest one
	while True:
		ip = ip - 1 if ip > 0 else n
		if not same(ring[ip], ring[hi]) or ip == hi:
			break
	inx = hi  # ... and after it
	while True:
		inx = (inx + 1) % n
		if not same(ring[inx], ring[hi]) or inx == hi:
			break
	prev, top, nxt = ring[ip], ring[hi], ring[inx]
	if same(prev, top) or same(nxt, top) or same(prev, nxt):
		return False
	disc = _orient(_frac(prev), _frac(top), _frac(nxt))
	if disc == 0:
		return prev[0] > nxt[0]
	return disc > 0


def _clean_ring(ring):
	"""Closed ring without repeated consecutive points, as float tuples; None if fewer than 3 distinct corners."""
	pts = []
	for x, y in np.asarray(ring, dtype=np.float64).reshape(-1, 2):
		p = (float(x), float(y))
		if not (math.isfinite(p[0]) and math.isfinite(p[1])):
			raise RepairError("non-finite coordinate")
		if not pts or pts[-1] != p:
			pts.append(p)
	if len(pts) >= 2 and pts[0] != pts[-1]:
		pts.append(pts[0])
	if len(pts) < 4:
		return None
	return pts


def _node(segments):
	"""segments: list of (p, q) float tuples.  Returns per segment the list of points (float tuples) at which it
	must be split: every point where another segment crosses or touches it, exact predicates."""
	m = len(segments)
	a = np.array([s[0] for s in segments])
	b = np.array([s[1] for s in segments])
	lo, hi = np.minimum(a, b), np.maximum(a, b)
	fa = [_frac(s[0]) for s in segments]
	fb = [_frac(s[1]) for s in segments]
	cuts = [[] for _ in range(m)]
	for i in range(m):
		# candidates: bounding boxes overlap (closed)
		cand = np.nonzero(
			(lo[i + 1:, 0] <= hi[i, 0]) & (hi[i + 1:, 0] >= lo[i, 0]) &
			(lo[i + 1:, 1] <= hi[i, 1]) & (hi[i + 1:, 1] >= lo[i, 1]))[0] + i + 1
		for j in cand:
			p, q, r, s = fa[i], fb[i], fa[j], fb[j]
			o1, o2 = _orient(p, q, r), _orient(p, q, s)
			o3, o4 = _orient(r, s, p), _orient(r, s, q)
			if o1 * o2 < 0 and o3 * o4 < 0:  # proper crossing: a new vertex
				d1 = (q[0] - p[0], q[1] - p[1])
				d2 = (s[0] - r[0], s[1] - r[1])
				den = d1[0] * d2[1] - d1[1] * d2[0]
				t = ((r[0] - p[0]) * d2[1] - (r[1] - p[1]) * d2[0]) / den
				x = (float(p[0] + t * d1[0]), float(p[1] + t * d1[1]))
				cuts[i].append(x)
				cuts[j].append(x)
				continue
			if o1 == 0 and _in_box(p, q, r):
				cuts[i].append(segments[j][0])
			if o2 == 0 and _in_box(p, q, s):
				cuts[i].append(segments[j][1])
			if o3 == 0 and _in_box(r, s, p):
				cuts[j].append(segments[i][0])
			if o4 == 0 and _in_box(r, s, q):
				cuts[j].append(segments[i][1])
	return cuts


def _winding(ring, p):
	"""Winding number (counter-clockwise positive) of the closed ring (float tuples) around p, which must not lie on
	the ring.  Exact."""
	fp = _frac(p)
	w = 0
	for k in range(len(ring) - 1):
		a, b = _frac(ring[k]), _frac(ring[k + 1])
		if a[1] <= fp[1]:
			if b[1] > fp[1] and _orient(a, b, fp) > 0:
				w += 1
		elif b[1] <= fp[1] and _orient(a, b, fp) < 0:
			w -= 1
	return w


def _signed_area(pts):
	s = 0.0
	x0, y0 = pts[0]
	for k in range(1, len(pts) - 1):
		s += (pts[k][0] - x0) * (pts[k + 1][1] - y0) - (pts[k + 1][0] - x0) * (pts[k][1] - y0)
	return s / 2.0


def _split_at_repeats(loop):
	"""A closed vertex-id loop [v0, v1, ..., v0] that may pass a vertex more than once -> simple loops."""
	out, stack, where = [], [], {}
	for v in loop[:-1]:
		if v in where:
			k = where[v]
			sub = stack[k:] + [v]
			for u in stack[k + 1:]:
				del where[u]
			del stack[k + 1:]
			if len(sub) >= 4:
				out.append(sub)
		else:
			where[v] = len(stack)
			stack.append(v)
	if len(stack) >= 3:
		out.append(stack + [stack[0]])
	return out


def buffer0_rings(rings):
	"""rings: [shell, hole, ...] as (n, 2) arrays (closed or not).  Returns the repaired area as a list of polygons,
	each [shell, hole, ...] of (n, 2) float64 arrays, closed, shells clockwise and holes counter-clockwise as GEOS'
	buffer writes them.  [] when nothing has depth >= 1."""
	return overlay([rings], lambda depth: depth[0] >= 1)


def intersection_rings(rings_a, rings_b):
	"""Area common to two polygons ([shell, hole, ...] each), as a list of polygons like buffer0_rings gives.
	What `a.intersection(b)` keeps of two areal geometries (GEOS OverlayOp: the faces of the noded arrangement
	that lie in both); ref: origami/batch/detect/compose.py:422 (`poly.intersection(box)`)."""
	return overlay([rings_a, rings_b], lambda depth: depth[0] >= 1 and depth[1] >= 1)


def overlay(groups, keep):
	"""groups: polygons as ring lists [shell, hole, ...]; every group keeps a depth of its own (see the file
	comment: +1 inside a shell, -1 inside a hole, rings counted in their own orientation).  The rings of all groups
	are noded together; the faces whose depth tuple satisfies `keep` are dissolved into polygons."""
	clean, sign = [], []
	for g, rings in enumerate(groups):
		for k, ring in enumerate(rings):
			pts = _clean_ring(ring)
			if pts is None:
				continue  # GEOS: a ring too short to enclose area adds no curve at distance 0
			clean.append(pts)
			unit = (1 if k == 0 else -1) * (1 if is_ccw_geos(pts) else -1)
			sign.append(tuple(unit if j == g else 0 for j in range(len(groups))))
	if not clean:
		return []
	zero = (0,) * len(groups)
	add = lambda x, y: tuple(a + b for a, b in zip(x, y))
	sub = lambda x, y: tuple(a - b for a, b in zip(x, y))
	times = lambda x, m: tuple(a * m for a in x)

	# ---- noding --------------------------------------------------------------------------------------------
	segments, owner = [], []
	for k, pts in enumerate(clean):
		for i in range(len(pts) - 1):
			segments.append((pts[i], pts[i + 1]))
			owner.append(k)
	cuts = _node(segments)
	vid, coords = {}, []

	def vertex(p):
		if p not in vid:
			vid[p] = len(coords)
			coords.append(p)
		return vid[p]

	delta = {}  # (a, b), a < b -> depth(left of a->b) - depth(right of a->b)
	ring_vertex = {}
	for (p, q), k, extra in zip(segments, owner, cuts):
		dx, dy = q[0] - p[0], q[1] - p[1]
		chain = sorted(set(extra) | {p, q}, key=lambda c: (c[0] - p[0]) * dx + (c[1] - p[1]) * dy)
		if chain[0] != p or chain[-1] != q:
			raise RepairError("a split point fell outside its segment")
		ids = [vertex(c) for c in chain]
		ring_vertex.setdefault(k, ids[0])
		for u, v in zip(ids[:-1], ids[1:]):
			if u == v:
				continue
			# the winding number left of a directed ring segment is one higher than right of it
			if u < v:
				delta[(u, v)] = add(delta.get((u, v), zero), sign[k])
			else:
				delta[(v, u)] = sub(delta.get((v, u), zero), sign[k])

	# ---- half-edges, faces ---------------------------------------------------------------------------------
	out_edges = {}
	for (u, v) in delta:
		out_edges.setdefault(u, []).append(v)
		out_edges.setdefault(v, []).append(u)
	for u, nbrs in out_edges.items():
		ux, uy = coords[u]
		nbrs.sort(key=lambda v: math.atan2(coords[v][1] - uy, coords[v][0] - ux))
	pos = {(u, v): i for u, nbrs in out_edges.items() for i, v in enumerate(nbrs)}

	def nxt(u, v):
		"""The half-edge after u->v on the boundary of the face to its left: at v, the first edge clockwise from v->u."""
		nbrs = out_edges[v]
		return v, nbrs[(pos[(v, u)] - 1) % len(nbrs)]

	face_of, faces = {}, []
	for u, nbrs in out_edges.items():
		for v in nbrs:
			if (u, v) in face_of:
				continue
			f = len(faces)
			loop, h = [], (u, v)
			while h not in face_of:
				face_of[h] = f
				loop.append(h)
				h = nxt(*h)
			if h != (u, v):
				raise RepairError("face walk did not close")
			faces.append(loop)

	# connected components of the arrangement and the unbounded face of each
	comp = {}
	for start in out_edges:
		if start in comp:
			continue
		comp[start] = start
		todo = [start]
		while todo:
			u = todo.pop()
			for v in out_edges[u]:
				if v not in comp:
					comp[v] = start
					todo.append(v)
	n_comp = len(set(comp.values()))
	if len(coords) - len(delta) + len(faces) != 2 * n_comp:  # V - E + F = 2 per component (each has its own outer face)
		raise RepairError("arrangement violates Euler's formula (V=%d E=%d F=%d C=%d)" % (
			len(coords), len(delta), len(faces), n_comp))
	area = [_signed_area([coords[h[0]] for h in loop] + [coords[loop[0][0]]]) for loop in faces]
	outer = {}
	for f, loop in enumerate(faces):
		c = comp[loop[0][0]]
		if c not in outer or area[f] < area[outer[c]]:
			outer[c] = f
	comp_of_ring = {k: comp[v] for k, v in ring_vertex.items()}

	depth = {}
	for c, f0 in outer.items():
		if area[f0] > 0:
			raise RepairError("no clockwise outer face")
		p = coords[faces[f0][0][0]]
		base = zero
		for k in range(len(clean)):
			if comp_of_ring[k] != c:
				base = add(base, times(sign[k], _winding(clean[k], p)))
		depth[f0] = base
		todo = [f0]
		while todo:
			f = todo.pop()
			for (u, v) in faces[f]:
				g = face_of[(v, u)]
				d = delta[(u, v)] if u < v else times(delta[(v, u)], -1)  # depth(left of u->v) - depth(right)
				if g not in depth:
					depth[g] = sub(depth[f], d)
					todo.append(g)
				elif depth[g] != sub(depth[f], d):
					raise RepairError("inconsistent face depths")

	# ---- dissolve the faces to keep (buffer(0): depth >= 1) ------------------------------------------------------
	inside = [bool(keep(depth[f])) for f in range(len(faces))]
	region = list(range(len(faces)))  # union-find over inside faces that share an edge

	def find(f):
		while region[f] != f:
			region[f] = region[region[f]]
			f = region[f]
		return f

	# The walk gives every connected component an outer face of its own; in the plane that face IS the face of
	# another component the component lies in (a hole ring floating inside its shell), or the unbounded one.
	if n_comp > 1:
		loops_xy = [[coords[h[0]] for h in loop] + [coords[loop[0][0]]] for loop in faces]
		for c, f0 in outer.items():
			p = coords[faces[f0][0][0]]
			best = None
			for f, loop in enumerate(faces):
				if comp[loop[0][0]] == c or f == outer[comp[loop[0][0]]]:
					continue
				if (best is None or area[f] < area[best]) and _winding(loops_xy[f], p) != 0:
					best = f
			if best is not None:
				if depth[best] != depth[f0]:
					raise RepairError("nested components disagree on the depth of their common face")
				region[find(f0)] = find(best)

	boundary = set()
	for (u, v) in delta:
		for h in ((u, v), (v, u)):
			f, g = face_of[h], face_of[(h[1], h[0])]
			if inside[f] and inside[g]:
				region[find(f)] = find(g)
			elif inside[f]:
				boundary.add(h)  # the repaired area lies to the left of h
	b_out = {}
	for (u, v) in boundary:
		b_out.setdefault(u, []).append(v)

	def b_next(u, v):
		nbrs = out_edges[v]
		i = pos[(v, u)]
		for step in range(1, len(nbrs) + 1):  # clockwise from v->u: through the inside, up to its boundary
			w = nbrs[(i - step) % len(nbrs)]
			if (v, w) in boundary:
				return v, w
		raise RepairError("open boundary")

	seen, polygons = set(), {}
	for h0 in sorted(boundary):
		if h0 in seen:
			continue
		loop, h = [h0[0]], h0
		while h not in seen:
			seen.add(h)
			loop.append(h[1])
			h = b_next(*h)
		if h != h0:
			raise RepairError("boundary walk did not close")
		r = find(face_of[h0])
		for sub in _split_at_repeats(loop):
			pts = [coords[v] for v in sub]
			a = _signed_area(pts)
			if a == 0:
				continue
			polygons.setdefault(r, []).append((a, pts))
	result = []
	for r in sorted(polygons, key=lambda r: min(p for _, pts in polygons[r] for p in pts)):
		loops = sorted(polygons[r], key=lambda t: -t[0])  # the shell is the one counter-clockwise loop
		if loops[0][0] <= 0 or (len(loops) > 1 and loops[1][0] > 0):
			raise RepairError("a region without exactly one outer ring")
		# GEOS writes buffer shells clockwise and holes counter-clockwise: reverse both
		result.append([np.array(pts[::-1], dtype=np.float64) for _, pts in loops])
	return result
