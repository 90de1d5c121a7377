"""The dewarp-path piece of the compose stage: page coordinates for PageXML.

ref: origami/batch/detect/compose.py:413-432 (`Document.rewarp`).  Regions and lines live in dewarped pixel
coordinates; PageXML wants them on the scanned (warped) page.  The reference sends one outline at a time through
`Grid.inverse` (a scipy RegularGridInterpolator over the dense grid, core/dewarp.py:561-576) and tidies the result
with shapely: an invalid outline becomes its convex hull, the outline is clipped to the page ("Page XML is very
picky about not specifying any negative coordinates"), and what is left must be one polygon — several parts are
replaced by their common convex hull, nothing left is an error.

Here all outlines of a page go through the inverse map in ONE launch (`ogd_grid_sample_bilinear`, evaluated from
the node grid: the dense grid is never built); validity, hull and clipping are host geometry
(origami_amd/geometry.py, origami_amd/repair.py:intersection_rings).  The rest of the compose stage (reading
order, OCR text, PageXML writing) is outside the dewarp path.
"""

import numpy as np

from ...geometry import Geom, convex_hull_ring
from ...repair import intersection_rings


def page_outline(warped_coords, page_size, original=None):
	"""What `rewarp` returns for one outline once it is in warped-page coordinates: the exterior ring of
	(valid outline or its convex hull) ∩ page, as an (n, 2) array, closed."""
	width, height = page_size
	ring = np.asarray(warped_coords, dtype=np.float64).reshape(-1, 2)
	if len(ring) and not np.array_equal(ring[0], ring[-1]):
		ring = np.vstack([ring, ring[:1]])  # shapely closes a Polygon's shell
	poly = Geom("Polygon", [ring])
	if not poly.is_valid:
		hull = convex_hull_ring(ring)
		poly = Geom("Polygon", [hull]) if hull is not None else Geom("Polygon", [])
	box = np.array([[0, 0], [width, 0], [width, height], [0, height], [0, 0]], dtype=np.float64)
	parts = [] if poly.is_empty else intersection_rings(poly.rings, [box])
	if not parts:
		raise RuntimeError("failed to rewarp coords %s as %s outside page" % (
			str(list(map(tuple, np.asarray(original if original is not None else warped_coords).tolist()))), poly.wkt))
	if len(parts) == 1:
		return parts[0][0]
	hull = convex_hull_ring(np.vstack([p[0] for p in parts]))
	if hull is None:
		raise RuntimeError("rewarped outline has no area inside the page")
	return hull


class Rewarper:
	"""`Document.rewarp` for a whole page: `many()` maps any number of outlines with one inverse-map launch."""

	def __init__(self, grid, page_size):
		self._grid = grid
		self._page_size = tuple(page_size)

	def __call__(self, coords):
		return self.many([coords])[0]

	def many(self, outlines):
		outlines = [np.asarray(c, dtype=np.float64).reshape(-1, 2) for c in outlines]
		if not outlines:
			return []
		ends = np.cumsum([len(c) for c in outlines])
		warped = self._grid.inverse(np.concatenate(outlines))
		return [
			page_outline(warped[b - len(c):b], self._page_size, original=c)
			for c, b in zip(outlines, ends)]
