"""The product's overlapped page pipeline (origami_amd/pipeline.py: pinned host -> H2D -> fused dewarp + K3 -> D2H
on three streams, N device slots) against the synchronous host entry points and the oracle — bit for bit, over
several ring cycles, with pages of different sizes, channel counts and interpolations in one run."""

import numpy as np
import pytest

from oracle import cv as ocv
from origami_amd import pinned, synth
from origami_amd.dewarp import Grid, INTER_LINEAR, INTER_NEAREST
from origami_amd.pipeline import DeviceArray, Event, PageJob, PagePipeline, Stream, dewarp_pages

pytestmark = pytest.mark.gpu


def _page_and_grid(seed, size, cell, channels=1):
	rng = np.random.default_rng(seed)
	warp = synth.Warp.random(rng, size)
	nodes = warp.node_grid(cell)
	pixels = synth.text_page(rng, size, channels)
	return pixels, nodes


SIZES = [((500, 650), 25), ((333, 411), 8), ((900, 700), 25), ((640, 480), 16), ((257, 1023), 9)]


def test_pinned_arrays_are_numpy_arrays_and_blocks_are_reused(gpu):
	a = pinned.empty((1000, 700), np.uint8)
	assert pinned.is_pinned(a) and pinned.is_pinned(a[10:20, 5:]) and not pinned.is_pinned(np.empty(4))
	a[:] = 7
	assert a.sum() == 7 * 1000 * 700
	address = pinned.block_of(a).address
	del a
	b = pinned.empty((999, 701), np.uint8)  # similar size: the cached block
	assert pinned.block_of(b).address == address
	f = pinned.empty((10, 2), np.float64)
	assert f.dtype == np.float64 and f.shape == (10, 2) and f.flags.c_contiguous
	c = pinned.as_pinned(b)
	assert c is b
	d = pinned.as_pinned(np.arange(12, dtype=np.int64).reshape(3, 4)[:, ::2])
	assert pinned.is_pinned(d) and d.tolist() == [[0, 2], [4, 6], [8, 10]]
	del b, c, d, f
	assert pinned.trim() > 0


def test_pipeline_equals_the_host_entry_point_over_several_ring_cycles(gpu):
	"""11 jobs through 3 slots (3.7 ring cycles): sizes, cells, channel counts and interpolations change from job
	to job, so every slot's buffers grow and are re-used for smaller pages; input from pinned and from ordinary
	memory.  Every result must equal ogd_dewarp_u8 of the same page bit for bit (which the kernel tests pin
	against the oracle), and the first ones are checked against the oracle directly."""
	jobs, want = [], []
	for k in range(11):
		size, cell = SIZES[k % len(SIZES)]
		channels = (1, 3, 1, 4)[k % 4]
		pixels, nodes = _page_and_grid(100 + k, size, cell, channels)
		interp = INTER_NEAREST if k % 5 == 3 else INTER_LINEAR
		sx, sy = (np.float32(0.5), np.float32(0.75)) if k % 3 == 2 else (1.0, 1.0)
		src = pinned.as_pinned(pixels) if k % 2 == 0 else pixels
		jobs.append(PageJob(src, nodes, cell, sx=sx, sy=sy, interpolation=interp, tag=k))
		want.append(gpu.dewarp_u8(pixels, nodes, cell, sx=sx, sy=sy, interpolation=interp))
	with PagePipeline(slots=3) as pipe:
		results = list(pipe.map(jobs))
		assert pipe.jobs_done == len(jobs) and pipe.in_flight == 0
	assert [r.tag for r in results] == list(range(11))
	for r, w in zip(results, want):
		assert r.done and pinned.is_pinned(r.page)
		np.testing.assert_array_equal(r.page, w, err_msg="job %d" % r.tag)
	for k in (0, 1, 2):
		job = jobs[k]
		dense = ocv.dense_grid(np.asarray(job.nodes), job.cell)
		dense[:, :, 0] *= np.float32(job.sx)
		dense[:, :, 1] *= np.float32(job.sy)
		np.testing.assert_array_equal(results[k].page, ocv.remap_u8(np.asarray(job.pixels), dense, job.interpolation))


def test_pipeline_warps_the_contours_of_each_page_in_the_same_job(gpu):
	"""K3 on the kernel stream of the job: the rings of each page through that page's transformer; the result equals
	Transformer.transform_table (one launch on the null stream) bit for bit, including rings longer than the 64-vertex
	pieces the pipeline cuts them into, a page without rings and rings leaving the grid's bounding box."""
	pages = []
	for k in range(5):
		size, cell = SIZES[k]
		pixels, nodes = _page_and_grid(300 + k, size, cell)
		grid = Grid(nodes, cell)
		rng = np.random.default_rng(k)
		rings = synth.polygon_rings(rng, size, 0 if k == 2 else 9, mu=30 if k != 3 else 150)
		if rings:
			rings[0] = rings[0] * 1.4 - 50.0  # partly outside the page and the node grid
		flat = np.concatenate(rings) if rings else np.empty((0, 2))
		offsets = np.zeros(len(rings) + 1, dtype=np.int64)
		np.cumsum([len(r) for r in rings], out=offsets[1:])
		pages.append((pixels, grid, (flat, offsets)))
	assert max(np.diff(pages[3][2][1])) > 64
	results = list(dewarp_pages(pages, slots=2))
	for r, (pixels, grid, (flat, offsets)) in zip(results, pages):
		np.testing.assert_array_equal(r.page, gpu.dewarp_u8(pixels, grid.points("sample"), grid.resolution))
		want = grid.transformer.transform_table(flat, offsets)
		assert r.rings.shape == want.shape
		np.testing.assert_array_equal(r.rings, want)


def test_pipeline_takes_device_resident_pixels_and_borrowed_streams(gpu):
	"""Pixels and node grid already on the device (DeviceArray) skip the upload; streams handed in are used, not
	replaced (a process must stay within its four hardware queues)."""
	size, cell = SIZES[0]
	pixels, nodes = _page_and_grid(7, size, cell)
	want = gpu.dewarp_u8(pixels, nodes, cell)
	streams = (Stream(), Stream(), Stream())
	d_pixels, d_nodes = DeviceArray.upload(pixels), DeviceArray.upload(nodes)
	with PagePipeline(slots=2, streams=streams) as pipe:
		results = [pipe.submit(PageJob(d_pixels, d_nodes, cell, tag=k)) for k in range(5)]
		assert pipe.in_flight <= 2
		for r in results:
			pipe.wait(r)
			np.testing.assert_array_equal(r.page, want)
	for s in streams:
		assert s.ptr is not None  # still the caller's
		s.close()


def test_events_time_a_stream(gpu):
	s = Stream()
	a, b = Event(timing=True), Event(timing=True)
	size, cell = SIZES[2]
	pixels, nodes = _page_and_grid(9, size, cell)
	with PagePipeline(slots=1, streams=(s, s, s)) as pipe:
		a.record(s)
		r = pipe.submit(PageJob(pinned.as_pinned(pixels), nodes, cell))
		b.record(s)
		b.synchronize()
		assert a.done() and b.done() and 0.0 < a.elapsed_ms(b) < 1000.0
		pipe.wait(r)
	np.testing.assert_array_equal(r.page, gpu.dewarp_u8(pixels, nodes, cell))
	s.close()


def test_page_decodes_into_pinned_memory_and_dewarps_from_it(gpu, tmp_path):
	"""Page: a PNG scan decoded by libogdewarp's inflate lands in pinned memory; Dewarper.dewarped comes back in
	pinned memory; same pixels as PIL + the oracle."""
	import PIL.Image
	from origami_amd.page import Page
	size, cell = SIZES[0]
	pixels, nodes = _page_and_grid(11, size, cell)
	path = tmp_path / "p.png"
	PIL.Image.fromarray(pixels).save(path)
	page = Page(path, Grid(nodes, cell))
	page.prefetch()
	assert page._pixels is not None and pinned.is_pinned(page._pixels)
	np.testing.assert_array_equal(np.asarray(page.warped), pixels)
	np.testing.assert_array_equal(np.asarray(page.dewarped), ocv.remap_u8(pixels, ocv.dense_grid(nodes, cell)))
