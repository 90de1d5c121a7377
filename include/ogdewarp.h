/*
 * ogdewarp.h — C ABI of libogdewarp.so: the MI355X (gfx950) dewarp hot path of Origami.
 *
 * The reference (poke1024/origami) has no FFI for this path; the seam is the set of third-party
 * calls inside origami/core/dewarp.py and origami/core/lingrid.py.  Every entry point below
 * replaces one of those call sites (cited as `ref:` = file:line under the reference tree).
 *
 * Conventions
 *  - plain pointers and sizes only; no torch / numpy types.
 *  - every function returns OGD_OK (0) or a negative ogd_status; the message of the last failure
 *    on the calling thread is returned by ogd_last_error().  Nothing here aborts the process.
 *  - "host" entry points take host buffers (C-contiguous unless a stride is passed), stage them
 *    through device memory owned by the library, and return when the result is in the host buffer.
 *  - "*_dev" entry points take DEVICE pointers plus a hipStream_t (passed as void*, NULL = the
 *    null stream) and only enqueue work: the caller synchronises.  They allocate nothing.
 *  - no HIP call happens before the first entry point is called (fork-safe: origami forks
 *    its pool workers, ref: origami/batch/core/processor.py:483).
 *  - all entry points are thread-safe; handles (ogd_tri) are immutable after creation.
 */
#ifndef OGDEWARP_H
#define OGDEWARP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ogd_status {
  OGD_OK = 0,
  OGD_ERR_INVALID = -1,  /* bad argument (shape, null pointer, unsupported channel count ...) */
  OGD_ERR_HIP = -2,      /* a HIP runtime call failed; ogd_last_error() has hipGetErrorString */
  OGD_ERR_NOMEM = -3,    /* host or device allocation failed */
  OGD_ERR_NODEVICE = -4, /* no gfx950 device visible */
  OGD_ERR_UNSUPPORTED = -5, /* input outside the domain of a fast set-up path; take the general one */
  OGD_ERR_RUNTIME = -6,    /* the algorithm failed where the reference raises RuntimeError (message alike) */
  OGD_ERR_ASSERT = -7      /* a result the reference asserts on (NaN in the node grid) */
} ogd_status;

#define OGD_INTER_NEAREST 0 /* cv2.INTER_NEAREST */
#define OGD_INTER_LINEAR 1  /* cv2.INTER_LINEAR  */

/* ---- library / device plumbing (no reference counterpart: the reference is CPU-only) ---- */
const char* ogd_last_error(void);
int ogd_version(void);
/* Number of visible GPUs (does not create a context). */
int ogd_device_count(int* count);
/* Bind the calling thread to `device` (-1: OGD_DEVICE env, else 0) and create its context. */
int ogd_init(int device);
/* PCI address of a GPU ("0000:c1:00.0"; does not create a context): a worker finds the NUMA node its GPU hangs
 * off under /sys/bus/pci/devices/<address>/numa_node and pins itself — and with it the pinned page buffers it is
 * about to allocate — to that node's cores (origami_amd/numa.py).  out: at least 16 bytes. */
int ogd_device_pci_bus_id(int device, char* out, int cap);
/* The host-buffer entry points keep the device blocks they used in a per-thread cache; this frees the
 * calling thread's idle ones (bytes freed are returned through freed_bytes if it is not NULL). */
int ogd_trim(size_t* freed_bytes);
/* Blocks of the calling thread's cache that are in use right now; 0 between calls (a block still counted after an
 * entry point returned will never be handed out again: what the leak tests watch). */
int ogd_pool_busy(int* blocks);
int ogd_dev_alloc(void** dptr, size_t bytes);
int ogd_dev_free(void* dptr);
int ogd_memcpy_h2d(void* dptr, const void* hptr, size_t bytes, void* stream);
int ogd_memcpy_d2h(void* hptr, const void* dptr, size_t bytes, void* stream);
int ogd_stream_sync(void* stream);
/* Strided forms (rows of `width` bytes, pitches in bytes), asynchronous on `stream` like the two above. */
int ogd_memcpy2d_h2d(void* dptr, size_t dpitch, const void* hptr, size_t hpitch, size_t width, size_t rows,
                     void* stream);
int ogd_memcpy2d_d2h(void* hptr, size_t hpitch, const void* dptr, size_t dpitch, size_t width, size_t rows,
                     void* stream);
/* Free / total device memory of the bound GPU (hipMemGetInfo). */
int ogd_mem_info(size_t* free_bytes, size_t* total_bytes);

/* ---- what a caller needs to overlap page copies with kernels (no reference counterpart) ------------
 * The reference opens and remaps one page at a time on the CPU (ref: origami/batch/core/io.py:275-280 ->
 * origami/core/page.py:82-88) and gets its parallelism from a process pool (ref: origami/batch/core/
 * processor.py:476-493).  On a GPU a broadsheet page is 48 MB up and 49 MB down for 0.07 ms of kernel, so the
 * page rate is the PCIe rate — if copies come from PINNED host memory (56 GB/s against 14 GB/s from pageable
 * memory) and run on streams of their own beside the kernels.  origami_amd/pipeline.py builds its page pipeline
 * (upload / kernel / download streams, N device slots) from these and the *_dev entry points.
 * ogd_host_alloc: page-locked host memory (hipHostMalloc); every host-buffer entry point also runs at the pinned
 * rate when handed such buffers.  Streams are created non-blocking (they do not synchronise with the null stream
 * the host-buffer entry points use).  A process gets GPU_MAX_HW_QUEUES (4) hardware queues: more streams than
 * that share queues and their copies take turns.  Events: timing = 0 creates them with timing disabled. */
int ogd_host_alloc(void** hptr, size_t bytes);
int ogd_host_free(void* hptr);
int ogd_stream_create(void** stream);
int ogd_stream_destroy(void* stream);
int ogd_stream_wait_event(void* stream, void* event);
int ogd_event_create(void** event, int timing);
int ogd_event_destroy(void* event);
int ogd_event_record(void* event, void* stream);
int ogd_event_sync(void* event);
int ogd_event_query(void* event, int* done);
int ogd_event_elapsed_ms(void* start, void* stop, float* ms);

/* ---- K1a: dense sampling grid -------------------------------------------------------------
 * ref: origami/core/dewarp.py:522-523  (cv2.resize(grid[:, :, c], (w*s, h*s), INTER_LINEAR) x2)
 * nodes: [h][w][2] float32 (x, y); out: [h*s][w*s][2] float32.  OpenCV pixel-centre alignment,
 * fp32 horizontal-then-vertical lerp.  sx, sy: the dewarp_image scale (ref: dewarp.py:622-623)
 * applied in fp32 to a copy (1.0f = none).
 */
int ogd_grid_fill(const float* nodes, int h, int w, int s, float sx, float sy, float* out);
int ogd_grid_fill_dev(const float* d_nodes, int h, int w, int s, float sx, float sy,
                      float* d_out, void* stream);
/* Rows y0 .. y0+nrows-1 of the same dense grid, written to d_out rows 0 .. nrows-1 (a band buffer).  A
 * caller that alternates band fills with ogd_remap_u8_dev on the band (map = d_out, Hd = nrows, dst
 * advanced by y0 rows) keeps the map in the 256 MB Infinity Cache instead of streaming it through HBM. */
int ogd_grid_fill_rows_dev(const float* d_nodes, int h, int w, int s, float sx, float sy,
                           int y0, int nrows, float* d_out, void* stream);
/* Same values sampled at dense pixels (ox*step, oy*step) only; out: [ceil(h*s/step)][ceil(w*s/step)][2].
 * ref: origami/core/dewarp.py:557-559 (points("full")[::r, ::r]) without materialising the map. */
int ogd_grid_points_strided(const float* nodes, int h, int w, int s, int step, float* out);

/* ---- K2: remap ------------------------------------------------------------------------------
 * ref: origami/core/dewarp.py:612-614 and :625-627  (cv2.remap(u8, map f32 (Hd,Wd,2), None, interp),
 * BORDER_CONSTANT 0).  src: [H][W][C] uint8, C in {1,3,4}, row stride in bytes; map_xy: [Hd][Wd][2]
 * float32 dense; dst: [Hd][Wd][C] uint8, row stride in bytes.  OpenCV fixed-point bilinear
 * (1/32 px, 15-bit weights) or round-half-even nearest.
 */
int ogd_remap_u8(const uint8_t* src, int H, int W, int C, ptrdiff_t src_stride,
                 const float* map_xy, int Hd, int Wd,
                 uint8_t* dst, ptrdiff_t dst_stride, int interp);
int ogd_remap_u8_dev(const uint8_t* d_src, int H, int W, int C, ptrdiff_t src_stride,
                     const float* d_map_xy, int Hd, int Wd,
                     uint8_t* d_dst, ptrdiff_t dst_stride, int interp, void* stream);
/* Fused K1a+K2: ref dewarp.py:608-627 as one pass; the dense map is never written to memory. */
int ogd_dewarp_u8(const uint8_t* src, int H, int W, int C, ptrdiff_t src_stride,
                  const float* nodes, int h, int w, int s, float sx, float sy,
                  uint8_t* dst, ptrdiff_t dst_stride, int interp);
int ogd_dewarp_u8_dev(const uint8_t* d_src, int H, int W, int C, ptrdiff_t src_stride,
                      const float* d_nodes, int h, int w, int s, float sx, float sy,
                      uint8_t* d_dst, ptrdiff_t dst_stride, int interp, void* stream);

/* ---- N4: the JPEG file of a dewarped page ---------------------------------------------------------
 * ref: origami/batch/core/io.py:606-610 (Writer.annotation: `image.save(path)` of the "L" image
 * page.dewarped to annotation.<name>.jpg), called from origami/batch/utils/dewarp.py:27-28.  PIL hands the
 * pixels to libjpeg(-turbo): baseline, one component, quality 75 unless given, standard Huffman tables,
 * JFIF 1.01 header.  These entry points write that file — the same bytes — with the DCT, quantisation,
 * Huffman coding and byte stuffing done on the device; only the code stream crosses PCIe.
 * img: [H][W] uint8, row stride in bytes; out: host buffer of `cap` bytes; *out_size receives the size of
 * the file (also when it does not fit: then OGD_ERR_INVALID is returned and nothing is written).
 * ogd_dewarp_jpeg_gray = ogd_dewarp_u8 (C = 1, scale 1, linear) followed by the encoder, the dewarped page
 * never leaves the device (what the batch stage `utils.dewarp` does per page).
 */
int ogd_jpeg_encode_gray(const uint8_t* img, int H, int W, ptrdiff_t stride, int quality,
                         uint8_t* out, size_t cap, size_t* out_size);
int ogd_jpeg_encode_gray_dev(const uint8_t* d_img, int H, int W, ptrdiff_t stride, int quality,
                             uint8_t* out, size_t cap, size_t* out_size, void* stream);
int ogd_dewarp_jpeg_gray(const uint8_t* src, int H, int W, ptrdiff_t src_stride,
                         const float* nodes, int h, int w, int s, int quality,
                         uint8_t* out, size_t cap, size_t* out_size);

/* ---- N4: decoding the page image ------------------------------------------------------------------
 * ref: origami/core/page.py:82-84 (`PIL.Image.open(path).convert("L")`).  A baseline JPEG decoder on the device
 * for 8-bit Huffman files, with or without restart intervals (DRI / RSTn), one component (grayscale) or three
 * (JFIF YCbCr in one interleaved scan, 4:4:4 / 4:2:2 / 4:2:0): the pixels `convert("L")` gives after PIL / libjpeg-turbo decoded the
 * file, bit for bit (Huffman decoding in parallel by fixed-point iteration over 512-bit chunks, islow IDCT,
 * libjpeg's fancy chroma upsampling and YCbCr -> RGB tables, PIL's RGB -> L).
 * ogd_jpeg_info: size and component count of any JPEG file; *supported = 1 if this decoder covers it (else
 * ogd_last_error() names the reason and the caller keeps its own decoder — progressive, arithmetic coding, 12 bit,
 * CMYK / Adobe, several scans ...).
 * ogd_jpeg_decode_gray: file bytes -> [H][W] uint8 (host, row stride in bytes); OGD_ERR_UNSUPPORTED for a file
 * outside the decoder's domain; OGD_ERR_RUNTIME for a scan this decoder gives up on (too few blocks, restart
 * markers that do not stand where the blocks say, a stream periodic throughout whose chunk states settle too
 * slowly): the caller keeps its own decoder then, too.  _dev: output in device memory.
 */
int ogd_jpeg_info(const uint8_t* file, size_t n, int* H, int* W, int* ncomp, int* supported);
/* The same from the marker segments alone, without walking the scan (which costs ~1.8 ms for a broadsheet page):
 * size and component count, and plausible = 0 if the headers already rule the file out.  For callers that size a
 * buffer and then let the decode call, which parses the whole file once, have the last word (OGD_ERR_UNSUPPORTED). */
int ogd_jpeg_header(const uint8_t* file, size_t n, int* H, int* W, int* ncomp, int* plausible);
/* scan file in -> dewarped page's file out (decode, ogd_dewarp_u8 with C = 1, scale 1, linear, encode: the
 * whole `utils.dewarp` step; only the two files cross PCIe).  OGD_ERR_UNSUPPORTED as ogd_jpeg_decode_gray. */
int ogd_jpeg_dewarp_jpeg_gray(const uint8_t* file, size_t n, const float* nodes, int h, int w, int s,
                              int quality, uint8_t* out, size_t cap, size_t* out_size);
int ogd_jpeg_decode_gray(const uint8_t* file, size_t n, uint8_t* out, int H, int W, ptrdiff_t stride);
int ogd_jpeg_decode_gray_dev(const uint8_t* file, size_t n, uint8_t* d_out, int H, int W, ptrdiff_t stride,
                             void* stream);

/* ---- K3 / K1b: piecewise-linear interpolation over a host Delaunay triangulation -------------
 * ref: origami/core/lingrid.py:286-290 (scipy LinearNDInterpolator x2), :174-189 (clip + NaN merge),
 *      :252-282 (dense variant, scipy griddata x2 + merge).
 * pts: [n][2] float64; simplices: [nt][3] int32 (Qhull, host-side); transform: [nt][3][2] float64 =
 * scipy Delaunay.transform (rows 0-1: inverse of the edge matrix, row 2: reference vertex), may be
 * NULL (then computed here in fp64); vals: [n][nv] float64, nv in {1,2}.
 */
typedef struct ogd_tri ogd_tri;
int ogd_tri_create(const double* pts, int n, const int32_t* simplices, int nt,
                   const double* transform, const double* vals, int nv, ogd_tri** out);
/* The same with scipy Delaunay.transform itself computed on the device (the LAPACK-style LU inverse with its
 * condition check, NaN rows for degenerate simplices: bit for bit what ogd_lattice_delaunay / ogd_delaunay_points
 * write with a non-NULL `transform`), for callers that got their simplices from those two with transform = NULL:
 * neither the host loop nor the 48 bytes per simplex of upload.  ref: lingrid.py:286-290. */
int ogd_tri_create_scipy(const double* pts, int n, const int32_t* simplices, int nt,
                         const double* vals, int nv, ogd_tri** out);
int ogd_tri_destroy(ogd_tri* t);
/* q: [nq][2] float64 -> out: [nq][nv] float64.  Points are clipped to bounds = {minx,miny,maxx,maxy}
 * (NULL = no clip), looked up in `inner` (may be NULL) and, where that yields NaN, in `extra`. */
int ogd_tri_eval(const ogd_tri* inner, const ogd_tri* extra, const double* bounds,
                 const double* q, int64_t nq, double* out);
int ogd_tri_eval_dev(const ogd_tri* inner, const ogd_tri* extra, const double* bounds,
                     const double* d_q, int64_t nq, double* d_out, void* stream);
/* Ragged batch: ring r owns points offsets[r] .. offsets[r+1]-1; one wavefront per ring.
 * ref: origami/batch/detect/dewarp.py:26 (shapely.ops.transform -> one Transformer call per ring).
 * The host form cuts rings longer than 64 vertices into 64-vertex pieces (one wavefront each), so that the
 * launch does not wait for its longest ring; a caller of the _dev form gets the same by passing the piece
 * boundaries as offsets (any partition of the points into intervals is valid). */
int ogd_tri_eval_ragged(const ogd_tri* inner, const ogd_tri* extra, const double* bounds,
                        const double* q, const int64_t* offsets, int nrings, double* out);
int ogd_tri_eval_ragged_dev(const ogd_tri* inner, const ogd_tri* extra, const double* bounds,
                            const double* d_q, const int64_t* d_offsets, int nrings,
                            double* d_out, void* stream);
/* The quadratic part of the validity test the stage applies to every warped polygon ring
 * (ref: origami/batch/detect/dewarp.py:35, shapely `geom.is_valid`): flags[r] = 1 if two non-adjacent
 * segments of closed ring r (points offsets[r] .. offsets[r+1]-1, first = last) touch or cross, else 0.
 * One wavefront per ring, exact orientation signs in fp64. */
int ogd_rings_self_cross(const double* q, const int64_t* offsets, int nrings, uint8_t* flags);
/* Dense variant on the lattice (x0 + ix*step, y0 + iy*step), ix < nx, iy < ny; component k of
 * lattice point (ix, iy) is written to out[ix*stride_x + iy*stride_y + k] (strides in elements).
 * out_f64 = 0: float32 output, 1: float64.  ref: origami/core/lingrid.py:257-277. */
int ogd_tri_rasterize(const ogd_tri* inner, const ogd_tri* extra, const double* bounds,
                      double x0, double y0, double step, int nx, int ny,
                      void* out, int64_t stride_x, int64_t stride_y, int out_f64);
int ogd_tri_rasterize_dev(const ogd_tri* inner, const ogd_tri* extra, const double* bounds,
                          double x0, double y0, double step, int nx, int ny,
                          void* d_out, int64_t stride_x, int64_t stride_y, int out_f64,
                          void* stream);

/* ---- inverse point map ------------------------------------------------------
 * ref: origami/core/dewarp.py:561-576 (scipy RegularGridInterpolator over the dense grid, linear,
 * extrapolating).  q_yx: [nq][2] float64 (y, x) in dewarped pixels -> out_xy: [nq][2] float64 warped
 * page position.  The dense grid is evaluated on the fly from the node grid (K1a arithmetic). */
int ogd_grid_sample_bilinear(const float* nodes, int h, int w, int s,
                             const double* q_yx, int64_t nq, double* out_xy);

/* ---- device-resident pages and per-line images ---------------------------------
 * ref: origami/core/page.py:82-88 (every later stage opens the page and dewarps it), origami/core/block.py:
 * 223-259 (Line.warped_grid + Line.dewarped_image: one small cv2.remap per text line, thousands per page),
 * origami/batch/core/lines.py:121-149 (LineExtractor).  A page is uploaded once; all lines of the page
 * are extracted by ONE launch. */
typedef struct ogd_page ogd_page;
int ogd_page_create(const uint8_t* src, int H, int W, int C, ptrdiff_t src_stride, ogd_page** out);
/* The page of a grayscale baseline JPEG scan, decoded straight into device memory (ogd_jpeg_decode_gray_dev);
 * OGD_ERR_UNSUPPORTED for other files: decode on the host and use ogd_page_create. */
int ogd_page_create_jpeg(const uint8_t* file, size_t n, ogd_page** out);
int ogd_page_destroy(ogd_page* page);
/* Dewarper.dewarped / dewarp_image of a resident page (fused K1a + K2), ref: dewarp.py:608-627. */
int ogd_page_dewarp_u8(const ogd_page* page, const float* nodes, int h, int w, int s, float sx, float sy,
                       uint8_t* dst, ptrdiff_t dst_stride, int interp);
/* The JPEG file of the dewarped page (ogd_dewarp_jpeg_gray) from a resident gray page: what `utils.dewarp` writes
 * (ref: origami/batch/utils/dewarp.py:27-28) when the page is already on the device because an earlier stage of
 * the same process put it there (ref: origami/batch/core/io.py:275-280, where every stage opens and remaps the page
 * again).  out / cap / out_size as ogd_jpeg_encode_gray. */
int ogd_page_dewarp_jpeg_gray(const ogd_page* page, const float* nodes, int h, int w, int s, int quality,
                              uint8_t* out, size_t cap, size_t* out_size);
/* lines: [n][6] float64 = p0.x, p0.y, right.x, right.y, up.x, up.y in dewarped pixels; line l yields a
 * ysteps[l] x xsteps[l] x C image (rows already flipped as block.py:257 does) at out + offsets[l];
 * offsets: [n + 1] byte offsets.  Sample (i, j) sits at p0 + linspace(0, up, ysteps)[i] +
 * linspace(0, right, xsteps)[j], goes through the inverse map (fp64 bilinear over the dense grid, cast to
 * float32) and is read from the warped page like cv2.remap does. */
int ogd_line_images_u8(const ogd_page* page, const float* nodes, int h, int w, int s,
                       const double* lines, const int32_t* xsteps, const int32_t* ysteps, int n,
                       int interp, uint8_t* out, const int64_t* offsets);

/* The same with the images left in DEVICE memory: d_out holds offsets[n] bytes and is complete when the call
 * returns (the launch goes to `stream`, which is synchronised before return; the node grid and the line records are
 * staged through the library's block cache).  For a GPU-resident consumer (an OCR front end on the same device):
 * of the 17 ms ogd_line_images_u8 takes for the 3000 lines of a broadsheet page, 15 ms are the 152 MB
 * device -> pageable-host copy that this form does not make. */
int ogd_line_images_u8_dev(const ogd_page* page, const float* nodes, int h, int w, int s,
                           const double* lines, const int32_t* xsteps, const int32_t* ysteps, int n,
                           int interp, uint8_t* d_out, const int64_t* offsets, void* stream);

/* ConfidenceSampler, ref: origami/batch/detect/lines.py:18-54: per line, the histogram of
 * cv2.remap(labels, warped_grid * (sx, sy), INTER_NEAREST) — np.bincount of the remapped labels; the remapped
 * image is never materialised.  labels: a one-channel ogd_page holding the label map; sx = w1 / w0,
 * sy = h1 / h0 (label map size / page size, applied in float32 as scale_grid() does); rows are NOT flipped
 * (irrelevant to a histogram); counts: [n][256] int32, overwritten. */
int ogd_line_label_counts(const ogd_page* labels, const float* nodes, int h, int w, int s, float sx, float sy,
                          const double* lines, const int32_t* xsteps, const int32_t* ysteps, int n,
                          int32_t* counts);

/* ---- Transformer set-up without Qhull --------------------------------------------
 * ref: origami/core/dewarp.py:143-161 (LinearNDInterpolator over the h x w node positions),
 * origami/core/lingrid.py:208-221, 286-290 (scipy.spatial.ConvexHull + Delaunay of the same points).
 * Host code, no device work: the Delaunay triangulation of a warped lattice from its known topology
 * (quads split, hull pockets ear-clipped, Lawson flips with exact predicates) instead of a general hull code.
 * pts: [h*w][2] float64 row-major lattice positions.  simplices: [cap][3] vertex indices (cap >= 2*h*w is
 * always enough); hull: [2*(h+w)] strict convex hull vertices, counter-clockwise like ConvexHull.vertices;
 * transform: NULL or [cap][3][2] in the layout of scipy.spatial.Delaunay.transform.
 * Returns OGD_ERR_UNSUPPORTED for a folded / degenerate lattice or coordinates outside the exact domain
 * (|x| < 2^20, multiples of 2^-40 — any float32 grid): the caller then runs Qhull as before. */
int ogd_lattice_delaunay(const double* pts, int h, int w, int32_t* simplices, int64_t cap,
                         int64_t* n_simplices, int32_t* hull, int* n_hull, double* transform);

/* ---- Delaunay triangulation of scattered points without Qhull (host code) --------------
 * ref: origami/core/lingrid.py:286-290 — scipy.interpolate.LinearNDInterpolator builds scipy.spatial.Delaunay
 * (Qhull) over the samples and over the extrapolation set of lingrid.py:192-250; origami/core/dewarp.py:108-116
 * and :143-161 reach it five times per page after ogd_lattice_delaunay.
 * Sweep over the sorted points + Lawson flips, predicates exact for any finite doubles.  pts: [n][2] float64;
 * simplices: [cap][3] (cap >= 2*n is always enough), positively oriented; transform: NULL or [cap][3][2] in the
 * layout of scipy.spatial.Delaunay.transform; n_ties: NULL or the number of interior edges whose two triangles
 * have cocircular vertices (the triangulation is unique, and equal to Qhull's, iff this is 0 — with ties the
 * caller should run Qhull, whose choice of diagonal cannot be predicted).
 * twin: NULL or [n]: twin[i] = -1 for a triangulated point, else the index of the coincident point (the lowest
 * index of its group) that stands for point i, which is then in no simplex (Qhull lists such points as
 * "coplanar" and keeps one of the group, not necessarily the same one); with NULL coincident points are refused.
 * Returns OGD_ERR_UNSUPPORTED for fewer than 3 distinct points, all-collinear input (Qhull raises QhullError
 * there, which the reference catches) and, with twin = NULL, coincident points. */
int ogd_delaunay_points(const double* pts, int64_t n, int32_t* simplices, int64_t cap,
                        int64_t* n_simplices, double* transform, int64_t* n_ties, int32_t* twin);

/* ---- WKT coordinate text (host code) ---------------------------------------------------------------
 * ref: origami/batch/detect/dewarp.py:39 (`geom.wkt`: shapely -> GEOS WKTWriter, trim, 16 significant digits).
 * Ring r (points offsets[r] .. offsets[r+1]-1 of pts [n][2] float64) becomes "(x y, x y, ...)" at
 * out[text_offsets[r] .. text_offsets[r+1]); numbers as printf("%.16g"), "nan", "inf", "-inf".
 * cap >= 52 * n + 2 * nrings bytes is always enough.  No terminating NUL is written. */
int ogd_wkt_ring_text(const double* pts, const int64_t* offsets, int nrings, char* out, int64_t cap,
                      int64_t* text_offsets);
/* The reading side, ref: origami/batch/core/io.py:142-144 (shapely.wkt.loads per .wkt member -> GEOS WKTReader).
 * n texts, text g at text[text_offsets[g] .. text_offsets[g+1]); the byte at text[text_offsets[n]] must be NUL.
 * POLYGON / LINESTRING / MULTIPOLYGON, optional Z, EMPTY, 2 to 4 numbers per coordinate (x, y kept).
 * kinds[g]: 0 polygon, 1 linestring, 2 multipolygon.  The rings of all geometries in order: points
 * ring_offsets[r] .. ring_offsets[r+1]-1 of pts [cap_pts][2]; ring_geom[r] = g; ring_part[r] = index of the
 * polygon inside a multipolygon (else 0).  A geometry without rings is EMPTY.  Capacities: one point per ','
 * plus one per text, one ring per '(' are always enough.  Anything else: OGD_ERR_UNSUPPORTED (the caller's
 * own parser then reports the error as the reference would). */
int ogd_wkt_parse(const char* text, const int64_t* text_offsets, int n, int8_t* kinds, double* pts,
                  int64_t cap_pts, int64_t* ring_offsets, int32_t* ring_geom, int32_t* ring_part,
                  int64_t cap_rings, int64_t* n_rings);
/* Signed area of every ring in the summation order of GEOS Area::ofRingSigned (what shapely's `geom.area`
 * adds up, ref: origami/batch/detect/dewarp.py:29-31): out[r] = sum_i (x_i - x_0) * (y_{i-1} - y_{i+1}) / 2. */
int ogd_rings_area_signed(const double* pts, const int64_t* offsets, int nrings, double* out);

/* ---- PNG page scans (host code) ------------------------------------------------------
 * ref: origami/core/page.py:82-88 (PIL.Image.open: the page's pixels, decoded anew by every stage).
 * A PNG's pixel data is one serial zlib stream; libogdewarp brings its own inflate (csrc/ogd_inflate.h: 64-bit
 * bit buffer, 10-bit tables with literal pairs, several symbols per refill) because that loop, not the GPU, bounds a PNG page.
 * ogd_png_info: size and channel count (1 grey, 3 RGB) of an 8-bit non-interlaced grey or RGB file;
 * OGD_ERR_UNSUPPORTED for every other kind (palette, 16 bit, alpha, Adam7: the caller keeps PIL for those).
 * ogd_png_decode_u8: the pixels, row y at out + y * out_stride, channels interleaved as PIL's "L" / "RGB" arrays. */
int ogd_png_info(const uint8_t* file, int64_t n, int* width, int* height, int* channels);
int ogd_png_decode_u8(const uint8_t* file, int64_t n, uint8_t* out, int64_t out_stride);
/* The zlib stream decoder on its own (what zlib.decompress does): n_out bytes written to out[cap];
 * OGD_ERR_INVALID for corrupt, truncated or too long data (Adler-32 verified). */
int ogd_inflate_zlib(const uint8_t* in, int64_t n_in, uint8_t* out, int64_t cap, int64_t* n_out);

/* ---- the stage's output archives (host code) -------------------------------------------
 * ref: origami/batch/core/io.py:516-518, 565-577 (contours.N.zip), origami/core/dewarp.py:545-553 (dewarp.zip): members
 * written with zipfile.ZIP_DEFLATED.  ogd_deflate_raw writes a raw deflate stream (what a zip member of method 8
 * holds) of dynamic-Huffman literal blocks, or stored blocks where those would be larger: no string matching, which
 * finds next to nothing in float32 node positions and coordinate text, at ~10x zlib's level-1 speed and within a few
 * per cent of its size.  cap >= n + n / 1000 + 64 is always enough.
 * ogd_deflate_many: members k = in[offsets[k] .. offsets[k+1]) -> out[out_offsets[k] .. out_offsets[k+1]) and their
 * CRC-32s, one call per archive. */
int ogd_deflate_raw(const uint8_t* in, int64_t n, uint8_t* out, int64_t cap, int64_t* n_out);
int ogd_deflate_many(const uint8_t* in, const int64_t* offsets, int n, uint8_t* out, int64_t cap,
                     int64_t* out_offsets, uint32_t* crcs);
/* PNG files of many small images in one call — the line images of a page (ref: origami/batch/export/lines.py:62-67:
 * `im.save(f, format='png', optimize=True)` per line).  8-bit grey (channels = 1) or RGB (3), rows filtered by the
 * minimum sum of absolute residuals, one IDAT chunk, the entropy coder of ogd_deflate_raw: the same pixels for any
 * PNG reader, not PIL's bytes.  Image k: [heights[k]][widths[k]][channels] uint8, contiguous, at images[k]; its file is
 * written at out + out_starts[k], where out_starts[k + 1] - out_starts[k] >= ogd_png_bound(heights[k], widths[k],
 * channels) bytes are free; out_sizes[k] receives its size, out_crcs[k] (may be NULL) the CRC-32 of the file (for
 * the zip member that stores it).  Host code, up to OGD_THREADS (default 4) threads. */
int64_t ogd_png_bound(int height, int width, int channels);
int ogd_png_encode_many(const uint8_t* const* images, const int32_t* widths, const int32_t* heights,
                        int channels, int n, uint8_t* out, const int64_t* out_starts, int64_t* out_sizes,
                        uint32_t* out_crcs);
/* The reading side (ref: origami/batch/core/io.py:142-144 and origami/core/dewarp.py:537-543 read the archives
 * through zipfile, one Python-level call per member): member k of an archive held in buf is the raw deflate stream
 * (methods[k] == 8) or the stored bytes (0) at buf[in_off[k] .. + in_len[k]); it is written to
 * out[out_off[k] .. out_off[k+1]) — the size the archive's directory states — and checked against crcs[k].
 * OGD_ERR_INVALID for a corrupt or truncated stream, a wrong size or a wrong CRC-32. */
int ogd_inflate_many(const uint8_t* buf, int64_t n_buf, const int64_t* in_off, const int64_t* in_len,
                     const int32_t* methods, int n, uint8_t* out, const int64_t* out_off, const uint32_t* crcs);

/* ---- the march of the node grid on the device -------------------------------------
 * ref: origami/core/dewarp.py:108-140 (Field), :328-339 (GridFactory.grid_shape), :421-439 (grid_h),
 * :441-493 (grid_hv without the border extension), :176-263 (ShapelyBatchIntersections).
 * A field is the angle interpolant lininterp(samples.points, samples.values, bounds) as an (inner, extra)
 * pair of scalar triangulations (inner may be NULL), or the constant angle phi0 when extra is NULL (a field
 * with <= 2 samples, ref: dewarp.py:112-113).  bounds = {0, 0, W, H} as passed to lininterp.
 *
 * ogd_march_shape: Field.estimate_extent of both fields -> n_cols (est_width), n_rows (est_height);
 * OGD_ERR_RUNTIME "n_steps exceeded" like the reference.  The caller applies max_grid_size.
 * ogd_march_grid: grid_h [n_rows][n_cols][2] float32 — as the reference leaves it, i.e. with the first and
 * last node of rows 1.. pushed outwards by make_row_slice's in-place extension (dewarp.py:205-211) — and
 * grid_hv [n_rows][n_cols][2] float32 before border extension.  num_slices = the reference's num_threads
 * (columns are marched per slice with that slice's ray length, dewarp.py:448-483).  OGD_ERR_RUNTIME
 * "failed to find intersection", OGD_ERR_ASSERT for NaN in grid_h. */
int ogd_march_shape(const ogd_tri* inner_h, const ogd_tri* extra_h, double phi0_h,
                    const ogd_tri* inner_v, const ogd_tri* extra_v, double phi0_v,
                    const double* bounds, int W, int H, int s, int* n_rows, int* n_cols);
int ogd_march_grid(const ogd_tri* inner_h, const ogd_tri* extra_h, double phi0_h,
                   const ogd_tri* inner_v, const ogd_tri* extra_v, double phi0_v,
                   const double* bounds, int s, int n_rows, int n_cols, int num_slices,
                   float* grid_h, float* grid_hv);
/* ogd_march_grid for callers that cannot allocate.  No counterpart in the reference: it exists for callers that
 * may not call hipMalloc — tight serving loops, harnesses that forbid allocation.  (It is a synchronous call like
 * ogd_march_grid — blocking copies of the results to the host on the null stream —, so it cannot itself be
 * captured into a HIP graph.)
 * Same arguments, same results (it runs the same kernels), but its device buffers — both node grids, the slice
 * ray lengths, the status record — are a fixed ~4 MiB array in the library's device image, held by every GPU the
 * library runs on until the process ends; no hipMalloc / hipFree or other allocating routine is called.
 * Limits: n_rows * n_cols <= 262144 nodes and at most 1024 column slices, else OGD_ERR_INVALID (use
 * ogd_march_grid).  All calls on one GPU share that array: calls that overlap on the same GPU race, so
 * serialise them per GPU (the array keeps the last call's contents; each call resets what it reads). */
int ogd_march_grid_noalloc(const ogd_tri* inner_h, const ogd_tri* extra_h, double phi0_h,
                           const ogd_tri* inner_v, const ogd_tri* extra_v, double phi0_v,
                           const double* bounds, int s, int n_rows, int n_cols, int num_slices,
                           float* grid_h, float* grid_hv);

#ifdef __cplusplus
}
#endif
#endif /* OGDEWARP_H */
