/* rpsf.h - C ABI of the MI355X patch-wise PSF-correction library (librpsf_hip.so).
 *
 * The reference (punch-mission/regularizepsf 1.2.0) is pure Python and has no FFI seam of its
 * own; the boundary it offers is the class API.  Each entry point below states which reference
 * lines it stands in for, so that a reference maintainer can bind it with ctypes (see
 * INTEGRATION.md for the stub).  Conventions:
 *   - plain pointers and sizes only; no C++ / torch types cross the boundary;
 *   - every function returns 0 on success or a negative RPSF_E_* code; rpsf_last_error() returns a
 *     thread-local human-readable message for the last failure on the calling thread;
 *   - host pointers are borrowed for the duration of the call only; a plan owns all of its device
 *     memory; nothing returned by the library is freed by the caller except through *_destroy/_free;
 *   - a plan is bound to one device and is not re-entrant; distinct plans may be used from
 *     distinct threads;
 *   - applies of distinct plans on distinct streams may run on the device at the same time.  The launches of the
 *     128- and 256-pixel plans are persistent (workgroups that stay resident and draw patches from queues) and contain
 *     workgroups that wait for others of the same launch; they need no particular number of resident workgroups to
 *     finish - every patch is drawn from a queue by whichever workgroups are resident - as long as the few (<= 128)
 *     summing workgroups at the head of all concurrent launches together leave one compute unit free
 *     (csrc/rpsf.hip, launch_patches, "FORWARD PROGRESS").
 */
#ifndef RPSF_H
#define RPSF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPSF_OK 0
#define RPSF_E_BADARG (-1)      /* null pointer, negative size, coordinate outside the padded image ... */
#define RPSF_E_UNSUPPORTED (-2) /* patch size outside 2..4096; or a size without a compiled plan (compiled: 16, 32, 64,
                                   128, 256) when libhipfft.so, which the fallback needs, cannot be loaded */
#define RPSF_E_HIP (-3)         /* HIP runtime error (message has the hipError string) */
#define RPSF_E_NOMEM (-4)       /* device allocation failed */
#define RPSF_E_RCCL (-5)        /* RCCL missing or failed */
#define RPSF_E_STATE (-6)       /* call order: e.g. apply before a transfer kernel was set */

/* np.pad modes evaluated inside the kernel (regularizepsf/transform.py:119-123 passes pad_mode to
 * np.pad).  Any other np.pad mode is handled by the Python layer, which pads on the host and calls
 * the library on the padded image with RPSF_PAD_CONSTANT. */
#define RPSF_PAD_CONSTANT 0
#define RPSF_PAD_SYMMETRIC 1
#define RPSF_PAD_REFLECT 2
#define RPSF_PAD_EDGE 3
#define RPSF_PAD_WRAP 4

typedef struct rpsf_plan rpsf_plan;

const char* rpsf_last_error(void);
int rpsf_device_count(int* count);
/* Compute-unit count and name of a device (for reporting). */
int rpsf_device_info(int device, int* compute_units, char* name, size_t name_len);

/* A plan is the device-side form of one ArrayPSFTransform: patch size N (psf_shape, square,
 * regularizepsf/transform.py:37-40), the patch corner list (transform.coordinates, (row, col) pairs,
 * transform.py:141-149) and, once installed, the transfer kernel.  N = 16, 32, 64, 128, 256 use the
 * hand-written kernels; any other N in 2..4096 (the reference takes every size) a hipFFT-based fallback with
 * the same results and tolerance (whole-image geometry only). */
int rpsf_plan_create(rpsf_plan** out, int device, int patch_size, int n_patches, const int32_t* coords_rc);
void rpsf_plan_destroy(rpsf_plan* plan);

/* Install the transfer kernel K (IndexedCube values of ArrayPSFTransform, complex64 interleaved,
 * shape (n_patches, N, N), transform.py:164).  The library folds it to its Hermitian part and
 * re-orders it into the layout the patch kernel streams.  host: pageable host memory. */
int rpsf_plan_set_transfer(rpsf_plan* plan, const float* k_c64_host);
/* Same, from a device-resident full K (e.g. produced by rpsf_build_transfer_device). */
int rpsf_plan_set_transfer_device(rpsf_plan* plan, const void* k_c64_device);
/* The same from the two PSF spectra K is built from (ArrayPSFTransform.construct, transform.py:78-82; complex64 (n_patches, N, N) each, on
 * the plan's device, e.g. left there by rpsf_psf_fft_device): the formula is evaluated where the packer reads K, so the full K is never
 * written or read back - 2 x n N^2 x 8 B in, the packed n N (N/2 + 1) x 8 B out, bit-identical to rpsf_build_transfer_device followed by
 * rpsf_plan_set_transfer_device. */
int rpsf_plan_set_transfer_spectra_device(rpsf_plan* plan, const void* s_c64_dev, const void* t_c64_dev, double alpha, double epsilon);
/* Overlap-add strategy (transform.py:167-169).  0 = automatic: on a regular half-overlap lattice of
 * corners (calculate_covering output) patches of equal lattice parity never overlap, so each patch
 * stores into one of four colour planes with plain coalesced stores and the planes are summed in a
 * fixed order (deterministic) - inside the same launch for 256-pixel patches, by a small kernel
 * otherwise; any other corner list falls back to float atomics.
 * 1 = force atomics, 2 = force planes (error if the corners are not a lattice), 3 = direct: every
 * lattice tile is accumulated in the output image itself, through the L2 of the XCD that runs most of
 * its patches, in processing order (deterministic; 128- and 256-pixel patches on a lattice only;
 * measured slower than the planes, kept for comparison).
 * 4 = sweep (what 0 selects for 16-, 32- and 64-pixel patches on a complete lattice of at least 2 x 2 patches):
 * a workgroup owns a region of OUTPUT pixels, walks every patch that touches it and adds the four
 * contributions of a pixel in LDS, in a fixed order (lattice rows top-down, even patch columns before odd
 * ones: bit-reproducible and independent of how the lattice is cut); every output pixel is written once,
 * by the one launch that is the whole apply.  Patches on region borders are computed by both neighbours. */
int rpsf_plan_set_overlap_mode(rpsf_plan* plan, int mode);
/* Sweep kernel (mode 4): cut the lattice into about `target_regions` regions of output pixels (one workgroup each; default: the device's
 * number of compute units).  More regions: shorter regions and a better balance over the CUs, more patches computed twice on region
 * borders.  The result does not depend on the cut, bit for bit.  rpsf_plan_sweep_info reports the cut in use: regions, jobs (slabs of
 * 128 / N patches), patch slots computed (/ the number of patches = the recompute factor), slabs per column parity and lattice row
 * of a region.  (The reference has no counterpart: transform.py:157-169 is one NumPy expression per step.) */
int rpsf_plan_set_sweep_regions(rpsf_plan* plan, int target_regions);
/* Launch options of a plan that tests and callers may pin (everything else is decided by the library; none changes a result beyond the
 * round-off of a different addition order, and the persistent / fused forms are bit-identical to their plain counterparts).
 * ENVIRONMENT: the shipped library reads exactly two variables, both for the host-side thread pool of the host-array entry points:
 *   RPSF_HOST_THREADS  (threads of the pool, default: the cores of the device's NUMA node, at most 16)
 *   RPSF_HOST_AFFINITY (0: do not bind the pool's threads to the device's NUMA node).
 * The development knobs of earlier rounds (RPSF_STRIPS, RPSF_SUM_FIRST, RPSF_V1, ...) are compiled in only with -DRPSF_DEV_ENV. */
enum {
  RPSF_OPT_PERSIST = 1,       /* 0 / 1: persistent patch workgroups for 128- and 256-pixel plans (default 1) */
  RPSF_OPT_FUSE = 2,          /* 0 / 1: colour-plane sum inside the patch launch where the geometry allows (default 1) */
  RPSF_OPT_K_CACHED = 3,      /* 0 / 1: transfer kernel by plain instead of streaming loads (default: by its size) */
  RPSF_OPT_PLANE_NT = 4,      /* -1 automatic, 0 / 1: streaming hint on the colour-plane stores of 128-pixel batches */
  RPSF_OPT_HOST_BANDS = 5,    /* -1 automatic, else the row bands a single host frame is cut into (0 or 1: none) */
  RPSF_OPT_STREAM_GROUP = 6,  /* 0 automatic, else frames per group of the streamed host path */
  RPSF_OPT_STREAM_DEPTH = 7,  /* 0 automatic, else groups in flight of the streamed host path */
  RPSF_OPT_DEBUG_ORPHAN = 8   /* testing aid of the direct overlap-add: every value-th workgroup behaves as if placed on a foreign XCD */
};
int rpsf_plan_set_option(rpsf_plan* plan, int option, int value);
int rpsf_plan_sweep_info(const rpsf_plan* plan, int* regions, long* jobs, long* patch_slots, int* slabs_per_phase);
/* Row bands the last single host frame (rpsf_apply_host / rpsf_apply, one frame) was cut into: 0 = it went as a whole.  A frame of B bands
 * cannot take less than PCIe's duplex time x (1 + 1/B) (bench.py's e2e.pcie_floor_ms). */
int rpsf_plan_host_bands(const rpsf_plan* plan, int* bands);
/* Start-up stagger of the patch kernel's first resident workgroups (microseconds, 0 = off): spreads
 * the gather / K-stream / store phases of different CUs in time so that HBM traffic overlaps compute.
 * Without this call the library decides: 12 us for the persistent launches of the 256-pixel plan from
 * 256 patches on (24 us from 2048 on), none otherwise. */
int rpsf_plan_set_stagger(rpsf_plan* plan, int microseconds);
/* The persistent launch of the 256-pixel plan keeps one workgroup on every CU until its patches are done, so a kernel
 * enqueued beside it on another stream - RCCL's send/recv of the seam rows (rpsf_comm_seam_exchange) - would wait for the
 * first of them to finish.  `cus` CUs (0..128; 8 is what the sharded apply asks for) are left without a patch workgroup. */
int rpsf_plan_set_reserved_cus(rpsf_plan* plan, int cus);
/* Partition the chip between the plan and a caller's own stream: the plan's stream (and every launch on it) is confined to the compute
 * units set in `mask` (bit i of word i / 32, hipExtStreamCreateWithCUMask's numbering; `words` uint32), its persistent launches are
 * sized for them; rpsf_stream_create makes a stream masked to whatever the caller passes (NULL: unmasked).  The pipelined seam exchange
 * gives RCCL's kernels and K4 a handful of CUs of their own this way (rpsf_plan_set_reserved_cus only leaves CUs unclaimed - a kernel
 * dispatched a moment before the persistent launch still lands on CUs that launch is waiting for). */
int rpsf_plan_set_cu_mask(rpsf_plan* plan, const uint32_t* mask, int words);
int rpsf_stream_create(int device, const uint32_t* mask, int words, void** stream);
int rpsf_stream_destroy(void* stream);
/* Opt-in image prefetch for streams of NEW frames (256-pixel plan, single-frame applies; ignored elsewhere): the summing workgroups at
 * the head of the persistent launch touch the image tile by tile, in the order in which the patches will gather it, as a paced side
 * job.  A frame that was not corrected a moment ago is not in the memory-side cache, and its first gathers pay the HBM latency
 * (transform.py:157-162 is the gather): -4 % per apply on a stream of different 4096^2 frames, nothing on a repeated one; frames
 * larger than the cache (8192^2) lose 8 %, which is why the library does not decide this by itself. */
int rpsf_plan_set_image_prefetch(rpsf_plan* plan, int on);
/* Development aid: in builds compiled with -DRPSF_STAMPS the patch kernel records 16 phase
 * timestamps per patch (10 ns ticks); this copies them out.  All zeros in a normal build. */
int rpsf_plan_debug_stamps(rpsf_plan* plan, unsigned long long* host, size_t count);
/* Bytes of packed transfer kernel the patch kernel reads per apply (for roofline accounting). */
int rpsf_plan_transfer_bytes(const rpsf_plan* plan, size_t* bytes);

/* Image geometry of one apply call.  The reference pads the image by 2N on every side with
 * np.pad(mode) and slices patch (r, c) at padded[r+2N : r+3N, c+2N : c+3N] (transform.py:119-123,
 * 141-149); here padding is an index map evaluated inside the kernel against the FULL image shape
 * (height, width).  origin_* is added to every patch corner (used when the caller hands over an
 * already padded image).  image_row0/image_rows and out_row0/out_rows describe which rows of the
 * full image / full output are resident at the device pointers (row-band sharding); for a whole
 * image they are 0 and height. */
typedef struct rpsf_geometry {
  int height, width;
  int pad_mode;
  float pad_value;
  int origin_row, origin_col;
  int image_row0, image_rows, ld_image;
  int out_row0, out_rows, ld_out;
} rpsf_geometry;

/* ArrayPSFTransform.apply, transform.py:117-177 without the saturation branch (:125-138,:171-172,
 * done by the Python layer): float32 image (height, width) in, float32 corrected image out. */
int rpsf_apply(rpsf_plan* plan, const float* image_host, int height, int width, int pad_mode, float pad_value,
               float* out_host);
/* The same call for the dtypes ArrayPSFTransform.apply really sees: the image as float32 or float64
 * (image_is_f64; apply casts with astype, transform.py:117) and the result as float32 or float64
 * (out_is_f64; the reference returns float64, transform.py:174-177).  The conversions run on the library's persistent
 * worker pool (created at the first such call of the process, RPSF_HOST_THREADS wide, default min(16, cores); no
 * thread or buffer is created per call), chunk by chunk through the plan's pinned staging, overlapped with the PCIe
 * copies.  rpsf_apply above is this call with float32 on both sides. */
int rpsf_apply_host(rpsf_plan* plan, const void* image_host, int image_is_f64, int height, int width, int pad_mode,
                    float pad_value, void* out_host, int out_is_f64);
/* ArrayPSFTransform.apply with a finite saturation_threshold, whole (transform.py:117-138,171-177): float64 copy, np.pad by
 * 2N (pad_mode's index map on the host), mask = padded > threshold, binary dilation with scipy's default cross element
 * (dilation >= 1 iterations), NaN, the sequential row-major nanmean fill over [i - w/2, i + w/2) (neighborhood_width >= 0),
 * the correction of the padded frame on the GPU, the raw values restored on the mask, the crop.  The host steps are the
 * reference's, in its order, on the plan's own scratch; only the rows the patches read and the rows the caller gets cross
 * PCIe.  (dilation < 1, negative widths and np.pad modes the kernel does not know stay with the Python layer's NumPy route.) */
int rpsf_apply_host_saturated(rpsf_plan* plan, const void* image_host, int image_is_f64, int height, int width, int pad_mode,
                              double threshold, int dilation, int neighborhood_width, void* out_host, int out_is_f64);
/* The same for a sequence of frames of one shape, one pointer per frame - the reference's example corrects a list of frames this way
 * (`[transform.apply(image, saturation_threshold=...) for image in images]`, docs/source/example.ipynb cell 25): the host steps of
 * frame i + 1 run while the GPU corrects frame i (two staging slots in turn). */
int rpsf_apply_frames_host_saturated(rpsf_plan* plan, const void* const* images_host, int image_is_f64, int n_frames, int height,
                                     int width, int pad_mode, double threshold, int dilation, int neighborhood_width,
                                     void* const* outs_host, int out_is_f64);
/* Same with image and output already resident on the plan's device; asynchronous on `stream`
 * (a hipStream_t, or NULL for the plan's own stream).  Every pixel of the resident output rows is written
 * (uncovered ones with 0).  image_dev / out_dev must be ordinary device memory of the plan's device
 * (hipMalloc: coarse-grained) - the atomic overlap-add mode uses hardware float atomics, which
 * fine-grained or host-mapped memory does not support.  A plan serves one apply at a time: a call on a
 * different stream than the previous one waits for that one (the plan's scratch is shared). */
int rpsf_apply_device(rpsf_plan* plan, const void* image_dev, void* out_dev, const rpsf_geometry* geom, void* stream);
/* Run `iters` back-to-back device-resident applies on the plan's stream, each bracketed by HIP events of its own, one
 * synchronisation at the end (the launches queue up as in a caller's loop): total_ms[i] covers the whole apply
 * (output clear + patch kernel), kernel_ms[i] the patch kernel alone.  Either array may be NULL. */
int rpsf_apply_device_timed(rpsf_plan* plan, const void* image_dev, void* out_dev, const rpsf_geometry* geom,
                            int iters, float* total_ms, float* kernel_ms);
/* `iters` applies back to back between ONE pair of HIP events on the plan's stream: the average device time of an apply in
 * a caller's loop (what bench.py's roofline line divides the algorithmic bytes by). */
int rpsf_apply_device_loop_ms(rpsf_plan* plan, const void* image_dev, void* out_dev, const rpsf_geometry* geom, int iters,
                              double* ms_per_apply);
/* A batch of n_frames frames of identical geometry corrected with the plan's (shared) transfer kernel -
 * what a caller of the reference does with a Python loop over ArrayPSFTransform.apply
 * (transform.py:85-177) on a stack of exposures.  The frames of one patch are scheduled next to each other so
 * the packed transfer kernel is read from HBM once per batch, not once per frame.
 * Host variants - the STREAMED path: what a user of the reference runs is `[transform.apply(image) for image in images]`
 * (docs/source/example.ipynb over transform.py:85-177), host arrays in and out, and every frame crosses PCIe twice, which
 * costs several times the kernel.  The frames go through the plan's staging slots in groups, up to four groups in flight on
 * three streams: H2D of group i + 1 || the shared-K launch of group i (rpsf_apply_batch_device's) || D2H of group i - 1, with
 * the dtype conversions of both directions on the persistent worker pool - PCIe in both directions is kept busy and is what
 * bounds the call (bench.py --config 5 --streamed reports it against rpsf_pcie_probe's floor).  Results are bit-identical to
 * the frame-by-frame loop.
 *   rpsf_apply_batch:        (n_frames, height, width) float32 stacks, C-contiguous;
 *   rpsf_apply_batch_host:   the same with float32 or float64 on either side (as rpsf_apply_host);
 *   rpsf_apply_frames_host:  one pointer per frame (a Python list of arrays needs no np.stack). */
int rpsf_apply_batch(rpsf_plan* plan, const float* images_host, int n_frames, int height, int width, int pad_mode,
                     float pad_value, float* outs_host);
int rpsf_apply_batch_host(rpsf_plan* plan, const void* images_host, int image_is_f64, int n_frames, int height, int width,
                          int pad_mode, float pad_value, void* outs_host, int out_is_f64);
int rpsf_apply_frames_host(rpsf_plan* plan, const void* const* images_host, int image_is_f64, int n_frames, int height,
                           int width, int pad_mode, float pad_value, void* const* outs_host, int out_is_f64);
/* Page-locked host memory (hipHostMalloc).  float32 frames kept in it - acquisition buffers, result rings - are read and
 * written by the copy engines directly: the host-array entry points above detect such pointers (hipPointerGetAttributes)
 * and skip the staging copy of the pageable path for every side that needs no dtype conversion. */
int rpsf_host_alloc(int device, size_t bytes, void** out);
int rpsf_host_free(void* ptr);
/* Width of the worker pool the host-array entry points convert on (creates it if this is the first use).  The workers are
 * pinned to cores of the NUMA node the (first) device hangs off, spread over its core complexes (RPSF_HOST_AFFINITY=0: not
 * pinned; RPSF_HOST_THREADS: width). */
int rpsf_host_threads(int* threads);
/* Self-test of that pool (no GPU involved; the CPU test suite runs it): `jobs` jobs of up to `parts` parts from each of `callers`
 * threads at once, every part checked to have run exactly once; *failures = number of jobs that came out wrong. */
int rpsf_host_pool_selftest(int callers, int jobs, int parts, int* failures);
/* That node (-1: unknown): a process that feeds the GPU from host arrays should run and allocate there - staging copies
 * from the other socket run at a third of the rate (scripts/micro/host_copy.hip). */
int rpsf_device_numa_node(int device, int* node);
/* What PCIe gives `bytes` on this box, from and to pinned host memory: one direction at a time and both at once (two
 * streams), best of `iters`, milliseconds.  The floor the streamed entry points are reported against (SURVEY 8d:
 * end-to-end figures are reported separately from the device-resident ones).  Any pointer may be NULL. */
int rpsf_pcie_probe(int device, size_t bytes, int iters, double* h2d_ms, double* d2h_ms, double* duplex_ms);
/* Device variant: frame f is at images_dev + f * image_stride and outs_dev + f * out_stride (strides in
 * floats); asynchronous on `stream` (NULL: the plan's own).  The plan keeps 16 bytes of scratch per output
 * pixel per frame in flight (large batches are cut into groups internally). */
int rpsf_apply_batch_device(rpsf_plan* plan, const void* images_dev, void* outs_dev, int n_frames, size_t image_stride,
                            size_t out_stride, const rpsf_geometry* geom, void* stream);
/* Timed variant, like rpsf_apply_device_timed: total_ms[i] covers the whole batch, kernel_ms[i] its patch kernel. */
int rpsf_apply_batch_device_timed(rpsf_plan* plan, const void* images_dev, void* outs_dev, int n_frames,
                                  size_t image_stride, size_t out_stride, const rpsf_geometry* geom, int iters,
                                  float* total_ms, float* kernel_ms);
/* The plan's own stream (hipStream_t), for callers that enqueue follow-up work such as the seam exchange. */
void* rpsf_plan_stream(rpsf_plan* plan);

/* ArrayPSFTransform.construct arithmetic, transform.py:78-82:
 *   K = conj(S) |S|^(alpha-1) / (|S|^(alpha+1) + (eps |T|)^(alpha+1)) * T
 * element-wise over `count` complex values; S, T, K are host arrays of complex64 (is_f64 = 0,
 * evaluated in float32 like NumPy does for complex64 input) or complex128 (is_f64 = 1). */
int rpsf_build_transfer(int device, size_t count, const void* s_host, const void* t_host, int is_f64, double alpha,
                        double epsilon, void* k_host);
/* Device-resident variant (pointers on `device`), asynchronous on stream. */
int rpsf_build_transfer_device(int device, size_t count, const void* s_dev, const void* t_dev, int is_f64,
                               double alpha, double epsilon, void* k_dev, void* stream);

/* Batched un-shifted 2-D FFT of real PSF cubes (ArrayPSF.__init__, psf.py:216-219), float32 in,
 * complex64 out, (count, N, N) host arrays. */
int rpsf_psf_fft(int device, int patch_size, int count, const float* values_host, float* fft_c64_host);
/* Same, leaving the spectra on the device (count * N * N complex64 at fft_c64_dev, allocated by the caller): the input
 * of rpsf_build_transfer_device, so that ArrayPSF -> construct (psf.py:216-219 -> transform.py:78-82) -> apply never
 * moves a spectrum or K over PCIe. */
int rpsf_psf_fft_device(int device, int patch_size, int count, const float* values_host, void* fft_c64_dev);
/* Built-in parametric PSF models rasterised on the device (kernel K6) and transformed there: replaces the host loop of
 * VariedFunctionalPSF.as_array_psf (regularizepsf/psf.py:159-165: one Python call of the model per patch on an N x N
 * meshgrid) followed by ArrayPSF.__init__ (psf.py:216-219) for models the library knows, so that model parameters ->
 * samples -> spectra -> K -> apply never leaves the GPU.  params_host: count x RPSF_MODEL_PARAMS doubles, per model
 *   RPSF_MODEL_ELLIPTICAL_GAUSSIAN: amplitude, row0, col0, sigma_row, sigma_col, theta, background, unused
 *   RPSF_MODEL_MOFFAT:              amplitude, row0, col0, alpha, beta, unused, background, unused
 * Element [i][j] of a patch is the model at (row = j, col = i), as the reference's meshgrid hands it over.  normalize != 0
 * scales every patch to unit sum.  values_f32_dev (optional, count * N * N floats) keeps the samples, fft_c64_dev
 * (count * N * N complex64) receives the spectra; both allocated by the caller on `device`. */
#define RPSF_MODEL_PARAMS 8
#define RPSF_MODEL_ELLIPTICAL_GAUSSIAN 0
#define RPSF_MODEL_MOFFAT 1
int rpsf_psf_model_fft_device(int device, int model, int patch_size, int count, const double* params_host, int normalize,
                              void* values_f32_dev, void* fft_c64_dev);

/* Host-side helper for the saturation branch of apply (transform.py:135-138): sequential, row-major
 * NaN-ignoring neighbourhood-mean fill of the masked pixels of the float64 padded image (no GPU involved). */
int rpsf_saturation_fill(double* padded, int rows, int cols, const uint8_t* mask, int neighborhood_width);

/* Device memory helpers for callers that keep frames resident (bench, tests, streaming). */
int rpsf_dev_alloc(int device, size_t bytes, void** out);
int rpsf_dev_free(int device, void* ptr);
int rpsf_memcpy_h2d(int device, void* dst_dev, const void* src_host, size_t bytes);
int rpsf_memcpy_d2h(int device, void* dst_host, const void* src_dev, size_t bytes);
int rpsf_device_synchronize(int device);

/* Multi-GPU row-band sharding (one process per GPU).  The caller splits the patch lattice into
 * contiguous row bands; each rank builds a plan for its band only (image rows it reads, patches it
 * owns) and, after the local apply, exchanges the seam rows its last lattice row spilled into the
 * next rank's band.  rpsf_comm_* wraps RCCL (loaded with dlopen) for exactly that neighbour
 * exchange; unique_id is 128 bytes, created on rank 0 and distributed by the caller. */
typedef struct rpsf_comm rpsf_comm;
int rpsf_comm_unique_id(void* id128);
int rpsf_comm_create(rpsf_comm** out, int device, int rank, int world, const void* id128);
void rpsf_comm_destroy(rpsf_comm* comm);
/* Send `send_count` floats at send_dev to rank+1 (if any) and receive `recv_count` floats from
 * rank-1 (if any) into recv_dev, then add recv_dev[0:recv_count] into accum_dev.  Any count may be 0. */
int rpsf_comm_seam_exchange_add(rpsf_comm* comm, const void* send_dev, size_t send_count, void* recv_dev,
                                size_t recv_count, void* accum_dev, void* stream);
/* The exchange alone, without the add - for callers that compute the spill rows first, on a stream of their own,
 * and let the transfer run beside the rest of the band (regularizepsf_amd/sharding.py).  stream NULL: the
 * communicator's own stream (rpsf_comm_stream). */
int rpsf_comm_seam_exchange(rpsf_comm* comm, const void* send_dev, size_t send_count, void* recv_dev, size_t recv_count,
                            void* stream);
void* rpsf_comm_stream(rpsf_comm* comm);
/* The number of ranks RCCL reports for the communicator (ncclCommCount) - evidence for a harness that the seam exchange
 * spans the GPUs it was launched on. */
int rpsf_comm_ranks(rpsf_comm* comm, int* ranks);
/* Make `waiter` (a hipStream_t) wait for everything enqueued on `signaller` so far. */
int rpsf_stream_wait(int device, void* waiter, void* signaller);
/* Events (hipEvent_t without timing) for dependencies that span steps: record on one stream now, make another stream wait for
 * it later (the sharded step's double-buffered seam rows). */
int rpsf_event_create(int device, void** event);
int rpsf_event_record(void* event, void* stream);
int rpsf_stream_wait_event(void* stream, void* event);
int rpsf_event_destroy(void* event);
/* accum_dev[0:count] += src_dev[0:count] (float32, on `device`, asynchronous on stream): the add of the seam
 * exchange by itself, for callers that move the seam rows with their own transport. */
int rpsf_add_rows(int device, void* accum_dev, const void* src_dev, size_t count, void* stream);
/* The same add on at most max_workgroups workgroups of 256 threads: for a caller that runs it beside a persistent patch launch
 * (the pipelined seam exchange of regularizepsf_amd/sharding.py), whose workgroups need whole CUs. */
int rpsf_add_rows_narrow(int device, void* accum_dev, const void* src_dev, size_t count, int max_workgroups, void* stream);
int rpsf_comm_barrier(rpsf_comm* comm, void* stream);
/* max over ranks of one double (used for whole-job timing) */
int rpsf_comm_allreduce_max(rpsf_comm* comm, double* value);

#ifdef __cplusplus
}
#endif
#endif /* RPSF_H */
