/*
 * fasttrack_amd.h - C ABI of the MI355X-native ORB tracking front end.
 *
 * This is the drop-in boundary: every entry point replaces one seam of the reference
 * (sfu-rsl/FastTrack, paths relative to its root) and takes plain pointers and sizes only - no
 * OpenCV / Eigen / torch types - so it binds from C++, C or ctypes alike.  INTEGRATION.md shows the
 * adapter a maintainer adds on the ORB-SLAM3 side.
 *
 * Conventions
 *  - every function returns FT_OK (0) or a negative ft_status; ft_last_error() returns the message of
 *    the calling thread's last failure (the reference prints and exit()s: src/Kernels/CudaUtils.cu:17-22).
 *  - handles are bound to one device; calls on different handles are re-entrant (the reference calls
 *    the left and right extractor from two host threads: src/Frame.cc:127-130); calls on the same
 *    handle must be serialised by the caller.
 *  - all compute runs in hand-written HIP kernels on the handle's device.  There is no CPU fallback:
 *    without a usable gfx950 device every compute entry point fails with FT_ERR_NO_DEVICE.
 *  - ft_keypoint has the field order and size (28 B) of cv::KeyPoint, so a std::vector<cv::KeyPoint>
 *    can be passed as ft_keypoint* directly.
 */
#ifndef FASTTRACK_AMD_H
#define FASTTRACK_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FT_API __attribute__((visibility("default")))

typedef enum ft_status {
    FT_OK = 0,
    FT_ERR_INVALID = -1,   /* bad argument */
    FT_ERR_NO_DEVICE = -2, /* no usable HIP device / extension not functional */
    FT_ERR_HIP = -3,       /* a HIP runtime call failed */
    FT_ERR_CAPACITY = -4,  /* caller buffer too small */
    FT_ERR_EMPTY = -5      /* empty input image (ORBextractor::operator() returns -1, ORBextractor.cc:1360) */
} ft_status;

/* cv::KeyPoint layout: pt.x pt.y size angle response octave class_id */
typedef struct ft_keypoint {
    float x, y, size, angle, response;
    int octave, class_id;
} ft_keypoint;

/* ------------------------------------------------------------------------------------------------
 * Library / device context.
 * Replaces KernelController::setCUDADevice / initializeKernels / shutdownKernels / saveKernelsStats
 * (include/Kernels/KernelController.h:15-29) and CudaUtils::loadSetting (include/Kernels/CudaUtils.h:21).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ft_context ft_context;

FT_API const char *ft_version(void);
FT_API const char *ft_last_error(void);
FT_API int ft_device_count(void); /* number of HIP devices, 0 if none (never fails) */
/* PCI bus id ("0000:c1:00.0") of a device: /sys/bus/pci/devices/<id>/numa_node names the host memory node next to it, so a
 * process that serves one GPU (reference: one KernelController per process, src/Kernels/KernelController.cu:24-29) can pin
 * its host threads there before it creates the context */
FT_API int ft_device_pci_bus_id(int device, char *buf, int len);
/* host_threads: workers for the host-side octree stage (0 = the CPUs this process may use:
 * hardware threads capped by affinity and by a cgroup CPU quota).
 * No side effects on the process environment.  Extractors for batches of more than 16 frames run on streams of the context
 * ("lanes") by a table made for the number of hardware queues the HIP runtime gives the process: GPU_MAX_HW_QUEUES, which
 * the runtime reads ONCE, at the first HIP call of the process.  An application that wants the eight-lane table (+10 % on
 * the headline workload over the four queues of the runtime's default) sets GPU_MAX_HW_QUEUES=10 before its first HIP call
 * (INTEGRATION.md section 6); ft_context_hw_queues() reports what the context was created under (4 = variable unset).
 * FT_LANE_MAP ("own" or whole sets of four lane numbers) overrides the table; a malformed value is FT_ERR_INVALID.
 * ft_context_destroy refuses (FT_ERR_INVALID) while extractors, front ends or tracked frames of the context are alive:
 * they run on its streams. */
FT_API int ft_context_create(int device, int host_threads, ft_context **out);
FT_API int ft_context_destroy(ft_context *ctx);
FT_API int ft_context_synchronize(ft_context *ctx);
FT_API int ft_context_device_name(ft_context *ctx, char *buf, int len);
FT_API int ft_context_hw_queues(const ft_context *ctx);
/* The lane table of the extractors created on the context from now on: n = 4 * sets entries, (stage A, stage B, octree 0,
 * octree 1) lane numbers in [0, 64) for the 1st, 2nd, ... extractor (wrapping around); n = 0: private streams for every
 * extractor.  What FT_LANE_MAP does through the environment; tools/lane_search.py finds a table for a given stream of frames. */
FT_API int ft_context_set_lane_map(ft_context *ctx, const int *map, int n);
/* Tuning options.  Every switch of the library that is not a debugging aid is a named integer option of the CONTEXT.
 * ft_context_create reads the initial values from the environment, once (variable = "FT_" + upper-case name):
 *
 *   name                 default  range    meaning
 *   pipeline_depth       0        0 .. 8   sub-batches a throughput batch is enqueued as (0 = automatic, 1 = no pipelining)
 *   device_octree        1        0 .. 1   DistributeOctTree on the device (0 = host thread pool)
 *   oct_hist             1        0 .. 1   histogram tier of the device octree for levels above 4096 candidates
 *   oct_hist_first       1        0 .. 2   histogram formulation for every level: 0 never, 1 latency-mode launches, 2 always
 *   oct_big              1        0 .. 1   sorted big tier of the device octree (up to 16384 keys per level)
 *   pyr_rows             1        0 .. 1   pyramid of launches of 8+ images: 1 = row-streaming kernel, 0 = tile kernel
 *   upload_kernel        1        0 .. 1   latency mode: frames go up through one kernel instead of DMA copies
 *   graph                1        0 .. 1   latency mode: batches of <= 8 frames are captured and replayed as HIP graphs
 *   paired               1        0 .. 1   latency-mode stereo front ends run both cameras through one set of launches
 *   pass_burst           12       2 .. 14  projection searches: claim passes enqueued per host round trip
 *   search_cache         2        0 .. 3   projection searches: 1 = later claim passes walk the cached candidate keys; 2 = and a
 *                                          batch of 24 or more frames resolves its claims in one launch behind the first pass (a
 *                                          workgroup per frame); 3 = every batch does
 *   search_grid          1        0 .. 1   projection searches: CSR grid of the frame built on the device
 *   blocking_sync        2        0 .. 2   host waits for the device: 0 = the runtime's default (it spins), 1 = sleeping
 *                                          (hipDeviceScheduleBlockingSync: a waiting thread leaves its core to the others - 10 - 20 us
 *                                          later wake-ups), 2 = sleeping when the process may use fewer than 8 CPUs.  Read by
 *                                          ft_context_create only (a device flag of the whole process)
 *
 * A value outside an option's range is FT_ERR_INVALID (from ft_context_set_option, and from ft_context_create when it comes
 * from the environment).
 * ft_context_set_option changes the context's value; extractors, front ends and tracked frames take the switches of their
 * context when they are CREATED (the search_* and pass_burst options are read per call).  `name` is the option name or its
 * environment spelling.  ft_option_describe enumerates the table (index 0, 1, ... until FT_ERR_INVALID).
 * Only the FT_DEBUG_* aids (FT_DEBUG_REPEAT, FT_DEBUG_OCC, FT_DEBUG_FAST, FT_DEBUG_OD_PROFILE, FT_DEBUG_OCT_PROFILE, and
 * FT_DEBUG_OCTREE_PATHS / _HIST_BINS / _HIST_STRICT of the host test entry ft_octree_distribute) and FT_LANE_MAP are read from
 * the environment anywhere else; none of them changes results. */
FT_API int ft_context_set_option(ft_context *ctx, const char *name, int value);
FT_API int ft_context_get_option(const ft_context *ctx, const char *name, int *value);
FT_API int ft_option_describe(int index, const char **name, const char **env, int *default_value, const char **doc);
FT_API int ft_option_range(const char *name, int *min_value, int *max_value);
FT_API int ft_context_host_threads(const ft_context *ctx);
/* per-stage wall/GPU timings of the calls made so far; the reference's REGISTER_STATS analogue
 * (include/Kernels/CudaUtils.h:14, src/Stats.cc:31-60).  Writes "<name>: <ms>" lines. */
FT_API int ft_context_save_stats(ft_context *ctx, const char *path);
/* kernel timing with HIP events on the launching stream (off by default); names are "kernel.<name>" */
FT_API int ft_context_set_kernel_timing(ft_context *ctx, int enabled);
FT_API int ft_context_get_stat(ft_context *ctx, const char *name, double *total_ms, long *calls);
FT_API int ft_context_reset_stats(ft_context *ctx);
/* device memory helpers so that a caller (or bench.py) can keep frames resident in HBM */
FT_API int ft_device_malloc(ft_context *ctx, size_t bytes, void **dptr);
FT_API int ft_device_free(ft_context *ctx, void *dptr);
/* pinned host memory: result arrays allocated here are filled by the device copies directly.  A block is released by
 * ft_host_free, or with its context: ft_context_destroy frees what is still allocated (do not use such a block afterwards). */
FT_API int ft_host_malloc(ft_context *ctx, size_t bytes, void **ptr);
FT_API int ft_host_free(ft_context *ctx, void *ptr);
FT_API int ft_memcpy_h2d(ft_context *ctx, void *dst, const void *src, size_t bytes);
FT_API int ft_memcpy_d2h(ft_context *ctx, void *dst, const void *src, size_t bytes);

/* ------------------------------------------------------------------------------------------------
 * ORB extractor.  Replaces ORB_SLAM3::ORBextractor (include/ORBextractor.h:100-197):
 *   ctor (nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, imageWidth, imageHeight)  :105-106
 *   int operator()(image, mask, keypoints, descriptors, vLappingArea)                       :113-115
 *   GetLevels/GetScaleFactor(s)/GetInverseScaleFactors/GetScaleSigmaSquares/...             :117-144
 *   mvImagePyramid (:146) and GetGPUPyramid() (:126-128)
 * Results equal the reference's CPU branch (KernelController::orbExtractionKernelRunStatus == 0).
 * max_batch image slots are allocated; slot b keeps the pyramid of the b-th image of the last call
 * resident in HBM for the stereo matcher.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ft_extractor ft_extractor;

FT_API int ft_extractor_create(ft_context *ctx, int nfeatures, float scale_factor, int nlevels, int ini_th_fast,
                               int min_th_fast, int image_width, int image_height, int max_batch,
                               ft_extractor **out);
FT_API int ft_extractor_destroy(ft_extractor *ex);
FT_API int ft_extractor_levels(const ft_extractor *ex);
FT_API int ft_extractor_max_batch(const ft_extractor *ex);
/* upper bound of keypoints operator() can return per image (octree may exceed nfeatures slightly) */
FT_API int ft_extractor_max_keypoints(const ft_extractor *ex);
/* any of the four output arrays may be NULL; each holds nlevels floats */
FT_API int ft_extractor_scale_factors(const ft_extractor *ex, float *scale, float *inv_scale, float *sigma2,
                                      float *inv_sigma2);
FT_API int ft_extractor_features_per_level(const ft_extractor *ex, int *quota);
FT_API int ft_extractor_level_size(const ft_extractor *ex, int level, int *width, int *height);

/* ORBextractor::operator() for one host image (row stride in bytes).  keypoints/descriptors receive
 * up to `capacity` entries laid out exactly as the reference does (ORBextractor.cc:1466-1487):
 * keypoints with x in [lap0, lap1] fill from the back, the others from the front; *n_mono is the
 * reference's return value.  FT_ERR_EMPTY for a null/empty image. */
FT_API int ft_extract(ft_extractor *ex, const uint8_t *image, int width, int height, int stride, int lap0,
                      int lap1, ft_keypoint *keypoints, uint8_t *descriptors, int capacity, int *n_keypoints,
                      int *n_mono);

/* Modes of an extractor / front end, fixed by max_batch at creation: max_batch <= 16 = LATENCY mode - private streams, and a
 * batch of at most 8 frames with a fixed call shape is captured once as a HIP graph and replayed (statistics
 * "extract.graph_launch" / "stereo.graph_launch" count the replays); max_batch > 16 = THROUGHPUT mode - the streams are lanes
 * of the context shared with the other wide extractors, which a capture cannot use, so such an object never takes the graph
 * path, whatever the size of the batch it is handed (FT_LANE_MAP=own gives wide extractors private streams and the graph path
 * back).  A caller that alternates between single frames and large batches creates one object for each. */

/* The same for `batch` <= max_batch images of identical size.  images[b] is a host pointer
 * (on_device = 0) or a device pointer on this context's device (on_device = 1, frames already in HBM).
 * Outputs are host arrays: keypoints[b*capacity + i], descriptors[(b*capacity + i)*32].  Arrays in pinned memory of the
 * context (ft_host_malloc) are written by the device itself, in the order above, for batches of more than eight images (no
 * staging copy and no pass of the host over the keypoints; statistic "extract.delivered_in_order_on_device"); any other
 * array is filled by the host from the library's staging.  The results are the same. */
FT_API int ft_extract_batch(ft_extractor *ex, const uint8_t *const *images, int batch, int on_device, int width,
                            int height, int stride, int lap0, int lap1, ft_keypoint *keypoints,
                            uint8_t *descriptors, int capacity, int *n_keypoints, int *n_mono);

/* mvImagePyramid[level] of slot `slot` copied to host (tight or strided rows) */
FT_API int ft_extractor_download_level(ft_extractor *ex, int slot, int level, uint8_t *dst, int dst_stride);
/* stage tap for parity tests: level `level` of slot `slot` after cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101)
 * (ORBextractor.cc:1456-1457), computed by the blur routines the descriptor kernel applies on the fly (the path itself
 * never materialises a blurred level) */
FT_API int ft_extractor_download_blurred_level(ft_extractor *ex, int slot, int level, uint8_t *dst, int dst_stride);
/* GetGPUPyramid(): device pointer + pitch of one level of one slot */
FT_API int ft_extractor_device_level(ft_extractor *ex, int slot, int level, const uint8_t **dptr, int *pitch);
/* stage taps for parity tests: FAST candidates handed to the octree, in the CPU's emission order,
 * as (x, y, score) triples relative to the level's (minBorderX, minBorderY) (ORBextractor.cc:1196-1198) */
FT_API int ft_extractor_download_candidates(ft_extractor *ex, int slot, int level, int *xys, int capacity,
                                            int *n);

/* stage tap for parity tests: the DEVICE formulation of DistributeOctTree (ORBextractor.cc:660-884) of one level on
 * caller-provided candidates ((x, y, score) triples relative to the level's (minBorderX, minBorderY), any order: the
 * kernels rank them by the reference's emission order themselves).  tiers: bit 1 allows the histogram tier
 * (k_octree_hist), bit 2 the sorted big tier (k_octree_big), bit 3 selects the kernel of latency-mode launches
 * (k_octree_auto: reported as tier 2); *tier = the tier that produced the result (1, 2, 3), 0 = the
 * level is beyond the allowed tiers (the pipeline would repair the image with the host octree).  out_xys: the retained
 * candidates in the reference's result order. */
FT_API int ft_extractor_octree_on_device(ft_extractor *ex, int level, const int *xys, int n, int tiers, int *out_xys,
                                         int capacity, int *n_out, int *tier);

/* Host-only stages, callable without a GPU (they never touch pixels):
 * ft_octree_distribute = ORBextractor::DistributeOctTree (ORBextractor.cc:660-884), the serial stage
 * that stays on the host in both branches of the reference.  xys: n (x, y, score) candidates relative to
 * (minX, minY) in FAST emission order; out_idx receives the retained candidate indices in the
 * reference's result order; *n_out their number (may exceed N by a few, as in the reference).
 * ft_level_geometry = the level / cell-grid / quota arithmetic of ORBextractor.cc:398-465,1120-1134,1499-1500;
 * every output array holds nlevels ints and may be NULL. */
FT_API int ft_octree_distribute(const int *xys, int n, int minX, int maxX, int minY, int maxY, int N, int *out_idx,
                                int capacity, int *n_out);
FT_API int ft_level_geometry(int width, int height, int nfeatures, float scale_factor, int nlevels, int *level_w,
                             int *level_h, int *quota, int *n_cols, int *n_rows, int *w_cell, int *h_cell);

/* ------------------------------------------------------------------------------------------------
 * Stereo matching, rectified pinhole pair.
 * Replaces KernelController::launchStereoMatchKernel (include/Kernels/KernelController.h:31-36) as
 * called from Frame::ComputeStereoMatchesGPU (src/Frame.cc:1007-1063); semantics follow the CPU
 * branch Frame::ComputeStereoMatches (src/Frame.cc:835-1005): row-band Hamming search, 11x11 SAD at
 * 11 shifts, parabola fit, and (apply_median_cut != 0) the 1.5*1.4*median SAD cut.
 * The pyramids are those resident in slot `slot` of exL / exR from their last extract call.
 * uright/depth: mvuRight / mvDepth (-1 = no match); sad (may be NULL): best SAD of kept matches, -1 else.
 * ---------------------------------------------------------------------------------------------- */
FT_API int ft_stereo_match(ft_extractor *exL, ft_extractor *exR, int slot, const ft_keypoint *keysL, int nL,
                           const ft_keypoint *keysR, int nR, const uint8_t *descL, const uint8_t *descR,
                           float mbf, float mb, int apply_median_cut, float *uright, float *depth, int *sad,
                           int *n_matches);

/* Fused stereo front end for throughput: extract left + right and stereo-match `batch` pairs with
 * keypoints and descriptors kept on the device between the stages (one stream per GPU).
 * Outputs per pair b (host arrays, `capacity` entries per image): keysL/descL/keysR/descR as
 * ft_extract_batch, uright/depth for the left keypoints. */
typedef struct ft_stereo_frontend ft_stereo_frontend;
FT_API int ft_stereo_frontend_create(ft_context *ctx, int nfeatures, float scale_factor, int nlevels,
                                     int ini_th_fast, int min_th_fast, int image_width, int image_height,
                                     int max_batch, float mbf, float mb, ft_stereo_frontend **out);
FT_API int ft_stereo_frontend_destroy(ft_stereo_frontend *fe);
FT_API ft_extractor *ft_stereo_frontend_left(ft_stereo_frontend *fe);
FT_API ft_extractor *ft_stereo_frontend_right(ft_stereo_frontend *fe);
/* Asynchronous halves of ft_stereo_frontend_process: submit enqueues everything (it returns once the host
 * octree of the last sub-batch is done and its kernels are queued), wait drains the device and finishes
 * the outputs.  With two front ends used alternately the drain of one batch overlaps with the next batch:
 *   submit(fe[k & 1], batch k);  wait(fe[(k - 1) & 1]);
 * Output arrays of a submitted batch must stay untouched until its wait returns.
 * INPUT FRAMES of a submitted batch must stay valid AND UNCHANGED until its wait returns, whatever memory they live
 * in: frames resident in HBM (on_device) and frames in pinned host memory (ft_host_malloc / hipHostMalloc /
 * hipHostRegister) are read in place by the device after submit has returned; only pageable host frames are copied
 * before submit returns.  A camera ring buffer must therefore not recycle a slot between submit and wait. */
FT_API int ft_stereo_frontend_submit(ft_stereo_frontend *fe, const uint8_t *const *imagesL,
                                     const uint8_t *const *imagesR, int batch, int on_device, int width, int height,
                                     int stride, ft_keypoint *keysL, uint8_t *descL, int *nL, ft_keypoint *keysR,
                                     uint8_t *descR, int *nR, int capacity, float *uright, float *depth,
                                     int *n_matches);
FT_API int ft_stereo_frontend_wait(ft_stereo_frontend *fe);
FT_API int ft_stereo_frontend_process(ft_stereo_frontend *fe, const uint8_t *const *imagesL,
                                      const uint8_t *const *imagesR, int batch, int on_device, int width,
                                      int height, int stride, ft_keypoint *keysL, uint8_t *descL, int *nL,
                                      ft_keypoint *keysR, uint8_t *descR, int *nR, int capacity, float *uright,
                                      float *depth, int *n_matches);
/* the descriptors of pair `slot` of the batch processed last, where the front end left them in HBM (n x 32 bytes in
 * the order of the host results; valid until the front end processes another batch): input of ft_bow_transform
 * (Frame::ComputeBoW) without a copy */
FT_API int ft_stereo_frontend_device_descriptors(ft_stereo_frontend *fe, int slot, int right, const uint8_t **dptr, int *n);

/* ------------------------------------------------------------------------------------------------
 * Fisheye stereo matching (matching part).
 * Replaces KernelController::launchFisheyeStereoMatchKernel(N, Nr, descL, descR, matches)
 * (include/Kernels/KernelController.h:38); semantics follow Frame::ComputeStereoFishEyeMatches
 * (src/Frame.cc:1231-1255): BFMatcher(NORM_HAMMING).knnMatch(k=2) + Lowe ratio 0.7.  The caller passes
 * the lapping-area subsets like the CPU branch (src/Frame.cc:1233-1237).  matches[i] = index into
 * descR or -1; best/second (may be NULL) = the two smallest distances.
 * ---------------------------------------------------------------------------------------------- */
FT_API int ft_fisheye_match(ft_context *ctx, const uint8_t *descL, int nL, const uint8_t *descR, int nR,
                            int *matches, int *best, int *second);

/* Complete Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1231-1271): the 2-NN + ratio test
 * above followed, on the device, by KannalaBrandt8::TriangulateMatches per surviving pair
 * (src/CameraModels/KannalaBrandt8.cpp:306-372: Newton unprojection, parallax, linear triangulation, depth and
 * reprojection tests).  keysL / keysR are the keypoints of the lapping-area subsets descL / descR refer to.
 * matches[i] = index into the right subset or -1 (mvLeftToRightMatch), depth[i] = mvDepth or -1,
 * p3d[3i..3i+2] = mvStereo3Dpoints.  The null vector of the 4x4 triangulation system is computed by a one-sided
 * Jacobi SVD in double (the reference uses Eigen::JacobiSVD in float): depth / p3d agree with the reference to
 * its own rounding error, not bit for bit.  Every other float step is the reference's operation for operation,
 * including tanf / atan2f / cosf / sinf as the host's glibc evaluates them (ft_selftest_libm). */
typedef struct ft_fisheye_rig {
    float cam1[8], cam2[8]; /* fx fy cx cy k1 k2 k3 k4 of mpCamera / mpCamera2 */
    float precision;        /* KannalaBrandt8::precision */
    float Rlr[9], tlr[3];   /* mRlr (row-major), mtlr */
} ft_fisheye_rig;
FT_API int ft_fisheye_stereo(ft_context *ctx, const ft_fisheye_rig *rig, const uint8_t *descL, const ft_keypoint *keysL,
                             int nL, const uint8_t *descR, const ft_keypoint *keysR, int nR, const float *level_sigma2,
                             int nlevels, int *matches, float *depth, float *p3d, int *n_matches);

/* ------------------------------------------------------------------------------------------------
 * Frame view for the projection matchers: the fields of ORB_SLAM3::Frame that
 * DATA_WRAPPER::CudaFrame::setMemory marshals (src/Kernels/CudaWrappers/CudaFrame.cu:77-181).
 * The 64x48 grid (Frame::AssignFeaturesToGrid, src/Frame.cc:409-440) is rebuilt on the device.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ft_frame_view {
    int N;     /* all keypoints (left + right when Nleft != -1) */
    int Nleft; /* -1: mono / rectified stereo (keys = mvKeysUn); else keys = mvKeys, keys_right = mvKeysRight */
    float mnMinX, mnMinY, mnMaxX, mnMaxY;
    float grid_inv_w, grid_inv_h; /* mfGridElementWidthInv / mfGridElementHeightInv */
    float mbf, mb;
    const ft_keypoint *keys;
    const ft_keypoint *keys_right;
    const uint8_t *descriptors; /* N x 32 (left rows then right rows) */
    const float *uright;        /* mvuRight[N] when Nleft == -1, NULL = all -1 */
    int *holder_obs;            /* in/out [N]: Observations() of mvpMapPoints[i], -1 when NULL */
    const int *left_to_right;   /* mvLeftToRightMatch[Nleft] or NULL */
    const int *right_to_left;   /* mvRightToLeftMatch[N-Nleft] or NULL */
    int cam_model;              /* 0 Pinhole, 1 KannalaBrandt8 */
    float cam[8];               /* fx fy cx cy k1 k2 k3 k4 */
    float Trl[12];              /* row-major 3x4 GetRelativePoseTrl() */
    const float *scale_factors; /* mvScaleFactors[nlevels] */
    int nlevels;
} ft_frame_view;

/* Frame::GetFeaturesInArea (src/Frame.cc:681-747) for nq queries at once, with the 64x48 grid of
 * Frame::AssignFeaturesToGrid / PosInGrid (:409-440, :749-759) evaluated on the device.  Query i = (x[i], y[i], r[i],
 * min_level[i], max_level[i], right[i]) (right may be NULL: left camera).  indices receives, per query, the keypoint
 * indices in the reference's order (cell column, cell row, then insertion order) at indices[i * capacity ...];
 * counts[i] = number of hits; when it exceeds capacity the first `capacity` hits in that order are stored. */
FT_API int ft_features_in_area(ft_context *ctx, const ft_frame_view *F, int nq, const float *x, const float *y,
                               const float *r, const int *min_level, const int *max_level, const uint8_t *right,
                               int *indices, int capacity, int *counts);

/* SoA arrays of the local map points, exactly what SearchLocalPointsKernel::launch builds
 * (src/Kernels/SearchLocalPointsKernel.cu:369-390) plus Observations(). */
typedef struct ft_local_points {
    int M;
    const uint8_t *skip; /* (!mbTrackInView && !mbTrackInViewR) || (bFarPoints && depth > thFar) || isBad() */
    const uint8_t *in_view, *in_view_r;
    const int *level, *level_r;           /* mnTrackScaleLevel, mnTrackScaleLevelR */
    const float *view_cos, *view_cos_r;   /* mTrackViewCos, mTrackViewCosR */
    const float *proj_x, *proj_y;         /* mTrackProjX, mTrackProjY */
    const float *proj_xr, *proj_yr;       /* mTrackProjXR, mTrackProjYR */
    const uint8_t *descriptors;           /* M x 32 */
    const int *observations;              /* pMP->Observations() */
} ft_local_points;

/* Replaces KernelController::launchSearchLocalPointsKernel (include/Kernels/KernelController.h:40-42)
 * AND the acceptance loop around it (src/ORBmatcher.cc:241-308); semantics follow the CPU branch of
 * ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>, th, bFarPoints, thFarPoints)
 * (src/ORBmatcher.cc:49-225) including its in-call claiming order.
 * assign[i] (size F->N) = index of the map point written to F.mvpMapPoints[i], else -1;
 * F->holder_obs is updated like mvpMapPoints.  The ten raw arrays (size M, any may be NULL) are the
 * reference kernel's outputs.  *n_matches = the reference's return value. */
FT_API int ft_search_local_points(ft_context *ctx, ft_frame_view *F, const ft_local_points *P, float th,
                                  float nn_ratio, int *assign, int *n_matches, int *best_dist, int *best_dist2,
                                  int *best_level, int *best_level2, int *best_idx, int *best_dist_r,
                                  int *best_dist2_r, int *best_level_r, int *best_level2_r, int *best_idx_r);

/* Last-frame map points for the motion-model search. */
typedef struct ft_last_points {
    int N;                      /* LastFrame.N */
    const uint8_t *valid;       /* mvpMapPoints[i] != NULL && !mvbOutlier[i] */
    const float *world_pos;     /* N x 3, pMP->GetWorldPos() */
    const uint8_t *descriptors; /* N x 32, pMP->GetDescriptor() */
    const int *observations;
    const int *octave; /* octave of last-frame keypoint i */
    const float *angle; /* angle of last-frame keypoint i (rotation histogram) */
} ft_last_points;

/* Replaces KernelController::launchPoseEstimationKernel (include/Kernels/KernelController.h:44-46) and
 * the histogram loop around it (src/ORBmatcher.cc:2013-2081); semantics follow the CPU branch of
 * ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (src/ORBmatcher.cc:1775-1990).
 * Tcw: row-major 3x4 current pose; forward/backward as computed at :1794-1795.
 * assign[i] (size Cur->N) = last-frame index whose map point ends up in CurrentFrame.mvpMapPoints[i]. */
FT_API int ft_search_last_frame(ft_context *ctx, ft_frame_view *Cur, const ft_last_points *L, const float *Tcw,
                                float th, int forward, int backward, int check_orientation, int *assign,
                                int *n_matches, int *best_dist, int *best_idx, int *best_dist_r,
                                int *best_idx_r);

/* The same search with the poses in the form the reference's CPU branch multiplies with.  CurrentFrame.GetPose() and
 * GetRelativePoseTrl() are Sophus::SE3f: a unit quaternion and a translation, applied to a point as
 * Thirdparty/Sophus/sophus/so3.hpp:358-367 + se3.hpp:321-324 do (uv = 2 (q.vec x p); p + w uv + q.vec x uv; + t) - NOT as a
 * matrix product; the two differ in the last bits of the projected point, i.e. in whether a keypoint that sits within ~1e-4 px
 * of a search window's edge is a candidate.  ft_search_last_frame (3x4 matrices, y = R x + t) is the form the reference's GPU
 * boundary takes (Eigen::Matrix4f transform_matrix, include/Kernels/KernelController.h:44-46); this one reproduces
 * `Tcw * x3Dw` (src/ORBmatcher.cc:1805) and `GetRelativePoseTrl() * x3Dc` (:1900) of the CPU branch operation for operation.
 * q = Eigen::Quaternionf::coeffs() = (x, y, z, w) of so3().unit_quaternion(); Trl may be NULL when Cur->Nleft == -1 (Cur->Trl is
 * not read by this call). */
typedef struct ft_se3 {
    float q[4];
    float t[3];
} ft_se3;
FT_API int ft_search_last_frame_se3(ft_context *ctx, ft_frame_view *Cur, const ft_last_points *L, const ft_se3 *Tcw,
                                    const ft_se3 *Trl, float th, int forward, int backward, int check_orientation, int *assign,
                                    int *n_matches, int *best_dist, int *best_idx, int *best_dist_r, int *best_idx_r);

/* ------------------------------------------------------------------------------------------------
 * Frustum test + scale prediction for the local map points: Frame::isInFrustum /
 * isInFrustumChecks (src/Frame.cc:536-610, 1308-1382) with MapPoint::PredictScale (src/MapPoint.cc:531-546),
 * the host loop in front of SearchByProjection (src/Tracking.cc:3503-3522).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ft_frame_pose {
    float Rcw[9]; /* mRcw, row-major */
    float tcw[3]; /* mtcw */
    float Ow[3];  /* mOw */
    float tlr[3]; /* mTlr.translation(), two-camera frames only (Frame.cc:1319) */
} ft_frame_pose;

typedef struct ft_map_points {
    int M;
    const uint8_t *skip;       /* NULL or: mnLastFrameSeen == frame id || isBad() (Tracking.cc:3507-3510) */
    const float *world_pos;    /* M x 3, GetWorldPos() */
    const float *normal;       /* M x 3, GetNormal() */
    const float *max_distance; /* mfMaxDistance (GetMaxDistanceInvariance() is 1.2f times this) */
    const float *min_distance; /* mfMinDistance (GetMinDistanceInvariance() is 0.8f times this) */
    const uint8_t *descriptors; /* M x 32, GetDescriptor()  - read by the searches only */
    const int *observations;    /* Observations()           - read by the searches only */
} ft_map_points;

/* The MapPoint tracking fields isInFrustum writes (arrays of M, each may be NULL): mbTrackInView(R),
 * mnTrackScaleLevel(R), mTrackViewCos(R), mTrackProjX/Y, mTrackProjXR/YR, mTrackDepth(R).  Fields the
 * reference leaves untouched for a point read level -1, view_cos 0, proj -1, depth 0. */
typedef struct ft_frustum_result {
    uint8_t *in_view, *in_view_r;
    int *level, *level_r;
    float *view_cos, *view_cos_r;
    float *proj_x, *proj_y, *proj_xr, *proj_yr;
    float *depth, *depth_r;
} ft_frustum_result;

/* F supplies the frame constants only (Nleft, bounds, camera, Trl, mbf, nlevels); its keypoint arrays are not
 * read.  log_scale_factor = Frame::mfLogScaleFactor.  *n_to_match = number of points in view of a camera. */
FT_API int ft_is_in_frustum(ft_context *ctx, const ft_frame_view *F, const ft_frame_pose *pose, const ft_map_points *P,
                            float viewing_cos_limit, float log_scale_factor, const ft_frustum_result *out,
                            int *n_to_match);

/* ------------------------------------------------------------------------------------------------
 * Device-resident frame for the projection searches.  The reference re-marshals the whole
 * Frame into a CudaFrame on every kernel call (src/Kernels/CudaWrappers/CudaFrame.cu:77-181).  Here the frame
 * is uploaded once (or bound to the buffers a stereo front end already holds in HBM: no copy at all) and is
 * then used by isInFrustum, SearchByProjection(last frame) and SearchByProjection(local map); the occupancy
 * of mvpMapPoints (holder_obs) carries over from one search to the next as in Tracking::TrackWithMotionModel
 * followed by Tracking::SearchLocalPoints, and the frustum outputs feed the local-map search without
 * leaving the device.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ft_tracked_frame ft_tracked_frame;
FT_API int ft_tracked_frame_create(ft_context *ctx, int max_keypoints, int max_points, ft_tracked_frame **out);
FT_API int ft_tracked_frame_destroy(ft_tracked_frame *tf);
/* copies keys / descriptors / uright / match tables / holder_obs of F to the device and builds its grid.  The arrays of F are
 * read before the call returns (packed into pinned staging of the frame); the device copies and the grid are ordered in front of
 * the searches that follow on the context's stream, the call itself does not wait for them. */
FT_API int ft_tracked_frame_upload(ft_tracked_frame *tf, const ft_frame_view *F);
/* pair `slot` of the batch the front end processed last (rectified stereo, Nleft == -1): keypoints, descriptors
 * and mvuRight are used where ft_stereo_frontend_* left them in HBM.  meta supplies the frame constants and
 * meta->keys (host copy, for the rotation histogram); holder_obs starts at meta->holder_obs or all -1.
 * The binding is valid until the front end processes another batch. */
FT_API int ft_tracked_frame_bind_stereo(ft_tracked_frame *tf, ft_stereo_frontend *fe, int slot, const ft_frame_view *meta);
/* ft_search_last_frame on the resident frame; updates the resident holder_obs */
FT_API int ft_tracked_frame_search_last_frame(ft_tracked_frame *tf, const ft_last_points *L, const float *Tcw, float th,
                                              int forward, int backward, int check_orientation, int *assign,
                                              int *n_matches);
/* the Sophus form of the poses (see ft_search_last_frame_se3) on the resident frame */
FT_API int ft_tracked_frame_search_last_frame_se3(ft_tracked_frame *tf, const ft_last_points *L, const ft_se3 *Tcw,
                                                  const ft_se3 *Trl, float th, int forward, int backward,
                                                  int check_orientation, int *assign, int *n_matches);
/* Tracking::SearchLocalPoints on the resident frame: isInFrustum (viewing_cos_limit, log_scale_factor) for all
 * points, then SearchByProjection(F, points, th, far_points, th_far_points) on the device-resident results.
 * frustum (may be NULL) receives the isInFrustum fields; assign / n_matches as in ft_search_local_points. */
FT_API int ft_tracked_frame_track_local_map(ft_tracked_frame *tf, const ft_frame_pose *pose, const ft_map_points *P,
                                            float viewing_cos_limit, float log_scale_factor, float th, float nn_ratio,
                                            int far_points, float th_far_points, const ft_frustum_result *frustum,
                                            int *n_to_match, int *assign, int *n_matches);
/* current holder_obs (size N of the resident frame) */
FT_API int ft_tracked_frame_holder_obs(ft_tracked_frame *tf, int *holder_obs);

/* ------------------------------------------------------------------------------------------------
 * B device-resident frames per call ("Batch API (B frames per launch)").  The reference's tracking thread
 * searches one frame at a time (Tracking::TrackWithMotionModel src/Tracking.cc:2911-2989, Tracking::SearchLocalPoints
 * :3472-3555; one kernel launch per call: src/Kernels/SearchLocalPointsKernel.cu:351-435, PoseEstimationKernel.cu:350-371),
 * which leaves a 256-CU device idle by construction.  A batch holds n_frames INDEPENDENT frames - the camera streams of one
 * time step, or any frames whose inputs the caller holds - and runs every stage as ONE launch over all of them: the grid
 * build, isInFrustum, the window scans of a search, and the in-call claiming of all frames (option search_cache = 2, batches of
 * 24 frames and more: one launch, a workgroup per frame walks the frame's points in index order; smaller batches, search_cache
 * <= 1, or a frame with a window of more than 511 candidates: claim passes, one launch per pass for all frames, per-frame
 * convergence).  Per frame the results are those of the
 * ft_tracked_frame_* call on that frame, bit for bit; the holder_obs of every frame carries over from one search to the next.
 * Arrays indexed by frame: frames[], L[], Tcw (12 floats per frame), forward / backward (NULL = all 0), poses[], P[],
 * frustum[] (NULL = not wanted), n_to_match[], assign[] (assign[f] has frames[f].N entries), n_matches[].
 * A batch object has a stream and a lock of its own: calls on one batch are serialised, two batches of one context used from
 * two host threads are two batches in flight.
 * The host's share of a search: the writes of a search (mvpMapPoints[kp] = pMP in point order, the rotation histogram and
 * ComputeThreeMaxima of SearchByProjection(CurrentFrame, LastFrame), src/ORBmatcher.cc:1880-1896, 1966-1987, 2210-2251) are
 * replayed ON THE DEVICE, the frames' holder_obs live in HBM, and the assignments come back through pinned memory - the host
 * builds the job records, enqueues, and copies the assignments into the caller's arrays.  Point arrays (ft_last_points /
 * ft_map_points) that ALL lie in pinned host memory (ft_host_malloc, hipHostMalloc, hipHostRegister) are read in place by the
 * device - no host copy; pageable ones are packed into the batch's pinned staging by the context's host threads first.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ft_tracked_batch ft_tracked_batch;
FT_API int ft_tracked_batch_create(ft_context *ctx, int max_frames, int max_keypoints, int max_points, ft_tracked_batch **out);
FT_API int ft_tracked_batch_destroy(ft_tracked_batch *tb);
/* ft_tracked_frame_upload for n_frames frames: one packed copy, all grids by one launch */
FT_API int ft_tracked_batch_upload(ft_tracked_batch *tb, int n_frames, const ft_frame_view *frames);
/* The frames of the batch straight from what two extractors left in HBM - no keypoint or descriptor crosses PCIe again.
 * Slot slot0 + f of exL / exR holds the left / right image of two-camera frame f (their last ft_extract_batch calls; lapping areas
 * [lap_l0, lap_l1] / [lap_r0, lap_r1] as passed there).  On the device: keypoints and descriptors are put into the reference's
 * order (lapping-area keypoints filled from the back, src/ORBextractor.cc:1466-1487); Frame::ComputeStereoFishEyeMatches
 * (src/Frame.cc:1231-1271) for every frame - 2-NN + ratio 0.7 between the lapping subsets (the seam of
 * launchFisheyeStereoMatchKernel, include/Kernels/KernelController.h:38) and, when `rig` is given, KannalaBrandt8::
 * TriangulateMatches on every surviving pair (depth > 0.0001 keeps it; level_sigma2 = mvLevelSigma2, nlevels floats) - fills
 * mvLeftToRightMatch / mvRightToLeftMatch; the grids are built.  rig == NULL: the matching alone (every ratio-test survivor stays).
 * meta[f]: the frame constants, N / Nleft (= the counts ft_extract_batch returned), holder_obs or NULL (all -1); its keys /
 * keys_right / descriptors / match tables are not read (the angles of the rotation histogram are read on the device).
 * exL / exR must not be used by another thread during the call; their next batch is ordered behind this call's reads of their slots.
 * Outputs, each an array of n_frames pointers or NULL: left_to_right[f] / right_to_left[f] (the match tables; both or neither),
 * depth[f] / p3d[f] (mvDepth [Nleft], mvStereo3Dpoints [3 Nleft]; with a rig only), n_stereo[f] (the frame's nMatches).  Without
 * outputs the call does not wait for the device. */
FT_API int ft_tracked_batch_bind_fisheye(ft_tracked_batch *tb, ft_extractor *exL, ft_extractor *exR, int slot0, int n_frames,
                                         int lap_l0, int lap_l1, int lap_r0, int lap_r1, const ft_frame_view *meta,
                                         const ft_fisheye_rig *rig, const float *level_sigma2, int *const *left_to_right,
                                         int *const *right_to_left, float *const *depth, float *const *p3d, int *n_stereo);
/* the same with a first slot per camera: exL == exR is allowed - ONE extractor whose last batch holds the left images in slots
 * slot0 .. and the right ones in slots slot0_right .. (disjoint ranges), e.g. a two-camera frame extracted as a batch of two */
FT_API int ft_tracked_batch_bind_fisheye_slots(ft_tracked_batch *tb, ft_extractor *exL, ft_extractor *exR, int slot0, int slot0_right,
                                               int n_frames, int lap_l0, int lap_l1, int lap_r0, int lap_r1,
                                               const ft_frame_view *meta, const ft_fisheye_rig *rig, const float *level_sigma2,
                                               int *const *left_to_right, int *const *right_to_left, float *const *depth,
                                               float *const *p3d, int *n_stereo);
/* ft_tracked_frame_search_last_frame for every frame of the batch (n_frames = the number uploaded) */
FT_API int ft_tracked_batch_search_last_frame(ft_tracked_batch *tb, int n_frames, const ft_last_points *L, const float *Tcw,
                                              float th, const int *forward, const int *backward, int check_orientation,
                                              int *const *assign, int *n_matches);
/* the Sophus form of the poses (ft_search_last_frame_se3); Trl[] may be NULL when no frame has two cameras */
FT_API int ft_tracked_batch_search_last_frame_se3(ft_tracked_batch *tb, int n_frames, const ft_last_points *L, const ft_se3 *Tcw,
                                                  const ft_se3 *Trl, float th, const int *forward, const int *backward,
                                                  int check_orientation, int *const *assign, int *n_matches);
/* ft_tracked_frame_track_local_map for every frame of the batch */
FT_API int ft_tracked_batch_track_local_map(ft_tracked_batch *tb, int n_frames, const ft_frame_pose *poses,
                                            const ft_map_points *P, float viewing_cos_limit, float log_scale_factor, float th,
                                            float nn_ratio, int far_points, float th_far_points,
                                            const ft_frustum_result *frustum, int *n_to_match, int *const *assign,
                                            int *n_matches);
/* The two halves of the three searches above, like ft_stereo_frontend_submit / _wait: submit_* enqueues the search (the claim
 * iteration up to the first point where the host has to look at the device's flag words - the whole search when the one-launch
 * resolution resolves every frame, the usual case) and returns; ft_tracked_batch_wait waits for the device, runs what is left
 * (claim passes for frames the resolution gave up on, small batches), and fills assign / n_matches / n_to_match / frustum.  One
 * search per batch may be in flight: a second submit (or upload / bind_fisheye) before the wait is FT_ERR_INVALID.  Between
 * submit and wait the OUTPUT arrays must stay untouched and point arrays in pinned memory (read in place, see above) must
 * stay valid and UNCHANGED; everything else of the arguments is copied before submit returns.  An input error only the device
 * sees when it reads the arrays in place (a last-frame octave outside the frame's levels) is reported by the wait.
 * ft_tracked_batch_wait without a submitted search returns FT_OK. */
FT_API int ft_tracked_batch_submit_search_last_frame(ft_tracked_batch *tb, int n_frames, const ft_last_points *L, const float *Tcw,
                                                     float th, const int *forward, const int *backward, int check_orientation,
                                                     int *const *assign, int *n_matches);
FT_API int ft_tracked_batch_submit_search_last_frame_se3(ft_tracked_batch *tb, int n_frames, const ft_last_points *L,
                                                         const ft_se3 *Tcw, const ft_se3 *Trl, float th, const int *forward,
                                                         const int *backward, int check_orientation, int *const *assign,
                                                         int *n_matches);
FT_API int ft_tracked_batch_submit_track_local_map(ft_tracked_batch *tb, int n_frames, const ft_frame_pose *poses,
                                                   const ft_map_points *P, float viewing_cos_limit, float log_scale_factor,
                                                   float th, float nn_ratio, int far_points, float th_far_points,
                                                   const ft_frustum_result *frustum, int *n_to_match, int *const *assign,
                                                   int *n_matches);
FT_API int ft_tracked_batch_wait(ft_tracked_batch *tb);
/* current holder_obs of frame `frame` (its N entries): copied down from HBM, where the searches keep it */
FT_API int ft_tracked_batch_holder_obs(ft_tracked_batch *tb, int frame, int *holder_obs);

/* ORBmatcher::DescriptorDistance for n pairs on the device (src/ORBmatcher.cc:2256-2272,
 * device copy src/Kernels/CudaUtils.cu:42-56).  a, b: n x 32 host bytes; dist: n ints. */
FT_API int ft_descriptor_distance(ft_context *ctx, const uint8_t *a, const uint8_t *b, int n, int *dist);

/* Host-libm self test.  The rBRIEF rotation (src/ORBextractor.cc:73-74: std::cos(float), std::sin(float)) and
 * MapPoint::PredictScale (src/MapPoint.cc:539: std::log(float)) are evaluated by the reference with the HOST libm;
 * the kernels reproduce glibc's cosf / sinf / logf bit for bit (fasttrack_amd/csrc/libm_f32.h).  This sweep evaluates
 * func (0 cosf, 1 sinf, 2 logf, 3 atanf, 4 atan2f, 5 tanf) on the device for the float bit patterns first_bits, first_bits + stride,
 * ... <= last_bits and compares with the same function of the calling process's libm.  mismatches == 0 over [0, 0x40c90fdb]
 * (cos, sin) and [1, 0x461c4000] (log, up to 1e4) means the device and this host's libm agree on every argument the path can
 * produce; atanf takes any range up to [0, 0xffffffff]; atan2f(y, x) sweeps y over the range and pairs each y with one x
 * derived from its bits (either sign, 2^-9 <= |x| < 2^7): the KannalaBrandt8 projection of two-camera frames; tanf (its
 * unprojection) is reproduced for |x| < 120, i.e. bits [0, 0x42efffff] and [0x80000000, 0xc2efffff]. */
FT_API int ft_selftest_libm(ft_context *ctx, int func, uint32_t first_bits, uint32_t last_bits, uint32_t stride,
                            unsigned long long *checked, unsigned long long *mismatches, uint32_t *first_bad);

/* ----------------------------------------------------------------------------------------------
 * Frame::ComputeBoW (src/Frame.cc:762-769): DBoW2's
 * TemplatedVocabulary<FORB::TDescriptor, FORB>::transform(features, BowVector&, FeatureVector&, levelsup)
 * (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1194) with the tree walk of every descriptor
 * (:1208-1253, FORB::distance Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101) on the device.
 *
 * A vocabulary is the k-ary tree of ORBvoc.txt: node 0 is the root, node i > 0 has parent[i] < i ... (any order of
 * the file is accepted), the children of a node are visited in ascending node id (the order loadFromTextFile
 * appends them, :1369-1420), leaves are the words, numbered in ascending node id.  scoring / weighting use the
 * reference's enum values (BowVector.h:39-56: weighting 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY; scoring 0 L1_NORM,
 * 1 L2_NORM, 2 CHI_SQUARE, 3 KL, 4 BHATTACHARYYA, 5 DOT_PRODUCT).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ft_vocabulary ft_vocabulary;
/* n_nodes includes the root (entry 0 of every array is ignored except is_leaf); descriptors: n_nodes x 32 bytes */
FT_API int ft_vocabulary_create(ft_context *ctx, int k, int L, int scoring, int weighting, int n_nodes,
                                const int *parent, const uint8_t *is_leaf, const uint8_t *descriptors,
                                const double *weights, ft_vocabulary **out);
/* the text format ORB-SLAM3 ships (TemplatedVocabulary::loadFromTextFile, :1338-1423): "k L scoring weighting", then
 * one line per node: parent is_leaf d0 .. d31 weight */
FT_API int ft_vocabulary_load_text(ft_context *ctx, const char *path, ft_vocabulary **out);
FT_API int ft_vocabulary_destroy(ft_vocabulary *voc);
FT_API int ft_vocabulary_info(const ft_vocabulary *voc, int *k, int *L, int *n_nodes, int *n_words);
/* transform of the n descriptors of one frame (host n x 32 bytes, or a device pointer with on_device != 0, e.g. the
 * descriptors a stereo front end left in HBM).
 * Per feature (each may be NULL): word_ids, node_ids (the ancestor at level L - levelsup, 0 = root if that level is
 * <= 0; the leaf itself if the walk ends above that level), weights (the word's weight; 0 = stopped word).
 * BowVector: bow_ids ascending with bow_values (after the weighting and the normalisation the scoring asks for),
 * n_bow entries (bow_capacity >= n is always enough).
 * FeatureVector in CSR form: fv_nodes ascending, the feature indices of node j (ascending, as push_back leaves them)
 * are fv_features[fv_offsets[j] .. fv_offsets[j + 1]), n_fv nodes (fv_nodes / fv_offsets hold fv_capacity and
 * fv_capacity + 1 entries, fv_features n). */
FT_API int ft_bow_transform(ft_vocabulary *voc, const uint8_t *descriptors, int n, int on_device, int levelsup,
                            unsigned *word_ids, unsigned *node_ids, double *weights, unsigned *bow_ids,
                            double *bow_values, int bow_capacity, int *n_bow, unsigned *fv_nodes, int *fv_offsets,
                            unsigned *fv_features, int fv_capacity, int *n_fv);

/* ----------------------------------------------------------------------------------------------
 * ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint*> &vpMapPointMatches)
 * (src/ORBmatcher.cc:322-524; Tracking::TrackReferenceKeyFrame src/Tracking.cc:2732-2744 and Relocalization): the step
 * that consumes the FeatureVectors of ft_bow_transform.  A side is its FeatureVector in that CSR form, its descriptors and,
 * for the rotation-consistency filter (mbCheckOrientation), its keypoint angles.  kf_has_point[i] != 0 <=> the keyframe's
 * i-th map point exists and is not bad (:345-351).  frame_nleft = Frame::Nleft (-1 = one camera; otherwise descriptors and
 * angles of the right camera follow the left ones, :383-421 incl. the disabled ratio test of the right camera, :453).
 * matches[i] (i < frame->n) = index of the keyframe feature whose map point vpMapPointMatches[i] receives, or -1;
 * *n_matches = the function's return value.  TH_LOW = 50, HISTO_LENGTH = 30 as in the reference (:42-43). */
typedef struct ft_bow_side {
    int n;                       /* features (descriptors, angles) */
    int n_nodes;                 /* FeatureVector entries */
    const unsigned *fv_nodes;    /* [n_nodes] ascending */
    const int *fv_offsets;       /* [n_nodes + 1] */
    const unsigned *fv_features; /* [fv_offsets[n_nodes]] */
    const uint8_t *descriptors;  /* n x 32 bytes, host */
    const float *angles;         /* [n] cv::KeyPoint::angle; may be NULL without check_orientation */
} ft_bow_side;
FT_API int ft_search_by_bow(ft_context *ctx, const ft_bow_side *kf, const uint8_t *kf_has_point, const ft_bow_side *frame,
                            int frame_nleft, float nn_ratio, int check_orientation, int *matches, int *n_matches);

#ifdef __cplusplus
}
#endif
#endif /* FASTTRACK_AMD_H */
