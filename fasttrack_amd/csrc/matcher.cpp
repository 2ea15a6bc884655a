// Host side of the stereo matchers: replaces KernelController::launchStereoMatchKernel /
// launchFisheyeStereoMatchKernel (reference include/Kernels/KernelController.h:31-38) and offers the
// fused extract+match front end used for throughput.  All compute is in kernels_match.hip.
#include <algorithm>
#include <cstring>

#include "ft_host.h"

#define FT_GRAPH_MAX_BATCH 8  // latency mode: batches up to this size are captured as graphs / run paired

#define FT_REQUIRE(cond, msg)               \
    do {                                    \
        if (!(cond)) {                      \
            ft_set_error(std::string(msg)); \
            return FT_ERR_INVALID;          \
        }                                   \
    } while (0)

static bool sameGeometry(const ft_extractor *a, const ft_extractor *b) {
    return a->width == b->width && a->height == b->height && a->nlevels == b->nlevels &&
           a->scaleFactor == b->scaleFactor && a->ctx == b->ctx;
}

extern "C" {

int ft_stereo_match(ft_extractor *exL, ft_extractor *exR, int slot, const ft_keypoint *keysL, int nL,
                    const ft_keypoint *keysR, int nR, const uint8_t *descL, const uint8_t *descR, float mbf,
                    float mb, int apply_median_cut, float *uright, float *depth, int *sad, int *n_matches) {
    FT_REQUIRE(exL && exR && uright && depth, "ft_stereo_match: null argument");
    FT_REQUIRE(sameGeometry(exL, exR), "ft_stereo_match: extractors differ in geometry or context");
    FT_REQUIRE(slot >= 0 && slot < exL->lastBatch && slot < exR->lastBatch, "ft_stereo_match: slot holds no pyramid");
    FT_REQUIRE(nL >= 0 && nR >= 0 && nL < 65536 && nR < 65536, "ft_stereo_match: keypoint count out of range");
    FT_REQUIRE(nL == 0 || (keysL && descL), "ft_stereo_match: null left arrays");
    FT_REQUIRE(nR == 0 || (keysR && descR), "ft_stereo_match: null right arrays");
    int rc = ft_set_device(exL->ctx);
    if (rc != FT_OK) return rc;
    if (nL == 0) {
        if (n_matches) *n_matches = 0;
        return FT_OK;
    }
    const int cap = std::max(std::max(nL, nR), 1);
    const FtGeom &g = exL->geom;
    static_assert(sizeof(ft_keypoint) == 28, "ft_keypoint must match cv::KeyPoint");
    // device scratch for the host arrays: grows to the largest request
    const int need = std::max(cap, g.maxKp);
    if (!exL->d_stKeys || need > exL->stCap) {
        hipFree(exL->d_stKeys);
        hipFree(exL->d_stDesc);
        hipFree(exL->d_stOut);
        hipFree(exL->d_stInt);
        hipFree(exL->d_stSorted);
        exL->d_stSorted = nullptr;
        exL->d_stKeys = nullptr;
        exL->d_stDesc = nullptr;
        exL->d_stOut = nullptr;
        exL->d_stInt = nullptr;
        FT_HIP(hipMalloc((void **)&exL->d_stKeys, sizeof(ft_keypoint) * 2 * need));
        FT_HIP(hipMalloc((void **)&exL->d_stDesc, (size_t)64 * need));
        FT_HIP(hipMalloc((void **)&exL->d_stOut, sizeof(float) * 2 * need));
        FT_HIP(hipMalloc((void **)&exL->d_stInt, sizeof(int) * (2 * (size_t)need + 8 + exL->height + 2)));
        FT_HIP(hipMalloc((void **)&exL->d_stSorted, sizeof(FtSortedR) * (size_t)need));
        exL->stCap = need;
    }
    const int C = exL->stCap;
    hipStream_t st = exL->stream;
    FT_HIP(hipStreamSynchronize(exR->stream));  // right pyramid must be complete
    FT_HIP(hipStreamSynchronize(exR->streamB));
    FT_HIP(hipStreamSynchronize(exL->streamB));
    FT_HIP(hipMemcpyAsync(exL->d_stKeys, keysL, sizeof(ft_keypoint) * nL, hipMemcpyHostToDevice, st));
    FT_HIP(hipMemcpyAsync(exL->d_stDesc, descL, (size_t)32 * nL, hipMemcpyHostToDevice, st));
    if (nR > 0) {
        FT_HIP(hipMemcpyAsync(exL->d_stKeys + C, keysR, sizeof(ft_keypoint) * nR, hipMemcpyHostToDevice, st));
        FT_HIP(hipMemcpyAsync(exL->d_stDesc + (size_t)32 * C, descR, (size_t)32 * nR, hipMemcpyHostToDevice, st));
    }
    int hdr[4] = {nL, nR, 0, 0};
    int *d_hdr = exL->d_stInt + 2 * C;
    FT_HIP(hipMemcpyAsync(d_hdr, hdr, sizeof hdr, hipMemcpyHostToDevice, st));
    FtStereoArgs a;
    a.keysL = exL->d_stKeys;
    a.keysR = exL->d_stKeys + C;
    a.descL = exL->d_stDesc;
    a.descR = exL->d_stDesc + (size_t)32 * C;
    a.nL = d_hdr;
    a.nR = d_hdr + 1;
    a.capacity = C;  // single pair: stride is irrelevant but bounds the grid
    a.mbf = mbf;
    a.mb = mb;
    a.uright = exL->d_stOut;
    a.depth = exL->d_stOut + C;
    a.sad = exL->d_stInt;
    a.hamIdx = exL->d_stInt + C;
    a.nMatches = d_hdr + 2;
    a.applyMedianCut = apply_median_cut;
    a.sorted = exL->d_stSorted;
    a.rowStart = exL->d_stInt + 2 * C + 4;
    a.rowStride = exL->height + 2;
    a.alignedLoads = exL->l0Aligned && exR->l0Aligned;
    rc = ft_launch_stereo_rowsort(st, g, 1, a);
    if (rc != FT_OK) return rc;
    rc = ft_launch_stereo_match(st, g, 1, exL->d_l0 + slot, exR->d_l0 + slot, exL->l0pitch, exR->l0pitch,
                                exL->d_pyr + (size_t)slot * g.pyrPerSlot, exR->d_pyr + (size_t)slot * g.pyrPerSlot, a);
    if (rc != FT_OK) return rc;
    rc = ft_launch_stereo_median(st, 1, a);
    if (rc != FT_OK) return rc;
    FT_HIP(hipMemcpyAsync(uright, a.uright, sizeof(float) * nL, hipMemcpyDeviceToHost, st));
    FT_HIP(hipMemcpyAsync(depth, a.depth, sizeof(float) * nL, hipMemcpyDeviceToHost, st));
    if (sad) FT_HIP(hipMemcpyAsync(sad, a.sad, sizeof(int) * nL, hipMemcpyDeviceToHost, st));
    int nm = 0;
    FT_HIP(hipMemcpyAsync(&nm, a.nMatches, sizeof(int), hipMemcpyDeviceToHost, st));
    FT_HIP(hipStreamSynchronize(st));
    if (n_matches) *n_matches = nm;
    return FT_OK;
}

// ---- fused front end ---------------------------------------------------------------------------
int ft_stereo_frontend_create(ft_context *ctx, int nfeatures, float scale_factor, int nlevels, int ini_th_fast,
                              int min_th_fast, int image_width, int image_height, int max_batch, float mbf, float mb,
                              ft_stereo_frontend **out) {
    FT_REQUIRE(ctx && out, "ft_stereo_frontend_create: null argument");
    *out = nullptr;
    ft_stereo_frontend *fe = new ft_stereo_frontend();
    fe->ctx = ctx;
    fe->mbf = mbf;
    fe->mb = mb;
    fe->maxBatch = max_batch;
    const bool pairedOn = ctx->tuning.paired != 0;
    fe->pairedCapable = pairedOn && max_batch >= 1 && max_batch <= FT_GRAPH_MAX_BATCH;
    int rc = ft_extractor_create(ctx, nfeatures, scale_factor, nlevels, ini_th_fast, min_th_fast, image_width,
                                 image_height, fe->pairedCapable ? 2 * max_batch : max_batch, &fe->exL);
    if (rc == FT_OK)
        rc = ft_extractor_create(ctx, nfeatures, scale_factor, nlevels, ini_th_fast, min_th_fast, image_width,
                                 image_height, max_batch, &fe->exR);
    if (rc != FT_OK) {
        ft_stereo_frontend_destroy(fe);
        return rc;
    }
    fe->capacity = fe->exL->geom.maxKp;
    const size_t n = (size_t)max_batch * fe->capacity;
    hipError_t e = hipMalloc((void **)&fe->d_uright, sizeof(float) * n);
    if (e == hipSuccess) e = hipMalloc((void **)&fe->d_depth, sizeof(float) * n);
    if (e == hipSuccess) e = hipMalloc((void **)&fe->d_sad, sizeof(int) * n);
    if (e == hipSuccess) e = hipMalloc((void **)&fe->d_nMatches, sizeof(int) * max_batch);
    if (e == hipSuccess) e = hipMalloc((void **)&fe->d_sorted, sizeof(FtSortedR) * n);
    if (e == hipSuccess) e = hipMalloc((void **)&fe->d_rowStart, sizeof(int) * (size_t)max_batch * (image_height + 2));
    if (e == hipSuccess) e = hipHostMalloc((void **)&fe->h_uright, sizeof(float) * n, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&fe->h_depth, sizeof(float) * n, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&fe->h_nMatches, sizeof(int) * max_batch, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&fe->evR, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&fe->evFork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&fe->evJoin, hipEventDisableTiming);
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&fe->evDone[i], hipEventDisableTiming);
    if (e != hipSuccess) {
        ft_stereo_frontend_destroy(fe);
        return ft_hip_fail(e, "stereo frontend allocation", __FILE__, __LINE__);
    }
    *out = fe;
    return FT_OK;
}

int ft_stereo_frontend_destroy(ft_stereo_frontend *fe) {
    if (!fe) return FT_OK;
    hipSetDevice(fe->ctx->device);
    ft_extractor_destroy(fe->exL);
    ft_extractor_destroy(fe->exR);
    hipFree(fe->d_uright);
    hipFree(fe->d_depth);
    hipFree(fe->d_sad);
    hipFree(fe->d_nMatches);
    hipFree(fe->d_sorted);
    hipFree(fe->d_rowStart);
    hipHostFree(fe->h_uright);
    hipHostFree(fe->h_depth);
    hipHostFree(fe->h_nMatches);
    if (fe->evR) hipEventDestroy(fe->evR);
    if (fe->evFork) hipEventDestroy(fe->evFork);
    if (fe->evJoin) hipEventDestroy(fe->evJoin);
    for (int i = 0; i < 2; i++)
        if (fe->evDone[i]) hipEventDestroy(fe->evDone[i]);
    if (fe->graphExec) hipGraphExecDestroy(fe->graphExec);
    delete fe;
    return FT_OK;
}

ft_extractor *ft_stereo_frontend_left(ft_stereo_frontend *fe) { return fe ? fe->exL : nullptr; }
ft_extractor *ft_stereo_frontend_right(ft_stereo_frontend *fe) { return fe ? fe->exR : nullptr; }

// result arrays in pinned host memory (ft_host_malloc) are filled by the D2H copies directly
static bool isPinnedHost(const void *p) { return ft_is_pinned_host(p); }


// Matching and result copies of the slots [b0, b0 + nb) once their descriptors are enqueued on the stage-B streams of the
// two extractors: row sort, stereo match, median cut on the left stage-B stream, then (unless one kernel delivers the whole
// batch at the end) the D2H copies of the rows.  Also the tail of the per-image host-octree repair.
static int frontendMatchAndDeliver(ft_stereo_frontend *fe, int s, int b0, int nb, bool dev, bool deliver, bool direct,
                                   ft_keypoint *keysL, uint8_t *descL, ft_keypoint *keysR, uint8_t *descR, int capacity,
                                   float *uright, float *depth) {
    ft_extractor *L = fe->exL, *R = fe->exR;
    const FtGeom &g = L->geom;
    const bool tm = fe->ctx->kernelTiming;
    hipStream_t st = L->streamB;
    int rc;
    // pageable result arrays go through the library's pinned staging buffers and a host memcpy in _wait
    // (with the device octree the per-image counts are unknown until the end: whole rows are copied, which the
    // caller's arrays must be able to hold)
    const size_t kp = sizeof(ft_keypoint);
    auto d2h = [&](void *user, void *staging, const void *dev, size_t elem, int b0, int nb, int maxN) -> hipError_t {
        // rows of `elem`-byte records: device/staging stride maxKp, user stride capacity
        const size_t o = (size_t)b0 * g.maxKp * elem;
        if (direct) {
            if (!user) return hipSuccess;
            return hipMemcpy2DAsync((uint8_t *)user + (size_t)b0 * capacity * elem, elem * capacity, (const uint8_t *)dev + o,
                                    elem * g.maxKp, elem * maxN, nb, hipMemcpyDeviceToHost, st);
        }
        return hipMemcpy2DAsync((uint8_t *)staging + o, elem * g.maxKp, (const uint8_t *)dev + o, elem * g.maxKp, elem * maxN,
                                nb, hipMemcpyDeviceToHost, st);
    };
    FT_HIP(hipEventRecord(R->evB[s], R->streamB));
    FT_HIP(hipStreamWaitEvent(st, R->evB[s], 0));
    // keypoints stay on the device between extraction and matching: pinhole stereo passes the lapping
    // area (0,0) (src/Frame.cc:127), no keypoint has x == 0, so mono order == extraction order.
    const size_t o = (size_t)b0 * g.maxKp;
    FtStereoArgs a;
    a.keysL = L->d_keys + o;
    a.keysR = R->d_keys + o;
    a.descL = L->d_desc + o * 32;
    a.descR = R->d_desc + o * 32;
    a.nL = L->d_nSel + b0;
    a.nR = R->d_nSel + b0;
    a.capacity = g.maxKp;
    a.mbf = fe->mbf;
    a.mb = fe->mb;
    a.uright = fe->d_uright + o;
    a.depth = fe->d_depth + o;
    a.sad = fe->d_sad + o;
    a.hamIdx = nullptr;
    a.nMatches = fe->d_nMatches + b0;
    a.applyMedianCut = 1;
    a.rowStride = L->height + 2;
    a.rowStart = fe->d_rowStart + (size_t)b0 * a.rowStride;
    a.sorted = fe->d_sorted + o;
    a.alignedLoads = L->l0Aligned && R->l0Aligned;
    L->evt.begin(tm, "kernel.stereo_rowsort", st);
    rc = ft_launch_stereo_rowsort(st, g, nb, a);
    L->evt.end(tm, st);
    if (rc != FT_OK) return rc;
    L->evt.begin(tm, "kernel.stereo_match", st);
    rc = ft_launch_stereo_match(st, g, nb, L->d_l0 + b0, R->d_l0 + b0, L->l0pitch, R->l0pitch,
                                L->d_pyr + (size_t)b0 * g.pyrPerSlot, R->d_pyr + (size_t)b0 * g.pyrPerSlot, a);
    L->evt.end(tm, st);
    if (rc != FT_OK) return rc;
    L->evt.begin(tm, "kernel.stereo_median", st);
    rc = ft_launch_stereo_median(st, nb, a);
    L->evt.end(tm, st);
    if (rc != FT_OK) return rc;
    int maxNL = 0, maxNR = 0;
    for (int b = b0; b < b0 + nb; b++) {
        maxNL = std::max(maxNL, dev ? g.maxKp : L->h_nSel[b]);
        maxNR = std::max(maxNR, dev ? g.maxKp : R->h_nSel[b]);
    }
    if (!dev && (maxNL > capacity || maxNR > capacity)) {
        ft_set_error("stereo front end: output capacity too small (use ft_extractor_max_keypoints)");
        return FT_ERR_CAPACITY;
    }
    if (deliver) return FT_OK;  // one kernel at the end delivers everything
    if (maxNL > 0) {
        FT_HIP(d2h(keysL, L->h_keys, L->d_keys, kp, b0, nb, maxNL));
        FT_HIP(d2h(descL, L->h_desc, L->d_desc, 32, b0, nb, maxNL));
        FT_HIP(d2h(uright, fe->h_uright, fe->d_uright, 4, b0, nb, maxNL));
        FT_HIP(d2h(depth, fe->h_depth, fe->d_depth, 4, b0, nb, maxNL));
    }
    if (maxNR > 0) {
        FT_HIP(d2h(keysR, R->h_keys, R->d_keys, kp, b0, nb, maxNR));
        FT_HIP(d2h(descR, R->h_desc, R->d_desc, 32, b0, nb, maxNR));
    }
    FT_HIP(hipMemcpyAsync(fe->h_nMatches + b0, fe->d_nMatches + b0, sizeof(int) * nb, hipMemcpyDeviceToHost, st));
    return FT_OK;
}

// enqueues one batch on the front end's streams.  capture != 0: L->stream is being captured into a graph - the other
// streams are forked from it first and joined back at the end, and nothing here may synchronise.
static int frontendEnqueue(ft_stereo_frontend *fe, const uint8_t *const *imagesL, const uint8_t *const *imagesR, int batch,
                           int on_device, int width, int height, int stride, ft_keypoint *keysL, uint8_t *descL, int *nL,
                           ft_keypoint *keysR, uint8_t *descR, int *nR, int capacity, float *uright, float *depth,
                           int *n_matches, bool direct, int capture, bool paired) {
    ft_extractor *L = fe->exL, *R = fe->exR;
    const FtGeom &g = L->geom;
    FtTimer tAll;
    fe->lastPaired = paired;
    if (paired) {
        // both cameras through the left extractor: slots [0, B) left, [B, 2B) right; one launch per kernel
        const int B = batch;
        std::vector<const uint8_t *> imgs(imagesL, imagesL + B);
        imgs.insert(imgs.end(), imagesR, imagesR + B);
        int rc = ft_extract_prepare(L, imgs.data(), 2 * B, on_device, width, height, stride);
        if (rc != FT_OK) return rc;
        fe->ctx->addStat("stereo.device_octree_batches", 0);
        rc = ft_extract_launch_a(L, 0, 2 * B, nullptr);
        if (rc == FT_OK) rc = ft_extract_launch_octree(L, 0, 0, 2 * B, L->evA[0]);
        if (rc != FT_OK) return rc;
        hipStream_t st = L->streamB;
        FT_HIP(hipStreamWaitEvent(st, L->evA[0], 0));
        rc = ft_extract_launch_b(L, 0, 2 * B, st);
        if (rc != FT_OK) return rc;
        const size_t oR = (size_t)B * g.maxKp;  // the right camera's half of the left extractor's arrays
        FtStereoArgs a;
        a.keysL = L->d_keys;
        a.keysR = L->d_keys + oR;
        a.descL = L->d_desc;
        a.descR = L->d_desc + oR * 32;
        a.nL = L->d_nSel;
        a.nR = L->d_nSel + B;
        a.capacity = g.maxKp;
        a.mbf = fe->mbf;
        a.mb = fe->mb;
        a.uright = fe->d_uright;
        a.depth = fe->d_depth;
        a.sad = fe->d_sad;
        a.hamIdx = nullptr;
        a.nMatches = fe->d_nMatches;
        a.applyMedianCut = 1;
        a.rowStride = L->height + 2;
        a.rowStart = fe->d_rowStart;
        a.sorted = fe->d_sorted;
        a.alignedLoads = L->l0Aligned;
        rc = ft_launch_stereo_rowsort(st, g, B, a);
        if (rc == FT_OK)
            rc = ft_launch_stereo_match(st, g, B, L->d_l0, L->d_l0 + B, L->l0pitch, L->l0pitch, L->d_pyr,
                                        L->d_pyr + (size_t)B * g.pyrPerSlot, a);
        if (rc == FT_OK) rc = ft_launch_stereo_median(st, B, a);
        if (rc != FT_OK) return rc;
        FtDeliverArgs d;
        d.keysL = a.keysL;
        d.keysR = a.keysR;
        d.descL = a.descL;
        d.descR = a.descR;
        d.uright = fe->d_uright;
        d.depth = fe->d_depth;
        d.nL = a.nL;
        d.nR = a.nR;
        d.nMatches = fe->d_nMatches;
        d.overflowL = d.overflowR = L->d_overflow;
        d.oKeysL = direct ? keysL : L->h_keys;
        d.oKeysR = direct ? keysR : R->h_keys;  // the right extractor lends its pinned staging and counters
        d.oDescL = direct ? descL : L->h_desc;
        d.oDescR = direct ? descR : R->h_desc;
        d.oUright = direct ? uright : fe->h_uright;
        d.oDepth = direct ? depth : fe->h_depth;
        d.oNL = L->h_nSel;
        d.oNR = R->h_nSel;
        d.oNMatches = fe->h_nMatches;
        d.oOverflowL = L->h_overflow;
        d.oOverflowR = R->h_overflow;
        d.srcStride = g.maxKp;
        d.dstStride = direct ? capacity : g.maxKp;
        rc = ft_launch_deliver(st, B, d);
        if (rc != FT_OK) return rc;
        R->lastBatch = B;
        if (capture) {
            FT_HIP(hipEventRecord(fe->evJoin, st));
            FT_HIP(hipStreamWaitEvent(L->stream, fe->evJoin, 0));
        }
        fe->ctx->addStat("stereo.submit.total", tAll.ms());
        return FT_OK;
    }
    if (capture) {  // fork: everything the right camera enqueues hangs off the captured stream
        FT_HIP(hipEventRecord(fe->evFork, L->stream));
        FT_HIP(hipStreamWaitEvent(R->stream, fe->evFork, 0));
    }
    int rc = ft_extract_prepare(L, imagesL, batch, on_device, width, height, stride);
    if (rc != FT_OK) return rc;
    rc = ft_extract_prepare(R, imagesR, batch, on_device, width, height, stride);
    if (rc != FT_OK) return rc;
    // software pipeline over sub-batches: while the host distributes the keypoints of sub-batch s, the
    // GPU already runs pyramid + FAST of sub-batch s+1 (stage-A streams) and descriptors / matching of
    // sub-batch s-1 (stage-B streams)
    const int S = ft_pipeline_depth(L->tune, batch, L->deviceOctree && R->deviceOctree);
    const int sb = (batch + S - 1) / S;
    // device octree: candidates, selection and counts stay on the device, so the whole batch is enqueued
    // without a single host synchronisation; otherwise the host octree of sub-batch s runs between the stages
    const bool dev = L->deviceOctree && R->deviceOctree;
    if (!dev) L->deviceOctree = R->deviceOctree = false;
    if (dev) fe->ctx->addStat("stereo.device_octree_batches", 0);
    for (int s = 0, b0 = 0; b0 < batch; s++, b0 += sb) {
        const int nb = std::min(sb, batch - b0);
        rc = ft_extract_launch_a(L, b0, nb, dev ? nullptr : L->evA[s]);
        if (rc == FT_OK && dev) rc = ft_extract_launch_octree(L, s, b0, nb, L->evA[s]);
        if (rc != FT_OK) return rc;
        rc = ft_extract_launch_a(R, b0, nb, dev ? nullptr : R->evA[s]);
        if (rc == FT_OK && dev) rc = ft_extract_launch_octree(R, s, b0, nb, R->evA[s]);
        if (rc != FT_OK) return rc;
    }
    hipStream_t st = L->streamB;
    // latency mode: the results of a small batch are written to pinned host memory by one kernel (FtDeliverArgs); large
    // batches keep the DMA copies, which cost no compute units
    const bool deliver = dev && batch <= FT_GRAPH_MAX_BATCH;
    double tOct = 0, tWait = 0, tLaunch = 0;
    for (int s = 0, b0 = 0; b0 < batch; s++, b0 += sb) {
        const int nb = std::min(sb, batch - b0);
        if (dev) {
            FT_HIP(hipStreamWaitEvent(L->streamB, L->evA[s], 0));
            FT_HIP(hipStreamWaitEvent(R->streamB, R->evA[s], 0));
        } else {
            FtTimer tW;
            FT_HIP(hipEventSynchronize(L->evA[s]));
            FT_HIP(hipEventSynchronize(R->evA[s]));
            tWait += tW.ms();
            FtTimer tO;
            ft_extractor *both[2] = {L, R};
            rc = ft_extract_octree_multi(both, 2, b0, nb);  // one job for both cameras: one critical path
            if (rc != FT_OK) return rc;
            tOct += tO.ms();
        }
        FtTimer tL;
        rc = ft_extract_launch_b(L, b0, nb, L->streamB);
        if (rc != FT_OK) return rc;
        rc = ft_extract_launch_b(R, b0, nb, R->streamB);
        if (rc != FT_OK) return rc;
        rc = frontendMatchAndDeliver(fe, s, b0, nb, dev, deliver, direct, keysL, descL, keysR, descR, capacity, uright, depth);
        if (rc != FT_OK) return rc;
        tLaunch += tL.ms();
    }
    if (deliver) {
        FtDeliverArgs d;
        d.keysL = L->d_keys;
        d.keysR = R->d_keys;
        d.descL = L->d_desc;
        d.descR = R->d_desc;
        d.uright = fe->d_uright;
        d.depth = fe->d_depth;
        d.nL = L->d_nSel;
        d.nR = R->d_nSel;
        d.nMatches = fe->d_nMatches;
        d.overflowL = L->d_overflow;
        d.overflowR = R->d_overflow;
        d.oKeysL = direct ? keysL : L->h_keys;
        d.oKeysR = direct ? keysR : R->h_keys;
        d.oDescL = direct ? descL : L->h_desc;
        d.oDescR = direct ? descR : R->h_desc;
        d.oUright = direct ? uright : fe->h_uright;
        d.oDepth = direct ? depth : fe->h_depth;
        d.oNL = L->h_nSel;
        d.oNR = R->h_nSel;
        d.oNMatches = fe->h_nMatches;
        d.oOverflowL = L->h_overflow;
        d.oOverflowR = R->h_overflow;
        d.srcStride = g.maxKp;
        d.dstStride = direct ? capacity : g.maxKp;
        rc = ft_launch_deliver(st, batch, d);
        if (rc != FT_OK) return rc;
    } else {
        rc = ft_extract_finish_counts(L, batch, st);
        if (rc == FT_OK) rc = ft_extract_finish_counts(R, batch, st);
        if (rc != FT_OK) return rc;
    }
    if (capture) {  // join: the stage-B stream of the left camera is the tail of everything
        FT_HIP(hipEventRecord(fe->evJoin, st));
        FT_HIP(hipStreamWaitEvent(L->stream, fe->evJoin, 0));
    }
    if (!dev) fe->ctx->addStat("stereo.octree(host,both)", tOct);
    fe->ctx->addStat("stereo.host_wait_stageA", tWait);
    fe->ctx->addStat("stereo.host_launch_stageB", tLaunch);
    fe->ctx->addStat("stereo.submit.total", tAll.ms());
    return FT_OK;
}

int ft_stereo_frontend_submit(ft_stereo_frontend *fe, const uint8_t *const *imagesL, const uint8_t *const *imagesR,
                              int batch, int on_device, int width, int height, int stride, ft_keypoint *keysL,
                              uint8_t *descL, int *nL, ft_keypoint *keysR, uint8_t *descR, int *nR, int capacity,
                              float *uright, float *depth, int *n_matches) {
    FT_REQUIRE(fe, "null front end");
    FT_REQUIRE(!fe->pending.active, "stereo front end: a submitted batch has not been waited for");
    if (!imagesL || !imagesR || width <= 0 || height <= 0) {
        ft_set_error("stereo front end: empty image");
        return FT_ERR_EMPTY;
    }
    ft_extractor *L = fe->exL, *R = fe->exR;
    const FtGeom &g = L->geom;
    int rc = ft_set_device(fe->ctx);
    if (rc != FT_OK) return rc;
    FT_REQUIRE(batch >= 1 && batch <= fe->maxBatch, "stereo front end: batch outside [1, max_batch]");
    rc = ft_extract_foreign_wait(L);  // (a tracked batch bound to these extractors may still read their slots; never inside a capture)
    if (rc == FT_OK) rc = ft_extract_foreign_wait(R);
    if (rc != FT_OK) return rc;
    const bool dev = L->deviceOctree && R->deviceOctree;
    const bool paired = fe->pairedCapable && dev && batch <= FT_GRAPH_MAX_BATCH && 2 * batch <= L->maxBatch;
    const bool direct = isPinnedHost(keysL) && isPinnedHost(descL) && isPinnedHost(keysR) && isPinnedHost(descR) &&
                        isPinnedHost(uright) && isPinnedHost(depth) && (!dev || capacity >= g.maxKp);
    // ---- latency mode: small batches with a fixed call shape run as one captured graph ----
    const bool graphsOn = fe->exL->tune.graph != 0;
    bool useGraph = graphsOn && !fe->graphDisabled && L->ownStreams && R->ownStreams && dev && direct && !fe->ctx->kernelTiming && batch >= 1 &&
                    batch <= FT_GRAPH_MAX_BATCH && batch <= fe->maxBatch && width == L->width && height == L->height &&
                    stride >= width;
    bool launched = false;
    if (useGraph) {
        for (int b = 0; b < batch; b++)
            if (!imagesL[b] || !imagesR[b]) useGraph = false;
    }
    if (useGraph) {
        ft_stereo_frontend::GraphKey key;
        key.batch = batch; key.onDevice = on_device; key.width = width; key.height = height; key.stride = stride;
        key.capacity = capacity;
        key.paired = paired ? 1 : 0;
        key.bigGridL = L->bigGrid + (L->histOn ? (1 << 24) : 0);
        key.bigGridR = R->bigGrid + (R->histOn ? (1 << 24) : 0);
        key.alignedL = key.alignedR = 1;
        if (on_device) {
            if (stride & 3) key.alignedL = key.alignedR = 0;
            for (int b = 0; b < batch; b++) {
                if ((uintptr_t)imagesL[b] & 3) key.alignedL = 0;
                if ((uintptr_t)imagesR[b] & 3) key.alignedR = 0;
            }
            if (paired) key.alignedL = key.alignedR = key.alignedL & key.alignedR;  // one launch reads both cameras
        } else if (ft_extract_ensure_stage(L) != FT_OK || ft_extract_ensure_stage(R) != FT_OK) {
            (void)hipGetLastError();
            useGraph = false;
        }
        const void *outs[6] = {keysL, descL, keysR, descR, uright, depth};
        for (int i = 0; i < 6; i++) key.out[i] = outs[i];
        if (!useGraph) {
        } else if (fe->graphExec && key == fe->graphKey) {
            // replay: device frames change through the level-0 pointer tables, host frames through the pinned staging
            // the captured uploads read
            if (paired) {
                std::vector<const uint8_t *> imgs(imagesL, imagesL + batch);
                imgs.insert(imgs.end(), imagesR, imagesR + batch);
                if (on_device)
                    for (int b = 0; b < 2 * batch; b++) L->h_l0[b] = imgs[b];
                else
                    ft_extract_restage(L, imgs.data(), 2 * batch, width, height, stride);
                L->lastBatch = 2 * batch;
                R->lastBatch = batch;
                fe->lastPaired = true;
            } else {
                if (on_device) {
                    for (int b = 0; b < batch; b++) {
                        L->h_l0[b] = imagesL[b];
                        R->h_l0[b] = imagesR[b];
                    }
                } else {
                    ft_extract_restage(L, imagesL, batch, width, height, stride);
                    ft_extract_restage(R, imagesR, batch, width, height, stride);
                }
                L->lastBatch = R->lastBatch = batch;
                fe->lastPaired = false;
            }
            fe->ctx->addStat("stereo.device_octree_batches", 0);
            FtTimer tG;
            FT_HIP(hipGraphLaunch(fe->graphExec, L->stream));
            fe->ctx->addStat("stereo.graph_launch", tG.ms());
            launched = true;
        } else {
            if (fe->graphExec) {
                hipGraphExecDestroy(fe->graphExec);
                fe->graphExec = nullptr;
            }
            hipGraph_t graph = nullptr;
            L->stageHost = R->stageHost = !on_device;
            hipError_t ce = hipStreamBeginCapture(L->stream, hipStreamCaptureModeThreadLocal);
            if (ce == hipSuccess) {
                rc = frontendEnqueue(fe, imagesL, imagesR, batch, on_device, width, height, stride, keysL, descL, nL, keysR,
                                     descR, nR, capacity, uright, depth, n_matches, direct, 1, paired);
                ce = hipStreamEndCapture(L->stream, &graph);
                if (rc == FT_OK && ce == hipSuccess && graph) ce = hipGraphInstantiate(&fe->graphExec, graph, nullptr, nullptr, 0);
                if (graph) hipGraphDestroy(graph);
            }
            L->stageHost = R->stageHost = false;
            if (rc != FT_OK || ce != hipSuccess || !fe->graphExec) {
                // capture is an optimisation: fall back to plain enqueueing for good
                (void)hipGetLastError();
                fe->graphDisabled = true;
                fe->graphExec = nullptr;
                fe->ctx->addStat("stereo.graph_capture_failed", 0);
            } else {
                fe->graphKey = key;
                fe->ctx->addStat("stereo.graph_captures", 0);
                FT_HIP(hipGraphLaunch(fe->graphExec, L->stream));
                launched = true;
            }
        }
    }
    if (!launched) {
        rc = frontendEnqueue(fe, imagesL, imagesR, batch, on_device, width, height, stride, keysL, descL, nL, keysR, descR,
                             nR, capacity, uright, depth, n_matches, direct, 0, paired);
        if (rc != FT_OK) return rc;
        // the end of this batch, marked NOW: ft_stereo_frontend_wait waits for these events and not for the streams - a
        // stream synchronisation enqueues its marker at the time of the call, behind whatever another front end has put
        // into the same hardware queue since (the runtime multiplexes all streams onto a few in-order queues), so waiting
        // for batch k-1 returned only when batch k was nearly through and the upload of batch k+1 started that late
        FT_HIP(hipEventRecord(fe->evDone[0], L->streamB));
        FT_HIP(hipEventRecord(fe->evDone[1], R->streamB));
    }
    // everything is enqueued; ft_stereo_frontend_wait waits for the end of the batch and finishes the outputs
    auto &P = fe->pending;
    P.imagesL.assign(imagesL, imagesL + batch);
    P.imagesR.assign(imagesR, imagesR + batch);
    P.onDevice = on_device;
    P.width = width;
    P.height = height;
    P.stride = stride;
    P.active = true;
    P.graph = launched;
    P.batch = batch;
    P.capacity = capacity;
    P.direct = direct;
    P.keysL = keysL; P.descL = descL; P.nL = nL;
    P.keysR = keysR; P.descR = descR; P.nR = nR;
    P.uright = uright; P.depth = depth; P.nMatches = n_matches;
    return FT_OK;
}

int ft_stereo_frontend_wait(ft_stereo_frontend *fe) {
    FT_REQUIRE(fe, "null front end");
    auto &P = fe->pending;
    if (!P.active) return FT_OK;
    int rc = ft_set_device(fe->ctx);
    if (rc != FT_OK) return rc;
    ft_extractor *L = fe->exL, *R = fe->exR;
    const FtGeom &g = L->geom;
    FtTimer tTail;
    P.active = false;
    if (P.graph) {
        FT_HIP(hipStreamSynchronize(L->stream));  // a captured batch completes on the stream it was launched on
        FT_HIP(hipStreamSynchronize(L->streamB));
        FT_HIP(hipStreamSynchronize(R->streamB));
    } else {
        FT_HIP(hipEventSynchronize(fe->evDone[0]));
        FT_HIP(hipEventSynchronize(fe->evDone[1]));
    }
    fe->ctx->addStat("stereo.host_tail_sync", tTail.ms());
    L->evt.resolve(fe->ctx);
    R->evt.resolve(fe->ctx);
    if (L->deviceOctree) ft_extract_update_big_grid(L);
    if (R->deviceOctree) ft_extract_update_big_grid(R);
    if ((L->deviceOctree && L->h_overflow[0]) || (R->deviceOctree && R->h_overflow[0])) {
        // a level of some image exceeded the device octree's limits (more than 16 384 candidates, or more than
        // FT_OCT_BIGCAP levels of the launch beyond 4 096)
        std::vector<int> sl, sr;
        std::vector<char> bad(P.batch, 0);
        int nBad = 0;
        if (!(P.graph || fe->lastPaired)) {
            rc = ft_extract_overflow_slots(L, P.batch, sl);
            if (rc == FT_OK) rc = ft_extract_overflow_slots(R, P.batch, sr);
            if (rc != FT_OK) return rc;
            for (int b : sl) bad[b] |= 1;
            for (int b : sr) bad[b] |= 2;
            for (int b = 0; b < P.batch; b++) nBad += bad[b] ? 1 : 0;
        }
        if (P.graph || fe->lastPaired || 4 * nBad > P.batch) {
            // latency mode (at most 8 pairs, possibly both cameras in one extractor), or most of the batch is beyond the
            // device octree: the whole batch is redone with the host-octree pipeline (sub-batches of 16 pairs, the host
            // octree of one overlapping with the kernels of the next).  Same inputs, same outputs, only slower.
            fe->ctx->addStat("stereo.device_octree_fallbacks", 1);
            for (int k = 1; k < nBad; k++) fe->ctx->addStat("stereo.device_octree_fallbacks", 1);
            L->h_overflow[0] = R->h_overflow[0] = 0;
            FT_HIP(hipMemset(L->d_overflow, 0, sizeof(int)));
            FT_HIP(hipMemset(R->d_overflow, 0, sizeof(int)));
            FT_HIP(hipMemset(L->d_ovSlot, 0, sizeof(int) * L->maxBatch));
            FT_HIP(hipMemset(R->d_ovSlot, 0, sizeof(int) * R->maxBatch));
            const bool dl = L->deviceOctree, dr = R->deviceOctree;
            L->deviceOctree = R->deviceOctree = false;
            std::vector<const uint8_t *> il = P.imagesL, ir = P.imagesR;
            rc = ft_stereo_frontend_submit(fe, il.data(), ir.data(), P.batch, P.onDevice, P.width, P.height, P.stride, P.keysL,
                                           P.descL, P.nL, P.keysR, P.descR, P.nR, P.capacity, P.uright, P.depth, P.nMatches);
            if (rc == FT_OK) rc = ft_stereo_frontend_wait(fe);
            L->deviceOctree = dl;
            R->deviceOctree = dr;
            return rc;
        }
        // throughput mode, a few pairs concerned: only they are redone, in place - host octree of the camera(s) that
        // overflowed (all of them as one parallel job), then descriptors, matching and result copies of each such pair;
        // every other pair keeps its device results
        std::vector<std::pair<ft_extractor *, int>> jobs;
        for (int b : sl) jobs.emplace_back(L, b);
        for (int b : sr) jobs.emplace_back(R, b);
        rc = ft_extract_repair_prepare(jobs);
        if (rc != FT_OK) return rc;
        for (int b = 0; b < P.batch; b++) {
            if (!bad[b]) continue;
            fe->ctx->addStat("stereo.device_octree_fallbacks", 1);
            if (bad[b] & 1) rc = ft_extract_repair_launch(L, b, L->streamB);
            if (rc == FT_OK && (bad[b] & 2)) rc = ft_extract_repair_launch(R, b, R->streamB);
            if (rc == FT_OK)
                rc = frontendMatchAndDeliver(fe, 0, b, 1, true, false, P.direct, P.keysL, P.descL, P.keysR, P.descR, P.capacity,
                                             P.uright, P.depth);
            if (rc != FT_OK) return rc;
        }
        FT_HIP(hipStreamSynchronize(L->streamB));
        FT_HIP(hipStreamSynchronize(R->streamB));
        L->evt.resolve(fe->ctx);
        R->evt.resolve(fe->ctx);
    }
    const size_t kp = sizeof(ft_keypoint);
    const int capacity = P.capacity;
    for (int b = 0; b < P.batch; b++) {
        const int nl = L->h_nSel[b], nr = R->h_nSel[b];
        if (nl > capacity || nr > capacity) {
            ft_set_error("stereo front end: output capacity too small (use ft_extractor_max_keypoints)");
            return FT_ERR_CAPACITY;
        }
        if (!P.direct) {
            if (P.keysL) memcpy(P.keysL + (size_t)b * capacity, L->h_keys + (size_t)b * g.maxKp, kp * nl);
            if (P.descL) memcpy(P.descL + (size_t)b * capacity * 32, L->h_desc + (size_t)b * g.maxKp * 32, (size_t)32 * nl);
            if (P.keysR) memcpy(P.keysR + (size_t)b * capacity, R->h_keys + (size_t)b * g.maxKp, kp * nr);
            if (P.descR) memcpy(P.descR + (size_t)b * capacity * 32, R->h_desc + (size_t)b * g.maxKp * 32, (size_t)32 * nr);
            if (P.uright) memcpy(P.uright + (size_t)b * capacity, fe->h_uright + (size_t)b * g.maxKp, 4 * (size_t)nl);
            if (P.depth) memcpy(P.depth + (size_t)b * capacity, fe->h_depth + (size_t)b * g.maxKp, 4 * (size_t)nl);
        }
        if (P.nL) P.nL[b] = nl;
        if (P.nR) P.nR[b] = nr;
        if (P.nMatches) P.nMatches[b] = fe->h_nMatches[b];
    }
    return FT_OK;
}

int ft_stereo_frontend_process(ft_stereo_frontend *fe, const uint8_t *const *imagesL, const uint8_t *const *imagesR,
                               int batch, int on_device, int width, int height, int stride, ft_keypoint *keysL,
                               uint8_t *descL, int *nL, ft_keypoint *keysR, uint8_t *descR, int *nR, int capacity,
                               float *uright, float *depth, int *n_matches) {
    FtTimer tAll;
    int rc = ft_stereo_frontend_submit(fe, imagesL, imagesR, batch, on_device, width, height, stride, keysL, descL, nL,
                                       keysR, descR, nR, capacity, uright, depth, n_matches);
    if (rc != FT_OK) return rc;
    rc = ft_stereo_frontend_wait(fe);
    if (rc == FT_OK) fe->ctx->addStat("stereo.process.total", tAll.ms());
    return rc;
}

// ---- fisheye 2-NN and descriptor distance --------------------------------------------------------
int ft_fisheye_match(ft_context *ctx, const uint8_t *descL, int nL, const uint8_t *descR, int nR, int *matches,
                     int *best, int *second) {
    FT_REQUIRE(ctx && matches, "ft_fisheye_match: null argument");
    FT_REQUIRE(nL >= 0 && nR >= 0 && nR < (1 << 20), "ft_fisheye_match: count out of range");
    FT_REQUIRE((nL == 0 || descL) && (nR == 0 || descR), "ft_fisheye_match: null descriptors");
    if (nL == 0) return FT_OK;
    int rc = ft_set_device(ctx);
    if (rc != FT_OK) return rc;
    std::lock_guard<std::mutex> lk(ctx->matchMutex);
    // layout in the context's grow-only scratch: descL | descR | matches, best, second
    const size_t oL = 0, oR = (((size_t)32 * nL) + 63) & ~(size_t)63, oOut = oR + ((((size_t)32 * std::max(nR, 1)) + 63) & ~(size_t)63);
    const size_t total = oOut + sizeof(int) * 3 * (size_t)nL;
    rc = ft_ensure_scratch(ctx, total, total);
    if (rc != FT_OK) return rc;
    uint8_t *dev = (uint8_t *)ctx->scratchDev, *pin = (uint8_t *)ctx->scratchPin;
    int *dOut = (int *)(dev + oOut), *hOut = (int *)(pin + oOut);
    hipStream_t st = ctx->stream;
    // pageable descriptors go up through the context's pinned scratch (one host memcpy each) instead of being staged and
    // waited for by the runtime copy by copy; descriptors the caller keeps in pinned memory are read where they are
    auto up = [&](size_t off, const uint8_t *src, size_t bytes) -> int {
        const uint8_t *from = src;
        if (!ft_is_pinned_host_range(src, bytes)) {
            memcpy(pin + off, src, bytes);
            from = pin + off;
        }
        FT_HIP(hipMemcpyAsync(dev + off, from, bytes, hipMemcpyHostToDevice, st));
        return FT_OK;
    };
    if ((rc = up(oL, descL, (size_t)32 * nL)) != FT_OK) return rc;
    if (nR > 0 && (rc = up(oR, descR, (size_t)32 * nR)) != FT_OK) return rc;
    rc = ft_launch_fisheye(st, dev + oL, nL, dev + oR, nR, dOut, dOut + nL, dOut + 2 * (size_t)nL);
    if (rc != FT_OK) return rc;
    FT_HIP(hipMemcpyAsync(hOut, dOut, sizeof(int) * 3 * (size_t)nL, hipMemcpyDeviceToHost, st));
    FT_HIP(hipStreamSynchronize(st));
    memcpy(matches, hOut, sizeof(int) * nL);
    if (best) memcpy(best, hOut + nL, sizeof(int) * nL);
    if (second) memcpy(second, hOut + 2 * (size_t)nL, sizeof(int) * nL);
    return FT_OK;
}

int ft_fisheye_stereo(ft_context *ctx, const ft_fisheye_rig *rig, const uint8_t *descL, const ft_keypoint *keysL, int nL,
                      const uint8_t *descR, const ft_keypoint *keysR, int nR, const float *level_sigma2, int nlevels,
                      int *matches, float *depth, float *p3d, int *n_matches) {
    FT_REQUIRE(ctx && rig && matches && depth && p3d && level_sigma2, "ft_fisheye_stereo: null argument");
    FT_REQUIRE(nL >= 0 && nR >= 0 && nR < (1 << 20), "ft_fisheye_stereo: count out of range");
    FT_REQUIRE(nlevels >= 1 && nlevels <= FT_MAX_LEVELS, "ft_fisheye_stereo: nlevels out of range");
    FT_REQUIRE((nL == 0 || (descL && keysL)) && (nR == 0 || (descR && keysR)), "ft_fisheye_stereo: null keypoints / descriptors");
    for (int i = 0; i < nL; i++) FT_REQUIRE(keysL[i].octave >= 0 && keysL[i].octave < nlevels, "ft_fisheye_stereo: left octave out of range");
    for (int i = 0; i < nR; i++) FT_REQUIRE(keysR[i].octave >= 0 && keysR[i].octave < nlevels, "ft_fisheye_stereo: right octave out of range");
    if (n_matches) *n_matches = 0;
    if (nL == 0) return FT_OK;
    int rc = ft_set_device(ctx);
    if (rc != FT_OK) return rc;
    FtFisheyeRig G;
    memset(&G, 0, sizeof G);
    memcpy(G.cam1, rig->cam1, sizeof G.cam1);
    memcpy(G.cam2, rig->cam2, sizeof G.cam2);
    G.precision = rig->precision;
    memcpy(G.Rlr, rig->Rlr, sizeof G.Rlr);
    memcpy(G.tlr, rig->tlr, sizeof G.tlr);
    for (int i = 0; i < nlevels; i++) G.sigma2[i] = level_sigma2[i];
    std::lock_guard<std::mutex> lk(ctx->matchMutex);
    auto up = [](size_t v) { return (v + 63) & ~(size_t)63; };
    const size_t nRa = (size_t)std::max(nR, 1);
    const size_t oDL = 0, oDR = up((size_t)32 * nL), oKL = oDR + up(32 * nRa), oKR = oKL + up(sizeof(ft_keypoint) * nL),
                 oOut = oKR + up(sizeof(ft_keypoint) * nRa);
    // outputs: matches | best | second | count(+pad) | depth | p3d
    const size_t outInts = 3 * (size_t)nL + 16, outBytes = sizeof(int) * outInts + sizeof(float) * 4 * (size_t)nL;
    rc = ft_ensure_scratch(ctx, oOut + outBytes, outBytes);
    if (rc != FT_OK) return rc;
    uint8_t *dev = (uint8_t *)ctx->scratchDev, *pin = (uint8_t *)ctx->scratchPin;
    int *dM = (int *)(dev + oOut), *dCount = dM + 3 * (size_t)nL;
    float *dOut = (float *)(dM + outInts);
    hipStream_t st = ctx->stream;
    FT_HIP(hipMemcpyAsync(dev + oDL, descL, (size_t)32 * nL, hipMemcpyHostToDevice, st));
    FT_HIP(hipMemcpyAsync(dev + oKL, keysL, sizeof(ft_keypoint) * nL, hipMemcpyHostToDevice, st));
    if (nR > 0) {
        FT_HIP(hipMemcpyAsync(dev + oDR, descR, (size_t)32 * nR, hipMemcpyHostToDevice, st));
        FT_HIP(hipMemcpyAsync(dev + oKR, keysR, sizeof(ft_keypoint) * nR, hipMemcpyHostToDevice, st));
    }
    FT_HIP(hipMemsetAsync(dCount, 0, sizeof(int), st));
    rc = ft_launch_fisheye(st, dev + oDL, nL, dev + oDR, nR, dM, dM + nL, dM + 2 * (size_t)nL);
    if (rc == FT_OK)
        rc = ft_launch_fisheye_triangulate(st, G, (const ft_keypoint *)(dev + oKL), nL, (const ft_keypoint *)(dev + oKR), dM, dOut,
                                           dOut + nL, dCount);
    if (rc != FT_OK) return rc;
    FT_HIP(hipMemcpyAsync(pin, dev + oOut, outBytes, hipMemcpyDeviceToHost, st));
    FT_HIP(hipStreamSynchronize(st));
    const int *hM = (const int *)pin;
    const float *hOut = (const float *)(hM + outInts);
    memcpy(matches, hM, sizeof(int) * nL);
    memcpy(depth, hOut, sizeof(float) * nL);
    memcpy(p3d, hOut + nL, sizeof(float) * 3 * (size_t)nL);
    if (n_matches) *n_matches = hM[3 * (size_t)nL];
    return FT_OK;
}

int ft_stereo_frontend_device_descriptors(ft_stereo_frontend *fe, int slot, int right, const uint8_t **dptr, int *n) {
    FT_REQUIRE(fe && dptr && n, "ft_stereo_frontend_device_descriptors: null argument");
    ft_extractor *ex = right ? fe->exR : fe->exL;
    FT_REQUIRE(slot >= 0 && slot < fe->exR->lastBatch, "ft_stereo_frontend_device_descriptors: slot holds no result");
    // a paired batch keeps the right camera's results in the upper half of the left extractor's arrays
    const int devSlot = (right && fe->lastPaired) ? fe->exR->lastBatch + slot : slot;
    *dptr = (fe->lastPaired ? fe->exL : ex)->d_desc + (size_t)devSlot * ex->geom.maxKp * 32;
    *n = ex->h_nSel[slot];
    return FT_OK;
}

int ft_descriptor_distance(ft_context *ctx, const uint8_t *a, const uint8_t *b, int n, int *dist) {
    FT_REQUIRE(ctx && n >= 0 && (n == 0 || (a && b && dist)), "ft_descriptor_distance: bad argument");
    if (n == 0) return FT_OK;
    int rc = ft_set_device(ctx);
    if (rc != FT_OK) return rc;
    std::lock_guard<std::mutex> lk(ctx->matchMutex);
    const size_t oA = 0, oB = (((size_t)32 * n) + 63) & ~(size_t)63, oD = 2 * oB;
    rc = ft_ensure_scratch(ctx, oD + sizeof(int) * (size_t)n, sizeof(int) * (size_t)n);
    if (rc != FT_OK) return rc;
    uint8_t *dev = (uint8_t *)ctx->scratchDev;
    hipStream_t st = ctx->stream;
    FT_HIP(hipMemcpyAsync(dev + oA, a, (size_t)32 * n, hipMemcpyHostToDevice, st));
    FT_HIP(hipMemcpyAsync(dev + oB, b, (size_t)32 * n, hipMemcpyHostToDevice, st));
    rc = ft_launch_hamming_pairs(st, dev + oA, dev + oB, n, (int *)(dev + oD));
    if (rc != FT_OK) return rc;
    FT_HIP(hipMemcpyAsync(ctx->scratchPin, dev + oD, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st));
    FT_HIP(hipStreamSynchronize(st));
    memcpy(dist, ctx->scratchPin, sizeof(int) * (size_t)n);
    return FT_OK;
}

}  // extern "C"
