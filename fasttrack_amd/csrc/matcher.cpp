// Host side of the stereo matchers: replaces KernelController::launchStereoMatchKernel /
// launchFisheyeStereoMatchKernel (reference include/Kernels/KernelController.h:31-38) and offers the
// fused extract+match front end used for throughput.  All compute is in kernels_match.hip.
#include <algorithm>
#include <cstring>

#include "ft_host.h"

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
        exL->d_stKeys = nullptr;
        exL->d_stDesc = nullptr;
        exL->d_stOut = nullptr;
        exL->d_stInt = nullptr;
        FT_HIP(hipMalloc((void **)&exL->d_stKeys, sizeof(ft_keypoint) * 2 * need));
        FT_HIP(hipMalloc((void **)&exL->d_stDesc, (size_t)64 * need));
        FT_HIP(hipMalloc((void **)&exL->d_stOut, sizeof(float) * 2 * need));
        FT_HIP(hipMalloc((void **)&exL->d_stInt, sizeof(int) * (2 * need + 4)));
        exL->stCap = need;
    }
    const int C = exL->stCap;
    hipStream_t st = exL->stream;
    FT_HIP(hipStreamSynchronize(exR->stream));  // right pyramid must be complete
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
    int rc = ft_extractor_create(ctx, nfeatures, scale_factor, nlevels, ini_th_fast, min_th_fast, image_width,
                                 image_height, max_batch, &fe->exL);
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
    if (e == hipSuccess) e = hipHostMalloc((void **)&fe->h_uright, sizeof(float) * n, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&fe->h_depth, sizeof(float) * n, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&fe->h_nMatches, sizeof(int) * max_batch, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&fe->evR, hipEventDisableTiming);
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
    hipHostFree(fe->h_uright);
    hipHostFree(fe->h_depth);
    hipHostFree(fe->h_nMatches);
    if (fe->evR) hipEventDestroy(fe->evR);
    delete fe;
    return FT_OK;
}

ft_extractor *ft_stereo_frontend_left(ft_stereo_frontend *fe) { return fe ? fe->exL : nullptr; }
ft_extractor *ft_stereo_frontend_right(ft_stereo_frontend *fe) { return fe ? fe->exR : nullptr; }

int ft_stereo_frontend_process(ft_stereo_frontend *fe, const uint8_t *const *imagesL, const uint8_t *const *imagesR,
                               int batch, int on_device, int width, int height, int stride, ft_keypoint *keysL,
                               uint8_t *descL, int *nL, ft_keypoint *keysR, uint8_t *descR, int *nR, int capacity,
                               float *uright, float *depth, int *n_matches) {
    FT_REQUIRE(fe, "null front end");
    if (!imagesL || !imagesR || width <= 0 || height <= 0) {
        ft_set_error("stereo front end: empty image");
        return FT_ERR_EMPTY;
    }
    ft_extractor *L = fe->exL, *R = fe->exR;
    const FtGeom &g = L->geom;
    FtTimer tAll;
    int rc = ft_extract_stage_a(L, imagesL, batch, on_device, width, height, stride);
    if (rc != FT_OK) return rc;
    rc = ft_extract_stage_a(R, imagesR, batch, on_device, width, height, stride);
    if (rc != FT_OK) return rc;
    FT_HIP(hipStreamSynchronize(L->stream));
    FtTimer tO;
    rc = ft_extract_octree(L, batch);  // overlaps with the right image's stage A still running
    if (rc != FT_OK) return rc;
    FT_HIP(hipStreamSynchronize(R->stream));
    rc = ft_extract_octree(R, batch);
    if (rc != FT_OK) return rc;
    fe->ctx->addStat("stereo.octree(host,both)", tO.ms());
    rc = ft_extract_stage_b(L, batch);
    if (rc != FT_OK) return rc;
    rc = ft_extract_stage_b(R, batch);
    if (rc != FT_OK) return rc;
    FT_HIP(hipEventRecord(fe->evR, R->stream));
    FT_HIP(hipStreamWaitEvent(L->stream, fe->evR, 0));
    // keypoints stay on the device between extraction and matching: pinhole stereo passes the lapping
    // area (0,0) (src/Frame.cc:127), no keypoint has x == 0, so mono order == extraction order.
    FtStereoArgs a;
    a.keysL = L->d_keys;
    a.keysR = R->d_keys;
    a.descL = L->d_desc;
    a.descR = R->d_desc;
    a.nL = L->d_nSel;
    a.nR = R->d_nSel;
    a.capacity = g.maxKp;
    a.mbf = fe->mbf;
    a.mb = fe->mb;
    a.uright = fe->d_uright;
    a.depth = fe->d_depth;
    a.sad = fe->d_sad;
    a.hamIdx = nullptr;
    a.nMatches = fe->d_nMatches;
    a.applyMedianCut = 1;
    const bool tm = fe->ctx->kernelTiming;
    L->evt.begin(tm, "kernel.stereo_match", L->stream);
    rc = ft_launch_stereo_match(L->stream, g, batch, L->d_l0, R->d_l0, L->l0pitch, R->l0pitch, L->d_pyr, R->d_pyr, a);
    L->evt.end(tm, L->stream);
    if (rc != FT_OK) return rc;
    L->evt.begin(tm, "kernel.stereo_median", L->stream);
    rc = ft_launch_stereo_median(L->stream, batch, a);
    L->evt.end(tm, L->stream);
    if (rc != FT_OK) return rc;
    int maxNL = 0, maxNR = 0;
    for (int b = 0; b < batch; b++) {
        maxNL = std::max(maxNL, L->h_nSel[b]);
        maxNR = std::max(maxNR, R->h_nSel[b]);
    }
    hipStream_t st = L->stream;
    const size_t kp = sizeof(ft_keypoint);
    if (maxNL > 0) {
        FT_HIP(hipMemcpy2DAsync(L->h_keys, kp * g.maxKp, L->d_keys, kp * g.maxKp, kp * maxNL, batch, hipMemcpyDeviceToHost, st));
        FT_HIP(hipMemcpy2DAsync(L->h_desc, (size_t)32 * g.maxKp, L->d_desc, (size_t)32 * g.maxKp, (size_t)32 * maxNL, batch, hipMemcpyDeviceToHost, st));
        FT_HIP(hipMemcpy2DAsync(fe->h_uright, 4 * (size_t)g.maxKp, fe->d_uright, 4 * (size_t)g.maxKp, 4 * (size_t)maxNL, batch, hipMemcpyDeviceToHost, st));
        FT_HIP(hipMemcpy2DAsync(fe->h_depth, 4 * (size_t)g.maxKp, fe->d_depth, 4 * (size_t)g.maxKp, 4 * (size_t)maxNL, batch, hipMemcpyDeviceToHost, st));
    }
    if (maxNR > 0) {
        FT_HIP(hipMemcpy2DAsync(R->h_keys, kp * g.maxKp, R->d_keys, kp * g.maxKp, kp * maxNR, batch, hipMemcpyDeviceToHost, st));
        FT_HIP(hipMemcpy2DAsync(R->h_desc, (size_t)32 * g.maxKp, R->d_desc, (size_t)32 * g.maxKp, (size_t)32 * maxNR, batch, hipMemcpyDeviceToHost, st));
    }
    FT_HIP(hipMemcpyAsync(fe->h_nMatches, fe->d_nMatches, sizeof(int) * batch, hipMemcpyDeviceToHost, st));
    FT_HIP(hipStreamSynchronize(st));
    L->evt.resolve(fe->ctx);
    R->evt.resolve(fe->ctx);
    for (int b = 0; b < batch; b++) {
        const int nl = L->h_nSel[b], nr = R->h_nSel[b];
        if (nl > capacity || nr > capacity) {
            ft_set_error("stereo front end: output capacity too small (use ft_extractor_max_keypoints)");
            return FT_ERR_CAPACITY;
        }
        if (keysL) memcpy(keysL + (size_t)b * capacity, L->h_keys + (size_t)b * g.maxKp, kp * nl);
        if (descL) memcpy(descL + (size_t)b * capacity * 32, L->h_desc + (size_t)b * g.maxKp * 32, (size_t)32 * nl);
        if (keysR) memcpy(keysR + (size_t)b * capacity, R->h_keys + (size_t)b * g.maxKp, kp * nr);
        if (descR) memcpy(descR + (size_t)b * capacity * 32, R->h_desc + (size_t)b * g.maxKp * 32, (size_t)32 * nr);
        if (uright) memcpy(uright + (size_t)b * capacity, fe->h_uright + (size_t)b * g.maxKp, 4 * (size_t)nl);
        if (depth) memcpy(depth + (size_t)b * capacity, fe->h_depth + (size_t)b * g.maxKp, 4 * (size_t)nl);
        if (nL) nL[b] = nl;
        if (nR) nR[b] = nr;
        if (n_matches) n_matches[b] = fe->h_nMatches[b];
    }
    fe->ctx->addStat("stereo.process.total", tAll.ms());
    return FT_OK;
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
    uint8_t *dL = nullptr, *dR = nullptr;
    int *dOut = nullptr;
    hipError_t e = hipMalloc((void **)&dL, (size_t)32 * nL);
    if (e == hipSuccess) e = hipMalloc((void **)&dR, (size_t)32 * std::max(nR, 1));
    if (e == hipSuccess) e = hipMalloc((void **)&dOut, sizeof(int) * 3 * nL);
    hipStream_t st = ctx->stream;
    if (e == hipSuccess) e = hipMemcpyAsync(dL, descL, (size_t)32 * nL, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && nR > 0) e = hipMemcpyAsync(dR, descR, (size_t)32 * nR, hipMemcpyHostToDevice, st);
    rc = FT_OK;
    if (e == hipSuccess) rc = ft_launch_fisheye(st, dL, nL, dR, nR, dOut, dOut + nL, dOut + 2 * nL);
    if (e == hipSuccess && rc == FT_OK) e = hipMemcpyAsync(matches, dOut, sizeof(int) * nL, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && rc == FT_OK && best) e = hipMemcpyAsync(best, dOut + nL, sizeof(int) * nL, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && rc == FT_OK && second) e = hipMemcpyAsync(second, dOut + 2 * nL, sizeof(int) * nL, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(dL);
    hipFree(dR);
    hipFree(dOut);
    if (e != hipSuccess) return ft_hip_fail(e, "ft_fisheye_match", __FILE__, __LINE__);
    return rc;
}

int ft_descriptor_distance(ft_context *ctx, const uint8_t *a, const uint8_t *b, int n, int *dist) {
    FT_REQUIRE(ctx && n >= 0 && (n == 0 || (a && b && dist)), "ft_descriptor_distance: bad argument");
    if (n == 0) return FT_OK;
    int rc = ft_set_device(ctx);
    if (rc != FT_OK) return rc;
    uint8_t *dA = nullptr, *dB = nullptr;
    int *dD = nullptr;
    hipStream_t st = ctx->stream;
    hipError_t e = hipMalloc((void **)&dA, (size_t)32 * n);
    if (e == hipSuccess) e = hipMalloc((void **)&dB, (size_t)32 * n);
    if (e == hipSuccess) e = hipMalloc((void **)&dD, sizeof(int) * n);
    if (e == hipSuccess) e = hipMemcpyAsync(dA, a, (size_t)32 * n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dB, b, (size_t)32 * n, hipMemcpyHostToDevice, st);
    rc = FT_OK;
    if (e == hipSuccess) rc = ft_launch_hamming_pairs(st, dA, dB, n, dD);
    if (e == hipSuccess && rc == FT_OK) e = hipMemcpyAsync(dist, dD, sizeof(int) * n, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(dA);
    hipFree(dB);
    hipFree(dD);
    if (e != hipSuccess) return ft_hip_fail(e, "ft_descriptor_distance", __FILE__, __LINE__);
    return rc;
}

}  // extern "C"
