// Host orchestration of the ORB extractor: replaces ORB_SLAM3::ORBextractor (reference
// include/ORBextractor.h:100-197, src/ORBextractor.cc).  Geometry/tables are computed here exactly as
// the reference's CPU branch does (float arithmetic, cvRound = half-to-even); pixels are only ever
// touched by the HIP kernels in kernels_extract.hip.  Compiled with -ffp-contract=off.
#include <algorithm>
#include <map>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "ft_host.h"

namespace {

inline int cvRoundF(float v) { return (int)lrintf(v); }
inline short satShort(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

#define FT_REQUIRE(cond, msg)                      \
    do {                                           \
        if (!(cond)) {                             \
            ft_set_error(std::string(msg));        \
            return FT_ERR_INVALID;                 \
        }                                          \
    } while (0)

// cv::resize INTER_LINEAR coefficient tables: index + two 11-bit weights per output
// column / row.  scale = 1 / (dsize / ssize) in double, fx in float, weights via cvRound.
void buildTaps(int ssize, int dsize, bool clampIndex, FtTap *out) {
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        if (clampIndex) {  // columns: OpenCV resets the fraction at both ends; rows are only clipped
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        out[d].s = (short)s;
        out[d].a0 = satShort(cvRoundF((1.f - f) * 2048));
        out[d].a1 = satShort(cvRoundF(f * 2048));
        out[d].pad = 0;
    }
}

int buildGeometry(ft_extractor *ex, std::vector<FtTap> &taps) {
    const int L = ex->nlevels;
    FtGeom &g = ex->geom;
    memset(&g, 0, sizeof g);
    g.nlevels = L;
    // ORBextractor.cc:398-414, 445-452
    ex->sf.assign(L, 1.f);
    ex->sigma2.assign(L, 1.f);
    for (int i = 1; i < L; i++) {
        ex->sf[i] = ex->sf[i - 1] * ex->scaleFactor;
        ex->sigma2[i] = ex->sf[i] * ex->sf[i];
    }
    ex->invsf.resize(L);
    ex->invsigma2.resize(L);
    for (int i = 0; i < L; i++) {
        ex->invsf[i] = 1.0f / ex->sf[i];
        ex->invsigma2[i] = 1.0f / ex->sigma2[i];
        g.sf[i] = ex->sf[i];
        g.invsf[i] = ex->invsf[i];
    }
    // per-level quotas, ORBextractor.cc:454-465
    ex->quota.assign(L, 0);
    {
        float factor = 1.0f / ex->scaleFactor;
        float nDesired = ex->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)L));
        int sum = 0;
        for (int level = 0; level < L - 1; level++) {
            ex->quota[level] = cvRoundF(nDesired);
            sum += ex->quota[level];
            nDesired *= factor;
        }
        ex->quota[L - 1] = std::max(ex->nfeatures - sum, 0);
    }
    int pyrOff = 0, cellBase = 0, stageBase = 0, candBase = 0, tapOff = 0, maxKp = 0;
    ex->levelMax.assign(L, 0);
    for (int l = 0; l < L; l++) {
        FtLevelGeom &v = g.lv[l];
        // ORBextractor.cc:1499-1500
        v.w = cvRoundF((float)ex->width * ex->invsf[l]);
        v.h = cvRoundF((float)ex->height * ex->invsf[l]);
        FT_REQUIRE(v.w < 4096 && v.h < 4096, "image too large: candidate packing holds 12-bit coordinates");
        v.pitch = (v.w + 63) & ~63;
        v.off = pyrOff;
        pyrOff += v.pitch * v.h;
        pyrOff = (pyrOff + 255) & ~255;
        // ORBextractor.cc:1120-1134
        const int minB = FT_EDGE_THRESHOLD - 3;
        v.maxBX = v.w - FT_EDGE_THRESHOLD + 3;
        v.maxBY = v.h - FT_EDGE_THRESHOLD + 3;
        const float W = 35;
        const float width = (float)(v.maxBX - minB), height = (float)(v.maxBY - minB);
        int nCols = (int)(width / W), nRows = (int)(height / W);
        if (nCols < 1 || nRows < 1) {
            nCols = nRows = 0;  // level too small for one cell (the reference would divide by zero)
            v.wCell = v.hCell = 1;
        } else {
            v.wCell = (int)std::ceil(width / nCols);
            v.hCell = (int)std::ceil(height / nRows);
        }
        v.nCols = nCols;
        v.nRows = nRows;
        v.cellBase = cellBase;
        v.cellCap = ((v.wCell + 1) / 2) * ((v.hCell + 1) / 2);  // strict 3x3 maxima cannot be adjacent
        FT_REQUIRE(v.cellCap < 65536 && v.wCell + 6 < 256 && v.hCell + 6 < 256, "FAST cell too large for the cell records");
        g.fastLv[l] = (unsigned)v.pitch | ((unsigned)v.cellCap << 16);
        v.stageBase = stageBase;
        v.candBase = candBase;
        v.candCap = nCols * nRows * v.cellCap;
        cellBase += nCols * nRows;
        stageBase += v.candCap;
        candBase += v.candCap;
        if (l >= 1) {
            const FtLevelGeom &p = g.lv[l - 1];
            v.area2x = (p.w == 2 * v.w && p.h == 2 * v.h) ? 1 : 0;
            v.xtab = tapOff;
            tapOff += v.w;
            v.ytab = tapOff;
            tapOff += v.h;
        }
        ex->levelMax[l] = ft::octree_max_result(minB, v.maxBX, minB, v.maxBY, ex->quota[l]);
        maxKp += ex->levelMax[l];
    }
    g.totalCells = cellBase;
    g.stagePerSlot = stageBase;
    g.candPerSlot = candBase;
    g.pyrPerSlot = pyrOff;
    g.maxKp = maxKp;
    taps.assign(std::max(tapOff, 1), FtTap{0, 0, 0, 0});
    for (int l = 1; l < L; l++) {
        buildTaps(g.lv[l - 1].w, g.lv[l].w, true, &taps[g.lv[l].xtab]);
        buildTaps(g.lv[l - 1].h, g.lv[l].h, false, &taps[g.lv[l].ytab]);
    }
    return FT_OK;
}

template <typename T>
int devAlloc(T **p, size_t n) {
    FT_HIP(hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    return FT_OK;
}
template <typename T>
int pinAlloc(T **p, size_t n, unsigned flags = hipHostMallocDefault) {
    FT_HIP(hipHostMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T), flags));
    return FT_OK;
}

void freeAll(ft_extractor *ex) {
    hipSetDevice(ex->ctx->device);
    if (ex->stream) hipStreamSynchronize(ex->stream);
    if (ex->streamB) hipStreamSynchronize(ex->streamB);
    for (int i = 0; i < FT_OCT_STREAMS; i++)
        if (ex->streamO[i]) hipStreamSynchronize(ex->streamO[i]);
    if (ex->octLayout.prof) {  // FT_OCT_PROFILE=1: phase times of the octree kernel (slot 0 of every launch)
        unsigned long long h[FT_MAX_LEVELS * 8];
        if (hipMemcpy(h, ex->octLayout.prof, sizeof h, hipMemcpyDeviceToHost) == hipSuccess)
            for (int l = 0; l < ex->nlevels; l++)
                fprintf(stderr, "[oct profile] level %d: codes %.1f us  sort %.1f us  rounds %.1f us (std::sort replay %.1f us, %llu elems in %llu sorts)  pick %.1f us\n",
                        l, h[l * 8] * 0.01, h[l * 8 + 1] * 0.01, h[l * 8 + 2] * 0.01, h[l * 8 + 4] * 0.01, h[l * 8 + 5], h[l * 8 + 6], h[l * 8 + 3] * 0.01);
        hipFree(ex->octLayout.prof);
        ex->octLayout.prof = nullptr;
    }
    ex->evt.destroy();
    for (int i = 0; i < FT_PIPE_MAX; i++) {
        if (ex->evA[i]) hipEventDestroy(ex->evA[i]);
        if (ex->evB[i]) hipEventDestroy(ex->evB[i]);
        if (ex->evO[i]) hipEventDestroy(ex->evO[i]);
    }
    if (ex->evJoin) hipEventDestroy(ex->evJoin);
    if (ex->evUp) hipEventDestroy(ex->evUp);
    if (ex->graphExec) hipGraphExecDestroy(ex->graphExec);
    for (int i = 0; i < FT_OCT_STREAMS; i++)
        if (ex->streamO[i]) {
            hipStreamSynchronize(ex->streamO[i]);
            if (ex->ownStreams) hipStreamDestroy(ex->streamO[i]);
        }
    if (ex->streamB) {
        hipStreamSynchronize(ex->streamB);
        if (ex->ownStreams) hipStreamDestroy(ex->streamB);
    }
    hipFree(ex->d_pyr);
    if (ex->h_stage) hipHostFree(ex->h_stage);
    if (ex->h_srcTab) hipHostFree(ex->h_srcTab);
    hipFree(ex->d_taps);
    hipFree(ex->d_cellTab);
    hipFree(ex->d_cellCount);
    hipFree(ex->d_stage);
    hipFree((void *)ex->d_l0);
    hipFree(ex->d_sel);
    hipFree(ex->d_nSel);
    hipFree(ex->d_selCount);
    hipFree(ex->d_overflow);
    hipFree(ex->d_ovSlot);
    hipFree(ex->d_bigCount);
    hipHostFree(ex->h_bigStat);
    hipHostFree(ex->h_histStat);
    hipFree(ex->d_bigList);
    hipFree(ex->d_sortList);
    hipFree(ex->d_octLow);
    if (ex->h_repCand) hipHostFree(ex->h_repCand);
    hipFree(ex->d_candDev);
    hipFree(ex->d_candCountDev);
    hipHostFree(ex->h_selCount);
    hipHostFree(ex->h_overflow);
    hipFree(ex->d_keys);
    hipFree(ex->d_desc);
    hipFree(ex->d_stKeys);
    hipFree(ex->d_stDesc);
    hipFree(ex->d_stOut);
    hipFree(ex->d_stInt);
    hipFree(ex->d_stSorted);
    hipHostFree((void *)ex->h_l0);
    hipHostFree(ex->h_cand);
    hipHostFree(ex->h_candCount);
    hipHostFree(ex->h_sel);
    hipHostFree(ex->h_nSel);
    hipHostFree(ex->h_nMono);
    hipHostFree(ex->h_keys);
    hipHostFree(ex->h_desc);
    if (ex->stream) {
        hipStreamSynchronize(ex->stream);
        if (ex->ownStreams) hipStreamDestroy(ex->stream);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// stages.  A batch is cut into sub-batches of consecutive slots so that the host octree of one
// sub-batch overlaps with the kernels of the next (stage A on ex->stream, stage B on ex->streamB).
// ------------------------------------------------------------------------------------------------
int ft_pipeline_depth(const ft_tuning &t, int batch, bool deviceOctree) {
    // Host octree: sub-batches of ~16 images keep every launch large enough to fill the 256 CUs while the host
    // octree of one sub-batch hides behind the kernels of the next.  Device octree: nothing waits for the host,
    // and fewer, wider launches win (measured on MI355X, 256 pairs of 1280x720: depth 8 46.9k fps, depth 4
    // 54.3k, depth 3 56.2k, depth 2 57.5k, depth 1 57.9k; 512 pairs: depth 4 58.0k, depth 2 58.5k) - sub-batches
    // only keep thin kernels (octree, copies) of one sub-batch under the wide kernels of the next; with all
    // kernels placing an image on one XCD the best shape is two sub-batches whatever the batch (512 pairs per
    // step: 65.0k, 384: 64.2k, 256: 63.7k, 768: 63.7k, 1024: 62.9k).  FT_PIPELINE_DEPTH overrides (1 = no pipelining).
    const int envDepth = t.pipeline_depth > 0 ? std::min(FT_PIPE_MAX, t.pipeline_depth) : 0;
    if (envDepth) return std::max(1, std::min(envDepth, batch));
    if (deviceOctree) return batch >= 256 ? 2 : 1;  // two sub-batches: 512 pairs 65.0k fps (3: 61.4k, 4: 60.1k, 1: 62.6k)
    return std::max(1, std::min(FT_PIPE_MAX, batch / 16));
}

// The next batch overwrites the keypoint / descriptor slots: behind whoever still reads them.  The event belongs to the reader (a
// tracked batch); it is waited for on the stage-A stream, which every other stream of the extractor follows through events.
int ft_extract_foreign_wait(ft_extractor *ex) {
    if (!ex->foreignReader) return FT_OK;
    hipEvent_t ev = ex->foreignReader;
    ex->foreignReader = nullptr;
    FT_HIP(hipStreamWaitEvent(ex->stream, ev, 0));
    return FT_OK;
}

int ft_extract_prepare(ft_extractor *ex, const uint8_t *const *images, int batch, int on_device, int width,
                       int height, int stride) {
    FT_REQUIRE(ex && images, "extract: null handle or image list");
    FT_REQUIRE(batch >= 1 && batch <= ex->maxBatch, "extract: batch outside [1, max_batch]");
    FT_REQUIRE(width == ex->width && height == ex->height, "extract: image size differs from the extractor's");
    FT_REQUIRE(stride >= width, "extract: stride smaller than width");
    int rc = ft_set_device(ex->ctx);
    if (rc != FT_OK) return rc;
    const FtGeom &g = ex->geom;
    const bool uploadKernel = ex->tune.upload_kernel != 0;
    if (!on_device && ex->stageHost && uploadKernel) {
        // graph path (latency mode): the host copies the frames into pinned staging, one (captured) kernel moves them
        // into the slot pyramids and writes the level-0 pointer table (ft_launch_upload)
        for (int b = 0; b < batch; b++) {
            if (!images[b]) {
                ft_set_error("extract: empty image");
                return FT_ERR_EMPTY;
            }
            ex->h_l0[b] = ex->d_pyr + (size_t)b * g.pyrPerSlot + g.lv[0].off;
        }
        ft_extract_restage(ex, images, batch, width, height, stride);
        ex->l0External = false;
        ex->l0Aligned = true;
        ex->l0pitch = g.lv[0].pitch;
        ex->lastBatch = batch;
        return ft_launch_upload(ex->stream, batch, ex->h_srcTab, width, height, ex->d_pyr + g.lv[0].off, g.lv[0].pitch,
                                g.pyrPerSlot, ex->d_l0);
    }
    // host frames laid out at a constant distance from each other (a camera ring buffer, a decoded clip): the whole batch
    // goes up as ONE strided copy - rows of the copy = frames - instead of one DMA operation per frame
    // Uploads of host frames outside a captured graph go through the context's upload stream, which carries nothing but
    // copies: the extractors' own streams are lanes they share with other extractors (ft_host.h), and a 0.5 GB copy in front
    // of another front end's kernels would hold those up.  (The previous batch of this extractor is complete - the caller has
    // waited for it - so nothing still reads the slots the copies overwrite.)
    hipStream_t us = ex->stream;
    if (!on_device && !ex->stageHost) {
        rc = ft_context_upload_stream(ex->ctx, &us);
        if (rc != FT_OK) return rc;
    }
    bool oneCopy = false;
    if (!on_device && !ex->stageHost && batch >= 4 && stride == width && g.lv[0].pitch == width) {
        oneCopy = images[0] != nullptr;
        const ptrdiff_t delta = oneCopy && images[1] ? images[1] - images[0] : 0;
        if (delta < (ptrdiff_t)width * height) oneCopy = false;
        for (int b = 1; b < batch && oneCopy; b++)
            if (!images[b] || images[b] - images[b - 1] != delta) oneCopy = false;
        // the distance between frames is the copy's source pitch: beyond what the device takes as a pitch (or when the
        // runtime refuses the copy for any other reason) the frames go up one by one, as frames at irregular distances do
        if (oneCopy && ex->ctx->maxPitch > 0 && (size_t)delta > ex->ctx->maxPitch) oneCopy = false;
        if (oneCopy) {
            const hipError_t ce = hipMemcpy2DAsync(ex->d_pyr + g.lv[0].off, g.pyrPerSlot, images[0], (size_t)delta, (size_t)width * height,
                                                   batch, hipMemcpyHostToDevice, us);
            if (ce != hipSuccess) {
                (void)hipGetLastError();
                oneCopy = false;
            }
        }
    }
    for (int b = 0; b < batch; b++) {
        if (!images[b]) {
            ft_set_error("extract: empty image");
            return FT_ERR_EMPTY;
        }
        uint8_t *slot0 = ex->d_pyr + (size_t)b * g.pyrPerSlot + g.lv[0].off;
        if (oneCopy) {
            ex->h_l0[b] = slot0;
        } else if (on_device) {
            ex->h_l0[b] = images[b];
        } else if (ex->stageHost) {
            // graph path: the frame is copied into pinned staging by the host, the (captured) upload reads from there
            ex->h_l0[b] = slot0;
            uint8_t *stg = ex->h_stage + (size_t)b * width * height;
            for (int y = 0; y < height; y++) memcpy(stg + (size_t)y * width, images[b] + (size_t)y * stride, width);
            FT_HIP(hipMemcpy2DAsync(slot0, g.lv[0].pitch, stg, width, width, height, hipMemcpyHostToDevice, ex->stream));
        } else {
            ex->h_l0[b] = slot0;
            FT_HIP(hipMemcpy2DAsync(slot0, g.lv[0].pitch, images[b], stride, width, height, hipMemcpyHostToDevice, us));
        }
    }
    if (us != ex->stream) {
        FT_HIP(hipEventRecord(ex->evUp, us));
        FT_HIP(hipStreamWaitEvent(ex->stream, ex->evUp, 0));
    }
    ex->l0External = on_device != 0;
    // dword tile loads need 4-byte aligned rows; the slot pyramids always are, caller frames may not be
    ex->l0Aligned = true;
    if (on_device) {
        if (stride & 3) ex->l0Aligned = false;
        for (int b = 0; b < batch; b++)
            if ((uintptr_t)images[b] & 3) ex->l0Aligned = false;
    }
    ex->l0pitch = on_device ? stride : g.lv[0].pitch;
    ex->lastBatch = batch;
    FT_HIP(hipMemcpyAsync((void *)ex->d_l0, (const void *)ex->h_l0, sizeof(uint8_t *) * batch, hipMemcpyHostToDevice,
                          ex->stream));
    return FT_OK;
}

// graph path with host frames: pinned staging for up to 8 slots (allocated once)
int ft_extract_ensure_stage(ft_extractor *ex) {
    if (!ex->h_srcTab) FT_HIP(hipHostMalloc((void **)&ex->h_srcTab, sizeof(FtSrcEntry) * 16, hipHostMallocDefault));
    // (up to 8 frames per camera; a paired stereo batch brings both cameras through one extractor)
    if (!ex->h_stage)
        FT_HIP(hipHostMalloc((void **)&ex->h_stage, (size_t)std::min(ex->maxBatch, 16) * ex->width * ex->height, hipHostMallocDefault));
    return FT_OK;
}
// replay of a captured batch with host frames: refresh the staging copies the captured uploads read
void ft_extract_restage(ft_extractor *ex, const uint8_t *const *images, int batch, int width, int height, int stride) {
    const bool uploadKernel = ex->tune.upload_kernel != 0;
    for (int b = 0; b < batch; b++) {
        // the upload kernel reads the caller's pinned frame in place - only when its whole extent is pinned
        if (uploadKernel && ft_is_pinned_host_range(images[b], (size_t)(height - 1) * stride + width)) {
            ex->h_srcTab[b].ptr = images[b];
            ex->h_srcTab[b].stride = stride;
            continue;
        }
        uint8_t *stg = ex->h_stage + (size_t)b * width * height;
        for (int y = 0; y < height; y++) memcpy(stg + (size_t)y * width, images[b] + (size_t)y * stride, width);
        ex->h_srcTab[b].ptr = stg;
        ex->h_srcTab[b].stride = width;
    }
}

// The host-mapped pinned candidate lists the host octree reads (2.7 MB per 1280x720 slot: 1.4 GB for 512 slots) exist only
// once the host octree is used - an extractor that stays on the device octree never allocates them.
int ft_extract_ensure_host_cand(ft_extractor *ex) {
    if (ex->h_cand) return FT_OK;
    const size_t n = std::max<size_t>((size_t)ex->maxBatch * ex->geom.candPerSlot, 1);
    FT_HIP(hipHostMalloc((void **)&ex->h_cand, n * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
    void *dp = nullptr;
    FT_HIP(hipHostGetDevicePointer(&dp, ex->h_cand, 0));
    ex->d_cand = (uint32_t *)dp;
    return FT_OK;
}

// stage A of slots [b0, b0+nb): pyramid + FAST + ordered compaction on ex->stream; `done` is recorded after it
int ft_extract_launch_a(ft_extractor *ex, int b0, int nb, hipEvent_t done) {
    const FtGeom &g = ex->geom;
    if (!ex->deviceOctree) {
        const int rce = ft_extract_ensure_host_cand(ex);
        if (rce != FT_OK) return rce;
    }
    const bool tm = ex->ctx->kernelTiming;
    const int al = ex->l0Aligned ? 1 : 0;
    const uint8_t *const *l0 = ex->d_l0 + b0;
    uint8_t *pyr = ex->d_pyr + (size_t)b0 * g.pyrPerSlot;
    int *cellCount = ex->d_cellCount + (size_t)b0 * g.totalCells;
    uint32_t *stage = ex->d_stage + (size_t)b0 * g.stagePerSlot;
    ex->evt.begin(tm, "kernel.pyr_down(all levels)", ex->stream);
    int rc = ft_launch_pyramid(ex->stream, g, nb, l0, ex->l0pitch, pyr, ex->d_taps, al, ex->tune.pyr_rows);
    ex->evt.end(tm, ex->stream);
    if (rc != FT_OK) return rc;
    ex->evt.begin(tm, "kernel.fast_cells", ex->stream);
    // the device octree ranks candidates by their coordinates: FAST need not deliver them in the reference's order; the host
    // octree gets the ordered per-cell kernel
    rc = ft_launch_fast_cells(ex->stream, g, nb, l0, ex->l0pitch, pyr, ex->iniTh, ex->minTh, al, cellCount, stage,
                              ex->deviceOctree ? 0 : 1, ex->d_cellTab);
    ex->evt.end(tm, ex->stream);
    if (rc != FT_OK) return rc;
    ex->evt.begin(tm, "kernel.compact", ex->stream);
    uint32_t *candDst = ex->deviceOctree ? ex->d_candDev : ex->d_cand;
    int *cntDst = ex->deviceOctree ? ex->d_candCountDev : ex->d_candCount;
    rc = ft_launch_compact(ex->stream, g, nb, cellCount, stage, candDst + (size_t)b0 * g.candPerSlot,
                           cntDst + (size_t)b0 * g.nlevels);
    ex->evt.end(tm, ex->stream);
    if (rc != FT_OK) return rc;
    if (done) FT_HIP(hipEventRecord(done, ex->stream));
    return FT_OK;
}

// device octree of slots [b0, b0+nb) of sub-batch `sub`: waits for their stage A (evA[sub], recorded here on
// ex->stream), runs on one of the octree streams and records `done`
int ft_extract_launch_octree(ft_extractor *ex, int sub, int b0, int nb, hipEvent_t done) {
    const FtGeom &g = ex->geom;
    hipStream_t so = ex->streamO[sub % FT_OCT_STREAMS];
    FT_HIP(hipEventRecord(ex->evO[sub], ex->stream));
    FT_HIP(hipStreamWaitEvent(so, ex->evO[sub], 0));
    FtOctArgs a = ex->octLayout;
    a.cand = ex->d_candDev + (size_t)b0 * g.candPerSlot;
    a.candCount = ex->d_candCountDev + (size_t)b0 * g.nlevels;
    a.sel = ex->d_sel + (size_t)b0 * g.maxKp;
    a.selCount = ex->d_selCount + (size_t)b0 * g.nlevels;
    a.overflow = ex->d_overflow;
    a.ovSlot = ex->d_ovSlot + b0;
    a.low = ex->d_octLow ? ex->d_octLow + (size_t)b0 * g.nlevels * FT_OCT_MAXN : nullptr;
    // Levels with more than FT_OCT_MAXN candidates go to the histogram tier.  A kernel more behind k_octree costs the
    // headline workload 1.2 % even when its 16 workgroups find nothing to do (A/B, 3 x 96 steps: 74.1 against 75.0 k frames/s:
    // one more launch in the chain of the octree lane), so the tier follows the demand: launched for the first batches of an
    // extractor and while the frames ask for it, with a grid sized by what the previous batches listed (its workgroups walk
    // the list), retired after FT_OCT_HIST_RETIRE batches without a listed level.  A dense frame that arrives while it is
    // retired is repaired on the host - that frame only, once - and brings the tier back for the next batch (the demand is
    // counted either way).  Small (latency-mode) batches: since a frame needed it (a kernel node more in the captured graph).
    const bool large = nb > 16 || ex->lastBatch > 16;
    const bool histNow = ex->histEnabled && (large ? ex->histGrid > 0 : ex->histOn);
    a.histCap = histNow ? nb * g.nlevels : 0;
    a.histGrid = large ? std::min(ex->histGrid, nb * g.nlevels) : nb * g.nlevels;
    a.histWanted = ex->histEnabled && !histNow ? 1 : 0;
    // launches of a frame or two: the histogram formulation for every level (no sort: the last level is done sooner)
    a.histFirst = ex->histEnabled && (ex->histFirstMode == 2 || (ex->histFirstMode == 1 && !large)) ? 1 : 0;
    if (a.histFirst) a.histCap = a.histGrid = a.histWanted = 0;  // no list: a workgroup per (level, image)
    a.sortCap = a.bigN ? std::min(ex->bigGrid, nb * g.nlevels) : 0;
    // one set of lists and counters per octree stream: the launches of two sub-batches run side by side
    a.bigCount = ex->d_bigCount + 4 * (sub % FT_OCT_STREAMS);
    a.bigList = ex->d_bigList + (size_t)(sub % FT_OCT_STREAMS) * ex->maxBatch * g.nlevels;
    a.sortList = ex->d_sortList + (size_t)(sub % FT_OCT_STREAMS) * ex->maxBatch * g.nlevels;
    const bool tm = ex->ctx->kernelTiming;
    ex->evt.begin(tm, "kernel.octree", so);
    int rc = ft_launch_octree(so, g, nb, a);
    ex->evt.end(tm, so);
    if (rc != FT_OK) return rc;
    if (done) FT_HIP(hipEventRecord(done, so));
    return FT_OK;
}

// host octree over the candidates of slots [b0, b0+nb) of one or two extractors (their stage A must
// have completed).  One task per (camera, level, image), issued level-major so the big level-0 trees
// start first; results land in disjoint sub-ranges of a preallocated scratch and are packed per slot.
int ft_extract_octree_multi(ft_extractor *const *exs, int nex, int b0, int nb) {
    ft_extractor *e0 = exs[0];
    const FtGeom &g = e0->geom;
    const int L = g.nlevels;
    const int perEx = nb * L;
    auto task = [&](int t, int) {
        static thread_local ft::OctreeWorkspace ws;
        static thread_local std::vector<int> keep;
        ft_extractor *ex = exs[t / perEx];
        const int r = t % perEx;
        const int level = r / nb, rel = r - level * nb, slot = b0 + rel;
        const FtLevelGeom &v = g.lv[level];
        const int n = ex->h_candCount[slot * L + level];
        const uint32_t *cand = ex->h_cand + (size_t)slot * g.candPerSlot + v.candBase;
        keep.clear();
        const int minB = FT_EDGE_THRESHOLD - 3;
        int k = ft::distribute_octree(cand, n, minB, v.maxBX, minB, v.maxBY, ex->quota[level], ws, keep);
        k = std::min(k, ex->levelMax[level]);
        FtSelKp *dst = ex->h_sel + (size_t)slot * g.maxKp + ex->levelOff[level];
        for (int i = 0; i < k; i++) {
            const uint32_t c = cand[keep[i]];
            // ORBextractor.cc:1211-1217: add the border offset back
            dst[i].x = (short)((c & 0xfffu) + minB);
            dst[i].y = (short)(((c >> 12) & 0xfffu) + minB);
            dst[i].level = (short)level;
            dst[i].response = (short)(c >> 24);
        }
        ex->h_selCount[(size_t)slot * L + level] = k;
    };
    e0->ctx->pool->parallel_for(nex * perEx, task);
    for (int e = 0; e < nex; e++) {
        ft_extractor *ex = exs[e];
        for (int slot = b0; slot < b0 + nb; slot++) {
            int n = 0;
            for (int l = 0; l < L; l++) n += ex->h_selCount[(size_t)slot * L + l];
            ex->h_nSel[slot] = n;
        }
    }
    return FT_OK;
}

int ft_extract_octree(ft_extractor *ex, int b0, int nb) { return ft_extract_octree_multi(&ex, 1, b0, nb); }

// After a batch has drained: size the sorted big tier (k_octree_big) of the following batches by what the histogram tier
// handed on in this one, and retire it after a run of batches without demand.
void ft_extract_update_big_grid(ft_extractor *ex) {
    if (!ex->octLayout.bigN && !ex->histEnabled) return;
    if (ex->lastBatch <= 16) {
        // latency mode (results delivered by one kernel, no demand counter on the host): an overflow of a frame switches the
        // histogram tier on for the frames that follow, one more the sorted big tier (the captured graph is re-captured: its
        // key holds both)
        if (ex->h_overflow[0]) {
            if (ex->histEnabled && ex->histFirstMode == 0 && !ex->histOn) ex->histOn = true;
            else if (ex->octLayout.bigN && ex->bigGrid == 0) ex->bigGrid = ex->maxBatch * ex->nlevels;
        }
        return;
    }
    int want = 0, wantHist = 0;  // the octree streams keep a list each (launches on different streams run side by side)
    for (int k = 0; k < FT_OCT_STREAMS; k++) {
        want = std::max(want, ex->h_bigStat[k]);
        wantHist = std::max(wantHist, ex->h_histStat[k]);
        ex->h_bigStat[k] = ex->h_histStat[k] = 0;
    }
    if (wantHist > 0) {  // a workgroup per listed level up to FT_OCT_HISTMAX (beyond that they walk the list)
        ex->histIdle = 0;
        ex->histGrid = std::max(std::max(ex->histGrid, FT_OCT_HISTMIN), std::min(FT_OCT_HISTMAX, ((wantHist + 63) / 64) * 64));
    } else if (ex->histGrid > 0 && ++ex->histIdle >= FT_OCT_HIST_RETIRE) {
        ex->histGrid = 0;  // retired until a frame asks for it again
    }
    if (!ex->octLayout.bigN) return;
    if (want > 0) {
        ex->bigIdle = 0;
        const int need = std::max(FT_OCT_BIGMIN, ((want + want / 4 + 63) / 64) * 64);
        if (need > ex->bigGrid) ex->bigGrid = std::min(need, ex->maxBatch * ex->nlevels);
    } else if (ex->bigGrid > 0 && ++ex->bigIdle >= 16) {
        ex->bigGrid = 0;
    }
}

// Which slots of the last device-octree batch need the host octree (a level beyond the kernels' limits).  Called after
// the batch has drained and the summary flag was seen; clears the device flags.
int ft_extract_overflow_slots(ft_extractor *ex, int batch, std::vector<int> &slots) {
    std::vector<int> f(batch, 0);
    FT_HIP(hipMemcpy(f.data(), ex->d_ovSlot, sizeof(int) * batch, hipMemcpyDeviceToHost));
    for (int b = 0; b < batch; b++)
        if (f[b]) slots.push_back(b);
    FT_HIP(hipMemset(ex->d_ovSlot, 0, sizeof(int) * batch));
    FT_HIP(hipMemset(ex->d_overflow, 0, sizeof(int)));
    ex->h_overflow[0] = 0;
    return FT_OK;
}

// Redo the octree of single slots on the host, in place (jobs = (extractor, slot) pairs; the extractors share geometry and
// context): the candidate lists of those slots come back from the device - in the order the device stage delivered them;
// they are put into the reference's emission order (cell row, cell column, row-major inside the cell), which the host
// octree's result depends on - and all (job, level) trees run as one parallel job.  ft_extract_repair_launch then uploads
// a slot's selection and enqueues its orientation + descriptors.  The other slots keep their device results.
int ft_extract_repair_prepare(const std::vector<std::pair<ft_extractor *, int>> &jobs) {
    if (jobs.empty()) return FT_OK;
    ft_extractor *e0 = jobs[0].first;
    const FtGeom &g = e0->geom;
    const int L = g.nlevels;
    // The candidate lists of the slots under repair come back into a pinned buffer of the extractor that holds just those
    // slots (grow-only, indexed by job) - not into the host-octree pipeline's array for ALL slots (max_batch x candPerSlot:
    // 1.4 GB of pinned memory for 512 slots, whose allocation in the middle of the pipeline stalled the other front end's
    // batch and turned a recoverable overflow into a hard error when it failed).
    std::vector<int> jobIdx(jobs.size());
    {
        std::map<ft_extractor *, int> perEx;
        for (size_t j = 0; j < jobs.size(); j++) jobIdx[j] = perEx[jobs[j].first]++;
        for (auto &kv : perEx) {
            ft_extractor *ex = kv.first;
            if (kv.second > ex->repCap) {
                if (ex->h_repCand) hipHostFree(ex->h_repCand);
                ex->h_repCand = nullptr;
                ex->repCap = 0;
                const int want = std::min(ex->maxBatch, std::max(kv.second + kv.second / 2, 4));
                FT_HIP(hipHostMalloc((void **)&ex->h_repCand, (size_t)want * g.candPerSlot * sizeof(uint32_t), hipHostMallocDefault));
                ex->repCap = want;
            }
        }
    }
    std::vector<int> counts(jobs.size() * L, 0);
    for (size_t j = 0; j < jobs.size(); j++) {
        ft_extractor *ex = jobs[j].first;
        const int slot = jobs[j].second;
        FT_HIP(hipMemcpy(counts.data() + j * L, ex->d_candCountDev + (size_t)slot * L, sizeof(int) * L, hipMemcpyDeviceToHost));
        for (int l = 0; l < L; l++) {
            const FtLevelGeom &v = g.lv[l];
            int &n = counts[j * L + l];
            n = std::max(0, std::min(n, v.candCap));
            if (n > 0)
                FT_HIP(hipMemcpyAsync(ex->h_repCand + (size_t)jobIdx[j] * g.candPerSlot + v.candBase,
                                      ex->d_candDev + (size_t)slot * g.candPerSlot + v.candBase, sizeof(uint32_t) * n,
                                      hipMemcpyDeviceToHost, ex->streamB));
        }
    }
    for (auto &j : jobs) FT_HIP(hipStreamSynchronize(j.first->streamB));
    auto task = [&](int t, int) {
        static thread_local ft::OctreeWorkspace ws;
        static thread_local std::vector<int> keep;
        const int j = t / L, level = t % L;
        ft_extractor *ex = jobs[j].first;
        const int slot = jobs[j].second;
        const FtLevelGeom &v = g.lv[level];
        const int n = counts[(size_t)j * L + level];
        uint32_t *cand = ex->h_repCand + (size_t)jobIdx[j] * g.candPerSlot + v.candBase;
        auto rank = [&](uint32_t q) -> uint64_t {
            const int x = (int)(q & 0xfffu) - 3, y = (int)((q >> 12) & 0xfffu) - 3;
            const int cj = std::min(x / std::max(v.wCell, 1), std::max(v.nCols - 1, 0));
            const int ci = std::min(y / std::max(v.hCell, 1), std::max(v.nRows - 1, 0));
            return ((uint64_t)ci << 48) | ((uint64_t)cj << 32) | ((uint64_t)y << 16) | (uint64_t)x;
        };
        std::sort(cand, cand + n, [&](uint32_t a, uint32_t b) { return rank(a) < rank(b); });
        keep.clear();
        const int minB = FT_EDGE_THRESHOLD - 3;
        int k = ft::distribute_octree(cand, n, minB, v.maxBX, minB, v.maxBY, ex->quota[level], ws, keep);
        k = std::min(k, ex->levelMax[level]);
        FtSelKp *dst = ex->h_sel + (size_t)slot * g.maxKp + ex->levelOff[level];
        for (int i = 0; i < k; i++) {
            const uint32_t c = cand[keep[i]];
            dst[i].x = (short)((c & 0xfffu) + minB);  // ORBextractor.cc:1211-1217: add the border offset back
            dst[i].y = (short)(((c >> 12) & 0xfffu) + minB);
            dst[i].level = (short)level;
            dst[i].response = (short)(c >> 24);
        }
        ex->h_selCount[(size_t)slot * L + level] = k;
    };
    e0->ctx->pool->parallel_for((int)jobs.size() * L, task);
    for (auto &j : jobs) {
        int n = 0;
        for (int l = 0; l < L; l++) n += j.first->h_selCount[(size_t)j.second * L + l];
        j.first->h_nSel[j.second] = n;
        j.first->ctx->addStat("extract.host_octree_repairs", 0);
    }
    return FT_OK;
}

int ft_extract_repair_launch(ft_extractor *ex, int slot, hipStream_t st) {
    const bool dev = ex->deviceOctree;
    ex->deviceOctree = false;  // stage B uploads the host selection of the slot
    const int rc = ft_extract_launch_b(ex, slot, 1, st);
    ex->deviceOctree = dev;
    return rc;
}

static int rangeMaxN(const ft_extractor *ex, int b0, int nb) {
    if (ex->deviceOctree) return ex->geom.maxKp;  // totals are not known on the host yet: copy whole rows
    int maxN = 0;
    for (int b = b0; b < b0 + nb; b++) maxN = std::max(maxN, ex->h_nSel[b]);
    return maxN;
}

// stage B of slots [b0, b0+nb) on `st`: (host octree only) upload the per-level selection and its counts,
// then orientation + descriptors -> d_keys / d_desc packed in level order, per-image totals -> d_nSel
int ft_extract_launch_b(ft_extractor *ex, int b0, int nb, hipStream_t st) {
    const FtGeom &g = ex->geom;
    if (!ex->deviceOctree) {
        FT_HIP(hipMemcpyAsync(ex->d_selCount + (size_t)b0 * g.nlevels, ex->h_selCount + (size_t)b0 * g.nlevels,
                              sizeof(int) * nb * g.nlevels, hipMemcpyHostToDevice, st));
        FT_HIP(hipMemcpyAsync(ex->d_sel + (size_t)b0 * g.maxKp, ex->h_sel + (size_t)b0 * g.maxKp,
                              sizeof(FtSelKp) * (size_t)nb * g.maxKp, hipMemcpyHostToDevice, st));
    }
    const bool tm = ex->ctx->kernelTiming;
    ex->evt.begin(tm, "kernel.orient_desc", st);
    int rc = ft_launch_orient_desc(st, g, nb, ex->d_l0 + b0, ex->l0pitch, ex->d_pyr + (size_t)b0 * g.pyrPerSlot,
                                   ex->l0Aligned ? 1 : 0, ex->d_sel + (size_t)b0 * g.maxKp,
                                   ex->d_selCount + (size_t)b0 * g.nlevels, ex->octLayout, ex->d_nSel + b0,
                                   ex->d_keys + (size_t)b0 * g.maxKp, ex->d_desc + (size_t)b0 * g.maxKp * 32);
    ex->evt.end(tm, st);
    return rc;
}

// device octree: the host learns the per-image totals (and whether a level exceeded the kernel's limits)
// only at the end of the batch
int ft_extract_finish_counts(ft_extractor *ex, int batch, hipStream_t st) {
    if (!ex->deviceOctree) return FT_OK;
    FtCountsArgs a;
    a.nSel = ex->d_nSel;
    a.oNSel = ex->h_nSel;
    a.overflow = ex->d_overflow;
    a.oOverflow = ex->h_overflow;
    // demand for the tiers behind k_octree during this batch (ft_extract_update_big_grid)
    a.bigCount = (ex->octLayout.bigN || ex->histEnabled) ? ex->d_bigCount : nullptr;
    a.oHist = ex->h_histStat;
    a.oBig = ex->h_bigStat;
    a.batch = batch;
    a.nStreams = FT_OCT_STREAMS;
    return ft_launch_finish_counts(st, a);
}

// async D2H of the keypoints / descriptors of slots [b0, b0+nb) on `st`
int ft_extract_download(ft_extractor *ex, int b0, int nb, hipStream_t st) {
    const FtGeom &g = ex->geom;
    const int maxN = rangeMaxN(ex, b0, nb);
    if (maxN == 0) return FT_OK;
    const size_t o = (size_t)b0 * g.maxKp;
    FT_HIP(hipMemcpy2DAsync(ex->h_keys + o, sizeof(ft_keypoint) * g.maxKp, ex->d_keys + o, sizeof(ft_keypoint) * g.maxKp,
                            sizeof(ft_keypoint) * maxN, nb, hipMemcpyDeviceToHost, st));
    FT_HIP(hipMemcpy2DAsync(ex->h_desc + o * 32, (size_t)32 * g.maxKp, ex->d_desc + o * 32, (size_t)32 * g.maxKp,
                            (size_t)32 * maxN, nb, hipMemcpyDeviceToHost, st));
    return FT_OK;
}

// ORBextractor.cc:1466-1487: keypoints inside the lapping area fill from the back
static int assembleOutputs(ft_extractor *ex, int batch, int lap0, int lap1, ft_keypoint *keypoints,
                           uint8_t *descriptors, int capacity, int *n_keypoints, int *n_mono) {
    const FtGeom &g = ex->geom;
    for (int b = 0; b < batch; b++)
        if (ex->h_nSel[b] > capacity) {
            ft_set_error("extract: output capacity too small (use ft_extractor_max_keypoints)");
            return FT_ERR_CAPACITY;
        }
    // (a wide batch is 10+ MB of keypoints and descriptors: one image per host thread of the context)
    const std::function<void(int, int)> one = [&](int b, int) {
        const int n = ex->h_nSel[b];
        const ft_keypoint *src = ex->h_keys + (size_t)b * g.maxKp;
        const uint8_t *sd = ex->h_desc + (size_t)b * g.maxKp * 32;
        ft_keypoint *dk = keypoints ? keypoints + (size_t)b * capacity : nullptr;
        uint8_t *dd = descriptors ? descriptors + (size_t)b * capacity * 32 : nullptr;
        int monoIndex = 0, stereoIndex = n - 1;
        for (int i = 0; i < n; i++) {
            const ft_keypoint &kp = src[i];
            int dst;
            if (kp.x >= lap0 && kp.x <= lap1) dst = stereoIndex--;
            else dst = monoIndex++;
            if (dk) dk[dst] = kp;
            if (dd) memcpy(dd + (size_t)dst * 32, sd + (size_t)i * 32, 32);
        }
        if (n_keypoints) n_keypoints[b] = n;
        if (n_mono) n_mono[b] = monoIndex;
    };
    if (batch >= 16) ex->ctx->pool->parallel_for(batch, one);
    else
        for (int b = 0; b < batch; b++) one(b, 0);
    return FT_OK;
}

extern "C" {

int ft_extractor_create(ft_context *ctx, int nfeatures, float scale_factor, int nlevels, int ini_th_fast,
                        int min_th_fast, int image_width, int image_height, int max_batch, ft_extractor **out) {
    FT_REQUIRE(ctx && out, "ft_extractor_create: null argument");
    *out = nullptr;
    FT_REQUIRE(nfeatures > 0, "nfeatures must be positive");
    FT_REQUIRE(nlevels >= 1 && nlevels <= FT_MAX_LEVELS, "nlevels outside [1, 12]");
    FT_REQUIRE(scale_factor > 1.0f, "scaleFactor must exceed 1");
    FT_REQUIRE(min_th_fast >= 1 && ini_th_fast >= min_th_fast && ini_th_fast < 255,
               "FAST thresholds must satisfy 1 <= minThFAST <= iniThFAST < 255");
    FT_REQUIRE(image_width >= 2 * FT_EDGE_THRESHOLD + 8 && image_height >= 2 * FT_EDGE_THRESHOLD + 8,
               "image smaller than the 19-px border allows");
    FT_REQUIRE(max_batch >= 1 && max_batch <= 4096, "max_batch outside [1, 4096]");
    int rc = ft_set_device(ctx);
    if (rc != FT_OK) return rc;
    ft_extractor *ex = new ft_extractor();
    ex->ctx = ctx;
    ex->nfeatures = nfeatures;
    ex->scaleFactor = scale_factor;
    ex->nlevels = nlevels;
    ex->iniTh = ini_th_fast;
    ex->minTh = min_th_fast;
    ex->width = image_width;
    ex->height = image_height;
    ex->maxBatch = max_batch;
    ex->tune = ctx->tuning;  // the extractor keeps the switches it was created under
    std::vector<FtTap> taps;
    rc = buildGeometry(ex, taps);
    if (rc != FT_OK) {
        delete ex;
        return rc;
    }
    const FtGeom &g = ex->geom;
    const size_t B = max_batch;
#define FT_TRY(x)            \
    do {                     \
        rc = (x);            \
        if (rc != FT_OK) {   \
            freeAll(ex);     \
            delete ex;       \
            return rc;       \
        }                    \
    } while (0)
    {
        hipStream_t lanes[FT_LANE_STREAMS];
        // latency-mode extractors (graph capture) own their streams; wide ones run on the context's lanes
        FT_TRY(ft_context_take_lanes(ctx, max_batch <= 16, image_width, image_height, lanes, &ex->ownStreams));
        ex->stream = lanes[0];
        ex->streamB = lanes[1];
        for (int i = 0; i < FT_OCT_STREAMS; i++) ex->streamO[i] = lanes[2 + i];
    }
    hipError_t se = hipSuccess;
    for (int i = 0; i < FT_PIPE_MAX && se == hipSuccess; i++) {
        se = hipEventCreateWithFlags(&ex->evA[i], hipEventDisableTiming);
        if (se == hipSuccess) se = hipEventCreateWithFlags(&ex->evB[i], hipEventDisableTiming);
        if (se == hipSuccess) se = hipEventCreateWithFlags(&ex->evO[i], hipEventDisableTiming);
        if (se == hipSuccess && i == 0) se = hipEventCreateWithFlags(&ex->evJoin, hipEventDisableTiming);
        if (se == hipSuccess && i == 0) se = hipEventCreateWithFlags(&ex->evUp, hipEventDisableTiming);
    }
    if (se != hipSuccess) {
        freeAll(ex);
        delete ex;
        return ft_hip_fail(se, "stream / event creation", __FILE__, __LINE__);
    }
    FT_TRY(devAlloc(&ex->d_pyr, B * g.pyrPerSlot));
    FT_TRY(devAlloc(&ex->d_taps, taps.size()));
    FT_TRY(devAlloc(&ex->d_cellTab, (size_t)std::max(g.totalCells, 1)));
    FT_TRY(devAlloc(&ex->d_cellCount, B * g.totalCells));
    FT_TRY(devAlloc(&ex->d_stage, B * g.stagePerSlot));
    FT_TRY(devAlloc(&ex->d_l0, B));
    FT_TRY(devAlloc(&ex->d_sel, B * g.maxKp));
    FT_TRY(devAlloc(&ex->d_nSel, B));
    FT_TRY(devAlloc(&ex->d_selCount, B * g.nlevels + FT_MAX_LEVELS));  // k_orient_desc reads FT_MAX_LEVELS counts per image at once
    FT_TRY(devAlloc(&ex->d_overflow, 1));
    FT_TRY(devAlloc(&ex->d_ovSlot, B));
    FT_TRY(devAlloc(&ex->d_bigCount, 4 * FT_OCT_STREAMS));
    FT_TRY(pinAlloc(&ex->h_bigStat, FT_OCT_STREAMS));
    FT_TRY(pinAlloc(&ex->h_histStat, FT_OCT_STREAMS));
    for (int k = 0; k < FT_OCT_STREAMS; k++) ex->h_bigStat[k] = ex->h_histStat[k] = 0;
    FT_TRY(devAlloc(&ex->d_bigList, (size_t)FT_OCT_STREAMS * B * g.nlevels));
    FT_TRY(devAlloc(&ex->d_sortList, (size_t)FT_OCT_STREAMS * B * g.nlevels));
    FT_TRY(pinAlloc(&ex->h_selCount, B * g.nlevels));
    FT_TRY(pinAlloc(&ex->h_overflow, 1));
    FT_TRY(devAlloc(&ex->d_keys, B * g.maxKp));
    FT_TRY(devAlloc(&ex->d_desc, B * g.maxKp * 32));
    FT_TRY(pinAlloc(&ex->h_l0, B));
    FT_TRY(pinAlloc(&ex->h_candCount, B * g.nlevels, hipHostMallocMapped | hipHostMallocCoherent));
    FT_TRY(pinAlloc(&ex->h_sel, B * g.maxKp));
    FT_TRY(pinAlloc(&ex->h_nSel, B));
    FT_TRY(pinAlloc(&ex->h_nMono, B));
    FT_TRY(pinAlloc(&ex->h_keys, B * g.maxKp));
    FT_TRY(pinAlloc(&ex->h_desc, B * g.maxKp * 32));
    {
        void *dp = nullptr;
        hipError_t e = hipHostGetDevicePointer(&dp, ex->h_candCount, 0);
        ex->d_candCount = (int *)dp;
        if (e != hipSuccess) {
            freeAll(ex);
            delete ex;
            return ft_hip_fail(e, "hipHostGetDevicePointer", __FILE__, __LINE__);
        }
    }
    {
        std::vector<FtCellRec> cellTab(std::max(g.totalCells, 1), FtCellRec{0, 0, 0, 0});
        for (int l = 0; l < nlevels; l++) {
            const FtLevelGeom &v = g.lv[l];
            for (int ci = 0; ci < v.nRows; ci++)
                for (int cj = 0; cj < v.nCols; cj++) {
                    // ORBextractor.cc:1138-1152: the cell's sub-image [iniX, maxX) x [iniY, maxY) and the skip rules;
                    // cv::FAST finds nothing in a sub-image under 7 px
                    const int iniX = 16 + cj * v.wCell, iniY = 16 + ci * v.hCell;
                    const int maxX = std::min(iniX + v.wCell + 6, v.maxBX), maxY = std::min(iniY + v.hCell + 6, v.maxBY);
                    int tw = maxX - iniX, th = maxY - iniY;
                    if (iniX >= v.maxBX - 6 || iniY >= v.maxBY - 3 || tw < 7 || th < 7) tw = th = 0;
                    const int c = ci * v.nCols + cj;
                    FtCellRec &r = cellTab[v.cellBase + c];
                    r.origin = (uint32_t)iniX | ((uint32_t)iniY << 16);
                    r.shape = (uint32_t)tw | ((uint32_t)th << 8) | ((uint32_t)l << 16);
                    r.srcOff = (uint32_t)(v.off + iniY * v.pitch + iniX);
                    r.outOff = (uint32_t)(v.stageBase + c * v.cellCap);
                }
        }
        hipError_t e = hipMemcpy(ex->d_cellTab, cellTab.data(), cellTab.size() * sizeof(FtCellRec), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(ex->d_taps, taps.data(), taps.size() * sizeof(FtTap), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(ex->d_nSel, 0, sizeof(int) * B);
        if (e != hipSuccess) {
            freeAll(ex);
            delete ex;
            return ft_hip_fail(e, "upload resize tables", __FILE__, __LINE__);
        }
    }
    memset(ex->h_nSel, 0, sizeof(int) * B);
    ex->levelOff.assign(nlevels + 1, 0);
    for (int l = 0; l < nlevels; l++) ex->levelOff[l + 1] = ex->levelOff[l] + ex->levelMax[l];
    memset(ex->h_selCount, 0, sizeof(int) * B * nlevels);
    ex->h_overflow[0] = 0;
    {
        FtOctArgs &o = ex->octLayout;
        memset(&o, 0, sizeof o);
        int maxQ = 0, need = 0;
        for (int l = 0; l < nlevels; l++) {
            o.quota[l] = ex->quota[l];
            o.levelMax[l] = ex->levelMax[l];
            o.selOff[l] = ex->levelOff[l];
            maxQ = std::max(maxQ, ex->quota[l]);
            const int minB = FT_EDGE_THRESHOLD - 3;
            int nIni = (int)std::round(static_cast<float>(g.lv[l].maxBX - minB) / (g.lv[l].maxBY - minB));
            if (nIni < 1) nIni = 1;
            need = std::max(need, ex->quota[l] + 4 * nIni);  // live nodes never exceed max(N + 2, 4 * nIni), + 4 in flight
        }
        o.poolCap = need + 16;
        // FT_DEVICE_OCTREE=0 keeps the octree on the host (also the path when a quota exceeds the kernel's limit)
        // the node pools must fit a CU's LDS beside FT_OCT_MAXN keys (per-level quotas up to ~1900); u16 node indices
        (void)maxQ;
        ex->deviceOctree = ex->tune.device_octree != 0 && o.poolCap < 30000 && ft_octree_smem_bytes(o.poolCap) <= 160 * 1024;
        o.histCap = o.sortCap = o.histGrid = o.histWanted = o.histFirst = 0;
        ex->histEnabled = ex->deviceOctree && ex->tune.oct_hist != 0 &&
                          ft_octree_hist_smem_bytes(o.poolCap) <= 160 * 1024;
        ex->histFirstMode = ex->tune.oct_hist_first;
        o.bigN = ex->deviceOctree && ex->tune.oct_big != 0 ? ft_octree_big_keys(o.poolCap) : 0;
        o.bigCount = ex->d_bigCount;
        o.bigList = ex->d_bigList;
        o.sortList = ex->d_sortList;
        o.low = nullptr;  // set per launch (ft_extract_launch_octree)
        if (ex->deviceOctree) {
            if (ft_debug_env("FT_DEBUG_OCT_PROFILE")) {  // per-level phase clocks of k_octree (slot 0) kept on the device
                FT_TRY(devAlloc(&o.prof, (size_t)FT_MAX_LEVELS * 8));
                hipMemset(o.prof, 0, sizeof(unsigned long long) * FT_MAX_LEVELS * 8);
            }
            FT_TRY(devAlloc(&ex->d_candDev, B * g.candPerSlot));
            FT_TRY(devAlloc(&ex->d_candCountDev, B * g.nlevels));
            // the first sorted tier in the compact LDS layout (49 instead of 64 KB per workgroup) wherever it fits
            if (ft_octree_smem_bytes(o.poolCap, true) < ft_octree_smem_bytes(o.poolCap, false))
                FT_TRY(devAlloc(&ex->d_octLow, B * g.nlevels * (size_t)FT_OCT_MAXN));
            hipError_t me = hipMemset(ex->d_overflow, 0, sizeof(int));
            if (me == hipSuccess) me = hipMemset(ex->d_ovSlot, 0, sizeof(int) * B);
            if (me == hipSuccess) me = hipMemset(ex->d_bigCount, 0, 4 * FT_OCT_STREAMS * sizeof(int));
            if (me != hipSuccess) {
                freeAll(ex);
                delete ex;
                return ft_hip_fail(me, "hipMemset", __FILE__, __LINE__);
            }
        }
    }
#undef FT_TRY
    ctx->liveObjects++;
    *out = ex;
    return FT_OK;
}

int ft_octree_distribute(const int *xys, int n, int minX, int maxX, int minY, int maxY, int N, int *out_idx,
                         int capacity, int *n_out) {
    FT_REQUIRE(n >= 0 && (n == 0 || xys) && n_out, "ft_octree_distribute: bad argument");
    FT_REQUIRE(maxX > minX && maxY > minY && N >= 0, "ft_octree_distribute: empty region");
    std::vector<uint32_t> packed(n);
    for (int i = 0; i < n; i++) {
        FT_REQUIRE(xys[3 * i] >= 0 && xys[3 * i] < 4096 && xys[3 * i + 1] >= 0 && xys[3 * i + 1] < 4096 &&
                       xys[3 * i + 2] >= 0 && xys[3 * i + 2] < 256,
                   "ft_octree_distribute: candidate out of range");
        packed[i] = ft_pack_cand(xys[3 * i], xys[3 * i + 1], xys[3 * i + 2]);
    }
    ft::OctreeWorkspace ws;
    std::vector<int> keep;
    // FT_DEBUG_OCTREE_PATHS=1 / 2 / 3 route this host entry point through the path-code formulations the device kernels
    // are built from (1: node list replay, 2: round formulation over sorted keys, 3: over a histogram), so they can be
    // checked without a GPU
    auto dbgInt = [](const char *name, int def) {
        const char *e = ft_debug_env(name);
        return e && *e ? atoi(e) : def;
    };
    const int usePaths = dbgInt("FT_DEBUG_OCTREE_PATHS", 0);
    int k;
    if (usePaths == 3 && n < 65535) {
        // 3: the histogram formulation of k_octree_hist (FT_DEBUG_OCTREE_HIST_BINS bins, default 8192); a level it gives up
        // on goes to the sorted rounds, as on the device.  FT_DEBUG_OCTREE_HIST_STRICT=1 reports the give-up instead.
        const int bins = dbgInt("FT_DEBUG_OCTREE_HIST_BINS", FT_OCT_HIST_BINS);
        k = ft::distribute_octree_hist(packed.data(), n, minX, maxX, minY, maxY, N, bins, keep);
        if (k == -2 && !dbgInt("FT_DEBUG_OCTREE_HIST_STRICT", 0)) {
            keep.clear();
            k = ft::distribute_octree_rounds(packed.data(), n, minX, maxX, minY, maxY, N, keep);
        }
    } else if (usePaths == 2 && n < 65535) k = ft::distribute_octree_rounds(packed.data(), n, minX, maxX, minY, maxY, N, keep);
    else if (usePaths == 1 && n < 65535) k = ft::distribute_octree_paths(packed.data(), n, minX, maxX, minY, maxY, N, keep);
    else k = ft::distribute_octree(packed.data(), n, minX, maxX, minY, maxY, N, ws, keep);
    if (k < 0) {
        ft_set_error("ft_octree_distribute: path-code workspace overflow");
        return FT_ERR_CAPACITY;
    }
    *n_out = k;
    if (k > capacity) {
        ft_set_error("ft_octree_distribute: capacity too small");
        return FT_ERR_CAPACITY;
    }
    for (int i = 0; i < k; i++) out_idx[i] = keep[i];
    return FT_OK;
}

int ft_level_geometry(int width, int height, int nfeatures, float scale_factor, int nlevels, int *level_w,
                      int *level_h, int *quota, int *n_cols, int *n_rows, int *w_cell, int *h_cell) {
    FT_REQUIRE(nlevels >= 1 && nlevels <= FT_MAX_LEVELS && scale_factor > 1.0f && nfeatures > 0, "bad parameters");
    ft_extractor ex;
    ex.nfeatures = nfeatures;
    ex.scaleFactor = scale_factor;
    ex.nlevels = nlevels;
    ex.width = width;
    ex.height = height;
    std::vector<FtTap> taps;
    int rc = buildGeometry(&ex, taps);
    if (rc != FT_OK) return rc;
    for (int l = 0; l < nlevels; l++) {
        const FtLevelGeom &v = ex.geom.lv[l];
        if (level_w) level_w[l] = v.w;
        if (level_h) level_h[l] = v.h;
        if (quota) quota[l] = ex.quota[l];
        if (n_cols) n_cols[l] = v.nCols;
        if (n_rows) n_rows[l] = v.nRows;
        if (w_cell) w_cell[l] = v.wCell;
        if (h_cell) h_cell[l] = v.hCell;
    }
    return FT_OK;
}

int ft_extractor_destroy(ft_extractor *ex) {
    if (!ex) return FT_OK;
    freeAll(ex);
    ex->ctx->liveObjects--;
    delete ex;
    return FT_OK;
}

int ft_extractor_levels(const ft_extractor *ex) { return ex ? ex->nlevels : 0; }
int ft_extractor_max_batch(const ft_extractor *ex) { return ex ? ex->maxBatch : 0; }
int ft_extractor_max_keypoints(const ft_extractor *ex) { return ex ? ex->geom.maxKp : 0; }

int ft_extractor_scale_factors(const ft_extractor *ex, float *scale, float *inv_scale, float *sigma2,
                               float *inv_sigma2) {
    FT_REQUIRE(ex, "null extractor");
    for (int i = 0; i < ex->nlevels; i++) {
        if (scale) scale[i] = ex->sf[i];
        if (inv_scale) inv_scale[i] = ex->invsf[i];
        if (sigma2) sigma2[i] = ex->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = ex->invsigma2[i];
    }
    return FT_OK;
}

int ft_extractor_features_per_level(const ft_extractor *ex, int *quota) {
    FT_REQUIRE(ex && quota, "null argument");
    for (int i = 0; i < ex->nlevels; i++) quota[i] = ex->quota[i];
    return FT_OK;
}

int ft_extractor_level_size(const ft_extractor *ex, int level, int *width, int *height) {
    FT_REQUIRE(ex && level >= 0 && level < ex->nlevels, "level out of range");
    if (width) *width = ex->geom.lv[level].w;
    if (height) *height = ex->geom.lv[level].h;
    return FT_OK;
}

int ft_extract_batch(ft_extractor *ex, const uint8_t *const *images, int batch, int on_device, int width,
                     int height, int stride, int lap0, int lap1, ft_keypoint *keypoints, uint8_t *descriptors,
                     int capacity, int *n_keypoints, int *n_mono) {
    FT_REQUIRE(ex, "null extractor");
    if (!images || width <= 0 || height <= 0) {
        ft_set_error("extract: empty image");
        return FT_ERR_EMPTY;
    }
    FtTimer tAll;
    int rc = ft_set_device(ex->ctx);
    if (rc != FT_OK) return rc;
    rc = ft_extract_foreign_wait(ex);  // (in front of any capture: the graph launch is enqueued behind it)
    if (rc != FT_OK) return rc;
    const int S = ft_pipeline_depth(ex->tune, batch, ex->deviceOctree);
    const int sb = (batch + S - 1) / S;
    // everything a batch needs with the device octree, enqueued without a host synchronisation (capture != 0: ex->stream is
    // being captured into a graph; the octree / stage-B streams fork from it through events and are joined back)
    const bool deliver = batch <= 8;  // (FtDeliverArgs, ft_internal.h)
    // a wide batch whose output arrays lie in pinned memory of this context (ft_host_malloc): the device writes the results there
    // itself, in the reference's output order (k_deliver_ordered) - no staging copy, no pass of the host over the keypoints
    static const bool stagedOnly = ft_debug_env("FT_DEBUG_STAGED_OUTPUTS") != nullptr;  // (A/B aid: the staging copy + the host's pass)
    const bool direct = !stagedOnly && !deliver && ex->deviceOctree && capacity > 0 && (keypoints || descriptors) &&
                        (!keypoints || ft_host_block_contains(ex->ctx, keypoints, sizeof(ft_keypoint) * (size_t)batch * capacity)) &&
                        (!descriptors || ft_host_block_contains(ex->ctx, descriptors, (size_t)32 * batch * capacity));
    bool deliveredDirect = false;
    auto deliverOrdered = [&](int b0, int nb) -> int {
        deliveredDirect = true;
        FtOrderedArgs a;
        a.keys = ex->d_keys;
        a.desc = ex->d_desc;
        a.nSel = ex->d_nSel;
        a.oKeys = keypoints;
        a.oDesc = descriptors;
        a.oMono = ex->h_nMono;
        a.srcStride = ex->geom.maxKp;
        a.capacity = capacity;
        a.b0 = b0;
        a.lap0 = (float)lap0;
        a.lap1 = (float)lap1;
        return ft_launch_deliver_ordered(ex->streamB, nb, a);
    };
    auto enqueueDevice = [&](int capture) -> int {
        int r = ft_extract_prepare(ex, images, batch, on_device, width, height, stride);
        if (r != FT_OK) return r;
        for (int s = 0, b0 = 0; b0 < batch; s++, b0 += sb) {
            const int nb = std::min(sb, batch - b0);
            r = ft_extract_launch_a(ex, b0, nb, nullptr);
            if (r == FT_OK) r = ft_extract_launch_octree(ex, s, b0, nb, ex->evA[s]);
            if (r != FT_OK) return r;
        }
        for (int s = 0, b0 = 0; b0 < batch; s++, b0 += sb) {
            const int nb = std::min(sb, batch - b0);
            FT_HIP(hipStreamWaitEvent(ex->streamB, ex->evA[s], 0));
            r = ft_extract_launch_b(ex, b0, nb, ex->streamB);
            if (r == FT_OK && !deliver) r = ft_extract_download(ex, b0, nb, ex->streamB);
            if (r != FT_OK) return r;
        }
        if (deliver) {  // latency mode: one kernel writes keypoints, descriptors and counters to the pinned staging
            FtDeliverArgs d;
            memset(&d, 0, sizeof d);
            d.keysL = ex->d_keys;
            d.descL = ex->d_desc;
            d.nL = ex->d_nSel;
            d.overflowL = ex->d_overflow;
            d.oKeysL = ex->h_keys;
            d.oDescL = ex->h_desc;
            d.oNL = ex->h_nSel;
            d.oOverflowL = ex->h_overflow;
            d.srcStride = d.dstStride = ex->geom.maxKp;
            r = ft_launch_deliver(ex->streamB, batch, d);
        } else {
            r = ft_extract_finish_counts(ex, batch, ex->streamB);
        }
        if (r != FT_OK) return r;
        if (capture) {
            FT_HIP(hipEventRecord(ex->evJoin, ex->streamB));
            FT_HIP(hipStreamWaitEvent(ex->stream, ex->evJoin, 0));
        }
        return FT_OK;
    };
    // images with a level beyond the device octree's limits (seen after the batch has drained): each is redone with the
    // host octree in place, the rest of the batch keeps its device results
    auto repairOverflow = [&]() -> int {
        std::vector<int> slots;
        int r = ft_extract_overflow_slots(ex, batch, slots);
        if (r != FT_OK) return r;
        std::vector<std::pair<ft_extractor *, int>> jobs;
        for (int b : slots) jobs.emplace_back(ex, b);
        r = ft_extract_repair_prepare(jobs);
        if (r != FT_OK) return r;
        for (int b : slots) {
            ex->ctx->addStat("extract.device_octree_fallbacks", 1);
            r = ft_extract_repair_launch(ex, b, ex->streamB);
            if (r == FT_OK) r = deliveredDirect ? deliverOrdered(b, 1) : ft_extract_download(ex, b, 1, ex->streamB);
            if (r != FT_OK) return r;
        }
        FT_HIP(hipStreamSynchronize(ex->streamB));
        return FT_OK;
    };
    bool done = false;
    const bool devWanted = ex->deviceOctree;
    const bool graphsOn = ex->tune.graph != 0;
    if (graphsOn && !ex->graphDisabled && ex->ownStreams && ex->deviceOctree && !ex->ctx->kernelTiming && batch >= 1 && batch <= 8 &&
        batch <= ex->maxBatch && width == ex->width && height == ex->height && stride >= width) {
        bool ok = true;
        for (int b = 0; b < batch; b++)
            if (!images[b]) ok = false;
        if (ok) {
            ft_extractor::GraphKey key;
            key.batch = batch; key.onDevice = on_device; key.width = width; key.height = height; key.stride = stride;
            key.aligned = 1;
            key.bigGrid = ex->bigGrid + (ex->histOn ? (1 << 24) : 0);
            if (on_device) {
                if (stride & 3) key.aligned = 0;
                for (int b = 0; b < batch; b++)
                    if ((uintptr_t)images[b] & 3) key.aligned = 0;
            } else if (ft_extract_ensure_stage(ex) != FT_OK) {
                ok = false;
                (void)hipGetLastError();
            }
            bool launched = false;
            if (!ok) {
            } else if (ex->graphExec && key == ex->graphKey) {
                if (on_device)
                    for (int b = 0; b < batch; b++) ex->h_l0[b] = images[b];
                else
                    ft_extract_restage(ex, images, batch, width, height, stride);
                ex->lastBatch = batch;
                FT_HIP(hipGraphLaunch(ex->graphExec, ex->stream));
                launched = true;
            } else {
                if (ex->graphExec) {
                    hipGraphExecDestroy(ex->graphExec);
                    ex->graphExec = nullptr;
                }
                hipGraph_t graph = nullptr;
                ex->stageHost = !on_device;
                hipError_t ce = hipStreamBeginCapture(ex->stream, hipStreamCaptureModeThreadLocal);
                if (ce == hipSuccess) {
                    rc = enqueueDevice(1);
                    ce = hipStreamEndCapture(ex->stream, &graph);
                    if (rc == FT_OK && ce == hipSuccess && graph) ce = hipGraphInstantiate(&ex->graphExec, graph, nullptr, nullptr, 0);
                    if (graph) hipGraphDestroy(graph);
                }
                if (rc != FT_OK || ce != hipSuccess || !ex->graphExec) {
                    (void)hipGetLastError();  // capture is an optimisation: plain enqueueing from now on
                    ex->graphDisabled = true;
                    ex->graphExec = nullptr;
                    ex->ctx->addStat("extract.graph_capture_failed", 0);
                    rc = FT_OK;
                    ex->stageHost = false;
                } else {
                    ex->stageHost = false;
                    ex->graphKey = key;
                    ex->ctx->addStat("extract.graph_captures", 0);
                    FT_HIP(hipGraphLaunch(ex->graphExec, ex->stream));
                    launched = true;
                }
            }
            if (launched) {
                ex->ctx->addStat("extract.device_octree_batches", 0);
                FT_HIP(hipStreamSynchronize(ex->stream));
                ft_extract_update_big_grid(ex);
                if (ex->h_overflow[0]) {  // some image met a level beyond the device octree's limits: that image is repaired
                    rc = repairOverflow();
                    if (rc != FT_OK) return rc;
                }
                done = true;
            }
        }
    }
    if (!done) {
        rc = ft_extract_prepare(ex, images, batch, on_device, width, height, stride);
        if (rc != FT_OK) return rc;
    }
    if (!done) {
        const bool dev = ex->deviceOctree;
        if (dev) ex->ctx->addStat("extract.device_octree_batches", 0);
        for (int s = 0, b0 = 0; b0 < batch; s++, b0 += sb) {
            const int nb = std::min(sb, batch - b0);
            rc = ft_extract_launch_a(ex, b0, nb, dev ? nullptr : ex->evA[s]);
            if (rc != FT_OK) return rc;
            if (dev) {
                rc = ft_extract_launch_octree(ex, s, b0, nb, ex->evA[s]);
                if (rc != FT_OK) return rc;
            }
        }
        double tOct = 0;
        for (int s = 0, b0 = 0; b0 < batch; s++, b0 += sb) {
            const int nb = std::min(sb, batch - b0);
            if (dev) {
                FT_HIP(hipStreamWaitEvent(ex->streamB, ex->evA[s], 0));
            } else {
                FT_HIP(hipEventSynchronize(ex->evA[s]));
                FtTimer tO;
                rc = ft_extract_octree(ex, b0, nb);
                if (rc != FT_OK) return rc;
                tOct += tO.ms();
            }
            rc = ft_extract_launch_b(ex, b0, nb, ex->streamB);
            if (rc != FT_OK) return rc;
            rc = (direct && dev) ? deliverOrdered(b0, nb) : ft_extract_download(ex, b0, nb, ex->streamB);
            if (rc != FT_OK) return rc;
        }
        rc = ft_extract_finish_counts(ex, batch, ex->streamB);
        if (rc != FT_OK) return rc;
        FT_HIP(hipStreamSynchronize(ex->streamB));
        FT_HIP(hipStreamSynchronize(ex->stream));
        ex->evt.resolve(ex->ctx);
        if (!dev) ex->ctx->addStat("extract.octree(host)", tOct);
        if (dev) ft_extract_update_big_grid(ex);
        if (dev && ex->h_overflow[0]) {
            rc = repairOverflow();
            if (rc != FT_OK) return rc;
        }
    }
    ex->deviceOctree = devWanted;
    if (deliveredDirect) {  // (the arrays are filled; what is left is the counts)
        for (int b = 0; b < batch; b++)
            if (ex->h_nSel[b] > capacity) {
                ft_set_error("extract: output capacity too small (use ft_extractor_max_keypoints)");
                return FT_ERR_CAPACITY;
            }
        for (int b = 0; b < batch; b++) {
            if (n_keypoints) n_keypoints[b] = ex->h_nSel[b];
            if (n_mono) n_mono[b] = ex->h_nMono[b];
        }
        ex->ctx->addStat("extract.delivered_in_order_on_device", batch);
        rc = FT_OK;
    } else {
        rc = assembleOutputs(ex, batch, lap0, lap1, keypoints, descriptors, capacity, n_keypoints, n_mono);
    }
    ex->ctx->addStat("extract.total", tAll.ms());
    return rc;
}

int ft_extract(ft_extractor *ex, const uint8_t *image, int width, int height, int stride, int lap0, int lap1,
               ft_keypoint *keypoints, uint8_t *descriptors, int capacity, int *n_keypoints, int *n_mono) {
    if (!image || width <= 0 || height <= 0) {
        ft_set_error("extract: empty image");
        return FT_ERR_EMPTY;
    }
    const uint8_t *imgs[1] = {image};
    return ft_extract_batch(ex, imgs, 1, 0, width, height, stride, lap0, lap1, keypoints, descriptors, capacity,
                            n_keypoints, n_mono);
}

int ft_extractor_device_level(ft_extractor *ex, int slot, int level, const uint8_t **dptr, int *pitch) {
    FT_REQUIRE(ex && dptr && pitch, "null argument");
    FT_REQUIRE(slot >= 0 && slot < ex->lastBatch, "slot holds no image");
    FT_REQUIRE(level >= 0 && level < ex->nlevels, "level out of range");
    const FtGeom &g = ex->geom;
    if (level == 0) {
        *dptr = ex->h_l0[slot];
        *pitch = ex->l0pitch;
    } else {
        *dptr = ex->d_pyr + (size_t)slot * g.pyrPerSlot + g.lv[level].off;
        *pitch = g.lv[level].pitch;
    }
    return FT_OK;
}

int ft_extractor_download_level(ft_extractor *ex, int slot, int level, uint8_t *dst, int dst_stride) {
    FT_REQUIRE(dst, "null destination");
    const uint8_t *src = nullptr;
    int pitch = 0;
    int rc = ft_extractor_device_level(ex, slot, level, &src, &pitch);
    if (rc != FT_OK) return rc;
    rc = ft_set_device(ex->ctx);
    if (rc != FT_OK) return rc;
    const FtLevelGeom &v = ex->geom.lv[level];
    FT_REQUIRE(dst_stride >= v.w, "destination stride smaller than the level width");
    FT_HIP(hipStreamSynchronize(ex->stream));
    FT_HIP(hipStreamSynchronize(ex->streamB));
    FT_HIP(hipMemcpy2D(dst, dst_stride, src, pitch, v.w, v.h, hipMemcpyDeviceToHost));
    return FT_OK;
}

int ft_extractor_download_blurred_level(ft_extractor *ex, int slot, int level, uint8_t *dst, int dst_stride) {
    FT_REQUIRE(dst, "null destination");
    const uint8_t *src = nullptr;
    int pitch = 0;
    int rc = ft_extractor_device_level(ex, slot, level, &src, &pitch);
    if (rc != FT_OK) return rc;
    rc = ft_set_device(ex->ctx);
    if (rc != FT_OK) return rc;
    const FtLevelGeom &v = ex->geom.lv[level];
    FT_REQUIRE(dst_stride >= v.w, "destination stride smaller than the level width");
    FT_HIP(hipStreamSynchronize(ex->stream));
    FT_HIP(hipStreamSynchronize(ex->streamB));
    uint8_t *tmp = nullptr;
    FT_HIP(hipMalloc((void **)&tmp, (size_t)v.w * v.h));
    rc = ft_launch_blur_level(ex->stream, src, pitch, v.w, v.h, tmp, v.w);
    hipError_t e = rc == FT_OK ? hipStreamSynchronize(ex->stream) : hipSuccess;
    if (rc == FT_OK && e == hipSuccess) e = hipMemcpy2D(dst, dst_stride, tmp, v.w, v.w, v.h, hipMemcpyDeviceToHost);
    hipFree(tmp);
    if (rc != FT_OK) return rc;
    FT_HIP(e);
    return FT_OK;
}

int ft_extractor_download_candidates(ft_extractor *ex, int slot, int level, int *xys, int capacity, int *n) {
    FT_REQUIRE(ex && n, "null argument");
    FT_REQUIRE(slot >= 0 && slot < ex->lastBatch, "slot holds no image");
    FT_REQUIRE(level >= 0 && level < ex->nlevels, "level out of range");
    int rc = ft_set_device(ex->ctx);
    if (rc != FT_OK) return rc;
    FT_HIP(hipStreamSynchronize(ex->stream));
    const FtGeom &g = ex->geom;
    std::vector<uint32_t> tmp;
    const uint32_t *c = ex->h_cand ? ex->h_cand + (size_t)slot * g.candPerSlot + g.lv[level].candBase : nullptr;
    int cnt = ex->h_candCount[slot * g.nlevels + level];
    if (ex->deviceOctree) {  // the lists never left the device
        FT_HIP(hipStreamSynchronize(ex->streamB));
        FT_HIP(hipMemcpy(&cnt, ex->d_candCountDev + slot * g.nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
        tmp.resize(std::max(cnt, 1));
        FT_HIP(hipMemcpy(tmp.data(), ex->d_candDev + (size_t)slot * g.candPerSlot + g.lv[level].candBase,
                         sizeof(uint32_t) * cnt, hipMemcpyDeviceToHost));
        // the device path delivers every cell's candidates in arbitrary order (k_octree ranks them by their
        // coordinates); present them in the reference's emission order: cell row, cell column, row-major in the cell
        const FtLevelGeom &L = g.lv[level];
        auto rank = [&](uint32_t v) -> uint64_t {
            const int x = (int)(v & 0xfffu) - 3, y = (int)((v >> 12) & 0xfffu) - 3;
            const int cj = std::min(x / std::max(L.wCell, 1), std::max(L.nCols - 1, 0));
            const int ci = std::min(y / std::max(L.hCell, 1), std::max(L.nRows - 1, 0));
            return ((uint64_t)ci << 48) | ((uint64_t)cj << 32) | ((uint64_t)y << 16) | (uint64_t)x;
        };
        std::sort(tmp.begin(), tmp.begin() + cnt, [&](uint32_t a, uint32_t b) { return rank(a) < rank(b); });
        c = tmp.data();
    }
    *n = cnt;
    if (xys) {
        for (int i = 0; i < cnt && i < capacity; i++) {
            xys[3 * i] = (int)(c[i] & 0xfffu);
            xys[3 * i + 1] = (int)((c[i] >> 12) & 0xfffu);
            xys[3 * i + 2] = (int)(c[i] >> 24);
        }
    }
    return FT_OK;
}

// Test tap: the device octree of ONE level on caller-provided candidates (slot 0 of the extractor).  tiers: bit 0 is always
// on (k_octree), bit 1 allows the histogram tier, bit 2 the sorted big tier.  *tier = the tier that produced the result
// (1, 2, 3) or 0 when the level was left to the host (overflow flag raised; *n_out = 0).
int ft_extractor_octree_on_device(ft_extractor *ex, int level, const int *xys, int n, int tiers, int *out_xys, int capacity,
                                  int *n_out, int *tier) {
    FT_REQUIRE(ex && xys && n_out && tier && n >= 0, "null argument");
    FT_REQUIRE(level >= 0 && level < ex->nlevels, "level out of range");
    FT_REQUIRE(ex->deviceOctree, "the extractor keeps the octree on the host");
    const FtGeom &g = ex->geom;
    FT_REQUIRE(n <= g.lv[level].candCap, "more candidates than the level's list holds");
    int rc = ft_set_device(ex->ctx);
    if (rc != FT_OK) return rc;
    std::vector<uint32_t> packed(std::max(n, 1));
    for (int i = 0; i < n; i++) {
        FT_REQUIRE(xys[3 * i] >= 3 && xys[3 * i] - 3 < g.lv[level].nCols * g.lv[level].wCell && xys[3 * i] < 4096 && xys[3 * i + 1] >= 3 &&
                       xys[3 * i + 1] - 3 < g.lv[level].nRows * g.lv[level].hCell && xys[3 * i + 1] < 4096 && xys[3 * i + 2] >= 1 &&
                       xys[3 * i + 2] < 256,
                   "candidate out of range");
        packed[i] = ft_pack_cand(xys[3 * i], xys[3 * i + 1], xys[3 * i + 2]);
    }
    hipStream_t so = ex->streamO[0];
    FT_HIP(hipStreamSynchronize(ex->stream));
    FT_HIP(hipStreamSynchronize(so));
    std::vector<int> counts(g.nlevels, 0);
    counts[level] = n;
    FT_HIP(hipMemcpy(ex->d_candCountDev, counts.data(), sizeof(int) * g.nlevels, hipMemcpyHostToDevice));
    if (n) FT_HIP(hipMemcpy(ex->d_candDev + g.lv[level].candBase, packed.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    FT_HIP(hipMemset(ex->d_bigCount, 0, 4 * sizeof(int)));
    FtOctArgs a = ex->octLayout;
    a.cand = ex->d_candDev;
    a.candCount = ex->d_candCountDev;
    a.sel = ex->d_sel;
    a.selCount = ex->d_selCount;
    a.overflow = ex->d_overflow;
    a.ovSlot = ex->d_ovSlot;
    a.histCap = (tiers & 2) && ex->histEnabled ? g.nlevels : 0;
    a.histGrid = g.nlevels;
    a.histWanted = 0;
    a.histFirst = (tiers & 8) && ex->histEnabled ? 1 : 0;  // bit 3: the histogram formulation for every level, as latency-mode launches run it
    a.sortCap = (tiers & 4) && a.bigN ? g.nlevels : 0;
    a.low = ex->d_octLow;
    a.bigCount = ex->d_bigCount;
    a.bigList = ex->d_bigList;
    a.sortList = ex->d_sortList;
    rc = ft_launch_octree(so, g, 1, a);
    if (rc != FT_OK) return rc;
    FT_HIP(hipStreamSynchronize(so));
    int cnt[4] = {0, 0, 0, 0}, ov = 0, k = 0;
    FT_HIP(hipMemcpy(cnt, ex->d_bigCount, sizeof cnt, hipMemcpyDeviceToHost));
    FT_HIP(hipMemcpy(&ov, ex->d_overflow, sizeof(int), hipMemcpyDeviceToHost));
    FT_HIP(hipMemcpy(&k, ex->d_selCount + level, sizeof(int), hipMemcpyDeviceToHost));
    FT_HIP(hipMemset(ex->d_overflow, 0, sizeof(int)));
    FT_HIP(hipMemset(ex->d_ovSlot, 0, sizeof(int)));
    FT_HIP(hipMemset(ex->d_bigCount, 0, 4 * sizeof(int)));
    *tier = ov ? 0 : cnt[1] > 0 ? 3 : (cnt[0] > 0 || a.histFirst) ? 2 : 1;
    if (ov) k = 0;
    *n_out = k;
    if (k > capacity) {
        ft_set_error("ft_extractor_octree_on_device: capacity too small");
        return FT_ERR_CAPACITY;
    }
    std::vector<FtSelKp> sel(std::max(k, 1));
    if (k) FT_HIP(hipMemcpy(sel.data(), ex->d_sel + ex->levelOff[level], sizeof(FtSelKp) * k, hipMemcpyDeviceToHost));
    const int minB = FT_EDGE_THRESHOLD - 3;
    for (int i = 0; i < k && out_xys; i++) {
        out_xys[3 * i] = sel[i].x - minB;
        out_xys[3 * i + 1] = sel[i].y - minB;
        out_xys[3 * i + 2] = sel[i].response;
    }
    return FT_OK;
}

}  // extern "C"
