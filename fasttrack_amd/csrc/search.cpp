// Host side of the two projection searches: replaces launchSearchLocalPointsKernel /
// launchPoseEstimationKernel (reference include/Kernels/KernelController.h:40-46) together with the
// acceptance loops the reference keeps in the caller (src/ORBmatcher.cc:241-308, 2013-2081).
// Windowing, level/box tests and every Hamming distance run on the device (kernels_search.hip); the
// host only marshals arrays, drives the fixed-point passes and replays the O(M) write list in map
// point order to produce mvpMapPoints / the rotation histogram, exactly as the reference's caller does.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "ft_host.h"
#include "ft_search.h"

#define FT_REQUIRE(cond, msg)               \
    do {                                    \
        if (!(cond)) {                      \
            ft_set_error(std::string(msg)); \
            return FT_ERR_INVALID;          \
        }                                   \
    } while (0)

namespace {

struct Arena {
    size_t off = 0;
    size_t take(size_t bytes) {
        const size_t o = off;
        off = (off + bytes + 63) & ~(size_t)63;
        return o;
    }
};

int ensureScratch(ft_context *ctx, size_t devBytes, size_t pinBytes) {
    if (devBytes > ctx->scratchDevBytes) {
        if (ctx->scratchDev) hipFree(ctx->scratchDev);
        ctx->scratchDev = nullptr;
        ctx->scratchDevBytes = 0;
        const size_t want = devBytes + devBytes / 2;
        FT_HIP(hipMalloc(&ctx->scratchDev, want));
        ctx->scratchDevBytes = want;
    }
    if (pinBytes > ctx->scratchPinBytes) {
        if (ctx->scratchPin) hipHostFree(ctx->scratchPin);
        ctx->scratchPin = nullptr;
        ctx->scratchPinBytes = 0;
        const size_t want = pinBytes + pinBytes / 2;
        FT_HIP(hipHostMalloc(&ctx->scratchPin, want, hipHostMallocDefault));
        ctx->scratchPinBytes = want;
    }
    return FT_OK;
}

struct FrameLayout {
    size_t keys, keysR, desc, uright, holder, l2r, r2l;
    int nLeftKeys, nRightKeys;
};

int checkFrame(const ft_frame_view *F) {
    FT_REQUIRE(F, "null frame view");
    FT_REQUIRE(F->N >= 0 && F->N < (1 << 24), "frame keypoint count out of range");
    FT_REQUIRE(F->Nleft == -1 || (F->Nleft >= 0 && F->Nleft <= F->N), "Nleft out of range");
    FT_REQUIRE(F->N == 0 || (F->keys && F->descriptors && F->holder_obs), "frame arrays are null");
    FT_REQUIRE(F->Nleft == -1 || F->N == F->Nleft || F->keys_right, "keys_right is null");
    FT_REQUIRE(F->Nleft == -1 || (F->left_to_right && F->right_to_left), "stereo match tables are null");
    FT_REQUIRE(F->scale_factors && F->nlevels >= 1 && F->nlevels <= FT_MAX_LEVELS, "scale factors missing");
    FT_REQUIRE(F->cam_model == 0 || F->cam_model == 1, "unknown camera model");
    return FT_OK;
}

void layoutFrame(const ft_frame_view *F, Arena &a, FrameLayout &L) {
    L.nLeftKeys = F->Nleft == -1 ? F->N : F->Nleft;
    L.nRightKeys = F->Nleft == -1 ? 0 : F->N - F->Nleft;
    L.keys = a.take(sizeof(ft_keypoint) * std::max(L.nLeftKeys, 1));
    L.keysR = a.take(sizeof(ft_keypoint) * std::max(L.nRightKeys, 1));
    L.desc = a.take((size_t)32 * std::max(F->N, 1));
    L.uright = a.take(sizeof(float) * std::max(F->N, 1));
    L.holder = a.take(sizeof(int) * std::max(F->N, 1));
    L.l2r = a.take(sizeof(int) * std::max(L.nLeftKeys, 1));
    L.r2l = a.take(sizeof(int) * std::max(L.nRightKeys, 1));
}

void stageFrame(const ft_frame_view *F, const FrameLayout &L, uint8_t *pin) {
    if (L.nLeftKeys) memcpy(pin + L.keys, F->keys, sizeof(ft_keypoint) * L.nLeftKeys);
    if (L.nRightKeys) memcpy(pin + L.keysR, F->keys_right, sizeof(ft_keypoint) * L.nRightKeys);
    if (F->N) memcpy(pin + L.desc, F->descriptors, (size_t)32 * F->N);
    if (F->uright && F->N) memcpy(pin + L.uright, F->uright, sizeof(float) * F->N);
    if (F->N) memcpy(pin + L.holder, F->holder_obs, sizeof(int) * F->N);
    if (F->Nleft != -1) {
        if (L.nLeftKeys) memcpy(pin + L.l2r, F->left_to_right, sizeof(int) * L.nLeftKeys);
        if (L.nRightKeys) memcpy(pin + L.r2l, F->right_to_left, sizeof(int) * L.nRightKeys);
    }
}

FtDevFrame devFrame(const ft_frame_view *F, const FrameLayout &L, uint8_t *dev) {
    FtDevFrame D;
    memset(&D, 0, sizeof D);
    D.N = F->N;
    D.Nleft = F->Nleft;
    D.mnMinX = F->mnMinX; D.mnMinY = F->mnMinY; D.mnMaxX = F->mnMaxX; D.mnMaxY = F->mnMaxY;
    D.invW = F->grid_inv_w; D.invH = F->grid_inv_h;
    D.mbf = F->mbf; D.mb = F->mb;
    D.keys = (const ft_keypoint *)(dev + L.keys);
    D.keysR = (const ft_keypoint *)(dev + L.keysR);
    D.desc = dev + L.desc;
    D.uright = F->uright ? (const float *)(dev + L.uright) : nullptr;
    D.holderObs = (const int *)(dev + L.holder);
    D.l2r = F->Nleft != -1 ? (const int *)(dev + L.l2r) : nullptr;
    D.r2l = F->Nleft != -1 ? (const int *)(dev + L.r2l) : nullptr;
    D.camModel = F->cam_model;
    memcpy(D.cam, F->cam, sizeof D.cam);
    memcpy(D.Trl, F->Trl, sizeof D.Trl);
    for (int i = 0; i < F->nlevels; i++) D.sf[i] = F->scale_factors[i];
    D.nlevels = F->nlevels;
    return D;
}

// runs `search` passes until a pass changes nothing; leaves the final results in resFinal (device)
template <typename SearchFn>
int fixedPoint(ft_context *ctx, hipStream_t st, int nPoints, int nKp, int *resA, int *resB, int *head, int *next,
               int *changed, SearchFn search, int **resFinal, int *passes) {
    int rc = ft_launch_fill_i32(st, head, nKp, -1);
    if (rc != FT_OK) return rc;
    rc = ft_launch_fill_i32(st, resB, 4 * nPoints, -2);
    if (rc != FT_OK) return rc;
    int *cur = resA, *prev = resB;
    int pass = 0;
    const int maxPasses = 2 * nPoints + 4;
    for (;;) {
        rc = search(cur);
        if (rc != FT_OK) return rc;
        rc = ft_launch_build_claims(st, cur, prev, nPoints, nKp, head, next, changed);
        if (rc != FT_OK) return rc;
        int h = 0;
        FT_HIP(hipMemcpyAsync(&h, changed, sizeof(int), hipMemcpyDeviceToHost, st));
        FT_HIP(hipStreamSynchronize(st));
        std::swap(cur, prev);
        pass++;
        if (!h) break;
        if (pass >= maxPasses) {
            ft_set_error("projection search: claim resolution did not converge");
            return FT_ERR_HIP;
        }
    }
    *resFinal = prev;
    *passes = pass;
    return FT_OK;
}

}  // namespace

extern "C" {

int ft_search_local_points(ft_context *ctx, ft_frame_view *F, const ft_local_points *P, float th, float nn_ratio,
                           int *assign, int *n_matches, int *best_dist, int *best_dist2, int *best_level,
                           int *best_level2, int *best_idx, int *best_dist_r, int *best_dist2_r, int *best_level_r,
                           int *best_level2_r, int *best_idx_r) {
    FT_REQUIRE(ctx && P && assign, "ft_search_local_points: null argument");
    int rc = checkFrame(F);
    if (rc != FT_OK) return rc;
    const int M = P->M, N = F->N;
    FT_REQUIRE(M >= 0 && M < (1 << 22), "map point count out of range");
    FT_REQUIRE(M == 0 || (P->skip && P->in_view && P->in_view_r && P->level && P->level_r && P->view_cos &&
                          P->view_cos_r && P->proj_x && P->proj_y && P->proj_xr && P->proj_yr && P->descriptors &&
                          P->observations),
               "local point arrays are null");
    for (int i = 0; i < N; i++) assign[i] = -1;
    if (n_matches) *n_matches = 0;
    int *outs[10] = {best_dist, best_dist2, best_level, best_level2, best_idx,
                     best_dist_r, best_dist2_r, best_level_r, best_level2_r, best_idx_r};
    if (M == 0 || N == 0) {
        for (int k = 0; k < 10; k++)
            if (outs[k])
                for (int i = 0; i < M; i++) outs[k][i] = (k % 5 == 0 || k % 5 == 1) ? 256 : -1;
        return FT_OK;
    }
    rc = ft_set_device(ctx);
    if (rc != FT_OK) return rc;
    std::lock_guard<std::mutex> lk(ctx->matchMutex);
    FtTimer tAll;
    // ---- layout: inputs | work | outputs ----
    Arena a;
    FrameLayout FL;
    layoutFrame(F, a, FL);
    const size_t oSkip = a.take(M), oIn = a.take(M), oInR = a.take(M);
    const size_t oLevel = a.take(4 * (size_t)M), oLevelR = a.take(4 * (size_t)M);
    const size_t oVc = a.take(4 * (size_t)M), oVcR = a.take(4 * (size_t)M);
    const size_t oPx = a.take(4 * (size_t)M), oPy = a.take(4 * (size_t)M), oPxr = a.take(4 * (size_t)M),
                 oPyr = a.take(4 * (size_t)M);
    const size_t oDesc = a.take(32 * (size_t)M), oObs = a.take(4 * (size_t)M);
    const size_t inputBytes = a.off;
    const size_t oResA = a.take(16 * (size_t)M), oResB = a.take(16 * (size_t)M), oHead = a.take(4 * (size_t)N),
                 oNext = a.take(16 * (size_t)M), oChanged = a.take(64);
    const size_t oRaw = a.take(40 * (size_t)M);
    const size_t total = a.off;
    const size_t outBytes = 16 * (size_t)M + 40 * (size_t)M + 64;
    rc = ensureScratch(ctx, total, std::max(inputBytes, outBytes));
    if (rc != FT_OK) return rc;
    uint8_t *pin = (uint8_t *)ctx->scratchPin, *dev = (uint8_t *)ctx->scratchDev;
    stageFrame(F, FL, pin);
    memcpy(pin + oSkip, P->skip, M);
    memcpy(pin + oIn, P->in_view, M);
    memcpy(pin + oInR, P->in_view_r, M);
    memcpy(pin + oLevel, P->level, 4 * (size_t)M);
    memcpy(pin + oLevelR, P->level_r, 4 * (size_t)M);
    memcpy(pin + oVc, P->view_cos, 4 * (size_t)M);
    memcpy(pin + oVcR, P->view_cos_r, 4 * (size_t)M);
    memcpy(pin + oPx, P->proj_x, 4 * (size_t)M);
    memcpy(pin + oPy, P->proj_y, 4 * (size_t)M);
    memcpy(pin + oPxr, P->proj_xr, 4 * (size_t)M);
    memcpy(pin + oPyr, P->proj_yr, 4 * (size_t)M);
    memcpy(pin + oDesc, P->descriptors, 32 * (size_t)M);
    memcpy(pin + oObs, P->observations, 4 * (size_t)M);
    hipStream_t st = ctx->stream;
    FT_HIP(hipMemcpyAsync(dev, pin, inputBytes, hipMemcpyHostToDevice, st));
    const FtDevFrame DF = devFrame(F, FL, dev);
    FtDevLocalPoints DP;
    DP.M = M;
    DP.skip = dev + oSkip; DP.inView = dev + oIn; DP.inViewR = dev + oInR;
    DP.level = (const int *)(dev + oLevel); DP.levelR = (const int *)(dev + oLevelR);
    DP.viewCos = (const float *)(dev + oVc); DP.viewCosR = (const float *)(dev + oVcR);
    DP.projX = (const float *)(dev + oPx); DP.projY = (const float *)(dev + oPy);
    DP.projXR = (const float *)(dev + oPxr); DP.projYR = (const float *)(dev + oPyr);
    DP.desc = dev + oDesc;
    FtClaims C;
    C.head = (const int *)(dev + oHead);
    C.next = (const int *)(dev + oNext);
    C.obs = (const int *)(dev + oObs);
    int *rawBase = (int *)(dev + oRaw);
    FtLocalRaw raw;
    raw.bestDist = rawBase; raw.bestDist2 = rawBase + M; raw.bestLevel = rawBase + 2 * M; raw.bestLevel2 = rawBase + 3 * M;
    raw.bestIdx = rawBase + 4 * M; raw.bestDistR = rawBase + 5 * M; raw.bestDist2R = rawBase + 6 * M;
    raw.bestLevelR = rawBase + 7 * M; raw.bestLevel2R = rawBase + 8 * M; raw.bestIdxR = rawBase + 9 * M;
    int *resFinal = nullptr, passes = 0;
    rc = fixedPoint(ctx, st, M, N, (int *)(dev + oResA), (int *)(dev + oResB), (int *)(dev + oHead), (int *)(dev + oNext),
                    (int *)(dev + oChanged),
                    [&](int *res) { return ft_launch_search_local(st, DF, DP, C, th, nn_ratio, res, raw); }, &resFinal,
                    &passes);
    if (rc != FT_OK) return rc;
    int *hRes = (int *)pin, *hRaw = (int *)(pin + 16 * (size_t)M + 64);
    FT_HIP(hipMemcpyAsync(hRes, resFinal, 16 * (size_t)M, hipMemcpyDeviceToHost, st));
    FT_HIP(hipMemcpyAsync(hRaw, rawBase, 40 * (size_t)M, hipMemcpyDeviceToHost, st));
    FT_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < 10; k++)
        if (outs[k]) memcpy(outs[k], hRaw + (size_t)k * M, 4 * (size_t)M);
    // replay the writes in map point order (ORBmatcher.cc:134-148 left, :203-214 right)
    int nm = 0;
    for (int i = 0; i < M; i++) {
        const int obs = P->observations[i];
        const int order[4] = {hRes[4 * i], hRes[4 * i + 1], hRes[4 * i + 3], hRes[4 * i + 2]};  // primL sideL sideR primR
        for (int k = 0; k < 4; k++) {
            const int kp = order[k];
            if (kp < 0) continue;
            F->holder_obs[kp] = obs;
            assign[kp] = i;
            nm++;
        }
    }
    if (n_matches) *n_matches = nm;
    ctx->addStat("search_local_points.total", tAll.ms());
    ctx->addStat("search_local_points.passes", passes);
    return FT_OK;
}

int ft_search_last_frame(ft_context *ctx, ft_frame_view *Cur, const ft_last_points *L, const float *Tcw, float th,
                         int forward, int backward, int check_orientation, int *assign, int *n_matches,
                         int *best_dist, int *best_idx, int *best_dist_r, int *best_idx_r) {
    FT_REQUIRE(ctx && L && Tcw && assign, "ft_search_last_frame: null argument");
    int rc = checkFrame(Cur);
    if (rc != FT_OK) return rc;
    const int M = L->N, N = Cur->N;
    FT_REQUIRE(M >= 0 && M < (1 << 22), "last-frame point count out of range");
    FT_REQUIRE(M == 0 || (L->valid && L->world_pos && L->descriptors && L->observations && L->octave && L->angle),
               "last-frame arrays are null");
    for (int i = 0; i < N; i++) assign[i] = -1;
    if (n_matches) *n_matches = 0;
    int *outs[4] = {best_dist, best_idx, best_dist_r, best_idx_r};
    if (M == 0 || N == 0) {
        for (int k = 0; k < 4; k++)
            if (outs[k])
                for (int i = 0; i < M; i++) outs[k][i] = (k % 2 == 0) ? 256 : -1;
        return FT_OK;
    }
    for (int i = 0; i < M; i++)
        FT_REQUIRE(!L->valid[i] || (L->octave[i] >= 0 && L->octave[i] < Cur->nlevels), "last-frame octave out of range");
    rc = ft_set_device(ctx);
    if (rc != FT_OK) return rc;
    std::lock_guard<std::mutex> lk(ctx->matchMutex);
    FtTimer tAll;
    Arena a;
    FrameLayout FL;
    layoutFrame(Cur, a, FL);
    const size_t oValid = a.take(M), oPos = a.take(12 * (size_t)M), oDesc = a.take(32 * (size_t)M),
                 oObs = a.take(4 * (size_t)M), oOct = a.take(4 * (size_t)M);
    const size_t inputBytes = a.off;
    const size_t oResA = a.take(16 * (size_t)M), oResB = a.take(16 * (size_t)M), oHead = a.take(4 * (size_t)N),
                 oNext = a.take(16 * (size_t)M), oChanged = a.take(64);
    const size_t oRaw = a.take(16 * (size_t)M);
    const size_t total = a.off;
    const size_t outBytes = 32 * (size_t)M + 64;
    rc = ensureScratch(ctx, total, std::max(inputBytes, outBytes));
    if (rc != FT_OK) return rc;
    uint8_t *pin = (uint8_t *)ctx->scratchPin, *dev = (uint8_t *)ctx->scratchDev;
    stageFrame(Cur, FL, pin);
    memcpy(pin + oValid, L->valid, M);
    memcpy(pin + oPos, L->world_pos, 12 * (size_t)M);
    memcpy(pin + oDesc, L->descriptors, 32 * (size_t)M);
    memcpy(pin + oObs, L->observations, 4 * (size_t)M);
    memcpy(pin + oOct, L->octave, 4 * (size_t)M);
    hipStream_t st = ctx->stream;
    FT_HIP(hipMemcpyAsync(dev, pin, inputBytes, hipMemcpyHostToDevice, st));
    const FtDevFrame DF = devFrame(Cur, FL, dev);
    FtDevLastPoints DL;
    DL.N = M;
    DL.valid = dev + oValid;
    DL.worldPos = (const float *)(dev + oPos);
    DL.desc = dev + oDesc;
    DL.octave = (const int *)(dev + oOct);
    FtClaims C;
    C.head = (const int *)(dev + oHead);
    C.next = (const int *)(dev + oNext);
    C.obs = (const int *)(dev + oObs);
    FtPose pose;
    memcpy(pose.m, Tcw, sizeof pose.m);
    int *rawBase = (int *)(dev + oRaw);
    FtLastRaw raw;
    raw.bestDist = rawBase; raw.bestIdx = rawBase + M; raw.bestDistR = rawBase + 2 * M; raw.bestIdxR = rawBase + 3 * M;
    int *resFinal = nullptr, passes = 0;
    rc = fixedPoint(ctx, st, M, N, (int *)(dev + oResA), (int *)(dev + oResB), (int *)(dev + oHead), (int *)(dev + oNext),
                    (int *)(dev + oChanged),
                    [&](int *res) { return ft_launch_search_last(st, DF, DL, C, pose, th, forward, backward, res, raw); },
                    &resFinal, &passes);
    if (rc != FT_OK) return rc;
    int *hRes = (int *)pin, *hRaw = (int *)(pin + 16 * (size_t)M + 64);
    FT_HIP(hipMemcpyAsync(hRes, resFinal, 16 * (size_t)M, hipMemcpyDeviceToHost, st));
    FT_HIP(hipMemcpyAsync(hRaw, rawBase, 16 * (size_t)M, hipMemcpyDeviceToHost, st));
    FT_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < 4; k++)
        if (outs[k]) memcpy(outs[k], hRaw + (size_t)k * M, 4 * (size_t)M);
    // replay in last-frame order; rotation histogram as in ORBmatcher.cc:1880-1896, 1942-1957, 1966-1987
    int nm = 0;
    std::vector<int> rotHist[FT_HISTO_LENGTH];
    const float factor = 1.0f / FT_HISTO_LENGTH;
    auto curAngle = [&](int idx) -> float {
        return (Cur->Nleft == -1) ? Cur->keys[idx].angle
               : (idx < Cur->Nleft) ? Cur->keys[idx].angle
                                    : Cur->keys_right[idx - Cur->Nleft].angle;
    };
    for (int i = 0; i < M; i++) {
        const int w2[2] = {hRes[4 * i], hRes[4 * i + 2]};
        for (int k = 0; k < 2; k++) {
            const int kp = w2[k];
            if (kp < 0) continue;
            Cur->holder_obs[kp] = L->observations[i];
            assign[kp] = i;
            nm++;
            if (check_orientation) {
                float rot = L->angle[i] - curAngle(kp);
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == FT_HISTO_LENGTH) bin = 0;
                if (bin >= 0 && bin < FT_HISTO_LENGTH) rotHist[bin].push_back(kp);  // the reference asserts
            }
        }
    }
    if (check_orientation) {
        // ComputeThreeMaxima, ORBmatcher.cc:2210-2251
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < FT_HISTO_LENGTH; i++) {
            const int s = (int)rotHist[i].size();
            if (s > max1) {
                max3 = max2; max2 = max1; max1 = s;
                ind3 = ind2; ind2 = ind1; ind1 = i;
            } else if (s > max2) {
                max3 = max2; max2 = s;
                ind3 = ind2; ind2 = i;
            } else if (s > max3) {
                max3 = s; ind3 = i;
            }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < FT_HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int kp : rotHist[i]) {
                    assign[kp] = -1;
                    Cur->holder_obs[kp] = -1;
                    nm--;
                }
    }
    if (n_matches) *n_matches = nm;
    ctx->addStat("search_last_frame.total", tAll.ms());
    ctx->addStat("search_last_frame.passes", passes);
    return FT_OK;
}

}  // extern "C"
