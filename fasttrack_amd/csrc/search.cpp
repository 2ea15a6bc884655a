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
    // the searches read a keypoint's octave back from four bits of a candidate key (make_key, kernels_search.hip)
    const int nL = F->Nleft == -1 ? F->N : F->Nleft, nR = F->Nleft == -1 ? 0 : F->N - F->Nleft;
    for (int i = 0; i < nL; i++) FT_REQUIRE(F->keys[i].octave >= 0 && F->keys[i].octave < F->nlevels, "keypoint octave outside [0, nlevels)");
    for (int i = 0; i < nR; i++)
        FT_REQUIRE(F->keys_right[i].octave >= 0 && F->keys_right[i].octave < F->nlevels, "right keypoint octave outside [0, nlevels)");
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

// Runs `search` passes until a pass changes nothing; the final results are on the host (through `download`) when it returns.
// A pass is ONE launch: the search kernel files every point's writes in the writer lists the next pass reads (three rotating
// head arrays: read / write / clear) and flags any change against the previous pass's results.  Passes are enqueued in
// bursts of FT_PASS_BURST without waiting in between; a pass first looks at the previous pass's flag and returns at once
// when the fixed point has been reached (it would reproduce its input), so the surplus passes of a burst cost an empty
// launch each while every avoided round trip (D2H of the flags + stream sync) costs ~100 us.  One memset (0xff: list heads
// = -1, flags = "unchanged") prepares a call.
#define FT_PASS_BURST_MAX 14  // flag slots per burst parity (16) and the 64-byte flag window of the pinned result area bound it
// passes per burst: with the candidate cache a pass is ~12 us and an early-exit pass ~5 us, a round trip to the host ~40 us,
// and a search needs 9 - 13 passes - one burst of 12 mostly does it (FT_PASS_BURST=<n> to experiment)
static int passBurst(const ft_context *ctx) { return std::min(std::max(ctx->tuning.pass_burst, 2), FT_PASS_BURST_MAX); }
// device buffers of the claim iteration: res 2 x 4 nPoints ints, head 3 x nKp directly followed by 16 flag ints (two burst
// parities x FT_PASS_BURST), next 2 x 4 nPoints
struct PassBufs {
    int *res, *head, *next;
    const int *obs;
    unsigned long long *cache;  // FT_CACHE_WORDS per point, or null (FT_SEARCH_CACHE=0)
};
// `download(res, flags, nFlagBytes)` enqueues ONE delivery kernel that writes the pass results `res`, whatever else the
// caller needs and the burst's flags into pinned host memory (hostFlags); it runs behind every burst, in front of the one
// stream synchronisation.
template <typename SearchFn, typename DownloadFn>
int fixedPoint(ft_context *ctx, hipStream_t st, int nPoints, int nKp, const PassBufs &B, FtClaims &C, SearchFn search,
               DownloadFn download, const int *hostFlags, int **resFinal, int *passes, int *burstHint = nullptr) {
    // Passes per burst.  Every surplus pass of a burst is an empty launch (4.5 us of dispatch for ~500 workgroups), every
    // burst that falls short a host round trip (~40 us).  A caller that searches frame after frame (ft_tracked_frame) hands in
    // the pass count of its previous search of the same kind: the first burst is that count + 1, later bursts are short.
    // Without a hint: option pass_burst (12) for every burst.
    const int burstMax = passBurst(ctx);
    int FT_PASS_BURST = burstHint && *burstHint > 0 ? std::min(std::max(*burstHint + 1, 4), FT_PASS_BURST_MAX) : burstMax;
    *resFinal = B.res;
    *passes = 0;
    if (nPoints <= 0) {
        int rc0 = download(B.res, nullptr, 0);
        if (rc0 != FT_OK) return rc0;
        FT_HIP(hipStreamSynchronize(st));
        return FT_OK;
    }
    const size_t K = ((size_t)std::max(nKp, 1) + 7) & ~(size_t)7;  // = passK(nKp)
    int *flags = B.head + 3 * K, *tab = flags + 32;  // (the table records are 32 bytes and 32-byte aligned: layoutPasses)
    const int fillWords = (int)(3 * K + 32 + 24 * K);
    {   // list heads = -1, flags = "unchanged" (-1), writer table empty (-1); candidate cache: ~0 in a slot's first word = "not built yet"
        const int rcf = ft_launch_fill_claims(st, B.head, fillWords, B.cache, 2 * nPoints, FT_CACHE_CAP + 1);
        if (rcf != FT_OK) return rcf;
    }
    C.cache = B.cache;
    int pass = 0, burst = 0;
    const int maxPasses = 2 * nPoints + 4 + burstMax;
    C.obs = B.obs;
    C.nKp = nKp;
    int *last = B.res;
    for (;; burst++) {
        int *fl = flags + 16 * (burst & 1), *flOther = flags + 16 * ((burst + 1) & 1);
        for (int b = 0; b < FT_PASS_BURST; b++, pass++) {
            C.firstPass = pass == 0;
            C.head = B.head + (size_t)(pass % 3) * K;
            C.headWrite = B.head + (size_t)((pass + 1) % 3) * K;
            C.headClear = B.head + (size_t)((pass + 2) % 3) * K;
            C.tab = tab + (size_t)(pass % 3) * 8 * K;
            C.tabWrite = tab + (size_t)((pass + 1) % 3) * 8 * K;
            C.tabClear = tab + (size_t)((pass + 2) % 3) * 8 * K;
            C.next = B.next + (size_t)((pass + 1) & 1) * 4 * nPoints;
            C.nextWrite = B.next + (size_t)(pass & 1) * 4 * nPoints;
            C.resPrev = B.res + (size_t)((pass + 1) & 1) * 4 * nPoints;
            C.flagCur = fl + b;
            C.flagPrev = b > 0 ? fl + b - 1 : nullptr;
            C.flagReset = flOther + b;
            last = B.res + (size_t)(pass & 1) * 4 * nPoints;
            const int rc = search(last);
            if (rc != FT_OK) return rc;
        }
        // after a converged burst both result buffers hold the fixed point (the last pass that ran reproduced its input)
        int rc = download(last, fl, sizeof(int) * FT_PASS_BURST);
        if (rc != FT_OK) return rc;
        FT_HIP(hipStreamSynchronize(st));
        const int *h = hostFlags;
        if (h[FT_PASS_BURST - 1] == -1) {  // the last pass of the burst changed nothing (or did not have to run)
            int ran = 0;
            while (ran < FT_PASS_BURST && h[ran] != -1) ran++;
            pass = pass - FT_PASS_BURST + std::min(ran + 1, FT_PASS_BURST);
            break;
        }
        if (pass >= maxPasses) {
            ft_set_error("projection search: claim resolution did not converge");
            return FT_ERR_HIP;
        }
        if (burstHint) FT_PASS_BURST = 4;  // the hint fell short: short bursts from here
    }
    *resFinal = last;
    *passes = pass;
    if (burstHint) *burstHint = pass;
    return FT_OK;
}

// arena space of the claim iteration for M points on a frame of N keypoints
struct PassLayout {
    size_t res, head, next, cache;
    bool haveCache;
};
// keypoint count of the claim buffers: a multiple of 8, so that the 32-byte table records behind 3 K heads + 32 flags are aligned
size_t passK(int N) { return ((size_t)std::max(N, 1) + 7) & ~(size_t)7; }
bool searchCacheOn(const ft_context *ctx) { return ctx->tuning.search_cache != 0; }
size_t searchCacheBytes(int M) { return 8 * (size_t)FT_CACHE_WORDS * (size_t)std::max(M, 1); }
// cacheInArena: the candidate cache (device only, 8 KB per point) lives at the end of the arena - the stand-alone searches,
// whose arena sizes the context's device scratch; a tracked frame owns a cache buffer of its own, so that its pinned mirror
// of the arena stays small
PassLayout layoutPasses(const ft_context *ctx, Arena &a, int M, int N, bool cacheInArena) {
    PassLayout L;
    L.res = a.take(32 * (size_t)M);
    L.head = a.take(4 * (3 * passK(N) + 32 + 24 * passK(N)));  // list heads, flags, writer table (fixedPoint)
    L.next = a.take(32 * (size_t)M);
    L.haveCache = cacheInArena && searchCacheOn(ctx);
    L.cache = L.haveCache ? a.take(searchCacheBytes(M)) : 0;
    return L;
}
PassBufs passBufs(const PassLayout &L, uint8_t *dev, const int *obs, unsigned long long *ownCache = nullptr) {
    PassBufs B;
    B.res = (int *)(dev + L.res);
    B.head = (int *)(dev + L.head);
    B.next = (int *)(dev + L.next);
    B.obs = obs;
    B.cache = L.haveCache ? (unsigned long long *)(dev + L.cache) : ownCache;
    return B;
}
// Frame::mGrid of a frame staged in the arena: CSR arrays behind the frame's own, built by one small launch
// cell starts per octave of both cameras (FT_MAX_LEVELS x 3073 ints each), then the entries as 16-byte search records and
// 32-byte descriptors
size_t gridIntBytes(int) { return (sizeof(int) * 2 * (size_t)FT_MAX_LEVELS * (FT_GRID_CELLS + 1) + 15) & ~(size_t)15; }
size_t gridBytes(int N) { return gridIntBytes(N) + 48 * (size_t)std::max(N, 1); }
size_t layoutGrid(Arena &a, int N) { return a.take(gridBytes(N)); }
int buildGrid(const ft_context *ctx, hipStream_t st, FtDevFrame &DF, int *grid) {
    if (!ctx->tuning.search_grid) return FT_OK;
    const int nL = DF.Nleft == -1 ? DF.N : DF.Nleft;
    int *startL = grid, *startR = grid + (size_t)FT_MAX_LEVELS * (FT_GRID_CELLS + 1);
    const bool two = DF.Nleft != -1;
    float4 *rec = (float4 *)((uint8_t *)grid + gridIntBytes(DF.N));
    uint8_t *gdesc = (uint8_t *)(rec + std::max(DF.N, 1));
    const int rc = ft_launch_build_grid(st, DF, startL, two ? startR : nullptr, rec, gdesc, two ? rec + nL : nullptr,
                                        two ? gdesc + (size_t)32 * nL : nullptr);
    if (rc != FT_OK) return rc;
    DF.gridStart[0] = startL;
    DF.gridStart[1] = two ? startR : nullptr;
    DF.gridRec[0] = rec;
    DF.gridDesc[0] = gdesc;
    DF.gridRec[1] = two ? rec + nL : nullptr;
    DF.gridDesc[1] = two ? gdesc + (size_t)32 * nL : nullptr;
    return FT_OK;
}

// Replays the writes of SearchByProjection(Frame, points) in map point order (ORBmatcher.cc:134-148 left,
// :203-214 right): res holds per point the keypoints written as (primary left, side left, primary right, side right).
int replayLocalWrites(const int *res, int M, const int *observations, int *holder, int *assign) {
    int nm = 0;
    for (int i = 0; i < M; i++) {
        const int obs = observations[i];
        const int order[4] = {res[4 * i], res[4 * i + 1], res[4 * i + 3], res[4 * i + 2]};  // primL sideL sideR primR
        for (int k = 0; k < 4; k++) {
            const int kp = order[k];
            if (kp < 0) continue;
            holder[kp] = obs;
            assign[kp] = i;
            nm++;
        }
    }
    return nm;
}

// Replays the writes of SearchByProjection(CurrentFrame, LastFrame) in last-frame order with the rotation histogram
// of ORBmatcher.cc:1880-1896, 1942-1957, 1966-1987 and ComputeThreeMaxima (:2210-2251).  curAngle(i) = angle of
// keypoint i of the current frame (left keypoints, then right).
template <typename AngleFn>
int replayLastFrameWrites(const int *res, int M, const ft_last_points *L, AngleFn curAngle, bool checkOrientation, int *holder,
                          int *assign) {
    int nm = 0;
    std::vector<int> rotHist[FT_HISTO_LENGTH];
    const float factor = 1.0f / FT_HISTO_LENGTH;
    for (int i = 0; i < M; i++) {
        const int w2[2] = {res[4 * i], res[4 * i + 2]};
        for (int k = 0; k < 2; k++) {
            const int kp = w2[k];
            if (kp < 0) continue;
            holder[kp] = L->observations[i];
            assign[kp] = i;
            nm++;
            if (checkOrientation) {
                float rot = L->angle[i] - curAngle(kp);
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == FT_HISTO_LENGTH) bin = 0;
                if (bin >= 0 && bin < FT_HISTO_LENGTH) rotHist[bin].push_back(kp);  // the reference asserts
            }
        }
    }
    if (checkOrientation) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < FT_HISTO_LENGTH; i++) {
            const int sz = (int)rotHist[i].size();
            if (sz > max1) {
                max3 = max2; max2 = max1; max1 = sz;
                ind3 = ind2; ind2 = ind1; ind1 = i;
            } else if (sz > max2) {
                max3 = max2; max2 = sz;
                ind3 = ind2; ind2 = i;
            } else if (sz > max3) {
                max3 = sz; ind3 = i;
            }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < FT_HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int kp : rotHist[i]) {
                    assign[kp] = -1;
                    holder[kp] = -1;
                    nm--;
                }
    }
    return nm;
}

// frame constants of a view (no arrays)
FtDevFrame devFrameConstants(const ft_frame_view *F) {
    FtDevFrame D;
    memset(&D, 0, sizeof D);
    D.N = F->N;
    D.Nleft = F->Nleft;
    D.mnMinX = F->mnMinX; D.mnMinY = F->mnMinY; D.mnMaxX = F->mnMaxX; D.mnMaxY = F->mnMaxY;
    D.invW = F->grid_inv_w; D.invH = F->grid_inv_h;
    D.mbf = F->mbf; D.mb = F->mb;
    D.camModel = F->cam_model;
    memcpy(D.cam, F->cam, sizeof D.cam);
    memcpy(D.Trl, F->Trl, sizeof D.Trl);
    for (int i = 0; i < F->nlevels && i < FT_MAX_LEVELS; i++) D.sf[i] = F->scale_factors ? F->scale_factors[i] : 1.f;
    D.nlevels = F->nlevels;
    return D;
}

// camera poses of isInFrustumChecks (Frame.cc:1312-1325); compiled without contraction, sums associated as Eigen does
FtFrustumPose frustumPose(const ft_frame_view *F, const ft_frame_pose *T) {
    FtFrustumPose P;
    memcpy(P.R[0], T->Rcw, sizeof P.R[0]);
    memcpy(P.t[0], T->tcw, sizeof P.t[0]);
    memcpy(P.twc[0], T->Ow, sizeof P.twc[0]);
    const float *Trl = F->Trl;
    auto sum3 = [](float e0, float e1, float e2) { return e0 + (e1 + e2); };  // Eigen's association (redux_novec_unroller)
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            P.R[1][3 * i + j] = sum3(Trl[4 * i] * T->Rcw[j], Trl[4 * i + 1] * T->Rcw[3 + j], Trl[4 * i + 2] * T->Rcw[6 + j]);
        P.t[1][i] = sum3(Trl[4 * i] * T->tcw[0], Trl[4 * i + 1] * T->tcw[1], Trl[4 * i + 2] * T->tcw[2]) + Trl[4 * i + 3];
        P.twc[1][i] = sum3(T->Rcw[i] * T->tlr[0], T->Rcw[3 + i] * T->tlr[1], T->Rcw[6 + i] * T->tlr[2]) + T->Ow[i];
    }
    return P;
}

FtFrustumPose frustumPose_fromDev(const FtDevFrame &DF, const ft_frame_pose *T) {
    ft_frame_view v;
    memset(&v, 0, sizeof v);
    memcpy(v.Trl, DF.Trl, sizeof v.Trl);
    return frustumPose(&v, T);
}

int checkMapPoints(const ft_map_points *P, bool forSearch) {
    FT_REQUIRE(P && P->M >= 0 && P->M < (1 << 22), "map point count out of range");
    FT_REQUIRE(P->M == 0 || (P->world_pos && P->normal && P->max_distance && P->min_distance), "map point arrays are null");
    FT_REQUIRE(!forSearch || P->M == 0 || (P->descriptors && P->observations), "map point descriptors / observations are null");
    return FT_OK;
}

struct FrustumLayout {
    size_t skip, pos, nrm, maxd, mind;                                      // inputs
    size_t inV, inVR, lvl, lvlR, vc, vcR, px, py, pxr, pyr, dep, depR, sskip, count;  // outputs
};

void layoutFrustum(int M, bool hasSkip, Arena &a, FrustumLayout &L, size_t *inputEnd) {
    const size_t m = (size_t)std::max(M, 1);
    L.skip = a.take(hasSkip ? m : 1);
    L.pos = a.take(12 * m);
    L.nrm = a.take(12 * m);
    L.maxd = a.take(4 * m);
    L.mind = a.take(4 * m);
    *inputEnd = a.off;
    L.inV = a.take(m); L.inVR = a.take(m);
    L.lvl = a.take(4 * m); L.lvlR = a.take(4 * m);
    L.vc = a.take(4 * m); L.vcR = a.take(4 * m);
    L.px = a.take(4 * m); L.py = a.take(4 * m); L.pxr = a.take(4 * m); L.pyr = a.take(4 * m);
    L.dep = a.take(4 * m); L.depR = a.take(4 * m);
    L.sskip = a.take(m);
    L.count = a.take(64);
}

void stageFrustum(const ft_map_points *P, const FrustumLayout &L, uint8_t *pin) {
    const size_t M = (size_t)P->M;
    if (!M) return;
    if (P->skip) memcpy(pin + L.skip, P->skip, M);
    memcpy(pin + L.pos, P->world_pos, 12 * M);
    memcpy(pin + L.nrm, P->normal, 12 * M);
    memcpy(pin + L.maxd, P->max_distance, 4 * M);
    memcpy(pin + L.mind, P->min_distance, 4 * M);
}

FtDevMapPoints devMapPoints(const ft_map_points *P, const FrustumLayout &L, uint8_t *dev) {
    FtDevMapPoints D;
    D.M = P->M;
    D.skip = P->skip ? dev + L.skip : nullptr;
    D.worldPos = (const float *)(dev + L.pos);
    D.normal = (const float *)(dev + L.nrm);
    D.maxDist = (const float *)(dev + L.maxd);
    D.minDist = (const float *)(dev + L.mind);
    return D;
}

FtFrustumOut devFrustumOut(const FrustumLayout &L, uint8_t *dev) {
    FtFrustumOut O;
    O.inView = dev + L.inV; O.inViewR = dev + L.inVR;
    O.level = (int *)(dev + L.lvl); O.levelR = (int *)(dev + L.lvlR);
    O.viewCos = (float *)(dev + L.vc); O.viewCosR = (float *)(dev + L.vcR);
    O.projX = (float *)(dev + L.px); O.projY = (float *)(dev + L.py);
    O.projXR = (float *)(dev + L.pxr); O.projYR = (float *)(dev + L.pyr);
    O.depth = (float *)(dev + L.dep); O.depthR = (float *)(dev + L.depR);
    O.searchSkip = dev + L.sskip;
    O.count = (int *)(dev + L.count);
    return O;
}

// D2H of the frustum fields the caller asked for (one contiguous copy of the output block, then scatter)
void unpackFrustum(int M, const FrustumLayout &L, size_t outBegin, uint8_t *pin, const ft_frustum_result *R, int *n_to_match);
int downloadFrustum(hipStream_t st, int M, const FrustumLayout &L, size_t outBegin, size_t outEnd, uint8_t *dev, uint8_t *pin,
                    const ft_frustum_result *R, int *n_to_match) {
    FT_HIP(hipMemcpyAsync(pin, dev + outBegin, outEnd - outBegin, hipMemcpyDeviceToHost, st));
    FT_HIP(hipStreamSynchronize(st));
    unpackFrustum(M, L, outBegin, pin, R, n_to_match);
    return FT_OK;
}
// the frustum fields of the output block [outBegin, ...) that sits at the start of `pin`
void unpackFrustum(int M, const FrustumLayout &L, size_t outBegin, uint8_t *pin, const ft_frustum_result *R, int *n_to_match) {
    auto at = [&](size_t off) { return pin + (off - outBegin); };
    if (n_to_match) *n_to_match = *(const int *)at(L.count);
    if (!R || !M) return;
    const size_t m = (size_t)M;
    if (R->in_view) memcpy(R->in_view, at(L.inV), m);
    if (R->in_view_r) memcpy(R->in_view_r, at(L.inVR), m);
    if (R->level) memcpy(R->level, at(L.lvl), 4 * m);
    if (R->level_r) memcpy(R->level_r, at(L.lvlR), 4 * m);
    if (R->view_cos) memcpy(R->view_cos, at(L.vc), 4 * m);
    if (R->view_cos_r) memcpy(R->view_cos_r, at(L.vcR), 4 * m);
    if (R->proj_x) memcpy(R->proj_x, at(L.px), 4 * m);
    if (R->proj_y) memcpy(R->proj_y, at(L.py), 4 * m);
    if (R->proj_xr) memcpy(R->proj_xr, at(L.pxr), 4 * m);
    if (R->proj_yr) memcpy(R->proj_yr, at(L.pyr), 4 * m);
    if (R->depth) memcpy(R->depth, at(L.dep), 4 * m);
    if (R->depth_r) memcpy(R->depth_r, at(L.depR), 4 * m);
}

}  // namespace

// Device-resident frame (ft_tracked_frame_*): owns (or borrows from a stereo front end) the keypoint / descriptor
// arrays in HBM; the scalar part of the frame, the host copy of the keypoints (angles for the rotation histogram)
// and the authoritative holder_obs live on the host and are cheap (a few KB per frame).
struct ft_tracked_frame {
    unsigned long long *d_cache = nullptr;  // candidate cache of the claim iteration (FtClaims::cache), maxPts points
    bool counted = false;  // registered with the context (ft_context::liveObjects)
    ft_context *ctx = nullptr;
    int maxKp = 0, maxPts = 0;
    // owned device storage
    ft_keypoint *d_keys = nullptr, *d_keysR = nullptr;
    uint8_t *d_desc = nullptr;
    float *d_uright = nullptr;
    int *d_holder = nullptr, *d_l2r = nullptr, *d_r2l = nullptr;
    int *d_grid = nullptr;                         // Frame::mGrid as CSR (both cameras), built when a frame is loaded
    uint8_t *d_work = nullptr, *h_work = nullptr;  // per-call arena (points, passes, outputs) and its pinned mirror
    int *h_holderUp = nullptr;                       // pinned source of the holder_obs uploads (see uploadHolder)
    uint8_t *h_frameUp = nullptr;                    // pinned staging of ft_tracked_frame_upload (all arrays of a frame)
    size_t frameUpBytes = 0;
    size_t workBytes = 0;
    // current frame
    bool loaded = false;
    FtDevFrame DF;
    std::vector<float> angles;  // angle of keypoint i (left then right)
    std::vector<int> holder;
    int passesLast = 0, passesLocal = 0;  // claim passes of the previous search of each kind: the next one's burst size (fixedPoint)
};

// holder_obs of the resident frame to the device, without a synchronisation: the pinned source belongs to the frame and is
// rewritten only by the next search on it, which is ordered behind this copy on the stream and synchronises the stream
// (fixedPoint) before the host gets here again
static int uploadHolder(ft_tracked_frame *tf, hipStream_t st) {
    const size_t bytes = sizeof(int) * tf->holder.size();
    if (!bytes) return FT_OK;
    memcpy(tf->h_holderUp, tf->holder.data(), bytes);
    FT_HIP(hipMemcpyAsync(tf->d_holder, tf->h_holderUp, bytes, hipMemcpyHostToDevice, st));
    return FT_OK;
}


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
    const PassLayout PL = layoutPasses(ctx, a, M, N, true);
    const size_t oGrid = layoutGrid(a, N);
    const size_t oRaw = a.take(40 * (size_t)M);
    const size_t total = a.off;
    const size_t outBytes = 16 * (size_t)M + 40 * (size_t)M + 64;
    rc = ft_ensure_scratch(ctx, total, std::max(inputBytes, outBytes));
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
    FtDevFrame DF = devFrame(F, FL, dev);
    rc = buildGrid(ctx, st, DF, (int *)(dev + oGrid));
    if (rc != FT_OK) return rc;
    FtDevLocalPoints DP;
    DP.M = M;
    DP.skip = dev + oSkip; DP.inView = dev + oIn; DP.inViewR = dev + oInR;
    DP.level = (const int *)(dev + oLevel); DP.levelR = (const int *)(dev + oLevelR);
    DP.viewCos = (const float *)(dev + oVc); DP.viewCosR = (const float *)(dev + oVcR);
    DP.projX = (const float *)(dev + oPx); DP.projY = (const float *)(dev + oPy);
    DP.projXR = (const float *)(dev + oPxr); DP.projYR = (const float *)(dev + oPyr);
    DP.desc = dev + oDesc;
    FtClaims C;
    C.obs = (const int *)(dev + oObs);
    int *rawBase = (int *)(dev + oRaw);
    FtLocalRaw raw;
    raw.bestDist = rawBase; raw.bestDist2 = rawBase + M; raw.bestLevel = rawBase + 2 * M; raw.bestLevel2 = rawBase + 3 * M;
    raw.bestIdx = rawBase + 4 * M; raw.bestDistR = rawBase + 5 * M; raw.bestDist2R = rawBase + 6 * M;
    raw.bestLevelR = rawBase + 7 * M; raw.bestLevel2R = rawBase + 8 * M; raw.bestIdxR = rawBase + 9 * M;
    int *resFinal = nullptr, passes = 0;
    rc = fixedPoint(ctx, st, M, N, passBufs(PL, dev, C.obs), C,
                    [&](int *res) { return ft_launch_search_local(st, DF, DP, C, th, nn_ratio, res, raw); },
                    [&](int *res, const int *fl, size_t flBytes) -> int {
                        return ft_launch_deliver_blocks(st, pin, res, 16 * (size_t)M, pin + 16 * (size_t)M + 64, rawBase, 40 * (size_t)M,
                                                        pin + 16 * (size_t)M, fl, flBytes);
                    },
                    (const int *)(pin + 16 * (size_t)M), &resFinal, &passes);
    if (rc != FT_OK) return rc;
    int *hRes = (int *)pin, *hRaw = (int *)(pin + 16 * (size_t)M + 64);
    for (int k = 0; k < 10; k++)
        if (outs[k]) memcpy(outs[k], hRaw + (size_t)k * M, 4 * (size_t)M);
    const int nm = replayLocalWrites(hRes, M, P->observations, F->holder_obs, assign);
    if (n_matches) *n_matches = nm;
    ctx->addStat("search_local_points.total", tAll.ms());
    ctx->addStat("search_local_points.passes", passes);
    return FT_OK;
}

namespace {
FtPose poseOfMatrix(const float *T) {
    FtPose p;
    memset(&p, 0, sizeof p);
    memcpy(p.m, T, sizeof p.m);
    return p;
}
int poseOfSe3(const ft_se3 *T, FtPose &p) {
    memset(&p, 0, sizeof p);
    const float n2 = T->q[0] * T->q[0] + T->q[1] * T->q[1] + T->q[2] * T->q[2] + T->q[3] * T->q[3];
    if (!(n2 > 0.99f && n2 < 1.01f)) {
        ft_set_error("ft_se3: q is not a unit quaternion (x, y, z, w)");
        return FT_ERR_INVALID;
    }
    p.m[0] = p.m[5] = p.m[10] = 1.f;
    p.m[3] = T->t[0]; p.m[7] = T->t[1]; p.m[11] = T->t[2];
    memcpy(p.q, T->q, sizeof p.q);
    p.quat = 1;
    return FT_OK;
}
// the right camera's pose of a frame in the Sophus form, in place of the matrix of its view
void setTrl(FtDevFrame &DF, const FtPose &trl) {
    memcpy(DF.Trl, trl.m, sizeof DF.Trl);
    memcpy(DF.TrlQ, trl.q, sizeof DF.TrlQ);
    DF.trlQuat = trl.quat;
}

int searchLastFrame(ft_context *ctx, ft_frame_view *Cur, const ft_last_points *L, const FtPose &pose, const FtPose *trl, float th,
                    int forward, int backward, int check_orientation, int *assign, int *n_matches, int *best_dist, int *best_idx,
                    int *best_dist_r, int *best_idx_r) {
    FT_REQUIRE(ctx && L && assign, "ft_search_last_frame: null argument");
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
    const PassLayout PL = layoutPasses(ctx, a, M, N, true);
    const size_t oGrid = layoutGrid(a, N);
    const size_t oRaw = a.take(16 * (size_t)M);
    const size_t total = a.off;
    const size_t outBytes = 32 * (size_t)M + 64;
    rc = ft_ensure_scratch(ctx, total, std::max(inputBytes, outBytes));
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
    FtDevFrame DF = devFrame(Cur, FL, dev);
    if (trl) setTrl(DF, *trl);
    rc = buildGrid(ctx, st, DF, (int *)(dev + oGrid));
    if (rc != FT_OK) return rc;
    FtDevLastPoints DL;
    DL.N = M;
    DL.valid = dev + oValid;
    DL.worldPos = (const float *)(dev + oPos);
    DL.desc = dev + oDesc;
    DL.octave = (const int *)(dev + oOct);
    FtClaims C;
    C.obs = (const int *)(dev + oObs);
    int *rawBase = (int *)(dev + oRaw);
    FtLastRaw raw;
    raw.bestDist = rawBase; raw.bestIdx = rawBase + M; raw.bestDistR = rawBase + 2 * M; raw.bestIdxR = rawBase + 3 * M;
    int *resFinal = nullptr, passes = 0;
    rc = fixedPoint(ctx, st, M, N, passBufs(PL, dev, C.obs), C,
                    [&](int *res) { return ft_launch_search_last(st, DF, DL, C, pose, th, forward, backward, res, raw); },
                    [&](int *res, const int *fl, size_t flBytes) -> int {
                        return ft_launch_deliver_blocks(st, pin, res, 16 * (size_t)M, pin + 16 * (size_t)M + 64, rawBase, 16 * (size_t)M,
                                                        pin + 16 * (size_t)M, fl, flBytes);
                    },
                    (const int *)(pin + 16 * (size_t)M), &resFinal, &passes);
    if (rc != FT_OK) return rc;
    int *hRes = (int *)pin, *hRaw = (int *)(pin + 16 * (size_t)M + 64);
    for (int k = 0; k < 4; k++)
        if (outs[k]) memcpy(outs[k], hRaw + (size_t)k * M, 4 * (size_t)M);
    auto curAngle = [&](int idx) -> float {
        return (Cur->Nleft == -1) ? Cur->keys[idx].angle
               : (idx < Cur->Nleft) ? Cur->keys[idx].angle
                                    : Cur->keys_right[idx - Cur->Nleft].angle;
    };
    const int nm = replayLastFrameWrites(hRes, M, L, curAngle, check_orientation != 0, Cur->holder_obs, assign);
    if (n_matches) *n_matches = nm;
    ctx->addStat("search_last_frame.total", tAll.ms());
    ctx->addStat("search_last_frame.passes", passes);
    return FT_OK;
}
}  // namespace

int ft_search_last_frame(ft_context *ctx, ft_frame_view *Cur, const ft_last_points *L, const float *Tcw, float th,
                         int forward, int backward, int check_orientation, int *assign, int *n_matches,
                         int *best_dist, int *best_idx, int *best_dist_r, int *best_idx_r) {
    FT_REQUIRE(Tcw, "ft_search_last_frame: null pose");
    return searchLastFrame(ctx, Cur, L, poseOfMatrix(Tcw), nullptr, th, forward, backward, check_orientation, assign, n_matches,
                           best_dist, best_idx, best_dist_r, best_idx_r);
}

int ft_search_last_frame_se3(ft_context *ctx, ft_frame_view *Cur, const ft_last_points *L, const ft_se3 *Tcw, const ft_se3 *Trl,
                             float th, int forward, int backward, int check_orientation, int *assign, int *n_matches,
                             int *best_dist, int *best_idx, int *best_dist_r, int *best_idx_r) {
    FT_REQUIRE(Tcw && Cur, "ft_search_last_frame_se3: null argument");
    FT_REQUIRE(Trl || Cur->Nleft == -1, "ft_search_last_frame_se3: a two-camera frame needs Trl");
    FtPose pose, trl;
    int rc = poseOfSe3(Tcw, pose);
    if (rc == FT_OK && Trl) rc = poseOfSe3(Trl, trl);
    if (rc != FT_OK) return rc;
    return searchLastFrame(ctx, Cur, L, pose, Trl ? &trl : nullptr, th, forward, backward, check_orientation, assign, n_matches,
                           best_dist, best_idx, best_dist_r, best_idx_r);
}

int ft_features_in_area(ft_context *ctx, const ft_frame_view *F, int nq, const float *x, const float *y, const float *r,
                        const int *min_level, const int *max_level, const uint8_t *right, int *indices, int capacity,
                        int *counts) {
    FT_REQUIRE(ctx && counts && nq >= 0 && capacity >= 0, "ft_features_in_area: bad argument");
    FT_REQUIRE(nq == 0 || (x && y && r && min_level && max_level), "ft_features_in_area: null query arrays");
    FT_REQUIRE(capacity == 0 || indices, "ft_features_in_area: null index array");
    FT_REQUIRE(nq < (1 << 22), "ft_features_in_area: too many queries");
    int rc = checkFrame(F);
    if (rc != FT_OK) return rc;
    FT_REQUIRE(F->N < (1 << 20), "ft_features_in_area: frame too large for the hit keys");
    if (nq == 0) return FT_OK;
    rc = ft_set_device(ctx);
    if (rc != FT_OK) return rc;
    std::lock_guard<std::mutex> lk(ctx->matchMutex);
    Arena a;
    FrameLayout FL;
    layoutFrame(F, a, FL);
    const size_t Q = (size_t)nq;
    const size_t oX = a.take(4 * Q), oY = a.take(4 * Q), oR = a.take(4 * Q), oMin = a.take(4 * Q), oMax = a.take(4 * Q),
                 oRight = a.take(Q);
    const size_t inputBytes = a.off;
    const size_t oCount = a.take(4 * Q), oOff = a.take(4 * Q);
    const size_t fixedBytes = a.off;
    rc = ft_ensure_scratch(ctx, fixedBytes, fixedBytes);
    if (rc != FT_OK) return rc;
    uint8_t *pin = (uint8_t *)ctx->scratchPin, *dev = (uint8_t *)ctx->scratchDev;
    stageFrame(F, FL, pin);
    memcpy(pin + oX, x, 4 * Q);
    memcpy(pin + oY, y, 4 * Q);
    memcpy(pin + oR, r, 4 * Q);
    memcpy(pin + oMin, min_level, 4 * Q);
    memcpy(pin + oMax, max_level, 4 * Q);
    if (right) memcpy(pin + oRight, right, Q);
    hipStream_t st = ctx->stream;
    FT_HIP(hipMemcpyAsync(dev, pin, inputBytes, hipMemcpyHostToDevice, st));
    FtDevFrame DF = devFrame(F, FL, dev);
    auto launch = [&](const int *offsets, unsigned *keys) {
        return ft_launch_features_in_area(st, DF, nq, (const float *)(dev + oX), (const float *)(dev + oY),
                                          (const float *)(dev + oR), (const int *)(dev + oMin), (const int *)(dev + oMax),
                                          right ? dev + oRight : nullptr, offsets, keys, (int *)(dev + oCount));
    };
    // pass 1 counts the hits of every query, pass 2 writes them at the query's offset (no capacity inside)
    rc = launch(nullptr, nullptr);
    if (rc != FT_OK) return rc;
    std::vector<int> hCount(nq), hOff(nq);
    FT_HIP(hipMemcpyAsync(hCount.data(), dev + oCount, 4 * Q, hipMemcpyDeviceToHost, st));
    FT_HIP(hipStreamSynchronize(st));
    size_t totalHits = 0;
    for (int q = 0; q < nq; q++) {
        hOff[q] = (int)totalHits;
        totalHits += (size_t)hCount[q];
        counts[q] = hCount[q];
    }
    if (totalHits == 0) return FT_OK;
    FT_REQUIRE(totalHits < (1u << 30), "ft_features_in_area: too many hits");
    const size_t oKeys = a.take(4 * totalHits);
    // growing the scratch buffer reallocates it: the frame and the queries are staged again in that case
    const void *devBefore = ctx->scratchDev;
    rc = ft_ensure_scratch(ctx, a.off, std::max(fixedBytes, 4 * totalHits));
    if (rc != FT_OK) return rc;
    pin = (uint8_t *)ctx->scratchPin;
    dev = (uint8_t *)ctx->scratchDev;
    if (ctx->scratchDev != devBefore) {
        stageFrame(F, FL, pin);
        memcpy(pin + oX, x, 4 * Q);
        memcpy(pin + oY, y, 4 * Q);
        memcpy(pin + oR, r, 4 * Q);
        memcpy(pin + oMin, min_level, 4 * Q);
        memcpy(pin + oMax, max_level, 4 * Q);
        if (right) memcpy(pin + oRight, right, Q);
        FT_HIP(hipMemcpyAsync(dev, pin, inputBytes, hipMemcpyHostToDevice, st));
        FT_HIP(hipStreamSynchronize(st));
        DF = devFrame(F, FL, dev);
    }
    FT_HIP(hipMemcpyAsync(dev + oOff, hOff.data(), 4 * Q, hipMemcpyHostToDevice, st));
    rc = launch((const int *)(dev + oOff), (unsigned *)(dev + oKeys));
    if (rc != FT_OK) return rc;
    unsigned *hKeys = (unsigned *)pin;
    FT_HIP(hipMemcpyAsync(hKeys, dev + oKeys, 4 * totalHits, hipMemcpyDeviceToHost, st));
    FT_HIP(hipStreamSynchronize(st));
    for (int q = 0; q < nq; q++) {
        unsigned *k = hKeys + hOff[q];
        std::sort(k, k + hCount[q]);  // (cell column, cell row, index): the order of the nested loops of Frame.cc:718-744
        const int m = std::min(hCount[q], capacity);
        for (int i = 0; i < m; i++) indices[(size_t)q * capacity + i] = (int)(k[i] & 0xfffffu);
    }
    return FT_OK;
}

int ft_is_in_frustum(ft_context *ctx, const ft_frame_view *F, const ft_frame_pose *pose, const ft_map_points *P,
                     float viewing_cos_limit, float log_scale_factor, const ft_frustum_result *out, int *n_to_match) {
    FT_REQUIRE(ctx && F && pose, "ft_is_in_frustum: null argument");
    FT_REQUIRE(F->nlevels >= 1 && F->nlevels <= FT_MAX_LEVELS, "ft_is_in_frustum: nlevels out of range");
    FT_REQUIRE(F->cam_model == 0 || F->cam_model == 1, "unknown camera model");
    int rc = checkMapPoints(P, false);
    if (rc != FT_OK) return rc;
    if (n_to_match) *n_to_match = 0;
    if (P->M == 0) return FT_OK;
    rc = ft_set_device(ctx);
    if (rc != FT_OK) return rc;
    std::lock_guard<std::mutex> lk(ctx->matchMutex);
    FtTimer tAll;
    Arena a;
    FrustumLayout L;
    size_t inputEnd = 0;
    layoutFrustum(P->M, P->skip != nullptr, a, L, &inputEnd);
    rc = ft_ensure_scratch(ctx, a.off, a.off);
    if (rc != FT_OK) return rc;
    uint8_t *pin = (uint8_t *)ctx->scratchPin, *dev = (uint8_t *)ctx->scratchDev;
    stageFrustum(P, L, pin);
    hipStream_t st = ctx->stream;
    FT_HIP(hipMemcpyAsync(dev, pin, inputEnd, hipMemcpyHostToDevice, st));
    const FtDevFrame DF = devFrameConstants(F);
    rc = ft_launch_frustum(st, DF, frustumPose(F, pose), devMapPoints(P, L, dev), viewing_cos_limit, log_scale_factor, 0, 0.f,
                           devFrustumOut(L, dev));
    if (rc != FT_OK) return rc;
    rc = downloadFrustum(st, P->M, L, inputEnd, a.off, dev, pin, out, n_to_match);
    ctx->addStat("is_in_frustum.total", tAll.ms());
    return rc;
}

int ft_tracked_frame_create(ft_context *ctx, int max_keypoints, int max_points, ft_tracked_frame **out) {
    FT_REQUIRE(ctx && out && max_keypoints > 0 && max_points > 0, "ft_tracked_frame_create: bad argument");
    FT_REQUIRE(max_keypoints < (1 << 24) && max_points < (1 << 22), "ft_tracked_frame_create: capacity out of range");
    int rc = ft_set_device(ctx);
    if (rc != FT_OK) return rc;
    ft_tracked_frame *tf = new ft_tracked_frame();
    tf->ctx = ctx;
    tf->maxKp = max_keypoints;
    tf->maxPts = max_points;
    const size_t K = (size_t)max_keypoints, M = (size_t)max_points;
    // arena of one call: map points (<= 72 B) + frustum outputs (<= 48 B) + passes / raw outputs (<= 104 B) per point
    tf->workBytes = 320 * M + 112 * K + 16384;  // (108 K: list heads and writer table of the claim iteration)
    hipError_t e = hipMalloc((void **)&tf->d_keys, sizeof(ft_keypoint) * K);
    if (e == hipSuccess) e = hipMalloc((void **)&tf->d_keysR, sizeof(ft_keypoint) * K);
    if (e == hipSuccess) e = hipMalloc((void **)&tf->d_desc, 32 * K);
    if (e == hipSuccess) e = hipMalloc((void **)&tf->d_uright, sizeof(float) * K);
    if (e == hipSuccess) e = hipMalloc((void **)&tf->d_holder, sizeof(int) * K);
    if (e == hipSuccess) e = hipMalloc((void **)&tf->d_l2r, sizeof(int) * K);
    if (e == hipSuccess) e = hipMalloc((void **)&tf->d_r2l, sizeof(int) * K);
    if (e == hipSuccess) e = hipMalloc((void **)&tf->d_work, tf->workBytes);
    if (e == hipSuccess) e = hipMalloc((void **)&tf->d_grid, gridBytes((int)K));
    if (e == hipSuccess && searchCacheOn(ctx)) e = hipMalloc((void **)&tf->d_cache, searchCacheBytes(max_points));
    if (e == hipSuccess) e = hipHostMalloc((void **)&tf->h_work, tf->workBytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&tf->h_holderUp, sizeof(int) * K, hipHostMallocDefault);
    tf->frameUpBytes = (2 * sizeof(ft_keypoint) + 32 + 4 * sizeof(int)) * K + 8 * 64;
    if (e == hipSuccess) e = hipHostMalloc((void **)&tf->h_frameUp, tf->frameUpBytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        ft_tracked_frame_destroy(tf);
        return ft_hip_fail(e, "ft_tracked_frame_create", __FILE__, __LINE__);
    }
    tf->counted = true;
    ctx->liveObjects++;
    *out = tf;
    return FT_OK;
}

int ft_tracked_frame_destroy(ft_tracked_frame *tf) {
    if (!tf) return FT_OK;
    ft_set_device(tf->ctx);
    hipStreamSynchronize(tf->ctx->stream);
    hipFree(tf->d_keys); hipFree(tf->d_keysR); hipFree(tf->d_desc); hipFree(tf->d_uright);
    hipFree(tf->d_holder); hipFree(tf->d_l2r); hipFree(tf->d_r2l); hipFree(tf->d_work); hipFree(tf->d_grid);
    if (tf->d_cache) hipFree(tf->d_cache);
    if (tf->h_work) hipHostFree(tf->h_work);
    if (tf->h_holderUp) hipHostFree(tf->h_holderUp);
    if (tf->h_frameUp) hipHostFree(tf->h_frameUp);
    if (tf->counted) tf->ctx->liveObjects--;
    delete tf;
    return FT_OK;
}

int ft_tracked_frame_upload(ft_tracked_frame *tf, const ft_frame_view *F) {
    FT_REQUIRE(tf, "null tracked frame");
    int rc = checkFrame(F);
    if (rc != FT_OK) return rc;
    FT_REQUIRE(F->N <= tf->maxKp, "ft_tracked_frame_upload: more keypoints than the frame was created for");
    rc = ft_set_device(tf->ctx);
    if (rc != FT_OK) return rc;
    std::lock_guard<std::mutex> lk(tf->ctx->matchMutex);
    hipStream_t st = tf->ctx->stream;
    const int nL = F->Nleft == -1 ? F->N : F->Nleft, nR = F->Nleft == -1 ? 0 : F->N - F->Nleft;
    // The caller's arrays are pageable as a rule: handed to hipMemcpyAsync as they are, each of the seven copies is staged by
    // the runtime and waited for (70 - 85 us per frame).  They are packed into the frame's own pinned buffer instead (one
    // pass of host memcpy) and go up from there as plain asynchronous copies; nothing is waited for here - the searches that
    // follow are ordered behind the copies on the stream, and the next upload waits for the stream before it repacks.
    FT_HIP(hipStreamSynchronize(st));
    {
        Arena up;
        auto put = [&](void *dst, const void *src, size_t bytes) -> int {
            if (!bytes) return FT_OK;
            const size_t o = up.take(bytes);
            memcpy(tf->h_frameUp + o, src, bytes);
            FT_HIP(hipMemcpyAsync(dst, tf->h_frameUp + o, bytes, hipMemcpyHostToDevice, st));
            return FT_OK;
        };
        if ((rc = put(tf->d_keys, F->keys, sizeof(ft_keypoint) * nL)) != FT_OK) return rc;
        if ((rc = put(tf->d_keysR, F->keys_right, sizeof(ft_keypoint) * nR)) != FT_OK) return rc;
        if ((rc = put(tf->d_desc, F->descriptors, (size_t)32 * F->N)) != FT_OK) return rc;
        if (F->uright && (rc = put(tf->d_uright, F->uright, sizeof(float) * F->N)) != FT_OK) return rc;
        if (F->Nleft != -1) {
            if ((rc = put(tf->d_l2r, F->left_to_right, sizeof(int) * nL)) != FT_OK) return rc;
            if ((rc = put(tf->d_r2l, F->right_to_left, sizeof(int) * nR)) != FT_OK) return rc;
        }
        if ((rc = put(tf->d_holder, F->holder_obs, sizeof(int) * F->N)) != FT_OK) return rc;
    }
    tf->DF = devFrameConstants(F);
    tf->DF.keys = tf->d_keys;
    tf->DF.keysR = tf->d_keysR;
    tf->DF.desc = tf->d_desc;
    tf->DF.uright = F->uright ? tf->d_uright : nullptr;
    tf->DF.holderObs = tf->d_holder;
    tf->DF.l2r = F->Nleft != -1 ? tf->d_l2r : nullptr;
    tf->DF.r2l = F->Nleft != -1 ? tf->d_r2l : nullptr;
    rc = buildGrid(tf->ctx, st, tf->DF, tf->d_grid);  // the grid of the frame, once: both searches look up their windows in it
    if (rc != FT_OK) return rc;
    tf->angles.resize(F->N);
    for (int i = 0; i < nL; i++) tf->angles[i] = F->keys[i].angle;
    for (int i = 0; i < nR; i++) tf->angles[nL + i] = F->keys_right[i].angle;
    tf->holder.assign(F->holder_obs, F->holder_obs + F->N);
    tf->loaded = true;
    return FT_OK;
}

int ft_tracked_frame_bind_stereo(ft_tracked_frame *tf, ft_stereo_frontend *fe, int slot, const ft_frame_view *meta) {
    FT_REQUIRE(tf && fe && meta, "ft_tracked_frame_bind_stereo: null argument");
    FT_REQUIRE(tf->ctx == fe->ctx, "tracked frame and front end belong to different contexts");
    FT_REQUIRE(!fe->pending.active, "ft_tracked_frame_bind_stereo: the front end has a submitted batch that was not waited for");
    ft_extractor *L = fe->exL;
    FT_REQUIRE(slot >= 0 && slot < fe->maxBatch, "ft_tracked_frame_bind_stereo: slot out of range");
    const int N = L->h_nSel[slot];
    FT_REQUIRE(meta->Nleft == -1, "ft_tracked_frame_bind_stereo: the stereo front end produces rectified frames (Nleft == -1)");
    FT_REQUIRE(meta->N == N, "ft_tracked_frame_bind_stereo: meta->N differs from the keypoint count of the slot");
    FT_REQUIRE(N <= tf->maxKp, "ft_tracked_frame_bind_stereo: more keypoints than the frame was created for");
    FT_REQUIRE(meta->scale_factors && meta->nlevels >= 1 && meta->nlevels <= FT_MAX_LEVELS, "scale factors missing");
    FT_REQUIRE(N == 0 || meta->keys, "ft_tracked_frame_bind_stereo: meta->keys (host copy of the keypoints) is null");
    int rc = ft_set_device(tf->ctx);
    if (rc != FT_OK) return rc;
    std::lock_guard<std::mutex> lk(tf->ctx->matchMutex);
    tf->DF = devFrameConstants(meta);
    tf->DF.keys = L->d_keys + (size_t)slot * L->geom.maxKp;
    tf->DF.keysR = tf->d_keysR;
    tf->DF.desc = L->d_desc + (size_t)slot * L->geom.maxKp * 32;
    tf->DF.uright = fe->d_uright + (size_t)slot * fe->capacity;
    tf->DF.holderObs = tf->d_holder;
    tf->DF.l2r = nullptr;
    tf->DF.r2l = nullptr;
    tf->holder.assign(N, -1);
    if (meta->holder_obs) tf->holder.assign(meta->holder_obs, meta->holder_obs + N);
    tf->angles.resize(N);
    for (int i = 0; i < N; i++) tf->angles[i] = meta->keys[i].angle;
    if (N) FT_HIP(hipMemcpy(tf->d_holder, tf->holder.data(), sizeof(int) * N, hipMemcpyHostToDevice));
    rc = buildGrid(tf->ctx, tf->ctx->stream, tf->DF, tf->d_grid);  // ordered in front of the searches on the context stream
    if (rc != FT_OK) return rc;
    tf->loaded = true;
    return FT_OK;
}

int ft_tracked_frame_holder_obs(ft_tracked_frame *tf, int *holder_obs) {
    FT_REQUIRE(tf && tf->loaded && holder_obs, "ft_tracked_frame_holder_obs: no frame loaded");
    if (!tf->holder.empty()) memcpy(holder_obs, tf->holder.data(), sizeof(int) * tf->holder.size());
    return FT_OK;
}

namespace {
int trackedSearchLastFrame(ft_tracked_frame *tf, const ft_last_points *L, const FtPose &pose, const FtPose *trl, float th,
                           int forward, int backward, int check_orientation, int *assign, int *n_matches) {
    FT_REQUIRE(tf && tf->loaded && L && assign, "ft_tracked_frame_search_last_frame: null argument / no frame loaded");
    ft_context *ctx = tf->ctx;
    const int M = L->N, N = tf->DF.N;
    FT_REQUIRE(M >= 0 && M <= tf->maxPts, "last-frame point count beyond the frame's capacity");
    FT_REQUIRE(M == 0 || (L->valid && L->world_pos && L->descriptors && L->observations && L->octave && L->angle),
               "last-frame arrays are null");
    for (int i = 0; i < N; i++) assign[i] = -1;
    if (n_matches) *n_matches = 0;
    if (M == 0 || N == 0) return FT_OK;
    for (int i = 0; i < M; i++)
        FT_REQUIRE(!L->valid[i] || (L->octave[i] >= 0 && L->octave[i] < tf->DF.nlevels), "last-frame octave out of range");
    int rc = ft_set_device(ctx);
    if (rc != FT_OK) return rc;
    std::lock_guard<std::mutex> lk(ctx->matchMutex);
    FtTimer tAll;
    Arena a;
    const size_t oValid = a.take(M), oPos = a.take(12 * (size_t)M), oDesc = a.take(32 * (size_t)M),
                 oObs = a.take(4 * (size_t)M), oOct = a.take(4 * (size_t)M);
    const size_t inputBytes = a.off;
    const PassLayout PL = layoutPasses(tf->ctx, a, M, N, false);
    FT_REQUIRE(a.off <= tf->workBytes, "tracked frame work arena too small");
    uint8_t *pin = tf->h_work, *dev = tf->d_work;
    memcpy(pin + oValid, L->valid, M);
    memcpy(pin + oPos, L->world_pos, 12 * (size_t)M);
    memcpy(pin + oDesc, L->descriptors, 32 * (size_t)M);
    memcpy(pin + oObs, L->observations, 4 * (size_t)M);
    memcpy(pin + oOct, L->octave, 4 * (size_t)M);
    hipStream_t st = ctx->stream;
    FT_HIP(hipMemcpyAsync(dev, pin, inputBytes, hipMemcpyHostToDevice, st));
    FtDevLastPoints DL;
    DL.N = M;
    DL.valid = dev + oValid;
    DL.worldPos = (const float *)(dev + oPos);
    DL.desc = dev + oDesc;
    DL.octave = (const int *)(dev + oOct);
    FtClaims C;
    C.obs = (const int *)(dev + oObs);
    FtLastRaw raw = {nullptr, nullptr, nullptr, nullptr};
    int *resFinal = nullptr, passes = 0;
    FtDevFrame DF = tf->DF;
    if (trl) setTrl(DF, *trl);
    rc = fixedPoint(ctx, st, M, N, passBufs(PL, dev, C.obs, tf->d_cache), C,
                    [&](int *res) { return ft_launch_search_last(st, DF, DL, C, pose, th, forward, backward, res, raw); },
                    [&](int *res, const int *fl, size_t flBytes) -> int {
                        return ft_launch_deliver_blocks(st, pin, res, 16 * (size_t)M, pin + 16 * (size_t)M, fl, flBytes, nullptr, nullptr, 0);
                    },
                    (const int *)(pin + 16 * (size_t)M), &resFinal, &passes, &tf->passesLast);
    if (rc != FT_OK) return rc;
    int *hRes = (int *)pin;
    const int nm = replayLastFrameWrites(hRes, M, L, [&](int idx) { return tf->angles[idx]; }, check_orientation != 0,
                                         tf->holder.data(), assign);
    // the occupancy the next search sees: uploaded from a pinned buffer of the frame's own, so that nothing has to wait
    // for the copy (the next call on this frame is ordered behind it on the stream and synchronises before it returns)
    rc = uploadHolder(tf, st);
    if (rc != FT_OK) return rc;
    if (n_matches) *n_matches = nm;
    ctx->addStat("tracked.search_last_frame.total", tAll.ms());
    ctx->addStat("tracked.search_last_frame.passes", passes);
    return FT_OK;
}
}  // namespace

int ft_tracked_frame_search_last_frame(ft_tracked_frame *tf, const ft_last_points *L, const float *Tcw, float th,
                                       int forward, int backward, int check_orientation, int *assign, int *n_matches) {
    FT_REQUIRE(Tcw, "ft_tracked_frame_search_last_frame: null pose");
    return trackedSearchLastFrame(tf, L, poseOfMatrix(Tcw), nullptr, th, forward, backward, check_orientation, assign, n_matches);
}

int ft_tracked_frame_search_last_frame_se3(ft_tracked_frame *tf, const ft_last_points *L, const ft_se3 *Tcw, const ft_se3 *Trl,
                                           float th, int forward, int backward, int check_orientation, int *assign, int *n_matches) {
    FT_REQUIRE(tf && tf->loaded && Tcw, "ft_tracked_frame_search_last_frame_se3: null argument / no frame loaded");
    FT_REQUIRE(Trl || tf->DF.Nleft == -1, "ft_tracked_frame_search_last_frame_se3: a two-camera frame needs Trl");
    FtPose pose, trl;
    int rc = poseOfSe3(Tcw, pose);
    if (rc == FT_OK && Trl) rc = poseOfSe3(Trl, trl);
    if (rc != FT_OK) return rc;
    return trackedSearchLastFrame(tf, L, pose, Trl ? &trl : nullptr, th, forward, backward, check_orientation, assign, n_matches);
}

int ft_tracked_frame_track_local_map(ft_tracked_frame *tf, const ft_frame_pose *pose, const ft_map_points *P,
                                     float viewing_cos_limit, float log_scale_factor, float th, float nn_ratio,
                                     int far_points, float th_far_points, const ft_frustum_result *frustum, int *n_to_match,
                                     int *assign, int *n_matches) {
    FT_REQUIRE(tf && tf->loaded && pose && assign, "ft_tracked_frame_track_local_map: null argument / no frame loaded");
    int rc = checkMapPoints(P, true);
    if (rc != FT_OK) return rc;
    ft_context *ctx = tf->ctx;
    const int M = P->M, N = tf->DF.N;
    FT_REQUIRE(M <= tf->maxPts, "map point count beyond the frame's capacity");
    for (int i = 0; i < N; i++) assign[i] = -1;
    if (n_matches) *n_matches = 0;
    if (n_to_match) *n_to_match = 0;
    if (M == 0) return FT_OK;
    rc = ft_set_device(ctx);
    if (rc != FT_OK) return rc;
    std::lock_guard<std::mutex> lk(ctx->matchMutex);
    FtTimer tAll;
    Arena a;
    FrustumLayout FL;
    size_t fInputEnd = 0;
    layoutFrustum(M, P->skip != nullptr, a, FL, &fInputEnd);
    const size_t fOutEnd = a.off;
    const size_t oDesc = a.take(32 * (size_t)M), oObs = a.take(4 * (size_t)M);
    const PassLayout PL = layoutPasses(tf->ctx, a, M, N, false);
    FT_REQUIRE(a.off <= tf->workBytes, "tracked frame work arena too small");
    uint8_t *pin = tf->h_work, *dev = tf->d_work;
    stageFrustum(P, FL, pin);
    memcpy(pin + oDesc, P->descriptors, 32 * (size_t)M);
    memcpy(pin + oObs, P->observations, 4 * (size_t)M);
    hipStream_t st = ctx->stream;
    FT_HIP(hipMemcpyAsync(dev, pin, fInputEnd, hipMemcpyHostToDevice, st));
    FT_HIP(hipMemcpyAsync(dev + oDesc, pin + oDesc, oObs + 4 * (size_t)M - oDesc, hipMemcpyHostToDevice, st));
    const FtDevFrame DF = tf->DF;
    const FtFrustumOut FO = devFrustumOut(FL, dev);
    rc = ft_launch_frustum(st, DF, frustumPose_fromDev(DF, pose), devMapPoints(P, FL, dev), viewing_cos_limit, log_scale_factor,
                           far_points, th_far_points, FO);
    if (rc != FT_OK) return rc;
    int nm = 0, passes = 0;
    if (N > 0) {
        // the frustum fields are the search's inputs where they are: no host round trip in between
        FtDevLocalPoints DP;
        DP.M = M;
        DP.skip = FO.searchSkip; DP.inView = FO.inView; DP.inViewR = FO.inViewR;
        DP.level = FO.level; DP.levelR = FO.levelR;
        DP.viewCos = FO.viewCos; DP.viewCosR = FO.viewCosR;
        DP.projX = FO.projX; DP.projY = FO.projY; DP.projXR = FO.projXR; DP.projYR = FO.projYR;
        DP.desc = dev + oDesc;
        FtClaims C;
        C.obs = (const int *)(dev + oObs);
        FtLocalRaw raw;
        memset(&raw, 0, sizeof raw);
        int *resFinal = nullptr;
        rc = fixedPoint(ctx, st, M, N, passBufs(PL, dev, C.obs, tf->d_cache), C,
                        [&](int *res) { return ft_launch_search_local(st, DF, DP, C, th, nn_ratio, res, raw); },
                        [&](int *res, const int *fl, size_t flBytes) -> int {  // pass results, frustum fields and flags: one kernel
                            return ft_launch_deliver_blocks(st, pin + fOutEnd, res, 16 * (size_t)M, pin, dev + fInputEnd, fOutEnd - fInputEnd,
                                                            pin + fOutEnd + 16 * (size_t)M, fl, flBytes);
                        },
                        (const int *)(pin + fOutEnd + 16 * (size_t)M), &resFinal, &passes, &tf->passesLocal);
        if (rc != FT_OK) return rc;
        int *hRes = (int *)(pin + fOutEnd);
        unpackFrustum(M, FL, fInputEnd, pin, frustum, n_to_match);
        nm = replayLocalWrites(hRes, M, P->observations, tf->holder.data(), assign);
        rc = uploadHolder(tf, st);
        if (rc != FT_OK) return rc;
    } else {
        rc = downloadFrustum(st, M, FL, fInputEnd, fOutEnd, dev, pin, frustum, n_to_match);
        if (rc != FT_OK) return rc;
    }
    if (n_matches) *n_matches = nm;
    ctx->addStat("tracked.track_local_map.total", tAll.ms());
    ctx->addStat("tracked.track_local_map.passes", passes);
    return FT_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------------------
// B device-resident frames searched through ONE set of launches (ft_tracked_batch_*; SURVEY.md 7 step 7 "Batch API (B frames
// per launch)").  One frame at a time is what the reference's tracking thread does (src/Tracking.cc:2911-2989, 3472-3555) and it
// leaves a 256-CU chip idle by construction: ~45 launches of a few hundred workgroups per frame.  A batch holds B independent
// frames - the camera streams of one time step, or any frames whose inputs the caller has - and runs every stage as one launch
// over all of them: grid build (blockIdx.z = frame), isInFrustum, and each pass of the two searches' claim iteration
// (blockIdx.y = frame, per-frame convergence flags; the batch runs max-over-frames passes).  Results per frame are those of
// ft_tracked_frame_* on that frame, bit for bit.
//
// Memory: ONE device arena per batch (the kernels re-derive every pointer of a job record from it: Rebase, kernels_search.hip)
//   work   | per call: job records, delivery records, the frames' point arrays (compact), frustum outputs
//   frames | keypoints, descriptors, uright, match tables, holder_obs of the uploaded frames (compact)
//   flags  | 32 words per frame;  counts | 1 word per frame (isInFrustum's nToMatch)
//   grid   | Frame::mGrid as CSR, per frame;  claims | res, list heads + writer table, next, per frame;  cache | per frame
// and two pinned buffers: the mirror of `work` + `frames` (inputs: one H2D copy per call) and the results.
// One search of a batch between its two halves (submit: everything up to the first point where the host must look at the flag
// words, enqueued; wait: the rest).  Holds what the second half needs of the call: nothing of the caller's argument arrays is
// referenced after submit except the OUTPUT arrays, whose pointers are copied here.
struct FtBatchCall {
    int kind = 0;  // 0 = none in flight, 1 = SearchByProjection(CurrentFrame, LastFrame), 2 = isInFrustum + SearchByProjection(Frame, points)
    int n = 0, maxPoints = 0, maxK = 8, maxM = 0, maxFrWords = 0, shInts = 0;
    const FtBatchJob *dJobs = nullptr;
    const FtDeliverRec *dRecs = nullptr;
    float th = 0.f, nnRatio = 0.f, viewingCosLimit = 0.f, logScaleFactor = 0.f, thFar = 0.f;
    int farPoints = 0, checkOrientation = 0;
    bool useResolve = false, frustumDone = false;
    // the claim iteration, split where the host first waits for the device
    int pass = 0, burst = 0, len = 0, prevLen = 0, parity = 0, nextB = 0;
    bool simple = false, awaitResolve = false, resolvedAll = false;
    // results in tb->h_out
    size_t oFlagsOut = 0, oNmOut = 0, oCountsOut = 0, oErrOut = 0;
    std::vector<size_t> outAssign, outFr, fInEnd;
    std::vector<FrustumLayout> FL;
    std::vector<int> M;
    // the caller's output arrays
    std::vector<int *> assign;
    bool assignDirect = false;  // every assign[f] lies in pinned memory: k_replay_batch writes there, the wait copies nothing
    bool frustumDirect = false;  // every array of every frustum[f] lies in pinned memory: a scatter launch behind k_frustum_batch fills them
    const FtGatherRec *dFrRecs = nullptr;
    int nFrRecs = 0;
    int *nMatches = nullptr, *nToMatch = nullptr;
    std::vector<ft_frustum_result> frustum;
    bool haveFrustum = false;
    FtTimer tAll;
};

struct ft_tracked_batch {
    ft_context *ctx = nullptr;
    bool counted = false;
    int maxFrames = 0, maxKp = 0, maxPts = 0;
    uint8_t *d_arena = nullptr;
    size_t arenaBytes = 0;
    size_t oWork = 0, workBytes = 0, oFrames = 0, framesBytes = 0, oFlags = 0, oCounts = 0, oGrid = 0, gridStride = 0, oClaims = 0,
           claimStride = 0, oCache = 0, cacheStride = 0;
    uint8_t *h_in = nullptr;   // pinned mirror of [work | frames]
    uint8_t *h_out = nullptr;  // pinned results
    size_t outBytes = 0;
    // the uploaded frames
    int nFrames = 0;
    std::vector<FtDevFrame> DF;
    // holder_obs (Observations() of mvpMapPoints[i], -1 = none) lives in HBM from the upload on: the searches read it there and
    // k_replay_batch updates it there; ft_tracked_batch_holder_obs copies a frame's array down on request
    std::vector<size_t> holderOff;  // byte offset of frame f's holder_obs inside the frames region
    size_t holderBegin = 0, holderEnd = 0;  // the holder_obs arrays of all frames are contiguous: one copy takes them up
    size_t oReplayed = 0;  // one int per frame: k_replay_batch's "this search's writes have been replayed" marker
    int passesLast = 0, passesLocal = 0;
    bool hasGrid = false;  // the frames' CSR grids were laid out and built at upload / bind time (search_grid as it was THEN)
    // the context's search options as the current call saw them (snapshotTuning, under ctx->matchMutex): the option may be set
    // from another thread while a call runs, and a call must not see two values of it
    int optSearchCache = 0, optPassBurst = 0;
    // a stream and a lock of the batch's own: two batches of one context used from two host threads are two batches in flight -
    // the passes of one run beside the host side (staging, replay) of the other
    hipStream_t stream = nullptr;
    std::mutex mu;
    FtEventTimer evt;  // ft_context_set_kernel_timing: HIP events around the batch's launches on the context's stream
    hipEvent_t evGather = nullptr;  // bind_fisheye: the gather from the extractors' slots has run (the extractors' next batch waits for it)
    hipEvent_t evMirror = nullptr;  // the last copy out of the pinned mirror h_in has run: the next call may repack it
    FtBatchCall call;  // the search between ft_tracked_batch_submit_* and ft_tracked_batch_wait (kind 0: none)
    size_t oErr = 0;   // one int per frame: input errors the kernels found (a last-frame octave outside the frame's levels)
};

namespace {

size_t batchClaimBytes(int maxKp, int maxPts) {
    const size_t K = passK(maxKp);
    return ((32 * (size_t)maxPts + 63) & ~(size_t)63) * 2 + ((4 * 27 * K + 63) & ~(size_t)63) + 4 * (16 + 2 * (size_t)maxPts) + 64;
}
// bytes of one frame's arrays in the frames region, upper bound
size_t batchFrameBytes(int maxKp) {
    const size_t K = (size_t)maxKp;
    return (2 * sizeof(ft_keypoint) + 32 + 4 * 4) * K + 8 * 64;
}
// per call and frame: point inputs (<= 69 B), frustum outputs (<= 47 B) per point + alignment slack
size_t batchWorkBytes(int maxPts) { return 128 * (size_t)maxPts + 26 * 64 + sizeof(FtBatchJob) + 4 * sizeof(FtDeliverRec); }
size_t batchOutBytes(int maxPts) { return (16 + 47) * (size_t)maxPts + 20 * 64; }

// the claim buffers of frame f for a search of nPoints points on nKp keypoints
void batchClaims(const ft_tracked_batch *tb, int f, int nKp, int nPoints, FtBatchJob &J) {
    uint8_t *c = tb->d_arena + tb->oClaims + (size_t)f * tb->claimStride;
    const size_t resBytes = (32 * (size_t)tb->maxPts + 63) & ~(size_t)63;
    J.res = (int *)c;
    J.next = (int *)(c + resBytes);
    J.head = (int *)(c + 2 * resBytes);
    J.K = (int)passK(nKp);
    J.tab = J.head + 3 * (size_t)J.K;
    J.slow = (int *)(c + 2 * resBytes + ((4 * 27 * passK(tb->maxKp) + 63) & ~(size_t)63));
    J.flags = (int *)(tb->d_arena + tb->oFlags) + FT_BATCH_FLAGS * (size_t)f;
    J.cache = tb->oCache ? (unsigned long long *)(tb->d_arena + tb->oCache + (size_t)f * tb->cacheStride) : nullptr;
    J.nKp = nKp;
    J.nPoints = nPoints;
    J.replayed = (int *)(tb->d_arena + tb->oReplayed) + f;
    J.err = (int *)(tb->d_arena + tb->oErr) + f;
}

// first pass of a batched search by the four-points-per-wave kernels (k_search_*_first) where the cache and the grid exist (and
// under search_cache = 2 the rest of the search by k_resolve_batch)
#ifndef FT_ROW_FIRST
#define FT_ROW_FIRST 1
#endif
// search_cache 2: the one-launch resolution for batches of FT_RESOLVE_MIN_FRAMES frames and more.  Its chain is as long for one
// frame as for 256 (a workgroup per frame: 0.26 / 0.5 ms per search at configs[3]) while a claim pass over few frames is a
// 10-us launch: one batch alone is served sooner by the passes up to ~56 frames (1 frame 0.19 against 0.33 ms, 32 frames 0.57
// against 0.65), several batches in flight more cheaply by the resolution from ~24 on (32 frames x 4 lanes: 22.4 k against
// 17.4 k frames/s) - EXPERIMENTS 10.8.  search_cache 3: every batch.
#ifndef FT_RESOLVE_MIN_FRAMES
#define FT_RESOLVE_MIN_FRAMES 24
#endif
// (callers hold tb->mu) the options a batch call works with: read once per call under the mutex ft_context_set_option writes under
void snapshotTuning(ft_tracked_batch *tb) {
    std::lock_guard<std::mutex> lk(tb->ctx->matchMutex);
    tb->optSearchCache = tb->ctx->tuning.search_cache;
    tb->optPassBurst = std::min(std::max(tb->ctx->tuning.pass_burst, 2), FT_PASS_BURST_MAX);
}
bool resolveWanted(const ft_tracked_batch *tb, int nFrames) {
    const int sc = tb->optSearchCache;
    return sc >= 3 || (sc == 2 && nFrames >= FT_RESOLVE_MIN_FRAMES);
}
// the row-first kernels and the one-launch resolution read the candidate cache AND the frames' grids: both must have been laid out
// when the batch was created / the frames were uploaded - the options' CURRENT values say nothing about that
bool rowsUsable(const ft_tracked_batch *tb) { return FT_ROW_FIRST && tb->oCache && tb->hasGrid; }
// The claim iteration of every frame of the batch (see fixedPoint): bursts of passes, one launch per pass for ALL frames, one
// delivery of the flag words + one synchronisation per burst.  A batch has 32 flag positions per burst parity: bursts of up to
// 30 passes (the slowest of many frames needs more passes than one frame does).  With the one-launch resolution
// (c.useResolve; k_resolve_batch: a workgroup per frame walks its points in index order) everything behind the first pass is ONE
// launch; a frame it resolved has all its flag words at -1 and is inert in later passes, and only if it gave up on a frame (a
// candidate list the cache could not hold) do the passes go on - for those frames.
// The iteration is written as two halves around its FIRST host synchronisation: callBegin enqueues everything up to it (the
// whole search when the resolution resolves every frame - the usual case) and returns; callFinish waits, looks at the flag
// words and runs whatever is left.  ft_tracked_batch_submit_* = callBegin, ft_tracked_batch_wait = callFinish.
int callLaunchPass(ft_tracked_batch *tb, FtBatchCall &c, int pass, int fCur, int fPrev, int fReset) {
    ft_context *ctx = tb->ctx;
    hipStream_t st = tb->stream;
    const bool local = c.kind == 2;
    if (local && !c.frustumDone) {  // behind the fill of the claim iteration (which zeroes the counts), in front of the first pass
        c.frustumDone = true;
        tb->evt.begin(ctx->kernelTiming, "kernel.frustum_batch", st);
        const int r = ft_launch_frustum_batch(st, tb->d_arena, c.dJobs, c.n, c.maxM, c.viewingCosLimit, c.logScaleFactor, c.farPoints, c.thFar);
        tb->evt.end(ctx->kernelTiming, st);
        if (r != FT_OK) return r;
    }
    const bool lean = pass > 0 && tb->oCache;
    const bool rows = pass == 0 && rowsUsable(tb);  // (fCur 0, fPrev -1, fReset = half: the kernel's own)
    int r;
    if (local) {
        tb->evt.begin(ctx->kernelTiming, lean ? "kernel.search_local_batch(later pass)" : "kernel.search_local_batch(first pass)", st);
        r = rows   ? ft_launch_search_local_first(st, tb->d_arena, c.dJobs, c.n, c.maxPoints, c.th, c.nnRatio)
            : lean ? ft_launch_search_local_batch_lean(st, tb->d_arena, c.dJobs, c.n, c.maxPoints, pass, fCur, fPrev, fReset, c.th, c.nnRatio)
                   : ft_launch_search_local_batch(st, tb->d_arena, c.dJobs, c.n, c.maxPoints, pass, fCur, fPrev, fReset, c.th, c.nnRatio);
    } else {
        tb->evt.begin(ctx->kernelTiming, lean ? "kernel.search_last_batch(later pass)" : "kernel.search_last_batch(first pass)", st);
        r = rows   ? ft_launch_search_last_first(st, tb->d_arena, c.dJobs, c.n, c.maxPoints, c.th)
            : lean ? ft_launch_search_last_batch_lean(st, tb->d_arena, c.dJobs, c.n, c.maxPoints, pass, fCur, fPrev, fReset, c.th)
                   : ft_launch_search_last_batch(st, tb->d_arena, c.dJobs, c.n, c.maxPoints, pass, fCur, fPrev, fReset, c.th);
    }
    tb->evt.end(ctx->kernelTiming, st);
    if (r == FT_OK && pass == 0 && tb->oCache) {
        tb->evt.begin(ctx->kernelTiming, "kernel.cache_partition_batch", st);
        r = ft_launch_cache_partition_batch(st, tb->d_arena, c.dJobs, c.n, c.maxPoints);
        tb->evt.end(ctx->kernelTiming, st);
    }
    return r;
}

// the flag words, the error words (and the counts) of every frame into pinned host memory; a local-map search's first delivery
// also carries the frustum fields (they do not change from burst to burst).  Records: [n] flags, [n + 1] errors, [n + 2] counts,
// [n + 3 + f] frustum fields of frame f.
int callDeliver(ft_tracked_batch *tb, FtBatchCall &c, int parity, bool first) {
    hipStream_t st = tb->stream;
    if (c.kind == 2 && !c.frustumDone) {  // no frame has keypoints: the frustum fields are still the call's result
        c.frustumDone = true;
        const int r = ft_launch_frustum_batch(st, tb->d_arena, c.dJobs, c.n, c.maxM, c.viewingCosLimit, c.logScaleFactor, c.farPoints, c.thFar);
        if (r != FT_OK) return r;
    }
    const bool frRecs = c.kind == 2 && first && c.haveFrustum && !c.frustumDirect;
    if (c.kind == 2 && first && c.frustumDirect) {
        const int r = ft_launch_gather_batch(st, c.dFrRecs, c.nFrRecs);
        if (r != FT_OK) return r;
    }
    const int nRecs = c.kind == 2 ? (frRecs ? c.n + 3 : 3) : 2;
    return ft_launch_deliver_batch(st, c.dRecs + c.n, nRecs, std::max(frRecs ? c.maxFrWords : 0, FT_BATCH_FLAGS * c.n), parity);
}

int callResolve(ft_tracked_batch *tb, FtBatchCall &c) {
    ft_context *ctx = tb->ctx;
    hipStream_t st = tb->stream;
    const bool local = c.kind == 2;
    tb->evt.begin(ctx->kernelTiming, local ? "kernel.resolve_batch(local map)" : "kernel.resolve_batch(last frame)", st);
    int r = ft_launch_resolve_batch(st, tb->d_arena, c.dJobs, c.n, local ? 1 : 0, c.nnRatio, c.shInts <= 12288 ? c.shInts : 0);
    tb->evt.end(ctx->kernelTiming, st);
    // the writes of the frames it resolved, replayed right behind it (a frame it gave up on waits for the passes)
    tb->evt.begin(ctx->kernelTiming, local ? "kernel.replay_batch(local map)" : "kernel.replay_batch(last frame)", st);
    if (r == FT_OK) r = ft_launch_replay_batch(st, tb->d_arena, c.dJobs, c.n, local ? 1 : 0, 0, c.checkOrientation, c.shInts, /*flagPos=*/0);
    tb->evt.end(ctx->kernelTiming, st);
    return r;
}

constexpr int kFlagHalf = FT_BATCH_FLAGS / 2, kLenMax = kFlagHalf - 2;

// passes [c.nextB, c.len) of burst c.burst, then its delivery
int callRunBurst(ft_tracked_batch *tb, FtBatchCall &c) {
    const int base = kFlagHalf * (c.burst & 1), other = kFlagHalf * ((c.burst + 1) & 1);
    for (int b = c.nextB; b < c.len; b++, c.pass++) {
        const int fPrev = b > 0 ? base + b - 1 : (c.burst > 0 ? other + c.prevLen - 1 : -1);
        int rc = callLaunchPass(tb, c, c.pass, base + b, fPrev, other + b);
        if (rc != FT_OK) return rc;
        c.parity = c.pass & 1;
        if (c.pass == 0 && c.useResolve) {  // the rest of the search in one launch; the host looks at the flag words before it goes on
            rc = callResolve(tb, c);
            if (rc == FT_OK) rc = callDeliver(tb, c, 0, true);
            c.pass++;
            c.nextB = b + 1;
            c.awaitResolve = true;
            return rc;
        }
    }
    c.nextB = c.len;
    // the frames this burst brought to their fixed point are replayed behind it (the kernel looks at the burst's last flag word
    // itself), in front of the delivery: when the host then finds every frame converged the search is complete - no launch and
    // no synchronisation of its own for the replay
    int rc = ft_launch_replay_batch(tb->stream, tb->d_arena, c.dJobs, c.n, c.kind == 2 ? 1 : 0, c.parity, c.checkOrientation, c.shInts,
                                    /*flagPos=*/base + c.len - 1);
    if (rc != FT_OK) return rc;
    return callDeliver(tb, c, c.parity, c.burst == 0 && !c.awaitResolve);
}

int callBegin(ft_tracked_batch *tb, FtBatchCall &c, int burstHint) {
    hipStream_t st = tb->stream;
    const int burstMax = std::min(tb->optPassBurst + 4, kLenMax);
    c.len = burstHint > 0 ? std::min(std::max(burstHint + 1, 4), kLenMax) : burstMax;
    c.pass = c.burst = c.prevLen = c.parity = c.nextB = 0;
    c.simple = c.awaitResolve = c.resolvedAll = false;
    int rc = ft_launch_fill_claims_batch(st, tb->d_arena, c.dJobs, c.n, 27 * c.maxK);
    if (rc != FT_OK) return rc;
    if (c.maxPoints <= 0) {  // nothing to search in any frame: the (empty) results and, for a local-map call, the frustum fields
        c.simple = true;
        rc = ft_launch_replay_batch(st, tb->d_arena, c.dJobs, c.n, c.kind == 2 ? 1 : 0, 0, c.checkOrientation, c.shInts, /*flagPos=*/-1);
        if (rc == FT_OK) rc = callDeliver(tb, c, 0, true);
        c.resolvedAll = true;
        return rc;
    }
    return callRunBurst(tb, c);
}

// the second half: waits for what callBegin enqueued, runs the bursts that are left (none when the resolution resolved every
// frame), has the writes of the frames the passes finished replayed; *passes = claim passes the search took
int callFinish(ft_tracked_batch *tb, FtBatchCall &c, int *passes) {
    ft_context *ctx = tb->ctx;
    hipStream_t st = tb->stream;
    const int *hostFlags = (const int *)(tb->h_out + c.oFlagsOut);
    const int burstMax = std::min(tb->optPassBurst + 4, kLenMax);
    const int maxPasses = 2 * c.maxPoints + 4 + burstMax;
    FT_HIP(hipStreamSynchronize(st));
    *passes = 0;
    if (c.simple) return FT_OK;
    bool firstDelivered = true;  // (callBegin's delivery carried the frustum fields)
    if (c.awaitResolve) {
        bool all = true;
        for (int f = 0; f < c.n && all; f++) all = hostFlags[FT_BATCH_FLAGS * (size_t)f] == -1;
        if (all) {
            c.resolvedAll = true;
            *passes = 2;
            return FT_OK;
        }
        ctx->addStat("tracked_batch.resolve_fallbacks", 1);
        c.awaitResolve = false;
        c.pass = 1;
        int rc = callRunBurst(tb, c);  // the rest of the first burst
        if (rc != FT_OK) return rc;
        FT_HIP(hipStreamSynchronize(st));
    }
    (void)firstDelivered;
    for (;;) {
        const int base = kFlagHalf * (c.burst & 1);
        bool all = true;
        int ranMax = 0;
        for (int f = 0; f < c.n; f++) {
            const int *h = hostFlags + FT_BATCH_FLAGS * (size_t)f + base;
            if (h[c.len - 1] != -1) all = false;
            int ran = 0;
            while (ran < c.len && h[ran] != -1) ran++;
            ranMax = std::max(ranMax, std::min(ran + 1, c.len));
        }
        if (all) {
            c.pass = c.pass - c.len + ranMax;
            break;
        }
        if (c.pass >= maxPasses) {
            ft_set_error("projection search (batch): claim resolution did not converge");
            return FT_ERR_HIP;
        }
        c.prevLen = c.len;
        c.len = std::min(c.len, 6);  // the first burst fell short: short bursts from here (never longer than the one before: the
                                     // flag words beyond a burst's length are not reset by the next one)
        c.burst++;
        c.nextB = 0;
        int rc = callRunBurst(tb, c);
        if (rc != FT_OK) return rc;
        FT_HIP(hipStreamSynchronize(st));
    }
    *passes = c.pass;  // (every frame's writes were replayed behind the burst that brought it to its fixed point: callRunBurst)
    return FT_OK;
}

int checkBatch(const ft_tracked_batch *tb, int n, const char *what) {
    if (!tb) {
        ft_set_error(std::string(what) + ": null batch");
        return FT_ERR_INVALID;
    }
    if (n != tb->nFrames || n <= 0) {
        ft_set_error(std::string(what) + ": n_frames differs from the number of frames uploaded");
        return FT_ERR_INVALID;
    }
    return FT_OK;
}

// LDS ints of k_replay_batch's last-writer table for the uploaded frames (0: a frame beyond the LDS, the table lives in HBM)
int replayShared(const ft_tracked_batch *tb) {
    int maxN = 1;
    for (const FtDevFrame &D : tb->DF) maxN = std::max(maxN, D.N);
    return maxN <= 15360 ? maxN : 0;
}

}  // namespace

extern "C" {

int ft_tracked_batch_create(ft_context *ctx, int max_frames, int max_keypoints, int max_points, ft_tracked_batch **out) {
    FT_REQUIRE(ctx && out && max_frames > 0 && max_keypoints > 0 && max_points > 0, "ft_tracked_batch_create: bad argument");
    FT_REQUIRE(max_frames <= 4096 && max_keypoints < (1 << 24) && max_points < (1 << 22), "ft_tracked_batch_create: capacity out of range");
    int rc = ft_set_device(ctx);
    if (rc != FT_OK) return rc;
    ft_tracked_batch *tb = new ft_tracked_batch();
    tb->ctx = ctx;
    tb->maxFrames = max_frames;
    tb->maxKp = max_keypoints;
    tb->maxPts = max_points;
    const size_t B = (size_t)max_frames;
    Arena a;
    tb->workBytes = (B * batchWorkBytes(max_points) + 4095) & ~(size_t)4095;
    tb->oWork = a.take(tb->workBytes);
    tb->framesBytes = (B * batchFrameBytes(max_keypoints) + 4095) & ~(size_t)4095;
    tb->oFrames = a.take(tb->framesBytes);
    tb->oFlags = a.take(B * FT_BATCH_FLAGS * sizeof(int));
    tb->oCounts = a.take(B * sizeof(int));
    tb->oReplayed = a.take(B * sizeof(int));
    tb->oErr = a.take(B * sizeof(int));
    tb->gridStride = (gridBytes(max_keypoints) + 255) & ~(size_t)255;
    tb->oGrid = a.take(B * tb->gridStride);
    tb->claimStride = (batchClaimBytes(max_keypoints, max_points) + 255) & ~(size_t)255;
    tb->oClaims = a.take(B * tb->claimStride);
    if (searchCacheOn(ctx)) {
        tb->cacheStride = searchCacheBytes(max_points);
        tb->oCache = a.take(B * tb->cacheStride);
    }
    tb->arenaBytes = a.off;
    // results of a search (per point) or of bind_fisheye (per keypoint: match tables, mvDepth, mvStereo3Dpoints)
    // (+ a search's assignments: 4 bytes per keypoint, and its match count)
    tb->outBytes = B * (std::max(batchOutBytes(max_points), 20 * (size_t)max_keypoints + 8 * 64) + 4 * (size_t)max_keypoints + 128) +
                   B * FT_BATCH_FLAGS * sizeof(int) + 4096;
    hipError_t e = hipMalloc((void **)&tb->d_arena, tb->arenaBytes);
    if (e == hipSuccess) e = hipHostMalloc((void **)&tb->h_in, tb->workBytes + tb->framesBytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&tb->h_out, tb->outBytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&tb->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&tb->evGather, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&tb->evMirror, hipEventDisableTiming);
    if (e != hipSuccess) {
        ft_tracked_batch_destroy(tb);
        return ft_hip_fail(e, "ft_tracked_batch_create", __FILE__, __LINE__);
    }
    tb->counted = true;
    ctx->liveObjects++;
    *out = tb;
    return FT_OK;
}

int ft_tracked_batch_destroy(ft_tracked_batch *tb) {
    if (!tb) return FT_OK;
    ft_set_device(tb->ctx);
    if (tb->stream) {
        hipStreamSynchronize(tb->stream);
        hipStreamDestroy(tb->stream);
    }
    if (tb->evMirror) hipEventDestroy(tb->evMirror);
    if (tb->evGather) {  // an extractor bound to this batch may still hold the event for its next batch: it goes with the context
        std::lock_guard<std::mutex> lk(tb->ctx->hostAllocMutex);
        tb->ctx->retiredEvents.push_back(tb->evGather);
    }
    if (tb->d_arena) hipFree(tb->d_arena);
    if (tb->h_in) hipHostFree(tb->h_in);
    if (tb->h_out) hipHostFree(tb->h_out);
    tb->evt.destroy();
    if (tb->counted) tb->ctx->liveObjects--;
    delete tb;
    return FT_OK;
}

int ft_tracked_batch_upload(ft_tracked_batch *tb, int n_frames, const ft_frame_view *frames) {
    FT_REQUIRE(tb && frames && n_frames > 0 && n_frames <= tb->maxFrames, "ft_tracked_batch_upload: bad argument");
    int nlevelsMax = 1;
    bool twoCam = false;
    for (int f = 0; f < n_frames; f++) {
        int rc = checkFrame(&frames[f]);
        if (rc != FT_OK) return rc;
        FT_REQUIRE(frames[f].N <= tb->maxKp, "ft_tracked_batch_upload: more keypoints than the batch was created for");
        nlevelsMax = std::max(nlevelsMax, frames[f].nlevels);
        twoCam = twoCam || frames[f].Nleft != -1;
    }
    ft_context *ctx = tb->ctx;
    int rc = ft_set_device(ctx);
    if (rc != FT_OK) return rc;
    std::lock_guard<std::mutex> lk(tb->mu);
    FT_REQUIRE(tb->call.kind == 0, "ft_tracked_batch_upload: a submitted search of this batch has not been waited for (ft_tracked_batch_wait)");
    FtTimer tAll;
    hipStream_t st = tb->stream;
    bool wantGrid;
    {
        std::lock_guard<std::mutex> lo(ctx->matchMutex);
        wantGrid = ctx->tuning.search_grid != 0;
    }
    tb->hasGrid = false;  // (until the launch that builds the grids of THESE frames is enqueued)
    FT_HIP(hipStreamSynchronize(st));  // the pinned mirror is repacked: nothing of an earlier call may still read it
    // layout of the frames region: the holder_obs arrays of all frames first (contiguous: refreshed after every search by one
    // copy), then every frame's arrays
    Arena a;
    tb->holderOff.assign(n_frames, 0);
    tb->holderBegin = a.off;
    for (int f = 0; f < n_frames; f++) tb->holderOff[f] = a.take(sizeof(int) * std::max(frames[f].N, 1));
    tb->holderEnd = a.off;
    struct Lay {
        size_t keys, keysR, desc, uright, l2r, r2l;
    };
    std::vector<Lay> lay(n_frames);
    for (int f = 0; f < n_frames; f++) {
        const ft_frame_view &F = frames[f];
        const int nL = F.Nleft == -1 ? F.N : F.Nleft, nR = F.Nleft == -1 ? 0 : F.N - F.Nleft;
        lay[f].keys = a.take(sizeof(ft_keypoint) * std::max(nL, 1));
        lay[f].keysR = a.take(sizeof(ft_keypoint) * std::max(nR, 1));
        lay[f].desc = a.take((size_t)32 * std::max(F.N, 1));
        lay[f].uright = a.take(sizeof(float) * std::max(F.N, 1));
        lay[f].l2r = a.take(sizeof(int) * std::max(nL, 1));
        lay[f].r2l = a.take(sizeof(int) * std::max(nR, 1));
    }
    FT_REQUIRE(a.off <= tb->framesBytes, "ft_tracked_batch_upload: frames region too small");
    tb->nFrames = n_frames;
    tb->DF.assign(n_frames, FtDevFrame());
    uint8_t *pinF = tb->h_in + tb->workBytes, *devF = tb->d_arena + tb->oFrames;
    FtBatchJob *hJobs = (FtBatchJob *)tb->h_in;
    FT_REQUIRE((size_t)n_frames * sizeof(FtBatchJob) <= tb->workBytes, "ft_tracked_batch_upload: work region too small");
    // the caller's arrays are pageable as a rule: packed into the pinned mirror by the context's host threads, one frame each
    const std::function<void(int, int)> stage = [&](int f, int) {
        const ft_frame_view &F = frames[f];
        const int nL = F.Nleft == -1 ? F.N : F.Nleft, nR = F.Nleft == -1 ? 0 : F.N - F.Nleft;
        if (nL) memcpy(pinF + lay[f].keys, F.keys, sizeof(ft_keypoint) * nL);
        if (nR) memcpy(pinF + lay[f].keysR, F.keys_right, sizeof(ft_keypoint) * nR);
        if (F.N) memcpy(pinF + lay[f].desc, F.descriptors, (size_t)32 * F.N);
        if (F.uright && F.N) memcpy(pinF + lay[f].uright, F.uright, sizeof(float) * F.N);
        if (F.Nleft != -1) {
            if (nL) memcpy(pinF + lay[f].l2r, F.left_to_right, sizeof(int) * nL);
            if (nR) memcpy(pinF + lay[f].r2l, F.right_to_left, sizeof(int) * nR);
        }
        if (F.N) memcpy(pinF + tb->holderOff[f], F.holder_obs, sizeof(int) * F.N);
        FtDevFrame &D = tb->DF[f];
        D = devFrameConstants(&F);
        D.keys = (const ft_keypoint *)(devF + lay[f].keys);
        D.keysR = (const ft_keypoint *)(devF + lay[f].keysR);
        D.desc = devF + lay[f].desc;
        D.uright = F.uright ? (const float *)(devF + lay[f].uright) : nullptr;
        D.holderObs = (const int *)(devF + tb->holderOff[f]);
        D.l2r = F.Nleft != -1 ? (const int *)(devF + lay[f].l2r) : nullptr;
        D.r2l = F.Nleft != -1 ? (const int *)(devF + lay[f].r2l) : nullptr;
        if (wantGrid) {  // the arrays k_build_grid_batch fills (buildGrid's layout, per frame)
            int *grid = (int *)(tb->d_arena + tb->oGrid + (size_t)f * tb->gridStride);
            const bool two = D.Nleft != -1;
            float4 *rec = (float4 *)((uint8_t *)grid + gridIntBytes(D.N));
            uint8_t *gdesc = (uint8_t *)(rec + std::max(D.N, 1));
            D.gridStart[0] = grid;
            D.gridStart[1] = two ? grid + (size_t)FT_MAX_LEVELS * (FT_GRID_CELLS + 1) : nullptr;
            D.gridRec[0] = rec;
            D.gridDesc[0] = gdesc;
            D.gridRec[1] = two ? rec + nL : nullptr;
            D.gridDesc[1] = two ? gdesc + (size_t)32 * nL : nullptr;
        }
        memset(&hJobs[f], 0, sizeof(FtBatchJob));
        hJobs[f].F = D;
    };
    ctx->pool->parallel_for(n_frames, stage);
    FT_HIP(hipMemcpyAsync(tb->d_arena + tb->oWork, tb->h_in, (size_t)n_frames * sizeof(FtBatchJob), hipMemcpyHostToDevice, st));
    FT_HIP(hipMemcpyAsync(devF, pinF, a.off, hipMemcpyHostToDevice, st));
    FT_HIP(hipEventRecord(tb->evMirror, st));
    if (wantGrid) {
        rc = ft_launch_build_grid_batch(st, tb->d_arena, (const FtBatchJob *)(tb->d_arena + tb->oWork), n_frames, nlevelsMax, twoCam);
        if (rc != FT_OK) return rc;
        tb->hasGrid = true;
    }
    ctx->addStat("tracked_batch.upload.total", tAll.ms());
    return FT_OK;
}

int ft_tracked_batch_holder_obs(ft_tracked_batch *tb, int frame, int *holder_obs) {
    FT_REQUIRE(tb && holder_obs, "ft_tracked_batch_holder_obs: bad argument");
    std::lock_guard<std::mutex> lk(tb->mu);  // (upload / bind_fisheye reassign the vectors)
    FT_REQUIRE(frame >= 0 && frame < tb->nFrames, "ft_tracked_batch_holder_obs: bad argument");
    FT_REQUIRE(tb->call.kind == 0, "ft_tracked_batch_holder_obs: a submitted search of this batch has not been waited for (ft_tracked_batch_wait)");
    int rc = ft_set_device(tb->ctx);
    if (rc != FT_OK) return rc;
    const int N = tb->DF[frame].N;
    if (N > 0) {  // (the array lives in HBM; the copy is ordered behind the batch's searches on its stream)
        FT_HIP(hipMemcpyAsync(holder_obs, tb->d_arena + tb->oFrames + tb->holderOff[frame], sizeof(int) * (size_t)N, hipMemcpyDeviceToHost, tb->stream));
        FT_HIP(hipStreamSynchronize(tb->stream));
    }
    return FT_OK;
}

}  // extern "C"

namespace {

// May the device read [p, p + bytes) in place?  A block of ft_host_malloc is known without asking the runtime; anything else is
// asked about (hipHostMalloc / hipHostRegister memory of the application qualifies, at a few microseconds per array).
bool readableInPlace(ft_context *ctx, const void *p, size_t bytes) {
    if (!p || bytes == 0) return true;
    return ft_host_block_contains(ctx, p, bytes) || ft_is_pinned_host_range(p, bytes);
}

// the records of a call's head in the pinned mirror: jobs | delivery records | gather records (the gather records only when the
// caller's arrays are read in place)
struct CallHead {
    size_t oJobs, oRecs, oGather;
    FtBatchJob *hJobs;
    FtDeliverRec *hRecs;
    FtGatherRec *hGather;
};
CallHead layoutHead(ft_tracked_batch *tb, Arena &a, int n, int nRecs, int nGather) {
    CallHead H;
    H.oJobs = a.take((size_t)n * sizeof(FtBatchJob));
    H.oRecs = a.take((size_t)nRecs * sizeof(FtDeliverRec));
    H.oGather = a.take((size_t)std::max(nGather, 1) * sizeof(FtGatherRec));
    H.hJobs = (FtBatchJob *)(tb->h_in + H.oJobs);
    H.hRecs = (FtDeliverRec *)(tb->h_in + H.oRecs);
    H.hGather = (FtGatherRec *)(tb->h_in + H.oGather);
    return H;
}

int requireIdle(ft_tracked_batch *tb, const char *what) {
    if (tb->call.kind != 0) {
        ft_set_error(std::string(what) + ": a submitted search of this batch has not been waited for (ft_tracked_batch_wait)");
        return FT_ERR_INVALID;
    }
    return FT_OK;
}

// SearchByProjection(CurrentFrame, LastFrame) of every frame, first half.  The point arrays are read IN PLACE by the device when
// every one of them lies in pinned host memory (one gather launch, no host copy; they must then stay unchanged until the wait
// returns); otherwise the host threads of the context pack them into the batch's pinned mirror and one copy takes them up.
int submitLastFrame(ft_tracked_batch *tb, int n, const ft_last_points *L, const FtPose *poses, const FtPose *trls, bool needTrl, float th,
                    const int *forward, const int *backward, int check_orientation, int *const *assign, int *n_matches) {
    int rc = requireIdle(tb, "ft_tracked_batch_search_last_frame");
    if (rc != FT_OK) return rc;
    rc = checkBatch(tb, n, "ft_tracked_batch_search_last_frame");
    if (rc != FT_OK) return rc;
    FT_REQUIRE(L && assign, "ft_tracked_batch_search_last_frame: null argument");
    ft_context *ctx = tb->ctx;
    snapshotTuning(tb);
    bool inPlace = true;
    for (int f = 0; f < n; f++) {
        FT_REQUIRE(!needTrl || trls || tb->DF[f].Nleft == -1, "ft_tracked_batch_search_last_frame_se3: a two-camera frame needs Trl");
        const int M = L[f].N;
        FT_REQUIRE(assign[f], "ft_tracked_batch_search_last_frame: null assign array");
        FT_REQUIRE(M >= 0 && M <= tb->maxPts, "last-frame point count beyond the batch's capacity");
        FT_REQUIRE(M == 0 || (L[f].valid && L[f].world_pos && L[f].descriptors && L[f].observations && L[f].octave && L[f].angle),
                   "last-frame arrays are null");
        const size_t m = (size_t)M;
        inPlace = inPlace && readableInPlace(ctx, L[f].valid, m) && readableInPlace(ctx, L[f].world_pos, 12 * m) &&
                  readableInPlace(ctx, L[f].descriptors, 32 * m) && readableInPlace(ctx, L[f].observations, 4 * m) &&
                  readableInPlace(ctx, L[f].octave, 4 * m) && readableInPlace(ctx, L[f].angle, 4 * m);
    }
    if (!inPlace)  // the host reads the arrays anyway: octaves checked here (in place: k_last_project_batch checks them, the wait reports)
        for (int f = 0; f < n; f++)
            for (int i = 0; i < L[f].N; i++)
                FT_REQUIRE(!L[f].valid[i] || (L[f].octave[i] >= 0 && L[f].octave[i] < tb->DF[f].nlevels), "last-frame octave out of range");
    rc = ft_set_device(ctx);
    if (rc != FT_OK) return rc;
    FtBatchCall &c = tb->call;
    c = FtBatchCall();
    c.assignDirect = true;
    for (int f = 0; f < n; f++) c.assignDirect = c.assignDirect && readableInPlace(ctx, assign[f], 4 * (size_t)std::max(tb->DF[f].N, 1));
    hipStream_t st = tb->stream;
    // layout of the call in the work region: job records | delivery records | gather records | per frame the point arrays
    Arena a;
    const CallHead H = layoutHead(tb, a, n, n + 2, inPlace ? 6 * n : 0);
    const size_t headEnd = a.off;
    struct Lay {
        size_t valid, pos, desc, obs, oct, ang, proj;
    };
    std::vector<Lay> lay(n);
    Arena o;  // results in tb->h_out: flag words, error words, match counts, every frame's assignments (k_replay_batch writes them there)
    c.oFlagsOut = o.take((size_t)n * FT_BATCH_FLAGS * sizeof(int));
    c.oErrOut = o.take((size_t)n * sizeof(int));
    c.oNmOut = o.take((size_t)n * sizeof(int));
    c.outAssign.resize(n);
    for (int f = 0; f < n; f++) {
        const size_t M = (size_t)std::max(L[f].N, 1);
        lay[f].valid = a.take(M);
        lay[f].pos = a.take(12 * M);
        lay[f].desc = a.take(32 * M);
        lay[f].obs = a.take(4 * M);
        lay[f].oct = a.take(4 * M);
        lay[f].ang = a.take(4 * M);
        c.outAssign[f] = o.take(4 * (size_t)std::max(tb->DF[f].N, 1));
    }
    const size_t inputEnd = a.off;  // what follows is device-only: the projections
    for (int f = 0; f < n; f++) lay[f].proj = a.take(sizeof(FtLastProj) * (size_t)std::max(L[f].N, 1));
    FT_REQUIRE(a.off <= tb->workBytes && o.off <= tb->outBytes, "tracked batch work arena too small");
    uint8_t *pin = tb->h_in, *dev = tb->d_arena + tb->oWork;
    FT_HIP(hipEventSynchronize(tb->evMirror));  // (the pinned mirror: the previous call's copy out of it - not the whole stream, whose
                                                // kernels - a bind_fisheye enqueued just before - may run on while this call is staged)
    const std::function<void(int, int)> stage = [&](int f, int) {
        const ft_last_points &P = L[f];
        const size_t M = (size_t)P.N;
        const int N = tb->DF[f].N;
        if (inPlace) {
            FtGatherRec *G = H.hGather + 6 * (size_t)f;
            G[0] = {dev + lay[f].valid, P.valid, (unsigned)M};
            G[1] = {dev + lay[f].pos, P.world_pos, (unsigned)(12 * M)};
            G[2] = {dev + lay[f].desc, P.descriptors, (unsigned)(32 * M)};
            G[3] = {dev + lay[f].obs, P.observations, (unsigned)(4 * M)};
            G[4] = {dev + lay[f].oct, P.octave, (unsigned)(4 * M)};
            G[5] = {dev + lay[f].ang, P.angle, (unsigned)(4 * M)};
        } else if (M) {
            memcpy(pin + lay[f].valid, P.valid, M);
            memcpy(pin + lay[f].pos, P.world_pos, 12 * M);
            memcpy(pin + lay[f].desc, P.descriptors, 32 * M);
            memcpy(pin + lay[f].obs, P.observations, 4 * M);
            memcpy(pin + lay[f].oct, P.octave, 4 * M);
            memcpy(pin + lay[f].ang, P.angle, 4 * M);
        }
        FtBatchJob &J = H.hJobs[f];
        memset(&J, 0, sizeof J);
        J.F = tb->DF[f];
        if (trls) setTrl(J.F, trls[f]);
        batchClaims(tb, f, N, N > 0 ? (int)M : 0, J);
        J.obs = (const int *)(dev + lay[f].obs);
        J.L.N = (int)M;
        J.L.valid = dev + lay[f].valid;
        J.L.worldPos = (const float *)(dev + lay[f].pos);
        J.L.desc = dev + lay[f].desc;
        J.L.octave = (const int *)(dev + lay[f].oct);
        J.L.angle = (const float *)(dev + lay[f].ang);
        J.proj = (FtLastProj *)(dev + lay[f].proj);
        J.Tcw = poses[f];
        J.forward = forward ? forward[f] : 0;
        J.backward = backward ? backward[f] : 0;
        J.assignOut = c.assignDirect ? assign[f] : (int *)(tb->h_out + c.outAssign[f]);
        J.nmOut = (int *)(tb->h_out + c.oNmOut) + f;
        memset(&H.hRecs[f], 0, sizeof(FtDeliverRec));  // (the points' results stay on the device: k_replay_batch turns them into assignments there)
    };
    if (inPlace)
        for (int f = 0; f < n; f++) stage(f, 0);  // (a few hundred bytes per frame: not worth waking the pool)
    else
        ctx->pool->parallel_for(n, stage);
    for (int f = 0; f < n; f++) {
        c.maxPoints = std::max(c.maxPoints, H.hJobs[f].nPoints);
        if (H.hJobs[f].nPoints > 0) c.maxK = std::max(c.maxK, H.hJobs[f].K);
    }
    H.hRecs[n] = {tb->h_out + c.oFlagsOut, {tb->d_arena + tb->oFlags, tb->d_arena + tb->oFlags}, FT_BATCH_FLAGS * n};
    H.hRecs[n + 1] = {tb->h_out + c.oErrOut, {tb->d_arena + tb->oErr, tb->d_arena + tb->oErr}, n};
    ctx->addStat("tracked_batch.search_last_frame.stage", c.tAll.ms());
    FT_HIP(hipMemcpyAsync(dev, pin, inPlace ? headEnd : inputEnd, hipMemcpyHostToDevice, st));
    FT_HIP(hipEventRecord(tb->evMirror, st));
    if (inPlace) {
        rc = ft_launch_gather_batch(st, (const FtGatherRec *)(dev + H.oGather), 6 * n);
        if (rc != FT_OK) return rc;
    }
    c.n = n;
    c.dJobs = (const FtBatchJob *)(dev + H.oJobs);
    c.dRecs = (const FtDeliverRec *)(dev + H.oRecs);
    c.th = th;
    c.checkOrientation = check_orientation;
    c.useResolve = rowsUsable(tb) && resolveWanted(tb, n);
    c.shInts = replayShared(tb);
    c.assign.assign(assign, assign + n);
    c.nMatches = n_matches;
    c.kind = 1;
    rc = callBegin(tb, c, tb->passesLast);
    if (rc != FT_OK) c.kind = 0;
    return rc;
}

// isInFrustum + SearchByProjection(Frame, local map points) of every frame, first half (inputs as submitLastFrame)
int submitLocalMap(ft_tracked_batch *tb, int n, const ft_frame_pose *poses, const ft_map_points *P, float viewing_cos_limit,
                   float log_scale_factor, float th, float nn_ratio, int far_points, float th_far_points,
                   const ft_frustum_result *frustum, int *n_to_match, int *const *assign, int *n_matches) {
    int rc = requireIdle(tb, "ft_tracked_batch_track_local_map");
    if (rc != FT_OK) return rc;
    rc = checkBatch(tb, n, "ft_tracked_batch_track_local_map");
    if (rc != FT_OK) return rc;
    snapshotTuning(tb);
    FT_REQUIRE(poses && P && assign, "ft_tracked_batch_track_local_map: null argument");
    ft_context *ctx = tb->ctx;
    bool inPlace = true;
    for (int f = 0; f < n; f++) {
        rc = checkMapPoints(&P[f], true);
        if (rc != FT_OK) return rc;
        FT_REQUIRE(P[f].M <= tb->maxPts, "map point count beyond the batch's capacity");
        FT_REQUIRE(assign[f], "ft_tracked_batch_track_local_map: null assign array");
        const size_t m = (size_t)P[f].M;
        inPlace = inPlace && readableInPlace(ctx, P[f].skip, m) && readableInPlace(ctx, P[f].world_pos, 12 * m) &&
                  readableInPlace(ctx, P[f].normal, 12 * m) && readableInPlace(ctx, P[f].max_distance, 4 * m) &&
                  readableInPlace(ctx, P[f].min_distance, 4 * m) && readableInPlace(ctx, P[f].descriptors, 32 * m) &&
                  readableInPlace(ctx, P[f].observations, 4 * m);
    }
    rc = ft_set_device(ctx);
    if (rc != FT_OK) return rc;
    FtBatchCall &c = tb->call;
    c = FtBatchCall();
    c.assignDirect = true;
    for (int f = 0; f < n; f++) c.assignDirect = c.assignDirect && readableInPlace(ctx, assign[f], 4 * (size_t)std::max(tb->DF[f].N, 1));
    hipStream_t st = tb->stream;
    Arena a;
    // the frustum fields: straight into the caller's arrays when every one of them lies in pinned memory (a scatter launch, 12
    // records per frame, behind k_frustum_batch), else through the batch's result buffer and the host's copies (unpackFrustum)
    bool frDirect = frustum != nullptr;
    auto field_ptrs = [](const ft_frustum_result &R, void *out[12]) {
        void *p[12] = {R.in_view, R.in_view_r, R.level, R.level_r, R.view_cos, R.view_cos_r, R.proj_x, R.proj_y, R.proj_xr, R.proj_yr, R.depth, R.depth_r};
        for (int k = 0; k < 12; k++) out[k] = p[k];
    };
    static const int kFieldBytes[12] = {1, 1, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4};
    for (int f = 0; f < n && frDirect; f++) {
        void *fp[12];
        field_ptrs(frustum[f], fp);
        for (int k = 0; k < 12; k++) frDirect = frDirect && readableInPlace(ctx, fp[k], (size_t)kFieldBytes[k] * (size_t)P[f].M);
    }
    const CallHead H = layoutHead(tb, a, n, 2 * n + 3, (inPlace ? 7 * n : 0) + (frDirect ? 12 * n : 0));
    const size_t headEnd = a.off;
    struct Lay {
        size_t fIn0, fOutEnd, desc, obs;
    };
    std::vector<Lay> lay(n);
    c.FL.resize(n);
    c.fInEnd.resize(n);
    c.outAssign.resize(n);
    c.outFr.resize(n);
    c.M.resize(n);
    Arena o;
    c.oFlagsOut = o.take((size_t)n * FT_BATCH_FLAGS * sizeof(int));
    c.oErrOut = o.take((size_t)n * sizeof(int));
    c.oCountsOut = o.take((size_t)n * sizeof(int));
    c.oNmOut = o.take((size_t)n * sizeof(int));
    // inputs of all frames first (one H2D copy when they are staged), then the frustum outputs (device only)
    for (int f = 0; f < n; f++) {
        const size_t M = (size_t)std::max(P[f].M, 1);
        lay[f].desc = a.take(32 * M);
        lay[f].obs = a.take(4 * M);
        c.M[f] = P[f].M;
    }
    // (FrustumLayout interleaves a frame's inputs and outputs: a staged copy covers both, the outputs' share is ~40 %)
    for (int f = 0; f < n; f++) {
        lay[f].fIn0 = a.off;
        layoutFrustum(P[f].M, P[f].skip != nullptr, a, c.FL[f], &c.fInEnd[f]);
        lay[f].fOutEnd = a.off;
        c.outAssign[f] = o.take(4 * (size_t)std::max(tb->DF[f].N, 1));
        c.outFr[f] = o.take(lay[f].fOutEnd - c.fInEnd[f]);
    }
    FT_REQUIRE(a.off <= tb->workBytes && o.off <= tb->outBytes, "tracked batch work arena too small");
    uint8_t *pin = tb->h_in, *dev = tb->d_arena + tb->oWork;
    FT_HIP(hipEventSynchronize(tb->evMirror));
    const std::function<void(int, int)> stage = [&](int f, int) {
        const ft_map_points &Q = P[f];
        const size_t M = (size_t)Q.M;
        const int N = tb->DF[f].N;
        const FrustumLayout &FL = c.FL[f];
        if (inPlace) {
            FtGatherRec *G = H.hGather + 7 * (size_t)f;
            G[0] = {dev + FL.skip, Q.skip, (unsigned)(Q.skip ? M : 0)};
            G[1] = {dev + FL.pos, Q.world_pos, (unsigned)(12 * M)};
            G[2] = {dev + FL.nrm, Q.normal, (unsigned)(12 * M)};
            G[3] = {dev + FL.maxd, Q.max_distance, (unsigned)(4 * M)};
            G[4] = {dev + FL.mind, Q.min_distance, (unsigned)(4 * M)};
            G[5] = {dev + lay[f].desc, Q.descriptors, (unsigned)(32 * M)};
            G[6] = {dev + lay[f].obs, Q.observations, (unsigned)(4 * M)};
        } else {
            stageFrustum(&Q, FL, pin);
            if (M) {
                memcpy(pin + lay[f].desc, Q.descriptors, 32 * M);
                memcpy(pin + lay[f].obs, Q.observations, 4 * M);
            }
        }
        FtBatchJob &J = H.hJobs[f];
        memset(&J, 0, sizeof J);
        J.F = tb->DF[f];
        batchClaims(tb, f, N, N > 0 ? (int)M : 0, J);
        J.obs = (const int *)(dev + lay[f].obs);
        J.MP = devMapPoints(&Q, FL, dev);
        J.T = frustumPose_fromDev(J.F, &poses[f]);
        J.O = devFrustumOut(FL, dev);
        J.O.count = (int *)(tb->d_arena + tb->oCounts) + f;
        J.P.M = (int)M;
        J.P.skip = J.O.searchSkip; J.P.inView = J.O.inView; J.P.inViewR = J.O.inViewR;
        J.P.level = J.O.level; J.P.levelR = J.O.levelR;
        J.P.viewCos = J.O.viewCos; J.P.viewCosR = J.O.viewCosR;
        J.P.projX = J.O.projX; J.P.projY = J.O.projY; J.P.projXR = J.O.projXR; J.P.projYR = J.O.projYR;
        J.P.desc = dev + lay[f].desc;
        J.assignOut = c.assignDirect ? assign[f] : (int *)(tb->h_out + c.outAssign[f]);
        J.nmOut = (int *)(tb->h_out + c.oNmOut) + f;
        if (frDirect) {
            FtGatherRec *G = H.hGather + (inPlace ? 7 * (size_t)n : 0) + 12 * (size_t)f;
            void *fp[12];
            field_ptrs(frustum[f], fp);
            const size_t srcOff[12] = {FL.inV, FL.inVR, FL.lvl, FL.lvlR, FL.vc, FL.vcR, FL.px, FL.py, FL.pxr, FL.pyr, FL.dep, FL.depR};
            for (int k = 0; k < 12; k++) G[k] = {fp[k], dev + srcOff[k], (unsigned)(fp[k] ? (size_t)kFieldBytes[k] * M : 0)};
        }
        memset(&H.hRecs[f], 0, sizeof(FtDeliverRec));
        FtDeliverRec &R2 = H.hRecs[n + 3 + f];  // (the frustum fields do not change from burst to burst: delivered with the first one)
        R2.dst = tb->h_out + c.outFr[f];
        R2.src[0] = R2.src[1] = dev + c.fInEnd[f];
        R2.words = M ? (int)((lay[f].fOutEnd - c.fInEnd[f]) / 4) : 0;
    };
    if (inPlace)
        for (int f = 0; f < n; f++) stage(f, 0);
    else
        ctx->pool->parallel_for(n, stage);
    for (int f = 0; f < n; f++) {
        c.maxPoints = std::max(c.maxPoints, H.hJobs[f].nPoints);
        c.maxM = std::max(c.maxM, P[f].M);
        if (H.hJobs[f].nPoints > 0) c.maxK = std::max(c.maxK, H.hJobs[f].K);
        c.maxFrWords = std::max(c.maxFrWords, H.hRecs[n + 3 + f].words);
    }
    H.hRecs[n] = {tb->h_out + c.oFlagsOut, {tb->d_arena + tb->oFlags, tb->d_arena + tb->oFlags}, FT_BATCH_FLAGS * n};
    H.hRecs[n + 1] = {tb->h_out + c.oErrOut, {tb->d_arena + tb->oErr, tb->d_arena + tb->oErr}, n};
    H.hRecs[n + 2] = {tb->h_out + c.oCountsOut, {tb->d_arena + tb->oCounts, tb->d_arena + tb->oCounts}, n};
    ctx->addStat("tracked_batch.track_local_map.stage", c.tAll.ms());
    FT_HIP(hipMemcpyAsync(dev, pin, inPlace ? headEnd : a.off, hipMemcpyHostToDevice, st));
    FT_HIP(hipEventRecord(tb->evMirror, st));
    if (inPlace) {
        rc = ft_launch_gather_batch(st, (const FtGatherRec *)(dev + H.oGather), 7 * n);
        if (rc != FT_OK) return rc;
    }
    c.n = n;
    c.dJobs = (const FtBatchJob *)(dev + H.oJobs);
    c.dRecs = (const FtDeliverRec *)(dev + H.oRecs);
    c.th = th;
    c.nnRatio = nn_ratio;
    c.viewingCosLimit = viewing_cos_limit;
    c.logScaleFactor = log_scale_factor;
    c.farPoints = far_points;
    c.thFar = th_far_points;
    c.useResolve = rowsUsable(tb) && resolveWanted(tb, n);
    c.shInts = replayShared(tb);
    c.assign.assign(assign, assign + n);
    c.nMatches = n_matches;
    c.nToMatch = n_to_match;
    c.haveFrustum = frustum != nullptr;
    c.frustumDirect = frDirect;
    c.dFrRecs = (const FtGatherRec *)(dev + H.oGather) + (inPlace ? 7 * (size_t)n : 0);
    c.nFrRecs = frDirect ? 12 * n : 0;
    if (frustum) c.frustum.assign(frustum, frustum + n);
    c.kind = 2;
    rc = callBegin(tb, c, tb->passesLocal);
    if (rc != FT_OK) c.kind = 0;
    return rc;
}

// the second half of either search: waits for the device, runs the claim passes the resolution left (none as a rule), hands the
// results to the caller's arrays
int waitCall(ft_tracked_batch *tb) {
    FtBatchCall &c = tb->call;
    if (c.kind == 0) return FT_OK;
    ft_context *ctx = tb->ctx;
    int rc = ft_set_device(ctx);
    const int kind = c.kind, n = c.n;
    const char *name = kind == 1 ? "search_last_frame" : "track_local_map";
    if (rc != FT_OK) {
        c.kind = 0;
        return rc;
    }
    FtTimer tDev;
    int passes = 0;
    rc = callFinish(tb, c, &passes);
    c.kind = 0;  // (whatever happened, the batch is free for the next call)
    if (rc != FT_OK) return rc;
    ctx->addStat((std::string("tracked_batch.") + name + ".device").c_str(), tDev.ms());
    tb->evt.resolve(ctx);
    (kind == 1 ? tb->passesLast : tb->passesLocal) = passes;
    FtTimer tRep;
    const int *hErr = (const int *)(tb->h_out + c.oErrOut);
    for (int f = 0; f < n; f++)
        if (hErr[f] & FT_JOB_ERR_OCTAVE) {
            ft_set_error("last-frame octave out of range");
            return FT_ERR_INVALID;
        }
    // what is left for the host: the assignments (and frustum fields) out of the pinned result buffer into the caller's arrays
    const int *hNm = (const int *)(tb->h_out + c.oNmOut), *hCounts = (const int *)(tb->h_out + c.oCountsOut);
    const std::function<void(int, int)> finish = [&](int f, int) {
        const int N = tb->DF[f].N;
        if (N > 0 && !c.assignDirect) memcpy(c.assign[f], tb->h_out + c.outAssign[f], sizeof(int) * (size_t)N);
        if (c.nMatches) c.nMatches[f] = hNm[f];
        if (kind == 2) {
            const int M = c.M[f];
            // unpackFrustum reads the count through the layout; the batch keeps the counts of all frames in one block
            if (M > 0 && c.haveFrustum && !c.frustumDirect) unpackFrustum(M, c.FL[f], c.fInEnd[f], tb->h_out + c.outFr[f], &c.frustum[f], nullptr);
            if (c.nToMatch) c.nToMatch[f] = M > 0 ? hCounts[f] : 0;
        }
    };
    if (n <= 8) for (int f = 0; f < n; f++) finish(f, 0);
    else ctx->pool->parallel_for(n, finish);
    ctx->addStat((std::string("tracked_batch.") + name + ".replay").c_str(), tRep.ms());
    ctx->addStat((std::string("tracked_batch.") + name + ".total").c_str(), c.tAll.ms());
    ctx->addStat((std::string("tracked_batch.") + name + ".passes").c_str(), passes);
    ctx->addStat((std::string("tracked_batch.") + name + ".frames").c_str(), n);
    return FT_OK;
}

int posesOfMatrices(const float *Tcw, int n, std::vector<FtPose> &poses) {
    poses.resize(n);
    for (int f = 0; f < n; f++) poses[f] = poseOfMatrix(Tcw + 12 * (size_t)f);
    return FT_OK;
}
int posesOfSe3(const ft_se3 *Tcw, const ft_se3 *Trl, int n, std::vector<FtPose> &poses, std::vector<FtPose> &trls) {
    poses.resize(n);
    trls.resize(n);
    for (int f = 0; f < n; f++) {  // (the batch's own state - frame count, camera counts - is checked under its lock)
        int rc = poseOfSe3(&Tcw[f], poses[f]);
        if (rc == FT_OK && Trl) rc = poseOfSe3(&Trl[f], trls[f]);
        if (rc != FT_OK) return rc;
    }
    return FT_OK;
}
}  // namespace

extern "C" {

int ft_tracked_batch_submit_search_last_frame(ft_tracked_batch *tb, int n_frames, const ft_last_points *L, const float *Tcw, float th,
                                              const int *forward, const int *backward, int check_orientation, int *const *assign,
                                              int *n_matches) {
    FT_REQUIRE(tb && Tcw && n_frames > 0, "ft_tracked_batch_search_last_frame: null batch or pose");
    std::vector<FtPose> poses;
    posesOfMatrices(Tcw, n_frames, poses);
    std::lock_guard<std::mutex> lk(tb->mu);  // (before anything of the batch is read: upload / bind_fisheye reassign it)
    return submitLastFrame(tb, n_frames, L, poses.data(), nullptr, false, th, forward, backward, check_orientation, assign, n_matches);
}

int ft_tracked_batch_submit_search_last_frame_se3(ft_tracked_batch *tb, int n_frames, const ft_last_points *L, const ft_se3 *Tcw,
                                                  const ft_se3 *Trl, float th, const int *forward, const int *backward,
                                                  int check_orientation, int *const *assign, int *n_matches) {
    FT_REQUIRE(tb && Tcw && n_frames > 0, "ft_tracked_batch_search_last_frame_se3: bad argument");
    std::vector<FtPose> poses, trls;
    int rc = posesOfSe3(Tcw, Trl, n_frames, poses, trls);
    if (rc != FT_OK) return rc;
    std::lock_guard<std::mutex> lk(tb->mu);
    return submitLastFrame(tb, n_frames, L, poses.data(), Trl ? trls.data() : nullptr, true, th, forward, backward, check_orientation, assign,
                           n_matches);
}

int ft_tracked_batch_submit_track_local_map(ft_tracked_batch *tb, int n_frames, const ft_frame_pose *poses, const ft_map_points *P,
                                            float viewing_cos_limit, float log_scale_factor, float th, float nn_ratio, int far_points,
                                            float th_far_points, const ft_frustum_result *frustum, int *n_to_match, int *const *assign,
                                            int *n_matches) {
    FT_REQUIRE(tb, "ft_tracked_batch_track_local_map: null batch");
    std::lock_guard<std::mutex> lk(tb->mu);
    return submitLocalMap(tb, n_frames, poses, P, viewing_cos_limit, log_scale_factor, th, nn_ratio, far_points, th_far_points, frustum,
                          n_to_match, assign, n_matches);
}

int ft_tracked_batch_wait(ft_tracked_batch *tb) {
    FT_REQUIRE(tb, "ft_tracked_batch_wait: null batch");
    std::lock_guard<std::mutex> lk(tb->mu);
    return waitCall(tb);
}

// the blocking forms: submit + wait
int ft_tracked_batch_search_last_frame(ft_tracked_batch *tb, int n_frames, const ft_last_points *L, const float *Tcw, float th,
                                       const int *forward, const int *backward, int check_orientation, int *const *assign,
                                       int *n_matches) {
    FT_REQUIRE(tb && Tcw && n_frames > 0, "ft_tracked_batch_search_last_frame: null batch or pose");
    std::vector<FtPose> poses;
    posesOfMatrices(Tcw, n_frames, poses);
    std::lock_guard<std::mutex> lk(tb->mu);
    int rc = submitLastFrame(tb, n_frames, L, poses.data(), nullptr, false, th, forward, backward, check_orientation, assign, n_matches);
    return rc == FT_OK ? waitCall(tb) : rc;
}

int ft_tracked_batch_search_last_frame_se3(ft_tracked_batch *tb, int n_frames, const ft_last_points *L, const ft_se3 *Tcw,
                                           const ft_se3 *Trl, float th, const int *forward, const int *backward,
                                           int check_orientation, int *const *assign, int *n_matches) {
    FT_REQUIRE(tb && Tcw && n_frames > 0, "ft_tracked_batch_search_last_frame_se3: bad argument");
    std::vector<FtPose> poses, trls;
    int rc = posesOfSe3(Tcw, Trl, n_frames, poses, trls);
    if (rc != FT_OK) return rc;
    std::lock_guard<std::mutex> lk(tb->mu);
    rc = submitLastFrame(tb, n_frames, L, poses.data(), Trl ? trls.data() : nullptr, true, th, forward, backward, check_orientation, assign,
                         n_matches);
    return rc == FT_OK ? waitCall(tb) : rc;
}

int ft_tracked_batch_track_local_map(ft_tracked_batch *tb, int n_frames, const ft_frame_pose *poses, const ft_map_points *P,
                                     float viewing_cos_limit, float log_scale_factor, float th, float nn_ratio, int far_points,
                                     float th_far_points, const ft_frustum_result *frustum, int *n_to_match, int *const *assign,
                                     int *n_matches) {
    FT_REQUIRE(tb, "ft_tracked_batch_track_local_map: null batch");
    std::lock_guard<std::mutex> lk(tb->mu);
    int rc = submitLocalMap(tb, n_frames, poses, P, viewing_cos_limit, log_scale_factor, th, nn_ratio, far_points, th_far_points, frustum,
                            n_to_match, assign, n_matches);
    return rc == FT_OK ? waitCall(tb) : rc;
}

}  // extern "C"

extern "C" {

int ft_tracked_batch_bind_fisheye(ft_tracked_batch *tb, ft_extractor *exL, ft_extractor *exR, int slot0, int n_frames, int lap_l0,
                                  int lap_l1, int lap_r0, int lap_r1, const ft_frame_view *meta, const ft_fisheye_rig *rig,
                                  const float *level_sigma2, int *const *left_to_right, int *const *right_to_left, float *const *depth,
                                  float *const *p3d, int *n_stereo) {
    return ft_tracked_batch_bind_fisheye_slots(tb, exL, exR, slot0, slot0, n_frames, lap_l0, lap_l1, lap_r0, lap_r1, meta, rig, level_sigma2,
                                               left_to_right, right_to_left, depth, p3d, n_stereo);
}

int ft_tracked_batch_bind_fisheye_slots(ft_tracked_batch *tb, ft_extractor *exL, ft_extractor *exR, int slot0, int slot0_right,
                                        int n_frames, int lap_l0, int lap_l1, int lap_r0, int lap_r1, const ft_frame_view *meta,
                                        const ft_fisheye_rig *rig, const float *level_sigma2, int *const *left_to_right,
                                        int *const *right_to_left, float *const *depth, float *const *p3d, int *n_stereo) {
    FT_REQUIRE(tb && exL && exR && meta && n_frames > 0 && n_frames <= tb->maxFrames && slot0 >= 0 && slot0_right >= 0,
               "ft_tracked_batch_bind_fisheye: bad argument");
    FT_REQUIRE(exL->ctx == tb->ctx && exR->ctx == tb->ctx, "ft_tracked_batch_bind_fisheye: extractors of another context");
    FT_REQUIRE(slot0 + n_frames <= exL->lastBatch && slot0_right + n_frames <= exR->lastBatch,
               "ft_tracked_batch_bind_fisheye: the extractors' last batches hold fewer images");
    FT_REQUIRE(exL != exR || slot0 + n_frames <= slot0_right || slot0_right + n_frames <= slot0,
               "ft_tracked_batch_bind_fisheye: one extractor for both cameras needs disjoint slot ranges");
    FT_REQUIRE(!rig || level_sigma2, "ft_tracked_batch_bind_fisheye: a rig needs level_sigma2 (mvLevelSigma2)");
    FT_REQUIRE((!depth && !p3d && !n_stereo) || rig, "ft_tracked_batch_bind_fisheye: depth / p3d / n_stereo come from the triangulation: pass a rig");
    FT_REQUIRE(!depth == !p3d, "ft_tracked_batch_bind_fisheye: depth and p3d go together");
    int nlevelsMax = 1, maxKp = 1;
    for (int f = 0; f < n_frames; f++) {
        const ft_frame_view &F = meta[f];
        FT_REQUIRE(F.Nleft >= 0 && F.N >= F.Nleft && F.N <= tb->maxKp, "ft_tracked_batch_bind_fisheye: keypoint counts out of range");
        FT_REQUIRE(F.Nleft == exL->h_nSel[slot0 + f] && F.N - F.Nleft == exR->h_nSel[slot0_right + f],
                   "ft_tracked_batch_bind_fisheye: meta's keypoint counts differ from the extractors' slots");
        FT_REQUIRE(F.scale_factors && F.nlevels >= 1 && F.nlevels <= FT_MAX_LEVELS, "scale factors missing");
        FT_REQUIRE(F.cam_model == 0 || F.cam_model == 1, "unknown camera model");
        nlevelsMax = std::max(nlevelsMax, F.nlevels);
        maxKp = std::max(maxKp, F.Nleft);
    }
    ft_context *ctx = tb->ctx;
    int rc = ft_set_device(ctx);
    if (rc != FT_OK) return rc;
    std::lock_guard<std::mutex> lk(tb->mu);
    FT_REQUIRE(tb->call.kind == 0, "ft_tracked_batch_bind_fisheye: a submitted search of this batch has not been waited for (ft_tracked_batch_wait)");
    FtTimer tAll;
    hipStream_t st = tb->stream;
    bool wantGrid;
    {
        std::lock_guard<std::mutex> lo(ctx->matchMutex);
        wantGrid = ctx->tuning.search_grid != 0;
    }
    tb->hasGrid = false;
    FT_HIP(hipStreamSynchronize(st));
    // frames region: holder_obs of all frames, the (monoLeft, monoRight) counts, then every frame's arrays (as ft_tracked_batch_upload)
    Arena a;
    tb->holderOff.assign(n_frames, 0);
    tb->holderBegin = a.off;
    for (int f = 0; f < n_frames; f++) tb->holderOff[f] = a.take(sizeof(int) * std::max(meta[f].N, 1));
    tb->holderEnd = a.off;
    const size_t oMono = a.take(sizeof(int) * 2 * (size_t)n_frames);
    const size_t oNst = a.take(sizeof(int) * (size_t)n_frames);
    struct Lay {
        size_t keys, keysR, desc, l2r, r2l, depth, p3d;
    };
    std::vector<Lay> lay(n_frames);
    for (int f = 0; f < n_frames; f++) {
        const int nL = meta[f].Nleft, nR = meta[f].N - nL;
        lay[f].keys = a.take(sizeof(ft_keypoint) * std::max(nL, 1));
        lay[f].keysR = a.take(sizeof(ft_keypoint) * std::max(nR, 1));
        lay[f].desc = a.take((size_t)32 * std::max(meta[f].N, 1));
        lay[f].l2r = a.take(sizeof(int) * std::max(nL, 1));
        lay[f].r2l = a.take(sizeof(int) * std::max(nR, 1));
        lay[f].depth = depth ? a.take(sizeof(float) * std::max(nL, 1)) : 0;   // (inside the uright / slack share of the frame's budget)
        lay[f].p3d = depth ? a.take(3 * sizeof(float) * std::max(nL, 1)) : 0;
    }
    FT_REQUIRE(a.off <= tb->framesBytes, "ft_tracked_batch_bind_fisheye: frames region too small");
    const bool wantTables = left_to_right && right_to_left;
    const bool wantOut = wantTables || depth || n_stereo;
    Arena o;
    std::vector<size_t> outL(n_frames), outR(n_frames), outD(n_frames), outP(n_frames);
    const size_t outN = o.take(sizeof(int) * (size_t)n_frames);
    for (int f = 0; f < n_frames; f++) {
        if (wantTables) {
            outL[f] = o.take(sizeof(int) * std::max(meta[f].Nleft, 1));
            outR[f] = o.take(sizeof(int) * std::max(meta[f].N - meta[f].Nleft, 1));
        }
        if (depth) {
            outD[f] = o.take(sizeof(float) * std::max(meta[f].Nleft, 1));
            outP[f] = o.take(3 * sizeof(float) * std::max(meta[f].Nleft, 1));
        }
    }
    FT_REQUIRE(o.off <= tb->outBytes, "ft_tracked_batch_bind_fisheye: result buffer too small");
    tb->nFrames = n_frames;
    tb->DF.assign(n_frames, FtDevFrame());
    uint8_t *pinF = tb->h_in + tb->workBytes, *devF = tb->d_arena + tb->oFrames;
    FtBatchJob *hJobs = (FtBatchJob *)tb->h_in;
    FtDeliverRec *hRecs = (FtDeliverRec *)(tb->h_in + (((size_t)n_frames * sizeof(FtBatchJob) + 63) & ~(size_t)63));
    // records per frame: l2r, r2l, depth, p3d (unused ones have 0 words); then the match counts; then the two tables of depth /
    // p3d pointers the triangulation kernel reads
    const int nRecs = 4 * n_frames + 1;
    float **hDepthTab = (float **)(hRecs + nRecs), **hP3dTab = hDepthTab + n_frames;
    const size_t headBytes = (size_t)((uint8_t *)(hP3dTab + n_frames) - tb->h_in);
    FT_REQUIRE(headBytes <= tb->workBytes, "ft_tracked_batch_bind_fisheye: work region too small");
    const std::function<void(int, int)> stage = [&](int f, int) {
        const ft_frame_view &F = meta[f];
        const int nL = F.Nleft, nR = F.N - nL;
        int *hold = (int *)(pinF + tb->holderOff[f]);
        for (int i = 0; i < F.N; i++) hold[i] = F.holder_obs ? F.holder_obs[i] : -1;
        FtDevFrame &D = tb->DF[f];
        D = devFrameConstants(&F);
        D.keys = (const ft_keypoint *)(devF + lay[f].keys);
        D.keysR = (const ft_keypoint *)(devF + lay[f].keysR);
        D.desc = devF + lay[f].desc;
        D.uright = nullptr;
        D.holderObs = (const int *)(devF + tb->holderOff[f]);
        D.l2r = (const int *)(devF + lay[f].l2r);
        D.r2l = (const int *)(devF + lay[f].r2l);
        if (wantGrid) {
            int *grid = (int *)(tb->d_arena + tb->oGrid + (size_t)f * tb->gridStride);
            float4 *rec = (float4 *)((uint8_t *)grid + gridIntBytes(D.N));
            uint8_t *gdesc = (uint8_t *)(rec + std::max(D.N, 1));
            D.gridStart[0] = grid;
            D.gridStart[1] = grid + (size_t)FT_MAX_LEVELS * (FT_GRID_CELLS + 1);
            D.gridRec[0] = rec;
            D.gridDesc[0] = gdesc;
            D.gridRec[1] = rec + nL;
            D.gridDesc[1] = gdesc + (size_t)32 * nL;
        }
        memset(&hJobs[f], 0, sizeof(FtBatchJob));
        hJobs[f].F = D;
        FtDeliverRec *R = hRecs + 4 * (size_t)f;
        memset(R, 0, 4 * sizeof(FtDeliverRec));
        if (wantTables) {
            R[0].dst = tb->h_out + outL[f];
            R[0].src[0] = R[0].src[1] = D.l2r;
            R[0].words = nL;
            R[1].dst = tb->h_out + outR[f];
            R[1].src[0] = R[1].src[1] = D.r2l;
            R[1].words = nR;
        }
        hDepthTab[f] = hP3dTab[f] = nullptr;
        if (depth) {
            hDepthTab[f] = (float *)(devF + lay[f].depth);
            hP3dTab[f] = (float *)(devF + lay[f].p3d);
            R[2].dst = tb->h_out + outD[f];
            R[2].src[0] = R[2].src[1] = hDepthTab[f];
            R[2].words = nL;
            R[3].dst = tb->h_out + outP[f];
            R[3].src[0] = R[3].src[1] = hP3dTab[f];
            R[3].words = 3 * nL;
        }
    };
    ctx->pool->parallel_for(n_frames, stage);
    hRecs[4 * (size_t)n_frames].dst = tb->h_out + outN;
    hRecs[4 * (size_t)n_frames].src[0] = hRecs[4 * (size_t)n_frames].src[1] = devF + oNst;
    hRecs[4 * (size_t)n_frames].words = n_frames;
    FT_HIP(hipMemcpyAsync(tb->d_arena + tb->oWork, tb->h_in, headBytes, hipMemcpyHostToDevice, st));
    if (tb->holderEnd > tb->holderBegin)
        FT_HIP(hipMemcpyAsync(devF + tb->holderBegin, pinF + tb->holderBegin, tb->holderEnd - tb->holderBegin, hipMemcpyHostToDevice, st));
    FT_HIP(hipEventRecord(tb->evMirror, st));
    const FtBatchJob *dJobs = (const FtBatchJob *)(tb->d_arena + tb->oWork);
    FtBindArgs A;
    A.keysL = exL->d_keys; A.keysR = exR->d_keys;
    A.descL = exL->d_desc; A.descR = exR->d_desc;
    A.strideL = exL->geom.maxKp; A.strideR = exR->geom.maxKp;
    A.slot0L = slot0;
    A.slot0R = slot0_right;
    A.lapL0 = lap_l0; A.lapL1 = lap_l1; A.lapR0 = lap_r0; A.lapR1 = lap_r1;
    A.mono = (int *)(devF + oMono);
    A.triangulate = rig ? 1 : 0;
    memset(&A.rig, 0, sizeof A.rig);
    if (rig) {
        memcpy(A.rig.cam1, rig->cam1, sizeof A.rig.cam1);
        memcpy(A.rig.cam2, rig->cam2, sizeof A.rig.cam2);
        A.rig.precision = rig->precision;
        memcpy(A.rig.Rlr, rig->Rlr, sizeof A.rig.Rlr);
        memcpy(A.rig.tlr, rig->tlr, sizeof A.rig.tlr);
        for (int i = 0; i < nlevelsMax; i++) A.rig.sigma2[i] = level_sigma2[i];
    }
    A.nMatches = (int *)(devF + oNst);
    const uint8_t *dWork = tb->d_arena + tb->oWork;
    A.depth = depth ? (float *const *)(dWork + ((uint8_t *)hDepthTab - tb->h_in)) : nullptr;
    A.p3d = depth ? (float *const *)(dWork + ((uint8_t *)hP3dTab - tb->h_in)) : nullptr;
    tb->evt.begin(ctx->kernelTiming, "kernel.lap_gather+fisheye_2nn_batch", st);
    rc = ft_launch_bind_fisheye_batch(st, tb->d_arena, dJobs, n_frames, maxKp, A);
    tb->evt.end(ctx->kernelTiming, st);
    if (rc == FT_OK && rig) {
        tb->evt.begin(ctx->kernelTiming, "kernel.fisheye_triangulate_batch", st);
        rc = ft_launch_fisheye_triangulate_batch(st, tb->d_arena, dJobs, n_frames, maxKp, A);
        tb->evt.end(ctx->kernelTiming, st);
    }
    if (rc != FT_OK) return rc;
    // the extractors' slots have been read: their next batch (which overwrites them) is ordered behind this point - the call may
    // return before the gather has run (no outputs asked for), and nothing else ties the extractors' streams to this one
    FT_HIP(hipEventRecord(tb->evGather, st));
    exL->foreignReader = tb->evGather;
    exR->foreignReader = tb->evGather;
    tb->evt.begin(ctx->kernelTiming, "kernel.build_grid_batch", st);
    if (rc == FT_OK && wantGrid) {
        rc = ft_launch_build_grid_batch(st, tb->d_arena, dJobs, n_frames, nlevelsMax, true);
        if (rc == FT_OK) tb->hasGrid = true;
    }
    tb->evt.end(ctx->kernelTiming, st);
    if (rc != FT_OK) return rc;
    if (wantOut) {
        const FtDeliverRec *dRecs = (const FtDeliverRec *)(tb->d_arena + tb->oWork + ((uint8_t *)hRecs - tb->h_in));
        rc = ft_launch_deliver_batch(st, dRecs, nRecs, 3 * maxKp, 0);
        if (rc != FT_OK) return rc;
        FT_HIP(hipStreamSynchronize(st));
        for (int f = 0; f < n_frames; f++) {
            const int nL = meta[f].Nleft, nR = meta[f].N - nL;
            if (wantTables && left_to_right[f] && nL) memcpy(left_to_right[f], tb->h_out + outL[f], sizeof(int) * nL);
            if (wantTables && right_to_left[f] && nR) memcpy(right_to_left[f], tb->h_out + outR[f], sizeof(int) * nR);
            if (depth && depth[f] && nL) memcpy(depth[f], tb->h_out + outD[f], sizeof(float) * nL);
            if (depth && p3d[f] && nL) memcpy(p3d[f], tb->h_out + outP[f], 3 * sizeof(float) * nL);
            if (n_stereo) n_stereo[f] = ((const int *)(tb->h_out + outN))[f];
        }
    }
    ctx->addStat("tracked_batch.bind_fisheye.total", tAll.ms());
    return FT_OK;
}

}  // extern "C"
