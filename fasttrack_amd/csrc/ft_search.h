// Device-side views used by the projection-search kernels (kernels_search.hip / search.cpp).
#pragma once

#include "ft_internal.h"

struct FtDevFrame {
    int N, Nleft;
    float mnMinX, mnMinY, mnMaxX, mnMaxY, invW, invH, mbf, mb;
    const ft_keypoint *keys, *keysR;  // device
    const uint8_t *desc;              // device N x 32
    const float *uright;              // device or null
    const int *holderObs;             // device [N] pre-call Observations() of mvpMapPoints[i] (-1 = none)
    const int *l2r, *r2l;             // device or null
    int camModel;
    float cam[8];
    float Trl[12];
    float TrlQ[4];  // trlQuat != 0: GetRelativePoseTrl() in the Sophus form (translation in Trl[3], [7], [11])
    int trlQuat;
    float sf[FT_MAX_LEVELS];
    int nlevels;
    // Frame::mGrid as one CSR per octave (k_build_grid; AssignFeaturesToGrid src/Frame.cc:409-440): the keypoints of octave o
    // in cell (cx, cy) are the entries gridStart[o * (FT_GRID_CELLS + 1) + cx * 48 + cy] .. [... + 1), so the part of a
    // window's column cx that lies in the search's level band is ONE contiguous range per octave; [0] left camera (or the
    // only one), [1] right camera.  Null = no grid: the searches then scan every keypoint.
    const int *gridStart[2];
    // the entries as search records {x, y, uright (or -1), index | octave << 24} and 32-byte descriptors - what a window
    // scan needs of a keypoint, as two contiguous reads per run of entries instead of the chain grid entry -> keypoint ->
    // descriptor / uright
    const float4 *gridRec[2];
    const uint8_t *gridDesc[2];
};
#define FT_GRID_CELLS (FT_GRID_COLS * FT_GRID_ROWS)

struct FtDevLocalPoints {
    int M;
    const uint8_t *skip, *inView, *inViewR;
    const int *level, *levelR;
    const float *viewCos, *viewCosR, *projX, *projY, *projXR, *projYR;
    const uint8_t *desc;
};

struct FtDevLastPoints {
    int N;
    const uint8_t *valid;
    const float *worldPos;
    const uint8_t *desc;
    const int *octave;
    const float *angle;  // batch form only: the last-frame keypoints' angles (rotation histogram of k_replay_batch), may be null
};

// The claim iteration (Jacobi passes over the sequential claiming of SearchByProjection): a pass reads the writer lists
// the previous pass built - head[kp] -> slot s (= 4 * point + write kind), next[s] - and builds the lists of the next
// pass while it runs, so a pass is ONE launch.  The list heads rotate through three arrays (read / write / clear), next
// and the results through two.  Flags: -1 = "this pass changed nothing", 0 = "changed" (one memset with 0xff prepares
// heads and flags of a call); a pass returns at once when the previous pass's flag says the fixed point was reached.
struct FtClaims {
    const int *head, *next;  // writer lists of the previous pass
    const int *obs;          // Observations() per map point
    int firstPass;           // no lists yet: the pre-call holders decide, and every result counts as changed
    int *headWrite, *nextWrite;  // lists this pass builds for the next one
    int *headClear;          // the heads the next pass will write: reset to -1 here
    // the writer table (kernels_search.hip, FT_TAB_ENTRIES): 8 ints per keypoint, rotating like the heads; head / next
    // only take the writers a record has no room for
    const int *tab;
    int *tabWrite, *tabClear;
    int nKp;
    const int *resPrev;      // results of the previous pass
    int *flagCur;            // this pass's flag (atomicAnd 0 on a change)
    const int *flagPrev;     // the previous pass's flag (null for the first pass of a burst)
    int *flagReset;          // the flag of the same position in the other burst parity: reset to -1 here
    int *flagStick = nullptr;  // batch form only: the 16 flag words of this burst (claims_begin_pass, kernels_search.hip)
    // Candidate cache (may be null).  What does NOT change from pass to pass - which keypoints of a point's window pass the
    // level band, the box and the uright test, and their Hamming distances - is computed once: the first pass that reaches a
    // (point, camera) window files the (distance, cell x, cell y, index) keys of all its candidates here, and every later
    // pass only walks those keys against the new lock state (a round of 64 keys = one coalesced load + the lock look-ups)
    // instead of walking the grid, the keypoints and the descriptors again.  A pass of the iteration is as long as its
    // slowest wave, and a window scan is a chain of ~5 dependent loads per 64 entries: 85 us per pass at th 15, whatever
    // the number of points.  Per (point, camera) FT_CACHE_CAP + 1 64-bit words: [0] = ~0 "not built" | number of
    // candidates (more than FT_CACHE_CAP: not cached, scan again) | any keypoint in the box << 32, then the keys, unordered.
    unsigned long long *cache;
};
#ifndef FT_CACHE_CAP
#define FT_CACHE_CAP 511
#endif
#define FT_CACHE_WORDS (2 * (FT_CACHE_CAP + 1))  // per point: left and right camera

// a rigid transform: row-major 3x4 (y = R x + t), or - quat != 0 - as Sophus::SE3f holds and applies it: unit quaternion
// q = (x, y, z, w), translation in m[3], m[7], m[11] (transform_pose, kernels_search.hip)
struct FtPose {
    float m[12];
    float q[4];
    int quat;
};

struct FtLocalRaw {
    int *bestDist, *bestDist2, *bestLevel, *bestLevel2, *bestIdx;
    int *bestDistR, *bestDist2R, *bestLevelR, *bestLevel2R, *bestIdxR;
};
struct FtLastRaw {
    int *bestDist, *bestIdx, *bestDistR, *bestIdxR;
};

// Frame::isInFrustum inputs / outputs (kernels_search.hip k_frustum)
struct FtDevMapPoints {
    int M;
    const uint8_t *skip;  // may be null
    const float *worldPos, *normal, *maxDist, *minDist;
};
struct FtFrustumPose {
    float R[2][9], t[2][3], twc[2][3];  // [0] left camera (mRcw, mtcw, mOw), [1] right camera (Frame.cc:1314-1320)
};
struct FtFrustumOut {
    uint8_t *inView, *inViewR;
    int *level, *levelR;
    float *viewCos, *viewCosR, *projX, *projY, *projXR, *projYR, *depth, *depthR;
    uint8_t *searchSkip;  // may be null: the `continue` conditions of ORBmatcher.cc:66-74 for the search that follows
    int *count;           // nToMatch (atomic; zeroed by the launcher)
};

// One frame of a batch of searches (ft_tracked_batch, search.cpp): what the batch kernels read of frame blockIdx.y, resident
// in HBM.  Every pointer of a job points into the device arena of its batch (`arena` of the launchers below: the kernels
// re-derive the pointers from it, see Rebase in kernels_search.hip).  The rotating buffers of the claim iteration are addressed by pass number (job_claims, kernels_search.hip):
// res 2 x 4 nPoints | head 3 x K | next 2 x 4 nPoints | tab 3 x 8 K (K = keypoints rounded up to 8) | flags FT_BATCH_FLAGS.
// projection of a last-frame point into the current frame's camera(s) (k_last_project_batch, kernels_search.hip)
struct FtLastProj {
    float u, v, invzc, ur, vr;
    int go;  // projects into the image (src/ORBmatcher.cc:1807-1822)
};
#define FT_BATCH_FLAGS 64  // flag words per frame of a batch: two burst parities x 32 positions (a power of two)
struct FtBatchJob {
    FtDevFrame F;
    int *res, *head, *next, *tab, *flags;
    int *slow;                  // [0], [1]: lengths of the slow lists of even / odd passes, then 2 x nPoints entries from word 16
    const int *obs;             // Observations() per point of the running search
    unsigned long long *cache;  // FT_CACHE_WORDS per point, or null
    int K, nKp, nPoints;
    // SearchByProjection(CurrentFrame, LastFrame)
    FtDevLastPoints L;
    FtPose Tcw;
    int forward, backward;
    FtLastProj *proj;  // [L.N], written by the first pass's projection launch
    // isInFrustum + SearchByProjection(Frame, local map points)
    FtDevMapPoints MP;
    FtFrustumPose T;
    FtFrustumOut O;
    FtDevLocalPoints P;
    // what the search leaves behind (k_replay_batch): assignOut [F.N] and nmOut [1] are PINNED HOST memory the kernel writes
    // directly (not in the arena: never rebased), replayed [1] (arena) is the frame's "done" marker: -1 until the frame's writes
    // have been replayed, then its match count
    int *assignOut, *nmOut, *replayed;
    int *err;  // [1] (arena), zeroed with the claim buffers: an input error a kernel found (FT_JOB_ERR_*), reported by the call's second half
};
#define FT_JOB_ERR_OCTAVE 1  // a valid last-frame point whose octave lies outside the frame's levels (the point is dropped)

// a block of bytes of PINNED HOST memory (the caller's arrays, read in place) gathered into the batch's arena by k_gather_batch
struct FtGatherRec {
    void *dst;        // device (arena)
    const void *src;  // pinned host
    unsigned bytes;
};
int ft_launch_gather_batch(hipStream_t st, const FtGatherRec *recs, int nRecs);
// a block of dwords delivered into pinned host memory by k_deliver_batch; src[parity of the last pass]
struct FtDeliverRec {
    void *dst;
    const void *src[2];
    int words;
};
// what two extractors left in HBM (slot slot0L + f / slot0R + f = left / right image of frame f of the batch), the lapping areas, and the
// per-frame (monoLeft, monoRight) counts the gather leaves for the matching
struct FtBindArgs {
    const ft_keypoint *keysL, *keysR;
    const uint8_t *descL, *descR;
    int strideL, strideR;  // keypoints per slot
    int slot0L, slot0R;  // first slot of the left / right extractor (one extractor may serve both cameras: disjoint slot ranges)
    int lapL0, lapL1, lapR0, lapR1;
    int *mono;  // [nFrames][2], device
    // the triangulation filter of Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1256-1271): 0 = the matching alone
    int triangulate;
    FtFisheyeRig rig;
    int *nMatches;      // [nFrames], device (zeroed by the gather)
    float *const *depth, *const *p3d;  // device tables of per-frame device arrays (mvDepth [Nleft], mvStereo3Dpoints [3 Nleft]), or null
};
// keypoints / descriptors of every frame into the reference's order, then the 2-NN + ratio matching of the lapping subsets
int ft_launch_bind_fisheye_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxKp, const FtBindArgs &A);
// KannalaBrandt8::TriangulateMatches for every ratio-test survivor of every frame (kernels_match.hip): keeps mvLeftToRightMatch
// where the depth is > 0.0001, fills mvRightToLeftMatch, mvDepth, mvStereo3Dpoints and the frame's match count
int ft_launch_fisheye_triangulate_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxKp, const FtBindArgs &A);
int ft_launch_fill_claims_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxWords);
// behind the first pass of a batch: every candidate list with its best candidates at the front (cache_partition, kernels_search.hip)
int ft_launch_cache_partition_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxPoints);
int ft_launch_deliver_batch(hipStream_t st, const FtDeliverRec *recs, int nRecs, int maxWords, int parity);
int ft_launch_build_grid_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxLevels, bool twoCam);
int ft_launch_frustum_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxM, float viewingCosLimit, float logScaleFactor,
                            int farPoints, float thFar);
// one pass of the claim iteration of every frame; fCur / fPrev / fReset: positions in each frame's flag words (fPrev < 0: none)
int ft_launch_search_last_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxPoints, int pass, int fCur, int fPrev,
                                int fReset, float th);
int ft_launch_search_local_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxPoints, int pass, int fCur, int fPrev,
                                 int fReset, float th, float nnRatio);
// the first pass with four points per wave (a point = a row of 16 lanes): every frame needs its grid and the candidate cache;
// fills the lists, results and writer table exactly as pass 0 of ft_launch_search_*_batch does
int ft_launch_search_last_first(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxPoints, float th);
int ft_launch_search_local_first(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxPoints, float th, float nnRatio);
// everything behind that first pass in one launch: a workgroup per frame walks the frame's points in index order (k_resolve_batch);
// a frame it resolves has all its flag words at -1 and its results in both result buffers, a frame it gives up on is untouched
// sharedInts = ints of LDS for the frame's last-writer table (>= the largest F.N; <= 12288), 0 = frames beyond that: the table lives in HBM
int ft_launch_resolve_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int local, float nnRatio, int sharedInts);
// The writes of a converged search replayed on the device, a workgroup per frame (k_replay_batch): assign[keypoint] = the last
// point that wrote it, holder_obs updated in place in HBM, the match count; last frame: with the rotation histogram and
// ComputeThreeMaxima when checkOrientation.  Only frames that have not been replayed yet and - flagPos >= 0 - whose flag word at that
// position says "converged" (a launch enqueued before the host has looked: behind k_resolve_batch position 0, behind a burst of
// claim passes its last position; -1: every frame); parity = result buffer of the last pass (a converged frame holds its results in both).  sharedInts = ints of LDS per workgroup
// for the last-writer table (>= the largest F.N), 0 = frames too large for the LDS: the table lives in the frame's writer table
int ft_launch_replay_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int local, int parity, int checkOrientation,
                           int sharedInts, int flagPos);
// a pass behind the first one: lean kernel (four points per wave from the candidate cache) + the general kernel on its slow list
int ft_launch_search_last_batch_lean(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxPoints, int pass, int fCur,
                                     int fPrev, int fReset, float th);
int ft_launch_search_local_batch_lean(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxPoints, int pass, int fCur,
                                      int fPrev, int fReset, float th, float nnRatio);
int ft_launch_fill_i32(hipStream_t st, int *p, int n, int v);
// p[0 .. n) = -1 and meta[i * strideWords] = ~0 for i < nMeta (meta may be null): the start of a claim iteration, one launch
int ft_launch_fill_claims(hipStream_t st, int *p, int n, unsigned long long *meta, int nMeta, int strideWords);
// p[i * strideWords] = v for i < n (64-bit words): the meta words of the candidate cache
int ft_launch_fill_stride_u64(hipStream_t st, unsigned long long *p, int n, int strideWords, unsigned long long v);
// one kernel copies up to three device blocks (sizes rounded up to dwords) into pinned host memory
int ft_launch_deliver_blocks(hipStream_t st, void *d0, const void *s0, size_t bytes0, void *d1, const void *s1, size_t bytes1,
                             void *d2, const void *s2, size_t bytes2);
int ft_launch_features_in_area(hipStream_t st, const FtDevFrame &F, int nq, const float *qx, const float *qy, const float *qr,
                               const int *qmin, const int *qmax, const uint8_t *qright, const int *offsets,
                               unsigned *outKeys, int *outCount);
int ft_launch_frustum(hipStream_t st, const FtDevFrame &F, const FtFrustumPose &T, const FtDevMapPoints &P,
                      float viewingCosLimit, float logScaleFactor, int farPoints, float thFar, const FtFrustumOut &O);
int ft_launch_search_local(hipStream_t st, const FtDevFrame &F, const FtDevLocalPoints &P, const FtClaims &C, float th,
                           float nnRatio, int *res, const FtLocalRaw &raw);
int ft_launch_search_last(hipStream_t st, const FtDevFrame &F, const FtDevLastPoints &L, const FtClaims &C,
                          const FtPose &Tcw, float th, int forward, int backward, int *res, const FtLastRaw &raw);
int ft_launch_build_grid(hipStream_t st, const FtDevFrame &F, int *gridStartL, int *gridStartR, float4 *recL, uint8_t *descL,
                         float4 *recR, uint8_t *descR);

#ifdef __HIPCC__
// Re-derives a pointer read from a job record from the arena pointer the kernel got as an argument (kernels_search.hip: a
// pointer out of memory is a generic pointer to the compiler): arena + (p - address of the arena, passed as an integer)
struct Rebase {
    uint8_t *arena;
    unsigned long long addr;  // (unsigned long long)arena
    template <class T>
    __device__ __forceinline__ T *operator()(T *p) const {
        return p ? (T *)(arena + ((unsigned long long)p - addr)) : nullptr;
    }
};
#endif
