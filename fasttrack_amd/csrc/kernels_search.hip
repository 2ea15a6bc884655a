// HIP kernels of the two projection searches for gfx950 (wave64).
//   k_search_local  ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>, th, ...)  (reference
//                   src/ORBmatcher.cc:49-225 + Frame::GetFeaturesInArea src/Frame.cc:681-747)
//   k_search_last   ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono)  (:1775-1960)
//
// One wave per map point; lanes stride over the frame's keypoints and test the 64x48-grid cell window,
// the level band and the box exactly as GetFeaturesInArea does, so no per-cell lists are needed (the
// reference's fixed 20-per-cell matrix overflows).  The scan order of the CPU loop,
// (cell x, cell y, keypoint index), is folded into a 64-bit key (distance, cx, cy, index): the two
// smallest keys of the wave are the CPU's (best, second best) including which octaves they carry.
//
// In-call claiming (a keypoint taken by an earlier map point with Observations() > 0 is skipped by
// later ones, ORBmatcher.cc:101-103,142) makes the CPU loop sequential.  It is reproduced exactly by a
// Jacobi iteration on that triangular dependency: every pass recomputes all points in parallel
// against the writes of the previous pass (per-keypoint linked lists of writers); point i is final
// after at most i+1 passes and the iteration stops when a pass changes nothing - the unique fixed
// point is the sequential result.
#include <algorithm>

#include "kb8_math.h"
#include "libm_f32.h"
#include "ft_search.h"
#include "wave_ops.h"

namespace {

__device__ __forceinline__ int hamming256(const unsigned long long a[4], const unsigned long long *b) {
    return __popcll(a[0] ^ b[0]) + __popcll(a[1] ^ b[1]) + __popcll(a[2] ^ b[2]) + __popcll(a[3] ^ b[3]);
}

// The array pointers of a frame as the scans use them.  The kernels of ONE frame get the frame as a kernel argument and
// its pointers are global pointers to the compiler; a pointer read from a record in memory (the job records of the batch
// kernels, FtBatchJob) is a GENERIC pointer - flat_load: both wait counters, an aperture check per access - and neither
// a cast through the global address space nor llvm.assume(!is_shared && !is_private) survives the optimiser.  What does:
// re-deriving the pointer from a pointer that IS a kernel argument - the batch's device arena, with the arena's address
// passed a second time as a plain integer, so that arena + (p - address) cannot be folded back into p.  Everything a job
// record points to lies inside the arena of its batch (search.cpp).
struct NoRebase {  // the pointers are kernel arguments already
    template <class T>
    __device__ __forceinline__ T *operator()(T *p) const {
        return p;
    }
};
struct FramePtrs {
    const ft_keypoint *keys, *keysR;
    const uint8_t *desc;
    const float *uright;
    const int *holderObs, *l2r, *r2l;
    const int *gridStart[2];
    const float4 *gridRec[2];
    const uint8_t *gridDesc[2];
};
template <class RB>
__device__ __forceinline__ FramePtrs frame_ptrs(const FtDevFrame &F, const RB &rb) {
    FramePtrs Q;
    Q.keys = rb(F.keys); Q.keysR = rb(F.keysR); Q.desc = rb(F.desc); Q.uright = rb(F.uright);
    Q.holderObs = rb(F.holderObs); Q.l2r = rb(F.l2r); Q.r2l = rb(F.r2l);
    Q.gridStart[0] = rb(F.gridStart[0]); Q.gridStart[1] = rb(F.gridStart[1]);
    Q.gridRec[0] = rb(F.gridRec[0]); Q.gridRec[1] = rb(F.gridRec[1]);
    Q.gridDesc[0] = rb(F.gridDesc[0]); Q.gridDesc[1] = rb(F.gridDesc[1]);
    return Q;
}
#define FT_NO_REBASE (NoRebase{})

#define KEY_NONE 0xffffffffffffffffull
// Candidate key: (distance, cell x, cell y, index) in the high bits - ascending keys are the scan order of the CPU loop, see
// the header - and below them what a later pass would otherwise have to fetch again through dependent loads: the keypoint's
// octave (four bits: checkFrame, search.cpp, admits octaves of [0, nlevels) only; masked here so that the keypoints of a
// BOUND frame, which no host check sees, can never spill into the index) and whether it was held before the call
// (mvpMapPoints[idx]->Observations() > 0).  The index is unique inside a window, so the low bits never decide a comparison.
__device__ __forceinline__ unsigned long long make_key(int dist, int cx, int cy, int idx, int octave, bool heldBefore) {
    return ((unsigned long long)dist << 41) | ((unsigned long long)cx << 35) | ((unsigned long long)cy << 29) |
           ((unsigned long long)idx << 5) | ((unsigned long long)(octave & 15) << 1) | (heldBefore ? 1ull : 0ull);
}
__device__ __forceinline__ int key_dist(unsigned long long k) { return (int)(k >> 41); }
__device__ __forceinline__ int key_idx(unsigned long long k) { return (int)((k >> 5) & 0xffffffull); }
__device__ __forceinline__ int key_octave(unsigned long long k) { return (int)((k >> 1) & 15ull); }
__device__ __forceinline__ bool key_held(unsigned long long k) { return (k & 1ull) != 0; }

// The words one pass of the claim iteration hands to the next - writer records, lists, results, flags - are written and read by
// DIFFERENT workgroups, but of DIFFERENT launches: a pass reads what the previous launch wrote (its own L1 starts empty) and
// writes what the next launch reads, never a word it also reads.  Plain loads and stores therefore do (rounds 3 - 4 kept them
// agent-scope atomics, sc1, for the sake of the persistent single-launch form, deleted in round 5): a record is two 16-byte
// loads that the L1 may keep for the other points of the CU that look at the same keypoint, a clear is 16-byte stores.  Only the
// read-modify-writes are atomics (record positions, list heads, the "changed" flag).
__device__ __forceinline__ int shared_load(const int *p) { return *p; }
__device__ __forceinline__ void shared_store(int *p, int v) { *p = v; }

// Writer table of a pass: per keypoint a 32-byte record {last position handed out, 7 entries}, entry = (4 * point + write
// kind) << 1 | (Observations() of the point > 0), -1 = empty; an eighth and later writer of one keypoint (never seen outside
// directed tests) goes to the overflow lists head / next, which hold the same entries.  One 32-byte read tells a later
// pass everything about a keypoint - where the linked lists of rounds 1-3 cost a dependent load per writer plus one for the
// writer's Observations() (a later pass is nothing but a chain of such round trips, ~1 us each).
#define FT_TAB_ENTRIES 7
// F.mvpMapPoints[kp] && ->Observations() > 0 as seen by map point i: the last writer j < i of the previous pass decides,
// else the pre-call holder (heldBefore)
struct LockRec {
    unsigned long long a, b, c, d;
};
__device__ __forceinline__ LockRec lock_record(const FtClaims &C, int kp) {
    const uint4 *rec = (const uint4 *)(C.tab + 8 * (size_t)kp);
    const uint4 lo = rec[0], hi = rec[1];
    LockRec r;
    r.a = (unsigned long long)lo.x | ((unsigned long long)lo.y << 32);
    r.b = (unsigned long long)lo.z | ((unsigned long long)lo.w << 32);
    r.c = (unsigned long long)hi.x | ((unsigned long long)hi.y << 32);
    r.d = (unsigned long long)hi.z | ((unsigned long long)hi.w << 32);
    return r;
}
// the decision of is_locked on a record that is already in registers (not a first pass)
__device__ __forceinline__ bool locked_by(const FtClaims &C, const LockRec &r, int kp, int i, bool heldBefore) {
    const unsigned long long a = r.a, b = r.b, c = r.c, d = r.d;
    const int last = (int)(unsigned)a;
    int best = -1;
    auto take = [&](int e) {
        if (e >= 0 && (e >> 3) < i && e > best) best = e;
    };
    take((int)(a >> 32)); take((int)(unsigned)b); take((int)(b >> 32)); take((int)(unsigned)c);
    take((int)(c >> 32)); take((int)(unsigned)d); take((int)(d >> 32));
    if (last >= FT_TAB_ENTRIES)
        for (int e = shared_load(&C.head[kp]); e >= 0; e = shared_load(&C.next[e >> 1])) take(e);
    return best >= 0 ? (best & 1) != 0 : heldBefore;
}
__device__ __forceinline__ bool is_locked(const FtClaims &C, int kp, int i, bool heldBefore) {
    if (C.firstPass) return heldBefore;
    return locked_by(C, lock_record(C, kp), kp, i, heldBefore);
}

// start of a claim-iteration pass (see FtClaims): false = the iteration has converged, nothing to do
// (blk of nblk: this workgroup among the frame's - the launch's own numbers unless the launcher laid the frames out itself)
__device__ __forceinline__ bool claims_begin_pass(const FtClaims &C, int blk, int nblk) {
    if (C.flagPrev && shared_load(C.flagPrev) == -1) {
        // batch form, first pass of a later burst: the frame had converged before this burst began.  Its flag words of this
        // burst's parity still hold what an earlier burst left there ("changed" for the passes it ran then): they all read
        // "unchanged" from here on, so that every later pass of the burst returns here as well.
        if (C.flagStick && blk == 0 && threadIdx.x < FT_BATCH_FLAGS / 2) shared_store(C.flagStick + threadIdx.x, -1);
        return false;
    }
    const int t = blk * blockDim.x + threadIdx.x, T = nblk * blockDim.x;
    for (int k = t; k < C.nKp; k += T) shared_store(&C.headClear[k], -1);
    uint4 *tc = (uint4 *)C.tabClear;  // (32-byte records, 32-byte aligned)
    for (int k = t; k < 2 * C.nKp; k += T) tc[k] = make_uint4(~0u, ~0u, ~0u, ~0u);
    if (t == 0) {
        shared_store(C.flagReset, -1);
        if (C.firstPass) shared_store(C.flagCur, 0);  // the first pass always "changes" its input
    }
    return true;
}
__device__ __forceinline__ bool claims_begin_pass(const FtClaims &C) { return claims_begin_pass(C, (int)blockIdx.x, (int)gridDim.x); }

// end of a point's turn in a pass: lane k files write kind k of point i - the result, the "changed" flag against the
// previous pass, and the entry in the writer table the NEXT pass will read (so a pass is one launch)
__device__ __forceinline__ void claims_file(const FtClaims &C, int *res, int i, int lane, const int r4[4]) {
    if (lane < 4) {
        const int kp = lane == 0 ? r4[0] : lane == 1 ? r4[1] : lane == 2 ? r4[2] : r4[3];
        const int s = 4 * i + lane;
        const int prev = C.firstPass ? 0 : shared_load(&C.resPrev[s]);
        if (!C.firstPass && kp != prev) atomicAnd(C.flagCur, 0);
        shared_store(&res[s], kp);
        if (kp >= 0) {
            const int e = (s << 1) | (C.obs[i] > 0 ? 1 : 0);
            int *rec = C.tabWrite + 8 * (size_t)kp;
            const int pos = atomicAdd(rec, 1) + 1;  // the record starts at -1
            if (pos < FT_TAB_ENTRIES) shared_store(rec + 1 + pos, e);
            else shared_store(&C.nextWrite[s], atomicExch(&C.headWrite[kp], e));
        }
    }
}

__device__ __forceinline__ unsigned div_magic_u(int d) { return d > 1 ? 0xffffffffu / (unsigned)d + 1u : 0u; }

struct Window {
    int minCX, maxCX, minCY, maxCY;
    bool empty;
};

// Frame::GetFeaturesInArea cell window (src/Frame.cc:689-711)
__device__ __forceinline__ Window cell_window(const FtDevFrame &F, float x, float y, float r) {
    Window w;
    w.empty = false;
    w.minCX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, F.mnMinX), r), F.invW)));
    if (w.minCX >= FT_GRID_COLS) w.empty = true;
    w.maxCX = min(FT_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, F.mnMinX), r), F.invW)));
    if (w.maxCX < 0) w.empty = true;
    w.minCY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, F.mnMinY), r), F.invH)));
    if (w.minCY >= FT_GRID_ROWS) w.empty = true;
    w.maxCY = min(FT_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, F.mnMinY), r), F.invH)));
    if (w.maxCY < 0) w.empty = true;
    return w;
}

// geometric part of GetFeaturesInArea for one keypoint: grid cell (Frame::PosInGrid, :749-759) inside
// the window, level band, box test.  Returns false when the keypoint is not a candidate.
__device__ __forceinline__ bool in_area(const FtDevFrame &F, const ft_keypoint &kp, const Window &w, float x, float y,
                                        float r, int minLevel, int maxLevel, int &cx, int &cy) {
    cx = (int)roundf(__fmul_rn(__fsub_rn(kp.x, F.mnMinX), F.invW));
    cy = (int)roundf(__fmul_rn(__fsub_rn(kp.y, F.mnMinY), F.invH));
    if (cx < 0 || cx >= FT_GRID_COLS || cy < 0 || cy >= FT_GRID_ROWS) return false;  // never entered the grid
    if (cx < w.minCX || cx > w.maxCX || cy < w.minCY || cy > w.maxCY) return false;
    const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
    if (checkLevels) {
        if (kp.octave < minLevel) return false;
        if (maxLevel >= 0 && kp.octave > maxLevel) return false;
    }
    const float dx = __fsub_rn(kp.x, x), dy = __fsub_rn(kp.y, y);
    return fabsf(dx) < r && fabsf(dy) < r;
}

// Frame::GetFeaturesInArea (src/Frame.cc:681-747) for a batch of queries: one wave per query, lanes stride over the
// keypoints of the requested camera; hits are appended as (cell x, cell y, index) keys whose ascending order is the
// order of the reference's nested cell loops (the host sorts the few hits of a query).
__global__ __launch_bounds__(256) void k_features_in_area(FtDevFrame F, int nq, const float *qx, const float *qy, const float *qr,
                                                          const int *qmin, const int *qmax, const uint8_t *qright,
                                                          const int *offsets, unsigned *outKeys, int *outCount) {
    const int lane = threadIdx.x & 63, wave = wave_index();
    const int q = blockIdx.x * 4 + wave;
    if (q >= nq) return;
    const float x = qx[q], y = qy[q], r = qr[q];
    const int minLevel = qmin[q], maxLevel = qmax[q];
    const bool right = qright && qright[q];
    const int n = F.Nleft == -1 ? F.N : (right ? F.N - F.Nleft : F.Nleft);
    const ft_keypoint *keys = (F.Nleft != -1 && right) ? F.keysR : F.keys;
    const Window w = cell_window(F, x, y, r);
    int count = 0;
    if (!w.empty) {
        for (int base = 0; base < n; base += 64) {
            const int idx = base + lane;
            bool hit = false;
            int cx = 0, cy = 0;
            if (idx < n) hit = in_area(F, keys[idx], w, x, y, r, minLevel, maxLevel, cx, cy);
            const unsigned long long b = __ballot(hit);
            if (hit) {
                // first pass (offsets == null) only counts; the second writes every hit at the query's offset
                const int pos = count + __popcll(b & ((1ull << lane) - 1ull));
                if (offsets) outKeys[(size_t)offsets[q] + pos] = ((unsigned)cx << 26) | ((unsigned)cy << 20) | (unsigned)idx;
            }
            count += __popcll(b);
        }
    }
    if (lane == 0 && !offsets) outCount[q] = count;
}

// two smallest keys of the wave (k0 < k1)
// key joins the two smallest keys seen (k0 <= k1), as selects: written as `if (key < k0) { k1 = k0; k0 = key; } else if
// (key < k1) k1 = key;` inside the window lambda the compiler selects between the ADDRESSES of k0 and k1 and keeps both in
// scratch memory - a load and a store per candidate
__device__ __forceinline__ void two_min_insert(unsigned long long &k0, unsigned long long &k1, unsigned long long key) {
    const unsigned long long larger = key < k0 ? k0 : key;
    k0 = key < k0 ? key : k0;
    k1 = larger < k1 ? larger : k1;
}
__device__ __forceinline__ void wave_two_min(unsigned long long &k0, unsigned long long &k1) {
    const unsigned long long m0 = wave_min_u64(k0);
    const unsigned long long cand = (k0 == m0) ? k1 : k0;
    const unsigned long long m1 = wave_min_u64(cand);
    k0 = m0;
    k1 = m1;
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- candidate cache of the claim iteration (FtClaims::cache) ----
// One LDS counter per wave hands out the positions while a window is scanned (the candidates turn up in divergent code).
struct CacheBuild {
    unsigned long long *slot;  // null: no cache
    int *counter;              // LDS, this wave's
    bool build;                // wave-uniform: this scan files its candidates
};
__device__ __forceinline__ void cache_begin(CacheBuild &B, int lane) {
    if (B.build) {
        if (lane == 0) *B.counter = 0;
        wave_lds_sync();
    }
}
__device__ __forceinline__ void cache_append(const CacheBuild &B, unsigned long long key) {
    const int pos = atomicAdd(B.counter, 1);
    if (pos < FT_CACHE_CAP) B.slot[1 + pos] = key;
}
// The best candidates first.  A later pass needs the smallest (two smallest) UNLOCKED keys of a list, and a key's order is
// its distance before anything else: with the keys of the FT_CACHE_HEAD (or a few more) smallest distances at the front of
// the list, a pass that finds enough unlocked keys among them need not look at the rest - at th 15 a window holds ~190
// candidates, and every candidate looked at is a 32-byte record read.  A kernel of its own does it once behind the first pass
// of a batch (k_cache_partition_batch; inside the search kernels it cost them 45 registers): the smallest distance D with at
// least FT_CACHE_HEAD keys <= D by bisection over the 9 bits of the distance (a count per step), then a stable partition of
// the list by dist <= D.  The head's length goes into the meta word; a list that is short, or whose head would not be short
// (many equal distances), keeps head = count - as every list of the single-frame path does.
#ifndef FT_CACHE_HEAD
#define FT_CACHE_HEAD 16
#endif
#ifndef FT_CACHE_HEAD_MAX
#define FT_CACHE_HEAD_MAX 48
#endif
// (PER = keys per lane: 64 PER >= n)
template <int PER>
__device__ __forceinline__ int cache_partition(unsigned long long *slot, int n, int lane) {
    unsigned long long k[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) k[j] = (lane + 64 * j < n) ? slot[1 + lane + 64 * j] : KEY_NONE;
    int lo = 0, hi = 256;  // smallest D in [0, 256] with count(dist <= D) >= FT_CACHE_HEAD (every distance is <= 256)
    while (lo < hi) {      // wave-uniform
        const int mid = (lo + hi) >> 1;
        int c = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) c += (k[j] != KEY_NONE && key_dist(k[j]) <= mid) ? 1 : 0;
        if (wave_sum_i32(c) >= FT_CACHE_HEAD) hi = mid;
        else lo = mid + 1;
    }
    int c = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) c += (k[j] != KEY_NONE && key_dist(k[j]) <= lo) ? 1 : 0;
    const int head = wave_sum_i32(c);
    if (head > FT_CACHE_HEAD_MAX || head >= n) return n;
    int front = 0, back = head;  // next free position of the two parts
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const bool have = k[j] != KEY_NONE, sel = have && key_dist(k[j]) <= lo;
        const unsigned long long bs = __ballot(sel), bo = __ballot(have && !sel);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (sel) slot[1 + front + __popcll(bs & below)] = k[j];
        else if (have) slot[1 + back + __popcll(bo & below)] = k[j];
        front += __popcll(bs);
        back += __popcll(bo);
    }
    return head;
}
__device__ __forceinline__ void cache_end(const CacheBuild &B, int lane, bool anyInBox) {
    if (B.build) {
        wave_lds_sync();
        const int n = *B.counter;
        if (lane == 0)  // (head = count: not partitioned)
            B.slot[0] = (unsigned long long)(unsigned)n | ((unsigned long long)(anyInBox ? 1 : 0) << 32) |
                        ((unsigned long long)(unsigned)min(n, FT_CACHE_CAP) << 40);
    }
}
// length of the list's head (cache_partition): <= the count
__device__ __forceinline__ int cache_head(unsigned long long meta) { return (int)((meta >> 40) & 0x3ffull); }
// 0 = not built yet, 1 = usable (count = candidates filed), 2 = built but too many candidates: scan the window again
// the same from a meta word that is already in a register (slot != null)
__device__ __forceinline__ int cache_state_of(unsigned long long meta, int &count, bool &anyInBox) {
    count = 0;
    anyInBox = false;
    if (meta == KEY_NONE) return 0;
    count = (int)(unsigned)meta;
    anyInBox = ((meta >> 32) & 1ull) != 0;
    return count <= FT_CACHE_CAP ? 1 : 2;
}
__device__ __forceinline__ int cache_state(const unsigned long long *slot, int &count, bool &anyInBox) {
    count = 0;
    anyInBox = false;
    if (!slot) return 2;
    const unsigned long long meta = slot[0];
    if (meta == KEY_NONE) return 0;
    count = (int)(unsigned)meta;
    anyInBox = ((meta >> 32) & 1ull) != 0;
    return count <= FT_CACHE_CAP ? 1 : 2;
}

// A keypoint of a window as the scans see it: position, octave, uright, descriptor - from the search records of the grid
// (one 16-byte and one 32-byte read at the entry's position) or, without a grid, from the frame's own arrays.
struct WinEntry {
    float x, y, uright;  // uright: < 0 = none (or a two-camera frame)
    int idx, octave, cx, cy;
    unsigned long long d[4];
};
// level band and box test of GetFeaturesInArea for a keypoint whose cell is already known to lie in the window
__device__ __forceinline__ bool in_box(const WinEntry &kp, float x, float y, float r, int minLevel, int maxLevel) {
    const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
    if (checkLevels) {
        if (kp.octave < minLevel) return false;
        if (maxLevel >= 0 && kp.octave > maxLevel) return false;
    }
    const float dx = __fsub_rn(kp.x, x), dy = __fsub_rn(kp.y, y);
    return fabsf(dx) < r && fabsf(dy) < r;
}

// The keypoints of camera `cam` whose grid cell lies in window w and whose octave lies in the level band of the search,
// handed to fn(entry) lane-parallel.  The frame's grid (k_build_grid) is a CSR PER OCTAVE: the keypoints of octave o in the
// cells (cx, minCY .. maxCY) are one contiguous range of entries - a map point looks at the keypoints GetFeaturesInArea would
// return for it (window AND level band: the band keeps 13 - 40 % of a window's keypoints, least where the windows are
// largest), where a grid over all octaves made the first pass of a search read every keypoint of the window.  Without a grid
// every keypoint's cell is computed and tested.
// minLevel / maxLevel as Frame::GetFeaturesInArea takes them (src/Frame.cc:714-729): no check at all unless minLevel > 0 or
// maxLevel >= 0; maxLevel < 0 = no upper bound.
template <class Fn>
__device__ __forceinline__ void for_window(const FtDevFrame &F, const FramePtrs &Q, int cam, const ft_keypoint *keys, int n, const Window &w,
                                           int minLevel, int maxLevel, int lane, Fn fn) {
    if (Q.gridStart[cam]) {
        // One lane per (octave, column of cells) range, a wave scan lays the ranges end to end, and the lanes take the
        // entries 64 at a time - two rounds per trip: record and descriptor of an entry sit at the entry's position, so a
        // round is one memory round trip, and a wide window a chain of them.
        const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
        const int lo = checkLevels ? min(max(minLevel, 0), F.nlevels - 1) : 0;  // (octaves beyond the table are filed under its last bucket)
        const int hi = (checkLevels && maxLevel >= 0) ? min(maxLevel, F.nlevels - 1) : F.nlevels - 1;
        const int ncolsW = w.maxCX - w.minCX + 1;
        const int npairs = (hi - lo + 1) * ncolsW;  // (<= 0: an empty band)
        const int *gs = Q.gridStart[cam];
        const float4 *rec = Q.gridRec[cam];
        const uint4 *gd = (const uint4 *)Q.gridDesc[cam];
        const unsigned colMagic = div_magic_u(ncolsW);
        for (int p0 = 0; p0 < npairs; p0 += 64) {
            const int np = min(64, npairs - p0);
            int b = 0, cnt = 0, myCol = 0;
            if (lane < np) {
                const int pidx = p0 + lane;
                const int oi = colMagic ? (int)__umulhi((unsigned)pidx, colMagic) : pidx;
                myCol = w.minCX + (pidx - oi * ncolsW);
                const int *col = gs + (size_t)(lo + oi) * (FT_GRID_CELLS + 1) + myCol * FT_GRID_ROWS;
                b = col[w.minCY];
                cnt = col[w.maxCY + 1] - b;
            }
            int incl = cnt;  // inclusive scan over the lanes
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int y = __shfl_up(incl, d);
                if (lane >= d) incl += y;
            }
            const int total = __builtin_amdgcn_readlane(incl, 63);
            auto locate = [&](int t, int &pos, int &cx) {
                int r = 0;  // the range entry t falls into: the number of ranges that end at or before t
                for (int c = 0; c < np - 1; c++) r += t >= __builtin_amdgcn_readlane(incl, c) ? 1 : 0;
                const int cb = __shfl(b, r), cEnd = __shfl(incl, r), cCnt = __shfl(cnt, r);
                cx = __shfl(myCol, r);
                pos = cb + (t - (cEnd - cCnt));
            };
            auto hand = [&](const float4 &r, const uint4 &d0, const uint4 &d1, int cx) {
                WinEntry e;
                e.x = r.x; e.y = r.y; e.uright = r.z;
                const int io = __float_as_int(r.w);
                e.idx = io & 0xffffff;
                e.octave = io >> 24;  // (signed: the keypoint's own octave, whatever bucket it was filed under)
                e.cx = cx;
                e.cy = (int)roundf(__fmul_rn(__fsub_rn(r.y, F.mnMinY), F.invH));  // Frame::PosInGrid, as k_build_grid filed it
                e.d[0] = (unsigned long long)d0.x | ((unsigned long long)d0.y << 32);
                e.d[1] = (unsigned long long)d0.z | ((unsigned long long)d0.w << 32);
                e.d[2] = (unsigned long long)d1.x | ((unsigned long long)d1.y << 32);
                e.d[3] = (unsigned long long)d1.z | ((unsigned long long)d1.w << 32);
                fn(e);
            };
            for (int t0 = 0; t0 < total; t0 += 128) {
                const int tA = t0 + lane, tB = t0 + 64 + lane;
                int posA, cxA, posB = 0, cxB = 0;
                locate(min(tA, total - 1), posA, cxA);
                const bool second = t0 + 64 < total;  // wave-uniform
                if (second) locate(min(tB, total - 1), posB, cxB);
                const float4 rA = rec[posA];
                const uint4 a0 = gd[2 * (size_t)posA], a1 = gd[2 * (size_t)posA + 1];
                float4 rB = rA;
                uint4 b0 = a0, b1 = a1;
                if (second) {
                    rB = rec[posB];
                    b0 = gd[2 * (size_t)posB];
                    b1 = gd[2 * (size_t)posB + 1];
                }
                if (tA < total) hand(rA, a0, a1, cxA);
                if (second && tB < total) hand(rB, b0, b1, cxB);
            }
        }
        return;
    }
    const uint8_t *desc = Q.desc + (cam == 0 ? 0 : (size_t)F.Nleft * 32);
    for (int idx = lane; idx < n; idx += 64) {
        const ft_keypoint kp = keys[idx];
        const int cx = (int)roundf(__fmul_rn(__fsub_rn(kp.x, F.mnMinX), F.invW));
        const int cy = (int)roundf(__fmul_rn(__fsub_rn(kp.y, F.mnMinY), F.invH));
        if (cx < 0 || cx >= FT_GRID_COLS || cy < 0 || cy >= FT_GRID_ROWS) continue;  // never entered the grid
        if (cx < w.minCX || cx > w.maxCX || cy < w.minCY || cy > w.maxCY) continue;
        WinEntry e;
        e.x = kp.x; e.y = kp.y;
        e.uright = (cam == 0 && F.Nleft == -1 && Q.uright) ? Q.uright[idx] : -1.0f;
        e.idx = idx; e.octave = kp.octave; e.cx = cx; e.cy = cy;
        const unsigned long long *dp = (const unsigned long long *)(desc + (size_t)idx * 32);
        e.d[0] = dp[0]; e.d[1] = dp[1]; e.d[2] = dp[2]; e.d[3] = dp[3];
        fn(e);
    }
}

// Frame::AssignFeaturesToGrid (src/Frame.cc:409-440) as one CSR per octave: workgroup (octave o, camera) counting-sorts the
// camera's keypoints of octave o by cell cx * 48 + cy in LDS and files them behind the keypoints of the lower octaves (their
// number is counted on the way).  An octave outside [0, nlevels) is filed under the nearest bucket; the searches test the
// keypoint's own octave anyway.  The order inside a cell is free (the searches order candidates by (distance, cx, cy, index)
// keys).  start: [nlevels][FT_GRID_CELLS + 1] absolute entry positions; rec / desc: the entries (ft_search.h).
__device__ __forceinline__ void build_grid_body(const FtDevFrame &F, const FramePtrs &Q, int oct, int cam, int *startL, int *startR,
                                                float4 *recL, uint8_t *descL, float4 *recR, uint8_t *descR) {
    __shared__ int cnt[FT_GRID_CELLS + 1];
    __shared__ int wsum[4], wbelow[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = cam == 0 ? (F.Nleft == -1 ? F.N : F.Nleft) : (F.Nleft == -1 ? 0 : F.N - F.Nleft);
    const ft_keypoint *keys = cam == 0 ? Q.keys : Q.keysR;
    int *start = (cam == 0 ? startL : startR);
    if (!start) return;
    start += (size_t)oct * (FT_GRID_CELLS + 1);
    for (int c = tid; c <= FT_GRID_CELLS; c += 256) cnt[c] = 0;
    __syncthreads();
    auto cellOf = [&](const ft_keypoint &kp) -> int {
        const int cx = (int)roundf(__fmul_rn(__fsub_rn(kp.x, F.mnMinX), F.invW));
        const int cy = (int)roundf(__fmul_rn(__fsub_rn(kp.y, F.mnMinY), F.invH));
        if (cx < 0 || cx >= FT_GRID_COLS || cy < 0 || cy >= FT_GRID_ROWS) return -1;
        return cx * FT_GRID_ROWS + cy;
    };
    int below = 0;  // keypoints of the grid in lower buckets
    for (int i = tid; i < n; i += 256) {
        const ft_keypoint kp = keys[i];
        const int c = cellOf(kp);
        if (c < 0) continue;
        const int bkt = min(max(kp.octave, 0), F.nlevels - 1);
        if (bkt < oct) below++;
        else if (bkt == oct) atomicAdd(&cnt[c], 1);
    }
    below = wave_sum_i32(below);
    if (lane == 0) wbelow[wave] = below;
    __syncthreads();
    // exclusive scan of the 3072 counts: 12 consecutive cells per thread
    constexpr int PER = FT_GRID_CELLS / 256;
    int local = 0;
    for (int k = 0; k < PER; k++) local += cnt[tid * PER + k];
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    const int base = wbelow[0] + wbelow[1] + wbelow[2] + wbelow[3];
    int run = base + incl - local;
    for (int w = 0; w < wave; w++) run += wsum[w];
    const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    for (int k = 0; k < PER; k++) {
        const int c = tid * PER + k, v = cnt[c];
        start[c] = run;
        cnt[c] = run;  // becomes the fill cursor of the cell
        run += v;
    }
    if (tid == 0) start[FT_GRID_CELLS] = base + total;
    __syncthreads();
    float4 *rec = cam == 0 ? recL : recR;
    uint8_t *gdesc = cam == 0 ? descL : descR;
    const uint8_t *desc = Q.desc + (cam == 0 ? 0 : (size_t)F.Nleft * 32);
    for (int i = tid; i < n; i += 256) {
        const ft_keypoint kp = keys[i];
        const int c = cellOf(kp);
        if (c < 0 || min(max(kp.octave, 0), F.nlevels - 1) != oct) continue;
        const int p = atomicAdd(&cnt[c], 1);
        const float ur = (cam == 0 && F.Nleft == -1 && Q.uright) ? Q.uright[i] : -1.0f;
        rec[p] = make_float4(kp.x, kp.y, ur, __int_as_float((i & 0xffffff) | (kp.octave << 24)));
        const uint4 *d = (const uint4 *)(desc + (size_t)i * 32);
        uint4 *o = (uint4 *)(gdesc + (size_t)p * 32);
        o[0] = d[0];
        o[1] = d[1];
    }
}
__global__ __launch_bounds__(256) void k_build_grid(FtDevFrame F, int *startL, int *startR, float4 *recL, uint8_t *descL, float4 *recR,
                                                    uint8_t *descR) {
    build_grid_body(F, frame_ptrs(F, FT_NO_REBASE), blockIdx.x, blockIdx.y, startL, startR, recL, descL, recR, descR);
}
// the grids of the frames of a batch (ft_tracked_batch): blockIdx.z = frame; the arrays are those F.gridStart / gridRec /
// gridDesc of the frame's job already point to
__global__ __launch_bounds__(256) void k_build_grid_batch(const FtBatchJob *__restrict__ jobs, Rebase rb) {
    const FtDevFrame &F = jobs[blockIdx.z].F;
    if ((int)blockIdx.x >= F.nlevels || (blockIdx.y == 1 && F.Nleft == -1)) return;
    const FramePtrs Q = frame_ptrs(F, rb);
    build_grid_body(F, Q, blockIdx.x, blockIdx.y, (int *)Q.gridStart[0], (int *)Q.gridStart[1], (float4 *)Q.gridRec[0],
                    (uint8_t *)Q.gridDesc[0], (float4 *)Q.gridRec[1], (uint8_t *)Q.gridDesc[1]);
}

// ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th, ...) for map point i by one wave (src/ORBmatcher.cc:49-225):
// r = (primary left, side left, primary right, side right) keypoints it writes; raw outputs as the reference kernel's
__device__ __forceinline__ void local_point(const FtDevFrame &F, const FramePtrs &Q, const FtDevLocalPoints &P, const FtClaims &C, float th,
                                            float nnRatio, int i, int lane, int r4[4], const FtLocalRaw &raw, int *ldsCounter) {
    int primL = -1, sideL = -1, primR = -1, sideR = -1;
    int bd = 256, bd2 = 256, bl = -1, bl2 = -1, bi = -1;
    int bdr = 256, bd2r = 256, blr = -1, bl2r = -1, bir = -1;
    bool skipRight = false;
    unsigned long long q[4];
    {
        const unsigned long long *p = (const unsigned long long *)(P.desc + (size_t)i * 32);
        q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; q[3] = p[3];
    }
    // Everything a pass needs of the point that does not depend on another load is requested HERE, before the first use: the
    // flags, both cache meta words and - speculatively - the first 64 cached keys of either camera.  A pass on cached
    // candidates was a chain of five dependent round trips (skip -> in view -> meta -> keys -> lock records); now it is two.
    unsigned long long *slotL = C.cache ? C.cache + (size_t)i * FT_CACHE_WORDS : nullptr;
    unsigned long long *slotR = slotL ? slotL + (FT_CACHE_CAP + 1) : nullptr;
    const bool twoCam = F.Nleft != -1;
    const uint8_t skipV = P.skip[i], inViewV = P.inView[i], inViewRV = twoCam ? P.inViewR[i] : (uint8_t)0;
    const int levelRV = twoCam ? P.levelR[i] : -1;
    const unsigned long long metaL = slotL ? slotL[0] : KEY_NONE, metaR = (slotR && twoCam) ? slotR[0] : KEY_NONE;
    const unsigned long long keyL0 = slotL ? slotL[1 + lane] : KEY_NONE, keyR0 = (slotR && twoCam) ? slotR[1 + lane] : KEY_NONE;
    if (!skipV) {
        const int nLeft = F.Nleft == -1 ? F.N : F.Nleft;
        if (inViewV) {
            unsigned long long k0 = KEY_NONE, k1 = KEY_NONE;
            CacheBuild cb;
            cb.slot = slotL;
            cb.counter = ldsCounter;
            int nCached;
            bool anyBox;
            const int cs = slotL ? cache_state_of(metaL, nCached, anyBox) : 2;
            cb.build = cs == 0;
            if (cs == 1) {
                for (int t = lane; t < nCached; t += 64) {
                    const unsigned long long key = t < 64 ? keyL0 : cb.slot[1 + t];
                    if (is_locked(C, key_idx(key), i, key_held(key))) continue;
                    two_min_insert(k0, k1, key);
                }
            } else {
                const int level = P.level[i];
                float r = ((double)P.viewCos[i] > 0.998) ? 2.5f : 4.0f;  // RadiusByViewingCos, ORBmatcher.cc:314-320
                if ((double)th != 1.0) r = __fmul_rn(r, th);
                const float rad = __fmul_rn(r, F.sf[level]);
                const float x = P.projX[i], y = P.projY[i];
                const Window w = cell_window(F, x, y, rad);
                cache_begin(cb, lane);
                if (!w.empty) {
                    const float pxr = (F.Nleft == -1 && Q.uright) ? P.projXR[i] : 0.f;
                    for_window(F, Q, 0, Q.keys, nLeft, w, level - 1, level, lane, [&](const WinEntry &kp) {
                        if (!in_box(kp, x, y, rad, level - 1, level)) return;
                        const int idx = kp.idx;
                        const bool held = Q.holderObs[idx] > 0;
                        const bool locked = is_locked(C, idx, i, held);
                        if (locked && !cb.build) return;
                        if (kp.uright > 0) {  // (mono-stereo frames only: the records of other frames hold -1)
                            const float er = fabsf(__fsub_rn(pxr, kp.uright));
                            if (er > rad) return;
                        }
                        const int dist = hamming256(q, kp.d);
                        const unsigned long long key = make_key(dist, kp.cx, kp.cy, idx, kp.octave, held);
                        if (cb.build) cache_append(cb, key);
                        if (locked) return;
                        two_min_insert(k0, k1, key);
                    });
                }
                cache_end(cb, lane, false);
            }
            wave_two_min(k0, k1);
            if (k0 != KEY_NONE) {
                bd = key_dist(k0);
                bi = key_idx(k0);
                bl = key_octave(k0);
            }
            if (k1 != KEY_NONE) {
                bd2 = key_dist(k1);
                bl2 = key_octave(k1);
            }
            if (bd <= FT_TH_HIGH) {
                if (bl == bl2 && (float)bd > __fmul_rn(nnRatio, (float)bd2)) {
                    skipRight = true;  // the reference's `continue` also skips the right-camera block
                } else {
                    primL = bi;
                    if (F.Nleft != -1 && Q.l2r[bi] != -1) sideL = Q.l2r[bi] + F.Nleft;
                }
            }
        }
        if (twoCam && inViewRV && !skipRight) {
            const int level = levelRV;
            if (level != -1) {
                const int nRight = F.N - F.Nleft;
                unsigned long long k0 = KEY_NONE, k1 = KEY_NONE;
                // this point's own left-block side write precedes its right-block search
                auto lockedR = [&](int g, bool held) -> bool { return (g == sideL) ? (C.obs[i] > 0) : is_locked(C, g, i, held); };
                // (the right block is not reached in every pass - skipRight depends on the locks - so its candidates are filed by
                // the first pass that gets here)
                CacheBuild cb;
                cb.slot = slotR;
                cb.counter = ldsCounter;
                int nCached;
                bool anyBox;
                const int cs = slotR ? cache_state_of(metaR, nCached, anyBox) : 2;
                cb.build = cs == 0;
                if (cs == 1) {
                    for (int t = lane; t < nCached; t += 64) {
                        const unsigned long long key = t < 64 ? keyR0 : cb.slot[1 + t];
                        if (lockedR(key_idx(key) + F.Nleft, key_held(key))) continue;
                        two_min_insert(k0, k1, key);
                    }
                } else {
                    const float r = ((double)P.viewCosR[i] > 0.998) ? 2.5f : 4.0f;
                    const float rad = __fmul_rn(r, F.sf[level]);
                    const float x = P.projXR[i], y = P.projYR[i];
                    const Window w = cell_window(F, x, y, rad);
                    cache_begin(cb, lane);
                    if (!w.empty) {
                        for_window(F, Q, 1, Q.keysR, nRight, w, level - 1, level, lane, [&](const WinEntry &kp) {
                            if (!in_box(kp, x, y, rad, level - 1, level)) return;
                            const int idx = kp.idx, g = idx + F.Nleft;
                            const bool held = Q.holderObs[g] > 0;
                            const bool locked = lockedR(g, held);
                            if (locked && !cb.build) return;
                            const int dist = hamming256(q, kp.d);
                            const unsigned long long key = make_key(dist, kp.cx, kp.cy, idx, kp.octave, held);
                            if (cb.build) cache_append(cb, key);
                            if (locked) return;
                            two_min_insert(k0, k1, key);
                        });
                    }
                    cache_end(cb, lane, false);
                }
                wave_two_min(k0, k1);
                if (k0 != KEY_NONE) {
                    bdr = key_dist(k0);
                    bir = key_idx(k0);
                    blr = key_octave(k0);
                }
                if (k1 != KEY_NONE) {
                    bd2r = key_dist(k1);
                    bl2r = key_octave(k1);
                }
                if (bdr <= FT_TH_HIGH && !(blr == bl2r && (float)bdr > __fmul_rn(nnRatio, (float)bd2r))) {
                    if (Q.r2l[bir] != -1) sideR = Q.r2l[bir];
                    primR = bir + F.Nleft;
                }
            }
        }
    }
    r4[0] = primL; r4[1] = sideL; r4[2] = primR; r4[3] = sideR;
    if (lane == 0 && raw.bestDist) {  // the reference kernel's raw outputs are optional (the resident-frame path skips them)
        raw.bestDist[i] = bd; raw.bestDist2[i] = bd2; raw.bestLevel[i] = bl; raw.bestLevel2[i] = bl2; raw.bestIdx[i] = bi;
        raw.bestDistR[i] = bdr; raw.bestDist2R[i] = bd2r; raw.bestLevelR[i] = blr; raw.bestLevel2R[i] = bl2r; raw.bestIdxR[i] = bir;
    }
}

// waves (= points) per workgroup of the search kernels: a pass is short, and its fixed cost is the dispatch of its workgroups
#ifndef FT_SEARCH_WPB
#define FT_SEARCH_WPB 4
#endif
__global__ __launch_bounds__(64 * FT_SEARCH_WPB) void k_search_local(FtDevFrame F, FtDevLocalPoints P, FtClaims C, float th,
                                                      float nnRatio, int *res, FtLocalRaw raw) {
    if (!claims_begin_pass(C)) return;
    const int lane = threadIdx.x & 63, wave = wave_index();
    const int i = blockIdx.x * FT_SEARCH_WPB + wave;
    if (i >= P.M) return;
    __shared__ int cacheCounter[FT_SEARCH_WPB];
    int r4[4];
    local_point(F, frame_ptrs(F, FT_NO_REBASE), P, C, th, nnRatio, i, lane, r4, raw, &cacheCounter[wave]);
    claims_file(C, res, i, lane, r4);
}

// camera models: src/CameraModels/Pinhole.cpp:43-49, KannalaBrandt8.cpp:67-84
__device__ __forceinline__ void project_cam(const FtDevFrame &F, const float p[3], float uv[2]) {
    if (F.camModel == 0) {
        uv[0] = __fadd_rn(__fdiv_rn(__fmul_rn(F.cam[0], p[0]), p[2]), F.cam[2]);
        uv[1] = __fadd_rn(__fdiv_rn(__fmul_rn(F.cam[1], p[1]), p[2]), F.cam[3]);
    } else {
        const float x2y2 = __fadd_rn(__fmul_rn(p[0], p[0]), __fmul_rn(p[1], p[1]));
        const float theta = ft_atan2_f(sqrtf(x2y2), p[2]);
        const float psi = ft_atan2_f(p[1], p[0]);
        const float t2 = __fmul_rn(theta, theta);
        const float t3 = __fmul_rn(theta, t2);
        const float t5 = __fmul_rn(t3, t2);
        const float t7 = __fmul_rn(t5, t2);
        const float t9 = __fmul_rn(t7, t2);
        const float r = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(theta, __fmul_rn(F.cam[4], t3)), __fmul_rn(F.cam[5], t5)),
                                            __fmul_rn(F.cam[6], t7)),
                                  __fmul_rn(F.cam[7], t9));
        uv[0] = __fadd_rn(__fmul_rn(__fmul_rn(F.cam[0], r), ft_cos_f(psi)), F.cam[2]);
        uv[1] = __fadd_rn(__fmul_rn(__fmul_rn(F.cam[1], r), ft_sin_f(psi)), F.cam[3]);
    }
}

__device__ __forceinline__ void transform34(const float *T, const float x[3], float y[3]) {
#pragma unroll
    for (int r = 0; r < 3; r++)
        y[r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[4 * r], x[0]), __fmul_rn(T[4 * r + 1], x[1])),
                                   __fmul_rn(T[4 * r + 2], x[2])),
                         T[4 * r + 3]);
}

// Sophus::SE3f * point as the reference's CPU branch evaluates `Tcw * x3Dw` (src/ORBmatcher.cc:1805) and `GetRelativePoseTrl() *
// x3Dc` (:1900): Thirdparty/Sophus/sophus/so3.hpp:358-367 - uv = q.vec().cross(p); uv += uv; p + q.w() * uv + q.vec().cross(uv) -
// then + translation (se3.hpp:321-324); every product and sum rounded on its own, coefficient order as Eigen's cross()
__device__ __forceinline__ void cross3_rn(const float a[3], const float b[3], float c[3]) {
    c[0] = __fsub_rn(__fmul_rn(a[1], b[2]), __fmul_rn(a[2], b[1]));
    c[1] = __fsub_rn(__fmul_rn(a[2], b[0]), __fmul_rn(a[0], b[2]));
    c[2] = __fsub_rn(__fmul_rn(a[0], b[1]), __fmul_rn(a[1], b[0]));
}
__device__ __forceinline__ void transform_pose(const float *m, const float *q, int quat, const float x[3], float y[3]) {
    if (!quat) {
        transform34(m, x, y);
        return;
    }
    float uv[3], c[3];
    cross3_rn(q, x, uv);
#pragma unroll
    for (int i = 0; i < 3; i++) uv[i] = __fadd_rn(uv[i], uv[i]);
    cross3_rn(q, uv, c);
#pragma unroll
    for (int i = 0; i < 3; i++) y[i] = __fadd_rn(__fadd_rn(__fadd_rn(x[i], __fmul_rn(q[3], uv[i])), c[i]), m[4 * i + 3]);
}

// ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) for last-frame point i by one wave (src/ORBmatcher.cc:
// 1775-1960): r = (left keypoint written, -1, right keypoint written, -1)
// pre (may be null): the projections of the points, computed once per point by k_last_project_batch - the pose transform and the
// camera model (two atan2f, a cosf and a sinf per camera for KannalaBrandt8) are the same number in all 64 lanes of a point's wave
template <bool PRE = false>
__device__ __forceinline__ void last_point(const FtDevFrame &F, const FramePtrs &Q, const FtDevLastPoints &Lp, const FtClaims &C, const FtPose &Tcw,
                                           float th, int bForward, int bBackward, int i, int lane, int r4[4], const FtLastRaw &raw,
                                           int *ldsCounter, const FtLastProj *pre = nullptr) {
    int primL = -1, primR = -1;
    int bd = 256, bi = -1, bdr = 256, bir = -1;
    // Later passes of the claim iteration on a point whose candidates are cached need neither the pose transform nor the camera
    // model (two atan2f, a cosf and a sinf for KannalaBrandt8 - evaluated by all 64 lanes, on the critical path of a pass that
    // is otherwise a handful of dependent loads): the cached keys already hold everything that depends on the projection.
    unsigned long long *slotL = C.cache ? C.cache + (size_t)i * FT_CACHE_WORDS : nullptr;
    unsigned long long *slotR = slotL ? slotL + (FT_CACHE_CAP + 1) : nullptr;
    // requested together, before the first use: validity, both meta words and (speculatively) the first 64 keys of either camera
    const bool twoCam = F.Nleft != -1;
    const uint8_t validV = Lp.valid[i];
    const unsigned long long metaL = slotL ? slotL[0] : KEY_NONE, metaR = (slotR && twoCam) ? slotR[0] : KEY_NONE;
    const unsigned long long keyL0 = slotL ? slotL[1 + lane] : KEY_NONE, keyR0 = (slotR && twoCam) ? slotR[1 + lane] : KEY_NONE;
    bool fromCache = false;
    int nCachedL = 0, nCachedR = 0;
    bool anyBoxL = false, anyBoxR = false;
    if (validV && slotL && cache_state_of(metaL, nCachedL, anyBoxL) == 1)
        fromCache = !twoCam || !anyBoxL || cache_state_of(metaR, nCachedR, anyBoxR) == 1;
    if (fromCache) {
        unsigned long long k0 = KEY_NONE;
        for (int t = lane; t < nCachedL; t += 64) {
            const unsigned long long key = t < 64 ? keyL0 : slotL[1 + t];
            if (is_locked(C, key_idx(key), i, key_held(key))) continue;
            k0 = key < k0 ? key : k0;
        }
        k0 = wave_min_u64(k0);
        if (anyBoxL) {
            if (k0 != KEY_NONE) {
                bd = key_dist(k0);
                bi = key_idx(k0);
            }
            if (bd <= FT_TH_HIGH) primL = bi;
            if (F.Nleft != -1) {
                unsigned long long kr = KEY_NONE;
                for (int t = lane; t < nCachedR; t += 64) {
                    const unsigned long long key = t < 64 ? keyR0 : slotR[1 + t];
                    if (is_locked(C, key_idx(key) + F.Nleft, i, key_held(key))) continue;
                    kr = key < kr ? key : kr;
                }
                kr = wave_min_u64(kr);
                if (kr != KEY_NONE) {
                    bdr = key_dist(kr);
                    bir = key_idx(kr);
                }
                if (bdr <= FT_TH_HIGH) primR = bir + F.Nleft;
            }
        }
    } else if (validV) {
        float xc[3] = {0.f, 0.f, 0.f}, uv[2] = {0.f, 0.f}, uvrPre[2] = {0.f, 0.f};
        float invzc;
        bool go;
        if constexpr (PRE) {
            const FtLastProj pj = pre[i];
            uv[0] = pj.u; uv[1] = pj.v; invzc = pj.invzc; uvrPre[0] = pj.ur; uvrPre[1] = pj.vr;
            go = pj.go != 0;
        } else {
            float xw[3] = {Lp.worldPos[3 * i], Lp.worldPos[3 * i + 1], Lp.worldPos[3 * i + 2]};
            transform_pose(Tcw.m, Tcw.q, Tcw.quat, xw, xc);
            invzc = (float)(1.0 / (double)xc[2]);
            go = !(invzc < 0);
            if (go) {
                project_cam(F, xc, uv);
                if (uv[0] < F.mnMinX || uv[0] > F.mnMaxX) go = false;
                if (uv[1] < F.mnMinY || uv[1] > F.mnMaxY) go = false;
            }
        }
        // a point that does not project into the image: an empty cache entry spares the later passes the projection
        if (!go && slotL && lane == 0 && metaL == KEY_NONE) slotL[0] = 0ull;
        if (go) {
            const int oct = Lp.octave[i];
            const float radius = __fmul_rn(th, F.sf[oct]);
            int minLevel, maxLevel;
            if (bForward) { minLevel = oct; maxLevel = -1; }
            else if (bBackward) { minLevel = 0; maxLevel = oct; }
            else { minLevel = oct - 1; maxLevel = oct + 1; }
            unsigned long long q[4];
            {
                const unsigned long long *p = (const unsigned long long *)(Lp.desc + (size_t)i * 32);
                q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; q[3] = p[3];
            }
            const int nLeft = F.Nleft == -1 ? F.N : F.Nleft;
            const Window w = cell_window(F, uv[0], uv[1], radius);
            unsigned long long k0 = KEY_NONE;
            int anyCand = 0;
            CacheBuild cb;
            cb.slot = C.cache ? C.cache + (size_t)i * FT_CACHE_WORDS : nullptr;
            cb.counter = ldsCounter;
            int nCached;
            bool anyBox;
            const int cs = cache_state(cb.slot, nCached, anyBox);
            cb.build = cs == 0;
            if (cs == 1) {
                anyCand = anyBox ? 1 : 0;
                for (int t = lane; t < nCached; t += 64) {
                    const unsigned long long key = cb.slot[1 + t];
                    if (is_locked(C, key_idx(key), i, key_held(key))) continue;
                    k0 = key < k0 ? key : k0;
                }
            } else {
                cache_begin(cb, lane);
                if (!w.empty) {
                    for_window(F, Q, 0, Q.keys, nLeft, w, minLevel, maxLevel, lane, [&](const WinEntry &kp) {
                        if (!in_box(kp, uv[0], uv[1], radius, minLevel, maxLevel)) return;
                        anyCand = 1;
                        const int idx = kp.idx;
                        const bool held = Q.holderObs[idx] > 0;
                        const bool locked = is_locked(C, idx, i, held);
                        if (locked && !cb.build) return;
                        if (kp.uright > 0) {
                            const float ur = __fsub_rn(uv[0], __fmul_rn(F.mbf, invzc));
                            const float er = fabsf(__fsub_rn(ur, kp.uright));
                            if (er > radius) return;
                        }
                        const int dist = hamming256(q, kp.d);
                        const unsigned long long key = make_key(dist, kp.cx, kp.cy, idx, kp.octave, held);
                        if (cb.build) cache_append(cb, key);
                        if (locked) return;
                        k0 = key < k0 ? key : k0;
                    });
                }
                anyCand = __any(anyCand);
                cache_end(cb, lane, anyCand != 0);
            }
            k0 = wave_min_u64(k0);
            // `if(vIndices2.empty()) continue;` (ORBmatcher.cc:1836) also skips the right-camera block
            if (anyCand) {
                if (k0 != KEY_NONE) {
                    bd = key_dist(k0);
                    bi = key_idx(k0);
                }
                if (bd <= FT_TH_HIGH) primL = bi;
                if (F.Nleft != -1) {
                    float xr[3], uvr[2];
                    if constexpr (PRE) {
                        uvr[0] = uvrPre[0];
                        uvr[1] = uvrPre[1];
                    } else {
                        transform_pose(F.Trl, F.TrlQ, F.trlQuat, xc, xr);
                        project_cam(F, xr, uvr);
                    }
                    const Window wr = cell_window(F, uvr[0], uvr[1], radius);
                    const int nRight = F.N - F.Nleft;
                    unsigned long long kr = KEY_NONE;
                    CacheBuild cbr;
                    cbr.slot = C.cache ? C.cache + (size_t)i * FT_CACHE_WORDS + (FT_CACHE_CAP + 1) : nullptr;
                    cbr.counter = ldsCounter;
                    int nCachedR;
                    bool anyBoxR;
                    const int csr = cache_state(cbr.slot, nCachedR, anyBoxR);
                    cbr.build = csr == 0;
                    if (csr == 1) {
                        for (int t = lane; t < nCachedR; t += 64) {
                            const unsigned long long key = cbr.slot[1 + t];
                            if (is_locked(C, key_idx(key) + F.Nleft, i, key_held(key))) continue;
                            kr = key < kr ? key : kr;
                        }
                    } else {
                        cache_begin(cbr, lane);
                        if (!wr.empty) {
                            for_window(F, Q, 1, Q.keysR, nRight, wr, minLevel, maxLevel, lane, [&](const WinEntry &kp) {
                                if (!in_box(kp, uvr[0], uvr[1], radius, minLevel, maxLevel)) return;
                                const int idx = kp.idx;
                                const bool held = Q.holderObs[idx + F.Nleft] > 0;
                                const bool locked = is_locked(C, idx + F.Nleft, i, held);
                                if (locked && !cbr.build) return;
                                const int dist = hamming256(q, kp.d);
                                const unsigned long long key = make_key(dist, kp.cx, kp.cy, idx, kp.octave, held);
                                if (cbr.build) cache_append(cbr, key);
                                if (locked) return;
                                kr = key < kr ? key : kr;
                            });
                        }
                        cache_end(cbr, lane, false);
                    }
                    kr = wave_min_u64(kr);
                    if (kr != KEY_NONE) {
                        bdr = key_dist(kr);
                        bir = key_idx(kr);
                    }
                    if (bdr <= FT_TH_HIGH) primR = bir + F.Nleft;
                }
            }
        }
    }
    r4[0] = primL; r4[1] = -1; r4[2] = primR; r4[3] = -1;
    if (lane == 0 && raw.bestDist) {
        raw.bestDist[i] = bd; raw.bestIdx[i] = bi; raw.bestDistR[i] = bdr; raw.bestIdxR[i] = bir;
    }
}

__global__ __launch_bounds__(64 * FT_SEARCH_WPB) void k_search_last(FtDevFrame F, FtDevLastPoints Lp, FtClaims C, FtPose Tcw, float th,
                                                     int bForward, int bBackward, int *res, FtLastRaw raw) {
    if (!claims_begin_pass(C)) return;
    const int lane = threadIdx.x & 63, wave = wave_index();
    const int i = blockIdx.x * FT_SEARCH_WPB + wave;
    if (i >= Lp.N) return;
    __shared__ int cacheCounter[FT_SEARCH_WPB];
    int r4[4];
    last_point(F, frame_ptrs(F, FT_NO_REBASE), Lp, C, Tcw, th, bForward, bBackward, i, lane, r4, raw, &cacheCounter[wave]);
    claims_file(C, res, i, lane, r4);
}

// the projections of SearchByProjection(CurrentFrame, LastFrame) once per point (thread per point, blockIdx.y = frame): exactly the
// expressions of last_point - Tcw * x3Dw, 1 / z, mpCamera->project, the bounds test, and for two-camera frames Trl * x3Dc and
// mpCamera2->project (src/ORBmatcher.cc:1805-1822, 1900-1902)
__global__ __launch_bounds__(256) void k_last_project_batch(const FtBatchJob *__restrict__ jobs, Rebase rb) {
    const FtBatchJob &J = jobs[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= J.L.N || J.nPoints <= 0) return;
    const FtDevFrame &F = J.F;
    FtLastProj pj;
    pj.u = pj.v = pj.invzc = pj.ur = pj.vr = 0.f;
    pj.go = 0;
    uint8_t *valid = const_cast<uint8_t *>(rb(J.L.valid));
    // a valid point's octave indexes the frame's scale factors in every kernel behind this one: a value outside the levels
    // (the host checks it when it copies the points; arrays read in place from pinned memory are checked here) drops the point
    // and marks the frame - the call's second half reports FT_ERR_INVALID
    if (valid[i] && ((unsigned)rb(J.L.octave)[i] >= (unsigned)F.nlevels)) {
        valid[i] = 0;
        int *err = rb(J.err);
        if (err) atomicOr(err, FT_JOB_ERR_OCTAVE);
    }
    if (valid[i]) {
        const float *wp = rb(J.L.worldPos);
        const float xw[3] = {wp[3 * i], wp[3 * i + 1], wp[3 * i + 2]};
        float xc[3], uv[2] = {0.f, 0.f};
        transform_pose(J.Tcw.m, J.Tcw.q, J.Tcw.quat, xw, xc);
        const float invzc = (float)(1.0 / (double)xc[2]);
        bool go = !(invzc < 0);
        if (go) {
            project_cam(F, xc, uv);
            if (uv[0] < F.mnMinX || uv[0] > F.mnMaxX) go = false;
            if (uv[1] < F.mnMinY || uv[1] > F.mnMaxY) go = false;
        }
        pj.u = uv[0]; pj.v = uv[1]; pj.invzc = invzc;
        pj.go = go ? 1 : 0;
        if (go && F.Nleft != -1) {
            float xr[3], uvr[2];
            transform_pose(F.Trl, F.TrlQ, F.trlQuat, xc, xr);
            project_cam(F, xr, uvr);
            pj.ur = uvr[0]; pj.vr = uvr[1];
        }
    }
    rb(J.proj)[i] = pj;
}

// ---- B frames per launch (ft_tracked_batch) --------------------------------------------------------------------------------
// One frame at a time leaves the chip idle by construction: a pass of the claim iteration is ~500 workgroups and a handful
// of dependent L2 round trips, 9 - 13 passes per search, each a launch.  Here blockIdx.y is the FRAME: everything a pass
// needs of a frame - the frame itself, its points, its rotating claim buffers - sits in a job record in HBM (read through
// scalar loads: the address is uniform), the pass number selects the buffers exactly as fixedPoint (search.cpp) does for a
// launch of its own, and every frame has its own convergence flags, so that the workgroups of a frame whose iteration has
// reached its fixed point return at once while the other frames go on: the batch runs max-over-frames passes.
__device__ __forceinline__ FtClaims job_claims(const FtBatchJob &J, const Rebase &rb, int pass, int fCur, int fPrev, int fReset, int *&res) {
    const size_t K = (size_t)J.K, R = (size_t)4 * J.nPoints;
    int *head = rb(J.head), *tab = rb(J.tab), *next = rb(J.next), *resB = rb(J.res), *flags = rb(J.flags);
    FtClaims C;
    C.firstPass = pass == 0;
    C.head = head + (size_t)(pass % 3) * K;
    C.headWrite = head + (size_t)((pass + 1) % 3) * K;
    C.headClear = head + (size_t)((pass + 2) % 3) * K;
    C.tab = tab + (size_t)(pass % 3) * 8 * K;
    C.tabWrite = tab + (size_t)((pass + 1) % 3) * 8 * K;
    C.tabClear = tab + (size_t)((pass + 2) % 3) * 8 * K;
    C.next = next + (size_t)((pass + 1) & 1) * R;
    C.nextWrite = next + (size_t)(pass & 1) * R;
    C.resPrev = resB + (size_t)((pass + 1) & 1) * R;
    C.obs = rb(J.obs);
    C.nKp = J.nKp;
    C.flagCur = flags + fCur;
    C.flagPrev = fPrev >= 0 ? flags + fPrev : nullptr;
    C.flagReset = flags + fReset;
    C.flagStick = (fPrev >= 0 && (fPrev / (FT_BATCH_FLAGS / 2)) != (fCur / (FT_BATCH_FLAGS / 2))) ? flags + (fCur & ~(FT_BATCH_FLAGS / 2 - 1)) : nullptr;
    C.cache = rb(J.cache);
    res = resB + (size_t)(pass & 1) * R;
    return C;
}

// slowList != 0: a later pass - the points the lean kernel (k_search_*_lean, below) could not serve from the candidate cache, by
// a grid-stride loop over the frame's slow list of this pass's parity; the pass's clears were done by the lean kernel
#ifndef FT_BATCH_WAVES
#define FT_BATCH_WAVES 6  // waves per SIMD the first-pass kernels are compiled for (80 registers, 12 - 32 bytes of scratch: 0.60 -> 0.545 ms; 8: spills, 0.82 ms)
#endif
template <bool slowList>
__global__ __launch_bounds__(64 * FT_SEARCH_WPB) __attribute__((amdgpu_waves_per_eu(FT_BATCH_WAVES, 8))) void k_search_last_batch(const FtBatchJob *__restrict__ jobs, Rebase rb, int pass, int fCur,
                                                                          int fPrev, int fReset, float th) {
    const FtBatchJob &J = jobs[blockIdx.y];
    if (J.nPoints <= 0) return;
    int *res;
    const FtClaims C = job_claims(J, rb, pass, fCur, fPrev, fReset, res);
    const int lane = threadIdx.x & 63, wave = wave_index();
    __shared__ int cacheCounter[FT_SEARCH_WPB];
    const FtLastRaw raw = {nullptr, nullptr, nullptr, nullptr};
    FtDevLastPoints L = J.L;
    L.valid = rb(L.valid); L.worldPos = rb(L.worldPos); L.desc = rb(L.desc); L.octave = rb(L.octave);
    const FramePtrs Q = frame_ptrs(J.F, rb);
    if constexpr (!slowList) {
        if (!claims_begin_pass(C)) return;
        const int i = blockIdx.x * FT_SEARCH_WPB + wave;
        if (i >= J.L.N) return;
        int r4[4];
        last_point<true>(J.F, Q, L, C, J.Tcw, th, J.forward, J.backward, i, lane, r4, raw, &cacheCounter[wave], rb(J.proj));
        claims_file(C, res, i, lane, r4);
        return;
    } else {
    if (C.flagPrev && shared_load(C.flagPrev) == -1) return;
    const int *slow = rb(J.slow);
    const int count = slow[pass & 1];
    for (int k = blockIdx.x * FT_SEARCH_WPB + wave; k < count; k += gridDim.x * FT_SEARCH_WPB) {
        const int i = slow[16 + (size_t)(pass & 1) * J.nPoints + k];
        int r4[4];
        last_point<true>(J.F, Q, L, C, J.Tcw, th, J.forward, J.backward, i, lane, r4, raw, &cacheCounter[wave], rb(J.proj));
        claims_file(C, res, i, lane, r4);
    }
    }
}

template <bool slowList>
__global__ __launch_bounds__(64 * FT_SEARCH_WPB) __attribute__((amdgpu_waves_per_eu(FT_BATCH_WAVES, 8))) void k_search_local_batch(const FtBatchJob *__restrict__ jobs, Rebase rb, int pass, int fCur,
                                                                           int fPrev, int fReset, float th, float nnRatio) {
    const FtBatchJob &J = jobs[blockIdx.y];
    if (J.nPoints <= 0) return;
    int *res;
    const FtClaims C = job_claims(J, rb, pass, fCur, fPrev, fReset, res);
    const int lane = threadIdx.x & 63, wave = wave_index();
    __shared__ int cacheCounter[FT_SEARCH_WPB];
    FtLocalRaw raw;
    raw.bestDist = nullptr;
    FtDevLocalPoints P = J.P;
    P.skip = rb(P.skip); P.inView = rb(P.inView); P.inViewR = rb(P.inViewR);
    P.level = rb(P.level); P.levelR = rb(P.levelR);
    P.viewCos = rb(P.viewCos); P.viewCosR = rb(P.viewCosR);
    P.projX = rb(P.projX); P.projY = rb(P.projY); P.projXR = rb(P.projXR); P.projYR = rb(P.projYR);
    P.desc = rb(P.desc);
    const FramePtrs Q = frame_ptrs(J.F, rb);
    if constexpr (!slowList) {
        if (!claims_begin_pass(C)) return;
        const int i = blockIdx.x * FT_SEARCH_WPB + wave;
        if (i >= J.P.M) return;
        int r4[4];
        local_point(J.F, Q, P, C, th, nnRatio, i, lane, r4, raw, &cacheCounter[wave]);
        claims_file(C, res, i, lane, r4);
        return;
    } else {
    if (C.flagPrev && shared_load(C.flagPrev) == -1) return;
    const int *slow = rb(J.slow);
    const int count = slow[pass & 1];
    for (int k = blockIdx.x * FT_SEARCH_WPB + wave; k < count; k += gridDim.x * FT_SEARCH_WPB) {
        const int i = slow[16 + (size_t)(pass & 1) * J.nPoints + k];
        int r4[4];
        local_point(J.F, Q, P, C, th, nnRatio, i, lane, r4, raw, &cacheCounter[wave]);
        claims_file(C, res, i, lane, r4);
    }
    }
}

// ---- first pass of a batch, four points per wave --------------------------------------------------------------------------
// The general kernel gives the window scan of ONE point a whole wave: at th 7 a window holds ~50 candidates of a few (octave,
// column) ranges - most lanes idle through ~600 instructions per point.  Here a point is a ROW of 16 lanes from the start
// (as in the lean kernels of the later passes): the ranges of its window one per lane (a DPP scan inside the row lays them end
// to end), the entries 16 at a time (the range of an entry by a few row-local shuffles), the candidates filed in the point's
// cache at positions handed out by a ballot of the row (no LDS counter), the minimum (two minima) by DPP steps inside the row.
// What is computed per entry - box, level band, uright test, Hamming distance, key - and what is filed are exactly the
// general kernel's (the order of a list is free), so the later passes cannot tell which kernel ran the first one.  First pass:
// nothing is locked but what was held before the call.
__device__ __forceinline__ int row_shfl(int v, int srcLane) { return __shfl(v, srcLane); }
// the two smallest keys of a row, in every lane of it
__device__ __forceinline__ void row_two_min(unsigned long long &k0, unsigned long long &k1) {
    const unsigned long long m0 = row_min_u64(k0);
    const unsigned long long cand = (k0 == m0) ? k1 : k0;
    k1 = row_min_u64(cand);
    k0 = m0;
}

// The window scan of a row's point, in three steps through a small LDS list of the row (FT_ROW_LIST entries; the lanes of a row
// belong to one wave, whose LDS operations are served in order - no barrier):
//   expand  a lane per (octave, column of cells) range, a DPP scan lays the ranges end to end, and every range lane writes the
//           grid positions of its entries (with the column in the top byte) at their places in the list - where the first form of
//           this loop looked the range of every entry up again, 16 entries at a time, by a chain of np - 1 shuffles;
//   filter  16 entries at a time: the 16-byte record, level band and box test of GetFeaturesInArea (in_box), the survivors packed
//           to the front of the list by a ballot of the row - the cell ranges of a window hold ~2.4 x the keypoints of the box, and
//           the other 58 % leave here without their descriptor having been loaded;
//   visit   fn(entry, real) for the survivors, 16 at a time: descriptor, cell row, and whatever the search does with them.
// Windows with more entries than the list holds go through it in parts.  What fn sees is what it saw before minus the entries
// in_box rejects (the order inside a list is free).
#define FT_ROW_LIST 64
__device__ __forceinline__ void row_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <class Fn>
__device__ __forceinline__ void row_for_window(const FtDevFrame &F, const FramePtrs &Q, int cam, const Window &w, int minLevel, int maxLevel,
                                               float bx, float by, float br, int sub, int rowBase, unsigned *list, Fn fn) {
    const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
    const int lo = checkLevels ? min(max(minLevel, 0), F.nlevels - 1) : 0;
    const int hi = (checkLevels && maxLevel >= 0) ? min(maxLevel, F.nlevels - 1) : F.nlevels - 1;
    const int ncolsW = w.maxCX - w.minCX + 1;
    const int npairs = (hi - lo + 1) * ncolsW;  // (<= 0: an empty band)
    const int *gs = Q.gridStart[cam];
    const float4 *rec = Q.gridRec[cam];
    const uint4 *gd = (const uint4 *)Q.gridDesc[cam];
    const unsigned colMagic = div_magic_u(ncolsW);
    for (int p0 = 0; p0 < npairs; p0 += 16) {  // (row-uniform)
        const int np = min(16, npairs - p0);
        int b = 0, cnt = 0, myCol = 0;
        if (sub < np) {
            const int pidx = p0 + sub;
            const int oi = colMagic ? (int)__umulhi((unsigned)pidx, colMagic) : pidx;
            myCol = w.minCX + (pidx - oi * ncolsW);
            const int *col = gs + (size_t)(lo + oi) * (FT_GRID_CELLS + 1) + myCol * FT_GRID_ROWS;
            b = col[w.minCY];
            cnt = col[w.maxCY + 1] - b;
        }
        int incl = cnt;  // inclusive scan over the 16 lanes of the row (lanes shifted in from outside the row read 0)
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);  // row_shr:1
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);  // row_shr:2
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);  // row_shr:4
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);  // row_shr:8
        const int total = row_shfl(incl, rowBase + 15);
        const int start = incl - cnt;
        const unsigned tag = (unsigned)b | ((unsigned)myCol << 24);  // (grid positions stay below 2^24, columns below 64)
        for (int w0 = 0; w0 < total; w0 += FT_ROW_LIST) {  // (row-uniform)
            const int nw = min(FT_ROW_LIST, total - w0);
            // expand: the part of this lane's range that falls into [w0, w0 + nw)
            for (int k = max(0, w0 - start), k1 = min(cnt, w0 + nw - start); k < k1; k++) list[start + k - w0] = tag + (unsigned)k;
            row_lds_sync();
            // filter: survivors of the box to the front (an entry is read before its group writes, and a group writes below its
            // own first entry + 16)
            int m = 0;
            for (int t0 = 0; t0 < nw; t0 += 16) {  // (row-uniform)
                const int t = t0 + sub;
                const unsigned v = list[min(t, nw - 1)];
                const float4 rr = rec[v & 0xffffffu];
                WinEntry e;
                e.x = rr.x; e.y = rr.y;
                e.octave = __float_as_int(rr.w) >> 24;
                const bool inb = t < nw && in_box(e, bx, by, br, minLevel, maxLevel);
                const unsigned bits = (unsigned)(__ballot(inb) >> rowBase) & 0xffffu;
                if (inb) list[m + __popc(bits & ((1u << sub) - 1u))] = v;
                m += __popc(bits);
            }
            row_lds_sync();
            // visit
            for (int s0 = 0; s0 < m; s0 += 16) {  // (row-uniform)
                const int sI = s0 + sub;
                const unsigned v = list[min(sI, m - 1)];
                const int pos = (int)(v & 0xffffffu);
                const float4 rr = rec[pos];
                const uint4 d0 = gd[2 * (size_t)pos], d1 = gd[2 * (size_t)pos + 1];
                WinEntry e;
                e.x = rr.x; e.y = rr.y; e.uright = rr.z;
                const int io = __float_as_int(rr.w);
                e.idx = io & 0xffffff;
                e.octave = io >> 24;
                e.cx = (int)(v >> 24);
                e.cy = (int)roundf(__fmul_rn(__fsub_rn(rr.y, F.mnMinY), F.invH));
                e.d[0] = (unsigned long long)d0.x | ((unsigned long long)d0.y << 32);
                e.d[1] = (unsigned long long)d0.z | ((unsigned long long)d0.w << 32);
                e.d[2] = (unsigned long long)d1.x | ((unsigned long long)d1.y << 32);
                e.d[3] = (unsigned long long)d1.z | ((unsigned long long)d1.w << 32);
                fn(e, sI < m);
            }
            row_lds_sync();  // (the next part - or the next window - overwrites the list)
        }
    }
}
// a candidate key into the point's list: positions by a ballot of the row (n = candidates filed so far, row-uniform)
__device__ __forceinline__ void row_cache_append(unsigned long long *slot, int &n, bool cand, unsigned long long key, int sub, int rowBase) {
    const unsigned bits = (unsigned)(__ballot(cand) >> rowBase) & 0xffffu;
    if (cand) {
        const int pos = n + __popc(bits & ((1u << sub) - 1u));
        if (pos < FT_CACHE_CAP) slot[1 + pos] = key;
    }
    n += __popc(bits);
}
__device__ __forceinline__ void row_cache_end(unsigned long long *slot, int n, bool anyInBox, int sub) {
    if (sub == 0)
        slot[0] = (unsigned long long)(unsigned)n | ((unsigned long long)(anyInBox ? 1 : 0) << 32) | ((unsigned long long)(unsigned)min(n, FT_CACHE_CAP) << 40);
}
__device__ __forceinline__ bool row_any(bool v, int rowBase) { return ((unsigned)(__ballot(v) >> rowBase) & 0xffffu) != 0u; }
// the first pass's claims_file for the point of a row: no previous results, every result counts as changed (the flag was set by
// claims_begin_pass)
__device__ __forceinline__ void claims_file_row_first(const FtClaims &C, int *res, int i, int sub, const int r4[4]) {
    if (sub < 4) {
        const int kp = sub == 0 ? r4[0] : sub == 1 ? r4[1] : sub == 2 ? r4[2] : r4[3];
        const int s = 4 * i + sub;
        res[s] = kp;
        if (kp >= 0) {
            const int e = (s << 1) | (C.obs[i] > 0 ? 1 : 0);
            int *rec = C.tabWrite + 8 * (size_t)kp;
            const int pos = atomicAdd(rec, 1) + 1;
            if (pos < FT_TAB_ENTRIES) rec[1 + pos] = e;
            else C.nextWrite[s] = atomicExch(&C.headWrite[kp], e);
        }
    }
}

__global__ __launch_bounds__(256) void k_search_last_first(const FtBatchJob *__restrict__ jobs, Rebase rb, float th, FtSlotGrid sg) {
    int frame, blk;
    if (!ft_slot_block(sg, frame, blk)) return;
    const FtBatchJob &J = jobs[frame];
    if (J.nPoints <= 0) return;
    int *res;
    const FtClaims C = job_claims(J, rb, 0, 0, -1, FT_BATCH_FLAGS / 2, res);
    claims_begin_pass(C, blk, sg.blocksPerSlot);
    __shared__ unsigned rowLists[16][FT_ROW_LIST];
    unsigned *list = rowLists[threadIdx.x >> 4];
    const int lane = threadIdx.x & 63, sub = lane & 15, rowBase = lane & 48;
    const int i = blk * 16 + (threadIdx.x >> 4);
    if (i >= J.L.N) return;
    const FtDevFrame &F = J.F;
    const FramePtrs Q = frame_ptrs(F, rb);
    const bool twoCam = F.Nleft != -1;
    unsigned long long *slotL = C.cache + (size_t)i * FT_CACHE_WORDS, *slotR = slotL + (FT_CACHE_CAP + 1);
    int primL = -1, primR = -1;
    if (rb(J.L.valid)[i]) {
        const FtLastProj pj = rb(J.proj)[i];
        if (!pj.go) {
            if (sub == 0) slotL[0] = 0ull;  // does not project into the image: an empty list spares the later passes the question
        } else {
            const int oct = rb(J.L.octave)[i];
            const float radius = __fmul_rn(th, F.sf[oct]);
            int minLevel, maxLevel;
            if (J.forward) { minLevel = oct; maxLevel = -1; }
            else if (J.backward) { minLevel = 0; maxLevel = oct; }
            else { minLevel = oct - 1; maxLevel = oct + 1; }
            unsigned long long q[4];
            {
                const unsigned long long *p = (const unsigned long long *)(rb(J.L.desc) + (size_t)i * 32);
                q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; q[3] = p[3];
            }
            const float u = pj.u, v = pj.v;
            const Window w = cell_window(F, u, v, radius);
            unsigned long long k0 = KEY_NONE;
            int n = 0;
            bool anyCand = false;
            if (!w.empty) {
                row_for_window(F, Q, 0, w, minLevel, maxLevel, u, v, radius, sub, rowBase, list, [&](const WinEntry &kp, bool real) {
                    const bool inb = real;  // (in_box held in the filter step)
                    anyCand = anyCand || inb;
                    bool cand = inb;
                    if (cand && kp.uright > 0) {
                        const float ur = __fsub_rn(u, __fmul_rn(F.mbf, pj.invzc));
                        if (fabsf(__fsub_rn(ur, kp.uright)) > radius) cand = false;
                    }
                    const bool held = cand && Q.holderObs[cand ? kp.idx : 0] > 0;
                    const unsigned long long key = make_key(hamming256(q, kp.d), kp.cx, kp.cy, kp.idx, kp.octave, held);
                    row_cache_append(slotL, n, cand, key, sub, rowBase);
                    if (cand && !held) k0 = key < k0 ? key : k0;
                });
            }
            anyCand = row_any(anyCand, rowBase);
            row_cache_end(slotL, n, anyCand, sub);
            k0 = row_min_u64(k0);
            if (anyCand) {  // `if(vIndices2.empty()) continue;` (ORBmatcher.cc:1836) also skips the right-camera block
                if (k0 != KEY_NONE && key_dist(k0) <= FT_TH_HIGH) primL = key_idx(k0);
                if (twoCam) {
                    const float ur = pj.ur, vr = pj.vr;
                    const Window wr = cell_window(F, ur, vr, radius);
                    unsigned long long kr = KEY_NONE;
                    int nr = 0;
                    if (!wr.empty) {
                        row_for_window(F, Q, 1, wr, minLevel, maxLevel, ur, vr, radius, sub, rowBase, list, [&](const WinEntry &kp, bool real) {
                            const bool cand = real;
                            const bool held = cand && Q.holderObs[(cand ? kp.idx : 0) + F.Nleft] > 0;
                            const unsigned long long key = make_key(hamming256(q, kp.d), kp.cx, kp.cy, kp.idx, kp.octave, held);
                            row_cache_append(slotR, nr, cand, key, sub, rowBase);
                            if (cand && !held) kr = key < kr ? key : kr;
                        });
                    }
                    row_cache_end(slotR, nr, false, sub);
                    kr = row_min_u64(kr);
                    if (kr != KEY_NONE && key_dist(kr) <= FT_TH_HIGH) primR = key_idx(kr) + F.Nleft;
                }
            }
        }
    }
    const int r4[4] = {primL, -1, primR, -1};
    claims_file_row_first(C, res, i, sub, r4);
}

__global__ __launch_bounds__(256) void k_search_local_first(const FtBatchJob *__restrict__ jobs, Rebase rb, float th, float nnRatio, FtSlotGrid sg) {
    int frame, blk;
    if (!ft_slot_block(sg, frame, blk)) return;
    const FtBatchJob &J = jobs[frame];
    if (J.nPoints <= 0) return;
    int *res;
    const FtClaims C = job_claims(J, rb, 0, 0, -1, FT_BATCH_FLAGS / 2, res);
    claims_begin_pass(C, blk, sg.blocksPerSlot);
    __shared__ unsigned rowLists[16][FT_ROW_LIST];
    unsigned *list = rowLists[threadIdx.x >> 4];
    const int lane = threadIdx.x & 63, sub = lane & 15, rowBase = lane & 48;
    const int i = blk * 16 + (threadIdx.x >> 4);
    if (i >= J.P.M) return;
    const FtDevFrame &F = J.F;
    const FramePtrs Q = frame_ptrs(F, rb);
    const bool twoCam = F.Nleft != -1;
    unsigned long long *slotL = C.cache + (size_t)i * FT_CACHE_WORDS, *slotR = slotL + (FT_CACHE_CAP + 1);
    const uint8_t skipV = rb(J.P.skip)[i], inViewV = rb(J.P.inView)[i], inViewRV = twoCam ? rb(J.P.inViewR)[i] : (uint8_t)0;
    const int levelRV = twoCam ? rb(J.P.levelR)[i] : -1;
    const int obsI = C.obs[i];
    int primL = -1, sideL = -1, primR = -1, sideR = -1;
    bool skipRight = false;
    if (!skipV) {
        unsigned long long q[4];
        {
            const unsigned long long *p = (const unsigned long long *)(rb(J.P.desc) + (size_t)i * 32);
            q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; q[3] = p[3];
        }
        if (inViewV) {
            const int level = rb(J.P.level)[i];
            float r = ((double)rb(J.P.viewCos)[i] > 0.998) ? 2.5f : 4.0f;  // RadiusByViewingCos, ORBmatcher.cc:314-320
            if ((double)th != 1.0) r = __fmul_rn(r, th);
            const float rad = __fmul_rn(r, F.sf[level]);
            const float x = rb(J.P.projX)[i], y = rb(J.P.projY)[i];
            const Window w = cell_window(F, x, y, rad);
            unsigned long long k0 = KEY_NONE, k1 = KEY_NONE;
            int n = 0;
            if (!w.empty) {
                const float pxr = (F.Nleft == -1 && Q.uright) ? rb(J.P.projXR)[i] : 0.f;
                row_for_window(F, Q, 0, w, level - 1, level, x, y, rad, sub, rowBase, list, [&](const WinEntry &kp, bool real) {
                    bool cand = real;
                    if (cand && kp.uright > 0 && fabsf(__fsub_rn(pxr, kp.uright)) > rad) cand = false;  // (mono-stereo frames only)
                    const bool held = cand && Q.holderObs[cand ? kp.idx : 0] > 0;
                    const unsigned long long key = make_key(hamming256(q, kp.d), kp.cx, kp.cy, kp.idx, kp.octave, held);
                    row_cache_append(slotL, n, cand, key, sub, rowBase);
                    if (cand && !held) two_min_insert(k0, k1, key);
                });
            }
            row_cache_end(slotL, n, false, sub);
            row_two_min(k0, k1);
            int bd = 256, bd2 = 256, bl = -1, bl2 = -1, bi = -1;
            if (k0 != KEY_NONE) { bd = key_dist(k0); bi = key_idx(k0); bl = key_octave(k0); }
            if (k1 != KEY_NONE) { bd2 = key_dist(k1); bl2 = key_octave(k1); }
            if (bd <= FT_TH_HIGH) {
                if (bl == bl2 && (float)bd > __fmul_rn(nnRatio, (float)bd2)) skipRight = true;
                else {
                    primL = bi;
                    if (twoCam) {
                        const int m = Q.l2r[bi];
                        if (m != -1) sideL = m + F.Nleft;
                    }
                }
            }
        }
        // (a point whose left block ended in the ratio test's `continue` files its right-camera candidates all the same: a later
        // pass may get past the test - the locks decide - and would otherwise have to come back here through the slow list)
        if (twoCam && inViewRV && levelRV != -1) {
            const int level = levelRV;
            const float r = ((double)rb(J.P.viewCosR)[i] > 0.998) ? 2.5f : 4.0f;
            const float rad = __fmul_rn(r, F.sf[level]);
            const float x = rb(J.P.projXR)[i], y = rb(J.P.projYR)[i];
            const Window w = cell_window(F, x, y, rad);
            unsigned long long k0 = KEY_NONE, k1 = KEY_NONE;
            int n = 0;
            if (!w.empty) {
                row_for_window(F, Q, 1, w, level - 1, level, x, y, rad, sub, rowBase, list, [&](const WinEntry &kp, bool real) {
                    const bool cand = real;
                    const int g = kp.idx + F.Nleft;
                    const bool held = cand && Q.holderObs[cand ? g : 0] > 0;
                    // this point's own left-block side write precedes its right-block search
                    const bool locked = (g == sideL) ? (obsI > 0) : held;
                    const unsigned long long key = make_key(hamming256(q, kp.d), kp.cx, kp.cy, kp.idx, kp.octave, held);
                    row_cache_append(slotR, n, cand, key, sub, rowBase);
                    if (cand && !locked) two_min_insert(k0, k1, key);
                });
            }
            row_cache_end(slotR, n, false, sub);
            row_two_min(k0, k1);
            int bdr = 256, bd2r = 256, blr = -1, bl2r = -1, bir = -1;
            if (k0 != KEY_NONE) { bdr = key_dist(k0); bir = key_idx(k0); blr = key_octave(k0); }
            if (k1 != KEY_NONE) { bd2r = key_dist(k1); bl2r = key_octave(k1); }
            if (!skipRight && bdr <= FT_TH_HIGH && !(blr == bl2r && (float)bdr > __fmul_rn(nnRatio, (float)bd2r))) {
                const int m = Q.r2l[bir];
                if (m != -1) sideR = m;
                primR = bir + F.Nleft;
            }
        }
    }
    const int r4[4] = {primL, sideL, primR, sideR};
    claims_file_row_first(C, res, i, sub, r4);
}

// ---- later passes of a batch: the lean kernels --------------------------------------------------------------------------------
// From the second pass on nearly every point finds its candidates in the cache the first pass filed, and its turn is a
// handful of loads: the cached keys, the 32-byte writer records of their keypoints, a minimum.  The general kernels above
// spend a whole wave (and ~90 registers, 30 KB of code) on it.  Here a point is a ROW of 16 lanes - four points per wave, the
// keys 16 at a time, the two smallest by DPP steps that never leave the row - and a point the cache cannot serve (a camera's
// candidates not filed yet: the right block is reached for the first time; more candidates than the cache holds) is handed
// to the general kernel through the frame's slow list (launched behind this one with slowList = 1).  Same reads of the
// previous pass's records, same keys, same comparisons: the results are those of the general kernel.
// claims_file for the point of a row: lane `sub` (0 .. 3) of the row files write kind sub
__device__ __forceinline__ void claims_file_row(const FtClaims &C, int *res, int i, int sub, const int r4[4]) {
    if (sub < 4) {
        const int kp = sub == 0 ? r4[0] : sub == 1 ? r4[1] : sub == 2 ? r4[2] : r4[3];
        const int s = 4 * i + sub;
        const int prev = shared_load(&C.resPrev[s]);
        if (kp != prev) atomicAnd(C.flagCur, 0);
        shared_store(&res[s], kp);
        if (kp >= 0) {
            const int e = (s << 1) | (C.obs[i] > 0 ? 1 : 0);
            int *rec = C.tabWrite + 8 * (size_t)kp;
            const int pos = atomicAdd(rec, 1) + 1;
            if (pos < FT_TAB_ENTRIES) shared_store(rec + 1 + pos, e);
            else shared_store(&C.nextWrite[s], atomicExch(&C.headWrite[kp], e));
        }
    }
}
__device__ __forceinline__ void slow_append(int *slow, int pass, int nPoints, int i) {
    const int pos = atomicAdd(&slow[pass & 1], 1);
    slow[16 + (size_t)(pass & 1) * nPoints + pos] = i;
}
#define FT_LEAN_PPB 16  // points per workgroup of the lean kernels: 4 waves x 4 rows

__global__ __launch_bounds__(256) void k_search_local_lean(const FtBatchJob *__restrict__ jobs, Rebase rb, int pass, int fCur, int fPrev,
                                                           int fReset, float nnRatio) {
    const FtBatchJob &J = jobs[blockIdx.y];
    if (J.nPoints <= 0) return;
    int *res;
    const FtClaims C = job_claims(J, rb, pass, fCur, fPrev, fReset, res);
    if (!claims_begin_pass(C)) return;
    int *slow = rb(J.slow);
    if (blockIdx.x == 0 && threadIdx.x == 0) slow[(pass + 1) & 1] = 0;  // the next pass's list starts empty
    const int sub = threadIdx.x & 15;
    const int i = blockIdx.x * FT_LEAN_PPB + (threadIdx.x >> 4);
    if (i >= J.P.M) return;
    const FtDevFrame &F = J.F;
    const bool twoCam = F.Nleft != -1;
    const uint8_t *skipP = rb(J.P.skip), *inViewP = rb(J.P.inView), *inViewRP = rb(J.P.inViewR);
    const int *levelRP = rb(J.P.levelR);
    const unsigned long long *slotL = C.cache + (size_t)i * FT_CACHE_WORDS, *slotR = slotL + (FT_CACHE_CAP + 1);
    const uint8_t skipV = skipP[i], inViewV = inViewP[i], inViewRV = twoCam ? inViewRP[i] : (uint8_t)0;
    const int levelRV = twoCam ? levelRP[i] : -1;
    const unsigned long long metaL = slotL[0], metaR = twoCam ? slotR[0] : KEY_NONE;
    const int obsI = C.obs[i];
    // A pass is a chain of dependent round trips, and the chip is full of such chains: the first 16 keys of both cameras'
    // lists are requested together with the flags and the meta words, and the lock records of both - the right camera's on
    // the chance that its block is reached - in ONE further trip (flags -> meta -> keys -> records left -> l2r -> records right
    // used to be six).  A key beyond a list's head is not a key: its "record" is the one of keypoint 0, read and dropped.
    const unsigned long long keyL0 = slotL[1 + sub], keyR0 = twoCam ? slotR[1 + sub] : KEY_NONE;
    const int headL0 = metaL == KEY_NONE ? 0 : min(cache_head(metaL), FT_CACHE_CAP), headR0 = metaR == KEY_NONE ? 0 : min(cache_head(metaR), FT_CACHE_CAP);
    const bool haveL0 = sub < headL0, haveR0 = sub < headR0;
    const int kpL0 = haveL0 ? key_idx(keyL0) : 0, kpR0 = haveR0 ? key_idx(keyR0) + F.Nleft : 0;
    const LockRec recL0 = lock_record(C, kpL0), recR0 = lock_record(C, kpR0);
    int primL = -1, sideL = -1, primR = -1, sideR = -1;
    bool skipRight = false, slowPoint = false;
    if (!skipV) {
        if (inViewV) {
            int nCached;
            bool anyBox;
            if (cache_state_of(metaL, nCached, anyBox) != 1) slowPoint = true;
            else {
                unsigned long long k0 = KEY_NONE, k1 = KEY_NONE;
                const int head = cache_head(metaL);
                if (haveL0 && !locked_by(C, recL0, kpL0, i, key_held(keyL0))) k0 = keyL0;
                for (int t = 16 + sub; t < head; t += 16) {
                    const unsigned long long key = slotL[1 + t];
                    if (is_locked(C, key_idx(key), i, key_held(key))) continue;
                    two_min_insert(k0, k1, key);
                }
                row_two_min(k0, k1);
                if (k1 == KEY_NONE && head < nCached) {  // fewer than two unlocked keys in the head: the rest of the list decides
                    for (int t = head + sub; t < nCached; t += 16) {
                        const unsigned long long key = slotL[1 + t];
                        if (is_locked(C, key_idx(key), i, key_held(key))) continue;
                        two_min_insert(k0, k1, key);
                    }
                    row_two_min(k0, k1);
                }
                int bd = 256, bd2 = 256, bl = -1, bl2 = -1, bi = -1;
                if (k0 != KEY_NONE) { bd = key_dist(k0); bi = key_idx(k0); bl = key_octave(k0); }
                if (k1 != KEY_NONE) { bd2 = key_dist(k1); bl2 = key_octave(k1); }
                if (bd <= FT_TH_HIGH) {
                    if (bl == bl2 && (float)bd > __fmul_rn(nnRatio, (float)bd2)) skipRight = true;
                    else {
                        primL = bi;
                        if (twoCam) {
                            const int m = rb(F.l2r)[bi];
                            if (m != -1) sideL = m + F.Nleft;
                        }
                    }
                }
            }
        }
        if (!slowPoint && twoCam && inViewRV && !skipRight && levelRV != -1) {
            int nCached;
            bool anyBox;
            if (cache_state_of(metaR, nCached, anyBox) != 1) slowPoint = true;
            else {
                unsigned long long k0 = KEY_NONE, k1 = KEY_NONE;
                const int head = cache_head(metaR);
                auto scan = [&](int from, int to) {
                    for (int t = from + sub; t < to; t += 16) {
                        const unsigned long long key = slotR[1 + t];
                        const int g = key_idx(key) + F.Nleft;
                        const bool locked = (g == sideL) ? (obsI > 0) : is_locked(C, g, i, key_held(key));
                        if (locked) continue;
                        two_min_insert(k0, k1, key);
                    }
                    row_two_min(k0, k1);
                };
                if (haveR0 && !((kpR0 == sideL) ? (obsI > 0) : locked_by(C, recR0, kpR0, i, key_held(keyR0)))) k0 = keyR0;
                scan(16, head);
                if (k1 == KEY_NONE && head < nCached) scan(head, nCached);
                int bdr = 256, bd2r = 256, blr = -1, bl2r = -1, bir = -1;
                if (k0 != KEY_NONE) { bdr = key_dist(k0); bir = key_idx(k0); blr = key_octave(k0); }
                if (k1 != KEY_NONE) { bd2r = key_dist(k1); bl2r = key_octave(k1); }
                if (bdr <= FT_TH_HIGH && !(blr == bl2r && (float)bdr > __fmul_rn(nnRatio, (float)bd2r))) {
                    const int m = rb(F.r2l)[bir];
                    if (m != -1) sideR = m;
                    primR = bir + F.Nleft;
                }
            }
        }
    }
    if (slowPoint) {
        if (sub == 0) slow_append(slow, pass, J.nPoints, i);
        return;
    }
    const int r4[4] = {primL, sideL, primR, sideR};
    claims_file_row(C, res, i, sub, r4);
}

__global__ __launch_bounds__(256) void k_search_last_lean(const FtBatchJob *__restrict__ jobs, Rebase rb, int pass, int fCur, int fPrev,
                                                          int fReset) {
    const FtBatchJob &J = jobs[blockIdx.y];
    if (J.nPoints <= 0) return;
    int *res;
    const FtClaims C = job_claims(J, rb, pass, fCur, fPrev, fReset, res);
    if (!claims_begin_pass(C)) return;
    int *slow = rb(J.slow);
    if (blockIdx.x == 0 && threadIdx.x == 0) slow[(pass + 1) & 1] = 0;
    const int sub = threadIdx.x & 15;
    const int i = blockIdx.x * FT_LEAN_PPB + (threadIdx.x >> 4);
    if (i >= J.L.N) return;
    const FtDevFrame &F = J.F;
    const bool twoCam = F.Nleft != -1;
    const unsigned long long *slotL = C.cache + (size_t)i * FT_CACHE_WORDS, *slotR = slotL + (FT_CACHE_CAP + 1);
    const uint8_t validV = rb(J.L.valid)[i];
    const unsigned long long metaL = slotL[0], metaR = twoCam ? slotR[0] : KEY_NONE;
    // (as in k_search_local_lean: the first 16 keys of both lists with the meta words, their lock records in one further trip)
    const unsigned long long keyL0 = slotL[1 + sub], keyR0 = twoCam ? slotR[1 + sub] : KEY_NONE;
    const int headL0 = metaL == KEY_NONE ? 0 : min(cache_head(metaL), FT_CACHE_CAP), headR0 = metaR == KEY_NONE ? 0 : min(cache_head(metaR), FT_CACHE_CAP);
    const bool haveL0 = sub < headL0, haveR0 = sub < headR0;
    const int kpL0 = haveL0 ? key_idx(keyL0) : 0, kpR0 = haveR0 ? key_idx(keyR0) + F.Nleft : 0;
    const LockRec recL0 = lock_record(C, kpL0), recR0 = lock_record(C, kpR0);
    int primL = -1, primR = -1;
    if (validV) {
        int nCachedL = 0, nCachedR = 0;
        bool anyBoxL = false, anyBoxR = false;
        bool fromCache = false;
        if (cache_state_of(metaL, nCachedL, anyBoxL) == 1) fromCache = !twoCam || !anyBoxL || cache_state_of(metaR, nCachedR, anyBoxR) == 1;
        if (!fromCache) {
            if (sub == 0) slow_append(slow, pass, J.nPoints, i);
            return;
        }
        auto scanMin = [&](const unsigned long long *slot, int from, int to, int base) -> unsigned long long {
            unsigned long long m = KEY_NONE;
            for (int t = from + sub; t < to; t += 16) {
                const unsigned long long key = slot[1 + t];
                if (is_locked(C, key_idx(key) + base, i, key_held(key))) continue;
                m = key < m ? key : m;
            }
            return row_min_u64(m);
        };
        // the head of the list first (cache_partition): an unlocked key there is smaller than every key behind it
        const int headL = cache_head(metaL);
        const unsigned long long firstL = (haveL0 && !locked_by(C, recL0, kpL0, i, key_held(keyL0))) ? keyL0 : KEY_NONE;
        unsigned long long k0 = headL > 16 ? scanMin(slotL, 16, headL, 0) : KEY_NONE;
        {
            const unsigned long long m = row_min_u64(firstL);
            k0 = m < k0 ? m : k0;
        }
        if (k0 == KEY_NONE && headL < nCachedL) k0 = scanMin(slotL, headL, nCachedL, 0);
        if (anyBoxL) {
            if (k0 != KEY_NONE && key_dist(k0) <= FT_TH_HIGH) primL = key_idx(k0);
            if (twoCam) {
                const int headR = cache_head(metaR);
                const unsigned long long firstR = (haveR0 && !locked_by(C, recR0, kpR0, i, key_held(keyR0))) ? keyR0 : KEY_NONE;
                unsigned long long kr = headR > 16 ? scanMin(slotR, 16, headR, F.Nleft) : KEY_NONE;
                {
                    const unsigned long long m = row_min_u64(firstR);
                    kr = m < kr ? m : kr;
                }
                if (kr == KEY_NONE && headR < nCachedR) kr = scanMin(slotR, headR, nCachedR, F.Nleft);
                if (kr != KEY_NONE && key_dist(kr) <= FT_TH_HIGH) primR = key_idx(kr) + F.Nleft;
            }
        }
    }
    const int r4[4] = {primL, -1, primR, -1};
    claims_file_row(C, res, i, sub, r4);
}

// ---- a batch's claims resolved in ONE launch: the points of a frame in index order, a chunk at a time --------------------------
// The passes above are a Jacobi iteration over ALL points of a frame: a point's locks depend on the writes of the points in
// front of it, a dependency chain of length c takes c passes, and every pass re-evaluates every point (13 - 20 passes of a batch
// at configs[3]).  But the dependency is triangular, and a batch has parallelism to spare ACROSS its frames.  So: one workgroup
// per frame walks the frame's points in index order, FT_RS_ROWS at a time (a point = a row of 16 lanes, as in the lean kernels).
// When a chunk is evaluated every point in front of it is FINAL: of their writes a keypoint needs to remember only the last
// (lastW[kp], an atomicMax of the writer-table entry: the largest (4 point + kind) - what locked_by picks from a record), and
// only the writes of the chunk's own points are still in motion - they are iterated inside the workgroup, on a hash table in
// LDS (keypoint -> bit mask of the chunk's rows that write it), until an iteration changes nothing.  An iteration after the
// first touches LDS only (keys and lastW values stay in registers).  A point is evaluated 2 - 3 times instead of 13 - 20, and
// a search is the first pass (window scans, k_search_*_first), the partition of the lists and this.
// It reads the candidate lists the first pass filed; a point whose list is not usable (more candidates than the cache holds)
// makes the workgroup give up on its frame: the frame's flag words stay as the first pass left them, the host sees it and
// continues with the passes above for such frames (resolved frames are inert there: all their flag words read "converged").
#ifndef FT_RS_W
#define FT_RS_W 16                       // lanes per point (a GROUP of lanes inside a DPP row): 16 or 8
#endif
#define FT_RS_LANES 1024                 // a workgroup
#define FT_RS_ROWS (FT_RS_LANES / FT_RS_W)   // points per chunk: 64 (128 with 8 lanes per point)
#define FT_RS_SLOTS (8 * FT_RS_ROWS)     // hash slots (<= 4 writes per point and chunk): a power of two
#define FT_RS_REG (48 / FT_RS_W)         // keys of a list's head a lane keeps in registers (x FT_RS_W lanes = FT_CACHE_HEAD_MAX)
#define FT_RS_MW (FT_RS_ROWS / 32)       // 32-bit words of a row mask
static_assert((FT_RS_W == 8 || FT_RS_W == 16) && FT_RS_W * FT_RS_REG == FT_CACHE_HEAD_MAX, "k_resolve_batch: a point is 8 or 16 lanes");
// (Round 6 measured 8 lanes per point - eight points per wave, 16 chunks of 128 points instead of 32 of 64: the last-frame
// resolution took the same 0.24 ms, the local-map one 0.63 instead of 0.46 (th 15: 0.50 / 1.20 against 0.41 / 0.94) - twice the
// points per chunk are more than twice the chunk: more of them collide inside it (more iterations), and six key registers per
// lane and camera spill.  EXPERIMENTS 11.9.)
struct RsShared {
    // two hash tables used alternately by the iterations of a chunk (iteration `it` reads table it & 1 and clears the other
    // one for its successor): keypoint -> rows of the chunk that write it (their results of the previous iteration)
    int kp[2][FT_RS_SLOTS];
    unsigned mask[2][FT_RS_MW][FT_RS_SLOTS];
    unsigned char obs[FT_RS_ROWS];  // Observations() > 0 of the chunk's points
    int vote[3];                    // "iteration it changed a result", slot it % 3
};
__device__ __forceinline__ unsigned rs_hash(int kp) { return ((unsigned)kp * 2654435761u) >> (32 - __builtin_ctz(FT_RS_SLOTS)); }
__device__ __forceinline__ void rs_clear(RsShared &S, int t) {
    for (int k = threadIdx.x; k < FT_RS_SLOTS; k += FT_RS_LANES) {
        S.kp[t][k] = -1;
#pragma unroll
        for (int w = 0; w < FT_RS_MW; w++) S.mask[t][w][k] = 0u;
    }
}
// a barrier for what the workgroup exchanges through LDS: outstanding loads from memory (the next chunk's prefetch) stay outstanding
__device__ __forceinline__ void rs_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ void rs_insert(RsShared &S, int t, int kp, int row) {
    unsigned h = rs_hash(kp);
    for (;;) {
        const int old = atomicCAS(&S.kp[t][h], -1, kp);
        if (old == -1 || old == kp) break;
        h = (h + 1) & (FT_RS_SLOTS - 1);
    }
    atomicOr(&S.mask[t][row >> 5][h], 1u << (row & 31));
}
// F.mvpMapPoints[kp] && ->Observations() > 0 as the point of row `row` sees it: the last writer in front of it - of this chunk
// (hash table t; useHash = 0: the chunk's first iteration, no writes of the chunk yet) or, if none, of the chunks before (lw = lastW[kp]) -
// decides, else the pre-call holder
__device__ __forceinline__ bool rs_locked(const RsShared &S, int t, bool useHash, int kp, int lw, bool held, int row) {
    if (useHash) {
        unsigned h = rs_hash(kp);
        for (;;) {
            const int k = S.kp[t][h];
            if (k == -1) break;
            if (k == kp) {
                // the highest row below `row` that writes the keypoint: the word of `row` cut off at its bit, then the words below
                int w = row >> 5;
                unsigned m = S.mask[t][w][h] & ((1u << (row & 31)) - 1u);
                while (m == 0u && w > 0) m = S.mask[t][--w][h];
                if (m) return S.obs[32 * w + 31 - __clz((int)m)] != 0;
                break;
            }
            h = (h + 1) & (FT_RS_SLOTS - 1);
        }
    }
    return lw >= 0 ? (lw & 1) != 0 : held;
}
// minima / maxima over the FT_RS_W lanes of a point, in every lane of it: DPP steps that stay inside the group (lane pairs, quads,
// halves of a row - and, for sixteen lanes, the row)
__device__ __forceinline__ unsigned long long grp_min_u64(unsigned long long v) {
#define FT_MIN64_STEP(ctrl)                                                                                   \
    {                                                                                                         \
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, ctrl, 0xF, 0xF, true); \
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(v >> 32), ctrl, 0xF, 0xF, true);   \
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;                                     \
        v = w < v ? w : v;                                                                                    \
    }
    FT_MIN64_STEP(0xB1) FT_MIN64_STEP(0x4E) FT_MIN64_STEP(0x141)
    if constexpr (FT_RS_W == 16) FT_MIN64_STEP(0x140)
#undef FT_MIN64_STEP
    return v;
}
__device__ __forceinline__ int grp_max_i32(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));
    if constexpr (FT_RS_W == 16) v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));
    return v;
}
__device__ __forceinline__ void grp_two_min(unsigned long long &k0, unsigned long long &k1) {
    const unsigned long long m0 = grp_min_u64(k0);
    const unsigned long long cand = (k0 == m0) ? k1 : k0;
    k1 = grp_min_u64(cand);
    k0 = m0;
}
// The last writers of the points in front of the running chunk, one word per keypoint of the frame.  LWLDS (round 6): the table
// lives in the workgroup's LDS for the whole walk (F.N ints: 16 KB at configs[3]) - a chunk's publication is an LDS atomic and
// the next chunk's look-ups are LDS reads, where round 5 went through L2 both ways (atomicMax, then device-scope loads that had
// to wait for it: one memory round trip on every chunk's critical path, 32 chunks per frame).  Frames too large for the LDS
// keep the table in HBM (buffer 0 of the list heads).
template <bool LWLDS>
__device__ __forceinline__ int rs_last_writer(const int *lastW, int kp) {
    if constexpr (LWLDS) return lastW[kp];
    else return __hip_atomic_load(lastW + kp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// results of a converged chunk: both result buffers (the host reads the one of the parity it is told), lastW for the chunks behind
__device__ __forceinline__ void rs_publish(int *res0, int *res1, int *lastW, int i, int sub, bool obsI, const int r4[4]) {
    if (sub < 4) {
        const int kp = sub == 0 ? r4[0] : sub == 1 ? r4[1] : sub == 2 ? r4[2] : r4[3];
        const int s = 4 * i + sub;
        res0[s] = kp;
        res1[s] = kp;
        if (kp >= 0) atomicMax(lastW + kp, (s << 1) | (obsI ? 1 : 0));  // (LDS or HBM: the address space decides the instruction)
    }
}
// what a point's turn needs that no other point's result changes - requested a chunk ahead
struct RsStatic {
    unsigned long long metaL, metaR, kL[FT_RS_REG], kR[FT_RS_REG];
    int obs;
    unsigned char f0, f1, f2;  // local map: skip, inView, inViewR; last frame: valid
    int levelR;
};
template <bool LOCAL>
__device__ __forceinline__ RsStatic rs_fetch(const FtBatchJob &J, const Rebase &rb, const unsigned long long *cache, const int *obsP, bool twoCam,
                                             int i, int sub) {
    RsStatic T;
    const unsigned long long *slotL = cache + (size_t)i * FT_CACHE_WORDS, *slotR = slotL + (FT_CACHE_CAP + 1);
    T.metaL = slotL[0];
    T.metaR = twoCam ? slotR[0] : KEY_NONE;
#pragma unroll
    for (int j = 0; j < FT_RS_REG; j++) {  // (whatever the lists' lengths: a key beyond a head is dropped when the meta word is there)
        T.kL[j] = slotL[1 + sub + FT_RS_W * j];
        T.kR[j] = twoCam ? slotR[1 + sub + FT_RS_W * j] : KEY_NONE;
    }
    T.obs = obsP[i];
    T.levelR = -1;
    if constexpr (LOCAL) {
        T.f0 = rb(J.P.skip)[i];
        T.f1 = rb(J.P.inView)[i];
        T.f2 = twoCam ? rb(J.P.inViewR)[i] : (unsigned char)0;
        if (twoCam) T.levelR = rb(J.P.levelR)[i];
    } else {
        T.f0 = rb(J.L.valid)[i];
        T.f1 = T.f2 = 0;
    }
    return T;
}

template <bool LOCAL, bool LWLDS>
__global__ __launch_bounds__(FT_RS_LANES) void k_resolve_batch(const FtBatchJob *__restrict__ jobs, Rebase rb, float nnRatio) {
    const FtBatchJob &J = jobs[blockIdx.x];
    if (J.nPoints <= 0) return;
    __shared__ RsShared S;
    extern __shared__ int rs_lw[];  // LWLDS: the frame's last-writer table
    const FtDevFrame &F = J.F;
    const bool twoCam = F.Nleft != -1;
    const int M = LOCAL ? J.P.M : J.L.N;
    const int row = threadIdx.x / FT_RS_W, sub = threadIdx.x % FT_RS_W;
    int *res0 = rb(J.res), *res1 = res0 + 4 * (size_t)J.nPoints;
    // (HBM form: buffer 0 of the list heads, all -1 after k_fill_claims_batch and not written by a first pass)
    int *lastW = LWLDS ? rs_lw : rb(J.head);
    if constexpr (LWLDS)
        for (int k = threadIdx.x; k < F.N; k += FT_RS_LANES) rs_lw[k] = -1;  // (the pre-scan's barrier below orders it)
    const int *obsP = rb(J.obs);
    const unsigned long long *cache = rb(J.cache);
    const int *l2r = rb(F.l2r), *r2l = rb(F.r2l);
    if (threadIdx.x < 3) S.vote[threadIdx.x] = 0;
    rs_clear(S, 1);  // (the table of a chunk's second iteration; the pre-scan's barrier below orders it)
    RsStatic T = rs_fetch<LOCAL>(J, rb, cache, obsP, twoCam, min(row, M - 1), sub);
    // Usable or not is decided for the WHOLE frame before the first chunk publishes anything (results, last writers): the meta
    // words of every list the frame's points will want, a point per lane.  A frame the kernel gives up on is untouched - the
    // claim passes that take over read the first pass's results and an all -1 last-writer buffer, as if this kernel had not run.
    // (Round 5 tested chunk by chunk: a list beyond the cache in a later chunk left the earlier chunks published.)
    {
        bool unusable = false;
        for (int p = threadIdx.x; p < M; p += FT_RS_LANES) {
            const unsigned long long *slotL = cache + (size_t)p * FT_CACHE_WORDS;
            const unsigned long long mL = slotL[0], mR = twoCam ? slotL[FT_CACHE_CAP + 1] : KEY_NONE;
            int nL = 0, nR = 0;
            bool boxL = false, boxR = false;
            const int stL = cache_state_of(mL, nL, boxL), stR = cache_state_of(mR, nR, boxR);
            bool wantL, wantR;
            if constexpr (LOCAL) {
                const bool skip = rb(J.P.skip)[p] != 0;
                wantL = !skip && rb(J.P.inView)[p] != 0;
                wantR = !skip && twoCam && rb(J.P.inViewR)[p] != 0 && rb(J.P.levelR)[p] != -1;
            } else {
                wantL = rb(J.L.valid)[p] != 0;
                wantR = wantL && twoCam && stL == 1 && boxL;
            }
            unusable = unusable || (wantL && stL != 1) || (wantR && stR != 1);
        }
        if (__syncthreads_or(unusable ? 1 : 0)) return;  // (the frame's flag words untouched: the host goes on with the passes)
    }
    for (int base = 0; base < M; base += FT_RS_ROWS) {  // (uniform)
        const int i = base + row;
        const bool act = i < M;
        const int ii = act ? i : M - 1;
        const unsigned long long *slotL = cache + (size_t)ii * FT_CACHE_WORDS, *slotR = slotL + (FT_CACHE_CAP + 1);
        bool wantL, wantR;
        if constexpr (LOCAL) {
            wantL = act && !T.f0 && T.f1;
            wantR = act && !T.f0 && twoCam && T.f2 && T.levelR != -1;
        } else {
            wantL = act && T.f0;
            wantR = wantL && twoCam;
        }
        const bool obsI = T.obs > 0;
        int nL = 0, nR = 0;
        bool anyBoxL = false, anyBoxR = false;
        const int stL = cache_state_of(T.metaL, nL, anyBoxL), stR = cache_state_of(T.metaR, nR, anyBoxR);
        (void)stR;
        if constexpr (!LOCAL) wantR = wantR && stL == 1 && anyBoxL;  // (`if(vIndices2.empty()) continue;` skips the right-camera block)
        const int headL = wantL ? cache_head(T.metaL) : 0, headR = wantR ? cache_head(T.metaR) : 0;
        if (!wantL) nL = 0;
        if (!wantR) nR = 0;
        unsigned long long kL[FT_RS_REG], kR[FT_RS_REG];
        int wL[FT_RS_REG], wR[FT_RS_REG], mL[FT_RS_REG], mR[FT_RS_REG];  // last writers; the keypoints' entries of the match tables
#pragma unroll
        for (int j = 0; j < FT_RS_REG; j++) {
            kL[j] = (sub + FT_RS_W * j < headL) ? T.kL[j] : KEY_NONE;
            kR[j] = (sub + FT_RS_W * j < headR) ? T.kR[j] : KEY_NONE;
        }
#pragma unroll
        for (int j = 0; j < FT_RS_REG; j++) {
            wL[j] = kL[j] != KEY_NONE ? rs_last_writer<LWLDS>(lastW, key_idx(kL[j])) : -1;
            wR[j] = kR[j] != KEY_NONE ? rs_last_writer<LWLDS>(lastW, key_idx(kR[j]) + F.Nleft) : -1;
            mL[j] = mR[j] = -1;
            if constexpr (LOCAL) {
                if (twoCam) {
                    if (kL[j] != KEY_NONE) mL[j] = l2r[key_idx(kL[j])];
                    if (kR[j] != KEY_NONE) mR[j] = r2l[key_idx(kR[j])];
                }
            }
        }
        if (base + FT_RS_ROWS < M) T = rs_fetch<LOCAL>(J, rb, cache, obsP, twoCam, min(i + FT_RS_ROWS, M - 1), sub);  // the next chunk's
        if (sub == 0) S.obs[row] = obsI ? 1 : 0;  // (read behind the barrier of the second iteration)
        if (threadIdx.x == 0) S.vote[1] = 0;      // (the second iteration's slot; the later ones are reset an iteration ahead)
        int r4[4] = {-1, -1, -1, -1};
        int it = 0;
        for (;; it++) {  // (uniform)
            const bool useHash = it > 0;
            const int ht = it & 1;
            // Two barriers per iteration behind the first: table ht is CLEAN here (cleared while the iteration before the last
            // one - or the previous chunk - was inserting: a barrier ago at least), the rows file their writes of the previous
            // iteration in it and clear the other table for the next iteration, barrier, everybody evaluates against it, barrier,
            // the vote.  (Round 5: one table, cleared between two barriers of its own - four barriers per iteration.)
            if (it > 0) {
                if (act && sub < 4) {
                    const int kp = sub == 0 ? r4[0] : sub == 1 ? r4[1] : sub == 2 ? r4[2] : r4[3];
                    if (kp >= 0) rs_insert(S, ht, kp, row);
                }
                rs_clear(S, ht ^ 1);
                if (threadIdx.x == 0) S.vote[(it + 1) % 3] = 0;  // (slot of the next iteration: last read two barriers ago)
                rs_barrier();
            }
            int primL = -1, sideL = -1, primR = -1, sideR = -1;
            if constexpr (LOCAL) {
                bool skipRight = false;
                if (wantL) {
                    unsigned long long k0 = KEY_NONE, k1 = KEY_NONE;
#pragma unroll
                    for (int j = 0; j < FT_RS_REG; j++)
                        if (kL[j] != KEY_NONE && !rs_locked(S, ht, useHash, key_idx(kL[j]), wL[j], key_held(kL[j]), row)) two_min_insert(k0, k1, kL[j]);
                    auto scan = [&](int from, int to) {
                        for (int t = from + sub; t < to; t += FT_RS_W) {
                            const unsigned long long key = slotL[1 + t];
                            const int kp = key_idx(key);
                            if (rs_locked(S, ht, useHash, kp, rs_last_writer<LWLDS>(lastW, kp), key_held(key), row)) continue;
                            two_min_insert(k0, k1, key);
                        }
                    };
                    if (headL > FT_RS_W * FT_RS_REG) scan(FT_RS_W * FT_RS_REG, headL);
                    grp_two_min(k0, k1);
                    if (k1 == KEY_NONE && headL < nL) {  // fewer than two unlocked keys in the head: the rest of the list decides
                        scan(headL, nL);
                        grp_two_min(k0, k1);
                    }
                    int bd = 256, bd2 = 256, bl = -1, bl2 = -1, bi = -1;
                    if (k0 != KEY_NONE) { bd = key_dist(k0); bi = key_idx(k0); bl = key_octave(k0); }
                    if (k1 != KEY_NONE) { bd2 = key_dist(k1); bl2 = key_octave(k1); }
                    if (bd <= FT_TH_HIGH) {
                        if (bl == bl2 && (float)bd > __fmul_rn(nnRatio, (float)bd2)) skipRight = true;
                        else {
                            primL = bi;
                            if (twoCam) {  // l2r[bi]: with the winner's lane, or (a key from beyond the registers) in memory
                                int m = INT_MIN;
#pragma unroll
                                for (int j = 0; j < FT_RS_REG; j++) m = kL[j] == k0 ? mL[j] : m;
                                m = grp_max_i32(m);
                                if (m == INT_MIN) m = l2r[bi];
                                if (m != -1) sideL = m + F.Nleft;
                            }
                        }
                    }
                }
                if (wantR && !skipRight) {
                    unsigned long long k0 = KEY_NONE, k1 = KEY_NONE;
                    // this point's own left-block side write precedes its right-block search
                    auto lockedR = [&](int g, int lw, bool held) -> bool { return (g == sideL) ? obsI : rs_locked(S, ht, useHash, g, lw, held, row); };
#pragma unroll
                    for (int j = 0; j < FT_RS_REG; j++)
                        if (kR[j] != KEY_NONE && !lockedR(key_idx(kR[j]) + F.Nleft, wR[j], key_held(kR[j]))) two_min_insert(k0, k1, kR[j]);
                    auto scan = [&](int from, int to) {
                        for (int t = from + sub; t < to; t += FT_RS_W) {
                            const unsigned long long key = slotR[1 + t];
                            const int g = key_idx(key) + F.Nleft;
                            if (lockedR(g, rs_last_writer<LWLDS>(lastW, g), key_held(key))) continue;
                            two_min_insert(k0, k1, key);
                        }
                    };
                    if (headR > FT_RS_W * FT_RS_REG) scan(FT_RS_W * FT_RS_REG, headR);
                    grp_two_min(k0, k1);
                    if (k1 == KEY_NONE && headR < nR) {
                        scan(headR, nR);
                        grp_two_min(k0, k1);
                    }
                    int bdr = 256, bd2r = 256, blr = -1, bl2r = -1, bir = -1;
                    if (k0 != KEY_NONE) { bdr = key_dist(k0); bir = key_idx(k0); blr = key_octave(k0); }
                    if (k1 != KEY_NONE) { bd2r = key_dist(k1); bl2r = key_octave(k1); }
                    if (bdr <= FT_TH_HIGH && !(blr == bl2r && (float)bdr > __fmul_rn(nnRatio, (float)bd2r))) {
                        int m = INT_MIN;
#pragma unroll
                        for (int j = 0; j < FT_RS_REG; j++) m = kR[j] == k0 ? mR[j] : m;
                        m = grp_max_i32(m);
                        if (m == INT_MIN) m = r2l[bir];
                        if (m != -1) sideR = m;
                        primR = bir + F.Nleft;
                    }
                }
            } else {
                auto listMin = [&](const unsigned long long *slot, const unsigned long long *kReg, const int *wReg, int head, int n, int off) {
                    unsigned long long m = KEY_NONE;
#pragma unroll
                    for (int j = 0; j < FT_RS_REG; j++)
                        if (kReg[j] != KEY_NONE && !rs_locked(S, ht, useHash, key_idx(kReg[j]) + off, wReg[j], key_held(kReg[j]), row))
                            m = kReg[j] < m ? kReg[j] : m;
                    auto scan = [&](int from, int to) {
                        for (int t = from + sub; t < to; t += FT_RS_W) {
                            const unsigned long long key = slot[1 + t];
                            const int g = key_idx(key) + off;
                            if (rs_locked(S, ht, useHash, g, rs_last_writer<LWLDS>(lastW, g), key_held(key), row)) continue;
                            m = key < m ? key : m;
                        }
                    };
                    if (head > FT_RS_W * FT_RS_REG) scan(FT_RS_W * FT_RS_REG, head);
                    m = grp_min_u64(m);
                    // the head of the list first (cache_partition): an unlocked key there is smaller than every key behind it
                    if (m == KEY_NONE && head < n) {
                        scan(head, n);
                        m = grp_min_u64(m);
                    }
                    return m;
                };
                if (wantL && anyBoxL) {
                    const unsigned long long k0 = listMin(slotL, kL, wL, headL, nL, 0);
                    if (k0 != KEY_NONE && key_dist(k0) <= FT_TH_HIGH) primL = key_idx(k0);
                    if (wantR) {
                        const unsigned long long kr = listMin(slotR, kR, wR, headR, nR, F.Nleft);
                        if (kr != KEY_NONE && key_dist(kr) <= FT_TH_HIGH) primR = key_idx(kr) + F.Nleft;
                    }
                }
            }
            const bool changed = act && (primL != r4[0] || sideL != r4[1] || primR != r4[2] || sideR != r4[3]);
            r4[0] = primL; r4[1] = sideL; r4[2] = primR; r4[3] = sideR;
            if (it == 0) continue;  // (the first iteration's results are what the second one starts from, changed or not)
            if (changed && sub == 0) S.vote[it % 3] = 1;
            rs_barrier();
            if (!S.vote[it % 3]) break;
#ifdef FT_RS_MAXIT
            if (it >= FT_RS_MAXIT) break;
#endif
        }
#ifndef FT_RS_NOPUB
        if (act) rs_publish(res0, res1, lastW, i, sub, obsI, r4);
#endif
        // The next chunk's rs_last_writer loads must see this chunk's atomicMax.  Both are device-scope operations that execute in
        // L2 (the atomic there, the sc1 load from there), issued by waves of ONE workgroup = one CU, and what orders them is the
        // CU's in-order vector-memory path: the ISA of the fence + barrier below is `s_waitcnt lgkmcnt(0) ; s_barrier` - NO
        // `vmcnt(0)`, the workgroup-scope release waits for nothing of the atomics - so a load issued behind the barrier is behind
        // every atomic issued in front of it in the same CU's queue to the same L2 channel (the same address).  LLVM's AMDGPU
        // memory model guarantees that order only in non-threadgroup-split mode (tgsplit: the waves of a workgroup may sit on
        // different CUs and a workgroup-scope release becomes a real wait); the build refuses tgsplit (csrc/Makefile: check-tgsplit,
        // tests/test_build_flags.py - the compiler defines no macro a static_assert could test).  An
        // agent-scope fence (__threadfence) would be safe everywhere and writes the L2 back, 30 us a time (EXPERIMENTS 10.7).
        // The next chunk's second iteration files into table 1: dirty when this chunk ended in an odd iteration (an even one cleared it)
        if (it & 1) rs_clear(S, 1);
        // (LWLDS: the table is in LDS - an LDS-only barrier, and none of the above applies)
        if constexpr (LWLDS) rs_barrier();
        else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
        }
    }
    // resolved: every flag word of the frame reads "converged"
    int *flags = rb(J.flags);
    if (threadIdx.x < FT_BATCH_FLAGS) flags[threadIdx.x] = -1;
}

// ---- the writes of a converged search, replayed where the results are ------------------------------------------------------------
// What the host did with a search's results until round 5 (replayLocalWrites / replayLastFrameWrites, search.cpp) - and what
// the reference does while it searches: CurrentFrame.mvpMapPoints[kp] = pMP in point order (src/ORBmatcher.cc:134-148, 203-214;
// 1860-1879, 1934-1941), the rotation histogram (:1880-1896, 1942-1957), ComputeThreeMaxima (:2210-2251) and the removal of
// the matches outside the three dominant bins (:1966-1987).  A workgroup per frame:
//   assign[kp]  = the LAST point that wrote keypoint kp (atomicMax of the point index over all writes), -1 if none - or if ANY
//                 write to kp fell into a removed histogram bin (the reference clears mvpMapPoints[kp] for every entry of such a bin,
//                 whoever wrote the keypoint last);
//   holder[kp]  = Observations() of that point, -1 where the histogram removed the keypoint, unchanged where nobody wrote;
//   nm          = writes - writes in removed bins (nmatches++ per write, nmatches-- per removed entry).
// The frame's holder_obs stays in HBM (the next search of the batch reads it there), assign and nm go straight into pinned host
// memory.  The last-writer table lives in LDS (F.N ints) or, for frames beyond it, in the frame's writer table (dead by now).
#define FT_REPLAY_REMOVED 0x40000000
template <bool LOCAL, bool INLDS>
__global__ __launch_bounds__(256) void k_replay_batch(const FtBatchJob *__restrict__ jobs, Rebase rb, int parity, int checkOrientation,
                                                      int flagPos) {
    const FtBatchJob &J = jobs[blockIdx.x];
    extern __shared__ int rp_last[];
    __shared__ int rp_hist[FT_HISTO_LENGTH], rp_keep, rp_sum[4];
    int *replayed = rb(J.replayed);
    const int *flags = rb(J.flags);
    const int N = J.F.N, M = J.nPoints;
    // (uniform) a frame that has been replayed already; flagPos >= 0 (a launch enqueued before the host has seen the flag words -
    // right behind k_resolve_batch: position 0, or behind a burst of claim passes: the burst's last position): only a frame whose
    // flag word there says "converged"; the others wait for the passes still to come
    if (*replayed >= 0 || (flagPos >= 0 && M > 0 && flags[flagPos] != -1)) return;
    const int tid = threadIdx.x;
    int *last = INLDS ? rp_last : rb(J.tab);
    int *assign = J.assignOut;
    int *holder = const_cast<int *>(rb(J.F.holderObs));
    const int *res = rb(J.res) + (size_t)parity * 4 * (size_t)M;
    const int *obs = rb(J.obs);
    const bool hist = !LOCAL && checkOrientation != 0;
    const int nLk = J.F.Nleft == -1 ? N : J.F.Nleft;
    const ft_keypoint *keys = rb(J.F.keys), *keysR = rb(J.F.keysR);
    const float *lastAngle = LOCAL ? nullptr : rb(J.L.angle);
    for (int kp = tid; kp < N; kp += 256) last[kp] = -1;
    if (tid < FT_HISTO_LENGTH) rp_hist[tid] = 0;
    if (!INLDS) __threadfence();
    __syncthreads();
    // rotation bin of the write (point i -> keypoint kp): src/ORBmatcher.cc:1882-1890, the host replay's expression operation by operation
    auto bin_of = [&](int i, int kp) -> int {
        const float cur = kp < nLk ? keys[kp].angle : keysR[kp - nLk].angle;
        float rot = __fsub_rn(lastAngle[i], cur);
        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
        int bin = (int)roundf(__fmul_rn(rot, 1.0f / FT_HISTO_LENGTH));
        if (bin == FT_HISTO_LENGTH) bin = 0;
        return bin;
    };
    int nm = 0;
    for (int s = tid; s < 4 * M; s += 256) {
        if (!LOCAL && (s & 1)) continue;  // last frame: the primary writes of the two cameras (res[4 i], res[4 i + 2])
        const int kp = res[s];
        if (kp < 0) continue;
        const int i = s >> 2;
        nm++;
        atomicMax(&last[kp], i);
        if (hist) {
            const int bin = bin_of(i, kp);
            if (bin >= 0 && bin < FT_HISTO_LENGTH) atomicAdd(&rp_hist[bin], 1);
        }
    }
    if (!INLDS) __threadfence();
    __syncthreads();
    if (hist) {
        if (tid == 0) {  // ComputeThreeMaxima (src/ORBmatcher.cc:2210-2251)
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int b = 0; b < FT_HISTO_LENGTH; b++) {
                const int sz = rp_hist[b];
                if (sz > max1) {
                    max3 = max2; max2 = max1; max1 = sz;
                    ind3 = ind2; ind2 = ind1; ind1 = b;
                } else if (sz > max2) {
                    max3 = max2; max2 = sz;
                    ind3 = ind2; ind2 = b;
                } else if (sz > max3) {
                    max3 = sz; ind3 = b;
                }
            }
            if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
            int keep = 0;
            if (ind1 >= 0) keep |= 1 << ind1;
            if (ind2 >= 0) keep |= 1 << ind2;
            if (ind3 >= 0) keep |= 1 << ind3;
            rp_keep = keep;
        }
        __syncthreads();
        const int keep = rp_keep;
        for (int s = tid; s < 4 * M; s += 256) {
            if (s & 1) continue;
            const int kp = res[s];
            if (kp < 0) continue;
            const int bin = bin_of(s >> 2, kp);
            if (bin >= 0 && bin < FT_HISTO_LENGTH && !((keep >> bin) & 1)) {
                atomicMax(&last[kp], FT_REPLAY_REMOVED);
                nm--;
            }
        }
        if (!INLDS) __threadfence();
        __syncthreads();
    }
    for (int kp = tid; kp < N; kp += 256) {
        const int a = INLDS ? last[kp] : __hip_atomic_load(last + kp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int out = -1;
        if (a >= FT_REPLAY_REMOVED) holder[kp] = -1;
        else if (a >= 0) {
            out = a;
            holder[kp] = obs[a];
        }
        assign[kp] = out;
    }
    nm = wave_sum_i32(nm);
    if ((tid & 63) == 0) rp_sum[tid >> 6] = nm;
    __syncthreads();
    if (tid == 0) {
        const int total = rp_sum[0] + rp_sum[1] + rp_sum[2] + rp_sum[3];
        *J.nmOut = total;
        *replayed = total;
    }
}

// ------------------------------------------------------------------------------------------------
// Frame::isInFrustum / isInFrustumChecks (src/Frame.cc:536-610, 1308-1382) with MapPoint::PredictScale
// (src/MapPoint.cc:531-546): one thread per local map point.  Float expressions are evaluated in the
// order the oracle states (no contraction); log(ratio) binds to logf (MapPoint.cc:539), reproduced by libm_f32.h.
// ------------------------------------------------------------------------------------------------
// Eigen's sum of three terms (dot, squaredNorm, a coefficient of a small matrix product): redux_novec_unroller splits the
// range in halves, e0 + (e1 + e2) (see the oracle's note at orc_is_in_frustum)
__device__ __forceinline__ float dot3(const float *a, const float *b) {
    return __fadd_rn(__fmul_rn(a[0], b[0]), __fadd_rn(__fmul_rn(a[1], b[1]), __fmul_rn(a[2], b[2])));
}
// sqrtf is correctly rounded here (-fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn maps to the native approximation
__device__ __forceinline__ float norm3(const float *a) { return sqrtf(dot3(a, a)); }

__device__ __forceinline__ int predict_scale(float maxDistanceRaw, float dist, float logScaleFactor, int nLevels) {
    const float ratio = __fdiv_rn(maxDistanceRaw, dist);
    const float lg = ft_libm::logf_glibc(ratio);
    int nScale = (int)ceilf(__fdiv_rn(lg, logScaleFactor));
    if (nScale < 0) nScale = 0;
    else if (nScale >= nLevels) nScale = nLevels - 1;
    return nScale;
}

__device__ __forceinline__ void frustum_point(const FtDevFrame &F, const FtFrustumPose &T, const FtDevMapPoints &P, float viewingCosLimit,
                                              float logScaleFactor, int farPoints, float thFar, const FtFrustumOut &O, int i) {
    if (i >= P.M) return;
    bool inView = false, inViewR = false;
    int level = -1, levelR = -1;
    float viewCosL = 0.f, viewCosR = 0.f, px = -1.f, py = -1.f, pxr = -1.f, pyr = -1.f, depth = 0.f, depthR = 0.f;
    if (!(P.skip && P.skip[i])) {
        const float Pw[3] = {P.worldPos[3 * i], P.worldPos[3 * i + 1], P.worldPos[3 * i + 2]};
        const float Pn[3] = {P.normal[3 * i], P.normal[3 * i + 1], P.normal[3 * i + 2]};
        const float maxRaw = P.maxDist[i];
        const float maxDistance = __fmul_rn(1.2f, maxRaw), minDistance = __fmul_rn(0.8f, P.minDist[i]);
        const int nCams = F.Nleft == -1 ? 1 : 2;
        for (int cam = 0; cam < nCams; cam++) {
            float Pc[3];
#pragma unroll
            for (int r = 0; r < 3; r++) Pc[r] = __fadd_rn(dot3(T.R[cam] + 3 * r, Pw), T.t[cam][r]);
            const float PcDist = norm3(Pc);
            if (Pc[2] < 0.0f) continue;
            float uv[2];
            project_cam(F, Pc, uv);
            if (uv[0] < F.mnMinX || uv[0] > F.mnMaxX) continue;
            if (uv[1] < F.mnMinY || uv[1] > F.mnMaxY) continue;
            if (F.Nleft == -1) {  // Frame.cc:564-565: set before the remaining checks
                px = uv[0];
                py = uv[1];
            }
            const float PO[3] = {__fsub_rn(Pw[0], T.twc[cam][0]), __fsub_rn(Pw[1], T.twc[cam][1]), __fsub_rn(Pw[2], T.twc[cam][2])};
            const float dist = norm3(PO);
            if (dist < minDistance || dist > maxDistance) continue;
            const float viewCos = __fdiv_rn(dot3(PO, Pn), dist);
            if (viewCos < viewingCosLimit) continue;
            const int lv = predict_scale(maxRaw, dist, logScaleFactor, F.nlevels);
            if (cam == 0) {
                inView = true;
                px = uv[0];
                py = uv[1];
                level = lv;
                viewCosL = viewCos;
                depth = PcDist;
                if (F.Nleft == -1) pxr = __fsub_rn(uv[0], __fmul_rn(F.mbf, __fdiv_rn(1.0f, Pc[2])));  // mTrackProjXR (:587)
            } else {
                inViewR = true;
                pxr = uv[0];
                pyr = uv[1];
                levelR = lv;
                viewCosR = viewCos;
                depthR = PcDist;
            }
        }
    }
    O.inView[i] = inView;
    O.inViewR[i] = inViewR;
    O.level[i] = level;
    O.levelR[i] = levelR;
    O.viewCos[i] = viewCosL;
    O.viewCosR[i] = viewCosR;
    O.projX[i] = px;
    O.projY[i] = py;
    O.projXR[i] = pxr;
    O.projYR[i] = pyr;
    O.depth[i] = depth;
    O.depthR[i] = depthR;
    // ORBmatcher.cc:66-74: not in view of either camera, or farther than thFarPoints (the caller's skip holds isBad())
    if (O.searchSkip)
        O.searchSkip[i] = (!inView && !inViewR) || (farPoints && depth > thFar) || (P.skip && P.skip[i]);
    if (inView || inViewR) atomicAdd(O.count, 1);
}
__global__ __launch_bounds__(256) void k_frustum(FtDevFrame F, FtFrustumPose T, FtDevMapPoints P, float viewingCosLimit,
                                                 float logScaleFactor, int farPoints, float thFar, FtFrustumOut O) {
    frustum_point(F, T, P, viewingCosLimit, logScaleFactor, farPoints, thFar, O, blockIdx.x * 256 + threadIdx.x);
}
// isInFrustum for the local map points of every frame of a batch: blockIdx.y = frame (the counts are zeroed by the launcher)
__global__ __launch_bounds__(256) void k_frustum_batch(const FtBatchJob *__restrict__ jobs, Rebase rb, float viewingCosLimit,
                                                       float logScaleFactor, int farPoints, float thFar) {
    const FtBatchJob &J = jobs[blockIdx.y];
    FtDevMapPoints P = J.MP;
    P.skip = rb(P.skip); P.worldPos = rb(P.worldPos); P.normal = rb(P.normal); P.maxDist = rb(P.maxDist); P.minDist = rb(P.minDist);
    FtFrustumOut O = J.O;
    O.inView = rb(O.inView); O.inViewR = rb(O.inViewR); O.level = rb(O.level); O.levelR = rb(O.levelR);
    O.viewCos = rb(O.viewCos); O.viewCosR = rb(O.viewCosR); O.projX = rb(O.projX); O.projY = rb(O.projY);
    O.projXR = rb(O.projXR); O.projYR = rb(O.projYR); O.depth = rb(O.depth); O.depthR = rb(O.depthR);
    O.searchSkip = rb(O.searchSkip); O.count = rb(O.count);
    frustum_point(J.F, J.T, P, viewingCosLimit, logScaleFactor, farPoints, thFar, O, blockIdx.x * 256 + threadIdx.x);
}

// Result delivery of a search: up to three device blocks (dword granularity) written straight into pinned host memory by
// one kernel - pass results, raw outputs / frustum fields, and the pass flags - instead of one DMA copy each (a small copy
// is a few microseconds of work behind tens of microseconds of queueing).
struct FtBlocks {
    void *dst[3];
    const void *src[3];
    int words[3];
};
__global__ __launch_bounds__(256) void k_deliver_blocks(FtBlocks b) {
    const int t = blockIdx.x * 256 + threadIdx.x, T = gridDim.x * 256;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        unsigned *d = (unsigned *)b.dst[k];
        const unsigned *s = (const unsigned *)b.src[k];
        for (int i = t; i < b.words[k]; i += T) d[i] = s[i];
    }
}

__global__ __launch_bounds__(256) void k_fill_stride_u64(unsigned long long *p, int n, int strideWords, unsigned long long v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[(size_t)i * strideWords] = v;
}

// start of a claim iteration: list heads, flags and writer table = -1, the cache's meta words = ~0 ("not built") - one launch
__global__ __launch_bounds__(256) void k_fill_claims(int *p, int n, unsigned long long *meta, int nMeta, int strideWords) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = -1;
    if (i < nMeta) meta[(size_t)i * strideWords] = ~0ull;
}

__global__ __launch_bounds__(256) void k_fill_i32(int *p, int n, int v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// start of the claim iteration of every frame of a batch (blockIdx.y = frame): list heads and writer table = -1 (27 K words
// behind J.head: layoutBatch, search.cpp), the frame's FT_BATCH_FLAGS flag words = -1, the cache's meta words = ~0, the frustum count = 0
__global__ __launch_bounds__(256) void k_fill_claims_batch(const FtBatchJob *__restrict__ jobs, Rebase rb) {
    const FtBatchJob &J = jobs[blockIdx.y];
    const int t = blockIdx.x * 256 + threadIdx.x, T = gridDim.x * 256;
    int *count = rb(J.O.count);
    if (t == 0 && count) *count = 0;
    int *replayed = rb(J.replayed);
    if (t == 0 && replayed) *replayed = -1;  // (k_replay_batch: this search's writes have not been replayed yet)
    int *err = rb(J.err);
    if (t == 0 && err) *err = 0;
    int *head = rb(J.head), *flags = rb(J.flags);
    if (t < FT_BATCH_FLAGS) flags[t] = -1;  // (also of a frame without points: "converged" is what the host reads there)
    if (J.nPoints <= 0) return;
    int *slow = rb(J.slow);
    if (t < 16) slow[t] = 0;
    unsigned long long *cache = rb(J.cache);
    const int words = 27 * J.K;
    for (int i = t; i < words; i += T) head[i] = -1;
    if (cache)
        for (int i = t; i < 2 * J.nPoints; i += T) cache[(size_t)i * (FT_CACHE_CAP + 1)] = ~0ull;
}

// ---- two-camera frames of a batch straight from what two extractors left in HBM (ft_tracked_batch_bind_fisheye) ----
// Step 1, workgroup (camera, frame): the keypoints and descriptors of the extractor's slot into the frame's arrays in the
// REFERENCE's order - ORBextractor::operator() fills keypoints inside the lapping area from the back and the others from the
// front (src/ORBextractor.cc:1466-1487; assembleOutputs, extractor.cpp, does the same for the host copies) - a stable
// partition by ranks from ballots; also: the camera's match table = -1, and the number of keypoints outside the lapping
// area (monoLeft / monoRight, src/Frame.cc:1144-1147) for step 2.
__global__ __launch_bounds__(256) void k_lap_gather_batch(const FtBatchJob *__restrict__ jobs, Rebase rb, FtBindArgs A) {
    const int cam = blockIdx.x, f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const FtDevFrame &F = jobs[f].F;
    const int slot = (cam == 0 ? A.slot0L : A.slot0R) + f;
    const int n = cam == 0 ? F.Nleft : F.N - F.Nleft;
    const ft_keypoint *src = (cam == 0 ? A.keysL : A.keysR) + (size_t)slot * (cam == 0 ? A.strideL : A.strideR);
    const uint4 *srcD = (const uint4 *)((cam == 0 ? A.descL : A.descR) + (size_t)slot * (cam == 0 ? A.strideL : A.strideR) * 32);
    ft_keypoint *dst = (ft_keypoint *)rb(cam == 0 ? F.keys : F.keysR);
    uint4 *dstD = (uint4 *)(rb((uint8_t *)F.desc) + (cam == 0 ? 0 : (size_t)F.Nleft * 32));
    int *tab = (int *)rb(cam == 0 ? F.l2r : F.r2l);
    const float lap0 = (float)(cam == 0 ? A.lapL0 : A.lapR0), lap1 = (float)(cam == 0 ? A.lapL1 : A.lapR1);
    __shared__ int wLap[4];
    int lapBefore = 0;  // lapping-area keypoints in front of this chunk
    for (int base = 0; base < n; base += 256) {
        const int i = base + tid;
        ft_keypoint kp;
        bool inLap = false;
        if (i < n) {
            kp = src[i];
            inLap = kp.x >= lap0 && kp.x <= lap1;
        }
        const unsigned long long b = __ballot(inLap);
        if (lane == 0) wLap[wave] = __popcll(b);
        __syncthreads();
        int before = lapBefore;
        for (int w = 0; w < wave; w++) before += wLap[w];
        const int chunkLap = wLap[0] + wLap[1] + wLap[2] + wLap[3];
        __syncthreads();
        if (i < n) {
            const int rankLap = before + __popcll(b & ((1ull << lane) - 1ull));
            const int d = inLap ? n - 1 - rankLap : i - rankLap;
            dst[d] = kp;
            dstD[2 * (size_t)d] = srcD[2 * (size_t)i];
            dstD[2 * (size_t)d + 1] = srcD[2 * (size_t)i + 1];
            tab[i] = -1;
        }
        lapBefore += chunkLap;
    }
    if (tid == 0) {
        A.mono[2 * f + cam] = n - lapBefore;
        if (cam == 0 && A.nMatches) A.nMatches[f] = 0;
    }
}

// Step 2: the matching part of Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1231-1255; the seam of the reference's
// launchFisheyeStereoMatchKernel, include/Kernels/KernelController.h:38) for every frame of the batch: BFMatcher(NORM_HAMMING)
// .knnMatch(k = 2) of the left lapping subset [monoLeft, Nleft) against the right one + Lowe's ratio 0.7, written as
// mvLeftToRightMatch / mvRightToLeftMatch (a right keypoint matched by several left ones keeps the last = largest index, as the
// reference's loop does).  A wave takes FE_Q queries: a lane holds one train descriptor of the current 64 in registers and
// meets the queries through LDS broadcasts, so a train descriptor is fetched once per FE_Q queries (the one-query-per-wave
// form of k_fisheye_2nn reads the whole train set per query: 128 MB of L2 traffic per 2000 x 2000 frame); keys
// (distance << 20 | train index), two smallest per lane and query, one wave reduction per query at the end.
#define FE_Q 16
// fillR2l = 0: the triangulation filter follows (k_fisheye_triangulate_batch), which writes mvRightToLeftMatch for the pairs it keeps
__global__ __launch_bounds__(256) void k_fisheye_2nn_batch(const FtBatchJob *__restrict__ jobs, Rebase rb, const int *__restrict__ mono,
                                                           int fillR2l) {
    const int f = blockIdx.y, lane = threadIdx.x & 63, wave = wave_index();
    const FtDevFrame &F = jobs[f].F;
    const int monoL = mono[2 * f], monoR = mono[2 * f + 1];
    const int nQ = F.Nleft - monoL, nT = (F.N - F.Nleft) - monoR;
    const int q0 = (blockIdx.x * 4 + wave) * FE_Q;
    if (q0 >= nQ) return;
    const uint8_t *desc = rb(F.desc);
    const uint4 *qd = (const uint4 *)(desc + (size_t)(monoL + q0) * 32);
    const uint4 *td = (const uint4 *)(desc + (size_t)(F.Nleft + monoR) * 32);
    __shared__ uint4 qs[4][FE_Q * 2];
    const int nq = min(FE_Q, nQ - q0);
    if (lane < 2 * nq) qs[wave][lane] = qd[lane];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    unsigned k0[FE_Q], k1[FE_Q];
#pragma unroll
    for (int q = 0; q < FE_Q; q++) k0[q] = k1[q] = 0xffffffffu;
    // (the next train descriptor is requested before the current one is compared with the sixteen queries: 320 vector
    // instructions cover its round trip, where two waves per SIMD - 175 registers - could not)
    uint4 an = make_uint4(0, 0, 0, 0), bn = an;
    if (lane < nT) {
        an = td[2 * (size_t)lane];
        bn = td[2 * (size_t)lane + 1];
    }
    for (int j = lane; j < nT; j += 64) {
        const uint4 a = an, b = bn;
        if (j + 64 < nT) {
            an = td[2 * (size_t)(j + 64)];
            bn = td[2 * (size_t)(j + 64) + 1];
        }
#pragma unroll
        for (int q = 0; q < FE_Q; q++) {
            const uint4 x = qs[wave][2 * q], y = qs[wave][2 * q + 1];  // (same address in every lane: a broadcast)
            const unsigned d = __popc(a.x ^ x.x) + __popc(a.y ^ x.y) + __popc(a.z ^ x.z) + __popc(a.w ^ x.w) + __popc(b.x ^ y.x) +
                               __popc(b.y ^ y.y) + __popc(b.z ^ y.z) + __popc(b.w ^ y.w);
            const unsigned key = (d << 20) | (unsigned)j;
            k1[q] = min(k1[q], max(k0[q], key));
            k0[q] = min(k0[q], key);
        }
    }
    int *l2r = (int *)rb(F.l2r), *r2l = (int *)rb(F.r2l);
#pragma unroll
    for (int q = 0; q < FE_Q; q++) {
        const unsigned m0 = wave_min_u32(k0[q]);
        const unsigned cand = (k0[q] == m0) ? k1[q] : k0[q];
        const unsigned m1 = wave_min_u32(cand);
        if (lane == 0 && q < nq) {
            const int d0 = (int)(m0 >> 20), d1 = (int)(m1 >> 20);
            if (nT >= 2 && (double)(float)d0 < (double)(float)d1 * 0.7) {
                const int t = monoR + (int)(m0 & 0xfffffu), qi = monoL + q0 + q;
                l2r[qi] = t;
                if (fillR2l) atomicMax(&r2l[t], qi);
            }
        }
    }
}

// the candidate lists the first pass of a batch filed: the best candidates to the front (cache_partition).  A wave takes
// FT_PART_LISTS lists one after the other (their meta words requested together: most lists are short and need nothing - a wave
// per list was bound by the rate waves can be launched at), with as many keys per lane as the list's length asks for
#define FT_PART_LISTS 4
__global__ __launch_bounds__(256) void k_cache_partition_batch(const FtBatchJob *__restrict__ jobs, Rebase rb) {
    const FtBatchJob &J = jobs[blockIdx.y];
    const int lane = threadIdx.x & 63, li0 = (blockIdx.x * 4 + wave_index()) * FT_PART_LISTS;
    const int nLists = 2 * J.nPoints;
    if (li0 >= nLists) return;
    unsigned long long *cache = rb(J.cache);
    auto slot_of = [&](int li) { return cache + (size_t)(li >> 1) * FT_CACHE_WORDS + (size_t)(li & 1) * (FT_CACHE_CAP + 1); };
    unsigned long long metas[FT_PART_LISTS];
#pragma unroll
    for (int k = 0; k < FT_PART_LISTS; k++) metas[k] = li0 + k < nLists ? slot_of(li0 + k)[0] : KEY_NONE;
#pragma unroll
    for (int k = 0; k < FT_PART_LISTS; k++) {
        const unsigned long long meta = metas[k];
        int n;
        bool anyBox;
        if (cache_state_of(meta, n, anyBox) != 1 || n <= FT_CACHE_HEAD_MAX || cache_head(meta) != n) continue;  // (wave-uniform)
        unsigned long long *slot = slot_of(li0 + k);
        int head;
        if (n <= 128) head = cache_partition<2>(slot, n, lane);
        else if (n <= 256) head = cache_partition<4>(slot, n, lane);
        else head = cache_partition<(FT_CACHE_CAP + 63) / 64>(slot, n, lane);
        if (lane == 0) slot[0] = (meta & ~(0x3ffull << 40)) | ((unsigned long long)(unsigned)head << 40);
    }
}

// Result delivery of a batch: record r (blockIdx.y) = one block of dwords written into pinned host memory; src[parity] lets a
// record follow the result buffer of the pass that ran last.
// The copy is bound by PCIe (a few hundred workgroups' stores in flight saturate it), so a record gets few workgroups that move
// 16 bytes per lane: the rest of the chip stays free for the kernels of the other batches in flight.
__global__ __launch_bounds__(256) void k_deliver_batch(const FtDeliverRec *__restrict__ recs, int parity) {
    const FtDeliverRec &R = recs[blockIdx.y];
    unsigned *d = (unsigned *)R.dst;
    const unsigned *s = (const unsigned *)R.src[parity];
    const int t = blockIdx.x * 256 + threadIdx.x, T = gridDim.x * 256;
    int done = 0;
    if ((((unsigned long long)(size_t)d | (unsigned long long)(size_t)s) & 15ull) == 0ull) {  // (uniform)
        const int quads = R.words >> 2;
        for (int i = t; i < quads; i += T) ((uint4 *)d)[i] = ((const uint4 *)s)[i];
        done = quads << 2;
    }
    for (int i = done + t; i < R.words; i += T) d[i] = s[i];
}

// The caller's point arrays, read in place out of pinned host memory, into the batch's arena: record r (blockIdx.y) = one array.
// 16 bytes per lane where source and destination allow it; the copy is PCIe-bound like the delivery, few workgroups per record.
__global__ __launch_bounds__(256) void k_gather_batch(const FtGatherRec *__restrict__ recs) {
    const FtGatherRec &R = recs[blockIdx.y];
    uint8_t *d = (uint8_t *)R.dst;
    const uint8_t *s = (const uint8_t *)R.src;
    const unsigned bytes = R.bytes;
    const unsigned t = blockIdx.x * 256 + threadIdx.x, T = gridDim.x * 256;
    unsigned done = 0;
    if ((((unsigned long long)(size_t)d | (unsigned long long)(size_t)s) & 15ull) == 0ull) {  // (uniform)
        const unsigned quads = bytes >> 4;
        for (unsigned i = t; i < quads; i += T) ((uint4 *)d)[i] = ((const uint4 *)s)[i];
        done = quads << 4;
    } else if ((((unsigned long long)(size_t)d | (unsigned long long)(size_t)s) & 3ull) == 0ull) {
        const unsigned words = bytes >> 2;
        for (unsigned i = t; i < words; i += T) ((unsigned *)d)[i] = ((const unsigned *)s)[i];
        done = words << 2;
    }
    for (unsigned i = done + t; i < bytes; i += T) d[i] = s[i];
}

}  // namespace

int ft_launch_gather_batch(hipStream_t st, const FtGatherRec *recs, int nRecs) {
    if (nRecs <= 0) return FT_OK;
    hipLaunchKernelGGL(k_gather_batch, dim3(2, nRecs), dim3(256), 0, st, recs);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_deliver_blocks(hipStream_t st, void *d0, const void *s0, size_t bytes0, void *d1, const void *s1, size_t bytes1,
                             void *d2, const void *s2, size_t bytes2) {
    FtBlocks b;
    b.dst[0] = d0; b.src[0] = s0; b.words[0] = (int)((bytes0 + 3) / 4);
    b.dst[1] = d1; b.src[1] = s1; b.words[1] = (int)((bytes1 + 3) / 4);
    b.dst[2] = d2; b.src[2] = s2; b.words[2] = (int)((bytes2 + 3) / 4);
    const int total = b.words[0] + b.words[1] + b.words[2];
    if (total <= 0) return FT_OK;
    hipLaunchKernelGGL(k_deliver_blocks, dim3(std::max(1, std::min(64, (total + 1023) / 1024))), dim3(256), 0, st, b);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_fill_stride_u64(hipStream_t st, unsigned long long *p, int n, int strideWords, unsigned long long v) {
    if (n <= 0) return FT_OK;
    hipLaunchKernelGGL(k_fill_stride_u64, dim3((n + 255) / 256), dim3(256), 0, st, p, n, strideWords, v);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_fill_claims(hipStream_t st, int *p, int n, unsigned long long *meta, int nMeta, int strideWords) {
    const int m = std::max(n, meta ? nMeta : 0);
    if (m <= 0) return FT_OK;
    hipLaunchKernelGGL(k_fill_claims, dim3((m + 255) / 256), dim3(256), 0, st, p, n, meta, meta ? nMeta : 0, strideWords);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_fill_i32(hipStream_t st, int *p, int n, int v) {
    if (n <= 0) return FT_OK;
    hipLaunchKernelGGL(k_fill_i32, dim3((n + 255) / 256), dim3(256), 0, st, p, n, v);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_frustum(hipStream_t st, const FtDevFrame &F, const FtFrustumPose &T, const FtDevMapPoints &P,
                      float viewingCosLimit, float logScaleFactor, int farPoints, float thFar, const FtFrustumOut &O) {
    int rc = ft_launch_fill_i32(st, O.count, 1, 0);
    if (rc != FT_OK) return rc;
    if (P.M <= 0) return FT_OK;
    hipLaunchKernelGGL(k_frustum, dim3((P.M + 255) / 256), dim3(256), 0, st, F, T, P, viewingCosLimit, logScaleFactor, farPoints,
                       thFar, O);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_features_in_area(hipStream_t st, const FtDevFrame &F, int nq, const float *qx, const float *qy, const float *qr,
                               const int *qmin, const int *qmax, const uint8_t *qright, const int *offsets,
                               unsigned *outKeys, int *outCount) {
    if (nq <= 0) return FT_OK;
    hipLaunchKernelGGL(k_features_in_area, dim3((nq + 3) / 4), dim3(256), 0, st, F, nq, qx, qy, qr, qmin, qmax, qright, offsets,
                       outKeys, outCount);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_build_grid(hipStream_t st, const FtDevFrame &F, int *gridStartL, int *gridStartR, float4 *recL, uint8_t *descL,
                         float4 *recR, uint8_t *descR) {
    hipLaunchKernelGGL(k_build_grid, dim3(F.nlevels, gridStartR ? 2 : 1), dim3(256), 0, st, F, gridStartL, gridStartR, recL, descL, recR,
                       descR);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_search_local(hipStream_t st, const FtDevFrame &F, const FtDevLocalPoints &P, const FtClaims &C, float th,
                           float nnRatio, int *res, const FtLocalRaw &raw) {
    if (P.M <= 0) return FT_OK;
    hipLaunchKernelGGL(k_search_local, dim3((P.M + FT_SEARCH_WPB - 1) / FT_SEARCH_WPB), dim3(64 * FT_SEARCH_WPB), 0, st, F, P, C, th, nnRatio, res, raw);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_search_last(hipStream_t st, const FtDevFrame &F, const FtDevLastPoints &L, const FtClaims &C,
                          const FtPose &Tcw, float th, int forward, int backward, int *res, const FtLastRaw &raw) {
    if (L.N <= 0) return FT_OK;
    hipLaunchKernelGGL(k_search_last, dim3((L.N + FT_SEARCH_WPB - 1) / FT_SEARCH_WPB), dim3(64 * FT_SEARCH_WPB), 0, st, F, L, C, Tcw, th, forward, backward, res, raw);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

// ---- launches of a batch of frames (ft_tracked_batch, search.cpp) ----
int ft_launch_fill_claims_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxWords);
#define FT_SLOW_BLOCKS 16  // workgroups per frame of a slow-list launch (a grid-stride loop serves longer lists)
static Rebase rebase_of(void *arena) { return Rebase{(uint8_t *)arena, (unsigned long long)(uintptr_t)arena}; }

int ft_launch_build_grid_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxLevels, bool twoCam) {
    if (nFrames <= 0) return FT_OK;
    hipLaunchKernelGGL(k_build_grid_batch, dim3(maxLevels, twoCam ? 2 : 1, nFrames), dim3(256), 0, st, jobs, rebase_of(arena));
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_frustum_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxM, float viewingCosLimit, float logScaleFactor,
                            int farPoints, float thFar) {
    if (nFrames <= 0 || maxM <= 0) return FT_OK;
    hipLaunchKernelGGL(k_frustum_batch, dim3((maxM + 255) / 256, nFrames), dim3(256), 0, st, jobs, rebase_of(arena), viewingCosLimit,
                       logScaleFactor, farPoints, thFar);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

// everything behind the first pass of a batch in one launch (k_resolve_batch): a workgroup per frame
int ft_launch_resolve_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int local, float nnRatio, int sharedInts) {
    if (nFrames <= 0) return FT_OK;
    const size_t sh = sizeof(int) * (size_t)sharedInts;  // the last-writer table of the largest frame; 0: frames beyond the LDS
    if (sharedInts > 0) {
        if (local) hipLaunchKernelGGL((k_resolve_batch<true, true>), dim3(nFrames), dim3(FT_RS_LANES), sh, st, jobs, rebase_of(arena), nnRatio);
        else hipLaunchKernelGGL((k_resolve_batch<false, true>), dim3(nFrames), dim3(FT_RS_LANES), sh, st, jobs, rebase_of(arena), nnRatio);
    } else {
        if (local) hipLaunchKernelGGL((k_resolve_batch<true, false>), dim3(nFrames), dim3(FT_RS_LANES), 0, st, jobs, rebase_of(arena), nnRatio);
        else hipLaunchKernelGGL((k_resolve_batch<false, false>), dim3(nFrames), dim3(FT_RS_LANES), 0, st, jobs, rebase_of(arena), nnRatio);
    }
    FT_HIP(hipGetLastError());
    return FT_OK;
}
int ft_launch_replay_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int local, int parity, int checkOrientation,
                           int sharedInts, int flagPos) {
    if (nFrames <= 0) return FT_OK;
    const Rebase rb = rebase_of(arena);
    const size_t sh = sizeof(int) * (size_t)sharedInts;
    if (sharedInts > 0) {
        if (local) hipLaunchKernelGGL((k_replay_batch<true, true>), dim3(nFrames), dim3(256), sh, st, jobs, rb, parity, checkOrientation, flagPos);
        else hipLaunchKernelGGL((k_replay_batch<false, true>), dim3(nFrames), dim3(256), sh, st, jobs, rb, parity, checkOrientation, flagPos);
    } else {
        if (local) hipLaunchKernelGGL((k_replay_batch<true, false>), dim3(nFrames), dim3(256), 0, st, jobs, rb, parity, checkOrientation, flagPos);
        else hipLaunchKernelGGL((k_replay_batch<false, false>), dim3(nFrames), dim3(256), 0, st, jobs, rb, parity, checkOrientation, flagPos);
    }
    FT_HIP(hipGetLastError());
    return FT_OK;
}
// the first pass with four points per wave (k_search_*_first): needs the candidate cache and the grid of every frame
int ft_launch_search_last_first(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxPoints, float th) {
    if (nFrames <= 0 || maxPoints <= 0) return FT_OK;
    hipLaunchKernelGGL(k_last_project_batch, dim3((maxPoints + 255) / 256, nFrames), dim3(256), 0, st, jobs, rebase_of(arena));
    dim3 grid;
    const FtSlotGrid sg = ft_slot_grid((maxPoints + 15) / 16, nFrames, grid);  // (a frame's workgroups on one XCD: its grid and its lists stay in that L2)
    hipLaunchKernelGGL(k_search_last_first, grid, dim3(256), 0, st, jobs, rebase_of(arena), th, sg);
    FT_HIP(hipGetLastError());
    return FT_OK;
}
int ft_launch_search_local_first(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxPoints, float th, float nnRatio) {
    if (nFrames <= 0 || maxPoints <= 0) return FT_OK;
    dim3 grid;
    const FtSlotGrid sg = ft_slot_grid((maxPoints + 15) / 16, nFrames, grid);
    hipLaunchKernelGGL(k_search_local_first, grid, dim3(256), 0, st, jobs, rebase_of(arena), th, nnRatio, sg);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_search_last_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxPoints, int pass, int fCur, int fPrev,
                                int fReset, float th) {
    if (nFrames <= 0 || maxPoints <= 0) return FT_OK;
    if (pass == 0)  // the points' projections, once (J.proj; the later passes and the slow lists read them too)
        hipLaunchKernelGGL(k_last_project_batch, dim3((maxPoints + 255) / 256, nFrames), dim3(256), 0, st, jobs, rebase_of(arena));
    hipLaunchKernelGGL(k_search_last_batch<false>, dim3((maxPoints + FT_SEARCH_WPB - 1) / FT_SEARCH_WPB, nFrames), dim3(64 * FT_SEARCH_WPB), 0, st,
                       jobs, rebase_of(arena), pass, fCur, fPrev, fReset, th);
    FT_HIP(hipGetLastError());
    return FT_OK;
}
// a later pass: the lean kernel for the points the candidate cache serves, the general kernel for its slow list
int ft_launch_search_last_batch_lean(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxPoints, int pass, int fCur,
                                     int fPrev, int fReset, float th) {
    if (nFrames <= 0 || maxPoints <= 0) return FT_OK;
    hipLaunchKernelGGL(k_search_last_lean, dim3((maxPoints + FT_LEAN_PPB - 1) / FT_LEAN_PPB, nFrames), dim3(256), 0, st, jobs,
                       rebase_of(arena), pass, fCur, fPrev, fReset);
    hipLaunchKernelGGL(k_search_last_batch<true>, dim3(FT_SLOW_BLOCKS, nFrames), dim3(64 * FT_SEARCH_WPB), 0, st, jobs, rebase_of(arena), pass,
                       fCur, fPrev, fReset, th);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_search_local_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxPoints, int pass, int fCur, int fPrev,
                                 int fReset, float th, float nnRatio) {
    if (nFrames <= 0 || maxPoints <= 0) return FT_OK;
    hipLaunchKernelGGL(k_search_local_batch<false>, dim3((maxPoints + FT_SEARCH_WPB - 1) / FT_SEARCH_WPB, nFrames), dim3(64 * FT_SEARCH_WPB), 0,
                       st, jobs, rebase_of(arena), pass, fCur, fPrev, fReset, th, nnRatio);
    FT_HIP(hipGetLastError());
    return FT_OK;
}
int ft_launch_search_local_batch_lean(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxPoints, int pass, int fCur,
                                      int fPrev, int fReset, float th, float nnRatio) {
    if (nFrames <= 0 || maxPoints <= 0) return FT_OK;
    hipLaunchKernelGGL(k_search_local_lean, dim3((maxPoints + FT_LEAN_PPB - 1) / FT_LEAN_PPB, nFrames), dim3(256), 0, st, jobs,
                       rebase_of(arena), pass, fCur, fPrev, fReset, nnRatio);
    hipLaunchKernelGGL(k_search_local_batch<true>, dim3(FT_SLOW_BLOCKS, nFrames), dim3(64 * FT_SEARCH_WPB), 0, st, jobs, rebase_of(arena), pass,
                       fCur, fPrev, fReset, th, nnRatio);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_fill_claims_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxWords) {
    if (nFrames <= 0) return FT_OK;
    hipLaunchKernelGGL(k_fill_claims_batch, dim3(std::max(1, std::min(64, (maxWords + 1023) / 1024)), nFrames), dim3(256), 0, st, jobs,
                       rebase_of(arena));
    FT_HIP(hipGetLastError());
    return FT_OK;
}

#ifndef FT_DELIVER_BLOCKS
#define FT_DELIVER_BLOCKS 2  // workgroups per record
#endif
int ft_launch_deliver_batch(hipStream_t st, const FtDeliverRec *recs, int nRecs, int maxWords, int parity) {
    if (nRecs <= 0 || maxWords <= 0) return FT_OK;
    hipLaunchKernelGGL(k_deliver_batch, dim3(std::max(1, std::min(FT_DELIVER_BLOCKS, (maxWords + 1023) / 1024)), nRecs), dim3(256), 0, st, recs, parity);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_bind_fisheye_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxKp, const FtBindArgs &A) {
    if (nFrames <= 0) return FT_OK;
    hipLaunchKernelGGL(k_lap_gather_batch, dim3(2, nFrames), dim3(256), 0, st, jobs, rebase_of(arena), A);
    hipLaunchKernelGGL(k_fisheye_2nn_batch, dim3((maxKp + 4 * FE_Q - 1) / (4 * FE_Q), nFrames), dim3(256), 0, st, jobs, rebase_of(arena),
                       (const int *)A.mono, A.triangulate ? 0 : 1);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_cache_partition_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxPoints) {
    if (nFrames <= 0 || maxPoints <= 0) return FT_OK;
    hipLaunchKernelGGL(k_cache_partition_batch, dim3((2 * maxPoints + 4 * FT_PART_LISTS - 1) / (4 * FT_PART_LISTS), nFrames), dim3(256), 0, st, jobs,
                       rebase_of(arena));
    FT_HIP(hipGetLastError());
    return FT_OK;
}
