// Device octree keypoint distribution (ORBextractor::DistributeOctTree, reference
// src/ORBextractor.cc:660-884) for gfx950: one wave per (level, image), everything in LDS.
//
//   1. path code per candidate (octree_paths.h), lane-parallel
//   2. bitonic sort of (code, index) keys: afterwards every node of the tree owns a contiguous key range
//   3. ROUNDS (octree.cpp distribute_octree_rounds is the single-thread statement of the same thing, checked
//      against the oracle on the CPU): the node list is an array; a whole pass of the reference - all nodes
//      of a breadth-first pass, or the leading nodes of the size-sorted careful pass - is split at once, one
//      lane per node (three boundary searches in the node's key range), and the new list positions are
//      prefix sums: children of the processed nodes in reverse push order, then the untouched nodes.
//      Only the careful pass's std::sort is sequential: its tie order is the reference's, so it is replayed
//      literally on lane 0 (octree_paths.h std_sort_replay).
//   4. per retained node the first-maximum response pick, one lane per node.
//
// The result - retained candidates AND their order - equals the host octree (octree.cpp).  Two tiers: k_octree sorts up to
// FT_OCT_MAXN = 4 096 candidates of a level in LDS; a level with more goes on a list for k_octree_big (the same code with
// an LDS layout for up to 16 384 keys).  A level beyond that - or a listed one the second tier has no workgroup for - raises
// the image's overflow flag, and the host redoes that image with the host octree (extractor.cpp ft_extract_repair_*).
#include "ft_internal.h"
#include "octree_paths.h"

namespace {

using ft::op::kMaxDepth;
using ft::op::Roots;
using ft::op::SortElem;
using ft::op::SortFrame;

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// exclusive prefix sum over the wave; total = sum of all lanes
__device__ __forceinline__ int wave_excl_scan(int v, int lane, int &total) {
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    total = __shfl(x, 63);
    return x - v;
}

// LDS layout of one wave's workspace (offsets in bytes from the dynamic LDS base)
struct OctLds {
    int keys, lohi[2], x01[2], dep[2], vSize, vPrev, ord, bq, pq, mark, stack, total;
};

__host__ __device__ inline OctLds oct_lds_layout(int cap, int maxN) {
    OctLds o;
    int p = 0;
    auto take = [&](int bytes) {
        const int at = p;
        p += (bytes + 15) & ~15;
        return at;
    };
    o.keys = take(maxN * 8);
    for (int i = 0; i < 2; i++) {
        o.lohi[i] = take(2 * cap * 4);
        o.x01[i] = take(2 * cap * 4);
        o.dep[i] = take(2 * cap);
    }
    o.vSize = take(cap * 8);
    o.vPrev = take(cap * 8);
    o.ord = take(cap * 2);
    o.bq = take(cap * 8);  // b1, b2, b3 of a processed node (u16 x 3, one pad)
    o.pq = take(cap * 4);  // exclusive prefixes: children | children with more than one key << 16
    o.mark = take(2 * cap);
    o.stack = take(64 * sizeof(SortFrame));
    o.total = p;
    return o;
}

// std::sort(a, a + n, compareNodes) of libstdc++ with one wave: the steps of octree_paths.h
// std_sort_replay_steps (checked against libstdc++ on the CPU), each of them lane-parallel.
// The introsort partitions of a range of at most 64 elements with the elements held in registers (lane i = element
// f + i): median-of-three and the pivot come from v_readlane, the candidate sets of the closed-form partition are two
// ballots, the k-th candidate from the left meets the k-th from the right through two 64-entry tables in LDS, and the
// swaps are one ds_bpermute per dword; the range stack lives in the lanes of one register.  Only the final state goes
// back to LDS.  Same steps as ss_partition_pairs (octree_paths.h), i.e. the same array as libstdc++'s loop.
__device__ void wave_sort_window(SortElem *a, int f, int l, int depth0, uint8_t *tabA, uint8_t *tabB, int lane) {
    const int len = l - f;
    unsigned key = 0xffffffffu, val = 0;
    if (lane < len) {
        const SortElem e = a[f + lane];
        key = e.key;
        val = e.val;
    }
    const unsigned long long below = (1ull << lane) - 1ull, above = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);
    unsigned frames = 0;  // lane s holds frame s: lo | hi << 8 | depth << 16
    int sp = 0;
    if (lane == 0) frames = (unsigned)(0 | (len << 8) | (depth0 << 16));
    sp = 1;
    while (sp > 0) {
        const unsigned fr = __builtin_amdgcn_readlane(frames, --sp);
        int lo = (int)(fr & 0xffu), hi = (int)((fr >> 8) & 0xffu), depth = (int)(fr >> 16);
        while (hi - lo > 16) {
            if (depth == 0) {  // heap sort at the depth limit: literal replay on one lane through LDS
                if (lane < len) a[f + lane] = SortElem{key, val};
                wave_lds_sync();
                if (lane == 0) ft::op::ss_heap_sort(a + f + lo, a + f + hi);
                wave_lds_sync();
                if (lane < len) {
                    const SortElem e = a[f + lane];
                    key = e.key;
                    val = e.val;
                }
                break;
            }
            --depth;
            // __move_median_to_first(first, first + 1, mid, last - 1)
            const int ia = lo + 1, ib = lo + (hi - lo) / 2, ic = hi - 1;
            const unsigned ka = __builtin_amdgcn_readlane(key, ia), kb = __builtin_amdgcn_readlane(key, ib),
                           kc = __builtin_amdgcn_readlane(key, ic);
            int pick;
            if (ka < kb) pick = (kb < kc) ? ib : (ka < kc) ? ic : ia;
            else pick = (ka < kc) ? ia : (kb < kc) ? ic : ib;
            const unsigned pv = (pick == ia) ? ka : (pick == ib) ? kb : kc;
            {  // swap(first, pick)
                const unsigned k0 = __builtin_amdgcn_readlane(key, lo), v0 = __builtin_amdgcn_readlane(val, lo);
                const unsigned k1 = pv, v1 = __builtin_amdgcn_readlane(val, pick);
                if (lane == lo) { key = k1; val = v1; }
                if (lane == pick) { key = k0; val = v0; }
            }
            // __unguarded_partition(first + 1, last, first) in closed form
            const bool in = lane > lo && lane < hi;
            const bool isA = in && !(key < pv), isB = in && !(pv < key);
            const unsigned long long mA = __ballot(isA), mB = __ballot(isB);
            const int nA = __popcll(mA), nB = __popcll(mB);
            const int ra = __popcll(mA & below), rb = __popcll(mB & above);
            if (isA) tabA[ra] = (uint8_t)lane;  // k-th candidate from the left
            if (isB) tabB[rb] = (uint8_t)lane;  // k-th candidate from the right
            wave_lds_sync();
            int partner = lane;
            bool swA = false;
            if (isA && ra < nB) {
                const int pb = tabB[ra];
                if (lane < pb) { partner = pb; swA = true; }
            }
            if (!swA && isB && rb < nA) {
                const int pa = tabA[rb];
                if (pa < lane) partner = pa;
            }
            const int K = __popcll(__ballot(swA));
            const int bK = K ? (int)tabB[K - 1] : hi;
            const int aK = K < nA ? (int)tabA[K] : 0x7fffffff;
            const int cut = aK < bK ? aK : bK;
            key = (unsigned)__shfl((int)key, partner);
            val = (unsigned)__shfl((int)val, partner);
            wave_lds_sync();  // the tables are read before the next partition rewrites them
            if (sp < 64) {
                if (lane == sp) frames = (unsigned)(cut | (hi << 8) | (depth << 16));
                sp++;
            }
            hi = cut;
        }
    }
    if (lane < len) a[f + lane] = SortElem{key, val};
    wave_lds_sync();
}

__device__ void wave_std_sort(SortElem *a, int n, SortFrame *stack, uint16_t *posA, uint16_t *posB, SortElem *tmp, int lane) {
    if (n <= 1) return;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) lg++;
    int sp = 0;
    if (lane == 0) stack[0] = SortFrame{0, (uint16_t)n, (uint16_t)(2 * lg), 0};
    sp = 1;
    wave_lds_sync();
    const unsigned long long below = (1ull << lane) - 1;
    while (sp > 0) {
        const SortFrame fr = stack[--sp];
        int f = fr.f, l = fr.l, depth = fr.depth;
        wave_lds_sync();  // the frame is read before a later push overwrites the slot
        while (l - f > 16) {
            if (l - f <= 64) {  // the rest of this range's partitions run in registers
                wave_sort_window(a, f, l, depth, (uint8_t *)posA, (uint8_t *)posB, lane);
                break;
            }
            if (depth == 0) {
                if (lane == 0) ft::op::ss_heap_sort(a + f, a + l);
                wave_lds_sync();
                break;
            }
            --depth;
            // __move_median_to_first(first, first + 1, mid, last - 1)
            const int ia = f + 1, ib = f + (l - f) / 2, ic = l - 1;
            const uint32_t ka = a[ia].key, kb = a[ib].key, kc = a[ic].key;
            int pick;
            if (ka < kb) pick = (kb < kc) ? ib : (ka < kc) ? ic : ia;
            else pick = (ka < kc) ? ia : (kb < kc) ? ic : ib;
            const uint32_t pv = (pick == ia) ? ka : (pick == ib) ? kb : kc;
            if (lane == 0) {
                const SortElem t0 = a[f], t1 = a[pick];
                a[f] = t1;
                a[pick] = t0;
            }
            wave_lds_sync();
            // candidates: A ascending, B ascending (read back reversed)
            int nA = 0, nB = 0;
            for (int i0 = f + 1; i0 < l; i0 += 64) {
                const int i = i0 + lane;
                const bool valid = i < l;
                const uint32_t x = valid ? a[i].key : 0u;
                const bool isA = valid && !(x < pv), isB = valid && !(pv < x);
                const unsigned long long bA = __ballot(isA), bB = __ballot(isB);
                if (isA) posA[nA + __popcll(bA & below)] = (uint16_t)i;
                if (isB) posB[nB + __popcll(bB & below)] = (uint16_t)i;
                nA += __popcll(bA);
                nB += __popcll(bB);
            }
            wave_lds_sync();
            int K = 0;
            const int nPair = min(nA, nB);
            for (int k0 = 0; k0 < nPair; k0 += 64) {
                const int k = k0 + lane;
                bool sw = false;
                int pa = 0, pb = 0;
                if (k < nPair) {
                    pa = posA[k];
                    pb = posB[nB - 1 - k];
                    sw = pa < pb;
                }
                const unsigned long long bs = __ballot(sw);
                if (sw) {
                    const SortElem t0 = a[pa], t1 = a[pb];
                    a[pa] = t1;
                    a[pb] = t0;
                }
                K += __popcll(bs);
                if (bs != ~0ull) break;
            }
            const int bK = K ? (int)posB[nB - K] : l;
            const int aK = K < nA ? (int)posA[K] : 0x7fffffff;
            const int cut = aK < bK ? aK : bK;
            wave_lds_sync();
            if (sp < 64) {
                if (lane == 0) stack[sp] = SortFrame{(uint16_t)cut, (uint16_t)l, (uint16_t)depth, 0};
                sp++;
            }
            l = cut;
        }
        wave_lds_sync();
    }
    // __final_insertion_sort as a windowed stable rank
    for (int i = lane; i < n; i += 64) {
        const SortElem e = a[i];
        // the 32 neighbouring keys are requested together (clamped indices; out-of-range neighbours do not count)
        uint32_t kl[16], kr[16];
#pragma unroll
        for (int d = 0; d < 16; d++) {
            kl[d] = a[max(i - 1 - d, 0)].key;
            kr[d] = a[min(i + 1 + d, n - 1)].key;
        }
        int pos = i;
#pragma unroll
        for (int d = 0; d < 16; d++) {
            pos -= (i - 1 - d >= 0) && kl[d] > e.key;
            pos += (i + 1 + d <= n - 1) && kr[d] < e.key;
        }
        tmp[pos] = e;
    }
    wave_lds_sync();
    for (int i = lane; i < n; i += 64) a[i] = tmp[i];
    wave_lds_sync();
}

constexpr int OCT_THREADS = 512;  // codes + key sort use the whole block, the rounds only its first wave

// one (level, image) with at most maxN candidates (maxN: the key capacity of the LDS layout of this launch)
__device__ __forceinline__ void oct_level(const FtGeom &g, const FtOctArgs &a, const int slot, const int level, const int maxN,
                                          uint8_t *smem) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const FtLevelGeom &L = g.lv[level];
    const int n = a.candCount[slot * g.nlevels + level];
    const int N = a.quota[level];
    int *cntOut = a.selCount + slot * g.nlevels + level;
    if (n <= 0) {
        if (tid == 0) *cntOut = 0;
        return;
    }
    const unsigned wMagic = L.wCell > 1 ? 0xffffffffu / (unsigned)L.wCell + 1u : 0u;
    const unsigned hMagic = L.hCell > 1 ? 0xffffffffu / (unsigned)L.hCell + 1u : 0u;
    // the sort key holds a 28-bit path code (at most 16 root nodes) and 7-bit cell / in-cell coordinates
    const int minB = FT_EDGE_THRESHOLD - 3;
    const Roots R = ft::op::make_roots(minB, L.maxBX, minB, L.maxBY);
    const bool keyFits = L.nCols <= 128 && L.nRows <= 128 && L.wCell <= 128 && L.hCell <= 128 && R.nIni <= 15;
    if (n > maxN && n <= a.bigN && keyFits && maxN < a.bigN) {
        // more candidates than this launch sorts in LDS: the level goes on the list of k_octree_big, which runs behind this
        // kernel with room for bigN keys per workgroup
        if (tid == 0) {
            const int idx = atomicAdd(a.bigCount, 1);
            atomicMax(a.bigCount + 1, idx + 1);  // the host sizes the second tier of the following batches by this
            if (idx < a.bigCap) a.bigList[idx] = (unsigned)slot | ((unsigned)level << 16);
            else {
                *cntOut = 0;
                atomicOr(a.overflow, 1);
                a.ovSlot[slot] = 1;
            }
        }
        return;
    }
    if (n > maxN || !keyFits) {  // beyond the device formulation: this image is redone with the host octree
        if (tid == 0) {
            *cntOut = 0;
            atomicOr(a.overflow, 1);
            a.ovSlot[slot] = 1;
        }
        return;
    }
    const uint32_t *cand = a.cand + (size_t)slot * g.candPerSlot + L.candBase;
    const int cap = a.poolCap;
    const OctLds o = oct_lds_layout(cap, maxN);
    unsigned long long *keys = (unsigned long long *)(smem + o.keys);  // code << 32 | index
    const uint32_t *codes = (const uint32_t *)keys;                     // codes[2 * i + 1] = code << 4 | response >> 4
    // the two copies of the node list are addressed as smem + offset (never through an array of pointers: that loses
    // the LDS address space and turns every node access into a flat instruction)
    auto lohiOf = [&](int b) -> uint32_t * { return (uint32_t *)(smem + (b ? o.lohi[1] : o.lohi[0])); };
    auto x01Of = [&](int b) -> uint32_t * { return (uint32_t *)(smem + (b ? o.x01[1] : o.x01[0])); };
    auto depOf = [&](int b) -> uint8_t * { return smem + (b ? o.dep[1] : o.dep[0]); };
    SortElem *vSize = (SortElem *)(smem + o.vSize);
    SortElem *vPrev = (SortElem *)(smem + o.vPrev);
    uint16_t *ord = (uint16_t *)(smem + o.ord);
    uint16_t *bq = (uint16_t *)(smem + o.bq);
    uint32_t *pq = (uint32_t *)(smem + o.pq);
    uint8_t *mark = smem + o.mark;
    SortFrame *stack = (SortFrame *)(smem + o.stack);

    const bool prof = a.prof && slot == 0;
    unsigned long long tPrev = prof ? wall_clock64() : 0;
    int profIdx = 0;
    auto tick = [&]() {
        if (!prof) return;
        const unsigned long long t = wall_clock64();
        if (tid == 0 && profIdx < 8) a.prof[level * 8 + profIdx] += t - tPrev;
        profIdx++;
        tPrev = t;
    };
    // ---- 1. path codes ----
    int nPad = 64;
    while (nPad < n) nPad <<= 1;
    for (int i = tid; i < nPad; i += OCT_THREADS) {
        unsigned long long k = ~0ull;
        if (i < n) {
            const uint32_t c = cand[i];
            // key = path code << 36 | response << 28 | emission rank.  Codes are unique (a depth-12 path identifies
            // the pixel), so the low 36 bits never decide the order; they carry what the pick at the end needs: the
            // response and the candidate's rank in the reference's emission order - cell row, cell column, then
            // row-major inside the cell (ORBextractor.cc:1136-1199) - which is a function of its coordinates, so the
            // order in which the FAST stage delivers the candidates does not matter.
            const int x = (int)(c & 0xfffu), y = (int)((c >> 12) & 0xfffu);
            const int cj = min(wMagic ? (int)__umulhi((unsigned)(x - 3), wMagic) : x - 3, L.nCols - 1);
            const int ci = min(hMagic ? (int)__umulhi((unsigned)(y - 3), hMagic) : y - 3, L.nRows - 1);
            const unsigned ekey = ((unsigned)ci << 21) | ((unsigned)cj << 14) | ((unsigned)(y - 3 - ci * L.hCell) << 7) |
                                  (unsigned)(x - 3 - cj * L.wCell);
            k = ((unsigned long long)ft::op::path_code(R, x, y) << 36) | ((unsigned long long)(c >> 24) << 28) | ekey;
        }
        keys[i] = k;
    }
    __syncthreads();
    tick();  // 0: codes
    // ---- 2. bitonic sort (keys are unique, so the order equals a stable sort by code) ----
    for (int k = 2; k <= nPad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (nPad >> 1); t += OCT_THREADS) {
                const int i = ((t / j) * (j << 1)) + (t % j);  // j is a power of two: shifts / masks
                const int l = i + j;
                const unsigned long long x = keys[i], y = keys[l];
                const bool up = (i & k) == 0;
                if ((x > y) == up) {
                    keys[i] = y;
                    keys[l] = x;
                }
            }
            __syncthreads();
        }
    }
    tick();  // 1: sort
    if (tid >= 64) return;  // the rest is one wave: no block barrier below this line
    auto lower_bound = [&](int lo, int hi, uint32_t target) {
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (codes[2 * mid + 1] < (target << 4)) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    };
    // ---- 3. rounds ----
    int curB = 0, start = cap, m = 0;
    for (int s0 = 0; s0 < R.nIni; s0 += 64) {  // root nodes (:672-705), empty ones dropped
        const int s = s0 + lane;
        int lo = 0, hi = 0;
        if (s < R.nIni) {
            lo = lower_bound(0, n, (uint32_t)s << (2 * kMaxDepth));
            hi = lower_bound(lo, n, (uint32_t)(s + 1) << (2 * kMaxDepth));
        }
        const bool has = hi > lo;
        const unsigned long long bal = __ballot(has);
        if (has) {
            const int pos = start + m + __popcll(bal & ((1ull << lane) - 1));
            int x0, x1;
            ft::op::root_bounds(R, s, x0, x1);
            lohiOf(0)[pos] = (uint32_t)lo | ((uint32_t)hi << 16);
            x01Of(0)[pos] = (uint32_t)x0 | ((uint32_t)x1 << 16);
            depOf(0)[pos] = 0;
        }
        m += __popcll(bal);
    }
    wave_lds_sync();
    int nV = 0;
    // splits the nodes ord[0..nOrd) of the current list in that order (with useStop: until the list holds N)
    auto split_round = [&](int nOrd, bool useStop) {
        const uint32_t *cl = lohiOf(curB), *cx = x01Of(curB);
        const uint8_t *cd = depOf(curB);
        uint32_t *nl = lohiOf(curB ^ 1), *nx = x01Of(curB ^ 1);
        uint8_t *ndp = depOf(curB ^ 1);
        // pass A: boundaries, child counts, prefixes, stop
        int carryP = 0, carryQ = 0, cum = m, nproc = nOrd;
        for (int r0 = 0; r0 < nOrd; r0 += 64) {
            const int r = r0 + lane;
            int nch = 0, nbig = 0;
            if (r < nOrd) {
                const int t = ord[r];
                const uint32_t lh = cl[t];
                const int lo = (int)(lh & 0xffffu), hi = (int)(lh >> 16), d = cd[t];
                int b1 = hi, b2 = hi, b3 = hi;
                if (d < kMaxDepth) {
                    const int shift = 2 * (kMaxDepth - 1 - d);
                    const uint32_t prefix = codes[2 * lo + 1] >> (shift + 2 + 4);
                    b1 = lower_bound(lo, hi, ((prefix << 2) | 1u) << shift);
                    b2 = lower_bound(b1, hi, ((prefix << 2) | 2u) << shift);
                    b3 = lower_bound(b2, hi, ((prefix << 2) | 3u) << shift);
                }
                bq[4 * r] = (uint16_t)b1;
                bq[4 * r + 1] = (uint16_t)b2;
                bq[4 * r + 2] = (uint16_t)b3;
                nch = (b1 > lo) + (b2 > b1) + (b3 > b2) + (hi > b3);
                nbig = (b1 - lo > 1) + (b2 - b1 > 1) + (b3 - b2 > 1) + (hi - b3 > 1);
            }
            int totP, totQ;
            const int exP = wave_excl_scan(nch, lane, totP);
            const int exQ = wave_excl_scan(nbig, lane, totQ);
            if (r < nOrd) pq[r] = (uint32_t)(carryP + exP) | ((uint32_t)(carryQ + exQ) << 16);
            if (useStop) {
                // list size after processing ranks <= r
                const int cumR = cum + (exP + nch) - (lane + 1);
                const unsigned long long hit = __ballot(r < nOrd && cumR >= N);
                if (hit) {
                    const int f = __ffsll((long long)hit) - 1;
                    nproc = r0 + f + 1;
                    carryP += __shfl(exP + nch, f);
                    carryQ += __shfl(exQ + nbig, f);
                    break;
                }
            }
            const int cntHere = min(64, nOrd - r0);
            carryP += totP;
            carryQ += totQ;
            cum += totP - cntHere;
        }
        const int C = carryP;
        // marks
        for (int t = start + lane; t < start + m; t += 64) mark[t] = 0;
        wave_lds_sync();
        for (int r = lane; r < nproc; r += 64) mark[ord[r]] = 1;
        wave_lds_sync();
        // pass C: children of the processed nodes
        for (int r = lane; r < nproc; r += 64) {
            const int t = ord[r];
            const uint32_t lh = cl[t], xx = cx[t];
            const int d = cd[t];
            const int x0 = (int)(xx & 0xffffu), x1 = (int)(xx >> 16);
            const int mx = x0 + ((x1 - x0 + 1) >> 1);
            const int b[5] = {(int)(lh & 0xffffu), bq[4 * r], bq[4 * r + 1], bq[4 * r + 2], (int)(lh >> 16)};
            const uint32_t pp = pq[r];
            int k = (int)(pp & 0xffffu), kb = (int)(pp >> 16);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int cnt = b[c + 1] - b[c];
                if (cnt == 0) continue;
                const int pos = cap - 1 - k;
                const int cx0 = (c & 1) ? mx : x0, cx1 = (c & 1) ? x1 : mx;
                nl[pos] = (uint32_t)b[c] | ((uint32_t)b[c + 1] << 16);
                nx[pos] = (uint32_t)cx0 | ((uint32_t)cx1 << 16);
                ndp[pos] = (uint8_t)(d + 1);
                if (cnt > 1) {
                    SortElem e;
                    e.key = ((uint32_t)cnt << 16) | (uint32_t)cx0;
                    e.val = (uint32_t)pos;
                    vSize[kb] = e;
                    kb++;
                }
                k++;
            }
        }
        // pass D: untouched nodes keep their order behind the new ones
        int u = 0;
        for (int t0 = start; t0 < start + m; t0 += 64) {
            const int t = t0 + lane;
            const bool keep = t < start + m && !mark[t];
            const unsigned long long bal = __ballot(keep);
            if (keep) {
                const int pos = cap + u + __popcll(bal & ((1ull << lane) - 1));
                nl[pos] = cl[t];
                nx[pos] = cx[t];
                ndp[pos] = cd[t];
            }
            u += __popcll(bal);
        }
        wave_lds_sync();
        start = cap - C;
        m = C + u;
        nV = carryQ;
        curB ^= 1;
    };
    bool finish = false;
    while (!finish) {
        int prevSize = m;
        // breadth-first pass (:719-797): every node with more than one key is split, in list order
        int nOrd = 0;
        {
            const uint32_t *cl = lohiOf(curB);
            for (int t0 = start; t0 < start + m; t0 += 64) {
                const int t = t0 + lane;
                bool big = false;
                if (t < start + m) {
                    const uint32_t lh = cl[t];
                    big = (int)(lh >> 16) - (int)(lh & 0xffffu) > 1;
                }
                const unsigned long long bal = __ballot(big);
                if (big) ord[nOrd + __popcll(bal & ((1ull << lane) - 1))] = (uint16_t)t;
                nOrd += __popcll(bal);
            }
            wave_lds_sync();
        }
        split_round(nOrd, false);
        if (m >= N || m == prevSize) {
            finish = true;
        } else if (m + 3 * nV > N) {
            // careful phase (:799-875): largest nodes first, stop as soon as the list holds N
            while (!finish) {
                prevSize = m;
                const int nPrev = nV;
                for (int k = lane; k < nPrev; k += 64) vPrev[k] = vSize[k];
                wave_lds_sync();
                const unsigned long long ts = prof ? wall_clock64() : 0;
                wave_std_sort(vPrev, nPrev, stack, bq, bq + cap, vSize, lane);
                if (prof && lane == 0) {
                    a.prof[level * 8 + 4] += wall_clock64() - ts;
                    a.prof[level * 8 + 5] += (unsigned long long)nPrev;
                    a.prof[level * 8 + 6] += 1;
                }
                for (int r = lane; r < nPrev; r += 64) ord[r] = (uint16_t)vPrev[nPrev - 1 - r].val;
                wave_lds_sync();
                split_round(nPrev, true);
                if (m >= N || m == prevSize) finish = true;
            }
        }
    }
    tick();  // 2: rounds
    // ---- 4. per retained node: largest response, earliest original index on ties ----
    const int kept = min(m, a.levelMax[level]);
    const uint32_t *cl = lohiOf(curB);
    FtSelKp *out = a.sel + (size_t)slot * g.maxKp + a.selOff[level];
    for (int t = lane; t < kept; t += 64) {
        const unsigned lh = cl[start + t];
        const int lo = (int)(lh & 0xffffu), hi = (int)(lh >> 16);
        // maximise (response, -emission rank): the first maximum in emission order (:863-881)
        const unsigned long long low = (1ull << 36) - 1, inv = (1ull << 28) - 1;
        unsigned long long best = (keys[lo] & low) ^ inv;
        for (int k = lo + 1; k < hi; k++) best = max(best, (keys[k] & low) ^ inv);
        const unsigned ek = (unsigned)(~best) & (unsigned)inv;
        const int bx = (int)((ek >> 14) & 127u) * L.wCell + (int)(ek & 127u) + 3;
        const int by = (int)((ek >> 21) & 127u) * L.hCell + (int)((ek >> 7) & 127u) + 3;
        const unsigned bestC = ft_pack_cand(bx, by, (int)(best >> 28));
        FtSelKp s;
        s.x = (short)((bestC & 0xfffu) + minB);  // ORBextractor.cc:1211-1217: add the border offset back
        s.y = (short)(((bestC >> 12) & 0xfffu) + minB);
        s.level = (short)level;
        s.response = (short)(bestC >> 24);
        out[t] = s;
    }
    tick();  // 3: pick
    if (lane == 0) *cntOut = kept;
}

__global__ __launch_bounds__(OCT_THREADS) void k_octree(FtGeom g, FtOctArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // a thin kernel that runs next to wide ones: its single critical wave per level should win the issue arbitration
    __builtin_amdgcn_s_setprio(3);
    // workgroups are dispatched in the order of their linear index: image fastest, so the level-0 workgroups of all
    // images (the long ones: most candidates, largest quota) start first and the short high levels fill in behind
    oct_level(g, a, blockIdx.x, blockIdx.y, FT_OCT_MAXN, smem);
}

// The levels k_octree left on the list (more than FT_OCT_MAXN candidates): one workgroup per entry with an LDS layout
// for a.bigN keys (up to the whole 160 KB of a CU).  A workgroup that needs a CU's whole LDS waits for a CU to drain even
// if it has nothing to do, so the kernel is only launched while the stream of frames needs it, with a grid the host sizes
// from the demand of the previous batches (ft_extractor::bigGrid); small batches always get one workgroup per level.
__global__ __launch_bounds__(OCT_THREADS) void k_octree_big(FtGeom g, FtOctArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int cnt = min(*a.bigCount, a.bigCap);
    if ((int)blockIdx.x >= cnt) return;
    const unsigned e = a.bigList[blockIdx.x];
    oct_level(g, a, (int)(e & 0xffffu), (int)(e >> 16), a.bigN, smem);
}

}  // namespace

size_t ft_octree_smem_bytes(int poolCap) { return (size_t)oct_lds_layout(poolCap, FT_OCT_MAXN).total; }
// key capacity of the second-tier kernel: the largest power of two (the sort is bitonic) whose layout fits one CU's LDS
int ft_octree_big_keys(int poolCap) {
    for (int n = 16384; n > FT_OCT_MAXN; n >>= 1)
        if ((size_t)oct_lds_layout(poolCap, n).total <= 160 * 1024) return n;
    return 0;
}

int ft_launch_octree(hipStream_t st, const FtGeom &g, int batch, const FtOctArgs &a) {
    const size_t smem = ft_octree_smem_bytes(a.poolCap);
    if (smem > 64 * 1024)  // large quotas: raise the dynamic LDS limit (per device, so not cached in a static)
        FT_HIP(hipFuncSetAttribute((const void *)k_octree, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const size_t smemBig = a.bigN ? (size_t)oct_lds_layout(a.poolCap, a.bigN).total : 0;
    if (a.bigN) FT_HIP(hipFuncSetAttribute((const void *)k_octree_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smemBig));
    for (int rep = ft_debug_repeat("octree"); rep > 0; rep--) {
        if (a.bigN) FT_HIP(hipMemsetAsync(a.bigCount, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_octree, dim3(batch, g.nlevels), dim3(OCT_THREADS), smem, st, g, a);
        if (a.bigN && a.bigCap > 0) hipLaunchKernelGGL(k_octree_big, dim3(a.bigCap), dim3(OCT_THREADS), smemBig, st, g, a);
    }
    FT_HIP(hipGetLastError());
    return FT_OK;
}
