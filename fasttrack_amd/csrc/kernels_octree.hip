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
// The result - retained candidates AND their order - equals the host octree (octree.cpp).  Three tiers:
//   k_octree       sorts up to FT_OCT_MAXN = 4 096 candidates of a level in LDS (the formulation above);
//   k_octree_hist  a level with more goes on a list for the HISTOGRAM formulation (octree.cpp HistKeys is its host
//                  statement): nothing is sorted - the candidates are counted per tree node of depth D (nIni * 4^D <=
//                  FT_OCT_HIST_BINS bins in code order, LDS atomics by the whole workgroup), the exclusive prefix sums of
//                  the bins are the positions the sorted array would have, so a node's child boundaries are three table
//                  look-ups instead of three binary searches, and the pick is one more pass over the candidates (bin ->
//                  final node -> LDS atomic maximum of (response, -emission rank)).  Any number of candidates up to
//                  65 535 per level, 16 KB of bins instead of 8 bytes per key: two or three workgroups per CU where the
//                  sorted second tier took a whole CU for a 512-thread bitonic sort of 16 384 keys;
//   k_octree_big   a level whose tree grows deeper than D somewhere (the quota is spent on a few dense clusters) is
//                  handed on to the sorted formulation with an LDS layout for up to 16 384 keys.
// A level beyond all that - or a listed one the next tier has no workgroup for - raises the image's overflow flag, and
// the host redoes that image with the host octree (extractor.cpp ft_extract_repair_*).
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

// exclusive prefix sum over the wave; total = sum of all lanes.  Six DPP adds (shifts inside the rows of 16 lanes, then the
// row totals broadcast into the following rows) - as six __shfl_up it was six ds_bpermute round trips, and the rounds
// scan twice per 64 nodes
__device__ __forceinline__ int wave_excl_scan(int v, int lane, int &total) {
    (void)lane;
    int x = v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);  // row_shr:1 (lanes shifted in from outside the row read 0)
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, true);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, true);  // row_bcast:31 into rows 2 and 3
    total = __builtin_amdgcn_readlane(x, 63);
    return x - v;
}

// LDS layout of one workgroup's workspace (offsets in bytes from the dynamic LDS base)
struct OctLds {
    int keys, lohi[2], x01[2], dep[2], pre[2], vSize, vPrev, ord, bq, pq, mark, stack, best, misc, total;
    int low;  // compact layout only (else 0): the low key dwords during the pick, in the upper half of the key array
};

// keyBytes: the sorted tiers hold 8 bytes per key, the histogram tier two bytes per bin (+ the end entry).
// compact (first sorted tier): the rounds need the keys' high dwords alone (the path codes), so behind the sort the
// workgroup packs those into the lower half of the key array, parks the low dwords (response, emission rank: the pick's
// business) in global memory, and the arrays only the rounds use - vSize .. mark - live in the upper half; the pick brings
// the low dwords back over them.  49 instead of 64 KB per workgroup at the headline's quotas: THREE workgroups per CU, and a
// quarter less LDS withheld from the kernels that share the CUs while one wave per workgroup walks the rounds (the
// footprint is what the tier costs the pipeline: option oct_smem_pad, EXPERIMENTS.md section 3.1).  Falls back to the plain layout
// when the round arrays do not fit half the key array (quotas beyond ~500 per level).
__host__ __device__ inline OctLds oct_lds_layout(int cap, int keyBytes, bool hist, bool compact = false) {
    OctLds o;
    int p = 0;
    auto take = [&](int bytes) {
        const int at = p;
        p += (bytes + 15) & ~15;
        return at;
    };
    auto round_arrays = [&]() {
        o.vSize = take(cap * 8);
        o.vPrev = take(cap * 8);
        o.ord = take(cap * 2);
        o.bq = take(cap * 8);  // b1, b2, b3 of a processed node (u16 x 3, one pad)
        o.pq = take(cap * 4);  // exclusive prefixes: children | children with more than one key << 16
        o.mark = take(2 * cap);
    };
    o.keys = take(keyBytes);
    o.low = 0;
    if (compact && !hist) {
        const int half = (keyBytes / 2 + 15) & ~15, end = p;
        p = o.keys + half;
        round_arrays();
        if (p <= end) o.low = o.keys + half;
        p = end;
    }
    for (int i = 0; i < 2; i++) {
        o.lohi[i] = take(2 * cap * 4);
        o.x01[i] = take(2 * cap * 4);
        o.dep[i] = take(2 * cap);
        o.pre[i] = hist ? take(2 * cap * 4) : 0;  // code prefix of a node (the sorted tiers read it from the node's first key)
    }
    if (!o.low) round_arrays();
    o.stack = take(64 * sizeof(SortFrame));
    // per final node (at most levelMax <= N + 3 < cap of them): the (response, -emission rank) maximum.  Needed only after the
    // rounds, so it takes the place of vSize (cap * 8 bytes, used by the rounds alone): 51 KB instead of 58 - THREE workgroups
    // of the histogram tier share a CU's 160 KB
    o.best = o.vSize;
    o.misc = take(64);                      // hand-over between the round wave and the workgroup; scan carries
    o.total = p;
    return o;
}
__host__ __device__ inline int oct_hist_key_bytes() { return 2 * FT_OCT_HIST_BINS + 16; }

// std::sort(a, a + n, compareNodes) of libstdc++ with one wave: the steps of octree_paths.h
// std_sort_replay_steps (checked against libstdc++ on the CPU), each of them lane-parallel.
// The introsort partitions of a range of at most 64 elements with the elements held in registers (lane i = element
// f + i): median-of-three and the pivot come from v_readlane, the candidate sets of the closed-form partition are two
// ballots, the k-th candidate from the left meets the k-th from the right through two 64-entry tables in LDS, and the
// swaps are one ds_bpermute per dword; the range stack lives in the lanes of one register.  Only the final state goes
// back to LDS.  Same steps as ss_partition_pairs (octree_paths.h), i.e. the same array as libstdc++'s loop.
__device__ __forceinline__ void wave_sort_window(SortElem *a, int f, int l, int depth0, uint8_t *tabA, uint8_t *tabB, int lane) {
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

__device__ __forceinline__ void wave_std_sort(SortElem *a, int n, SortFrame *stack, uint16_t *posA, uint16_t *posB, SortElem *tmp, int lane) {
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

constexpr int OCT_THREADS = 512;  // codes + key sort / histogram / pick use the whole block, the rounds only its first wave
#ifndef OCT_SORT_STAGES
#define OCT_SORT_STAGES 2  // bitonic stages per LDS round trip (2^S keys per thread)
#endif
static_assert(OCT_SORT_STAGES >= 1 && OCT_SORT_STAGES <= 4, "bitonic_pass is instantiated for 1 .. 4 stages");

// S consecutive stages (partner distances 2^p .. 2^(p-S+1)) of the bitonic merge of size k over keys[0, nPad), by the whole
// workgroup; ends with the block barrier
template <int S>
__device__ __forceinline__ void bitonic_pass(unsigned long long *keys, int nPad, int k, int p, int tid) {
    constexpr int E = 1 << S;
    const int q = p - S + 1;  // lowest distance bit
    for (int t = tid; t < (nPad >> S); t += OCT_THREADS) {
        const int i = ((t >> q) << (p + 1)) + (t & ((1 << q) - 1));  // bits q .. p of the index are zero
        unsigned long long e[E];
#pragma unroll
        for (int r = 0; r < E; r++) e[r] = keys[i + (r << q)];
        const bool up = (i & k) == 0;
#pragma unroll
        for (int d = E >> 1; d > 0; d >>= 1) {
#pragma unroll
            for (int r = 0; r < E; r++) {
                if (r & d) continue;
                const unsigned long long x = e[r], y = e[r | d];
                const unsigned long long lo = x < y ? x : y, hi = x < y ? y : x;
                e[r] = up ? lo : hi;
                e[r | d] = up ? hi : lo;
            }
        }
#pragma unroll
        for (int r = 0; r < E; r++) keys[i + (r << q)] = e[r];
    }
    __syncthreads();
}

// hand-over words in OctLds::misc (ints)
enum { OM_START = 0, OM_M = 1, OM_CURB = 2, OM_GAVEUP = 3, OM_WAVESUM = 4 /* .. +8 */ };

// one (level, image).  Sorted tiers (HIST = false): at most maxN candidates (the key capacity of the LDS layout of this
// launch).  Histogram tier (HIST = true): any number up to 65 535.  Returns false when the histogram tier gave up on the
// level (a node deeper than its table would have to be split); nothing has been written then.
template <bool HIST>
__device__ __forceinline__ bool oct_level(const FtGeom &g, const FtOctArgs &a, const int slot, const int level, const int maxN,
                                          uint8_t *smem, uint32_t *lowG = nullptr) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const FtLevelGeom &L = g.lv[level];
    const int n = a.candCount[slot * g.nlevels + level];
    const int N = a.quota[level];
    int *cntOut = a.selCount + slot * g.nlevels + level;
    if (n <= 0) {
        if (tid == 0) *cntOut = 0;
        return true;
    }
    const unsigned wMagic = L.wCell > 1 ? 0xffffffffu / (unsigned)L.wCell + 1u : 0u;
    const unsigned hMagic = L.hCell > 1 ? 0xffffffffu / (unsigned)L.hCell + 1u : 0u;
    // the sort key holds a 28-bit path code (at most 16 root nodes) and 7-bit cell / in-cell coordinates
    const int minB = FT_EDGE_THRESHOLD - 3;
    const Roots R = ft::op::make_roots(minB, L.maxBX, minB, L.maxBY);
    const bool keyFits = L.nCols <= 128 && L.nRows <= 128 && L.wCell <= 128 && L.hCell <= 128 && R.nIni <= 15;
    auto fail = [&]() {  // beyond the device formulation: this image is redone with the host octree
        if (tid == 0) {
            *cntOut = 0;
            atomicOr(a.overflow, 1);
            a.ovSlot[slot] = 1;
        }
    };
    // files the level under the next tier (count at cnt[0], demand since the host last looked at cnt[2]); false = no room
    auto hand_on = [&](int *cnt, unsigned *list, int cap) -> bool {
        const int idx = atomicAdd(cnt, 1);
        atomicMax(cnt + 2, idx + 1);
        if (idx >= cap) return false;
        list[idx] = (unsigned)slot | ((unsigned)level << 16);
        return true;
    };
    if constexpr (!HIST) {
        if (n > maxN && keyFits && (a.histCap > 0 || a.histWanted || maxN < a.bigN)) {
            // more candidates than this launch sorts in LDS: the level goes on the list of the histogram tier (or, with that
            // tier switched off, of k_octree_big while it has room for the keys), which runs behind this kernel
            if (tid == 0) {
                bool ok;
                // (with the histogram tier retired for lack of demand - histCap 0, histWanted set - the level still counts as
                // demand, so that the host brings the tier back for the next batch; this one is repaired on the host)
                if ((a.histCap > 0 || a.histWanted) && n <= 65535) ok = hand_on(a.bigCount, a.bigList, a.histCap);
                else if (n <= a.bigN) ok = hand_on(a.bigCount + 1, a.sortList, a.sortCap);
                else ok = false;
                if (!ok) {
                    *cntOut = 0;
                    atomicOr(a.overflow, 1);
                    a.ovSlot[slot] = 1;
                }
            }
            return true;
        }
        if (n > maxN || !keyFits) {
            fail();
            return true;
        }
    } else {
        if (n > 65535 || !keyFits) {
            fail();
            return true;
        }
    }
    const uint32_t *cand = a.cand + (size_t)slot * g.candPerSlot + L.candBase;
    const int cap = a.poolCap;
    // lowG (first sorted tier): maxN dwords of global scratch of this (level, image) - the compact layout, if it fits
    const OctLds o = oct_lds_layout(cap, HIST ? oct_hist_key_bytes() : maxN * 8, HIST, lowG != nullptr);
    const bool compact = !HIST && o.low != 0;
    unsigned long long *keys = (unsigned long long *)(smem + o.keys);  // sorted tiers: code << 36 | response << 28 | rank
    // the high dword of key i = code << 4 | response >> 4: every second dword of the key array, or (compact layout, behind
    // the sort) dword i of its lower half
    const uint32_t *codes = (const uint32_t *)keys + (compact ? 0 : 1);
    const int codeShift = compact ? 0 : 1;
    uint32_t *bins32 = (uint32_t *)(smem + o.keys);                     // histogram tier: two 16-bit counters per dword,
    uint16_t *bins = (uint16_t *)(smem + o.keys);                       // then bins[b] = candidates in bins below b
    int *misc = (int *)(smem + o.misc);
    // the two copies of the node list are addressed as smem + offset (never through an array of pointers: that loses
    // the LDS address space and turns every node access into a flat instruction)
    auto lohiOf = [&](int b) -> uint32_t * { return (uint32_t *)(smem + (b ? o.lohi[1] : o.lohi[0])); };
    auto x01Of = [&](int b) -> uint32_t * { return (uint32_t *)(smem + (b ? o.x01[1] : o.x01[0])); };
    auto depOf = [&](int b) -> uint8_t * { return smem + (b ? o.dep[1] : o.dep[0]); };
    auto preOf = [&](int b) -> uint32_t * { return (uint32_t *)(smem + (b ? o.pre[1] : o.pre[0])); };
    SortElem *vSize = (SortElem *)(smem + o.vSize);
    SortElem *vPrev = (SortElem *)(smem + o.vPrev);
    uint16_t *ord = (uint16_t *)(smem + o.ord);
    uint16_t *bq = (uint16_t *)(smem + o.bq);
    uint32_t *pq = (uint32_t *)(smem + o.pq);
    uint8_t *mark = smem + o.mark;
    SortFrame *stack = (SortFrame *)(smem + o.stack);
    // histogram tier: depth of the table and its size
    const int D = HIST ? ft::op::hist_depth(R.nIni, FT_OCT_HIST_BINS) : 0;
    const int nBins = HIST ? R.nIni << (2 * D) : 0;
    const int perThread = HIST ? (nBins + OCT_THREADS - 1) / OCT_THREADS : 0;

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
    // emission rank of a candidate - cell row, cell column, then row-major inside the cell (ORBextractor.cc:1136-1199) - a
    // function of its coordinates, so the order in which the FAST stage delivers the candidates does not matter
    auto emission_key = [&](int x, int y) -> unsigned {
        const int cj = min(wMagic ? (int)__umulhi((unsigned)(x - 3), wMagic) : x - 3, L.nCols - 1);
        const int ci = min(hMagic ? (int)__umulhi((unsigned)(y - 3), hMagic) : y - 3, L.nRows - 1);
        return ((unsigned)ci << 21) | ((unsigned)cj << 14) | ((unsigned)(y - 3 - ci * L.hCell) << 7) | (unsigned)(x - 3 - cj * L.wCell);
    };
    if constexpr (!HIST) {
        // ---- 1. path codes ----
        int nPad = 64;
        while (nPad < n) nPad <<= 1;
        for (int i = tid; i < nPad; i += OCT_THREADS) {
            unsigned long long k = ~0ull;
            if (i < n) {
                const uint32_t c = cand[i];
                // key = path code << 36 | response << 28 | emission rank.  Codes are unique (a depth-12 path identifies
                // the pixel), so the low 36 bits never decide the order; they carry what the pick at the end needs: the
                // response and the candidate's rank in the reference's emission order.
                const int x = (int)(c & 0xfffu), y = (int)((c >> 12) & 0xfffu);
                k = ((unsigned long long)ft::op::path_code(R, x, y) << 36) | ((unsigned long long)(c >> 24) << 28) | emission_key(x, y);
            }
            keys[i] = k;
        }
        __syncthreads();
        tick();  // 0: codes
        // ---- 2. bitonic sort (keys are unique, so the order equals a stable sort by code) ----
        // (pair t of a stage with partner distance j = 2^sh: shifts and masks - written as t / j and t % j the compiler
        // emits two integer divisions per pair, which was a third of the sort's time.  Ending the stages that stay inside
        // one wave's run of keys with a wave-level fence instead of the block barrier - 60 of the 66 stages of 2 048 keys -
        // gains nothing on top: A/B 75.63 against 75.71 k frames/s.)
        // Up to OCT_SORT_STAGES stages per pass: a thread takes the 2^S keys whose indices differ in the bits of the S partner
        // distances 2^p .. 2^(p-S+1), runs those compare-exchange stages on them in registers and writes them back.  S = 2:
        // 36 passes (LDS round trip + block barrier each) instead of 66 for 2 048 keys, 16 instead of 23 us per level under
        // load; S = 3 / 4 (26 / 21 passes, but a half / a quarter of the threads with work and longer strides): 20 / 26 us.
        // The direction of a merge of size k depends on bit k of the index alone, which the keys of a thread share.
        {
            int m = 1;  // log2(k)
            for (int k = 2; k <= nPad; k <<= 1, m++) {
                int p = m - 1;  // the merge's stages: partner distances 2^p .. 2^0
                const int first = m % OCT_SORT_STAGES;  // the odd stages first, in a smaller pass
                if (first == 1) bitonic_pass<1>(keys, nPad, k, p, tid);
                else if (first == 2) bitonic_pass<2>(keys, nPad, k, p, tid);
                else if (first == 3) bitonic_pass<3>(keys, nPad, k, p, tid);
                p -= first;
                for (; p >= OCT_SORT_STAGES - 1; p -= OCT_SORT_STAGES) bitonic_pass<OCT_SORT_STAGES>(keys, nPad, k, p, tid);
            }
        }
        tick();  // 1: sort
        if (compact) {
            // high dwords to the lower half of the key array, low dwords to global memory: eight keys per thread and trip
            // are read before the first one is overwritten (chunk c's packed dwords land on keys of chunks <= c / 2)
            uint32_t *packed = (uint32_t *)keys;
            for (int base = 0; base < n; base += 8 * OCT_THREADS) {
                unsigned long long kk[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int i = base + j * OCT_THREADS + tid;
                    kk[j] = i < n ? keys[i] : 0ull;
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int i = base + j * OCT_THREADS + tid;
                    if (i < n) {
                        packed[i] = (uint32_t)(kk[j] >> 32);
                        lowG[i] = (uint32_t)kk[j];
                    }
                }
                __syncthreads();  // (also: the stores to lowG are done - the pick's wave reads them back)
            }
        }
        if (tid >= 64) return true;  // the rest is one wave: no block barrier below this line
    } else {
        // ---- 1. histogram over the nodes of depth D: bins in code order ----
        for (int i = tid; i <= (nBins >> 1); i += OCT_THREADS) bins32[i] = 0;
        if (tid == 0) misc[OM_GAVEUP] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += OCT_THREADS) {
            const uint32_t c = cand[i];
            const unsigned b = ft::op::path_prefix(R, (int)(c & 0xfffu), (int)((c >> 12) & 0xfffu), D);
            atomicAdd(&bins32[b >> 1], 1u << ((b & 1u) * 16u));  // a half never carries: n <= 65 535
        }
        __syncthreads();
        tick();  // 0: histogram
        // ---- 2. exclusive prefix sums in place: bins[b] = position of the bin's first key in the (virtual) sorted array ----
        {
            const int b0 = min(tid * perThread, nBins), b1 = min(b0 + perThread, nBins);
            int sum = 0;
            for (int b = b0; b < b1; b++) sum += bins[b];
            int tot;
            int ex = wave_excl_scan(sum, lane, tot);
            if (lane == 63) misc[OM_WAVESUM + (tid >> 6)] = tot;
            __syncthreads();
            for (int w = 0; w < (tid >> 6); w++) ex += misc[OM_WAVESUM + w];
            for (int b = b0; b < b1; b++) {
                const int c = bins[b];
                bins[b] = (uint16_t)ex;
                ex += c;
            }
            if (tid == OCT_THREADS - 1) bins[nBins] = (uint16_t)n;
            __syncthreads();
        }
        tick();  // 1: scan
    }
    int curB = 0, start = cap, m = 0;
    bool gaveUp = false;
    if (tid < 64) {
    auto lower_bound = [&](int lo, int hi, uint32_t target) {
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (codes[mid << codeShift] < (target << 4)) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    };
    // ---- 3. rounds ----
    for (int s0 = 0; s0 < R.nIni; s0 += 64) {  // root nodes (:672-705), empty ones dropped
        const int s = s0 + lane;
        int lo = 0, hi = 0;
        if (s < R.nIni) {
            if constexpr (HIST) {
                lo = bins[s << (2 * D)];
                hi = bins[(s + 1) << (2 * D)];
            } else {
                lo = lower_bound(0, n, (uint32_t)s << (2 * kMaxDepth));
                hi = lower_bound(lo, n, (uint32_t)(s + 1) << (2 * kMaxDepth));
            }
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
            if constexpr (HIST) preOf(0)[pos] = (uint32_t)s;
        }
        m += __popcll(bal);
    }
    wave_lds_sync();
    int nV = 0;
    // splits the nodes ord[0..nOrd) of the current list in that order (with useStop: until the list holds N)
    auto split_round = [&](int nOrd, bool useStop) {
        const uint32_t *cl = lohiOf(curB), *cx = x01Of(curB), *cp = preOf(curB);
        const uint8_t *cd = depOf(curB);
        uint32_t *nl = lohiOf(curB ^ 1), *nx = x01Of(curB ^ 1), *np = preOf(curB ^ 1);
        uint8_t *ndp = depOf(curB ^ 1);
        // pass A: boundaries, child counts, prefixes, stop
        int carryP = 0, carryQ = 0, cum = m, nproc = nOrd;
        for (int r0 = 0; r0 < nOrd; r0 += 64) {
            const int r = r0 + lane;
            int nch = 0, nbig = 0;
            bool deep = false;
            if (r < nOrd) {
                const int t = ord[r];
                const uint32_t lh = cl[t];
                const int lo = (int)(lh & 0xffffu), hi = (int)(lh >> 16), d = cd[t];
                int b1 = hi, b2 = hi, b3 = hi;
                if constexpr (HIST) {
                    if (d < D) {  // the children's first bins
                        const int sh = 2 * (D - 1 - d);
                        const uint32_t p4 = cp[t] << 2;
                        b1 = bins[(p4 | 1u) << sh];
                        b2 = bins[(p4 | 2u) << sh];
                        b3 = bins[(p4 | 3u) << sh];
                    } else {
                        deep = true;  // the table ends at depth D
                    }
                } else if (d < kMaxDepth) {
                    // the first keys of children 1, 2, 3: three binary searches over [lo, hi), advanced together - one
                    // LDS latency per step for all three instead of three dependent chains (b1 <= b2 <= b3 by the order
                    // of the targets; searching each in the whole range finds the same positions)
                    const int shift = 2 * (kMaxDepth - 1 - d);
                    const uint32_t prefix = codes[lo << codeShift] >> (shift + 2 + 4);
                    const uint32_t t1 = (((prefix << 2) | 1u) << shift) << 4, t2 = (((prefix << 2) | 2u) << shift) << 4,
                                   t3 = (((prefix << 2) | 3u) << shift) << 4;
                    int l1 = lo, h1 = hi, l2 = lo, h2 = hi, l3 = lo, h3 = hi;
                    while (l1 < h1 || l2 < h2 || l3 < h3) {
                        const int m1 = (l1 + h1) >> 1, m2 = (l2 + h2) >> 1, m3 = (l3 + h3) >> 1;
                        const uint32_t c1 = codes[min(m1, hi - 1) << codeShift], c2 = codes[min(m2, hi - 1) << codeShift],
                                       c3 = codes[min(m3, hi - 1) << codeShift];
                        if (l1 < h1) { if (c1 < t1) l1 = m1 + 1; else h1 = m1; }
                        if (l2 < h2) { if (c2 < t2) l2 = m2 + 1; else h2 = m2; }
                        if (l3 < h3) { if (c3 < t3) l3 = m3 + 1; else h3 = m3; }
                    }
                    b1 = l1;
                    b2 = l2;
                    b3 = l3;
                }
                bq[4 * r] = (uint16_t)b1;
                bq[4 * r + 1] = (uint16_t)b2;
                bq[4 * r + 2] = (uint16_t)b3;
                nch = (b1 > lo) + (b2 > b1) + (b3 > b2) + (hi > b3);
                nbig = (b1 - lo > 1) + (b2 - b1 > 1) + (b3 - b2 > 1) + (hi - b3 > 1);
            }
            if constexpr (HIST) {
                if (__any(deep)) {  // wave-uniform
                    gaveUp = true;
                    return;
                }
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
                    carryP += __builtin_amdgcn_readlane(exP + nch, f);  // (f is wave-uniform)
                    carryQ += __builtin_amdgcn_readlane(exQ + nbig, f);
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
            uint32_t p4 = 0;
            if constexpr (HIST) p4 = cp[t] << 2;
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
                if constexpr (HIST) np[pos] = p4 | (uint32_t)c;
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
                if constexpr (HIST) np[pos] = cp[t];
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
        if (gaveUp) break;
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
                if (gaveUp) break;
                if (m >= N || m == prevSize) finish = true;
            }
            if (gaveUp) break;
        }
    }
    if constexpr (HIST) {
        if (lane == 0) {
            misc[OM_START] = start;
            misc[OM_M] = m;
            misc[OM_CURB] = curB;
            misc[OM_GAVEUP] = gaveUp ? 1 : 0;
        }
    }
    }  // tid < 64
    tick();  // 2: rounds
    FtSelKp *out = a.sel + (size_t)slot * g.maxKp + a.selOff[level];
    const unsigned long long low = (1ull << 36) - 1, inv = (1ull << 28) - 1;
    auto write_out = [&](int t, unsigned long long best) {  // best = (response << 28 | rank) ^ inv: the first maximum in emission order
        const unsigned ek = (unsigned)(~best) & (unsigned)inv;
        const int bx = (int)((ek >> 14) & 127u) * L.wCell + (int)(ek & 127u) + 3;
        const int by = (int)((ek >> 21) & 127u) * L.hCell + (int)((ek >> 7) & 127u) + 3;
        FtSelKp s;
        s.x = (short)(bx + minB);  // ORBextractor.cc:1211-1217: add the border offset back
        s.y = (short)(by + minB);
        s.level = (short)level;
        s.response = (short)(best >> 28);
        out[t] = s;
    };
    if constexpr (!HIST) {
        // ---- 4. per retained node: largest response, earliest emission rank on ties (:863-881) ----
        const int kept = min(m, a.levelMax[level]);
        const uint32_t *cl = lohiOf(curB);
        if (compact) {
            // the low dwords come back from global memory into the upper half of the key array (the round arrays there are
            // dead); agent-scope loads: other waves of the workgroup wrote them
            uint32_t *lowL = (uint32_t *)(smem + o.low);
            wave_lds_sync();
            for (int base = 0; base < n; base += 8 * 64) {
                uint32_t v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int i = base + j * 64 + lane;
                    v[j] = i < n ? __hip_atomic_load(lowG + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int i = base + j * 64 + lane;
                    if (i < n) lowL[i] = v[j];
                }
            }
            wave_lds_sync();
            for (int t = lane; t < kept; t += 64) {
                const unsigned lh = cl[start + t];
                const int lo = (int)(lh & 0xffffu), hi = (int)(lh >> 16);
                unsigned long long best = 0;
                for (int k = lo; k < hi; k++)
                    best = max(best, (((unsigned long long)(codes[k] & 0xfu) << 32) | lowL[k]) ^ inv);
                write_out(t, best);
            }
        } else {
            for (int t = lane; t < kept; t += 64) {
                const unsigned lh = cl[start + t];
                const int lo = (int)(lh & 0xffffu), hi = (int)(lh >> 16);
                unsigned long long best = (keys[lo] & low) ^ inv;
                for (int k = lo + 1; k < hi; k++) best = max(best, (keys[k] & low) ^ inv);
                write_out(t, best);
            }
        }
        tick();  // 3: pick
        if (lane == 0) *cntOut = kept;
        return true;
    } else {
        __syncthreads();
        if (misc[OM_GAVEUP]) return false;  // uniform
        start = misc[OM_START];
        m = misc[OM_M];
        curB = misc[OM_CURB];
        const int kept = min(m, a.levelMax[level]);
        // ---- 4. bin -> final node.  Every node marks the first bin of its subtree with its list position + 1; a bin belongs to
        // the last mark at or below it (the final nodes tile the code space except where there are no candidates).  The table
        // of bin starts is not needed any more: the owners take its place.
        unsigned long long *best = (unsigned long long *)(smem + o.best);
        __syncthreads();  // every thread has read the hand-over words (the scan below reuses them)
        for (int b = tid; b < nBins; b += OCT_THREADS) bins[b] = 0;
        for (int t = tid; t < kept; t += OCT_THREADS) best[t] = 0;
        __syncthreads();
        {
            const uint32_t *cp = preOf(curB);
            const uint8_t *cd = depOf(curB);
            for (int t = tid; t < m; t += OCT_THREADS) bins[cp[start + t] << (2 * (D - (int)cd[start + t]))] = (uint16_t)(t + 1);
        }
        __syncthreads();
        {
            const int b0 = min(tid * perThread, nBins), b1 = min(b0 + perThread, nBins);
            int last = 0;
            for (int b = b0; b < b1; b++) {
                const int v = bins[b];
                last = v ? v : last;
            }
            // the last mark of the threads below: inside the wave by ballot, across the waves through LDS
            const unsigned long long has = __ballot(last != 0);
            const unsigned long long below = has & ((1ull << lane) - 1ull);
            const int src = below ? 63 - __clzll((long long)below) : 0;
            const int fromWave = __shfl(last, src);
            const int waveLast = __shfl(last, has ? 63 - __clzll((long long)has) : 0);
            if (lane == 0) misc[OM_WAVESUM + (tid >> 6)] = has ? waveLast : 0;
            __syncthreads();
            int carry = 0;
            for (int w = 0; w < (tid >> 6); w++) {
                const int v = misc[OM_WAVESUM + w];
                carry = v ? v : carry;
            }
            if (below) carry = fromWave;
            for (int b = b0; b < b1; b++) {
                const int v = bins[b];
                carry = v ? v : carry;
                bins[b] = (uint16_t)carry;
            }
        }
        __syncthreads();
        tick();  // 3: owners
        // ---- 5. pick: one pass over the candidates, LDS atomic maximum per final node ----
        for (int i = tid; i < n; i += OCT_THREADS) {
            const uint32_t c = cand[i];
            const int x = (int)(c & 0xfffu), y = (int)((c >> 12) & 0xfffu);
            const int t1 = bins[ft::op::path_prefix(R, x, y, D)];
            if (t1 >= 1 && t1 <= kept)
                atomicMax(&best[t1 - 1], ((((unsigned long long)(c >> 24)) << 28) | (unsigned long long)emission_key(x, y)) ^ inv);
        }
        __syncthreads();
        for (int t = tid; t < kept; t += OCT_THREADS) write_out(t, best[t]);
        tick();  // 4: pick
        if (tid == 0) *cntOut = kept;
        return true;
    }
}

__global__ __launch_bounds__(OCT_THREADS) void k_octree(FtGeom g, FtOctArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // a thin kernel that runs next to wide ones: its single critical wave per level should win the issue arbitration
    __builtin_amdgcn_s_setprio(3);
    // workgroups are dispatched in the order of their linear index: image fastest, so the level-0 workgroups of all
    // images (the long ones: most candidates, largest quota) start first and the short high levels fill in behind
    oct_level<false>(g, a, blockIdx.x, blockIdx.y, FT_OCT_MAXN, smem,
                     a.low ? a.low + ((size_t)blockIdx.x * g.nlevels + blockIdx.y) * FT_OCT_MAXN : nullptr);
}

// The levels k_octree left on the list (more than FT_OCT_MAXN candidates): histogram formulation.  The workgroups walk the
// list (entry blockIdx.x, + gridDim.x, ...): the grid is sized by the demand of the previous batches, not by the worst case
// (a grid of one workgroup per level of the launch cost the headline workload 5 % although none of them had an entry: 2 048
// workgroups of 58 KB LDS still have to be placed).  A level the formulation gives up on is handed on to k_octree_big (or
// flagged for the host when that tier has no room).
__global__ __launch_bounds__(OCT_THREADS) void k_octree_hist(FtGeom g, FtOctArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int cnt = min(a.bigCount[0], a.histCap);
    if ((int)blockIdx.x >= cnt) return;
    __builtin_amdgcn_s_setprio(3);
    for (int ei = (int)blockIdx.x; ei < cnt; ei += (int)gridDim.x) {
        const unsigned e = a.bigList[ei];
        const int slot = (int)(e & 0xffffu), level = (int)(e >> 16);
        const bool done = oct_level<true>(g, a, slot, level, 0, smem);
        if (!done && threadIdx.x == 0) {
            const int n = a.candCount[slot * g.nlevels + level];
            bool ok = false;
            if (n <= a.bigN) {
                const int idx = atomicAdd(a.bigCount + 1, 1);
                atomicMax(a.bigCount + 3, idx + 1);  // the host sizes the third tier of the following batches by this
                if (idx < a.sortCap) {
                    a.sortList[idx] = e;
                    ok = true;
                }
            }
            if (!ok) {
                a.selCount[slot * g.nlevels + level] = 0;
                atomicOr(a.overflow, 1);
                a.ovSlot[slot] = 1;
            }
        }
        __syncthreads();  // the next level reuses the LDS
    }
}

// Launches of a frame or two (latency mode; FtOctArgs::histFirst): one workgroup per (level, image) picks the formulation
// by the level's numbers.  Plenty of candidates for the quota (n >= 1.5 N) -> the histogram formulation: without the 512-
// thread bitonic sort (29 of the 102 us of a 1280x720 level 0) and with table look-ups instead of binary searches in the
// rounds the level - and with it the frame - is done sooner (extraction of a 1280x720 frame 0.269 -> 0.239 ms).  Fewer
// candidates than that and the tree is split until every node holds one key, far below the histogram's depth (measured on
// the test frames: every level with n < 1.3 N gives up, none above) -> the sorted formulation at once; it also takes over,
// in the same workgroup, a level the histogram gave up on after all (nothing has been written by then).  Only a level with
// more than FT_OCT_MAXN candidates AND a tree deeper than the table leaves the kernel: to k_octree_big, or to the host.
// (Launches that fill the chip keep the sorted tier first and the histogram tier behind it: there the two formulations
// cost the same, EXPERIMENTS.md section 3.5.)
__global__ __launch_bounds__(OCT_THREADS) void k_octree_auto(FtGeom g, FtOctArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __builtin_amdgcn_s_setprio(3);
    const int slot = (int)blockIdx.x, level = (int)blockIdx.y;
    const int n = a.candCount[slot * g.nlevels + level];
    const int N = a.quota[level];
    bool done = false;
    if (2 * n >= 3 * N || n > FT_OCT_MAXN) done = oct_level<true>(g, a, slot, level, 0, smem);  // wave- and block-uniform
    if (done) return;
    if (n <= FT_OCT_MAXN) {
        __syncthreads();  // the histogram formulation's LDS is dead
        // (n <= FT_OCT_MAXN: the sorted formulation never lists the level, whatever the list capacities say.  No modified copy
        // of the arguments: a 300-byte struct on the stack means scratch memory and flat accesses for the whole kernel)
        oct_level<false>(g, a, slot, level, FT_OCT_MAXN, smem, a.low ? a.low + ((size_t)slot * g.nlevels + level) * FT_OCT_MAXN : nullptr);
        return;
    }
    if (threadIdx.x == 0) {
        bool ok = false;
        if (n <= a.bigN) {
            const int idx = atomicAdd(a.bigCount + 1, 1);
            atomicMax(a.bigCount + 3, idx + 1);  // the host sizes the sorted big tier of the following frames by this
            if (idx < a.sortCap) {
                a.sortList[idx] = (unsigned)slot | ((unsigned)level << 16);
                ok = true;
            }
        }
        if (!ok) {
            a.selCount[slot * g.nlevels + level] = 0;
            atomicOr(a.overflow, 1);
            a.ovSlot[slot] = 1;
        }
    }
}

// The levels the histogram tier gave up on: the sorted formulation with an LDS layout for a.bigN keys (up to the whole
// 160 KB of a CU).  A workgroup that needs a CU's whole LDS waits for a CU to drain even if it has nothing to do, so the
// kernel is only launched while the stream of frames needs it, with a grid the host sizes from the demand of the previous
// batches (ft_extractor::bigGrid).
__global__ __launch_bounds__(OCT_THREADS) void k_octree_big(FtGeom g, FtOctArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int cnt = min(a.bigCount[1], a.sortCap);
    if ((int)blockIdx.x >= cnt) return;
    const unsigned e = a.sortList[blockIdx.x];
    oct_level<false>(g, a, (int)(e & 0xffffu), (int)(e >> 16), a.bigN, smem);
}

}  // namespace

size_t ft_octree_smem_bytes(int poolCap, bool compact) { return (size_t)oct_lds_layout(poolCap, FT_OCT_MAXN * 8, false, compact).total; }
size_t ft_octree_hist_smem_bytes(int poolCap) { return (size_t)oct_lds_layout(poolCap, oct_hist_key_bytes(), true).total; }
// key capacity of the sorted big tier: the largest power of two (the sort is bitonic) whose layout fits one CU's LDS
int ft_octree_big_keys(int poolCap) {
    for (int n = 16384; n > FT_OCT_MAXN; n >>= 1)
        if ((size_t)oct_lds_layout(poolCap, n * 8, false).total <= 160 * 1024) return n;
    return 0;
}

int ft_launch_octree(hipStream_t st, const FtGeom &g, int batch, const FtOctArgs &a) {
    // smemPad (option oct_smem_pad, a probe): extra bytes of LDS per workgroup of the sorted tier - what its footprint costs
    // the kernels that share the CUs with it
    const size_t smem = ft_octree_smem_bytes(a.poolCap, a.low != nullptr);
    if (ft_debug_env("FT_DEBUG_OCC")) fprintf(stderr, "[ft] k_octree: %zu B of LDS per workgroup (pool %d)\n", smem, a.poolCap);
    if (smem > 64 * 1024)  // large quotas: raise the dynamic LDS limit (per device, so not cached in a static)
        FT_HIP(hipFuncSetAttribute((const void *)k_octree, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const size_t smemHist = a.histCap > 0 ? ft_octree_hist_smem_bytes(a.poolCap) : 0;
    if (smemHist > 64 * 1024)
        FT_HIP(hipFuncSetAttribute((const void *)k_octree_hist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smemHist));
    const size_t smemBig = a.bigN ? (size_t)oct_lds_layout(a.poolCap, a.bigN * 8, false).total : 0;
    if (a.bigN && a.sortCap > 0)
        FT_HIP(hipFuncSetAttribute((const void *)k_octree_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smemBig));
    if (a.histFirst) {
        const size_t smemAll = std::max(ft_octree_hist_smem_bytes(a.poolCap), smem);  // (a.low: the sorted fallback in the compact layout)
        if (smemAll > 64 * 1024)
            FT_HIP(hipFuncSetAttribute((const void *)k_octree_auto, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smemAll));
        for (int rep = ft_debug_repeat("octree"); rep > 0; rep--) {
            if (a.bigN) FT_HIP(hipMemsetAsync(a.bigCount, 0, 2 * sizeof(int), st));
            hipLaunchKernelGGL(k_octree_auto, dim3(batch, g.nlevels), dim3(OCT_THREADS), smemAll, st, g, a);
            if (a.bigN && a.sortCap > 0) hipLaunchKernelGGL(k_octree_big, dim3(a.sortCap), dim3(OCT_THREADS), smemBig, st, g, a);
        }
        FT_HIP(hipGetLastError());
        return FT_OK;
    }
    for (int rep = ft_debug_repeat("octree"); rep > 0; rep--) {
        // the list counters of the launch ([2] and [3], the demand, are the host's to reset)
        if (a.bigN || a.histCap > 0 || a.histWanted) FT_HIP(hipMemsetAsync(a.bigCount, 0, 2 * sizeof(int), st));
        hipLaunchKernelGGL(k_octree, dim3(batch, g.nlevels), dim3(OCT_THREADS), smem, st, g, a);
        if (a.histCap > 0) hipLaunchKernelGGL(k_octree_hist, dim3(std::min(a.histCap, std::max(a.histGrid, 1))), dim3(OCT_THREADS), smemHist, st, g, a);
        if (a.bigN && a.sortCap > 0) hipLaunchKernelGGL(k_octree_big, dim3(a.sortCap), dim3(OCT_THREADS), smemBig, st, g, a);
    }
    FT_HIP(hipGetLastError());
    return FT_OK;
}
