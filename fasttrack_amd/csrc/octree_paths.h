// Octree keypoint distribution reformulated for the GPU ("path codes"), usable from host and device code.
//
// The reference (ORBextractor::DistributeOctTree, src/ORBextractor.cc:660-884) moves keypoint vectors from
// node to node.  The geometry of the subdivision does not depend on the data: a node's children are
// [x0, x0+ceil(w/2)) x [y0, y0+ceil(h/2)) etc. (DivideNode :510-566), and a key is sent to a child by two
// comparisons against the parent's mid lines.  So every key's whole root-to-leaf path - the initial column
// node (pt.x / hX, :686) followed by one quadrant digit per depth - can be computed independently, and
// after sorting the keys by path code every node at every depth owns a CONTIGUOUS range of the sorted
// array.  DivideNode then costs three boundary searches instead of a partition pass, no key ever moves, and
// only the node bookkeeping (list order, front insertion, the size-sorted "careful" phase with its
// std::sort tie behaviour, first-maximum pick) stays sequential.  That part is replayed literally.
//
// Everything here is plain C++ over caller-provided arrays so that the same code runs as a single-thread
// host function (tests, CPU check of the device logic) and inside the device kernel (uniform control flow
// in one wave, arrays in LDS).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define OP_HD __host__ __device__
#else
#define OP_HD
#endif

namespace ft {
namespace op {

constexpr int kMaxDepth = 12;  // 2^12 >= any level dimension (coordinates are 12-bit)

// ---- libstdc++ std::sort, replayed on (key, payload) pairs compared by key only -----------------------
// key = size << 16 | UL.x reproduces compareNodes (ORBextractor.cc:626-641); the sequence of comparisons
// and moves is exactly __introsort_loop + __final_insertion_sort of bits/stl_algo.h (threshold 16,
// median-of-three to first, unguarded Hoare partition, heap sort at depth 2*lg(n)), so ties end up where
// the reference's std::sort leaves them.
struct SortElem {
    uint32_t key;
    uint32_t val;
};

OP_HD inline bool se_less(const SortElem &a, const SortElem &b) { return a.key < b.key; }
OP_HD inline void se_swap(SortElem &a, SortElem &b) {
    const SortElem t = a;
    a = b;
    b = t;
}

OP_HD inline void ss_unguarded_linear_insert(SortElem *last) {
    const SortElem val = *last;
    SortElem *next = last - 1;
    while (se_less(val, *next)) {
        *last = *next;
        last = next;
        --next;
    }
    *last = val;
}

OP_HD inline void ss_insertion_sort(SortElem *first, SortElem *last) {
    if (first == last) return;
    for (SortElem *i = first + 1; i != last; ++i) {
        if (se_less(*i, *first)) {
            const SortElem val = *i;
            for (SortElem *p = i; p != first; --p) *p = *(p - 1);  // move_backward(first, i, i + 1)
            *first = val;
        } else {
            ss_unguarded_linear_insert(i);
        }
    }
}

OP_HD inline void ss_adjust_heap(SortElem *first, int holeIndex, int len, SortElem value) {
    const int topIndex = holeIndex;
    int secondChild = holeIndex;
    while (secondChild < (len - 1) / 2) {
        secondChild = 2 * (secondChild + 1);
        if (se_less(first[secondChild], first[secondChild - 1])) secondChild--;
        first[holeIndex] = first[secondChild];
        holeIndex = secondChild;
    }
    if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
        secondChild = 2 * (secondChild + 1);
        first[holeIndex] = first[secondChild - 1];
        holeIndex = secondChild - 1;
    }
    // __push_heap
    int parent = (holeIndex - 1) / 2;
    while (holeIndex > topIndex && se_less(first[parent], value)) {
        first[holeIndex] = first[parent];
        holeIndex = parent;
        parent = (holeIndex - 1) / 2;
    }
    first[holeIndex] = value;
}

OP_HD inline void ss_heap_sort(SortElem *first, SortElem *last) {  // __partial_sort(first, last, last)
    const int len = (int)(last - first);
    if (len >= 2) {  // __make_heap
        int parent = (len - 2) / 2;
        for (;;) {
            const SortElem value = first[parent];
            ss_adjust_heap(first, parent, len, value);
            if (parent == 0) break;
            parent--;
        }
    }
    while (last - first > 1) {  // __sort_heap / __pop_heap
        --last;
        const SortElem value = *last;
        *last = *first;
        ss_adjust_heap(first, 0, (int)(last - first), value);
    }
}

OP_HD inline SortElem *ss_partition_pivot(SortElem *first, SortElem *last) {
    SortElem *mid = first + (last - first) / 2;
    // __move_median_to_first(first, first + 1, mid, last - 1)
    SortElem *a = first + 1, *b = mid, *c = last - 1;
    if (se_less(*a, *b)) {
        if (se_less(*b, *c)) se_swap(*first, *b);
        else if (se_less(*a, *c)) se_swap(*first, *c);
        else se_swap(*first, *a);
    } else if (se_less(*a, *c)) se_swap(*first, *a);
    else if (se_less(*b, *c)) se_swap(*first, *c);
    else se_swap(*first, *b);
    // __unguarded_partition(first + 1, last, first)
    SortElem *lo = first + 1, *hi = last;
    for (;;) {
        while (se_less(*lo, *first)) ++lo;
        --hi;
        while (se_less(*first, *hi)) --hi;
        if (!(lo < hi)) return lo;
        se_swap(*lo, *hi);
        ++lo;
    }
}

// std::sort(first, last, compareNodes).  The recursion of __introsort_loop (recurse right, loop left) is
// unrolled with an explicit stack of (first, last, depth) ranges processed in the same order; `stack` holds
// 64 frames (the device passes LDS, a private array would live in scratch memory).
struct SortFrame {
    uint16_t f, l, depth, pad;
};

OP_HD inline void std_sort_replay(SortElem *first, SortElem *last, SortFrame *stack) {
    const int n = (int)(last - first);
    if (n <= 0) return;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) lg++;
    int sp = 0;
    stack[sp++] = SortFrame{0, (uint16_t)n, (uint16_t)(2 * lg), 0};
    while (sp > 0) {
        const SortFrame fr = stack[--sp];
        int f = fr.f, l = fr.l, depth = fr.depth;
        while (l - f > 16) {
            if (depth == 0) {
                ss_heap_sort(first + f, first + l);
                break;
            }
            --depth;
            const int cut = (int)(ss_partition_pivot(first + f, first + l) - first);
            // the reference recurses into [cut, last) first and then continues with [first, cut): the two
            // ranges are disjoint, so running the left one now and the right one later gives the same array
            if (sp < 64) stack[sp++] = SortFrame{(uint16_t)cut, (uint16_t)l, (uint16_t)depth, 0};
            l = cut;
        }
    }
    // __final_insertion_sort
    if (n > 16) {
        ss_insertion_sort(first, first + 16);
        for (SortElem *i = first + 16; i != last; ++i) ss_unguarded_linear_insert(i);
    } else {
        ss_insertion_sort(first, last);
    }
}

OP_HD inline void std_sort_replay(SortElem *first, SortElem *last) {
    SortFrame stack[64];
    std_sort_replay(first, last, stack);
}

// ---- the same std::sort as data-parallel steps (what the device kernel executes with one wave) ----------
// (a) __unguarded_partition is a two-pointer loop, but which elements it swaps is a closed form: with
//     A = positions of [first+1, last) holding an element that is not < pivot, ascending, and
//     B = positions holding an element that is not > pivot, descending, it swaps A[k] <-> B[k] for every
//     k < K, K = number of leading pairs with A[k] < B[k] (all swaps touch positions outside (A[k], B[k]), so
//     the scans between them see original data), and returns min(A[K], B[K-1]) (B[-1] = last).
// (b) __final_insertion_sort is a stable insertion sort of an array whose inversions all lie inside the
//     <= 16-element ranges introsort left unsorted: every element moves to
//     i - #{j in [i-16, i): key_j > key_i} + #{j in (i, i+16]: key_j < key_i}.
// This host statement exists so that the claim is checked against libstdc++ on the CPU
// (tests/cpp/test_sort_replay.cpp); posA / posB / tmp hold (last - first) entries.
inline int ss_partition_pairs(SortElem *a, int f, int l, uint16_t *posA, uint16_t *posB) {
    SortElem *first = a + f;
    SortElem *pa = first + 1, *pb = first + (l - f) / 2, *pc = a + l - 1;
    if (se_less(*pa, *pb)) {
        if (se_less(*pb, *pc)) se_swap(*first, *pb);
        else if (se_less(*pa, *pc)) se_swap(*first, *pc);
        else se_swap(*first, *pa);
    } else if (se_less(*pa, *pc)) se_swap(*first, *pa);
    else if (se_less(*pb, *pc)) se_swap(*first, *pc);
    else se_swap(*first, *pb);
    const uint32_t pv = first->key;
    int nA = 0, nB = 0;
    for (int i = f + 1; i < l; i++)
        if (!(a[i].key < pv)) posA[nA++] = (uint16_t)i;
    for (int i = l - 1; i > f; i--)
        if (!(pv < a[i].key)) posB[nB++] = (uint16_t)i;
    int K = 0;
    while (K < nA && K < nB && posA[K] < posB[K]) K++;
    for (int k = 0; k < K; k++) se_swap(a[posA[k]], a[posB[k]]);
    const int bK = K ? posB[K - 1] : l;
    return (K < nA && posA[K] < bK) ? posA[K] : bK;
}

inline void std_sort_replay_steps(SortElem *first, SortElem *last, SortFrame *stack, uint16_t *posA, uint16_t *posB,
                                  SortElem *tmp) {
    const int n = (int)(last - first);
    if (n <= 0) return;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) lg++;
    int sp = 0;
    stack[sp++] = SortFrame{0, (uint16_t)n, (uint16_t)(2 * lg), 0};
    while (sp > 0) {
        const SortFrame fr = stack[--sp];
        int f = fr.f, l = fr.l, depth = fr.depth;
        while (l - f > 16) {
            if (depth == 0) {
                ss_heap_sort(first + f, first + l);
                break;
            }
            --depth;
            const int cut = ss_partition_pairs(first, f, l, posA, posB);
            if (sp < 64) stack[sp++] = SortFrame{(uint16_t)cut, (uint16_t)l, (uint16_t)depth, 0};
            l = cut;
        }
    }
    for (int i = 0; i < n; i++) {
        int pos = i;
        for (int j = (i > 16 ? i - 16 : 0); j < i; j++) pos -= first[j].key > first[i].key;
        for (int j = i + 1; j <= i + 16 && j < n; j++) pos += first[j].key < first[i].key;
        tmp[pos] = first[i];
    }
    for (int i = 0; i < n; i++) first[i] = tmp[i];
}

// ---- path codes ---------------------------------------------------------------------------------------
struct Roots {
    int nIni;
    float hX;
    int H;  // maxY - minY
};

OP_HD inline Roots make_roots(int minX, int maxX, int minY, int maxY) {
    Roots r;
    // ORBextractor.cc:664-666: nIni = round(float(maxX-minX)/(maxY-minY)), hX = float(maxX-minX)/nIni
    r.nIni = (int)roundf(static_cast<float>(maxX - minX) / (maxY - minY));
    if (r.nIni < 1) r.nIni = 1;
    r.hX = static_cast<float>(maxX - minX) / r.nIni;
    r.H = maxY - minY;
    return r;
}

OP_HD inline void root_bounds(const Roots &r, int s, int &x0, int &x1) {
    x0 = (int)(r.hX * static_cast<float>(s));      // :676
    x1 = (int)(r.hX * static_cast<float>(s + 1));  // :677
}

// code = root << 24 | q_1 << 22 | q_2 << 20 | ... | q_12 (q_d = quadrant chosen at depth d: 0 n1, 1 n2, 2 n3, 3 n4)
OP_HD inline uint32_t path_code(const Roots &r, int x, int y) {
    int s = (int)((float)x / r.hX);  // vpIniNodes[kp.pt.x / hX] (:690)
    if (s >= r.nIni) s = r.nIni - 1;
    int x0, x1, y0 = 0, y1 = r.H;
    root_bounds(r, s, x0, x1);
    uint32_t code = (uint32_t)s;
    for (int d = 0; d < kMaxDepth; d++) {
        // DivideNode: halfX = ceil(float(UR.x-UL.x)/2) == (w+1)>>1 for the non-negative ints at hand
        const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
        const int q = (x < mx ? 0 : 1) + (y < my ? 0 : 2);
        if (q & 1) x0 = mx; else x1 = mx;
        if (q & 2) y0 = my; else y1 = my;
        code = (code << 2) | (uint32_t)q;
    }
    return code;
}

// The first D digits of the path code: root << 2D | q_1 << 2(D-1) | ... | q_D, i.e. the index of the candidate's node of
// depth D in code order (path_code(...) >> 2 * (kMaxDepth - D)).  The histogram formulation (k_octree_hist) counts the
// candidates per such node instead of sorting them.
OP_HD inline uint32_t path_prefix(const Roots &r, int x, int y, int D) {
    int s = (int)((float)x / r.hX);
    if (s >= r.nIni) s = r.nIni - 1;
    int x0, x1, y0 = 0, y1 = r.H;
    root_bounds(r, s, x0, x1);
    uint32_t code = (uint32_t)s;
    for (int d = 0; d < D; d++) {
        const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
        const int q = (x < mx ? 0 : 1) + (y < my ? 0 : 2);
        if (q & 1) x0 = mx; else x1 = mx;
        if (q & 2) y0 = my; else y1 = my;
        code = (code << 2) | (uint32_t)q;
    }
    return code;
}

// depth of the histogram: the deepest D whose nIni * 4^D nodes fit maxBins (0 when not even the roots do)
OP_HD inline int hist_depth(int nIni, int maxBins) {
    int D = 0;
    while (D < kMaxDepth && ((long long)nIni << (2 * (D + 1))) <= (long long)maxBins) D++;
    return D;
}

// ---- tree replay over sorted codes ----------------------------------------------------------------------
struct Node {
    uint16_t x0, y0, x1, y1;
    uint16_t lo, hi;  // range in the sorted key array
    uint16_t prev, next;
    uint8_t depth;    // number of quadrant digits fixed (0 for a root)
    uint8_t noMore;
};
constexpr uint16_t kNil = 0xffff;

struct Workspace {
    Node *pool;          // [poolCap]
    uint16_t *freeList;  // [poolCap]
    SortElem *vSize;     // [poolCap]
    SortElem *vPrev;     // [poolCap]
    int poolCap;
};

// first index in [lo, hi) whose code is >= target (codes ascending); codeAt(i) returns the i-th sorted code
template <class CodeAt>
OP_HD inline int lower_bound_code(CodeAt codeAt, int lo, int hi, uint32_t target) {
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (codeAt(mid) < target) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// Replays DistributeOctTree on keys sorted by (path code, original index).
//   codeAt(i)      i-th path code of the sorted keys, i in [0, n)
//   emit(pos, lo, hi) is called once per retained node in the reference's result (list) order with the node's
//                  range in the sorted array; the caller picks the key with the largest response there,
//                  earliest original index on ties (the reference keeps the first maximum in emission order)
// returns the number of retained nodes, or -1 when the workspace is too small.
template <class CodeAt, class Emit>
OP_HD inline int distribute(CodeAt codeAt, int n, const Roots &R, int N, Workspace &ws, Emit emit) {
    if (n <= 0) return 0;
    Node *pool = ws.pool;
    int nFree = 0, poolUsed = 0;
    auto alloc = [&]() -> int {
        if (nFree > 0) return ws.freeList[--nFree];
        if (poolUsed >= ws.poolCap) return -1;
        return poolUsed++;
    };
    int head = kNil, tail = kNil, size = 0;
    auto push_back = [&](int i) {
        pool[i].prev = (uint16_t)tail;
        pool[i].next = kNil;
        if (tail != kNil) pool[tail].next = (uint16_t)i;
        else head = i;
        tail = i;
        size++;
    };
    auto push_front = [&](int i) {
        pool[i].next = (uint16_t)head;
        pool[i].prev = kNil;
        if (head != kNil) pool[head].prev = (uint16_t)i;
        else tail = i;
        head = i;
        size++;
    };
    auto erase = [&](int i) -> int {
        const int p = pool[i].prev, nx = pool[i].next;
        if (p != kNil) pool[p].next = (uint16_t)nx;
        else head = nx;
        if (nx != kNil) pool[nx].prev = (uint16_t)p;
        else tail = p;
        size--;
        ws.freeList[nFree++] = (uint16_t)i;
        return nx;
    };
    // roots: keys of root s are the codes with top bits == s
    for (int s = 0; s < R.nIni; s++) {
        const int lo = lower_bound_code(codeAt, 0, n, (uint32_t)s << (2 * kMaxDepth));
        const int hi = lower_bound_code(codeAt, lo, n, (uint32_t)(s + 1) << (2 * kMaxDepth));
        if (hi == lo) continue;  // empty initial nodes are erased at once (:704-705)
        const int i = alloc();
        if (i < 0) return -1;
        int x0, x1;
        root_bounds(R, s, x0, x1);
        Node nd;
        nd.x0 = (uint16_t)x0; nd.x1 = (uint16_t)x1; nd.y0 = 0; nd.y1 = (uint16_t)R.H;
        nd.lo = (uint16_t)lo; nd.hi = (uint16_t)hi;
        nd.prev = nd.next = kNil;
        nd.depth = 0;
        nd.noMore = (hi - lo) == 1;
        pool[i] = nd;
        push_back(i);
    }
    int nVSize = 0;
    bool overflow = false;
    // DivideNode + "add childs if they contain points" (:741-789 / :808-848)
    auto divide_and_push = [&](int ni, int *nToExpand) {
        const Node nd = pool[ni];
        const int d = nd.depth;
        int b[5];
        b[0] = nd.lo;
        b[4] = nd.hi;
        if (d >= kMaxDepth) {  // cannot happen for distinct pixels; keep everything in n1
            b[1] = b[2] = b[3] = nd.hi;
        } else {
            const int shift = 2 * (kMaxDepth - 1 - d);
            const uint32_t prefix = codeAt((int)nd.lo) >> (shift + 2);
            for (int q = 1; q < 4; q++) b[q] = lower_bound_code(codeAt, b[q - 1], nd.hi, ((prefix << 2) | (uint32_t)q) << shift);
        }
        const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
        const int bx[4][4] = {{nd.x0, nd.y0, mx, my}, {mx, nd.y0, nd.x1, my}, {nd.x0, my, mx, nd.y1}, {mx, my, nd.x1, nd.y1}};
        for (int q = 0; q < 4; q++) {
            const int cnt = b[q + 1] - b[q];
            if (cnt == 0) continue;
            const int ci = alloc();
            if (ci < 0) {
                overflow = true;
                return;
            }
            Node ch;
            ch.x0 = (uint16_t)bx[q][0]; ch.y0 = (uint16_t)bx[q][1]; ch.x1 = (uint16_t)bx[q][2]; ch.y1 = (uint16_t)bx[q][3];
            ch.lo = (uint16_t)b[q]; ch.hi = (uint16_t)b[q + 1];
            ch.prev = ch.next = kNil;
            ch.depth = (uint8_t)(d + 1);
            ch.noMore = cnt == 1;
            pool[ci] = ch;
            push_front(ci);
            if (cnt > 1) {
                if (nToExpand) (*nToExpand)++;
                ws.vSize[nVSize].key = ((uint32_t)cnt << 16) | (uint32_t)ch.x0;
                ws.vSize[nVSize].val = (uint32_t)ci;
                nVSize++;
            }
        }
    };
    bool finish = false;
    while (!finish && !overflow) {
        int prevSize = size;
        int nToExpand = 0;
        nVSize = 0;
        int it = head;
        while (it != kNil && !overflow) {
            if (pool[it].noMore) {
                it = pool[it].next;
                continue;
            }
            // the children take pool slots before the parent is released, exactly like the reference
            // constructs n1..n4 before lNodes.erase(lit)
            divide_and_push(it, &nToExpand);
            it = erase(it);
        }
        if (overflow) break;
        if (size >= N || size == prevSize) {
            finish = true;
        } else if (size + nToExpand * 3 > N) {
            while (!finish && !overflow) {
                prevSize = size;
                const int nPrev = nVSize;
                for (int k = 0; k < nPrev; k++) ws.vPrev[k] = ws.vSize[k];
                nVSize = 0;
                std_sort_replay(ws.vPrev, ws.vPrev + nPrev);
                for (int j = nPrev - 1; j >= 0; j--) {
                    const int ni = (int)ws.vPrev[j].val;
                    divide_and_push(ni, nullptr);
                    if (overflow) break;
                    erase(ni);
                    if (size >= N) break;
                }
                if (size >= N || size == prevSize) finish = true;
            }
        }
    }
    if (overflow) return -1;
    int kept = 0;
    for (int it = head; it != kNil; it = pool[it].next) {
        emit(kept, (int)pool[it].lo, (int)pool[it].hi);
        kept++;
    }
    return kept;
}

}  // namespace op
}  // namespace ft
