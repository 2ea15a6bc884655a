// See octree.h.  Compiled with -ffp-contract=off: the float expressions below must evaluate exactly
// as the reference's (src/ORBextractor.cc:510-566, 660-884).
#include "octree.h"
#include "octree_paths.h"

#include <algorithm>
#include <cmath>

namespace ft {

namespace {

inline float candX(uint32_t c) { return (float)(c & 0xfffu); }
inline float candY(uint32_t c) { return (float)((c >> 12) & 0xfffu); }
inline float candR(uint32_t c) { return (float)(c >> 24); }

struct NodeList {
    std::vector<OctreeWorkspace::Node> &pool;
    int head = -1, tail = -1, size = 0;
    explicit NodeList(std::vector<OctreeWorkspace::Node> &p) : pool(p) {}
    void push_back(int i) {
        pool[i].prev = tail;
        pool[i].next = -1;
        if (tail >= 0) pool[tail].next = i;
        else head = i;
        tail = i;
        size++;
    }
    void push_front(int i) {
        pool[i].next = head;
        pool[i].prev = -1;
        if (head >= 0) pool[head].prev = i;
        else tail = i;
        head = i;
        size++;
    }
    int erase(int i) {  // returns the element after i
        const int p = pool[i].prev, n = pool[i].next;
        if (p >= 0) pool[p].next = n;
        else head = n;
        if (n >= 0) pool[n].prev = p;
        else tail = p;
        size--;
        return n;
    }
};

// ExtractorNode::DivideNode: split node `ni` into four children occupying the parent's key range.
// Keys are (candidate | index << 32) records that ping-pong between two buffers: one counting pass
// (quadrant codes kept in a byte array) and one stable scatter, no copy back, no indirection.
// Only non-empty children are created; child[q] = -1 otherwise.  Candidate coordinates are small
// non-negative integers, so the reference's float comparisons `pt.x < UR.x` are exact as int compares.
void divide(OctreeWorkspace &ws, int ni, int child[4]) {
    const OctreeWorkspace::Node nd = ws.pool[ni];
    const int halfX = (int)std::ceil(static_cast<float>(nd.x1 - nd.x0) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(nd.y1 - nd.y0) / 2);
    const int mx = nd.x0 + halfX, my = nd.y0 + halfY;
    const uint64_t *src = (nd.buf ? ws.keysB : ws.keysA).data();
    uint64_t *dst = (nd.buf ? ws.keysA : ws.keysB).data();
    uint8_t *quad = ws.quad.data();
    int cnt[4] = {0, 0, 0, 0};
    // quadrant: 0 = n1 (left, top) 1 = n2 (right, top) 2 = n3 (left, bottom) 3 = n4 (right, bottom)
    for (int k = nd.begin; k < nd.end; k++) {
        const uint32_t c = (uint32_t)src[k];
        const int q = ((int)(c & 0xfffu) < mx ? 0 : 1) + ((int)((c >> 12) & 0xfffu) < my ? 0 : 2);
        quad[k] = (uint8_t)q;
        cnt[q]++;
    }
    int off[4];
    off[0] = nd.begin;
    off[1] = off[0] + cnt[0];
    off[2] = off[1] + cnt[1];
    off[3] = off[2] + cnt[2];
    int cur[4] = {off[0], off[1], off[2], off[3]};
    for (int k = nd.begin; k < nd.end; k++) dst[cur[quad[k]]++] = src[k];
    const int bx[4][4] = {{nd.x0, nd.y0, mx, my}, {mx, nd.y0, nd.x1, my}, {nd.x0, my, mx, nd.y1}, {mx, my, nd.x1, nd.y1}};
    for (int q = 0; q < 4; q++) {
        if (cnt[q] == 0) {
            child[q] = -1;
            continue;
        }
        OctreeWorkspace::Node ch;
        ch.x0 = bx[q][0];
        ch.y0 = bx[q][1];
        ch.x1 = bx[q][2];
        ch.y1 = bx[q][3];
        ch.begin = off[q];
        ch.end = off[q] + cnt[q];
        ch.prev = ch.next = -1;
        ch.noMore = cnt[q] == 1;
        ch.buf = nd.buf ^ 1;
        child[q] = (int)ws.pool.size();
        ws.pool.push_back(ch);
    }
}

}  // namespace

int octree_max_result(int minX, int maxX, int minY, int maxY, int N) {
    int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    if (nIni < 1) nIni = 1;
    return std::max(4 * nIni, N + 3);
}

int distribute_octree(const uint32_t *cand, int n, int minX, int maxX, int minY, int maxY, int N,
                      OctreeWorkspace &ws, std::vector<int> &out) {
    if (n <= 0) return 0;
    ws.pool.clear();
    ws.pool.reserve(4 * (size_t)std::max(N, 64) + 64);
    ws.keysA.resize(n);
    ws.keysB.resize(n);
    ws.quad.resize(n);
    int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    if (nIni < 1) nIni = 1;  // the reference divides by zero here for very tall images
    const float hX = static_cast<float>(maxX - minX) / nIni;
    NodeList list(ws.pool);
    // initial column nodes; keys are bucketed stably so each node sees them in emission order
    std::vector<int> &slotOf = ws.scratch;
    slotOf.resize(n);
    std::vector<int> cnt(nIni + 1, 0);
    for (int i = 0; i < n; i++) {
        int s = (int)(candX(cand[i]) / hX);
        if (s >= nIni) s = nIni - 1;
        slotOf[i] = s;
        cnt[s + 1]++;
    }
    for (int s = 0; s < nIni; s++) cnt[s + 1] += cnt[s];
    {
        std::vector<int> cur(cnt.begin(), cnt.end() - 1);
        for (int i = 0; i < n; i++) ws.keysA[cur[slotOf[i]]++] = (uint64_t)cand[i] | ((uint64_t)i << 32);
    }
    for (int s = 0; s < nIni; s++) {
        OctreeWorkspace::Node nd;
        nd.x0 = (int)(hX * static_cast<float>(s));
        nd.x1 = (int)(hX * static_cast<float>(s + 1));
        nd.y0 = 0;
        nd.y1 = maxY - minY;
        nd.begin = cnt[s];
        nd.end = cnt[s + 1];
        nd.prev = nd.next = -1;
        nd.noMore = (nd.end - nd.begin) == 1;
        nd.buf = 0;
        ws.pool.push_back(nd);
        if (nd.end > nd.begin) list.push_back((int)ws.pool.size() - 1);  // empty nodes are erased at once
    }
    auto &vSize = ws.sizeAndNode;
    auto &vPrev = ws.prevSizeAndNode;
    vSize.clear();
    auto cmp = [&](const std::pair<int, int> &a, const std::pair<int, int> &b) {
        if (a.first < b.first) return true;
        if (a.first > b.first) return false;
        return ws.pool[a.second].x0 < ws.pool[b.second].x0;
    };
    auto pushChildren = [&](const int child[4], int *nToExpand) {
        for (int q = 0; q < 4; q++) {
            const int ci = child[q];
            if (ci < 0) continue;
            const int sz = ws.pool[ci].end - ws.pool[ci].begin;
            list.push_front(ci);
            if (sz > 1) {
                if (nToExpand) (*nToExpand)++;
                vSize.push_back(std::make_pair(sz, ci));
            }
        }
    };
    bool finish = false;
    while (!finish) {
        int prevSize = list.size;
        int nToExpand = 0;
        vSize.clear();
        int it = list.head;
        while (it >= 0) {
            if (ws.pool[it].noMore) {
                it = ws.pool[it].next;
                continue;
            }
            int child[4];
            divide(ws, it, child);
            pushChildren(child, &nToExpand);
            it = list.erase(it);
        }
        if (list.size >= N || list.size == prevSize) {
            finish = true;
        } else if (list.size + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = list.size;
                vPrev = vSize;
                vSize.clear();
                std::sort(vPrev.begin(), vPrev.end(), cmp);
                for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
                    int child[4];
                    divide(ws, vPrev[j].second, child);
                    pushChildren(child, nullptr);
                    list.erase(vPrev[j].second);
                    if (list.size >= N) break;
                }
                if (list.size >= N || list.size == prevSize) finish = true;
            }
        }
    }
    int kept = 0;
    for (int it = list.head; it >= 0; it = ws.pool[it].next) {
        const OctreeWorkspace::Node &nd = ws.pool[it];
        const uint64_t *keys = (nd.buf ? ws.keysB : ws.keysA).data();
        uint64_t best = keys[nd.begin];
        unsigned maxResponse = (uint32_t)best >> 24;
        for (int k = nd.begin + 1; k < nd.end; k++) {
            const unsigned r = (uint32_t)keys[k] >> 24;
            if (r > maxResponse) {
                best = keys[k];
                maxResponse = r;
            }
        }
        out.push_back((int)(best >> 32));
        kept++;
    }
    return kept;
}

// Single-thread host run of the path-code formulation (octree_paths.h) that the device kernel executes:
// codes -> sort by (code, index) -> tree replay.  Used by tests to check the formulation against
// distribute_octree on the CPU before it ever runs on a GPU.
int distribute_octree_paths(const uint32_t *cand, int n, int minX, int maxX, int minY, int maxY, int N,
                            std::vector<int> &out) {
    if (n <= 0) return 0;
    const op::Roots R = op::make_roots(minX, maxX, minY, maxY);
    std::vector<uint64_t> keys(n);
    for (int i = 0; i < n; i++)
        keys[i] = ((uint64_t)op::path_code(R, (int)(cand[i] & 0xfffu), (int)((cand[i] >> 12) & 0xfffu)) << 32) | (uint32_t)i;
    std::sort(keys.begin(), keys.end());
    std::vector<uint32_t> codes(n);
    for (int i = 0; i < n; i++) codes[i] = (uint32_t)(keys[i] >> 32);
    const int poolCap = N + 4 * R.nIni + 16;
    std::vector<op::Node> pool(poolCap);
    std::vector<uint16_t> freeList(poolCap);
    std::vector<op::SortElem> vSize(poolCap), vPrev(poolCap);
    op::Workspace ws{pool.data(), freeList.data(), vSize.data(), vPrev.data(), poolCap};
    std::vector<int> res;
    auto emit = [&](int, int lo, int hi) {
        int best = (int)(uint32_t)keys[lo];
        for (int k = lo + 1; k < hi; k++) {
            const int i = (int)(uint32_t)keys[k];
            const unsigned r = cand[i] >> 24, rb = cand[best] >> 24;
            if (r > rb || (r == rb && i < best)) best = i;
        }
        res.push_back(best);
    };
    const int k = op::distribute([&](int i) { return codes[i]; }, n, R, N, ws, emit);
    if (k < 0) return -1;
    for (int i = 0; i < k && i < (int)res.size(); i++) out.push_back(res[i]);
    return k;
}

// Single-thread host statement of the ROUND formulation the device kernels execute (kernels_octree.hip):
// the node list is an array, and a whole pass of the reference - every node of a breadth-first pass, or
// the first `nproc` nodes of the size-sorted careful pass - is split at once.  The list after a pass is
//   [children of the last processed node, n4..n1] ... [children of the first processed node, n4..n1]
//   followed by the nodes that were not split, in their old order
// (std::list::push_front of every child + erase of the parent, ORBextractor.cc:741-789 / :808-848), so
// every position is a prefix sum.  Only the careful pass's std::sort stays sequential (replayed literally).
//
// Where a node's keys lie is the business of a policy:
//   SortedKeys (k_octree, k_octree_big): the keys are sorted by path code, a node owns a contiguous range and its
//       children's boundaries are three binary searches;
//   HistKeys (k_octree_hist): nothing is sorted.  The candidates are COUNTED per node of depth D (a histogram over
//       the code prefixes of depth D, nIni * 4^D bins in code order); the exclusive prefix sums of the bins are the
//       positions the sorted array would have, so the boundaries of a node of depth < D are table look-ups, and the
//       pick at the end is one more pass over the candidates (bin -> final node -> maximum of (response, -rank)).
//       A node of depth D that would have to be split is beyond the table: the policy gives up (-2) and the level
//       goes to the sorted formulation.
namespace {

struct RNode {
    int lo, hi, x0, x1, depth;
    uint32_t pre;  // code prefix of the node: root, then one quadrant digit per depth
};

struct SortedKeys {
    const uint32_t *cand;
    int n;
    op::Roots R;
    std::vector<uint64_t> keys;
    SortedKeys(const uint32_t *c, int n_, const op::Roots &r) : cand(c), n(n_), R(r), keys(n_) {
        for (int i = 0; i < n; i++)
            keys[i] = ((uint64_t)op::path_code(R, (int)(cand[i] & 0xfffu), (int)((cand[i] >> 12) & 0xfffu)) << 32) | (uint32_t)i;
        std::sort(keys.begin(), keys.end());
    }
    uint32_t codeAt(int i) const { return (uint32_t)(keys[i] >> 32); }
    void rootRange(int s, int &lo, int &hi) const {
        auto at = [&](int i) { return codeAt(i); };
        lo = op::lower_bound_code(at, 0, n, (uint32_t)s << (2 * op::kMaxDepth));
        hi = op::lower_bound_code(at, lo, n, (uint32_t)(s + 1) << (2 * op::kMaxDepth));
    }
    bool children(const RNode &nd, int b[5]) const {
        b[0] = nd.lo;
        b[1] = b[2] = b[3] = b[4] = nd.hi;
        if (nd.depth < op::kMaxDepth) {
            auto at = [&](int i) { return codeAt(i); };
            const int shift = 2 * (op::kMaxDepth - 1 - nd.depth);
            for (int k = 1; k < 4; k++) b[k] = op::lower_bound_code(at, b[k - 1], nd.hi, ((nd.pre << 2) | (uint32_t)k) << shift);
        }
        return true;
    }
    void pick(const RNode *nodes, int m, std::vector<int> &out) const {
        for (int t = 0; t < m; t++) {
            int best = (int)(uint32_t)keys[nodes[t].lo];
            for (int k = nodes[t].lo + 1; k < nodes[t].hi; k++) {
                const int i = (int)(uint32_t)keys[k];
                const unsigned r = cand[i] >> 24, rb = cand[best] >> 24;
                if (r > rb || (r == rb && i < best)) best = i;
            }
            out.push_back(best);
        }
    }
};

struct HistKeys {
    const uint32_t *cand;
    int n, D;
    op::Roots R;
    std::vector<int> start;  // [nBins + 1]: candidates in bins below b
    HistKeys(const uint32_t *c, int n_, const op::Roots &r, int maxBins) : cand(c), n(n_), D(op::hist_depth(r.nIni, maxBins)), R(r) {
        const int nBins = r.nIni << (2 * D);
        start.assign(nBins + 1, 0);
        for (int i = 0; i < n; i++) start[op::path_prefix(R, (int)(cand[i] & 0xfffu), (int)((cand[i] >> 12) & 0xfffu), D) + 1]++;
        for (int b = 0; b < nBins; b++) start[b + 1] += start[b];
    }
    void rootRange(int s, int &lo, int &hi) const {
        lo = start[(size_t)s << (2 * D)];
        hi = start[(size_t)(s + 1) << (2 * D)];
    }
    bool children(const RNode &nd, int b[5]) const {
        if (nd.depth >= D) return false;  // the table ends here
        const int sh = 2 * (D - 1 - nd.depth);
        b[0] = nd.lo;
        for (int k = 1; k < 4; k++) b[k] = start[(size_t)((nd.pre << 2) | (uint32_t)k) << sh];
        b[4] = nd.hi;
        return true;
    }
    void pick(const RNode *nodes, int m, std::vector<int> &out) const {
        // bin -> final node: every node marks its first bin, a bin belongs to the last mark at or below it (the final
        // nodes tile the code space except for regions without candidates)
        const int nBins = R.nIni << (2 * D);
        std::vector<int> owner(nBins, -1), best(m, -1);
        for (int t = 0; t < m; t++) owner[(size_t)nodes[t].pre << (2 * (D - nodes[t].depth))] = t;
        for (int b = 1; b < nBins; b++)
            if (owner[b] < 0) owner[b] = owner[b - 1];
        for (int i = 0; i < n; i++) {
            const int t = owner[op::path_prefix(R, (int)(cand[i] & 0xfffu), (int)((cand[i] >> 12) & 0xfffu), D)];
            if (t < 0) continue;
            // first maximum in emission order = largest response, smallest index (the caller's order IS the emission order here)
            if (best[t] < 0 || (cand[i] >> 24) > (cand[best[t]] >> 24)) best[t] = i;
        }
        for (int t = 0; t < m; t++) out.push_back(best[t]);
    }
};

// returns the number of retained nodes, -2 when the policy gave up
template <class Keys>
int rounds_impl(const Keys &K, const op::Roots &R, int N, std::vector<int> &out) {
    const int cap = std::max(N + 3, 4 * R.nIni) + 16;
    std::vector<RNode> bufA(2 * cap), bufB(2 * cap);
    RNode *cur = bufA.data(), *nxt = bufB.data();
    std::vector<op::SortElem> vSize(cap), vPrev(cap);
    std::vector<int> ord(cap), P(cap), Q(cap), b1(cap), b2(cap), b3(cap);
    std::vector<uint8_t> mark(2 * cap);
    int start = cap, m = 0;
    for (int s = 0; s < R.nIni; s++) {
        int lo, hi;
        K.rootRange(s, lo, hi);
        if (hi == lo) continue;
        int x0, x1;
        op::root_bounds(R, s, x0, x1);
        cur[start + m++] = RNode{lo, hi, x0, x1, 0, (uint32_t)s};
    }
    int nV = 0;
    bool gaveUp = false;
    // splits ord[0..nOrd) (absolute positions in cur) in that order; with useStop the pass ends after the
    // split that brings the list to N nodes
    auto split_round = [&](int nOrd, bool useStop) {
        int cum = m, nproc = nOrd, p = 0, q = 0;
        for (int r = 0; r < nOrd; r++) {
            const RNode nd = cur[ord[r]];
            int b[5];
            if (!K.children(nd, b)) {
                gaveUp = true;
                return;
            }
            b1[r] = b[1]; b2[r] = b[2]; b3[r] = b[3];
            int nch = 0, nbig = 0;
            for (int k = 0; k < 4; k++) {
                nch += b[k + 1] > b[k];
                nbig += b[k + 1] - b[k] > 1;
            }
            P[r] = p; Q[r] = q;
            p += nch; q += nbig;
            cum += nch - 1;
            if (useStop && cum >= N) {
                nproc = r + 1;
                break;
            }
        }
        const int C = p;
        for (int t = start; t < start + m; t++) mark[t] = 0;
        for (int r = 0; r < nproc; r++) mark[ord[r]] = 1;
        for (int r = 0; r < nproc; r++) {
            const RNode nd = cur[ord[r]];
            const int b[5] = {nd.lo, b1[r], b2[r], b3[r], nd.hi};
            const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1);
            int k = 0, kb = 0;
            for (int c = 0; c < 4; c++) {
                const int cnt = b[c + 1] - b[c];
                if (cnt == 0) continue;
                const int pos = cap - 1 - (P[r] + k);
                nxt[pos] = RNode{b[c], b[c + 1], (c & 1) ? mx : nd.x0, (c & 1) ? nd.x1 : mx, nd.depth + 1, (nd.pre << 2) | (uint32_t)c};
                if (cnt > 1) {
                    vSize[Q[r] + kb].key = ((uint32_t)cnt << 16) | (uint32_t)nxt[pos].x0;
                    vSize[Q[r] + kb].val = (uint32_t)pos;
                    kb++;
                }
                k++;
            }
        }
        int u = 0;
        for (int t = start; t < start + m; t++)
            if (!mark[t]) nxt[cap + u++] = cur[t];
        start = cap - C;
        m = C + u;
        nV = q;
        std::swap(cur, nxt);
    };
    bool finish = false;
    while (!finish) {
        int prevSize = m;
        int nOrd = 0;
        for (int t = start; t < start + m; t++)
            if (cur[t].hi - cur[t].lo > 1) ord[nOrd++] = t;
        split_round(nOrd, false);
        if (gaveUp) return -2;
        if (m >= N || m == prevSize) {
            finish = true;
        } else if (m + 3 * nV > N) {
            while (!finish) {
                prevSize = m;
                const int nPrev = nV;
                for (int k = 0; k < nPrev; k++) vPrev[k] = vSize[k];
                op::std_sort_replay(vPrev.data(), vPrev.data() + nPrev);
                for (int r = 0; r < nPrev; r++) ord[r] = (int)vPrev[nPrev - 1 - r].val;
                split_round(nPrev, true);
                if (gaveUp) return -2;
                if (m >= N || m == prevSize) finish = true;
            }
        }
    }
    K.pick(cur + start, m, out);
    return m;
}

}  // namespace

int distribute_octree_rounds(const uint32_t *cand, int n, int minX, int maxX, int minY, int maxY, int N,
                             std::vector<int> &out) {
    if (n <= 0) return 0;
    if (n > 65535) return -1;
    const op::Roots R = op::make_roots(minX, maxX, minY, maxY);
    return rounds_impl(SortedKeys(cand, n, R), R, N, out);
}

int distribute_octree_hist(const uint32_t *cand, int n, int minX, int maxX, int minY, int maxY, int N, int maxBins,
                           std::vector<int> &out) {
    if (n <= 0) return 0;
    if (n > 65535) return -1;
    const op::Roots R = op::make_roots(minX, maxX, minY, maxY);
    if (R.nIni > maxBins) return -2;
    return rounds_impl(HistKeys(cand, n, R, maxBins), R, N, out);
}

}  // namespace ft
