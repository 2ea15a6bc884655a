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
    auto bestOf = [&](int lo, int hi) {
        int best = (int)(uint32_t)keys[lo];
        for (int k = lo + 1; k < hi; k++) {
            const int i = (int)(uint32_t)keys[k];
            const unsigned r = cand[i] >> 24, rb = cand[best] >> 24;
            if (r > rb || (r == rb && i < best)) best = i;
        }
        return best;
    };
    std::vector<int> res(std::max(N + 8, 4 * R.nIni + 8));
    const int k = op::distribute(codes.data(), n, R, N, ws, bestOf, res.data(), (int)res.size());
    if (k < 0) return -1;
    for (int i = 0; i < k && i < (int)res.size(); i++) out.push_back(res[i]);
    return k;
}

}  // namespace ft
