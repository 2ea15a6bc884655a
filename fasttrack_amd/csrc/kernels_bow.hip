// HIP kernels of the bag-of-words steps for gfx950 (wave64):
//   k_search_by_bow  ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (reference src/ORBmatcher.cc:322-488): the matching
//                inside the vocabulary nodes the two FeatureVectors share, one wave per node (see there)
//   k_bow_walk   TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)
//                (reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1208-1253) for every descriptor of a frame,
//                distances as FORB::distance (Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101)
// A group of 16 lanes walks one descriptor down the tree (four descriptors per wave): at every level each lane of the
// group takes one child of the current node (children beyond 16 in further rounds), reads its 32-byte descriptor as two
// 16-byte loads and forms the key (distance << 8 | child rank); the minimum key of the group - four DPP steps inside
// the row of 16 - is the reference's "first child with the smallest distance".  A level is one memory round trip.
#include "ft_internal.h"
#include "wave_ops.h"

namespace {

// minimum over the 16 lanes of a DPP row, returned in every lane of the row
__device__ __forceinline__ unsigned row_min_u32(unsigned v) {
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));  // row_half_mirror
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

__global__ __launch_bounds__(256) void k_bow_walk(FtBowTree t, const uint8_t *desc, int n, int nidLevel,
                                                  unsigned *wordOut, unsigned *nodeOut, double *weightOut) {
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const int i = (int)((blockIdx.x * 256u + threadIdx.x) >> 4);  // descriptor of this 16-lane group
    const bool live = i < n;
    // the group's descriptor (clamped read for the idle groups of the last wave: they walk descriptor n - 1)
    const unsigned long long *dp = (const unsigned long long *)(desc + (size_t)min(i, n - 1) * 32);
    const unsigned long long d0 = gload<unsigned long long>(dp), d1 = gload<unsigned long long>(dp + 1),
                             d2 = gload<unsigned long long>(dp + 2), d3 = gload<unsigned long long>(dp + 3);
    unsigned node = 0, nid = 0;  // root; nid stays 0 when nidLevel <= 0 (:1218)
    int level = 0;
    int first = t.childStart[0], count = t.childStart[1] - first;
    while (count > 0) {  // !isLeaf()
        level++;
        unsigned best = 0xffffffffu;
        for (int c0 = 0; c0 < count; c0 += 16) {
            const int c = c0 + sub;
            unsigned key = 0xffffffffu;
            if (c < count) {
                const unsigned child = t.childList[first + c];
                const unsigned long long *q = (const unsigned long long *)(t.desc + (size_t)child * 32);
                const int dist = __popcll(d0 ^ q[0]) + __popcll(d1 ^ q[1]) + __popcll(d2 ^ q[2]) + __popcll(d3 ^ q[3]);
                key = ((unsigned)dist << 16) | (unsigned)c;  // count <= 65535 children (k <= 20 in the reference's loader)
            }
            best = min(best, row_min_u32(key));
        }
        node = t.childList[first + (int)(best & 0xffffu)];
        if (level == nidLevel) nid = node;
        first = t.childStart[node];
        count = t.childStart[node + 1] - first;
    }
    if (nidLevel > level) nid = node;  // the walk ended above the requested level (the reference leaves *nid unset)
    if (live && sub == 0) {
        if (wordOut) wordOut[i] = t.wordId[node];
        if (nodeOut) nodeOut[i] = nid;
        if (weightOut) weightOut[i] = t.weight[node];
    }
}


// SearchByBoW.  A frame feature sits in exactly one node of the FeatureVector, so the "already matched" test of the reference
// (vpMapPointMatches[realIdxF], :365 / :387) only ever looks at assignments made inside the same node: the nodes are
// independent, the keyframe features of a node are sequential (each sees the claims of the ones in front of it), the frame
// features of a node are parallel.  One wave per keyframe node: binary search for the node on the frame side (the
// reference's lower_bound walk visits exactly the common keys), then for every keyframe feature with a map point the 64
// lanes take the node's frame features 64 at a time and keep their two smallest (distance << 20 | position) keys; the
// wave's smallest key is the reference's "first feature with the smallest distance" (strict < in scan order, :373), the
// second smallest key carries bestDist2.  Two-camera frames (Nleft != -1) keep a second pair of keys for the right camera
// (:404-421).  Claims live in matches[] and are written and read back by the SAME lane (position & 63), so no ordering
// between lanes is needed.
__global__ __launch_bounds__(256) void k_search_by_bow(FtBowSide K, const uint8_t *kfHasPoint, FtBowSide F, int nleft, float nnRatio,
                                                       int *matches) {
    const int lane = threadIdx.x & 63;
    const int a = (int)blockIdx.x * 4 + wave_index();
    if (a >= K.nNodes) return;
    const unsigned node = K.nodes[a];
    int lo = 0, hi = F.nNodes;  // lower_bound
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (F.nodes[mid] < node) lo = mid + 1;
        else hi = mid;
    }
    if (lo >= F.nNodes || F.nodes[lo] != node) return;
    const int fBeg = F.offsets[lo], fCnt = F.offsets[lo + 1] - fBeg;
    const int kBeg = K.offsets[a], kEnd = K.offsets[a + 1];
    constexpr unsigned NONE = (256u << 20) | 0xfffffu;  // bestDist = 256, no index
    auto twoMin = [](unsigned &x0, unsigned &x1) {
        const unsigned m0 = wave_min_u32(x0);
        const unsigned m1 = wave_min_u32(x0 == m0 ? x1 : x0);
        x0 = m0;
        x1 = m1;
    };
    constexpr int R = 4;  // frame features per lane held in registers
    if (fCnt <= 64 * R) {
        // The usual node (a few dozen to a few hundred features) lives in registers: lane l holds the node's frame features
        // l, l + 64, ... - index, descriptor, "still free" - and, 64 at a time, its keyframe features - index, map-point flag,
        // descriptor; the sequential walk over the keyframe features broadcasts one of them per step (v_readlane) and touches
        // no memory until the final stores.
        unsigned fIdx[R];
        unsigned long long f0[R], f1[R], f2[R], f3[R];
        int mine[R];
        unsigned freeMask = 0, rightMask = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int p = r * 64 + lane;
            const bool have = p < fCnt;
            fIdx[r] = have ? F.features[fBeg + p] : 0u;
            const unsigned long long *q = (const unsigned long long *)(F.desc + (size_t)fIdx[r] * 32);
            f0[r] = q[0]; f1[r] = q[1]; f2[r] = q[2]; f3[r] = q[3];
            mine[r] = -1;
            freeMask |= (have ? 1u : 0u) << r;
            rightMask |= ((nleft != -1 && (int)fIdx[r] >= nleft) ? 1u : 0u) << r;
        }
        auto bcast64 = [](unsigned long long v, int j) {
            return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), j) << 32) |
                   (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, j);
        };
        for (int k0i = kBeg; k0i < kEnd; k0i += 64) {
            const int kCnt = min(64, kEnd - k0i);
            const bool haveK = lane < kCnt;
            const unsigned kIdx = haveK ? K.features[k0i + lane] : 0u;
            const unsigned long long *qk = (const unsigned long long *)(K.desc + (size_t)kIdx * 32);
            const unsigned long long d0 = qk[0], d1 = qk[1], d2 = qk[2], d3 = qk[3];
            const int has = haveK ? (int)kfHasPoint[kIdx] : 0;
            for (int jj = 0; jj < kCnt; jj++) {
                const int j = __builtin_amdgcn_readfirstlane(jj);
                if (!__builtin_amdgcn_readlane(has, j)) continue;
                const int kf = __builtin_amdgcn_readlane((int)kIdx, j);
                const unsigned long long b0 = bcast64(d0, j), b1 = bcast64(d1, j), b2 = bcast64(d2, j), b3 = bcast64(d3, j);
                unsigned l0 = NONE, l1 = NONE, r0 = NONE, r1 = NONE;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const unsigned dist = (unsigned)(__popcll(b0 ^ f0[r]) + __popcll(b1 ^ f1[r]) + __popcll(b2 ^ f2[r]) + __popcll(b3 ^ f3[r]));
                    const unsigned key = (freeMask >> r & 1u) ? ((dist << 20) | (unsigned)(r * 64 + lane)) : NONE;
                    const bool right = rightMask >> r & 1u;
                    const unsigned keyL = right ? NONE : key, keyR = right ? key : NONE;
                    const unsigned largerL = max(keyL, l0), largerR = max(keyR, r0);
                    l0 = min(l0, keyL);
                    l1 = min(l1, largerL);
                    r0 = min(r0, keyR);
                    r1 = min(r1, largerR);
                }
                twoMin(l0, l1);
                if ((int)(l0 >> 20) > 50) continue;
                auto claim = [&](unsigned key) {
                    const int p = (int)(key & 0xfffffu);
                    if ((p & 63) == lane) {
#pragma unroll
                        for (int r = 0; r < R; r++)
                            if ((p >> 6) == r) {
                                mine[r] = kf;
                                freeMask &= ~(1u << r);
                            }
                    }
                };
                if (__fmul_rn(nnRatio, (float)(l1 >> 20)) > (float)(l0 >> 20)) claim(l0);
                if (nleft != -1) {
                    twoMin(r0, r1);
                    if ((int)(r0 >> 20) <= 50) claim(r0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++)
            if (mine[r] >= 0) matches[fIdx[r]] = mine[r];
        return;
    }
    for (int ik = kBeg; ik < kEnd; ik++) {
        const unsigned kfIdx = K.features[ik];
        if (!kfHasPoint[kfIdx]) continue;  // !pMP || pMP->isBad() (:347-351)
        const unsigned long long *dk = (const unsigned long long *)(K.desc + (size_t)kfIdx * 32);
        const unsigned long long k0 = dk[0], k1 = dk[1], k2 = dk[2], k3 = dk[3];
        unsigned l0 = NONE, l1 = NONE, r0 = NONE, r1 = NONE;
        for (int p = lane; p < fCnt; p += 64) {
            const unsigned fIdx = F.features[fBeg + p];
            if (__hip_atomic_load(matches + fIdx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0) continue;
            const unsigned long long *q = (const unsigned long long *)(F.desc + (size_t)fIdx * 32);
            const unsigned dist = (unsigned)(__popcll(k0 ^ q[0]) + __popcll(k1 ^ q[1]) + __popcll(k2 ^ q[2]) + __popcll(k3 ^ q[3]));
            const unsigned key = (dist << 20) | (unsigned)p;
            const bool right = nleft != -1 && (int)fIdx >= nleft;
            // both pairs take a key, one of them the neutral NONE (selects between values, not between the pairs' addresses)
            const unsigned keyL = right ? NONE : key, keyR = right ? key : NONE;
            const unsigned largerL = max(keyL, l0), largerR = max(keyR, r0);
            l0 = min(l0, keyL);
            l1 = min(l1, largerL);
            r0 = min(r0, keyR);
            r1 = min(r1, largerR);
        }
        twoMin(l0, l1);
        const int bestDist1 = (int)(l0 >> 20), bestDist2 = (int)(l1 >> 20);
        if (bestDist1 > 50) continue;  // TH_LOW (:424); the right camera is only looked at inside this branch (:451)
        const int kf = (int)kfIdx;
        if (__fmul_rn(nnRatio, (float)bestDist2) > (float)bestDist1) {  // (float)bestDist1 < mfNNratio * (float)bestDist2
            const int p = (int)(l0 & 0xfffffu);
            if ((p & 63) == lane) __hip_atomic_store(matches + F.features[fBeg + p], kf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (nleft != -1) {
            twoMin(r0, r1);
            if ((int)(r0 >> 20) <= 50) {  // the right camera's ratio test is disabled in the reference ("|| true", :453)
                const int p = (int)(r0 & 0xfffffu);
                if ((p & 63) == lane) __hip_atomic_store(matches + F.features[fBeg + p], kf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

}  // namespace

int ft_launch_bow_walk(hipStream_t st, const FtBowTree &t, const uint8_t *desc, int n, int nidLevel, unsigned *wordOut,
                       unsigned *nodeOut, double *weightOut) {
    if (n <= 0) return FT_OK;
    const int groupsPerBlock = 256 / 16;
    hipLaunchKernelGGL(k_bow_walk, dim3((n + groupsPerBlock - 1) / groupsPerBlock), dim3(256), 0, st, t, desc, n, nidLevel,
                       wordOut, nodeOut, weightOut);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_search_by_bow(hipStream_t st, const FtBowSide &K, const uint8_t *kfHasPoint, const FtBowSide &F, int nleft,
                            float nnRatio, int *matches) {
    if (K.nNodes <= 0 || F.nNodes <= 0) return FT_OK;
    k_search_by_bow<<<(K.nNodes + 3) / 4, 256, 0, st>>>(K, kfHasPoint, F, nleft, nnRatio, matches);
    return hipGetLastError() == hipSuccess ? FT_OK : FT_ERR_HIP;
}
