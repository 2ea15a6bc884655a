// HIP kernel of the bag-of-words transform for gfx950 (wave64):
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
