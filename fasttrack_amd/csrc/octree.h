// Host octree keypoint distribution - the serial stage that stays on the host in both branches of the
// reference (SURVEY.md section 3 "Hot loops", section 8 row a5).  Semantics of
// ORBextractor::DistributeOctTree / ExtractorNode::DivideNode / compareNodes
// (reference src/ORBextractor.cc:660-884, :510-566, :626-641): same node visiting order, same
// front-insertion order, same std::sort call on the same (size, UL.x) sequence, same first-maximum
// pick - so the retained keypoints AND their output order are identical.
//
// Implementation: every node owns a contiguous sub-range of a key array (packed candidate + original
// index) that is stably 4-way partitioned into a second array when the node splits (ping-pong, no
// indirection), and the std::list of the reference becomes an intrusive doubly linked list over a pool.
#pragma once

#include <stdint.h>

#include <utility>
#include <vector>

namespace ft {

struct OctreeWorkspace {
    struct Node {
        int x0, y0, x1, y1;  // UL = (x0,y0), BR = (x1,y1)
        int begin, end;      // candidate sub-range in perm
        int prev, next;      // list links (-1 = none)
        bool noMore;
        uint8_t buf;         // which key buffer holds the range (keys ping-pong on every split)
    };
    std::vector<Node> pool;
    std::vector<uint64_t> keysA, keysB;  // candidate | index << 32
    std::vector<uint8_t> quad;
    std::vector<int> scratch;
    std::vector<std::pair<int, int>> sizeAndNode, prevSizeAndNode;  // (size, node index)
};

// cand: n packed candidates (x | y<<12 | score<<24), coordinates relative to (minX, minY), in the
// emission order of the FAST stage.  Appends the indices of the retained candidates to `out` in the
// reference's result order and returns how many were retained.
int distribute_octree(const uint32_t *cand, int n, int minX, int maxX, int minY, int maxY, int N,
                      OctreeWorkspace &ws, std::vector<int> &out);

// the same result through the path-code formulation of octree_paths.h (what the device kernel runs),
// single-threaded on the host; returns -1 when n or N exceed what the 16-bit node records hold
int distribute_octree_paths(const uint32_t *cand, int n, int minX, int maxX, int minY, int maxY, int N,
                            std::vector<int> &out);

// the same result through the round formulation (arrays + prefix sums, see octree.cpp) that the device
// kernel executes with one wave per level
int distribute_octree_rounds(const uint32_t *cand, int n, int minX, int maxX, int minY, int maxY, int N,
                             std::vector<int> &out);

// the round formulation over a HISTOGRAM of the candidates instead of sorted keys (what k_octree_hist executes, see
// octree.cpp): -2 when a node deeper than the histogram (nIni * 4^D <= maxBins bins) would have to be split
int distribute_octree_hist(const uint32_t *cand, int n, int minX, int maxX, int minY, int maxY, int N, int maxBins,
                           std::vector<int> &out);

// upper bound of what distribute_octree can return for a level (used to size output buffers)
int octree_max_result(int minX, int maxX, int minY, int maxY, int N);

}  // namespace ft
