// Internal types shared by the host orchestration (.cpp) and the HIP kernels (.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/fasttrack_amd.h"

#define FT_MAX_LEVELS 12
#define FT_EDGE_THRESHOLD 19  // include/ORBextractor.h:31 of the reference
#define FT_PATCH_SIZE 31
#define FT_HALF_PATCH 15
#define FT_TH_HIGH 100  // src/ORBmatcher.cc:41
#define FT_TH_LOW 50
#define FT_HISTO_LENGTH 30
#define FT_GRID_COLS 64  // include/Frame.h:47
#define FT_GRID_ROWS 48

// Geometry of one pyramid level; identical for every image slot of an extractor.
struct FtLevelGeom {
    int w, h, pitch;   // level size and row pitch (bytes) inside the slot's pyramid buffer
    int off;           // byte offset of the level inside the slot's pyramid buffer (level 0 unused
                       // when the caller's frame is already in HBM)
    int maxBX, maxBY;  // w-16, h-16 (minBorder is always 16 = EDGE_THRESHOLD-3)
    int nCols, nRows, wCell, hCell;  // FAST cell grid (ORBextractor.cc:1128-1134)
    int cellBase;      // first cell of this level in the per-slot cell arrays
    int cellCap;       // staging entries per cell of this level
    int stageBase;     // entry offset of the level's first cell in the slot's staging buffer
    int candBase;      // entry offset of the level's dense candidate list
    int candCap;       // capacity of that list
    int xtab, ytab;    // offsets of the resize tables of this level (level >= 1)
    int area2x;        // level is an exact 2x decimation of the previous one (INTER_AREA path)
};

struct FtGeom {
    int nlevels;
    int totalCells;
    int stagePerSlot;  // staging entries per slot
    int candPerSlot;   // dense candidate entries per slot
    int pyrPerSlot;    // pyramid bytes per slot
    int maxKp;         // keypoint capacity per slot
    float sf[FT_MAX_LEVELS];     // mvScaleFactor
    float invsf[FT_MAX_LEVELS];  // mvInvScaleFactor
    unsigned fastLv[FT_MAX_LEVELS];  // k_fast_cells: row pitch | cellCap << 16 of the level (one scalar load)
    FtLevelGeom lv[FT_MAX_LEVELS];
};

// k_fast_cells: everything a wave needs to know about its cell, built with the extractor (one 16-byte scalar load where the
// kernel used to derive it from the level geometry with ~70 scalar instructions and a dozen dependent scalar loads)
struct FtCellRec {
    uint32_t origin;  // iniX | iniY << 16: top-left corner of the cell's sub-image (ORBextractor.cc:1138-1147)
    uint32_t shape;   // tile width | tile height << 8 | level << 16; width 0 = the reference skips the cell (:1141,1150)
    uint32_t srcOff;  // byte offset of pixel (iniX, iniY) in the slot's pyramid buffer (levels >= 1)
    uint32_t outOff;  // entry offset of the cell's staging slot in the slot's staging buffer
};

// candidate / keypoint packing: x | y << 12 | score << 24 (x, y < 4096; score <= 254)
static inline __host__ __device__ uint32_t ft_pack_cand(int x, int y, int s) {
    return (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)s << 24);
}

// Image -> XCD placement for kernels that read patches of an image's pyramid (k_orient_desc, k_stereo_match).
// Workgroup b of a 1-D launch lands on XCD b % 8 (round-robin placement of the dispatcher, used for speed only:
// a different placement changes nothing but the hit rate).  With the plain (blocksPerSlot, batch) grid every
// XCD touches every image, so each of the eight private 4 MB L2s tries to hold all pyramids of the launch; here
// all workgroups of an image run on one XCD and an XCD walks its images one after the other, so the L2 holds
// the ~3 MB pyramid it is currently sampling.  Launches with fewer than 8 images keep the plain grid (all XCDs
// busy on one image).  Returns false for the padding workgroups.
struct FtSlotGrid {
    int blocksPerSlot, batch, xcdMap;
    unsigned magic;  // floor(2^32 / blocksPerSlot) + 1, 0 for blocksPerSlot == 1
};
static inline FtSlotGrid ft_slot_grid(int blocksPerSlot, int batch, dim3 &grid) {
    FtSlotGrid sg;
    sg.blocksPerSlot = blocksPerSlot;
    sg.batch = batch;
    sg.xcdMap = batch >= 8;
    sg.magic = blocksPerSlot > 1 ? 0xffffffffu / (unsigned)blocksPerSlot + 1u : 0u;
    grid = sg.xcdMap ? dim3((unsigned)blocksPerSlot * 8u * (unsigned)((batch + 7) / 8), 1, 1) : dim3(blocksPerSlot, batch, 1);
    return sg;
}
#if defined(__HIPCC__)
__device__ __forceinline__ bool ft_slot_block(const FtSlotGrid &sg, int &slot, int &blk) {
    if (!sg.xcdMap) {
        slot = blockIdx.y;
        blk = blockIdx.x;
        return true;
    }
    const int xcd = (int)(blockIdx.x & 7u), j = (int)(blockIdx.x >> 3);
    const int grp = sg.magic ? (int)__umulhi((unsigned)j, sg.magic) : j;
    blk = j - grp * sg.blocksPerSlot;
    slot = grp * 8 + xcd;
    return slot < sg.batch;
}
#endif

// selected keypoint handed from the host octree to the orientation/descriptor kernel
struct FtSelKp {
    short x, y;  // level coordinates (border included)
    short level;
    short response;
};

// resize tap: source index and the two 11-bit fixed-point weights
struct FtTap {
    short s, a0, a1, pad;
};

// device octree (kernels_octree.hip): a level with up to FT_OCT_MAXN candidates is sorted in the LDS layout of the
// main kernel, one with up to bigN (16384 when the node pools leave room, see ft_octree_big_keys) by the second-tier
// kernel that runs behind it; an image with a level beyond that is redone with the host octree (that image only).
// The per-level quota is bounded by the LDS the node pools need (ft_octree_smem_bytes <= 160 KB).
#define FT_OCT_MAXN 4096
#define FT_OCT_HIST_BINS 8192  // bins of the histogram tier (k_octree_hist): nodes of depth D, nIni * 4^D <= this
#define FT_OCT_HISTMIN 16      // workgroups of the histogram tier while no batch has asked for more (each walks the list)
#define FT_OCT_HIST_RETIRE 2   // batches without a listed level after which the histogram tier is not launched any more
#ifndef FT_OCT_HISTMAX
#define FT_OCT_HISTMAX 768     // ... and at most (three fit a CU)
#endif
#define FT_OCT_BIGMIN 64   // smallest grid of the sorted big tier (k_octree_big) once a stream of frames needs it
struct FtOctArgs {
    const uint32_t *cand;   // device dense candidate lists [slot * candPerSlot + candBase]
    const int *candCount;   // device [slot * nlevels + level]
    FtSelKp *sel;           // device [slot * maxKp + selOff[level] + i]
    int *selCount;          // device [slot * nlevels + level]
    int *overflow;          // device flag: some image of the launch needs the host octree
    int *ovSlot;            // device [batch]: which ones
    // Levels with more than FT_OCT_MAXN candidates.  bigCount (per octree stream): [0] (slot, level) pairs listed for the
    // histogram tier by the running launch, [1] pairs that tier handed on to the sorted big tier, [2] / [3] the maxima of
    // [0] / [1] since the host last looked.  histCap / sortCap: capacity of the lists = grids of
    // k_octree_big (0 = tier not launched; k_octree_hist walks its list with histGrid workgroups); bigN: keys per
    // workgroup of k_octree_big.
    int *bigCount;
    unsigned *bigList, *sortList;
    // first sorted tier, compact LDS layout (kernels_octree.hip oct_lds_layout): FT_OCT_MAXN dwords of scratch per (image,
    // level) of the launch, [(slot * nlevels + level) * FT_OCT_MAXN]; null = the plain layout
    uint32_t *low;
    int histCap, histGrid, sortCap, bigN;  // histGrid: workgroups of k_octree_hist (they walk the list)
    int histWanted;                        // the histogram tier exists but is not launched now: count its demand all the same
    int histFirst;                         // the histogram formulation runs every level (k_octree_hist_all); give-ups -> k_octree_big
    int quota[FT_MAX_LEVELS], levelMax[FT_MAX_LEVELS], selOff[FT_MAX_LEVELS];
    int poolCap, keyBytes;
    unsigned long long *prof;  // FT_OCT_PROFILE=1: per-level phase times of slot 0 (wall_clock64 ticks), else null
};

// vocabulary tree on the device (kernels_bow.hip): children of node i are childList[childStart[i] .. childStart[i+1])
// in ascending node id; a node without children is a word
struct FtBowTree {
    const int *childStart;      // [nNodes + 1]
    const unsigned *childList;  // [nNodes - 1]
    const uint8_t *desc;        // [nNodes * 32]
    const unsigned *wordId;     // [nNodes] (valid for leaves)
    const double *weight;       // [nNodes]
    int nNodes;
};
int ft_launch_bow_walk(hipStream_t st, const FtBowTree &t, const uint8_t *desc, int n, int nidLevel, unsigned *wordOut,
                       unsigned *nodeOut, double *weightOut);
// one side of ORBmatcher::SearchByBoW on the device: a DBoW2 FeatureVector in CSR form and the descriptors it indexes
struct FtBowSide {
    int n, nNodes;
    const unsigned *nodes;     // [nNodes] ascending
    const int *offsets;        // [nNodes + 1]
    const unsigned *features;  // [offsets[nNodes]]
    const uint8_t *desc;       // [n * 32]
};
int ft_launch_search_by_bow(hipStream_t st, const FtBowSide &K, const uint8_t *kfHasPoint, const FtBowSide &F, int nleft,
                            float nnRatio, int *matches);

void ft_set_error(const std::string &msg);
// Profiling aid: FT_DEBUG_REPEAT=<name>[,<name>...] makes the launcher of that kernel (pyr, fast, compact, octree,
// orient, rowsort, stereo, median) enqueue it twice.  Every kernel is idempotent, so results do not change; the
// increase of the step time is the kernel's marginal cost inside the overlapped pipeline, which neither the
// per-launch event times nor the rocprof durations (both inflated by concurrency) show.
int ft_debug_repeat(const char *name);
int ft_hip_fail(hipError_t e, const char *what, const char *file, int line);
const char *ft_debug_env(const char *name);  // FT_DEBUG_* aids only (context.cpp)

#define FT_HIP(call)                                                              \
    do {                                                                          \
        hipError_t _e = (call);                                                   \
        if (_e != hipSuccess) return ft_hip_fail(_e, #call, __FILE__, __LINE__);  \
    } while (0)

// ---- kernel launchers (kernels_extract.hip) -------------------------------------------------
// l0: device array [batch] of level-0 pointers; pyr: base of the slot pyramids
int ft_launch_pyramid(hipStream_t st, const FtGeom &g, int batch, const uint8_t *const *l0, int l0pitch,
                      uint8_t *pyr, const FtTap *taps, int alignedLoads, int rowsKernel);
// the cells a k_fast_cells launch covers: up to two runs of consecutive cells of the per-image cell table
struct FtCellRanges {
    int lo0, n0, lo1, n1;
};
int ft_launch_fast_cells(hipStream_t st, const FtGeom &g, int batch, const uint8_t *const *l0, int l0pitch,
                         const uint8_t *pyr, int iniTh, int minTh, int alignedLoads, int *cellCount,
                         uint32_t *stage, int ordered, const FtCellRec *cellTab);
int ft_launch_compact(hipStream_t st, const FtGeom &g, int batch, const int *cellCount, const uint32_t *stage,
                      uint32_t *cand, int *candCount);
// sel is laid out per level (slot * maxKp + selOff[level] + i) with per-level counts; the kernel packs the
// results in level order (slot * maxKp + k) and stores the per-image total in nSel
int ft_launch_orient_desc(hipStream_t st, const FtGeom &g, int batch, const uint8_t *const *l0, int l0pitch,
                          const uint8_t *pyr, int alignedLoads, const FtSelKp *sel, const int *selCount,
                          const FtOctArgs &layout, int *nSel, ft_keypoint *keys, uint8_t *desc);
int ft_launch_octree(hipStream_t st, const FtGeom &g, int batch, const FtOctArgs &a);
// test tap: 7x7 Gaussian of a whole level through k_orient_desc's blur routines (dst on the device)
int ft_launch_blur_level(hipStream_t st, const uint8_t *img, int pitch, int w, int h, uint8_t *dst, int dstPitch);
size_t ft_octree_smem_bytes(int poolCap, bool compact = false);
size_t ft_octree_hist_smem_bytes(int poolCap);
int ft_octree_big_keys(int poolCap);
size_t ft_fast_smem_bytes(const FtGeom &g);

// ---- kernel launchers (kernels_match.hip) ---------------------------------------------------
// A right keypoint as k_stereo_rowsort lays it down in row-bucket order: everything the left keypoint's scan needs in
// one 48-byte entry, so the scan is one coalesced read instead of the chain index -> keypoint -> descriptor.
struct __attribute__((aligned(16))) FtSortedR {
    float x, y;
    int octave, idx;  // idx = position in the caller's right keypoint array
    unsigned long long desc[4];
};
struct FtStereoArgs {
    const ft_keypoint *keysL, *keysR;  // device
    const uint8_t *descL, *descR;      // device, n x 32
    const int *nL, *nR;                // device [batch]
    int capacity;                      // stride (entries) between images in the arrays above
    float mbf, mb;
    float *uright, *depth;             // device [batch*capacity]
    int *sad;                          // device [batch*capacity]
    int *hamIdx;                       // device [batch*capacity] (debug tap, may be null)
    int *nMatches;                     // device [batch]
    int applyMedianCut;
    int *rowStart;                     // device [batch * rowStride]: right keypoints bucketed by (int)y
    FtSortedR *sorted;                 // device [batch * capacity]: right keypoints in bucket order
    int rowStride;                     // >= level-0 height + 2
    int alignedLoads;                  // level-0 frames of both cameras are 4-byte aligned (base and stride): dword reads of image rows
};
// Result delivery of a small batch (latency mode): one kernel writes everything the host needs - keypoints and
// descriptors of both cameras, mvuRight, mvDepth and the counters - straight into pinned host memory, instead of
// eleven small copies through the DMA queue (each a few microseconds of work behind ~10 us of latency).
struct FtDeliverArgs {
    const ft_keypoint *keysL, *keysR;  // device, rows of srcStride records per image
    const uint8_t *descL, *descR;
    const float *uright, *depth;
    const int *nL, *nR, *nMatches;      // device [batch]
    const int *overflowL, *overflowR;   // device flags of the two extractors
    ft_keypoint *oKeysL, *oKeysR;       // pinned host (any of the row destinations may be null), rows of dstStride records
    uint8_t *oDescL, *oDescR;
    float *oUright, *oDepth;
    int *oNL, *oNR, *oNMatches, *oOverflowL, *oOverflowR;
    int srcStride, dstStride;
};
int ft_launch_deliver(hipStream_t st, int batch, const FtDeliverArgs &a);
// ft_extract_batch's results of slots [b0, b0 + nb) in the reference's output order, into the caller's pinned arrays (k_deliver_ordered)
struct FtOrderedArgs {
    const ft_keypoint *keys;  // device, rows of srcStride records per slot
    const uint8_t *desc;
    const int *nSel;          // device [slots]
    ft_keypoint *oKeys;       // pinned host, rows of `capacity` records per slot (may be null)
    uint8_t *oDesc;           // (may be null)
    int *oMono;               // pinned host [slots]: keypoints outside the lapping area
    int srcStride, capacity, b0;
    float lap0, lap1;
};
int ft_launch_deliver_ordered(hipStream_t st, int nb, const FtOrderedArgs &a);
// the end-of-batch counters of an extractor into pinned host memory (k_finish_counts)
struct FtCountsArgs {
    const int *nSel;      // device [batch]
    int *oNSel;           // pinned host [batch]
    const int *overflow;  // device flag
    int *oOverflow;       // pinned host
    int *bigCount;        // device: per octree stream k the demand words [4 k + 2], [4 k + 3] (read and zeroed); null = none
    int *oHist, *oBig;    // pinned host [nStreams]
    int batch, nStreams;
};
int ft_launch_finish_counts(hipStream_t st, const FtCountsArgs &a);
// Frame upload of a small batch (latency mode): one kernel reads the frames from pinned host memory
// and writes them as level 0 of the slot pyramids (row pitch `pitch`, `slotBytes` apart), and fills
// the level-0 pointer table - instead of one DMA copy per frame plus one for the table.
// Each frame is described by an entry of a table in pinned host memory that the kernel reads when it runs (so a captured
// launch follows the frames of every replay): a frame the caller keeps in pinned memory is read where it is, any other
// frame from the library's pinned staging copy.
struct FtSrcEntry {
    const uint8_t *ptr;
    long long stride;
};
int ft_launch_upload(hipStream_t st, int batch, const FtSrcEntry *srcTab, int width, int height, uint8_t *slot0, int pitch,
                     size_t slotBytes, const uint8_t **l0Table);
struct FtFisheyeRig {
    float cam1[8], cam2[8], precision, Rlr[9], tlr[3];
    float sigma2[FT_MAX_LEVELS];
};
// KannalaBrandt8::TriangulateMatches for every left keypoint with a 2-NN match (matches[i] >= 0); rejected pairs
// get matches[i] = -1; nMatches counts the survivors
int ft_launch_fisheye_triangulate(hipStream_t st, const FtFisheyeRig &rig, const ft_keypoint *keysL, int nL,
                                  const ft_keypoint *keysR, int *matches, float *depth, float *p3d, int *nMatches);
int ft_launch_stereo_match(hipStream_t st, const FtGeom &g, int batch, const uint8_t *const *l0L,
                           const uint8_t *const *l0R, int l0pitchL, int l0pitchR, const uint8_t *pyrL,
                           const uint8_t *pyrR, const FtStereoArgs &a);
int ft_launch_stereo_median(hipStream_t st, int batch, const FtStereoArgs &a);
int ft_launch_stereo_rowsort(hipStream_t st, const FtGeom &g, int batch, const FtStereoArgs &a);
int ft_launch_fisheye(hipStream_t st, const uint8_t *descL, int nL, const uint8_t *descR, int nR, int *matches,
                      int *best, int *second);
int ft_launch_hamming_pairs(hipStream_t st, const uint8_t *a, const uint8_t *b, int n, int *dist);
