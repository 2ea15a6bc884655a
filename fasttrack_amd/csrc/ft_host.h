// Host-side object definitions behind the opaque C handles.
#pragma once

#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "ft_internal.h"
#include "octree.h"
#include "thread_pool.h"

#define FT_PIPE_MAX 8
// Streams of an extractor: stage A, stage B and FT_OCT_STREAMS octree streams (a long, thin k_octree never sits in front
// of a wide kernel of the same stream; measured on MI355X, default bench: 1 octree stream 32.3 k frames/s, 2: 37.4 k,
// 4: 35.5 k).  The streams belong to the CONTEXT, which keeps FT_LANE_SETS sets of them ("lanes") and hands the sets to
// the extractors in turn: the left and the right extractor of a stereo front end get different sets, the second front end
// of a double-buffered caller gets the same two again.  Why: the runtime multiplexes streams onto a few in-order hardware
// queues in an order of its own (least used queue at first use), and which streams end up sharing a queue decides the
// throughput - with 16 private streams the default bench ran 49 k to 71 k frames/s depending on GPU_MAX_HW_QUEUES (5 .. 16).
// Which stream goes to which lane is a table (ft_context::laneMap, FT_LANE_MAP) found by a search on the default bench
// (tools/lane_search.py: random maps 45 - 66 k frames/s, hill climbing from the best 73.4 k).  Extractors for small batches
// (latency mode) keep streams of their own: their batches are captured as HIP graphs, and a stream under capture cannot be
// shared with another host thread.
// Size classes of the shipped tables: frames of fewer pixels than this take laneMapSmall.  (What the frames CONTAIN - dense
// texture, host-in uploads - also shifts the balance, but is not known when an extractor is created: an application that
// knows its stream searches a table on it with tools/lane_search.py and sets it with ft_context_set_lane_map / FT_LANE_MAP.)
#define FT_LANE_SMALL_PIXELS ((size_t)600 * 1000)
#define FT_LANE_MAP_SMALL {1, 2, 3, 4, 2, 1, 3, 1, 5, 4, 3, 2, 1, 1, 0, 1}  // 752x480: 151.6 k frames/s against 147.6 k with the 1280x720 table; climbed again late in round 3: 158.4 against 156.0 k; round 4: the last entry moved, 164.5 against 163.2 k
#define FT_OCT_STREAMS 2
#define FT_LANE_STREAMS (2 + FT_OCT_STREAMS)

// ---- tuning options --------------------------------------------------------------------------------------------------
// Every switch of the library that is not a debugging aid, in ONE table: name (as ft_context_set_option takes it), the
// environment variable that sets its initial value, the default, and what it does.  ft_context_create reads the environment
// ONCE into ft_context::tuning; ft_context_set_option changes the context's copy; extractors, front ends and tracked frames
// take their switches from the context when they are CREATED (an option set later applies to objects created later; the
// search switches are read per call).  Nothing else in the library reads FT_* variables except the FT_DEBUG_* aids
// (ft_debug_env, context.cpp) and FT_LANE_MAP (a list, ft_context_create).
#define FT_TUNING_OPTIONS(X)                                                                                                  \
    X(pipeline_depth, "FT_PIPELINE_DEPTH", 0, 0, 8, "sub-batches a throughput batch is enqueued as (0 = automatic, 1 = no pipelining, <= 8)") \
    X(device_octree, "FT_DEVICE_OCTREE", 1, 0, 1, "DistributeOctTree on the device (0 = host thread pool)")                  \
    X(oct_hist, "FT_OCT_HIST", 1, 0, 1, "histogram tier of the device octree for levels above 4096 candidates")              \
    X(oct_hist_first, "FT_OCT_HIST_FIRST", 1, 0, 2, "histogram formulation for every level: 0 never, 1 latency-mode launches, 2 always") \
    X(oct_big, "FT_OCT_BIG", 1, 0, 1, "sorted big tier of the device octree (up to 16384 keys per level)")                   \
    X(pyr_rows, "FT_PYR_ROWS", 1, 0, 1, "pyramid of launches of 8+ images: 1 = row-streaming kernel (k_pyr_rows), 0 = tile kernel (k_pyr_down)") \
    X(upload_kernel, "FT_UPLOAD_KERNEL", 1, 0, 1, "latency mode: frames go up through k_upload instead of DMA copies")       \
    X(graph, "FT_GRAPH", 1, 0, 1, "latency mode: batches of <= 8 frames are captured and replayed as HIP graphs")            \
    X(paired, "FT_PAIRED", 1, 0, 1, "latency-mode stereo front ends run both cameras through one set of launches")           \
    X(pass_burst, "FT_PASS_BURST", 12, 2, 14, "projection searches: claim passes enqueued per host round trip")              \
    X(search_cache, "FT_SEARCH_CACHE", 2, 0, 3, "projection searches: 1 = later claim passes walk the cached candidate keys, 2 = and a batch of 24+ frames resolves its claims in one launch, 3 = every batch does") \
    X(search_grid, "FT_SEARCH_GRID", 1, 0, 1, "projection searches: CSR grid of the frame built on the device") \
    X(blocking_sync, "FT_BLOCKING_SYNC", 2, 0, 2, "host waits for the device: 0 = the runtime's default (it spins), 1 = sleeping (hipDeviceScheduleBlockingSync), 2 = sleeping when the process may use fewer than 8 CPUs")

struct ft_tuning {
#define FT_X(field, env, def, lo, hi, doc) int field = def;
    FT_TUNING_OPTIONS(FT_X)
#undef FT_X
};
bool ft_tuning_from_env(ft_tuning &t, std::string &err);             // defaults overridden by the FT_* variables that are set; false: a value out of range
int *ft_tuning_field(ft_tuning &t, const char *name);                // by option name or by environment name; null = unknown
const char *ft_debug_env(const char *name);                          // FT_DEBUG_* aids: the one other place that reads the environment
const char *ft_read_env(const char *name);                           // context.cpp only: the single getenv of the library

struct ft_context {
    int device = 0;
    std::string deviceName;
    ft::ThreadPool *pool = nullptr;
    hipStream_t stream = nullptr;  // context-level stream for the stand-alone matchers
    // lanes[set * FT_LANE_STREAMS + {0: stage A, 1: stage B, 2..: octree}], created on demand; laneSets = 0: every extractor
    // creates (and owns) its streams
    std::mutex laneMutex;
    std::vector<hipStream_t> lanes;
    hipStream_t uploadStream = nullptr;  // host frames of whole batches go up here (copies only)
    std::vector<int> laneMap;  // [extractor k mod sets][role] -> lane; empty: private streams
    std::vector<int> laneMapSmall;  // the same for frames below FT_LANE_SMALL_PIXELS (a table of its own: other kernel balance)
    bool userLaneMap = false;       // laneMap came from FT_LANE_MAP / ft_context_set_lane_map: it serves every shape
    int nextLaneSet = 0;
    size_t maxPitch = 0;                // hipDeviceAttributeMaxPitch (0 = unknown): bound of the one-copy upload's source pitch
    int hwQueues = 4;                   // GPU_MAX_HW_QUEUES of the process environment when the context was created (4 = unset)
    std::atomic<int> liveObjects{0};    // extractors (incl. those of front ends) alive on this context: they hold its lanes
    // grow-only scratch of the stand-alone matchers (one call at a time per context)
    std::mutex matchMutex;
    void *scratchDev = nullptr, *scratchPin = nullptr;
    size_t scratchDevBytes = 0, scratchPinBytes = 0;
    std::mutex hostAllocMutex;
    // events of destroyed tracked batches that an extractor may still hold as its foreignReader: destroyed with the context
    std::vector<hipEvent_t> retiredEvents;
    std::map<uintptr_t, size_t> hostBlocks;  // address -> bytes of the live ft_host_malloc blocks (ft_host_block_contains)
    std::vector<void *> hostAllocs;  // live ft_host_malloc blocks
    bool kernelTiming = false;  // ft_context_set_kernel_timing
    ft_tuning tuning;           // FT_TUNING_OPTIONS: read from the environment by ft_context_create, changed by ft_context_set_option
    std::mutex statsMutex;
    std::map<std::string, std::pair<double, long>> stats;  // name -> (total ms, calls)
    void addStat(const char *name, double ms) {
        std::lock_guard<std::mutex> lk(statsMutex);
        auto &s = stats[name];
        s.first += ms;
        s.second += 1;
    }
};

// HIP-event timing of kernel groups on the stream they are launched on (bench.py's roofline leg).
// begin()/end() bracket launches; resolve() is called after the stream has been synchronised.
struct FtEventTimer {
    struct Rec {
        const char *name;
        hipEvent_t a, b;
    };
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    std::vector<Rec> recs;
    hipEvent_t get() {
        if (used == pool.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            pool.push_back(e);
        }
        return pool[used++];
    }
    void begin(bool enabled, const char *name, hipStream_t st) {
        if (!enabled) return;
        Rec r{name, get(), get()};
        if (!r.a || !r.b) return;
        hipEventRecord(r.a, st);
        recs.push_back(r);
    }
    void end(bool enabled, hipStream_t st) {
        if (!enabled || recs.empty()) return;
        hipEventRecord(recs.back().b, st);
    }
    void resolve(ft_context *ctx) {
        for (auto &r : recs) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) ctx->addStat(r.name, ms);
        }
        recs.clear();
        used = 0;
    }
    void destroy() {
        for (auto e : pool) hipEventDestroy(e);
        pool.clear();
    }
};

struct FtTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double ms() const {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
};

struct ft_extractor {
    ft_context *ctx = nullptr;
    ft_tuning tune;  // copy of ctx->tuning taken when the extractor was created
    int nfeatures = 0, nlevels = 0, iniTh = 0, minTh = 0, width = 0, height = 0, maxBatch = 0;
    float scaleFactor = 1.2f;
    std::vector<float> sf, invsf, sigma2, invsigma2;
    std::vector<int> quota;
    std::vector<int> levelMax;  // per-level bound of octree results
    std::vector<int> levelOff;  // prefix sums of levelMax
    FtGeom geom{};
    hipStream_t stream = nullptr;   // stage A (pyramid, FAST, compaction)
    hipStream_t streamB = nullptr;  // stage B (orientation + descriptors), matching, result copies
    hipEvent_t evA[FT_PIPE_MAX] = {}, evB[FT_PIPE_MAX] = {};
    // the device octree is a long, narrow kernel (one wave per level and image): it runs on its own streams
    // so that the wide stage-A kernels of the following sub-batches are not queued behind it
    hipStream_t streamO[FT_OCT_STREAMS] = {};
    bool ownStreams = false;  // false: the streams are lanes of the context
    hipEvent_t evO[FT_PIPE_MAX] = {};
    // latency mode of ft_extract / ft_extract_batch: a small batch with a fixed call shape is captured once as a HIP
    // graph (see ft_stereo_frontend::GraphKey); the key is everything baked into the nodes
    struct GraphKey {
        int batch = 0, onDevice = 0, width = 0, height = 0, stride = 0, aligned = 0, bigGrid = 0;
        bool operator==(const GraphKey &o) const {
            return batch == o.batch && onDevice == o.onDevice && width == o.width && height == o.height && stride == o.stride &&
                   aligned == o.aligned && bigGrid == o.bigGrid;
        }
    } graphKey;
    hipGraphExec_t graphExec = nullptr;
    bool graphDisabled = false;
    hipEvent_t evJoin = nullptr;
    hipEvent_t evUp = nullptr;  // the batch of host frames has arrived (recorded on the upload stream of the context)
    // somebody else's kernel still reads this extractor's keypoint / descriptor slots (ft_tracked_batch_bind_fisheye's gather on the
    // batch's stream): recorded there, waited for by the extractor's next batch before it overwrites them (ft_extract_foreign_wait)
    hipEvent_t foreignReader = nullptr;
    // device buffers
    uint8_t *d_pyr = nullptr;
    FtTap *d_taps = nullptr;
    FtCellRec *d_cellTab = nullptr;  // per FAST cell: origin, tile shape, level, source and staging offsets
    int *d_cellCount = nullptr;
    uint32_t *d_stage = nullptr;
    const uint8_t **d_l0 = nullptr;
    const uint8_t **h_l0 = nullptr;  // pinned
    uint8_t *h_stage = nullptr;      // pinned staging of host frames for the graph path (up to 16 slots, allocated on first use)
    FtSrcEntry *h_srcTab = nullptr;  // pinned: where k_upload finds each frame (the caller's pinned memory or h_stage)
    bool stageHost = false;          // ft_extract_prepare: host frames go through h_stage (their pointers may change between replays)
    float *d_sf = nullptr;           // scale factors on device [nlevels] then inverse [nlevels]
    // host-mapped pinned buffers written by the device
    uint32_t *h_cand = nullptr, *d_cand = nullptr;
    int *h_candCount = nullptr, *d_candCount = nullptr;
    // device octree (kernels_octree.hip): candidates, selection and counts never leave the device
    bool deviceOctree = false;
    uint32_t *d_candDev = nullptr;
    int *d_candCountDev = nullptr;
    int *d_selCount = nullptr, *h_selCount = nullptr;  // [maxBatch * nlevels]
    int *d_overflow = nullptr, *h_overflow = nullptr;
    int *d_ovSlot = nullptr;         // [maxBatch] which slots overflowed (read only after the summary flag was seen)
    int *d_bigCount = nullptr;       // octree tiers beyond FT_OCT_MAXN candidates, per octree stream: FtOctArgs::bigCount (4 ints)
    int *h_bigStat = nullptr;        // pinned copies of the sorted big tier's demand ([3])
    unsigned *d_bigList = nullptr;   // per octree stream [maxBatch * nlevels] (slot, level) pairs for the histogram tier
    unsigned *d_sortList = nullptr;  // ... and the pairs that tier handed on to k_octree_big
    uint32_t *d_octLow = nullptr;    // low key dwords of the first sorted tier while its rounds run (FtOctArgs::low)
    uint32_t *h_repCand = nullptr;   // pinned candidate lists of the slots under per-image repair (repCap slots, grow-only)
    int repCap = 0;
    bool histEnabled = false;        // k_octree_hist available (FT_OCT_HIST=0 switches it off)
    int histFirstMode = 1;           // FT_OCT_HIST_FIRST: 0 never, 1 (default) in latency-mode launches, 2 always
    bool histOn = false;             // latency mode: launched since a frame overflowed the first tier (large batches: always)
    int histGrid = FT_OCT_HISTMIN;   // workgroups of k_octree_hist for large batches: by the demand of the previous ones
    int histIdle = 0;
    int *h_histStat = nullptr;       // pinned copies of the histogram tier's demand ([1])
    int bigGrid = 0, bigIdle = 0;    // grid of k_octree_big: 0 until the histogram tier hands levels on, sized by that demand
    FtOctArgs octLayout{};
    // selected keypoints host -> device
    FtSelKp *h_sel = nullptr, *d_sel = nullptr;
    int *h_nSel = nullptr, *d_nSel = nullptr;
    int *h_nMono = nullptr;  // pinned: keypoints outside the lapping area per slot (k_deliver_ordered)
    // results device -> host
    ft_keypoint *d_keys = nullptr, *h_keys = nullptr;
    uint8_t *d_desc = nullptr, *h_desc = nullptr;
    // scratch of the host-array stereo API (ft_stereo_match), allocated on first use
    ft_keypoint *d_stKeys = nullptr;  // [2 * maxKp] left then right
    uint8_t *d_stDesc = nullptr;      // [2 * maxKp * 32]
    float *d_stOut = nullptr;         // [2 * maxKp] uright then depth
    int *d_stInt = nullptr;           // [2 * stCap + 4] sad, hamming idx, then nL nR nMatches, then the row starts
    FtSortedR *d_stSorted = nullptr;  // [stCap] right keypoints in row-bucket order
    int stCap = 0;
    FtEventTimer evt;
    // state of the last call (consumed by the stereo matcher)
    int lastBatch = 0;
    int l0pitch = 0;
    bool l0External = false;
    bool l0Aligned = true;
};

struct ft_stereo_frontend {
    ft_context *ctx = nullptr;
    ft_extractor *exL = nullptr, *exR = nullptr;
    float mbf = 0, mb = 0;
    int capacity = 0;
    float *d_uright = nullptr, *d_depth = nullptr, *h_uright = nullptr, *h_depth = nullptr;
    int *d_sad = nullptr, *d_nMatches = nullptr, *h_nMatches = nullptr;
    // Latency front ends (max_batch <= 8): the left extractor is created with room for 2 x max_batch images and a small
    // batch runs PAIRED - left frames in its slots [0, B), right frames in [B, 2B) - so that every kernel of the
    // extraction is launched once for both cameras (half the launches; the two cameras no longer queue behind each other
    // in the graph's launch order).  The right extractor then only lends its pinned host buffers to the results.
    int maxBatch = 0;
    bool pairedCapable = false, lastPaired = false;
    FtSortedR *d_sorted = nullptr;  // right keypoints bucketed by row (k_stereo_rowsort)
    int *d_rowStart = nullptr;
    hipEvent_t evR = nullptr;
    // Latency mode: a small batch with a fixed call shape (same sizes, same result arrays, same kind of input) is
    // captured once as a HIP graph - ~45 enqueue calls on seven streams become one launch.  graphKey holds everything
    // that is baked into the captured nodes.
    struct GraphKey {
        int batch = 0, onDevice = 0, width = 0, height = 0, stride = 0, capacity = 0, alignedL = 0, alignedR = 0, paired = 0;
        int bigGridL = 0, bigGridR = 0;
        const void *out[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        bool operator==(const GraphKey &o) const {
            if (batch != o.batch || onDevice != o.onDevice || width != o.width || height != o.height || stride != o.stride ||
                capacity != o.capacity || alignedL != o.alignedL || alignedR != o.alignedR || paired != o.paired ||
                bigGridL != o.bigGridL || bigGridR != o.bigGridR)
                return false;
            for (int i = 0; i < 6; i++)
                if (out[i] != o.out[i]) return false;
            return true;
        }
    } graphKey;
    hipGraphExec_t graphExec = nullptr;
    bool graphDisabled = false;
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    hipEvent_t evDone[2] = {nullptr, nullptr};  // end of the submitted batch on the two stage-B streams
    struct Pending {  // batch enqueued by ft_stereo_frontend_submit, finished by ft_stereo_frontend_wait
        bool active = false, direct = false, graph = false;
        int batch = 0, capacity = 0;
        std::vector<const uint8_t *> imagesL, imagesR;  // kept for the host-octree fallback
        int onDevice = 0, width = 0, height = 0, stride = 0;
        ft_keypoint *keysL = nullptr, *keysR = nullptr;
        uint8_t *descL = nullptr, *descR = nullptr;
        int *nL = nullptr, *nR = nullptr, *nMatches = nullptr;
        float *uright = nullptr, *depth = nullptr;
    } pending;
};

int ft_set_device(const ft_context *ctx);
int ft_context_take_lanes(ft_context *ctx, bool wantPrivate, int width, int height, hipStream_t *out, bool *owned);  // out[FT_LANE_STREAMS]
int ft_context_upload_stream(ft_context *ctx, hipStream_t *out);  // the copy-only stream of the context (created on demand)
bool ft_is_pinned_host(const void *p);  // null counts as pinned (nothing to copy)
bool ft_is_pinned_host_range(const void *p, size_t bytes);  // [p, p + bytes) inside ONE pinned host allocation
bool ft_host_block_contains(ft_context *ctx, const void *p, size_t bytes);
int ft_extract_foreign_wait(ft_extractor *ex);  // orders the extractor's next batch behind ex->foreignReader (never inside a stream capture)
int ft_ensure_scratch(ft_context *ctx, size_t devBytes, size_t pinBytes);  // grow-only matcher scratch (hold matchMutex)
int ft_usable_cpus();
int ft_hw_queues_hint();
bool ft_parse_lane_map(const char *text, std::vector<int> &map, std::string &err);
int ft_pipeline_depth(const ft_tuning &t, int batch, bool deviceOctree);
// validation, level-0 pointers / uploads of a whole batch (async on ex->stream)
int ft_extract_prepare(ft_extractor *ex, const uint8_t *const *images, int batch, int on_device, int width, int height,
                       int stride);
// per sub-batch of slots [b0, b0+nb): see extractor.cpp
int ft_extract_launch_a(ft_extractor *ex, int b0, int nb, hipEvent_t done);
int ft_extract_ensure_stage(ft_extractor *ex);
int ft_extract_ensure_host_cand(ft_extractor *ex);
void ft_extract_restage(ft_extractor *ex, const uint8_t *const *images, int batch, int width, int height, int stride);
int ft_extract_octree(ft_extractor *ex, int b0, int nb);
int ft_extract_overflow_slots(ft_extractor *ex, int batch, std::vector<int> &slots);
void ft_extract_update_big_grid(ft_extractor *ex);
int ft_extract_repair_prepare(const std::vector<std::pair<ft_extractor *, int>> &jobs);
int ft_extract_repair_launch(ft_extractor *ex, int slot, hipStream_t st);
int ft_extract_octree_multi(ft_extractor *const *exs, int nex, int b0, int nb);
int ft_extract_launch_octree(ft_extractor *ex, int sub, int b0, int nb, hipEvent_t done);  // device octree, own streams
int ft_extract_launch_b(ft_extractor *ex, int b0, int nb, hipStream_t st);
int ft_extract_finish_counts(ft_extractor *ex, int batch, hipStream_t st);  // device mode: totals + overflow flag to host
int ft_extract_download(ft_extractor *ex, int b0, int nb, hipStream_t st);
