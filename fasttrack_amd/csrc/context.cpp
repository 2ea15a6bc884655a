// Library / device context: replaces KernelController::setCUDADevice / initializeKernels /
// shutdownKernels / saveKernelsStats (reference include/Kernels/KernelController.h:15-29).
#include <sched.h>

#include <algorithm>
#include <cerrno>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "ft_host.h"

static thread_local std::string g_lastError;

void ft_set_error(const std::string &msg) { g_lastError = msg; }

int ft_hip_fail(hipError_t e, const char *what, const char *file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    g_lastError = buf;
    // a missing device must be unmistakable: there is no CPU fallback behind this ABI
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver) return FT_ERR_NO_DEVICE;
    return FT_ERR_HIP;
}

// CPUs this process may actually use: hardware threads, capped by the scheduler affinity and by a
// cgroup CPU quota (containers often expose 256 logical CPUs with a 16-CPU quota; oversubscribing the
// quota gets every thread throttled).
int ft_usable_cpus() {
    int n = (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, CPU_COUNT(&set));
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
        char q[64];
        long period = 0;
        if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long quota = atol(q);
            if (quota > 0) n = std::min<long>(n, std::max<long>(1, (quota + period - 1) / period));
        }
        fclose(f);
    } else {
        long quota = -1, period = 0;
        if (FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(fq, "%ld", &quota) != 1) quota = -1;
            fclose(fq);
        }
        if (FILE *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(fp, "%ld", &period) != 1) period = 0;
            fclose(fp);
        }
        if (quota > 0 && period > 0) n = std::min<long>(n, std::max<long>(1, (quota + period - 1) / period));
    }
    return std::max(n, 1);
}

int ft_set_device(const ft_context *ctx) {
    FT_HIP(hipSetDevice(ctx->device));
    return FT_OK;
}

int ft_context_upload_stream(ft_context *ctx, hipStream_t *out) {
    std::lock_guard<std::mutex> lk(ctx->laneMutex);
    if (!ctx->uploadStream) FT_HIP(hipStreamCreateWithFlags(&ctx->uploadStream, hipStreamNonBlocking));
    *out = ctx->uploadStream;
    return FT_OK;
}

// the streams of the next extractor (see ft_host.h); out[FT_LANE_STREAMS]
int ft_context_take_lanes(ft_context *ctx, bool wantPrivate, int width, int height, hipStream_t *out, bool *owned) {
    std::lock_guard<std::mutex> lk(ctx->laneMutex);
    // which table: the user's (FT_LANE_MAP / ft_context_set_lane_map) for every shape, else the one searched for frames of
    // this size class (ft_host.h)
    const std::vector<int> &laneMap = ctx->userLaneMap || (size_t)width * height >= FT_LANE_SMALL_PIXELS ? ctx->laneMap : ctx->laneMapSmall;
    if (wantPrivate || laneMap.empty()) {
        for (int i = 0; i < FT_LANE_STREAMS; i++) out[i] = nullptr;
        for (int i = 0; i < FT_LANE_STREAMS; i++) {
            const hipError_t e = hipStreamCreateWithFlags(&out[i], hipStreamNonBlocking);
            if (e != hipSuccess) {
                for (int k = 0; k < i; k++) {
                    hipStreamDestroy(out[k]);
                    out[k] = nullptr;
                }
                return ft_hip_fail(e, "hipStreamCreateWithFlags", __FILE__, __LINE__);
            }
        }
        *owned = true;
        return FT_OK;
    }
    const size_t sets = laneMap.size() / FT_LANE_STREAMS;
    const size_t set = (size_t)(ctx->nextLaneSet++) % sets;
    for (int i = 0; i < FT_LANE_STREAMS; i++) {
        const size_t lane = (size_t)laneMap[set * FT_LANE_STREAMS + i];
        if (ctx->lanes.size() <= lane) ctx->lanes.resize(lane + 1, nullptr);
        if (!ctx->lanes[lane]) FT_HIP(hipStreamCreateWithFlags(&ctx->lanes[lane], hipStreamNonBlocking));
        out[i] = ctx->lanes[lane];
    }
    *owned = false;
    return FT_OK;
}

// hardware queues the HIP runtime multiplexes this process's streams onto: GPU_MAX_HW_QUEUES as the process environment
// holds it (the runtime reads it once, when it initialises), 4 when it is unset
int ft_hw_queues_hint() {
    const char *e = ft_read_env("GPU_MAX_HW_QUEUES");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 4;
}

// ---- the environment is read HERE and nowhere else (ft_host.h, FT_TUNING_OPTIONS) ----------------------------------------
const char *ft_read_env(const char *name) { return getenv(name); }
const char *ft_debug_env(const char *name) { return strncmp(name, "FT_DEBUG_", 9) == 0 ? ft_read_env(name) : nullptr; }

static const struct {
    const char *name, *env, *doc;
    int def, lo, hi;
} kTuningTable[] = {
#define FT_X(field, env, def, lo, hi, doc) {#field, env, doc, def, lo, hi},
    FT_TUNING_OPTIONS(FT_X)
#undef FT_X
};
static const int kTuningCount = (int)(sizeof kTuningTable / sizeof kTuningTable[0]);

int *ft_tuning_field(ft_tuning &t, const char *name) {
    if (!name) return nullptr;
#define FT_X(field, env, def, lo, hi, doc) \
    if (strcmp(name, #field) == 0 || strcmp(name, env) == 0) return &t.field;
    FT_TUNING_OPTIONS(FT_X)
#undef FT_X
    return nullptr;
}

// the range of option `name` (by option or environment name); false = unknown
static bool tuningRange(const char *name, int &lo, int &hi) {
    for (int i = 0; i < kTuningCount; i++)
        if (strcmp(name, kTuningTable[i].name) == 0 || strcmp(name, kTuningTable[i].env) == 0) {
            lo = kTuningTable[i].lo;
            hi = kTuningTable[i].hi;
            return true;
        }
    return false;
}

// a value outside an option's range is an error (FT_ERR_INVALID from ft_context_create), not a silently different setting
bool ft_tuning_from_env(ft_tuning &t, std::string &err) {
    for (int i = 0; i < kTuningCount; i++) {
        const char *e = ft_read_env(kTuningTable[i].env);
        if (!e || !*e) continue;
        // the whole text must be a number: "abc" or "1x" is an error, not 0 or 1
        char *end = nullptr;
        errno = 0;
        const long lv = strtol(e, &end, 10);
        while (end && (*end == ' ' || *end == '\t')) end++;
        if (end == e || (end && *end) || errno == ERANGE) {
            err = std::string(kTuningTable[i].env) + "=" + e + " is not an integer";
            return false;
        }
        const int v = lv < INT_MIN ? INT_MIN : lv > INT_MAX ? INT_MAX : (int)lv;
        if (v < kTuningTable[i].lo || v > kTuningTable[i].hi) {
            err = std::string(kTuningTable[i].env) + "=" + e + " is outside [" + std::to_string(kTuningTable[i].lo) + ", " +
                  std::to_string(kTuningTable[i].hi) + "]";
            return false;
        }
        *ft_tuning_field(t, kTuningTable[i].name) = v;
    }
    return true;
}

// "own" (private streams: empty map) or whole sets of four lane numbers in [0, 64); anything else is an error, not a
// silently different table
bool ft_parse_lane_map(const char *text, std::vector<int> &map, std::string &err) {
    std::vector<int> m;
    const char *p = text;
    while (*p == ' ' || *p == '\t' || *p == ',') p++;
    if (strncmp(p, "own", 3) == 0) {
        p += 3;
        while (*p == ' ' || *p == '\t') p++;
        if (*p) {
            err = "unexpected text after \"own\"";
            return false;
        }
        map.clear();
        return true;
    }
    while (*p) {
        char *end = nullptr;
        const long v = strtol(p, &end, 10);
        if (end == p) {
            err = std::string("not a number: \"") + p + "\"";
            return false;
        }
        if (v < 0 || v >= 64) {
            err = "lane " + std::to_string(v) + " outside [0, 64)";
            return false;
        }
        m.push_back((int)v);
        p = end;
        while (*p == ' ' || *p == '\t' || *p == ',') p++;
    }
    if (m.empty() || m.size() % FT_LANE_STREAMS != 0) {
        err = std::to_string(m.size()) + " entries: need whole sets of (stage A, stage B, octree 0, octree 1)";
        return false;
    }
    map = m;
    return true;
}

extern "C" {

#include "version.inc"  // FT_CSRC_HASH: sha1 of the csrc sources this library was built from (Makefile)
const char *ft_version(void) { return "fasttrack_amd 0.5 (gfx950) csrc:" FT_CSRC_HASH; }

const char *ft_last_error(void) { return g_lastError.c_str(); }

int ft_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// "0000:c1:00.0" of a HIP device: the key to /sys/bus/pci/devices/<id>/numa_node, by which a rank pins its host
// threads next to its GPU before it creates the context (fasttrack_amd/shard.py)
int ft_device_pci_bus_id(int device, char *buf, int len) {
    if (!buf || len < 16) {
        ft_set_error("ft_device_pci_bus_id: buffer too small");
        return FT_ERR_INVALID;
    }
    FT_HIP(hipDeviceGetPCIBusId(buf, len, device));
    return FT_OK;
}

int ft_context_create(int device, int host_threads, ft_context **out) {
    if (!out) {
        ft_set_error("ft_context_create: out is null");
        return FT_ERR_INVALID;
    }
    *out = nullptr;
    // The HIP runtime multiplexes the streams of a process onto GPU_MAX_HW_QUEUES in-order hardware queues (4 unless the
    // variable says otherwise), handed out as the streams are first used; streams that share a queue run one kernel at a time,
    // whatever their events say, and which streams share what is worth +-20 % of the throughput (49 k - 71 k frames/s over
    // GPU_MAX_HW_QUEUES = 5 .. 16 with private streams).  So the wide extractors run on "lanes" of the context (ft_host.h), by a
    // table made for the number of queues the process HAS: the runtime reads the variable when it initialises - at the first
    // HIP call of the process, which need not be ours - so the library does not touch the environment (round 2 did: a
    // setenv that came too late whenever the caller, a device probe or another HIP user had initialised the runtime, and a
    // write to process-wide state from library code).  The application sets GPU_MAX_HW_QUEUES (10 serves the eight-lane
    // table: a queue per lane, the upload stream and the matchers' stream) before its first HIP call - bench.py and the
    // Python driver do - and ft_context_hw_queues() / the "context.hw_queues" statistic say which table is in use.
    std::vector<int> userMap;
    const char *lm = ft_read_env("FT_LANE_MAP");
    // exported but empty (or blank) = unset: "FT_LANE_MAP= cmd" is how a shell neutralises a variable
    if (lm && lm[strspn(lm, " \t")] == 0) lm = nullptr;
    const bool haveUserMap = lm != nullptr;
    if (haveUserMap) {  // checked before anything else: a typo must not pass as a different table
        std::string err;
        if (!ft_parse_lane_map(lm, userMap, err)) {
            ft_set_error("FT_LANE_MAP: " + err);
            return FT_ERR_INVALID;
        }
    }
    ft_tuning tuning;
    {
        std::string err;
        if (!ft_tuning_from_env(tuning, err)) {
            ft_set_error("ft_context_create: " + err);
            return FT_ERR_INVALID;
        }
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        ft_set_error("no HIP device available: the fasttrack_amd kernels need a gfx950 GPU (there is no CPU fallback)");
        return FT_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        ft_set_error("ft_context_create: device index out of range");
        return FT_ERR_INVALID;
    }
    FT_HIP(hipSetDevice(device));
    // How a host thread waits for the device.  The runtime's default spins: fine while every waiting thread has a core of its
    // own, ruinous when more threads wait than the process has cores (a spinning waiter keeps the thread that has launches
    // to make off the core for a scheduler slice).  A device flag of the whole process; a process that has set its own
    // keeps it (the call then fails and is ignored).
    if (tuning.blocking_sync == 1 || (tuning.blocking_sync == 2 && ft_usable_cpus() < 8)) {
        if (hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) (void)hipGetLastError();
    }
    hipDeviceProp_t prop;
    FT_HIP(hipGetDeviceProperties(&prop, device));
    ft_context *ctx = new ft_context();
    ctx->device = device;
    ctx->deviceName = std::string(prop.name[0] ? prop.name : "AMD GPU") + " (" + prop.gcnArchName + ")";
    {
        int mp = 0;
        if (hipDeviceGetAttribute(&mp, hipDeviceAttributeMaxPitch, device) == hipSuccess && mp > 0) ctx->maxPitch = (size_t)mp;
        else (void)hipGetLastError();
    }
    if (host_threads <= 0) host_threads = ft_usable_cpus();
    if (host_threads < 1) host_threads = 1;
    ctx->pool = new ft::ThreadPool(host_threads - 1);
    // FT_LANE_MAP="a b o0 o1  a b o0 o1 ..." : lane of (stage A, stage B, octree 0, octree 1) for the 1st, 2nd, ... extractor
    // created on the context (the list wraps around); FT_LANE_MAP=own gives every extractor four streams of its own
    ctx->tuning = tuning;
    ctx->hwQueues = ft_hw_queues_hint();
    // the shipped table was searched with 10 queues (a queue per lane, the upload stream, the matchers' stream): with fewer
    // the lanes share queues in an order the runtime picks, which is what the table exists to avoid
    if (ctx->hwQueues >= 10) {
        // searched with 10 queues (tools/lane_search.py): on 1280x720 / 512 pairs, and on 752x480 / 512 pairs for the small class
        // (climbed again at the end of round 3, when the octree tier had become short: two entries moved, +0.7 % on the
        // headline scenes, +1.1 % on dense ones)
        ctx->laneMap = {1, 2, 7, 4, 5, 3, 3, 1, 1, 4, 3, 2, 1, 1, 4, 3};  // (round 4: three entries moved by two more climbs, +0.7 % and +0.5 %; profiles/r04_lane_search.txt)
        ctx->laneMapSmall = FT_LANE_MAP_SMALL;
    } else ctx->laneMap.clear();  // the runtime's default of four queues: private streams, placed by the runtime (66 k frames/s on the headline
                                // workload; a lane per stage shared by the cameras and front ends - {0,1,2,3} for everyone - ran 55 k)
    if (haveUserMap) {
        ctx->laneMap = userMap;
        ctx->userLaneMap = true;
    }
    hipError_t se = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
        delete ctx->pool;
        delete ctx;
        return ft_hip_fail(se, "hipStreamCreateWithFlags", __FILE__, __LINE__);
    }
    *out = ctx;
    return FT_OK;
}

int ft_context_destroy(ft_context *ctx) {
    if (!ctx) return FT_OK;
    if (ctx->liveObjects.load() > 0) {
        // extractors / front ends of this context still hold its lane streams: destroying those under them would leave
        // their own destructors synchronising dead handles
        ft_set_error("ft_context_destroy: extractors, front ends or tracked frames of this context are still alive - destroy them first");
        return FT_ERR_INVALID;
    }
    hipSetDevice(ctx->device);
    for (hipStream_t s : ctx->lanes)
        if (s) {
            hipStreamSynchronize(s);
            hipStreamDestroy(s);
        }
    if (ctx->uploadStream) {
        hipStreamSynchronize(ctx->uploadStream);
        hipStreamDestroy(ctx->uploadStream);
    }
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    if (ctx->scratchDev) hipFree(ctx->scratchDev);
    if (ctx->scratchPin) hipHostFree(ctx->scratchPin);
    for (void *p : ctx->hostAllocs) hipHostFree(p);  // ft_host_malloc blocks the caller did not free: they go with the context
    for (hipEvent_t e : ctx->retiredEvents) hipEventDestroy(e);
    delete ctx->pool;
    delete ctx;
    return FT_OK;
}

int ft_context_synchronize(ft_context *ctx) {
    if (!ctx) return FT_ERR_INVALID;
    FT_HIP(hipSetDevice(ctx->device));
    FT_HIP(hipDeviceSynchronize());
    return FT_OK;
}

int ft_context_device_name(ft_context *ctx, char *buf, int len) {
    if (!ctx || !buf || len <= 0) return FT_ERR_INVALID;
    snprintf(buf, len, "%s", ctx->deviceName.c_str());
    return FT_OK;
}

int ft_context_host_threads(const ft_context *ctx) { return ctx ? ctx->pool->size() : 0; }

int ft_context_hw_queues(const ft_context *ctx) { return ctx ? ctx->hwQueues : 0; }

int ft_context_set_lane_map(ft_context *ctx, const int *map, int n) {
    if (!ctx || n < 0 || (n > 0 && !map)) return FT_ERR_INVALID;
    if (n % FT_LANE_STREAMS != 0) {
        ft_set_error("ft_context_set_lane_map: need whole sets of (stage A, stage B, octree 0, octree 1)");
        return FT_ERR_INVALID;
    }
    for (int i = 0; i < n; i++)
        if (map[i] < 0 || map[i] >= 64) {
            ft_set_error("ft_context_set_lane_map: lane outside [0, 64)");
            return FT_ERR_INVALID;
        }
    std::lock_guard<std::mutex> lk(ctx->laneMutex);
    ctx->laneMap.assign(map, map + n);  // n == 0: private streams for every extractor
    ctx->userLaneMap = true;
    ctx->nextLaneSet = 0;
    return FT_OK;
}

int ft_context_set_option(ft_context *ctx, const char *name, int value) {
    int *f = ctx ? ft_tuning_field(ctx->tuning, name) : nullptr;
    if (!f) {
        ft_set_error(std::string("ft_context_set_option: unknown option \"") + (name ? name : "(null)") + "\"");
        return FT_ERR_INVALID;
    }
    int lo = 0, hi = 0;
    if (!tuningRange(name, lo, hi) || value < lo || value > hi) {
        ft_set_error(std::string("ft_context_set_option: ") + name + " = " + std::to_string(value) + " is outside [" +
                     std::to_string(lo) + ", " + std::to_string(hi) + "]");
        return FT_ERR_INVALID;
    }
    // the search_* and pass_burst options are read per call, under the same mutex
    std::lock_guard<std::mutex> lk(ctx->matchMutex);
    *f = value;
    return FT_OK;
}

int ft_context_get_option(const ft_context *ctx, const char *name, int *value) {
    const int *f = ctx && value ? ft_tuning_field(const_cast<ft_context *>(ctx)->tuning, name) : nullptr;
    if (!f) {
        ft_set_error(std::string("ft_context_get_option: unknown option \"") + (name ? name : "(null)") + "\"");
        return FT_ERR_INVALID;
    }
    *value = *f;
    return FT_OK;
}

int ft_option_range(const char *name, int *min_value, int *max_value) {
    int lo = 0, hi = 0;
    if (!name || !tuningRange(name, lo, hi)) return FT_ERR_INVALID;
    if (min_value) *min_value = lo;
    if (max_value) *max_value = hi;
    return FT_OK;
}

int ft_option_describe(int index, const char **name, const char **env, int *default_value, const char **doc) {
    if (index < 0 || index >= kTuningCount) return FT_ERR_INVALID;
    if (name) *name = kTuningTable[index].name;
    if (env) *env = kTuningTable[index].env;
    if (default_value) *default_value = kTuningTable[index].def;
    if (doc) *doc = kTuningTable[index].doc;
    return FT_OK;
}

int ft_context_save_stats(ft_context *ctx, const char *path) {
    if (!ctx || !path) return FT_ERR_INVALID;
    FILE *f = fopen(path, "w");
    if (!f) {
        ft_set_error(std::string("cannot open ") + path);
        return FT_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(ctx->statsMutex);
    fprintf(f, "context.hw_queues: %d (GPU_MAX_HW_QUEUES as the process environment held it when the context was created)\n", ctx->hwQueues);
    for (auto &kv : ctx->stats)
        fprintf(f, "%s: %.4f ms total, %ld calls, %.4f ms/call\n", kv.first.c_str(), kv.second.first, kv.second.second,
                kv.second.second ? kv.second.first / kv.second.second : 0.0);
    fclose(f);
    return FT_OK;
}

int ft_context_set_kernel_timing(ft_context *ctx, int enabled) {
    if (!ctx) return FT_ERR_INVALID;
    ctx->kernelTiming = enabled != 0;
    return FT_OK;
}

int ft_context_get_stat(ft_context *ctx, const char *name, double *total_ms, long *calls) {
    if (!ctx || !name) return FT_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->statsMutex);
    auto it = ctx->stats.find(name);
    if (total_ms) *total_ms = it == ctx->stats.end() ? 0.0 : it->second.first;
    if (calls) *calls = it == ctx->stats.end() ? 0 : it->second.second;
    return FT_OK;
}

int ft_context_reset_stats(ft_context *ctx) {
    if (!ctx) return FT_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->statsMutex);
    ctx->stats.clear();
    return FT_OK;
}

int ft_device_malloc(ft_context *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return FT_ERR_INVALID;
    FT_HIP(hipSetDevice(ctx->device));
    FT_HIP(hipMalloc(dptr, bytes));
    return FT_OK;
}

int ft_device_free(ft_context *ctx, void *dptr) {
    if (!ctx) return FT_ERR_INVALID;
    FT_HIP(hipSetDevice(ctx->device));
    FT_HIP(hipFree(dptr));
    return FT_OK;
}

int ft_host_malloc(ft_context *ctx, size_t bytes, void **ptr) {
    if (!ctx || !ptr) return FT_ERR_INVALID;
    FT_HIP(hipSetDevice(ctx->device));
    FT_HIP(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
    std::lock_guard<std::mutex> lk(ctx->hostAllocMutex);
    ctx->hostAllocs.push_back(*ptr);  // released by ft_host_free, or with the context
    ctx->hostBlocks[(uintptr_t)*ptr] = bytes ? bytes : 1;
    return FT_OK;
}

int ft_host_free(ft_context *ctx, void *ptr) {
    if (!ctx) return FT_ERR_INVALID;
    FT_HIP(hipSetDevice(ctx->device));
    {
        std::lock_guard<std::mutex> lk(ctx->hostAllocMutex);
        auto it = std::find(ctx->hostAllocs.begin(), ctx->hostAllocs.end(), ptr);
        if (it != ctx->hostAllocs.end()) ctx->hostAllocs.erase(it);
        ctx->hostBlocks.erase((uintptr_t)ptr);
    }
    FT_HIP(hipHostFree(ptr));
    return FT_OK;
}

int ft_memcpy_h2d(ft_context *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return FT_ERR_INVALID;
    FT_HIP(hipSetDevice(ctx->device));
    FT_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return FT_OK;
}

int ft_memcpy_d2h(ft_context *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return FT_ERR_INVALID;
    FT_HIP(hipSetDevice(ctx->device));
    FT_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return FT_OK;
}

}  // extern "C"

int ft_debug_repeat(const char *name) {
    static const char *env = ft_debug_env("FT_DEBUG_REPEAT");
    if (!env || !*env) return 1;
    const size_t n = strlen(name);
    for (const char *p = env; *p;) {
        const char *e = strchr(p, ',');
        const size_t len = e ? (size_t)(e - p) : strlen(p);
        if (len == n && strncmp(p, name, n) == 0) return 2;
        p += len;
        if (*p == ',') p++;
    }
    return 1;
}

bool ft_is_pinned_host(const void *p) {
    if (!p) return true;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// [p, p + bytes) inside a block of ft_host_malloc: pinned for certain, and known without asking the runtime (a call that checks
// hundreds of arrays - ft_tracked_batch_submit_* - cannot afford a hipPointerGetAttributes each)
bool ft_host_block_contains(ft_context *ctx, const void *p, size_t bytes) {
    if (!ctx || !p) return false;
    const uintptr_t a = (uintptr_t)p;
    std::lock_guard<std::mutex> lk(ctx->hostAllocMutex);
    auto it = ctx->hostBlocks.upper_bound(a);  // the first block that starts behind a: the one in front of it may hold a
    if (it == ctx->hostBlocks.begin()) return false;
    --it;
    return a + bytes <= it->first + it->second;
}

// A frame the device may read in place: first AND last byte lie in the same pinned / registered host allocation.
// (A frame that starts inside a pinned region and runs past it - wrong stride or height, a view at the tail of a
// registered range - would be a device access to unpinned memory: a fatal GPU fault, not an error code.)
bool ft_is_pinned_host_range(const void *p, size_t bytes) {
    if (!p || bytes == 0) return false;
    if (!ft_is_pinned_host(p)) return false;
    const uint8_t *last = (const uint8_t *)p + (bytes - 1);
    if (!ft_is_pinned_host(last)) return false;
    void *b0 = nullptr, *b1 = nullptr;
    size_t s0 = 0, s1 = 0;
    if (hipMemGetAddressRange((hipDeviceptr_t *)&b0, &s0, (hipDeviceptr_t)p) != hipSuccess ||
        hipMemGetAddressRange((hipDeviceptr_t *)&b1, &s1, (hipDeviceptr_t)last) != hipSuccess) {
        (void)hipGetLastError();
        // the runtime cannot name the allocation (e.g. hipHostRegister'ed ranges on some stacks): be conservative
        return false;
    }
    return b0 == b1;
}

// grow-only device / pinned scratch of the matchers (callers hold ctx->matchMutex)
int ft_ensure_scratch(ft_context *ctx, size_t devBytes, size_t pinBytes) {
    if (devBytes > ctx->scratchDevBytes) {
        if (ctx->scratchDev) hipFree(ctx->scratchDev);
        ctx->scratchDev = nullptr;
        ctx->scratchDevBytes = 0;
        const size_t want = devBytes + devBytes / 2;
        FT_HIP(hipMalloc(&ctx->scratchDev, want));
        ctx->scratchDevBytes = want;
    }
    if (pinBytes > ctx->scratchPinBytes) {
        if (ctx->scratchPin) hipHostFree(ctx->scratchPin);
        ctx->scratchPin = nullptr;
        ctx->scratchPinBytes = 0;
        const size_t want = pinBytes + pinBytes / 2;
        FT_HIP(hipHostMalloc(&ctx->scratchPin, want, hipHostMallocDefault));
        ctx->scratchPinBytes = want;
    }
    return FT_OK;
}

