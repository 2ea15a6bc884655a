// Library / device context: replaces KernelController::setCUDADevice / initializeKernels /
// shutdownKernels / saveKernelsStats (reference include/Kernels/KernelController.h:15-29).
#include <sched.h>

#include <algorithm>
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

extern "C" {

const char *ft_version(void) { return "fasttrack_amd 0.1 (gfx950)"; }

const char *ft_last_error(void) { return g_lastError.c_str(); }

int ft_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ft_context_create(int device, int host_threads, ft_context **out) {
    if (!out) {
        ft_set_error("ft_context_create: out is null");
        return FT_ERR_INVALID;
    }
    *out = nullptr;
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
    hipDeviceProp_t prop;
    FT_HIP(hipGetDeviceProperties(&prop, device));
    ft_context *ctx = new ft_context();
    ctx->device = device;
    ctx->deviceName = std::string(prop.name[0] ? prop.name : "AMD GPU") + " (" + prop.gcnArchName + ")";
    if (host_threads <= 0) host_threads = ft_usable_cpus();
    if (host_threads < 1) host_threads = 1;
    ctx->pool = new ft::ThreadPool(host_threads - 1);
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
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    if (ctx->scratchDev) hipFree(ctx->scratchDev);
    if (ctx->scratchPin) hipHostFree(ctx->scratchPin);
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

int ft_context_save_stats(ft_context *ctx, const char *path) {
    if (!ctx || !path) return FT_ERR_INVALID;
    FILE *f = fopen(path, "w");
    if (!f) {
        ft_set_error(std::string("cannot open ") + path);
        return FT_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(ctx->statsMutex);
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
    return FT_OK;
}

int ft_host_free(ft_context *ctx, void *ptr) {
    if (!ctx) return FT_ERR_INVALID;
    FT_HIP(hipSetDevice(ctx->device));
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
    static const char *env = getenv("FT_DEBUG_REPEAT");
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

