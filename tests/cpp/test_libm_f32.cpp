// Host check of fasttrack_amd/csrc/libm_f32.h against the libm this process is linked with:
//   * cosf_glibc / sinf_glibc == cosf / sinf for every float of [-6.2831860, 6.2831860] (the rBRIEF rotation angle of
//     /root/reference/src/ORBextractor.cc:72-74 is fastAtan2 degrees * (float)(pi/180); KB8's psi is in [-pi, pi]),
//   * logf_glibc == logf for every positive float up to 1e4 (PredictScale's ratio, /root/reference/src/MapPoint.cc:539)
//     plus every float above it at a coarse stride.
// Build with -DFT_LIBM_CONTRACT=1 (glibc's *_fma ifunc variant, the one an AVX2+FMA host runs) or =0 (the *_sse2
// variant); -ffp-contract=off in both so that only the explicit fma calls fuse.
// usage: test_libm_f32 [stride] [threads]     stride 1 = exhaustive (default 1), threads default 8
// Prints one line per function: checked / mismatches, and the first mismatching arguments.  Exit code 1 on mismatch.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../../fasttrack_amd/csrc/libm_f32.h"

static float from_bits(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static uint32_t to_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

template <class Mine, class Host>
static unsigned long long sweep(const char *name, uint32_t lo, uint32_t hi, uint32_t stride, int threads, Mine mine,
                                Host host) {
    std::atomic<unsigned long long> bad{0}, checked{0};
    std::atomic<int> shown{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&, t] {
            unsigned long long b = 0, c = 0;
            for (uint64_t u = (uint64_t)lo + (uint64_t)t * stride; u <= hi; u += (uint64_t)stride * threads) {
                const float x = from_bits((uint32_t)u);
                const float m = mine(x), h = host(x);
                if (to_bits(m) != to_bits(h)) {
                    b++;
                    if (shown.fetch_add(1) < 5) printf("  %s(%.9g = 0x%08x): mine %.9g host %.9g\n", name, x, (uint32_t)u, m, h);
                }
                c++;
            }
            bad += b;
            checked += c;
        });
    for (auto &th : pool) th.join();
    printf("%s: checked %llu mismatches %llu\n", name, checked.load(), bad.load());
    return bad.load();
}

int main(int argc, char **argv) {
    const uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1u;
    const int threads = argc > 2 ? atoi(argv[2]) : 8;
    printf("FT_LIBM_CONTRACT=%d stride %u\n", FT_LIBM_CONTRACT, stride);
    unsigned long long bad = 0;
    const uint32_t top = to_bits(6.2831860f);  // just above 360 * (float)(pi / 180)
    // volatile function pointers: the compiler must call libm, not fold or substitute a builtin
    float (*volatile hcos)(float) = cosf;
    float (*volatile hsin)(float) = sinf;
    float (*volatile hlog)(float) = logf;
    bad += sweep("cosf", 0, top, stride, threads, [](float x) { return ft_libm::cosf_glibc(x); }, [&](float x) { return hcos(x); });
    bad += sweep("sinf", 0, top, stride, threads, [](float x) { return ft_libm::sinf_glibc(x); }, [&](float x) { return hsin(x); });
    // negative arguments: KannalaBrandt8::project takes cos / sin of psi = atan2f(y, x) in [-pi, pi]
    // (/root/reference/src/CameraModels/KannalaBrandt8.cpp:74-75)
    bad += sweep("cosf(neg)", 0x80000000u, 0x80000000u | top, stride, threads, [](float x) { return ft_libm::cosf_glibc(x); },
                 [&](float x) { return hcos(x); });
    bad += sweep("sinf(neg)", 0x80000000u, 0x80000000u | top, stride, threads, [](float x) { return ft_libm::sinf_glibc(x); },
                 [&](float x) { return hsin(x); });
    // every positive float (subnormals included) up to 1e4, then the rest of the finite range at 64x the stride
    bad += sweep("logf(0,1e4]", 1, to_bits(1e4f), stride, threads, [](float x) { return ft_libm::logf_glibc(x); },
                 [&](float x) { return hlog(x); });
    bad += sweep("logf(1e4,max]", to_bits(1e4f), 0x7f7fffffu, stride * 64u, threads,
                 [](float x) { return ft_libm::logf_glibc(x); }, [&](float x) { return hlog(x); });
    // atanf: every float (NaNs compare by kind); atan2f: the KB8 shapes and a dense random sample
    float (*volatile hatan)(float) = atanf;
    float (*volatile hatan2)(float, float) = atan2f;
    {
        std::atomic<unsigned long long> b{0}, c{0};
        std::atomic<int> shown{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++)
            pool.emplace_back([&, t] {
                unsigned long long bb = 0, cc = 0;
                for (uint64_t u = (uint64_t)t * stride; u <= 0xffffffffull; u += (uint64_t)stride * threads) {
                    const float x = from_bits((uint32_t)u);
                    const float m = ft_libm::atanf_glibc(x), h = hatan(x);
                    if (to_bits(m) != to_bits(h) && !(m != m && h != h)) {
                        bb++;
                        if (shown.fetch_add(1) < 5) printf("  atanf(%.9g = 0x%08x): mine %.9g host %.9g\n", x, (uint32_t)u, m, h);
                    }
                    cc++;
                }
                b += bb;
                c += cc;
            });
        for (auto &th : pool) th.join();
        printf("atanf(all): checked %llu mismatches %llu\n", c.load(), b.load());
        bad += b.load();
    }
    {
        std::atomic<unsigned long long> b{0}, c{0};
        std::atomic<int> shown{0};
        std::vector<std::thread> pool;
        const unsigned long long per = (1ull << 31) / stride / threads;
        for (int t = 0; t < threads; t++)
            pool.emplace_back([&, t] {
                unsigned long long bb = 0, cc = 0;
                uint64_t s = 0x9e3779b97f4a7c15ull * (t + 1);
                auto next = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 32); };
                for (unsigned long long i = 0; i < per; i++) {
                    float y, x;
                    switch (i & 3) {
                        case 0: y = from_bits(next()); x = from_bits(next()); break;                          // any two floats
                        case 1: y = (float)(int32_t)next() * 0x1p-20f; x = (float)(int32_t)next() * 0x1p-20f; break;  // |.| < 2048: image / ray coordinates
                        case 2: y = std::fabs((float)(int32_t)next() * 0x1p-24f); x = (float)(int32_t)next() * 0x1p-22f; break;  // theta = atan2f(r >= 0, z)
                        default: y = from_bits(next()); x = (i & 4) ? 1.0f : from_bits((next() & 0x80000000u) | (i & 8 ? 0x7f800000u : 0u)); break;  // x = 1, +-0, +-inf
                    }
                    const float m = ft_libm::atan2f_glibc(y, x), h = hatan2(y, x);
                    if (to_bits(m) != to_bits(h) && !(m != m && h != h)) {
                        bb++;
                        if (shown.fetch_add(1) < 5) printf("  atan2f(%.9g, %.9g): mine %.9g host %.9g\n", y, x, m, h);
                    }
                    cc++;
                }
                b += bb;
                c += cc;
            });
        for (auto &th : pool) th.join();
        printf("atan2f(sample): checked %llu mismatches %llu\n", c.load(), b.load());
        bad += b.load();
    }
    // tanf on every float of (-120, 120)
    float (*volatile htan)(float) = tanf;
    for (uint32_t sign : {0u, 0x80000000u}) {
        std::atomic<unsigned long long> b{0}, c{0};
        std::atomic<int> shown{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++)
            pool.emplace_back([&, t] {
                unsigned long long bb = 0, cc = 0;
                for (uint64_t u = (uint64_t)t * stride; u < 0x42f00000ull; u += (uint64_t)stride * threads) {
                    const float x = from_bits((uint32_t)u | sign);
                    bool exact;
                    const float m = ft_libm::tanf_glibc(x, &exact), h = htan(x);
                    if (!exact || to_bits(m) != to_bits(h)) {
                        bb++;
                        if (shown.fetch_add(1) < 8) printf("  tanf(%.9g = 0x%08x): mine %.9g host %.9g\n", x, (uint32_t)u | sign, m, h);
                    }
                    cc++;
                }
                b += bb;
                c += cc;
            });
        for (auto &th : pool) th.join();
        printf("tanf(%s120): checked %llu mismatches %llu\n", sign ? "-" : "+", c.load(), b.load());
        bad += b.load();
    }
    return bad ? 1 : 0;
}
