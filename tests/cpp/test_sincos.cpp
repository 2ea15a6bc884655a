// Host check of fasttrack_amd/csrc/sincos_poly.h: for floats a in [0, 2 pi] (the angle computeOrbDescriptor rotates
// by), (float)cos((double)a) and (float)sin((double)a) from the polynomial equal libm's, bit for bit.
// usage: test_sincos [stride]   (stride 1 = every float of the interval, about 1.09e9 values; default 997)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../fasttrack_amd/csrc/sincos_poly.h"

int main(int argc, char **argv) {
    const uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 997u;
    const float top = 6.2831860f;  // just above 360 * (float)(pi / 180)
    uint32_t last;
    memcpy(&last, &top, 4);
    unsigned long long checked = 0, bad = 0;
    for (uint64_t bits = 0; bits <= last; bits += stride) {
        const uint32_t b32 = (uint32_t)bits;
        float a;
        memcpy(&a, &b32, 4);
        double sn, cs;
        ft_sincos_0_2pi((double)a, sn, cs);
        const float c0 = (float)cos((double)a), s0 = (float)sin((double)a);
        const float c1 = (float)cs, s1 = (float)sn;
        if (memcmp(&c0, &c1, 4) || memcmp(&s0, &s1, 4)) {
            if (bad < 10) printf("a=%.9g cos %.9g vs %.9g sin %.9g vs %.9g\n", a, c0, c1, s0, s1);
            bad++;
        }
        checked++;
    }
    // the angles that actually occur: fastAtan2 degrees (multiples are not special, so a dense sweep) times pi/180
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    for (int i = 0; i <= 3600000; i++) {
        const float a = (float)(i * 1e-4) * factorPI;
        double sn, cs;
        ft_sincos_0_2pi((double)a, sn, cs);
        const float c0 = (float)cos((double)a), s0 = (float)sin((double)a), c1 = (float)cs, s1 = (float)sn;
        if (memcmp(&c0, &c1, 4) || memcmp(&s0, &s1, 4)) bad++;
        checked++;
    }
    printf("checked %llu mismatches %llu\n", checked, bad);
    return bad ? 1 : 0;
}
