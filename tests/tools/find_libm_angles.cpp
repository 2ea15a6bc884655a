// Finds keypoint angles (degrees, as cv::fastAtan2 could return them) at which the rBRIEF rotation of
// /root/reference/src/ORBextractor.cc:72-81 samples a DIFFERENT pixel with libm's cosf / sinf than with the narrowed double
// cos / sin the oracle used until round 3.  The angles feed tests/tools/make_golden_vectors.py (libm_rotation case).
// usage: find_libm_angles [how_many=48] [seed=1]   -> prints one angle bit pattern (hex) per line
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../oracle/orb_pattern.inc"

int main(int argc, char **argv) {
    const int want = argc > 1 ? atoi(argv[1]) : 48;
    uint64_t s = argc > 2 ? strtoull(argv[2], 0, 10) : 1;
    const float factorPI = (float)(M_PI / 180.f);
    int found = 0;
    unsigned long long tried = 0, trigDiff = 0;
    while (found < want) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const float deg = (float)((s >> 11) * (1.0 / 9007199254740992.0) * 360.0);
        if (!(deg < 360.f)) continue;
        tried++;
        const float ang = deg * factorPI;
        const float a = cosf(ang), b = sinf(ang);
        const float a2 = (float)cos((double)ang), b2 = (float)sin((double)ang);
        if (a == a2 && b == b2) continue;
        trigDiff++;
        bool differs = false;
        for (int i = 0; i < 512 && !differs; i++) {
            const float px = kOrcPattern31[2 * i], py = kOrcPattern31[2 * i + 1];
            differs = lrintf(px * b + py * a) != lrintf(px * b2 + py * a2) || lrintf(px * a - py * b) != lrintf(px * a2 - py * b2);
        }
        if (differs) {
            uint32_t u;
            memcpy(&u, &deg, 4);
            printf("0x%08x\n", u);
            found++;
        }
    }
    fprintf(stderr, "tried %llu angles, %llu with a different cosf/sinf, %d with a different sample\n", tried, trigDiff, found);
    return 0;
}
