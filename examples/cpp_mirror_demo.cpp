// Minimal C++ use of the mirror header: extract one image with the reference's call shape, then stereo-match.
// Build: g++ -std=c++17 -Iinclude examples/cpp_mirror_demo.cpp -Lfasttrack_amd -lfasttrack_amd -o demo
// (run with LD_LIBRARY_PATH=fasttrack_amd; needs an MI355X - there is no CPU fallback).
#include <cstdio>
#include <vector>

#include "fasttrack_amd.hpp"

int main() {
    try {
        fasttrack::Context ctx(0);
        const int W = 640, H = 480;
        std::vector<uint8_t> img((size_t)W * H);
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) img[(size_t)y * W + x] = (uint8_t)(((x / 16 + y / 16) & 1) ? 200 : 60);
        fasttrack::ORBextractor<> left(ctx, 1000, 1.2f, 8, 20, 7, W, H), right(ctx, 1000, 1.2f, 8, 20, 7, W, H);
        std::vector<ft_keypoint> keysL, keysR;
        std::vector<uint8_t> descL, descR;
        std::vector<int> lap = {0, 0};
        fasttrack::ImageView view{img.data(), W, H, W};
        const int monoL = left(view, keysL, descL, lap);
        right(view, keysR, descR, lap);
        std::vector<std::pair<int, int>> vDistIdx;
        std::vector<float> mvuRight, mvDepth;
        fasttrack::KernelController::launchStereoMatchKernel(left, right, keysL, keysR, descL.data(), descR.data(), 50.f,
                                                            0.11f, true, vDistIdx, mvuRight, mvDepth);
        std::printf("%zu keypoints (mono index %d), %zu stereo matches\n", keysL.size(), monoL, vDistIdx.size());
    } catch (const fasttrack::Error &e) {
        std::fprintf(stderr, "%s (status %d)\n", e.what(), e.status);
        return 1;
    }
    return 0;
}
