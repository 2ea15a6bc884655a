// C++ use of the mirror header with the reference's call shapes: ORBextractor::operator() on the left and the right
// image, KernelController::launchStereoMatchKernel, then KernelController::launchSearchLocalPointsKernel on the
// extracted frame.
// Build: g++ -std=c++17 -Iinclude examples/cpp_mirror_demo.cpp -Lfasttrack_amd -lfasttrack_amd -o demo
// (run with LD_LIBRARY_PATH=fasttrack_amd; needs an MI355X - there is no CPU fallback).
//   demo                                     a built-in checkerboard frame, prints counts
//   demo W H NF left.raw right.raw MBF MB out.bin [points.bin TH]
//        frames from raw 8-bit files (W x H bytes each); everything the calls returned goes to out.bin as int32 /
//        raw records (see dump() below) - tests/test_cpp_header.py writes the inputs and compares out.bin with the
//        oracle.  points.bin: int32 M, then the ft_local_points arrays in the order of the struct.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fasttrack_amd.hpp"

static std::vector<uint8_t> readFile(const char *path) {
    std::vector<uint8_t> v;
    FILE *f = std::fopen(path, "rb");
    if (!f) {
        std::fprintf(stderr, "cannot open %s\n", path);
        std::exit(2);
    }
    std::fseek(f, 0, SEEK_END);
    v.resize((size_t)std::ftell(f));
    std::fseek(f, 0, SEEK_SET);
    if (!v.empty() && std::fread(v.data(), 1, v.size(), f) != v.size()) std::exit(2);
    std::fclose(f);
    return v;
}

template <class T>
static void dump(FILE *f, const std::vector<T> &v) {
    const int n = (int)v.size();
    std::fwrite(&n, sizeof n, 1, f);
    if (n) std::fwrite(v.data(), sizeof(T), v.size(), f);
}

int main(int argc, char **argv) {
    try {
        // the application's choice, before the first HIP call of the process: a hardware queue per lane of the context
        setenv("GPU_MAX_HW_QUEUES", "10", 0);
        fasttrack::Context ctx(0);
        const bool files = argc >= 9;
        const int W = files ? std::atoi(argv[1]) : 640, H = files ? std::atoi(argv[2]) : 480, NF = files ? std::atoi(argv[3]) : 1000;
        std::vector<uint8_t> imgL((size_t)W * H), imgR;
        if (files) {
            imgL = readFile(argv[4]);
            imgR = readFile(argv[5]);
            if (imgL.size() != (size_t)W * H || imgR.size() != (size_t)W * H) {
                std::fprintf(stderr, "frame files must hold W x H bytes\n");
                return 2;
            }
        } else {
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) imgL[(size_t)y * W + x] = (uint8_t)(((x / 16 + y / 16) & 1) ? 200 : 60);
            imgR = imgL;
        }
        const float mbf = files ? (float)std::atof(argv[6]) : 50.f, mb = files ? (float)std::atof(argv[7]) : 0.11f;
        fasttrack::ORBextractor<> left(ctx, NF, 1.2f, 8, 20, 7, W, H), right(ctx, NF, 1.2f, 8, 20, 7, W, H);
        std::vector<ft_keypoint> keysL, keysR;
        std::vector<uint8_t> descL, descR;
        std::vector<int> lap = {0, 0};
        const int monoL = left(fasttrack::ImageView{imgL.data(), W, H, W}, keysL, descL, lap);
        const int monoR = right(fasttrack::ImageView{imgR.data(), W, H, W}, keysR, descR, lap);
        std::vector<std::pair<int, int>> vDistIdx;
        std::vector<float> mvuRight, mvDepth;
        fasttrack::KernelController::launchStereoMatchKernel(left, right, keysL, keysR, descL.data(), descR.data(), mbf, mb, true,
                                                            vDistIdx, mvuRight, mvDepth);
        std::printf("%zu + %zu keypoints (mono index %d / %d), %zu stereo matches\n", keysL.size(), keysR.size(), monoL, monoR,
                    vDistIdx.size());
        if (!files) return 0;
        FILE *out = std::fopen(argv[8], "wb");
        if (!out) return 2;
        dump(out, keysL); dump(out, descL); dump(out, keysR); dump(out, descR); dump(out, mvuRight); dump(out, mvDepth);
        std::vector<int> sadIdx;
        for (auto &p : vDistIdx) { sadIdx.push_back(p.first); sadIdx.push_back(p.second); }
        dump(out, sadIdx);
        if (argc >= 11) {
            // ORBmatcher::SearchByProjection(F, vpMapPoints, th) through the reference's kernel seam, on the frame just extracted
            const std::vector<uint8_t> pb = readFile(argv[9]);
            const float th = (float)std::atof(argv[10]);
            const uint8_t *p = pb.data();
            ft_local_points P{};
            P.M = *(const int *)p; p += 4;
            const size_t M = (size_t)P.M;
            auto take = [&](size_t bytes) { const uint8_t *q = p; p += bytes; return q; };
            P.skip = take(M); P.in_view = take(M); P.in_view_r = take(M);
            P.level = (const int *)take(4 * M); P.level_r = (const int *)take(4 * M);
            P.view_cos = (const float *)take(4 * M); P.view_cos_r = (const float *)take(4 * M);
            P.proj_x = (const float *)take(4 * M); P.proj_y = (const float *)take(4 * M);
            P.proj_xr = (const float *)take(4 * M); P.proj_yr = (const float *)take(4 * M);
            P.descriptors = take(32 * M); P.observations = (const int *)take(4 * M);
            if ((size_t)(p - pb.data()) != pb.size()) {
                std::fprintf(stderr, "points file has the wrong size\n");
                return 2;
            }
            const std::vector<float> sf = left.GetScaleFactors();
            std::vector<int> holder(keysL.size(), -1);
            ft_frame_view F{};
            F.N = (int)keysL.size(); F.Nleft = -1;
            F.mnMinX = 0.f; F.mnMinY = 0.f; F.mnMaxX = (float)W; F.mnMaxY = (float)H;  // Frame::ComputeImageBounds without distortion
            F.grid_inv_w = 64.f / (F.mnMaxX - F.mnMinX); F.grid_inv_h = 48.f / (F.mnMaxY - F.mnMinY);  // FRAME_GRID_COLS / ROWS
            F.mbf = mbf; F.mb = mb;
            F.keys = keysL.data(); F.descriptors = descL.data(); F.uright = mvuRight.data(); F.holder_obs = holder.data();
            F.scale_factors = sf.data(); F.nlevels = (int)sf.size();
            std::vector<int> assign, raw[10];
            for (auto &r : raw) r.assign(M ? M : 1, 0);
            const int nm = fasttrack::KernelController::launchSearchLocalPointsKernel(
                ctx, F, P, th, 0.8f, assign, raw[0].data(), raw[1].data(), raw[2].data(), raw[3].data(), raw[4].data(), raw[5].data(),
                raw[6].data(), raw[7].data(), raw[8].data(), raw[9].data());
            std::printf("SearchByProjection: %d matches of %d points\n", nm, P.M);
            std::vector<int> nmv = {nm};
            dump(out, nmv); dump(out, assign); dump(out, holder);
            for (int k : {2, 4}) dump(out, raw[k]);  // h_bestDist, h_bestIdx
        }
        std::fclose(out);
    } catch (const fasttrack::Error &e) {
        std::fprintf(stderr, "%s (status %d)\n", e.what(), e.status);
        return 1;
    }
    return 0;
}
