// fasttrack_amd.hpp - header-only C++ mirror of the reference's operator interface for the hot path,
// on top of the C ABI (fasttrack_amd.h).  Same names, argument meaning and return conventions as
//   ORB_SLAM3::ORBextractor           (reference include/ORBextractor.h:100-197)
//   KernelController::launch*         (reference include/Kernels/KernelController.h:13-55)
// but free of OpenCV / Eigen: images are (pointer, width, height, stride), keypoints are any struct
// with cv::KeyPoint's 28-byte layout (ft_keypoint, or cv::KeyPoint itself in an ORB-SLAM3 build), and
// descriptors are row-major N x 32 bytes (what a CV_8U cv::Mat holds).  INTEGRATION.md shows the
// few lines that bind it to cv::Mat / ORB_SLAM3::Frame.
//
// Errors: the reference prints and exit()s (src/Kernels/CudaUtils.cu:17-22); here every failure throws
// fasttrack::Error carrying ft_last_error().  There is no CPU run mode to fall back to: the five
// setGPURunMode flags of the reference have no equivalent.
#pragma once

#include <cstring>
#include <map>
#include <stdexcept>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include "fasttrack_amd.h"

namespace fasttrack {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string &m) : std::runtime_error("fasttrack_amd: " + m), status(s) {}
};
inline void check(int status) {
    if (status != FT_OK) throw Error(status, ft_last_error());
}

// KernelController::setCUDADevice + initializeKernels + shutdownKernels + saveKernelsStats
class Context {
public:
    explicit Context(int device = 0, int hostThreads = 0) { check(ft_context_create(device, hostThreads, &h_)); }
    // ft_context_destroy refuses while extractors, front ends or tracked frames of the context are alive (they hold its
    // streams): a destructor cannot throw, so that ordering mistake is reported on stderr instead of leaking silently.
    // Declare the Context BEFORE the objects that use it (members are destroyed in reverse order).
    ~Context() {
        if (h_ && ft_context_destroy(h_) != FT_OK)
            std::fprintf(stderr, "fasttrack_amd: context not destroyed (leaked): %s\n", ft_last_error());
    }
    void setOption(const char *name, int value) { check(ft_context_set_option(h_, name, value)); }
    int getOption(const char *name) const {
        int v = 0;
        check(ft_context_get_option(h_, name, &v));
        return v;
    }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    ft_context *handle() const { return h_; }
    void synchronize() { check(ft_context_synchronize(h_)); }
    void saveKernelsStats(const std::string &file_path) { check(ft_context_save_stats(h_, file_path.c_str())); }

private:
    ft_context *h_ = nullptr;
};

// Non-owning 8-bit single-channel image (cv::Mat of type CV_8UC1: data, cols, rows, step[0]).
struct ImageView {
    const uint8_t *data = nullptr;
    int cols = 0, rows = 0, step = 0;
    bool empty() const { return !data || cols <= 0 || rows <= 0; }
};

template <class KeyPointT = ft_keypoint>
class ORBextractor {
    static_assert(sizeof(KeyPointT) == sizeof(ft_keypoint), "KeyPointT must have cv::KeyPoint's 28-byte layout");

public:
    // ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int imageWidth, int imageHeight)
    ORBextractor(Context &ctx, int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                 int imageWidth, int imageHeight, int maxBatch = 1)
        : nfeatures(nfeatures), scaleFactor(scaleFactor), nlevels(nlevels), iniThFAST(iniThFAST), minThFAST(minThFAST) {
        check(ft_extractor_create(ctx.handle(), nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, imageWidth,
                                  imageHeight, maxBatch, &h_));
        mvScaleFactor.resize(nlevels);
        mvInvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels);
        mvInvLevelSigma2.resize(nlevels);
        check(ft_extractor_scale_factors(h_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                                         mvInvLevelSigma2.data()));
    }
    ~ORBextractor() { ft_extractor_destroy(h_); }
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // int operator()(InputArray image, InputArray mask, vector<KeyPoint>& keypoints, OutputArray descriptors,
    //                std::vector<int>& vLappingArea)        (the mask is ignored by the reference too)
    // descriptors: resized to keypoints.size() * 32 bytes (rows of a CV_8U N x 32 matrix).
    // Returns the reference's monoIndex, or -1 for an empty image (ORBextractor.cc:1360-1361).
    int operator()(const ImageView &image, std::vector<KeyPointT> &keypoints, std::vector<uint8_t> &descriptors,
                   const std::vector<int> &vLappingArea) {
        if (image.empty()) return -1;
        const int cap = ft_extractor_max_keypoints(h_);
        keypoints.resize(cap);
        descriptors.resize((size_t)cap * 32);
        int n = 0, nMono = 0;
        const int lap0 = vLappingArea.size() > 0 ? vLappingArea[0] : 0, lap1 = vLappingArea.size() > 1 ? vLappingArea[1] : 0;
        const int st = ft_extract(h_, image.data, image.cols, image.rows, image.step, lap0, lap1,
                                  reinterpret_cast<ft_keypoint *>(keypoints.data()), descriptors.data(), cap, &n, &nMono);
        if (st == FT_ERR_EMPTY) return -1;
        check(st);
        keypoints.resize(n);
        descriptors.resize((size_t)n * 32);
        return nMono;
    }

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // mvImagePyramid[level]: host copy of one level of the last image (the reference keeps bordered
    // cv::Mat ROIs; nothing on the path reads the border, so tight rows are returned)
    std::vector<uint8_t> imagePyramidLevel(int level, int &cols, int &rows, int slot = 0) {
        check(ft_extractor_level_size(h_, level, &cols, &rows));
        std::vector<uint8_t> out((size_t)cols * rows);
        check(ft_extractor_download_level(h_, slot, level, out.data(), cols));
        return out;
    }
    // GetGPUPyramid(): device pointer + pitch of a level
    const uint8_t *GetGPUPyramid(int level, int &pitch, int slot = 0) {
        const uint8_t *p = nullptr;
        check(ft_extractor_device_level(h_, slot, level, &p, &pitch));
        return p;
    }
    ft_extractor *handle() const { return h_; }

    const int nfeatures;
    const float scaleFactor;
    const int nlevels, iniThFAST, minThFAST;

private:
    ft_extractor *h_ = nullptr;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

// Static facade with the reference's launch* names.  Outputs follow the reference's own conventions:
// vectors are resized by the callee, int arrays are caller-allocated.
struct KernelController {
    // launchStereoMatchKernel(vRowIndices, d_pyrL, d_pyrR, pyrL, pyrR, keysL, keysR, descL, descR, minD, maxD,
    //                         thOrbDist, mbf, pyramidOnGpu, vDistIdx, mvuRight, mvDepth)     KernelController.h:31-36
    // The row table and the pyramids live on the device already (slot `slot` of both extractors), minD/maxD
    // follow from mb/mbf as in Frame.cc:864-867.  vDistIdx receives (SAD, iL) of every refined match so the
    // caller's median cut (Frame.cc:1049-1062) runs unchanged when applyMedianCut is false.
    template <class KP>
    static void launchStereoMatchKernel(ORBextractor<KP> &left, ORBextractor<KP> &right, const std::vector<KP> &mvKeys,
                                        const std::vector<KP> &mvKeysRight, const uint8_t *mDescriptors,
                                        const uint8_t *mDescriptorsRight, float mbf, float mb, bool applyMedianCut,
                                        std::vector<std::pair<int, int>> &vDistIdx, std::vector<float> &mvuRight,
                                        std::vector<float> &mvDepth, int slot = 0) {
        const int N = (int)mvKeys.size(), Nr = (int)mvKeysRight.size();
        mvuRight.assign(N, -1.0f);
        mvDepth.assign(N, -1.0f);
        std::vector<int> sad(N > 0 ? N : 1, -1);
        int nm = 0;
        check(ft_stereo_match(left.handle(), right.handle(), slot, reinterpret_cast<const ft_keypoint *>(mvKeys.data()), N,
                              reinterpret_cast<const ft_keypoint *>(mvKeysRight.data()), Nr, mDescriptors,
                              mDescriptorsRight, mbf, mb, applyMedianCut ? 1 : 0, mvuRight.data(), mvDepth.data(),
                              sad.data(), &nm));
        vDistIdx.clear();
        for (int i = 0; i < N; i++)
            if (sad[i] >= 0) vDistIdx.push_back(std::make_pair(sad[i], i));
    }

    // launchFisheyeStereoMatchKernel(N, Nr, descL, descR, matches)                         KernelController.h:38
    static void launchFisheyeStereoMatchKernel(Context &ctx, int N, int Nr, const uint8_t *mDescriptors,
                                               const uint8_t *mDescriptorsRight, int *matches) {
        check(ft_fisheye_match(ctx.handle(), mDescriptors, N, mDescriptorsRight, Nr, matches, nullptr, nullptr));
    }

    // Frame::ComputeStereoFishEyeMatches complete (src/Frame.cc:1231-1271): ratio-tested 2-NN matches filtered by
    // KannalaBrandt8::TriangulateMatches.  Arrays are indexed like the lapping-area subsets.
    static int computeStereoFishEyeMatches(Context &ctx, const ft_fisheye_rig &rig, int N, int Nr, const uint8_t *descL,
                                           const ft_keypoint *keysL, const uint8_t *descR, const ft_keypoint *keysR,
                                           const std::vector<float> &mvLevelSigma2, std::vector<int> &leftToRight,
                                           std::vector<float> &mvDepth, std::vector<float> &mvStereo3Dpoints) {
        leftToRight.assign(N > 0 ? N : 1, -1);
        mvDepth.assign(N > 0 ? N : 1, -1.0f);
        mvStereo3Dpoints.assign(3 * (size_t)(N > 0 ? N : 1), 0.f);
        int nm = 0;
        check(ft_fisheye_stereo(ctx.handle(), &rig, descL, keysL, N, descR, keysR, Nr, mvLevelSigma2.data(),
                                (int)mvLevelSigma2.size(), leftToRight.data(), mvDepth.data(), mvStereo3Dpoints.data(), &nm));
        leftToRight.resize(N);
        mvDepth.resize(N);
        mvStereo3Dpoints.resize(3 * (size_t)N);
        return nm;
    }

    // launchSearchLocalPointsKernel(F, vmp, th, bFarPoints, thFarPoints, 10 x int*)        KernelController.h:40-42
    // F / P are the POD views of Frame and of the map points (INTEGRATION.md shows how they are filled);
    // in addition to the ten raw arrays the call returns the final assignment and nmatches, i.e. the
    // acceptance loop of ORBmatcher.cc:241-308 is included.
    static int launchSearchLocalPointsKernel(Context &ctx, ft_frame_view &F, const ft_local_points &P, float th,
                                             float mfNNratio, std::vector<int> &assign, int *h_bestLevel,
                                             int *h_bestLevel2, int *h_bestDist, int *h_bestDist2, int *h_bestIdx,
                                             int *h_bestLevelR, int *h_bestLevelR2, int *h_bestDistR, int *h_bestDistR2,
                                             int *h_bestIdxR) {
        assign.assign(F.N > 0 ? F.N : 1, -1);
        int nm = 0;
        check(ft_search_local_points(ctx.handle(), &F, &P, th, mfNNratio, assign.data(), &nm, h_bestDist, h_bestDist2,
                                     h_bestLevel, h_bestLevel2, h_bestIdx, h_bestDistR, h_bestDistR2,
                                     h_bestLevelR, h_bestLevelR2, h_bestIdxR));
        assign.resize(F.N);
        return nm;
    }

    // launchPoseEstimationKernel(Cur, Last, th, bForward, bBackward, Tcw, 4 x int*)        KernelController.h:44-46
    // Tcw: row-major 3x4 (the top rows of the reference's Eigen::Matrix4f).
    static int launchPoseEstimationKernel(Context &ctx, ft_frame_view &CurrentFrame, const ft_last_points &LastFrame,
                                          float th, bool bForward, bool bBackward, const float *Tcw,
                                          bool mbCheckOrientation, std::vector<int> &assign, int *h_bestDist,
                                          int *h_bestIdx2, int *h_bestDistR, int *h_bestIdxR2) {
        assign.assign(CurrentFrame.N > 0 ? CurrentFrame.N : 1, -1);
        int nm = 0;
        check(ft_search_last_frame(ctx.handle(), &CurrentFrame, &LastFrame, Tcw, th, bForward ? 1 : 0, bBackward ? 1 : 0,
                                   mbCheckOrientation ? 1 : 0, assign.data(), &nm, h_bestDist, h_bestIdx2, h_bestDistR,
                                   h_bestIdxR2));
        assign.resize(CurrentFrame.N);
        return nm;
    }
    // The same with the poses as the CPU branch multiplies with them (Sophus::SE3f: `Tcw * x3Dw`, ORBmatcher.cc:1805, is a
    // quaternion rotation): Tcw = se3Of(CurrentFrame.GetPose()), Trl = se3Of(CurrentFrame.GetRelativePoseTrl()) or nullptr for
    // one camera.  Bit-for-bit the CPU branch; the matrix overload above is the reference's GPU boundary.
    static int launchPoseEstimationKernel(Context &ctx, ft_frame_view &CurrentFrame, const ft_last_points &LastFrame,
                                          float th, bool bForward, bool bBackward, const ft_se3 &Tcw, const ft_se3 *Trl,
                                          bool mbCheckOrientation, std::vector<int> &assign, int *h_bestDist,
                                          int *h_bestIdx2, int *h_bestDistR, int *h_bestIdxR2) {
        assign.assign(CurrentFrame.N > 0 ? CurrentFrame.N : 1, -1);
        int nm = 0;
        check(ft_search_last_frame_se3(ctx.handle(), &CurrentFrame, &LastFrame, &Tcw, Trl, th, bForward ? 1 : 0, bBackward ? 1 : 0,
                                       mbCheckOrientation ? 1 : 0, assign.data(), &nm, h_bestDist, h_bestIdx2, h_bestDistR,
                                       h_bestIdxR2));
        assign.resize(CurrentFrame.N);
        return nm;
    }
};

// ft_se3 of a Sophus::SE3f (or anything with unit_quaternion().coeffs() = x y z w and translation()): the pose form of the
// CPU branch's point transforms
template <class SE3f>
inline ft_se3 se3Of(const SE3f &T) {
    ft_se3 r;
    const auto q = T.unit_quaternion().coeffs();
    const auto t = T.translation();
    for (int i = 0; i < 4; i++) r.q[i] = q[i];
    for (int i = 0; i < 3; i++) r.t[i] = t[i];
    return r;
}

// Frame::isInFrustum for all local map points (src/Frame.cc:536-610, 1308-1382; the loop of
// src/Tracking.cc:3503-3522).  The vectors receive the MapPoint tracking fields; returns nToMatch.
struct FrustumFields {
    std::vector<uint8_t> mbTrackInView, mbTrackInViewR;
    std::vector<int> mnTrackScaleLevel, mnTrackScaleLevelR;
    std::vector<float> mTrackViewCos, mTrackViewCosR, mTrackProjX, mTrackProjY, mTrackProjXR, mTrackProjYR, mTrackDepth,
        mTrackDepthR;
    ft_frustum_result view(int M) {
        const size_t m = M > 0 ? (size_t)M : 1;
        mbTrackInView.assign(m, 0); mbTrackInViewR.assign(m, 0);
        mnTrackScaleLevel.assign(m, -1); mnTrackScaleLevelR.assign(m, -1);
        for (auto *v : {&mTrackViewCos, &mTrackViewCosR, &mTrackProjX, &mTrackProjY, &mTrackProjXR, &mTrackProjYR, &mTrackDepth,
                        &mTrackDepthR})
            v->assign(m, 0.f);
        ft_frustum_result r;
        r.in_view = mbTrackInView.data(); r.in_view_r = mbTrackInViewR.data();
        r.level = mnTrackScaleLevel.data(); r.level_r = mnTrackScaleLevelR.data();
        r.view_cos = mTrackViewCos.data(); r.view_cos_r = mTrackViewCosR.data();
        r.proj_x = mTrackProjX.data(); r.proj_y = mTrackProjY.data();
        r.proj_xr = mTrackProjXR.data(); r.proj_yr = mTrackProjYR.data();
        r.depth = mTrackDepth.data(); r.depth_r = mTrackDepthR.data();
        return r;
    }
};

inline int isInFrustum(Context &ctx, const ft_frame_view &F, const ft_frame_pose &pose, const ft_map_points &P,
                       float viewingCosLimit, float mfLogScaleFactor, FrustumFields &out) {
    const ft_frustum_result r = out.view(P.M);
    int n = 0;
    check(ft_is_in_frustum(ctx.handle(), &F, &pose, &P, viewingCosLimit, mfLogScaleFactor, &r, &n));
    return n;
}

// Device-resident Frame for the projection searches (ft_tracked_frame_*): upload (or bind to a stereo front end
// slot) once per frame, then SearchByProjection(last frame) and SearchLocalPoints without re-marshalling.
class TrackedFrame {
public:
    TrackedFrame(Context &ctx, int maxKeypoints, int maxPoints) {
        check(ft_tracked_frame_create(ctx.handle(), maxKeypoints, maxPoints, &h_));
    }
    ~TrackedFrame() { ft_tracked_frame_destroy(h_); }
    TrackedFrame(const TrackedFrame &) = delete;
    TrackedFrame &operator=(const TrackedFrame &) = delete;
    void upload(const ft_frame_view &F) {
        check(ft_tracked_frame_upload(h_, &F));
        N_ = F.N;
    }
    void bindStereo(ft_stereo_frontend *fe, int slot, const ft_frame_view &meta) {
        check(ft_tracked_frame_bind_stereo(h_, fe, slot, &meta));
        N_ = meta.N;
    }
    // ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono)                       ORBmatcher.cc:1775
    int SearchByProjection(const ft_last_points &LastFrame, const float *Tcw, float th, bool bForward, bool bBackward,
                           bool mbCheckOrientation, std::vector<int> &assign) {
        assign.assign(N_ > 0 ? N_ : 1, -1);
        int nm = 0;
        check(ft_tracked_frame_search_last_frame(h_, &LastFrame, Tcw, th, bForward, bBackward, mbCheckOrientation,
                                                 assign.data(), &nm));
        assign.resize(N_);
        return nm;
    }
    int SearchByProjection(const ft_last_points &LastFrame, const ft_se3 &Tcw, const ft_se3 *Trl, float th, bool bForward,
                           bool bBackward, bool mbCheckOrientation, std::vector<int> &assign) {
        assign.assign(N_ > 0 ? N_ : 1, -1);
        int nm = 0;
        check(ft_tracked_frame_search_last_frame_se3(h_, &LastFrame, &Tcw, Trl, th, bForward, bBackward, mbCheckOrientation,
                                                     assign.data(), &nm));
        assign.resize(N_);
        return nm;
    }
    // Tracking::SearchLocalPoints: isInFrustum(pMP, 0.5) for every point + SearchByProjection(F, points, th, ...)
    int SearchLocalPoints(const ft_frame_pose &pose, const ft_map_points &P, float viewingCosLimit, float mfLogScaleFactor,
                          float th, float mfNNratio, bool bFarPoints, float thFarPoints, FrustumFields *fields,
                          int *nToMatch, std::vector<int> &assign) {
        assign.assign(N_ > 0 ? N_ : 1, -1);
        ft_frustum_result r;
        if (fields) r = fields->view(P.M);
        int nm = 0;
        check(ft_tracked_frame_track_local_map(h_, &pose, &P, viewingCosLimit, mfLogScaleFactor, th, mfNNratio, bFarPoints,
                                               thFarPoints, fields ? &r : nullptr, nToMatch, assign.data(), &nm));
        assign.resize(N_);
        return nm;
    }
    std::vector<int> holderObservations() {
        std::vector<int> h(N_ > 0 ? N_ : 1);
        check(ft_tracked_frame_holder_obs(h_, h.data()));
        h.resize(N_);
        return h;
    }
    ft_tracked_frame *handle() { return h_; }

private:
    ft_tracked_frame *h_ = nullptr;
    int N_ = 0;
};

// B frames per launch (ft_tracked_batch_*): the frames B camera streams deliver for one time step - or any B frames whose
// inputs the caller holds - searched through ONE set of launches; per frame the results of TrackedFrame's calls.  A batch has
// a stream of its own: several batches used from several host threads are several steps in flight.
class TrackedBatch {
public:
    TrackedBatch(Context &ctx, int maxFrames, int maxKeypoints, int maxPoints) {
        check(ft_tracked_batch_create(ctx.handle(), maxFrames, maxKeypoints, maxPoints, &h_));
    }
    ~TrackedBatch() { ft_tracked_batch_destroy(h_); }
    TrackedBatch(const TrackedBatch &) = delete;
    TrackedBatch &operator=(const TrackedBatch &) = delete;
    void upload(const std::vector<ft_frame_view> &frames) {
        check(ft_tracked_batch_upload(h_, (int)frames.size(), frames.data()));
        loaded(frames);
    }
    // the two-camera frames of the batch from what two extractors left in HBM (their last operator() batches): keypoints into
    // the reference's lapping-area order, the matching of Frame::ComputeStereoFishEyeMatches, the grids - all on the device
    // rig != nullptr: with KannalaBrandt8::TriangulateMatches on every matched pair (mvLevelSigma2 = levelSigma2), as
    // Frame::ComputeStereoFishEyeMatches; mvDepth / mvStereo3Dpoints / nMatches per frame on request
    void bindFisheye(ft_extractor *left, ft_extractor *right, int slot0, const int lapLeft[2], const int lapRight[2],
                     const std::vector<ft_frame_view> &meta, const ft_fisheye_rig *rig = nullptr, const float *levelSigma2 = nullptr,
                     int *const *leftToRight = nullptr, int *const *rightToLeft = nullptr, float *const *mvDepth = nullptr,
                     float *const *mvStereo3Dpoints = nullptr, int *nMatches = nullptr) {
        check(ft_tracked_batch_bind_fisheye(h_, left, right, slot0, (int)meta.size(), lapLeft[0], lapLeft[1], lapRight[0], lapRight[1],
                                            meta.data(), rig, levelSigma2, leftToRight, rightToLeft, mvDepth, mvStereo3Dpoints, nMatches));
        loaded(meta);
    }
    // ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) for every frame; Tcw: 12 floats per frame     ORBmatcher.cc:1775
    void SearchByProjection(const std::vector<ft_last_points> &LastFrames, const float *Tcw, float th, bool mbCheckOrientation,
                            std::vector<std::vector<int>> &assign, std::vector<int> &nmatches, const int *bForward = nullptr,
                            const int *bBackward = nullptr) {
        prepare(assign, nmatches);
        check(ft_tracked_batch_search_last_frame(h_, (int)N_.size(), LastFrames.data(), Tcw, th, bForward, bBackward, mbCheckOrientation,
                                                 ptrs_.data(), nmatches.data()));
        finish(assign);
    }
    // Tracking::SearchLocalPoints for every frame: isInFrustum + SearchByProjection(F, points, th, ...)              Tracking.cc:3472
    void SearchLocalPoints(const std::vector<ft_frame_pose> &poses, const std::vector<ft_map_points> &P, float viewingCosLimit,
                           float mfLogScaleFactor, float th, float mfNNratio, bool bFarPoints, float thFarPoints,
                           const ft_frustum_result *fields, std::vector<int> &nToMatch, std::vector<std::vector<int>> &assign,
                           std::vector<int> &nmatches) {
        prepare(assign, nmatches);
        nToMatch.assign(N_.size(), 0);
        check(ft_tracked_batch_track_local_map(h_, (int)N_.size(), poses.data(), P.data(), viewingCosLimit, mfLogScaleFactor, th, mfNNratio,
                                               bFarPoints, thFarPoints, fields, nToMatch.data(), ptrs_.data(), nmatches.data()));
        finish(assign);
    }
    // The same two searches in two halves (ft_tracked_batch_submit_* / ft_tracked_batch_wait): submit enqueues and returns, Wait() fills
    // assign / nmatches (/ nToMatch) - they and the point arrays (when those lie in pinned memory) must stay as they are until then.
    void SubmitSearchByProjection(const std::vector<ft_last_points> &LastFrames, const float *Tcw, float th, bool mbCheckOrientation,
                                  std::vector<std::vector<int>> &assign, std::vector<int> &nmatches, const int *bForward = nullptr,
                                  const int *bBackward = nullptr) {
        prepare(assign, nmatches);
        pending_ = &assign;
        check(ft_tracked_batch_submit_search_last_frame(h_, (int)N_.size(), LastFrames.data(), Tcw, th, bForward, bBackward, mbCheckOrientation,
                                                        ptrs_.data(), nmatches.data()));
    }
    void SubmitSearchLocalPoints(const std::vector<ft_frame_pose> &poses, const std::vector<ft_map_points> &P, float viewingCosLimit,
                                 float mfLogScaleFactor, float th, float mfNNratio, bool bFarPoints, float thFarPoints,
                                 const ft_frustum_result *fields, std::vector<int> &nToMatch, std::vector<std::vector<int>> &assign,
                                 std::vector<int> &nmatches) {
        prepare(assign, nmatches);
        nToMatch.assign(N_.size(), 0);
        pending_ = &assign;
        check(ft_tracked_batch_submit_track_local_map(h_, (int)N_.size(), poses.data(), P.data(), viewingCosLimit, mfLogScaleFactor, th,
                                                      mfNNratio, bFarPoints, thFarPoints, fields, nToMatch.data(), ptrs_.data(), nmatches.data()));
    }
    void Wait() {
        check(ft_tracked_batch_wait(h_));
        if (pending_) finish(*pending_);
        pending_ = nullptr;
    }
    std::vector<int> holderObservations(int frame) {
        std::vector<int> h(N_[frame] > 0 ? N_[frame] : 1);
        check(ft_tracked_batch_holder_obs(h_, frame, h.data()));
        h.resize(N_[frame]);
        return h;
    }
    ft_tracked_batch *handle() { return h_; }

private:
    void loaded(const std::vector<ft_frame_view> &frames) {
        N_.clear();
        for (const ft_frame_view &F : frames) N_.push_back(F.N);
    }
    void prepare(std::vector<std::vector<int>> &assign, std::vector<int> &nmatches) {
        assign.resize(N_.size());
        ptrs_.resize(N_.size());
        nmatches.assign(N_.size(), 0);
        for (size_t f = 0; f < N_.size(); f++) {
            assign[f].assign(N_[f] > 0 ? N_[f] : 1, -1);
            ptrs_[f] = assign[f].data();
        }
    }
    void finish(std::vector<std::vector<int>> &assign) {
        for (size_t f = 0; f < N_.size(); f++) assign[f].resize(N_[f]);
    }
    ft_tracked_batch *h_ = nullptr;
    std::vector<int> N_;
    std::vector<int *> ptrs_;
    std::vector<std::vector<int>> *pending_ = nullptr;
};

// ORBVocabulary (DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) for Frame::ComputeBoW: same call as
// mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4) (src/Frame.cc:762-769), with the tree walk on the
// device.  BowVector / FeatureVector are the std::map types of DBoW2 (BowVector.h:59, FeatureVector.h:24).
class ORBVocabulary {
public:
    typedef std::map<unsigned, double> BowVector;
    typedef std::map<unsigned, std::vector<unsigned>> FeatureVector;
    ORBVocabulary(Context &ctx, const std::string &textFile) { check(ft_vocabulary_load_text(ctx.handle(), textFile.c_str(), &h_)); }
    ~ORBVocabulary() { ft_vocabulary_destroy(h_); }
    ORBVocabulary(const ORBVocabulary &) = delete;
    ORBVocabulary &operator=(const ORBVocabulary &) = delete;
    // descriptors: N x 32 bytes (cv::Mat mDescriptors is continuous); onDevice: a pointer into HBM, e.g. from
    // ft_stereo_frontend_device_descriptors
    void transform(const uint8_t *descriptors, int N, BowVector &v, FeatureVector &fv, int levelsup, bool onDevice = false) {
        v.clear();
        fv.clear();
        if (N <= 0) return;
        std::vector<unsigned> ids(N), nodes(N), feats(N);
        std::vector<double> vals(N);
        std::vector<int> offs(N + 1);
        int nb = 0, nf = 0;
        check(ft_bow_transform(h_, descriptors, N, onDevice ? 1 : 0, levelsup, nullptr, nullptr, nullptr, ids.data(), vals.data(), N,
                               &nb, nodes.data(), offs.data(), feats.data(), N, &nf));
        for (int j = 0; j < nb; j++) v.emplace_hint(v.end(), ids[j], vals[j]);
        for (int j = 0; j < nf; j++)
            fv.emplace_hint(fv.end(), nodes[j], std::vector<unsigned>(feats.begin() + offs[j], feats.begin() + offs[j + 1]));
    }
    ft_vocabulary *handle() { return h_; }

private:
    ft_vocabulary *h_ = nullptr;
};

// ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches) (src/ORBmatcher.cc:322-524) on the
// std::map FeatureVectors of ORBVocabulary::transform.  kfHasPoint[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();
// kfAngles / fAngles = cv::KeyPoint::angle per feature (right-camera features after the left ones, as the reference indexes
// them).  matches[i] = index into vpMapPointsKF or -1: `vpMapPointMatches[i] = matches[i] < 0 ? nullptr : vpMapPointsKF[matches[i]]`.
// Returns nmatches.
inline int SearchByBoW(Context &ctx, const ORBVocabulary::FeatureVector &kfFeatVec, const uint8_t *kfDescriptors, int kfN,
                       const std::vector<uint8_t> &kfHasPoint, const float *kfAngles, const ORBVocabulary::FeatureVector &fFeatVec,
                       const uint8_t *fDescriptors, int fN, const float *fAngles, int fNleft, float nnRatio, bool checkOrientation,
                       std::vector<int> &matches) {
    struct Csr {
        std::vector<unsigned> nodes, feats;
        std::vector<int> offs;
    };
    auto flatten = [](const ORBVocabulary::FeatureVector &fv) {
        Csr c;
        c.offs.push_back(0);
        for (const auto &e : fv) {
            c.nodes.push_back(e.first);
            c.feats.insert(c.feats.end(), e.second.begin(), e.second.end());
            c.offs.push_back((int)c.feats.size());
        }
        return c;
    };
    const Csr k = flatten(kfFeatVec), f = flatten(fFeatVec);
    auto side = [](const Csr &c, int n, const uint8_t *d, const float *a) {
        ft_bow_side s;
        s.n = n;
        s.n_nodes = (int)c.nodes.size();
        s.fv_nodes = c.nodes.data();
        s.fv_offsets = c.offs.data();
        s.fv_features = c.feats.data();
        s.descriptors = d;
        s.angles = a;
        return s;
    };
    const ft_bow_side K = side(k, kfN, kfDescriptors, kfAngles), F = side(f, fN, fDescriptors, fAngles);
    matches.assign(fN > 0 ? fN : 1, -1);
    int nm = 0;
    check(ft_search_by_bow(ctx.handle(), &K, kfHasPoint.data(), &F, fNleft, nnRatio, checkOrientation ? 1 : 0, matches.data(), &nm));
    matches.resize(fN > 0 ? fN : 0);
    return nm;
}

}  // namespace fasttrack
