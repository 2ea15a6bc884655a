/*
 * orb_oracle.h - CPU restatement of the reference's *CPU branch* of the tracking front end.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the checker /
 * the timed CPU baseline ("restated CPU path", kind "port").  The product (fasttrack_amd/) never
 * links, imports or falls back to this code.
 *
 * PARITY UNPINNED: the reference has no tests, golden vectors or fixtures for this path
 *  and cannot be built here (needs OpenCV >= 4.4, Eigen, Pangolin, Boost,
 * CUDA; none present, no network).  The arithmetic that lives inside OpenCV (cv::resize, cv::FAST,
 * cv::GaussianBlur, cv::fastAtan2, cvRound, BFMatcher) is restated from the published OpenCV 4.x
 * algorithms; each such routine is isolated in one function below.  What IS
 * pinned against reference data: the rBRIEF pattern, the FAST ring offsets, the FAST-9 arc table
 * (reference src/fast.cu:24), umax, the per-level quota formula and TH_HIGH/TH_LOW/HISTO_LENGTH
 * (tests/test_oracle_tables.py, fixtures under tests/golden/).
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field order and size (28 B) as cv::KeyPoint: pt.x pt.y size angle response octave class_id */
typedef struct orc_keypoint {
    float x, y, size, angle, response;
    int octave, class_id;
} orc_keypoint;

/* POD view of the parts of ORB_SLAM3::Frame the matchers read (include/Frame.h). */
typedef struct orc_frame {
    int N;     /* all keypoints (left + right when Nleft != -1) */
    int Nleft; /* -1: mono / rectified stereo (keys = mvKeysUn); else fisheye stereo (keys = mvKeys) */
    float mnMinX, mnMinY, mnMaxX, mnMaxY;
    float grid_inv_w, grid_inv_h; /* mfGridElementWidthInv / HeightInv (Frame.cc:184-185) */
    float mbf, mb;
    const orc_keypoint *keys;       /* [Nleft == -1 ? N : Nleft] */
    const orc_keypoint *keys_right; /* [N - Nleft] when Nleft != -1 */
    const uint8_t *descriptors;     /* N x 32 */
    const float *uright;            /* mvuRight[N] when Nleft == -1 (NULL = all -1) */
    int *holder_obs;                /* in/out [N]: Observations() of mvpMapPoints[i], -1 when NULL */
    const int *left_to_right;       /* mvLeftToRightMatch[Nleft] or NULL */
    const int *right_to_left;       /* mvRightToLeftMatch[N-Nleft] or NULL */
    int cam_model;                  /* 0 pinhole, 1 KannalaBrandt8 */
    float cam[8];                   /* fx fy cx cy k1..k4 */
    float Trl[12];                  /* row-major 3x4, GetRelativePoseTrl() */
    const float *scale_factors;     /* mvScaleFactors[nlevels] */
    int nlevels;
} orc_frame;

/* ---- tables (ORBextractor.cc:393-499) ---- */
void orc_scale_factors(float scale_factor, int nlevels, float *sf, float *inv_sf);
void orc_features_per_level(int nfeatures, float scale_factor, int nlevels, int *out);
void orc_umax(int *out16);
void orc_level_sizes(int w, int h, float scale_factor, int nlevels, int *lw, int *lh);
const signed char *orc_pattern(void); /* 1024 int8 */

/* ---- OpenCV-boundary primitives ---- */
int orc_cv_round_f(float v);
int orc_cv_round_d(double v);
void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh,
                          int dstride);
void orc_gaussian_blur7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride);
void orc_gaussian_kernel7_fixed(int *k7); /* 8-bit fixed-point taps of GaussianBlur(7x7, sigma 2) */
/* cv::FAST(img, kps, threshold, nonmax) TYPE_9_16; out = (x,y,score) triples, row-major order */
int orc_fast9_16(const uint8_t *img, int w, int h, int stride, int threshold, int nonmax, int *xys, int cap);
int orc_fast_is_corner(const uint8_t *center, int stride, int threshold);
int orc_fast_corner_score(const uint8_t *center, int stride, int threshold);
int orc_fast_mask_has_arc9(unsigned mask16);
float orc_fast_atan2(float y, float x);
float orc_ic_angle(const uint8_t *img, int stride, float x, float y);
void orc_brief_descriptor(const uint8_t *blurred, int stride, float x, float y, float angle_deg,
                          uint8_t *desc32);
int orc_descriptor_distance(const uint8_t *a, const uint8_t *b);

/* ---- octree (ORBextractor.cc:510-566, 626-641, 660-884) ---- */
/* xys: n (x,y,score) candidates in emission order, coordinates relative to (minX,minY).
 * out_idx: indices of retained candidates in result (list) order.  returns count. */
int orc_distribute_octree(const int *xys, int n, int minX, int maxX, int minY, int maxY, int N, int *out_idx,
                          int cap);

/* ---- extractor (ORBextractor::operator(), CPU branch) ---- */
typedef struct orc_extractor orc_extractor;
orc_extractor *orc_extractor_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th);
void orc_extractor_destroy(orc_extractor *ex);
/* returns total keypoints (or -1 on empty input); *n_mono = return value of operator() */
int orc_extract(orc_extractor *ex, const uint8_t *img, int w, int h, int stride, int lap0, int lap1,
                orc_keypoint *kps, uint8_t *desc, int cap, int *n_mono);
/* stage-by-stage introspection of the last orc_extract call */
int orc_get_level(const orc_extractor *ex, int level, const uint8_t **data, int *w, int *h, int *stride);
int orc_get_blurred(const orc_extractor *ex, int level, const uint8_t **data, int *w, int *h, int *stride);
int orc_get_candidates(const orc_extractor *ex, int level, int *xys, int cap); /* pre-octree, rel. to border */
int orc_get_level_keypoints(const orc_extractor *ex, int level, orc_keypoint *out, uint8_t *desc, int cap);
/* pyramid only */
int orc_compute_pyramid(orc_extractor *ex, const uint8_t *img, int w, int h, int stride);

/* ---- Frame::ComputeStereoMatches (Frame.cc:835-1005) ---- */
/* exL/exR hold the pyramids of the last extract.  best_dist[i] = SAD of accepted matches else -1;
 * hamming_idx[i] = best right index of the Hamming stage (-1 none) for stage-wise checks.
 * apply_median_cut != 0 applies Frame.cc:991-1004.  returns number of surviving matches. */
int orc_stereo_match(const orc_extractor *exL, const orc_extractor *exR, const orc_keypoint *keysL, int nL,
                     const orc_keypoint *keysR, int nR, const uint8_t *descL, const uint8_t *descR, float mbf,
                     float mb, float *uright, float *depth, int *best_dist, int *hamming_idx,
                     int apply_median_cut);

/* ---- Frame::ComputeStereoFishEyeMatches matching part (Frame.cc:1231-1255) ---- */
/* matches[i] = train index passing Lowe 0.7 ratio else -1; best/second distances also returned */
int orc_fisheye_match(const uint8_t *descL, int nL, const uint8_t *descR, int nR, int *matches, int *best,
                      int *second);

/* ---- Frame grid (Frame.cc:409-440, 681-759) ---- */
int orc_features_in_area(const orc_frame *F, float x, float y, float r, int min_level, int max_level,
                         int right, int *out, int cap);

/* ---- ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>, th, bFar, thFar) (ORBmatcher.cc:49-225) */
typedef struct orc_local_points {
    int M;
    const uint8_t *skip; /* 1: (!inView && !inViewR) || far || isBad()  -> continue */
    const uint8_t *in_view, *in_view_r;
    const int *level, *level_r;
    const float *view_cos, *view_cos_r;
    const float *proj_x, *proj_y, *proj_xr; /* mTrackProjX/Y, mTrackProjXR (also used as right-cam X) */
    const float *proj_yr;
    const uint8_t *descriptors; /* M x 32 */
    const int *observations;    /* pMP->Observations() */
} orc_local_points;
/* assign[i] (size F->N): index of the map point written to F.mvpMapPoints[i] by this call, else -1.
 * raw outputs (size M each, may be NULL) mirror the arrays of launchSearchLocalPointsKernel. */
int orc_search_local_points(orc_frame *F, const orc_local_points *P, float th, float nn_ratio, int *assign,
                            int *best_dist, int *best_dist2, int *best_level, int *best_level2, int *best_idx,
                            int *best_dist_r, int *best_dist2_r, int *best_level_r, int *best_level2_r,
                            int *best_idx_r);

/* ---- ORBmatcher::SearchByProjection(Cur, Last, th, bMono) (ORBmatcher.cc:1775-1990) ---- */
typedef struct orc_last_points {
    int N;                      /* LastFrame.N */
    const uint8_t *valid;       /* mvpMapPoints[i] != NULL && !mvbOutlier[i] */
    const float *world_pos;     /* N x 3 */
    const uint8_t *descriptors; /* N x 32, pMP->GetDescriptor() */
    const int *observations;
    const int *octave;   /* last-frame keypoint octave of i */
    const float *angle;  /* last-frame keypoint angle of i */
} orc_last_points;
/* Tcw: row-major 3x4.  assign[i] (size Cur->N) = last-frame index whose map point was written, else -1
 * (after the rotation-consistency filter when check_orientation).  returns nmatches. */
int orc_search_last_frame(orc_frame *Cur, const orc_last_points *L, const float *Tcw, float th, int forward,
                          int backward, int check_orientation, int *assign, int *best_dist, int *best_idx,
                          int *best_dist_r, int *best_idx_r);
/* the same search with the poses as Sophus::SE3f holds them (unit quaternion x y z w + translation) and applies them to a point
 * (Thirdparty/Sophus/sophus/so3.hpp:358-367, se3.hpp:321-324): what the CPU branch evaluates at src/ORBmatcher.cc:1805 / :1900.
 * q_trl / t_trl: GetRelativePoseTrl(), may be NULL when Cur->Nleft == -1. */
int orc_search_last_frame_se3(orc_frame *Cur, const orc_last_points *L, const float *q_tcw, const float *t_tcw, const float *q_trl,
                              const float *t_trl, float th, int forward, int backward, int check_orientation, int *assign,
                              int *best_dist, int *best_idx, int *best_dist_r, int *best_idx_r);
void orc_se3_transform(const float q[4], const float t[3], const float p[3], float y[3]);
void orc_three_maxima(const int *hist_sizes, int L, int *ind1, int *ind2, int *ind3);

/* ---- Frame::ComputeStereoFishEyeMatches complete (src/Frame.cc:1231-1271): 2-NN + ratio test, then
 *      KannalaBrandt8::TriangulateMatches per surviving pair (src/CameraModels/KannalaBrandt8.cpp:306-372:
 *      unproject by Newton iteration :114-143, parallax, Triangulate :397-409, depth and reprojection tests).
 *      Eigen::JacobiSVD is not available here: the null vector of the 4x4 system is computed by a one-sided
 *      Jacobi SVD in double and narrowed to float (the reference's float JacobiSVD agrees to its own rounding
 *      error) - floating-point outputs of this function are compared with a tolerance. ---- */
typedef struct orc_fisheye_rig {
    float cam1[8], cam2[8]; /* fx fy cx cy k1..k4 of mpCamera / mpCamera2 */
    float precision;        /* KannalaBrandt8::precision (1e-6) */
    float Rlr[9], tlr[3];   /* mRlr, mtlr */
} orc_fisheye_rig;
/* code[i]: the value TriangulateMatches returns (-1 parallax, -2 z1 <= 0, -3 z2 <= 0, -4 / -5 reprojection, else z1);
 * p3d receives x3D when code > 0 */
void orc_kb8_triangulate(const orc_fisheye_rig *rig, int n, const float *xy1, const float *xy2, const float *sigma1,
                         const float *sigma2, float *code, float *p3d);
/* keysL / keysR: the lapping-area subsets like descL / descR; level_sigma2 = mvLevelSigma2.
 * matches[i] = j or -1, depth[i] = z1 or -1, p3d[3i..] = mvStereo3Dpoints; returns nMatches. */
int orc_fisheye_stereo(const orc_fisheye_rig *rig, const uint8_t *descL, const orc_keypoint *keysL, int nL,
                       const uint8_t *descR, const orc_keypoint *keysR, int nR, const float *level_sigma2,
                       int *matches, float *depth, float *p3d);

/* ---- Frame::isInFrustum / isInFrustumChecks + MapPoint::PredictScale
 *      (src/Frame.cc:536-610, 1308-1382; src/MapPoint.cc:531-546): the per-map-point step in front of
 *      SearchByProjection (src/Tracking.cc:3503-3522) ---- */
typedef struct orc_frame_pose {
    float Rcw[9]; /* mRcw row-major */
    float tcw[3]; /* mtcw */
    float Ow[3];  /* mOw */
    float tlr[3]; /* mTlr.translation() (two-camera frames only) */
} orc_frame_pose;
typedef struct orc_map_points {
    int M;
    const uint8_t *skip;       /* mnLastFrameSeen == frame id || isBad(): not tested (Tracking.cc:3507-3510) */
    const float *world_pos;    /* M x 3 GetWorldPos() */
    const float *normal;       /* M x 3 GetNormal() */
    const float *max_distance; /* mfMaxDistance (GetMaxDistanceInvariance = 1.2f * this) */
    const float *min_distance; /* mfMinDistance (GetMinDistanceInvariance = 0.8f * this) */
} orc_map_points;
/* outputs (size M): the MapPoint tracking fields; values the reference leaves untouched are written as
 * level -1, view_cos 0, proj -1, depth 0.  returns the number of points with in_view || in_view_r (nToMatch). */
int orc_is_in_frustum(const orc_frame *F, const orc_frame_pose *T, const orc_map_points *P, float viewing_cos_limit,
                      float log_scale_factor, uint8_t *in_view, uint8_t *in_view_r, int *level, int *level_r,
                      float *view_cos, float *view_cos_r, float *proj_x, float *proj_y, float *proj_xr,
                      float *proj_yr, float *depth, float *depth_r);

/* ---- Frame::ComputeBoW (src/Frame.cc:762-769): DBoW2 TemplatedVocabulary<FORB::TDescriptor, FORB>
 *      (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h) restated: loadFromTextFile :1338-1423, transform of a frame
 *      :1127-1194, of one feature :1208-1253, FORB::distance FORB.cpp:81-101, BowVector.cpp:32-85,
 *      FeatureVector.cpp:31-45 ---- */
typedef struct orc_vocabulary orc_vocabulary;
/* node 0 is the root; children are appended in ascending node id, words numbered in node order */
orc_vocabulary *orc_vocabulary_create(int k, int L, int scoring, int weighting, int n_nodes, const int *parent,
                                      const uint8_t *is_leaf, const uint8_t *descriptors, const double *weights);
orc_vocabulary *orc_vocabulary_load_text(const char *path);
void orc_vocabulary_destroy(orc_vocabulary *v);
int orc_vocabulary_nodes(const orc_vocabulary *v);
int orc_vocabulary_words(const orc_vocabulary *v);
/* per feature: word_ids, node_ids, weights (each may be NULL); BowVector as (ids, values) in map order, returns its
 * size in *n_bow; FeatureVector in CSR form (fv_nodes, fv_offsets[n_fv + 1], fv_features[n]) */
void orc_bow_transform(const orc_vocabulary *v, const uint8_t *descriptors, int n, int levelsup, unsigned *word_ids,
                       unsigned *node_ids, double *weights, unsigned *bow_ids, double *bow_values, int *n_bow,
                       unsigned *fv_nodes, int *fv_offsets, unsigned *fv_features, int *n_fv);

/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) - src/ORBmatcher.cc:322-524 (TrackReferenceKeyFrame,
 * Relocalization).  A side = its FeatureVector in the CSR form of orc_bow_transform, its descriptors and keypoint angles;
 * kf has_point[i] != 0 <=> vpMapPointsKF[i] is a good map point; f_nleft = Frame::Nleft (-1: one camera; angles / descriptors
 * of the right camera follow the left ones).  matches[i] = the keyframe feature whose map point vpMapPointMatches[i] holds,
 * or -1.  Returns nmatches. */
typedef struct orc_bow_side {
    int n;                      /* features */
    int n_nodes;                /* FeatureVector entries */
    const unsigned *fv_nodes;   /* [n_nodes] ascending */
    const int *fv_offsets;      /* [n_nodes + 1] */
    const unsigned *fv_features; /* [fv_offsets[n_nodes]] */
    const uint8_t *descriptors; /* n x 32 */
    const float *angles;        /* [n] cv::KeyPoint::angle */
} orc_bow_side;
int orc_search_by_bow(const orc_bow_side *kf, const uint8_t *kf_has_point, const orc_bow_side *f, int f_nleft, float nn_ratio,
                      int check_orientation, int *matches);

#ifdef __cplusplus
}
#endif
#endif
