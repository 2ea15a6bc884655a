// Device-side views used by the projection-search kernels (kernels_search.hip / search.cpp).
#pragma once

#include "ft_internal.h"

struct FtDevFrame {
    int N, Nleft;
    float mnMinX, mnMinY, mnMaxX, mnMaxY, invW, invH, mbf, mb;
    const ft_keypoint *keys, *keysR;  // device
    const uint8_t *desc;              // device N x 32
    const float *uright;              // device or null
    const int *holderObs;             // device [N] pre-call Observations() of mvpMapPoints[i] (-1 = none)
    const int *l2r, *r2l;             // device or null
    int camModel;
    float cam[8];
    float Trl[12];
    float sf[FT_MAX_LEVELS];
    int nlevels;
};

struct FtDevLocalPoints {
    int M;
    const uint8_t *skip, *inView, *inViewR;
    const int *level, *levelR;
    const float *viewCos, *viewCosR, *projX, *projY, *projXR, *projYR;
    const uint8_t *desc;
};

struct FtDevLastPoints {
    int N;
    const uint8_t *valid;
    const float *worldPos;
    const uint8_t *desc;
    const int *octave;
};

// writer lists of the previous pass: head[kp] -> slot s (= 4*point + write kind), next[s]
struct FtClaims {
    const int *head, *next;  // writer lists of the previous pass
    const int *obs;          // Observations() per map point
    // housekeeping of the claim iteration, done by the search kernel itself so that a pass is two launches:
    int *headNext;           // the list heads this pass's k_build_claims will fill: reset to -1 here
    int nKp;
    int *changedCur;         // this pass's "something changed" flag: reset to 0 here
    const int *changedPrev;  // the previous pass's flag (null for the first pass of a burst): 0 = fixed point reached,
                             // this pass would reproduce its input and returns at once
};

struct FtPose {
    float m[12];
};

struct FtLocalRaw {
    int *bestDist, *bestDist2, *bestLevel, *bestLevel2, *bestIdx;
    int *bestDistR, *bestDist2R, *bestLevelR, *bestLevel2R, *bestIdxR;
};
struct FtLastRaw {
    int *bestDist, *bestIdx, *bestDistR, *bestIdxR;
};

// Frame::isInFrustum inputs / outputs (kernels_search.hip k_frustum)
struct FtDevMapPoints {
    int M;
    const uint8_t *skip;  // may be null
    const float *worldPos, *normal, *maxDist, *minDist;
};
struct FtFrustumPose {
    float R[2][9], t[2][3], twc[2][3];  // [0] left camera (mRcw, mtcw, mOw), [1] right camera (Frame.cc:1314-1320)
};
struct FtFrustumOut {
    uint8_t *inView, *inViewR;
    int *level, *levelR;
    float *viewCos, *viewCosR, *projX, *projY, *projXR, *projYR, *depth, *depthR;
    uint8_t *searchSkip;  // may be null: the `continue` conditions of ORBmatcher.cc:66-74 for the search that follows
    int *count;           // nToMatch (atomic; zeroed by the launcher)
};

int ft_launch_fill_i32(hipStream_t st, int *p, int n, int v);
int ft_launch_features_in_area(hipStream_t st, const FtDevFrame &F, int nq, const float *qx, const float *qy, const float *qr,
                               const int *qmin, const int *qmax, const uint8_t *qright, const int *offsets,
                               unsigned *outKeys, int *outCount);
int ft_launch_frustum(hipStream_t st, const FtDevFrame &F, const FtFrustumPose &T, const FtDevMapPoints &P,
                      float viewingCosLimit, float logScaleFactor, int farPoints, float thFar, const FtFrustumOut &O);
int ft_launch_search_local(hipStream_t st, const FtDevFrame &F, const FtDevLocalPoints &P, const FtClaims &C, float th,
                           float nnRatio, int *res, const FtLocalRaw &raw);
int ft_launch_search_last(hipStream_t st, const FtDevFrame &F, const FtDevLastPoints &L, const FtClaims &C,
                          const FtPose &Tcw, float th, int forward, int backward, int *res, const FtLastRaw &raw);
int ft_launch_build_claims(hipStream_t st, const int *res, const int *prevRes, int nPoints, int *head, int *next,
                           int *changed, const int *changedPrev);
