/*
 * orb_oracle.cpp - CPU restatement of the reference's CPU branch (see orb_oracle.h for the role and
 * the "parity unpinned" caveat).  Build: oracle/Makefile (g++ -O3 -ffp-contract=off, no fast-math).
 *
 * File:line citations are relative to /root/reference.
 */
#include "orb_oracle.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <utility>
#include <vector>

#include "orb_pattern.inc"

namespace {

constexpr int kPatchSize = 31;      // include/ORBextractor.h:29
constexpr int kHalfPatch = 15;      // :30
constexpr int kEdgeThreshold = 19;  // :31
constexpr int TH_HIGH = 100;        // src/ORBmatcher.cc:41
constexpr int TH_LOW = 50;          // :42
constexpr int HISTO_LENGTH = 30;    // :43
constexpr int GRID_ROWS = 48;       // include/Frame.h:46
constexpr int GRID_COLS = 64;       // :47

// ---- OpenCV scalar helpers: cvRound = round-half-to-even (cvtss2si / cvtsd2si) ----
inline int cvRoundF(float v) { return (int)lrintf(v); }
inline int cvRoundD(double v) { return (int)lrint(v); }
inline int cvFloorD(double v) { return (int)std::floor(v); }
inline int cvCeilD(double v) { return (int)std::ceil(v); }
inline short satShort(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
inline int reflect101(int i, int n) {
    // BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}

// FAST ring: offsets16 of cv::FAST == points[32] at src/ORBextractor.cc:418-419
const int kRing[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                          {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> px;  // tight rows
    const uint8_t *row(int y) const { return px.data() + (size_t)y * w; }
    uint8_t *row(int y) { return px.data() + (size_t)y * w; }
};

}  // namespace

struct orc_extractor {
    int nfeatures, nlevels, iniTh, minTh;
    float scaleFactor;
    std::vector<float> sf, invsf;
    std::vector<int> quota;
    int umax[16];
    // state of the last call
    std::vector<Image> pyr, blurred;
    std::vector<std::vector<int>> cand;                // per level (x,y,score) rel. to minBorder
    std::vector<std::vector<orc_keypoint>> levelKeys;  // post-octree, level coords, with angle
    std::vector<std::vector<uint8_t>> levelDesc;
};

extern "C" {

int orc_cv_round_f(float v) { return cvRoundF(v); }
int orc_cv_round_d(double v) { return cvRoundD(v); }
const signed char *orc_pattern(void) { return kOrcPattern31; }

// ------------------------------------------------------------------------------------------------
// tables - src/ORBextractor.cc:393-414 (scale chain), :454-465 (quotas), :478-493 (umax)
// ------------------------------------------------------------------------------------------------
void orc_scale_factors(float scale_factor, int nlevels, float *sf, float *inv_sf) {
    sf[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) sf[i] = sf[i - 1] * scale_factor;
    if (inv_sf)
        for (int i = 0; i < nlevels; i++) inv_sf[i] = 1.0f / sf[i];
}

void orc_features_per_level(int nfeatures, float scale_factor, int nlevels, int *out) {
    float factor = 1.0f / scale_factor;
    float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        out[level] = cvRoundF(nDesired);
        sum += out[level];
        nDesired *= factor;
    }
    out[nlevels - 1] = std::max(nfeatures - sum, 0);
}

void orc_umax(int *umax) {
    int v, v0;
    int vmax = cvFloorD(kHalfPatch * sqrt(2.f) / 2 + 1);
    int vmin = cvCeilD(kHalfPatch * sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (v = 0; v <= kHalfPatch; ++v) umax[v] = 0;
    for (v = 0; v <= vmax; ++v) umax[v] = cvRoundD(sqrt(hp2 - v * v));
    for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

// level size - src/ORBextractor.cc:1499-1500
void orc_level_sizes(int w, int h, float scale_factor, int nlevels, int *lw, int *lh) {
    std::vector<float> sf(nlevels), inv(nlevels);
    orc_scale_factors(scale_factor, nlevels, sf.data(), inv.data());
    for (int l = 0; l < nlevels; l++) {
        lw[l] = cvRoundF((float)w * inv[l]);
        lh[l] = cvRoundF((float)h * inv[l]);
    }
}

// ------------------------------------------------------------------------------------------------
// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for 8UC1 - SURVEY A.1 (call site
// src/ORBextractor.cc:1508).  Fixed point, INTER_RESIZE_COEF_BITS = 11.
// ------------------------------------------------------------------------------------------------
void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh,
                          int dstride) {
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    if (sw == 2 * dw && sh == 2 * dh) {
        // OpenCV switches exact 2x INTER_LINEAR decimation to the INTER_AREA fast path
        for (int y = 0; y < dh; y++) {
            const uint8_t *s0 = src + (size_t)(2 * y) * sstride, *s1 = s0 + sstride;
            for (int x = 0; x < dw; x++)
                dst[(size_t)y * dstride + x] =
                    (uint8_t)((s0[2 * x] + s0[2 * x + 1] + s1[2 * x] + s1[2 * x + 1] + 2) >> 2);
        }
        return;
    }
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloorD(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            xmax = std::min(xmax, dx);
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        ialpha[dx * 2] = satShort(cvRoundF((1.f - fx) * 2048));
        ialpha[dx * 2 + 1] = satShort(cvRoundF(fx * 2048));
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloorD(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[dy * 2] = satShort(cvRoundF((1.f - fy) * 2048));
        ibeta[dy * 2 + 1] = satShort(cvRoundF(fy * 2048));
    }
    std::vector<int> r0(dw), r1(dw);
    auto hresize = [&](int sy, std::vector<int> &D) {
        sy = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);
        const uint8_t *S = src + (size_t)sy * sstride;
        int dx = 0;
        for (; dx < xmax; dx++) {
            int sx = xofs[dx];
            D[dx] = S[sx] * ialpha[dx * 2] + S[sx + 1] * ialpha[dx * 2 + 1];
        }
        for (; dx < dw; dx++) D[dx] = S[xofs[dx]] * 2048;
    };
    for (int dy = 0; dy < dh; dy++) {
        hresize(yofs[dy], r0);
        hresize(yofs[dy] + 1, r1);
        const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int x = 0; x < dw; x++)
            D[x] = (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
    }
}

// ------------------------------------------------------------------------------------------------
// cv::GaussianBlur(8U, Size(7,7), 2, 2, BORDER_REFLECT_101) - SURVEY A.2 (call site
// src/ORBextractor.cc:1456-1457).  Fixed-point kernel with error diffusion, taps sum to 256.
// ------------------------------------------------------------------------------------------------
void orc_gaussian_kernel7_fixed(int *k7) {
    const int n = 7;
    const double sigma = 2.0;
    double v[7], sum = 0;
    const double scale2X = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        v[i] = std::exp(scale2X * x * x);
        sum += v[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) v[i] *= sum;
    double err = 0;
    int acc = 0;
    for (int i = 0; i < n / 2; i++) {
        double adj = v[i] * 256.0 + err;
        int q = cvRoundD(adj);
        err = adj - q;
        k7[i] = q;
        k7[n - 1 - i] = q;
        acc += q;
    }
    k7[n / 2] = 256 - 2 * acc;
}

void orc_gaussian_blur7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride) {
    int k[7];
    orc_gaussian_kernel7_fixed(k);
    std::vector<uint16_t> hbuf((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t *S = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            unsigned a = 0;
            for (int t = 0; t < 7; t++) a += (unsigned)k[t] * S[reflect101(x + t - 3, w)];
            hbuf[(size_t)y * w + x] = (uint16_t)a;  // <= 255*256, no saturation
        }
    }
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            unsigned a = 0;
            for (int t = 0; t < 7; t++) a += (unsigned)k[t] * hbuf[(size_t)reflect101(y + t - 3, h) * w + x];
            dst[(size_t)y * dstride + x] = (uint8_t)((a + 32768u) >> 16);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// cv::FAST TYPE_9_16 - SURVEY A.3 (call sites src/ORBextractor.cc:1157,1176)
// ------------------------------------------------------------------------------------------------
int orc_fast_mask_has_arc9(unsigned m) {
    m &= 0xffffu;
    unsigned d = m | (m << 16);
    for (int s = 0; s < 16; s++)
        if (((d >> s) & 0x1ffu) == 0x1ffu) return 1;
    return 0;
}

int orc_fast_is_corner(const uint8_t *c, int stride, int t) {
    const int v = c[0];
    unsigned dark = 0, bright = 0;
    for (int k = 0; k < 16; k++) {
        int p = c[kRing[k][0] + kRing[k][1] * stride];
        if (p - v < -t) dark |= 1u << k;
        if (p - v > t) bright |= 1u << k;
    }
    return orc_fast_mask_has_arc9(dark) || orc_fast_mask_has_arc9(bright);
}

// cornerScore<16> of OpenCV's fast_score.cpp, loops kept in its order
int orc_fast_corner_score(const uint8_t *c, int stride, int threshold) {
    const int K = 8, N = K * 3 + 1;
    int k, v = c[0];
    short d[N];
    for (k = 0; k < N; k++) d[k] = (short)(v - c[kRing[k & 15][0] + kRing[k & 15][1] * stride]);
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]);
        a = std::min(a, (int)d[k + 5]);
        a = std::min(a, (int)d[k + 6]);
        a = std::min(a, (int)d[k + 7]);
        a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]);
        b = std::max(b, (int)d[k + 4]);
        b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]);
        b = std::max(b, (int)d[k + 7]);
        b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

// OpenCV's high-speed test: a 9-arc contains one pixel of every opposite pair (k, k+8), so if some
// pair has neither pixel brighter nor darker the centre cannot be a corner.  Necessary condition only.
static inline bool fastQuickReject(const uint8_t *c, int stride, int t) {
    const int v = c[0];
    auto cls = [&](int k) -> int {
        int d = c[kRing[k][0] + kRing[k][1] * stride] - v;
        return d < -t ? 1 : (d > t ? 2 : 0);
    };
    int d = cls(0) | cls(8);
    if (d == 0) return true;
    d &= cls(2) | cls(10);
    d &= cls(4) | cls(12);
    d &= cls(6) | cls(14);
    if (d == 0) return true;
    d &= cls(1) | cls(9);
    d &= cls(3) | cls(11);
    d &= cls(5) | cls(13);
    d &= cls(7) | cls(15);
    return d == 0;
}

int orc_fast9_16(const uint8_t *img, int w, int h, int stride, int threshold, int nonmax, int *xys, int cap) {
    int n = 0;
    if (w < 7 || h < 7) return 0;
    std::vector<uint8_t> score((size_t)w * h, 0);
    std::vector<uint8_t> is((size_t)w * h, 0);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const uint8_t *c = img + (size_t)y * stride + x;
            if (fastQuickReject(c, stride, threshold)) continue;
            if (orc_fast_is_corner(c, stride, threshold)) {
                is[(size_t)y * w + x] = 1;
                score[(size_t)y * w + x] = nonmax ? (uint8_t)orc_fast_corner_score(c, stride, threshold) : 0;
            }
        }
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            if (!is[(size_t)y * w + x]) continue;
            int s = score[(size_t)y * w + x];
            if (nonmax) {
                const uint8_t *p = &score[(size_t)y * w + x];
                if (!(s > p[-1] && s > p[1] && s > p[-w - 1] && s > p[-w] && s > p[-w + 1] && s > p[w - 1] &&
                      s > p[w] && s > p[w + 1]))
                    continue;
            }
            if (n < cap) { xys[3 * n] = x; xys[3 * n + 1] = y; xys[3 * n + 2] = s; }
            n++;
        }
    return n;
}

// ------------------------------------------------------------------------------------------------
// cv::fastAtan2 - SURVEY A.4 (call site src/ORBextractor.cc:65)
// ------------------------------------------------------------------------------------------------
float orc_fast_atan2(float y, float x) {
    static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = std::abs(x), ay = std::abs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// IC_Angle - src/ORBextractor.cc:39-66
float orc_ic_angle(const uint8_t *img, int stride, float x, float y) {
    static int umax[16];
    static bool init = false;
    if (!init) { orc_umax(umax); init = true; }
    int m_01 = 0, m_10 = 0;
    const uint8_t *center = img + (size_t)cvRoundF(y) * stride + cvRoundF(x);
    for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m_10 += u * center[u];
    for (int v = 1; v <= kHalfPatch; ++v) {
        int v_sum = 0;
        int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

// computeOrbDescriptor - src/ORBextractor.cc:68-108.  `cos(angle)` / `sin(angle)` on a float under
// `using namespace std` (:34) bind to std::cos(float) / std::sin(float) = libm's cosf / sinf (SURVEY A.7
// misread this as the double routines).  The host libm is a third party the oracle calls, not restates:
// glibc 2.35 here and on the GPU box; its float routines are not correctly rounded, so another libm may
// differ in the last bit on a few per cent of angles.
void orc_brief_descriptor(const uint8_t *img, int stride, float x, float y, float angle_deg, uint8_t *desc) {
    const float factorPI = (float)(M_PI / 180.f);
    float angle = (float)angle_deg * factorPI;
    float a = cosf(angle), b = sinf(angle);
    const uint8_t *center = img + (size_t)cvRoundF(y) * stride + cvRoundF(x);
    const signed char *pat = kOrcPattern31;
    auto value = [&](int idx) -> int {
        const float px = (float)pat[2 * idx], py = (float)pat[2 * idx + 1];
        return center[cvRoundF(px * b + py * a) * stride + cvRoundF(px * a - py * b)];
    };
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int j = 0; j < 8; j++) {
            int t0 = value(2 * j), t1 = value(2 * j + 1);
            val |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

// ORBmatcher::DescriptorDistance - src/ORBmatcher.cc:2256-2272
int orc_descriptor_distance(const uint8_t *a, const uint8_t *b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        unsigned int v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Octree distribution - src/ORBextractor.cc:510-566 (DivideNode), :626-641 (compareNodes),
// :660-884 (DistributeOctTree).  Nodes are axis-aligned boxes (UL,UR,BL,BR collapse to x0,y0,x1,y1)
// holding candidate indices; list discipline (push_front / erase) and std::sort are kept because the
// output order and the tie behaviour depend on them.
// ------------------------------------------------------------------------------------------------
namespace {

struct OctNode {
    int x0, y0, x1, y1;
    std::vector<int> keys;
    bool noMore = false;
    std::list<OctNode>::iterator self;
};

struct OctCtx {
    const int *xys;
    float px(int i) const { return (float)xys[3 * i]; }
    float py(int i) const { return (float)xys[3 * i + 1]; }
    float resp(int i) const { return (float)xys[3 * i + 2]; }
};

void divideNode(const OctCtx &c, const OctNode &n, OctNode ch[4]) {
    const int halfX = (int)ceil(static_cast<float>(n.x1 - n.x0) / 2);
    const int halfY = (int)ceil(static_cast<float>(n.y1 - n.y0) / 2);
    const int mx = n.x0 + halfX, my = n.y0 + halfY;
    ch[0] = OctNode{n.x0, n.y0, mx, my, {}, false, {}};
    ch[1] = OctNode{mx, n.y0, n.x1, my, {}, false, {}};
    ch[2] = OctNode{n.x0, my, mx, n.y1, {}, false, {}};
    ch[3] = OctNode{mx, my, n.x1, n.y1, {}, false, {}};
    for (int q = 0; q < 4; q++) ch[q].keys.reserve(n.keys.size());
    for (int i : n.keys) {
        if (c.px(i) < mx) {
            if (c.py(i) < my) ch[0].keys.push_back(i);
            else ch[2].keys.push_back(i);
        } else if (c.py(i) < my)
            ch[1].keys.push_back(i);
        else
            ch[3].keys.push_back(i);
    }
    for (int q = 0; q < 4; q++)
        if (ch[q].keys.size() == 1) ch[q].noMore = true;
}

bool compareNodes(const std::pair<int, OctNode *> &e1, const std::pair<int, OctNode *> &e2) {
    if (e1.first < e2.first) return true;
    if (e1.first > e2.first) return false;
    return e1.second->x0 < e2.second->x0;
}

}  // namespace

extern "C" int orc_distribute_octree(const int *xys, int n, int minX, int maxX, int minY, int maxY, int N,
                                     int *out_idx, int cap) {
    OctCtx c{xys};
    int nIni = (int)round(static_cast<float>(maxX - minX) / (maxY - minY));
    if (nIni < 1) nIni = 1;  // reference divides by zero for very tall images; guarded here
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::list<OctNode> nodes;
    std::vector<OctNode *> ini(nIni);
    for (int i = 0; i < nIni; i++) {
        OctNode ni;
        ni.x0 = (int)(hX * static_cast<float>(i));
        ni.x1 = (int)(hX * static_cast<float>(i + 1));
        ni.y0 = 0;
        ni.y1 = maxY - minY;
        ni.keys.reserve(n);
        nodes.push_back(ni);
        ini[i] = &nodes.back();
    }
    for (int i = 0; i < n; i++) {
        int slot = (int)(c.px(i) / hX);
        if (slot >= nIni) slot = nIni - 1;  // cannot happen for in-range candidates
        ini[slot]->keys.push_back(i);
    }
    for (auto it = nodes.begin(); it != nodes.end();) {
        if (it->keys.size() == 1) { it->noMore = true; ++it; }
        else if (it->keys.empty()) it = nodes.erase(it);
        else ++it;
    }
    bool finish = false;
    std::vector<std::pair<int, OctNode *>> sizeAndNode;
    sizeAndNode.reserve(nodes.size() * 4);
    auto pushChildren = [&](OctNode ch[4], int *nToExpand) {
        for (int q = 0; q < 4; q++) {
            if (ch[q].keys.empty()) continue;
            nodes.push_front(ch[q]);
            if (ch[q].keys.size() > 1) {
                if (nToExpand) (*nToExpand)++;
                sizeAndNode.push_back(std::make_pair((int)ch[q].keys.size(), &nodes.front()));
                nodes.front().self = nodes.begin();
            }
        }
    };
    while (!finish) {
        int prevSize = (int)nodes.size();
        auto it = nodes.begin();
        int nToExpand = 0;
        sizeAndNode.clear();
        while (it != nodes.end()) {
            if (it->noMore) { ++it; continue; }
            OctNode ch[4];
            divideNode(c, *it, ch);
            pushChildren(ch, &nToExpand);
            it = nodes.erase(it);
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) {
            finish = true;
        } else if (((int)nodes.size() + nToExpand * 3) > N) {
            while (!finish) {
                prevSize = (int)nodes.size();
                std::vector<std::pair<int, OctNode *>> prev = sizeAndNode;
                sizeAndNode.clear();
                std::sort(prev.begin(), prev.end(), compareNodes);
                for (int j = (int)prev.size() - 1; j >= 0; j--) {
                    OctNode ch[4];
                    divideNode(c, *prev[j].second, ch);
                    pushChildren(ch, nullptr);
                    nodes.erase(prev[j].second->self);
                    if ((int)nodes.size() >= N) break;
                }
                if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) finish = true;
            }
        }
    }
    int cnt = 0;
    for (auto &nd : nodes) {
        int best = nd.keys[0];
        float maxResponse = c.resp(best);
        for (size_t k = 1; k < nd.keys.size(); k++)
            if (c.resp(nd.keys[k]) > maxResponse) {
                best = nd.keys[k];
                maxResponse = c.resp(best);
            }
        if (cnt < cap) out_idx[cnt] = best;
        cnt++;
    }
    return cnt;
}

// ------------------------------------------------------------------------------------------------
// Extractor - ORBextractor ctor :393-499, ComputePyramid :1495-1520, ComputeKeyPointsOctTree
// :1112-1227, operator() CPU branch :1356-1493
// ------------------------------------------------------------------------------------------------
extern "C" {

orc_extractor *orc_extractor_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th) {
    orc_extractor *ex = new orc_extractor();
    ex->nfeatures = nfeatures;
    ex->nlevels = nlevels;
    ex->iniTh = ini_th;
    ex->minTh = min_th;
    ex->scaleFactor = scale_factor;
    ex->sf.resize(nlevels);
    ex->invsf.resize(nlevels);
    orc_scale_factors(scale_factor, nlevels, ex->sf.data(), ex->invsf.data());
    ex->quota.resize(nlevels);
    orc_features_per_level(nfeatures, scale_factor, nlevels, ex->quota.data());
    orc_umax(ex->umax);
    return ex;
}

void orc_extractor_destroy(orc_extractor *ex) { delete ex; }

// The reference keeps each level inside a bordered buffer (EDGE_THRESHOLD = 19, REFLECT_101); no stage
// on this path reads the border (FAST tests x in [19, w-20], patches reach <= 18 px, SAD windows are
// bounds-checked), so levels are stored tight here.
int orc_compute_pyramid(orc_extractor *ex, const uint8_t *img, int w, int h, int stride) {
    if (!img || w <= 0 || h <= 0) return -1;
    ex->pyr.assign(ex->nlevels, Image());
    for (int level = 0; level < ex->nlevels; ++level) {
        float scale = ex->invsf[level];
        int lw = cvRoundF((float)w * scale), lh = cvRoundF((float)h * scale);
        Image &L = ex->pyr[level];
        L.w = lw;
        L.h = lh;
        L.px.resize((size_t)lw * lh);
        if (level == 0) {
            for (int y = 0; y < h; y++) memcpy(L.row(y), img + (size_t)y * stride, w);
        } else {
            const Image &P = ex->pyr[level - 1];
            orc_resize_linear_u8(P.px.data(), P.w, P.h, P.w, L.px.data(), lw, lh, lw);
        }
    }
    return 0;
}

static void computeKeyPointsOctTree(orc_extractor *ex) {
    const int nlevels = ex->nlevels;
    ex->cand.assign(nlevels, {});
    ex->levelKeys.assign(nlevels, {});
    const float W = 35;
    for (int level = 0; level < nlevels; ++level) {
        const Image &im = ex->pyr[level];
        const int minBorderX = kEdgeThreshold - 3;
        const int minBorderY = minBorderX;
        const int maxBorderX = im.w - kEdgeThreshold + 3;
        const int maxBorderY = im.h - kEdgeThreshold + 3;
        std::vector<int> &cand = ex->cand[level];
        const float width = (float)(maxBorderX - minBorderX);
        const float height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / W);
        const int nRows = (int)(height / W);
        if (nCols < 1 || nRows < 1) continue;  // reference would divide by zero; level too small
        const int wCell = (int)ceil(width / nCols);
        const int hCell = (int)ceil(height / nRows);
        std::vector<int> cell;
        for (int i = 0; i < nRows; i++) {
            const float iniY = (float)(minBorderY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = (float)maxBorderY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(minBorderX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = (float)maxBorderX;
                const int x0 = (int)iniX, y0 = (int)iniY, cw = (int)maxX - x0, chh = (int)maxY - y0;
                cell.resize((size_t)3 * cw * chh);
                int nc = orc_fast9_16(im.row(y0) + x0, cw, chh, im.w, ex->iniTh, 1, cell.data(), cw * chh);
                if (nc == 0) nc = orc_fast9_16(im.row(y0) + x0, cw, chh, im.w, ex->minTh, 1, cell.data(), cw * chh);
                for (int k = 0; k < nc; k++) {
                    cand.push_back(cell[3 * k] + j * wCell);
                    cand.push_back(cell[3 * k + 1] + i * hCell);
                    cand.push_back(cell[3 * k + 2]);
                }
            }
        }
        const int n = (int)cand.size() / 3;
        std::vector<int> keep(n > 0 ? n : 1);
        int nk = n ? orc_distribute_octree(cand.data(), n, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                           ex->quota[level], keep.data(), n)
                   : 0;
        const int scaledPatchSize = (int)(kPatchSize * ex->sf[level]);
        std::vector<orc_keypoint> &keys = ex->levelKeys[level];
        keys.resize(nk);
        for (int k = 0; k < nk; k++) {
            orc_keypoint &kp = keys[k];
            kp.x = (float)cand[3 * keep[k]];
            kp.y = (float)cand[3 * keep[k] + 1];
            kp.response = (float)cand[3 * keep[k] + 2];
            kp.x += minBorderX;
            kp.y += minBorderY;
            kp.octave = level;
            kp.size = (float)scaledPatchSize;
            kp.angle = -1;
            kp.class_id = -1;
        }
    }
    for (int level = 0; level < nlevels; ++level)
        for (auto &kp : ex->levelKeys[level])
            kp.angle = orc_ic_angle(ex->pyr[level].px.data(), ex->pyr[level].w, kp.x, kp.y);
}

int orc_extract(orc_extractor *ex, const uint8_t *img, int w, int h, int stride, int lap0, int lap1,
                orc_keypoint *kps, uint8_t *desc, int cap, int *n_mono) {
    if (orc_compute_pyramid(ex, img, w, h, stride) != 0) return -1;
    computeKeyPointsOctTree(ex);
    const int nlevels = ex->nlevels;
    int nkeypoints = 0;
    for (int l = 0; l < nlevels; l++) nkeypoints += (int)ex->levelKeys[l].size();
    ex->blurred.assign(nlevels, Image());
    ex->levelDesc.assign(nlevels, {});
    int monoIndex = 0, stereoIndex = nkeypoints - 1;
    for (int level = 0; level < nlevels; ++level) {
        std::vector<orc_keypoint> &keys = ex->levelKeys[level];
        const int nl = (int)keys.size();
        if (nl == 0) continue;
        const Image &im = ex->pyr[level];
        Image &bl = ex->blurred[level];
        bl.w = im.w;
        bl.h = im.h;
        bl.px.resize(im.px.size());
        orc_gaussian_blur7_u8(im.px.data(), im.w, im.h, im.w, bl.px.data(), im.w);
        std::vector<uint8_t> &d = ex->levelDesc[level];
        d.resize((size_t)nl * 32);
        for (int i = 0; i < nl; i++)
            orc_brief_descriptor(bl.px.data(), bl.w, keys[i].x, keys[i].y, keys[i].angle, &d[(size_t)i * 32]);
        const float scale = ex->sf[level];
        for (int i = 0; i < nl; i++) {
            orc_keypoint kp = keys[i];
            if (level != 0) { kp.x *= scale; kp.y *= scale; }
            int dst;
            if (kp.x >= lap0 && kp.x <= lap1) dst = stereoIndex--;
            else dst = monoIndex++;
            if (dst < cap) {
                if (kps) kps[dst] = kp;
                if (desc) memcpy(desc + (size_t)dst * 32, &d[(size_t)i * 32], 32);
            }
        }
    }
    if (n_mono) *n_mono = monoIndex;
    return nkeypoints;
}

static int getImage(const std::vector<Image> &v, int level, const uint8_t **data, int *w, int *h, int *stride) {
    if (level < 0 || level >= (int)v.size() || v[level].px.empty()) return -1;
    *data = v[level].px.data();
    *w = v[level].w;
    *h = v[level].h;
    *stride = v[level].w;
    return 0;
}
int orc_get_level(const orc_extractor *ex, int level, const uint8_t **data, int *w, int *h, int *stride) {
    return getImage(ex->pyr, level, data, w, h, stride);
}
int orc_get_blurred(const orc_extractor *ex, int level, const uint8_t **data, int *w, int *h, int *stride) {
    return getImage(ex->blurred, level, data, w, h, stride);
}
int orc_get_candidates(const orc_extractor *ex, int level, int *xys, int cap) {
    if (level < 0 || level >= (int)ex->cand.size()) return -1;
    int n = (int)ex->cand[level].size() / 3;
    if (xys) memcpy(xys, ex->cand[level].data(), sizeof(int) * 3 * std::min(n, cap));
    return n;
}
int orc_get_level_keypoints(const orc_extractor *ex, int level, orc_keypoint *out, uint8_t *desc, int cap) {
    if (level < 0 || level >= (int)ex->levelKeys.size()) return -1;
    int n = (int)ex->levelKeys[level].size();
    int m = std::min(n, cap);
    if (out) memcpy(out, ex->levelKeys[level].data(), sizeof(orc_keypoint) * m);
    if (desc && level < (int)ex->levelDesc.size() && !ex->levelDesc[level].empty())
        memcpy(desc, ex->levelDesc[level].data(), (size_t)32 * m);
    return n;
}

// ------------------------------------------------------------------------------------------------
// Frame::ComputeStereoMatches - src/Frame.cc:835-1005
// ------------------------------------------------------------------------------------------------
int orc_stereo_match(const orc_extractor *exL, const orc_extractor *exR, const orc_keypoint *keysL, int N,
                     const orc_keypoint *keysR, int Nr, const uint8_t *descL, const uint8_t *descR, float mbf,
                     float mb, float *uright, float *depth, int *best_dist_out, int *hamming_idx,
                     int apply_median_cut) {
    for (int i = 0; i < N; i++) {
        uright[i] = -1.0f;
        depth[i] = -1.0f;
        if (best_dist_out) best_dist_out[i] = -1;
        if (hamming_idx) hamming_idx[i] = -1;
    }
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = exL->pyr[0].h;
    const std::vector<float> &sf = exL->sf;
    const std::vector<float> &invsf = exL->invsf;
    std::vector<std::vector<size_t>> vRowIndices(nRows);
    for (int iR = 0; iR < Nr; iR++) {
        const float kpY = keysR[iR].y;
        const float r = 2.0f * sf[keysR[iR].octave];
        const int maxr = (int)ceil(kpY + r);
        const int minr = (int)floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);  // reference indexes unchecked
    }
    const float minZ = mb;
    const float minD = 0;
    const float maxD = mbf / minZ;
    std::vector<std::pair<int, int>> vDistIdx;
    vDistIdx.reserve(N);
    for (int iL = 0; iL < N; iL++) {
        const orc_keypoint &kpL = keysL[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y;
        const float uL = kpL.x;
        const int row = (int)vL;
        if (row < 0 || row >= nRows) continue;
        const std::vector<size_t> &vCandidates = vRowIndices[row];
        if (vCandidates.empty()) continue;
        const float minU = uL - maxD;
        const float maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH;
        size_t bestIdxR = 0;
        const uint8_t *dL = descL + (size_t)iL * 32;
        for (size_t iC = 0; iC < vCandidates.size(); iC++) {
            const size_t iR = vCandidates[iC];
            const orc_keypoint &kpR = keysR[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orc_descriptor_distance(dL, descR + iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {
            if (hamming_idx) hamming_idx[iL] = (int)bestIdxR;
            const float uR0 = keysR[bestIdxR].x;
            const float scaleFactor = invsf[kpL.octave];
            const float scaleduL = roundf(kpL.x * scaleFactor);
            const float scaledvL = roundf(kpL.y * scaleFactor);
            const float scaleduR0 = roundf(uR0 * scaleFactor);
            const int w = 5;
            const Image &imL = exL->pyr[kpL.octave];
            const Image &imR = exR->pyr[kpL.octave];
            const int yl0 = (int)(scaledvL - w), xl0 = (int)(scaleduL - w);
            int bestDistS = INT_MAX;
            int bestincR = 0;
            const int L = 5;
            float vDists[2 * 5 + 1];
            const float iniu = scaleduR0 + L - w;
            const float endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= imR.w) continue;
            // cv::Mat::rowRange/colRange would assert on a window leaving the level; keypoints are
            // >= 19 px inside so this cannot trigger, guarded for safety
            if (yl0 < 0 || yl0 + 2 * w + 1 > imL.h || xl0 < 0 || xl0 + 2 * w + 1 > imL.w) continue;
            if ((int)scaleduR0 - L - w < 0) continue;  // the reference only tests +L (:938-940); same remark
            for (int incR = -L; incR <= +L; incR++) {
                const int xr0 = (int)(scaleduR0 + incR - w);
                int sad = 0;
                for (int yy = 0; yy < 2 * w + 1; yy++) {
                    const uint8_t *a = imL.row(yl0 + yy) + xl0;
                    const uint8_t *b = imR.row(yl0 + yy) + xr0;
                    for (int xx = 0; xx < 2 * w + 1; xx++) sad += std::abs((int)a[xx] - (int)b[xx]);
                }
                float dist = (float)sad;  // cv::norm(NORM_L1) returns double, stored in a float
                if (dist < bestDistS) { bestDistS = (int)dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1];
            const float dist2 = vDists[L + bestincR];
            const float dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = sf[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) {
                    disparity = 0.01;
                    bestuR = uL - 0.01;
                }
                depth[iL] = mbf / disparity;
                uright[iL] = bestuR;
                if (best_dist_out) best_dist_out[iL] = bestDistS;
                vDistIdx.push_back(std::pair<int, int>(bestDistS, iL));
            }
        }
    }
    int nm = (int)vDistIdx.size();
    if (apply_median_cut && !vDistIdx.empty()) {  // empty case is UB in the reference (:992)
        std::sort(vDistIdx.begin(), vDistIdx.end());
        const float median = (float)vDistIdx[vDistIdx.size() / 2].first;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
            if (vDistIdx[i].first < thDist) break;
            uright[vDistIdx[i].second] = -1;
            depth[vDistIdx[i].second] = -1;
            nm--;
        }
    }
    return nm;
}

// ------------------------------------------------------------------------------------------------
// Frame::ComputeStereoFishEyeMatches, matching part - src/Frame.cc:1231-1255.
// BFMatcher(NORM_HAMMING).knnMatch(k=2): two smallest distances, earlier train index first on ties
// (SURVEY A.5); accept when best < second * 0.7 (double arithmetic, :1255).
// ------------------------------------------------------------------------------------------------
int orc_fisheye_match(const uint8_t *descL, int nL, const uint8_t *descR, int nR, int *matches, int *best,
                      int *second) {
    int n = 0;
    for (int i = 0; i < nL; i++) {
        int d0 = INT_MAX, d1 = INT_MAX, i0 = -1;
        for (int j = 0; j < nR; j++) {
            int d = orc_descriptor_distance(descL + (size_t)i * 32, descR + (size_t)j * 32);
            if (d < d1) {
                if (d < d0) { d1 = d0; d0 = d; i0 = j; }
                else d1 = d;
            }
        }
        matches[i] = -1;
        if (best) best[i] = nR >= 1 ? d0 : -1;
        if (second) second[i] = nR >= 2 ? d1 : -1;
        if (nR >= 2 && (float)d0 < (float)d1 * 0.7) { matches[i] = i0; n++; }
    }
    return n;
}

// ------------------------------------------------------------------------------------------------
// Frame grid - PosInGrid :749-759, AssignFeaturesToGrid :409-440, GetFeaturesInArea :681-747
// ------------------------------------------------------------------------------------------------
struct OrcGrid {
    std::vector<int> cell[GRID_COLS][GRID_ROWS];
    std::vector<int> cellR[GRID_COLS][GRID_ROWS];
};

static void buildGrid(const orc_frame *F, OrcGrid &g) {
    for (int i = 0; i < F->N; i++) {
        const orc_keypoint &kp = (F->Nleft == -1) ? F->keys[i] : (i < F->Nleft) ? F->keys[i] : F->keys_right[i - F->Nleft];
        int posX = (int)round((kp.x - F->mnMinX) * F->grid_inv_w);
        int posY = (int)round((kp.y - F->mnMinY) * F->grid_inv_h);
        if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) continue;
        if (F->Nleft == -1 || i < F->Nleft) g.cell[posX][posY].push_back(i);
        else g.cellR[posX][posY].push_back(i - F->Nleft);
    }
}

static std::vector<int> featuresInArea(const orc_frame *F, const OrcGrid &g, float x, float y, float r,
                                       int minLevel, int maxLevel, bool bRight) {
    std::vector<int> out;
    float factorX = r, factorY = r;
    const int nMinCellX = std::max(0, (int)floor((x - F->mnMinX - factorX) * F->grid_inv_w));
    if (nMinCellX >= GRID_COLS) return out;
    const int nMaxCellX = std::min((int)GRID_COLS - 1, (int)ceil((x - F->mnMinX + factorX) * F->grid_inv_w));
    if (nMaxCellX < 0) return out;
    const int nMinCellY = std::max(0, (int)floor((y - F->mnMinY - factorY) * F->grid_inv_h));
    if (nMinCellY >= GRID_ROWS) return out;
    const int nMaxCellY = std::min((int)GRID_ROWS - 1, (int)ceil((y - F->mnMinY + factorY) * F->grid_inv_h));
    if (nMaxCellY < 0) return out;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const std::vector<int> &vCell = (!bRight) ? g.cell[ix][iy] : g.cellR[ix][iy];
            for (size_t j = 0; j < vCell.size(); j++) {
                const orc_keypoint &kpUn = (F->Nleft == -1) ? F->keys[vCell[j]]
                                           : (!bRight)      ? F->keys[vCell[j]]
                                                            : F->keys_right[vCell[j]];
                if (bCheckLevels) {
                    if (kpUn.octave < minLevel) continue;
                    if (maxLevel >= 0)
                        if (kpUn.octave > maxLevel) continue;
                }
                const float distx = kpUn.x - x;
                const float disty = kpUn.y - y;
                if (fabs(distx) < factorX && fabs(disty) < factorY) out.push_back(vCell[j]);
            }
        }
    return out;
}

int orc_features_in_area(const orc_frame *F, float x, float y, float r, int min_level, int max_level,
                         int right, int *out, int cap) {
    OrcGrid *g = new OrcGrid();
    buildGrid(F, *g);
    std::vector<int> v = featuresInArea(F, *g, x, y, r, min_level, max_level, right != 0);
    delete g;
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = v[i];
    return (int)v.size();
}

static inline int octaveOf(const orc_frame *F, int idx) {
    return (F->Nleft == -1) ? F->keys[idx].octave : (idx < F->Nleft) ? F->keys[idx].octave : F->keys_right[idx - F->Nleft].octave;
}

// ------------------------------------------------------------------------------------------------
// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints)
// CPU branch - src/ORBmatcher.cc:49-225, RadiusByViewingCos :314-320.
// holder_obs[i] models F.mvpMapPoints[i]: -1 = NULL, else Observations() of the holder.
// ------------------------------------------------------------------------------------------------
int orc_search_local_points(orc_frame *F, const orc_local_points *P, float th, float nn_ratio, int *assign,
                            int *o_bd, int *o_bd2, int *o_bl, int *o_bl2, int *o_bi, int *o_bdr, int *o_bd2r,
                            int *o_blr, int *o_bl2r, int *o_bir) {
    OrcGrid *g = new OrcGrid();
    buildGrid(F, *g);
    for (int i = 0; i < F->N; i++) assign[i] = -1;
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    auto radiusByViewingCos = [](float viewCos) -> float { return viewCos > 0.998 ? 2.5f : 4.0f; };
    for (int iMP = 0; iMP < P->M; iMP++) {
        if (o_bd) { o_bd[iMP] = 256; o_bd2[iMP] = 256; o_bl[iMP] = -1; o_bl2[iMP] = -1; o_bi[iMP] = -1; }
        if (o_bdr) { o_bdr[iMP] = 256; o_bd2r[iMP] = 256; o_blr[iMP] = -1; o_bl2r[iMP] = -1; o_bir[iMP] = -1; }
        if (P->skip[iMP]) continue;
        const uint8_t *MPdescriptor = P->descriptors + (size_t)iMP * 32;
        if (P->in_view[iMP]) {
            const int nPredictedLevel = P->level[iMP];
            float r = radiusByViewingCos(P->view_cos[iMP]);
            if (bFactor) r *= th;
            const std::vector<int> vIndices = featuresInArea(F, *g, P->proj_x[iMP], P->proj_y[iMP],
                                                             r * F->scale_factors[nPredictedLevel],
                                                             nPredictedLevel - 1, nPredictedLevel, false);
            if (!vIndices.empty()) {
                int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
                for (int idx : vIndices) {
                    if (F->holder_obs[idx] > 0) continue;
                    if (F->Nleft == -1 && F->uright && F->uright[idx] > 0) {
                        const float er = fabs(P->proj_xr[iMP] - F->uright[idx]);
                        if (er > r * F->scale_factors[nPredictedLevel]) continue;
                    }
                    const int dist = orc_descriptor_distance(MPdescriptor, F->descriptors + (size_t)idx * 32);
                    if (dist < bestDist) {
                        bestDist2 = bestDist;
                        bestDist = dist;
                        bestLevel2 = bestLevel;
                        bestLevel = octaveOf(F, idx);
                        bestIdx = idx;
                    } else if (dist < bestDist2) {
                        bestLevel2 = octaveOf(F, idx);
                        bestDist2 = dist;
                    }
                }
                if (o_bd) { o_bd[iMP] = bestDist; o_bd2[iMP] = bestDist2; o_bl[iMP] = bestLevel; o_bl2[iMP] = bestLevel2; o_bi[iMP] = bestIdx; }
                if (bestDist <= TH_HIGH) {
                    if (bestLevel == bestLevel2 && bestDist > nn_ratio * bestDist2) continue;
                    if (bestLevel != bestLevel2 || bestDist <= nn_ratio * bestDist2) {
                        F->holder_obs[bestIdx] = P->observations[iMP];
                        assign[bestIdx] = iMP;
                        if (F->Nleft != -1 && F->left_to_right[bestIdx] != -1) {
                            int j = F->left_to_right[bestIdx] + F->Nleft;
                            F->holder_obs[j] = P->observations[iMP];
                            assign[j] = iMP;
                            nmatches++;
                        }
                        nmatches++;
                    }
                }
            }
        }
        if (F->Nleft != -1 && P->in_view_r[iMP]) {
            const int nPredictedLevel = P->level_r[iMP];
            if (nPredictedLevel != -1) {
                float r = radiusByViewingCos(P->view_cos_r[iMP]);
                const std::vector<int> vIndices = featuresInArea(F, *g, P->proj_xr[iMP], P->proj_yr[iMP],
                                                                 r * F->scale_factors[nPredictedLevel],
                                                                 nPredictedLevel - 1, nPredictedLevel, true);
                if (vIndices.empty()) continue;
                int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
                for (int idx : vIndices) {
                    if (F->holder_obs[idx + F->Nleft] > 0) continue;
                    const int dist = orc_descriptor_distance(MPdescriptor, F->descriptors + (size_t)(idx + F->Nleft) * 32);
                    if (dist < bestDist) {
                        bestDist2 = bestDist;
                        bestDist = dist;
                        bestLevel2 = bestLevel;
                        bestLevel = F->keys_right[idx].octave;
                        bestIdx = idx;
                    } else if (dist < bestDist2) {
                        bestLevel2 = F->keys_right[idx].octave;
                        bestDist2 = dist;
                    }
                }
                if (o_bdr) { o_bdr[iMP] = bestDist; o_bd2r[iMP] = bestDist2; o_blr[iMP] = bestLevel; o_bl2r[iMP] = bestLevel2; o_bir[iMP] = bestIdx; }
                if (bestDist <= TH_HIGH) {
                    if (bestLevel == bestLevel2 && bestDist > nn_ratio * bestDist2) continue;
                    if (F->Nleft != -1 && F->right_to_left[bestIdx] != -1) {
                        int j = F->right_to_left[bestIdx];
                        F->holder_obs[j] = P->observations[iMP];
                        assign[j] = iMP;
                        nmatches++;
                    }
                    F->holder_obs[bestIdx + F->Nleft] = P->observations[iMP];
                    assign[bestIdx + F->Nleft] = iMP;
                    nmatches++;
                }
            }
        }
    }
    delete g;
    return nmatches;
}

// ORBmatcher::ComputeThreeMaxima - src/ORBmatcher.cc:2210-2251
void orc_three_maxima(const int *histo, int L, int *ind1, int *ind2, int *ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            *ind3 = *ind2; *ind2 = *ind1; *ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            *ind3 = *ind2; *ind2 = i;
        } else if (s > max3) {
            max3 = s; *ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

// camera projection - src/CameraModels/Pinhole.cpp:43-49, KannalaBrandt8.cpp:67-84.  `cos(psi)` / `sin(psi)` there are unqualified
// calls with a float argument in a file WITHOUT a using-directive (none is reachable from KannalaBrandt8.h either): they are cosf /
// sinf only if the translation unit has seen libstdc++'s <math.h> wrapper, which pulls the std:: float overloads into the global
// namespace; with <cmath> alone they are ::cos(double) - a double product narrowed at the store (checked with this image's g++ 11).
// The reference's own headers do not include <math.h> on that path for GCC (Thirdparty/g2o/g2o/stuff/macros.h:104 is its
// "unknown compiler" branch); KannalaBrandt8.h -> TwoViewReconstruction.h:22 includes <opencv2/opencv.hpp>, and OpenCV's flann
// headers (opencv2/flann/lsh_table.h) do include <math.h> - OpenCV is absent here, so this binding is part of "unpinned: OpenCV".
// cosf / sinf it is, as in rounds 1-3.
static void projectCam(const orc_frame *F, const float p[3], float uv[2]) {
    if (F->cam_model == 0) {
        uv[0] = F->cam[0] * p[0] / p[2] + F->cam[2];
        uv[1] = F->cam[1] * p[1] / p[2] + F->cam[3];
    } else {
        const float x2_plus_y2 = p[0] * p[0] + p[1] * p[1];
        const float theta = atan2f(sqrtf(x2_plus_y2), p[2]);
        const float psi = atan2f(p[1], p[0]);
        const float theta2 = theta * theta;
        const float theta3 = theta * theta2;
        const float theta5 = theta3 * theta2;
        const float theta7 = theta5 * theta2;
        const float theta9 = theta7 * theta2;
        const float r = theta + F->cam[4] * theta3 + F->cam[5] * theta5 + F->cam[6] * theta7 + F->cam[7] * theta9;
        uv[0] = F->cam[0] * r * cosf(psi) + F->cam[2];
        uv[1] = F->cam[1] * r * sinf(psi) + F->cam[3];
    }
}

// rigid transform y = R x + t with T row-major 3x4; evaluation order fixed as ((r0*x + r1*y) + r2*z) + t
static void transform34(const float *T, const float x[3], float y[3]) {
    for (int r = 0; r < 3; r++) y[r] = ((T[4 * r] * x[0] + T[4 * r + 1] * x[1]) + T[4 * r + 2] * x[2]) + T[4 * r + 3];
}

// ------------------------------------------------------------------------------------------------
// ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) CPU branch -
// src/ORBmatcher.cc:1775-1990.  bForward/bBackward (:1794-1795) are inputs, as in the reference's
// own launchPoseEstimationKernel boundary (include/Kernels/KernelController.h:44-46).
// ------------------------------------------------------------------------------------------------
// Sophus::SE3f * point as the CPU branch evaluates `Tcw * x3Dw` (src/ORBmatcher.cc:1805) and `GetRelativePoseTrl() * x3Dc`
// (:1900): Sophus::SE3Base::operator*(point) = so3() * p + translation() (Thirdparty/Sophus/sophus/se3.hpp:321-324) with
// SO3Base::operator*(point) (so3.hpp:358-367): uv = q.vec().cross(p); uv += uv; return p + q.w() * uv + q.vec().cross(uv).
// Eigen's cross() (Eigen/src/Geometry/OrthoMethods.h) returns the evaluated vector (a1 b2 - a2 b1, a2 b0 - a0 b2,
// a0 b1 - a1 b0); the sum is coefficient-wise (p + w uv) + cross.  q = (x, y, z, w) = Eigen::Quaternionf::coeffs().
static void cross3(const float a[3], const float b[3], float c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
void orc_se3_transform(const float q[4], const float t[3], const float p[3], float y[3]) {
    float uv[3], c[3];
    cross3(q, p, uv);
    for (int i = 0; i < 3; i++) uv[i] = uv[i] + uv[i];
    cross3(q, uv, c);
    for (int i = 0; i < 3; i++) {
        const float r = (p[i] + q[3] * uv[i]) + c[i];
        y[i] = r + t[i];
    }
}

extern "C++" {
template <class TcwFn, class TrlFn>
static int searchLastFrameImpl(orc_frame *Cur, const orc_last_points *Lp, TcwFn applyTcw, TrlFn applyTrl, float th, int bForward,
                               int bBackward, int check_orientation, int *assign, int *o_bd, int *o_bi, int *o_bdr,
                               int *o_bir) {
    OrcGrid *g = new OrcGrid();
    buildGrid(Cur, *g);
    for (int i = 0; i < Cur->N; i++) assign[i] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    auto curAngle = [&](int idx) -> float {
        return (Cur->Nleft == -1) ? Cur->keys[idx].angle : (idx < Cur->Nleft) ? Cur->keys[idx].angle : Cur->keys_right[idx - Cur->Nleft].angle;
    };
    for (int i = 0; i < Lp->N; i++) {
        if (o_bd) { o_bd[i] = 256; o_bi[i] = -1; }
        if (o_bdr) { o_bdr[i] = 256; o_bir[i] = -1; }
        if (!Lp->valid[i]) continue;
        float x3Dc[3];
        applyTcw(Lp->world_pos + 3 * i, x3Dc);
        const float invzc = 1.0 / x3Dc[2];
        if (invzc < 0) continue;
        float uv[2];
        projectCam(Cur, x3Dc, uv);
        if (uv[0] < Cur->mnMinX || uv[0] > Cur->mnMaxX) continue;
        if (uv[1] < Cur->mnMinY || uv[1] > Cur->mnMaxY) continue;
        const int nLastOctave = Lp->octave[i];
        float radius = th * Cur->scale_factors[nLastOctave];
        std::vector<int> vIndices2;
        if (bForward) vIndices2 = featuresInArea(Cur, *g, uv[0], uv[1], radius, nLastOctave, -1, false);
        else if (bBackward) vIndices2 = featuresInArea(Cur, *g, uv[0], uv[1], radius, 0, nLastOctave, false);
        else vIndices2 = featuresInArea(Cur, *g, uv[0], uv[1], radius, nLastOctave - 1, nLastOctave + 1, false);
        if (vIndices2.empty()) continue;
        const uint8_t *dMP = Lp->descriptors + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            if (Cur->holder_obs[i2] > 0) continue;
            if (Cur->Nleft == -1 && Cur->uright && Cur->uright[i2] > 0) {
                const float ur = uv[0] - Cur->mbf * invzc;
                const float er = fabs(ur - Cur->uright[i2]);
                if (er > radius) continue;
            }
            const int dist = orc_descriptor_distance(dMP, Cur->descriptors + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (o_bd) { o_bd[i] = bestDist; o_bi[i] = bestIdx2; }
        if (bestDist <= TH_HIGH) {
            Cur->holder_obs[bestIdx2] = Lp->observations[i];
            assign[bestIdx2] = i;
            nmatches++;
            if (check_orientation) {
                float rot = Lp->angle[i] - curAngle(bestIdx2);
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
        if (Cur->Nleft != -1) {
            float x3Dr[3];
            applyTrl(x3Dc, x3Dr);
            float uvr[2];
            projectCam(Cur, x3Dr, uvr);
            float radiusR = th * Cur->scale_factors[nLastOctave];
            std::vector<int> vR;
            if (bForward) vR = featuresInArea(Cur, *g, uvr[0], uvr[1], radiusR, nLastOctave, -1, true);
            else if (bBackward) vR = featuresInArea(Cur, *g, uvr[0], uvr[1], radiusR, 0, nLastOctave, true);
            else vR = featuresInArea(Cur, *g, uvr[0], uvr[1], radiusR, nLastOctave - 1, nLastOctave + 1, true);
            int bestDistR = 256, bestIdxR = -1;
            for (int i2 : vR) {
                if (Cur->holder_obs[i2 + Cur->Nleft] > 0) continue;
                const int dist = orc_descriptor_distance(dMP, Cur->descriptors + (size_t)(i2 + Cur->Nleft) * 32);
                if (dist < bestDistR) { bestDistR = dist; bestIdxR = i2; }
            }
            if (o_bdr) { o_bdr[i] = bestDistR; o_bir[i] = bestIdxR; }
            if (bestDistR <= TH_HIGH) {
                Cur->holder_obs[bestIdxR + Cur->Nleft] = Lp->observations[i];
                assign[bestIdxR + Cur->Nleft] = i;
                nmatches++;
                if (check_orientation) {
                    float rot = Lp->angle[i] - Cur->keys_right[bestIdxR].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(bestIdxR + Cur->Nleft);
                }
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        orc_three_maxima(sizes, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) {
                    assign[rotHist[i][j]] = -1;
                    Cur->holder_obs[rotHist[i][j]] = -1;
                    nmatches--;
                }
    }
    delete g;
    return nmatches;
}
}  // extern "C++"

// the pose as a row-major 3x4 matrix, y = R x + t (the form the reference's GPU boundary takes: Eigen::Matrix4f
// transform_matrix, include/Kernels/KernelController.h:44-46)
int orc_search_last_frame(orc_frame *Cur, const orc_last_points *Lp, const float *Tcw, float th, int bForward,
                          int bBackward, int check_orientation, int *assign, int *o_bd, int *o_bi, int *o_bdr,
                          int *o_bir) {
    return searchLastFrameImpl(
        Cur, Lp, [&](const float *x, float *y) { transform34(Tcw, x, y); }, [&](const float *x, float *y) { transform34(Cur->Trl, x, y); },
        th, bForward, bBackward, check_orientation, assign, o_bd, o_bi, o_bdr, o_bir);
}

// the poses as Sophus::SE3f holds and applies them - what the CPU branch computes (q = x y z w; trl may be NULL for one camera)
int orc_search_last_frame_se3(orc_frame *Cur, const orc_last_points *Lp, const float *q_tcw, const float *t_tcw, const float *q_trl,
                              const float *t_trl, float th, int bForward, int bBackward, int check_orientation, int *assign,
                              int *o_bd, int *o_bi, int *o_bdr, int *o_bir) {
    return searchLastFrameImpl(
        Cur, Lp, [&](const float *x, float *y) { orc_se3_transform(q_tcw, t_tcw, x, y); },
        [&](const float *x, float *y) { orc_se3_transform(q_trl, t_trl, x, y); }, th, bForward, bBackward, check_orientation, assign,
        o_bd, o_bi, o_bdr, o_bir);
}

// ------------------------------------------------------------------------------------------------
// Frame::isInFrustum (src/Frame.cc:536-610), Frame::isInFrustumChecks (:1308-1382),
// MapPoint::PredictScale (src/MapPoint.cc:531-546).  Eigen evaluates the sums of its fixed-size float expressions -
// dot(), squaredNorm() and every coefficient of a small matrix product, which is `(lhs.row(i).transpose().cwiseProduct(
// rhs.col(j))).sum()` (Eigen/src/Core/ProductEvaluators.h) - through redux_novec_unroller (Eigen/src/Core/Redux.h), which
// splits a range of Length terms at Length / 2: three terms associate as e0 + (e1 + e2), NOT left to right.  (Eigen is a
// third party absent from this image: restated from its published source, 3.3 / 3.4; tests/tools/eigen_order_probe.cpp
// prints the association a real Eigen uses.)  No contraction.  `log(ratio)` binds to logf (float argument; a using-directive reaches MapPoint.cc through
// MapPoint.h -> Frame.h:31 -> ORBVocabulary.h:24 -> Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:36), `ceil` to ceilf.
// ------------------------------------------------------------------------------------------------
static inline float sum3(float e0, float e1, float e2) { return e0 + (e1 + e2); }
static inline float dot3(const float *a, const float *b) { return sum3(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
static inline float norm3(const float *a) { return sqrtf(dot3(a, a)); }

static int predictScale(float maxDistanceRaw, float currentDist, float logScaleFactor, int nLevels) {
    const float ratio = maxDistanceRaw / currentDist;
    int nScale = (int)ceilf(logf(ratio) / logScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= nLevels) nScale = nLevels - 1;
    return nScale;
}

int orc_is_in_frustum(const orc_frame *F, const orc_frame_pose *T, const orc_map_points *P, float viewingCosLimit,
                      float logScaleFactor, uint8_t *in_view, uint8_t *in_view_r, int *level, int *level_r,
                      float *view_cos, float *view_cos_r, float *proj_x, float *proj_y, float *proj_xr,
                      float *proj_yr, float *depth, float *depth_r) {
    // right camera (Frame.cc:1314-1320): mR = Rrl * mRcw, mt = Rrl * mtcw + trl, twc = mRwc * tlr + mOw
    float Rr[9], tr[3], twcR[3];
    {
        const float *Trl = F->Trl;
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++)
                Rr[3 * i + j] = sum3(Trl[4 * i] * T->Rcw[j], Trl[4 * i + 1] * T->Rcw[3 + j], Trl[4 * i + 2] * T->Rcw[6 + j]);
            tr[i] = sum3(Trl[4 * i] * T->tcw[0], Trl[4 * i + 1] * T->tcw[1], Trl[4 * i + 2] * T->tcw[2]) + Trl[4 * i + 3];
            // mRwc = mRcw^T
            twcR[i] = sum3(T->Rcw[i] * T->tlr[0], T->Rcw[3 + i] * T->tlr[1], T->Rcw[6 + i] * T->tlr[2]) + T->Ow[i];
        }
    }
    int nToMatch = 0;
    for (int i = 0; i < P->M; i++) {
        in_view[i] = 0; in_view_r[i] = 0;
        level[i] = -1; level_r[i] = -1;
        view_cos[i] = 0; view_cos_r[i] = 0;
        proj_x[i] = -1; proj_y[i] = -1; proj_xr[i] = -1; proj_yr[i] = -1;
        depth[i] = 0; depth_r[i] = 0;
        if (P->skip && P->skip[i]) continue;
        const float *Pw = P->world_pos + 3 * i, *Pn = P->normal + 3 * i;
        const float maxDistance = 1.2f * P->max_distance[i], minDistance = 0.8f * P->min_distance[i];
        if (F->Nleft == -1) {  // Frame.cc:538-597
            float Pc[3];
            for (int r = 0; r < 3; r++) Pc[r] = dot3(T->Rcw + 3 * r, Pw) + T->tcw[r];
            const float Pc_dist = norm3(Pc);
            const float PcZ = Pc[2];
            const float invz = 1.0f / PcZ;
            if (PcZ < 0.0f) continue;
            float uv[2];
            projectCam(F, Pc, uv);
            if (uv[0] < F->mnMinX || uv[0] > F->mnMaxX) continue;
            if (uv[1] < F->mnMinY || uv[1] > F->mnMaxY) continue;
            proj_x[i] = uv[0];
            proj_y[i] = uv[1];
            const float PO[3] = {Pw[0] - T->Ow[0], Pw[1] - T->Ow[1], Pw[2] - T->Ow[2]};
            const float dist = norm3(PO);
            if (dist < minDistance || dist > maxDistance) continue;
            const float viewCos = dot3(PO, Pn) / dist;
            if (viewCos < viewingCosLimit) continue;
            const int nPredictedLevel = predictScale(P->max_distance[i], dist, logScaleFactor, F->nlevels);
            in_view[i] = 1;
            proj_xr[i] = uv[0] - F->mbf * invz;
            depth[i] = Pc_dist;
            level[i] = nPredictedLevel;
            view_cos[i] = viewCos;
            nToMatch++;
        } else {  // Frame.cc:599-609 -> isInFrustumChecks for both cameras
            for (int right = 0; right < 2; right++) {
                const float *R = right ? Rr : T->Rcw, *t = right ? tr : T->tcw, *twc = right ? twcR : T->Ow;
                float Pc[3];
                for (int r = 0; r < 3; r++) Pc[r] = dot3(R + 3 * r, Pw) + t[r];
                const float Pc_dist = norm3(Pc);
                if (Pc[2] < 0.0f) continue;
                float uv[2];
                projectCam(F, Pc, uv);  // mpCamera2 has the parameters of mpCamera in this POD view
                if (uv[0] < F->mnMinX || uv[0] > F->mnMaxX) continue;
                if (uv[1] < F->mnMinY || uv[1] > F->mnMaxY) continue;
                const float PO[3] = {Pw[0] - twc[0], Pw[1] - twc[1], Pw[2] - twc[2]};
                const float dist = norm3(PO);
                if (dist < minDistance || dist > maxDistance) continue;
                const float viewCos = dot3(PO, Pn) / dist;
                if (viewCos < viewingCosLimit) continue;
                const int nPredictedLevel = predictScale(P->max_distance[i], dist, logScaleFactor, F->nlevels);
                if (right) {
                    in_view_r[i] = 1;
                    proj_xr[i] = uv[0]; proj_yr[i] = uv[1];
                    level_r[i] = nPredictedLevel; view_cos_r[i] = viewCos; depth_r[i] = Pc_dist;
                } else {
                    in_view[i] = 1;
                    proj_x[i] = uv[0]; proj_y[i] = uv[1];
                    level[i] = nPredictedLevel; view_cos[i] = viewCos; depth[i] = Pc_dist;
                }
            }
            if (in_view[i] || in_view_r[i]) nToMatch++;
        }
    }
    return nToMatch;
}

// ------------------------------------------------------------------------------------------------
// KannalaBrandt8::unproject (:114-143), ::project (:67-84), ::Triangulate (:397-409), ::TriangulateMatches
// (:306-372), Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1231-1271)
// ------------------------------------------------------------------------------------------------
static void kb8Unproject(const float *cam, float precision, float px, float py, float r[3]) {
    const float pwx = (px - cam[2]) / cam[0], pwy = (py - cam[3]) / cam[1];
    float scale = 1.f;
    float theta_d = sqrtf(pwx * pwx + pwy * pwy);
    theta_d = fminf(fmaxf((float)(-3.1415926535897932384626433832795 / 2.f), theta_d), (float)(3.1415926535897932384626433832795 / 2.f));
    if (theta_d > 1e-8) {
        float theta = theta_d;
        for (int j = 0; j < 10; j++) {
            const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
            const float k0_theta2 = cam[4] * theta2, k1_theta4 = cam[5] * theta4;
            const float k2_theta6 = cam[6] * theta6, k3_theta8 = cam[7] * theta8;
            const float theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                                    (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
            theta = theta - theta_fix;
            if (fabsf(theta_fix) < precision) break;
        }
        scale = tanf(theta) / theta_d;
    }
    r[0] = pwx * scale;
    r[1] = pwy * scale;
    r[2] = 1.f;
}

static void kb8Project(const float *cam, const float p[3], float uv[2]) {
    orc_frame f;
    memset(&f, 0, sizeof f);
    f.cam_model = 1;
    memcpy(f.cam, cam, sizeof f.cam);
    projectCam(&f, p, uv);
}

// right singular vector of the smallest singular value of the 4x4 matrix A (row-major), one-sided Jacobi in double
static void nullVector4(const double A[16], double v[4]) {
    double U[16], V[16];
    for (int i = 0; i < 16; i++) { U[i] = A[i]; V[i] = (i % 5 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 30; sweep++) {
        bool rotated = false;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 4; i++) {
                    alpha += U[4 * i + p] * U[4 * i + p];
                    beta += U[4 * i + q] * U[4 * i + q];
                    gamma += U[4 * i + p] * U[4 * i + q];
                }
                if (fabs(gamma) <= 1e-15 * sqrt(alpha * beta) || gamma == 0.0) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < 4; i++) {
                    const double up = U[4 * i + p], uq = U[4 * i + q];
                    U[4 * i + p] = c * up - sn * uq;
                    U[4 * i + q] = sn * up + c * uq;
                    const double vp = V[4 * i + p], vq = V[4 * i + q];
                    V[4 * i + p] = c * vp - sn * vq;
                    V[4 * i + q] = sn * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    int best = 0;
    double bn = 0;
    for (int j = 0; j < 4; j++) {
        double nj = 0;
        for (int i = 0; i < 4; i++) nj += U[4 * i + j] * U[4 * i + j];
        if (j == 0 || nj < bn) { bn = nj; best = j; }
    }
    for (int i = 0; i < 4; i++) v[i] = V[4 * i + best];
}

static float kb8TriangulateMatches(const orc_fisheye_rig *rig, float x1, float y1, float x2, float y2, float sigmaLevel,
                                   float unc, float p3D[3]) {
    float r1[3], r2[3];
    kb8Unproject(rig->cam1, rig->precision, x1, y1, r1);
    kb8Unproject(rig->cam2, rig->precision, x2, y2, r2);
    const float *R12 = rig->Rlr, *t12 = rig->tlr;
    float r21[3];
    for (int i = 0; i < 3; i++) r21[i] = dot3(R12 + 3 * i, r2);
    const float cosParallaxRays = dot3(r1, r21) / (norm3(r1) * norm3(r21));
    if (cosParallaxRays > 0.9998) return -1;
    // Tcw1 = [I | 0], Tcw2 = [R21 | -R21 t12], R21 = R12^T
    float R21[9], t2[3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R21[3 * i + j] = R12[3 * j + i];
    for (int i = 0; i < 3; i++) t2[i] = -dot3(R21 + 3 * i, t12);  // (-R21) * t12, negation is exact
    float Tcw1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, Tcw2[12];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Tcw2[4 * i + j] = R21[3 * i + j];
        Tcw2[4 * i + 3] = t2[i];
    }
    // Triangulate: rows p.x*T.row(2) - T.row(0), p.y*T.row(2) - T.row(1) in float (:400-403)
    float Af[16];
    for (int j = 0; j < 4; j++) {
        Af[j] = r1[0] * Tcw1[8 + j] - Tcw1[j];
        Af[4 + j] = r1[1] * Tcw1[8 + j] - Tcw1[4 + j];
        Af[8 + j] = r2[0] * Tcw2[8 + j] - Tcw2[j];
        Af[12 + j] = r2[1] * Tcw2[8 + j] - Tcw2[4 + j];
    }
    double A[16], v[4];
    for (int i = 0; i < 16; i++) A[i] = Af[i];
    nullVector4(A, v);
    const float x3Dh[4] = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    const float x3D[3] = {x3Dh[0] / x3Dh[3], x3Dh[1] / x3Dh[3], x3Dh[2] / x3Dh[3]};
    const float z1 = x3D[2];
    if (z1 <= 0) return -2;
    const float z2 = dot3(R21 + 6, x3D) + Tcw2[11];
    if (z2 <= 0) return -3;
    float uv1[2];
    kb8Project(rig->cam1, x3D, uv1);
    const float errX1 = uv1[0] - x1, errY1 = uv1[1] - y1;
    if ((errX1 * errX1 + errY1 * errY1) > 5.991 * sigmaLevel) return -4;
    float x3D2[3];
    for (int i = 0; i < 3; i++) x3D2[i] = dot3(R21 + 3 * i, x3D) + t2[i];
    float uv2[2];
    kb8Project(rig->cam2, x3D2, uv2);
    const float errX2 = uv2[0] - x2, errY2 = uv2[1] - y2;
    if ((errX2 * errX2 + errY2 * errY2) > 5.991 * unc) return -5;
    p3D[0] = x3D[0]; p3D[1] = x3D[1]; p3D[2] = x3D[2];
    return z1;
}

void orc_kb8_triangulate(const orc_fisheye_rig *rig, int n, const float *xy1, const float *xy2, const float *sigma1,
                         const float *sigma2, float *code, float *p3d) {
    for (int i = 0; i < n; i++) {
        float p[3] = {0, 0, 0};
        code[i] = kb8TriangulateMatches(rig, xy1[2 * i], xy1[2 * i + 1], xy2[2 * i], xy2[2 * i + 1], sigma1[i], sigma2[i], p);
        p3d[3 * i] = p[0]; p3d[3 * i + 1] = p[1]; p3d[3 * i + 2] = p[2];
    }
}

int orc_fisheye_stereo(const orc_fisheye_rig *rig, const uint8_t *descL, const orc_keypoint *keysL, int nL,
                       const uint8_t *descR, const orc_keypoint *keysR, int nR, const float *level_sigma2, int *matches,
                       float *depth, float *p3d) {
    std::vector<int> knn(nL > 0 ? nL : 1);
    orc_fisheye_match(descL, nL, descR, nR, knn.data(), nullptr, nullptr);
    int nMatches = 0;
    for (int i = 0; i < nL; i++) {
        matches[i] = -1;
        depth[i] = -1.0f;
        p3d[3 * i] = p3d[3 * i + 1] = p3d[3 * i + 2] = 0.f;
        const int j = knn[i];
        if (j < 0) continue;
        float p[3];
        const float d = kb8TriangulateMatches(rig, keysL[i].x, keysL[i].y, keysR[j].x, keysR[j].y,
                                              level_sigma2[keysL[i].octave], level_sigma2[keysR[j].octave], p);
        if (d > 0.0001f) {
            matches[i] = j;
            depth[i] = d;
            p3d[3 * i] = p[0]; p3d[3 * i + 1] = p[1]; p3d[3 * i + 2] = p[2];
            nMatches++;
        }
    }
    return nMatches;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Frame::ComputeBoW (src/Frame.cc:762-769) = mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4):
// DBoW2's TemplatedVocabulary restated with the same containers (std::map, std::vector) and loops.
// ------------------------------------------------------------------------------------------------
#include <fstream>
#include <map>
#include <sstream>

struct orc_vocabulary {
    struct Node {  // TemplatedVocabulary.h:297-330
        double weight = 0;
        std::vector<unsigned> children;
        unsigned parent = 0;
        uint8_t descriptor[32] = {0};
        unsigned word_id = 0;
        bool isLeaf() const { return children.empty(); }
    };
    int k = 0, L = 0, scoring = 0, weighting = 0;
    std::vector<Node> nodes;
    int nWords = 0;
};

namespace {

// FORB::distance (FORB.cpp:81-101): the parallel bit count over eight 32-bit words
int forb_distance(const uint8_t *a, const uint8_t *b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        unsigned int v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

// TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup) (:1208-1253)
void voc_transform_one(const orc_vocabulary *voc, const uint8_t *feature, unsigned &word_id, double &weight, unsigned &nid,
                       int levelsup) {
    const int nid_level = voc->L - levelsup;
    nid = 0;  // root when nid_level <= 0 (:1218); the reference leaves it unset if the walk ends above nid_level
    unsigned final_id = 0;
    int current_level = 0;
    do {
        ++current_level;
        const std::vector<unsigned> &nodes = voc->nodes[final_id].children;
        final_id = nodes[0];
        double best_d = forb_distance(feature, voc->nodes[final_id].descriptor);
        for (size_t j = 1; j < nodes.size(); j++) {
            const unsigned id = nodes[j];
            const double d = forb_distance(feature, voc->nodes[id].descriptor);
            if (d < best_d) {
                best_d = d;
                final_id = id;
            }
        }
        if (current_level == nid_level) nid = final_id;
    } while (!voc->nodes[final_id].isLeaf());
    if (nid_level > current_level) nid = final_id;  // defined here as the leaf (fasttrack_amd.h: ft_bow_transform)
    word_id = voc->nodes[final_id].word_id;
    weight = voc->nodes[final_id].weight;
}

}  // namespace

extern "C" {

orc_vocabulary *orc_vocabulary_create(int k, int L, int scoring, int weighting, int n_nodes, const int *parent,
                                      const uint8_t *is_leaf, const uint8_t *descriptors, const double *weights) {
    orc_vocabulary *v = new orc_vocabulary;
    v->k = k;
    v->L = L;
    v->scoring = scoring;
    v->weighting = weighting;
    v->nodes.resize(n_nodes);
    for (int nid = 1; nid < n_nodes; nid++) {  // the body of loadFromTextFile's loop (:1384-1419)
        v->nodes[nid].parent = (unsigned)parent[nid];
        v->nodes[parent[nid]].children.push_back((unsigned)nid);
        memcpy(v->nodes[nid].descriptor, descriptors + (size_t)32 * nid, 32);
        v->nodes[nid].weight = weights[nid];
        if (is_leaf[nid]) v->nodes[nid].word_id = (unsigned)v->nWords++;
    }
    return v;
}

orc_vocabulary *orc_vocabulary_load_text(const char *path) {
    std::ifstream f(path);
    if (!f.good()) return nullptr;
    std::string s;
    std::getline(f, s);
    std::stringstream ss;
    ss << s;
    int k = -1, L = -1, n1 = -1, n2 = -1;
    ss >> k;
    ss >> L;
    ss >> n1;
    ss >> n2;
    if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) return nullptr;  // :1359
    std::vector<int> parent(1, 0);
    std::vector<uint8_t> leaf(1, 0), desc(32, 0);
    std::vector<double> weight(1, 0.0);
    while (!f.eof()) {
        std::string snode;
        std::getline(f, snode);
        if (snode.find_first_not_of(" \t\r\n") == std::string::npos) continue;  // (a trailing empty line: see bow.cpp)
        std::stringstream ssnode;
        ssnode << snode;
        int pid = 0, nIsLeaf = 0;
        ssnode >> pid;
        ssnode >> nIsLeaf;
        const size_t at = desc.size();
        desc.resize(at + 32, 0);
        for (int iD = 0; iD < 32; iD++) {  // FORB::fromString (FORB.cpp:120-135)
            int n = 0;
            ssnode >> n;
            if (!ssnode.fail()) desc[at + iD] = (unsigned char)n;
        }
        double w = 0.0;
        ssnode >> w;
        parent.push_back(pid);
        leaf.push_back(nIsLeaf > 0 ? 1 : 0);
        weight.push_back(w);
    }
    return orc_vocabulary_create(k, L, n1, n2, (int)parent.size(), parent.data(), leaf.data(), desc.data(), weight.data());
}

void orc_vocabulary_destroy(orc_vocabulary *v) { delete v; }
int orc_vocabulary_nodes(const orc_vocabulary *v) { return (int)v->nodes.size(); }
int orc_vocabulary_words(const orc_vocabulary *v) { return v->nWords; }

void orc_bow_transform(const orc_vocabulary *voc, const uint8_t *descriptors, int n, int levelsup, unsigned *word_ids,
                       unsigned *node_ids, double *weights, unsigned *bow_ids, double *bow_values, int *n_bow,
                       unsigned *fv_nodes, int *fv_offsets, unsigned *fv_features, int *n_fv) {
    std::map<unsigned, double> v;                 // BowVector
    std::map<unsigned, std::vector<unsigned>> fv;  // FeatureVector
    if (voc->nodes.size() > 1) {                  // !empty() (:1134)
        const bool must = voc->scoring != 5;      // mustNormalize: all but DotProductScoring (ScoringObject.h:73-89)
        const bool l2 = voc->scoring == 1;
        const bool tf = voc->weighting == 0 || voc->weighting == 1;  // TF_IDF || TF (:1145)
        for (int i_feature = 0; i_feature < n; i_feature++) {
            unsigned id, nid;
            double w;
            voc_transform_one(voc, descriptors + (size_t)32 * i_feature, id, w, nid, levelsup);
            if (word_ids) word_ids[i_feature] = id;
            if (node_ids) node_ids[i_feature] = nid;
            if (weights) weights[i_feature] = w;
            if (w > 0) {  // not stopped
                if (tf) {  // BowVector::addWeight (BowVector.cpp:32-44)
                    auto vit = v.lower_bound(id);
                    if (vit != v.end() && !(v.key_comp()(id, vit->first))) vit->second += w;
                    else v.insert(vit, std::make_pair(id, w));
                } else {  // BowVector::addIfNotExist (:48-56)
                    auto vit = v.lower_bound(id);
                    if (vit == v.end() || (v.key_comp()(id, vit->first))) v.insert(vit, std::make_pair(id, w));
                }
                fv[nid].push_back((unsigned)i_feature);  // FeatureVector::addFeature (FeatureVector.cpp:31-45)
            }
        }
        if (tf && !v.empty() && !must) {  // :1164-1170
            const double nd = v.size();
            for (auto vit = v.begin(); vit != v.end(); vit++) vit->second /= nd;
        }
        if (must) {  // BowVector::normalize (BowVector.cpp:60-85)
            double norm = 0.0;
            if (!l2) {
                for (auto it = v.begin(); it != v.end(); ++it) norm += fabs(it->second);
            } else {
                for (auto it = v.begin(); it != v.end(); ++it) norm += it->second * it->second;
                norm = sqrt(norm);
            }
            if (norm > 0.0)
                for (auto it = v.begin(); it != v.end(); ++it) it->second /= norm;
        }
    }
    int m = 0;
    for (auto it = v.begin(); it != v.end(); ++it, ++m) {
        if (bow_ids) bow_ids[m] = it->first;
        if (bow_values) bow_values[m] = it->second;
    }
    if (n_bow) *n_bow = m;
    int j = 0, off = 0;
    for (auto it = fv.begin(); it != fv.end(); ++it, ++j) {
        if (fv_nodes) fv_nodes[j] = it->first;
        if (fv_offsets) fv_offsets[j] = off;
        for (unsigned f : it->second) {
            if (fv_features) fv_features[off] = f;
            off++;
        }
    }
    if (fv_offsets) fv_offsets[j] = off;
    if (n_fv) *n_fv = j;
}

// ------------------------------------------------------------------------------------------------
// ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches) - src/ORBmatcher.cc:322-524.
// The two FeatureVectors are walked like the reference's std::map iterators (equal node: match the node's features;
// otherwise lower_bound on the other side).  TH_LOW = 50 (:42), HISTO_LENGTH = 30 (:43).
// ------------------------------------------------------------------------------------------------
int orc_search_by_bow(const orc_bow_side *K, const uint8_t *kf_has_point, const orc_bow_side *F, int f_nleft, float nn_ratio,
                      int check_orientation, int *matches) {
    const int TH_LOW = 50;
    for (int i = 0; i < F->n; i++) matches[i] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    auto pushRot = [&](int kfIdx, int fIdx) {
        float rot = K->angles[kfIdx] - F->angles[fIdx];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back(fIdx);
    };
    int a = 0, b = 0;
    while (a < K->n_nodes && b < F->n_nodes) {
        if (K->fv_nodes[a] == F->fv_nodes[b]) {
            for (int ik = K->fv_offsets[a]; ik < K->fv_offsets[a + 1]; ik++) {
                const int realIdxKF = (int)K->fv_features[ik];
                if (!kf_has_point[realIdxKF]) continue;  // !pMP || pMP->isBad()
                const uint8_t *dKF = K->descriptors + 32 * (size_t)realIdxKF;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                int bestDist1R = 256, bestIdxFR = -1, bestDist2R = 256;
                for (int jf = F->fv_offsets[b]; jf < F->fv_offsets[b + 1]; jf++) {
                    const int realIdxF = (int)F->fv_features[jf];
                    if (matches[realIdxF] >= 0) continue;
                    const int dist = orc_descriptor_distance(dKF, F->descriptors + 32 * (size_t)realIdxF);
                    if (f_nleft == -1) {
                        if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                        else if (dist < bestDist2) bestDist2 = dist;
                    } else {
                        if (realIdxF < f_nleft && dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                        else if (realIdxF < f_nleft && dist < bestDist2) bestDist2 = dist;
                        if (realIdxF >= f_nleft && dist < bestDist1R) { bestDist2R = bestDist1R; bestDist1R = dist; bestIdxFR = realIdxF; }
                        else if (realIdxF >= f_nleft && dist < bestDist2R) bestDist2R = dist;
                    }
                }
                if (bestDist1 <= TH_LOW) {
                    if ((float)bestDist1 < nn_ratio * (float)bestDist2) {
                        matches[bestIdxF] = realIdxKF;
                        if (check_orientation) pushRot(realIdxKF, bestIdxF);
                        nmatches++;
                    }
                    if (bestDist1R <= TH_LOW) {  // the ratio test of the right camera is disabled in the reference ("|| true", :460)
                        matches[bestIdxFR] = realIdxKF;
                        if (check_orientation) pushRot(realIdxKF, bestIdxFR);
                        nmatches++;
                    }
                }
            }
            a++;
            b++;
        } else if (K->fv_nodes[a] < F->fv_nodes[b]) {
            while (a < K->n_nodes && K->fv_nodes[a] < F->fv_nodes[b]) a++;  // lower_bound(Fit->first)
        } else {
            while (b < F->n_nodes && F->fv_nodes[b] < K->fv_nodes[a]) b++;
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1, sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        orc_three_maxima(sizes, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : rotHist[i]) {
                matches[idx] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

}  // extern "C"
