// HIP kernels of the ORB extractor for gfx950 (CDNA4, wave64).  No MFMA anywhere: the path is
// integer / bitwise.  Results equal the reference's CPU branch:
//   k_pyr_down     cv::resize INTER_LINEAR 8UC1 fixed point     (ORBextractor.cc:1495-1520)
//   k_fast_cells   per-cell cv::FAST-9/16 + NMS + threshold fallback (ORBextractor.cc:1136-1199, A.3)
//   k_compact      cell-row-major ordered candidate list          (ORBextractor.cc:1186-1198)
//   k_orient_desc  IC_Angle + 7x7 fixed-point blur + rBRIEF       (ORBextractor.cc:39-108,1456-1462, A.2/A.4/A.6/A.7)
#include "ft_internal.h"

namespace {

__constant__ signed char c_pattern[1024] = {
#include "orb_pattern.inc"
};
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
// fixed-point taps of GaussianBlur(7x7, sigma 2): error-diffused, sum 256
__constant__ int c_gauss[7] = {18, 34, 48, 56, 48, 34, 18};

__device__ __forceinline__ const uint8_t *level_ptr(const FtGeom &g, int level, int slot, const uint8_t *const *l0,
                                                    int l0pitch, const uint8_t *pyr, int &pitch) {
    if (level == 0) {
        pitch = l0pitch;
        return l0[slot];
    }
    pitch = g.lv[level].pitch;
    return pyr + (size_t)slot * g.pyrPerSlot + g.lv[level].off;
}

// ------------------------------------------------------------------------------------------------
// Pyramid: level l from level l-1 (one launch per level; grid.z = image slot)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pyr_down(FtGeom g, int level, const uint8_t *const *l0, int l0pitch,
                                                  uint8_t *pyr, const FtTap *taps) {
    const int dx = blockIdx.x * 64 + threadIdx.x;
    const int dy = blockIdx.y * 4 + threadIdx.y;
    const int slot = blockIdx.z;
    const FtLevelGeom &D = g.lv[level];
    if (dx >= D.w || dy >= D.h) return;
    int spitch;
    const uint8_t *S = level_ptr(g, level - 1, slot, l0, l0pitch, pyr, spitch);
    const int sw = g.lv[level - 1].w, sh = g.lv[level - 1].h;
    uint8_t *out = pyr + (size_t)slot * g.pyrPerSlot + D.off + (size_t)dy * D.pitch + dx;
    if (D.area2x) {
        const uint8_t *r0 = S + (size_t)(2 * dy) * spitch + 2 * dx, *r1 = r0 + spitch;
        *out = (uint8_t)((r0[0] + r0[1] + r1[0] + r1[1] + 2) >> 2);
        return;
    }
    const FtTap xt = taps[D.xtab + dx], yt = taps[D.ytab + dy];
    const int sx0 = xt.s, sx1 = min(sx0 + 1, sw - 1);
    const int sy0 = min(max((int)yt.s, 0), sh - 1), sy1 = min(max((int)yt.s + 1, 0), sh - 1);
    const uint8_t *r0 = S + (size_t)sy0 * spitch, *r1 = S + (size_t)sy1 * spitch;
    const int h0 = r0[sx0] * xt.a0 + r0[sx1] * xt.a1;
    const int h1 = r1[sx0] * xt.a0 + r1[sx1] * xt.a1;
    *out = (uint8_t)((((yt.a0 * (h0 >> 4)) >> 16) + ((yt.a1 * (h1 >> 4)) >> 16) + 2) >> 2);
}

// ------------------------------------------------------------------------------------------------
// FAST-9/16 per cell.  One workgroup per (cell, image).  The cell's (wCell+6)x(hCell+6) uint8 tile is
// staged in LDS; scores (largest threshold at which the pixel is still a corner) go to an LDS score
// plane whose zero rim implements "a neighbour belonging to another cell counts as 0"; survivors are
// compacted in row-major order with ballot/popcount prefix sums (no atomics: the octree's result
// depends on candidate order).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool has_arc9(unsigned m) {
    unsigned d = m | (m << 16);
    unsigned a = d & (d >> 1);
    a &= a >> 2;
    a &= a >> 4;
    a &= d >> 8;
    return (a & 0xffffu) != 0;
}

// ring offsets (x,y) k = 0..15: ORBextractor.cc:418-419
#define FT_RING(F)                                                                                            \
    F(0, 0, 3) F(1, 1, 3) F(2, 2, 2) F(3, 3, 1) F(4, 3, 0) F(5, 3, -1) F(6, 2, -2) F(7, 1, -3) F(8, 0, -3)    \
    F(9, -1, -3) F(10, -2, -2) F(11, -3, -1) F(12, -3, 0) F(13, -3, 1) F(14, -2, 2) F(15, -1, 3)

__device__ __forceinline__ int fast_score(const int d[16]) {
    // score = max over the 16 arcs of 9 of min(d) (either polarity) - 1  == cornerScore<16> for corners
    int mn2[16], mx2[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mn2[k] = min(d[k], d[(k + 1) & 15]);
        mx2[k] = max(d[k], d[(k + 1) & 15]);
    }
    int mn4[16], mx4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mn4[k] = min(mn2[k], mn2[(k + 2) & 15]);
        mx4[k] = max(mx2[k], mx2[(k + 2) & 15]);
    }
    int best = -256, worst = 256;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int mn9 = min(min(mn4[k], mn4[(k + 4) & 15]), d[(k + 8) & 15]);
        int mx9 = max(max(mx4[k], mx4[(k + 4) & 15]), d[(k + 8) & 15]);
        best = max(best, mn9);
        worst = min(worst, mx9);
    }
    return max(best, -worst) - 1;
}

__global__ __launch_bounds__(256) void k_fast_cells(FtGeom g, const uint8_t *const *l0, int l0pitch,
                                                    const uint8_t *pyr, int iniTh, int minTh, int *cellCount,
                                                    uint32_t *stage) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = blockIdx.y, cell = blockIdx.x;
    int level = 0;
    while (level + 1 < g.nlevels && cell >= g.lv[level + 1].cellBase) level++;
    const FtLevelGeom &L = g.lv[level];
    const int c = cell - L.cellBase;
    const int ci = c / L.nCols, cj = c - ci * L.nCols;
    const int iniX = 16 + cj * L.wCell, iniY = 16 + ci * L.hCell;
    const int maxX = min(iniX + L.wCell + 6, L.maxBX), maxY = min(iniY + L.hCell + 6, L.maxBY);
    int *cnt = cellCount + (size_t)slot * g.totalCells + cell;
    const int tw = maxX - iniX, th = maxY - iniY;
    // ORBextractor.cc:1141,1150 skip rules; cv::FAST finds nothing in a sub-image under 7 px
    if (iniX >= L.maxBX - 6 || iniY >= L.maxBY - 3 || tw < 7 || th < 7) {
        if (tid == 0) *cnt = 0;
        return;
    }
    const int tp = (tw + 3) & ~3;            // tile pitch
    const int pw = tw - 6, ph = th - 6;      // tested region
    const int sp = pw + 2;                   // score plane pitch (1-px zero rim)
    uint8_t *tile = smem;                    // th * tp
    uint8_t *score = tile + (((L.hCell + 6) * ((L.wCell + 6 + 3) & ~3) + 15) & ~15);
    int *wcnt = (int *)(score + ((((L.hCell + 2) * (L.wCell + 2)) + 15) & ~15));
    int pitch;
    const uint8_t *img = level_ptr(g, level, slot, l0, l0pitch, pyr, pitch);
    const uint8_t *src = img + (size_t)iniY * pitch + iniX;
    for (int i = tid; i < tw * th; i += 256) {
        int y = i / tw, x = i - y * tw;
        tile[y * tp + x] = src[(size_t)y * pitch + x];
    }
    for (int i = tid; i < sp * (ph + 2); i += 256) score[i] = 0;
    __syncthreads();
    const int npx = pw * ph;
    for (int i = tid; i < npx; i += 256) {
        const int y = i / pw, x = i - y * pw;
        const uint8_t *cpx = tile + (y + 3) * tp + (x + 3);
        const int v = cpx[0];
        int d[16];
#define FT_LD(k, ox, oy) d[k] = v - (int)cpx[(oy)*tp + (ox)];
        FT_RING(FT_LD)
#undef FT_LD
        unsigned dark = 0, bright = 0;  // ring pixel darker / brighter than the centre by more than minTh
#pragma unroll
        for (int k = 0; k < 16; k++) {
            dark |= (d[k] > minTh ? 1u : 0u) << k;
            bright |= (d[k] < -minTh ? 1u : 0u) << k;
        }
        if (has_arc9(dark) || has_arc9(bright)) score[(y + 1) * sp + (x + 1)] = (uint8_t)fast_score(d);
    }
    __syncthreads();
    // NMS: strictly greater than the 8 neighbours (cv::FAST); the result replaces the raw tile
    uint8_t *surv = tile;
    int localHi = 0;
    for (int i = tid; i < npx; i += 256) {
        const int y = i / pw, x = i - y * pw;
        const uint8_t *s = score + (y + 1) * sp + (x + 1);
        const int v = s[0];
        bool keep = v > 0 && v > s[-1] && v > s[1] && v > s[-sp - 1] && v > s[-sp] && v > s[-sp + 1] &&
                    v > s[sp - 1] && v > s[sp] && v > s[sp + 1];
        surv[i] = keep ? 1 : 0;
        if (keep && v >= iniTh) localHi = 1;
    }
    // cell-level threshold fallback (ORBextractor.cc:1157-1177): if any survivor reaches iniThFAST only
    // those are emitted, otherwise every minThFAST survivor is
    const int anyHi = __syncthreads_or(localHi);
    const int emitTh = anyHi ? iniTh : minTh;
    uint32_t *out = stage + (size_t)slot * g.stagePerSlot + L.stageBase + (size_t)c * L.cellCap;
    int running = 0;
    for (int base = 0; base < npx; base += 256) {
        const int i = base + tid;
        int y = 0, x = 0, sc = 0;
        bool f = false;
        if (i < npx) {
            y = i / pw;
            x = i - y * pw;
            sc = score[(y + 1) * sp + (x + 1)];
            f = surv[i] && sc >= emitTh;
        }
        const unsigned long long b = __ballot(f);
        if (lane == 0) wcnt[wave] = __popcll(b);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int cw = wcnt[w];
            if (w < wave) woff += cw;
            tot += cw;
        }
        if (f) {
            const int pos = running + woff + __popcll(b & ((1ull << lane) - 1ull));
            // keypoint (x+3, y+3) in the cell sub-image, shifted by (j*wCell, i*hCell): ORBextractor.cc:1196-1197
            if (pos < L.cellCap) out[pos] = ft_pack_cand(x + 3 + cj * L.wCell, y + 3 + ci * L.hCell, sc);
        }
        running += tot;
        __syncthreads();
    }
    if (tid == 0) *cnt = min(running, L.cellCap);
}

// ------------------------------------------------------------------------------------------------
// Ordered compaction: one workgroup per (level, image).  Exclusive scan of the cell counts in cell
// order, then a coalesced gather into the dense list (which lives in host-mapped pinned memory so
// the host octree can read it after a single stream sync).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compact(FtGeom g, const int *cellCount, const uint32_t *stage,
                                                 uint32_t *cand, int *candCount) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int *offs = (int *)smem;  // nCells + 1
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int level = blockIdx.x, slot = blockIdx.y;
    const FtLevelGeom &L = g.lv[level];
    const int nCells = L.nCols * L.nRows;
    const int *cnt = cellCount + (size_t)slot * g.totalCells + L.cellBase;
    const int per = (nCells + 255) / 256;
    const int c0 = tid * per, c1 = min(c0 + per, nCells);
    int local = 0;
    for (int c = c0; c < c1; c++) local += cnt[c];
    int incl = local;  // inclusive scan across the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if (w < wave) wbase += wsum[w];
        total += wsum[w];
    }
    int run = wbase + incl - local;
    for (int c = c0; c < c1; c++) {
        offs[c] = run;
        run += cnt[c];
    }
    if (tid == 0) offs[nCells] = total;
    __syncthreads();
    total = min(total, L.candCap);
    const uint32_t *st = stage + (size_t)slot * g.stagePerSlot + L.stageBase;
    uint32_t *dst = cand + (size_t)slot * g.candPerSlot + L.candBase;
    for (int o = tid; o < total; o += 256) {
        int lo = 0, hi = nCells;  // last cell with offs[cell] <= o
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (offs[mid] <= o) lo = mid;
            else hi = mid;
        }
        dst[o] = st[(size_t)lo * L.cellCap + (o - offs[lo])];
    }
    if (tid == 0) candCount[slot * g.nlevels + level] = total;
}

// ------------------------------------------------------------------------------------------------
// Orientation + descriptor: one wave per retained keypoint, four keypoints per workgroup.
// The 43x43 unblurred patch (31-px disc for the moments, 37x37 sample window + 3-px blur halo) is
// staged in LDS with BORDER_REFLECT_101 at the level's edges; the 7x7 Gaussian is applied on the fly
// as two integer 7-tap passes in LDS, so no blurred pyramid is ever written to HBM.
// ------------------------------------------------------------------------------------------------
#define OD_R 21                 // patch radius: 18 (max rotated pattern offset) + 3 (blur)
#define OD_P (2 * OD_R + 1)     // 43
#define OD_PP 44                // raw pitch
#define OD_B 37                 // blurred window
#define OD_WAVE_BYTES (OD_P * OD_PP + OD_P * OD_B * 2 + OD_B * OD_B + 3)

__device__ __forceinline__ int reflect101(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    // cv::fastAtan2; every operation rounded separately (no FMA contraction)
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

__global__ __launch_bounds__(256) void k_orient_desc(FtGeom g, const uint8_t *const *l0, int l0pitch,
                                                     const uint8_t *pyr, const FtSelKp *sel, const int *nSel,
                                                     ft_keypoint *keysOut, uint8_t *descOut) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = blockIdx.y;
    const int k = blockIdx.x * 4 + wave;
    const bool valid = k < nSel[slot];
    uint8_t *raw = smem + (size_t)wave * ((OD_WAVE_BYTES + 15) & ~15);
    unsigned short *hb = (unsigned short *)(raw + OD_P * OD_PP);
    uint8_t *bl = (uint8_t *)(hb + OD_P * OD_B);
    int cx = 0, cy = 0, level = 0, response = 0;
    if (valid) {
        const FtSelKp s = sel[(size_t)slot * g.maxKp + k];
        cx = s.x;
        cy = s.y;
        level = s.level;
        response = s.response;
        int pitch;
        const uint8_t *img = level_ptr(g, level, slot, l0, l0pitch, pyr, pitch);
        const int w = g.lv[level].w, h = g.lv[level].h;
        for (int i = lane; i < OD_P * OD_P; i += 64) {
            const int r = i / OD_P, c = i - r * OD_P;
            const int gy = reflect101(cy - OD_R + r, h), gx = reflect101(cx - OD_R + c, w);
            raw[r * OD_PP + c] = img[(size_t)gy * pitch + gx];
        }
    }
    __syncthreads();
    float angle = 0.f;
    if (valid) {
        // IC_Angle: integer moments over the 31-px disc (two patch rows per step: lanes 0-31 / 32-63)
        int m10 = 0, m01 = 0;
        const int u = (lane & 31) - 15;
        for (int r = 0; r < 32; r += 2) {
            const int v = r + (lane >> 5) - 15;
            if (v <= 15 && u <= 15) {
                const int av = v < 0 ? -v : v, au = u < 0 ? -u : u;
                if (au <= c_umax[av]) {
                    const int I = raw[(OD_R + v) * OD_PP + (OD_R + u)];
                    m10 += u * I;
                    m01 += v * I;
                }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            m10 += __shfl_xor(m10, o);
            m01 += __shfl_xor(m01, o);
        }
        angle = fast_atan2_deg((float)m01, (float)m10);
        // horizontal 7-tap pass: hb[r][c] for r in 0..42, c in 0..36 (16-bit, <= 255*256)
        for (int i = lane; i < OD_P * OD_B; i += 64) {
            const int r = i / OD_B, c = i - r * OD_B;
            const uint8_t *p = raw + r * OD_PP + c;
            unsigned a = 0;
#pragma unroll
            for (int t = 0; t < 7; t++) a += (unsigned)c_gauss[t] * p[t];
            hb[r * OD_B + c] = (unsigned short)a;
        }
    }
    __syncthreads();
    if (valid) {
        for (int i = lane; i < OD_B * OD_B; i += 64) {
            const int r = i / OD_B, c = i - r * OD_B;
            unsigned a = 0;
#pragma unroll
            for (int t = 0; t < 7; t++) a += (unsigned)c_gauss[t] * hb[(r + t) * OD_B + c];
            bl[i] = (uint8_t)((a + 32768u) >> 16);
        }
    }
    __syncthreads();
    if (valid) {
        // computeOrbDescriptor: angle in radians as float, cos/sin in double then narrowed
        const float factorPI = (float)(3.14159265358979323846 / 180.f);
        const float ar = __fmul_rn(angle, factorPI);
        const float ca = (float)cos((double)ar), sb = (float)sin((double)ar);
        unsigned long long words[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int p = q * 64 + lane;  // pair index: byte p/8, bit p%8
            const float x0 = (float)c_pattern[4 * p], y0 = (float)c_pattern[4 * p + 1];
            const float x1 = (float)c_pattern[4 * p + 2], y1 = (float)c_pattern[4 * p + 3];
            const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, sb), __fmul_rn(y0, ca)));
            const int c0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, ca), __fmul_rn(y0, sb)));
            const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, sb), __fmul_rn(y1, ca)));
            const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, ca), __fmul_rn(y1, sb)));
            const int t0 = bl[(18 + r0) * OD_B + (18 + c0)];
            const int t1 = bl[(18 + r1) * OD_B + (18 + c1)];
            words[q] = __ballot(t0 < t1);
        }
        if (lane == 0) {
            const size_t o = (size_t)slot * g.maxKp + k;
            // ORBextractor.cc:1209-1221 (octave, size = int(PATCH_SIZE * sf)) and :1472-1475 (pt *= scale)
            const float scale = g.sf[level];
            ft_keypoint kp;
            kp.x = level ? __fmul_rn((float)cx, scale) : (float)cx;
            kp.y = level ? __fmul_rn((float)cy, scale) : (float)cy;
            kp.size = (float)(int)__fmul_rn((float)FT_PATCH_SIZE, scale);
            kp.angle = angle;
            kp.response = (float)response;
            kp.octave = level;
            kp.class_id = -1;
            keysOut[o] = kp;
            unsigned long long *d = (unsigned long long *)(descOut + o * 32);
            d[0] = words[0];
            d[1] = words[1];
            d[2] = words[2];
            d[3] = words[3];
        }
    }
}

}  // namespace

size_t ft_fast_smem_bytes(const FtGeom &g) {
    size_t mx = 0;
    for (int l = 0; l < g.nlevels; l++) {
        const FtLevelGeom &L = g.lv[l];
        size_t tile = (size_t)(((L.hCell + 6) * ((L.wCell + 6 + 3) & ~3) + 15) & ~15);
        size_t sc = (size_t)((((L.hCell + 2) * (L.wCell + 2)) + 15) & ~15);
        // surv bytes reuse the tile: needs wCell*hCell <= tile bytes (true: (w+6)*(h+6) > w*h)
        mx = std::max(mx, tile + sc + 16);
    }
    return mx;
}

int ft_launch_pyramid(hipStream_t st, const FtGeom &g, int batch, const uint8_t *const *l0, int l0pitch,
                      uint8_t *pyr, const FtTap *taps) {
    for (int level = 1; level < g.nlevels; level++) {
        const FtLevelGeom &D = g.lv[level];
        dim3 grid((D.w + 63) / 64, (D.h + 3) / 4, batch), block(64, 4, 1);
        hipLaunchKernelGGL(k_pyr_down, grid, block, 0, st, g, level, l0, l0pitch, pyr, taps);
    }
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_fast_cells(hipStream_t st, const FtGeom &g, int batch, const uint8_t *const *l0, int l0pitch,
                         const uint8_t *pyr, int iniTh, int minTh, int *cellCount, uint32_t *stage) {
    dim3 grid(g.totalCells, batch, 1), block(256, 1, 1);
    hipLaunchKernelGGL(k_fast_cells, grid, block, ft_fast_smem_bytes(g), st, g, l0, l0pitch, pyr, iniTh, minTh,
                       cellCount, stage);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_compact(hipStream_t st, const FtGeom &g, int batch, const int *cellCount, const uint32_t *stage,
                      uint32_t *cand, int *candCount) {
    int maxCells = 0;
    for (int l = 0; l < g.nlevels; l++) maxCells = std::max(maxCells, g.lv[l].nCols * g.lv[l].nRows);
    dim3 grid2(g.nlevels, batch, 1), block(256, 1, 1);
    hipLaunchKernelGGL(k_compact, grid2, block, (size_t)(maxCells + 1) * sizeof(int), st, g, cellCount, stage, cand,
                       candCount);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_orient_desc(hipStream_t st, const FtGeom &g, int batch, const uint8_t *const *l0, int l0pitch,
                          const uint8_t *pyr, const FtSelKp *sel, const int *nSel, ft_keypoint *keys,
                          uint8_t *desc) {
    dim3 grid((g.maxKp + 3) / 4, batch, 1), block(256, 1, 1);
    const size_t smem = 4 * (size_t)((OD_WAVE_BYTES + 15) & ~15);
    hipLaunchKernelGGL(k_orient_desc, grid, block, smem, st, g, l0, l0pitch, pyr, sel, nSel, keys, desc);
    FT_HIP(hipGetLastError());
    return FT_OK;
}
