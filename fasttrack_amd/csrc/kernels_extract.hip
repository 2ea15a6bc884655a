// HIP kernels of the ORB extractor for gfx950 (CDNA4, wave64).  No MFMA anywhere: the path is
// integer / bitwise.  Results equal the reference's CPU branch:
//   k_pyr_down     cv::resize INTER_LINEAR 8UC1 fixed point     (ORBextractor.cc:1495-1520)
//   k_fast_cells   per-cell cv::FAST-9/16 + NMS + threshold fallback (ORBextractor.cc:1136-1199, A.3)
//   k_compact      cell-row-major ordered candidate list          (ORBextractor.cc:1186-1198)
//   k_orient_desc  IC_Angle + 7x7 fixed-point blur + rBRIEF       (ORBextractor.cc:39-108,1456-1462, A.2/A.4/A.6/A.7)
#include "ft_internal.h"
#include "libm_f32.h"
#include "wave_ops.h"

namespace {

// umax of IC_Angle (ORBextractor.cc:478-493): 15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3

// IC_Angle with v_dot4_u32_u8: row v of the 31-px disc is eight dwords of four pixels, dword j holding
// u = -15 + 4j .. -12 + 4j.  W has u + 16 in the bytes inside the disc (|u| <= umax[|v|]), M has 1 there; both
// are 0 outside, so m10 = sum dot4(D, W) - 16 * dot4(D, M) and m01 = sum v * dot4(D, M).  Row 16 is all zero
// (padding of the 4 x 8-row walk).
struct FtMomentTab {
    unsigned W[17 * 8], M[17 * 8];
};
constexpr FtMomentTab ft_make_moment_tab() {
    constexpr int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    FtMomentTab t{};
    for (int av = 0; av < 16; av++)
        for (int j = 0; j < 8; j++)
            for (int k = 0; k < 4; k++) {
                const int u = -15 + 4 * j + k;
                const int au = u < 0 ? -u : u;
                if (u <= 15 && au <= umax[av]) {
                    t.W[av * 8 + j] |= (unsigned)(u + 16) << (8 * k);
                    t.M[av * 8 + j] |= 1u << (8 * k);
                }
            }
    return t;
}
__constant__ FtMomentTab c_mom = ft_make_moment_tab();
// the rBRIEF pattern as floats: (x0, y0, x1, y1) per test
struct FtPatternF {
    float4 p[256];
};
constexpr FtPatternF ft_make_pattern_f() {
    constexpr signed char pat[1024] = {
#include "orb_pattern.inc"
    };
    FtPatternF t{};
    for (int i = 0; i < 256; i++) {
        t.p[i].x = (float)pat[4 * i];
        t.p[i].y = (float)pat[4 * i + 1];
        t.p[i].z = (float)pat[4 * i + 2];
        t.p[i].w = (float)pat[4 * i + 3];
    }
    return t;
}
__constant__ FtPatternF c_patternF = ft_make_pattern_f();
// GaussianBlur(7x7, sigma 2) fixed-point taps {18,34,48,56,48,34,18} (error-diffused, sum 256) are literals in k_orient_desc


// i / d for small operands with a precomputed magic = floor(2^32 / d) + 1 (exact for i < 2^32 / d);
// d == 1 makes the magic wrap to 0, so it is special-cased.
__host__ __device__ constexpr unsigned div_magic_of(unsigned d) { return d > 1 ? 0xffffffffu / d + 1u : 0u; }
// The divisors that occur per wave (dwords per footprint row, columns behind the 32-column halves of a cell) are
// small: their magics come from a constant table (one scalar load) instead of a ~25-instruction integer division.
struct FtDivMagicTab {
    unsigned m[128];
};
constexpr FtDivMagicTab ft_make_div_magic_tab() {
    FtDivMagicTab t{};
    for (unsigned d = 0; d < 128; d++) t.m[d] = d ? div_magic_of(d) : 0u;
    return t;
}
__constant__ FtDivMagicTab c_divMagic = ft_make_div_magic_tab();
__device__ __forceinline__ unsigned div_magic_small(unsigned d) { return d < 128u ? c_divMagic.m[d] : div_magic_of(d); }
__device__ __forceinline__ int div_by(int i, unsigned magic) { return magic ? (int)__umulhi((unsigned)i, magic) : i; }

__device__ __forceinline__ const uint8_t *level_ptr(const FtGeom &g, int level, int slot, const uint8_t *const *l0,
                                                    int l0pitch, const uint8_t *pyr, int &pitch) {
    if (level == 0) {
        pitch = l0pitch;
        return l0[slot];
    }
    pitch = g.lv[level].pitch;
    return pyr + (size_t)slot * g.pyrPerSlot + g.lv[level].off;
}

// LDS hand-off between the lanes of ONE wave: the LDS queue of a wave is served in order, so only the
// compiler has to be kept from moving accesses across this point.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------
// Pyramid: level l from level l-1 (one launch per level; grid.z = image slot)
// ------------------------------------------------------------------------------------------------
// One WAVE per 64 x 8 output tile (no workgroup barrier, like k_fast_cells): the source footprint of the
// tile (rows sy(first)..sy(last)+1, columns sx(first)..sx(last)+1) is staged in LDS with aligned dword
// loads, then every lane produces a 4 x 4 block of outputs from 2x2 taps read from LDS and stores four dwords.
#define PD_TW 64
#define PD_TH 16
#define PD_NLOAD 9  // footprint dwords per lane in flight at once
// A wave's life is a chain of memory round trips, so the chain is kept short: the source footprint of the tile is
// bounded arithmetically (fixed-point scale with a two-pixel margin instead of reading the first and last tap),
// and the taps the outputs need - four in x and four in y per lane - are requested together with the footprint,
// before anything waits.
__global__ __launch_bounds__(64) void k_pyr_down(FtGeom g, int level, const uint8_t *const *l0, int l0pitch,
                                                 uint8_t *pyr, const FtTap *taps, int alignedLoads, int ldsPitch,
                                                 int rowsAlloc, unsigned sxQ16, unsigned syQ16, FtSlotGrid sg, int tilesX,
                                                 unsigned txMagic) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x;
    // all tiles of an image on one XCD (ft_slot_grid): neighbouring tiles share source lines (footprint margins, 64-B
    // lines), which then come from that XCD's L2 instead of being fetched once per XCD
    int slot, tile;
    if (!ft_slot_block(sg, slot, tile)) return;
    const int tileY = div_by(tile, txMagic), tileX = tile - tileY * tilesX;
    const FtLevelGeom &D = g.lv[level];
    const int dx0 = tileX * PD_TW, dy0 = tileY * PD_TH;
    const int dx1 = min(dx0 + PD_TW, D.w) - 1, dy1 = min(dy0 + PD_TH, D.h) - 1;  // last output col / row
    int spitch;
    const uint8_t *S = level_ptr(g, level - 1, slot, l0, l0pitch, pyr, spitch);
    const int sw = g.lv[level - 1].w, sh = g.lv[level - 1].h;
    int sxa, sxb, sya, syb;  // source footprint (inclusive)
    if (D.area2x) {
        sxa = 2 * dx0; sxb = 2 * dx1 + 1; sya = 2 * dy0; syb = 2 * dy1 + 1;
    } else {
        // first tap of output d is floor((d + 0.5) * scale - 0.5); ((2d + 1) * scaleQ16) >> 17 is within one of it
        sxa = max((int)(((unsigned)(2 * dx0 + 1) * sxQ16) >> 17) - 2, 0);
        sxb = min((int)(((unsigned)(2 * dx1 + 1) * sxQ16) >> 17) + 2, sw - 1);
        sya = max((int)(((unsigned)(2 * dy0 + 1) * syQ16) >> 17) - 2, 0);
        syb = min((int)(((unsigned)(2 * dy1 + 1) * syQ16) >> 17) + 2, sh - 1);
    }
    // compute mapping: lane = (group of 4 output columns, group of 4 output rows): 16 x 4 groups cover the 64 x 16 tile,
    // every lane produces a 4 x 4 block and stores it as four dwords (a row of the tile is 64 B = one store segment)
    const int xg = lane & 15, yg = lane >> 4;
    const int bx = dx0 + 4 * xg, by = dy0 + 4 * yg;
    FtTap xt[4], yt[4];
    if (!D.area2x) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            xt[k] = taps[D.xtab + min(bx + k, dx1)];
            yt[k] = taps[D.ytab + min(by + k, dy1)];
        }
    }
    const int rows = syb - sya + 1;
    int ax = 0;
    if (alignedLoads) {
        ax = sxa & 3;
        const int nd = (sxb - sxa + 1 + ax + 3) >> 2;  // dwords per row; over-read stays inside the aligned row pitch
        const uint8_t *src = S + (size_t)sya * spitch + (sxa - ax);
        // all loads of the footprint are issued before the first LDS store (PD_NLOAD x 64 dwords cover the usual
        // footprint; a larger one takes further rounds).  Element i = lane + 64 j of the (rows x nd) footprint is
        // walked incrementally - 64 elements further is dy rows and dx dwords further, one more row on wrap-around -
        // so a load costs a few full-rate adds instead of a division and 64-bit address arithmetic.
        const int n = nd * rows;
        const int spare = ldsPitch * rowsAlloc;  // one dword behind the tile (see the launcher)
        const unsigned ndMagic = div_magic_small((unsigned)nd);
        const int dy = div_by(64, ndMagic), dx = 64 - dy * nd;
        int x = lane - div_by(lane, ndMagic) * nd;
        unsigned gOff = (unsigned)(div_by(lane, ndMagic) * spitch + 4 * x);
        int lOff = div_by(lane, ndMagic) * ldsPitch + 4 * x;
        const unsigned gStep = (unsigned)(dy * spitch + 4 * dx), gStepW = gStep + (unsigned)(spitch - 4 * nd);
        const int lStep = dy * ldsPitch + 4 * dx, lStepW = lStep + ldsPitch - 4 * nd;
        for (int i0 = 0; i0 < n; i0 += 64 * PD_NLOAD) {
            unsigned v[PD_NLOAD];
            int off[PD_NLOAD];
#pragma unroll
            for (int k = 0; k < PD_NLOAD; k++) {
                // lanes past the footprint re-read its first dword and park it in the spare dword behind the tile:
                // selects instead of exec-mask branches around every load and store
                const bool ok = i0 + 64 * k + lane < n;
                off[k] = ok ? lOff : spare;
                v[k] = gload<unsigned>(src + (ok ? gOff : 0u));
                x += dx;
                const bool wrap = x >= nd;
                x -= wrap ? nd : 0;
                gOff += wrap ? gStepW : gStep;
                lOff += wrap ? lStepW : lStep;
            }
#pragma unroll
            for (int k = 0; k < PD_NLOAD; k++) *(unsigned *)(smem + off[k]) = v[k];
        }
    } else {
        const int cw = sxb - sxa + 1;
        const unsigned cwMagic = div_magic_of((unsigned)cw);
        const uint8_t *src = S + (size_t)sya * spitch + sxa;
        for (int i = lane; i < cw * rows; i += 64) {
            const int y = div_by(i, cwMagic), x = i - y * cw;
            smem[y * ldsPitch + x] = gload<uint8_t>(src + (size_t)y * spitch + x);
        }
    }
    wave_lds_sync();
    const uint8_t *T = smem + ax;  // source pixel (sx, sy) at T[(sy - sya) * ldsPitch + (sx - sxa)]
    if (bx > dx1 || by > dy1) return;
    // uniform level base + 32-bit lane offset (row * pitch + column): scalar-base stores, no 64-bit lane arithmetic
    uint8_t *outLevel = pyr + (size_t)slot * g.pyrPerSlot + D.off;
    // the row pitch of a level is a multiple of 64 B, so the dword store may run past the last column of the level
    // (into the row's padding) but never into the next row
    if (D.area2x) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int dy = by + j;
            if (dy > dy1) break;
            unsigned pk = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int dxk = min(bx + k, dx1);
                const uint8_t *r0 = T + __mul24(2 * dy - sya, ldsPitch) + (2 * dxk - sxa), *r1 = r0 + ldsPitch;
                pk |= (unsigned)((r0[0] + r0[1] + r1[0] + r1[1] + 2) >> 2) << (8 * k);
            }
            gstore<unsigned>(outLevel + (unsigned)vmad24(dy, D.pitch, bx), pk);
        }
        return;
    }
    int cx0[4], cx1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        cx0[k] = xt[k].s - sxa;
        cx1[k] = min(xt[k].s + 1, sw - 1) - sxa;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int dy = by + j;
        if (dy > dy1) break;
        const int sy0 = min(max((int)yt[j].s, 0), sh - 1) - sya, sy1 = min(max((int)yt[j].s + 1, 0), sh - 1) - sya;
        const uint8_t *r0 = T + __mul24(sy0, ldsPitch), *r1 = T + __mul24(sy1, ldsPitch);
        unsigned pk = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int h0 = r0[cx0[k]] * xt[k].a0 + r0[cx1[k]] * xt[k].a1;
            const int h1 = r1[cx0[k]] * xt[k].a0 + r1[cx1[k]] * xt[k].a1;
            // weights <= 2^11 and h >> 4 < 2^15: 24-bit multiplies (full rate; the 32-bit v_mul_lo is quarter rate)
            pk |= (unsigned)(((vmul24((int)yt[j].a0, h0 >> 4) >> 16) + (vmul24((int)yt[j].a1, h1 >> 4) >> 16) + 2) >> 2) << (8 * k);
        }
        gstore<unsigned>(outLevel + (unsigned)vmad24(dy, D.pitch, bx), pk);
    }
}

// ------------------------------------------------------------------------------------------------
// Pyramid, row-streaming form (the default): one WAVE per strip of 128 output columns x PR_RB output rows.
// A lane owns two adjacent output columns for the whole strip: their taps (source column, weights) live in registers,
// the eight source bytes that hold both columns' two taps are ONE aligned 8-byte load per source row, and the
// horizontal interpolation of a column is one v_perm (two bytes -> a pair of u16) + one v_dot2_u32_u16.  The wave
// walks the source rows of its strip once, top to bottom: cv::resize needs rows sy and sy + 1 for an output row and
// sy advances by one or two, so the horizontally interpolated row is kept in registers and serves as the lower row of
// one output row and the upper row of the next (1.2 instead of 2 interpolations per output row at scale 1.2).  No LDS,
// no workgroup barrier; row taps are wave-uniform (one lane holds the taps of one output row, v_readlane hands them
// out); the next source row is requested before the current one is used.  Per output pixel ~9 VALU instructions
// instead of ~32 in the tile kernel above, whose arithmetic it repeats bit for bit.
// ------------------------------------------------------------------------------------------------
#define PR_COLS 128
#ifndef PR_RB
#define PR_RB 32
#endif
#ifndef PR_PF
#define PR_PF 4       // source rows in flight ahead of the one being used
#endif
typedef unsigned short ft_us2 __attribute__((ext_vector_type(2)));
typedef unsigned ft_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned udot2_u16(unsigned a, unsigned b) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(ft_us2, a), __builtin_bit_cast(ft_us2, b), 0u, false);
}
// (b * a) >> 16 for a wave-uniform b < 2^11 handed in as b << 16 and a < 2^16: the high half of a 32 x 32-bit product - one
// instruction where a 24-bit multiply and a shift took two
__device__ __forceinline__ unsigned umulhi_su(unsigned bShifted16, unsigned a) {
    unsigned r;
    asm("v_mul_hi_u32 %0, %1, %2" : "=v"(r) : "s"(bShifted16), "v"(a));
    return r;
}
template <bool AREA>
__global__ __launch_bounds__(64) void k_pyr_rows(FtGeom g, int level, const uint8_t *const *l0, int l0pitch, uint8_t *pyr,
                                                 const FtTap *taps, FtSlotGrid sg, int stripsX, unsigned sxMagic,
                                                 int readableEnd) {
    const int lane = threadIdx.x;
    int slot, tile;
    if (!ft_slot_block(sg, slot, tile)) return;
    const int ty = div_by(tile, sxMagic), tx = tile - ty * stripsX;
    const FtLevelGeom &D = g.lv[level];
    int spitch;
    const uint8_t *S = level_ptr(g, level - 1, slot, l0, l0pitch, pyr, spitch);
    const int sw = g.lv[level - 1].w, sh = g.lv[level - 1].h;
    const int dx = tx * PR_COLS + 2 * lane;  // first of this lane's two output columns
    const int j0 = ty * PR_RB, nrows = min(PR_RB, D.h - j0);
    // row taps: lane r holds the two source rows (clamped to the level, packed sy0 | sy1 << 16) and the two weights
    // (b0 | b1 << 16) of output row j0 + r - unpacked and clamped once here, by the vector unit for all rows at a time;
    // the loop below reads a row's pair with two v_readlane and four scalar instructions
    unsigned rowW0 = 0, rowW1 = 0;
    if constexpr (!AREA) {
        const ft_u2 t = gload<ft_u2>(taps + D.ytab + j0 + min(lane, nrows - 1));
        const int sy = (int)(short)(t.x & 0xffffu);
        rowW0 = (unsigned)min(max(sy, 0), sh - 1) | ((unsigned)min(max(sy + 1, 0), sh - 1) << 16);
        rowW1 = (t.x >> 16) | (t.y << 16);
        // the lane behind the last output row holds a row pair no source row ever matches: the loop below runs out of output
        // rows by reading it, without a clamp and a test per row (nrows <= PR_RB < 64)
        if (lane >= nrows) rowW0 = 0xffffffffu;
    }
    // column taps; columns beyond the level repeat its last column (their stores are masked)
    const int dxa = min(dx, D.w - 1), dxb = min(dx + 1, D.w - 1);
    int sxa, sxb, cxa, cxb;
    unsigned wa, wb;
    if constexpr (AREA) {
        sxa = 2 * dxa; sxb = 2 * dxb; cxa = sxa + 1; cxb = sxb + 1;
        wa = wb = 0x00010001u;  // s0 + s1
    } else {
        const ft_u2 ta = gload<ft_u2>(taps + D.xtab + dxa), tb = gload<ft_u2>(taps + D.xtab + dxb);
        sxa = (int)(short)(ta.x & 0xffffu); sxb = (int)(short)(tb.x & 0xffffu);
        cxa = min(sxa + 1, sw - 1); cxb = min(sxb + 1, sw - 1);
        wa = (ta.x >> 16) | (ta.y << 16);  // a0 | a1 << 16
        wb = (tb.x >> 16) | (tb.y << 16);
    }
    // the 8-byte window [base, base + 8) of a source row that holds all four taps of the lane (the launcher has checked
    // that it does); it never reaches past the readable end of a row
    const int base = min(sxa & ~3, readableEnd - 8);
    const unsigned selA = (unsigned)(sxa - base) | 0x0c00u | ((unsigned)(cxa - base) << 16) | 0x0c000000u;
    const unsigned selB = (unsigned)(sxb - base) | 0x0c00u | ((unsigned)(cxb - base) << 16) | 0x0c000000u;
    auto rowTap = [&](int r, int &sy0, int &sy1, unsigned &b0, unsigned &b1) {
        if constexpr (AREA) {
            sy0 = 2 * (j0 + r); sy1 = r < nrows ? sy0 + 1 : -1; b0 = b1 = 0;
        } else {
            const unsigned w0 = (unsigned)__builtin_amdgcn_readlane((int)rowW0, r), w1 = (unsigned)__builtin_amdgcn_readlane((int)rowW1, r);
            sy0 = (int)(w0 & 0xffffu); sy1 = (int)(w0 >> 16);
            b0 = w1 << 16; b1 = w1 & 0xffff0000u;  // the weights as b << 16 (umulhi_su)
        }
    };
    int jr = 0, sy0, sy1;
    unsigned b0, b1;
    rowTap(0, sy0, sy1, b0, b1);
    int lastSy0, rLast;
    unsigned bx0, bx1;
    rowTap(nrows - 1, lastSy0, rLast, bx0, bx1);
    int r = sy0;
    // Fixed wave-uniform bases + 32-bit offsets (row offset: scalar, lane offset: vector): scalar-base loads and stores, no
    // 64-bit arithmetic in the loop.  The kernel issues MORE scalar than vector instructions per wave (833 against 721 in
    // round 3's profile, and a SIMD issues one of each per slot at best), so the loop's bookkeeping is written for the
    // scalar count: a row's offset is one multiply of its clamped index (pointer += select(pitch, 0) took five
    // instructions), the destination advances by one 32-bit add, the row taps end in a sentinel.
    const uint8_t *srcBase = S;
    uint8_t *dstBase = pyr + (size_t)slot * g.pyrPerSlot + D.off + (size_t)j0 * D.pitch;
    unsigned dOff = 0;
    const unsigned laneSrc = (unsigned)base, laneDst = (unsigned)dx;
    const bool store2 = dx + 1 < D.w, store1 = dx < D.w;
    // PR_PF source rows are in flight ahead of the one being used (a ring of registers, the loop unrolled over it): a
    // wave's life is a chain of dependent row loads, and a store counts on the same counter as a load, so with one row
    // in flight every wait would also wait for the stores just issued.
    // The loop exists twice: for strips inside the level, where every lane stores its two columns (no exec masks in the
    // loop), and for the last strip of a row of strips.
    auto run = [&](auto edgeTag) {
        constexpr bool EDGE = decltype(edgeTag)::value;
        int pfIdx = r;  // the next row to request: min(pfIdx, rLast) (the ring runs up to PR_PF - 1 rows past the strip's last)
        auto prefetch = [&]() -> ft_u2 {
            unsigned offS = laneSrc + (unsigned)min(pfIdx, rLast) * (unsigned)spitch;
            // (the empty asm keeps the zero-extension of the offset inside this block, where instruction selection can fold
            // it into the scalar-base addressing mode)
            asm volatile("" : "+v"(offS));
            const ft_u2 v = gload<ft_u2>(srcBase + offS);
            pfIdx++;
            return v;
        };
        ft_u2 q[PR_PF];
#pragma unroll
        for (int k = 0; k < PR_PF; k++) q[k] = prefetch();
        unsigned hpA = 0, hpB = 0, hcA = 0, hcB = 0;
        // a source row is used up by its two horizontal interpolations; the next row of the ring is requested into the same
        // registers right behind them (requested before, it would need registers of its own and the ring would have to be
        // copied around at the end of every trip - behind a wait for all of its loads)
        auto take = [&](ft_u2 &q) {
            hpA = hcA; hpB = hcB;
            hcA = udot2_u16(__builtin_amdgcn_perm(q.y, q.x, selA), wa);
            hcB = udot2_u16(__builtin_amdgcn_perm(q.y, q.x, selB), wb);
            if constexpr (!AREA) { hcA >>= 4; hcB >>= 4; }
            q = prefetch();
        };
        auto step = [&]() {
            while (sy1 == r) {  // wave-uniform; behind the last output row sy1 matches no row
                // both taps on one source row happens only where cv::resize clamps the rows (top and bottom edge): monotonic,
                // so the upper-row registers may simply be overwritten
                if (__builtin_expect(sy0 == r, 0)) {  // wave-uniform and rare: a branch, not two selects per row
                    hpA = hcA; hpB = hcB;
                    asm volatile("" : "+v"(hpA), "+v"(hpB));
                }
                unsigned oA, oB;
                if constexpr (AREA) {
                    oA = (hpA + hcA + 2u) >> 2; oB = (hpB + hcB + 2u) >> 2;
                } else {
                    // ((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2: each product's high half in one instruction
                    // (v_mul_hi_u32 with the weight as b << 16), one three-operand add, one shift - four instructions per
                    // output value where round 5 had six (24-bit multiply-add, shift, multiply, shift, add, shift)
                    oA = (umulhi_su(b0, hpA) + umulhi_su(b1, hcA) + 2u) >> 2;
                    oB = (umulhi_su(b0, hpB) + umulhi_su(b1, hcB) + 2u) >> 2;
                }
                unsigned offD = laneDst + dOff;
                asm volatile("" : "+v"(offD));
                if constexpr (!EDGE) {
                    gstore<unsigned short>(dstBase + offD, (unsigned short)(oA | (oB << 8)));
                } else {
                    if (store2) gstore<unsigned short>(dstBase + offD, (unsigned short)(oA | (oB << 8)));
                    else if (store1) gstore<uint8_t>(dstBase + offD, (uint8_t)oA);
                }
                dOff += (unsigned)D.pitch;
                jr++;
                rowTap(jr, sy0, sy1, b0, b1);  // (row nrows: the sentinel)
            }
            r++;
        };
        // whole trips of PR_PF: the steps behind the last source row see the last row again (the prefetch pointer stops
        // there) and write nothing, since every output row is done by then (sy1 = -1)
        for (int left = rLast - r + 1; left > 0; left -= PR_PF) {
#pragma unroll
            for (int k = 0; k < PR_PF; k++) {
                take(q[k]);
                step();
            }
        }
    };
    if (tx * PR_COLS + PR_COLS <= D.w) run(std::false_type{});  // wave-uniform
    else run(std::true_type{});
}

// ------------------------------------------------------------------------------------------------
// FAST-9/16 per cell.  One workgroup per (cell, image).  The cell's (wCell+6)x(hCell+6) uint8 tile is
// staged in LDS; scores (largest threshold at which the pixel is still a corner) go to an LDS score
// plane whose zero rim implements "a neighbour belonging to another cell counts as 0"; survivors are
// compacted in row-major order with ballot/popcount prefix sums (no atomics: the octree's result
// depends on candidate order).
// ------------------------------------------------------------------------------------------------
typedef unsigned short fs_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(fs_us2, a), __builtin_bit_cast(fs_us2, b)));
}
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(fs_us2, a), __builtin_bit_cast(fs_us2, b)));
}
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

__device__ __forceinline__ int fast_score(int v, const int p[16]) {
    // score = max over the 16 arcs of 9 of min(v - p_i) (either polarity) - 1  == cornerScore<16> for corners.
    // min over an arc of (v - p_i) = v - max over the arc of p_i, so the arc extrema are taken on the ring pixels
    // themselves and v enters once at the end: score = max(v - min_k max9_k(p), max_k min9_k(p) - v) - 1.
    // Arc extrema from runs of three: ext9[k] = ext3(ext3[k], ext3[k+3], ext3[k+6]) - v_min3 / v_max3 make every line
    // below one instruction per element (64 in all, plus 16 for the two final reductions).
    int mn3[16], mx3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mn3[k] = min(min(p[k], p[(k + 1) & 15]), p[(k + 2) & 15]);
        mx3[k] = max(max(p[k], p[(k + 1) & 15]), p[(k + 2) & 15]);
    }
    int mn9[16], mx9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mn9[k] = min(min(mn3[k], mn3[(k + 3) & 15]), mn3[(k + 6) & 15]);
        mx9[k] = max(max(mx3[k], mx3[(k + 3) & 15]), mx3[(k + 6) & 15]);
    }
    int darkest = max(max(mn9[0], mn9[1]), mn9[2]), brightest = min(min(mx9[0], mx9[1]), mx9[2]);
#pragma unroll
    for (int k = 3; k < 15; k += 2) {
        darkest = max(max(darkest, mn9[k]), mn9[k + 1]);
        brightest = min(min(brightest, mx9[k]), mx9[k + 1]);
    }
    darkest = max(darkest, mn9[15]);      // max over arcs of the arc's darkest pixel
    brightest = min(brightest, mx9[15]);  // min over arcs of the arc's brightest pixel
    return max(v - brightest, darkest - v) - 1;
}

// cornerScore of TWO candidates per lane: their ring pixels travel in the halves of a register, and the arc extrema are
// taken with v_pk_minimum3_f16 / v_pk_maximum3_f16 (gfx950) - three operands AND two halves per instruction, half the
// instructions of the v_min3_u32 / v_max3_u32 network above per candidate.  The bit patterns 0x0000 .. 0x00ff are f16
// denormals n * 2^-24, which order like the integers they are; tools/pkmin3_probe.hip checks on the hardware that neither
// instruction flushes or canonicalises them (all 2 x 16.7 M operand triples, both halves).  v = the two centre pixels.
__device__ __forceinline__ unsigned pk_min3_h(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned pk_max3_h(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
typedef short fs_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned fast_score2(unsigned v, const unsigned p[16]) {
    unsigned mn3[16], mx3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mn3[k] = pk_min3_h(p[k], p[(k + 1) & 15], p[(k + 2) & 15]);
        mx3[k] = pk_max3_h(p[k], p[(k + 1) & 15], p[(k + 2) & 15]);
    }
    unsigned mn9[16], mx9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mn9[k] = pk_min3_h(mn3[k], mn3[(k + 3) & 15], mn3[(k + 6) & 15]);
        mx9[k] = pk_max3_h(mx3[k], mx3[(k + 3) & 15], mx3[(k + 6) & 15]);
    }
    unsigned darkest = pk_max3_h(mn9[0], mn9[1], mn9[2]), brightest = pk_min3_h(mx9[0], mx9[1], mx9[2]);
#pragma unroll
    for (int k = 3; k < 15; k += 2) {
        darkest = pk_max3_h(darkest, mn9[k], mn9[k + 1]);
        brightest = pk_min3_h(brightest, mx9[k], mx9[k + 1]);
    }
    const fs_s2 dk = __builtin_bit_cast(fs_s2, pk_max3_h(darkest, mn9[15], mn9[15]));
    const fs_s2 br = __builtin_bit_cast(fs_s2, pk_min3_h(brightest, mx9[15], mx9[15]));
    const fs_s2 vv = __builtin_bit_cast(fs_s2, v);
    const fs_s2 sc = __builtin_elementwise_max(vv - br, dk - vv) - (fs_s2)(1);
    return __builtin_bit_cast(unsigned, sc);
}

// One WAVE per (cell, image): no workgroup barrier anywhere, so the ~20 cells resident on a CU hide each other's
// load and LDS latency (LDS per wave is kept near 6 KB); measured VALU bound (EXPERIMENTS.md section 3).
// TP = LDS pitch of the tile and of the score plane; TP > 0 makes every ring / neighbour offset an
// instruction immediate, TP == 0 is the any-size fallback.
// LDS carve (bytes): tile th*tp | score (ph+2)*tp | candidate ring FC_CAND u16 | corner list FC_CORN u16 ;
// the survivor flags reuse the tile once the scores are final.
#ifndef FC_CAND
#define FC_CAND 512   // candidates buffered between the rejection test and the score pass
#endif
#ifndef FC_XCD_RUN
#define FC_XCD_RUN 8
#endif
#ifndef FC_UNROLL
#define FC_UNROLL 2
#endif
#ifndef FC_ROWS
#define FC_ROWS 4     // rows per trip of the column-mapped rejection pass (>= 3)
#endif
#ifndef FC_NMS_REG
#define FC_NMS_REG 2  // corner-list chunks (64 corners each) whose NMS flags stay in registers
#endif
#ifndef FC_CORN
#define FC_CORN 384   // corners kept for NMS / emission; a cell with more falls back to scanning the plane (256: the cells of
                      // dense frames, ~330 corners, all took the scan: dense scenes +2.4 % with 384, the headline +0.7 %; 448: -1 %)
#endif
#ifndef FC_WAVES_PER_EU
#define FC_WAVES_PER_EU 1
#endif
__host__ __device__ __forceinline__ int fc_pitch(int wCell, int TP) { return TP ? TP : ((wCell + 12) & ~3); }
__host__ __device__ __forceinline__ int fc_tile_bytes(int wCell, int hCell, int TP) {
    // the flags of the slow path need one byte per tested pixel
    const int t = (hCell + 6) * fc_pitch(wCell, TP), f = wCell * hCell;
    return ((t > f ? t : f) + 15) & ~15;
}
__host__ __device__ __forceinline__ int fc_score_bytes(int wCell, int hCell, int TP) { return (((hCell + 2) * fc_pitch(wCell, TP)) + 15) & ~15; }
__host__ __device__ __forceinline__ int fc_list_bytes() { return 2 * (FC_CAND + FC_CORN) + 16; }  // + the spare entry

// dbg (FT_DEBUG_FAST, a timing probe - results are wrong with 1, 2, 4, 16 (16 drops the candidates: phase A alone); 8 only switches the paired score rounds off): 1 replaces the score network by a three-pixel hash, 2 stops in
// front of NMS / emission, 4 right behind the staging of the tile; tools/fast_probe.py times the kernel with each of them.
// ORDERED = false (device octree, which ranks candidates by their coordinates): the rejection pass is free to
// visit the pixels in any order and uses all 64 lanes (see below); ORDERED = true delivers every cell's
// candidates row-major, as the host octree expects them.
template <int TP, bool ORDERED>
__global__ __launch_bounds__(64, FC_WAVES_PER_EU) void k_fast_cells(FtGeom g, const uint8_t *const *l0, int l0pitch, const uint8_t *pyr,
                                                   int iniTh, int minTh, int alignedLoads, int *cellCount,
                                                   uint32_t *stage, const FtCellRec *cellTab, FtSlotGrid sg, int dbg, int tileBytes,
                                                   int scoreBytesMax, FtCellRanges cr) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x;
    int slot = blockIdx.y, cellSg = 0;
    if (sg.xcdMap && !ft_slot_block(sg, slot, cellSg)) return;
    const uint8_t *img0 = l0[slot];  // level 0 is the caller's frame; requested before anything depends on the level
    // XCD-aware mapping: workgroup b runs on XCD b % 8 (observed placement, used for speed only).  Launches of eight
    // or more images put all cells of an image on one XCD (ft_slot_grid, above): neighbouring cells - their tiles
    // overlap by 6 px in both directions and share 64-B lines - hit the same private L2, which also still holds what
    // the pyramid kernel wrote for that image.  Smaller launches deal runs of FC_XCD_RUN consecutive cells round-robin
    // to the XCDs, so that at least horizontal neighbours share an L2 while all XCDs stay busy on the few images.
    const int j = (int)(blockIdx.x >> 3), xcd = (int)(blockIdx.x & 7);
    // the launch's cells: one or two runs of consecutive cells of the image's cell table (ft_launch_fast_cells: the levels whose
    // tiles fit this variant's pitch)
    const int cj = sg.xcdMap ? cellSg : ((j / FC_XCD_RUN) * 8 + xcd) * FC_XCD_RUN + (j % FC_XCD_RUN);
    if (cj >= cr.n0 + cr.n1) return;
    const int cell = cj < cr.n0 ? cr.lo0 + cj : cr.lo1 + (cj - cr.n0);
    // the cell's record, built with the extractor: origin, tile size, level, source and staging offsets in one scalar load
    const FtCellRec rec = cellTab[cell];
    const int iniX = (int)(rec.origin & 0xffffu), iniY = (int)(rec.origin >> 16);
    const int tw = (int)(rec.shape & 0xffu), th = (int)((rec.shape >> 8) & 0xffu), level = (int)((rec.shape >> 16) & 0xffu);
    int *cnt = cellCount + (size_t)slot * g.totalCells + cell;
    // ORBextractor.cc:1141,1150 skip rules; cv::FAST finds nothing in a sub-image under 7 px (the record says so)
    if (tw == 0) {
        if (lane == 0) *cnt = 0;
        return;
    }
    const unsigned lv = g.fastLv[level];
    const int cellCap = (int)(lv >> 16);
    const int pw = tw - 6, ph = th - 6;  // tested region
    const int npx = pw * ph;
    // the any-size variant derives its LDS pitch from the level's cell width; the fixed-pitch variants carve the LDS by the
    // launch-wide maxima (the launcher sizes the allocation by them anyway)
    const int tp = TP ? TP : fc_pitch(g.lv[level].wCell, 0);
    // division by the row length: only the any-size variant unpacks pixel codes with it (the fixed-pitch variants
    // need it on the rare score-plane scan alone and compute it there)
    const unsigned pwMagic = TP ? 0u : div_magic_of((unsigned)pw);
    uint8_t *tile = smem;
    uint8_t *score = tile + (TP ? tileBytes : fc_tile_bytes(g.lv[level].wCell, g.lv[level].hCell, 0));
    const int scoreBytes = TP ? ((ph + 2) * TP + 15) & ~15 : fc_score_bytes(g.lv[level].wCell, g.lv[level].hCell, 0);
    unsigned short *cand = (unsigned short *)(score + (TP ? scoreBytesMax : scoreBytes));
    unsigned short *corn = cand + FC_CAND;
    const int pitch = level ? (int)(lv & 0xffffu) : l0pitch;
    // pixel (iniX, iniY) of the level
    const uint8_t *org = level ? pyr + ((size_t)slot * g.pyrPerSlot + rec.srcOff) : img0 + ((size_t)iniY * l0pitch + iniX);
    // ---- stage the tile: aligned dword rows when the level allows it (coalesced 4-byte lanes) ----
    int ax = 0;
    if (alignedLoads) {
        ax = iniX & 3;
        const int nd = (tw + ax + 3) >> 2;
        const uint8_t *src = org - ax;
        if constexpr (TP != 0) {
            // fixed (row, dword) per lane: TP/4 dwords span a tile row, 64 / (TP/4) rows per trip.  Rows and dwords beyond
            // the tile are clamped to its last row / dword instead of being masked: such a lane loads and stores the
            // same element as the lane that owns it, so the whole copy is branch-free - no per-load division, no 64-bit
            // arithmetic, no exec-mask bookkeeping.  Nine trips are in flight before the first LDS store.
            constexpr int DW = TP / 4, RPT = 64 / DW;
            const int rr = lane / DW, cc4 = 4 * min(lane - rr * DW, nd - 1);
            if (th > 8 * RPT && th <= 9 * RPT) {  // wave-uniform
                // the usual cell (41 - 45 tile rows at TP 48): the first eight row groups exist for every lane, so their
                // rows need no clamp - a load is one address add, its LDS store an immediate offset; only the ninth is clamped
                const unsigned goff = (unsigned)vmad24(rr, pitch, cc4);
                unsigned *tl = (unsigned *)(tile + vmad24(rr, TP, cc4));
                const int row8 = min(8 * RPT + rr, th - 1);
                unsigned v[9];
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = gload<unsigned>(src + (goff + (unsigned)(k * RPT * pitch)));
                v[8] = gload<unsigned>(src + (unsigned)vmad24(row8, pitch, cc4));
#pragma unroll
                for (int k = 0; k < 8; k++) tl[k * RPT * (TP / 4)] = v[k];
                *(unsigned *)(tile + vmad24(row8, TP, cc4)) = v[8];
            } else
            for (int r0 = 0; r0 < th; r0 += 9 * RPT) {
                unsigned v[9];
                int row[9];
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    row[k] = min(r0 + k * RPT + rr, th - 1);
                    v[k] = gload<unsigned>(src + (unsigned)vmad24(row[k], pitch, cc4));
                }
#pragma unroll
                for (int k = 0; k < 9; k++) *(unsigned *)(tile + vmad24(row[k], TP, cc4)) = v[k];
            }
        } else {
            const unsigned ndMagic = div_magic_of((unsigned)nd);
            // every load of the tile is issued before the first LDS store
            for (int i0 = 0; i0 < nd * th; i0 += 64 * 9) {
                unsigned v[9];
                int off[9];
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    const int i = i0 + 64 * k + lane, ic = min(i, nd * th - 1);
                    const int y = div_by(ic, ndMagic), x = ic - y * nd;
                    off[k] = i < nd * th ? y * tp + 4 * x : -1;
                    v[k] = gload<unsigned>(src + (size_t)y * pitch + 4 * x);
                }
#pragma unroll
                for (int k = 0; k < 9; k++)
                    if (off[k] >= 0) *(unsigned *)(tile + off[k]) = v[k];
            }
        }
    } else {
        const unsigned twMagic = div_magic_of((unsigned)tw);
        const uint8_t *src = org;
        for (int i = lane; i < tw * th; i += 64) {
            const int y = div_by(i, twMagic), x = i - y * tw;
            tile[y * tp + x] = gload<uint8_t>(src + (size_t)y * pitch + x);
        }
    }
    // the whole score plane (a multiple of 16 bytes, 16-byte aligned) is cleared with 16-byte stores: two per lane
    for (int i = lane; i < scoreBytes >> 4; i += 64) ((uint4 *)score)[i] = make_uint4(0, 0, 0, 0);
    wave_lds_sync();
    if (dbg & 4) {
        if (lane == 0) *cnt = (int)tile[lane] & 0;
        return;
    }
    const uint8_t *t0 = tile + ax;  // pixel (x, y) of the cell sub-image at t0[y * tp + x]
    // ---- phase A / B rounds.  A: high-speed rejection (OpenCV's opposite-pair test without the
    // polarity): a 9-arc contains one pixel of every opposite pair, so for the two compass pairs (k = 0, 4)
    // max(|d_k|, |d_k+8|) must exceed the threshold; survivors are appended in row-major order to the
    // candidate ring (the any-size variant also tests the two diagonal pairs).  B (whenever the ring fills, and at the end): score = largest threshold at which
    // the pixel is still a corner (cornerScore<16>); corner at minThFAST <=> score >= minThFAST.
    int nc = 0, ncorn = 0;
    // A pixel of the tested region travels through the candidate ring and the corner list as a 16-bit code: with a
    // fixed pitch (pw < 64) that is (y << 6 | x), packed and unpacked with a shift and a mask; the any-size variant
    // keeps the linear index y * pw + x and pays a division per unpack.
    auto pixCode = [&](int y, int x) -> int { return TP > 0 ? ((y << 6) | x) : y * pw + x; };
    auto pixY = [&](int code) -> int { return TP > 0 ? code >> 6 : div_by(code, pwMagic); };
    auto pixX = [&](int code, int y) -> int { return TP > 0 ? (code & 63) : code - y * pw; };
    // phase B over the buffered candidates: score, corner list (row-major), ring reset
    auto flushB = [&]() {
        if (dbg & 16) {  // probe: phase A alone (the candidates are dropped)
            nc = 0;
            return;
        }
        wave_lds_sync();
        // Branch-free rounds: a lane beyond the last candidate repeats the last one (same score, same store) and is kept out
        // of the corner list by its flag; a candidate that is no corner stores 0 over the 0 the plane already holds; a lane
        // with nothing for the corner list writes the spare entry behind the lists.
        int jb = 0;
        if constexpr (TP > 0) {
            // rounds of 128: two candidates per lane through the packed network (fast_score2); the corner list takes the
            // first 64 of a round before the second 64, so its order is the ring's
            for (; nc - jb > 64 && !(dbg & 9); jb += 128) {  // dbg 8: A/B switch, rounds of 64 only
                const int jA = jb + lane, jB = jb + 64 + lane;
                const int cA = cand[jA], cB = cand[min(jB, nc - 1)];
                const int yA = pixY(cA), xA = pixX(cA, yA), yB = pixY(cB), xB = pixX(cB, yB);
                const uint8_t *wA = t0 + yA * tp + xA, *wB = t0 + yB * tp + xB;
                unsigned v2;
                unsigned ring2[16];
                // (Tried, round 3: the 7 x 7 window as seven unaligned ds_read_b64 / _b32 per candidate + v_perm_b32 - 14 LDS
                // instructions per round instead of 34 - is correct and TWICE as slow, 0.63 against 0.34 ms per launch of 128
                // frames: unaligned LDS accesses are replayed.  ds_read_u8 + ds_read_u8_d16_hi into one register, to save the 17
                // packing instructions: gfx950 runs with SRAM ECC, where a d16 load writes the whole register - wrong results.)
                v2 = (unsigned)wA[3 * tp + 3] | ((unsigned)wB[3 * tp + 3] << 16);
#define FT_LD2(k, ox, oy) ring2[k] = (unsigned)wA[((oy) + 3) * tp + (ox) + 3] | ((unsigned)wB[((oy) + 3) * tp + (ox) + 3] << 16);
                FT_RING(FT_LD2)
#undef FT_LD2
                const unsigned sc2 = fast_score2(v2, ring2);
                const int scA = (int)(short)(sc2 & 0xffffu), scB = (int)(short)(sc2 >> 16);
                const bool cornA = scA >= minTh, cornB = scB >= minTh;
                score[(yA + 1) * tp + (xA + 1)] = (uint8_t)(cornA ? scA : 0);
                score[(yB + 1) * tp + (xB + 1)] = (uint8_t)(cornB ? scB : 0);
                const bool isB = cornB && jB < nc;
                const unsigned long long ba = __builtin_amdgcn_ballot_w64(cornA), bb = __builtin_amdgcn_ballot_w64(isB);
                const int posA = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ba >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ba, (unsigned)ncorn));
                ncorn += __popcll(ba);
                const int posB = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bb, (unsigned)ncorn));
                ncorn += __popcll(bb);
                corn[cornA && posA < FC_CORN ? posA : FC_CORN] = (unsigned short)cA;  // corn + FC_CORN = the spare entry
                corn[isB && posB < FC_CORN ? posB : FC_CORN] = (unsigned short)cB;
            }
        }
        for (; jb < nc; jb += 64) {
            const int j = jb + lane;
            const int ci2 = cand[min(j, nc - 1)];
            const int y = pixY(ci2), x = pixX(ci2, y);
            // top-left corner of the pixel's 7 x 7 ring window: with a fixed pitch every ring offset is a non-negative
            // immediate of the LDS read (a negative one costs an address add of its own)
            const uint8_t *win = t0 + y * tp + x;
            const int v = win[3 * tp + 3];
            int ringPx[16];
#define FT_LD(k, ox, oy) ringPx[k] = (int)win[((oy) + 3) * tp + (ox) + 3];
            FT_RING(FT_LD)
#undef FT_LD
            const int sc = (dbg & 1) ? ((v + ringPx[0] + ringPx[5] + ringPx[11]) & 31) : fast_score(v, ringPx);
            const bool corner = sc >= minTh;
            score[(y + 1) * tp + (x + 1)] = (uint8_t)(corner ? sc : 0);
            const bool isCorner = corner && j < nc;
            const unsigned long long cb = __builtin_amdgcn_ballot_w64(isCorner);
            const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(cb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cb, (unsigned)ncorn));
            corn[isCorner && pos < FC_CORN ? pos : FC_CORN] = (unsigned short)ci2;  // corn + FC_CORN = the spare entry
            ncorn += __popcll(cb);
        }
        nc = 0;
        wave_lds_sync();
    };
#define FT_AD(ox, oy) __builtin_amdgcn_sad_u8(v, (unsigned)cpx[(oy)*tp + (ox)], 0u)
    // Appends the lanes of a row whose compass pairs pass to the candidate ring, in lane order, without a branch: the
    // other lanes store to the spare entry behind the lists.  (The two diagonal pairs used to be tested first; on
    // textured images that cost more instructions per row than the few candidates it removed cost in phase B, where
    // a partly filled round of 64 is as expensive as a full one.)
    auto pushRow = [&](bool pass, int code, int n) -> int {
        const unsigned long long b = __ballot(pass);
        const int pos = pass ? (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, (unsigned)n))
                             : FC_CAND + FC_CORN;
        cand[pos] = (unsigned short)code;
        return n + __popcll(b);
    };
    if constexpr (TP > 0 && !ORDERED) {
        // Two half-cells side by side: lanes 0-31 walk the rows of the upper half, lanes 32-63 the rows of the lower
        // half, 32 columns each (a 36-pixel-wide cell fills 56 % of a 64-lane row, two 32-wide halves fill it); the
        // columns beyond 32 are tested afterwards in linear order.  Inside a half the rows slide as in the ordered
        // pass (shared vertical differences, immediates only).
        const int half = lane >> 5, cx = lane & 31;
        const int hRows = (ph + 1) >> 1;          // rows of the upper half; the lower half has ph - hRows
        const int y0h = half * hRows;              // first row of this lane's half
        const int wMain = min(pw, 32);
        const bool actC = cx < wMain;
        const uint8_t *col = t0 + min(cx, wMain - 1) + 3 + y0h * TP;  // (x + 3, tile row y0h)
        const int codeBase = (y0h << 6) | cx;
        unsigned pA = col[3 * TP], pB = col[4 * TP], pC = col[5 * TP];
        // Rows travel in PAIRS (y, y + 1) packed in the halves of a register: v_sad_u8 / v_sad_hi_u8 produce the halves,
        // v_pk_max_u16 / v_pk_min_u16 combine them.  The vertical difference |p(y) - p(y + 3)| is the south difference of
        // row y and the north difference of row y + 3, so a north pair is an alignbit of two earlier south pairs.
        // History: hist0 = [., dS(y - 3)], hist1 = [dS(y - 2), dS(y - 1)].
        unsigned hist0 = __builtin_amdgcn_sad_hi_u8(pA, (unsigned)col[0], 0u);
        unsigned hist1 = __builtin_amdgcn_sad_hi_u8(pC, (unsigned)col[2 * TP], __builtin_amdgcn_sad_u8(pB, (unsigned)col[TP], 0u));
        const unsigned th = (unsigned)minTh;
        const int rowsHalf = half ? ph - hRows : hRows;
        // blocks of at most 28 rows (seven trips): a lane's verdicts of a block fit one 32-bit mask
        for (int yb = 0; yb < hRows; yb += 28) {
        const int yEnd = min(yb + 28, hRows);
        unsigned cmask = 0;
        auto trip = [&](int y) {
            const uint8_t *r = col + y * TP;
            unsigned pv[7];
            pv[0] = pA; pv[1] = pB; pv[2] = pC;
#pragma unroll
            for (int k = 0; k < 4; k++) pv[k + 3] = r[(k + 6) * TP];
            const unsigned ps0 = __builtin_amdgcn_sad_hi_u8(pv[1], pv[4], __builtin_amdgcn_sad_u8(pv[0], pv[3], 0u));
            const unsigned ps1 = __builtin_amdgcn_sad_hi_u8(pv[3], pv[6], __builtin_amdgcn_sad_u8(pv[2], pv[5], 0u));
            const unsigned pn0 = __builtin_amdgcn_alignbit(hist1, hist0, 16);  // [dS(y - 3), dS(y - 2)]
            const unsigned pn1 = __builtin_amdgcn_alignbit(ps0, hist1, 16);    // [dS(y - 1), dS(y)]
            const unsigned pe0 = __builtin_amdgcn_sad_hi_u8(pv[1], (unsigned)r[4 * TP + 3], __builtin_amdgcn_sad_u8(pv[0], (unsigned)r[3 * TP + 3], 0u));
            const unsigned pw0 = __builtin_amdgcn_sad_hi_u8(pv[1], (unsigned)r[4 * TP - 3], __builtin_amdgcn_sad_u8(pv[0], (unsigned)r[3 * TP - 3], 0u));
            const unsigned pe1 = __builtin_amdgcn_sad_hi_u8(pv[3], (unsigned)r[6 * TP + 3], __builtin_amdgcn_sad_u8(pv[2], (unsigned)r[5 * TP + 3], 0u));
            const unsigned pw1 = __builtin_amdgcn_sad_hi_u8(pv[3], (unsigned)r[6 * TP - 3], __builtin_amdgcn_sad_u8(pv[2], (unsigned)r[5 * TP - 3], 0u));
            const unsigned M0 = pk_min_u16(pk_max_u16(pn0, ps0), pk_max_u16(pe0, pw0));
            const unsigned M1 = pk_min_u16(pk_max_u16(pn1, ps1), pk_max_u16(pe1, pw1));
            hist0 = ps0; hist1 = ps1;
            pA = pv[4]; pB = pv[5]; pC = pv[6];
            // Every lane files the verdicts of its column in a bit mask of its own: the compare of a half of M against the
            // threshold lands in vcc and v_addc shifts it into the mask (mask = 2 * mask + vcc) - two vector instructions per row
            // and no scalar work at all, where a ballot per row with its popcount, position arithmetic and exec juggling cost
            // a dozen scalar instructions for every row that holds a candidate.
            asm("v_cmp_gt_u32_sdwa vcc, %1, %2 src0_sel:WORD_0 src1_sel:DWORD\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                "v_cmp_gt_u32_sdwa vcc, %1, %2 src0_sel:WORD_1 src1_sel:DWORD\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                "v_cmp_gt_u32_sdwa vcc, %3, %2 src0_sel:WORD_0 src1_sel:DWORD\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                "v_cmp_gt_u32_sdwa vcc, %3, %2 src0_sel:WORD_1 src1_sel:DWORD\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                : "+v"(cmask)
                : "v"(M0), "v"(th), "v"(M1)
                : "vcc");
        };
        // two trips per iteration: the register rotation (pA / pB / pC, the two history pairs) between them is renaming
        // instead of four moves per trip
        int y = yb;
        for (; y + 4 < yEnd; y += 8) {
            trip(y);
            trip(y + 4);
        }
        if (y < yEnd) trip(y);
        {
            // row yb + r of a half sits at bit T - 1 - r (T = rows walked in the block, a multiple of four); rows the half does
            // not have and columns beyond the cell are masked out here, once
            const int T = (yEnd - yb + 3) & ~3;
            const int cnt = min(max(rowsHalf - yb, 0), T);
            const unsigned valid = actC ? (((1u << cnt) - 1u) << (T - cnt)) : 0u;
            cmask &= valid;
            const int codeTop = codeBase + ((yb + T - 1) << 6);
            // Emission.  The lanes' candidate counts are scanned once (six DPP adds), which gives every lane the ring slot of
            // its first candidate; then every lane writes its own candidates one after the other, lowest bit first - as many
            // rounds as the fullest column holds candidates, each round five vector instructions, a write and the loop test
            // (a ballot + popcount + position per round, as the rows used to cost, is what this avoids).
            int incl = __popc(cmask);
            const int own = incl;
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);  // row_shr:1
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);  // row_shr:2
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);  // row_shr:4
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);  // row_shr:8
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, true);  // row_bcast:15 into rows 1 and 3
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, true);  // row_bcast:31 into rows 2 and 3
            const int total = __builtin_amdgcn_readlane(incl, 63);
            // (every step leaves room for one more round of 64 in the ring: the appends of the remaining columns rely on it)
            if (nc + total > FC_CAND - 64) flushB();  // wave-uniform; leaves nc = 0
            if (total <= FC_CAND - 64) {
                unsigned short *slot = cand + (nc + incl - own);
                while (__builtin_amdgcn_ballot_w64(cmask != 0u)) {  // wave-uniform
                    // a lane whose mask is empty writes the spare entry behind the lists: no exec juggling in the loop
                    *(cmask ? slot : cand + (FC_CAND + FC_CORN)) = (unsigned short)(codeTop - (__builtin_ctz(cmask | 0x80000000u) << 6));
                    cmask &= cmask - 1u;
                    slot++;
                }
                nc += total;
            } else {
                // more candidates in one block than the ring holds (dense texture): a round per candidate rank with a flush
                // whenever the ring fills
                for (;;) {
                    const unsigned long long b = __builtin_amdgcn_ballot_w64(cmask != 0u);
                    if (!b) break;  // wave-uniform
                    const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, (unsigned)nc));
                    const int bit = __builtin_ctz(cmask | 0x80000000u);
                    if (__builtin_amdgcn_inverse_ballot_w64(b)) cand[pos] = (unsigned short)(codeTop - (bit << 6));
                    cmask &= cmask - 1u;
                    nc += __popcll(b);
                    if (nc > FC_CAND - 64) flushB();  // wave-uniform
                }
            }
        }
        }
        // columns 32 .. pw-1 (four of them for a 36-pixel cell), every row, in linear order
        const int wRem = pw - wMain, nRem = wRem * ph;
        if (wRem > 0) {
            const unsigned remMagic = div_magic_small((unsigned)wRem);
            for (int base = 0; base < nRem; base += 64) {
                const int i = base + lane;
                const int ii = min(i, nRem - 1);
                const int yy = div_by(ii, remMagic), xx = wMain + ii - yy * wRem;
                const uint8_t *cpx = t0 + (yy + 3) * TP + (xx + 3);
                const unsigned v = cpx[0];
                const unsigned m0 = max(FT_AD(0, 3), FT_AD(0, -3));
                const unsigned m1 = max(FT_AD(3, 0), FT_AD(-3, 0));
                const unsigned m01r = i < nRem ? min(m0, m1) : 0u;
                if (__any(m01r > (unsigned)minTh)) nc = pushRow(m01r > (unsigned)minTh, pixCode(yy, xx), nc);
                if (nc > FC_CAND - 64) flushB();  // wave-uniform
            }
        }
        flushB();
    } else if constexpr (TP > 0) {
        // Fixed pitch (pw <= TP - 6 < 64): lane = column, rows walked top to bottom three at a time.  Every tile
        // offset is an immediate; the vertical differences |p(y) - p(y+3)| serve row y (south) and row y + 3
        // (north), and the centre of row y + 3 is the south pixel of row y, so a row costs three LDS reads and
        // seven VALU instructions per lane before the wave-level early out.
        const bool act = lane < pw;
        const uint8_t *col = t0 + min(lane, pw - 1) + 3;  // (x + 3, tile row 0)
        unsigned pA = col[3 * TP], pB = col[4 * TP], pC = col[5 * TP];
        unsigned dA = __builtin_amdgcn_sad_u8(pA, (unsigned)col[0], 0u);
        unsigned dB = __builtin_amdgcn_sad_u8(pB, (unsigned)col[TP], 0u);
        unsigned dC = __builtin_amdgcn_sad_u8(pC, (unsigned)col[2 * TP], 0u);
        // FC_ROWS rows per trip: all their LDS reads are issued before the first use, one wave-level early out
        // covers the trip
        for (int y = 0; y < ph; y += FC_ROWS) {
            const uint8_t *r = col + y * TP;
            unsigned pv[FC_ROWS + 3], m01[FC_ROWS], dSv[FC_ROWS], mAny = 0;
            pv[0] = pA; pv[1] = pB; pv[2] = pC;
#pragma unroll
            for (int k = 0; k < FC_ROWS; k++) pv[k + 3] = r[(k + 6) * TP];
#pragma unroll
            for (int k = 0; k < FC_ROWS; k++) {
                const unsigned v = pv[k];
                const uint8_t *cpx = r + (k + 3) * TP;
                const unsigned dS = dSv[k] = __builtin_amdgcn_sad_u8(v, pv[k + 3], 0u);
                const unsigned dN = k == 0 ? dA : k == 1 ? dB : k == 2 ? dC : dSv[k >= 3 ? k - 3 : 0];
                const unsigned m1 = max(FT_AD(3, 0), FT_AD(-3, 0));
                m01[k] = (act && y + k < ph) ? min(max(dN, dS), m1) : 0u;
                mAny = max(mAny, m01[k]);
            }
            dA = dSv[FC_ROWS - 3]; dB = dSv[FC_ROWS - 2]; dC = dSv[FC_ROWS - 1];
            pA = pv[FC_ROWS]; pB = pv[FC_ROWS + 1]; pC = pv[FC_ROWS + 2];
            if (__any(mAny > (unsigned)minTh)) {
#pragma unroll
                for (int k = 0; k < FC_ROWS; k++) nc = pushRow(m01[k] > (unsigned)minTh, pixCode(y + k, lane), nc);
            }
            if (nc > FC_CAND - FC_ROWS * 64 || y + FC_ROWS >= ph) flushB();  // wave-uniform
        }
    } else {
        // any-size fallback: pixels in linear order, FC_UNROLL 64-pixel chunks per trip
        for (int base = 0; base < npx; base += 64 * FC_UNROLL) {
            unsigned m01[FC_UNROLL];
            const uint8_t *cp[FC_UNROLL];
            unsigned mAny = 0;
#pragma unroll
            for (int u = 0; u < FC_UNROLL; u++) {
                const int i = base + 64 * u + lane;
                const int ii = min(i, npx - 1);
                const int y = div_by(ii, pwMagic), x = ii - y * pw;
                const uint8_t *cpx = t0 + (y + 3) * tp + (x + 3);
                cp[u] = cpx;
                const unsigned v = cpx[0];
                const unsigned m0 = max(FT_AD(0, 3), FT_AD(0, -3));
                const unsigned m1 = max(FT_AD(3, 0), FT_AD(-3, 0));
                m01[u] = i < npx ? min(m0, m1) : 0u;
                mAny = max(mAny, m01[u]);
            }
            // wave-level early out: when the compass pairs already reject every pixel of the trip (flat regions)
            // the two diagonal pairs are not even read
            if (__any(mAny > (unsigned)minTh)) {
#pragma unroll
                for (int u = 0; u < FC_UNROLL; u++) {
                    const int i = base + 64 * u + lane;
                    const uint8_t *cpx = cp[u];
                    const unsigned v = cpx[0];
                    const unsigned m2 = max(FT_AD(2, 2), FT_AD(-2, -2));
                    const unsigned m3 = max(FT_AD(2, -2), FT_AD(-2, 2));
                    const bool pass = min(m01[u], min(m2, m3)) > (unsigned)minTh;
                    const unsigned long long b = __ballot(pass);
                    if (pass) cand[nc + __popcll(b & ((1ull << lane) - 1ull))] = (unsigned short)i;
                    nc += __popcll(b);
                }
            }
            if (nc > FC_CAND - 64 * FC_UNROLL || base + 64 * FC_UNROLL >= npx) flushB();  // wave-uniform
        }
    }
#undef FT_AD
    wave_lds_sync();
    // ---- NMS (strictly greater than the 8 neighbours, cv::FAST), cell-level threshold fallback
    // (ORBextractor.cc:1157-1177: if any survivor reaches iniThFAST only those are emitted, otherwise every
    // minThFAST survivor is) and emission.  The corner list is in row-major order in the ORDERED variant; a cell with more
    // than FC_CORN corners scans the score plane instead (same order).
    if (dbg & 2) {
        if (lane == 0) *cnt = 0;
        return;
    }
    const bool useList = ncorn <= FC_CORN;
    const int nItems = useList ? ncorn : npx;
    auto nms = [&](int item, int &pix) -> int {
        int y, x;
        if (useList) {
            pix = (int)corn[item];
            y = pixY(pix), x = pixX(pix, y);
        } else {  // score-plane scan: item is the linear index of the pixel
            const unsigned mg = TP ? div_magic_of((unsigned)pw) : pwMagic;
            y = div_by(item, mg), x = item - y * pw;
            pix = pixCode(y, x);
        }
        // top-left neighbour of the pixel: every offset below is a non-negative immediate with a fixed pitch
        const uint8_t *s = score + y * tp + x;
        const unsigned v = s[tp + 1];
        // strictly greater than all eight neighbours <=> greater than their maximum (three v_max3 and a v_max, no
        // short-circuit branches); the maximum is >= 0, so a kept pixel has a score
        const unsigned m = max(max(max(max((unsigned)s[0], (unsigned)s[1]), (unsigned)s[2]),
                                   max(max((unsigned)s[tp], (unsigned)s[tp + 2]), (unsigned)s[2 * tp])),
                               max((unsigned)s[2 * tp + 1], (unsigned)s[2 * tp + 2]));
        return v > m ? (v >= (unsigned)iniTh ? 2 : 1) : 0;
    };
    uint32_t *out = stage + ((size_t)slot * g.stagePerSlot + rec.outOff);
    int run = 0;
    auto emit = [&](int fl, int pix, int need) {
        const bool f = fl >= need;
        const unsigned long long b = __builtin_amdgcn_ballot_w64(f);
        const int y = pixY(pix), x = pixX(pix, y);
        const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, (unsigned)run));
        // keypoint (x+3, y+3) in the cell sub-image, shifted by (j*wCell, i*hCell): ORBextractor.cc:1196-1197
        const uint32_t packed = ft_pack_cand(x + (iniX - 13), y + (iniY - 13), score[(y + 1) * tp + (x + 1)]);
        if (f && pos < cellCap) out[pos] = packed;
        run += __popcll(b);
    };
    if (nItems <= 64 * FC_NMS_REG) {
        // the usual case: the NMS flags of the whole cell stay in registers, the corner list is read once
        int fl[FC_NMS_REG], px[FC_NMS_REG], anyHi = 0;
#pragma unroll
        for (int q = 0; q < FC_NMS_REG; q++) {
            fl[q] = 0;
            px[q] = 0;
            if (q * 64 < nItems) {  // wave-uniform
                // lanes beyond the list look at its last entry and drop the verdict (no divergent branch)
                const int f = nms(min(q * 64 + lane, nItems - 1), px[q]);
                fl[q] = q * 64 + lane < nItems ? f : 0;
                anyHi |= __any(fl[q] == 2);
            }
        }
        const int need = anyHi ? 2 : 1;
#pragma unroll
        for (int q = 0; q < FC_NMS_REG; q++)
            if (q * 64 < nItems) emit(fl[q], px[q], need);
    } else {
        // More corners than the registers hold flags for (dense frames).  The threshold decision needs to know whether ANY
        // survivor reaches iniThFAST before the first one is emitted; round 3 computed every verdict, parked it in LDS and walked
        // the list a second time.  Now: a cell whose best corner is below iniThFAST cannot have such a survivor, and one whose
        // best corner reaches it nearly always has (only ties between neighbouring maxima can take them all out) - so the
        // decision is taken from the maximum score (one read per corner), the verdicts are computed and emitted in ONE walk,
        // and the rare cell where the strong corners all fell to ties is walked again for the weak ones.
        auto itemPix = [&](int it) -> int {
            if (useList) return (int)corn[it];
            const unsigned mg = TP ? div_magic_of((unsigned)pw) : pwMagic;
            const int y = div_by(it, mg);
            return pixCode(y, it - y * pw);
        };
        unsigned mx = 0;
        for (int base = 0; base < nItems; base += 64) {
            const int pix = itemPix(min(base + lane, nItems - 1));
            const int y = pixY(pix), x = pixX(pix, y);
            mx = max(mx, (unsigned)score[(y + 1) * tp + (x + 1)]);
        }
        int need = __any(mx >= (unsigned)iniTh) ? 2 : 1;
        for (;;) {
            int anyHi = 0;
            run = 0;
            for (int base = 0; base < nItems; base += 64) {
                const int it = base + lane;
                int pix = 0;
                const int f = nms(min(it, nItems - 1), pix);
                const int fl = it < nItems ? f : 0;
                anyHi |= __any(fl == 2);
                emit(fl, pix, need);
            }
            if (need == 1 || anyHi) break;  // wave-uniform
            need = 1;                       // every strong corner lost a tie: the cell emits its weak survivors
        }
    }
    if (lane == 0) *cnt = min(run, cellCap);
}

// ------------------------------------------------------------------------------------------------
// Compaction in cell order: one workgroup per (level, image).  Exclusive scan of the cell counts in cell
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
#define OD_PP 48                // raw pitch: 12 dwords cover the 43 bytes at any 4-byte phase
#define OD_HP 40                // pitch (u16) of the horizontally blurred rows: raw byte positions 0..39
#ifndef OD_LAYOUT
#define OD_LAYOUT 1             // 0: horizontally blurred rows row-major (seven scattered u16 reads per sample); 1: column-major
#endif                          // (the seven values of a sample are consecutive: two ds_read2_b32 + three v_alignbit)
#ifndef OD_RP
#define OD_RP 44                // column-major layout: u16 slots per column (43 rows + 1: dword-aligned columns)
#endif
#define OD_RAW_BYTES (OD_P * OD_PP)                  // 2064
#define OD_HB_BYTES (OD_LAYOUT ? OD_HP * OD_RP * 2 + 16 : OD_P * OD_HP * 2)  // 3536 / 3440
#define OD_WAVE_BYTES ((OD_RAW_BYTES + OD_HB_BYTES + 15) & ~15)
#define OD_WAVES 4
#ifndef OD_MFMA
#define OD_MFMA 0               // 1: horizontal 7-tap pass on the matrix pipe (v_mfma_i32_16x16x32_i8): patch rows x banded tap matrix,
#endif                          // pixels kept in LDS as p - 128, rows kept as h - 32768.  Exact (the 47 extraction tests), 10 % fewer
                                // vector instructions, and 0.5 % SLOWER in the pipeline: a matrix product holds the SIMD's vector
                                // issue for its 16 cycles (tools/mfma_overlap.hip), and the banded form uses 16 % of its
                                // multiply-adds (EXPERIMENTS 11.11).  Kept as a build option for the A/B (tools/od_ab.sh)
#if OD_MFMA && !OD_LAYOUT
#error "OD_MFMA writes the column-major layout"
#endif
#define OD_BIAS (OD_MFMA ? 0x80808080u : 0u)
#ifndef OD_KPW_WIDE
#define OD_KPW_WIDE 2           // keypoints per wave (1 or 2) in launches of 8+ images: loads of all of them in flight before the first is processed
#endif

__device__ __forceinline__ int reflect101(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// The 7x7 sigma-2 Gaussian of cv::GaussianBlur (8-bit fixed point, taps {18,34,48,56,48,34,18}) as the
// two routines k_orient_desc applies on the fly; k_blur_level (a test tap) runs the same two routines over a whole level.
// Horizontal: four consecutive outputs h[j] = sum_t k[t] * byte[j + t] (j = 0..3, t = 0..6) from three aligned dwords
// holding bytes 0..11.  The shifted windows of outputs 1 .. 3 are expressed through shifted WEIGHTS on the aligned
// dwords (ten v_dot4 with constant weight vectors) instead of byte-aligning the data first.
__device__ __forceinline__ void od_hblur4(unsigned d0, unsigned d1, unsigned d2, unsigned &h0, unsigned &h1, unsigned &h2, unsigned &h3) {
#define FT_W4(a, b, c, d) ((unsigned)(a) | ((unsigned)(b) << 8) | ((unsigned)(c) << 16) | ((unsigned)(d) << 24))
    const unsigned W0a = FT_W4(18, 34, 48, 56), W0b = FT_W4(48, 34, 18, 0);
    const unsigned W1a = FT_W4(0, 18, 34, 48), W1b = FT_W4(56, 48, 34, 18);
    const unsigned W2a = FT_W4(0, 0, 18, 34), W2b = FT_W4(48, 56, 48, 34), W2c = FT_W4(18, 0, 0, 0);
    const unsigned W3a = FT_W4(0, 0, 0, 18), W3b = FT_W4(34, 48, 56, 48), W3c = FT_W4(34, 18, 0, 0);
#undef FT_W4
    h0 = __builtin_amdgcn_udot4(d1, W0b, __builtin_amdgcn_udot4(d0, W0a, 0u, false), false);
    h1 = __builtin_amdgcn_udot4(d1, W1b, __builtin_amdgcn_udot4(d0, W1a, 0u, false), false);
    h2 = __builtin_amdgcn_udot4(d2, W2c, __builtin_amdgcn_udot4(d1, W2b, __builtin_amdgcn_udot4(d0, W2a, 0u, false), false), false);
    h3 = __builtin_amdgcn_udot4(d2, W3c, __builtin_amdgcn_udot4(d1, W3b, __builtin_amdgcn_udot4(d0, W3a, 0u, false), false), false);
}
// Vertical: blurred pixel = (sum_s k[s] * h[s] + 2^15) >> 16 from the seven horizontally blurred values of its column
// (each <= 255 * 256; the sums stay below 2^24, so 24-bit multiply-adds are exact)
__device__ __forceinline__ unsigned od_vblur7(unsigned p0, unsigned p1, unsigned p2, unsigned p3, unsigned p4, unsigned p5, unsigned p6) {
    unsigned v;
    asm("v_mad_u32_u24 %0, %1, 18, %2" : "=v"(v) : "v"(p0 + p6), "s"(32768u));
    asm("v_mad_u32_u24 %0, %1, 34, %2" : "=v"(v) : "v"(p1 + p5), "v"(v));
    asm("v_mad_u32_u24 %0, %1, 48, %2" : "=v"(v) : "v"(p2 + p4), "v"(v));
    asm("v_mad_u32_u24 %0, %1, 56, %2" : "=v"(v) : "v"(p3), "v"(v));
    return v >> 16;
}

// The same sum from the column-major layout, where the seven values of a column are consecutive u16: four dwords that
// hold them (starting at the dword that holds the first one; sh = 16 if the first value is that dword's high half, else 0)
// -> three v_alignbit + one shift put the pairs (t0,t1) (t2,t3) (t4,t5) (t6,-) into place, four v_dot2_u32_u16 weigh them.
typedef unsigned short od_us2 __attribute__((ext_vector_type(2)));
typedef int od_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned od_dot2(unsigned a, unsigned w, unsigned acc) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(od_us2, a), __builtin_bit_cast(od_us2, w), acc, false);
}
__device__ __forceinline__ unsigned od_vblur7_pairs(unsigned e0, unsigned e1, unsigned e2, unsigned e3) {
    const unsigned W01 = 18u | (34u << 16), W23 = 48u | (56u << 16), W45 = 48u | (34u << 16), W6 = 18u;
    return od_dot2(e3, W6, od_dot2(e2, W45, od_dot2(e1, W23, od_dot2(e0, W01, 32768u)))) >> 16;
}
__device__ __forceinline__ unsigned od_vblur7_dwords(unsigned d0, unsigned d1, unsigned d2, unsigned d3, unsigned sh) {
    return od_vblur7_pairs(__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh),
                           __builtin_amdgcn_alignbit(d3, d2, sh), d3 >> (sh & 31u));  // (only the low five bits of sh count)
}

// the same from values stored as h - 32768 (signed 16 bit, what the matrix pipe leaves of the biased pixels): signed dot products,
// the 256 * 32768 the seven taps took off added back with the rounding constant
typedef short od_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int od_sdot2(unsigned a, unsigned w, int acc) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(od_s2, a), __builtin_bit_cast(od_s2, w), acc, false);
}
__device__ __forceinline__ unsigned od_vblur7_dwords_s(unsigned d0, unsigned d1, unsigned d2, unsigned d3, unsigned sh) {
    const unsigned W01 = 18u | (34u << 16), W23 = 48u | (56u << 16), W45 = 48u | (34u << 16), W6 = 18u;
    const unsigned e0 = __builtin_amdgcn_alignbit(d1, d0, sh), e1 = __builtin_amdgcn_alignbit(d2, d1, sh),
                   e2 = __builtin_amdgcn_alignbit(d3, d2, sh), e3 = d3 >> (sh & 31u);
    return (unsigned)od_sdot2(e3, W6, od_sdot2(e2, W45, od_sdot2(e1, W23, od_sdot2(e0, W01, 32768 + 256 * 32768)))) >> 16;
}

// Test tap (a7, GaussianBlur directly): the blurred image of one pyramid level, BORDER_REFLECT_101, computed with
// od_hblur4 / od_vblur7.  One thread per aligned group of four output pixels; speed is irrelevant here.
__global__ void k_blur_level(const uint8_t *img, int pitch, int w, int h, uint8_t *dst, int dstPitch) {
    const int gx = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
    if (gx >= w || y >= h) return;
    unsigned hb[7][4];
    for (int s = 0; s < 7; s++) {
        const uint8_t *row = img + (size_t)reflect101(y + s - 3, h) * pitch;
        unsigned d[3] = {0u, 0u, 0u};
        for (int b = 0; b < 12; b++) d[b >> 2] |= (unsigned)row[reflect101(gx - 3 + b, w)] << (8 * (b & 3));
        od_hblur4(d[0], d[1], d[2], hb[s][0], hb[s][1], hb[s][2], hb[s][3]);
    }
    for (int j = 0; j < 4 && gx + j < w; j++) {
#if OD_LAYOUT
        // the column as k_orient_desc keeps it: consecutive u16, once starting in a low half and once in a high half
        const unsigned t[8] = {hb[0][j], hb[1][j], hb[2][j], hb[3][j], hb[4][j], hb[5][j], hb[6][j], 0u};
        unsigned v;
        if ((gx + j + y) & 1) v = od_vblur7_dwords(t[0] << 16, t[1] | (t[2] << 16), t[3] | (t[4] << 16), t[5] | (t[6] << 16), 16u);
        else v = od_vblur7_dwords(t[0] | (t[1] << 16), t[2] | (t[3] << 16), t[4] | (t[5] << 16), t[6], 0u);
        dst[(size_t)y * dstPitch + gx + j] = (uint8_t)v;
#else
        dst[(size_t)y * dstPitch + gx + j] = (uint8_t)od_vblur7(hb[0][j], hb[1][j], hb[2][j], hb[3][j], hb[4][j], hb[5][j], hb[6][j]);
#endif
    }
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

// FT_DEBUG_OD_PROFILE=1: where a wave of k_orient_desc spends its life - 100 MHz wall-clock stamps at the phase boundaries of
// wave 0 of every 16th workgroup, summed here ([0] = sampled waves, [1 + p] = ticks of phase p); the launcher prints them
__device__ unsigned long long g_odProf[16];
template <int OD_KPW>
__global__ __launch_bounds__(64 * OD_WAVES) void k_orient_desc(FtGeom g, const uint8_t *const *l0, int l0pitch,
                                                               const uint8_t *pyr, int alignedLoads, const FtSelKp *sel,
                                                               const int *selCount, FtOctArgs lay, int *nSel,
                                                               ft_keypoint *keysOut, uint8_t *descOut, FtSlotGrid sg, int prof) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index is uniform by construction; as a scalar it makes the keypoint lookup below (level search, entry
    // address, level geometry, image pointer, row addressing) SALU work instead of 64 identical VALU lanes
    const int wave = wave_index();
    int slot, blk;
    if (!ft_slot_block(sg, slot, blk)) return;
    const bool profW = prof && wave == 0 && (blockIdx.x & 15u) == 0;
    unsigned long long tPrev = profW ? wall_clock64() : 0;
    int profIdx = 0;
    auto tick = [&]() {
        if (!profW) return;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the phase's memory operations belong to it
        const unsigned long long t = wall_clock64();
        if (lane == 0) atomicAdd(&g_odProf[1 + profIdx], t - tPrev);
        profIdx++;
        tPrev = t;
    };
    // OD_KPW keypoints per wave, one after the other through the same LDS buffers - with the LOADS of all of them issued up
    // front: level lookup, selection entries, image pointers and the nine patch dwords per lane of every keypoint are
    // requested before the first one is processed, so the dependent chain selCount -> sel entry -> image pointer -> patch
    // (three to four memory round trips) is paid once per OD_KPW keypoints and the later patches arrive behind the
    // arithmetic of the earlier ones (SQ counters of round 2: the waves of this kernel waited 42 % of their time, 26 % active).
    // Only the loaded dwords are kept (9 registers per extra keypoint) - no tables are hoisted across keypoints, which is what
    // cost the four-keypoints-per-wave variant of round 2 its occupancy.
    const int kFirst = (blk * OD_WAVES + wave) * OD_KPW;
    // the octree leaves its result per level; keypoint k of the image (level order) is entry k - prefix of
    // the level that contains it
    // Lane l holds the count of level l (one vector load), a DPP scan over the row of 16 lanes turns the counts into
    // inclusive prefixes, and the level of keypoint k is the first lane whose prefix exceeds k (one ballot): a dozen
    // instructions where the scalar unit used to walk the FT_MAX_LEVELS levels one by one (150 scalar instructions per wave -
    // and the CU's four SIMDs share one scalar unit).
    int total = 0;
    const int cnt = lane < g.nlevels ? selCount[slot * g.nlevels + lane] : 0;
    int incl = cnt;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);  // row_shr:1 (lanes shifted in from outside the row read 0)
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);  // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);  // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);  // row_shr:8
    total = __builtin_amdgcn_readlane(incl, 15);
    if (kFirst == 0 && lane == 0) nSel[slot] = total;
    if (kFirst >= total) return;  // waves are independent: no workgroup barrier below
    const int nKp = min(OD_KPW, total - kFirst);
    uint8_t *raw = smem + (size_t)wave * OD_WAVE_BYTES;
    unsigned short *hb = (unsigned short *)(raw + OD_RAW_BYTES);
    // per keypoint of the wave: the selection entry and what the patch load needs (all wave-uniform)
    FtSelKp sk[OD_KPW];
    const uint8_t *imgK[OD_KPW];
    int pitchK[OD_KPW];
    bool interiorK[OD_KPW];
    unsigned pv[OD_KPW][9];
    const int rrL = (lane * 43) >> 9, ccL = lane - rrL * 12;  // lane / 12 for lane < 64
#pragma unroll
    for (int q = 0; q < OD_KPW; q++) {
        const int kk = min(kFirst + q, total - 1);  // (a keypoint beyond the last repeats it; its loads are harmless, it is not processed)
        const unsigned below = (unsigned)__builtin_amdgcn_ballot_w64(kk < incl) & 0xffffu;
        const int selLevel = __builtin_ctz(below | 0x8000u);
        const int prefix = __builtin_amdgcn_readlane(incl - cnt, selLevel);
        sk[q] = sel[(size_t)slot * g.maxKp + lay.selOff[selLevel] + (kk - prefix)];
    }
#pragma unroll
    for (int q = 0; q < OD_KPW; q++) {
        const int level = sk[q].level;
        imgK[q] = level_ptr(g, level, slot, l0, l0pitch, pyr, pitchK[q]);
        const int px0 = sk[q].x - OD_R, py0 = sk[q].y - OD_R;
        const int axq = px0 & 3;
        interiorK[q] = alignedLoads && px0 >= 0 && py0 >= 0 && py0 + OD_P <= g.lv[level].h && (px0 - axq) + OD_PP <= g.lv[level].w;
        if (interiorK[q] && lane < 60) {
            // 43 rows x 12 aligned dwords: coalesced 4-byte lanes, pixel (r, c) lands at raw[r*48 + ax + c].  Five rows
            // per step on lanes 0-59 with a fixed (row, dword) per lane, so a step is one load and one LDS store with
            // immediate offsets; all nine row groups are requested before the first LDS store
            const uint8_t *src = imgK[q] + (size_t)py0 * pitchK[q] + (px0 - axq);
            const unsigned laneOff = (unsigned)(rrL * pitchK[q] + 4 * ccL);
#pragma unroll
            for (int it = 0; it < 9; it++)
                pv[q][it] = (it < 8 || rrL < 3) ? gload<unsigned>(src + (laneOff + (unsigned)(it * 5 * pitchK[q]))) : 0u;  // scalar base + 32-bit lane offset
        }
    }
    // The keypoints of the wave go through the LDS buffers one after the other, but the wave-uniform arithmetic between the
    // moments and the samples - fastAtan2, the angle in radians, sin / cos in double: a fifth of the kernel's vector
    // instructions, all 64 lanes computing the same number - is done ONCE for both: keypoint 0's moments on lanes 0-31,
    // keypoint 1's on lanes 32-63.  Order: raw 0 -> moments 0 -> hblur 0 | raw 1 (the raw buffer is free again) -> moments 1 |
    // angles of both | samples 0 (hb 0) | hblur 1 -> samples 1.
    int axK[OD_KPW], m10K[OD_KPW], m01K[OD_KPW];
    auto stage = [&](const int q) {
        const FtSelKp s = sk[q];
        const int level = s.level;
        const int pitch = pitchK[q];
        const uint8_t *img = imgK[q];
        const int w = g.lv[level].w, h = g.lv[level].h;
        const int px0 = s.x - OD_R, py0 = s.y - OD_R;
        int ax = px0 & 3;
        if (interiorK[q]) {
            unsigned *rawLane = (unsigned *)raw + rrL * 12 + ccL;
            if (lane < 60) {
#pragma unroll
                for (int it = 0; it < 9; it++)
                    if (it < 8 || rrL < 3) rawLane[it * 60] = pv[q][it] ^ OD_BIAS;
            }
        } else {
            // BORDER_REFLECT_101 of the blur at the level's edges (keypoints are >= 19 px inside, the
            // patch reaches 21) or unaligned caller frames
            ax = 0;
            for (int i = lane; i < OD_P * OD_P; i += 64) {
                const int r = i / OD_P, c = i - r * OD_P;
                const int gy = reflect101(py0 + r, h), gx = reflect101(px0 + c, w);
                raw[r * OD_PP + c] = gload<uint8_t>(img + (size_t)gy * pitch + gx) ^ (uint8_t)OD_BIAS;
            }
        }
        axK[q] = ax;
    };
    auto moments = [&](const int q) {
        const int ax = axK[q];
        // IC_Angle: integer moments over the 31-px disc, four pixels per v_dot4 (tables above): lane = (row of an
        // 8-row group, dword of the row), four groups cover rows v = -15 .. 16 (row 16 has zero weights)
        int m10 = 0, m01 = 0;
        {
            const int rsub = lane >> 3, jd = lane & 7;
            const int sh = (ax + 6) & 3;
            // dword j of row v starts at raw byte (v + 21) * 48 + ax + 6 + 4j
            const unsigned *w = (const unsigned *)raw + (rsub + 6) * 12 + ((ax + 6) >> 2) + jd;
            int v = rsub - 15;
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const unsigned lo = w[it * 96], hi = w[it * 96 + 1];
                const unsigned D = __builtin_amdgcn_alignbyte(hi, lo, sh) ^ OD_BIAS;  // (the patch holds p - 128 for the matrix pipe)
                const int av = v < 0 ? -v : v;
                const unsigned dw = __builtin_amdgcn_udot4(D, c_mom.W[av * 8 + jd], 0u, false);
                const unsigned dm = __builtin_amdgcn_udot4(D, c_mom.M[av * 8 + jd], 0u, false);
                m10 += (int)dw - 16 * (int)dm;
                m01 += __mul24(v, (int)dm);  // |v| <= 16, dm <= 4 * 255 * 15
                v += 8;
            }
            m10 = wave_sum_i32(m10);
            m01 = wave_sum_i32(m01);
        }
        m10K[q] = m10;
        m01K[q] = m01;
    };
#if OD_MFMA
    // the tap operand of the lane: bytes e = 0 .. 7 are T[k = 8 g + e][b = n] = tap[8 g + e - n] (0 outside 0 .. 6) - an
    // 8-byte window of the tap sequence
    long tapOp;
    {
        const int s8 = 8 * (8 * (lane >> 4) - (lane & 15));
        const unsigned long long T = 0x0012223038302212ull;  // 18 34 48 56 48 34 18, lowest byte first
        tapOp = (long)(s8 >= 0 ? (s8 < 64 ? T >> s8 : 0ull) : (s8 > -64 ? T << -s8 : 0ull));
    }
#endif
    auto hblur = [&]() {
        // horizontal 7-tap pass on the packed bytes: a task = (row, aligned group of 4 raw byte positions);
        // two v_dot4_u32_u8 per output on byte windows cut out with v_alignbyte.  hb[r][b] = sum_t k[t] *
        // raw[r][b + t] for raw byte positions b in [0, 40) (16 bit, <= 255 * 256); output column c of the
        // blurred window is b = ax + c.
        {
#if OD_MFMA
            // hb - 32768 (43 rows x 40 positions, kept as SIGNED 16 bit: od_vblur7_dwords_s adds the 256 * 32768 back) = raw (43 x
            // 48 signed bytes p - 128) x T (banded: T[k][b] = tap[k - b]), as nine
            // 16 x 16 tiles with K = 32: tile (i, j) reads the bytes 16 j + 8 g .. + 7 of rows 16 i + n (lane = 16 g + n: one
            // ds_read_b64), and because the window of a tile starts at its own first output position the tap operand is the
            // same for all nine.  The accumulator of a lane holds rows 16 i + 4 g .. + 3 of output position 16 j + n: four
            // consecutive u16 of a column of hbT - two v_perm and one ds_write_b64.  (Rows 43 .. 47 and positions 40 .. 47
            // read whatever lies behind the patch - finite integers - and are not stored, row 43 - the dummy slot - aside.)
            const int n = lane & 15, g4 = lane >> 4;
            const uint8_t *aLane = raw + n * OD_PP + 8 * g4;
            uint8_t *oLane = (uint8_t *)hb + 2 * OD_RP * n + 8 * g4;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                // the three tiles of a row block: operands first, the three products back to back, then the stores
                long A[3];
                od_v4i d[3];
#pragma unroll
                for (int j = 0; j < 3; j++) A[j] = *(const long *)(aLane + i * 16 * OD_PP + 16 * j);
#pragma unroll
                for (int j = 0; j < 3; j++) d[j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(A[j], tapOp, od_v4i{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    uint2 pk;
                    pk.x = __builtin_amdgcn_perm((unsigned)d[j].y, (unsigned)d[j].x, 0x05040100u);
                    pk.y = __builtin_amdgcn_perm((unsigned)d[j].w, (unsigned)d[j].z, 0x05040100u);
                    if ((i < 2 || g4 < 3) && (j < 2 || n < 8)) *(uint2 *)(oLane + j * 16 * 2 * OD_RP + 32 * i) = pk;
                }
            }
#elif OD_LAYOUT
            // column-major output (hbT[column][row], OD_RP u16 slots per column): a lane takes a PAIR of rows (2P, 2P + 1) of its
            // group of four columns and stores the two values of a column as one dword; six row pairs per step on lanes 0-59
            const int rr = (lane * 13) >> 7, gq = lane - rr * 10;  // lane / 10 for lane < 64
            const unsigned *rwLane = (const unsigned *)(raw + 2 * rr * OD_PP) + gq;
            unsigned *hbLane = (unsigned *)hb + 4 * gq * (OD_RP / 2) + rr;
            if (lane < 60) {
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    if (it == 3 && rr > 3) break;  // row pairs 22, 23 do not exist (rows 0 .. 42; row 43 is a dummy)
                    const unsigned *rw = rwLane + it * (12 * OD_PP / 4);
                    unsigned a0, a1, a2, a3, b0, b1, b2, b3;
                    od_hblur4(rw[0], rw[1], rw[2], a0, a1, a2, a3);
                    od_hblur4(rw[OD_PP / 4], rw[OD_PP / 4 + 1], rw[OD_PP / 4 + 2], b0, b1, b2, b3);
                    unsigned *o = hbLane + it * 6;
                    o[0] = __builtin_amdgcn_perm(b0, a0, 0x05040100u);  // a | b << 16 (both < 2^16)
                    o[OD_RP / 2] = __builtin_amdgcn_perm(b1, a1, 0x05040100u);
                    o[2 * (OD_RP / 2)] = __builtin_amdgcn_perm(b2, a2, 0x05040100u);
                    o[3 * (OD_RP / 2)] = __builtin_amdgcn_perm(b3, a3, 0x05040100u);
                }
            }
#else
            // six rows per step on lanes 0-59, (row, group) fixed per lane: every LDS offset of a step is an immediate
            const int rr = (lane * 13) >> 7, gq = lane - rr * 10;  // lane / 10 for lane < 64
            const unsigned *rwLane = (const unsigned *)(raw + rr * OD_PP) + gq;
            unsigned short *hbLane = hb + rr * OD_HP + 4 * gq;
            if (lane < 60) {
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    if (it == 7 && rr > 0) break;  // rows 42 .. 47: only row 42 exists
                    const unsigned *rw = rwLane + it * (6 * OD_PP / 4);
                    const unsigned d0 = rw[0], d1 = rw[1], d2 = rw[2];
                    unsigned h0, h1, h2, h3;
                    od_hblur4(d0, d1, d2, h0, h1, h2, h3);
                    uint2 pk;
                    pk.x = __builtin_amdgcn_perm(h1, h0, 0x05040100u);  // h0 | h1 << 16 (both < 2^16) in one instruction
                    pk.y = __builtin_amdgcn_perm(h3, h2, 0x05040100u);
                    *(uint2 *)(hbLane + it * 6 * OD_HP) = pk;
                }
            }
#endif
        }
    };
    typedef float v2f __attribute__((ext_vector_type(2)));
    auto describe = [&](const int q, const float angle, const float ca, const float sb) {
        const int ax = axK[q];
        const int k = kFirst + q;
        const FtSelKp s = sk[q];
        const int cx = s.x, cy = s.y, level = s.level, response = s.response;
        // vertical 7-tap pass only where the pattern samples (512 of the 1369 window positions): the blurred
        // pixel at offset (r, c) from the keypoint is (sum_s k[s] * hb[18 + r + s][ax + 18 + c] + 2^15) >> 16
        // LDS byte offset of hb[18 + r][ax + 18 + c] = r * 80 + (2 c + hbase): one shift-add and one 24-bit multiply-add;
        // the seven taps are immediates from there, combined with 24-bit multiply-adds (sums stay below 2^24)
#if OD_LAYOUT
        // column-major: the seven values of a sample are consecutive u16 of column ax + 18 + c starting at row 18 + r: byte
        // offset 2 * ((ax + 18 + c) * OD_RP + 18 + r) in the wave's hbT
        // r and c arrive as the BIT PATTERNS of (offset + 1.5 * 2^23): adding that constant to a float rounds it to the
        // nearest integer, ties to even - cvRound - and leaves 0x4B400000 + offset in the register (|offset| < 2^22), one
        // instruction where v_rndne + v_cvt_i32 took two.  The constant parts come out in the wash: (r << 1) wraps to
        // 2 * offset + 2 * 0x4B400000, the 24-bit multiply sees 0x400000 + c, and both surpluses are folded into hbase.
        // (hbase carries the LDS address of hb itself, and the sample pointer is made from the integer: as smem + offset the
        // compiler adds the base of the dynamic LDS - zero, but a symbol to it - to every sample's address)
        typedef __attribute__((address_space(3))) const unsigned od_lds_u32;
        const int hbase = (int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)(uint8_t *)hb +
                          2 * ((ax + 18) * OD_RP + 18) - (int)(2u * 0x4B400000u + 0x400000u * (unsigned)(2 * OD_RP));
        auto blurred = [&](unsigned rBits, unsigned cBits) -> unsigned {
            const int byteIdx = vmad24((int)cBits, 2 * OD_RP, (int)((rBits << 1) + (unsigned)hbase));
            od_lds_u32 *p = (od_lds_u32 *)(uintptr_t)(unsigned)(byteIdx & ~3);
            // (bit shift of the window: 16 if the first value is a high half.  byteIdx is even, and both v_alignbit and the
            // shift take the low five bits of their operand: byteIdx << 3 serves without masking bit 1 out first)
#if OD_MFMA
            return od_vblur7_dwords_s(p[0], p[1], p[2], p[3], (unsigned)byteIdx << 3);
#else
            return od_vblur7_dwords(p[0], p[1], p[2], p[3], (unsigned)byteIdx << 3);
#endif
        };
#else
        const int hbase = (int)((const uint8_t *)(hb + 18 * OD_HP + ax + 18) - smem);
        auto blurred = [&](int r, int c) -> unsigned {
            const unsigned short *p = (const unsigned short *)(smem + vmad24(r, 2 * OD_HP, (c << 1) + hbase));
            return od_vblur7(p[0], p[OD_HP], p[2 * OD_HP], p[3 * OD_HP], p[4 * OD_HP], p[5 * OD_HP], p[6 * OD_HP]);
        };
#endif
        const v2f scPair = {sb, ca}, csPair = {ca, sb};
        unsigned long long words[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int p = q * 64 + lane;  // pair index: byte p/8, bit p%8
            const float4 pt = c_patternF.p[p];
            // (x, y) pairs times (sin, cos) and (cos, sin) pairs: packed fp32 multiplies, every product rounded on its own
            const v2f p0 = {pt.x, pt.y}, p1 = {pt.z, pt.w};
            const v2f a0 = p0 * scPair, b0 = p0 * csPair, a1 = p1 * scPair, b1 = p1 * csPair;
#if OD_LAYOUT
            const float rnd = 12582912.f;  // 1.5 * 2^23
            const unsigned r0 = __float_as_uint(__fadd_rn(__fadd_rn(a0.x, a0.y), rnd));  // cvRound(x0 sin + y0 cos), biased
            const unsigned c0 = __float_as_uint(__fadd_rn(__fsub_rn(b0.x, b0.y), rnd));  // cvRound(x0 cos - y0 sin), biased
            const unsigned r1 = __float_as_uint(__fadd_rn(__fadd_rn(a1.x, a1.y), rnd));
            const unsigned c1 = __float_as_uint(__fadd_rn(__fsub_rn(b1.x, b1.y), rnd));
#else
            const int r0 = __float2int_rn(__fadd_rn(a0.x, a0.y));  // x0 sin + y0 cos
            const int c0 = __float2int_rn(__fsub_rn(b0.x, b0.y));  // x0 cos - y0 sin
            const int r1 = __float2int_rn(__fadd_rn(a1.x, a1.y));
            const int c1 = __float2int_rn(__fsub_rn(b1.x, b1.y));
#endif
            words[q] = __ballot(blurred(r0, c0) < blurred(r1, c1));
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
    };
    tick();  // 0: level lookup, selection entries, patch loads (all keypoints of the wave) have arrived
    stage(0);
    wave_lds_sync();
    tick();  // 1: patch 0 staged in LDS
    moments(0);
    tick();  // 2: moments 0
    hblur();
    wave_lds_sync();  // hb 0 is complete, the raw buffer is free
    tick();  // 3: horizontal blur 0
    if constexpr (OD_KPW > 1) {
        if (nKp > 1) {  // wave-uniform
            stage(1);
            wave_lds_sync();
            moments(1);
        }
    }
    tick();  // 4: patch 1 staged + its moments
    // lane group of keypoint q for the shared angle arithmetic: the halves of the wave
    constexpr int GROUP = 32;
    float mY = (float)m01K[0], mX = (float)m10K[0];
    if constexpr (OD_KPW > 1) {
        if (nKp > 1 && lane >= GROUP) {
            mY = (float)m01K[1];
            mX = (float)m10K[1];
        }
    }
    const float angleL = fast_atan2_deg(mY, mX);
    // computeOrbDescriptor (ORBextractor.cc:72-74): angle in radians as float, then std::cos(float) / std::sin(float)
    // = glibc's cosf / sinf, reproduced bit for bit (libm_f32.h)
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float ar = __fmul_rn(angleL, factorPI);
    const float caL = ft_libm::cosf_glibc(ar), sbL = ft_libm::sinf_glibc(ar);
    auto lane_value = [&](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
    tick();  // 5: angles of all keypoints of the wave
    describe(0, lane_value(angleL, 0), lane_value(caL, 0), lane_value(sbL, 0));
    tick();  // 6: samples + stores of keypoint 0
    if constexpr (OD_KPW > 1) {
        if (nKp > 1) {
            wave_lds_sync();  // the samples in front have read hb (and patch 1 is staged)
            hblur();
            wave_lds_sync();
            tick();  // 7: horizontal blur 1
            describe(1, lane_value(angleL, GROUP), lane_value(caL, GROUP), lane_value(sbL, GROUP));
            tick();  // 8: samples + stores of keypoint 1
        }
    }
    if (profW && lane == 0) atomicAdd(&g_odProf[0], 1ull);
}

}  // namespace

static int fast_tile_pitch(const FtGeom &g) {
    // fixed LDS pitch (all ring offsets immediates) when every level's cell fits: wCell + 6 + 3 <= TP
    int need = 0;
    for (int l = 0; l < g.nlevels; l++)
        if (g.lv[l].nCols * g.lv[l].nRows > 0) need = std::max(need, g.lv[l].wCell + 9);
    return need <= 48 ? 48 : (need <= 64 ? 64 : 0);
}

size_t ft_fast_smem_bytes(const FtGeom &g) {
    const int TP = fast_tile_pitch(g);
    // the fixed-pitch variants carve tile | score plane | lists at the largest tile and the largest score plane of any
    // level; the any-size variant at the sizes of the cell's own level
    size_t mxTile = 0, mxScore = 0, mxSum = 0;
    for (int l = 0; l < g.nlevels; l++) {
        const FtLevelGeom &L = g.lv[l];
        const size_t t = (size_t)fc_tile_bytes(L.wCell, L.hCell, TP), sc = (size_t)fc_score_bytes(L.wCell, L.hCell, TP);
        mxTile = std::max(mxTile, t);
        mxScore = std::max(mxScore, sc);
        mxSum = std::max(mxSum, t + sc);
    }
    return (TP ? mxTile + mxScore : mxSum) + fc_list_bytes();
}

// the row-streaming kernel needs every lane's four taps inside one aligned 8-byte window of a source row
static bool pyr_rows_fits(const FtGeom &g, int level) {
    const FtLevelGeom &D = g.lv[level], &P = g.lv[level - 1];
    // the window starts at sx(dx) & ~3 (<= 3 bytes before the first tap), sx(dx + 1) <= sx(dx) + ceil(scale), and the
    // second tap of column dx + 1 is one further: 3 + ceil(scale) + 1 <= 7 holds for every scale <= 3
    return (double)P.w / (double)D.w < 2.5 && P.w >= 8;
}

int ft_launch_pyramid(hipStream_t st, const FtGeom &g, int batch, const uint8_t *const *l0, int l0pitch,
                      uint8_t *pyr, const FtTap *taps, int alignedLoads, int rowsKernel) {
    const bool rowsOn = rowsKernel != 0;
    for (int rep = ft_debug_repeat("pyr"); rep > 0; rep--)
    for (int level = 1; level < g.nlevels; level++) {
        const FtLevelGeom &D = g.lv[level], &P = g.lv[level - 1];
        // a launch of a few images is latency bound: the tile kernel's many short waves finish a level sooner than the
        // row-streaming kernel's few long ones (752x480 frame: 0.19 against 0.23 ms); wide launches take the streaming kernel
        if (rowsOn && batch >= 8 && alignedLoads && pyr_rows_fits(g, level)) {
            const int stripsX = (D.w + PR_COLS - 1) / PR_COLS, stripsY = (D.h + PR_RB - 1) / PR_RB;
            dim3 grid, block(64, 1, 1);
            const FtSlotGrid sg = ft_slot_grid(stripsX * stripsY, batch, grid);
            // bytes of a source row that may be read: the whole pitch of a slot level, the width rounded up to a dword
            // (inside the 4-byte aligned stride) of a caller's frame
            const int readableEnd = level == 1 ? std::min((P.w + 3) & ~3, l0pitch) : P.pitch;
            if (D.area2x)
                hipLaunchKernelGGL(k_pyr_rows<true>, grid, block, 0, st, g, level, l0, l0pitch, pyr, taps, sg, stripsX,
                                   div_magic_of((unsigned)stripsX), readableEnd);
            else
                hipLaunchKernelGGL(k_pyr_rows<false>, grid, block, 0, st, g, level, l0, l0pitch, pyr, taps, sg, stripsX,
                                   div_magic_of((unsigned)stripsX), readableEnd);
            continue;
        }
        // LDS footprint of a 64 x 8 output tile: ceil(tile * scale) + the second tap + the margin of the
        // arithmetic footprint bound + alignment slack
        const int cols = (int)((long long)PD_TW * P.w / D.w) + 8, rowsN = (int)((long long)PD_TH * P.h / D.h) + 8;
        const int ldsPitch = (cols + 3 + 3) & ~3;
        // scale = source size / destination size in 16.16, rounded up (the error stays far below one pixel)
        const unsigned sxQ16 = (unsigned)(((unsigned long long)P.w << 16) / (unsigned)D.w) + 1u;
        const unsigned syQ16 = (unsigned)(((unsigned long long)P.h << 16) / (unsigned)D.h) + 1u;
        const int tilesX = (D.w + PD_TW - 1) / PD_TW, tilesY = (D.h + PD_TH - 1) / PD_TH;
        dim3 grid, block(64, 1, 1);
        const FtSlotGrid sg = ft_slot_grid(tilesX * tilesY, batch, grid);
        hipLaunchKernelGGL(k_pyr_down, grid, block, (size_t)ldsPitch * rowsN + 4, st, g, level, l0, l0pitch, pyr, taps,
                           alignedLoads, ldsPitch, rowsN, sxQ16, syQ16, sg, tilesX, div_magic_of((unsigned)tilesX));
    }
    FT_HIP(hipGetLastError());
    return FT_OK;
}

// LDS of a launch over the levels of `mask` at pitch TP (fixed-pitch variants: the largest tile and the largest score plane of
// those levels; the any-size variant: the largest sum)
static size_t fast_smem_of(const FtGeom &g, unsigned mask, int TP, int *tileBytes, int *scoreBytes) {
    size_t mxTile = 0, mxScore = 0, mxSum = 0;
    for (int l = 0; l < g.nlevels; l++) {
        const FtLevelGeom &L = g.lv[l];
        if (!((mask >> l) & 1u) || L.nCols * L.nRows <= 0) continue;
        const size_t t = (size_t)fc_tile_bytes(L.wCell, L.hCell, TP), sc = (size_t)fc_score_bytes(L.wCell, L.hCell, TP);
        mxTile = std::max(mxTile, t);
        mxScore = std::max(mxScore, sc);
        mxSum = std::max(mxSum, t + sc);
    }
    if (tileBytes) *tileBytes = (int)mxTile;
    if (scoreBytes) *scoreBytes = (int)mxScore;
    return (TP ? mxTile + mxScore : mxSum) + fc_list_bytes();
}

// the cells of the levels of `mask` as runs of consecutive cells; false: more than two runs (or none)
static bool fast_cell_ranges(const FtGeom &g, unsigned mask, FtCellRanges &cr) {
    int runs = 0, lo[2] = {0, 0}, n[2] = {0, 0};
    bool open = false;
    for (int l = 0; l < g.nlevels; l++) {
        const int cells = g.lv[l].nCols * g.lv[l].nRows;
        if (cells <= 0) continue;  // (no cells: the level neither starts nor ends a run)
        if ((mask >> l) & 1u) {
            if (!open) {
                if (runs == 2) return false;
                lo[runs] = g.lv[l].cellBase;
                n[runs] = 0;
                runs++;
                open = true;
            }
            n[runs - 1] += cells;
        } else {
            open = false;
        }
    }
    cr.lo0 = lo[0]; cr.n0 = n[0]; cr.lo1 = lo[1]; cr.n1 = n[1];
    return runs > 0;
}

int ft_launch_fast_cells(hipStream_t st, const FtGeom &g, int batch, const uint8_t *const *l0, int l0pitch,
                         const uint8_t *pyr, int iniTh, int minTh, int alignedLoads, int *cellCount,
                         uint32_t *stage, int ordered, const FtCellRec *cellTab) {
    if (g.totalCells == 0) return FT_OK;  // every level is too small for a 35-px cell: no candidates
    typedef void (*FastFn)(FtGeom, const uint8_t *const *, int, const uint8_t *, int, int, int, int *, uint32_t *, const FtCellRec *,
                           FtSlotGrid, int, int, int, FtCellRanges);
    static const int dbg = ft_debug_env("FT_DEBUG_FAST") ? atoi(ft_debug_env("FT_DEBUG_FAST")) : 0;
    static const bool occDbg = ft_debug_env("FT_DEBUG_OCC") != nullptr;  // resident workgroups per CU as the runtime computes them
    static const bool noSplit = ft_debug_env("FT_DEBUG_FAST_NOSPLIT") != nullptr;  // A/B aid: one launch at the widest pitch
    const unsigned all = (1u << g.nlevels) - 1u;
    // One launch per tile pitch.  The pitch of a launch is that of its widest cell (every ring offset an instruction
    // immediate), and with it the LDS per wave: an image size whose small levels need 64 bytes (512 x 512: cells of 44 and 47
    // pixels on levels 5 and 6, 25 of 490 cells) used to put EVERY cell at 64 - 19 waves per CU instead of 25.  Now the
    // levels that fit 48 bytes run at 48 and the few others in a launch of their own (the same stream: no ordering needed,
    // the cells write disjoint staging slots).
    unsigned mask48 = 0, mask64 = 0, maskAny = 0;
    for (int l = 0; l < g.nlevels; l++) {
        if (g.lv[l].nCols * g.lv[l].nRows <= 0) continue;
        const int need = g.lv[l].wCell + 9;
        (need <= 48 ? mask48 : need <= 64 ? mask64 : maskAny) |= 1u << l;
    }
    struct Part { unsigned mask; int TP; };
    Part parts[3];
    int nParts = 0;
    FtCellRanges probe;
    const bool splittable = !noSplit && !maskAny && mask48 && mask64 && fast_cell_ranges(g, mask48, probe) && fast_cell_ranges(g, mask64, probe);
    if (splittable) {
        parts[nParts++] = {mask48, 48};
        parts[nParts++] = {mask64, 64};
    } else {
        parts[nParts++] = {all, fast_tile_pitch(g)};
    }
    const int runBlock = 8 * FC_XCD_RUN;  // grid padded to whole rounds of the XCD mapping
    for (int pi = 0; pi < nParts; pi++) {
        const int TP = parts[pi].TP;
        FtCellRanges cr;
        if (!fast_cell_ranges(g, parts[pi].mask, cr)) {  // (the unsplit launch: all cells, one run)
            cr.lo0 = 0; cr.n0 = g.totalCells; cr.lo1 = 0; cr.n1 = 0;
        }
        if (nParts == 1) { cr.lo0 = 0; cr.n0 = g.totalCells; cr.lo1 = 0; cr.n1 = 0; }
        const int nCells = cr.n0 + cr.n1;
        // LDS carve of the fixed-pitch variants: the largest tile and score plane of the launch's levels
        int tileBytes = 0, scoreBytes = 0;
        const size_t smem = fast_smem_of(g, parts[pi].mask, TP, &tileBytes, &scoreBytes);
        dim3 grid(((nCells + runBlock - 1) / runBlock) * runBlock, batch, 1), block(64, 1, 1);
        FtSlotGrid sg;
        sg.blocksPerSlot = 0; sg.batch = batch; sg.xcdMap = 0; sg.magic = 0;
        if (batch >= 8) sg = ft_slot_grid(nCells, batch, grid);  // image -> XCD; smaller launches keep the cell-run mapping
        const FastFn fn = TP == 48 ? (ordered ? k_fast_cells<48, true> : k_fast_cells<48, false>)
                          : TP == 64 ? (ordered ? k_fast_cells<64, true> : k_fast_cells<64, false>)
                                     : k_fast_cells<0, true>;  // any-size cells: the linear pass is ordered anyway
        if (smem > 64 * 1024)  // very wide cells (tiny images with one cell column): raise the dynamic LDS limit
            FT_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        if (occDbg) {
            int nb = 0;
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)fn, 64, smem);
            fprintf(stderr, "[ft] k_fast_cells<%d,%d>: %d cells per image, %zu B of LDS per wave, %d waves per CU\n", TP, ordered, nCells, smem, nb);
        }
        for (int rep = ft_debug_repeat("fast"); rep > 0; rep--)
            hipLaunchKernelGGL(fn, grid, block, smem, st, g, l0, l0pitch, pyr, iniTh, minTh, alignedLoads, cellCount, stage, cellTab, sg, dbg,
                               tileBytes, scoreBytes, cr);
    }
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_compact(hipStream_t st, const FtGeom &g, int batch, const int *cellCount, const uint32_t *stage,
                      uint32_t *cand, int *candCount) {
    int maxCells = 0;
    for (int l = 0; l < g.nlevels; l++) maxCells = std::max(maxCells, g.lv[l].nCols * g.lv[l].nRows);
    dim3 grid2(g.nlevels, batch, 1), block(256, 1, 1);
    for (int rep = ft_debug_repeat("compact"); rep > 0; rep--)
    hipLaunchKernelGGL(k_compact, grid2, block, (size_t)(maxCells + 1) * sizeof(int), st, g, cellCount, stage, cand,
                       candCount);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_blur_level(hipStream_t st, const uint8_t *img, int pitch, int w, int h, uint8_t *dst, int dstPitch) {
    dim3 grid(((w + 3) / 4 + 63) / 64, h, 1), block(64, 1, 1);
    hipLaunchKernelGGL(k_blur_level, grid, block, 0, st, img, pitch, w, h, dst, dstPitch);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_orient_desc(hipStream_t st, const FtGeom &g, int batch, const uint8_t *const *l0, int l0pitch,
                          const uint8_t *pyr, int alignedLoads, const FtSelKp *sel, const int *selCount,
                          const FtOctArgs &layout, int *nSel, ft_keypoint *keys, uint8_t *desc) {
    dim3 grid, block(64 * OD_WAVES, 1, 1);
    // Launches that fill the chip (8+ images) give a wave several keypoints, which hides the load chains behind arithmetic;
    // a frame or two at a time (latency mode) leaves SIMDs idle as it is, and a wave that works through two keypoints one
    // after the other only doubles the time to the last result (stereo pair 752x480: 0.284 against 0.244 ms): one each.
    const int kpw = batch >= 8 ? OD_KPW_WIDE : 1;
    const FtSlotGrid sg = ft_slot_grid((g.maxKp + OD_WAVES * kpw - 1) / (OD_WAVES * kpw), batch, grid);
    const size_t smem = OD_WAVES * (size_t)OD_WAVE_BYTES;
    static const bool occDbg = ft_debug_env("FT_DEBUG_OCC") != nullptr;
    if (occDbg) {
        int nb = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kpw == 1 ? (const void *)k_orient_desc<1> : (const void *)k_orient_desc<OD_KPW_WIDE>,
                                                     64 * OD_WAVES, smem);
        fprintf(stderr, "[ft] k_orient_desc<%d>: %zu B of LDS per workgroup, %d workgroups per CU\n", kpw, smem, nb);
    }
    static const int prof = ft_debug_env("FT_DEBUG_OD_PROFILE") ? 1 : 0;
    for (int rep = ft_debug_repeat("orient"); rep > 0; rep--) {
        if (kpw == 1)
            hipLaunchKernelGGL(k_orient_desc<1>, grid, block, smem, st, g, l0, l0pitch, pyr, alignedLoads, sel, selCount, layout,
                               nSel, keys, desc, sg, 0);
        else
            hipLaunchKernelGGL(k_orient_desc<OD_KPW_WIDE>, grid, block, smem, st, g, l0, l0pitch, pyr, alignedLoads, sel, selCount,
                               layout, nSel, keys, desc, sg, prof);
    }
    if (prof && kpw > 1) {
        static int launches = 0;
        if (++launches % 64 == 0) {  // (synchronises: a debugging aid)
            unsigned long long h[16];
            if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_odProf), sizeof h) == hipSuccess && h[0]) {
                static const char *names[9] = {"loads (level, selection, patches)", "stage 0", "moments 0", "hblur 0", "stage 1 + moments 1",
                                               "angles", "samples 0 + stores", "hblur 1", "samples 1 + stores"};
                double tot = 0;
                for (int k = 0; k < 9; k++) tot += (double)h[1 + k];
                fprintf(stderr, "[ft] k_orient_desc<2> wave life, %llu sampled waves, %.2f us per wave:\n", h[0], tot / h[0] / 100.0);
                for (int k = 0; k < 9; k++) fprintf(stderr, "[ft]   %-36s %6.2f us  %5.1f %%\n", names[k], h[1 + k] / (double)h[0] / 100.0, 100.0 * h[1 + k] / tot);
            }
        }
    }
    FT_HIP(hipGetLastError());
    return FT_OK;
}
