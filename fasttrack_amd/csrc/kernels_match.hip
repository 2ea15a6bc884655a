// HIP kernels of the Hamming matchers for gfx950 (wave64): 256-bit descriptors as 4 x uint64 with
// __popcll, wave-wide argmin on packed (distance, index) keys so ties resolve exactly like the
// reference's sequential "dist < bestDist" scans.
//   k_stereo_match   Frame::ComputeStereoMatches  (reference src/Frame.cc:835-989)
//   k_stereo_median  the 1.5*1.4*median SAD cut    (src/Frame.cc:991-1004)
//   k_fisheye_2nn    BFMatcher knnMatch(k=2)+ratio (src/Frame.cc:1231-1255)
//   k_hamming_pairs  ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:2256-2272)
#include "ft_internal.h"
#include "ft_search.h"
#include "kb8_math.h"
#include "wave_ops.h"

namespace {

__device__ __forceinline__ const uint8_t *level_ptr(const FtGeom &g, int level, int slot, const uint8_t *const *l0,
                                                    int l0pitch, const uint8_t *pyr, int &pitch) {
    if (level == 0) {
        pitch = l0pitch;
        return l0[slot];
    }
    pitch = g.lv[level].pitch;
    return pyr + (size_t)slot * g.pyrPerSlot + g.lv[level].off;
}

__device__ __forceinline__ int hamming256(const unsigned long long a[4], const unsigned long long *b) {
    return __popcll(a[0] ^ b[0]) + __popcll(a[1] ^ b[1]) + __popcll(a[2] ^ b[2]) + __popcll(a[3] ^ b[3]);
}

// One wave per left keypoint.
__global__ __launch_bounds__(256) void k_stereo_match(FtGeom g, const uint8_t *const *l0L, const uint8_t *const *l0R,
                                                      int l0pitchL, int l0pitchR, const uint8_t *pyrL,
                                                      const uint8_t *pyrR, FtStereoArgs a, FtSlotGrid sg) {
    const int lane = threadIdx.x & 63, wave = wave_index();  // scalar: the left keypoint's data become SALU / s_load work
    int slot, blk;
    if (!ft_slot_block(sg, slot, blk)) return;
    const int iL = blk * 4 + wave;
    const size_t base = (size_t)slot * a.capacity;
    // the keypoint is requested together with the count that decides whether it exists (the arrays hold `capacity`
    // entries per image, so the read is in bounds either way): one memory round trip instead of two
    const ft_keypoint kpL = a.keysL[base + min(iL, a.capacity - 1)];
    const int nL = a.nL[slot];
    if (iL >= nL) return;
    float outU = -1.0f, outD = -1.0f;
    int outSad = -1, outHam = -1;
    const int levelL = kpL.octave;
    const float vL = kpL.y, uL = kpL.x;
    const int nRows = g.lv[0].h;
    const int row = (int)vL;
    const float minZ = a.mb;
    const float minD = 0.f;
    const float maxD = __fdiv_rn(a.mbf, minZ);
    const float minU = __fsub_rn(uL, maxD);
    const float maxU = __fsub_rn(uL, minD);
    unsigned long long dL[4];
    {
        const unsigned long long *p = (const unsigned long long *)(a.descL + (base + iL) * 32);
        dL[0] = p[0]; dL[1] = p[1]; dL[2] = p[2]; dL[3] = p[3];
    }
    unsigned best = 0xffffffffu;  // (dist << 16) | iR
    float bestX = 0.f;            // x of this lane's best right keypoint
    if (row >= 0 && row < nRows && !(maxU < 0)) {
        // right keypoints are bucketed by (int)y (k_stereo_rowsort); the band of keypoint iR covers `row` only
        // if |y_R - row| < r + 1 with r = 2 * sf[octave] <= 2 * max scale factor, so scanning the buckets
        // [row - reach, row + reach] sees every candidate of the reference's vRowIndices[row]; the exact band,
        // octave and disparity tests are applied to each of them.
        const int reach = (int)ceilf(2.0f * g.sf[g.nlevels - 1]) + 1;
        const int *rs = a.rowStart + (size_t)slot * a.rowStride;
        const int t0 = rs[max(row - reach, 0)], t1 = rs[min(row + reach + 1, nRows)];
        const FtSortedR *srt = a.sorted + base;
        for (int t = t0 + lane; t < t1; t += 64) {
            const FtSortedR kpR = srt[t];  // consecutive lanes read consecutive 48-byte entries
            // row band of the right keypoint (Frame.cc:852-862): rows floor(y-r) .. ceil(y+r), r = 2*sf[octave]
            const float r = __fmul_rn(2.0f, g.sf[kpR.octave]);
            const int maxr = (int)ceilf(__fadd_rn(kpR.y, r));
            const int minr = (int)floorf(__fsub_rn(kpR.y, r));
            if (row < minr || row > maxr) continue;
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = hamming256(dL, kpR.desc);
                const unsigned key = ((unsigned)dist << 16) | (unsigned)kpR.idx;
                if (key < best) {
                    best = key;
                    bestX = uR;
                }
            }
        }
    }
    const unsigned myBest = best;
    best = wave_min_u32(best);
    // x of the winner travels with it: the lane that holds the minimum (keys are unique: they contain the index)
    const unsigned long long winner = __ballot(myBest == best);
    const float uR0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bestX), (int)__ffsll((long long)winner) - 1));
    const int bestDist = (int)(best >> 16);
    const int thOrbDist = (FT_TH_HIGH + FT_TH_LOW) / 2;
    // bestDist starts at TH_HIGH in the reference, so only dist < TH_HIGH ever registers; < thOrbDist is stricter
    if (best != 0xffffffffu && bestDist < thOrbDist) {
        const int bestIdxR = (int)(best & 0xffffu);
        outHam = bestIdxR;
        const float scaleFactor = g.invsf[levelL];
        const float scaleduL = roundf(__fmul_rn(kpL.x, scaleFactor));
        const float scaledvL = roundf(__fmul_rn(kpL.y, scaleFactor));
        const float scaleduR0 = roundf(__fmul_rn(uR0, scaleFactor));
        const int w = 5, Ls = 5;
        int pitchL, pitchR;
        const uint8_t *imL = level_ptr(g, levelL, slot, l0L, l0pitchL, pyrL, pitchL);
        const uint8_t *imR = level_ptr(g, levelL, slot, l0R, l0pitchR, pyrR, pitchR);
        const int lw = g.lv[levelL].w, lh = g.lv[levelL].h;
        const float iniu = __fsub_rn(__fadd_rn(scaleduR0, (float)Ls), (float)w);
        const float endu = __fadd_rn(__fadd_rn(__fadd_rn(scaleduR0, (float)Ls), (float)w), 1.0f);
        const int yl0 = (int)(scaledvL - w), xl0 = (int)(scaleduL - w), xr00 = (int)scaleduR0 - Ls - w;
        bool ok = !(iniu < 0 || endu >= (float)lw);
        // windows that would leave the level make cv::Mat::rowRange/colRange throw in the reference;
        // unreachable for extractor keypoints (>= 19 px inside), guarded against stray reads
        ok = ok && yl0 >= 0 && yl0 + 2 * w + 1 <= lh && xl0 >= 0 && xl0 + 2 * w + 1 <= lw && xr00 >= 0 &&
             xr00 + 2 * (w + Ls) + 1 <= lw;
        if (ok) {
            int bestS = 0x7fffffff, bestinc = 0;
            float dists[11];
            const int xl0a = xl0 & ~3, xr0a = xr00 & ~3;
            // the aligned windows (16 bytes of a left row, 24 + 4 of a right row) stay inside the rows: wave-uniform
            if (a.alignedLoads && xl0a + 16 <= lw && xr0a + 28 <= lw) {
                // ---- SAD search on packed bytes (round 4).  The 11 x 11 left patch and the 11 x 21 right strip are fetched as
                // aligned dwords - one load instruction per wave for each (44 and 66 dwords) where every lane used to issue 24
                // byte loads - and parked in the wave's 476 bytes of LDS.  Then lane (shift group g = lane / 16, row r =
                // lane % 16) takes row r at shift s = 4 k + g in round k = 0, 1, 2: the 11 bytes of the left row and of the
                // shifted right row are three dwords each (v_alignbyte cuts them out, the twelfth byte is masked), three
                // v_sad_u8 add up the row's absolute differences, four DPP adds inside the row of 16 lanes sum the 11 rows.
                __shared__ unsigned sadBuf[4][44 + 77 + 3];
                unsigned *ldsL = sadBuf[wave], *ldsR = ldsL + 44;  // right rows at a pitch of 7 dwords (the last one: padding)
                {
                    const int rl = lane >> 2, dl = lane & 3;  // lanes 0 .. 43
                    const unsigned vL = gload<unsigned>(imL + (unsigned)((yl0 + min(rl, 10)) * pitchL + xl0a + 4 * dl));  // (scalar base + 32-bit lane offset)
                    const int rr0 = (lane * 43) >> 8, dr0 = lane - 6 * rr0;  // lane / 6 for lane < 64: rows 0 .. 10
                    const unsigned vR0 = gload<unsigned>(imR + (unsigned)((yl0 + rr0) * pitchR + xr0a + 4 * dr0));
                    // dwords 64, 65 (row 10, dwords 4, 5) and the padding dword of every row: lanes 0 .. 12
                    const int i1 = lane < 2 ? 64 + lane : 0;
                    const unsigned vR1 = gload<unsigned>(imR + (unsigned)((yl0 + (lane < 2 ? 10 : min(lane - 2, 10))) * pitchR + xr0a +
                                                                    (lane < 2 ? 4 * (4 + lane) : 24)));
                    if (lane < 44) ldsL[lane] = vL;
                    ldsR[rr0 * 7 + dr0] = vR0;
                    if (lane < 2) ldsR[10 * 7 + (i1 - 60)] = vR1;
                    else if (lane < 13) ldsR[(lane - 2) * 7 + 6] = vR1;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int g4 = lane >> 4, r = lane & 15, rc = min(r, 10);
                const int offL = xl0 - xl0a, offR = xr00 - xr0a;
                const unsigned l0d = ldsL[rc * 4], l1d = ldsL[rc * 4 + 1], l2d = ldsL[rc * 4 + 2], l3d = ldsL[rc * 4 + 3];
                const unsigned Lw0 = __builtin_amdgcn_alignbyte(l1d, l0d, (unsigned)offL), Lw1 = __builtin_amdgcn_alignbyte(l2d, l1d, (unsigned)offL),
                               Lw2 = __builtin_amdgcn_alignbyte(l3d, l2d, (unsigned)offL) & 0x00ffffffu;
                unsigned sums[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int sft = min(4 * k + g4, 10);  // (shift 11 does not exist: the fourth group of the last round repeats shift 10)
                    const int o = offR + sft;            // byte offset of the shifted row inside the aligned right row: 0 .. 13
                    const unsigned *pr = ldsR + rc * 7 + (o >> 2);
                    const unsigned r0 = pr[0], r1 = pr[1], r2 = pr[2], r3 = pr[3];
                    const unsigned sh = (unsigned)(o & 3);
                    const unsigned Rw0 = __builtin_amdgcn_alignbyte(r1, r0, sh), Rw1 = __builtin_amdgcn_alignbyte(r2, r1, sh),
                                   Rw2 = __builtin_amdgcn_alignbyte(r3, r2, sh) & 0x00ffffffu;
                    unsigned d = __builtin_amdgcn_sad_u8(Lw2, Rw2, __builtin_amdgcn_sad_u8(Lw1, Rw1, __builtin_amdgcn_sad_u8(Lw0, Rw0, 0u)));
                    d = r < 11 ? d : 0u;
                    // sum over the row of 16 lanes (every lane of the row ends up with it)
                    d += (unsigned)__builtin_amdgcn_update_dpp(0, (int)d, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                    d += (unsigned)__builtin_amdgcn_update_dpp(0, (int)d, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
                    d += (unsigned)__builtin_amdgcn_update_dpp(0, (int)d, 0x141, 0xF, 0xF, true);  // row_half_mirror
                    d += (unsigned)__builtin_amdgcn_update_dpp(0, (int)d, 0x140, 0xF, 0xF, true);  // row_mirror
                    sums[k] = d;
                }
#pragma unroll
                for (int sft = 0; sft < 11; sft++) {
                    const int dS = __builtin_amdgcn_readlane((int)sums[sft >> 2], 16 * (sft & 3));
                    dists[sft] = (float)dS;
                    if (dS < bestS) {
                        bestS = dS;
                        bestinc = sft - Ls;
                    }
                }
            } else {
            // 11x11 left patch: each lane owns pixels lane and lane+64 (< 121)
            const int p0 = lane, p1 = lane + 64;
            const int r0 = p0 / 11, c0 = p0 - r0 * 11, r1 = p1 / 11, c1 = p1 - r1 * 11;
            // every load of the SAD search is issued before the first reduction: the left patch and, per lane, the 11
            // consecutive right-image bytes its two pixels slide over (a wave used to pay the memory latency 11 times)
            const uint8_t *pl0 = imL + (size_t)(yl0 + r0) * pitchL + xl0 + c0;
            const uint8_t *pl1 = imL + (size_t)(yl0 + min(r1, 10)) * pitchL + xl0 + c1;
            const uint8_t *pr0 = imR + (size_t)(yl0 + r0) * pitchR + xr00 + c0;
            const uint8_t *pr1 = imR + (size_t)(yl0 + min(r1, 10)) * pitchR + xr00 + c1;
            // the lanes whose second pixel does not exist (p1 >= 121) carry 0 on both sides
            const bool two = p1 < 121;
            const unsigned a0 = gload<uint8_t>(pl0);
            const unsigned a1r = gload<uint8_t>(pl1);
            unsigned b0[11], b1[11];
#pragma unroll
            for (int s = 0; s < 11; s++) {
                b0[s] = gload<uint8_t>(pr0 + s);
                b1[s] = gload<uint8_t>(pr1 + s);
            }
            const unsigned a1 = two ? a1r : 0u;
#pragma unroll
            for (int s = 0; s < 11; s++) b1[s] = two ? b1[s] : 0u;
            // two shifts per register: v_sad_u8 adds |a - b| to the low half, v_sad_hi_u8 to the high half (a lane's two
            // pixels give at most 510, a wave's sum at most 121 * 255 < 2^16), so six wave sums serve the eleven shifts
#pragma unroll
            for (int s = 0; s < 11; s += 2) {
                unsigned d = __builtin_amdgcn_sad_u8(a1, b1[s], __builtin_amdgcn_sad_u8(a0, b0[s], 0u));
                if (s + 1 < 11) d = __builtin_amdgcn_sad_hi_u8(a1, b1[s + 1], __builtin_amdgcn_sad_hi_u8(a0, b0[s + 1], d));
                const unsigned sum = (unsigned)wave_sum_i32((int)d);
                const int dLo = (int)(sum & 0xffffu), dHi = (int)(sum >> 16);
                dists[s] = (float)dLo;
                if (dLo < bestS) {
                    bestS = dLo;
                    bestinc = s - Ls;
                }
                if (s + 1 < 11) {
                    dists[s + 1] = (float)dHi;
                    if (dHi < bestS) {
                        bestS = dHi;
                        bestinc = s + 1 - Ls;
                    }
                }
            }
            }
            if (!(bestinc == -Ls || bestinc == Ls)) {
                float dist1 = 0, dist2 = 0, dist3 = 0;
#pragma unroll
                for (int s = 1; s < 10; s++)
                    if (s == bestinc + Ls) {
                        dist1 = dists[s - 1];
                        dist2 = dists[s];
                        dist3 = dists[s + 1];
                    }
                const float den = __fmul_rn(2.0f, __fsub_rn(__fadd_rn(dist1, dist3), __fmul_rn(2.0f, dist2)));
                const float deltaR = __fdiv_rn(__fsub_rn(dist1, dist3), den);
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = __fmul_rn(g.sf[levelL], __fadd_rn(__fadd_rn(scaleduR0, (float)bestinc), deltaR));
                    float disparity = __fsub_rn(uL, bestuR);
                    if (disparity >= minD && disparity < maxD) {
                        if (disparity <= 0) {
                            disparity = 0.01f;
                            bestuR = (float)((double)uL - 0.01);
                        }
                        outD = __fdiv_rn(a.mbf, disparity);
                        outU = bestuR;
                        outSad = bestS;
                    }
                }
            }
        }
    }
    if (lane == 0) {
        a.uright[base + iL] = outU;
        a.depth[base + iL] = outD;
        a.sad[base + iL] = outSad;
        if (a.hamIdx) a.hamIdx[base + iL] = outHam;
    }
}

// Counting sort of the right keypoints of one pair by (int)y: rowStart[r] .. rowStart[r+1] index `order`.
// Keypoints outside [0, H) (cannot come from the extractor) are dropped, as the reference's unchecked
// vRowIndices access would be out of bounds for them.
__global__ __launch_bounds__(256) void k_stereo_rowsort(FtGeom g, FtStereoArgs a) {
    extern __shared__ __attribute__((aligned(16))) int sh[];  // H + 1 counters
    __shared__ int wsum[4];
    const int slot = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = g.lv[0].h;
    const int nR = a.nR[slot];
    const size_t base = (size_t)slot * a.capacity;
    for (int i = tid; i <= H; i += 256) sh[i] = 0;
    __syncthreads();
    for (int i = tid; i < nR; i += 256) {
        const int y = (int)a.keysR[base + i].y;
        if (y >= 0 && y < H) atomicAdd(&sh[y], 1);
    }
    __syncthreads();
    // exclusive scan of sh[0..H): each thread owns a contiguous chunk
    const int per = (H + 255) / 256;
    const int c0 = min(tid * per, H), c1 = min(c0 + per, H);
    int local = 0;
    for (int c = c0; c < c1; c++) local += sh[c];
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; w++) wbase += wsum[w];
    int run = wbase + incl - local;
    int *rs = a.rowStart + (size_t)slot * a.rowStride;
    for (int c = c0; c < c1; c++) {
        const int n = sh[c];
        rs[c] = run;
        sh[c] = run;  // becomes the scatter cursor
        run += n;
    }
    if (tid == 255) rs[H] = run;  // the last thread's chunk ends at H
    __syncthreads();
    FtSortedR *srt = a.sorted + base;
    for (int i = tid; i < nR; i += 256) {
        const ft_keypoint kp = a.keysR[base + i];
        const int y = (int)kp.y;
        if (y >= 0 && y < H) {
            FtSortedR e;
            e.x = kp.x;
            e.y = kp.y;
            e.octave = kp.octave;
            e.idx = i;
            const unsigned long long *d = (const unsigned long long *)(a.descR + (base + i) * 32);
            e.desc[0] = d[0]; e.desc[1] = d[1]; e.desc[2] = d[2]; e.desc[3] = d[3];
            srt[atomicAdd(&sh[y], 1)] = e;
        }
    }
}

// One WAVE per pair (no barriers, no atomics): the accepted SADs are pulled into registers once, the
// median (element size/2 of the sorted list) is found by a bitwise search on the value with ballot-free
// wave sums, then every match with SAD >= 1.5f*1.4f*median is dropped (src/Frame.cc:991-1004).
#define SM_PER_LANE 40  // 64 * 40 = 2560 left keypoints per pair; larger inputs take the looped path
__global__ __launch_bounds__(64) void k_stereo_median(FtStereoArgs a) {
    const int slot = blockIdx.x, lane = threadIdx.x;
    const int nL = a.nL[slot];
    const size_t base = (size_t)slot * a.capacity;
    const bool inRegs = nL <= 64 * SM_PER_LANE;
    int v[SM_PER_LANE];
    int m = 0;
    if (inRegs) {
#pragma unroll
        for (int k = 0; k < SM_PER_LANE; k++) {
            const int i = k * 64 + lane;
            v[k] = i < nL ? a.sad[base + i] : -1;
            m += v[k] >= 0 ? 1 : 0;
        }
    } else {
        for (int i = lane; i < nL; i += 64) m += a.sad[base + i] >= 0 ? 1 : 0;
    }
    const int total = wave_sum_i32(m);
    if (total == 0 || !a.applyMedianCut) {
        if (lane == 0) a.nMatches[slot] = total;
        return;
    }
    const int k = total / 2;  // 0-based rank in ascending order
    // smallest value with count(sad <= value) >= k + 1; SAD <= 121*255 < 2^15
    int lo = 0;
    for (int bit = 14; bit >= 0; bit--) {
        const int probe = lo + (1 << bit) - 1;  // is the answer <= probe ?
        int c = 0;
        if (inRegs) {
#pragma unroll
            for (int q = 0; q < SM_PER_LANE; q++) c += (v[q] >= 0 && v[q] <= probe) ? 1 : 0;
        } else {
            for (int i = lane; i < nL; i += 64) {
                const int s = a.sad[base + i];
                c += (s >= 0 && s <= probe) ? 1 : 0;
            }
        }
        if (wave_sum_i32(c) < k + 1) lo += 1 << bit;
    }
    const float median = (float)lo;
    const float thDist = __fmul_rn(1.5f * 1.4f, median);
    int removed = 0;
    for (int i = lane; i < nL; i += 64) {
        const int s = a.sad[base + i];
        if (s >= 0 && !((float)s < thDist)) {
            a.uright[base + i] = -1.f;
            a.depth[base + i] = -1.f;
            a.sad[base + i] = -1;
            removed++;
        }
    }
    removed = wave_sum_i32(removed);
    if (lane == 0) a.nMatches[slot] = total - removed;
}

// Brute-force 2-NN: one wave per query, lanes stride over the train set; the two smallest
// (distance, index) keys of the wave are the reference's (best, second) with earlier index first.
__global__ __launch_bounds__(256) void k_fisheye_2nn(const uint8_t *descL, int nL, const uint8_t *descR, int nR,
                                                     int *matches, int *bestOut, int *secondOut) {
    const int lane = threadIdx.x & 63, wave = wave_index();
    const int i = blockIdx.x * 4 + wave;
    if (i >= nL) return;
    unsigned long long q[4];
    {
        const unsigned long long *p = (const unsigned long long *)(descL + (size_t)i * 32);
        q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; q[3] = p[3];
    }
    unsigned k0 = 0xffffffffu, k1 = 0xffffffffu;  // two smallest keys seen by this lane
    for (int j = lane; j < nR; j += 64) {
        const unsigned key = ((unsigned)hamming256(q, (const unsigned long long *)(descR + (size_t)j * 32)) << 20) | (unsigned)j;
        if (key < k0) { k1 = k0; k0 = key; }
        else if (key < k1) k1 = key;
    }
    const unsigned m0 = wave_min_u32(k0);
    // second smallest overall: lanes whose k0 was the global minimum contribute their k1 instead
    const unsigned cand = (k0 == m0) ? k1 : k0;
    const unsigned m1 = wave_min_u32(cand);
    if (lane == 0) {
        const int d0 = (int)(m0 >> 20), d1 = (int)(m1 >> 20);
        int match = -1;
        if (nR >= 2 && (double)(float)d0 < (double)(float)d1 * 0.7) match = (int)(m0 & 0xfffffu);
        matches[i] = match;
        if (bestOut) bestOut[i] = nR >= 1 ? d0 : -1;
        if (secondOut) secondOut[i] = nR >= 2 ? d1 : -1;
    }
}

__global__ __launch_bounds__(256) void k_hamming_pairs(const uint8_t *a, const uint8_t *b, int n, int *dist) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long *pa = (const unsigned long long *)(a + (size_t)i * 32);
    unsigned long long q[4] = {pa[0], pa[1], pa[2], pa[3]};
    dist[i] = hamming256(q, (const unsigned long long *)(b + (size_t)i * 32));
}

// ------------------------------------------------------------------------------------------------
// KannalaBrandt8::TriangulateMatches (src/CameraModels/KannalaBrandt8.cpp:306-372) for the pairs that passed the
// ratio test: one thread per left keypoint.  Float steps follow the reference's order without contraction; the
// null vector of the 4x4 system comes from a one-sided Jacobi SVD in double (see fasttrack_amd.h).
// ------------------------------------------------------------------------------------------------
// Eigen's association of a three-term sum: e0 + (e1 + e2) (redux_novec_unroller; the oracle's note at orc_is_in_frustum)
__device__ __forceinline__ float fdot3(const float *a, const float *b) {
    return __fadd_rn(__fmul_rn(a[0], b[0]), __fadd_rn(__fmul_rn(a[1], b[1]), __fmul_rn(a[2], b[2])));
}
__device__ __forceinline__ float fnorm3(const float *a) { return sqrtf(fdot3(a, a)); }

__device__ void kb8_unproject(const float *cam, float precision, float px, float py, float r[3]) {
    const float pwx = __fdiv_rn(__fsub_rn(px, cam[2]), cam[0]), pwy = __fdiv_rn(__fsub_rn(py, cam[3]), cam[1]);
    float scale = 1.f;
    float theta_d = sqrtf(__fadd_rn(__fmul_rn(pwx, pwx), __fmul_rn(pwy, pwy)));
    theta_d = fminf(fmaxf((float)(-3.1415926535897932384626433832795 / 2.f), theta_d), (float)(3.1415926535897932384626433832795 / 2.f));
    if ((double)theta_d > 1e-8) {
        float theta = theta_d;
        for (int j = 0; j < 10; j++) {
            const float t2 = __fmul_rn(theta, theta), t4 = __fmul_rn(t2, t2), t6 = __fmul_rn(t4, t2), t8 = __fmul_rn(t4, t4);
            const float k0 = __fmul_rn(cam[4], t2), k1 = __fmul_rn(cam[5], t4), k2 = __fmul_rn(cam[6], t6), k3 = __fmul_rn(cam[7], t8);
            const float num = __fsub_rn(__fmul_rn(theta, __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(1.f, k0), k1), k2), k3)), theta_d);
            const float den = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(1.f, __fmul_rn(3.f, k0)), __fmul_rn(5.f, k1)), __fmul_rn(7.f, k2)),
                                        __fmul_rn(9.f, k3));
            const float fix = __fdiv_rn(num, den);
            theta = __fsub_rn(theta, fix);
            if (fabsf(fix) < precision) break;
        }
        scale = __fdiv_rn(ft_tan_f(theta), theta_d);
    }
    r[0] = __fmul_rn(pwx, scale);
    r[1] = __fmul_rn(pwy, scale);
    r[2] = 1.f;
}

__device__ void kb8_project(const float *cam, const float p[3], float uv[2]) {
    const float x2y2 = __fadd_rn(__fmul_rn(p[0], p[0]), __fmul_rn(p[1], p[1]));
    const float theta = ft_atan2_f(sqrtf(x2y2), p[2]);
    const float psi = ft_atan2_f(p[1], p[0]);
    const float t2 = __fmul_rn(theta, theta), t3 = __fmul_rn(theta, t2), t5 = __fmul_rn(t3, t2), t7 = __fmul_rn(t5, t2),
                t9 = __fmul_rn(t7, t2);
    const float r = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(theta, __fmul_rn(cam[4], t3)), __fmul_rn(cam[5], t5)), __fmul_rn(cam[6], t7)),
                              __fmul_rn(cam[7], t9));
    uv[0] = __fadd_rn(__fmul_rn(__fmul_rn(cam[0], r), ft_cos_f(psi)), cam[2]);
    uv[1] = __fadd_rn(__fmul_rn(__fmul_rn(cam[1], r), ft_sin_f(psi)), cam[3]);
}

// right singular vector of the smallest singular value of a 4x4 matrix: one-sided (Hestenes) Jacobi in double
__device__ void null_vector4(const double A[16], double v[4]) {
    double U[16], V[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        U[i] = A[i];
        V[i] = (i % 5 == 0) ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 30; sweep++) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = p + 1; q < 4; q++) {
                double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    alpha = __dadd_rn(alpha, __dmul_rn(U[4 * i + p], U[4 * i + p]));
                    beta = __dadd_rn(beta, __dmul_rn(U[4 * i + q], U[4 * i + q]));
                    gamma = __dadd_rn(gamma, __dmul_rn(U[4 * i + p], U[4 * i + q]));
                }
                if (fabs(gamma) <= __dmul_rn(1e-15, sqrt(__dmul_rn(alpha, beta))) || gamma == 0.0) continue;
                rotated = true;
                const double zeta = __ddiv_rn(__dsub_rn(beta, alpha), __dmul_rn(2.0, gamma));
                const double t = __ddiv_rn(zeta >= 0 ? 1.0 : -1.0, __dadd_rn(fabs(zeta), sqrt(__dadd_rn(1.0, __dmul_rn(zeta, zeta)))));
                const double c = __ddiv_rn(1.0, sqrt(__dadd_rn(1.0, __dmul_rn(t, t)))), sn = __dmul_rn(c, t);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const double up = U[4 * i + p], uq = U[4 * i + q];
                    U[4 * i + p] = __dsub_rn(__dmul_rn(c, up), __dmul_rn(sn, uq));
                    U[4 * i + q] = __dadd_rn(__dmul_rn(sn, up), __dmul_rn(c, uq));
                    const double vp = V[4 * i + p], vq = V[4 * i + q];
                    V[4 * i + p] = __dsub_rn(__dmul_rn(c, vp), __dmul_rn(sn, vq));
                    V[4 * i + q] = __dadd_rn(__dmul_rn(sn, vp), __dmul_rn(c, vq));
                }
            }
        if (!rotated) break;
    }
    int best = 0;
    double bn = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        double nj = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) nj = __dadd_rn(nj, __dmul_rn(U[4 * i + j], U[4 * i + j]));
        if (j == 0 || nj < bn) {
            bn = nj;
            best = j;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = best == 0 ? V[4 * i] : best == 1 ? V[4 * i + 1] : best == 2 ? V[4 * i + 2] : V[4 * i + 3];
}

__device__ float kb8_triangulate_matches(const FtFisheyeRig &g, float x1, float y1, float x2, float y2, float sigmaLevel, float unc,
                                         float p3D[3]) {
    float r1[3], r2[3];
    kb8_unproject(g.cam1, g.precision, x1, y1, r1);
    kb8_unproject(g.cam2, g.precision, x2, y2, r2);
    const float *R12 = g.Rlr, *t12 = g.tlr;
    float r21[3];
#pragma unroll
    for (int i = 0; i < 3; i++) r21[i] = fdot3(R12 + 3 * i, r2);
    const float cosParallaxRays = __fdiv_rn(fdot3(r1, r21), __fmul_rn(fnorm3(r1), fnorm3(r21)));
    if ((double)cosParallaxRays > 0.9998) return -1.f;
    float R21[9], t2[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) R21[3 * i + j] = R12[3 * j + i];
#pragma unroll
    for (int i = 0; i < 3; i++) t2[i] = -fdot3(R21 + 3 * i, t12);
    // rows of A: p.x * T.row(2) - T.row(0), p.y * T.row(2) - T.row(1) with Tcw1 = [I | 0], Tcw2 = [R21 | t2]
    const float T1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    float T2[12];
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) T2[4 * i + j] = R21[3 * i + j];
        T2[4 * i + 3] = t2[i];
    }
    double A[16], v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        A[j] = (double)__fsub_rn(__fmul_rn(r1[0], T1[8 + j]), T1[j]);
        A[4 + j] = (double)__fsub_rn(__fmul_rn(r1[1], T1[8 + j]), T1[4 + j]);
        A[8 + j] = (double)__fsub_rn(__fmul_rn(r2[0], T2[8 + j]), T2[j]);
        A[12 + j] = (double)__fsub_rn(__fmul_rn(r2[1], T2[8 + j]), T2[4 + j]);
    }
    null_vector4(A, v);
    const float h3 = (float)v[3];
    const float x3D[3] = {__fdiv_rn((float)v[0], h3), __fdiv_rn((float)v[1], h3), __fdiv_rn((float)v[2], h3)};
    const float z1 = x3D[2];
    if (z1 <= 0) return -2.f;
    const float z2 = __fadd_rn(fdot3(R21 + 6, x3D), T2[11]);
    if (z2 <= 0) return -3.f;
    float uv1[2];
    kb8_project(g.cam1, x3D, uv1);
    const float ex1 = __fsub_rn(uv1[0], x1), ey1 = __fsub_rn(uv1[1], y1);
    if ((double)__fadd_rn(__fmul_rn(ex1, ex1), __fmul_rn(ey1, ey1)) > 5.991 * (double)sigmaLevel) return -4.f;
    float x3D2[3];
#pragma unroll
    for (int i = 0; i < 3; i++) x3D2[i] = __fadd_rn(fdot3(R21 + 3 * i, x3D), t2[i]);
    float uv2[2];
    kb8_project(g.cam2, x3D2, uv2);
    const float ex2 = __fsub_rn(uv2[0], x2), ey2 = __fsub_rn(uv2[1], y2);
    if ((double)__fadd_rn(__fmul_rn(ex2, ex2), __fmul_rn(ey2, ey2)) > 5.991 * (double)unc) return -5.f;
    p3D[0] = x3D[0];
    p3D[1] = x3D[1];
    p3D[2] = x3D[2];
    return z1;
}

__global__ __launch_bounds__(64) void k_fisheye_triangulate(FtFisheyeRig rig, const ft_keypoint *keysL, int nL, const ft_keypoint *keysR,
                                                           int *matches, float *depth, float *p3d, int *nMatches) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= nL) return;
    const int j = matches[i];
    float d = -1.0f, p[3] = {0.f, 0.f, 0.f};
    int m = -1;
    if (j >= 0) {
        const ft_keypoint a = keysL[i], b = keysR[j];
        const float z = kb8_triangulate_matches(rig, a.x, a.y, b.x, b.y, rig.sigma2[a.octave], rig.sigma2[b.octave], p);
        if (z > 0.0001f) {  // Frame.cc:1263
            d = z;
            m = j;
            atomicAdd(nMatches, 1);
        } else {
            p[0] = p[1] = p[2] = 0.f;
        }
    }
    matches[i] = m;
    depth[i] = d;
    p3d[3 * i] = p[0];
    p3d[3 * i + 1] = p[1];
    p3d[3 * i + 2] = p[2];
}

// The same for every frame of a tracked batch (blockIdx.y = frame): the pairs k_fisheye_2nn_batch left in mvLeftToRightMatch
// (right-camera indices) are triangulated; a pair that fails loses its entry, a pair that stays enters mvRightToLeftMatch (the
// largest left index per right keypoint = the last writer of the reference's loop, src/Frame.cc:1262) and mvDepth / mvStereo3Dpoints
__global__ __launch_bounds__(256) void k_fisheye_triangulate_batch(const FtBatchJob *__restrict__ jobs, Rebase rb, FtBindArgs A) {
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const FtDevFrame &F = jobs[f].F;
    if (i >= F.Nleft) return;
    int *l2r = (int *)rb(F.l2r), *r2l = (int *)rb(F.r2l);
    const int j = l2r[i];
    float d = -1.0f, p[3] = {0.f, 0.f, 0.f};
    if (j >= 0) {
        const ft_keypoint a = rb(F.keys)[i], b = rb(F.keysR)[j];
        const float z = kb8_triangulate_matches(A.rig, a.x, a.y, b.x, b.y, A.rig.sigma2[a.octave], A.rig.sigma2[b.octave], p);
        if (z > 0.0001f) {  // Frame.cc:1263
            d = z;
            atomicMax(&r2l[j], i);
            atomicAdd(&A.nMatches[f], 1);
        } else {
            p[0] = p[1] = p[2] = 0.f;
            l2r[i] = -1;
        }
    }
    if (A.depth) {
        float *dep = rb(A.depth[f]), *pp = rb(A.p3d[f]);
        dep[i] = d;
        pp[3 * i] = p[0];
        pp[3 * i + 1] = p[1];
        pp[3 * i + 2] = p[2];
    }
}

}  // namespace

int ft_launch_fisheye_triangulate_batch(hipStream_t st, void *arena, const FtBatchJob *jobs, int nFrames, int maxKp, const FtBindArgs &A) {
    if (nFrames <= 0 || maxKp <= 0) return FT_OK;
    const Rebase rb{(uint8_t *)arena, (unsigned long long)(uintptr_t)arena};
    hipLaunchKernelGGL(k_fisheye_triangulate_batch, dim3((maxKp + 255) / 256, nFrames), dim3(256), 0, st, jobs, rb, A);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

namespace {
// grid (DL_BLOCKS, batch): the blocks of an image share the rows of its six result arrays dword by dword (every record
// size is a multiple of four bytes); the stores go over the host link, the counters with them
#define DL_BLOCKS 24
__global__ __launch_bounds__(256) void k_deliver(FtDeliverArgs a) {
    const int slot = blockIdx.y;
    const int nl = a.nL[slot], nr = a.nR ? a.nR[slot] : 0;  // one camera only: the right-hand pointers are null
    const int t = blockIdx.x * 256 + threadIdx.x, T = DL_BLOCKS * 256;
    auto rows = [&](const void *src, void *dst, int recBytes, int n) {
        if (!dst || n <= 0) return;
        const unsigned *s = (const unsigned *)((const uint8_t *)src + (size_t)slot * a.srcStride * recBytes);
        unsigned *d = (unsigned *)((uint8_t *)dst + (size_t)slot * a.dstStride * recBytes);
        const int words = n * (recBytes >> 2);
        for (int i = t; i < words; i += T) d[i] = s[i];
    };
    rows(a.keysL, a.oKeysL, (int)sizeof(ft_keypoint), nl);
    rows(a.descL, a.oDescL, 32, nl);
    rows(a.uright, a.oUright, 4, nl);
    rows(a.depth, a.oDepth, 4, nl);
    rows(a.keysR, a.oKeysR, (int)sizeof(ft_keypoint), nr);
    rows(a.descR, a.oDescR, 32, nr);
    if (t == 0) {
        a.oNL[slot] = nl;
        if (a.oNR) a.oNR[slot] = nr;
        if (a.oNMatches) a.oNMatches[slot] = a.nMatches[slot];
        if (slot == 0) {
            *a.oOverflowL = *a.overflowL;
            if (a.oOverflowR) *a.oOverflowR = *a.overflowR;
        }
    }
}
}  // namespace

namespace {
// The results of ft_extract_batch in the order ORBextractor::operator() returns them (src/ORBextractor.cc:1466-1487: keypoints
// outside the lapping area fill the output from the front, those inside it from the back), written straight into the caller's
// arrays in pinned host memory: a workgroup per image, the position of a keypoint from a running count of the lapping-area
// keypoints in front of it (ballots, wave totals through LDS).  Replaces the copy of whole rows into the library's staging and
// the host's pass over them (ft_extract_batch: 2 x 15 MB per 128 two-camera frames of 2 000 features, through the context's
// host threads).  An image with more keypoints than the caller's rows hold is left alone (the host reports FT_ERR_CAPACITY).
__global__ __launch_bounds__(256) void k_deliver_ordered(FtOrderedArgs a) {
    const int slot = a.b0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.nSel[slot];
    __shared__ int wcnt[2][4];
    if (n > a.capacity) {
        if (tid == 0) a.oMono[slot] = 0;
        return;
    }
    const unsigned *src = (const unsigned *)(a.keys + (size_t)slot * a.srcStride);
    const uint4 *sd = (const uint4 *)(a.desc + (size_t)slot * a.srcStride * 32);
    unsigned *dk = a.oKeys ? (unsigned *)(a.oKeys + (size_t)slot * a.capacity) : nullptr;
    uint4 *dd = a.oDesc ? (uint4 *)(a.oDesc + (size_t)slot * a.capacity * 32) : nullptr;
    constexpr int W = (int)sizeof(ft_keypoint) / 4;
    int lapBefore = 0;
    for (int base = 0, r = 0; base < n; base += 256, r ^= 1) {  // (uniform)
        const int i = base + tid;
        const bool real = i < n;
        unsigned kw[W];
#pragma unroll
        for (int k = 0; k < W; k++) kw[k] = real ? src[(size_t)i * W + k] : 0u;
        const float x = __uint_as_float(kw[0]);  // ft_keypoint::x is the first field
        const bool inLap = real && x >= a.lap0 && x <= a.lap1;
        const unsigned long long bal = __ballot(inLap);
        if (lane == 0) wcnt[r][wave] = __popcll(bal);
        __syncthreads();  // (two buffers: the next round's writes cannot pass a reader of this one)
        int below = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            below += w < wave ? wcnt[r][w] : 0;
            total += wcnt[r][w];
        }
        if (real) {
            const int rankLap = lapBefore + below + __popcll(bal & ((1ull << lane) - 1ull));
            const int dst = inLap ? n - 1 - rankLap : i - rankLap;
            if (dk) {
#pragma unroll
                for (int k = 0; k < W; k++) dk[(size_t)dst * W + k] = kw[k];
            }
            if (dd) {
                dd[2 * (size_t)dst] = sd[2 * (size_t)i];
                dd[2 * (size_t)dst + 1] = sd[2 * (size_t)i + 1];
            }
        }
        lapBefore += total;
    }
    if (tid == 0) a.oMono[slot] = n - lapBefore;
}

__global__ __launch_bounds__(256) void k_upload(const FtSrcEntry *srcTab, int width, int height, uint8_t *slot0, int pitch,
                                                size_t slotBytes, const uint8_t **l0Table) {
    const int slot = blockIdx.y;
    const FtSrcEntry e = srcTab[slot];  // (pinned host memory, written by the host just before the launch or replay)
    const uint8_t *src = e.ptr;
    const size_t sstride = (size_t)e.stride;
    uint8_t *dst = slot0 + (size_t)slot * slotBytes;
    const int t = blockIdx.x * 256 + threadIdx.x, T = gridDim.x * 256;
    if ((width & 3) == 0 && (((uintptr_t)src | sstride) & 3) == 0) {  // dword columns (the slot pitch is a multiple of 64 bytes)
        const int wq = width >> 2, n = wq * height;
        for (int i = t; i < n; i += T) {
            const int y = i / wq, x = i - y * wq;
            ((unsigned *)(dst + (size_t)y * pitch))[x] = ((const unsigned *)(src + (size_t)y * sstride))[x];
        }
    } else {
        const int n = width * height;
        for (int i = t; i < n; i += T) {
            const int y = i / width, x = i - y * width;
            dst[(size_t)y * pitch + x] = src[(size_t)y * sstride + x];
        }
    }
    if (t == 0) l0Table[slot] = dst;
}
}  // namespace

int ft_launch_upload(hipStream_t st, int batch, const FtSrcEntry *srcTab, int width, int height, uint8_t *slot0, int pitch,
                     size_t slotBytes, const uint8_t **l0Table) {
    hipLaunchKernelGGL(k_upload, dim3(48, batch), dim3(256), 0, st, srcTab, width, height, slot0, pitch, slotBytes, l0Table);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

namespace {
// the counters the host reads at the end of a wide batch - per-image totals, the overflow flag, the demand of the octree tiers
// behind k_octree (reset here for the next batch) - written to pinned host memory by ONE small kernel: they were six or seven
// 4-byte copies and two fills through the DMA queue, ~18 us of link latency each, one after the other at the tail of every
// ft_extract_batch / front-end batch
__global__ __launch_bounds__(256) void k_finish_counts(FtCountsArgs a) {
    for (int i = threadIdx.x; i < a.batch; i += 256) a.oNSel[i] = a.nSel[i];
    if (threadIdx.x == 0) *a.oOverflow = *a.overflow;
    if (a.bigCount && (int)threadIdx.x < a.nStreams) {
        const int k = threadIdx.x;
        a.oHist[k] = a.bigCount[4 * k + 2];
        a.oBig[k] = a.bigCount[4 * k + 3];
        a.bigCount[4 * k + 2] = 0;
        a.bigCount[4 * k + 3] = 0;
    }
}
}  // namespace

int ft_launch_finish_counts(hipStream_t st, const FtCountsArgs &a) {
    hipLaunchKernelGGL(k_finish_counts, dim3(1), dim3(256), 0, st, a);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_deliver_ordered(hipStream_t st, int nb, const FtOrderedArgs &a) {
    if (nb <= 0) return FT_OK;
    hipLaunchKernelGGL(k_deliver_ordered, dim3(nb), dim3(256), 0, st, a);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_deliver(hipStream_t st, int batch, const FtDeliverArgs &a) {
    hipLaunchKernelGGL(k_deliver, dim3(DL_BLOCKS, batch), dim3(256), 0, st, a);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_stereo_match(hipStream_t st, const FtGeom &g, int batch, const uint8_t *const *l0L,
                           const uint8_t *const *l0R, int l0pitchL, int l0pitchR, const uint8_t *pyrL,
                           const uint8_t *pyrR, const FtStereoArgs &a) {
    dim3 grid, block(256, 1, 1);
    const FtSlotGrid sg = ft_slot_grid((a.capacity + 3) / 4, batch, grid);
    for (int rep = ft_debug_repeat("stereo"); rep > 0; rep--)
    hipLaunchKernelGGL(k_stereo_match, grid, block, 0, st, g, l0L, l0R, l0pitchL, l0pitchR, pyrL, pyrR, a, sg);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_stereo_rowsort(hipStream_t st, const FtGeom &g, int batch, const FtStereoArgs &a) {
    for (int rep = ft_debug_repeat("rowsort"); rep > 0; rep--)
    hipLaunchKernelGGL(k_stereo_rowsort, dim3(batch), dim3(256), (size_t)(g.lv[0].h + 1) * sizeof(int), st, g, a);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_stereo_median(hipStream_t st, int batch, const FtStereoArgs &a) {
    hipLaunchKernelGGL(k_stereo_median, dim3(batch), dim3(64), 0, st, a);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_fisheye(hipStream_t st, const uint8_t *descL, int nL, const uint8_t *descR, int nR, int *matches,
                      int *best, int *second) {
    if (nL <= 0) return FT_OK;
    hipLaunchKernelGGL(k_fisheye_2nn, dim3((nL + 3) / 4), dim3(256), 0, st, descL, nL, descR, nR, matches, best, second);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_hamming_pairs(hipStream_t st, const uint8_t *a, const uint8_t *b, int n, int *dist) {
    if (n <= 0) return FT_OK;
    hipLaunchKernelGGL(k_hamming_pairs, dim3((n + 255) / 256), dim3(256), 0, st, a, b, n, dist);
    FT_HIP(hipGetLastError());
    return FT_OK;
}

int ft_launch_fisheye_triangulate(hipStream_t st, const FtFisheyeRig &rig, const ft_keypoint *keysL, int nL,
                                  const ft_keypoint *keysR, int *matches, float *depth, float *p3d, int *nMatches) {
    if (nL <= 0) return FT_OK;
    hipLaunchKernelGGL(k_fisheye_triangulate, dim3((nL + 63) / 64), dim3(64), 0, st, rig, keysL, nL, keysR, matches, depth, p3d, nMatches);
    FT_HIP(hipGetLastError());
    return FT_OK;
}
