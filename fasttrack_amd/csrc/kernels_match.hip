// HIP kernels of the Hamming matchers for gfx950 (wave64): 256-bit descriptors as 4 x uint64 with
// __popcll, wave-wide argmin on packed (distance, index) keys so ties resolve exactly like the
// reference's sequential "dist < bestDist" scans.
//   k_stereo_match   Frame::ComputeStereoMatches  (reference src/Frame.cc:835-989)
//   k_stereo_median  the 1.5*1.4*median SAD cut    (src/Frame.cc:991-1004)
//   k_fisheye_2nn    BFMatcher knnMatch(k=2)+ratio (src/Frame.cc:1231-1255)
//   k_hamming_pairs  ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:2256-2272)
#include "ft_internal.h"

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

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = min(v, (unsigned)__shfl_xor((int)v, o));
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One wave per left keypoint.
__global__ __launch_bounds__(256) void k_stereo_match(FtGeom g, const uint8_t *const *l0L, const uint8_t *const *l0R,
                                                      int l0pitchL, int l0pitchR, const uint8_t *pyrL,
                                                      const uint8_t *pyrR, FtStereoArgs a, FtSlotGrid sg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int slot, blk;
    if (!ft_slot_block(sg, slot, blk)) return;
    const int iL = blk * 4 + wave;
    const int nL = a.nL[slot];
    if (iL >= nL) return;
    const size_t base = (size_t)slot * a.capacity;
    float outU = -1.0f, outD = -1.0f;
    int outSad = -1, outHam = -1;
    const ft_keypoint kpL = a.keysL[base + iL];
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
    if (row >= 0 && row < nRows && !(maxU < 0)) {
        // right keypoints are bucketed by (int)y (k_stereo_rowsort); the band of keypoint iR covers `row` only
        // if |y_R - row| < r + 1 with r = 2 * sf[octave] <= 2 * max scale factor, so scanning the buckets
        // [row - reach, row + reach] sees every candidate of the reference's vRowIndices[row]; the exact band,
        // octave and disparity tests are applied to each of them.
        const int reach = (int)ceilf(2.0f * g.sf[g.nlevels - 1]) + 1;
        const int *rs = a.rowStart + (size_t)slot * a.rowStride;
        const int t0 = rs[max(row - reach, 0)], t1 = rs[min(row + reach + 1, nRows)];
        const int *ord = a.order + base;
        for (int t = t0 + lane; t < t1; t += 64) {
            const int iR = ord[t];
            const ft_keypoint kpR = a.keysR[base + iR];
            // row band of the right keypoint (Frame.cc:852-862): rows floor(y-r) .. ceil(y+r), r = 2*sf[octave]
            const float r = __fmul_rn(2.0f, g.sf[kpR.octave]);
            const int maxr = (int)ceilf(__fadd_rn(kpR.y, r));
            const int minr = (int)floorf(__fsub_rn(kpR.y, r));
            if (row < minr || row > maxr) continue;
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = hamming256(dL, (const unsigned long long *)(a.descR + (base + iR) * 32));
                best = min(best, ((unsigned)dist << 16) | (unsigned)iR);
            }
        }
    }
    best = wave_min_u32(best);
    const int bestDist = (int)(best >> 16);
    const int thOrbDist = (FT_TH_HIGH + FT_TH_LOW) / 2;
    // bestDist starts at TH_HIGH in the reference, so only dist < TH_HIGH ever registers; < thOrbDist is stricter
    if (best != 0xffffffffu && bestDist < thOrbDist) {
        const int bestIdxR = (int)(best & 0xffffu);
        outHam = bestIdxR;
        const float uR0 = a.keysR[base + bestIdxR].x;
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
            // 11x11 left patch: each lane owns pixels lane and lane+64 (< 121)
            const int p0 = lane, p1 = lane + 64;
            const int r0 = p0 / 11, c0 = p0 - r0 * 11, r1 = p1 / 11, c1 = p1 - r1 * 11;
            const int a0 = imL[(size_t)(yl0 + r0) * pitchL + xl0 + c0];
            const int a1 = p1 < 121 ? imL[(size_t)(yl0 + r1) * pitchL + xl0 + c1] : 0;
            int bestS = 0x7fffffff, bestinc = 0;
            float dists[11];
#pragma unroll
            for (int s = 0; s < 11; s++) {
                const int xr = xr00 + s;
                int d = abs(a0 - (int)imR[(size_t)(yl0 + r0) * pitchR + xr + c0]);
                if (p1 < 121) d += abs(a1 - (int)imR[(size_t)(yl0 + r1) * pitchR + xr + c1]);
                d = wave_sum_i32(d);
                dists[s] = (float)d;
                if (d < bestS) {
                    bestS = d;
                    bestinc = s - Ls;
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
    int *ord = a.order + base;
    for (int i = tid; i < nR; i += 256) {
        const int y = (int)a.keysR[base + i].y;
        if (y >= 0 && y < H) ord[atomicAdd(&sh[y], 1)] = i;
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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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

}  // namespace

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
