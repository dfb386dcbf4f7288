// mgm_wta.hip -- K4+K5+K6: ordered sum of the per-pass Lr volumes, over-count
// correction, winner-takes-all and (optionally) V-fit refinement, one wavefront
// per pixel.
//
//   S accumulation in pass order      mgm_core.cc:582-587   (S = ((0+L0)+L1)+...)
//   S -= (NDIR-1)*C, first finite     mgm_core.cc:592-609
//   strict minimum wins
//   subpixel_refinement_sgm + VfitMinimum   mgm_refine.h:51-68, refine.h:70-92
//
// Compiled with default floating point: S may hold NaN (inf - inf) and the
// refinement must propagate NaN/inf exactly as IEEE arithmetic does on the CPU.
#include <algorithm>
#include <cstdlib>

#include "mgm_device.h"

namespace mgm {

__device__ __forceinline__ bool finite_bits(float x)
{
    return (__builtin_bit_cast(unsigned, x) & 0x7f800000u) != 0x7f800000u;
}

// refine.h:70-92
__device__ __forceinline__ void vfit(float v0, float v1, float v2, float &v_min, float &x_min)
{
    if ((v1 > v0) && (v1 > v2)) {
        v_min = v1;
        x_min = 0.0f;
        return;
    }
    float slope = v2 - v1;
    if ((v2 - v1) < (v0 - v1)) slope = v0 - v1;
    x_min = (v0 - v2) / (2.0f * slope);
    v_min = v2 + (x_min - 1.0f) * slope;
}

// PPW slabs per wave and iteration: all of them are requested before the first is consumed.
// One KiB per stream per wave leaves HBM at ~4.7 TB/s, two at ~6 TB/s (tools/microbench/bw.hip).
// MAXD: upper bound of NDIR the instance is built for (4 or 8): with at most 4 directions twice the slabs fit the
// registers of a wave.
// SUB: pixels per slab.  A slab is the 64*LPL consecutive floats a wave loads per stream; with L = 64*LPL/SUB labels
// that is SUB consecutive pixels, each reduced by its own 64/SUB lanes -- 128 labels run as LPL = 4, SUB = 2 (16-byte
// loads, two arg-mins per butterfly) instead of LPL = 2, SUB = 1.  SUB > 1 needs EXACT and npix % SUB == 0.
template <int LPL, int PPW, bool EXACT, int MAXD = kMaxDirs, int SUB = 1>
__global__ void __launch_bounds__(256) k_wta(const WtaParams P)
{
    static_assert(SUB == 1 || EXACT, "several pixels per slab only without padding lanes");
    constexpr int LP = LPL * 64;
    constexpr int LANES = 64 / SUB;  // lanes per pixel
    __shared__ float sS[4][LP];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int L = P.L;          // label stride of a pixel (a multiple of 64/SUB... when EXACT: L * SUB == LP)
    const int Lr_ = P.Lreal;    // labels that exist
    constexpr bool exact = EXACT;  // no label slot is padding, loads need no guard
    const int Ls = SUB > 1 ? LP : L;          // floats per slab
    const int m0 = lane * LPL;                // this lane's offset in the slab
    const int sub = lane / LANES;             // its pixel of the slab
    const int o0 = (lane % LANES) * LPL;      // its first label
    const bool c8 = P.C8 && exact;  // compact costs: one byte per label (wave-uniform)
    const long long nslab = P.npix / SUB;
    // grid-stride over groups of PPW consecutive slabs
    for (long long g0 = ((long long)blockIdx.x * 4 + wv) * PPW; g0 < nslab; g0 += (long long)gridDim.x * 4 * PPW) {
        float c[PPW][LPL], l[MAXD][PPW][LPL];
#pragma unroll
        for (int u = 0; u < PPW; u++) {
            const long long g = g0 + u < nslab ? g0 + u : nslab - 1;
            if (c8 && P.cbytes == 2) {  // (two bytes per cost: absolute differences of colour pairs, ...)
                const unsigned short *q = reinterpret_cast<const unsigned short *>(P.C8) + g * Ls + m0;
#pragma unroll
                for (int k = 0; k < LPL; k++) c[u][k] = c16_decode(q[k]);
            } else if (c8) {
                const uint8_t *q = P.C8 + g * Ls + m0;
#pragma unroll
                for (int k = 0; k < LPL; k++) c[u][k] = c8_decode(q[k]);
            } else {
                const float *q = P.C + g * Ls + m0;
#pragma unroll
                for (int k = 0; k < LPL; k++) c[u][k] = (exact || m0 + k < L) ? q[k] : f_inf();
            }
        }
#pragma unroll
        for (int p = 0; p < MAXD; p++) {
            if (p < P.NDIR) {
#pragma unroll
                for (int u = 0; u < PPW; u++) {
                    const long long g = g0 + u < nslab ? g0 + u : nslab - 1;
                    const float *q = P.Lr + (long long)p * P.nvol + g * Ls + m0;
#pragma unroll
                    for (int k = 0; k < LPL; k++) l[p][u][k] = (exact || m0 + k < L) ? q[k] : f_inf();
                }
            }
        }
#pragma unroll
        for (int u = 0; u < PPW; u++) {
            if (g0 + u >= nslab) break;
            const long long pix = (g0 + u) * SUB + sub;  // (per lane group)
            // S = ((0 + L0) + L1) + ... in pass order (mgm_core.cc:582-587)
            float S[LPL];
#pragma unroll
            for (int k = 0; k < LPL; k++) S[k] = 0.0f;
#pragma unroll
            for (int p = 0; p < MAXD; p++) {
                if (p < P.NDIR) {
#pragma unroll
                    for (int k = 0; k < LPL; k++) S[k] = S[k] + l[p][u][k];
                }
            }
            if (P.FIX == 1) {
                const float f = (float)(P.NDIR - 1);
#pragma unroll
                for (int k = 0; k < LPL; k++) S[k] = S[k] - f * c[u][k];
            }
            // what a disparity outside the volume -- or, in a ragged volume, outside the pixel's own range -- holds in the
            // reference's S: never incremented (0), then the over-count term with C = +INF (mgm_core.cc:582-599)
            float vout = 0.0f;
            if (P.FIX == 1) vout = vout - (float)(P.NDIR - 1) * f_inf();
            if (P.clo) {
                const int cl = (int)P.clo[pix] - P.dmin, ch = (int)P.chi[pix] - P.dmin;
#pragma unroll
                for (int k = 0; k < LPL; k++)
                    if (o0 + k < cl || o0 + k > ch) S[k] = vout;
            }
            if (P.S) {
                float *q = P.S + pix * Lr_ + o0;
#pragma unroll
                for (int k = 0; k < LPL; k++)
                    if (o0 + k < Lr_) q[k] = S[k];
            }
            // first strict minimum among finite entries, ascending o
            float best = f_inf();
            int bi = 0x7fffffff;
            const bool windowed = P.wlo != nullptr;
            int wl = 0, wh = 0;  // window in label indices
            if (windowed) {
                wl = (int)P.wlo[pix] - P.dmin;  // Dvec(int min, int max) of allocate_costvolume: float -> int
                wh = (int)P.whi[pix] - P.dmin;
            }
#pragma unroll
            for (int k = 0; k < LPL; k++) {
                const float v = S[k];
                if (o0 + k < Lr_ && (!windowed || (o0 + k >= wl && o0 + k <= wh)) && finite_bits(v) && best > v) {
                    best = v;
                    bi = o0 + k;
                }
            }
#pragma unroll
            for (int d = LANES / 2; d >= 1; d >>= 1) {  // (xor partners stay inside the pixel's lane group)
                const float ov = __shfl_xor(best, d);
                const int oi = __shfl_xor(bi, d);
                if (ov < best || (ov == best && oi < bi)) {
                    best = ov;
                    bi = oi;
                }
            }
            if (windowed) {
                // window labels outside the volume, in scan order: below it, (the volume), above it
                if (finite_bits(vout)) {
                    if (wl < 0 && wl <= wh && !(best < vout)) {
                        best = vout;
                        bi = wl;
                    }
                    const int hi0 = wl > Lr_ ? wl : Lr_;
                    if (wh >= Lr_ && hi0 <= wh && vout < best) {
                        best = vout;
                        bi = hi0;
                    }
                }
            }
            float outv, outc = best;
            if (bi == 0x7fffffff) outv = __builtin_nanf("");  // the reference leaves minP uninitialised here
            else outv = (float)(bi + P.dmin);
            if (P.refine == 1 && !windowed) {  // (wave-uniform) every lane stages its part of S, each pixel reads its own
#pragma unroll
                for (int k = 0; k < LPL; k++) sS[wv][m0 + k] = S[k];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (bi != 0x7fffffff && bi - 1 >= 0 && bi + 2 <= Lr_ - 1) {  // mgm_refine.h:58
                    const float *sp = sS[wv] + sub * L;
                    const float v0 = sp[bi - 1], v1 = sp[bi], v2 = sp[bi + 1];
                    float vmin, dx;
                    vfit(v0, v1, v2, vmin, dx);
                    outv = (float)(bi + P.dmin) + dx;
                    outc = vmin;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next slab overwrites sS
            }
            if (lane % LANES == 0) {
                P.out[pix] = outv;
                P.outcost[pix] = outc;
            }
        }
    }
}

// 192 and 384 labels: the same work with 16-BYTE loads.  Three labels per lane are 12-byte pieces (k_wta<3>: 5.6 TB/s of
// reads where 16-byte pieces reach 6.4, round 2), six are a 16- and an 8-byte piece -- so here a wave takes a GROUP of
// NPX consecutive pixels that fills three 1-KiB slabs exactly (4 x 192 or 2 x 384 labels = 768 floats) and lane l of
// slab j holds the four floats 4*(64 j + l) .. of the group: chunk t = 64 j + l belongs to pixel t / (L/4) and starts at
// its label 4*(t % (L/4)) (a chunk never straddles pixels: L is a multiple of four).  Every lane therefore holds three
// chunks of up to three different pixels; the arg-min of pixel q is a wave-wide butterfly over what each lane holds OF
// THAT PIXEL.  Plain volumes only (no padding, no windows, no ragged ranges): everything else takes k_wta.
template <int L, int MAXD>
__global__ void __launch_bounds__(256) k_wta_q(const WtaParams P)
{
    static_assert(L == 192 || L == 384, "groups of 768 floats");
    constexpr int NPX = 768 / L;   // pixels per group
    constexpr int CPP = L / 4;     // chunks per pixel
    __shared__ float sS[4][768];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool c8 = P.C8 != nullptr;
    const long long ngroup = P.npix / NPX;
    int cpix[3], clab[3];  // per slab: this lane's chunk -> pixel of the group, first label
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int t = 64 * j + lane;
        cpix[j] = t / CPP;
        clab[j] = (t % CPP) * 4;
    }
    for (long long g = (long long)blockIdx.x * 4 + wv; g < ngroup; g += (long long)gridDim.x * 4) {
        float c[3][4], l[MAXD][3][4];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            if (c8 && P.cbytes == 2) {
                const uint2 w = *reinterpret_cast<const uint2 *>(P.C8 + (g * 768 + 256 * j + 4 * lane) * 2);
                c[j][0] = c16_decode(w.x & 65535u); c[j][1] = c16_decode(w.x >> 16);
                c[j][2] = c16_decode(w.y & 65535u); c[j][3] = c16_decode(w.y >> 16);
            } else if (c8) {
                const unsigned w = *reinterpret_cast<const unsigned *>(P.C8 + g * 768 + 256 * j + 4 * lane);
#pragma unroll
                for (int k = 0; k < 4; k++) c[j][k] = c8_decode((w >> (8 * k)) & 255u);
            } else {
                const float4 q = *reinterpret_cast<const float4 *>(P.C + g * 768 + 256 * j + 4 * lane);
                c[j][0] = q.x; c[j][1] = q.y; c[j][2] = q.z; c[j][3] = q.w;
            }
        }
#pragma unroll
        for (int p = 0; p < MAXD; p++)
            if (p < P.NDIR) {
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const float4 q = *reinterpret_cast<const float4 *>(P.Lr + (long long)p * P.nvol + g * 768 + 256 * j + 4 * lane);
                    l[p][j][0] = q.x; l[p][j][1] = q.y; l[p][j][2] = q.z; l[p][j][3] = q.w;
                }
            }
        // S = ((0 + L0) + L1) + ... in pass order, then the over-count term (mgm_core.cc:582-599)
        float S[3][4];
        const float f = (float)(P.NDIR - 1);
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float a = 0.0f;
#pragma unroll
                for (int p = 0; p < MAXD; p++)
                    if (p < P.NDIR) a = a + l[p][j][k];
                if (P.FIX == 1) a = a - f * c[j][k];
                S[j][k] = a;
            }
        if (P.S) {
#pragma unroll
            for (int j = 0; j < 3; j++) *reinterpret_cast<float4 *>(P.S + g * 768 + 256 * j + 4 * lane) = make_float4(S[j][0], S[j][1], S[j][2], S[j][3]);
        }
        // per chunk: first strict minimum among its finite entries, ascending label
        float cb[3];
        int ci[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            cb[j] = f_inf();
            ci[j] = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (finite_bits(S[j][k]) && cb[j] > S[j][k]) {
                    cb[j] = S[j][k];
                    ci[j] = clab[j] + k;
                }
        }
        if (P.refine == 1) {
#pragma unroll
            for (int j = 0; j < 3; j++) *reinterpret_cast<float4 *>(&sS[wv][256 * j + 4 * lane]) = make_float4(S[j][0], S[j][1], S[j][2], S[j][3]);
        }
#pragma unroll
        for (int q = 0; q < NPX; q++) {
            float best = f_inf();
            int bi = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < 3; j++)  // (a lane's chunks of one pixel lie in ascending label order over j)
                if (cpix[j] == q && (cb[j] < best)) {
                    best = cb[j];
                    bi = ci[j];
                }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const float ov = __shfl_xor(best, d);
                const int oi = __shfl_xor(bi, d);
                if (ov < best || (ov == best && oi < bi)) {
                    best = ov;
                    bi = oi;
                }
            }
            float outv, outc = best;
            if (bi == 0x7fffffff) outv = __builtin_nanf("");  // the reference leaves minP uninitialised here
            else outv = (float)(bi + P.dmin);
            if (P.refine == 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (bi != 0x7fffffff && bi - 1 >= 0 && bi + 2 <= L - 1) {  // mgm_refine.h:58
                    const float *sp = sS[wv] + q * L;
                    float vmin, dx;
                    vfit(sp[bi - 1], sp[bi], sp[bi + 1], vmin, dx);
                    outv = (float)(bi + P.dmin) + dx;
                    outc = vmin;
                }
            }
            if (lane == 0) {
                P.out[g * NPX + q] = outv;
                P.outcost[g * NPX + q] = outc;
            }
        }
        if (P.refine == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next group overwrites sS
    }
}

// Any label count (used beyond 2048 labels, where k_wta has no instance): one wavefront per pixel, the labels strided over
// the lanes.  Same arithmetic and the same rules as k_wta -- S = ((0 + L0) + L1) + ... in pass order, the over-count
// term, the first strict minimum among the finite entries of the pixel's window, V-fit on the winner's neighbours --
// with S recomputed for the three labels of the fit instead of being staged.  Nothing here is tuned: the reference's Dvec
// has no label limit (dvec.cc:60), and this keeps the library from having one.
__global__ void __launch_bounds__(256) k_wta_any(const WtaParams P)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int L = P.L, Lr_ = P.Lreal;
    const float f = (float)(P.NDIR - 1);
    float vout = 0.0f;  // what a disparity outside the volume / the pixel's own range holds in the reference's S
    if (P.FIX == 1) vout = vout - f * f_inf();
    for (long long pix = (long long)blockIdx.x * 4 + wv; pix < P.npix; pix += (long long)gridDim.x * 4) {
        int cl = 0, ch = Lr_ - 1;
        if (P.clo) {
            cl = (int)P.clo[pix] - P.dmin;
            ch = (int)P.chi[pix] - P.dmin;
        }
        const bool windowed = P.wlo != nullptr;
        int wl = 0, wh = 0;
        if (windowed) {
            wl = (int)P.wlo[pix] - P.dmin;
            wh = (int)P.whi[pix] - P.dmin;
        }
        auto S_at = [&](int o) {
            float a = 0.0f;
            for (int p = 0; p < P.NDIR; p++) a = a + P.Lr[(long long)p * P.nvol + pix * L + o];
            if (P.FIX == 1)
                a = a - f * (!P.C8 ? P.C[pix * L + o]
                                   : (P.cbytes == 2 ? c16_decode(reinterpret_cast<const unsigned short *>(P.C8)[pix * L + o]) : c8_decode(P.C8[pix * L + o])));
            if (o < cl || o > ch) a = vout;
            return a;
        };
        float best = f_inf();
        int bi = 0x7fffffff;
        for (int o = lane; o < Lr_; o += 64) {
            const float v = S_at(o);
            if (P.S) P.S[pix * Lr_ + o] = v;
            if ((!windowed || (o >= wl && o <= wh)) && finite_bits(v) && best > v) {
                best = v;
                bi = o;
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const float ov = __shfl_xor(best, d);
            const int oi = __shfl_xor(bi, d);
            if (ov < best || (ov == best && oi < bi)) {
                best = ov;
                bi = oi;
            }
        }
        if (windowed && finite_bits(vout)) {  // window labels outside the volume, in scan order: below it, (the volume), above it
            if (wl < 0 && wl <= wh && !(best < vout)) {
                best = vout;
                bi = wl;
            }
            const int hi0 = wl > Lr_ ? wl : Lr_;
            if (wh >= Lr_ && hi0 <= wh && vout < best) {
                best = vout;
                bi = hi0;
            }
        }
        float outv, outc = best;
        if (bi == 0x7fffffff) outv = __builtin_nanf("");  // the reference leaves minP uninitialised here
        else outv = (float)(bi + P.dmin);
        if (P.refine == 1 && !windowed && bi != 0x7fffffff && bi - 1 >= 0 && bi + 2 <= Lr_ - 1) {  // mgm_refine.h:58
            float vmin, dx;
            vfit(S_at(bi - 1), S_at(bi), S_at(bi + 1), vmin, dx);
            outv = (float)(bi + P.dmin) + dx;
            outc = vmin;
        }
        if (lane == 0) {
            P.out[pix] = outv;
            P.outcost[pix] = outc;
        }
    }
}

hipError_t launch_wta(const WtaParams &p, hipStream_t s)
{
    long long nb = (p.npix + 3) / 4;  // (an upper bound: waves take several pixels per iteration)
    static int per_cu = -1;  // MGM_HIP_WTA_WG_PER_CU=n overrides the grid bound (A/B timing)
    if (per_cu < 0) {
        per_cu = (int)tune_num("wta_wg_per_cu", 0);
        if (per_cu < 0) per_cu = 0;
    }
    static int packed = -1;  // MGM_HIP_WTA_PACKED=0: one pixel per slab also at 128 / 64 labels (A/B timing)
    if (packed < 0) packed = tune_num("wta_packed", 1) != 0;
    const bool use_packed = packed && p.Lreal == p.L && (p.L == 128 || p.L == 64) && p.npix % (256 / p.L) == 0;
    // A bounded grid (workgroups of 4 waves), grid-stride beyond it.  Measured at 1920x1080 (8 / 4 directions): one
    // pixel per slab is fastest at ~768 workgroups per CU (2.99 ms at 16 -> 2.73 ms: 6.4 TB/s, the read ceiling of the
    // part), several pixels per slab at ~128 (0.83 -> 0.80 ms); far larger grids lose again.
    const long long cap = (long long)(p.num_cu > 0 ? p.num_cu : 256) * (per_cu ? per_cu : (use_packed ? 128 : 768));
    if (nb > cap) nb = cap;
    const dim3 grid((unsigned)nb), block(256);
    static int wide4 = -1;  // MGM_HIP_WTA_WIDE4=0: the 8-direction instance also for NDIR <= 4 (A/B timing)
    if (wide4 < 0) wide4 = tune_num("wta_wide4", 1) != 0;
    if (p.L > kMaxLPL * 64) {  // beyond the widest k_wta instance
        hipLaunchKernelGGL(k_wta_any, grid, block, 0, s, p);
        return hipGetLastError();
    }
    static int quad = -1;  // MGM_HIP_WTA_QUAD=0: 192 / 384 labels on k_wta<3> / <6> (A/B timing)
    if (quad < 0) quad = tune_num("wta_quad", 1) != 0;
    if (quad && p.Lreal == p.L && (p.L == 192 || p.L == 384) && p.npix % (768 / p.L) == 0 && !p.wlo && !p.clo && p.refine <= 1) {
        long long nq = (p.npix / (768 / p.L) + 3) / 4;
        const long long capq = (long long)(p.num_cu > 0 ? p.num_cu : 256) * (per_cu ? per_cu : 256);
        if (nq > capq) nq = capq;
        const dim3 gq((unsigned)nq);
        if (p.L == 192) {
            if (p.NDIR <= 4) hipLaunchKernelGGL((k_wta_q<192, 4>), gq, block, 0, s, p);
            else hipLaunchKernelGGL((k_wta_q<192, kMaxDirs>), gq, block, 0, s, p);
        } else {
            if (p.NDIR <= 4) hipLaunchKernelGGL((k_wta_q<384, 4>), gq, block, 0, s, p);
            else hipLaunchKernelGGL((k_wta_q<384, kMaxDirs>), gq, block, 0, s, p);
        }
        return hipGetLastError();
    }
    // 128 and 64 labels: two / four pixels per 256-float slab (16-byte loads, one butterfly for all of them)
    if (use_packed) {
        if (p.L == 128) {
            if (p.NDIR <= 4) hipLaunchKernelGGL((k_wta<4, 4, true, 4, 2>), grid, block, 0, s, p);
            else hipLaunchKernelGGL((k_wta<4, 2, true, kMaxDirs, 2>), grid, block, 0, s, p);
        } else {
            if (p.NDIR <= 4) hipLaunchKernelGGL((k_wta<4, 4, true, 4, 4>), grid, block, 0, s, p);
            else hipLaunchKernelGGL((k_wta<4, 2, true, kMaxDirs, 4>), grid, block, 0, s, p);
        }
        return hipGetLastError();
    }
    switch (pass_lpl(p.L)) {
#define WTA_CASE(LPL, PPW)                                                                  \
    case LPL:                                                                               \
        if (p.L == 64 * LPL && p.NDIR <= 4 && wide4) hipLaunchKernelGGL((k_wta<LPL, 2 * PPW, true, 4>), grid, block, 0, s, p); \
        else if (p.L == 64 * LPL) hipLaunchKernelGGL((k_wta<LPL, PPW, true>), grid, block, 0, s, p);  \
        else hipLaunchKernelGGL((k_wta<LPL, 1, false>), grid, block, 0, s, p);               \
        break;
        WTA_CASE(1, 4)
        WTA_CASE(2, 4)
        WTA_CASE(3, 2)
        WTA_CASE(4, 3)
        WTA_CASE(6, 1)
        WTA_CASE(8, 1)
#undef WTA_CASE
        // 513..2048 labels: one pixel per wave and iteration, guarded loads
        // (768 and 1024 labels exactly: the second pass-kernel build takes them since round 4, with compact costs)
        case 12:
            if (p.L == 768) hipLaunchKernelGGL((k_wta<12, 1, true>), grid, block, 0, s, p);
            else hipLaunchKernelGGL((k_wta<12, 1, false>), grid, block, 0, s, p);
            break;
        case 16:
            if (p.L == 1024) hipLaunchKernelGGL((k_wta<16, 1, true>), grid, block, 0, s, p);
            else hipLaunchKernelGGL((k_wta<16, 1, false>), grid, block, 0, s, p);
            break;
        case 24: hipLaunchKernelGGL((k_wta<24, 1, false>), grid, block, 0, s, p); break;
        case 32: hipLaunchKernelGGL((k_wta<32, 1, false>), grid, block, 0, s, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// refine.h:40-68
__device__ __forceinline__ void parabolafit(const float (&v)[4], float &v_min, float &x_min)
{
    if (v[1] > v[0] && v[1] > v[2]) {
        x_min = 0;
        v_min = v[1];
        return;
    }
    const float c = v[1];
    const float b = (v[2] - v[0]) / 2;
    const float a = (v[2] - 2 * v[1] + v[0]) / 2;
    float x = -b / (2 * a);
    if (x > 1) x = 1;
    if (x < -1) x = -1;
    v_min = (a * x + b) * x + c;
    x_min = x;
}

// refine.h:6-38 (the doubling of a and b and the clamp of a are the reference's)
__device__ __forceinline__ void parabolafit_ocv(const float (&v)[4], float &v_min, float &x_min)
{
    if (v[1] > v[0] && v[1] > v[2]) {
        x_min = 0;
        v_min = v[1];
        return;
    }
    const float c = v[1];
    float b = (v[2] - v[0]) / 2;
    float a = (v[2] - 2 * v[1] + v[0]) / 2;
    a *= 2;
    b *= 2;
    a = a > 1.0 ? a : 1.0;
    float x = (-b + a) / (2 * a);
    if (x > 1) x = 1;
    if (x < -1) x = -1;
    v_min = (a * x + b) * x + c;
    x_min = x;
}

// refine.h:94-99: evaluated in double (the literals are doubles), narrowed on return
__device__ __forceinline__ float cubic_interp(const float (&p)[4], const float x)
{
    return p[1] + 0.5 * x *
                      (p[2] - p[0] + x * (2.0 * p[0] - 5.0 * p[1] + 4.0 * p[2] - p[3] + x * (3.0 * (p[1] - p[2]) + p[3] - p[0])));
}

// refine.h:102-145: p = {S[o-1], S[o], S[o+1], S[o+2]}; the cubic treats p[1]..p[2] as the unit interval
__device__ __forceinline__ void cubicfit(const float (&p)[4], float &out_pmin, float &out_xmin)
{
    float pmin, xmin;
    if (p[1] < p[2]) {
        pmin = p[1];
        xmin = 0.0;
    } else {
        pmin = p[2];
        xmin = 1.0;
    }
    const double a = 0.5 * 3.0 * (3.0 * (p[1] - p[2]) + p[3] - p[0]);
    const double b = 2.0 * p[0] - 5.0 * p[1] + 4.0 * p[2] - p[3];
    const double c = 0.5 * (p[2] - p[0]);
    const double discr = b * b - 4.0 * a * c;
    if (discr >= 0) {
        const double z1 = (-b + __builtin_sqrt(discr)) / (2.0 * a);
        const double z2 = (-b - __builtin_sqrt(discr)) / (2.0 * a);
        if (z1 > 0.0 && z1 < 1.0) {
            const float tmp = cubic_interp(p, z1);
            if (tmp < pmin) {
                pmin = tmp;
                xmin = z1;
            }
        }
        if (z2 > 0.0 && z2 < 1.0) {
            const float tmp = cubic_interp(p, z2);
            if (tmp < pmin) {
                pmin = tmp;
                xmin = z2;
            }
        }
    }
    out_pmin = pmin;
    out_xmin = xmin;
}

// K4-K6 on the RANGE-PROPORTIONAL layout (mgm_pass_rel.hip): one wavefront per pixel, one label slot per lane.  The same
// arithmetic and rules as k_wta / k_wta_any + k_refine on the dense hull -- S = ((0 + L0) + L1) + ... in pass order, the
// over-count term, the first strict minimum among the finite entries of the pixel's window scanned by rising disparity,
// the refinement gate of mgm_refine.h:58 on the window -- with "a disparity the pixel does not own" = `vout` (what the
// reference's S holds there: 0 - (NDIR-1)*INF, or 0 without the over-count fix) instead of a stored value.
// SPL, CB (round 6): label slots per lane (4 / 8: 64 / 128 slots per pixel) and bytes per cost code (1 / 2), as in k_pass_rel.
template <int SPL, int CB>
__global__ void __launch_bounds__(256) k_wta_rel(const WtaRelParams P)
{
    constexpr int SLOTS = 16 * SPL;
    // FOUR pixels per wave: a pixel's 64 slots on a row of 16 lanes, 4 per lane (16-byte loads; the first version had one slot per
    // lane, one pixel per wave: 4-byte loads and six LDS-crossbar rounds per pixel, 1.5 ms per 1920x1080 volume where its 4.4 GB ask for 0.7)
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, grp = lane >> 4;
    const float f = (float)(P.NDIR - 1);
    float vout = 0.0f;
    if (P.FIX == 1) vout = vout - f * f_inf();
    const bool vfin = finite_bits(vout);
    const long long nquads = (P.npix + 3) / 4;
    for (long long q4 = (long long)blockIdx.x * 4 + wv; q4 < nquads; q4 += (long long)gridDim.x * 4) {
        const long long pix0 = q4 * 4 + grp;
        const bool live = pix0 < P.npix;
        const long long pix = live ? pix0 : P.npix - 1;
        const int4 rec = reinterpret_cast<const int4 *>(P.base)[pix];  // (the pixel's record: disparity of slot 0, its own range)
        const int b = rec.x, lo = rec.y, hi = rec.z;
        const bool windowed = P.wlo != nullptr;
        const int wl = windowed ? (int)P.wlo[pix] : lo, wh = windowed ? (int)P.whi[pix] : hi;
        typedef float f4 __attribute__((ext_vector_type(4)));
        float a[SPL];
#pragma unroll
        for (int q = 0; q < SPL; q++) a[q] = 0.0f;
        for (int p = 0; p < P.NDIR; p++) {
#pragma unroll
            for (int h = 0; h < SPL / 4; h++) {
                const f4 t = reinterpret_cast<const f4 *>(P.Lr + (long long)p * P.nvol + pix * SLOTS + SPL * li)[h];
#pragma unroll
                for (int q = 0; q < 4; q++) a[4 * h + q] = a[4 * h + q] + t[q];
            }
        }
        if (P.FIX == 1) {
            constexpr int NWORD = SPL * CB / 4;
            unsigned cw[NWORD];
#pragma unroll
            for (int w = 0; w < NWORD; w++) cw[w] = reinterpret_cast<const unsigned *>(P.c8 + (pix * SLOTS + SPL * li) * CB)[w];
#pragma unroll
            for (int q = 0; q < SPL; q++)
                a[q] = a[q] - f * (CB == 1 ? c8_decode((cw[q / 4] >> (8 * (q % 4))) & 255u)
                                           : (CB == 2 ? c16_decode((cw[q / 2] >> (16 * (q % 2))) & 65535u) : __builtin_bit_cast(float, cw[q * CB / 4])));
        }
        float val[SPL];
        // (2) its own disparities inside the window: the first strict minimum by rising disparity = the smallest (value, disparity)
        float cb = f_inf();
        int ci = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < SPL; q++) {
            const int d = b + SPL * li + q;
            const bool own = d >= lo && d <= hi;
            val[q] = own ? a[q] : vout;
            if (own && d >= wl && d <= wh && finite_bits(val[q]) && val[q] < cb) {
                cb = val[q];
                ci = d;
            }
        }
        auto take = [&](float ov, int oi) {
            if (ov < cb || (ov == cb && oi < ci)) {
                cb = ov;
                ci = oi;
            }
        };
        // butterfly over the row of 16 lanes: quad swaps, then the two mirrors
        take(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cb), 0xB1, 0xf, 0xf, false)), __builtin_amdgcn_update_dpp(0, ci, 0xB1, 0xf, 0xf, false));
        take(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cb), 0x4E, 0xf, 0xf, false)), __builtin_amdgcn_update_dpp(0, ci, 0x4E, 0xf, 0xf, false));
        take(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cb), 0x141, 0xf, 0xf, false)), __builtin_amdgcn_update_dpp(0, ci, 0x141, 0xf, 0xf, false));
        take(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cb), 0x140, 0xf, 0xf, false)), __builtin_amdgcn_update_dpp(0, ci, 0x140, 0xf, 0xf, false));
        // (1) the window's disparities below the pixel's own range: all `vout`, the first of them is the candidate
        float best = f_inf();
        int bi = 0x7fffffff;
        if (wl < lo && wl <= wh && vfin) {
            best = vout;
            bi = wl;
        }
        if (ci != 0x7fffffff && best > cb) {
            best = cb;
            bi = ci;
        }
        // (3) the window's disparities above the own range
        const int hi0 = wl > hi + 1 ? wl : hi + 1;
        if (wh > hi && hi0 <= wh && vfin && vout < best) {
            best = vout;
            bi = hi0;
        }
        float outv, outc = best;
        if (bi == 0x7fffffff) outv = __builtin_nanf("");  // the reference leaves minP uninitialised here
        else outv = (float)bi;
        // the four values around the minimum, from the lanes of the row that hold them (fetched whether or not they are used: every
        // lane takes part in the crossbar reads)
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int sl = (bi == 0x7fffffff ? b : bi) - 1 + k - b;  // slot of that disparity (row-uniform)
            const int e = sl & (SPL - 1);
            float mine = val[0];
#pragma unroll
            for (int q = 1; q < SPL; q++) mine = e == q ? val[q] : mine;
            const float x = __shfl(mine, (lane & 48) | ((sl / SPL) & 15));
            v[k] = (sl >= 0 && sl < SLOTS) ? x : vout;
        }
        if (P.refine >= 1 && bi != 0x7fffffff && bi - 1 >= wl && bi + 2 <= wh) {  // mgm_refine.h:58 (S allocated over the window)
            float vmin = outc, dx = 0;
            if (P.refine == 1) vfit(v[0], v[1], v[2], vmin, dx);
            else if (P.refine == 2) parabolafit(v, vmin, dx);
            else if (P.refine == 3) cubicfit(v, vmin, dx);
            else parabolafit_ocv(v, vmin, dx);
            outv = (float)bi + dx;
            outc = vmin;
        }
        if (li == 0 && live) {
            P.out[pix] = outv;
            P.outcost[pix] = outc;
        }
    }
}
// The corrected aggregated volume mgm() returns (mgm_core.cc:426, 582-601), on the dense hull, from the range-proportional Lr volumes
// (round 6): one thread per (pixel, label).  A label the pixel owns: S = ((0 + L0) + L1) + ... in pass order, minus (NDIR - 1) C
// with the over-count fix; a label it does not own holds what the dense-hull kernels leave there (C = +INF: +INF, or INF - INF =
// NaN with the fix) -- the reference's S has no such label.
__global__ void __launch_bounds__(256) k_rel_S(const WtaRelParams P, long long n, int L, int dmin, float *__restrict__ S)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const long long pix = t / L;
    const int d = dmin + (int)(t - pix * L);
    const int4 rec = reinterpret_cast<const int4 *>(P.base)[pix];
    const float f = (float)(P.NDIR - 1);
    float s = f_inf();
    if (d >= rec.y && d <= rec.z) {
        const long long k = pix * P.slots + (d - rec.x);
        s = 0.0f;
        for (int p = 0; p < P.NDIR; p++) s = s + P.Lr[(long long)p * P.nvol + k];
        if (P.FIX == 1)
            s = s - f * (P.cb == 1 ? c8_decode((unsigned)P.c8[k])
                                   : (P.cb == 2 ? c16_decode((unsigned)reinterpret_cast<const uint16_t *>(P.c8)[k]) : reinterpret_cast<const float *>(P.c8)[k]));
    } else if (P.FIX == 1)
        s = s - f * f_inf();
    S[t] = s;
}
hipError_t launch_rel_S(const WtaRelParams &p, int L, int dmin, float *S, hipStream_t s)
{
    const long long n = p.npix * L;
    hipLaunchKernelGGL(k_rel_S, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, L, dmin, S);
    return hipGetLastError();
}
hipError_t launch_wta_rel(const WtaRelParams &p, hipStream_t s)
{
    const long long groups = (p.npix + 15) / 16;  // four waves of four pixels per block
    const long long cap = (long long)p.num_cu * 64;
    const dim3 grid((unsigned)std::max(1ll, std::min(groups, cap)));
    if (p.slots == 128) {
        if (p.cb == 4) hipLaunchKernelGGL((k_wta_rel<8, 4>), grid, dim3(256), 0, s, p);
        else if (p.cb == 2) hipLaunchKernelGGL((k_wta_rel<8, 2>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_wta_rel<8, 1>), grid, dim3(256), 0, s, p);
    } else {
        if (p.cb == 4) hipLaunchKernelGGL((k_wta_rel<4, 4>), grid, dim3(256), 0, s, p);
        else if (p.cb == 2) hipLaunchKernelGGL((k_wta_rel<4, 2>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_wta_rel<4, 1>), grid, dim3(256), 0, s, p);
    }
    return hipGetLastError();
}

// Stand-alone refinement on a materialised (corrected) S: one thread per pixel (subpixel_refinement_sgm,
// mgm_refine.h:40-70; method = index into its table: 1 vfit, 2 parabola, 3 cubic, 4 parabolaOCV).  With range
// images (wlo/whi, see WtaParams) the gate uses the pixel's window and disparities outside the volume read `vout`.
__global__ void __launch_bounds__(256) k_refine(const float *__restrict__ S, long long npix, int L, int dmin, int method,
                                                const float *__restrict__ wlo, const float *__restrict__ whi, float vout,
                                                float *__restrict__ out, float *__restrict__ outcost)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float minP = out[i];
    if (!(minP == minP)) return;  // NaN label (no finite S): undefined in the reference
    const int o = (int)minP;
    const int lo = wlo ? (int)wlo[i] : dmin, hi = whi ? (int)whi[i] : dmin + L - 1;
    if (o - 1 >= lo && o + 2 <= hi) {
        const float *Si = S + i * L;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int d = o - 1 + k - dmin;
            v[k] = (d >= 0 && d < L) ? Si[d] : vout;
        }
        float vmin = outcost[i], dx = 0;
        if (method == 1) vfit(v[0], v[1], v[2], vmin, dx);
        else if (method == 2) parabolafit(v, vmin, dx);
        else if (method == 3) cubicfit(v, vmin, dx);
        else parabolafit_ocv(v, vmin, dx);
        out[i] = (float)o + dx;
        outcost[i] = vmin;
    }
}

hipError_t launch_refine(const float *S, long long npix, int L, int dmin, int method, const float *wlo, const float *whi,
                         float vout, float *out, float *outcost, hipStream_t s)
{
    if (method < 1 || method > 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_refine, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, S, npix, L, dmin, method, wlo, whi, vout,
                       out, outcost);
    return hipGetLastError();
}

// ---- update_dmin_dmax (mgm.cc:120-158) + the two remove_nonfinite_values_Img calls that follow it (387-388) ------
// global finite minimum / maximum of the disparity map (image_minmax, img_tools.h:183-199): floats ordered
// through their bit patterns so that integer atomics can reduce them
__device__ __forceinline__ unsigned f2ord(float f)
{
    const unsigned u = __builtin_bit_cast(unsigned, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o)
{
    return __builtin_bit_cast(float, (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__global__ void __launch_bounds__(256) k_minmax_init(unsigned *mm)
{
    mm[0] = f2ord(__builtin_huge_valf());   // gmin = +INF
    mm[1] = f2ord(-__builtin_huge_valf());  // gmax = -INF
}
__global__ void __launch_bounds__(256) k_minmax(const float *__restrict__ u, long long n, unsigned *mm)
{
    float lo = __builtin_huge_valf(), hi = -__builtin_huge_valf();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = u[i];
        if (finite_bits(v)) {
            lo = v < lo ? v : lo;
            hi = v > hi ? v : hi;
        }
    }
    atomicMin(mm + 0, f2ord(lo));
    atomicMax(mm + 1, f2ord(hi));
}
__global__ void __launch_bounds__(256) k_update_ranges(const float *__restrict__ outoff, int nx, int ny, int slack, int r,
                                                       const unsigned *__restrict__ mm, float *__restrict__ dminI,
                                                       float *__restrict__ dmaxI)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)nx * ny) return;
    const int i = (int)(idx % nx), j = (int)(idx / nx);
    const float gmin = ord2f(mm[0]), gmax = ord2f(mm[1]);
    float dmin = __builtin_huge_valf(), dmax = -__builtin_huge_valf();
    for (int dj = -r; dj <= r; dj++)
        for (int di = -r; di <= r; di++) {
            int x = i + di, y = j + dj;  // valneumann
            x = x >= 0 ? x : 0;
            x = x < nx ? x : nx - 1;
            y = y >= 0 ? y : 0;
            y = y < ny ? y : ny - 1;
            const float v = outoff[x + (long long)y * nx];
            const float a = finite_bits(v) ? v - slack : gmin - slack;
            const float b = finite_bits(v) ? v + slack : gmax + slack;
            dmin = __builtin_fminf(dmin, a);
            dmax = __builtin_fmaxf(dmax, b);
        }
    float lo = dminI[idx], hi = dmaxI[idx];
    if (finite_bits(dmin)) {
        lo = dmin;
        hi = dmax;
    }
    // remove_nonfinite_values_Img(dminI, gmin), (dmaxI, gmax)
    dminI[idx] = finite_bits(lo) ? lo : gmin;
    dmaxI[idx] = finite_bits(hi) ? hi : gmax;
}

hipError_t launch_update_ranges(const float *outoff, int nx, int ny, int slack, int radius, float *dminI, float *dmaxI,
                                float *scratch2, hipStream_t s)
{
    unsigned *mm = reinterpret_cast<unsigned *>(scratch2);
    const long long n = (long long)nx * ny;
    if (slack < 0) slack = -slack;
    hipLaunchKernelGGL(k_minmax_init, dim3(1), dim3(1), 0, s, mm);
    long long nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_minmax, dim3((unsigned)nb), dim3(256), 0, s, outoff, n, mm);
    hipLaunchKernelGGL(k_update_ranges, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, outoff, nx, ny, slack, radius, mm, dminI,
                       dmaxI);
    return hipGetLastError();
}

}  // namespace mgm
