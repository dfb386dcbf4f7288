// mgm_cost_fast.hip -- K2, the restructured cost-volume kernels of the costs the hot paths use (round 4): single-word census
// (compact copy only, padded layouts), absolute / squared differences (the same), Birchfield-Tomasi, census over several
// descriptor words and differences without a compact form (fp32, four pixels per lane), clipped NCC (window statistics once
// per pixel).  launch_cost (mgm_cost.hip) tries launch_cost_fast first; the general kernel there takes whatever is left
// (ragged volumes, prefiltered NCC / Birchfield-Tomasi, label counts that are not multiples of four).
//
// Compiled with default (NaN-honouring) floating point, like mgm_cost.hip.
#include "mgm_cost_common.h"

namespace mgm {

// K2 for the case whose costs are known to fit the compact form (single-word census, trunc = +INF or an
// integer <= 254; see mgm_costvolume_build_dev): integer arithmetic only, the compact volume only.
// One wavefront per pixel; lane l owns the LPL consecutive labels l*LPL.. -- LPL bytes, one store.
//   cost = min(popcount(cu ^ cv), trunc), trunc for a hypothesis outside the right image
//   (mgm_costvolume.h:65-78, 401-412); a pixel without a finite cost is all zeros (414-421).
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
template <int LPL>
__global__ void __launch_bounds__(256) k_cost_census8(const uint32_t *__restrict__ cu, const uint32_t *__restrict__ cv,
                                                      int nx, int ny, int vnx, int vny, int dmin, unsigned tb,
                                                      uint8_t *__restrict__ C8, int Lreal)
{
    constexpr int L = LPL * 64;
    const long long npix = (long long)nx * ny;
    const int lane = threadIdx.x & 63;
    for (long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); pix < npix; pix += (long long)gridDim.x * 4) {
        const int x = (int)(pix % nx), y = (int)(pix / nx);
        const uint32_t wu = cu[pix];
        const int q0 = x + dmin + lane * LPL;
        const bool yin = y < vny;
        const uint32_t *row = cv + (long long)(yin ? y : 0) * vnx;
        unsigned b[LPL];
        if (yin && q0 >= 0 && q0 + LPL <= vnx && (lane + 1) * LPL <= Lreal) {  // the whole group lies inside the right image
            uint32_t wv[LPL];
            if constexpr (LPL % 4 == 0) {
#pragma unroll
                for (int h = 0; h < LPL / 4; h++) {
                    const u32x4_a4 t = *reinterpret_cast<const u32x4_a4 *>(row + q0 + 4 * h);
                    wv[4 * h] = t.x; wv[4 * h + 1] = t.y; wv[4 * h + 2] = t.z; wv[4 * h + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < LPL; k++) wv[k] = row[q0 + k];
            }
#pragma unroll
            for (int k = 0; k < LPL; k++) {
                const unsigned pc = (unsigned)__builtin_popcount(wu ^ wv[k]);
                b[k] = pc < tb ? pc : tb;
            }
        } else {
#pragma unroll
            for (int k = 0; k < LPL; k++) {
                const int q = q0 + k;
                const bool in = yin && q >= 0 && q < vnx;
                const unsigned pc = (unsigned)__builtin_popcount(wu ^ row[in ? q : 0]);
                b[k] = in ? (pc < tb ? pc : tb) : tb;
                if (lane * LPL + k >= Lreal) b[k] = 255u;  // a slot of the padded layout: +INF
            }
        }
        bool fin = false;
#pragma unroll
        for (int k = 0; k < LPL; k++) fin |= b[k] != 255u;
        if (__builtin_amdgcn_ballot_w64(fin) == 0ull) {  // no valid hypothesis: zeros (the slots of a padded layout stay +INF)
#pragma unroll
            for (int k = 0; k < LPL; k++) b[k] = lane * LPL + k >= Lreal ? 255u : 0u;
        }
        uint8_t *dst = C8 + pix * L + lane * LPL;
        if constexpr (LPL == 1) {
            dst[0] = (uint8_t)b[0];
        } else if constexpr (LPL == 2) {
            *reinterpret_cast<unsigned short *>(dst) = (unsigned short)(b[0] | (b[1] << 8));
        } else if constexpr (LPL % 4 == 0) {
#pragma unroll
            for (int h = 0; h < LPL / 4; h++)
                reinterpret_cast<unsigned *>(dst)[h] = b[4 * h] | (b[4 * h + 1] << 8) | (b[4 * h + 2] << 16) | (b[4 * h + 3] << 24);
        } else {
#pragma unroll
            for (int k = 0; k < LPL; k++) dst[k] = (uint8_t)b[k];
        }
    }
}

// The same costs for label counts that divide 1024, sixteen labels per lane: a wave writes 1 KiB = 1024 / L whole
// pixels per iteration with 16-byte stores (the 4-byte version above spends its time in per-pixel index arithmetic
// and load latency: one pixel per wave and iteration).
template <int L>
__global__ void __launch_bounds__(256) k_cost_census8w(const uint32_t *__restrict__ cu, const uint32_t *__restrict__ cv,
                                                       int nx, int ny, int vnx, int vny, int dmin, unsigned tb,
                                                       uint8_t *__restrict__ C8)
{
    static_assert(L == 64 || L == 128 || L == 256 || L == 512, "whole pixels per KiB");
    constexpr int LP = L / 16;    // lanes per pixel
    constexpr int PPC = 64 / LP;  // pixels per wave and iteration
    const long long npix = (long long)nx * ny;
    const long long nchunk = (npix + PPC - 1) / PPC;
    const int lane = threadIdx.x & 63, sub = lane / LP, part = lane % LP;
    const unsigned long long group = (LP == 64 ? ~0ull : ((1ull << (LP % 64)) - 1ull)) << (sub * LP);
    for (long long chunk = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); chunk < nchunk; chunk += (long long)gridDim.x * 4) {
        const long long pix = chunk * PPC + sub;
        const bool live = pix < npix;
        const unsigned p32 = live ? (unsigned)pix : 0u;  // (npix < 2^31: checked by the caller)
        const int y = (int)(p32 / (unsigned)nx), x = (int)(p32 - (unsigned)y * (unsigned)nx);
        const uint32_t wu = cu[p32];
        const int q0 = x + dmin + part * 16;
        const bool yin = y < vny;
        const uint32_t *row = cv + (long long)(yin ? y : 0) * vnx;
        unsigned w[4];
        bool fin = false;
        if (yin && q0 >= 0 && q0 + 16 <= vnx) {  // the lane's sixteen labels lie inside the right image
            u32x4_a4 t[4];
#pragma unroll
            for (int h = 0; h < 4; h++) t[h] = *reinterpret_cast<const u32x4_a4 *>(row + q0 + 4 * h);
#pragma unroll
            for (int h = 0; h < 4; h++) {
                const unsigned v[4] = {t[h].x, t[h].y, t[h].z, t[h].w};
                unsigned b[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const unsigned pc = (unsigned)__builtin_popcount(wu ^ v[k]);
                    b[k] = pc < tb ? pc : tb;
                    fin |= b[k] != 255u;
                }
                w[h] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
            }
        } else {
#pragma unroll
            for (int h = 0; h < 4; h++) {
                unsigned b[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int q = q0 + 4 * h + k;
                    const bool in = yin && q >= 0 && q < vnx;
                    const unsigned pc = (unsigned)__builtin_popcount(wu ^ row[in ? q : 0]);
                    b[k] = in ? (pc < tb ? pc : tb) : tb;
                    fin |= b[k] != 255u;
                }
                w[h] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
            }
        }
        const bool anyfinite = (__builtin_amdgcn_ballot_w64(fin) & group) != 0ull;  // of this pixel's labels
        if (live) {
            uint4 o;
            o.x = anyfinite ? w[0] : 0u; o.y = anyfinite ? w[1] : 0u; o.z = anyfinite ? w[2] : 0u; o.w = anyfinite ? w[3] : 0u;
            *reinterpret_cast<uint4 *>(C8 + pix * L + part * 16) = o;
        }
    }
}

// The same again with FOUR consecutive pixels of a row per lane (any image width: see W4; any compact
// label count -- at 192 / 384 labels a pixel group takes 12 / 24 lanes and the last 4 / 16 lanes of the wave idle): the sixteen
// labels of a lane slide along the right image by one word per pixel, so the four pixels share 19 census words where
// four separate lanes load 64 -- the kernel above is bound by those (L1-resident, unaligned) loads, not by its stores.
template <int L, bool W4>  // W4: the image width is a multiple of four (every group whole and 16-byte aligned); else the last
                            // group of a row holds fewer pixels and the loads / stores are guarded
__global__ void __launch_bounds__(256) k_cost_census8x(const uint32_t *__restrict__ cu, const uint32_t *__restrict__ cv,
                                                       int nx, int ny, int vnx, int vny, int dmin, unsigned tb,
                                                       uint8_t *__restrict__ C8, int Lreal)
{
    static_assert(L % 16 == 0 && L >= 16 && L <= 1024, "sixteen labels per lane");
    constexpr int LP = L / 16;    // lanes per pixel group
    constexpr int G = 64 / LP;    // groups of four pixels per wave and iteration (192 / 384 labels: 4 / 16 lanes of the wave idle)
    const int gpr = (nx + 3) / 4;  // groups of four pixels per row
    const long long ngrp = (long long)gpr * ny, nchunk = (ngrp + G - 1) / G;
    const int lane = threadIdx.x & 63, sub = lane / LP, part = lane % LP;
    const unsigned long long group = (LP == 64 ? ~0ull : ((1ull << (LP % 64)) - 1ull)) << ((sub * LP) & 63);
    for (long long chunk = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); chunk < nchunk; chunk += (long long)gridDim.x * 4) {
        const long long grp = chunk * G + sub;
        const bool live = sub < G && grp < ngrp;
        const unsigned g32 = live ? (unsigned)grp : 0u;  // (npix < 2^31: checked by the caller)
        const int y = (int)(g32 / (unsigned)gpr), x = (int)(g32 - (unsigned)y * (unsigned)gpr) * 4;  // x .. x+3: one row
        const long long pix0 = (long long)y * nx + x;
        const int nhere = W4 ? 4 : (nx - x < 4 ? nx - x : 4);  // pixels of this group that exist
        unsigned wu[4];
        if (W4) {
            const uint4 wu4 = *reinterpret_cast<const uint4 *>(cu + pix0);
            wu[0] = wu4.x; wu[1] = wu4.y; wu[2] = wu4.z; wu[3] = wu4.w;
        } else if (nhere == 4) {
            const u32x4_a4 wu4 = *reinterpret_cast<const u32x4_a4 *>(cu + pix0);
            wu[0] = wu4.x; wu[1] = wu4.y; wu[2] = wu4.z; wu[3] = wu4.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) wu[i] = cu[pix0 + (i < nhere ? i : 0)];
        }
        const int q0 = x + dmin + part * 16;
        const bool yin = y < vny;
        const uint32_t *row = cv + (long long)(yin ? y : 0) * vnx;
        unsigned w[4][4];
        bool fin[4] = {false, false, false, false};
        unsigned pad[4];  // what a pixel without a valid hypothesis gets: zeros, the slots of a padded layout +INF
#pragma unroll
        for (int h = 0; h < 4; h++) {
            pad[h] = 0u;
#pragma unroll
            for (int k = 0; k < 4; k++) pad[h] |= (part * 16 + 4 * h + k >= Lreal ? 255u : 0u) << (8 * k);
        }
        if (yin && q0 >= 0 && q0 + 20 <= vnx && (part + 1) * 16 <= Lreal) {  // every word the four pixels need lies inside the right image
            unsigned v[20];
#pragma unroll
            for (int h = 0; h < 5; h++) {
                const u32x4_a4 t = *reinterpret_cast<const u32x4_a4 *>(row + q0 + 4 * h);
                v[4 * h] = t.x; v[4 * h + 1] = t.y; v[4 * h + 2] = t.z; v[4 * h + 3] = t.w;
            }
            // (a bit count is at most 32: never the +INF code, and clipped only by a truncation below 32 -- wave-uniform)
            const bool clip = tb < 32u;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                fin[i] = true;
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    unsigned b[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) b[k] = (unsigned)__builtin_popcount(wu[i] ^ v[i + 4 * h + k]);
                    if (clip) {
#pragma unroll
                        for (int k = 0; k < 4; k++) b[k] = b[k] < tb ? b[k] : tb;
                    }
                    w[i][h] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    unsigned b[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int q = q0 + i + 4 * h + k;
                        const bool in = yin && q >= 0 && q < vnx;
                        const unsigned pc = (unsigned)__builtin_popcount(wu[i] ^ row[in ? q : 0]);
                        b[k] = in ? (pc < tb ? pc : tb) : tb;
                        if (part * 16 + 4 * h + k >= Lreal) b[k] = 255u;  // a slot of the padded layout: +INF
                        fin[i] |= b[k] != 255u;
                    }
                    w[i][h] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
                }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool anyfinite = (__builtin_amdgcn_ballot_w64(fin[i]) & group) != 0ull;  // of this pixel's labels
            if (live && i < nhere) {
                uint4 o;
                o.x = anyfinite ? w[i][0] : pad[0]; o.y = anyfinite ? w[i][1] : pad[1]; o.z = anyfinite ? w[i][2] : pad[2]; o.w = anyfinite ? w[i][3] : pad[3];
                *reinterpret_cast<uint4 *>(C8 + (pix0 + i) * L + part * 16) = o;
            }
        }
    }
}

// ---- absolute / squared differences, compact form only (round 4) ---------------------------------------------------
// computeC_AD / computeC_SD (mgm_costvolume.h:23-44) for a volume that is EXPECTED to fit the compact form (8-bit images:
// whole-number differences): only the compact copy is written, CB bytes per cost, and the flag word says afterwards whether
// every cost really had that form -- if one did not, mgm_costvolume_build_dev runs the general kernel, which writes the fp32
// volume.  Work layout of k_cost_census8x: four consecutive pixels of a row and sixteen labels per lane, the four pixels
// share NL + 3 samples of the right image per channel; one 16-byte store per lane and pixel (NL = 16 or 8 labels).
//   * The sum over the channels runs in channel order from 0, as there.  x = max(d, -d) enters as |d| (a source modifier):
//     the two differ in the sign of a zero or of a NaN only, and 0 + x, x * x and "NaN loses the comparison with truncDist"
//     hide both.
//   * truncDist is +INF or a non-negative number here (the caller checks), so min(e, truncDist) is v_min_f32: a NaN cost
//     becomes truncDist exactly as with the reference's comparison.
//   * Encoding: convert, and keep the largest cost and the largest fractional part of the lane's costs of a pixel -- only a
//     lane that saw a cost outside [0, LIM] or a fraction (labels
//     outside the right image with truncDist = +INF; volumes that will be filled again) takes the careful c8_encode /
//     c16_encode path.
// NCH = the channel count (1 or 3: the right-image samples of all channels are loaded up front), or 0 = any (channel loop
// outermost, 64 accumulators).
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
// four consecutive samples starting at p (a group of four pixels of one row; `n` < 4 of them exist at the end of a row whose
// width is not a multiple of four: the others repeat the first)
__device__ __forceinline__ void load4_row(const float *__restrict__ p, int n, float (&o)[4])
{
    if (n >= 4) {
        const f32x4_a4 w = *reinterpret_cast<const f32x4_a4 *>(p);
        o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = w.w;
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = p[i < n ? i : 0];
    }
}
template <int NL>
__device__ __forceinline__ void diff_load(const CostParams &P, int t, long long pix0, int nlive, long long npix, long long vpix, int y, bool yin, int q0,
                                          bool inside, float (&ut)[4], float (&vt)[NL + 4])
{
    load4_row(P.u + (long long)t * npix + pix0, nlive, ut);
    const float *row = P.v + (long long)t * vpix + (long long)(yin ? y : 0) * P.vnx;
    if (inside) {
#pragma unroll
        for (int h = 0; h < NL / 4 + 1; h++) {
            const f32x4_a4 w = *reinterpret_cast<const f32x4_a4 *>(row + q0 + 4 * h);
            vt[4 * h] = w.x; vt[4 * h + 1] = w.y; vt[4 * h + 2] = w.z; vt[4 * h + 3] = w.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NL + 3; k++) {
            const int q = q0 + k;
            vt[k] = row[(yin && q >= 0 && q < P.vnx) ? q : 0];
        }
        vt[NL + 3] = 0.0f;
    }
}
template <int CB, int NCH, bool SD>
__global__ void __launch_bounds__(256) k_cost_diffx(const CostParams P)
{
    constexpr unsigned LIM = CB == 2 ? 65534u : 254u;  // the largest finite code
    constexpr int NC = NCH ? NCH : 1;
    constexpr int NL = 16 / CB;      // labels per lane: sixteen bytes, one store per pixel
    const int L = P.L, LP = L / NL;  // lanes per group of four pixels
    const int G = 64 / LP;           // groups per wave and iteration (192 / 384 / 768 labels: the last lanes of the wave idle)
    const int nx = P.nx, vnx = P.vnx;
    const long long npix = (long long)nx * P.ny, vpix = (long long)vnx * P.vny;
    const int gpr = (nx + 3) / 4;  // groups of four pixels per row (the last one of a row may hold fewer)
    const long long ngrp = (long long)gpr * P.ny, nchunk = (ngrp + G - 1) / G;
    const int lane = threadIdx.x & 63, sub = lane / LP, part = lane - sub * LP;
    const unsigned long long group = (LP == 64 ? ~0ull : ((1ull << (LP & 63)) - 1ull)) << ((sub * LP) & 63);
    const float trunc = P.trunc, tclamp = __builtin_fminf(trunc, (float)(LIM + 2u));
    const bool padlane = (part + 1) * NL > P.Lreal;  // this lane holds label slots of a padded layout (P.Lreal < P.L)
    bool odd = false, hopeless = CB == 2;  // a cost without the compact form of this width / of either width
    for (long long chunk = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); chunk < nchunk; chunk += (long long)gridDim.x * 4) {
        const long long grp = chunk * G + sub;
        const bool live = sub < G && grp < ngrp;
        const unsigned g32 = live ? (unsigned)grp : 0u;  // (npix < 2^31: checked by the caller)
        const int y = (int)(g32 / (unsigned)gpr), x = (int)(g32 - (unsigned)y * (unsigned)gpr) * 4;  // x .. x+3: one row
        const long long pix0 = (long long)y * nx + x;
        const int nlive = live ? (nx - x < 4 ? nx - x : 4) : 0, nload = nx - x < 4 ? nx - x : 4;
        const int q0 = x + P.dmin + part * NL;
        const bool yin = y < P.vny;
        const bool inside = yin && q0 >= 0 && q0 + NL + 4 <= vnx;  // every sample the four pixels need lies inside the right image
        float uu[NC][4], v[NC][NL + 4];
        float e[NCH ? 1 : 4][NL];
        if constexpr (NCH != 0) {
#pragma unroll
            for (int t = 0; t < NCH; t++) diff_load<NL>(P, t, pix0, nload, npix, vpix, y, yin, q0, inside, uu[t], v[t]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int k = 0; k < NL; k++) e[i][k] = 0.0f;
            for (int t = 0; t < P.nch; t++) {
                diff_load<NL>(P, t, pix0, nload, npix, vpix, y, yin, q0, inside, uu[0], v[0]);
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int k = 0; k < NL; k++) {
                        const float d = uu[0][i] - v[0][i + k];
                        e[i][k] += SD ? d * d : __builtin_fabsf(d);
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float c[NL];
            unsigned b[NL];
            float top = 0.0f, frac = 0.0f;  // the largest clamped cost and the largest fractional part
#pragma unroll
            for (int k = 0; k < NL; k++) {
                if constexpr (NCH != 0) {
                    float a = 0.0f;  // (0 + x: the compiler drops it; x >= +0)
#pragma unroll
                    for (int t = 0; t < NCH; t++) {
                        const float d = uu[t][i] - v[t][i + k];
                        a += SD ? d * d : __builtin_fabsf(d);
                    }
                    c[k] = a;
                } else {
                    c[k] = e[i][k];
                }
                if (!inside) {  // a label outside the right image costs truncDist (mgm_costvolume.h:401-412)
                    const int q = q0 + i + k;
                    c[k] = (yin && q >= 0 && q < vnx) ? c[k] : trunc;
                }
                float cc = __builtin_fminf(c[k], tclamp);  // in [0, LIM + 2]: the conversion is defined
                if (padlane && part * NL + k >= P.Lreal) cc = (float)(LIM + 1u);  // a slot of the padded layout: +INF, as its code
                b[k] = (unsigned)cc;
                top = __builtin_fmaxf(top, cc);
                frac = __builtin_fmaxf(frac, __builtin_amdgcn_fractf(cc));
            }
            bool fin = true;
            if (frac > 0.0f || top > (float)LIM) {  // the careful path
                fin = false;
#pragma unroll
                for (int k = 0; k < NL; k++) {
                    const float ct = (padlane && part * NL + k >= P.Lreal) ? __builtin_huge_valf() : ((c[k] < trunc) ? c[k] : trunc);
                    fin |= finite_bits(ct);
                    b[k] = CB == 2 ? c16_encode(ct) : c8_encode(ct);
                    odd |= b[k] > LIM + 1u && i < nlive;  // (i >= nlive: not a pixel of the image, see load4_row)
                    if (CB == 1) hopeless |= c16_encode(ct) > 65535u;  // (... nor in two bytes)
                }
            }
            // no valid hypothesis for this pixel => all labels cost 0 (mgm_costvolume.h:414-421)
            const bool anyfinite = (__builtin_amdgcn_ballot_w64(fin) & group) != 0ull;
            if (i >= nlive) continue;
            uint8_t *dst = P.C8 + ((pix0 + i) * L + part * NL) * CB;
            if constexpr (CB == 2) {
                unsigned w[4];
#pragma unroll
                for (int k = 0; k < 4; k++)
                    w[k] = anyfinite ? ((b[2 * k] & 65535u) | (b[2 * k + 1] << 16))
                                     : ((padlane && part * NL + 2 * k >= P.Lreal ? 65535u : 0u) | (padlane && part * NL + 2 * k + 1 >= P.Lreal ? 65535u << 16 : 0u));
                reinterpret_cast<uint4 *>(dst)[0] = make_uint4(w[0], w[1], w[2], w[3]);
            } else {
                unsigned w[4];
#pragma unroll
                for (int k = 0; k < 4; k++)
                    w[k] = anyfinite ? ((b[4 * k] & 255u) | ((b[4 * k + 1] & 255u) << 8) | ((b[4 * k + 2] & 255u) << 16) | (b[4 * k + 3] << 24))
                                     : ((padlane && part * NL + 4 * k >= P.Lreal ? 255u : 0u) | (padlane && part * NL + 4 * k + 1 >= P.Lreal ? 255u << 8 : 0u) |
                                        (padlane && part * NL + 4 * k + 2 >= P.Lreal ? 255u << 16 : 0u) | (padlane && part * NL + 4 * k + 3 >= P.Lreal ? 255u << 24 : 0u));
                reinterpret_cast<uint4 *>(dst)[0] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
    }
    // flag bit 0: some cost has no compact form of this width; bit 3: ... and a wider one would not help either
    if (__builtin_amdgcn_ballot_w64(odd) != 0ull && lane == 0) flag_once(P.bad8, 1u);
    if (__builtin_amdgcn_ballot_w64(odd && hopeless) != 0ull && lane == 0) flag_once(P.bad8, 8u);
}
template <int CB, bool SD>
static void launch_diffx(const CostParams &p, hipStream_t s)
{
    long long nw = ((long long)((p.nx + 3) / 4) * p.ny * 4 * p.L * CB / 4096 + 3) / 4 + 1;
    if (nw > 256 * 32) nw = 256 * 32;
    const dim3 grid((unsigned)nw), block(256);
    if (p.nch == 1) hipLaunchKernelGGL((k_cost_diffx<CB, 1, SD>), grid, block, 0, s, p);
    else if (p.nch == 3) hipLaunchKernelGGL((k_cost_diffx<CB, 3, SD>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_cost_diffx<CB, 0, SD>), grid, block, 0, s, p);
}

// ---- Birchfield-Tomasi costs, restructured (round 4) --------------------------------------------------------------------
// computeC_BTAD / computeC_BTSD (mgm_costvolume.h:82-135) look at three samples of each image per cell -- but the interval a
// sample spans depends on its own image alone: k_bt_spans writes the two ends once per sample (2*nch planes per image, same
// operations as bt_span above), and k_cost_btx is left with two three-way maxima and a minimum per cell and channel.  Work
// layout of k_cost_diffx: a wave takes four consecutive pixels of a row, a lane four consecutive labels of them (and the
// next 256 labels in its next turn): one 16-byte store of fp32 costs per lane and pixel -- the costs are multiples of one
// half, there is no compact form for them.
__global__ void __launch_bounds__(256) k_bt_spans(const float *__restrict__ u, int nx, int ny, int nch, float *__restrict__ sp)
{
    const long long npix = (long long)nx * ny;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * nch) return;
    const int t = (int)(idx / npix);
    const long long p = idx - t * npix;
    const int y = (int)(p / nx), x = (int)(p - (long long)y * nx);
    const BtSpan a = bt_span(u + t * npix + (long long)y * nx, nx, x);
    sp[idx] = a.lo;
    sp[idx + npix * nch] = a.hi;
}
// FN = the cost function (CostParams::costfn): 4 / 5 Birchfield-Tomasi as described; 0 / 1 absolute / squared differences and 2
// census over several descriptor words, for the volumes of those that have no compact form (float-valued or blurred
// images, costs that are thirds or halves of bit counts) and used to take the general kernel: the same layout, fp32 out.
template <int FN, bool W4>  // W4: the image width is a multiple of four (every group is whole: no guarded loads and stores)
__global__ void __launch_bounds__(256) k_cost_btx(const CostParams P)
{
    constexpr bool BT = FN >= 4, SD = FN == 5 || FN == 1;
    const int nx = P.nx, vnx = P.vnx, L = P.L, nch = P.nch;
    const long long npix = (long long)nx * P.ny, vpix = (long long)vnx * P.vny;
    const int gpr = (nx + 3) / 4;  // groups of four pixels per row (the last one of a row may hold fewer)
    const long long ngroup = (long long)gpr * P.ny;
    const int lane = threadIdx.x & 63;
    const float trunc = P.trunc;
    bool nanv = false;
    for (long long grp = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); grp < ngroup; grp += (long long)gridDim.x * 4) {
        const int y = (int)(grp / gpr), x = (int)(grp - (long long)y * gpr) * 4;  // x .. x+3: one row
        const long long pix0 = (long long)y * nx + x;
        const int nlive = W4 ? 4 : (nx - x < 4 ? nx - x : 4);
        const bool yin = y < P.vny;
        bool fin[4] = {false, false, false, false};
        for (int o0 = lane * 4; o0 < L; o0 += 256) {
            const int q0 = x + P.dmin + o0;
            const bool inside = yin && q0 >= 0 && q0 + 8 <= vnx;  // every sample the four pixels need lies inside the right image
            float e[4][4];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int k = 0; k < 4; k++) e[i][k] = 0.0f;
            for (int t = 0; t < nch; t++) {
                float ac[4], al[4] = {}, ah[4] = {};
                load4_row(P.u + (long long)t * npix + pix0, nlive, ac);
                if constexpr (BT) {
                    load4_row(P.ncc_u + (long long)t * npix + pix0, nlive, al);
                    load4_row(P.ncc_u + (long long)(nch + t) * npix + pix0, nlive, ah);
                }
                const long long rowoff = (long long)(yin ? y : 0) * vnx;
                const float *rc = P.v + (long long)t * vpix + rowoff;
                const float *rl = BT ? P.ncc_v + (long long)t * vpix + rowoff : rc, *rh = BT ? P.ncc_v + (long long)(nch + t) * vpix + rowoff : rc;
                float bc[8], bl[8] = {}, bh[8] = {};
                if (inside) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const f32x4_a4 c4 = *reinterpret_cast<const f32x4_a4 *>(rc + q0 + 4 * h);
                        bc[4 * h] = c4.x; bc[4 * h + 1] = c4.y; bc[4 * h + 2] = c4.z; bc[4 * h + 3] = c4.w;
                        if constexpr (BT) {
                            const f32x4_a4 l4 = *reinterpret_cast<const f32x4_a4 *>(rl + q0 + 4 * h);
                            const f32x4_a4 h4 = *reinterpret_cast<const f32x4_a4 *>(rh + q0 + 4 * h);
                            bl[4 * h] = l4.x; bl[4 * h + 1] = l4.y; bl[4 * h + 2] = l4.z; bl[4 * h + 3] = l4.w;
                            bh[4 * h] = h4.x; bh[4 * h + 1] = h4.y; bh[4 * h + 2] = h4.z; bh[4 * h + 3] = h4.w;
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 7; k++) {
                        const int q = q0 + k;
                        const int qq = (yin && q >= 0 && q < vnx) ? q : 0;
                        bc[k] = rc[qq];
                        if constexpr (BT) {
                            bl[k] = rl[qq];
                            bh[k] = rh[qq];
                        }
                    }
                    bc[7] = 0.0f;
                }
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if constexpr (BT) {
                            const float a_to_b = tri_high(0.0f, ac[i] - bh[i + k], bl[i + k] - ac[i]);
                            const float b_to_a = tri_high(0.0f, bc[i + k] - ah[i], al[i] - bc[i + k]);
                            const float r = __builtin_fabsf(a_to_b < b_to_a ? a_to_b : b_to_a);
                            e[i][k] += SD ? r * r : r;
                        } else if constexpr (FN == 2) {  // the samples are descriptor words (mgm_costvolume.h:65-78)
                            e[i][k] += (float)__builtin_popcount(__builtin_bit_cast(unsigned, ac[i]) ^ __builtin_bit_cast(unsigned, bc[i + k]));
                        } else {  // computeC_AD / computeC_SD (23-44)
                            float d = ac[i] - bc[i + k];
                            d = (d > -d) ? d : -d;
                            e[i][k] += SD ? d * d : d;
                        }
                    }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float c[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int q = q0 + i + k;
                    float v = e[i][k];
                    if constexpr (FN == 2) v = (float)((double)v * 1.0 / (double)nch);
                    c[k] = (inside || (yin && q >= 0 && q < vnx)) ? v : trunc;  // outside the right image: truncDist (401-412)
                    c[k] = (c[k] < trunc) ? c[k] : trunc;
                    fin[i] |= finite_bits(c[k]);
                    nanv |= c[k] != c[k];
                }
                if (i < nlive) *reinterpret_cast<float4 *>(P.C + (pix0 + i) * L + o0) = make_float4(c[0], c[1], c[2], c[3]);
            }
        }
        // no valid hypothesis for this pixel => all labels cost 0 (mgm_costvolume.h:414-421)
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (__builtin_amdgcn_ballot_w64(fin[i]) == 0ull && i < nlive)
                for (int o0 = lane * 4; o0 < L; o0 += 256) *reinterpret_cast<float4 *>(P.C + (pix0 + i) * L + o0) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (P.bad8 && __builtin_amdgcn_ballot_w64(nanv) != 0ull && lane == 0) flag_once(P.bad8, 2u);
}

// ---- clipped NCC, restructured (round 4) ---------------------------------------------------------------------------
// computeC_clippedNCC (mgm_costvolume.h:137-165) accumulates five window sums per (pixel, label, channel) -- but mu1 and s1
// depend on the left pixel alone and mu2, s2 on the right pixel alone: only the cross term is per cell.  Each sum is a
// sequential fp32 accumulation over the window in the reference's (i outer, j inner) order, so computing it ONCE per pixel
// in that order gives the same bits as computing it per label; likewise s - mu*mu (one rounded product, one rounded
// difference).  k_ncc_stats does that for both images (and notes whether the window lies inside the image and is NaN-free:
// otherwise the reference returns INFINITY whatever the other window holds); k_cost_ncc then needs 25 products per cell
// instead of 125 operations and 50 loads, with the rows of both images staged in LDS.
__global__ void __launch_bounds__(256) k_ncc_stats(const float *__restrict__ u, int nx, int ny, int nch, int hw, float *__restrict__ st)
{
    const long long npix = (long long)nx * ny;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix) return;
    const int x = (int)(idx % nx), y = (int)(idx / nx);
    bool ok = x - hw >= 0 && y - hw >= 0 && x + hw < nx && y + hw < ny;
    for (int t = 0; t < nch; t++) {
        float mu = 0, s2 = 0;
        int n = 0;
        if (ok)
            for (int i = -hw; i <= hw; i++)
                for (int j = -hw; j <= hw; j++) {
                    const float v = u[(x + i) + (long long)(y + j) * nx + t * npix];
                    ok = ok && (v == v);
                    mu += v;
                    s2 += v * v;
                    n++;
                }
        n = n ? n : 1;
        mu /= n;
        s2 /= n;
        st[idx + (long long)t * npix] = mu;
        st[idx + (long long)(nch + t) * npix] = s2 - mu * mu;
    }
    st[idx + (long long)(2 * nch) * npix] = ok ? 1.0f : 0.0f;
}

// One wavefront per pixel, lane l takes the labels l, l+64, ...; a workgroup of four waves walks PXB consecutive pixels of
// one image row with the 2*hw+1 rows of both images around it in LDS (conflict-free: consecutive lanes read consecutive
// words; the left window is a broadcast read).
constexpr int kNccPxb = 32;       // pixels of a row per workgroup
constexpr int kNccMaxHw = 3;      // windows up to 7x7 (CENSUS_NCC_WIN <= 7); wider ones take the general kernel
constexpr int kNccMaxL = 1024;    // LDS: (PXB + L + 2*hw) floats per row and channel
template <int HW>
__global__ void __launch_bounds__(256) k_cost_ncc(const CostParams P)
{
    constexpr int WIN = 2 * HW + 1;
    extern __shared__ float ncc_lds[];
    const int nch = P.nch, L = P.L;
    const int ntx = (P.nx + kNccPxb - 1) / kNccPxb;
    const int y = blockIdx.x / ntx, x0 = (blockIdx.x % ntx) * kNccPxb;
    const int uw = kNccPxb + 2 * HW;           // staged columns of the left image: x0-HW ..
    const int vw = kNccPxb + L - 1 + 2 * HW;   // ... of the right image: x0+dmin-HW ..
    float *Lu = ncc_lds;                       // [nch][WIN][uw]
    float *Lv = Lu + nch * WIN * uw;           // [nch][WIN][vw]
    const long long npix = (long long)P.nx * P.ny, vpix = (long long)P.vnx * P.vny;
    for (int k = threadIdx.x; k < nch * WIN * uw; k += blockDim.x) {
        const int c = k % uw, r = (k / uw) % WIN, t = k / (uw * WIN);
        const int xx = x0 - HW + c, yy = y - HW + r;
        Lu[k] = (xx >= 0 && xx < P.nx && yy >= 0 && yy < P.ny) ? P.u[xx + (long long)yy * P.nx + t * npix] : 0.0f;
    }
    for (int k = threadIdx.x; k < nch * WIN * vw; k += blockDim.x) {
        const int c = k % vw, r = (k / vw) % WIN, t = k / (vw * WIN);
        const int xx = x0 + P.dmin - HW + c, yy = y - HW + r;
        Lv[k] = (xx >= 0 && xx < P.vnx && yy >= 0 && yy < P.vny) ? P.v[xx + (long long)yy * P.vnx + t * vpix] : 0.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool yin = y < P.vny;
    for (int xl = wave; xl < kNccPxb; xl += 4) {
        const int x = x0 + xl;
        if (x >= P.nx) break;
        const long long pix = (long long)y * P.nx + x;
        float *Cp = P.C + pix * L;
        const bool ok1 = P.ncc_u[pix + (long long)(2 * nch) * npix] != 0.0f;
        bool anyfinite = false, nanv = false;
        for (int o = lane; o < L; o += 64) {
            const int qx = x + o + P.dmin;
            float e = P.trunc;
            if (yin && qx >= 0 && qx < P.vnx) {
                const long long q = (long long)y * P.vnx + qx;
                if (!ok1 || P.ncc_v[q + (long long)(2 * nch) * vpix] == 0.0f) {
                    e = __builtin_huge_valf();
                } else {
                    float NCC = 0;
                    for (int t = 0; t < nch; t++) {
                        const float *a = Lu + (t * WIN) * uw + xl;            // left window: column xl + (i + HW), row j + HW
                        const float *b = Lv + (t * WIN) * vw + xl + o;        // right window: column xl + o + (i + HW)
                        float prod = 0;
#pragma unroll
                        for (int i = 0; i < WIN; i++)
#pragma unroll
                            for (int j = 0; j < WIN; j++) prod += a[j * uw + i] * b[j * vw + i];
                        prod /= (WIN * WIN);
                        const float mu1 = P.ncc_u[pix + (long long)t * npix], mu2 = P.ncc_v[q + (long long)t * vpix];
                        const float var = P.ncc_u[pix + (long long)(nch + t) * npix] * P.ncc_v[q + (long long)(nch + t) * vpix];
                        const double den = (0.0000001 > var) ? 0.0000001 : (double)var;
                        NCC = (float)(NCC + (prod - mu1 * mu2) / __builtin_sqrt(den));
                    }
                    const float m = (NCC < nch) ? NCC : (float)nch;
                    const float c = (0 > m) ? 0 : m;
                    const float clipped = nch - c;
                    e = clipped * 64;
                }
            }
            e = (e < P.trunc) ? e : P.trunc;
            Cp[o] = e;
            anyfinite |= finite_bits(e);
            nanv |= e != e;
        }
        // no valid hypothesis for this pixel => all labels cost 0 (mgm_costvolume.h:414-421)
        if (__builtin_amdgcn_ballot_w64(anyfinite) == 0ull)
            for (int o = lane; o < L; o += 64) Cp[o] = 0.0f;
        if (P.bad8 && __builtin_amdgcn_ballot_w64(nanv) != 0ull && lane == 0) flag_once(P.bad8, 2u);
    }
}

template <int FN>
static void launch_btx(const CostParams &p, long long nw, hipStream_t s)
{
    if (p.nx % 4) hipLaunchKernelGGL((k_cost_btx<FN, false>), dim3((unsigned)nw), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_cost_btx<FN, true>), dim3((unsigned)nw), dim3(256), 0, s, p);
}
template <int L>
static void launch_census8x(const CostParams &p, dim3 grid, unsigned tb, hipStream_t s)
{
    if (p.nx % 4) hipLaunchKernelGGL((k_cost_census8x<L, false>), grid, dim3(256), 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal);
    else hipLaunchKernelGGL((k_cost_census8x<L, true>), grid, dim3(256), 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal);
}
// Launches the restructured kernel that serves `p`, if there is one (*taken), else leaves the volume to the general kernel.
hipError_t launch_cost_fast(const CostParams &p, hipStream_t s, bool *taken)
{
    *taken = true;
    const long long npix = (long long)p.nx * p.ny;
    if (p.costfn == 3 && p.ncc_u && p.ncc_v && p.C && !p.C8 && !p.rlo && p.hwin >= 1 && p.hwin <= kNccMaxHw && p.L <= kNccMaxL && p.nch <= 4) {
        const long long vpix = (long long)p.vnx * p.vny;
        hipLaunchKernelGGL(k_ncc_stats, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, p.u, p.nx, p.ny, p.nch, p.hwin, p.ncc_u);
        hipLaunchKernelGGL(k_ncc_stats, dim3((unsigned)((vpix + 255) / 256)), dim3(256), 0, s, p.v, p.vnx, p.vny, p.nch, p.hwin, p.ncc_v);
        const int win = 2 * p.hwin + 1;
        const size_t lds = sizeof(float) * (size_t)p.nch * win * ((kNccPxb + 2 * p.hwin) + (kNccPxb + p.L - 1 + 2 * p.hwin));
        const dim3 grid((unsigned)(((p.nx + kNccPxb - 1) / kNccPxb) * (long long)p.ny));
        hipError_t e = hipSuccess;
        switch (p.hwin) {
            case 1:
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cost_ncc<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e == hipSuccess) hipLaunchKernelGGL(k_cost_ncc<1>, grid, dim3(256), lds, s, p);
                break;
            case 2:
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cost_ncc<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e == hipSuccess) hipLaunchKernelGGL(k_cost_ncc<2>, grid, dim3(256), lds, s, p);
                break;
            default:
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cost_ncc<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e == hipSuccess) hipLaunchKernelGGL(k_cost_ncc<3>, grid, dim3(256), lds, s, p);
                break;
        }
        return e != hipSuccess ? e : hipGetLastError();
    }
    if (p.costfn >= 4 && p.ncc_u && p.ncc_v && p.C && !p.C8 && !p.rlo && p.L % 4 == 0) {
        const long long vpix = (long long)p.vnx * p.vny;
        hipLaunchKernelGGL(k_bt_spans, dim3((unsigned)((npix * p.nch + 255) / 256)), dim3(256), 0, s, p.u, p.nx, p.ny, p.nch, p.ncc_u);
        hipLaunchKernelGGL(k_bt_spans, dim3((unsigned)((vpix * p.nch + 255) / 256)), dim3(256), 0, s, p.v, p.vnx, p.vny, p.nch, p.ncc_v);
        long long nw = ((long long)((p.nx + 3) / 4) * p.ny + 3) / 4;
        if (nw > 256 * 64) nw = 256 * 64;
        if (nw < 1) nw = 1;
        if (p.costfn == 5) launch_btx<5>(p, nw, s);
        else launch_btx<4>(p, nw, s);
        return hipGetLastError();
    }
    // differences / multi-word census without a compact form: fp32 volume only (see mgm_costvolume_build_dev)
    if (p.costfn <= 2 && p.C && !p.C8 && !p.rlo && p.L % 4 == 0) {
        long long nw = ((long long)((p.nx + 3) / 4) * p.ny + 3) / 4;
        if (nw > 256 * 64) nw = 256 * 64;
        if (nw < 1) nw = 1;
        if (p.costfn == 0) launch_btx<0>(p, nw, s);
        else if (p.costfn == 1) launch_btx<1>(p, nw, s);
        else launch_btx<2>(p, nw, s);
        return hipGetLastError();
    }
    // (k_cost_diffx takes truncDist = +INF or a non-negative number, sign bit clear; anything else goes to k_cost below)
    if (!p.C && p.C8 && (p.costfn == 0 || p.costfn == 1) && !p.rlo && npix < 0x7fffffffll && c8_supported(p.L) &&
        (p.cbytes == 1 || p.cbytes == 2) && p.L * p.cbytes <= 1024 && p.trunc >= 0.0f && !__builtin_signbit(p.trunc)) {
        if (p.cbytes == 2) p.costfn == 1 ? launch_diffx<2, true>(p, s) : launch_diffx<2, false>(p, s);
        else p.costfn == 1 ? launch_diffx<1, true>(p, s) : launch_diffx<1, false>(p, s);
        return hipGetLastError();
    }
    if (!p.C && p.C8 && p.costfn == 2 && p.nch == 1 && c8_supported(p.L)) {
        const unsigned tb = p.trunc == __builtin_huge_valf() ? 255u : (unsigned)p.trunc;
        long long nb = (npix + 3) / 4;
        if (nb > 256 * 32) nb = 256 * 32;
        const dim3 block(256);
        if (npix < 0x7fffffffll) {  // (every compact label count is a multiple of 16)
            long long nw = ((long long)((p.nx + 3) / 4) * p.ny * 4 * p.L / 4096 + 3) / 4 + 1;
            if (nw > 256 * 32) nw = 256 * 32;
            const dim3 gridw((unsigned)nw);
            switch (p.L) {
                case 64: launch_census8x<64>(p, gridw, tb, s); break;
                case 128: launch_census8x<128>(p, gridw, tb, s); break;
                case 192: launch_census8x<192>(p, gridw, tb, s); break;
                case 256: launch_census8x<256>(p, gridw, tb, s); break;
                case 384: launch_census8x<384>(p, gridw, tb, s); break;
                case 512: launch_census8x<512>(p, gridw, tb, s); break;
                case 768: launch_census8x<768>(p, gridw, tb, s); break;
                default: launch_census8x<1024>(p, gridw, tb, s); break;
            }
            return hipGetLastError();
        }
        if ((p.L == 64 || p.L == 128 || p.L == 256 || p.L == 512) && npix < 0x7fffffffll && p.Lreal == p.L) {
            long long nw = (npix * p.L / 1024 + 3) / 4 + 1;
            if (nw > 256 * 32) nw = 256 * 32;
            const dim3 gridw((unsigned)nw);
            switch (p.L) {
                case 64: hipLaunchKernelGGL(k_cost_census8w<64>, gridw, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8); break;
                case 128: hipLaunchKernelGGL(k_cost_census8w<128>, gridw, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8); break;
                case 256: hipLaunchKernelGGL(k_cost_census8w<256>, gridw, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8); break;
                default: hipLaunchKernelGGL(k_cost_census8w<512>, gridw, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8); break;
            }
            return hipGetLastError();
        }
        const dim3 grid((unsigned)nb);
        switch (p.L / 64) {
            case 1: hipLaunchKernelGGL(k_cost_census8<1>, grid, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
            case 2: hipLaunchKernelGGL(k_cost_census8<2>, grid, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
            case 3: hipLaunchKernelGGL(k_cost_census8<3>, grid, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
            case 4: hipLaunchKernelGGL(k_cost_census8<4>, grid, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
            case 6: hipLaunchKernelGGL(k_cost_census8<6>, grid, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
            case 8: hipLaunchKernelGGL(k_cost_census8<8>, grid, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
            case 12: hipLaunchKernelGGL(k_cost_census8<12>, grid, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
            default: hipLaunchKernelGGL(k_cost_census8<16>, grid, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
        }
        return hipGetLastError();
    }
    *taken = false;
    return hipSuccess;
}

}  // namespace mgm
