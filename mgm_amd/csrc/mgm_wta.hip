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

template <int LPL>
__global__ void __launch_bounds__(256) k_wta(const WtaParams P)
{
    constexpr int LP = LPL * 64;
    __shared__ float sS[4][LP];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int L = P.L;
    const bool exact = (L == LP);
    const int o0 = lane * LPL;
    // grid-stride over pixels: a bounded number of workgroups, each wave streams many pixels
    for (long long pix = (long long)blockIdx.x * 4 + wv; pix < P.npix; pix += (long long)gridDim.x * 4) {

    float c[LPL], S[LPL];
    if (P.C8 && exact) {  // compact costs: one byte per label (wave-uniform branch)
        const uint8_t *q = P.C8 + pix * L + o0;
#pragma unroll
        for (int k = 0; k < LPL; k++) c[k] = c8_decode(q[k]);
    } else {
        const float *q = P.C + pix * L + o0;
#pragma unroll
        for (int k = 0; k < LPL; k++) c[k] = (exact || o0 + k < L) ? q[k] : f_inf();
    }
    // all NDIR slabs are requested before the first one is consumed (independent loads in flight),
    // then summed in pass order: S = ((0 + L0) + L1) + ...
    float l[kMaxDirs][LPL];
#pragma unroll
    for (int p = 0; p < kMaxDirs; p++) {
        if (p < P.NDIR) {
            const float *q = P.Lr + (long long)p * P.nvol + pix * L + o0;
#pragma unroll
            for (int k = 0; k < LPL; k++) l[p][k] = (exact || o0 + k < L) ? q[k] : f_inf();
        }
    }
#pragma unroll
    for (int k = 0; k < LPL; k++) S[k] = 0.0f;
#pragma unroll
    for (int p = 0; p < kMaxDirs; p++) {
        if (p < P.NDIR) {
#pragma unroll
            for (int k = 0; k < LPL; k++) S[k] = S[k] + l[p][k];
        }
    }
    if (P.FIX == 1) {
        const float f = (float)(P.NDIR - 1);
#pragma unroll
        for (int k = 0; k < LPL; k++) S[k] = S[k] - f * c[k];
    }
    if (P.S) {
        float *q = P.S + pix * L + o0;
#pragma unroll
        for (int k = 0; k < LPL; k++)
            if (exact || o0 + k < L) q[k] = S[k];
    }

    // first strict minimum among finite entries, ascending o
    float best = f_inf();
    int bi = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < LPL; k++) {
        const float v = S[k];
        if ((exact || o0 + k < L) && finite_bits(v) && best > v) {
            best = v;
            bi = o0 + k;
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
    float outv, outc = best;
    if (bi == 0x7fffffff) {
        outv = __builtin_nanf("");  // the reference leaves minP uninitialised here
    } else {
        outv = (float)(bi + P.dmin);
        if (P.refine == 1 && bi - 1 >= 0 && bi + 2 <= L - 1) {  // mgm_refine.h:58
#pragma unroll
            for (int k = 0; k < LPL; k++) sS[wv][o0 + k] = S[k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const float v0 = sS[wv][bi - 1], v1 = sS[wv][bi], v2 = sS[wv][bi + 1];
            float vmin, dx;
            vfit(v0, v1, v2, vmin, dx);
            outv = (float)(bi + P.dmin) + dx;
            outc = vmin;
        }
    }
    if (lane == 0) {
        P.out[pix] = outv;
        P.outcost[pix] = outc;
    }
    }  // pixel loop
}

hipError_t launch_wta(const WtaParams &p, hipStream_t s)
{
    long long nb = (p.npix + 3) / 4;
    if (nb > 256 * 16) nb = 256 * 16;  // 16 workgroups of 4 waves per CU, grid-stride beyond
    const dim3 grid((unsigned)nb), block(256);
    switch (pass_lpl(p.L)) {
        case 1: hipLaunchKernelGGL(k_wta<1>, grid, block, 0, s, p); break;
        case 2: hipLaunchKernelGGL(k_wta<2>, grid, block, 0, s, p); break;
        case 3: hipLaunchKernelGGL(k_wta<3>, grid, block, 0, s, p); break;
        case 4: hipLaunchKernelGGL(k_wta<4>, grid, block, 0, s, p); break;
        case 6: hipLaunchKernelGGL(k_wta<6>, grid, block, 0, s, p); break;
        case 8: hipLaunchKernelGGL(k_wta<8>, grid, block, 0, s, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Stand-alone refinement on a materialised (corrected) S: one thread per pixel.
__global__ void __launch_bounds__(256) k_refine(const float *__restrict__ S, long long npix, int L, int dmin,
                                                float *__restrict__ out, float *__restrict__ outcost)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float minP = out[i];
    if (!(minP == minP)) return;  // NaN label (no finite S): undefined in the reference
    const int o = (int)minP;
    if (o - 1 >= dmin && o + 2 <= dmin + L - 1) {
        const float *Si = S + i * L + (o - dmin);
        float vmin, dx;
        vfit(Si[-1], Si[0], Si[1], vmin, dx);
        out[i] = (float)o + dx;
        outcost[i] = vmin;
    }
}

hipError_t launch_refine(const float *S, long long npix, int L, int dmin, int method, float *out, float *outcost,
                         hipStream_t s)
{
    if (method != 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_refine, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, S, npix, L, dmin, out, outcost);
    return hipGetLastError();
}

}  // namespace mgm
