// mgm_pass_exact.hip -- K3, the SLOW, operand-order-faithful build: the reference's four update functions
// (mgm_core.cc:66-90, 95-144, 197-219, 229-281) with every minimum written as the reference writes it --
// `a < b ? a : b` (mgm_core.cc:48), fmin3 by two strict `>` tests (54-60), Dvec::get_minvalue by a strict `<` scan in
// label order (dvec.cc:81-88) -- and compiled WITHOUT -fno-honor-nans, so that a NaN operand gives what it gives in the
// reference: which operand holds the NaN decides the result, something v_min_f32 and the fused DPP scans of the fast
// builds do not reproduce.  Taken only for the volumes that can produce NaNs (mgm_api.hip, run_passes):
//   * an uploaded volume that holds NaN costs;
//   * a volume built by `-p census` with a non-census distance from descriptors of more than 24 bits: the costs are
//     differences of descriptor WORDS read as floats (mgm_costvolume.h:355-362), NaN patterns included;
//   * a ragged volume with P2 = +INF: a pixel whose neighbours' ranges miss its own gets an all-INF slab, the next one
//     INF - INF = NaN (mgm_core.cc:242-271 on Dvec::operator[]'s +INF, dvec.cc:129).
//
// Shape: the reference's own schedule (mgm_core.cc:505-579) -- one launch per diagonal ii of the slope-2 sweep, one
// wavefront per pixel of it, the labels strided over the lanes; Lr starts as a copy of C (495-498: here every pixel writes its
// WHOLE slab when its diagonal comes -- each pixel lies on exactly one -- so no copy is made first), every visited pixel
// leaves its slab minimum (577) in `mins`.  Round 6: the passes of a volume are independent of each other (their sum is taken
// later, in order), so ONE launch per diagonal serves all of them (blockIdx.y = pass): 8 x fewer launches of 8 x the waves --
// the kernel is bound by its launches (one per diagonal: ~4900 for a 1920x1080 image), not by its arithmetic.  Ragged volumes keep the dense hull layout of the fast path (foreign labels
// hold +INF = what Dvec::operator[] returns for them), a pixel only writes the labels of its own range (Dvec::set_nolock,
// dvec.cc:111-118), and the FH functions convolve over the RECEIVING pixel's range with
// FixBounrady_for_minConvTruncatedLinear (166-186) where the reference applies it (update_cost2_trunclinear only).
// Nothing here is tuned: a 1920x1080x256 volume takes on the order of a second.
#include <algorithm>

#include "mgm_device.h"

namespace mgm {


__device__ __forceinline__ float ref_min(float a, float b) { return (a < b) ? a : b; }  // __min, mgm_core.cc:48
__device__ __forceinline__ float ref_fmin3(float a, float b, float c)                   // mgm_core.cc:54-60
{
    float m = a;
    if (m > b) m = b;
    if (m > c) m = c;
    return m;
}
// e / howmany as the reference's fp32 division (the proven sequences of mgm_pass_common.h; NaN and INF pass through)
__device__ __forceinline__ float ref_div(float e, int n)
{
    if (n == 1) return e;
    if (n == 2) return e * 0.5f;
    if (n == 4) return e * 0.25f;
    const float c = 0x1.555556p-2f;
    const float q0 = e * c;
    const float r = __builtin_fmaf(-3.0f, q0, e);
    const float q = __builtin_fmaf(r, c, q0);
    return __builtin_amdgcn_div_fixupf(q, 3.0f, e);
}

// LDS: up to four convolution arrays of the receiving pixel's range
extern __shared__ float exact_smem[];

__global__ void __launch_bounds__(64) k_pass_exact(const ExactParams P)
{
    const int lane = threadIdx.x;
    const ExactPass &G = P.pass[blockIdx.y];
    if ((int)blockIdx.x >= G.njj) return;
    const int jj = G.jj0 + blockIdx.x;
    const int maxii = G.row_major ? P.nx : P.ny, maxjj = G.row_major ? P.ny : P.nx;
    int x = P.ii - 2 * jj, y = jj;
    if (x < 0 || x >= maxii || jj >= maxjj) return;
    if (!G.row_major) {
        const int t = x;
        x = y;
        y = t;
    }
    if (G.inc_x == 0) x = (P.nx - 1) - x;
    if (G.inc_y == 0) y = (P.ny - 1) - y;
    const long long npix = (long long)P.nx * P.ny, pidx = (long long)x + (long long)y * P.nx;
    const int L = P.L;
    float *const Lr = G.Lr;
    float *const mins = P.mins + (long long)blockIdx.y * npix;
    float *Lp = Lr + pidx * L;
    const float *Cp = P.C + pidx * L;
    int rl = 0, rh = L - 1;  // the pixel's own labels (dense indices)
    if (P.rlo) {
        rl = (int)P.rlo[pidx] - P.dmin;
        rh = (int)P.rhi[pidx] - P.dmin;
    }
    long long nidx[4];
    bool inside = true;
    for (int k = 0; k < 4; k++) {
        const int qx = x + G.d[k][0], qy = y + G.d[k][1];
        if (!(qx >= 0 && qy >= 0 && qx < P.nx && qy < P.ny)) inside = false;
        nidx[k] = (long long)qx + (long long)qy * P.nx;
    }
    const int hm = P.MGM;
    if (inside) {  // mgm_core.cc:538-541: all four neighbours inside, whatever MGM is
        const float *Ln[4];
        float mn[4], D[4];
        int nl[4], nh[4];  // the neighbours' own ranges (FixBoundary)
        for (int k = 0; k < 4; k++) {
            Ln[k] = Lr + nidx[k] * L;
            mn[k] = (k < hm) ? mins[nidx[k]] : __builtin_huge_valf();
            D[k] = P.w8 ? P.w8[pidx + (long long)G.wplane[k] * npix] : 1.0f;
            nl[k] = 0;
            nh[k] = L - 1;
            if (P.rlo) {
                nl[k] = (int)P.rlo[nidx[k]] - P.dmin;
                nh[k] = (int)P.rhi[nidx[k]] - P.dmin;
            }
        }
        auto at = [&](const float *a, int o) { return (o >= 0 && o < L) ? a[o] : __builtin_huge_valf(); };  // Dvec::operator[]
        // the labels of the hull this pixel does not own keep what C holds there (+INF): Lr = CC (495-498), set_nolock drops them
        for (int o = lane; o < L; o += 64)
            if (o < rl || o > rh) Lp[o] = Cp[o];
        if (P.mode == 0) {
            for (int o = rl + lane; o <= rh; o += 64) {
                float e = 0;
                for (int k = 0; k < 2; k++) {
                    const float vL0 = Ln[k][o];
                    const float vLP1 = ref_min(at(Ln[k], o - 1), at(Ln[k], o + 1)) + P.P1;
                    const float vLP2 = mn[k] + P.P2;
                    e += (ref_fmin3(vL0, vLP1, vLP2) - mn[k]) * 0.5f;  // (x / 2 exactly)
                }
                Lp[o] = Cp[o] + e;
            }
        } else if (P.mode == 1) {
            for (int o = rl + lane; o <= rh; o += 64) {
                float e = 0;
                for (int k = 0; k < hm; k++) {
                    const float vL0 = Ln[k][o];
                    const float vLP1 = ref_min(at(Ln[k], o - 1), at(Ln[k], o + 1)) + P.P1 * D[k];
                    const float vLP2 = mn[k] + P.P2 * D[k];
                    e += ref_fmin3(vL0, vLP1, vLP2) - mn[k];
                }
                Lp[o] = Cp[o] + ref_div(e, hm);
            }
        } else {
            // FH: the neighbours' values over THIS pixel's range, convolved there (197-219, 229-281)
            const int NN = rh - rl + 1;
            const int nk = P.mode == 2 ? 2 : hm;
            // the convolution arrays: in LDS up to 8192 labels, beyond that in a slice of global scratch of this workgroup's
            // own (one wave per workgroup; the block barriers below order its stores and loads)
            float *const conv = P.fhscratch ? P.fhscratch + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 * (size_t)L : exact_smem;
            if (NN > 0) {
                for (int k = 0; k < nk; k++)
                    for (int o = lane; o < NN; o += 64) conv[(size_t)k * NN + o] = Ln[k][rl + o];
                __threadfence_block();
                __syncthreads();
                if (lane < nk) {  // one lane per neighbour: the recurrences are sequential in fp32 (152-163)
                    const int k = lane;
                    float *M = conv + (size_t)k * NN;
                    const float p1 = P.mode == 2 ? P.P1 : P.P1 * D[k], p2 = P.mode == 2 ? P.P2 : P.P2 * D[k];
                    if (P.mode == 2) {  // FixBounrady_for_minConvTruncatedLinear (166-186): I = the neighbour's own labels
                        const int imin = nl[k], imax = nh[k], mmin = rl, mmax = rh;
                        const float *I = Ln[k] + imin;
                        if (imin < mmin) {
                            float T = I[0];
                            for (int o = imin + 1; o <= mmin; o++) {
                                const float Inext = o <= imax ? I[o - imin] : __builtin_huge_valf();
                                T = ref_min(T + p1, Inext);
                            }
                            M[0] = ref_min(M[0], T);
                        }
                        if (imax > mmax) {
                            float T = I[imax - imin];
                            for (int o = imax - 1; o >= mmax; o--) {
                                const float Inext = o >= imin ? I[o - imin] : __builtin_huge_valf();
                                T = ref_min(T + p1, Inext);
                            }
                            M[mmax - mmin] = ref_min(M[mmax - mmin], T);
                        }
                    }
                    for (int o = 1; o < NN; o++) M[o] = ref_min(M[o - 1] + p1, M[o]);
                    for (int o = NN - 2; o >= 0; o--) M[o] = ref_min(M[o + 1] + p1, M[o]);
                    if (p2 < __builtin_huge_valf())
                        for (int o = 0; o < NN; o++) M[o] = ref_min(M[o], mn[k] + p2);
                }
                __threadfence_block();
                __syncthreads();
                for (int o = lane; o < NN; o += 64) {
                    float v;
                    if (P.mode == 2) {
                        v = Cp[rl + o] + (conv[o] - mn[0] + conv[(size_t)NN + o] - mn[1]) * 0.5f;
                    } else {
                        float e = conv[o] - mn[0];
                        for (int k = 1; k < hm; k++) e += conv[(size_t)k * NN + o] - mn[k];
                        v = Cp[rl + o] + ref_div(e, hm);
                    }
                    Lp[rl + o] = v;
                }
            }
        }
        __syncthreads();  // (the slab is complete before its minimum is taken; one wave per block)
    } else {  // a pixel of the frame: never updated, its Lr is its cost slab (495-498)
        for (int o = lane; o < L; o += 64) Lp[o] = Cp[o];
        __syncthreads();
    }
    // Dvec::get_minvalue (dvec.cc:81-88): strict `<` scan in label order from +INF -- NaNs never enter, and of several
    // equal minima (+0 and -0) the FIRST is kept.  Lane l scans the labels l*chunk .. in order, then the lanes combine in
    // lane order with the same strict test.
    const int NNr = rh - rl + 1;
    const int chunk = NNr > 0 ? (NNr + 63) / 64 : 0;
    float m = __builtin_huge_valf();
    for (int q = 0; q < chunk; q++) {
        const int o = rl + lane * chunk + q;
        if (o <= rh) {
            const float v = Lp[o];
            if (v < m) m = v;
        }
    }
    for (int off = 1; off < 64; off <<= 1) {  // after the step with `off`, lane l holds the scan result of lanes l .. l+2*off-1
        const float hi = __shfl_down(m, off);
        if (lane + off < 64 && hi < m) m = hi;
    }
    if (lane == 0) mins[pidx] = m;
}

hipError_t launch_pass_exact(const ExactParams &base, hipStream_t s)
{
    ExactParams p = base;
    size_t shmem = (p.mode >= 2) ? sizeof(float) * 4 * (size_t)p.L : 0;
    if (shmem > 128 * 1024) {  // (more than 8192 labels: the caller provides global scratch, mgm_plan.hip)
        if (!p.fhscratch) return hipErrorInvalidValue;
        shmem = 0;
    } else
        p.fhscratch = nullptr;
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_pass_exact), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    int last = 0;
    for (int q = 0; q < p.npass; q++) last = std::max(last, p.pass[q].row_major ? p.nx + 2 * p.ny : p.ny + 2 * p.nx);
    for (int ii = 0; ii < last; ii++) {  // mgm_core.cc:505: for (ii = 0; ii < maxii + 2*maxjj; ii++), every pass of the launch at once
        int most = 0;
        for (int q = 0; q < p.npass; q++) {
            const int maxii = p.pass[q].row_major ? p.nx : p.ny, maxjj = p.pass[q].row_major ? p.ny : p.nx;
            // jj with 0 <= ii - 2*jj < maxii
            int jlo = ii - (maxii - 1);
            jlo = jlo <= 0 ? 0 : (jlo + 1) / 2;
            int jhi = ii / 2;
            if (jhi > maxjj - 1) jhi = maxjj - 1;
            p.pass[q].jj0 = jlo;
            p.pass[q].njj = jhi < jlo ? 0 : jhi - jlo + 1;
            most = std::max(most, p.pass[q].njj);
        }
        if (most == 0) continue;
        p.ii = ii;
        hipLaunchKernelGGL(k_pass_exact, dim3((unsigned)most, (unsigned)p.npass), dim3(64), shmem, s, p);
    }
    return hipGetLastError();
}

}  // namespace mgm
