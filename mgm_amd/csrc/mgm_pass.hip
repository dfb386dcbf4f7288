// mgm_pass.hip -- K3: the MGM scan-line recursion, all directions in ONE launch.
//
// Replaces the per-pass diagonal loop of the reference, mgm_core.cc:489-579,
// and its update functions update_cost2 (66-90), update_costW (95-144),
// update_cost2_trunclinear (197-219), update_costW_trunclinear (229-281) with
// minConvTruncatedLinear (152-163).
//
// Structure (see DESIGN.md "K3"):
//   * one wavefront owns one scan line and walks along it; lane l holds the
//     LPL contiguous disparities o = l*LPL .. l*LPL+LPL-1 of the current pixel;
//   * a workgroup = R waves = a band of R consecutive lines, run in lock-step
//     on a slope-2 diagonal (the reference's own schedule, mgm_core.cc:505-511):
//     at step s wave r is at pixel i = s-1-2r.  The slab a pixel publishes to
//     its successors goes through a 2-deep LDS ring to the next line's wave;
//   * the last line of a band hands its slabs to the first line of the next
//     band (another workgroup, any CU/XCD) through a small global buffer
//     written with agent-scope (sc1, write-through) stores and a progress
//     word; the consumer polls the word relaxed and reads with sc1 loads;
//   * work items (pass, band) are handed out by an atomic ticket in an order
//     in which every item depends only on lower tickets, so a spinning
//     workgroup always waits on one that is already running: no residency
//     assumption, no dependence on dispatch order or XCD placement;
//   * every pass writes its own Lr volume; K5 (mgm_wta.hip) sums them in pass
//     order, which keeps the reference's fp32 summation order (582-587).
//
// What a pixel publishes ("W"), chosen so that no consumer repeats work:
//   unit weights, Hirschmueller : T[o] = fmin3(L[o], min(L[o-1],L[o+1])+P1, m+P2)
//   unit weights, FH            : T[o] = minconv(L)[o]            (then min(.,m+P2))
//   weighted,     Hirschmueller : L[o] and N[o] = min(L[o-1],L[o+1])
//   weighted,     FH            : L[o]
// plus m = min_o L[o] (the cached Dvec minimum, dvec.cc:81-88).
//
// This translation unit is compiled with -fno-honor-nans: the Lr recursion is
// NaN-free (costs are finite or +INF and every pixel has a finite cost, see
// mgm_costvolume.h:414-421), and it lets v_min_f32 be used without
// canonicalisation.  Nothing here may rely on NaN semantics.
#include "mgm_pass_common.h"

namespace mgm {

// ---- the kernel ------------------------------------------------------------------
template <int LPL, bool FH, bool WEIGHTED, int R>
__global__ void __launch_bounds__(R * 64) k_pass(const PassParams P)
{
    constexpr int LP = LPL * 64;
    constexpr int NS = (WEIGHTED && !FH) ? 2 : 1;
    using NbT = Nb<LPL, NS>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *ring = smem;                        // [R][2][NS][LP]
    float *ringm = smem + R * 2 * NS * LP;     // [R][2]
    int *s_task = reinterpret_cast<int *>(ringm + R * 2);

    const int tid = threadIdx.x, lane = tid & 63;
    const int r = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) *s_task = (int)atomicAdd(P.ticket, 1u);
    __syncthreads();
    const int2 tk = P.tasks[*s_task];
    const int vp = tk.x, band = tk.y;  // vp = volume*8 + pass
    const int pass = vp & (kMaxDirs - 1);
    const PassVolume &V = P.vol[vp / kMaxDirs];
    const PassGeom &g = P.g[pass];
    const int NL = g.NL, LL = g.LL, L = P.L, MGM = P.MGM, form = g.form;
    const float P1 = P.P1, P2 = P.P2;
    const bool exact = (L == LP);
    const int j = band * R + r;
    const bool line_ok = j < NL;
    const bool has_prev = line_ok && (j >= 1);
    const bool from_global = (r == 0) && (band > 0);
    const bool to_lds = (r < R - 1) && (j + 1 < NL);
    const bool to_global = (r == R - 1) && (band + 1 < g.nbands);

    const float *__restrict__ Cb = V.C;
    float *__restrict__ Lrb = V.Lr + (long long)(pass - P.pass0) * P.nvol;
    const long long pix0 = g.base + (long long)j * g.jstep;
    const long long istep = g.istep;

    float *hand_out = P.hand + ((long long)(vp * 2 + (band & 1)) * P.LLmax) * (NS * LP);
    float *handm_out = P.handm + (long long)(vp * 2 + (band & 1)) * P.LLmax;
    const float *hand_in = P.hand + ((long long)(vp * 2 + ((band + 1) & 1)) * P.LLmax) * (NS * LP);
    const float *handm_in = P.handm + (long long)(vp * 2 + ((band + 1) & 1)) * P.LLmax;
    unsigned *prog_out = P.prog + vp * P.maxbands + band;
    const unsigned *prog_in = prog_out - 1;  // only dereferenced when band > 0

    float Cpf[PF][LPL];
    NbT Hpf[PF] = {};
    NbT nb_b = {}, nb_s = {}, nb_f = {}, nb_i = {};  // back, same, fwd (previous line), inline (this line)
    unsigned known = 0;
    bool dead = false;  // watchdog fired: stop polling, results are garbage, host reports it

#pragma unroll
    for (int u = 0; u < PF; u++) {
#pragma unroll
        for (int k = 0; k < LPL; k++) Cpf[u][k] = f_inf();
    }

    // wait until the producer band has published slabs [0, need)
    auto ensure = [&](unsigned need) {
        if (known >= need || dead) return;
        unsigned spins = 0;
        for (;;) {
            known = __hip_atomic_load(prog_in, RLX_AGENT);
            if (known >= need) break;
            __builtin_amdgcn_s_sleep(4);
            if (((++spins) & 1023u) == 0) {
                if (spins > SPIN_LIMIT || __hip_atomic_load(P.err, RLX_AGENT) != 0) {
                    if (lane == 0) __hip_atomic_store(P.err, 1u, RLX_AGENT);
                    dead = true;
                    break;
                }
            }
        }
    };
    auto issue_prefetch = [&](int s, float(&cdst)[LPL], NbT &hdst) {
        const int i = s - 1 - 2 * r;
        if (line_ok && i >= 0 && i < LL) load_slab<LPL>(Cb + (pix0 + (long long)i * istep) * L, lane, L, exact, cdst);
        if (from_global) {
            const int h = i + 1;  // r == 0 => h = s
            if (h >= 0 && h < LL) {
                ensure((unsigned)h + 1u);
#pragma unroll
                for (int q = 0; q < NS; q++) load_slab_sc1<LPL>(hand_in + ((long long)h * NS + q) * LP, lane, hdst.w[q]);
                hdst.m = __builtin_bit_cast(
                    float, __hip_atomic_load(reinterpret_cast<const unsigned *>(handm_in + h), RLX_AGENT));
            }
        }
    };

    if (from_global) ensure((unsigned)(LL < PF ? LL : PF));
#pragma unroll
    for (int u = 0; u < PF; u++) issue_prefetch(u, Cpf[u], Hpf[u]);

    const int nsteps = LL + 1 + 2 * (R - 1);
    for (int s0 = 0; s0 < nsteps; s0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int s = s0 + u;
            const int i = s - 1 - 2 * r;

            // (a) slide the previous-line window and fetch slab i+1 of line j-1
            if (has_prev && i >= -1 && i + 1 < LL) {
                nb_b = nb_s;
                nb_s = nb_f;
                if (r > 0) {
                    const float *src = ring + ((r - 1) * 2 + ((i + 1) & 1)) * NS * LP + lane * LPL;
#pragma unroll
                    for (int q = 0; q < NS; q++)
#pragma unroll
                        for (int k = 0; k < LPL; k++) nb_f.w[q][k] = src[q * LP + k];
                    nb_f.m = ringm[(r - 1) * 2 + ((i + 1) & 1)];
                } else {
                    nb_f = Hpf[u];
                }
            }

            // (b) the pixel itself
            if (line_ok && i >= 0 && i < LL) {
                const long long pix = pix0 + (long long)i * istep;
                float Lv[LPL];
                const bool interior = has_prev && i >= 1 && i <= LL - 2;  // mgm_core.cc:538-541
                if (interior) {
                    if constexpr (!WEIGHTED) {
                        if (form == 0) combine_unit<LPL>(Cpf[u], nb_i, nb_s, nb_b, nb_f, MGM, FH, Lv);
                        else combine_unit<LPL>(Cpf[u], nb_f, nb_b, nb_s, nb_i, MGM, FH, Lv);
                    } else {
                        float D[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) D[k] = V.w8[(long long)g.wplane[k] * P.npix + pix];
                        int rl = 0, rh = 0x7fffffff;  // the pixel's own label range (ragged volumes, FH only)
                        if (FH && V.rlo) {
                            rl = (int)V.rlo[pix] - P.dmin;
                            rh = (int)V.rhi[pix] - P.dmin;
                        }
                        if constexpr (!FH) {
                            if (form == 0) combine_whirsch<LPL>(Cpf[u], nb_i, nb_s, nb_b, nb_f, D, P1, P2, MGM, Lv);
                            else combine_whirsch<LPL>(Cpf[u], nb_f, nb_b, nb_s, nb_i, D, P1, P2, MGM, Lv);
                        } else {
                            if (form == 0) combine_wfh<LPL>(Cpf[u], nb_i, nb_s, nb_b, nb_f, D, P1, P2, MGM, lane, L, Lv, rl, rh);
                            else combine_wfh<LPL>(Cpf[u], nb_f, nb_b, nb_s, nb_i, D, P1, P2, MGM, lane, L, Lv, rl, rh);
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < LPL; k++) Lv[k] = Cpf[u][k];
                }
                store_slab<LPL>(Lrb + pix * L, lane, L, exact, Lv);

                // what this pixel publishes
                const float m = slab_min<LPL>(Lv);
                nb_i.m = m;
                if constexpr (!WEIGHTED) {
                    if constexpr (!FH) {
                        float N[LPL];
                        neighbour_min<LPL>(Lv, N);
                        const float cap = m + P2;
#pragma unroll
                        for (int k = 0; k < LPL; k++) nb_i.w[0][k] = fminf(fminf(Lv[k], N[k] + P1), cap);
                    } else {
#pragma unroll
                        for (int k = 0; k < LPL; k++) nb_i.w[0][k] = Lv[k];
                        fh_minconv<LPL>(nb_i.w[0], m, P1, P2, lane, L);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < LPL; k++) nb_i.w[0][k] = Lv[k];
                    if constexpr (!FH) neighbour_min<LPL>(Lv, nb_i.w[NS - 1]);
                }
                if (to_lds) {
                    float *dst = ring + (r * 2 + (i & 1)) * NS * LP + lane * LPL;
#pragma unroll
                    for (int q = 0; q < NS; q++)
#pragma unroll
                        for (int k = 0; k < LPL; k++) dst[q * LP + k] = nb_i.w[q][k];
                    if (lane == 0) ringm[r * 2 + (i & 1)] = m;
                }
                if (to_global) {
#pragma unroll
                    for (int q = 0; q < NS; q++)
                        store_slab_sc1<LPL>(hand_out + ((long long)i * NS + q) * LP, lane, nb_i.w[q]);
                    if (lane == 0)
                        __hip_atomic_store(reinterpret_cast<unsigned *>(handm_out + i), __builtin_bit_cast(unsigned, m),
                                           RLX_AGENT);
                    if (((i + 1) % CH) == 0 || i == LL - 1) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (lane == 0) __hip_atomic_store(prog_out, (unsigned)(i + 1), RLX_AGENT);
                    }
                }
            }

            // (c) prefetch for step s + PF
            issue_prefetch(s + PF, Cpf[u], Hpf[u]);

            // (d) everybody's slab for this step is in LDS before anyone reads it
            lds_barrier();
        }
    }
}

// ---- self-test: div3_exact against IEEE division over every fp32 bit pattern ---------
__global__ void __launch_bounds__(256) k_selftest_div3(unsigned long long *nbad)
{
    unsigned long long bad = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride) {
        const float x = __builtin_bit_cast(float, (unsigned)b);
        const unsigned a = __builtin_bit_cast(unsigned, x / 3.0f);
        const unsigned q = __builtin_bit_cast(unsigned, div3_exact(x));
        const bool a_nan = (a & 0x7fffffffu) > 0x7f800000u, q_nan = (q & 0x7fffffffu) > 0x7f800000u;
        if (a != q && !(a_nan && q_nan)) bad++;
    }
    if (bad) atomicAdd(nbad, bad);
}
hipError_t launch_selftest_div3(unsigned long long *nbad, hipStream_t s)
{
    hipLaunchKernelGGL(k_selftest_div3, dim3(4096), dim3(256), 0, s, nbad);
    return hipGetLastError();
}

// ---- launcher ------------------------------------------------------------------
int pass_ns(bool fh, bool weighted) { return (weighted && !fh) ? 2 : 1; }
int pass_lpl(int L)
{
    const int lpl = (L + 63) / 64;
    if (lpl == 5) return 6;
    if (lpl == 7) return 8;
    if (lpl <= 8) return lpl;
    return lpl <= 12 ? 12 : (lpl <= 16 ? 16 : (lpl <= 24 ? 24 : 32));  // 513..2048 labels: bands of four lines
}

template <int LPL, bool FH, bool WEIGHTED, int R>
static hipError_t launch_one(const PassParams &p, int ntasks, hipStream_t s)
{
    constexpr int LP = LPL * 64;
    constexpr int NS = (WEIGHTED && !FH) ? 2 : 1;
    const size_t shmem = sizeof(float) * (R * 2 * NS * LP + R * 2) + 16;
    auto kern = k_pass<LPL, FH, WEIGHTED, R>;
    if (shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(ntasks), dim3(R * 64), shmem, s, p);
    return hipGetLastError();
}

template <int LPL, int R>
static hipError_t launch_lpl(const PassParams &p, int ntasks, bool fh, int wmode, hipStream_t s)
{
    if (!wmode) return fh ? launch_one<LPL, true, false, R>(p, ntasks, s) : launch_one<LPL, false, false, R>(p, ntasks, s);
    return fh ? launch_one<LPL, true, true, R>(p, ntasks, s) : launch_one<LPL, false, true, R>(p, ntasks, s);
}

hipError_t launch_pass(const PassParams &p, int ntasks, int R, bool fh, int wmode, hipStream_t s)
{
    const int lpl = pass_lpl(p.L);
    if (R == 4)  // more than 512 labels (any label count up to 2048): the slabs of a line need most of a SIMD's registers
        switch (lpl) {
            case 12: return launch_lpl<12, 4>(p, ntasks, fh, wmode, s);
            case 16: return launch_lpl<16, 4>(p, ntasks, fh, wmode, s);
            case 24: return launch_lpl<24, 4>(p, ntasks, fh, wmode, s);
            case 32: return launch_lpl<32, 4>(p, ntasks, fh, wmode, s);
            default: return hipErrorInvalidValue;
        }
    if (R != 16) return hipErrorInvalidValue;
    switch (lpl) {
        case 1: return launch_lpl<1, 16>(p, ntasks, fh, wmode, s);
        case 2: return launch_lpl<2, 16>(p, ntasks, fh, wmode, s);
        case 3: return launch_lpl<3, 16>(p, ntasks, fh, wmode, s);
        case 4: return launch_lpl<4, 16>(p, ntasks, fh, wmode, s);
        case 6: return launch_lpl<6, 16>(p, ntasks, fh, wmode, s);
        case 8: return launch_lpl<8, 16>(p, ntasks, fh, wmode, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mgm
