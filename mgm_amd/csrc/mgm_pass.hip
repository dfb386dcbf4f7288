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
#include "mgm_device.h"

namespace mgm {

constexpr int PF = 4;                    // prefetch depth (steps)
constexpr int CH = 8;                    // pixels per inter-band progress publication
constexpr unsigned SPIN_LIMIT = 1u << 22;  // watchdog for the inter-band poll

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- slab I/O ---------------------------------------------------------------
template <int LPL>
__device__ __forceinline__ void load_slab(const float *__restrict__ p, int lane, int L, bool exact, float (&v)[LPL])
{
    const float *q = p + lane * LPL;
    if (exact) {
#pragma unroll
        for (int k = 0; k < LPL; k++) v[k] = q[k];
    } else {
#pragma unroll
        for (int k = 0; k < LPL; k++) v[k] = (lane * LPL + k < L) ? q[k] : f_inf();
    }
}
template <int LPL>
__device__ __forceinline__ void store_slab(float *__restrict__ p, int lane, int L, bool exact, const float (&v)[LPL])
{
    float *q = p + lane * LPL;
    if (exact) {
#pragma unroll
        for (int k = 0; k < LPL; k++) q[k] = v[k];
    } else {
#pragma unroll
        for (int k = 0; k < LPL; k++)
            if (lane * LPL + k < L) q[k] = v[k];
    }
}
// inter-workgroup hand-off: write-through stores / L1-bypassing loads (sc1)
template <int LPL>
__device__ __forceinline__ void store_slab_sc1(float *p, int lane, const float (&v)[LPL])
{
    unsigned *q = reinterpret_cast<unsigned *>(p) + lane * LPL;
#pragma unroll
    for (int k = 0; k < LPL; k++) __hip_atomic_store(q + k, __builtin_bit_cast(unsigned, v[k]), RLX_AGENT);
}
template <int LPL>
__device__ __forceinline__ void load_slab_sc1(const float *p, int lane, float (&v)[LPL])
{
    const unsigned *q = reinterpret_cast<const unsigned *>(p) + lane * LPL;
#pragma unroll
    for (int k = 0; k < LPL; k++) v[k] = __builtin_bit_cast(float, __hip_atomic_load(q + k, RLX_AGENT));
}

// ---- per-slab transforms ------------------------------------------------------
template <int LPL>
__device__ __forceinline__ float slab_min(const float (&v)[LPL])
{
    float m = v[0];
#pragma unroll
    for (int k = 1; k < LPL; k++) m = fminf(m, v[k]);
    return wave_min(m);
}

// N[o] = min(L[o-1], L[o+1]) with +INF outside the label range (dvec.cc:129)
template <int LPL>
__device__ __forceinline__ void neighbour_min(const float (&Lv)[LPL], float (&N)[LPL])
{
    const float left = dpp_shr1(Lv[LPL - 1], f_inf());
    const float right = dpp_shl1(Lv[0], f_inf());
#pragma unroll
    for (int k = 0; k < LPL; k++) {
        const float lo = k ? Lv[k - 1] : left;
        const float hi = (k < LPL - 1) ? Lv[k + 1] : right;
        N[k] = fminf(lo, hi);
    }
}

// Exact minConvTruncatedLinear (mgm_core.cc:152-163) on a slab spread over the
// wave.  The reference runs two SEQUENTIAL fp32 recurrences over o,
//   fwd: M[o] = min(M[o-1] + P1, M[o])      bwd: M[o] = min(M[o+1] + P1, M[o]),
// each add rounded, so x + n*P1 in one rounding is not equivalent.  Here every
// lane runs the recurrence exactly over its own LPL elements given a carry
// from its neighbour lane; the 64 carries are first GUESSED with a log-step
// scan (single-rounded ramps) and then iterated to the fixed point
//   c_l = carry_out(lane l | carry_in = c_{l-1}),
// which is unique and equals the sequential result (lane 0 has no carry-in, so
// after n sweeps lanes 0..n-1 are exact; the loop ends when a sweep changes
// nothing, normally the first).  `valid` masks label slots >= L.
template <int LPL>
__device__ __forceinline__ void fh_minconv(float (&M)[LPL], float m, float P1, float P2, int lane, int L)
{
    const float rampP = (float)LPL * P1;
    // ---------------- forward ----------------
    {
        float a = M[0];
#pragma unroll
        for (int k = 1; k < LPL; k++) a = fminf(M[k], a + P1);
        float c = a;  // carry-out ignoring carry-in: exact for lane 0, a guess elsewhere
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float t = __shfl_up(c, d) + (float)d * rampP;
            if (lane >= d) c = fminf(c, t);
        }
        float f[LPL];
        for (int it = 0; it < 66; it++) {
            const float cin = dpp_shr1(c, f_inf());
            f[0] = fminf(M[0], cin + P1);
#pragma unroll
            for (int k = 1; k < LPL; k++) f[k] = fminf(M[k], f[k - 1] + P1);
            const bool same = (f[LPL - 1] == c);
            c = f[LPL - 1];
            if (__builtin_amdgcn_ballot_w64(!same) == 0ull) break;
        }
#pragma unroll
        for (int k = 0; k < LPL; k++) M[k] = (lane * LPL + k < L) ? f[k] : f_inf();
    }
    // ---------------- backward ----------------
    {
        float a = M[LPL - 1];
#pragma unroll
        for (int k = LPL - 2; k >= 0; k--) a = fminf(M[k], a + P1);
        float c = a;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float t = __shfl_down(c, d) + (float)d * rampP;
            if (lane + d < 64) c = fminf(c, t);
        }
        float f[LPL];
        for (int it = 0; it < 66; it++) {
            const float cin = dpp_shl1(c, f_inf());
            f[LPL - 1] = fminf(M[LPL - 1], cin + P1);
#pragma unroll
            for (int k = LPL - 2; k >= 0; k--) f[k] = fminf(M[k], f[k + 1] + P1);
            const bool same = (f[0] == c);
            c = f[0];
            if (__builtin_amdgcn_ballot_w64(!same) == 0ull) break;
        }
#pragma unroll
        for (int k = 0; k < LPL; k++) M[k] = f[k];
    }
    if (P2 < f_inf()) {
        const float cap = m + P2;
#pragma unroll
        for (int k = 0; k < LPL; k++) M[k] = fminf(M[k], cap);
    }
}

// ---- the four reference update functions ---------------------------------------
// Neighbour k's published slab(s) and minimum.
template <int LPL, int NS>
struct Nb {
    float w[NS][LPL];
    float m;
};

// unit weights: w[0] = T.  MGM, FH are wave-uniform run-time values.
template <int LPL>
__device__ __forceinline__ void combine_unit(const float (&C)[LPL], const Nb<LPL, 1> &n1, const Nb<LPL, 1> &n2,
                                             const Nb<LPL, 1> &n3, const Nb<LPL, 1> &n4, int MGM, bool FH,
                                             float (&out)[LPL])
{
    if (MGM == 2) {
        if (!FH) {  // update_cost2: e=0; e+=(t1-m1)/2; e+=(t2-m2)/2
#pragma unroll
            for (int k = 0; k < LPL; k++) {
                float e = 0.0f;
                e += (n1.w[0][k] - n1.m) * 0.5f;
                e += (n2.w[0][k] - n2.m) * 0.5f;
                out[k] = C[k] + e;
            }
        } else {  // update_cost2_trunclinear: (M1 - m1 + M2 - m2)/2
#pragma unroll
            for (int k = 0; k < LPL; k++) out[k] = C[k] + (((n1.w[0][k] - n1.m) + n2.w[0][k]) - n2.m) * 0.5f;
        }
        return;
    }
    // update_costW / update_costW_trunclinear with DeltaI = 1
#pragma unroll
    for (int k = 0; k < LPL; k++) {
        float e;
        if (!FH) {
            e = 0.0f;
            e += n1.w[0][k] - n1.m;
        } else {
            e = n1.w[0][k] - n1.m;
        }
        if (MGM >= 2) e += n2.w[0][k] - n2.m;  // MGM == 2 never reaches here
        if (MGM >= 3) e += n3.w[0][k] - n3.m;
        if (MGM >= 4) e += n4.w[0][k] - n4.m;
        float q;
        if (MGM == 1) q = e;
        else if (MGM == 3) q = e / 3.0f;
        else q = e * 0.25f;
        out[k] = C[k] + q;
    }
}

// weighted Hirschmueller (update_costW): w[0] = L, w[1] = N
template <int LPL>
__device__ __forceinline__ float hirsch_w_term(const Nb<LPL, 2> &n, int k, float p1, float p2)
{
    const float t = fminf(fminf(n.w[0][k], n.w[1][k] + p1), n.m + p2);
    return t - n.m;
}
template <int LPL>
__device__ __forceinline__ void combine_whirsch(const float (&C)[LPL], const Nb<LPL, 2> &n1, const Nb<LPL, 2> &n2,
                                                const Nb<LPL, 2> &n3, const Nb<LPL, 2> &n4, const float (&D)[4],
                                                float P1, float P2, int MGM, float (&out)[LPL])
{
    const float a1 = P1 * D[0], b1 = P2 * D[0], a2 = P1 * D[1], b2 = P2 * D[1];
    const float a3 = P1 * D[2], b3 = P2 * D[2], a4 = P1 * D[3], b4 = P2 * D[3];
#pragma unroll
    for (int k = 0; k < LPL; k++) {
        float e = 0.0f;
        e += hirsch_w_term<LPL>(n1, k, a1, b1);
        if (MGM >= 2) e += hirsch_w_term<LPL>(n2, k, a2, b2);
        if (MGM >= 3) e += hirsch_w_term<LPL>(n3, k, a3, b3);
        if (MGM >= 4) e += hirsch_w_term<LPL>(n4, k, a4, b4);
        out[k] = C[k] + e / (float)MGM;
    }
}
// weighted FH (update_costW_trunclinear): w[0] = L; the min-convolution depends
// on the consumer's weights, so it runs here, once per neighbour.
template <int LPL>
__device__ __forceinline__ void combine_wfh(const float (&C)[LPL], const Nb<LPL, 1> &n1, const Nb<LPL, 1> &n2,
                                            const Nb<LPL, 1> &n3, const Nb<LPL, 1> &n4, const float (&D)[4], float P1,
                                            float P2, int MGM, int lane, int L, float (&out)[LPL])
{
    float e[LPL], M[LPL];
#pragma unroll
    for (int k = 0; k < LPL; k++) M[k] = n1.w[0][k];
    fh_minconv<LPL>(M, n1.m, P1 * D[0], P2 * D[0], lane, L);
#pragma unroll
    for (int k = 0; k < LPL; k++) e[k] = M[k] - n1.m;
    if (MGM >= 2) {
#pragma unroll
        for (int k = 0; k < LPL; k++) M[k] = n2.w[0][k];
        fh_minconv<LPL>(M, n2.m, P1 * D[1], P2 * D[1], lane, L);
#pragma unroll
        for (int k = 0; k < LPL; k++) e[k] += M[k] - n2.m;
    }
    if (MGM >= 3) {
#pragma unroll
        for (int k = 0; k < LPL; k++) M[k] = n3.w[0][k];
        fh_minconv<LPL>(M, n3.m, P1 * D[2], P2 * D[2], lane, L);
#pragma unroll
        for (int k = 0; k < LPL; k++) e[k] += M[k] - n3.m;
    }
    if (MGM >= 4) {
#pragma unroll
        for (int k = 0; k < LPL; k++) M[k] = n4.w[0][k];
        fh_minconv<LPL>(M, n4.m, P1 * D[3], P2 * D[3], lane, L);
#pragma unroll
        for (int k = 0; k < LPL; k++) e[k] += M[k] - n4.m;
    }
#pragma unroll
    for (int k = 0; k < LPL; k++) out[k] = C[k] + e[k] / (float)MGM;
}

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
    const int pass = tk.x, band = tk.y;
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

    const float *__restrict__ Cb = P.C;
    float *__restrict__ Lrb = P.Lr + (long long)pass * P.nvol;
    const long long pix0 = g.base + (long long)j * g.jstep;
    const long long istep = g.istep;

    float *hand_out = P.hand + ((long long)(pass * 2 + (band & 1)) * P.LLmax) * (NS * LP);
    float *handm_out = P.handm + (long long)(pass * 2 + (band & 1)) * P.LLmax;
    const float *hand_in = P.hand + ((long long)(pass * 2 + ((band + 1) & 1)) * P.LLmax) * (NS * LP);
    const float *handm_in = P.handm + (long long)(pass * 2 + ((band + 1) & 1)) * P.LLmax;
    unsigned *prog_out = P.prog + pass * P.maxbands + band;
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
                        for (int k = 0; k < 4; k++) D[k] = P.w8[(long long)g.wplane[k] * P.npix + pix];
                        if constexpr (!FH) {
                            if (form == 0) combine_whirsch<LPL>(Cpf[u], nb_i, nb_s, nb_b, nb_f, D, P1, P2, MGM, Lv);
                            else combine_whirsch<LPL>(Cpf[u], nb_f, nb_b, nb_s, nb_i, D, P1, P2, MGM, Lv);
                        } else {
                            if (form == 0) combine_wfh<LPL>(Cpf[u], nb_i, nb_s, nb_b, nb_f, D, P1, P2, MGM, lane, L, Lv);
                            else combine_wfh<LPL>(Cpf[u], nb_f, nb_b, nb_s, nb_i, D, P1, P2, MGM, lane, L, Lv);
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

// ---- launcher ------------------------------------------------------------------
int pass_ns(bool fh, bool weighted) { return (weighted && !fh) ? 2 : 1; }
int pass_lpl(int L)
{
    const int lpl = (L + 63) / 64;
    if (lpl == 5) return 6;
    if (lpl == 7) return 8;
    return lpl;
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
