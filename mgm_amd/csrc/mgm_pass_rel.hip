// mgm_pass_rel.hip -- K3 for RAGGED volumes in a RANGE-PROPORTIONAL layout (round 5).
//
// The reference allocates and visits, per pixel, only the disparities of that pixel's own range
// (mgm_costvolume.h:275-299, dvec.cc:55-64): a coarse-to-fine caller that passes range images (-m/-M, or main()'s own
// TSGM_ITER loop) with windows of +-24 labels inside a hull of 256 pays for 49 labels, not 256.  The dense [y][x][hull]
// layout of the other kernels pays for the hull.  Here every pixel p carries a slab of 64 label SLOTS placed at its own
// window: slot k <-> disparity base(p) + k, base(p) = lo(p) - 1 (one empty slot below the window, at least one above:
// windows of up to 62 labels), costs as bytes (255 = +INF = "no such label", what a read past a Dvec returns,
// dvec.cc:129).  Bytes and steps follow sum_p(range), as in the reference.
//
// The recursion is the reference's (mgm_core.cc:489-579) with the update functions evaluated on the CONSUMER side --
// update_costW (95-144) / update_costW_trunclinear (229-281), which is also what the reference calls for unit weights
// with TSGM != 2 (563-575): a neighbour q publishes its raw slab L_q (Hirschmueller: and N_q[k] = min(L_q[k-1], L_q[k+1])),
// its minimum and its base; the receiving pixel reads slot k + base(p) - base(q) of it -- the SAME disparity -- and +INF
// where that falls outside q's 64 slots (q's window lies strictly inside them, so every label beyond is a label q does not
// have: Dvec::get returns INFINITY there).  FH potentials convolve over the receiving pixel's own range
// (mgm_core.cc:242-271: M[] is copied over [Lp.min, Lp.max]): combine_wfh masks to it.  Same neighbours, same operands,
// same order as the dense kernels on the hull => the same bits on every label that exists.
//
// Structure: the first build's (mgm_pass.hip) -- one wavefront per scan line, 15 lines per workgroup in lock-step on the
// slope-2 diagonal (+ a loader wave, see k_pass_rel), a 4-deep LDS ring per line (a slab is read by the next line at three consecutive steps, each time at
// another shift), the band hand-off through global memory with progress words, work items by atomic ticket.  One label
// slot per lane.  Not built here (the dense path keeps them): TSGM = 2 without weights (update_cost2 /
// update_cost2_trunclinear are other functions), windows wider than 62 labels, costs that are not bytes, P2 = +INF.
#include "mgm_pass_common.h"

namespace mgm {

constexpr int RR = 15;   // lines per band = compute waves of a workgroup (+ 1 loader wave)
constexpr int RD4 = 4;   // ring slots per line
constexpr int LD = 4;    // steps of global loads the loader wave keeps in flight

// ---- the relative copy of a ragged volume ------------------------------------------------------------------------------
// one thread per (pixel, slot): rel8[p][k] = byte code of C[p][base(p) + k - dmin] inside the pixel's window, 255 elsewhere;
// flag |= 1 if a window is wider than 62 labels, |= 2 if a cost has no byte form
__global__ void __launch_bounds__(256) k_rel_gather(const float *__restrict__ C, const float *__restrict__ rlo, const float *__restrict__ rhi, long long npix,
                                                    int L, int dmin, uint8_t *__restrict__ rel8, int *__restrict__ relb, unsigned *flag)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long p = t >> 6;
    const int k = (int)(t & 63);
    if (p >= npix) return;
    const int lo = (int)rlo[p], hi = (int)rhi[p];
    const int b = lo - 1;
    if (k == 0) {  // the pixel's record: disparity of slot 0, its own range
        relb[p * 4 + 0] = b;
        relb[p * 4 + 1] = lo;
        relb[p * 4 + 2] = hi;
        relb[p * 4 + 3] = 0;
    }
    unsigned bad = 0;
    if (hi - lo + 1 > 62 || hi < lo) bad |= 1u;
    const int d = b + k;  // disparity of this slot
    unsigned code = 255u;
    if (d >= lo && d <= hi && d - dmin >= 0 && d - dmin < L) {
        code = c8_encode(C[p * L + (d - dmin)]);
        if (code > 255u) {
            bad |= 2u;
            code = 255u;
        }
    }
    rel8[p * 64 + k] = (uint8_t)code;
    if (bad && *flag != (*flag | bad)) atomicOr(flag, bad);  // (look before raising: one word for everybody)
}
hipError_t launch_rel_gather(const float *C, const float *rlo, const float *rhi, long long npix, int L, int dmin, uint8_t *rel8, int *relb,
                             unsigned *flag, hipStream_t s)
{
    const long long n = npix * 64;
    hipLaunchKernelGGL(k_rel_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, C, rlo, rhi, npix, L, dmin, rel8, relb, flag);
    return hipGetLastError();
}

// write-through (sc1) stores in 16-byte pieces: narrow sc1 stores are one fabric write each (MI355X_MICROARCH.md, "stores of each
// flavour"; the second build's hand-off learned it in round 3) -- the band hand-off below leaves as 16 x 16 bytes per slab, not 64 x 4
typedef float relf4 __attribute__((ext_vector_type(4)));
typedef float relf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rel_st_sc1_x4(float *p, relf4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void rel_st_sc1_x2(float *p, relf2 v)
{
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// ---- the kernel ----------------------------------------------------------------------------------------------------------
// Memory side on a LOADER wave (the 16th of the workgroup), as in the second build: everything a step reads from memory --
// the cost bytes and records of the band's 15 pixels (one 16-byte load per lane: lane 4 r + c fetches bytes 16 c .. of line
// r's pixel), their edge weights, and the previous band's hand-off slab -- is requested LD steps ahead into registers and
// written to LDS rings one step ahead; the compute waves touch only LDS and issue stores, so nothing makes them wait for
// memory (the first version loaded in the compute waves and spent 4-9 us per step in vmcnt(0) drains: 19 ms for a
// 1920x1080 volume of 49-label windows; the step is now what its arithmetic costs).
// PUBE (unit weights, Hirschmueller): the transform does not depend on the reader then, so the producer publishes
// E[k] = fmin(fmin(L[k], N[k] + P1), m + P2) - m once (one slab; its "minimum" word carries FAR = (m + P2) - m, what the
// expression gives for a disparity the neighbour does not have: L = N = +INF there) and the reader only adds -- the
// terms of update_costW with DeltaI = 1 (mgm_core.cc:104-137; P1 * 1.0f is P1), in its order.
template <bool FH, bool PUBE>
__global__ void __launch_bounds__((RR + 1) * 64) k_pass_rel(const RelParams P)
{
    static_assert(!(FH && PUBE), "FH potentials convolve over the RECEIVING pixel's range: consumer side only");
    constexpr int NS = (FH || PUBE) ? 1 : 2;
    constexpr int HS = NS * 64 + 4;  // floats per hand-off slot: slab(s), minimum, base (+ 2 of padding: 16-byte pieces)
    using NbT = Nb<1, NS>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *ring = smem;                                   // [RR + 1][RD4][NS][64]   (row RR: the previous band's last line)
    float *ringm = ring + (RR + 1) * RD4 * NS * 64;       // [RR + 1][RD4]
    int *ringb = reinterpret_cast<int *>(ringm + (RR + 1) * RD4);  // [RR + 1][RD4]
    int *mring = ringb + (RR + 1) * RD4;                  // [4][RR][4]   records of the step's pixels: base, lo, hi
    float *wring = reinterpret_cast<float *>(mring + 4 * RR * 4);  // [4][RR][4]   edge weights of the step's pixels
    uint8_t *cring = reinterpret_cast<uint8_t *>(wring + 4 * RR * 4);  // [4][RR][64] cost bytes of the step's pixels
    int *s_task = reinterpret_cast<int *>(cring + 4 * RR * 64);

    const int tid = threadIdx.x, lane = tid & 63;
    const int r = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) *s_task = (int)atomicAdd(P.ticket, 1u);
    __syncthreads();
    const int2 tk = P.tasks[*s_task];
    const int vp = tk.x, band = tk.y;  // vp = volume*8 + pass
    const int pass = vp & (kMaxDirs - 1);
    const RelVolume &V = P.vol[vp / kMaxDirs];
    const PassGeom &g = P.g[pass];
    const int NL = g.NL, LL = g.LL, MGM = P.MGM, form = g.form;
    const float P1 = P.P1, P2 = P.P2;
    const long long istep = g.istep;
    const int nsteps = (LL + 1 + 2 * (RR - 1) + 3) / 4 * 4;

    float *hand_out = P.hand + ((long long)(vp * 2 + (band & 1)) * P.LLmax) * HS;
    const float *hand_in = P.hand + ((long long)(vp * 2 + ((band + 1) & 1)) * P.LLmax) * HS;
    unsigned *prog_out = P.prog + vp * P.maxbands + band;
    const unsigned *prog_in = prog_out - 1;  // only dereferenced when band > 0

    if (r == RR) {
        // =========================== loader wave ===========================
        const bool from_global = band > 0;
        const int cl = lane >> 2, cpart = lane & 3;                 // cost bytes: line, 16-byte piece
        const int cj = band * RR + (cl < RR ? cl : RR - 1);
        const bool c_ok = cl < RR && cj < NL;
        const long long cpix0 = g.base + (long long)(cj < NL ? cj : NL - 1) * g.jstep;
        const int ml = lane < RR ? lane : RR - 1;                   // records: lane = line
        const int mj = band * RR + ml;
        const bool m_ok = lane < RR && mj < NL;
        const long long mpix0 = g.base + (long long)(mj < NL ? mj : NL - 1) * g.jstep;
        const int wk = lane & 3;                                    // weights: line lane / 4, neighbour lane % 4 (the cost lanes' split)
        const long long wplane = (long long)g.wplane[wk] * P.npix;
        uint4 Cst[LD];
        int4 Mst[LD];
        float Wst[LD], Hst[2][NS], Hx[2];  // (the hand-off slab is requested TWO steps ahead only: every step of lead is a step of lag per band)
        // the producer's progress word, read WITHOUT waiting: requested every step, looked at two steps later -- in the steady state
        // (this band a hand-off lag behind its predecessor) that value already covers what the step needs and the blocking poll
        // below never runs (it did every fourth step -- the word moves in fours --, ~1.5 us of round trip each: a third of the step)
        unsigned Pst[2] = {0u, 0u};
        unsigned known = 0;
        bool dead = false;
        auto ensure = [&](unsigned need) {  // wait until the producer band has published slabs [0, need)
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
        auto issue = [&](int t, int u) {  // what step t reads, into register stage u
            {
                const int i = t - 1 - 2 * cl;
                const bool ok = c_ok && i >= 0 && i < LL;
                const long long pix = cpix0 + (long long)(ok ? i : 0) * istep;
                Cst[u] = ok ? *reinterpret_cast<const uint4 *>(V.c8 + pix * 64 + cpart * 16) : make_uint4(~0u, ~0u, ~0u, ~0u);
                Wst[u] = (ok && P.weighted) ? V.w8[wplane + pix] : 1.0f;
            }
            {
                const int i = t - 1 - 2 * ml;
                const bool ok = m_ok && i >= 0 && i < LL;
                const long long pix = mpix0 + (long long)(ok ? i : 0) * istep;
                Mst[u] = ok ? reinterpret_cast<const int4 *>(V.base)[pix] : make_int4(0, 0, 0, 0);
            }
        };
        auto issue_hand = [&](int t, int u) {  // the previous band's slab that step t reads
            if (from_global) {
                const int h = t;  // the first line of the band is at pixel t - 1 and reads pixel t of the line before it
                known = Pst[u] > known ? Pst[u] : known;  // (requested two steps ago)
                Pst[u] = __hip_atomic_load(prog_in, RLX_AGENT);
                if (h >= 0 && h < LL) {
                    ensure((unsigned)h + 1u);
                    const unsigned *src = reinterpret_cast<const unsigned *>(hand_in + (long long)h * HS);
#pragma unroll
                    for (int q = 0; q < NS; q++) Hst[u][q] = __builtin_bit_cast(float, __hip_atomic_load(src + q * 64 + lane, RLX_AGENT));
                    Hx[u] = __builtin_bit_cast(float, __hip_atomic_load(src + NS * 64 + (lane & 1), RLX_AGENT));
                }
            }
        };
        auto commit = [&](int t, int u) {  // register stage u -> the rings, for step t
            const int sl = t & 3;
            if (cl < RR) {
                *reinterpret_cast<uint4 *>(cring + ((sl * RR + cl) * 64 + cpart * 16)) = Cst[u];
                wring[(sl * RR + cl) * 4 + wk] = Wst[u];
            }
            if (lane < RR) *reinterpret_cast<int4 *>(mring + (sl * RR + lane) * 4) = Mst[u];
        };
        auto commit_hand = [&](int t, int u) {
            if (from_global && t >= 0 && t < LL) {
                const int slot = t & (RD4 - 1);
                float *dst = ring + ((RR * RD4 + slot) * NS) * 64 + lane;
#pragma unroll
                for (int q = 0; q < NS; q++) dst[q * 64] = Hst[u][q];
                if (lane == 0) ringm[RR * RD4 + slot] = Hx[u];
                if (lane == 1) ringb[RR * RD4 + slot] = __builtin_bit_cast(int, Hx[u]);
            }
        };
#pragma unroll
        for (int u = 0; u < LD; u++) issue(u, u);
        issue_hand(0, 0);
        issue_hand(1, 1);
        commit(0, 0);
        commit_hand(0, 0);
        issue(LD, 0);
        issue_hand(2, 0);
        lds_barrier();  // B0
        for (int s0 = 0; s0 < nsteps; s0 += LD) {
#pragma unroll
            for (int u = 0; u < LD; u++) {
                const int s = s0 + u;
                commit(s + 1, (u + 1) % LD);
                commit_hand(s + 1, (u + 1) % 2);
                issue(s + 1 + LD, (u + 1) % LD);
                issue_hand(s + 3, (u + 1) % 2);
                lds_barrier();
            }
        }
        return;
    }

    // ============================= compute waves =============================
    const int j = band * RR + r;
    const bool line_ok = j < NL;
    const bool has_prev = line_ok && (j >= 1);
    const bool to_global = (r == RR - 1) && (band + 1 < g.nbands);
    const int prow = r > 0 ? r - 1 : RR;  // ring row of the line before this one
    float *__restrict__ Lrb = V.Lr + (long long)(pass - P.pass0) * P.nvol;
    const long long pix0 = g.base + (long long)j * g.jstep;

    // neighbour `row`/`pixel n` of the pixel with base bp: the same disparities, +INF where n has no slot for them
    auto fetch = [&](int row, int n, int bp, NbT &nb) {
        const int slot = n & (RD4 - 1);
        const int sh = bp - ringb[row * RD4 + slot];
        const int idx = lane + sh;
        const bool in = (unsigned)idx < 64u;
        const float *src = ring + ((row * RD4 + slot) * NS) * 64;
        nb.m = ringm[row * RD4 + slot];
#pragma unroll
        for (int q = 0; q < NS; q++) nb.w[q][0] = in ? src[q * 64 + (in ? idx : 0)] : (PUBE ? nb.m : f_inf());
    };

    lds_barrier();  // B0: the loader's first step has landed
    for (int s = 0; s < nsteps; s++) {
        const int i = s - 1 - 2 * r;
        if (line_ok && i >= 0 && i < LL) {
            const long long pix = pix0 + (long long)i * istep;
            const int sl = s & 3;
            const int4 rec = *reinterpret_cast<const int4 *>(mring + (sl * RR + r) * 4);
            const int bp = rec.x;
            float Cv[1], Lv[1];
            Cv[0] = c8_decode((unsigned)cring[(sl * RR + r) * 64 + lane]);
            const bool interior = has_prev && i >= 1 && i <= LL - 2;  // mgm_core.cc:538-541
            if (interior) {
                NbT nb_i, nb_s, nb_b, nb_f;
                // (only the neighbours the update reads: MGM of them, in the pass's order)
                const bool f0 = form == 0;
                nb_i.w[0][0] = nb_s.w[0][0] = nb_b.w[0][0] = nb_f.w[0][0] = f_inf();
                if constexpr (NS == 2) nb_i.w[1][0] = nb_s.w[1][0] = nb_b.w[1][0] = nb_f.w[1][0] = f_inf();
                nb_i.m = nb_s.m = nb_b.m = nb_f.m = 0.0f;
                if (f0 || MGM >= 4) fetch(r, i - 1, bp, nb_i);
                if (f0 ? MGM >= 2 : MGM >= 3) fetch(prow, i, bp, nb_s);
                if (f0 ? MGM >= 3 : MGM >= 2) fetch(prow, i - 1, bp, nb_b);
                if (!f0 || MGM >= 4) fetch(prow, i + 1, bp, nb_f);
                const float4 w4 = *reinterpret_cast<const float4 *>(wring + (sl * RR + r) * 4);
                const float D[4] = {w4.x, w4.y, w4.z, w4.w};
                if constexpr (PUBE) {
                    // e = 0; e += t1 - m1; ... in the pass's order (0 + x is x: x >= +0); Lp = C + e / howmany
                    const NbT &n1 = f0 ? nb_i : nb_f, &n2 = f0 ? nb_s : nb_b, &n3 = f0 ? nb_b : nb_s, &n4 = f0 ? nb_f : nb_i;
                    float e = n1.w[0][0];
                    if (MGM >= 2) e += n2.w[0][0];
                    if (MGM >= 3) e += n3.w[0][0];
                    if (MGM >= 4) e += n4.w[0][0];
                    Lv[0] = Cv[0] + div_small_rt(e, MGM);
                } else if constexpr (!FH) {
                    if (f0) combine_whirsch<1>(Cv, nb_i, nb_s, nb_b, nb_f, D, P1, P2, MGM, Lv);
                    else combine_whirsch<1>(Cv, nb_f, nb_b, nb_s, nb_i, D, P1, P2, MGM, Lv);
                } else {
                    const int rl = rec.y - bp, rh = rec.z - bp;  // the pixel's own range, in slots
                    if (f0) combine_wfh<1>(Cv, nb_i, nb_s, nb_b, nb_f, D, P1, P2, MGM, lane, 64, Lv, rl, rh);
                    else combine_wfh<1>(Cv, nb_f, nb_b, nb_s, nb_i, D, P1, P2, MGM, lane, 64, Lv, rl, rh);
                }
            } else {
                Lv[0] = Cv[0];
            }
            Lrb[pix * 64 + lane] = Lv[0];

            // what this pixel publishes: its raw slab (Hirschmueller: and the neighbour minima), minimum, base
            const float m = slab_min<1>(Lv);
            float N[1] = {f_inf()};
            if constexpr (!FH) neighbour_min<1>(Lv, N);
            float pub0 = Lv[0], pubm = m;  // what goes out: the raw slab and its minimum -- or (PUBE) E and FAR
            if constexpr (PUBE) {
                const float cap = m + P2;
                pub0 = fminf(fminf(Lv[0], N[0] + P1), cap) - m;
                pubm = cap - m;
            }
            {
                const int slot = i & (RD4 - 1);
                float *dst = ring + ((r * RD4 + slot) * NS) * 64 + lane;
                dst[0] = pub0;
                if constexpr (NS == 2) dst[64] = N[0];
                if (lane == 0) {
                    ringm[r * RD4 + slot] = pubm;
                    ringb[r * RD4 + slot] = bp;
                }
            }
            if (to_global) {
                // the slab(s) just written to the ring, read back as 16-byte pieces by 16 lanes per slab (same wave: the LDS
                // write has retired) and stored write-through; minimum and base as one 8-byte piece
                float *dstg = hand_out + (long long)i * HS;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane < 16 * NS) {
                    const float *srcl = ring + ((r * RD4 + (i & (RD4 - 1))) * NS) * 64 + lane * 4;
                    const relf4 v = {srcl[0], srcl[1], srcl[2], srcl[3]};
                    rel_st_sc1_x4(dstg + lane * 4, v);
                }
                if (lane == 0) {
                    const relf2 v = {pubm, __builtin_bit_cast(float, bp)};
                    rel_st_sc1_x2(dstg + NS * 64, v);
                }
                // Progress is published PL steps LATE, every fourth pixel: this wave only issues stores, they retire in order,
                // so once at most PL steps' worth of them are outstanding every store of pixel i - PL has reached memory -- a
                // counted wait instead of draining the queue (which stalled the whole band for a store round trip every eighth
                // step, and the band behind it for up to eight pixels more)
                constexpr int SPS = 1 + 1 + 1 + 1, PL = 3;  // store instructions per step (Lr, slab pieces, minimum + base; + the word itself)
                if (i == LL - 1) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(prog_out, (unsigned)LL, RLX_AGENT);
                } else if (i >= PL && ((i - PL + 1) & 3) == 0) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PL * SPS) : "memory");
                    if (lane == 0) __hip_atomic_store(prog_out, (unsigned)(i - PL + 1), RLX_AGENT);
                }
            }
        }
        // everybody's slab for this step is in LDS before anyone reads it
        lds_barrier();
    }
}

template <bool FH, bool PUBE>
static hipError_t launch_rel_one(const RelParams &p, int ntasks, bool one_per_cu, hipStream_t s)
{
    constexpr int NS = (FH || PUBE) ? 1 : 2;
    size_t shmem = sizeof(float) * ((size_t)(RR + 1) * RD4 * NS * 64 + 2 * (RR + 1) * RD4 + 2 * 4 * RR * 4) + 4 * RR * 64 + 16;
    // Occupancy through the LDS request, as for the second build: two of these workgroups fit a CU and each then steps ~1.7x
    // slower -- right for a batch (throughput), wrong for a launch bound by its chains of bands (one or two volumes)
    if (one_per_cu && shmem < 81 * 1024) shmem = 81 * 1024;
    auto kern = k_pass_rel<FH, PUBE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(ntasks), dim3((RR + 1) * 64), shmem, s, p);
    return hipGetLastError();
}
// pube: unit weights with Hirschmueller potentials (the producer publishes E)
hipError_t launch_pass_rel(const RelParams &p, int ntasks, bool fh, bool pube, bool one_per_cu, hipStream_t s)
{
    if (fh) return launch_rel_one<true, false>(p, ntasks, one_per_cu, s);
    return pube ? launch_rel_one<false, true>(p, ntasks, one_per_cu, s) : launch_rel_one<false, false>(p, ntasks, one_per_cu, s);
}
int pass_rel_lines() { return RR; }
int pass_rel_hand_floats(bool one_slab) { return (one_slab ? 1 : 2) * 64 + 4; }

}  // namespace mgm
