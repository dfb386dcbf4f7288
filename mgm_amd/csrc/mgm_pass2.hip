// mgm_pass2.hip -- K3, second build: the same recursion and dataflow as
// mgm_pass.hip (read its header first), with the memory side moved off the
// compute waves:
//
//   * a workgroup = NC compute waves (one scan line each, lock-step on the
//     slope-2 diagonal) + NL loader waves;
//   * the loader waves stream every operand that comes from memory -- the C
//     slab of each line D steps ahead, and the previous band's hand-off slabs
//     -- into LDS rings with LDS-DMA (global_load_lds_dwordx4, no VGPR round
//     trip) and retire them with a COUNTED s_waitcnt vmcnt(n*(D-1)) followed
//     by the step barrier, so D-1 steps of loads stay in flight across every
//     barrier;
//   * compute waves only read LDS and issue stores (their Lr slab, and for the
//     last line of the band the sc1 hand-off), so the compiler has no load to
//     wait for: the first build lost ~1.5 us per step to vmcnt(0) drains the
//     compiler placed on in-flight prefetch registers;
//   * between bands the unweighted kernels hand over SELF-VALIDATING slabs (the
//     launch's tag in the sign bit of every word, one slot per band: see TAGS in
//     k_pass2) -- no progress words, no publication lag; the weighted kernels
//     keep the first build's progress-word protocol;
//   * chain-bound launches walk the lines of passes 4-7 (2 or 3 neighbours: no
//     in-line dependency) as TWO strips from the image edges inwards (see
//     `strips` in pass2_item), and their workgroups stay and work per-XCD queues
//     of work items off (see k_pass2, XCDQ).
//
// Used when every slab is a whole number of 16-byte DMA pieces (L == 64*LPL,
// LPL in {1,2,3,4,6,8}); other label counts run padded to the next such count,
// or -- direction-sharded odd label counts, more than 512 labels, negative
// penalties -- take the first build.
#include <type_traits>

#include "mgm_pass_common.h"

namespace mgm {

typedef __attribute__((address_space(3))) void *lds_vptr;
typedef const __attribute__((address_space(1))) void *glb_vptr;

// One 16-byte-per-lane LDS-DMA piece: lane l moves src[l*4 .. l*4+3] to
// dst_base[l*4 ..] (LDS destination = wave-uniform base + 16*lane).
template <int AUX>
__device__ __forceinline__ void dma16(const float *src_lane, float *dst_base)
{
    __builtin_amdgcn_global_load_lds((glb_vptr)src_lane, (lds_vptr)dst_base, 16, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ void dma4(const void *src_lane, void *dst_base)
{
    __builtin_amdgcn_global_load_lds((glb_vptr)src_lane, (lds_vptr)dst_base, 4, 0, AUX);
}
constexpr int AUX_SC1 = 16;  // agent-scope (L1-bypassing) cache policy bit

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// (timeline words go out through a GLOBAL-address-space pointer made from the integer value, and the switch is an int: a
// null test of the generic pointer inside the queue kernels trips this compiler -- "Illegal instruction detected: Operand
// has incorrect register class", V_CMP_NE_U32 on src_shared_base -- as the development timers did, mgm_pass2.hip:launch2_one)
__device__ __forceinline__ void tl_store(const PassParams &P, long long idx, unsigned long long v)
{
    typedef __attribute__((address_space(1))) unsigned long long *gptr;
    gptr g = (gptr)(unsigned long long)P.tl_addr;
    g[idx] = v;
}
__device__ __forceinline__ void step_barrier(bool skip = false)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!skip) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// LDS read the compiler must not order against pending LDS-DMA itself (it would
// drain vmcnt): the landing of the word is guaranteed by the counted wait.
__device__ __forceinline__ unsigned lds_read_u32_opaque(const unsigned *p)
{
    unsigned v;
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned *)p;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 lds_read_b128_opaque(const float *p)
{
    u32x4 v;
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float *)p;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}

// Write-through (sc1) slab store in the widest pieces the slab allows: narrow sc1 stores are one
// fabric write each, so 4 x dword costs ~6x the time of one dwordx4 (MI355X_MICROARCH.md).
// Inline asm because clang has no 16-byte agent-scope store; the trailing s_nop keeps the data
// registers intact until the store has read them.  Not counted by the compiler's vmcnt
// bookkeeping -- compute waves issue no loads, and every wait on these stores is explicit.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_sc1_x4(float *p, float a, float b, float c, float d)
{
    const f32x4 v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_sc1_x2(float *p, float a, float b)
{
    const f32x2 v = {a, b};
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_sc1_x1(float *p, float a)
{
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(a) : "memory");
}
template <int LPL>
__device__ __forceinline__ void store_slab_sc1_wide(float *slab, int lane, const float (&v)[LPL])
{
    float *p = slab + lane * LPL;
    if constexpr (LPL == 1) st_sc1_x1(p, v[0]);
    else if constexpr (LPL == 2) st_sc1_x2(p, v[0], v[1]);
    else if constexpr (LPL == 3) { st_sc1_x2(p, v[0], v[1]); st_sc1_x1(p + 2, v[2]); }
    else if constexpr (LPL == 4) st_sc1_x4(p, v[0], v[1], v[2], v[3]);
    else if constexpr (LPL == 6) { st_sc1_x4(p, v[0], v[1], v[2], v[3]); st_sc1_x2(p + 4, v[4], v[5]); }
    else {
        static_assert(LPL % 4 == 0, "whole 16-byte pieces");
#pragma unroll
        for (int k = 0; k < LPL; k += 4) st_sc1_x4(p + k, v[k], v[k + 1], v[k + 2], v[k + 3]);
    }
}
// the same slab with plain stores (XCDQ: the reader is on this XCD)
template <int LPL>
__device__ __forceinline__ void store_slab_plain(float *slab, int lane, const float (&v)[LPL])
{
    float *p = slab + lane * LPL;
    if constexpr (LPL % 4 == 0) {
#pragma unroll
        for (int k = 0; k < LPL; k += 4) {
            const f32x4 q = {v[k], v[k + 1], v[k + 2], v[k + 3]};
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p + k), "v"(q) : "memory");
        }
    } else {
#pragma unroll
        for (int k = 0; k < LPL; k++) asm volatile("global_store_dword %0, %1, off\n\ts_nop 1" ::"v"(p + k), "v"(v[k]) : "memory");
    }
}
template <int LPL>
constexpr int sc1_store_count() { return LPL == 3 || LPL == 6 ? 2 : (LPL + 3) / 4; }

#ifndef MGM_P2_WAVES_PER_EU
#define MGM_P2_WAVES_PER_EU 8   // (compact unweighted kernels only; every other kernel is built for 4)
#endif
#ifndef MGM_P2_ONEB_WPE
#define MGM_P2_ONEB_WPE 4       // ... the queue kernels of launches that run ONE band per CU (k_pass2, ONEB)
#endif
#ifndef MGM_P2_NC
#define MGM_P2_NC 14
#endif
#ifndef MGM_P2_LDS_KB
#define MGM_P2_LDS_KB 78
#endif
#ifndef MGM_P2_DEV
#define MGM_P2_DEV 0   // 1: in-kernel timers and experiment switches (development builds: MGM_P2_DEFINES=-DMGM_P2_DEV=1);
#endif                 // the run-time tests alone cost the issue-bound FH kernel 6 %, so product builds compile them out

#ifndef MGM_P2_TIMELINE
#define MGM_P2_TIMELINE 0  // 1: the queue kernels record, per work item, start / end / time waited for the predecessor band / where it
#endif                     // ran (PassParams::tl; MGM_HIP_TIMELINE=<file>; tools/timeline.py).  Variant builds only: tools/sweep_build.sh tl -DMGM_P2_TIMELINE=1
#ifndef MGM_P2_C8_NL
#define MGM_P2_C8_NL 1
#endif
#ifndef MGM_P2_C8_NC
#define MGM_P2_C8_NC (16 - MGM_P2_C8_NL)   // lines per band with compact costs (tuning experiments: 7)
#endif
#ifndef MGM_P2_MAXD
#define MGM_P2_MAXD 2   // steps of DMA in flight, the shallow build (throughput-bound launches of many volumes)
#endif
#ifndef MGM_P2_DEEPD
#define MGM_P2_DEEPD 4  // ... the deep build: launches in which a band's step waits for its DMA (see DEEP in k_pass2)
#endif
#ifndef MGM_P2_PUBLAG
#define MGM_P2_PUBLAG 3
#endif
#ifndef MGM_P2_PUBEVERY
#define MGM_P2_PUBEVERY 4
#endif
#ifndef MGM_P2_LEAD
#define MGM_P2_LEAD 2
#endif
// ---- geometry of the build --------------------------------------------------------
template <int LPL, int NS, bool HASM, int C8, int MAXD = MGM_P2_MAXD, int EXTRA = 0>
struct Plan {
    static constexpr int LP = LPL * 64;
    static constexpr int IPS = (LPL * 16 + 63) / 64;  // DMA pieces per fp32 slab
    // compact costs: a slab is LP bytes = LPS lanes of one DMA piece, so one 64-lane piece carries the
    // slabs of LPD different lines (every lane has its own source address)
    // (C8 = bytes per compact cost: 1, or 2 since round 4 -- 0: fp32 costs)
    static constexpr int LPS = LPL * 4 * (C8 ? C8 : 1);
    static constexpr int LPD = C8 ? 64 / LPS : 1;
    static constexpr int NL = (LPL <= 4 && MGM_P2_NC > 7) ? (C8 ? MGM_P2_C8_NL : 2) : 1;  // loader waves
    static constexpr int NC = (LPL <= 4) ? (C8 ? MGM_P2_C8_NC : MGM_P2_NC) : 7;   // compute waves = lines per band
    static constexpr int NCA = (NL == 2) ? NC / 2 : NC;                               // lines served by loader A
    static constexpr int NDMA = C8 ? (NC + LPD - 1) / LPD : NCA * IPS;                // C pieces per step (loader A)
    // DMA instructions per step: loader A = its C pieces + hand-off slabs [+ minimum + progress word: the kernels whose
    // hand-off slabs are not self-validating, see TAGS in k_pass2]
    // compact costs with two loaders: A = hand-off only, B = all C pieces
    // (EXTRA: one more 4-byte piece per step -- the weight-selector words of the lines, k_pass2 W2)
    static constexpr int nA = ((C8 && NL == 2) ? 0 : NDMA) + NS * IPS + (HASM ? 2 : 0) + EXTRA;
    static constexpr int nB = (C8 && NL == 2) ? NDMA : (NC - NCA) * IPS;
    // Ring geometry: RT = T-ring slots per line (2 with barriers), RDEPTH = steps of C / hand-off data the rings
    // hold, D = steps of DMA kept in flight (D <= RDEPTH-1).  The largest of a few candidates that fits in LDS.
    static constexpr int cring_floats(int rdepth) { return C8 ? rdepth * NDMA * 256 : NC * rdepth * LP; }
    static constexpr int lds_floats3(int rt, int rdepth)
    {
        return NC * rt * NS * LP       // T ring
               + NC * rt               // T minima
               + rdepth * NS * LP      // hand-off ring
               + 2 * rdepth + 8 + 32   // hand-off minima, progress words, task word (+ spare)
               + cring_floats(rdepth)  // C ring
               + (C8 ? rdepth * 16 : 0)   // per line and ring slot: does the compact slab hold a +INF code?
               + EXTRA * rdepth * 64;     // per line and ring slot: the pixel's weight-selector word (W2)
    }
    static constexpr bool fits(int rt, int rdepth, int d)
    {
        return d >= 2 && d <= rdepth - 1 && lds_floats3(rt, rdepth) * 4 <= ((C8 && LPL <= 4 && NS == 1) ? MGM_P2_LDS_KB : 160) * 1024 &&
               nA * (d - 1) <= 63 && nB * (d - 1) <= 63;
    }
    static constexpr int pick(int what)  // 0: RT, 1: RDEPTH, 2: D
    {
        const int rts[2] = {2, 2};
        const int rds[3] = {MAXD + 1, 4, 3};
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 3; b++)
                for (int d = MAXD; d >= 2; d--)
                    if (fits(rts[a], rds[b], d)) return what == 0 ? rts[a] : (what == 1 ? rds[b] : d);
        return 0;
    }
    static constexpr int RT = pick(0), RDEPTH = pick(1), D = pick(2);
    static constexpr int rd(int) { return RDEPTH; }
    static constexpr int lds_floats(int) { return lds_floats3(RT, RDEPTH); }
    static_assert(D >= 2, "no feasible pipeline depth");
    static_assert(!C8 || LPD >= 1, "a compact slab must fit a DMA piece");  // (64 % LPS lanes of a piece may idle: L = 192, 384)
    static_assert((RT & (RT - 1)) == 0, "RT must be a power of two");
};

// unit weights, slabs hold E = T - m (every unit-weight case but FH with MGM == 2).  E >= +0, so the reference's
// `e = 0; e += ...` needs no addition for its first term (+0 + x == x bit for bit when x >= +0).  With MGM == 2
// (update_cost2: (t_1 - m_1)/2 + (t_2 - m_2)/2) the slabs hold the HALVES, taken once by the line that publishes them
// instead of once per reader -- the same correctly rounded value either way.
template <int LPL, int MGM, bool FH>
__device__ __forceinline__ void combine_unit_E(const float (&C)[LPL], const float (&e1)[LPL], const float (&e2)[LPL],
                                               const float (&e3)[LPL], const float (&e4)[LPL], float (&out)[LPL])
{
    if constexpr (MGM == 2) {  // update_cost2 (FH with MGM == 2 never gets here)
#pragma unroll
        for (int k = 0; k < LPL; k++) out[k] = C[k] + (e1[k] + e2[k]);
    } else {
#pragma unroll
        for (int k = 0; k < LPL; k++) {
            float e = e1[k];
            if constexpr (MGM >= 3) e += e2[k];
            if constexpr (MGM >= 3) e += e3[k];
            if constexpr (MGM >= 4) e += e4[k];
            out[k] = C[k] + div_small<MGM>(e);
        }
    }
}

// TWO-VALUED WEIGHTS (k_pass2, W2).  update_costW / update_costW_trunclinear (mgm_core.cc:95-144, 229-281) scale the penalties of
// neighbour k by the weight D_k of the RECEIVING pixel's edge, so the transform of a slab depends on its reader -- the general
// weighted kernels therefore hand over L (and N) and transform on the consumer side, three or four times per pixel.  But the
// weights the reference itself makes (compute_mgm_weights, mgm_weights.h:63-85: aP2 where the image is flat, 1 elsewhere)
// take TWO values, 1 and a: the producer can publish both transforms, E_1 = T(L; P1, P2) - m and E_a = T(L; P1*a, P2*a) - m
// (P1*a and P2*a rounded once, as the reference's products are), and every reader picks one per neighbour by a bit of its
// own pixel's selector word.  What is handed over is E >= +0 again: self-validating slabs, deep rings, per-XCD queues and
// compact costs all apply, and the consumer's update is the unweighted one.  The association is update_costW's for every
// TSGM, 2 included: e = ((0 + e_1) + e_2) + ..., Lp = C + e / TSGM.
template <int LPL, int MGM>
__device__ __forceinline__ void combine_w2(const float (&C)[LPL], const Nb<LPL, 2> &n1, const Nb<LPL, 2> &n2, const Nb<LPL, 2> &n3,
                                           const Nb<LPL, 2> &n4, bool a1, bool a2, bool a3, bool a4, float (&out)[LPL])
{
#pragma unroll
    for (int k = 0; k < LPL; k++) {
        float e = a1 ? n1.w[1][k] : n1.w[0][k];
        if constexpr (MGM >= 2) e += a2 ? n2.w[1][k] : n2.w[0][k];
        if constexpr (MGM >= 3) e += a3 ? n3.w[1][k] : n3.w[0][k];
        if constexpr (MGM >= 4) e += a4 ? n4.w[1][k] : n4.w[0][k];
        out[k] = C[k] + div_small<MGM>(e);
    }
}

// SUBV > 1: the wave's 256 label slots hold the slabs of SUBV different VOLUMES of the launch (128 labels: 2, 64: 4),
// same pass, same line, same pixel -- every lane group walks its own volume, nothing crosses between them, and a step
// that is mostly fixed cost (barrier, LDS round trips, DMA issue) serves SUBV volumes.  The LDS rings, the hand-off
// slabs and the compact-cost pieces simply carry [volume 0 | volume 1 | ...]; what differs per lane group is the base
// pointers, the slab minimum and the edge lanes of the label neighbourhood / of the FH scans.  LPL = 4, compact
// costs, no weights, the kernels that publish E (everything but FH with TSGM = 2).
// DEEP: the rings hold MGM_P2_DEEPD steps of DMA instead of MGM_P2_MAXD.  With two steps in flight a load has ONE step to
// land, and a step that is shorter than the memory latency (~0.8 us under load) waits for it: measured (round 3, same
// box, shallow -> deep) 4096x4096x192 x 1: K3 32.5 -> 27.7 ms; 1920x1080x128 x 1: 2.69 -> 2.34; 256 labels x 1: 5.71 -> 5.50
// (Hirschmueller), 7.23 -> 7.03 (FH); x 2: 11.24 -> 10.95; eight / sixteen 128-label volumes 9.11 -> 8.69 / 16.6 -> 15.9;
// twelve 256-label volumes 48.5 -> 48.2.  The default of every compact unweighted launch (mgm_api.hip, run_passes); the
// shallow build stays for A/B runs (MGM_HIP_DEEP=0).
// where the work-item word lives in the workgroup's LDS (the layout of pass2_item, below)
template <int LPL, bool FH, bool WEIGHTED, int MGM, int C8, bool DEEP, bool W2 = false>
struct P2Lds {
    static constexpr int NS = (W2 || (WEIGHTED && !FH)) ? 2 : 1;
    static constexpr bool pubE = W2 || (!WEIGHTED && !(FH && MGM == 2));
    using PL = Plan<LPL, NS, !pubE, C8, DEEP ? MGM_P2_DEEPD : MGM_P2_MAXD, W2 ? 1 : 0>;
    static constexpr int RD = PL::rd(PL::D);
    static constexpr int task_off = PL::NC * PL::RT * NS * PL::LP + RD * NS * PL::LP + PL::cring_floats(RD) + PL::NC * PL::RT + RD + RD;  // floats
};

template <int LPL, bool FH, bool WEIGHTED, int MGM, int C8, int SUBV, bool DEEP, bool XCDQ, bool W2 = false>
__device__ __forceinline__ void pass2_item(const PassParams &P, const int ticket)
{
    static_assert(!W2 || (!WEIGHTED && C8 && DEEP && SUBV == 1 && LPL <= 4), "two-valued weights: the compact kernels with deep rings");
    static_assert(!DEEP || (C8 && !WEIGHTED && (W2 || !(FH && MGM == 2))), "the deep rings exist for the compact kernels that publish E");
    static_assert(SUBV == 1 || (LPL == 4 && C8 == 1 && !WEIGHTED && !(FH && MGM == 2)), "volumes share a wave only in the compact unweighted kernels that publish E");
    static_assert(!W2 || C8 == 1, "two-valued weights: one byte per cost");
    constexpr int LANES = 64 / SUBV;  // lanes per volume
    constexpr int NS = (W2 || (WEIGHTED && !FH)) ? 2 : 1;
    constexpr bool pubE = W2 || (!WEIGHTED && !(FH && MGM == 2));  // slabs carry E = T - m; minima not needed
    // Inter-band hand-off of the kernels that publish E.  E >= +0 always (T >= m; the host sends negative penalties to
    // the first build), so the sign bit of every word is free: the last line of a band stores its slabs with the sign
    // bits set to the LAUNCH's tag bit.  Every band has its own hand-off slots, written exactly once per launch, and
    // consecutive launches on the same slots alternate the tag (the host clears the region whenever the geometry
    // changes), so whatever a slot held before carries the other sign.  The consumer's loader simply DMAs the slab it
    // needs next and looks at the signs of what landed: every word validates itself -- no progress word, no "stores
    // have landed" wait on the producer side, no publication lag: a band trails its predecessor by the visibility
    // latency of the stores plus the DMA depth (~4 steps instead of ~10).  A stale slab is fetched again until it is
    // valid (slow path).  No assumption on timing or placement: a slot can only ever hold this launch's slab or an
    // older launch's.
    constexpr bool TAGS = pubE;
    // Loader-side +INF flags of the compact cost slabs (see flag_inf): only where the step is long enough for the loader to
    // have the time -- the FH kernels (cfg3 x 12: K3 50.1 -> 48.9 ms); the Hirschmueller kernels, whose loader is on the
    // critical path of a short step, lose 8-25 % with it (cfg2 x 16, cfg4) and keep decoding every byte.
    constexpr bool CFLAG = C8 && FH;
    using PL = Plan<LPL, NS, !pubE, C8, DEEP ? MGM_P2_DEEPD : MGM_P2_MAXD, W2 ? 1 : 0>;
    static_assert(!W2 || PL::NL == 1, "two-valued weights: one loader wave");
    constexpr int LP = PL::LP, NC = PL::NC, NCA = PL::NCA, D = PL::D, IPS = PL::IPS;
    constexpr int LPS = PL::LPS, LPD = PL::LPD, NDMA = PL::NDMA;
    constexpr int LPW = NCA;   // C lines per loader wave (NC - NCA == NCA when there are two loaders)
    constexpr int RD = PL::rd(D);  // C / hand-off ring depth (steps)
    constexpr int RT = PL::RT;     // T-ring slots per line
    using NbT = Nb<LPL, NS>;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Tring = smem;                          // [NC][RT][NS][LP]
    float *Hring = Tring + NC * RT * NS * LP;     // [RD][NS][LP]
    float *Cring = Hring + RD * NS * LP;          // fp32: [NC][RD][LP]; compact: [RD][NDMA][1 KiB]
    float *Tm = Cring + PL::cring_floats(RD);     // [NC][RT]
    float *Hm = Tm + NC * RT;                     // [RD]
    unsigned *Hprog = reinterpret_cast<unsigned *>(Hm + RD);  // [RD]
    int *s_task = reinterpret_cast<int *>(smem + P2Lds<LPL, FH, WEIGHTED, MGM, C8, DEEP, W2>::task_off);  // (= Hprog + RD; the kernel's ticket word)
    unsigned *Cflag = reinterpret_cast<unsigned *>(s_task + 40);  // [RD][16] (compact costs; behind the spare words)
    unsigned *Wring = Cflag + RD * 16;                             // [RD][64] (W2: lane r = line r's weight-selector word)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int2 tk = P.tasks[ticket];
    const int vp = tk.x, band = tk.y & 0xffff, strip = (tk.y >> 16) & 0xff;  // vp = volume*8 + pass
    const bool plain_out = XCDQ && ((tk.y >> 24) & 1) != 0;  // the next band of the chain runs on this XCD (see k_pass2)
    const int pass = vp & (kMaxDirs - 1);
    const int vgrp = (vp / kMaxDirs) * SUBV;  // first volume of this work item
    const PassVolume &V = P.vol[vgrp];
#if MGM_P2_DEV  // in-kernel timers (MGM_HIP_DEBUG_STATS) and experiment switches (MGM_HIP_XFLAGS)
    unsigned long long *dbg = P.dbg ? P.dbg + (long long)ticket * 16 : nullptr;
    const int xflags = P.xflags;
#else
    constexpr unsigned long long *dbg = nullptr;
    constexpr int xflags = 0;
#endif
    if (dbg && tid == 0) dbg[0] = wall_clock64();
    const PassGeom &g = P.g[pass];
    const int NLn = g.NL, LL = g.LL, L = P.L, form = g.form;
    const float P1 = P.P1, P2 = P.P2;
    const int SL = g.slope;  // wave r is at pixel i = s - 1 - SL*r at step s
    // Two strips per line (form 1 with 2 or 3 neighbours: a pixel depends on the previous line only): this workgroup walks
    // strip `strip` of its band's lines, strip 0 = [0, split) upwards from pixel 0, strip 1 = [split, LL) downwards from
    // pixel LL-1 -- in MIRRORED coordinates i' = LL-1-i, in which it is again a walk upwards from 0 (the two neighbours
    // that swap roles, fwd and back, enter a commutative fp32 sum).  Both strips run from the image edge inwards and meet
    // in the middle; so that neither ever needs the other's lines of the SAME band, line r of the band is walked over
    // [0, W - r), W = strip length + NC-1: the last line covers exactly the strip, the lines above it a little more (the
    // pixels both workgroups compute get the same values twice).  What a strip needs from the other comes from the
    // previous band's hand-off slots, which are indexed by absolute pixel and validate themselves.  The chain of a pass
    // is then 2 steps per line + HALF a line, and a band lives half as long.
    const bool strips = TAGS && g.nstrips == 2;
    const bool mirror = strips && strip == 1;
    const long long istep = mirror ? -g.istep : g.istep;
    const long long gbase = mirror ? g.base + (long long)(LL - 1) * g.istep : g.base;
    const int W = !strips ? LL : min(LL, (strip == 0 ? g.split : LL - g.split) + NC - 1);  // pixels of the tile's first line
    const int nsteps = ((W + 1 + SL * (NC - 1)) + 2) / 3 * 3;
    const bool from_global = band > 0;

    constexpr int NSLP = NS * LP;
    // flag protocol: two slots per pass, alternating with the band's parity; TAGS: one slot per band
    const long long hslab = (long long)(vp / kMaxDirs) * P.hand_vstride + g.hand_base;
    float *hand_out = TAGS ? P.hand + (hslab + (long long)band * LL) * NSLP : P.hand + ((long long)(vp * 2 + (band & 1)) * P.LLmax) * NSLP;
    float *handm_out = P.handm + (long long)(vp * 2 + (band & 1)) * P.LLmax;
    // (band 0 has no predecessor; its loader still issues the DMAs -- every step the same count -- from its own slots)
    const float *hand_in = TAGS ? P.hand + (hslab + (long long)(band > 0 ? band - 1 : 0) * LL) * NSLP
                                : P.hand + ((long long)(vp * 2 + ((band + 1) & 1)) * P.LLmax) * NSLP;
    const float *handm_in = P.handm + (long long)(vp * 2 + ((band + 1) & 1)) * P.LLmax;
    unsigned *prog_out = P.prog + vp * P.maxbands + band;
    const unsigned *prog_in = from_global ? prog_out - 1 : prog_out;

    const unsigned tag_in = P.hand_tag[pass], tag_out = tag_in;  // sign bits of the slabs this pass hands over in this launch

    if (wave >= NC) {
        // =========================== loader waves ===========================
        const int wl = wave - NC;
        const int r0 = wl == 0 ? 0 : NCA;
        unsigned known = 0;
        bool dead = false;
        unsigned long long n_slow = 0, t_slow = 0, n_spin = 0, t_ret = 0, t_bar = 0, t_iss = 0, tl_wait = 0;
        (void)tl_wait;
        (void)n_slow;
        (void)n_spin;

        // Per C piece: the (clamped) source pointer of the NEXT target step and its pixel index.
        // Target step t wants pixel i = t-1-SL*r of line r; out-of-range steps re-read an end pixel
        // (the slot they fill is never consumed) so that every step issues the same DMA count.
        // fp32 costs: one piece = one line's slab (lane l moves bytes 16l..16l+15 of it).
        // compact costs: one piece = the slabs of LPD lines; lane l serves line q*LPD + l/LPS.
        constexpr int NPIECE = C8 ? NDMA : LPW;
        const float *cptr[NPIECE];
        int ci[NPIECE];
        const long long cstride = C8 ? (istep * L * C8) / 4 : istep * L;  // in floats (compact: L * C8 bytes per pixel)
#pragma unroll
        for (int q = 0; q < NPIECE; q++) {
            int r = C8 ? q * LPD + lane / LPS : r0 + q;
            r = r < NC ? r : NC - 1;
            int j = band * NC + r;
            j = j < NLn ? j : NLn - 1;
            if constexpr (C8 && SUBV > 1) {
                constexpr int CPV = LPS / SUBV;  // 16-byte chunks of a line's slab that belong to one volume
                const int chunk = lane % LPS;
                const uint8_t *c8 = P.vol[vgrp + chunk / CPV].C8;
                cptr[q] = reinterpret_cast<const float *>(c8 + (gbase + (long long)j * g.jstep) * L + (chunk % CPV) * 16);
            } else if constexpr (C8 != 0)
                cptr[q] = reinterpret_cast<const float *>(V.C8 + (gbase + (long long)j * g.jstep) * L * C8 + (lane % LPS) * 16);
            else
                cptr[q] = V.C + (gbase + (long long)j * g.jstep) * L + lane * 4;
            ci[q] = -1 - SL * r;
        }
        // W2: the weight-selector word of every line's pixel travels like the costs -- lane r fetches line r's word of the
        // target step (4 bytes per lane, one DMA instruction per step)
        const unsigned *wptr = nullptr;
        int wi = 0;
        if constexpr (W2) {
            const int r = lane < NC ? lane : NC - 1;
            int j = band * NC + r;
            j = j < NLn ? j : NLn - 1;
            wptr = V.wsel + (gbase + (long long)j * g.jstep);
            wi = -1 - SL * r;
        }
        // hand-off slab wanted by wave 0 at step t: pixel t (its fwd neighbour) with slope 2, pixel t-1
        // (its same neighbour) with slope 1; clamped to [0, LL-1]
        const int Hmax = W < LL ? W : LL - 1;  // last pixel of the previous band's last line this tile reads
        const long long hstep = mirror ? -(long long)NSLP : (long long)NSLP;  // (slots are indexed by the absolute pixel)
        const float *hptr = hand_in + (mirror ? (long long)(LL - 1) * NSLP : 0) + lane * 4;
        const float *hmptr = handm_in;
        int ht = SL == 2 ? 0 : -1;

        auto issue = [&](int slot) {  // everything the step `ht` needs, into ring slot `slot`
            const bool c_duty = !(C8 && PL::NL == 2) || wl == 1;  // with compact costs and two loaders, B fetches C
#pragma unroll
            for (int q = 0; q < NPIECE; q++) {
                if (!c_duty) break;
                if constexpr (C8) {
                    dma16<0>(cptr[q], Cring + (slot * NDMA + q) * 256);
                } else {
                    float *dst = Cring + ((r0 + q) * RD + slot) * LP;
#pragma unroll
                    for (int c = 0; c < IPS; c++)
                        if (c * 64 + lane < ((xflags & 2) ? 1 : LPL * 16)) dma16<0>(cptr[q] + c * 256, dst + c * 256);
                }
                const bool adv = (ci[q] >= 0) && (ci[q] < W - 1);
                cptr[q] += adv ? cstride : 0;
                ci[q]++;
            }
            if constexpr (W2) {
                dma4<0>(wptr, Wring + slot * 64);
                const bool wadv = (wi >= 0) && (wi < W - 1);
                wptr += wadv ? istep : 0;
                wi++;
            }
            if (wl == 0) {
                const int h = ht < 0 ? 0 : (ht <= Hmax ? ht : Hmax);
                if (!TAGS && from_global && !dead && !(xflags & 4) && known < (unsigned)h + 1u) {
                    // slow path: the producer band is not far enough ahead.  Poll the word through
                    // LDS-DMA as well (no VGPR load, so nothing makes the compiler drain us elsewhere).
                    // Wait for a LEAD beyond the bare need: the producer publishes one pixel per step, so
                    // returning at the first sufficient value would put us back here on the next step.
                    unsigned spins = 0;
                    const unsigned long long t0 = dbg ? wall_clock64() : 0;
                    constexpr unsigned LEAD = MGM_P2_LEAD;
                    const unsigned want = (unsigned)h + 1u + LEAD < (unsigned)LL ? (unsigned)h + 1u + LEAD : (unsigned)LL;
                    n_slow++;
                    for (;;) {
                        n_spin++;
                        if (lane == 0) dma4<AUX_SC1>(prog_in, Hprog + slot);
                        wait_vmcnt<0>();
                        known = __builtin_amdgcn_readfirstlane(lds_read_u32_opaque(Hprog + slot));
                        if (known >= want) break;
                        __builtin_amdgcn_s_sleep(8);
                        if (((++spins) & 255u) == 0) {
                            if (lane == 0) dma4<AUX_SC1>(P.err, Hprog + slot);
                            wait_vmcnt<0>();
                            const unsigned e = __builtin_amdgcn_readfirstlane(lds_read_u32_opaque(Hprog + slot));
                            if (spins > (SPIN_LIMIT >> 2) || e != 0) {
                                if (lane == 0) __hip_atomic_store(P.err, 1u, RLX_AGENT);
                                dead = true;
                                break;
                            }
                        }
                    }
                    if (dbg) t_slow += wall_clock64() - t0;
                }
#pragma unroll
                for (int q = 0; q < NS; q++)
#pragma unroll
                    for (int c = 0; c < IPS; c++)
                        if (c * 64 + lane < LPL * 16)
                            dma16<AUX_SC1>(hptr + q * LP + c * 256, Hring + (slot * NS + q) * LP + c * 256);
                if constexpr (!TAGS) {
                    if (lane == 0) dma4<AUX_SC1>(hmptr, Hm + slot);
                    if (lane == 0) dma4<AUX_SC1>(prog_in, Hprog + slot);
                }
                const bool adv = ht >= 0 && ht < Hmax;
                hptr += adv ? hstep : 0;
                hmptr += adv ? 1 : 0;
                ht++;
            }
        };
        auto retire = [&]() {  // all but the newest D-1 steps of DMA have landed
            if (wl == 0) wait_vmcnt<PL::nA *(D - 1)>();
            else wait_vmcnt<PL::nB *(D - 1)>();
        };

        // TAGS: the hand-off slab of step `t` (pixel t, or t-1 with slope 1) has landed in ring slot `vslot`; make sure
        // it is the predecessor band's, fetching it again until it is
        auto validate = [&](int t, int vslot) {
            if constexpr (TAGS) {
                const int h = SL == 2 ? t : t - 1;
                if (wl != 0 || !from_global || h < 0 || h > Hmax || dead || (xflags & 4)) return;
                unsigned spins = 0;
                const unsigned long long t0 = dbg ? wall_clock64() : 0;
                unsigned long long tl0 = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int q = 0; q < NS; q++)
#pragma unroll
                        for (int c = 0; c < IPS; c++)
                            if (c * 64 + lane < LPL * 16) {
                                const u32x4 v = lds_read_b128_opaque(Hring + vslot * NSLP + q * LP + c * 256 + lane * 4);
                                // all four sign bits must equal the expected tag
                                ok = ok && (tag_in ? ((v.x & v.y & v.z & v.w) >> 31) != 0u : ((v.x | v.y | v.z | v.w) >> 31) == 0u);
                            }
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                    if (spins == 0) n_slow++;
                    if constexpr (MGM_P2_TIMELINE == 2 && XCDQ)
                        if (spins == 0 && P.tl_on) tl0 = wall_clock64();
                    n_spin++;
                    __builtin_amdgcn_s_sleep(4);
#pragma unroll
                    for (int q = 0; q < NS; q++)
#pragma unroll
                        for (int c = 0; c < IPS; c++)
                            if (c * 64 + lane < LPL * 16)
                                dma16<AUX_SC1>(hand_in + (long long)(mirror ? LL - 1 - h : h) * NSLP + q * LP + c * 256 + lane * 4,
                                               Hring + vslot * NSLP + q * LP + c * 256);
                    wait_vmcnt<0>();
                    if (((++spins) & 255u) == 0) {
                        if (lane == 0) dma4<AUX_SC1>(P.err, Hprog);
                        wait_vmcnt<0>();
                        const unsigned e = __builtin_amdgcn_readfirstlane(lds_read_u32_opaque(Hprog));
                        if (spins > (SPIN_LIMIT >> 2) || e != 0) {
                            if (lane == 0) __hip_atomic_store(P.err, 1u, RLX_AGENT);
                            dead = true;
                            break;
                        }
                    }
                }
                if (dbg) t_slow += wall_clock64() - t0;
                if constexpr (MGM_P2_TIMELINE == 2 && XCDQ)
                    if (tl0) tl_wait += wall_clock64() - tl0;
            }
        };

        // Compact costs: 255 stands for +INF, and turning it back costs the compute waves two instructions per label.
        // Most slabs hold no such code (labels whose right pixel lies outside the image, padded label slots), so the
        // loader -- which in the FH kernels has issue slots to spare -- looks at every compact piece once it has landed and leaves one
        // flag per line: the compute wave then converts the bytes alone (mgm_device.h, c8_decode).
        auto flag_inf = [&](int vslot) {
            if constexpr (CFLAG) {
                if (!(!(C8 && PL::NL == 2) || wl == 1)) return;  // (the loader that fetches the compact pieces)
#pragma unroll
                for (int q = 0; q < NDMA; q++) {
                    const u32x4 v = lds_read_b128_opaque(Cring + (vslot * NDMA + q) * 256 + lane * 4);
                    // != 0 iff a byte (two-byte costs: an aligned halfword) of w is all ones, the +INF code
                    auto ff = [](unsigned w) { return C8 == 2 ? (((~w) - 0x00010001u) & w & 0x80008000u) : (((~w) - 0x01010101u) & w & 0x80808080u); };
                    const unsigned long long bal = __builtin_amdgcn_ballot_w64((ff(v.x) | ff(v.y) | ff(v.z) | ff(v.w)) != 0u);
                    const int line = q * LPD + lane;
                    if (lane < LPD && line < NC)
                        Cflag[vslot * 16 + line] = ((bal >> ((lane * LPS) & 63)) & (LPS >= 64 ? ~0ull : ((1ull << (LPS & 63)) - 1ull))) != 0ull ? 1u : 0u;
                }
            }
        };

        int slot = 0;
        for (int t = 0; t < D; t++) {  // prologue: steps 0..D-1
            issue(slot);
            slot = slot + 1 == RD ? 0 : slot + 1;
        }
        retire();
        validate(0, 0);
        flag_inf(0);
        if (dbg && wl == 0 && lane == 0) {
            dbg[1] = wall_clock64();
            dbg[5] = t_slow;
        }
        step_barrier((xflags & 8) != 0);  // B0
        int uslot = 0;   // slot of the step that is about to run
        for (int s = 0; s < nsteps; s++) {
            if (!TAGS && wl == 0 && from_global) {
                // freshest progress word that has landed: the one issued for step s
                const unsigned k = __builtin_amdgcn_readfirstlane(lds_read_u32_opaque(Hprog + uslot));
                known = k > known ? k : known;
            }
            const unsigned long long ta = dbg ? wall_clock64() : 0;
            issue(slot);
            slot = slot + 1 == RD ? 0 : slot + 1;
            uslot = uslot + 1 == RD ? 0 : uslot + 1;
            const unsigned long long tb = dbg ? wall_clock64() : 0;
            retire();
            validate(s + 1, uslot);  // what the compute waves read after this barrier
            flag_inf(uslot);
            const unsigned long long tc = dbg ? wall_clock64() : 0;
            step_barrier((xflags & 8) != 0);
            if (dbg) {
                const unsigned long long td = wall_clock64();
                t_iss += tb - ta;
                t_ret += tc - tb;
                t_bar += td - tc;
            }
        }
        if constexpr (MGM_P2_TIMELINE != 0 && XCDQ)
            if (P.tl_on && wl == 0 && lane == 0) {
                tl_store(P, (long long)ticket * 8 + 2, MGM_P2_TIMELINE == 2 ? tl_wait : 0ull);  // (2: clock reads in the poll loop -- this compiler rejects them)
                tl_store(P, (long long)ticket * 8 + 6, n_spin);
                tl_store(P, (long long)ticket * 8 + 3, n_slow);
                tl_store(P, (long long)ticket * 8 + 5, (unsigned long long)nsteps);
            }
        if (dbg && wl == 0 && lane == 0) {
            dbg[2] = wall_clock64();
            dbg[6] = t_slow;
            dbg[7] = (unsigned long long)nsteps;
            dbg[8] = t_iss;
            dbg[9] = t_ret;
            dbg[10] = t_bar;
        }
        return;
    }

    // ============================= compute waves =============================
    const int r = wave;
    const int j = band * NC + r;
    const bool line_ok = j < NLn;
    const bool has_prev = line_ok && (j >= 1);
    const bool to_lds = (r < NC - 1) && (j + 1 < NLn);
    const bool to_global = (r == NC - 1) && (band + 1 < g.nbands);
    float *__restrict__ Lrb = P.vol[vgrp + (SUBV > 1 ? lane / LANES : 0)].Lr + (long long)(pass - P.pass0) * P.nvol;
    const long long pix0 = gbase + (long long)j * g.jstep;
    const int Wr = strips ? W - r : LL;           // this line is walked over [0, Wr)
    const int Wx = Wr + 1 < LL ? Wr + 1 : LL;     // ... and reads the slabs [0, Wx) of the line before it
    const float *fwd_src0 = r > 0 ? Tring + (r - 1) * RT * NSLP + lane * LPL : Hring + lane * LPL;
    const float *fwd_m0 = r > 0 ? Tm + (r - 1) * RT : Hm;
    // fp32: own ring [RD][LP]; compact: byte (r%LPD)*LPS*16 + lane*LPL of piece r/LPD of the step's slot
    const float *c_src0 = C8 ? Cring + (r / LPD) * 256 : Cring + r * RD * LP + lane * LPL;
    const int c8_byte = (r % LPD) * LPS * 16 + lane * LPL * (C8 ? C8 : 1);
    float *t_dst0 = Tring + r * RT * NSLP + lane * LPL;

    // The whole line walk, specialised on the neighbour order of the pass (FORM).
    auto run = [&](auto formc, auto slopec) {
        constexpr int FORM = decltype(formc)::value;
        constexpr int SLOPE = decltype(slopec)::value;  // 1 only with FORM == 0 and MGM <= 3
        NbT wA = {}, wB = {}, wC = {}, nb_i = {};

        // one step: X receives the newest slab of the previous line.  Slope 2: that is the fwd
        // neighbour (i+1), Z = same, Y = back.  Slope 1: it is the same neighbour (i), Z = back.
        constexpr int NEWOFF = SLOPE == 2 ? 1 : 0;  // index of the slab fetched this step, relative to i
        unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
        unsigned long long fh_sweeps = 0, fh_n = 0, fh_rep = 0;
        unsigned fh_max = 0;
        const bool prof = dbg && r == NC / 2;
        // where this lane's part of the Lr slab of step s goes: a running pointer (one 64-bit add per step instead of
        // rebuilding pixel index * label stride from half a dozen scalars that would otherwise have to stay live -- the FH
        // kernels spill SGPRs, and every restore is a VALU slot)
        const int wb[4] = {g.wplane[0], g.wplane[1], g.wplane[2], g.wplane[3]};  // (W2) weight plane of neighbour k
        float *qs = Lrb + (pix0 + (long long)(-1 - SLOPE * r) * istep) * L + (lane % LANES) * LPL;
        const long long dq = istep * L;
        auto step = [&](int s, int cslot, NbT &X, const NbT &Y, const NbT &Z) {
            const int i = s - 1 - SLOPE * r;
            float *const q_here = qs;
            qs += dq;
            const unsigned long long c0 = prof ? clock64() : 0;
            if (has_prev && i + NEWOFF >= 0 && i + NEWOFF < Wx) {
                const int sl = r > 0 ? ((i + NEWOFF) & (RT - 1)) : cslot;
                const float *src = fwd_src0 + sl * NSLP;
#pragma unroll
                for (int q = 0; q < NS; q++)
#pragma unroll
                    for (int k = 0; k < LPL; k++) X.w[q][k] = src[q * LP + k];
                if constexpr (TAGS)
                    if (r == 0) {  // (wave-uniform) the slab came from the previous band: E >= +0, drop the hand-off tag
#pragma unroll
                        for (int q = 0; q < NS; q++)
#pragma unroll
                            for (int k = 0; k < LPL; k++) X.w[q][k] = __builtin_fabsf(X.w[q][k]);
                    }
                if constexpr (!pubE) X.m = fwd_m0[sl];
            }
            if (line_ok && i >= 0 && i < Wr) {
                long long pix = 0;
                if constexpr (WEIGHTED || MGM_P2_DEV) pix = pix0 + (long long)i * istep;
                float Cv[LPL], Lv[LPL];
                if constexpr (C8 == 2) {
                    // two bytes per cost: convert the halfwords; the +INF code (all ones) is patched in only where the slab
                    // holds one -- flagged by the loader (FH kernels) or found by the wave itself
                    const unsigned char *src = reinterpret_cast<const unsigned char *>(c_src0 + cslot * NDMA * 256) + c8_byte;
                    unsigned hv[LPL];
                    if constexpr (LPL % 2 == 0) {
#pragma unroll
                        for (int h = 0; h < LPL / 2; h++) {
                            const unsigned w = reinterpret_cast<const unsigned *>(src)[h];
                            hv[2 * h] = w & 65535u;
                            hv[2 * h + 1] = w >> 16;
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < LPL; k++) hv[k] = reinterpret_cast<const unsigned short *>(src)[k];
                    }
#pragma unroll
                    for (int k = 0; k < LPL; k++) Cv[k] = (float)hv[k];
                    bool patch;
                    if constexpr (CFLAG) patch = __builtin_amdgcn_readfirstlane((int)Cflag[cslot * 16 + r]) != 0;
                    else {
                        bool any = false;
#pragma unroll
                        for (int k = 0; k < LPL; k++) any |= hv[k] == 65535u;
                        patch = __builtin_amdgcn_ballot_w64(any) != 0ull;
                    }
                    if (patch) {
#pragma unroll
                        for (int k = 0; k < LPL; k++) Cv[k] = hv[k] == 65535u ? f_inf() : Cv[k];
                    }
                } else if constexpr (C8 == 1) {
                    const unsigned char *src = reinterpret_cast<const unsigned char *>(c_src0 + cslot * NDMA * 256) + c8_byte;
                    // (wave-uniform) no +INF code in this slab: the bytes are the costs
                    const bool plain = CFLAG && __builtin_amdgcn_readfirstlane((int)Cflag[cslot * 16 + r]) == 0;
                    if constexpr (CFLAG && LPL == 4) {  // convert the bytes, then patch the +INF codes in where the slab has any
                        const unsigned w = reinterpret_cast<const unsigned *>(src)[0];
#pragma unroll
                        for (int k = 0; k < 4; k++) Cv[k] = (float)((w >> (8 * k)) & 255u);
                        if (!plain) {
#pragma unroll
                            for (int k = 0; k < 4; k++) Cv[k] = ((w >> (8 * k)) & 255u) == 255u ? f_inf() : Cv[k];
                        }
                    } else if (plain) {
                        if constexpr (LPL == 1) {
                            Cv[0] = (float)*src;
                        } else if constexpr (LPL == 2) {
                            const unsigned w = *reinterpret_cast<const unsigned short *>(src);
                            Cv[0] = (float)(w & 255u);
                            Cv[1] = (float)(w >> 8);
                        } else if constexpr (LPL % 4 == 0) {
#pragma unroll
                            for (int h = 0; h < LPL / 4; h++) {
                                const unsigned w = reinterpret_cast<const unsigned *>(src)[h];
#pragma unroll
                                for (int k = 0; k < 4; k++) Cv[h * 4 + k] = (float)((w >> (8 * k)) & 255u);
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < LPL; k++) Cv[k] = (float)src[k];
                        }
                    } else if constexpr (LPL == 1) {
                        Cv[0] = c8_decode(*src);
                    } else if constexpr (LPL == 2) {
                        const unsigned w = *reinterpret_cast<const unsigned short *>(src);
                        Cv[0] = c8_decode(w & 255u);
                        Cv[1] = c8_decode(w >> 8);
                    } else if constexpr (LPL % 4 == 0) {
                        // (kernels whose loader leaves no flags) convert the bytes; the +INF code is patched in only if
                        // some lane of the wave holds one -- 4 slots for the test instead of 2 per label
                        unsigned w[LPL / 4], any = 0u;
#pragma unroll
                        for (int h = 0; h < LPL / 4; h++) {
                            w[h] = reinterpret_cast<const unsigned *>(src)[h];
                            any |= (0xFEFEFEFEu - w[h]) & w[h] & 0x80808080u;  // != 0 iff a byte of w is 0xFF
#pragma unroll
                            for (int k = 0; k < 4; k++) Cv[h * 4 + k] = (float)((w[h] >> (8 * k)) & 255u);
                        }
                        if (__builtin_amdgcn_ballot_w64(any != 0u) != 0ull) {
#pragma unroll
                            for (int k = 0; k < LPL; k++) Cv[k] = ((w[k / 4] >> (8 * (k % 4))) & 255u) == 255u ? f_inf() : Cv[k];
                        }
                    } else {  // 3 or 6 labels per lane: the lane's bytes are not word-aligned
#pragma unroll
                        for (int k = 0; k < LPL; k++) Cv[k] = c8_decode(src[k]);
                    }
                } else {
                    const float *src = c_src0 + cslot * LP;
#pragma unroll
                    for (int k = 0; k < LPL; k++) Cv[k] = src[k];
                }
                const bool interior = has_prev && i >= 1 && i <= LL - 2;  // mgm_core.cc:538-541
                if (prof) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    asm volatile("" : "+v"(Cv[0]), "+v"(X.w[0][0]));
                    ph[0] += clock64() - c0;  // LDS reads landed
                }
                const unsigned long long c1 = prof ? clock64() : 0;
                // neighbours of the previous line by role
                const NbT &nb_same = SLOPE == 2 ? Z : X;
                const NbT &nb_back = SLOPE == 2 ? Y : Z;
                const NbT &nb_fwd = X;  // only read with SLOPE == 2
                if (interior) {
                    if constexpr (W2) {
                        // the selector word of THIS pixel: bit p = "the weight of plane p is not 1" (k_wsel)
                        const unsigned wm = (unsigned)__builtin_amdgcn_readfirstlane((int)Wring[cslot * 64 + r]);
                        const bool a0 = (wm >> wb[0]) & 1u, a1 = (wm >> wb[1]) & 1u, a2 = (wm >> wb[2]) & 1u, a3 = (wm >> wb[3]) & 1u;
                        if constexpr (FORM == 0) combine_w2<LPL, MGM>(Cv, nb_i, nb_same, nb_back, nb_fwd, a0, a1, a2, a3, Lv);
                        else combine_w2<LPL, MGM>(Cv, nb_fwd, nb_back, nb_same, nb_i, a0, a1, a2, a3, Lv);
                    } else if constexpr (!WEIGHTED) {
                        if constexpr (pubE) {
                            if constexpr (FORM == 0)
                                combine_unit_E<LPL, MGM, FH>(Cv, nb_i.w[0], nb_same.w[0], nb_back.w[0], nb_fwd.w[0], Lv);
                            else
                                combine_unit_E<LPL, MGM, FH>(Cv, nb_fwd.w[0], nb_back.w[0], nb_same.w[0], nb_i.w[0], Lv);
                        } else {
                            if constexpr (FORM == 0) combine_unit<LPL>(Cv, nb_i, nb_same, nb_back, nb_fwd, MGM, FH, Lv);
                            else combine_unit<LPL>(Cv, nb_fwd, nb_back, nb_same, nb_i, MGM, FH, Lv);
                        }
                    } else {
                        float Dw[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) Dw[k] = V.w8[(long long)g.wplane[k] * P.npix + pix];
                        int rl = 0, rh = 0x7fffffff;  // the pixel's own label range (ragged volumes, FH only)
                        if (FH && V.rlo) {
                            rl = (int)V.rlo[pix] - P.dmin;
                            rh = (int)V.rhi[pix] - P.dmin;
                        }
                        if constexpr (!FH) {
                            if constexpr (FORM == 0)
                                combine_whirsch<LPL>(Cv, nb_i, nb_same, nb_back, nb_fwd, Dw, P1, P2, MGM, Lv);
                            else
                                combine_whirsch<LPL>(Cv, nb_fwd, nb_back, nb_same, nb_i, Dw, P1, P2, MGM, Lv);
                        } else {
                            if (MGM == 2 && P.fh2_ragged) {  // (all-ones weights: the unweighted TSGM = 2 function of the reference)
                                if constexpr (FORM == 0) combine_fh2_ragged<LPL>(Cv, nb_i, nb_same, P1, P2, lane, P.Lreal, rl, rh, Lv);
                                else combine_fh2_ragged<LPL>(Cv, nb_fwd, nb_back, P1, P2, lane, P.Lreal, rl, rh, Lv);
                            } else if constexpr (FORM == 0)
                                combine_wfh<LPL>(Cv, nb_i, nb_same, nb_back, nb_fwd, Dw, P1, P2, MGM, lane, P.Lreal, Lv, rl, rh);
                            else
                                combine_wfh<LPL>(Cv, nb_fwd, nb_back, nb_same, nb_i, Dw, P1, P2, MGM, lane, P.Lreal, Lv, rl, rh);
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < LPL; k++) Lv[k] = Cv[k];
                }
                if (prof) {
                    asm volatile("" : "+v"(Lv[0]), "+v"(Lv[LPL - 1]));
                    ph[1] += clock64() - c1;  // combine
                }
                const unsigned long long c2 = prof ? clock64() : 0;
                if (!(xflags & 1)) {
#ifdef MGM_P2_XFLAG16  // xflags & 16 (timing experiment, wrong results): every Lr store lands in a small cache-resident window
                    float *q = Lrb + ((xflags & 16) ? (pix & 255) : pix) * L + (lane % LANES) * LPL;
#else
                    float *q = q_here;
#endif
#pragma unroll
                    for (int k = 0; k < LPL; k++) q[k] = Lv[k];
                }
                float m;
                if constexpr (SUBV == 1) {
                    m = slab_min<LPL>(Lv);
                } else {  // one minimum per lane group
                    m = Lv[0];
#pragma unroll
                    for (int k = 1; k < LPL; k++) m = fminf(m, Lv[k]);
                    m = dpp_min_row_shr1(m, m);
                    m = dpp_min_row_shr2(m, m);
                    m = dpp_min_row_shr4(m, m);
                    m = dpp_min_row_shr8(m, m);                      // lane 15 of every row: the row's minimum
                    if constexpr (SUBV == 2) m = dpp_min_bcast15(m, m);  // rows 1 and 3 take in rows 0 and 2
                    m = __shfl(m, lane | (LANES - 1));              // the group's last lane has it
                }
                nb_i.m = m;
                if (prof) {
                    float mm = m;
                    asm volatile("" : "+v"(mm));
                    ph[2] += clock64() - c2;  // store issue + wave min
                }
                const unsigned long long c3 = prof ? clock64() : 0;
                if constexpr (W2) {
                    const float P1a = V.p1a, P2a = V.p2a;  // P1 * a, P2 * a: rounded once, as the reference's products are
                    if constexpr (!FH) {
                        float N[LPL];
                        neighbour_min<LPL>(Lv, N);
                        const float cap = m + P2, capa = m + P2a;
#pragma unroll
                        for (int k = 0; k < LPL; k++) {
                            nb_i.w[0][k] = fminf(fminf(Lv[k], N[k] + P1), cap) - m;
                            nb_i.w[1][k] = fminf(fminf(Lv[k], N[k] + P1a), capa) - m;
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < LPL; k++) nb_i.w[0][k] = nb_i.w[1][k] = Lv[k];
                        unsigned sw = 0;
                        // (round 5) only the transforms some reader of this pixel picks in this pass: bits 8 + 2*pass (E_1) and
                        // 9 + 2*pass (E_a) of the pixel's word (k_wneed) -- wave-uniform; the slab nobody reads goes out as L - m
                        const unsigned need = ((unsigned)__builtin_amdgcn_readfirstlane((int)Wring[cslot * 64 + r]) >> (8 + 2 * pass)) & 3u;
                        if (need & 1u) fh_minconv<LPL, false, 1>(nb_i.w[0], m, P1, P2, lane, P.Lreal, sw);
                        if (need & 2u) fh_minconv<LPL, false, 1>(nb_i.w[1], m, P1a, P2a, lane, P.Lreal, sw);
#pragma unroll
                        for (int k = 0; k < LPL; k++) {
                            nb_i.w[0][k] -= m;
                            nb_i.w[1][k] -= m;
                        }
                    }
                } else if constexpr (!WEIGHTED) {
                    if constexpr (!FH) {
                        float N[LPL];
                        neighbour_min<LPL>(Lv, N, SUBV > 1 && lane % LANES == 0, SUBV > 1 && lane % LANES == LANES - 1);
                        const float cap = m + P2;
#pragma unroll
                        for (int k = 0; k < LPL; k++) nb_i.w[0][k] = fminf(fminf(Lv[k], N[k] + P1), cap);
                    } else {
#pragma unroll
                        for (int k = 0; k < LPL; k++) nb_i.w[0][k] = Lv[k];
                        unsigned sw = 0;
                        fh_minconv<LPL, false, SUBV>(nb_i.w[0], m, P1, P2, lane, P.Lreal, sw);
                        if (prof) { fh_sweeps += sw; fh_max = sw > fh_max ? sw : fh_max; fh_n++; fh_rep += sw > 2; }
                    }
                    if constexpr (pubE) {
#pragma unroll
                        for (int k = 0; k < LPL; k++) nb_i.w[0][k] = MGM == 2 ? (nb_i.w[0][k] - m) * 0.5f : nb_i.w[0][k] - m;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < LPL; k++) nb_i.w[0][k] = Lv[k];
                    if constexpr (!FH) neighbour_min<LPL>(Lv, nb_i.w[NS - 1]);
                }
                if (prof) {
                    asm volatile("" : "+v"(nb_i.w[0][0]), "+v"(nb_i.w[0][LPL - 1]));
                    ph[3] += clock64() - c3;  // publish transform
                }
                const unsigned long long c4 = prof ? clock64() : 0;
                if (to_lds) {
                    float *dst = t_dst0 + (i & (RT - 1)) * NSLP;
#pragma unroll
                    for (int q = 0; q < NS; q++)
#pragma unroll
                        for (int k = 0; k < LPL; k++) dst[q * LP + k] = nb_i.w[q][k];
                    if constexpr (!pubE)
                        if (lane == 0) Tm[r * RT + (i & (RT - 1))] = m;
                }
                if (prof) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    ph[4] += clock64() - c4;  // LDS write retired
                }
                if constexpr (TAGS) {
                    if (to_global) {
#pragma unroll
                        for (int q = 0; q < NS; q++) {
                            float tagged[LPL];
#pragma unroll
                            for (int k = 0; k < LPL; k++)
                                tagged[k] = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, nb_i.w[q][k]) & 0x7fffffffu) | tag_out);  // (a NaN -- INF costs with P2 = INF -- may carry a sign of its own)
                            float *hs = hand_out + ((long long)(mirror ? LL - 1 - i : i) * NS + q) * LP;
                            if constexpr (XCDQ) {
                                if (plain_out) store_slab_plain<LPL>(hs, lane, tagged);
                                else store_slab_sc1_wide<LPL>(hs, lane, tagged);
                            } else
                                store_slab_sc1_wide<LPL>(hs, lane, tagged);
                        }
                    }
                } else if (to_global) {
#pragma unroll
                    for (int q = 0; q < NS; q++)
                        store_slab_sc1_wide<LPL>(hand_out + ((long long)i * NS + q) * LP, lane, nb_i.w[q]);
                    if constexpr (!pubE)
                        if (lane == 0) st_sc1_x1(handm_out + i, m);
                    // Publish progress PUBLAG steps late, every PUBEVERY pixels: this wave only issues
                    // stores, and they retire in order, so once at most PUBLAG*SPS newer stores are
                    // outstanding every store of step i-PUBLAG (hand-off slab, minimum) has reached
                    // memory.  SPS is a lower bound of the stores issued per step (the Lr slab + the
                    // hand-off slabs), which only makes the wait conservative.  The word is not written
                    // every step: write-through stores to one address serialise at ~1.5 us each.
                    constexpr int SPS = 1 + NS * sc1_store_count<LPL>();
                    constexpr int PUBLAG = (63 / SPS) < MGM_P2_PUBLAG ? (63 / SPS) : MGM_P2_PUBLAG;
                    constexpr int PUBEVERY = MGM_P2_PUBEVERY;
                    if (i == LL - 1) {
                        wait_vmcnt<0>();
                        if (lane == 0) __hip_atomic_store(prog_out, (unsigned)LL, RLX_AGENT);
                    } else if (((i - PUBLAG + 1) % PUBEVERY) == 0 && i - PUBLAG >= 0) {
                        wait_vmcnt<PUBLAG * SPS>();
                        if (lane == 0) __hip_atomic_store(prog_out, (unsigned)(i - PUBLAG + 1), RLX_AGENT);
                    }
                }
            }
        };

        int cslot = 0;
        unsigned long long t_cbar = 0;
        step_barrier((xflags & 8) != 0);  // B0: the loaders' prologue has landed
        for (int s = 0; s < nsteps; s += 3) {
            step(s, cslot, wA, wB, wC);
            cslot = cslot + 1 == RD ? 0 : cslot + 1;
            const unsigned long long t0 = (dbg && r == NC / 2) ? wall_clock64() : 0;
            step_barrier((xflags & 8) != 0);
            if (dbg && r == NC / 2) t_cbar += wall_clock64() - t0;
            step(s + 1, cslot, wB, wC, wA);
            cslot = cslot + 1 == RD ? 0 : cslot + 1;
            step_barrier((xflags & 8) != 0);
            step(s + 2, cslot, wC, wA, wB);
            cslot = cslot + 1 == RD ? 0 : cslot + 1;
            step_barrier((xflags & 8) != 0);
        }
        if (dbg && r == NC / 2 && lane == 0) {
            dbg[14] = t_cbar * 3;
            dbg[3] = ph[0]; dbg[4] = ph[1]; dbg[11] = ph[2]; dbg[12] = ph[3]; dbg[13] = ph[4];
            dbg[15] = ((fh_sweeps & 0xfffff) << 44) | ((unsigned long long)(fh_max & 0xff) << 36) | ((fh_n & 0x3ffff) << 18) | (fh_rep & 0x3ffff);
        }
    };
    if (form != 0) run(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
    else if (SL == 2) run(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
    else if constexpr (MGM <= 3) run(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
}

// The kernel: one work item per workgroup, taken by ticket -- or, XCDQ, the work items of the XCD the workgroup finds
// itself on.  XCDQ (launches in which the chains of bands matter; mgm_api.hip, run_passes): the host deals the bands of
// every pass in blocks of consecutive bands -- or whole passes -- to eight queues, one per XCD; a workgroup reads its XCC id at run time and
// works through THAT queue, one item after the other, until it is empty.  A band whose successor sits in the same queue
// hands its slabs over with PLAIN stores: they stay in the XCD's L2, where the successor's L2-served (sc1) loads find
// them after an L2 round trip instead of a trip through the fabric (write-through stores drop the line from the L2 --
// MI355X_MICROARCH, "stores of each flavour") -- the hand-off lag of a band shrinks, and with it the chain of a pass;
// and a band that follows another on a CU starts without a workgroup having to be dispatched first, which measured
// as the larger half of the gain.
// The last band of a block hands over write-through as before.  Nothing depends on WHERE a workgroup runs except
// through the id it reads itself; progress needs one resident workgroup per XCD (every queue is in global ticket
// order, so the earliest unfinished item of the launch is always at the head of its queue with all it needs finished),
// and the last workgroup to leave checks that every queue was worked off -- a queue without workgroups raises the
// watchdog word instead of leaving lines unwritten.
// ONEB: the build for launches that run ONE band per CU (chain-bound: mgm_api.hip, wg_per_cu == 1).  The compact
// unweighted kernels are otherwise capped at 64 VGPRs so that two bands fit a CU, and under that cap every FH instance
// with the queue loop spilled 9-14 VGPRs (40-52 bytes of scratch per lane) and ~50 SGPRs -- scratch traffic and
// v_readlane restores on the critical chain of exactly the launches that are bound by the length of a step.
template <int LPL, bool FH, bool WEIGHTED, int MGM, int C8, int SUBV = 1, bool DEEP = false, bool XCDQ = false, bool ONEB = false, bool W2 = false>
__global__ void __launch_bounds__((Plan<LPL, 1, true, C8>::NC + Plan<LPL, 1, true, C8>::NL) * 64,
                                  (LPL >= 12 ? 2 : ONEB ? MGM_P2_ONEB_WPE : (C8 && LPL <= 4 && !WEIGHTED) ? MGM_P2_WAVES_PER_EU : 4)) k_pass2(const PassParams P)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int *s_ticket = reinterpret_cast<int *>(smem + P2Lds<LPL, FH, WEIGHTED, MGM, C8, DEEP, W2>::task_off);
    if constexpr (!XCDQ) {
        if (threadIdx.x == 0) *s_ticket = (int)atomicAdd(P.ticket, 1u);
        __syncthreads();
        pass2_item<LPL, FH, WEIGHTED, MGM, C8, SUBV, DEEP, false, W2>(P, *s_ticket);
    } else {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        const int xq = P.xcdq == 2 ? 0 : (int)(xcc & 7u);  // (2: one queue for all -- the persistent loop alone, an A/B setting)
        const int2 qi = P.tasks[xq - 8];  // (first ticket, count) of this XCD's queue: the table's header
        for (;;) {
            if (threadIdx.x == 0) *s_ticket = (int)atomicAdd(P.qticket + xq, 1u);
            __syncthreads();
            const int t = *s_ticket;
            if (t >= qi.y) break;
            if constexpr (MGM_P2_TIMELINE != 0)
                if (P.tl_on && threadIdx.x == 0) {
                    unsigned hwid;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                    tl_store(P, (long long)(qi.x + t) * 8 + 0, wall_clock64());
                    tl_store(P, (long long)(qi.x + t) * 8 + 4, ((unsigned long long)xcc << 32) | hwid);
                }
            pass2_item<LPL, FH, WEIGHTED, MGM, C8, SUBV, DEEP, true, W2>(P, qi.x + t);
            wait_vmcnt<0>();   // (the loader's DMAs beyond the last step)
            __syncthreads();   // LDS and s_ticket are free again
            if constexpr (MGM_P2_TIMELINE != 0)
                if (P.tl_on && threadIdx.x == 0) tl_store(P, (long long)(qi.x + t) * 8 + 1, wall_clock64());
        }
        if (threadIdx.x == 0) {
            const unsigned left = atomicAdd(P.qticket + 8, 1u) + 1u;
            if (left == gridDim.x) {
                bool all = true;
                for (int q = 0; q < 8; q++) all = all && __hip_atomic_load(P.qticket + q, RLX_AGENT) >= (unsigned)P.tasks[q - 8].y;
                if (!all) __hip_atomic_store(P.err, 1u, RLX_AGENT);
            }
        }
    }
}

// ---- launcher (one translation unit per LPL: -DMGM_P2_LPL=n) -----------------------
template <int LPL, bool FH, bool WEIGHTED, int MGM, int C8, int SUBV = 1, bool DEEP = false, bool XCDQ = false, bool ONEB = false, bool W2 = false>
static hipError_t launch2_c8(const PassParams &p, int ntasks, hipStream_t s)
{
    using PL = typename P2Lds<LPL, FH, WEIGHTED, MGM, C8, DEEP, W2>::PL;
    size_t shmem = sizeof(float) * (size_t)PL::lds_floats(PL::D);
    // Occupancy is chosen per launch through the LDS request: the compact unweighted kernels are built for two
    // workgroups per CU (<= 64 VGPRs, < 80 KB of LDS).  Two bands per CU hide each other's barrier and LDS stalls --
    // right when the launch is throughput-bound (a batch of volumes); a single volume is bound by the chain of
    // bands, where the doubled step latency costs more than it gives, so it asks for > half the LDS and runs alone.
    if (p.wg_per_cu < 2 && shmem < 81 * 1024) shmem = 81 * 1024;
    auto kern = k_pass2<LPL, FH, WEIGHTED, MGM, C8, SUBV, DEEP, XCDQ, ONEB, W2>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(ntasks), dim3((PL::NC + PL::NL) * 64), shmem, s, p);
    return hipGetLastError();
}

template <int LPL, bool FH, bool WEIGHTED, int MGM>
static hipError_t launch2_one(const PassParams &p, int ntasks, hipStream_t s)
{
    // (The kernels with per-XCD queues do not exist in development builds: with the timers compiled in, this compiler
    // fails on some of their instances -- "Illegal instruction detected: Operand has incorrect register class",
    // V_CMP_NE_U32 on src_shared_base.  The host does not ask for them there, mgm_api.hip.)
    if constexpr (LPL == 4 && !WEIGHTED && !(FH && MGM == 2)) {  // several volumes per wave (128 / 64 labels)
        if (p.subv > 1 && p.xcdq) return hipErrorInvalidValue;  // (no gain measured there: cfg2 x 4..16, cfg5 x 4..16 within noise)
        if (p.subv == 2 && p.vol[0].C8) {
            return p.deep ? launch2_c8<LPL, FH, WEIGHTED, MGM, true, 2, true>(p, ntasks, s) : launch2_c8<LPL, FH, WEIGHTED, MGM, true, 2>(p, ntasks, s);
        }
        if (p.subv == 4 && p.vol[0].C8) {
            return p.deep ? launch2_c8<LPL, FH, WEIGHTED, MGM, true, 4, true>(p, ntasks, s) : launch2_c8<LPL, FH, WEIGHTED, MGM, true, 4>(p, ntasks, s);
        }
    }
    if (p.subv > 1) return hipErrorInvalidValue;
    if constexpr (!WEIGHTED && !(FH && MGM == 2) && LPL <= 8)
        if (p.vol[0].C8 && p.cbytes == 2) {  // two bytes per cost (round 4): the kernels with deep rings only
            if (!p.deep) return hipErrorInvalidValue;
            if constexpr (!MGM_P2_DEV) {
                if (p.xcdq && p.oneb) return launch2_c8<LPL, FH, WEIGHTED, MGM, 2, 1, true, true, true>(p, ntasks, s);
                if (p.xcdq) return launch2_c8<LPL, FH, WEIGHTED, MGM, 2, 1, true, true, false>(p, ntasks, s);
            }
            if (p.xcdq) return hipErrorInvalidValue;
            return launch2_c8<LPL, FH, WEIGHTED, MGM, 2, 1, true>(p, ntasks, s);
        }
    if (p.vol[0].C8 && p.cbytes == 2) return hipErrorInvalidValue;
    if constexpr (!WEIGHTED && !(FH && MGM == 2))
        if (p.vol[0].C8 && p.deep) {
            if constexpr (!MGM_P2_DEV) {
                // (the Hirschmueller launches with queues run one band per CU unless told otherwise, the FH ones up to a
                // load/chain of 1.5: mgm_api.hip)
                if (p.xcdq && p.oneb) return launch2_c8<LPL, FH, WEIGHTED, MGM, true, 1, true, true, true>(p, ntasks, s);
                if (p.xcdq) return launch2_c8<LPL, FH, WEIGHTED, MGM, true, 1, true, true, false>(p, ntasks, s);
            }
            if (p.xcdq) return hipErrorInvalidValue;
            return launch2_c8<LPL, FH, WEIGHTED, MGM, true, 1, true>(p, ntasks, s);
        }
    if (p.vol[0].C8) return launch2_c8<LPL, FH, WEIGHTED, MGM, true>(p, ntasks, s);
    return launch2_c8<LPL, FH, WEIGHTED, MGM, false>(p, ntasks, s);
}

template <int LPL, bool FH, bool WEIGHTED>
static hipError_t launch2_mgm(const PassParams &p, int ntasks, hipStream_t s)
{
    switch (p.MGM) {
        case 1: return launch2_one<LPL, FH, WEIGHTED, 1>(p, ntasks, s);
        case 2: return launch2_one<LPL, FH, WEIGHTED, 2>(p, ntasks, s);
        case 3: return launch2_one<LPL, FH, WEIGHTED, 3>(p, ntasks, s);
        case 4: return launch2_one<LPL, FH, WEIGHTED, 4>(p, ntasks, s);
        default: return hipErrorInvalidValue;
    }
}

#ifndef MGM_P2_LPL
#error "compile with -DMGM_P2_LPL=<1|2|3|4|6|8|12|16>"
#endif
// two-valued weights (wmode 2): the compact kernels with deep rings and per-XCD queues, one band per CU
template <int LPL, bool FH>
static hipError_t launch2_w2(const PassParams &p, int ntasks, hipStream_t s)
{
    if constexpr (LPL <= 4 && !MGM_P2_DEV) {
        if (!p.vol[0].C8 || !p.deep || !p.xcdq || p.subv > 1 || !p.vol[0].wsel) return hipErrorInvalidValue;
        switch (p.MGM) {
            case 1: return launch2_c8<LPL, FH, false, 1, true, 1, true, true, true, true>(p, ntasks, s);
            case 2: return launch2_c8<LPL, FH, false, 2, true, 1, true, true, true, true>(p, ntasks, s);
            case 3: return launch2_c8<LPL, FH, false, 3, true, 1, true, true, true, true>(p, ntasks, s);
            case 4: return launch2_c8<LPL, FH, false, 4, true, 1, true, true, true, true>(p, ntasks, s);
            default: return hipErrorInvalidValue;
        }
    }
    return hipErrorInvalidValue;
}

template <>
hipError_t launch_pass2_lpl<MGM_P2_LPL>(const PassParams &p, int ntasks, bool fh, int wmode, hipStream_t s)
{
    constexpr int LPL = MGM_P2_LPL;
    if (wmode == 2) return fh ? launch2_w2<LPL, true>(p, ntasks, s) : launch2_w2<LPL, false>(p, ntasks, s);
    if (!wmode) return fh ? launch2_mgm<LPL, true, false>(p, ntasks, s) : launch2_mgm<LPL, false, false>(p, ntasks, s);
    if constexpr (LPL >= 12) return hipErrorInvalidValue;  // (768 / 1024 labels: the weighted kernels' rings do not fit the LDS; first build)
    else return fh ? launch2_mgm<LPL, true, true>(p, ntasks, s) : launch2_mgm<LPL, false, true>(p, ntasks, s);
}

}  // namespace mgm
