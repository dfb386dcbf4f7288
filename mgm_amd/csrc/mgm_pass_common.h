// mgm_pass_common.h -- device code shared by the two builds of K3 (mgm_pass.hip,
// mgm_pass2.hip): slab I/O, the per-slab transforms and the four reference
// update functions.  See mgm_pass.hip for the reference citations.
//
// Every translation unit including this is compiled with -fno-honor-nans: the
// Lr recursion is NaN-free (costs are finite or +INF and every pixel has a
// finite cost, mgm_costvolume.h:414-421).  Nothing here may rely on NaN
// semantics.
#pragma once
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
__device__ __forceinline__ void neighbour_min(const float (&Lv)[LPL], float (&N)[LPL], bool first_lane = false,
                                              bool last_lane = false)
{
    // first_lane / last_lane: this lane starts / ends a label range inside the wave (several volumes per wave)
    float left = dpp_shr1(Lv[LPL - 1], f_inf());
    float right = dpp_shl1(Lv[0], f_inf());
    left = first_lane ? f_inf() : left;
    right = last_lane ? f_inf() : right;
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
// Repair of a rejected carry guess: all 64 carries at once, exactly, when the additions inside a
// binade are exact (P1 a multiple of twice the float grid there, e.g. P1 = 2 and values < 2^23).
// Then the only roundings of a sequential chain z (+) P1 (+) P1 ... happen on the steps that land in
// a higher binade, each one RNE rounding of an exactly known sum to that binade's grid; and since
// the steps that follow add multiples of the grid, the roundings commute with them: the chain equals
// the EXACT sum z' + (n-1)*P1, z' = z (+) P1, rounded successively to the grid of every binade from
// z' upwards (a grid at or below that of z' changes nothing, so starting at P1's binade serves every
// origin).  Successive rounding is monotone, so the winning origin is the one with the smallest
// exact sum: a min-plus scan in f64, which holds these sums exactly.  If the premise does not hold
// the result is merely another guess; the caller's fixed-point sweeps remain the ground truth.
// GROUPS: label ranges per wave (several volumes per wave): lane groups of 64/GROUPS lanes scan independently.
// (The min-plus scan of the f64 sums runs on DPP moves -- two 32-bit moves per value -- with the same row / row-broadcast
// structure as the fp32 guess in fh_scan: the first version used LDS-crossbar shuffles, ~2000 cycles per repair, and a
// repair anywhere in a band stalls all its lock-stepped lines.  0.5-1 % of the slabs of real census data take this path.)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_mov_f64(double v, double fill)  // lanes without a source lane (or outside ROWMASK) get `fill`
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v), f = __builtin_bit_cast(unsigned long long, fill);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)f, (int)(unsigned)b, CTRL, ROWMASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(f >> 32), (int)(unsigned)(b >> 32), CTRL, ROWMASK, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int LPL, bool FWD, int GROUPS = 1>
__device__ __forceinline__ float fh_repair(float a, float P1, int lane_in)
{
    constexpr int GL = 64 / GROUPS;
    // (cold path: keep its lane masks and f64 constants from being hoisted into the caller's hot loop, where they
    // would push live scalar registers into spills)
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    asm volatile("" : "+v"(P1));
    const double P1d = (double)P1, Rd = (double)LPL * (double)P1;
    const double inf = (double)f_inf();
    const int li = lane & 15, row = lane >> 4, lg = lane % GL;
    auto dmin = [](double x, double y) { return y < x ? y : x; };
    const double b = (double)(a + P1);  // first step of every chain that starts at this lane's carry-out
    // s = origin' + LPL*P1 from the neighbour lane (the surplus P1 comes off below); nothing enters a label range from outside
    double s = FWD ? dpp_mov_f64<0x138, 0xf>(b, inf) : dpp_mov_f64<0x130, 0xf>(b, inf);  // wave_shr:1 / wave_shl:1
    s = (FWD ? lg == 0 : lg == GL - 1) ? inf : s + Rd;
    if constexpr (FWD) {
        s = dmin(s, dpp_mov_f64<0x111, 0xf>(s, inf) + Rd);         // row_shr:1
        s = dmin(s, dpp_mov_f64<0x112, 0xf>(s, inf) + 2.0 * Rd);   // row_shr:2
        s = dmin(s, dpp_mov_f64<0x114, 0xf>(s, inf) + 4.0 * Rd);   // row_shr:4
        s = dmin(s, dpp_mov_f64<0x118, 0xf>(s, inf) + 8.0 * Rd);   // row_shr:8
        if constexpr (GROUPS <= 2)  // rows 1 and 3 take in the row before them (lane 15 of it) ...
            s = dmin(s, dpp_mov_f64<0x142, 0xa>(s, inf) + (double)(li + 1) * Rd);
        if constexpr (GROUPS == 1)  // ... rows 2 and 3 everything up to lane 31
            s = dmin(s, dpp_mov_f64<0x143, 0xc>(s, inf) + (double)(li + 1 + (row == 3 ? 16 : 0)) * Rd);
    } else {
        s = dmin(s, dpp_mov_f64<0x101, 0xf>(s, inf) + Rd);         // row_shl:1
        s = dmin(s, dpp_mov_f64<0x102, 0xf>(s, inf) + 2.0 * Rd);
        s = dmin(s, dpp_mov_f64<0x104, 0xf>(s, inf) + 4.0 * Rd);
        s = dmin(s, dpp_mov_f64<0x108, 0xf>(s, inf) + 8.0 * Rd);
        if constexpr (GROUPS == 1) {  // the totals of the rows after this one sit at their first lanes
            const double t3 = readlane_f64(s, 48);
            const double t2 = dmin(readlane_f64(s, 32), t3 + 16.0 * Rd);
            const double t1 = dmin(readlane_f64(s, 16), t2 + 16.0 * Rd);
            const double tp = row == 0 ? t1 : (row == 1 ? t2 : t3);
            s = dmin(s, row <= 2 ? tp + (double)(16 - li) * Rd : inf);
        } else if constexpr (GROUPS == 2) {
            const double t1 = readlane_f64(s, 16), t3 = readlane_f64(s, 48);
            s = dmin(s, (row & 1) ? inf : (row == 0 ? t1 : t3) + (double)(16 - li) * Rd);
        }
    }
    s -= P1d;
    if (P1 > 0.0f) {
        double B = (double)__builtin_bit_cast(float, (__builtin_bit_cast(unsigned, P1) & 0x7f800000u) + 0x00800000u);
        for (int it = 0; it < 64; it++) {
            const bool reach = s < inf && s >= B;  // the chain lands in the binade [B, 2B)
            if (__builtin_amdgcn_ballot_w64(reach) == 0ull) break;
            const double magic = B * 805306368.0;  // 1.5 * 2^52 * (B * 2^-23): rounds to multiples of the binade's grid
            const double r = (s + magic) - magic;
            s = reach ? r : s;
            B += B;
        }
    }
    return fminf(a, (float)s);
}

// Second guess after a rejected one, still in fp32 (cold path, ~25 instructions): what makes the first guess fail is
// ONE addition that crosses three or more binades -- in practice a hop that starts from the near-zero value at the
// winning label -- because a single rounding of x + n*P1 then differs from the reference's rounding at every binade on
// the way (crossing one or two binades in one addition gives the same bits: the two roundings cannot disagree).  So:
// every lane first advances its carry-out by one whole lane with LPL exact unit steps (the value is then >= LPL*P1 = R),
// and the min-plus scan over those cuts every hop into pieces that cross at most two binades from there: 4R at a time
// from a value >= R, then 12R from >= 5R, then 16R.  Exact under fh_repair's premise (additions inside a binade exact);
// emulated on the CPU against the sequential recurrence on 3*10^6 slabs of census data for P1 in {0.75, 2, 2.5, 3, 8}:
// no miss (tools/fh_emul.py).  The caller verifies it like any guess; fh_repair remains behind it.
template <int LPL, bool FWD, int GROUPS = 1>
__device__ __forceinline__ float fh_careful(float a, float P1, int lane_in)
{
    constexpr int GL = 64 / GROUPS;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));  // (cold path: keep its per-lane constants out of the caller's hot loop)
    asm volatile("" : "+v"(P1));
    const float R = (float)LPL * P1, inf = f_inf();
    const int li = lane & 15, row = lane >> 4, lg = lane % GL;
    float b = a;
#pragma unroll
    for (int q = 0; q < LPL; q++) b = b + P1;  // the chain out of this lane, through the whole next lane
    float s = FWD ? dpp_shr1(b, inf) : dpp_shl1(b, inf);
    s = (FWD ? lg == 0 : lg == GL - 1) ? inf : s;
    auto cut2 = [&](float off, float &p1, float &p2) {  // off = p1 + p2, p1 <= 4R
        p1 = fminf(off, 4.0f * R);
        p2 = off - p1;
    };
    if constexpr (FWD) {
        s = dpp_min_row_shr1(s, s + R);
        s = dpp_min_row_shr2(s, s + 2.0f * R);
        s = dpp_min_row_shr4(s, s + 4.0f * R);
        s = dpp_min_row_shr8(s, (s + 4.0f * R) + 4.0f * R);
        if constexpr (GROUPS <= 2) {  // rows 1 and 3 take in lane 15 of the row before them, (li+1) lanes away
            float p1, p2;
            cut2((float)(li + 1) * R, p1, p2);
            const bool on = GROUPS == 1 ? row >= 1 : (row & 1);
            float t = dpp_add_bcast15(s, on ? p1 : inf);
            t = t + (on ? p2 : 0.0f);
            s = fminf(s, t);
        }
        if constexpr (GROUPS == 1) {  // rows 2 and 3 take in lane 31
            const float off = (float)(li + 1 + (row == 3 ? 16 : 0)) * R;
            const float p1 = fminf(off, 4.0f * R), p2 = fminf(off - p1, 12.0f * R), p3 = (off - p1) - p2;
            const bool on = row >= 2;
            float t = dpp_add_bcast31(s, on ? p1 : inf);
            t = (t + (on ? p2 : 0.0f)) + (on ? p3 : 0.0f);
            s = fminf(s, t);
        }
    } else {
        s = dpp_min_row_shl1(s, s + R);
        s = dpp_min_row_shl2(s, s + 2.0f * R);
        s = dpp_min_row_shl4(s, s + 4.0f * R);
        s = dpp_min_row_shl8(s, (s + 4.0f * R) + 4.0f * R);
        auto add16 = [&](float x) { return (x + 4.0f * R) + 12.0f * R; };  // sixteen lanes further
        float p1, p2;
        cut2((float)(16 - li) * R, p1, p2);
        if constexpr (GROUPS == 1) {
            const float t3 = readlane_f(s, 48);
            const float t2 = fminf(readlane_f(s, 32), add16(t3));
            const float t1 = fminf(readlane_f(s, 16), add16(t2));
            const float tp = row == 0 ? t1 : (row == 1 ? t2 : t3);
            s = fminf(s, row <= 2 ? (tp + p1) + p2 : inf);
        } else if constexpr (GROUPS == 2) {
            const float t1 = readlane_f(s, 16), t3 = readlane_f(s, 48);
            s = fminf(s, (row & 1) ? inf : ((row == 0 ? t1 : t3) + p1) + p2);
        }
    }
    return fminf(a, s);
}

// One direction of minConvTruncatedLinear.  FWD: M[o] = min(M[o-1] + P1, M[o]) for rising o.
// Written for the VALU issue budget (four waves share a SIMD: every slot costs 16 cycles): the
// guess is a min-plus scan of fused DPP instructions, 2 slots per log-step.
template <int LPL, bool FWD, int GROUPS = 1>
__device__ __forceinline__ void fh_scan(float (&M)[LPL], float P1, int lane, unsigned &sweeps)
{
    static_assert(GROUPS == 1 || GROUPS == 2 || GROUPS == 4, "lane groups of 64, 32 or 16");
    constexpr int GL = 64 / GROUPS;            // lanes per label range
    constexpr int K0 = FWD ? 0 : LPL - 1;      // first label of the lane in scan order
    constexpr int K1 = FWD ? LPL - 1 : 0;      // last
    constexpr int DK = FWD ? 1 : -1;
    const float rampP = (float)LPL * P1;

    // carry-out ignoring carry-in: exact for the first lane, a candidate origin elsewhere.  (The cold paths below compute
    // it again -- `again` keeps that from being merged with the hot path's value, whose register the scan can then
    // overwrite in place instead of copying it first.)
    auto origin = [&](bool again) {
        float a = M[K0];
        if (again) asm volatile("" : "+v"(a));
#pragma unroll
        for (int q = 1; q < LPL; q++) a = fminf(M[K0 + q * DK], a + P1);  // (compile-time indices: no scratch)
        return a;
    };
    const float a = origin(false);

    // cheap guess of the 64 carries: min-plus scan with single-rounded ramps.  Inside each row of
    // 16 lanes with DPP row shifts; across rows FWD with the row broadcasts, BWD through the row
    // totals read into SGPRs -- no LDS-crossbar permutes on this path.  The per-lane constants
    // depend on (lane, P1) only: with unit weights they are hoisted out of the line walk.
    const int li = lane & 15, row = lane >> 4;
    const float r1 = rampP, r2 = 2.0f * rampP, r4 = 4.0f * rampP, r8 = 8.0f * rampP, r16 = 16.0f * rampP;
    float c = a;
    if constexpr (FWD) {
        c = dpp_min_row_shr1(c, c + r1);
        c = dpp_min_row_shr2(c, c + r2);
        c = dpp_min_row_shr4(c, c + r4);
        c = dpp_min_row_shr8(c, c + r8);
        // lane 15 of the row before (rows 1..3), then lane 31 (rows 2..3), at their distance in lanes -- as far as those
        // rows belong to the same label range
        if constexpr (GROUPS <= 2) {
            const float offA = (GROUPS == 1 ? row >= 1 : (row & 1)) ? (float)(li + 1) * rampP : f_inf();
            c = fminf(c, dpp_add_bcast15(c, offA));
        }
        if constexpr (GROUPS == 1) {
            const float offB = row >= 2 ? (float)(li + 1 + (row == 3 ? 16 : 0)) * rampP : f_inf();
            c = fminf(c, dpp_add_bcast31(c, offB));
        }
    } else {
        c = dpp_min_row_shl1(c, c + r1);
        c = dpp_min_row_shl2(c, c + r2);
        c = dpp_min_row_shl4(c, c + r4);
        c = dpp_min_row_shl8(c, c + r8);
        if constexpr (GROUPS == 1) {
            // the mirror image of the forward scan's two row broadcasts: rows 0 and 2 take in the first lane of the row
            // after them, then rows 0 and 1 take in lane 32, which by then covers rows 2..3.  Per-lane offsets (+INF
            // where a hop does not apply); 5 VALU slots where chaining the three row totals first took 9.
            const float offA0 = row == 0 ? (float)(16 - li) * rampP : f_inf();
            const float offA2 = row == 2 ? (float)(16 - li) * rampP : f_inf();
            const float offB = row <= 1 ? (float)(32 - lane) * rampP : f_inf();
            c = fminf(fminf(c, readlane_f(c, 16) + offA0), readlane_f(c, 48) + offA2);
            c = fminf(c, readlane_f(c, 32) + offB);
        } else if constexpr (GROUPS == 2) {  // rows 0 and 2 take in the row after them (same label range)
            const float t1 = readlane_f(c, 16), t3 = readlane_f(c, 48);
            const float tp = row == 0 ? t1 : t3;
            const float offD = (row & 1) ? f_inf() : (float)(16 - li) * rampP;
            c = fminf(c, tp + offD);
        }
        (void)r16;
    }
    // no carry into the first lane of a label range
    const float p1edge = (FWD ? lane % GL == 0 : lane % GL == GL - 1) ? f_inf() : P1;
    float f[LPL];
    int boosted = 0;
    for (int it = 0; it < 70; it++) {
        // exact in-lane recurrence given the neighbour's carry
        const float cin = FWD ? dpp_add_wave_shr1(c, p1edge) : dpp_add_wave_shl1(c, p1edge);  // carry + P1
        f[K0] = fminf(M[K0], cin);
#pragma unroll
        for (int q = 1; q < LPL; q++) f[K0 + q * DK] = fminf(M[K0 + q * DK], f[K0 + (q - 1) * DK] + P1);
        const bool same = (f[K1] == c);
        c = f[K1];
        sweeps++;
        if (__builtin_amdgcn_ballot_w64(!same) == 0ull) break;
        if (boosted == 0) {
            // The guess was wrong somewhere: one of its hops crossed binades more than twice (a ramp from the near-zero
            // value at the winning label, or over a stretch of +INF costs).  Plain sweeps would repair one lane per
            // sweep.  Second guess: the same scan with carries advanced by a lane and cut hops, exact whenever the
            // additions inside a binade are; and if the next sweep rejects that too, every carry is re-derived in f64.
            boosted = 1;
#ifdef MGM_FH_NOREPAIR  // timing experiment (WRONG results): what do the repairs cost?
            break;
#endif
            c = fh_careful<LPL, FWD, GROUPS>(origin(true), P1, lane);
        } else if (boosted == 1) {
            boosted = 2;
            c = fh_repair<LPL, FWD, GROUPS>(origin(true), P1, lane);
        }
    }
#pragma unroll
    for (int k = 0; k < LPL; k++) M[k] = f[k];
}

// L: labels of a range (<= 64*LPL/GROUPS); the label slots beyond it are padding.  (FULL is a leftover tag: the padding
// check is made at run time.)  GROUPS: independent label ranges in the wave, each with its own minimum m (per lane).
template <int LPL, bool FULL = false, int GROUPS = 1>
__device__ __forceinline__ void fh_minconv(float (&M)[LPL], float m, float P1, float P2, int lane, int L, unsigned &sweeps)
{
    constexpr int GL = 64 / GROUPS;
    fh_scan<LPL, true, GROUPS>(M, P1, lane, sweeps);
    if (L < GL * LPL) {  // (wave-uniform) label slots >= L hold +INF on entry; the forward scan has filled them with ramp values
#pragma unroll
        for (int k = 0; k < LPL; k++) M[k] = ((lane % GL) * LPL + k < L) ? M[k] : f_inf();
    }
    fh_scan<LPL, false, GROUPS>(M, P1, lane, sweeps);
    if (P2 < f_inf()) {
        const float cap = m + P2;
#pragma unroll
        for (int k = 0; k < LPL; k++) M[k] = fminf(M[k], cap);
    }
}
// ---- NK min-convolutions SIDE BY SIDE (round 6; the range-proportional kernels' three-or-four neighbours of a pixel) ----------
// fh_scan is one dependent chain from its first instruction to its last (origin -> log-step guess -> sweep), ending in a
// wave-wide vote and a branch: a wave that runs three of them one after the other issues a VALU instruction every ~10 clocks
// (profiles/r05_rel_phases.txt: 3900 clocks per step for ~400 instructions), and the compiler cannot interleave them across
// their loops.  Here the NK guesses and the NK first sweeps are ONE straight-line block (independent chains the scheduler
// interleaves) with ONE vote; a rejected guess anywhere (rare) sends every array through fh_scan from scratch -- the fixed point
// of the sweeps is unique and equals the sequential recurrence whatever the starting carries, so the result is fh_scan's bit
// for bit.  (fh_scan itself is left untouched: the dense kernels' hot loops are built on it.)
template <int LPL, bool FWD, int GROUPS, int NK>
__device__ __forceinline__ void fh_scan_multi(float (&M)[NK][LPL], const float (&P1)[NK], int lane, unsigned &sweeps)
{
    static_assert(GROUPS == 4, "rows of 16 lanes: the only user is the range-proportional pass kernel");
    constexpr int GL = 64 / GROUPS;
    constexpr int K0 = FWD ? 0 : LPL - 1;
    constexpr int K1 = FWD ? LPL - 1 : 0;
    constexpr int DK = FWD ? 1 : -1;
    // every stage is written ACROSS the NK arrays (k innermost): NK independent chains per stage
    // carry-out of every lane ignoring its carry-in (fh_scan's `origin`)
    float c[NK], t[NK], rp[NK];
#pragma unroll
    for (int k = 0; k < NK; k++) c[k] = M[k][K0];
#pragma unroll
    for (int q = 1; q < LPL; q++)
#pragma unroll
        for (int k = 0; k < NK; k++) c[k] = fminf(M[k][K0 + q * DK], c[k] + P1[k]);
    // the guess: min-plus scan with single-rounded ramps inside each row of 16 lanes (fh_scan with GROUPS = 4)
#pragma unroll
    for (int k = 0; k < NK; k++) rp[k] = (float)LPL * P1[k];
#pragma unroll
    for (int st = 0; st < 4; st++) {
#pragma unroll
        for (int k = 0; k < NK; k++) t[k] = c[k] + (float)(1 << st) * rp[k];
        if constexpr (FWD) {
            if (st == 0) dpp_min_row_shr1_n<NK>(c, t);
            else if (st == 1) dpp_min_row_shr2_n<NK>(c, t);
            else if (st == 2) dpp_min_row_shr4_n<NK>(c, t);
            else dpp_min_row_shr8_n<NK>(c, t);
        } else {
            if (st == 0) dpp_min_row_shl1_n<NK>(c, t);
            else if (st == 1) dpp_min_row_shl2_n<NK>(c, t);
            else if (st == 2) dpp_min_row_shl4_n<NK>(c, t);
            else dpp_min_row_shl8_n<NK>(c, t);
        }
    }
    // one sweep: the exact in-lane recurrence given the neighbour lane's carry; the guess holds iff the carries reproduce
    const bool edge = FWD ? lane % GL == 0 : lane % GL == GL - 1;  // no carry into the first lane of a label range
    float pe[NK], cin[NK], f[NK][LPL];
#pragma unroll
    for (int k = 0; k < NK; k++) pe[k] = edge ? f_inf() : P1[k];
    if constexpr (FWD) dpp_add_wave_shr1_n<NK>(cin, c, pe);
    else dpp_add_wave_shl1_n<NK>(cin, c, pe);
#pragma unroll
    for (int k = 0; k < NK; k++) f[k][K0] = fminf(M[k][K0], cin[k]);
#pragma unroll
    for (int q = 1; q < LPL; q++)
#pragma unroll
        for (int k = 0; k < NK; k++) f[k][K0 + q * DK] = fminf(M[k][K0 + q * DK], f[k][K0 + (q - 1) * DK] + P1[k]);
    bool same = true;
#pragma unroll
    for (int k = 0; k < NK; k++) same = same && (f[k][K1] == c[k]);
    sweeps += NK;
    if (__builtin_amdgcn_ballot_w64(!same) == 0ull) {
#pragma unroll
        for (int k = 0; k < NK; k++)
#pragma unroll
            for (int q = 0; q < LPL; q++) M[k][q] = f[k][q];
    } else {  // (cold) some guess was rejected: every array settles by itself
#pragma unroll
        for (int k = 0; k < NK; k++) fh_scan<LPL, FWD, GROUPS>(M[k], P1[k], lane, sweeps);
    }
}
// minConvTruncatedLinear of NK arrays whose label slots all exist (64 slots per row of 16 lanes: no padding to restore between
// the two passes); m, P1, P2 per array
template <int LPL, int GROUPS, int NK>
__device__ __forceinline__ void fh_minconv_multi(float (&M)[NK][LPL], const float (&m)[NK], const float (&P1)[NK], const float (&P2)[NK], int lane,
                                                 unsigned &sweeps)
{
    fh_scan_multi<LPL, true, GROUPS, NK>(M, P1, lane, sweeps);
    fh_scan_multi<LPL, false, GROUPS, NK>(M, P1, lane, sweeps);
    // the cap, unconditionally: with P2 = +INF it is min(M, +INF) = M (no NaNs here: the caller's volumes have none and its P2 is finite)
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const float cap = m[k] + P2[k];
#pragma unroll
        for (int q = 0; q < LPL; q++) M[k][q] = fminf(M[k][q], cap);
    }
}
template <int LPL, bool FULL = false>
__device__ __forceinline__ void fh_minconv(float (&M)[LPL], float m, float P1, float P2, int lane, int L)
{
    unsigned sweeps = 0;
    fh_minconv<LPL, FULL>(M, m, P1, P2, lane, L, sweeps);
}

// ---- exact small-integer division -------------------------------------------------
// RN(x / 3) in three FMA-class operations.  c = RN(1/3); q0 = RN(x*c);
// r = x - 3*q0 (exact in one fma); q = RN(q0 + r*c).  Checked exhaustively
// against IEEE division for all 2^32 inputs (tests/test_div3.py on the CPU and
// mgm_selftest_div3 on the device): equal for every finite x except -0, and
// wrong only for -0 / +-INF, which v_div_fixup_f32 repairs.
__device__ __forceinline__ float div3_exact(float x)
{
    const float c = 0x1.555556p-2f;
    const float q0 = x * c;
    const float r = __builtin_fmaf(-3.0f, q0, x);
    const float q = __builtin_fmaf(r, c, q0);
    return __builtin_amdgcn_div_fixupf(q, 3.0f, x);
}
// e / howmany for howmany = 1..4, bit-identical to the fp32 division of the reference
template <int N>
__device__ __forceinline__ float div_small(float e)
{
    static_assert(N >= 1 && N <= 4, "MGM is 1..4");
    if constexpr (N == 1) return e;
    else if constexpr (N == 2) return e * 0.5f;
    else if constexpr (N == 3) return div3_exact(e);
    else return e * 0.25f;
}
__device__ __forceinline__ float div_small_rt(float e, int n)
{
    return n == 1 ? e : (n == 2 ? e * 0.5f : (n == 3 ? div3_exact(e) : e * 0.25f));
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
        out[k] = C[k] + div_small_rt(e, MGM);
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
        out[k] = C[k] + div_small_rt(e, MGM);
    }
}
// weighted FH (update_costW_trunclinear): w[0] = L; the min-convolution depends
// on the consumer's weights, so it runs here, once per neighbour.
template <int LPL, bool FULL = false>
__device__ __forceinline__ void combine_wfh(const float (&C)[LPL], const Nb<LPL, 1> &n1, const Nb<LPL, 1> &n2,
                                            const Nb<LPL, 1> &n3, const Nb<LPL, 1> &n4, const float (&D)[4], float P1,
                                            float P2, int MGM, int lane, int L, float (&out)[LPL], int rl = 0,
                                            int rh = 0x7fffffff)
{
    // [rl, rh]: the receiving pixel's own label range in a ragged volume.  The reference copies the neighbour's
    // values over THAT range (foreign labels read +INF) and convolves there (mgm_core.cc:242-271): masking the slab
    // to the range before the convolution is the same thing, since +INF + P1 never wins a minimum.
    auto take = [&](const Nb<LPL, 1> &n, float (&M)[LPL]) {
#pragma unroll
        for (int k = 0; k < LPL; k++) {
            const int o = lane * LPL + k;
            M[k] = (o >= rl && o <= rh) ? n.w[0][k] : f_inf();
        }
    };
    float e[LPL], M[LPL];
    take(n1, M);
    fh_minconv<LPL, FULL>(M, n1.m, P1 * D[0], P2 * D[0], lane, L);
#pragma unroll
    for (int k = 0; k < LPL; k++) e[k] = M[k] - n1.m;
    if (MGM >= 2) {
        take(n2, M);
        fh_minconv<LPL, FULL>(M, n2.m, P1 * D[1], P2 * D[1], lane, L);
#pragma unroll
        for (int k = 0; k < LPL; k++) e[k] += M[k] - n2.m;
    }
    if (MGM >= 3) {
        take(n3, M);
        fh_minconv<LPL, FULL>(M, n3.m, P1 * D[2], P2 * D[2], lane, L);
#pragma unroll
        for (int k = 0; k < LPL; k++) e[k] += M[k] - n3.m;
    }
    if (MGM >= 4) {
        take(n4, M);
        fh_minconv<LPL, FULL>(M, n4.m, P1 * D[3], P2 * D[3], lane, L);
#pragma unroll
        for (int k = 0; k < LPL; k++) e[k] += M[k] - n4.m;
    }
#pragma unroll
    for (int k = 0; k < LPL; k++) out[k] = C[k] + div_small_rt(e[k], MGM);
}


// update_cost2_trunclinear on a RAGGED volume (mgm_core.cc:197-219): the two neighbour slabs are copied over the
// receiving pixel's label range [rl, rh], FixBounrady_for_minConvTruncatedLinear (166-186) then folds what lies
// outside that range into its two end labels -- the forward recurrence over the neighbour's labels up to rl, the
// backward one down to rh, both on the neighbour's RAW values -- and the min-convolution runs over the range.
// In the dense layout a neighbour's missing labels hold +INF, so those two recurrences are plain scans of its
// whole slab read at labels rl and rh.
template <int LPL>
__device__ __forceinline__ void combine_fh2_ragged(const float (&C)[LPL], const Nb<LPL, 1> &n1, const Nb<LPL, 1> &n2,
                                                   float P1, float P2, int lane, int L, int rl, int rh, float (&out)[LPL])
{
    auto at = [&](const float (&v)[LPL], int o) {  // v's value at label o (wave-uniform o)
        float x = v[0];
#pragma unroll
        for (int k = 1; k < LPL; k++) x = (o % LPL == k) ? v[k] : x;
        return __shfl(x, o / LPL);
    };
    auto prepare = [&](const Nb<LPL, 1> &n, float (&M)[LPL]) {
        float F[LPL], B[LPL];
        unsigned sweeps = 0;
#pragma unroll
        for (int k = 0; k < LPL; k++) F[k] = B[k] = n.w[0][k];
        fh_scan<LPL, true>(F, P1, lane, sweeps);
        fh_scan<LPL, false>(B, P1, lane, sweeps);
        const float TL = at(F, rl), TR = at(B, rh);
#pragma unroll
        for (int k = 0; k < LPL; k++) {
            const int o = lane * LPL + k;
            float x = (o >= rl && o <= rh) ? n.w[0][k] : f_inf();
            if (o == rl) x = fminf(x, TL);
            if (o == rh) x = fminf(x, TR);
            M[k] = x;
        }
        fh_minconv<LPL>(M, n.m, P1, P2, lane, L);
    };
    float M1[LPL], M2[LPL];
    prepare(n1, M1);
    prepare(n2, M2);
#pragma unroll
    for (int k = 0; k < LPL; k++) out[k] = C[k] + (((M1[k] - n1.m) + M2[k]) - n2.m) * 0.5f;
}

}  // namespace mgm
