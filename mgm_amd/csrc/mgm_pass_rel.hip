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
// Structure: bands of 16 scan lines per workgroup in lock-step on the slope-2 diagonal (round 6: slope 1 where the fwd neighbour is not read;
// the form-1 passes with TSGM <= 3 ACROSS their lines, bands of 16 anti-diagonals all at the same step: see `diag` in k_pass_rel) -- 4 compute waves of FOUR lines each
// (a pixel's 64 slots on a row of 16 lanes, 4 per lane: see k_pass_rel) + a loader wave that feeds LDS rings by LDS-DMA --, a
// 4-deep LDS ring per line (a slab is read by the next line at three consecutive steps, each time at another shift), the band
// hand-off through global memory in self-validating slots (the launch's tag in every word's sign bit), work items by atomic ticket.  Not built here (the dense path keeps them): FH with TSGM = 2 without weights
// (update_cost2_trunclinear and its boundary fix-up), windows wider than 62 labels, costs that are not bytes, P2 = +INF, negative
// penalties or weights (the tags).  (Round 6: TSGM = 2 without weights, Hirschmueller -- update_cost2 -- runs here, PUBE with halved terms.)
#include <algorithm>

#include "mgm_pass_common.h"

namespace mgm {

#ifndef MGM_REL_PHASES
#define MGM_REL_PHASES 0  // 1 (development build, MGM_REL_DEFINES=-DMGM_REL_PHASES=1): per work item, the clocks every wave spent computing /
#endif                    // publishing / at the step barrier, and the loader issuing / waiting for its DMAs / at the barrier (with MGM_HIP_TIMELINE)
constexpr int NW = 4;        // compute waves of a workgroup (+ 1 loader wave)
constexpr int GL = 4;        // scan lines per wave: lane groups of 16 lanes, 4 label slots per lane
constexpr int RR = NW * GL;  // lines per band
constexpr int RD4 = 4;   // ring slots per line
constexpr int SD = 8;    // slots of the rings the loader wave fills (costs, records, weights, the previous band's slabs)

typedef __attribute__((address_space(3))) void *rel_lds_vptr;
typedef const __attribute__((address_space(1))) void *rel_glb_vptr;
// LDS-DMA: lane l moves 16 (4) bytes from its own source address to dst_base + 16 l (4 l): no VGPR round trip, so the
// compiler has nothing to wait for, and the loader retires its loads with COUNTED waits (mgm_pass2.hip does the same)
template <int AUX>
__device__ __forceinline__ void rel_dma16(const void *src_lane, void *dst_base)
{
    __builtin_amdgcn_global_load_lds((rel_glb_vptr)src_lane, (rel_lds_vptr)dst_base, 16, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ void rel_dma4(const void *src_lane, void *dst_base)
{
    __builtin_amdgcn_global_load_lds((rel_glb_vptr)src_lane, (rel_lds_vptr)dst_base, 4, 0, AUX);
}
constexpr int REL_SC1 = 16;  // agent-scope (L1-bypassing) cache policy bit
template <int N>
__device__ __forceinline__ void rel_wait_vmcnt()
{
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// LDS read the compiler must not order against pending LDS-DMA itself (it would drain vmcnt): the counted wait has made
// sure the word landed
__device__ __forceinline__ unsigned rel_lds_read_opaque(const unsigned *p)
{
    unsigned v;
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned *)p;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}
typedef unsigned rel_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rel_u4 rel_lds_read128_opaque(const float *p)
{
    rel_u4 v;
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float *)p;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}
__device__ __forceinline__ void rel_lds_write128_opaque(float *p, rel_u4 v)
{
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)p;
    asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(v) : "memory");
}
constexpr int REL_BIAS = 1 << 24;  // a hand-off slot holds base + REL_BIAS: a positive word, its sign bit free for the tag

// ---- the relative copy of a ragged volume ------------------------------------------------------------------------------
// Round 6: the layout comes in two widths and two cost sizes -- SLOTS = 64 or 128 label slots per pixel (windows of up to 62 /
// 126 labels: 4 or 8 slots per lane on the pixel's row of 16 lanes), costs as one byte (255 = +INF) or two (65535 = +INF:
// absolute differences of colour pairs, squared differences).  The host tries the narrowest form first and widens by what
// the flag word says (mgm_api.hip, rel_fill).
// one thread per (pixel, slot): rel[p][k] = code of C[p][base(p) + k - dmin] inside the pixel's window, all-ones elsewhere;
// flag |= 1 if a window is wider than SLOTS - 2 labels, |= 2 if a cost does not have the form
__global__ void __launch_bounds__(256) k_rel_gather(const float *__restrict__ C, const float *__restrict__ rlo, const float *__restrict__ rhi, long long npix,
                                                    int L, int dmin, int lgslots, int cb, uint8_t *__restrict__ rel8, int *__restrict__ relb, unsigned *flag)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long p = t >> lgslots;
    const int slots = 1 << lgslots;
    const int k = (int)(t & (slots - 1));
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
    if (hi - lo + 1 > slots - 2 || hi < lo) bad |= 1u;
    const int d = b + k;  // disparity of this slot
    if (cb == 4) {  // the cost itself (fp32: NCC, Birchfield-Tomasi, census over several words ...); a NaN cost has no place here (bit 1)
        float x = __builtin_huge_valf();
        if (d >= lo && d <= hi && d - dmin >= 0 && d - dmin < L) {
            x = C[p * L + (d - dmin)];
            if (x != x || x < 0.0f) {  // (NaN: the operand-order-faithful kernel's; a negative cost: the hand-off tags need L >= +0)
                bad |= 2u;
                x = __builtin_huge_valf();
            }
        }
        reinterpret_cast<float *>(rel8)[p * slots + k] = x;
    } else {
        const unsigned none = cb == 1 ? 255u : 65535u;
        unsigned code = none;
        if (d >= lo && d <= hi && d - dmin >= 0 && d - dmin < L) {
            code = cb == 1 ? c8_encode(C[p * L + (d - dmin)]) : c16_encode(C[p * L + (d - dmin)]);
            if (code > none) {
                bad |= 2u;
                code = none;
            }
        }
        if (cb == 1) rel8[p * slots + k] = (uint8_t)code;
        else reinterpret_cast<uint16_t *>(rel8)[p * slots + k] = (uint16_t)code;
    }
    if (bad && *flag != (*flag | bad)) atomicOr(flag, bad);  // (look before raising: one word for everybody)
}
hipError_t launch_rel_gather(const float *C, const float *rlo, const float *rhi, long long npix, int L, int dmin, int slots, int cb, uint8_t *rel8, int *relb,
                             unsigned *flag, hipStream_t s)
{
    const long long n = npix * slots;
    hipLaunchKernelGGL(k_rel_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, C, rlo, rhi, npix, L, dmin, slots == 128 ? 7 : 6, cb, rel8, relb, flag);
    return hipGetLastError();
}

// The relative copy of a single-word CENSUS volume straight from the descriptor words (integer costs, trunc = +INF or a whole
// number up to 254: every cost is a byte and none is NaN by construction) -- no dense hull in between: a 1920x1080 volume of
// 55-label windows in a hull of 256 cost 1.2 ms (general kernel, fp32 + compact hull) + 0.45 ms (k_rel_gather); this writes
// its 133 MB alone.  cost = min(popcount(cu ^ cv), trunc), trunc for a hypothesis outside
// the right image (mgm_costvolume.h:65-78, 401-412); a pixel without a finite cost in its range is all zeros there (414-421).
template <int SPL>
__global__ void __launch_bounds__(256) k_cost_census_rel(const uint32_t *__restrict__ cu, const uint32_t *__restrict__ cv, int nx, int ny, int vnx, int vny,
                                                         int dmin, int L, unsigned tb, const float *__restrict__ rlo, const float *__restrict__ rhi,
                                                         uint8_t *__restrict__ rel8, int *__restrict__ relb, unsigned *flag)
{
    constexpr int SLOTS = 16 * SPL;
    // four pixels per wave: a row of 16 lanes per pixel, SPL slots (4-byte stores) per lane
    const long long npix = (long long)nx * ny;
    const int li = threadIdx.x & 15;
    const long long pix = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + ((threadIdx.x >> 4) & 3);
    if (pix >= npix) return;
    const int y = (int)(pix / nx), x = (int)(pix - (long long)y * nx);
    const int lo = (int)rlo[pix], hi = (int)rhi[pix];
    const int b = lo - 1;
    const bool yin = y < vny;
    const uint32_t wu = cu[pix];
    const uint32_t *row = cv + (long long)(yin ? y : 0) * vnx;
    unsigned code[SPL];
    bool fin = false;
#pragma unroll
    for (int q = 0; q < SPL; q++) {
        const int d = b + SPL * li + q;
        const bool inw = d >= lo && d <= hi && d - dmin >= 0 && d - dmin < L;  // a label of this pixel
        const int qx = x + d;
        const bool in = yin && qx >= 0 && qx < vnx;
        const unsigned pc = (unsigned)__builtin_popcount(wu ^ row[in ? qx : 0]);
        code[q] = inw ? (in ? (pc < tb ? pc : tb) : tb) : 255u;
        fin = fin || (inw && code[q] != 255u);
    }
    // no finite cost in the pixel's range (its row of 16 lanes): zeros there
    const unsigned long long any = __builtin_amdgcn_ballot_w64(fin);
    if (((any >> (threadIdx.x & 48)) & 0xffffull) == 0ull) {
#pragma unroll
        for (int q = 0; q < SPL; q++) {
            const int d = b + SPL * li + q;
            const bool inw = d >= lo && d <= hi && d - dmin >= 0 && d - dmin < L;
            code[q] = inw ? 0u : 255u;
        }
    }
#pragma unroll
    for (int w = 0; w < SPL / 4; w++)
        reinterpret_cast<unsigned *>(rel8 + pix * SLOTS)[li * (SPL / 4) + w] = code[4 * w] | (code[4 * w + 1] << 8) | (code[4 * w + 2] << 16) | (code[4 * w + 3] << 24);
    if (li == 0) {
        *reinterpret_cast<int4 *>(relb + pix * 4) = make_int4(b, lo, hi, 0);
        const unsigned bad = (hi - lo + 1 > SLOTS - 2 || hi < lo) ? 1u : 0u;
        if (bad && *flag != (*flag | bad)) atomicOr(flag, bad);
    }
}
hipError_t launch_cost_census_rel(const uint32_t *cu, const uint32_t *cv, int nx, int ny, int vnx, int vny, int dmin, int L, float trunc, const float *rlo,
                                  const float *rhi, int slots, uint8_t *rel8, int *relb, unsigned *flag, hipStream_t s)
{
    const unsigned tb = trunc == __builtin_huge_valf() ? 255u : (unsigned)trunc;
    const long long npix = (long long)nx * ny;
    const unsigned grid = (unsigned)((npix + 15) / 16);
    if (slots == 128) hipLaunchKernelGGL(k_cost_census_rel<8>, dim3(grid), dim3(256), 0, s, cu, cv, nx, ny, vnx, vny, dmin, L, tb, rlo, rhi, rel8, relb, flag);
    else hipLaunchKernelGGL(k_cost_census_rel<4>, dim3(grid), dim3(256), 0, s, cu, cv, nx, ny, vnx, vny, dmin, L, tb, rlo, rhi, rel8, relb, flag);
    return hipGetLastError();
}
// ... and the dense fp32 hull of such a volume, for whoever asks for it (ensure_f32): +INF wherever a pixel has no label
__global__ void __launch_bounds__(256) k_rel_expand(const uint8_t *__restrict__ rel8, const int *__restrict__ relb, long long n, int L, int dmin,
                                                    int slots, int cb, float *__restrict__ C)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const long long pix = t / L;
    const int o = (int)(t - pix * L);
    const int k = dmin + o - relb[pix * 4];
    float v = __builtin_huge_valf();
    if (k >= 0 && k < slots)
        v = cb == 1 ? c8_decode((unsigned)rel8[pix * slots + k])
                    : (cb == 2 ? c16_decode((unsigned)reinterpret_cast<const uint16_t *>(rel8)[pix * slots + k]) : reinterpret_cast<const float *>(rel8)[pix * slots + k]);
    C[t] = v;
}
hipError_t launch_rel_expand(const uint8_t *rel8, const int *relb, long long npix, int L, int dmin, int slots, int cb, float *C, hipStream_t s)
{
    const long long n = npix * L;
    hipLaunchKernelGGL(k_rel_expand, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rel8, relb, n, L, dmin, slots, cb, C);
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

// minimum over a row of 16 lanes, in every lane of the row (quad swaps, then the two mirrors: 4 DPP slots, no LDS crossbar)
__device__ __forceinline__ float rel_row_min(float v)
{
    asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
    return v;
}

// ---- the kernel ----------------------------------------------------------------------------------------------------------
// FOUR scan lines per wavefront: a pixel's 64 label slots sit on a ROW of 16 lanes (the unit of the DPP shifts), 4 slots per
// lane, so one issue slot serves four pixels -- the first version of this file put one slot on a lane, one pixel on a wave,
// and its step cost what a 256-label step of the dense kernels costs (min-convolutions: one per NEIGHBOUR here, the receiving
// pixel's range shapes them) for a quarter of the labels: 14 ms per 1920x1080 volume of 55-label windows, VALU-bound.  The
// lane groups of a wave are lines 4w .. 4w+3 of the band, each at its own pixel of the slope-2 diagonal; everything that
// crosses lanes stays inside a row (rel_row_min, the scans with GROUPS = 4, neighbour_min with its edge flags), and lines
// talk through the LDS rings exactly as before.
// Memory side on a LOADER wave (the last of the workgroup), as in the second build: everything a step reads from memory --
// the cost bytes and records of the band's 16 pixels, their edge weights, and the previous band's hand-off slot -- goes
// straight into LDS rings by LDS-DMA, P.ld steps ahead, retired by COUNTED waits; the compute waves touch only LDS and
// issue stores.  (With register loads the compiler drained vmcnt at every step: global_load .. s_waitcnt vmcnt(0).)
// What bounds a step (profiles/r05_rel_phases.txt, a -DMGM_REL_PHASES=1 build): the compute waves' own instruction stream --
// ~1800 clocks Hirschmueller, ~3900 FH (the loader issues its DMAs in 150 and waits for the predecessor band's slot for the rest);
// one wave per SIMD issues a wave64 instruction every four clocks whatever their dependences (three min-convolutions side by side were no faster), so a single launch is bound
// by instructions per step x steps of its chain of bands, a batch by VALU throughput.
// PUBE (unit weights, Hirschmueller): the transform does not depend on the reader then, so the producer publishes
// E[k] = fmin(fmin(L[k], N[k] + P1), m + P2) - m once (one slab; its "minimum" word carries FAR = (m + P2) - m, what the
// expression gives for a disparity the neighbour does not have: L = N = +INF there) and the reader only adds -- the
// terms of update_costW with DeltaI = 1 (mgm_core.cc:104-137; P1 * 1.0f is P1), in its order.
// NK (round 6; FH only): the neighbour count TSGM as a compile-time constant -- the pixel's NK min-convolutions then run SIDE BY
// SIDE in one instruction stream (fh_minconv_multi: independent dependency chains, one vote per direction instead of one per
// array and direction); 0: the run-time loop over the neighbours, one convolution after the other.
// Slope (round 6): form-0 passes with TSGM <= 3 never read the fwd neighbour (i + 1, j - 1), so a line only has to stay ONE pixel
// behind the line before it (g.slope, as in the second build, mgm_pass2.hip): a band then hands over R + lag steps after it
// started instead of 2 R + lag -- the chain of a 1920-column pass of 120 bands drops from 120 x 36 + 1110 to 120 x 20 + 1095.
// SPL, CB (round 6): label slots per lane (4: 64 slots per pixel, windows of up to 62 labels; 8: 128 slots, up to 126) and bytes per
// cost (1, or 2: colour AD / SD).  Everything below is written for SPL values per lane; the loader's DMA counts follow.
// FH2 (round 6): update_cost2_trunclinear (mgm_core.cc:197-219) -- FH potentials with TSGM = 2 and NO weights, the one update function
// with FixBounrady_for_minConvTruncatedLinear (166-186): before the receiving pixel convolves a neighbour's values over its own range
// [pa, pb], what lies OUTSIDE that range in the neighbour's range [qa, qb] is folded into the two end labels -- T_left = the forward
// recurrence over the neighbour's labels from qa up to pa, T_right = the backward one from qb down to pb, each continued as a plain
// ramp where it leaves the neighbour's range.  Those recurrences run over labels the receiving pixel's 64 / 128 slots may not even
// hold, so the PRODUCER computes them once, in its own frame -- F = the forward pass of its raw slab, B = the backward pass --
// and publishes [L][F][B]; a reader picks F at pa (or at qb and ramps on), B at pb (or at qa), and min()s them into its two end
// labels before the convolution.  Then e = (M1 - m1 + M2 - m2) / 2 in the reference's association.
template <bool FH, bool PUBE, int NK = 0, int SPL = 4, int CB = 1, bool FH2 = false>
__global__ void __launch_bounds__((NW + 1) * 64) k_pass_rel(const RelParams P)
{
    static_assert(!FH2 || (FH && NK == 2), "update_cost2_trunclinear: FH, two neighbours, side by side");
    static_assert(!(FH && PUBE), "FH potentials convolve over the RECEIVING pixel's range: consumer side only");
    static_assert(NK == 0 || FH, "side-by-side convolutions: FH only");
    static_assert((SPL == 4 || SPL == 8) && (CB == 1 || CB == 2 || CB == 4), "64 or 128 slots; one- or two-byte cost codes, or the fp32 cost itself");
    constexpr int SLOTS = 16 * SPL;          // label slots per pixel
    constexpr int CBY = SLOTS * CB;          // cost bytes per pixel
    constexpr int CPP = CBY / 16;            // ... in 16-byte pieces
    constexpr int CI = RR * CPP / 64;        // DMA instructions that fetch the cost pieces of the band's RR pixels (1, 2 or 4)
    constexpr int NS = (FH || PUBE) ? 1 : 2;
    // A ring entry = a hand-off slot, whole 16-byte pieces.  One slab (FH, PUBE): [4 guard words][64 values][4 guard words][minimum,
    // base, 2 of padding] -- the guards hold what a disparity the pixel does not have reads as (+INF; PUBE: FAR), so a reader takes
    // its four values at ONE clamped index, no comparison or selection per value (the 64 + 64 variant: [L][N][minimum, base, padding])
    constexpr bool GUARD = NS == 1;
    constexpr int GO = GUARD ? SPL : 0;                 // where the values start (a lane's worth of guard words before and after them)
    constexpr int FOFF = SLOTS + 2 * SPL, BOFF = FOFF + SLOTS;  // FH2: the forward and backward passes of the raw slab, behind the guard-framed values
    constexpr int HOFF = FH2 ? BOFF + SLOTS : (GUARD ? SLOTS + 2 * SPL : NS * SLOTS);  // where the header (minimum, base, highest slot) sits
    constexpr int HS = HOFF + 4;                        // floats per entry / hand-off slot
    constexpr int NPIECE = HS / 4;                      // its 16-byte pieces
    constexpr int HI = (NPIECE + 63) / 64;              // DMA instructions per hand-off slot (2 only for two slabs of 128)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *ring = smem;                                    // [RR][RD4][HS]   what the lines of the band publish
    float *hring = ring + RR * RD4 * HS;                   // [SD][HS]        the previous band's last line, by pixel & (SD - 1)
    float *hring2 = hring + SD * HS;                       // [SD][HS]        (launches with anti-diagonal passes only) its last line but one
    int *mring = reinterpret_cast<int *>(hring2 + (P.diag_any ? SD * HS : 0));  // [SD][RR][4]    records of the step's pixels: base, lo, hi
    float *wring = reinterpret_cast<float *>(mring + SD * RR * 4);  // [SD][RR][4]   edge weights of the step's pixels
    uint8_t *cring = reinterpret_cast<uint8_t *>(wring + SD * RR * 4);  // [SD][RR][CBY] cost bytes of the step's pixels
    unsigned *hprog = reinterpret_cast<unsigned *>(cring + SD * RR * CBY);  // [4]   scratch of the loader's slow path (the error word)
    int *s_task = reinterpret_cast<int *>(hprog + 4);
    constexpr int LTW = 24;           // words per line of the table below: 4 of ranges + 4 per neighbour (+ 4 spare)
    int *ltab = s_task + 4;           // [RR][LTW]  the geometry table of the band's lines (compute waves, below)

    const int tid = threadIdx.x, lane = tid & 63;
    const int r = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) {
        *s_task = (int)atomicAdd(P.ticket, 1u);
        if (P.tl) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            P.tl[(long long)*s_task * 8 + 0] = wall_clock64();
            P.tl[(long long)*s_task * 8 + 4] = ((unsigned long long)(xcc & 15u) << 32) | hwid;
        }
    }
    if (!P.weighted)  // unit weights: the ring holds ones, nothing is fetched
        for (int k = tid; k < SD * RR * 4; k += (NW + 1) * 64) wring[k] = 1.0f;
    if constexpr (GUARD)  // entries nobody has written yet read as "no such disparity" everywhere
        for (int k = tid; k < (RR * RD4 + (P.diag_any ? 2 : 1) * SD) * HS; k += (NW + 1) * 64) ring[k] = f_inf();
    __syncthreads();
    const int2 tk = P.tasks[*s_task];
    const int vp = tk.x, band = tk.y & 0xffff, strip = (tk.y >> 16) & 0xff;  // vp = volume*8 + pass
    // the bands of the launch's LONGEST chains issue first where two workgroups share a SIMD (the host marks them: run_rel)
    if ((tk.y >> 24) & 1) __builtin_amdgcn_s_setprio(2);
    const int pass = vp & (kMaxDirs - 1);
    const RelVolume &V = P.vol[vp / kMaxDirs];
    const PassGeom &g = P.g[pass];
    const int NL = g.NL, LL = g.LL, MGM = P.MGM, form = g.form;
    const float P1 = P.P1, P2 = P.P2;
    const int SL = g.slope;  // line l of the band is at pixel s - 1 - SL l at step s
    // TWO STRIPS per line (round 6; form 1 with TSGM <= 3: a pixel depends on the line before only, never on its own line -- the
    // second build's `strips`, mgm_pass2.hip): this work item walks strip `strip` of its band's lines, strip 0 = [0, split) upwards
    // from pixel 0, strip 1 = [split, LL) downwards from pixel LL - 1 -- in MIRRORED coordinates i' = LL - 1 - i, in which it is
    // again a walk upwards from 0 with the fwd and back neighbours at i' - 1 and i' + 1.  So that neither strip ever needs the
    // other's lines of the SAME band, line l of the band is walked over [0, W - l), W = strip length + lines - 1: the band's last
    // line covers exactly the strip, the lines above it a little more (the pixels both items compute get the same values twice).
    // What a strip needs from the other comes through the previous band's hand-off slots, which are indexed by ABSOLUTE pixel and
    // validate themselves.  The chain of a pass is then 2 steps per line + HALF a line.  Everything below runs in LOCAL
    // coordinates (rings, steps); only addresses and hand-off slots are absolute.
    const bool strips = g.nstrips == 2;
    const bool mirror = strips && strip == 1;
    const int slen = !strips ? LL : (strip == 0 ? g.split : LL - g.split);
    // ANTI-DIAGONALS (round 6; g.diag: form 1 with TSGM <= 3 again).  All three neighbours of such a pixel sit on the line before it --
    // (i + 1, j - 1), (i - 1, j - 1), (i, j - 1) -- so the dependency depth of the pass is its number of LINES, and a walk along the lines
    // (two steps of lag per line: 2 NL steps, whatever the strips) wastes it.  Here the walk runs ACROSS the lines: step u is line j = u, and
    // the "lines" of the bands are the anti-diagonals d = i + j, on which the three neighbours become (u - 1, d), (u - 1, d - 2),
    // (u - 1, d - 1): the previous step of the own diagonal and of the two before it.  Every diagonal of a band is at the same step
    // (slope 0), a band needs the last TWO diagonals of the band before it one step earlier, and the chain of a pass is
    // NL + bands x (1 + lag) steps instead of 2 NL + LL / 2 + bands x lag.  Diagonal d has its pixels at u in [max(0, d - LL + 1),
    // min(NL - 1, d)]; the band walks the union of its diagonals' ranges, [dlo, dhi], in local steps i = u - dlo.
    const bool diag = g.diag != 0;
    const int NLd = diag ? NL + LL - 1 : NL;                                   // lines of the walk
    const int dlo = diag ? max(0, band * RR - LL + 1) : 0;                     // first line j of the original pass this band touches
    const int W = diag ? min(NL - 1, band * RR + RR - 1) - dlo + 1 : (!strips ? LL : min(LL, slen + RR - 1));  // pixels of the band's first line / steps of the band
    const long long istep = diag ? g.jstep - g.istep : (mirror ? -g.istep : g.istep);  // along the walk
    const long long lstep = diag ? g.istep : g.jstep;                                   // from line to line
    const long long gbase = diag ? g.base + (long long)dlo * istep : (mirror ? g.base + (long long)(LL - 1) * g.istep : g.base);
    const int nsteps = (W + 1 + SL * (RR - 1) + 3) / 4 * 4;
    // line l of the band is walked over local pixels [la(l), lb(l)]
    auto la = [&](int l) { return diag ? max(0, min(band * RR + l, NLd - 1) - LL + 1) - dlo : 0; };
    auto lb = [&](int l) { return diag ? min(NL - 1, min(band * RR + l, NLd - 1)) - dlo : W - 1; };
    const int wmax = g.wmax;

    // SELF-VALIDATING hand-off slots, one per (volume, pass, band, pixel), written once per launch with the launch's tag in the
    // sign bit of every word (values, minimum and biased base are non-negative; the padding words carry the tag alone): the
    // reading band fetches a slot when it wants it and looks at the sign bits -- no progress words, no publication lag, no
    // counted waits on the writing side (the second build's protocol, mgm_pass2.hip TAGS).  With progress words a band ran
    // ~51 steps behind its predecessor where the geometry asks for 32: 120 bands x 19 steps of a 6600-step chain.
    const long long hslot0 = (long long)(vp / kMaxDirs) * P.hand_vstride + g.hand_base;
    // (anti-diagonals: two lines of wmax slots per band, by the band's local step)
    const long long hper = diag ? 2LL * wmax : (long long)LL;
    float *hand_out = P.hand + (hslot0 + (long long)band * hper) * HS;
    const float *hand_in = P.hand + (hslot0 + (long long)(band > 0 ? band - 1 : 0) * hper) * HS;  // (band 0 issues the same DMAs, from its own slots)
    const unsigned tag = P.tag;

    if (r == NW) {
        // =========================== loader wave ===========================
        // Step t reads: the cost bytes (one 16-byte piece per lane: lane 4 l + c has piece c of line l's pixel), the records
        // (lane l < RR), the edge weights (lane 4 l + k: neighbour k of line l's pixel) of pixel t - 1 - 2 l of every line l,
        // and pixel t of the previous band's last line.  All of it is requested LD steps ahead, straight into the rings'
        // slot t & (SD - 1), every step the same number of DMA instructions (addresses clamped at the ends of a line: what
        // lands for a pixel that does not exist is never looked at), retired by count: nothing here waits for a round trip
        // except the slow path of the hand-off.
        const bool from_global = band > 0;
        const bool weighted = P.weighted != 0;
        // cost pieces: instruction n moves piece (lane + 64 n) % CPP of line (lane + 64 n) / CPP -- [line][piece] order in the ring
        const uint8_t *cptr[CI];
        int ci[CI], ca[CI], cb[CI];  // (the pointer stands at pixel clamp(ci, ca, cb) of its line)
#pragma unroll
        for (int n = 0; n < CI; n++) {
            const int cl = (lane + 64 * n) / CPP;
            const int cj = min(band * RR + cl, NLd - 1);
            ca[n] = la(cl), cb[n] = lb(cl);
            cptr[n] = V.c8 + (gbase + (long long)cj * lstep + (long long)ca[n] * istep) * CBY + ((lane + 64 * n) % CPP) * 16;
            ci[n] = -1 - SL * cl;
        }
        const int wl = lane >> 2;  // edge weights: lane 4 l + k has neighbour k of line l's pixel
        const int wa = la(wl), wb = lb(wl);
        const long long wpix0 = gbase + (long long)min(band * RR + wl, NLd - 1) * lstep + (long long)wa * istep;
        const float *wptr = weighted ? V.w8 + (long long)g.wplane[lane & 3] * P.npix + wpix0 : nullptr;
        int wi = -1 - SL * wl;
        const int ml = lane < RR ? lane : RR - 1;
        const int mj = min(band * RR + ml, NLd - 1);
        const int ma = la(ml), mb = lb(ml);
        const int4 *mptr = reinterpret_cast<const int4 *>(V.base) + (gbase + (long long)mj * lstep + (long long)ma * istep);
        int mi = -1 - SL * ml;
        // the previous band's slots of local pixel 0 (anti-diagonals: its local step of the same line j, dsh further on; first the last
        // line but one, wmax slots on the last)
        const int dsh = diag ? dlo - max(0, (band > 0 ? band - 1 : 0) * RR - LL + 1) : 0;
        const int hlast = diag ? wmax - 1 - dsh : LL - 1;  // the last local pixel that has a slot
        const float *hptr = hand_in + (mirror ? (long long)(LL - 1) * HS : (long long)dsh * HS) + lane * 4;  // (local pixel 0)
        const long long hstep = mirror ? -(long long)HS : (long long)HS;
        int ht = 0;
        bool dead = false;
        unsigned long long tl_wait = 0, n_slow = 0, n_spin = 0;
        auto issue = [&]() {  // everything step `ht` reads
            const int slot = ht & (SD - 1);
#pragma unroll
            for (int n = 0; n < CI; n++) {
                rel_dma16<0>(cptr[n], cring + slot * RR * CBY + n * 1024);
                const bool adv = ci[n] >= ca[n] && ci[n] < cb[n];
                cptr[n] += adv ? istep * CBY : 0;
                ci[n]++;
            }
            if (weighted) {
                rel_dma4<0>(wptr, wring + slot * RR * 4);
                const bool adv = wi >= wa && wi < wb;
                wptr += adv ? istep : 0;
                wi++;
            }
            if (lane < RR) rel_dma16<0>(mptr, mring + slot * RR * 4);
            {
                const bool adv = mi >= ma && mi < mb;
                mptr += adv ? istep : 0;
                mi++;
            }
            // the previous band's slot of pixel ht, whatever it holds by now: validate() looks at it when its step comes
#pragma unroll
            for (int n = 0; n < HI; n++)
                if (lane + 64 * n < NPIECE) rel_dma16<REL_SC1>(hptr + (diag ? (long long)wmax * HS : 0) + 256 * n, hring + slot * HS + 256 * n);
            if (diag)
#pragma unroll
                for (int n = 0; n < HI; n++)
                    if (lane + 64 * n < NPIECE) rel_dma16<REL_SC1>(hptr + 256 * n, hring2 + slot * HS + 256 * n);
            hptr += (ht < hlast) ? hstep : 0;
            ht++;
        };
        // The slot of pixel t has landed in the hand ring: make sure it is THIS launch's (every word's sign bit = the tag),
        // fetching it again until it is; then take the tags off (and the bias off the base) so that the compute waves read it like
        // any ring entry.
        auto validate_one = [&](float *ent, const float *slot) {
            rel_u4 v[HI];
            unsigned spins = 0;
            unsigned long long w0 = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int n = 0; n < HI; n++)
                    if (lane + 64 * n < NPIECE) {
                        v[n] = rel_lds_read128_opaque(ent + (lane + 64 * n) * 4);
                        ok = ok && (tag ? ((v[n].x & v[n].y & v[n].z & v[n].w) >> 31) != 0u : ((v[n].x | v[n].y | v[n].z | v[n].w) >> 31) == 0u);
                    }
                if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                if (spins == 0) {
                    n_slow++;
                    if (P.tl) w0 = wall_clock64();
                }
                n_spin++;
                __builtin_amdgcn_s_sleep(2);
#pragma unroll
                for (int n = 0; n < HI; n++)
                    if (lane + 64 * n < NPIECE) rel_dma16<REL_SC1>(slot + (lane + 64 * n) * 4, ent + 256 * n);
                rel_wait_vmcnt<0>();
                if (((++spins) & 255u) == 0) {
                    if (lane == 0) rel_dma4<REL_SC1>(P.err, hprog);
                    rel_wait_vmcnt<0>();
                    const unsigned e = (unsigned)__builtin_amdgcn_readfirstlane((int)rel_lds_read_opaque(hprog));
                    if (spins > (SPIN_LIMIT >> 2) || e != 0) {
                        if (lane == 0) __hip_atomic_store(P.err, 1u, RLX_AGENT);
                        dead = true;
                        break;
                    }
                }
            }
            if (P.tl && w0) tl_wait += wall_clock64() - w0;
#pragma unroll
            for (int n = 0; n < HI; n++)
                if (lane + 64 * n < NPIECE) {
                    v[n].x &= 0x7fffffffu, v[n].y &= 0x7fffffffu, v[n].z &= 0x7fffffffu, v[n].w &= 0x7fffffffu;
                    if (lane + 64 * n == HOFF / 4) v[n].y -= (unsigned)REL_BIAS;  // the piece (minimum, base + bias, padding, padding)
                    rel_lds_write128_opaque(ent + (lane + 64 * n) * 4, v[n]);
                }
        };
        auto validate = [&](int t) {
            if (!from_global || dead) return;
            if (!diag) {
                if (t >= LL || t > W) return;  // (beyond local pixel W nobody of this item reads: the other strip's business)
                validate_one(hring + (t & (SD - 1)) * HS, hand_in + (long long)(mirror ? LL - 1 - t : t) * HS);
            } else {
                if (t >= W) return;
                const int u = dlo + t;  // the slots of line j = u: written where the previous band's last two diagonals have a pixel there
#pragma unroll
                for (int which = 1; which >= 0; which--) {
                    const int dp = (band - 1) * RR + RR - 2 + which;
                    if (u >= max(0, dp - LL + 1) && u <= min(NL - 1, dp) && !dead)
                        validate_one((which ? hring : hring2) + (t & (SD - 1)) * HS, hand_in + ((long long)which * wmax + t + dsh) * HS);
                }
            }
        };
        const int LD = P.ld;  // steps of DMA in flight (2 .. 5: the rings have SD = 8 slots, three of them being read)
        auto retire = [&]() {  // all but the newest LD - 1 steps of DMA have landed
            const int n = (CI + 1 + (diag ? 2 * HI : HI) + (weighted ? 1 : 0)) * (LD - 1);  // (3 .. 14 instructions per step, 1 .. 4 steps)
            switch (n) {
#define MGM_REL_W(k) case k: rel_wait_vmcnt<k>(); break;
#define MGM_REL_W4(k) MGM_REL_W(k) MGM_REL_W(k + 1) MGM_REL_W(k + 2) MGM_REL_W(k + 3)
                MGM_REL_W(3) MGM_REL_W4(4) MGM_REL_W4(8) MGM_REL_W4(12) MGM_REL_W4(16) MGM_REL_W4(20) MGM_REL_W4(24) MGM_REL_W4(28) MGM_REL_W4(32) MGM_REL_W4(36)
                MGM_REL_W4(40) MGM_REL_W4(44) MGM_REL_W4(48) MGM_REL_W4(52) MGM_REL_W(56)
#undef MGM_REL_W4
#undef MGM_REL_W
            default: rel_wait_vmcnt<0>(); break;  // (a count not listed: wait for everything -- correct, only slower)
            }
        };
        // Step s + 1 reads, of the previous band's line(s), the slots of local pixels <= s + 1 where the walk reads the fwd neighbour
        // (slope 2), <= s where it does not (slope 1), <= s - 1 on the anti-diagonals (every neighbour one step back): the slot made sure of
        // during step s is s + vlead -- each one less is a step less of distance between a band and the band before it
        const int vlead = diag ? -1 : (SL == 1 ? 0 : 1);
#pragma unroll 1
        for (int u = 0; u < LD; u++) issue();
        retire();
        if (vlead > 0) validate(0);
        lds_barrier();  // B0
        unsigned long long lph[3] = {0, 0, 0};
        (void)lph;
#pragma unroll 1
        for (int s = 0; s < nsteps; s++) {
            const unsigned long long l0 = MGM_REL_PHASES ? clock64() : 0;
            issue();   // step s + LD
            const unsigned long long l1 = MGM_REL_PHASES ? clock64() : 0;
            retire();  // step s + 1 is in the rings
            if (s + vlead >= 0) validate(s + vlead);
            const unsigned long long l2 = MGM_REL_PHASES ? clock64() : 0;
            lds_barrier();
            if constexpr (MGM_REL_PHASES != 0) {
                const unsigned long long l3 = clock64();
                lph[0] += l1 - l0, lph[1] += l2 - l1, lph[2] += l3 - l2;
            }
        }
        rel_wait_vmcnt<0>();  // (the DMAs beyond the last step)
        if constexpr (MGM_REL_PHASES != 0)
            if (P.tl && lane == 0) {
                unsigned long long *w = P.tl + (long long)gridDim.x * 8 + (long long)*s_task * 16 + 12;
                w[0] = lph[0], w[1] = lph[1], w[2] = lph[2];
            }
        if (P.tl && lane == 0) {
            unsigned long long *w = P.tl + (long long)*s_task * 8;
            w[1] = wall_clock64();
            w[2] = tl_wait;
            w[3] = n_slow;
            w[5] = (unsigned long long)nsteps;
            w[6] = n_spin;
        }
        return;
    }

    // ============================= compute waves =============================
    const int grp = lane >> 4, li = lane & 15;  // the lane's line within the wave; its label slots are SPL li .. SPL li + SPL - 1
    const int ln = GL * r + grp;                // line within the band = ring row
    const int j = band * RR + ln;
    const bool line_ok = j < NLd;
    const bool has_prev = line_ok && (j >= 1);
    const bool to_global = (r == NW - 1) && (band + 1 < g.nbands);  // (the wave that holds the band's last line)
    const int prow = ln > 0 ? ln - 1 : RR;  // ring row of the line before this one
    const int prow2 = ln > 1 ? ln - 2 : RR + 1 - ln;  // ... and of the one before that (anti-diagonals; RR + 1: the second hand ring)
    float *__restrict__ Lrb = V.Lr + (long long)(pass - P.pass0) * P.nvol;
    const long long pix0 = gbase + (long long)(line_ok ? j : 0) * lstep;
    // Everything the walk's geometry decides -- line walk or anti-diagonals, the pass's form, a mirrored strip -- is settled HERE, per LINE,
    // in a small LDS table the step loop reads the same way whatever the geometry:
    //   the line's pixels [Wa, Wb]; those among them that take neighbours [Ia, Ib] (mgm_core.cc:538-541: not on the image's border);
    //   neighbour k's ring row (its first entry + the mask of its ring) and its step relative to this pixel's.
    // (The first anti-diagonal build selected at every step: 15 more live scalar registers, spilled, x 4 13.0 -> 17.1 ms with the switch off;
    // the second kept these in registers per lane: 100 instead of 92, four waves per SIMD instead of five, and the third workgroup of a
    // CU often found no room -- 14.3 ms; forcing 96 registers spilled the Lr pointer: a scratch reload + vmcnt(0) behind every step's
    // stores.  A read of 16 bytes per use instead: the values are dead again before the convolutions need the registers.)
    const int Wa = la(ln);
    const int Wl = diag ? lb(ln) + 1 : (!strips ? LL : min(LL, slen + RR - 1 - ln));
    const int Wb = line_ok ? Wl - 1 : Wa - 1;
    const int i0d = j - dlo;  // anti-diagonals: the pixel's place on its line of the original pass is i0d - i, its line dlo + i
    // (g.swap: the roles of i and j exchanged -- the pixels that take neighbours are those of the lines 1 .. NL - 2 from pixel 1 on)
    const bool swp = g.swap != 0;
    const int Ia = diag ? max(Wa, max(1 - dlo, i0d - LL + 2)) : 1;
    const int Ib = diag ? min(Wb, i0d - 1) : (swp ? ((has_prev && j <= NL - 2) ? Wb : 0) : (has_prev ? min(Wb, LL - 2) : 0));
    const bool f0 = form == 0;
    int *const lt = ltab + ln * LTW;
    if (li == 0) *reinterpret_cast<int4 *>(lt) = make_int4(Wa, Wb, Ia, Ib);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // form 0: the pixel before on this line, then the line before at i, i - 1, i + 1; form 1: the same four the other way round (mgm_core.cc:520-575);
        // a mirrored strip: the fwd neighbour i + 1 is local i' - 1; anti-diagonals: fwd, back, same = the previous step of this line, of the
        // line two before, of the line before
        const bool own = f0 ? k == 0 : k == 3;
        const int di0 = f0 ? (k == 1 ? 0 : (k == 2 ? -1 : 1)) : (k == 0 ? 1 : (k == 1 ? -1 : 0));
        // exchanged roles (form 0, TSGM <= 3): the first neighbour (i - 1, j) of the pass is the same pixel of the line before, the second
        // (i, j - 1) the pixel before on this line, the third (i - 1, j - 1) stays what it was
        const int row = diag ? (k == 0 ? ln : (k == 1 ? prow2 : prow)) : (swp ? (k == 1 ? ln : prow) : (own ? ln : prow));
        const int nstep = swp ? (k == 0 ? 0 : -1) : ((diag || own) ? -1 : (mirror ? -di0 : di0));
        const int nent = (int)((row >= RR ? (row == RR ? hring : hring2) : ring + row * RD4 * HS) - smem);
        if (li == 0) *reinterpret_cast<int4 *>(lt + 4 + 4 * k) = make_int4(nent, row >= RR ? SD - 1 : RD4 - 1, nstep, 0);
    }
    // slab(s) of neighbour k as the pixel with base bp sees them: the same disparities, +INF (PUBE: FAR) where the neighbour has no slot for them
    // (read at the END of a step for the next one -- behind the convolutions, ahead of the barrier's wait: no register of it is live while
    // the convolutions run, and no step begins with a round trip to LDS: x 1 7.3 -> 6.7 ms against reading at the step's head)
    constexpr int NKK = NK > 0 ? NK : 4;
    int4 lr, nt[NKK];
    auto read_table = [&]() {
        asm volatile("" ::: "memory");  // (the table must not come back as loop-invariant registers)
        lr = *reinterpret_cast<const int4 *>(lt);  // Wa, Wb, Ia, Ib
#pragma unroll
        for (int k = 0; k < NKK; k++) nt[k] = *reinterpret_cast<const int4 *>(lt + 4 + 4 * k);
    };
    auto entry = [&](int k, int i) -> const float * { return smem + nt[k].x + ((i + nt[k].z) & nt[k].y) * HS; };
    // the hand-off to the next band: the lanes that hold its source lines (the band's last line; anti-diagonals: its last two), slot of local pixel 0, stride
    // (these two stay in registers: the band's last wave needs them behind its convolutions, where a table read would be a round trip)
    const bool pubs = to_global && (diag ? grp >= GL - 2 : grp == GL - 1);
    const int hpub0 = (diag ? (grp - (GL - 2)) * wmax : (mirror ? LL - 1 : 0)) * HS;  // (floats: < 2^31, the host checks)
    const int hpubs = mirror ? -HS : HS;

    lds_barrier();  // B0: the loader's first step has landed
    read_table();
    unsigned long long cph[3] = {0, 0, 0}, nsweeps = 0;
    (void)cph;
    (void)nsweeps;
    for (int s = 0; s < nsteps; s++) {
        const unsigned long long c0 = MGM_REL_PHASES ? clock64() : 0;
        unsigned long long c1 = c0;
        const int i = s - 1 - SL * ln;
        const bool act = i >= lr.x && i <= lr.y;
        if (__builtin_amdgcn_ballot_w64(act) != 0ull) {
            const long long pix = pix0 + (long long)(act ? i : 0) * istep;
            const int sl = s & (SD - 1);
            const int4 rec = *reinterpret_cast<const int4 *>(mring + (sl * RR + ln) * 4);
            const int bp = rec.x;
            float Cv[SPL];
            {
                constexpr int NWORD = SPL * CB / 4;  // the lane's cost codes as 32-bit words
                unsigned cw[NWORD];
#pragma unroll
                for (int w = 0; w < NWORD; w++) cw[w] = reinterpret_cast<const unsigned *>(cring + (sl * RR + ln) * CBY + SPL * CB * li)[w];
#pragma unroll
                for (int q = 0; q < SPL; q++)
                    Cv[q] = CB == 1 ? c8_decode((cw[q / 4] >> (8 * (q % 4))) & 255u)
                                    : (CB == 2 ? c16_decode((cw[q / 2] >> (16 * (q % 2))) & 65535u) : __builtin_bit_cast(float, cw[q * CB / 4]));
            }
            const bool interior = i >= lr.z && i <= lr.w;  // mgm_core.cc:538-541
            const float4 w4 = *reinterpret_cast<const float4 *>(wring + (sl * RR + ln) * 4);
            const float D[4] = {w4.x, w4.y, w4.z, w4.w};
            const int rl = rec.y - bp, rh = rec.z - bp;  // the pixel's own range, in slots
            // the MGM neighbours the update reads, in the pass's order (form 0: the pixel before on this line, then the line
            // before at i, i - 1, i + 1; the other form: the same four the other way round) -- mgm_core.cc:520-575
            float e[SPL];
#pragma unroll
            for (int q = 0; q < SPL; q++) e[q] = 0.0f;
            if constexpr (NK > 0) {
                // the NK neighbours' slabs over this pixel's range, convolved side by side, summed in the pass's order
                float Mk[NK][SPL], mk[NK], p1k[NK], p2k[NK];
#pragma unroll
                for (int k = 0; k < NK; k++) {
                    const float *src = entry(k, i);
                    const relf4 hdr = *reinterpret_cast<const relf4 *>(src + HOFF);  // (minimum, base, highest slot, -): one unconditional 16-byte read
                    const float hmin = hdr.x, hbase = hdr.y, hhi = hdr.z;  // (scalars first: __builtin_bit_cast of the vector ELEMENT hdr[1] read element 0)
                    mk[k] = interior ? hmin : 0.0f;
                    const int sh = bp - __builtin_bit_cast(int, hbase);
                    const int idx0 = interior ? min(max(SPL * li + sh, -SPL), SLOTS) + GO : 0;
                    float wv[SPL];  // (read first, unconditionally: a select per value, not an exec-masked LDS read per value)
#pragma unroll
                    for (int q = 0; q < SPL; q++) wv[q] = src[idx0 + q];
#pragma unroll
                    for (int q = 0; q < SPL; q++) {
                        const int o = SPL * li + q;
                        Mk[k][q] = (o >= rl && o <= rh) ? wv[q] : f_inf();
                    }
                    p1k[k] = P1 * D[k];
                    p2k[k] = P2 * D[k];
                    if constexpr (FH2) {
                        // FixBounrady_for_minConvTruncatedLinear (mgm_core.cc:166-186), slots of the NEIGHBOUR's frame: its range is
                        // [1, qh], this pixel's range there [rl + sh, rh + sh]
                        const int qh = interior ? __builtin_bit_cast(int, hhi) : 1;
                        const int pa = rl + sh, pb = rh + sh;
                        const bool hasL = interior && 1 < pa, hasR = interior && qh > pb;
                        float TL = src[FOFF + (hasL ? min(pa, qh) : 1)];  // the forward pass at pa -- or at the neighbour's last label, a ramp from there
                        float TR = src[BOFF + (hasR ? max(pb, 1) : 1)];   // the backward pass at pb -- or at its first label
                        const int nL = hasL ? max(0, pa - qh) : 0, nR = hasR ? max(0, 1 - pb) : 0;
                        for (int t = 0; __builtin_amdgcn_ballot_w64(t < nL || t < nR) != 0ull; t++) {  // T = min(T + P1, +INF), one label at a time
                            TL = t < nL ? TL + p1k[k] : TL;
                            TR = t < nR ? TR + p1k[k] : TR;
                        }
#pragma unroll
                        for (int q = 0; q < SPL; q++) {
                            const int o = SPL * li + q;
                            if (hasL && o == rl) Mk[k][q] = fminf(Mk[k][q], TL);
                            if (hasR && o == rh) Mk[k][q] = fminf(Mk[k][q], TR);
                        }
                    }
                }
                unsigned sw = 0;
                fh_minconv_multi<SPL, GL, NK>(Mk, mk, p1k, p2k, lane, sw);
                if constexpr (MGM_REL_PHASES != 0) nsweeps += sw;
                if constexpr (FH2) {  // (M1[o] - min1 + M2[o] - min2) / 2, left to right (mgm_core.cc:216)
#pragma unroll
                    for (int q = 0; q < SPL; q++) e[q] = (((Mk[0][q] - mk[0]) + Mk[1][q]) - mk[1]) * 0.5f;
                } else
#pragma unroll
                for (int k = 0; k < NK; k++)
#pragma unroll
                    for (int q = 0; q < SPL; q++) e[q] = k == 0 ? Mk[k][q] - mk[k] : e[q] + (Mk[k][q] - mk[k]);
            } else
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (k < MGM) {
                    const float *src = entry(k, i);
                    const float m = interior ? src[HOFF] : 0.0f;                                   // minimum (or FAR)
                    const int sh = bp - reinterpret_cast<const int *>(src)[HOFF + 1];              // base
                    float w[NS][SPL];
                    if constexpr (GUARD) {
                        // the lane's values at one index clamped into [first guard, last guard]: whatever lies outside the neighbour's
                        // slots reads a guard word (a pixel that takes no neighbours reads guard words only)
                        const int idx0 = interior ? min(max(SPL * li + sh, -SPL), SLOTS) + GO : 0;
#pragma unroll
                        for (int q = 0; q < SPL; q++) w[0][q] = src[idx0 + q];
                    } else {
                        const float far = PUBE ? m : f_inf();
#pragma unroll
                        for (int q = 0; q < SPL; q++) {
                            const int idx = SPL * li + q + sh;
                            const bool in = interior && (unsigned)idx < (unsigned)SLOTS;
#pragma unroll
                            for (int t = 0; t < NS; t++) w[t][q] = in ? src[t * SLOTS + (in ? idx : 0)] : far;
                        }
                    }
                    if constexpr (PUBE) {
                        // e = 0; e += t1 - m1; ... (0 + x is x: x >= +0); update_cost2 (TSGM = 2 without weights, mgm_core.cc:66-90)
                        // halves every term before it is added: e += (t1 - m1) / 2; e += (t2 - m2) / 2; L = C + e
                        const float hf = P.cost2 ? 0.5f : 1.0f;
#pragma unroll
                        for (int q = 0; q < SPL; q++) e[q] = k == 0 ? w[0][q] * hf : e[q] + w[0][q] * hf;
                    } else if constexpr (!FH) {  // update_costW (mgm_core.cc:95-144): e = 0; e += fmin3(L, N + P1 D, m + P2 D) - m
                        const float a = P1 * D[k], b = P2 * D[k];
#pragma unroll
                        for (int q = 0; q < SPL; q++) e[q] += fminf(fminf(w[0][q], w[NS - 1][q] + a), m + b) - m;
                    } else {  // update_costW_trunclinear (229-281): the neighbour's values over THIS pixel's range, convolved there
                        float M[SPL];
#pragma unroll
                        for (int q = 0; q < SPL; q++) {
                            const int o = SPL * li + q;
                            M[q] = (o >= rl && o <= rh) ? w[0][q] : f_inf();
                        }
                        unsigned sw = 0;
                        fh_minconv<SPL, false, GL>(M, m, P1 * D[k], P2 * D[k], lane, SLOTS, sw);
                        if constexpr (MGM_REL_PHASES != 0) nsweeps += sw;
#pragma unroll
                        for (int q = 0; q < SPL; q++) e[q] = k == 0 ? M[q] - m : e[q] + (M[q] - m);
                    }
                }
            }
            float Lv[SPL];
#pragma unroll
            for (int q = 0; q < SPL; q++)  // (NK: the divisor folds at compile time; update_cost2 has divided already)
                Lv[q] = interior ? Cv[q] + (((PUBE && P.cost2) || FH2) ? e[q] : div_small_rt(e[q], NK > 0 ? NK : MGM)) : Cv[q];
            if constexpr (MGM_REL_PHASES != 0) {
                asm volatile("" : "+v"(Lv[0]), "+v"(Lv[1]), "+v"(Lv[2]), "+v"(Lv[3]));
                c1 = clock64();
            }
            read_table();  // (for the next step: the convolutions are over, the stores and the publication below cover the reads)
            if (act)
#pragma unroll
                for (int h = 0; h < SPL / 4; h++)
                    reinterpret_cast<relf4 *>(Lrb + pix * SLOTS + SPL * li)[h] = relf4{Lv[4 * h], Lv[4 * h + 1], Lv[4 * h + 2], Lv[4 * h + 3]};

            // what this pixel publishes: its raw slab (Hirschmueller: and the neighbour minima), minimum, base -- or (PUBE) E and FAR
            float lmin = Lv[0];
#pragma unroll
            for (int q = 1; q < SPL; q++) lmin = fminf(lmin, Lv[q]);
            const float m = rel_row_min(lmin);
            float N[SPL];
#pragma unroll
            for (int q = 0; q < SPL; q++) N[q] = f_inf();
            if constexpr (!FH) neighbour_min<SPL>(Lv, N, li == 0, li == 15);
            float pubv[SPL];
#pragma unroll
            for (int q = 0; q < SPL; q++) pubv[q] = Lv[q];
            float pubm = m;
            if constexpr (PUBE) {
                const float cap = m + P2;
#pragma unroll
                for (int q = 0; q < SPL; q++) pubv[q] = fminf(fminf(Lv[q], N[q] + P1), cap) - m;
                pubm = cap - m;
            }
            float Fv[SPL], Bv[SPL];  // FH2: the forward and the backward pass of the raw slab in this pixel's own frame
            if constexpr (FH2) {
#pragma unroll
                for (int q = 0; q < SPL; q++) Fv[q] = Bv[q] = Lv[q];
                unsigned sw = 0;
                fh_scan<SPL, true, GL>(Fv, P1, lane, sw);
                fh_scan<SPL, false, GL>(Bv, P1, lane, sw);
            }
            const float farv = PUBE ? pubm : f_inf();  // what a disparity this pixel does not have reads as
            const relf4 far4 = {farv, farv, farv, farv};
            const relf4 hdr4 = {pubm, __builtin_bit_cast(float, bp), __builtin_bit_cast(float, rh), 0.0f};  // (minimum, base, highest slot of the range)
            constexpr int GP = SPL / 4;  // 16-byte pieces of a guard (a lane's worth of words) and of a lane's values
            if (act) {
                float *ent = ring + (ln * RD4 + (i & (RD4 - 1))) * HS;
#pragma unroll
                for (int h = 0; h < GP; h++) {
                    reinterpret_cast<relf4 *>(ent + GO + SPL * li)[h] = relf4{pubv[4 * h], pubv[4 * h + 1], pubv[4 * h + 2], pubv[4 * h + 3]};
                    if constexpr (NS == 2) reinterpret_cast<relf4 *>(ent + SLOTS + SPL * li)[h] = relf4{N[4 * h], N[4 * h + 1], N[4 * h + 2], N[4 * h + 3]};
                    if constexpr (FH2) {
                        reinterpret_cast<relf4 *>(ent + FOFF + SPL * li)[h] = relf4{Fv[4 * h], Fv[4 * h + 1], Fv[4 * h + 2], Fv[4 * h + 3]};
                        reinterpret_cast<relf4 *>(ent + BOFF + SPL * li)[h] = relf4{Bv[4 * h], Bv[4 * h + 1], Bv[4 * h + 2], Bv[4 * h + 3]};
                    }
                }
                if constexpr (GUARD) {  // lanes 0 .. GP-1: the guard before the values, GP .. 2 GP - 1: the one after, 2 GP: the header
                    if (li < 2 * GP + 1)
                        *reinterpret_cast<relf4 *>(ent + (li < GP ? 4 * li : (li < 2 * GP ? GO + SLOTS + 4 * (li - GP) : HOFF))) = li == 2 * GP ? hdr4 : far4;
                } else {
                    if (li == 0) *reinterpret_cast<relf2 *>(ent + HOFF) = relf2{pubm, __builtin_bit_cast(float, bp)};
                }
            }
            if (to_global) {
                // (strips: the band's last line covers exactly its strip -- no slot is written twice; anti-diagonals: the last TWO lines, each where it has a pixel)
                if (__builtin_amdgcn_ballot_w64(pubs && act) != 0ull) {
                    // the hand-off to the next band: write-through 16-byte stores straight from the registers, every word tagged
                    float *dstg = hand_out + (hpub0 + i * hpubs);
                    if (pubs && act) {
                        auto tagged = [&](relf4 x) {
                            rel_u4 u = __builtin_bit_cast(rel_u4, x);
                            u.x |= tag, u.y |= tag, u.z |= tag, u.w |= tag;
                            return __builtin_bit_cast(relf4, u);
                        };
#pragma unroll
                        for (int h = 0; h < GP; h++) {
                            rel_st_sc1_x4(dstg + GO + SPL * li + 4 * h, tagged(relf4{pubv[4 * h], pubv[4 * h + 1], pubv[4 * h + 2], pubv[4 * h + 3]}));
                            if constexpr (NS == 2) rel_st_sc1_x4(dstg + SLOTS + SPL * li + 4 * h, tagged(relf4{N[4 * h], N[4 * h + 1], N[4 * h + 2], N[4 * h + 3]}));
                            if constexpr (FH2) {
                                rel_st_sc1_x4(dstg + FOFF + SPL * li + 4 * h, tagged(relf4{Fv[4 * h], Fv[4 * h + 1], Fv[4 * h + 2], Fv[4 * h + 3]}));
                                rel_st_sc1_x4(dstg + BOFF + SPL * li + 4 * h, tagged(relf4{Bv[4 * h], Bv[4 * h + 1], Bv[4 * h + 2], Bv[4 * h + 3]}));
                            }
                        }
                        const relf4 hd = {pubm, __builtin_bit_cast(float, bp + REL_BIAS), __builtin_bit_cast(float, rh), 0.0f};
                        if constexpr (GUARD) {
                            if (li < 2 * GP + 1)
                                rel_st_sc1_x4(dstg + (li < GP ? 4 * li : (li < 2 * GP ? GO + SLOTS + 4 * (li - GP) : HOFF)), tagged(li == 2 * GP ? hd : far4));
                        } else {
                            if (li == 0) rel_st_sc1_x4(dstg + HOFF, tagged(hd));
                        }
                    }
                }
            }
        }
        // everybody's slab for this step is in LDS before anyone reads it
        const unsigned long long c2 = MGM_REL_PHASES ? clock64() : 0;
        lds_barrier();
        if constexpr (MGM_REL_PHASES != 0) {
            const unsigned long long c3 = clock64();
            cph[0] += c1 - c0, cph[1] += c2 - c1, cph[2] += c3 - c2;
        }
    }
    if constexpr (MGM_REL_PHASES != 0)
        if (P.tl && lane == 0) {
            unsigned long long *w = P.tl + (long long)gridDim.x * 8 + (long long)*s_task * 16 + r * 3;
            w[0] = cph[0], w[1] = cph[1], w[2] = cph[2];
            if (r == 0) w[15] = nsweeps;
        }
}

template <bool FH, bool PUBE, int NK, int SPL, int CB, bool FH2 = false>
static hipError_t launch_rel_one(const RelParams &p, int ntasks, int wg_per_cu, hipStream_t s)
{
    constexpr int NS = (FH || PUBE) ? 1 : 2;
    constexpr int SLOTS = 16 * SPL;
    constexpr int HS = (FH2 ? 3 * SLOTS + 2 * SPL : (NS == 1 ? SLOTS + 2 * SPL : NS * SLOTS)) + 4;
    size_t shmem = sizeof(float) * ((size_t)RR * RD4 * HS + (size_t)(p.diag_any ? 2 : 1) * SD * HS + 2 * SD * RR * 4) + (size_t)SD * RR * SLOTS * CB + sizeof(unsigned) * (SD + 4) + 16 + sizeof(int) * RR * 24;
    // Occupancy through the LDS request, as for the second build: wg_per_cu workgroups (of 4 compute waves: one per SIMD) share a
    // CU -- one for a launch bound by its chains of bands, more for a batch (throughput)
    if (wg_per_cu >= 1) {
        const size_t want = (size_t)(160 * 1024) / (size_t)(wg_per_cu + 1) + 1024;  // more than a (wg_per_cu + 1)-th of the LDS
        if (shmem < want) shmem = want;
    }
    auto kern = k_pass_rel<FH, PUBE, NK, SPL, CB, FH2>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(ntasks), dim3((NW + 1) * 64), shmem, s, p);
    return hipGetLastError();
}
template <int SPL, int CB>
static hipError_t launch_rel_fmt(const RelParams &p, int ntasks, bool fh, bool pube, int wg_per_cu, hipStream_t s)
{
    if (fh) {
        if constexpr (SPL == 4 && CB == 1)  // (the one-after-the-other build is kept for the form round 5 measured, as an A/B switch)
            if (!p.fh_multi) return launch_rel_one<true, false, 0, SPL, CB>(p, ntasks, wg_per_cu, s);
        if (p.fh2) {  // update_cost2_trunclinear (TSGM = 2, no weights)
            if constexpr (SPL == 8 && CB == 4) return hipErrorInvalidValue;  // (its rings + the cost pieces exceed the LDS: the host keeps this one on the dense hull)
            else return launch_rel_one<true, false, 2, SPL, CB, true>(p, ntasks, wg_per_cu, s);
        }
        switch (p.MGM) {
        case 1: return launch_rel_one<true, false, 1, SPL, CB>(p, ntasks, wg_per_cu, s);
        case 2: return launch_rel_one<true, false, 2, SPL, CB>(p, ntasks, wg_per_cu, s);
        case 3: return launch_rel_one<true, false, 3, SPL, CB>(p, ntasks, wg_per_cu, s);
        default: return launch_rel_one<true, false, 4, SPL, CB>(p, ntasks, wg_per_cu, s);
        }
    }
    return pube ? launch_rel_one<false, true, 0, SPL, CB>(p, ntasks, wg_per_cu, s) : launch_rel_one<false, false, 0, SPL, CB>(p, ntasks, wg_per_cu, s);
}
// pube: unit weights with Hirschmueller potentials (the producer publishes E); wg_per_cu: workgroups per CU (0: what fits);
// p.slots (64 / 128) and p.cb (1 / 2): the volumes' range-proportional format
hipError_t launch_pass_rel(const RelParams &p, int ntasks, bool fh, bool pube, int wg_per_cu, hipStream_t s)
{
    if (p.slots == 128)
        return p.cb == 4 ? launch_rel_fmt<8, 4>(p, ntasks, fh, pube, wg_per_cu, s)
                         : (p.cb == 2 ? launch_rel_fmt<8, 2>(p, ntasks, fh, pube, wg_per_cu, s) : launch_rel_fmt<8, 1>(p, ntasks, fh, pube, wg_per_cu, s));
    return p.cb == 4 ? launch_rel_fmt<4, 4>(p, ntasks, fh, pube, wg_per_cu, s)
                     : (p.cb == 2 ? launch_rel_fmt<4, 2>(p, ntasks, fh, pube, wg_per_cu, s) : launch_rel_fmt<4, 1>(p, ntasks, fh, pube, wg_per_cu, s));
}
int pass_rel_lines() { return RR; }
int pass_rel_phases() { return MGM_REL_PHASES ? 16 : 0; }
// (one slab: a lane's worth of guard words + the values + another guard + the header's 4; fh2: + the forward and backward passes)
// LDS bytes of a workgroup (launch_rel_one's request): with anti-diagonal passes a second hand ring -- the host asks before it plans them
size_t pass_rel_lds_bytes(bool one_slab, int slots, int cb, bool fh2, bool diag)
{
    const size_t HS = (size_t)pass_rel_hand_floats(one_slab, slots, fh2);
    return sizeof(float) * ((size_t)RR * RD4 * HS + (size_t)(diag ? 2 : 1) * SD * HS + 2 * SD * RR * 4) + (size_t)SD * RR * slots * cb + sizeof(unsigned) * (SD + 4) + 16 + sizeof(int) * RR * 24;
}
int pass_rel_hand_floats(bool one_slab, int slots, bool fh2) { return (fh2 ? 3 * slots + 2 * (slots / 16) : (one_slab ? slots + 2 * (slots / 16) : 2 * slots)) + 4; }

}  // namespace mgm
