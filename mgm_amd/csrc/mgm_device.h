// mgm_device.h -- shared host/device declarations of libmgm_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mgm {

constexpr int kMaxDirs = 8;
constexpr int kMaxLPL = 32;          // disparities per lane of the fast kernels -> L <= 2048 (the fastest stop at 8: 512 labels)
constexpr int kMaxLabels = 1 << 22;  // (index arithmetic; the reference's Dvec has no limit, dvec.cc:60) beyond 2048 labels: the generic
                                     // kernels (mgm_pass_exact.hip, k_wta_any); FH beyond 8192: convolution arrays in global scratch
constexpr int kWave = 64;            // CDNA wavefront
constexpr int kCensusMaxWords = 8;   // 32-bit census words per pixel

// Geometry of one pass in canonical coordinates (i = position along the scan
// line, j = line index).  Derived on the host from the reference's pass table
// (mgm_core.cc:463-471): pixel(i,j) = base + i*istep + j*jstep.  In these
// coordinates every pass has the same four neighbours
//   inline (i-1,j)   same (i,j-1)   back (i-1,j-1)   fwd (i+1,j-1)
// and only the ORDER in which they are summed differs:
//   form 0 (passes 0-3): inline, same, back, fwd
//   form 1 (passes 4-7): fwd, back, same, inline
struct PassGeom {
    int NL, LL;        // number of lines, pixels per line
    int form;          // 0 / 1
    int nbands;        // ceil(NL / R)
    int slope;         // pixels of lead a line keeps over the next one: 2, or 1 when no fwd neighbour is used
    long long base;    // pixel index of (0,0)
    long long istep;   // pixel-index step along the line
    long long jstep;   // pixel-index step between lines
    int wplane[4];     // weight plane of neighbour k (mgm_core.cc:481-484)
    int nstrips, split;   // 2: the lines of this pass are walked as two strips [0, split) and [split, LL) by two workgroups per band
                          // (k_pass2, TAGS, form 1 with 2 or 3 neighbours: no in-line dependency), both from the image edge inwards
    long long hand_base;  // self-validating hand-off slabs (k_pass2, TAGS): first slab of this pass within a volume's region
    int swap;             // k_pass_rel only (round 6): a form-0 pass with 2 or 3 neighbours walked with the roles of i and j exchanged (NL, LL, istep,
                          // jstep are the exchanged ones): the in-line neighbour and the one on the line before swap places, the third stays
    int diag, wmax;       // k_pass_rel only (round 6): 1 = the pass is walked along the ANTI-DIAGONALS of (i, j), all lines of a band at the
                          // same step (form 1 with 2 or 3 neighbours: every neighbour sits on the line before); wmax: hand-off slots per line
};

// One launch of the pass kernel may aggregate several cost volumes of identical geometry (the
// left->right and right->left volumes of a stereo pair, consecutive pairs): work items are then (volume, pass,
// band), and the long dependency chains of one volume's column passes are hidden behind the other
// volumes' work.
constexpr int kMaxBatch = 16;
struct PassVolume {
    const float *C;     // [npix][L]
    const uint8_t *C8;  // [npix][L] compact costs, PassParams::cbytes bytes each (integers, all-ones = +INF: c8_encode /
                        // c16_encode) or nullptr (all volumes alike)
    float *Lr;          // NDIR volumes, pass p at Lr + (p - pass0)*nvol
    const float *w8;    // 8 planes [npix] or nullptr (all volumes alike)
    const float *rlo, *rhi;  // ragged volume: per-pixel range images (only the weighted FH kernels read them), or nullptr
    // two-valued weights (k_pass2, W2): one selector word per pixel, bit p = "the weight of plane p is not 1" (k_wsel), and
    // the penalties scaled by the other value a: P1*a, P2*a (P2*a = +INF where the cap cannot bind, like PassParams::P2)
    const unsigned *wsel;
    float p1a, p2a;
};
struct PassParams {
    PassVolume vol[kMaxBatch];
    float *hand;        // hand-off slabs  [volume*8 + pass][2][LLmax][NS*LP]; the kernels with self-validating slabs (k_pass2,
                        // TAGS): [volume][pass: g.hand_base][band][LL][LP], every slot written once per launch
    long long hand_vstride;   // ... slabs per volume (group)
    unsigned hand_tag[kMaxDirs];  // ... per pass, the tag of this launch: sign bit every handed-over word carries (0 or 0x80000000).
                              // A pass's slots are written once per launch OF THAT PASS, so the tag alternates per pass
                              // (direction-sharded callers launch the passes of one volume one at a time)
    float *handm;       // hand-off minima [volume*8 + pass][2][LLmax]
    unsigned *prog;     // progress words  [volume*8 + pass][maxbands]
    unsigned *ticket;   // work-item ticket counter
    unsigned *err;      // watchdog word
    const int2 *tasks;  // ticket -> (volume*8 + pass, band + (strip << 16) + (plain hand-off << 24)); tasks[-8 .. -1]: the
                        // XCD queues (first ticket, count) of an xcdq launch
    unsigned *qticket;  // xcdq: the eight queues' ticket counters + the count of workgroups that have left
    int xcdq;                 // 1: per-XCD work queues (k_pass2, XCDQ)
    int subv;                 // volumes per wave (1; 2 at 128 labels, 4 at 64: k_pass2<..., SUBV>); work items then address groups of volumes
    int wg_per_cu;            // 1 or 2 workgroups per compute unit (second build; see launch2_c8)
    int deep;                 // 1: the build with deeper DMA rings (k_pass2, DEEP; compact unweighted kernels)
    int oneb;                 // 1: the queue kernels built for one band per CU (k_pass2, ONEB: no 64-VGPR cap)
    int cbytes;               // bytes per compact cost of PassVolume::C8 (1 or 2)
    int xflags;               // development experiments (MGM_HIP_XFLAGS): 1 skip Lr stores, 2 skip C DMA, 4 ignore
                              // inter-band waits, 8 skip step barriers, 16 Lr stores into a cache-resident window (-DMGM_P2_XFLAG16 builds only); all of them need a -DMGM_P2_DEV=1 build
    unsigned long long *dbg;  // nullptr, or 8 words per ticket of timing diagnostics (MGM_HIP_DEBUG_STATS)
    unsigned long long tl_addr;  // device address of 8 words per ticket: the queue kernels' timeline (-DMGM_P2_TIMELINE=1 builds; MGM_HIP_TIMELINE)
    int tl_on;                   // ... 1: record it
    long long npix, nvol;
    int L, MGM, NDIR, dmin;
    int fh2_ragged;     // 1: FH, TSGM = 2, no weights, ragged volume: update_cost2_trunclinear with its boundary fix-up
    int Lreal;          // labels that exist (<= L): a label count the second build does not take runs padded to L with +INF costs
    int pass0;          // pass p writes its Lr volume to slot p - pass0
    int LLmax, maxbands;
    float P1, P2;
    PassGeom g[kMaxDirs];
};

struct WtaParams {
    const float *C;
    const uint8_t *C8;  // compact copy of C (see PassParams) or nullptr
    int cbytes;         // bytes per compact cost (1 or 2)
    const float *Lr;    // NDIR volumes
    float *S;           // nullptr or corrected volume out
    float *out, *outcost;
    long long npix, nvol;
    int L, NDIR, FIX, dmin, refine;  // refine: 0 none, 1 vfit
    int Lreal;                       // labels that exist (<= L, the stride of C / C8 / Lr); S is written with stride Lreal
    // mgm() called with range images narrower or wider than the volume's own range (mgm.cc:377-388, every TSGM_ITER
    // iteration after the first): the winner is sought among the disparities [(int)wlo, (int)whi] of each pixel; a
    // disparity of that window outside the volume holds S = 0 - (NDIR-1)*INF (0 without the over-count fix).
    const float *wlo, *whi;          // nullptr: the whole range
    // ragged C (CostParams::rlo/rhi): a disparity outside the pixel's own range does not exist in C either
    const float *clo, *chi;
    int num_cu;                      // compute units of the device (grid sizing)
};

// ---- ragged volumes in the range-proportional layout (mgm_pass_rel.hip, k_wta_rel): 64 label slots per pixel placed at the
// pixel's own window, slot k <-> disparity base + k
struct RelVolume {
    const uint8_t *c8;       // [npix][slots] cost codes of cb bytes (all ones = +INF: the slot is not a disparity of the pixel, or its cost is +INF)
    const int *base;         // [npix][4] the pixel's record: disparity of slot 0 (= its lowest disparity - 1), lowest, highest, 0
    const float *rlo, *rhi;  // [npix] the pixel's own range (the volume's range images)
    float *Lr;               // NDIR volumes [npix][slots], pass p at Lr + (p - pass0)*nvol
    const float *w8;         // 8 planes [npix] or nullptr
};
struct RelParams {
    RelVolume vol[kMaxBatch];
    float *hand;        // self-validating hand-off slots [volume][pass: g.hand_base][band][LL][NS*64 + 4]: slab(s), minimum, base + bias, padding;
                        // every word of a slot carries the launch's tag in its sign bit (mgm_pass_rel.hip)
    long long hand_vstride;  // slots per volume
    unsigned tag;       // 0 or 0x80000000
    unsigned *ticket, *err;
    const int2 *tasks;  // ticket -> (volume*8 + pass, band)
    long long npix, nvol;
    int MGM, NDIR, pass0, LLmax, maxbands, weighted;
    int ld;             // steps of LDS-DMA the loader keeps in flight (2..5; every step of lead is a step of lag per band)
    int slots, cb;      // the volumes' range-proportional format: 64 or 128 label slots per pixel, 1 or 2 bytes per cost (round 6)
    int fh2;            // FH potentials, TSGM = 2, no weights: update_cost2_trunclinear with its boundary fix-up (k_pass_rel, FH2)
    int cost2;          // TSGM = 2 without weights, Hirschmueller: update_cost2's association (every term halved before the sum)
    int fh_multi;       // FH: the pixel's TSGM min-convolutions side by side (k_pass_rel<true, false, TSGM>) instead of one after the other
    int diag_any;       // some pass of the launch walks anti-diagonals (g.diag): the workgroups carry a second hand ring
    float P1, P2;
    unsigned long long *tl;  // nullptr, or 8 words per work item (MGM_HIP_TIMELINE; tools/timeline.py): start, end, waited, slow paths, where, steps, polls
    PassGeom g[kMaxDirs];
};
hipError_t launch_rel_gather(const float *C, const float *rlo, const float *rhi, long long npix, int L, int dmin, int slots, int cb, uint8_t *rel8, int *relb,
                             unsigned *flag, hipStream_t s);
hipError_t launch_cost_census_rel(const uint32_t *cu, const uint32_t *cv, int nx, int ny, int vnx, int vny, int dmin, int L, float trunc, const float *rlo,
                                  const float *rhi, int slots, uint8_t *rel8, int *relb, unsigned *flag, hipStream_t s);
hipError_t launch_rel_expand(const uint8_t *rel8, const int *relb, long long npix, int L, int dmin, int slots, int cb, float *C, hipStream_t s);
hipError_t launch_pass_rel(const RelParams &p, int ntasks, bool fh, bool pube, int wg_per_cu, hipStream_t s);
int pass_rel_lines();
int pass_rel_phases();  // words per work item of phase clocks behind the timeline words (0: not a -DMGM_REL_PHASES=1 build)
int pass_rel_hand_floats(bool one_slab, int slots, bool fh2);
size_t pass_rel_lds_bytes(bool one_slab, int slots, int cb, bool fh2, bool diag);
struct WtaRelParams {
    const uint8_t *c8;
    const int *base;
    const float *rlo, *rhi;
    const float *Lr;
    const float *wlo, *whi;  // nullptr: the window is the pixel's own range
    float *out, *outcost;
    long long npix, nvol;
    int NDIR, FIX, refine;   // refine: index into the reference's table (0 none, 1 vfit, 2 parabola, 3 cubic, 4 parabolaOCV)
    int num_cu;
    int slots, cb;           // the volume's range-proportional format
};
hipError_t launch_wta_rel(const WtaRelParams &p, hipStream_t s);
hipError_t launch_rel_S(const WtaRelParams &p, int L, int dmin, float *S, hipStream_t s);  // the corrected S on the dense hull from the range-proportional Lr volumes

// the slow, operand-order-faithful pass kernel (mgm_pass_exact.hip): one pass of one volume
struct ExactPass {       // one pass of the reference's table (mgm_core.cc:463-471, 481-484)
    float *Lr;           // this pass's volume [npix][L]: written whole by the launch (a pixel that is not updated keeps C, 495-498)
    int d[4][2];         // neighbour offsets
    int wplane[4];       // weight plane of neighbour k
    int inc_x, inc_y, row_major;
    int jj0, njj;        // this launch: first jj of the diagonal that lies inside the image, and how many
};
struct ExactParams {
    const float *C;      // [npix][L]
    float *mins;         // [npass][npix] slab minima of the pixels visited so far
    float *fhscratch;    // FH potentials with more than 8192 labels: npass x max(nx, ny) x 4 x L floats (the convolution arrays of a
                         // diagonal's pixels, which fit the LDS up to 8192 labels), else unused
    const float *w8;     // 8 planes or nullptr
    const float *rlo, *rhi;  // ragged volume: per-pixel range images, or nullptr
    int nx, ny, L, dmin;
    float P1, P2;
    int MGM, mode;       // mode: 0 update_cost2, 1 update_costW, 2 update_cost2_trunclinear, 3 update_costW_trunclinear
    int npass;           // passes of this launch: blockIdx.y (round 6: the passes are independent -- one launch per DIAGONAL serves them all)
    int ii;              // diagonal
    ExactPass pass[kMaxDirs];
};
hipError_t launch_pass_exact(const ExactParams &p, hipStream_t s);

// launchers (one per translation unit)
hipError_t launch_pass(const PassParams &p, int ntasks, int R, bool fh, int wmode, hipStream_t s);
int pass_ns(bool fh, bool weighted);  // slabs per hand-off slot
int pass_lpl(int L);                  // disparities per lane the pass kernel is instantiated for
// second build (LDS-DMA loader waves); pass2_lines(L) = lines per band, 0 if L is not supported by it
int pass2_lines(int L, bool c8);
bool pass2_devtools();  // built with -DMGM_P2_DEV=1 (MGM_HIP_DEBUG_STATS / MGM_HIP_XFLAGS are honoured)
bool pass2_timeline();  // built with -DMGM_P2_TIMELINE=1 (MGM_HIP_TIMELINE is honoured)
hipError_t launch_pass2(const PassParams &p, int ntasks, bool fh, int wmode, hipStream_t s);
template <int LPL>
hipError_t launch_pass2_lpl(const PassParams &p, int ntasks, bool fh, int wmode, hipStream_t s);
// compact-cost support: labels per lane for which the C8 forms of K3 / k_wta exist
inline bool c8_supported(int L) { return L == 64 || L == 128 || L == 192 || L == 256 || L == 384 || L == 512 || L == 768 || L == 1024; }
hipError_t launch_compact(const float *C, long long n, uint8_t *C8, int cbytes, unsigned *bad8, hipStream_t s);
hipError_t launch_nanscan(const float *C, long long n, unsigned *flag, hipStream_t s);
hipError_t launch_pad(const float *C, long long npix, int L, int LP, float *Cp, uint8_t *C8p, int cbytes, unsigned *bad8, hipStream_t s);
hipError_t launch_expand(const uint8_t *C8, int cbytes, long long n, float *C, hipStream_t s);
hipError_t launch_expand_padded(const uint8_t *C8, int cbytes, long long npix, int L, int LP, float *C, hipStream_t s);
hipError_t launch_wta(const WtaParams &p, hipStream_t s);
long long tune_num(const char *key, long long dflt);  // development switches (MGM_HIP_TUNE; mgm_ctx.hip)
hipError_t launch_median(const float *u, int nx, int ny, int nch, int radius, float *out, hipStream_t s);
hipError_t launch_leftright(const float *dx, int nc, int nr, const float *Rdx, int Rnc, float threshold, float *out,
                            hipStream_t s);
hipError_t launch_backproject(const float *u, int nx, int ny, int nch, const float *v, int vnx, int vny, const float *disp,
                              float *out, hipStream_t s);
hipError_t launch_refine(const float *S, long long npix, int L, int dmin, int method, const float *wlo, const float *whi,
                         float vout, float *out, float *outcost, hipStream_t s);
hipError_t launch_update_ranges(const float *outoff, int nx, int ny, int slack, int radius, float *dminI, float *dmaxI,
                                float *scratch2, hipStream_t s);
hipError_t launch_census(const float *u, int nx, int ny, int nch, int winradius, uint32_t *out, hipStream_t s);
struct CostParams {
    const float *u, *v;          // planar images (float, or census words reinterpreted)
    const uint32_t *cu, *cv;     // census words (planar) when prefiltered
    float *C;
    uint8_t *C8;                 // compact copy, written alongside C
    int cbytes;                  // its bytes per cost (1 or 2)
    unsigned *bad8;              // set to 1 if some cost is not representable in the compact form
    int nx, ny, vnx, vny, nch;   // nch = channels of the (prefiltered) images
    int dmin, L;
    int Lreal;                   // k_cost_diffx / k_cost_census8*: the label slots Lreal..L-1 of the (padded) layout get +INF; else = L
    int costfn;                  // 0 ad, 1 sd, 2 census, 3 ncc, 4 btad, 5 btsd
    int hwin;                    // ncc: half window (CENSUS_NCC_WIN / 2)
    float trunc;                 // truncDist * nch
    // ragged volume (per-pixel ranges from range images, mgm_costvolume.h:276-299, 323): pixel p only has the
    // disparities [(int)rlo(p), (int)rhi(p)]; the others are +INF in the dense layout and exempt from the
    // "no finite cost => zeros" rule.  nullptr: every pixel has the whole range.
    const float *rlo, *rhi;
    // ncc: scratch for the per-pixel window statistics of the two images (k_ncc_stats): (2*nch + 1) planes of nx*ny /
    // vnx*vny floats each -- mean and variance term per channel, then the "window inside the image and NaN-free" flag --
    // or nullptr (the general kernel recomputes every window for every label).  Birchfield-Tomasi costs: the two ends of the
    // interval every sample spans (k_bt_spans), 2*nch planes per image.
    float *ncc_u, *ncc_v;
};
hipError_t launch_cost(const CostParams &p, hipStream_t s);
hipError_t launch_cost_fast(const CostParams &p, hipStream_t s, bool *taken);  // (mgm_cost_fast.hip; called by launch_cost)
hipError_t launch_filter2d(const float *u, int nx, int ny, int nch, const float *taps, int fnx, int fny, float *out,
                           hipStream_t s);
hipError_t launch_weights(const float *u, int nx, int ny, int nch, float aP, float aThresh, float *w8,
                          hipStream_t s);
hipError_t launch_selftest_div3(unsigned long long *nbad, hipStream_t s);
hipError_t launch_any_not_one(const float *w, long long n, unsigned *flag, hipStream_t s);
// what values do the weights take?  out[0] = 1 if some weight is not 1.0; out[1], out[2] = smallest and largest bit pattern
// among the weights that are not 1.0 and are positive finite floats; out[3] = 1 if some weight is neither (negative, 0, NaN, INF)
hipError_t launch_weight_values(const float *w, long long n, unsigned *out4, hipStream_t s);
// selector words of two-valued weights: sel[p] bit k = (w[k*npix + p] != 1.0f), k = 0..7
hipError_t launch_wsel(const float *w8, long long npix, unsigned *sel, hipStream_t s);
// ... and which of the two transforms of a pixel anybody reads: out[q] = sel[q] | need << 8 (k_wneed; d[pass][k] = neighbour offsets
// of the reference's pass table, plane[pass][k] = their weight planes)
hipError_t launch_wneed(const unsigned *sel, int nx, int ny, int MGM, const int (*d)[4][2], const int (*plane)[4], unsigned *out, hipStream_t s);
hipError_t launch_probe_streams(float *base, long long stride, int nstreams, long long floats_per_stream, hipStream_t s);
hipError_t launch_check_tags(const float *slabs, long long nwords, unsigned tag, unsigned *count, hipStream_t s);
hipError_t launch_xcc_census(unsigned *mask, hipStream_t s);

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ float f_inf() { return __builtin_huge_valf(); }

// Compact cost: an fp32 cost that is an integer in [0, 254] or +INF is stored as one byte
// (255 = +INF).  Census costs with one descriptor word and AD costs of 8-bit images qualify; the
// conversion back is exact.  Returns 256 if x is not representable.
__device__ __forceinline__ unsigned c8_encode(float x)
{
    if (x == __builtin_huge_valf()) return 255u;
    const float r = __builtin_rintf(x);
    const bool neg0 = __builtin_bit_cast(unsigned, x) == 0x80000000u;  // -0 would come back as +0
    return (x >= 0.0f && x <= 254.0f && r == x && !neg0) ? (unsigned)r : 256u;
}
__device__ __forceinline__ float c8_decode(unsigned b) { return b == 255u ? __builtin_huge_valf() : (float)b; }
// The same with TWO bytes per cost (round 4): integers in [0, 65534] or +INF (65535) -- absolute differences summed over
// the channels of a colour pair (up to 765), squared differences of one channel (up to 65025).  Returns 65536 if x is not
// representable.  cb = bytes per compact cost (1 or 2) picks between the two forms.
__device__ __forceinline__ unsigned c16_encode(float x)
{
    if (x == __builtin_huge_valf()) return 65535u;
    const float r = __builtin_rintf(x);
    const bool neg0 = __builtin_bit_cast(unsigned, x) == 0x80000000u;
    return (x >= 0.0f && x <= 65534.0f && r == x && !neg0) ? (unsigned)r : 65536u;
}
__device__ __forceinline__ float c16_decode(unsigned h) { return h == 65535u ? __builtin_huge_valf() : (float)h; }

// DPP lane shifts over the whole wave (gfx9 wave_shr / wave_shl).
// shr1: lane l receives lane l-1 (lane 0 gets `fill`).
__device__ __forceinline__ float dpp_shr1(float v, float fill)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill),
                                                                 __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
// shl1: lane l receives lane l+1 (lane 63 gets `fill`).
__device__ __forceinline__ float dpp_shl1(float v, float fill)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill),
                                                                 __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
// generic DPP move: lanes without a source (row edge / masked row) receive `fill`
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_mov(float v, float fill)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill),
                                                                 __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, false));
}
__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// Fused DPP arithmetic, written as inline assembly: the compiler keeps a DPP move, its fill value and
// the arithmetic apart (four issue slots where one or two do), and with four waves per SIMD every
// VALU slot of the scan-line kernels costs 16 cycles of the SIMD.  The s_nop ahead of each DPP
// instruction covers the "VALU writes a VGPR, DPP reads it" hazard (2 wait states), which the
// compiler's hazard recogniser does not see through inline assembly.
//   MGM_DPP_MIN(name, ctrl):  c = min(c, t[source lane]);  lanes without a source lane keep c
//   MGM_DPP_ADD(name, ctrl):  t = c[source lane] + addend; lanes without a source lane read 0,
//                             so `addend` must be +INF there
#define MGM_DPP_MIN(name, ctrl)                                                                            \
    __device__ __forceinline__ float name(float c, float t)                                                \
    {                                                                                                      \
        asm("s_nop 1\n\tv_min_f32_dpp %0, %1, %0 " ctrl " bank_mask:0xf" : "+v"(c) : "v"(t));              \
        return c;                                                                                          \
    }
#define MGM_DPP_ADD(name, ctrl)                                                                            \
    __device__ __forceinline__ float name(float c, float addend)                                           \
    {                                                                                                      \
        float t;                                                                                           \
        asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 " ctrl " bank_mask:0xf bound_ctrl:0" : "=v"(t) : "v"(c), "v"(addend)); \
        return t;                                                                                          \
    }
MGM_DPP_MIN(dpp_min_row_shr1, "row_shr:1 row_mask:0xf")
MGM_DPP_MIN(dpp_min_row_shr2, "row_shr:2 row_mask:0xf")
MGM_DPP_MIN(dpp_min_row_shr4, "row_shr:4 row_mask:0xf")
MGM_DPP_MIN(dpp_min_row_shr8, "row_shr:8 row_mask:0xf")
MGM_DPP_MIN(dpp_min_row_shl1, "row_shl:1 row_mask:0xf")
MGM_DPP_MIN(dpp_min_row_shl2, "row_shl:2 row_mask:0xf")
MGM_DPP_MIN(dpp_min_row_shl4, "row_shl:4 row_mask:0xf")
MGM_DPP_MIN(dpp_min_row_shl8, "row_shl:8 row_mask:0xf")
MGM_DPP_MIN(dpp_min_bcast15, "row_bcast:15 row_mask:0xa")
MGM_DPP_MIN(dpp_min_bcast31, "row_bcast:31 row_mask:0xc")
MGM_DPP_ADD(dpp_add_bcast15, "row_bcast:15 row_mask:0xf")
MGM_DPP_ADD(dpp_add_bcast31, "row_bcast:31 row_mask:0xf")
MGM_DPP_ADD(dpp_add_wave_shr1, "wave_shr:1 row_mask:0xf")
MGM_DPP_ADD(dpp_add_wave_shl1, "wave_shl:1 row_mask:0xf")
#undef MGM_DPP_MIN
#undef MGM_DPP_ADD
// The same on NK independent values in ONE block (round 6, fh_scan_multi): one s_nop for the block, the NK DPP instructions back to
// back -- the chains of NK side-by-side scans interleave at every stage instead of running one after the other (the compiler keeps
// inline-assembly blocks in source order).
#define MGM_DPP_MIN_N(name, ctrl)                                                                                                     \
    template <int NK>                                                                                                                 \
    __device__ __forceinline__ void name(float (&c)[NK], const float (&t)[NK])                                                        \
    {                                                                                                                                 \
        static_assert(NK >= 1 && NK <= 4, "one to four chains");                                                                      \
        if constexpr (NK == 1)                                                                                                        \
            asm("s_nop 1\n\tv_min_f32_dpp %0, %1, %0 " ctrl " bank_mask:0xf" : "+v"(c[0]) : "v"(t[0]));                               \
        else if constexpr (NK == 2)                                                                                                   \
            asm("s_nop 1\n\tv_min_f32_dpp %0, %2, %0 " ctrl " bank_mask:0xf\n\tv_min_f32_dpp %1, %3, %1 " ctrl " bank_mask:0xf"      \
                : "+v"(c[0]), "+v"(c[1])                                                                                              \
                : "v"(t[0]), "v"(t[1]));                                                                                              \
        else if constexpr (NK == 3)                                                                                                   \
            asm("s_nop 1\n\tv_min_f32_dpp %0, %3, %0 " ctrl " bank_mask:0xf\n\tv_min_f32_dpp %1, %4, %1 " ctrl                       \
                " bank_mask:0xf\n\tv_min_f32_dpp %2, %5, %2 " ctrl " bank_mask:0xf"                                                   \
                : "+v"(c[0]), "+v"(c[1]), "+v"(c[2])                                                                                  \
                : "v"(t[0]), "v"(t[1]), "v"(t[2]));                                                                                   \
        else                                                                                                                          \
            asm("s_nop 1\n\tv_min_f32_dpp %0, %4, %0 " ctrl " bank_mask:0xf\n\tv_min_f32_dpp %1, %5, %1 " ctrl                       \
                " bank_mask:0xf\n\tv_min_f32_dpp %2, %6, %2 " ctrl " bank_mask:0xf\n\tv_min_f32_dpp %3, %7, %3 " ctrl                \
                " bank_mask:0xf"                                                                                                      \
                : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])                                                                      \
                : "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]));                                                                        \
    }
#define MGM_DPP_ADD_N(name, ctrl)                                                                                                     \
    template <int NK>                                                                                                                 \
    __device__ __forceinline__ void name(float (&t)[NK], const float (&c)[NK], const float (&a)[NK])                                  \
    {                                                                                                                                 \
        static_assert(NK >= 1 && NK <= 4, "one to four chains");                                                                      \
        if constexpr (NK == 1)                                                                                                        \
            asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 " ctrl " bank_mask:0xf bound_ctrl:0" : "=&v"(t[0]) : "v"(c[0]), "v"(a[0]));      \
        else if constexpr (NK == 2)                                                                                                   \
            asm("s_nop 1\n\tv_add_f32_dpp %0, %2, %4 " ctrl " bank_mask:0xf bound_ctrl:0\n\tv_add_f32_dpp %1, %3, %5 " ctrl          \
                " bank_mask:0xf bound_ctrl:0"                                                                                         \
                : "=&v"(t[0]), "=&v"(t[1])                                                                                            \
                : "v"(c[0]), "v"(c[1]), "v"(a[0]), "v"(a[1]));                                                                        \
        else if constexpr (NK == 3)                                                                                                   \
            asm("s_nop 1\n\tv_add_f32_dpp %0, %3, %6 " ctrl " bank_mask:0xf bound_ctrl:0\n\tv_add_f32_dpp %1, %4, %7 " ctrl          \
                " bank_mask:0xf bound_ctrl:0\n\tv_add_f32_dpp %2, %5, %8 " ctrl " bank_mask:0xf bound_ctrl:0"                         \
                : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2])                                                                               \
                : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(a[0]), "v"(a[1]), "v"(a[2]));                                                  \
        else                                                                                                                          \
            asm("s_nop 1\n\tv_add_f32_dpp %0, %4, %8 " ctrl " bank_mask:0xf bound_ctrl:0\n\tv_add_f32_dpp %1, %5, %9 " ctrl          \
                " bank_mask:0xf bound_ctrl:0\n\tv_add_f32_dpp %2, %6, %10 " ctrl                                                     \
                " bank_mask:0xf bound_ctrl:0\n\tv_add_f32_dpp %3, %7, %11 " ctrl " bank_mask:0xf bound_ctrl:0"                        \
                : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])                                                                  \
                : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));                            \
    }
MGM_DPP_MIN_N(dpp_min_row_shr1_n, "row_shr:1 row_mask:0xf")
MGM_DPP_MIN_N(dpp_min_row_shr2_n, "row_shr:2 row_mask:0xf")
MGM_DPP_MIN_N(dpp_min_row_shr4_n, "row_shr:4 row_mask:0xf")
MGM_DPP_MIN_N(dpp_min_row_shr8_n, "row_shr:8 row_mask:0xf")
MGM_DPP_MIN_N(dpp_min_row_shl1_n, "row_shl:1 row_mask:0xf")
MGM_DPP_MIN_N(dpp_min_row_shl2_n, "row_shl:2 row_mask:0xf")
MGM_DPP_MIN_N(dpp_min_row_shl4_n, "row_shl:4 row_mask:0xf")
MGM_DPP_MIN_N(dpp_min_row_shl8_n, "row_shl:8 row_mask:0xf")
MGM_DPP_ADD_N(dpp_add_wave_shr1_n, "wave_shr:1 row_mask:0xf")
MGM_DPP_ADD_N(dpp_add_wave_shl1_n, "wave_shl:1 row_mask:0xf")
#undef MGM_DPP_MIN_N
#undef MGM_DPP_ADD_N

// wave-wide minimum, result uniform (SGPR).  NaN-free inputs only.
__device__ __forceinline__ float wave_min(float v)
{
    v = dpp_min_row_shr1(v, v);
    v = dpp_min_row_shr2(v, v);
    v = dpp_min_row_shr4(v, v);
    v = dpp_min_row_shr8(v, v);  // lane 15 of every row: the row's minimum
    v = dpp_min_bcast15(v, v);   // rows 1 and 3 take in the row before them
    v = dpp_min_bcast31(v, v);   // rows 2 and 3 take in lane 31
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#endif

}  // namespace mgm
