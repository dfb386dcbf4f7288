// mgm_host.h -- host-side internals shared by the translation units behind the C ABI of libmgm_hip.so
// (mgm_ctx.hip: contexts, device containers, compact cost copies; mgm_plan.hip: the launch plan of the pass kernels and the
// winner search; mgm_api.hip: cost volumes, aggregation calls, the steps around them).  Nothing here is exported through
// include/mgm_hip.h; no compute happens on the host and there is no CPU fallback.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mgm_hip.h"
#include "mgm_device.h"

using namespace mgm;

struct mgm_img {
    float *d;
    int nx, ny, nch;
    int device;  // the device the pixels live on (mgm_img_device)
};
struct mgm_cv {
    float *d;
    int nx, ny, dmin, dmax;
    // compact (one byte per cost) copy used by K3 / k_wta when every cost is an integer 0..254 or +INF
    uint8_t *d8 = nullptr;
    int cbytes = 1;            // bytes per cost of the compact copy: 1 (0..254, 255 = +INF) or 2 (0..65534, 65535 = +INF; round 4:
                               // absolute differences of colour pairs, squared differences)
    size_t d8_cap = 0;         // bytes allocated at d8
    mutable int pad_hint = 1;  // a label count that runs padded: the compact form its padded copy took last time (0: none did)
    unsigned *bad8 = nullptr;  // device word: 1 = not representable
    int c8_state = 0;          // 0 none, 1 written (validity not read back yet), 2 valid, -1 invalid
    // K2 skips the fp32 write when its costs are known to fit the compact form (single-word census):
    // nothing on the hot path reads `d` then, and it is decoded from d8 if somebody asks for it.
    int f32_state = 1;         // 1 current, 0 stale (d8 holds the volume)
    int diff_fails = 0;        // AD / SD fillings of this volume in a row that did not fit the compact form: after two, refills go straight
                               // to the fp32 kernel (a filling that fits resets the count)
    bool diff_wide = false;    // ... did not fit ONE byte per cost but does fit two (a grey pair with a difference of 255): refills start there
    // A label count that the pass kernels run PADDED (151 -> 192, ...): K2 may write the padded compact copy itself --
    // [npix][p8_L] costs of p8_cb bytes, the label slots beyond the real count +INF -- instead of an fp32 volume that every
    // aggregation call pads and encodes again (run_passes).  p8_state 2: valid (and then the ONLY copy until somebody asks
    // for the fp32 volume: f32_state 0), 0: none.
    uint8_t *p8 = nullptr;
    size_t p8_cap = 0;
    int p8_L = 0, p8_cb = 1, p8_state = 0;
    mgm_ctx *owner = nullptr;
    // ragged volume: the per-pixel range images it was built from (device, nx*ny floats each), else nullptr.
    // dmin/dmax are then the hull of all ranges; labels outside a pixel's own range hold +INF.
    float *rlo = nullptr, *rhi = nullptr;
    // built by `-p census` with a non-census distance from descriptors of more than 24 bits: costs are differences of
    // descriptor WORDS read as floats (mgm_costvolume.h:355-362), NaN patterns included.  The volume itself is
    // reproduced bit for bit; what the reference's aggregation makes of NaN costs depends on operand order.
    bool nan_words = false;
    // Does the volume hold NaN costs?  The scan-line kernels are compiled NaN-free (mgm_pass_common.h) and what the
    // reference makes of a NaN cost depends on the operand order of its minima, so such a volume is refused by
    // mgm_aggregate* instead of being aggregated into something unspecified.  0 not scanned (uploaded / written through
    // mgm_cv_device_ptr), 1 flag word on the device is current but not read back, 2 clean, -1 holds NaN.
    int nan_state = 0;
    // bumped whenever the contents may have changed: contexts remember (pointer, generation) of the volumes of their
    // last aggregation, so a refilled volume, or a new one at a recycled address, is not mistaken for one of them
    unsigned long long gen = 0;
    // ragged volume, range-proportional copy (round 5; mgm_pass_rel.hip): 64 cost bytes per pixel placed at its own window +
    // a 16-byte record per pixel (disparity of slot 0, lo, hi) + a flag word, one allocation [npix*64 bytes][npix*16 bytes][flag]; rel_state 0 none, 1 written (flag
    // not read back yet), 2 usable, -1 not usable (a window wider than 62 labels, a cost that is not a byte)
    // (round 6) the copy's FORMAT: rel_slots = 64 or 128 label slots per pixel (windows of up to 62 / 126 labels), rel_cb = 1 or 2 bytes
    // per cost code; [npix * rel_slots * rel_cb bytes of costs][npix * 16 bytes of records][flag word].  A gathered copy starts in the
    // narrowest form its cost function allows and is gathered again wider if the flag word asks for it (rel_resolve).
    uint8_t *relbuf = nullptr;
    size_t rel_cap = 0;
    int rel_state = 0;
    int rel_slots = 64, rel_cb = 1;
    int rel_hint_slots = 64;  // the width the last direct filling of this volume needed: a refill starts there (a failed attempt costs ~2.5 ms of flag traffic)
    size_t rel_cost_bytes() const { return (size_t)nx * ny * (size_t)rel_slots * (size_t)rel_cb; }
    int *rel_records() const { return reinterpret_cast<int *>(relbuf + rel_cost_bytes()); }
    unsigned *rel_flag() const { return reinterpret_cast<unsigned *>(relbuf + rel_cost_bytes() + (size_t)nx * ny * 16); }
    // the relative copy is the ONLY copy (single-word census: K2 wrote it straight from the descriptors, f32_state 0);
    // ensure_f32 expands it into the dense hull on demand
    bool rel_only = false;
    // a caller-provided volume whose refill FAILED half-way holds neither its old costs nor new ones: mgm_aggregate*
    // refuses it (MGM_ERR_INVALID) until a later mgm_costvolume_build* has filled it
    bool unfilled = false;
};
unsigned long long next_cv_generation();

struct Buf {  // grow-only device scratch
    void *p = nullptr;
    size_t cap = 0;
};

struct Timing {
    const char *name;
    hipEvent_t a, b;
};

struct mgm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // workspace
    Buf exact_mins, exact_scratch;  // slab minima / (FH beyond 8192 labels) convolution arrays of the operand-order-faithful pass kernel
    Buf lr, hand, hand2, handm, words, tasks, census_u, census_v, dbg, stmp, ones8;  // hand: self-validating slabs (TAGS); hand2: the other kernels' slots
    // range-proportional aggregation of ragged volumes (mgm_pass_rel.hip): its Lr volumes [volume][pass][npix][64], hand-off slots,
    // task table (+ what it was made for), and what the last such aggregation ran on (mgm_wta_windowed_dev searches it again)
    Buf lr_rel, hand_rel, tasks_rel;
    std::string tasks_rel_key;
    int ntasks_rel = 0;
    int rel_last_batch = 0, rel_last_ndir = 0;
    long long rel_last_stride = 0;
    const mgm_cv *rel_last_cvs[kMaxBatch] = {};
    unsigned long long rel_last_gens[kMaxBatch] = {};
    // Pipelined contexts (mgm_ctx_set_pipeline, depth >= 2): aggregation calls are DEFERRED and gathered -- up to `depth`
    // calls of the same geometry and settings become ONE launch of the pass kernel (see PendingAgg, pipe_flush)
    struct PendingAgg {
        std::vector<const mgm_cv *> C;
        std::vector<const mgm_img *> w8;  // empty: unweighted
        std::vector<mgm_img *> out, outcost;
        float P1, P2;
        int NDIR, MGM, use_fh, fix_overcount;
        std::string refine;
        bool has_refine;
    };
    int pipe_depth = 1;
    std::vector<PendingAgg> pend;
    // mgm_ctx_set_placement_tries: how many physical placements of a NEW Lr workspace the context may try (0: take what the
    // allocator gives); placed_ptr / placed_cap: the allocation that has been through it
    int place_tries = 0;
    const void *placed_ptr = nullptr;
    size_t placed_cap = 0;
    size_t ws_limit = 0;  // mgm_ctx_set_workspace_limit: cap on the Lr + hand-off workspace of one pass launch (0 = none)
    int debug_stats = 0;  // MGM_HIP_DEBUG_STATS=1: per-workgroup timing summary of K3 on stderr
    unsigned *h_words = nullptr;  // pinned mirror of the control words
    // cached task table key
    int tk_nx = -1, tk_ny = -1, tk_ndir = -1, tk_r = -1;
    // task tables of earlier launch shapes (a caller that launches the passes of a volume one by one alternates between
    // eight of them): {nx, ny, key, R} -> device table; `tasks` is the one in use
    struct TaskTab {
        int nx, ny, key, R, ntasks;
        Buf buf;
        bool one_queue;  // dealt to ONE queue for all XCDs (the plan's simulation preferred it: mgm_plan.hip)
    };
    bool tk_one_queue = false;  // ... of the table in use
    std::vector<TaskTab> ttabs;
    int force_build = 0;  // 0 auto, 1 first build only (MGM_HIP_PASS_BUILD=1)
    int ntasks = 0;
    // last aggregate (for mgm_debug_download_lr)
    long long last_nvol = 0;    // floats per volume
    long long last_stride = 0;  // floats between the Lr volumes of consecutive passes (>= last_nvol)
    int last_ndir = 0;
    int last_batch = 0;
    int last_L = 0, last_Lk = 0;          // labels of the last aggregation, and the label stride its kernels ran with (>= last_L)
    Buf padf[kMaxBatch], pad8[kMaxBatch];  // padded copies of the cost volumes of a launch whose label count was padded
    Buf wsel[kMaxBatch], wvals;            // two-valued weights (k_pass2, W2): selector words per volume; the value scan's words
    bool last_pad_c8 = false;
    const uint8_t *last_pad_ptr[kMaxBatch] = {};  // ... where they are: the context's pad8 buffers, or the volumes' own padded copies (mgm_cv::p8)
    int last_pad_cb = 1;  // ... bytes per compact cost of those padded copies
    const mgm_cv *last_cvs[kMaxBatch] = {};  // the volumes of the last aggregation (identity only, never dereferenced) ...
    unsigned long long last_gens[kMaxBatch] = {};  // ... and their generations at that time
    bool pending_check = false;
    // self-validating hand-off slabs (k_pass2, TAGS): what the region was last cleared for, and the tag of its last launch
    std::string hand_key;
    std::string hand_rel_key;      // the range-proportional kernels' slots (same protocol): geometry they were last written for ...
    unsigned hand_rel_tag = 0;     // ... and the tag they carry
    unsigned hand_tags[kMaxDirs] = {};  // per pass: the tag its slots carry after its last launch
    int num_cu = 256;  // hipDeviceProp_t::multiProcessorCount
    int xcc_mask = -1;  // XCC ids the workgroups of a launch see (k_xcc_census; -1: not looked yet)
    // timing
    bool timing = false;
    std::vector<Timing> tim;
};


constexpr int kR = 16;       // lines per band (waves per workgroup) of the pass kernel
constexpr int kCtrlWords = 4 + kMaxBatch * kMaxDirs * 4096;  // ticket, err, flag, pad, prog[volume*8 + pass][maxbands]
constexpr int kMaxBands = 4096;

// Development switches (A/B timing, tests of the fall-back paths), read once per process; everything is on by default.
struct DevSwitches {
    bool c8;         // MGM_HIP_C8=0: never use the compact (1 byte per label) cost volumes
    bool lazy_f32;   // MGM_HIP_LAZY_F32=0: always materialise the fp32 volume next to the compact one
    bool pad;        // MGM_HIP_PAD=0: no padding of label counts to the next count of the second build
    int subv;        // MGM_HIP_SUBV=0: one volume per wave also at 128 / 64 labels; 2: volumes share waves whenever they can
    int deep;        // MGM_HIP_DEEP=0|1: never / always the pass kernels with deep DMA rings (default: by the launch's shape)
    int wg_per_cu;   // MGM_HIP_WG_PER_CU=1|2: override the occupancy heuristic of the pass kernel (0 = heuristic)
    int xflags;      // MGM_HIP_XFLAGS: experiment bits of development builds (mgm_device.h)
    int strips;      // MGM_HIP_STRIPS=0|1: never / always walk the lines of passes 4-7 as two strips (default: chain-bound launches only)
    int xcdq;        // MGM_HIP_XCDQ=0|1: never / whenever possible the per-XCD work queues of k_pass2 (default: chain-bound launches)
    int xcdq_k;      // MGM_HIP_XCDQ_K: consecutive bands of a pass per queue block (0: a pass stays on one XCD; default: by the launch's shape)
    bool w2;         // MGM_HIP_W2=0: two-valued weights take the general weighted kernels too (A/B)
    bool oneb;       // MGM_HIP_ONEB=0: launches that run one band per CU keep the queue kernels capped at 64 VGPRs (A/B)
    long long lr_pad;  // MGM_HIP_LR_PAD: floats between consecutive Lr volumes beyond their size, in 256-byte blocks (67)
};
const DevSwitches &dev();
long long lr_pad_floats();

// Development switches live behind ONE variable: MGM_HIP_TUNE="key=value,key=value" (keys are the lower-case names of
// DevSwitches' comments: deep, xcdq, xcdq_k, strips, wg_per_cu, subv, c8, pad, lazy_f32, w2, oneb, lr_pad, xflags, pass_build,
// debug_stats, check_tags, show_plan, wta_*); the individual MGM_HIP_<KEY> variables of rounds 1-3 are still read (tests and
// tools use them).  A build with -DMGM_HIP_RELEASE compiles the parser out: every switch then has its default.
// (declared in mgm_device.h: mgm::tune_num(key, default))

int fail(mgm_ctx *c, int code, const std::string &msg);
int hipfail(mgm_ctx *c, hipError_t e, const char *what);
#define HIPCHK(c, call)                                          \
    do {                                                         \
        hipError_t e__ = (call);                                 \
        if (e__ != hipSuccess) return hipfail((c), e__, #call);  \
    } while (0)
hipError_t dev_malloc(void **p, size_t bytes);
int reserve(mgm_ctx *c, Buf &b, size_t bytes);
int ensure_words(mgm_ctx *c);

struct TimeScope {  // brackets one kernel launch with events when timing is on
    mgm_ctx *c;
    Timing t{};
    bool on;
    TimeScope(mgm_ctx *ctx, const char *name) : c(ctx), on(ctx->timing)
    {
        if (!on) return;
        t.name = name;
        if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) {
            on = false;
            return;
        }
        (void)hipEventRecord(t.a, c->stream);
    }
    ~TimeScope()
    {
        if (!on) return;
        (void)hipEventRecord(t.b, c->stream);
        c->tim.push_back(t);
    }
};

// The reference's pass table, mgm_core.cc:463-471, as data.
struct RefPass {
    int d[4][2];
    int inc_x, inc_y, row_major;
};
static const RefPass kPasses[8] = {
    {{{-1, 0}, {0, -1}, {-1, -1}, {1, -1}}, 1, 1, 1}, {{{1, 0}, {0, 1}, {1, 1}, {-1, 1}}, 0, 0, 1},
    {{{0, 1}, {-1, 0}, {-1, 1}, {-1, -1}}, 1, 0, 0},  {{{0, -1}, {1, 0}, {1, -1}, {1, 1}}, 0, 1, 0},
    {{{-1, -1}, {1, -1}, {0, -1}, {1, 0}}, 0, 1, 1},  {{{1, -1}, {1, 1}, {1, 0}, {0, 1}}, 0, 0, 0},
    {{{1, 1}, {-1, 1}, {0, 1}, {-1, 0}}, 1, 0, 1},    {{{-1, 1}, {-1, -1}, {-1, 0}, {0, -1}}, 1, 1, 0},
};
static const int kPassToChannel[4][8] = {  // mgm_core.cc:481-484
    {0, 1, 2, 3, 4, 5, 6, 7}, {3, 2, 0, 1, 5, 6, 7, 4}, {4, 6, 7, 5, 3, 1, 2, 0}, {5, 7, 4, 6, 1, 2, 0, 3}};

int distance_index(const char *n);
int prefilter_index(const char *n);
int refinement_index(const char *n);
bool make_geom(int pass, int nx, int ny, int R, int MGM, bool slope1_ok, PassGeom &g);
int check_watchdog(mgm_ctx *c, bool block = true);

// pipelined contexts (mgm_ctx_set_pipeline): see mgm_api.hip
int pipe_flush(mgm_ctx *c);
inline int pipe_join(mgm_ctx *c) { return (c && !c->pend.empty()) ? pipe_flush(c) : MGM_OK; }
bool pipe_uses(const mgm_ctx *c, const void *obj);

// volumes and their compact copies (mgm_ctx.hip)
int cv_alloc_f32(mgm_ctx *c, mgm_cv *cv);
int cv_create(mgm_ctx *c, int nx, int ny, int dmin, int dmax, bool alloc_f32, mgm_cv **out);
int ensure_f32(mgm_ctx *c, const mgm_cv *ccv);
int c8_alloc(mgm_ctx *c, mgm_cv *cv, int cb = 1);
int c8_resolve(mgm_ctx *c, const mgm_cv *ccv, bool *use);

// the launch plan (mgm_plan.hip)
int padded_labels(int L);
int p8_alloc(mgm_ctx *c, mgm_cv *cv, int LP, int cb);
int run_passes(mgm_ctx *c, const mgm_cv *const *Cs, const mgm_img *const *w8s, int nb, float P1, float P2, int MGM, int use_fh, int first,
               int count, bool allow_pad = false, int slot0 = 0, int nslots = 0, int layout_ndir = 0);
int run_wta(mgm_ctx *c, const mgm_cv *C, long long pix0, long long npix, const float *lr, long long lr_stride, int NDIR, int fix_overcount,
            int ridx, float *out, float *outcost, float *Sout, const float *wlo = nullptr, const float *whi = nullptr, int slot = -1);
// the range-proportional path of ragged volumes (mgm_plan.hip): is this call one it takes?  then the passes + the winner search
bool rel_enabled();
int rel_resolve(mgm_ctx *c, const mgm_cv *cv, bool *usable);
int rel_alloc(mgm_ctx *c, mgm_cv *cv, int slots, int cb);  // (re)allocates relbuf for the format and sets rel_slots / rel_cb; MGM_OK also when the device has no room (relbuf stays null)
int weights_have_odd_values(mgm_ctx *c, const mgm_img *const *w8s, int nb, long long npix, bool *odd, bool *any = nullptr);
int run_rel(mgm_ctx *c, const mgm_cv *const *Cs, const mgm_img *const *w8s, int nb, float P1, float P2, int MGM, int use_fh, int NDIR,
            int fix_overcount, int ridx, mgm_img *const *outs, mgm_img *const *outcosts, mgm_cv **S = nullptr);
int run_wta_rel(mgm_ctx *c, const mgm_cv *C, int slot, int NDIR, int fix_overcount, int ridx, const float *wlo, const float *whi, float *out,
                float *outcost, float *Sout = nullptr);
int run_wta_refine(mgm_ctx *c, const mgm_cv *C, long long pix0, long long npix, const float *lr, long long lr_stride, int NDIR,
                   int fix_overcount, int ridx, float *out, float *outcost, float *Sout, const float *wlo = nullptr,
                   const float *whi = nullptr, int slot = -1);
