// mgm_api.hip -- the C ABI of libmgm_hip.so (include/mgm_hip.h): contexts, device
// containers, and the host-side orchestration of the kernels.  No compute
// happens on the host and there is no CPU fallback.
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
};
static unsigned long long next_cv_generation()
{
    static unsigned long long g = 0;
    return ++g;
}

namespace {

struct Buf {  // grow-only device scratch
    void *p = nullptr;
    size_t cap = 0;
};

struct Timing {
    const char *name;
    hipEvent_t a, b;
};

}  // namespace

struct mgm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // workspace
    Buf exact_mins;  // slab minima of the operand-order-faithful pass kernel (mgm_pass_exact.hip)
    Buf lr, hand, hand2, handm, words, tasks, census_u, census_v, dbg, stmp, ones8;  // hand: self-validating slabs (TAGS); hand2: the other kernels' slots
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
    };
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
    int last_pad_cb = 1;  // ... bytes per compact cost of those padded copies
    const mgm_cv *last_cvs[kMaxBatch] = {};  // the volumes of the last aggregation (identity only, never dereferenced) ...
    unsigned long long last_gens[kMaxBatch] = {};  // ... and their generations at that time
    bool pending_check = false;
    // self-validating hand-off slabs (k_pass2, TAGS): what the region was last cleared for, and the tag of its last launch
    std::string hand_key;
    unsigned hand_tags[kMaxDirs] = {};  // per pass: the tag its slots carry after its last launch
    int num_cu = 256;  // hipDeviceProp_t::multiProcessorCount
    int xcc_mask = -1;  // XCC ids the workgroups of a launch see (k_xcc_census; -1: not looked yet)
    // timing
    bool timing = false;
    std::vector<Timing> tim;
};

namespace {

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
static const DevSwitches &dev()
{
    static const DevSwitches d = [] {
        auto on = [](const char *n) { const char *e = getenv(n); return !(e && atoi(e) == 0); };
        auto num = [](const char *n, long long dflt) { const char *e = getenv(n); return e ? atoll(e) : dflt; };
        return DevSwitches{on("MGM_HIP_C8"), on("MGM_HIP_LAZY_F32"), on("MGM_HIP_PAD"), (int)num("MGM_HIP_SUBV", 1), (int)num("MGM_HIP_DEEP", -1),
                           (int)num("MGM_HIP_WG_PER_CU", 0), (int)num("MGM_HIP_XFLAGS", 0), (int)num("MGM_HIP_STRIPS", -1), (int)num("MGM_HIP_XCDQ", -1), (int)num("MGM_HIP_XCDQ_K", -1), on("MGM_HIP_W2"), on("MGM_HIP_ONEB"), 64ll * num("MGM_HIP_LR_PAD", 67)};
    }();
    return d;
}
long long lr_pad_floats() { return dev().lr_pad; }

int fail(mgm_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    return code;
}
int hipfail(mgm_ctx *c, hipError_t e, const char *what)
{
    return fail(c, MGM_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(c, call)                                          \
    do {                                                         \
        hipError_t e__ = (call);                                 \
        if (e__ != hipSuccess) return hipfail((c), e__, #call);  \
    } while (0)

// hipMalloc whose failure does not outlive the call: the runtime keeps the last error until somebody reads it, and every
// launch wrapper here ends with `return hipGetLastError()` -- without this, the first kernel launched after an
// MGM_ERR_NOMEM return (the smaller chunk mgm_aggregate_batch_dev retries with, or simply the caller's next call) would
// report the stale hipErrorOutOfMemory as its own.
hipError_t dev_malloc(void **p, size_t bytes)
{
    const hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        *p = nullptr;
        (void)hipGetLastError();
    }
    return e;
}

int reserve(mgm_ctx *c, Buf &b, size_t bytes)
{
    if (bytes <= b.cap) return MGM_OK;
    if (b.p) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    hipError_t e = dev_malloc(&b.p, bytes);
    if (e != hipSuccess) {
        return fail(c, MGM_ERR_NOMEM, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
    }
    b.cap = bytes;
    return MGM_OK;
}

// the control words of the pass kernel (and a little scratch for others): zeroed when they come into being -- word 1, the
// watchdog word, is never reset by a launch (check_watchdog)
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

int ensure_words(mgm_ctx *c)
{
    const void *before = c->words.p;
    if (int r = reserve(c, c->words, sizeof(unsigned) * (4 + (size_t)kMaxBatch * kMaxDirs * 4096))) return r;
    if (c->words.p != before) HIPCHK(c, hipMemsetAsync(c->words.p, 0, c->words.cap, c->stream));
    return MGM_OK;
}

// name tables with the reference's silent fall-back to entry 0
int distance_index(const char *n)  // mgm_costvolume.h:170-190
{
    static const char *t[] = {"ad", "sd", "census", "ncc", "btad", "btsd", nullptr};
    int r = 0;
    for (int i = 0; t[i]; i++)
        if (n && !strcmp(n, t[i])) r = i;
    return r;
}
int prefilter_index(const char *n)  // mgm_costvolume.h:194-207
{
    static const char *t[] = {"none", "census", "sobelx", "gblur", nullptr};
    int r = 0;
    for (int i = 0; t[i]; i++)
        if (n && !strcmp(n, t[i])) r = i;
    return r;
}
int refinement_index(const char *n)  // mgm_refine.h:15-35
{
    static const char *t[] = {"none", "vfit", "parabola", "cubic", "parabolaOCV", nullptr};
    int r = 0;
    for (int i = 0; t[i]; i++)
        if (n && !strcmp(n, t[i])) r = i;
    return r;
}

// The reference's pass table, mgm_core.cc:463-471, as data.
struct RefPass {
    int d[4][2];
    int inc_x, inc_y, row_major;
};
const RefPass kPasses[8] = {
    {{{-1, 0}, {0, -1}, {-1, -1}, {1, -1}}, 1, 1, 1}, {{{1, 0}, {0, 1}, {1, 1}, {-1, 1}}, 0, 0, 1},
    {{{0, 1}, {-1, 0}, {-1, 1}, {-1, -1}}, 1, 0, 0},  {{{0, -1}, {1, 0}, {1, -1}, {1, 1}}, 0, 1, 0},
    {{{-1, -1}, {1, -1}, {0, -1}, {1, 0}}, 0, 1, 1},  {{{1, -1}, {1, 1}, {1, 0}, {0, 1}}, 0, 0, 0},
    {{{1, 1}, {-1, 1}, {0, 1}, {-1, 0}}, 1, 0, 1},    {{{-1, 1}, {-1, -1}, {-1, 0}, {0, -1}}, 1, 1, 0},
};
const int kPassToChannel[4][8] = {  // mgm_core.cc:481-484
    {0, 1, 2, 3, 4, 5, 6, 7}, {3, 2, 0, 1, 5, 6, 7, 4}, {4, 6, 7, 5, 3, 1, 2, 0}, {5, 7, 4, 6, 1, 2, 0, 3}};

// Canonical geometry of a pass (see PassGeom).  Returns false if the table
// entry does not reduce to one of the two canonical neighbour orders.
bool make_geom(int pass, int nx, int ny, int R, int MGM, bool slope1_ok, PassGeom &g)
{
    const RefPass &rp = kPasses[pass];
    const long long sx = rp.inc_x ? 1 : -1, sy = rp.inc_y ? 1 : -1;
    g.base = (long long)(rp.inc_y ? 0 : ny - 1) * nx + (rp.inc_x ? 0 : nx - 1);
    if (rp.row_major) {
        g.NL = ny;
        g.LL = nx;
        g.istep = sx;
        g.jstep = sy * nx;
    } else {
        g.NL = nx;
        g.LL = ny;
        g.istep = sy * nx;
        g.jstep = sx;
    }
    int kind[4];
    for (int k = 0; k < 4; k++) {
        const int dx = rp.d[k][0], dy = rp.d[k][1];
        const int di = rp.row_major ? dx * (int)sx : dy * (int)sy;
        const int dj = rp.row_major ? dy * (int)sy : dx * (int)sx;
        if (di == -1 && dj == 0) kind[k] = 0;        // inline
        else if (di == 0 && dj == -1) kind[k] = 1;   // same
        else if (di == -1 && dj == -1) kind[k] = 2;  // back
        else if (di == 1 && dj == -1) kind[k] = 3;   // fwd
        else return false;
        g.wplane[k] = kPassToChannel[k][pass];
    }
    if (kind[0] == 0 && kind[1] == 1 && kind[2] == 2 && kind[3] == 3) g.form = 0;
    else if (kind[0] == 3 && kind[1] == 2 && kind[2] == 1 && kind[3] == 0) g.form = 1;
    else return false;
    g.nbands = (g.NL + R - 1) / R;
    // form 0 sums inline, same, back, fwd: with MGM <= 3 the fwd neighbour (i+1, j-1) is never read,
    // so a line only has to stay ONE pixel behind the previous one (second K3 build only)
    g.slope = (slope1_ok && g.form == 0 && MGM <= 3) ? 1 : 2;
    g.nstrips = 1;
    g.split = g.LL;
    g.hand_base = 0;
    return true;
}

}  // namespace

// The watchdog word of the pass kernel is STICKY on the device: no launch resets it, a copy of it follows every pass
// launch into h_words[1], and only the host clears it, once it has seen it set.  A hand-off time-out of one launch is
// therefore reported by whichever call next finds the stream idle (block = false: pass launches look without waiting --
// nothing on the hot path synchronises for it) or synchronises anyway (block = true), and cannot be overwritten by a
// later launch's copy.
static int check_watchdog(mgm_ctx *c, bool block = true)
{
    if (!c->pending_check) return MGM_OK;
    if (block) HIPCHK(c, hipStreamSynchronize(c->stream));
    else if (hipStreamQuery(c->stream) != hipSuccess) return MGM_OK;  // still running: the word is looked at later
    c->pending_check = false;
    if (c->h_words[1] != 0) {
        c->h_words[1] = 0;
        if (c->words.p) (void)hipMemsetAsync((unsigned *)c->words.p + 1, 0, sizeof(unsigned), c->stream);
        c->hand_key.clear();  // (the launch may have left its hand-off slots half written)
        return fail(c, MGM_ERR_INTERNAL, "pass kernel watchdog: inter-band hand-off timed out");
    }
    return MGM_OK;
}

// ---- pipelined contexts -------------------------------------------------------------------------------------------------
// Every entry point that is not part of gathering a step (cost volume build into a volume no pending call uses, weights,
// the aggregation calls themselves) first runs what has been deferred: downloads, post-processing, frees, synchronisation,
// the direction-sharded building blocks.  A no-op for plain contexts.
static int pipe_flush(mgm_ctx *c);
static int pipe_join(mgm_ctx *c) { return (c && !c->pend.empty()) ? pipe_flush(c) : MGM_OK; }
static bool pipe_uses(const mgm_ctx *c, const void *obj)  // is `obj` (a volume or an image) an operand of a deferred call?
{
    if (!c || !obj) return false;
    for (const auto &q : c->pend) {
        for (const mgm_cv *x : q.C) if (x == obj) return true;
        for (const mgm_img *x : q.w8) if (x == obj) return true;
        for (const mgm_img *x : q.out) if (x == obj) return true;
        for (const mgm_img *x : q.outcost) if (x == obj) return true;
    }
    return false;
}

// ---------------------------------------------------------------------------
extern "C" {

const char *mgm_version(void) { return "mgm-hip 0.1 (gfx950)"; }

int mgm_ctx_create(int device, mgm_ctx **out)
{
    if (!out) return MGM_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return MGM_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return MGM_ERR_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return MGM_ERR_HIP;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return MGM_ERR_HIP;  // the kernels exist for gfx950 only
    mgm_ctx *c = new mgm_ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return MGM_ERR_HIP;
    }
    if (hipHostMalloc((void **)&c->h_words, 16 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return MGM_ERR_HIP;
    }
    memset(c->h_words, 0, 16 * sizeof(unsigned));
    if (const char *e = getenv("MGM_HIP_PASS_BUILD")) c->force_build = atoi(e);
    if (const char *e = getenv("MGM_HIP_DEBUG_STATS")) c->debug_stats = atoi(e);
    *out = c;
    return MGM_OK;
}

int mgm_ctx_destroy(mgm_ctx *c)
{
    if (!c) return MGM_OK;
    (void)hipSetDevice(c->device);
    (void)pipe_join(c);  // (deferred calls of a pipelined context still write the caller's images)
    (void)hipStreamSynchronize(c->stream);
    std::vector<Buf *> bufs = {&c->lr, &c->hand, &c->hand2, &c->handm, &c->exact_mins, &c->words, &c->census_u, &c->census_v, &c->dbg, &c->stmp, &c->ones8};
    for (int v = 0; v < kMaxBatch; v++) {
        bufs.push_back(&c->padf[v]);
        bufs.push_back(&c->pad8[v]);
        bufs.push_back(&c->wsel[v]);
    }
    bufs.push_back(&c->wvals);
    for (auto &t : c->ttabs)
        if (t.buf.p) (void)hipFree(t.buf.p);
    for (Buf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (auto &t : c->tim) {
        (void)hipEventDestroy(t.a);
        (void)hipEventDestroy(t.b);
    }
    if (c->h_words) (void)hipHostFree(c->h_words);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return MGM_OK;
}

// Pipelined context: see mgm_hip.h.  depth 1 switches it off (after running whatever was deferred).
int mgm_ctx_set_pipeline(mgm_ctx *c, int depth)
{
    if (!c) return MGM_ERR_INVALID;
    if (depth < 1 || depth > kMaxBatch) return fail(c, MGM_ERR_INVALID, "mgm_ctx_set_pipeline: depth must be 1..16");
    if (int r = mgm_ctx_synchronize(c)) return r;
    c->pipe_depth = depth;
    return MGM_OK;
}

// The workspace (Lr volumes, hand-off slots, census images, ...) only ever grows with the largest call seen; this hands
// it back to the device.  The next call allocates what it needs again; mgm_wta_windowed_dev / mgm_debug_download_lr /
// mgm_lr_device_ptr have nothing to work on until the next aggregation.
int mgm_ctx_trim(mgm_ctx *c)
{
    if (int jr = pipe_join(c)) return jr;  // (pipelined context: run what has been deferred first)
    if (!c) return MGM_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (int r = mgm_ctx_synchronize(c)) return r;
    std::vector<Buf *> bufs = {&c->lr, &c->hand, &c->hand2, &c->handm, &c->exact_mins, &c->census_u, &c->census_v, &c->dbg, &c->stmp, &c->ones8};
    for (int v = 0; v < kMaxBatch; v++) {
        bufs.push_back(&c->padf[v]);
        bufs.push_back(&c->pad8[v]);
        bufs.push_back(&c->wsel[v]);
    }
    bufs.push_back(&c->wvals);
    for (Buf *b : bufs) {
        if (b->p) (void)hipFree(b->p);
        b->p = nullptr;
        b->cap = 0;
    }
    c->hand_key.clear();
    c->tk_nx = c->tk_ny = c->tk_ndir = c->tk_r = -1;
    c->ntasks = 0;
    for (auto &t : c->ttabs)
        if (t.buf.p) (void)hipFree(t.buf.p);
    c->ttabs.clear();
    c->tasks = Buf{};
    c->last_ndir = c->last_batch = 0;
    for (int v = 0; v < kMaxBatch; v++) c->last_cvs[v] = nullptr;
    return MGM_OK;
}

int mgm_ctx_set_workspace_limit(mgm_ctx *c, unsigned long long bytes)
{
    if (!c) return MGM_ERR_INVALID;
    c->ws_limit = (size_t)bytes;
    return MGM_OK;
}

int mgm_ctx_mem_info(mgm_ctx *c, unsigned long long *free_bytes, unsigned long long *total_bytes)
{
    if (!c) return MGM_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    size_t f = 0, t = 0;
    HIPCHK(c, hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return MGM_OK;
}

const char *mgm_last_error(const mgm_ctx *c) { return c ? c->err.c_str() : "null context"; }

void *mgm_ctx_stream(mgm_ctx *c) { return c ? (void *)c->stream : nullptr; }

int mgm_ctx_synchronize(mgm_ctx *c)
{
    if (!c) return MGM_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (int r = pipe_join(c)) return r;  // (pipelined context: run what has been deferred)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_watchdog(c);
}

int mgm_timing_enable(mgm_ctx *c, int enable)
{
    if (!c) return MGM_ERR_INVALID;
    c->timing = enable != 0;
    return MGM_OK;
}
int mgm_timing_reset(mgm_ctx *c)
{
    if (int jr = pipe_join(c)) return jr;  // (pipelined context: run what has been deferred first)
    if (!c) return MGM_ERR_INVALID;
    (void)hipStreamSynchronize(c->stream);
    for (auto &t : c->tim) {
        (void)hipEventDestroy(t.a);
        (void)hipEventDestroy(t.b);
    }
    c->tim.clear();
    return MGM_OK;
}
int mgm_timing_count(mgm_ctx *c) { return c ? (int)c->tim.size() : 0; }
int mgm_timing_get(mgm_ctx *c, int idx, const char **name, float *ms)
{
    if (!c || idx < 0 || idx >= (int)c->tim.size()) return MGM_ERR_INVALID;
    HIPCHK(c, hipEventSynchronize(c->tim[idx].b));
    float t = 0;
    HIPCHK(c, hipEventElapsedTime(&t, c->tim[idx].a, c->tim[idx].b));
    if (name) *name = c->tim[idx].name;
    if (ms) *ms = t;
    return MGM_OK;
}

// ---- images ---------------------------------------------------------------
int mgm_img_create(mgm_ctx *c, int nx, int ny, int nch, mgm_img **out)
{
    if (!c || !out || nx <= 0 || ny <= 0 || nch <= 0) return fail(c, MGM_ERR_INVALID, "mgm_img_create: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    mgm_img *im = new mgm_img{nullptr, nx, ny, nch, c->device};
    hipError_t e = dev_malloc((void **)&im->d, sizeof(float) * (size_t)nx * ny * nch);
    if (e != hipSuccess) {
        delete im;
        return fail(c, MGM_ERR_NOMEM, std::string("mgm_img_create: ") + hipGetErrorString(e));
    }
    *out = im;
    return MGM_OK;
}
int mgm_img_upload(mgm_ctx *c, const float *host, int nx, int ny, int nch, mgm_img **out)
{
    if (!host) return fail(c, MGM_ERR_INVALID, "mgm_img_upload: null host pointer");
    int r = mgm_img_create(c, nx, ny, nch, out);
    if (r) return r;
    hipError_t e = hipMemcpyAsync((*out)->d, host, sizeof(float) * (size_t)nx * ny * nch, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {  // nothing this call created outlives its failure
        r = hipfail(c, e, "mgm_img_upload: copy");
        mgm_img_free(c, *out);
        *out = nullptr;
        return r;
    }
    return MGM_OK;
}
int mgm_img_download(mgm_ctx *c, const mgm_img *im, float *host)
{
    if (int jr = pipe_join(c)) return jr;  // (pipelined context: run what has been deferred first)
    if (!c || !im || !host) return fail(c, MGM_ERR_INVALID, "mgm_img_download: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(host, im->d, sizeof(float) * (size_t)im->nx * im->ny * im->nch, hipMemcpyDeviceToHost,
                             c->stream));
    return mgm_ctx_synchronize(c);
}
int mgm_img_dims(const mgm_img *im, int *nx, int *ny, int *nch)
{
    if (!im) return MGM_ERR_INVALID;
    if (nx) *nx = im->nx;
    if (ny) *ny = im->ny;
    if (nch) *nch = im->nch;
    return MGM_OK;
}
void *mgm_img_device_ptr(mgm_img *im) { return im ? im->d : nullptr; }
int mgm_img_device(const mgm_img *im) { return im ? im->device : -1; }
int mgm_img_free(mgm_ctx *c, mgm_img *im)
{
    if (int jr = pipe_join(c)) return jr;  // (pipelined context: run what has been deferred first)
    if (!im) return MGM_OK;
    if (c) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
    }
    (void)hipFree(im->d);
    delete im;
    return MGM_OK;
}

// ---- volumes --------------------------------------------------------------
// The fp32 array of a volume is allocated when somebody needs it: a volume K2 fills in the compact form only (single-word
// census costs) never does on the hot path.
static int cv_alloc_f32(mgm_ctx *c, mgm_cv *cv)
{
    if (cv->d) return MGM_OK;
    const size_t n = (size_t)cv->nx * cv->ny * (size_t)(cv->dmax - cv->dmin + 1);
    hipError_t e = dev_malloc((void **)&cv->d, sizeof(float) * n);
    if (e != hipSuccess) {
        cv->d = nullptr;
        return fail(c, MGM_ERR_NOMEM, std::string("cost volume (fp32): ") + hipGetErrorString(e));
    }
    return MGM_OK;
}
static int cv_create(mgm_ctx *c, int nx, int ny, int dmin, int dmax, bool alloc_f32, mgm_cv **out)
{
    if (!c || !out || nx <= 0 || ny <= 0 || dmax < dmin) return fail(c, MGM_ERR_INVALID, "mgm_cv_create: bad arguments");
    const long long L = (long long)dmax - dmin + 1;
    if (L > kMaxLabels)
        return fail(c, MGM_ERR_UNSUPPORTED, "more than 8192 disparity labels per pixel are not supported");
    HIPCHK(c, hipSetDevice(c->device));
    mgm_cv *cv = new mgm_cv();
    cv->d = nullptr;
    cv->nx = nx;
    cv->ny = ny;
    cv->dmin = dmin;
    cv->dmax = dmax;
    cv->owner = c;
    if (alloc_f32)
        if (int r = cv_alloc_f32(c, cv)) {
            delete cv;
            return r;
        }
    if (dev_malloc((void **)&cv->bad8, 64) != hipSuccess) {
        if (cv->d) (void)hipFree(cv->d);
        delete cv;
        return fail(c, MGM_ERR_NOMEM, "mgm_cv_create: flag word");
    }
    cv->gen = next_cv_generation();
    *out = cv;
    return MGM_OK;
}
int mgm_cv_create(mgm_ctx *c, int nx, int ny, int dmin, int dmax, mgm_cv **out) { return cv_create(c, nx, ny, dmin, dmax, true, out); }
int mgm_cv_upload(mgm_ctx *c, const float *dense, int nx, int ny, int dmin, int dmax, mgm_cv **out)
{
    if (!dense) return fail(c, MGM_ERR_INVALID, "mgm_cv_upload: null host pointer");
    int r = mgm_cv_create(c, nx, ny, dmin, dmax, out);
    if (r) return r;
    const size_t n = (size_t)nx * ny * (size_t)(dmax - dmin + 1);
    hipError_t e = hipMemcpyAsync((*out)->d, dense, sizeof(float) * n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        r = hipfail(c, e, "mgm_cv_upload: copy");
        mgm_cv_free(c, *out);
        *out = nullptr;
        return r;
    }
    (*out)->c8_state = 0;
    (*out)->nan_state = 0;
    return MGM_OK;
}
// make cv->d current (see mgm_cv::f32_state); enqueued on the context's stream
static int ensure_f32(mgm_ctx *c, const mgm_cv *ccv)
{
    mgm_cv *cv = const_cast<mgm_cv *>(ccv);
    if (cv->f32_state) return MGM_OK;
    if (!cv->d8 || cv->c8_state < 1) return fail(c, MGM_ERR_INTERNAL, "cost volume has neither an fp32 nor a compact copy");
    if (int r = cv_alloc_f32(c, cv)) return r;
    TimeScope t(c, "k_expand");
    HIPCHK(c, launch_expand(cv->d8, (long long)cv->nx * cv->ny * (cv->dmax - cv->dmin + 1), cv->d, c->stream));
    cv->f32_state = 1;
    return MGM_OK;
}
int mgm_cv_download(mgm_ctx *c, const mgm_cv *cv, float *dense)
{
    if (int jr = pipe_join(c)) return jr;  // (pipelined context: run what has been deferred first)
    if (!c || !cv || !dense) return fail(c, MGM_ERR_INVALID, "mgm_cv_download: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    if (int r = ensure_f32(c, cv)) return r;
    const size_t n = (size_t)cv->nx * cv->ny * (size_t)(cv->dmax - cv->dmin + 1);
    HIPCHK(c, hipMemcpyAsync(dense, cv->d, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    return mgm_ctx_synchronize(c);
}
int mgm_cv_dims(const mgm_cv *cv, int *nx, int *ny, int *dmin, int *dmax)
{
    if (!cv) return MGM_ERR_INVALID;
    if (nx) *nx = cv->nx;
    if (ny) *ny = cv->ny;
    if (dmin) *dmin = cv->dmin;
    if (dmax) *dmax = cv->dmax;
    return MGM_OK;
}
int mgm_cv_device(const mgm_cv *cv) { return (cv && cv->owner) ? cv->owner->device : -1; }
void *mgm_cv_device_ptr(mgm_cv *cv)
{
    if (cv) (void)pipe_join(cv->owner);
    if (!cv) return nullptr;
    if (cv->owner && ensure_f32(cv->owner, cv)) return nullptr;
    cv->c8_state = 0;  // the caller may write through the pointer: re-derive the compact copy at the next use
    cv->nan_state = 0;
    cv->gen = next_cv_generation();
    return cv->d;
}
int mgm_cv_free(mgm_ctx *c, mgm_cv *cv)
{
    if (int jr = pipe_join(c)) return jr;  // (pipelined context: run what has been deferred first)
    if (!cv) return MGM_OK;
    if (c) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
    }
    for (mgm_ctx *o : {c, cv->owner})
        if (o)
            for (int v = 0; v < kMaxBatch; v++)
                if (o->last_cvs[v] == cv) o->last_cvs[v] = nullptr;
    if (cv->d) (void)hipFree(cv->d);
    if (cv->d8) (void)hipFree(cv->d8);
    if (cv->bad8) (void)hipFree(cv->bad8);
    if (cv->rlo) (void)hipFree(cv->rlo);
    if (cv->rhi) (void)hipFree(cv->rhi);
    delete cv;
    return MGM_OK;
}

// ---- compact costs -----------------------------------------------------------
static int c8_alloc(mgm_ctx *c, mgm_cv *cv, int cb = 1)
{
    const size_t n = (size_t)cv->nx * cv->ny * (size_t)(cv->dmax - cv->dmin + 1) * cb + 64;
    if (cv->d8 && cv->d8_cap < n) {  // (refilled with a cost that takes the wider form)
        HIPCHK(c, hipStreamSynchronize(c->stream));
        (void)hipFree(cv->d8);
        cv->d8 = nullptr;
    }
    if (!cv->d8) {
        if (dev_malloc((void **)&cv->d8, n) != hipSuccess) {
            cv->d8 = nullptr;
            cv->d8_cap = 0;
            return fail(c, MGM_ERR_NOMEM, "hipMalloc of the compact cost volume failed");
        }
        cv->d8_cap = n;
    }
    cv->cbytes = cb;
    return MGM_OK;
}
// Decide (once per filling of the volume) whether the compact copy can stand in for C, and whether the volume
// holds NaN costs (mgm_cv::nan_state).  Costs one 4-byte device->host read per filling; MGM_HIP_C8=0 disables the
// compact path.  An uploaded volume is scanned here: by k_compact where it gets a compact copy, else by k_nanscan.
static int c8_resolve(mgm_ctx *c, const mgm_cv *ccv, bool *use)
{
    mgm_cv *cv = const_cast<mgm_cv *>(ccv);
    *use = false;
    const int L = cv->dmax - cv->dmin + 1;
    const bool enabled = dev().c8 && c8_supported(L);
    const long long n = (long long)cv->nx * cv->ny * L;
    bool launched = false;
    if (enabled && cv->c8_state == 0) {  // uploaded / externally written volume: make the compact copy now (one byte per cost)
        int r = c8_alloc(c, cv, 1);
        if (r) return r;
        HIPCHK(c, hipMemsetAsync(cv->bad8, 0, 4, c->stream));
        TimeScope t(c, "k_compact");
        HIPCHK(c, launch_compact(cv->d, n, cv->d8, cv->bad8, c->stream));
        cv->c8_state = 1;
        cv->nan_state = 1;
        launched = true;
    }
    if (cv->nan_state == 0) {
        if (!launched) HIPCHK(c, hipMemsetAsync(cv->bad8, 0, 4, c->stream));
        TimeScope t(c, "k_nanscan");
        HIPCHK(c, launch_nanscan(cv->d, n, cv->bad8, c->stream));
        cv->nan_state = 1;
    }
    if (cv->c8_state == 1 || cv->nan_state == 1) {
        HIPCHK(c, hipMemcpyAsync(c->h_words + 3, cv->bad8, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (cv->c8_state == 1) cv->c8_state = (c->h_words[3] & 1u) ? -1 : 2;
        if (cv->nan_state == 1) cv->nan_state = (c->h_words[3] & 2u) ? -1 : 2;
        if (cv->c8_state < 0 && !cv->f32_state)
            return fail(c, MGM_ERR_INTERNAL, "cost volume predicted to fit the compact form does not");
    }
    *use = enabled && cv->c8_state == 2;
    return MGM_OK;
}

// ---- cost volume ------------------------------------------------------------
static int costvolume_build(mgm_ctx *c, const mgm_img *u, const mgm_img *v, int dmin, int dmax, const mgm_img *rloI,
                            const mgm_img *rhiI, const char *prefilter, const char *distance, float truncDist, int census_win,
                            mgm_cv **out);

int mgm_costvolume_build_dev(mgm_ctx *c, const mgm_img *u, const mgm_img *v, int dmin, int dmax, const char *prefilter,
                             const char *distance, float truncDist, int census_win, mgm_cv **out)
{
    return costvolume_build(c, u, v, dmin, dmax, nullptr, nullptr, prefilter, distance, truncDist, census_win, out);
}

int mgm_costvolume_build_ranged_dev(mgm_ctx *c, const mgm_img *u, const mgm_img *v, const mgm_img *dminI, const mgm_img *dmaxI,
                                    int hull_min, int hull_max, const char *prefilter, const char *distance, float truncDist,
                                    int census_win, mgm_cv **out)
{
    if (!c || !u || !dminI || !dmaxI) return fail(c, MGM_ERR_INVALID, "mgm_costvolume_build_ranged: null argument");
    for (const mgm_img *im : {dminI, dmaxI})
        if (im->nx != u->nx || im->ny != u->ny || im->nch != 1)
            return fail(c, MGM_ERR_INVALID, "mgm_costvolume_build_ranged: the range images must have the left image's size");
    return costvolume_build(c, u, v, hull_min, hull_max, dminI, dmaxI, prefilter, distance, truncDist, census_win, out);
}

static int costvolume_fill(mgm_ctx *c, const mgm_img *u, const mgm_img *v, int dmin, int dmax, const mgm_img *rloI,
                           const mgm_img *rhiI, const char *prefilter, const char *distance, float truncDist, int census_win,
                           mgm_cv **out);

// A volume this call created does not outlive a failure of the call (a caller-provided one stays the caller's).
static int costvolume_build(mgm_ctx *c, const mgm_img *u, const mgm_img *v, int dmin, int dmax, const mgm_img *rloI,
                            const mgm_img *rhiI, const char *prefilter, const char *distance, float truncDist, int census_win,
                            mgm_cv **out)
{
    if (!c || !u || !v || !out) return fail(c, MGM_ERR_INVALID, "mgm_costvolume_build: null argument");
    const bool provided = *out != nullptr;
    const int r = costvolume_fill(c, u, v, dmin, dmax, rloI, rhiI, prefilter, distance, truncDist, census_win, out);
    if (r != MGM_OK && !provided && *out) {
        const std::string msg = c->err;  // (mgm_cv_free synchronises and may touch the message)
        mgm_cv_free(c, *out);
        *out = nullptr;
        c->err = msg;
    }
    return r;
}

static int costvolume_fill(mgm_ctx *c, const mgm_img *u, const mgm_img *v, int dmin, int dmax, const mgm_img *rloI,
                           const mgm_img *rhiI, const char *prefilter, const char *distance, float truncDist, int census_win,
                           mgm_cv **out)
{
    if (u->nch != v->nch) return fail(c, MGM_ERR_INVALID, "mgm_costvolume_build: channel counts differ");
    HIPCHK(c, hipSetDevice(c->device));
    int dist = distance_index(distance), pre = prefilter_index(prefilter);
    const int costfn = dist;  // the function is picked BEFORE the consistency fix (mgm_costvolume.h:355)
    if (dist == 2 || pre == 1) {  // 358-362
        dist = 2;
        pre = 1;
    }

    int r = MGM_OK;
    if (*out) {  // caller-provided volume to refill (must have the right geometry)
        if ((*out)->nx != u->nx || (*out)->ny != u->ny || (*out)->dmin != dmin || (*out)->dmax != dmax)
            return fail(c, MGM_ERR_INVALID, "mgm_costvolume_build: *C is non-NULL but has a different geometry");
        // (pipelined context: a deferred aggregation still wants the costs this volume holds now)
        if (pipe_uses(c, *out) && (r = pipe_flush(c))) return r;
    } else if ((r = cv_create(c, u->nx, u->ny, dmin, dmax, false, out))) {
        return r;
    }
    (*out)->nan_words = false;
    (*out)->gen = next_cv_generation();
    CostParams p{};
    p.C = (*out)->d;
    p.C8 = nullptr;
    p.bad8 = (*out)->bad8;
    HIPCHK(c, hipMemsetAsync((*out)->bad8, 0, 4, c->stream));
    (*out)->c8_state = 0;
    (*out)->nan_state = 1;  // K2 flags NaN costs as it writes them
    if (rloI) {  // the volume keeps its own copy of the range images: K4-K6 need them again
        const size_t nb = sizeof(float) * (size_t)u->nx * u->ny;
        for (float **q : {&(*out)->rlo, &(*out)->rhi})
            if (!*q && dev_malloc((void **)q, nb) != hipSuccess) {
                *q = nullptr;
                return fail(c, MGM_ERR_NOMEM, "mgm_costvolume_build: range images");
            }
        HIPCHK(c, hipMemcpyAsync((*out)->rlo, rloI->d, nb, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync((*out)->rhi, rhiI->d, nb, hipMemcpyDeviceToDevice, c->stream));
    } else if ((*out)->rlo) {  // a refilled volume that used to be ragged
        (void)hipFree((*out)->rlo);
        (void)hipFree((*out)->rhi);
        (*out)->rlo = (*out)->rhi = nullptr;
    }
    p.rlo = (*out)->rlo;
    p.rhi = (*out)->rhi;
    // (NCC costs are (nch - clipped NCC) * 64 and Birchfield-Tomasi costs are built on half-way interpolants: practically
    // never whole numbers, so no compact copy is attempted -- one byte store per label of K2, for nothing)
    const bool may_be_integer = costfn <= 2;
    // Which compact form: census costs are bit counts (one byte); absolute differences of a one-channel 8-bit pair stay below
    // 256, of a colour pair below 766, squared differences below 65026 per channel: two bytes (up to 512 labels: the pass
    // kernels that read them).  The flag word tells afterwards whether every cost really had the form.
    const int cb = (costfn == 2 || (costfn == 0 && u->nch == 1) || dmax - dmin + 1 > 512) ? 1 : 2;
    if (c8_supported(dmax - dmin + 1) && dev().c8) {
        if (may_be_integer) {
            if ((r = c8_alloc(c, *out, cb))) return r;
            p.C8 = (*out)->d8;
            p.cbytes = cb;
            (*out)->c8_state = 1;
        } else
            (*out)->c8_state = -1;
    }
    (*out)->f32_state = 1;
    p.nx = u->nx;
    p.ny = u->ny;
    p.vnx = v->nx;
    p.vny = v->ny;
    p.dmin = dmin;
    p.L = dmax - dmin + 1;
    p.costfn = costfn;
    p.hwin = census_win / 2;  // computeC_clippedNCC: CENSUS_NCC_WIN()/2
    p.nch = u->nch;
    p.u = u->d;
    p.v = v->d;
    if (pre == 1) {
        const int wr = census_win / 2, side = 2 * wr + 1;
        const int nbits = u->nch * (side * side - 1);
        if (wr < 1 || nbits % 8)  // census_tools.cc:81 asserts this
            return fail(c, MGM_ERR_INVALID, "census: nch*(win*win-1) must be a positive multiple of 8");
        const int nwords = (nbits / 8 + 3) / 4;
        if (nwords > kCensusMaxWords) return fail(c, MGM_ERR_UNSUPPORTED, "census descriptor longer than 256 bits");
        (*out)->nan_words = costfn != 2 && nbits > 24;
        if ((r = reserve(c, c->census_u, sizeof(uint32_t) * (size_t)u->nx * u->ny * nwords))) return r;
        if ((r = reserve(c, c->census_v, sizeof(uint32_t) * (size_t)v->nx * v->ny * nwords))) return r;
        {
            TimeScope t(c, "k_census");
            HIPCHK(c, launch_census(u->d, u->nx, u->ny, u->nch, wr, (uint32_t *)c->census_u.p, c->stream));
        }
        {
            TimeScope t(c, "k_census");
            HIPCHK(c, launch_census(v->d, v->nx, v->ny, v->nch, wr, (uint32_t *)c->census_v.p, c->stream));
        }
        p.cu = (const uint32_t *)c->census_u.p;
        p.cv = (const uint32_t *)c->census_v.p;
        p.u = (const float *)c->census_u.p;  // -p census with an ad/sd cost: words read as floats
        p.v = (const float *)c->census_v.p;
        p.nch = nwords;
    }
    if (pre == 2 || pre == 3) {  // sobelx / gblur of both images (mgm_costvolume.h:366-373), then AD or SD on them
        const size_t nu = (size_t)u->nx * u->ny * u->nch, nv = (size_t)v->nx * v->ny * v->nch;
        if ((r = reserve(c, c->census_u, sizeof(float) * nu))) return r;
        if ((r = reserve(c, c->census_v, sizeof(float) * nv))) return r;
        TimeScope t(c, "k_filter2d");
        if (pre == 2) {
            static const float sob[9] = {-1, 0, 1, -2, 0, 2, -1, 0, 1};  // img_tools.h:129-137
            HIPCHK(c, launch_filter2d(u->d, u->nx, u->ny, u->nch, sob, 3, 3, (float *)c->census_u.p, c->stream));
            HIPCHK(c, launch_filter2d(v->d, v->nx, v->ny, v->nch, sob, 3, 3, (float *)c->census_v.p, c->stream));
        } else {
            // gblur_gray with sigma = 1 (img_tools.h:140-180): the taps are computed on the host exactly as there
            const float sigma = 1.0f;
            const float radius = 3 * fabsf(sigma);
            int rr = (int)ceil((double)(1 + 2 * radius));
            rr = rr < 1 ? 1 : (rr > 39 ? 39 : rr);
            float k[39];
            const int cw = (rr - 1) / 2;
            float m = 0;
            for (int i = 0; i < rr; i++) {
                const float x = (float)hypot((double)(i - cw), 0.0);
                const float g = (float)exp((double)(-x * x / (2 * sigma * sigma)));  // (double-precision exp, as compiled there)
                k[i] = g;
                m += g;
            }
            for (int i = 0; i < rr; i++) k[i] /= m;
            if ((r = reserve(c, c->stmp, sizeof(float) * std::max(nu, nv)))) return r;
            HIPCHK(c, launch_filter2d(u->d, u->nx, u->ny, u->nch, k, rr, 1, (float *)c->stmp.p, c->stream));
            HIPCHK(c, launch_filter2d((const float *)c->stmp.p, u->nx, u->ny, u->nch, k, 1, rr, (float *)c->census_u.p, c->stream));
            HIPCHK(c, launch_filter2d(v->d, v->nx, v->ny, v->nch, k, rr, 1, (float *)c->stmp.p, c->stream));
            HIPCHK(c, launch_filter2d((const float *)c->stmp.p, v->nx, v->ny, v->nch, k, 1, rr, (float *)c->census_v.p, c->stream));
        }
        p.u = (const float *)c->census_u.p;
        p.v = (const float *)c->census_v.p;
    }
    if (costfn == 3 && pre == 0 && !rloI && u->nch <= 4) {
        // clipped NCC on the plain images: the per-pixel window statistics are computed once (k_ncc_stats), in the census
        // buffers, which this combination leaves free
        if ((r = reserve(c, c->census_u, sizeof(float) * (size_t)u->nx * u->ny * (2 * u->nch + 1)))) return r;
        if ((r = reserve(c, c->census_v, sizeof(float) * (size_t)v->nx * v->ny * (2 * v->nch + 1)))) return r;
        p.ncc_u = (float *)c->census_u.p;
        p.ncc_v = (float *)c->census_v.p;
    }
    p.trunc = truncDist * (float)p.nch;  // mgm_costvolume.h:401,405
    // A census cost over one descriptor word is a bit count 0..32, clipped to `trunc`: with trunc = +INF
    // or an integer up to 254 every cost fits the compact form, and the fp32 volume -- which neither K3
    // nor k_wta reads then -- is only materialised on demand (ensure_f32).
    if (p.C8 && !p.rlo && costfn == 2 && p.nch == 1 &&
        (p.trunc == __builtin_huge_valf() || (p.trunc >= 0.0f && p.trunc <= 254.0f && p.trunc == rintf(p.trunc))) &&
        dev().lazy_f32) {
        p.C = nullptr;
        (*out)->f32_state = 0;
        // these kernels (k_cost_census8*) compute min(popcount, trunc) in integers: every cost has a compact form and none
        // is NaN BY CONSTRUCTION, so there is no flag to read back -- a refilled volume costs no synchronisation
        (*out)->c8_state = 2;
        (*out)->nan_state = 2;
    } else {
        if ((r = cv_alloc_f32(c, *out))) return r;
        p.C = (*out)->d;
    }
    {
        TimeScope t(c, "k_cost");
        HIPCHK(c, launch_cost(p, c->stream));
    }
    return MGM_OK;
}

int mgm_costvolume_build(mgm_ctx *c, const float *u, const float *v, int nx, int ny, int nch, int vnx, int vny,
                         const float *dminI, const float *dmaxI, const char *prefilter, const char *distance,
                         float truncDist, int census_win, mgm_cv **out)
{
    if (!c || !u || !v || !dminI || !dmaxI || !out) return fail(c, MGM_ERR_INVALID, "mgm_costvolume_build: null argument");
    // Dvec::init receives the float range values converted to int (dvec.cc:55-60)
    int dmin = (int)dminI[0], dmax = (int)dmaxI[0];
    bool ragged = false;
    for (long long i = 0; i < (long long)nx * ny; i++) {
        const int lo = (int)dminI[i], hi = (int)dmaxI[i];
        if (hi < lo) return fail(c, MGM_ERR_INVALID, "mgm_costvolume_build: a pixel's range is empty (dmax < dmin)");
        ragged |= lo != dmin || hi != dmax;
    }
    if (ragged)  // the dense layout spans the hull of all ranges
        for (long long i = 0; i < (long long)nx * ny; i++) {
            dmin = std::min(dmin, (int)dminI[i]);
            dmax = std::max(dmax, (int)dmaxI[i]);
        }
    mgm_img *du = nullptr, *dv = nullptr, *dlo = nullptr, *dhi = nullptr;
    *out = nullptr;
    int r = mgm_img_upload(c, u, nx, ny, nch, &du);
    if (!r) r = mgm_img_upload(c, v, vnx, vny, nch, &dv);
    if (!r && ragged) r = mgm_img_upload(c, dminI, nx, ny, 1, &dlo);
    if (!r && ragged) r = mgm_img_upload(c, dmaxI, nx, ny, 1, &dhi);
    if (!r)
        r = ragged ? mgm_costvolume_build_ranged_dev(c, du, dv, dlo, dhi, dmin, dmax, prefilter, distance, truncDist, census_win, out)
                   : mgm_costvolume_build_dev(c, du, dv, dmin, dmax, prefilter, distance, truncDist, census_win, out);
    if (!r) r = mgm_ctx_synchronize(c);
    for (mgm_img *im : {du, dv, dlo, dhi}) mgm_img_free(c, im);
    return r;
}

// ---- weights ------------------------------------------------------------------
int mgm_weights_dev(mgm_ctx *c, const mgm_img *u, float aP, float aThresh, mgm_img **w8)
{
    if (!c || !u || !w8) return fail(c, MGM_ERR_INVALID, "mgm_weights: null argument");
    const bool provided = *w8 != nullptr;  // (a caller that computes weights pair after pair refills its image)
    int r = MGM_OK;
    if (provided) {
        if ((*w8)->nx != u->nx || (*w8)->ny != u->ny || (*w8)->nch != 8)
            return fail(c, MGM_ERR_INVALID, "mgm_weights: *w8 is non-NULL but is not an nx*ny*8 image");
        if (pipe_uses(c, *w8))  // (pipelined context: a deferred aggregation still wants the weights it holds now)
            if ((r = pipe_flush(c))) return r;
    } else if ((r = mgm_img_create(c, u->nx, u->ny, 8, w8)))
        return r;
    HIPCHK(c, hipSetDevice(c->device));
    TimeScope t(c, "k_weights");
    const hipError_t e = launch_weights(u->d, u->nx, u->ny, u->nch, aP, aThresh, (*w8)->d, c->stream);
    if (e != hipSuccess) {
        r = hipfail(c, e, "k_weights");
        if (!provided) {
            mgm_img_free(c, *w8);
            *w8 = nullptr;
        }
        return r;
    }
    return MGM_OK;
}

// ---- aggregation ----------------------------------------------------------------
// K3 for the passes [first, first+count) of the reference's table; pass p's Lr volume goes to
// workspace slot p - first.  Shared by mgm_aggregate_dev and the direction-sharded multi-GPU path.
// K3 over `nb` cost volumes of identical geometry in one launch (see PassVolume).  Volume v's Lr volumes
// end up at lr + v*count*lr_stride.
// smallest label count the second K3 build takes that holds L labels (0: none)
static int padded_labels(int L)
{
    for (int lp : {64, 128, 192, 256, 384, 512, 768, 1024})
        if (lp >= L) return lp;
    return 0;
}

// K3 for the volumes whose aggregation can meet NaNs: the slow, operand-order-faithful kernel (mgm_pass_exact.hip) --
// the reference's own update functions and schedule, one launch per diagonal, every minimum as the reference writes it.
static int run_passes_exact(mgm_ctx *c, const mgm_cv *const *Cs, const mgm_img *const *w8s, int nb, float P1, float P2, int MGM,
                            bool fh, bool weighted, int first, int count, int slot0, int nslots)
{
    const mgm_cv *C = Cs[0];
    const int nx = C->nx, ny = C->ny, L = C->dmax - C->dmin + 1;
    const long long npix = (long long)nx * ny, nvol = npix * L;
    const long long lr_stride = nvol + lr_pad_floats();
    int r;
    if ((r = reserve(c, c->lr, sizeof(float) * (size_t)lr_stride * nslots * nb))) return r;
    if ((r = reserve(c, c->exact_mins, sizeof(float) * (size_t)npix))) return r;
    ExactParams p{};
    p.nx = nx;
    p.ny = ny;
    p.L = L;
    p.P1 = P1;
    p.P2 = P2;
    p.MGM = MGM;
    // which of the four update functions (mgm_core.cc:548-571)
    p.mode = weighted ? (fh ? 3 : 1) : (fh ? (MGM == 2 ? 2 : 3) : (MGM == 2 ? 0 : 1));
    p.mins = (float *)c->exact_mins.p;
    for (int v = 0; v < nb; v++) {
        if ((r = ensure_f32(c, Cs[v]))) return r;
        p.C = Cs[v]->d;
        p.dmin = Cs[v]->dmin;
        p.w8 = weighted ? w8s[v]->d : nullptr;
        p.rlo = Cs[v]->rlo;
        p.rhi = Cs[v]->rhi;
        for (int q = first; q < first + count; q++) {
            const RefPass &rp = kPasses[q];
            for (int k = 0; k < 4; k++) {
                p.d[k][0] = rp.d[k][0];
                p.d[k][1] = rp.d[k][1];
                p.wplane[k] = kPassToChannel[k][q];
            }
            p.inc_x = rp.inc_x;
            p.inc_y = rp.inc_y;
            p.row_major = rp.row_major;
            p.Lr = (float *)c->lr.p + ((size_t)v * nslots + slot0 + (q - first)) * lr_stride;
            HIPCHK(c, hipMemcpyAsync(p.Lr, p.C, sizeof(float) * (size_t)nvol, hipMemcpyDeviceToDevice, c->stream));  // Lr = CC (495-498)
            TimeScope t(c, "k_pass_exact");
            HIPCHK(c, launch_pass_exact(p, c->stream));
        }
    }
    c->last_nvol = nvol;
    c->last_stride = lr_stride;
    c->last_ndir = nslots;
    c->last_batch = nb;
    c->last_L = L;
    c->last_Lk = L;
    c->last_pad_c8 = false;
    for (int v = 0; v < kMaxBatch; v++) {
        c->last_cvs[v] = v < nb ? Cs[v] : nullptr;
        c->last_gens[v] = v < nb ? Cs[v]->gen : 0;
    }
    return MGM_OK;
}

// slot0 / nslots: pass p's Lr volume goes to workspace slot slot0 + (p - first) of nslots (a caller that launches the
// passes of one volume one at a time keeps them all: mgm_aggregate_passes_at_dev); layout_ndir: the hand-off region is
// laid out for the passes [0, layout_ndir) whichever of them this launch runs, so that such a caller's launches share it.
static int run_passes(mgm_ctx *c, const mgm_cv *const *Cs, const mgm_img *const *w8s, int nb, float P1, float P2, int MGM,
                      int use_fh, int first, int count, bool allow_pad = false, int slot0 = 0, int nslots = 0, int layout_ndir = 0)
{
    if (nslots <= 0) nslots = slot0 + count;
    if (layout_ndir < first + count) layout_ndir = first + count;
    const mgm_cv *C = Cs[0];
    const int nx = C->nx, ny = C->ny, Lreal = C->dmax - C->dmin + 1;
    const int PEND = first + count;
    HIPCHK(c, hipSetDevice(c->device));
    // The second build's unweighted kernels keep the sign bit of the slabs they hand from band to band for a validity
    // tag, which needs E = T - m >= +0, i.e. non-negative penalties (mgm_pass2.hip, TAGS): anything else takes the first build.
    const bool first_build = c->force_build == 1 || !(P1 >= 0.0f) || !(P2 >= 0.0f);
    if (int r0 = check_watchdog(c, false)) return r0;  // (without waiting: the word is sticky on the device)

    // A label count the second build does not take (not 64, 128, 192, 256, 384 or 512) runs PADDED: the kernels see
    // the next such count, the extra label slots hold +INF costs -- "no such label", exactly what a read past a Dvec
    // returns (dvec.cc:129) -- and stay +INF through every update: C = +INF there and every pixel of a volume with a
    // uniform range has a finite minimum, so the added term is finite.
    int L = Lreal;
    bool padded = false;
    // (weights with more than 512 labels run on the first build, which takes any label count as it is: no padding then)
    if (allow_pad && !first_build && pass2_lines(Lreal, false) == 0 && dev().pad && !(w8s && w8s[0] && Lreal > 512)) {
        const int lp = padded_labels(Lreal);
        if (lp) {
            L = lp;
            padded = true;
        }
    }
    const long long npix = (long long)nx * ny, nvol = npix * L;
    const int lpl = pass_lpl(L), LP = lpl * 64;
    int r;
    if ((r = ensure_words(c))) return r;
    unsigned *words = (unsigned *)c->words.p;
    HIPCHK(c, hipMemsetAsync(words, 0, sizeof(unsigned), c->stream));  // the ticket (the progress words: below, where they are used)

    // weighted? (mgm_core.cc:420-423: any value != 1.0 switches every update) -- and what values do the weights take: the
    // planes compute_mgm_weights makes hold 1 and ONE other value, which the pass kernel exploits (k_pass2, W2)
    bool weighted = false, w2cand = false;
    float w2a[kMaxBatch] = {};
    if (w8s && w8s[0]) {
        if ((r = reserve(c, c->wvals, sizeof(unsigned) * 4 * kMaxBatch))) return r;
        unsigned init[4 * kMaxBatch], got[4 * kMaxBatch];
        for (int v = 0; v < nb; v++) init[4 * v] = 0u, init[4 * v + 1] = 0xffffffffu, init[4 * v + 2] = 0u, init[4 * v + 3] = 0u;
        HIPCHK(c, hipMemcpyAsync(c->wvals.p, init, sizeof(unsigned) * 4 * nb, hipMemcpyHostToDevice, c->stream));
        for (int v = 0; v < nb; v++)
            if (w8s[v]) HIPCHK(c, launch_weight_values(w8s[v]->d, npix * 8, (unsigned *)c->wvals.p + 4 * v, c->stream));
        HIPCHK(c, hipMemcpyAsync(got, c->wvals.p, sizeof(unsigned) * 4 * nb, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        w2cand = true;
        for (int v = 0; v < nb; v++) {
            const bool wv = w8s[v] && got[4 * v] != 0;
            if (v && wv != weighted)
                return fail(c, MGM_ERR_UNSUPPORTED, "batched volumes must be all weighted or all unweighted");
            weighted = wv;
            w2cand = w2cand && wv && got[4 * v + 3] == 0 && got[4 * v + 1] == got[4 * v + 2];
            memcpy(&w2a[v], &got[4 * v + 1], 4);
        }
        w2cand = w2cand && weighted;
    }
    const bool fh = use_fh > 0;
    const bool weighted_given = weighted;  // (before the ragged FH path borrows the weighted kernels below)
    // FH potentials on a ragged volume: the min-convolution runs over the RECEIVING pixel's range (mgm_core.cc:242-271), so
    // it cannot be done once by the producer.  The weighted FH kernels convolve on the consumer side anyway: use them,
    // with all-ones weights if the caller has none (update_costW_trunclinear with DeltaI = 1 is what the reference calls
    // then, mgm_core.cc:563-570) -- except for TSGM = 2 without weights, which is update_cost2_trunclinear with its
    // boundary fix-up (166-186, 197-219) and is not built.
    bool ragged = false;
    for (int v = 0; v < nb; v++) ragged |= Cs[v]->rlo != nullptr;
    // Volumes whose aggregation can meet NaNs take the slow kernel that keeps the operand order of the reference's minima
    // (mgm_pass_exact.hip; the fast builds are compiled NaN-free):
    //   * costs that are descriptor WORDS differenced as floats (-p census with a non-census distance, > 24 bits);
    //   * a ragged volume with P2 = +INF: the dense layout relies on every slab keeping a finite minimum, which a finite
    //     P2 guarantees (every term is capped at m + P2); with P2 = +INF a pixel whose neighbours' ranges miss its own gets
    //     an all-INF slab and the next one INF - INF = NaN;
    //   * (found below, by the scan of an uploaded volume) NaN costs.
    //   * more than 2048 labels: no fast kernel is built that wide (the reference's Dvec has no label limit, dvec.cc:60).
    bool exact = (ragged && !(P2 < __builtin_huge_valf())) || Lreal > kMaxLPL * 64;
    for (int v = 0; v < nb; v++) exact |= Cs[v]->nan_words;
    if (exact) return run_passes_exact(c, Cs, w8s, nb, P1, P2, MGM, fh, weighted_given, first, count, slot0, nslots);
    const float *ones8 = nullptr;
    if (fh && ragged)
        for (int v = 1; v < nb; v++)
            if (Cs[v]->dmin != Cs[0]->dmin) return fail(c, MGM_ERR_UNSUPPORTED, "batched ragged volumes must share their hull under FH potentials");
    bool fh2_ragged = false;
    if (fh && ragged && !weighted) {
        // TSGM = 2 without weights is update_cost2_trunclinear with its boundary fix-up (166-186, 197-219): the second
        // build has it (combine_fh2_ragged); the first build does not
        fh2_ragged = MGM == 2;
        if (fh2_ragged && (first_build || (pass2_lines(L, false) == 0)))
            return fail(c, MGM_ERR_UNSUPPORTED, "FH potentials with TSGM=2 and no weights on a ragged cost volume need the second build");
        if ((r = reserve(c, c->ones8, sizeof(float) * (size_t)npix * 8))) return r;
        std::vector<float> one((size_t)npix * 8, 1.0f);
        HIPCHK(c, hipMemcpyAsync(c->ones8.p, one.data(), sizeof(float) * one.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        ones8 = (const float *)c->ones8.p;
        weighted = true;
    }
    int NS = pass_ns(fh, weighted);  // (two slabs per slot also for two-valued weights: decided below)

    // compact costs (one byte per label) when the volume allows it
    bool use_c8 = true;
    std::vector<char> c8ok(nb, 0);
    for (int v = 0; v < nb; v++) {  // (also resolves mgm_cv::nan_state: one scan per filling of a volume)
        bool u = false;
        if ((r = c8_resolve(c, Cs[v], &u))) return r;
        c8ok[v] = u;
        exact |= Cs[v]->nan_state < 0;
    }
    if (exact) return run_passes_exact(c, Cs, w8s, nb, P1, P2, MGM, fh, weighted_given, first, count, slot0, nslots);
    int cb = 1;  // bytes per compact cost of this launch
    if (padded) {
        // padded copies of the costs: a compact form if every volume allows it -- the one that worked for the first volume
        // last time first (mgm_cv::pad_hint), then the other --, else fp32
        int tries[3] = {Cs[0]->pad_hint == 2 ? 2 : 1, Cs[0]->pad_hint == 2 ? 1 : 2, 0};
        if (Cs[0]->pad_hint == 0) tries[0] = 0;
        use_c8 = false;
        for (int t = 0; t < 3 && dev().c8 && !use_c8; t++) {
            const int tb = tries[t];
            if (tb == 0 || (tb == 2 && L > 512)) break;
            HIPCHK(c, hipMemsetAsync(words + 3, 0, sizeof(unsigned), c->stream));
            for (int v = 0; v < nb; v++) {
                if ((r = ensure_f32(c, Cs[v]))) return r;
                if ((r = reserve(c, c->pad8[v], (size_t)npix * L * tb))) return r;
                TimeScope ts(c, "k_pad");
                HIPCHK(c, launch_pad(Cs[v]->d, npix, Lreal, L, nullptr, (uint8_t *)c->pad8[v].p, tb, words + 3, c->stream));
            }
            HIPCHK(c, hipMemcpyAsync(c->h_words + 3, words + 3, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->h_words[3] == 0) {
                use_c8 = true;
                cb = tb;
            }
        }
        for (int v = 0; v < nb; v++) Cs[v]->pad_hint = use_c8 ? cb : 0;
        if (!use_c8)
            for (int v = 0; v < nb; v++) {
                if ((r = ensure_f32(c, Cs[v]))) return r;
                if ((r = reserve(c, c->padf[v], sizeof(float) * (size_t)npix * L))) return r;
                TimeScope t(c, "k_pad");
                HIPCHK(c, launch_pad(Cs[v]->d, npix, Lreal, L, (float *)c->padf[v].p, nullptr, 1, nullptr, c->stream));
            }
    } else {
        for (int v = 0; v < nb; v++) use_c8 = use_c8 && c8ok[v] && Cs[v]->cbytes == Cs[0]->cbytes;
        cb = use_c8 ? Cs[0]->cbytes : 1;
    }
    // Two bytes per cost: read by the unweighted kernels with deep rings that publish E, up to 512 labels (k_pass2, C8 == 2);
    // everything else reads the fp32 volume (which K2 always writes next to a two-byte copy).
    if (use_c8 && cb == 2 && (weighted || (fh && MGM == 2) || pass_lpl(L) > 8 || dev().deep == 0)) {
        if (padded)
            for (int v = 0; v < nb; v++) {
                if ((r = reserve(c, c->padf[v], sizeof(float) * (size_t)npix * L))) return r;
                TimeScope t(c, "k_pad");
                HIPCHK(c, launch_pad(Cs[v]->d, npix, Lreal, L, (float *)c->padf[v].p, nullptr, 1, nullptr, c->stream));
            }
        use_c8 = false;
    }
    // (768 / 1024 labels with weights that are not two-valued-and-narrow: the weighted kernels of the second build stop at
    // 512 labels -- two slabs per slot do not fit the LDS beyond -- so those take the first build, which has no compact costs)
    const bool wide_weighted = weighted && lpl > 8;
    if (first_build || wide_weighted) use_c8 = false;
    // second build (LDS-DMA loaders) whenever the slabs are whole DMA pieces
    // 128 / 64 labels: 2 / 4 volumes of the launch share every wave of the 256-label kernels (k_pass2<..., SUBV>) -- a
    // step is mostly fixed cost, so it may as well serve several volumes.  Compact costs, no weights, not FH with
    // TSGM = 2 (whose slabs travel with their minimum), and a volume count that divides.
    // Only from two such groups on, though: sharing a wave halves the band-steps but makes every step the longer step of
    // the 256-label kernels, and a launch of one group is bound by its chain of bands, i.e. by the step (round 3,
    // 1920x1080x128 x 2: K3 3.74 ms sharing, 3.11 ms as two plain work items; x 4: the same either way).
    int subv = 1;
    if (!first_build && use_c8 && cb == 1 && !weighted && !(fh && MGM == 2) && (L == 128 || L == 64) && nb % (256 / L) == 0 &&
        (dev().subv == 2 || (dev().subv == 1 && nb / (256 / L) >= 2)))
        subv = 256 / L;
    const int ngroups = nb / subv;  // work items address groups of `subv` volumes
    const int Lk = L * subv;        // label slots of a wave
    const int R2 = (first_build || wide_weighted) ? 0 : pass2_lines(Lk, use_c8);
    const int R = R2 ? R2 : (lpl > 8 ? 4 : kR);  // (more than 512 labels: the first build with bands of four lines)
    PassParams p{};
    int maxLL = 0, maxbands = 0;
    for (int q = 0; q < std::max(PEND, layout_ndir); q++) {
        if (!make_geom(q, nx, ny, R, MGM, R2 != 0, p.g[q])) return fail(c, MGM_ERR_INTERNAL, "pass table does not reduce to canonical form");
        maxLL = std::max(maxLL, p.g[q].LL);
        maxbands = std::max(maxbands, p.g[q].nbands);
    }
    if (maxbands > kMaxBands) return fail(c, MGM_ERR_UNSUPPORTED, "image side exceeds 65536 pixels");
    // Two-valued weights (k_pass2, W2): the compact kernels with deep rings and per-XCD queues, every volume's weights 1 and
    // one other positive value.  Anything they do not cover -- fp32 costs, more than 256 labels, launches too small for the
    // queues, a partitioned device, FH on ragged volumes (which borrows the weighted kernels above) -- keeps the general
    // weighted kernels.
    bool w2 = w2cand && dev().w2 && R2 && use_c8 && cb == 1 && lpl <= 4 && !ones8 && !pass2_devtools() && dev().xcdq != 0 && dev().deep != 0;
    if (w2) {
        int items = 0;
        for (int q = first; q < PEND; q++) items += nb * p.g[q].nbands;
        w2 = items >= 32;
    }
    if (w2 && c->xcc_mask < 0) {
        HIPCHK(c, hipMemsetAsync(words + 3, 0, sizeof(unsigned), c->stream));
        HIPCHK(c, launch_xcc_census(words + 3, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_words + 3, words + 3, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->xcc_mask = (int)c->h_words[3];
    }
    w2 = w2 && c->xcc_mask == 0xff;
    const bool wk = weighted && !w2;  // the general weighted kernels (consumer-side transforms, progress words)

    // Consecutive passes' volumes are staggered by an odd number of 256-byte blocks so that the
    // NDIR slabs of one pixel (read together by k_wta) do not fall on the same HBM channel.
    const long long lr_stride = nvol + lr_pad_floats();
    if ((r = reserve(c, c->lr, sizeof(float) * (size_t)lr_stride * nslots * nb))) return r;
    if (w2) NS = 2;
    const int LPk = (subv > 1 ? Lk : LP) * (w2 ? 2 : 1);  // floats per hand-off slot of the self-validating protocol
    // The second build's unweighted kernels hand slabs from band to band that validate themselves (mgm_pass2.hip, TAGS):
    // one slot per (volume, pass, band, pixel), written once per launch OF THAT PASS with the tag in the sign bits.  A
    // pass's tag alternates between its consecutive launches over the same slots; a different geometry clears the region
    // first (all-ones words) and starts every pass again with tag 0.  The region is laid out for the passes
    // [0, layout_ndir) and is this protocol's alone (the other kernels' slots live in `hand2`), so neither a caller that
    // launches the passes one by one nor one that alternates weighted and unweighted runs makes it be cleared again.
    const bool tags = R2 && !wk && (w2 || !(fh && MGM == 2));
    std::string tag_key;
    float *hand_ptr = nullptr;
    if (tags) {
        long long per_vol = 0;
        for (int q = 0; q < layout_ndir; q++) {
            p.g[q].hand_base = per_vol;
            per_vol += (long long)p.g[q].nbands * p.g[q].LL;
        }
        p.hand_vstride = per_vol;
        const size_t bytes = sizeof(float) * (size_t)ngroups * per_vol * LPk;
        const void *before = c->hand.p;
        if ((r = reserve(c, c->hand, bytes))) return r;
        char key[160];
        snprintf(key, sizeof key, "%d %d %d %d %d %d %d", nx, ny, LPk, ngroups, layout_ndir, R, MGM <= 3 ? 1 : 0);  // (LPk tells the W2 layout apart)
        if (c->hand.p != before || c->hand_key != key) {
            HIPCHK(c, hipMemsetAsync(c->hand.p, 0xff, bytes, c->stream));
            c->hand_key = key;
            for (int q = 0; q < kMaxDirs; q++) c->hand_tags[q] = 0x80000000u;  // (what the cleared words look like)
        }
        for (int q = first; q < PEND; q++) {
            c->hand_tags[q] ^= 0x80000000u;
            p.hand_tag[q] = c->hand_tags[q];
        }
        // The tags are only good for a launch that really rewrites every slot of its passes: until the pass kernel has
        // been enqueued the region counts as unknown (the next call clears it), so an error return between here and the
        // launch cannot leave slots behind that carry the tag of the launch after next.
        tag_key = c->hand_key;
        c->hand_key.clear();
        hand_ptr = (float *)c->hand.p;
    } else {
        if ((r = reserve(c, c->hand2, sizeof(float) * (size_t)nb * kMaxDirs * 2 * maxLL * NS * (subv > 1 ? Lk : LP)))) return r;
        hand_ptr = (float *)c->hand2.p;
        // progress words of this protocol: [volume*8 + pass][band]
        HIPCHK(c, hipMemsetAsync(words + 4, 0, sizeof(unsigned) * (size_t)nb * kMaxDirs * kMaxBands, c->stream));
    }
    if ((r = reserve(c, c->handm, sizeof(float) * (size_t)nb * kMaxDirs * 2 * maxLL))) return r;

    double load_ratio = 0;  // band-steps per CU over the longest chain of the launch
    {
        // Two bands per CU pay when the launch is bound by throughput, not by the longest chain of bands: compare the
        // band-steps one CU has to run with the critical path of the slowest pass (steps of slope*lines + line length
        // + the hand-off lag per band: ~3 steps with self-validating slabs, ~10 with progress words).  Measured on
        // 1920x1080 (round 2, after the hand-off rewrite): the FH kernels -- long dependent instruction chains per step --
        // gain from the second band from a ratio of ~1.8 on (three cfg3 volumes per launch; 12 volumes: 64 -> 51 ms); the
        // Hirschmueller kernels only at large batches of 256 labels (+3 % at 12 volumes), and lose 3-10 % at 128 labels
        // or small batches: their steps are short enough for one band to keep the CU's issue slots busy.
        double work = 0, chain = 0;
        const double lag = tags ? 3.0 : 10.0;
        for (int q = first; q < PEND; q++) {
            const PassGeom &g = p.g[q];
            work += (double)ngroups * g.nbands * (g.LL + g.slope * R);
            chain = std::max(chain, (double)g.slope * g.NL + g.LL + lag * g.nbands);
        }
        // (round 3, with the XCD queues: two 256-label FH volumes, ratio 1.66, K3 10.29 -> 9.93 ms with the second band; one
        // volume -- 0.83 -- loses 20 % with it: the FH threshold moved from 1.8 to 1.5)
        p.wg_per_cu = (work / (double)c->num_cu > (fh ? 1.5 : 8.0) * chain) ? 2 : 1;
        load_ratio = work / (double)c->num_cu / chain;
        // Deep DMA rings (k_pass2, DEEP) for every compact unweighted launch: same-process A/B runs of round 3
        // (tools/ab_env.sh, shallow -> deep) give -13 % of K3 for one 128-label volume, -15 % at 4096x4096x192, -3 % for
        // one or two 256-label FH volumes, -3 % for 8 or 16 128-label volumes, and 0..-1 % for twelve 256-label ones.
        p.deep = (tags && use_c8) ? 1 : 0;
    }
    if (dev().deep >= 0) p.deep = (tags && use_c8 && dev().deep) ? 1 : 0;
    if (dev().wg_per_cu) p.wg_per_cu = dev().wg_per_cu;
    // Per-XCD work queues (k_pass2, XCDQ): launches in which the chains of bands matter.  The workgroups stay and work a
    // queue off (a band that follows another on a CU starts at once instead of waiting for a workgroup to be dispatched),
    // and most hand-offs stay inside an XCD's L2.  Same-box A/B runs (round 3, 1920x1080, K3 without -> with queues):
    // 256 labels FH x 1 6.9 -> 6.4 ms, x 2 11.0 -> 10.3, x 3 14.6 -> 13.9, x 4 18.6 -> 17.8, x 6 and x 12 (load/chain 5
    // and 10) 0 .. +1 %; Hirschmueller x 1 5.4 -> 4.85, x 2 8.6 -> 8.15; 128 labels x 1 2.38 -> 2.07, x 3 4.40 -> 4.24;
    // 4096x4096x192 x 1 +-0, x 2 (load/chain 5.4) +1 %.  One queue for all XCDs (MGM_HIP_XCDQ=2: the staying workgroups
    // alone) gives 6.5, 5.2 and 2.08 ms for the three single volumes, 16.3 instead of 15.3 for three 256-label ones.
    // Needs all eight XCC ids to show up in a launch (a partitioned device shows fewer), and a launch large enough for
    // the dispatcher's round robin to have put several workgroups on every XCD: a queue is only worked off by
    // workgroups that find themselves on its XCD -- a small launch keeps the single ticket counter.
    bool xcdq = false;
    int nitems = 0;  // work items of the launch (before strips) = its workgroups
    for (int q = first; q < PEND; q++) nitems += ngroups * p.g[q].nbands;
    // Hirschmueller potentials (short steps: the second band per CU never gave them more than 3 %): with the queues, ONE band per
    // CU is the better schedule at every batch size -- same-box A/B runs of 256-label volumes, two bands per CU without
    // queues -> one with: x 8 0.964 -> 0.985 of the roofline, x 12 0.957 -> 0.981 (K3 46.9 -> 44.9 ms); 4096x4096x192 x 2 +-0 --,
    // so they take the queues whatever the load; the FH kernels, which need the second band from a load/chain of 1.5 on,
    // below a load/chain of 4.
    const bool always_q = !fh;
    if (tags && p.deep && subv == 1 && R2 && nitems >= 32 && !pass2_devtools() && (w2 || dev().xcdq >= 1 || (dev().xcdq < 0 && (always_q || load_ratio < 4.0)))) {
        if (c->xcc_mask < 0) {
            HIPCHK(c, hipMemsetAsync(words + 3, 0, sizeof(unsigned), c->stream));
            HIPCHK(c, launch_xcc_census(words + 3, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->h_words + 3, words + 3, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            c->xcc_mask = (int)c->h_words[3];
        }
        xcdq = c->xcc_mask == 0xff;
    }
    if (xcdq && always_q && !dev().wg_per_cu) p.wg_per_cu = 1;
    if (w2) {
        if (!xcdq || !p.deep) return fail(c, MGM_ERR_INTERNAL, "two-valued weights: the launch plan lost its queues");
        p.wg_per_cu = 1;  // (two slabs per slot: one band per CU)
    }
    // Two strips per line: the passes without an in-line dependency -- form 1 with 2 or 3 neighbours -- walk their lines
    // from both image edges inwards (mgm_pass2.hip): half the line length in the critical path of a pass, bands that
    // live half as long, for twice the work items, each with its own pipeline ramp and hand-off lag (and 1-2 % of the
    // pixels of such a pass computed twice).  Round 3, same-box A/B runs:
    //   * with the deep rings alone the strips LOSE on whole volumes (strips -> none, 1920x1080: 256 labels x 1 K3 7.60 ->
    //     7.21 ms FH, 5.42 -> 5.15 Hirschmueller; 4096x4096x192 27.5 -> 26.2; two or three volumes -2..-5 % too) and win
    //     where a launch runs only a FEW passes of one volume (a rank of a direction-sharded run; 4096x4096x192,
    //     tools/time_passes.py, none -> strips: one pass 8.8 -> 7.8-8.0 ms, two 10.7 -> 9.7, four 15.5 -> 14.9-15.2);
    //   * with the XCD queues -- a finished strip's successor starts at once -- they win wherever the chains dominate
    //     (none -> strips, 1920x1080x256: FH x 1 6.47 -> 6.18, x 2 10.25 -> 10.0, x 3 14.0 -> 13.5 but x 4 17.7 -> 18.1;
    //     Hirschmueller x 1 4.85 -> 4.70, x 2 and x 3 +-0; 4096x4096x192 x 1 (load/chain 2.7) 26.5 -> 27.0, x 2 50.9 -> 52.6;
    //     a rank's four passes of 4096x4096x192 9.6 -> 8.4 with queues and strips together): on below a load/chain of 2.
    bool any_strips = false;
    if (tags && !w2 && (dev().strips == 1 || (dev().strips < 0 && ((ngroups == 1 && count <= 4 && p.wg_per_cu == 1) || (xcdq && load_ratio < 2.0)))))
        for (int q = first; q < PEND; q++)
            if (p.g[q].form == 1 && (MGM == 2 || MGM == 3) && p.g[q].LL >= 8 * R) {
                p.g[q].nstrips = 2;
                p.g[q].split = p.g[q].LL / 2;
                any_strips = true;
            }
    // bands per queue block: a pass stays on one XCD when the passes of the launch fill the eight queues evenly; otherwise
    // blocks of two bands, which spread four or twelve passes over all XCDs at the price of every second hand-off
    // crossing.
    // Measured (K3, block 0 / 1 / 2, no queues): 1920x1080x256 FH x 2 10.2 / 10.9 / 11.3 (11.4), x 3 14.2 / 15.5 / 15.5 (16.5),
    // x 4 17.7 / 18.4 / 17.9 (19.1); Hirschmueller x 3 13.0 / 13.35 / 13.4 (13.5); 128 labels x 1 (four passes) 2.85 / 2.08 /
    // 2.07 (2.35); 4096x4096x192 x 1 27.2 / 27.5 / 26.4 (27.65) -- lines that long keep far more bands in flight than an
    // XCD has CUs, and a pinned pass that takes longer than the others leaves the other XCDs idle at the end.
    int QK = ((ngroups * count) % 8 == 0 && maxLL <= 3000) ? 0 : 2;
    if (dev().xcdq_k >= 0) QK = dev().xcdq_k;
    if (QK <= 0) QK = 1 << 20;
    if (getenv("MGM_HIP_SHOW_PLAN"))  // development aid: what the launch heuristics decided
        fprintf(stderr, "[mgm plan] %dx%dx%d passes %d..%d x %d volumes: load/chain %.2f, %d wg/cu, deep %d, subv %d, strips %d, xcd queues %d (block %d; xcc ids seen 0x%x)\n", nx, ny, L,
                first, PEND - 1, nb, load_ratio, p.wg_per_cu, p.deep, subv, any_strips ? 1 : 0, xcdq ? 1 : 0, QK >= (1 << 20) ? 0 : QK, (unsigned)c->xcc_mask);
    // task table: ticket -> (pass, band [, strip]); item (p, b, .) always follows the items (p, b-1, .)
    const int tk_key = ((((PEND * 16 + first) * kMaxBatch + nb - 1) * 8 + subv) * 2 + (any_strips ? 1 : 0)) * 2 + (xcdq ? 1 : 0);
    if (c->tk_nx != nx || c->tk_ny != ny || c->tk_ndir != tk_key || c->tk_r != R)
        for (auto &t : c->ttabs)
            if (t.nx == nx && t.ny == ny && t.key == tk_key && t.R == R) {  // a shape seen before: its table is still on the device
                c->tasks = t.buf;
                c->ntasks = t.ntasks;
                c->tk_nx = nx, c->tk_ny = ny, c->tk_ndir = tk_key, c->tk_r = R;
                break;
            }
    if (c->tk_nx != nx || c->tk_ny != ny || c->tk_ndir != tk_key || c->tk_r != R) {
        // Passes with more bands (the column passes of a wide image) have the longer dependency
        // chain, so tickets are dealt by RELATIVE progress b / nbands(pass): every pass advances at
        // the rate that lets all of them finish together.  Within a pass the order is still by band.
        std::vector<int2> tasks;
        for (int v = 0; v < ngroups; v++)
            for (int q = first; q < PEND; q++)
                for (int b = 0; b < p.g[q].nbands; b++)
                    for (int st = 0; st < p.g[q].nstrips; st++) tasks.push_back(make_int2(v * kMaxDirs + q, b + (st << 16)));
        std::stable_sort(tasks.begin(), tasks.end(), [&](const int2 &a, const int2 &b) {
            const long long ka = (long long)(a.y & 0xffff) * p.g[b.x % kMaxDirs].nbands, kb = (long long)(b.y & 0xffff) * p.g[a.x % kMaxDirs].nbands;
            return ka != kb ? ka < kb : a.x < b.x;
        });
        if (c->ttabs.size() >= 24) {  // (bounded: drop the oldest; the stream is synchronised below before anything is reused)
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->ttabs.front().buf.p == c->tasks.p) {  // (the table the cached key still names)
                c->tasks = Buf{};
                c->tk_nx = c->tk_ny = c->tk_ndir = c->tk_r = -1;
            }
            if (c->ttabs.front().buf.p) (void)hipFree(c->ttabs.front().buf.p);
            c->ttabs.erase(c->ttabs.begin());
        }
        // The table's header: the eight XCD queues (first ticket, count).  xcdq: the sorted items are dealt to the queues
        // in blocks of QK consecutive bands of a pass, consecutive blocks to consecutive queues, the passes staggered;
        // every queue keeps the global order (what the progress argument of k_pass2 rests on), and an item whose
        // successor band sits in the same queue is marked for a plain hand-off (bit 24).
        std::vector<int2> table(8, make_int2(0, 0));
        if (xcdq) {
            std::vector<int2> qs[8];
            for (const int2 &t : tasks) {
                const int v = t.x / kMaxDirs, q = t.x % kMaxDirs, b = t.y & 0xffff;
                const int chain = v * count + (q - first);
                const bool same = b + 1 < p.g[q].nbands && (b + 1) / QK == b / QK;
                if (dev().xcdq == 2) qs[0].push_back(t);  // (A/B setting: one queue, write-through hand-offs)
                else qs[(b / QK + chain) % 8].push_back(make_int2(t.x, t.y | (same ? 1 << 24 : 0)));
            }
            int at = 0;
            for (int k = 0; k < 8; k++) {
                table[k] = make_int2(at, (int)qs[k].size());
                at += (int)qs[k].size();
                table.insert(table.end(), qs[k].begin(), qs[k].end());
            }
        } else
            table.insert(table.end(), tasks.begin(), tasks.end());
        // (the table in use and its key change together, and only once the new table is on the device: a failure on the
        // way leaves the context with the table -- and the key -- it had)
        Buf fresh{};
        if ((r = reserve(c, fresh, sizeof(int2) * table.size()))) return r;
        hipError_t ce = hipMemcpyAsync(fresh.p, table.data(), sizeof(int2) * table.size(), hipMemcpyHostToDevice, c->stream);
        if (ce == hipSuccess) ce = hipStreamSynchronize(c->stream);
        if (ce != hipSuccess) {
            (void)hipFree(fresh.p);
            return hipfail(c, ce, "task table upload");
        }
        c->ttabs.push_back(mgm_ctx::TaskTab{nx, ny, tk_key, R, (int)tasks.size(), fresh});
        c->tasks = fresh;
        c->ntasks = (int)tasks.size();
        c->tk_nx = nx;
        c->tk_ny = ny;
        c->tk_ndir = tk_key;
        c->tk_r = R;
    }

    for (int v = 0; v < nb; v++) {
        if (!use_c8 && (r = ensure_f32(c, Cs[v]))) return r;
        p.vol[v].C = padded ? (const float *)c->padf[v].p : Cs[v]->d;
        p.vol[v].C8 = use_c8 ? (padded ? (const uint8_t *)c->pad8[v].p : Cs[v]->d8) : nullptr;
        p.vol[v].Lr = (float *)c->lr.p + ((size_t)v * nslots + slot0) * lr_stride;
        p.vol[v].w8 = ones8 ? ones8 : (weighted ? w8s[v]->d : nullptr);
        p.vol[v].rlo = (fh && ragged) ? Cs[v]->rlo : nullptr;
        p.vol[v].rhi = (fh && ragged) ? Cs[v]->rhi : nullptr;
        if (w2) {
            if ((r = reserve(c, c->wsel[v], sizeof(unsigned) * (size_t)npix))) return r;
            HIPCHK(c, launch_wsel(w8s[v]->d, npix, (unsigned *)c->wsel[v].p, c->stream));
            p.vol[v].wsel = (const unsigned *)c->wsel[v].p;
            p.vol[v].p1a = P1 * w2a[v];  // (fp32 products, rounded once: what update_costW computes for D = a)
            p.vol[v].p2a = P2 * w2a[v];
            // (FH: the cap min(., m + P2*a) is skipped where it cannot bind, as for the unit penalties below)
            if (fh && p.vol[v].p1a >= 0.0f && p.vol[v].p2a >= 4.0f * (float)Lk * p.vol[v].p1a + 4096.0f) p.vol[v].p2a = __builtin_huge_valf();
        }
    }
    p.hand = hand_ptr;
    p.handm = (float *)c->handm.p;
    p.ticket = words + 0;
    p.err = words + 1;
    p.prog = words + 4;
    p.tasks = (const int2 *)c->tasks.p + 8;  // (behind the header)
    p.xcdq = xcdq ? (dev().xcdq == 2 ? 2 : 1) : 0;
    p.oneb = (xcdq && p.wg_per_cu < 2 && dev().oneb) ? 1 : 0;
    p.cbytes = use_c8 ? cb : 1;
    p.qticket = words + 4;  // (the progress words of the other protocol: the kernels with tags do not use them)
    if (xcdq) HIPCHK(c, hipMemsetAsync(words + 4, 0, 9 * sizeof(unsigned), c->stream));
    p.npix = npix;
    p.nvol = lr_stride;
    p.L = L;
    p.Lreal = Lreal;
    p.subv = subv;
    p.fh2_ragged = fh2_ragged ? 1 : 0;
    p.MGM = MGM;
    p.dmin = C->dmin;
    p.NDIR = PEND;
    p.pass0 = first;
    p.LLmax = maxLL;
    p.maxbands = kMaxBands;
    p.P1 = P1;
    p.P2 = P2;
    // FH potentials, unweighted, compact costs: min(minconv(L)[o], m + P2) is minconv(L)[o] itself whenever P2 exceeds the
    // longest ramp of a slab by a wide margin -- every label is reached from the slab's minimum in at most L-1 steps of
    // P1, the costs are integers <= 254, so every value of a slab stays below 254 + (L-1)*P1 and the rounding of a ramp
    // of L-1 additions at that magnitude is far below one step.  The kernels skip the cap for P2 = INF (wave-uniform),
    // so it is passed as INF then: same bits, five instructions of the FH step fewer (the reference's own example,
    // P1 = 2, P2 = 20000, is such a case).
    if (fh && tags && use_c8 && P1 >= 0.0f && P2 >= 4.0f * (float)Lk * P1 + 4096.0f) p.P2 = __builtin_huge_valf();
    p.dbg = nullptr;
    p.xflags = 0;
    p.xflags = dev().xflags;
    if ((p.xflags || c->debug_stats) && R2 && !pass2_devtools())
        return fail(c, MGM_ERR_UNSUPPORTED, "MGM_HIP_XFLAGS / MGM_HIP_DEBUG_STATS need a development build of the pass kernels "
                                            "(MGM_P2_DEFINES=-DMGM_P2_DEV=1 python -m mgm_amd.build --force)");
    if (c->debug_stats && R2) {
        if ((r = reserve(c, c->dbg, sizeof(unsigned long long) * 16 * (size_t)c->ntasks))) return r;
        HIPCHK(c, hipMemsetAsync(c->dbg.p, 0, sizeof(unsigned long long) * 16 * (size_t)c->ntasks, c->stream));
        p.dbg = (unsigned long long *)c->dbg.p;
    }
    {
        TimeScope t(c, R2 ? "k_pass2" : "k_pass");
        if (R2) HIPCHK(c, launch_pass2(p, c->ntasks, fh, w2 ? 2 : (wk ? 1 : 0), c->stream));
        else HIPCHK(c, launch_pass(p, c->ntasks, R, fh, weighted ? 1 : 0, c->stream));
    }
    if (tags) c->hand_key = tag_key;  // enqueued: every slot of the region will carry this launch's tag
    if (tags) {
        // MGM_HIP_CHECK_TAGS=1 (debug; synchronises): the invariant the tag protocol rests on, checked after the launch --
        // every word of the slots of this launch's passes (all bands but the last, which hands nothing over) carries the
        // launch's tag.  A slot the kernel skipped would keep its OLD tag and validate a stale slab two launches later.
        const char *chk = getenv("MGM_HIP_CHECK_TAGS");
        if (chk && atoi(chk) != 0) {
            HIPCHK(c, hipMemsetAsync(words + 3, 0, sizeof(unsigned), c->stream));
            for (int v = 0; v < ngroups; v++)
                for (int q = first; q < PEND; q++) {
                    const long long nw = (long long)(p.g[q].nbands - 1) * p.g[q].LL * LPk;
                    if (nw <= 0) continue;
                    HIPCHK(c, launch_check_tags(hand_ptr + ((long long)v * p.hand_vstride + p.g[q].hand_base) * LPk, nw, p.hand_tag[q], words + 3, c->stream));
                }
            HIPCHK(c, hipMemcpyAsync(c->h_words + 3, words + 3, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->h_words[3] != 0) {
                c->hand_key.clear();
                return fail(c, MGM_ERR_INTERNAL, "hand-off slots: " + std::to_string(c->h_words[3]) + " words do not carry the launch's tag");
            }
        }
    }
    HIPCHK(c, hipMemcpyAsync(c->h_words + 1, words + 1, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    c->pending_check = true;
    if (p.dbg) {  // development aid: where does K3's time go?
        std::vector<unsigned long long> d((size_t)c->ntasks * 16);
        std::vector<int2> tk(c->ntasks);
        HIPCHK(c, hipMemcpyAsync(d.data(), p.dbg, d.size() * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(tk.data(), (const int2 *)c->tasks.p + 8, tk.size() * sizeof(int2), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < c->ntasks; i++) {
            t0 = std::min(t0, d[i * 16 + 0]);
            t1 = std::max(t1, d[i * 16 + 2]);
        }
        const double tick = 1e-2;  // wall_clock64: 100 MHz -> 0.01 us
        fprintf(stderr, "[mgm stats] %d workgroups, kernel span %.1f us\n", c->ntasks, (t1 - t0) * tick);
        if (c->debug_stats >= 2)  // one line per work item: pass, band, strip, ticket, start / first step / end (us), time in the slow path
            for (int i = 0; i < c->ntasks; i++)
                fprintf(stderr, "[mgm item] %d %d %d %d %.1f %.1f %.1f %.1f\n", tk[i].x % kMaxDirs, tk[i].y & 0xffff, (tk[i].y >> 16) & 0xff, i,
                        (d[i * 16 + 0] - t0) * tick, (d[i * 16 + 1] - t0) * tick, (d[i * 16 + 2] - t0) * tick, d[i * 16 + 6] * tick);
        for (int q = first; q < PEND; q++) {
            double run = 0, slow = 0, pro = 0, nslow = 0, nspin = 0, steps = 0, first = 1e30, last = 0;
            double ai = 0, ar = 0, ab = 0, bi = 0, br = 0, bb = 0, cb = 0, fsw = 0, fn = 0, fmx = 0, frep = 0;
            int n = 0;
            double fa = 0, fb = 0, fc = 0;
            bool dec = false;
            for (int i = 0; i < c->ntasks; i++)
                if (tk[i].x % kMaxDirs == q) {
                    n++;
                    if (d[i * 16 + 1] >> 63) {  // barrier-free build: failed polls of the profiled wave by cause
                        dec = true;
                        fa += (double)((d[i * 16 + 1] >> 42) & 0x1fffff);
                        fb += (double)((d[i * 16 + 1] >> 21) & 0x1fffff);
                        fc += (double)(d[i * 16 + 1] & 0x1fffff);
                        d[i * 16 + 1] = d[i * 16 + 0];
                    }
                    run += (d[i * 16 + 2] - d[i * 16 + 1]) * tick;
                    pro += (d[i * 16 + 1] - d[i * 16 + 0]) * tick;
                    slow += d[i * 16 + 6] * tick;
                    nslow += d[i * 16 + 3];
                    nspin += d[i * 16 + 4];
                    steps = (double)d[i * 16 + 7];
                    ai += d[i * 16 + 8] * tick; ar += d[i * 16 + 9] * tick; ab += d[i * 16 + 10] * tick;
                    bi += d[i * 16 + 11] * tick; br += d[i * 16 + 12] * tick; bb += d[i * 16 + 13] * tick;
                    cb += d[i * 16 + 14] * tick;
                    fsw += (double)(d[i * 16 + 15] >> 44); fn += (double)((d[i * 16 + 15] >> 18) & 0x3ffff);
                    frep += (double)(d[i * 16 + 15] & 0x3ffff);
                    fmx = std::max(fmx, (double)((d[i * 16 + 15] >> 36) & 0xff));
                    first = std::min(first, (double)(d[i * 16 + 0] - t0) * tick);
                    last = std::max(last, (double)(d[i * 16 + 2] - t0) * tick);
                }
            fprintf(stderr,
                    "[mgm stats] pass %d: %d bands x %.0f steps; per band: prologue %.1f us, main loop %.1f us (%.3f us/step), "
                    "slow-path %.1f us in %.1f polls (%.0f spins); pass active %.1f..%.1f us\n",
                    q, n, steps, pro / n, run / n, run / n / steps, slow / n, nslow / n, nspin / n, first, last);
            fprintf(stderr,
                    "[mgm stats]         loader A: issue %.0f retire %.0f barrier %.0f us | compute wave: barrier-wait %.0f us; "
                    "kcycles per band: lds-read %.0f combine %.0f store+min %.0f transform %.0f lds-write %.0f\n",
                    ai / n, ar / n, ab / n, cb / n, nslow / n / 1e3, nspin / n / 1e3, bi / tick / n / 1e3, br / tick / n / 1e3,
                    bb / tick / n / 1e3);
            if (dec)
                fprintf(stderr, "[mgm stats]         failed polls per band: previous line %.0f, next line %.0f, DMA %.0f\n", fa / n,
                        fb / n, fc / n);
            if (fn > 0)
                fprintf(stderr, "[mgm stats]         FH min-convolution: %.3f sweeps per slab (fwd+bwd, minimum 2), worst %.0f, %.2f%% of slabs repaired\n",
                        fsw / fn, fmx, 100.0 * frep / fn);
        }
    }
    c->last_nvol = nvol;
    c->last_stride = lr_stride;
    c->last_ndir = nslots;
    c->last_batch = nb;
    c->last_L = Lreal;
    c->last_Lk = L;
    c->last_pad_c8 = padded && use_c8;
    c->last_pad_cb = cb;
    for (int v = 0; v < kMaxBatch; v++) {
        c->last_cvs[v] = v < nb ? Cs[v] : nullptr;
        c->last_gens[v] = v < nb ? Cs[v]->gen : 0;
    }

    return MGM_OK;
}

// K4-K6 over `npix` pixels starting at pixel `pix0` of C, reading pass p's Lr from lr + p*lr_stride.
// `slot` >= 0: the volume was slot `slot` of the context's last aggregation; if that launch ran with a padded label
// count, its padded cost copies and label stride are used (see run_passes).  slot < 0: plain [pix][L] layout.
static int run_wta(mgm_ctx *c, const mgm_cv *C, long long pix0, long long npix, const float *lr, long long lr_stride, int NDIR,
                   int fix_overcount, int ridx, float *out, float *outcost, float *Sout, const float *wlo = nullptr,
                   const float *whi = nullptr, int slot = -1)
{
    const int Lreal = C->dmax - C->dmin + 1;
    const bool padded = slot >= 0 && c->last_Lk > c->last_L && c->last_L == Lreal;
    const int L = padded ? c->last_Lk : Lreal;
    WtaParams w{};
    if (padded) {
        w.C = c->last_pad_c8 ? nullptr : (const float *)c->padf[slot].p + pix0 * L;
        w.cbytes = c->last_pad_c8 ? c->last_pad_cb : 1;
        w.C8 = c->last_pad_c8 ? (const uint8_t *)c->pad8[slot].p + pix0 * L * w.cbytes : nullptr;
    } else {
        // (two-byte costs: the exact k_wta instances and k_wta_q read them -- label counts of the compact pass kernels)
        w.cbytes = C->cbytes;
        w.C8 = (C->c8_state == 2 && c->force_build != 1) ? C->d8 + pix0 * L * w.cbytes : nullptr;
        if (!w.C8)
            if (int r = ensure_f32(c, C)) return r;
        w.C = C->d ? C->d + pix0 * L : nullptr;
    }
    w.Lr = lr;
    w.S = Sout;
    w.out = out;
    w.outcost = outcost;
    w.npix = npix;
    w.nvol = lr_stride;
    w.L = L;
    w.Lreal = Lreal;
    w.NDIR = NDIR;
    w.FIX = fix_overcount;
    w.dmin = C->dmin;
    w.refine = ridx;
    // range images are whole-image arrays; this call may cover a slab of rows starting at pix0
    w.wlo = wlo ? wlo + pix0 : nullptr;
    w.whi = whi ? whi + pix0 : nullptr;
    w.clo = C->rlo ? C->rlo + pix0 : nullptr;
    w.chi = C->rhi ? C->rhi + pix0 : nullptr;
    w.num_cu = c->num_cu;
    TimeScope t(c, "k_wta");
    HIPCHK(c, launch_wta(w, c->stream));
    return MGM_OK;
}

// K4-K6 with any refinement of the reference's table: none/vfit are fused into k_wta; parabola, cubic and
// parabolaOCV (refine.h:6-145) run as a second kernel on the corrected S (the caller's, or a scratch volume).
static int run_wta_refine(mgm_ctx *c, const mgm_cv *C, long long pix0, long long npix, const float *lr, long long lr_stride,
                          int NDIR, int fix_overcount, int ridx, float *out, float *outcost, float *Sout,
                          const float *wlo = nullptr, const float *whi = nullptr, int slot = -1)
{
    if (!wlo && C->rlo) {  // a ragged volume: S is allocated from the same range images (mgm_core.cc:426)
        wlo = C->rlo;
        whi = C->rhi;
    }
    if (ridx <= 1 && !wlo) return run_wta(c, C, pix0, npix, lr, lr_stride, NDIR, fix_overcount, ridx, out, outcost, Sout, nullptr, nullptr, slot);
    if (ridx == 0) return run_wta(c, C, pix0, npix, lr, lr_stride, NDIR, fix_overcount, 0, out, outcost, Sout, wlo, whi, slot);
    const int L = C->dmax - C->dmin + 1;
    int r;
    if (!Sout) {
        if ((r = reserve(c, c->stmp, sizeof(float) * (size_t)npix * L))) return r;
        Sout = (float *)c->stmp.p;
    }
    if ((r = run_wta(c, C, pix0, npix, lr, lr_stride, NDIR, fix_overcount, 0, out, outcost, Sout, wlo, whi, slot))) return r;
    // what a disparity of a pixel's window outside the volume holds: S stays 0, minus (NDIR-1)*C with C = +INF
    float vout = 0.0f;
    if (fix_overcount == 1) vout = vout - (float)(NDIR - 1) * __builtin_huge_valf();
    TimeScope t(c, "k_refine");
    HIPCHK(c, launch_refine(Sout, npix, L, C->dmin, ridx, wlo ? wlo + pix0 : nullptr, whi ? whi + pix0 : nullptr, vout, out,
                            outcost, c->stream));
    return MGM_OK;
}

// Arguments of an aggregation call, checked (shared by the immediate and the deferred path)
static int check_aggregate_args(mgm_ctx *c, int n, const mgm_cv *const *C, const mgm_img *const *w8, int NDIR, int MGM, mgm_img *const *out,
                                mgm_img *const *outcost)
{
    if (!c || !C || !out || !outcost || n < 1) return fail(c, MGM_ERR_INVALID, "mgm_aggregate: null argument");
    if (n > kMaxBatch) return fail(c, MGM_ERR_INVALID, "mgm_aggregate_batch: at most 16 volumes per call");
    if (NDIR < 1 || NDIR > kMaxDirs)  // the reference reads past its 8-entry table for -O 16 (mgm_core.cc:489)
        return fail(c, MGM_ERR_INVALID, "NDIR must be 1..8");
    if (MGM < 1 || MGM > 4) return fail(c, MGM_ERR_INVALID, "MGM (TSGM) must be 1..4");
    for (int v = 0; v < n; v++)
        if (!C[v] || !out[v] || !outcost[v]) return fail(c, MGM_ERR_INVALID, "mgm_aggregate: null argument");
    const int nx = C[0]->nx, ny = C[0]->ny, L = C[0]->dmax - C[0]->dmin + 1;
    for (int v = 0; v < n; v++) {
        if (C[v]->nx != nx || C[v]->ny != ny || C[v]->dmax - C[v]->dmin + 1 != L)
            return fail(c, MGM_ERR_INVALID, "mgm_aggregate_batch: the volumes must have the same size and label count");
        if (out[v]->nx != nx || out[v]->ny != ny || outcost[v]->nx != nx || outcost[v]->ny != ny)
            return fail(c, MGM_ERR_INVALID, "mgm_aggregate: output image size mismatch");
        if (w8 && w8[v] && (w8[v]->nx != nx || w8[v]->ny != ny || w8[v]->nch != 8))
            return fail(c, MGM_ERR_INVALID, "mgm_aggregate: weights must be nx*ny*8");
        if (w8 && (w8[v] == nullptr) != (w8[0] == nullptr))
            return fail(c, MGM_ERR_INVALID, "mgm_aggregate_batch: weights for all volumes or for none");
    }
    return MGM_OK;
}

// The aggregation itself, now: one pass launch over the batch (several if it does not fit), then the winner search per volume.
static int aggregate_batch_now(mgm_ctx *c, int n, const mgm_cv *const *C, const mgm_img *const *w8, float P1, float P2, int NDIR,
                               int MGM, int use_fh, int fix_overcount, const char *refine, mgm_img *const *out,
                               mgm_img *const *outcost, mgm_cv **S)
{
    const int nx = C[0]->nx, ny = C[0]->ny, L = C[0]->dmax - C[0]->dmin + 1;
    const int ridx = refinement_index(refine);
    HIPCHK(c, hipSetDevice(c->device));
    int r = MGM_OK;
    const long long npix = (long long)nx * ny;
    if (S)
        for (int v = 0; v < n; v++) S[v] = nullptr;
    // The Lr volumes of a launch take NDIR x W x H x L floats per volume.  A batch that does not fit the caller's
    // workspace limit (mgm_ctx_set_workspace_limit), or the device (hipMalloc fails), is run as several launches over
    // the largest sub-batches that do -- multiples of four / two volumes first, so that volumes keep sharing waves at
    // 64 / 128 labels -- instead of failing with MGM_ERR_NOMEM: same results, the later volumes just wait their turn.
    int chunk = n;
    if (c->ws_limit) {
        const int Lk = padded_labels(L) ? padded_labels(L) : L;
        const double per_vol = 4.0 * ((double)npix * Lk + (double)lr_pad_floats()) * NDIR * 1.07;  // (+ the hand-off slots: ~7 %)
        while (chunk > 1 && per_vol * chunk > (double)c->ws_limit) chunk--;
        if (chunk >= 4) chunk -= chunk % 4;
        else if (chunk == 3) chunk = 2;
    }
    for (int v0 = 0; v0 < n && !r;) {
        int m = std::min(chunk, n - v0);
        r = run_passes(c, C + v0, (w8 && w8[0]) ? w8 + v0 : nullptr, m, P1, P2, MGM, use_fh, 0, NDIR, /*allow_pad=*/true);
        if (r == MGM_ERR_NOMEM && m > 1) {  // does not fit the device either: halve and try again
            chunk = m > 4 ? (m / 2) - (m / 2) % 2 : m / 2;
            chunk = std::max(chunk, 1);
            r = MGM_OK;
            continue;
        }
        for (int v = 0; v < m && !r; v++) {
            float *Sout = nullptr;
            if (S) {
                if ((r = mgm_cv_create(c, nx, ny, C[v0 + v]->dmin, C[v0 + v]->dmax, &S[v0 + v]))) break;
                Sout = S[v0 + v]->d;
            }
            const float *lr = (const float *)c->lr.p + (size_t)v * NDIR * c->last_stride;
            r = run_wta_refine(c, C[v0 + v], 0, npix, lr, c->last_stride, NDIR, fix_overcount, ridx, out[v0 + v]->d, outcost[v0 + v]->d,
                               Sout, nullptr, nullptr, v);
        }
        v0 += m;
    }
    if (r && S) {  // no S volume of a failed call is handed out
        const std::string msg = c->err;
        for (int v = 0; v < n; v++) {
            if (S[v]) mgm_cv_free(c, S[v]);
            S[v] = nullptr;
        }
        c->err = msg;
    }
    return r;
}

// Pipelined context: everything that has been deferred, as ONE batch (the calls were checked to fit together when they
// were queued).  An error is the error of the call that made the flush happen.
static int pipe_flush(mgm_ctx *c)
{
    if (c->pend.empty()) return MGM_OK;
    std::vector<mgm_ctx::PendingAgg> q;
    q.swap(c->pend);  // (nothing below may see them as pending any more)
    std::vector<const mgm_cv *> Cs;
    std::vector<const mgm_img *> Ws;
    std::vector<mgm_img *> Os, Ks;
    for (const auto &a : q) {
        Cs.insert(Cs.end(), a.C.begin(), a.C.end());
        Ws.insert(Ws.end(), a.w8.begin(), a.w8.end());
        Os.insert(Os.end(), a.out.begin(), a.out.end());
        Ks.insert(Ks.end(), a.outcost.begin(), a.outcost.end());
    }
    const auto &a = q[0];
    return aggregate_batch_now(c, (int)Cs.size(), Cs.data(), Ws.empty() ? nullptr : Ws.data(), a.P1, a.P2, a.NDIR, a.MGM, a.use_fh,
                               a.fix_overcount, a.has_refine ? a.refine.c_str() : nullptr, Os.data(), Ks.data(), nullptr);
}

int mgm_aggregate_batch_dev(mgm_ctx *c, int n, const mgm_cv *const *C, const mgm_img *const *w8, float P1, float P2, int NDIR,
                            int MGM, int use_fh, int fix_overcount, const char *refine, mgm_img *const *out,
                            mgm_img *const *outcost, mgm_cv **S)
{
    if (int r = check_aggregate_args(c, n, C, w8, NDIR, MGM, out, outcost)) return r;
    if (c->pipe_depth < 2) return aggregate_batch_now(c, n, C, w8, P1, P2, NDIR, MGM, use_fh, fix_overcount, refine, out, outcost, S);
    // Pipelined context (mgm_ctx_set_pipeline): the call is DEFERRED -- remembered, not run -- until `depth` calls have been
    // gathered, and those then run as one batch: ONE launch of the pass kernel over all their volumes, where the chains of
    // bands of one volume fill the gaps of the others'.  A call that does not fit what is waiting (other geometry or
    // settings, weighted against unweighted, S wanted, an operand that a waiting call already uses as an output, more than
    // 16 volumes together) makes the waiting ones run first.
    const bool weighted_call = w8 && w8[0];
    bool fits = !S;
    int waiting = 0;
    if (!c->pend.empty()) {
        const auto &a = c->pend[0];
        for (const auto &q : c->pend) waiting += (int)q.C.size();
        const mgm_cv *A = a.C[0];
        fits = fits && A->nx == C[0]->nx && A->ny == C[0]->ny && A->dmax - A->dmin == C[0]->dmax - C[0]->dmin && a.P1 == P1 && a.P2 == P2 &&
               a.NDIR == NDIR && a.MGM == MGM && a.use_fh == use_fh && a.fix_overcount == fix_overcount && a.has_refine == (refine != nullptr) &&
               (!refine || a.refine == refine) && !a.w8.empty() == weighted_call && waiting + n <= kMaxBatch;
        for (int v = 0; v < n && fits; v++) fits = !pipe_uses(c, out[v]) && !pipe_uses(c, outcost[v]);
    }
    for (int v = 0; v < n && fits; v++)  // (within the call: an image cannot be two outputs)
        for (int u = 0; u < n; u++) fits = fits && out[v] != outcost[u] && (u == v || (out[v] != out[u] && outcost[v] != outcost[u]));
    if (!fits) {
        if (int r = pipe_flush(c)) return r;
        if (S) return aggregate_batch_now(c, n, C, w8, P1, P2, NDIR, MGM, use_fh, fix_overcount, refine, out, outcost, S);
    }
    mgm_ctx::PendingAgg a;
    a.C.assign(C, C + n);
    if (weighted_call) a.w8.assign(w8, w8 + n);
    a.out.assign(out, out + n);
    a.outcost.assign(outcost, outcost + n);
    a.P1 = P1, a.P2 = P2, a.NDIR = NDIR, a.MGM = MGM, a.use_fh = use_fh, a.fix_overcount = fix_overcount;
    a.has_refine = refine != nullptr;
    a.refine = refine ? refine : "";
    c->pend.push_back(std::move(a));
    if ((int)c->pend.size() >= c->pipe_depth) return pipe_flush(c);
    return MGM_OK;
}

int mgm_aggregate_dev(mgm_ctx *c, const mgm_cv *C, const mgm_img *w8, float P1, float P2, int NDIR, int MGM, int use_fh,
                      int fix_overcount, const char *refine, mgm_img *out, mgm_img *outcost, mgm_cv **S)
{
    return mgm_aggregate_batch_dev(c, 1, &C, w8 ? &w8 : nullptr, P1, P2, NDIR, MGM, use_fh, fix_overcount, refine, &out,
                                   &outcost, S);
}

// ---- direction sharding (multi-GPU): run a subset of the passes, sum slabs of Lr volumes ----------
int mgm_aggregate_passes_dev(mgm_ctx *c, const mgm_cv *C, const mgm_img *w8, float P1, float P2, int MGM, int use_fh,
                             int first_pass, int n_passes)
{
    if (int jr = pipe_join(c)) return jr;  // (pipelined context: run what has been deferred first)
    if (!c || !C) return fail(c, MGM_ERR_INVALID, "mgm_aggregate_passes: null argument");
    if (first_pass < 0 || n_passes < 1 || first_pass + n_passes > kMaxDirs)
        return fail(c, MGM_ERR_INVALID, "mgm_aggregate_passes: passes must lie in 0..7");
    if (MGM < 1 || MGM > 4) return fail(c, MGM_ERR_INVALID, "MGM (TSGM) must be 1..4");
    if (w8 && (w8->nx != C->nx || w8->ny != C->ny || w8->nch != 8))
        return fail(c, MGM_ERR_INVALID, "mgm_aggregate_passes: weights must be nx*ny*8");
    HIPCHK(c, hipSetDevice(c->device));
    return run_passes(c, &C, w8 ? &w8 : nullptr, 1, P1, P2, MGM, use_fh, first_pass, n_passes);
}

// The same with the caller saying where the Lr volumes go: pass p lands in workspace slot slot0 + (p - first_pass) of
// n_slots, and the context lays its hand-off region out for the passes [0, NDIR_total).  A caller that launches the
// passes of one volume one at a time (to send pass k's slabs while pass k+1 runs) keeps all of them this way.
int mgm_aggregate_passes_at_dev(mgm_ctx *c, const mgm_cv *C, const mgm_img *w8, float P1, float P2, int MGM, int use_fh,
                                int first_pass, int n_passes, int slot0, int n_slots, int NDIR_total)
{
    if (int jr = pipe_join(c)) return jr;  // (pipelined context: run what has been deferred first)
    if (!c || !C) return fail(c, MGM_ERR_INVALID, "mgm_aggregate_passes_at: null argument");
    if (first_pass < 0 || n_passes < 1 || first_pass + n_passes > kMaxDirs || NDIR_total > kMaxDirs)
        return fail(c, MGM_ERR_INVALID, "mgm_aggregate_passes_at: passes must lie in 0..7");
    if (slot0 < 0 || n_slots < slot0 + n_passes || n_slots > kMaxDirs)
        return fail(c, MGM_ERR_INVALID, "mgm_aggregate_passes_at: the passes do not fit the slots");
    if (MGM < 1 || MGM > 4) return fail(c, MGM_ERR_INVALID, "MGM (TSGM) must be 1..4");
    if (w8 && (w8->nx != C->nx || w8->ny != C->ny || w8->nch != 8))
        return fail(c, MGM_ERR_INVALID, "mgm_aggregate_passes_at: weights must be nx*ny*8");
    HIPCHK(c, hipSetDevice(c->device));
    return run_passes(c, &C, w8 ? &w8 : nullptr, 1, P1, P2, MGM, use_fh, first_pass, n_passes, false, slot0, n_slots, NDIR_total);
}

void *mgm_lr_device_ptr(mgm_ctx *c, int slot)
{
    (void)pipe_join(c);
    if (!c || !c->lr.p || slot < 0 || slot >= c->last_ndir || c->last_Lk != c->last_L) return nullptr;
    return (float *)c->lr.p + (size_t)slot * c->last_stride;
}

int mgm_wta_rows_dev(mgm_ctx *c, const mgm_cv *C, int row0, int nrows, const void *lr_slabs, int NDIR, int fix_overcount,
                     const char *refine, void *out_rows, void *outcost_rows)
{
    if (int jr = pipe_join(c)) return jr;  // (pipelined context: run what has been deferred first)
    if (!c || !C || !lr_slabs || !out_rows || !outcost_rows) return fail(c, MGM_ERR_INVALID, "mgm_wta_rows: null argument");
    if (row0 < 0 || nrows < 1 || row0 + nrows > C->ny || NDIR < 1 || NDIR > kMaxDirs)
        return fail(c, MGM_ERR_INVALID, "mgm_wta_rows: bad row range or NDIR");
    const int ridx = refinement_index(refine);
    HIPCHK(c, hipSetDevice(c->device));
    const long long L = C->dmax - C->dmin + 1, slab = (long long)nrows * C->nx * L;
    return run_wta_refine(c, C, (long long)row0 * C->nx, (long long)nrows * C->nx, (const float *)lr_slabs, slab, NDIR,
                   fix_overcount, ridx, (float *)out_rows, (float *)outcost_rows, nullptr);
}

int mgm_aggregate(mgm_ctx *c, const mgm_cv *C, const float *w8, float P1, float P2, int NDIR, int MGM, int use_fh,
                  int fix_overcount, const char *refine, float *out, float *outcost, mgm_cv **S)
{
    if (!c || !C || !out || !outcost) return fail(c, MGM_ERR_INVALID, "mgm_aggregate: null argument");
    mgm_img *dw = nullptr, *dout = nullptr, *dcost = nullptr;
    int r = MGM_OK;
    if (w8) r = mgm_img_upload(c, w8, C->nx, C->ny, 8, &dw);
    if (!r) r = mgm_img_create(c, C->nx, C->ny, 1, &dout);
    if (!r) r = mgm_img_create(c, C->nx, C->ny, 1, &dcost);
    if (!r) r = mgm_aggregate_dev(c, C, dw, P1, P2, NDIR, MGM, use_fh, fix_overcount, refine, dout, dcost, S);
    if (!r) r = mgm_img_download(c, dout, out);
    if (!r) r = mgm_img_download(c, dcost, outcost);
    mgm_img_free(c, dw);
    mgm_img_free(c, dout);
    mgm_img_free(c, dcost);
    return r;
}

int mgm_debug_download_lr(mgm_ctx *c, int pass, float *dense)
{
    if (int jr = pipe_join(c)) return jr;  // (pipelined context: run what has been deferred first)
    if (!c || !dense || pass < 0 || pass >= c->last_ndir || !c->lr.p)
        return fail(c, MGM_ERR_INVALID, "mgm_debug_download_lr: nothing to download");
    HIPCHK(c, hipSetDevice(c->device));
    // (a launch with a padded label count keeps last_Lk floats per pixel, of which the first last_L exist)
    HIPCHK(c, hipMemcpy2DAsync(dense, sizeof(float) * c->last_L, (const float *)c->lr.p + (size_t)pass * c->last_stride,
                               sizeof(float) * c->last_Lk, sizeof(float) * c->last_L, (size_t)(c->last_nvol / c->last_Lk),
                               hipMemcpyDeviceToHost, c->stream));
    return mgm_ctx_synchronize(c);
}

// ---- self-tests -------------------------------------------------------------------
int mgm_selftest_div3(mgm_ctx *c, unsigned long long *nbad)
{
    if (int jr = pipe_join(c)) return jr;  // (pipelined context: run what has been deferred first)
    if (!c || !nbad) return fail(c, MGM_ERR_INVALID, "mgm_selftest_div3: null argument");
    HIPCHK(c, hipSetDevice(c->device));
    int r;
    if ((r = ensure_words(c))) return r;
    unsigned long long *d = (unsigned long long *)c->words.p + 1;  // (words 2 and 3: scratch; word 1 is the sticky watchdog word)
    HIPCHK(c, hipMemsetAsync(d, 0, sizeof(unsigned long long), c->stream));
    HIPCHK(c, launch_selftest_div3(d, c->stream));
    HIPCHK(c, hipMemcpyAsync(nbad, d, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MGM_OK;
}

// ---- refinement -----------------------------------------------------------------
int mgm_refine_dev(mgm_ctx *c, const mgm_cv *S, const char *method, mgm_img *out, mgm_img *outcost)
{
    if (int jr = pipe_join(c)) return jr;  // (pipelined context: run what has been deferred first)
    if (!c || !S || !out || !outcost) return fail(c, MGM_ERR_INVALID, "mgm_refine: null argument");
    if (out->nx != S->nx || out->ny != S->ny || outcost->nx != S->nx || outcost->ny != S->ny)
        return fail(c, MGM_ERR_INVALID, "mgm_refine: image size mismatch");
    const int m = refinement_index(method);
    if (m == 0) return MGM_OK;  // "none" and unknown names (mgm_refine.h:28-35)
    HIPCHK(c, hipSetDevice(c->device));
    TimeScope t(c, "k_refine");
    HIPCHK(c, launch_refine(S->d, (long long)S->nx * S->ny, S->dmax - S->dmin + 1, S->dmin, m, nullptr, nullptr, 0.0f, out->d,
                            outcost->d, c->stream));
    return MGM_OK;
}

int mgm_wta_windowed_dev(mgm_ctx *c, const mgm_cv *C, int NDIR, int fix_overcount, const char *refine, const mgm_img *dminI,
                         const mgm_img *dmaxI, mgm_img *out, mgm_img *outcost)
{
    if (int jr = pipe_join(c)) return jr;  // (pipelined context: run what has been deferred first)
    if (!c || !C || !dminI || !dmaxI || !out || !outcost) return fail(c, MGM_ERR_INVALID, "mgm_wta_windowed: null argument");
    const int nx = C->nx, ny = C->ny, L = C->dmax - C->dmin + 1;
    for (const mgm_img *im : {dminI, dmaxI, (const mgm_img *)out, (const mgm_img *)outcost})
        if (im->nx != nx || im->ny != ny || im->nch != 1) return fail(c, MGM_ERR_INVALID, "mgm_wta_windowed: image size mismatch");
    int slot = -1;
    for (int v = 0; v < c->last_batch; v++)
        if (c->last_cvs[v] == C && c->last_gens[v] == C->gen) slot = v;
    if (!c->lr.p || slot < 0 || c->last_ndir != NDIR || c->last_L != L)
        return fail(c, MGM_ERR_INVALID, "mgm_wta_windowed: this volume was not part of the context's last aggregation with NDIR passes");
    HIPCHK(c, hipSetDevice(c->device));
    return run_wta_refine(c, C, 0, (long long)nx * ny, (const float *)c->lr.p + (size_t)slot * NDIR * c->last_stride, c->last_stride, NDIR, fix_overcount,
                          refinement_index(refine), out->d, outcost->d, nullptr, dminI->d, dmaxI->d, slot);
}

int mgm_update_ranges_dev(mgm_ctx *c, const mgm_img *outoff, mgm_img *dminI, mgm_img *dmaxI, int slack, int radius)
{
    if (int jr = pipe_join(c)) return jr;  // (pipelined context: run what has been deferred first)
    if (!c || !outoff || !dminI || !dmaxI) return fail(c, MGM_ERR_INVALID, "mgm_update_ranges: null argument");
    for (const mgm_img *im : {(const mgm_img *)dminI, (const mgm_img *)dmaxI})
        if (im->nx != outoff->nx || im->ny != outoff->ny || im->nch != 1 || outoff->nch != 1)
            return fail(c, MGM_ERR_INVALID, "mgm_update_ranges: image size mismatch");
    if (radius < 0 || radius > 16) return fail(c, MGM_ERR_INVALID, "mgm_update_ranges: radius must be 0..16");
    HIPCHK(c, hipSetDevice(c->device));
    int r;
    if ((r = ensure_words(c))) return r;
    TimeScope t(c, "k_update_ranges");
    // (the two words of the global minimum / maximum live at the end of the control block, which K3 does not use)
    HIPCHK(c, launch_update_ranges(outoff->d, outoff->nx, outoff->ny, slack, radius, dminI->d, dmaxI->d,
                                   (float *)c->words.p + kCtrlWords - 2, c->stream));
    return MGM_OK;
}

int mgm_median_dev(mgm_ctx *c, const mgm_img *in, int radius, mgm_img *out)
{
    if (int jr = pipe_join(c)) return jr;  // (pipelined context: run what has been deferred first)
    if (!c || !in || !out || in == out) return fail(c, MGM_ERR_INVALID, "mgm_median: bad arguments");
    if (out->nx != in->nx || out->ny != in->ny || out->nch != in->nch) return fail(c, MGM_ERR_INVALID, "mgm_median: image size mismatch");
    if (radius < 1 || radius > 1024) return fail(c, MGM_ERR_INVALID, "mgm_median: radius must be 1..1024");
    // Beyond radius 7 the order statistic costs 33 sweeps of the (2r+1)^2 window per pixel, read straight from memory
    // (no tiling): bounded here so that the call finishes in about a minute at most -- 1920x1080 up to radius ~190;
    // the launches themselves are cut into pieces of bounded work (mgm_post.hip).
    constexpr double kMedianMaxReads = 1.0e13;
    if (radius > 7 && 33.0 * (2.0 * radius + 1.0) * (2.0 * radius + 1.0) * (double)in->nx * in->ny * in->nch > kMedianMaxReads)
        return fail(c, MGM_ERR_UNSUPPORTED, "mgm_median: window too large for this image (33 * (2r+1)^2 * pixels must stay below 1e13)");
    HIPCHK(c, hipSetDevice(c->device));
    TimeScope t(c, "k_median");
    HIPCHK(c, launch_median(in->d, in->nx, in->ny, in->nch, radius, out->d, c->stream));
    return MGM_OK;
}

int mgm_leftright_dev(mgm_ctx *c, const mgm_img *d, const mgm_img *other, float tau, mgm_img *out)
{
    if (int jr = pipe_join(c)) return jr;  // (pipelined context: run what has been deferred first)
    if (!c || !d || !other || !out || out == other) return fail(c, MGM_ERR_INVALID, "mgm_leftright: bad arguments");
    if (d->nch != 1 || other->nch != 1 || out->nch != 1 || out->nx != d->nx || out->ny != d->ny || other->ny < d->ny)
        return fail(c, MGM_ERR_INVALID, "mgm_leftright: image size mismatch");
    HIPCHK(c, hipSetDevice(c->device));
    TimeScope t(c, "k_leftright");
    HIPCHK(c, launch_leftright(d->d, d->nx, d->ny, other->d, other->nx, tau, out->d, c->stream));
    return MGM_OK;
}

int mgm_backproject_dev(mgm_ctx *c, const mgm_img *u, const mgm_img *v, const mgm_img *disp, mgm_img *out)
{
    if (int jr = pipe_join(c)) return jr;  // (pipelined context: run what has been deferred first)
    if (!c || !u || !v || !disp || !out) return fail(c, MGM_ERR_INVALID, "mgm_backproject: null argument");
    if (u->nch != v->nch || disp->nx != u->nx || disp->ny != u->ny || disp->nch != 1 || out->nx != u->nx || out->ny != u->ny ||
        out->nch != u->nch)
        return fail(c, MGM_ERR_INVALID, "mgm_backproject: image size mismatch");
    HIPCHK(c, hipSetDevice(c->device));
    TimeScope t(c, "k_backproject");
    HIPCHK(c, launch_backproject(u->d, u->nx, u->ny, u->nch, v->d, v->nx, v->ny, disp->d, out->d, c->stream));
    return MGM_OK;
}

int mgm_refine(mgm_ctx *c, const mgm_cv *S, const char *method, float *out, float *outcost)
{
    if (!c || !S || !out || !outcost) return fail(c, MGM_ERR_INVALID, "mgm_refine: null argument");
    mgm_img *dout = nullptr, *dcost = nullptr;
    int r = mgm_img_upload(c, out, S->nx, S->ny, 1, &dout);
    if (!r) r = mgm_img_upload(c, outcost, S->nx, S->ny, 1, &dcost);
    if (!r) r = mgm_refine_dev(c, S, method, dout, dcost);
    if (!r) r = mgm_img_download(c, dout, out);
    if (!r) r = mgm_img_download(c, dcost, outcost);
    mgm_img_free(c, dout);
    mgm_img_free(c, dcost);
    return r;
}

}  // extern "C"
