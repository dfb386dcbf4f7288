// mgm_ctx.hip -- contexts, device containers (images, volumes and their compact copies), timing: the part of the C ABI of
// libmgm_hip.so (include/mgm_hip.h) that owns memory.  See mgm_host.h.
#include "mgm_host.h"

unsigned long long next_cv_generation()
{
    static unsigned long long g = 0;
    return ++g;
}

// ---- development switches: ONE place, ONE variable -------------------------------------------------------------------------
// MGM_HIP_TUNE="key=value,key=value,..." -- or, as in rounds 1-3 (tests and tools still use them), the individual variables
// MGM_HIP_<KEY>.  Read once per process.  -DMGM_HIP_RELEASE compiles all of it out: every switch keeps its default.
long long mgm::tune_num(const char *key, long long dflt)
{
#ifdef MGM_HIP_RELEASE
    (void)key;
    return dflt;
#else
    static const std::vector<std::pair<std::string, long long>> table = [] {
        std::vector<std::pair<std::string, long long>> t;
        if (const char *e = getenv("MGM_HIP_TUNE")) {
            std::string s(e);
            size_t i = 0;
            while (i < s.size()) {
                size_t j = s.find(',', i);
                if (j == std::string::npos) j = s.size();
                const std::string item = s.substr(i, j - i);
                const size_t q = item.find('=');
                if (q != std::string::npos && q > 0) t.emplace_back(item.substr(0, q), atoll(item.c_str() + q + 1));
                i = j + 1;
            }
        }
        return t;
    }();
    for (const auto &kv : table)
        if (kv.first == key) return kv.second;
    std::string legacy = "MGM_HIP_";
    for (const char *q = key; *q; q++) legacy += (char)toupper((unsigned char)*q);
    if (const char *e = getenv(legacy.c_str())) return atoll(e);
    return dflt;
#endif
}

const DevSwitches &dev()
{
    static const DevSwitches d = [] {
        auto on = [](const char *n) { return tune_num(n, 1) != 0; };
        return DevSwitches{on("c8"), on("lazy_f32"), on("pad"), (int)tune_num("subv", 1), (int)tune_num("deep", -1),
                           (int)tune_num("wg_per_cu", 0), (int)tune_num("xflags", 0), (int)tune_num("strips", -1), (int)tune_num("xcdq", -1),
                           (int)tune_num("xcdq_k", -1), on("w2"), on("oneb"), 64ll * tune_num("lr_pad", 67)};
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
    // (round 6) the runtime keeps its last error until somebody reads it, and every launch wrapper ends with `return hipGetLastError()`:
    // an error that was returned DIRECTLY by a call (hipFuncSetAttribute refusing an LDS request, ...) would otherwise be reported once
    // more by the next, innocent, launch of the process
    (void)hipGetLastError();
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
    g.diag = 0;
    g.wmax = 0;
    g.swap = 0;
    return true;
}



// The watchdog word of the pass kernel is STICKY on the device: no launch resets it, a copy of it follows every pass
// launch into h_words[1], and only the host clears it, once it has seen it set.  A hand-off time-out of one launch is
// therefore reported by whichever call next finds the stream idle (block = false: pass launches look without waiting --
// nothing on the hot path synchronises for it) or synchronises anyway (block = true), and cannot be overwritten by a
// later launch's copy.
int check_watchdog(mgm_ctx *c, bool block)
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


bool pipe_uses(const mgm_ctx *c, const void *obj)  // is `obj` (a volume or an image) an operand of a deferred call?
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
    c->force_build = (int)tune_num("pass_build", 0);
    c->debug_stats = (int)tune_num("debug_stats", 0);
    *out = c;
    return MGM_OK;
}

int mgm_ctx_destroy(mgm_ctx *c)
{
    if (!c) return MGM_OK;
    (void)hipSetDevice(c->device);
    (void)pipe_join(c);  // (deferred calls of a pipelined context still write the caller's images)
    (void)hipStreamSynchronize(c->stream);
    std::vector<Buf *> bufs = {&c->lr, &c->hand, &c->hand2, &c->handm, &c->exact_mins, &c->exact_scratch, &c->words, &c->census_u, &c->census_v, &c->dbg, &c->stmp, &c->ones8,
                               &c->lr_rel, &c->hand_rel, &c->tasks_rel};  // (the range-proportional kernels' workspace: round 5 forgot it here)
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
    std::vector<Buf *> bufs = {&c->lr, &c->hand, &c->hand2, &c->handm, &c->exact_mins, &c->exact_scratch, &c->census_u, &c->census_v, &c->dbg, &c->stmp, &c->ones8,
                               &c->lr_rel, &c->hand_rel, &c->tasks_rel};
    c->tasks_rel_key.clear();
    c->hand_rel_key.clear();
    c->rel_last_batch = 0;
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

int mgm_ctx_set_placement_tries(mgm_ctx *c, int tries)
{
    if (!c || tries < 0 || tries > 8) return MGM_ERR_INVALID;
    c->place_tries = tries;
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
// Refill an existing image from the host (same size): no allocation, what a caller with a stream of same-sized inputs wants.
int mgm_img_update(mgm_ctx *c, mgm_img *im, const float *host)
{
    if (int jr = pipe_join(c)) return jr;  // (pipelined context: a deferred call may still read or write the image)
    if (!c || !im || !host) return fail(c, MGM_ERR_INVALID, "mgm_img_update: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(im->d, host, sizeof(float) * (size_t)im->nx * im->ny * im->nch, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // (the host buffer is the caller's again on return)
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
extern "C++" int cv_alloc_f32(mgm_ctx *c, mgm_cv *cv)
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
extern "C++" int cv_create(mgm_ctx *c, int nx, int ny, int dmin, int dmax, bool alloc_f32, mgm_cv **out)
{
    if (!c || !out || nx <= 0 || ny <= 0 || dmax < dmin) return fail(c, MGM_ERR_INVALID, "mgm_cv_create: bad arguments");
    const long long L = (long long)dmax - dmin + 1;
    if (L > kMaxLabels)
        return fail(c, MGM_ERR_UNSUPPORTED, "more than 4 194 304 disparity labels per pixel are not supported");
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
    (*out)->p8_state = 0;
    (*out)->nan_state = 0;
    return MGM_OK;
}
// make cv->d current (see mgm_cv::f32_state); enqueued on the context's stream
extern "C++" int ensure_f32(mgm_ctx *c, const mgm_cv *ccv)
{
    mgm_cv *cv = const_cast<mgm_cv *>(ccv);
    if (cv->f32_state) return MGM_OK;
    if (cv->rel_only && cv->rel_state == 2 && cv->relbuf) {  // K2 wrote the range-proportional copy alone (ragged census volume)
        if (int r = cv_alloc_f32(c, cv)) return r;
        const long long npix = (long long)cv->nx * cv->ny;
        TimeScope t(c, "k_expand");
        HIPCHK(c, launch_rel_expand(cv->relbuf, cv->rel_records(), npix, cv->dmax - cv->dmin + 1, cv->dmin, cv->rel_slots, cv->rel_cb, cv->d, c->stream));
        cv->f32_state = 1;
        return MGM_OK;
    }
    if (cv->p8_state == 2) {  // K2 wrote the padded compact copy alone
        if (int r = cv_alloc_f32(c, cv)) return r;
        TimeScope t(c, "k_expand");
        HIPCHK(c, launch_expand_padded(cv->p8, cv->p8_cb, (long long)cv->nx * cv->ny, cv->dmax - cv->dmin + 1, cv->p8_L, cv->d, c->stream));
        cv->f32_state = 1;
        return MGM_OK;
    }
    if (!cv->d8 || cv->c8_state < 1) return fail(c, MGM_ERR_INTERNAL, "cost volume has neither an fp32 nor a compact copy");
    if (int r = cv_alloc_f32(c, cv)) return r;
    TimeScope t(c, "k_expand");
    HIPCHK(c, launch_expand(cv->d8, cv->cbytes, (long long)cv->nx * cv->ny * (cv->dmax - cv->dmin + 1), cv->d, c->stream));
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
    cv->rel_only = false;  // (and the range-proportional copy no longer stands for the volume)
    cv->rel_state = cv->rel_state == 2 || cv->rel_state == 1 ? 0 : cv->rel_state;
    cv->p8_state = 0;
    cv->nan_state = 0;
    cv->gen = next_cv_generation();
    return cv->d;
}
int mgm_cv_free(mgm_ctx *c, mgm_cv *cv)
{
    if (int jr = pipe_join(c)) return jr;  // (pipelined context: run what has been deferred first)
    if (!cv) return MGM_OK;
    // (freed through another context, or with none: the context that made the volume may still hold deferred calls on it)
    if (cv->owner && cv->owner != c && pipe_uses(cv->owner, cv)) (void)pipe_join(cv->owner);
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
    if (cv->p8) (void)hipFree(cv->p8);
    if (cv->bad8) (void)hipFree(cv->bad8);
    if (cv->relbuf) (void)hipFree(cv->relbuf);
    for (mgm_ctx *o : {c, cv->owner})
        if (o)
            for (int v = 0; v < kMaxBatch; v++)
                if (o->rel_last_cvs[v] == cv) o->rel_last_cvs[v] = nullptr;
    if (cv->rlo) (void)hipFree(cv->rlo);
    if (cv->rhi) (void)hipFree(cv->rhi);
    delete cv;
    return MGM_OK;
}

}  // extern "C"

// ---- compact costs -----------------------------------------------------------
int c8_alloc(mgm_ctx *c, mgm_cv *cv, int cb)
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
// room for a padded compact copy of LP label slots, cb bytes each (mgm_cv::p8)
int p8_alloc(mgm_ctx *c, mgm_cv *cv, int LP, int cb)
{
    const size_t n = (size_t)cv->nx * cv->ny * (size_t)LP * cb + 64;
    if (cv->p8 && cv->p8_cap < n) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        (void)hipFree(cv->p8);
        cv->p8 = nullptr;
    }
    if (!cv->p8) {
        if (dev_malloc((void **)&cv->p8, n) != hipSuccess) {
            cv->p8 = nullptr;
            cv->p8_cap = 0;
            return fail(c, MGM_ERR_NOMEM, "hipMalloc of the padded compact cost volume failed");
        }
        cv->p8_cap = n;
    }
    cv->p8_L = LP;
    cv->p8_cb = cb;
    return MGM_OK;
}
// Decide (once per filling of the volume) whether the compact copy can stand in for C, and whether the volume
// holds NaN costs (mgm_cv::nan_state).  Costs one 4-byte device->host read per filling; MGM_HIP_C8=0 disables the
// compact path.  An uploaded volume is scanned here: by k_compact where it gets a compact copy, else by k_nanscan.
int c8_resolve(mgm_ctx *c, const mgm_cv *ccv, bool *use)
{
    mgm_cv *cv = const_cast<mgm_cv *>(ccv);
    *use = false;
    const int L = cv->dmax - cv->dmin + 1;
    const bool enabled = dev().c8 && c8_supported(L);
    const long long n = (long long)cv->nx * cv->ny * L;
    bool launched = false;
    if (enabled && cv->c8_state == 0) {  // uploaded / externally written volume: make the compact copy now (one byte per cost)
        int r = ensure_f32(c, cv);  // (a ragged census volume that only has its range-proportional copy)
        if (r) return r;
        r = c8_alloc(c, cv, 1);
        if (r) return r;
        HIPCHK(c, hipMemsetAsync(cv->bad8, 0, 4, c->stream));
        TimeScope t(c, "k_compact");
        HIPCHK(c, launch_compact(cv->d, n, cv->d8, 1, cv->bad8, c->stream));
        cv->c8_state = 1;
        cv->nan_state = 1;
        launched = true;
    }
    if (cv->nan_state == 0) {
        if (int r = ensure_f32(c, cv)) return r;
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
        // an uploaded volume of whole numbers beyond 254 (absolute differences of a colour pair computed elsewhere): the
        // two-byte form, where the pass kernels read it (up to 512 labels) -- k_compact says whether it would fit
        if (launched && cv->c8_state < 0 && cv->nan_state == 2 && !(c->h_words[3] & 8u) && L <= 512 && cv->f32_state) {
            if (int r = c8_alloc(c, cv, 2)) return r;
            HIPCHK(c, hipMemsetAsync(cv->bad8, 0, 4, c->stream));
            {
                TimeScope t(c, "k_compact");
                HIPCHK(c, launch_compact(cv->d, n, cv->d8, 2, cv->bad8, c->stream));
            }
            HIPCHK(c, hipMemcpyAsync(c->h_words + 3, cv->bad8, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            cv->c8_state = (c->h_words[3] & 1u) ? -1 : 2;
        }
        if (cv->c8_state < 0 && !cv->f32_state)
            return fail(c, MGM_ERR_INTERNAL, "cost volume predicted to fit the compact form does not");
    }
    *use = enabled && cv->c8_state == 2;
    return MGM_OK;
}
