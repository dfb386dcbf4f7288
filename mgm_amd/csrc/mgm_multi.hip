// mgm_multi.hip -- several GPUs of ONE node behind the C ABI (include/mgm_hip.h, "mgm_multi_*"): the passes of one
// volume sharded by DIRECTION over the devices, the per-direction Lr volumes summed in pass order after an exchange of
// ROW SLABS over xGMI (BASELINE cfg4).  The CPU analogue in the reference is mgm_naive_parallelism
// (mgm_core.cc:632-831): passes in parallel on private Lr volumes (710-717), then accumulated (798-805) -- there in
// thread-finish order, here always in pass order, which keeps the result bit-identical to mgm().
//
// One host thread drives all devices: every device has its own mgm_ctx (device + stream + workspace) and a second
// stream for the exchange; all calls are asynchronous, ordered by events.  The exchange runs in ROUNDS -- round r moves
// the r-th pass of every rank -- over one of three transports:
//   rccl      ONE grouped set of ncclSend / ncclRecv per round (RCCL keeps one xGMI link per peer busy: an all-to-all,
//             7 x 1/8 of a volume out and in per GPU at 8 GPUs).  An all-reduce is not used: it would fix neither the
//             order of the fp32 additions nor use more than one link at a time (ring);
//   peer      the RECEIVING device pulls each slab with hipMemcpyPeerAsync on its exchange stream once the owner's
//             "pass done" event has fired -- no communicator, no IPC; what a handle falls back to when librccl or its
//             communicator is not to be had (MGM_MULTI_TRANSPORT=peer asks for it);
//   loopback  a device appears several times in the list (MGM_MULTI_LOOPBACK=1, tests): the "ranks" are contexts on one
//             GPU and a slab is a device-to-device copy.  Partition, buffers, ordering and the row-slab WTA are the code
//             that runs on a real node; only the copy call differs from `peer`.
// With MGM_MULTI_OVERLAP=1 a rank that owns several passes launches them one per launch (mgm_aggregate_passes_at_dev)
// and round r is posted right behind pass r, so its slabs travel while pass r+1 runs.
//
// Failure discipline: every buffer is sized BEFORE anything is enqueued; a transfer group is closed on every path that
// opened it (GroupGuard); when a rank's launch fails nothing further is posted, every rank's queued work is drained, and
// only then is the error returned; the final wait polls the streams against MGM_MULTI_TIMEOUT_S and aborts the
// communicators (ncclCommAbort) instead of blocking forever on a peer that never delivers.
//
// librccl is loaded on first use (dlopen), so that single-GPU users of libmgm_hip.so do not carry it; everything here
// is written against the public ABI of mgm_hip.h only.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mgm_hip.h"

namespace {

struct Rccl {  // the entry points used, resolved from librccl.so at run time
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string &err)
    {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) {
            err = std::string("librccl not found: ") + dlerror();
            return false;
        }
        auto sym = [&](const char *n) { return dlsym(lib, n); };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        CommAbort = (decltype(CommAbort))sym("ncclCommAbort");
        CommGetAsyncError = (decltype(CommGetAsyncError))sym("ncclCommGetAsyncError");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) {
            err = "librccl lacks an entry point";
            dlclose(lib);
            lib = nullptr;
            return false;
        }
        return true;
    }
};
Rccl g_rccl;
thread_local std::string g_create_err = "no mgm_multi_create has failed on this thread";

struct Grow {  // grow-only device buffer on one device
    void *p = nullptr;
    size_t cap = 0;
};

enum Transport { T_RCCL = 0, T_PEER = 1, T_LOOPBACK = 2 };
const char *const kTransportName[] = {"rccl", "peer", "loopback"};
constexpr int kRounds = 8;

}  // namespace

struct mgm_multi {
    int n = 0;
    Transport tr = T_RCCL;
    bool dead = false;     // a transfer timed out and the transport was aborted: only mgm_multi_destroy is left
    bool overlap = false;  // MGM_MULTI_OVERLAP=1
    double timeout_s = 120.0;
    std::vector<int> dev;
    std::vector<mgm_ctx *> ctx;
    std::vector<ncclComm_t> comm;
    std::vector<Grow> recv, rows;          // per rank: [NDIR][nrows][nx][L] slabs received; its rows' results (rank 0: + the gather staging)
    std::vector<hipStream_t> xs;           // per rank: the exchange stream
    std::vector<hipEvent_t> pass_done;     // [rank * kRounds + round]: that pass's Lr volume is complete
    std::vector<hipEvent_t> xdone, rows_done;  // per rank: all its slabs have arrived; its rows are searched
    std::string err;
};

namespace {

int fail(mgm_multi *m, int code, const std::string &msg)
{
    if (m) m->err = msg;
    return code;
}
int reserve(mgm_multi *m, int k, Grow &b, size_t bytes)
{
    if (bytes <= b.cap) return MGM_OK;
    if (hipSetDevice(m->dev[k]) != hipSuccess) return fail(m, MGM_ERR_HIP, "hipSetDevice");
    if (b.p) {
        (void)hipStreamSynchronize((hipStream_t)mgm_ctx_stream(m->ctx[k]));
        (void)hipStreamSynchronize(m->xs[k]);
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    if (hipMalloc(&b.p, bytes) != hipSuccess) {
        b.p = nullptr;
        (void)hipGetLastError();  // (the runtime keeps the error until it is read: the next launch wrapper would report it)
        return fail(m, MGM_ERR_NOMEM, "mgm_multi: hipMalloc(" + std::to_string(bytes) + ") on device " + std::to_string(m->dev[k]));
    }
    b.cap = bytes;
    return MGM_OK;
}

struct GroupGuard {  // an RCCL group that is closed on every path out of its scope
    bool open = false;
    bool start()
    {
        open = g_rccl.GroupStart() == ncclSuccess;
        return open;
    }
    ncclResult_t end()
    {
        if (!open) return ncclSuccess;
        open = false;
        return g_rccl.GroupEnd();
    }
    ~GroupGuard()
    {
        if (open) (void)g_rccl.GroupEnd();
    }
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Everything queued on every rank's streams has run (or the deadline passed: false).
bool drain(mgm_multi *m, double deadline, std::string &why)
{
    for (;;) {
        bool busy = false;
        for (int k = 0; k < m->n && !busy; k++) {
            (void)hipSetDevice(m->dev[k]);
            for (hipStream_t s : {(hipStream_t)mgm_ctx_stream(m->ctx[k]), m->xs[k]}) {
                const hipError_t e = hipStreamQuery(s);
                if (e == hipErrorNotReady) busy = true;
                else if (e != hipSuccess) {
                    why = std::string("device ") + std::to_string(m->dev[k]) + ": " + hipGetErrorString(e);
                    return false;
                }
            }
            if (m->tr == T_RCCL && k < (int)m->comm.size() && m->comm[k] && g_rccl.CommGetAsyncError) {
                ncclResult_t ae = ncclSuccess;
                if (g_rccl.CommGetAsyncError(m->comm[k], &ae) == ncclSuccess && ae != ncclSuccess && ae != ncclInProgress) {
                    why = std::string("RCCL asynchronous error on rank ") + std::to_string(k) + ": " + g_rccl.GetErrorString(ae);
                    return false;
                }
            }
        }
        if (!busy) return true;
        if (now_s() > deadline) {
            why = "the exchange did not complete within " + std::to_string((int)m->timeout_s) + " s (MGM_MULTI_TIMEOUT_S)";
            return false;
        }
        usleep(100);
    }
}

void abort_transport(mgm_multi *m)
{
    m->dead = true;
    if (m->tr == T_RCCL && g_rccl.CommAbort)
        for (auto &c : m->comm)
            if (c) {
                (void)g_rccl.CommAbort(c);
                c = nullptr;
            }
}

}  // namespace

extern "C" {

// Contiguous blocks of passes and of rows, one per rank, sizes differing by at most one (the earlier ranks get the
// longer blocks).  Exported as data so that the partition can be checked without a GPU.
int mgm_multi_plan(int n, int NDIR, int ny, int *first_pass, int *n_passes, int *row0, int *nrows)
{
    if (n < 1 || NDIR < 1 || ny < 1 || !first_pass || !n_passes || !row0 || !nrows) return MGM_ERR_INVALID;
    int p = 0, r = 0;
    for (int k = 0; k < n; k++) {
        const int np = NDIR / n + (k < NDIR % n ? 1 : 0), nr = ny / n + (k < ny % n ? 1 : 0);
        first_pass[k] = p;
        n_passes[k] = np;
        row0[k] = r;
        nrows[k] = nr;
        p += np;
        r += nr;
    }
    return MGM_OK;
}

int mgm_multi_create(const int *device_ids, int n, mgm_multi **out)
{
    if (!out) return MGM_ERR_INVALID;
    *out = nullptr;
    if (!device_ids || n < 1 || n > 64) {
        g_create_err = "mgm_multi_create: 1..64 device ids";
        return MGM_ERR_INVALID;
    }
    bool dup = false;
    for (int a = 0; a < n; a++)
        for (int b = 0; b < a; b++) dup |= device_ids[a] == device_ids[b];
    const char *lb = getenv("MGM_MULTI_LOOPBACK");
    if (dup && !(lb && atoi(lb) == 1)) {
        g_create_err = "mgm_multi_create: a device id appears twice (loopback ranks need MGM_MULTI_LOOPBACK=1)";
        return MGM_ERR_INVALID;
    }
    mgm_multi *m = new mgm_multi();
    m->n = n;
    // loopback only where the caller really passed one device several times: distinct devices never take it silently
    const char *tr = getenv("MGM_MULTI_TRANSPORT");
    m->tr = dup ? T_LOOPBACK : ((tr && !strcmp(tr, "peer")) ? T_PEER : T_RCCL);
    if (const char *e = getenv("MGM_MULTI_OVERLAP")) m->overlap = atoi(e) == 1;
    if (const char *e = getenv("MGM_MULTI_TIMEOUT_S")) m->timeout_s = std::max(1.0, atof(e));
    m->dev.assign(device_ids, device_ids + n);
    m->ctx.assign(n, nullptr);
    m->recv.resize(n);
    m->rows.resize(n);
    m->xs.assign(n, nullptr);
    m->pass_done.assign((size_t)n * kRounds, nullptr);
    m->xdone.assign(n, nullptr);
    m->rows_done.assign(n, nullptr);
    int r = MGM_OK;
    std::string why;
    for (int k = 0; k < n && !r; k++) {
        r = mgm_ctx_create(device_ids[k], &m->ctx[k]);
        if (r) {
            why = "mgm_ctx_create(" + std::to_string(device_ids[k]) + ") failed: no usable gfx950 device of that ordinal";
            break;
        }
        bool ok = hipSetDevice(device_ids[k]) == hipSuccess && hipStreamCreateWithFlags(&m->xs[k], hipStreamNonBlocking) == hipSuccess;
        for (int q = 0; q < kRounds && ok; q++) ok = hipEventCreateWithFlags(&m->pass_done[(size_t)k * kRounds + q], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&m->xdone[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&m->rows_done[k], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            r = MGM_ERR_HIP;
            why = "stream / event creation on device " + std::to_string(device_ids[k]);
        }
    }
    if (!r && m->tr == T_RCCL) {
        std::string e;
        if (!g_rccl.load(e)) {
            m->tr = T_PEER;  // degrade: the exchange does not need a communicator
            m->err = "transport peer (" + e + ")";
        } else {
            m->comm.assign(n, nullptr);
            const ncclResult_t ne = g_rccl.CommInitAll(m->comm.data(), n, device_ids);
            if (ne != ncclSuccess) {
                m->comm.clear();
                m->tr = T_PEER;
                m->err = std::string("transport peer (ncclCommInitAll: ") + g_rccl.GetErrorString(ne) + ")";
            }
        }
    }
    if (!r && m->tr == T_PEER && n > 1)  // direct copies over xGMI where the devices can see each other (else the runtime stages them)
        for (int a = 0; a < n; a++)
            for (int b = 0; b < n; b++) {
                int can = 0;
                if (a == b || hipDeviceCanAccessPeer(&can, device_ids[a], device_ids[b]) != hipSuccess || !can) continue;
                if (hipSetDevice(device_ids[a]) == hipSuccess) {
                    const hipError_t pe = hipDeviceEnablePeerAccess(device_ids[b], 0);
                    if (pe != hipSuccess) (void)hipGetLastError();  // (already enabled: fine)
                }
            }
    if (r) {
        g_create_err = why;
        mgm_multi_destroy(m);
        (void)hipGetLastError();  // (leave no stale HIP error behind for the caller's next launch to trip over)
        return r;
    }
    *out = m;
    return MGM_OK;
}

int mgm_multi_destroy(mgm_multi *m)
{
    if (!m) return MGM_OK;
    for (int k = 0; k < m->n; k++) {
        if (!m->ctx[k]) continue;  // (a create that failed at this rank: nothing of it exists, and its device may not either)
        (void)hipSetDevice(m->dev[k]);
        if (!m->dead) {
            if (m->ctx[k]) (void)mgm_ctx_synchronize(m->ctx[k]);
            if (m->xs[k]) (void)hipStreamSynchronize(m->xs[k]);
        }
        if (k < (int)m->comm.size() && m->comm[k]) (void)g_rccl.CommDestroy(m->comm[k]);
        if (m->recv[k].p) (void)hipFree(m->recv[k].p);
        if (m->rows[k].p) (void)hipFree(m->rows[k].p);
        for (int q = 0; q < kRounds; q++)
            if (m->pass_done[(size_t)k * kRounds + q]) (void)hipEventDestroy(m->pass_done[(size_t)k * kRounds + q]);
        if (m->xdone[k]) (void)hipEventDestroy(m->xdone[k]);
        if (m->rows_done[k]) (void)hipEventDestroy(m->rows_done[k]);
        if (m->xs[k]) (void)hipStreamDestroy(m->xs[k]);
        if (m->ctx[k] && !m->dead) (void)mgm_ctx_destroy(m->ctx[k]);  // (a dead handle's contexts may still have a transfer pending: leaked)
    }
    delete m;
    return MGM_OK;
}

int mgm_multi_size(const mgm_multi *m) { return m ? m->n : 0; }
mgm_ctx *mgm_multi_ctx(mgm_multi *m, int rank) { return (m && rank >= 0 && rank < m->n) ? m->ctx[rank] : nullptr; }
const char *mgm_multi_last_error(const mgm_multi *m) { return m ? m->err.c_str() : g_create_err.c_str(); }
const char *mgm_multi_transport(const mgm_multi *m) { return m ? kTransportName[m->tr] : ""; }

int mgm_multi_aggregate(mgm_multi *m, const mgm_cv *const *C, const mgm_img *const *w8, float P1, float P2, int NDIR, int MGM,
                        int use_fh, int fix_overcount, const char *refine, mgm_img *out0, mgm_img *outcost0)
{
    if (!m || !C || !out0 || !outcost0) return fail(m, MGM_ERR_INVALID, "mgm_multi_aggregate: null argument");
    if (m->dead) return fail(m, MGM_ERR_HIP, "mgm_multi_aggregate: the transport of this handle was aborted after a time-out");
    if (NDIR < 1 || NDIR > 8) return fail(m, MGM_ERR_INVALID, "NDIR must be 1..8");
    const int n = m->n;
    int nx = 0, ny = 0, dmin = 0, dmax = 0;
    for (int k = 0; k < n; k++) {
        int a, b, c, d;
        if (!C[k] || mgm_cv_dims(C[k], &a, &b, &c, &d)) return fail(m, MGM_ERR_INVALID, "mgm_multi_aggregate: one cost volume per device");
        if (k == 0) nx = a, ny = b, dmin = c, dmax = d;
        else if (a != nx || b != ny || c != dmin || d != dmax) return fail(m, MGM_ERR_INVALID, "mgm_multi_aggregate: the devices' cost volumes differ in geometry");
        // a volume or weight image on another rank's device would fault in that rank's kernels: refuse it here
        if (mgm_cv_device(C[k]) != m->dev[k]) return fail(m, MGM_ERR_INVALID, "mgm_multi_aggregate: C[" + std::to_string(k) + "] does not live on device " + std::to_string(m->dev[k]));
        if (w8 && w8[k] && mgm_img_device(w8[k]) != m->dev[k])
            return fail(m, MGM_ERR_INVALID, "mgm_multi_aggregate: w8[" + std::to_string(k) + "] does not live on device " + std::to_string(m->dev[k]));
    }
    {
        int a, b, c;
        if (mgm_img_dims(out0, &a, &b, &c) || a != nx || b != ny || mgm_img_dims(outcost0, &a, &b, &c) || a != nx || b != ny)
            return fail(m, MGM_ERR_INVALID, "mgm_multi_aggregate: output image size mismatch");
        if (mgm_img_device(out0) != m->dev[0] || mgm_img_device(outcost0) != m->dev[0])
            return fail(m, MGM_ERR_INVALID, "mgm_multi_aggregate: out0 / outcost0 must live on device_ids[0]");
    }
    const size_t L = (size_t)(dmax - dmin + 1), rowf = (size_t)nx * L;  // floats per image row of a volume
    std::vector<int> first(n), cnt(n), row0(n), nrows(n);
    mgm_multi_plan(n, NDIR, ny, first.data(), cnt.data(), row0.data(), nrows.data());
    int rounds = 0;
    for (int k = 0; k < n; k++) rounds = std::max(rounds, cnt[k]);
    const bool per_pass = m->overlap && rounds > 1;  // one launch per pass, each round posted behind its pass
    int r;
    // 0. every buffer, before anything is enqueued: all passes of this device's rows in pass order; its rows' two result
    //    images -- on rank 0 also the staging area of the RCCL gather
    size_t staging = 0;
    for (int k = 1; k < n; k++) staging += 2 * (size_t)nrows[k] * nx;
    for (int k = 0; k < n; k++) {
        if ((r = reserve(m, k, m->recv[k], sizeof(float) * (size_t)NDIR * std::max(nrows[k], 1) * rowf))) return r;
        const size_t own = 2 * (size_t)std::max(nrows[k], 1) * nx;
        if ((r = reserve(m, k, m->rows[k], sizeof(float) * (k == 0 ? std::max(own, std::max<size_t>(staging, 1)) : own)))) return r;
    }
    auto ctx_stream = [&](int k) { return (hipStream_t)mgm_ctx_stream(m->ctx[k]); };
    auto lr_of = [&](int k, int p) { return (const float *)mgm_lr_device_ptr(m->ctx[k], p - first[k]); };
    auto ev_of = [&](int k, int round) { return m->pass_done[(size_t)k * kRounds + (per_pass ? round : 0)]; };
    int rc = MGM_OK;      // first failure; nothing further is posted once it is set, and it is returned after the drain
    std::string rc_msg;
    auto hipbad = [&](const char *what) {
        if (!rc) rc = MGM_ERR_HIP, rc_msg = std::string(what) + ": " + hipGetErrorString(hipGetLastError());
    };

    // 1. + 3. passes and exchange, round by round
    auto launch_round = [&](int round) {  // every rank's pass(es) of this round (all of them when !per_pass)
        for (int k = 0; k < n && !rc; k++) {
            if (cnt[k] <= round) continue;  // (more devices than passes: such a rank only sums and searches its rows)
            int e = per_pass ? mgm_aggregate_passes_at_dev(m->ctx[k], C[k], w8 ? w8[k] : nullptr, P1, P2, MGM, use_fh, first[k] + round, 1, round, cnt[k], NDIR)
                             : mgm_aggregate_passes_dev(m->ctx[k], C[k], w8 ? w8[k] : nullptr, P1, P2, MGM, use_fh, first[k], cnt[k]);
            if (e) {
                rc = e;
                rc_msg = std::string("device ") + std::to_string(m->dev[k]) + ": " + mgm_last_error(m->ctx[k]);
                return;
            }
            if (hipSetDevice(m->dev[k]) != hipSuccess || hipEventRecord(ev_of(k, round), ctx_stream(k)) != hipSuccess) hipbad("hipEventRecord (pass done)");
        }
    };
    auto post_round = [&](int round) {  // the slabs of every rank's `round`-th pass go to the rows' owners
        if (rc) return;
        for (int k = 0; k < n; k++)
            if (cnt[k] > round && !lr_of(k, first[k] + round)) {
                rc = MGM_ERR_INTERNAL, rc_msg = "mgm_multi_aggregate: no Lr volume (padded label count?)";
                return;
            }
        GroupGuard grp;
        if (m->tr == T_RCCL && n > 1 && !grp.start()) {
            rc = MGM_ERR_HIP, rc_msg = "ncclGroupStart";
            return;
        }
        for (int k = 0; k < n && !rc; k++) {
            if (hipSetDevice(m->dev[k]) != hipSuccess) {
                hipbad("hipSetDevice");
                break;
            }
            hipStream_t xk = m->xs[k];
            float *rk = (float *)m->recv[k].p;
            const size_t slab_k = (size_t)nrows[k] * rowf;
            // (the exchange stream of a rank that sends this round starts behind its pass)
            if (cnt[k] > round && hipStreamWaitEvent(xk, ev_of(k, round), 0) != hipSuccess) hipbad("hipStreamWaitEvent (own pass)");
            for (int g = 0; g < n && !rc; g++) {  // what lands in rank k's buffer: rank g's pass of this round
                if (cnt[g] <= round || !nrows[k]) continue;
                const int p = first[g] + round;
                float *dst = rk + (size_t)p * slab_k;
                const float *src = lr_of(g, p) + (size_t)row0[k] * rowf;
                if (g == k) {
                    if (hipMemcpyAsync(dst, src, sizeof(float) * slab_k, hipMemcpyDeviceToDevice, xk) != hipSuccess) hipbad("hipMemcpyAsync (own slab)");
                } else if (m->tr == T_RCCL) {
                    const ncclResult_t e = g_rccl.Recv(dst, slab_k, ncclFloat, g, m->comm[k], xk);
                    if (e != ncclSuccess && !rc) rc = MGM_ERR_HIP, rc_msg = std::string("ncclRecv: ") + g_rccl.GetErrorString(e);
                } else {  // pull it: ordered behind the owner's pass by its event
                    hipError_t e = hipStreamWaitEvent(xk, ev_of(g, round), 0);
                    if (e == hipSuccess)
                        e = m->tr == T_PEER ? hipMemcpyPeerAsync(dst, m->dev[k], src, m->dev[g], sizeof(float) * slab_k, xk)
                                            : hipMemcpyAsync(dst, src, sizeof(float) * slab_k, hipMemcpyDeviceToDevice, xk);
                    if (e != hipSuccess) hipbad("slab copy");
                }
            }
            if (m->tr == T_RCCL && cnt[k] > round)  // what rank k sends
                for (int g = 0; g < n && !rc; g++) {
                    if (g == k || !nrows[g]) continue;
                    const ncclResult_t e = g_rccl.Send(lr_of(k, first[k] + round) + (size_t)row0[g] * rowf, (size_t)nrows[g] * rowf, ncclFloat, g, m->comm[k], xk);
                    if (e != ncclSuccess) rc = MGM_ERR_HIP, rc_msg = std::string("ncclSend: ") + g_rccl.GetErrorString(e);
                }
        }
        const ncclResult_t ge = grp.end();  // (closed whatever happened inside)
        if (ge != ncclSuccess && !rc) rc = MGM_ERR_HIP, rc_msg = std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(ge);
    };
    if (per_pass) {
        for (int round = 0; round < rounds && !rc; round++) {
            launch_round(round);
            post_round(round);
        }
    } else {
        launch_round(0);
        for (int round = 0; round < rounds && !rc; round++) post_round(round);
    }
    // 4. every device finishes its rows: ordered sum over the passes, over-count fix, WTA, refinement -- behind its slabs
    float *o0 = (float *)mgm_img_device_ptr(out0), *c0 = (float *)mgm_img_device_ptr(outcost0);
    for (int k = 0; k < n && !rc; k++) {
        if (hipSetDevice(m->dev[k]) != hipSuccess || hipEventRecord(m->xdone[k], m->xs[k]) != hipSuccess ||
            hipStreamWaitEvent(ctx_stream(k), m->xdone[k], 0) != hipSuccess) {
            hipbad("exchange-done event");
            break;
        }
        if (!nrows[k]) continue;
        float *ok = k == 0 ? o0 + (size_t)row0[0] * nx : (float *)m->rows[k].p;
        float *ck = k == 0 ? c0 + (size_t)row0[0] * nx : (float *)m->rows[k].p + (size_t)nrows[k] * nx;
        if ((r = mgm_wta_rows_dev(m->ctx[k], C[k], row0[k], nrows[k], m->recv[k].p, NDIR, fix_overcount, refine, ok, ck))) {
            rc = r, rc_msg = std::string("device ") + std::to_string(m->dev[k]) + ": " + mgm_last_error(m->ctx[k]);
            break;
        }
        if (k > 0 && hipEventRecord(m->rows_done[k], ctx_stream(k)) != hipSuccess) hipbad("hipEventRecord (rows)");
    }
    // 5. the rows travel to device 0 (two W x nrows images per device)
    if (n > 1 && !rc) {
        hipStream_t s0 = ctx_stream(0);
        if (m->tr != T_RCCL) {
            for (int k = 1; k < n && !rc; k++) {
                if (!nrows[k]) continue;
                const size_t cntk = (size_t)nrows[k] * nx;
                const float *src = (const float *)m->rows[k].p;
                hipError_t e = hipSetDevice(m->dev[0]);
                if (e == hipSuccess) e = hipStreamWaitEvent(s0, m->rows_done[k], 0);
                if (e == hipSuccess)
                    e = m->tr == T_PEER ? hipMemcpyPeerAsync(o0 + (size_t)row0[k] * nx, m->dev[0], src, m->dev[k], sizeof(float) * cntk, s0)
                                        : hipMemcpyAsync(o0 + (size_t)row0[k] * nx, src, sizeof(float) * cntk, hipMemcpyDeviceToDevice, s0);
                if (e == hipSuccess)
                    e = m->tr == T_PEER ? hipMemcpyPeerAsync(c0 + (size_t)row0[k] * nx, m->dev[0], src + cntk, m->dev[k], sizeof(float) * cntk, s0)
                                        : hipMemcpyAsync(c0 + (size_t)row0[k] * nx, src + cntk, sizeof(float) * cntk, hipMemcpyDeviceToDevice, s0);
                if (e != hipSuccess) hipbad("row copy");
            }
        } else {
            // device 0 receives each rank's two images back to back into its staging area (sized in step 0), then places them
            GroupGuard grp;
            if (!grp.start()) rc = MGM_ERR_HIP, rc_msg = "ncclGroupStart (rows)";
            float *st = (float *)m->rows[0].p;
            size_t off = 0;
            for (int k = 1; k < n && !rc; k++) {
                if (!nrows[k]) continue;
                const size_t cntk = (size_t)nrows[k] * nx;
                if (hipSetDevice(m->dev[k]) != hipSuccess) hipbad("hipSetDevice");
                else if (g_rccl.Send(m->rows[k].p, 2 * cntk, ncclFloat, 0, m->comm[k], ctx_stream(k)) != ncclSuccess) rc = MGM_ERR_HIP, rc_msg = "ncclSend (rows)";
                if (!rc && hipSetDevice(m->dev[0]) != hipSuccess) hipbad("hipSetDevice");
                else if (!rc && g_rccl.Recv(st + off, 2 * cntk, ncclFloat, k, m->comm[0], s0) != ncclSuccess) rc = MGM_ERR_HIP, rc_msg = "ncclRecv (rows)";
                off += 2 * cntk;
            }
            const ncclResult_t ge = grp.end();
            if (ge != ncclSuccess && !rc) rc = MGM_ERR_HIP, rc_msg = std::string("ncclGroupEnd (rows): ") + g_rccl.GetErrorString(ge);
            off = 0;
            if (!rc && hipSetDevice(m->dev[0]) != hipSuccess) hipbad("hipSetDevice");
            for (int k = 1; k < n && !rc; k++) {
                if (!nrows[k]) continue;
                const size_t cntk = (size_t)nrows[k] * nx;
                if (hipMemcpyAsync(o0 + (size_t)row0[k] * nx, st + off, sizeof(float) * cntk, hipMemcpyDeviceToDevice, s0) != hipSuccess ||
                    hipMemcpyAsync(c0 + (size_t)row0[k] * nx, st + off + cntk, sizeof(float) * cntk, hipMemcpyDeviceToDevice, s0) != hipSuccess)
                    hipbad("hipMemcpyAsync (rows)");
                off += 2 * cntk;
            }
        }
    }
    // 6. everything has been enqueued (or a failure stopped the posting: whatever was posted is complete on both sides,
    //    one host thread posts for every rank).  Wait for all of it against the deadline; the call returns when the result
    //    is on device 0, no watchdog fired anywhere, and no rank's stream has anything left that touches a buffer.
    std::string why;
    if (!drain(m, now_s() + m->timeout_s, why)) {
        abort_transport(m);
        return fail(m, MGM_ERR_HIP, "mgm_multi_aggregate: " + why + (rc ? " (after: " + rc_msg + ")" : ""));
    }
    for (int k = n - 1; k >= 0; k--)
        if ((r = mgm_ctx_synchronize(m->ctx[k])) && !rc) rc = r, rc_msg = std::string("device ") + std::to_string(m->dev[k]) + ": " + mgm_last_error(m->ctx[k]);
    if (rc) return fail(m, rc, rc_msg);
    return MGM_OK;
}

}  // extern "C"
