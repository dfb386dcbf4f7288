// mgm_multi.hip -- several GPUs of ONE node behind the C ABI (include/mgm_hip.h, "mgm_multi_*"): the passes of one
// volume sharded by DIRECTION over the devices, the per-direction Lr volumes summed in pass order after an exchange of
// ROW SLABS over xGMI (BASELINE cfg4).  The CPU analogue in the reference is mgm_naive_parallelism
// (mgm_core.cc:632-831): passes in parallel on private Lr volumes (710-717), then accumulated (798-805) -- there in
// thread-finish order, here always in pass order, which keeps the result bit-identical to mgm().
//
// One host thread drives all devices: every device has its own mgm_ctx (device + stream + workspace), all calls are
// asynchronous on those streams, and the exchange is ONE grouped set of ncclSend / ncclRecv per device (RCCL keeps one
// xGMI link per peer busy: an all-to-all, 7 x 1/8 of a volume out and in per GPU at 8 GPUs).  An all-reduce is not
// used: it would fix neither the order of the fp32 additions nor use more than one link at a time (ring).
//
// librccl is loaded on first use (dlopen), so that single-GPU users of libmgm_hip.so do not carry it; everything here
// is written against the public ABI of mgm_hip.h only.
//
// Loopback mode (MGM_MULTI_LOOPBACK=1, tests): the same device may appear several times in the list; the "ranks" are
// then contexts on one GPU and a transfer is a device-to-device copy ordered by events instead of a send/recv pair.
// Partition, buffers, ordering and the row-slab WTA are the code that runs on a real node; only the transport differs.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
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
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) {
            err = "librccl lacks an entry point";
            return false;
        }
        return true;
    }
};
Rccl g_rccl;

struct Grow {  // grow-only device buffer on one device
    void *p = nullptr;
    size_t cap = 0;
};

}  // namespace

struct mgm_multi {
    int n = 0;
    bool loopback = false;
    std::vector<int> dev;
    std::vector<mgm_ctx *> ctx;
    std::vector<ncclComm_t> comm;
    std::vector<Grow> recv, rows;        // per rank: [NDIR][nrows][nx][L] slabs received; [2][nrows][nx] results of its rows
    std::vector<hipEvent_t> passes_done;  // per rank: its Lr volumes are complete (loopback transport)
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
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    if (hipMalloc(&b.p, bytes) != hipSuccess) {
        b.p = nullptr;
        return fail(m, MGM_ERR_NOMEM, "mgm_multi: hipMalloc(" + std::to_string(bytes) + ") on device " + std::to_string(m->dev[k]));
    }
    b.cap = bytes;
    return MGM_OK;
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
    if (!device_ids || n < 1 || n > 64) return MGM_ERR_INVALID;
    bool dup = false;
    for (int a = 0; a < n; a++)
        for (int b = 0; b < a; b++) dup |= device_ids[a] == device_ids[b];
    const char *lb = getenv("MGM_MULTI_LOOPBACK");
    if (dup && !(lb && atoi(lb) == 1)) return MGM_ERR_INVALID;  // RCCL needs distinct devices
    mgm_multi *m = new mgm_multi();
    m->n = n;
    m->loopback = dup || (lb && atoi(lb) == 1);
    m->dev.assign(device_ids, device_ids + n);
    m->ctx.assign(n, nullptr);
    m->recv.resize(n);
    m->rows.resize(n);
    m->passes_done.assign(n, nullptr);
    int r = MGM_OK;
    for (int k = 0; k < n && !r; k++) {
        r = mgm_ctx_create(device_ids[k], &m->ctx[k]);
        if (!r && hipEventCreateWithFlags(&m->passes_done[k], hipEventDisableTiming) != hipSuccess) r = MGM_ERR_HIP;
    }
    if (!r && !m->loopback) {
        std::string e;
        if (!g_rccl.load(e)) r = MGM_ERR_HIP;
        else {
            m->comm.assign(n, nullptr);
            if (g_rccl.CommInitAll(m->comm.data(), n, device_ids) != ncclSuccess) {
                m->comm.clear();
                r = MGM_ERR_HIP;
            }
        }
    }
    if (r) {
        mgm_multi_destroy(m);
        return r;
    }
    *out = m;
    return MGM_OK;
}

int mgm_multi_destroy(mgm_multi *m)
{
    if (!m) return MGM_OK;
    for (int k = 0; k < m->n; k++) {
        (void)hipSetDevice(m->dev[k]);
        if (m->ctx[k]) (void)mgm_ctx_synchronize(m->ctx[k]);
        if (k < (int)m->comm.size() && m->comm[k]) (void)g_rccl.CommDestroy(m->comm[k]);
        if (m->recv[k].p) (void)hipFree(m->recv[k].p);
        if (m->rows[k].p) (void)hipFree(m->rows[k].p);
        if (m->passes_done[k]) (void)hipEventDestroy(m->passes_done[k]);
        if (m->ctx[k]) (void)mgm_ctx_destroy(m->ctx[k]);
    }
    delete m;
    return MGM_OK;
}

int mgm_multi_size(const mgm_multi *m) { return m ? m->n : 0; }
mgm_ctx *mgm_multi_ctx(mgm_multi *m, int rank) { return (m && rank >= 0 && rank < m->n) ? m->ctx[rank] : nullptr; }
const char *mgm_multi_last_error(const mgm_multi *m) { return m ? m->err.c_str() : "null handle"; }

int mgm_multi_aggregate(mgm_multi *m, const mgm_cv *const *C, const mgm_img *const *w8, float P1, float P2, int NDIR, int MGM,
                        int use_fh, int fix_overcount, const char *refine, mgm_img *out0, mgm_img *outcost0)
{
    if (!m || !C || !out0 || !outcost0) return fail(m, MGM_ERR_INVALID, "mgm_multi_aggregate: null argument");
    if (NDIR < 1 || NDIR > 8) return fail(m, MGM_ERR_INVALID, "NDIR must be 1..8");
    const int n = m->n;
    int nx = 0, ny = 0, dmin = 0, dmax = 0;
    for (int k = 0; k < n; k++) {
        int a, b, c, d;
        if (!C[k] || mgm_cv_dims(C[k], &a, &b, &c, &d)) return fail(m, MGM_ERR_INVALID, "mgm_multi_aggregate: one cost volume per device");
        if (k == 0) nx = a, ny = b, dmin = c, dmax = d;
        else if (a != nx || b != ny || c != dmin || d != dmax) return fail(m, MGM_ERR_INVALID, "mgm_multi_aggregate: the devices' cost volumes differ in geometry");
    }
    {
        int a, b, c;
        if (mgm_img_dims(out0, &a, &b, &c) || a != nx || b != ny || mgm_img_dims(outcost0, &a, &b, &c) || a != nx || b != ny)
            return fail(m, MGM_ERR_INVALID, "mgm_multi_aggregate: output image size mismatch");
    }
    const size_t L = (size_t)(dmax - dmin + 1), rowf = (size_t)nx * L;  // floats per image row of a volume
    std::vector<int> first(n), cnt(n), row0(n), nrows(n);
    mgm_multi_plan(n, NDIR, ny, first.data(), cnt.data(), row0.data(), nrows.data());
    auto owner = [&](int p) {
        for (int k = 0; k < n; k++)
            if (p >= first[k] && p < first[k] + cnt[k]) return k;
        return -1;
    };
    int r;
    // 1. every device runs its block of passes (asynchronous: the devices work concurrently)
    for (int k = 0; k < n; k++) {
        if (!cnt[k]) continue;  // more devices than passes: this one only sums and searches its rows
        if ((r = mgm_aggregate_passes_dev(m->ctx[k], C[k], w8 ? w8[k] : nullptr, P1, P2, MGM, use_fh, first[k], cnt[k])))
            return fail(m, r, std::string("device ") + std::to_string(m->dev[k]) + ": " + mgm_last_error(m->ctx[k]));
        if (m->loopback) {
            if (hipSetDevice(m->dev[k]) != hipSuccess || hipEventRecord(m->passes_done[k], (hipStream_t)mgm_ctx_stream(m->ctx[k])) != hipSuccess)
                return fail(m, MGM_ERR_HIP, "hipEventRecord");
        }
    }
    // 2. receive buffers: all passes of this device's rows, in pass order
    for (int k = 0; k < n; k++) {
        if ((r = reserve(m, k, m->recv[k], sizeof(float) * (size_t)NDIR * std::max(nrows[k], 1) * rowf))) return r;
        if ((r = reserve(m, k, m->rows[k], sizeof(float) * 2 * (size_t)std::max(nrows[k], 1) * nx))) return r;
    }
    // 3. the exchange.  Rank k sends, for each of its passes, rank g's rows to g; and receives its own rows of every other
    //    rank's passes.  Between any two ranks the messages are posted in pass order on both sides.
    auto lr_of = [&](int k, int p) { return (const float *)mgm_lr_device_ptr(m->ctx[k], p - first[k]); };
    for (int k = 0; k < n; k++)
        for (int p = first[k]; p < first[k] + cnt[k]; p++)
            if (!lr_of(k, p)) return fail(m, MGM_ERR_INTERNAL, "mgm_multi_aggregate: no Lr volume (padded label count?)");
    if (!m->loopback && n > 1)
        if (g_rccl.GroupStart() != ncclSuccess) return fail(m, MGM_ERR_HIP, "ncclGroupStart");
    for (int k = 0; k < n; k++) {
        if (hipSetDevice(m->dev[k]) != hipSuccess) return fail(m, MGM_ERR_HIP, "hipSetDevice");
        hipStream_t sk = (hipStream_t)mgm_ctx_stream(m->ctx[k]);
        float *rk = (float *)m->recv[k].p;
        const size_t slab_k = (size_t)nrows[k] * rowf;
        for (int p = 0; p < NDIR; p++) {  // what lands in rank k's buffer
            const int g = owner(p);
            if (!nrows[k]) continue;
            if (g == k) {
                if (hipMemcpyAsync(rk + (size_t)p * slab_k, lr_of(k, p) + (size_t)row0[k] * rowf, sizeof(float) * slab_k, hipMemcpyDeviceToDevice, sk) != hipSuccess)
                    return fail(m, MGM_ERR_HIP, "hipMemcpyAsync (own slab)");
            } else if (m->loopback) {
                if (hipStreamWaitEvent(sk, m->passes_done[g], 0) != hipSuccess ||
                    hipMemcpyAsync(rk + (size_t)p * slab_k, lr_of(g, p) + (size_t)row0[k] * rowf, sizeof(float) * slab_k, hipMemcpyDeviceToDevice, sk) != hipSuccess)
                    return fail(m, MGM_ERR_HIP, "hipMemcpyAsync (loopback slab)");
            } else {
                const ncclResult_t e = g_rccl.Recv(rk + (size_t)p * slab_k, slab_k, ncclFloat, g, m->comm[k], sk);
                if (e != ncclSuccess) return fail(m, MGM_ERR_HIP, std::string("ncclRecv: ") + g_rccl.GetErrorString(e));
            }
        }
        if (!m->loopback)
            for (int p = first[k]; p < first[k] + cnt[k]; p++)  // what rank k sends
                for (int g = 0; g < n; g++) {
                    if (g == k || !nrows[g]) continue;
                    const ncclResult_t e = g_rccl.Send(lr_of(k, p) + (size_t)row0[g] * rowf, (size_t)nrows[g] * rowf, ncclFloat, g, m->comm[k], sk);
                    if (e != ncclSuccess) return fail(m, MGM_ERR_HIP, std::string("ncclSend: ") + g_rccl.GetErrorString(e));
                }
    }
    if (!m->loopback && n > 1)
        if (g_rccl.GroupEnd() != ncclSuccess) return fail(m, MGM_ERR_HIP, "ncclGroupEnd");
    // 4. every device finishes its rows: ordered sum over the passes, over-count fix, WTA, refinement
    float *o0 = (float *)mgm_img_device_ptr(out0), *c0 = (float *)mgm_img_device_ptr(outcost0);
    for (int k = 0; k < n; k++) {
        if (!nrows[k]) continue;
        float *ok = k == 0 ? o0 + (size_t)row0[0] * nx : (float *)m->rows[k].p;
        float *ck = k == 0 ? c0 + (size_t)row0[0] * nx : (float *)m->rows[k].p + (size_t)nrows[k] * nx;
        if ((r = mgm_wta_rows_dev(m->ctx[k], C[k], row0[k], nrows[k], m->recv[k].p, NDIR, fix_overcount, refine, ok, ck)))
            return fail(m, r, std::string("device ") + std::to_string(m->dev[k]) + ": " + mgm_last_error(m->ctx[k]));
        if (m->loopback && k > 0)
            if (hipSetDevice(m->dev[k]) != hipSuccess || hipEventRecord(m->passes_done[k], (hipStream_t)mgm_ctx_stream(m->ctx[k])) != hipSuccess)
                return fail(m, MGM_ERR_HIP, "hipEventRecord");
    }
    // 5. the rows travel to device 0 (two W x nrows images per device)
    if (n > 1) {
        if (!m->loopback && g_rccl.GroupStart() != ncclSuccess) return fail(m, MGM_ERR_HIP, "ncclGroupStart");
        for (int k = 1; k < n; k++) {
            if (!nrows[k]) continue;
            const size_t cntk = (size_t)nrows[k] * nx;
            hipStream_t s0 = (hipStream_t)mgm_ctx_stream(m->ctx[0]), sk = (hipStream_t)mgm_ctx_stream(m->ctx[k]);
            if (m->loopback) {
                if (hipSetDevice(m->dev[0]) != hipSuccess || hipStreamWaitEvent(s0, m->passes_done[k], 0) != hipSuccess ||
                    hipMemcpyAsync(o0 + (size_t)row0[k] * nx, m->rows[k].p, sizeof(float) * cntk, hipMemcpyDeviceToDevice, s0) != hipSuccess ||
                    hipMemcpyAsync(c0 + (size_t)row0[k] * nx, (float *)m->rows[k].p + cntk, sizeof(float) * cntk, hipMemcpyDeviceToDevice, s0) != hipSuccess)
                    return fail(m, MGM_ERR_HIP, "hipMemcpyAsync (loopback rows)");
            } else {
                if (hipSetDevice(m->dev[k]) != hipSuccess) return fail(m, MGM_ERR_HIP, "hipSetDevice");
                if (g_rccl.Send(m->rows[k].p, 2 * cntk, ncclFloat, 0, m->comm[k], sk) != ncclSuccess) return fail(m, MGM_ERR_HIP, "ncclSend (rows)");
            }
        }
        if (!m->loopback) {
            // device 0 receives each rank's two images back to back into a staging area, then places them
            if (hipSetDevice(m->dev[0]) != hipSuccess) return fail(m, MGM_ERR_HIP, "hipSetDevice");
            hipStream_t s0 = (hipStream_t)mgm_ctx_stream(m->ctx[0]);
            size_t total = 0;
            for (int k = 1; k < n; k++) total += 2 * (size_t)nrows[k] * nx;
            // (the staging area follows rank 0's own receive buffer use: a separate allocation)
            static_assert(sizeof(float) == 4, "");
            Grow &st = m->rows[0];
            if ((r = reserve(m, 0, st, sizeof(float) * std::max<size_t>(total, 1)))) {
                (void)g_rccl.GroupEnd();
                return r;
            }
            size_t off = 0;
            for (int k = 1; k < n; k++) {
                if (!nrows[k]) continue;
                const size_t cntk = (size_t)nrows[k] * nx;
                if (g_rccl.Recv((float *)st.p + off, 2 * cntk, ncclFloat, k, m->comm[0], s0) != ncclSuccess) {
                    (void)g_rccl.GroupEnd();
                    return fail(m, MGM_ERR_HIP, "ncclRecv (rows)");
                }
                off += 2 * cntk;
            }
            if (g_rccl.GroupEnd() != ncclSuccess) return fail(m, MGM_ERR_HIP, "ncclGroupEnd");
            off = 0;
            for (int k = 1; k < n; k++) {
                if (!nrows[k]) continue;
                const size_t cntk = (size_t)nrows[k] * nx;
                if (hipMemcpyAsync(o0 + (size_t)row0[k] * nx, (float *)st.p + off, sizeof(float) * cntk, hipMemcpyDeviceToDevice, s0) != hipSuccess ||
                    hipMemcpyAsync(c0 + (size_t)row0[k] * nx, (float *)st.p + off + cntk, sizeof(float) * cntk, hipMemcpyDeviceToDevice, s0) != hipSuccess)
                    return fail(m, MGM_ERR_HIP, "hipMemcpyAsync (rows)");
                off += 2 * cntk;
            }
        }
    }
    // 6. everything has been enqueued; the call returns when the result is on device 0 (and no watchdog fired anywhere)
    for (int k = n - 1; k >= 0; k--)
        if ((r = mgm_ctx_synchronize(m->ctx[k]))) return fail(m, r, std::string("device ") + std::to_string(m->dev[k]) + ": " + mgm_last_error(m->ctx[k]));
    return MGM_OK;
}

}  // extern "C"
