// mgm_main.cc -- the `mgm` command line on top of libmgm_hip.so.
//
// A from-scratch restatement of the reference's host program (mgm.cc:266-450): same
// options, same environment parameters, same stdout, same left->right / right->left
// orchestration and post-processing -- with the hot path (cost volume, aggregation, WTA,
// refinement; mgm.cc:372-385 and 405-414) running on the MI355X through the C ABI of
// include/mgm_hip.h.  Everything here is host glue; no stereo arithmetic of the path is
// computed on the CPU.  Image files (src/imgio.h): PNG, TIFF, PGM/PPM, PFM and .npy in -- what the
// reference's iio decodes them to -- and float TIFF, PFM or .npy out.
//
// A command line is a JOB with three stages: DECODE (host: the input files -> planar float images), DEVICE (this thread:
// uploads, every kernel of the path, downloads; the job's stdout is written here, in order) and ENCODE (host: the output
// files).  A 1920x1080x256 pair is ~17 ms of device work between ~19 ms of decoding and ~9 ms of encoding, so the program
// is organised around its HOST costs:
//   * one-shot (`mgm u v out`): both inputs are decoded on their own threads while this thread brings the device context
//     up, and the context is torn down on a thread of its own while the outputs are encoded and written;
//   * resident (`mgm --batch FILE`): the three stages of consecutive jobs OVERLAP -- a decoder thread works ahead of the
//     device stage, a writer thread behind it -- and the device objects of a pair (images, volumes, maps) are kept and
//     refilled by the next pair of the same geometry instead of being freed and allocated again.  A failing job (bad
//     option, unreadable file, device-side refusal) fails alone: every device-side error is an exception caught per job.
//     A line whose input file is an earlier line's OUTPUT (a coarse-to-fine chain: -m/-M written by the line before) is
//     decoded only once that output has been written: lines behave as if run one after the other.
// MGM_HIP_STATS=1 prints the wall-time breakdown on stderr; MGM_HIP_KERNELS=1 the names of the kernels a job launched.
//
// WITH_MGM2=1 (mgm_naive_parallelism, mgm_core.cc:632-831): every pass on its own private Lr volume, all passes in
// flight at once, then the volumes accumulated into S -- which is how the device path is organised anyway
// (DESIGN.md 4).  The reference accumulates in thread-finish order (798-805); here the order is always 0, 1, ..., 7:
// what its default build (Makefile:1, no OpenMP) and any run with one thread compute, and one member of the outcome
// set of a threaded run.  So the flag is accepted and changes nothing.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../include/mgm_hip.h"
#include "imgio.h"

// ---- MGM_HIP_STATS=1: where the wall time of the run goes (stderr, one line) ------------------
struct Stopwatch {
    bool on = getenv("MGM_HIP_STATS") && atoi(getenv("MGM_HIP_STATS")) != 0;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
    std::string line;
    void mark(const char *what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        char buf[96];
        snprintf(buf, sizeof buf, " %s %.1f", what, std::chrono::duration<double, std::milli>(now - last).count());
        line += buf;
        last = now;
    }
    // milliseconds between exec and main(): dynamic loading of the HIP runtime and its static initialisers (10 ms ticks)
    static double premain_ms()
    {
        double up = 0;
        long long start = 0;
        if (FILE *f = fopen("/proc/uptime", "r")) {
            if (fscanf(f, "%lf", &up) != 1) up = 0;
            fclose(f);
        }
        if (FILE *f = fopen("/proc/self/stat", "r")) {
            char buf[1024];
            const size_t n = fread(buf, 1, sizeof buf - 1, f);
            fclose(f);
            buf[n] = 0;
            if (const char *q = strrchr(buf, ')')) {  // fields after "(comm)": state is #3, starttime #22
                int field = 2;
                for (q++; *q && field < 22; q++)
                    if (*q == ' ') field++;
                start = atoll(q);
            }
        }
        return (up - (double)start / 100.0) * 1e3;
    }
    double pre = on ? premain_ms() : 0;
    void report(const char *stage)
    {
        if (!on) return;
        fprintf(stderr, "[mgm stats] exec -> main() ~%.0f ms (10 ms ticks);[mgm stats] %s, ms:%s | total %.1f\n", pre, stage, line.c_str(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

// ---- environment "smart parameters" (smartparameter.h:26-50): double, read once --------------
static double env_param(const char *name, double dflt)
{
    const char *s = getenv(name);
    double y;
    if (s && sscanf(s, "%lf", &y) == 1) return y;
    return dflt;
}

// ---- pick_option (mgm.cc:165-179): single-dash name, value = next argv, both removed ----------
static const char *pick_option(int *argc, char **argv, const char *opt, const char *dflt)
{
    for (int i = 0; i < *argc - 1; i++)
        if (argv[i][0] == '-' && 0 == strcmp(argv[i] + 1, opt)) {
            const char *r = argv[i + 1];
            *argc -= 2;
            for (int j = i; j < *argc; j++) argv[j] = argv[j + 2];
            return r;
        }
    return dflt;
}

static void remove_nonfinite(HostImg &u, float v)  // img_tools.h:37-41
{
    for (float &x : u.data)
        if (!std::isfinite(x)) x = v;
}

struct Opts {
    int dmin, dmax, NDIR;
    float P1, P2, aP2, aThresh, truncDist;
    std::string distance, prefilter, refine;
    int TSGM, FH, FIX, census_win;
};

// A device-side failure: thrown, never exit() -- in resident mode the job fails and the next one runs.
// code(): the process exit code of a one-shot run (2: valid in the reference, refused here).
struct DeviceError : std::runtime_error {
    int rc;
    DeviceError(int rc_, const std::string &m) : std::runtime_error(m), rc(rc_) {}
    int code() const { return rc == MGM_ERR_UNSUPPORTED ? 2 : 1; }
};
[[noreturn]] static void die(mgm_ctx *ctx, int rc, const char *what)
{
    char head[96];
    snprintf(head, sizeof head, "%s failed (%d): ", what, rc);
    throw DeviceError(rc, std::string(head) + (ctx ? mgm_last_error(ctx) : ""));
}

// One run of the path: mgm.cc:372 + 376-385 (u,v) or 373 + 405-414 (v,u with the negated range), in stages so
// that the two runs of a pair can share one launch of the pass kernel (mgm_aggregate_batch_dev).
struct Run {
    mgm_img *du = nullptr, *dv = nullptr, *dw = nullptr, *dout = nullptr, *dcost = nullptr, *dspare = nullptr;
    mgm_cv *C = nullptr;
    bool weighted_msg = false, borrowed_uv = false, ragged = false;
    int nx = 0, ny = 0, nch = 0, vnx = 0, vny = 0, dmin = 0, dmax = 0;  // (what the kept objects were made for)
};

static void free_run(mgm_ctx *ctx, Run &r)
{
    mgm_cv_free(ctx, r.C);
    if (!r.borrowed_uv) {
        mgm_img_free(ctx, r.du);
        mgm_img_free(ctx, r.dv);
    }
    for (mgm_img *im : {r.dw, r.dout, r.dcost, r.dspare}) mgm_img_free(ctx, im);
    r = Run();
}

// the device image of `h`: a kept one of the same size refilled, else a new one
static void upload_into(mgm_ctx *ctx, const HostImg &h, mgm_img **im)
{
    int nx = 0, ny = 0, nch = 0, rc;
    if (*im && mgm_img_dims(*im, &nx, &ny, &nch) == MGM_OK && nx == h.nx && ny == h.ny && nch == h.nch) {
        if ((rc = mgm_img_update(ctx, *im, h.data.data()))) die(ctx, rc, "upload");
        return;
    }
    mgm_img_free(ctx, *im);
    *im = nullptr;
    if ((rc = mgm_img_upload(ctx, h.data.data(), h.nx, h.ny, h.nch, im))) die(ctx, rc, "upload");
}
static void image_like(mgm_ctx *ctx, int nx, int ny, int nch, mgm_img **im)  // a kept image of that size, or a new one
{
    int ax = 0, ay = 0, ac = 0, rc;
    if (*im && mgm_img_dims(*im, &ax, &ay, &ac) == MGM_OK && ax == nx && ay == ny && ac == nch) return;
    mgm_img_free(ctx, *im);
    *im = nullptr;
    if ((rc = mgm_img_create(ctx, nx, ny, nch, im))) die(ctx, rc, "mgm_img_create");
}

// `r` may hold the objects of an earlier pair (resident mode): what fits is refilled in place, the rest replaced.
// `uv_from`: the run of the same pair the other way round, whose uploaded images this one borrows (v,u instead of u,v).
static void prepare_run(mgm_ctx *ctx, const HostImg &u, const HostImg &v, int dmin, int dmax, const Opts &o, Run &r,
                        const HostImg *lo = nullptr, const HostImg *hi = nullptr, const Run *uv_from = nullptr)
{
    int rc;
    bool ragged = false;
    if (lo) {  // range images (-m/-M, mgm.cc:342-353): Dvec takes them as ints (mgm_costvolume.h:323)
        dmin = (int)lo->data[0];
        dmax = (int)hi->data[0];
        for (size_t i = 0; i < lo->data.size(); i++) ragged |= (int)lo->data[i] != dmin || (int)hi->data[i] != dmax;
    }
    if (ragged)
        for (size_t i = 0; i < lo->data.size(); i++) {
            dmin = std::min(dmin, (int)lo->data[i]);
            dmax = std::max(dmax, (int)hi->data[i]);
        }
    const bool keep = r.C && !r.ragged && !ragged && r.nx == u.nx && r.ny == u.ny && r.nch == u.nch && r.vnx == v.nx && r.vny == v.ny &&
                      r.dmin == dmin && r.dmax == dmax && r.borrowed_uv == (uv_from != nullptr);
    if (!keep) free_run(ctx, r);
    r.nx = u.nx, r.ny = u.ny, r.nch = u.nch, r.vnx = v.nx, r.vny = v.ny, r.dmin = dmin, r.dmax = dmax, r.ragged = ragged;
    r.weighted_msg = false;
    if (uv_from) {
        r.du = uv_from->dv;
        r.dv = uv_from->du;
        r.borrowed_uv = true;
    } else {
        upload_into(ctx, u, &r.du);
        upload_into(ctx, v, &r.dv);
    }
    // compute_mgm_weights(u, aP2, aThresh)   [aP1 is parsed and unused in the reference too, mgm.cc:372].  With -aP2 1 (the
    // default) every weight is 1 whatever the image (ws(), mgm_weights.h:38-42, returns aP3 or 1): mgm() then takes the
    // unweighted update functions (mgm_core.cc:420-423) -- no planes to compute, upload or scan.
    if (o.aP2 != 1.0f) {
        if ((rc = mgm_weights_dev(ctx, r.du, o.aP2, o.aThresh, &r.dw))) die(ctx, rc, "mgm_weights");
        // mgm() announces the weighted mode on stdout (mgm_core.cc:420-423)
        std::vector<float> w((size_t)u.npix() * 8);
        if ((rc = mgm_img_download(ctx, r.dw, w.data()))) die(ctx, rc, "download");
        r.weighted_msg = std::any_of(w.begin(), w.end(), [](float x) { return x != 1.0f; });
    } else if (r.dw) {
        mgm_img_free(ctx, r.dw);
        r.dw = nullptr;
    }
    if (ragged) {
        mgm_img *dlo = nullptr, *dhi = nullptr;
        if ((rc = mgm_img_upload(ctx, lo->data.data(), u.nx, u.ny, 1, &dlo)) || (rc = mgm_img_upload(ctx, hi->data.data(), u.nx, u.ny, 1, &dhi))) {
            mgm_img_free(ctx, dlo);
            die(ctx, rc, "upload");
        }
        rc = mgm_costvolume_build_ranged_dev(ctx, r.du, r.dv, dlo, dhi, dmin, dmax, o.prefilter.c_str(), o.distance.c_str(), o.truncDist,
                                             o.census_win, &r.C);
        mgm_img_free(ctx, dlo);
        mgm_img_free(ctx, dhi);
        if (rc) die(ctx, rc, "mgm_costvolume_build_ranged");
    } else if ((rc = mgm_costvolume_build_dev(ctx, r.du, r.dv, dmin, dmax, o.prefilter.c_str(), o.distance.c_str(), o.truncDist, o.census_win, &r.C))) {
        die(ctx, rc, "mgm_costvolume_build");
    }
    image_like(ctx, u.nx, u.ny, 1, &r.dout);
    image_like(ctx, u.nx, u.ny, 1, &r.dcost);
}

static void aggregate_run(mgm_ctx *ctx, const Opts &o, Run &r)
{
    const int rc = mgm_aggregate_dev(ctx, r.C, r.dw, o.P1, o.P2, o.NDIR, o.TSGM, o.FH, o.FIX, o.refine.c_str(), r.dout, r.dcost, nullptr);
    if (rc) die(ctx, rc, "mgm_aggregate");
}

// MGM_DEVICES=a,b,...: the passes of each mgm() run sharded over several GPUs of the node (mgm_multi_aggregate: every
// device builds the cost volume from the images, runs its block of passes, row slabs travel over RCCL, ordered sum and
// WTA per device, rows gathered on the first device).  `r` has been prepared on the first device.
static void aggregate_run_multi(mgm_multi *m, const HostImg &u, const HostImg &v, int dmin, int dmax, const Opts &o, Run &r)
{
    const int n = mgm_multi_size(m);
    std::vector<Run> shadow(n);
    std::vector<const mgm_cv *> Cs(n);
    std::vector<const mgm_img *> Ws(n);
    Cs[0] = r.C;
    Ws[0] = r.dw;
    auto drop = [&] {
        for (int k = 1; k < n; k++) free_run(mgm_multi_ctx(m, k), shadow[k]);
    };
    try {
        for (int k = 1; k < n; k++) {
            prepare_run(mgm_multi_ctx(m, k), u, v, dmin, dmax, o, shadow[k]);
            Cs[k] = shadow[k].C;
            Ws[k] = shadow[k].dw;
        }
    } catch (...) {
        drop();
        throw;
    }
    const int rc = mgm_multi_aggregate(m, Cs.data(), r.dw ? Ws.data() : nullptr, o.P1, o.P2, o.NDIR, o.TSGM, o.FH, o.FIX, o.refine.c_str(), r.dout, r.dcost);
    drop();
    if (rc) {
        char head[96];
        snprintf(head, sizeof head, "mgm_multi_aggregate failed (%d): ", rc);
        throw DeviceError(rc, std::string(head) + mgm_multi_last_error(m));
    }
}

// what mgm() and print_solution_energy put on stdout for one run (mgm_core.cc:420-423, 491; mgm_print_energy.h:109-111)
static void report_run(const Opts &o, const Run &r)
{
    if (r.weighted_msg) printf(" USING IMAGE DEPENDENT WEIGHTS\n");
    for (int p = 0; p < o.NDIR; p++) printf("%d", p);
    printf("\n");
    fflush(stdout);
}

static HostImg download(mgm_ctx *ctx, const mgm_img *im, int nx, int ny, int nch)
{
    HostImg h;
    h.nx = nx;
    h.ny = ny;
    h.nch = nch;
    h.data.resize((size_t)nx * ny * nch);
    const int rc = mgm_img_download(ctx, im, h.data.data());
    if (rc) die(ctx, rc, "download");
    return h;
}

// outoff = median_filter(outoff, MEDIAN) on the device (mgm.cc:396, 419), any radius
static void median_run(mgm_ctx *ctx, Run &r, int radius)
{
    int rc;
    image_like(ctx, r.nx, r.ny, 1, &r.dspare);
    if ((rc = mgm_median_dev(ctx, r.dout, radius, r.dspare))) die(ctx, rc, "mgm_median");
    std::swap(r.dout, r.dspare);
}

// iterations 2..TSGM_ITER of main()'s loop (mgm.cc:377-388): the ranges narrow around the previous solution
// (update_dmin_dmax), the volume -- and with it every scan-line pass -- stays the same, so only the winner search and
// the refinement are redone, on the Lr volumes the context still holds
static void iterate_run(mgm_ctx *ctx, const Opts &o, Run &r, int iterations, int dmin, int dmax, const HostImg *lo0 = nullptr,
                        const HostImg *hi0 = nullptr)
{
    if (iterations < 2) return;
    std::vector<float> lo((size_t)r.nx * r.ny, (float)dmin), hi((size_t)r.nx * r.ny, (float)dmax);
    if (lo0) {
        lo = lo0->data;
        hi = hi0->data;
    }
    mgm_img *dlo = nullptr, *dhi = nullptr;
    int rc;
    if ((rc = mgm_img_upload(ctx, lo.data(), r.nx, r.ny, 1, &dlo)) || (rc = mgm_img_upload(ctx, hi.data(), r.nx, r.ny, 1, &dhi))) {
        mgm_img_free(ctx, dlo);
        die(ctx, rc, "upload");
    }
    for (int it = 1; it < iterations; it++) {
        const char *what = "mgm_update_ranges";
        if (!(rc = mgm_update_ranges_dev(ctx, r.dout, dlo, dhi, 3, 2))) {
            what = "mgm_wta_windowed";
            rc = mgm_wta_windowed_dev(ctx, r.C, o.NDIR, o.FIX, o.refine.c_str(), dlo, dhi, r.dout, r.dcost);
        }
        if (rc) {
            mgm_img_free(ctx, dlo);
            mgm_img_free(ctx, dhi);
            die(ctx, rc, what);
        }
        report_run(o, r);
    }
    mgm_img_free(ctx, dlo);
    mgm_img_free(ctx, dhi);
}

// The device side of the process: created for the first job, kept for the following ones (resident mode) together with
// the device objects of the last pair (refilled by the next pair of the same geometry).
struct Session {
    mgm_ctx *ctx = nullptr;
    mgm_multi *multi = nullptr;
    std::vector<int> devs;
    bool tried = false, resident = false;
    int rc_ctx = 0;
    Run L, R;
};

// ---- a job ------------------------------------------------------------------------------------------------------------
struct Job {
    std::vector<std::string> tok;  // the command line (argv[0] included)
    // parsed
    Opts o{};
    std::string min_file, max_file, nolr_file, f_u, f_v, f_out, f_cost, f_back;
    int early = -1;           // >= 0: the command line ends before any work with this code ...
    std::string early_out;    // ... after these lines on stdout
    std::string early_err;    // ... and stderr
    // decoded
    HostImg u, v, rlo, rhi;
    std::exception_ptr decode_err;
    // results of the device stage
    HostImg outoff, outcost, syn, nolr;
    int rc = 0;  // the job's exit code
};

static void parse_job(Job &j)
{
    std::vector<char *> av;
    for (auto &t : j.tok) av.push_back(&t[0]);
    av.push_back(nullptr);
    int argc = (int)j.tok.size();
    char **argv = av.data();
    auto early = [&](int code, const char *out, const std::string &err = "") {
        j.early = code;
        if (out) j.early_out = std::string(out) + "\n";
        j.early_err = err;
    };
    if (argc < 2 || !strcmp(argv[1], "-h")) return early(0, "usage:\n\tmgm [-options] u v out [cost [backflow]]");
    if (!strcmp(argv[1], "-?")) return early(0, "Compute stereo disparities by the MGM algorithm.");
    if (!strcmp(argv[1], "--version")) return early(0, "mgm 2.0 (mgm-hip, MI355X)");
    if (!strcmp(argv[1], "--help"))
        return early(0, "mgm [options] in_u in_v out_disp [out_cost [out_backflow]]   (in: png tif pgm ppm pfm npy; out: tif pfm npy)\n"
                        "options: -r dmin(-30) -R dmax(30) -O NDIR(4) -P1 (8) -P2 (32) -p prefilter(none) -t distance(ad)\n"
                        "         -truncDist (inf) -s subpix(none) -aP1 (1) -aP2 (1) -aThresh (5) -m FILE -M FILE -l FILE\n"
                        "environment: CENSUS_NCC_WIN=3 TESTLRRL=1 TESTLRRL_TAU=1.0 MEDIAN=0 TSGM=4 TSGM_ITER=1\n"
                        "             TSGM_FIX_OVERCOUNT=1 USE_TRUNCATED_LINEAR_POTENTIALS=0 MGM_DEVICE=0 MGM_DEVICES=0,1,...\n"
                        "resident mode: mgm --batch FILE|-   (one such command line per line of FILE, one device context for all)");
    if (argc < 4)
        return early(1, nullptr, std::string("too few parameters\n   usage: ") + argv[0] +
                                     "  [-r dmin -R dmax] [-m dminImg -M dmaxImg] [-O NDIR: 2, (4), 8] u v out [cost [backflow]]\n");
    // mgm.cc:303-318, in the same order (the order matters for pick_option's argv surgery)
    j.min_file = pick_option(&argc, argv, "m", "");
    j.max_file = pick_option(&argc, argv, "M", "");
    Opts &o = j.o;
    o.dmin = atoi(pick_option(&argc, argv, "r", "-30"));
    o.dmax = atoi(pick_option(&argc, argv, "R", "30"));
    o.NDIR = atoi(pick_option(&argc, argv, "O", "4"));
    o.P1 = (float)atof(pick_option(&argc, argv, "P1", "8"));
    o.P2 = (float)atof(pick_option(&argc, argv, "P2", "32"));
    pick_option(&argc, argv, "aP1", "1");  // accepted and unused, as in the reference
    o.aP2 = (float)atof(pick_option(&argc, argv, "aP2", "1"));
    o.aThresh = (float)atof(pick_option(&argc, argv, "aThresh", "5"));
    o.distance = pick_option(&argc, argv, "t", "ad");
    o.prefilter = pick_option(&argc, argv, "p", "none");
    o.refine = pick_option(&argc, argv, "s", "none");
    o.truncDist = (float)atof(pick_option(&argc, argv, "truncDist", "inf"));
    j.nolr_file = pick_option(&argc, argv, "l", "");
    if (argc > 1) j.f_u = argv[1];
    if (argc > 2) j.f_v = argv[2];
    if (argc > 3) j.f_out = argv[3];
    if (argc > 4) j.f_cost = argv[4];
    if (argc > 5) j.f_back = argv[5];
    if (argc < 4) return early(1, nullptr, "too few parameters\n");
    o.TSGM = (int)env_param("TSGM", 4);
    o.FH = (int)env_param("USE_TRUNCATED_LINEAR_POTENTIALS", 0);
    o.FIX = (int)env_param("TSGM_FIX_OVERCOUNT", 1);
    o.census_win = (int)env_param("CENSUS_NCC_WIN", 3);
}

// DECODE: the input files of a job (any thread; touches neither the device nor stdout).  Both images on their own threads.
static void decode_job(Job &j)
{
    if (j.early >= 0) return;
    std::exception_ptr eu, ev;
    std::thread tv([&] { try { j.v = imgio::read(j.f_v); remove_nonfinite(j.v, 0); } catch (...) { ev = std::current_exception(); } });
    try {
        j.u = imgio::read(j.f_u);
        remove_nonfinite(j.u, 0);
    } catch (...) {
        eu = std::current_exception();
    }
    tv.join();
    j.decode_err = eu ? eu : ev;
    if (!j.decode_err && !j.min_file.empty()) {
        try {
            j.rlo = imgio::read(j.min_file);
            j.rhi = imgio::read(j.max_file);
        } catch (...) {
            j.decode_err = std::current_exception();
        }
    }
}

// ENCODE: the output files of a job (any thread)
static void encode_job(Job &j)
{
    if (j.early >= 0 || j.rc != 0) return;
    try {
        if (!j.nolr_file.empty()) imgio::write(j.nolr_file, j.nolr);
        imgio::write(j.f_out, j.outoff);
        if (!j.f_cost.empty()) imgio::write(j.f_cost, j.outcost);
        if (!j.f_back.empty()) imgio::write(j.f_back, j.syn);
    } catch (const std::exception &e) {
        fprintf(stderr, "mgm: %s\n", e.what());
        j.rc = 1;
    }
}

static void bring_up(Session &S, int ITER, bool ranged)
{
    if (S.tried) return;  // (resident mode: the first job brings the device side up, the others find it there)
    S.tried = true;
    if (const char *dl = getenv("MGM_DEVICES"))  // "0,1,2,3": several GPUs of this node
        for (const char *q = dl; *q;) {
            char *end;
            const long d = strtol(q, &end, 10);
            if (end == q) break;
            S.devs.push_back((int)d);
            q = *end == ',' ? end + 1 : end;
        }
    if (S.devs.size() > 1 && (ITER > 1 || ranged)) {
        fprintf(stderr, "mgm: MGM_DEVICES: TSGM_ITER > 1 and range images run on the first device only\n");
        S.devs.resize(1);
    }
    if (S.devs.size() > 1) {
        if ((S.rc_ctx = mgm_multi_create(S.devs.data(), (int)S.devs.size(), &S.multi)) == 0) S.ctx = mgm_multi_ctx(S.multi, 0);
    } else {
        S.rc_ctx = mgm_ctx_create(S.devs.size() == 1 ? S.devs[0] : (int)env_param("MGM_DEVICE", 0), &S.ctx);
        // MGM_PLACE_TRIES=n: a context that stays may try n physical placements of its workspace and keep the fastest (off by default: three
        // more allocations of a 34 GB workspace cost ~2 s, what 100 further pairs gain back)
        if (S.rc_ctx == 0 && S.resident) (void)mgm_ctx_set_placement_tries(S.ctx, (int)env_param("MGM_PLACE_TRIES", 0));
        if (S.rc_ctx == 0 && getenv("MGM_HIP_KERNELS") && atoi(getenv("MGM_HIP_KERNELS")) != 0) (void)mgm_timing_enable(S.ctx, 1);
    }
}

// DEVICE: everything between the decoded inputs and the downloaded maps; the job's stdout.  `decoded`: waits for the
// job's DECODE stage (which may run beside the context's creation).  Returns the job's exit code; device objects stay in
// S.L / S.R (kept for the next pair, or torn down by the caller).
static int device_job(Session &S, Job &j, const std::function<void()> &decoded, Stopwatch &sw)
{
    Opts o = j.o;
    printf("%d %d\n", o.dmin, o.dmax);  // mgm.cc:328
    fflush(stdout);
    const double TSGM_ITER = env_param("TSGM_ITER", 1), TESTLRRL = env_param("TESTLRRL", 1);
    const double TAU = env_param("TESTLRRL_TAU", 1.0), MEDIAN = env_param("MEDIAN", 0);
    (void)env_param("WITH_MGM2", 0);  // accepted: see the header comment
    // main()'s loops are `for (int i = 0; i < TSGM_ITER(); i++)` on a DOUBLE (mgm.cc:377, 406): ceil() iterations; none at
    // all for TSGM_ITER <= 0 -- the maps then stay the zero images they were allocated as (mgm.cc:360-365)
    const int ITER = TSGM_ITER > 0 ? (int)std::ceil(TSGM_ITER) : 0;
    const bool ranged = !j.min_file.empty();
    try {
        bring_up(S, ITER, ranged);
        sw.mark("context");
        decoded();
        sw.mark("decode(rest)");
        if (j.decode_err) std::rethrow_exception(j.decode_err);
        mgm_ctx *ctx = S.ctx;
        mgm_multi *multi = S.multi;
        if (S.rc_ctx && S.devs.size() > 1) {
            fprintf(stderr, "mgm: MGM_DEVICES: cannot set up %d devices (mgm_multi_create = %d): %s\n", (int)S.devs.size(), S.rc_ctx, mgm_multi_last_error(nullptr));
            return 1;
        }
        if (S.rc_ctx) {
            fprintf(stderr, "mgm: no usable MI355X device (mgm_ctx_create = %d); there is no CPU path\n", S.rc_ctx);
            return 1;
        }
        if (multi && ranged) {
            fprintf(stderr, "mgm: --batch with MGM_DEVICES: range images are not supported on several devices\n");
            return 1;
        }
        const HostImg &u = j.u, &v = j.v;
        HostImg &rlo = j.rlo, &rhi = j.rhi;  // -m / -M range images of the left->right run (mgm.cc:342-353); the right->left run keeps -r/-R
        if (ranged) {
            if (rlo.nx != u.nx || rlo.ny != u.ny || rhi.nx != u.nx || rhi.ny != u.ny || rlo.nch != 1 || rhi.nch != 1) {
                fprintf(stderr, "mgm: the -m/-M images must have the size of the left image\n");
                return 1;
            }
            remove_nonfinite(rlo, (float)o.dmin);
            remove_nonfinite(rhi, (float)o.dmax);
            for (size_t i = 0; i < rlo.data.size(); i++)
                if (rhi.data[i] < rlo.data[i] + 1) rhi.data[i] = ceilf(rlo.data[i] + 1);
        }
        const HostImg *plo = ranged ? &rlo : nullptr, *phi = ranged ? &rhi : nullptr;
        o.P1 *= u.nch;  // mgm.cc:356-357
        o.P2 *= u.nch;

        int rc;
        Run &L = S.L, &R = S.R;
        const bool both = TESTLRRL != 0;
        if (!both && R.C) free_run(ctx, R);  // (kept from an earlier pair, not wanted by this one)
        prepare_run(ctx, u, v, o.dmin, o.dmax, o, L, plo, phi);
        bool together = false, r_ready = false;
        auto zero_run = [&](Run &r) {  // no iteration: mgm() is never called, outoff / outcost keep their zeros
            const std::vector<float> z((size_t)r.nx * r.ny, 0.0f);
            for (mgm_img **im : {&r.dout, &r.dcost})
                if ((rc = mgm_img_update(ctx, *im, z.data()))) die(ctx, rc, "upload");
        };
        // the right->left run reads the same two images the other way round: it borrows the left->right run's uploads
        // (not when the ranges are images: that run keeps -r/-R and is prepared -- and freed -- on its own)
        auto prepare_R = [&] {
            prepare_run(ctx, v, u, -o.dmax, -o.dmin, o, R, nullptr, nullptr, &L);  // mgm.cc:366, 405
            r_ready = true;
        };
        if (ITER == 0) {
            zero_run(L);
            together = true;  // (nothing to aggregate)
        }
        if (ITER > 0 && !multi && both && !plo && u.nx == v.nx && u.ny == v.ny && env_param("MGM_BATCH_LR", 1) != 0) {
            // both runs of the pair (mgm.cc:376-385 and 405-414) through ONE launch of the pass kernel
            prepare_R();
            const mgm_cv *Cs[2] = {L.C, R.C};
            const mgm_img *Ws[2] = {L.dw, R.dw};
            mgm_img *Os[2] = {L.dout, R.dout}, *Cc[2] = {L.dcost, R.dcost};
            rc = mgm_aggregate_batch_dev(ctx, 2, Cs, L.dw ? Ws : nullptr, o.P1, o.P2, o.NDIR, o.TSGM, o.FH, o.FIX, o.refine.c_str(), Os, Cc, nullptr);
            if (rc == MGM_OK) together = true;
            else if (rc == MGM_ERR_NOMEM) {  // 2*NDIR Lr volumes do not fit: one run after the other, as the reference does
                free_run(ctx, R);
                r_ready = false;
            } else if (rc != MGM_ERR_UNSUPPORTED) die(ctx, rc, "mgm_aggregate_batch");
            // (UNSUPPORTED: one image weighted, the other not -- the two runs take different update functions)
        }
        sw.mark("upload+enqueue");
        if (ITER > 0) {
            if (multi) aggregate_run_multi(multi, u, v, o.dmin, o.dmax, o, L);
            else if (!together) aggregate_run(ctx, o, L);
            report_run(o, L);
            iterate_run(ctx, o, L, ITER, o.dmin, o.dmax, plo, phi);
        }
        if (MEDIAN != 0) median_run(ctx, L, (int)MEDIAN);
        if (!j.nolr_file.empty()) j.nolr = download(ctx, L.dout, L.nx, L.ny, 1);
        if (both) {
            if (!r_ready) prepare_R();
            if (ITER == 0) {
                zero_run(R);
            } else {
                if (multi) aggregate_run_multi(multi, v, u, -o.dmax, -o.dmin, o, R);
                else if (!together) aggregate_run(ctx, o, R);
                report_run(o, R);
                iterate_run(ctx, o, R, ITER, -o.dmax, -o.dmin);
            }
            if (MEDIAN != 0) median_run(ctx, R, (int)MEDIAN);
            // leftright_test both ways on copies of the unchecked maps (mgm.cc:420-423)
            image_like(ctx, L.nx, L.ny, 1, &L.dspare);
            image_like(ctx, R.nx, R.ny, 1, &R.dspare);
            if ((rc = mgm_leftright_dev(ctx, R.dout, L.dout, (float)TAU, R.dspare)) ||
                (rc = mgm_leftright_dev(ctx, L.dout, R.dout, (float)TAU, L.dspare)))
                die(ctx, rc, "mgm_leftright");
            std::swap(L.dout, L.dspare);
            std::swap(R.dout, R.dspare);
        }
        sw.mark("enqueue(rest)");
        j.outoff = download(ctx, L.dout, L.nx, L.ny, 1);
        sw.mark("device+download");
        if (!j.f_cost.empty()) j.outcost = download(ctx, L.dcost, L.nx, L.ny, 1);
        // back-projected image (mgm.cc:433-443)
        if (!j.f_back.empty()) {
            mgm_img *dsyn = nullptr;
            if ((rc = mgm_img_create(ctx, u.nx, u.ny, u.nch, &dsyn))) die(ctx, rc, "mgm_img_create");
            if ((rc = mgm_backproject_dev(ctx, L.du, L.dv, L.dout, dsyn))) {
                mgm_img_free(ctx, dsyn);
                die(ctx, rc, "mgm_backproject");
            }
            try {
                j.syn = download(ctx, dsyn, u.nx, u.ny, u.nch);
            } catch (...) {
                mgm_img_free(ctx, dsyn);
                throw;
            }
            mgm_img_free(ctx, dsyn);
        }
        sw.mark("download(rest)");
        // MGM_HIP_KERNELS=1: which kernels the job ran, in launch order, on stderr (tests assert which pass kernel a command line took)
        if (getenv("MGM_HIP_KERNELS") && atoi(getenv("MGM_HIP_KERNELS")) != 0 && !multi) {
            const int n = mgm_timing_count(ctx);
            fprintf(stderr, "[mgm kernels]");
            for (int k = 0; k < n; k++) {
                const char *name = nullptr;
                float ms = 0;
                if (mgm_timing_get(ctx, k, &name, &ms) == 0 && name) fprintf(stderr, " %s", name);
            }
            fprintf(stderr, "\n");
            (void)mgm_timing_reset(ctx);
        }
    } catch (const DeviceError &e) {
        fprintf(stderr, "mgm: %s\n", e.what());
        // whatever the failed job left half-made goes; the context stays (resident mode: the next job starts clean)
        if (S.ctx) {
            free_run(S.ctx, S.R);
            free_run(S.ctx, S.L);
        }
        return e.code();
    } catch (const std::exception &e) {
        fprintf(stderr, "mgm: %s\n", e.what());
        if (S.ctx) {
            free_run(S.ctx, S.R);
            free_run(S.ctx, S.L);
        }
        return 1;
    }
    return 0;
}

static void teardown(Session &S)
{
    if (S.ctx) {
        free_run(S.ctx, S.R);
        free_run(S.ctx, S.L);
    }
    if (S.multi) mgm_multi_destroy(S.multi);
    else if (S.ctx) mgm_ctx_destroy(S.ctx);
    S.ctx = nullptr;
    S.multi = nullptr;
}

// ---- one-shot ---------------------------------------------------------------------------------------------------------
static int run_one(int argc, char **argv)
{
    Job j;
    for (int i = 0; i < argc; i++) j.tok.push_back(argv[i]);
    parse_job(j);
    if (j.early >= 0) {
        fputs(j.early_out.c_str(), stdout);
        fputs(j.early_err.c_str(), stderr);
        return j.early;
    }
    Stopwatch sw;
    Session S;
    // decode both inputs on their own threads while this one brings the device up (process start + HIP initialisation
    // + context is most of a run's wall time: the device work of a full-HD pair is ~17 ms)
    std::thread dec([&] { decode_job(j); });
    bool joined = false;
    j.rc = device_job(S, j, [&] { dec.join(); joined = true; }, sw);
    if (!joined) dec.join();
    if (j.rc) {
        fflush(stdout);
        fflush(stderr);
        return j.rc;  // (the process ends through the runtime's own handlers: a failed run is not timed)
    }
    // the device side is torn down (volumes, tens of GB of workspace) while the outputs are encoded and written
    std::thread down([&] { teardown(S); });
    encode_job(j);
    sw.mark("encode+write");
    down.join();
    sw.mark("teardown(rest)");
    sw.report("one-shot");
    // Everything is written and the device side is down: leave without the HIP runtime's own exit handlers (unloading
    // code objects, closing the device: tens of ms that produce nothing).  MGM_HIP_ORDERLY_EXIT=1 keeps them.
    fflush(stdout);
    fflush(stderr);
    if (!(getenv("MGM_HIP_ORDERLY_EXIT") && atoi(getenv("MGM_HIP_ORDERLY_EXIT")))) _exit(j.rc);
    return j.rc;
}

// ---- RESIDENT MODE: `mgm --batch FILE` (FILE = - : stdin) ---------------------------------------------------------------
// Every non-empty line of FILE is one command line of this program without its name -- options, u, v, out [cost
// [backflow]] -- and all of them run in THIS process, on one device context: HIP start-up (~0.1 s) and the allocation of
// the tens of GB of workspace are paid once.  Per job, stdout and every output file are what the one-shot command writes
// (the environment parameters are the process's, the same for all lines).  A `#` at the start of a token starts a comment.
// The stages of consecutive jobs overlap: job n+1 (and n+2) is decoded, and job n-1 encoded, while job n is on the
// device; stdout keeps the order of the lines.  Exit code: 0 if every line succeeded, else 1 (the remaining lines still
// run: a failing job -- unreadable input, bad option, device-side refusal, unwritable output -- fails alone).
static int run_batch(const char *file)
{
    FILE *f = strcmp(file, "-") ? fopen(file, "r") : stdin;
    if (!f) {
        fprintf(stderr, "mgm: --batch: cannot open %s\n", file);
        return 1;
    }
    std::vector<std::unique_ptr<Job>> jobs;
    std::vector<size_t> lineno;
    {
        char *buf = nullptr;
        size_t cap = 0, ln = 0;
        ssize_t n;
        while ((n = getline(&buf, &cap, f)) >= 0) {  // (lines of any length)
            ln++;
            std::unique_ptr<Job> j(new Job);
            j->tok.push_back("mgm");
            const std::string l(buf, (size_t)n);
            size_t i = 0;
            while (i < l.size()) {
                while (i < l.size() && strchr(" \t\r\n", l[i])) i++;
                size_t k = i;
                while (k < l.size() && !strchr(" \t\r\n", l[k])) k++;
                if (k > i && l[i] == '#') break;  // a comment: only where a token starts (a '#' inside a path is part of it)
                if (k > i) j->tok.push_back(l.substr(i, k - i));
                i = k;
            }
            if (j->tok.size() > 1) {
                jobs.push_back(std::move(j));
                lineno.push_back(ln);
            }
        }
        free(buf);
    }
    if (f != stdin) fclose(f);
    const size_t N = jobs.size();
    for (auto &j : jobs) parse_job(*j);

    // the decoder works at most AHEAD jobs ahead of the device stage (each decoded pair holds its images in memory)
    constexpr size_t AHEAD = 2;
    std::mutex mu;
    std::condition_variable cv;
    size_t decoded = 0, on_device = 0, to_write = 0, written = 0;  // jobs [0, decoded) are decoded; the device stage is at job on_device; [0, to_write) may be written; [0, written) are
    //  a line may consume what an earlier line produces (coarse-to-fine: the -m/-M range images of line k written by line k - 1):
    // the decoder then waits for the writer -- needs[k] = how many jobs must have been WRITTEN before job k's inputs are read
    std::vector<size_t> needs(N, 0);
    for (size_t k = 0; k < N; k++) {
        const Job &j = *jobs[k];
        for (size_t m = 0; m < k; m++) {
            const Job &e = *jobs[m];
            for (const std::string *in : {&j.f_u, &j.f_v, &j.min_file, &j.max_file})
                for (const std::string *out : {&e.f_out, &e.f_cost, &e.f_back, &e.nolr_file})
                    if (!in->empty() && *in == *out) needs[k] = m + 1;
        }
    }
    std::thread decoder([&] {
        for (size_t k = 0; k < N; k++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return k < on_device + 1 + AHEAD && written >= needs[k]; });
            }
            decode_job(*jobs[k]);
            {
                std::lock_guard<std::mutex> lk(mu);
                decoded = k + 1;
            }
            cv.notify_all();
        }
    });
    int bad = 0;
    std::thread writer([&] {
        for (size_t k = 0; k < N; k++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return k < to_write; });
            }
            Job &j = *jobs[k];
            encode_job(j);
            if (j.rc != 0) {
                fprintf(stderr, "mgm: --batch: line %zu failed\n", lineno[k]);
                std::lock_guard<std::mutex> lk(mu);
                bad = 1;
            }
            // the job is done: its images go
            j.u = j.v = j.rlo = j.rhi = j.outoff = j.outcost = j.syn = j.nolr = HostImg();
            {
                std::lock_guard<std::mutex> lk(mu);
                written = k + 1;
            }
            cv.notify_all();
        }
    });
    Session S;
    S.resident = true;
    for (size_t k = 0; k < N; k++) {
        Job &j = *jobs[k];
        {
            std::lock_guard<std::mutex> lk(mu);
            on_device = k;
        }
        cv.notify_all();
        if (j.early >= 0) {
            fputs(j.early_out.c_str(), stdout);
            fputs(j.early_err.c_str(), stderr);
            j.rc = j.early;
        } else {
            Stopwatch sw;
            j.rc = device_job(S, j, [&] {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return decoded > k; });
            }, sw);
            sw.report("resident job, device stage");
        }
        fflush(stdout);
        {
            std::lock_guard<std::mutex> lk(mu);
            to_write = k + 1;
        }
        cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        on_device = N;
    }
    cv.notify_all();
    decoder.join();
    writer.join();
    return bad;  // (the process leaves through _exit: no teardown)
}

int main(int argc, char **argv)
{
    if (argc >= 3 && !strcmp(argv[1], "--batch")) {
        const int r = run_batch(argv[2]);
        fflush(stdout);
        fflush(stderr);
        _exit(r);  // (as the one-shot run: without the HIP runtime's exit handlers)
    }
    return run_one(argc, argv);
}
