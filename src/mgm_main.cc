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
// Wall time (MGM_HIP_STATS=1 prints the breakdown on stderr): a 1920x1080x256 pair is ~17 ms of device work, so the
// program is organised around its HOST costs -- both input images are decoded on their own threads while the main thread
// brings the device context up, and the context is torn down (tens of GB of workspace handed back) on a thread of its
// own while the outputs are encoded and written.
//
// WITH_MGM2=1 (mgm_naive_parallelism, mgm_core.cc:632-831): every pass on its own private Lr volume, all passes in
// flight at once, then the volumes accumulated into S -- which is how the device path is organised anyway
// (DESIGN.md 4).  The reference accumulates in thread-finish order (798-805); here the order is always 0, 1, ..., 7:
// what its default build (Makefile:1, no OpenMP) and any run with one thread compute, and one member of the outcome
// set of a threaded run.  So the flag is accepted and changes nothing.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
    void report()
    {
        if (!on) return;
        fprintf(stderr, "[mgm stats] exec -> main() ~%.0f ms (10 ms ticks);", pre);
        fprintf(stderr, "[mgm stats] ms since main():%s | total %.1f\n", line.c_str(),
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

struct Run;
static void free_run(mgm_ctx *ctx, Run &r);

struct Opts {
    int dmin, dmax, NDIR;
    float P1, P2, aP2, aThresh, truncDist;
    const char *distance, *prefilter, *refine;
    int TSGM, FH, FIX, census_win;
};

static void die(mgm_ctx *ctx, int rc, const char *what)
{
    fprintf(stderr, "mgm: %s failed (%d): %s\n", what, rc, ctx ? mgm_last_error(ctx) : "");
    exit(rc == MGM_ERR_UNSUPPORTED ? 2 : 1);
}

// One run of the path: mgm.cc:372 + 376-385 (u,v) or 373 + 405-414 (v,u with the negated range), in three stages so
// that the two runs of a pair can share one launch of the pass kernel (mgm_aggregate_batch_dev).
struct Run {
    mgm_img *du = nullptr, *dv = nullptr, *dw = nullptr, *dout = nullptr, *dcost = nullptr;
    mgm_cv *C = nullptr;
    bool weighted_msg = false;
    int nx = 0, ny = 0;
};

static void prepare_run(mgm_ctx *ctx, const HostImg &u, const HostImg &v, int dmin, int dmax, const Opts &o, Run &r,
                        const HostImg *lo = nullptr, const HostImg *hi = nullptr)
{
    int rc;
    r.nx = u.nx;
    r.ny = u.ny;
    if ((rc = mgm_img_upload(ctx, u.data.data(), u.nx, u.ny, u.nch, &r.du))) die(ctx, rc, "upload");
    if ((rc = mgm_img_upload(ctx, v.data.data(), v.nx, v.ny, v.nch, &r.dv))) die(ctx, rc, "upload");
    // compute_mgm_weights(u, aP2, aThresh)   [aP1 is parsed and unused in the reference too, mgm.cc:372]
    if ((rc = mgm_weights_dev(ctx, r.du, o.aP2, o.aThresh, &r.dw))) die(ctx, rc, "mgm_weights");
    if (o.aP2 != 1.0f) {  // mgm() announces the weighted mode on stdout (mgm_core.cc:420-423)
        std::vector<float> w((size_t)u.npix() * 8);
        if ((rc = mgm_img_download(ctx, r.dw, w.data()))) die(ctx, rc, "download");
        r.weighted_msg = std::any_of(w.begin(), w.end(), [](float x) { return x != 1.0f; });
    }
    bool ragged = false;
    if (lo) {  // range images (-m/-M, mgm.cc:342-353): Dvec takes them as ints (mgm_costvolume.h:323)
        dmin = (int)lo->data[0];
        dmax = (int)hi->data[0];
        for (size_t i = 0; i < lo->data.size(); i++) ragged |= (int)lo->data[i] != dmin || (int)hi->data[i] != dmax;
    }
    if (ragged) {
        for (size_t i = 0; i < lo->data.size(); i++) {
            dmin = std::min(dmin, (int)lo->data[i]);
            dmax = std::max(dmax, (int)hi->data[i]);
        }
        mgm_img *dlo = nullptr, *dhi = nullptr;
        if ((rc = mgm_img_upload(ctx, lo->data.data(), u.nx, u.ny, 1, &dlo)) || (rc = mgm_img_upload(ctx, hi->data.data(), u.nx, u.ny, 1, &dhi)))
            die(ctx, rc, "upload");
        rc = mgm_costvolume_build_ranged_dev(ctx, r.du, r.dv, dlo, dhi, dmin, dmax, o.prefilter, o.distance, o.truncDist,
                                             o.census_win, &r.C);
        mgm_img_free(ctx, dlo);
        mgm_img_free(ctx, dhi);
        if (rc) die(ctx, rc, "mgm_costvolume_build_ranged");
    } else if ((rc = mgm_costvolume_build_dev(ctx, r.du, r.dv, dmin, dmax, o.prefilter, o.distance, o.truncDist, o.census_win, &r.C))) {
        die(ctx, rc, "mgm_costvolume_build");
    }
    if ((rc = mgm_img_create(ctx, u.nx, u.ny, 1, &r.dout)) || (rc = mgm_img_create(ctx, u.nx, u.ny, 1, &r.dcost)))
        die(ctx, rc, "mgm_img_create");
}

static void aggregate_run(mgm_ctx *ctx, const Opts &o, Run &r)
{
    const int rc = mgm_aggregate_dev(ctx, r.C, r.dw, o.P1, o.P2, o.NDIR, o.TSGM, o.FH, o.FIX, o.refine, r.dout, r.dcost, nullptr);
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
    for (int k = 1; k < n; k++) {
        prepare_run(mgm_multi_ctx(m, k), u, v, dmin, dmax, o, shadow[k]);
        Cs[k] = shadow[k].C;
        Ws[k] = shadow[k].dw;
    }
    const int rc = mgm_multi_aggregate(m, Cs.data(), Ws.data(), o.P1, o.P2, o.NDIR, o.TSGM, o.FH, o.FIX, o.refine, r.dout, r.dcost);
    if (rc) {
        fprintf(stderr, "mgm: mgm_multi_aggregate failed (%d): %s\n", rc, mgm_multi_last_error(m));
        exit(rc == MGM_ERR_UNSUPPORTED ? 2 : 1);
    }
    for (int k = 1; k < n; k++) free_run(mgm_multi_ctx(m, k), shadow[k]);
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

static void free_run(mgm_ctx *ctx, Run &r)
{
    mgm_cv_free(ctx, r.C);
    for (mgm_img *im : {r.du, r.dv, r.dw, r.dout, r.dcost}) mgm_img_free(ctx, im);
    r = Run();
}

// outoff = median_filter(outoff, MEDIAN) on the device (mgm.cc:396, 419), any radius
static void median_run(mgm_ctx *ctx, Run &r, int radius)
{
    mgm_img *tmp = nullptr;
    int rc = mgm_img_create(ctx, r.nx, r.ny, 1, &tmp);
    if (rc) die(ctx, rc, "mgm_img_create");
    if ((rc = mgm_median_dev(ctx, r.dout, radius, tmp))) die(ctx, rc, "mgm_median");
    mgm_img_free(ctx, r.dout);
    r.dout = tmp;
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
    if ((rc = mgm_img_upload(ctx, lo.data(), r.nx, r.ny, 1, &dlo)) || (rc = mgm_img_upload(ctx, hi.data(), r.nx, r.ny, 1, &dhi)))
        die(ctx, rc, "upload");
    for (int it = 1; it < iterations; it++) {
        if ((rc = mgm_update_ranges_dev(ctx, r.dout, dlo, dhi, 3, 2))) die(ctx, rc, "mgm_update_ranges");
        if ((rc = mgm_wta_windowed_dev(ctx, r.C, o.NDIR, o.FIX, o.refine, dlo, dhi, r.dout, r.dcost))) die(ctx, rc, "mgm_wta_windowed");
        report_run(o, r);
    }
    mgm_img_free(ctx, dlo);
    mgm_img_free(ctx, dhi);
}

// The device side of the process: created for the first pair, kept for the following ones (resident mode).
struct Session {
    mgm_ctx *ctx = nullptr;
    mgm_multi *multi = nullptr;
    std::vector<int> devs;
    bool tried = false;
    int rc_ctx = 0;
};

static int run_pair(Session &S, int argc, char **argv, bool last);

// RESIDENT MODE (round 4): `mgm --batch FILE` (FILE = - : stdin).  Every non-empty line of FILE is one command line of this
// program without its name -- options, u, v, out [cost [backflow]] -- and all of them run in THIS process, on one device
// context: HIP start-up (~0.1 s) and the allocation of the tens of GB of workspace are paid once, a further pair costs its
// decoding, ~10-20 ms of device work and its encoding.  Per pair, stdout and every output file are what the one-shot
// command writes (the environment parameters are the process's, the same for all lines); `#` starts a comment.  Exit code:
// 0 if every line succeeded, else 1 (the remaining lines still run).
static int run_batch(const char *file)
{
    FILE *f = strcmp(file, "-") ? fopen(file, "r") : stdin;
    if (!f) {
        fprintf(stderr, "mgm: --batch: cannot open %s\n", file);
        return 1;
    }
    std::vector<std::string> lines;
    char buf[16384];
    while (fgets(buf, sizeof buf, f)) {
        std::string l(buf);
        const size_t h = l.find('#');
        if (h != std::string::npos) l.resize(h);
        if (l.find_first_not_of(" \t\r\n") != std::string::npos) lines.push_back(l);
    }
    if (f != stdin) fclose(f);
    Session S;
    int bad = 0;
    for (size_t k = 0; k < lines.size(); k++) {
        std::vector<std::string> tok{"mgm"};
        size_t i = 0;
        const std::string &l = lines[k];
        while (i < l.size()) {
            while (i < l.size() && strchr(" \t\r\n", l[i])) i++;
            size_t j = i;
            while (j < l.size() && !strchr(" \t\r\n", l[j])) j++;
            if (j > i) tok.push_back(l.substr(i, j - i));
            i = j;
        }
        std::vector<char *> av;
        for (auto &t : tok) av.push_back(&t[0]);
        av.push_back(nullptr);
        if (run_pair(S, (int)tok.size(), av.data(), /*last=*/false) != 0) {  // (the process leaves through _exit: no teardown)
            fprintf(stderr, "mgm: --batch: line %zu failed\n", k + 1);
            bad = 1;
        }
        fflush(stdout);
    }
    return bad;
}

int main(int argc, char **argv)
{
    if (argc >= 3 && !strcmp(argv[1], "--batch")) {
        const int r = run_batch(argv[2]);
        fflush(stdout);
        fflush(stderr);
        _exit(r);  // (as the one-shot run: without the HIP runtime's exit handlers)
    }
    Session S;
    return run_pair(S, argc, argv, true);
}

static int run_pair(Session &S, int argc, char **argv, bool last)
{
    if (argc < 2 || !strcmp(argv[1], "-h")) return 0 * puts("usage:\n\tmgm [-options] u v out [cost [backflow]]");
    if (!strcmp(argv[1], "-?")) return 0 * puts("Compute stereo disparities by the MGM algorithm.");
    if (!strcmp(argv[1], "--version")) return 0 * puts("mgm 2.0 (mgm-hip, MI355X)");
    if (!strcmp(argv[1], "--help"))
        return 0 * puts("mgm [options] in_u in_v out_disp [out_cost [out_backflow]]   (in: png tif pgm ppm pfm npy; out: tif pfm npy)\n"
                        "options: -r dmin(-30) -R dmax(30) -O NDIR(4) -P1 (8) -P2 (32) -p prefilter(none) -t distance(ad)\n"
                        "         -truncDist (inf) -s subpix(none) -aP1 (1) -aP2 (1) -aThresh (5) -m FILE -M FILE -l FILE\n"
                        "environment: CENSUS_NCC_WIN=3 TESTLRRL=1 TESTLRRL_TAU=1.0 MEDIAN=0 TSGM=4 TSGM_ITER=1\n"
                        "             TSGM_FIX_OVERCOUNT=1 USE_TRUNCATED_LINEAR_POTENTIALS=0 MGM_DEVICE=0 MGM_DEVICES=0,1,...\n"
                        "resident mode: mgm --batch FILE|-   (one such command line per line of FILE, one device context for all)");
    if (argc < 4) {
        fprintf(stderr, "too few parameters\n   usage: %s  [-r dmin -R dmax] [-m dminImg -M dmaxImg] [-O NDIR: 2, (4), 8] u v out "
                        "[cost [backflow]]\n", argv[0]);
        return 1;
    }
    // mgm.cc:303-318, in the same order (the order matters for pick_option's argv surgery)
    const char *min_file = pick_option(&argc, argv, "m", "");
    const char *max_file = pick_option(&argc, argv, "M", "");
    Opts o;
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
    const char *nolr_file = pick_option(&argc, argv, "l", "");
    const char *f_u = argc > 1 ? argv[1] : nullptr, *f_v = argc > 2 ? argv[2] : nullptr;
    const char *f_out = argc > 3 ? argv[3] : nullptr, *f_cost = argc > 4 ? argv[4] : nullptr;
    const char *f_back = argc > 5 ? argv[5] : nullptr;
    if (!f_u || !f_v || !f_out) { fprintf(stderr, "too few parameters\n"); return 1; }

    printf("%d %d\n", o.dmin, o.dmax);  // mgm.cc:328
    fflush(stdout);

    o.TSGM = (int)env_param("TSGM", 4);
    o.FH = (int)env_param("USE_TRUNCATED_LINEAR_POTENTIALS", 0);
    o.FIX = (int)env_param("TSGM_FIX_OVERCOUNT", 1);
    o.census_win = (int)env_param("CENSUS_NCC_WIN", 3);
    const double TSGM_ITER = env_param("TSGM_ITER", 1), TESTLRRL = env_param("TESTLRRL", 1);
    const double TAU = env_param("TESTLRRL_TAU", 1.0), MEDIAN = env_param("MEDIAN", 0);
    (void)env_param("WITH_MGM2", 0);  // accepted: see the header comment
    // main()'s loops are `for (int i = 0; i < TSGM_ITER(); i++)` on a DOUBLE (mgm.cc:377, 406): ceil() iterations; none at
    // all for TSGM_ITER <= 0 -- the maps then stay the zero images they were allocated as (mgm.cc:360-365)
    const int ITER = TSGM_ITER > 0 ? (int)std::ceil(TSGM_ITER) : 0;

    Stopwatch sw;
    try {
        // decode both inputs on their own threads while this one brings the device up (process start + HIP initialisation
        // + context is most of a run's wall time: the device work of a full-HD pair is ~17 ms)
        HostImg u, v;
        std::exception_ptr eu, ev;
        std::thread tu([&] { try { u = imgio::read(f_u); remove_nonfinite(u, 0); } catch (...) { eu = std::current_exception(); } });
        std::thread tv([&] { try { v = imgio::read(f_v); remove_nonfinite(v, 0); } catch (...) { ev = std::current_exception(); } });
        int rc;
        if (!S.tried) {  // (resident mode: the first pair brings the device side up, the others find it there)
            S.tried = true;
            if (const char *dl = getenv("MGM_DEVICES"))  // "0,1,2,3": several GPUs of this node
                for (const char *q = dl; *q;) {
                    char *end;
                    const long d = strtol(q, &end, 10);
                    if (end == q) break;
                    S.devs.push_back((int)d);
                    q = *end == ',' ? end + 1 : end;
                }
            if (S.devs.size() > 1 && (ITER > 1 || min_file[0])) {
                fprintf(stderr, "mgm: MGM_DEVICES: TSGM_ITER > 1 and range images run on the first device only\n");
                S.devs.resize(1);
            }
            if (S.devs.size() > 1) {
                if ((S.rc_ctx = mgm_multi_create(S.devs.data(), (int)S.devs.size(), &S.multi)) == 0) S.ctx = mgm_multi_ctx(S.multi, 0);
            } else {
                S.rc_ctx = mgm_ctx_create(S.devs.size() == 1 ? S.devs[0] : (int)env_param("MGM_DEVICE", 0), &S.ctx);
            }
        }
        mgm_ctx *ctx = S.ctx;
        mgm_multi *multi = S.multi;
        if (multi && min_file[0]) {
            fprintf(stderr, "mgm: --batch with MGM_DEVICES: range images are not supported on several devices\n");
            tu.join();
            tv.join();
            return 1;
        }
        const std::vector<int> &devs = S.devs;
        const int rc_ctx = S.rc_ctx;
        sw.mark("context");
        tu.join();
        tv.join();
        sw.mark("decode(rest)");
        if (eu) std::rethrow_exception(eu);
        if (ev) std::rethrow_exception(ev);
        if (rc_ctx && devs.size() > 1) {
            fprintf(stderr, "mgm: MGM_DEVICES: cannot set up %d devices (mgm_multi_create = %d): %s\n", (int)devs.size(), rc_ctx, mgm_multi_last_error(nullptr));
            return 1;
        }
        if (rc_ctx) { fprintf(stderr, "mgm: no usable MI355X device (mgm_ctx_create = %d); there is no CPU path\n", rc_ctx); return 1; }
        HostImg rlo, rhi;  // -m / -M range images of the left->right run (mgm.cc:342-353); the right->left run keeps -r/-R
        if (min_file[0]) {
            rlo = imgio::read(min_file);
            rhi = imgio::read(max_file);
            if (rlo.nx != u.nx || rlo.ny != u.ny || rhi.nx != u.nx || rhi.ny != u.ny || rlo.nch != 1 || rhi.nch != 1) {
                fprintf(stderr, "mgm: the -m/-M images must have the size of the left image\n");
                return 1;
            }
            remove_nonfinite(rlo, (float)o.dmin);
            remove_nonfinite(rhi, (float)o.dmax);
            for (size_t i = 0; i < rlo.data.size(); i++)
                if (rhi.data[i] < rlo.data[i] + 1) rhi.data[i] = ceilf(rlo.data[i] + 1);
        }
        const HostImg *plo = min_file[0] ? &rlo : nullptr, *phi = min_file[0] ? &rhi : nullptr;
        o.P1 *= u.nch;  // mgm.cc:356-357
        o.P2 *= u.nch;


        HostImg outoff, outcost;
        Run L, R;
        prepare_run(ctx, u, v, o.dmin, o.dmax, o, L, plo, phi);
        bool together = false;
        auto zero_run = [&](Run &r) {  // no iteration: mgm() is never called, outoff / outcost keep their zeros
            const std::vector<float> z((size_t)r.nx * r.ny, 0.0f);
            for (mgm_img **im : {&r.dout, &r.dcost}) {
                mgm_img_free(ctx, *im);
                *im = nullptr;
                if ((rc = mgm_img_upload(ctx, z.data(), r.nx, r.ny, 1, im))) die(ctx, rc, "upload");
            }
        };
        if (ITER == 0) {
            zero_run(L);
            together = true;  // (nothing to aggregate)
        }
        if (ITER > 0 && !multi && TESTLRRL != 0 && !plo && u.nx == v.nx && u.ny == v.ny && env_param("MGM_BATCH_LR", 1) != 0) {
            // both runs of the pair (mgm.cc:376-385 and 405-414) through ONE launch of the pass kernel
            prepare_run(ctx, v, u, -o.dmax, -o.dmin, o, R);  // mgm.cc:366, 405
            const mgm_cv *Cs[2] = {L.C, R.C};
            const mgm_img *Ws[2] = {L.dw, R.dw};
            mgm_img *Os[2] = {L.dout, R.dout}, *Cc[2] = {L.dcost, R.dcost};
            rc = mgm_aggregate_batch_dev(ctx, 2, Cs, Ws, o.P1, o.P2, o.NDIR, o.TSGM, o.FH, o.FIX, o.refine, Os, Cc, nullptr);
            if (rc == MGM_OK) together = true;
            else if (rc == MGM_ERR_NOMEM) free_run(ctx, R);  // 2*NDIR Lr volumes do not fit: one run after the other, as the reference does
            else if (rc != MGM_ERR_UNSUPPORTED) die(ctx, rc, "mgm_aggregate_batch");
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
        if (nolr_file[0]) imgio::write(nolr_file, download(ctx, L.dout, L.nx, L.ny, 1));
        if (TESTLRRL != 0) {
            if (!R.C) prepare_run(ctx, v, u, -o.dmax, -o.dmin, o, R);  // mgm.cc:366, 405
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
            mgm_img *Lchk = nullptr, *Rchk = nullptr;
            if ((rc = mgm_img_create(ctx, L.nx, L.ny, 1, &Lchk)) || (rc = mgm_img_create(ctx, R.nx, R.ny, 1, &Rchk)))
                die(ctx, rc, "mgm_img_create");
            if ((rc = mgm_leftright_dev(ctx, R.dout, L.dout, (float)TAU, Rchk)) ||
                (rc = mgm_leftright_dev(ctx, L.dout, R.dout, (float)TAU, Lchk)))
                die(ctx, rc, "mgm_leftright");
            mgm_img_free(ctx, L.dout);
            mgm_img_free(ctx, R.dout);
            L.dout = Lchk;
            R.dout = Rchk;
        }
        sw.mark("enqueue(rest)");
        outoff = download(ctx, L.dout, L.nx, L.ny, 1);
        sw.mark("device+download");
        outcost = download(ctx, L.dcost, L.nx, L.ny, 1);
        // back-projected image (mgm.cc:433-443)
        HostImg syn;
        if (f_back) {
            mgm_img *dsyn = nullptr;
            if ((rc = mgm_img_create(ctx, u.nx, u.ny, u.nch, &dsyn))) die(ctx, rc, "mgm_img_create");
            if ((rc = mgm_backproject_dev(ctx, L.du, L.dv, L.dout, dsyn))) die(ctx, rc, "mgm_backproject");
            syn = download(ctx, dsyn, u.nx, u.ny, u.nch);
            mgm_img_free(ctx, dsyn);
        }
        // the device side is torn down (volumes, tens of GB of workspace) while the outputs are encoded and written
        std::thread teardown([&] {
            free_run(ctx, L);
            free_run(ctx, R);
            if (!last) return;  // (resident mode: the context and its workspace stay for the next pair)
            if (multi) mgm_multi_destroy(multi);
            else mgm_ctx_destroy(ctx);
            S.ctx = nullptr;
            S.multi = nullptr;
        });
        std::exception_ptr ew;
        try {
            imgio::write(f_out, outoff);
            if (f_cost) imgio::write(f_cost, outcost);
            if (f_back) imgio::write(f_back, syn);
        } catch (...) {
            ew = std::current_exception();
        }
        sw.mark("encode+write");
        teardown.join();
        sw.mark("teardown(rest)");
        sw.report();
        if (ew) std::rethrow_exception(ew);
        // Everything is written and the device side is down: leave without the HIP runtime's own exit handlers (unloading
        // code objects, closing the device: tens of ms that produce nothing).  MGM_HIP_ORDERLY_EXIT=1 keeps them.
        fflush(stdout);
        fflush(stderr);
        if (last && !(getenv("MGM_HIP_ORDERLY_EXIT") && atoi(getenv("MGM_HIP_ORDERLY_EXIT")))) _exit(0);
    } catch (const std::exception &e) {
        fprintf(stderr, "mgm: %s\n", e.what());
        return 1;
    }
    return 0;
}
