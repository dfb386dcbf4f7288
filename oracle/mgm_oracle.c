/*
 * mgm_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, dense-layout CPU restatement of the hot path of gfacciol/mgm
 * (cost volume -> multi-direction MGM aggregation -> WTA -> sub-pixel fit).
 * It exists to CHECK the HIP path and to be timed as the "port" CPU baseline;
 * it is never linked into, imported by, or called from the product
 * (mgm_amd/, include/, src/).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit against
 * the compiled reference (oracle/_ref, built from /root/reference by
 * oracle/Makefile) in tests/test_oracle_vs_ref.py, and against the golden
 * vectors under tests/golden/ that were produced by that same reference build
 * (tests/golden/make_golden.py).
 *
 * Layout conventions (the reference's, made dense):
 *   images   : planar float, data[x + y*nx + c*nx*ny]          (img.h:35-51)
 *   volumes  : [y][x][o] fp32, o = 0..L-1 <-> disparity dmin+o  (dvec.cc:49-131,
 *              mgm_costvolume.h:311-327 with a uniform [dmin,dmax] range)
 *   weights  : 8 planes W,E,S,N,NW,NE,SE,SW                     (mgm_weights.h:69)
 *
 * Arithmetic is fp32 with the reference's exact association; compile with
 * -ffp-contract=off and without -ffast-math (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_CENSUS_WORDS 16

/* number of OpenMP threads used by every loop below (default: 1) */
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n < 1 ? 1 : n);
#else
    (void)n;
#endif
}

/* mgm_costvolume.h:16-17 */
#define MIN_(a, b) (((a) < (b)) ? (a) : (b))
#define MAX_(a, b) (((a) > (b)) ? (a) : (b))

/* mgm_core.c:54-60 */
static inline float fmin3(float a, float b, float c)
{
    float m = a;
    if (m > b) m = b;
    if (m > c) m = c;
    return m;
}

/* Check of the HIP kernels' exact division by three (mgm_pass_common.h div3_exact) with C fmaf:
 * counts inputs in [start, end) (stride `step` over float bit patterns) whose
 * q = fma(fma(-3, x*c, x), c, x*c) differs from x/3; -0 and +-inf are excluded
 * (the device finishes with v_div_fixup_f32, tested on the device by mgm_selftest_div3). */
unsigned long long orc_check_div3(unsigned long long start, unsigned long long end, unsigned long long step)
{
    const float c = 1.0f / 3.0f;
    unsigned long long bad = 0;
    for (unsigned long long b = start; b < end; b += step) {
        uint32_t u = (uint32_t)b, a, q;
        float x, ref, q0, r, qq;
        memcpy(&x, &u, 4);
        if (!isfinite(x) || u == 0x80000000u) continue;
        ref = x / 3.0f;
        q0 = x * c;
        r = fmaf(-3.0f, q0, x);
        qq = fmaf(r, c, q0);
        memcpy(&a, &ref, 4);
        memcpy(&q, &qq, 4);
        if (a != q) bad++;
    }
    return bad;
}

/* ------------------------------------------------------------------------ */
/* Census transform: census_tools.cc:16-57, 76-99, 127-153.                  */
/* Bits "centre < neighbour" in order (channel, dy, dx), centre skipped, NaN */
/* outside => 0; packed MSB-first into bytes; bytes laid into 32-bit words   */
/* (little-endian memcpy into floats in the reference).  Output is planar    */
/* [word][y][x] of the raw 32-bit patterns.                                  */
/* ------------------------------------------------------------------------ */
int orc_census_nwords(int nch, int winradius)
{
    int side = 2 * winradius + 1;
    int nbits = nch * (side * side - 1);
    int nbytes = nbits / 8; /* census_tools.cc:81 asserts nbits % 8 == 0 */
    return (nbytes + 3) / 4;
}

int orc_census(const float *u, int nx, int ny, int nch, int winradius, uint32_t *out)
{
    int side = 2 * winradius + 1;
    int nbits = nch * (side * side - 1);
    if (nbits % 8) return -1;
    int nbytes = nbits / 8;
    int nwords = (nbytes + 3) / 4;
    if (nwords > ORC_MAX_CENSUS_WORDS) return -2;
    size_t npix = (size_t)nx * ny;
    for (int y = 0; y < ny; y++)
        for (int x = 0; x < nx; x++) {
            unsigned char bytes[4 * ORC_MAX_CENSUS_WORDS];
            memset(bytes, 0, sizeof bytes);
            int cx = 0;
            for (int l = 0; l < nch; l++)
                for (int j = -winradius; j <= winradius; j++)
                    for (int i = -winradius; i <= winradius; i++) {
                        if (!i && !j) continue;
                        float a = u[x + (size_t)y * nx + l * npix];
                        int xx = x + i, yy = y + j;
                        int bit = 0;
                        if (xx >= 0 && xx < nx && yy >= 0 && yy < ny) {
                            float b = u[xx + (size_t)yy * nx + l * npix];
                            bit = a < b;
                        } /* else b = NaN => a<b false (census_tools.cc:31-33) */
                        bytes[cx >> 3] = (unsigned char)(bytes[cx >> 3] * 2 + bit);
                        cx++;
                    }
            for (int w = 0; w < nwords; w++) {
                uint32_t word;
                memcpy(&word, bytes + 4 * w, 4);
                out[x + (size_t)y * nx + w * npix] = word;
            }
        }
    return nwords;
}

/* ------------------------------------------------------------------------ */
/* Prefilters used with non-census costs: img_tools.h:105-180.               */
/* ------------------------------------------------------------------------ */
static float valneumann(const float *u, int nx, int ny, int x, int y, int c)
{
    int xx = x, yy = y;
    xx = x >= 0 ? xx : 0;
    xx = x < nx ? xx : nx - 1;
    yy = y >= 0 ? yy : 0;
    yy = y < ny ? yy : ny - 1;
    return u[xx + (size_t)yy * nx + (size_t)c * nx * ny];
}

/* apply_filter with a 2-D (single-channel) kernel, img_tools.h:105-127 */
static void apply_filter2d(const float *u, int nx, int ny, int nch, const float *f, int fnx, int fny,
                           float *out)
{
    int hfnx = fnx / 2, hfny = fny / 2;
    for (int c = 0; c < nch; c++)
        for (int j = 0; j < ny; j++)
            for (int i = 0; i < nx; i++) {
                float v = 0;
                for (int jj = 0; jj < fny; jj++)
                    for (int ii = 0; ii < fnx; ii++)
                        v += valneumann(u, nx, ny, i + ii - hfnx, j + jj - hfny, c) * f[ii + jj * fnx];
                out[i + (size_t)j * nx + (size_t)c * nx * ny] = v;
            }
}

/* img_tools.h:140-180 with sigma = 1 */
static void gblur_sigma1(const float *u, int nx, int ny, int nch, float *out)
{
    float sigma = 1.0f;
    float radius = 3 * fabsf(sigma);
    int r = (int)ceil(1 + 2 * radius);
    if (r < 1) r = 1;
    if (r > 39) r = 39;
    float k[39];
    int cw = (r - 1) / 2;
    float m = 0;
    for (int i = 0; i < r; i++) {
        float x = (float)hypot(i - cw, 0);
        float v = (float)exp(-x * x / (2 * sigma * sigma));
        k[i] = v;
        m += v;
    }
    for (int i = 0; i < r; i++) k[i] /= m;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)nx * ny * nch);
    apply_filter2d(u, nx, ny, nch, k, r, 1, tmp);
    apply_filter2d(tmp, nx, ny, nch, k, 1, r, out);
    free(tmp);
}

/* ------------------------------------------------------------------------ */
/* Matching costs: mgm_costvolume.h:23-44 (AD, SD), 65-78 (census).          */
/* ------------------------------------------------------------------------ */
static const unsigned char popcnt8[256] = {
#define B2(n) n, n + 1, n + 1, n + 2
#define B4(n) B2(n), B2(n + 1), B2(n + 1), B2(n + 2)
#define B6(n) B4(n), B4(n + 1), B4(n + 1), B4(n + 2)
    B6(0), B6(1), B6(1), B6(2)};

enum { DIST_AD = 0, DIST_SD = 1, DIST_CENSUS = 2, DIST_NCC = 3, DIST_BTAD = 4, DIST_BTSD = 5 };
enum { PRE_NONE = 0, PRE_CENSUS = 1, PRE_SOBELX = 2, PRE_GBLUR = 3 };

/* name tables with the reference's silent fall-back to index 0:
 * mgm_costvolume.h:170-190, 194-207 */
int orc_distance_index(const char *name)
{
    static const char *t[] = {"ad", "sd", "census", "ncc", "btad", "btsd", 0};
    int r = 0;
    for (int i = 0; t[i]; i++)
        if (!strcmp(name, t[i])) r = i;
    return r;
}
int orc_prefilter_index(const char *name)
{
    static const char *t[] = {"none", "census", "sobelx", "gblur", 0};
    int r = 0;
    for (int i = 0; t[i]; i++)
        if (!strcmp(name, t[i])) r = i;
    return r;
}

static float cost_ad(const float *u, const float *v, int nx, int ny, int vnx, int vny, int nch, int px,
                     int py, int qx, int qy)
{
    /* p is always inside u here; q inside v was checked by the caller */
    (void)ny;
    float tmp = 0;
    for (int t = 0; t < nch; t++) {
        float x = u[px + (size_t)py * nx + (size_t)t * nx * ny] -
                  v[qx + (size_t)qy * vnx + (size_t)t * vnx * vny];
        x = MAX_(x, -x);
        tmp += x;
    }
    return tmp;
}

static float cost_sd(const float *u, const float *v, int nx, int ny, int vnx, int vny, int nch, int px,
                     int py, int qx, int qy)
{
    float tmp = 0;
    for (int t = 0; t < nch; t++) {
        float x = u[px + (size_t)py * nx + (size_t)t * nx * ny] -
                  v[qx + (size_t)qy * vnx + (size_t)t * vnx * vny];
        x = MAX_(x, -x);
        tmp += x * x;
    }
    return tmp;
}

/* byte-LUT popcount of XOR over every byte of every word; "r * 1.0 / u.nch"
 * is a double division narrowed to float on return (mgm_costvolume.h:77). */
static float cost_census(const uint32_t *cu, const uint32_t *cv, int nx, int ny, int vnx, int vny,
                         int nwords, int px, int py, int qx, int qy)
{
    float r = 0;
    for (int t = 0; t < nwords; t++) {
        uint32_t a = cu[px + (size_t)py * nx + (size_t)t * nx * ny];
        uint32_t b = cv[qx + (size_t)qy * vnx + (size_t)t * vnx * vny];
        uint32_t x = a ^ b;
        int cnt = popcnt8[x & 255] + popcnt8[(x >> 8) & 255] + popcnt8[(x >> 16) & 255] + popcnt8[x >> 24];
        r += (float)cnt;
    }
    return (float)(r * 1.0 / nwords);
}

/* Birchfield-Tomasi dissimilarity of one channel, mgm_costvolume.h:82-110: half-sample interpolation along x
 * (the "/2.0" is a double division narrowed back to float), then the symmetric interval distance. */
#define MIN3_(a, b, c) (((a) < (b)) ? (((a) < (c)) ? (a) : (c)) : (((c) < (b)) ? (c) : (b)))
#define MAX3_(a, b, c) (((a) > (b)) ? (((a) > (c)) ? (a) : (c)) : (((c) > (b)) ? (c) : (b)))
static float btad1(const float *u, const float *v, int nx, int ny, int vnx, int vny, int t, int px, int py, int qx, int qy)
{
    const float *pu = u + (size_t)t * nx * ny + (size_t)py * nx;
    const float *pv = v + (size_t)t * vnx * vny + (size_t)qy * vnx;
    float IL = pu[px];
    float ILp = IL, ILm = IL;
    if (px < nx - 1) ILp = (float)((IL + pu[px + 1]) / 2.0);
    if (px >= 1) ILm = (float)((IL + pu[px - 1]) / 2.0);
    float IR = pv[qx];
    float IRp = IR, IRm = IR;
    if (qx < vnx - 1) IRp = (float)((IR + pv[qx + 1]) / 2.0);
    if (qx >= 1) IRm = (float)((IR + pv[qx - 1]) / 2.0);
    float IminR = MIN3_(IRm, IRp, IR);
    float ImaxR = MAX3_(IRm, IRp, IR);
    float IminL = MIN3_(ILm, ILp, IL);
    float ImaxL = MAX3_(ILm, ILp, IL);
    float dLR = MAX3_(0, IL - ImaxR, IminR - IL);
    float dRL = MAX3_(0, IR - ImaxL, IminL - IR);
    float BT = MIN_(dLR, dRL);
    return (float)fabs(BT);
}

/* computeC_BTAD / computeC_BTSD, mgm_costvolume.h:114-135 */
static float cost_bt(const float *u, const float *v, int nx, int ny, int vnx, int vny, int nch, int px, int py, int qx,
                     int qy, int squared)
{
    float val = 0;
    for (int t = 0; t < nch; t++) {
        float x = btad1(u, v, nx, ny, vnx, vny, t, px, py, qx, qy);
        val += squared ? x * x : x;
    }
    return val;
}

/* computeC_clippedNCC, mgm_costvolume.h:137-165: sums in float, the normalisation in double (the 0.0000001 and
 * sqrt are double), a sample outside either image (valnan) or NaN in the window => INFINITY. */
static float cost_ncc(const float *u, const float *v, int nx, int ny, int vnx, int vny, int nch, int px, int py, int qx,
                      int qy, int hwindow)
{
    float NCC = 0;
    for (int t = 0; t < nch; t++) {
        float mu1 = 0, mu2 = 0, s1 = 0, s2 = 0, prod = 0;
        int n = 0;
        for (int i = -hwindow; i <= hwindow; i++)
            for (int j = -hwindow; j <= hwindow; j++) {
                int ax = px + i, ay = py + j, bx = qx + i, by = qy + j;
                if (ax < 0 || ay < 0 || ax >= nx || ay >= ny || bx < 0 || by < 0 || bx >= vnx || by >= vny) return INFINITY;
                float v1 = u[ax + (size_t)ay * nx + (size_t)t * nx * ny];
                float v2 = v[bx + (size_t)by * vnx + (size_t)t * vnx * vny];
                if (isnan(v1) || isnan(v2)) return INFINITY;
                mu1 += v1;
                mu2 += v2;
                s1 += v1 * v1;
                s2 += v2 * v2;
                prod += v1 * v2;
                n++;
            }
        mu1 /= n;
        mu2 /= n;
        s1 /= n;
        s2 /= n;
        prod /= n;
        float var = (s1 - mu1 * mu1) * (s2 - mu2 * mu2);
        double den = (0.0000001 > var) ? 0.0000001 : (double)var;
        NCC = (float)(NCC + (prod - mu1 * mu2) / sqrt(den));
    }
    float m = (NCC < nch) ? NCC : (float)nch;
    float c = (0 > m) ? 0 : m;
    float clipped = nch - c;
    return clipped * 64;
}

/*
 * allocate_and_fill_sgm_costvolume, mgm_costvolume.h:337-424, uniform range.
 * u: nx*ny*nch, v: vnx*vny*nch.  C: [ny][nx][L], L = dmax-dmin+1.
 * `distance`/`prefilter` are indices from orc_*_index().  Note the reference
 * picks the cost FUNCTION before the "census forces both" fix (355 vs 358-362):
 * -p census with -t ad transforms the images but keeps AD on the float-typed
 * census words.
 */
int orc_costvolume(const float *in_u, const float *in_v, int nx, int ny, int nch, int vnx, int vny, int dmin,
                   int dmax, int prefilter, int distance, float truncDist, int census_win, float *C)
{
    int L = dmax - dmin + 1;
    int costfn = distance; /* picked first (355) */
    if (distance == DIST_CENSUS || prefilter == PRE_CENSUS) {
        distance = DIST_CENSUS;
        prefilter = PRE_CENSUS;
    }

    const float *u = in_u, *v = in_v;
    float *fu = 0, *fv = 0;
    uint32_t *cu = 0, *cv = 0;
    int cnch = nch; /* channel count of the prefiltered images (u.nch at 401,405) */

    if (prefilter == PRE_CENSUS) {
        int wr = census_win / 2;
        int nwords = orc_census_nwords(nch, wr);
        cu = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)nx * ny * nwords);
        cv = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)vnx * vny * nwords);
        if (orc_census(in_u, nx, ny, nch, wr, cu) < 0 || orc_census(in_v, vnx, vny, nch, wr, cv) < 0) {
            free(cu);
            free(cv);
            return -11;
        }
        cnch = nwords;
        /* AD/SD on census words reinterpret the words as floats */
        u = (const float *)cu;
        v = (const float *)cv;
    } else if (prefilter == PRE_SOBELX) {
        static const float sob[] = {-1, 0, 1, -2, 0, 2, -1, 0, 1};
        fu = (float *)malloc(sizeof(float) * (size_t)nx * ny * nch);
        fv = (float *)malloc(sizeof(float) * (size_t)vnx * vny * nch);
        apply_filter2d(in_u, nx, ny, nch, sob, 3, 3, fu);
        apply_filter2d(in_v, vnx, vny, nch, sob, 3, 3, fv);
        u = fu;
        v = fv;
    } else if (prefilter == PRE_GBLUR) {
        fu = (float *)malloc(sizeof(float) * (size_t)nx * ny * nch);
        fv = (float *)malloc(sizeof(float) * (size_t)vnx * vny * nch);
        gblur_sigma1(in_u, nx, ny, nch, fu);
        gblur_sigma1(in_v, vnx, vny, nch, fv);
        u = fu;
        v = fv;
    }

    float tr = truncDist * cnch; /* 401, 405 */
#pragma omp parallel for
    for (int jj = 0; jj < ny; jj++)
        for (int ii = 0; ii < nx; ii++) {
            float *Cp = C + ((size_t)jj * nx + ii) * L;
            int allinvalid = 1;
            for (int o = dmin; o <= dmax; o++) {
                int qx = ii + o, qy = jj;
                float e = tr;
                if (qx >= 0 && qy >= 0 && qx < vnx && qy < vny) {
                    if (costfn == DIST_CENSUS)
                        e = cost_census(cu, cv, nx, ny, vnx, vny, cnch, ii, jj, qx, qy);
                    else if (costfn == DIST_AD)
                        e = cost_ad(u, v, nx, ny, vnx, vny, cnch, ii, jj, qx, qy);
                    else if (costfn == DIST_SD)
                        e = cost_sd(u, v, nx, ny, vnx, vny, cnch, ii, jj, qx, qy);
                    else if (costfn == DIST_NCC)
                        e = cost_ncc(u, v, nx, ny, vnx, vny, cnch, ii, jj, qx, qy, census_win / 2);
                    else
                        e = cost_bt(u, v, nx, ny, vnx, vny, cnch, ii, jj, qx, qy, costfn == DIST_BTSD);
                }
                e = MIN_(e, tr);
                Cp[o - dmin] = e;
                if (isfinite(e)) allinvalid = 0;
            }
            if (allinvalid) /* 414-421 */
                for (int o = 0; o < L; o++) Cp[o] = 0;
        }
    free(fu);
    free(fv);
    free(cu);
    free(cv);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Edge weights: mgm_weights.h:26-85.                                        */
/* ------------------------------------------------------------------------ */
void orc_weights(const float *u, int nx, int ny, int nch, float aP, float aThresh, float *w)
{
    static const int sx[8] = {-1, 1, 0, 0, -1, 1, 1, -1};
    static const int sy[8] = {0, 0, 1, -1, -1, -1, 1, 1};
    size_t npix = (size_t)nx * ny;
    for (int o = 0; o < 8; o++)
        for (int j = 0; j < ny; j++)
            for (int i = 0; i < nx; i++) {
                float wvalue = 1.0;
                int qx = i + sx[o], qy = j + sy[o];
                if (qx >= 0 && qy >= 0 && qx < nx && qy < ny) {
                    float d = 0;
                    for (int c = 0; c < nch; c++) {
                        float diff = u[i + (size_t)j * nx + c * npix] - u[qx + (size_t)qy * nx + c * npix];
                        d += diff * diff;
                    }
                    float Delta = d / nch;
                    if (fabs(Delta) < aThresh * aThresh) wvalue = aP;
                    else wvalue = 1;
                }
                w[i + (size_t)j * nx + o * npix] = wvalue;
            }
}

/* ------------------------------------------------------------------------ */
/* MGM aggregation: mgm_core.cc:408-613.                                     */
/* ------------------------------------------------------------------------ */

/* pass table, mgm_core.cc:463-471: dir1..dir4 (dx,dy), inc_x, inc_y, row_major */
typedef struct {
    int d[4][2];
    int inc_x, inc_y, row_major;
} pass_t;
static const pass_t PASSES[8] = {
    {{{-1, 0}, {0, -1}, {-1, -1}, {1, -1}}, 1, 1, 1}, {{{1, 0}, {0, 1}, {1, 1}, {-1, 1}}, 0, 0, 1},
    {{{0, 1}, {-1, 0}, {-1, 1}, {-1, -1}}, 1, 0, 0},  {{{0, -1}, {1, 0}, {1, -1}, {1, 1}}, 0, 1, 0},
    {{{-1, -1}, {1, -1}, {0, -1}, {1, 0}}, 0, 1, 1},  {{{1, -1}, {1, 1}, {1, 0}, {0, 1}}, 0, 0, 0},
    {{{1, 1}, {-1, 1}, {0, 1}, {-1, 0}}, 1, 0, 1},    {{{-1, 1}, {-1, -1}, {-1, 0}, {0, -1}}, 1, 1, 0},
};
/* mgm_core.cc:481-484 */
static const int P2C[4][8] = {
    {0, 1, 2, 3, 4, 5, 6, 7}, {3, 2, 0, 1, 5, 6, 7, 4}, {4, 6, 7, 5, 3, 1, 2, 0}, {5, 7, 4, 6, 1, 2, 0, 3}};

/* Dvec::get_minvalue, dvec.cc:81-88 */
static float slab_min(const float *a, int L)
{
    float m = INFINITY;
    for (int o = 0; o < L; o++)
        if (a[o] < m) m = a[o];
    return m;
}

/* Dvec::operator[], dvec.cc:129: out of range reads +INF */
#define AT(a, o, L) (((o) >= 0 && (o) < (L)) ? (a)[o] : INFINITY)

/* minConvTruncatedLinear, mgm_core.cc:152-163 */
static void minconv(float *M, int mm, float minMall, float P1, float P2)
{
    for (int o = 1; o < mm; o++) M[o] = MIN_(M[o - 1] + P1, M[o]);
    for (int o = mm - 2; o >= 0; o--) M[o] = MIN_(M[o + 1] + P1, M[o]);
    if (P2 < INFINITY)
        for (int o = 0; o < mm; o++) M[o] = MIN_(M[o], minMall + P2);
}

/* One pixel update.  Ln[k] = neighbour slabs, mn[k] = their cached minima,
 * D[k] = weights.  `mode` selects among the four reference functions:
 *   0 update_cost2            (mgm_core.cc:66-90)
 *   1 update_costW            (95-144)
 *   2 update_cost2_trunclinear(197-219; FixBoundary is a no-op for uniform ranges)
 *   3 update_costW_trunclinear(229-281)
 */
static void update_pixel(float *Lp, const float *Cp, const float *const Ln[4], const float mn[4],
                         const float D[4], float P1, float P2, int howmany, int mode, int L, float *scratch)
{
    if (mode == 0) {
        const float *Lq = Ln[0], *Lr = Ln[1];
        float min1 = mn[0], min2 = mn[1];
        for (int o = 0; o < L; o++) {
            float C = Cp[o];
            float vL0 = Lq[o];
            float vLP1 = MIN_(AT(Lq, o - 1, L), AT(Lq, o + 1, L)) + P1;
            float vLP2 = min1 + P2;
            float v2L0 = Lr[o];
            float v2LP1 = MIN_(AT(Lr, o - 1, L), AT(Lr, o + 1, L)) + P1;
            float v2LP2 = min2 + P2;
            float e = 0;
            e += (fmin3(vL0, vLP1, vLP2) - min1) / 2;
            e += (fmin3(v2L0, v2LP1, v2LP2) - min2) / 2;
            Lp[o] = C + e;
        }
    } else if (mode == 1) {
        for (int o = 0; o < L; o++) {
            float C = Cp[o];
            float e = 0;
            for (int k = 0; k < howmany; k++) {
                const float *Lq = Ln[k];
                float vL0 = Lq[o];
                float vLP1 = MIN_(AT(Lq, o - 1, L), AT(Lq, o + 1, L)) + P1 * D[k];
                float vLP2 = mn[k] + P2 * D[k];
                e += fmin3(vL0, vLP1, vLP2) - mn[k];
            }
            Lp[o] = C + e / howmany;
        }
    } else if (mode == 2) {
        float *M1 = scratch, *M2 = scratch + L;
        memcpy(M1, Ln[0], sizeof(float) * L);
        minconv(M1, L, mn[0], P1, P2);
        memcpy(M2, Ln[1], sizeof(float) * L);
        minconv(M2, L, mn[1], P1, P2);
        for (int o = 0; o < L; o++) Lp[o] = Cp[o] + (M1[o] - mn[0] + M2[o] - mn[1]) / 2;
    } else {
        for (int k = 0; k < howmany; k++) {
            float *M = scratch + (size_t)k * L;
            memcpy(M, Ln[k], sizeof(float) * L);
            minconv(M, L, mn[k], P1 * D[k], P2 * D[k]);
        }
        for (int o = 0; o < L; o++) {
            float e = scratch[o] - mn[0];
            for (int k = 1; k < howmany; k++) e += scratch[(size_t)k * L + o] - mn[k];
            Lp[o] = Cp[o] + e / howmany;
        }
    }
}

/*
 * mgm(), mgm_core.cc:408-613, uniform ranges.
 *   C       [ny][nx][L]       (not modified)
 *   w8      8 planes or NULL (=> all ones)
 *   S       [ny][nx][L] out: corrected aggregated volume (returned by mgm())
 *   out     [ny][nx] labels as dmin + argmin; outcost: the minimum
 *   Lr_dump NULL, or NDIR volumes receiving each pass's Lr (test aid)
 * If a pixel has no finite S the reference leaves `minP` uninitialised
 * (mgm_core.cc:594); this restatement writes NaN there.
 */
int orc_mgm(const float *C, int nx, int ny, int L, int dmin, const float *w8, float P1, float P2, int NDIR,
            int MGM, int FH, int FIX, float *S, float *out, float *outcost, float *Lr_dump)
{
    if (NDIR < 1 || NDIR > 8 || MGM < 1 || MGM > 4) return -1;
    size_t npix = (size_t)nx * ny;
    size_t nvol = npix * L;
    int weighted = 0;
    if (w8)
        for (size_t i = 0; i < npix * 8; i++)
            if (w8[i] != 1.0) weighted = 1; /* mgm_core.cc:420-422 */

    int mode;
    if (weighted) mode = FH ? 3 : 1;
    else if (FH) mode = (MGM == 2) ? 2 : 3;
    else mode = (MGM == 2) ? 0 : 1;

    float *Lr = (float *)malloc(sizeof(float) * nvol);
    float *mins = (float *)malloc(sizeof(float) * npix);
    if (!Lr || !mins) return -2;
    memset(S, 0, sizeof(float) * nvol); /* allocate_costvolume zero-initialises (426) */

    for (int pass = 0; pass < NDIR; pass++) {
        pass_t dir = PASSES[pass];
        memcpy(Lr, C, sizeof(float) * nvol); /* 495-498 */
        /* the reference computes minima lazily; frame pixels keep Lr = C */
        int maxii = nx, maxjj = ny;
        if (!dir.row_major) {
            maxii = ny;
            maxjj = nx;
        }
        /* Slope-2 diagonal schedule of mgm_core.cc:505-511.  Any topological
         * order gives the same result; keeping the diagonals lets OpenMP run
         * the same parallel loop as the reference (one team per pass, one
         * barrier per diagonal). */
#pragma omp parallel
        {
            float *scratch = (float *)malloc(sizeof(float) * 4 * (size_t)L);
            for (int ii = 0; ii < maxii + 2 * maxjj; ii++) {
#pragma omp for schedule(static, 1)
                for (int jj = 0; jj < maxjj; jj++) {
                    int x = ii - 2 * jj, y = jj;
                    if (x < 0 || x >= maxii) continue;
                    int maxnx = maxii, maxny = maxjj;
                    if (!dir.row_major) {
                        int t = x;
                        x = y;
                        y = t;
                        t = maxnx;
                        maxnx = maxny;
                        maxny = t;
                    }
                    if (dir.inc_x == 0) x = (maxnx - 1) - x;
                    if (dir.inc_y == 0) y = (maxny - 1) - y;
                    size_t pidx = (size_t)x + (size_t)y * nx;
                    size_t nidx[4];
                    int inside = 1;
                    for (int k = 0; k < 4; k++) {
                        int qx = x + dir.d[k][0], qy = y + dir.d[k][1];
                        if (!(qx >= 0 && qy >= 0 && qx < nx && qy < ny)) inside = 0;
                        nidx[k] = (size_t)qx + (size_t)qy * nx;
                    }
                    if (inside) { /* 538-541: all four, whatever MGM is */
                        const float *Ln[4];
                        float mn[4], D[4];
                        for (int k = 0; k < 4; k++) {
                            Ln[k] = Lr + nidx[k] * L;
                            mn[k] = (k < MGM) ? mins[nidx[k]] : INFINITY;
                            D[k] = weighted ? w8[pidx + (size_t)P2C[k][pass] * npix] : 1.0f;
                        }
                        update_pixel(Lr + pidx * L, C + pidx * L, Ln, mn, D, P1, P2, MGM, mode, L, scratch);
                    }
                    mins[pidx] = slab_min(Lr + pidx * L, L); /* 577 */
                } /* implicit barrier: one diagonal completes before the next starts */
            }
            free(scratch);
        }
        if (Lr_dump) memcpy(Lr_dump + (size_t)pass * nvol, Lr, sizeof(float) * nvol);
#pragma omp parallel for
        for (size_t i = 0; i < nvol; i++) S[i] += Lr[i]; /* 582-587 */
    }

    /* over-count fix + WTA, 592-609 */
#pragma omp parallel for
    for (size_t i = 0; i < npix; i++) {
        float minP = NAN;
        float minL = INFINITY;
        float *Si = S + i * L;
        const float *Ci = C + i * L;
        for (int o = 0; o < L; o++) {
            if (FIX == 1) Si[o] = Si[o] - (NDIR - 1) * Ci[o];
            if (isfinite(Si[o]))
                if (minL > Si[o]) {
                    minL = Si[o];
                    minP = (float)(o + dmin);
                }
        }
        out[i] = minP;
        outcost[i] = minL;
    }
    free(Lr);
    free(mins);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* RAGGED ranges (round 6): the same path with a per-pixel Dvec range.        */
/*                                                                          */
/* Containers: the reference allocates one Dvec per pixel over [min(p),     */
/* max(p)] = ((int)dminI[p], (int)dmaxI[p]) (mgm_costvolume.h:311-327,      */
/* dvec.cc:55-64) and every read outside a Dvec's own range gives +INF      */
/* (dvec.cc:129).  Here a volume is the dense HULL [ny][nx][L], label o <->  */
/* disparity dmin + o, with integer images lo/hi (absolute disparities,      */
/* dmin <= lo <= hi <= dmin+L-1): a pixel OWNS lo..hi, every other slot of    */
/* its slab holds +INF -- which is what a neighbour reads there.             */
/* ------------------------------------------------------------------------ */

/* allocate_and_fill_sgm_costvolume (mgm_costvolume.h:337-424) with range images.  Slots a pixel does not own are
 * written +INF and take no part in the "no finite hypothesis => zeros" rule (414-421 loops over the pixel's Dvec). */
int orc_costvolume_ranged(const float *in_u, const float *in_v, int nx, int ny, int nch, int vnx, int vny, int dmin,
                          int L, const int *lo, const int *hi, int prefilter, int distance, float truncDist,
                          int census_win, float *C)
{
    /* A cost depends on (p, o) alone, so the uniform restatement over the hull gives every label's cost (steps 0-2 and
     * 4.1-4.3); the per-pixel rule (414-421) is then applied to each pixel's OWN range.  The hull call has applied that rule
     * hull-wide: a slab it zeroed had no finite cost anywhere, so the own range has none either and is zeroed too. */
    int r = orc_costvolume(in_u, in_v, nx, ny, nch, vnx, vny, dmin, dmin + L - 1, prefilter, distance, truncDist,
                           census_win, C);
    if (r) return r;
    for (int jj = 0; jj < ny; jj++)
        for (int ii = 0; ii < nx; ii++) {
            size_t p = (size_t)jj * nx + ii;
            float *Cp = C + p * L;
            int a = lo[p] - dmin, b = hi[p] - dmin;
            if (a < 0 || b >= L || a > b) return -21;
            int allinvalid = 1;
            for (int o = a; o <= b; o++)
                if (isfinite(Cp[o])) allinvalid = 0;
            if (allinvalid)
                for (int o = a; o <= b; o++) Cp[o] = 0;
            for (int o = 0; o < a; o++) Cp[o] = INFINITY;
            for (int o = b + 1; o < L; o++) Cp[o] = INFINITY;
        }
    return 0;
}

/* FixBounrady_for_minConvTruncatedLinear, mgm_core.cc:166-186.  Lq = the neighbour's hull slab (index = label - dmin),
 * [imin, imax] its own range, M the receiver's working row over [mmin, mmax]; all as hull indices. */
static void fix_boundary(const float *Lq, int imin, int imax, float *M, int mmin, int mmax, float P1)
{
    if (imin < mmin) {
        float T = Lq[imin];
        for (int o = imin + 1; o <= mmin; o++) {
            float Inext = o <= imax ? Lq[o] : INFINITY;
            T = MIN_(T + P1, Inext);
        }
        M[0] = MIN_(M[0], T);
    }
    if (imax > mmax) {
        float T = Lq[imax];
        for (int o = imax - 1; o >= mmax; o--) {
            float Inext = o >= imin ? Lq[o] : INFINITY;
            T = MIN_(T + P1, Inext);
        }
        M[mmax - mmin] = MIN_(M[mmax - mmin], T);
    }
}

/* update_pixel for a receiver that owns [a, b] (hull indices); qa/qb = the neighbours' own ranges.  Neighbour slabs hold
 * +INF outside their ranges, so Lq[o] and AT(Lq, o+-1) are Dvec::operator[] (dvec.cc:129). */
static void update_pixel_r(float *Lp, const float *Cp, const float *const Ln[4], const float mn[4], const float D[4],
                           float P1, float P2, int howmany, int mode, int L, int a, int b, const int qa[4],
                           const int qb[4], float *scratch)
{
    int NN = b - a + 1;
    if (mode == 0) { /* update_cost2, mgm_core.cc:66-90 */
        const float *Lq = Ln[0], *Lr = Ln[1];
        float min1 = mn[0], min2 = mn[1];
        for (int o = a; o <= b; o++) {
            float C = Cp[o];
            float vL0 = Lq[o];
            float vLP1 = MIN_(AT(Lq, o - 1, L), AT(Lq, o + 1, L)) + P1;
            float vLP2 = min1 + P2;
            float v2L0 = Lr[o];
            float v2LP1 = MIN_(AT(Lr, o - 1, L), AT(Lr, o + 1, L)) + P1;
            float v2LP2 = min2 + P2;
            float e = 0;
            e += (fmin3(vL0, vLP1, vLP2) - min1) / 2;
            e += (fmin3(v2L0, v2LP1, v2LP2) - min2) / 2;
            Lp[o] = C + e;
        }
    } else if (mode == 1) { /* update_costW, 95-144 */
        for (int o = a; o <= b; o++) {
            float C = Cp[o];
            float e = 0;
            for (int k = 0; k < howmany; k++) {
                const float *Lq = Ln[k];
                float vL0 = Lq[o];
                float vLP1 = MIN_(AT(Lq, o - 1, L), AT(Lq, o + 1, L)) + P1 * D[k];
                float vLP2 = mn[k] + P2 * D[k];
                e += fmin3(vL0, vLP1, vLP2) - mn[k];
            }
            Lp[o] = C + e / howmany;
        }
    } else if (mode == 2) { /* update_cost2_trunclinear, 197-219: the only caller of the boundary fix-up */
        float *M1 = scratch, *M2 = scratch + L;
        memcpy(M1, Ln[0] + a, sizeof(float) * NN);
        fix_boundary(Ln[0], qa[0], qb[0], M1, a, b, P1);
        minconv(M1, NN, mn[0], P1, P2);
        memcpy(M2, Ln[1] + a, sizeof(float) * NN);
        fix_boundary(Ln[1], qa[1], qb[1], M2, a, b, P1);
        minconv(M2, NN, mn[1], P1, P2);
        for (int o = a; o <= b; o++) Lp[o] = Cp[o] + (M1[o - a] - mn[0] + M2[o - a] - mn[1]) / 2;
    } else { /* update_costW_trunclinear, 229-281: convolution over the receiver's range, no fix-up */
        for (int k = 0; k < howmany; k++) {
            float *M = scratch + (size_t)k * L;
            memcpy(M, Ln[k] + a, sizeof(float) * NN);
            minconv(M, NN, mn[k], P1 * D[k], P2 * D[k]);
        }
        for (int o = a; o <= b; o++) {
            float e = scratch[o - a] - mn[0];
            for (int k = 1; k < howmany; k++) e += scratch[(size_t)k * L + (o - a)] - mn[k];
            Lp[o] = Cp[o] + e / howmany;
        }
    }
}

/*
 * mgm(), mgm_core.cc:408-613, with range images.
 *   C        [ny][nx][L] hull; only the slots a pixel owns (lo..hi) are read
 *   lo, hi   the ranges CC was allocated with (absolute disparities)
 *   slo, shi NULL, or the range images mgm() itself is called with when they differ from CC's (main()'s TSGM_ITER loop,
 *            mgm.cc:377-388: CC keeps its ranges, S is allocated from the narrowed images, mgm_core.cc:426); with
 *            sdmin/sL = the hull S is returned on
 *   S        NULL or [ny][nx][sL]: the corrected aggregated volume; slots outside a pixel's S-range hold +INF
 *   out/outcost as orc_mgm; Lr_dump NULL, or one hull volume (+INF outside a pixel's range) per pass whose bit is set in
 *            dump_mask, in pass order (a full-size test keeps two of the eight)
 * S[i] accumulates Lr[i][o] for o in Lr[i]'s range that also lie in S[i]'s (increment_nolock drops the others,
 * dvec.cc:117-125); the search runs over S[i]'s range with CC[i][o] = +INF where CC does not own o (592-609).
 */
int orc_mgm_ranged(const float *C, int nx, int ny, int L, int dmin, const int *lo, const int *hi, const int *slo,
                   const int *shi, int sdmin, int sL, const float *w8, float P1, float P2, int NDIR, int MGM, int FH,
                   int FIX, float *S, float *out, float *outcost, float *Lr_dump, unsigned dump_mask)
{
    if (NDIR < 1 || NDIR > 8 || MGM < 1 || MGM > 4) return -1;
    int ndumped = 0;
    if (!slo || !shi) {
        slo = lo;
        shi = hi;
        sdmin = dmin;
        sL = L;
    }
    size_t npix = (size_t)nx * ny;
    size_t nvol = npix * L;
    for (size_t i = 0; i < npix; i++) {
        if (lo[i] < dmin || hi[i] > dmin + L - 1 || lo[i] > hi[i]) return -21;
        if (slo[i] < sdmin || shi[i] > sdmin + sL - 1 || slo[i] > shi[i]) return -22;
    }
    int weighted = 0;
    if (w8)
        for (size_t i = 0; i < npix * 8; i++)
            if (w8[i] != 1.0) weighted = 1; /* mgm_core.cc:420-422 */

    int mode;
    if (weighted) mode = FH ? 3 : 1;
    else if (FH) mode = (MGM == 2) ? 2 : 3;
    else mode = (MGM == 2) ? 0 : 1;

    float *Cm = (float *)malloc(sizeof(float) * nvol); /* C with +INF where a pixel does not own the label */
    float *Lr = (float *)malloc(sizeof(float) * nvol);
    float *mins = (float *)malloc(sizeof(float) * npix);
    float *Sx = (float *)malloc(sizeof(float) * npix * (size_t)sL);
    if (!Cm || !Lr || !mins || !Sx) return -2;
    for (size_t i = 0; i < npix; i++) {
        int a = lo[i] - dmin, b = hi[i] - dmin;
        for (int o = 0; o < L; o++) Cm[i * L + o] = (o >= a && o <= b) ? C[i * L + o] : INFINITY;
    }
    memset(Sx, 0, sizeof(float) * npix * (size_t)sL); /* allocate_costvolume zero-initialises (426) */

    for (int pass = 0; pass < NDIR; pass++) {
        pass_t dir = PASSES[pass];
        memcpy(Lr, Cm, sizeof(float) * nvol); /* 495-498 */
        int maxii = nx, maxjj = ny;
        if (!dir.row_major) {
            maxii = ny;
            maxjj = nx;
        }
#pragma omp parallel
        {
            float *scratch = (float *)malloc(sizeof(float) * 4 * (size_t)L);
            for (int ii = 0; ii < maxii + 2 * maxjj; ii++) {
#pragma omp for schedule(static, 1)
                for (int jj = 0; jj < maxjj; jj++) {
                    int x = ii - 2 * jj, y = jj;
                    if (x < 0 || x >= maxii) continue;
                    int maxnx = maxii, maxny = maxjj;
                    if (!dir.row_major) {
                        int t = x;
                        x = y;
                        y = t;
                        t = maxnx;
                        maxnx = maxny;
                        maxny = t;
                    }
                    if (dir.inc_x == 0) x = (maxnx - 1) - x;
                    if (dir.inc_y == 0) y = (maxny - 1) - y;
                    size_t pidx = (size_t)x + (size_t)y * nx;
                    size_t nidx[4];
                    int inside = 1;
                    for (int k = 0; k < 4; k++) {
                        int qx = x + dir.d[k][0], qy = y + dir.d[k][1];
                        if (!(qx >= 0 && qy >= 0 && qx < nx && qy < ny)) inside = 0;
                        nidx[k] = (size_t)qx + (size_t)qy * nx;
                    }
                    if (inside) { /* 538-541 */
                        const float *Ln[4];
                        float mn[4], D[4];
                        int qa[4], qb[4];
                        for (int k = 0; k < 4; k++) {
                            Ln[k] = Lr + nidx[k] * L;
                            mn[k] = (k < MGM) ? mins[nidx[k]] : INFINITY;
                            D[k] = weighted ? w8[pidx + (size_t)P2C[k][pass] * npix] : 1.0f;
                            qa[k] = lo[nidx[k]] - dmin;
                            qb[k] = hi[nidx[k]] - dmin;
                        }
                        update_pixel_r(Lr + pidx * L, Cm + pidx * L, Ln, mn, D, P1, P2, MGM, mode, L, lo[pidx] - dmin,
                                       hi[pidx] - dmin, qa, qb, scratch);
                    }
                    /* Dvec::get_minvalue over the pixel's own range (dvec.cc:81-88) */
                    mins[pidx] = slab_min(Lr + pidx * L + (lo[pidx] - dmin), hi[pidx] - lo[pidx] + 1);
                }
            }
            free(scratch);
        }
        if (Lr_dump && ((dump_mask >> pass) & 1u)) memcpy(Lr_dump + (size_t)(ndumped++) * nvol, Lr, sizeof(float) * nvol);
#pragma omp parallel for
        for (size_t i = 0; i < npix; i++) /* 582-587 */
            for (int d = lo[i]; d <= hi[i]; d++)
                if (d >= slo[i] && d <= shi[i]) Sx[i * sL + (d - sdmin)] += Lr[i * L + (d - dmin)];
    }

#pragma omp parallel for
    for (size_t i = 0; i < npix; i++) { /* 592-609 */
        float minP = NAN;
        float minL = INFINITY;
        float *Si = Sx + i * sL;
        for (int d = slo[i]; d <= shi[i]; d++) {
            float Cd = (d >= lo[i] && d <= hi[i]) ? C[i * L + (d - dmin)] : INFINITY;
            if (FIX == 1) Si[d - sdmin] = Si[d - sdmin] - (NDIR - 1) * Cd;
            if (isfinite(Si[d - sdmin]))
                if (minL > Si[d - sdmin]) {
                    minL = Si[d - sdmin];
                    minP = (float)d;
                }
        }
        out[i] = minP;
        outcost[i] = minL;
        if (S)
            for (int o = 0; o < sL; o++)
                S[i * sL + o] = (o + sdmin >= slo[i] && o + sdmin <= shi[i]) ? Si[o] : INFINITY;
    }
    free(Cm);
    free(Lr);
    free(mins);
    free(Sx);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Sub-pixel refinement: mgm_refine.h:40-70 + refine.h.                      */
/* ------------------------------------------------------------------------ */
enum { REF_NONE = 0, REF_VFIT = 1, REF_PARABOLA = 2, REF_CUBIC = 3, REF_PARABOLA_OCV = 4 };

int orc_refinement_index(const char *name) /* mgm_refine.h:28-35 */
{
    static const char *t[] = {"none", "vfit", "parabola", "cubic", "parabolaOCV", ""};
    int r = 0;
    for (int i = 0; i < 6; i++)
        if (!strcmp(name, t[i])) r = i;
    return r == 5 ? 0 : r; /* the "" sentinel row has f == NULL => none */
}

/* refine.h:70-92 */
static void vfit(const float v[3], float *v_min, float *x_min)
{
    if ((v[1] > v[0]) && (v[1] > v[2])) {
        *v_min = v[1];
        *x_min = 0;
        return;
    }
    float slope = v[2] - v[1];
    if ((v[2] - v[1]) < (v[0] - v[1])) slope = v[0] - v[1];
    *x_min = (v[0] - v[2]) / (2 * slope);
    *v_min = v[2] + (*x_min - 1) * slope;
}

/* refine.h:40-68 */
static void parabolafit(const float v[3], float *v_min, float *x_min)
{
    if (v[1] > v[0] && v[1] > v[2]) {
        *x_min = 0;
        *v_min = v[1];
        return;
    }
    float c = v[1];
    float b = (v[2] - v[0]) / 2;
    float a = (v[2] - 2 * v[1] + v[0]) / 2;
    float x = -b / (2 * a);
    if (x > 1) x = 1;
    if (x < -1) x = -1;
    *v_min = (a * x + b) * x + c;
    *x_min = x;
}

/* refine.h:6-38 (including the "don't make any sense" lines, verbatim maths) */
static void parabolafit_ocv(const float v[3], float *v_min, float *x_min)
{
    if (v[1] > v[0] && v[1] > v[2]) {
        *x_min = 0;
        *v_min = v[1];
        return;
    }
    float c = v[1];
    float b = (v[2] - v[0]) / 2;
    float a = (v[2] - 2 * v[1] + v[0]) / 2;
    a *= 2;
    b *= 2;
    a = a > 1.0 ? a : 1.0;
    float x = (-b + a) / (2 * a);
    if (x > 1) x = 1;
    if (x < -1) x = -1;
    *v_min = (a * x + b) * x + c;
    *x_min = x;
}

/* refine.h:94-99: evaluated in double (0.5, 2.0 ... literals), narrowed on return */
static float cubic_interp(const float p[4], const float x)
{
    return p[1] + 0.5 * x *
                      (p[2] - p[0] + x * (2.0 * p[0] - 5.0 * p[1] + 4.0 * p[2] - p[3] + x * (3.0 * (p[1] - p[2]) + p[3] - p[0])));
}

/* refine.h:102-145.  NOTE the reference receives v = {S[o-1],S[o],S[o+1],S[o+2]}
 * and the cubic treats p[1]..p[2] as the unit interval. */
static void cubicfit(const float p[4], float *out_pmin, float *out_xmin)
{
    float pmin, xmin;
    if (p[1] < p[2]) {
        pmin = p[1];
        xmin = 0.0;
    } else {
        pmin = p[2];
        xmin = 1.0;
    }
    double a, b, c, z1, z2, discr;
    a = 0.5 * 3.0 * (3.0 * (p[1] - p[2]) + p[3] - p[0]);
    b = 2.0 * p[0] - 5.0 * p[1] + 4.0 * p[2] - p[3];
    c = 0.5 * (p[2] - p[0]);
    discr = b * b - 4.0 * a * c;
    if (discr >= 0) {
        z1 = (-b + sqrt(discr)) / (2.0 * a);
        z2 = (-b - sqrt(discr)) / (2.0 * a);
        if (z1 > 0.0 && z1 < 1.0) {
            float tmp = cubic_interp(p, z1);
            if (tmp < pmin) {
                pmin = tmp;
                xmin = z1;
            }
        }
        if (z2 > 0.0 && z2 < 1.0) {
            float tmp = cubic_interp(p, z2);
            if (tmp < pmin) {
                pmin = tmp;
                xmin = z2;
            }
        }
    }
    *out_pmin = pmin;
    *out_xmin = xmin;
}

/*
 * subpixel_refinement_sgm (mgm_refine.h:40-70) for a uniform range.  S is the
 * CORRECTED volume returned by orc_mgm.  The "o-1 >= min && o+2 <= max" gate
 * applies to every method (mgm_refine.h:58).
 */
int orc_refine(const float *S, int nx, int ny, int L, int dmin, int method, float *out, float *outcost)
{
    if (method == REF_NONE) return 0;
    if (method < 0 || method > REF_PARABOLA_OCV) return -1;
    size_t N = (size_t)nx * ny;
#pragma omp parallel for
    for (size_t i = 0; i < N; i++) {
        float minP = out[i];
        float minL = outcost[i];
        if (!(minP == minP)) continue; /* NaN label: UB in the reference */
        int o = (int)minP;
        if (o - 1 >= dmin && o + 2 <= dmin + L - 1) {
            const float *Si = S + i * L + (o - dmin);
            float v[4] = {Si[-1], Si[0], Si[1], Si[2]};
            float dx = 0;
            if (method == REF_VFIT) vfit(v, &minL, &dx);
            else if (method == REF_PARABOLA) parabolafit(v, &minL, &dx);
            else if (method == REF_CUBIC) cubicfit(v, &minL, &dx);
            else parabolafit_ocv(v, &minL, &dx);
            minP = o + dx;
        }
        out[i] = minP;
        outcost[i] = minL;
    }
    return 0;
}

/* subpixel_refinement_sgm (mgm_refine.h:40-70) on a ragged S: the gate is the pixel's own S range (58). */
int orc_refine_ranged(const float *S, int nx, int ny, int L, int dmin, const int *lo, const int *hi, int method,
                      float *out, float *outcost)
{
    if (method == REF_NONE) return 0;
    if (method < 0 || method > REF_PARABOLA_OCV) return -1;
    size_t N = (size_t)nx * ny;
#pragma omp parallel for
    for (size_t i = 0; i < N; i++) {
        float minP = out[i];
        float minL = outcost[i];
        if (!(minP == minP)) continue; /* NaN label: UB in the reference */
        int o = (int)minP;
        if (o - 1 >= lo[i] && o + 2 <= hi[i]) {
            const float *Si = S + i * L + (o - dmin);
            float v[4] = {Si[-1], Si[0], Si[1], Si[2]};
            float dx = 0;
            if (method == REF_VFIT) vfit(v, &minL, &dx);
            else if (method == REF_PARABOLA) parabolafit(v, &minL, &dx);
            else if (method == REF_CUBIC) cubicfit(v, &minL, &dx);
            else parabolafit_ocv(v, &minL, &dx);
            minP = o + dx;
        }
        out[i] = minP;
        outcost[i] = minL;
    }
    return 0;
}
