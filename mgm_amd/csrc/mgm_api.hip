// mgm_api.hip -- the C ABI of libmgm_hip.so (include/mgm_hip.h), compute side: cost volumes, edge weights, the aggregation
// calls (immediate, batched, deferred), the direction-sharding building blocks and the steps main() applies around the path.
// Contexts and containers: mgm_ctx.hip; the launch plan: mgm_plan.hip.  No compute happens on the host and there is no CPU
// fallback.
#include "mgm_host.h"

extern "C" {

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
    if (*out) {
        // A provided volume whose refill failed after its state had been touched (a reservation that ran out of memory, a
        // kernel launch error) must not pass for a filled one: no compact copy, no "NaN-free" verdict, and the mark
        // that makes mgm_aggregate* refuse it.
        (*out)->unfilled = r != MGM_OK;
        if (r != MGM_OK) {
            (*out)->c8_state = -1;
            (*out)->p8_state = 0;
            (*out)->nan_state = 0;
        }
    }
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

    if (pre == 1) {  // (checked before a provided volume is touched: a bad window leaves it as it was)
        const int wr0 = census_win / 2, side0 = 2 * wr0 + 1, nbits0 = u->nch * (side0 * side0 - 1);
        if (wr0 < 1 || nbits0 % 8)  // census_tools.cc:81 asserts this
            return fail(c, MGM_ERR_INVALID, "census: nch*(win*win-1) must be a positive multiple of 8");
        if ((nbits0 / 8 + 3) / 4 > kCensusMaxWords) return fail(c, MGM_ERR_UNSUPPORTED, "census descriptor longer than 256 bits");
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
    (*out)->p8_state = 0;
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
    // Neither is one attempted for census over several descriptor words (thirds or halves of bit counts), for differences of
    // blurred images or of descriptor words read as floats, nor for a volume whose last two fillings did not fit: those
    // write the fp32 volume alone (k_cost_btx).
    const int census_words = pre == 1 ? (u->nch * ((census_win / 2 * 2 + 1) * (census_win / 2 * 2 + 1) - 1) / 8 + 3) / 4 : 0;
    const bool may_be_integer = costfn == 2 ? census_words == 1 : (costfn <= 1 && (pre == 0 || pre == 2) && (*out)->diff_fails < 2);
    // Which compact form: census costs are bit counts (one byte); absolute differences of a one-channel 8-bit pair stay below
    // 256, of a colour pair below 766, squared differences below 65026 per channel: two bytes (up to 512 labels: the pass
    // kernels that read them).  The flag word tells afterwards whether every cost really had the form.
    const int cb = (costfn == 2 || (costfn == 0 && u->nch == 1 && !(*out)->diff_wide) || dmax - dmin + 1 > 512) ? 1 : 2;
    // (a ragged single-word census volume first tries to be written as its range-proportional copy ALONE, below: no compact hull
    // is allocated for it unless that fails)
    const bool rel_direct_candidate = rloI && costfn == 2 && census_words == 1 && rel_enabled() && dev().lazy_f32 && tune_num("rel_direct", 1) != 0 &&
                                      (truncDist == __builtin_huge_valf() || (truncDist >= 0.0f && truncDist <= 254.0f && truncDist == rintf(truncDist)));
    auto setup_c8 = [&]() -> int {
        if (c8_supported(dmax - dmin + 1) && dev().c8) {
            if (may_be_integer) {
                if (int rr = c8_alloc(c, *out, cb)) return rr;
                p.C8 = (*out)->d8;
                p.cbytes = cb;
                (*out)->c8_state = 1;
            } else
                (*out)->c8_state = -1;
        }
        return MGM_OK;
    };
    if (!rel_direct_candidate && (r = setup_c8())) return r;
    (*out)->f32_state = 1;
    p.nx = u->nx;
    p.ny = u->ny;
    p.vnx = v->nx;
    p.vny = v->ny;
    p.dmin = dmin;
    p.L = dmax - dmin + 1;
    p.Lreal = p.L;
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
    if (costfn >= 4 && pre == 0 && !rloI) {
        // Birchfield-Tomasi on the plain images: the interval every sample spans is computed once (k_bt_spans), in the census
        // buffers, which this combination leaves free
        if ((r = reserve(c, c->census_u, sizeof(float) * (size_t)u->nx * u->ny * 2 * u->nch))) return r;
        if ((r = reserve(c, c->census_v, sizeof(float) * (size_t)v->nx * v->ny * 2 * v->nch))) return r;
        p.ncc_u = (float *)c->census_u.p;
        p.ncc_v = (float *)c->census_v.p;
    }
    p.trunc = truncDist * (float)p.nch;  // mgm_costvolume.h:401,405
    // A label count that the pass kernels run padded (151 -> 192 slots, ...): the same two families of costs write the PADDED
    // compact copy themselves (mgm_cv::p8; the slots beyond the real count +INF) instead of an fp32 volume that every
    // aggregation call would pad and encode again.  The flag word is read back at once; a volume that does not fit takes
    // the general kernel below, fp32 volume and all, and so do its refills.
    const int LP = (!c8_supported(p.L) && dev().c8 && dev().pad && dev().lazy_f32 && !p.rlo) ? padded_labels(p.L) : 0;
    (*out)->rel_only = false;
    const bool census_fits = costfn == 2 && p.nch == 1 &&
                             (p.trunc == __builtin_huge_valf() || (p.trunc >= 0.0f && p.trunc <= 254.0f && p.trunc == rintf(p.trunc)));
    const bool diff_may_fit = (costfn == 0 || costfn == 1) && (pre == 0 || pre == 2) && p.trunc >= 0.0f && !std::signbit(p.trunc) &&
                              (long long)u->nx * u->ny < 0x7fffffffll;  // (what k_cost_diffx takes)
    // A ragged single-word census volume: the RANGE-PROPORTIONAL copy alone, straight from the descriptor words (mgm_pass_rel.hip,
    // k_cost_census_rel) -- neither the fp32 hull nor its compact twin is written; whoever wants the hull gets it from
    // ensure_f32.  The flag word (a window wider than 62 labels) is read back at once: such a volume takes the general path.
    if (rel_direct_candidate && p.rlo && census_fits) {
        for (int slots = (*out)->rel_hint_slots == 128 ? 128 : 64; slots <= 128; slots *= 2) {  // (round 6: windows of up to 62 labels in 64 slots per pixel, else up to 126 in 128)
            if ((r = rel_alloc(c, *out, slots, 1))) return r;
            if (!(*out)->relbuf) break;
            unsigned *flag = (*out)->rel_flag();
            HIPCHK(c, hipMemsetAsync(flag, 0, 4, c->stream));
            {
                TimeScope t(c, "k_cost");
                HIPCHK(c, launch_cost_census_rel(p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, dmin, p.L, p.trunc, p.rlo, p.rhi, slots, (*out)->relbuf, (*out)->rel_records(),
                                                 flag, c->stream));
            }
            if ((r = ensure_words(c))) return r;
            HIPCHK(c, hipMemcpyAsync(c->h_words + 3, flag, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->h_words[3] == 0u) {
                (*out)->rel_hint_slots = slots;
                (*out)->rel_state = 2;
                (*out)->rel_only = true;
                (*out)->f32_state = 0;
                (*out)->c8_state = 0;   // (no compact hull either: c8_resolve makes one from the expanded volume if a dense launch wants it)
                (*out)->nan_state = 2;  // integer costs: NaN-free by construction
                return MGM_OK;
            }
            if (tune_num("rel_wide", 1) == 0) break;
        }
        (*out)->rel_state = 0;
    }
    if (rel_direct_candidate && (r = setup_c8())) return r;  // (it did not work out: a window wider than 62 labels -- the general path)
    if (LP && (*out)->diff_fails < 2 && (census_fits || diff_may_fit)) {
        int pcb = (costfn == 2 || (costfn == 0 && u->nch == 1 && !(*out)->diff_wide) || LP > 512) ? 1 : 2;
        for (;;) {
            if ((r = p8_alloc(c, *out, LP, pcb))) return r;
            CostParams q = p;
            q.C = nullptr;
            q.C8 = (*out)->p8;
            q.cbytes = pcb;
            q.L = LP;
            {
                TimeScope t(c, "k_cost");
                HIPCHK(c, launch_cost(q, c->stream));
            }
            bool fits = census_fits;  // (min(popcount, trunc) in integers: fits and is NaN-free by construction, nothing to read back)
            if (!fits) {
                HIPCHK(c, hipMemcpyAsync(c->h_words + 3, (*out)->bad8, 4, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                fits = c->h_words[3] == 0u;
            }
            if (fits) {
                (*out)->diff_fails = 0;
                (*out)->p8_state = 2;
                (*out)->f32_state = 0;
                (*out)->nan_state = 2;
                return MGM_OK;
            }
            HIPCHK(c, hipMemsetAsync((*out)->bad8, 0, 4, c->stream));
            if (pcb == 1 && c->h_words[3] == 1u && LP <= 512) {  // one byte per cost was too narrow, two will do
                (*out)->diff_wide = true;
                pcb = 2;
                continue;
            }
            break;
        }
        (*out)->diff_fails++;
    }
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
    } else if (p.C8 && !p.rlo && (costfn == 0 || costfn == 1) && (pre == 0 || pre == 2) && dev().lazy_f32 && (*out)->diff_fails < 2) {
        // Absolute / squared differences of (sobelx-filtered) 8-bit images are whole numbers: write the compact copy alone and
        // read the flag word back -- the read-back the aggregation would do anyway (c8_resolve).  A volume that does not fit
        // (float-valued images, a fractional truncDist) is filled again by the general kernel, fp32 volume and all, and so
        // are its refills.
        p.C = nullptr;
        (*out)->f32_state = 0;
        for (;;) {
            {
                TimeScope t(c, "k_cost");
                HIPCHK(c, launch_cost(p, c->stream));
            }
            HIPCHK(c, hipMemcpyAsync(c->h_words + 3, (*out)->bad8, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->h_words[3] == 0u) {
                (*out)->diff_fails = 0;
                (*out)->c8_state = 2;
                (*out)->nan_state = 2;
                return MGM_OK;
            }
            HIPCHK(c, hipMemsetAsync((*out)->bad8, 0, 4, c->stream));
            // one byte per cost was too narrow (a grey pair with a difference of 255), two would do (k_cost_diffx says so)
            if (p.cbytes == 1 && c->h_words[3] == 1u && p.L <= 512) {
                (*out)->diff_wide = true;
                if ((r = c8_alloc(c, *out, 2))) return r;
                p.C8 = (*out)->d8;
                p.cbytes = 2;
                continue;
            }
            break;
        }
        (*out)->diff_fails++;
        if ((r = cv_alloc_f32(c, *out))) return r;
        p.C = (*out)->d;
        p.C8 = nullptr;  // (no compact form: the fp32 volume alone)
        (*out)->c8_state = -1;
        (*out)->nan_state = 1;
        (*out)->f32_state = 1;
    } else {
        if ((r = cv_alloc_f32(c, *out))) return r;
        p.C = (*out)->d;
    }
    {
        TimeScope t(c, "k_cost");
        HIPCHK(c, launch_cost(p, c->stream));
    }
    // A ragged volume also gets its RANGE-PROPORTIONAL copy (mgm_pass_rel.hip): 64 cost bytes per pixel at the pixel's own
    // window -- what the aggregation then walks instead of the hull, if every window is at most 62 labels wide and every
    // cost a byte (the flag word is read back by the first aggregation).
    (*out)->rel_state = 0;
    if (p.rlo && p.C && rel_enabled()) {
        // the narrowest form the cost function can have: one byte for single-word census and grey-level absolute differences, two for
        // the other absolute / squared differences; the flag word (read back by the first aggregation, rel_resolve) widens it
        // (... and the fp32 cost itself -- four bytes -- for what has no integer form: NCC, Birchfield-Tomasi, census over several words,
        // differences of filtered images)
        const int rcb = ((costfn == 2 && census_words == 1) || (costfn == 0 && u->nch == 1 && pre == 0)) ? 1
                        : (((costfn == 0 || costfn == 1) && (pre == 0 || pre == 2)) ? 2 : 4);
        if ((r = rel_alloc(c, *out, 64, rcb))) return r;
        if ((*out)->relbuf) {
            HIPCHK(c, hipMemsetAsync((*out)->rel_flag(), 0, 4, c->stream));
            TimeScope t(c, "k_rel_gather");
            HIPCHK(c, launch_rel_gather(p.C, p.rlo, p.rhi, (long long)u->nx * u->ny, p.L, dmin, 64, rcb, (*out)->relbuf, (*out)->rel_records(), (*out)->rel_flag(),
                                        c->stream));
            (*out)->rel_state = 1;
        }
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
    for (int v = 0; v < n; v++)
        if (C[v]->unfilled) return fail(c, MGM_ERR_INVALID, "mgm_aggregate: the last mgm_costvolume_build of this volume failed; it holds no costs");
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
    // Ragged volumes whose range-proportional copies are usable take the kernels that walk the pixels' own windows
    // (mgm_pass_rel.hip): not when S is wanted (a dense volume), with TSGM = 2 (unweighted: other update functions), with
    // P2 = +INF (all-INF slabs: the operand-order-faithful kernel).
    // Measured (round 5, 1920x1080, windows of 49 labels in a hull of 256, K3 + search): FH 21.3 + 3.9 ms on the hull (the general
    // weighted kernels: the min-convolution runs over the RECEIVING pixel's range) -> 14.0 + 1.5 ms; free-form weights,
    // Hirschmueller: 11.4 + 3.9 -> 7.1 + 1.5; unit weights, Hirschmueller: 4.75 + 3.9 on the hull's queue kernels against
    // 7.1 + 1.5 + 0.45 (gather) -- a tie, and those keep the hull (MGM_HIP_TUNE=rel=2 sends them here too).
    // (round 5, later: unit-weight Hirschmueller volumes publish E on the producer side there, k_pass_rel PUBE: one volume 7.5 + 1.5 +
    // 0.45 ms against 4.6 + 3.9 -- still a tie --, two 10.5 + 2.9 + 0.9 against 9.2 + 7.7, four 16.0 + 5.9 + 1.7 against 15.2 + 14.8:
    // batches of them go)
    // (round 5, last: four scan lines per wave -- two volumes 9.3 + 3.0, four 10.5 + 5.9 --, and census volumes whose ONLY copy is the
    // range-proportional one: the hull would first have to be expanded and encoded again, 2.2 ms, so those go too: 8.1 + 1.5)
    bool only_rel = true;
    for (int v = 0; v < n; v++) only_rel = only_rel && C[v]->rel_only;
    // (round 6: with slope 1, two strips and the 5.6 ms launch the tie is gone -- one unit-weight Hirschmueller volume that HAS its hull,
    // 1920x1080, windows of 49 labels: 5.3 + 3.7 ms on the hull's queue kernels against 5.6 + 0.8 here, the gather paid either way:
    // every usable copy is walked; tune rel_tie=0 restores round 5's rule)
    const bool rel_pays = use_fh > 0 || (w8 && w8[0]) || n >= 2 || only_rel || tune_num("rel", 1) >= 2 || tune_num("rel_tie", 1) != 0;
    //  the hand-off of k_pass_rel keeps the launch's tag in the sign bit of every published word, so every L / E / minimum
    // must be >= +0: non-negative penalties (the dense path's own gate, run_passes `first_build`) and, for weighted launches,
    // positive finite weights (a negative or NaN weight makes N + P1 D - m negative or NaN): those take the dense kernels.
    // (round 6) TSGM = 2: with weights it is update_costW[_trunclinear] like every other TSGM; without -- weight images of ones
    // count as none, mgm_core.cc:420-423 -- Hirschmueller is update_cost2 (built: PUBE with halved terms), FH is
    // update_cost2_trunclinear with its boundary fix-up (not built here: the dense hull has it).
    bool rel_sign_ok = P1 >= 0.0f && P2 >= 0.0f;
    bool rel_weighted = w8 && w8[0];
    // (round 6, last: S wanted is no reason for the hull any more -- k_rel_S expands the corrected sum from the range-proportional Lr volumes)
    const bool rel_candidate = (!S || tune_num("rel_S", 1) != 0) && P2 < __builtin_huge_valf() && rel_pays && rel_sign_ok && rel_enabled();
    if (rel_candidate && rel_weighted) {
        bool odd = false, any = false;
        if ((r = weights_have_odd_values(c, w8, n, npix, &odd, &any))) return r;
        rel_sign_ok = !odd;
        if (MGM == 2 && !any) rel_weighted = false;  // planes of ones: the reference runs unweighted
    }
    const bool rel_fn_ok = MGM != 2 || rel_weighted || use_fh <= 0 || tune_num("rel_fh2", 1) != 0;  // (round 6, last: update_cost2_trunclinear is built too, k_pass_rel FH2)
    if (rel_candidate && rel_sign_ok && rel_fn_ok) {
        bool all = true;
        for (int v = 0; v < n && all; v++) {
            bool u = false;
            if ((r = rel_resolve(c, C[v], &u))) return r;
            all = u && C[v]->rel_slots == C[0]->rel_slots && C[v]->rel_cb == C[0]->rel_cb;  // (one format per launch)
        }
        // the one combination whose LDS rings do not fit a CU: update_cost2_trunclinear ([L][F][B] entries) on 128 slots of fp32 costs
        // (116 KB of rings + 64 KB of cost pieces): the dense hull keeps it
        if (all && use_fh > 0 && MGM == 2 && !rel_weighted && C[0]->rel_slots == 128 && C[0]->rel_cb == 4) all = false;
        if (all) {
            //  the range-proportional launch honours the workspace limit too, and a batch the device cannot hold is run
            // in halves instead of failing: NDIR x npix x slots floats per volume (+ ~16 % of hand-off slots: two lines per band on the anti-diagonal passes)
            int chunk = n;
            const double per_vol = 4.0 * ((double)npix * C[0]->rel_slots + (double)lr_pad_floats()) * NDIR * 1.16;
            if (c->ws_limit)
                while (chunk > 1 && per_vol * chunk > (double)c->ws_limit) chunk--;
            for (int v0 = 0; v0 < n;) {
                const int m = std::min(chunk, n - v0);
                r = run_rel(c, C + v0, rel_weighted ? w8 + v0 : nullptr, m, P1, P2, MGM, use_fh, NDIR, fix_overcount, ridx, out + v0, outcost + v0, S ? S + v0 : nullptr);
                if (r == MGM_ERR_NOMEM && m > 1) {
                    chunk = std::max(1, m / 2);
                    (void)hipGetLastError();
                    c->err.clear();
                    continue;
                }
                if (r) return r;
                v0 += m;
            }
            return MGM_OK;
        }
    }
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
        // mgm_ctx_set_placement_tries: the workspace has just been (re)allocated -- time this very launch on a few physical
        // placements and keep the fastest (the launch is idempotent: same inputs, same Lr volumes, whichever allocation)
        if (r == MGM_OK && c->place_tries >= 2 && c->lr.p && (c->lr.p != c->placed_ptr || c->lr.cap != c->placed_cap) && c->lr.cap >= (1ull << 28)) {
            auto timed = [&](float *ms) -> int {
                struct Ev {  // (destroyed on every exit path)
                    hipEvent_t e = nullptr;
                    ~Ev() { if (e) (void)hipEventDestroy(e); }
                } a, b;
                HIPCHK(c, hipEventCreate(&a.e));
                HIPCHK(c, hipEventCreate(&b.e));
                HIPCHK(c, hipEventRecord(a.e, c->stream));
                int rr = run_passes(c, C + v0, (w8 && w8[0]) ? w8 + v0 : nullptr, m, P1, P2, MGM, use_fh, 0, NDIR, true);
                if (rr == MGM_OK) {
                    HIPCHK(c, hipEventRecord(b.e, c->stream));
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                    HIPCHK(c, hipEventElapsedTime(ms, a.e, b.e));
                }
                return rr;
            };
            float best = 0;
            if ((r = timed(&best))) break;  // (the first run above was the warm-up)
            struct Held {  // the allocations that are only held so that the next try lands elsewhere: freed on every exit path
                std::vector<Buf> v;
                void release() { for (Buf &h : v) if (h.p) (void)hipFree(h.p); v.clear(); }
                ~Held() { release(); }
                void push_back(const Buf &b) { v.push_back(b); }
            } held;
            for (int t = 1; t < c->place_tries && r == MGM_OK; t++) {
                size_t fr = 0, tot = 0;
                if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < c->lr.cap + (1ull << 30)) break;  // no room for a second workspace
                Buf old = c->lr;
                c->lr = Buf{};
                float ms = 0;
                int rr = run_passes(c, C + v0, (w8 && w8[0]) ? w8 + v0 : nullptr, m, P1, P2, MGM, use_fh, 0, NDIR, true);  // allocates; warm-up
                if (rr == MGM_OK) rr = timed(&ms);
                if (rr != MGM_OK) {  // (out of memory after all: back to what we had)
                    if (c->lr.p) (void)hipFree(c->lr.p);
                    c->lr = old;
                    (void)hipGetLastError();
                    c->err.clear();
                    r = run_passes(c, C + v0, (w8 && w8[0]) ? w8 + v0 : nullptr, m, P1, P2, MGM, use_fh, 0, NDIR, true);
                    break;
                }
                if (tune_num("show_plan", 0)) fprintf(stderr, "[mgm place] try %d: %.3f ms (best so far %.3f)\n", t, ms, best);
                if (ms < best * 0.985f) {  // the new placement is the faster one: the old allocation is only held (so that the next try lands elsewhere)
                    held.push_back(old);
                    best = ms;
                } else {  // keep the old one: hold the new one instead, and run the launch on the old workspace again (its Lr volumes are what k_wta reads)
                    held.push_back(c->lr);
                    c->lr = old;
                    r = run_passes(c, C + v0, (w8 && w8[0]) ? w8 + v0 : nullptr, m, P1, P2, MGM, use_fh, 0, NDIR, true);
                }
            }
            HIPCHK(c, hipStreamSynchronize(c->stream));
            held.release();
            c->placed_ptr = c->lr.p;
            c->placed_cap = c->lr.cap;
        }
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
extern "C++" int pipe_flush(mgm_ctx *c)
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
    int r = aggregate_batch_now(c, (int)Cs.size(), Cs.data(), Ws.empty() ? nullptr : Ws.data(), a.P1, a.P2, a.NDIR, a.MGM, a.use_fh,
                                a.fix_overcount, a.has_refine ? a.refine.c_str() : nullptr, Os.data(), Ks.data(), nullptr);
    if (r == MGM_ERR_UNSUPPORTED && q.size() > 1) {
        // Calls that each are fine may not go into ONE launch (e.g. weight planes that are all ones beside real ones: "all
        // weighted or all unweighted" is decided on the values): run them as the caller issued them.
        r = MGM_OK;
        for (const auto &b : q) {
            const int rb = aggregate_batch_now(c, (int)b.C.size(), b.C.data(), b.w8.empty() ? nullptr : b.w8.data(), b.P1, b.P2, b.NDIR,
                                               b.MGM, b.use_fh, b.fix_overcount, b.has_refine ? b.refine.c_str() : nullptr,
                                               const_cast<mgm_img **>(b.out.data()), const_cast<mgm_img **>(b.outcost.data()), nullptr);
            if (rb != MGM_OK && r == MGM_OK) r = rb;
        }
    }
    return r;
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
    if (C->unfilled) return fail(c, MGM_ERR_INVALID, "mgm_aggregate_passes: the last mgm_costvolume_build of this volume failed; it holds no costs");
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
    if (C->unfilled) return fail(c, MGM_ERR_INVALID, "mgm_aggregate_passes: the last mgm_costvolume_build of this volume failed; it holds no costs");
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
    if (c && dense && c->rel_last_batch > 0 && c->last_ndir == 0 && c->rel_last_cvs[0] && c->lr_rel.p) {
        // the last aggregation ran on the range-proportional copy (k_pass_rel): volume 0's pass, expanded on the host to the dense
        // hull -- label o <-> slot o + dmin - base(p) of the pixel's 64, +INF where the pixel has no such label (test aid only)
        if (pass < 0 || pass >= c->rel_last_ndir) return fail(c, MGM_ERR_INVALID, "mgm_debug_download_lr: no such pass");
        const mgm_cv *C = c->rel_last_cvs[0];
        const size_t npix = (size_t)C->nx * C->ny;
        const int L = C->dmax - C->dmin + 1;
        HIPCHK(c, hipSetDevice(c->device));
        const size_t slots = (size_t)C->rel_slots;
        std::vector<float> slabs(npix * slots);
        std::vector<int> rec(npix * 4);
        HIPCHK(c, hipMemcpyAsync(slabs.data(), (const float *)c->lr_rel.p + (size_t)pass * c->rel_last_stride, sizeof(float) * npix * slots,
                                 hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(rec.data(), C->rel_records(), sizeof(int) * npix * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (size_t p = 0; p < npix; p++) {
            const int b = rec[p * 4], lo = rec[p * 4 + 1], hi = rec[p * 4 + 2];
            for (int o = 0; o < L; o++) {
                const int d = C->dmin + o;
                dense[p * L + o] = (d >= lo && d <= hi) ? slabs[p * slots + (d - b)] : __builtin_huge_valf();
            }
        }
        return MGM_OK;
    }
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
// Diagnostic: the rate (GB/s) at which the store pattern of the pass kernels lands on the context's Lr workspace as it is
// placed now -- `nstreams` volumes at the last aggregation's stride written side by side (tools/bimodal_probe.py).
int mgm_debug_probe_workspace(mgm_ctx *c, int nstreams, float *gbps)
{
    if (int jr = pipe_join(c)) return jr;
    if (!c || !gbps || nstreams < 1 || nstreams > 64) return fail(c, MGM_ERR_INVALID, "mgm_debug_probe_workspace: bad arguments");
    if (!c->lr.p || c->last_stride <= 0) return fail(c, MGM_ERR_INVALID, "mgm_debug_probe_workspace: no aggregation has run on this context");
    HIPCHK(c, hipSetDevice(c->device));
    const long long stride = c->last_stride;
    const long long have = (long long)(c->lr.cap / sizeof(float));
    if ((long long)nstreams * stride > have) nstreams = (int)(have / stride);
    if (nstreams < 1) return fail(c, MGM_ERR_INVALID, "mgm_debug_probe_workspace: workspace smaller than one volume");
    const long long per = std::min<long long>(stride, (1ll << 28)) / 4 * 4;  // at most 1 GiB per stream
    hipEvent_t a, b;
    HIPCHK(c, hipEventCreate(&a));
    HIPCHK(c, hipEventCreate(&b));
    HIPCHK(c, launch_probe_streams((float *)c->lr.p, stride, nstreams, per, c->stream));  // (warm)
    HIPCHK(c, hipEventRecord(a, c->stream));
    HIPCHK(c, launch_probe_streams((float *)c->lr.p, stride, nstreams, per, c->stream));
    HIPCHK(c, hipEventRecord(b, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *gbps = (float)((double)per * 4.0 * nstreams / (ms * 1e-3) / 1e9);
    return MGM_OK;
}

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
    for (int v = 0; v < c->rel_last_batch; v++)  // the context's last aggregation of this volume ran on its range-proportional copy
        if (c->rel_last_cvs[v] == C && c->rel_last_gens[v] == C->gen && c->rel_last_ndir == NDIR) {
            HIPCHK(c, hipSetDevice(c->device));
            return run_wta_rel(c, C, v, NDIR, fix_overcount, refinement_index(refine), dminI->d, dmaxI->d, out->d, outcost->d);
        }
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
