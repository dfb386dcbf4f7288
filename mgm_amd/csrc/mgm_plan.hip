// mgm_plan.hip -- the launch plan: which pass kernel a batch of volumes takes (build, label width, compact costs, queues,
// strips, occupancy), its workspace and hand-off regions, the task table; and the winner search behind it.  See mgm_host.h.
#include "mgm_host.h"

#include <functional>

// ---- aggregation ----------------------------------------------------------------
// K3 for the passes [first, first+count) of the reference's table; pass p's Lr volume goes to
// workspace slot p - first.  Shared by mgm_aggregate_dev and the direction-sharded multi-GPU path.
// K3 over `nb` cost volumes of identical geometry in one launch (see PassVolume).  Volume v's Lr volumes
// end up at lr + v*count*lr_stride.
// smallest label count the second K3 build takes that holds L labels (0: none)
// How many per-XCD work queues a launch can use on this device: the XCC ids 0 .. n-1 a 2048-workgroup launch saw
// (k_xcc_census), if they are exactly that -- 8 on an MI355X in SPX mode, fewer on a partitioned device, 0 = no queues.
static int xcc_queues(int mask)
{
    int n = 0;
    while (n < 8 && ((mask >> n) & 1)) n++;
    return (n >= 2 && mask == (1 << n) - 1) ? n : 0;
}

int padded_labels(int L)
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
    if ((r = reserve(c, c->exact_mins, sizeof(float) * (size_t)npix * count))) return r;
    const bool big_fh = fh && (size_t)L * 16 > 128 * 1024;  // the four convolution arrays of a pixel do not fit the LDS
    if (big_fh && (r = reserve(c, c->exact_scratch, sizeof(float) * (size_t)count * std::max(nx, ny) * 4 * (size_t)L))) return r;
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
    p.fhscratch = big_fh ? (float *)c->exact_scratch.p : nullptr;
    for (int v = 0; v < nb; v++) {
        if ((r = ensure_f32(c, Cs[v]))) return r;
        p.C = Cs[v]->d;
        p.dmin = Cs[v]->dmin;
        p.w8 = weighted ? w8s[v]->d : nullptr;
        p.rlo = Cs[v]->rlo;
        p.rhi = Cs[v]->rhi;
        // (round 6) all passes of the volume in ONE sweep of launches, one per diagonal (they are independent: blockIdx.y = pass);
        // every pixel writes its whole slab when its diagonal comes, so Lr = CC (495-498) needs no copy
        p.npass = count;
        for (int q = first; q < first + count; q++) {
            const RefPass &rp = kPasses[q];
            ExactPass &g = p.pass[q - first];
            for (int k = 0; k < 4; k++) {
                g.d[k][0] = rp.d[k][0];
                g.d[k][1] = rp.d[k][1];
                g.wplane[k] = kPassToChannel[k][q];
            }
            g.inc_x = rp.inc_x;
            g.inc_y = rp.inc_y;
            g.row_major = rp.row_major;
            g.Lr = (float *)c->lr.p + ((size_t)v * nslots + slot0 + (q - first)) * lr_stride;
        }
        TimeScope t(c, "k_pass_exact");
        HIPCHK(c, launch_pass_exact(p, c->stream));
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
    c->rel_last_batch = 0;  // (an earlier range-proportional aggregation of the same volume is no longer the context's last)
    return MGM_OK;
}

// Weighted launches of the range-proportional kernels: does any weight image hold a value that is not positive and finite
// (<= 0, NaN, INF)?  One small scan + read-back per call, the same kernel run_passes uses to recognise two-valued weights.
int weights_have_odd_values(mgm_ctx *c, const mgm_img *const *w8s, int nb, long long npix, bool *odd, bool *any)
{
    *odd = false;
    if (any) *any = false;
    int r;
    if ((r = reserve(c, c->wvals, sizeof(unsigned) * 4 * kMaxBatch))) return r;
    unsigned init[4 * kMaxBatch], got[4 * kMaxBatch];
    for (int v = 0; v < nb; v++) init[4 * v] = 0u, init[4 * v + 1] = 0xffffffffu, init[4 * v + 2] = 0u, init[4 * v + 3] = 0u;
    HIPCHK(c, hipMemcpyAsync(c->wvals.p, init, sizeof(unsigned) * 4 * nb, hipMemcpyHostToDevice, c->stream));
    for (int v = 0; v < nb; v++)
        if (w8s[v]) HIPCHK(c, launch_weight_values(w8s[v]->d, npix * 8, (unsigned *)c->wvals.p + 4 * v, c->stream));
    HIPCHK(c, hipMemcpyAsync(got, c->wvals.p, sizeof(unsigned) * 4 * nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int v = 0; v < nb; v++) {
        *odd = *odd || (w8s[v] && got[4 * v + 3] != 0);
        if (any) *any = *any || (w8s[v] && got[4 * v] != 0);  // a value != 1.0: the run is a weighted one (mgm_core.cc:420-423)
    }
    return MGM_OK;
}

// The launch as a list-scheduling problem, simulated on the host (run_passes, order 2).  A chain = the bands of one pass (or
// one strip of it) of one volume, band b READY once band b-1 (of both strips) started `skew` steps earlier; `nqueues` queues of
// `slots` band slots each, band b of chain k belonging to queue (b / QK + k.chain) % nqueues; a free slot starts, among the
// heads of the chains whose next band belongs to its queue, the ready one with the longest remaining chain (none ready: the
// one that will be first).  Returns the makespan in steps; `order` = the items by their start in that schedule -- per queue,
// the order in which the queue hands them out.
struct SimChain {
    int x, st, nb, sib, chain;
    double skew, len;
    // bands that differ (the anti-diagonal passes of k_pass_rel): band b is ready bskew[b] steps after band b-1 started, runs blen[b]
    // steps, and brem[b] steps of the chain remain behind its start; empty: skew / len for every band
    std::vector<double> bskew, blen, brem;
    double skew_of(int b) const { return bskew.empty() ? skew : bskew[b]; }
    double len_of(int b) const { return blen.empty() ? len : blen[b]; }
    double rem_of(int b) const { return brem.empty() ? (double)(nb - 1 - b) * skew + len : brem[b]; }
    double gap_of(int b) const { return bskew.empty() ? skew : std::min(bskew[b], 4.0); }  // (a band cannot end before its predecessor + this)
};
static double simulate_schedule(const std::vector<SimChain> &chains, int nqueues, int slots, int QK, std::vector<int2> &order)
{
    const int n = (int)chains.size();
    std::vector<std::vector<double>> start(n), end(n);
    std::vector<int> next(n, 0);
    size_t total = 0;
    for (int i = 0; i < n; i++) {
        start[i].assign(chains[i].nb, 0.0);
        end[i].assign(chains[i].nb, 0.0);
        total += (size_t)chains[i].nb;
    }
    auto queue_of = [&](int i, int b) { return nqueues <= 1 ? 0 : (b / QK + chains[i].chain) % nqueues; };
    typedef std::pair<double, int> Slot;  // (free at, queue)
    std::vector<Slot> heap;
    for (int q = 0; q < std::max(1, nqueues); q++)
        for (int k = 0; k < slots; k++) heap.push_back(Slot(0.0, q));
    std::make_heap(heap.begin(), heap.end(), std::greater<Slot>());
    order.clear();
    const double INF = 1e300;
    double makespan = 0.0;
    size_t guard = 0;
    while (order.size() < total && !heap.empty() && guard++ < 64 * total + 4096) {
        std::pop_heap(heap.begin(), heap.end(), std::greater<Slot>());
        const Slot sl = heap.back();
        heap.pop_back();
        const double t = sl.first;
        int best = -1;
        double best_rem = -1, best_ready = INF;
        bool best_is_ready = false, later = false;
        for (int i = 0; i < n; i++) {
            const SimChain &k = chains[i];
            if (next[i] >= k.nb) continue;
            const int b = next[i];
            if (queue_of(i, b) != sl.second) {
                for (int bb = b + 1; bb < k.nb && !later; bb += std::max(1, QK)) later = queue_of(i, bb) == sl.second;
                continue;
            }
            double ready = 0.0;
            if (b > 0) {
                ready = start[i][b - 1] + k.skew_of(b);
                if (k.sib >= 0) ready = next[k.sib] > b - 1 ? std::max(ready, start[k.sib][b - 1] + k.skew) : INF;
            }
            const double rem = k.rem_of(b);
            const bool is_ready = ready <= t;
            const bool better = best < 0 || (is_ready != best_is_ready ? is_ready : (is_ready ? rem > best_rem : (ready != best_ready ? ready < best_ready : rem > best_rem)));
            if (better) best = i, best_rem = rem, best_ready = ready, best_is_ready = is_ready;
        }
        if (best < 0 || best_ready >= INF) {
            // nothing of this queue can start yet (its next bands follow bands of other queues, or the other strip): look again
            // when the next slot frees; a queue with nothing left retires its slots
            if (best < 0 && !later) continue;
            const double again = heap.empty() ? t + 1.0 : std::max(t, heap.front().first) + 1.0;
            heap.push_back(Slot(again, sl.second));
            std::push_heap(heap.begin(), heap.end(), std::greater<Slot>());
            continue;
        }
        const SimChain &k = chains[best];
        const int b = next[best]++;
        const double st_eff = std::max(t, best_ready);
        double en = st_eff + k.len_of(b);
        if (b > 0) en = std::max(en, end[best][b - 1] + k.gap_of(b));  // (it cannot overtake its predecessor)
        start[best][b] = st_eff;
        end[best][b] = en;
        makespan = std::max(makespan, en);
        heap.push_back(Slot(en, sl.second));
        std::push_heap(heap.begin(), heap.end(), std::greater<Slot>());
        order.push_back(make_int2(k.x, b + (k.st << 16)));
    }
    if (order.size() < total) {  // (cannot happen; never lose an item to the model)
        for (int i = 0; i < n; i++)
            for (int b = next[i]; b < chains[i].nb; b++) order.push_back(make_int2(chains[i].x, b + (chains[i].st << 16)));
        makespan = INF;
    }
    return makespan;
}

// slot0 / nslots: pass p's Lr volume goes to workspace slot slot0 + (p - first) of nslots (a caller that launches the
// passes of one volume one at a time keeps them all: mgm_aggregate_passes_at_dev); layout_ndir: the hand-off region is
// laid out for the passes [0, layout_ndir) whichever of them this launch runs, so that such a caller's launches share it.
int run_passes(mgm_ctx *c, const mgm_cv *const *Cs, const mgm_img *const *w8s, int nb, float P1, float P2, int MGM,
               int use_fh, int first, int count, bool allow_pad, int slot0, int nslots, int layout_ndir)
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
    bool weighted = false, w2cand = false, wodd = false;  // wodd: a weight that is not positive and finite (<= 0, NaN, INF)
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
            // Planes of ones beside real weights: with TSGM != 2 the reference calls the SAME update function either way
            // (update_costW[_trunclinear] with DeltaI = 1.0, mgm_core.cc:563-575), so the launch simply runs weighted; with
            // TSGM = 2 the unweighted volume is update_cost2's, a different function: not in one launch.
            if (v && wv != weighted && (MGM == 2 || !w8s[v]))
                return fail(c, MGM_ERR_UNSUPPORTED, "batched volumes must be all weighted or all unweighted");
            weighted = weighted || wv;
            wodd = wodd || (wv && got[4 * v + 3] != 0);
            w2cand = w2cand && (!wv || (got[4 * v + 3] == 0 && got[4 * v + 1] == got[4 * v + 2]));
            if (wv) memcpy(&w2a[v], &got[4 * v + 1], 4);
            else w2a[v] = 1.0f;  // (planes of ones in a weighted launch: no selector bit is set, the other value is never used)
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
    //   * (round 6, found by the ragged parity tests of tests/) FH potentials on a ragged volume with a NEGATIVE slope -- P1 < 0, or a
    //     weight <= 0 / NaN scaling it: the fast kernels mask the neighbour's slab to the receiving pixel's range and convolve
    //     over the whole hull, which equals the reference's convolution over the range (mgm_core.cc:242-271) only while the
    //     ramp the forward pass leaves ABOVE the range cannot flow back into it (M[rh] + 2 P1 >= M[rh]).
    bool exact = (ragged && !(P2 < __builtin_huge_valf())) || Lreal > kMaxLPL * 64 || (ragged && use_fh > 0 && (!(P1 >= 0.0f) || wodd));
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
    // padded launch whose volumes all carry the padded compact copy K2 wrote (mgm_cv::p8): nothing to pad or encode
    bool own_padded = padded && dev().c8;
    for (int v = 0; v < nb && own_padded; v++)
        own_padded = Cs[v]->p8_state == 2 && Cs[v]->p8_L == L && Cs[v]->p8_cb == Cs[0]->p8_cb;
    if (own_padded) {
        cb = Cs[0]->p8_cb;
    } else if (padded) {
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
    // everything else reads the fp32 volume.  Since round 4 K2 writes AD / SD volumes compact-ONLY (f32_state 0), so a
    // weighted launch, FH with TSGM = 2 or a shallow-ring launch on such a volume pays an expansion (ensure_f32: k_expand
    // plus a W*H*L fp32 allocation) the first time -- correct, but four times the cost bytes; bench.py's cost_bytes prices
    // weighted launches on fp32 costs for that reason.
    if (use_c8 && cb == 2 && (weighted || (fh && MGM == 2) || pass_lpl(L) > 8 || dev().deep == 0)) {
        own_padded = false;
        if (padded)
            for (int v = 0; v < nb; v++) {
                if ((r = ensure_f32(c, Cs[v]))) return r;
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
    // (FH on a ragged volume WITH two-valued weights as well: the producer-side transforms of W2 convolve over the SENDING pixel's
    // slab -- found by the long random campaign of tests/test_gpu_rel.py in round 5, where the range-proportional kernels and the
    // reference agreed and this path did not)
    bool w2 = w2cand && dev().w2 && R2 && use_c8 && cb == 1 && lpl <= 4 && !ones8 && !(fh && ragged) && !pass2_devtools() && dev().xcdq != 0 && dev().deep != 0;
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
    w2 = w2 && xcc_queues(c->xcc_mask) > 0;
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
        xcdq = xcc_queues(c->xcc_mask) > 0;
    }
    const int nq = xcdq ? xcc_queues(c->xcc_mask) : 8;  // queues of the launch (the XCDs of the device)
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
    bool one_queue = xcdq && (dev().xcdq == 2 || tune_num("one_queue", -1) > 0);  // (order 2: decided by the simulated schedules, below)
    int QK = ((ngroups * count) % nq == 0 && maxLL <= 3000) ? 0 : 2;
    if (dev().xcdq_k >= 0) QK = dev().xcdq_k;
    if (QK <= 0) QK = 1 << 20;
    if (tune_num("show_plan", 0))  // development aid: what the launch heuristics decided
        fprintf(stderr, "[mgm plan] %dx%dx%d passes %d..%d x %d volumes: load/chain %.2f, %d wg/cu, deep %d, subv %d, strips %d, xcd queues %d (block %d; xcc ids seen 0x%x)\n", nx, ny, L,
                first, PEND - 1, nb, load_ratio, p.wg_per_cu, p.deep, subv, any_strips ? 1 : 0, xcdq ? 1 : 0, QK >= (1 << 20) ? 0 : QK, (unsigned)c->xcc_mask);
    // task table: ticket -> (pass, band [, strip]); item (p, b, .) always follows the items (p, b-1, .)
    const int tk_key = (((((PEND * 16 + first) * kMaxBatch + nb - 1) * 8 + subv) * 2 + (any_strips ? 1 : 0)) * 2 + (xcdq ? 1 : 0)) * 4 +
                       (p.wg_per_cu >= 2 ? 2 : 0);  // (the simulated schedule depends on the band slots)
    if (c->tk_nx != nx || c->tk_ny != ny || c->tk_ndir != tk_key || c->tk_r != R)
        for (auto &t : c->ttabs)
            if (t.nx == nx && t.ny == ny && t.key == tk_key && t.R == R) {  // a shape seen before: its table is still on the device
                c->tasks = t.buf;
                c->ntasks = t.ntasks;
                c->tk_one_queue = t.one_queue;
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
        // Round 5 (tools/timeline.py on the queue kernels: 14-24 % of a single launch's CU-time was TAIL -- compute units
        // with nothing left while the 2 ms items of the row passes, dealt last like everybody's last bands, ran out): the
        // items are dealt by their LATEST START TIME instead, i.e. longest remaining chain first -- item (p, b) is followed,
        // band after band, by (nbands - 1 - b) hand-offs of slope * R + lag steps and its own walk; what has the longest
        // way to go to the end of its pass starts first, and the launch ends on the SHORT items (the strips of the
        // column passes).  Within a pass the order is still by band (the remaining chain shrinks with b).
        const int order = (int)tune_num("order", 2);  // (0: the relative-progress order of rounds 1-4, 1: longest remaining chain first -- for A/B runs)
        if (order == 0) {
            std::stable_sort(tasks.begin(), tasks.end(), [&](const int2 &a, const int2 &b) {
                const long long ka = (long long)(a.y & 0xffff) * p.g[b.x % kMaxDirs].nbands, kb = (long long)(b.y & 0xffff) * p.g[a.x % kMaxDirs].nbands;
                return ka != kb ? ka < kb : a.x < b.x;
            });
        } else if (order >= 2) {
            // LIST SCHEDULING, simulated (simulate_schedule, above): the tickets come out in the order in which a machine of
            // band slots that always starts, among the items whose predecessor band is far enough ahead (READY), the one with
            // the longest remaining chain would start them.  The launch then follows that schedule by itself -- every free
            // workgroup takes the next ticket of its queue -- as far as its step times match the model's (one step = one time
            // unit for every pass), and an item taken early merely waits, as it always could.  What the plain
            // longest-remaining-chain order gets wrong is the START of the launch: it hands the first 256 tickets to some sixty
            // consecutive bands of the two longest chains, of which band k cannot move before k * (slope * R) steps have
            // passed (timeline, round 5: 43 % of the CUs waiting through the first millisecond).
            // The same simulation DECIDES between the per-XCD queues and one queue for all XCDs (write-through hand-offs
            // everywhere): a pass pinned to an XCD runs in whole rounds of that XCD's 32 CUs -- 72 row bands of 2.3 ms are three
            // rounds, the last one a quarter full (1920x1080x256 FH x 1, K3: pinned 6.08 ms, one queue 5.76; two volumes, two
            // passes per XCD: 10.25 against 11.04) -- so the plan takes the single queue where its simulated makespan is
            // shorter by more than what the crossing hand-offs cost (4 %).
            const double lagS = tags ? 5.0 : 12.0;
            std::vector<SimChain> ch;
            for (int v = 0; v < ngroups; v++)
                for (int q = first; q < PEND; q++)
                    for (int st = 0; st < p.g[q].nstrips; st++) {
                        const PassGeom &g = p.g[q];
                        SimChain k;
                        k.x = v * kMaxDirs + q, k.st = st, k.nb = g.nbands, k.chain = v * count + (q - first);
                        k.sib = g.nstrips == 2 ? (int)ch.size() + (st == 0 ? 1 : -1) : -1;
                        k.skew = (double)g.slope * R + lagS;
                        k.len = (g.nstrips == 2 ? (st == 0 ? g.split : g.LL - g.split) + R - 1 : g.LL) + (double)g.slope * (R - 1) + 3.0;
                        ch.push_back(k);
                    }
            const int slots = std::max(1, c->num_cu * std::max(1, p.wg_per_cu));
            std::vector<int2> ord_one, ord_q;
            const double t_one = simulate_schedule(ch, 1, slots, 1 << 20, ord_one);
            double t_q = t_one;
            if (xcdq) t_q = simulate_schedule(ch, nq, std::max(1, slots / nq), QK, ord_q);
            const long long forced = tune_num("one_queue", -1);
            // (launches that run two bands per CU keep their queues: with the doubled step the pinned dealing measured 5 % FASTER
            // for two and four 256-label FH volumes although the model says otherwise -- 9.67 against 10.18 ms, 17.7 against 18.3)
            one_queue = xcdq && (dev().xcdq == 2 || forced > 0 || (forced < 0 && p.wg_per_cu < 2 && t_one * 1.04 < t_q));
            const bool sim_ok = ((xcdq && !one_queue) ? t_q : t_one) < 1e299 && ((xcdq && !one_queue) ? ord_q : ord_one).size() == tasks.size();
            if (sim_ok) {
                tasks = (xcdq && !one_queue) ? ord_q : ord_one;
            } else {
                // (never seen: the simulation gave up.  Its leftovers are NOT a valid order -- a strip's band would precede the
                // other strip's band before it -- so the launch takes the sorted order: longest remaining chain first, which is one)
                const double lag1 = tags ? 3.0 : 10.0;
                auto rem1 = [&](const int2 &t) {
                    const PassGeom &g = p.g[t.x % kMaxDirs];
                    const int b = t.y & 0xffff, st = (t.y >> 16) & 0xff;
                    const double walk = (g.nstrips == 2 ? (st == 0 ? g.split : g.LL - g.split) + R - 1 : g.LL) + (double)g.slope * R;
                    return (double)(g.nbands - 1 - b) * (g.slope * R + lag1) + walk;
                };
                std::stable_sort(tasks.begin(), tasks.end(), [&](const int2 &a, const int2 &b) {
                    const double ra = rem1(a), rb = rem1(b);
                    return ra != rb ? ra > rb : a.x < b.x;
                });
            }
            if (tune_num("show_plan", 0))
                fprintf(stderr, "[mgm plan] simulated makespan (steps): one queue %.0f, %d queues (block %d) %.0f -> %s\n", t_one, nq, QK >= (1 << 20) ? 0 : QK, t_q,
                        one_queue ? "one queue" : (xcdq ? "per-XCD queues" : "ticket counter"));
        } else {
            const double lag = tags ? 3.0 : 10.0;
            auto remaining = [&](const int2 &t) {
                const PassGeom &g = p.g[t.x % kMaxDirs];
                const int b = t.y & 0xffff, st = (t.y >> 16) & 0xff;
                const double walk = (g.nstrips == 2 ? (st == 0 ? g.split : g.LL - g.split) + R - 1 : g.LL) + (double)g.slope * R;
                return (double)(g.nbands - 1 - b) * (g.slope * R + lag) + walk;
            };
            std::stable_sort(tasks.begin(), tasks.end(), [&](const int2 &a, const int2 &b) {
                const double ra = remaining(a), rb = remaining(b);
                return ra != rb ? ra > rb : a.x < b.x;
            });
        }
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
                if (one_queue) qs[0].push_back(t);  // (one queue for all XCDs, write-through hand-offs)
                else qs[(b / QK + chain) % nq].push_back(make_int2(t.x, t.y | (same ? 1 << 24 : 0)));
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
        c->ttabs.push_back(mgm_ctx::TaskTab{nx, ny, tk_key, R, (int)tasks.size(), fresh, one_queue});
        c->tasks = fresh;
        c->ntasks = (int)tasks.size();
        c->tk_one_queue = one_queue;
        c->tk_nx = nx;
        c->tk_ny = ny;
        c->tk_ndir = tk_key;
        c->tk_r = R;
    }

    one_queue = xcdq && c->tk_one_queue;  // (what the table in use was dealt for)
    for (int v = 0; v < nb; v++) {
        if (!use_c8 && (r = ensure_f32(c, Cs[v]))) return r;
        p.vol[v].C = padded ? (const float *)c->padf[v].p : Cs[v]->d;
        p.vol[v].C8 = use_c8 ? (padded ? (own_padded ? Cs[v]->p8 : (const uint8_t *)c->pad8[v].p) : Cs[v]->d8) : nullptr;
        p.vol[v].Lr = (float *)c->lr.p + ((size_t)v * nslots + slot0) * lr_stride;
        p.vol[v].w8 = ones8 ? ones8 : (weighted ? w8s[v]->d : nullptr);
        p.vol[v].rlo = (fh && ragged) ? Cs[v]->rlo : nullptr;
        p.vol[v].rhi = (fh && ragged) ? Cs[v]->rhi : nullptr;
        if (w2) {
            if ((r = reserve(c, c->wsel[v], sizeof(unsigned) * 2 * (size_t)npix))) return r;
            HIPCHK(c, launch_wsel(w8s[v]->d, npix, (unsigned *)c->wsel[v].p, c->stream));
            {   // ... + which of a pixel's two transforms its readers pick (k_wneed): what the kernel gets is the combined word
                int dd[8][4][2], pl[8][4];
                for (int q = 0; q < 8; q++)
                    for (int k = 0; k < 4; k++) {
                        dd[q][k][0] = kPasses[q].d[k][0];
                        dd[q][k][1] = kPasses[q].d[k][1];
                        pl[q][k] = kPassToChannel[k][q];
                    }
                HIPCHK(c, launch_wneed((const unsigned *)c->wsel[v].p, nx, ny, MGM, dd, pl, (unsigned *)c->wsel[v].p + npix, c->stream));
            }
            p.vol[v].wsel = (const unsigned *)c->wsel[v].p + npix;
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
    p.xcdq = xcdq ? (one_queue ? 2 : 1) : 0;
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
    p.tl_addr = 0;
    p.tl_on = 0;
    p.xflags = 0;
    p.xflags = dev().xflags;
    // MGM_HIP_TIMELINE=<file> with a -DMGM_P2_TIMELINE=1 build of the pass kernels: one line per work item of every queue
    // launch (tools/timeline.py reads them) -- where the compute units' time goes in a single launch
    const char *tl_file = (xcdq && R2 && pass2_timeline()) ? getenv("MGM_HIP_TIMELINE") : nullptr;
    if (tl_file && *tl_file) {
        if ((r = reserve(c, c->dbg, sizeof(unsigned long long) * 8 * (size_t)c->ntasks))) return r;
        HIPCHK(c, hipMemsetAsync(c->dbg.p, 0, sizeof(unsigned long long) * 8 * (size_t)c->ntasks, c->stream));
        p.tl_addr = (unsigned long long)(uintptr_t)c->dbg.p;
        p.tl_on = 1;
    }
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
        if (tune_num("check_tags", 0) != 0) {
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
    if (p.tl_on) {
        std::vector<unsigned long long> d((size_t)c->ntasks * 8);
        std::vector<int2> tk((size_t)c->ntasks + 8);
        HIPCHK(c, hipMemcpyAsync(d.data(), c->dbg.p, d.size() * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(tk.data(), (const int2 *)c->tasks.p, tk.size() * sizeof(int2), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (FILE *f = fopen(tl_file, "a")) {
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int i = 0; i < c->ntasks; i++) {
                if (d[(size_t)i * 8]) t0 = std::min(t0, d[(size_t)i * 8]);
                t1 = std::max(t1, d[(size_t)i * 8 + 1]);
            }
            const double tick = 1e-2;  // wall_clock64: 100 MHz -> 0.01 us
            fprintf(f, "launch %d %d %d %d %d %.1f volumes %d fh %d mgm %d strips %d\n", nx, ny, L, c->ntasks, p.wg_per_cu, (double)(t1 - t0) * tick, nb,
                    fh ? 1 : 0, MGM, any_strips ? 1 : 0);
            for (int i = 0; i < c->ntasks; i++) {
                int queue = 0;
                for (int q = 0; q < 8; q++)
                    if (i >= tk[q].x && i < tk[q].x + tk[q].y) queue = q;
                const int2 t = tk[(size_t)i + 8];
                const unsigned long long *w = &d[(size_t)i * 8];
                fprintf(f, "item %d %d %d %d %d %.2f %.2f %.2f %llu %llu %llu %llu %llu\n", i, t.x, t.y & 0xffff, (t.y >> 16) & 0xff, queue,
                        (double)(w[0] - t0) * tick, (double)(w[1] - t0) * tick, (double)w[2] * tick, w[3], w[4] & 0xffffffffull, w[4] >> 32, w[5], w[6]);
            }
            fclose(f);
        }
    }
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
    c->rel_last_batch = 0;  // (the context's last aggregation is this dense one)
    c->last_nvol = nvol;
    c->last_stride = lr_stride;
    c->last_ndir = nslots;
    c->last_batch = nb;
    c->last_L = Lreal;
    c->last_Lk = L;
    c->last_pad_c8 = padded && use_c8;
    c->last_pad_cb = cb;
    for (int v = 0; v < nb; v++) c->last_pad_ptr[v] = (padded && use_c8) ? (own_padded ? Cs[v]->p8 : (const uint8_t *)c->pad8[v].p) : nullptr;
    for (int v = 0; v < kMaxBatch; v++) {
        c->last_cvs[v] = v < nb ? Cs[v] : nullptr;
        c->last_gens[v] = v < nb ? Cs[v]->gen : 0;
    }

    return MGM_OK;
}

// K4-K6 over `npix` pixels starting at pixel `pix0` of C, reading pass p's Lr from lr + p*lr_stride.
// `slot` >= 0: the volume was slot `slot` of the context's last aggregation; if that launch ran with a padded label
// count, its padded cost copies and label stride are used (see run_passes).  slot < 0: plain [pix][L] layout.
int run_wta(mgm_ctx *c, const mgm_cv *C, long long pix0, long long npix, const float *lr, long long lr_stride, int NDIR,
            int fix_overcount, int ridx, float *out, float *outcost, float *Sout, const float *wlo, const float *whi, int slot)
{
    const int Lreal = C->dmax - C->dmin + 1;
    const bool padded = slot >= 0 && c->last_Lk > c->last_L && c->last_L == Lreal;
    const int L = padded ? c->last_Lk : Lreal;
    WtaParams w{};
    if (padded) {
        w.C = c->last_pad_c8 ? nullptr : (const float *)c->padf[slot].p + pix0 * L;
        w.cbytes = c->last_pad_c8 ? c->last_pad_cb : 1;
        w.C8 = c->last_pad_c8 ? c->last_pad_ptr[slot] + pix0 * L * w.cbytes : nullptr;
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
int run_wta_refine(mgm_ctx *c, const mgm_cv *C, long long pix0, long long npix, const float *lr, long long lr_stride,
                   int NDIR, int fix_overcount, int ridx, float *out, float *outcost, float *Sout,
                   const float *wlo, const float *whi, int slot)
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


// ---- ragged volumes on their range-proportional copies (mgm_pass_rel.hip, k_wta_rel; round 5) -------------------------------
// MGM_HIP_REL=0 (read at every call: tests switch it inside one process): ragged volumes keep the dense-hull kernels
bool rel_enabled()
{
    const char *e = getenv("MGM_HIP_REL");
    return !(e && atoi(e) == 0) && tune_num("rel", 1) != 0;
}

int rel_alloc(mgm_ctx *c, mgm_cv *cv, int slots, int cb)
{
    const size_t npix = (size_t)cv->nx * cv->ny, need = npix * (size_t)slots * (size_t)cb + npix * 16 + 16;
    HIPCHK(c, hipSetDevice(c->device));
    if (cv->rel_cap < need) {
        if (cv->relbuf) {
            HIPCHK(c, hipStreamSynchronize(c->stream));  // (a kernel may still be reading the old copy)
            (void)hipFree(cv->relbuf);
        }
        cv->relbuf = nullptr;
        cv->rel_cap = 0;
        if (dev_malloc((void **)&cv->relbuf, need) == hipSuccess) cv->rel_cap = need;
        else (void)hipGetLastError();
    }
    cv->rel_slots = slots;
    cv->rel_cb = cb;
    return MGM_OK;
}

int rel_resolve(mgm_ctx *c, const mgm_cv *ccv, bool *usable)
{
    mgm_cv *cv = const_cast<mgm_cv *>(ccv);
    *usable = false;
    if (!cv->rlo || !cv->relbuf || cv->rel_state == 0 || cv->rel_state == -1) return MGM_OK;
    if (cv->rel_state == 1) {
        HIPCHK(c, hipSetDevice(c->device));
        if (int r = ensure_words(c)) return r;
        // the flag word of the gathered copy: bit 0 = a window wider than the format's slots - 2, bit 1 = a cost without the format's
        // code.  (round 6) The copy is then gathered again one step wider -- 64 -> 128 slots, one -> two bytes per cost -- while the
        // fp32 hull it is gathered from is current; what fits neither keeps the dense hull.
        for (int round = 0; round < 4; round++) {
            HIPCHK(c, hipMemcpyAsync(c->h_words + 3, cv->rel_flag(), 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            const unsigned f = c->h_words[3];
            if (f == 0u) {
                cv->rel_state = 2;
                break;
            }
            const int slots = (f & 1u) ? cv->rel_slots * 2 : cv->rel_slots, cb = (f & 2u) ? cv->rel_cb * 2 : cv->rel_cb;
            if (slots > 128 || cb > 4 || !cv->f32_state || !cv->d || tune_num("rel_wide", 1) == 0) {  // (cb 4: the fp32 cost itself)
                cv->rel_state = -1;
                break;
            }
            if (int r = rel_alloc(c, cv, slots, cb)) return r;
            if (!cv->relbuf) {
                cv->rel_state = -1;
                break;
            }
            HIPCHK(c, hipMemsetAsync(cv->rel_flag(), 0, 4, c->stream));
            TimeScope t(c, "k_rel_gather");
            HIPCHK(c, launch_rel_gather(cv->d, cv->rlo, cv->rhi, (long long)cv->nx * cv->ny, cv->dmax - cv->dmin + 1, cv->dmin, slots, cb, cv->relbuf,
                                        cv->rel_records(), cv->rel_flag(), c->stream));
        }
        if (cv->rel_state == 1) cv->rel_state = -1;
    }
    *usable = cv->rel_state == 2;
    return MGM_OK;
}

int run_wta_rel(mgm_ctx *c, const mgm_cv *C, int slot, int NDIR, int fix_overcount, int ridx, const float *wlo, const float *whi, float *out,
                float *outcost, float *Sout)
{
    const size_t npix = (size_t)C->nx * C->ny;
    WtaRelParams w{};
    w.c8 = C->relbuf;
    w.base = C->rel_records();
    w.slots = C->rel_slots;
    w.cb = C->rel_cb;
    w.rlo = C->rlo;
    w.rhi = C->rhi;
    w.Lr = (const float *)c->lr_rel.p + (size_t)slot * NDIR * c->rel_last_stride;
    w.wlo = wlo;
    w.whi = whi;
    w.out = out;
    w.outcost = outcost;
    w.npix = (long long)npix;
    w.nvol = c->rel_last_stride;
    w.NDIR = NDIR;
    w.FIX = fix_overcount;
    w.refine = ridx;
    w.num_cu = c->num_cu;
    {
        TimeScope t(c, "k_wta");
        HIPCHK(c, launch_wta_rel(w, c->stream));
    }
    if (Sout) {  // the caller wants the corrected aggregated volume mgm() returns: on the dense hull
        TimeScope t(c, "k_rel_S");
        HIPCHK(c, launch_rel_S(w, C->dmax - C->dmin + 1, C->dmin, Sout, c->stream));
    }
    return MGM_OK;
}

int run_rel(mgm_ctx *c, const mgm_cv *const *Cs, const mgm_img *const *w8s, int nb, float P1, float P2, int MGM, int use_fh, int NDIR,
            int fix_overcount, int ridx, mgm_img *const *outs, mgm_img *const *outcosts, mgm_cv **S)
{
    const mgm_cv *C = Cs[0];
    const int nx = C->nx, ny = C->ny;
    const long long npix = (long long)nx * ny;
    const bool fh = use_fh > 0;
    const bool pube = !fh && !(w8s && w8s[0]);  // unit weights, Hirschmueller: the producer publishes E (k_pass_rel, PUBE)
    const int slots = C->rel_slots, rcb = C->rel_cb;  // (every volume of the launch has this format: aggregate_batch_now)
    const bool fh2 = fh && MGM == 2 && !(w8s && w8s[0]);  // update_cost2_trunclinear (the caller passes weights only if some weight != 1)
    const int R = pass_rel_lines(), HS = pass_rel_hand_floats(fh || pube, slots, fh2);
    int r;
    HIPCHK(c, hipSetDevice(c->device));
    if (int r0 = check_watchdog(c, false)) return r0;
    if ((r = ensure_words(c))) return r;
    unsigned *words = (unsigned *)c->words.p;
    RelParams p{};
    int maxLL = 0, maxbands = 0;
    int swapmask = 0;  // the passes walked with exchanged roles: part of what the hand-off region's layout and the task table depend on
    for (int q = 0; q < NDIR; q++) {
        // (round 6) form-0 passes with TSGM <= 3 walk slope 1 (make_geom decides; tune rel_slope1=0: slope 2 everywhere, as in round 5)
        if (!make_geom(q, nx, ny, R, MGM, tune_num("rel_slope1", 1) != 0, p.g[q])) return fail(c, MGM_ERR_INTERNAL, "pass table does not reduce to canonical form");
        maxLL = std::max(maxLL, p.g[q].LL);
        maxbands = std::max(maxbands, p.g[q].nbands);
        // (round 6) two strips per line for the form-1 passes: no pixel of those passes depends on its own line with TSGM <= 3, so the
        // two halves of a band's lines are two work items (k_pass_rel).  Measured, 1920x1080, windows of 49 labels: FH x 1 8.91 -> 8.12 ms,
        // x 4 13.33 -> 12.89; Hirschmueller x 1 6.13 -> 5.53 (tune rel_strips=0: none)
        // (round 6, later) ... and better: those passes ACROSS their lines, bands of anti-diagonals in lock step (k_pass_rel, g.diag) -- the
        // chain of a pass is NL + bands x lag steps instead of 2 NL + LL / 2 + bands x lag.  Two hand-off lines per band and a second hand
        // ring in LDS: where that does not fit (the three-slab entries of 128 slots with two-byte costs) the strips stay.  tune rel_diag=0: strips
        // (round 6, last) form-0 passes with TSGM <= 3 and more lines than pixels per line (the column passes of a landscape image) are walked
        // with the roles of i and j exchanged: (i - 1, j), (i, j - 1), (i - 1, j - 1) is symmetric in them, the depth NL + LL stays, but a band
        // trails the band before it by ~2 x 16 steps in practice (the pace of a chain is that of its slowest band), so FEWER bands of longer
        // lines end sooner: 68 x D + 1920 against 120 x D + 1080.  tune rel_swap=0: none
        // (FH x 1 5.73 -> 5.27 ms, x 2 7.34 -> 7.04, windows of 101 labels 8.58 -> 7.9, x 4 unchanged; the short steps of the Hirschmueller launches lose
        // 1-2 % with it -- 4.17 / 5.03 / 8.05 -> 4.19 / 5.16 / 8.16 -- and keep their walks: rel_swap=2 forces it there too)
        if ((tune_num("rel_swap", 1) >= 2 || (tune_num("rel_swap", 1) == 1 && fh)) && p.g[q].form == 0 && MGM <= 3 && p.g[q].slope == 1 && p.g[q].NL > p.g[q].LL) {
            PassGeom &g = p.g[q];
            std::swap(g.NL, g.LL);
            std::swap(g.istep, g.jstep);
            g.swap = 1;
            swapmask |= 1 << q;
            g.nbands = (g.NL + R - 1) / R;
            g.split = g.LL;
            maxLL = std::max(maxLL, g.LL);
        }
        const bool diag_ok = tune_num("rel_diag", 1) != 0 && p.g[q].form == 1 && MGM <= 3 && pass_rel_lds_bytes(fh || pube, slots, rcb, fh2, true) <= (size_t)160 * 1024;
        if (diag_ok) {
            PassGeom &g = p.g[q];
            g.diag = 1;
            g.slope = 0;
            g.nbands = (g.NL + g.LL - 1 + R - 1) / R;
            g.wmax = std::min(g.NL, g.LL + R);
            p.diag_any = 1;
        } else if (tune_num("rel_strips", 1) != 0 && p.g[q].form == 1 && MGM <= 3 && p.g[q].LL >= 8 * R) {
            p.g[q].nstrips = 2;
            p.g[q].split = p.g[q].LL / 2;
        }
        maxbands = std::max(maxbands, p.g[q].nbands);
    }
    if (maxbands > kMaxBands) return fail(c, MGM_ERR_UNSUPPORTED, "image side exceeds 65536 pixels");
    const long long stride = npix * slots + lr_pad_floats();
    if ((r = reserve(c, c->lr_rel, sizeof(float) * (size_t)stride * NDIR * nb))) return r;
    // self-validating hand-off slots, one per (volume, pass, band, pixel): written once per launch with the launch's tag (see
    // k_pass_rel); another geometry clears the region (all-ones words) and starts again with tag 0
    long long per_vol = 0;
    for (int q = 0; q < NDIR; q++) {
        p.g[q].hand_base = per_vol;
        per_vol += (long long)p.g[q].nbands * (p.g[q].diag ? 2LL * p.g[q].wmax : (long long)p.g[q].LL);
    }
    p.hand_vstride = per_vol;
    std::string tag_key;
    {
        const size_t bytes = sizeof(float) * (size_t)nb * per_vol * HS;
        const void *before = c->hand_rel.p;
        if ((r = reserve(c, c->hand_rel, bytes))) return r;
        char hk[128];
        snprintf(hk, sizeof hk, "%d %d %d %d %d %d %d %d", nx, ny, NDIR, nb, HS, R, slots, p.diag_any + 2 * swapmask);
        if (c->hand_rel.p != before || c->hand_rel_key != hk) {
            HIPCHK(c, hipMemsetAsync(c->hand_rel.p, 0xff, bytes, c->stream));
            c->hand_rel_key = hk;
            c->hand_rel_tag = 0x80000000u;  // (what the cleared words look like)
        }
        c->hand_rel_tag ^= 0x80000000u;
        p.tag = c->hand_rel_tag;
        // until the launch has been enqueued the region counts as unknown (an error return in between must not leave slots behind
        // that carry the tag of the launch after next)
        tag_key = c->hand_rel_key;
        c->hand_rel_key.clear();
    }
    // the task table: the simulated list schedule of the launch (one ticket counter), cached per shape
    char key[96];
    // workgroups (4 compute waves + the loader) per CU: tune rel_wg forces it
    const long long wgs = tune_num("rel_wg", 0);
    // (measured, round 6, 1920x1080 windows of 49 labels, FH with the side-by-side convolutions: x 1 7.87 / 8.20 / 8.53 ms at 1 / 2 / 3 per CU,
    // x 2 13.0 / 8.6 / 9.6, x 4 24.3 / 14.1 / 12.9; Hirschmueller the same order)
    // (with the anti-diagonal passes a single launch is no longer one long chain: x 1 6.89 / 6.32 / 6.47 ms at 1 / 2 / 3, x 2 11.5 / 7.82 / 7.76,
    // x 4 22.0 / 13.0 / 12.3; Hirschmueller x 1 5.45 / 4.62 / 4.36)
    const int rel_wg = wgs > 0 ? (int)std::min(wgs, 6LL) : (nb <= 1 ? (fh ? 2 : 3) : 3);
    snprintf(key, sizeof key, "%d %d %d %d %d %d %d %d %d %lld", nx, ny, NDIR, nb, rel_wg, p.g[0].slope, MGM, p.g[NDIR - 1].nstrips, p.diag_any + 2 * swapmask, tune_num("rel_prio", nb <= 1 ? 5 : 0) + 1000 * tune_num("rel_lag", 0) + 100000 * tune_num("rel_lagd", 0) + 10000000 * tune_num("rel_slots", 100));
    if (c->tasks_rel_key != key) {
        std::vector<SimChain> ch;
        for (int v = 0; v < nb; v++)
            for (int q = 0; q < NDIR; q++)
                for (int st = 0; st < p.g[q].nstrips; st++) {
                    const PassGeom &g = p.g[q];
                    SimChain k;
                    k.x = v * kMaxDirs + q, k.st = st, k.nb = g.nbands, k.chain = v * NDIR + q;
                    k.sib = g.nstrips == 2 ? (int)ch.size() + (st == 0 ? 1 : -1) : -1;  // (a band waits for BOTH strips of the band before it)
                    // (slope of the lock-step diagonal x lines + the lag the MODEL assumes per band.  The tickets are the start order of the
                    // simulated schedule, so these constants decide who holds a band slot while it waits: with the hand-off's real ~4 steps on
                    // every chain, the anti-diagonal bands -- ready every few steps -- took the first 400 of 512 slots and sat in them.  Measured
                    // (tools/ab_rel_lag.sh): 1-2 steps on the line walks, 7 on the anti-diagonals: x 1 6.15 -> 5.72 ms, Hirschmueller x 2 5.42 -> 5.0,
                    // windows of 101 labels 9.09 -> 8.5, four volumes unchanged; tune rel_lag / rel_lagd: added to them)
                    k.skew = (double)g.slope * R + (g.slope == 1 ? 1.0 : 2.0) + (double)tune_num("rel_lag", 0);
                    k.len = (g.nstrips == 2 ? (st == 0 ? g.split : g.LL - g.split) + R - 1 : g.LL) + (double)g.slope * (R - 1) + 1.0;
                    if (g.diag) {  // band b walks lines [lo_b, hi_b] of the pass, one step behind band b - 1 (+ the lag)
                        auto lo = [&](int b) { return std::max(0, b * R - g.LL + 1); };
                        auto hi = [&](int b) { return std::min(g.NL - 1, b * R + R - 1); };
                        k.bskew.assign(g.nbands, 0.0), k.blen.assign(g.nbands, 0.0), k.brem.assign(g.nbands, 0.0);
                        for (int b = 0; b < g.nbands; b++) {
                            k.bskew[b] = b > 0 ? (double)(lo(b) - lo(b - 1)) + 7.0 + (double)tune_num("rel_lagd", 0) : 0.0;
                            k.blen[b] = (double)(hi(b) - lo(b)) + 2.0;
                        }
                        for (int b = g.nbands - 1; b >= 0; b--) k.brem[b] = b == g.nbands - 1 ? k.blen[b] : std::max(k.blen[b], k.bskew[b + 1] + k.brem[b + 1]);
                    }
                    ch.push_back(k);
                }
        std::vector<int2> order;
        (void)simulate_schedule(ch, 1, std::max(1, (int)(c->num_cu * rel_wg * tune_num("rel_slots", 100) / 100)), 1 << 20, order);
        // workgroups that share a CU slow each other down (a step of 1.07 us becomes ~1.5): the bands of the longest chains -- what the
        // launch ends with -- get the issue priority (bit 24 of the task word; tune rel_prio=0: none, =100: every chain within x % of the longest)
        {
            const double pct = (double)tune_num("rel_prio", nb <= 1 ? 5 : 0);  // (x 1 6.25 -> 6.03 ms, Hirschmueller 4.36 -> 4.22; x 4 12.3 -> 12.4: not there)
            double longest = 0.0;
            for (const SimChain &k : ch) longest = std::max(longest, k.rem_of(0));
            for (int2 &t : order)
                for (const SimChain &k : ch)
                    if (k.x == t.x && k.st == ((t.y >> 16) & 0xff)) {
                        if (pct > 0.0 && k.rem_of(0) >= longest * (1.0 - pct / 100.0) && rel_wg > 1) t.y |= 1 << 24;
                        break;
                    }
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));  // (a launch that still reads the old table)
        if ((r = reserve(c, c->tasks_rel, sizeof(int2) * order.size()))) return r;
        HIPCHK(c, hipMemcpyAsync(c->tasks_rel.p, order.data(), sizeof(int2) * order.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->tasks_rel_key = key;
        c->ntasks_rel = (int)order.size();
    }
    bool weighted = false;
    for (int v = 0; v < nb; v++) {
        p.vol[v].c8 = Cs[v]->relbuf;
        p.vol[v].base = Cs[v]->rel_records();
        p.vol[v].rlo = Cs[v]->rlo;
        p.vol[v].rhi = Cs[v]->rhi;
        p.vol[v].Lr = (float *)c->lr_rel.p + (size_t)v * NDIR * stride;
        p.vol[v].w8 = (w8s && w8s[v]) ? w8s[v]->d : nullptr;
        weighted = weighted || p.vol[v].w8;
    }
    if (weighted)
        for (int v = 0; v < nb; v++)
            if (!p.vol[v].w8) return fail(c, MGM_ERR_INVALID, "mgm_aggregate_batch: weights for all volumes or for none");
    p.hand = (float *)c->hand_rel.p;
    p.ticket = words + 0;
    p.err = words + 1;
    p.tasks = (const int2 *)c->tasks_rel.p;
    p.npix = npix;
    p.nvol = stride;
    p.MGM = MGM;
    p.NDIR = NDIR;
    p.pass0 = 0;
    p.LLmax = maxLL;
    p.maxbands = kMaxBands;
    p.weighted = weighted ? 1 : 0;
    p.P1 = P1;
    p.P2 = P2;
    // the loader's lead: every step of it is a step of lag per band (round 5: 3 for batches; with the anti-diagonal passes 2 is as good there)
    p.ld = (int)std::min(5LL, std::max(2LL, tune_num("rel_ld", 2)));  // (2 against 3: x 2 7.76 / 7.81 ms, x 4 12.30 / 12.39)
    p.fh_multi = tune_num("rel_multi", 1) != 0 ? 1 : 0;
    p.cost2 = (pube && MGM == 2) ? 1 : 0;
    p.fh2 = fh2 ? 1 : 0;
    p.slots = slots;
    p.cb = rcb;
    p.tl = nullptr;
    // MGM_HIP_TIMELINE=<file>: one line per work item (tools/timeline.py) -- where the compute units' time goes
    const char *tl_file = getenv("MGM_HIP_TIMELINE");
    if (tl_file && *tl_file) {
        if ((r = reserve(c, c->dbg, sizeof(unsigned long long) * (8 + pass_rel_phases()) * (size_t)c->ntasks_rel))) return r;
        HIPCHK(c, hipMemsetAsync(c->dbg.p, 0, sizeof(unsigned long long) * (8 + pass_rel_phases()) * (size_t)c->ntasks_rel, c->stream));
        p.tl = (unsigned long long *)c->dbg.p;
    }
    HIPCHK(c, hipMemsetAsync(words, 0, sizeof(unsigned), c->stream));
    {
        TimeScope t(c, "k_pass_rel");
        HIPCHK(c, launch_pass_rel(p, c->ntasks_rel, fh, pube, rel_wg, c->stream));
    }
    c->hand_rel_key = tag_key;  // enqueued: every slot of the region will carry this launch's tag
    HIPCHK(c, hipMemcpyAsync(c->h_words + 1, words + 1, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    c->pending_check = true;
    if (p.tl) {
        const int nph = pass_rel_phases();
        std::vector<unsigned long long> d((size_t)c->ntasks_rel * (8 + nph));
        std::vector<int2> tk((size_t)c->ntasks_rel);
        HIPCHK(c, hipMemcpyAsync(d.data(), c->dbg.p, d.size() * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(tk.data(), c->tasks_rel.p, tk.size() * sizeof(int2), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (FILE *f = fopen(tl_file, "a")) {
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int i = 0; i < c->ntasks_rel; i++) {
                if (d[(size_t)i * 8]) t0 = std::min(t0, d[(size_t)i * 8]);
                t1 = std::max(t1, d[(size_t)i * 8 + 1]);
            }
            const double tick = 1e-2;  // wall_clock64: 100 MHz -> 0.01 us
            fprintf(f, "launch %d %d %d %d %d %.1f volumes %d fh %d mgm %d rel 1\n", nx, ny, 64, c->ntasks_rel, rel_wg, (double)(t1 - t0) * tick, nb, fh ? 1 : 0, MGM);
            for (int i = 0; i < c->ntasks_rel; i++) {
                const unsigned long long *w = &d[(size_t)i * 8];
                fprintf(f, "item %d %d %d %d %d %.2f %.2f %.2f %llu %llu %llu %llu %llu\n", i, tk[i].x, tk[i].y & 0xffff, 0, 0, (double)(w[0] - t0) * tick,
                        (double)(w[1] - t0) * tick, (double)w[2] * tick, w[3], w[4] & 0xffffffffull, w[4] >> 32, w[5], w[6]);
                if (nph) {  // (development build) clocks per wave: compute, publish, barrier x 4 waves; loader: issue, retire, barrier
                    const unsigned long long *q = &d[(size_t)c->ntasks_rel * 8 + (size_t)i * nph];
                    fprintf(f, "phases %d", i);
                    for (int k = 0; k < 16; k++) fprintf(f, " %llu", q[k]);
                    fprintf(f, "\n");
                }
            }
            fclose(f);
        }
    }
    // what the context's last aggregation was: this one (the dense Lr workspace no longer belongs to these volumes)
    c->last_batch = 0;
    c->last_ndir = 0;
    for (int v = 0; v < kMaxBatch; v++) c->last_cvs[v] = nullptr;
    c->rel_last_batch = nb;
    c->rel_last_ndir = NDIR;
    c->rel_last_stride = stride;
    for (int v = 0; v < kMaxBatch; v++) {
        c->rel_last_cvs[v] = v < nb ? Cs[v] : nullptr;
        c->rel_last_gens[v] = v < nb ? Cs[v]->gen : 0;
    }
    for (int v = 0; v < nb; v++) {
        float *Sout = nullptr;
        if (S) {
            if ((r = mgm_cv_create(c, nx, ny, Cs[v]->dmin, Cs[v]->dmax, &S[v]))) return r;
            Sout = S[v]->d;
        }
        if ((r = run_wta_rel(c, Cs[v], v, NDIR, fix_overcount, ridx, nullptr, nullptr, outs[v]->d, outcosts[v]->d, Sout))) return r;
    }
    return MGM_OK;
}
