/*
 * mgm_hip.h -- C ABI of libmgm_hip.so, the MI355X (gfx950) MGM stereo core.
 *
 * This is the drop-in boundary for the hot path of gfacciol/mgm:
 *
 *   allocate_and_fill_sgm_costvolume()  mgm_costvolume.h:337-424   -> mgm_costvolume_build[_dev]
 *   compute_mgm_weights()               mgm_weights.h:63-85        -> mgm_weights[_dev]
 *   mgm()                               mgm_core.cc:408-613        -> mgm_aggregate[_dev]
 *   subpixel_refinement_sgm()           mgm_refine.h:40-70         -> mgm_refine[_dev]
 *
 * which the reference's main() calls at mgm.cc:372-385 (and again for the
 * right-to-left run at mgm.cc:405-414).  INTEGRATION.md shows the replacement
 * of those call sites.
 *
 * Conventions
 *   - plain C: opaque handles, raw pointers, ints; every function returns an
 *     mgm_status (0 = MGM_OK) and never throws or aborts across the boundary;
 *     mgm_last_error(ctx) gives the text of the last failure on that context.
 *     A call that fails hands out nothing it created: an output handle is
 *     either not written at all or reset to NULL, never left pointing at a
 *     half-made object (a volume passed in to be refilled stays the caller's).
 *   - images are the reference's `struct Img` layout (img.h:35-51): planar
 *     float32, data[x + y*nx + c*nx*ny].
 *   - cost volumes are the reference's per-pixel `Dvec` order (dvec.cc:49-131)
 *     made dense: float32 [y][x][o], o = 0..L-1 <-> disparity dmin+o,
 *     L = dmax-dmin+1 (dvec.cc:60).  Reads outside [dmin,dmax] behave as +INF.
 *   - weights are 8 planes in the order W,E,S,N,NW,NE,SE,SW (mgm_weights.h:69).
 *   - one mgm_ctx per host thread; a ctx is bound to one device and owns one
 *     HIP stream.  The *_dev entry points enqueue on that stream and return
 *     without synchronising (results are ordered on the stream); entry points
 *     that take or return HOST buffers synchronise before returning.
 *   - there is NO CPU fallback: without a usable gfx950 device every compute
 *     entry point fails with MGM_ERR_HIP.
 */
#ifndef MGM_HIP_H_
#define MGM_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mgm_ctx mgm_ctx; /* device + stream + workspace */
typedef struct mgm_img mgm_img; /* device-resident planar float image */
typedef struct mgm_cv mgm_cv;   /* device-resident dense volume [ny][nx][L] */

typedef enum mgm_status {
    MGM_OK = 0,
    MGM_ERR_INVALID = 1,     /* bad argument */
    MGM_ERR_UNSUPPORTED = 2, /* valid in the reference, not built yet */
    MGM_ERR_HIP = 3,         /* HIP runtime error / no device */
    MGM_ERR_NOMEM = 4,
    MGM_ERR_INTERNAL = 5     /* in-kernel watchdog fired (dataflow hand-off timed out) */
} mgm_status;

/* ---- context ---------------------------------------------------------- */
/* MGM_ERR_HIP if `device` does not exist or is not a gfx950 part. */
int mgm_ctx_create(int device, mgm_ctx **ctx);
int mgm_ctx_destroy(mgm_ctx *ctx);
const char *mgm_last_error(const mgm_ctx *ctx);
int mgm_ctx_synchronize(mgm_ctx *ctx);
/* Hand the context's grow-only workspace (Lr volumes of the largest aggregation so far, hand-off slots, ...) back to the
 * device; it is allocated again on demand.  Synchronises. */
int mgm_ctx_trim(mgm_ctx *ctx);
/* Cap, in bytes, on the workspace ONE pass launch may take (the NDIR Lr volumes of every batched volume and the
 * hand-off slots; 0 = no cap, the default).  mgm_aggregate_batch_dev runs a batch that exceeds it -- or that the
 * device cannot hold -- as several launches over the largest sub-batches that fit instead of returning MGM_ERR_NOMEM. */
int mgm_ctx_set_workspace_limit(mgm_ctx *ctx, unsigned long long bytes);
/* PIPELINED CONTEXT for a caller that has ONE volume (or one small batch) at a time -- a stream of stereo pairs.
 * depth 1 (the default): every call runs when it is made.  depth D = 2..16: mgm_aggregate[_batch]_dev calls are DEFERRED --
 * checked, remembered, MGM_OK returned -- until D of them have been gathered, and those then run as ONE batch: one launch
 * of the pass kernel over all their volumes (what mgm_aggregate_batch_dev does for a caller that has the volumes
 * together: a single volume's launch is bound by its chains of bands, and the other volumes' bands fill its gaps; round
 * 4, 1920x1080x256, 8 directions, FH: 0.70 of the roofline one volume at a time, 0.82 with D = 2, 0.87 with D = 4).  Every
 * volume gets exactly the result of an unpipelined call.  The workspace holds the Lr volumes of D calls.
 * What the caller must know: (1) results exist once the D-th call of a group, mgm_ctx_synchronize or any other entry point
 * of the context that is not part of gathering a step has returned (downloads, post-processing, frees and the rest first
 * run what has been deferred); (2) consecutive calls need DIFFERENT cost volumes, weight images and output images (D sets,
 * used in turn) -- refilling a volume or weights that a deferred call still needs, or naming an output image twice, is
 * still correct (the deferred calls then run first) but defeats the gathering; (3) an error of the deferred work is
 * returned by the call that made it run.  A call that does not fit the waiting ones (other geometry or settings, S
 * wanted) makes them run first.  Synchronises; MGM_ERR_INVALID for a depth outside 1..16. */
int mgm_ctx_set_pipeline(mgm_ctx *ctx, int depth);
/* PLACEMENT of the Lr workspace.  Identical launches run on plateaus up to 16 % apart depending on which physical pages the
 * allocator handed out for the workspace (docs/experiments.md, rounds 4 and 5: same code, data and virtual address).  With
 * tries = n >= 2 the context, whenever an aggregation has just (re)allocated its workspace, times that aggregation's pass
 * launch on up to n allocations -- the earlier ones held meanwhile, so that each lands on other pages -- and keeps the
 * fastest.  Results do not change (the launch is simply repeated on the same inputs); the first aggregation after a
 * (re)allocation takes n - 1 allocations and 2 n - 1 launches longer, and needs room for two workspaces (skipped when the
 * device does not have it).  0 (default) / 1: take what the allocator gives.  For long-lived contexts (resident mode,
 * services, benchmarks); a one-shot command line would pay more than it gains. */
int mgm_ctx_set_placement_tries(mgm_ctx *ctx, int tries);
/* Free and total device memory in bytes as the runtime sees them now (hipMemGetInfo on the context's device): what a
 * caller sizes its batches -- or mgm_ctx_set_workspace_limit -- with.  Either pointer may be NULL. */
int mgm_ctx_mem_info(mgm_ctx *ctx, unsigned long long *free_bytes, unsigned long long *total_bytes);
void *mgm_ctx_stream(mgm_ctx *ctx); /* the hipStream_t everything is enqueued on */
const char *mgm_version(void);

/* Per-kernel timing with HIP events on the ctx stream.  While enabled every
 * kernel launch is bracketed by an event pair; mgm_timing_get() synchronises
 * and reports elapsed milliseconds per launch in launch order. */
int mgm_timing_enable(mgm_ctx *ctx, int enable);
int mgm_timing_reset(mgm_ctx *ctx);
int mgm_timing_count(mgm_ctx *ctx);
int mgm_timing_get(mgm_ctx *ctx, int idx, const char **kernel_name, float *ms);

/* ---- images (struct Img, img.h:9-59) ----------------------------------- */
int mgm_img_create(mgm_ctx *ctx, int nx, int ny, int nch, mgm_img **img);
int mgm_img_upload(mgm_ctx *ctx, const float *host, int nx, int ny, int nch, mgm_img **img);
/* Refill an existing image from a host buffer of its own size (no allocation; synchronises, so the buffer is the caller's
 * again on return): a caller with a stream of same-sized inputs keeps its device images (src/mgm_main.cc, resident mode). */
int mgm_img_update(mgm_ctx *ctx, mgm_img *img, const float *host);
int mgm_img_download(mgm_ctx *ctx, const mgm_img *img, float *host);
int mgm_img_dims(const mgm_img *img, int *nx, int *ny, int *nch);
/* The raw device pointer.  Does NOT synchronise and does NOT run deferred work: on a pipelined context
 * (mgm_ctx_set_pipeline) call mgm_ctx_synchronize first if the image is an output of a deferred aggregation. */
void *mgm_img_device_ptr(mgm_img *img);
int mgm_img_device(const mgm_img *img); /* HIP device ordinal the pixels live on (-1: NULL) */
int mgm_img_free(mgm_ctx *ctx, mgm_img *img);

/* ---- volumes (costvolume_t, mgm_costvolume.h:311-327) ------------------ */
int mgm_cv_create(mgm_ctx *ctx, int nx, int ny, int dmin, int dmax, mgm_cv **cv);
int mgm_cv_upload(mgm_ctx *ctx, const float *dense, int nx, int ny, int dmin, int dmax, mgm_cv **cv);
int mgm_cv_download(mgm_ctx *ctx, const mgm_cv *cv, float *dense);
int mgm_cv_dims(const mgm_cv *cv, int *nx, int *ny, int *dmin, int *dmax);
void *mgm_cv_device_ptr(mgm_cv *cv);
int mgm_cv_device(const mgm_cv *cv); /* HIP device ordinal of the context that made the volume (-1: NULL) */
int mgm_cv_free(mgm_ctx *ctx, mgm_cv *cv);

/* ---- cost volume: allocate_and_fill_sgm_costvolume --------------------- */
/* prefilter in {"none","census","sobelx","gblur"}, distance in {"ad","sd","census",
 * "ncc","btad","btsd"}; unknown names silently select the first entry, as the
 * reference does (mgm_costvolume.h:184-190, 201-207).  census_win is the value
 * of the reference's CENSUS_NCC_WIN environment parameter (mgm_costvolume.h:61).
 * Every entry of both tables is built on the device.  *C must be NULL (a new volume is allocated) or a volume of
 * the same geometry, which is then refilled in place (no allocation, no sync). */
int mgm_costvolume_build_dev(mgm_ctx *ctx, const mgm_img *u, const mgm_img *v, int dmin, int dmax,
                             const char *prefilter, const char *distance, float truncDist, int census_win,
                             mgm_cv **C);
/* Host-buffer form taking the reference's per-pixel range images (dminI/dmaxI, mgm.cc:338-353), converted to
 * int as Dvec's constructor does (mgm_costvolume.h:323).  Uniform ranges take the fast path.  RAGGED ranges
 * (-m/-M files) give a volume over the hull of all ranges in which a pixel only owns the disparities of its own
 * range -- the others read +INF, as Dvec::operator[] does (dvec.cc:129), and are exempt from the "no finite cost"
 * rule; mgm_aggregate* then searches the winner and gates the refinement inside each pixel's range.  The hull may
 * span any number of labels the device can hold (the reference's Dvec has no limit, dvec.cc:60; an index-arithmetic cap of
 * 4 194 304 aside): up to 1024 on the fast kernels, up to 2048 on the first build of the pass kernel, beyond that on the
 * generic ones (a second or more per full-HD volume); batched ragged volumes must share hull_min under FH potentials.
 * A ragged volume whose windows are at most 126 labels wide also keeps a RANGE-PROPORTIONAL copy (64 or 128 label slots per pixel at
 * the pixel's own window, mgm_costvolume.h:275-299), which mgm_aggregate* then walks instead of the hull: same results, bytes and
 * steps proportional to the labels that exist. */
int mgm_costvolume_build(mgm_ctx *ctx, const float *u, const float *v, int nx, int ny, int nch, int vnx, int vny,
                         const float *dminI, const float *dmaxI, const char *prefilter, const char *distance,
                         float truncDist, int census_win, mgm_cv **C);

/* Device form of the ragged build: dminI/dmaxI are nx*ny device images, [hull_min, hull_max] must contain every
 * pixel's (int) range. */
int mgm_costvolume_build_ranged_dev(mgm_ctx *ctx, const mgm_img *u, const mgm_img *v, const mgm_img *dminI,
                                    const mgm_img *dmaxI, int hull_min, int hull_max, const char *prefilter,
                                    const char *distance, float truncDist, int census_win, mgm_cv **C);

/* ---- edge weights: compute_mgm_weights --------------------------------- */
/* compute_mgm_weights (mgm_weights.h:63-85; called at mgm.cc:372 when -aP2 != 1): 8 planes of nx*ny weights.  *w8 == NULL
 * on entry: a new image is created; non-NULL: that nx*ny*8 image is refilled (like *C of mgm_costvolume_build_dev). */
int mgm_weights_dev(mgm_ctx *ctx, const mgm_img *u, float aP, float aThresh, mgm_img **w8);

/* ---- aggregation + WTA: mgm() ------------------------------------------ */
/* C is not modified.  w8 may be NULL (all ones).  A volume whose aggregation can meet NaNs -- NaN costs (an uploaded
 * volume is scanned once per filling; `-p census` with another distance from descriptors of more than 24 bits), or a ragged
 * volume with P2 = +INF (all-INF slabs, then INF - INF) -- is aggregated by a slow kernel that keeps the OPERAND ORDER of
 * the reference's minima (`a < b ? a : b`, mgm_core.cc:48-60; dvec.cc:81-88), which decides what a NaN does there: the
 * results are still the reference's, at a second or so per full-HD volume instead of milliseconds.  As in the reference
 * (mgm_core.cc:420-423) a single weight != 1.0 anywhere switches the whole run
 * to the weighted update functions.  P1/P2 are used as given (the caller has
 * already multiplied by the channel count, mgm.cc:356-357).
 *   NDIR 1..8, MGM (TSGM) 1..4, use_fh: USE_TRUNCATED_LINEAR_POTENTIALS,
 *   fix_overcount: TSGM_FIX_OVERCOUNT.
 * refine: NULL/"none" gives mgm()'s own outputs (integer labels dmin+argmin);
 *   "vfit" additionally applies subpixel_refinement_sgm in the same kernel; "parabola", "cubic" and
 *   "parabolaOCV" apply it as a second kernel on the corrected S.
 * out/outcost: nx*ny images.  S: NULL, or receives the corrected aggregated
 *   volume that mgm() returns (costs one extra volume write). */
int mgm_aggregate_dev(mgm_ctx *ctx, const mgm_cv *C, const mgm_img *w8, float P1, float P2, int NDIR, int MGM,
                      int use_fh, int fix_overcount, const char *refine, mgm_img *out, mgm_img *outcost,
                      mgm_cv **S);
int mgm_aggregate(mgm_ctx *ctx, const mgm_cv *C, const float *w8, float P1, float P2, int NDIR, int MGM, int use_fh,
                  int fix_overcount, const char *refine, float *out, float *outcost, mgm_cv **S);

/* n (1..16) volumes of identical size and label count aggregated by ONE launch of the pass kernel:
 * the two runs of mgm() that main() makes for a stereo pair, left->right (mgm.cc:376-385) and
 * right->left (mgm.cc:405-414), or the volumes of consecutive pairs.  Every volume gets exactly the
 * result mgm_aggregate_dev would give it; the point is throughput -- the scan-line passes of one
 * volume form dependency chains (band after band) that leave compute units waiting, and the other
 * volumes' bands fill those gaps (a throughput-bound launch puts two bands on a compute unit; at 128 / 64
 * labels two / four volumes share every wavefront).  The workspace holds NDIR fp32 volumes per batched volume.  C, out, outcost: arrays of n handles.  w8: NULL, or an array of n
 * weight images (for all volumes or none; all weighted or all unweighted).  S: NULL, or an array of
 * n handles to receive the corrected aggregated volumes. */
int mgm_aggregate_batch_dev(mgm_ctx *ctx, int n, const mgm_cv *const *C, const mgm_img *const *w8, float P1, float P2,
                            int NDIR, int MGM, int use_fh, int fix_overcount, const char *refine, mgm_img *const *out,
                            mgm_img *const *outcost, mgm_cv **S);

/* ---- one volume sharded by DIRECTION over several GPUs ---------
 * Each rank holds the full C (rebuilt from the images: cheaper than broadcasting it) and runs
 * a contiguous subset of the passes; pass first_pass+k's Lr volume stays in workspace slot k
 * (mgm_lr_device_ptr).  The ranks then exchange ROW SLABS of those volumes (grouped
 * send/recv over RCCL -- mgm_amd/dist.py; an all-reduce would change the fp32 summation order)
 * and every rank finishes its rows with mgm_wta_rows_dev, whose `lr_slabs` is a device buffer
 * [NDIR][nrows][nx][L] holding ALL passes of rows row0..row0+nrows-1 in pass order;
 * out_rows / outcost_rows are device buffers of nrows*nx floats. */
int mgm_aggregate_passes_dev(mgm_ctx *ctx, const mgm_cv *C, const mgm_img *w8, float P1, float P2, int MGM, int use_fh,
                             int first_pass, int n_passes);
/* The same with the caller placing the Lr volumes: pass first_pass+k lands in workspace slot slot0+k of n_slots, and
 * the hand-off region is laid out once for the passes [0, NDIR_total).  For callers that launch the passes of a
 * volume ONE AT A TIME so that pass k's slabs travel while pass k+1 runs (mgm_multi_aggregate, mgm_amd/dist.py). */
int mgm_aggregate_passes_at_dev(mgm_ctx *ctx, const mgm_cv *C, const mgm_img *w8, float P1, float P2, int MGM, int use_fh,
                                int first_pass, int n_passes, int slot0, int n_slots, int NDIR_total);
void *mgm_lr_device_ptr(mgm_ctx *ctx, int slot);
int mgm_wta_rows_dev(mgm_ctx *ctx, const mgm_cv *C, int row0, int nrows, const void *lr_slabs, int NDIR,
                     int fix_overcount, const char *refine, void *out_rows, void *outcost_rows);

/* ---- the same, driven from ONE host thread for n GPUs of a node: mgm_multi ---------------------------------------
 * An mgm_ctx per device plus a transport for the row slabs.  The CPU analogue in the reference is
 * mgm_naive_parallelism (mgm_core.cc:632-831, WITH_MGM2=1): passes in parallel on private Lr volumes, then accumulated
 * -- here always in pass order, so the result is mgm()'s bit for bit.
 *   mgm_multi_create     device_ids: n distinct HIP devices (gfx950).  Transport (mgm_multi_transport tells which):
 *                          "rccl"      grouped ncclSend/ncclRecv on a communicator over the devices (librccl is loaded on
 *                                      first use) -- the default;
 *                          "peer"      the receiving device PULLS each slab with a device-to-device copy on a second
 *                                      stream, ordered by events: MGM_MULTI_TRANSPORT=peer, and what the handle falls back
 *                                      to when librccl cannot be loaded or its communicator cannot be made;
 *                          "loopback"  a device id appears several times (allowed only with MGM_MULTI_LOOPBACK=1): the
 *                                      ranks are contexts on one GPU, copies as for "peer" -- the n-rank path on a one-GPU
 *                                      box (tests).
 *                        On failure *m is NULL and mgm_multi_last_error(NULL) holds the reason.
 *   mgm_multi_ctx        rank k's context: build C[k] on it (mgm_costvolume_build_dev from images uploaded there).
 *   mgm_multi_aggregate  mgm() of ONE volume: C[k] = the same cost volume on every device (rebuilt there from the
 *                        images: cheaper than moving it; MGM_ERR_INVALID if C[k] or w8[k] does not live on device k, or
 *                        out0 / outcost0 not on device_ids[0]), w8 NULL or one weight image per device; rank k runs the
 *                        passes of mgm_multi_plan's block k, the row slabs move (one peer per xGMI link, all links at
 *                        once), every rank sums its rows in pass order and searches them, and the rows are gathered into
 *                        out0 / outcost0.  Synchronous; an exchange that does not complete within
 *                        MGM_MULTI_TIMEOUT_S (default 120) seconds returns MGM_ERR_HIP after aborting the transport (the
 *                        handle is then only good for mgm_multi_destroy).  A rank's failure is reported after EVERY
 *                        rank's queued work has drained, never in the middle of a transfer group.
 *                        MGM_MULTI_OVERLAP=1: ranks with several passes launch them one per launch and each pass's
 *                        slabs leave behind it, while the next pass runs.
 *   mgm_multi_plan       the partition as data: contiguous blocks of passes and of rows per rank (sizes differ by <= 1). */
typedef struct mgm_multi mgm_multi;
int mgm_multi_create(const int *device_ids, int n, mgm_multi **m);
int mgm_multi_destroy(mgm_multi *m);
int mgm_multi_size(const mgm_multi *m);
mgm_ctx *mgm_multi_ctx(mgm_multi *m, int rank);
const char *mgm_multi_last_error(const mgm_multi *m); /* m == NULL: why the last mgm_multi_create of this thread failed */
const char *mgm_multi_transport(const mgm_multi *m);  /* "rccl", "peer" or "loopback" */
int mgm_multi_plan(int n, int NDIR, int ny, int *first_pass, int *n_passes, int *row0, int *nrows);
int mgm_multi_aggregate(mgm_multi *m, const mgm_cv *const *C, const mgm_img *const *w8, float P1, float P2, int NDIR, int MGM,
                        int use_fh, int fix_overcount, const char *refine, mgm_img *out0, mgm_img *outcost0);

/* Test/diagnostic aid: copy pass `pass`'s Lr volume of the LAST mgm_aggregate
 * call on this ctx into `dense` ([ny][nx][L]). */
int mgm_debug_download_lr(mgm_ctx *ctx, int pass, float *dense);

/* Diagnostic (tools/bimodal_probe.py): the rate, in GB/s, at which the store pattern of the pass kernels -- `nstreams`
 * volumes at the stride of the context's last aggregation, written side by side -- lands on the Lr workspace where the
 * allocator placed it.  MGM_ERR_INVALID before the context's first aggregation. */
int mgm_debug_probe_workspace(mgm_ctx *ctx, int nstreams, float *gbps);

/* Device self-test: compares the kernels' three-operation exact x/3 against IEEE
 * division for all 2^32 float32 inputs; *nbad receives the number of inputs
 * whose results differ (NaN == NaN). */
int mgm_selftest_div3(mgm_ctx *ctx, unsigned long long *nbad);

/* ---- sub-pixel refinement: subpixel_refinement_sgm ---------------------- */
/* method in {"none","vfit","parabola","cubic","parabolaOCV"}; unknown => none
 * (mgm_refine.h:28-35).  out/outcost are read and updated.  (vfit is also fused into mgm_aggregate*'s
 * WTA kernel; the other methods run as a second kernel on the corrected S, in mgm_aggregate* too.) */
int mgm_refine_dev(mgm_ctx *ctx, const mgm_cv *S, const char *method, mgm_img *out, mgm_img *outcost);
int mgm_refine(mgm_ctx *ctx, const mgm_cv *S, const char *method, float *out, float *outcost);

/* ---- TSGM_ITER > 1: main()'s loop mgm.cc:377-388 ------------------------------------------------------------
 * The reference builds the cost volume once and calls mgm() TSGM_ITER times with range images that
 * update_dmin_dmax narrows around the previous solution.  The volume is the same every time, hence so are the
 * scan-line passes: only the winner search and the refinement see the new ranges.  mgm_wta_windowed_dev does
 * exactly that part again on the Lr volumes the context still holds from its last mgm_aggregate_dev(C, NDIR)
 * call: the winner of pixel p is the first strict minimum of S over the disparities [(int)dminI(p), (int)dmaxI(p)]
 * (mgm_core.cc:592-609 with S allocated from those images, 426), the refinement gate (mgm_refine.h:58) uses the
 * same window, and a disparity of the window outside C's range holds what it holds there: 0 - (NDIR-1)*INF, or 0
 * without the over-count fix.  Any refinement name. */
int mgm_wta_windowed_dev(mgm_ctx *ctx, const mgm_cv *C, int NDIR, int fix_overcount, const char *refine,
                         const mgm_img *dminI, const mgm_img *dmaxI, mgm_img *out, mgm_img *outcost);
/* update_dmin_dmax (mgm.cc:120-158; main() uses slack 3, radius 2) followed by the two
 * remove_nonfinite_values_Img calls (mgm.cc:387-388): dminI/dmaxI are updated in place from the disparity map. */
int mgm_update_ranges_dev(mgm_ctx *ctx, const mgm_img *outoff, mgm_img *dminI, mgm_img *dmaxI, int slack, int radius);

/* ---- what main() does to the disparity maps right after the path (device images in, device images out) ---- */
/* median_filter (img_tools.h:203-238, called at mgm.cc:396, 419 when MEDIAN != 0): per channel, the window
 * (2*radius+1)^2 clipped at the border, NaN samples ignored, the upper median v[n/2]; an all-NaN window
 * leaves the pixel unchanged.  Any radius >= 1 (beyond 7 the order statistic is found by radix selection instead of
 * pairwise counting: 33 sweeps of the window per pixel, so the call is bounded -- MGM_ERR_UNSUPPORTED when
 * 33 * (2*radius+1)^2 * nx*ny*nch exceeds 1e13 window reads, i.e. beyond radius ~190 at 1920x1080; up to that it
 * takes at most about a minute).  out must have in's size (and must not be in). */
int mgm_median_dev(mgm_ctx *ctx, const mgm_img *in, int radius, mgm_img *out);
/* leftright_test (mgm.cc:68-91, called at 420-423): out[x,y] = d[x,y] if Lx = round(x + d) lies inside `other`
 * and |Lx + other[Lx,y] - x| <= tau, NaN otherwise.  d and out have one size, `other` may have another width. */
int mgm_leftright_dev(mgm_ctx *ctx, const mgm_img *d, const mgm_img *other, float tau, mgm_img *out);
/* the back-projected image main() writes as its optional third output (mgm.cc:433-443): v sampled at x + disp
 * (integer truncation of the reference's float index), u where that falls outside v. */
int mgm_backproject_dev(mgm_ctx *ctx, const mgm_img *u, const mgm_img *v, const mgm_img *disp, mgm_img *out);

#ifdef __cplusplus
}
#endif
#endif /* MGM_HIP_H_ */
