/*
 * cmlhip.h — C ABI of the MI355X (gfx950) photometric hot path for libCML / MODSLAM.
 *
 * This is the drop-in boundary: the unchanged cml:: host classes (DSOTracker,
 * DSOBundleAdjustment) keep their public C++ API and call these entry points from
 * the bodies of computeResidual/computeHessian (tracker) and
 * linearizeAll/solveSystem (bundle adjustment).  The reference has no C ABI for
 * this path; each entry point below cites the reference code
 * (path:line under the reference tree, abbreviations: BA.cpp =
 * src/cml/optimization/dso/DSOBundleAdjustment.cpp, TR.cpp =
 * src/cml/optimization/dso/DSOTracker.cpp, ACC.h =
 * src/cml/optimization/dso/MatrixAccumulators.h) whose inner loop it replaces.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer argument is HOST memory that is
 *    borrowed for the duration of the call, unless its name ends in _dev;
 *  - all matrices are ROW-MAJOR unless stated; poses are world->camera (R, t),
 *    p_cam = R p_world + t (src/cml/map/Camera.h:307-315);
 *  - every function returns a cmlhip_status (0 = ok); nothing throws or aborts;
 *    non-finite results are reported as CMLHIP_ERR_NONFINITE so the host branches
 *    at BA.cpp:836-841,1484-1492 and TR.cpp:121-138 still fire;
 *  - one context owns one HIP stream; contexts are independent and re-entrant
 *    (tracker ctx and BA ctx may be driven from two host threads);
 *  - there is NO CPU fallback: if no gfx950 device is usable, create() fails.
 */
#ifndef CMLHIP_H
#define CMLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMLHIP_ABI_VERSION 1
#define CMLHIP_PATTERN 8        /* star8, src/cml/types.h:1381-1407; DSOMAXRESPERPOINT, DSOResidual.h:10 */
#define CMLHIP_CPARS 4          /* calibration block size, ACC.h:26 */
#define CMLHIP_MAX_FRAMES 32    /* window size limit of the upload / accumulate / Schur kernels (reference default maxFrames = 6, BA.h:271) */
/* Hard caps that are REFUSALS (CMLHIP_ERR_INVALID with a message in cmlhip_last_error), not fallbacks:
 *   - cmlhip_ba_iteration_batch and the hybrid (ORB) term of cmlhip_ba_iteration_async: windows whose factorisation is LDS-resident,
 *     8N (+4 with optimize_calibration) <= 160, i.e. N <= CMLHIP_MAX_SOLVE_FRAMES (19 with the calibration block).
 *     cmlhip_ba_solve and the plain cmlhip_ba_iteration_async take every window up to CMLHIP_MAX_FRAMES: above that size the
 *     factorisation runs in global memory (one workgroup, ~1 ms at N = 32; a correctness path — the reference's windows hold 6-7 keyframes).
 *   - cmlhip_pnp_optimize: at most CMLHIP_PNP_MAX_MATCHES matches (LDS-resident match list).
 *   - the dense LDL^T is unpivoted (Eigen's ldlt() pivots): it relies on the Jacobi scaling 1/sqrt(diag + 10) of BA.cpp:1312-1316,
 *     which bounds the scaled diagonal; a non-finite or non-positive pivot is reported as CMLHIP_ERR_NONFINITE. */
#define CMLHIP_MAX_SOLVE_FRAMES 20
#define CMLHIP_PNP_MAX_MATCHES 2560
#define CMLHIP_RJ_FLOATS 74     /* sizeof(DSORawResidualJacobian)/4, DSOResidual.h:22-69 */

typedef enum {
    CMLHIP_OK = 0,
    CMLHIP_ERR_INVALID = 1,      /* bad argument / size over the limits given at create */
    CMLHIP_ERR_HIP = 2,          /* HIP runtime error, see cmlhip_last_error */
    CMLHIP_ERR_NONFINITE = 3,    /* a result contains NaN/Inf */
    CMLHIP_ERR_NOT_FOUND = 4,    /* unknown image id / level */
    CMLHIP_ERR_STATE = 5,        /* call order violated (e.g. linearize before upload) */
    CMLHIP_ERR_TIMEOUT = 6       /* a bounded device-side wait between workgroups gave up (the launch could not be co-resident): results void */
} cmlhip_status;

/* residual states, DSOResidual.h:14-16 */
enum { CMLHIP_RES_IN = 0, CMLHIP_RES_OOB = 1, CMLHIP_RES_OUTLIER = 2 };
/* accumulation modes, DSOResidual.h:18-20 */
enum { CMLHIP_MODE_ACTIVE = 0, CMLHIP_MODE_LINEARIZED = 1, CMLHIP_MODE_MARGINALIZED = 2 };
/* pyramid texel storage */
enum { CMLHIP_TEXEL_F32 = 0, CMLHIP_TEXEL_F16 = 1 };

typedef struct cmlhip_ctx cmlhip_ctx;

typedef struct {
    int device_id;       /* HIP device ordinal */
    int max_frames;      /* N  <= CMLHIP_MAX_FRAMES */
    int max_points;      /* P  */
    int max_residuals;   /* R  */
    int max_tracker_points; /* per level, upper bound of DSOTrackerPrivateLevel::n() (TR.h:62) */
    int max_reproj_obs;  /* ORB observations (BA.cpp:2607-2659) */
    int texel_format;    /* CMLHIP_TEXEL_F32 | CMLHIP_TEXEL_F16 (config E: fp16 taps, fp32 accumulate).  With fp16 texels a frame that a
                          * BA window names keeps a second, tiled copy of its level 0 (8 more bytes per pixel) for the resident loop's
                          * residual kernel; it is built on the device at the first cmlhip_ba_upload_window that names the image and
                          * is refreshed by a same-size cmlhip_pyramid_put.  A window holds pointers into its images: rebuilding
                          * (cmlhip_pyramid_build, a put with another size) or dropping an image the uploaded window names releases
                          * those blocks and INVALIDATES the window — the BA entry points then return CMLHIP_ERR_STATE until the next
                          * cmlhip_ba_upload_window. */
} cmlhip_limits;

/* ---------------------------------------------------------------- context */
int  cmlhip_abi_version(void);
int  cmlhip_device_count(void);
int  cmlhip_create(cmlhip_ctx** out, const cmlhip_limits* limits);
/* Several contexts on ONE device at the same time (sequence shards per GPU, BASELINE.json configs[3] scaled down to one device): n_contexts tells
 * this context how many launch beside each other.  Launches whose workgroups WAIT for one another (the tracker batch: G workgroups per hypothesis
 * exchanging partial sums) must be resident as a whole; they size themselves for 1 / n_contexts of the device's capacity.  Default 1. */
int  cmlhip_set_device_share(cmlhip_ctx* ctx, int n_contexts);
void cmlhip_destroy(cmlhip_ctx* ctx);
const char* cmlhip_last_error(const cmlhip_ctx* ctx);
int  cmlhip_synchronize(cmlhip_ctx* ctx);
/* raw hipStream_t of the context (for hipEvent timing by the caller) */
void* cmlhip_stream(cmlhip_ctx* ctx);

/* ---------------------------------------------------------------- pyramids
 * Device cache of GradientImage = Array2D<Vector3f>{I, dI/dx, dI/dy}
 * (src/cml/types.h:915, src/cml/image/Array2D.h:288-327), keyed by the host
 * image id (Array2D::getId(), src/cml/capture/CaptureImage.cpp:12-18,264).
 * Host layout: AoS, 3 floats per texel, x fastest, data[(y*w+x)*3 + c].
 * Device layout: 4 floats (or 4 halves) per texel {I,dx,dy,0} so one texel is one
 * aligned 16-B (8-B) load and a bilinear tap pair is one aligned 32-B segment.
 */
int cmlhip_pyramid_put(cmlhip_ctx* ctx, uint64_t image_id, int level,
                       const float* aos3, int w, int h);
/* Build the whole pyramid on the device from the level-0 gray image: 2x2 box mean
 * (Array2D.h:388-401), central-difference gradient with a zero 1-px border
 * (Array2D.h:288-327), level sizes by integer halving
 * (src/cml/capture/CaptureImage.cpp:39-78).  levels <= 8. Also keeps the gray
 * levels (needed by tracker reference colors, TR.cpp:702). */
int cmlhip_pyramid_build(cmlhip_ctx* ctx, uint64_t image_id, const float* gray,
                         int w, int h, int levels);
/* The same build, handed to the context's image worker: the call returns once the levels are allocated; the staging copy of `gray`, its
 * transfer and the kernels of every level run on a host thread and a stream of their own — beside whatever the context is doing (the
 * reference builds a frame's pyramid on the capture thread, ahead of the SLAM thread: capture/CaptureImage.cpp:39-78,216-221).  `gray` must stay
 * valid and unchanged until the first call that names image_id returns (that call orders itself behind the build) or until
 * cmlhip_pyramid_drop(image_id).  An image_id that is already in the cache is rebuilt synchronously (cmlhip_pyramid_build). */
int cmlhip_pyramid_build_async(cmlhip_ctx* ctx, uint64_t image_id, const float* gray, int w, int h, int levels);
int cmlhip_pyramid_drop(cmlhip_ctx* ctx, uint64_t image_id);   /* CaptureImage::makeUnactive, CaptureImage.cpp:364-403 */
int cmlhip_pyramid_level_size(cmlhip_ctx* ctx, uint64_t image_id, int level, int* w, int* h);
/* read one level back as AoS3 floats (tests) */
int cmlhip_pyramid_get(cmlhip_ctx* ctx, uint64_t image_id, int level, float* aos3_out);

/* ---------------------------------------------------------------- tracker (a2-a5)
 * Replaces DSOTracker::computeResidual + computeHessian (TR.cpp:248-492) and
 * makeCoarseDepthL0 (TR.cpp:494-724).
 */
typedef struct {
    float huber;             /* mHuberThreshold 9 (TR.h:478) */
    float cutoff;            /* mCutoffThreshold * levelCutoffRepeat (TR.cpp:63,74) */
    float cutoff_base;       /* mCutoffThreshold 20 (numRobust test, TR.cpp:386) */
    float scale_rot, scale_trans, scale_a, scale_b; /* 1, 0.5, 10, 1000 (TR.h:484-487) */
} cmlhip_tracker_params;

typedef struct {
    float  E;                /* TR.cpp:410 */
    int    numTermsInE;
    int    numSaturated;
    int    numRobust;
    int    numWarped;        /* survivors written to the warped buffer before padding (TR.cpp:372-384) */
    float  flow[3];          /* TR.cpp:412-414 */
    double H[64];            /* trackerContext->hessian, scaled (TR.cpp:474-484); valid if want_hessian */
    double b[8];             /* trackerContext->jacobian, scaled (TR.cpp:475,485-488) */
    float  H9[81];           /* raw Accumulator9::H before /n and scaling (ACC.h:1026-1045) */
} cmlhip_tracker_result;

/* reference point list of one level: n x {u, v, idepth, color} (TR.h:46-60) */
int cmlhip_tracker_set_reference(cmlhip_ctx* ctx, int level, const float* uvic, int n);
/* Build the per-level lists on the device from the active points (makeCoarseDepthL0).
 * pts: n x {Ku, Kv, new_idepth, weight} (doubles) already projected into the reference frame by
 * the host (TR.cpp:521-553 uses Frame/MapPoint accessors that stay on the host);
 * the splat, 2x2 sum, dilation, normalisation and compaction (TR.cpp:550-719) run on
 * the device against pyramid `ref_image_id` (gray levels). n_out[level] = list sizes. */
int cmlhip_tracker_make_coarse_depth(cmlhip_ctx* ctx, uint64_t ref_image_id, int levels,
                                     const double* pts, int n, int* n_out);
/* read a level's list back (tests): uvic_out has room for w*h*4 floats; returns n */
int cmlhip_tracker_get_reference(cmlhip_ctx* ctx, int level, float* uvic_out, int* n_out);

/* One computeResidual (+ computeHessian when want_hessian) at `level` of image
 * `new_image_id`.  R, t = refToNew (double, cast to float as TR.cpp:270-271);
 * K = {fx, fy, cx, cy} of that level (src/cml/map/InternalCalibration.h:116-127);
 * aff = mLastReferenceExposure.to(exposure) = {a, b} (TR.cpp:272); b0 = reference
 * exposure parameter b (TR.cpp:428). */
int cmlhip_tracker_eval(cmlhip_ctx* ctx, uint64_t new_image_id, int level,
                        const double R[9], const double t[3], const double K[4],
                        const double aff[2], double b0,
                        const cmlhip_tracker_params* prm, int want_hessian,
                        cmlhip_tracker_result* out);
/* DSOTracker::optimize (TR.cpp:15-246) resident on the device, batched over motion hypotheses: ONE launch runs the whole
 * coarse-to-fine Levenberg-Marquardt loop of every candidate refToNew (one workgroup each) against the reference lists set by
 * cmlhip_tracker_set_reference / _make_coarse_depth and the pyramid `new_image_id`; one readback returns every result.
 * This is what DSOTracker::trackWithMotionModel (DSOTracker.h:238-383) runs once per hypothesis, sequentially, through
 * 25-40 cmlhip_tracker_eval calls each.  The only coupling between the reference's tries — a try is abandoned when the rmse of a
 * level pass exceeds 1.5 x that of the best try so far (TR.cpp:183-189) — only shortens a try: the kernel records the rmse of
 * every level pass (pass_level / pass_rmse) and the caller applies the rule afterwards while replaying the winner selection of
 * DSOTracker.h:262-313 (cml_amd::DSOTracker::trackWithMotionModelBatched).
 * ref_exposure = {a, b, exposure time} of the reference, init_exposure = {a, b, exposure time} every try starts from.
 * A hypothesis is spread over G workgroups (G = 8, 4, 2 or 1: as many as keep the whole launch resident, asked of the device); the
 * sums of a level are formed in parts that depend on the level's size alone, in a fixed order, so a hypothesis gives the same bits
 * whatever batch it travels in.  CMLHIP_ERR_TIMEOUT: a workgroup never met its partners (the launch was not co-resident); results void. */
#define CMLHIP_TRACKER_MAX_STEPS 256
typedef struct { double R[9], t[3]; } cmlhip_tracker_hypothesis;
typedef struct {
    double R[9], t[3];                    /* refToNew after the last accepted step */
    double a, b;                          /* exposure parameters of the new frame */
    int    isCorrect, tooManySaturated;   /* TR.cpp:239-240 (the second literally carries haveGoodPoints) */
    float  E[5]; int numTermsInE[5], numSaturated[5], numRobust[5], iterations[5];
    double levelCutoffRepeat[5], relAff[2], covariance[6];
    float  flow[3];
    int    n_pass, pass_level[8]; double pass_rmse[8];   /* level passes in execution order (a level may repeat once, TR.cpp:192-195) */
    int    n_steps; unsigned char step_level[CMLHIP_TRACKER_MAX_STEPS], step_accept[CMLHIP_TRACKER_MAX_STEPS];   /* the trials, TR.cpp:163 */
    double eval_us, algebra_us;           /* where the workgroup's time went: residual / Hessian evaluations, lane-0 algebra between them */
} cmlhip_tracker_opt_result;
/* The reference leaves its loop over the motion hypotheses behind the first good try: `haveOneGood && achievedRes < lastCoarseRMSE * 1.5`
 * (DSOTracker.h:306-309) — with the usual constant-velocity guess in front, behind the FIRST.  rmse_bar > 0 (the caller's lastCoarseRMSE * 1.5)
 * lets the batches that follow do the same: when hypothesis 0 ends adopted (isCorrect, finite E/n of level 0) with E/n below the bar, the
 * other hypotheses give up at their next exchange and return n_steps = -1 (nothing else of such a result is meaningful).  The caller's replay
 * of the selection (DSOTracker.h:262-313) decides as before; should it ever ask for a result that was given up (a bar that differs by a
 * rounding), it runs the batch again with rmse_bar = 0.  rmse_bar <= 0 (default): every hypothesis runs to its end. */
int cmlhip_tracker_set_early_exit(cmlhip_ctx* ctx, double rmse_bar);
int cmlhip_tracker_optimize_batch(cmlhip_ctx* ctx, uint64_t new_image_id, int levels, const double K0[4] /* level-0 fx fy cx cy */,
                                  const double ref_exposure[3], const double init_exposure[3], const cmlhip_tracker_params* prm,
                                  int optimize_a, int optimize_b, double saturated_ratio_threshold,
                                  int n_hypotheses, const cmlhip_tracker_hypothesis* hypotheses, cmlhip_tracker_opt_result* results);
/* The same in two halves: _async enqueues the batch and returns; _wait blocks until it (and whatever the caller enqueued behind it with a completion
 * ticket of its own: cmlhip_tracer_trace_resident_tracked_async) has run, then hands the n_hyp results over.  One batch in flight per context. */
int cmlhip_tracker_optimize_batch_async(cmlhip_ctx* ctx, uint64_t new_image_id, int levels, const double K0[4], const double ref_exposure[3],
                                        const double init_exposure[3], const cmlhip_tracker_params* prm, int optimize_a, int optimize_b,
                                        double saturated_ratio_th, int n_hyp, const cmlhip_tracker_hypothesis* hyp);
int cmlhip_tracker_optimize_wait(cmlhip_ctx* ctx, cmlhip_tracker_opt_result* out /* n_hyp of the batch in flight */);
/* warped buffer readback (tests): SoA rows idepth,u,v,dx,dy,residual,weight,refcolor
 * (TR.h:98-135), each numWarped long, in reference-list order. */
int cmlhip_tracker_get_warped(cmlhip_ctx* ctx, float* out8xn, int capacity, int* n_out);

/* ---------------------------------------------------------------- bundle adjustment (a6-a15) */
typedef struct {
    double fx, fy, cx, cy;   /* level-0 pinhole cached by BA (BA.cpp:419-425) */
    int    w, h;
    float  huber;            /* 9    BA.h:243 */
    float  outlier_th_sum;   /* 2500 BA.h:244 */
    double scale_f, scale_c; /* 50, 50 BA.h:252-253 */
    int    optimize_a, optimize_b; /* BA.h:276-277 */
} cmlhip_ba_params;

typedef struct {
    uint64_t image_id;       /* target image of residuals into this frame */
    float    frame_energy_th;/* DSOFrame::frameEnergyTH (DSOFrame.h:35) */
    float    b0;             /* DSOFrame::getB0 = state_zero[7]*scaleB (DSOFrame.h:197-199) */
} cmlhip_ba_frame;

typedef struct {
    float  x, y;             /* Corner mX,mY (src/cml/types.h:1254) */
    double idepth;           /* MapPoint::getReferenceInverseDepth (MapObject.h:110) */
    float  idepth_zero;      /* DSOPoint::idepth_zero (DSOPoint.h:73) */
    float  prior;            /* DSOPoint::priorF (BA.cpp:1182) */
    float  colors[CMLHIP_PATTERN];   /* DSOPoint::colors (DSOPoint.h:54, DSOContext.h:87-91) */
    float  weights[CMLHIP_PATTERN];  /* DSOPoint::weights (BA.cpp:410) */
    int    host;             /* DSOFrame::id of the reference frame */
} cmlhip_ba_point;

typedef struct {
    int point;               /* index into points[] */
    int target;              /* DSOFrame::id of elements.frame */
    int state;               /* state_state      (DSOResidual.h:152) */
    int is_linearized;       /* DSOResidual.h:96 */
} cmlhip_ba_residual;

/* DSOFramePrecomputed for one (host,target) pair (DSOFrame.h:248-291), index host*N+target */
typedef struct {
    double R[9], t[3];       /* trialRefToTarget (current state)           */
    double R0[9], t0[3];     /* PRE_RTll_0, PRE_tTll_0 (evaluation point)  */
    double aff_a, aff_b;     /* exposureTransition = host.aff_g2l().to(target.aff_g2l()) */
} cmlhip_ba_pair;

typedef struct {
    double energy;           /* sum of returned energies (BA.cpp:1565,1608) */
    int    n_in, n_oob, n_outlier;   /* counts of state_NewState after the pass */
    float  new_frame_energy_th;      /* setNewFrameEnergyTH result for frame N-1 (BA.cpp:2419-2464) */
} cmlhip_ba_lin_result;

int cmlhip_ba_set_params(cmlhip_ctx* ctx, const cmlhip_ba_params* prm);
/* Upload the window (BA::run preamble, BA.cpp:753-779).  Builds the two index maps the
 * kernels use: residuals grouped by point (CSR) and by (host,target) pair with
 * pair index htIDX = host + target*N (BA.cpp:1677). */
int cmlhip_ba_upload_window(cmlhip_ctx* ctx, int N, const cmlhip_ba_frame* frames,
                            int P, const cmlhip_ba_point* points,
                            int R, const cmlhip_ba_residual* residuals);
/* ---- The window kept across keyframes.  The reference edits its window in place: BA::addPoints appends points and their residuals
 * (BA.cpp:382-415), BA::addNewFrame appends one residual per active point (:417-462), removePoint / removeFrame drop entries and renumber the
 * frames (DSOContext.h:94-111,154-174), and BA::run (:744-910) works on what is there.  These calls give a caller the same hand-over: the
 * library keeps the window (points, residuals {point, target, state, linearized}) between keyframes, the caller sends the EDITS, and
 * cmlhip_ba_window_commit makes the edited window the device window of the next run — index maps, device order and every derived table
 * exactly as cmlhip_ba_upload_window builds them from the same lists (that call IS reset + append + commit).
 *   numbering: points and residuals are numbered in append order; cmlhip_ba_window_compact drops the entries whose flag is 0 and renumbers
 *   the survivors by rank (a surviving residual must name a surviving point) — the caller renumbers its own lists the same way;
 *   cmlhip_ba_window_retire_frame is removeFrame's renumbering: frame ids above `frame` move down by one, entries that still name the frame
 *   get -1 and must be dropped by the next compact.
 *   commit: `frames` as for cmlhip_ba_upload_window (image ids, thresholds, b0 of the N frames as they are NOW); idepth / idepth_zero / prior
 *   (P each, or NULL = keep the values appended) refresh the per-point values a run changes; reset_states != 0 applies BA::run's resetOOB
 *   (:766-779): every residual gets state IN / not linearized except the n_lin listed ones, which keep the given state and stay LINEARIZED. */
int cmlhip_ba_window_reset(cmlhip_ctx* ctx);
int cmlhip_ba_window_append_points(cmlhip_ctx* ctx, int n, const cmlhip_ba_point* points);
int cmlhip_ba_window_append_residuals(cmlhip_ctx* ctx, int n, const cmlhip_ba_residual* residuals);
int cmlhip_ba_window_retire_frame(cmlhip_ctx* ctx, int frame);
int cmlhip_ba_window_compact(cmlhip_ctx* ctx, int n_points, const unsigned char* point_alive, int n_residuals, const unsigned char* residual_alive);
int cmlhip_ba_window_counts(cmlhip_ctx* ctx, int* P, int* R);      /* entries the library holds (committed or not) */
/* Owner token of the kept window: a number that changes with every cmlhip_ba_window_reset (cmlhip_ba_upload_window resets).  A caller that appends to a
 * window it handed over earlier compares the number it remembered: equal sizes alone do not say that the entries are still its own. */
int cmlhip_ba_window_generation(cmlhip_ctx* ctx, unsigned* generation);
int cmlhip_ba_window_commit(cmlhip_ctx* ctx, int N, const cmlhip_ba_frame* frames, const double* idepth, const float* idepth_zero,
                            const float* prior, int reset_states, int n_lin, const int* lin_residuals, const int* lin_states);
/* Upload scope: between begin and end the host-to-device copies of cmlhip_ba_set_params, cmlhip_ba_window_commit, cmlhip_ba_set_pairs,
 * cmlhip_ba_set_arithmetic, cmlhip_ba_set_resident_outputs, cmlhip_ba_set_resident_state / _prior / _indirect (M = 0) and cmlhip_ba_resident_convergence are collected and leave as ONE
 * packed block when the scope ends (the few kernels those calls launch behind their own copies run then, in call order) — the preamble of BA::run
 * (BA.cpp:744-802) as one transfer instead of four.  Any OTHER call on the context ends the scope first; ending a scope that is not open is a no-op. */
int cmlhip_upload_scope_begin(cmlhip_ctx* ctx);
int cmlhip_upload_scope_end(cmlhip_ctx* ctx);
/* sizes of the uploaded window (N frames, P points, R residuals) */
int cmlhip_ba_window_size(cmlhip_ctx* ctx, int* N, int* P, int* R);
/* per-iteration state: N*N pair transforms + frame thresholds (ba_update_state) */
int cmlhip_ba_set_pairs(cmlhip_ctx* ctx, const cmlhip_ba_pair* pairs /* N*N, host*N+target */);
int cmlhip_ba_set_frame_energy_th(cmlhip_ctx* ctx, const float* th /* N */);
/* DSOFrame::getB0 of every frame again (it follows state_zero: DSOFrame.h:197-199).  BA::run re-anchors the newest frame before its closing
 * linearizeAll(true) (setEvalPT, BA.cpp:885-894): the b0 handed over with the window is stale for residuals hosted by that frame afterwards. */
int cmlhip_ba_set_frame_b0(cmlhip_ctx* ctx, const float* b0 /* N */);
/* Arithmetic of the residual kernels of the device-resident loop (cmlhip_ba_iteration_async / _batch: ba_linearize_rs.hip for windows of
 * >= 36 864 residuals, ba_linearize_rs4.hip below).  CMLHIP_ARITH_EXACT (default): DSOBundleAdjustmentLinearizationContext::linearize statement
 * for statement, bit-identical to the reference's contraction-free reading (BA.cpp:62-316).  CMLHIP_ARITH_RELAXED (opt-in): the same formulas with
 * fused multiply-adds, one Newton step on the projection's reciprocal, and the photometric terms / pattern sums of a pixel (BA.cpp:214-271) in fp32
 * where the reference widens to double and rounds every partial sum back to float — what a -ffast-math Release build of the reference is allowed
 * to do.  Per-residual outputs then agree with the exact mode to ~1e-6 relative (tests/test_relaxed_arithmetic_gpu.py: bars 1e-4, classification
 * identical up to a reported count of residuals at a threshold).  The record kernel (cmlhip_ba_linearize, the closing pass of a run, the
 * marginalisation passes) and every other call stay exact in both modes. */
#define CMLHIP_ARITH_EXACT   0
#define CMLHIP_ARITH_RELAXED 1
int cmlhip_ba_set_arithmetic(cmlhip_ctx* ctx, int mode);
/* What the residual kernels of the device-resident loop (cmlhip_ba_iteration_async / _batch, the resident form of cmlhip_ba_linearize_apply and the
 * closing pass of cmlhip_ba_finish_run) STORE per residual besides the state the next pass reads.  CMLHIP_RESIDENT_OUTPUTS_FULL (default): everything
 * DSOBundleAdjustmentLinearizationContext::linearize leaves in the residual (BA.cpp:131 setCenterProjectedTo, :297-314 state_NewEnergy /
 * state_NewEnergyWithOutlier / the returned energy).  CMLHIP_RESIDENT_OUTPUTS_LEAN: centerProjectedTo and the returned energy are not stored and
 * state_NewEnergyWithOutlier only for residuals into the newest frame — the ones setNewFrameEnergyTH reads (BA.cpp:2419-2464); nothing BA::run,
 * the marginalisation calls or the getters of the host mirror consume is affected (cmlhip_ba_get_center_projected and the new_energy_wo array of
 * cmlhip_ba_get_states then hold the values of the last FULL or record-kernel pass).  The host mirror runs LEAN unless keepResidualEnergies is set:
 * 20 of 121 stored bytes per residual and five partial-line store instructions per wave less under the texel gather. */
#define CMLHIP_RESIDENT_OUTPUTS_FULL 0
#define CMLHIP_RESIDENT_OUTPUTS_LEAN 1
int cmlhip_ba_set_resident_outputs(cmlhip_ctx* ctx, int mode);
int cmlhip_ba_get_resident_outputs(cmlhip_ctx* ctx, int* mode);
int cmlhip_ba_set_idepth(cmlhip_ctx* ctx, const double* idepth /* P */, const float* idepth_zero /* P or NULL */);
int cmlhip_ba_get_idepth(cmlhip_ctx* ctx, double* idepth /* P */);

/* linearizeAll(false) minus the host bookkeeping: DSOBundleAdjustmentLinearizationContext::linearize
 * for every active (non-linearized) residual (BA.cpp:62-316,1551-1565) followed by
 * setNewFrameEnergyTH (BA.cpp:2419-2464; the new threshold is stored for frame N-1). */
int cmlhip_ba_linearize(cmlhip_ctx* ctx, cmlhip_ba_lin_result* out);
/* applyActiveRes(copyJacobians) (BA.cpp:2045-2093) */
int cmlhip_ba_apply(cmlhip_ctx* ctx, int copy_jacobians);
/* cmlhip_ba_linearize followed by cmlhip_ba_apply(ctx, 1) as ONE pass over the residuals (the preamble of BA::run, BA.cpp:785-790: linearizeAll(false),
 * then applyRes(r, true) of every residual with nothing in between).  Same results as the two calls. */
int cmlhip_ba_linearize_apply(cmlhip_ctx* ctx, cmlhip_ba_lin_result* out);   /* out == NULL: enqueue only (no host wait); the pass's tail (energy sum, new threshold) rides in
                                                                               * the next resident iteration's solve launch (or runs when a getter / cmlhip_ba_finish_run asks);
                                                                               * its energy comes back as cmlhip_ba_finish_run's `first`.  Follow it with an iteration, a getter
                                                                               * or cmlhip_ba_finish_run: another call would leave frameEnergyTH of the newest frame pending */
/* The tail of DSOBundleAdjustment::run in one call and ONE readback: linearizeAll(true) (BA.cpp:896 = linearize + applyRes(true),
 * :1551-1569) followed by everything the host writes back afterwards — residual states / energies (:1571-1640), the points'
 * inverse depths and the per-point accumulators (HdiF -> setInverseDepthHessian, :1889-1901).  pairs must be current.
 * Any output pointer may be NULL; point_acc is P x 14 as cmlhip_ba_get_point_acc returns it.
 * linearizeAll(true) also REMOVES every active residual that is not good afterwards (toRemove, :1595-1598,1624-1638): after the readback
 * the device retires them too — state OOB (absorbing, :68-72,2055-2059) — so that the residual loop of cmlhip_ba_relinearize_points
 * (tryMarginalize walks the point's REMAINING residuals, :2291) cannot revive a residual the host has already dropped. */
int cmlhip_ba_finish_keyframe(cmlhip_ctx* ctx, cmlhip_ba_lin_result* lin, int* state, int* new_state, float* energy,
                              float* new_energy, float* new_energy_without_outlier, unsigned char* is_good,
                              double* idepth, float* point_acc);

/* solveSystem accumulation (BA.cpp:1354-1385): addToHessianTop (ACTIVE and LINEARIZED),
 * stitchDoubleTop, addToHessianSC, stitchDoubleSC.
 * adHost/adTarget: N*N 8x8 row-major, index h + t*N (BA.cpp:1094-1095);
 * adHTdeltaF: N*N x 8 (BA.cpp:1113); cdelta: mCDeltaF (BA.cpp:1118);
 * prior: N x 8 DSOFrame::prior, delta_prior: N x 8 (BA.cpp:1169-1175); cprior: mCPrior.
 * Outputs are (8N+4)^2 / (8N+4) doubles, row-major; any may be NULL. */
typedef struct {
    const double* adHost; const double* adTarget; const float* adHTdeltaF;
    const double* cdelta;         /* 4 */
    const double* prior;          /* N*8 */
    const double* delta_prior;    /* N*8 */
    const double* cprior;         /* 4 */
} cmlhip_ba_accum_in;
int cmlhip_ba_accumulate(cmlhip_ctx* ctx, const cmlhip_ba_accum_in* in,
                         double* HA, double* bA, double* HL, double* bL,
                         double* Hsc, double* bsc);
/* solveLevenbergMarquardt (BA.cpp:1284-1320): H = HL+HM+HA, diag*(1+lambda),
 * -Hsc/(1+lambda), Jacobi scaling, factorisation of the trailing 8N block
 * (or the full system when optimize_calibration), x[0:4] = 0 otherwise.
 * HM/bM may be NULL (disableMarginalization, BA.cpp:1395-1398). x: 8N+4. */
int cmlhip_ba_solve(cmlhip_ctx* ctx, double lambda, const double* HM, const double* bM,
                    int optimize_calibration, double* x);
/* resubstitution (BA.cpp:1427-1487): x is the (possibly orthogonalised) solution;
 * step[p] = -HdiF * (bdSum - cstep.Hcd - sum xAd[h,t].JpJdF). Returns NONFINITE like
 * BA.cpp:1484-1491. */
int cmlhip_ba_backsub(cmlhip_ctx* ctx, const double* x, double* step /* P, may be NULL */);
/* point part of doStepFromBackup (BA.cpp:976-994): idepth = backup + step if finite and > 0;
 * also returns sumID, sumNID, numID (float accumulators of BA.cpp:988-990). */
int cmlhip_ba_backup_points(cmlhip_ctx* ctx);                      /* BA.cpp:919-922 */
int cmlhip_ba_step_points(cmlhip_ctx* ctx, float sums_out[3]);
/* point part of loadSateBackup (BA.cpp:938-942): idepth = idepth_zero = idepth_backup */
int cmlhip_ba_restore_points(cmlhip_ctx* ctx);

/* ---- readbacks (parity tests, Statistic norms BA.cpp:1415-1425, host bookkeeping) */
int cmlhip_ba_get_states(cmlhip_ctx* ctx, int* state, int* new_state, float* energy,
                         float* new_energy, float* new_energy_with_outlier,
                         unsigned char* is_good /* isActiveAndIsGoodNEW */);
/* raw Jacobian records in the reference layout (DSOResidual.h:22-69):
 * resF[8] Jpdxi[2][6] Jpdc[2][4] Jpdd[2] JIdx[2][8] JabF[2][8] JIdx2[4] JabJIdx[4] Jab2[4]
 * (2x2 blocks column-major as Eigen stores them). which: 0 = rJ (latest linearize), 1 = efsJ. */
int cmlhip_ba_get_rj(cmlhip_ctx* ctx, int which, float* out /* R*74 */);
int cmlhip_ba_get_jpjdf(cmlhip_ctx* ctx, float* out /* R*8 */);
int cmlhip_ba_get_center_projected(cmlhip_ctx* ctx, float* out /* R*3 */);
/* What the residual kernel reads besides points and residuals, as the device holds it NOW: the N*N DSOFramePrecomputed records
 * (DSOFrame.h:248-291; layout of cmlhip_ba_set_pairs — inside the resident loop they are rewritten by the device frame step every
 * iteration) and the per-frame frameEnergyTH / b0 (DSOFrame.h:35,197-199).  A plain copy with no side effect (the pending
 * setNewFrameEnergyTH of the last resident pass is NOT run): the values are exactly those the last residual pass used, which is what
 * lets a checker replay that pass (tests/resident_check.py, bench.py's parity gate).  Any pointer may be NULL. */
int cmlhip_ba_get_pairs(cmlhip_ctx* ctx, cmlhip_ba_pair* pairs /* N*N */, float* frame_energy_th /* N */, float* b0 /* N */);
/* per point: Hdd_accAF, bd_accAF, Hcd_accAF[4], Hdd_accLF, bd_accLF, Hcd_accLF[4], HdiF, bdSumF (14 floats) */
int cmlhip_ba_get_point_acc(cmlhip_ctx* ctx, float* out /* P*14 */);
/* raw per-pair 13x13 accumulators (AccumulatorApprox::H after finish, ACC.h:639-673), index h + t*N */
int cmlhip_ba_get_pair_acc(cmlhip_ctx* ctx, int mode, float* out /* N*N*169 */);
/* index maps built by upload_window (bit-exact bookkeeping):
 * pair_of[r] = host + target*N; by_point_offsets[P+1], by_point[R]; by_pair_offsets[N*N+1], by_pair[R] */
int cmlhip_ba_get_index_maps(cmlhip_ctx* ctx, int* pair_of, int* by_point_offsets, int* by_point,
                             int* by_pair_offsets, int* by_pair);

/* ---------------------------------------------------------------- hybrid ORB term (a16)
 * addIndirectToProblem (BA.cpp:2574-2729) + ReprojectionError::jacobian
 * (src/cml/optimization/Residual.h:59-100).  obs: n x {frame, point, gt_x, gt_y}
 * (gt = undistorted feature position, Residual.h:15); poses: N x {R[9], t[3]} world->cam;
 * points: M x 3 world coordinates; fxfy: level-0 focal lengths (Tukey threshold 3/|(fx,fy)|).
 * Outputs: M6 = pose block of J J^T (6N x 6N, before damping), b6 = 6N, used[n] = 1 when the
 * observation passed the res/finite tests (BA.cpp:2630). */
typedef struct { int frame; int point; double gx, gy; } cmlhip_reproj_obs;
int cmlhip_reproj_accumulate(cmlhip_ctx* ctx, int N, const double* poses /* N*12 */,
                             int M, const double* points /* M*3 */,
                             int n, const cmlhip_reproj_obs* obs, double fx, double fy,
                             double* M6 /* (6N)^2 */, double* b6 /* 6N */,
                             double* Jpoints /* M*3 or NULL */, unsigned char* used /* n or NULL */);
/* indirectX = ldlt(M6 with diag*(1+lambda)).solve(-b6) (BA.cpp:2695-2700) on the device */
int cmlhip_reproj_solve(cmlhip_ctx* ctx, int N, double lambda, double* x6 /* 6N */);

/* ---------------------------------------------------------------- immature points: DSOTracer
 * trace(): epipolar search of every immature point in a new frame (DSOTracer.cpp:585-823); optimizeImmaturePoint():
 * the per-point Gauss-Newton on activation (DSOTracer.cpp:280-404, linearizeResidual :406-494).  The pattern is star8. */
enum { CMLHIP_IPS_GOOD = 0, CMLHIP_IPS_OOB = 1, CMLHIP_IPS_OUTLIER = 2, CMLHIP_IPS_SKIPPED = 3, CMLHIP_IPS_BADCONDITION = 4,
       CMLHIP_IPS_UNINITIALIZED = 5 };                              /* DSOTracerStatus, DSOPoint.h:12-19 */
typedef struct {            /* DSOTracerPointPrivate (DSOTracer.h:17-33) + the MapPoint fields the tracer reads */
    float  x, y;            /* reference corner, level 0 */
    int    host;            /* index of the reference frame in the per-call frame list */
    int    last_status;     /* lastTraceStatus (in/out) */
    double idepth_min, idepth_max;                                   /* iDepthMin / iDepthMax (in/out; max may be NaN) */
    double gradH[4];        /* row-major 2x2 */
    double energy_th, quality;
    double last_uv[2], last_pixel_interval;                          /* lastTraceUV, lastTracePixelInterval (out) */
    float  gray[8];         /* MapPoint::getGrayPatch at the pattern pixels (MapObject.h:392-401) */
    float  dpatch[24];      /* MapPoint::getDerivativePatch (I, dI/dx, dI/dy) at the pattern pixels (:403-412) */
} cmlhip_immature_point;
typedef struct {            /* reference frame -> traced frame, formed as DSOTracer.cpp:608-610 */
    double KRKi[9], Kt[3];  /* K R K^-1 and K t of host -> frame (level 0) */
    double aff_a, aff_b;    /* referenceFrame.getExposure().to(frame.getExposure()) */
} cmlhip_trace_pair;
typedef struct {            /* DSOTracer.h:188-206 (values as Parameter stores them: float literals widened to double) */
    double max_pix_search, max_slack_interval, trace_step_size, min_improvement_factor, min_trace_test_radius,
           extra_slack_on_th, huber_th, outlier_th_sum_component, min_idepth_h_act;
    int    gn_its_on_activation, pad;
} cmlhip_tracer_params;
/* traceNewCoarse's per-point work: points[i] is traced in `image_id` with pairs[points[i].host]; every field trace() writes
 * is updated in place.  A point whose host IS the traced frame must not be passed (trace() returns early for it). */
int cmlhip_trace_points(cmlhip_ctx* ctx, uint64_t image_id, const cmlhip_tracer_params* prm, int n_hosts,
                        const cmlhip_trace_pair* pairs, int n, cmlhip_immature_point* points);
/* The same with the immature set resident on the device (one launch + a 24-byte readback per traced frame): upload once,
 * trace every point whose host is neither `skip_host` (the traced frame itself, DSOTracer.cpp:597-599: counted with its old
 * status) nor negative (not in the window: untouched, not counted); counts[s] = number of points with status s afterwards. */
int cmlhip_tracer_set_points(cmlhip_ctx* ctx, int n, const cmlhip_immature_point* points);
int cmlhip_tracer_trace_resident(cmlhip_ctx* ctx, uint64_t image_id, const cmlhip_tracer_params* prm, int n_hosts,
                                 const cmlhip_trace_pair* pairs, int skip_host, int counts[6]);
int cmlhip_tracer_get_points(cmlhip_ctx* ctx, int n, cmlhip_immature_point* points);
/* The resident set between keyframes, edited where it lies (DSOTracer's list: points leave — activated, dropped, their host frame marginalised,
 * DSOTracer.cpp:20-26,124-134,216-247 — and makeNewTraces appends, :496-541): the new set is the kept points in the given order (slot i <- old slot
 * keep[i], with host index hosts[i] in the caller's current frame list) followed by the n_new new records.  cmlhip_tracer_get_state returns the seven
 * fields trace() writes (56 bytes per point instead of the 232-byte record) — what activatePoints' candidate tests read (DSOTracer.cpp:128-178). */
typedef struct { double idepth_min, idepth_max, quality, last_uv[2], last_pixel_interval; int last_status, pad; } cmlhip_immature_state;
int cmlhip_tracer_edit_points(cmlhip_ctx* ctx, int n_keep, const int* keep, const int* hosts, int n_new, const cmlhip_immature_point* new_points);
int cmlhip_tracer_get_state(cmlhip_ctx* ctx, int n, cmlhip_immature_state* out);
/* One enqueue and ONE host wait for a tracked frame (Hybrid.cpp:383-442: trackWithMotionModel, then traceNewCoarse against the pose it found).
 * cmlhip_tracker_optimize_batch_async enqueues the hypothesis batch; cmlhip_tracer_trace_resident_tracked_async then enqueues, on the same stream,
 * the trace of the resident immature set against the pose of the batch's FIRST hypothesis — the try the reference's loop ends on whenever it is good
 * (DSOTracker.h:306-309): the pairs host -> frame (frame = refToNew o reference; K R K^-1, K t, exposure transfer with exposure times 1,
 * DSOTracer.cpp:606-608, Exposure.h:119-123) are formed on the device from that result, `hosts` (world -> camera pose and exposure of every window
 * frame) and `reference`.  cmlhip_tracker_optimize_wait returns when the whole chain has run.  The caller replays the selection of
 * DSOTracker.h:262-313 on the results: when it adopts the first try, cmlhip_tracer_trace_resident_finish(keep = 1) returns the status histogram and
 * the pairs that were used; otherwise keep = 0 restores every traced point (the kernel journals what it overwrites) and the caller traces again with
 * the pairs of the pose it did select (cmlhip_tracer_trace_resident). */
typedef struct { double R[9], t[3], a, b; } cmlhip_frame_pose;          /* world -> camera, exposure parameters */
int cmlhip_tracer_trace_resident_tracked_async(cmlhip_ctx* ctx, uint64_t image_id, const cmlhip_tracer_params* prm, int n_hosts,
                                               const cmlhip_frame_pose* hosts, const cmlhip_frame_pose* reference, const double K[4], int skip_host);
/* Optional, BEFORE cmlhip_tracker_optimize_batch_async: the window of the trace that will follow the batch (the very arguments the _tracked_async call
 * then passes).  The batch's launch carries it and the workgroup that ends the first hypothesis forms the pairs host -> frame right behind its result;
 * the trace behind it reads them instead of every wave deriving its own, and the publishing launch copies them (same function, same bits — only where
 * it runs changes).  One shot: consumed by the next batch; a _tracked_async call whose window differs from the prepared one works as without it.
 * Windows wider than 8 frames are accepted and ignored (their pairs come from a launch of their own, as before). */
int cmlhip_tracer_tracked_prepare(cmlhip_ctx* ctx, int n_hosts, const cmlhip_frame_pose* hosts, const cmlhip_frame_pose* reference, const double K[4]);
int cmlhip_tracer_trace_resident_finish(cmlhip_ctx* ctx, int keep, int counts[6], cmlhip_trace_pair* pairs_out /* n_hosts or NULL */);
typedef struct {            /* host -> target of the activation window: Camera::to and Exposure::to (DSOTracer.cpp:418-420) */
    double R[9], t[3], aff_a, aff_b;
} cmlhip_activation_pair;
/* optimizeImmaturePoint for n points over the N frames `image_ids` (level-0 pinhole K = fx,fy,cx,cy): pairs[h*N+t].
 * result[i] = 1 activate / 0 keep immature / -1 remove (the return value of the reference); idepth[i] is the optimised
 * inverse depth when result is 1; res_state[i*N + t] = final state_state of the residual into frame t (-1 for t == host). */
int cmlhip_optimize_immature_points(cmlhip_ctx* ctx, int N, const uint64_t* image_ids, const double K[4],
                                    const cmlhip_activation_pair* pairs, const cmlhip_tracer_params* prm, int min_obs,
                                    int n, const cmlhip_immature_point* points, int* result, float* idepth, int* res_state);
/* The same for points of the device-resident set (cmlhip_tracer_set_points / _edit_points), named by their slots: the records do not travel.  The
 * hosts the set carries must index the frame list `image_ids`. */
int cmlhip_optimize_immature_points_resident(cmlhip_ctx* ctx, int N, const uint64_t* image_ids, const double K[4],
                                             const cmlhip_activation_pair* pairs, const cmlhip_tracer_params* prm, int min_obs,
                                             int n, const int* slots, int* result, float* idepth, int* res_state);

/* ---------------------------------------------------------------- coarse initializer: DSOInitializer
 * calcResAndGS (DSOInitializer.cpp:451-750): the photometric residuals/Jacobians of every initializer point of one pyramid
 * level against the frame being tracked, the 9x9 Gauss-Newton system (Accumulator9), its Schur complement on the inverse
 * depths and the per-point JbBuffer the idepth step (doStep, :869-909) is formed from.  One launch per LM evaluation. */
typedef struct {            /* the fields of DSOInitializerPoint (DSOInitializer.h:11-58) calcResAndGS reads and writes */
    float p_pattern[8][3];  /* pPattern: homogeneous pixel of each pattern position in the reference (setFirst, :66) */
    float color[8];         /* reference gray at the pattern positions (:67) */
    float idepth_new, iR, outlier_th;
    float energy[2];
    int   is_good;          /* in/out: cleared for good when a pattern pixel leaves the image (:508) */
    /* written by the call */
    int   is_good_new;
    float energy_new[2], maxstep, last_hessian_new;
    float jb[10];           /* mJbBuffer_new[i]: only touched for points that were good on entry, as in the reference */
    float pad[3];
} cmlhip_init_point;       /* 224 bytes */
typedef struct {
    float RKi[9], t[3];     /* (refToNew.R * K^-1).cast<float>(), refToNew.t.cast<float>() at this level (:473-474) */
    float fx, fy, cx, cy;   /* K(lvl) as float (:457-460) */
    float aff_a, aff_b;     /* r2new_aff: exposure ratio, 0 (:476-479) */
    float huber, alpha_w, alpha_k, coupling_weight;                  /* mHuberThreshold, mAlphaW, mAlphaK, mCouplingWeight */
    float tlog[3];          /* SE3(camera).log().head<3>() as float (:738) */
    float pad;
    double t_sqnorm;        /* refToNew.getTranslation().squaredNorm() in scalar_t (:674) */
} cmlhip_init_params;
/* H_out / H_out_sc 8x8 row-major, b_out / b_out_sc 8, res = (E.A, alphaEnergy, E.num) as the reference returns them. */
int cmlhip_initializer_calc_res_and_gs(cmlhip_ctx* ctx, uint64_t image_id, int level, const cmlhip_init_params* prm,
                                       int n, cmlhip_init_point* points, float* H_out, float* b_out,
                                       float* H_out_sc, float* b_out_sc, float res[3]);

/* ---------------------------------------------------------------- ORB side: pose-only optimisation
 * IndirectCameraOptimizer::optimize (src/cml/optimization/g2o/IndirectCameraOptimizer.cpp:4-195 with g2o's Levenberg,
 * :197-382 with Gauss-Newton) and evaluateOutliers (:384-427): one free VertexSE3Expmap, fixed points,
 * EdgeSE3ProjectXYZ with a Huber kernel of delta sqrt(5.991), 4 rounds of 10 iterations with the observations re-classified
 * after each round and the kernel removed for the last one.  The whole optimisation is ONE launch of one workgroup. */
typedef struct {
    double X[3];            /* pMP->getWorldCoordinate().absolute() (:71) */
    double obs[2];          /* feature point of the frame (:66) */
    double inv_sigma2;      /* edge information (:87-88: 1 / descriptor distance; :281: 1 / scaleFactor^2) */
    double info;            /* vnInfo: the information evaluateOutliers tests with (:89, :285) */
} cmlhip_pnp_match;        /* 56 bytes */
enum { CMLHIP_PNP_LEVENBERG = 0, CMLHIP_PNP_GAUSS_NEWTON = 1 };
typedef struct {
    int    is_ok;           /* IndirectCameraOptimizerResult::isOk */
    int    rounds;          /* rounds completed before returning */
    int    n_bad;           /* outliers after the last evaluateOutliers */
    int    lm_iterations[4];/* solve() calls of each round */
    int    pad;
    double R[9], t[3];      /* result.camera (world -> camera); the pose reached when is_ok == 0 */
    double covariance[6];   /* diagonal of Hpp^-1 (:177-190) when asked for */
    double chi2[4];         /* active robust chi2 at the last linearisation of each round */
} cmlhip_pnp_result;
/* R, t: the pose every round starts from (`camera` when given, else frame->getCamera(), :132-135); K = fx, fy, cx, cy
 * of level 0; outliers (n bytes, in/out) as the reference's List<bool>; check_outliers = mCheckOutliers. */
int cmlhip_pnp_optimize(cmlhip_ctx* ctx, const double R[9], const double t[3], const double K[4], int n,
                        const cmlhip_pnp_match* matches, unsigned char* outliers, int algorithm, int check_outliers,
                        int compute_covariance, cmlhip_pnp_result* out);

/* ---------------------------------------------------------------- ORB side: local bundle adjustment
 * IndirectBundleAdjustment::localOptimize / startOptimization / the removal test of apply()
 * (src/cml/optimization/g2o/IndirectBundleAdjustment.cpp:7-236,:322-334) with the graph handed over as arrays.
 * The host keeps the covisibility search that selects local keyframes, fixed keyframes and local points (:9-35). */
typedef struct {
    double R[9], t[3];      /* frame->getCamera(): world -> camera (:72, :86) */
    double K[4];            /* fx, fy, cx, cy of frame->getK(0) (:150-154) */
    int    fixed;           /* 1: lFixedCameras (or every frame when fixFrames) */
    int    pad;
} cmlhip_lba_frame;        /* 136 bytes */
typedef struct {
    int    frame;           /* index into the frame array */
    int    pad;
    double obs[2];          /* corner.point0() (:139) */
    double inv_sigma2;      /* 1 / scaleFactor^2 (:141-143) */
} cmlhip_lba_edge;         /* 32 bytes */
typedef struct {
    int    ok;
    int    n_bad;           /* edges that fail apply()'s test chi2 > 5.991 || !isDepthPositive (:327) */
    int    iterations_done[2];   /* optimize() iterations of the first pass and of the refinement pass */
    double chi2[2];         /* active robust chi2 after each pass (Levenberg mode) */
} cmlhip_lba_result;
/* Edges are point-major, in the order :120-165 creates them: point p owns edges [point_offsets[p], point_offsets[p+1]).
 * fix_frames != 0 (mBaMode != BAINDIRECT, indirect/Mapping.cpp:89): g2o's StructureOnlySolver<3> — poses stay, every point is
 * refined on its own (one launch, a lane per point).  fix_frames == 0: g2o's Levenberg over BlockSolver_6_3 with the points
 * marginalised; frames with fixed == 0 are optimised (1..32 of them: the reduced pose system is factorised in the LDS of one
 * CU), the others only constrain the points.  points (n_points x 3) and the R, t of the free frames are updated in place;
 * edge_bad (one byte per edge) receives apply()'s removal test. */
/* pbStopFlag of localOptimize (IndirectBundleAdjustment.cpp:7, handed to g2o by setForceStopFlag :65-67): a byte the caller may set
 * from another thread while cmlhip_lba_optimize runs.  g2o tests it before every optimize() iteration (sparse_optimizer.cpp,
 * `!terminate()`); here it is tested before every Levenberg iteration and, in the structure-only mode (one launch per pass), before
 * each pass.  iterations_done reports what ran.  NULL (the default) = no flag.  The pointer must stay valid until it is reset. */
int cmlhip_lba_set_stop_flag(cmlhip_ctx* ctx, const unsigned char* flag);
int cmlhip_lba_optimize(cmlhip_ctx* ctx, int n_frames, cmlhip_lba_frame* frames, int n_points, double* points,
                        const int* point_offsets, const cmlhip_lba_edge* edges, int fix_frames, int num_iterations,
                        int refine_iterations, unsigned char* edge_bad, cmlhip_lba_result* out);

/* ---------------------------------------------------------------- marginalisation (once per keyframe), SURVEY §8 a15
 * tryMarginalize's residual loop (BA.cpp:2291-2304) for the points that are about to be marginalised: every residual of the
 * listed points is reset (resetOOB), re-linearised at the current state, committed (applyRes(true)) and, when good,
 * fixed (fixLinearization, BA.cpp:2210-2238: res_toZero = resF - J*delta, isLinearized = true).  `in` supplies adHTdeltaF /
 * cdelta; pairs must be current (cmlhip_ba_set_pairs). */
int cmlhip_ba_relinearize_points(cmlhip_ctx* ctx, const cmlhip_ba_accum_in* in, int n, const int* point_idx, int* n_good);
/* The same pass with what tryMarginalize's host loop then reads of it (BA.cpp:2296-2304) in the same readback: packed[r] = state | isActiveAndIsGoodNEW << 2 |
 * isLinearized << 3 | state_NewState << 4 of residual r and — where asked for — its three energies, caller's order: one wait where
 * cmlhip_ba_relinearize_points + cmlhip_ba_get_states + cmlhip_ba_get_res_to_zero were three. */
int cmlhip_ba_relinearize_points_packed(cmlhip_ctx* ctx, const cmlhip_ba_accum_in* in, int n, const int* point_idx, int* n_good, unsigned char* packed /* R */,
                                        float* energy /* R or NULL */, float* new_energy /* R or NULL */, float* new_energy_wo /* R or NULL */);
/* marginalizePointsF (BA.cpp:2466-2500): MARGINALIZED-mode accumulation of the listed points only.
 * M, Mb = stitchDoubleTop(usePrior = false); Msc, Mbsc = stitchDoubleSC with shiftPriorToZero = false.  The caller adds
 * 0.25 * (M - Msc) and 0.25 * (Mb - Mbsc) to the prior (BA.cpp:2502-2507).  (8N+4)^2 / (8N+4) doubles each. */
int cmlhip_ba_marginalize_points(cmlhip_ctx* ctx, const cmlhip_ba_accum_in* in, int n, const int* point_idx,
                                 double* M, double* Mb, double* Msc, double* Mbsc);
/* calcLEnergy (BA.cpp:2119-2208) without its forceAccept early-out: prior terms + the sum over the LINEARIZED good residuals */
int cmlhip_ba_lin_energy(cmlhip_ctx* ctx, const cmlhip_ba_accum_in* in, double* energy, int* num_linearized);
/* res_toZeroF (R x 8) and isLinearized (R) as the device holds them */
int cmlhip_ba_get_res_to_zero(cmlhip_ctx* ctx, float* res_to_zero, unsigned char* is_linearized);

/* ---------------------------------------------------------------- device-resident Gauss-Newton iterations
 * The loop body of BA::run (BA.cpp:804-880) under forceAccept / fixLambda without a host round trip per iteration:
 *   accumulate -> Schur + system -> solve (+ orthogonalize) -> back-substitution + doStepFromBackup of the points AND of the
 *   frames (state += -x, PRE_worldToCam = exp(state_scaled) * worldToCam_evalPT, DSOFramePrecomputed of the N^2 pairs,
 *   adHTdeltaF / delta_prior of computeDelta) -> linearizeAll + applyRes(true) + setNewFrameEnergyTH.
 * The host sets the frame states once (after BA::run's preamble), enqueues k iterations, and reads the states back. */
typedef struct {
    double eval_q[4], eval_t[3];   /* worldToCam_evalPT as Sophus stores it: unit quaternion (w,x,y,z) + translation (DSOFrame.h:88) */
    double state[10];              /* DSOFrame::state, unscaled (DSOFrame.h:110-124); [8],[9] unused by the BA */
    double state_zero[10];         /* DSOFrame::state_zero */
    double prior_zero[8];          /* DSOFrame::prior_zero */
    double ab_exposure;            /* exposure time of the frame (aff_g2l, DSOFrame.h:216-222) */
    int    fix_pose;               /* doStepFromBackup(fixCamera): pose part of the step dropped (BA.cpp:957-960) */
    int    pad;
} cmlhip_ba_frame_state;

/* everything an iteration reads besides the residuals: adjoints and priors (as cmlhip_ba_accumulate takes them), the frame
 * states, the state scales {translation, rotation, a, b} (BA.h:246-251) and, optionally, an orthonormal basis U (7 x (8N+4),
 * row-major, zero rows allowed) of the gauge nullspace: from the third iteration on x -= U^T (U x) (BA.cpp:1196-1261,1404). */
int cmlhip_ba_set_resident_state(cmlhip_ctx* ctx, const cmlhip_ba_accum_in* in, const cmlhip_ba_frame_state* frames,
                                 const double scales[4], const double* nullspace_basis /* 7*(8N+4) or NULL */);
/* The hybrid ORB term INSIDE the resident iteration (MODSLAM's mixedBundleAdjustment, BA.cpp:1327-1329 -> addIndirectToProblem
 * :2574-2729): the indirect points (M x world XYZ) and observations are kept on the device; every cmlhip_ba_iteration_async then
 * evaluates the reprojection Jacobians on the CURRENT resident frame poses, solves the per-frame 6x6 systems and replaces the pose
 * part of x by that solution (the reference's literal weighting, :2714-2727) before the nullspace projection — no host round trip.
 * Only windows with more than 4 frames mix (:1327).  Call after cmlhip_ba_set_resident_state; M = 0 switches the term off.
 * get: x of the last iteration (8N+4), the last indirect solution (6N) and the per-point Jacobian sums (3M, for setUncertainty :2690). */
int cmlhip_ba_set_resident_indirect(cmlhip_ctx* ctx, int M, const double* points_xyz, int n_obs, const cmlhip_reproj_obs* obs, double fx, double fy);
int cmlhip_ba_get_resident_indirect(cmlhip_ctx* ctx, double* x, double* x6, double* Jpoints);
/* The marginalisation prior INSIDE the resident iteration (solveSystem with disableMarginalization == false, BA.cpp:1389-1401: HM =
 * mMarginalizedHessian, bM_top = mMarginalizedB + mMarginalizedHessian * getFramesDelta()): HM ((8N+4)^2) and the RAW bM (8N+4,
 * mMarginalizedB itself) are kept on the device; every iteration adds HM to the system and forms bM_top from the resident frame states
 * — the frame step of iteration i writes the right-hand side iteration i + 1 takes.  Call after cmlhip_ba_set_resident_state (which
 * switches the prior off again); NULL pointers switch it off.  Valid under forceAccept, where calcMEnergy / calcLEnergy return 0
 * (BA.cpp:2100-2102, 2123-2125) and the prior therefore only enters the solve. */
int cmlhip_ba_set_resident_prior(cmlhip_ctx* ctx, const double* HM, const double* bM);
/* Mirror of BA::run's early exit (`if (canbreak && it >= 1) break`, BA.cpp:879, canbreak from doStepFromBackup :996-1027 with
 * thOptIterations): after the call, the iteration whose step passes the test is the last one that runs — the kernels of the
 * iterations enqueued behind it return at once.  th <= 0 switches the test off (every enqueued iteration runs). */
int cmlhip_ba_resident_convergence(cmlhip_ctx* ctx, double th_opt_iterations);
/* number of iterations that ran and the photometric energy after each of them (statEnergyP); synchronises */
int cmlhip_ba_get_resident_log(cmlhip_ctx* ctx, int* iterations, double* energies, int capacity);
/* frame states after the iterations enqueued so far (synchronises); pre_w2c: N x 7 (q, t) of PRE_worldToCam, may be NULL;
 * last_pass: energy / census / setNewFrameEnergyTH of the last residual pass (what cmlhip_ba_linearize returns), may be NULL */
int cmlhip_ba_get_resident_state(cmlhip_ctx* ctx, cmlhip_ba_frame_state* frames, double* pre_w2c, cmlhip_ba_lin_result* last_pass);

/* The tail of DSOBundleAdjustment::run (BA.cpp:882-910) when the loop ran resident, with ONE host wait: everything the three getters above return
 * (after the iterations enqueued so far), then — reanchor_newest != 0 — the re-anchoring of the newest frame's evaluation point ON THE DEVICE
 * (setEvalPT(PRE_worldToCam, (0,..,0,a,b)), :885-894: evaluation point = current pose, PRE_RTll_0 / PRE_tTll_0 of the pairs that name the frame,
 * b0) and the closing linearizeAll(true) with cmlhip_ba_finish_keyframe's outputs.  `first` = the ENERGY of the preamble pass enqueued by
 * cmlhip_ba_linearize_apply(ctx, NULL) (statEnergyP's first entry, BA.cpp:792; filled when cmlhip_ba_resident_convergence armed the control block,
 * 0 otherwise; its counts and threshold fields are 0: that pass's tail ran inside the first solve launch and only logged its energy); `last` = the
 * last iteration's pass; frames / pre_w2c are the states BEFORE the re-anchoring (the host mirror
 * re-anchors its own copy from pre_w2c[N-1], the pose the device used).  Any pointer may be NULL.  Replaces: the host round trip between
 * BA::run's loop and its closing pass (frame states up, DSOFramePrecomputed + b0 down). */
typedef struct {
    cmlhip_ba_frame_state* frames;      /* N */
    double* pre_w2c;                    /* N x 7 (q, t) of PRE_worldToCam */
    cmlhip_ba_lin_result* first;        /* preamble pass (cmlhip_ba_linearize_apply with out == NULL): `energy` only, see above */
    cmlhip_ba_lin_result* last;         /* last pass of the loop */
    int* iterations; double* energies; int capacity;      /* as cmlhip_ba_get_resident_log */
    double* x;                          /* 8N+4: x of the last solve */
    /* compact forms of the closing pass's outputs, packed on the device in the CALLER's order (no permutation on the host, a tenth of the bytes):
     * state_good[r] = state_state | isActiveAndIsGoodNEW << 2 (R bytes); hdi[p] = HdiF of point p (P floats: setInverseDepthHessian, BA.cpp:1889-1901) */
    unsigned char* state_good;
    float* hdi;
} cmlhip_ba_resident_out;
int cmlhip_ba_finish_run(cmlhip_ctx* ctx, int reanchor_newest, const cmlhip_ba_resident_out* resident, cmlhip_ba_lin_result* lin, int* state,
                         int* new_state, float* energy, float* new_energy, float* new_energy_without_outlier, unsigned char* is_good,
                         double* idepth, float* point_acc);

/* ---------------------------------------------------------------- timing helpers (bench.py)
 * HIP events on the context stream; ms between the two most recent marks. */
int cmlhip_event_mark(cmlhip_ctx* ctx, int which /* 0 = start, 1 = stop */);
int cmlhip_event_elapsed_ms(cmlhip_ctx* ctx, float* ms);
/* attach the two events to the NEXT instrumented dispatch itself (tracker evaluation, tracker optimisation batch, the kernels of the
 * resident iteration) instead of recording them around it: cmlhip_event_elapsed_ms then returns that kernel's own duration
 * (begin / end timestamps of the dispatch — the quantity rocprofv3 --kernel-trace reports), without launch overhead */
int cmlhip_profile_next_launch(cmlhip_ctx* ctx);
/* enqueue-only variants for throughput measurement: no host readback, no sync */
int cmlhip_ba_linearize_async(cmlhip_ctx* ctx);
/* one resident iteration (see above); without cmlhip_ba_set_resident_state only the points are stepped */
int cmlhip_ba_iteration_async(cmlhip_ctx* ctx, double lambda);
/* Several windows per launch (throughput mode: more sequence shards than GPUs, north_star "independent keyframe windows / sequence
 * shards"): ONE resident iteration of each of the S windows held by ctxs[0..S), in the five launches one window takes (gridDim.y =
 * window, S solve workgroups side by side) on the stream of ctxs[0].  Every window must be uploaded, have its resident state set
 * (cmlhip_ba_set_resident_state); the windows of a batch share one residual-kernel regime (all R < 36 k: the 4-lane kernel, or all
 * R >= 36 k: the lane-per-residual kernel) and one point-slice class, carry no hybrid term, no LINEARIZED residuals and no
 * convergence control, and fit the batched solve / back-substitution (reduced system resident in LDS; N < 12 or P < 2048) —
 * anything else is refused with CMLHIP_ERR_INVALID / _STATE, never routed elsewhere.  A window's
 * result is bit-identical to the one cmlhip_ba_iteration_async gives it (same kernel bodies, same arguments).  Synchronise through
 * cmlhip_synchronize(ctxs[0]) before reading any of the windows back or using their contexts on their own again. */
int cmlhip_ba_iteration_batch(cmlhip_ctx* const* ctxs, int n_windows, double lambda);
/* Per-kernel HIP-event timing of the iteration pipeline: when enabled, cmlhip_ba_iteration_async attaches HIP events to the
 * DISPATCHES themselves (hipExtLaunchKernelGGL start / stop events, i.e. the begin / end timestamps of the kernel, the same
 * quantity rocprofv3 --kernel-trace reports): (a) begin and end of the residual/Jacobian kernel, (b) begin of the
 * accumulate kernel and end of the back-substitution kernel (the Schur-reduce + solve group, launch gaps included).
 * read returns the mean durations in ms over the recorded iterations and resets the recorder. */
int cmlhip_profile_enable(cmlhip_ctx* ctx, int max_iterations);
/* record only every stride-th iteration (default 1), so that the event records do not perturb a timed run */
int cmlhip_profile_stride(cmlhip_ctx* ctx, int stride);
/* which of the two groups carry events: 1 = (a) the residual kernel, 2 = (b) the Schur-reduce + solve group, 3 = both (default).
 * An event-carrying dispatch costs the pipeline ~3 us (its completion is signalled to the host side of the queue), so a timed run
 * that needs only the roofline kernel's duration asks for (a) alone (the dispatch ahead of the residual kernel keeps its end event, so
 * that the kernel's begin timestamp is taken as rocprofv3 takes it); the group that is not selected reads back as 0. */
int cmlhip_profile_select(cmlhip_ctx* ctx, int mask);
/* development aid: in-kernel phase timestamps (wall clock, 10 ns ticks), 16 slots per kernel: [0,16) residual kernel,
 * [16,32) accumulate, [32,48) system tiles, [48,64) solve, [64,80) back-substitution; then, from slot 128, per-workgroup
 * {begin, end} pairs for the five kernels (1024 workgroups each).  `out` holds CMLHIP_DEBUG_SLOTS values.
 * Reads the previous values, then (re)arms. */
#define CMLHIP_DEBUG_SLOTS (128 + 5 * 1024 * 2)
int cmlhip_debug_timestamps(cmlhip_ctx* ctx, int enable, long long* out);
/* mean durations as defined above; empty_bracket_ms is kept for ABI stability and is always 0 (the events are the
 * dispatches' own timestamps: there is no bracket overhead to subtract) */
int cmlhip_profile_read(cmlhip_ctx* ctx, float* linearize_ms, float* schur_solve_ms, float* empty_bracket_ms, int* n_recorded);

#ifdef __cplusplus
}
#endif
#endif /* CMLHIP_H */
