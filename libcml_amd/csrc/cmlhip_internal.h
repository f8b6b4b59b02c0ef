// cmlhip_internal.h — shared declarations of the gfx950 device layer behind include/cmlhip.h.
// MI355X only: wave = 64 lanes, no portability shims.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_ext.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <map>
#include <unordered_map>
#include <vector>
#include <deque>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include "../../include/cmlhip.h"
#include "trace_pairs.h"

#define CML_WAVE 64
#define RS_TILE_DEFAULT 64

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    unsigned gen = 0;        // bumped by every (re)allocation: "is this the buffer I initialised" must not be an address compare (a free + malloc may return the address)
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct PyrLevel {
    int w = 0, h = 0;
    void* grad = nullptr;   // float4 {I,dx,dy,0} (or 4 halves) per texel, x fastest
    float* gray = nullptr;  // may be null when only the gradient image was put
    void* tiled = nullptr;  // fp16 level 0 only: the same texels in 128-byte tiles (cml_tiled_level0), built when a BA window first names the frame
    bool tiled_valid = false;
};
struct Pyramid {
    int levels = 0;
    PyrLevel lv[8];
    // cmlhip_pyramid_build_async: the levels are allocated, their contents arrive from the context's image worker on its own stream.
    // `ready` is recorded behind the last kernel; the first consumer (cml_find_pyr) waits for the worker to have enqueued everything and
    // orders the context's stream behind the event.
    bool pending = false;
    bool failed = false;          // the image worker could not copy / build it: the entry stays (its blocks are released by cmlhip_pyramid_drop) but is never handed out
    hipEvent_t ready = nullptr;
};
struct PyrJob { uint64_t id; const float* src; int levels; int w[8], h[8]; float* gray[8]; void* grad[8]; hipEvent_t ready; };

// per-frame device descriptor used by the BA kernels
// the flat records cross the ABI by value from other languages (ctypes / numpy dtypes in libcml_amd/abi.py): pin their sizes
static_assert(sizeof(cmlhip_immature_point) == 232, "cmlhip_immature_point layout");
static_assert(sizeof(cmlhip_init_point) == 224, "cmlhip_init_point layout");
static_assert(sizeof(cmlhip_pnp_match) == 56, "cmlhip_pnp_match layout");
static_assert(sizeof(cmlhip_lba_frame) == 136 && sizeof(cmlhip_lba_edge) == 32, "cmlhip_lba_* layout");
static_assert(sizeof(cmlhip_ba_pair) == 26 * sizeof(double), "cmlhip_ba_pair layout");

struct FrameDev {
    const void* grad0;      // level-0 gradient image of the frame
    float frame_energy_th;
    float b0;
    const void* grad0t;     // fp16 texels: the level-0 image again in 128-byte tiles (null otherwise), see cml_tiled_level0
};
// Tiled fp16 level 0 (read by the lane-per-residual kernel of the resident loop).  One 128-byte line = a tile of 4 rows x 4 texels
// plus, in every row, a copy of the first texel of the next tile (5 texels x 6 bytes {I, dI/dx, dI/dy} + 2 bytes pad = 32 bytes per
// row): the two texels of a bilinear row always lie in ONE tile row, and the 6 x 6 texel footprint of a pattern covers on average
// (1 + 5/4)^2 = 5.1 lines instead of the 7.9 of the row-major image (6 rows x (1 + 5/16)).
#define CML_TILE_W 4
#define CML_TILE_H 4
static inline size_t cml_tiled_bytes(int w, int h) { return (size_t)((w + CML_TILE_W - 1) / CML_TILE_W) * ((h + CML_TILE_H - 1) / CML_TILE_H) * 128; }

// the BA window as the library keeps it between keyframes (cmlhip_ba_window_*): SoA, caller order, the layout of the device's point arrays.
// The arrays live in ONE pinned, device-mapped host block sized by the limits given at create: the commit's scatter kernel reads them where they
// are (no staging copy); `busy` is recorded behind that kernel and waited for before the next edit touches the block.
struct WindowShadow {
    void* block = nullptr; size_t capP = 0, capR = 0;
    size_t P = 0, R = 0;
    float *x = nullptr, *y = nullptr, *idz = nullptr, *prior = nullptr, *colors = nullptr, *weights = nullptr;     // colors / weights: 8 per point
    double* idepth = nullptr;
    int* host = nullptr;
    int *rpoint = nullptr, *rtarget = nullptr, *rstate = nullptr;
    unsigned char* rlin = nullptr;
    hipEvent_t busy = nullptr; bool busy_pending = false;
};

struct cmlhip_ctx {
    cmlhip_limits lim{};
    int device = 0;
    unsigned attr_done = 0;       // hipFuncSetAttribute(MaxDynamicSharedMemorySize) already applied on this context's device, one bit per kernel
    hipStream_t stream = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    std::string err;
    std::unordered_map<uint64_t, Pyramid> pyr;
    // image worker (cmlhip_pyramid_build_async): one host thread per context that copies the caller's image
    // and builds the levels on pyr_stream — beside whatever the context's own stream is running (the reference builds a frame's pyramid on
    // its capture thread, ahead of the SLAM thread: capture/CaptureImage.cpp).  The worker touches neither the cache map nor the pool.
    std::thread pyr_thread; std::mutex pyr_mu; std::condition_variable pyr_cv;
    std::deque<PyrJob> pyr_jobs; bool pyr_quit = false, pyr_started = false;
    std::unordered_map<uint64_t, int> pyr_state;              // image id -> 1 with the worker, 2 enqueued on pyr_stream (ready recorded), -1 failed
    hipStream_t pyr_stream = nullptr;
    std::multimap<size_t, void*> img_pool;                    // released pyramid levels by byte size: a new frame reuses them (no hipMalloc / hipFree per frame)
    size_t img_pool_bytes = 0;
    DevBuf img_tmp;
    DevBuf h2d_blob, h2d_desc;                                // packed upload block and its segment table (batched cml_h2d)
    bool h2d_batching = false, h2d_inplace = false; size_t h2d_batch_start = 0;
    // cmlhip_upload_scope_begin / _end: the uploads of SEVERAL calls (window, pair records, resident state, prior) leave in one packed block; the kernels
    // those calls would launch behind their own upload wait in `deferred` and run, in order, when the scope ends
    bool h2d_scope = false; std::vector<std::function<int()>> deferred;
    bool d2h_batching = false; std::vector<unsigned long long> d2h_segs; std::vector<void*> d2h_dst;
    DevBuf d2h_blob; void* pinned_d2h = nullptr; size_t pinned_d2h_bytes = 0;
    std::vector<unsigned long long> h2d_segs;                 // (dst pointer, offset in the block, bytes) triples                                           // AoS3 staging of pyramid_put / pyramid_get
    void* pinned = nullptr;       // pinned host staging (readbacks / small uploads)
    size_t pinned_bytes = 0, pinned_off = 0;
    // Kernel timing of the resident iteration (cmlhip_profile_enable): the events ride ON the dispatches (hipExtLaunchKernelGGL
    // start / stop events = the dispatch's own begin / end timestamps, what rocprofv3 --kernel-trace reads), not around them
    hipEvent_t batch_ev = nullptr;                            // cml_iteration_batch: orders this context's pending stream work ahead of a batch on another context's stream
    hipEvent_t ext_start = nullptr, ext_stop = nullptr;       // consumed by the next CML_LAUNCH_EV
    hipEvent_t ext_stop_if_merged = nullptr;                  // stop event for the solve launch when the back-substitution rides in it
    std::vector<hipEvent_t> prof_ev;   // 4 events per recorded iteration: K3 begin, K6 end, K1 begin, K1 end
    int prof_cap = 0, prof_n = 0, prof_stride = 1, prof_tick = 0, prof_mask = 3;    // prof_mask: 1 = the residual kernel, 2 = the Schur-reduce + solve group

    // ---------------- BA window
    cmlhip_ba_params ba_prm{};
    bool ba_prm_set = false, ba_uploaded = false, ba_pairs_set = false;
    std::vector<uint64_t> ba_image_ids;                      // level-0 images the uploaded window points into (see window_forget_image)
    int N = 0, P = 0, R = 0, n_lin = 0, n_newframe = 0;
    std::vector<int> h_pair_of, h_by_point_off, h_by_point, h_by_pair_off, h_by_pair;    // caller numbering (cmlhip_ba_get_index_maps)
    std::vector<int> h_dev_of, h_caller_of;                   // caller r -> device r' (pair-sorted) and back
    DevBuf c_point, c_target, c_state, c_lin, c_dev_of, c_bpos;   // the window's residual lists in CALLER order + their device / by-point positions (input of k_window_expand)
    bool h_maps_valid = false;
    WindowShadow win; std::vector<int> h_tiles, h_tile_off, w_cnt_p, w_cnt_q;     // window kept across keyframes + commit scratch (allocated once)
    DevBuf frames, pairs;                                     // FrameDev[N], cmlhip_ba_pair[N*N]
    DevBuf pt_x, pt_y, pt_idepth, pt_idepth_zero, pt_prior, pt_host, pt_colors, pt_weights, pt_backup;
    DevBuf pt_acc;                                            // P x 16 floats: HddA bdA HcdA[4] HddL bdL HcdL[4] HdiF bdSum pad pad
    DevBuf pt_step;                                           // P doubles
    DevBuf r_dead;                                            // residuals the closing linearizeAll(true) of a run removed (BA.cpp:1595-1598,1624-1638): tryMarginalize's pass must not revive them
    DevBuf r_host;                                            // static copy of the point's host per residual (saves a dependent load)
    DevBuf r_point, r_target, r_state, r_new_state, r_energy, r_new_energy, r_new_energy_wo, r_ret_energy;
    DevBuf r_good, r_lin, r_sel, r_center, r_jpjdf, r_rtz, rj[2];
    DevBuf by_point_off, by_point, by_pair_off, by_pair, newframe_res;
    DevBuf point_code, point_tgt, point_pos; int pt_stride = 0; // [P][pt_stride]: efsJ code per point slot (kept by applyRes), static target | lin << 8, slot of r
    DevBuf pair_code, pair_pos; int pair_stride = 0;          // [N*N][pair_stride] efsJ code (2r+sel, -1 = not in the ACTIVE sum) kept by applyRes; position of r
    // resident residual kernel (ba_linearize_rs.hip)
    DevBuf rs_tiles, rs_tile_off, rs_part, r_px, r_py, r_colors, r_weights, r_idepth, point_res; int n_tiles = 0;
    std::vector<int> h_rs_pair_tab; int rs_pair_n = 0, rs_pair_max_tiles = 0;    // 4-lane kernel, 2-D launch: {first residual, residuals, first tile, host | target << 16} per pair that has residuals
    int rs_tile = RS_TILE_DEFAULT;                            // residuals per wave tile of this window: 64 = lane per residual (large windows), 16 = 4 lanes per residual
    bool r_idepth_dirty = true;                               // pt_idepth was written by something else than the resident point step
    bool efs_in_partials = false;                             // the last residual pass was the resident kernel: the pair blocks of the good
                                                              // residuals live in rs_part, their efsJ records are NOT materialised
    bool rs_ok = true;                                        // use the resident kernel in cmlhip_ba_iteration_async (CMLHIP_NO_RS=1 in the environment: the record-writing kernel)
    int lin_partial_n = 0;                                    // number of energy partials the last residual pass wrote
    DevBuf acc_pair[2];                                       // N*N x 96 floats (91 used) ACTIVE / LINEARIZED
    DevBuf acc_num[2];                                        // N*N ints
    DevBuf pair_blocks;                                       // N*N x PAIR_BLK doubles (stitched per-pair blocks)
    DevBuf adH, adT, adHTd, vec_small;                        // adjoints, adHTdeltaF, {cdelta,cprior,prior,delta_prior}
    DevBuf HA, bA, HL, bL, Hsc, bsc, HM, bM, xvec, Hf, bf;    // (8N+4)^2 / (8N+4) doubles; Hf/bf = final LM system
    double trk_early_rmse = 0.0;                              // cmlhip_tracker_set_early_exit: > 0: hypothesis 0 may end the batch (tracker_opt.hip)
    bool arith_relaxed = false;                               // cmlhip_ba_set_arithmetic(CMLHIP_ARITH_RELAXED): see ba_linearize_rs_body.inc
    unsigned win_generation = 0;                              // bumped by every cmlhip_ba_window_reset (and so by cmlhip_ba_upload_window): the owner token of the kept window
    int device_share = 1;                                     // cmlhip_set_device_share: contexts that launch on this device at the same time (sequence shards per GPU)
    bool rs_lean = false;                                     // cmlhip_ba_set_resident_outputs(CMLHIP_RESIDENT_OUTPUTS_LEAN): RsArgs::lean of the resident residual kernels
    DevBuf bM_raw; bool resident_prior = false;               // resident loop with the marginalisation prior: mMarginalizedB as handed over; bM then holds bM_raw + HM * delta of the CURRENT frame states (cmlhip_ba_set_resident_prior)
    DevBuf tr_points, tr_pairs, tr_out;
    DevBuf ini_points, ini_partial;          // coarse initializer (initializer.hip)
    DevBuf pnp_matches, pnp_flags, pnp_out;  // pose-only optimisation (pnp.hip)
    DevBuf lba_frames, lba_cams, lba_points, lba_off, lba_edges, lba_err, lba_flags, lba_work;   // local bundle adjustment (lba.hip)
    const volatile unsigned char* lba_stop = nullptr;         // the caller's pbStopFlag (cmlhip_lba_set_stop_flag): g2o's forceStopFlag
    DevBuf tr_resident; int tr_resident_n = 0;                // immature set kept on the device (cmlhip_tracer_set_points)                       // immature-point tracer staging
    DevBuf trk_pose0;                                         // {R, t, a, b} of the pending batch's first result, on the device
    DevBuf tr_resident2, tr_edit, tr_state;                   // cmlhip_tracer_edit_points (the set rebuilt into the second buffer), cmlhip_tracer_get_state
    DevBuf tr_hosts, tr_journal; void* tr_host = nullptr; void* tr_host_dev = nullptr;      // cmlhip_tracer_trace_resident_tracked_async: host poses, the rollback journal, mapped block {counts | pairs}
    bool tr_spec_pending = false; int tr_spec_hosts = 0, tr_spec_skip = -2;
    TrackedReq tr_req; bool tr_req_valid = false, tr_req_consumed = false;      // cmlhip_tracer_tracked_prepare: the window the next tracker launch carries (its tail forms the pairs)
    DevBuf tr_counts;                                         // status histogram of the tracked trace (cleared by its publishing kernel)
    DevBuf pt_mask, marg_scratch;                             // marginalisation passes: per-point selection, block partials
    DevBuf frame_state, pre_w2c, null_basis;                  // device-resident iterations (cmlhip_ba_set_resident_state)
    DevBuf rr_scratch;                                        // R-length readbacks permuted back to the caller's order on the device (ResRead, ba_api.hip)
    DevBuf run_pack;                                          // cmlhip_ba_finish_run: state | good << 2 per caller residual, HdiF per point (k_ba_pack_closing)
    DevBuf run_snap;                                          // cmlhip_ba_finish_run: the loop's last summary + frame states, kept across the re-anchoring and the closing pass
    bool resident_on = false, have_null = false, lin_finish_pending = false; int resident_iter = 0; double res_scales[4] = {1, 1, 1, 1}; double conv_th = 0; bool conv_on = false;
    DevBuf dbg; bool dbg_on = false;                          // phase timestamps (tools)
    std::vector<std::pair<const char*, hipEvent_t>> marks; size_t marks_n = 0;   // development (CMLHIP_RUN_MARKS=1): events behind the stages of a run(), printed by cmlhip_ba_finish_run
    DevBuf step_partial;                                      // per-block {sumID, sumNID, numID, pad} of the point update
    int n_lin_partial = 0; double last_lambda = 1e-5; bool last_have_hm = false; double sys_lambda = 1e-5;
    DevBuf G;                                                 // P x ldg doubles (Schur rows [g | bdSum])
    DevBuf syrk_part;                                         // partial SYRK tiles
    DevBuf xad;                                               // wide windows: x . adjoints table of the back-substitution (k_ba_xad)
    DevBuf solve_image;                                       // wide windows: the assembled, scaled LM system in the solver's LDS layout
    DevBuf scal;                                              // small scalar scratch (energy partials, counters, th)
    DevBuf lin_partial;                                       // per-block energy / count partials
    // ---------------- tracker
    DevBuf trk_ref[8]; int trk_n[8] = {0}; int trk_last_n = 0;                    // per-level uvic lists
    DevBuf trk_warped;                                        // n x 8 floats + flag
    DevBuf trk_partial, trk_out, trk_hyp, trk_opt_out;        // (trk_hyp / trk_opt_out: cmlhip_tracker_optimize_batch)
    float* trk_host = nullptr; unsigned trk_seq = 0;           // mapped, coherent host buffer: the tracker kernel writes its rows + a per-workgroup
                                                              // sequence flag straight to host memory, the caller polls (no memcpy, no stream sync)
    DevBuf cd_idepth[8], cd_wsum[8], cd_wbak[8], cd_cnt, cd_pts;      // makeCoarseDepth scratch
    // ---------------- reproj
    DevBuf rp_obs, rp_poses, rp_points, rp_M, rp_b, rp_Jp, rp_used, rp_x, rp_off, rp_orig; int rp_acc_N = 0;
    DevBuf trk_early;                                         // cmlhip_tracker_set_early_exit: the word hypothesis 0 raises (own buffer, cleared per armed launch)
    DevBuf trk_xch;                                           // cmlhip_tracker_optimize_batch: partial sums + tickets of the workgroups of a hypothesis
    void* trk_opt_host = nullptr; size_t trk_opt_host_bytes = 0;   // mapped, coherent host block of cmlhip_tracker_optimize_batch: hypotheses in, results out
    void* trk_opt_host_dev = nullptr;                         // its device address
    int trk_pending_n = 0; size_t trk_pending_res_off = 0, trk_pending_late_off = 0;   // cmlhip_tracker_optimize_batch_async: a batch is in flight (results at res_off of the block)
    void* done_word = nullptr; void* done_word_dev = nullptr; unsigned done_ticket = 0; bool done_pending = false;   // completion tickets (cml_done_enqueue / cml_done_wait, tracker_opt.hip)
    unsigned trk_xch_gen = 0; int trk_epoch = 0;              // tracker exchange buffer: cleared once per ALLOCATION (DevBuf::gen) and when the 16-bit launch number wraps (tracker_opt.hip)
    int trk_capacity[2] = {0, 0};                              // workgroups of k_tracker_optimize<half> the device holds at once (CUs x occupancy), 0 = not asked yet
    DevBuf x_ticket; bool x_ticket_zeroed = false, backsub_merged = false; int x_ticket_seq = 0;      // K6 inside the K5 launch (BacksubCall)
    DevBuf batch_main, batch_rs; std::vector<unsigned char> batch_main_host, batch_rs_host; unsigned attr_done_batch = 0;   // cmlhip_ba_iteration_batch (kept by the first context of the batch)
    DevBuf rr_obs, rr_off, rr_orig, rr_points, rr_jp, rr_used, rr_x, rr_ready; std::vector<int> rr_point_of;     // the resident hybrid term's own buffers
    bool rp_resident = false; int rp_res_M = 0, rp_res_n = 0; double rp_res_fx = 0, rp_res_fy = 0;   // hybrid term inside the resident iteration (cmlhip_ba_set_resident_indirect)
};

// every extern "C" entry selects its context's device first: the current device is per-thread state and a process may own
// contexts on several GPUs
void cml_mark(cmlhip_ctx* c, const char* what);               // development: no-op unless CMLHIP_RUN_MARKS is set
void cml_marks_dump(cmlhip_ctx* c);
void cml_scope_abort(cmlhip_ctx* c);                          // a staging call failed inside an open scope: drop the scope's block and deferred launches
int cml_scope_end(cmlhip_ctx* c);                              // flush the open upload scope (if any) and run what was deferred
// (a call that is not part of an open upload scope ends it first; when the scope's packed flush or one of its deferred kernels fails, THIS call reports
//  that status instead of proceeding on a half-built device window)
#define CML_DEV(ctx) do { if (ctx) { (void)hipSetDevice((ctx)->device); if ((ctx)->h2d_scope) { const int rc_scope_ = cml_scope_end(ctx); \
        if (rc_scope_) { if ((ctx)->err.empty()) (ctx)->err = "an upload scope that this call ended failed to flush"; return rc_scope_; } } } } while (0)
#define CML_DEV_SCOPED(ctx) do { if (ctx) (void)hipSetDevice((ctx)->device); } while (0)      /* entries that stage into an open upload scope */

#define CML_CHECK(ctx, call)                                                                      \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                      \
            return CMLHIP_ERR_HIP;                                                                \
        }                                                                                         \
    } while (0)

#define CML_REQUIRE(ctx, cond, code, msg)                                                         \
    do {                                                                                          \
        if (!(cond)) { (ctx)->err = (msg); return (code); }                                       \
    } while (0)

int cml_ensure(cmlhip_ctx* c, DevBuf& b, size_t bytes);          // grow-only device allocation
void cml_free(DevBuf& b);
int cml_h2d(cmlhip_ctx* c, void* dst, const void* src, size_t bytes);
int cml_h2d_inplace(cmlhip_ctx* c, void* dst, const void* src, size_t bytes);   // batch mode: a segment read where it lies (pinned, device-mapped source)
void cml_window_free(cmlhip_ctx* c);
void* cml_h2d_stage(cmlhip_ctx* c, void* dst, size_t bytes);      // batch mode: the staging bytes themselves (nullptr: use cml_h2d)
// Batched form for the many small arrays of a window upload: between begin and flush every cml_h2d only stages its bytes;
// flush moves the packed block with ONE copy and scatters it to the destinations with one small kernel.
void cml_h2d_batch_begin(cmlhip_ctx* c);
int cml_zero(cmlhip_ctx* c, void* dst, size_t bytes);        // hipMemsetAsync(0), or a zero segment of the open batch
int cml_fill_ff(cmlhip_ctx* c, void* dst, size_t bytes);      // 0xff fill (ints: -1), batched like cml_zero
int cml_fill_7f(cmlhip_ctx* c, void* dst, size_t bytes);      // 0x7f fill, batched like cml_zero
int cml_h2d_batch_flush(cmlhip_ctx* c);   // async on ctx stream via pinned staging
int cml_d2h(cmlhip_ctx* c, void* dst, const void* src, size_t bytes);   // sync readback (inside a batch: recorded, delivered by the flush)
// Several arrays, one round trip: between begin and flush cml_d2h only records; flush gathers the pieces into one device block,
// copies it once into pinned memory, waits once and hands the pieces out.
void cml_d2h_batch_begin(cmlhip_ctx* c);
int cml_d2h_batch_flush(cmlhip_ctx* c);
const Pyramid* cml_find_pyr(cmlhip_ctx* c, uint64_t id);
int cml_tiled_level0(cmlhip_ctx* c, uint64_t id, const void** out);      // builds (once) and returns the tiled fp16 level 0 of a cached pyramid
int cml_done_enqueue(cmlhip_ctx* c);                       // a ticket behind everything enqueued on the context's stream so far ...
int cml_done_embed(cmlhip_ctx* c, unsigned* ticket, volatile unsigned** word_dev);      // ... or written by a kernel of the caller's own as its last act
int cml_done_wait(cmlhip_ctx* c);                          // ... and the host's wait for the newest one (spin on a mapped word; the stream when none is pending)
const cmlhip_tracker_opt_result* cml_tracker_pending_result_dev(cmlhip_ctx* c, int i);

static inline int cml_div_up(int a, int b) { return (a + b - 1) / b; }

// launch `kern<<<grid, block, shmem, c->stream>>>(args...)`; when the context carries pending profile events they are attached to
// this dispatch and cleared
#define CML_LAUNCH_EV(c, kern, grid, block, shmem, ...)                                                              \
    do {                                                                                                            \
        if ((c)->ext_start || (c)->ext_stop) {                                                                      \
            hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), (std::uint32_t)(shmem), (c)->stream, (c)->ext_start, (c)->ext_stop, 0, __VA_ARGS__); \
            (c)->ext_start = nullptr; (c)->ext_stop = nullptr;                                                      \
        } else {                                                                                                    \
            kern<<<grid, block, shmem, (c)->stream>>>(__VA_ARGS__);                                                 \
        }                                                                                                           \
    } while (0)
