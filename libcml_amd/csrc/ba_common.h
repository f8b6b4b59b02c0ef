// ba_common.h — kernel argument block and record layout shared by the BA kernels.
#pragma once
#include "cmlhip_internal.h"

// 74-float DSORawResidualJacobian (DSOResidual.h:44-68) + 6 extra floats, padded to 80 (320 B, 16-B aligned)
enum { O_RES = 0, O_XI0 = 8, O_XI1 = 14, O_C0 = 20, O_C1 = 24, O_DD = 28, O_JI0 = 30, O_JI1 = 38,
       O_JAB0 = 46, O_JAB1 = 54, O_JI2 = 62, O_JABJI = 66, O_JAB2 = 70,
       O_X_JIR = 74,    // JI^T r   (2)  ACTIVE-mode inner products of BA.cpp:1719-1729, produced by linearize
       O_X_JABR = 76,   // Jab^T r  (2)
       O_X_RR = 78,     // r^T r    (1) + pad
       RJ_STRIDE = 80 };

// per-residual reduced record kept by applyRes (r_jpjdf, PS_STRIDE floats): JpJdF[8] (BA.cpp:2066-2080) and this residual's terms
// of the point sums of BA.cpp:1747-1750: Hcd[4] at PS_HCD, Hdd at PS_HDD, bd (ACTIVE mode) at PS_BD; what the point rows of
// k_ba_acc and k_ba_backsub read instead of the 320-B record
enum { PS_HCD = 8, PS_HDD = 12, PS_BD = 13, PS_STRIDE = 16 };

#define ACC_STRIDE 96          // floats per (host,target) accumulator: 55 (10x10 upper) + 30 (10x3) + 6 (3x3 upper) = 91
#define PT_ACC_STRIDE 16       // HddA bdA HcdA[4] HddL bdL HcdL[4] HdiF bdSum pad pad
// per-pair stitched fp64 blocks (stitchDoubleTop, BA.cpp:1827-1843): HH TT HT (64 each) HC TC (32 each) bH bT (8 each) CC (16) bC (4)
enum { PB_HH = 0, PB_TT = 64, PB_HT = 128, PB_HC = 192, PB_TC = 224, PB_BH = 256, PB_BT = 264, PB_CC = 272, PB_BC = 288, PB_STRIDE = 296 };

#define DBG_T(A, slot) do { if ((A).dbg && threadIdx.x == 0 && blockIdx.x == 0) (A).dbg[slot] = wall_clock64(); } while (0)

// per-workgroup {begin,end} stamps: kernel k in 0..4 (linearize, acc, system, solve, backsub)
#define DBG_BLK(dbgp, k, which) do { if ((dbgp) && threadIdx.x == 0 && blockIdx.x < 1024) (dbgp)[128 + ((k) * 1024 + blockIdx.x) * 2 + (which)] = wall_clock64(); } while (0)
#define DBG_BLK_END(dbgp, k) do { if (dbgp) { __syncthreads(); DBG_BLK(dbgp, k, 1); } } while (0)

struct LinSummary {
    double energy;
    int n_in, n_oob, n_outlier;
    float new_frame_energy_th;
    float sums[4];            // step_points: sumID sumNID numID, pad
    int nonfinite;            // backsub: #points with a non-finite step
    int pad;
};

// control block of the device-resident loop when it mirrors BA::run's early exit (BA.cpp:879 `if (canbreak && it >= 1) break`):
// lives in the context's scalar scratch; every kernel of an iteration returns at once when `stop` is set
struct ResidentCtl {
    int stop;                 // sticky
    int iters_done;           // iterations that really ran
    float frame_sums[4];      // sumA sumB sumT sumR of the last frame step (doStepFromBackup, BA.cpp:957-972)
    int stop_lin;             // `stop` as the residual kernel sees it: raised by the first launch AFTER the one that set `stop`, so that
                              // the residual pass which detects convergence still runs to completion in every workgroup
    int pad;
    double energy[40];        // photometric energy after iteration i (statEnergyP)
    double energy0;           // ... and of the pass AHEAD of the first iteration (run()'s preamble), when its tail rides in the first solve launch
};
#define CML_CTL_OFFSET 640
#define CML_CLOSE_OFFSET 192            // LinSummary of run()'s closing pass (cmlhip_ba_finish_run): the loop's last summary stays at offset 0
#define CML_ZERO_WORD_OFFSET 1008       // a word of the scalar scratch that is cleared with it at upload and never written

struct BAArgs {
    int N, P, R, w, h, opt_a, opt_b, n;            // n = 8N+4
    double fx, fy, cx, cy, fxi, fyi, huber_d, oth_d, scale_f, scale_c;
    const FrameDev* frames;
    const cmlhip_ba_pair* pairs;
    const float* pt_x; const float* pt_y; double* pt_idepth; float* pt_idepth_zero; const float* pt_prior;
    const int* pt_host; const float* pt_colors; const float* pt_weights; float* pt_backup; float* pt_acc; double* pt_step;
    const int* r_point; const int* r_host; const int* r_target; int* r_state; int* r_new_state;
    float* r_energy; float* r_new_energy; float* r_new_energy_wo; float* r_ret_energy;
    unsigned char* r_good; const unsigned char* r_lin; unsigned char* r_sel;
    unsigned char* r_lin_rw; int* point_tgt_rw;    // writable views for the marginalisation kernels (isLinearized changes there)
    unsigned char* r_dead;                         // 1: removed by the closing pass of the last run (absorbing, like OOB); never null after an upload
    float* r_center; float* r_jpjdf; float* r_rtz; float* rj0; float* rj1;
    const int* by_point_off; const int* by_point; const int* by_pair_off; const int* by_pair;
    double* r_idepth; const int* point_res;         // per-residual copy of the point's inverse depth; [P][pt_stride] residual of the slot (-1 = empty)
    int* point_code; const int* point_tgt; const int* point_pos; int pt_stride;   // [P][pt_stride]: 2r+sel of the point's good residuals else -1; target | lin << 8 (-1 = empty slot); slot of r
    int* pair_code; const int* pair_pos; int pair_stride;     // [N*N][pair_stride]: 2r+sel of the ACTIVE good residuals of the pair, else -1 (written by applyRes); slot of r
    double* lin_partial;          // per-block {energy, n_in, n_oob, n_outlier} of the residual kernel (may be null)
    long long* dbg;               // optional phase timestamps (wall_clock64, 100 MHz): 16 slots per kernel, see cmlhip_debug_read
    ResidentCtl* ctl;             // null: no convergence test / early exit (bench, tests of single iterations)
    int it_index; int n_step_blocks; double th_opt;   // iteration number, #blocks of step_partial, thOptIterations (BA.h:244)
    const float* step_partial_ro;
    const unsigned char* pt_mask; // optional per-point selection (marginalisation passes); null = every point
    int fuse_apply;               // residual kernel also performs applyRes(copyJacobians=true) (valid when the step is always accepted)
    int records_only;             // k_ba_linearize: only re-create the efsJ record of every good residual (cml_materialize_records)
    int no_lookahead;             // development (CMLHIP_NO_LOOKAHEAD): the plain factorisation loop for wide systems too
};

#define CML_DEBUG_RS_TILES 4096                 // development: per-tile stamps of k_ba_lin_rs behind the CMLHIP_DEBUG_SLOTS (cmlhip_debug_timestamps)
#define RS_TILE 64                                 // residuals per wave tile of the lane-per-residual kernel (16 for the 4-lane kernel: cmlhip_ctx::rs_tile)
// resident residual kernel (ba_linearize_rs.hip): wave tiles of <= RS_TILE residuals of one (host,target) pair
#define RS_LEAN_BIT 0x10000
#define RS_STAGGER_OF(flags) (((flags) >> 8) & 255)
struct RsArgs {
    const int4* tiles; int ntiles;                 // {first residual, count, host, target}
    const float* r_px; const float* r_py;          // the point's pixel, per residual
    const float* r_colors; const float* r_weights; // [R][8] the point's pattern colours / weights, per residual
    const double* r_idepth;                        // the point's inverse depth, per residual (kept current by k_ba_backsub's point step; refreshed when another path wrote pt_idepth)
    int dbg_flags;                                 // ONE word (the kernel sits at the scalar-register limit): bits 0-7 development switches (CMLHIP_RS_DBG);
                                                   // bits 8-15 RS_STAGGER: lane-per-residual kernel, phase shift of the waves sharing a SIMD, x 0.43 us x the wave's slot;
                                                   // bit 16 RS_LEAN: cmlhip_ba_set_resident_outputs(CMLHIP_RESIDENT_OUTPUTS_LEAN) — centerProjectedTo and the returned energy are
                                                   // not stored, state_NewEnergyWithOutlier only for residuals into the newest frame (setNewFrameEnergyTH's input, BA.cpp:2419-2464)
    const int* stop_lin;                           // never null: ResidentCtl::stop_lin of the window, or a word that stays zero (read with the inputs, tested behind them)
    float* part;                                   // [ntiles][64][4]: the wave's 16x16 fp32 tile of its pair's 13x13 block (MFMA D layout)
};
struct BatchRs { BAArgs A; RsArgs X; int blocks; int pad; };      // one window of a batched residual launch
int cml_launch_linearize_rs(cmlhip_ctx* c, const BAArgs& A);
int cml_launch_linearize_rs4(cmlhip_ctx* c, const BAArgs& A, RsArgs X);
int cml_materialize_records(cmlhip_ctx* c);
// several windows per launch (cmlhip_ba_iteration_batch): per-window {BAArgs, RsArgs, blocks} records appended to `blob`
int cml_fill_rs4_batch(cmlhip_ctx* c, const BAArgs& A, std::vector<unsigned char>& blob, int& blocks);
int cml_launch_linearize_rs4_batch(cmlhip_ctx* c0, const void* dev_records, int S, int max_blocks);
int cml_launch_linearize_rs_batch(cmlhip_ctx* c0, const void* dev_records, int S, int max_blocks);      // windows in the throughput regime (tiles of 64)
void cml_refresh_r_idepth(cmlhip_ctx* c, const BAArgs& A, hipStream_t stream);
int cml_iteration_batch(cmlhip_ctx* const* ctxs, int S, double lambda);        // re-create the efsJ records the resident kernel did not write (no state change)

// point slices of the Schur SYRK (k_ba_system): more slices = more CUs pulling rows, but more partials for the consumer to add
// (a power of two <= 8: the solve kernel is instantiated per slice count so that it issues exactly the loads it needs)
static inline int cml_sys_slices(int P) { return P <= 512 ? 1 : (P <= 1024 ? 2 : (P <= 2048 ? 4 : 8)); }
int cml_make_ba_args(cmlhip_ctx* c, BAArgs& A);
int cml_launch_linearize(cmlhip_ctx* c, const BAArgs& A);
int cml_launch_lin_finish(cmlhip_ctx* c, const BAArgs& A, size_t out_offset = 0);     // out_offset: where in the scalar scratch the LinSummary goes
int cml_launch_apply(cmlhip_ctx* c, const BAArgs& A, int copy);
// K3 (pair blocks + point rows) and K4 (system tiles: H_A, H_L, H_sc and the final LM system for `lambda` / optional HM)
int cml_launch_accumulate(cmlhip_ctx* c, const BAArgs& A, double lambda, bool have_hm, bool do_backup, bool system_only = false, bool marg = false);
struct ReprojArgs;
int cml_launch_solve(cmlhip_ctx* c, const BAArgs& A, int optcal, bool with_lin_finish, bool ortho = false, const double* indirect_x = nullptr, const ReprojArgs* rp = nullptr,
                     bool merge_backsub = false);      // merge_backsub: the back-substitution rides in this launch (c->backsub_merged tells whether it did)
void cml_resident_reproj_args(cmlhip_ctx* c, double lambda, int ticket, ReprojArgs* out);      // reproj.hip: the hybrid term of the resident iteration
int cml_launch_reproj_resident(cmlhip_ctx* c, double lambda);        // reproj.hip: addIndirectToProblem on the resident frame states -> rp_x
int cml_launch_schur_out(cmlhip_ctx* c, const BAArgs& A);
int cml_launch_marg_reset(cmlhip_ctx* c, const BAArgs& A);
int cml_launch_marg_fix(cmlhip_ctx* c, const BAArgs& A, const float* adHTd, const double* cdelta, int* counter);
int cml_launch_lin_energy(cmlhip_ctx* c, const BAArgs& A, const float* adHTd, const double* cdelta, double* partial, int* num);     // H_sc / b_sc for host readback
int cml_launch_backsub(cmlhip_ctx* c, const BAArgs& A, bool do_step);
int cml_launch_backup_points(cmlhip_ctx* c, const BAArgs& A);
int cml_launch_step_points(cmlhip_ctx* c, const BAArgs& A);
int cml_launch_restore_points(cmlhip_ctx* c, const BAArgs& A);

// Jp*delta of a linearised residual, in Eigen's evaluation order (pinned on the reference's vendored Eigen, tests/golden):
// Vector6f.dot(Vector8f.head<6>()) is ((x0 + x2) + (x1 + x3)) + (x4 + x5); the 4-dot is (c0 + c2) + (c1 + c3) against an
// evaluated Vector4f (BA.cpp:1699, 2166) and (c0 + c1) + (c2 + c3) when the cast expression stays inside the dot (BA.cpp:2219).
__device__ __forceinline__ float cml_jp_delta(const float* Jxi, const float* dp, const float* Jc, const double* cdelta, float jpdd, float dd, bool cast_in_dot) {
#pragma clang fp contract(off)
    const float x0 = Jxi[0] * dp[0], x1 = Jxi[1] * dp[1], x2 = Jxi[2] * dp[2], x3 = Jxi[3] * dp[3], x4 = Jxi[4] * dp[4], x5 = Jxi[5] * dp[5];
    const float c0 = Jc[0] * (float)cdelta[0], c1 = Jc[1] * (float)cdelta[1], c2 = Jc[2] * (float)cdelta[2], c3 = Jc[3] * (float)cdelta[3];
    const float d6 = ((x0 + x2) + (x1 + x3)) + (x4 + x5);
    const float d4 = cast_in_dot ? (c0 + c1) + (c2 + c3) : (c0 + c2) + (c1 + c3);
    return (d6 + d4) + jpdd * dd;
}
