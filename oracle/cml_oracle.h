/*
 * cml_oracle.h — CPU restatement (plain C) of libCML/MODSLAM's photometric hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (libcml_amd/, include/) may
 * include, link or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker.
 *
 * Pinning status: the reference's own translation units
 * include the cmake-generated cml/config.h and one of them needs Qt, so the reference
 * path is UNBUILDABLE in this image under the build rules, and the reference holds no
 * tests or golden vectors for this path.  The DSO-specific functions
 * below are therefore "PARITY UNPINNED" against reference outputs; what IS pinned:
 *   - the third-party arithmetic at the boundary (Eigen 3.4.0 LDLT / inverse /
 *     JacobiSVD pseudo-inverse, Sophus 1.1.0 SE3 exp/log/Adj/Dx_exp_x) against the
 *     vendored headers compiled from /root/reference/thirdparty (oracle/_ref, and the
 *     committed vectors tests/golden/thirdparty_*.json);
 *   - mathematical self-consistency: analytic Jacobians vs finite differences of the
 *     restated residual, Schur-complement solution vs a dense full-system solve.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference tree; BA.cpp = src/cml/optimization/dso/DSOBundleAdjustment.cpp,
 * TR.cpp = src/cml/optimization/dso/DSOTracker.cpp,
 * ACC.h = src/cml/optimization/dso/MatrixAccumulators.h).
 */
#ifndef CML_ORACLE_H
#define CML_ORACLE_H

#include "../include/cmlhip.h"

#ifdef __cplusplus
extern "C" {
/* ------------------------------------------------------------------ ORB side: pose-only optimisation */
/* IndirectCameraOptimizer::optimize + evaluateOutliers over the vendored g2o, IndirectCameraOptimizer.cpp:4-427 */
void orc_pnp_optimize(const double R0[9], const double t0[3], const double K[4], int n, const cmlhip_pnp_match* m,
                      unsigned char* outliers, int algorithm, int check_outliers, int compute_covariance, cmlhip_pnp_result* out);

/* ------------------------------------------------------------------ ORB side: local bundle adjustment */
/* IndirectBundleAdjustment::localOptimize + startOptimization + apply's edge test, IndirectBundleAdjustment.cpp:7-334 */
int orc_lba_optimize(int n_frames, cmlhip_lba_frame* frames, int n_points, double* points, const int* point_offsets,
                     const cmlhip_lba_edge* edges, int fix_frames, int num_iterations, int refine_iterations,
                     unsigned char* edge_bad, cmlhip_lba_result* out);
int orc_ldlt3(const double A[9], const double b[3], double x[3]);   /* Eigen::LDLT<Matrix3d>: returns isPositive() */
/* the SE3Quat / Eigen arithmetic shared by orc_pnp.c and orc_lba.c (orc_g2o.h), exported for pinning; quaternions x, y, z, w */
void orc_g2o_from_Rt(const double R[9], const double t[3], double q[4], double tt[3]);
void orc_g2o_exp(const double u[6], double q[4], double tt[3]);
void orc_g2o_mul(const double qa[4], const double ta[3], const double qb[4], const double tb[3], double q[4], double tt[3]);
void orc_g2o_map(const double q[4], const double t[3], const double X[3], double out[3]);
void orc_g2o_to_matrix(const double q[4], double R[9]);
void orc_g2o_inv3(const double A[9], double Ai[9]);
int  orc_g2o_llt_solve(const double* A, int n, const double* b, double* x);

#endif

/* ------------------------------------------------------------------ images */
/* level sizes: src/cml/capture/CaptureImage.cpp:39-78 (pyramidSize == -1 branch). returns #levels */
int  orc_pyramid_sizes(int w, int h, int* ws, int* hs, int max_levels);
/* 2x2 box mean: src/cml/image/Array2D.h:388-401 */
void orc_reduce_by_two(const float* in, int w, int h, float* out);
/* {I, 0.5(I[x+1]-I[x-1]), 0.5(I[y+1]-I[y-1])}, zero 1-px border: Array2D.h:288-327 */
void orc_gradient_image(const float* gray, int w, int h, float* aos3);
/* bilinear 3-channel tap: Array2D.h:265-286 */
void orc_interpolate3(const float* aos3, int w, float x, float y, float out[3]);

/* ------------------------------------------------------------------ SE3 (Sophus 1.1.0 semantics) */
typedef struct { double q[4]; /* w,x,y,z */ double t[3]; } orc_se3;
void orc_se3_identity(orc_se3* T);
void orc_se3_from_Rt(const double R[9], const double t[3], orc_se3* T);
void orc_se3_matrix(const orc_se3* T, double R[9]);
void orc_se3_exp(const double xi[6], orc_se3* T);              /* sophus/se3.hpp:776-797, so3.hpp:599-635 */
void orc_se3_log(const orc_se3* T, double xi[6]);              /* se3.hpp:224-257, so3.hpp:248-294 */
void orc_se3_mul(const orc_se3* A, const orc_se3* B, orc_se3* C);
void orc_se3_inv(const orc_se3* A, orc_se3* C);
void orc_se3_adj(const orc_se3* T, double A[36]);              /* se3.hpp:104-112 */
void orc_se3_dx_exp_x(const double xi[6], double J[42]);       /* se3.hpp Dx_exp_x, 7x6 row-major, rows: t(3) then q(x,y,z,w) */
/* Exposure::to: src/cml/map/Exposure.h:119-123 */
void orc_exposure_to(double a_from, double b_from, double t_from,
                     double a_to, double b_to, double t_to, double* a, double* b);

/* ------------------------------------------------------------------ dense (Eigen 3.4.0 semantics) */
/* LDLT<Lower> with diagonal pivoting + solve: Eigen/src/Cholesky/LDLT.h:300-396,560-600. A row-major n x n.
 * returns 0 ok. */
int  orc_ldlt_solve(const double* A, const double* b, int n, double* x);
/* general inverse by partial-pivot LU (Eigen PartialPivLU for n>4) */
int  orc_inverse(const double* A, int n, double* Ainv);
/* b -= N (N^T N)^+ N^T b with singular values <= delta*max dropped: BA.cpp:1196-1261 */
void orc_orthogonalize(double* b, int n, const double* Ncols /* n x m, column j at Ncols[j*n] */, int m, double delta);

/* ------------------------------------------------------------------ BA window (flat mirror of DSOContext) */
typedef struct {
    cmlhip_ba_params prm;
    int N, P, R;
    /* frames */
    const float* image[CMLHIP_MAX_FRAMES];    /* level-0 AoS3 gradient image of each frame (borrowed) */
    float frame_energy_th[CMLHIP_MAX_FRAMES];
    float b0[CMLHIP_MAX_FRAMES];
    cmlhip_ba_pair* pairs;                    /* N*N, host*N+target */
    /* points */
    cmlhip_ba_point* points;
    float* idepth_backup;
    float* Hdd_accAF; float* bd_accAF; float* Hcd_accAF;   /* P, P, P*4 */
    float* Hdd_accLF; float* bd_accLF; float* Hcd_accLF;
    float* HdiF; float* bdSumF; double* step;
    /* residuals */
    int* r_point; int* r_target; int* r_state; int* r_new_state; unsigned char* r_lin; unsigned char* r_good;
    float* r_energy; float* r_new_energy; float* r_new_energy_wo;
    float* r_center;      /* R*3 */
    float* rJ;            /* R*74 */
    float* efsJ;          /* R*74 */
    float* JpJdF;         /* R*8 */
    float* res_toZeroF;   /* R*8 */
    /* index maps */
    int* pair_of; int* by_point_off; int* by_point; int* by_pair_off; int* by_pair;
    /* raw accumulators */
    float* accA;          /* N*N*169 (finished AccumulatorApprox::H), index h + t*N */
    float* accL;
    int*   accA_num; int* accL_num;
} orc_ba_window;

orc_ba_window* orc_ba_create(const cmlhip_ba_params* prm, int N, const cmlhip_ba_frame* frames,
                             const float* const* images, int P, const cmlhip_ba_point* points,
                             int R, const cmlhip_ba_residual* residuals);
void orc_ba_destroy(orc_ba_window* w);
void orc_ba_set_pairs(orc_ba_window* w, const cmlhip_ba_pair* pairs);

/* DSOBundleAdjustmentLinearizationContext::linearize, BA.cpp:62-316. returns the energy it returns */
double orc_ba_linearize_one(orc_ba_window* w, int r);
/* linearizeAll(false) loop + setNewFrameEnergyTH: BA.cpp:1551-1565,1610,2419-2464 */
void orc_ba_linearize_all(orc_ba_window* w, cmlhip_ba_lin_result* out);
/* applyRes: BA.cpp:2051-2093 */
void orc_ba_apply(orc_ba_window* w, int copy_jacobians);
/* addToHessianTop x2 + stitchDoubleTop x2 + addToHessianSC + stitchDoubleSC: BA.cpp:1354-1385,1648-2043 */
void orc_ba_accumulate(orc_ba_window* w, const cmlhip_ba_accum_in* in,
                       double* HA, double* bA, double* HL, double* bL, double* Hsc, double* bsc);
/* solveLevenbergMarquardt without the indirect term/orthogonalize: BA.cpp:1284-1320 */
int  orc_ba_solve(const orc_ba_window* w, double lambda, const double* HA, const double* bA,
                  const double* HL, const double* bL, const double* HM, const double* bM,
                  const double* Hsc, const double* bsc, int optimize_calibration, double* x);
/* resubstitution: BA.cpp:1427-1487 */
int  orc_ba_backsub(orc_ba_window* w, const cmlhip_ba_accum_in* in, const double* x);
void orc_ba_backup_points(orc_ba_window* w);
void orc_ba_restore_points(orc_ba_window* w);                  /* BA.cpp:938-942 */
void orc_ba_step_points(orc_ba_window* w, float sums[3]);      /* BA.cpp:976-994 */
/* marginalisation, SURVEY §8 a15 */
void orc_ba_fix_linearization(orc_ba_window* w, int r, const cmlhip_ba_accum_in* in);              /* BA.cpp:2210-2238 */
int  orc_ba_relinearize_points(orc_ba_window* w, int n, const int* pts, const cmlhip_ba_accum_in* in);   /* BA.cpp:2291-2304 */
void orc_ba_marginalize_points(orc_ba_window* w, int n, const int* pts, const cmlhip_ba_accum_in* in,
                               double* M, double* Mb, double* Msc, double* Mbsc);                      /* BA.cpp:2466-2513 */
void orc_ba_marginalize_frame(double* HM, double* bM, int N, int frame, const double prior[8], const double delta_prior[8]);   /* BA.cpp:483-558 */
double orc_ba_calc_m_energy(const double* HM, const double* bM, int n, const double* delta);          /* BA.cpp:2095-2117 */
double orc_ba_calc_l_energy(const orc_ba_window* w, const cmlhip_ba_accum_in* in, int* num_out);      /* BA.cpp:2119-2208 */

/* host-side frame algebra used by the tests to build inputs the way the reference does */
typedef struct {
    orc_se3 w2c_eval;            /* worldToCam_evalPT */
    double state[10], state_zero[10], state_scaled[10], step[10], state_backup[10];
    double ab_exposure;
    orc_se3 PRE_w2c, PRE_c2w;
    double prior[8], delta[8], delta_prior[8], prior_zero[10];
    double ns_pose[36];          /* 6x6 col-major: column i at [i*6] */
    double ns_scale[6];
    double ns_affine[8];         /* 4x2 */
    int keyid;
} orc_frame;
typedef struct { double trans, rot, a, b; } orc_scales;
void orc_frame_set_state(orc_frame* f, const double state[10], const orc_scales* s);          /* DSOFrame.h:110-124 */
void orc_frame_set_state_scaled(orc_frame* f, const double ss[10], const orc_scales* s);      /* DSOFrame.h:126-142 */
void orc_frame_set_state_zero(orc_frame* f, const double sz[10], const orc_scales* s);        /* DSOFrame.h:154-186 */
void orc_frame_set_evalpt_scaled(orc_frame* f, const orc_se3* w2c, double aff_a, double aff_b,
                                 const orc_scales* s);                                        /* DSOFrame.h:99-108 */
void orc_frame_precompute(const orc_frame* host, const orc_frame* target, cmlhip_ba_pair* out);/* DSOFrame.h:259-273 */
/* computeAdjoints BA.cpp:1062-1097: adHost/adTarget N*N*64, index h+t*N, row-major 8x8 */
void orc_ba_compute_adjoints(const orc_frame* frames, int N, const orc_scales* s, double* adHost, double* adTarget);
/* computeDelta BA.cpp:1103-1177 (frame part): adHTdeltaF N*N*8; sets prior/delta/delta_prior */
void orc_ba_compute_delta(orc_frame* frames, int N, const double* adHost, const double* adTarget,
                          int optimize_a, int optimize_b, float* adHTdeltaF);
/* computeNullspaces BA.cpp:2365-2417: 7 vectors (6 pose + 1 scale) of length 8N+4, column j at out[j*n] */
void orc_ba_nullspaces(const orc_frame* frames, int N, const orc_scales* s, double* out7);

/* ------------------------------------------------------------------ tracker */
/* makeCoarseDepthL0 device part (TR.cpp:550-719): pts n x {Ku,Kv,new_idepth,weight}; gray[l] = gray level images.
 * lists[l] gets n_out[l] x {u,v,idepth,color}. scratch is managed internally. */
void orc_tracker_make_coarse_depth(const double* pts, int n, int levels, const int* ws, const int* hs,
                                   const float* const* gray, float** lists, int* n_out);
/* computeResidual + computeHessian: TR.cpp:248-492. warped: 8 x cap SoA rows (may be NULL) */
void orc_tracker_eval(const float* aos3, int w, int h, const float* uvic, int n, int level,
                      const double R[9], const double t[3], const double K[4], const double aff[2], double b0,
                      const cmlhip_tracker_params* prm, int want_hessian,
                      cmlhip_tracker_result* out, float* warped, int cap);

/* DSOTracker::optimize (TR.cpp:15-246) and trackWithMotionModel (TR.h:238-383) on top of orc_tracker_eval */
typedef struct { int level, iteration, accept; double lambda, E_new, E_old; int n_new, n_old; } orc_trk_step;
typedef struct {
    int levels;                                  /* pyramid levels of the frame to track (maxLevel = min(levels - 1, 4)) */
    const float* aos3[5]; int w[5], h[5];        /* its gradient images */
    const float* uvic[5]; int n[5];              /* reference lists (makeCoarseDepthL0) */
    double K[4];                                 /* level-0 pinhole fx fy cx cy */
    double ref_a, ref_b, ref_t, new_t;           /* reference exposure parameters / time, exposure time of the new frame */
    cmlhip_tracker_params prm;                   /* huber, cutoff_base, scales (cutoff is derived per level) */
    int optimize_a, optimize_b; double saturated_ratio_th;
    int have_last; double last_rmse[5];          /* mLastResidual.isCorrect, mLastResidual.rmse(level) */
} orc_trk_problem;
typedef struct {
    int isCorrect, tooManySaturated;
    double E[5]; int numTermsInE[5], numSaturated[5], numRobust[5]; double levelCutoffRepeat[5];
    double relAff[2], covariance[6], flow[3]; int n_steps;
} orc_trk_result;
int orc_tracker_optimize(const orc_trk_problem* P, orc_se3* refToNew, double* cur_a, double* cur_b, orc_trk_result* out,
                         orc_trk_step* log, int log_cap);
int orc_tracker_track_with_motion_model(orc_trk_problem* P, int n_hyp, const orc_se3* hyp, double init_a, double init_b, double last_coarse_rmse,
                                        int failure_mode, orc_se3* best_pose, double* best_a, double* best_b, orc_trk_result* best,
                                        double* achieved_out, int* winner, int* tries);

/* ------------------------------------------------------------------ hybrid ORB term */
/* ReprojectionError::jacobian, src/cml/optimization/Residual.h:59-100 ; returns 0 if not finite */
int  orc_reproj_jacobian(const double R[9], const double t[3], const double X[3], double gx, double gy,
                         double fx, double fy, double* residual, double Jt[3], double Jq[4], double Jp[3]);
/* addIndirectToProblem accumulation, BA.cpp:2607-2687 */
void orc_reproj_accumulate(int N, const double* poses, int M, const double* points, int n,
                           const cmlhip_reproj_obs* obs, double fx, double fy,
                           double* M6, double* b6, double* Jpoints, unsigned char* used);

#ifdef __cplusplus
}
/* ------------------------------------------------------------------ ORB side: pose-only optimisation */
/* IndirectCameraOptimizer::optimize + evaluateOutliers over the vendored g2o, IndirectCameraOptimizer.cpp:4-427 */
void orc_pnp_optimize(const double R0[9], const double t0[3], const double K[4], int n, const cmlhip_pnp_match* m,
                      unsigned char* outliers, int algorithm, int check_outliers, int compute_covariance, cmlhip_pnp_result* out);

/* ------------------------------------------------------------------ ORB side: local bundle adjustment */
/* IndirectBundleAdjustment::localOptimize + startOptimization + apply's edge test, IndirectBundleAdjustment.cpp:7-334 */
int orc_lba_optimize(int n_frames, cmlhip_lba_frame* frames, int n_points, double* points, const int* point_offsets,
                     const cmlhip_lba_edge* edges, int fix_frames, int num_iterations, int refine_iterations,
                     unsigned char* edge_bad, cmlhip_lba_result* out);
int orc_ldlt3(const double A[9], const double b[3], double x[3]);   /* Eigen::LDLT<Matrix3d>: returns isPositive() */
/* the SE3Quat / Eigen arithmetic shared by orc_pnp.c and orc_lba.c (orc_g2o.h), exported for pinning; quaternions x, y, z, w */
void orc_g2o_from_Rt(const double R[9], const double t[3], double q[4], double tt[3]);
void orc_g2o_exp(const double u[6], double q[4], double tt[3]);
void orc_g2o_mul(const double qa[4], const double ta[3], const double qb[4], const double tb[3], double q[4], double tt[3]);
void orc_g2o_map(const double q[4], const double t[3], const double X[3], double out[3]);
void orc_g2o_to_matrix(const double q[4], double R[9]);
void orc_g2o_inv3(const double A[9], double Ai[9]);
int  orc_g2o_llt_solve(const double* A, int n, const double* b, double* x);

#endif

/* Eigen expression shapes of the projection arithmetic (pinned on the vendored Eigen, see orc_base.c) */
void orc_eig_matvec3f_affine(const float M[9], const float v[3], const float t[3], float s, int sign, float out[3]);
void orc_eig_matvec3f_noalias(const float M[9], const float v[3], const float t[3], float s, float out[3]);
void orc_eig_homog3d(const double R[9], const double v[2], const double t[3], double s, double out[3]);
void orc_eig_matvec3d(const double M[9], const double v[3], double out[3]);
void orc_eig_matmul3f(const float A[9], const float B[9], float out[9]);
void orc_eig_matmul3d(const double A[9], const double B[9], double out[9]);
void orc_eig_inverse3f(const float m[9], float o[9]);
void orc_eig_inverse3d(const double m[9], double o[9]);
float orc_eig_jp_delta(const float Jxi[6], const float dp[8], const float Jc[4], const double cdelta[4], float Jpdd, float dd, int cast_in_dot);
double orc_eig_calib_dot(const double step[4], const float a[4], const float l[4]);
double orc_eig_row8_dot_cast(const double xa[8], const float J[8]);

/* ------------------------------------------------------------------ immature points: DSOTracer */
/* DSOTracer::trace, DSOTracer.cpp:585-823: aos3 = level-0 gradient image of the traced frame (channel 0 = gray) */
int orc_trace_point(const float* aos3, int w, int h, const cmlhip_trace_pair* pair, const cmlhip_tracer_params* prm,
                    cmlhip_immature_point* point);
/* DSOTracer::optimizeImmaturePoint + linearizeResidual, DSOTracer.cpp:280-494 */
int orc_optimize_immature_point(int N, const float* const* images, int w, int h, const double K[4], const cmlhip_activation_pair* pairs,
                                const cmlhip_tracer_params* prm, int min_obs, const cmlhip_immature_point* point, float* idepth_out,
                                int* res_state);

/* ------------------------------------------------------------------ coarse initializer */
/* DSOInitializer::calcResAndGS, DSOInitializer.cpp:451-750: aos3 = gradient image of the tracked frame at the level */
void orc_init_calc_res_and_gs(const float* aos3, int w, int h, const cmlhip_init_params* prm, int n, cmlhip_init_point* points,
                              float* H_out, float* b_out, float* H_out_sc, float* b_out_sc, float res[3]);

/* ------------------------------------------------------------------ ORB side: pose-only optimisation */
/* IndirectCameraOptimizer::optimize + evaluateOutliers over the vendored g2o, IndirectCameraOptimizer.cpp:4-427 */
void orc_pnp_optimize(const double R0[9], const double t0[3], const double K[4], int n, const cmlhip_pnp_match* m,
                      unsigned char* outliers, int algorithm, int check_outliers, int compute_covariance, cmlhip_pnp_result* out);

/* ------------------------------------------------------------------ ORB side: local bundle adjustment */
/* IndirectBundleAdjustment::localOptimize + startOptimization + apply's edge test, IndirectBundleAdjustment.cpp:7-334 */
int orc_lba_optimize(int n_frames, cmlhip_lba_frame* frames, int n_points, double* points, const int* point_offsets,
                     const cmlhip_lba_edge* edges, int fix_frames, int num_iterations, int refine_iterations,
                     unsigned char* edge_bad, cmlhip_lba_result* out);
int orc_ldlt3(const double A[9], const double b[3], double x[3]);   /* Eigen::LDLT<Matrix3d>: returns isPositive() */
/* the SE3Quat / Eigen arithmetic shared by orc_pnp.c and orc_lba.c (orc_g2o.h), exported for pinning; quaternions x, y, z, w */
void orc_g2o_from_Rt(const double R[9], const double t[3], double q[4], double tt[3]);
void orc_g2o_exp(const double u[6], double q[4], double tt[3]);
void orc_g2o_mul(const double qa[4], const double ta[3], const double qb[4], const double tb[3], double q[4], double tt[3]);
void orc_g2o_map(const double q[4], const double t[3], const double X[3], double out[3]);
void orc_g2o_to_matrix(const double q[4], double R[9]);
void orc_g2o_inv3(const double A[9], double Ai[9]);
int  orc_g2o_llt_solve(const double* A, int n, const double* b, double* x);

#endif
