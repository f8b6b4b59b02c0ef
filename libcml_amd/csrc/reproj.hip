// reproj.hip — hybrid ORB term: reprojection Jacobian accumulation into the pose system.
// Replaces DSOBundleAdjustment::addIndirectToProblem's accumulation (BA.cpp:2607-2700) and
// ReprojectionError::jacobian (src/cml/optimization/Residual.h:59-100, src/cml/map/Camera.h:317-386,
// src/cml/maths/Derivative.h:28-116, src/cml/maths/Rotation.cpp:205-290).
//
// The reference builds a sparse J of size (6N+3M) x (N*M), forms the DENSE H = J J^T ((6N+3M)^2 doubles, tens of
// MB) and then keeps only the 6N x 6N pose block.  Each column of J touches one frame and one point, so that pose
// block is block-diagonal: M6[i,i] = sum_j f_ij f_ij^T.  Here ONE WORKGROUP OWNS ONE FRAME: the observations are sorted by frame on
// upload (stable: the caller's order inside a frame), each lane walks its share of the frame's list in order, and the 21+6 unique
// sums are reduced in a FIXED order (lane partials -> wave butterfly -> the four waves in sequence) — no atomics anywhere, so the term
// and everything mixed from it are bit-reproducible from run to run (round 2 summed with fp64 atomicAdd and was not).  The per-point
// Jacobian sums (setUncertainty, BA.cpp:2690) leave the device per observation and are added on the host in the caller's order.
// All arithmetic is fp64.
#include "cmlhip_internal.h"
#include "../host/se3.h"

#include "reproj_dev.h"

__global__ __launch_bounds__(RP_THREADS) void k_reproj_frames(ReprojArgs a) {
    __shared__ double lds[RP_LDS_DOUBLES];
    reproj_frame_block(a, blockIdx.x, lds);
}

__global__ void k_reproj_solve(int N, double lambda, const double* __restrict__ M6, const double* __restrict__ b6, double* __restrict__ x6) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= N) return;
    const int m = 6 * N;
    double A[36];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) A[i * 6 + j] = M6[(size_t)(6 * f + i) * m + 6 * f + j];
    double wA[36], wx[6];
    int wtr[6];
    reproj_solve6(A, b6 + 6 * f, lambda, x6 + 6 * f, wA, wx, wtr);
}

// observations sorted by frame (stable), CSR offsets, and the caller's index of every sorted observation
static void sort_by_frame(int N, int n, const cmlhip_reproj_obs* obs, std::vector<cmlhip_reproj_obs>& sorted, std::vector<int>& off, std::vector<int>& orig) {
    off.assign(N + 1, 0);
    for (int k = 0; k < n; k++) off[obs[k].frame + 1]++;
    for (int i = 0; i < N; i++) off[i + 1] += off[i];
    sorted.resize(n ? n : 1); orig.resize(n ? n : 1);
    std::vector<int> cur(off.begin(), off.end() - 1);
    for (int k = 0; k < n; k++) { const int p = cur[obs[k].frame]++; sorted[p] = obs[k]; orig[p] = k; }
}

struct ReprojBufs { DevBuf *obs, *off, *orig, *points, *jp, *used, *x; };
static int upload_obs(cmlhip_ctx* c, const ReprojBufs& B, int N, int M, const double* points, int n, const cmlhip_reproj_obs* obs) {
    std::vector<cmlhip_reproj_obs> sorted; std::vector<int> off, orig;
    sort_by_frame(N, n, obs, sorted, off, orig);
    int rc;
#define ENS(buf, bytes) if ((rc = cml_ensure(c, *(buf), (size_t)(bytes)))) return rc
    ENS(B.obs, sizeof(cmlhip_reproj_obs) * sorted.size()); ENS(B.off, 4 * (size_t)(N + 1)); ENS(B.orig, 4 * orig.size());
    ENS(B.points, 8 * 3 * (size_t)(M ? M : 1)); ENS(B.jp, 8 * 3 * (size_t)(n ? n : 1)); ENS(B.used, (size_t)(n ? n : 1)); ENS(B.x, 8 * 6 * (size_t)N);
#undef ENS
    if (M && (rc = cml_h2d(c, B.points->p, points, 8 * 3 * (size_t)M))) return rc;
    if ((rc = cml_h2d(c, B.obs->p, sorted.data(), sizeof(cmlhip_reproj_obs) * sorted.size()))) return rc;
    if ((rc = cml_h2d(c, B.off->p, off.data(), 4 * off.size()))) return rc;
    return cml_h2d(c, B.orig->p, orig.data(), 4 * orig.size());
}

// per-point sums of the point Jacobians over the used observations (BA.cpp:2657-2659), added in the caller's observation order
static int read_point_jacobians(cmlhip_ctx* c, DevBuf& jp, int M, int n, const cmlhip_reproj_obs* obs_caller_order, const int* point_of, double* Jpoints) {
    std::vector<double> j(3 * (size_t)(n ? n : 1));
    int rc = n ? cml_d2h(c, j.data(), jp.p, 8 * 3 * (size_t)n) : 0;
    if (rc) return rc;
    for (int k = 0; k < 3 * M; k++) Jpoints[k] = 0.0;
    for (int k = 0; k < n; k++) {
        const int p = obs_caller_order ? obs_caller_order[k].point : point_of[k];
        for (int cc = 0; cc < 3; cc++) Jpoints[3 * (size_t)p + cc] += j[3 * (size_t)k + cc];
    }
    return CMLHIP_OK;
}

static ReprojArgs resident_args(cmlhip_ctx* c, double lambda) {
    ReprojArgs a = {};
    a.N = c->N; a.poses = nullptr; a.fs = c->frame_state.as<cmlhip_ba_frame_state>(); a.sc_t = c->res_scales[0]; a.sc_r = c->res_scales[1];
    a.off = c->rr_off.as<int>(); a.obs = c->rr_obs.as<cmlhip_reproj_obs>(); a.orig = c->rr_orig.as<int>(); a.points = c->rr_points.as<double>();
    a.fx = c->rp_res_fx; a.fy = c->rp_res_fy; a.lambda = lambda; a.M6 = nullptr; a.b6 = nullptr; a.x6 = c->rr_x.as<double>();
    a.jp_obs = c->rr_jp.as<double>(); a.used = c->rr_used.as<unsigned char>();
    a.ready = nullptr; a.ticket = 0;
    return a;
}
void cml_resident_reproj_args(cmlhip_ctx* c, double lambda, int ticket, ReprojArgs* out) {
    *out = resident_args(c, lambda);
    out->ready = c->rr_ready.as<int>(); out->ticket = ticket;
}

// addIndirectToProblem inside the device-resident iteration (BA.cpp:1327-1329, 2574-2729): ONE launch, a workgroup per frame — pose
// from the resident frame state, the frame's observations, the fixed-order sums and the damped 6x6 solve; the solve kernel of the
// iteration then replaces the pose part of x by rr_x (the literal weighting of :2714-2727) before the nullspace projection.
int cml_launch_reproj_resident(cmlhip_ctx* c, double lambda) {
    k_reproj_frames<<<c->N, RP_THREADS, 0, c->stream>>>(resident_args(c, lambda));
    CML_CHECK(c, hipGetLastError());
    return CMLHIP_OK;
}

extern "C" {

int cmlhip_ba_set_resident_indirect(cmlhip_ctx* c, int M, const double* points, int n, const cmlhip_reproj_obs* obs, double fx, double fy) { CML_DEV_SCOPED(c);
    if (!c || M < 0 || n < 0 || (M > 0 && !points) || (n > 0 && !obs)) return CMLHIP_ERR_INVALID;
    c->rp_resident = false;
    if (M == 0) return CMLHIP_OK;                                  // BA.cpp:2587-2589: nothing to mix (no device work: an open upload scope stays open)
    if (c->h2d_scope) { const int rcs = cml_scope_end(c); if (rcs) return rcs; }
    CML_REQUIRE(c, c->ba_uploaded && c->resident_on, CMLHIP_ERR_STATE, "cmlhip_ba_set_resident_state not called for this window");
    CML_REQUIRE(c, n <= c->lim.max_reproj_obs, CMLHIP_ERR_INVALID, "observations exceed max_reproj_obs");
    const int N = c->N;
    for (int k = 0; k < n; k++)
        CML_REQUIRE(c, obs[k].frame >= 0 && obs[k].frame < N && obs[k].point >= 0 && obs[k].point < M, CMLHIP_ERR_INVALID, "bad observation index");
    // the resident term owns its buffers (rr_*): a host-path cmlhip_reproj_accumulate between iterations cannot disturb it
    const ReprojBufs B{&c->rr_obs, &c->rr_off, &c->rr_orig, &c->rr_points, &c->rr_jp, &c->rr_used, &c->rr_x};
    int rc = upload_obs(c, B, N, M, points, n, obs);
    if (rc) return rc;
    if ((rc = cml_ensure(c, c->rr_ready, 4 * (size_t)N))) return rc;
    if ((rc = cml_zero(c, c->rr_ready.p, 4 * (size_t)N))) return rc;          // tickets of the frame workgroups (iteration number + 1)
    c->rr_point_of.resize(n);
    for (int k = 0; k < n; k++) c->rr_point_of[k] = obs[k].point;
    c->rp_res_M = M; c->rp_res_n = n; c->rp_res_fx = fx; c->rp_res_fy = fy;
    c->rp_resident = true;
    return CMLHIP_OK;
}

int cmlhip_ba_get_resident_indirect(cmlhip_ctx* c, double* x, double* x6, double* Jpoints) { CML_DEV(c);
    if (!c) return CMLHIP_ERR_INVALID;
    CML_REQUIRE(c, c->ba_uploaded && c->resident_on, CMLHIP_ERR_STATE, "cmlhip_ba_set_resident_state not called for this window");
    int rc;
    if (x && (rc = cml_d2h(c, x, c->xvec.p, 8 * (8 * (size_t)c->N + 4)))) return rc;
    if (c->rp_resident) {
        if (x6 && (rc = cml_d2h(c, x6, c->rr_x.p, 8 * 6 * (size_t)c->N))) return rc;
        if (Jpoints && (rc = read_point_jacobians(c, c->rr_jp, c->rp_res_M, c->rp_res_n, nullptr, c->rr_point_of.data(), Jpoints))) return rc;
    }
    return CMLHIP_OK;
}

int cmlhip_reproj_accumulate(cmlhip_ctx* c, int N, const double* poses, int M, const double* points, int n,
                             const cmlhip_reproj_obs* obs, double fx, double fy, double* M6, double* b6, double* Jpoints,
                             unsigned char* used) { CML_DEV(c);
    if (!c || N < 1 || N > CMLHIP_MAX_FRAMES || M < 0 || n < 0 || !poses || (M > 0 && !points) || (n > 0 && !obs)) return CMLHIP_ERR_INVALID;
    CML_REQUIRE(c, n <= c->lim.max_reproj_obs, CMLHIP_ERR_INVALID, "observations exceed max_reproj_obs");
    for (int k = 0; k < n; k++)
        CML_REQUIRE(c, obs[k].frame >= 0 && obs[k].frame < N && obs[k].point >= 0 && obs[k].point < M, CMLHIP_ERR_INVALID, "bad observation index");
    const int m = 6 * N;
    int rc;
    const ReprojBufs B{&c->rp_obs, &c->rp_off, &c->rp_orig, &c->rp_points, &c->rp_Jp, &c->rp_used, &c->rp_x};
    if ((rc = upload_obs(c, B, N, M, points, n, obs))) return rc;
    if ((rc = cml_ensure(c, c->rp_poses, 8 * 12 * (size_t)N))) return rc;
    if ((rc = cml_ensure(c, c->rp_M, 8 * (size_t)m * m))) return rc;
    if ((rc = cml_ensure(c, c->rp_b, 8 * (size_t)m))) return rc;
    if ((rc = cml_h2d(c, c->rp_poses.p, poses, 8 * 12 * (size_t)N))) return rc;
    CML_CHECK(c, hipMemsetAsync(c->rp_M.p, 0, 8 * (size_t)m * m, c->stream));
    ReprojArgs a = {};
    a.N = N; a.poses = c->rp_poses.as<double>(); a.off = c->rp_off.as<int>(); a.obs = c->rp_obs.as<cmlhip_reproj_obs>(); a.orig = c->rp_orig.as<int>();
    a.points = c->rp_points.as<double>(); a.fx = fx; a.fy = fy; a.lambda = 0; a.M6 = c->rp_M.as<double>(); a.b6 = c->rp_b.as<double>(); a.x6 = nullptr;
    a.jp_obs = c->rp_Jp.as<double>(); a.used = c->rp_used.as<unsigned char>();
    k_reproj_frames<<<N, RP_THREADS, 0, c->stream>>>(a);
    CML_CHECK(c, hipGetLastError());
    c->rp_acc_N = N;
    if (M6 && (rc = cml_d2h(c, M6, c->rp_M.p, 8 * (size_t)m * m))) return rc;
    if (b6 && (rc = cml_d2h(c, b6, c->rp_b.p, 8 * (size_t)m))) return rc;
    if (Jpoints && M && (rc = read_point_jacobians(c, c->rp_Jp, M, n, obs, nullptr, Jpoints))) return rc;
    if (used && n && (rc = cml_d2h(c, used, c->rp_used.p, (size_t)n))) return rc;
    return CMLHIP_OK;
}

int cmlhip_reproj_solve(cmlhip_ctx* c, int N, double lambda, double* x6) { CML_DEV(c);
    if (!c || N < 1 || N > CMLHIP_MAX_FRAMES || !x6) return CMLHIP_ERR_INVALID;
    CML_REQUIRE(c, c->rp_M.p && c->rp_x.p && c->rp_acc_N == N, CMLHIP_ERR_STATE, "cmlhip_reproj_accumulate not called for this N");
    k_reproj_solve<<<1, 64, 0, c->stream>>>(N, lambda, c->rp_M.as<double>(), c->rp_b.as<double>(), c->rp_x.as<double>());
    CML_CHECK(c, hipGetLastError());
    int rc = cml_d2h(c, x6, c->rp_x.p, 8 * 6 * (size_t)N);
    if (rc) return rc;
    for (int i = 0; i < 6 * N; i++) if (!std::isfinite(x6[i])) return CMLHIP_ERR_NONFINITE;   // BA.cpp:2702-2704
    return CMLHIP_OK;
}

}  // extern "C"
