// capi.cpp — extern "C" handles onto the C++ host mirror (for the Python tests / bench; a C++ caller uses the
// classes directly).  Nothing here computes: it forwards to cml_amd::DSOBundleAdjustment / DSOTracker.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "DSOBundleAdjustment.h"
#include "DSOTracker.h"
#include "HostLap.h"
#include "DSOTracer.h"
#include "IndirectG2O.h"
#include "DSOInitializer.h"

using namespace cml_amd;

extern "C" {

void* cmlhost_ba_create(cmlhip_ctx* ctx) { return new DSOBundleAdjustment(ctx); }
void cmlhost_ba_destroy(void* h) { delete static_cast<DSOBundleAdjustment*>(h); }
void cmlhost_ba_set_calibration(void* h, double fx, double fy, double cx, double cy, int w, int hgt) {
    static_cast<DSOBundleAdjustment*>(h)->setCalibration(fx, fy, cx, cy, w, hgt);
}
int cmlhost_ba_set_param(void* h, const char* name, double v) {
    DSOBundleAdjustment* b = static_cast<DSOBundleAdjustment*>(h);
    const std::string n(name);
    if (n == "iterations") b->mNumIterations = (int)v;
    else if (n == "fixedLambda") b->mFixedLambda = v;
    else if (n == "fixLambda") b->mFixLambda = v != 0;
    else if (n == "forceAccept") b->mForceAccept = v != 0;
    else if (n == "optimizeLightA") b->mOptimizeA = v != 0;
    else if (n == "optimizeLightB") b->mOptimizeB = v != 0;
    else if (n == "disableMarginalization") b->mDisableMarginalization = v != 0;
    else if (n == "optimizeCalibration") b->mOptimizeCalibration = v != 0;
    else if (n == "Huber threshold") b->mHuberThreshold = v;
    else if (n == "outlierTHSumComponent") b->mSettingOutlierTHSumComponent = v;
    else if (n == "ThOptIterations") b->mThOptIterations = v;
    else if (n == "iDepth Fix Prior") b->mIdepthFixPrior = (int)v;
    else if (n == "Solver mode delta") b->mSolverModeDelta = v;
    else if (n == "mixedBundleAdjustment") b->mMixedBundleAdjustment = v != 0;
    else if (n == "residentLoop") b->mResidentLoop = v != 0;
    else if (n == "keepResidualEnergies") b->mKeepResidualEnergies = v != 0;
    else if (n == "leanResidentOutputs") b->mLeanResidentOutputs = v != 0;
    else if (n == "relaxedArithmetic") b->mRelaxedArithmetic = v != 0;
    else if (n == "Minimum iDepth Hessian Marginlaization") b->mMinIdepthHMarg = v;
    else if (n == "maxFrames") b->mMaxFrames = (int)v;
    else if (n == "frameMinAge") b->mMinFrameAge = (int)v;
    else return 1;      // unknown keys are an error, like the reference's YAML loader (AbstractSlam.h:70-83)
    return 0;
}
int cmlhost_ba_add_frame(void* h, uint64_t image_id, const double R[9], const double t[3], double a, double b, double exposure) {
    return static_cast<DSOBundleAdjustment*>(h)->addNewFrame(image_id, SE3::fromRt(R, t), Exposure(exposure, a, b));
}
// test hook: drifted state of an older keyframe (state != state_zero)
void cmlhost_ba_set_frame_state(void* h, int f, const double state[10]) {
    DSOBundleAdjustment* b = static_cast<DSOBundleAdjustment*>(h);
    double sc[4] = {b->mScaleTranslation, b->mScaleRotation, b->mScaleLightA, b->mScaleLightB};
    b->getFrames()[f].setState(state, sc);
}
void cmlhost_ba_set_frame_energy_th(void* h, int f, double th) { static_cast<DSOBundleAdjustment*>(h)->getFrames()[f].frameEnergyTH = th; }
// INDEX LIFETIME.  Point / residual indices (what cmlhost_ba_add_point(s), cmlhost_tracer_add_activated_to_ba return, what cmlhost_ba_outliers and the
// export calls report) number the object's lists as they stand NOW.  Entries that were dropped (outliers, marginalised points, residuals the closing pass
// of run() removed, everything of a marginalised frame) stay in the lists, flagged dead, until the lists are renumbered — by the next cmlhost_ba_add_frame
// (BA::addNewFrame) or, if something was dropped since, at the start of the next run().  A renumbering keeps the order of the survivors and invalidates
// every index handed out before it; cmlhost_ba_outliers refers to the lists as they were when the last run() returned, so read it before the next
// cmlhost_ba_add_frame.  (The reference has no such indices: its sets simply lose the objects.)
int cmlhost_ba_add_point(void* h, float x, float y, double idepth, int host, const float colors[8], const float weights[8], int prior) {
    return static_cast<DSOBundleAdjustment*>(h)->addPoint(x, y, idepth, host, colors, weights, prior != 0);
}
// n points at once (addPoints, BA.cpp:382-415): xy n x 2, colors / weights n x 8; returns the index of the first
int cmlhost_ba_add_points(void* h, int n, const float* xy, const double* idepth, const int* host, const float* colors, const float* weights, int prior) {
    DSOBundleAdjustment* b = static_cast<DSOBundleAdjustment*>(h);
    const int first = (int)b->getPoints().size();
    for (int i = 0; i < n; i++) b->addPoint(xy[2 * i], xy[2 * i + 1], idepth[i], host[i], colors + 8 * (size_t)i, weights + 8 * (size_t)i, prior != 0);
    b->handOverNewEntries();                                 // the library's window gets the batch now (BA::addPoints builds its residuals here, not in run())
    return first;
}
int cmlhost_ba_run(void* h, int updatePointsOnly) { return static_cast<DSOBundleAdjustment*>(h)->run(updatePointsOnly != 0) ? 1 : 0; }
int cmlhost_ba_run_host_loop(void* h, int updatePointsOnly) { return static_cast<DSOBundleAdjustment*>(h)->runHostLoop(updatePointsOnly != 0) ? 1 : 0; }
int cmlhost_ba_run_resident(void* h, int updatePointsOnly) { return static_cast<DSOBundleAdjustment*>(h)->runResident(updatePointsOnly != 0) ? 1 : 0; }
int cmlhost_ba_begin_resident(void* h, int updatePointsOnly) { return static_cast<DSOBundleAdjustment*>(h)->beginResident(updatePointsOnly != 0) ? 1 : 0; }
int cmlhost_ba_iterate_resident(void* h, int k, double lambda) { return static_cast<DSOBundleAdjustment*>(h)->iterateResident(k, lambda) ? 1 : 0; }
int cmlhost_ba_end_resident(void* h, double* lastEnergy) { return static_cast<DSOBundleAdjustment*>(h)->endResident(lastEnergy) ? 1 : 0; }
// ---- marginalisation (BA.h:34-46)
void cmlhost_ba_flag_frame(void* h, int f, int flag) { static_cast<DSOBundleAdjustment*>(h)->getFrames()[f].flaggedForMarginalization = flag != 0; }
void cmlhost_ba_flag_frames_for_marginalization(void* h, int immature) { static_cast<DSOBundleAdjustment*>(h)->flagFramesForMarginalization(immature); }
void cmlhost_ba_flag_frames_for_marginalization_v(void* h, int n, const int* immaturePerFrame) {
    static_cast<DSOBundleAdjustment*>(h)->flagFramesForMarginalization(std::vector<int>(immaturePerFrame, immaturePerFrame + n));
}
// ---- the mirror's whole state as flat records (sequence checker: an oracle window is built from exactly what the mirror holds)
struct cmlhost_ba_frame_rec {
    double eval_q[4], eval_t[3], pre_q[4], pre_t[3], state[10], state_zero[10], prior_zero[10], ab_exposure, frameEnergyTH;
    uint64_t image_id;
    int id, keyid, flagged, numMarginalized, numResidualsOut, pad;
};
struct cmlhost_ba_point_rec {
    double idepth;
    float x, y, colors[8], weights[8], idepth_zero, priorF, idepth_hessian, pad0;
    int host, hasDepthPrior, numGoodResiduals, lastResidual[2], lastResidualState[2], toMarginalize, marginalized, alive;
};
struct cmlhost_ba_residual_rec {
    double state_energy, state_NewEnergy;
    int point, target, state_state, state_NewState, isLinearized, good, alive, pad;
};
int cmlhost_ba_export(void* h, cmlhost_ba_frame_rec* fr, cmlhost_ba_point_rec* pt, cmlhost_ba_residual_rec* rs) {
    DSOBundleAdjustment* b = static_cast<DSOBundleAdjustment*>(h);
    if (fr) for (size_t i = 0; i < b->getFrames().size(); i++) {
        const DSOFrame& F = b->getFrames()[i]; cmlhost_ba_frame_rec& o = fr[i];
        std::memset(&o, 0, sizeof o);
        std::memcpy(o.eval_q, F.worldToCam_evalPT.q, sizeof o.eval_q); std::memcpy(o.eval_t, F.worldToCam_evalPT.t, sizeof o.eval_t);
        std::memcpy(o.pre_q, F.PRE_worldToCam.q, sizeof o.pre_q); std::memcpy(o.pre_t, F.PRE_worldToCam.t, sizeof o.pre_t);
        std::memcpy(o.state, F.state, sizeof o.state); std::memcpy(o.state_zero, F.state_zero, sizeof o.state_zero);
        std::memcpy(o.prior_zero, F.prior_zero, sizeof o.prior_zero);
        o.ab_exposure = F.ab_exposure; o.frameEnergyTH = F.frameEnergyTH; o.image_id = F.image_id; o.id = F.id; o.keyid = F.keyid;
        o.flagged = F.flaggedForMarginalization; o.numMarginalized = F.numMarginalized; o.numResidualsOut = F.numResidualsOut;
    }
    if (pt) for (size_t i = 0; i < b->getPoints().size(); i++) {
        const DSOPoint& P = b->getPoints()[i]; cmlhost_ba_point_rec& o = pt[i];
        std::memset(&o, 0, sizeof o);
        o.idepth = P.idepth; o.x = P.x; o.y = P.y; std::memcpy(o.colors, P.colors, sizeof o.colors); std::memcpy(o.weights, P.weights, sizeof o.weights);
        o.idepth_zero = P.idepth_zero; o.priorF = P.priorF; o.idepth_hessian = P.idepth_hessian; o.host = P.host; o.hasDepthPrior = P.hasDepthPrior;
        o.numGoodResiduals = P.numGoodResiduals;
        for (int q = 0; q < 2; q++) { o.lastResidual[q] = P.lastResidual[q]; o.lastResidualState[q] = P.lastResidualState[q]; }
        o.toMarginalize = P.toMarginalize; o.marginalized = P.marginalized; o.alive = P.alive;
    }
    if (rs) for (size_t i = 0; i < b->getResiduals().size(); i++) {
        const DSOResidual& R = b->getResiduals()[i]; cmlhost_ba_residual_rec& o = rs[i];
        std::memset(&o, 0, sizeof o);
        o.state_energy = R.state_energy; o.state_NewEnergy = R.state_NewEnergy; o.point = R.point; o.target = R.target; o.state_state = R.state_state;
        o.state_NewState = R.state_NewState; o.isLinearized = R.isLinearized; o.good = R.isActiveAndIsGoodNEW; o.alive = R.alive;
    }
    return (int)sizeof(cmlhost_ba_frame_rec) | ((int)sizeof(cmlhost_ba_point_rec) << 10) | ((int)sizeof(cmlhost_ba_residual_rec) << 20);
}
int cmlhost_ba_try_marginalize(void* h) { return static_cast<DSOBundleAdjustment*>(h)->tryMarginalize() ? 1 : 0; }
int cmlhost_ba_marginalize_points(void* h) { return static_cast<DSOBundleAdjustment*>(h)->marginalizePointsF() ? 1 : 0; }
int cmlhost_ba_marginalize_frames(void* h, int* removed, int cap) {
    const std::vector<int> r = static_cast<DSOBundleAdjustment*>(h)->marginalizeFrames();
    for (int i = 0; i < (int)r.size() && i < cap; i++) removed[i] = r[i];
    return (int)r.size();
}
int cmlhost_ba_get_prior(void* h, double* HM, double* bM) {          // returns n = 8N+4
    DSOBundleAdjustment* b = static_cast<DSOBundleAdjustment*>(h);
    const int n = (int)b->marginalizedB().size();
    if (HM) std::memcpy(HM, b->marginalizedHessian().data(), sizeof(double) * n * n);
    if (bM) std::memcpy(bM, b->marginalizedB().data(), sizeof(double) * n);
    return n;
}
void cmlhost_ba_get_point_flags(void* h, unsigned char* toMarg, unsigned char* marginalized, float* idepthHessian) {
    const auto& P = static_cast<DSOBundleAdjustment*>(h)->getPoints();
    for (size_t i = 0; i < P.size(); i++) { if (toMarg) toMarg[i] = P[i].toMarginalize; if (marginalized) marginalized[i] = P[i].marginalized; if (idepthHessian) idepthHessian[i] = P[i].idepth_hessian; }
}
void cmlhost_ba_set_indirect_points(void* h, int M, const double* xyz, int n, const cmlhip_reproj_obs* obs) {   // addIndirectToProblem's inputs
    static_cast<DSOBundleAdjustment*>(h)->setIndirectPoints(std::vector<double>(xyz, xyz + 3 * (size_t)M), std::vector<cmlhip_reproj_obs>(obs, obs + n));
}
int cmlhost_ba_get_indirect(void* h, double* x6, double* uncertainty, double* x) {   // last indirectX (6N), point uncertainties (M), last x (8N+4); returns 6N or 0
    DSOBundleAdjustment* b = static_cast<DSOBundleAdjustment*>(h);
    if (x6) std::copy(b->lastIndirectX().begin(), b->lastIndirectX().end(), x6);
    if (uncertainty) std::copy(b->indirectUncertainty().begin(), b->indirectUncertainty().end(), uncertainty);
    if (x) std::copy(b->lastX().begin(), b->lastX().end(), x);
    return (int)b->lastIndirectX().size();
}
// getGoodPointsForTracking (BA.h:76-85) + the host half of DSOTracker::makeCoarseDepthL0 (TR.cpp:521-553): every live point whose newest residual is
// IN, projected from its host frame into keyframe `kf` — (u, v, idepth in kf, weight) per point, 4 doubles each; returns the count (cap: room in `out`)
int cmlhost_ba_coarse_depth_points(void* h, int kf, const double K[4], double* out, int cap) {
    DSOBundleAdjustment* b = static_cast<DSOBundleAdjustment*>(h);
    const auto& F = b->getFrames(); const auto& Pts = b->getPoints();
    if (kf < 0 || kf >= (int)F.size()) return -1;
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    std::vector<double> Rr(9 * F.size()), tr(3 * F.size());
    double Rn[9]; F[kf].PRE_worldToCam.matrix(Rn);
    const double* tn = F[kf].PRE_worldToCam.t;
    for (size_t hh = 0; hh < F.size(); hh++) {                       // host -> kf: R = Rn Rh^T, t = tn - R th (Camera::to)
        double Rh[9]; F[hh].PRE_worldToCam.matrix(Rh);
        const double* th = F[hh].PRE_worldToCam.t;
        double* R = &Rr[9 * hh];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = Rn[i * 3] * Rh[j * 3] + Rn[i * 3 + 1] * Rh[j * 3 + 1] + Rn[i * 3 + 2] * Rh[j * 3 + 2];
        for (int i = 0; i < 3; i++) tr[3 * hh + i] = tn[i] - (R[i * 3] * th[0] + R[i * 3 + 1] * th[1] + R[i * 3 + 2] * th[2]);
    }
    int n = 0;
    for (const auto& P : Pts) {
        if (!P.alive || P.lastResidual[0] < 0 || P.lastResidualState[0] != DSORES_IN) continue;
        if (n >= cap) return -2;
        const double* R = &Rr[9 * (size_t)P.host]; const double* t = &tr[3 * (size_t)P.host];
        const double idp = (double)P.idepth;
        const double r0 = ((double)P.x - cx) * (1.0 / fx), r1 = ((double)P.y - cy) * (1.0 / fy);
        const double p0 = R[0] * r0 + R[1] * r1 + R[2] + idp * t[0], p1 = R[3] * r0 + R[4] * r1 + R[5] + idp * t[1], p2 = R[6] * r0 + R[7] * r1 + R[8] + idp * t[2];
        const double unc = 1.0 / ((double)P.idepth_hessian + 0.01);                        // DSOPoint::updatePointUncertainty (DSOPoint.h:107-117)
        const float wgt = std::sqrt((float)(1e-3 / (unc + 1e-12)));
        double* o = out + 4 * (size_t)n;
        o[0] = (p0 / p2) * fx + cx; o[1] = (p1 / p2) * fy + cy; o[2] = (1.0 / p2) * idp; o[3] = (double)wgt;
        n++;
    }
    return n;
}
// development (CMLHOST_TIMING=sum): forget the laps collected so far (a warm-up pass)
void cmlhost_laps_reset() { cml_amd::HostLapTable::get().rows.clear(); }
void cmlhost_ba_run_timing(void* h, double us[6]) { for (int i = 0; i < 6; i++) us[i] = static_cast<DSOBundleAdjustment*>(h)->lastRunUs[i]; }
int cmlhost_ba_rejected(void* h) { return static_cast<DSOBundleAdjustment*>(h)->statRejected; }
double cmlhost_ba_last_lambda(void* h) { return static_cast<DSOBundleAdjustment*>(h)->lastLambda; }
double cmlhost_ba_calc_m_energy(void* h) { return static_cast<DSOBundleAdjustment*>(h)->calcMEnergy(); }
double cmlhost_ba_calc_l_energy(void* h) { return static_cast<DSOBundleAdjustment*>(h)->calcLEnergy(); }
const char* cmlhost_ba_last_error(void* h) { return static_cast<DSOBundleAdjustment*>(h)->lastError().c_str(); }
int cmlhost_ba_counts(void* h, int* nframes, int* npoints, int* nresiduals, int* noutliers, int* iterations) {
    DSOBundleAdjustment* b = static_cast<DSOBundleAdjustment*>(h);
    *nframes = (int)b->getFrames().size(); *npoints = (int)b->getPoints().size(); *nresiduals = (int)b->getResiduals().size();
    *noutliers = (int)b->getOutliers().size(); *iterations = b->lastIterations;
    return 0;
}
// frame f: R[9] t[3] of PRE_worldToCam, aff a b (state_scaled[6:8]), state[10], frameEnergyTH
void cmlhost_ba_get_frame(void* h, int f, double R[9], double t[3], double ab[2], double state[10], double* th) {
    const DSOFrame& F = static_cast<DSOBundleAdjustment*>(h)->getFrames()[f];
    F.PRE_worldToCam.matrix(R);
    std::memcpy(t, F.PRE_worldToCam.t, 3 * sizeof(double));
    ab[0] = F.state_scaled[6]; ab[1] = F.state_scaled[7];
    std::memcpy(state, F.state, 10 * sizeof(double));
    *th = F.frameEnergyTH;
}
void cmlhost_ba_get_points(void* h, double* idepth, unsigned char* alive, int* numGood) {
    auto& P = static_cast<DSOBundleAdjustment*>(h)->getPoints();
    for (size_t i = 0; i < P.size(); i++) { idepth[i] = P[i].idepth; alive[i] = P[i].alive; numGood[i] = P[i].numGoodResiduals; }
}
void cmlhost_ba_get_residual_states(void* h, int* state, unsigned char* alive, unsigned char* good) {
    auto& R = static_cast<DSOBundleAdjustment*>(h)->getResiduals();
    for (size_t i = 0; i < R.size(); i++) { state[i] = R[i].state_state; alive[i] = R[i].alive; good[i] = R[i].isActiveAndIsGoodNEW; }
}
void cmlhost_ba_get_outliers(void* h, int* out) {
    auto& o = static_cast<DSOBundleAdjustment*>(h)->getOutliers();
    for (size_t i = 0; i < o.size(); i++) out[i] = o[i];
}
// host algebra exposed for parity tests against the oracle's frame algebra
void cmlhost_ba_get_algebra(void* h, double* adHost, double* adTarget, float* adHTdeltaF, cmlhip_ba_pair* pairs, double* prior,
                            double* delta_prior, double* nullspaces7) {
    DSOBundleAdjustment* b = static_cast<DSOBundleAdjustment*>(h);
    b->computeAdjoints();
    b->computeDelta();
    const size_t N = b->getFrames().size();
    std::memcpy(adHost, b->adHost().data(), 8 * 64 * N * N);
    std::memcpy(adTarget, b->adTarget().data(), 8 * 64 * N * N);
    std::memcpy(adHTdeltaF, b->adHTdeltaF().data(), 4 * 8 * N * N);
    std::vector<cmlhip_ba_pair> p;
    b->framePairs(p);
    std::memcpy(pairs, p.data(), sizeof(cmlhip_ba_pair) * N * N);
    for (size_t i = 0; i < N; i++) for (int k = 0; k < 8; k++) { prior[8 * i + k] = b->getFrames()[i].prior[k]; delta_prior[8 * i + k] = b->getFrames()[i].delta_prior[k]; }
    std::vector<double> ns;
    b->computeNullspaces(ns);
    std::memcpy(nullspaces7, ns.data(), 8 * ns.size());
}
void cmlhost_ba_orthogonalize(void* h, double* x, int n) {
    std::vector<double> v(x, x + n);
    static_cast<DSOBundleAdjustment*>(h)->orthogonalize(v);
    std::memcpy(x, v.data(), 8 * (size_t)n);
}
int cmlhost_ba_stats(void* h, double* energyP, int cap) {
    auto& s = static_cast<DSOBundleAdjustment*>(h)->statEnergyP;
    int n = (int)s.size() < cap ? (int)s.size() : cap;
    for (int i = 0; i < n; i++) energyP[i] = s[s.size() - n + i];
    return n;
}

// ---------------------------------------------------------------------------------------------- tracker
void* cmlhost_tracker_create(cmlhip_ctx* ctx) { return new DSOTracker(ctx); }
void cmlhost_tracker_destroy(void* h) { delete static_cast<DSOTracker*>(h); }
void cmlhost_tracker_set_calibration(void* h, double fx, double fy, double cx, double cy) { static_cast<DSOTracker*>(h)->setCalibration(fx, fy, cx, cy); }
int cmlhost_tracker_set_param(void* h, const char* name, double v) {
    DSOTracker* t = static_cast<DSOTracker*>(h);
    const std::string n(name);
    if (n == "optimizeLightA") t->mOptimizeA = v != 0;
    else if (n == "optimizeLightB") t->mOptimizeB = v != 0;
    else if (n == "Cutoff threshold") t->mCutoffThreshold = v;
    else if (n == "Huber threshold") t->mHuberThreshold = v;
    else if (n == "saturatedThreshold") t->mSaturatedRatioThreshold = v;
    else if (n == "maxLevel") t->maxLevelOverride = (int)v;
    else if (n == "failureMode") t->mFailureMode = (int)v;
    else if (n == "lastCoarseRMSE") t->mLastCoarseRMSE = v;
    else if (n == "batchedFirstAlone") t->mBatchedFirstAlone = v != 0;
    else if (n == "batchedEarlyExit") t->mBatchedEarlyExit = v != 0;
    else return 1;
    return 0;
}
int cmlhost_tracker_make_coarse_depth(void* h, uint64_t ref_image, int levels, const double* pts, int n, int* n_out) {
    return static_cast<DSOTracker*>(h)->makeCoarseDepthL0(ref_image, levels, pts, n, n_out) ? 1 : 0;
}
// optimize(): refToNew in/out as R[9], t[3]; reference exposure (a,b,t) and current exposure in/out
int cmlhost_tracker_optimize(void* h, uint64_t new_image, int levels, double R[9], double t[3], const double refExp[3], double curExp[3],
                             double* E, int* numTerms, int* numSat, double flow[3], double relAff[2], double cov[6], int* isCorrect,
                             int* tooManySaturated, int* iterationsPerLevel) {
    DSOTracker* T = static_cast<DSOTracker*>(h);
    SE3 refToNew = SE3::fromRt(R, t);
    Exposure ref(refExp[2], refExp[0], refExp[1]), cur(curExp[2], curExp[0], curExp[1]);
    DSOTracker::Residual res = T->optimize(new_image, levels, refToNew, ref, cur);
    refToNew.matrix(R);
    std::memcpy(t, refToNew.t, 3 * sizeof(double));
    curExp[0] = cur.a; curExp[1] = cur.b;
    for (int l = 0; l < levels && l < (int)res.E.size(); l++) { E[l] = res.E[l]; numTerms[l] = res.numTermsInE[l]; numSat[l] = res.numSaturated[l]; iterationsPerLevel[l] = res.iterations[l]; }
    for (int k = 0; k < 3; k++) flow[k] = res.flowVector[k];
    relAff[0] = res.relAff[0]; relAff[1] = res.relAff[1];
    for (int k = 0; k < 6; k++) cov[k] = res.covariance[k];
    *isCorrect = res.isCorrect; *tooManySaturated = res.tooManySaturated;
    return res.isCorrect ? 1 : 0;
}
const char* cmlhost_tracker_last_error(void* h) { return static_cast<DSOTracker*>(h)->lastError().c_str(); }
// tests: install the evaluation provider (the oracle's computeResidual + computeHessian) / read the trial log of the last optimize()
void cmlhost_tracker_set_eval(void* h, DSOTracker::EvalFn fn, void* user) { DSOTracker* t = static_cast<DSOTracker*>(h); t->evalOverride = fn; t->evalUser = user; }
int cmlhost_tracker_steps(void* h, int cap, int* level, int* iteration, int* accept, double* lambda) {
    const auto& s = static_cast<DSOTracker*>(h)->lastSteps;
    for (int i = 0; i < (int)s.size() && i < cap; i++) { level[i] = s[i].level; iteration[i] = s[i].iteration; accept[i] = s[i].accept; lambda[i] = s[i].lambda; }
    return (int)s.size();
}
void cmlhost_tracker_set_last_residual(void* h, int isCorrect, int levels, const double* rmse) {
    DSOTracker* t = static_cast<DSOTracker*>(h);
    t->mLastResidual = DSOTracker::Residual();
    t->mLastResidual.isCorrect = isCorrect != 0;
    t->mLastResidual.E.assign(levels, 0.0); t->mLastResidual.numTermsInE.assign(levels, 1);
    for (int l = 0; l < levels; l++) t->mLastResidual.E[l] = rmse[l];
}
// trackWithMotionModel(): hypotheses as n x {R[9], t[3]}; outputs the adopted try
int cmlhost_tracker_track_with_motion_model(void* h, uint64_t new_image, int levels, int n_hyp, const double* hypRt, const double refExp[3], const double initExp[3],
                                            double R[9], double t[3], double outExp[2], double* E, int* numTerms, int* numSat, int* isCorrect,
                                            int* tooManySaturated, int* winner, int* tries, double* lastCoarseRMSE, int batched) {
    DSOTracker* T = static_cast<DSOTracker*>(h);
    std::vector<SE3> hyp(n_hyp);
    for (int i = 0; i < n_hyp; i++) hyp[i] = SE3::fromRt(hypRt + 12 * i, hypRt + 12 * i + 9);
    Exposure ref(refExp[2], refExp[0], refExp[1]), init(initExp[2], initExp[0], initExp[1]), best = init;
    SE3 bestT;
    DSOTracker::Residual res;
    const bool ok = batched ? T->trackWithMotionModelBatched(new_image, levels, n_hyp, hyp.data(), ref, init, bestT, best, res, winner, tries)
                            : T->trackWithMotionModel(new_image, levels, n_hyp, hyp.data(), ref, init, bestT, best, res, winner, tries);
    if (ok) {
        bestT.matrix(R); std::memcpy(t, bestT.t, 3 * sizeof(double));
        outExp[0] = best.a; outExp[1] = best.b;
        for (int l = 0; l < levels && l < (int)res.E.size(); l++) { E[l] = res.E[l]; numTerms[l] = res.numTermsInE[l]; numSat[l] = res.numSaturated[l]; }
    }
    *isCorrect = res.isCorrect; *tooManySaturated = res.tooManySaturated; *lastCoarseRMSE = T->mLastCoarseRMSE;
    return ok ? 1 : 0;
}

// One tracked frame, one host wait (Hybrid.cpp:383-442: trackWithMotionModel, then traceNewCoarse against the pose it found): the hypothesis batch and,
// behind it on the same stream, the trace of the resident immature set against the batch's first result are enqueued together; the selection is replayed
// after the single wait.  kept = 1: the first try was adopted and the trace stands (counts / pairs_used are filled); kept = 0: another try won, tracking
// failed or fell back — every traced point was restored and the caller traces again with the pose it goes on with (cmlhost_tracer_trace).
// hosts: n_frames x {R[9], t[3], a, b} world -> camera and exposure of the window's keyframes (frame_ids order), ref_index the keyframe the
// hypotheses are relative to.
int cmlhost_frame_track_and_trace(void* trk, void* trc, uint64_t new_image, int levels, int n_hyp, const double* hypRt, const double refExp[3], const double initExp[3],
                                  int traced_frame_id, int n_frames, const int* frame_ids, const double* hosts, int ref_index, const double K[4],
                                  double R[9], double t[3], double outExp[2], double* E, int* numTerms, int* numSat, int* isCorrect, int* tooManySaturated,
                                  int* winner, int* tries, double* lastCoarseRMSE, int* kept, int counts[6], cmlhip_trace_pair* pairs_used) {
    DSOTracker* T = static_cast<DSOTracker*>(trk);
    cml_amd::DSOTracer* Tr = static_cast<cml_amd::DSOTracer*>(trc);
    *kept = 0;
    if (n_frames < 1 || ref_index < 0 || ref_index >= n_frames) return -1;
    std::vector<SE3> hyp(n_hyp);
    for (int i = 0; i < n_hyp; i++) hyp[i] = SE3::fromRt(hypRt + 12 * i, hypRt + 12 * i + 9);
    Exposure ref(refExp[2], refExp[0], refExp[1]), init(initExp[2], initExp[0], initExp[1]), best = init;
    std::vector<int> ids(frame_ids, frame_ids + n_frames);
    std::vector<cmlhip_frame_pose> hp(n_frames);
    for (int h = 0; h < n_frames; h++) {
        std::memcpy(hp[h].R, hosts + 14 * (size_t)h, 9 * sizeof(double)); std::memcpy(hp[h].t, hosts + 14 * (size_t)h + 9, 3 * sizeof(double));
        hp[h].a = hosts[14 * (size_t)h + 12]; hp[h].b = hosts[14 * (size_t)h + 13];
    }
    static const bool timing = cml_amd::HostLap::modeOf() == 1;
    const auto T0 = std::chrono::steady_clock::now();
    auto us = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - T0).count(); };
    if (!Tr->prepareTracked(hp, hp[ref_index], K)) return -1;                        // the batch's launch carries the trace's window: pairs formed at its tail
    if (!T->trackWithMotionModelBatchedEnqueue(new_image, levels, n_hyp, hyp.data(), ref, init)) return -1;
    const double t_a = us();
    const bool traced = Tr->traceNewCoarseTrackedAsync(new_image, traced_frame_id, ids, hp, hp[ref_index], K);
    const double t_b = us();
    SE3 bestT;
    DSOTracker::Residual res;
    bool retried = false;
    const bool ok = T->trackWithMotionModelBatchedFinish(bestT, best, res, winner, tries, &retried);      // the ONE wait of the frame
    if (timing) fprintf(stderr, "  [frame] tracker enqueued %.0f us | trace enqueued %.0f | waited + replayed %.0f\n", t_a, t_b, us());
    if (ok) {
        bestT.matrix(R); std::memcpy(t, bestT.t, 3 * sizeof(double));
        outExp[0] = best.a; outExp[1] = best.b;
        for (int l = 0; l < levels && l < (int)res.E.size(); l++) { E[l] = res.E[l]; numTerms[l] = res.numTermsInE[l]; numSat[l] = res.numSaturated[l]; }
    }
    *isCorrect = res.isCorrect; *tooManySaturated = res.tooManySaturated; *lastCoarseRMSE = T->mLastCoarseRMSE;
    if (traced) {
        const bool keep = ok && *winner == 0 && !retried;
        std::vector<cmlhip_trace_pair> pr;
        if (!Tr->finishTracked(keep, counts, &pr)) return -1;
        if (keep) { *kept = 1; if (pairs_used) std::memcpy(pairs_used, pr.data(), sizeof(cmlhip_trace_pair) * pr.size()); }
    } else return -1;
    return ok ? 1 : 0;
}

// ---- DSOTracer mirror
void* cmlhost_tracer_create(cmlhip_ctx* ctx) { return new cml_amd::DSOTracer(ctx); }
void cmlhost_tracer_destroy(void* h) { delete static_cast<cml_amd::DSOTracer*>(h); }
int cmlhost_tracer_add_point(void* h, float x, float y, int host_frame_id, const float gray[8], const float dpatch[24], const double gradH[4], float type) {
    return static_cast<cml_amd::DSOTracer*>(h)->addImmaturePoint(x, y, host_frame_id, gray, dpatch, gradH, type);
}
// n points of one keyframe at once (makeNewTraces): xy n x 2, gray n x 8, dpatch n x 24, gradH n x 4; returns the index of the first
int cmlhost_tracer_add_points(void* h, int n, const float* xy, int host_frame_id, const float* gray, const float* dpatch, const double* gradH) {
    cml_amd::DSOTracer* t = static_cast<cml_amd::DSOTracer*>(h);
    const int first = (int)t->size();
    for (int i = 0; i < n; i++) t->addImmaturePoint(xy[2 * i], xy[2 * i + 1], host_frame_id, gray + 8 * (size_t)i, dpatch + 24 * (size_t)i, gradH + 4 * (size_t)i, 1.f);
    return first;
}
int cmlhost_tracer_prepare_resident(void* h, int n_frames, const int* frame_ids) {
    std::vector<int> ids(frame_ids, frame_ids + n_frames);
    return static_cast<cml_amd::DSOTracer*>(h)->prepareResident(ids) ? 1 : 0;
}
void cmlhost_tracer_compact(void* h) { static_cast<cml_amd::DSOTracer*>(h)->compact(); }
void cmlhost_tracer_get_frame_ids(void* h, int* out) {
    auto& P = static_cast<cml_amd::DSOTracer*>(h)->peek();
    for (size_t i = 0; i < P.size(); i++) out[i] = P[i].frame_id;
}
int cmlhost_tracer_trace(void* h, uint64_t image_id, int traced_frame_id, int n_frames, const int* frame_ids, const cmlhip_trace_pair* pairs, int counts[6]) {
    std::vector<int> ids(frame_ids, frame_ids + n_frames);
    std::vector<cmlhip_trace_pair> pr(pairs, pairs + n_frames);
    return static_cast<cml_amd::DSOTracer*>(h)->traceNewCoarse(image_id, traced_frame_id, ids, pr, counts) ? 1 : 0;
}
int cmlhost_tracer_activate(void* h, int n_frames, const int* frame_ids, const uint64_t* image_ids, const double K[4], int w, int hgt,
                            const cmlhip_activation_pair* pairs, int* activated, int cap) {
    std::vector<int> ids(frame_ids, frame_ids + n_frames);
    std::vector<uint64_t> im(image_ids, image_ids + n_frames);
    std::vector<cmlhip_activation_pair> pr(pairs, pairs + (size_t)n_frames * n_frames);
    std::vector<int> act;
    if (!static_cast<cml_amd::DSOTracer*>(h)->activatePoints(ids, im, K, w, hgt, pr, act)) return -1;
    for (size_t i = 0; i < act.size() && (int)i < cap; i++) activated[i] = act[i];
    return (int)act.size();
}
int cmlhost_tracer_count(void* h) { return (int)static_cast<cml_amd::DSOTracer*>(h)->size(); }
// DSOTracer::activatePoints hands its activated points to BA::addPoints (DSOTracer.cpp:199-210 -> BA.cpp:343-415): pixel, activated inverse depth,
// host keyframe (window index of the point's frame id), the gray patch as colours, gradient weights sqrt(c / (c + |grad|^2)), c = 50^2 (BA.cpp:405-411).
// xy_out (n x 2 ints, optional): the pixels handed over (the caller's pixel selector keeps them occupied).  Returns the first BA point index or -1.
int cmlhost_tracer_add_activated_to_ba(void* tr, void* ba, int n, const int* idx, int n_frames, const int* frame_ids, int* xy_out) {
    auto& P = static_cast<cml_amd::DSOTracer*>(tr)->points();
    DSOBundleAdjustment* b = static_cast<DSOBundleAdjustment*>(ba);
    const int first = (int)b->getPoints().size();
    for (int k = 0; k < n; k++) {
        if (idx[k] < 0 || idx[k] >= (int)P.size()) return -1;
        const auto& q = P[(size_t)idx[k]];
        int host = -1;
        for (int f = 0; f < n_frames; f++) if (frame_ids[f] == q.frame_id) { host = f; break; }
        if (host < 0) return -1;
        float w[8];
        for (int j = 0; j < 8; j++) {
            const double gx = (double)q.d.dpatch[3 * j + 1], gy = (double)q.d.dpatch[3 * j + 2];
            w[j] = (float)std::sqrt(2500.0 / (2500.0 + (gx * gx + gy * gy)));
        }
        b->addPoint(q.d.x, q.d.y, (double)q.idepth, host, q.d.gray, w, false);
        if (xy_out) { xy_out[2 * k] = (int)q.d.x; xy_out[2 * k + 1] = (int)q.d.y; }
    }
    b->handOverNewEntries();                                 // the library's window gets the batch now (BA::addPoints builds its residuals here, not in run())
    return first;
}
// immature points still alive per frame id (what flagFramesForMarginalization weighs, BA.cpp:428-462 via DSOContext's per-frame groups)
void cmlhost_tracer_immature_counts(void* h, int n_frames, const int* frame_ids, int* counts) {
    auto& P = static_cast<cml_amd::DSOTracer*>(h)->peek();
    for (int k = 0; k < n_frames; k++) counts[k] = 0;
    for (size_t i = 0; i < P.size(); i++) {
        if (!P[i].alive || P[i].activated) continue;
        for (int k = 0; k < n_frames; k++) if (frame_ids[k] == P[i].frame_id) { counts[k]++; break; }
    }
}
void cmlhost_tracer_get_points(void* h, cmlhip_immature_point* out, unsigned char* alive, unsigned char* activated, float* idepth) {
    auto& P = static_cast<cml_amd::DSOTracer*>(h)->points();
    for (size_t i = 0; i < P.size(); i++) { out[i] = P[i].d; alive[i] = P[i].alive; activated[i] = P[i].activated; idepth[i] = P[i].idepth; }
}
const char* cmlhost_tracer_last_error(void* h) { return static_cast<cml_amd::DSOTracer*>(h)->lastError().c_str(); }


// ---- IndirectCameraOptimizer / IndirectBundleAdjustment mirrors (flat records for ctypes)
struct cmlhost_matching { int has_map_point; int level; double X[3]; double obs[2]; double scale_factor_base; double descriptor_distance; };
struct cmlhost_lba_frame { int id; int pad; double R[9], t[3], K[4]; };
struct cmlhost_lba_point { int id; int reference_frame_id; double X[3]; };
struct cmlhost_lba_apparition { int point; int frame_id; double obs[2]; int level; int pad; double scale_factor_base; };

static std::vector<cml_amd::IndirectCameraOptimizer::Matching> to_matchings(int n, const cmlhost_matching* m) {
    std::vector<cml_amd::IndirectCameraOptimizer::Matching> v((size_t)n);
    for (int i = 0; i < n; i++) {
        v[i].hasMapPoint = m[i].has_map_point != 0; v[i].level = m[i].level; v[i].scaleFactorBase = m[i].scale_factor_base;
        v[i].descriptorDistance = m[i].descriptor_distance;
        for (int k = 0; k < 3; k++) v[i].X[k] = m[i].X[k];
        v[i].obs[0] = m[i].obs[0]; v[i].obs[1] = m[i].obs[1];
    }
    return v;
}
static void put_result(const cml_amd::IndirectCameraOptimizerResult& r, int* is_ok, double R[9], double t[3], double cov[6]) {
    *is_ok = r.isOk ? 1 : 0;
    for (int k = 0; k < 9; k++) R[k] = r.R[k];
    for (int k = 0; k < 3; k++) t[k] = r.t[k];
    for (int k = 0; k < 6; k++) cov[k] = r.covariance[k];
}
int cmlhost_pnp_optimize(cmlhip_ctx* ctx, int check_outliers, const double frameR[9], const double frameT[3], const double* cameraR, const double* cameraT,
                         const double K[4], int n, const cmlhost_matching* m, unsigned char* outliers, int n_outliers_in, int compute_cov,
                         int* is_ok, double R[9], double t[3], double cov[6]) {
    cml_amd::IndirectCameraOptimizer opt(ctx);
    opt.mCheckOutliers = check_outliers != 0;
    std::vector<bool> out((size_t)n_outliers_in);
    for (int i = 0; i < n_outliers_in; i++) out[i] = outliers[i] != 0;
    const auto r = opt.optimize(frameR, frameT, cameraR, cameraT, K, to_matchings(n, m), out, compute_cov != 0);
    for (int i = 0; i < n; i++) outliers[i] = out[i] ? 1 : 0;
    put_result(r, is_ok, R, t, cov);
    return opt.lastError().empty() ? 0 : 1;
}
int cmlhost_pnp_optimize_points(cmlhip_ctx* ctx, int check_outliers, const double frameR[9], const double frameT[3], const double K[4], int n,
                                const cmlhost_matching* m, int* outlier_idx, int* n_outlier_idx, int compute_cov, int* is_ok, double R[9], double t[3], double cov[6]) {
    cml_amd::IndirectCameraOptimizer opt(ctx);
    opt.mCheckOutliers = check_outliers != 0;
    std::vector<int> idx;
    const auto r = opt.optimize(frameR, frameT, K, to_matchings(n, m), idx, compute_cov != 0);
    for (size_t i = 0; i < idx.size(); i++) outlier_idx[i] = idx[i];
    *n_outlier_idx = (int)idx.size();
    put_result(r, is_ok, R, t, cov);
    return opt.lastError().empty() ? 0 : 1;
}
void* cmlhost_lba_create(cmlhip_ctx* ctx) { return new cml_amd::IndirectBundleAdjustment(ctx); }
void cmlhost_lba_destroy(void* h) { delete static_cast<cml_amd::IndirectBundleAdjustment*>(h); }
void cmlhost_lba_set_params(void* h, int num_iteration, int refine_iteration, int remove_edge) {
    auto* b = static_cast<cml_amd::IndirectBundleAdjustment*>(h);
    b->mNumIteration = num_iteration; b->mRefineIteration = refine_iteration; b->mRemoveEdge = remove_edge != 0;
}
static std::vector<cml_amd::IndirectBundleAdjustment::Frame> to_frames(int n, const cmlhost_lba_frame* f) {
    std::vector<cml_amd::IndirectBundleAdjustment::Frame> v((size_t)n);
    for (int i = 0; i < n; i++) {
        v[i].id = f[i].id;
        for (int k = 0; k < 9; k++) v[i].R[k] = f[i].R[k];
        for (int k = 0; k < 3; k++) v[i].t[k] = f[i].t[k];
        for (int k = 0; k < 4; k++) v[i].K[k] = f[i].K[k];
    }
    return v;
}
int cmlhost_lba_local_optimize(void* h, int n_local, const cmlhost_lba_frame* local, int n_fixed, const cmlhost_lba_frame* fixed, int n_points,
                               const cmlhost_lba_point* points, int n_app, const cmlhost_lba_apparition* app, int fix_frames,
                               unsigned char* stop_flag /* the reference's pbStopFlag for THIS call (IndirectBundleAdjustment.h:27), may be null */) {
    std::vector<cml_amd::IndirectBundleAdjustment::Point> P((size_t)n_points);
    for (int i = 0; i < n_points; i++) { P[i].id = points[i].id; P[i].referenceFrameId = points[i].reference_frame_id; for (int k = 0; k < 3; k++) P[i].X[k] = points[i].X[k]; }
    for (int a = 0; a < n_app; a++) {
        cml_amd::IndirectBundleAdjustment::Apparition A;
        A.frameId = app[a].frame_id; A.obs[0] = app[a].obs[0]; A.obs[1] = app[a].obs[1]; A.level = app[a].level; A.scaleFactorBase = app[a].scale_factor_base;
        P[(size_t)app[a].point].apparitions.push_back(A);
    }
    return static_cast<cml_amd::IndirectBundleAdjustment*>(h)->localOptimize(to_frames(n_local, local), to_frames(n_fixed, fixed), P, fix_frames != 0,
                                                                             reinterpret_cast<bool*>(stop_flag)) ? 1 : 0;
}
int cmlhost_lba_apply(void* h, int n_local, cmlhost_lba_frame* local_out, int n_points, double* X_out, int* removals, int cap, cmlhip_lba_result* res) {
    std::vector<cml_amd::IndirectBundleAdjustment::Frame> F; std::vector<cml_amd::IndirectBundleAdjustment::Point> P;
    std::vector<cml_amd::IndirectBundleAdjustment::Removal> Rm;
    auto* b = static_cast<cml_amd::IndirectBundleAdjustment*>(h);
    b->apply(F, P, Rm);
    for (int i = 0; i < n_local && i < (int)F.size(); i++) {
        local_out[i].id = F[i].id;
        for (int k = 0; k < 9; k++) local_out[i].R[k] = F[i].R[k];
        for (int k = 0; k < 3; k++) local_out[i].t[k] = F[i].t[k];
        for (int k = 0; k < 4; k++) local_out[i].K[k] = F[i].K[k];
    }
    for (int i = 0; i < n_points && i < (int)P.size(); i++) for (int k = 0; k < 3; k++) X_out[3 * i + k] = P[i].X[k];
    for (size_t i = 0; i < Rm.size() && (int)i < cap; i++) { removals[2 * i] = Rm[i].frameId; removals[2 * i + 1] = Rm[i].pointId; }
    if (res) *res = b->result();
    return (int)Rm.size();
}
const char* cmlhost_lba_last_error(void* h) { return static_cast<cml_amd::IndirectBundleAdjustment*>(h)->lastError().c_str(); }


// ---- DSOInitializer mirror
static cml_amd::SE3 se3_from_qt(const double qt[7]) { cml_amd::SE3 T; for (int k = 0; k < 4; k++) T.q[k] = qt[k]; for (int k = 0; k < 3; k++) T.t[k] = qt[4 + k]; return T; }
void* cmlhost_init_create(cmlhip_ctx* ctx) { return new cml_amd::DSOInitializer(ctx); }
void cmlhost_init_destroy(void* h) { delete static_cast<cml_amd::DSOInitializer*>(h); }
int cmlhost_init_set_first(void* h, int n_levels, const int* w, const int* hgt, const double* K4, const float* const* gray, const int* n_px,
                           const int* px, const int* py, const double ref_qt[7], double ref_exposure) {
    std::vector<cml_amd::DSOInitializer::LevelInput> L((size_t)n_levels);
    size_t at = 0;
    for (int l = 0; l < n_levels; l++) {
        L[l].w = w[l]; L[l].h = hgt[l]; for (int k = 0; k < 4; k++) L[l].K[k] = K4[4 * l + k];
        L[l].gray = gray[l];
        L[l].px.assign(px + at, px + at + n_px[l]); L[l].py.assign(py + at, py + at + n_px[l]);
        at += (size_t)n_px[l];
    }
    return static_cast<cml_amd::DSOInitializer*>(h)->setFirst(L, se3_from_qt(ref_qt), ref_exposure) ? 1 : 0;
}
int cmlhost_init_try(void* h, uint64_t image_id, const double frame_qt[7], double exposure) {
    return static_cast<cml_amd::DSOInitializer*>(h)->tryInitialize(image_id, se3_from_qt(frame_qt), exposure);
}
void cmlhost_init_state(void* h, double cur_qt[7], int* snapped, int* frame_id, int counts[3], float* rescale) {
    auto* I = static_cast<cml_amd::DSOInitializer*>(h);
    const cml_amd::SE3& T = I->currentCamera();
    for (int k = 0; k < 4; k++) cur_qt[k] = T.q[k];
    for (int k = 0; k < 3; k++) cur_qt[4 + k] = T.t[k];
    *snapped = I->snapped() ? 1 : 0; *frame_id = I->frameID();
    counts[0] = I->numCalcCalls; counts[1] = I->numAccepted; counts[2] = I->numRejected;
    *rescale = I->rescaleFactor();
}
int cmlhost_init_level_size(void* h, int lvl) { return (int)static_cast<cml_amd::DSOInitializer*>(h)->points(lvl).size(); }
void cmlhost_init_get_points(void* h, int lvl, float* xy, float* iR, float* idepth, unsigned char* good, float* last_hessian, int* parent, int* neighbours) {
    const auto& P = static_cast<cml_amd::DSOInitializer*>(h)->points(lvl);
    for (size_t i = 0; i < P.size(); i++) {
        xy[2 * i] = P[i].px; xy[2 * i + 1] = P[i].py; iR[i] = P[i].d.iR; idepth[i] = P[i].idepth; good[i] = P[i].d.is_good ? 1 : 0;
        last_hessian[i] = P[i].lastHessian; parent[i] = P[i].parent;
        for (int k = 0; k < 10; k++) neighbours[10 * i + k] = P[i].neighbours[k];
    }
}
const char* cmlhost_init_last_error(void* h) { return static_cast<cml_amd::DSOInitializer*>(h)->lastError().c_str(); }

}  // extern "C"
