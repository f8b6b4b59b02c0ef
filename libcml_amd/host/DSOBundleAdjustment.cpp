// DSOBundleAdjustment.cpp — host mirror of CML::Optimization::DSOBundleAdjustment over the C ABI.
// BA.cpp = src/cml/optimization/dso/DSOBundleAdjustment.cpp in the reference tree.
#include "DSOBundleAdjustment.h"
#include "HostLap.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstdio>

namespace cml_amd {


// ------------------------------------------------------------------------------------------------ DSOFrame
static void updatePRE(DSOFrame& f) {                                           // DSOFrame.h:119-120
    f.PRE_worldToCam = SE3::exp(f.state_scaled) * f.worldToCam_evalPT;
    f.PRE_camToWorld = f.PRE_worldToCam.inverse();
}
void DSOFrame::setState(const double s[10], const double sc[4]) {             // DSOFrame.h:110-124
    double tmp[10];
    std::memcpy(tmp, s, sizeof tmp);
    std::memcpy(state, tmp, sizeof tmp);
    for (int i = 0; i < 3; i++) { state_scaled[i] = sc[0] * tmp[i]; state_scaled[3 + i] = sc[1] * tmp[3 + i]; }
    state_scaled[6] = sc[2] * tmp[6]; state_scaled[7] = sc[3] * tmp[7]; state_scaled[8] = sc[2] * tmp[8]; state_scaled[9] = sc[3] * tmp[9];
    updatePRE(*this);
}
void DSOFrame::setStateScaled(const double ss[10], const double sc[4]) {      // DSOFrame.h:126-142
    double tmp[10];
    std::memcpy(tmp, ss, sizeof tmp);
    std::memcpy(state_scaled, tmp, sizeof tmp);
    for (int i = 0; i < 3; i++) { state[i] = tmp[i] / sc[0]; state[3 + i] = tmp[3 + i] / sc[1]; }
    state[6] = tmp[6] / sc[2]; state[7] = tmp[7] / sc[3]; state[8] = tmp[8] / sc[2]; state[9] = tmp[9] / sc[3];
    updatePRE(*this);
}
void DSOFrame::setStateZero(const double sz[10], const double sc[4]) {        // DSOFrame.h:154-186
    double tmp[10];
    std::memcpy(tmp, sz, sizeof tmp);
    std::memcpy(state_zero, tmp, sizeof tmp);
    const SE3 Ti = worldToCam_evalPT.inverse();
    for (int i = 0; i < 6; i++) {
        double eps[6] = {0, 0, 0, 0, 0, 0}, lp[6], lm[6];
        eps[i] = 1e-3;
        const SE3 Ep = SE3::exp(eps);
        eps[i] = -1e-3;
        const SE3 Em = SE3::exp(eps);
        ((worldToCam_evalPT * Ep) * Ti).log(lp);
        ((worldToCam_evalPT * Em) * Ti).log(lm);
        for (int k = 0; k < 6; k++) nullspaces_pose[i * 6 + k] = (lp[k] - lm[k]) / (2e-3);
    }
    SE3 Pp = worldToCam_evalPT, Pm = worldToCam_evalPT;
    for (int k = 0; k < 3; k++) { Pp.t[k] *= 1.00001; Pm.t[k] /= 1.00001; }
    double lp[6], lm[6];
    (Pp * Ti).log(lp);
    (Pm * Ti).log(lm);
    for (int k = 0; k < 6; k++) nullspaces_scale[k] = (lp[k] - lm[k]) / (2e-3);
    std::memset(nullspaces_affine, 0, sizeof nullspaces_affine);
    nullspaces_affine[0] = 1;
    nullspaces_affine[4 + 1] = (double)std::exp((float)(state_zero[6] * sc[2])) * ab_exposure;
}
void DSOFrame::setEvalPT(const SE3& w2c, const double s[10], const double sc[4]) {     // DSOFrame.h:88-95
    worldToCam_evalPT = w2c;
    setState(s, sc);
    setStateZero(s, sc);
}
void DSOFrame::setEvalPT_scaled(const SE3& w2c, const Exposure& aff, const double sc[4]) {   // DSOFrame.h:99-108
    double init[10] = {0, 0, 0, 0, 0, 0, aff.a, aff.b, 0, 0};
    worldToCam_evalPT = w2c;
    setStateScaled(init, sc);
    setStateZero(state, sc);
}
void DSOFrame::doStepFromBackup(const double sc[4]) {                         // DSOFrame.h:82-84
    double s[10];
    for (int i = 0; i < 10; i++) s[i] = state_backup[i] + step[i];
    setState(s, sc);
}
void DSOFrame::setStep(const double s[10]) {                                  // DSOFrame.h:205-214
    for (int i = 0; i < 10; i++)
        if (!std::isfinite(s[i])) { std::memset(step, 0, sizeof step); return; }
    std::memcpy(step, s, sizeof step);
}

// ------------------------------------------------------------------------------------------------ BA
DSOBundleAdjustment::DSOBundleAdjustment(cmlhip_ctx* ctx) : mCtx(ctx) {
    mMarginalizedHessian.assign((CMLHIP_CPARS + 8) * (CMLHIP_CPARS + 8), 0.0);      // BA.cpp:323-327
    mMarginalizedB.assign(CMLHIP_CPARS + 8, 0.0);
}

bool DSOBundleAdjustment::fail(const std::string& what, int rc) {
    mError = what + " (status " + std::to_string(rc) + "): " + (mCtx ? cmlhip_last_error(mCtx) : "no context");
    return false;
}

void DSOBundleAdjustment::setCalibration(double fx, double fy, double cx, double cy, int w, int h) {
    mPrm.fx = fx; mPrm.fy = fy; mPrm.cx = cx; mPrm.cy = cy; mPrm.w = w; mPrm.h = h;
    mHaveCalib = true;
}

// The library keeps the window between keyframes (cmlhip_ba_window_*, include/cmlhip.h) index for index with mPoints / mResiduals: entries this
// object has not handed over yet are appended, in list order.  A window the library no longer holds (somebody uploaded another one through the
// context) is handed over again from the start.
bool DSOBundleAdjustment::syncWindowAppends() {
    int wp = 0, wr = 0;
    int rc = cmlhip_ba_window_counts(mCtx, &wp, &wr);
    if (rc) return fail("cmlhip_ba_window_counts", rc);
    unsigned gen = 0;
    if ((rc = cmlhip_ba_window_generation(mCtx, &gen))) return fail("cmlhip_ba_window_generation", rc);
    // (the sizes alone do not say whose entries they are: another HostBA on this context, a checker replay or a direct cmlhip_ba_upload_window
    //  may have left a window of the same size — the generation number changes with every reset)
    if (gen != mWinGeneration || wp != (int)mWinPoints || wr != (int)mWinResiduals || mWinPoints > mPoints.size() || mWinResiduals > mResiduals.size()) {
        if ((rc = cmlhip_ba_window_reset(mCtx))) return fail("cmlhip_ba_window_reset", rc);
        if ((rc = cmlhip_ba_window_generation(mCtx, &mWinGeneration))) return fail("cmlhip_ba_window_generation", rc);
        mWinPoints = mWinResiduals = 0;
    }
    if (mWinPoints < mPoints.size()) {
        std::vector<cmlhip_ba_point>& pts = mUploadPoints;
        pts.resize(mPoints.size() - mWinPoints);
        for (size_t p = mWinPoints; p < mPoints.size(); p++) {
            const DSOPoint& P = mPoints[p];
            cmlhip_ba_point& q = pts[p - mWinPoints];
            q.x = P.x; q.y = P.y; q.idepth = P.idepth; q.idepth_zero = P.idepth_zero; q.prior = P.priorF; q.host = P.host;
            std::memcpy(q.colors, P.colors, sizeof q.colors);
            std::memcpy(q.weights, P.weights, sizeof q.weights);
        }
        if ((rc = cmlhip_ba_window_append_points(mCtx, (int)pts.size(), pts.data()))) return fail("cmlhip_ba_window_append_points", rc);
        mWinPoints = mPoints.size();
    }
    if (mWinResiduals < mResiduals.size()) {
        std::vector<cmlhip_ba_residual>& rs = mUploadResiduals;
        rs.resize(mResiduals.size() - mWinResiduals);
        for (size_t r = mWinResiduals; r < mResiduals.size(); r++) {
            const DSOResidual& R = mResiduals[r];
            rs[r - mWinResiduals] = cmlhip_ba_residual{R.point < 0 ? 0 : R.point, R.target, R.state_state, R.isLinearized ? 1 : 0};
        }
        if ((rc = cmlhip_ba_window_append_residuals(mCtx, (int)rs.size(), rs.data()))) return fail("cmlhip_ba_window_append_residuals", rc);
        mWinResiduals = mResiduals.size();
    }
    return true;
}

void DSOBundleAdjustment::compactDead() {
    mOutliers.clear(); mActive.clear(); mActivePoints.clear();
    if (mDeadSinceCompact == 0) { mPointSlot.assign(mPoints.size(), -1); return; }      // nothing was dropped since the lists were last renumbered
    // Round 6: the lists are compacted IN PLACE (stable) and the per-point residual lists keep their storage — copying 15 000 72-byte residuals and 2 700
    // points into fresh vectors and rebuilding a vector<vector<int>> (one allocation per point) was 120 of addNewFrame's 165 us at the sliding window.
    const size_t P0 = mPoints.size(), R0 = mResiduals.size();
    HostLap lap("compactDead");
    std::vector<int>& pmap = mCompactPmap; std::vector<int>& rmap = mCompactRmap;
    std::vector<unsigned char>& pAlive = mCompactPAlive; std::vector<unsigned char>& rAlive = mCompactRAlive;
    pmap.assign(P0, -1); rmap.assign(R0, -1); pAlive.assign(P0, 0); rAlive.assign(R0, 0);
    size_t np = 0;
    for (size_t p = 0; p < P0; p++) if (mPoints[p].alive) { pmap[p] = (int)np; pAlive[p] = 1; np++; }
    mLinearizedAlive = 0;
    size_t nr = 0;
    for (size_t r = 0; r < R0; r++) {
        const DSOResidual& R = mResiduals[r];
        if (!R.alive || R.point < 0 || pmap[R.point] < 0) continue;
        rmap[r] = (int)nr; rAlive[r] = 1; nr++;
        mLinearizedAlive += R.isLinearized;
    }
    lap("maps");
    // the library's copy of the window is renumbered the same way (everything appended first, so that the lists have the same length)
    if (syncWindowAppends()) {
        lap("appends synced");
        const int rc = cmlhip_ba_window_compact(mCtx, (int)pAlive.size(), pAlive.data(), (int)rAlive.size(), rAlive.data());
        if (rc) { cmlhip_ba_window_reset(mCtx); mWinPoints = mWinResiduals = 0; }        // (handed over again from the start by the next run)
        else { mWinPoints = np; mWinResiduals = nr; }
    } else { cmlhip_ba_window_reset(mCtx); mWinPoints = mWinResiduals = 0; mError.clear(); }
    lap("window compacted");
    if (mPointRes.size() < P0) mPointRes.resize(P0);
    for (size_t p = 0; p < P0; p++) {
        const int q = pmap[p];
        if (q < 0) continue;
        if ((size_t)q != p) { mPoints[q] = mPoints[p]; mPointRes[q].swap(mPointRes[p]); }
        mPointRes[q].clear();                                                  // (keeps its capacity: refilled below without allocating)
        for (int k = 0; k < 2; k++) mPoints[q].lastResidual[k] = mPoints[q].lastResidual[k] >= 0 ? rmap[mPoints[q].lastResidual[k]] : -1;
    }
    mPoints.resize(np);
    for (size_t q = np; q < mPointRes.size(); q++) mPointRes[q].clear();      // (the dropped points' lists keep their storage for the points to come: no shrink)
    for (size_t r = 0; r < R0; r++) {
        const int q = rmap[r];
        if (q < 0) continue;
        if ((size_t)q != r) mResiduals[q] = mResiduals[r];
        mResiduals[q].point = pmap[mResiduals[q].point];
        mPointRes[mResiduals[q].point].push_back(q);
    }
    mResiduals.resize(nr);
    mPointSlot.assign(mPoints.size(), -1);
    mDeadSinceCompact = 0;
    lap("lists compacted");
}

int DSOBundleAdjustment::addNewFrame(uint64_t image_id, const SE3& worldToCam, const Exposure& exposure) {
    double sc[4];
    scales(sc);
    HostLap lap("addNewFrame");
    compactDead();                                                  // (between keyframes nothing refers to point / residual indices)
    lap("compactDead");
    DSOFrame f;
    f.id = (int)mFrames.size();
    f.keyid = mFrameKeyCounter++;                                   // DSOContext.h:49-50
    f.image_id = image_id;
    f.ab_exposure = exposure.t;
    f.setEvalPT_scaled(worldToCam, exposure, sc);                   // BA.cpp:434
    mFrames.push_back(f);
    // grow the marginalisation prior by one 8-block of zeros, BA.cpp:439-443
    const int n = 8 * (int)mFrames.size() + CMLHIP_CPARS, o = n - 8;
    std::vector<double> H((size_t)n * n, 0.0), b(n, 0.0);
    for (int i = 0; i < o && (size_t)o * o == mMarginalizedHessian.size(); i++) {
        for (int j = 0; j < o; j++) H[(size_t)i * n + j] = mMarginalizedHessian[(size_t)i * o + j];
        b[i] = mMarginalizedB[i];
    }
    mMarginalizedHessian.swap(H);
    mMarginalizedB.swap(b);
    computeAdjoints();
    computeDelta();
    lap("prior+adjoints+delta");
    // residuals of the existing points into the new frame, BA.cpp:456-460 (createResidual, :336-380)
    const int t = f.id;
    std::vector<SE3> relT(mFrames.size());                             // host -> new frame at the evaluation points, once per host (not per point)
    std::vector<double> relR(9 * mFrames.size());
    for (int h = 0; h < (int)mFrames.size(); h++) { relT[h] = mFrames[t].worldToCam_evalPT * mFrames[h].worldToCam_evalPT.inverse(); relT[h].matrix(&relR[9 * h]); }
    for (int p = 0; p < (int)mPoints.size(); p++) {
        if (!mPoints[p].alive || mPoints[p].host == t) continue;
        const DSOPoint& P = mPoints[p];
        const SE3& ht = relT[P.host];
        const double* R = &relR[9 * P.host];
        const double rx = ((double)P.x - mPrm.cx) * (1.0 / mPrm.fx), ry = ((double)P.y - mPrm.cy) * (1.0 / mPrm.fy);
        const double px = R[0] * rx + R[1] * ry + R[2] + ht.t[0] * P.idepth, py = R[3] * rx + R[4] * ry + R[5] + ht.t[1] * P.idepth,
                     pz = R[6] * rx + R[7] * ry + R[8] + ht.t[2] * P.idepth;
        const double Ku = (px / pz) * mPrm.fx + mPrm.cx, Kv = (py / pz) * mPrm.fy + mPrm.cy;
        DSOResidual r;
        r.point = p; r.target = t;
        r.centerProjectedTo[0] = (float)Ku; r.centerProjectedTo[1] = (float)Kv; r.centerProjectedTo[2] = (float)((1.0 / pz) * P.idepth);
        const bool inside = Ku >= 0 && Kv >= 0 && Ku < mPrm.w && Kv < mPrm.h;     // Frame::isInside(p, 0, 0), src/cml/map/Frame.h:136-138
        r.state_state = inside ? DSORES_IN : DSORES_OOB;
        r.state_NewState = DSORES_OUTLIER;
        mResiduals.push_back(r);
        mPointRes[p].push_back((int)mResiduals.size() - 1);
        mPoints[p].lastResidual[0] = (int)mResiduals.size() - 1;            // target is getFrames().back(), BA.cpp:374-375
        mPoints[p].lastResidualState[0] = r.state_state;
    }
    lap("residuals created");
    handOverNewEntries();
    lap("handed over");
    return f.id;
}

int DSOBundleAdjustment::addPoint(float x, float y, double idepth, int host, const float colors[8], const float weights[8],
                                  bool hasDepthPrior) {
    DSOPoint P;
    P.x = x; P.y = y; P.idepth = idepth; P.host = host;
    std::memcpy(P.colors, colors, sizeof P.colors);
    std::memcpy(P.weights, weights, sizeof P.weights);
    P.idepth_zero = (float)idepth;                                   // BA.cpp:394
    P.hasDepthPrior = hasDepthPrior;
    const int p = (int)mPoints.size();
    mPoints.push_back(P);
    if (mPointRes.size() < mPoints.size()) mPointRes.resize(mPoints.size());
    mPointRes[p].clear();
    if (mPointRes[p].capacity() < 8) mPointRes[p].reserve(8);                  // one allocation for a point's life (it grew 1 -> 2 -> 4 -> 8)
    const int NF = (int)mFrames.size();
    if (mRelValidFor != NF) {                                        // host -> target at the evaluation points for every pair, rebuilt when the window changed
        mRelT.assign((size_t)NF * NF, SE3()); mRelR.assign(9 * (size_t)NF * NF, 0.0);
        for (int h = 0; h < NF; h++) {
            const SE3 hi = mFrames[h].worldToCam_evalPT.inverse();
            for (int t2 = 0; t2 < NF; t2++) { mRelT[(size_t)h * NF + t2] = mFrames[t2].worldToCam_evalPT * hi; mRelT[(size_t)h * NF + t2].matrix(&mRelR[9 * ((size_t)h * NF + t2)]); }
        }
        mRelValidFor = NF;
    }
    for (int t = 0; t < NF; t++) {                                   // BA.cpp:398-400
        if (t == host) continue;
        const SE3& ht = mRelT[(size_t)host * NF + t];
        const double* R = &mRelR[9 * ((size_t)host * NF + t)];
        const double rx = ((double)x - mPrm.cx) * (1.0 / mPrm.fx), ry = ((double)y - mPrm.cy) * (1.0 / mPrm.fy);
        const double px = R[0] * rx + R[1] * ry + R[2] + ht.t[0] * idepth, py = R[3] * rx + R[4] * ry + R[5] + ht.t[1] * idepth,
                     pz = R[6] * rx + R[7] * ry + R[8] + ht.t[2] * idepth;
        const double Ku = (px / pz) * mPrm.fx + mPrm.cx, Kv = (py / pz) * mPrm.fy + mPrm.cy;
        DSOResidual r;
        r.point = p; r.target = t;
        r.centerProjectedTo[0] = (float)Ku; r.centerProjectedTo[1] = (float)Kv; r.centerProjectedTo[2] = (float)((1.0 / pz) * idepth);
        const bool inside = Ku >= 0 && Kv >= 0 && Ku < mPrm.w && Kv < mPrm.h;
        r.state_state = inside ? DSORES_IN : DSORES_OOB;
        mResiduals.push_back(r);
        mPointRes[p].push_back((int)mResiduals.size() - 1);
        const int nf = (int)mFrames.size();
        if (t == nf - 1) { mPoints[p].lastResidual[0] = (int)mResiduals.size() - 1; mPoints[p].lastResidualState[0] = r.state_state; }        // BA.cpp:374-378
        else if (nf >= 2 && t == nf - 2) { mPoints[p].lastResidual[1] = (int)mResiduals.size() - 1; mPoints[p].lastResidualState[1] = r.state_state; }
    }
    return p;
}

void DSOBundleAdjustment::computeAdjoints() {                                 // BA.cpp:1030-1101
    const int N = (int)mFrames.size();
    mRelValidFor = -1;                                              // (called whenever frames or evaluation points changed: the addPoint table follows)
    double sc[4];
    scales(sc);
    mAdHost.assign((size_t)N * N * 64, 0.0);
    mAdTarget.assign((size_t)N * N * 64, 0.0);
    for (int h = 0; h < N; h++)
        for (int t = 0; t < N; t++) {
            const SE3 ht = mFrames[t].worldToCam_evalPT * mFrames[h].worldToCam_evalPT.inverse();
            double Adj[36], la, lb;
            ht.Adj(Adj);
            mFrames[h].aff_g2l_0(sc).to(mFrames[t].aff_g2l_0(sc), la, lb);
            double* AH = &mAdHost[64 * (size_t)(h + t * N)];
            double* AT = &mAdTarget[64 * (size_t)(h + t * N)];
            for (int i = 0; i < 6; i++) {
                for (int j = 0; j < 6; j++) AH[i * 8 + j] = -Adj[j * 6 + i];
                AT[i * 8 + i] = 1;
            }
            AT[6 * 8 + 6] = -la; AH[6 * 8 + 6] = la; AT[7 * 8 + 7] = -1; AH[7 * 8 + 7] = la;
            const double rs[8] = {sc[0], sc[0], sc[0], sc[1], sc[1], sc[1], sc[2], sc[3]};
            for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) { AH[i * 8 + j] *= rs[i]; AT[i * 8 + j] *= rs[i]; }
        }
}

void DSOBundleAdjustment::computeDelta() {                                    // BA.cpp:1103-1194
    const int N = (int)mFrames.size();
    mAdHTdeltaF.assign((size_t)N * N * 8, 0.f);
    for (int h = 0; h < N; h++)
        for (int t = 0; t < N; t++) {
            const int idx = h + t * N;
            for (int j = 0; j < 8; j++) {
                double s = 0, s2 = 0;
                for (int i = 0; i < 8; i++) {
                    s += (mFrames[h].state[i] - mFrames[h].state_zero[i]) * mAdHost[64 * (size_t)idx + i * 8 + j];
                    s2 += (mFrames[t].state[i] - mFrames[t].state_zero[i]) * mAdTarget[64 * (size_t)idx + i * 8 + j];
                }
                mAdHTdeltaF[8 * (size_t)idx + j] = (float)(s + s2);
            }
        }
    // mCDeltaF = calibration - mCalibZero: the calibration is not optimised by this path, it stays 0 (BA.cpp:1118)
    const float rotPrior = 1e11f, transPrior = 1e10f, affBPrior = 1e14f, affAPrior = 1e14f;
    float modeA = 1e12f, modeB = 1e8f;
    if (!mOptimizeA) modeA = -1;
    if (!mOptimizeB) modeB = -1;
    for (auto& f : mFrames) {
        std::memset(f.prior, 0, sizeof f.prior);
        if (f.keyid == 0) {
            f.prior[0] = f.prior[1] = f.prior[2] = transPrior; f.prior[3] = f.prior[4] = f.prior[5] = rotPrior;
            f.prior[6] = affAPrior; f.prior[7] = affBPrior;
        } else {
            f.prior[6] = modeA < 0 ? affAPrior : modeA;
            f.prior[7] = modeB < 0 ? affBPrior : modeB;
        }
        for (int i = 0; i < 8; i++) { f.delta[i] = f.state[i] - f.state_zero[i]; f.delta_prior[i] = f.state[i] - f.prior_zero[i]; }
    }
    for (auto& p : mPoints) {                                                 // :1179-1184
        p.priorF = p.hasDepthPrior ? (float)mIdepthFixPrior : 0.f;
        p.deltaF = (float)(p.idepth - (double)p.idepth_zero);
    }
}

void DSOBundleAdjustment::framePairs(std::vector<cmlhip_ba_pair>& out) const {   // DSOFrame.h:259-273
    const int N = (int)mFrames.size();
    out.resize((size_t)N * N);
    for (int h = 0; h < N; h++)
        for (int t = 0; t < N; t++) {
            cmlhip_ba_pair& p = out[(size_t)h * N + t];
            const SE3 ll = mFrames[t].PRE_worldToCam * mFrames[h].PRE_camToWorld;
            ll.matrix(p.R);
            std::memcpy(p.t, ll.t, sizeof p.t);
            const SE3 l0 = mFrames[t].worldToCam_evalPT * mFrames[h].worldToCam_evalPT.inverse();
            l0.matrix(p.R0);
            std::memcpy(p.t0, l0.t, sizeof p.t0);
            mFrames[h].aff_g2l().to(mFrames[t].aff_g2l(), p.aff_a, p.aff_b);
        }
}

void DSOBundleAdjustment::computeNullspaces(std::vector<double>& out) const {    // BA.cpp:2365-2417 (pose x6 + scale)
    const int N = (int)mFrames.size(), n = 8 * N + CMLHIP_CPARS;
    out.assign((size_t)7 * n, 0.0);
    for (int i = 0; i < 6; i++)
        for (int f = 0; f < N; f++)
            for (int k = 0; k < 6; k++)
                out[(size_t)i * n + 4 + 8 * f + k] = mFrames[f].nullspaces_pose[i * 6 + k] * (k < 3 ? 1.0 / mScaleTranslation : 1.0 / mScaleRotation);
    for (int f = 0; f < N; f++)
        for (int k = 0; k < 6; k++)
            out[(size_t)6 * n + 4 + 8 * f + k] = mFrames[f].nullspaces_scale[k] * (k < 3 ? 1.0 / mScaleTranslation : 1.0 / mScaleRotation);
}

// symmetric Jacobi eigen-decomposition (m <= 16)
static void jacobiEig(double* A, int m, double* V) {
    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) V[i * m + j] = (i == j);
    for (int sweep = 0; sweep < 100; sweep++) {
        double off = 0;
        for (int i = 0; i < m; i++) for (int j = i + 1; j < m; j++) off += A[i * m + j] * A[i * m + j];
        if (off < 1e-300) break;
        for (int p = 0; p < m; p++)
            for (int q = p + 1; q < m; q++) {
                const double apq = A[p * m + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double tau = (A[q * m + q] - A[p * m + p]) / (2 * apq);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1 + tau * tau));
                const double c = 1 / std::sqrt(1 + t * t), s = t * c;
                for (int k = 0; k < m; k++) { const double a = A[k * m + p], b = A[k * m + q]; A[k * m + p] = c * a - s * b; A[k * m + q] = s * a + c * b; }
                for (int k = 0; k < m; k++) { const double a = A[p * m + k], b = A[q * m + k]; A[p * m + k] = c * a - s * b; A[q * m + k] = s * a + c * b; }
                for (int k = 0; k < m; k++) { const double a = V[k * m + p], b = V[k * m + q]; V[k * m + p] = c * a - s * b; V[k * m + q] = s * a + c * b; }
            }
    }
}

// inverse of a fixed 8x8 the way Eigen computes Matrix<8,8>::inverse(): partial-pivot LU, then solve against the identity
static void inverse8(const double* A, double* Ainv) {
    double LU[64];
    int piv[8];
    std::memcpy(LU, A, sizeof LU);
    for (int i = 0; i < 8; i++) piv[i] = i;
    for (int k = 0; k < 8; k++) {
        int best = k; double bv = std::fabs(LU[k * 8 + k]);
        for (int i = k + 1; i < 8; i++) if (std::fabs(LU[i * 8 + k]) > bv) { bv = std::fabs(LU[i * 8 + k]); best = i; }
        if (best != k) { for (int j = 0; j < 8; j++) std::swap(LU[k * 8 + j], LU[best * 8 + j]); std::swap(piv[k], piv[best]); }
        for (int i = k + 1; i < 8; i++) {
            LU[i * 8 + k] /= LU[k * 8 + k];
            for (int j = k + 1; j < 8; j++) LU[i * 8 + j] -= LU[i * 8 + k] * LU[k * 8 + j];
        }
    }
    for (int c = 0; c < 8; c++) {
        double y[8];
        for (int i = 0; i < 8; i++) { double s = (piv[i] == c) ? 1.0 : 0.0; for (int j = 0; j < i; j++) s -= LU[i * 8 + j] * y[j]; y[i] = s; }
        for (int i = 7; i >= 0; i--) { double s = y[i]; for (int j = i + 1; j < 8; j++) s -= LU[i * 8 + j] * Ainv[j * 8 + c]; Ainv[i * 8 + c] = s / LU[i * 8 + i]; }
    }
}

// b -= N (N^T N)^+ N^T b, singular values <= delta*max dropped (BA.cpp:1196-1261).  With N = U S V^T the reference's
// 0.5 (N Npi^T + (N Npi^T)^T) is U_kept U_kept^T; U_kept is built from the eigen-decomposition of N^T N.
void DSOBundleAdjustment::nullspaceBasis(std::vector<double>& U) const {
    std::vector<double> ns;
    computeNullspaces(ns);
    const int m = 7, n = 8 * (int)mFrames.size() + CMLHIP_CPARS;
    for (int j = 0; j < m; j++) {
        double s = 0;
        for (int i = 0; i < n; i++) s += ns[(size_t)j * n + i] * ns[(size_t)j * n + i];
        s = std::sqrt(s);
        for (int i = 0; i < n; i++) ns[(size_t)j * n + i] /= s;
    }
    double G[49], V[49];
    for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++) {
            double s = 0;
            for (int k = 0; k < n; k++) s += ns[(size_t)i * n + k] * ns[(size_t)j * n + k];
            G[i * m + j] = s;
        }
    jacobiEig(G, m, V);
    double smax = 0;
    for (int i = 0; i < m; i++) smax = std::max(smax, G[i * m + i] > 0 ? std::sqrt(G[i * m + i]) : 0.0);
    U.assign((size_t)m * n, 0.0);
    for (int e = 0; e < m; e++) {
        const double sv = G[e * m + e] > 0 ? std::sqrt(G[e * m + e]) : 0.0;
        if (!(sv > mSolverModeDelta * smax)) continue;                         // dropped direction: zero row
        for (int i = 0; i < n; i++) {
            double s = 0;
            for (int j = 0; j < m; j++) s += ns[(size_t)j * n + i] * V[j * m + e];
            U[(size_t)e * n + i] = s / sv;
        }
    }
}

void DSOBundleAdjustment::orthogonalize(std::vector<double>& x) const {
    std::vector<double> U;
    nullspaceBasis(U);
    const int m = 7, n = (int)x.size();
    std::vector<double> proj(n, 0.0);
    for (int e = 0; e < m; e++) {
        double dot = 0;
        for (int i = 0; i < n; i++) dot += U[(size_t)e * n + i] * x[i];
        for (int i = 0; i < n; i++) proj[i] += U[(size_t)e * n + i] * dot;
    }
    for (int i = 0; i < n; i++) x[i] -= proj[i];
}

// The window of this run on the device: the library already holds everything up to the last keyframe (cmlhip_ba_window_*); what BA::addNewFrame
// and BA::addPoints added since is appended, the per-point values a run changes (inverse depth, its linearisation point, the prior) and the
// frames' thresholds / b0 are refreshed, and the commit applies the preamble's resetOOB (BA.cpp:766-779) — no rebuild, no copy of the whole window.
bool DSOBundleAdjustment::uploadWindow() {
    if (!mHaveCalib) { mError = "setCalibration not called"; return false; }
    mPrm.huber = (float)mHuberThreshold; mPrm.outlier_th_sum = (float)mSettingOutlierTHSumComponent;
    mPrm.scale_f = mScaleF; mPrm.scale_c = mScaleC; mPrm.optimize_a = mOptimizeA; mPrm.optimize_b = mOptimizeB;
    int rc = cmlhip_ba_set_params(mCtx, &mPrm);
    if (rc) return fail("cmlhip_ba_set_params", rc);
    HostLap lapU("uploadWindow");
    if (mDeadSinceCompact) compactDead();                    // (entries dropped since the last addNewFrame: the window holds live entries only)
    lapU("compact");
    if (!syncWindowAppends()) return false;
    lapU("appends");
    const int N = (int)mFrames.size();
    std::vector<cmlhip_ba_frame> fr(N);
    for (int i = 0; i < N; i++) {
        fr[i].image_id = mFrames[i].image_id;
        fr[i].frame_energy_th = (float)mFrames[i].frameEnergyTH;
        fr[i].b0 = mFrames[i].getB0((float)mScaleLightB);
    }
    const size_t P = mPoints.size(), R = mResiduals.size();
    mDynIdepth.resize(P); mDynZero.resize(P); mDynPrior.resize(P);
    for (size_t p = 0; p < P; p++) { const DSOPoint& Q = mPoints[p]; mDynIdepth[p] = Q.idepth; mDynZero[p] = Q.idepth_zero; mDynPrior[p] = Q.priorF; }
    // residuals that stay LINEARIZED over this run keep their state (:766-779); everything else is reset by the commit (and, for this object's own
    // copies of the fields, by the closing pass's bookkeeping)
    std::vector<int> linIdx, linState;
    if (mLinearizedAlive > 0 && !mAddLinearizedPoints)
        for (size_t r = 0; r < R; r++) if (mResiduals[r].isLinearized) { linIdx.push_back((int)r); linState.push_back(mResiduals[r].state_state); }
    if (mAddLinearizedPoints && mLinearizedAlive > 0) { for (auto& Rr : mResiduals) Rr.isLinearized = false; mLinearizedAlive = 0; }
    lapU("dyn arrays");
    rc = cmlhip_ba_window_commit(mCtx, N, fr.data(), mDynIdepth.data(), mDynZero.data(), mDynPrior.data(), 1, (int)linIdx.size(), linIdx.data(), linState.data());
    if (rc) return fail("cmlhip_ba_window_commit", rc);
    lapU("commit");
    // every entry is live: the device numbering is the lists' own
    mActive.resize(R); mActivePoints.resize(P); mPointSlot.resize(P);
    for (size_t r = 0; r < R; r++) mActive[r] = (int)r;
    for (size_t p = 0; p < P; p++) { mActivePoints[p] = (int)p; mPointSlot[p] = (int)p; }
    mPairsValid = false;                                     // (an upload forgets the pair records)
    lapU("identity maps");
    return true;
}

// cmlhip_ba_set_pairs unless the device already holds exactly these records for this upload (run()'s preamble and beginResident compute the
// same N^2 records from the same frame states)
int DSOBundleAdjustment::setPairs(const std::vector<cmlhip_ba_pair>& pairs) {
    if (mPairsValid && mPairsSent.size() == pairs.size() && std::memcmp(mPairsSent.data(), pairs.data(), sizeof(cmlhip_ba_pair) * pairs.size()) == 0) return CMLHIP_OK;
    const int rc = cmlhip_ba_set_pairs(mCtx, pairs.data());
    mPairsValid = rc == CMLHIP_OK;
    if (mPairsValid) mPairsSent = pairs;
    return rc;
}

bool DSOBundleAdjustment::linearizeAll(bool fixLinearization, double energy[3], std::vector<double>* idepthOut, std::vector<float>* pointAccOut, bool applyToo, bool enqueueOnly) {   // BA.cpp:1497-1646
    HostLap lapL("linearizeAll");
    std::vector<cmlhip_ba_pair> pairs;
    framePairs(pairs);
    int rc = setPairs(pairs);
    if (rc) return fail("cmlhip_ba_set_pairs", rc);
    lapL("pairs set");
    cmlhip_ba_lin_result lr;
    const int R = (int)mActive.size();
    std::vector<int> st, ns;
    std::vector<float> e, ne, nw;
    std::vector<unsigned char> good;
    if (fixLinearization) {
        // linearize + applyRes(r, true) (:1568-1569) + every array the host writes back afterwards: one device call, one readback
        // (the host logic behind this pass reads state_state and isActiveAndIsGoodNEW only — isOOB, the removal rule, numGoodResiduals;
        //  the energies and state_NewState stay on the device unless mKeepResidualEnergies asks for them)
        st.resize(R); good.resize(R);
        if (mKeepResidualEnergies) { ns.resize(R); e.resize(R); ne.resize(R); nw.resize(R); }
        if (idepthOut) idepthOut->resize(mActivePoints.size());
        if (pointAccOut) pointAccOut->resize(14 * mActivePoints.size() + 14);
        rc = cmlhip_ba_finish_keyframe(mCtx, &lr, st.data(), mKeepResidualEnergies ? ns.data() : nullptr, mKeepResidualEnergies ? e.data() : nullptr,
                                       mKeepResidualEnergies ? ne.data() : nullptr, mKeepResidualEnergies ? nw.data() : nullptr, good.data(),
                                       idepthOut ? idepthOut->data() : nullptr, pointAccOut ? pointAccOut->data() : nullptr);
        if (rc && rc != CMLHIP_ERR_NONFINITE) return fail("cmlhip_ba_finish_keyframe", rc);
    } else {
        if (applyToo && enqueueOnly) {                                              // run()'s preamble when the loop is resident: no host wait, the summary comes back with
            rc = cmlhip_ba_linearize_apply(mCtx, nullptr);                          // cmlhip_ba_finish_run's one copy
            if (rc) return fail("cmlhip_ba_linearize_apply", rc);
            energy[0] = energy[1] = energy[2] = 0;
            return true;
        }
        rc = applyToo ? cmlhip_ba_linearize_apply(mCtx, &lr) : cmlhip_ba_linearize(mCtx, &lr);
        if (rc && rc != CMLHIP_ERR_NONFINITE) return fail("cmlhip_ba_linearize", rc);
    }
    lapL("device call done");
    energy[0] = lr.energy; energy[1] = 0; energy[2] = 0;
    mFrames.back().frameEnergyTH = lr.new_frame_energy_th;                // setNewFrameEnergyTH, :1610
    if (fixLinearization) closingBookkeeping(st, good, ns, e, ne, nw, nullptr);
    lapL("bookkeeping done");
    return true;
}

// host bookkeeping behind linearizeAll(true), BA.cpp:1571-1640, from the arrays the device's closing pass returned (caller order = mActive)
// (packed != nullptr: state | good << 2 per residual, as cmlhip_ba_finish_run packs them — st / good are then not read)
void DSOBundleAdjustment::closingBookkeeping(const std::vector<int>& st, const std::vector<unsigned char>& good, const std::vector<int>& ns,
                                             const std::vector<float>& e, const std::vector<float>& ne, const std::vector<float>& nw, const unsigned char* packed) {
    const int R = (int)mActive.size();
    std::vector<int>& nres = mScratchCount;
    nres.assign(mPoints.size(), 0);
    const bool allActive = (size_t)R == mResiduals.size();                      // (the committed window is the whole list: the census below needs no second pass)
    for (int k = 0; k < R; k++) {
        DSOResidual& Rr = mResiduals[mActive[k]];
        const int st_k = packed ? (packed[k] & 3) : st[k];
        Rr.state_state = st_k; Rr.isActiveAndIsGoodNEW = packed ? ((packed[k] >> 2) & 1) != 0 : good[k] != 0;
        if (mKeepResidualEnergies) { Rr.state_NewState = ns[k]; Rr.state_energy = e[k]; Rr.state_NewEnergy = ne[k]; Rr.state_NewEnergyWithOutlier = nw[k]; }
        else {
            Rr.state_NewState = st_k;                                           // applyNewState: state_state = state_NewState (DSOResidual.h)
            if (!Rr.isLinearized) Rr.state_NewEnergy = Rr.state_energy = 0;     // resetOOB of the preamble (:766-779); the energies stay on the device in this mode
        }
        DSOPoint& Pp = mPoints[Rr.point];
        for (int q = 0; q < 2; q++) if (Pp.lastResidual[q] == mActive[k]) Pp.lastResidualState[q] = Rr.state_state;   // setResidualState, :1618-1622
        if (Rr.isLinearized) { nres[Rr.point] += Rr.alive; continue; }
        if (Rr.isActiveAndIsGoodNEW) Pp.numGoodResiduals++;                     // :1592
        else {                                                                  // toRemove, :1595-1598,1624-1638
            Rr.alive = false; mDeadSinceCompact++;
            mFrames[Rr.target].numResidualsOut++;                               // removeResiduals, DSOContext.h:210
            for (int q = 0; q < 2; q++) if (Pp.lastResidual[q] == mActive[k]) Pp.lastResidual[q] = -1;
        }
        nres[Rr.point] += Rr.alive;
    }
    if (!allActive) { nres.assign(mPoints.size(), 0); for (const auto& Rr : mResiduals) if (Rr.alive) nres[Rr.point]++; }
    for (int p = 0; p < (int)mPoints.size(); p++)                               // points left without residual, :1638-1640
        if (mPoints[p].alive && nres[p] == 0) { mPoints[p].alive = false; mDeadSinceCompact++; mOutliers.push_back(p); }
}

void DSOBundleAdjustment::backupState() {                                     // BA.cpp:912-926
    for (auto& f : mFrames) f.backupState();
    cmlhip_ba_backup_points(mCtx);
}

bool DSOBundleAdjustment::solveSystem(int iteration, double lambda) {         // BA.cpp:1339-1495
    const int N = (int)mFrames.size(), n = 8 * N + CMLHIP_CPARS;
    if (mFixLambda) lambda = mFixedLambda;
    std::vector<double> prior(8 * (size_t)N), dprior(8 * (size_t)N);
    for (int i = 0; i < N; i++) for (int k = 0; k < 8; k++) { prior[8 * i + k] = mFrames[i].prior[k]; dprior[8 * i + k] = mFrames[i].delta_prior[k]; }
    double cdelta[4] = {mCDeltaF[0], mCDeltaF[1], mCDeltaF[2], mCDeltaF[3]}, cprior[4] = {mCPriorValue, mCPriorValue, mCPriorValue, mCPriorValue};
    cmlhip_ba_accum_in in{mAdHost.data(), mAdTarget.data(), mAdHTdeltaF.data(), cdelta, prior.data(), dprior.data(), cprior};
    const bool wantStats = true;
    std::vector<double> HA, bA, Hsc, bsc;
    if (wantStats) { HA.resize((size_t)n * n); bA.resize(n); Hsc.resize((size_t)n * n); bsc.resize(n); }
    int rc = cmlhip_ba_accumulate(mCtx, &in, wantStats ? HA.data() : nullptr, wantStats ? bA.data() : nullptr, nullptr, nullptr,
                                  wantStats ? Hsc.data() : nullptr, wantStats ? bsc.data() : nullptr);
    if (rc) return fail("cmlhip_ba_accumulate", rc);
    const double* HM = nullptr; const double* bM = nullptr;
    std::vector<double> bMtop;
    if (!mDisableMarginalization) {                                           // BA.cpp:1389-1401
        bMtop.assign(n, 0.0);
        std::vector<double> d(n, 0.0);
        for (int h = 0; h < N; h++) for (int k = 0; k < 8; k++) d[4 + 8 * h + k] = mFrames[h].delta[k];
        for (int i = 0; i < n; i++) {
            double s = mMarginalizedB[i];
            for (int j = 0; j < n; j++) s += mMarginalizedHessian[(size_t)i * n + j] * d[j];
            bMtop[i] = s;
        }
        HM = mMarginalizedHessian.data(); bM = bMtop.data();
    } else {
        std::fill(mMarginalizedHessian.begin(), mMarginalizedHessian.end(), 0.0);     // :1395-1398
        std::fill(mMarginalizedB.begin(), mMarginalizedB.end(), 0.0);
    }
    mX.assign(n, 0.0);
    rc = cmlhip_ba_solve(mCtx, lambda, HM, bM, mOptimizeCalibration ? 1 : 0, mX.data());
    if (rc == CMLHIP_ERR_NONFINITE) { mError = "non-finite solution"; /* the reference dumps the system and carries on, :1323-1325 */ }
    else if (rc) return fail("cmlhip_ba_solve", rc);
    if (N > 4 && !addIndirectToProblem(mX)) return false;                     // :1327-1329
    if (iteration >= 2) orthogonalize(mX);                                    // :1404 mustOrthogonalize
    auto norm = [](const std::vector<double>& v) { double s = 0; for (double a : v) s += a * a; return std::sqrt(s); };
    if (wantStats) { statHessianP.push_back(norm(HA)); statHessianSC.push_back(norm(Hsc)); statBP.push_back(norm(bA)); statBSC.push_back(norm(bsc)); }
    statXNorm.push_back(norm(mX));
    for (int h = 0; h < N; h++) {                                             // :1433-1441
        double st[10] = {0};
        for (int k = 0; k < 8; k++) st[k] = -mX[4 + 8 * h + k];
        mFrames[h].setStep(st);
    }
    rc = cmlhip_ba_backsub(mCtx, mX.data(), nullptr);                          // :1455-1487
    if (rc == CMLHIP_ERR_NONFINITE) { mError = "points without a finite step"; return false; }   // :1489-1492
    if (rc) return fail("cmlhip_ba_backsub", rc);
    return true;
}

// ------------------------------------------------------------------------------------------------ hybrid ORB term
void DSOBundleAdjustment::setIndirectPoints(const std::vector<double>& worldXYZ, const std::vector<cmlhip_reproj_obs>& observations) {
    mIndirectPoints = worldXYZ;
    mIndirectObs = observations;
    mIndirectUncertainty.assign(worldXYZ.size() / 3, 0.0);
}

// addIndirectToProblem, BA.cpp:2574-2729.  Device: the per-observation Jacobians (ReprojectionError::jacobian through
// Dx_exp_x), the pose block of J J^T, b and the per-point Jacobian sums (cmlhip_reproj_accumulate), then
// ldlt(M with diag*(1+fixedLambda)).solve(-bM) (cmlhip_reproj_solve).  Host: the literal weighting of :2714-2727 —
// numIndirectPoint = 1 and numDirectPoint = 0 are constants there, so the pose part of x is REPLACED by the indirect solution.
void DSOBundleAdjustment::indirectUncertaintyFrom(const std::vector<double>& Jp) {
    const int M = (int)(mIndirectPoints.size() / 3);
    for (int j = 0; j < M; j++) {                                             // (Jp Jp^T).inverse().diagonal().norm(), :2690-2692: the inverse of
        const double* a = &Jp[3 * j];                                         // a rank-one 3x3 by cofactors / determinant, whatever that gives
        double A[9], C[3];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) A[3 * r + c] = a[r] * a[c];
        C[0] = A[4] * A[8] - A[5] * A[7]; C[1] = A[0] * A[8] - A[2] * A[6]; C[2] = A[0] * A[4] - A[1] * A[3];
        const double det = A[0] * C[0] - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
        const double d0 = C[0] / det, d1 = C[1] / det, d2 = C[2] / det;
        mIndirectUncertainty[j] = std::sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    }
}

bool DSOBundleAdjustment::addIndirectToProblem(std::vector<double>& X) {
    if (!mMixedBundleAdjustment) return true;                                 // :2575-2577
    const int N = (int)mFrames.size(), M = (int)(mIndirectPoints.size() / 3), n = (int)mIndirectObs.size();
    if (M == 0) return true;                                                  // :2587-2589
    std::vector<double> poses(12 * (size_t)N), Jp(3 * (size_t)M);
    for (int i = 0; i < N; i++) {                                             // frame->getCamera() is cameraOf(PRE_worldToCam) (:934, :966)
        mFrames[i].PRE_worldToCam.matrix(&poses[12 * i]);
        for (int k = 0; k < 3; k++) poses[12 * i + 9 + k] = mFrames[i].PRE_worldToCam.t[k];
    }
    int rc = cmlhip_reproj_accumulate(mCtx, N, poses.data(), M, mIndirectPoints.data(), n, mIndirectObs.data(), mPrm.fx, mPrm.fy,
                                      nullptr, nullptr, Jp.data(), nullptr);
    if (rc) return fail("cmlhip_reproj_accumulate", rc);
    indirectUncertaintyFrom(Jp);
    mIndirectX.assign(6 * (size_t)N, 0.0);
    rc = cmlhip_reproj_solve(mCtx, N, mFixedLambda, mIndirectX.data());       // :2695-2700
    if (rc && rc != CMLHIP_ERR_NONFINITE) return fail("cmlhip_reproj_solve", rc);
    for (double v : mIndirectX) if (!std::isfinite(v)) return true;           // :2702-2704
    for (int i = 0; i < N; i++) {                                             // :2714-2727
        const int numIndirectPoint = 1, numDirectPoint = 0;
        const double indirectRatio = (double)numIndirectPoint / (double)(numIndirectPoint + numDirectPoint);
        const double directRatio = 1.0 - indirectRatio;
        for (int k = 0; k < 6; k++) X[4 + 8 * i + k] = X[4 + 8 * i + k] * directRatio + mIndirectX[6 * i + k] * indirectRatio;
    }
    return true;
}

bool DSOBundleAdjustment::doStepFromBackup(bool fixCamera) {                  // BA.cpp:948-1028
    double sc[4];
    scales(sc);
    float sumA = 0, sumB = 0, sumT = 0, sumR = 0;
    for (auto& f : mFrames) {
        if (fixCamera) for (int i = 0; i < 6; i++) f.step[i] = 0;
        f.doStepFromBackup(sc);
        sumA += (float)(f.step[6] * f.step[6]);
        sumB += (float)(f.step[7] * f.step[7]);
        sumT += (float)(f.step[0] * f.step[0] + f.step[1] * f.step[1] + f.step[2] * f.step[2]);
        sumR += (float)(f.step[3] * f.step[3] + f.step[4] * f.step[4] + f.step[5] * f.step[5]);
    }
    float sums[3] = {0, 0, 0};
    cmlhip_ba_step_points(mCtx, sums);
    float sumID = sums[0], sumNID = sums[1];
    const float numID = sums[2];
    const float nf = (float)mFrames.size();
    sumA /= nf; sumB /= nf; sumR /= nf; sumT /= nf; sumID /= numID; sumNID /= numID;
    (void)sumID;
    computeDelta();
    return std::sqrt(sumA) < 0.0005 * mThOptIterations && std::sqrt(sumB) < 0.00005 * mThOptIterations &&
           std::sqrt(sumR) < 0.00005 * mThOptIterations && std::sqrt(sumT) * sumNID < 0.00005 * mThOptIterations;
}

bool DSOBundleAdjustment::runPreamble(double lastEnergy[3], bool enqueueOnly) {                // BA.cpp:744-802
    HostLap lap("preamble");
    const auto T0 = lap.t0;
    mOutliers.clear();
    mError.clear();
    lastIterations = 0;
    int alivePts = 0;
    for (const auto& p : mPoints) alivePts += p.alive;
    if (alivePts == 0) { mError = "No points..."; return false; }             // :759-762
    computeAdjoints();
    computeDelta();
    lap("adjoints+delta");
    if (!uploadWindow()) return false;
    lap("uploadWindow");
    lastRunUs[0] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - T0).count();
    if (!linearizeAll(false, lastEnergy, nullptr, nullptr, true, enqueueOnly)) return false;   // linearizeAll(false) + applyActiveRes(true), :785-790, one pass on the device
    lap("linearizeAll");
    if (!enqueueOnly) statEnergyP.push_back(lastEnergy[0] / std::max<size_t>(1, mActive.size()));
    return true;
}

bool DSOBundleAdjustment::runEpilogue(double lastEnergy[3]) {                // BA.cpp:882-910
    double sc[4];
    scales(sc);
    // re-anchor the newest frame's evaluation point, :885-894
    DSOFrame& fb = mFrames.back();
    double nz[10] = {0};
    nz[6] = fb.state[6]; nz[7] = fb.state[7];
    fb.setEvalPT(fb.PRE_worldToCam, nz, sc);
    computeAdjoints();
    computeDelta();
    {   // getB0 follows state_zero (DSOFrame.h:197-199): the device's copy of the re-anchored frame's b0 is refreshed before the closing pass
        std::vector<float> b0(mFrames.size());
        for (size_t i = 0; i < mFrames.size(); i++) b0[i] = mFrames[i].getB0((float)mScaleLightB);
        const int rcb = getenv("CMLHOST_NO_B0_REFRESH") ? 0 : cmlhip_ba_set_frame_b0(mCtx, b0.data());      // (development switch: the stale value)
        if (rcb) return fail("cmlhip_ba_set_frame_b0", rcb);
    }
    std::vector<double> idp;
    std::vector<float> pacc;
    if (!linearizeAll(true, lastEnergy, &idp, &pacc)) return false;           // :896 (+ the inverse depths and point accumulators, same readback)
    if (!std::isfinite(lastEnergy[0])) { mError = "Not finite energy"; return false; }
    // write the optimised inverse depths back (MapPoint::setReferenceInverseDepth in the reference)
    for (size_t k = 0; k < mActivePoints.size(); k++) {
        DSOPoint& P = mPoints[mActivePoints[k]];
        P.idepth = idp[k];
        P.idepth_zero = (float)idp[k];
        const float hdi = pacc[14 * k + 12];                                      // setInverseDepthHessian(H), HdiF = 1/H, BA.cpp:1889-1901
        P.idepth_hessian = hdi > 0 ? 1.0f / hdi : 0.f;
    }
    return true;
}

bool DSOBundleAdjustment::run(bool updatePointsOnly) {                        // BA.cpp:744-910
    // forceAccept + fixLambda (the reference's defaults, BA.h:265-270): every step is accepted and lambda never changes, so the loop
    // body has no host decision left except the early exit, which the device mirrors; the marginalisation prior, when enabled, only
    // enters the solve under forceAccept (calcMEnergy / calcLEnergy return 0, BA.cpp:2100-2102,2123-2125) and is resident too
    // (the hybrid ORB term is mixed into x inside the device solve: cmlhip_ba_set_resident_indirect)
    if (mResidentLoop && mForceAccept && mFixLambda && mNumIterations <= 40) return runResident(updatePointsOnly);
    return runHostLoop(updatePointsOnly);
}

bool DSOBundleAdjustment::runHostLoop(bool updatePointsOnly) {
    double sc[4];
    scales(sc);
    double lastEnergy[3], newEnergy[3];
    if (!runPreamble(lastEnergy)) return false;
    double lastEnergyL = calcLEnergy(), lastEnergyM = calcMEnergy();          // BA.cpp:783-784 (0 under forceAccept, :2100-2102,2123-2125)
    int rc;
    double lambda = mFixedLambda;
    for (int it = 0; it < mNumIterations; it++) {
        lastIterations = it + 1;
        backupState();
        if (!solveSystem(it, lambda)) return false;                           // :813-816
        const bool canbreak = doStepFromBackup(updatePointsOnly);
        if (!linearizeAll(false, newEnergy)) return false;
        const double newEnergyL = calcLEnergy(), newEnergyM = calcMEnergy();
        const double newTotal = newEnergy[0] + newEnergy[1] + newEnergyL + newEnergyM;       // :830-831
        const double lastTotal = lastEnergy[0] + lastEnergy[1] + lastEnergyL + lastEnergyM;
        if (!std::isfinite(newTotal)) { mError = "non finite energy"; return false; }     // :836-841
        if (newTotal < lastTotal || mForceAccept) {
            statEnergyP.push_back(newEnergy[0]);
            rc = cmlhip_ba_apply(mCtx, 1);
            if (rc) return fail("cmlhip_ba_apply", rc);
            for (int k = 0; k < 3; k++) lastEnergy[k] = newEnergy[k];
            lastEnergyL = newEnergyL; lastEnergyM = newEnergyM;
            lambda *= 0.25;
        } else {                                                              // loadSateBackup, :871-875
            for (auto& f : mFrames) f.loadSateBackup(sc);
            rc = cmlhip_ba_restore_points(mCtx);
            if (rc) return fail("cmlhip_ba_restore_points", rc);
            computeDelta();
            if (!linearizeAll(false, lastEnergy)) return false;
            lastEnergyL = calcLEnergy(); lastEnergyM = calcMEnergy();
            lambda *= 1e2;
            statRejected++;
        }
        lastLambda = lambda;
        if (canbreak && it >= 1) break;                                       // :879
    }
    return runEpilogue(lastEnergy);
}

// ------------------------------------------------------------------------------------------------ marginalisation
void DSOBundleAdjustment::fillAccumIn(cmlhip_ba_accum_in& in, std::vector<double>& prior, std::vector<double>& dprior, double cdelta[4], double cprior[4]) {
    const int N = (int)mFrames.size();
    prior.assign(8 * (size_t)N, 0.0); dprior.assign(8 * (size_t)N, 0.0);
    for (int i = 0; i < N; i++) for (int k = 0; k < 8; k++) { prior[8 * i + k] = mFrames[i].prior[k]; dprior[8 * i + k] = mFrames[i].delta_prior[k]; }
    for (int i = 0; i < 4; i++) { cdelta[i] = mCDeltaF[i]; cprior[i] = mCPriorValue; }
    in = cmlhip_ba_accum_in{mAdHost.data(), mAdTarget.data(), mAdHTdeltaF.data(), cdelta, prior.data(), dprior.data(), cprior};
}

void DSOBundleAdjustment::removePointsWithoutResidual() {                     // DSOContext.h:218-229
    for (int p = 0; p < (int)mPoints.size(); p++) {          // (a live point's own list says whether a live residual is left: no pass over every residual)
        if (!mPoints[p].alive) continue;
        bool any = false;
        for (int ri : mPointRes[p]) if (mResiduals[ri].alive) { any = true; break; }
        if (!any) { mPoints[p].alive = false; mDeadSinceCompact++; }
    }
}

void DSOBundleAdjustment::removePoint(int p, bool marginalize, bool sweep) {  // DSOContext.h:94-111,204-215
    if (!mPoints[p].alive) return;
    char seen[CMLHIP_MAX_FRAMES] = {0};                      // (no allocation per removed point)
    for (int ri : mPointRes[p]) {
        DSOResidual& r = mResiduals[ri];
        if (!r.alive) continue;
        if (marginalize && !seen[r.target]) { seen[r.target] = 1; mFrames[r.target].numMarginalized++; }
        r.alive = false;
        mFrames[r.target].numResidualsOut++;
    }
    mPoints[p].alive = false; mDeadSinceCompact++;
    if (sweep) removePointsWithoutResidual();        // (callers that remove many points sweep once behind their loop: the end state is the same)
}

void DSOBundleAdjustment::removeFrame(int f) {                                // DSOContext.h:154-174
    for (int p = 0; p < (int)mPoints.size(); p++) if (mPoints[p].alive && mPoints[p].host == f) removePoint(p, false, false);
    for (auto& r : mResiduals) if (r.alive && r.target == f) { r.alive = false; mDeadSinceCompact++; mFrames[f].numResidualsOut++; }
    removePointsWithoutResidual();
    mDeadSinceCompact++;                                                       // (the frame's entries carry -1 from here on: the next addNewFrame / run renumbers)
    if (syncWindowAppends()) {                                                 // the library's copy of the window: same renumbering of the frame ids
        if (cmlhip_ba_window_retire_frame(mCtx, f)) { cmlhip_ba_window_reset(mCtx); mWinPoints = mWinResiduals = 0; }
    } else { cmlhip_ba_window_reset(mCtx); mWinPoints = mWinResiduals = 0; mError.clear(); }
    mFrames.erase(mFrames.begin() + f);
    for (int i = 0; i < (int)mFrames.size(); i++) mFrames[i].id = i;          // makeFrameId
    for (auto& P : mPoints) { if (P.host == f) P.host = -1; else if (P.host > f) P.host--; }
    for (auto& r : mResiduals) { if (r.target == f) r.target = -1; else if (r.target > f) r.target--; }
}

bool DSOBundleAdjustment::isOOB(int p, const std::vector<int>& toMarg) const {   // BA.cpp:2515-2554 (toKeep is never filled, :2253-2258)
    const int setting_minGoodActiveResForMarg = 3, setting_minGoodResForMarg = 4;
    const DSOPoint& P = mPoints[p];
    int visInToMarg = 0, numIn = 0;
    for (int ri : mPointRes[p]) {
        const DSOResidual& r = mResiduals[ri];
        if (!r.alive || r.state_state != DSORES_IN) continue;
        numIn++;
        for (int k : toMarg) if (r.target == k) visInToMarg++;
    }
    if (numIn >= setting_minGoodActiveResForMarg && P.numGoodResiduals > setting_minGoodResForMarg + 10 &&
        numIn - visInToMarg < setting_minGoodActiveResForMarg) return true;
    if (P.lastResidualState[0] == DSORES_OOB) return true;
    if (numIn < 2) return false;
    if (P.lastResidualState[0] == DSORES_OUTLIER && P.lastResidualState[1] == DSORES_OUTLIER) return true;
    return false;
}

void DSOBundleAdjustment::flagFramesForMarginalization(int numImmaturePerFrame) {
    flagFramesForMarginalization(std::vector<int>(mFrames.size(), numImmaturePerFrame));
}

void DSOBundleAdjustment::flagFramesForMarginalization(const std::vector<int>& immaturePerFrame) {   // BA.cpp:603-716
    const int N = (int)mFrames.size();
    int flagged = 0;
    double sc[4];
    scales(sc);
    std::vector<int> nresOfFrame(N, 0);
    for (const auto& r : mResiduals) if (r.alive) nresOfFrame[r.target]++;
    for (int i = 0; i < N; i++) {
        DSOFrame& f = mFrames[i];
        const double in = nresOfFrame[i] + (i < (int)immaturePerFrame.size() ? immaturePerFrame[i] : 0);
        const double out = f.numMarginalized + f.numResidualsOut;
        double a, b;
        mFrames.back().aff_g2l().to(f.aff_g2l(), a, b);                         // frameBack exposure -> frame exposure
        const double setting_minPointsRemaining = 0.05, setting_maxLogAffFacInWindow = 0.7;
        const int setting_minFrames = mMaxFrames - 2;
        const bool notEnough = in < setting_minPointsRemaining * (in + out);
        const bool tooBig = std::fabs(std::log(a)) > setting_maxLogAffFacInWindow && N - flagged > setting_minFrames;
        if (notEnough || tooBig) { f.flaggedForMarginalization = true; flagged++; }
    }
    if (N - flagged >= mMaxFrames) {                                           // marginalize one, :648-708
        double smallestScore = 1;
        int toMarginalize = -1;
        const DSOFrame& latest = mFrames.back();
        for (int ri = 0; ri < N; ri++) {
            const DSOFrame& ref = mFrames[ri];
            if (ref.keyid > latest.keyid - mMinFrameAge || ref.keyid == 0) continue;
            double distScore = 0;
            for (int ti = 0; ti < N; ti++) {
                if (ti == ri) continue;
                const DSOFrame& tg = mFrames[ti];
                if (tg.keyid > latest.keyid - mMinFrameAge + 1) continue;
                const SE3 rt = tg.PRE_worldToCam * ref.PRE_camToWorld;            // reference camera -> target camera
                const double d = std::sqrt(rt.t[0] * rt.t[0] + rt.t[1] * rt.t[1] + rt.t[2] * rt.t[2]);
                distScore += 1.0 / (1e-5 + d);
            }
            const SE3 rb = latest.PRE_worldToCam * ref.PRE_camToWorld;
            distScore *= -std::sqrt(std::sqrt(rb.t[0] * rb.t[0] + rb.t[1] * rb.t[1] + rb.t[2] * rb.t[2]));
            if (distScore < smallestScore) { smallestScore = distScore; toMarginalize = ri; }
        }
        if (toMarginalize >= 0) { mFrames[toMarginalize].flaggedForMarginalization = true; flagged++; }
    }
}

bool DSOBundleAdjustment::tryMarginalize() {                                  // BA.cpp:2240-2363
    HostLap lap("tryMarginalize");
    const int setting_minGoodActiveResForMarg = 3, setting_minGoodResForMarg = 4;
    std::vector<int> toMarg;
    for (int i = 0; i < (int)mFrames.size(); i++) if (mFrames[i].flaggedForMarginalization) toMarg.push_back(i);
    std::vector<int> nres(mPoints.size(), 0);
    for (const auto& r : mResiduals) if (r.alive) nres[r.point]++;
    std::vector<int> toDrop, candidates;
    for (int p = 0; p < (int)mPoints.size(); p++) {
        const DSOPoint& P = mPoints[p];
        if (!P.alive) continue;
        if (P.idepth < 0 || nres[p] == 0) toDrop.push_back(p);
        else if (isOOB(p, toMarg) || mFrames[P.host].flaggedForMarginalization) {
            if (nres[p] >= setting_minGoodActiveResForMarg && P.numGoodResiduals >= setting_minGoodResForMarg) candidates.push_back(p);
            else toDrop.push_back(p);
        }
    }
    lap("classified");
    // residual loop of the candidates on the device (:2291-2304): resetOOB, linearize, applyRes(true), fixLinearization
    if (!candidates.empty()) {
        std::vector<int> slots;
        for (int p : candidates) {
            if (mPointSlot[p] < 0) { mError = "tryMarginalize: point is not in the uploaded window"; return false; }
            slots.push_back(mPointSlot[p]);
        }
        computeDelta();
        std::vector<cmlhip_ba_pair> pairs;
        framePairs(pairs);
        int rc = setPairs(pairs);
        if (rc) return fail("cmlhip_ba_set_pairs", rc);
        lap("delta+pairs set");
        cmlhip_ba_accum_in in; std::vector<double> prior, dprior; double cdelta[4], cprior[4];
        fillAccumIn(in, prior, dprior, cdelta, cprior);
        int ngood = 0;
        const int R = (int)mActive.size();
        {   // state / good / LINEARIZED / new state of every residual as one byte, and the three energies, IN THE PASS'S OWN READBACK (one wait; it was three)
            std::vector<unsigned char> pk(R);
            std::vector<float> e(R), ne(R), nw(R);
            rc = cmlhip_ba_relinearize_points_packed(mCtx, &in, (int)slots.size(), slots.data(), &ngood, pk.data(), e.data(), ne.data(), nw.data());
            if (rc) return fail("cmlhip_ba_relinearize_points_packed", rc);
            lap("device pass + readback");
            std::vector<char> isCand(mPoints.size(), 0);
            for (int p : candidates) isCand[p] = 1;
            for (int k = 0; k < R; k++) {
                DSOResidual& Rr = mResiduals[mActive[k]];
                if (!isCand[Rr.point]) continue;
                Rr.state_state = pk[k] & 3; Rr.state_NewState = (pk[k] >> 4) & 3; Rr.state_energy = e[k]; Rr.state_NewEnergy = ne[k];
                Rr.state_NewEnergyWithOutlier = nw[k]; Rr.isActiveAndIsGoodNEW = (pk[k] & 4) != 0; Rr.isLinearized = (pk[k] & 8) != 0;
                mLinearizedAlive += (pk[k] & 8) != 0;
            }
        }
    }
    for (int p : candidates) {
        if (mPoints[p].idepth_hessian > mMinIdepthHMarg) mPoints[p].toMarginalize = true;       // :2316-2325
        else toDrop.push_back(p);
    }
    lap("states assigned");
    for (int p : toDrop) { removePoint(p, false, false); mOutliers.push_back(p); }    // :2344-2348
    removePointsWithoutResidual();
    lap("points removed");
    return true;
}

bool DSOBundleAdjustment::marginalizePointsF() {                              // BA.cpp:2466-2513
    HostLap lap("marginalizePointsF");
    computeDelta();
    computeAdjoints();
    lap("delta+adjoints");
    std::vector<int> pts, slots;
    for (int p = 0; p < (int)mPoints.size(); p++)
        if (mPoints[p].toMarginalize) { pts.push_back(p); slots.push_back(mPointSlot[p]); }
    const int n = 8 * (int)mFrames.size() + CMLHIP_CPARS;
    std::vector<double> M((size_t)n * n, 0.0), Mb(n, 0.0), Msc((size_t)n * n, 0.0), Mbsc(n, 0.0);
    if (!pts.empty()) {
        cmlhip_ba_accum_in in; std::vector<double> prior, dprior; double cdelta[4], cprior[4];
        fillAccumIn(in, prior, dprior, cdelta, cprior);
        const int rc = cmlhip_ba_marginalize_points(mCtx, &in, (int)slots.size(), slots.data(), M.data(), Mb.data(), Msc.data(), Mbsc.data());
        if (rc) return fail("cmlhip_ba_marginalize_points", rc);
    }
    lap("device accumulate + readback");
    for (int p : pts) {                                                        // :2490-2497
        mPoints[p].marginalized = true; mPoints[p].toMarginalize = false;
        removePoint(p, true, false);
    }
    removePointsWithoutResidual();
    const double setting_margWeightFac = 0.5 * 0.5;                            // :2502
    for (size_t i = 0; i < M.size(); i++) mMarginalizedHessian[i] += setting_margWeightFac * (M[i] - Msc[i]);
    for (int i = 0; i < n; i++) mMarginalizedB[i] += setting_margWeightFac * (Mb[i] - Mbsc[i]);
    lap("points removed + prior");
    return true;
}

void DSOBundleAdjustment::marginalizeFrame(int frame) {                       // BA.cpp:464-601
    const int N = (int)mFrames.size(), odim = 8 * N + CMLHIP_CPARS, ndim = odim - 8;
    std::vector<double> H((size_t)odim * odim), b(odim);
    std::vector<int> perm;
    const int io = 8 * frame + CMLHIP_CPARS;
    for (int i = 0; i < odim; i++) if (i < io || i >= io + 8) perm.push_back(i);   // frame block to the end, order of the rest kept (:489-508)
    for (int i = 0; i < 8; i++) perm.push_back(io + i);
    for (int i = 0; i < odim; i++) {
        b[i] = mMarginalizedB[perm[i]];
        for (int j = 0; j < odim; j++) H[(size_t)i * odim + j] = mMarginalizedHessian[(size_t)perm[i] * odim + perm[j]];
    }
    const DSOFrame& F = mFrames[frame];
    for (int i = 0; i < 8; i++) {                                              // :511-513
        H[(size_t)(ndim + i) * odim + ndim + i] += F.prior[i];
        b[ndim + i] += F.prior[i] * F.delta_prior[i];
    }
    std::vector<double> SVec(odim);
    for (int i = 0; i < odim; i++) SVec[i] = std::sqrt(std::fabs(H[(size_t)i * odim + i]) + 10.0);
    for (int i = 0; i < odim; i++) {
        for (int j = 0; j < odim; j++) H[(size_t)i * odim + j] = (1.0 / SVec[i]) * H[(size_t)i * odim + j] * (1.0 / SVec[j]);
        b[i] = (1.0 / SVec[i]) * b[i];
    }
    double hpi[64], hinv[64];
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) hpi[i * 8 + j] = H[(size_t)(ndim + i) * odim + ndim + j];
    inverse8(hpi, hinv);                                                       // Matrix<8,8>::inverse(): partial-pivot LU (:532-535)
    std::vector<double> bli((size_t)ndim * 8);
    for (int i = 0; i < ndim; i++)
        for (int j = 0; j < 8; j++) {
            double s = 0;
            for (int q = 0; q < 8; q++) s += H[(size_t)(ndim + q) * odim + i] * hinv[q * 8 + j];
            bli[(size_t)i * 8 + j] = s;
        }
    for (int i = 0; i < ndim; i++) {                                           // Schur complement, :538-541
        for (int j = 0; j < ndim; j++) {
            double s = 0;
            for (int q = 0; q < 8; q++) s += bli[(size_t)i * 8 + q] * H[(size_t)(ndim + q) * odim + j];
            H[(size_t)i * odim + j] -= s;
        }
        double s = 0;
        for (int q = 0; q < 8; q++) s += bli[(size_t)i * 8 + q] * b[ndim + q];
        b[i] -= s;
    }
    for (int i = 0; i < odim; i++) {                                           // unscale, :544-545
        for (int j = 0; j < odim; j++) H[(size_t)i * odim + j] = SVec[i] * H[(size_t)i * odim + j] * SVec[j];
        b[i] = SVec[i] * b[i];
    }
    std::vector<double> Hn((size_t)ndim * ndim), bn(ndim);
    for (int i = 0; i < ndim; i++) {                                           // :548-549
        for (int j = 0; j < ndim; j++) Hn[(size_t)i * ndim + j] = 0.5 * (H[(size_t)i * odim + j] + H[(size_t)j * odim + i]);
        bn[i] = b[i];
    }
    mMarginalizedHessian.swap(Hn);
    mMarginalizedB.swap(bn);
    removeFrame(frame);                                                        // :552
    computeDelta();                                                            // :598-599
    computeAdjoints();
}

std::vector<int> DSOBundleAdjustment::marginalizeFrames() {                   // BA.cpp:718-742
    std::vector<int> removed;
    for (;;) {
        int f = -1;
        for (int i = 0; i < (int)mFrames.size(); i++) if (mFrames[i].flaggedForMarginalization) { f = i; break; }
        if (f < 0) break;
        removed.push_back(f + (int)removed.size());                             // id at call time of the first removal
        marginalizeFrame(f);
    }
    return removed;
}

double DSOBundleAdjustment::calcMEnergy() const {                             // BA.cpp:2095-2117
    if (mForceAccept) return 0;
    const int N = (int)mFrames.size(), n = 8 * N + CMLHIP_CPARS;
    std::vector<double> d(n, 0.0);
    for (int i = 0; i < 4; i++) d[i] = mCDeltaF[i];
    for (int h = 0; h < N; h++) for (int k = 0; k < 8; k++) d[4 + 8 * h + k] = mFrames[h].delta[k];
    double e = 0;
    for (int i = 0; i < n; i++) {
        double s = 2 * mMarginalizedB[i];
        for (int j = 0; j < n; j++) s += mMarginalizedHessian[(size_t)i * n + j] * d[j];
        e += d[i] * s;
    }
    return std::fabs(e);
}

double DSOBundleAdjustment::calcLEnergy() {                                   // BA.cpp:2119-2208
    if (mForceAccept) return 0;
    cmlhip_ba_accum_in in; std::vector<double> prior, dprior; double cdelta[4], cprior[4];
    fillAccumIn(in, prior, dprior, cdelta, cprior);
    double e = 0; int num = 0;
    if (cmlhip_ba_lin_energy(mCtx, &in, &e, &num)) return 0;
    return e;
}

// ------------------------------------------------------------------------------------------------ device-resident loop
bool DSOBundleAdjustment::beginResident(bool updatePointsOnly) {
    if (!mForceAccept || !mFixLambda) { mError = "resident iterations need forceAccept and fixLambda (every step accepted, BA.h:265-267)"; return false; }
    if (mDisableMarginalization) {                                             // solveSystem zeroes the prior in this mode, BA.cpp:1395-1398
        std::fill(mMarginalizedHessian.begin(), mMarginalizedHessian.end(), 0.0);
        std::fill(mMarginalizedB.begin(), mMarginalizedB.end(), 0.0);
    }
    const int N = (int)mFrames.size();
    HostLap lapB("beginResident");
    double sc[4];
    scales(sc);
    computeDelta();
    std::vector<cmlhip_ba_pair> pairs;
    framePairs(pairs);
    lapB("delta+pairs");
    int rc = cmlhip_ba_set_arithmetic(mCtx, mRelaxedArithmetic ? CMLHIP_ARITH_RELAXED : CMLHIP_ARITH_EXACT);
    if (rc) return fail("cmlhip_ba_set_arithmetic", rc);
    rc = cmlhip_ba_set_resident_outputs(mCtx, (mLeanResidentOutputs && !mKeepResidualEnergies) ? CMLHIP_RESIDENT_OUTPUTS_LEAN : CMLHIP_RESIDENT_OUTPUTS_FULL);
    if (rc) return fail("cmlhip_ba_set_resident_outputs", rc);
    rc = setPairs(pairs);                                    // (unchanged since run()'s preamble: not sent again — a new set would also discard the pass's pair tiles)
    if (rc) return fail("cmlhip_ba_set_pairs", rc);
    std::vector<double> prior(8 * (size_t)N), dprior(8 * (size_t)N);
    for (int i = 0; i < N; i++) for (int k = 0; k < 8; k++) { prior[8 * i + k] = mFrames[i].prior[k]; dprior[8 * i + k] = mFrames[i].delta_prior[k]; }
    double cdelta[4] = {mCDeltaF[0], mCDeltaF[1], mCDeltaF[2], mCDeltaF[3]}, cprior[4] = {mCPriorValue, mCPriorValue, mCPriorValue, mCPriorValue};
    cmlhip_ba_accum_in in{mAdHost.data(), mAdTarget.data(), mAdHTdeltaF.data(), cdelta, prior.data(), dprior.data(), cprior};
    std::vector<cmlhip_ba_frame_state> fs(N);
    for (int i = 0; i < N; i++) {
        const DSOFrame& f = mFrames[i];
        std::memcpy(fs[i].eval_q, f.worldToCam_evalPT.q, sizeof fs[i].eval_q);
        std::memcpy(fs[i].eval_t, f.worldToCam_evalPT.t, sizeof fs[i].eval_t);
        std::memcpy(fs[i].state, f.state, sizeof fs[i].state);
        std::memcpy(fs[i].state_zero, f.state_zero, sizeof fs[i].state_zero);
        std::memcpy(fs[i].prior_zero, f.prior_zero, sizeof fs[i].prior_zero);
        fs[i].ab_exposure = f.ab_exposure;
        fs[i].fix_pose = updatePointsOnly ? 1 : 0;
        fs[i].pad = 0;
    }
    lapB("states");
    std::vector<double> U;
    nullspaceBasis(U);
    lapB("nullspace basis");
    rc = cmlhip_ba_set_resident_state(mCtx, &in, fs.data(), sc, U.data());
    if (rc) return fail("cmlhip_ba_set_resident_state", rc);
    lapB("set_resident_state");
    // marginalisation prior (BA.cpp:1389-1401): HM and the raw bM stay on the device, bM_top follows the frame states there
    rc = cmlhip_ba_set_resident_prior(mCtx, mDisableMarginalization ? nullptr : mMarginalizedHessian.data(), mDisableMarginalization ? nullptr : mMarginalizedB.data());
    if (rc) return fail("cmlhip_ba_set_resident_prior", rc);
    // hybrid ORB term (BA.cpp:1327-1329, 2574-2729) evaluated and mixed on the device inside every iteration
    const int M = mMixedBundleAdjustment ? (int)(mIndirectPoints.size() / 3) : 0;
    rc = cmlhip_ba_set_resident_indirect(mCtx, M, M ? mIndirectPoints.data() : nullptr, M ? (int)mIndirectObs.size() : 0,
                                         M ? mIndirectObs.data() : nullptr, mPrm.fx, mPrm.fy);
    if (rc) return fail("cmlhip_ba_set_resident_indirect", rc);
    lapB("prior+indirect");
    return true;
}

bool DSOBundleAdjustment::iterateResident(int k, double lambda) {
    if (k > 0) mPairsValid = false;                          // (the device's frame step rewrites the pair records: the next setPairs must send them)
    for (int i = 0; i < k; i++) {
        const int rc = cmlhip_ba_iteration_async(mCtx, lambda);
        if (rc) return fail("cmlhip_ba_iteration_async", rc);
    }
    return true;
}

bool DSOBundleAdjustment::endResident(double* lastEnergy) {
    mPairsValid = false;                                     // (the device's frame step has rewritten the pair records)
    const int N = (int)mFrames.size();
    double sc[4];
    scales(sc);
    std::vector<cmlhip_ba_frame_state> fs(N);
    cmlhip_ba_lin_result last{};
    int rc = cmlhip_ba_get_resident_state(mCtx, fs.data(), nullptr, &last);
    if (rc) return fail("cmlhip_ba_get_resident_state", rc);
    for (int i = 0; i < N; i++) {
        DSOFrame& f = mFrames[i];
        for (int k = 0; k < 10; k++) { f.step[k] = fs[i].state[k] - f.state[k]; }
        f.setState(fs[i].state, sc);
    }
    mFrames.back().frameEnergyTH = last.new_frame_energy_th;                  // setNewFrameEnergyTH of the last pass
    if (lastEnergy) *lastEnergy = last.energy;
    {   // x of the last solve; with the hybrid term also the last indirect solution and the point uncertainties (:2690-2692)
        const int M = mMixedBundleAdjustment ? (int)(mIndirectPoints.size() / 3) : 0;
        const bool mixed = M > 0 && N > 4;
        mX.assign(8 * (size_t)N + CMLHIP_CPARS, 0.0);
        std::vector<double> Jp(3 * (size_t)(mixed ? M : 0));
        if (mixed) mIndirectX.assign(6 * (size_t)N, 0.0);
        rc = cmlhip_ba_get_resident_indirect(mCtx, mX.data(), mixed ? mIndirectX.data() : nullptr, mixed ? Jp.data() : nullptr);
        if (rc) return fail("cmlhip_ba_get_resident_indirect", rc);
        if (mixed) indirectUncertaintyFrom(Jp);
    }
    computeDelta();
    return std::isfinite(last.energy);
}

// run() with the loop resident and ONE host wait: the preamble pass, the resident state, the iterations, the re-anchoring of the newest frame and the
// closing pass are enqueued back to back; cmlhip_ba_finish_run brings everything back in one copy.  (The hybrid term reads its point Jacobians back
// through a path of its own and an empty loop leaves no pose on the device: both take the step-by-step flow, runResidentStepwise.)
bool DSOBundleAdjustment::runResident(bool updatePointsOnly) {
    if (mMixedBundleAdjustment || mNumIterations < 1 || mFrames.empty() || getenv("CMLHOST_RUN_STEPWISE")) return runResidentStepwise(updatePointsOnly);
    double lastEnergy[3];
    const auto T0 = std::chrono::steady_clock::now();
    auto us = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - T0).count(); };
    HostLap lap("run");
    // ---- preamble (BA.cpp:744-802) and the loop's resident state: ONE packed copy (upload scope), then the kernels
    mOutliers.clear();
    mError.clear();
    lastIterations = 0;
    int alivePts = 0;
    for (const auto& p : mPoints) alivePts += p.alive;
    if (alivePts == 0) { mError = "No points..."; return false; }             // :759-762
    computeAdjoints();
    computeDelta();
    int rc = cmlhip_upload_scope_begin(mCtx);
    if (rc) return fail("cmlhip_upload_scope_begin", rc);
    struct ScopeGuard { cmlhip_ctx* c; ~ScopeGuard() { cmlhip_upload_scope_end(c); } } scopeGuard{mCtx};      // (error returns leave no scope open)
    rc = cmlhip_ba_set_resident_outputs(mCtx, (mLeanResidentOutputs && !mKeepResidualEnergies) ? CMLHIP_RESIDENT_OUTPUTS_LEAN : CMLHIP_RESIDENT_OUTPUTS_FULL);   // (the preamble pass already runs lean)
    if (rc) return fail("cmlhip_ba_set_resident_outputs", rc);
    if (!uploadWindow()) return false;
    {
        std::vector<cmlhip_ba_pair> pairs;
        framePairs(pairs);
        if ((rc = setPairs(pairs))) return fail("cmlhip_ba_set_pairs", rc);
    }
    rc = cmlhip_upload_scope_end(mCtx);                      // window + pair records: one packed copy
    if (rc) return fail("cmlhip_upload_scope_end", rc);
    lastRunUs[0] = us();
    lap("window committed");
    rc = cmlhip_ba_linearize_apply(mCtx, nullptr);           // linearizeAll(false) + applyActiveRes(true), :785-790: enqueued (the device works on it while the
    if (rc) return fail("cmlhip_ba_linearize_apply", rc);    // loop's state is prepared below); its tail rides in the first solve launch
    lastRunUs[1] = us() - lastRunUs[0];
    rc = cmlhip_upload_scope_begin(mCtx);
    if (rc) return fail("cmlhip_upload_scope_begin", rc);
    if (!beginResident(updatePointsOnly)) return false;      // adjoints, frame states, prior, gauge basis: the second (small) packed copy
    rc = cmlhip_ba_resident_convergence(mCtx, mThOptIterations);             // `if (canbreak && it >= 1) break`, BA.cpp:879
    if (rc) return fail("cmlhip_ba_resident_convergence", rc);
    rc = cmlhip_upload_scope_end(mCtx);
    if (rc) return fail("cmlhip_upload_scope_end", rc);
    const double t_begin = us();
    lastRunUs[2] = t_begin - lastRunUs[0] - lastRunUs[1];
    lap("resident state staged");
    if (!iterateResident(mNumIterations, mFixedLambda)) return false;
    lastLambda = mFixedLambda;
    const double t_enq = us();
    lastRunUs[3] = t_enq - t_begin;
    lap("iterations enqueued");
    // ---- everything back in one copy
    const int N = (int)mFrames.size(), R = (int)mActive.size();
    double sc[4];
    scales(sc);
    std::vector<cmlhip_ba_frame_state> fs(N);
    std::vector<double> pre(7 * (size_t)N), en(mNumIterations, 0.0);
    cmlhip_ba_lin_result first{}, last{}, lr{};
    int its = 0;
    mX.assign(8 * (size_t)N + CMLHIP_CPARS, 0.0);
    // the closing pass's per-residual state / good flag come back as ONE byte each, already in this object's order, and HdiF as one float per point
    std::vector<unsigned char> sg(R);
    std::vector<float> hdiv(mActivePoints.size());
    cmlhip_ba_resident_out ro{fs.data(), pre.data(), &first, &last, &its, en.data(), (int)en.size(), mX.data(), sg.data(), hdiv.data()};
    std::vector<int> st, ns;
    std::vector<float> e, ne, nw;
    std::vector<unsigned char> good;
    if (mKeepResidualEnergies) { ns.resize(R); e.resize(R); ne.resize(R); nw.resize(R); }
    std::vector<double> idp(mActivePoints.size());
    rc = cmlhip_ba_finish_run(mCtx, 1, &ro, &lr, nullptr, mKeepResidualEnergies ? ns.data() : nullptr, mKeepResidualEnergies ? e.data() : nullptr,
                              mKeepResidualEnergies ? ne.data() : nullptr, mKeepResidualEnergies ? nw.data() : nullptr, nullptr, idp.data(), nullptr);
    if (rc && rc != CMLHIP_ERR_NONFINITE) return fail("cmlhip_ba_finish_run", rc);
    const double t_end = us();
    lastRunUs[4] = t_end - t_enq;
    lap("finish_run returned");
    mPairsValid = false;                                     // (the device's frame step and the re-anchoring have rewritten the pair records)
    statEnergyP.push_back(first.energy / std::max<size_t>(1, mActive.size()));      // the preamble's entry, :792
    for (int i = 0; i < N; i++) {                                              // the loop's frame states (endResident)
        DSOFrame& f = mFrames[i];
        for (int k = 0; k < 10; k++) f.step[k] = fs[i].state[k] - f.state[k];
        f.setState(fs[i].state, sc);
    }
    lastIterations = its;
    for (int i = 0; i < its && i < (int)en.size(); i++) statEnergyP.push_back(en[i]);
    if (!std::isfinite(last.energy)) { mError = "non finite energy"; return false; }
    // re-anchor the newest frame's evaluation point, :885-894 — at the pose the DEVICE re-anchored at (its own PRE_worldToCam of the last step)
    DSOFrame& fb = mFrames.back();
    double nz[10] = {0};
    nz[6] = fb.state[6]; nz[7] = fb.state[7];
    SE3 anchor;
    std::memcpy(anchor.q, &pre[7 * (size_t)(N - 1)], sizeof anchor.q);
    std::memcpy(anchor.t, &pre[7 * (size_t)(N - 1) + 4], sizeof anchor.t);
    fb.setEvalPT(anchor, nz, sc);
    lap("frames set");
    computeAdjoints();
    computeDelta();
    lap("adjoints+delta");
    lastEnergy[0] = lr.energy; lastEnergy[1] = lastEnergy[2] = 0;
    fb.frameEnergyTH = lr.new_frame_energy_th;                                 // setNewFrameEnergyTH of the closing pass, :1610
    closingBookkeeping(st, good, ns, e, ne, nw, sg.data());
    lap("closing bookkeeping");
    if (!std::isfinite(lastEnergy[0])) { mError = "Not finite energy"; return false; }
    for (size_t k = 0; k < mActivePoints.size(); k++) {                        // MapPoint::setReferenceInverseDepth / setInverseDepthHessian
        DSOPoint& P = mPoints[mActivePoints[k]];
        P.idepth = idp[k];
        P.idepth_zero = (float)idp[k];
        const float hdi = hdiv[k];
        P.idepth_hessian = hdi > 0 ? 1.0f / hdi : 0.f;
    }
    lastRunUs[5] = us() - t_end;
    lap("bookkeeping done");
    return true;
}

bool DSOBundleAdjustment::runResidentStepwise(bool updatePointsOnly) {
    double lastEnergy[3];
    const auto T0 = std::chrono::steady_clock::now();
    HostLap lap("run (stepwise)");
    auto us = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - T0).count(); };
    if (!runPreamble(lastEnergy)) return false;
    lap("preamble done");
    const double t_pre = us();
    lastRunUs[1] = t_pre - lastRunUs[0];
    if (!beginResident(updatePointsOnly)) return false;
    lap("beginResident done");
    const double t_begin = us();
    lastRunUs[2] = t_begin - t_pre;
    int rc = cmlhip_ba_resident_convergence(mCtx, mThOptIterations);         // `if (canbreak && it >= 1) break`, BA.cpp:879
    if (rc) return fail("cmlhip_ba_resident_convergence", rc);
    if (!iterateResident(mNumIterations, mFixedLambda)) return false;
    lastLambda = mFixedLambda;
    double e = 0;
    lap("iterations enqueued");
    const double t_enq = us();
    lastRunUs[3] = t_enq - t_begin;
    if (!endResident(&e)) { mError = "non finite energy"; return false; }
    lap("endResident done");
    const double t_end = us();
    lastRunUs[4] = t_end - t_enq;
    std::vector<double> en(mNumIterations > 0 ? mNumIterations : 1, 0.0);
    int its = 0;
    rc = cmlhip_ba_get_resident_log(mCtx, &its, en.data(), (int)en.size());
    if (rc) return fail("cmlhip_ba_get_resident_log", rc);
    lastIterations = its;
    for (int i = 0; i < its && i < (int)en.size(); i++) statEnergyP.push_back(en[i]);
    lastEnergy[0] = e;
    const bool ok = runEpilogue(lastEnergy);
    lap("epilogue done");
    lastRunUs[5] = us() - t_end;
    return ok;
}

}  // namespace cml_amd
