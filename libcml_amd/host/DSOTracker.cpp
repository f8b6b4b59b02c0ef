// DSOTracker.cpp — host mirror of DSOTracker::optimize (TR.cpp = src/cml/optimization/dso/DSOTracker.cpp).
#include "DSOTracker.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace cml_amd {

bool ldltSolveSmall(const double* Ain, const double* b, int n, double* x) {
    double A[64], temp[8];
    int tr[8];
    for (int i = 0; i < n * n; i++) A[i] = Ain[i];
#define M(i, j) A[(i) * n + (j)]
    if (n == 1) tr[0] = 0;
    else
        for (int k = 0; k < n; k++) {
            int big = k; double best = std::fabs(M(k, k));
            for (int i = k + 1; i < n; i++) if (std::fabs(M(i, i)) > best) { best = std::fabs(M(i, i)); big = i; }
            tr[k] = big;
            if (k != big) {
                for (int j = 0; j < k; j++) std::swap(M(k, j), M(big, j));
                for (int i = big + 1; i < n; i++) std::swap(M(i, k), M(i, big));
                std::swap(M(k, k), M(big, big));
                for (int i = k + 1; i < big; i++) std::swap(M(i, k), M(big, i));
            }
            if (k > 0) {
                double s = 0;
                for (int j = 0; j < k; j++) { temp[j] = M(j, j) * M(k, j); s += M(k, j) * temp[j]; }
                M(k, k) -= s;
                for (int i = k + 1; i < n; i++) { double s2 = 0; for (int j = 0; j < k; j++) s2 += M(i, j) * temp[j]; M(i, k) -= s2; }
            }
            const double akk = M(k, k);
            if (k == 0 && !(std::fabs(akk) > 0.0)) { for (int j = 0; j < n; j++) tr[j] = j; break; }
            if (std::fabs(akk) > 0.0) for (int i = k + 1; i < n; i++) M(i, k) /= akk;
        }
    for (int i = 0; i < n; i++) x[i] = b[i];
    for (int k = 0; k < n; k++) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
    for (int i = 0; i < n; i++) { double s = x[i]; for (int j = 0; j < i; j++) s -= M(i, j) * x[j]; x[i] = s; }
    for (int i = 0; i < n; i++) x[i] = (std::fabs(M(i, i)) > 2.2250738585072014e-308) ? x[i] / M(i, i) : 0.0;
    for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int j = i + 1; j < n; j++) s -= M(j, i) * x[j]; x[i] = s; }
    for (int k = n - 1; k >= 0; k--) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
#undef M
    for (int i = 0; i < n; i++) if (!std::isfinite(x[i])) return false;
    return true;
}

void inverseSmall(const double* Ain, int n, double* Ai) {
    double A[64];
    for (int i = 0; i < n * n; i++) A[i] = Ain[i];
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) Ai[i * n + j] = (i == j);
    for (int k = 0; k < n; k++) {
        int p = k;
        for (int i = k + 1; i < n; i++) if (std::fabs(A[i * n + k]) > std::fabs(A[p * n + k])) p = i;
        if (p != k) for (int j = 0; j < n; j++) { std::swap(A[k * n + j], A[p * n + j]); std::swap(Ai[k * n + j], Ai[p * n + j]); }
        const double d = A[k * n + k];
        for (int j = 0; j < n; j++) { A[k * n + j] /= d; Ai[k * n + j] /= d; }
        for (int i = 0; i < n; i++)
            if (i != k) {
                const double f = A[i * n + k];
                if (f != 0) for (int j = 0; j < n; j++) { A[i * n + j] -= f * A[k * n + j]; Ai[i * n + j] -= f * Ai[k * n + j]; }
            }
    }
}

bool DSOTracker::makeCoarseDepthL0(uint64_t ref_image_id, int levels, const double* pts, int n, int* n_out) {
    const int rc = cmlhip_tracker_make_coarse_depth(mCtx, ref_image_id, levels, pts, n, n_out);
    if (rc) { mError = std::string("cmlhip_tracker_make_coarse_depth: ") + cmlhip_last_error(mCtx); return false; }
    return true;
}

DSOTracker::Residual DSOTracker::optimize(uint64_t new_image_id, int pyramidLevels, SE3& refToNewInOut,
                                          const Exposure& refExposure, Exposure& currentExposure) {
    SE3 currentRefToNew = refToNewInOut;          // the reference writes `camera` only on the normal exit (TR.cpp:237): an aborted call leaves the caller's pose untouched
    const int maxIterations[] = {10, 20, 50, 50, 50};                     // TR.cpp:23
    int maxLevel = std::min(pyramidLevels - 1, 4);
    if (maxLevelOverride >= 0) maxLevel = std::min(maxLevel, maxLevelOverride);
    Residual oldR, newR;
    for (Residual* r : {&oldR, &newR}) {
        r->numTermsInE.assign(maxLevel + 1, 0); r->numRobust.assign(maxLevel + 1, 0); r->numSaturated.assign(maxLevel + 1, 0);
        r->E.assign(maxLevel + 1, 0.0); r->iterations.assign(maxLevel + 1, 0);
    }
    std::vector<double> levelCutoffRepeat(maxLevel + 1, 1.0);
    bool haveRepeated = false;
    SE3 newRefToNew;
    Exposure newExposure = currentExposure;
    double H[64] = {0}, bvec[8] = {0}, Hn[64], bn[8];
    cmlhip_tracker_params prm;
    prm.huber = (float)mHuberThreshold; prm.cutoff_base = (float)mCutoffThreshold;
    prm.scale_rot = (float)mScaleRotation; prm.scale_trans = (float)mScaleTranslation; prm.scale_a = (float)mScaleLightA; prm.scale_b = (float)mScaleLightB;

    auto eval = [&](int level, const SE3& T, const Exposure& ex, Residual& out, bool wantH, double* Ho, double* bo) -> bool {
        double R[9], K[4], aff[2];
        T.matrix(R);
        const double d = (double)(1 << level);
        K[0] = mK[0] / d; K[1] = mK[1] / d; K[2] = (mK[2] + 0.5) / d - 0.5; K[3] = (mK[3] + 0.5) / d - 0.5;   // InternalCalibration.h:116-127
        refExposure.to(ex, aff[0], aff[1]);
        prm.cutoff = (float)(mCutoffThreshold * levelCutoffRepeat[level]);
        cmlhip_tracker_result tr;
        const int rc = evalOverride ? evalOverride(evalUser, level, R, T.t, K, aff, refExposure.b, &prm, &tr)
                                    : cmlhip_tracker_eval(mCtx, new_image_id, level, R, T.t, K, aff, refExposure.b, &prm, wantH ? 1 : 0, &tr);
        if (rc && rc != CMLHIP_ERR_NONFINITE) { mError = std::string("cmlhip_tracker_eval: ") + cmlhip_last_error(mCtx); return false; }
        out.E[level] = tr.E; out.numTermsInE[level] = tr.numTermsInE; out.numSaturated[level] = tr.numSaturated; out.numRobust[level] = tr.numRobust;
        for (int k = 0; k < 3; k++) out.flowVector[k] = tr.flow[k];
        if (wantH) { std::memcpy(Ho, tr.H, sizeof tr.H); std::memcpy(bo, tr.b, sizeof tr.b); }
        return true;
    };

    lastSteps.clear();
    for (int level = maxLevel; level >= 0; level--) {
        levelCutoffRepeat[level] = 1;
        if (!eval(level, currentRefToNew, currentExposure, oldR, true, H, bvec)) { oldR.isCorrect = false; return oldR; }
        if (oldR.numTermsInE[level] < 20) { oldR.isCorrect = false; return oldR; }                        // TR.cpp:65-69
        while ((oldR.numSaturated[level] / (double)oldR.numTermsInE[level]) > 0.6 && levelCutoffRepeat[level] < 50) {   // :71-75
            levelCutoffRepeat[level] *= 2;
            if (!eval(level, currentRefToNew, currentExposure, oldR, true, H, bvec)) { oldR.isCorrect = false; return oldR; }
        }
        if (oldR.numTermsInE[level] - oldR.numSaturated[level] < 10) { oldR.isCorrect = false; return oldR; }            // :77-81
        double lambda = 0.01;
        const double lambdaExtrapolationLimit = 0.001;
        for (int iteration = 0; iteration < maxIterations[level]; iteration++) {
            oldR.iterations[level] = iteration + 1;
            double D[64], inc[8] = {0};
            std::memcpy(D, H, sizeof D);
            for (int i = 0; i < 8; i++) D[i * 8 + i] *= (1 + lambda);
            bool ok = true;
            double nb[8];
            for (int i = 0; i < 8; i++) nb[i] = -bvec[i];
            if (mOptimizeA && mOptimizeB) ok = ldltSolveSmall(D, nb, 8, inc);                               // :96-98
            else if (mOptimizeA && !mOptimizeB) {                                                        // :99-102
                double S[49], x7[7];
                for (int i = 0; i < 7; i++) for (int j = 0; j < 7; j++) S[i * 7 + j] = D[i * 8 + j];
                ok = ldltSolveSmall(S, nb, 7, x7);
                for (int i = 0; i < 7; i++) inc[i] = x7[i];
                inc[7] = 0;
            } else if (!mOptimizeA && mOptimizeB) {                                                      // :103-114
                double Hs[64], bs[8], S[49], nb7[7], x7[7];
                std::memcpy(Hs, D, sizeof Hs); std::memcpy(bs, bvec, sizeof bs);
                for (int i = 0; i < 8; i++) Hs[i * 8 + 6] = Hs[i * 8 + 7];
                for (int j = 0; j < 8; j++) Hs[6 * 8 + j] = Hs[7 * 8 + j];
                bs[6] = bs[7];
                for (int i = 0; i < 7; i++) { for (int j = 0; j < 7; j++) S[i * 7 + j] = Hs[i * 8 + j]; nb7[i] = -bs[i]; }
                ok = ldltSolveSmall(S, nb7, 7, x7);
                for (int i = 0; i < 6; i++) inc[i] = x7[i];
                inc[6] = 0; inc[7] = x7[6];
            } else {                                                                                     // :115-119
                double S[36], x6[6];
                for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) S[i * 6 + j] = D[i * 8 + j];
                ok = ldltSolveSmall(S, nb, 6, x6);
                for (int i = 0; i < 6; i++) inc[i] = x6[i];
            }
            if (!ok) { oldR.isCorrect = false; return oldR; }                                            // :121-138
            double extrapFac = 1;
            if (lambda < lambdaExtrapolationLimit) extrapFac = std::sqrt(std::sqrt(lambdaExtrapolationLimit / lambda));
            for (int i = 0; i < 8; i++) inc[i] *= extrapFac;
            double incS[8];
            std::memcpy(incS, inc, sizeof incS);
            for (int i = 0; i < 3; i++) { incS[i] *= mScaleRotation; incS[3 + i] *= mScaleTranslation; }   // the literal lane/scale pairing, :144-148
            incS[6] *= mScaleLightA; incS[7] *= mScaleLightB;
            newRefToNew = SE3::exp(incS) * currentRefToNew;                                              // :155-157
            newExposure = currentExposure;
            newExposure.a += incS[6]; newExposure.b += incS[7];                                          // :159
            if (!eval(level, newRefToNew, newExposure, newR, true, Hn, bn)) { oldR.isCorrect = false; return oldR; }
            const bool accept = (newR.E[level] / (double)newR.numTermsInE[level]) < (oldR.E[level] / (double)oldR.numTermsInE[level]);   // :163
            lastSteps.push_back(Step{level, iteration, accept ? 1 : 0, lambda, newR.E[level], oldR.E[level], newR.numTermsInE[level], oldR.numTermsInE[level]});
            if (accept) {
                std::memcpy(H, Hn, sizeof H); std::memcpy(bvec, bn, sizeof bvec);                        // computeHessian at the accepted state, :166
                const std::vector<int> its = oldR.iterations;
                oldR = newR; oldR.iterations = its;
                currentRefToNew = newRefToNew;
                currentExposure.a = newExposure.a; currentExposure.b = newExposure.b;
                lambda *= 0.5;
            } else {
                lambda *= 4;
            }
            double nrm = 0;
            for (int i = 0; i < 8; i++) nrm += inc[i] * inc[i];
            if (std::sqrt(nrm) < 1e-3) break;                                                             // :176-179
        }
        if (mLastResidual.isCorrect && level < (int)mLastResidual.E.size() && oldR.rmse(level) > 1.5 * mLastResidual.rmse(level)) {   // :183-189
            oldR.isCorrect = false;
            return oldR;
        }
        if (levelCutoffRepeat[level] > 1 && !haveRepeated) { level++; haveRepeated = true; }            // :192-195
    }
    double relA, relB;
    refExposure.to(currentExposure, relA, relB);                                                         // :203
    bool haveGoodLight = true;
    if (mOptimizeA) { if (std::fabs(currentExposure.a) > 1.2) haveGoodLight = false; }
    else if (std::fabs(std::log((float)relA)) > 1.5) haveGoodLight = false;
    if (mOptimizeB) { if (std::fabs(currentExposure.b) > 200) haveGoodLight = false; }
    else if (std::fabs((float)relB) > 200) haveGoodLight = false;
    bool haveGoodPoints = true;
    if ((double)oldR.numSaturated[0] / (double)oldR.numTermsInE[0] > mSaturatedRatioThreshold) haveGoodPoints = false;   // :231-235
    oldR.isCorrect = haveGoodLight;
    oldR.tooManySaturated = haveGoodPoints;                                                              // sic, :240
    oldR.levelCutoffRepeat = levelCutoffRepeat;
    oldR.relAff[0] = relA; oldR.relAff[1] = relB;
    double Hi[64];
    inverseSmall(H, 8, Hi);                                                                              // :243
    for (int k = 0; k < 6; k++) oldR.covariance[k] = Hi[k * 8 + k];
    refToNewInOut = currentRefToNew;
    return oldR;
}

bool DSOTracker::trackWithMotionModel(uint64_t new_image_id, int pyramidLevels, int n_hyp, const SE3* hyp, const Exposure& referenceExposure,
                                      const Exposure& initialExposure, SE3& bestRefToNew, Exposure& bestExposure, Residual& residual,
                                      int* winner, int* tries) {
    residual = Residual();
    bool haveOneGood = false;
    Residual trackingResult, testTrackingResult;
    double achievedRes = std::numeric_limits<double>::max();
    if (winner) *winner = -1;
    auto rmse0 = [](const Residual& r) { return (!r.E.empty() && r.numTermsInE[0] > 0) ? r.rmse() : std::numeric_limits<double>::quiet_NaN(); };
    int i = 0;
    for (; i < n_hyp; i++) {
        SE3 testRefToNew = hyp[i];
        Exposure testExposure = initialExposure;                                           // :268-269
        mLastResidual = trackingResult;                                                    // :272
        testTrackingResult = optimize(new_image_id, pyramidLevels, testRefToNew, referenceExposure, testExposure);
        const double rm = rmse0(testTrackingResult);
        auto adopt = [&]() {
            haveOneGood = true; bestRefToNew = testRefToNew; bestExposure = testExposure; trackingResult = testTrackingResult;
            if (winner) *winner = i;
        };
        if (trackingResult.tooManySaturated == true && testTrackingResult.tooManySaturated == false && testTrackingResult.isCorrect && std::isfinite(rm)) adopt();   // :280-285
        if (testTrackingResult.isCorrect && std::isfinite(rm) && !(rm >= achievedRes)) {   // do we have a new winner? :288-296
            if (trackingResult.tooManySaturated || !testTrackingResult.tooManySaturated) adopt();
        }
        if (haveOneGood) {                                                                 // take over achieved res (always), :299-304
            if (!testTrackingResult.numTermsInE.empty() && testTrackingResult.numTermsInE[0] > 0 && rm < achievedRes) achievedRes = rm;
        }
        const float setting_reTrackThreshold = 1.5f;
        if (haveOneGood && achievedRes < mLastCoarseRMSE * setting_reTrackThreshold) { i++; break; }   // :306-309
        if (haveOneGood && i >= 50) { i++; break; }                                        // :311-313
    }
    if (tries) *tries = i;
    if (!haveOneGood) {
        if ((mFailureMode == 1 || mFailureMode == 2) && n_hyp > 0) {                       // :324-352 (mode 2: the caller then overrides the camera with frame * cameras[0])
            bestRefToNew = hyp[0];
            bestExposure = initialExposure;
            mLastResidual = trackingResult;
            trackingResult = optimize(new_image_id, pyramidLevels, bestRefToNew, referenceExposure, bestExposure);
            if (winner) *winner = 0;
            haveOneGood = true;
        } else {
            return false;                                                                  // :353-355
        }
    } else {
        mLastCoarseRMSE = achievedRes;                                                     // :357 (mFirstRMSE of the reference keyframe is the caller's)
    }
    residual = trackingResult;
    return haveOneGood;
}

int DSOTracker::launchPending(double bar) {
    PendingBatch& B = mPending;
    int rc = cmlhip_tracker_set_early_exit(mCtx, bar);
    if (rc) { mError = std::string("cmlhip_tracker_set_early_exit: ") + cmlhip_last_error(mCtx); return rc; }
    const double refE[3] = {B.ref.a, B.ref.b, B.ref.t}, initE[3] = {B.init.a, B.init.b, B.init.t};
    rc = cmlhip_tracker_optimize_batch_async(mCtx, B.image, B.levels, mK, refE, initE, &B.prm, mOptimizeA ? 1 : 0, mOptimizeB ? 1 : 0,
                                             mSaturatedRatioThreshold, (int)B.H.size(), B.H.data());
    (void)cmlhip_tracker_set_early_exit(mCtx, 0.0);
    if (rc) mError = std::string("cmlhip_tracker_optimize_batch_async: ") + cmlhip_last_error(mCtx);
    return rc;
}

bool DSOTracker::trackWithMotionModelBatchedEnqueue(uint64_t new_image_id, int pyramidLevels, int n_hyp, const SE3* hyp, const Exposure& referenceExposure,
                                                    const Exposure& initialExposure) {
    PendingBatch& B = mPending;
    if (B.active) { mError = "trackWithMotionModelBatchedEnqueue: the previous batch was not finished"; return false; }
    if (n_hyp <= 0) return false;
    B.image = new_image_id; B.pyramidLevels = pyramidLevels; B.ref = referenceExposure; B.init = initialExposure;
    B.hyp.assign(hyp, hyp + n_hyp);
    B.H.resize(n_hyp);
    for (int i = 0; i < n_hyp; i++) { hyp[i].matrix(B.H[i].R); std::memcpy(B.H[i].t, hyp[i].t, sizeof B.H[i].t); }
    B.prm.huber = (float)mHuberThreshold; B.prm.cutoff_base = (float)mCutoffThreshold; B.prm.cutoff = B.prm.cutoff_base;
    B.prm.scale_rot = (float)mScaleRotation; B.prm.scale_trans = (float)mScaleTranslation; B.prm.scale_a = (float)mScaleLightA; B.prm.scale_b = (float)mScaleLightB;
    B.levels = std::min(pyramidLevels, 5);
    if (maxLevelOverride >= 0) B.levels = std::min(B.levels, maxLevelOverride + 1);
    // early exit on the device: the bar of the replay's break (lastCoarseRMSE * setting_reTrackThreshold), valid while the selection state is
    // still empty — i.e. for the FIRST try only (the try the motion model puts its best guess in).  Results behind it come back "given up"
    // (n_steps = -1); the replay never reads them unless its own test of try 0 disagrees with the device's by a rounding, in which
    // case the batch runs again in full.
    B.bar = (mBatchedEarlyExit && n_hyp > 1 && std::isfinite((double)mLastCoarseRMSE) && mLastCoarseRMSE > 0) ? (double)(mLastCoarseRMSE * 1.5f) : 0.0;
    if (launchPending(B.bar)) return false;
    B.active = true;
    return true;
}

bool DSOTracker::trackWithMotionModelBatchedFinish(SE3& bestRefToNew, Exposure& bestExposure, Residual& residual, int* winner, int* tries, bool* retried) {
    PendingBatch& B = mPending;
    residual = Residual();
    if (winner) *winner = -1;
    if (tries) *tries = 0;
    if (retried) *retried = false;
    if (!B.active) { mError = "trackWithMotionModelBatchedFinish: no batch in flight"; return false; }
    B.active = false;
    const int n_hyp = (int)B.H.size(), levels = B.levels;
    const Exposure& initialExposure = B.init;
    const Exposure& referenceExposure = B.ref;
    const SE3* hyp = B.hyp.data();
    std::vector<cmlhip_tracker_opt_result> R(n_hyp);
  for (int attempt = 0; attempt < 2; attempt++) {
    bool hit_given_up = false;
    int rc = cmlhip_tracker_optimize_wait(mCtx, R.data());
    if (rc) { mError = std::string("cmlhip_tracker_optimize_batch: ") + cmlhip_last_error(mCtx); return false; }
    auto toResidual = [&](const cmlhip_tracker_opt_result& r) {
        Residual o;
        o.E.assign(r.E, r.E + levels); o.numTermsInE.assign(r.numTermsInE, r.numTermsInE + levels); o.numSaturated.assign(r.numSaturated, r.numSaturated + levels);
        o.numRobust.assign(r.numRobust, r.numRobust + levels); o.iterations.assign(r.iterations, r.iterations + levels);
        o.levelCutoffRepeat.assign(r.levelCutoffRepeat, r.levelCutoffRepeat + levels);
        for (int k = 0; k < 3; k++) o.flowVector[k] = r.flow[k];
        o.isCorrect = r.isCorrect != 0; o.tooManySaturated = r.tooManySaturated != 0;
        o.relAff[0] = r.relAff[0]; o.relAff[1] = r.relAff[1];
        for (int k = 0; k < 6; k++) o.covariance[k] = r.covariance[k];
        return o;
    };
    bool haveOneGood = false;
    Residual trackingResult;
    double achievedRes = std::numeric_limits<double>::max();
    int i = 0;
    for (; i < n_hyp; i++) {
        if (R[i].n_steps < 0) { hit_given_up = true; break; }                            // given up on the device behind try 0's early exit: not a result
        Residual test = toResidual(R[i]);
        double rm = (test.numTermsInE[0] > 0) ? test.rmse() : std::numeric_limits<double>::quiet_NaN();
        if (trackingResult.isCorrect) {                                                    // mLastResidual = trackingResult, TR.cpp:183-189
            for (int p = 0; p < R[i].n_pass; p++) {
                const int lv = R[i].pass_level[p];
                if (R[i].pass_rmse[p] > 1.5 * trackingResult.rmse(lv)) {
                    test.isCorrect = false; test.tooManySaturated = true;                  // the try returns at that level
                    rm = (lv == 0) ? R[i].pass_rmse[p] : std::numeric_limits<double>::quiet_NaN();   // its level-0 slot is only filled when it got there
                    if (lv != 0) test.numTermsInE[0] = 0;
                    break;
                }
            }
        }
        auto adopt = [&]() {
            haveOneGood = true; bestRefToNew = SE3::fromRt(R[i].R, R[i].t); bestExposure = initialExposure; bestExposure.a = R[i].a; bestExposure.b = R[i].b;
            trackingResult = test;
            if (winner) *winner = i;
        };
        if (trackingResult.tooManySaturated == true && test.tooManySaturated == false && test.isCorrect && std::isfinite(rm)) adopt();   // :280-285
        if (test.isCorrect && std::isfinite(rm) && !(rm >= achievedRes)) {                 // :288-296
            if (trackingResult.tooManySaturated || !test.tooManySaturated) adopt();
        }
        if (haveOneGood && test.numTermsInE[0] > 0 && rm < achievedRes) achievedRes = rm;  // :299-304
        const float setting_reTrackThreshold = 1.5f;
        if (haveOneGood && achievedRes < mLastCoarseRMSE * setting_reTrackThreshold) { i++; break; }   // :306-309
        if (haveOneGood && i >= 50) { i++; break; }                                        // :311-313
    }
    if (hit_given_up && attempt == 0) {                      // (the device's test of try 0 and this one differ by a rounding: all hypotheses, in full)
        if (winner) *winner = -1;
        if (retried) *retried = true;
        if (launchPending(0.0)) return false;
        continue;
    }
    if (tries) *tries = i;
    if (!haveOneGood) {
        if ((mFailureMode == 1 || mFailureMode == 2) && n_hyp > 0) {                       // :324-352: optimize(cameras[0]) with mLastResidual = trackingResult (not correct: no abort)
            bestRefToNew = hyp[0];                                                         // the reference re-runs optimize() on cameras[0]; an aborted run leaves the pose at cameras[0]
            bestExposure = initialExposure;
            mLastResidual = trackingResult;
            trackingResult = optimize(B.image, B.pyramidLevels, bestRefToNew, referenceExposure, bestExposure);
            if (winner) *winner = 0;
            if (retried) *retried = true;                    // (the pose is not the batch's first result: a trace speculated on it must not be kept)
            haveOneGood = true;
        } else {
            return false;
        }
    } else {
        mLastCoarseRMSE = achievedRes;
    }
    residual = trackingResult;
    return haveOneGood;
  }
    return false;
}

bool DSOTracker::trackWithMotionModelBatched(uint64_t new_image_id, int pyramidLevels, int n_hyp, const SE3* hyp, const Exposure& referenceExposure,
                                             const Exposure& initialExposure, SE3& bestRefToNew, Exposure& bestExposure, Residual& residual,
                                             int* winner, int* tries) {
    residual = Residual();
    if (winner) *winner = -1;
    if (tries) *tries = 0;
    if (n_hyp <= 0) return false;
    // Policy.  Round 2 (one workgroup per hypothesis: ~1 ms per launch whatever the number of hypotheses, a host-driven optimize ~0.3 ms)
    // tried the FIRST hypothesis alone through the host-driven loop and launched the batch only when it did not end the search.  Since
    // round 3 a hypothesis is spread over up to 8 workgroups and the whole batch costs what one device-resident optimize costs (0.21 ms
    // for one, 0.24 ms for fifty: DESIGN §7) — less than the 14-25 evaluations of a host-driven loop with a launch, a mapped-memory poll
    // and host algebra each (0.40 ms per frame in the sequence of round 4).  So the batch runs at once; the hypotheses behind the
    // reference's early exit are computed speculatively and discarded by the replay.  mBatchedFirstAlone restores the old order.
    if (mBatchedFirstAlone) {
        SE3 T0 = hyp[0];
        Exposure e0 = initialExposure;
        mLastResidual = Residual();
        const Residual r0 = optimize(new_image_id, pyramidLevels, T0, referenceExposure, e0);
        const double rm0 = (!r0.numTermsInE.empty() && r0.numTermsInE[0] > 0) ? r0.rmse() : std::numeric_limits<double>::quiet_NaN();
        const bool good0 = r0.isCorrect && std::isfinite(rm0);                              // the two adoption tests of :280-296 on an empty history
        if (n_hyp == 1 || (good0 && rm0 < mLastCoarseRMSE * 1.5f)) {
            if (winner) *winner = good0 ? 0 : -1;
            if (tries) *tries = 1;
            if (good0) { bestRefToNew = T0; bestExposure = e0; residual = r0; mLastCoarseRMSE = rm0; return true; }
            if (n_hyp == 1 && !(mFailureMode == 1 || mFailureMode == 2)) return false;
        }
    }
    if (!trackWithMotionModelBatchedEnqueue(new_image_id, pyramidLevels, n_hyp, hyp, referenceExposure, initialExposure)) return false;
    return trackWithMotionModelBatchedFinish(bestRefToNew, bestExposure, residual, winner, tries);
}

}  // namespace cml_amd
