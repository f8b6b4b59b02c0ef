// DSOTracker.h — host-side mirror of CML::Optimization::DSOTracker (src/cml/optimization/dso/DSOTracker.h:200-520,
// DSOTracker.cpp:15-246) over the C ABI: the coarse-to-fine Levenberg-Marquardt control flow stays on the host,
// computeResidual + computeHessian are one fused device launch (cmlhip_tracker_eval).
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/cmlhip.h"
#include "se3.h"

namespace cml_amd {

class DSOTracker {
public:
    struct Residual {                                   // DSOTracker.h:202-232
        std::vector<double> E;
        std::vector<int> numTermsInE, numSaturated, numRobust, iterations;
        double flowVector[3] = {0, 0, 0};
        bool isCorrect = false, tooManySaturated = true;
        std::vector<double> levelCutoffRepeat;
        double relAff[2] = {0, 0};
        double covariance[6] = {999999, 999999, 999999, 999999, 999999, 999999};
        double rmse(size_t i = 0) const { return E[i] / (double)numTermsInE[i]; }
    };

    // one Levenberg-Marquardt trial of optimize(): the accept / reject decision of TR.cpp:163 and what it compared
    struct Step { int level, iteration, accept; double lambda, E_new, E_old; int n_new, n_old; };
    // computeResidual + computeHessian provider.  Default: cmlhip_tracker_eval on the device.  Tests install the oracle's evaluation
    // here to compare the control flow of optimize() with the oracle's restatement on identical numbers.
    typedef int (*EvalFn)(void* user, int level, const double R[9], const double t[3], const double K[4], const double aff[2], double b0,
                          const cmlhip_tracker_params* prm, cmlhip_tracker_result* out);

    explicit DSOTracker(cmlhip_ctx* ctx) : mCtx(ctx) {}

    // parameters, DSOTracker.h:473-520
    double mHuberThreshold = 9.0, mCutoffThreshold = 20.0;
    double mScaleRotation = 1.0, mScaleTranslation = 0.5, mScaleLightA = 10.0, mScaleLightB = 1000.0;
    bool mOptimizeA = true, mOptimizeB = true, mBackupSolver = false;
    double mSaturatedRatioThreshold = 0.33;
    int maxLevelOverride = -1;
    int mFailureMode = 0;                               // DSOTracker.h:517
    bool mBatchedEarlyExit = true;                      // trackWithMotionModelBatched: hypothesis 0 may end the batch on the device (cmlhip_tracker_set_early_exit with lastCoarseRMSE * 1.5: the break of DSOTracker.h:306-309 behind the first try)
    bool mBatchedFirstAlone = false;                    // trackWithMotionModelBatched: try hypothesis 0 through the host-driven loop before launching the batch (the round-2 policy)
    double mLastCoarseRMSE = 100;                       // DSOTracker.h:470
    Residual mLastResidual;                             // set by trackWithMotionModel before every try (DSOTracker.h:272), read by optimize (TR.cpp:183-189)
    std::vector<Step> lastSteps;                        // the trials of the last optimize()
    EvalFn evalOverride = nullptr; void* evalUser = nullptr;

    void setCalibration(double fx, double fy, double cx, double cy) { mK[0] = fx; mK[1] = fy; mK[2] = cx; mK[3] = cy; }
    // makeCoarseDepthL0 (DSOTracker.cpp:494-724): pts = n x {Ku,Kv,new_idepth,weight} projected by the caller (:521-540)
    bool makeCoarseDepthL0(uint64_t ref_image_id, int levels, const double* pts, int n, int* n_out);
    // optimize (DSOTracker.cpp:15-246): refToNew and currentExposure are updated in place
    Residual optimize(uint64_t new_image_id, int pyramidLevels, SE3& refToNew, const Exposure& referenceExposure, Exposure& currentExposure);
    // trackWithMotionModel (DSOTracker.h:238-383): the hypothesis loop around optimize.  hyp = reference->getCamera().to(c) for every
    // camera c of Map::multiConstantVelocityMotionModel (the caller owns the map).  Every try starts from initialExposure's
    // parameters.  On success bestRefToNew / bestExposure / residual hold the adopted try (the caller applies
    // frame->setCamera(reference.compose(bestRefToNew)), setExposureParameters); *tries = hypotheses run, *winner = adopted index.
    bool trackWithMotionModel(uint64_t new_image_id, int pyramidLevels, int n_hyp, const SE3* hyp, const Exposure& referenceExposure,
                              const Exposure& initialExposure, SE3& bestRefToNew, Exposure& bestExposure, Residual& residual,
                              int* winner, int* tries);
    // The same procedure with every hypothesis optimised SIDE BY SIDE on the device (cmlhip_tracker_optimize_batch: one launch, one
    // readback), then the reference's sequential winner selection (DSOTracker.h:262-313) replayed on the results, including the
    // abort of TR.cpp:183-189 (a try whose level rmse exceeds 1.5 x the best try's so far counts as failed), which the kernel
    // reports per level pass.  Hypotheses beyond the reference's early exit are computed speculatively and discarded.
    bool trackWithMotionModelBatched(uint64_t new_image_id, int pyramidLevels, int n_hyp, const SE3* hyp, const Exposure& referenceExposure,
                                     const Exposure& initialExposure, SE3& bestRefToNew, Exposure& bestExposure, Residual& residual,
                                     int* winner, int* tries);
    // The same in two halves, for callers that enqueue more work behind the batch before they wait (one host wait per tracked frame: the immature
    // points are traced behind it, DSOTracer::traceNewCoarseTrackedAsync): Enqueue launches the batch and returns; Finish waits, then replays the
    // selection exactly as trackWithMotionModelBatched does.  retried (optional): the replay asked for a result the device's early exit had given
    // up and the whole batch ran again, synchronously.
    bool trackWithMotionModelBatchedEnqueue(uint64_t new_image_id, int pyramidLevels, int n_hyp, const SE3* hyp, const Exposure& referenceExposure,
                                            const Exposure& initialExposure);
    bool trackWithMotionModelBatchedFinish(SE3& bestRefToNew, Exposure& bestExposure, Residual& residual, int* winner, int* tries, bool* retried = nullptr);
    const std::string& lastError() const { return mError; }

private:
    struct PendingBatch {
        bool active = false;
        uint64_t image = 0; int pyramidLevels = 0, levels = 0;
        std::vector<SE3> hyp;
        std::vector<cmlhip_tracker_hypothesis> H;
        cmlhip_tracker_params prm;
        Exposure ref, init;
        double bar = 0.0;
    } mPending;
    int launchPending(double bar);                            // cmlhip_tracker_set_early_exit + cmlhip_tracker_optimize_batch_async on mPending
    cmlhip_ctx* mCtx;
    double mK[4] = {1, 1, 0, 0};
    std::string mError;
};

// x = A.ldlt().solve(b), Eigen 3.4.0 semantics (Cholesky/LDLT.h:300-396,560-600), n <= 8
bool ldltSolveSmall(const double* A, const double* b, int n, double* x);
void inverseSmall(const double* A, int n, double* Ai);

}  // namespace cml_amd
