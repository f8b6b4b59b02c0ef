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

    explicit DSOTracker(cmlhip_ctx* ctx) : mCtx(ctx) {}

    // parameters, DSOTracker.h:473-520
    double mHuberThreshold = 9.0, mCutoffThreshold = 20.0;
    double mScaleRotation = 1.0, mScaleTranslation = 0.5, mScaleLightA = 10.0, mScaleLightB = 1000.0;
    bool mOptimizeA = true, mOptimizeB = true, mBackupSolver = false;
    double mSaturatedRatioThreshold = 0.33;
    int maxLevelOverride = -1;

    void setCalibration(double fx, double fy, double cx, double cy) { mK[0] = fx; mK[1] = fy; mK[2] = cx; mK[3] = cy; }
    // makeCoarseDepthL0 (DSOTracker.cpp:494-724): pts = n x {Ku,Kv,new_idepth,weight} projected by the caller (:521-540)
    bool makeCoarseDepthL0(uint64_t ref_image_id, int levels, const double* pts, int n, int* n_out);
    // optimize (DSOTracker.cpp:15-246): refToNew and currentExposure are updated in place
    Residual optimize(uint64_t new_image_id, int pyramidLevels, SE3& refToNew, const Exposure& referenceExposure, Exposure& currentExposure);
    const std::string& lastError() const { return mError; }

private:
    cmlhip_ctx* mCtx;
    double mK[4] = {1, 1, 0, 0};
    std::string mError;
};

// x = A.ldlt().solve(b), Eigen 3.4.0 semantics (Cholesky/LDLT.h:300-396,560-600), n <= 8
bool ldltSolveSmall(const double* A, const double* b, int n, double* x);
void inverseSmall(const double* A, int n, double* Ai);

}  // namespace cml_amd
