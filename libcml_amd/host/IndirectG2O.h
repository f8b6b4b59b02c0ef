// IndirectG2O.h — host-side mirrors of CML::Optimization::G2O::IndirectCameraOptimizer
// (src/cml/optimization/g2o/IndirectCameraOptimizer.{h,cpp}) and IndirectBundleAdjustment
// (src/cml/optimization/g2o/IndirectBundleAdjustment.{h,cpp}) over the C ABI.  What the reference does on the host around
// g2o stays on the host here — collecting the matchings, the early returns, the information weights from the pyramid level,
// the write-back and the edge removal policy — and the g2o graph + optimize() calls become one device call each.
// The reference reaches everything through PFrame / PPoint / Matching; this mirror takes the same quantities flat.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/cmlhip.h"

namespace cml_amd {

struct IndirectCameraOptimizerResult {          // IndirectCameraOptimizer.h:14-19
    bool isOk = false;
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};      // camera (world -> camera)
    double covariance[6] = {0, 0, 0, 0, 0, 0};
};

class IndirectCameraOptimizer {
public:
    explicit IndirectCameraOptimizer(cmlhip_ctx* ctx) : mCtx(ctx) {}

    struct Matching {                           // Matching + the Corner / MapPoint fields optimize() reads
        bool hasMapPoint = true;                // matchings[i].getMapPoint().isNull() -> outlier, skipped (:57-62)
        double X[3] = {0, 0, 0};                // pMP->getWorldCoordinate().absolute()
        double obs[2] = {0, 0};                 // getFeaturePoint(frame).point0()
        int level = 0;                          // corner level: scaleFactor = pow(mScaleFactor, level) (types.h:1163-1165)
        double scaleFactorBase = 1.2;           // Corner::mScaleFactor (SCALEFACTOR)
        double descriptorDistance = 1;          // matchings[i].getDescriptorDistance()
    };

    bool mCheckOutliers = true;                 // IndirectCameraOptimizer.h: "checkOutliers"

    // optimize(frame, camera, matchings, outliers, computeCovariance), IndirectCameraOptimizer.cpp:4-195 (g2o Levenberg):
    // frameR/frameT = frame->getCamera(); camera (may be null) = the Optional<Camera> the rounds start from.
    IndirectCameraOptimizerResult optimize(const double frameR[9], const double frameT[3], const double* cameraR, const double* cameraT,
                                           const double K[4], const std::vector<Matching>& matchings, std::vector<bool>& outliers, bool computeCovariance);
    // optimize(frame, outliersPoints, computeCovariance), :197-382 (g2o Gauss-Newton over the frame's indirect map points):
    // outlierIndices receives the indices of the points that end as outliers (the reference returns the PPoints).
    IndirectCameraOptimizerResult optimize(const double frameR[9], const double frameT[3], const double K[4], const std::vector<Matching>& points,
                                           std::vector<int>& outlierIndices, bool computeCovariance);
    const std::string& lastError() const { return mError; }

private:
    cmlhip_ctx* mCtx;
    std::string mError;
};

class IndirectBundleAdjustment {
public:
    explicit IndirectBundleAdjustment(cmlhip_ctx* ctx) : mCtx(ctx) {}

    struct Frame { int id = 0; double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0}, K[4] = {1, 1, 0, 0}; };
    struct Apparition { int frameId = 0; double obs[2] = {0, 0}; int level = 0; double scaleFactorBase = 1.2; };
    struct Point { int id = 0; double X[3] = {0, 0, 0}; int referenceFrameId = -1; std::vector<Apparition> apparitions; };   // getIndirectApparitions()

    // parameters, names and defaults of IndirectBundleAdjustment.h:64-72
    int mNumIteration = 5, mRefineIteration = 0;
    bool mRemoveEdge = true;

    // localOptimize (IndirectBundleAdjustment.cpp:7-208) with the covisibility search already done by the caller:
    // localKeyFrames = lLocalKeyFrames, fixedCameras = lFixedCameras, points = lLocalIndirectPoints (with ALL their indirect
    // apparitions; those into frames outside the two sets are ignored, :131).  Returns what the reference returns.
    // pbStopFlag as in the reference's signature (IBA.h:27): tested before the optimisation starts (:173-178, returns false), handed to
    // the solver (setForceStopFlag, :65-67 -> cmlhip_lba_set_stop_flag) and tested again before the refinement pass (:193-198)
    bool localOptimize(const std::vector<Frame>& localKeyFrames, const std::vector<Frame>& fixedCameras, const std::vector<Point>& points, bool fixFrames,
                       bool* pbStopFlag = nullptr);
    // apply() (:238-337): the optimised keyframe cameras and point positions, and the (frameId, pointId) observations the
    // reference would remove (chi2 > 5.991 or negative depth, mRemoveEdge, not the point's reference frame, :325-334)
    struct Removal { int frameId, pointId; };
    void apply(std::vector<Frame>& localKeyFramesOut, std::vector<Point>& pointsOut, std::vector<Removal>& removals) const;
    const cmlhip_lba_result& result() const { return mResult; }
    const std::string& lastError() const { return mError; }

private:
    cmlhip_ctx* mCtx;
    std::string mError;
    bool mHaveSolution = false;
    std::vector<Frame> mLocal;
    std::vector<Point> mPoints;
    std::vector<cmlhip_lba_frame> mFrames;
    std::vector<double> mX;
    std::vector<int> mOff, mEdgeFrameId, mEdgePoint;
    std::vector<cmlhip_lba_edge> mEdges;
    std::vector<unsigned char> mBad;
    cmlhip_lba_result mResult{};
};

}  // namespace cml_amd
