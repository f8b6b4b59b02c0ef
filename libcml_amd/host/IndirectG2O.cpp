// IndirectG2O.cpp — see IndirectG2O.h.  Line references: ICO.cpp = src/cml/optimization/g2o/IndirectCameraOptimizer.cpp,
// IBA.cpp = src/cml/optimization/g2o/IndirectBundleAdjustment.cpp.
#include "IndirectG2O.h"
#include <cmath>
#include <cstring>
#include <map>

namespace cml_amd {

static void fill(IndirectCameraOptimizerResult& r, const cmlhip_pnp_result& p) {
    r.isOk = p.is_ok != 0;
    std::memcpy(r.R, p.R, sizeof r.R); std::memcpy(r.t, p.t, sizeof r.t); std::memcpy(r.covariance, p.covariance, sizeof r.covariance);
}

IndirectCameraOptimizerResult IndirectCameraOptimizer::optimize(const double frameR[9], const double frameT[3], const double* cameraR, const double* cameraT,
                                                                const double K[4], const std::vector<Matching>& matchings, std::vector<bool>& outliers,
                                                                bool computeCovariance) {
    IndirectCameraOptimizerResult result;
    const int N = (int)matchings.size();
    if ((int)outliers.size() != N) outliers.assign(N, false);                       // ICO.cpp:46-49
    std::vector<cmlhip_pnp_match> m; std::vector<unsigned char> flags; std::vector<int> index;
    m.reserve(N); flags.reserve(N); index.reserve(N);
    for (int i = 0; i < N; i++) {
        if (!matchings[i].hasMapPoint) { outliers[i] = true; continue; }            // "G2O : MapPoint is null", :57-62
        cmlhip_pnp_match e;
        std::memcpy(e.X, matchings[i].X, sizeof e.X); std::memcpy(e.obs, matchings[i].obs, sizeof e.obs);
        const double scaleFactor = std::pow(matchings[i].scaleFactorBase, matchings[i].level);      // processScaleFactorFromLevel
        e.inv_sigma2 = 1.0 / matchings[i].descriptorDistance;                       // :87-88
        e.info = 1.0 / (scaleFactor * scaleFactor);                                 // vnInfo, :89
        m.push_back(e); flags.push_back(outliers[i] ? 1 : 0); index.push_back(i);
    }
    // the device call makes the two early returns of :121-129 itself (and reports them as !is_ok)
    const double* R0 = cameraR ? cameraR : frameR; const double* t0 = cameraT ? cameraT : frameT;   // initialCamera, :132-135
    cmlhip_pnp_result p;
    const int rc = cmlhip_pnp_optimize(mCtx, R0, t0, K, (int)m.size(), m.data(), flags.data(), CMLHIP_PNP_LEVENBERG, mCheckOutliers ? 1 : 0,
                                       computeCovariance ? 1 : 0, &p);
    if (rc != CMLHIP_OK) { mError = cmlhip_last_error(mCtx); return result; }
    for (size_t k = 0; k < index.size(); k++) outliers[index[k]] = flags[k] != 0;
    fill(result, p);
    return result;
}

IndirectCameraOptimizerResult IndirectCameraOptimizer::optimize(const double frameR[9], const double frameT[3], const double K[4], const std::vector<Matching>& points,
                                                                std::vector<int>& outlierIndices, bool computeCovariance) {
    IndirectCameraOptimizerResult result;
    std::vector<cmlhip_pnp_match> m; std::vector<int> index;
    for (int i = 0; i < (int)points.size(); i++) {
        if (!points[i].hasMapPoint) continue;                                       // index without a valid value, ICO.cpp:246-250
        cmlhip_pnp_match e;
        std::memcpy(e.X, points[i].X, sizeof e.X); std::memcpy(e.obs, points[i].obs, sizeof e.obs);
        const double scaleFactor = std::pow(points[i].scaleFactorBase, points[i].level);
        e.inv_sigma2 = 1.0 / (scaleFactor * scaleFactor);                           // :280-283: the level weight on the edge as well
        e.info = e.inv_sigma2;                                                      // :285
        m.push_back(e); index.push_back(i);
    }
    std::vector<unsigned char> flags(m.size(), 0);                                  // outliers.emplace_back(false), :300
    cmlhip_pnp_result p;
    const int rc = cmlhip_pnp_optimize(mCtx, frameR, frameT, K, (int)m.size(), m.data(), flags.data(), CMLHIP_PNP_GAUSS_NEWTON, mCheckOutliers ? 1 : 0,
                                       computeCovariance ? 1 : 0, &p);
    if (rc != CMLHIP_OK) { mError = cmlhip_last_error(mCtx); return result; }
    // the reference only reports the outlier points when all four rounds ran (:352-358 sits after the loop's early returns)
    if (p.rounds == 4) for (size_t k = 0; k < index.size(); k++) if (flags[k]) outlierIndices.push_back(index[k]);
    fill(result, p);
    return result;
}

bool IndirectBundleAdjustment::localOptimize(const std::vector<Frame>& localKeyFrames, const std::vector<Frame>& fixedCameras, const std::vector<Point>& points,
                                             bool fixFrames, bool* pbStopFlag) {
    mHaveSolution = false;
    if (points.empty()) { mError = "G2O BA : No points"; return false; }                               // IBA.cpp:37-40
    if (localKeyFrames.size() <= 2) { mError = "G2O BA : Not enough frames"; return false; }           // :43-46
    if (fixedCameras.size() < 3) { mError = "G2O BA : Not enough fixed cameras"; return false; }       // :95-98
    mLocal = localKeyFrames; mPoints = points;
    mFrames.clear(); mX.clear(); mOff.assign(1, 0); mEdges.clear(); mEdgeFrameId.clear(); mEdgePoint.clear();
    std::map<int, int> slot;                                                        // frame id -> index in the frame array
    auto add = [&](const Frame& f, bool fixed) {
        cmlhip_lba_frame o;
        std::memcpy(o.R, f.R, sizeof o.R); std::memcpy(o.t, f.t, sizeof o.t); std::memcpy(o.K, f.K, sizeof o.K);
        o.fixed = fixed ? 1 : 0; o.pad = 0;
        slot[f.id] = (int)mFrames.size(); mFrames.push_back(o);
    };
    for (const Frame& f : localKeyFrames) add(f, fixFrames);                        // vSE3->setFixed(fixFrames), :74
    for (const Frame& f : fixedCameras) add(f, true);                               // :88
    int numIndirect = 0;
    for (size_t p = 0; p < points.size(); p++) {                                    // :120-165
        mX.push_back(points[p].X[0]); mX.push_back(points[p].X[1]); mX.push_back(points[p].X[2]);
        for (const Apparition& a : points[p].apparitions) {
            const auto it = slot.find(a.frameId);
            if (it == slot.end()) continue;                                         // neither local nor fixed, :131
            cmlhip_lba_edge e;
            e.frame = it->second; e.pad = 0; e.obs[0] = a.obs[0]; e.obs[1] = a.obs[1];
            const double scaleFactor = std::pow(a.scaleFactorBase, a.level);
            e.inv_sigma2 = 1.0 / (scaleFactor * scaleFactor);                       // :141-143
            mEdges.push_back(e); mEdgeFrameId.push_back(a.frameId); mEdgePoint.push_back((int)p);
            numIndirect++;
        }
        mOff.push_back((int)mEdges.size());
    }
    if (numIndirect < 10) { mError = "G2O Ba : Not enough indirect points"; return false; }            // :167-170
    if (pbStopFlag != nullptr && *pbStopFlag) { mError = "G2O Ba: Stop flag is set to true"; return false; }   // :173-178
    mBad.assign(mEdges.size(), 0);
    static_assert(sizeof(bool) == 1, "pbStopFlag is read as a byte");
    cmlhip_lba_set_stop_flag(mCtx, reinterpret_cast<const unsigned char*>(pbStopFlag));                 // setForceStopFlag, :65-67; also the test before the refinement pass, :193-198
    struct Reset { cmlhip_ctx* c; ~Reset() { cmlhip_lba_set_stop_flag(c, nullptr); } } reset{mCtx};
    const int rc = cmlhip_lba_optimize(mCtx, (int)mFrames.size(), mFrames.data(), (int)points.size(), mX.data(), mOff.data(), mEdges.data(),
                                       fixFrames ? 1 : 0, mNumIteration, mRefineIteration, mBad.data(), &mResult);
    if (rc != CMLHIP_OK) { mError = cmlhip_last_error(mCtx); return false; }
    mHaveSolution = true;
    return true;
}

void IndirectBundleAdjustment::apply(std::vector<Frame>& localKeyFramesOut, std::vector<Point>& pointsOut, std::vector<Removal>& removals) const {
    localKeyFramesOut.clear(); pointsOut.clear(); removals.clear();
    if (!mHaveSolution) return;                                                     // mOptimizer == nullptr, IBA.cpp:241-243
    localKeyFramesOut = mLocal;
    for (size_t f = 0; f < mLocal.size(); f++) {                                    // pKF->setCamera, :301-305
        std::memcpy(localKeyFramesOut[f].R, mFrames[f].R, sizeof mFrames[f].R);
        std::memcpy(localKeyFramesOut[f].t, mFrames[f].t, sizeof mFrames[f].t);
    }
    pointsOut = mPoints;
    for (size_t p = 0; p < mPoints.size(); p++) for (int k = 0; k < 3; k++) pointsOut[p].X[k] = mX[3 * p + k];     // setWorldCoordinate, :309-318
    for (size_t e = 0; e < mEdges.size(); e++)                                      // :322-334
        if (mBad[e] && mRemoveEdge && mPoints[mEdgePoint[e]].referenceFrameId != mEdgeFrameId[e])
            removals.push_back(Removal{mEdgeFrameId[e], mPoints[mEdgePoint[e]].id});
}

}  // namespace cml_amd
