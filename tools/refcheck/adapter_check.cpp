// adapter_check.cpp — COMPILE-ONLY check of the drop-in boundary from the reference's side (build container only; nothing here is
// shipped or linked into the product, and no reference source is copied: this translation unit only CALLS the reference's headers).
//
// It contains the adapter a libCML maintainer writes inside DSOBundleAdjustment::run / DSOTracker: the flat
// records of include/cmlhip.h filled from the reference's own types, member by member.  If the ABI forgot a field, misnamed one or
// assumed a type the reference does not have, this file does not compile.  tools/refcheck/check.sh runs the compiler in
// -fsyntax-only mode against /root/reference (with cml/config.h produced from the reference's config.h.in in a temporary
// directory, the one substitution its cmake performs).
#include <cml/optimization/dso/DSOContext.h>
#include <cml/optimization/dso/DSOFrame.h>
#include <cml/optimization/dso/DSOPoint.h>
#include <cml/optimization/dso/DSOResidual.h>
#include <cml/optimization/dso/DSOTracker.h>

#include "../../include/cmlhip.h"

#include <vector>

namespace CML { namespace Optimization {

// cmlhip_ba_frame <- DSOFrame (DSOFrame.h:17-246) + the frame's level-0 gradient image id (Array2D::getId)
inline cmlhip_ba_frame make_frame(PFrame frame, DSOFrame& data, float scaleLightB) {
    cmlhip_ba_frame f;
    f.image_id = (uint64_t)frame->getCaptureFrame().getDerivativeImage(0).getId();
    f.frame_energy_th = (float)data.frameEnergyTH;                     // DSOFrame.h:35
    f.b0 = data.getB0(scaleLightB);                                    // DSOFrame.h:197-199
    return f;
}

// cmlhip_ba_point <- MapPoint + DSOPoint (DSOPoint.h:44-167); host = DSOFrame::id of the reference frame
inline cmlhip_ba_point make_point(PPoint point, DSOPoint& data, int hostId) {
    cmlhip_ba_point p;
    const Corner corner = point->getReferenceCorner();                  // MapObject.h:108
    p.x = (float)corner.x(); p.y = (float)corner.y();
    p.idepth = point->getReferenceInverseDepth();                       // MapObject.h:110
    p.idepth_zero = data.idepth_zero;                                   // DSOPoint.h:73
    p.prior = data.priorF;                                              // DSOPoint.h:69
    for (int i = 0; i < CMLHIP_PATTERN; i++) {
        p.colors[i] = (float)data.colors[i];                            // DSOPoint.h:54
        p.weights[i] = (float)data.weights[i];                          // DSOPoint.h:65 (BA.cpp:405-411)
    }
    p.host = hostId;
    return p;
}

// cmlhip_ba_residual <- DSOResidual (DSOResidual.h:72-156)
inline cmlhip_ba_residual make_residual(DSOResidual* r, int pointIndex, int targetId) {
    cmlhip_ba_residual o;
    o.point = pointIndex;
    o.target = targetId;
    o.state = (int)r->getState();                                       // DSORES_IN = 0, OOB = 1, OUTLIER = 2 == CMLHIP_RES_*
    o.is_linearized = r->isLinearized ? 1 : 0;
    return o;
}
static_assert((int)DSORES_IN == CMLHIP_RES_IN && (int)DSORES_OOB == CMLHIP_RES_OOB && (int)DSORES_OUTLIER == CMLHIP_RES_OUTLIER,
              "residual state codes of include/cmlhip.h must be the reference's DSOResidualState values");

// cmlhip_ba_pair <- DSOFramePrecomputed (DSOFrame.h:248-291)
inline cmlhip_ba_pair make_pair(const DSOFramePrecomputed& pre) {
    cmlhip_ba_pair o;
    const Matrix33& R = pre.trialRefToTarget.getRotationMatrix();       // BA.cpp:98-99
    const Vector3& t = pre.trialRefToTarget.getTranslation();
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) { o.R[3 * i + j] = R(i, j); o.R0[3 * i + j] = pre.PRE_RTll_0(i, j); }
        o.t[i] = t[i]; o.t0[i] = pre.PRE_tTll_0[i];
    }
    const Vector2 ab = pre.exposureTransition.getParameters();          // Exposure.h:34-36
    o.aff_a = ab[0]; o.aff_b = ab[1];
    return o;
}

// write-back after cmlhip_ba_finish_keyframe: what BA::run stores per residual and per point (BA.cpp:1571-1640, 1889-1901)
inline void write_back(DSOResidual* r, int state, double energy, double newEnergyWithOutlier, bool good) {
    r->setState((DSOResidualState)state);
    r->state_energy = energy;
    r->state_NewEnergyWithOutlier = newEnergyWithOutlier;
    r->isActiveAndIsGoodNEW = good;
}
inline void write_back(PPoint point, DSOPoint& data, double idepth, float HdiF) {
    point->setReferenceInverseDepth(idepth);
    data.HdiF = HdiF;
    data.setInverseDepthHessian(HdiF != 0 ? 1.0f / HdiF : 0.0f);
}

// cmlhip_ba_frame_state (device-resident loop) <- DSOFrame: evaluation point, state vectors, prior zero, exposure time
inline cmlhip_ba_frame_state make_frame_state(const DSOFrame& f, bool fixPose) {
    cmlhip_ba_frame_state s;
    const Sophus::SE3<scalar_t> ev = f.get_worldToCam_evalPT();
    const auto q = ev.unit_quaternion();
    s.eval_q[0] = q.w(); s.eval_q[1] = q.x(); s.eval_q[2] = q.y(); s.eval_q[3] = q.z();
    for (int i = 0; i < 3; i++) s.eval_t[i] = ev.translation()[i];
    const Vector<10> st = f.get_state(), sz = f.get_state_zero();
    for (int i = 0; i < 10; i++) { s.state[i] = st[i]; s.state_zero[i] = sz[i]; s.prior_zero[i] = f.prior_zero[i]; }
    s.ab_exposure = f.ab_exposure;
    s.fix_pose = fixPose ? 1 : 0;
    s.pad = 0;
    return s;
}

// tracker: cmlhip_tracker_params <- DSOTracker's parameters are private Parameter members (DSOTracker.h:473-520); the adapter lives
// INSIDE the class, so it is checked as a member-style template over anything exposing them
template <typename Tracker>
inline cmlhip_tracker_params make_tracker_params(Tracker& t, double cutoffRepeat) {
    cmlhip_tracker_params p;
    p.huber = (float)t.mHuberThreshold.f();
    p.cutoff_base = (float)t.mCutoffThreshold.f();
    p.cutoff = (float)(t.mCutoffThreshold.f() * cutoffRepeat);
    p.scale_rot = (float)t.mScaleRotation.f(); p.scale_trans = (float)t.mScaleTranslation.f();
    p.scale_a = (float)t.mScaleLightA.f(); p.scale_b = (float)t.mScaleLightB.f();
    return p;
}

// tracker result -> DSOTracker::Residual (DSOTracker.h:200-232)
inline void fill_residual(DSOTracker::Residual& res, int level, const cmlhip_tracker_result& r) {
    res.E[level] = r.E; res.numTermsInE[level] = r.numTermsInE; res.numSaturated[level] = r.numSaturated; res.numRobust[level] = r.numRobust;
    res.flowVector = Vector3(r.flow[0], r.flow[1], r.flow[2]);
}

}}  // namespace CML::Optimization

// every adapter is instantiated so that -fsyntax-only checks its body
void cmlhip_refcheck_instantiate(CML::PFrame frame, CML::PPoint point, CML::Optimization::DSOFrame& fd, CML::Optimization::DSOPoint& pd,
                                 CML::Optimization::DSOResidual* r, const CML::Optimization::DSOFramePrecomputed& pre,
                                 CML::Optimization::DSOTracker::Residual& res, const cmlhip_tracker_result& tr) {
    using namespace CML::Optimization;
    std::vector<cmlhip_ba_frame> frames{make_frame(frame, fd, 1000.0f)};
    std::vector<cmlhip_ba_point> points{make_point(point, pd, fd.id)};
    std::vector<cmlhip_ba_residual> residuals{make_residual(r, 0, fd.id)};
    std::vector<cmlhip_ba_pair> pairs{make_pair(pre)};
    std::vector<cmlhip_ba_frame_state> states{make_frame_state(fd, false)};
    write_back(r, CMLHIP_RES_IN, 0.0, 0.0, true);
    write_back(point, pd, 0.1, 1.0f);
    fill_residual(res, 0, tr);
    (void)frames; (void)points; (void)residuals; (void)pairs; (void)states;
}
