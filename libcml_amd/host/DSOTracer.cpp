// DSOTracer.cpp — host mirror of CML::Optimization::DSOTracer over the C ABI (TRC.cpp = src/cml/optimization/dso/DSOTracer.cpp).
#include "DSOTracer.h"
#include "HostLap.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace cml_amd {

DSOTracer::DSOTracer(cmlhip_ctx* ctx) : mCtx(ctx) {
    prm.max_pix_search = (double)0.027f; prm.max_slack_interval = (double)1.5f; prm.trace_step_size = (double)1.0f;
    prm.min_improvement_factor = (double)2.0f; prm.min_trace_test_radius = (double)2.0f; prm.extra_slack_on_th = (double)1.2f;
    prm.huber_th = (double)9.0f; prm.outlier_th_sum_component = (double)(50.0f * 50.0f); prm.min_idepth_h_act = (double)100.0f;
    prm.gn_its_on_activation = 3; prm.pad = 0;
}

int DSOTracer::addImmaturePoint(float x, float y, int host_frame_id, const float gray[8], const float dpatch[24], const double gradH[4], float type) {
    ImmaturePoint P;
    std::memset(&P.d, 0, sizeof P.d);
    P.d.x = x; P.d.y = y; P.d.host = -1; P.frame_id = host_frame_id;
    P.d.last_status = CMLHIP_IPS_UNINITIALIZED;                                 // DSOTracer.h:21-28
    P.d.idepth_min = 1.0 / 1000.0; P.d.idepth_max = NAN;
    P.d.last_uv[0] = P.d.last_uv[1] = -1; P.d.last_pixel_interval = -1;
    P.d.quality = 10000;
    std::memcpy(P.d.gradH, gradH, sizeof P.d.gradH);
    std::memcpy(P.d.gray, gray, sizeof P.d.gray);
    std::memcpy(P.d.dpatch, dpatch, sizeof P.d.dpatch);
    P.d.energy_th = 8 * mSettingOutlierTH;                                      // TRC.cpp:527
    P.my_type = type;
    if (!std::isfinite(P.d.energy_th)) return -1;                               // :531-534
    pullResident();
    mResDirty = true; mResSlotsValid = false;
    mPoints.push_back(P);
    return (int)mPoints.size() - 1;
}

void DSOTracer::compact() {
    HostLap lap("tracer compact");
    pullResident();
    lap("state pulled");
    mResDirty = true; mResSlotsValid = false;
    std::vector<ImmaturePoint> keep;
    std::vector<int> moved(mPoints.size(), -1);                                 // old index -> new index
    for (size_t i = 0; i < mPoints.size(); i++) if (mPoints[i].alive && !mPoints[i].activated) { moved[i] = (int)keep.size(); keep.push_back(mPoints[i]); }
    for (int& w : mResWho) if (w >= 0) w = moved[w];                            // (the device's slots still hold the points that left: dropped at the next edit)
    mPoints.swap(keep);
    lap("list rebuilt");
}

static int indexOf(const std::vector<int>& ids, int id) {
    for (size_t i = 0; i < ids.size(); i++) if (ids[i] == id) return (int)i;
    return -1;
}

bool DSOTracer::pullResident() {
    if (!mHostStale) return true;
    mHostStale = false;
    if (mResWho.empty()) return true;
    // the seven fields trace() writes (56 bytes per point), not the 232-byte records
    std::vector<cmlhip_immature_state>& st = mStateBuf;                         // (kept across calls: 56 bytes x ~2 500 points zero-filled per call otherwise)
    if (st.size() < mResWho.size()) st.resize(mResWho.size());
    const int rc = cmlhip_tracer_get_state(mCtx, (int)mResWho.size(), st.data());
    if (rc) { mError = std::string("cmlhip_tracer_get_state: ") + cmlhip_last_error(mCtx); return false; }
    for (size_t k = 0; k < mResWho.size(); k++) {
        if (mResWho[k] < 0) continue;
        cmlhip_immature_point& d = mPoints[mResWho[k]].d;
        d.idepth_min = st[k].idepth_min; d.idepth_max = st[k].idepth_max; d.quality = st[k].quality;
        d.last_uv[0] = st[k].last_uv[0]; d.last_uv[1] = st[k].last_uv[1]; d.last_pixel_interval = st[k].last_pixel_interval; d.last_status = st[k].last_status;
    }
    return true;
}

// The device's set follows this list by EDITS (cmlhip_tracer_edit_points): the points that stay keep their records where they are — moved to the front
// in their old order, with their host index in the current frame list — and only the points added since the last call travel (makeNewTraces' ~500 per
// keyframe instead of the whole set).
bool DSOTracer::syncResident(const std::vector<int>& frame_ids) {
    if (!mResDirty && frame_ids == mResFrameIds) return true;
    HostLap lap("syncResident");
    if (!pullResident()) return false;
    lap("state pulled");
    std::vector<int> keep, hosts, who;
    std::vector<cmlhip_immature_point> fresh;
    std::vector<int> freshWho;
    // mResWho lists the points in slot order; compact() has kept it current (indices into mPoints, -1 for a point that left the list)
    for (size_t k = 0; k < mResWho.size(); k++) {
        const int i = mResWho[k];
        if (i < 0) continue;
        ImmaturePoint& P = mPoints[i];
        P.res_slot = -1;
        if (!P.alive || P.activated) continue;
        const int h = indexOf(frame_ids, P.frame_id);
        if (h < 0) { P.alive = false; continue; }                               // reference frame left the group, TRC.cpp:20-26
        P.d.host = h;
        keep.push_back((int)k); hosts.push_back(h); who.push_back(i);
    }
    for (int i = 0; i < (int)mPoints.size(); i++) mPoints[i].res_slot = -1;
    for (size_t k = 0; k < who.size(); k++) mPoints[who[k]].res_slot = (int)k;
    for (int i = 0; i < (int)mPoints.size(); i++) {
        ImmaturePoint& P = mPoints[i];
        if (!P.alive || P.activated || P.res_slot >= 0) continue;
        if (P.was_resident) continue;                                           // (left the set above: host gone)
        const int h = indexOf(frame_ids, P.frame_id);
        if (h < 0) { P.alive = false; continue; }
        P.d.host = h;
        P.res_slot = (int)(who.size() + fresh.size());
        fresh.push_back(P.d); freshWho.push_back(i);
    }
    lap("edit lists");
    const int rc = cmlhip_tracer_edit_points(mCtx, (int)keep.size(), keep.data(), hosts.data(), (int)fresh.size(), fresh.data());
    if (rc) { mError = std::string("cmlhip_tracer_edit_points: ") + cmlhip_last_error(mCtx); return false; }
    who.insert(who.end(), freshWho.begin(), freshWho.end());
    mResWho.swap(who);
    for (int i : mResWho) mPoints[i].was_resident = true;
    mResFrameIds = frame_ids;
    mResDirty = false; mResSlotsValid = true;
    lap("device edit");
    return true;
}

bool DSOTracer::traceNewCoarse(uint64_t traced_image_id, int traced_frame_id, const std::vector<int>& frame_ids,
                               const std::vector<cmlhip_trace_pair>& pairs, int counts[6]) {
    for (int c = 0; c < 6; c++) counts[c] = 0;
    if (!syncResident(frame_ids)) return false;
    if (mResWho.empty() || pairs.empty()) return true;
    const int skip = indexOf(frame_ids, traced_frame_id);                       // trace() returns the old status for the traced frame's own points, :597-599
    const int rc = cmlhip_tracer_trace_resident(mCtx, traced_image_id, &prm, (int)pairs.size(), pairs.data(), skip >= 0 ? skip : -2, counts);
    if (rc) { mError = std::string("cmlhip_tracer_trace_resident: ") + cmlhip_last_error(mCtx); return false; }
    mHostStale = true;
    return true;
}

bool DSOTracer::traceNewCoarseTrackedAsync(uint64_t traced_image_id, int traced_frame_id, const std::vector<int>& frame_ids,
                                           const std::vector<cmlhip_frame_pose>& hosts, const cmlhip_frame_pose& reference, const double K[4]) {
    if (hosts.size() != frame_ids.size() || hosts.empty()) { mError = "traceNewCoarseTrackedAsync: one pose per window frame"; return false; }
    if (!syncResident(frame_ids)) return false;
    const int skip = indexOf(frame_ids, traced_frame_id);
    const int rc = cmlhip_tracer_trace_resident_tracked_async(mCtx, traced_image_id, &prm, (int)hosts.size(), hosts.data(), &reference, K, skip >= 0 ? skip : -2);
    if (rc) { mError = std::string("cmlhip_tracer_trace_resident_tracked_async: ") + cmlhip_last_error(mCtx); return false; }
    mTrackedPending = true;
    return true;
}

bool DSOTracer::prepareTracked(const std::vector<cmlhip_frame_pose>& hosts, const cmlhip_frame_pose& reference, const double K[4]) {
    if (hosts.empty()) { mError = "prepareTracked: no window"; return false; }
    const int rc = cmlhip_tracer_tracked_prepare(mCtx, (int)hosts.size(), hosts.data(), &reference, K);
    if (rc) { mError = std::string("cmlhip_tracer_tracked_prepare: ") + cmlhip_last_error(mCtx); return false; }
    return true;
}

bool DSOTracer::finishTracked(bool keep, int counts[6], std::vector<cmlhip_trace_pair>* pairs_out) {
    if (!mTrackedPending) { mError = "finishTracked: nothing in flight"; return false; }
    mTrackedPending = false;
    if (pairs_out) pairs_out->resize(mResFrameIds.size());
    const int rc = cmlhip_tracer_trace_resident_finish(mCtx, keep ? 1 : 0, counts, pairs_out ? pairs_out->data() : nullptr);
    if (rc) { mError = std::string("cmlhip_tracer_trace_resident_finish: ") + cmlhip_last_error(mCtx); return false; }
    if (keep) mHostStale = true;
    return true;
}

bool DSOTracer::activatePoints(const std::vector<int>& frame_ids, const std::vector<uint64_t>& image_ids, const double K[4], int w, int h,
                               const std::vector<cmlhip_activation_pair>& pairs, std::vector<int>& activated, const SpacingPolicy& spacing) {
    const int N = (int)frame_ids.size(), last = N - 1;
    HostLap lap("activatePoints");
    if (!pullResident()) return false;
    lap("state pulled");
    mResDirty = true;                                                           // (points leave the set here: activated, dropped, out of the image)
    activated.clear();
    numSkippedBecauseStatus = numSkippedBecausePixelInterval = numSkippedBecauseQuality = numSkippedBecauseDepth = 0;
    numDeletedBecauseOutlier = numDeletedBecauseOOB = numMapped = numNonMapped = numDropped = 0;
    std::vector<cmlhip_immature_point> batch;
    std::vector<int> who;
    for (int i = 0; i < (int)mPoints.size(); i++) {
        ImmaturePoint& P = mPoints[i];
        if (!P.alive || P.activated) continue;
        const int hst = indexOf(frame_ids, P.frame_id);
        if (hst == last) continue;                                              // TRC.cpp:120-122
        if (hst < 0) { P.alive = false; continue; }                             // :124-127
        if (!std::isfinite(P.d.idepth_max) || P.d.last_status == CMLHIP_IPS_OUTLIER) { P.alive = false; numDeletedBecauseOutlier++; continue; }   // :129-134
        const bool okStatus = P.d.last_status == CMLHIP_IPS_GOOD || P.d.last_status == CMLHIP_IPS_SKIPPED ||
                              P.d.last_status == CMLHIP_IPS_BADCONDITION || P.d.last_status == CMLHIP_IPS_OOB;
        const bool okInterval = P.d.last_pixel_interval < 8;
        const bool okQuality = P.d.quality > mSettingsMinTraceQuality;
        const bool okDepth = (P.d.idepth_max + P.d.idepth_min) > 0;
        if (!okStatus) numSkippedBecauseStatus++;
        if (!okInterval) numSkippedBecausePixelInterval++;
        if (!okQuality) numSkippedBecauseQuality++;
        if (!okDepth) numSkippedBecauseDepth++;
        if (!(okStatus && okInterval && okQuality && okDepth)) {
            if (P.d.last_status == CMLHIP_IPS_OOB) { P.alive = false; numDeletedBecauseOOB++; }    // :170-176
            continue;
        }
        // projection into the last frame at the centre of the interval, TRC.cpp:181-183
        const double idepth = (P.d.idepth_min + P.d.idepth_max) / 2.0;
        const cmlhip_activation_pair& hl = pairs[(size_t)hst * N + last];
        const double ux = ((double)P.d.x - K[2]) * (1.0 / K[0]), uy = ((double)P.d.y - K[3]) * (1.0 / K[1]);
        const double p0 = hl.R[0] * ux + hl.R[1] * uy + hl.R[2] + hl.t[0] * idepth, p1 = hl.R[3] * ux + hl.R[4] * uy + hl.R[5] + hl.t[1] * idepth,
                     p2 = hl.R[6] * ux + hl.R[7] * uy + hl.R[8] + hl.t[2] * idepth;
        const double u = (p0 / p2) * K[0] + K[2], v = (p1 / p2) * K[1] + K[3];
        if (!(u >= 0 && v >= 0 && u < w && v < h)) { P.alive = false; continue; }                 // :185,193-197
        if (spacing && !spacing(u, v, P.my_type)) continue;                                        // DistanceMap test, :186-191
        P.d.host = hst;
        who.push_back(i);                                                       // (the 232-byte records are copied only if they have to travel, below)
    }
    lap("candidates selected");
    if (who.empty()) return true;
    std::vector<int> result(who.size()), states(who.size() * (size_t)N);
    std::vector<float> idp(who.size());
    // the candidates are points of the device-resident set: named by their slots when the set is current for this frame list (the list it was edited
    // with, with frames appended behind it — the new keyframe), their 232-byte records do not travel again
    bool resident = mResSlotsValid && mResFrameIds.size() <= frame_ids.size();
    for (size_t f = 0; resident && f < mResFrameIds.size(); f++) resident = mResFrameIds[f] == frame_ids[f];
    std::vector<int> slots(who.size());
    for (size_t k = 0; resident && k < who.size(); k++) { slots[k] = mPoints[who[k]].res_slot; resident = slots[k] >= 0; }
    int rc;
    if (resident) {
        rc = cmlhip_optimize_immature_points_resident(mCtx, N, image_ids.data(), K, pairs.data(), &prm, 1, (int)slots.size(), slots.data(), result.data(), idp.data(), states.data());
        if (rc) { mError = std::string("cmlhip_optimize_immature_points_resident: ") + cmlhip_last_error(mCtx); return false; }
    } else {
        batch.reserve(who.size());
        for (int i : who) batch.push_back(mPoints[i].d);
        rc = cmlhip_optimize_immature_points(mCtx, N, image_ids.data(), K, pairs.data(), &prm, 1, (int)batch.size(), batch.data(), result.data(), idp.data(), states.data());
        if (rc) { mError = std::string("cmlhip_optimize_immature_points: ") + cmlhip_last_error(mCtx); return false; }
    }
    lap("device optimisation");
    for (size_t k = 0; k < who.size(); k++) {                                   // TRC.cpp:216-247
        ImmaturePoint& P = mPoints[who[k]];
        if (result[k] == 1) {
            P.activated = true; P.idepth = idp[k];
            P.n_res_state = std::min(N, (int)CMLHIP_MAX_FRAMES);
            for (int f = 0; f < P.n_res_state; f++) P.res_state[f] = (signed char)states[k * (size_t)N + f];
            activated.push_back(who[k]);
            numMapped++;
        } else if (result[k] == -1 || P.d.last_status == CMLHIP_IPS_OOB) { P.alive = false; numDropped++; }
        else numNonMapped++;
    }
    lap("results applied");
    return true;
}

}  // namespace cml_amd
