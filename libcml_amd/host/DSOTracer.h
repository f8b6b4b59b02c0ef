// DSOTracer.h — host-side mirror of CML::Optimization::DSOTracer (src/cml/optimization/dso/DSOTracer.h:35-206,
// DSOTracer.cpp:13-278) over the C ABI: the immature-point bookkeeping and the activation policy stay on the host, the
// per-point epipolar search (trace) and the activation Gauss-Newton (optimizeImmaturePoint) are device calls.
// The reference reaches points/frames through Map/MapPoint/PrivateData; this mirror takes the same quantities flat.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>
#include "../../include/cmlhip.h"
#include "se3.h"

namespace cml_amd {

class DSOTracer {
public:
    explicit DSOTracer(cmlhip_ctx* ctx);

    // parameters, names and defaults of DSOTracer.h:188-206 (float literals widened, like Parameter stores them)
    cmlhip_tracer_params prm;
    double mSettingOutlierTH = (double)(12.0f * 12.0f), mSettingsMinTraceQuality = (double)3.0f;

    struct ImmaturePoint {                      // DSOTracerPointPrivate + the MapPoint fields, see cmlhip_immature_point
        cmlhip_immature_point d;
        int frame_id = -1;                      // stable id of the host keyframe (index into the caller's frame list at call time is d.host)
        float my_type = 1;
        bool alive = true, activated = false;
        float idepth = 0;                       // set on activation
        signed char res_state[CMLHIP_MAX_FRAMES] = {};   // per frame of the activation window (-1 host), n_res_state entries — inline: no allocation per activated point
        int n_res_state = 0;
        int res_slot = -1;                      // the point's slot in the device-resident set (-1: not there)
        bool was_resident = false;              // has been in the device's set (a point is added to it once)
    };

    // makeNewTraces' per-point record (DSOTracer.cpp:496-541): gradH/patches are computed by the caller from the host image
    int addImmaturePoint(float x, float y, int host_frame_id, const float gray[8], const float dpatch[24], const double gradH[4], float type = 1);
    // traceNewCoarse (DSOTracer.cpp:13-57): frame_ids[h] = stable id of window frame h, pairs[h] = host h -> traced frame.
    // Points hosted by a frame that is no longer in the list are removed (:20-26).  counts = {good, oob, outlier, skipped, badcondition, uninitialized}
    bool traceNewCoarse(uint64_t traced_image_id, int traced_frame_id, const std::vector<int>& frame_ids, const std::vector<cmlhip_trace_pair>& pairs,
                        int counts[6]);
    // activatePoints (DSOTracer.cpp:59-278) without the map bookkeeping: candidate tests (:128-178), optional spacing policy
    // (the reference's DistanceMap, :180-196) as a callback on the point's projection in the last frame, batch
    // optimizeImmaturePoint on the device, result handling (:216-247).  Returns the indices of the activated points.
    using SpacingPolicy = std::function<bool(double u, double v, float type)>;
    bool activatePoints(const std::vector<int>& frame_ids, const std::vector<uint64_t>& image_ids, const double K[4], int w, int h,
                        const std::vector<cmlhip_activation_pair>& pairs, std::vector<int>& activated, const SpacingPolicy& spacing = nullptr);

    // traceNewCoarse behind a tracker batch that is still in flight (cmlhip_tracer_trace_resident_tracked_async): the pairs are formed on the device from
    // the batch's first result, `hosts` (world -> camera pose and exposure of every window frame, in frame_ids order) and `reference` (the keyframe the
    // hypotheses are relative to).  finishTracked(keep): keep = the caller's replay of the selection adopted that first try — counts / the pairs that were
    // used come back; otherwise every traced point is restored and the caller traces again (traceNewCoarse) with the pose it did select.
    bool traceNewCoarseTrackedAsync(uint64_t traced_image_id, int traced_frame_id, const std::vector<int>& frame_ids, const std::vector<cmlhip_frame_pose>& hosts,
                                    const cmlhip_frame_pose& reference, const double K[4]);
    // optional, ahead of DSOTracker::trackWithMotionModelBatchedEnqueue: the window traceNewCoarseTrackedAsync will pass (cmlhip_tracer_tracked_prepare)
    bool prepareTracked(const std::vector<cmlhip_frame_pose>& hosts, const cmlhip_frame_pose& reference, const double K[4]);
    bool finishTracked(bool keep, int counts[6], std::vector<cmlhip_trace_pair>* pairs_out);

    // bring the device's set up to date NOW (the keyframe's work: makeNewTraces has added its points, marginalizeFrames has removed frames) instead of in
    // front of the next frame's trace
    bool prepareResident(const std::vector<int>& frame_ids) { return syncResident(frame_ids); }
    void compact();                             // forget the points that were activated or removed (getMap().removeMapPoint in the reference): indices change
    // The immature set lives ON THE DEVICE between keyframes (cmlhip_tracer_set_points / _trace_resident: a traced frame moves 24 bytes, not the set);
    // this object's copy of the fields trace() writes is refreshed when somebody looks (points(), activatePoints, compact).
    std::vector<ImmaturePoint>& points() { pullResident(); return mPoints; }
    size_t size() const { return mPoints.size(); }
    const std::vector<ImmaturePoint>& peek() const { return mPoints; }     // the host-owned fields (frame_id, alive, activated, idepth, static patches) are always current
    const std::string& lastError() const { return mError; }
    int numSkippedBecauseStatus = 0, numSkippedBecausePixelInterval = 0, numSkippedBecauseQuality = 0, numSkippedBecauseDepth = 0,
        numDeletedBecauseOutlier = 0, numDeletedBecauseOOB = 0, numMapped = 0, numNonMapped = 0, numDropped = 0;

private:
    bool syncResident(const std::vector<int>& frame_ids);    // the device's set = the live, not yet activated points, hosts indexed against frame_ids (re-sent only when something changed)
    bool pullResident();
    cmlhip_ctx* mCtx;
    std::vector<ImmaturePoint> mPoints;
    std::vector<cmlhip_immature_state> mStateBuf;     // pullResident's readback buffer
    std::vector<int> mResWho, mResFrameIds;     // device slot -> index into mPoints; the frame list the device's host indices refer to
    bool mResDirty = true, mHostStale = false, mTrackedPending = false;
    bool mResSlotsValid = false;                // every live point's res_slot names its record on the device (no point added / list compacted since the last edit)
    std::string mError;
};

}  // namespace cml_amd
