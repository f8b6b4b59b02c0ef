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
        std::vector<int> res_state;             // per frame of the activation window (-1 host)
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

    void compact();                             // forget the points that were activated or removed (getMap().removeMapPoint in the reference): indices change
    std::vector<ImmaturePoint>& points() { return mPoints; }
    const std::string& lastError() const { return mError; }
    int numSkippedBecauseStatus = 0, numSkippedBecausePixelInterval = 0, numSkippedBecauseQuality = 0, numSkippedBecauseDepth = 0,
        numDeletedBecauseOutlier = 0, numDeletedBecauseOOB = 0, numMapped = 0, numNonMapped = 0, numDropped = 0;

private:
    cmlhip_ctx* mCtx;
    std::vector<ImmaturePoint> mPoints;
    std::string mError;
};

}  // namespace cml_amd
