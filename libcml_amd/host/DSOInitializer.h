// DSOInitializer.h — host-side mirror of CML::Optimization::DSOInitializer (src/cml/optimization/dso/DSOInitializer.{h,cpp})
// over the C ABI.  The per-evaluation work, calcResAndGS (:451-750), is the device call cmlhip_initializer_calc_res_and_gs;
// everything around it — the point records of setFirst (:7-107), the neighbour tables of makeNN (:919-984), the per-level
// Levenberg loop of tryInitialize (:115-341), doStep / applyStep / optReg / calcEC / resetPoints / propagateUp / propagateDown
// (:752-917) and the snapping / success logic — is O(npts) scalar work on a handful of frames per sequence and stays on the
// host, statement for statement.  The pixel selection (Features::PixelSelector) stays with the caller: setFirst takes the
// selected pixels of every level.  The reference reaches images and poses through PFrame; this mirror takes them flat.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/cmlhip.h"
#include "se3.h"

namespace cml_amd {

class DSOInitializer {
public:
    explicit DSOInitializer(cmlhip_ctx* ctx) : mCtx(ctx) {}

    // parameters, names and defaults of DSOInitializer.h:141-170
    float mRegulalizationWeight = 0.45f, mHuberThreshold = 9.0f, mScaleRotation = 1.0f, mScaleTranslation = 0.5f, mScaleLightA = 10.0f,
          mScaleLightB = 1000.0f, mSettingOutlierTH = 12.0f * 12.0f, mNNWeight = 0.25f;
    int mSettingsDesiredPointDensity = 1800;

    struct Point {                               // DSOInitializerPoint, DSOInitializer.h:11-58
        cmlhip_init_point d;                     // the fields calcResAndGS reads and writes
        float px, py;                            // p
        float idepth = 1, initialiR = 1, iRSumNum = 0, lastHessian = 0;
        int parent = -1; float parentDist = -1;
        int neighbours[10]; float neighboursDist[10];
        float jb[10];                            // mJbBuffer[i] (the accepted row; d.jb is mJbBuffer_new[i])
    };
    struct LevelInput {                          // one pyramid level of the reference frame
        int w = 0, h = 0; double K[4] = {1, 1, 0, 0};
        const float* gray = nullptr;             // reference->getCaptureFrame().getGrayImage(lvl), row-major w x h
        std::vector<int> px, py;                 // the pixels PixelSelector kept at this level (x, y integer positions, :44-50)
    };

    // setFirst (:7-107): builds the point records (p = pixel + 0.1, pattern positions, reference colours, outlierTH) and makeNN.
    // referenceCamera = reference->getCamera(), referenceExposure = getExposure().getExposureFromCamera()
    bool setFirst(const std::vector<LevelInput>& levels, const SE3& referenceCamera, double referenceExposure);
    // tryInitialize (:115-341) for one new frame whose pyramid is in the device cache under `imageId`:
    // frameCamera / frameExposure = the frame's camera and exposure on entry.  Returns -1 / 0 / 1 like the reference.
    int tryInitialize(uint64_t imageId, const SE3& frameCamera, double frameExposure);

    const SE3& currentCamera() const { return mCurrentCamera; }
    bool snapped() const { return mSnapped; }
    int frameID() const { return mFrameID; }
    const std::vector<Point>& points(int lvl) const { return mPoints[lvl]; }
    int numLevels() const { return (int)mPoints.size(); }
    // onInitializationSuccess (:343-440), the numeric part: rescale factor 0.5 / median(iR of the good level-0 points); the
    // rescaled camera; inverse depths iR * rescaleFactor of the (at most pointDensity) evenly strided good points
    float rescaleFactor() const { return mRescaleFactor; }
    void initializedPoints(std::vector<int>& index, std::vector<float>& idepth) const;
    const std::string& lastError() const { return mError; }
    int numCalcCalls = 0, numAccepted = 0, numRejected = 0;

private:
    void makeNN();
    void resetPoints(int lvl);
    void doStep(int lvl, float lambda, const float inc[8]);
    void applyStep(int lvl);
    void optReg(int lvl);
    void calcEC(int lvl, float out[3]) const;
    void propagateUp(int srcLvl);
    void propagateDown(int srcLvl);
    bool calcResAndGS(int lvl, uint64_t imageId, float H[64], float b[8], float Hsc[64], float bsc[8], const SE3& camera, double exposure, float res[3]);

    cmlhip_ctx* mCtx;
    std::string mError;
    std::vector<std::vector<Point>> mPoints;
    std::vector<LevelInput> mLevels;             // w, h, K only (gray pointers are not kept)
    SE3 mReferenceCamera, mCurrentCamera;
    double mReferenceExposure = 1, mCurrentExposure = 1;
    float mAlphaK = 2.5f * 2.5f, mAlphaW = 150 * 150, mRegWeight = 0.45f, mCouplingWeight = 1;
    bool mSnapped = false, mIsInit = false, mSuccess = false, mFresh = false;
    int mFrameID = 0, mSnappedAt = 0;
    float mRescaleFactor = 1;
};

}  // namespace cml_amd
