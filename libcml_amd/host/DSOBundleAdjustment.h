// DSOBundleAdjustment.h — host-side mirror of the reference operator interface for the sliding-window BA,
// driving the gfx950 device layer through the C ABI (include/cmlhip.h).
//
// Mirrors CML::Optimization::DSOBundleAdjustment (src/cml/optimization/dso/DSOBundleAdjustment.h:26-101,235-288)
// and its helper types DSOFrame / DSOPoint / DSOResidual (DSOFrame.h, DSOPoint.h, DSOResidual.h): same method
// names, argument meaning, parameter names/defaults and error behaviour (bool returns, never throws).  The
// reference reaches frames/points through Map/Frame/MapPoint/PrivateData objects; this mirror takes the same
// quantities as flat values (image id, pose, exposure, corner, idepth, colours, weights), which is exactly what
// the modified bodies of the reference methods would pass down.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/cmlhip.h"
#include "se3.h"

namespace cml_amd {

enum DSOResidualState { DSORES_IN = 0, DSORES_OOB, DSORES_OUTLIER };          // DSOResidual.h:14-16

struct DSOFrame {                                                              // DSOFrame.h:17-246
    int id = -1, keyid = -1;
    uint64_t image_id = 0;
    double delta[8] = {0}, delta_prior[8] = {0}, prior[8] = {0}, prior_zero[10] = {0};
    double frameEnergyTH = 8 * 8 * 8;
    double nullspaces_pose[36] = {0}, nullspaces_scale[6] = {0}, nullspaces_affine[8] = {0};   // column-major like Eigen
    SE3 PRE_worldToCam, PRE_camToWorld;
    double ab_exposure = 1;
    bool flaggedForMarginalization = false;
    int numMarginalized = 0, numResidualsOut = 0;                    // DSOFrame.h:229-244
    double state[10] = {0}, state_zero[10] = {0}, state_backup[10] = {0}, state_scaled[10] = {0}, step[10] = {0};
    SE3 worldToCam_evalPT;

    void setState(const double s[10], const double sc[4]);
    void setStateScaled(const double ss[10], const double sc[4]);
    void setStateZero(const double sz[10], const double sc[4]);
    void setEvalPT(const SE3& w2c, const double s[10], const double sc[4]);
    void setEvalPT_scaled(const SE3& w2c, const Exposure& aff, const double sc[4]);
    void backupState() { for (int i = 0; i < 10; i++) state_backup[i] = state[i]; }
    void loadSateBackup(const double sc[4]) { setState(state_backup, sc); }
    void doStepFromBackup(const double sc[4]);
    void setStep(const double s[10]);
    Exposure aff_g2l() const { return Exposure(ab_exposure, state_scaled[6], state_scaled[7]); }
    Exposure aff_g2l_0(const double sc[4]) const { return Exposure(ab_exposure, state_zero[6] * sc[2], state_zero[7] * sc[3]); }
    float getB0(float scaleB) const { return (float)(state_zero[7] * scaleB); }
};

struct DSOPoint {                                                              // DSOPoint.h:44-167 (+ the MapPoint fields BA reads)
    float x = 0, y = 0;
    double idepth = 0;
    int host = -1;
    float colors[8] = {0}, weights[8] = {0};
    float idepth_zero = 0, idepth_backup = 0, deltaF = 0, priorF = 0;
    float HdiF = 0, bdSumF = 0;
    double step = 0;
    bool hasDepthPrior = false;
    int numGoodResiduals = 0;
    float idepth_hessian = 0, maxRelBaseline = 0;
    int lastResidual[2] = {-1, -1};                                  // residual index / state of the newest two frames (DSOPoint.h:120-140)
    int lastResidualState[2] = {DSORES_OOB, DSORES_OOB};
    bool toMarginalize = false, marginalized = false;                // groups DSOTOMARGINALIZE / DSOMARGINALIZED
    double uncertainty = 0;
    bool alive = true;
};

struct DSOResidual {                                                           // DSOResidual.h:72-156
    int point = -1, target = -1;
    int state_state = DSORES_IN, state_NewState = DSORES_OUTLIER;
    double state_energy = 0, state_NewEnergy = 0, state_NewEnergyWithOutlier = 0;
    bool isLinearized = false, isActiveAndIsGoodNEW = false;
    float centerProjectedTo[3] = {0, 0, 0};
    bool alive = true;
};

class DSOBundleAdjustment {
public:
    explicit DSOBundleAdjustment(cmlhip_ctx* ctx);

    // ---- parameters, names and defaults of BA.h:235-288
    int    mNumIterations = 4;
    double mHuberThreshold = 9.0, mSettingOutlierTHSumComponent = 50.0 * 50.0, mThOptIterations = 1.2;
    double mScaleRotation = 1.0, mScaleTranslation = 0.5, mScaleLightA = 10.0, mScaleLightB = 1000.0, mScaleF = 50.0, mScaleC = 50.0;
    bool   mForceAccept = true, mFixLambda = true;
    double mFixedLambda = 1e-5;
    int    mIdepthFixPrior = 50 * 50;
    double mSolverModeDelta = 0.00001;
    bool   mOptimizeA = true, mOptimizeB = true, mMixedBundleAdjustment = false, mAddLinearizedPoints = false;
    bool   mDisableMarginalization = true, mOptimizeCalibration = false, mAbortBAOnFailture = false;
    double mMinIdepthHMarg = 50.0;                           // BA.h:263
    int    mMaxFrames = 6, mMinFrameAge = 1;                 // BA.h:271-272
    bool   mResidentLoop = true;                             // run(): keep the iteration loop on the device when the parameters allow it
    bool   mRelaxedArithmetic = false;                       // cmlhip_ba_set_arithmetic(CMLHIP_ARITH_RELAXED) for the iterations of the device-resident loop (include/cmlhip.h; default: exact)
    bool   mKeepResidualEnergies = false;                    // run()'s closing pass also reads state_energy / state_NewEnergy / state_NewState of every residual back (nothing on the host uses them)
    bool   mLeanResidentOutputs = true;                      // cmlhip_ba_set_resident_outputs(LEAN) for the resident passes of run(): nothing this class reads is affected (FULL when mKeepResidualEnergies)
    double mCPriorValue = 5e9;                               // BA.cpp:2136-2137 (mCPrior is only assigned inside calcLEnergy)

    // ---- reference interface (BA.h:28-85), flat arguments
    void setCalibration(double fx, double fy, double cx, double cy, int w, int h);      // BA.cpp:419-425
    int  addNewFrame(uint64_t image_id, const SE3& worldToCam, const Exposure& exposure);       // BA.cpp:417-462; returns DSOFrame::id
    int  addPoint(float x, float y, double idepth, int host, const float colors[8], const float weights[8], bool hasDepthPrior);  // addPoints, BA.cpp:382-415
    bool run(bool updatePointsOnly = false);                                                     // BA.cpp:744-910
    // addNewFrame / a batch of addPoint calls hand the new points and residuals to the library's window (cmlhip_ba_window_append_*) when they are made —
    // where the reference's addPoints / addNewFrame build them — so that run() finds the window complete; run() hands over whatever is still missing
    void handOverNewEntries() { if (!syncWindowAppends()) mError.clear(); }
    // The same run with the iteration loop resident on the device (forceAccept + fixLambda, no early break):
    // preamble and epilogue as run(), mNumIterations x cmlhip_ba_iteration_async in between, no host round trip per iteration.
    bool runResident(bool updatePointsOnly = false);
    bool runResidentStepwise(bool updatePointsOnly = false);                                     // the same with a host wait behind every stage (hybrid term; CMLHOST_RUN_STEPWISE=1)
    bool runHostLoop(bool updatePointsOnly = false);                                             // the literal loop: one device call per reference statement
    // pieces of runResident, for callers that keep iterating (bench): states to the device / k iterations / states back
    bool beginResident(bool updatePointsOnly = false);
    bool iterateResident(int k, double lambda);
    bool endResident(double* lastEnergy = nullptr);
    void nullspaceBasis(std::vector<double>& U7n) const;                                         // orthonormal basis used by orthogonalize
    // ---- marginalisation (BA.h:34-46), once per keyframe after run()
    void flagFramesForMarginalization(int numImmaturePerFrame = 0);                              // BA.cpp:603-716
    // the same with the immature count of EVERY frame (frame->getReferenceGroupMapPoints(immatureGroup).size(), BA.cpp:617), and — as the
    // reference calls it from addNewFrame BEFORE the frame is added (BA.cpp:428) — the exposure the new frame's affine test runs against
    // is that of the newest frame of the window (getFrames().back(), :614)
    void flagFramesForMarginalization(const std::vector<int>& immaturePerFrame);
    bool tryMarginalize();                                                                       // BA.cpp:2240-2363 (device: relinearize + fixLinearization)
    bool marginalizePointsF();                                                                   // BA.cpp:2466-2513 (device: MARGINALIZED accumulation)
    std::vector<int> marginalizeFrames();                                                        // BA.cpp:718-742; returns the removed DSOFrame ids (before renumbering)
    void marginalizeFrame(int frameId);                                                          // BA.cpp:464-601
    double calcMEnergy() const;                                                                  // BA.cpp:2095-2117
    double calcLEnergy();                                                                        // BA.cpp:2119-2208
    const std::vector<double>& marginalizedHessian() const { return mMarginalizedHessian; }
    const std::vector<double>& marginalizedB() const { return mMarginalizedB; }
    const std::vector<int>& getOutliers() const { return mOutliers; }                            // point indices dropped by the last run
    void computeNullspaces(std::vector<double>& out7) const;                                     // BA.cpp:2365-2417
    // ---- hybrid ORB term (BA.cpp:2574-2729, "mixedBundleAdjustment"): the INDIRECTGROUP map points of the window's frames
    // (world coordinates) and their feature observations {DSOFrame id, point index, undistorted position}, handed over flat;
    // solveSystem mixes the indirect pose solution into x between the solve and the orthogonalisation (BA.cpp:1327-1329)
    void setIndirectPoints(const std::vector<double>& worldXYZ, const std::vector<cmlhip_reproj_obs>& observations);
    void indirectUncertaintyFrom(const std::vector<double>& Jp);                // setUncertainty of the indirect points, BA.cpp:2690-2692
    bool addIndirectToProblem(std::vector<double>& X);
    const std::vector<double>& indirectUncertainty() const { return mIndirectUncertainty; }     // MapPoint::setUncertainty, BA.cpp:2690-2692
    const std::vector<double>& lastIndirectX() const { return mIndirectX; }
    const std::vector<double>& lastX() const { return mX; }

    // ---- state access for the caller (what the reference writes back through Frame/MapPoint setters)
    std::vector<DSOFrame>& getFrames() { return mFrames; }
    std::vector<DSOPoint>& getPoints() { return mPoints; }
    std::vector<DSOResidual>& getResiduals() { return mResiduals; }
    const std::string& lastError() const { return mError; }

    // ---- statistics of the last run (BA.h:215-233)
    std::vector<double> statEnergyP, statXNorm, statHessianP, statHessianSC, statBP, statBSC;
    int lastIterations = 0, statRejected = 0;                // iterations of the last run / rejected steps since construction
    double lastRunUs[6] = {0, 0, 0, 0, 0, 0};                // host clock of the last runResident(): window commit | preamble pass enqueued | resident state staged | iterations enqueued | cmlhip_ba_finish_run (the one host wait) | bookkeeping
    double lastLambda = 0;
    // exposed for tests
    void computeAdjoints();
    void computeDelta();
    const std::vector<double>& adHost() const { return mAdHost; }
    const std::vector<double>& adTarget() const { return mAdTarget; }
    const std::vector<float>& adHTdeltaF() const { return mAdHTdeltaF; }
    void framePairs(std::vector<cmlhip_ba_pair>& out) const;                                     // DSOFramePrecomputed, DSOFrame.h:248-291
    void orthogonalize(std::vector<double>& x) const;                                            // BA.cpp:1196-1261

private:
    std::vector<cmlhip_ba_point> mUploadPoints; std::vector<cmlhip_ba_residual> mUploadResiduals;      // uploadWindow's arrays, kept between keyframes
    int setPairs(const std::vector<cmlhip_ba_pair>& pairs);
    std::vector<cmlhip_ba_pair> mPairsSent; bool mPairsValid = false;      // the pair records the device holds for the current upload
    bool uploadWindow();
    bool syncWindowAppends();                                                  // hands the library the points / residuals added since the last hand-over (cmlhip_ba_window_append_*)
    size_t mWinPoints = 0, mWinResiduals = 0;                                  // how much of mPoints / mResiduals the library's window holds (index for index)
    unsigned mWinGeneration = ~0u;                          // cmlhip_ba_window_generation of the window this object's entries live in (the owner token)
    int mDeadSinceCompact = 0, mLinearizedAlive = 0;                           // entries dropped since the lists were renumbered; residuals that may carry isLinearized
    std::vector<int> mScratchCount;
    std::vector<double> mDynIdepth; std::vector<float> mDynZero, mDynPrior;    // per-point values refreshed at every commit
    bool isOOB(int p, const std::vector<int>& toMarg) const;                  // BA.cpp:2515-2554
    void removePoint(int p, bool marginalize, bool sweep = true);             // DSOContext.h:94-111 (sweep: removePointsWithoutResidual behind it, :218-229)
    void compactDead();                                                       // drops dead points / residuals from the lists and renumbers (the reference's sets simply lose them)
    void removeFrame(int f);                                                  // DSOContext.h:154-174
    void removePointsWithoutResidual();
    void fillAccumIn(cmlhip_ba_accum_in& in, std::vector<double>& prior, std::vector<double>& dprior, double cdelta[4], double cprior[4]);
    bool runPreamble(double lastEnergy[3], bool enqueueOnly = false);
    void closingBookkeeping(const std::vector<int>& st, const std::vector<unsigned char>& good, const std::vector<int>& ns,
                            const std::vector<float>& e, const std::vector<float>& ne, const std::vector<float>& nw, const unsigned char* packed);
    bool runEpilogue(double lastEnergy[3]);
    bool linearizeAll(bool fixLinearization, double energy[3], std::vector<double>* idepthOut = nullptr, std::vector<float>* pointAccOut = nullptr, bool applyToo = false, bool enqueueOnly = false);
    bool solveSystem(int iteration, double lambda);
    bool doStepFromBackup(bool fixCamera);
    void backupState();
    void scales(double sc[4]) const { sc[0] = mScaleTranslation; sc[1] = mScaleRotation; sc[2] = mScaleLightA; sc[3] = mScaleLightB; }
    bool fail(const std::string& what, int rc);

    cmlhip_ctx* mCtx;
    cmlhip_ba_params mPrm{};
    bool mHaveCalib = false;
    int mFrameKeyCounter = 0;
    std::vector<DSOFrame> mFrames;
    std::vector<DSOPoint> mPoints;
    std::vector<DSOResidual> mResiduals;
    std::vector<SE3> mRelT; std::vector<double> mRelR; int mRelValidFor = -1;   // addPoint: host -> target poses at the evaluation points (rebuilt after computeAdjoints)
    std::vector<std::vector<int>> mPointRes;        // residual indices of every point (DSOPoint::residuals, DSOPoint.h:87), dead ones included until compactDead
    std::vector<int> mCompactPmap, mCompactRmap; std::vector<unsigned char> mCompactPAlive, mCompactRAlive;      // compactDead's renumbering tables (kept: no allocation per keyframe)
    std::vector<int> mActive;                       // indices of residuals uploaded (alive), device order
    std::vector<int> mActivePoints, mPointSlot;     // device point order <-> mPoints
    std::vector<int> mOutliers;
    std::vector<double> mAdHost, mAdTarget, mMarginalizedHessian, mMarginalizedB, mX;
    std::vector<float> mAdHTdeltaF;
    std::vector<double> mIndirectPoints, mIndirectUncertainty, mIndirectX;
    std::vector<cmlhip_reproj_obs> mIndirectObs;
    double mCDeltaF[4] = {0, 0, 0, 0};
    std::string mError;
};

}  // namespace cml_amd
