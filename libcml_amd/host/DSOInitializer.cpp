// DSOInitializer.cpp — see DSOInitializer.h.  Line references: src/cml/optimization/dso/DSOInitializer.cpp unless noted.
#include "DSOInitializer.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace cml_amd {

static const int kStar8[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};   // types.h:1381-1393
static const float SCALEFACTOR = 2;                                                                      // types.h:1074

// Array2D<float>::interpolate, image/Array2D.h:242-262
static float interpolate(const float* img, int w, float x, float y) {
    const int ix = (int)x, iy = (int)y;
    const float dx = x - (float)ix, dy = y - (float)iy, dxdy = dx * dy;
    const int i1 = iy * w + ix, i2 = i1 + w;
    return img[i1] * (1 - dx - dy + dxdy) + img[i1 + 1] * (dx - dxdy) + img[i2] * (dy - dxdy) + img[i2 + 1] * dxdy;
}
static float median(std::vector<float> v) {                               // maths/Utils.h:153-225 (running median of a stream)
    std::sort(v.begin(), v.end());
    const size_t n = v.size();
    return (n & 1) ? v[n / 2] : (float)(((double)v[n / 2 - 1] + (double)v[n / 2]) / 2.0);
}
// the small double products in Eigen's evaluation order: packet rows (e0 + e1) + e2, scalar row e0 + (e1 + e2)
static void matmul3d(const double A[9], const double B[9], double out[9]) {
    for (int j = 0; j < 3; j++) {
        for (int i = 0; i < 2; i++) out[3 * i + j] = (A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j]) + A[3 * i + 2] * B[6 + j];
        out[6 + j] = A[6] * B[j] + (A[7] * B[3 + j] + A[8] * B[6 + j]);
    }
}
static void inverse3d(const double m[9], double o[9]) {                   // Eigen/src/LU/InverseImpl.h:139-176
#define COF(i, j) (m[(((i) + 1) % 3) * 3 + (((j) + 1) % 3)] * m[(((i) + 2) % 3) * 3 + (((j) + 2) % 3)] - m[(((i) + 1) % 3) * 3 + (((j) + 2) % 3)] * m[(((i) + 2) % 3) * 3 + (((j) + 1) % 3)])
    const double c0 = COF(0, 0), c1 = COF(1, 0), c2 = COF(2, 0);
    const double invdet = 1.0 / ((c0 * m[0] + c1 * m[3]) + c2 * m[6]);
    for (int i = 0; i < 3; i++) for (int j = 1; j < 3; j++) o[j * 3 + i] = COF(i, j) * invdet;
    o[0] = c0 * invdet; o[1] = c1 * invdet; o[2] = c2 * invdet;
#undef COF
}
// Eigen::LDLT<Matrix<float,6,6>>::solve (Eigen/src/Cholesky/LDLT.h:300-396,560-600): diagonal pivoting, tiny pivots pseudo-inverted
static void ldlt6f(const float Ain[36], const float b[6], float x[6]) {
    const int n = 6;
    float A[36], temp[6];
    int tr[6];
    std::memcpy(A, Ain, sizeof A);
#define M(i, j) A[(i) * n + (j)]
    for (int k = 0; k < n; k++) {
        int big = k;
        float best = std::fabs(M(k, k));
        for (int i = k + 1; i < n; i++) if (std::fabs(M(i, i)) > best) { best = std::fabs(M(i, i)); big = i; }
        tr[k] = big;
        if (k != big) {
            const int s = n - big - 1;
            for (int j = 0; j < k; j++) std::swap(M(k, j), M(big, j));
            for (int i = 0; i < s; i++) std::swap(M(big + 1 + i, k), M(big + 1 + i, big));
            std::swap(M(k, k), M(big, big));
            for (int i = k + 1; i < big; i++) std::swap(M(i, k), M(big, i));
        }
        const int rs = n - k - 1;
        if (k > 0) {
            for (int j = 0; j < k; j++) temp[j] = M(j, j) * M(k, j);
            float s = 0;
            for (int j = 0; j < k; j++) s += M(k, j) * temp[j];
            M(k, k) -= s;
            for (int i = 0; i < rs; i++) { float s2 = 0; for (int j = 0; j < k; j++) s2 += M(k + 1 + i, j) * temp[j]; M(k + 1 + i, k) -= s2; }
        }
        const float akk = M(k, k);
        const bool valid = std::fabs(akk) > 0.0f;
        if (k == 0 && !valid) { for (int j = 0; j < n; j++) tr[j] = j; break; }
        if (rs > 0 && valid) for (int i = 0; i < rs; i++) M(k + 1 + i, k) /= akk;
    }
    for (int i = 0; i < n; i++) x[i] = b[i];
    for (int k = 0; k < n; k++) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
    for (int i = 0; i < n; i++) { float s = x[i]; for (int j = 0; j < i; j++) s -= M(i, j) * x[j]; x[i] = s; }
    for (int i = 0; i < n; i++) { if (std::fabs(M(i, i)) > 1.17549435e-38f) x[i] /= M(i, i); else x[i] = 0; }
    for (int i = n - 1; i >= 0; i--) { float s = x[i]; for (int j = i + 1; j < n; j++) s -= M(j, i) * x[j]; x[i] = s; }
    for (int k = n - 1; k >= 0; k--) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
#undef M
}

// ------------------------------------------------------------------------------------------------ setFirst, :7-107
bool DSOInitializer::setFirst(const std::vector<LevelInput>& levels, const SE3& referenceCamera, double referenceExposure) {
    const int nl = std::min(5, (int)levels.size());                        // mNumPyramidLevel, :11
    mPoints.assign(nl, {});
    mLevels.assign(levels.begin(), levels.begin() + nl);
    mReferenceCamera = referenceCamera; mReferenceExposure = referenceExposure;
    mIsInit = false;
    for (int lvl = 0; lvl < nl; lvl++) {
        const LevelInput& L = levels[lvl];
        for (size_t q = 0; q < L.px.size(); q++) {
            const int x = L.px[q], y = L.py[q];
            const int pad = 2;
            if (!(y >= pad + 1 && y < L.h - pad - 2 && x >= pad + 1 && x < L.w - pad - 2)) continue;   // the loop bounds of :44-45
            Point p;
            std::memset(&p.d, 0, sizeof p.d);
            p.px = x + 0.1; p.py = y + 0.1;                                // :53-54 (double literal, float store)
            p.idepth = 1; p.d.iR = 1; p.d.is_good = 1; p.lastHessian = 0; p.d.last_hessian_new = 0; p.d.idepth_new = 1;
            for (int idx = 0; idx < 8; idx++) {
                const float posx = p.px + (float)kStar8[idx][0], posy = p.py + (float)kStar8[idx][1];
                p.d.p_pattern[idx][0] = posx; p.d.p_pattern[idx][1] = posy; p.d.p_pattern[idx][2] = 1;
                p.d.color[idx] = interpolate(L.gray, L.w, posx, posy);     // :67
            }
            p.d.outlier_th = 8 * mSettingOutlierTH;                        // :84
            for (int k = 0; k < 10; k++) { p.neighbours[k] = -1; p.neighboursDist[k] = 0; p.jb[k] = 0; }
            mPoints[lvl].push_back(p);
        }
        mLevels[lvl].px.clear(); mLevels[lvl].py.clear(); mLevels[lvl].gray = nullptr;
        if (mPoints[lvl].size() < 10) { mError = "fewer than 10 points at level " + std::to_string(lvl); return false; }   // :92-94
    }
    makeNN();                                                              // :97
    mSnapped = false; mFrameID = 0; mSnappedAt = 0; mSuccess = false;      // :99-102
    mIsInit = true; mFresh = true;
    return true;
}

// ------------------------------------------------------------------------------------------------ makeNN, :919-984 (PointGrid, utils/KDTree.h)
namespace {
struct NN { int index; double distance; };
struct PointGrid {
    static const int GW = 64, GH = 64;
    const std::vector<DSOInitializer::Point>& pts; int maxx, maxy;
    std::vector<int> cell[GW][GH];
    PointGrid(const std::vector<DSOInitializer::Point>& p, int w, int h) : pts(p), maxx(w), maxy(h) {
        for (size_t i = 0; i < p.size(); i++) {
            int gx, gy; gridPos(p[i].px, p[i].py, gx, gy);
            if (gx < 0 || gy < 0 || gx >= GW || gy >= GH) continue;
            cell[gx][gy].push_back((int)i);
        }
    }
    void gridPos(float x, float y, int& gx, int& gy) const { gx = (int)((x - 0) * GW / (maxx - 0)); gy = (int)((y - 0) * GH / (maxy - 0)); }
    void searchInRadiusNum(float x, float y, size_t num, std::vector<NN>& result) const {
        result.clear();
        int cx, cy; gridPos(x, y, cx, cy);
        size_t radius = 0;
        int x0 = cx, x1 = cx, y0 = cy, y1 = cy;
        while (radius < (size_t)GW) {
            x0 = cx - (int)radius; x1 = cx + (int)radius; y0 = cy - (int)radius; y1 = cy + (int)radius;
            size_t n = 0;
            for (int gx = x0; gx <= x1; gx++) { if (gx < 0 || gx >= GW) continue; for (int gy = y0; gy <= y1; gy++) { if (gy < 0 || gy >= GH) continue; n += cell[gx][gy].size(); } }
            if (n >= num) break;
            radius++;
        }
        for (int gx = x0; gx <= x1; gx++) {
            if (gx < 0 || gx >= GW) continue;
            for (int gy = y0; gy <= y1; gy++) {
                if (gy < 0 || gy >= GH) continue;
                for (int index : cell[gx][gy]) {
                    const double ddx = (double)x - (double)pts[index].px, ddy = (double)y - (double)pts[index].py;
                    result.push_back(NN{index, std::sqrt(ddx * ddx + ddy * ddy)});
                }
            }
        }
        std::sort(result.begin(), result.end(), [](const NN& a, const NN& b) { return a.distance < b.distance; });
        if (result.size() > num) result.resize(num);
    }
};
}  // namespace

void DSOInitializer::makeNN() {
    const float NNDistFactor = 0.05;
    const int nl = (int)mPoints.size(), nn = 10;
    std::vector<PointGrid*> idx(nl);
    for (int i = 0; i < nl; i++) idx[i] = new PointGrid(mPoints[i], mLevels[i].w, mLevels[i].h);
    std::vector<NN> r, r1;
    for (int lvl = 0; lvl < nl; lvl++)
        for (Point& p : mPoints[lvl]) {
            idx[lvl]->searchInRadiusNum(p.px, p.py, nn, r);
            float sumDF = 0;
            for (int k = 0; k < nn && k < (int)r.size(); k++) {
                p.neighbours[k] = r[k].index;
                const float df = expf(-r[k].distance * NNDistFactor);
                sumDF += df; p.neighboursDist[k] = df;
            }
            for (int k = 0; k < nn; k++) p.neighboursDist[k] *= 10 / sumDF;
            if (lvl < nl - 1) {
                idx[lvl + 1]->searchInRadiusNum(p.px / SCALEFACTOR, p.py / SCALEFACTOR, 1, r1);
                p.parent = r1.empty() ? 0 : r1[0].index;
                p.parentDist = r1.empty() ? -1 : expf(-r1[0].distance * NNDistFactor);
            } else { p.parent = -1; p.parentDist = -1; }
        }
    for (PointGrid* g : idx) delete g;
}

// ------------------------------------------------------------------------------------------------ the per-level helpers, :752-917
void DSOInitializer::resetPoints(int lvl) {                                // :844-875
    const int top = (int)mPoints.size() - 1;
    for (Point& p : mPoints[lvl]) {
        p.d.energy[0] = 0; p.d.energy[1] = 0;
        p.d.idepth_new = p.idepth;
        if (lvl == top && !p.d.is_good) {
            float snd = 0, sn = 0;
            for (int n = 0; n < 10; n++) {
                if (p.neighbours[n] == -1) continue;
                const Point& o = mPoints[lvl][p.neighbours[n]];
                if (!o.d.is_good) continue;
                snd += o.d.iR; sn += 1;
            }
            if (sn > 0) { p.d.is_good = 1; p.d.iR = p.idepth = p.d.idepth_new = snd / sn; }
        }
    }
}
void DSOInitializer::doStep(int lvl, float lambda, const float inc[8]) {   // :877-905
    const float maxPixelStep = 0.25, idMaxStep = 1e10;
    for (Point& p : mPoints[lvl]) {
        if (!p.d.is_good) continue;
        float dot = 0;                                                     // mJbBuffer[i].head<8>().dot(inc): one SSE packet pair, (0+4 .. ) reduced
        { const float z[8] = {p.jb[0] * inc[0], p.jb[1] * inc[1], p.jb[2] * inc[2], p.jb[3] * inc[3], p.jb[4] * inc[4], p.jb[5] * inc[5], p.jb[6] * inc[6], p.jb[7] * inc[7]};
          const float a0 = z[0] + z[4], a1 = z[1] + z[5], a2 = z[2] + z[6], a3 = z[3] + z[7];
          dot = (a0 + a2) + (a1 + a3); }
        const float b = p.jb[8] + dot;
        float step = -b * p.jb[9] / (1 + lambda);
        float maxstep = maxPixelStep * p.d.maxstep;
        if (maxstep > idMaxStep) maxstep = idMaxStep;
        if (step > maxstep) step = maxstep;
        if (step < -maxstep) step = -maxstep;
        float newIdepth = p.idepth + step;
        if (newIdepth < 1e-3) newIdepth = 1e-3;
        if (newIdepth > 50) newIdepth = 50;
        p.d.idepth_new = newIdepth;
    }
}
void DSOInitializer::applyStep(int lvl) {                                  // :907-917 (+ std::swap(mJbBuffer, mJbBuffer_new))
    for (Point& p : mPoints[lvl]) {
        if (!p.d.is_good) { p.idepth = p.d.idepth_new = p.d.iR; }
        else {
            p.d.energy[0] = p.d.energy_new[0]; p.d.energy[1] = p.d.energy_new[1];
            p.d.is_good = p.d.is_good_new;
            p.idepth = p.d.idepth_new;
            p.lastHessian = p.d.last_hessian_new;
        }
        for (int k = 0; k < 10; k++) std::swap(p.jb[k], p.d.jb[k]);
    }
}
void DSOInitializer::optReg(int lvl) {                                     // :810-842
    if (!mSnapped) { for (Point& p : mPoints[lvl]) p.d.iR = p.initialiR; return; }
    for (Point& p : mPoints[lvl]) {
        if (!p.d.is_good) continue;
        std::vector<float> idnn;
        for (int j = 0; j < 10; j++) {
            if (p.neighbours[j] == -1) continue;
            const Point& o = mPoints[lvl][j];                              // literal: the reference indexes the level by j, not by neighbours[j] (:825)
            if (!o.d.is_good) continue;
            idnn.push_back(o.d.iR);
        }
        if (idnn.size() > 2) p.d.iR = (1 - mRegWeight) * p.idepth + mRegWeight * median(idnn);
        p.d.iR = (1 - mNNWeight) * p.d.iR + mNNWeight * p.initialiR;
    }
}
void DSOInitializer::calcEC(int lvl, float out[3]) const {                 // :788-808 (AccumulatorX<2>: plain float sums at these sizes' tolerance)
    if (!mSnapped) { out[0] = 0; out[1] = 0; out[2] = (float)mPoints[lvl].size(); return; }
    float a = 0, b = 0; int num = 0;
    for (const Point& p : mPoints[lvl]) {
        if (!p.d.is_good_new) continue;
        const float rOld = p.idepth - p.d.iR, rNew = p.d.idepth_new - p.d.iR;
        a += rOld * rOld; b += rNew * rNew; num++;
    }
    out[0] = mCouplingWeight * a; out[1] = mCouplingWeight * b; out[2] = (float)num;
}
void DSOInitializer::propagateUp(int srcLvl) {                             // :752-782
    std::vector<Point>& up = mPoints[srcLvl + 1];
    for (Point& p : up) { p.d.iR = 0; p.iRSumNum = 0; }
    for (const Point& p : mPoints[srcLvl]) {
        if (!p.d.is_good) continue;
        Point& par = up[p.parent];
        par.d.iR += p.d.iR * p.lastHessian; par.iRSumNum += p.lastHessian;
    }
    for (Point& p : up) if (p.iRSumNum > 0) { p.idepth = p.d.iR = (p.d.iR / p.iRSumNum); p.d.is_good = 1; }
    optReg(srcLvl + 1);
}
void DSOInitializer::propagateDown(int srcLvl) {                           // :784-808
    const int dst = srcLvl - 1;
    for (Point& p : mPoints[dst]) {
        const Point& par = mPoints[srcLvl][p.parent];
        if (!par.d.is_good || par.lastHessian < 0.1) continue;
        if (!p.d.is_good) { p.d.iR = p.idepth = p.d.idepth_new = par.d.iR; p.d.is_good = 1; p.lastHessian = 0; }
        else {
            const float newiR = (p.d.iR * p.lastHessian * SCALEFACTOR + par.d.iR * par.lastHessian) / (p.lastHessian * SCALEFACTOR + par.lastHessian);
            p.d.iR = p.idepth = p.d.idepth_new = newiR;
        }
    }
    optReg(srcLvl - 1);
}

// calcResAndGS (:451-750): the constants of :455-480, then the device call
bool DSOInitializer::calcResAndGS(int lvl, uint64_t imageId, float H[64], float b[8], float Hsc[64], float bsc[8], const SE3& camera, double exposure, float res[3]) {
    const LevelInput& L = mLevels[lvl];
    const double K[9] = {L.K[0], 0, L.K[2], 0, L.K[1], L.K[3], 0, 0, 1};
    double Ki[9], R[9], RKi[9];
    inverse3d(K, Ki);                                                      // Matrix33 Ki = K.inverse(), :456
    const SE3 refToNew = camera * mReferenceCamera.inverse();              // mReference->getCamera().to(camera), Camera.h:297
    refToNew.matrix(R);
    matmul3d(R, Ki, RKi);                                                  // (R * Ki).cast<float>(), :471
    cmlhip_init_params P;
    std::memset(&P, 0, sizeof P);
    for (int i = 0; i < 9; i++) P.RKi[i] = (float)RKi[i];
    for (int i = 0; i < 3; i++) P.t[i] = (float)refToNew.t[i];
    P.fx = (float)K[0]; P.fy = (float)K[4]; P.cx = (float)K[2]; P.cy = (float)K[5];
    P.aff_a = (float)(exposure / mReferenceExposure); P.aff_b = 0;        // :476-479
    P.huber = mHuberThreshold; P.alpha_w = mAlphaW; P.alpha_k = mAlphaK; P.coupling_weight = mCouplingWeight;
    double xi[6];
    camera.log(xi);                                                        // SE3(camera).log().head<3>(), :738
    for (int i = 0; i < 3; i++) P.tlog[i] = (float)xi[i];
    P.t_sqnorm = refToNew.t[0] * refToNew.t[0] + refToNew.t[1] * refToNew.t[1] + refToNew.t[2] * refToNew.t[2];
    std::vector<Point>& pts = mPoints[lvl];
    std::vector<cmlhip_init_point> d(pts.size());
    for (size_t i = 0; i < pts.size(); i++) d[i] = pts[i].d;
    numCalcCalls++;
    const int rc = cmlhip_initializer_calc_res_and_gs(mCtx, imageId, lvl, &P, (int)d.size(), d.data(), H, b, Hsc, bsc, res);
    if (rc != CMLHIP_OK && rc != CMLHIP_ERR_NONFINITE) { mError = cmlhip_last_error(mCtx); return false; }
    for (size_t i = 0; i < pts.size(); i++) pts[i].d = d[i];
    return true;
}

// ------------------------------------------------------------------------------------------------ tryInitialize, :115-341
int DSOInitializer::tryInitialize(uint64_t imageId, const SE3& frameCamera, double frameExposure) {
    if (!mIsInit) { mError = "setFirst has not succeeded"; return 0; }
    if (mFresh) { mCurrentCamera = frameCamera; mFresh = false; }         // :128-133
    mCurrentExposure = frameExposure;
    const int nl = (int)mPoints.size();
    const float wM[8] = {mScaleRotation, mScaleRotation, mScaleRotation, mScaleTranslation, mScaleTranslation, mScaleTranslation, mScaleLightA, mScaleLightB};
    int maxIterations[8] = {5, 5, 10, 30, 50, 50, 50, 50};                // :160-165
    mAlphaK = 2.5 * 2.5; mAlphaW = 150 * 150; mRegWeight = mRegulalizationWeight; mCouplingWeight = 1;   // :168-171
    if (!mSnapped) {                                                       // :173-192 (inverse depth map of ones)
        for (int k = 0; k < 3; k++) mCurrentCamera.t[k] = mReferenceCamera.t[k];
        for (int lvl = 0; lvl < nl; lvl++)
            for (Point& p : mPoints[lvl]) { p.initialiR = 1; p.d.iR = 1; p.d.idepth_new = 1; p.lastHessian = 0; }
    }
    for (int lvl = nl - 1; lvl >= 0; lvl--) {
        if (lvl < nl - 1) propagateDown(lvl + 1);
        float H[64], Hsc[64], b[8], bsc[8], resOld[3];
        resetPoints(lvl);
        if (!calcResAndGS(lvl, imageId, H, b, Hsc, bsc, mCurrentCamera, mCurrentExposure, resOld)) return -1;
        if (resOld[2] == 0) return -1;
        applyStep(lvl);
        float lambda = 0.1, eps = 1e-4;
        int fails = 0, iteration = 0;
        const float norm = 0.01f / (float)(mLevels[lvl].w * mLevels[lvl].h);
        while (true) {
            float Hl[64], bl[8];
            for (int i = 0; i < 64; i++) Hl[i] = H[i];
            for (int i = 0; i < 8; i++) Hl[i * 9] *= (1 + lambda);
            for (int i = 0; i < 64; i++) Hl[i] -= Hsc[i] * (1 / (1 + lambda));
            for (int i = 0; i < 8; i++) bl[i] = b[i] - bsc[i] * (1 / (1 + lambda));
            for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) Hl[i * 8 + j] = wM[i] * Hl[i * 8 + j] * wM[j] * norm;     // :222
            for (int i = 0; i < 8; i++) bl[i] = wM[i] * bl[i] * norm;
            float inc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, H6[36], x6[6];
            for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) H6[i * 6 + j] = Hl[i * 8 + j];
            ldlt6f(H6, bl, x6);                                            // fixAffine, :227-230
            for (int i = 0; i < 6; i++) inc[i] = -(wM[i] * x6[i]);
            double xi[6];
            for (int i = 0; i < 6; i++) xi[i] = (double)inc[i];
            bool finite = true;
            for (int i = 0; i < 6; i++) finite = finite && std::isfinite(xi[i]);
            if (!finite) { mError = "SE3::exp failed"; return -1; }
            const SE3 newCamera = SE3::exp(xi) * mCurrentCamera;           // mCurrentCamera.compose(Camera(se3)), Camera.h:289-291
            const double newExposure = mCurrentExposure;                   // add(inc[6], inc[7]) with a fixed affine
            doStep(lvl, lambda, inc);
            float Hn[64], Hscn[64], bn[8], bscn[8], resNew[3], regEnergy[3];
            if (!calcResAndGS(lvl, imageId, Hn, bn, Hscn, bscn, newCamera, newExposure, resNew)) return -1;
            if (resNew[2] == 0) return -1;
            calcEC(lvl, regEnergy);
            const float eTotalNew = (resNew[0] + resNew[1] + regEnergy[1]), eTotalOld = (resOld[0] + resOld[1] + regEnergy[0]);
            const bool accept = eTotalOld > eTotalNew;
            if (accept) {
                if (resNew[1] == mAlphaK * mPoints[lvl].size()) mSnapped = true;       // :269-271
                std::memcpy(H, Hn, sizeof H); std::memcpy(b, bn, sizeof b); std::memcpy(Hsc, Hscn, sizeof Hsc); std::memcpy(bsc, bscn, sizeof bsc);
                std::memcpy(resOld, resNew, sizeof resOld);
                mCurrentCamera = newCamera; mCurrentExposure = newExposure;
                applyStep(lvl);
                optReg(lvl);
                lambda *= 0.5; fails = 0;
                if (lambda < 0.0001) lambda = 0.0001;
                numAccepted++;
            } else {
                fails++; lambda *= 4;
                if (lambda > 10000) lambda = 10000;
                numRejected++;
            }
            float n2 = 0;
            for (int i = 0; i < 8; i++) n2 += inc[i] * inc[i];
            if (!(std::sqrt(n2) > eps) || iteration >= maxIterations[lvl] || fails >= 2) break;
            iteration++;
        }
    }
    for (int i = 0; i < nl - 1; i++) propagateUp(i);                       // :309-311
    mFrameID++;
    if (!mSnapped) mSnappedAt = 0;
    if (mSnapped && mSnappedAt == 0) mSnappedAt = mFrameID;
    mSuccess = mSnapped && mFrameID > mSnappedAt + 5;
    if (mSuccess) {                                                        // onInitializationSuccess, :343-375
        std::vector<float> allIR;
        for (const Point& p : mPoints[0]) if (p.d.is_good) allIR.push_back(p.d.iR);
        if (!allIR.empty()) {
            mRescaleFactor = 0.5f / median(allIR);
            for (int k = 0; k < 3; k++) mCurrentCamera.t[k] = mCurrentCamera.t[k] / mRescaleFactor;
        }
    }
    return mSuccess ? 1 : 0;
}

void DSOInitializer::initializedPoints(std::vector<int>& index, std::vector<float>& idepth) const {     // :408-433
    index.clear(); idepth.clear();
    const size_t n = mPoints[0].size(), desired = std::min(n, (size_t)mSettingsDesiredPointDensity);
    for (size_t j = 0; j < desired; j++) {
        const size_t i = j * n / desired;
        if (!mPoints[0][i].d.is_good) continue;
        index.push_back((int)i); idepth.push_back(mPoints[0][i].d.iR * mRescaleFactor);
    }
}

}  // namespace cml_amd
