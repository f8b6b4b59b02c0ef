// trace_pairs.h — host -> frame pairs of the tracked trace (DSOTracer.cpp:606-608: K R K^-1, K t; Exposure.h:119-123 with exposure times 1), formed on the
// device from the first hypothesis' result of a tracker batch.  One function for every caller (k_tracker_optimize's tail, the trace launch, the pairs
// launch of wide windows): the same inputs give the same bits wherever it runs.  Compile the including unit with FP contraction off.
#pragma once
#include "../../include/cmlhip.h"

#define TR_INLINE_HOSTS 8
struct TrackedReq {                            // the window of a tracked trace, small enough to travel in kernel arguments (up to TR_INLINE_HOSTS frames)
    cmlhip_frame_pose ref; double K[4]; int n_hosts, pad;
    cmlhip_frame_pose hosts[TR_INLINE_HOSTS];
};
__device__ __forceinline__ void tp_mul33(const double* A_, const double* B_, double* C_) {       // C = A B, row by row, left to right
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C_[3 * i + j] = (A_[3 * i] * B_[j] + A_[3 * i + 1] * B_[3 + j]) + A_[3 * i + 2] * B_[6 + j];
}
// host -> frame for one host; pose0 = {R[9], t[3], a, b} of refToNew and the new frame's exposure
__device__ __forceinline__ void tp_pair(const double* pose0, const cmlhip_frame_pose& ref, const double* K, const cmlhip_frame_pose& H, cmlhip_trace_pair& P) {
    double R[9], t[3];
    for (int k = 0; k < 9; k++) R[k] = pose0[k];
    for (int k = 0; k < 3; k++) t[k] = pose0[9 + k];
    const double an = pose0[12], bn = pose0[13];
    double Rn[9], tn[3];
    tp_mul33(R, ref.R, Rn);                                                                     // frame = refToNew o reference
    for (int i = 0; i < 3; i++) tn[i] = ((R[3 * i] * ref.t[0] + R[3 * i + 1] * ref.t[1]) + R[3 * i + 2] * ref.t[2]) + t[i];
    double HT[9], Rr[9], tr[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) HT[3 * i + j] = H.R[3 * j + i];
    tp_mul33(Rn, HT, Rr);                                                                       // Rn Rh^T
    for (int i = 0; i < 3; i++) tr[i] = tn[i] - ((Rr[3 * i] * H.t[0] + Rr[3 * i + 1] * H.t[1]) + Rr[3 * i + 2] * H.t[2]);
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    const double Km[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1}, Ki[9] = {1.0 / fx, 0, -cx / fx, 0, 1.0 / fy, -cy / fy, 0, 0, 1};
    double KR[9];
    tp_mul33(Km, Rr, KR);
    tp_mul33(KR, Ki, P.KRKi);
    for (int i = 0; i < 3; i++) P.Kt[i] = (Km[3 * i] * tr[0] + Km[3 * i + 1] * tr[1]) + Km[3 * i + 2] * tr[2];
    const double a = exp(an - H.a);
    P.aff_a = a; P.aff_b = bn - a * H.b;
}
