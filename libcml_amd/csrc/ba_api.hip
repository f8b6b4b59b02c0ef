// ba_api.hip — extern "C" entry points of the bundle-adjustment part of include/cmlhip.h.
// Host side: index bookkeeping (bit-exact maps), uploads, launches, readbacks.  No CPU compute fallback.
#include "cmlhip_internal.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "ba_common.h"
#include "ba_frames.h"
#include "ba_finish.h"
#include "reproj_dev.h"
#include <cmath>

static int ba_check(cmlhip_ctx* c, bool need_pairs) {
    if (!c) return CMLHIP_ERR_INVALID;
    CML_REQUIRE(c, c->ba_prm_set, CMLHIP_ERR_STATE, "cmlhip_ba_set_params not called");
    CML_REQUIRE(c, c->ba_uploaded, CMLHIP_ERR_STATE, "cmlhip_ba_upload_window not called");
    if (need_pairs) CML_REQUIRE(c, c->ba_pairs_set, CMLHIP_ERR_STATE, "cmlhip_ba_set_pairs not called");
    return CMLHIP_OK;
}

// a launch that reads what the caller has just staged: inside an upload scope (cmlhip_upload_scope_begin) it waits for the scope's one packed copy
template <class F> static int cml_defer(cmlhip_ctx* c, F&& f) {
    if (c->h2d_scope) { c->deferred.emplace_back(std::forward<F>(f)); return CMLHIP_OK; }
    return f();
}

int cml_make_ba_args(cmlhip_ctx* c, BAArgs& A) {
    const cmlhip_ba_params& p = c->ba_prm;
    A.N = c->N; A.P = c->P; A.R = c->R; A.w = p.w; A.h = p.h; A.opt_a = p.optimize_a; A.opt_b = p.optimize_b; A.n = 8 * c->N + 4;
    A.fx = p.fx; A.fy = p.fy; A.cx = p.cx; A.cy = p.cy; A.fxi = 1.0 / p.fx; A.fyi = 1.0 / p.fy;
    A.huber_d = (double)p.huber; A.oth_d = (double)p.outlier_th_sum; A.scale_f = p.scale_f; A.scale_c = p.scale_c;
    A.frames = c->frames.as<FrameDev>(); A.pairs = c->pairs.as<cmlhip_ba_pair>();
    A.pt_x = c->pt_x.as<float>(); A.pt_y = c->pt_y.as<float>(); A.pt_idepth = c->pt_idepth.as<double>();
    A.pt_idepth_zero = c->pt_idepth_zero.as<float>(); A.pt_prior = c->pt_prior.as<float>(); A.pt_host = c->pt_host.as<int>();
    A.r_dead = c->r_dead.as<unsigned char>();
    A.pt_colors = c->pt_colors.as<float>(); A.pt_weights = c->pt_weights.as<float>(); A.pt_backup = c->pt_backup.as<float>();
    A.pt_acc = c->pt_acc.as<float>(); A.pt_step = c->pt_step.as<double>();
    A.r_point = c->r_point.as<int>(); A.r_host = c->r_host.as<int>(); A.r_target = c->r_target.as<int>(); A.r_state = c->r_state.as<int>();
    A.r_new_state = c->r_new_state.as<int>(); A.r_energy = c->r_energy.as<float>(); A.r_new_energy = c->r_new_energy.as<float>();
    A.r_new_energy_wo = c->r_new_energy_wo.as<float>(); A.r_ret_energy = c->r_ret_energy.as<float>();
    A.r_good = c->r_good.as<unsigned char>(); A.r_lin = c->r_lin.as<unsigned char>(); A.r_sel = c->r_sel.as<unsigned char>();
    A.ctl = nullptr; A.it_index = 0; A.n_step_blocks = (c->P * 8 + 255) / 256; A.th_opt = 0; A.step_partial_ro = c->step_partial.as<float>();
    A.r_lin_rw = c->r_lin.as<unsigned char>(); A.point_tgt_rw = c->point_tgt.as<int>(); A.pt_mask = nullptr;
    A.r_center = c->r_center.as<float>(); A.r_jpjdf = c->r_jpjdf.as<float>(); A.r_rtz = c->r_rtz.as<float>();
    A.rj0 = c->rj[0].as<float>(); A.rj1 = c->rj[1].as<float>();
    A.by_point_off = c->by_point_off.as<int>(); A.by_point = c->by_point.as<int>();
    A.by_pair_off = c->by_pair_off.as<int>(); A.by_pair = c->by_pair.as<int>();
    A.point_code = c->point_code.as<int>(); A.point_tgt = c->point_tgt.as<int>(); A.point_pos = c->point_pos.as<int>(); A.pt_stride = c->pt_stride;
    A.r_idepth = c->r_idepth.as<double>(); A.point_res = c->point_res.as<int>();
    A.pair_code = c->pair_code.as<int>(); A.pair_pos = c->pair_pos.as<int>(); A.pair_stride = c->pair_stride;
    A.lin_partial = c->lin_partial.as<double>(); A.fuse_apply = 0; A.records_only = 0;
    { static const bool nla = getenv("CMLHIP_NO_LOOKAHEAD") != nullptr; A.no_lookahead = nla ? 1 : 0; }
    A.dbg = c->dbg_on ? c->dbg.as<long long>() : nullptr;
    return CMLHIP_OK;
}

static inline int ldg_of(int n) { return ((n + 1 + 15) / 16) * 16; }

// R-length readbacks: the device holds residuals in pair-sorted order r' (see cmlhip_ba_upload_window); the caller gets its own
// numbering back.  add() records a device array (element size in bytes): the permutation back to the caller's order runs ON THE DEVICE
// (k_res_to_caller: out[r] = dev[c_dev_of[r]] into a scratch block that the copy then takes as it is — a host-side gather of 15 000 entries per array
// was 10-15 us each); arrays that do not fit the scratch block fall back to the host permutation in deliver().
__global__ void k_res_to_caller(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, const int* __restrict__ dev_of, int R, int esz) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const size_t k = (size_t)dev_of[r];
    if ((esz & 3) == 0) {
        const unsigned* s4 = reinterpret_cast<const unsigned*>(src + k * esz); unsigned* d4 = reinterpret_cast<unsigned*>(dst + (size_t)r * esz);
        for (int i = 0; i < esz / 4; i++) d4[i] = s4[i];
    } else for (int i = 0; i < esz; i++) dst[(size_t)r * esz + i] = src[k * esz + i];
}
struct ResRead {
    cmlhip_ctx* c;
    struct Item { void* out; size_t esz; std::vector<unsigned char> tmp; const void* dev; };     // host-permuted fallback (dev: still to be copied)
    struct Direct { void* out; const void* src; size_t bytes; };                               // already in the caller's order in the scratch block
    std::vector<Item> items;
    std::vector<Direct> direct;
    size_t scratch_off = 0;
    explicit ResRead(cmlhip_ctx* c_) : c(c_) { items.reserve(8); }
    void add(void* out, const void* dev, size_t esz) {
        if (!out || c->R == 0) return;
        const size_t R = (size_t)c->R, bytes = esz * R, need = scratch_off + ((bytes + 255) & ~size_t(255));
        if (c->c_dev_of.p && c->c_dev_of.bytes >= 4 * R) {
            // (the scratch block is sized once, by the limit given at create: no re-allocation between the add()s of one readback)
            const size_t want = 64 * std::max(R, (size_t)c->lim.max_residuals);
            if (scratch_off == 0 && c->rr_scratch.bytes < want) (void)cml_ensure(c, c->rr_scratch, want);
            if (c->rr_scratch.bytes >= need) {
                unsigned char* dst = c->rr_scratch.as<unsigned char>() + scratch_off;
                k_res_to_caller<<<cml_div_up((int)R, 256), 256, 0, c->stream>>>(static_cast<const unsigned char*>(dev), dst, c->c_dev_of.as<int>(), (int)R, (int)esz);
                scratch_off = need;
                if (c->d2h_batching) cml_d2h(c, out, dst, bytes);            // recorded: delivered by the batch's flush, straight into `out`
                else direct.push_back(Direct{out, dst, bytes});
                return;
            }
        }
        items.push_back(Item{out, esz, std::vector<unsigned char>(bytes), dev});
        if (c->d2h_batching) { cml_d2h(c, items.back().tmp.data(), dev, bytes); items.back().dev = nullptr; }
    }
    int read_now() {
        for (auto& d : direct) { int rc = cml_d2h(c, d.out, d.src, d.bytes); if (rc) return rc; }
        direct.clear();
        for (auto& it : items)
            if (it.dev) { int rc = cml_d2h(c, it.tmp.data(), it.dev, it.esz * (size_t)c->R); if (rc) return rc; it.dev = nullptr; }
        deliver();
        return CMLHIP_OK;
    }
    void deliver() {
        const size_t R = (size_t)c->R;
        const int* dev_of = c->h_dev_of.data();
        for (auto& it : items) {
            if (it.esz == 4) {                                   // typed gathers: a memcpy call per element was most of a readback's host time
                uint32_t* o = static_cast<uint32_t*>(it.out); const uint32_t* t = reinterpret_cast<const uint32_t*>(it.tmp.data());
                for (size_t r = 0; r < R; r++) o[r] = t[dev_of[r]];
            } else if (it.esz == 1) {
                unsigned char* o = static_cast<unsigned char*>(it.out); const unsigned char* t = it.tmp.data();
                for (size_t r = 0; r < R; r++) o[r] = t[dev_of[r]];
            } else {
                unsigned char* o = static_cast<unsigned char*>(it.out);
                for (size_t r = 0; r < R; r++) memcpy(o + it.esz * r, it.tmp.data() + it.esz * (size_t)dev_of[r], it.esz);
            }
        }
        items.clear();
    }
};

struct ExpandArgs {
    int R, P, N, pt_stride, pair_stride;
    // caller order (the library's window shadows, copied as they are) + the two positions the host's counting pass assigned
    const int* c_point; const int* c_target; const int* c_state; const unsigned char* c_lin; const int* c_dev_of; const int* c_bpos;
    const int* pt_host; const float* pt_x; const float* pt_y; const float* pt_colors; const float* pt_weights;
    const int* by_point_off; const int* by_pair_off;
    // device order
    int* r_point; int* r_target; int* r_state; unsigned char* r_lin; int* r_host; int* r_new_state; int* by_pair; int* pair_pos; int* by_point;
    float* r_px; float* r_py; float* r_colors; float* r_weights;
    int* point_tgt; int* point_pos; int* point_res;
    const double* pt_idepth; double* r_idepth;
};
// one thread per residual of the CALLER's list: r' = c_dev_of[r] is its place in the pair-sorted device order, c_bpos[r] its place in the by-point
// lists; every R-length device array, the per-residual copies of the point's static inputs and the point-slot tables are written from here
__global__ void k_window_expand(ExpandArgs E) {
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    for (int r = i0; r < E.R; r += gridDim.x * blockDim.x) {
        const int k = E.c_dev_of[r], p = E.c_point[r], t = E.c_target[r], host = E.pt_host[p], q = host + t * E.N, lin = E.c_lin[r];
        E.r_point[k] = p; E.r_target[k] = t; E.r_state[k] = E.c_state[r]; E.r_lin[k] = (unsigned char)lin;
        E.r_host[k] = host;
        E.r_new_state[k] = CMLHIP_RES_OUTLIER;
        E.by_pair[k] = k;
        E.pair_pos[k] = q * E.pair_stride + (k - E.by_pair_off[q]);
        E.r_px[k] = E.pt_x[p]; E.r_py[k] = E.pt_y[p];
        E.r_idepth[k] = E.pt_idepth[p];                          // (the resident kernels address the inverse depth by the residual too: k_ba_idepth_to_res's work)
        const float4* c4 = reinterpret_cast<const float4*>(E.pt_colors + 8 * (size_t)p); const float4* w4 = reinterpret_cast<const float4*>(E.pt_weights + 8 * (size_t)p);
        float4* oc = reinterpret_cast<float4*>(E.r_colors + 8 * (size_t)k); float4* ow = reinterpret_cast<float4*>(E.r_weights + 8 * (size_t)k);
        oc[0] = c4[0]; oc[1] = c4[1]; ow[0] = w4[0]; ow[1] = w4[1];
        const int jb = E.c_bpos[r];                              // device lists hold r'; a point's residuals stay in the caller's list order (BA.cpp:1469-1479 walks them in that order)
        E.by_point[jb] = k;
        const int s_ = p * E.pt_stride + (jb - E.by_point_off[p]);   // the point's slot table: the residual in the slot (device id), its target | lin << 8
        E.point_res[s_] = k;
        E.point_tgt[s_] = t | (lin ? 256 : 0);
        E.point_pos[k] = s_;
    }
}

extern "C" {

int cmlhip_ba_set_params(cmlhip_ctx* c, const cmlhip_ba_params* prm) { CML_DEV_SCOPED(c);
    if (!c || !prm) return CMLHIP_ERR_INVALID;
    CML_REQUIRE(c, prm->w > 4 && prm->h > 4 && prm->fx != 0 && prm->fy != 0, CMLHIP_ERR_INVALID, "bad BA params");
    c->ba_prm = *prm;
    c->ba_prm_set = true;
    return CMLHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The window kept by the library across keyframes (cmlhip_ba_window_*): SoA shadows in the layout the device arrays have, edited in place by
// what BA::addPoints (BA.cpp:382-415), BA::addNewFrame (:417-462), removePoint / removeFrame (DSOContext.h:94-111,154-174) do to the
// reference's sets — append, retire, renumber — so that a keyframe's run() hands over a DELTA instead of rebuilding and re-copying the whole
// window.  cmlhip_ba_window_commit turns the shadows into the device window: index bookkeeping (exact: htIDX = host + target*N, BA.cpp:1677,
// CSR by point and by pair, pair-sorted device order) in two passes over the int shadows, the arrays memcpy'd into the packed copy.
// cmlhip_ba_upload_window is reset + append + commit: one code path, a window built by edits is the window a fresh upload builds.
static int window_commit(cmlhip_ctx* c, int N, const cmlhip_ba_frame* frames);

static int window_alloc(cmlhip_ctx* c) {
    WindowShadow& W = c->win;
    if (W.block) return CMLHIP_OK;
    const size_t cp = (size_t)std::max(c->lim.max_points, 1), cr = (size_t)std::max(c->lim.max_residuals, 1);
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t bytes = al(8 * cp) + 4 * al(4 * cp) + 2 * al(32 * cp) + al(4 * cp) + 3 * al(4 * cr) + al(cr);
    CML_CHECK(c, hipHostMalloc(&W.block, bytes, hipHostMallocMapped | hipHostMallocCoherent));
    CML_CHECK(c, hipEventCreateWithFlags(&W.busy, hipEventDisableTiming));
    char* q = static_cast<char*>(W.block);
    auto take = [&](size_t b) { char* r = q; q += al(b); return r; };
    W.idepth = (double*)take(8 * cp);
    W.x = (float*)take(4 * cp); W.y = (float*)take(4 * cp); W.idz = (float*)take(4 * cp); W.prior = (float*)take(4 * cp);
    W.colors = (float*)take(32 * cp); W.weights = (float*)take(32 * cp);
    W.host = (int*)take(4 * cp);
    W.rpoint = (int*)take(4 * cr); W.rtarget = (int*)take(4 * cr); W.rstate = (int*)take(4 * cr);
    W.rlin = (unsigned char*)take(cr);
    W.capP = cp; W.capR = cr; W.P = W.R = 0;
    return CMLHIP_OK;
}
// an edit of the shadows waits for the last commit's scatter kernel to have read them (normally long done: run() ends with a host wait)
static int window_writable(cmlhip_ctx* c) {
    int rc = window_alloc(c);
    if (rc) return rc;
    WindowShadow& W = c->win;
    if (W.busy_pending) {
        if (c->h2d_scope && (rc = cml_scope_end(c))) return rc;      // (the commit is still waiting in an open scope: it leaves first)
        CML_CHECK(c, hipEventSynchronize(W.busy));
        W.busy_pending = false;
    }
    return CMLHIP_OK;
}

int cmlhip_ba_window_reset(cmlhip_ctx* c) { CML_DEV_SCOPED(c);
    if (!c) return CMLHIP_ERR_INVALID;
    c->win.P = c->win.R = 0;
    c->win_generation += 1;                                  // whoever kept entries in the old window sees that it is gone (cmlhip_ba_window_generation)
    return CMLHIP_OK;
}
int cmlhip_ba_window_generation(cmlhip_ctx* c, unsigned* generation) {
    if (!c || !generation) return CMLHIP_ERR_INVALID;
    *generation = c->win_generation;
    return CMLHIP_OK;
}

int cmlhip_ba_window_append_points(cmlhip_ctx* c, int n, const cmlhip_ba_point* pts) { CML_DEV_SCOPED(c);
    if (!c || n < 0 || (n > 0 && !pts)) return CMLHIP_ERR_INVALID;
    int rc = window_writable(c);
    if (rc) return rc;
    WindowShadow& W = c->win;
    const size_t P0 = W.P;
    CML_REQUIRE(c, P0 + (size_t)n <= W.capP, CMLHIP_ERR_INVALID, "window exceeds the limits given at create");
    for (int i = 0; i < n; i++) {
        const cmlhip_ba_point& q = pts[i];
        const size_t p = P0 + i;
        W.x[p] = q.x; W.y[p] = q.y; W.idepth[p] = q.idepth; W.idz[p] = q.idepth_zero; W.prior[p] = q.prior; W.host[p] = q.host;
        memcpy(&W.colors[8 * p], q.colors, 32); memcpy(&W.weights[8 * p], q.weights, 32);
    }
    W.P = P0 + n;
    return CMLHIP_OK;
}

int cmlhip_ba_window_append_residuals(cmlhip_ctx* c, int n, const cmlhip_ba_residual* res) { CML_DEV_SCOPED(c);
    if (!c || n < 0 || (n > 0 && !res)) return CMLHIP_ERR_INVALID;
    int rc = window_writable(c);
    if (rc) return rc;
    WindowShadow& W = c->win;
    const size_t R0 = W.R, P = W.P;
    CML_REQUIRE(c, R0 + (size_t)n <= W.capR, CMLHIP_ERR_INVALID, "window exceeds the limits given at create");
    for (int i = 0; i < n; i++) CML_REQUIRE(c, res[i].point >= 0 && (size_t)res[i].point < P, CMLHIP_ERR_INVALID, "residual index out of range");
    for (int i = 0; i < n; i++) { W.rpoint[R0 + i] = res[i].point; W.rtarget[R0 + i] = res[i].target; W.rstate[R0 + i] = res[i].state; W.rlin[R0 + i] = res[i].is_linearized != 0; }
    W.R = R0 + n;
    return CMLHIP_OK;
}

// removeFrame's renumbering (DSOContext.h:154-174 + makeFrameId): ids above `frame` move down by one; whatever still names the frame itself gets -1
// (the caller drops those entries with the next cmlhip_ba_window_compact, as the reference's sets lose them)
int cmlhip_ba_window_retire_frame(cmlhip_ctx* c, int frame) { CML_DEV_SCOPED(c);
    if (!c || frame < 0) return CMLHIP_ERR_INVALID;
    int rc = window_writable(c);
    if (rc) return rc;
    WindowShadow& W = c->win;
    for (size_t p = 0; p < W.P; p++) { int& h = W.host[p]; if (h == frame) h = -1; else if (h > frame) h--; }
    for (size_t r = 0; r < W.R; r++) { int& t = W.rtarget[r]; if (t == frame) t = -1; else if (t > frame) t--; }
    return CMLHIP_OK;
}

// the survivors keep their order and are renumbered by rank (points AND residuals; a residual's point index follows); a residual that survives
// must name a surviving point
int cmlhip_ba_window_compact(cmlhip_ctx* c, int n_points, const unsigned char* point_alive, int n_res, const unsigned char* res_alive) { CML_DEV_SCOPED(c);
    if (!c || (n_points > 0 && !point_alive) || (n_res > 0 && !res_alive)) return CMLHIP_ERR_INVALID;
    int rc = window_writable(c);
    if (rc) return rc;
    WindowShadow& W = c->win;
    CML_REQUIRE(c, (size_t)n_points == W.P && (size_t)n_res == W.R, CMLHIP_ERR_STATE, "cmlhip_ba_window_compact: the caller's lists and the window differ in length");
    for (int r = 0; r < n_res; r++)                          // checked before anything moves: a refused call leaves the window as it was
        CML_REQUIRE(c, !res_alive[r] || point_alive[W.rpoint[r]], CMLHIP_ERR_INVALID, "cmlhip_ba_window_compact: a surviving residual names a dropped point");
    std::vector<int>& pmap = c->w_cnt_p;
    pmap.resize((size_t)n_points);
    size_t np = 0;
    for (int p = 0; p < n_points; p++) {
        pmap[p] = point_alive[p] ? (int)np : -1;
        if (!point_alive[p]) continue;
        if (np != (size_t)p) {
            W.x[np] = W.x[p]; W.y[np] = W.y[p]; W.idepth[np] = W.idepth[p]; W.idz[np] = W.idz[p]; W.prior[np] = W.prior[p]; W.host[np] = W.host[p];
            memcpy(&W.colors[8 * np], &W.colors[8 * (size_t)p], 32); memcpy(&W.weights[8 * np], &W.weights[8 * (size_t)p], 32);
        }
        np++;
    }
    W.P = np;
    size_t nr = 0;
    for (int r = 0; r < n_res; r++) {
        if (!res_alive[r]) continue;
        W.rpoint[nr] = pmap[W.rpoint[r]]; W.rtarget[nr] = W.rtarget[r]; W.rstate[nr] = W.rstate[r]; W.rlin[nr] = W.rlin[r];
        nr++;
    }
    W.R = nr;
    return CMLHIP_OK;
}

int cmlhip_ba_window_counts(cmlhip_ctx* c, int* P, int* R) { CML_DEV_SCOPED(c);
    if (!c) return CMLHIP_ERR_INVALID;
    if (P) *P = (int)c->win.P;
    if (R) *R = (int)c->win.R;
    return CMLHIP_OK;
}

int cmlhip_ba_window_commit(cmlhip_ctx* c, int N, const cmlhip_ba_frame* frames, const double* idepth, const float* idepth_zero, const float* prior,
                            int reset_states, int n_lin, const int* lin_residuals, const int* lin_states) { CML_DEV_SCOPED(c);
    if (!c || !frames || n_lin < 0 || (n_lin > 0 && (!lin_residuals || !lin_states))) return CMLHIP_ERR_INVALID;
    int rc = window_writable(c);
    if (rc) return rc;
    WindowShadow& W = c->win;
    const size_t P = W.P, R = W.R;
    if (idepth) memcpy(W.idepth, idepth, 8 * P);
    if (idepth_zero) memcpy(W.idz, idepth_zero, 4 * P);
    if (prior) memcpy(W.prior, prior, 4 * P);
    if (reset_states) {                                      // resetOOB of BA::run's preamble (BA.cpp:766-779) for everything but the listed LINEARIZED residuals
        for (size_t r = 0; r < R; r++) W.rstate[r] = (int)CMLHIP_RES_IN;
        memset(W.rlin, 0, R);
    }
    for (int i = 0; i < n_lin; i++) {
        CML_REQUIRE(c, lin_residuals[i] >= 0 && (size_t)lin_residuals[i] < R, CMLHIP_ERR_INVALID, "residual index out of range");
        W.rstate[lin_residuals[i]] = lin_states[i]; W.rlin[lin_residuals[i]] = 1;
    }
    rc = window_commit(c, N, frames);
    if (rc) { cml_scope_abort(c); c->ba_uploaded = false; }      // (a refused commit leaves NO window: not half of the new one, not the old one under new shadows)
    return rc;
}

int cmlhip_ba_upload_window(cmlhip_ctx* c, int N, const cmlhip_ba_frame* frames, int P, const cmlhip_ba_point* points,
                            int R, const cmlhip_ba_residual* res) { CML_DEV(c);
    if (!c || !frames || (P > 0 && !points) || (R > 0 && !res)) return CMLHIP_ERR_INVALID;
    CML_REQUIRE(c, c->ba_prm_set, CMLHIP_ERR_STATE, "cmlhip_ba_set_params not called");
    CML_REQUIRE(c, N >= 1 && N <= c->lim.max_frames && P >= 0 && P <= c->lim.max_points && R >= 0 && R <= c->lim.max_residuals,
                CMLHIP_ERR_INVALID, "window exceeds the limits given at create");
    int rc;
    if ((rc = window_writable(c))) return rc;
    CML_REQUIRE(c, (size_t)P <= c->win.capP && (size_t)R <= c->win.capR, CMLHIP_ERR_INVALID, "window exceeds the limits given at create");
    if ((rc = cmlhip_ba_window_reset(c))) return rc;
    if ((rc = cmlhip_ba_window_append_points(c, P, points))) return rc;
    if ((rc = cmlhip_ba_window_append_residuals(c, R, res))) return rc;
    rc = window_commit(c, N, frames);
    if (rc) c->ba_uploaded = false;
    return rc;
}

static int window_commit(cmlhip_ctx* c, int N, const cmlhip_ba_frame* frames) {
    WindowShadow& W = c->win;
    const int P = (int)W.P, R = (int)W.R;
    CML_REQUIRE(c, c->ba_prm_set, CMLHIP_ERR_STATE, "cmlhip_ba_set_params not called");
    CML_REQUIRE(c, N >= 1 && N <= c->lim.max_frames && P <= c->lim.max_points && R <= c->lim.max_residuals,
                CMLHIP_ERR_INVALID, "window exceeds the limits given at create");
    (void)hipSetDevice(c->device);
    const auto T0_ = std::chrono::steady_clock::now();
    static const bool timing_ = getenv("CMLHIP_TIMING") != nullptr;
    auto lap_ = [&](const char* w) { if (timing_ || getenv("CMLHIP_TIMING")) fprintf(stderr, "      [upload] %-18s %.0f us\n", w, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - T0_).count()); };
    // ---- frames: resolve pyramids
    std::vector<FrameDev> fd(N);
    for (int i = 0; i < N; i++) {
        const Pyramid* py = cml_find_pyr(c, frames[i].image_id);
        CML_REQUIRE(c, py && py->lv[0].grad, CMLHIP_ERR_NOT_FOUND, "frame image not in the pyramid cache");
        CML_REQUIRE(c, py->lv[0].w == c->ba_prm.w && py->lv[0].h == c->ba_prm.h, CMLHIP_ERR_INVALID, "image size != BA params");
        fd[i].grad0 = py->lv[0].grad; fd[i].frame_energy_th = frames[i].frame_energy_th; fd[i].b0 = frames[i].b0;
        fd[i].grad0t = nullptr;
        if (c->lim.texel_format == CMLHIP_TEXEL_F16) {              // the tiled copy the lane-per-residual kernel gathers from (built once per image)
            if (int rc_t = cml_tiled_level0(c, frames[i].image_id, &fd[i].grad0t)) return rc_t;
        }
    }
    // ---- index bookkeeping (exact): htIDX = host + target*N (BA.cpp:1677), CSR by point and by pair — pass 1: counts
    const int NN = N * N;
    c->h_pair_of.resize(R);
    c->h_by_point_off.assign(P + 1, 0); c->h_by_pair_off.assign(NN + 1, 0);
    int n_lin = 0, n_newframe = 0;
    {
        const int* rp = W.rpoint; const int* rt = W.rtarget; const int* hst = W.host;
        int* pair_of = c->h_pair_of.data(); int* cp = c->h_by_point_off.data() + 1; int* cq = c->h_by_pair_off.data() + 1;
        const unsigned char* rl = W.rlin;
        bool ok = true;
        for (int r = 0; r < R; r++) {
            const int p = rp[r], t = rt[r];
            const int host = hst[p];                         // (p was range-checked when the residual was appended; compaction keeps it valid)
            ok = ok && ((unsigned)t < (unsigned)N) && ((unsigned)host < (unsigned)N) && host != t;
            if (!ok) break;
            const int q = host + t * N;
            pair_of[r] = q; cp[p]++; cq[q]++;
            n_newframe += t == N - 1; n_lin += rl[r];
        }
        CML_REQUIRE(c, ok, CMLHIP_ERR_INVALID, "residual index out of range / bad host frame (BA.cpp:338-340)");
    }
    int mx_pt = 1, mx_pair = 1;
    for (int p = 0; p < P; p++) { mx_pt = std::max(mx_pt, c->h_by_point_off[p + 1]); c->h_by_point_off[p + 1] += c->h_by_point_off[p]; }
    for (int q = 0; q < NN; q++) { mx_pair = std::max(mx_pair, c->h_by_pair_off[q + 1]); c->h_by_pair_off[q + 1] += c->h_by_pair_off[q]; }
    c->N = N; c->P = P; c->R = R; c->n_lin = n_lin; c->n_newframe = n_newframe;
    const int n = 8 * N + 4, ldg = ldg_of(n), ntile = ldg / 16;
    // ---- allocations: by the LIMITS given at create (max_frames / max_points / max_residuals), not by this window's sizes — a window that grows keyframe by
    //      keyframe (2 -> 7 frames at the start of every sequence) otherwise re-allocates most of its ~90 buffers at every run (measured: 150-500 us of
    //      hipFree / hipMalloc per run() while the window grows, against 70-90 us for the whole commit at the sliding size).  Only the two tables whose
    //      stride is data-dependent (largest pair, fullest point) follow the window, with slack.
    int rc = 0;
    const size_t Nc = (size_t)std::max(N, c->lim.max_frames), Pc = (size_t)std::max(P, c->lim.max_points), Rc = (size_t)std::max(R, c->lim.max_residuals);
    const size_t NNc = Nc * Nc, nc = 8 * Nc + 4, ldgc = (size_t)ldg_of((int)nc), ntilec = ldgc / 16;
#define ENS(buf, bytes) if ((rc = cml_ensure(c, buf, (size_t)(bytes)))) return rc
    ENS(c->frames, sizeof(FrameDev) * Nc); ENS(c->pairs, sizeof(cmlhip_ba_pair) * NNc);
    ENS(c->pt_x, 4 * Pc); ENS(c->pt_y, 4 * Pc); ENS(c->pt_idepth, 8 * Pc); ENS(c->pt_idepth_zero, 4 * Pc); ENS(c->pt_prior, 4 * Pc);
    ENS(c->pt_host, 4 * Pc); ENS(c->pt_colors, 32 * Pc); ENS(c->pt_weights, 32 * Pc); ENS(c->pt_backup, 4 * Pc);
    ENS(c->pt_acc, 4 * PT_ACC_STRIDE * Pc); ENS(c->pt_step, 8 * Pc);
    ENS(c->r_point, 4 * Rc); ENS(c->r_host, 4 * Rc); ENS(c->r_target, 4 * Rc); ENS(c->r_state, 4 * Rc); ENS(c->r_new_state, 4 * Rc);
    ENS(c->r_energy, 4 * Rc); ENS(c->r_new_energy, 4 * Rc); ENS(c->r_new_energy_wo, 4 * Rc); ENS(c->r_ret_energy, 4 * Rc);
    ENS(c->r_good, Rc); ENS(c->r_lin, Rc); ENS(c->r_sel, Rc); ENS(c->r_dead, Rc); ENS(c->r_center, 12 * Rc); ENS(c->r_jpjdf, 4 * (size_t)PS_STRIDE * Rc); ENS(c->r_rtz, 32 * Rc);
    ENS(c->rj[0], 4 * (size_t)RJ_STRIDE * Rc); ENS(c->rj[1], 4 * (size_t)RJ_STRIDE * Rc);
    ENS(c->by_point_off, 4 * (Pc + 1)); ENS(c->by_point, 4 * Rc); ENS(c->by_pair_off, 4 * (NNc + 1)); ENS(c->by_pair, 4 * Rc);
    ENS(c->newframe_res, 4 * std::max((size_t)n_newframe, Pc));                 // (one residual per point into the newest frame at most)
    for (int m = 0; m < 2; m++) { ENS(c->acc_pair[m], 4 * ACC_STRIDE * NNc); ENS(c->acc_num[m], 4 * NNc); }
    ENS(c->pair_blocks, 2 * 8 * (size_t)PB_STRIDE * NNc);
    ENS(c->adH, 8 * 64 * NNc); ENS(c->adT, 8 * 64 * NNc); ENS(c->adHTd, 4 * 8 * NNc); ENS(c->vec_small, 8 * (8 + 16 * Nc));
    ENS(c->HA, 8 * nc * nc); ENS(c->HL, 8 * nc * nc); ENS(c->Hsc, 8 * nc * nc); ENS(c->HM, 8 * nc * nc);
    ENS(c->bA, 8 * nc); ENS(c->bL, 8 * nc); ENS(c->bsc, 8 * nc); ENS(c->bM, 8 * nc); ENS(c->xvec, 8 * nc);
    ENS(c->Hf, 8 * nc * nc); ENS(c->bf, 8 * nc);
    // ---- wave tiles of the resident residual kernel: <= RS_TILE consecutive device residuals of ONE pair each
    // Tile size by regime: a window that gives the lane-per-residual kernel at least one wave per SIMD runs it (throughput);
    // smaller windows are latency-bound and take 4 lanes per residual.  CMLHIP_RS_TILE=16|64 forces one (development).
    {
        const char* e = getenv("CMLHIP_RS_TILE");
        const int forced = e ? atoi(e) : 0;
        // (crossover measured with tools/probe_rs_regime.sh: R = 25 200: 11.2 (4 lanes) / 15.5 us (lane per residual); 35 000: 15.3 / 15.9; the
        //  lane-per-residual kernel stays at 15-16 us up to one wave per SIMD = 65 000 residuals, the 4-lane kernel starts a second round of
        //  waves at 2048 x 16 = 32 768)
        c->rs_tile = (forced == 16 || forced == 64) ? forced : (R >= 36 * 1024 ? 64 : 16);
    }
    const int TS = c->rs_tile;
    std::vector<int>& tiles = c->h_tiles; std::vector<int>& tile_off = c->h_tile_off;
    tiles.clear(); tile_off.assign(NN + 1, 0);
    c->h_rs_pair_tab.clear(); c->rs_pair_max_tiles = 0;
    for (int q = 0; q < NN; q++) {
        const int host = q % N, target = q / N;                  // htIDX = host + target * N
        for (int i = c->h_by_pair_off[q]; i < c->h_by_pair_off[q + 1]; i += TS) {
            tiles.push_back(i); tiles.push_back(std::min(TS, c->h_by_pair_off[q + 1] - i)); tiles.push_back(host); tiles.push_back(target);
        }
        tile_off[q + 1] = (int)tiles.size() / 4;
        if (tile_off[q + 1] > tile_off[q]) {                     // the pair table of the 2-D launch (ba_linearize_rs4.hip): travels in the kernel-argument segment
            c->h_rs_pair_tab.push_back(c->h_by_pair_off[q]); c->h_rs_pair_tab.push_back(c->h_by_pair_off[q + 1] - c->h_by_pair_off[q]);
            c->h_rs_pair_tab.push_back(tile_off[q]); c->h_rs_pair_tab.push_back(host | (target << 16));
            c->rs_pair_max_tiles = std::max(c->rs_pair_max_tiles, tile_off[q + 1] - tile_off[q]);
        }
    }
    c->rs_pair_n = (int)c->h_rs_pair_tab.size() / 4;
    c->n_tiles = (int)tiles.size() / 4;
    const size_t tiles_c = std::max((size_t)c->n_tiles, Rc / TS + NNc);          // (every pair may end in a partly filled tile)
    ENS(c->rs_tiles, 16 * std::max(tiles_c, (size_t)1)); ENS(c->rs_tile_off, 4 * (NNc + 1));
    ENS(c->rs_part, 1024 * std::max(tiles_c, (size_t)1));
    ENS(c->r_px, 4 * Rc); ENS(c->r_py, 4 * Rc); ENS(c->r_colors, 32 * Rc); ENS(c->r_weights, 32 * Rc); ENS(c->r_idepth, 8 * Rc);
    c->r_idepth_dirty = true;
    c->n_lin_partial = std::max((R + 31) / 32, c->n_tiles);
    c->lin_partial_n = 0; c->efs_in_partials = false;
    ENS(c->lin_partial, 32 * (std::max((Rc + 31) / 32, tiles_c) + 1)); ENS(c->step_partial, 16 * ((Pc + 31) / 32 + 1));
    ENS(c->G, 8 * (Pc * ldgc + Pc));
    ENS(c->syrk_part, 8 * 256 * (ntilec * (ntilec + 1) / 2) * (size_t)cml_sys_slices((int)Pc));
    ENS(c->xad, 8 * 8 * NNc);
    ENS(c->solve_image, 8 * ((ntilec * (ntilec + 1) / 2) * 16 * 17 + 32 * ntilec));      // k_ba_assemble (wide windows)
    ENS(c->scal, 1024);
    c->pair_stride = (mx_pair + 3) & ~3;
    c->pt_stride = (mx_pt + 7) & ~7;
    const size_t pt_tot = (size_t)std::max(P, 1) * c->pt_stride;
    const size_t pt_tot_c = Pc * (size_t)std::max(c->pt_stride, (int)(((Nc - 1) + 7) & ~size_t(7)));      // a point has at most one residual per other frame
    const size_t pair_need = (size_t)NN * c->pair_stride;
    if (c->pair_code.bytes < 4 * pair_need) ENS(c->pair_code, 4 * (pair_need + pair_need / 2));          // data-dependent stride: follows the window, with slack
    ENS(c->pair_pos, 4 * std::max(Rc, (size_t)1));
    ENS(c->point_code, 4 * std::max(pt_tot_c, pt_tot)); ENS(c->point_tgt, 4 * std::max(pt_tot_c, pt_tot)); ENS(c->point_pos, 4 * std::max(Rc, (size_t)1)); ENS(c->point_res, 4 * std::max(pt_tot_c, pt_tot));
#undef ENS
    lap_("counts+ensure");
    // ---- SoA staging + upload: everything below is staged and leaves in ONE copy + one scatter / fill kernel (cml_h2d_batch_flush)
    cml_h2d_batch_begin(c);
    struct BatchGuard { cmlhip_ctx* c; ~BatchGuard() { if (c->h2d_batching && !c->h2d_scope) { c->h2d_batching = false; c->h2d_segs.clear(); } } } batch_guard{c};   // error returns close the batch
    // the device-order arrays are written where the packed copy starts from (cml_h2d_stage); the vectors exist only when the ring has no room
    struct Staged { void* p = nullptr; std::vector<unsigned char> fb; void* dst = nullptr; size_t bytes = 0; };
    auto stage = [&](Staged& s, DevBuf& buf, size_t bytes) -> void* {
        s.dst = buf.p; s.bytes = bytes;
        s.p = cml_h2d_stage(c, buf.p, bytes);
        if (!s.p) { s.fb.resize(bytes ? bytes : 1); s.p = s.fb.data(); }
        return s.p;
    };
    auto commit = [&](Staged& s) -> int { return s.fb.empty() ? CMLHIP_OK : cml_h2d(c, s.dst, s.fb.data(), s.bytes); };
    // points: the shadows ARE the device layout (caller order), and they sit in pinned, device-mapped memory: the scatter kernel reads them in place
#define UPS(buf, ptr, bytes) if ((rc = cml_h2d_inplace(c, (buf).p, (ptr), (bytes)))) return rc
    UPS(c->pt_x, W.x, 4 * (size_t)P); UPS(c->pt_y, W.y, 4 * (size_t)P); UPS(c->pt_idepth, W.idepth, 8 * (size_t)P); UPS(c->pt_idepth_zero, W.idz, 4 * (size_t)P);
    UPS(c->pt_prior, W.prior, 4 * (size_t)P); UPS(c->pt_host, W.host, 4 * (size_t)P); UPS(c->pt_colors, W.colors, 32 * (size_t)P); UPS(c->pt_weights, W.weights, 32 * (size_t)P);
    // ---- pass 2: DEVICE residual order = (host,target)-pair-sorted: r' = position in the by-pair list.  Every R-length device array, the
    //      by-point lists and the new-frame list hold r'; the ABI keeps the caller's numbering (inputs are permuted here, readbacks
    //      are permuted back, cmlhip_ba_get_index_maps reports the caller-order maps).  Residuals of one pair are contiguous on the
    //      device, so the residual kernel of the resident loop reads the pair record through scalar loads.
    //      The host only ASSIGNS the two positions of every residual (sequential writes); the permutation itself runs on the device (k_window_expand).
    c->h_dev_of.resize(R); c->h_maps_valid = false;
    if ((rc = cml_ensure(c, c->c_point, 4 * Rc))) return rc;
    if ((rc = cml_ensure(c, c->c_target, 4 * Rc))) return rc;
    if ((rc = cml_ensure(c, c->c_state, 4 * Rc))) return rc;
    if ((rc = cml_ensure(c, c->c_lin, Rc))) return rc;
    if ((rc = cml_ensure(c, c->c_dev_of, 4 * Rc))) return rc;
    if ((rc = cml_ensure(c, c->c_bpos, 4 * Rc))) return rc;
    {
        Staged s_dv, s_bp, s_nf;
        int* dvd = (int*)stage(s_dv, c->c_dev_of, 4 * (size_t)R); int* bpd = (int*)stage(s_bp, c->c_bpos, 4 * (size_t)R);
        int* nfd = (int*)stage(s_nf, c->newframe_res, 4 * (size_t)n_newframe);
        c->w_cnt_p.assign(P, 0); c->w_cnt_q.assign(NN, 0);
        int* c1 = c->w_cnt_p.data(); int* c2 = c->w_cnt_q.data();
        const int* rp = W.rpoint; const int* rt = W.rtarget;
        const int* pair_of = c->h_pair_of.data(); const int* opt = c->h_by_point_off.data(); const int* oq = c->h_by_pair_off.data();
        int* dev_of = c->h_dev_of.data();
        int nf = 0;
        for (int r = 0; r < R; r++) {
            const int p = rp[r], q = pair_of[r];
            const int k = oq[q] + c2[q]++;                   // device id
            dev_of[r] = k; dvd[r] = k;
            bpd[r] = opt[p] + c1[p]++;
            if (rt[r] == N - 1) nfd[nf++] = k;
        }
        for (Staged* s : {&s_dv, &s_bp, &s_nf}) if ((rc = commit(*s))) return rc;
    }
    UPS(c->c_point, W.rpoint, 4 * (size_t)R); UPS(c->c_target, W.rtarget, 4 * (size_t)R); UPS(c->c_state, W.rstate, 4 * (size_t)R); UPS(c->c_lin, W.rlin, (size_t)R);
    W.busy_pending = true;                                   // (recorded behind the scatter kernel when the batch leaves: cml_h2d_batch_flush)
#define UP(buf, vec) if ((rc = cml_h2d(c, (buf).p, (vec).data(), (vec).size() * sizeof((vec)[0])))) return rc
    UP(c->frames, fd);
    UP(c->by_point_off, c->h_by_point_off);
    UP(c->by_pair_off, c->h_by_pair_off);
    if ((rc = cml_fill_ff(c, c->pair_code.p, 4 * (size_t)NN * c->pair_stride))) return rc;      // nothing is good before the first applyRes
    if ((rc = cml_fill_ff(c, c->point_code.p, 4 * pt_tot))) return rc;
    if ((rc = cml_fill_ff(c, c->point_tgt.p, 4 * pt_tot))) return rc;          // empty slots: -1 (filled by k_window_expand where a residual sits)
    if ((rc = cml_fill_ff(c, c->point_res.p, 4 * pt_tot))) return rc;
    if (c->n_tiles) { UP(c->rs_tiles, tiles); }
    UP(c->rs_tile_off, tile_off);
#undef UP
#undef UPS
    // resetOOB (DSOResidual.h:83-88): energies 0, flags cleared — only the R (P) entries of this window, not the buffers' high-water capacity
    //  (+ 64 entries of slack: a tile's tail lanes may look past R)
    auto zr = [&](DevBuf& b, size_t elem, size_t count) -> int { return cml_zero(c, b.p, std::min(b.bytes, elem * (count + 64))); };
    const size_t Rz = (size_t)R, Pz = (size_t)P;
    if ((rc = zr(c->r_energy, 4, Rz))) return rc;
    if ((rc = zr(c->r_new_energy, 4, Rz))) return rc;
    if ((rc = zr(c->r_new_energy_wo, 4, Rz))) return rc;
    if ((rc = zr(c->r_ret_energy, 4, Rz))) return rc;
    if ((rc = zr(c->r_good, 1, Rz))) return rc;
    if ((rc = zr(c->r_sel, 1, Rz))) return rc;
    if ((rc = zr(c->r_dead, 1, Rz))) return rc;
    if ((rc = zr(c->r_center, 12, Rz))) return rc;
    if ((rc = zr(c->r_jpjdf, 4 * (size_t)PS_STRIDE, Rz))) return rc;
    if ((rc = zr(c->r_rtz, 32, Rz))) return rc;
    if ((rc = zr(c->rj[0], 4 * (size_t)RJ_STRIDE, Rz))) return rc;
    if ((rc = zr(c->rj[1], 4 * (size_t)RJ_STRIDE, Rz))) return rc;
    if ((rc = zr(c->pt_acc, 4 * (size_t)PT_ACC_STRIDE, Pz))) return rc;
    if ((rc = zr(c->pt_step, 8, Pz))) return rc;
    if ((rc = zr(c->pt_backup, 4, Pz))) return rc;
    if ((rc = cml_zero(c, c->scal.p, c->scal.bytes))) return rc;
    if ((rc = cml_zero(c, c->pair_blocks.p, 2 * 8 * (size_t)PB_STRIDE * NN))) return rc;
    if ((rc = cml_zero(c, c->lin_partial.p, 32 * (size_t)(c->n_lin_partial + 1)))) return rc;
    if ((rc = cml_zero(c, c->step_partial.p, 16 * (size_t)((P + 31) / 32 + 1)))) return rc;
    lap_("staged");
    if ((rc = cml_h2d_batch_flush(c))) return rc;
    if (R > 0) {
        ExpandArgs E;
        E.R = R; E.P = P; E.N = N; E.pt_stride = c->pt_stride; E.pair_stride = c->pair_stride;
        E.c_point = c->c_point.as<int>(); E.c_target = c->c_target.as<int>(); E.c_state = c->c_state.as<int>(); E.c_lin = c->c_lin.as<unsigned char>();
        E.c_dev_of = c->c_dev_of.as<int>(); E.c_bpos = c->c_bpos.as<int>();
        E.pt_host = c->pt_host.as<int>(); E.pt_x = c->pt_x.as<float>(); E.pt_y = c->pt_y.as<float>(); E.pt_colors = c->pt_colors.as<float>(); E.pt_weights = c->pt_weights.as<float>();
        E.by_point_off = c->by_point_off.as<int>(); E.by_pair_off = c->by_pair_off.as<int>();
        E.r_point = c->r_point.as<int>(); E.r_target = c->r_target.as<int>(); E.r_state = c->r_state.as<int>(); E.r_lin = c->r_lin.as<unsigned char>();
        E.r_host = c->r_host.as<int>(); E.r_new_state = c->r_new_state.as<int>(); E.by_pair = c->by_pair.as<int>(); E.pair_pos = c->pair_pos.as<int>(); E.by_point = c->by_point.as<int>();
        E.r_px = c->r_px.as<float>(); E.r_py = c->r_py.as<float>(); E.r_colors = c->r_colors.as<float>(); E.r_weights = c->r_weights.as<float>();
        E.point_tgt = c->point_tgt.as<int>(); E.point_pos = c->point_pos.as<int>(); E.point_res = c->point_res.as<int>();
        E.pt_idepth = c->pt_idepth.as<double>(); E.r_idepth = c->r_idepth.as<double>();
        c->r_idepth_dirty = false;                               // (written by the expand kernel, which runs ahead of anything that reads it)
        const int work = R;
        if ((rc = cml_defer(c, [c, E, work]() -> int {
                k_window_expand<<<cml_div_up(std::min(work, 1 << 20), 256), 256, 0, c->stream>>>(E);
                CML_CHECK(c, hipGetLastError());
                return CMLHIP_OK; }))) return rc;
    }
    lap_("flushed");
    c->ba_uploaded = true;
    c->ba_image_ids.clear();
    for (int i = 0; i < N; i++) c->ba_image_ids.push_back(frames[i].image_id);
    c->ba_pairs_set = false;
    c->resident_on = false; c->resident_iter = 0;
    return CMLHIP_OK;
}

int cmlhip_ba_window_size(cmlhip_ctx* c, int* N, int* P, int* R) { CML_DEV(c);
    if (!c) return CMLHIP_ERR_INVALID;
    if (N) *N = c->ba_uploaded ? c->N : 0;
    if (P) *P = c->ba_uploaded ? c->P : 0;
    if (R) *R = c->ba_uploaded ? c->R : 0;
    return CMLHIP_OK;
}

int cmlhip_ba_set_pairs(cmlhip_ctx* c, const cmlhip_ba_pair* pairs) { CML_DEV_SCOPED(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    if (c->h2d_scope && c->efs_in_partials && (rc = cml_scope_end(c))) return rc;      // (records to re-create: a launch)
    if ((rc = cml_materialize_records(c))) return rc;        // the resident kernel keeps Jacobians in reduced form: re-create the records first
    if (!pairs) return CMLHIP_ERR_INVALID;
    rc = cml_h2d(c, c->pairs.p, pairs, sizeof(cmlhip_ba_pair) * c->N * c->N);
    if (rc) return rc;
    c->ba_pairs_set = true;
    return CMLHIP_OK;
}

int cmlhip_ba_set_frame_energy_th(cmlhip_ctx* c, const float* th) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    if (!th) return CMLHIP_ERR_INVALID;
    std::vector<FrameDev> fd(c->N);
    rc = cml_d2h(c, fd.data(), c->frames.p, sizeof(FrameDev) * c->N);
    if (rc) return rc;
    for (int i = 0; i < c->N; i++) fd[i].frame_energy_th = th[i];
    return cml_h2d(c, c->frames.p, fd.data(), sizeof(FrameDev) * c->N);
}

int cmlhip_ba_set_arithmetic(cmlhip_ctx* c, int mode) { CML_DEV_SCOPED(c);
    if (!c || (mode != CMLHIP_ARITH_EXACT && mode != CMLHIP_ARITH_RELAXED)) return CMLHIP_ERR_INVALID;
    c->arith_relaxed = mode == CMLHIP_ARITH_RELAXED;
    return CMLHIP_OK;
}

int cmlhip_ba_set_resident_outputs(cmlhip_ctx* c, int mode) { CML_DEV_SCOPED(c);
    if (!c || (mode != CMLHIP_RESIDENT_OUTPUTS_FULL && mode != CMLHIP_RESIDENT_OUTPUTS_LEAN)) return CMLHIP_ERR_INVALID;
    c->rs_lean = mode == CMLHIP_RESIDENT_OUTPUTS_LEAN;
    return CMLHIP_OK;
}
int cmlhip_ba_get_resident_outputs(cmlhip_ctx* c, int* mode) {
    if (!c || !mode) return CMLHIP_ERR_INVALID;
    *mode = c->rs_lean ? CMLHIP_RESIDENT_OUTPUTS_LEAN : CMLHIP_RESIDENT_OUTPUTS_FULL;
    return CMLHIP_OK;
}

// DSOFrame::getB0 follows state_zero (DSOFrame.h:197-199): run()'s epilogue re-anchors the newest frame (setEvalPT, BA.cpp:885-894), so
// the b0 uploaded with the window is stale for residuals HOSTED by that frame from then on (the closing linearizeAll(true), tryMarginalize)
struct B0Args { float b0[CMLHIP_MAX_FRAMES]; };
__global__ void k_set_frame_b0(FrameDev* fd, int N, B0Args a) { const int i = threadIdx.x; if (i < N) fd[i].b0 = a.b0[i]; }
int cmlhip_ba_set_frame_b0(cmlhip_ctx* c, const float* b0) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    if (!b0) return CMLHIP_ERR_INVALID;
    B0Args a;
    for (int i = 0; i < CMLHIP_MAX_FRAMES; i++) a.b0[i] = i < c->N ? b0[i] : 0.f;
    k_set_frame_b0<<<1, 64, 0, c->stream>>>(c->frames.as<FrameDev>(), c->N, a);
    CML_CHECK(c, hipGetLastError());
    return CMLHIP_OK;
}

int cmlhip_ba_set_idepth(cmlhip_ctx* c, const double* idepth, const float* idepth_zero) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    if ((rc = cml_materialize_records(c))) return rc;        // the resident kernel keeps Jacobians in reduced form: re-create the records first
    if (!idepth) return CMLHIP_ERR_INVALID;
    c->r_idepth_dirty = true;
    rc = cml_h2d(c, c->pt_idepth.p, idepth, 8 * (size_t)c->P);
    if (!rc && idepth_zero) rc = cml_h2d(c, c->pt_idepth_zero.p, idepth_zero, 4 * (size_t)c->P);
    return rc;
}
int cmlhip_ba_get_idepth(cmlhip_ctx* c, double* idepth) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    return cml_d2h(c, idepth, c->pt_idepth.p, 8 * (size_t)c->P);
}

int cmlhip_ba_linearize_async(cmlhip_ctx* c) { CML_DEV(c);
    int rc = ba_check(c, true);
    if (rc) return rc;
    if ((rc = cml_materialize_records(c))) return rc;        // the resident kernel keeps Jacobians in reduced form: re-create the records first
    BAArgs A;
    cml_make_ba_args(c, A);
    cml_launch_linearize(c, A);
    cml_launch_lin_finish(c, A);
    CML_CHECK(c, hipGetLastError());
    return CMLHIP_OK;
}

int cmlhip_ba_linearize(cmlhip_ctx* c, cmlhip_ba_lin_result* out) { CML_DEV(c);
    int rc = cmlhip_ba_linearize_async(c);
    if (rc) return rc;
    LinSummary S;
    rc = cml_d2h(c, &S, c->scal.p, sizeof S);
    if (rc) return rc;
    if (out) {
        out->energy = S.energy; out->n_in = S.n_in; out->n_oob = S.n_oob; out->n_outlier = S.n_outlier;
        out->new_frame_energy_th = S.new_frame_energy_th;
    }
    return std::isfinite(S.energy) ? CMLHIP_OK : CMLHIP_ERR_NONFINITE;
}

// linearizeAll(false) + applyRes(r, true) of BA::run's preamble (BA.cpp:785-790: nothing sits between them) as ONE pass over the residuals
int cmlhip_ba_linearize_apply(cmlhip_ctx* c, cmlhip_ba_lin_result* out) { CML_DEV(c);
    int rc = ba_check(c, true);
    if (rc) return rc;
    if ((rc = cml_materialize_records(c))) return rc;
    BAArgs A;
    cml_make_ba_args(c, A);
    A.fuse_apply = 1;
    if (c->rs_ok && c->n_tiles > 0 && c->n_lin == 0) {      // the resident loop's own residual kernel: the pass leaves the pair tiles its first accumulation reads
        const bool relaxed = c->arith_relaxed;              // include/cmlhip.h: only cmlhip_ba_iteration_async / _batch honour CMLHIP_ARITH_RELAXED; this pass
        c->arith_relaxed = false;                           // commits states and frameEnergyTH through applyRes(true), so it is always the exact kernel
        cml_launch_linearize_rs(c, A);                      // (28 us of record-path accumulation become 10 at the sequence's window; records re-materialise on demand)
        c->arith_relaxed = relaxed;
        c->efs_in_partials = true; c->lin_partial_n = c->n_tiles;
    } else cml_launch_linearize(c, A);
    if (!out) cml_mark(c, "pass0");
    if (!out) {                                             // enqueue only: the pass's tail (energy sum, setNewFrameEnergyTH) stays pending — it rides in the next
        c->lin_finish_pending = true;                       // iteration's solve launch like every later pass's (its energy is logged as ResidentCtl::energy0 when
        CML_CHECK(c, hipGetLastError());                    // cmlhip_ba_resident_convergence armed the log), or runs when a getter / cmlhip_ba_finish_run asks
        return CMLHIP_OK;
    }
    cml_launch_lin_finish(c, A);
    CML_CHECK(c, hipGetLastError());
    LinSummary S;
    if ((rc = cml_d2h(c, &S, c->scal.p, sizeof S))) return rc;
    out->energy = S.energy; out->n_in = S.n_in; out->n_oob = S.n_oob; out->n_outlier = S.n_outlier; out->new_frame_energy_th = S.new_frame_energy_th;
    return std::isfinite(S.energy) ? CMLHIP_OK : CMLHIP_ERR_NONFINITE;
}

int cmlhip_ba_apply(cmlhip_ctx* c, int copy) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    if ((rc = cml_materialize_records(c))) return rc;        // the resident kernel keeps Jacobians in reduced form: re-create the records first
    BAArgs A;
    cml_make_ba_args(c, A);
    cml_launch_apply(c, A, copy);
    CML_CHECK(c, hipGetLastError());
    return CMLHIP_OK;
}

__global__ void k_ba_retire(BAArgs A) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.R || A.r_lin[r] || A.r_good[r]) return;
    A.r_dead[r] = 1;
    A.r_state[r] = CMLHIP_RES_OOB;
}
int cmlhip_ba_finish_keyframe(cmlhip_ctx* c, cmlhip_ba_lin_result* lin, int* state, int* new_state, float* energy, float* new_energy,
                              float* new_energy_wo, unsigned char* is_good, double* idepth, float* point_acc) { CML_DEV(c);
    int rc = ba_check(c, true);
    if (rc) return rc;
    if ((rc = cml_materialize_records(c))) return rc;        // (pair codes and selectors of the residuals this pass does not evaluate: state OOB)
    BAArgs A;
    cml_make_ba_args(c, A);
    A.fuse_apply = 1;                                        // linearize + applyRes(r, true) in ONE pass over the residuals (BA.cpp:1568-1569: nothing sits between them)
    cml_launch_linearize(c, A);
    cml_launch_lin_finish(c, A);
    CML_CHECK(c, hipGetLastError());
    const size_t R = c->R, P = c->P;
    LinSummary S;
    std::vector<float> pacc(point_acc ? PT_ACC_STRIDE * P : 0);
    ResRead rr(c);
    cml_d2h_batch_begin(c);
    cml_d2h(c, &S, c->scal.p, sizeof S);
    rr.add(state, c->r_state.p, 4); rr.add(new_state, c->r_new_state.p, 4); rr.add(energy, c->r_energy.p, 4);
    rr.add(new_energy, c->r_new_energy.p, 4); rr.add(new_energy_wo, c->r_new_energy_wo.p, 4); rr.add(is_good, c->r_good.p, 1);
    if (idepth) cml_d2h(c, idepth, c->pt_idepth.p, 8 * P);
    if (point_acc && P) cml_d2h(c, pacc.data(), c->pt_acc.p, 4 * pacc.size());
    if ((rc = cml_d2h_batch_flush(c))) return rc;
    rr.deliver();
    if (point_acc) for (size_t p = 0; p < P; p++) memcpy(point_acc + 14 * p, &pacc[PT_ACC_STRIDE * p], 14 * 4);
    if (lin) { lin->energy = S.energy; lin->n_in = S.n_in; lin->n_oob = S.n_oob; lin->n_outlier = S.n_outlier; lin->new_frame_energy_th = S.new_frame_energy_th; }
    // linearizeAll(true) REMOVES every active residual that is not good (toRemove, BA.cpp:1595-1598,1624-1638): from here on it is gone for
    // the window on the device as well — state OOB (absorbing: linearize and applyRes return at once, BA.cpp:68-72,2055-2059), never reset by
    // the residual loop of tryMarginalize (the reference walks the point's remaining residuals only, BA.cpp:2291)
    if (R) k_ba_retire<<<cml_div_up((int)R, 256), 256, 0, c->stream>>>(A);
    CML_CHECK(c, hipGetLastError());
    return std::isfinite(S.energy) ? CMLHIP_OK : CMLHIP_ERR_NONFINITE;
}

// BA::run's re-anchoring of the newest frame (BA.cpp:885-894: setEvalPT(PRE_worldToCam, (0,...,0,a,b))) on the device, between the last iteration
// and the closing linearizeAll(true): the frame's evaluation point becomes its current pose, its state keeps the affine entries only, the
// DSOFramePrecomputed of every pair that names it gets its PRE_RTll_0 / PRE_tTll_0 from the new evaluation points (DSOFrame.h:261-267) and its b0
// follows state_zero (DSOFrame.h:197-199).  Saves run() the host round trip (frame states back, pairs + b0 down) ahead of the closing pass.
// The host mirror adopts the same evaluation point from the pose read back (pre_w2c).
__device__ __forceinline__ void reanchor_newest_block(cmlhip_ba_frame_state* __restrict__ fs, const double* __restrict__ pre_w2c, cmlhip_ba_pair* __restrict__ pairs,
                                                      FrameDev* __restrict__ frames, int N, double scale_b, cmlhip_ba_frame_state* __restrict__ snap, double (*s_ev)[7]) {
    using cml_amd::SE3;
    const int tid = threadIdx.x, f = N - 1;
    if (tid < N) {
        cmlhip_ba_frame_state& S = fs[tid];
        snap[tid] = S;                                       // the loop's result, as the host reads it back (the edit below is the run's epilogue)
        if (tid == f) {
            for (int k = 0; k < 4; k++) { S.eval_q[k] = pre_w2c[7 * f + k]; s_ev[f][k] = S.eval_q[k]; }
            for (int k = 0; k < 3; k++) { S.eval_t[k] = pre_w2c[7 * f + 4 + k]; s_ev[f][4 + k] = S.eval_t[k]; }
            for (int k = 0; k < 10; k++) { const double v = (k == 6 || k == 7) ? S.state[k] : 0.0; S.state[k] = v; S.state_zero[k] = v; }
            frames[f].b0 = (float)(S.state_zero[7] * scale_b);
        } else {
            for (int k = 0; k < 4; k++) s_ev[tid][k] = S.eval_q[k];
            for (int k = 0; k < 3; k++) s_ev[tid][4 + k] = S.eval_t[k];
        }
    }
    __syncthreads();
    for (int e = tid; e < 2 * N; e += blockDim.x) {          // pairs (h, f) and (f, t)
        const int h = e < N ? e : f, t = e < N ? f : e - N;
        if (e >= N && t == f) continue;                      // (f, f) once
        SE3 Et, Eh;
        for (int k = 0; k < 4; k++) { Et.q[k] = s_ev[t][k]; Eh.q[k] = s_ev[h][k]; }
        for (int k = 0; k < 3; k++) { Et.t[k] = s_ev[t][4 + k]; Eh.t[k] = s_ev[h][4 + k]; }
        const SE3 l0 = Et * Eh.inverse();
        cmlhip_ba_pair& P = pairs[(size_t)h * N + t];
        double R0[9];
        l0.matrix(R0);
        for (int k = 0; k < 9; k++) P.R0[k] = R0[k];
        for (int k = 0; k < 3; k++) P.t0[k] = l0.t[k];
    }
}
__global__ void k_ba_reanchor_newest(cmlhip_ba_frame_state* __restrict__ fs, const double* __restrict__ pre_w2c, cmlhip_ba_pair* __restrict__ pairs,
                                     FrameDev* __restrict__ frames, int N, double scale_b, cmlhip_ba_frame_state* __restrict__ snap) {
    __shared__ double s_ev[CMLHIP_MAX_FRAMES][7];
    reanchor_newest_block(fs, pre_w2c, pairs, frames, N, scale_b, snap, s_ev);
}
// the tail of the loop's last residual pass (workgroup 0: energy sum, census, setNewFrameEnergyTH) and the re-anchoring (workgroup 1) in one launch:
// they touch different records (frames[N-1].frame_energy_th | frames[N-1].b0, the pair records, the frame states)
__global__ __launch_bounds__(1024) void k_ba_finish_reanchor(BAArgs A, const int* __restrict__ newframe_res, int n_newframe, const double* __restrict__ lin_partial, int n_partial,
                                                             LinSummary* __restrict__ out, FrameDev* __restrict__ frames_rw, cmlhip_ba_frame_state* __restrict__ fs,
                                                             const double* __restrict__ pre_w2c, cmlhip_ba_pair* __restrict__ pairs, double scale_b, cmlhip_ba_frame_state* __restrict__ snap) {
    __shared__ unsigned s_u32[264];
    __shared__ double s_f64[1024];
    if (blockIdx.x == 0) lin_finish_block(A, newframe_res, n_newframe, lin_partial, n_partial, out, frames_rw, s_u32, s_f64);
    else reanchor_newest_block(fs, pre_w2c, pairs, frames_rw, A.N, scale_b, snap, reinterpret_cast<double (*)[7]>(s_f64));
}

// what the host's bookkeeping behind the closing pass reads, packed in the CALLER's order: one byte per residual, one float per point
__global__ void k_ba_pack_closing(int R, int P, const int* __restrict__ c_dev_of, const int* __restrict__ r_state, const unsigned char* __restrict__ r_good,
                                  const float* __restrict__ pt_acc, unsigned char* __restrict__ sg, float* __restrict__ hdi) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R) { const int k = c_dev_of[i]; sg[i] = (unsigned char)((r_state[k] & 3) | (r_good[k] ? 4 : 0)); }
    if (i < P) hdi[i] = pt_acc[(size_t)PT_ACC_STRIDE * i + 12];
}

// The tail of DSOBundleAdjustment::run with the loop resident on the device, in ONE readback: what cmlhip_ba_get_resident_state / _log / _indirect return
// after the iterations (frame states, PRE_worldToCam, the preamble pass's and the last pass's summaries, the energy log, x), then — reanchor_newest —
// the newest frame re-anchored on the device (k_ba_reanchor_newest) and cmlhip_ba_finish_keyframe's closing pass with its outputs.
int cmlhip_ba_finish_run(cmlhip_ctx* c, int reanchor_newest, const cmlhip_ba_resident_out* ro, cmlhip_ba_lin_result* lin, int* state, int* new_state,
                         float* energy, float* new_energy, float* new_energy_wo, unsigned char* is_good, double* idepth, float* point_acc) { CML_DEV(c);
    int rc = ba_check(c, true);
    if (rc) return rc;
    CML_REQUIRE(c, c->resident_on, CMLHIP_ERR_STATE, "cmlhip_ba_set_resident_state not called for this window");
    CML_REQUIRE(c, !reanchor_newest || c->resident_iter > 0, CMLHIP_ERR_STATE, "cmlhip_ba_finish_run: no iteration has run (PRE_worldToCam is not on the device)");
    BAArgs A;
    cml_make_ba_args(c, A);
    const size_t N = c->N, n = 8 * N + 4, R = c->R, P = c->P;
    if ((rc = cml_ensure(c, c->run_snap, sizeof(cmlhip_ba_frame_state) * N))) return rc;
    // the closing pass as the loop's own residual kernel (Jacobians in reduced form: whoever reads a 74-float record afterwards — the marginalisation
    // passes — re-creates them on demand, cml_materialize_records); windows the resident kernel does not take go through the record-writing kernel
    const bool rs_close = c->rs_ok && c->n_tiles > 0 && c->n_lin == 0;
    if (!rs_close && (rc = cml_materialize_records(c))) return rc;      // (with the loop's own pair records, as cmlhip_ba_set_pairs does ahead of a new set)
    // (the loop's last summary stays at offset 0 of the scalar scratch: the closing pass writes its own slot; the frame states of the loop are kept by
    //  the re-anchoring before it edits them)
    const bool pend = c->lin_finish_pending;
    if (pend && c->conv_on) A.ctl = reinterpret_cast<ResidentCtl*>(c->scal.as<char>() + CML_CTL_OFFSET);
    if (pend && reanchor_newest) {
        k_ba_finish_reanchor<<<2, 1024, 0, c->stream>>>(A, c->newframe_res.as<int>(), c->n_newframe, c->lin_partial.as<double>(), c->lin_partial_n, c->scal.as<LinSummary>(),
                                                          c->frames.as<FrameDev>(), c->frame_state.as<cmlhip_ba_frame_state>(), c->pre_w2c.as<double>(), c->pairs.as<cmlhip_ba_pair>(),
                                                          c->res_scales[3], c->run_snap.as<cmlhip_ba_frame_state>());
    } else {
        if (pend) cml_launch_lin_finish(c, A);               // the tail of the last residual pass normally rides in the NEXT solve launch
        if (reanchor_newest)
            k_ba_reanchor_newest<<<1, 64, 0, c->stream>>>(c->frame_state.as<cmlhip_ba_frame_state>(), c->pre_w2c.as<double>(), c->pairs.as<cmlhip_ba_pair>(),
                                                            c->frames.as<FrameDev>(), (int)N, c->res_scales[3], c->run_snap.as<cmlhip_ba_frame_state>());
    }
    CML_CHECK(c, hipGetLastError());
    c->lin_finish_pending = false;
    A.ctl = nullptr;
    A.fuse_apply = 1;
    if (rs_close) {
        const bool relaxed = c->arith_relaxed;              // (exact, like every pass outside the iterations: include/cmlhip.h)
        c->arith_relaxed = false;
        cml_launch_linearize_rs(c, A);
        c->arith_relaxed = relaxed;
        c->efs_in_partials = true; c->lin_partial_n = c->n_tiles;
    } else cml_launch_linearize(c, A);
    cml_launch_lin_finish(c, A, CML_CLOSE_OFFSET);
    CML_CHECK(c, hipGetLastError());
    cml_mark(c, "closing");
    const bool packed = ro && (ro->state_good || ro->hdi);
    const size_t sg_bytes = (R + 255) & ~size_t(255);
    if (packed) {
        if ((rc = cml_ensure(c, c->run_pack, sg_bytes + 4 * P + 256))) return rc;
        const int work = (int)std::max(R, P);
        if (work > 0) k_ba_pack_closing<<<cml_div_up(work, 256), 256, 0, c->stream>>>((int)R, (int)P, c->c_dev_of.as<int>(), c->r_state.as<int>(), c->r_good.as<unsigned char>(),
                                                                                     c->pt_acc.as<float>(), c->run_pack.as<unsigned char>(), reinterpret_cast<float*>(c->run_pack.as<char>() + sg_bytes));
        CML_CHECK(c, hipGetLastError());
    }
    LinSummary S, Sfirst, Slast;
    ResidentCtl ctl;
    std::vector<float> pacc(point_acc ? PT_ACC_STRIDE * P : 0);
    ResRead rr(c);
    cml_d2h_batch_begin(c);
    cml_d2h(c, &S, c->scal.as<char>() + CML_CLOSE_OFFSET, sizeof S);
    if (ro) {
        cml_d2h(c, &Slast, c->scal.p, sizeof Slast);
        cml_d2h(c, &ctl, c->scal.as<char>() + CML_CTL_OFFSET, sizeof ctl);
        if (ro->frames) cml_d2h(c, ro->frames, reanchor_newest ? c->run_snap.p : c->frame_state.p, sizeof(cmlhip_ba_frame_state) * N);
        if (ro->pre_w2c) cml_d2h(c, ro->pre_w2c, c->pre_w2c.p, 8 * 7 * N);
        if (ro->x) cml_d2h(c, ro->x, c->xvec.p, 8 * n);
        if (ro->state_good && R) cml_d2h(c, ro->state_good, c->run_pack.p, R);
        if (ro->hdi && P) cml_d2h(c, ro->hdi, c->run_pack.as<char>() + sg_bytes, 4 * P);
    }
    rr.add(state, c->r_state.p, 4); rr.add(new_state, c->r_new_state.p, 4); rr.add(energy, c->r_energy.p, 4);
    rr.add(new_energy, c->r_new_energy.p, 4); rr.add(new_energy_wo, c->r_new_energy_wo.p, 4); rr.add(is_good, c->r_good.p, 1);
    if (idepth) cml_d2h(c, idepth, c->pt_idepth.p, 8 * P);
    if (point_acc && P) cml_d2h(c, pacc.data(), c->pt_acc.p, 4 * pacc.size());
    if ((rc = cml_d2h_batch_flush(c))) return rc;
    cml_mark(c, "gathered"); (void)hipStreamSynchronize(c->stream); cml_marks_dump(c);
    rr.deliver();
    if (point_acc) for (size_t p = 0; p < P; p++) memcpy(point_acc + 14 * p, &pacc[PT_ACC_STRIDE * p], 14 * 4);
    auto put = [](cmlhip_ba_lin_result* o, const LinSummary& s) { if (o) { o->energy = s.energy; o->n_in = s.n_in; o->n_oob = s.n_oob; o->n_outlier = s.n_outlier; o->new_frame_energy_th = s.new_frame_energy_th; } };
    put(lin, S);
    if (ro) {
        memset(&Sfirst, 0, sizeof Sfirst); Sfirst.energy = ctl.energy0;      // the preamble's tail rode in the first solve launch: its energy is in the log
        put(ro->first, Sfirst); put(ro->last, Slast);
        const int nit = c->conv_on ? ctl.iters_done : c->resident_iter;
        if (ro->iterations) *ro->iterations = nit;
        if (ro->energies) for (int i = 0; i < ro->capacity && i < nit && i < 40; i++) ro->energies[i] = ctl.energy[i];
    }
    if (R) k_ba_retire<<<cml_div_up((int)R, 256), 256, 0, c->stream>>>(A);
    CML_CHECK(c, hipGetLastError());
    if (ro && !std::isfinite(Slast.energy)) return CMLHIP_ERR_NONFINITE;
    return std::isfinite(S.energy) ? CMLHIP_OK : CMLHIP_ERR_NONFINITE;
}

static int upload_accum_in(cmlhip_ctx* c, const cmlhip_ba_accum_in* in) {
    const int N = c->N;
    int rc;
    if ((rc = cml_h2d(c, c->adH.p, in->adHost, 8 * 64 * (size_t)N * N))) return rc;
    if ((rc = cml_h2d(c, c->adT.p, in->adTarget, 8 * 64 * (size_t)N * N))) return rc;
    if ((rc = cml_h2d(c, c->adHTd.p, in->adHTdeltaF, 4 * 8 * (size_t)N * N))) return rc;
    std::vector<double> v(8 + 16 * N);
    for (int i = 0; i < 4; i++) { v[i] = in->cdelta[i]; v[4 + i] = in->cprior[i]; }
    for (int i = 0; i < 8 * N; i++) { v[8 + i] = in->prior[i]; v[8 + 8 * N + i] = in->delta_prior[i]; }
    return cml_h2d(c, c->vec_small.p, v.data(), v.size() * 8);
}

int cmlhip_ba_accumulate(cmlhip_ctx* c, const cmlhip_ba_accum_in* in, double* HA, double* bA, double* HL, double* bL,
                         double* Hsc, double* bsc) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    if (!in || !in->adHost || !in->adTarget || !in->adHTdeltaF || !in->cdelta || !in->prior || !in->delta_prior || !in->cprior)
        return CMLHIP_ERR_INVALID;
    if ((rc = upload_accum_in(c, in))) return rc;
    BAArgs A;
    cml_make_ba_args(c, A);
    cml_launch_accumulate(c, A, c->last_lambda, false, false);
    if (Hsc || bsc) cml_launch_schur_out(c, A);
    CML_CHECK(c, hipGetLastError());
    const size_t n = 8 * (size_t)c->N + 4;
    if (HA && (rc = cml_d2h(c, HA, c->HA.p, 8 * n * n))) return rc;
    if (bA && (rc = cml_d2h(c, bA, c->bA.p, 8 * n))) return rc;
    if (HL && (rc = cml_d2h(c, HL, c->HL.p, 8 * n * n))) return rc;
    if (bL && (rc = cml_d2h(c, bL, c->bL.p, 8 * n))) return rc;
    if (Hsc && (rc = cml_d2h(c, Hsc, c->Hsc.p, 8 * n * n))) return rc;
    if (bsc && (rc = cml_d2h(c, bsc, c->bsc.p, 8 * n))) return rc;
    return CMLHIP_OK;
}

int cmlhip_ba_solve(cmlhip_ctx* c, double lambda, const double* HM, const double* bM, int optcal, double* x) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    const size_t n = 8 * (size_t)c->N + 4;
    const bool have = HM && bM;
    if (have) {
        if ((rc = cml_h2d(c, c->HM.p, HM, 8 * n * n))) return rc;
        if ((rc = cml_h2d(c, c->bM.p, bM, 8 * n))) return rc;
    }
    BAArgs A;
    cml_make_ba_args(c, A);
    c->last_lambda = lambda; c->last_have_hm = have;
    cml_launch_accumulate(c, A, lambda, have, false, true); // pair blocks / Schur rows are current; rebuilds the final system for this lambda / HM
    if ((rc = cml_launch_solve(c, A, optcal, false))) return rc;   // (a window too wide for the LDS-resident factorisation is refused here)
    CML_CHECK(c, hipGetLastError());
    int flag = 0;
    if ((rc = cml_d2h(c, &flag, c->scal.as<char>() + 256, sizeof(int)))) return rc;
    if (x && (rc = cml_d2h(c, x, c->xvec.p, 8 * n))) return rc;
    return flag ? CMLHIP_ERR_NONFINITE : CMLHIP_OK;
}

int cmlhip_ba_backsub(cmlhip_ctx* c, const double* x, double* step) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    const size_t n = 8 * (size_t)c->N + 4;
    if (x && (rc = cml_h2d(c, c->xvec.p, x, 8 * n))) return rc;
    BAArgs A;
    cml_make_ba_args(c, A);
    CML_CHECK(c, hipMemsetAsync(&c->scal.as<LinSummary>()->nonfinite, 0, sizeof(int), c->stream));
    cml_launch_backsub(c, A, false);
    CML_CHECK(c, hipGetLastError());
    LinSummary S;
    if ((rc = cml_d2h(c, &S, c->scal.p, sizeof S))) return rc;
    if (step && (rc = cml_d2h(c, step, c->pt_step.p, 8 * (size_t)c->P))) return rc;
    return S.nonfinite ? CMLHIP_ERR_NONFINITE : CMLHIP_OK;
}

int cmlhip_ba_backup_points(cmlhip_ctx* c) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    BAArgs A;
    cml_make_ba_args(c, A);
    cml_launch_backup_points(c, A);
    CML_CHECK(c, hipGetLastError());
    return CMLHIP_OK;
}

int cmlhip_ba_restore_points(cmlhip_ctx* c) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    if ((rc = cml_materialize_records(c))) return rc;        // the resident kernel keeps Jacobians in reduced form: re-create the records first
    BAArgs A;
    cml_make_ba_args(c, A);
    c->r_idepth_dirty = true;
    cml_launch_restore_points(c, A);
    CML_CHECK(c, hipGetLastError());
    return CMLHIP_OK;
}

static int read_step_sums(cmlhip_ctx* c, float sums[3]) {
    const int nb = (c->P + 255) / 256;
    if (nb == 0) { sums[0] = sums[1] = sums[2] = 0.f; return CMLHIP_OK; }      // a window without points: empty sums, no copy
    std::vector<float> part(4 * (size_t)nb);
    int rc = cml_d2h(c, part.data(), c->step_partial.p, 16 * (size_t)nb);
    if (rc) return rc;
    float a = 0, b = 0, n = 0;
    for (int i = 0; i < nb; i++) { a += part[4 * i]; b += part[4 * i + 1]; n += part[4 * i + 2]; }
    sums[0] = a; sums[1] = b; sums[2] = n;
    return CMLHIP_OK;
}

int cmlhip_ba_step_points(cmlhip_ctx* c, float sums[3]) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    if ((rc = cml_materialize_records(c))) return rc;        // the resident kernel keeps Jacobians in reduced form: re-create the records first
    BAArgs A;
    cml_make_ba_args(c, A);
    c->r_idepth_dirty = true;
    cml_launch_step_points(c, A);
    CML_CHECK(c, hipGetLastError());
    if (sums) return read_step_sums(c, sums);
    return CMLHIP_OK;
}

// one Gauss-Newton iteration enqueued back to back, no host round trip (iterations 0 and 1 of BA::run, which do
// not orthogonalise: BA.cpp:1404 `iteration >= 2`): backup -> accumulate -> solve -> backsub -> step -> linearize -> apply.
// The adjoints / priors of the last cmlhip_ba_accumulate call are reused.
int cmlhip_ba_iteration_async(cmlhip_ctx* c, double lambda) { CML_DEV(c);
    int rc = ba_check(c, true);
    if (rc) return rc;
    BAArgs A;
    cml_make_ba_args(c, A);
    A.fuse_apply = 1;                                        // the step is always accepted here (forceAccept, BA.h:265)
    if (c->resident_on && c->conv_on) {                  // mirror of run()'s early exit: see ResidentCtl
        A.ctl = reinterpret_cast<ResidentCtl*>(c->scal.as<char>() + CML_CTL_OFFSET);
        A.it_index = c->resident_iter; A.th_opt = c->conv_th;
    }
    const bool prof = c->prof_cap > 0 && c->prof_n < c->prof_cap && (c->prof_tick++ % c->prof_stride) == 0;
    hipEvent_t* ev = prof ? &c->prof_ev[4 * (size_t)c->prof_n] : nullptr;
    // (the dispatch AHEAD of the residual kernel always carries its end event when the residual kernel is timed: a dispatch that follows
    //  one without a completion signal takes its begin timestamp while the predecessor is still draining — measured 9.8 us against the
    //  8.8 us rocprofv3 reports for the same kernel, whose tracer puts a signal on every dispatch)
    const bool prof_ss = prof && (c->prof_mask & 2), prof_k1 = prof && (c->prof_mask & 1);
    if (prof_ss) c->ext_start = ev[0];                       // begin timestamp of the K3 dispatch
    cml_launch_accumulate(c, A, lambda, c->resident_on && c->resident_prior, true);        // K3 (+ backup) and K4 (+ the marginalisation prior when it is resident)
    const bool mix = c->resident_on && c->rp_resident && c->N > 4;      // addIndirectToProblem, BA.cpp:1327-1329 (only with more than 4 frames)
    ReprojArgs rp;
    if (mix) cml_resident_reproj_args(c, lambda, c->resident_iter + 1, &rp);   // its per-frame workgroups ride in the solve launch
    // K5: solve (+ hybrid term beside it, + orthogonalize, BA.cpp:1404) || energy threshold of the previous residual pass
    static const bool no_merge = getenv("CMLHIP_NO_MERGE") != nullptr;          // development: K6 as its own launch
    c->ext_stop_if_merged = (prof && !no_merge) ? ev[1] : nullptr;      // (merged: the end of the K5 + K6 dispatch closes the Schur-reduce + solve group)
    if ((rc = cml_launch_solve(c, A, 0, true, c->resident_on && c->have_null && c->resident_iter >= 2, mix ? c->rr_x.as<double>() : nullptr, mix ? &rp : nullptr, !no_merge))) return rc;
    c->resident_iter++;
    if (!c->backsub_merged) {
        if (prof) c->ext_stop = ev[1];                        // end timestamp of the K6 dispatch
        cml_launch_backsub(c, A, true);                       // K6: back-substitution + point update
    }
    if (prof_k1) { c->ext_start = ev[2]; c->ext_stop = ev[3]; } // begin / end timestamps of the K1 dispatch itself
    if (c->rs_ok) {                                          // K1: residuals + Jacobians + applyRes, Jacobians kept in reduced form
        cml_launch_linearize_rs(c, A);
        c->efs_in_partials = true; c->lin_partial_n = c->n_tiles;
    } else {
        cml_launch_linearize(c, A);
    }
    c->ext_start = c->ext_stop = nullptr;
    c->lin_finish_pending = true;
    if (prof) c->prof_n++;
    CML_CHECK(c, hipGetLastError());
    cml_mark(c, "iter");
    return CMLHIP_OK;
}

int cmlhip_ba_iteration_batch(cmlhip_ctx* const* ctxs, int S, double lambda) {
    if (!ctxs || S < 1 || S > 64 || !ctxs[0]) return CMLHIP_ERR_INVALID;
    cmlhip_ctx* c0 = ctxs[0];
    CML_DEV(c0);
    for (int k = 0; k < S; k++) {
        cmlhip_ctx* c = ctxs[k];
        if (!c) return CMLHIP_ERR_INVALID;
        for (int q = 0; q < k; q++) if (ctxs[q] == c) { c0->err = "cmlhip_ba_iteration_batch: a context appears twice"; return CMLHIP_ERR_INVALID; }
        int rc = ba_check(c, true);
        if (rc) { if (c != c0) c0->err = c->err; return rc; }
        CML_REQUIRE(c0, c->device == c0->device && c->lim.texel_format == c0->lim.texel_format, CMLHIP_ERR_INVALID, "cmlhip_ba_iteration_batch: contexts of one device and one texel format");
        CML_REQUIRE(c0, c->resident_on, CMLHIP_ERR_STATE, "cmlhip_ba_iteration_batch: cmlhip_ba_set_resident_state not called for a window");
        CML_REQUIRE(c0, !(c->rp_resident && c->N > 4), CMLHIP_ERR_STATE, "cmlhip_ba_iteration_batch: the hybrid term is not batched (use cmlhip_ba_iteration_async)");
        CML_REQUIRE(c0, c->n_lin == 0 && !c->conv_on, CMLHIP_ERR_STATE, "cmlhip_ba_iteration_batch: no LINEARIZED residuals and no convergence control in a batch");
    }
    return cml_iteration_batch(ctxs, S, lambda);
}

// ---------------------------------------------------------------------------------------------- marginalisation
static int upload_point_mask(cmlhip_ctx* c, int n, const int* idx) {
    std::vector<unsigned char> m((size_t)std::max(c->P, 1), 0);
    for (int i = 0; i < n; i++) {
        if (idx[i] < 0 || idx[i] >= c->P) { c->err = "point index out of range"; return CMLHIP_ERR_INVALID; }
        m[idx[i]] = 1;
    }
    int rc = cml_ensure(c, c->pt_mask, m.size());
    if (rc) return rc;
    return cml_h2d(c, c->pt_mask.p, m.data(), m.size());
}

// the inputs of a marginalisation pass (adjoints, deltas, priors, the point mask, a cleared counter) in ONE packed upload: a scatter kernel reading the
// pinned staging block instead of five copies / fills of their own (each a launch on the stream and a runtime call on the host)
static int upload_marg_inputs(cmlhip_ctx* c, const cmlhip_ba_accum_in* in, int n, const int* point_idx, int* counter_to_clear) {
    cml_h2d_batch_begin(c);
    int rc = upload_accum_in(c, in);
    if (!rc) rc = upload_point_mask(c, n, point_idx);
    if (!rc && counter_to_clear) rc = cml_zero(c, counter_to_clear, sizeof(int));            // (a zero segment of the same batch)
    const int rf = cml_h2d_batch_flush(c);
    return rc ? rc : rf;
}

int cmlhip_ba_relinearize_points(cmlhip_ctx* c, const cmlhip_ba_accum_in* in, int n, const int* point_idx, int* n_good) { CML_DEV(c);
    int rc = ba_check(c, true);
    if (rc) return rc;
    if ((rc = cml_materialize_records(c))) return rc;        // the resident kernel keeps Jacobians in reduced form: re-create the records first
    if (!in || !in->adHost || !in->adTarget || !in->adHTdeltaF || !in->cdelta || !in->prior || !in->delta_prior || !in->cprior || n < 0 || (n > 0 && !point_idx))
        return CMLHIP_ERR_INVALID;
    int* counter = reinterpret_cast<int*>(c->scal.as<char>() + 512);
    if ((rc = upload_marg_inputs(c, in, n, point_idx, counter))) return rc;
    BAArgs A;
    cml_make_ba_args(c, A);
    A.pt_mask = c->pt_mask.as<unsigned char>();
    A.lin_partial = nullptr;                                 // a subset pass: no window energy
    A.fuse_apply = 1;                                        // applyRes(r, true), BA.cpp:2299
    cml_launch_marg_reset(c, A);
    cml_launch_linearize(c, A);
    cml_launch_marg_fix(c, A, c->adHTd.as<float>(), c->vec_small.as<double>(), counter);
    CML_CHECK(c, hipGetLastError());
    int ng = 0;
    if ((rc = cml_d2h(c, &ng, counter, sizeof(int)))) return rc;
    if (ng > 0) c->n_lin = std::max(c->n_lin, 1);            // the LINEARIZED blocks of the regular accumulation are live from now on
    if (n_good) *n_good = ng;
    return CMLHIP_OK;
}

// the same pass with what tryMarginalize's host loop reads of it (BA.cpp:2296-2304: state, isActiveAndIsGoodNEW, isLinearized of the candidates'
// residuals) packed as ONE byte per residual in the caller's order, behind the counter in ONE readback — it was the counter, six R-length arrays
// and the LINEARIZED flags in three synchronous calls
// (e0 .. e2: the three energy arrays the caller asked for, in its order too — they were a launch each)
__global__ void k_ba_pack_marg(int R, const int* __restrict__ c_dev_of, const int* __restrict__ r_state, const int* __restrict__ r_new_state,
                               const unsigned char* __restrict__ r_good, const unsigned char* __restrict__ r_lin, unsigned char* __restrict__ out,
                               const float* __restrict__ e0, const float* __restrict__ e1, const float* __restrict__ e2, float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ o2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R) {
        const int k = c_dev_of[i];
        out[i] = (unsigned char)((r_state[k] & 3) | (r_good[k] ? 4 : 0) | (r_lin[k] ? 8 : 0) | ((r_new_state[k] & 3) << 4));
        if (o0) o0[i] = e0[k];
        if (o1) o1[i] = e1[k];
        if (o2) o2[i] = e2[k];
    }
}
int cmlhip_ba_relinearize_points_packed(cmlhip_ctx* c, const cmlhip_ba_accum_in* in, int n, const int* point_idx, int* n_good, unsigned char* packed,
                                        float* energy, float* new_energy, float* new_energy_wo) { CML_DEV(c);
    int rc = ba_check(c, true);
    if (rc) return rc;
    if ((rc = cml_materialize_records(c))) return rc;
    if (!in || !in->adHost || !in->adTarget || !in->adHTdeltaF || !in->cdelta || !in->prior || !in->delta_prior || !in->cprior || n < 0 || (n > 0 && !point_idx) || !packed)
        return CMLHIP_ERR_INVALID;
    int* counter = reinterpret_cast<int*>(c->scal.as<char>() + 512);
    if ((rc = upload_marg_inputs(c, in, n, point_idx, counter))) return rc;
    BAArgs A;
    cml_make_ba_args(c, A);
    A.pt_mask = c->pt_mask.as<unsigned char>();
    A.lin_partial = nullptr;
    A.fuse_apply = 1;
    cml_launch_marg_reset(c, A);
    cml_launch_linearize(c, A);
    cml_launch_marg_fix(c, A, c->adHTd.as<float>(), c->vec_small.as<double>(), counter);
    const size_t R = c->R, Rb = (R + 255) & ~size_t(255), Rf = (4 * R + 255) & ~size_t(255);
    if ((rc = cml_ensure(c, c->run_pack, Rb + 3 * Rf + 4 * (size_t)c->P + 256))) return rc;
    unsigned char* pk = c->run_pack.as<unsigned char>();
    float* o0 = energy ? reinterpret_cast<float*>(pk + Rb) : nullptr;
    float* o1 = new_energy ? reinterpret_cast<float*>(pk + Rb + Rf) : nullptr;
    float* o2 = new_energy_wo ? reinterpret_cast<float*>(pk + Rb + 2 * Rf) : nullptr;
    if (R > 0) k_ba_pack_marg<<<cml_div_up((int)R, 256), 256, 0, c->stream>>>((int)R, c->c_dev_of.as<int>(), c->r_state.as<int>(), c->r_new_state.as<int>(),
                                                                             c->r_good.as<unsigned char>(), c->r_lin.as<unsigned char>(), pk,
                                                                             c->r_energy.as<float>(), c->r_new_energy.as<float>(), c->r_new_energy_wo.as<float>(), o0, o1, o2);
    CML_CHECK(c, hipGetLastError());
    int ng = 0;
    cml_d2h_batch_begin(c);
    cml_d2h(c, &ng, counter, sizeof(int));
    if (R > 0) {
        cml_d2h(c, packed, pk, R);
        if (o0) cml_d2h(c, energy, o0, 4 * R);
        if (o1) cml_d2h(c, new_energy, o1, 4 * R);
        if (o2) cml_d2h(c, new_energy_wo, o2, 4 * R);
    }
    if ((rc = cml_d2h_batch_flush(c))) return rc;
    if (ng > 0) c->n_lin = std::max(c->n_lin, 1);
    if (n_good) *n_good = ng;
    return CMLHIP_OK;
}

int cmlhip_ba_marginalize_points(cmlhip_ctx* c, const cmlhip_ba_accum_in* in, int n, const int* point_idx, double* M, double* Mb,
                                 double* Msc, double* Mbsc) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    if ((rc = cml_materialize_records(c))) return rc;        // the resident kernel keeps Jacobians in reduced form: re-create the records first
    if (!in || !in->adHost || !in->adTarget || !in->adHTdeltaF || !in->cdelta || !in->prior || !in->delta_prior || !in->cprior || n < 0 || (n > 0 && !point_idx))
        return CMLHIP_ERR_INVALID;
    if ((rc = upload_marg_inputs(c, in, n, point_idx, nullptr))) return rc;
    BAArgs A;
    cml_make_ba_args(c, A);
    A.pt_mask = c->pt_mask.as<unsigned char>();
    cml_launch_accumulate(c, A, 0.0, false, false, false, true);
    cml_launch_schur_out(c, A);
    CML_CHECK(c, hipGetLastError());
    const size_t nn = 8 * (size_t)c->N + 4;
    cml_d2h_batch_begin(c);                                  // the four results in ONE gather + copy + wait (they were four synchronous copies)
    if (M) cml_d2h(c, M, c->HA.p, 8 * nn * nn);
    if (Mb) cml_d2h(c, Mb, c->bA.p, 8 * nn);
    if (Msc) cml_d2h(c, Msc, c->Hsc.p, 8 * nn * nn);
    if (Mbsc) cml_d2h(c, Mbsc, c->bsc.p, 8 * nn);
    return cml_d2h_batch_flush(c);
}

int cmlhip_ba_lin_energy(cmlhip_ctx* c, const cmlhip_ba_accum_in* in, double* energy, int* num) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    if ((rc = cml_materialize_records(c))) return rc;        // the resident kernel keeps Jacobians in reduced form: re-create the records first
    if (!in || !in->adHTdeltaF || !in->cdelta || !in->prior || !in->delta_prior || !in->adHost || !in->adTarget || !in->cprior) return CMLHIP_ERR_INVALID;
    if ((rc = upload_accum_in(c, in))) return rc;
    const int nb = (c->P + 255) / 256;
    if ((rc = cml_ensure(c, c->marg_scratch, 16 * (size_t)std::max(nb, 1)))) return rc;
    BAArgs A;
    cml_make_ba_args(c, A);
    double* part = c->marg_scratch.as<double>();
    int* nums = reinterpret_cast<int*>(part + std::max(nb, 1));
    cml_launch_lin_energy(c, A, c->adHTd.as<float>(), c->vec_small.as<double>(), part, nums);
    CML_CHECK(c, hipGetLastError());
    std::vector<double> hp(std::max(nb, 1)); std::vector<int> hn(std::max(nb, 1));
    if (nb > 0) {
        cml_d2h_batch_begin(c);
        cml_d2h(c, hp.data(), part, 8 * (size_t)nb);
        cml_d2h(c, hn.data(), nums, 4 * (size_t)nb);
        if ((rc = cml_d2h_batch_flush(c))) return rc;
    }
    double F = 0;                                            // BA.cpp:2131-2142
    for (int i = 0; i < 8 * c->N; i++) F += in->delta_prior[i] * in->prior[i] * in->delta_prior[i];
    for (int i = 0; i < 4; i++) F += in->cdelta[i] * 5e9 * in->cdelta[i];
    double E = 0; int k = 0;
    for (int i = 0; i < nb; i++) { E += hp[i]; k += hn[i]; }
    if (energy) *energy = E + F;
    if (num) *num = k;
    return CMLHIP_OK;
}

int cmlhip_ba_get_res_to_zero(cmlhip_ctx* c, float* rtz, unsigned char* is_lin) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    ResRead rr(c);
    rr.add(rtz, c->r_rtz.p, 32); rr.add(is_lin, c->r_lin.p, 1);
    return rr.read_now();
}

int cmlhip_ba_set_resident_state(cmlhip_ctx* c, const cmlhip_ba_accum_in* in, const cmlhip_ba_frame_state* frames, const double scales[4],
                                 const double* nullspace_basis) { CML_DEV_SCOPED(c);
    int rc = ba_check(c, true);
    if (rc) return rc;
    if (!in || !in->adHost || !in->adTarget || !in->adHTdeltaF || !in->cdelta || !in->prior || !in->delta_prior || !in->cprior || !frames || !scales)
        return CMLHIP_ERR_INVALID;
    const int N = c->N, n = 8 * N + 4;
    const size_t Ncap = (size_t)std::max(N, c->lim.max_frames);           // (by the limit: a growing window does not re-allocate at every keyframe)
    if ((rc = cml_ensure(c, c->frame_state, sizeof(cmlhip_ba_frame_state) * Ncap))) return rc;
    if ((rc = cml_ensure(c, c->pre_w2c, 8 * 7 * Ncap))) return rc;
    if (nullspace_basis && (rc = cml_ensure(c, c->null_basis, 8 * 7 * (8 * Ncap + 4)))) return rc;
    cml_h2d_batch_begin(c);                                              // adjoints, deltas, priors, frame states, gauge basis: one copy
    struct BatchGuard { cmlhip_ctx* c; ~BatchGuard() { if (c->h2d_batching && !c->h2d_scope) { c->h2d_batching = false; c->h2d_segs.clear(); } } } batch_guard{c};
    if ((rc = upload_accum_in(c, in))) return rc;
    if ((rc = cml_h2d(c, c->frame_state.p, frames, sizeof(cmlhip_ba_frame_state) * (size_t)N))) return rc;
    if ((rc = cml_zero(c, c->pre_w2c.p, 8 * 7 * (size_t)N))) return rc;
    for (int i = 0; i < 4; i++) c->res_scales[i] = scales[i];
    c->have_null = nullspace_basis != nullptr;
    if (c->have_null && (rc = cml_h2d(c, c->null_basis.p, nullspace_basis, 8 * 7 * (size_t)n))) return rc;
    if ((rc = cml_h2d_batch_flush(c))) return rc;
    c->resident_on = true; c->resident_iter = 0; c->conv_th = 0; c->conv_on = false; c->rp_resident = false; c->resident_prior = false;
    return CMLHIP_OK;
}

// bM_top of the frame states as they stand (iteration 0 of a resident run; afterwards the frame step keeps it current)
__global__ void k_ba_prior_rhs(const cmlhip_ba_frame_state* __restrict__ fs, int N, FrameStepArgs F) {
    __shared__ double s_delta[CMLHIP_MAX_FRAMES][8];
    for (int e = threadIdx.x; e < N * 8; e += blockDim.x) s_delta[e >> 3][e & 7] = fs[e >> 3].state[e & 7] - fs[e >> 3].state_zero[e & 7];
    __syncthreads();
    frame_prior_rhs(F, s_delta);
}

int cmlhip_ba_set_resident_prior(cmlhip_ctx* c, const double* HM, const double* bM) { CML_DEV_SCOPED(c);
    int rc = ba_check(c, true);
    if (rc) return rc;
    CML_REQUIRE(c, c->resident_on, CMLHIP_ERR_STATE, "cmlhip_ba_set_resident_state not called for this window");
    if (!HM || !bM) { c->resident_prior = false; return CMLHIP_OK; }
    const size_t n = 8 * (size_t)c->N + 4;
    if ((rc = cml_ensure(c, c->bM_raw, 8 * (8 * (size_t)std::max(c->N, c->lim.max_frames) + 4)))) return rc;
    if ((rc = cml_h2d(c, c->HM.p, HM, 8 * n * n))) return rc;
    if ((rc = cml_h2d(c, c->bM_raw.p, bM, 8 * n))) return rc;
    FrameStepArgs F = {};
    F.n = (int)n; F.HM = c->HM.as<double>(); F.bM_raw = c->bM_raw.as<double>(); F.bM_top = c->bM.as<double>();
    if ((rc = cml_defer(c, [c, F]() -> int {
            k_ba_prior_rhs<<<1, 128, 0, c->stream>>>(c->frame_state.as<cmlhip_ba_frame_state>(), c->N, F);
            CML_CHECK(c, hipGetLastError());
            return CMLHIP_OK; }))) return rc;
    c->resident_prior = true;
    return CMLHIP_OK;
}

int cmlhip_ba_resident_convergence(cmlhip_ctx* c, double th_opt_iterations) { CML_DEV_SCOPED(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    CML_REQUIRE(c, c->resident_on, CMLHIP_ERR_STATE, "cmlhip_ba_set_resident_state not called for this window");
    c->conv_th = th_opt_iterations > 0 ? th_opt_iterations : 0.0;     // 0: the test can never pass, the log is still kept
    c->conv_on = true;
    return cml_zero(c, c->scal.as<char>() + CML_CTL_OFFSET, sizeof(ResidentCtl));      // (a zero segment of the open upload scope, or a memset on the stream)
}

int cmlhip_ba_get_resident_log(cmlhip_ctx* c, int* iterations, double* energies, int capacity) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    CML_REQUIRE(c, c->resident_on, CMLHIP_ERR_STATE, "cmlhip_ba_set_resident_state not called for this window");
    if (c->lin_finish_pending) {
        BAArgs A;
        cml_make_ba_args(c, A);
        if (c->conv_on) A.ctl = reinterpret_cast<ResidentCtl*>(c->scal.as<char>() + CML_CTL_OFFSET);
        cml_launch_lin_finish(c, A);
        CML_CHECK(c, hipGetLastError());
        c->lin_finish_pending = false;
    }
    ResidentCtl h;
    if ((rc = cml_d2h(c, &h, c->scal.as<char>() + CML_CTL_OFFSET, sizeof h))) return rc;
    const int n = c->conv_on ? h.iters_done : c->resident_iter;
    if (iterations) *iterations = n;
    if (energies) for (int i = 0; i < capacity && i < n && i < 40; i++) energies[i] = h.energy[i];
    return CMLHIP_OK;
}

int cmlhip_ba_get_resident_state(cmlhip_ctx* c, cmlhip_ba_frame_state* frames, double* pre_w2c, cmlhip_ba_lin_result* last) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    CML_REQUIRE(c, c->resident_on, CMLHIP_ERR_STATE, "cmlhip_ba_set_resident_state not called for this window");
    if (c->lin_finish_pending) {                             // the tail of the last residual pass normally rides in the NEXT solve launch
        BAArgs A;
        cml_make_ba_args(c, A);
        if (c->conv_on) A.ctl = reinterpret_cast<ResidentCtl*>(c->scal.as<char>() + CML_CTL_OFFSET);
        cml_launch_lin_finish(c, A);
        CML_CHECK(c, hipGetLastError());
        c->lin_finish_pending = false;
    }
    if (last) {
        LinSummary S;
        if ((rc = cml_d2h(c, &S, c->scal.p, sizeof S))) return rc;
        last->energy = S.energy; last->n_in = S.n_in; last->n_oob = S.n_oob; last->n_outlier = S.n_outlier;
        last->new_frame_energy_th = S.new_frame_energy_th;
    }
    if (frames && (rc = cml_d2h(c, frames, c->frame_state.p, sizeof(cmlhip_ba_frame_state) * (size_t)c->N))) return rc;
    if (pre_w2c && (rc = cml_d2h(c, pre_w2c, c->pre_w2c.p, 8 * 7 * (size_t)c->N))) return rc;
    return CMLHIP_OK;
}

int cmlhip_debug_timestamps(cmlhip_ctx* c, int enable, long long* out128) { CML_DEV(c);
    const size_t NS = CMLHIP_DEBUG_SLOTS;
    if (!c) return CMLHIP_ERR_INVALID;
    const size_t NX = CML_DEBUG_RS_TILES * 8;              // development: 7 phase stamps + hw id per tile of the lane-per-residual kernel (-DCML_RS_STAMPS)
    int rc = cml_ensure(c, c->dbg, (NS + NX) * sizeof(long long));
    if (rc) return rc;
    if (out128 && (rc = cml_d2h(c, out128, c->dbg.p, NS * sizeof(long long)))) return rc;
    if (out128) {
        if (const char* f = getenv("CMLHIP_RS_TS_FILE")) {
            std::vector<long long> x(NX);
            if ((rc = cml_d2h(c, x.data(), (char*)c->dbg.p + NS * sizeof(long long), NX * sizeof(long long)))) return rc;
            if (FILE* fp = fopen(f, "wb")) { fwrite(x.data(), sizeof(long long), NX, fp); fclose(fp); }
        }
    }
    c->dbg_on = enable != 0;
    if (enable) CML_CHECK(c, hipMemsetAsync(c->dbg.p, 0, (NS + NX) * sizeof(long long), c->stream));
    return CMLHIP_OK;
}

int cmlhip_profile_enable(cmlhip_ctx* c, int max_iterations) { CML_DEV(c);
    if (!c || max_iterations < 0) return CMLHIP_ERR_INVALID;
    (void)hipStreamSynchronize(c->stream);
    for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
    c->prof_ev.clear();
    c->prof_cap = max_iterations; c->prof_n = 0; c->prof_tick = 0;
    c->prof_ev.resize(4 * (size_t)max_iterations);
    for (auto& e : c->prof_ev) CML_CHECK(c, hipEventCreate(&e));
    return CMLHIP_OK;
}

int cmlhip_profile_stride(cmlhip_ctx* c, int stride) { CML_DEV(c);
    if (!c || stride < 1) return CMLHIP_ERR_INVALID;
    c->prof_stride = stride; c->prof_tick = 0;
    return CMLHIP_OK;
}

int cmlhip_profile_select(cmlhip_ctx* c, int mask) { CML_DEV(c);
    if (!c || (mask & ~3) || !mask) return CMLHIP_ERR_INVALID;
    c->prof_mask = mask;
    return CMLHIP_OK;
}

int cmlhip_profile_read(cmlhip_ctx* c, float* lin_ms, float* ss_ms, float* empty_ms, int* n) { CML_DEV(c);
    if (!c) return CMLHIP_ERR_INVALID;
    CML_CHECK(c, hipStreamSynchronize(c->stream));
    double a = 0, b = 0, e = 0;
    for (int i = 0; i < c->prof_n; i++) {
        float m0 = 0, m1 = 0;
        if (c->prof_mask & 2) CML_CHECK(c, hipEventElapsedTime(&m0, c->prof_ev[4 * (size_t)i + 0], c->prof_ev[4 * (size_t)i + 1]));   // K3 begin -> K6 end
        if (c->prof_mask & 1) CML_CHECK(c, hipEventElapsedTime(&m1, c->prof_ev[4 * (size_t)i + 2], c->prof_ev[4 * (size_t)i + 3]));   // K1 begin -> K1 end
        b += m0; a += m1;
    }
    if (n) *n = c->prof_n;
    if (lin_ms) *lin_ms = c->prof_n ? (float)(a / c->prof_n) : 0.f;
    if (ss_ms) *ss_ms = c->prof_n ? (float)(b / c->prof_n) : 0.f;
    if (empty_ms) *empty_ms = c->prof_n ? (float)(e / c->prof_n) : 0.f;
    c->prof_n = 0;
    return CMLHIP_OK;
}

// ---------------------------------------------------------------------------------------------- readbacks
int cmlhip_ba_get_states(cmlhip_ctx* c, int* state, int* new_state, float* energy, float* new_energy, float* new_energy_wo,
                         unsigned char* is_good) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    ResRead rr(c);
    cml_d2h_batch_begin(c);                                             // six arrays, one round trip
    rr.add(state, c->r_state.p, 4); rr.add(new_state, c->r_new_state.p, 4); rr.add(energy, c->r_energy.p, 4);
    rr.add(new_energy, c->r_new_energy.p, 4); rr.add(new_energy_wo, c->r_new_energy_wo.p, 4); rr.add(is_good, c->r_good.p, 1);
    if ((rc = cml_d2h_batch_flush(c))) return rc;
    rr.deliver();
    return CMLHIP_OK;
}

int cmlhip_ba_get_pairs(cmlhip_ctx* c, cmlhip_ba_pair* pairs, float* frame_energy_th, float* b0) { CML_DEV(c);
    int rc = ba_check(c, true);
    if (rc) return rc;
    if (pairs && (rc = cml_d2h(c, pairs, c->pairs.p, sizeof(cmlhip_ba_pair) * (size_t)c->N * c->N))) return rc;
    if (frame_energy_th || b0) {
        std::vector<FrameDev> fd(c->N);
        if ((rc = cml_d2h(c, fd.data(), c->frames.p, sizeof(FrameDev) * (size_t)c->N))) return rc;
        for (int i = 0; i < c->N; i++) { if (frame_energy_th) frame_energy_th[i] = fd[i].frame_energy_th; if (b0) b0[i] = fd[i].b0; }
    }
    return CMLHIP_OK;
}

int cmlhip_ba_get_rj(cmlhip_ctx* c, int which, float* out) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    if ((rc = cml_materialize_records(c))) return rc;        // the resident kernel keeps Jacobians in reduced form: re-create the records first
    if (!out) return CMLHIP_ERR_INVALID;
    const size_t R = c->R;
    std::vector<float> b0(RJ_STRIDE * R), b1(RJ_STRIDE * R);
    std::vector<unsigned char> sel(R);
    if ((rc = cml_d2h(c, b0.data(), c->rj[0].p, 4 * RJ_STRIDE * R))) return rc;
    if ((rc = cml_d2h(c, b1.data(), c->rj[1].p, 4 * RJ_STRIDE * R))) return rc;
    if ((rc = cml_d2h(c, sel.data(), c->r_sel.p, R))) return rc;
    for (size_t r = 0; r < R; r++) {
        const size_t k = (size_t)c->h_dev_of[r];
        const bool efs_in_1 = sel[k] != 0;
        const float* src = ((which == 1) == efs_in_1 ? b1.data() : b0.data()) + RJ_STRIDE * k;
        memcpy(out + CMLHIP_RJ_FLOATS * r, src, 4 * CMLHIP_RJ_FLOATS);
    }
    return CMLHIP_OK;
}

int cmlhip_ba_get_jpjdf(cmlhip_ctx* c, float* out) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    std::vector<float> wide(PS_STRIDE * (size_t)c->R);               // the device keeps 16 floats per residual (JpJdF + point-sum terms)
    if (c->R && (rc = cml_d2h(c, wide.data(), c->r_jpjdf.p, 4 * wide.size()))) return rc;
    for (size_t r = 0; r < (size_t)c->R; r++) memcpy(out + 8 * r, &wide[PS_STRIDE * (size_t)c->h_dev_of[r]], 32);
    return CMLHIP_OK;
}
int cmlhip_ba_get_center_projected(cmlhip_ctx* c, float* out) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    ResRead rr(c);
    rr.add(out, c->r_center.p, 12);
    return rr.read_now();
}
int cmlhip_ba_get_point_acc(cmlhip_ctx* c, float* out) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    std::vector<float> t(PT_ACC_STRIDE * (size_t)c->P);
    if ((rc = cml_d2h(c, t.data(), c->pt_acc.p, 4 * t.size()))) return rc;
    for (int p = 0; p < c->P; p++) memcpy(out + 14 * (size_t)p, &t[PT_ACC_STRIDE * (size_t)p], 14 * 4);
    return CMLHIP_OK;
}
int cmlhip_ba_get_pair_acc(cmlhip_ctx* c, int mode, float* out) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    if (mode < 0 || mode > 1 || !out) return CMLHIP_ERR_INVALID;
    const int NN = c->N * c->N;
    std::vector<float> t(ACC_STRIDE * (size_t)NN);
    if ((rc = cml_d2h(c, t.data(), c->acc_pair[mode].p, 4 * t.size()))) return rc;
    for (int q = 0; q < NN; q++) {
        float* H = out + 169 * (size_t)q;
        const float* a = &t[ACC_STRIDE * (size_t)q];
        memset(H, 0, 169 * 4);
        int idx = 0;
        for (int r = 0; r < 10; r++) for (int cc = r; cc < 10; cc++) { H[r * 13 + cc] = H[cc * 13 + r] = a[idx]; idx++; }
        idx = 0;
        for (int r = 0; r < 10; r++) for (int cc = 0; cc < 3; cc++) { H[r * 13 + cc + 10] = H[(cc + 10) * 13 + r] = a[55 + idx]; idx++; }
        H[10 * 13 + 10] = a[85]; H[10 * 13 + 11] = H[11 * 13 + 10] = a[86]; H[10 * 13 + 12] = H[12 * 13 + 10] = a[87];
        H[11 * 13 + 11] = a[88]; H[11 * 13 + 12] = H[12 * 13 + 11] = a[89]; H[12 * 13 + 12] = a[90];
    }
    return CMLHIP_OK;
}
int cmlhip_ba_get_index_maps(cmlhip_ctx* c, int* pair_of, int* bpo, int* bp, int* bqo, int* bq) { CML_DEV(c);
    int rc = ba_check(c, false);
    if (rc) return rc;
    const size_t R = (size_t)c->R;
    if (pair_of) memcpy(pair_of, c->h_pair_of.data(), 4 * R);
    if (bpo) memcpy(bpo, c->h_by_point_off.data(), 4 * (size_t)(c->P + 1));
    if (bqo) memcpy(bqo, c->h_by_pair_off.data(), 4 * (size_t)(c->N * c->N + 1));
    if ((bp || bq) && R) {
        // the lists as the DEVICE holds them (k_window_expand wrote them), translated back to the caller's numbering
        std::vector<int> caller_of(R), dbp(R), dbq(R);
        for (size_t r = 0; r < R; r++) caller_of[(size_t)c->h_dev_of[r]] = (int)r;
        if ((rc = cml_d2h(c, dbp.data(), c->by_point.p, 4 * R))) return rc;
        if ((rc = cml_d2h(c, dbq.data(), c->by_pair.p, 4 * R))) return rc;
        for (size_t i = 0; i < R; i++) {
            if ((unsigned)dbp[i] >= R || (unsigned)dbq[i] >= R) { c->err = "cmlhip_ba_get_index_maps: a device list entry is out of range"; return CMLHIP_ERR_STATE; }
            if (bp) bp[i] = caller_of[(size_t)dbp[i]];
            if (bq) bq[i] = caller_of[(size_t)dbq[i]];
        }
    }
    return CMLHIP_OK;
}

}  // extern "C"
