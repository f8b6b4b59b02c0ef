// cmlhip_ctx.hip — context, device memory, pyramid cache and the image kernels.
// Reference: GradientImage layout/ownership src/cml/types.h:915, src/cml/capture/CaptureImage.cpp:209-403;
// image ops src/cml/image/Array2D.h:288-327 (gradient), :388-401 (reduceByTwo).
#include "cmlhip_internal.h"
#include <cstdlib>
#include <algorithm>

void cml_window_free(cmlhip_ctx* c) {
    WindowShadow& W = c->win;
    if (W.busy) (void)hipEventDestroy(W.busy);
    if (W.block) (void)hipHostFree(W.block);
    W = WindowShadow{};
}
void cml_mark(cmlhip_ctx* c, const char* what) {
    static const bool on = getenv("CMLHIP_RUN_MARKS") != nullptr;
    if (!on) return;
    if (c->marks_n == c->marks.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; c->marks.push_back({what, e}); }
    c->marks[c->marks_n].first = what;
    (void)hipEventRecord(c->marks[c->marks_n].second, c->stream);
    c->marks_n++;
}
void cml_marks_dump(cmlhip_ctx* c) {          // (call behind a stream synchronise)
    if (c->marks_n < 2) { c->marks_n = 0; return; }
    fprintf(stderr, "  [device timeline, us since '%s']", c->marks[0].first);
    for (size_t i = 1; i < c->marks_n; i++) { float ms = 0; (void)hipEventElapsedTime(&ms, c->marks[0].second, c->marks[i].second); fprintf(stderr, " %s %.1f |", c->marks[i].first, 1e3 * ms); }
    fprintf(stderr, "\n");
    c->marks_n = 0;
}
int cml_ensure(cmlhip_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.bytes >= bytes) return CMLHIP_OK;
    if (b.p) { hipStreamSynchronize(c->stream); (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }
    size_t cap = (bytes + 255) & ~size_t(255);
    CML_CHECK(c, hipMalloc(&b.p, cap));
    b.bytes = cap; b.gen++;
    return CMLHIP_OK;
}
void cml_free(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.bytes = 0;
}
int cml_h2d(cmlhip_ctx* c, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return CMLHIP_OK;
    // The source is a borrowed (usually pageable) host buffer that the caller may free or reuse as soon as we return,
    // so it is copied into a pinned staging ring owned by the context first; the device copy is then truly async.
    const size_t cap = 8u << 20;
    if (!c->pinned) {
        CML_CHECK(c, hipHostMalloc(&c->pinned, cap, hipHostMallocMapped | hipHostMallocCoherent));
        c->pinned_bytes = cap;
        c->pinned_off = 0;
    }
    if (bytes > cap / 2) {                          // big uploads (images): synchronous copy, no staging
        CML_CHECK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        CML_CHECK(c, hipStreamSynchronize(c->stream));
        return CMLHIP_OK;
    }
    if (c->pinned_off + bytes > cap) {              // ring wrap: everything staged so far must have been consumed
        if (c->h2d_scope) {                         // (inside an upload scope: what is staged leaves now, with the kernels waiting for it; the scope stays open)
            const int rc = cml_scope_end(c); if (rc) return rc;
            c->h2d_scope = true; c->h2d_batching = true;
        } else if (c->h2d_batching) { const int rc = cml_h2d_batch_flush(c); if (rc) return rc; c->h2d_batching = true; }
        CML_CHECK(c, hipStreamSynchronize(c->stream));
        c->pinned_off = 0;
        c->h2d_batch_start = 0;
    }
    char* stage = static_cast<char*>(c->pinned) + c->pinned_off;
    memcpy(stage, src, bytes);
    if (c->h2d_batching) {                          // staged only: the flush copies the packed block once
        c->h2d_segs.push_back((unsigned long long)(uintptr_t)dst);
        c->h2d_segs.push_back((unsigned long long)(c->pinned_off - c->h2d_batch_start));
        c->h2d_segs.push_back((unsigned long long)bytes);
        c->pinned_off += (bytes + 255) & ~size_t(255);
        return CMLHIP_OK;
    }
    c->pinned_off += (bytes + 255) & ~size_t(255);
    CML_CHECK(c, hipMemcpyAsync(dst, stage, bytes, hipMemcpyHostToDevice, c->stream));
    return CMLHIP_OK;
}
// Batch mode, direct scatter: `src` lies in pinned, device-mapped host memory that stays untouched until the scatter kernel has run (the window
// shadows): registered as a segment of the packed block by its distance from the block's base — nothing is copied on the host.  Otherwise: cml_h2d.
int cml_h2d_inplace(cmlhip_ctx* c, void* dst, const void* src, size_t bytes) {
    static const bool staged_copies = getenv("CMLHIP_STAGED_COPIES") != nullptr;
    if (bytes == 0) return CMLHIP_OK;
    if (!c->h2d_batching || !c->pinned || staged_copies) return cml_h2d(c, dst, src, bytes);
    const char* base = static_cast<const char*>(c->pinned) + c->h2d_batch_start;
    c->h2d_segs.push_back((unsigned long long)(uintptr_t)dst);
    c->h2d_segs.push_back((unsigned long long)(static_cast<const char*>(src) - base));       // (modulo 2^64: the kernel adds it back to the base)
    c->h2d_segs.push_back((unsigned long long)bytes);
    c->h2d_inplace = true;
    return CMLHIP_OK;
}
// Batch mode only: room for `bytes` in the packed block, registered for `dst` — the caller WRITES its array there instead of building it in
// pageable memory and having cml_h2d copy it (the window upload builds ~0.6 MB of SoA arrays per keyframe).  nullptr: no room / no ring
// yet / not batching — the caller falls back to cml_h2d.
void* cml_h2d_stage(cmlhip_ctx* c, void* dst, size_t bytes) {
    if (!c->h2d_batching || !c->pinned || bytes == 0 || c->pinned_off + bytes > c->pinned_bytes) return nullptr;
    char* stage = static_cast<char*>(c->pinned) + c->pinned_off;
    c->h2d_segs.push_back((unsigned long long)(uintptr_t)dst);
    c->h2d_segs.push_back((unsigned long long)(c->pinned_off - c->h2d_batch_start));
    c->h2d_segs.push_back((unsigned long long)bytes);
    c->pinned_off += (bytes + 255) & ~size_t(255);
    return stage;
}
// one workgroup row per segment: 16-byte words, then the byte tail (block offsets and DevBuf bases are 256-byte aligned)
__global__ void k_h2d_scatter(const unsigned long long* __restrict__ segs, const char* __restrict__ blob) {
    const unsigned long long* S = segs + 3 * (size_t)blockIdx.y;
    char* dst = reinterpret_cast<char*>((uintptr_t)S[0]);
    const bool zero = S[1] >= ~2ull;                  // a fill segment (~0: zeros, ~1: all bits set = -1 as int, ~2: every byte 0x7f)
    const unsigned fillw = S[1] == ~1ull ? 0xffffffffu : (S[1] == ~2ull ? 0x7f7f7f7fu : 0u);
    const char* src = zero ? blob : blob + S[1];
    const size_t bytes = (size_t)S[2], words = bytes / 16;
    const bool aligned = (((uintptr_t)dst) & 15) == 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (zero) {
        if (aligned) {
            for (size_t i = t0; i < words; i += stride) reinterpret_cast<uint4*>(dst)[i] = make_uint4(fillw, fillw, fillw, fillw);
            for (size_t i = words * 16 + t0; i < bytes; i += stride) dst[i] = (char)fillw;
        } else for (size_t i = t0; i < bytes; i += stride) dst[i] = (char)fillw;
    } else if (aligned) {
        for (size_t i = t0; i < words; i += stride) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
        for (size_t i = words * 16 + t0; i < bytes; i += stride) dst[i] = src[i];
    } else {
        for (size_t i = t0; i < bytes; i += stride) dst[i] = src[i];
    }
}
int cml_zero(cmlhip_ctx* c, void* dst, size_t bytes) {
    if (bytes == 0) return CMLHIP_OK;
    if (c->h2d_batching) {
        c->h2d_segs.push_back((unsigned long long)(uintptr_t)dst); c->h2d_segs.push_back(~0ull); c->h2d_segs.push_back((unsigned long long)bytes);
        return CMLHIP_OK;
    }
    CML_CHECK(c, hipMemsetAsync(dst, 0, bytes, c->stream));
    return CMLHIP_OK;
}
int cml_fill_7f(cmlhip_ctx* c, void* dst, size_t bytes) {       // every byte 0x7f (ints: 0x7f7f7f7f, "nobody yet" of the coarse-depth owner map)
    if (bytes == 0) return CMLHIP_OK;
    if (c->h2d_batching) {
        c->h2d_segs.push_back((unsigned long long)(uintptr_t)dst); c->h2d_segs.push_back(~2ull); c->h2d_segs.push_back((unsigned long long)bytes);
        return CMLHIP_OK;
    }
    CML_CHECK(c, hipMemsetAsync(dst, 0x7f, bytes, c->stream));
    return CMLHIP_OK;
}
int cml_fill_ff(cmlhip_ctx* c, void* dst, size_t bytes) {       // every byte 0xff (ints: -1)
    if (bytes == 0) return CMLHIP_OK;
    if (c->h2d_batching) {
        c->h2d_segs.push_back((unsigned long long)(uintptr_t)dst); c->h2d_segs.push_back(~1ull); c->h2d_segs.push_back((unsigned long long)bytes);
        return CMLHIP_OK;
    }
    CML_CHECK(c, hipMemsetAsync(dst, 0xff, bytes, c->stream));
    return CMLHIP_OK;
}
void cml_h2d_batch_begin(cmlhip_ctx* c) {
    if (c->h2d_scope && c->h2d_batching) return;    // the scope's batch is already open
    c->h2d_batching = true; c->h2d_segs.clear(); c->h2d_batch_start = c->pinned ? c->pinned_off : 0;
}
static int h2d_batch_flush_now(cmlhip_ctx* c);
int cml_h2d_batch_flush(cmlhip_ctx* c) {
    if (c->h2d_scope) return CMLHIP_OK;             // leaves when the scope ends
    return h2d_batch_flush_now(c);
}
int cml_scope_end(cmlhip_ctx* c) {
    if (!c->h2d_scope) return CMLHIP_OK;
    c->h2d_scope = false;
    int rc = h2d_batch_flush_now(c);
    std::vector<std::function<int()>> run;
    run.swap(c->deferred);
    for (auto& f : run) { const int r = f(); if (!rc) rc = r; }
    cml_mark(c, "scope");
    return rc;
}
// a call that stages into an open scope has failed half-way: nothing of the scope leaves (a block with half a window in it must not be scattered, and the
// kernels waiting for it must not run); the caller sees the failing call's status
void cml_scope_abort(cmlhip_ctx* c) {
    if (!c->h2d_scope) return;
    c->h2d_scope = false; c->h2d_batching = false; c->h2d_inplace = false;
    c->h2d_segs.clear(); c->deferred.clear();
    c->win.busy_pending = false;
}
extern "C" int cmlhip_upload_scope_begin(cmlhip_ctx* c) {
    if (!c) return CMLHIP_ERR_INVALID;
    (void)hipSetDevice(c->device);
    if (c->h2d_scope) return CMLHIP_OK;
    if (c->pinned && c->pinned_off > c->pinned_bytes / 4) {     // room for a window's worth of arrays without a wrap inside the scope
        CML_CHECK(c, hipStreamSynchronize(c->stream));
        c->pinned_off = 0;
    }
    cml_h2d_batch_begin(c);
    c->h2d_scope = true;
    return CMLHIP_OK;
}
extern "C" int cmlhip_upload_scope_end(cmlhip_ctx* c) {
    if (!c) return CMLHIP_ERR_INVALID;
    (void)hipSetDevice(c->device);
    return cml_scope_end(c);
}
static int h2d_batch_flush_now(cmlhip_ctx* c) {
    c->h2d_batching = false;
    const size_t nseg = c->h2d_segs.size() / 3;
    if (nseg == 0) return CMLHIP_OK;
    const size_t blob = c->pinned_off - c->h2d_batch_start;
    int rc;
    const size_t tbytes = sizeof(unsigned long long) * 3 * nseg;
    const size_t cap = c->pinned_bytes;
    const bool room = c->pinned && c->pinned_off + tbytes <= cap;       // the table rides at the end of the pinned block
    static const bool staged_copies = getenv("CMLHIP_STAGED_COPIES") != nullptr;
    if (!staged_copies && c->pinned) {
        // the scatter kernel pulls the packed block (and the segments registered in place) straight out of pinned, device-mapped host memory: no
        // blit of the block into device memory first — two copies and their dispatch gaps (16 + 4 us and ~10 us of gaps per window upload in
        // the trace) become the kernel's own reads across the link
        const unsigned long long* tab;
        if (room) {
            char* tstage = static_cast<char*>(c->pinned) + c->pinned_off;
            memcpy(tstage, c->h2d_segs.data(), tbytes);
            c->pinned_off += (tbytes + 255) & ~size_t(255);
            tab = reinterpret_cast<const unsigned long long*>(tstage);
        } else {                                        // no room behind the block: the table goes through a plain synchronous copy
            if ((rc = cml_ensure(c, c->h2d_desc, tbytes))) return rc;
            CML_CHECK(c, hipMemcpyAsync(c->h2d_desc.p, c->h2d_segs.data(), tbytes, hipMemcpyHostToDevice, c->stream));
            CML_CHECK(c, hipStreamSynchronize(c->stream));
            tab = c->h2d_desc.as<unsigned long long>();
        }
        k_h2d_scatter<<<dim3(32, (unsigned)nseg), 256, 0, c->stream>>>(tab, static_cast<const char*>(c->pinned) + c->h2d_batch_start);
        CML_CHECK(c, hipGetLastError());
        if (c->win.busy_pending && c->win.busy) CML_CHECK(c, hipEventRecord(c->win.busy, c->stream));      // the window shadows are free again behind this kernel
        c->h2d_segs.clear(); c->h2d_inplace = false;
        return CMLHIP_OK;
    }
    if ((rc = cml_ensure(c, c->h2d_blob, blob + 16))) return rc;
    if ((rc = cml_ensure(c, c->h2d_desc, tbytes))) return rc;
    if (!room) {
        CML_CHECK(c, hipMemcpyAsync(c->h2d_blob.p, static_cast<char*>(c->pinned) + c->h2d_batch_start, blob, hipMemcpyHostToDevice, c->stream));
        CML_CHECK(c, hipMemcpyAsync(c->h2d_desc.p, c->h2d_segs.data(), tbytes, hipMemcpyHostToDevice, c->stream));
        CML_CHECK(c, hipStreamSynchronize(c->stream));
    } else {
        char* tstage = static_cast<char*>(c->pinned) + c->pinned_off;
        memcpy(tstage, c->h2d_segs.data(), tbytes);
        c->pinned_off += (tbytes + 255) & ~size_t(255);
        CML_CHECK(c, hipMemcpyAsync(c->h2d_blob.p, static_cast<char*>(c->pinned) + c->h2d_batch_start, blob, hipMemcpyHostToDevice, c->stream));
        CML_CHECK(c, hipMemcpyAsync(c->h2d_desc.p, tstage, tbytes, hipMemcpyHostToDevice, c->stream));
    }
    k_h2d_scatter<<<dim3(32, (unsigned)nseg), 256, 0, c->stream>>>(c->h2d_desc.as<unsigned long long>(), c->h2d_blob.as<char>());
    CML_CHECK(c, hipGetLastError());
    if (c->win.busy_pending && c->win.busy) CML_CHECK(c, hipEventRecord(c->win.busy, c->stream));
    c->h2d_segs.clear();
    return CMLHIP_OK;
}
int cml_d2h(cmlhip_ctx* c, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return CMLHIP_OK;
    if (c->d2h_batching) {
        const size_t ns = c->d2h_segs.size();                        // the next piece starts behind the last one, 256-byte aligned
        const size_t off = ns ? (size_t)c->d2h_segs[ns - 2] + (((size_t)c->d2h_segs[ns - 1] + 255) & ~size_t(255)) : 0;
        c->d2h_segs.push_back((unsigned long long)(uintptr_t)src); c->d2h_segs.push_back((unsigned long long)off); c->d2h_segs.push_back((unsigned long long)bytes);
        c->d2h_dst.push_back(dst);
        return CMLHIP_OK;
    }
    CML_CHECK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    CML_CHECK(c, hipStreamSynchronize(c->stream));
    return CMLHIP_OK;
}
__global__ void k_d2h_gather(const unsigned long long* __restrict__ segs, char* __restrict__ blob) {
    const unsigned long long* S = segs + 3 * (size_t)blockIdx.y;
    const char* src = reinterpret_cast<const char*>((uintptr_t)S[0]);
    char* dst = blob + S[1];
    const size_t bytes = (size_t)S[2], words = bytes / 16;
    const bool aligned = (((uintptr_t)src) & 15) == 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (aligned) {
        for (size_t i = t0; i < words; i += stride) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
        for (size_t i = words * 16 + t0; i < bytes; i += stride) dst[i] = src[i];
    } else for (size_t i = t0; i < bytes; i += stride) dst[i] = src[i];
}
struct D2hSegs { unsigned long long s[3 * 40]; };          // up to 40 pieces per readback ride in the kernel arguments (960 bytes)
__global__ void k_d2h_gather_direct(D2hSegs T, char* __restrict__ blob) {
    const unsigned long long* S = T.s + 3 * (size_t)blockIdx.y;
    const char* src = reinterpret_cast<const char*>((uintptr_t)S[0]);
    char* dst = blob + S[1];
    const size_t bytes = (size_t)S[2], words = bytes / 16;
    const bool aligned = (((uintptr_t)src) & 15) == 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (aligned) {
        for (size_t i = t0; i < words; i += stride) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
        for (size_t i = words * 16 + t0; i < bytes; i += stride) dst[i] = src[i];
    } else for (size_t i = t0; i < bytes; i += stride) dst[i] = src[i];
}
void cml_d2h_batch_begin(cmlhip_ctx* c) { c->d2h_batching = true; c->d2h_segs.clear(); c->d2h_dst.clear(); }
int cml_d2h_batch_flush(cmlhip_ctx* c) {
    c->d2h_batching = false;
    const size_t nseg = c->d2h_dst.size();
    if (nseg == 0) return CMLHIP_OK;
    const size_t total = (size_t)c->d2h_segs[3 * (nseg - 1) + 1] + (((size_t)c->d2h_segs[3 * (nseg - 1) + 2] + 255) & ~size_t(255));
    int rc;
    if ((rc = cml_ensure(c, c->d2h_blob, total))) return rc;
    if ((rc = cml_ensure(c, c->h2d_desc, sizeof(unsigned long long) * 3 * nseg))) return rc;
    if (c->pinned_d2h_bytes < total) {
        if (c->pinned_d2h) (void)hipHostFree(c->pinned_d2h);
        c->pinned_d2h = nullptr; c->pinned_d2h_bytes = 0;
        const size_t cap = std::max(total, (size_t)1 << 20);
        CML_CHECK(c, hipHostMalloc(&c->pinned_d2h, cap, hipHostMallocMapped | hipHostMallocCoherent));
        c->pinned_d2h_bytes = cap;
    }
    static const bool staged_copies = getenv("CMLHIP_STAGED_COPIES") != nullptr;
    if (!staged_copies && sizeof(unsigned long long) * 3 * nseg <= sizeof(D2hSegs)) {
        // the gather kernel takes its table in the kernel-argument segment and writes the pieces straight into the pinned (device-mapped, coherent)
        // host block: no table copy ahead of it, no device-to-host blit behind it
        D2hSegs T;
        memcpy(T.s, c->d2h_segs.data(), sizeof(unsigned long long) * 3 * nseg);
        k_d2h_gather_direct<<<dim3(16, (unsigned)nseg), 256, 0, c->stream>>>(T, static_cast<char*>(c->pinned_d2h));
        CML_CHECK(c, hipGetLastError());
        CML_CHECK(c, hipStreamSynchronize(c->stream));          // (polling an event behind the kernel instead was measured: 12 us SLOWER per run())
    } else {
        if ((rc = cml_h2d(c, c->h2d_desc.p, c->d2h_segs.data(), sizeof(unsigned long long) * 3 * nseg))) return rc;
        k_d2h_gather<<<dim3(16, (unsigned)nseg), 256, 0, c->stream>>>(c->h2d_desc.as<unsigned long long>(), c->d2h_blob.as<char>());
        CML_CHECK(c, hipMemcpyAsync(c->pinned_d2h, c->d2h_blob.p, total, hipMemcpyDeviceToHost, c->stream));
        CML_CHECK(c, hipStreamSynchronize(c->stream));
    }
    for (size_t k = 0; k < nseg; k++) memcpy(c->d2h_dst[k], static_cast<char*>(c->pinned_d2h) + c->d2h_segs[3 * k + 1], (size_t)c->d2h_segs[3 * k + 2]);
    c->d2h_segs.clear(); c->d2h_dst.clear();
    return CMLHIP_OK;
}
// a pyramid that is still being built by the image worker: wait until everything is ENQUEUED on the worker's stream (host side), then order
// the context's stream behind its last kernel (device side: no host wait for the build itself)
static int pyr_settle(cmlhip_ctx* c, uint64_t id, Pyramid& P) {
    if (!P.pending) return CMLHIP_OK;
    int st;
    {
        std::unique_lock<std::mutex> lk(c->pyr_mu);
        c->pyr_cv.wait(lk, [&] { auto it = c->pyr_state.find(id); return it == c->pyr_state.end() || it->second != 1; });
        auto it = c->pyr_state.find(id);
        st = it == c->pyr_state.end() ? 2 : it->second;
        if (it != c->pyr_state.end()) c->pyr_state.erase(it);
    }
    P.pending = false;
    if (st < 0) {                                           // sticky: every later look-up of this id fails too (the levels were never filled)
        P.failed = true;
        if (c->pyr_stream) (void)hipStreamSynchronize(c->pyr_stream);      // whatever the worker did enqueue has left the blocks before they can be released
        c->err = "cmlhip_pyramid_build_async: the image worker failed to copy / build the pyramid";
        return CMLHIP_ERR_HIP;
    }
    CML_CHECK(c, hipStreamWaitEvent(c->stream, P.ready, 0));
    return CMLHIP_OK;
}
const Pyramid* cml_find_pyr(cmlhip_ctx* c, uint64_t id) {
    auto it = c->pyr.find(id);
    if (it == c->pyr.end()) return nullptr;
    if (it->second.pending && pyr_settle(c, id, it->second)) return nullptr;
    if (it->second.failed) { c->err = "the pyramid of this image id was never built (cmlhip_pyramid_build_async failed): drop the id and build it again"; return nullptr; }
    return &it->second;
}

static void pyr_worker_main(cmlhip_ctx* c);

extern "C" {

int cmlhip_abi_version(void) { return CMLHIP_ABI_VERSION; }

int cmlhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int cmlhip_create(cmlhip_ctx** out, const cmlhip_limits* lim) {
    if (!out || !lim) return CMLHIP_ERR_INVALID;
    *out = nullptr;
    if (lim->max_frames < 1 || lim->max_frames > CMLHIP_MAX_FRAMES) return CMLHIP_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || lim->device_id >= n) return CMLHIP_ERR_HIP;   // no CPU fallback
    if (hipSetDevice(lim->device_id) != hipSuccess) return CMLHIP_ERR_HIP;
    cmlhip_ctx* c = new cmlhip_ctx();
    c->lim = *lim;
    c->device = lim->device_id;
    c->rs_ok = getenv("CMLHIP_NO_RS") == nullptr;        // development switch: resident loop on the record-writing residual kernel
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return CMLHIP_ERR_HIP; }
    (void)hipEventCreate(&c->ev[0]);
    (void)hipEventCreate(&c->ev[1]);
    // the image worker's stream and thread (cmlhip_pyramid_build_async) are set up HERE: creating a stream takes 4-9 ms on this stack,
    // which belongs to the start of a sequence, not in front of its second frame; the thread sleeps on a condition variable until used
    if (hipStreamCreateWithFlags(&c->pyr_stream, hipStreamNonBlocking) == hipSuccess) {
        c->pyr_thread = std::thread(pyr_worker_main, c);
        c->pyr_started = true;
    }
    *out = c;
    return CMLHIP_OK;
}

void cmlhip_destroy(cmlhip_ctx* c) {
    if (!c) return;
    if (c->h2d_scope) { (void)hipSetDevice(c->device); (void)cml_scope_end(c); }
    (void)hipSetDevice(c->device);
    if (c->pyr_started) {                                    // the image worker: finish what is queued, then leave
        { std::lock_guard<std::mutex> lk(c->pyr_mu); c->pyr_quit = true; }
        c->pyr_cv.notify_all();
        c->pyr_thread.join();
        (void)hipStreamSynchronize(c->pyr_stream);
        (void)hipStreamDestroy(c->pyr_stream);
    }
    (void)hipStreamSynchronize(c->stream);
    for (auto& kv : c->pyr) if (kv.second.ready) (void)hipEventDestroy(kv.second.ready);
    for (auto& kv : c->pyr)
        for (int l = 0; l < 8; l++) { if (kv.second.lv[l].grad) (void)hipFree(kv.second.lv[l].grad); if (kv.second.lv[l].gray) (void)hipFree(kv.second.lv[l].gray); if (kv.second.lv[l].tiled) (void)hipFree(kv.second.lv[l].tiled); }
    for (auto& kv : c->img_pool) (void)hipFree(kv.second);
    c->img_pool.clear();
    DevBuf* all[] = {&c->img_tmp, &c->h2d_blob, &c->h2d_desc, &c->d2h_blob, &c->frames, &c->pairs, &c->pt_x, &c->pt_y, &c->pt_idepth, &c->pt_idepth_zero, &c->pt_prior, &c->pt_host,
                     &c->pt_colors, &c->pt_weights, &c->pt_backup, &c->pt_acc, &c->pt_step, &c->r_point, &c->r_host, &c->r_target,
                     &c->r_state, &c->r_new_state, &c->r_energy, &c->r_new_energy, &c->r_new_energy_wo, &c->r_ret_energy,
                     &c->r_good, &c->r_lin, &c->r_sel, &c->r_dead, &c->r_center, &c->r_jpjdf, &c->r_rtz, &c->rj[0], &c->rj[1],
                     &c->trk_hyp, &c->trk_opt_out, &c->rs_tiles, &c->rs_tile_off, &c->rs_part, &c->r_idepth, &c->point_res, &c->r_px, &c->r_py, &c->r_colors, &c->r_weights, &c->by_point_off, &c->by_point, &c->by_pair_off, &c->by_pair, &c->pair_code, &c->pair_pos, &c->point_code, &c->point_tgt, &c->point_pos, &c->frame_state, &c->pre_w2c, &c->null_basis, &c->pt_mask, &c->marg_scratch, &c->tr_points, &c->tr_pairs, &c->tr_out, &c->tr_resident, &c->tr_counts, &c->tr_resident2, &c->tr_edit, &c->tr_state, &c->tr_hosts, &c->tr_journal, &c->trk_pose0, &c->ini_points, &c->ini_partial, &c->pnp_matches, &c->pnp_flags, &c->pnp_out, &c->lba_frames, &c->lba_cams, &c->lba_points, &c->lba_off, &c->lba_edges, &c->lba_err, &c->lba_flags, &c->lba_work, &c->newframe_res, &c->acc_pair[0],
                     &c->acc_pair[1], &c->acc_num[0], &c->acc_num[1], &c->pair_blocks, &c->adH, &c->adT, &c->adHTd,
                     &c->vec_small, &c->HA, &c->bA, &c->HL, &c->bL, &c->Hsc, &c->bsc, &c->HM, &c->bM, &c->xvec, &c->G,
                     &c->syrk_part, &c->solve_image, &c->xad, &c->scal, &c->lin_partial, &c->trk_warped, &c->trk_partial, &c->trk_out, &c->cd_cnt, &c->cd_pts,
                     &c->Hf, &c->bf, &c->step_partial, &c->rp_obs, &c->rp_poses, &c->rp_points, &c->rp_M, &c->rp_b, &c->rp_Jp, &c->rp_used, &c->rp_x, &c->rp_off, &c->rp_orig,
                     &c->rr_obs, &c->rr_off, &c->rr_orig, &c->rr_points, &c->rr_jp, &c->rr_used, &c->rr_x, &c->rr_ready, &c->trk_xch, &c->x_ticket, &c->batch_main, &c->batch_rs,
                     &c->trk_early, &c->rr_scratch, &c->run_pack, &c->run_snap, &c->c_point, &c->c_target, &c->c_state, &c->c_lin, &c->c_dev_of, &c->c_bpos};
    for (DevBuf* b : all) cml_free(*b);
    for (int l = 0; l < 8; l++) { cml_free(c->trk_ref[l]); cml_free(c->cd_idepth[l]); cml_free(c->cd_wsum[l]); cml_free(c->cd_wbak[l]); }
    cml_window_free(c);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pinned_d2h) (void)hipHostFree(c->pinned_d2h);
    if (c->trk_host) (void)hipHostFree(c->trk_host);
    if (c->trk_opt_host) (void)hipHostFree(c->trk_opt_host);
    if (c->done_word) (void)hipHostFree(c->done_word);
    if (c->tr_host) (void)hipHostFree(c->tr_host);
    for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
    (void)hipEventDestroy(c->ev[0]);
    (void)hipEventDestroy(c->ev[1]);
    if (c->batch_ev) (void)hipEventDestroy(c->batch_ev);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

int cmlhip_set_device_share(cmlhip_ctx* c, int n_contexts) {
    if (!c || n_contexts < 1) return CMLHIP_ERR_INVALID;
    c->device_share = n_contexts;
    return CMLHIP_OK;
}
const char* cmlhip_last_error(const cmlhip_ctx* c) { return c ? c->err.c_str() : "null context"; }

int cmlhip_synchronize(cmlhip_ctx* c) { CML_DEV(c);
    if (!c) return CMLHIP_ERR_INVALID;
    CML_CHECK(c, hipStreamSynchronize(c->stream));
    return CMLHIP_OK;
}
void* cmlhip_stream(cmlhip_ctx* c) { return c ? (void*)c->stream : nullptr; }

int cmlhip_event_mark(cmlhip_ctx* c, int which) { CML_DEV(c);
    if (!c || which < 0 || which > 1) return CMLHIP_ERR_INVALID;
    CML_CHECK(c, hipEventRecord(c->ev[which], c->stream));
    return CMLHIP_OK;
}
int cmlhip_profile_next_launch(cmlhip_ctx* c) { CML_DEV(c);
    if (!c) return CMLHIP_ERR_INVALID;
    c->ext_start = c->ev[0]; c->ext_stop = c->ev[1];          // consumed by the next instrumented dispatch (CML_LAUNCH_EV)
    return CMLHIP_OK;
}
int cmlhip_event_elapsed_ms(cmlhip_ctx* c, float* ms) { CML_DEV(c);
    if (!c || !ms) return CMLHIP_ERR_INVALID;
    CML_CHECK(c, hipEventSynchronize(c->ev[1]));
    CML_CHECK(c, hipEventElapsedTime(ms, c->ev[0], c->ev[1]));
    return CMLHIP_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ image kernels
template <bool HALF>
__device__ __forceinline__ void store_texel(void* img, size_t idx, float I, float dx, float dy) {
    if (HALF) {
        __half2 a = __floats2half2_rn(I, dx), b = __floats2half2_rn(dy, 0.f);
        uint2 v;
        v.x = *reinterpret_cast<unsigned*>(&a);
        v.y = *reinterpret_cast<unsigned*>(&b);
        reinterpret_cast<uint2*>(img)[idx] = v;
    } else {
        reinterpret_cast<float4*>(img)[idx] = make_float4(I, dx, dy, 0.f);
    }
}

// AoS3 host layout -> device texels
template <bool HALF>
__global__ void k_expand_aos3(const float* __restrict__ aos3, void* __restrict__ img, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    store_texel<HALF>(img, i, aos3[3 * (size_t)i], aos3[3 * (size_t)i + 1], aos3[3 * (size_t)i + 2]);
}

// Array2D::reduceByTwo, Array2D.h:388-401
__global__ void k_reduce_by_two(const float* __restrict__ in, int w, int h, float* __restrict__ out) {
    int nw = w / 2, nh = h / 2;
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= nw || y >= nh) return;
    const float* r0 = in + (size_t)(2 * y) * w + 2 * x;
    const float* r1 = r0 + w;
    out[(size_t)y * nw + x] = (((r0[0] + r0[1]) + r1[0]) + r1[1]) / 4.0f;
}

// Array2D::gradientImage, Array2D.h:288-327 (zero 1-px border)
template <bool HALF>
__global__ void k_gradient(const float* __restrict__ g, int w, int h, void* __restrict__ img) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    size_t i = (size_t)y * w + x;
    if (x < 1 || y < 1 || x >= w - 1 || y >= h - 1) { store_texel<HALF>(img, i, 0.f, 0.f, 0.f); return; }
    store_texel<HALF>(img, i, g[i], (g[i + 1] - g[i - 1]) * 0.5f, (g[i + w] - g[i - w]) * 0.5f);
}

template <bool HALF>
__global__ void k_collapse_aos3(const void* __restrict__ img, float* __restrict__ aos3, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float I, dx, dy;
    if (HALF) {
        uint2 v = reinterpret_cast<const uint2*>(img)[i];
        __half2 a = *reinterpret_cast<__half2*>(&v.x), b = *reinterpret_cast<__half2*>(&v.y);
        I = __low2float(a); dx = __high2float(a); dy = __low2float(b);
    } else {
        float4 v = reinterpret_cast<const float4*>(img)[i];
        I = v.x; dx = v.y; dy = v.z;
    }
    aos3[3 * (size_t)i] = I; aos3[3 * (size_t)i + 1] = dx; aos3[3 * (size_t)i + 2] = dy;
}

static size_t texel_bytes(const cmlhip_ctx* c) { return c->lim.texel_format == CMLHIP_TEXEL_F16 ? 8 : 16; }

// Image levels come from and go back to a per-context pool keyed by byte size: a sequence allocates its pyramid levels once and
// every later frame (same sizes) reuses them.  The pool is capped; beyond the cap a released level really is freed.
static const size_t IMG_POOL_CAP = (size_t)8 << 30;
static int pool_alloc(cmlhip_ctx* c, size_t bytes, void** out) {
    auto it = c->img_pool.find(bytes);
    if (it != c->img_pool.end()) { *out = it->second; c->img_pool.erase(it); c->img_pool_bytes -= bytes; return CMLHIP_OK; }
    CML_CHECK(c, hipMalloc(out, bytes));
    return CMLHIP_OK;
}
static void pool_release(cmlhip_ctx* c, void* p, size_t bytes) {
    if (!p) return;
    static const bool off = getenv("CMLHIP_NO_IMAGE_POOL") != nullptr;      // measurement switch: free and allocate per frame
    if (off || c->img_pool_bytes + bytes > IMG_POOL_CAP) { (void)hipFree(p); return; }
    c->img_pool.emplace(bytes, p); c->img_pool_bytes += bytes;
}
// A BA window holds raw pointers into the level-0 images it names (FrameDev::grad0 / grad0t): when such an image is rebuilt with
// another size, rebuilt from gray or dropped, its blocks go back to the pool and the window must not be used again before the next
// cmlhip_ba_upload_window (the entry points then fail with CMLHIP_ERR_STATE instead of reading recycled memory)
static void window_forget_image(cmlhip_ctx* c, uint64_t id) {
    for (uint64_t u : c->ba_image_ids)
        if (u == id) { c->ba_uploaded = false; c->resident_on = false; c->ba_image_ids.clear(); return; }
}
static void free_level(cmlhip_ctx* c, PyrLevel& L) {
    const size_t n = (size_t)L.w * L.h;
    pool_release(c, L.grad, n * texel_bytes(c));
    pool_release(c, L.gray, n * sizeof(float));
    pool_release(c, L.tiled, cml_tiled_bytes(L.w, L.h));
    L = PyrLevel();
}

// one thread per tile row (32 bytes): texels x0 .. x0+4 of row y (clamped at the image border) as 5 x {I, dI/dx, dI/dy} halves
__global__ void k_tile_level0_f16(const uint2* __restrict__ img, int w, int h, int tw, int th, uint4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tw * th * CML_TILE_H) return;
    const int r = i & (CML_TILE_H - 1), t = i / CML_TILE_H, tx = t % tw, ty = t / tw;
    const int y = min(CML_TILE_H * ty + r, h - 1);
    unsigned short hv[16];
#pragma unroll
    for (int cidx = 0; cidx < 5; cidx++) {
        const uint2 v = img[(size_t)y * w + min(CML_TILE_W * tx + cidx, w - 1)];       // {I | dx << 16, dy | 0}
        hv[3 * cidx] = (unsigned short)(v.x & 0xffffu); hv[3 * cidx + 1] = (unsigned short)(v.x >> 16); hv[3 * cidx + 2] = (unsigned short)(v.y & 0xffffu);
    }
    hv[15] = 0;
    uint4 o0, o1;
    o0.x = hv[0] | ((unsigned)hv[1] << 16); o0.y = hv[2] | ((unsigned)hv[3] << 16); o0.z = hv[4] | ((unsigned)hv[5] << 16); o0.w = hv[6] | ((unsigned)hv[7] << 16);
    o1.x = hv[8] | ((unsigned)hv[9] << 16); o1.y = hv[10] | ((unsigned)hv[11] << 16); o1.z = hv[12] | ((unsigned)hv[13] << 16); o1.w = hv[14] | ((unsigned)hv[15] << 16);
    out[2 * (size_t)i] = o0; out[2 * (size_t)i + 1] = o1;
}

int cml_tiled_level0(cmlhip_ctx* c, uint64_t id, const void** out) {
    auto it = c->pyr.find(id);
    if (it != c->pyr.end() && it->second.pending && pyr_settle(c, id, it->second)) return CMLHIP_ERR_HIP;
    if (it == c->pyr.end() || !it->second.lv[0].grad || c->lim.texel_format != CMLHIP_TEXEL_F16) return CMLHIP_ERR_NOT_FOUND;
    PyrLevel& L = it->second.lv[0];
    if (!L.tiled) { if (int rc = pool_alloc(c, cml_tiled_bytes(L.w, L.h), &L.tiled)) return rc; L.tiled_valid = false; }
    if (!L.tiled_valid) {
        const int tw = (L.w + CML_TILE_W - 1) / CML_TILE_W, th = (L.h + CML_TILE_H - 1) / CML_TILE_H, nthr = tw * th * CML_TILE_H;
        k_tile_level0_f16<<<cml_div_up(nthr, 256), 256, 0, c->stream>>>(reinterpret_cast<const uint2*>(L.grad), L.w, L.h, tw, th, reinterpret_cast<uint4*>(L.tiled));
        CML_CHECK(c, hipGetLastError());
        L.tiled_valid = true;
    }
    *out = L.tiled;
    return CMLHIP_OK;
}

extern "C" {

int cmlhip_pyramid_put(cmlhip_ctx* c, uint64_t id, int level, const float* aos3, int w, int h) { CML_DEV(c);
    if (!c || !aos3 || level < 0 || level >= 8 || w <= 0 || h <= 0) return CMLHIP_ERR_INVALID;
    Pyramid& P = c->pyr[id];
    if (P.pending) (void)pyr_settle(c, id, P);              // (an image the worker is still building: ordered ahead of the put)
    PyrLevel& L = P.lv[level];
    (void)hipStreamSynchronize(c->stream);
    if (L.w != w || L.h != h) { if (level == 0 && L.grad) window_forget_image(c, id); free_level(c, L); }
    L.tiled_valid = false;
    size_t n = (size_t)w * h;
    int rc;
    if (!L.grad && (rc = pool_alloc(c, n * texel_bytes(c), &L.grad))) return rc;
    L.w = w; L.h = h;
    if (level + 1 > P.levels) P.levels = level + 1;
    if ((rc = cml_ensure(c, c->img_tmp, n * 3 * sizeof(float)))) return rc;
    float* tmp = c->img_tmp.as<float>();
    if ((rc = cml_h2d(c, tmp, aos3, n * 3 * sizeof(float)))) return rc;
    int blocks = cml_div_up((int)n, 256);
    if (c->lim.texel_format == CMLHIP_TEXEL_F16) k_expand_aos3<true><<<blocks, 256, 0, c->stream>>>(tmp, L.grad, (int)n);
    else k_expand_aos3<false><<<blocks, 256, 0, c->stream>>>(tmp, L.grad, (int)n);
    CML_CHECK(c, hipGetLastError());
    if (level == 0 && L.tiled) {                                        // a window may already hold the pointer of the tiled copy: keep it in step
        const void* t_ = nullptr;
        if (int rc_t = cml_tiled_level0(c, id, &t_)) return rc_t;
    }
    return CMLHIP_OK;                                                   // (the staging buffer is reused in stream order)
}

int cmlhip_pyramid_build(cmlhip_ctx* c, uint64_t id, const float* gray, int w, int h, int levels) { CML_DEV(c);
    if (!c || !gray || levels < 1 || levels > 8 || w <= 0 || h <= 0) return CMLHIP_ERR_INVALID;
    Pyramid& P = c->pyr[id];
    if (P.pending) (void)pyr_settle(c, id, P);
    (void)hipStreamSynchronize(c->stream);
    if (P.lv[0].grad) window_forget_image(c, id);
    for (int l = 0; l < 8; l++) free_level(c, P.lv[l]);
    P.failed = false;                                        // a synchronous rebuild of an id whose asynchronous build failed repairs the entry
    P.levels = levels;
    int cw = w, ch = h;
    for (int l = 0; l < levels; l++) {
        PyrLevel& L = P.lv[l];
        L.w = cw; L.h = ch;
        size_t n = (size_t)cw * ch;
        int rc;
        if ((rc = pool_alloc(c, n * sizeof(float), (void**)&L.gray))) return rc;
        if ((rc = pool_alloc(c, n * texel_bytes(c), &L.grad))) return rc;
        if (l == 0) {
            if ((rc = cml_h2d(c, L.gray, gray, n * sizeof(float)))) return rc;
        } else {
            const PyrLevel& U = P.lv[l - 1];
            dim3 g(cml_div_up(cw, 256), ch);
            k_reduce_by_two<<<g, 256, 0, c->stream>>>(U.gray, U.w, U.h, L.gray);
        }
        dim3 g(cml_div_up(cw, 256), ch);
        if (c->lim.texel_format == CMLHIP_TEXEL_F16) k_gradient<true><<<g, 256, 0, c->stream>>>(L.gray, cw, ch, L.grad);
        else k_gradient<false><<<g, 256, 0, c->stream>>>(L.gray, cw, ch, L.grad);
        cw /= 2; ch /= 2;      // == (int)(w / 2^l) of CaptureImage.cpp:39-78 (floor of a floor)
        if (cw < 1 || ch < 1) { P.levels = l + 1; break; }
    }
    CML_CHECK(c, hipGetLastError());
    return CMLHIP_OK;
}

// ---- cmlhip_pyramid_build_async.  The caller's thread allocates the levels (pool + cache map are its own) and hands the rest to the
// context's image worker: the copy of the (pageable) image to the device and the reduce / gradient kernels of every level on the
// worker's stream.  Neither the staging of the image (~0.1 ms on a host core) nor the transfer nor the kernels sit on the caller's thread or on the
// context's stream; the first call that names the image (cml_find_pyr) orders the stream behind the build.
static void pyr_worker_main(cmlhip_ctx* c) {
    (void)hipSetDevice(c->device);
    for (;;) {
        PyrJob J;
        {
            std::unique_lock<std::mutex> lk(c->pyr_mu);
            c->pyr_cv.wait(lk, [&] { return c->pyr_quit || !c->pyr_jobs.empty(); });
            if (c->pyr_jobs.empty()) return;                 // (quit with nothing left)
            J = c->pyr_jobs.front(); c->pyr_jobs.pop_front();
        }
        // (the caller's image is pageable memory: the runtime stages it through its own pinned pool and this call returns when the source
        //  has been read — on this thread that is exactly what is wanted, and the context needs no staging buffers of its own: two
        //  hipHostMalloc of an image cost 4-10 ms at the first call)
        const size_t bytes = (size_t)J.w[0] * J.h[0] * sizeof(float);
        bool ok = hipMemcpyAsync(J.gray[0], J.src, bytes, hipMemcpyHostToDevice, c->pyr_stream) == hipSuccess;
        for (int l = 0; ok && l < J.levels; l++) {
            const dim3 g(cml_div_up(J.w[l], 256), J.h[l]);
            if (l > 0) k_reduce_by_two<<<g, 256, 0, c->pyr_stream>>>(J.gray[l - 1], J.w[l - 1], J.h[l - 1], J.gray[l]);
            if (c->lim.texel_format == CMLHIP_TEXEL_F16) k_gradient<true><<<g, 256, 0, c->pyr_stream>>>(J.gray[l], J.w[l], J.h[l], J.grad[l]);
            else k_gradient<false><<<g, 256, 0, c->pyr_stream>>>(J.gray[l], J.w[l], J.h[l], J.grad[l]);
        }
        ok = ok && hipGetLastError() == hipSuccess && hipEventRecord(J.ready, c->pyr_stream) == hipSuccess;
        { std::lock_guard<std::mutex> lk(c->pyr_mu); c->pyr_state[J.id] = ok ? 2 : -1; }
        c->pyr_cv.notify_all();
    }
}

int cmlhip_pyramid_build_async(cmlhip_ctx* c, uint64_t id, const float* gray, int w, int h, int levels) { CML_DEV(c);
    if (!c || !gray || levels < 1 || levels > 8 || w <= 0 || h <= 0) return CMLHIP_ERR_INVALID;
    if (c->pyr.find(id) != c->pyr.end()) return cmlhip_pyramid_build(c, id, gray, w, h, levels);     // an id that is in use: the synchronous path (its blocks may be in flight)
    if (!c->pyr_started) return cmlhip_pyramid_build(c, id, gray, w, h, levels);      // (no worker: its stream could not be created)
    Pyramid& P = c->pyr[id];
    PyrJob J;
    memset(&J, 0, sizeof J);
    J.id = id; J.src = gray;
    int cw = w, ch = h, rc = 0;
    P.levels = levels;
    for (int l = 0; l < levels; l++) {
        PyrLevel& L = P.lv[l];
        L.w = cw; L.h = ch;
        const size_t n = (size_t)cw * ch;
        if ((rc = pool_alloc(c, n * sizeof(float), (void**)&L.gray)) || (rc = pool_alloc(c, n * texel_bytes(c), &L.grad))) break;
        J.w[l] = cw; J.h[l] = ch; J.gray[l] = L.gray; J.grad[l] = L.grad;
        cw /= 2; ch /= 2;
        if (cw < 1 || ch < 1) { P.levels = l + 1; break; }
    }
    if (rc) { for (int l = 0; l < 8; l++) free_level(c, P.lv[l]); c->pyr.erase(id); return rc; }
    J.levels = P.levels;
    if (!P.ready && hipEventCreateWithFlags(&P.ready, hipEventDisableTiming) != hipSuccess) {      // no entry with unfilled levels is left behind
        for (int l = 0; l < 8; l++) free_level(c, P.lv[l]);
        c->pyr.erase(id);
        c->err = "cmlhip_pyramid_build_async: hipEventCreateWithFlags failed";
        return CMLHIP_ERR_HIP;
    }
    J.ready = P.ready;
    P.pending = true;
    { std::lock_guard<std::mutex> lk(c->pyr_mu); c->pyr_state[id] = 1; c->pyr_jobs.push_back(J); }
    c->pyr_cv.notify_all();
    return CMLHIP_OK;
}

int cmlhip_pyramid_drop(cmlhip_ctx* c, uint64_t id) { CML_DEV(c);
    if (!c) return CMLHIP_ERR_INVALID;
    auto it = c->pyr.find(id);
    if (it == c->pyr.end()) return CMLHIP_ERR_NOT_FOUND;
    if (it->second.pending) (void)pyr_settle(c, id, it->second);
    (void)hipStreamSynchronize(c->stream);
    window_forget_image(c, id);
    for (int l = 0; l < 8; l++) free_level(c, it->second.lv[l]);
    if (it->second.ready) (void)hipEventDestroy(it->second.ready);
    c->pyr.erase(it);
    return CMLHIP_OK;
}

int cmlhip_pyramid_level_size(cmlhip_ctx* c, uint64_t id, int level, int* w, int* h) { CML_DEV(c);
    if (!c) return CMLHIP_ERR_INVALID;
    const Pyramid* P = cml_find_pyr(c, id);
    if (!P || level < 0 || level >= P->levels || !P->lv[level].grad) return CMLHIP_ERR_NOT_FOUND;
    if (w) *w = P->lv[level].w;
    if (h) *h = P->lv[level].h;
    return CMLHIP_OK;
}

int cmlhip_pyramid_get(cmlhip_ctx* c, uint64_t id, int level, float* out) { CML_DEV(c);
    if (!c || !out) return CMLHIP_ERR_INVALID;
    const Pyramid* P = cml_find_pyr(c, id);
    if (!P || level < 0 || level >= P->levels || !P->lv[level].grad) return CMLHIP_ERR_NOT_FOUND;
    const PyrLevel& L = P->lv[level];
    size_t n = (size_t)L.w * L.h;
    int rc;
    if ((rc = cml_ensure(c, c->img_tmp, n * 3 * sizeof(float)))) return rc;
    float* tmp = c->img_tmp.as<float>();
    int blocks = cml_div_up((int)n, 256);
    if (c->lim.texel_format == CMLHIP_TEXEL_F16) k_collapse_aos3<true><<<blocks, 256, 0, c->stream>>>(L.grad, tmp, (int)n);
    else k_collapse_aos3<false><<<blocks, 256, 0, c->stream>>>(L.grad, tmp, (int)n);
    return cml_d2h(c, out, tmp, n * 3 * sizeof(float));
}

}  // extern "C"
