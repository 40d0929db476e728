// api.hip — the C-ABI of include/ggr_raster.h: orchestration of the gfx950 kernels.
//
// Replaces `RasterizeGaussiansCUDA` / `RasterizeGaussiansBackwardCUDA` of the extension GGRt
// imports at reference ggrt/model/pixelsplat/decoder/cuda_splatting.py:6-9.
// No global mutable state: the only statics are thread_local — an error string and, per device, one pinned word +
// one event for the num_rendered read-back, a side stream — handed back to a process-wide pool when the thread ends (below).
#include "../../include/ggr_raster.h"
#include "ggr_common.h"
#include <algorithm>
#include <mutex>
#include <vector>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

namespace {

thread_local char g_err[512] = "";
const char kSpinFault[] = "depth sort: a look-back spin hit its bound (GPU preempted or halted?); frame not rendered";
const char kRangeFault[] = "depth sort: a sort key beyond 30 bits reached the sort (keys must come from preprocess_fwd); frame not rendered";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(GGR_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define KCHECK(dbg, s, what)                                                                    \
    do {                                                                                        \
        hipError_t e_ = hipGetLastError();                                                      \
        if (e_ == hipSuccess && (dbg)) e_ = hipStreamSynchronize(s);                            \
        if (e_ != hipSuccess) return fail(GGR_E_HIP, "%s: %s", what, hipGetErrorString(e_));    \
    } while (0)

// Optional per-stage HIP-event timing (profiling mode only: stage_ms != NULL).  mark(stage): the time since the previous
// mark belongs to `stage` (the forward's stages are not walked in index order: the per-tile depth sort sits behind the scatter).
struct StageTimer {
    hipStream_t s;
    float* out;
    int n;
    hipEvent_t ev[24];
    int stage_of[24];
    int used = 0;
    StageTimer(hipStream_t s_, float* out_, int n_) : s(s_), out(out_), n(n_) {
        if (out) for (int i = 0; i < 24; i++) (void)hipEventCreate(&ev[i]);
        mark(-1);
    }
    void mark(int stage) {
        if (out && used < 24) { (void)hipEventRecord(ev[used], s); stage_of[used] = stage; used++; }
    }
    void finish() {
        if (!out) return;
        (void)hipStreamSynchronize(s);
        for (int i = 1; i < used; i++) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[i - 1], ev[i]);
            if (stage_of[i] >= 0 && stage_of[i] < n) out[stage_of[i]] += ms;
        }
        for (int i = 0; i < 24; i++) (void)hipEventDestroy(ev[i]);
        out = nullptr;
    }
    ~StageTimer() { finish(); }
};

int validate(const GgrSettings* st, const GgrForwardIn* in) {
    if (!st || !in) return fail(GGR_E_INVALID, "null settings / inputs");
    if (st->num_points < 0 || st->image_width < 0 || st->image_height < 0)
        return fail(GGR_E_INVALID, "negative size");
    if ((in->shs == nullptr) == (in->colors_precomp == nullptr) && st->num_points > 0)
        return fail(GGR_E_INVALID, "Please provide excatly one of either SHs or precomputed colors!");
    const bool has_sr = in->scales != nullptr && in->rotations != nullptr;
    const bool any_sr = in->scales != nullptr || in->rotations != nullptr;
    if (st->num_points > 0 && ((!has_sr && in->cov3D_precomp == nullptr) || (any_sr && in->cov3D_precomp != nullptr)))
        return fail(GGR_E_INVALID,
                    "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    if (st->sh_max_degree != 0 && st->sh_max_degree != 3 && st->sh_max_degree != 4)
        return fail(GGR_E_INVALID, "sh_max_degree must be 0 (default), 3 or 4");
    if ((st->depth_sort & ~GGR_DEPTH_SORT_NO_BUCKETS) < GGR_DEPTH_SORT_AUTO || (st->depth_sort & ~GGR_DEPTH_SORT_NO_BUCKETS) > GGR_DEPTH_SORT_PER_TILE)
        return fail(GGR_E_INVALID, "depth_sort must be 0 (auto), 1 (global) or 2 (per tile), optionally | GGR_DEPTH_SORT_NO_BUCKETS");
    if ((st->scissor[0] | st->scissor[1] | st->scissor[2] | st->scissor[3]) &&
        (st->scissor[0] < 0 || st->scissor[1] < 0 || st->scissor[2] <= st->scissor[0] || st->scissor[3] <= st->scissor[1]))
        return fail(GGR_E_INVALID, "scissor must be x0 < x1, y0 < y1, all >= 0 (or all zero for none)");
    if (in->shs && st->sh_stride < (st->sh_degree > 3 ? 16 : (st->sh_degree + 1) * (st->sh_degree + 1)))
        return fail(GGR_E_INVALID, "sh_stride %d too small for sh_degree %d", st->sh_stride, st->sh_degree);
    const int64_t tiles = (int64_t)((st->image_width + GGR_TILE - 1) / GGR_TILE) * ((st->image_height + GGR_TILE - 1) / GGR_TILE);
    if (tiles > (1 << 24)) return fail(GGR_E_LIMIT, "image has %lld tiles; at most 2^24 supported", (long long)tiles);
    if (st->image_width > 65535 * GGR_TILE)   // (the packed tile rect holds 16-bit tile coordinates)
        return fail(GGR_E_LIMIT, "image width %d exceeds %d px (65535 tile columns)", st->image_width, 65535 * GGR_TILE);
    if (st->image_height > 65535 * GGR_TILE)
        return fail(GGR_E_LIMIT, "image height %d exceeds %d px (65535 tile rows)", st->image_height, 65535 * GGR_TILE);
    return GGR_OK;
}

// Per host thread and device: the library's two small host-side resources (the read-back slot and the side stream below).
// A thread OWNS its slot while it lives — no lock on the forward's path — and hands it back to a process-wide pool when it
// ends, without a HIP call (a thread_local destructor can run after the HIP runtime is gone): the next thread that needs
// one for that device takes it from the pool, so short-lived host threads (autograd workers, data-loader threads that
// render) reuse a handful of slots instead of leaking a pinned line, a stream and six events each.  The pool
// itself is never destroyed (static destruction order).
#define GGR_MAX_DEVICES 64
template <class T>
struct SlotPool {
    std::mutex mu;
    std::vector<T*> free_slots[GGR_MAX_DEVICES];
    int created = 0;   // slots ever allocated (ggr_debug_host_slots)
    static SlotPool& get() { static SlotPool* p = new SlotPool; return *p; }
};
template <class T>
struct ThreadSlots {
    T* slot[GGR_MAX_DEVICES] = {};
    ~ThreadSlots() {
        SlotPool<T>& pool = SlotPool<T>::get();
        std::lock_guard<std::mutex> lk(pool.mu);
        for (int d = 0; d < GGR_MAX_DEVICES; d++)
            if (slot[d]) pool.free_slots[d].push_back(slot[d]);
    }
    // this thread's slot for the current device (nullptr: no current device, or its index is beyond GGR_MAX_DEVICES)
    T* current() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GGR_MAX_DEVICES) { (void)hipGetLastError(); return nullptr; }
        if (!slot[dev]) {
            SlotPool<T>& pool = SlotPool<T>::get();
            std::lock_guard<std::mutex> lk(pool.mu);
            if (!pool.free_slots[dev].empty()) { slot[dev] = pool.free_slots[dev].back(); pool.free_slots[dev].pop_back(); }
            else { slot[dev] = new T; pool.created++; }
        }
        return slot[dev];
    }
};
// per host thread and device: one pinned word + one event for the num_rendered read-back of the exact mode (a
// thread has at most one forward between its launch and its sync, so the slot is never shared)
#define GGR_READBACK_ARMED GGR_HOST_ARMED  // (counts stop at 0x7FFFFFFF; GGR_HOST_FAULT_* report a sort fault)
struct ReadbackSlot {
    uint32_t* host = nullptr;
    hipEvent_t ev = nullptr;        // behind the kernel that writes the word
    hipEvent_t ev_start = nullptr;  // in front of the tile-list kernels: the wait's time bound starts here
};
ReadbackSlot* readback_slot() {
    static thread_local ThreadSlots<ReadbackSlot> slots;
    ReadbackSlot* rp = slots.current();
    if (!rp) return nullptr;
    ReadbackSlot& r = *rp;
    if (!r.host) {
        void* p = nullptr;
        // coherent (fine-grained) on purpose: the kernel's system-scope store must become visible to the spinning
        // host while the kernel is still running; a non-coherent mapping would only show it at the kernel's end
        if (hipHostMalloc(&p, 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&r.ev, hipEventDisableTiming) != hipSuccess) { (void)hipHostFree(p); return nullptr; }
        if (hipEventCreateWithFlags(&r.ev_start, hipEventDisableTiming) != hipSuccess) {
            (void)hipEventDestroy(r.ev); r.ev = nullptr; (void)hipHostFree(p); return nullptr;
        }
        r.host = (uint32_t*)p;
    }
    return &r;
}

// The forward's side stream (forward_impl): the SH colour half of the per-Gaussian stage runs on it beside the depth sort and
// the tile-list kernels.  One per (host thread, device), like the read-back slot: created on first use, lowest priority (the
// kernels of the caller's stream — the critical path — win the CUs whenever both have workgroups to place), ordered against
// the caller's stream by two events per forward (fork behind the geometry kernel, join in front of the blend).  Holds no
// state between calls; two forwards of one thread on two streams share it and merely serialise their colour kernels.
// ROCm maps a process's streams onto a FEW hardware queues (four by default, GPU_MAX_HW_QUEUES): a side stream created after
// the host has made three or more streams of its own lands on the queue of the caller's stream, and then every kernel the
// forward chains on the caller's stream behind the fork is held up while the colour kernel runs — measured at C5' 0.80-0.86
// instead of 0.69 ms per step (geometry stage 35 -> 63 µs, depth sort 113 -> 169-250, tile counts 47 -> 78;
// tools/experiments/r06_slow_after_graph.py; bench.py's own graph legs did it to the legs behind them).  So a side stream is
// PROBED against the caller's stream before it is used — the forward's own pattern in miniature: a kernel on the caller's
// stream, the fork, a 100 µs one-wave spin on the candidate, a chain of eight empty kernels on the caller's stream, HIP events
// around the chain, against the same with an empty kernel instead of the spin.  A candidate that stretches the chain is
// dropped and the next one tried (the runtime hands its queues out round robin); no candidate after six: no split for this
// caller stream.  Once per (host thread, device, caller stream): ≈ 0.5 ms and one synchronisation of the caller's stream.
// GPU_MAX_HW_QUEUES=16 in the environment of the process avoids the sharing at its root (measured: 0.69 with eight host streams).
struct SideStream {
    hipStream_t stream = nullptr;     // for the caller stream the call came with (set by side_stream())
    hipEvent_t fork = nullptr, join = nullptr;
    hipEvent_t t0 = nullptr, t1 = nullptr;   // timing brackets of the colour kernel (stage profiling; the probe)
    bool failed = false;
    struct Pair { hipStream_t caller; hipStream_t side; bool usable; } pairs[4];
    int npairs = 0;
};
// true: `cand` runs concurrently with `s`
bool side_stream_is_concurrent(SideStream& r, hipStream_t s, hipStream_t cand) {
    if (hipStreamSynchronize(s) != hipSuccess || hipStreamSynchronize(cand) != hipSuccess) { (void)hipGetLastError(); return false; }
    ReadbackSlot* rb = readback_slot();   // (its pinned, coherent line: word 8 tells the host that the spin is on the device)
    if (!rb) return false;
    volatile uint32_t* started = rb->host + 8;
    // the forward's own pattern: a kernel on the caller's stream, the fork (event on the caller's stream, the candidate waits
    // for it), a kernel on the candidate, then a CHAIN of kernels on the caller's stream — timed with the candidate's kernel a
    // 200 µs spin that is known to be running, and, for comparison, an empty one
    float best = 1e9f, base = 1e9f;
    for (int rep = 0; rep < 4; rep++) {   // (the first launches of a fresh stream pay for its set-up)
        const bool spin = (rep & 1) != 0;
        ggr::launch_noop(s);
        if (hipEventRecord(r.fork, s) != hipSuccess || hipStreamWaitEvent(cand, r.fork, 0) != hipSuccess) break;
        if (spin) {
            *started = 0u;
            ggr::launch_spin(20000ull /*200 µs*/, (uint32_t*)started, cand);
            timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            while (*started == 0u) {   // (bounded: 50 ms)
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6 > 50.0) break;
            }
        } else {
            ggr::launch_noop(cand);
        }
        if (hipEventRecord(r.t0, s) != hipSuccess) break;
        for (int k = 0; k < 8; k++) ggr::launch_noop(s);
        if (hipEventRecord(r.t1, s) != hipSuccess || hipEventSynchronize(r.t1) != hipSuccess) break;
        float ms = 1e9f;
        if (hipEventElapsedTime(&ms, r.t0, r.t1) == hipSuccess) { if (spin) best = std::min(best, ms); else base = std::min(base, ms); }
        (void)hipStreamSynchronize(cand);
    }
    (void)hipGetLastError();
    // measured: 0.017-0.018 ms either way on a candidate with a queue of its own, 0.017-0.023 against 0.053-0.054 on one that
    // shares the caller's
    const bool ok = best < 1.5f * base + 0.010f;
    if (getenv("GGR_SIDE_STREAM_DEBUG"))
        fprintf(stderr, "[ggr] side-stream probe: caller %p candidate %p: chain of 8 kernels %.3f ms beside an empty kernel, %.3f beside a "
                        "running spin -> %s\n", (void*)s, (void*)cand, base, best, ok ? "concurrent" : "shares the caller's queue");
    return ok;
}
SideStream* side_stream(hipStream_t caller) {
    static thread_local ThreadSlots<SideStream> slots;
    SideStream* rp = slots.current();
    if (!rp) return nullptr;
    SideStream& r = *rp;
    if (r.failed) return nullptr;
    if (!r.fork) {
        bool ok = hipEventCreateWithFlags(&r.fork, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&r.join, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreate(&r.t0) == hipSuccess && hipEventCreate(&r.t1) == hipSuccess;
        if (!ok) { r.failed = true; (void)hipGetLastError(); return nullptr; }
    }
    for (int i = 0; i < r.npairs; i++)
        if (r.pairs[i].caller == caller) { r.stream = r.pairs[i].side; return r.pairs[i].usable ? &r : nullptr; }
    if (r.npairs == 4) return nullptr;   // (a fifth caller stream of one thread: no split for it)
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // lo = least priority (numerically greatest)
    const char* pe = getenv("GGR_COLOUR_PRIO");   // dev: "high" / "normal" instead of the lowest priority
    const int prio = (pe && *pe == 'h') ? hi : (pe && *pe == 'n') ? (lo + hi) / 2 : lo;
    const char* np = getenv("GGR_SIDE_STREAM_PROBE");   // "0": take the first stream unprobed (the behaviour until round 5)
    const bool probe = !(np && *np == '0');
    SideStream::Pair pr{caller, nullptr, false};
    // (an earlier caller stream's side stream may serve this one too)
    for (int i = 0; i < r.npairs && !pr.usable; i++)
        if (r.pairs[i].usable && (!probe || side_stream_is_concurrent(r, caller, r.pairs[i].side))) { pr.side = r.pairs[i].side; pr.usable = true; }
    hipStream_t rejected[6];
    int nrej = 0;
    for (int attempt = 0; attempt < 6 && !pr.usable; attempt++) {
        hipStream_t cand = nullptr;
        if (hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, prio) != hipSuccess) { (void)hipGetLastError(); break; }
        if (!probe || side_stream_is_concurrent(r, caller, cand)) { pr.side = cand; pr.usable = true; }
        else rejected[nrej++] = cand;   // (kept until the search ends: destroyed at once, its queue slot would be handed out again)
    }
    for (int i = 0; i < nrej; i++) (void)hipStreamDestroy(rejected[i]);
    r.pairs[r.npairs++] = pr;
    r.stream = pr.side;
    return pr.usable ? &r : nullptr;
}
// GGR_SPLIT_COLOUR=0: the per-Gaussian stage as ONE kernel on the caller's stream (dev / A-B measurements)
// (read on every forward: a test or a host can switch within a process)
bool split_colour_enabled() {
    const char* e = getenv("GGR_SPLIT_COLOUR");
    return !(e && *e == '0');
}

// Blocks of the colour kernel's launch (per Gaussian set): a few PERSISTENT blocks per CU.  Unthrottled (one block per
// 256-Gaussian chunk) the kernel finishes in 50 µs at C3 and doubles the duration of the depth-sort passes it runs beside (a
// sort tile waits for a CU's wave slots and LDS, every later tile for its look-back): forward +13 µs instead of −30.  One
// block per CU streams ≈ 2.7 TB/s and has until the blend needs the colours — the depth sort and the tile-list kernels,
// whose duration grows with the number of (view, Gaussian) pairs as the colour kernel's bytes do; beyond ≈ 2 M pairs one
// block per CU is no longer enough (C6′, 4.9 M pairs: the blend waited 0.35 ms for the colours), so the count grows with the
// pairs: 1 + pairs / 2 M, at most 4.  GGR_COLOUR_BLOCKS_PER_CU overrides (0 = unthrottled).
#define GGR_SPLIT_MAX_POINTS 2500000
int colour_grid_blocks(size_t pairs) {
    const char* e_blocks = getenv("GGR_COLOUR_BLOCKS_PER_CU");   // (read per call, like GGR_SPLIT_COLOUR)
    const int forced = (e_blocks && *e_blocks) ? atoi(e_blocks) : -1;
    // (the CU count of the CURRENT device, looked up once per device — ADVICE r5: a process that drives several devices must
    //  not throttle all of them by the first one's count)
    static int cus_of[GGR_MAX_DEVICES] = {0};
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < GGR_MAX_DEVICES) {
        int c = __atomic_load_n(&cus_of[dev], __ATOMIC_RELAXED);
        if (c == 0) {
            hipDeviceProp_t prop;
            c = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
            __atomic_store_n(&cus_of[dev], c, __ATOMIC_RELAXED);
        }
        cus = c;
    } else {
        (void)hipGetLastError();
    }
    const int per_cu = forced >= 0 ? forced : (int)std::min<size_t>(4, 1 + pairs / 2000000);
    return per_cu <= 0 ? 0 : per_cu * cus;
}

// The exact mode's wait for num_rendered: the host watches the pinned word (armed with a sentinel no count can take)
// and, every 1024 polls, asks `query` (hipEventQuery of the event behind the kernel that writes the word) whether the
// kernel has ended.  Leaves on: the word changed; the query says "done" (the word is then re-read once — a
// non-coherent mapping shows it only now); the query returns ANYTHING other than hipErrorNotReady (stream in error,
// device lost: the word will never change) → GGR_E_HIP; or `timeout_s` seconds without either (a hung GPU) →
// GGR_E_HIP.  The clock of that bound only starts once `query_start` (the event recorded on the stream right IN FRONT
// of the tile-list kernels, or NULL: at once) reports completion: work that was queued on the stream ahead of this
// forward — a long evaluation queue, a device shared with another process — is not this forward's hang.
// The bound itself is GGR_READBACK_TIMEOUT_S seconds (environment, default 30; ≤ 0 = wait for as long as the event
// query keeps answering "not ready").  Work queued AHEAD of the forward has its own, generous bound from this function's
// entry (GGR_READBACK_QUEUE_TIMEOUT_S, default 600 s, ≤ 0 = none): a device hung in earlier work never starts the first
// clock.  The device-side spins are bounded the same way — with both bounds at their defaults nothing in a forward waits
// forever, on a GPU that makes progress or on one that does not.
typedef hipError_t (*ReadbackQueryFn)(void*);
double readback_timeout_s() {
    const char* e = getenv("GGR_READBACK_TIMEOUT_S");
    if (e && *e) { char* end = nullptr; const double v = strtod(e, &end); if (end != e) return v; }
    return 30.0;
}
// bound on the wait for work queued on the stream BEFORE this forward (the per-forward bound above only runs from the
// tile-list kernels' turn): generous, since a caller may legitimately have minutes of work in front; <= 0 disables
double readback_queue_timeout_s() {
    const char* e = getenv("GGR_READBACK_QUEUE_TIMEOUT_S");
    if (e && *e) { char* end = nullptr; const double v = strtod(e, &end); if (end != e) return v; }
    return 600.0;
}
int wait_readback(volatile uint32_t* hw, ReadbackQueryFn query, void* ctx, double timeout_s, uint32_t* value,
                  ReadbackQueryFn query_start = nullptr, void* ctx_start = nullptr) {
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const timespec t_entry = t0;
    const double queue_timeout_s = readback_queue_timeout_s();
    bool started = query_start == nullptr;
    uint32_t v = *hw;
    for (uint32_t spins = 0; v == GGR_READBACK_ARMED; v = *hw) {
        if ((++spins & 0x3FFu) == 0) {
            const hipError_t q = query(ctx);
            if (q == hipSuccess) { v = *hw; break; }
            if (q != hipErrorNotReady)
                return fail(GGR_E_HIP, "num_rendered read-back: the stream is in error (%s); frame not rendered", hipGetErrorString(q));
            if (!started) {   // still behind earlier work of the stream: the bound does not run yet
                const hipError_t qs = query_start(ctx_start);
                if (qs == hipSuccess) { started = true; clock_gettime(CLOCK_MONOTONIC, &t0); }
                else if (qs != hipErrorNotReady)
                    return fail(GGR_E_HIP, "num_rendered read-back: the stream is in error (%s); frame not rendered", hipGetErrorString(qs));
                else {   // a device hung in work queued BEFORE this forward never starts the clock above: second, absolute bound
                    timespec t1;
                    clock_gettime(CLOCK_MONOTONIC, &t1);
                    if (queue_timeout_s > 0.0 &&
                        (double)(t1.tv_sec - t_entry.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t_entry.tv_nsec) > queue_timeout_s)
                        return fail(GGR_E_HIP, "num_rendered read-back: the stream's earlier work has not finished within %.0f s "
                                               "(GGR_READBACK_QUEUE_TIMEOUT_S); frame not rendered", queue_timeout_s);
                }
                __builtin_ia32_pause();
                continue;
            }
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if (timeout_s > 0.0 && (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > timeout_s)
                return fail(GGR_E_HIP, "num_rendered read-back: no answer from the GPU within %.0f s of the tile-list kernels' turn "
                                       "(GGR_READBACK_TIMEOUT_S); frame not rendered", timeout_s);
        }
        __builtin_ia32_pause();
    }
    if (v == GGR_READBACK_ARMED) return fail(GGR_E_HIP, "num_rendered read-back: the tile-list kernel ended without writing the count");
    *value = v;
    return GGR_OK;
}
hipError_t query_event(void* ev) { return hipEventQuery((hipEvent_t)ev); }

InputForm input_form(const GgrSettings* st, const GgrForwardIn* in, int sets, const void* dL_dshs = nullptr) {
    InputForm f;
    // flat float4 staging of odd-length SH rows needs every set's rows (and gradient rows) to start 16-B aligned
    f.sh_aligned = (((uintptr_t)in->shs | (uintptr_t)dL_dshs) & 15) == 0 &&
                   (sets <= 1 || ((int64_t)st->num_points * st->sh_stride * 3) % 4 == 0);
    f.cov_stride = in->cov3D_full ? 9 : 6;
    f.sh_channel_major = in->sh_channel_major ? 1 : 0;
    f.aux_affine = (in->aux_affine && !in->aux_precomp) ? 1 : 0;
    f.aux_a = in->aux_a;
    f.aux_b = in->aux_b;
    f.sh_cap = st->sh_max_degree == 4 ? 4 : 3;  // default 3: INTEGRATION.md §7
    f.tight_rects = st->reference_rects ? 0 : 1;
    const int gx = (st->image_width + GGR_TILE - 1) / GGR_TILE, gy = (st->image_height + GGR_TILE - 1) / GGR_TILE;
    f.sc_x0 = 0; f.sc_y0 = 0; f.sc_x1 = gx; f.sc_y1 = gy;
    const int32_t* sc = st->scissor;
    if (sc[0] | sc[1] | sc[2] | sc[3]) {   // the tiles that overlap the pixel window (validate() checked its shape)
        f.sc_x0 = min(gx, sc[0] / GGR_TILE); f.sc_y0 = min(gy, sc[1] / GGR_TILE);
        f.sc_x1 = min(gx, (sc[2] + GGR_TILE - 1) / GGR_TILE); f.sc_y1 = min(gy, (sc[3] + GGR_TILE - 1) / GGR_TILE);
    }
    return f;
}

size_t tiles_of(int W, int H) { return (size_t)((W + GGR_TILE - 1) / GGR_TILE) * ((H + GGR_TILE - 1) / GGR_TILE); }

// the reference's call: one camera, taken from the settings
ViewSet single_view(const GgrSettings* st, const GgrForwardIn* in) {
    ViewSet vs;
    vs.V = 1; vs.sets = 1; vs.vps = 1;
    vs.view = st->viewmatrix; vs.proj = st->projmatrix; vs.campos = st->campos; vs.bg = st->bg;
    vs.tanfov = st->tanfov_dev; vs.input_scale = in->input_scale;
    vs.tanfovx = st->tanfovx; vs.tanfovy = st->tanfovy;
    return vs;
}

int view_set(const GgrSettings* st, const GgrViews* v, ViewSet* vs) {
    if (!v || v->num_views < 1) return fail(GGR_E_INVALID, "GgrViews: num_views must be >= 1");
    if (!v->viewmatrix || !v->projmatrix || !v->campos || !v->bg)
        return fail(GGR_E_INVALID, "GgrViews: null camera array");
    const int64_t V = v->num_views;
    const int64_t sets = v->num_sets > 1 ? v->num_sets : 1;
    if (V % sets != 0) return fail(GGR_E_INVALID, "GgrViews: num_views must be a multiple of num_sets");
    if (sets > GGR_SORT_MAX_SEGMENTS) return fail(GGR_E_LIMIT, "GgrViews: at most 64 Gaussian sets per launch set");
    const int64_t gy = (st->image_height + GGR_TILE - 1) / GGR_TILE;
    if (V * st->num_points >= 0x7FFFFFFFll) return fail(GGR_E_LIMIT, "num_views x num_points too large");
    if (V * gy > 65535) return fail(GGR_E_LIMIT, "num_views x tile rows exceeds 65535");
    if (V * (int64_t)tiles_of(st->image_width, st->image_height) > (1 << 24)) return fail(GGR_E_LIMIT, "more than 2^24 tiles over all views");
    vs->V = (int)V; vs->sets = (int)sets; vs->vps = (int)(V / sets);
    vs->view = v->viewmatrix; vs->proj = v->projmatrix; vs->campos = v->campos; vs->bg = v->bg;
    vs->tanfov = v->tanfov; vs->input_scale = v->input_scale;
    vs->tanfovx = st->tanfovx; vs->tanfovy = st->tanfovy;
    return GGR_OK;
}

}  // namespace

extern "C" {

int ggr_abi_version(void) { return GGR_ABI_VERSION; }
// the sha256 of the sources this library was compiled from (ggrt_official_amd/_build.py source_hash(): the build passes
// it as -DGGR_SOURCE_HASH).  The marker in front lets the build read it from the file's bytes without loading it.
#ifndef GGR_SOURCE_HASH
#define GGR_SOURCE_HASH "0000000000000000000000000000000000000000000000000000000000000000"
#endif
const char* ggr_source_hash(void) {
    static const char tagged[] = "ggr-source-hash:" GGR_SOURCE_HASH;
    return tagged + 16;
}
const char* ggr_last_error(void) { return g_err; }

size_t ggr_geom_bytes(int32_t P) { return ggr_carve_geom(nullptr, (size_t)(P > 0 ? P : 0)).bytes; }
size_t ggr_image_bytes(int32_t W, int32_t H) { return ggr_carve_image(nullptr, W, H).bytes; }
size_t ggr_binning_bytes(int64_t N, int32_t, int32_t) { return ggr_point_list_bytes((size_t)(N > 0 ? N : 0)); }
size_t ggr_work_bytes(int32_t P, int32_t W, int32_t H) {
    return ggr::plan_tile_lists((size_t)(P > 0 ? P : 0), tiles_of(W, H)).work_bytes;
}
size_t ggr_backward_scratch_bytes(int32_t P) { return ggr_carve_bwd(nullptr, (size_t)(P > 0 ? P : 0)).bytes; }

}  // extern "C"

namespace {

// forward of V views of the same P Gaussians (V = 1: ggr_forward).  All per-Gaussian state is per (view, Gaussian),
// the tiles of the views are stacked (ggr_common.h ViewSet): one preprocess launch, ONE depth sort over the V·P keys,
// one tile-list build over the V·T tiles, one blend launch.
int forward_impl(const GgrSettings* st, const ViewSet& vs, const GgrForwardIn* in, GgrForwardOut* out, GgrAllocFn alloc,
                 void* alloc_ctx, void* stream) {
    if (!out || !out->out_color || !out->geom_buffer || !out->image_buffer || !alloc)
        return fail(GGR_E_INVALID, "null output / buffer / allocator");
    if (st->num_points > 0 && !out->radii) return fail(GGR_E_INVALID, "null radii");
    hipStream_t s = (hipStream_t)stream;
    const int P1 = st->num_points, W = st->image_width, H = st->image_height, NV = vs.V;
    const int P = P1 * NV;                      // (view, Gaussian) pairs
    const int gx = (W + GGR_TILE - 1) / GGR_TILE;
    const size_t tiles = tiles_of(W, H) * (size_t)NV;
    const bool dbg = st->debug != 0;

    const size_t segs = ggr_sort_segments((size_t)NV);  // the depth sort runs one segment per view
    // (no_backward: the caller may have brought the smaller ggr_geom_bytes_inference buffer — nothing beyond it is touched)
    GeomLayout g = ggr_carve_geom(out->geom_buffer, (size_t)P, segs, /*with_jac=*/out->no_backward == 0);
    ImageLayout im = ggr_carve_image(out->image_buffer, W, H, NV);
    StageTimer tm(s, out->stage_ms, GGR_FWD_STAGES);

    // How the lists get their depth order (GgrSettings.depth_sort; tile_sort.hip).  Per tile: no global depth sort — the
    // tile-list builder walks the Gaussians in id order and every tile's list is sorted on its own afterwards.  A list longer
    // than GGR_TSORT_CAP_LARGE cannot be sorted that way: with a read-back (exact mode, capacity_is_hint) the call then
    // rebuilds the lists with the global sort; the pure sync-free mode has no read-back, so AUTO keeps the global sort there.
    const bool sync_free = out->binning_capacity > 0 && out->binning_buffer != nullptr;
    const bool hinted = sync_free && out->capacity_is_hint != 0;   // exact mode with a guessed buffer: N is awaited at the END
    if (out->binning_capacity >= 0x7FFFFFFF) return fail(GGR_E_LIMIT, "binning_capacity too large");
    int depth_sort = st->depth_sort;
    // The global sort has two forms (binning.hip): three stable passes, or ONE partition pass into depth buckets + every bucket
    // sorted in LDS.  The bucket form can meet a bucket it cannot sort (> 8192 different keys inside 1/4096 of the frame's
    // depth range): it says so through the read-back and the call then sorts again in three passes — so it needs a read-back,
    // like the per-tile form.  GGR_DEPTH_SORT_NO_BUCKETS (or GGR_GLOBAL_SORT=3pass) keeps the three passes.
    bool three_pass = (depth_sort & GGR_DEPTH_SORT_NO_BUCKETS) != 0;
    depth_sort &= ~GGR_DEPTH_SORT_NO_BUCKETS;
    if (const char* e = getenv("GGR_DEPTH_SORT")) {   // dev / A-B override of AUTO (read per call): "global" | "per_tile"
        if (depth_sort == GGR_DEPTH_SORT_AUTO && *e)
            depth_sort = (*e == 'g' || *e == '1') ? GGR_DEPTH_SORT_GLOBAL : (*e == 'p' || *e == '2') ? GGR_DEPTH_SORT_PER_TILE : depth_sort;
    }
    if (const char* e = getenv("GGR_GLOBAL_SORT")) { if (*e == '3' || *e == 'l') three_pass = true; }   // dev / A-B: "3pass" | "buckets"
    // AUTO: per tile where its lists are short — at most 256 (view, Gaussian) pairs per tile on average (a 1080p frame with 1 M
    // Gaussians: 123; its longest list: 1 100 entries) and, when the caller knows, a longest list of at most 4096 entries — and
    // the call has a read-back to fall back with.  Measured (NOTES r6, fwd+bwd): C3 0.78 against 0.80 ms, 200 k Gaussians at
    // 504 × 378 0.22 against 0.30; GGRt's own shapes — 1 M pixel-aligned Gaussians on 660 tiles, every list 4 000-6 000 entries —
    // 0.77 against 0.69: those keep the global sort.
    const uint32_t len_hint_in = out->max_list_len > 0 ? (uint32_t)out->max_list_len : 0u;
    if (depth_sort == GGR_DEPTH_SORT_AUTO)
        depth_sort = ((!sync_free || hinted) && (size_t)P <= 256 * tiles && len_hint_in <= 4096) ? GGR_DEPTH_SORT_PER_TILE
                                                                                             : GGR_DEPTH_SORT_GLOBAL;
    bool per_tile = depth_sort == GGR_DEPTH_SORT_PER_TILE && P > 0 && tiles > 0;
    const bool buckets = !three_pass && (!sync_free || hinted) && P > 0 && ggr::radix_sort_buckets_ok((size_t)P / segs);
    out->depth_sort_used = per_tile ? GGR_DEPTH_SORT_PER_TILE : buckets ? GGR_DEPTH_SORT_GLOBAL : GGR_DEPTH_SORT_GLOBAL_3PASS;
    const uint32_t len_hint = len_hint_in;
    out->max_list_len = -1;

    const ggr::TileListPlan plan = ggr::plan_tile_lists((size_t)P, tiles);
    // (per tile: the id-order scatter writes (id, key) entries that the per-tile sort reads — 8 B per list entry of scratch.
    //  With the list buffer's size known up front it rides on the work area; in the exact mode on the list buffer's tail)
    const size_t pairs_up_front = (per_tile && sync_free) ? ggr_pair_list_bytes((size_t)out->binning_capacity) : 0;
    void* work = alloc(alloc_ctx, plan.work_bytes + pairs_up_front);  // 1st allocator call: transient work area
    if (!work) return fail(GGR_E_ALLOC, "work-area allocator returned NULL");
    uint2* pair_list = pairs_up_front ? (uint2*)((char*)work + plan.work_bytes) : nullptr;
    uint2* rect_sorted = nullptr;
    uint32_t *totals_area = nullptr, totals_words = 0;   // the tile-list builder's per-tile / per-group totals (start from zero)
    ggr::tile_list_gather_targets(plan, work, tiles, &rect_sorted, &totals_area, &totals_words);

    // 1. per-Gaussian projection.  With SH colours the stage is split (preprocess.hip PART): the geometry half runs here, in
    //    front of the depth sort; the colour half — the SH rows, 4/5 of the stage's bytes, needed by the blend only — runs on
    //    the side stream beside the latency-bound sort / tile-list kernels (which leave HBM and most CUs idle) and is joined
    //    in front of the blend.  Not while the caller's stream is being captured into a graph (a thread-local side stream
    //    would be pulled into the capture), not in debug mode (one kernel at a time), not for precomputed colours.
    const InputForm inf = input_form(st, in, vs.sets);
    SideStream* side = nullptr;
    // … nor for more than GGR_SPLIT_MAX_POINTS Gaussians per set: the colour kernel's bytes grow with them faster than the
    // window beside the binning does (C6′, 4.9 M Gaussians of 25 coefficients: 1.9 GB to move within ≈ 0.47 ms — at that rate
    // the depth sort beside it takes 0.66 instead of 0.29 ms; step 2.43 ms as one kernel, 2.51 split)
    // … nor per tile: beside the tile counts and the scatter the colour kernel costs them what it saves (count +22, scatter +12 µs
    // at C3), beside the per-tile sort it takes twice its time and the blend waits for it (NOTES r6)
    if (in->shs && P1 > 0 && P1 <= GGR_SPLIT_MAX_POINTS && !dbg && split_colour_enabled() &&
        (!per_tile || getenv("GGR_COLOUR_FORK") != nullptr)) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) side = side_stream(s);
        else (void)hipGetLastError();
    }
    // (per tile: the kernel also clears the tile-list totals — the depth sort's last pass does it otherwise — and leaves the
    //  sort's work area alone)
    ggr::launch_preprocess_fwd(P1, st->sh_degree, st->sh_stride, in->means3D, in->shs, in->colors_precomp,
                               in->opacities, in->scales, in->rotations, st->scale_modifier, in->cov3D_precomp,
                               in->aux_precomp, vs, W, H, out->radii, g, inf, s, side ? GGR_PRE_GEOMETRY : GGR_PRE_ALL, 0,
                               out->no_backward ? 0 : 1, per_tile ? totals_area : nullptr, per_tile ? totals_words : 0u,
                               per_tile ? 1 : 0);
    KCHECK(dbg, s, "preprocess_fwd");
    bool colour_pending = side != nullptr;
    auto fork_colour = [&]() -> int {   // the colour kernel behind everything queued on `s` so far, on the side stream
        if (!colour_pending) return GGR_OK;
        colour_pending = false;
        HIP_TRY(hipEventRecord(side->fork, s));
        HIP_TRY(hipStreamWaitEvent(side->stream, side->fork, 0));
        // (the launch's grid is (blocks, Gaussian sets): the persistent blocks are shared out between the sets)
        const int cg = colour_grid_blocks((size_t)P);
        const int colour_grid = cg <= 0 ? 0 : std::max(1, cg / std::max(1, vs.sets));
        if (out->stage_ms) (void)hipEventRecord(side->t0, side->stream);
        ggr::launch_preprocess_fwd(P1, st->sh_degree, st->sh_stride, in->means3D, in->shs, in->colors_precomp,
                                   in->opacities, in->scales, in->rotations, st->scale_modifier, in->cov3D_precomp,
                                   in->aux_precomp, vs, W, H, out->radii, g, inf, side->stream, GGR_PRE_COLOUR,
                                   colour_grid, out->no_backward ? 0 : 1);
        if (out->stage_ms) (void)hipEventRecord(side->t1, side->stream);
        HIP_TRY(hipEventRecord(side->join, side->stream));
        return GGR_OK;
    };
    // (started right behind the geometry kernel: started behind the depth sort, or behind the tile counts, it delays the
    //  blend by more than it spares the sort — NOTES r5)
    int colour_fork_at = 0;   // 0: behind the geometry kernel, 1: behind the tile counts, 2: behind the scatter (dev: NOTES r6)
    if (const char* e = getenv("GGR_COLOUR_FORK")) { if (*e >= '0' && *e <= '2') colour_fork_at = *e - '0'; }
    if (colour_fork_at == 0) { const int rc = fork_colour(); if (rc != GGR_OK) return rc; }
    // (every return below this point must leave the caller's stream ordered behind the side stream: JoinGuard.  The guard
    //  also covers a colour kernel that was never started: join() launches it first)
    struct JoinGuard {
        SideStream* sd; hipStream_t s; bool done = false;
        void join() { if (sd && !done) { (void)hipStreamWaitEvent(s, sd->join, 0); done = true; } }
        ~JoinGuard() { join(); }
    } joiner{side, s};
    auto fork_colour_at = [&](int where) -> int { return (colour_pending && colour_fork_at == where) ? fork_colour() : GGR_OK; };
    tm.mark(GGR_FWD_PREPROCESS);

    // sync-free mode: the caller brought a list buffer of `binning_capacity` entries → no read-back, no host
    // sync, no second allocator call; the whole forward (and backward) is then hipGraph-capturable
    uint32_t capacity = sync_free ? (uint32_t)out->binning_capacity : 0xFFFFFFFFu;
    // exact mode: num_rendered travels to the host through a pinned word written by the scan kernel, with an
    // event right behind that kernel — the host wakes up while the last scan kernel still runs and has the list
    // buffer allocated and the scatter queued by the time the GPU gets there (a device→host memcpy into pageable
    // memory + stream sync left the GPU idle for that long)
    ReadbackSlot* rb = ((!sync_free || hinted) && P > 0) ? readback_slot() : nullptr;
    const uint32_t* sort_fault = ggr::radix_sort_fault_word(g.hist);

    // ---- the pieces of the list build --------------------------------------------------------------------------------
    // 2. stable sort of the Gaussians by depth bits (ties keep ascending id).  Its last pass also drops every
    //    Gaussian's tile rect at its sorted position (into the tile-list work area) and clears the per-tile totals.
    uint32_t *dk = nullptr, *order = nullptr;
    auto global_sort = [&](bool area_cleared, bool bucket_form) {
        // (preprocess already wrote the keys into g.keys_a; the values are the identity, formed by the sort's first pass)
        ggr::radix_sort_pairs(g.keys_a, g.keys_b, g.vals_a, g.vals_b, g.hist, (size_t)P, (uint32_t)segs, &dk, &order, s,
                              /*hist_zeroed=*/area_cleared /*by preprocess_fwd*/,
                              /*block_max_ready=*/(uint32_t)((P1 + GGR_PRE_THREADS - 1) / GGR_PRE_THREADS) * (uint32_t)vs.sets /*likewise*/,
                              /*identity_vals=*/true /*preprocess_fwd writes no values: the first pass forms them*/,
                              g.rect, rect_sorted, totals_area, totals_words, bucket_form);
    };
    // 3. per-(chunk, tile) counts → list positions, tile ranges, num_rendered (+ the longest list)
    auto count_pass = [&](bool id_order, bool to_host) {
        if (to_host && rb) {
            ((volatile uint32_t*)rb->host)[1] = 0u;
            *(volatile uint32_t*)rb->host = GGR_READBACK_ARMED;
            (void)hipEventRecord(rb->ev_start, s);
        }
        ggr::launch_tile_list_count(plan, (size_t)P, tiles, gx, order, g.rect, work, im.ranges, g.counters, capacity, s,
                                    /*rects_gathered=*/true, (to_host && rb) ? rb->host : nullptr, (to_host && rb) ? rb->ev : nullptr,
                                    id_order ? nullptr : sort_fault, id_order, id_order ? GGR_TSORT_CAP_LARGE : 0xFFFFFFFFu);
    };
    // 4. in-order scatter of the ids into the per-tile lists (+ per tile: the depth order, list by list)
    uint32_t* point_list = nullptr;
    uint32_t sorted_upto = 0;   // per tile: lists of up to this many entries have been depth-sorted in `point_list`
    // the per-tile sort of the lists with sorted_upto < length <= upto: one launch per length class it spans (tile_sort.hip)
    auto tile_sort_upto = [&](uint32_t upto) {
        upto = std::min<uint32_t>(upto, GGR_TSORT_CAP_LARGE);
        if (upto <= sorted_upto) return;
        // One launch when a class with at least four workgroups per CU covers everything (lists up to 3072 entries: a few tiles
        // just beyond 2048 do not pay for a launch of their own); otherwise the lists up to 2048 in their class and the
        // longer ones in theirs.  The launch with the largest class also copies the ids of lists beyond it: unsorted, but
        // nothing uninitialised is left for the blend.
        const uint32_t one_launch = 3072;
        if (upto <= one_launch || sorted_upto >= GGR_TSORT_CAP_SMALL) {
            ggr::launch_tile_depth_sort(tiles, im.ranges, point_list, pair_list, sorted_upto, upto, s, 1, g.counters + 3);
        } else {
            ggr::launch_tile_depth_sort(tiles, im.ranges, point_list, pair_list, sorted_upto, GGR_TSORT_CAP_SMALL, s, 0, g.counters + 3);
            ggr::launch_tile_depth_sort(tiles, im.ranges, point_list, pair_list, GGR_TSORT_CAP_SMALL, upto, s, 1, g.counters + 3);
        }
        sorted_upto = upto;
    };
    auto scatter_pass = [&](bool id_order, uint32_t cap_entries, uint32_t expect_longest) {
        ggr::launch_tile_list_scatter(plan, (size_t)P, tiles, gx, id_order ? nullptr : order, g.rect, work, point_list,
                                      cap_entries, s, id_order ? g.keys_a : nullptr, id_order ? pair_list : nullptr);
        tm.mark(GGR_FWD_TILE_SCATTER);
        (void)fork_colour_at(2);
        sorted_upto = 0;
        if (id_order) {
            tile_sort_upto(expect_longest);
            tm.mark(GGR_FWD_TILE_SORT);
        }
    };
    // 5. blend (clears the caller's backward scratch on the side; an image without tiles launches nothing)
    const bool scissored = (st->scissor[0] | st->scissor[1] | st->scissor[2] | st->scissor[3]) != 0;
    auto blend_pass = [&]() {
        if (colour_pending) (void)fork_colour();   // (a frame without list entries: nothing started it yet)
        joiner.join();   // the colour records (side stream) are complete before the blend reads them
        ggr::launch_blend_fwd(W, H, im.ranges, point_list, g.splat, g.colour, vs.bg, out->out_color, out->no_backward ? nullptr : im.final_T,
                              im.n_contrib, out->out_depth, out->no_backward ? nullptr : im.ckpt, im.ckpt_slots, im.tile_top, NV,
                              scissored ? 1 : 0, out->backward_scratch, ggr_carve_bwd(nullptr, (size_t)P1, (size_t)NV).bytes, s);
    };
    // what the host word(s) said: N, or a fault; the longest list
    uint32_t num_rendered = 0, longest = 0;
    bool bucket_fault = false;   // the depth sort's bucket form left a bucket unsorted: the lists must be built again
    bool bucket_slow = false;    // … or sorted everything, much of it the slow way: advisory (GGR_DEPTH_SORT_GLOBAL_SLOW)
    auto read_counts = [&]() -> int {
        if (rb) {
            // the single host sync of forward.  N is written by the FIRST block of the last tile-list kernel: the host
            // watches the pinned word instead of waiting for that kernel's end (wait_readback above)
            const int rc = wait_readback((volatile uint32_t*)rb->host, query_event, (void*)rb->ev, readback_timeout_s(), &num_rendered,
                                         query_event, (void*)rb->ev_start);
            if (rc != GGR_OK) return rc;
            longest = ((volatile uint32_t*)rb->host)[1];   // (stored before N's release store)
            bucket_fault = (longest >> 31) != 0u;
            bucket_slow = ((longest >> 30) & 1u) != 0u;
            longest &= 0x3FFFFFFFu;
        } else {   // no pinned slot (hipHostMalloc / hipEventCreate failed): copy + sync — a frame is never returned unchecked
            uint32_t w4[4] = {0u, 0u, 0u, 0u};
            HIP_TRY(hipMemcpyAsync(w4, g.counters, 16, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            num_rendered = (w4[1] & 2u) ? GGR_HOST_FAULT_SPIN : (w4[1] & 4u) ? GGR_HOST_FAULT_RANGE : w4[0];
            longest = w4[2];
            bucket_fault = (w4[1] & 16u) != 0u;
            bucket_slow = (w4[1] & 32u) != 0u;
        }
        // raised by the tile-list kernel that writes N (bin_group_prefix_kernel)
        if (num_rendered == GGR_HOST_FAULT_SPIN) return fail(GGR_E_HIP, "%s", kSpinFault);
        if (num_rendered == GGR_HOST_FAULT_RANGE) return fail(GGR_E_LIMIT, "%s", kRangeFault);
        if (num_rendered >= 0x7FFFFFFFu) return fail(GGR_E_LIMIT, "num_rendered %u too large", num_rendered);
        out->num_rendered = (int64_t)num_rendered;
        out->max_list_len = (int32_t)longest;
        return GGR_OK;
    };
    // a list too long for the per-tile sort: the lists once more, through the global depth sort (the counts restart from zero:
    // the sort's last pass clears the totals; its own work area was left untouched by preprocess_fwd)
    bool rebuilt = false;
    auto rebuild_global = [&]() {
        out->depth_sort_used = per_tile ? GGR_DEPTH_SORT_GLOBAL_3PASS : GGR_DEPTH_SORT_GLOBAL_FELL_BACK;
        per_tile = false;
        rebuilt = true;
        global_sort(/*area_cleared=*/false, /*bucket_form=*/false);
        tm.mark(GGR_FWD_DEPTH_SORT);
        count_pass(/*id_order=*/false, /*to_host=*/false);
        tm.mark(GGR_FWD_TILE_COUNT);
    };

    if (P > 0 && tiles > 0) {
        if (!per_tile) {
            global_sort(/*area_cleared=*/true, buckets);
            KCHECK(dbg, s, "depth sort");
        }
        tm.mark(GGR_FWD_DEPTH_SORT);
        count_pass(per_tile, /*to_host=*/true);
    } else {
        tm.mark(GGR_FWD_DEPTH_SORT);
        HIP_TRY(hipMemsetAsync(im.ranges, 0, (tiles ? tiles : 1) * sizeof(uint2), s));
        HIP_TRY(hipMemsetAsync(g.counters, 0, 16, s));
        rb = nullptr;
    }
    KCHECK(dbg, s, "tile_list_count");
    { const int rc = fork_colour_at(1); if (rc != GGR_OK) return rc; }
    if (dbg && sync_free && P > 0 && tiles > 0) {  // debug mode may sync: check the sort's fault bit right here
        uint32_t two[2] = {0u, 0u};
        HIP_TRY(hipMemcpy(two, g.counters, 8, hipMemcpyDeviceToHost));
        if (two[1] & 2u) return fail(GGR_E_HIP, "%s", kSpinFault);
        if (two[1] & 4u) return fail(GGR_E_LIMIT, "%s", kRangeFault);
    }
    if (!sync_free) {
        if (P > 0 && tiles > 0) { const int rc = read_counts(); if (rc != GGR_OK) return rc; }
        else { out->num_rendered = 0; out->max_list_len = 0; }
        tm.mark(GGR_FWD_TILE_COUNT);
        if ((per_tile && longest > GGR_TSORT_CAP_LARGE) || (!per_tile && bucket_fault)) rebuild_global();
        // 2nd call: kept for backward (per tile: + the (id, key) scratch on its tail)
        void* bin_mem = alloc(alloc_ctx, ggr_point_list_bytes(num_rendered) + (per_tile ? ggr_pair_list_bytes(num_rendered) : 0));
        if (!bin_mem) return fail(GGR_E_ALLOC, "binning allocator returned NULL");
        out->binning_buffer = bin_mem;
        point_list = (uint32_t*)bin_mem;
        if (per_tile) pair_list = (uint2*)((char*)bin_mem + ggr_point_list_bytes(num_rendered));
    } else {
        out->num_rendered = -1;  // known on the device only: ggr_forward_status reads it (and the overflow flag)
        tm.mark(GGR_FWD_TILE_COUNT);
        point_list = (uint32_t*)out->binning_buffer;
    }
    if (P > 0 && tiles > 0 && (sync_free || num_rendered > 0)) {
        // (how long the lists are that the per-tile sort is launched for: known in the exact mode, the caller's guess with a
        //  guessed buffer — checked and repaired at the end —, everything it can take in the pure sync-free mode)
        const uint32_t expect = !sync_free ? longest : hinted ? (len_hint ? len_hint : GGR_TSORT_CAP_SMALL) : GGR_TSORT_CAP_LARGE;
        scatter_pass(per_tile, capacity, expect);
        KCHECK(dbg, s, "tile_list_scatter");
    } else {
        tm.mark(GGR_FWD_TILE_SCATTER);
    }

    if (out->backward_scratch && tiles == 0)
        HIP_TRY(hipMemsetAsync(out->backward_scratch, 0, ggr_carve_bwd(nullptr, (size_t)P1, (size_t)NV).bytes, s));
    blend_pass();
    KCHECK(dbg, s, "blend_fwd");
    tm.mark(GGR_FWD_BLEND);
    if (hinted && P > 0 && tiles > 0) {   // num_rendered, while the device works on scatter and blend; the guesses must have held
        out->num_rendered = 0;
        { const int rc = read_counts(); if (rc != GGR_OK) return rc; }
        const bool cut = num_rendered > capacity;
        const bool too_long = (per_tile && longest > GGR_TSORT_CAP_LARGE) || (!per_tile && bucket_fault);   // → the lists once more, three-pass global sort
        const bool unsorted_left = per_tile && !too_long && longest > sorted_upto;   // the length guess was too small
        if (cut) {
            // The guess did not hold.  Everything up to the tile ranges is valid (the counts do not depend on the list
            // buffer); only the ranges and the lists were cut at the guessed capacity and the blend ran on cut lists.  With an allocator at hand the call repairs
            // itself: the exact buffer (the allocator's second call, as in the exact mode), the tile ranges, the scatter and the blend once
            // more — 0.21 ms at C3 instead of a whole second forward (round 5; until then GGR_E_CAPACITY and the host
            // repeated the call: 1.25 instead of 0.90 ms).  capacity_is_hint = 2 tells the host that this happened.
            void* bin_mem = alloc ? alloc(alloc_ctx, ggr_point_list_bytes(num_rendered) +
                                                         ((per_tile && !too_long) ? ggr_pair_list_bytes(num_rendered) : 0)) : nullptr;
            if (!bin_mem)
                return fail(GGR_E_CAPACITY, "num_rendered %u exceeds the hinted capacity %u", num_rendered, capacity);
            if (per_tile && !too_long) pair_list = (uint2*)((char*)bin_mem + ggr_point_list_bytes(num_rendered));
            out->binning_buffer = bin_mem;
            out->binning_capacity = (int64_t)num_rendered;
            out->capacity_is_hint = 2;
            point_list = (uint32_t*)bin_mem;
            capacity = 0xFFFFFFFFu;
        }
        if (too_long) {
            rebuild_global();    // (also rewrites the ranges, uncut)
        } else if (cut) {
            ggr::launch_tile_list_ranges(plan, tiles, work, im.ranges, s);   // (they were cut at the guessed capacity)
            // … which also leaves the status word's overflow bit standing: cleared below
        }
        if (cut || too_long) {
            if (!too_long) {
                HIP_TRY(hipMemsetAsync(g.counters + 1, 0, 4, s));   // the frame is complete after all (ggr_forward_status)
                HIP_TRY(hipMemsetAsync(g.counters + 3, 0, 4, s));   // (the per-tile sort below counts its slow route anew)
            }
            scatter_pass(per_tile, 0xFFFFFFFFu, longest);
            KCHECK(dbg, s, "tile_list_scatter (repair)");
        } else if (unsorted_left) {
            tile_sort_upto(longest);
            KCHECK(dbg, s, "tile_depth_sort (repair)");
        }
        if (cut || too_long || unsorted_left) {
            if (!cut) out->capacity_is_hint = 3;   // repaired for the list LENGTH guess alone: the caller's buffer stays
            blend_pass();
            KCHECK(dbg, s, "blend_fwd (repair)");
        }
    } else if (hinted) {
        out->num_rendered = 0;
        out->max_list_len = 0;
    }
    if (rebuilt) {   // the global sort of the rebuild ran behind the read-back: its fault word is checked here (rare path: one sync)
        uint32_t two[2] = {0u, 0u};
        HIP_TRY(hipMemcpyAsync(two, g.counters, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (two[1] & 2u) return fail(GGR_E_HIP, "%s", kSpinFault);
        if (two[1] & 4u) return fail(GGR_E_LIMIT, "%s", kRangeFault);
    }
    if (bucket_slow && out->depth_sort_used == GGR_DEPTH_SORT_GLOBAL) out->depth_sort_used = GGR_DEPTH_SORT_GLOBAL_SLOW;
    tm.finish();
    if (out->stage_ms && side) {   // the colour kernel's own duration (it ran BESIDE stages 1-3, not in addition to them)
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, side->t0, side->t1) == hipSuccess) out->stage_ms[GGR_FWD_COLOUR] += ms;
        else (void)hipGetLastError();
    }
    return GGR_OK;
}

int backward_impl(const GgrSettings* st, const ViewSet& vs, const GgrBackwardIn* in, GgrBackwardOut* out, void* stream) {
    if (!out) return fail(GGR_E_INVALID, "null gradient output struct");
    const int NV = vs.V;
    if (st->num_points == 0) {  // nothing to differentiate; camera gradients are zero
        hipStream_t s0 = (hipStream_t)stream;
        if (out->dL_dviewmatrix) HIP_TRY(hipMemsetAsync(out->dL_dviewmatrix, 0, 64 * (size_t)NV, s0));
        if (out->dL_dprojmatrix) HIP_TRY(hipMemsetAsync(out->dL_dprojmatrix, 0, 64 * (size_t)NV, s0));
        if (out->dL_dcampos) HIP_TRY(hipMemsetAsync(out->dL_dcampos, 0, 12 * (size_t)NV, s0));
        return GGR_OK;
    }
    if (!out->dL_dmeans3D || !out->dL_dmeans2D || !out->dL_dopacities || !out->dL_dcov3D)
        return fail(GGR_E_INVALID, "null gradient output");
    if (!in->geom_buffer || !in->image_buffer || !in->scratch || !in->dL_dout_color)
        return fail(GGR_E_INVALID, "null saved buffer / scratch / upstream gradient");
    const bool has_cp = in->fwd.colors_precomp != nullptr;
    if (has_cp ? !out->dL_dcolors_precomp : !out->dL_dshs) return fail(GGR_E_INVALID, "null colour gradient output");
    const bool has_sr = in->fwd.scales != nullptr;
    if (has_sr && (!out->dL_dscales || !out->dL_drotations)) return fail(GGR_E_INVALID, "null scale/rotation gradient output");
    if (in->fwd.aux_precomp && in->dL_dout_depth && !out->dL_daux) return fail(GGR_E_INVALID, "null dL_daux output");
    const int npose = (out->dL_dviewmatrix != nullptr) + (out->dL_dprojmatrix != nullptr) + (out->dL_dcampos != nullptr);
    if (npose != 0 && npose != 3) return fail(GGR_E_INVALID, "camera gradients: give all three outputs or none");
    hipStream_t s = (hipStream_t)stream;
    const int P = st->num_points, W = st->image_width, H = st->image_height;
    const bool dbg = st->debug != 0;
    if (in->num_rendered != 0 && !in->binning_buffer) return fail(GGR_E_INVALID, "null binning buffer");

    GeomLayout g = ggr_carve_geom((void*)in->geom_buffer, (size_t)P * NV, ggr_sort_segments((size_t)NV));
    ImageLayout im = ggr_carve_image((void*)in->image_buffer, W, H, NV);
    const uint32_t* point_list = (const uint32_t*)in->binning_buffer;
    BwdScratch sc = ggr_carve_bwd(in->scratch, (size_t)P, (size_t)NV);

    StageTimer tm(s, out->stage_ms, GGR_BWD_STAGES);
    if (!in->scratch_zeroed)  // (else: this frame's forward cleared it inside its blend kernel)
        HIP_TRY(hipMemsetAsync(in->scratch, 0, sc.bytes, s));  // dL_dmeans2D / dL_dopacities are written by preprocess_bwd
    tm.mark(GGR_BWD_CLEAR);

    if (in->num_rendered != 0) {  // (-1: sync-free forward, count known on the device only)
        ggr::launch_blend_bwd(W, H, im.ranges, point_list, g.splat, g.colour, vs.bg, im.final_T, im.n_contrib,
                              in->dL_dout_color, in->dL_dout_depth, sc.grad2d, im.tile_top, im.ckpt,
                              im.ckpt_slots, im.bwd_segments, NV, s);
        KCHECK(dbg, s, "blend_bwd");
    }
    tm.mark(GGR_BWD_BLEND);
    const float* cov = in->fwd.cov3D_precomp ? in->fwd.cov3D_precomp : g.cov3D;
    ggr::launch_preprocess_bwd(P, st->sh_degree, st->sh_stride, in->fwd.means3D, in->fwd.shs, g.sh_jac, has_cp ? 1 : 0,
                               in->fwd.scales, in->fwd.rotations, st->scale_modifier, cov, vs, W, H, in->radii, g.clamped,
                               sc.grad2d, in->dL_dout_depth ? 1 : 0, out->dL_dmeans3D, out->dL_dmeans2D,
                               out->dL_dopacities, out->dL_dshs, out->dL_dcolors_precomp, out->dL_dcov3D,
                               out->dL_dscales, out->dL_drotations, in->fwd.aux_precomp ? out->dL_daux : nullptr,
                               npose ? sc.pose_acc : nullptr, out->dL_dviewmatrix, out->dL_dprojmatrix,
                               out->dL_dcampos, input_form(st, &in->fwd, vs.sets, out->dL_dshs), in->fwd.cov3D_precomp ? 1 : 0, s);
    KCHECK(dbg, s, "preprocess_bwd");
    tm.mark(GGR_BWD_PREPROCESS);
    return GGR_OK;
}

}  // namespace

extern "C" {

int ggr_forward(const GgrSettings* st, const GgrForwardIn* in, GgrForwardOut* out, GgrAllocFn alloc,
                void* alloc_ctx, void* stream) {
    g_err[0] = 0;
    int rc = validate(st, in);
    if (rc) return rc;
    return forward_impl(st, single_view(st, in), in, out, alloc, alloc_ctx, stream);
}

int ggr_backward(const GgrSettings* st, const GgrBackwardIn* in, GgrBackwardOut* out, void* stream) {
    g_err[0] = 0;
    if (!in) return fail(GGR_E_INVALID, "null inputs");
    int rc = validate(st, &in->fwd);
    if (rc) return rc;
    return backward_impl(st, single_view(st, &in->fwd), in, out, stream);
}

int ggr_forward_views(const GgrSettings* st, const GgrViews* views, const GgrForwardIn* in, GgrForwardOut* out,
                      GgrAllocFn alloc, void* alloc_ctx, void* stream) {
    g_err[0] = 0;
    int rc = validate(st, in);
    if (rc) return rc;
    ViewSet vs;
    if ((rc = view_set(st, views, &vs)) != 0) return rc;
    return forward_impl(st, vs, in, out, alloc, alloc_ctx, stream);
}

int ggr_backward_views(const GgrSettings* st, const GgrViews* views, const GgrBackwardIn* in, GgrBackwardOut* out,
                       void* stream) {
    g_err[0] = 0;
    if (!in) return fail(GGR_E_INVALID, "null inputs");
    int rc = validate(st, &in->fwd);
    if (rc) return rc;
    ViewSet vs;
    if ((rc = view_set(st, views, &vs)) != 0) return rc;
    return backward_impl(st, vs, in, out, stream);
}

size_t ggr_geom_bytes_inference(int32_t P, int32_t V) {
    const size_t v = (size_t)(V > 0 ? V : 1);
    return ggr_carve_geom(nullptr, (size_t)(P > 0 ? P : 0) * v, ggr_sort_segments(v), /*with_jac=*/false).bytes;
}
size_t ggr_image_bytes_inference(int32_t W, int32_t H, int32_t V) { return ggr_carve_image(nullptr, W, H, V > 0 ? V : 1).bytes_no_ckpt; }
size_t ggr_geom_bytes_views(int32_t P, int32_t V) {
    const size_t v = (size_t)(V > 0 ? V : 1);
    return ggr_carve_geom(nullptr, (size_t)(P > 0 ? P : 0) * v, ggr_sort_segments(v)).bytes;
}
size_t ggr_image_bytes_views(int32_t W, int32_t H, int32_t V) { return ggr_carve_image(nullptr, W, H, V > 0 ? V : 1).bytes; }
size_t ggr_work_bytes_views(int32_t P, int32_t W, int32_t H, int32_t V) {
    const size_t v = (size_t)(V > 0 ? V : 1);
    return ggr::plan_tile_lists((size_t)(P > 0 ? P : 0) * v, tiles_of(W, H) * v).work_bytes;
}
size_t ggr_backward_scratch_bytes_views(int32_t P, int32_t V) {
    return ggr_carve_bwd(nullptr, (size_t)(P > 0 ? P : 0), (size_t)(V > 0 ? V : 1)).bytes;
}

int ggr_camera_setup(int32_t n, const float* extrinsics, const float* intrinsics, const float* near, const float* far,
                     int32_t scale_invariant, float* viewmatrix, float* projmatrix, float* campos, float* tanfov,
                     float* scale, void* stream) {
    g_err[0] = 0;
    if (n < 0 || (n > 0 && (!extrinsics || !intrinsics || !near || !far || !viewmatrix || !projmatrix || !campos ||
                            !tanfov || !scale)))
        return fail(GGR_E_INVALID, "bad arguments");
    ggr::launch_camera_setup(n, extrinsics, intrinsics, near, far, scale_invariant, viewmatrix, projmatrix, campos,
                             tanfov, scale, (hipStream_t)stream);
    KCHECK(false, (hipStream_t)stream, "camera_setup");
    return GGR_OK;
}

int ggr_sort_stats_async(const void* geom_buffer, int32_t P, uint32_t* host_words, void* stream) {
    g_err[0] = 0;
    if (!geom_buffer || !host_words) return fail(GGR_E_INVALID, "null geom buffer / destination");
    GeomLayout g = ggr_carve_geom((void*)geom_buffer, (size_t)(P > 0 ? P : 0));
    // (the counters' place in the buffer depends on P alone: whether the forward was an inference one does not matter)
    HIP_TRY(hipMemcpyAsync(host_words, g.counters, 16, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return GGR_OK;
}

int ggr_forward_status(const void* geom_buffer, int32_t P, int64_t* num_rendered, int32_t* overflow, void* stream) {
    g_err[0] = 0;
    if (!geom_buffer) return fail(GGR_E_INVALID, "null geom buffer");
    GeomLayout g = ggr_carve_geom((void*)geom_buffer, (size_t)(P > 0 ? P : 0));
    uint32_t host[2] = {0u, 0u};
    HIP_TRY(hipMemcpyAsync(host, g.counters, 8, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (num_rendered) *num_rendered = (int64_t)host[0];
    // (bit 3: a tile list too long for the per-tile depth sort in a forward that could not fall back — the frame is as
    //  incomplete as one whose lists were cut)
    if (overflow) *overflow = (int32_t)((host[1] & 1u) | ((host[1] >> 3) & 1u));
    if (host[1] & 2u) return fail(GGR_E_HIP, "%s", kSpinFault);
    if (host[1] & 4u) return fail(GGR_E_LIMIT, "%s", kRangeFault);
    return GGR_OK;
}

namespace {
struct FakeQuery { int scenario; int calls; volatile uint32_t* word; timespec t0; double late_s; };
hipError_t fake_query(void* p) {
    FakeQuery* q = (FakeQuery*)p;
    q->calls++;
    switch (q->scenario) {
        case 0: if (q->calls == 3) *q->word = 1234u; return hipErrorNotReady;
        case 1: return q->calls < 2 ? hipErrorNotReady : hipErrorLaunchFailure;
        case 2: return hipErrorNotReady;
        case 4: return hipErrorNotReady;   // (the word arrives through fake_start, below)
        default: return hipSuccess;
    }
}
// scenario 4: the stream is busy with EARLIER work for `late_s` seconds (longer than the time bound), then the
// tile-list kernels get their turn and the word arrives a few polls later — must succeed
hipError_t fake_start(void* p) {
    FakeQuery* q = (FakeQuery*)p;
    timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double dt = (double)(t1.tv_sec - q->t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - q->t0.tv_nsec);
    if (dt < q->late_s) return hipErrorNotReady;
    *q->word = 4321u;
    return hipSuccess;
}
}  // namespace

int ggr_debug_readback_wait(int32_t scenario, double timeout_s, uint32_t* value) {
    g_err[0] = 0;
    if (scenario < 0 || scenario > 4 || !value) return fail(GGR_E_INVALID, "bad arguments");
    volatile uint32_t word = GGR_READBACK_ARMED;
    FakeQuery q{scenario, 0, &word, {}, 3.0 * timeout_s};
    clock_gettime(CLOCK_MONOTONIC, &q.t0);
    if (scenario == 4) return wait_readback(&word, fake_query, &q, timeout_s, value, fake_start, &q);
    return wait_readback(&word, fake_query, &q, timeout_s, value);
}

int ggr_debug_counters(uint64_t* out, int32_t reset) {
    g_err[0] = 0;
    if (!out) return fail(GGR_E_INVALID, "null output");
    unsigned long long f[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    HIP_TRY(hipDeviceSynchronize());
    ggr::blend_fwd_counters(f, reset);
    ggr::blend_bwd_counters(b, reset);
    for (int i = 0; i < 4; i++) { out[i] = f[i]; out[4 + i] = b[i]; }
#ifdef GGR_DEV_COUNTERS
    return GGR_OK;
#else
    return fail(GGR_E_INVALID, "this library was built without -DGGR_DEV_COUNTERS: the counters are zero");
#endif
}

int ggr_debug_host_slots(int32_t* readback_slots, int32_t* side_streams) {
    g_err[0] = 0;
    if (readback_slots) { SlotPool<ReadbackSlot>& p = SlotPool<ReadbackSlot>::get(); std::lock_guard<std::mutex> lk(p.mu); *readback_slots = p.created; }
    if (side_streams) { SlotPool<SideStream>& p = SlotPool<SideStream>::get(); std::lock_guard<std::mutex> lk(p.mu); *side_streams = p.created; }
    return GGR_OK;
}

int ggr_debug_copy(const void* src, void* dst, size_t bytes, int32_t blocks, void* stream) {
    g_err[0] = 0;
    if (!src || !dst || (bytes & 15) || (((uintptr_t)src | (uintptr_t)dst) & 15)) return fail(GGR_E_INVALID, "bad arguments (16-byte granularity)");
    ggr::launch_copy_f4(src, dst, bytes, blocks, (hipStream_t)stream);
    KCHECK(false, (hipStream_t)stream, "copy_f4");
    return GGR_OK;
}

int ggr_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float*, uint8_t* present,
                     void* stream) {
    g_err[0] = 0;
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) return fail(GGR_E_INVALID, "bad arguments");
    ggr::launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
    KCHECK(false, (hipStream_t)stream, "mark_visible");
    return GGR_OK;
}

int ggr_debug_unpack_geom(const void* geom_buffer, int32_t P, float* depth, float* xy, float* conic_opacity,
                          float* rgb, int32_t* tiles_touched, uint8_t* clamped, void* stream) {
    g_err[0] = 0;
    if (!geom_buffer) return fail(GGR_E_INVALID, "null geom buffer");
    GeomLayout g = ggr_carve_geom((void*)geom_buffer, (size_t)(P > 0 ? P : 0));
    ggr::launch_unpack_geom(g, P, depth, xy, conic_opacity, rgb, tiles_touched, clamped, (hipStream_t)stream);
    KCHECK(false, (hipStream_t)stream, "unpack_geom");
    return GGR_OK;
}

int ggr_debug_unpack_binning(const void* binning_buffer, const void* image_buffer, int64_t N, int32_t W, int32_t H,
                             uint32_t* point_list, int32_t* ranges, float* final_T, int32_t* n_contrib, void* stream) {
    g_err[0] = 0;
    hipStream_t s = (hipStream_t)stream;
    if (!image_buffer) return fail(GGR_E_INVALID, "null image buffer");
    ImageLayout im = ggr_carve_image((void*)image_buffer, W, H);
    const size_t tiles = tiles_of(W, H);
    if (point_list && N > 0) {
        if (!binning_buffer) return fail(GGR_E_INVALID, "null binning buffer");
        HIP_TRY(hipMemcpyAsync(point_list, binning_buffer, (size_t)N * 4, hipMemcpyDeviceToDevice, s));
    }
    if (ranges && tiles) HIP_TRY(hipMemcpyAsync(ranges, im.ranges, tiles * 8, hipMemcpyDeviceToDevice, s));
    if (final_T) HIP_TRY(hipMemcpyAsync(final_T, im.final_T, (size_t)W * H * 4, hipMemcpyDeviceToDevice, s));
    if (n_contrib) HIP_TRY(hipMemcpyAsync(n_contrib, im.n_contrib, (size_t)W * H * 4, hipMemcpyDeviceToDevice, s));
    return GGR_OK;
}

}  // extern "C"
