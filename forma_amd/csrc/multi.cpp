// multi.cpp — several MI355X behind ONE context (forma_hip_create_multi, include/forma_hip.h).
//
// The reference renderer is one object with one `render` call (forma/src/cpu/renderer.rs:61-84) whose parallelism is
// internal (a Rayon pool over tile rows, painter/mod.rs:741-776).  The same shape here: the caller sees one context and
// one forma_hip_render; inside, every device has a full single-device context (a "kid": own stream, own buffers, the
// whole scene) and a dedicated host thread, and a frame is
//
//   kid g: lines [cuts[g], cuts[g + 1])  --k_line_* / k_rasterize-->  its pixel segments, in line order
//          --k_owner_count / _scan / _scatter-->  G buckets, bucket r = segments whose tile row device r owns
//   ONE all-to-all over xGMI: RCCL, grouped ncclAllToAll of the padded buckets — each carries its {count, overflow} header
//          as its last word — on the kids' streams: equal splits, so no count has to reach the host first; every GPU pair
//          has its own link
//   kid g: received buckets, rank-major = global line order  --stable radix sort (in place through the chunk map), carry
//          pre-pass, painter-->  its band of tile rows  --hipMemcpy2D-->  its rows of the caller's buffer
//
// FRAMES IN FLIGHT (forma_hip_set_frames_in_flight on the multi-device context, round 4).  A band frame is ~250 us of kernels
// that are bound by their own dependent round trips, not by the chip: ONE frame at a time leaves every device mostly idle and
// puts two host barriers and the collective's enqueue on the critical path.  With F slots every device holds F frame slots
// (full contexts that borrow the device's scene: own stream, own per-frame and exchange buffers) and every slot has its own
// set of communicators, so the collectives of consecutive frames never share a queue.  A device-resident, cache-less frame is
// ENQUEUED on the next slot by the device threads — buckets, the slot's all-to-all, the owner's sort + paint, all stream-ordered
// — and the call returns; it is verified when its slot comes round again (or when any call needs its result), re-planned and
// re-run if a bucket outgrew the plan.  The host barriers between the two halves remain, but they only order ENQUEUES: the
// devices work on the other slots' frames meanwhile.  Frames into caller memory, cache frames and timed frames keep the
// synchronous contract (cpu/buffer/mod.rs:43-49).
//
// `tile_y` is the most significant key field and the cover carry never crosses tile rows (painter/mod.rs:518-522), so a
// band is a complete sort + paint problem; the stable partition and the rank-major concatenation keep every band's sorted
// stream bit-identical to the corresponding slice of the single-device stream.
//
// The PLAN (which lines a device rasterizes, which rows it owns, how big a bucket may get) is made from measurements of
// the scene itself on the first frame of a geometry / canvas size: prefix sums of the line lengths -> line shares of equal
// pixel-segment counts; per-device tile-row histograms -> bands of equal pixel-segment counts and the largest bucket.
// A frame whose buckets outgrow the plan fails on the device (FORMA_E_CAPACITY inside), the context re-plans and re-runs it.
//
// RCCL is loaded with dlopen on first use: single-device users of libforma_hip.so never map the 570 MB library, and the
// process may already hold one (PyTorch ships its own copy under the same soname; the loader then hands back that one).
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>

#include <dlfcn.h>
#include <rccl/rccl.h>            // types and prototypes only: every call goes through the table below

#include "ctx.h"

namespace {

// ---- RCCL, resolved at run time ------------------------------------------------------------------------------------------
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllToAll) AllToAll = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    char why[256] = {0};
};

RcclApi* rccl_api() {
    static std::mutex m;
    static RcclApi api;
    static bool tried = false;
    std::lock_guard<std::mutex> lock(m);
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) { api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (api.handle) break; }
        if (!api.handle) { snprintf(api.why, sizeof api.why, "dlopen(librccl): %s", dlerror()); return &api; }
#define RCCL_SYM(field, name) api.field = (decltype(api.field))dlsym(api.handle, name); \
        if (!api.field) { snprintf(api.why, sizeof api.why, "librccl lacks %s", name); api.handle = nullptr; return &api; }
        RCCL_SYM(CommInitAll, "ncclCommInitAll") RCCL_SYM(CommDestroy, "ncclCommDestroy") RCCL_SYM(GroupStart, "ncclGroupStart")
        RCCL_SYM(GroupEnd, "ncclGroupEnd") RCCL_SYM(AllToAll, "ncclAllToAll") RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
    }
    return &api;
}

// ---- host-side synchronisation of the per-device threads ------------------------------------------------------------------------
// Frames are short (a few hundred microseconds), so waiting threads spin first and only then sleep.
inline void cpu_relax() {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}
struct SpinBarrier {
    std::atomic<int> arrived{0};
    std::atomic<unsigned> phase{0};
    void wait(int n) {
        const unsigned ph = phase.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            arrived.store(0, std::memory_order_relaxed);
            phase.store(ph + 1, std::memory_order_release);
            return;
        }
        for (unsigned spins = 0; phase.load(std::memory_order_acquire) == ph; spins++) {
            if (spins < 20000) cpu_relax(); else std::this_thread::yield();
        }
    }
};

struct FrameJob {
    enum Mode { FULL, DEFER, SETTLE } mode = FULL;  // whole frame now | enqueue on a slot | complete a slot's deferred frame
    int slot = 0;
    uint8_t* dst = nullptr;
    uint32_t width = 0, height = 0;
    size_t stride = 0;
    uint8_t channels[4] = {0, 1, 2, 3};
    float clear[4] = {0, 0, 0, 0};
    bool has_crop = false;
    forma_rect_t crop = {0, 0, 0, 0};
    int cache_id = -1;
    bool timings = false;
};

// the caller's current device is the caller's business: every entry point of a multi-device context restores it (the single-
// device entry points it drives call hipSetDevice for each device in turn)
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

}  // namespace

struct MultiState {
    int G = 0, F = 1;                         // devices, frame slots
    forma_hip_ctx* owner = nullptr;
    forma_hip_ctx* kid[FORMA_MAX_RANKS] = {nullptr};
    int dev[FORMA_MAX_RANKS] = {0};
    bool duplicates = false;                  // a device is listed twice: rehearsal on one GPU, device copies instead of RCCL
    bool use_rccl = false;
    // LAYOUT of a frame across the devices (forma_hip_multi_layout, forma_hip.h).  EXCHANGE: the north star's — every device
    // rasterizes a share of the LINES, the pixel segments travel to the owner of their tile row (one all-to-all).  BANDS: no
    // exchange at all — every device holds the whole scene anyway, so it prepares ALL lines but culls them to its band of tile
    // rows in the frame's first kernel (LineSource::band_lo / band_hi: a line outside the band has length 0; the segments of a
    // line that crosses the band's edge are flagged out of it by k_rasterize), then runs the plain single-device frame on its
    // band.  What BANDS repeats on every device is k_line_len over all lines (14 us for the 1.6 M lines of the 4K scene); what
    // it saves is the three bucketing kernels, the collective, and k_sort_hist on the received stream (the rasterizer's fused
    // histograms need the keys it made itself).  AUTO picks by that trade (choose_layout).
    int layout_req = FORMA_LAYOUT_AUTO, layout = FORMA_LAYOUT_EXCHANGE;
    bool kids_banded = false;                 // the kids carry forma_hip_set_band restrictions (BANDS plan): an EXCHANGE plan lifts them
    ncclComm_t comm[FORMA_MAX_FRAMES_IN_FLIGHT][FORMA_MAX_RANKS] = {{nullptr}};   // one set of communicators per frame slot
    hipEvent_t ev_bucket[FORMA_MAX_FRAMES_IN_FLIGHT][FORMA_MAX_RANKS] = {{nullptr}};
    // the plan
    bool planned = false;
    uint32_t plan_w = 0, plan_h = 0, cap = 0;
    uint32_t edges[FORMA_MAX_RANKS + 1] = {0};
    size_t cuts[FORMA_MAX_RANKS + 1] = {0};
    uint32_t plans_made = 0;
    bool cache_used[32] = {false};
    // frame slots
    bool slot_pending[FORMA_MAX_FRAMES_IN_FLIGHT] = {false};
    FrameJob slot_job[FORMA_MAX_FRAMES_IN_FLIGHT];                          // what a pending slot was asked to render
    uint32_t slot_edges[FORMA_MAX_FRAMES_IN_FLIGHT][FORMA_MAX_RANKS + 1] = {{0}};   // the bands its frame was rendered with
    unsigned next_slot = 0;
    int last_slot = 0;
    // the per-device threads
    std::thread th[FORMA_MAX_RANKS];
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    std::atomic<uint64_t> job_gen{0};
    std::atomic<int> done{0};
    bool quit = false;
    FrameJob job;
    SpinBarrier bar;
    int rc_stage[FORMA_MAX_RANKS] = {0};      // result of the first half of the frame, read by everybody after barrier A
    int rc_coll = 0;                          // result of the collective (leader), read after barrier B
    int rc[FORMA_MAX_RANKS] = {0};
    forma_timings_t tm[FORMA_MAX_RANKS];
    forma_rect_t band_crop[FORMA_MAX_RANKS];
    // what the last frame was (inspection calls)
    bool last_valid = false;
    uint32_t last_w = 0, last_h = 0;
};

namespace {

#define MFAIL(code, what) fd_fail(ctx, code, what)

void copy_err(forma_hip_ctx* ctx, const forma_hip_ctx* from) { memcpy(ctx->err, from->err, sizeof ctx->err); }

// the context of device g's frame slot s (slot 0 is the device's own context, the others borrow its scene)
forma_hip_ctx* slot_ctx(const MultiState* M, int g, int s) {
    forma_hip_ctx* k = M->kid[g];
    return (s > 0 && (size_t)s < k->slots.size()) ? k->slots[s] : k;
}

// ---- planning (host logic; the Python launcher's copy is forma_amd/sharding.py) --------------------------------------------------
// lines 0..n cut into G contiguous ranges carrying (nearly) equal pixel-segment counts
void line_shares(const std::vector<uint32_t>& sums, int G, size_t* cuts) {
    const size_t n = sums.size();
    const double total = n ? (double)sums[n - 1] : 0.0;
    cuts[0] = 0;
    for (int r = 1; r < G; r++) {
        size_t c;
        if (total > 0) c = (size_t)(std::lower_bound(sums.begin(), sums.end(), (uint32_t)std::min(total * r / G, 4294967295.0)) - sums.begin()) + 1;
        else c = n * r / G;
        cuts[r] = std::min(std::max(c, cuts[r - 1]), n);
    }
    cuts[G] = n;
}

// tiles_h tile rows cut into G contiguous bands with (nearly) equal pixel-segment counts; with fewer rows than devices the
// first tiles_h bands hold one row each and the rest are empty
void band_edges(const std::vector<uint64_t>& hist, uint32_t tiles_h, int G, uint32_t* edges) {
    std::vector<double> cum(tiles_h);
    double acc = 0;
    for (uint32_t r = 0; r < tiles_h; r++) { acc += (double)hist[r]; cum[r] = acc; }
    edges[0] = 0;
    for (int r = 1; r < G; r++) {
        uint32_t e;
        if (acc > 0) e = (uint32_t)(std::lower_bound(cum.begin(), cum.end(), acc * r / G) - cum.begin()) + 1;
        else e = (uint32_t)((uint64_t)tiles_h * r / G);
        e = std::max(e, std::min(edges[r - 1] + 1, tiles_h));                  // every band keeps a row while rows are left ...
        const uint32_t left = (uint32_t)(G - r);                             // ... and leaves one for each band after it
        if (tiles_h >= left) e = std::min(e, tiles_h - left);
        e = std::max(e, edges[r - 1]);
        edges[r] = std::min(e, tiles_h);
    }
    edges[G] = tiles_h;
}

uint32_t pair_capacity(uint64_t max_pair) {          // 6 % slack, whole 2048-segment blocks
    const uint64_t c = max_pair + max_pair / 16 + 4096;
    return (uint32_t)std::min<uint64_t>((c + 2047) / 2048 * 2048, 0x3FFFFFFFull);
}

int create_slot_transport(MultiState* M, int s);
void fall_back_to_copies(MultiState* M, const char* why);

// EXCHANGE against BANDS, per device and frame, in bytes that cross HBM: the exchange re-touches the device's N / G pixel
// segments about four times (k_owner_count reads them, k_owner_scatter reads and writes them, k_sort_hist reads what arrived)
// and pays four more launches and the collective (a fixed 8 MB-equivalent each: ~2 us of a 4 TB/s stream); BANDS prepares the
// (G - 1) / G of the lines that are not the device's share once more (12 B each).  With the bench scenes (8-170 pixel segments per
// line) BANDS wins at every device count; EXCHANGE needs a scene of many short lines — more lines than pixel segments.
int choose_layout(size_t n_lines, uint64_t n_segments, int G) {
    const double exchange_extra = 32.0 * (double)n_segments / G + 5.0 * 8.0e6;
    const double bands_extra = 12.0 * (double)n_lines * (G - 1) / G;
    return bands_extra > exchange_extra ? FORMA_LAYOUT_EXCHANGE : FORMA_LAYOUT_BANDS;
}

// BANDS: tile-row bands of (nearly) equal pixel-segment counts from ONE row histogram of the whole scene (device 0); device g is
// restricted to its band (forma_hip_set_band), nobody to a line range.  No capacities: nothing can overflow.
int make_plan_bands(forma_hip_ctx* ctx, uint32_t width, uint32_t height) {
    MultiState* M = ctx->multi;
    const int G = M->G;
    const uint32_t tiles_h = (height + 15) / 16;
    int rc;
    for (int g = 0; g < G; g++) {
        if ((rc = fd_set_line_range(M->kid[g], false, 0, 0))) { copy_err(ctx, M->kid[g]); return rc; }
        if ((rc = forma_hip_set_band(M->kid[g], 0, 0))) { copy_err(ctx, M->kid[g]); return rc; }
    }
    std::vector<uint32_t> hist(2048, 0u);
    if ((rc = fd_row_histogram(M->kid[0], width, height, hist.data(), nullptr))) { copy_err(ctx, M->kid[0]); return rc; }
    std::vector<uint64_t> all(std::max<uint32_t>(tiles_h, 1), 0ull);
    for (uint32_t r = 0; r < tiles_h && r < 2047; r++) all[r] = hist[r];
    const bool had_plan = M->plans_made > 0 && M->plan_w == width && M->plan_h == height;
    uint32_t old_edges[FORMA_MAX_DEVICES + 1];
    for (int g = 0; g <= G; g++) old_edges[g] = M->edges[g];
    band_edges(all, tiles_h, G, M->edges);
    bool bands_moved = !had_plan;
    for (int g = 0; g <= G; g++) bands_moved |= old_edges[g] != M->edges[g];
    for (int g = 0; g < G; g++) {
        // (an empty band — more devices than tile rows — gets no restriction and no frames: device_frame skips it)
        if (M->edges[g] < M->edges[g + 1] && (rc = forma_hip_set_band(M->kid[g], M->edges[g], M->edges[g + 1]))) { copy_err(ctx, M->kid[g]); return rc; }
        for (int c = 0; c < 32 && bands_moved; c++)
            if (M->cache_used[c] && (rc = forma_hip_cache_clear(M->kid[g], c))) { copy_err(ctx, M->kid[g]); return rc; }
    }
    M->kids_banded = true;
    M->cap = 0;
    M->planned = true; M->plan_w = width; M->plan_h = height; M->plans_made++;
    return FORMA_OK;
}

int make_plan(forma_hip_ctx* ctx, uint32_t width, uint32_t height) {
    MultiState* M = ctx->multi;
    const int G = M->G;
    const uint32_t tiles_h = (height + 15) / 16;
    int rc;
    if (M->layout == FORMA_LAYOUT_BANDS) return make_plan_bands(ctx, width, height);
    if (M->kids_banded) {                                  // (the last plan was a BANDS plan: every device sees all tile rows again)
        for (int g = 0; g < G; g++) if ((rc = forma_hip_set_band(M->kid[g], 0, 0))) { copy_err(ctx, M->kid[g]); return rc; }
        M->kids_banded = false;
    }
    // the exchange's transport, made when the first EXCHANGE plan is: communicators (or events for the copy transport) per slot
    for (int sl = 0; sl < M->F; sl++) {
        rc = create_slot_transport(M, sl);
        if (rc == FORMA_E_COMM) {                          // (a working peer-copy exchange beats FORMA_E_COMM)
            fall_back_to_copies(M, rccl_api()->handle ? "ncclCommInitAll failed" : rccl_api()->why);
            rc = create_slot_transport(M, sl);
        }
        if (rc) return MFAIL(rc, "multi-device context: cannot create a frame slot's transport");
    }
    std::vector<uint32_t> sums;
    if ((rc = fd_line_sums(M->kid[0], width, height, sums))) { copy_err(ctx, M->kid[0]); return rc; }
    line_shares(sums, G, M->cuts);
    std::vector<uint32_t> hist((size_t)G * 2048, 0u);
    for (int g = 0; g < G; g++) {
        if ((rc = fd_set_line_range(M->kid[g], true, M->cuts[g], M->cuts[g + 1]))) { copy_err(ctx, M->kid[g]); return rc; }
        if ((rc = fd_row_histogram(M->kid[g], width, height, &hist[(size_t)g * 2048], nullptr))) { copy_err(ctx, M->kid[g]); return rc; }
    }
    std::vector<uint64_t> all(std::max<uint32_t>(tiles_h, 1), 0ull);
    for (int g = 0; g < G; g++) for (uint32_t r = 0; r < tiles_h && r < 2047; r++) all[r] += hist[(size_t)g * 2048 + r];
    const bool had_plan = M->plans_made > 0 && M->plan_w == width && M->plan_h == height;
    uint32_t old_edges[FORMA_MAX_DEVICES + 1];
    for (int g = 0; g <= G; g++) old_edges[g] = M->edges[g];
    band_edges(all, tiles_h, G, M->edges);
    bool bands_moved = !had_plan;
    for (int g = 0; g <= G; g++) bands_moved |= old_edges[g] != M->edges[g];
    uint64_t max_pair = 0;
    for (int s = 0; s < G; s++)
        for (int g = 0; g < G; g++) {
            uint64_t c = 0;
            for (uint32_t r = M->edges[g]; r < M->edges[g + 1] && r < 2047; r++) c += hist[(size_t)s * 2048 + r];
            max_pair = std::max(max_pair, c);
        }
    M->cap = pair_capacity(max_pair);
    if ((uint64_t)M->cap * G >= (1ull << 30)) return MFAIL(FORMA_E_CAPACITY, "multi-device plan: more than 2^30-1 bucket slots per device");
    for (int g = 0; g < G; g++) {
        for (int sl = 0; sl < M->F; sl++) {                  // every frame slot of the device: its own buckets
            forma_hip_ctx* k = slot_ctx(M, g, sl);
            if ((rc = forma_hip_exchange_plan(k, M->edges, (uint32_t)G, M->cap))) { copy_err(ctx, k); return rc; }
            k->xuse_recv = G == 1 && M->use_rccl;            // a world of one still runs its collective (the point of the rehearsal)
        }
        // a band that moved invalidates what a buffer-layer cache remembers about its rows: start over (everything repaints once).
        // (A plan made again for the same scene — after forma_hip_trim — lands on the same bands and keeps the caches.)
        for (int c = 0; c < 32 && bands_moved; c++)
            if (M->cache_used[c] && (rc = forma_hip_cache_clear(M->kid[g], c))) { copy_err(ctx, M->kid[g]); return rc; }
    }
    M->planned = true; M->plan_w = width; M->plan_h = height; M->plans_made++;
    return FORMA_OK;
}

// ---- one device's part of a frame (its own thread) ----------------------------------------------------------------------------------
void add_timings(forma_timings_t& t, const forma_timings_t& a) {
    t.prepare_us += a.prepare_us; t.rasterize_us += a.rasterize_us; t.sort_us += a.sort_us; t.carry_us += a.carry_us;
    t.paint_us += a.paint_us; t.total_us += a.total_us; t.d2h_us += a.d2h_us; t.exchange_us += a.exchange_us;
    t.sort_pass_us = std::max(t.sort_pass_us, a.sort_pass_us);
    t.n_lines += a.n_lines; t.n_sort_passes = std::max(t.n_sort_passes, a.n_sort_passes);
    t.n_segments = std::max(t.n_segments, a.n_segments);   // (the first half counts what was rasterized, the second what is sorted)
    t.n_runs += a.n_runs; t.n_tile_entries += a.n_tile_entries; t.n_tiles_written += a.n_tiles_written;
}

void device_frame(MultiState* M, int g) {
    const FrameJob& J = M->job;
    forma_hip_ctx* kid = slot_ctx(M, g, J.slot);
    const int G = M->G;
    if (M->layout == FORMA_LAYOUT_BANDS) {
        // the plain single-device frame of this device's band: forma_hip_render on the device's own context, cropped to the band
        // (it paints and copies out the band's rows only); with frame slots a device-resident frame is enqueued by the device's
        // own slot logic (api.cpp render_impl) and settled there
        forma_hip_ctx* k0 = M->kid[g];
        memset(&M->tm[g], 0, sizeof M->tm[g]);
        if (J.mode == FrameJob::SETTLE) { M->rc[g] = forma_hip_sync(k0); return; }
        const forma_rect_t& bc = M->band_crop[g];
        // (a band outside the caller's crop still rasterizes and sorts — an empty crop paints nothing — so that the context's
        //  sorted stream is whole, as in the EXCHANGE layout; only a device without a band, more devices than tile rows, sits out)
        if (M->edges[g] >= M->edges[g + 1]) { M->rc[g] = FORMA_OK; return; }
        M->rc[g] = forma_hip_render(k0, J.dst, J.width, J.height, J.stride, J.channels, J.clear, &bc, J.cache_id, J.timings ? &M->tm[g] : nullptr);
        return;
    }
    if (J.mode == FrameJob::SETTLE) {                       // complete this device's part of the slot's deferred frame
        M->rc[g] = fd_gsp_settle(kid);
        return;
    }
    forma_timings_t t1, t2;
    memset(&t1, 0, sizeof t1); memset(&t2, 0, sizeof t2);
    int rc = forma_hip_rasterize_bucket_frame(kid, J.width, J.height, J.timings ? &t1 : nullptr);
    if (rc == FORMA_OK && !M->use_rccl && G > 1 && hipEventRecord(M->ev_bucket[J.slot][g], kid->stream) != hipSuccess)
        rc = fd_fail(kid, FORMA_E_HIP, "hipEventRecord (bucket)");
    M->rc_stage[g] = rc;
    M->bar.wait(G);                                        // A: every device has enqueued its buckets
    bool ok = true;
    for (int r = 0; r < G; r++) ok = ok && M->rc_stage[r] == FORMA_OK;
    if (ok && M->use_rccl) {
        // The canonical single-process form: ONE thread issues the collective for every communicator inside a group, each
        // on its device's stream, behind the bucket kernels already enqueued there.  Equal splits (the pair capacity + the
        // bucket's header word {count, overflow}): no count has to reach the host before the exchange can be enqueued, and
        // ONE collective per frame moves everything.  Every frame slot has its own communicators.
        if (g == 0) {
            RcclApi* R = rccl_api();
            ncclResult_t r = R->GroupStart();
            for (int q = 0; q < G && r == ncclSuccess; q++) {
                forma_hip_ctx* k = slot_ctx(M, q, J.slot);
                r = R->AllToAll(k->xsend.p, k->xrecv.p, (size_t)M->cap + 1, ncclUint64, M->comm[J.slot][q], k->stream);   // data + header of every bucket
            }
            const ncclResult_t e = R->GroupEnd();
            if (r == ncclSuccess) r = e;
            M->rc_coll = FORMA_OK;
            if (r != ncclSuccess) {
                snprintf(kid->err, sizeof kid->err, "RCCL all-to-all: %s", R->GetErrorString(r));
                M->rc_coll = FORMA_E_COMM;
            }
        }
        M->bar.wait(G);                                    // B: the collective is enqueued on every stream
        if (M->rc_coll) { ok = false; rc = M->rc_coll; if (g) memcpy(kid->err, slot_ctx(M, 0, J.slot)->err, sizeof kid->err); }
    } else if (ok && G > 1) {
        // rehearsal / FORMA_HIP_DEBUG=xchg=copy / no usable RCCL: the same data movement with device copies — recv[g][s] =
        // send[s][g], each behind the sender's bucket kernels (event) and in front of this device's gather (stream order)
        for (int s = 0; s < G && rc == FORMA_OK; s++) {
            forma_hip_ctx* src = slot_ctx(M, s, J.slot);
            if (hipStreamWaitEvent(kid->stream, M->ev_bucket[J.slot][s], 0) != hipSuccess) { rc = fd_fail(kid, FORMA_E_HIP, "hipStreamWaitEvent"); break; }
            const size_t W = (size_t)M->cap + 1;          // a bucket: data + header
            hipError_t e1;
            if (M->dev[s] == M->dev[g])
                e1 = hipMemcpyAsync(kid->xrecv.as<uint64_t>() + (size_t)s * W, src->xsend.as<uint64_t>() + (size_t)g * W, W * 8,
                                    hipMemcpyDeviceToDevice, kid->stream);
            else
                e1 = hipMemcpyPeerAsync(kid->xrecv.as<uint64_t>() + (size_t)s * W, M->dev[g], src->xsend.as<uint64_t>() + (size_t)g * W,
                                        M->dev[s], W * 8, kid->stream);
            if (e1 != hipSuccess) rc = fd_fail(kid, FORMA_E_HIP, "bucket copy", e1);
        }
        ok = rc == FORMA_OK;
    }
    if (ok) {
        if (J.mode == FrameJob::DEFER)
            rc = fd_gsp_defer(kid, J.width, J.height, J.channels, J.clear, &M->band_crop[g]);
        else
            rc = fd_gather_sort_paint(kid, J.dst, J.width, J.height, J.stride, J.channels, J.clear, &M->band_crop[g], J.cache_id,
                                      J.timings ? &t2 : nullptr);
    } else if (rc == FORMA_OK) {
        rc = FORMA_E_STATE;                                // another device failed: this one did not paint
        snprintf(kid->err, sizeof kid->err, "another device of the context failed its part of the frame");
    }
    if (J.timings) { add_timings(t1, t2); M->tm[g] = t1; }
    M->rc[g] = rc;
}

void worker_main(MultiState* M, int g) {
    (void)hipSetDevice(M->dev[g]);
    uint64_t seen = 0;
    for (;;) {
        // spin briefly for the next frame (back-to-back frames), then sleep
        unsigned spins = 0;
        while (M->job_gen.load(std::memory_order_acquire) == seen) {
            if (++spins < 40000) { cpu_relax(); continue; }
            std::unique_lock<std::mutex> lock(M->m);
            M->cv_job.wait(lock, [&] { return M->job_gen.load(std::memory_order_acquire) != seen || M->quit; });
            break;
        }
        if (M->quit) return;
        seen = M->job_gen.load(std::memory_order_acquire);
        device_frame(M, g);
        if (M->done.fetch_add(1, std::memory_order_acq_rel) + 1 == M->G) {
            std::lock_guard<std::mutex> lock(M->m);
            M->cv_done.notify_all();
        }
    }
}

// hands M->job to the device threads and waits until every one is through with it.  *overflow: some device's owner half met a
// bucket beyond the pair capacity (the frame is void everywhere: re-plan and run it again)
int run_job(forma_hip_ctx* ctx, bool* overflow) {
    MultiState* M = ctx->multi;
    M->done.store(0, std::memory_order_release);
    {
        std::lock_guard<std::mutex> lock(M->m);
        M->job_gen.fetch_add(1, std::memory_order_acq_rel);
    }
    M->cv_job.notify_all();
    unsigned spins = 0;
    while (M->done.load(std::memory_order_acquire) != M->G) {
        if (++spins < 40000) { cpu_relax(); continue; }
        std::unique_lock<std::mutex> lock(M->m);
        M->cv_done.wait(lock, [&] { return M->done.load(std::memory_order_acquire) == M->G; });
    }
    // the most specific failure wins: a capacity overflow (re-plan) over a plain error over "another device failed"
    int first = FORMA_OK, pick = -1;
    bool over = false;
    for (int g = 0; g < M->G; g++) {
        const int r = M->rc[g];
        forma_hip_ctx* k = slot_ctx(M, g, M->job.slot);
        if (r == FORMA_E_CAPACITY && k->xoverflowed) over = true;
        if (r == FORMA_OK) continue;
        const bool better = pick < 0 || (r == FORMA_E_CAPACITY && first != FORMA_E_CAPACITY) || (first == FORMA_E_STATE && r != FORMA_E_STATE);
        if (better) { first = r; pick = g; }
    }
    if (pick >= 0) copy_err(ctx, slot_ctx(M, pick, M->job.slot));
    if (overflow) *overflow = over;
    return first;
}

// a device paints (and copies out) the intersection of its band with the crop, tile-rounded like Rect::new (renderer.rs:43-52)
void set_band_crops(MultiState* M, uint32_t width, uint32_t height, const forma_rect_t* crop_or_null) {
    const uint32_t tiles_h = (height + 15) / 16;
    for (int g = 0; g < M->G; g++) {
        uint32_t ty0 = M->edges[g], ty1 = M->edges[g + 1];
        uint32_t x0 = 0, x1 = width;
        if (crop_or_null) {
            ty0 = std::max(ty0, crop_or_null->y0 / 16); ty1 = std::min(ty1, std::min(tiles_h, (crop_or_null->y1 + 15) / 16));
            x0 = crop_or_null->x0; x1 = crop_or_null->x1;
        }
        if (ty0 >= ty1) { ty0 = M->edges[g]; ty1 = ty0; }              // nothing of the band is inside the crop
        M->band_crop[g] = forma_rect_t{x0, x1, ty0 * 16, std::min(ty1 * 16, height)};
        if (ty0 == ty1) M->band_crop[g].y1 = M->band_crop[g].y0;
    }
}

// devices work side by side: a stage takes as long as its slowest device; counts add up
void sum_timings(const MultiState* M, forma_timings_t* timings) {
    memset(timings, 0, sizeof *timings);
    for (int g = 0; g < M->G; g++) {
        const forma_timings_t& t = M->tm[g];
        timings->prepare_us = std::max(timings->prepare_us, t.prepare_us); timings->rasterize_us = std::max(timings->rasterize_us, t.rasterize_us);
        timings->sort_us = std::max(timings->sort_us, t.sort_us); timings->sort_pass_us = std::max(timings->sort_pass_us, t.sort_pass_us);
        timings->carry_us = std::max(timings->carry_us, t.carry_us); timings->paint_us = std::max(timings->paint_us, t.paint_us);
        timings->total_us = std::max(timings->total_us, t.total_us); timings->d2h_us = std::max(timings->d2h_us, t.d2h_us);
        timings->exchange_us = std::max(timings->exchange_us, t.exchange_us);
        timings->n_lines += t.n_lines; timings->n_segments += t.n_segments; timings->n_runs += t.n_runs;
        timings->n_tile_entries += t.n_tile_entries; timings->n_tiles_written += t.n_tiles_written;
        timings->n_sort_passes = std::max(timings->n_sort_passes, t.n_sort_passes);
    }
}

// One whole frame on frame slot `slot`, complete when it returns: plan if there is none, run, re-plan and run again when a
// bucket outgrew the plan.  The caller has settled every other slot if a new plan may be needed (a plan re-sizes the exchange
// buffers of ALL slots).
int full_frame(forma_hip_ctx* ctx, const FrameJob& want, forma_timings_t* timings) {
    MultiState* M = ctx->multi;
    const int G = M->G;
    for (int attempt = 0; attempt < 3; attempt++) {
        if (!M->planned || M->plan_w != want.width || M->plan_h != want.height) {
            const int rc = make_plan(ctx, want.width, want.height);
            if (rc) return rc;
        }
        M->job = want;
        M->job.mode = FrameJob::FULL;
        set_band_crops(M, want.width, want.height, want.has_crop ? &want.crop : nullptr);
        bool overflow = false;
        const int rc = run_job(ctx, &overflow);
        if (rc == FORMA_E_CAPACITY && overflow && attempt < 2) {            // a bucket outgrew the plan: measure again
            M->planned = false;
            continue;
        }
        if (rc) return rc;
        for (int g = 0; g <= G; g++) M->slot_edges[want.slot][g] = M->edges[g];
        M->last_valid = true; M->last_w = want.width; M->last_h = want.height; M->last_slot = want.slot;
        if (timings) sum_timings(M, timings);
        return FORMA_OK;
    }
    return MFAIL(FORMA_E_CAPACITY, "multi-device plan did not converge");
}

// Every deferred frame is completed.  A frame whose buckets outgrew the plan is void on every device: once ALL slots have been
// looked at (a new plan re-sizes everybody's exchange buffers) such frames are run again, whole, under a new plan.  The first
// error that remains is returned (and is the context's error text).
int drain_slots(forma_hip_ctx* ctx) {
    MultiState* M = ctx->multi;
    int first = FORMA_OK;
    char first_err[sizeof ctx->err] = {0};
    if (M->layout == FORMA_LAYOUT_BANDS) {                  // the devices' own frame slots hold what is in flight
        for (int g = 0; g < M->G; g++) {
            const int rc = forma_hip_sync(M->kid[g]);
            if (rc && !first) { first = rc; memcpy(first_err, M->kid[g]->err, sizeof first_err); }
        }
        if (first) memcpy(ctx->err, first_err, sizeof ctx->err);
        return first;
    }
    bool rerun[FORMA_MAX_FRAMES_IN_FLIGHT] = {false};
    bool any = false;
    for (int k = 0; k < M->F; k++) {
        const int s = (int)((M->next_slot + (unsigned)k) % (unsigned)M->F);    // oldest first
        if (!M->slot_pending[s]) continue;
        M->slot_pending[s] = false;
        M->job = M->slot_job[s];
        M->job.mode = FrameJob::SETTLE; M->job.slot = s;
        bool overflow = false;
        const int rc = run_job(ctx, &overflow);
        if (rc == FORMA_E_CAPACITY && overflow) { rerun[s] = true; any = true; continue; }
        if (rc && !first) { first = rc; memcpy(first_err, ctx->err, sizeof first_err); }
    }
    if (any) {
        M->planned = false;
        for (int k = 0; k < M->F; k++) {
            const int s = (int)((M->next_slot + (unsigned)k) % (unsigned)M->F);
            if (!rerun[s]) continue;
            FrameJob want = M->slot_job[s];
            want.slot = s; want.dst = nullptr; want.cache_id = -1; want.timings = false;
            const int rc = full_frame(ctx, want, nullptr);
            if (rc && !first) { first = rc; memcpy(first_err, ctx->err, sizeof first_err); }
        }
    }
    if (first) memcpy(ctx->err, first_err, sizeof ctx->err);
    return first;
}

int create_slot_transport(MultiState* M, int s) {          // communicators / events of frame slot s
    for (int g = 0; g < M->G; g++) {
        if (M->ev_bucket[s][g]) continue;
        (void)hipSetDevice(M->dev[g]);
        if (hipEventCreateWithFlags(&M->ev_bucket[s][g], hipEventDisableTiming) != hipSuccess) return FORMA_E_HIP;
    }
    if (M->use_rccl && !M->comm[s][0]) {
        RcclApi* R = rccl_api();
        if (!R->handle || R->CommInitAll(M->comm[s], M->G, M->dev) != ncclSuccess) {
            for (int g = 0; g < M->G; g++) M->comm[s][g] = nullptr;
            return FORMA_E_COMM;
        }
    }
    return FORMA_OK;
}
void destroy_slot_transport(MultiState* M, int s) {
    if (M->comm[s][0]) { RcclApi* R = rccl_api(); for (int g = 0; g < M->G; g++) if (M->comm[s][g] && R->handle) (void)R->CommDestroy(M->comm[s][g]); }
    for (int g = 0; g < M->G; g++) {
        M->comm[s][g] = nullptr;
        if (M->ev_bucket[s][g]) { (void)hipSetDevice(M->dev[g]); (void)hipEventDestroy(M->ev_bucket[s][g]); M->ev_bucket[s][g] = nullptr; }
    }
}
// No usable RCCL (library missing, communicator creation failed): the exchange still works with peer copies behind events —
// slower to enqueue, the same bytes over the same links.  Said once on stderr: a deployment wants to know.
void fall_back_to_copies(MultiState* M, const char* why) {
    for (int s = 0; s < FORMA_MAX_FRAMES_IN_FLIGHT; s++)
        if (M->comm[s][0]) { RcclApi* R = rccl_api(); for (int g = 0; g < M->G; g++) { if (M->comm[s][g] && R->handle) (void)R->CommDestroy(M->comm[s][g]); M->comm[s][g] = nullptr; } }
    M->use_rccl = false;
    if (!M->duplicates) {                                   // copies between distinct devices: peer access both ways
        for (int i = 0; i < M->G; i++) {
            (void)hipSetDevice(M->dev[i]);
            for (int j = 0; j < M->G; j++) if (j != i) (void)hipDeviceEnablePeerAccess(M->dev[j], 0);   // (already enabled is fine)
        }
        (void)hipGetLastError();
    }
    fprintf(stderr, "[forma_hip] multi-device context: RCCL is not usable (%s); pixel segments are exchanged with peer copies instead\n", why);
    M->planned = false;                                    // (xuse_recv of a world of one depends on the transport)
}

}  // namespace

// ---- entry points (dispatched from api.cpp when ctx->multi) -----------------------------------------------------------------------
int multi_create(forma_hip_ctx** out, const int* devices, int n) {
    *out = nullptr;
    DeviceGuard guard;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return FORMA_E_NO_DEVICE;
    for (int i = 0; i < n; i++) if (devices[i] < 0 || devices[i] >= count) return FORMA_E_ARG;
    forma_hip_ctx* ctx = new (std::nothrow) forma_hip_ctx();
    MultiState* M = new (std::nothrow) MultiState();
    if (!ctx || !M) { delete ctx; delete M; return FORMA_E_INTERNAL; }
    ctx->multi = M; ctx->device = devices[0];
    M->owner = ctx; M->G = n;
    for (int i = 0; i < n; i++) { M->dev[i] = devices[i]; for (int j = 0; j < i; j++) if (devices[j] == devices[i]) M->duplicates = true; }
    M->use_rccl = !M->duplicates && !forma_debug_parse().xchg_copy;
    int rc = FORMA_OK;
    for (int i = 0; i < n && rc == FORMA_OK; i++) rc = forma_hip_create(&M->kid[i], devices[i]);
    if (rc == FORMA_OK && !M->use_rccl && !M->duplicates) {             // copies between distinct devices: peer access both ways
        for (int i = 0; i < n; i++) {
            (void)hipSetDevice(devices[i]);
            for (int j = 0; j < n; j++) if (j != i) (void)hipDeviceEnablePeerAccess(devices[j], 0);   // (already enabled is fine)
        }
        (void)hipGetLastError();
    }
    // (the exchange's transport — RCCL communicators per frame slot, or events for the copy transport — is made by the first
    //  EXCHANGE plan: a context that only ever renders BANDS frames never loads librccl)
    {
        const ForMaDebug dbg = forma_debug_parse();
        if (dbg.multi_layout == 1) M->layout_req = FORMA_LAYOUT_EXCHANGE;
        else if (dbg.multi_layout == 2) M->layout_req = FORMA_LAYOUT_BANDS;
        if (n == 1) M->layout_req = FORMA_LAYOUT_EXCHANGE;      // (force_exchange: the rehearsal of the exchange path with a world of one)
        M->layout = M->layout_req == FORMA_LAYOUT_BANDS ? FORMA_LAYOUT_BANDS : FORMA_LAYOUT_EXCHANGE;
    }
    if (rc != FORMA_OK) { multi_destroy(ctx); return rc; }
    for (int g = 0; g < n; g++) M->th[g] = std::thread(worker_main, M, g);
    *out = ctx;
    return FORMA_OK;
}

void multi_destroy(forma_hip_ctx* ctx) {
    DeviceGuard guard;
    MultiState* M = ctx->multi;
    if (M) {
        if (M->th[0].joinable()) (void)drain_slots(ctx);
        {
            std::lock_guard<std::mutex> lock(M->m);
            M->quit = true;
            M->job_gen.fetch_add(1, std::memory_order_acq_rel);
        }
        M->cv_job.notify_all();
        for (int g = 0; g < M->G; g++) if (M->th[g].joinable()) M->th[g].join();
        for (int s = 0; s < FORMA_MAX_FRAMES_IN_FLIGHT; s++) destroy_slot_transport(M, s);
        for (int g = 0; g < M->G; g++) if (M->kid[g]) forma_hip_destroy(M->kid[g]);
        delete M;
    }
    ctx->multi = nullptr;
    delete ctx;
}

forma_hip_ctx* multi_first(forma_hip_ctx* ctx) { return ctx->multi->kid[0]; }

int multi_sync(forma_hip_ctx* ctx) {
    DeviceGuard guard;
    return drain_slots(ctx);
}

int multi_set_frames_in_flight(forma_hip_ctx* ctx, int n) {
    DeviceGuard guard;
    MultiState* M = ctx->multi;
    int rc = drain_slots(ctx);
    if (rc) return rc;
    for (int g = 0; g < M->G; g++)
        if ((rc = forma_hip_set_frames_in_flight(M->kid[g], n))) { copy_err(ctx, M->kid[g]); return rc; }
    for (int s = n; s < FORMA_MAX_FRAMES_IN_FLIGHT; s++) destroy_slot_transport(M, s);
    // (the new slots' communicators / events are made by the next EXCHANGE plan: make_plan)
    M->F = n; M->next_slot = 0; M->last_slot = 0;
    M->planned = false;                                    // the new slots need their exchange buffers: plan again
    return FORMA_OK;
}

void multi_info(forma_hip_ctx* ctx, forma_context_info_t* out) {
    MultiState* M = ctx->multi;
    out->n_devices = (uint32_t)M->G; out->frames_in_flight = (uint32_t)M->F;
    out->transport = M->layout == FORMA_LAYOUT_BANDS ? FORMA_TRANSPORT_NONE
                     : (M->G > 1 || M->use_rccl ? (M->use_rccl ? FORMA_TRANSPORT_RCCL : FORMA_TRANSPORT_COPY) : FORMA_TRANSPORT_NONE);
    out->layout = (uint32_t)M->layout;
    for (int g = 0; g < M->G && g < FORMA_MAX_DEVICES; g++) out->devices[g] = M->dev[g];
}

int multi_set_layout(forma_hip_ctx* ctx, int layout) {
    DeviceGuard guard;
    MultiState* M = ctx->multi;
    if (layout != FORMA_LAYOUT_AUTO && layout != FORMA_LAYOUT_EXCHANGE && layout != FORMA_LAYOUT_BANDS) return MFAIL(FORMA_E_ARG, "layout: FORMA_LAYOUT_AUTO / _EXCHANGE / _BANDS");
    { const int rc = drain_slots(ctx); if (rc) return rc; }
    M->layout_req = layout;
    if (layout != FORMA_LAYOUT_AUTO) M->layout = layout;   // (what forma_hip_context_info says until the next plan; AUTO: decided then)
    M->planned = false;                                    // the next frame plans for the layout asked for
    return FORMA_OK;
}

#define EACH_KID(call)                                                              \
    do {                                                                            \
        MultiState* M = ctx->multi;                                                 \
        for (int g = 0; g < M->G; g++) {                                            \
            forma_hip_ctx* k = M->kid[g];                                           \
            const int rc = (call);                                                  \
            if (rc) { copy_err(ctx, k); return rc; }                                \
        }                                                                           \
    } while (0)
#define MULTI_ENTER()                                                               \
    DeviceGuard guard;                                                              \
    { const int _rc = drain_slots(ctx); if (_rc) return _rc; }

int multi_set_geometry(forma_hip_ctx* ctx, const float* x, const float* y, const uint32_t* line_slot, size_t n_points) {
    MULTI_ENTER();
    ctx->multi->planned = false;                          // new geometry: new line shares, new bands
    EACH_KID(forma_hip_set_geometry(k, x, y, line_slot, n_points));
    EACH_KID(fd_set_line_range(k, false, 0, 0));
    return FORMA_OK;
}
int multi_set_geoms(forma_hip_ctx* ctx, const forma_geom_t* geoms, size_t n_geoms) {
    // (transforms move segments between bands: the plan stays — its capacity carries 6 % slack and a frame that outgrows it
    //  re-plans; a layer that is switched on or off changes far less than that in practice)
    MULTI_ENTER();
    EACH_KID(forma_hip_set_geoms(k, geoms, n_geoms));
    return FORMA_OK;
}
int multi_set_styles(forma_hip_ctx* ctx, const uint32_t* style_offsets, size_t n_orders, const uint32_t* style_words, size_t n_words,
                     const uint8_t* unchanged) {
    MULTI_ENTER();
    EACH_KID(forma_hip_set_styles(k, style_offsets, n_orders, style_words, n_words, unchanged));
    return FORMA_OK;
}
int multi_set_images(forma_hip_ctx* ctx, const forma_image_t* images, size_t n_images, const uint16_t* texels, size_t n_texels) {
    MULTI_ENTER();
    EACH_KID(forma_hip_set_images(k, images, n_images, texels, n_texels));
    return FORMA_OK;
}
int multi_trim(forma_hip_ctx* ctx) {
    MULTI_ENTER();
    EACH_KID(forma_hip_trim(k));
    ctx->multi->planned = false;                           // (the plan's measurements live in the kids' buffers: plan again)
    ctx->multi->last_valid = false;
    return FORMA_OK;
}

int multi_cache_clear(forma_hip_ctx* ctx, int cache_id) {
    MULTI_ENTER();
    EACH_KID(forma_hip_cache_clear(k, cache_id));
    return FORMA_OK;
}

int multi_render(forma_hip_ctx* ctx, uint8_t* dst, uint32_t width, uint32_t height, size_t stride_bytes, const uint8_t channels[4],
                 const float clear_color[4], const forma_rect_t* crop_or_null, int cache_id, forma_timings_t* timings) {
    DeviceGuard guard;
    MultiState* M = ctx->multi;
    if (cache_id >= 0) M->cache_used[cache_id] = true;
    FrameJob want;
    want.dst = dst; want.width = width; want.height = height; want.stride = stride_bytes;
    memcpy(want.channels, channels, 4); memcpy(want.clear, clear_color, 16);
    want.has_crop = crop_or_null != nullptr; if (crop_or_null) want.crop = *crop_or_null;
    want.cache_id = cache_id; want.timings = timings != nullptr;
    if (!M->planned || M->plan_w != width || M->plan_h != height) {
        // a plan is due: settle what is in flight under the old one, then decide the layout the new plan is made for
        { const int rc = drain_slots(ctx); if (rc) return rc; }
        int lay = M->layout_req;
        if (lay == FORMA_LAYOUT_AUTO) {
            std::vector<uint32_t> sums;
            const int rc = fd_line_sums(M->kid[0], width, height, sums);
            if (rc) { copy_err(ctx, M->kid[0]); return rc; }
            lay = choose_layout(sums.size(), sums.empty() ? 0u : sums.back(), M->G);
        }
        M->layout = lay;
    }
    if (M->layout == FORMA_LAYOUT_BANDS) {
        if (!M->planned || M->plan_w != width || M->plan_h != height) { const int rc = make_plan(ctx, width, height); if (rc) return rc; }
        M->job = want;
        M->job.mode = FrameJob::FULL; M->job.slot = 0;
        set_band_crops(M, width, height, crop_or_null);
        const int rc = run_job(ctx, nullptr);
        if (rc) return rc;
        for (int g = 0; g <= M->G; g++) M->slot_edges[0][g] = M->edges[g];
        M->last_valid = true; M->last_w = width; M->last_h = height; M->last_slot = 0;
        if (timings) sum_timings(M, timings);
        return FORMA_OK;
    }
    // Frames in flight: a device-resident frame without a cache is ENQUEUED on the next frame slot of every device and this
    // call returns; it is verified when the slot comes round again or when any call needs its result.  Everything else keeps
    // the synchronous contract of the reference: `dst` is fully written when the call returns.
    if (M->F > 1 && !dst && cache_id < 0 && !timings) {
        const int s = (int)(M->next_slot % (unsigned)M->F);
        int rc;
        if (M->slot_pending[s] || !M->planned || M->plan_w != width || M->plan_h != height) {
            // the slot still owes a frame — or a plan has to be made, which re-sizes every slot's exchange buffers: settle
            // (a settle that has to re-plan settles everybody; the common case is one SETTLE job on this slot)
            bool simple = M->slot_pending[s] && M->planned && M->plan_w == width && M->plan_h == height;
            if (simple) {
                M->slot_pending[s] = false;
                M->job = M->slot_job[s];
                M->job.mode = FrameJob::SETTLE; M->job.slot = s;
                bool overflow = false;
                rc = run_job(ctx, &overflow);
                if (rc == FORMA_E_CAPACITY && overflow) {                  // void under this plan: everybody settles, then it runs again
                    if ((rc = drain_slots(ctx))) return rc;
                    M->planned = false;
                    FrameJob again = M->slot_job[s];
                    again.slot = s; again.dst = nullptr; again.cache_id = -1; again.timings = false;
                    if ((rc = full_frame(ctx, again, nullptr))) return rc;
                } else if (rc) return rc;
            } else if ((rc = drain_slots(ctx))) return rc;
        }
        M->next_slot++;
        want.slot = s;
        if (!M->planned || M->plan_w != width || M->plan_h != height) {    // first frame of a plan: whole and synchronous
            return full_frame(ctx, want, nullptr);
        }
        M->job = want;
        M->job.mode = FrameJob::DEFER;
        set_band_crops(M, width, height, crop_or_null);
        bool overflow = false;
        rc = run_job(ctx, &overflow);
        if (rc == FORMA_E_CAPACITY && overflow) {                          // (a slot without predictions ran synchronously and overflowed)
            for (int g = 0; g < M->G; g++) slot_ctx(M, g, s)->xpending = false;
            if ((rc = drain_slots(ctx))) return rc;
            M->planned = false;
            return full_frame(ctx, want, nullptr);
        }
        if (rc) { for (int g = 0; g < M->G; g++) slot_ctx(M, g, s)->xpending = false; return rc; }
        M->slot_job[s] = want;
        for (int g = 0; g <= M->G; g++) M->slot_edges[s][g] = M->edges[g];
        M->slot_pending[s] = true;
        M->last_valid = true; M->last_w = width; M->last_h = height; M->last_slot = s;
        return FORMA_OK;
    }
    { const int rc = drain_slots(ctx); if (rc) return rc; }
    want.slot = 0;
    return full_frame(ctx, want, timings);
}

int multi_read_segments(forma_hip_ctx* ctx, int which, uint64_t* out, size_t capacity, size_t* out_n) {
    MULTI_ENTER();
    MultiState* M = ctx->multi;
    *out_n = 0;
    if (which != 1) return MFAIL(FORMA_E_STATE, "a multi-device context holds no single unsorted stream (which = 1: the sorted stream of the painted rows)");
    if (!M->last_valid) return MFAIL(FORMA_E_STATE, "no frame rendered yet");
    const int s = M->last_slot;
    if (M->layout == FORMA_LAYOUT_BANDS) {
        // every device's sorted stream holds its band's rows plus what it flagged out of the band (tile row -1, in front): the
        // painted rows of the frame are the bands' own rows, concatenated in band order
        std::vector<uint64_t> part;
        size_t at = 0;
        for (int g = 0; g < M->G; g++) {
            const uint32_t e0 = M->slot_edges[s][g], e1 = M->slot_edges[s][g + 1];
            if (e0 >= e1) continue;
            forma_hip_ctx* k = M->kid[g];
            size_t n = 0;
            int rc = forma_hip_read_segments(k, 1, nullptr, 0, &n);
            if (rc && rc != FORMA_E_CAPACITY) { copy_err(ctx, k); return rc; }
            part.resize(n);
            if (n && (rc = forma_hip_read_segments(k, 1, part.data(), n, &n))) { copy_err(ctx, k); return rc; }
            const auto lo = std::lower_bound(part.begin(), part.end(), (uint64_t)(e0 + 1) << 53);
            const auto hi = std::lower_bound(part.begin(), part.end(), (uint64_t)(e1 + 1) << 53);
            const size_t m = (size_t)(hi - lo);
            if (out && at + m <= capacity) memcpy(out + at, &*part.begin() + (lo - part.begin()), m * 8);
            at += m;
        }
        *out_n = at;
        if (at > capacity) return MFAIL(FORMA_E_CAPACITY, "segment capacity too small");
        if (at && !out) return FORMA_E_ARG;
        return FORMA_OK;
    }
    size_t total = 0;
    for (int g = 0; g < M->G; g++) total += slot_ctx(M, g, s)->n_seg;
    *out_n = total;
    if (total > capacity) return MFAIL(FORMA_E_CAPACITY, "segment capacity too small");
    if (total == 0) return FORMA_OK;
    if (!out) return FORMA_E_ARG;
    size_t at = 0;
    for (int g = 0; g < M->G; g++) {                      // bands ascend in tile_y, the most significant key field
        size_t n = 0;
        forma_hip_ctx* k = slot_ctx(M, g, s);
        const int rc = fd_read_sorted(k, out + at, capacity - at, &n);
        if (rc) { copy_err(ctx, k); return rc; }
        at += n;
    }
    return FORMA_OK;
}

int multi_read_image(forma_hip_ctx* ctx, uint8_t* dst, size_t stride_bytes) {
    MULTI_ENTER();
    MultiState* M = ctx->multi;
    if (!M->last_valid) return MFAIL(FORMA_E_STATE, "no image on the device");
    if ((size_t)M->last_w * 4 > stride_bytes) return MFAIL(FORMA_E_ARG, "width exceeds width stride");
    const int s = M->last_slot;
    for (int g = 0; g < M->G; g++) {
        if (M->layout == FORMA_LAYOUT_BANDS && M->slot_edges[s][g] >= M->slot_edges[s][g + 1]) continue;
        forma_hip_ctx* k = M->layout == FORMA_LAYOUT_BANDS ? fd_last_slot(M->kid[g]) : slot_ctx(M, g, s);
        const int rc = fd_copy_image_rows(k, dst, stride_bytes, M->slot_edges[s][g] * 16, std::min(M->slot_edges[s][g + 1] * 16, M->last_h));
        if (rc) { copy_err(ctx, k); return rc; }
    }
    return FORMA_OK;
}

int multi_tiles_written(forma_hip_ctx* ctx, uint8_t* flags, size_t n_tiles) {
    MULTI_ENTER();
    MultiState* M = ctx->multi;
    if (!M->last_valid) return MFAIL(FORMA_E_STATE, "no frame rendered yet");
    const uint32_t tiles_w = (M->last_w + 15) / 16, tiles_h = (M->last_h + 15) / 16;
    const size_t T = (size_t)tiles_w * tiles_h;
    if (n_tiles < T) return MFAIL(FORMA_E_CAPACITY, "tile flag capacity too small");
    memset(flags, 0, n_tiles);
    std::vector<uint8_t> tmp(T);
    const int s = M->last_slot;
    for (int g = 0; g < M->G; g++) {
        const uint32_t e0 = M->slot_edges[s][g], e1 = M->slot_edges[s][g + 1];
        if (e0 >= e1) continue;
        forma_hip_ctx* k = M->layout == FORMA_LAYOUT_BANDS ? fd_last_slot(M->kid[g]) : slot_ctx(M, g, s);
        const int rc = fd_tiles_written(k, tmp.data(), T);
        if (rc) { copy_err(ctx, k); return rc; }
        for (uint32_t ty = e0; ty < e1 && ty < tiles_h; ty++)
            memcpy(flags + (size_t)ty * tiles_w, tmp.data() + (size_t)ty * tiles_w, tiles_w);
    }
    return FORMA_OK;
}
