// C-ABI entry points (include/gsrast_amd.h): chunk layout, gsr_forward orchestration,
// error reporting and the stage-level entry points the parity tests call.
//
// gsr_forward follows reference apps/gsrast/gscuda/GSCuda.cu:695-811 (gscuda::forward);
// the chunk carving follows AuxBuffer.cu:13-21 (obtain) and :44-89 (fromChunk).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <vector>

#include "gsr_common.hpp"
#include "blockbin.hpp"
#include "radix_sort.hpp"

namespace gsr {

static thread_local int g_last_error = GSR_OK;
static thread_local char g_hip_error[256] = "";

void set_hip_error(hipError_t e, const char* what) {
    snprintf(g_hip_error, sizeof(g_hip_error), "%s: %s", what, hipGetErrorString(e));
}

int record_error(int code) { g_last_error = code; return code; }

namespace {

// obtain(): AuxBuffer.cu:13-21 — align the running pointer up, hand out `bytes`.
template <typename T>
inline void obtain(char*& chunk, T*& out, size_t bytes, size_t align = 128) {
    const size_t offset = reinterpret_cast<size_t>(chunk);
    const size_t aligned = align * ((offset + align - 1) / align);
    out = reinterpret_cast<T*>(aligned);
    chunk = reinterpret_cast<char*>(aligned + bytes);
}

inline int fail(int code) { g_last_error = code; return code; }

inline size_t align128(size_t v) { return (v + 127) / 128 * 128; }

// Layout of GeometryState::scanningSpace (the reference keeps CUB's scan temp there,
// AuxBuffer.cu:49-51; this library keeps all of its per-Gaussian scratch there).
struct GeoScratch {
    char* scan_temp;          // partial sums of the two prefix scans
    uint32_t* depth_key;      // u32[N] depth bits or ~0 (written by preprocess)
    uint32_t* rect_idx;       // u32[N] packed band-clipped rectangle in index order (written by preprocess)
    uint32_t* sort_info;      // [0] distinct top-byte digits of the visible depth keys, [1] visible Gaussians V,
                              // [2..3] u64: sum of tilesTouched without the u32 wrap-around, [4] tiles with a list
    uint32_t* vis_partial;    // per 4096-key chunk: visible keys before it (compaction)
    uint32_t* main_partial;   // the same for the keys with the main top byte only (the depth order's side way, radix_sort.hip)
    uint32_t* big_partial;    // per 4096 Gaussians: the instances of those that touch kBigSplatTiles tiles or more
    uint4* wave_sums;         // per 64 Gaussians: {tilesTouched summed, with a tile, instances of the big ones, with another top byte} (written by the preprocess, summed by the scan)
    uint32_t *side_k, *side_v, *side_r;   // the side list (kDepthSideMax entries); its words: sort_info[8..10]
    uint32_t *c_k, *c_v;      // the visible (depth key, index) pairs in index order: the sort's input
    uint32_t *a_k, *a_v;      // depth-sort ping (three passes end here; kDepthSideMax elements of room in front of each a_*)
    uint32_t *b_k, *b_v;      // depth-sort pong = result (sorted depth bits, sorted index)
    uint32_t *c_r, *a_r, *b_r;  // the packed rectangles of the same Gaussians, moved with the pairs (tile grids up to 255 x 255)
    SweepScratch sweep;       // onesweep status words for the N-sized sort: pass 0 (+ error word, digit histograms)
    SweepScratch sweep_more[3];   // passes 1-3: their own look-back words, so one clear up front covers all four
    char* emit_scratch;       // column-major emission: [chunk][column] table, block partials, column starts
    char* block_scratch;      // block binning: [chunk][block] table, partials, block meta, tile counts / starts
    size_t bytes;
};
GeoScratch carve_geo_scratch(char* base, size_t n) {
    GeoScratch g;
    size_t off = 0;
    g.scan_temp = base + off; off += align128(scan_temp_bytes(n));
    g.depth_key = reinterpret_cast<uint32_t*>(base + off); off += align128(4 * n);
    g.rect_idx = reinterpret_cast<uint32_t*>(base + off); off += align128(4 * n);
    g.sort_info = reinterpret_cast<uint32_t*>(base + off); off += 128;
    g.vis_partial = reinterpret_cast<uint32_t*>(base + off); off += depth_compact_scratch_bytes(n);
    g.main_partial = reinterpret_cast<uint32_t*>(base + off); off += depth_compact_scratch_bytes(n);
    g.big_partial = reinterpret_cast<uint32_t*>(base + off); off += depth_compact_scratch_bytes(n);
    g.wave_sums = reinterpret_cast<uint4*>(base + off); off += align128(16 * ((n + 63) / 64));
    g.side_k = reinterpret_cast<uint32_t*>(base + off); off += align128(4 * kDepthSideMax);
    g.side_v = reinterpret_cast<uint32_t*>(base + off); off += align128(4 * kDepthSideMax);
    g.side_r = reinterpret_cast<uint32_t*>(base + off); off += align128(4 * kDepthSideMax);
    // (three groups of three consecutive arrays: between its passes the depth order keeps the triples as 12-byte RECORDS in
    // the room of a group — {a_k, a_v, a_r} and {b_k, b_v, b_r}, or {c_k, c_v, c_r} where there is no compaction to fill them)
    g.c_k = reinterpret_cast<uint32_t*>(base + off); off += align128(4 * n);
    g.c_v = reinterpret_cast<uint32_t*>(base + off); off += align128(4 * n);
    g.c_r = reinterpret_cast<uint32_t*>(base + off); off += align128(4 * n);
    off += 4 * kDepthSideMax; g.a_k = reinterpret_cast<uint32_t*>(base + off); off += align128(4 * n);
    off += 4 * kDepthSideMax; g.a_v = reinterpret_cast<uint32_t*>(base + off); off += align128(4 * n);
    off += 4 * kDepthSideMax; g.a_r = reinterpret_cast<uint32_t*>(base + off); off += align128(4 * n);
    g.b_k = reinterpret_cast<uint32_t*>(base + off); off += align128(4 * n);
    g.b_v = reinterpret_cast<uint32_t*>(base + off); off += align128(4 * n);
    g.b_r = reinterpret_cast<uint32_t*>(base + off); off += align128(4 * n);
    g.sweep = carve_sweep_scratch(base + off, n); off += sweep_scratch_bytes(n);
    for (auto& sw : g.sweep_more) { sw = carve_sweep_scratch(base + off, n); off += sweep_scratch_bytes(n); }
    g.emit_scratch = base + off; off += align128(emit_scratch_bytes(n));
    g.block_scratch = base + off; off += align128(blockbin_geo_bytes(n));
    g.bytes = off;
    return g;
}

// Layout of BinningState::sortingSpace: one scratch copy of the pairs + onesweep status.
struct BinScratch {
    uint64_t* tmp_k;
    uint32_t* tmp_v;
    SweepScratch sweep;       // pass 1 (also holds the two tile-digit histograms)
    SweepScratch sweep2;      // pass 2: its own look-back words, so both clears precede pass 1
    size_t bytes;
};
BinScratch carve_bin_scratch(char* base, size_t r) {
    BinScratch b;
    size_t off = 0;
    b.tmp_k = reinterpret_cast<uint64_t*>(base + off); off += align128(8 * r);
    b.tmp_v = reinterpret_cast<uint32_t*>(base + off); off += align128(4 * r);
    b.sweep = carve_sweep_scratch(base + off, r); off += sweep_scratch_bytes(r);
    b.sweep2 = carve_sweep_scratch(base + off, r); off += sweep_scratch_bytes(r);
    b.bytes = off;
    return b;
}

// Pinned landing zone for the numRendered read-back plus the events / side stream of a call: one per host
// thread AND device (events and streams belong to the device that was current when they were created). These are
// resources, not state: nothing a later call needs to know about an earlier one is kept here — that travels in the
// gsr_forward_receipt, and lives in the caller's chunks.
}  // namespace
}  // namespace gsr

// A tile history (include/gsrast_amd.h, GSR_FLAG_NO_TILE_HISTORY): how long the tiles of a view's two last frames took, the
// order the next blend takes them in (TileOrder, blend_core.hpp), and what the host remembers of tile_order_kernel's
// statistics. One per view — the caller's own (gsr_tile_history_create), or one the library keeps per host thread, device
// and stream. It decides WHEN a tile is composited, never what comes out.
struct gsr_tile_history {
    uint32_t magic = 0;
    int device = -1;
    uint32_t* ticks[2] = {nullptr, nullptr};  // device: tile times (10 ns) of the two last frames; ticks[cur] receives the next one's
    uint32_t* order = nullptr;                // device: the blend's workgroup order
    uint32_t* deep = nullptr;                 // device word: how many leading entries of the order are DEEP tiles (blend.hip)
    uint32_t* stats = nullptr;                // pinned host words, written by tile_order_kernel: [0] fresh, [1] longest tile, [2] mean,
    uint32_t* stats_dev = nullptr;            //   [3] similarity of the two frames x 1000, [4] order dropped (they do not resemble each other)
    int cur = 0;
    int dims[4] = {0, 0, 0, 0};               // width, height, tile rows [begin, end) the ticks belong to
    uint32_t order_serial = 0;                // the call whose blend took `order` (0: none)
    bool wanted = false;                      // the last statistics say the frame ends on a few slow tiles (or is a light one)
    bool decorrelated = false;                // ... and that the two last frames did not resemble each other
    bool overlapped = false;                  // the last block-plan call ran its blend beside the emission
    bool block_fed = false;                   // ... and read the block lists (else the sorted lists: a tile's time then says less about the block-fed blend)
    uint32_t mean = 0, longest = 0;           // mean and longest tile time of the last statistics; mean 0: none yet for this size
    uint32_t calls = 0;                       // calls since the ticks were last cleared
    uint32_t last_serial = 0;                 // the owning thread's call counter at its last use (the library's own histories: which to give up)
    bool used = false;
    hipStream_t last_stream = nullptr;        // the stream of the call that used it last: what orders two calls' kernels
    hipEvent_t ev_order = nullptr;            // "the order is sorted" (recorded on the library's second stream)
    hipEvent_t ev_switch = nullptr;           // a caller's own history taken to another stream: that stream waits for the old one's tail
};

namespace gsr {
namespace {
constexpr uint32_t kHistoryMagic = 0x54485347u;   // "GSHT"
// Instances per visible Gaussian (R / V) at which the plans and the blend's feed change hands (each with the frames it was
// measured on; `profiles/r05_trained_like.txt` has all of them on a scene of flat, opaque splats on surfaces):
constexpr uint64_t kBlockPlanMinInstances = 6;    // block plan from here on, sort plan below
constexpr uint64_t kBlockFeedMinInstances = 48;   // a SERIAL blend reads the block lists from here on, the sorted lists below
constexpr uint32_t kBigSplatTiles = 256;          // "a big splat" (16 x 16 tiles and more) for the plan's choice
constexpr uint64_t kOverlapMinInstances = 16;     // the blend may run beside the emission (block-fed) from here on — when the tile times say it is the shorter of the two
// A blend fed from the sorted lists gives EVERY tile four waves (deep tiles, blend.hip) below this many instances per visible
// Gaussian: short lists of small splats, where a frame ends on the lone waves of its few deep tiles and four waves per tile
// cost the others next to nothing. Blend, one wave per tile -> four, bench scene (`profiles/r06_deep_tiles.txt`): R/V = 3.6: 0.66 ->
// 0.30 ms, 5.3: 0.54 -> 0.31, 7.3: 0.54 -> 0.43, 11: 0.60 -> 0.59, 16: 0.63 -> 0.66, 23: 0.46 -> 0.56 (long lists that few
// records of survive: the walk is the work, and the four waves meet at a barrier every 256 entries of it); on 49 unrelated
// views of the same scene (bench.py's random views, serial blends): 0.57-0.97 of one wave's time up to 12, 0.83-1.00 at 12-16,
// 0.94-1.06 at 16-24, up to 1.23 beyond. The switch is at 16 — where the blend may start to run beside the emission
// (kOverlapMinInstances), fed from the block lists, which have no deep tiles. Above it: one
// wave per tile (four for the history's slowest tiles only was built and measured neutral there: GSR_DEEP_BY_HISTORY).
constexpr uint64_t kDeepAllMaxInstances = 16;
constexpr unsigned long long kColorsTicksPerMega = 2040;   // colors_visible_kernel alone: 10 ns units per million Gaussians (50 M: 1.02 ms, 64 bytes fetched per Gaussian at the memory's request rate)
constexpr size_t kMaxDefaultHistories = 8;        // streams per host thread and device that get a history of the library's own

int tile_history_new(gsr_tile_history** out) {
    gsr_tile_history* h = new gsr_tile_history;
    auto fail_with = [&](hipError_t e, const char* what) {
        set_hip_error(e, what);
        if (h->ticks[0]) (void)hipFree(h->ticks[0]);
        if (h->stats) (void)hipHostFree(h->stats);
        if (h->ev_order) (void)hipEventDestroy(h->ev_order);
        if (h->ev_switch) (void)hipEventDestroy(h->ev_switch);
        delete h;
        return GSR_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipGetDevice(&h->device)) != hipSuccess) return fail_with(e, "hipGetDevice");
    uint32_t* dev = nullptr;
    if ((e = hipMalloc(reinterpret_cast<void**>(&dev), sizeof(uint32_t) * (3 * kTileOrderMax + 32))) != hipSuccess) return fail_with(e, "hipMalloc (tile history)");
    h->ticks[0] = dev; h->ticks[1] = dev + kTileOrderMax; h->order = dev + 2 * kTileOrderMax; h->deep = dev + 3 * kTileOrderMax;
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&h->stats), 64, hipHostMallocMapped)) != hipSuccess) return fail_with(e, "hipHostMalloc (tile history)");
    memset(h->stats, 0, 64);
    if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&h->stats_dev), h->stats, 0)) != hipSuccess) return fail_with(e, "hipHostGetDevicePointer");
    if ((e = hipEventCreateWithFlags(&h->ev_order, hipEventDisableTiming)) != hipSuccess) return fail_with(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&h->ev_switch, hipEventDisableTiming)) != hipSuccess) return fail_with(e, "hipEventCreate");
    h->magic = kHistoryMagic;
    *out = h;
    return GSR_OK;
}

constexpr uint32_t kAsyncSlots = 64;       // error-word slots handed out in turn, one per gsr_forward call
constexpr uint32_t kAsyncBase = 16;        // first slot word inside the pinned block
struct Readback {
    uint32_t* host_dev = nullptr;      // the same words as the device sees them (pinned host memory is mapped)
    uint32_t* host = nullptr;          // [3] top digits, [4] V, [6..7] u64 un-wrapped instance count, [10] side way taken, [11] side keys below the main top
                                       // byte, [12] side keys as the scan counted them, [13] as the compaction listed them (all written by the
                                       // kernels that compute them), [8..9] staged count;
                                       // from [kAsyncBase]: kAsyncSlots x {N-sized sort gave up, R-sized sort gave up (both
                                       // written by the kernels themselves), serial of the owning call, 0}
    uint32_t serial = 0;               // calls made so far by this thread on this device
    unsigned long long* staged_dev = nullptr;
    unsigned long long* staged_host = nullptr;
    hipEvent_t ev[2 * GSR_NUM_STAGES] = {};   // [2s] start, [2s+1] end of stage s
    bool events = false;
    bool recorded[GSR_NUM_STAGES] = {};
    int begin_of[GSR_NUM_STAGES] = {};        // event index a stage starts at (default 2s)
    void ev_alias_begin(int stage, int after_stage) { begin_of[stage] = 2 * after_stage + 1; }
    hipEvent_t ev_r = nullptr;                // "numRendered has landed in host memory"
    hipStream_t side = nullptr;               // block plan, GSR_FLAG_OVERLAP_EMIT: the blend runs here, beside the emission
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int ensure_side() {
        if (!side) {
            // (a priority of its own: HIP maps the streams of a priority onto a few hardware queues, and in a process with
            // many streams — torch.distributed and RCCL bring theirs — this one landed on the caller's queue: its kernels then
            // ran in front of the caller's instead of beside them, forced-distributed bench 1.37 -> 1.45 ms. The lower
            // priority also suits what it carries: work that is to fill gaps, never to be waited for)
            // (measured, forced-distributed / plain bench: normal 1.441 / 1.211, lowest 1.214 / 1.208, highest 1.237 / 1.243 ms)
            int prio_low = 0, prio_high = 0;
            if (hipDeviceGetStreamPriorityRange(&prio_low, &prio_high) != hipSuccess) { (void)hipGetLastError(); prio_low = 0; }
            GSR_HIP_TRY(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, prio_low));
            GSR_HIP_TRY(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
            GSR_HIP_TRY(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        }
        return GSR_OK;
    }
    int ensure() {
        if (!host) {
            const size_t bytes = sizeof(uint32_t) * (kAsyncBase + 4 * kAsyncSlots);
            GSR_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&host), bytes, hipHostMallocMapped));
            memset(host, 0, bytes);
            GSR_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&host_dev), host, 0));
            staged_host = reinterpret_cast<unsigned long long*>(host + 8);
        }
        if (!ev_r) GSR_HIP_TRY(hipEventCreateWithFlags(&ev_r, hipEventDisableTiming));
        return GSR_OK;
    }
    // the tile histories this thread's calls without one of their own take: one per stream (see TileHistory below)
    std::vector<gsr_tile_history*> default_histories;
    hipEvent_t ev_colors = nullptr;           // "geomState.rgb is written" (colors_visible_kernel on the side stream)
    hipEvent_t ev_pre_blend = nullptr;        // "the blend is about to start" (colours beside the blend)
    int ensure_colors() {
        { const int rc = ensure_side(); if (rc != GSR_OK) return rc; }
        if (!ev_colors) {
            GSR_HIP_TRY(hipEventCreateWithFlags(&ev_colors, hipEventDisableTiming));
            GSR_HIP_TRY(hipEventCreateWithFlags(&ev_pre_blend, hipEventDisableTiming));
        }
        return GSR_OK;
    }
    int ensure_staged() {
        if (!staged_dev) GSR_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&staged_dev), sizeof(unsigned long long)));
        return GSR_OK;
    }
    int ensure_events() {
        if (!events) {
            for (auto& e : ev) GSR_HIP_TRY(hipEventCreate(&e));
            events = true;
        }
        return GSR_OK;
    }
};
// Everything the calling thread owns, per device; given back by gsr_thread_release and when the thread ends.
struct ThreadResources {
    std::map<int, Readback> by_device;
    void release();
    ~ThreadResources() { release(); }
};
static thread_local ThreadResources g_thread;

// What the environment asks for, read once per process (A/B runs and the tests set these before the first call).
struct EnvKnobs {
    bool tile_history;         // GSR_TILE_HISTORY=0: no call reads or writes a tile history
    int colors_beside;         // GSR_COLORS_BESIDE=0|1|2: geomState.rgb inside the preprocess / beside the depth sort / beside the blend; -1: by size
    int fused_depth;           // GSR_FUSED_DEPTH=0|1: the depth order with / without the compaction whatever the size; -1: by size
    int colors_early_pct;      // GSR_COLORS_EARLY_PCT=0..100: of the colours written beside the blend, the share that starts right behind the preprocess; -1: the default
    int depth_records;         // GSR_DEPTH_RECORDS=0|1: the depth order's triples as three arrays / as 12-byte records between its passes; -1: records without the compaction
    long block_feed_min;       // GSR_BLOCK_FEED_MIN=n: kBlockFeedMinInstances for this process (A/B runs); -1: the constant
    long deep_all_max;         // GSR_DEEP_ALL_MAX=n: kDeepAllMaxInstances for this process (A/B runs); -1: the constant
    bool deep_waves_auto;      // GSR_DEEP_WAVES_AUTO=0: deep tiles always get four waves (A/B runs)
    bool deep_by_history;      // GSR_DEEP_BY_HISTORY=1: above that, the history's slowest tiles get four waves (tile_order_kernel's
                               // count; measured neutral, `profiles/r06_deep_tiles.txt`: off by default)
};
EnvKnobs read_env_knobs() {
    EnvKnobs e;
    const char* h = getenv("GSR_TILE_HISTORY");
    e.tile_history = !(h && h[0] == '0');
    const char* c = getenv("GSR_COLORS_BESIDE");
    e.colors_beside = c && c[0] >= '0' && c[0] <= '2' ? c[0] - '0' : -1;
    const char* f = getenv("GSR_FUSED_DEPTH");
    e.fused_depth = f && (f[0] == '0' || f[0] == '1') ? f[0] - '0' : -1;
    const char* ce = getenv("GSR_COLORS_EARLY_PCT");
    e.colors_early_pct = ce && ce[0] >= '0' && ce[0] <= '9' ? std::min(100, atoi(ce)) : -1;
    const char* dr = getenv("GSR_DEPTH_RECORDS");
    e.depth_records = dr && (dr[0] == '0' || dr[0] == '1') ? dr[0] - '0' : -1;
    const char* b = getenv("GSR_BLOCK_FEED_MIN");
    e.block_feed_min = b && b[0] ? atol(b) : -1;
    const char* d = getenv("GSR_DEEP_ALL_MAX");
    e.deep_all_max = d && d[0] ? atol(d) : -1;
    const char* dw = getenv("GSR_DEEP_WAVES_AUTO");
    e.deep_waves_auto = !(dw && dw[0] == '0');
    const char* dh = getenv("GSR_DEEP_BY_HISTORY");
    e.deep_by_history = dh && dh[0] == '1';
    return e;
}
// (read when the first call needs them; gsr_reread_environment — the tests' and A/B scripts' way of changing a knob inside one
// process — reads them again: no call may be in flight on another thread meanwhile)
EnvKnobs& env_knobs_storage() {
    static EnvKnobs k = read_env_knobs();
    return k;
}
const EnvKnobs& env_knobs() { return env_knobs_storage(); }

// The calling thread's resources for the CURRENT device.
int current_readback(Readback*& out) {
    int dev = 0;
    GSR_HIP_TRY(hipGetDevice(&dev));
    out = &g_thread.by_device[dev];
    return GSR_OK;
}

void destroy_history(gsr_tile_history* h) {
    h->magic = 0;
    (void)hipFree(h->ticks[0]);
    (void)hipHostFree(h->stats);
    (void)hipEventDestroy(h->ev_order);
    (void)hipEventDestroy(h->ev_switch);
    delete h;
}

// Gives back what the calling thread's calls have made the library allocate, for every device: the second stream (drained
// first: nothing of the library's is in flight afterwards — work on the CALLER's streams is the caller's to wait for before
// it frees the chunks), the pinned words, every event, the staged-record counter and the histories the library kept for
// calls without one of their own. A later call of the thread starts from nothing again.
void ThreadResources::release() {
    if (by_device.empty()) return;
    int before = -1;
    const bool have_device = hipGetDevice(&before) == hipSuccess;
    for (auto& kv : by_device) {
        Readback& rb = kv.second;
        if (hipSetDevice(kv.first) != hipSuccess) { (void)hipGetLastError(); continue; }     // (the runtime is gone: nothing left to free)
        if (rb.side) { (void)hipStreamSynchronize(rb.side); (void)hipStreamDestroy(rb.side); }
        for (gsr_tile_history* h : rb.default_histories) destroy_history(h);
        if (rb.events) for (auto& e : rb.ev) (void)hipEventDestroy(e);
        for (hipEvent_t e : {rb.ev_r, rb.ev_fork, rb.ev_join, rb.ev_colors, rb.ev_pre_blend})
            if (e) (void)hipEventDestroy(e);
        if (rb.staged_dev) (void)hipFree(rb.staged_dev);
        if (rb.host) (void)hipHostFree(rb.host);
        (void)hipGetLastError();
    }
    by_device.clear();
    if (have_device) (void)hipSetDevice(before);
    (void)hipGetLastError();
}

}  // namespace

const uint32_t* tile_order_of_call(const gsr_forward_receipt& r, int row_begin, int row_end) {
    if (r.serial == 0u) return nullptr;
    const int dims[4] = {r.width, r.height, row_begin, row_end};
    auto fits = [&](const gsr_tile_history* h) {
        return h && h->magic == kHistoryMagic && h->order_serial == r.serial && memcmp(dims, h->dims, sizeof(dims)) == 0;
    };
    if (r.tile_history) return fits(r.tile_history) ? r.tile_history->order : nullptr;    // (the caller's own: its to share between threads)
    Readback* rb = nullptr;
    if (current_readback(rb) != GSR_OK) return nullptr;
    for (const gsr_tile_history* h : rb->default_histories)
        if (fits(h)) return h->order;
    return nullptr;
}

DeviceShape device_shape_of(int cus) {
    DeviceShape s;
    s.cus = cus > 0 ? cus : 1;
    s.blend_slots = (uint32_t)s.cus * 4u * 5u;
    s.blend_slots_beside = (uint32_t)s.cus * 4u * 3u;
    s.light_frame_ticks = 25000ull * (unsigned long long)s.blend_slots;
    return s;
}

int current_device_shape(DeviceShape* out) {
    static thread_local std::map<int, DeviceShape> cache;
    int dev = 0;
    GSR_HIP_TRY(hipGetDevice(&dev));
    auto it = cache.find(dev);
    if (it == cache.end()) {
        int cus = 0;
        GSR_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        it = cache.emplace(dev, device_shape_of(cus)).first;
    }
    *out = it->second;
    return GSR_OK;
}

// What a gsr_backward call may read of the forward call that issued `r` (see gsr_backward_args.receipt): derived from
// the receipt and the chunk layouts alone — no state of this library is consulted, so any host thread may ask, after any
// number of other calls, as long as the chunks are as that call left them.
int lists_of_receipt(const gsr_forward_receipt& r, int n, int width, int height, int row_begin, int row_end,
                     const void* point_list, BlockFeed* feed, bool* from_blocks, bool* lists_written) {
    *from_blocks = false;
    *lists_written = true;
    if (r.magic != GSR_RECEIPT_MAGIC) return GSR_ERR_INVALID_ARG;
    if (r.num_gaussians != n || r.width != width || r.height != height || r.tile_row_begin != row_begin ||
        r.tile_row_end != row_end || !r.geometry_chunk || !r.image_chunk)
        return GSR_ERR_INVALID_ARG;
    if (r.num_rendered == 0) return GSR_OK;                     // (no binning chunk, no lists: the caller zeroes its outputs)
    if (!r.binning_chunk) return GSR_ERR_INVALID_ARG;
    gsr_binning_state bin;
    gsr_binning_from_chunk(r.binning_chunk, r.num_rendered, &bin);
    if (point_list != bin.values) return GSR_ERR_INVALID_ARG;
    *lists_written = !(r.plan_used & GSR_PLAN_LISTS_SKIPPED);
    // (not after a blend from the sorted lists: BlockMeta::walked, which bounds the per-entry sums, is the block-fed
    // blend's by-product; the backward then takes the sorted lists for every tile)
    if ((r.plan_used & 0xFFu) == GSR_PLAN_BLOCKS && !(r.plan_used & GSR_PLAN_BLEND_FROM_LISTS)) {
        gsr_geometry_state geom;
        gsr_geometry_from_chunk(r.geometry_chunk, n, &geom);
        const GeoScratch gs = carve_geo_scratch(geom.scanning_space, (size_t)n);
        const int grid_x = (width + kTile - 1) / kTile, grid_y = (height + kTile - 1) / kTile;
        // the block lists (read by nothing else once the forward call is complete) and, for the per-entry gradient sums,
        // the 8 R bytes of keysUnsorted: the (rectangle | depth) halves of the block-list entries there are dead after
        // the unit masks and the emission
        *feed = block_feed((int)r.num_visible, grid_x, grid_y, r.num_rendered, gs.block_scratch, bin.values_unsorted, bin.sorting_space);
        feed->acc = reinterpret_cast<float*>(bin.keys_unsorted);
        feed->acc_floats = 2ull * (unsigned long long)r.num_rendered;
        *from_blocks = true;
    }
    if (!*from_blocks && !*lists_written) return GSR_ERR_INVALID_ARG;
    return GSR_OK;
}

}  // namespace gsr

using namespace gsr;

extern "C" {

char* gsr_geometry_from_chunk(char* chunk, int n, gsr_geometry_state* s) {
    const size_t N = (size_t)(n < 0 ? 0 : n);
    obtain(chunk, s->tiles_touched, sizeof(uint32_t) * N);
    s->scan_size = carve_geo_scratch(nullptr, N).bytes;
    s->num_rendered = 0;
    obtain(chunk, s->scanning_space, s->scan_size);
    obtain(chunk, s->depths, sizeof(float) * N);
    obtain(chunk, s->clamped, sizeof(uint8_t) * N * 3);
    obtain(chunk, s->internal_radii, sizeof(int32_t) * N);
    obtain(chunk, s->means2D, sizeof(float) * 2 * N);
    obtain(chunk, s->cov3D, sizeof(float) * 6 * N);
    obtain(chunk, s->conic_opacity, sizeof(float) * 4 * N);
    obtain(chunk, s->rgb, sizeof(float) * 3 * N);
    obtain(chunk, s->point_offsets, sizeof(uint32_t) * N);
    return chunk;
}

char* gsr_image_from_chunk(char* chunk, int size, gsr_image_state* s) {
    const size_t P = (size_t)(size < 0 ? 0 : size);
    obtain(chunk, s->ranges, sizeof(uint32_t) * 2 * P);
    obtain(chunk, s->n_contrib, sizeof(uint32_t) * P);
    obtain(chunk, s->accum_alpha, sizeof(float) * P);
    return chunk;
}

char* gsr_binning_from_chunk(char* chunk, size_t size, gsr_binning_state* s) {
    obtain(chunk, s->keys_unsorted, sizeof(uint64_t) * size);
    obtain(chunk, s->keys, sizeof(uint64_t) * size);
    obtain(chunk, s->values_unsorted, sizeof(uint32_t) * size);
    obtain(chunk, s->values, sizeof(uint32_t) * size);
    s->sorting_size = std::max(std::max(carve_bin_scratch(nullptr, size).bytes, sort_temp_bytes(size)), blockbin_bin_bytes(size));
    obtain(chunk, s->sorting_space, s->sorting_size);
    return chunk;
}

size_t gsr_required_geometry(int n) { gsr_geometry_state s; return reinterpret_cast<size_t>(gsr_geometry_from_chunk(nullptr, n, &s)); }
size_t gsr_required_image(int size) { gsr_image_state s; return reinterpret_cast<size_t>(gsr_image_from_chunk(nullptr, size, &s)); }
size_t gsr_required_binning(size_t size) { gsr_binning_state s; return reinterpret_cast<size_t>(gsr_binning_from_chunk(nullptr, size, &s)); }

uint32_t gsr_higher_msb(uint32_t n) {   // GSCuda.cu:481-502
    int msb = (int)sizeof(uint32_t) * 4;
    int step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return (uint32_t)msb;
}

int gsr_last_error(void) { return g_last_error; }
const char* gsr_last_hip_error(void) { return g_hip_error; }
const char* gsr_error_string(int code) {
    switch (code) {
        case GSR_OK: return "ok";
        case GSR_ERR_INVALID_ARG: return "invalid argument";
        case GSR_ERR_ALLOC: return "chunk allocator returned NULL";
        case GSR_ERR_HIP: return "HIP runtime error";
        case GSR_ERR_NO_DEVICE: return "no HIP device";
        case GSR_ERR_TOO_LARGE: return "numRendered exceeds 32-bit offsets";
        case GSR_ERR_INTERNAL: return "radix sort look-back gave up (bounded spin expired)";
        case GSR_ERR_STALE_RECEIPT: return "the receipt's error slot has been handed to a later call";
        default: return "unknown error";
    }
}

size_t gsr_scan_temp_bytes(size_t n) { return scan_temp_bytes(n); }
int gsr_inclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, char* temp, void* stream) {
    g_hip_error[0] = 0;
    if (n && (!in || !out || !temp)) return fail(GSR_ERR_INVALID_ARG);
    return fail(launch_inclusive_scan(in, out, n, temp, (hipStream_t)stream));
}
size_t gsr_sort_temp_bytes(size_t n) { return sort_temp_bytes(n); }
int gsr_sort_pairs_u64_u32(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* values_in,
                           uint32_t* values_out, size_t n, int begin_bit, int end_bit, char* temp, void* stream) {
    g_hip_error[0] = 0;
    if (n && (!keys_in || !keys_out || !values_in || !values_out || !temp)) return fail(GSR_ERR_INVALID_ARG);
    return fail(launch_sort_pairs(keys_in, keys_out, values_in, values_out, n, begin_bit, end_bit, temp, (hipStream_t)stream));
}

int gsr_colors_from_dc(int n, const float* shs, float* colors, void* stream) {
    g_hip_error[0] = 0;
    if (n <= 0) return fail(GSR_OK);
    if (!shs || !colors) return fail(GSR_ERR_INVALID_ARG);
    return fail(launch_colors_from_dc(n, shs, colors, (hipStream_t)stream));
}

int gsr_blend_expf(int n, const float* in, float* out, void* stream) {
    g_hip_error[0] = 0;
    if (n > 0 && (!in || !out)) return fail(GSR_ERR_INVALID_ARG);
    return fail(launch_exp_test(n, in, out, (hipStream_t)stream));
}

int gsr_footprint_misses_tile(int n, const float* means2D, const float* conic_opacity, const int32_t* tile_xy, int width,
                              int height, uint8_t* misses, void* stream) {
    g_hip_error[0] = 0;
    if (n > 0 && (!means2D || !conic_opacity || !tile_xy || !misses || width <= 0 || height <= 0)) return fail(GSR_ERR_INVALID_ARG);
    return fail(launch_footprint_test(n, means2D, conic_opacity, tile_xy, width, height, misses, (hipStream_t)stream));
}

int gsr_tile_history_create(gsr_tile_history** out) {
    g_hip_error[0] = 0;
    if (!out) return fail(GSR_ERR_INVALID_ARG);
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(GSR_ERR_NO_DEVICE);
    return fail(tile_history_new(out));
}

int gsr_tile_history_destroy(gsr_tile_history* h) {
    g_hip_error[0] = 0;
    if (!h) return fail(GSR_OK);
    if (h->magic != kHistoryMagic) return fail(GSR_ERR_INVALID_ARG);
    destroy_history(h);
    return fail(GSR_OK);
}

void gsr_reread_environment(void) { env_knobs_storage() = read_env_knobs(); }

int gsr_thread_release(void) {
    g_hip_error[0] = 0;
    g_thread.release();
    return fail(GSR_OK);
}

void gsr_device_shape(int cus, uint32_t out[4]) {
    const DeviceShape s = device_shape_of(cus);
    out[0] = (uint32_t)s.cus; out[1] = s.blend_slots; out[2] = s.blend_slots_beside; out[3] = (uint32_t)(s.light_frame_ticks / 25000ull);
}

int gsr_tile_history_stats(const gsr_tile_history* h, uint32_t out[6]) {
    if (!h || h->magic != kHistoryMagic || !out) return fail(GSR_ERR_INVALID_ARG);
    out[0] = h->stats[0] != 0u ? h->stats[2] : h->mean;          // (words the last sort has left and no call has read yet come first)
    out[1] = h->stats[1];
    out[2] = h->stats[3];
    out[3] = (h->stats[0] != 0u ? h->stats[4] != 0u : h->decorrelated) ? 1u : 0u;
    out[4] = h->calls;
    out[5] = h->overlapped ? 1u : 0u;
    return fail(GSR_OK);
}

int gsr_tile_history_forget_stream(gsr_tile_history* h) {
    g_hip_error[0] = 0;
    if (!h || h->magic != kHistoryMagic) return fail(GSR_ERR_INVALID_ARG);
    h->used = false;
    h->last_stream = nullptr;
    return fail(GSR_OK);
}

int gsr_tile_history_times(const gsr_tile_history* h, uint32_t* times, int count, uint32_t* deep_tiles) {
    g_hip_error[0] = 0;
    if (!h || h->magic != kHistoryMagic || !times || count < 0 || count > kTileOrderMax) return fail(GSR_ERR_INVALID_ARG);
    // (a tool's call: it synchronises the device — the history's calls may be on any stream)
    GSR_HIP_TRY(hipDeviceSynchronize());
    GSR_HIP_TRY(hipMemcpy(times, h->ticks[h->cur ^ 1], sizeof(uint32_t) * (size_t)count, hipMemcpyDeviceToHost));
    if (deep_tiles) GSR_HIP_TRY(hipMemcpy(deep_tiles, h->deep, sizeof(uint32_t), hipMemcpyDeviceToHost));
    return fail(GSR_OK);
}

int gsr_poll_async_error(const gsr_forward_receipt* r) {
    if (!r || r->magic != GSR_RECEIPT_MAGIC || !r->async_words) return fail(GSR_ERR_INVALID_ARG);
    const volatile uint32_t* w = r->async_words;
    // (the kernels write the owning call's serial, not 1: a late writer of an older call cannot raise the new owner's flag)
    if (w[0] == r->serial || w[1] == r->serial) return fail(GSR_ERR_INTERNAL);
    if (w[2] != r->serial) return fail(GSR_ERR_STALE_RECEIPT);   // the slot has a new owner: nothing is known about that call any more
    return GSR_OK;
}

int gsr_forward(gsr_forward_args* a) {
    g_hip_error[0] = 0;
    if (!a || a->struct_size != sizeof(gsr_forward_args)) return fail(GSR_ERR_INVALID_ARG);
    a->num_rendered = 0;
    a->records_staged = 0;
    a->plan_used = 0;
    memset(a->stage_ms, 0, sizeof(a->stage_ms));
    memset(&a->receipt, 0, sizeof(a->receipt));
    const int n = a->num_gaussians;
    if (n <= 0 || a->width <= 0 || a->height <= 0 || !a->geometry_alloc || !a->binning_alloc || !a->image_alloc ||
        !a->background || !a->means3D || !a->opacities || !a->view_matrix || !a->proj_matrix || !a->out_color ||
        (!a->shs && !a->colors_precomp) || (!a->cov3D_precomp && (!a->scales || !a->rotations)))
        return fail(GSR_ERR_INVALID_ARG);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(GSR_ERR_NO_DEVICE);

    hipStream_t stream = (hipStream_t)a->stream;
    const bool profile = (a->flags & GSR_FLAG_PROFILE) != 0;
    const bool count_staged = (a->flags & GSR_FLAG_COUNT_STAGED) != 0;
    const bool inria = (a->flags & GSR_FLAG_SEMANTICS_INRIA) != 0;
    if (inria && !a->cam_pos && !a->colors_precomp) return fail(GSR_ERR_INVALID_ARG);
    const int32_t* rects_in = inria ? nullptr : a->rects;      // upstream rectangles are radius-based
    int rc;
    Readback* rbp = nullptr;
    if ((rc = current_readback(rbp)) != GSR_OK) return fail(rc);
    Readback& g_rb = *rbp;
    if ((rc = g_rb.ensure()) != GSR_OK) return fail(rc);
    if (profile && (rc = g_rb.ensure_events()) != GSR_OK) return fail(rc);
    if (count_staged && (rc = g_rb.ensure_staged()) != GSR_OK) return fail(rc);

    FrameDims d;
    d.width = a->width;
    d.height = a->height;
    d.grid_x = (a->width + kTile - 1) / kTile;
    d.grid_y = (a->height + kTile - 1) / kTile;
    d.row_begin = 0;
    d.row_end = d.grid_y;
    if (a->tile_row_begin != 0 || a->tile_row_end != 0) {
        if (a->tile_row_begin < 0 || a->tile_row_end > d.grid_y || a->tile_row_begin > a->tile_row_end)
            return fail(GSR_ERR_INVALID_ARG);
        d.row_begin = a->tile_row_begin;
        d.row_end = a->tile_row_end;
    }
    const int num_tiles = d.grid_x * d.grid_y;

    // GSCuda.cu:723-729
    char* geo_chunk = a->geometry_alloc(a->geometry_user, gsr_required_geometry(n));
    if (!geo_chunk) return fail(GSR_ERR_ALLOC);
    gsr_geometry_state geom;
    gsr_geometry_from_chunk(geo_chunk, n, &geom);
    int32_t* radii = a->radii ? a->radii : geom.internal_radii;

    // GSCuda.cu:734-736
    const int P = a->width * a->height;
    char* img_chunk = a->image_alloc(a->image_user, gsr_required_image(P) + 128);
    if (!img_chunk) return fail(GSR_ERR_ALLOC);
    gsr_image_state img;
    gsr_image_from_chunk(img_chunk, P, &img);

#define GSR_BEGIN(s) do { if (profile) GSR_HIP_TRY(hipEventRecord(g_rb.ev[2 * (s)], stream)); } while (0)
#define GSR_END(s) do { if (profile) { GSR_HIP_TRY(hipEventRecord(g_rb.ev[2 * (s) + 1], stream)); g_rb.recorded[s] = true; } } while (0)
#define GSR_STEP(call) do { rc = (call); if (rc != GSR_OK) return fail(rc); } while (0)

    const GeoScratch gs = carve_geo_scratch(geom.scanning_space, (size_t)n);
    // This call's error words: the next of the slots (the call that owned it 64 calls ago is long complete: the host
    // has waited for every call's read-back since). Zeroed here, written only by a kernel whose bounded look-back spin
    // gave up, read by gsr_poll_async_error through the receipt.
    if (++g_rb.serial == 0u) ++g_rb.serial;                    // (0 is what an untouched error word holds)
    const uint32_t serial = g_rb.serial;
    const uint32_t slot_at = kAsyncBase + 4u * (serial % kAsyncSlots);
    g_rb.host[slot_at] = g_rb.host[slot_at + 1] = g_rb.host[slot_at + 3] = 0;
    g_rb.host[slot_at + 2] = serial;
    uint32_t* const err_n = g_rb.host_dev + slot_at;          // N-sized depth sort
    uint32_t* const err_r = g_rb.host_dev + slot_at + 1;      // R-sized sort (sort plan, generic plan)
    // what the caller takes away from this call (complete once GSR_OK is certain)
    auto issue_receipt = [&](uint32_t nv, char* bin_chunk_or_null) {
        gsr_forward_receipt& rc_ = a->receipt;
        rc_.plan_used = a->plan_used;
        rc_.num_gaussians = n; rc_.width = a->width; rc_.height = a->height;
        rc_.tile_row_begin = d.row_begin; rc_.tile_row_end = d.row_end;
        rc_.num_rendered = a->num_rendered; rc_.num_visible = nv;
        rc_.serial = serial;
        rc_.geometry_chunk = geo_chunk; rc_.image_chunk = img_chunk; rc_.binning_chunk = bin_chunk_or_null;
        rc_.async_words = g_rb.host + slot_at;
        rc_.tile_history = a->tile_history;
        rc_.magic = GSR_RECEIPT_MAGIC;
    };
    for (bool& r : g_rb.recorded) r = false;
    for (int s = 0; s < GSR_NUM_STAGES; ++s) g_rb.begin_of[s] = 2 * s;

    // Slow tiles first (TileOrder, blend_core.hpp): the order of this call's blend workgroups is sorted from the tile times of
    // the history's last frame while the depth sort runs, on the library's second stream. Which history: the caller's, or
    // this thread's own for the call's stream (a call of another size starts from zeros = patch order).
    DeviceShape shape;
    GSR_STEP(current_device_shape(&shape));
    const EnvKnobs& env = env_knobs();
    gsr_tile_history* hist = nullptr;
    if (env.tile_history && !(a->flags & GSR_FLAG_NO_TILE_HISTORY) && tile_order_workgroups(d) <= kTileOrderMax &&
        d.grid_x * d.grid_y <= kTileOrderMax) {
        if (a->tile_history) {
            int dev_now = -1;
            GSR_HIP_TRY(hipGetDevice(&dev_now));
            if (a->tile_history->magic != kHistoryMagic || a->tile_history->device != dev_now) return fail(GSR_ERR_INVALID_ARG);
            hist = a->tile_history;
            if (hist->used && hist->last_stream != stream) {
                // (the caller has taken its history to another stream: this call's kernels go behind what the old stream holds
                // now — if that stream is gone, so is its work)
                if (hipEventRecord(hist->ev_switch, hist->last_stream) == hipSuccess) GSR_HIP_TRY(hipStreamWaitEvent(stream, hist->ev_switch, 0));
                else (void)hipGetLastError();
            }
        } else {
            for (gsr_tile_history* h : g_rb.default_histories)
                if (h->last_stream == stream) { hist = h; break; }
            if (!hist && g_rb.default_histories.size() >= kMaxDefaultHistories) {
                // A ninth stream: the history this thread has not used for the longest time goes — its stream may be gone
                // (a caller that makes a stream per frame), so nothing is asked of that stream: freeing the history's memory
                // waits for the device to be done with it (hipFree). Rare by construction; before round 6 the calls on
                // further streams simply ran without a history, for good.
                size_t lru = 0;
                for (size_t i = 1; i < g_rb.default_histories.size(); ++i)
                    if ((int32_t)(g_rb.default_histories[i]->last_serial - g_rb.default_histories[lru]->last_serial) < 0) lru = i;
                destroy_history(g_rb.default_histories[lru]);
                g_rb.default_histories.erase(g_rb.default_histories.begin() + (long)lru);
            }
            if (!hist) {
                // (a history is an accelerator: if the device has no memory left for one, the call runs without)
                if (tile_history_new(&hist) == GSR_OK) g_rb.default_histories.push_back(hist);
                else { hist = nullptr; (void)hipGetLastError(); g_hip_error[0] = 0; }
            }
        }
        if (hist) { hist->last_stream = stream; hist->used = true; hist->last_serial = g_rb.serial; GSR_STEP(g_rb.ensure_side()); }       // (the stream the sort of the order runs on)
    }
    const bool history = hist != nullptr;
    // The order costs a launch on the second stream and the host a few microseconds, and it pays on frames that END on a
    // few slow tiles and on light frames: it is sorted when the last statistics (the history's pinned host words, left by
    // tile_order_kernel: fresh, the longest tile and the mean) say the longest tile takes 2.5 times what the tiles would take spread evenly over the chip's
    // 5 120 wave slots — and every fourth call, to have fresh statistics (a camera that leaves the cloud is noticed within
    // five frames). The ticks are recorded by every call.
    bool order_now = false;
    uint32_t* t_ticks = nullptr;
    if (history) {
        const int dims_now[4] = {a->width, a->height, d.row_begin, d.row_end};
        if (memcmp(dims_now, hist->dims, sizeof(dims_now)) != 0) {
            GSR_HIP_TRY(hipMemsetAsync(hist->ticks[0], 0, sizeof(uint32_t) * 2 * (size_t)kTileOrderMax, stream));
            memcpy(hist->dims, dims_now, sizeof(dims_now));
            hist->wanted = hist->decorrelated = hist->overlapped = false;
            hist->calls = 0; hist->stats[0] = 0; hist->mean = hist->longest = 0; hist->order_serial = 0;
        }
        if (hist->stats[0] != 0u) {
            const unsigned long long tiles = (unsigned long long)(d.row_end - d.row_begin) * (unsigned long long)d.grid_x;
            hist->wanted = 2ull * (unsigned long long)shape.blend_slots * hist->stats[1] > 5ull * tiles * hist->stats[2] ||
                           (hist->stats[2] != 0u && tiles * hist->stats[2] < shape.light_frame_ticks);      // (or a light frame: tile_order_kernel)
            hist->mean = hist->stats[2];
            hist->longest = hist->stats[1];
            hist->decorrelated = hist->stats[4] != 0u;
            hist->stats[0] = 0;
        }
        // (while the frames do not resemble each other the sort runs every call: it is what looks whether they do again — a
        // camera cut is over after three frames — and it hands out the patch order as long as they do not)
        order_now = hist->wanted || hist->decorrelated || (hist->calls % 4u) == 1u;       // (call 0 has no ticks yet)
        ++hist->calls;
        t_ticks = hist->ticks[hist->cur];                 // (this call's times; the order is sorted from the other set)
        hist->cur ^= 1;
    }
    // geomState.rgb (GSCuda.cu:362-366) is a strided read nothing needs before the blend: by default it is written by a kernel
    // of its own on the second stream while the scan and the depth sort run (launch_colors_visible, preprocess.hip). Whatever way the
    // call ends, the caller's stream has waited for it (the chunk is the caller's).
    // Up to 16 M Gaussians beside the depth sort: there its kernels wait on latency and the colours cost them 0.05 ms for the
    // 0.10 ms the preprocess saves — bench frame 1.315 -> 1.268 ms. At 50 M they are bound by HBM themselves and lose what
    // the preprocess gains (6.10 -> 6.19 ms): there the colours are written beside the BLEND — vector-bound —, which takes a
    // record's colour straight from the SH array meanwhile (TileFeed::dc_stride). GSR_COLORS_BESIDE = 0 / 1 / 2 (environment,
    // for A/B runs and the tests): inside the preprocess / beside the depth sort / beside the blend, whatever the size.
    // ... where there IS a blend to hide behind: a frame whose blend runs beside the emission (the history's last frame did) would
    // write its colours behind that blend, beside the rest of the emission — bound by the memory as well, and the frame ends with
    // it; beside the depth sort they cost less (20 M Gaussians of the bench scene, 919 M instances: 3.87 -> 3.76 ms).
    const int colors_forced = env.colors_beside;
    const bool colors_movable = !inria && !a->colors_precomp && !(a->flags & GSR_FLAG_SERIAL_EMIT);
    const bool blend_beside_emission = history && hist->mean != 0u && hist->overlapped;
    const int colors_mode = !colors_movable ? 0 : (colors_forced >= 0 ? colors_forced : ((n <= (1 << 24) || blend_beside_emission) ? 1 : 2));
    const bool colors_beside = colors_mode == 1;
    // (SideJoin: whatever way the call is left — a failing step included — the caller's stream waits for what this call
    // has put on the second stream: for `pending`, an event already recorded there, or, while `tail` is armed, for an
    // event recorded behind everything the second stream holds at that moment)
    struct SideJoin {
        hipStream_t stream;
        hipEvent_t pending;
        Readback* rb;
        bool tail;
        ~SideJoin() {
            if (tail && rb->side && rb->ev_join) {
                if (hipEventRecord(rb->ev_join, rb->side) == hipSuccess) (void)hipStreamWaitEvent(stream, rb->ev_join, 0);
                else (void)hipGetLastError();
            } else if (pending) {
                (void)hipStreamWaitEvent(stream, pending, 0);
            }
        }
    } colors_join{stream, nullptr, &g_rb, false};
    if (colors_mode != 0) GSR_STEP(g_rb.ensure_colors());
    GSR_BEGIN(GSR_STAGE_PREPROCESS);
    const bool xy_plan = d.grid_x <= 255 && d.grid_y <= 255;
    if (inria)
        GSR_STEP(launch_preprocess_inria(*a, geom, radii, gs.depth_key, xy_plan ? gs.rect_idx : nullptr, d, stream, gs.wave_sums, kBigSplatTiles));
    else
        GSR_STEP(launch_preprocess(*a, geom, radii, gs.depth_key, xy_plan ? gs.rect_idx : nullptr, d, stream, gs.wave_sums,
                                   colors_mode != 0, kBigSplatTiles));   // :744-768
    GSR_END(GSR_STAGE_PREPROCESS);
    // Colours beside the blend (mode 2) where the blend is SHORTER than the colours kernel (50 M Gaussians: 1.02 ms of colours
    // alone, 1.37 beside a blend of 0.70 — the frame ended 0.65 ms after its blend): the Gaussians [0, colors_early) get theirs
    // right behind the preprocess — beside the scan and the digit counts, which wait on LDS atomics and latency, and on into the
    // first depth pass, which pays for it (336 -> 545 us with two fifths of them) —, the rest beside the blend, which then
    // outlasts it or nearly: 50 M 4.93-5.07 -> 4.78-4.94 ms. The share: what the history's blend leaves uncovered of
    // kColorsTicksPerMega x N, at most half; none without a history (GSR_COLORS_EARLY_PCT: a fixed share, A/B runs).
    size_t colors_early = 0;
    if (colors_mode == 2) {
        unsigned long long pct = 0;
        if (env.colors_early_pct >= 0) {
            pct = (unsigned long long)env.colors_early_pct;
        } else if (history && hist->mean != 0u && !hist->decorrelated) {
            const unsigned long long tiles = (unsigned long long)(d.row_end - d.row_begin) * (unsigned long long)d.grid_x;
            const unsigned long long blend_ticks = std::max((unsigned long long)hist->mean * tiles / (unsigned long long)shape.blend_slots,
                                                            (unsigned long long)hist->longest);
            const unsigned long long colors_ticks = kColorsTicksPerMega * (unsigned long long)n / 1000000ull;
            if (colors_ticks > blend_ticks) pct = std::min(50ull, 100ull * (colors_ticks - blend_ticks) / colors_ticks);
        }
        colors_early = (size_t)n * (size_t)pct / 100u;
    }
    if (colors_beside || colors_early != 0u) {
        // Forked right behind the preprocess (tilesTouched is final there): the scan's and the compaction's small launches
        // leave most of the chip idle, and what the colours kernel gets done beside them it does not take from the depth passes
        // (forked behind the read-back's event instead — no event of its own on the caller's stream — the three passes took
        // 151 us for their 108: bench frame 1.215 -> 1.205 ms, from outside the cloud 1.52 -> 1.49, (0,0,-30) 1.42 -> 1.39).
        GSR_HIP_TRY(hipEventRecord(g_rb.ev_pre_blend, stream));
        GSR_HIP_TRY(hipStreamWaitEvent(g_rb.side, g_rb.ev_pre_blend, 0));
        colors_join.tail = true;
        GSR_STEP(launch_colors_visible(colors_beside ? n : (int)colors_early, geom.tiles_touched, a->shs, geom.rgb, g_rb.side));
        GSR_HIP_TRY(hipEventRecord(g_rb.ev_colors, g_rb.side));
        colors_join.pending = g_rb.ev_colors;
        colors_join.tail = false;
    }
    // (mode 2: what is still to be written, by the launches further down)
    const int colors_rest = n - (int)colors_early;
    const uint32_t* const rest_tiles = geom.tiles_touched + colors_early;
    const float* const rest_shs = a->shs ? a->shs + 48u * colors_early : nullptr;
    float* const rest_rgb = geom.rgb + 3u * colors_early;
    GSR_BEGIN(GSR_STAGE_SCAN);
    // (the same pass counts the Gaussians with a tile per 4096: the offsets of the depth order's compaction below)
    // Its first launch also clears the depth order's four scratch areas (look-back words, tickets, the digit histograms:
    // adjacent in the chunk), and its one-workgroup launch leaves V and the un-wrapped instance count in the pinned host
    // words themselves: a memset and a copy command of their own were two more 5 us stops on this chain of small launches.
    GSR_STEP(launch_inclusive_scan(geom.tiles_touched, geom.point_offsets, (size_t)n,      // :771
                                   gs.scan_temp, stream, reinterpret_cast<unsigned long long*>(gs.sort_info + 2),
                                   gs.vis_partial, gs.sort_info + 1, g_rb.host_dev + 4,
                                   gs.sweep.ticket, 4 * sweep_scratch_bytes((size_t)n),
                                   gs.wave_sums, gs.main_partial, kDepthSideMax, gs.sort_info + 8,
                                   gs.big_partial, kBigSplatTiles));
    GSR_END(GSR_STAGE_SCAN);
    // The sort of reference :794-797 is an LSD radix sort of (tile | depth) keys. Its low
    // half is the same for every key of a Gaussian, so those digit passes run once per
    // Gaussian BEFORE duplication (N keys, not R): depth order here, tile order below.
    GSR_BEGIN(GSR_STAGE_DEPTH_ORDER);
    SweepScratch four[4] = {gs.sweep, gs.sweep_more[0], gs.sweep_more[1], gs.sweep_more[2]};
    // "a bounded look-back spin gave up" is written by the kernel straight into the pinned host words (it never
    // happens on a healthy device; a copy at the end of every frame for it cost 5 us of stream time)
    for (auto& f : four) { f.error_word = err_n; f.error_value = serial; }
    // Only Gaussians with at least one tile in this call take part from here on (V of N: 52 % on the
    // bench frame, a few per cent per rank when the frame is sharded): their (depth key, index) pairs
    // are compacted in index order — the same kernels count the digits of the four sort passes.
    // With a packed rectangle per Gaussian (grids up to 255 x 255) the rectangle travels with the index through the depth
    // passes: both binning plans want it in depth order, and gathering it by index afterwards is a random 4-byte read per
    // Gaussian (0.93 ms of the 50 M frame).
    // (the depth keys are float bits of NDC z: nearly every visible Gaussian has z in [0.5, 1) and the top byte 0x3F; the
    // handful that does not — 23 of 3 M on the bench frame — goes a side way instead of costing everybody a fourth pass)
    DepthSide side;
    side.words = gs.sort_info + 8;
    side.main_partial = gs.main_partial;
    side.keys = gs.side_k; side.vals = gs.side_v; side.rects = xy_plan ? gs.side_r : nullptr;
    side.capacity = kDepthSideMax;
    // Scenes beyond 16 M Gaussians (there every one of these kernels is bound by HBM): no compaction — its 20 N bytes buy
    // nothing where nearly every Gaussian is visible (50 M: 0.20 ms). The digit counts come from a pass over the keys alone
    // and the first depth pass reads the per-Gaussian arrays itself, leaving out what has no tile (onesweep_kernel, DROP).
    // (GSR_FUSED_DEPTH = 0 / 1 in the environment, for A/B runs and the tests: never / whatever the size)
    const bool fused_depth = xy_plan && (env.fused_depth >= 0 ? env.fused_depth == 1 : n > (1 << 24));
    if (fused_depth)
        GSR_STEP(sort_u32_prepare_counts(gs.depth_key, (uint32_t)n, four, gs.sort_info, stream, g_rb.host_dev + 3, &side));
    else
        GSR_STEP(sort_u32_prepare(gs.depth_key, (uint32_t)n, gs.c_k, gs.c_v, gs.vis_partial, four, gs.sort_info, stream, true,
                                  xy_plan ? gs.rect_idx : nullptr, xy_plan ? gs.c_r : nullptr, g_rb.host_dev + 3, &side));
    // :772 — the pipeline's one device->host read: the binning chunk is sized by R. The host waits for the
    // copies only (an event). They also bring V and whether the fourth depth pass is needed: depth keys are
    // float bits, and when every visible Gaussian has the same top byte (NDC z in [0.5, 1)) that pass would
    // move nothing.
    // (no copy command: the kernels that computed the three figures wrote them into the pinned words as well; numRendered
    // is the low word of the un-wrapped instance count)
    GSR_HIP_TRY(hipEventRecord(g_rb.ev_r, stream));
    // The first three depth passes are needed whatever the read-back says, so they are queued BEFORE the host waits
    // (grids sized for N keys, the true count V read on the device): the device sorts while the host sleeps.
    // Between the passes the (key, index, rectangle) triples travel as 12-byte RECORDS — a digit's run leaves a tile as one
    // piece instead of three, a lane fetches its key's triple with one load (50 M Gaussians: 347 + 2 x 322 -> 340 + 309 + 294 us)
    // — in the room of the compaction's arrays where there is no compaction, else in that of the first pass's destination
    // (dead before the last pass writes its three arrays there), and in the other pair's. Where the passes are bound by
    // latency, not by the memory (up to 16 M Gaussians: with the compaction), it changes nothing (bench frame and the path's poses:
    // +-0.003 ms) and the arrays stay. GSR_DEPTH_RECORDS=0 / 1: never / on both routes.
    const bool depth_records = xy_plan && (env.depth_records >= 0 ? env.depth_records == 1 : fused_depth);
    if (fused_depth)
        GSR_STEP(sort_u32_passes(gs.depth_key, nullptr, (uint32_t)n, gs.a_k, gs.a_v, gs.b_k, gs.b_v, four, 0, 3, stream, gs.sort_info + 1,
                                 gs.rect_idx, gs.a_r, gs.b_r, &side, depth_records ? gs.c_k : nullptr, depth_records ? gs.b_k : nullptr));
    else
        GSR_STEP(sort_u32_passes(gs.c_k, gs.c_v, (uint32_t)n, gs.a_k, gs.a_v, gs.b_k, gs.b_v, four, 0, 3, stream, gs.sort_info + 1,
                                 xy_plan ? gs.c_r : nullptr, xy_plan ? gs.a_r : nullptr, xy_plan ? gs.b_r : nullptr, nullptr,
                                 depth_records ? gs.a_k : nullptr, depth_records ? gs.b_k : nullptr));
    if (order_now) {
        // (behind the same event — it follows the history's last blend in stream order — and queued while the host would
        // only wait: nothing is added to the caller's stream, and by the time the blend is launched the order is there)
        // From here until this call's blend has been launched the order belongs to no call (a backward of an earlier one
        // must not take it: order_serial).
        hist->order_serial = 0;
        GSR_HIP_TRY(hipStreamWaitEvent(g_rb.side, g_rb.ev_r, 0));
        bool sorted = false;
        GSR_STEP(launch_tile_order(d, hist->ticks[hist->cur], t_ticks, hist->order, hist->stats_dev, g_rb.side, &sorted, hist->deep, shape));
        if (sorted) GSR_HIP_TRY(hipEventRecord(hist->ev_order, g_rb.side));
        else order_now = false;                               // (no room for the sort on this device: patch order)
    }
    GSR_HIP_TRY(hipEventSynchronize(g_rb.ev_r));
    // The reference's offsets are u32 (AuxBuffer.cuh:51): a frame whose instance count does not fit them would size
    // the binning chunk by the wrapped count while the emission writes per true count. Refused before anything
    // R-sized is touched (launch_sort_pairs draws the same line at n >= 0xFFFFFFFF).
    const unsigned long long true_total = (unsigned long long)g_rb.host[6] | ((unsigned long long)g_rb.host[7] << 32);
    if (true_total >= 0xFFFFFFFFull) return fail(GSR_ERR_TOO_LARGE);
    const bool side_way = g_rb.host[10] != 0u;             // (then the stream's keys share their top byte: three passes)
    const bool four_passes = g_rb.host[3] > 1u;
    const int nv = (int)g_rb.host[4];                      // V: the length of every depth-ordered array below
    const uint32_t side_m = side_way ? g_rb.host[12] : 0u, side_lo = side_way ? g_rb.host[11] : 0u;
    // (host[13]: the keys the compaction actually put on the side list — it must be the count the scan decided on, or
    // depth_side_kernel would rank entries of an earlier frame)
    // (without the compaction the side list is filled by the first depth pass, which may still be running: its count is not known here)
    if (side_way && (four_passes || side_m > kDepthSideMax || side_lo > side_m || side_m > (uint32_t)nv || (!fused_depth && g_rb.host[13] != side_m)))
        return fail(GSR_ERR_INTERNAL);
    if (four_passes)
        GSR_STEP(sort_u32_passes(gs.c_k, gs.c_v, (uint32_t)nv, gs.a_k, gs.a_v, gs.b_k, gs.b_v, four, 3, 4, stream, nullptr,
                                 xy_plan ? gs.c_r : nullptr, xy_plan ? gs.a_r : nullptr, xy_plan ? gs.b_r : nullptr));
    if (side_way)
        GSR_STEP(launch_depth_side(side, side_m, side_lo, (uint32_t)nv - side_m, gs.a_k, gs.a_v, xy_plan ? gs.a_r : nullptr, stream));
    // depth-sorted keys / indices, and the other pair of buffers (free from here on)
    uint32_t* const sorted_k = four_passes ? gs.b_k : gs.a_k - side_lo;
    uint32_t* const sorted_v = four_passes ? gs.b_v : gs.a_v - side_lo;
    const uint32_t* const sorted_r = four_passes ? gs.b_r : gs.a_r - side_lo;      // (xy_plan only)
    uint32_t* const spare_k = four_passes ? gs.a_k : gs.b_k;
    const uint32_t R = (uint32_t)true_total;               // (= pointOffsets[N - 1], GSCuda.cu:772)
    a->num_rendered = R;
    const float t_cutoff = inria ? 0.0001f : 0.001f;                                        // :653 / upstream
    if (R == 0) {
        // (colours beside the blend: there is no blend — the zeros of a frame without a tile are written here)
        if (colors_mode == 2) GSR_STEP(launch_colors_visible(colors_rest, rest_tiles, rest_shs, rest_rgb, stream));
        if (!inria) { issue_receipt((uint32_t)nv, nullptr); return fail(GSR_OK); }          // :775-778
        // upstream still runs the tile loop: every pixel gets the background
        GSR_HIP_TRY(hipMemsetAsync(img.ranges, 0, sizeof(uint32_t) * 2 * (size_t)num_tiles, stream));
        GSR_STEP(launch_blend(d, img.ranges, nullptr, geom.means2D, a->colors_precomp ? a->colors_precomp : geom.rgb,
                              geom.conic_opacity, img.accum_alpha, img.n_contrib, a->background, a->out_color, nullptr,
                              t_cutoff, stream));
        issue_receipt((uint32_t)nv, nullptr);               // (after the last step that can fail)
        return fail(GSR_OK);
    }

    char* bin_chunk = a->binning_alloc(a->binning_user, gsr_required_binning(R) + 128);    // :782-784
    if (!bin_chunk) return fail(GSR_ERR_ALLOC);
    gsr_binning_state bin;
    gsr_binning_from_chunk(bin_chunk, R, &bin);
    const BinScratch bs = carve_bin_scratch(bin.sorting_space, R);

    // Tile grids up to 255 x 255: the tile-column pass is produced directly by a column-major
    // emission and only the tile-row pass runs as a sort. Larger grids: depth-ordered emission and
    // 8-bit digit passes over the tile bits.
    // Two binning plans give the same sorted lists. "blocks": the lists are written directly by
    // tile-block owners (blockbin.hip), nothing R-sized is sorted. "sort": column-major emission +
    // one onesweep pass on the tile row. The block plan pays per (Gaussian, block) entry, so scenes
    // of tiny splats (few tiles per Gaussian) stay on the sort plan unless a flag forces one.
    bool use_blocks = xy_plan && blockbin_supported(d.grid_x, d.grid_y) && !(a->flags & GSR_FLAG_PLAN_SORT);
    // The block plan pays per (Gaussian, block) entry and per unit, the sort plan 36 bytes per instance: what decides is
    // the instances per VISIBLE Gaussian. Measured (binning without the blend, sort / blocks): R/V = 2.7 (50 M tiny splats)
    // 5.6 / 5.9 ms, 5.3 (the bench scene from far away) 0.98 / 1.00 ms, 7.5: 2.08 / 1.81 ms, 11: 2.76 / 1.58 ms, 88: 2x.
    // ... of the splats that ARE small: the sort plan's emission walks a Gaussian's columns and keys chunk by chunk of 512
    // Gaussians, and a few hundred background splats that cover a thousand tiles each (any trained scene seen from outside)
    // make its slowest chunks five times the others — 1 M flat splats + 500 huge ones from 48 units away, R/V = 2.8:
    // 1.53 against 1.08 ms; 5.8 M: 2.46 / 2.14 (`profiles/r05_trained_like.txt`). The scan has counted the instances of the
    // splats of kBigSplatTiles tiles and more (host words 14-15): with an eighth of the frame's instances in such splats the
    // frame goes to the block plan whatever its average.
    const unsigned long long big_instances = (unsigned long long)g_rb.host[14] | ((unsigned long long)g_rb.host[15] << 32);
    if (use_blocks && !(a->flags & GSR_FLAG_PLAN_BLOCKS))
        use_blocks = (uint64_t)R >= kBlockPlanMinInstances * (uint64_t)nv || 8ull * big_instances >= (unsigned long long)R;
    a->plan_used = use_blocks ? GSR_PLAN_BLOCKS : (xy_plan ? GSR_PLAN_SORT : GSR_PLAN_GENERIC);
    // (the block plan has no R-sized sort: sortingSpace then holds its unit tables, not look-back words)
    if (!use_blocks)
        GSR_HIP_TRY(hipMemsetAsync(bs.sweep.error_word, 0, 128 + 256 * sizeof(uint32_t), stream));   // error word + tile-row histogram
    // (before the streams fork: the blend may run on the side stream)
    if (count_staged) GSR_HIP_TRY(hipMemsetAsync(g_rb.staged_dev, 0, sizeof(unsigned long long), stream));
    bool forked = false, blend_from_lists = false;
    // (the other plans' blends run behind their lists, fed from them: what the history says of THIS frame's blend)
    if (history && !use_blocks) { hist->overlapped = false; hist->block_fed = false; }
    if (use_blocks) {
        // keysUnsorted / valuesUnsorted hold the block lists (rectangle | depth bits, index) in this plan
        GSR_STEP(launch_block_binning(nv, sorted_k, sorted_v, sorted_r, d.grid_x, d.grid_y, R, gs.block_scratch,
                                      bin.keys_unsorted, bin.values_unsorted, bin.sorting_space, img.ranges, inria, stream,
                                      profile ? g_rb.ev[2 * GSR_STAGE_DEPTH_ORDER + 1] : nullptr, gs.sort_info + 4,
                                      (a->flags & GSR_FLAG_NO_SORTED_LISTS) ? bin.values : nullptr, shape.cus));
        if (profile) {
            // depth order + block lists | unit masks + prefixes + ranges (recorded as "sort_pass1") | emission
            g_rb.ev_alias_begin(GSR_STAGE_SORT_PASS1, GSR_STAGE_DEPTH_ORDER);
            GSR_HIP_TRY(hipEventRecord(g_rb.ev[2 * GSR_STAGE_SORT_PASS1 + 1], stream));
            g_rb.recorded[GSR_STAGE_DEPTH_ORDER] = g_rb.recorded[GSR_STAGE_SORT_PASS1] = true;
        }
        // The blend of this plan reads the block lists, not the sorted lists, so it does not depend on the
        // emission. With GSR_FLAG_OVERLAP_EMIT the emission (bound by the HBM write path) runs beside the
        // blend (bound by vector ALU work) on a second stream, and the caller's stream waits for it before
        // gsr_forward's work is complete: 5 % shorter frames, but each of the two kernels runs ~20 % longer
        // while they share the chip, so per-kernel times are no longer those of the kernels alone.
        // By default the library decides: beside each other when the blend — what the tiles of the last calls took, spread
        // over the chip's 5 120 wave slots — is expected to be the shorter of the two (the emission: 12 R bytes at 5 TB/s);
        // a blend already running beside the emission takes about twice as long per tile, hence the second threshold.
        bool overlap = (a->flags & GSR_FLAG_OVERLAP_EMIT) != 0;
        // (not while the history's frames do not resemble each other: the last frame's tile times then say nothing about this one)
        if (!overlap && !(a->flags & GSR_FLAG_SERIAL_EMIT) && history && hist->mean != 0u && !hist->decorrelated &&
            (uint64_t)R >= kOverlapMinInstances * (uint64_t)nv) {
            const unsigned long long tiles = (unsigned long long)(d.row_end - d.row_begin) * (unsigned long long)d.grid_x;
            // how long the blend will take: the tiles' times spread over the chip's 5 120 wave slots — beside the emission, whose
            // persistent workgroups keep their registers, over the 3 072 it gets there —, but never less than the longest tile
            // (frames of small splats end on a few lone waves: the mean alone said 0.16 ms for a blend of 0.36)
            unsigned long long blend_ticks = std::max((unsigned long long)hist->mean * tiles /
                                                          (unsigned long long)(hist->overlapped ? shape.blend_slots_beside : shape.blend_slots),
                                                      (unsigned long long)hist->longest);
            // (times of a blend fed from the SORTED lists: out of the block lists a tile walks every unit of its block for its
            // entries — measured on the stand-in, block feed over sorted-list feed: 1.1 at 88 instances per visible Gaussian,
            // 1.44 at 23, 3 at 5 = 1 + 10 V / R)
            if (!hist->block_fed) blend_ticks = blend_ticks * ((unsigned long long)R + 10ull * (unsigned long long)nv) / (unsigned long long)R;
            // The emission: 12 R bytes at 4 TB/s (measured 4.9 on the bench frame's 3.2 GB, 3.8 on 1 GB, 3.3-4.6 on 0.46 GB) — then the blend must be the shorter of the two, beside a kernel that fills
            // the memory pipes it is throttled (the stand-in from outside the cloud, R/V = 22: 1.66 -> 1.99 ms) —, but never
            // under the 0.07 ms a wave takes for its share of one unit: a light frame's emission leaves the chip idle, and a
            // blend of up to twice that still gains beside it (1 M flat splats, frames of 0.4 ms; `profiles/r05_trained_like.txt`).
            // Once overlapped, the times are those of a blend that shares the chip: while the emission runs it advances at 0.46
            // of its pace (bench frame: 0.11 ms alone, 0.24 beside an emission that outlasts it), so b' = b / 0.46 if that ends
            // inside the emission e, else e + (b - 0.46 e). The time it would take alone is taken back out of b' and held to
            // the same limit, a tenth more (the stand-in from outside the cloud, entered from an overlapped pose, stayed
            // overlapped under a looser bound: 1.48 -> 1.70 ms, for good).
            const unsigned long long emit_bw = 12ull * (unsigned long long)R / 40000ull, emit_floor = 7000ull;
            unsigned long long limit = emit_bw >= emit_floor ? emit_bw : 2ull * emit_floor;
            // (a frame that is block-fed either way — 48 instances per visible Gaussian and more — changes nothing but the
            // company its blend keeps: there a blend of up to twice the emission still gains, 1 M-splat stand-in, emission
            // 0.10 ms, blend 0.13-0.24: 6-10 %; three times loses: the bench frame with faint splats)
            if (hist->block_fed && (uint64_t)R >= kBlockFeedMinInstances * (uint64_t)nv) limit *= 2ull;
            if (hist->overlapped) {
                const unsigned long long e = std::max(emit_bw, emit_floor);
                const unsigned long long alone = blend_ticks <= e ? blend_ticks * 46ull / 100ull : blend_ticks - e * 54ull / 100ull;
                overlap = 10ull * alone < 11ull * limit;
            } else {
                overlap = blend_ticks < limit;
            }
        }
        const bool serial = !overlap || (a->flags & GSR_FLAG_NO_SORTED_LISTS);
        if (history) hist->overlapped = !serial;
        if (!serial) a->plan_used |= GSR_PLAN_EMIT_OVERLAPPED;
        // Which lists feed the blend. Out of the block lists a tile walks every unit of its block and picks its entries
        // by mask: as good as the sorted list where a Gaussian covers most tiles of its blocks, but with small splats a
        // tile owns a few of a unit's 2048 entries and pays a round trip to memory per unit for them (the bench scene
        // from outside the cloud, R/V = 23: 0.65 against 0.45 ms; from far away, R/V = 5: 1.98 against 0.65 ms; bench
        // frame, R/V = 88: equal). With the sorted lists written anyway, sparse frames blend from them.
        blend_from_lists = serial && !(a->flags & GSR_FLAG_NO_SORTED_LISTS) &&
                           (uint64_t)R < (env.block_feed_min >= 0 ? (uint64_t)env.block_feed_min : kBlockFeedMinInstances) * (uint64_t)nv;
        if (history) hist->block_fed = !blend_from_lists;
        // (the emission stays on the caller's stream and is launched first: its persistent workgroups must be resident
        // before the blend's thousands of waves arrive — the other way round the blend takes every register file and the
        // emission starts when the blend is nearly over: no gain)
        if (!serial) {
            GSR_STEP(g_rb.ensure_side());
            GSR_HIP_TRY(hipEventRecord(g_rb.ev_fork, stream));
            GSR_HIP_TRY(hipStreamWaitEvent(g_rb.side, g_rb.ev_fork, 0));
            forked = true;
            colors_join.tail = true;                          // (until the join at the end of the call has been queued)
        }
        hipStream_t emit_stream = stream;
        // (What a gsr_backward call after this one may use — the block lists and, for its per-entry gradient sums, the
        // bytes of keysUnsorted — it works out from the receipt: lists_of_receipt.)
        if (blend_from_lists) a->plan_used |= GSR_PLAN_BLEND_FROM_LISTS;
        // GSR_FLAG_NO_SORTED_LISTS: this plan's blend reads the block lists, and no caller of the reference reads
        // BinningState (GSGaussians.cpp:214-219 maps GeometryState only): a forward-only caller may skip the 12 R
        // bytes of sorted keys / values altogether. keys / values are then left unwritten.
        if (a->flags & GSR_FLAG_NO_SORTED_LISTS) {
            a->plan_used |= GSR_PLAN_LISTS_SKIPPED;        // (values[0] = GSR_LISTS_SKIPPED_STAMP: written with the tile ranges)
        } else {
            if (profile) GSR_HIP_TRY(hipEventRecord(g_rb.ev[2 * GSR_STAGE_DUPLICATE], emit_stream));
            GSR_STEP(launch_block_emit(nv, d.grid_x, d.grid_y, R, gs.block_scratch, bin.keys_unsorted, bin.values_unsorted,
                                       bin.sorting_space, bin.keys, bin.values, emit_stream, forked, shape.cus));
            if (profile) {
                GSR_HIP_TRY(hipEventRecord(g_rb.ev[2 * GSR_STAGE_DUPLICATE + 1], emit_stream));
                g_rb.recorded[GSR_STAGE_DUPLICATE] = true;
            }
        }
    } else if (xy_plan) {
        uint32_t* hist_y = bs.sweep.hist;
        SweepScratch bsw = bs.sweep;
        bsw.error_word = err_r; bsw.error_value = serial;
        if (d.grid_y > 1) GSR_STEP(sweep_clear(bsw, R, (uint32_t)d.grid_y, stream));
        // one pass: with a single tile row the column-major list is already the sorted list
        uint64_t* emit_k = d.grid_y > 1 ? bin.keys_unsorted : bin.keys;
        uint32_t* emit_v = d.grid_y > 1 ? bin.values_unsorted : bin.values;
        // the depth-order stage ends and the emission stage starts at an event inside the launcher
        GSR_STEP(launch_emit_columns(nv, sorted_k, sorted_v, sorted_r, d.grid_x, d.grid_y, gs.emit_scratch, hist_y,
                                     emit_k, emit_v, stream, profile ? g_rb.ev[2 * GSR_STAGE_DEPTH_ORDER + 1] : nullptr,
                                     profile ? g_rb.ev[2 * GSR_STAGE_DUPLICATE] : nullptr));   // :787
        if (profile) {
            g_rb.recorded[GSR_STAGE_DEPTH_ORDER] = true;
            GSR_HIP_TRY(hipEventRecord(g_rb.ev[2 * GSR_STAGE_DUPLICATE + 1], stream));
            g_rb.recorded[GSR_STAGE_DUPLICATE] = true;
        }
        if (d.grid_y > 1) {
            DigitSpec sy;
            sy.mode = kDigitTileY; sy.shift = 0; sy.nbins = (uint32_t)d.grid_y; sy.grid_x = (uint32_t)d.grid_x;
            sy.inv_grid_x = 1.0f / (float)d.grid_x;
            GSR_BEGIN(GSR_STAGE_SORT_PASS2);
            GSR_STEP(sweep_pass_u64(bin.keys_unsorted, bin.values_unsorted, bin.keys, bin.values, R, sy, hist_y, bsw, stream, true));
            GSR_END(GSR_STAGE_SORT_PASS2);
        } else {
            // keep the "unsorted" arrays populated as the reference does
            GSR_HIP_TRY(hipMemcpyAsync(bin.keys_unsorted, bin.keys, 8 * (size_t)R, hipMemcpyDeviceToDevice, stream));
            GSR_HIP_TRY(hipMemcpyAsync(bin.values_unsorted, bin.values, 4 * (size_t)R, hipMemcpyDeviceToDevice, stream));
        }
    } else {
        GSR_STEP(launch_gather_counts(nv, sorted_k, sorted_v, geom.tiles_touched, spare_k, stream));
        GSR_STEP(launch_inclusive_scan(spare_k, spare_k, (size_t)nv, gs.scan_temp, stream));
        GSR_END(GSR_STAGE_DEPTH_ORDER);
        GSR_BEGIN(GSR_STAGE_DUPLICATE);
        GSR_STEP(launch_duplicate(nv, sorted_k, sorted_v, spare_k, geom, radii, rects_in, d, bin.keys_unsorted,
                                  bin.values_unsorted, nullptr, nullptr, stream));         // :787
        GSR_END(GSR_STAGE_DUPLICATE);
        const int end_bit = 32 + (int)gsr_higher_msb((uint32_t)num_tiles);                 // :791
        GSR_BEGIN(GSR_STAGE_SORT_PASS2);
        GSR_STEP(launch_sort_pairs(bin.keys_unsorted, bin.keys, bin.values_unsorted, bin.values, R, 32, end_bit,
                                   bin.sorting_space, stream, err_r, serial));
        GSR_END(GSR_STAGE_SORT_PASS2);
    }
    // :800-801 — under the block plan the ranges are the tile starts it has already computed
    if (!use_blocks) {
        GSR_BEGIN(GSR_STAGE_RANGES);
        GSR_STEP(launch_tile_ranges(bin.keys, R, img.ranges, num_tiles, inria, stream, gs.sort_info + 4));
        GSR_END(GSR_STAGE_RANGES);
    }
    const float* colors = a->colors_precomp ? a->colors_precomp : geom.rgb;                // :803
    hipStream_t blend_stream = forked ? g_rb.side : stream;
    // (the colours: long since written — the blend's stream is made to wait only if they are not; a blend on the side stream
    // follows them in stream order, and the caller's stream joins that stream below)
    if (colors_join.pending) {
        if (blend_stream == stream && hipEventQuery(g_rb.ev_colors) != hipSuccess) {
            (void)hipGetLastError();
            GSR_HIP_TRY(hipStreamWaitEvent(stream, g_rb.ev_colors, 0));
        }
        colors_join.pending = nullptr;
        a->plan_used |= GSR_PLAN_COLORS_BESIDE;
    }
    // (the order: long since sorted — the blend's stream is made to wait only if it is not)
    if (order_now && hipEventQuery(hist->ev_order) != hipSuccess) {
        (void)hipGetLastError();                              // ("not ready" is no error of this call)
        GSR_HIP_TRY(hipStreamWaitEvent(blend_stream, hist->ev_order, 0));
    }
    const uint32_t* const t_order = order_now ? hist->order : nullptr;
    if (order_now && !hist->decorrelated) a->plan_used |= GSR_PLAN_TILES_REORDERED;
    if (history && hist->decorrelated) a->plan_used |= GSR_PLAN_TILE_ORDER_DROPPED;
    // Colours beside the blend (scenes beyond 16 M Gaussians): the blend takes them from the SH array; geomState.rgb is
    // written meanwhile on the other stream — or, where the blend itself runs on the second stream beside the emission,
    // behind it there — and the caller's stream waits for it before the call's work is complete.
    const bool colors_late = colors_mode == 2;
    if (colors_late && !forked) {
        GSR_HIP_TRY(hipEventRecord(g_rb.ev_pre_blend, stream));
        GSR_HIP_TRY(hipStreamWaitEvent(g_rb.side, g_rb.ev_pre_blend, 0));
        colors_join.tail = true;
        GSR_STEP(launch_colors_visible(colors_rest, rest_tiles, rest_shs, rest_rgb, g_rb.side));
        GSR_HIP_TRY(hipEventRecord(g_rb.ev_colors, g_rb.side));
        colors_join.pending = g_rb.ev_colors;                 // (joined when this function is left)
        colors_join.tail = false;
    }
    if (colors_late) { colors = a->shs; a->plan_used |= GSR_PLAN_COLORS_BESIDE; }
    // (deep tiles, blend.hip: the leading entries of an order sorted for THIS call; the block-fed blend has none)
    const bool list_fed = !(use_blocks && !blend_from_lists);
    const bool deep_wanted = env.deep_by_history && list_fed && t_order != nullptr && !hist->decorrelated && !(a->flags & GSR_FLAG_NO_DEEP_TILES);
    const uint32_t deep_forced = a->flags & (GSR_FLAG_DEEP_TILES_ALL | GSR_FLAG_DEEP_WAVES_8 | GSR_FLAG_DEEP_WAVES_16);
    // (not where geomState.rgb is written BESIDE the blend — scenes beyond 16 M Gaussians, colors_late above —: eight deep
    // workgroups a CU hold every vector register of its SIMDs, the colours kernel waits for them to retire and the frame for
    // the colours kernel: 50 M Gaussians 5.09 -> 5.33 ms, 5.13-5.22 with the blend kept to 5-6 workgroups a CU by idle LDS;
    // with the colours passed as colorsPrecomp there is no such kernel: 4.37 -> 4.29)
    const bool colours_beside_blend = colors_mode == 2;
    const bool deep_by_rule = !colours_beside_blend && !(a->flags & GSR_FLAG_NO_DEEP_TILES) &&
                              (uint64_t)R < (env.deep_all_max >= 0 ? (uint64_t)env.deep_all_max : kDeepAllMaxInstances) * (uint64_t)nv;
    const bool deep_all = list_fed && (deep_forced != 0u || deep_by_rule);
    // How many waves a deep tile gets: four — or eight, sixteen where the history says the frame's work sits in few tiles:
    // tiles x mean / longest is how many tiles AS LONG AS THE LONGEST the frame amounts to; with fewer of them than the chip has
    // SIMDs eight waves per tile win, with fewer than a quarter sixteen (measured, blend with 4 / 8 / 16 waves per tile,
    // `profiles/r06_deep_tiles.txt` — a trained-like scene of 5.83 M splats from 32 / 48 / 70 units away, 358 / 200 / 96 such
    // tiles: 1.15 / 1.01 / 1.03, 1.82 / 1.66 / 1.17, 3.20 / 2.88 / 2.20 ms (one wave per tile: 1.77, 3.06, 5.86); the bench
    // scene from 40 / 50 units, 948 / 422: 0.297 / 0.283 / 0.68 and 0.240 / 0.212 / 0.30; from 30 units, 1 609: 0.31 / 0.38 / 1.03)
    int deep_waves = (deep_forced & GSR_FLAG_DEEP_WAVES_16) ? 16 : ((deep_forced & GSR_FLAG_DEEP_WAVES_8) ? 8 : 4);
    if (deep_all && deep_forced == 0u && env.deep_waves_auto && history && hist->mean != 0u && hist->longest != 0u && !hist->decorrelated) {
        const unsigned long long tiles = (unsigned long long)(d.row_end - d.row_begin) * (unsigned long long)d.grid_x;
        const unsigned long long as_longest = tiles * (unsigned long long)hist->mean / (unsigned long long)hist->longest;
        const unsigned long long simds = 4ull * (unsigned long long)shape.cus;
        deep_waves = 4ull * as_longest <= simds ? 16 : (as_longest <= simds ? 8 : 4);
    }
    if (deep_wanted || deep_all) a->plan_used |= GSR_PLAN_DEEP_TILES;
    if (profile) GSR_HIP_TRY(hipEventRecord(g_rb.ev[2 * GSR_STAGE_BLEND], blend_stream));
    if (use_blocks && !blend_from_lists)
        GSR_STEP(launch_blend_blocks(nv, d, R, gs.block_scratch, bin.values_unsorted, bin.sorting_space, img.ranges, geom.means2D,
                                     colors, geom.conic_opacity, img.accum_alpha, img.n_contrib, a->background, a->out_color,
                                     count_staged ? g_rb.staged_dev : nullptr, t_cutoff, blend_stream, t_order, t_ticks, colors_late));
    else
        GSR_STEP(launch_blend(d, img.ranges, bin.values, geom.means2D, colors, geom.conic_opacity, img.accum_alpha,
                              img.n_contrib, a->background, a->out_color, count_staged ? g_rb.staged_dev : nullptr,
                              t_cutoff, blend_stream, gs.sort_info + 4, R, t_order, t_ticks, colors_late,
                              deep_wanted ? hist->deep : nullptr, deep_all, deep_waves));   // :804-810
    if (order_now) hist->order_serial = serial;               // (the blend that takes the order is in its stream: a backward of this call may take it too)
    if (profile) { GSR_HIP_TRY(hipEventRecord(g_rb.ev[2 * GSR_STAGE_BLEND + 1], blend_stream)); g_rb.recorded[GSR_STAGE_BLEND] = true; }
    if (forked) {                                                           // the image is complete when the side stream is
        if (colors_late) GSR_STEP(launch_colors_visible(colors_rest, rest_tiles, rest_shs, rest_rgb, g_rb.side));   // (beside the rest of the emission)
        GSR_HIP_TRY(hipEventRecord(g_rb.ev_join, g_rb.side));
        GSR_HIP_TRY(hipStreamWaitEvent(stream, g_rb.ev_join, 0));
        colors_join.tail = false;
    }

    if (profile || count_staged) {
        if (count_staged)
            GSR_HIP_TRY(hipMemcpyAsync(g_rb.staged_host, g_rb.staged_dev, sizeof(unsigned long long),
                                       hipMemcpyDeviceToHost, stream));
        GSR_HIP_TRY(hipStreamSynchronize(stream));
        if (count_staged) a->records_staged = *g_rb.staged_host;
        if (profile) {
            for (int s = 0; s < GSR_NUM_STAGES; ++s) {
                if (!g_rb.recorded[s]) continue;
                float ms = 0.0f;
                GSR_HIP_TRY(hipEventElapsedTime(&ms, g_rb.ev[g_rb.begin_of[s]], g_rb.ev[2 * s + 1]));
                a->stage_ms[s] = ms;
            }
        }
    }
#undef GSR_BEGIN
#undef GSR_END
#undef GSR_STEP
    issue_receipt((uint32_t)nv, bin_chunk);
    return fail(GSR_OK);
}

}  // extern "C"
