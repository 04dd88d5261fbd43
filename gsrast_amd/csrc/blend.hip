// Per-tile front-to-back alpha blend of the sort plan: one 16x16 screen tile per wavefront (four pixels per
// lane), Gaussian records pulled from HBM 64 at a time into wave-private LDS and composited with early-outs
// decided by wave64 __ballot. (The block plan's blend, fed from the block lists, is in blockbin.hip; both
// share blend_core.hpp.)
//
// Semantics follow reference apps/gsrast/gscuda/GSCuda.cu:543-677 (renderCUDA):
//   integer pixel centres (no +0.5), power = -0.5(A dx^2 + C dy^2) - B dx dy, skip power > 0,
//   alpha = min(0.99, opacity * exp(power)), skip alpha < 1/255, stop when T(1-alpha) < 0.001,
//   C += rgb * alpha * T, out = C + T * background, finalT, nContrib = index (1-based) of the
//   last contributing record. The batch size (256) and the "whole tile done" test at the top
//   of each batch are the reference's, so the number of records staged (R_f) is identical.
// Differences that do not change any pixel: a record no lane of the wave can see (power > 0 or below the
// 1/255 cut for every lane) is skipped after the power evaluation by one ballot.

#include <algorithm>
#include <map>
#include <type_traits>

#include "blend_core.hpp"

namespace gsr {
namespace {

struct BlendParams {
    const uint2* ranges;
    const uint32_t* point_list;
    const float2* means2D;
    const float* colors;           // vec3, 12-byte stride
    const float4* conic_opacity;
    float* final_t;
    uint32_t* n_contrib;
    const float* background;
    float* out_color;
    unsigned long long* staged_counter;
    float t_cutoff;                // transmittance below which a pixel is finished (0.001 gscuda, 1e-4 upstream)
    FrameDims dims;
    int num_tiles;                 // tiles in [row_begin,row_end)
    const uint32_t* nonempty;      // tiles of the frame that have a list (device word), or null: never four waves per tile
    int base_workgroups;           // workgroups of one wave per tile; the launch holds four times as many when nonempty is given
    uint32_t num_rendered;         // R of the call (the lists' total length)
    TileOrder history;             // longest tiles first (blend_core.hpp)
    uint32_t dc_stride;            // 0, or 48: `colors` is the SH array (TileFeed::dc_stride)
};

// One wave per tile leaves most of the chip idle when few tiles have a list, and the frame lasts as long as the slowest
// of them. Four waves then share a tile: one 16 x 4 strip each, every wave walking the whole list but keeping, at
// staging, only what can reach ITS strip. That repeats the walk four times and divides the compositing by up to four,
// so it pays where the chip has room for the repeated walk (few tiles), or where the walk is short and the compositing
// is the work (a scene seen from far away: short lists of splats smaller than a tile — eye (0,0,-50) on the bench
// scene, 3 712 tiles with a list, 210 entries each on average: blend 0.84 -> 0.34 ms). With thousands of deep lists
// the repeated walk costs more than the shorter chains save (full-screen frames with four waves per tile: 0.46 -> 1.27
// ms from outside the cloud, 0.66 -> 1.01 ms from (0,0,-30): `gpurun_out/s12`).
constexpr uint32_t kStripTilesAny = 1536;        // tiles with a list up to which four waves share a tile whatever the lists
constexpr uint32_t kStripTilesShort = 4096;      // ... and up to which they do when the lists are short:
constexpr uint32_t kStripMeanList = 1024;        // entries per tile with a list, on average
constexpr uint32_t kPriorityMeanList = 8192;     // entries per tile with a list from which deep tiles are given issue priority

// ---- one wave per tile, four pixels per lane ------------------------------------------------
// Lane l owns pixels (x = l & 15, y = (l >> 4) + 4 k), k = 0..3, so slot k of the wave is the
// 16 x 4 strip of rows 4k..4k+3. dx and the terms that only depend on it are computed once per
// record and lane; the per-row terms run as packed f32 pairs (v_pk_mul_f32 / v_pk_add_f32: two
// IEEE single operations per issue, same rounding as the scalar form, no fused multiply-add), so
// the power evaluation costs 18 issues for four pixels instead of 44. Records are staged 64 at
// a time, one per lane, in wave-private LDS: no workgroup barrier anywhere. The batch size of the
// reference (256) only survives as the granularity of the "whole tile done" test and of the
// staged-record count R_f, which therefore stay identical to the reference's.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5))) void blend_wave_kernel(const BlendParams p) {
    __shared__ StagedRecords s_staged;
    exp_table_init(s_staged.exp_tab, (int)threadIdx.x);      // (wave-private LDS: ordered inside the wave)

    // workgroups [k * base, (k + 1) * base) are strip k of the tiles: the three extra sets leave at once unless the frame
    // has few tiles with a list (they are the END of the launch, and a tile's four waves run on one XCD)
    const int strip = (int)blockIdx.x / p.base_workgroups;
    bool strips = false, prioritise = false;
    if (p.nonempty != nullptr) {
        const uint32_t ne = *p.nonempty;
        strips = ne <= kStripTilesAny || (ne <= kStripTilesShort && (unsigned long long)p.num_rendered <= (unsigned long long)kStripMeanList * ne);
        prioritise = !strips && (unsigned long long)p.num_rendered >= (unsigned long long)kPriorityMeanList * ne;    // (see set_tile_priority)
    }
    if (strip != 0 && !strips) return;
    const uint32_t clock_begin = tile_clock();
    const int wg = (int)blockIdx.x - strip * p.base_workgroups;
    const int tile_local = tile_of_workgroup(p.history.order ? (int)p.history.order[wg] : wg, p.dims.grid_x, p.dims.row_end - p.dims.row_begin);
    if (tile_local < 0) return;
    const int tile = p.dims.row_begin * p.dims.grid_x + tile_local;
    const int tx = tile % p.dims.grid_x, ty = tile / p.dims.grid_x;
    const int lane = threadIdx.x;
    TileLanes s;
    tile_lanes_init(s, tx, ty, lane, p.dims.width, p.dims.height, strips ? strip : -1);
    const uint2 range = p.ranges[tile];
    const uint32_t total = range.y - range.x;          // unsigned wrap as in the reference
    unsigned long long staged = 0;
    bool all_done = tile_lanes_all_done(s);

    TileFeed feed;
    feed.means2D = p.means2D; feed.colors = p.colors; feed.conic_opacity = p.conic_opacity;
    feed.dc_stride = p.dc_stride;
    feed.box = tile_box(tx, ty, p.dims.width, p.dims.height);
    if (strips) {
        feed.box.y_lo = (float)(ty * kTile + 4 * strip);
        feed.box.y_hi = (float)min(ty * kTile + 4 * strip + 3, p.dims.height - 1);
    }
    feed.total = total; feed.t_cutoff = p.t_cutoff;
    // batch k = list positions [64 k, 64 k + 64); ids are fetched two batches ahead, records one batch ahead
    auto next_batch = [&](uint32_t pos) {
        RecordBatch nb;
        nb.valid = pos < total;
        if (nb.valid) {
            const uint32_t cnt = min((uint32_t)kWave, total - pos);
            nb.mask = cnt == kWave ? ~0ull : ((1ull << cnt) - 1ull);
            nb.pos = pos;
            if ((uint32_t)lane < cnt) nb.id = p.point_list[range.x + pos + (uint32_t)lane];
        }
        return nb;
    };
    // (two copies of the loop: the one without the priority updates is, instruction for instruction, the loop of the frames
    // that do not use them — with one loop and a flag tested inside it, eye (0,0,-30) ran 3 % slower for nothing)
    auto walk = [&](auto with_priority) {
        if (with_priority.value) set_tile_priority(total);
        RecordBatch b0 = next_batch(0);
        fetch_records(b0, feed);
        RecordBatch b1 = next_batch(kWave);
        for (uint32_t pos = 2 * kWave; b0.valid && !all_done; pos += kWave) {
            fetch_records(b1, feed);
            RecordBatch b2 = next_batch(pos);
            all_done = stage_and_composite(s, feed, s_staged, b0, staged);
            b0 = b1;
            b1 = b2;
            if (with_priority.value) set_tile_priority(total - min(total, pos));
        }
    };
    if (prioritise) walk(std::true_type{});
    else walk(std::false_type{});
    tile_lanes_write(s, p.dims.width, p.dims.height, p.background, p.final_t, p.n_contrib, p.out_color);
    if (p.staged_counter && lane == 0) atomicAdd(p.staged_counter, staged);
    if (p.history.ticks && lane == 0 && strip == 0) p.history.ticks[tile] = tile_clock() - clock_begin;
}

// Workgroup numbers of the patch order, the SLOW tiles of the frame before first — those that took more than twice the
// mean, longest first — and everybody else behind them in patch order: one workgroup, a bitonic sort of (class, workgroup)
// in LDS. All tiles sorted by their time (or the slow ones counted from 0.5 to 1.5 times the mean), measured on ten frames:
// the frames of short lists gain more — bench frame 0.136 -> 0.112 ms, (0,0,-9) 0.221 -> 0.162, (0,0,-14) 0.46 -> 0.42 —
// but the frames whose tiles all take long lose 6-19 % ((0,0,-20), faint splats, 4K from outside): neighbours no longer run
// together on one XCD's L2. What separates the two groups on every frame measured is the tile time per wave slot (78-243
// us against 313-2 131 us): below 250 us the frame is sorted whole, above it only the tiles beyond twice the mean move —
// there nobody loses. The padding of the patch grid sorts to the end. `stats` (pinned host words): the longest tile, the mean.
//
// Whether the frame before says anything about this one is read off the frame before THAT (ticks_before): similarity =
// sum over the tiles of min(t / T, t' / T'), each frame's times as shares of its own total T — how much of the time goes
// to the same tiles, whatever the level (the same frame blended beside the emission takes twice as long in every tile).
// A camera that moves, or stands still, keeps it near one; a caller that draws another view every call (a trainer; two
// views alternating on one history) does not, and an order made from an unrelated frame is a random order: 5-28 % slower
// than the patch order. Below kMinSimilarity the patch order goes out (stats[4] = 1: the host then stops asking for the
// order except every fourth call, which looks again). A frame before without any times (the history is new, or was
// cleared) counts as similar.
// (bench scene, 1920 x 1080, scripts/history_similarity.py, `profiles/r05_history_similarity.txt`)
// along bench.py's camera path (0.31 units a frame) 0.84 to 0.97, median 0.90; between unrelated views 0.46 to 0.90, median 0.81
constexpr uint32_t kMinSimilarity = 800;         // of 1000
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint32_t* __restrict__ ticks, const uint32_t* __restrict__ ticks_before,
                                                          uint32_t* __restrict__ order,
                                                          int workgroups, int padded, int grid_x, int row_begin, int rows,
                                                          uint32_t* __restrict__ stats) {
    extern __shared__ uint32_t s_key[];
    __shared__ unsigned long long s_sum, s_common, s_before;
    __shared__ uint32_t s_cnt, s_max;
    if (threadIdx.x == 0) { s_sum = 0; s_cnt = 0; s_max = 0; s_common = 0; s_before = 0; }
    __syncthreads();
    unsigned long long sum = 0, sum_before = 0;
    uint32_t cnt = 0, mx = 0;
    for (int i = threadIdx.x; i < workgroups; i += 1024) {
        const int tile_local = tile_of_workgroup(i, grid_x, rows);
        if (tile_local >= 0) {
            const uint32_t t = ticks[row_begin * grid_x + tile_local];
            sum += t; mx = max(mx, t); ++cnt;
            sum_before += ticks_before[row_begin * grid_x + tile_local];
        }
    }
    atomicAdd(&s_sum, sum);
    atomicAdd(&s_cnt, cnt);
    atomicMax(&s_max, mx);
    atomicAdd(&s_before, sum_before);
    __syncthreads();
    // shares in units of 2^-20 of the frame's total (a tile's time is below 2^32, times 2^20 fits 64 bits)
    unsigned long long common = 0;
    const unsigned long long total = max(s_sum, 1ull), total_before = max(s_before, 1ull);
    for (int i = threadIdx.x; i < workgroups; i += 1024) {
        const int tile_local = tile_of_workgroup(i, grid_x, rows);
        if (tile_local >= 0) {
            const unsigned long long a = ((unsigned long long)ticks[row_begin * grid_x + tile_local] << 20) / total;
            const unsigned long long b = ((unsigned long long)ticks_before[row_begin * grid_x + tile_local] << 20) / total_before;
            common += min(a, b);
        }
    }
    atomicAdd(&s_common, common);
    __syncthreads();
    const uint32_t similarity = s_before == 0ull ? 1000u : (uint32_t)min(1000ull, (1000ull * s_common) >> 20);
    const bool dropped = similarity < kMinSimilarity;
    // (a LIGHT frame — less than 250 us of tile time per wave slot — is sorted whole: its lists are short, it lasts as long as
    // its longest tiles whatever the neighbours do; see kLightFrameTicks)
    const unsigned long long slow_from = s_sum < kLightFrameTicks ? 0ull : 2ull * s_sum / max(s_cnt, 1u);
    // what the host decides on whether the next calls need an order at all: {fresh, longest tile, mean, similarity, dropped}
    if (stats && threadIdx.x == 0) {
        stats[1] = s_max; stats[2] = (uint32_t)(s_sum / max(s_cnt, 1u)); stats[3] = similarity; stats[4] = dropped ? 1u : 0u;
        __threadfence_system();
        stats[0] = 1u;
    }
    if (dropped) {                                   // (the patch order itself)
        for (int i = threadIdx.x; i < workgroups; i += 1024) order[i] = (uint32_t)i;
        return;
    }
    for (int i = threadIdx.x; i < padded; i += 1024) {
        uint32_t key = 0xFFF00000u | (uint32_t)i;
        if (i < workgroups) {
            const int tile_local = tile_of_workgroup(i, grid_x, rows);
            if (tile_local >= 0) {
                const uint32_t t = ticks[row_begin * grid_x + tile_local];
                key = ((t > slow_from ? 511u - quantise_ticks(t) : 512u) << 20) | (uint32_t)i;
            }
        }
        s_key[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= padded; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < padded / 2; t += 1024) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;       // the pair (lo, lo + j)
                const bool up = (lo & k) == 0;
                const uint32_t a = s_key[lo], b = s_key[hi];
                if ((a > b) == up) { s_key[lo] = b; s_key[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < workgroups; i += 1024) order[i] = s_key[i] & 0xFFFFFu;
}

// Stage entry point for the tests: the footprint test of blend_core.hpp on n (record, tile) pairs.
__global__ __launch_bounds__(256) void footprint_test_kernel(int n, const float2* __restrict__ xy, const float4* __restrict__ co,
                                                             const int2* __restrict__ tile, int width, int height,
                                                             uint8_t* __restrict__ misses) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    misses[i] = record_misses_tile(xy[i], co[i], tile_box(tile[i].x, tile[i].y, width, height)) ? 1 : 0;
}

// Stage entry point for the tests: the blend's exponential (blend_core.hpp, exp_ref), argument by argument.
__global__ __launch_bounds__(64) void exp_test_kernel(int n, const float* __restrict__ in, float* __restrict__ out) {
    __shared__ unsigned long long s_exp[32];
    exp_table_init(s_exp, (int)threadIdx.x);
    for (int i = blockIdx.x * 64 + threadIdx.x; i < n; i += gridDim.x * 64) out[i] = exp_ref(in[i], s_exp);
}

}  // namespace

int launch_exp_test(int n, const float* in, float* out, hipStream_t stream) {
    if (n <= 0) return GSR_OK;
    hipLaunchKernelGGL(exp_test_kernel, dim3((unsigned)std::min(65536, (n + 63) / 64)), dim3(64), 0, stream, n, in, out);
    GSR_LAUNCH_CHECK("exp_test_kernel");
    return GSR_OK;
}

int launch_footprint_test(int n, const float* xy, const float* conic_opacity, const int32_t* tile_xy, int width, int height,
                          uint8_t* misses, hipStream_t stream) {
    if (n <= 0) return GSR_OK;
    hipLaunchKernelGGL(footprint_test_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n,
                       reinterpret_cast<const float2*>(xy), reinterpret_cast<const float4*>(conic_opacity),
                       reinterpret_cast<const int2*>(tile_xy), width, height, misses);
    GSR_LAUNCH_CHECK("footprint_test_kernel");
    return GSR_OK;
}

int launch_blend(const FrameDims& d, const uint32_t* ranges, const uint32_t* point_list,
                 const float* means2D, const float* colors, const float* conic_opacity,
                 float* final_t, uint32_t* n_contrib, const float* background, float* out_color,
                 unsigned long long* staged_counter, float t_cutoff, hipStream_t stream, const uint32_t* nonempty_tiles,
                 uint32_t num_rendered, const uint32_t* tile_order, uint32_t* tile_ticks, bool colors_are_shs) {
    BlendParams p;
    p.dc_stride = colors_are_shs ? 48u : 0u;
    p.history.order = tile_order; p.history.ticks = tile_ticks;
    p.num_rendered = num_rendered;
    p.ranges = reinterpret_cast<const uint2*>(ranges);
    p.point_list = point_list;
    p.means2D = reinterpret_cast<const float2*>(means2D);
    p.colors = colors;
    p.conic_opacity = reinterpret_cast<const float4*>(conic_opacity);
    p.final_t = final_t;
    p.n_contrib = n_contrib;
    p.background = background;
    p.out_color = out_color;
    p.staged_counter = staged_counter;
    p.t_cutoff = t_cutoff;
    p.dims = d;
    p.num_tiles = (d.row_end - d.row_begin) * d.grid_x;
    if (p.num_tiles <= 0) return GSR_OK;
    // (not when the staged records are counted: that count is per tile, the reference's "whole tile done" test)
    p.nonempty = staged_counter ? nullptr : nonempty_tiles;
    p.base_workgroups = patch_workgroups(d.grid_x, d.row_end - d.row_begin);
    hipLaunchKernelGGL(blend_wave_kernel, dim3((unsigned)(p.base_workgroups * (p.nonempty ? 4 : 1))), dim3(kWave), 0, stream, p);
    GSR_LAUNCH_CHECK("blend_wave_kernel");
    return GSR_OK;
}

// (see TileOrder, blend_core.hpp) `order` receives patch_workgroups(...) entries; asynchronous on stream
int tile_order_workgroups(const FrameDims& d) { return patch_workgroups(d.grid_x, d.row_end - d.row_begin); }
int launch_tile_order(const FrameDims& d, const uint32_t* ticks, const uint32_t* ticks_before, uint32_t* order, uint32_t* stats,
                      hipStream_t stream, bool* sorted) {
    *sorted = false;
    const int workgroups = tile_order_workgroups(d);
    if (workgroups <= 0) return GSR_OK;
    int padded = 2;
    while (padded < workgroups) padded <<= 1;
    if (padded > kTileOrderMax) return GSR_ERR_INVALID_ARG;
    if ((size_t)padded * sizeof(uint32_t) > 48 * 1024) {
        // more than the default 48 KB of dynamic LDS: asked for once per device; a device that has not got the 128 KB goes
        // without the order (the call is none the worse for it)
        static thread_local std::map<int, bool> lds_ok;
        int dev = 0;
        GSR_HIP_TRY(hipGetDevice(&dev));
        auto it = lds_ok.find(dev);
        if (it == lds_ok.end()) {
            const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(tile_order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                kTileOrderMax * (int)sizeof(uint32_t)) == hipSuccess;
            if (!ok) (void)hipGetLastError();
            it = lds_ok.emplace(dev, ok).first;
        }
        if (!it->second) return GSR_OK;
    }
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), (size_t)padded * sizeof(uint32_t), stream, ticks, ticks_before, order,
                       workgroups, padded, d.grid_x, d.row_begin, d.row_end - d.row_begin, stats);
    GSR_LAUNCH_CHECK("tile_order_kernel");
    *sorted = true;
    return GSR_OK;
}

}  // namespace gsr
