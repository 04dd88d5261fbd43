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

#include <stdio.h>
#include <stdlib.h>

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
    const uint32_t* deep_count;    // device word: leading entries of the order that are DEEP tiles (four waves, one walk), or null
    int deep_all;                  // every tile is one (GSR_FLAG_DEEP_TILES_ALL)
    uint32_t dc_stride;            // 0, or 48: `colors` is the SH array (TileFeed::dc_stride)
};

// One wave per tile leaves most of the chip idle when few tiles have a list, and the frame lasts as long as the slowest
// of them: calls with few tiles with a list (a rank's band of a sharded frame; a scene seen from far away: short lists of
// splats smaller than a tile) give every tile the four waves of a deep tile (below), decided on the device from the
// count of tiles that got a list. Rounds 2-5 had four INDEPENDENT waves for such tiles, each walking the whole list for
// its 16 x 4 strip (eye (0,0,-50) on the bench scene, 3 712 tiles with a list, 210 entries each: blend 0.84 -> 0.34 ms
// then, 0.24 with the shared walk, 0.21 with eight waves).
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
//
// ---- DEEP tiles: four waves per tile, ONE walk ------------------------------------------------
// A frame's blend lasts as long as its slowest tiles, and a wave alone on its SIMD issues a vector instruction every five
// cycles where the SIMD could take one every two (scripts/micro/valu_issue.hip): from (0,0,-30) a tenth of the tiles runs
// beyond 335 us, one wave each, while the rest of the chip has long finished (`profiles/r04_blend_tile_times.txt`). A deep
// tile gets a whole workgroup — which tiles: every tile of a frame of fewer than 16 instances per visible Gaussian (the
// host's rule, api.hip: kDeepAllMaxInstances), every tile of a call with few tiles (`strips`, above), or, behind
// GSR_DEEP_BY_HISTORY, the `deep_count` leading entries of the history's order (tile_order_kernel) —: the four waves WALK
// the list together — a round is 256 list entries, wave w fetches, culls and stages entries
// [64 w, 64 w + 64) of it into segment w of the shared staging area, with every survivor the strips it can reach at all —
// and each wave COMPOSITES one 16 x 4 strip over the four segments in list order, taking only the slots that can reach its
// strip. The list is read once and culled once per tile; per record a wave pays the filter and one strip's evaluation
// instead of four. T, nContrib and every decision are those of one wave per tile (same functions, same order per pixel);
// the staged-record count follows from WHERE the tile finished: the reference stages a batch of 256 when its first
// position is reached with some pixel unfinished, i.e. every batch that starts before the record the last pixel
// finished on. One or two barriers a round (two staging areas or one), which wait for the wave's LDS traffic only.
// Workgroups of the MIXED launch (blend_group_kernel): [0, base) one DEEP tile each (those beyond the deep count leave at
// once), then groups of four ordinary tiles, a wave each (no barrier there: the waves never meet).
constexpr int kGroupWaves = 4;
constexpr uint32_t kDeepGainX16 = 40;            // a deep tile takes 1 / 2.5 of one wave's time: what its recorded time counts as (tile_order_kernel)
constexpr uint32_t kDeepFracX16 = 6;             // GSR_DEEP_BY_HISTORY: deep from 3/8 of the longest estimate
constexpr uint32_t kDeepFloorTicks = 4000;       // ... but never below 40 us
constexpr uint32_t kDeepFlag = 0x80000000u;      // in a tile's recorded time: it was composited by four waves
// MODE 0, blend_wave_kernel: the launch of the frames that have no use for deep tiles (the host knows: launch_blend) — a workgroup is ONE
// wave and one tile, instruction for instruction the kernel of the rounds before (the workgroups of four hold their LDS and
// their place until the slowest of their four tiles is through: 3-8 % of the blend at 13-21 instances per visible Gaussian).
// MODE 2, blend_deep_kernel: EVERY tile of the launch is a deep one (the host decides: launch_blend) — no ordinary tile's
// registers to carry (one pixel per lane: 64 of them instead of 96), one staging area instead of two (13 KB of LDS instead of
// 27: a second barrier a round), so that eight of these workgroups fit a CU where five of the mixed kind do: at 4.8 waves per
// SIMD the mixed kernel kept the vector pipes busy half of the time (`profiles/r06_deep_tiles.txt`).
enum { kBlendSingle = 0, kBlendGrouped = 1, kBlendDeepOnly = 2 };
// W: the waves of a workgroup — 4; 8 or 16 in the deep-only kernel for frames whose work sits in a few hundred tiles (a far
// view of a dense scene): strips of 16 x 2 or 16 x 1 pixels, the upper lanes of a wave idle — on a chip with a wave per
// SIMD idle lanes cost nothing, and two waves on a SIMD issue twice what one does (`profiles/r06_micro_valu_issue.txt`).
template <int MODE, int W = kGroupWaves>
__device__ __forceinline__ void blend_wave_body(const BlendParams& p) {
    constexpr bool GROUPED = MODE != kBlendSingle, DEEP_ONLY = MODE == kBlendDeepOnly;
    static_assert(W == kGroupWaves || DEEP_ONLY, "more than four waves: the deep-only kernel");
    constexpr int kWavesHere = GROUPED ? W : 1;
    constexpr int kRows = kTile / (GROUPED ? W : 4);
    constexpr uint32_t kBuffers = DEEP_ONLY ? 1u : 2u;
    // (one area, two uses: a wave's own staging records in the ordinary mode; two rounds of four segments in the deep one)
    constexpr size_t kStageBytes = !GROUPED ? sizeof(StagedRecords)
                                   : DEEP_ONLY ? sizeof(DeepSegment) * W
                                   : (sizeof(DeepSegment) * 2 * kGroupWaves > sizeof(StagedRecords) * kGroupWaves ? sizeof(DeepSegment) * 2 * kGroupWaves
                                                                                                                : sizeof(StagedRecords) * kGroupWaves);
    __shared__ __attribute__((aligned(16))) unsigned char s_stage[kStageBytes];
    __shared__ unsigned long long s_exp[kWavesHere][32];   // exp_ref's table, a copy per wave
    __shared__ uint32_t s_count[2][kWavesHere];            // deep: survivors per segment of a round
    __shared__ uint32_t s_done[kWavesHere], s_done_at[kWavesHere];     // deep: strip finished, and on which record
    const int wave = GROUPED ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0, lane = (int)threadIdx.x & 63;     // (wave: a scalar)
    unsigned long long* const exp_tab = s_exp[wave];
    exp_table_init(exp_tab, lane);           // (written and read by that wave only)

    bool strips = false, prioritise = false;
    if (p.nonempty != nullptr) {
        const uint32_t ne = *p.nonempty;
        strips = GROUPED && (ne <= kStripTilesAny || (ne <= kStripTilesShort && (unsigned long long)p.num_rendered <= (unsigned long long)kStripMeanList * ne));
        prioritise = !strips && (unsigned long long)p.num_rendered >= (unsigned long long)kPriorityMeanList * ne;    // (see set_tile_priority)
    }
    const int base = p.base_workgroups;
    bool deep = false;
    int entry = (int)blockIdx.x;
    if constexpr (GROUPED) {
        const int deep_tiles = (DEEP_ONLY || strips || p.deep_all) ? base : (p.deep_count ? (int)min(*p.deep_count, (uint32_t)base) : 0);
        deep = (int)blockIdx.x < base;
        if (deep) {
            if (entry >= deep_tiles) return;
        } else {
            // (the four waves of a workgroup take entries 8 apart: entries e and e + 8 were one XCD's when an entry was a workgroup
            // of its own — the tiles of a 4 x 2 patch share their records in that XCD's L2, tile_of_workgroup)
            const int g = (int)blockIdx.x - base;
            entry = deep_tiles + 32 * (g >> 3) + 8 * wave + (g & 7);
            if (strips || p.deep_all || entry >= base) return;
        }
    }
    const uint32_t clock_begin = tile_clock();
    const int tile_local = tile_of_workgroup(p.history.order ? (int)p.history.order[entry] : entry, p.dims.grid_x, p.dims.row_end - p.dims.row_begin);
    if (tile_local < 0) return;
    const int tile = p.dims.row_begin * p.dims.grid_x + tile_local;
    const int tx = tile % p.dims.grid_x, ty = tile / p.dims.grid_x;
    const uint2 range = p.ranges[tile];
    const uint32_t total = range.y - range.x;          // unsigned wrap as in the reference

    TileFeed feed;
    feed.means2D = p.means2D; feed.colors = p.colors; feed.conic_opacity = p.conic_opacity;
    feed.dc_stride = p.dc_stride;
    feed.box = tile_box(tx, ty, p.dims.width, p.dims.height);
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
    if constexpr (GROUPED) if (deep) {
        constexpr uint32_t kRound = W * kWave;
        DeepSegment (*const segs)[W] = reinterpret_cast<DeepSegment (*)[W]>(s_stage);      // [round & 1][wave] (one area in the deep-only kernel)
        StripLanes s;
        strip_lanes_init(s, tx, ty, wave, lane, p.dims.width, p.dims.height, kRows);
        auto everybody_done = [&] {
            uint32_t all = 1u;
#pragma unroll
            for (int g = 0; g < W; ++g) all &= s_done[g];
            return all != 0u;
        };
        bool my_done = strip_lanes_all_done(s);            // (a strip below the image: finished from the start)
        if (lane == 0) { s_done[wave] = my_done ? 1u : 0u; s_done_at[wave] = 0u; }
        // (the barrier of a round waits for the wave's LDS traffic only — __syncthreads would also wait for the loads of the
        // rounds to come, which are in flight on purpose)
        auto round_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
        auto composite_round = [&](uint32_t half) {
#pragma nounroll
            for (int g = 0; g < W; ++g) {
                const uint32_t cnt = s_count[half][g];
                if (cnt == 0u) continue;
                const DeepSegment& seg = segs[half][g];
                const unsigned long long slots = __ballot((uint32_t)lane < cnt && ((seg.touch[lane] >> wave) & 1u) != 0u);
                uint32_t at = 0;
                if (slots != 0ull && composite_strip(s, seg, slots, feed.t_cutoff, exp_tab, &at)) {
                    my_done = true;
                    if (lane == 0) { s_done[wave] = 1u; s_done_at[wave] = at; }
                    break;
                }
            }
        };
        RecordBatch b0 = next_batch((uint32_t)(wave * kWave));
        fetch_records(b0, feed);
        RecordBatch b1 = next_batch(kRound + (uint32_t)(wave * kWave));
        if constexpr (kBuffers == 1u) {
            // one staging area: stage round r, barrier, composite it, barrier (b0: round r, its records there; b1: round r + 1, its ids)
            round_barrier();                             // (the flags above)
            for (uint32_t pos = 0; pos < total; pos += kRound) {
                fetch_records(b1, feed);
                const RecordBatch b2 = next_batch(pos + 2 * kRound + (uint32_t)(wave * kWave));
                const uint32_t kept = b0.valid ? stage_batch_deep<W>(feed, segs[0][wave], b0, p.dims.height) : 0u;
                if (lane == 0) s_count[0][wave] = kept;
                round_barrier();                         // the round's four segments are staged
                if (!my_done) composite_round(0u);
                round_barrier();                         // everybody is through with them; the flags of this round are visible
                if (everybody_done()) break;
                b0 = b1;
                b1 = b2;
            }
        } else {
            // Round r: stage r + 1 into the other half of the area (everybody left round r - 1, which read it, through the
            // barrier before), composite r, one barrier.
            fetch_records(b1, feed);
            RecordBatch b2 = next_batch(2 * kRound + (uint32_t)(wave * kWave));
            {
                const uint32_t kept = b0.valid ? stage_batch_deep<W>(feed, segs[0][wave], b0, p.dims.height) : 0u;
                if (lane == 0) s_count[0][wave] = kept;
            }
            round_barrier();
            for (uint32_t pos = 0, r = 0; pos < total; pos += kRound, ++r) {
                // (b1: round r + 1, its records on their way since the round before; b2: round r + 2, its ids)
                fetch_records(b2, feed);
                const RecordBatch b3 = next_batch(pos + 3 * kRound + (uint32_t)(wave * kWave));
                const uint32_t kept = b1.valid ? stage_batch_deep<W>(feed, segs[(r + 1u) & 1u][wave], b1, p.dims.height) : 0u;
                if (lane == 0) s_count[(r + 1u) & 1u][wave] = kept;
                if (!my_done) composite_round(r & 1u);
                round_barrier();                            // round r is composited, round r + 1 staged, the flags are visible
                if (everybody_done()) break;
                b1 = b2;
                b2 = b3;
            }
        }
        strip_lanes_write(s, p.dims.width, p.dims.height, p.background, p.final_t, p.n_contrib, p.out_color);
        if (wave == 0 && lane == 0) {
            if (p.staged_counter) {
                // (the loop was left through a barrier: the flags are final)
                const bool finished = everybody_done();
                uint32_t last = 0;
#pragma unroll
                for (int g = 0; g < W; ++g) last = max(last, s_done_at[g]);
                const unsigned long long batches = ((unsigned long long)last + (unsigned long long)kBatch - 1ull) / (unsigned long long)kBatch;
                atomicAdd(p.staged_counter, finished ? min((unsigned long long)total, batches * (unsigned long long)kBatch) : (unsigned long long)total);
            }
            if (p.history.ticks) p.history.ticks[tile] = ((tile_clock() - clock_begin) & ~kDeepFlag) | kDeepFlag;
        }
        return;
    }
    if constexpr (!DEEP_ONLY) {
    StagedRecords& mine = reinterpret_cast<StagedRecords*>(s_stage)[wave];
    TileLanes s;
    tile_lanes_init(s, tx, ty, lane, p.dims.width, p.dims.height, -1);
    unsigned long long staged = 0;
    bool all_done = tile_lanes_all_done(s);
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
            all_done = stage_and_composite(s, feed, mine, b0, staged, exp_tab);
            b0 = b1;
            b1 = b2;
            if (with_priority.value) set_tile_priority(total - min(total, pos));
        }
    };
    if (prioritise) walk(std::true_type{});
    else walk(std::false_type{});
    tile_lanes_write(s, p.dims.width, p.dims.height, p.background, p.final_t, p.n_contrib, p.out_color);
    if (p.staged_counter && lane == 0) atomicAdd(p.staged_counter, staged);
    if (p.history.ticks && lane == 0) p.history.ticks[tile] = (tile_clock() - clock_begin) & ~kDeepFlag;
    }
}

__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(5))) void blend_wave_kernel(const BlendParams p) { blend_wave_body<kBlendSingle>(p); }
__global__ __launch_bounds__(kGroupWaves * kWave) __attribute__((amdgpu_waves_per_eu(4))) void blend_group_kernel(const BlendParams p) { blend_wave_body<kBlendGrouped>(p); }
__global__ __launch_bounds__(kGroupWaves * kWave) __attribute__((amdgpu_waves_per_eu(8))) void blend_deep_kernel(const BlendParams p) { blend_wave_body<kBlendDeepOnly>(p); }
__global__ __launch_bounds__(8 * kWave) __attribute__((amdgpu_waves_per_eu(8))) void blend_deep8_kernel(const BlendParams p) { blend_wave_body<kBlendDeepOnly, 8>(p); }
__global__ __launch_bounds__(16 * kWave) __attribute__((amdgpu_waves_per_eu(4))) void blend_deep16_kernel(const BlendParams p) { blend_wave_body<kBlendDeepOnly, 16>(p); }


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
// Which of the slow tiles get four waves (blend_wave_kernel, DEEP tiles). A tile that was composited by four waves left its
// time flagged (kDeepFlag): what it would take one wave is estimated as gain x that time, and every decision below is
// taken on those estimates, so that a tile does not leave the class because the class made it faster. With the deep tiles
// taking 1 / gain of their time, the launch lasts about max(longest / gain, longest ordinary tile, everybody's time spread
// over the chip's wave slots): a tile is DEEP when its estimate lies beyond max(frac x longest estimate, that spread, floor)
// — compared as quantised classes, so that the deep tiles are exactly the order's leading entries.
struct DeepRule {
    uint32_t gain_x16;        // one wave's time over four waves' (x 16)
    uint32_t frac_x16;        // of the longest tile's estimate (x 16)
    uint32_t floor_ticks;     // never below this (10 ns units): a short tile gains nothing from three barriers a round
    uint32_t wave_slots;      // wave slots of the chip the frame's tile time is spread over (DeviceShape::blend_slots)
    unsigned long long light_ticks;   // DeviceShape::light_frame_ticks
};
__device__ __forceinline__ uint32_t tile_estimate(uint32_t recorded, uint32_t gain_x16) {
    const uint32_t t = recorded & ~kDeepFlag;
    if (gain_x16 == 0u) return t;                // (no deep tiles wanted: GSR_DEEP=0)
    return (recorded & kDeepFlag) ? (uint32_t)min(0x7FFFFFFFull, ((unsigned long long)t * gain_x16) >> 4) : t;
}
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint32_t* __restrict__ ticks, const uint32_t* __restrict__ ticks_before,
                                                          uint32_t* __restrict__ order,
                                                          int workgroups, int padded, int grid_x, int row_begin, int rows,
                                                          uint32_t* __restrict__ stats, uint32_t* __restrict__ deep_count, const DeepRule rule) {
    extern __shared__ uint32_t s_key[];
    __shared__ unsigned long long s_sum, s_common, s_before, s_est;
    __shared__ uint32_t s_cnt, s_max, s_max_est, s_deep;
    if (threadIdx.x == 0) { s_sum = 0; s_cnt = 0; s_max = 0; s_common = 0; s_before = 0; s_est = 0; s_max_est = 0; s_deep = 0; }
    __syncthreads();
    unsigned long long sum = 0, sum_before = 0, sum_est = 0;
    uint32_t cnt = 0, mx = 0, mx_est = 0;
    for (int i = threadIdx.x; i < workgroups; i += 1024) {
        const int tile_local = tile_of_workgroup(i, grid_x, rows);
        if (tile_local >= 0) {
            const uint32_t rec = ticks[row_begin * grid_x + tile_local];
            const uint32_t t = rec & ~kDeepFlag, est = tile_estimate(rec, rule.gain_x16);
            sum += t; mx = max(mx, t); ++cnt;
            sum_est += est; mx_est = max(mx_est, est);
            sum_before += tile_estimate(ticks_before[row_begin * grid_x + tile_local], rule.gain_x16);
        }
    }
    atomicAdd(&s_sum, sum);
    atomicAdd(&s_cnt, cnt);
    atomicMax(&s_max, mx);
    atomicAdd(&s_before, sum_before);
    atomicAdd(&s_est, sum_est);
    atomicMax(&s_max_est, mx_est);
    __syncthreads();
    // shares in units of 2^-20 of the frame's total (a tile's time is below 2^32, times 2^20 fits 64 bits)
    unsigned long long common = 0;
    const unsigned long long total = max(s_est, 1ull), total_before = max(s_before, 1ull);
    for (int i = threadIdx.x; i < workgroups; i += 1024) {
        const int tile_local = tile_of_workgroup(i, grid_x, rows);
        if (tile_local >= 0) {
            const unsigned long long a = ((unsigned long long)tile_estimate(ticks[row_begin * grid_x + tile_local], rule.gain_x16) << 20) / total;
            const unsigned long long b = ((unsigned long long)tile_estimate(ticks_before[row_begin * grid_x + tile_local], rule.gain_x16) << 20) / total_before;
            common += min(a, b);
        }
    }
    atomicAdd(&s_common, common);
    __syncthreads();
    const uint32_t similarity = s_before == 0ull ? 1000u : (uint32_t)min(1000ull, (1000ull * s_common) >> 20);
    const bool dropped = similarity < kMinSimilarity;
    // (a LIGHT frame — less than 250 us of tile time per wave slot — is sorted whole: its lists are short, it lasts as long as
    // its longest tiles whatever the neighbours do; see DeviceShape::light_frame_ticks)
    const unsigned long long slow_from = s_est < rule.light_ticks ? 0ull : 2ull * s_est / max(s_cnt, 1u);
    const unsigned long long deep_from = max(max(((unsigned long long)s_max_est * rule.frac_x16) >> 4, s_est / max(rule.wave_slots, 1u)),
                                             max((unsigned long long)rule.floor_ticks, slow_from));
    const uint32_t deep_class = quantise_ticks((uint32_t)min(deep_from, 0x7FFFFFFFull));
    // what the host decides on whether the next calls need an order at all: {fresh, longest tile, mean, similarity, dropped}
    if (stats && threadIdx.x == 0) {
        // (longest and mean in ONE-WAVE terms — a deep tile's time times the gain —: what the host's estimates of a blend's
        // length are calibrated in, whichever way the frame before was composited)
        stats[1] = s_max_est; stats[2] = (uint32_t)(s_est / max(s_cnt, 1u)); stats[3] = similarity; stats[4] = dropped ? 1u : 0u;
        __threadfence_system();
        stats[0] = 1u;
    }
    if (dropped) {                                   // (the patch order itself)
        for (int i = threadIdx.x; i < workgroups; i += 1024) order[i] = (uint32_t)i;
        if (deep_count && threadIdx.x == 0) *deep_count = 0u;
        return;
    }
    uint32_t deep = 0;
    for (int i = threadIdx.x; i < padded; i += 1024) {
        uint32_t key = 0xFFF00000u | (uint32_t)i;
        if (i < workgroups) {
            const int tile_local = tile_of_workgroup(i, grid_x, rows);
            if (tile_local >= 0) {
                const uint32_t t = tile_estimate(ticks[row_begin * grid_x + tile_local], rule.gain_x16);
                const uint32_t q = quantise_ticks(t);
                key = ((t > slow_from ? 511u - q : 512u) << 20) | (uint32_t)i;
                if (t > slow_from && q > deep_class) ++deep;
            }
        }
        s_key[i] = key;
    }
    atomicAdd(&s_deep, deep);
    __syncthreads();
    if (deep_count && threadIdx.x == 0) *deep_count = rule.gain_x16 != 0u ? s_deep : 0u;
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
                 uint32_t num_rendered, const uint32_t* tile_order, uint32_t* tile_ticks, bool colors_are_shs, const uint32_t* deep_count,
                 bool deep_all, int deep_waves) {
    BlendParams p;
    p.deep_count = tile_order ? deep_count : nullptr;      // (the deep tiles are the order's leading entries)
    p.deep_all = deep_all ? 1 : 0;
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
    // Deep tiles are possible when the host asks for them (deep_all), when the history may name some, or when the frame may
    // turn out to have few tiles with a list (the kernel's `strips`: at most 1 536 tiles, or lists of 1 024 entries on 4 096):
    // workgroups [0, base) then take a deep tile each (or nothing), and behind them come the ordinary tiles, four to a
    // workgroup. Every other frame is launched a wave per workgroup, as ever.
    const bool few_tiles = p.nonempty != nullptr && ((uint32_t)p.num_tiles <= kStripTilesAny ||
                                                     (unsigned long long)num_rendered <= (unsigned long long)kStripMeanList * kStripTilesShort);
    if (p.deep_all && deep_waves == 16)
        hipLaunchKernelGGL(blend_deep16_kernel, dim3((unsigned)p.base_workgroups), dim3(16 * kWave), 0, stream, p);
    else if (p.deep_all && deep_waves == 8)
        hipLaunchKernelGGL(blend_deep8_kernel, dim3((unsigned)p.base_workgroups), dim3(8 * kWave), 0, stream, p);
    else if (p.deep_all)
        hipLaunchKernelGGL(blend_deep_kernel, dim3((unsigned)p.base_workgroups), dim3(kGroupWaves * kWave), 0, stream, p);
    else if (p.deep_count || few_tiles)
        hipLaunchKernelGGL(blend_group_kernel, dim3((unsigned)(p.base_workgroups + (p.base_workgroups + 31) / 32 * 8)),
                           dim3(kGroupWaves * kWave), 0, stream, p);
    else
        hipLaunchKernelGGL(blend_wave_kernel, dim3((unsigned)p.base_workgroups), dim3(kWave), 0, stream, p);
    GSR_LAUNCH_CHECK("blend_wave_kernel");
    return GSR_OK;
}

// (see TileOrder, blend_core.hpp) `order` receives patch_workgroups(...) entries; asynchronous on stream
int tile_order_workgroups(const FrameDims& d) { return patch_workgroups(d.grid_x, d.row_end - d.row_begin); }
int launch_tile_order(const FrameDims& d, const uint32_t* ticks, const uint32_t* ticks_before, uint32_t* order, uint32_t* stats,
                      hipStream_t stream, bool* sorted, uint32_t* deep_count, const DeviceShape& shape) {
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
    // (GSR_DEEP = "gain_x16,frac_x16,floor_ticks" in the environment, read once: A/B runs and the tests; gain 0: no deep tiles)
    static const DeepRule env_rule = [] {
        DeepRule r{kDeepGainX16, kDeepFracX16, kDeepFloorTicks, 0u, 0ull};
        if (const char* e = getenv("GSR_DEEP")) {
            unsigned g = r.gain_x16, f = r.frac_x16, fl = r.floor_ticks;
            const int got = sscanf(e, "%u,%u,%u", &g, &f, &fl);
            if (got >= 1) r.gain_x16 = g;
            if (got >= 2) r.frac_x16 = f;
            if (got >= 3) r.floor_ticks = fl;
        }
        return r;
    }();
    DeepRule rule = env_rule;
    rule.wave_slots = shape.blend_slots;
    rule.light_ticks = shape.light_frame_ticks;
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), (size_t)padded * sizeof(uint32_t), stream, ticks, ticks_before, order,
                       workgroups, padded, d.grid_x, d.row_begin, d.row_end - d.row_begin, stats, deep_count, rule);
    GSR_LAUNCH_CHECK("tile_order_kernel");
    *sorted = true;
    return GSR_OK;
}

}  // namespace gsr
