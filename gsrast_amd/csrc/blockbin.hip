// Block binning: the sorted (tile | depth, index) lists written directly, no pass over R keys.
//
// Reference semantics: apps/gsrast/gscuda/GSCuda.cu:422-475 (duplicateWithKeys) emits one pair per
// (Gaussian, covered tile) and :794-797 sorts all R of them by (tile, depth). The sorted list is,
// tile after tile, the Gaussians covering that tile in (depth, index) order. With the Gaussians
// already in that order (the depth half of the key is sorted once per Gaussian, before
// duplication) the list of a tile is a FILTER of the depth-ordered Gaussian list, so it can be
// written front to back by whoever owns the tile. Owners are wavefronts, one lane per tile of an
// 8 x 8 tile block:
//   coarse_count / blockscan_* / coarse_emit : the depth-ordered list is split, stably, into one
//       list per tile block (entries = rectangle, depth bits, index; 12 B, E <= R of them); a
//       chunk's entries leave in (block, rank) order, one per lane: whole lines per block
//   blend_blocks : ... on frames of large splats (48 or more instances per visible Gaussian, or
//       when the sorted lists are not written); sparser frames blend from the sorted lists (blend.hip)
//   unit_masks   : a unit = 2048 consecutive entries of one block list; its coverage bit masks
//       (entries x tile columns / rows, transposed with v_writelane) and, from their bit counts,
//       the keys per (unit, tile)
//   block_prefix / tile_start : prefix of those counts down the units of a block, then over the
//       tiles in tile order = where every (unit, tile) run starts in the sorted list — and, per
//       tile, its range [start, end) (identifyTileRanges, GSCuda.cu:504-538)
//   block_emit   : one wavefront per unit: per tile, (column mask & row mask) says which entries of
//       each batch of 64 cover it; they are compacted by v_mbcnt rank through a small LDS run
//       buffer and stored as dense, line-aligned groups of 128 keys — every (unit, tile) run is
//       written front to back by one wave.
//   blend_blocks : the per-tile blend (GSCuda.cu:543-677) fed from the same block lists and masks,
//       so it does not wait for — and can run beside — the emission.
// The result is bit-identical to the stable 64-bit sort (same lists, same order inside a tile).
// R-sized traffic: 12 R bytes written once (the reference's emit + 6-pass sort moves > 150 R).
#include <stdlib.h>

#include <map>

#include "blend_core.hpp"
#include "blockbin.hpp"

namespace gsr {
namespace {

constexpr int kCoarse = 1024;            // Gaussians per workgroup of the coarse passes ...
constexpr int kCoarseSmall = 512;        // ... with more than 256 blocks (4K): half the LDS mask table, two workgroups per CU
constexpr int kScanRows = 64;            // table rows per workgroup of the block scan

inline size_t align128(size_t v) { return (v + 127) / 128 * 128; }

// ---- coarse pass 1: entries per (chunk of 1024 depth-consecutive Gaussians, block) ------------
// sorted_rect: the visible Gaussians' packed rectangles in depth order (they travel through the depth sort with the indices)
template <int CHUNK>
__global__ __launch_bounds__(CHUNK) void coarse_count_kernel(int n, const uint32_t* __restrict__ sorted_rect, int nbx, int nbp,
                                                               uint32_t* __restrict__ table) {
    __shared__ uint32_t s_cnt[kMaxBlocks];
    if ((int)threadIdx.x < nbp) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int r = blockIdx.x * CHUNK + threadIdx.x;
    const uint32_t packed = (r < n) ? sorted_rect[r] : 0u;
    if (packed) {
        const uint32_t x0 = packed & 0xFFu, w = (packed >> 8) & 0xFFu, y0 = (packed >> 16) & 0xFFu, h = packed >> 24;
        const uint32_t bx0 = x0 / kBW, bx1 = (x0 + w - 1) / kBW, by0 = y0 / kBH, by1 = (y0 + h - 1) / kBH;
        for (uint32_t by = by0; by <= by1; ++by)
            for (uint32_t bx = bx0; bx <= bx1; ++bx) atomicAdd(&s_cnt[by * nbx + bx], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x < nbp) table[(size_t)blockIdx.x * nbp + threadIdx.x] = s_cnt[threadIdx.x];
}

// ---- exclusive prefix of table[row][b] down the rows, for every block column b ----------------
__global__ __launch_bounds__(kMaxBlocks) void blockscan_reduce_kernel(const uint32_t* __restrict__ table, uint32_t rows, int nbp,
                                                                      uint32_t* __restrict__ partial) {
    const uint32_t r0 = blockIdx.x * kScanRows, r1 = min(rows, r0 + kScanRows);
    uint32_t s = 0;
#pragma unroll 8
    for (uint32_t r = r0; r < r1; ++r) s += table[(size_t)r * nbp + threadIdx.x];
    partial[(size_t)blockIdx.x * nbp + threadIdx.x] = s;
}

// block-wide exclusive scan of one value per thread (blockDim.x <= 512, multiple of 64)
__device__ __forceinline__ uint32_t block_exclusive(uint32_t v, uint32_t* s_ws, uint32_t* total) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nw = blockDim.x / kWave;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += o;
    }
    __syncthreads();
    if (lane == kWave - 1) s_ws[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int w = 0; w < nw; ++w) {
        if (w < wave) base += s_ws[w];
        tot += s_ws[w];
    }
    *total = tot;
    return base + incl - v;
}

// One workgroup: per block column the exclusive prefix of the row-group partials; then, across the
// blocks, where each block list starts and which units it is cut into.
__global__ __launch_bounds__(kMaxBlocks) void blockscan_partials_kernel(uint32_t* __restrict__ partial, uint32_t groups, int nb,
                                                                        BlockMeta meta) {
    __shared__ uint32_t s_ws[kMaxBlocks / kWave];
    const int nbp = meta.nbp;
    uint32_t running = 0;
    // (32 rows' loads are issued before the first of their stores: the compiler cannot know that a store does not touch the next
    // row, and a row at a time is a round trip each — 47 of them on the bench frame)
    for (uint32_t g0 = 0; g0 < groups; g0 += 32u) {
        uint32_t v[32];
#pragma unroll
        for (uint32_t k = 0; k < 32u; ++k) v[k] = (g0 + k < groups) ? partial[(size_t)(g0 + k) * nbp + threadIdx.x] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 32u; ++k) {
            if (g0 + k < groups) partial[(size_t)(g0 + k) * nbp + threadIdx.x] = running;
            running += v[k];
        }
    }
    const uint32_t len = ((int)threadIdx.x < nb) ? running : 0u;
    uint32_t total_entries, total_units;
    const uint32_t ls = block_exclusive(len, s_ws, &total_entries);
    const uint32_t us = block_exclusive((len + kUnit - 1) / kUnit, s_ws, &total_units);
    meta.list_start()[threadIdx.x] = ls;
    meta.unit_start()[threadIdx.x] = us;
    meta.walked()[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        meta.list_start()[nbp] = total_entries;
        meta.unit_start()[nbp] = total_units;
        meta.tickets()[0] = 0;
        meta.tickets()[1] = 0;
    }
}

__global__ __launch_bounds__(kMaxBlocks) void blockscan_apply_kernel(uint32_t* __restrict__ table, uint32_t rows, int nbp,
                                                                     const uint32_t* __restrict__ partial,
                                                                     const uint32_t* __restrict__ list_start) {
    const uint32_t r0 = blockIdx.x * kScanRows, r1 = min(rows, r0 + kScanRows);
    uint32_t running = partial[(size_t)blockIdx.x * nbp + threadIdx.x] + list_start[threadIdx.x];
    for (uint32_t b0 = r0; b0 < r1; b0 += 32u) {          // (loads of 32 rows before their stores, as above)
        uint32_t v[32];
#pragma unroll
        for (uint32_t k = 0; k < 32u; ++k) v[k] = (b0 + k < r1) ? table[(size_t)(b0 + k) * nbp + threadIdx.x] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 32u; ++k) {
            if (b0 + k < r1) table[(size_t)(b0 + k) * nbp + threadIdx.x] = running;
            running += v[k];
        }
    }
}

// ---- coarse pass 2: write the block lists ------------------------------------------------------
// Order inside a chunk: per block a 1024-bit mask of the chunk's Gaussians touching it; the rank of Gaussian g in
// block b is the number of set bits below g. The chunk's entries are then written in (block, rank) order, one entry per
// lane: consecutive lanes write consecutive entries of one block's list, so a wave's stores are a few whole lines.
// (One lane per Gaussian looping over its blocks — the version before — sent every entry to a different list: two
// store requests per entry, 24 M per frame, which is what the kernel's 0.12 ms were made of.)
// The t-th set bit of m (t < popc(m)).
__device__ __forceinline__ uint32_t nth_set_bit(uint32_t m, uint32_t t) {
    uint32_t at = 0;
#pragma unroll
    for (uint32_t half = 16; half >= 1; half >>= 1) {
        const uint32_t c = (uint32_t)__popc(m & ((1u << half) - 1u));
        const bool up = t >= c;
        t -= up ? c : 0u;
        at += up ? half : 0u;
        m = up ? (m >> half) : m;
    }
    return at;
}

template <int CHUNK>
__global__ __launch_bounds__(CHUNK) void coarse_emit_kernel(int n, const uint32_t* __restrict__ sorted_depth,
                                                              const uint32_t* __restrict__ sorted_idx,
                                                              const uint32_t* __restrict__ rect_packed,
                                                              const uint32_t* __restrict__ table, int nbx, int nb, int nbp,
                                                              uint64_t* __restrict__ ent_rd, uint32_t* __restrict__ ent_idx) {
    extern __shared__ uint32_t s_dyn[];
    constexpr int W = CHUNK / 32;                // mask words per block
    constexpr int WP = W + 1;                    // (rows one word apart in the banks)
    uint32_t* s_mask = s_dyn;                    // [nb][WP]
    uint32_t* s_pre = s_mask + (size_t)nb * WP;  // [nb][WP] set bits in the words below; [W] = the block's total
    uint32_t* s_start = s_pre + (size_t)nb * WP; // [kMaxBlocks] where the block's entries start among the chunk's (~0 past nb)
    uint32_t* s_row = s_start + kMaxBlocks;      // [nb] where the chunk's entries of the block start in the block lists
    uint32_t* s_rect = s_row + nbp;              // [CHUNK] what an entry carries
    uint32_t* s_depth = s_rect + CHUNK;
    uint32_t* s_idx = s_depth + CHUNK;
    __shared__ uint32_t s_ws[CHUNK / kWave];
    const int r = blockIdx.x * CHUNK + threadIdx.x;
    // (every load of the chunk is issued before the first wait: one round trip to memory per workgroup, not two)
    const uint32_t rect = (r < n) ? rect_packed[r] : 0u;
    const uint32_t depth_in = (r < n) ? sorted_depth[r] : 0u, idx_in = (r < n) ? sorted_idx[r] : 0u;
    const uint32_t row_in = ((int)threadIdx.x < nbp) ? table[(size_t)blockIdx.x * nbp + threadIdx.x] : 0u;
    s_rect[threadIdx.x] = rect;
    s_depth[threadIdx.x] = depth_in;
    s_idx[threadIdx.x] = idx_in;
    if ((int)threadIdx.x < nbp) s_row[threadIdx.x] = row_in;
    for (int i = threadIdx.x; i < nb * WP; i += CHUNK) s_mask[i] = 0;
    __syncthreads();
    const uint32_t x0 = rect & 0xFFu, w = (rect >> 8) & 0xFFu, y0 = (rect >> 16) & 0xFFu, h = rect >> 24;
    uint32_t bx0 = 0, bx1 = 0, by0 = 1, by1 = 0;
    if (rect) { bx0 = x0 / kBW; bx1 = (x0 + w - 1) / kBW; by0 = y0 / kBH; by1 = (y0 + h - 1) / kBH; }
    const uint32_t word = threadIdx.x >> 5, bit = 1u << (threadIdx.x & 31);
    for (uint32_t by = by0; by <= by1; ++by)
        for (uint32_t bx = bx0; bx <= bx1; ++bx) atomicOr(&s_mask[(by * nbx + bx) * WP + word], bit);
    __syncthreads();
    // W consecutive lanes scan the W words of one block
    for (int i = threadIdx.x; i < nb * W; i += CHUNK) {
        const int at = (i / W) * WP + (i & (W - 1));
        const uint32_t c = (uint32_t)__popc(s_mask[at]);
        // (inclusive prefix over the W lanes that share a block: DPP row shifts — W = 32: + the row broadcast — instead of
        // five LDS permutes)
        uint32_t incl;
        if constexpr (W == 32) incl = prefix32_inclusive(c);
        else {
            static_assert(W == 16 || W == 32, "a block's mask words are one or two DPP rows");
            incl = c;
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, false);
        }
        s_pre[at] = incl - c;
        if ((i & (W - 1)) == W - 1) s_pre[(i / W) * WP + W] = incl;
    }
    __syncthreads();
    // exclusive prefix of the blocks' totals (blocks past nb: ~0, so that the search below needs no bounds)
    uint32_t chunk_total;
    {
        const uint32_t a = ((int)threadIdx.x < nb) ? s_pre[threadIdx.x * WP + W] : 0u;
        const uint32_t ex = block_exclusive(a, s_ws, &chunk_total);
        if (threadIdx.x < (uint32_t)kMaxBlocks) s_start[threadIdx.x] = ((int)threadIdx.x < nb) ? ex : 0xFFFFFFFFu;
    }
    __syncthreads();
    const uint32_t top = nb > 256 ? 256u : 128u;
    for (uint32_t j = threadIdx.x; j < chunk_total; j += CHUNK) {
        // the last block that starts at or before j (blocks without an entry share their successor's start and are skipped)
        uint32_t b = 0;
        for (uint32_t step = top; step >= 1; step >>= 1)
            if (s_start[b + step] <= j) b += step;
        const uint32_t t = j - s_start[b];
        const uint32_t* pre = s_pre + b * WP;
        uint32_t wd = 0;
#pragma unroll
        for (uint32_t step = W / 2; step >= 1; step >>= 1)
            if (pre[wd + step] <= t) wd += step;
        const uint32_t g = wd * 32u + nth_set_bit(s_mask[b * WP + wd], t - pre[wd]);
        const uint32_t pos = s_row[b] + t;
        ent_rd[pos] = (uint64_t)s_rect[g] | ((uint64_t)s_depth[g] << 32);
        ent_idx[pos] = s_idx[g];
    }
}

// ---- units ---------------------------------------------------------------------------------------
struct UnitInfo { uint32_t bx, by, e0, e1; };

// Which block does unit u belong to: the number of blocks whose first unit is <= u, minus one.
// (blocks without entries have no units; the last block with unit_start <= u is the owner)
__device__ __forceinline__ UnitInfo locate_unit(uint32_t u, const BlockMeta& meta, int nb, int nbx) {
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t* us = meta.unit_start();
    uint32_t below = 0;
    for (int k = 0; k < meta.nbp; k += kWave) {
        const int b = k + lane;
        below += (uint32_t)__popcll(__ballot(b < nb && us[b] <= u));
    }
    const uint32_t b = below - 1u;
    UnitInfo ui;
    ui.bx = b % (uint32_t)nbx;
    ui.by = b / (uint32_t)nbx;
    ui.e0 = meta.list_start()[b] + (u - us[b]) * kUnit;
    ui.e1 = min(meta.list_start()[b + 1], ui.e0 + kUnit);
    return ui;
}

// Coverage masks of one unit, transposed: for each of the 8 tile columns and 8 tile rows of the block (k = column 0..7 |
// 8 + row) and each batch w of 64 entries, the 64-bit mask of the batch's entries that cover it. Coverage of tile (c, r)
// by batch w is then m[c][w] & m[8 + r][w]: a rectangle is a column range x a row range.
// Lane = entry on the way in: every entry's 16 coverage bits go to LDS as a half-word. Lane = (batch w, half h) on the way
// out: it reads the 32 half-words of its half batch as 16 dwords and transposes them in its registers — a packed pair of
// 16 x 16 bit-matrix transposes, four rounds of eight masked swaps — into the 16 half masks it owns. (Sixteen ballots per
// batch, each filed with two v_writelane, were 2 000 vector instructions per unit; this is 200, on all 64 lanes.)
// The entries of a half batch are stored so that dword i pairs entry i (low half-word) with entry i + 16: then dword k
// after the transpose IS the 32-bit mask of column / row k, entry i at bit i.
__device__ __forceinline__ void transpose_16x16_pairs(uint32_t (&d)[16]) {
#pragma unroll
    for (int round = 0; round < 4; ++round) {
        const int sh = 8 >> round;
        const uint32_t m = round == 0 ? 0x00FF00FFu : (round == 1 ? 0x0F0F0F0Fu : (round == 2 ? 0x33333333u : 0x55555555u));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i & sh) continue;
            const uint32_t t = ((d[i] >> sh) ^ d[i + sh]) & m;
            d[i + sh] ^= t;
            d[i] ^= t << sh;
        }
    }
}

// Sums of 32 per-lane values over the 64 lanes, all at once: lane l returns the total of x[l >> 1]. Every fold halves the
// registers while it halves the lanes a value lives in: v_permlane32_swap / v_permlane16_swap (gfx950) for the lane bits 5
// and 4, then a select and one DPP add per pair of registers (row_ror:8, row_half_mirror, two quad permutations). 70
// instructions where 32 separate prefix sums with their read-backs took 400. (No carries between packed fields expected.)
__device__ __forceinline__ uint32_t wave_sums_of_32(uint32_t (&x)[32]) {
    const int lane = threadIdx.x & (kWave - 1);
#define GSR_DPP_U(v, ctrl) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xf, 0xf, false))
    uint32_t y[16], z[8], u[4], v[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {               // lanes 0-31: x[i] (lanes l and l + 32 added); lanes 32-63: x[i + 16]
        const auto r = __builtin_amdgcn_permlane32_swap(x[i], x[i + 16], false, false);
        y[i] = r[0] + r[1];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {                // rows of 16 lanes: x[i], x[i + 8], x[i + 16], x[i + 24]
        const auto r = __builtin_amdgcn_permlane16_swap(y[i], y[i + 8], false, false);
        z[i] = r[0] + r[1];
    }
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = (b3 ? z[i + 4] : z[i]) + GSR_DPP_U(b3 ? z[i] : z[i + 4], 0x128);     // row_ror:8
#pragma unroll
    for (int i = 0; i < 2; ++i) v[i] = (b2 ? u[i + 2] : u[i]) + GSR_DPP_U(b2 ? u[i] : u[i + 2], 0x141);     // row_half_mirror
    uint32_t w = (b1 ? v[1] : v[0]) + GSR_DPP_U(b1 ? v[0] : v[1], 0x4E);                                    // quad_perm [2,3,0,1]
    w += GSR_DPP_U(w, 0xB1);                                                                                 // quad_perm [1,0,3,2]
#undef GSR_DPP_U
    return w;
}

// Per unit: the transposed coverage masks (kept for the emission kernel, 4 KB per unit) and, from
// their bit counts, the keys per tile of the block.
__global__ __launch_bounds__(256) void unit_masks_kernel(BlockMeta meta, int nb, int nbx, const uint64_t* __restrict__ ent_rd,
                                                         uint2* __restrict__ unit_masks, uint32_t* __restrict__ cnt) {
    __shared__ uint16_t s_bits[256 / kWave][kUnit];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint32_t total = meta.unit_start()[meta.nbp];
    const uint32_t* ent_rd32 = reinterpret_cast<const uint32_t*>(ent_rd);
    // where this lane's entry of a batch goes (see above), and which half batch it owns on the way out
    const uint32_t q = (uint32_t)lane & 31u;
    uint16_t* const put = s_bits[wave] + (((uint32_t)lane & 32u) | ((q & 15u) << 1) | (q >> 4));
    const uint4* const take = reinterpret_cast<const uint4*>(s_bits[wave] + 64u * q + ((uint32_t)lane & 32u));
    // units cost the same here (<= 2048 entries each): dealt round-robin, no work queue (thousands of
    // waves taking tickets from one counter serialise on it for longer than the kernel's own work)
    const uint32_t nwaves = gridDim.x * (blockDim.x / kWave);
    for (uint32_t u = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; u < total; u += nwaves) {
        const UnitInfo ui = locate_unit(u, meta, nb, nbx);
        const uint32_t e0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ui.e0);
        const uint32_t e1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ui.e1);
        const uint32_t bx0 = ui.bx * kBW, by0 = ui.by * kBH;
        // all of the unit's rectangles are fetched up front: one memory round trip, not one per batch
        uint32_t rects[kBatches];
#pragma unroll
        for (int w = 0; w < kBatches; ++w) {
            const uint32_t i = e0 + (uint32_t)w * kWave + (uint32_t)lane;
            rects[w] = (i < e1) ? ent_rd32[2 * (size_t)i] : 0u;
        }
#pragma unroll
        for (int w = 0; w < kBatches; ++w) {
            const uint32_t rect = rects[w];
            const uint32_t x0 = rect & 0xFFu, rw = (rect >> 8) & 0xFFu, y0 = (rect >> 16) & 0xFFu, rh = rect >> 24;
            const uint32_t cx0 = max(x0, bx0) - bx0, cx1 = min(x0 + rw, bx0 + kBW) - bx0;
            const uint32_t cy0 = max(y0, by0) - by0, cy1 = min(y0 + rh, by0 + kBH) - by0;
            // bits 0..7: tile columns covered, bits 8..15: tile rows covered (0 for lanes past the end)
            const uint32_t bits = (((1u << (cx1 - cx0)) - 1u) << cx0) | (((1u << (cy1 - cy0)) - 1u) << (cy0 + 8u));
            put[w * kWave] = (uint16_t)(rect ? bits : 0u);
        }
        // (wave-private LDS: the writes above and the reads below are ordered inside the wave)
        uint32_t d[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 v = take[i];
            d[4 * i] = v.x; d[4 * i + 1] = v.y; d[4 * i + 2] = v.z; d[4 * i + 3] = v.w;
        }
        transpose_16x16_pairs(d);
        // lane (w, h) holds half h of the 64-bit mask of batch w: the uint2 of (k, w) is written by two lanes
        uint32_t* dst = reinterpret_cast<uint32_t*>(unit_masks + (size_t)u * 16 * kBatches) + 2u * q + ((uint32_t)lane >> 5);
#pragma unroll
        for (int k = 0; k < 16; ++k) dst[k * 2 * kBatches] = d[k];
        // keys per tile = bits of (column mask & row mask) summed over the half batches (lanes); the two tiles 2 p, 2 p + 1
        // of a tile row share a sum, 16 bits each (<= 2048 keys per tile and unit)
        uint32_t x[32];
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            const int c0 = (2 * p) & 7, r = 8 + ((2 * p) >> 3);
            x[p] = (uint32_t)__popc(d[c0] & d[r]) | ((uint32_t)__popc(d[c0 + 1] & d[r]) << 16);
        }
        const uint32_t pair = wave_sums_of_32(x);      // lane l: tiles 2 (l >> 1), 2 (l >> 1) + 1 = l & ~1, l | 1
        cnt[(size_t)u * 64 + lane] = (lane & 1) ? (pair >> 16) : (pair & 0xFFFFu);
    }
}

// Per block: exclusive prefix of cnt[unit][tile] down the block's units; totals = keys per tile.
__global__ __launch_bounds__(256) void block_prefix_kernel(BlockMeta meta, int nbx, int gx, int gy, uint32_t* __restrict__ cnt,
                                                           uint32_t* __restrict__ tile_count) {
    __shared__ uint32_t s_sum[4][64];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint32_t b = blockIdx.x;
    const uint32_t u0 = meta.unit_start()[b], u1 = meta.unit_start()[b + 1];
    const uint32_t per = (u1 - u0 + 3) / 4;
    const uint32_t a = min(u1, u0 + (uint32_t)wave * per), z = min(u1, a + per);
    uint32_t s = 0;
#pragma unroll 8
    for (uint32_t u = a; u < z; ++u) s += cnt[(size_t)u * 64 + lane];
    s_sum[wave][lane] = s;
    __syncthreads();
    uint32_t running = 0, total = 0;
    for (int w = 0; w < 4; ++w) {
        if (w < wave) running += s_sum[w][lane];
        total += s_sum[w][lane];
    }
    for (uint32_t u0b = a; u0b < z; u0b += 32u) {          // (loads of 32 units before their stores: cnt is read and written)
        uint32_t v[32];
#pragma unroll
        for (uint32_t k = 0; k < 32u; ++k) v[k] = (u0b + k < z) ? cnt[(size_t)(u0b + k) * 64 + lane] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 32u; ++k) {
            if (u0b + k < z) cnt[(size_t)(u0b + k) * 64 + lane] = running;
            running += v[k];
        }
    }
    const uint32_t tx = (b % (uint32_t)nbx) * kBW + (uint32_t)(lane & 7), ty = (b / (uint32_t)nbx) * kBH + (uint32_t)(lane >> 3);
    if (wave == 0 && tx < (uint32_t)gx && ty < (uint32_t)gy) tile_count[ty * gx + tx] = total;
}

// Exclusive prefix over the tiles in tile order (one workgroup; T <= 65 025) = where every tile's list
// starts, which is also the tile's range: [start, start + count) as identifyTileRanges leaves it
// (GSCuda.cu:504-538) — (0, 0) for tiles without a key, and for the lone tile of an R == 1 frame unless
// `close_single` (the upstream behaviour) is asked for.
__global__ __launch_bounds__(1024) void tile_start_kernel(const uint32_t* __restrict__ tile_count, uint32_t tiles,
                                                          uint32_t* __restrict__ tile_start, uint2* __restrict__ ranges,
                                                          uint32_t r_total, bool close_single, uint32_t* __restrict__ nonempty,
                                                          uint32_t* __restrict__ skipped_stamp) {
    __shared__ uint32_t s_ws[16];
    // GSR_FLAG_NO_SORTED_LISTS: the sorted values stay unwritten; their first word says so (include/gsrast_amd.h)
    if (threadIdx.x == 0 && skipped_stamp) *skipped_stamp = GSR_LISTS_SKIPPED_STAMP;
    const uint32_t per = (tiles + 1023) / 1024;
    const uint32_t a = min(tiles, threadIdx.x * per), z = min(tiles, a + per);
    uint32_t s = 0;
#pragma unroll 8
    for (uint32_t t = a; t < z; ++t) s += tile_count[t];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t incl = s;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += o;
    }
    if (lane == kWave - 1) s_ws[wave] = incl;
    __syncthreads();
    uint32_t running = incl - s;
    for (int w = 0; w < wave; ++w) running += s_ws[w];
    const bool closed = r_total > 1u || close_single;
    int with_list = 0;
#pragma unroll 8
    for (uint32_t t = a; t < z; ++t) {
        const uint32_t c = tile_count[t];
        tile_start[t] = running;
        ranges[t] = (c != 0u && closed) ? make_uint2(running, running + c) : make_uint2(0u, 0u);
        with_list += (c != 0u && closed) ? 1 : 0;
        running += c;
    }
    if (threadIdx.x == 1023) tile_start[tiles] = running;
    // the tiles that got a list (the blend from the sorted lists asks whether they can fill the chip, blend.hip)
    __shared__ int s_with[16];
    for (int off = 32; off > 0; off >>= 1) with_list += __shfl_xor(with_list, off, kWave);
    if (lane == 0) s_with[wave] = with_list;
    __syncthreads();
    if (threadIdx.x == 0 && nonempty) {
        int tot = 0;
        for (int w = 0; w < 16; ++w) tot += s_with[w];
        *nonempty = (uint32_t)tot;
    }
}

// ---- emission ------------------------------------------------------------------------------------
// One wavefront per unit (<= 2048 consecutive entries of one block list = 32 batches of 64).
//   masks : lane = entry. Per batch, 16 ballots give, for each of the 8 tile columns and 8 tile rows
//           of the block, which of the 64 entries cover it; v_writelane files them TRANSPOSED: lane w
//           of register xm[c] / ym[r] holds the 64-entry mask of batch w. (coverage of tile (c, r) by
//           batch w is xm[c] & ym[r]: rectangles are products of a column range and a row range)
//   tiles : per tile two LDS reads and one AND leave lane w holding the tile's mask of batch w; bit
//           counts + a 32-lane DPP prefix give where each batch's keys start in the tile's run. Then,
//           batch by batch, EXEC = that mask and the covered entries store their key / value at
//           start + v_mbcnt rank: compacted, in entry order, one contiguous burst per (tile, batch).
// Every (unit, tile) run is therefore written front to back by one wave in consecutive bursts (whole
// lines form in L2); there is no per-entry serial work and no staging of keys in LDS.
constexpr int kEmitWaves = 4;                // independent waves per workgroup
constexpr uint32_t kDenseUnitKeys = 40000;   // keys per unit (2048 entries) from which units are dealt in quarters
constexpr uint32_t kGroup = 128;             // keys per dense store group: 8 key lines + 4 value lines
constexpr uint32_t kRun = 512;               // slots of the per-wave run buffer (>= kGroup - 1 + 4 batches of 64)
constexpr int kCheck = 4;                    // batches between two looks at the run buffer's fill

// Narrow path (the ends of a run, see the cut below): keys [a, b) of the run buffer, one per lane, plain stores — the
// lines they touch are shared with the neighbouring unit's run and should meet in L2 (streaming stores here: 0.69 -> 0.78 ms).
constexpr uint32_t kLine = 32;               // keys per 128-byte line of the value array (= two lines of the key array)
constexpr uint32_t kLongRun = 512;           // runs longer than this may be cut at lines, the others at whole groups
__device__ __forceinline__ void store_run_sparse(const uint2* run_buf, uint32_t a, uint32_t b, uint32_t tile,
                                                 uint64_t* __restrict__ keys, uint32_t* __restrict__ values) {
    const uint32_t lane = threadIdx.x & (kWave - 1);
    for (uint32_t g = a + lane; g < b; g += kWave) {
        const uint2 q = run_buf[g & (kRun - 1)];
        keys[g] = ((uint64_t)tile << 32) | q.x;
        values[g] = q.y;
    }
}
// Dense path: the `count` keys [a, a + count), a a multiple of kLine, count a multiple of kLine and <= kGroup: two keys
// per lane, whole lines of both arrays only.
__device__ __forceinline__ void store_run_group(const uint2* run_buf, uint32_t a, uint32_t count, uint32_t tile,
                                                uint64_t* __restrict__ keys, uint32_t* __restrict__ values) {
    const uint32_t lane = threadIdx.x & (kWave - 1);
    if (2u * lane >= count) return;
    const uint4 q = *reinterpret_cast<const uint4*>(run_buf + ((a + 2u * lane) & (kRun - 1)));    // (a pair never wraps: a + 2 lane is even)
    // streaming (nt) stores: 12 R bytes go out once and at most 1 % of them is read back by the blend
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x4 kk = {q.x, tile, q.z, tile};      // two consecutive 64-bit keys
    const u32x2 vv = {q.y, q.w};
    __builtin_nontemporal_store(kk, reinterpret_cast<u32x4*>(keys + a + 2u * lane));
    __builtin_nontemporal_store(vv, reinterpret_cast<u32x2*>(values + a + 2u * lane));
}

__global__ __launch_bounds__(kEmitWaves* kWave) void block_emit_kernel(BlockMeta meta, int nb, int nbx, int gx, int gy,
                                                                        const uint64_t* __restrict__ ent_rd,
                                                                        const uint32_t* __restrict__ ent_idx,
                                                                        const uint2* __restrict__ unit_masks,
                                                                        const uint32_t* __restrict__ cnt,
                                                                        const uint32_t* __restrict__ tile_start,
                                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ values,
                                                                        uint32_t r_total) {
    __shared__ uint2 s_maskT[kEmitWaves][16][kBatches];      // [column 0..7 | row 0..7][batch]
    __shared__ __attribute__((aligned(16))) uint2 s_run[kEmitWaves][kRun + kWave];   // {depth bits, index}, slot = output index mod 512; + one scrap slot per lane
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint32_t total_units = meta.unit_start()[meta.nbp];
    const uint32_t* ent_rd32 = reinterpret_cast<const uint32_t*>(ent_rd);
    uint2* run_buf = s_run[wave];
    uint2* scrap = run_buf + kRun + lane;
    // Work items differ in key count, so they come from a work queue — except the first one of every wave
    // (ticket = wave number): thousands of waves hitting one counter at launch would queue up on it.
    // An item is a unit or, on frames of large splats, a quarter of a unit's tiles (tile rows 0-1 | 2-3 | 4-5 | 6-7 of the
    // block; each quarter loads the unit's entries and masks for itself). With whole units the bench frame has 2.9 units
    // per wave and units of very different weight (the splats next to the camera cover their whole block): some waves end
    // on their second unit and others on their fourth, and the waves were resident for 83 % of the launch on average
    // (SQ_WAVE_CYCLES). Measured, whole units / halves / quarters: bench frame (22 keys per entry) 0.688 / 0.656 / 0.634 ms;
    // 3840 x 2160 (35 keys per entry, 8 units per wave) 2.997 / 2.959 / 2.903 ms; but eye (0,0,-9) (17 keys per entry)
    // 0.591 / - / 0.603 ms and from outside the cloud (11 keys per entry) 0.361 / 0.372 / 0.397 ms: there the units are
    // alike and an item's own loads (24 KB of entries and masks) weigh more than the tail. Eighths: 0.683 ms on the bench
    // frame, 2.844 ms at 3840 x 2160. The four quarters of a unit taken by the four waves of one workgroup together (one
    // ticket per workgroup, so that the loads miss the caches once): 0.650 ms on the bench frame, the waves wait for the
    // slowest quarter. One queue per XCD (XCD x takes the units u = x mod 8, so that the four quarters of a unit are loaded
    // through one L2): 0.703 ms on the bench frame, 3.17 ms at 3840 x 2160 — the quarters' re-reads (FETCH_SIZE 84 -> 248 MB
    // per launch) are not what the kernel waits for. Nor does dealing the first 50 / 70 / 85 % of the units statically
    // (workgroup g: units g, g + G, ..., a quarter per wave, no waiting) and only the rest from the queue help: 0.645 /
    // 0.654 / 0.704 ms. What pays is the fine-grained dynamic dealing itself. (`gpurun_out/r3b` - `r3d`.)
    const uint32_t nwaves = gridDim.x * kEmitWaves;
    const uint32_t keys_per_unit = r_total / max(total_units, 1u);
    uint32_t parts = (total_units >= 32u * nwaves || keys_per_unit < kDenseUnitKeys) ? 1u : 4u;
    // A LIGHT frame has fewer units than the launch has waves (1 M splats: 800 units for 4 096 waves): a wave alone on its
    // SIMD takes 0.11 ms for the 64 tiles x 32 batches of its one unit, whatever the keys — the frame's emission lasted
    // 0.13 ms for 0.17 GB. With room for them the items are made smaller, down to one tile row of the block each, as long as
    // every wave still gets at most one.
#ifndef GSR_LIGHT_PARTS
#define GSR_LIGHT_PARTS 8
#endif
    if (parts == 1u) {
        while (parts < (uint32_t)GSR_LIGHT_PARTS && total_units * parts * 2u <= nwaves) parts *= 2u;
    }
    const uint32_t total_items = total_units * parts, tiles_per_item = 64u / parts;
    bool first = true;
    for (;;) {
        uint32_t item = blockIdx.x * kEmitWaves + (uint32_t)wave;
        if (!first) {
            if (lane == 0) item = nwaves + atomicAdd(&meta.tickets()[1], 1u);
            item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
        }
        first = false;
        if (item >= total_items) break;
        const uint32_t u = item / parts, t_first = (item % parts) * tiles_per_item;
        const UnitInfo ui = locate_unit(u, meta, nb, nbx);
        const uint32_t e0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ui.e0);
        const uint32_t e1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ui.e1);
        const uint32_t bx0 = ui.bx * kBW, by0 = ui.by * kBH;
        // lane = tile: where this unit's keys of the tile start in the sorted list
        const uint32_t tx = bx0 + (uint32_t)(lane & 7), ty = by0 + (uint32_t)(lane >> 3);
        const bool in_grid = tx < (uint32_t)gx && ty < (uint32_t)gy;
        const uint32_t tile_v = ty * (uint32_t)gx + tx;
        const uint32_t base_v = in_grid ? tile_start[tile_v] + cnt[(size_t)u * 64 + lane] : 0u;

        // ---- the unit's transposed coverage masks (unit_masks_kernel): 4 KB, HBM -> LDS ----
        if (lane < kBatches) {
            const uint2* src = unit_masks + (size_t)u * 16 * kBatches + lane;
#pragma unroll
            for (int k = 0; k < 16; ++k) s_maskT[wave][k][lane] = src[k * kBatches];
        }
        // ---- the entries' {depth bits, index}: lane = entry, register pair b = batch ----
        uint2 ent[kBatches];
#pragma unroll
        for (int b = 0; b < kBatches; ++b) {
            const uint32_t i = min(e0 + (uint32_t)(b * kWave + lane), e1 - 1u);   // (entries past the end have empty masks)
            ent[b].x = ent_rd32[2 * (size_t)i + 1];
            ent[b].y = ent_idx[i];
        }
        __builtin_amdgcn_wave_barrier();

        // ---- tiles ----
        for (uint32_t t = t_first; t < t_first + tiles_per_item; ++t) {
            const uint2 xm = s_maskT[wave][t & 7u][lane & (kBatches - 1)];
            const uint2 ym = s_maskT[wave][8u + (t >> 3)][lane & (kBatches - 1)];
            const uint32_t tm_lo = (lane < kBatches) ? (xm.x & ym.x) : 0u, tm_hi = (lane < kBatches) ? (xm.y & ym.y) : 0u;
            const uint32_t c = (uint32_t)__popc(tm_lo) + (uint32_t)__popc(tm_hi);
            const uint32_t incl = prefix32_inclusive(c);
            const uint32_t run = (uint32_t)__builtin_amdgcn_readlane((int)incl, kBatches - 1);
            if (run == 0u) continue;
            const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)base_v, (int)t);
            const uint32_t tile = (uint32_t)__builtin_amdgcn_readlane((int)tile_v, (int)t);
            if (base + run > r_total || base + run < base) continue;      // (never beyond the arrays)
            // The run is compacted through the run buffer: batch by batch the covered entries write
            // {depth, index} at slot (output index mod 512), output index = where the batch starts +
            // v_mbcnt rank; the other lanes write their scrap slot (no branch, so consecutive batches overlap
            // freely; the compiler computes the slot under an EXEC mask — a v_cndmask on the mask's SGPR pair
            // instead, forced through inline assembly, measured the same: 0.637 ms either way). Every 4 batches the
            // complete groups of the output are stored densely. (Splitting a group's store into the LDS read at one
            // look and the HBM stores at the next, so that the read's latency is covered: no change either. Only the
            // covered lanes writing — EXEC = the mask around the slot arithmetic AND the LDS write, in inline assembly, no
            // branch: no scrap slots, no bank conflicts between the two kinds of lanes (4.6e7 conflict cycles per launch),
            // a third of the LDS write traffic — 0.6335 against 0.635 ms: the kernel does not wait for its LDS.)
            const uint32_t start_v = base + incl - c;                 // lane w: output index of batch w's first key
            // Where the groups are cut. Coarse: at multiples of 128 keys — the run's keys in front of its first multiple and
            // behind its last one take the narrow path (18 % of all keys on the bench frame). Fine: at lines of the value
            // array (32 keys = 128 bytes there, 256 bytes of keys): only fewer than 32 keys at either end are narrow, the
            // groups in between are still whole lines of both arrays. A run goes fine if it is long and its first four
            // batches already reach the first line boundary (dense coverage); the others stay coarse, where the fine cut's
            // extra store sequences cost more than they save. (Two copies of the loop: the choice is made once per run.)
            const uint32_t last = base + run;
            const uint32_t origin_fine = (base + kLine - 1u) & ~(kLine - 1u);
            const bool fine = run > kLongRun && base + (uint32_t)__builtin_amdgcn_readlane((int)incl, kCheck - 1) >= origin_fine;
            uint32_t flushed = base;
            if (fine) {
#pragma unroll
                for (int w = 0; w < kBatches; ++w) {
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)tm_lo, w);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)tm_hi, w);
                    const uint32_t start = (uint32_t)__builtin_amdgcn_readlane((int)start_v, w);
                    const unsigned long long m = ((unsigned long long)hi << 32) | lo;
                    const uint32_t o = start + __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
                    uint2* dst = __builtin_amdgcn_inverse_ballot_w64(m) ? run_buf + (o & (kRun - 1)) : scrap;
                    *dst = ent[w];
                    if ((w % kCheck) == kCheck - 1) {
                        const uint32_t end = base + (uint32_t)__builtin_amdgcn_readlane((int)incl, w);
                        if (w == kCheck - 1) {
                            __builtin_amdgcn_wave_barrier();
                            store_run_sparse(run_buf, base, origin_fine, tile, keys, values);
                            __builtin_amdgcn_wave_barrier();
                            flushed = origin_fine;
                        }
                        while (end >= flushed + kGroup) {
                            __builtin_amdgcn_wave_barrier();
                            store_run_group(run_buf, flushed, kGroup, tile, keys, values);
                            __builtin_amdgcn_wave_barrier();
                            flushed += kGroup;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                const uint32_t whole = (last - flushed) & ~(kLine - 1u);       // < kGroup: every complete group has left
                store_run_group(run_buf, flushed, whole, tile, keys, values);
                store_run_sparse(run_buf, flushed + whole, last, tile, keys, values);
            } else {
#pragma unroll
                for (int w = 0; w < kBatches; ++w) {
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)tm_lo, w);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)tm_hi, w);
                    const uint32_t start = (uint32_t)__builtin_amdgcn_readlane((int)start_v, w);
                    const unsigned long long m = ((unsigned long long)hi << 32) | lo;
                    const uint32_t o = start + __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
                    uint2* dst = __builtin_amdgcn_inverse_ballot_w64(m) ? run_buf + (o & (kRun - 1)) : scrap;
                    *dst = ent[w];
                    if ((w % kCheck) == kCheck - 1) {
                        const uint32_t end = base + (uint32_t)__builtin_amdgcn_readlane((int)incl, w);
                        while (end >= (flushed & ~(kGroup - 1)) + kGroup) {
                            const uint32_t boundary = (flushed & ~(kGroup - 1)) + kGroup;
                            __builtin_amdgcn_wave_barrier();
                            if ((flushed & (kGroup - 1)) == 0u) store_run_group(run_buf, flushed, kGroup, tile, keys, values);
                            else store_run_sparse(run_buf, flushed, boundary, tile, keys, values);
                            __builtin_amdgcn_wave_barrier();
                            flushed = boundary;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                store_run_sparse(run_buf, flushed, last, tile, keys, values);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- blend straight from the block lists -------------------------------------------------------------
// The tile's records, in list order, are the entries of its block whose (column mask & row mask) bit is
// set — the same filter the emission applies. Reading them from the block lists makes the blend
// independent of the emission (beside which it may run, on a second stream — or which may not run at all:
// GSR_FLAG_NO_SORTED_LISTS). Per batch of 64 entries the covered ones fetch their record, the ones whose
// footprint can reach the tile are staged compacted (v_mbcnt rank) in wave-private LDS and composited by the
// shared core (blend_core.hpp); the 256-record batches of the reference survive as the granularity of the
// staged-record count.
struct BlockBlendParams {
    BlockMeta meta;
    int nbx;
    const uint2* unit_masks;
    const uint32_t* ent_idx;
    const uint2* ranges;
    const float2* means2D;
    const float* colors;
    const float4* conic_opacity;
    float* final_t;
    uint32_t* n_contrib;
    const float* background;
    float* out_color;
    unsigned long long* staged_counter;
    float t_cutoff;
    FrameDims dims;
    int num_tiles;
    int waves_per_tile;          // 1, or 4 (one 16 x 4 strip per wave) when the call has few tiles
    TileOrder history;           // longest tiles first (blend_core.hpp)
    uint32_t dc_stride;          // 0, or 48: `colors` is the SH array (TileFeed::dc_stride)
};

__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(5))) void blend_blocks_kernel(const BlockBlendParams p) {
    __shared__ StagedRecords s_staged;
    exp_table_init(s_staged.exp_tab, (int)threadIdx.x);      // (wave-private LDS: ordered inside the wave)
    const uint32_t clock_begin = tile_clock();
    const int wpt = p.waves_per_tile;
    const int wg = (int)blockIdx.x / wpt;
    const int tile_local = tile_of_workgroup(p.history.order ? (int)p.history.order[wg] : wg, p.dims.grid_x, p.dims.row_end - p.dims.row_begin);
    if (tile_local < 0) return;
    const int tile = p.dims.row_begin * p.dims.grid_x + tile_local;
    const int tx = tile % p.dims.grid_x, ty = tile / p.dims.grid_x;
    const int lane = threadIdx.x;
    TileLanes s;
    tile_lanes_init(s, tx, ty, lane, p.dims.width, p.dims.height, wpt == 1 ? -1 : (int)blockIdx.x % wpt);
    const uint2 range = p.ranges[tile];
    const uint32_t total = range.y - range.x;
    unsigned long long staged = 0;
    bool all_done = tile_lanes_all_done(s) || total == 0u;

    const uint32_t b = (uint32_t)(ty / kBH) * (uint32_t)p.nbx + (uint32_t)(tx / kBW);
    const uint32_t col = (uint32_t)(tx % kBW), row = 8u + (uint32_t)(ty % kBH);
    const uint32_t u0 = p.meta.unit_start()[b], u1 = p.meta.unit_start()[b + 1];
    const uint32_t list0 = p.meta.list_start()[b];
    TileFeed feed;
    feed.means2D = p.means2D; feed.colors = p.colors; feed.conic_opacity = p.conic_opacity;
    feed.dc_stride = p.dc_stride;
    feed.box = tile_box(tx, ty, p.dims.width, p.dims.height);
    feed.total = total; feed.t_cutoff = p.t_cutoff;
    // The tile's batches, in list order: batch w of unit u, for every (u, w) whose mask is not empty. An iterator walks
    // them; ids are fetched two batches ahead and records one batch ahead of the batch being composited.
    uint32_t it_u = u0, it_nz = 0, it_pos = 0;          // unit, its batches still to come (bit w), list positions handed out
    uint32_t tm_lo = 0, tm_hi = 0;                      // lane w < 32: the tile's mask of batch w of unit it_u
    bool it_loaded = false;
    auto next_batch = [&]() {
        RecordBatch nb;
        for (;;) {
            if (!it_loaded) {
                if (it_u >= u1) return nb;               // (valid = false)
                const uint2* um = p.unit_masks + (size_t)it_u * 16 * kBatches + (lane & (kBatches - 1));
                const uint2 xm = um[col * kBatches], ym = um[row * kBatches];
                tm_lo = (lane < kBatches) ? (xm.x & ym.x) : 0u;
                tm_hi = (lane < kBatches) ? (xm.y & ym.y) : 0u;
                it_nz = (uint32_t)__ballot((tm_lo | tm_hi) != 0u);
                it_loaded = true;
            }
            if (it_nz != 0u) break;
            ++it_u;
            it_loaded = false;
        }
        const int w = __builtin_amdgcn_readfirstlane(__ffs((int)it_nz) - 1);
        it_nz &= it_nz - 1u;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)tm_lo, w);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)tm_hi, w);
        nb.mask = ((unsigned long long)hi << 32) | lo;
        nb.pos = it_pos;
        nb.valid = true;
        it_pos += nb.count();
        if (nb.present()) nb.id = p.ent_idx[list0 + (it_u - u0) * kUnit + (uint32_t)w * kWave + (uint32_t)lane];
        if (it_nz == 0u) { ++it_u; it_loaded = false; }
        return nb;
    };
    RecordBatch b0, b1;
    if (!all_done) {
        b0 = next_batch();
        fetch_records(b0, feed);
        b1 = next_batch();
    }
    while (b0.valid && !all_done) {
        fetch_records(b1, feed);
        RecordBatch b2 = next_batch();
        all_done = stage_and_composite(s, feed, s_staged, b0, staged);
        b0 = b1;
        b1 = b2;
    }
    tile_lanes_write(s, p.dims.width, p.dims.height, p.background, p.final_t, p.n_contrib, p.out_color);
    if (p.staged_counter && lane == 0) atomicAdd(p.staged_counter, staged);
    // how far into the block's list this tile looked (gsr_backward after GSR_FLAG_NO_SORTED_LISTS keeps per-entry sums
    // for that part of the list only)
    const uint32_t units = it_u - u0 + (it_loaded ? 1u : 0u);
    if (lane == 0 && units != 0u) atomicMax(&p.meta.walked()[b], units);
    if (p.history.ticks && lane == 0 && (int)blockIdx.x % wpt == 0) p.history.ticks[tile] = tile_clock() - clock_begin;
}

}  // namespace

bool blockbin_supported(int grid_x, int grid_y) {
    if (grid_x > 255 || grid_y > 255) return false;
    const int nb = ((grid_x + kBW - 1) / kBW) * ((grid_y + kBH - 1) / kBH);
    return nb <= kMaxBlocks;
}

static size_t blockbin_table_bytes(size_t n) { return align128(((n + kCoarseSmall - 1) / kCoarseSmall) * (size_t)kMaxBlocks * 4); }
static size_t blockbin_partial_bytes(size_t n) {
    const size_t chunks = (n + kCoarseSmall - 1) / kCoarseSmall;
    return align128(((chunks + kScanRows - 1) / kScanRows) * (size_t)kMaxBlocks * 4);
}
// per-Gaussian scratch (geometry chunk): chunk table, row-group partials, block meta, tile counts / starts
size_t blockbin_geo_bytes(size_t n) {
    return blockbin_table_bytes(n) + blockbin_partial_bytes(n) + align128(kBlockMetaWords * 4) +
           2 * align128((65536 + 1) * 4);
}
// per-instance scratch (binning chunk): keys per (unit, tile) + the units' transposed coverage masks
static size_t blockbin_cnt_bytes(size_t r) { return align128((r / kUnit + kMaxBlocks + 1) * 64 * 4); }
size_t blockbin_bin_bytes(size_t r) { return blockbin_cnt_bytes(r) + align128((r / kUnit + kMaxBlocks + 1) * 16 * kBatches * 8); }

namespace {
// Where the plan's tables live inside the two scratch areas.
struct PlanTables {
    int nbx, nb, nbp, chunk;
    uint32_t chunks, groups, tiles, max_units;
    uint32_t *table, *partial, *tile_count, *tile_start, *cnt;
    uint2* unit_masks;
    BlockMeta meta;
};
PlanTables plan_tables(int n, int grid_x, int grid_y, uint32_t r_total, char* geo_scratch, char* bin_scratch) {
    PlanTables t;
    t.nbx = (grid_x + kBW - 1) / kBW;
    t.nb = t.nbx * ((grid_y + kBH - 1) / kBH);
    t.nbp = (t.nb + kWave - 1) / kWave * kWave;
    t.chunk = t.nb > 256 ? kCoarseSmall : kCoarse;
    t.chunks = (uint32_t)((n + t.chunk - 1) / t.chunk);
    t.groups = (t.chunks + kScanRows - 1) / kScanRows;
    t.tiles = (uint32_t)(grid_x * grid_y);
    t.max_units = r_total / kUnit + (uint32_t)t.nb + 1u;
    char* p = geo_scratch;
    t.table = reinterpret_cast<uint32_t*>(p); p += blockbin_table_bytes((size_t)n);
    t.partial = reinterpret_cast<uint32_t*>(p); p += blockbin_partial_bytes((size_t)n);
    t.meta.w = reinterpret_cast<uint32_t*>(p); t.meta.nbp = t.nbp; p += align128(kBlockMetaWords * 4);
    t.tile_count = reinterpret_cast<uint32_t*>(p); p += align128((65536 + 1) * 4);
    t.tile_start = reinterpret_cast<uint32_t*>(p);
    t.cnt = reinterpret_cast<uint32_t*>(bin_scratch);
    t.unit_masks = reinterpret_cast<uint2*>(bin_scratch + blockbin_cnt_bytes(r_total));
    return t;
}
}  // namespace

BlockFeed block_feed(int n, int grid_x, int grid_y, uint32_t r_total, char* geo_scratch, const uint32_t* ent_idx, char* bin_scratch) {
    const PlanTables t = plan_tables(n, grid_x, grid_y, r_total, geo_scratch, bin_scratch);
    BlockFeed f;
    f.meta = t.meta;
    f.nbx = t.nbx;
    f.unit_masks = t.unit_masks;
    f.prefix = t.cnt;
    f.ent_idx = ent_idx;
    f.acc = nullptr;
    f.acc_floats = 0;
    return f;
}

// Everything of the block plan up to (not including) the emission: block lists, unit masks, prefixes and
// the tile ranges. sorted_depth / sorted_idx / sorted_rect: the n visible Gaussians in depth order (depth bits, index,
// packed rectangle). ent_rd / ent_idx: R-sized scratch for the block lists. ev_coarse_end: optional event recorded
// after the block lists (stage timing).
int launch_block_binning(int n, const uint32_t* sorted_depth, const uint32_t* sorted_idx, const uint32_t* sorted_rect,
                         int grid_x, int grid_y, uint32_t r_total, char* geo_scratch, uint64_t* ent_rd,
                         uint32_t* ent_idx, char* bin_scratch, uint32_t* ranges, bool close_single, hipStream_t stream,
                         hipEvent_t ev_coarse_end, uint32_t* nonempty_tiles, uint32_t* skipped_stamp, int cus) {
    const PlanTables t = plan_tables(n, grid_x, grid_y, r_total, geo_scratch, bin_scratch);
    if (t.chunk == kCoarse)
        hipLaunchKernelGGL(coarse_count_kernel<kCoarse>, dim3(t.chunks), dim3(kCoarse), 0, stream, n, sorted_rect, t.nbx, t.nbp,
                           t.table);
    else
        hipLaunchKernelGGL(coarse_count_kernel<kCoarseSmall>, dim3(t.chunks), dim3(kCoarseSmall), 0, stream, n, sorted_rect, t.nbx,
                           t.nbp, t.table);
    GSR_LAUNCH_CHECK("coarse_count_kernel");
    hipLaunchKernelGGL(blockscan_reduce_kernel, dim3(t.groups), dim3(t.nbp), 0, stream, t.table, t.chunks, t.nbp, t.partial);
    GSR_LAUNCH_CHECK("blockscan_reduce_kernel");
    hipLaunchKernelGGL(blockscan_partials_kernel, dim3(1), dim3(t.nbp), 0, stream, t.partial, t.groups, t.nb, t.meta);
    GSR_LAUNCH_CHECK("blockscan_partials_kernel");
    hipLaunchKernelGGL(blockscan_apply_kernel, dim3(t.groups), dim3(t.nbp), 0, stream, t.table, t.chunks, t.nbp, t.partial,
                       t.meta.list_start());
    GSR_LAUNCH_CHECK("blockscan_apply_kernel");
    const size_t mask_bytes = ((size_t)t.nb * (t.chunk / 32 + 1) * 2 + kMaxBlocks + t.nbp + 3 * (size_t)t.chunk) * 4;
    // More than the default 48 KB of dynamic LDS: asked for once per device and host thread (the most a grid of kMaxBlocks
    // blocks can need), not on every call.
#define GSR_STEP_LDS(kernel, chunk_)                                                                                          \
    do {                                                                                                                      \
        static thread_local std::map<int, size_t> granted;   /* dynamic LDS this thread has been granted, per device */       \
        int dev_ = 0;                                                                                                         \
        GSR_HIP_TRY(hipGetDevice(&dev_));                                                                                     \
        if (granted[dev_] < mask_bytes) {                                                                                     \
            /* the most a grid of kMaxBlocks blocks can need, so that no later frame asks again; a device that has not got */ \
            /* that much is asked for what THIS frame needs, and only a frame that does not fit fails                     */ \
            const size_t most_ = ((size_t)kMaxBlocks * ((chunk_) / 32 + 1) * 2 + kMaxBlocks + kMaxBlocks + 3 * (size_t)(chunk_)) * 4; \
            const size_t want_ = most_ < 160 * 1024 ? most_ : 160 * 1024;                                                     \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                    (int)want_) == hipSuccess) {                                                              \
                granted[dev_] = want_;                                                                                        \
            } else {                                                                                                          \
                (void)hipGetLastError();                                                                                      \
                GSR_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                (int)mask_bytes));                                                            \
                granted[dev_] = mask_bytes;                                                                                   \
            }                                                                                                                 \
        }                                                                                                                     \
    } while (0)
    if (t.chunk == kCoarse) {
        if (mask_bytes > 48 * 1024) GSR_STEP_LDS(coarse_emit_kernel<kCoarse>, kCoarse);
        hipLaunchKernelGGL(coarse_emit_kernel<kCoarse>, dim3(t.chunks), dim3(kCoarse), mask_bytes, stream, n, sorted_depth,
                           sorted_idx, sorted_rect, t.table, t.nbx, t.nb, t.nbp, ent_rd, ent_idx);
    } else {
        if (mask_bytes > 48 * 1024) GSR_STEP_LDS(coarse_emit_kernel<kCoarseSmall>, kCoarseSmall);
        hipLaunchKernelGGL(coarse_emit_kernel<kCoarseSmall>, dim3(t.chunks), dim3(kCoarseSmall), mask_bytes, stream, n, sorted_depth,
                           sorted_idx, sorted_rect, t.table, t.nbx, t.nb, t.nbp, ent_rd, ent_idx);
    }
#undef GSR_STEP_LDS
    GSR_LAUNCH_CHECK("coarse_emit_kernel");
    if (ev_coarse_end) GSR_HIP_TRY(hipEventRecord(ev_coarse_end, stream));

    // persistent grid: enough waves to fill the chip, never more than there can be units
    const uint32_t count_wgs = std::min<uint32_t>((t.max_units + 3) / 4, (uint32_t)cus * 8u);
    hipLaunchKernelGGL(unit_masks_kernel, dim3(count_wgs), dim3(256), 0, stream, t.meta, t.nb, t.nbx, ent_rd, t.unit_masks, t.cnt);
    GSR_LAUNCH_CHECK("unit_masks_kernel");
    hipLaunchKernelGGL(block_prefix_kernel, dim3(t.nb), dim3(256), 0, stream, t.meta, t.nbx, grid_x, grid_y, t.cnt, t.tile_count);
    GSR_LAUNCH_CHECK("block_prefix_kernel");
    hipLaunchKernelGGL(tile_start_kernel, dim3(1), dim3(1024), 0, stream, t.tile_count, t.tiles, t.tile_start,
                       reinterpret_cast<uint2*>(ranges), r_total, close_single, nonempty_tiles, skipped_stamp);
    GSR_LAUNCH_CHECK("tile_start_kernel");
    return GSR_OK;
}

// The emission: the sorted keys / values written from the tables launch_block_binning left.
int launch_block_emit(int n, int grid_x, int grid_y, uint32_t r_total, char* geo_scratch, const uint64_t* ent_rd,
                      const uint32_t* ent_idx, char* bin_scratch, uint64_t* keys, uint32_t* values, hipStream_t stream, bool beside_blend,
                      int cus) {
    const PlanTables t = plan_tables(n, grid_x, grid_y, r_total, geo_scratch, bin_scratch);
    // Workgroups (of four waves) per CU: two where the Gaussians cover many tiles each (long, dense runs: three or four
    // measured 1.5 % slower on the bench frame and at 4K), four where they cover few (short runs, the waves wait more than
    // they store: 0.416 -> 0.366 ms from outside the cloud, R / V = 23 against 88 on the bench frame).
    // With the blend beside it (second stream) always two: four of these workgroups hold 448 of a SIMD's 512 vector registers
    // and no wave of the blend fits until they retire — the stand-in from outside the cloud, overlapped: blend 0.96 ms for
    // 0.45 alone, the frame 1.70 for 1.50.
    const uint32_t per_cu = ((uint64_t)r_total >= 48ull * (uint64_t)n || beside_blend) ? 2u : 4u;
    // (a unit may be dealt as four items, see the kernel: a wave per item on small frames)
    const uint32_t emit_wgs = std::min<uint32_t>(t.max_units, (uint32_t)cus * per_cu);
    hipLaunchKernelGGL(block_emit_kernel, dim3(emit_wgs), dim3(kEmitWaves * kWave), 0, stream, t.meta, t.nb, t.nbx, grid_x, grid_y,
                       ent_rd, ent_idx, t.unit_masks, t.cnt, t.tile_start, keys, values, r_total);
    GSR_LAUNCH_CHECK("block_emit_kernel");
    return GSR_OK;
}

// The blend, fed from the block lists (independent of launch_block_emit).
int launch_blend_blocks(int n, const FrameDims& d, uint32_t r_total, char* geo_scratch, const uint32_t* ent_idx, char* bin_scratch,
                        const uint32_t* ranges, const float* means2D, const float* colors, const float* conic_opacity,
                        float* final_t, uint32_t* n_contrib, const float* background, float* out_color,
                        unsigned long long* staged_counter, float t_cutoff, hipStream_t stream,
                        const uint32_t* tile_order, uint32_t* tile_ticks, bool colors_are_shs) {
    const PlanTables t = plan_tables(n, d.grid_x, d.grid_y, r_total, geo_scratch, bin_scratch);
    BlockBlendParams p;
    p.dc_stride = colors_are_shs ? 48u : 0u;
    p.history.order = tile_order; p.history.ticks = tile_ticks;
    p.meta = t.meta;
    p.nbx = t.nbx;
    p.unit_masks = t.unit_masks;
    p.ent_idx = ent_idx;
    p.ranges = reinterpret_cast<const uint2*>(ranges);
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
    // A call with few tiles (one rank's band of a sharded frame) cannot fill the chip with one wave per
    // tile: four waves per tile then, one 16 x 4 strip each. (Not when the staged records are counted:
    // that count is per tile, the reference's "whole tile done" test.)
    p.waves_per_tile = (p.num_tiles <= 1536 && !staged_counter) ? 4 : 1;     // (measured: 960 tiles 0.17 -> 0.13 ms, 4080 tiles 0.16 -> 0.29 ms)
    hipLaunchKernelGGL(blend_blocks_kernel, dim3((unsigned)(patch_workgroups(d.grid_x, d.row_end - d.row_begin) * p.waves_per_tile)),
                       dim3(kWave), 0, stream, p);
    GSR_LAUNCH_CHECK("blend_blocks_kernel");
    return GSR_OK;
}

}  // namespace gsr
