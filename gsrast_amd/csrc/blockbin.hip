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
//       list per tile block (entries = rectangle, depth bits, index; 12 B, E <= R of them)
//   unit_count   : a unit = 2048 consecutive entries of one block list; keys per (unit, tile) from
//       a 9 x 9 difference array in LDS
//   block_prefix / tile_start : prefix of those counts down the units of a block, then over the
//       tiles in tile order = where every (unit, tile) run starts in the sorted list
//   block_emit   : one wavefront per unit walks its entries; lane (tx, ty) appends the entry to
//       its tile's 32-slot LDS ring when the rectangle covers the tile, and every time a ring
//       crosses a 32-key boundary of the OUTPUT index the wave stores those 32 keys / values as
//       whole 128-byte lines. Only the first and last line of a (unit, tile) run are partial.
// The result is bit-identical to the stable 64-bit sort (same lists, same order inside a tile).
// R-sized traffic: 12 R bytes written once (the reference's emit + 6-pass sort moves > 150 R).
#include "gsr_common.hpp"

namespace gsr {
namespace {

constexpr int kBW = 8, kBH = 8;          // tiles per block: one lane per tile
constexpr int kCoarse = 1024;            // Gaussians per workgroup of the coarse passes
constexpr int kUnit = 2048;              // block-list entries per emission unit
constexpr int kScanRows = 64;            // table rows per workgroup of the block scan
constexpr int kMaxBlocks = 512;          // 8 x 8-tile blocks per frame (4K: 30 x 17 = 510)
constexpr int kRing = 32;                // keys staged per tile
constexpr int kRingStride = 2 * kRing + 4;   // dwords per tile ring: 32 x {depth, idx} + pad (bank skew, 16-B aligned)

struct BlockMeta {                       // u32 words in HBM
    // [0, nbp]            list_start : entry index where the list of block b starts (nbp + 1 words)
    // [nbp+1, 2nbp+1]     unit_start : first unit of block b (nbp + 1 words; [nb] = total units)
    // then                ticket_count, ticket_emit (work queues of the two persistent kernels)
    uint32_t* w;
    int nbp;
    __host__ __device__ uint32_t* list_start() const { return w; }
    __host__ __device__ uint32_t* unit_start() const { return w + nbp + 1; }
    __host__ __device__ uint32_t* tickets() const { return w + 2 * (nbp + 1); }
};

inline size_t align128(size_t v) { return (v + 127) / 128 * 128; }

// ---- coarse pass 1: entries per (chunk of 1024 depth-consecutive Gaussians, block) ------------
__global__ __launch_bounds__(kCoarse) void coarse_count_kernel(int n, const uint32_t* __restrict__ sorted_depth,
                                                               const uint32_t* __restrict__ sorted_idx,
                                                               const uint32_t* __restrict__ rect_by_index, int nbx, int nbp,
                                                               uint32_t* __restrict__ rect_packed,
                                                               uint32_t* __restrict__ table) {
    __shared__ uint32_t s_cnt[kMaxBlocks];
    if ((int)threadIdx.x < nbp) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int r = blockIdx.x * kCoarse + threadIdx.x;
    uint32_t packed = 0;
    if (r < n && sorted_depth[r] != 0xFFFFFFFFu) packed = rect_by_index[sorted_idx[r]];
    if (r < n) rect_packed[r] = packed;
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
#pragma unroll 8
    for (uint32_t g = 0; g < groups; ++g) {
        const uint32_t v = partial[(size_t)g * nbp + threadIdx.x];
        partial[(size_t)g * nbp + threadIdx.x] = running;
        running += v;
    }
    const uint32_t len = ((int)threadIdx.x < nb) ? running : 0u;
    uint32_t total_entries, total_units;
    const uint32_t ls = block_exclusive(len, s_ws, &total_entries);
    const uint32_t us = block_exclusive((len + kUnit - 1) / kUnit, s_ws, &total_units);
    meta.list_start()[threadIdx.x] = ls;
    meta.unit_start()[threadIdx.x] = us;
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
#pragma unroll 8
    for (uint32_t r = r0; r < r1; ++r) {
        const size_t cell = (size_t)r * nbp + threadIdx.x;
        const uint32_t v = table[cell];
        table[cell] = running;
        running += v;
    }
}

// ---- coarse pass 2: write the block lists ------------------------------------------------------
// Order inside a chunk: per block a 1024-bit mask of the chunk's Gaussians touching it; the rank of
// Gaussian g in block b is the number of set bits below g.
__global__ __launch_bounds__(kCoarse) void coarse_emit_kernel(int n, const uint32_t* __restrict__ sorted_depth,
                                                              const uint32_t* __restrict__ sorted_idx,
                                                              const uint32_t* __restrict__ rect_packed,
                                                              const uint32_t* __restrict__ table, int nbx, int nb, int nbp,
                                                              uint64_t* __restrict__ ent_rd, uint32_t* __restrict__ ent_idx) {
    extern __shared__ uint32_t s_dyn[];
    uint32_t* s_mask = s_dyn;                    // [nb][32]
    uint32_t* s_pre = s_dyn + (size_t)nb * 32;   // [nb][32] set bits in the words below
    const int r = blockIdx.x * kCoarse + threadIdx.x;
    const uint32_t rect = (r < n) ? rect_packed[r] : 0u;
    if (__syncthreads_or(rect != 0u) == 0) return;          // culled tail of the depth order
    for (int i = threadIdx.x; i < nb * 32; i += kCoarse) s_mask[i] = 0;
    __syncthreads();
    const uint32_t x0 = rect & 0xFFu, w = (rect >> 8) & 0xFFu, y0 = (rect >> 16) & 0xFFu, h = rect >> 24;
    uint32_t bx0 = 0, bx1 = 0, by0 = 1, by1 = 0;
    if (rect) { bx0 = x0 / kBW; bx1 = (x0 + w - 1) / kBW; by0 = y0 / kBH; by1 = (y0 + h - 1) / kBH; }
    const uint32_t word = threadIdx.x >> 5, bit = 1u << (threadIdx.x & 31);
    for (uint32_t by = by0; by <= by1; ++by)
        for (uint32_t bx = bx0; bx <= bx1; ++bx) atomicOr(&s_mask[(by * nbx + bx) * 32 + word], bit);
    __syncthreads();
    // 32 consecutive lanes scan the 32 words of one block
    for (int i = threadIdx.x; i < nb * 32; i += kCoarse) {
        const uint32_t c = (uint32_t)__popc(s_mask[i]);
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 32);
            if ((int)(threadIdx.x & 31) >= off) incl += o;
        }
        s_pre[i] = incl - c;
    }
    __syncthreads();
    if (!rect) return;
    const uint64_t rd = (uint64_t)rect | ((uint64_t)sorted_depth[r] << 32);
    const uint32_t idx = sorted_idx[r];
    const uint32_t* row = table + (size_t)blockIdx.x * nbp;
    for (uint32_t by = by0; by <= by1; ++by)
        for (uint32_t bx = bx0; bx <= bx1; ++bx) {
            const uint32_t b = by * nbx + bx;
            const uint32_t pos = row[b] + s_pre[b * 32 + word] + (uint32_t)__popc(s_mask[b * 32 + word] & (bit - 1u));
            ent_rd[pos] = rd;
            ent_idx[pos] = idx;
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

// Keys per (unit, tile of its block). Persistent waves, one unit at a time from a ticket counter.
__global__ __launch_bounds__(256) void unit_count_kernel(BlockMeta meta, int nb, int nbx, const uint64_t* __restrict__ ent_rd,
                                                         uint32_t* __restrict__ cnt) {
    __shared__ uint32_t s_diff[4][96];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t* diff = s_diff[wave];
    const uint32_t total = meta.unit_start()[meta.nbp];
    for (;;) {
        uint32_t u = 0;
        if (lane == 0) u = atomicAdd(&meta.tickets()[0], 1u);
        u = (uint32_t)__builtin_amdgcn_readfirstlane((int)u);
        if (u >= total) break;
        const UnitInfo ui = locate_unit(u, meta, nb, nbx);
        diff[lane] = 0;
        if (lane < 32) diff[64 + lane] = 0;
        __builtin_amdgcn_wave_barrier();
        const uint32_t tx0 = ui.bx * kBW, ty0 = ui.by * kBH;
        for (uint32_t e = ui.e0 + (uint32_t)lane; e < ui.e1; e += kWave) {
            const uint32_t rect = (uint32_t)ent_rd[e];
            const uint32_t x0 = rect & 0xFFu, w = (rect >> 8) & 0xFFu, y0 = (rect >> 16) & 0xFFu, h = rect >> 24;
            const uint32_t cx0 = max(x0, tx0) - tx0, cx1 = min(x0 + w, tx0 + kBW) - tx0;
            const uint32_t cy0 = max(y0, ty0) - ty0, cy1 = min(y0 + h, ty0 + kBH) - ty0;
            atomicAdd(&diff[cy0 * 9 + cx0], 1u);
            atomicSub(&diff[cy0 * 9 + cx1], 1u);
            atomicSub(&diff[cy1 * 9 + cx0], 1u);
            atomicAdd(&diff[cy1 * 9 + cx1], 1u);
        }
        __builtin_amdgcn_wave_barrier();
        // 2-D inclusive prefix over the 8 x 8 cells: along x inside groups of 8 lanes, then along y
        uint32_t v = diff[(lane >> 3) * 9 + (lane & 7)];
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            const uint32_t o = __shfl_up(v, off, 8);
            if ((lane & 7) >= off) v += o;
        }
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(v, off, kWave);
            if (lane >= off) v += o;
        }
        cnt[(size_t)u * 64 + lane] = v;
        __builtin_amdgcn_wave_barrier();
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
#pragma unroll 8
    for (uint32_t u = a; u < z; ++u) {
        const uint32_t v = cnt[(size_t)u * 64 + lane];
        cnt[(size_t)u * 64 + lane] = running;
        running += v;
    }
    const uint32_t tx = (b % (uint32_t)nbx) * kBW + (uint32_t)(lane & 7), ty = (b / (uint32_t)nbx) * kBH + (uint32_t)(lane >> 3);
    if (wave == 0 && tx < (uint32_t)gx && ty < (uint32_t)gy) tile_count[ty * gx + tx] = total;
}

// Exclusive prefix over the tiles in tile order (one workgroup; T <= 65 025).
__global__ __launch_bounds__(1024) void tile_start_kernel(const uint32_t* __restrict__ tile_count, uint32_t tiles,
                                                          uint32_t* __restrict__ tile_start) {
    __shared__ uint32_t s_ws[16];
    const uint32_t per = (tiles + 1023) / 1024;
    const uint32_t a = min(tiles, threadIdx.x * per), z = min(tiles, a + per);
    uint32_t s = 0;
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
    for (uint32_t t = a; t < z; ++t) {
        tile_start[t] = running;
        running += tile_count[t];
    }
    if (threadIdx.x == 1023) tile_start[tiles] = running;
}

// ---- emission ------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) Key2 { uint32_t d0, t0, d1, t1; };   // two consecutive 64-bit keys
struct __attribute__((aligned(8))) Val2 { uint32_t a, b; };

// Stores the staged keys of the tiles in `m` (one bit per lane = tile): the 32-key output line that
// ends at the tile's write position (or holds it, for the final partial line). Four tiles per
// iteration: 16 lanes per tile, two keys per lane — a 16-byte key store and an 8-byte value store.
__device__ __forceinline__ void flush_lines(unsigned long long m, const uint32_t* ring, uint32_t pos, uint32_t base,
                                            uint32_t tile, uint64_t* __restrict__ keys, uint32_t* __restrict__ values) {
    const int lane = threadIdx.x & (kWave - 1);
    const int grp = lane >> 4, sub = lane & 15;
    while (m) {
        int j = 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int jj = m ? (__ffsll((long long)m) - 1) : 64;
            m &= m - 1ull;
            if (grp == g) j = jj;
        }
        const bool valid = j < 64;
        const int src = valid ? j : lane;
        const uint32_t pj = (uint32_t)__shfl((int)pos, src, kWave);
        const uint32_t bj = (uint32_t)__shfl((int)base, src, kWave);
        const uint32_t tj = (uint32_t)__shfl((int)tile, src, kWave);
        const uint32_t g0 = ((pj - 1u) & ~(uint32_t)(kRing - 1)) + 2u * (uint32_t)sub;
        const uint4 q = *reinterpret_cast<const uint4*>(ring + src * kRingStride + 4 * sub);
        const bool v0 = valid && g0 >= bj && g0 < pj;
        const bool v1 = valid && g0 + 1u >= bj && g0 + 1u < pj;
        if (v0 && v1) {
            Key2 k;
            k.d0 = q.x; k.t0 = tj; k.d1 = q.z; k.t1 = tj;
            *reinterpret_cast<Key2*>(keys + g0) = k;
            Val2 v;
            v.a = q.y; v.b = q.w;
            *reinterpret_cast<Val2*>(values + g0) = v;
        } else {
            if (v0) { keys[g0] = ((uint64_t)tj << 32) | q.x; values[g0] = q.y; }
            if (v1) { keys[g0 + 1u] = ((uint64_t)tj << 32) | q.z; values[g0 + 1u] = q.w; }
        }
    }
}

__global__ __launch_bounds__(kWave) void block_emit_kernel(BlockMeta meta, int nb, int nbx, int gx, int gy,
                                                           const uint64_t* __restrict__ ent_rd,
                                                           const uint32_t* __restrict__ ent_idx,
                                                           const uint32_t* __restrict__ cnt,
                                                           const uint32_t* __restrict__ tile_start,
                                                           uint64_t* __restrict__ keys, uint32_t* __restrict__ values) {
    __shared__ __attribute__((aligned(16))) uint32_t ring[kWave * kRingStride];
    const int lane = threadIdx.x;
    const uint32_t total = meta.unit_start()[meta.nbp];
    uint32_t* my_ring = ring + lane * kRingStride;
    for (;;) {
        uint32_t u = 0;
        if (lane == 0) u = atomicAdd(&meta.tickets()[1], 1u);
        u = (uint32_t)__builtin_amdgcn_readfirstlane((int)u);
        if (u >= total) break;
        const UnitInfo ui = locate_unit(u, meta, nb, nbx);
        const uint32_t tx = ui.bx * kBW + (uint32_t)(lane & 7), ty = ui.by * kBH + (uint32_t)(lane >> 3);
        const bool in_grid = tx < (uint32_t)gx && ty < (uint32_t)gy;
        const uint32_t tile = ty * (uint32_t)gx + tx;
        const uint32_t base = in_grid ? tile_start[tile] + cnt[(size_t)u * 64 + lane] : 0u;
        uint32_t pos = base;
        uint32_t e = ui.e0;
        // entries are fetched 64 at a time, one batch ahead of the one being walked
        uint32_t ne = min(64u, ui.e1 - e);
        uint64_t rd = ((uint32_t)lane < ne) ? ent_rd[e + lane] : 0ull;
        uint32_t id = ((uint32_t)lane < ne) ? ent_idx[e + lane] : 0u;
        while (ne) {
            const uint32_t e_next = e + ne;
            const uint32_t ne_next = min(64u, ui.e1 - e_next);
            const uint64_t rd_next = ((uint32_t)lane < ne_next) ? ent_rd[e_next + lane] : 0ull;
            const uint32_t id_next = ((uint32_t)lane < ne_next) ? ent_idx[e_next + lane] : 0u;
            const uint32_t rect_v = (uint32_t)rd, depth_v = (uint32_t)(rd >> 32);
            for (uint32_t j = 0; j < ne; ++j) {
                const uint32_t rect = (uint32_t)__builtin_amdgcn_readlane((int)rect_v, (int)j);
                const uint32_t depth = (uint32_t)__builtin_amdgcn_readlane((int)depth_v, (int)j);
                const uint32_t idx = (uint32_t)__builtin_amdgcn_readlane((int)id, (int)j);
                const uint32_t x0 = rect & 0xFFu, w = (rect >> 8) & 0xFFu, y0 = (rect >> 16) & 0xFFu, h = rect >> 24;
                const bool cov = (tx - x0) < w && (ty - y0) < h;
                bool full = false;
                if (cov) {
                    *reinterpret_cast<uint2*>(my_ring + 2u * (pos & (kRing - 1))) = make_uint2(depth, idx);
                    ++pos;
                    full = (pos & (kRing - 1)) == 0u;
                }
                const unsigned long long fm = __ballot(full);
                if (fm) {
                    __builtin_amdgcn_wave_barrier();
                    flush_lines(fm, ring, pos, base, tile, keys, values);
                    __builtin_amdgcn_wave_barrier();
                }
            }
            e = e_next; ne = ne_next; rd = rd_next; id = id_next;
        }
        // the last, partial line of every tile
        const unsigned long long tm = __ballot((pos & (kRing - 1)) != 0u && pos != base);
        __builtin_amdgcn_wave_barrier();
        if (tm) flush_lines(tm, ring, pos, base, tile, keys, values);
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

bool blockbin_supported(int grid_x, int grid_y) {
    if (grid_x > 255 || grid_y > 255) return false;
    const int nb = ((grid_x + kBW - 1) / kBW) * ((grid_y + kBH - 1) / kBH);
    return nb <= kMaxBlocks;
}

static size_t blockbin_table_bytes(size_t n) { return align128(((n + kCoarse - 1) / kCoarse) * (size_t)kMaxBlocks * 4); }
static size_t blockbin_partial_bytes(size_t n) {
    const size_t chunks = (n + kCoarse - 1) / kCoarse;
    return align128(((chunks + kScanRows - 1) / kScanRows) * (size_t)kMaxBlocks * 4);
}
// per-Gaussian scratch (geometry chunk): chunk table, row-group partials, block meta, tile counts / starts
size_t blockbin_geo_bytes(size_t n) {
    return blockbin_table_bytes(n) + blockbin_partial_bytes(n) + align128((2 * (kMaxBlocks + 1) + 2) * 4) +
           2 * align128((65536 + 1) * 4);
}
// per-instance scratch (binning chunk): keys per (unit, tile)
size_t blockbin_bin_bytes(size_t r) { return align128((r / kUnit + kMaxBlocks + 1) * 64 * 4); }

// rect_packed: out, u32[n] in depth order. ent_rd / ent_idx: R-sized scratch for the block lists.
// ev_*: optional events recorded between the three groups of kernels (stage timing).
int launch_block_binning(int n, const uint32_t* sorted_depth, const uint32_t* sorted_idx, const uint32_t* rect_by_index,
                         int grid_x, int grid_y, uint32_t r_total, uint32_t* rect_packed, char* geo_scratch, uint64_t* ent_rd,
                         uint32_t* ent_idx, char* bin_scratch, uint64_t* keys, uint32_t* values, hipStream_t stream,
                         hipEvent_t ev_coarse_end, hipEvent_t ev_prefix_end) {
    const int nbx = (grid_x + kBW - 1) / kBW, nby = (grid_y + kBH - 1) / kBH, nb = nbx * nby;
    const int nbp = (nb + kWave - 1) / kWave * kWave;
    const uint32_t chunks = (uint32_t)((n + kCoarse - 1) / kCoarse);
    const uint32_t groups = (chunks + kScanRows - 1) / kScanRows;
    const uint32_t tiles = (uint32_t)(grid_x * grid_y);
    char* p = geo_scratch;
    uint32_t* table = reinterpret_cast<uint32_t*>(p); p += blockbin_table_bytes((size_t)n);
    uint32_t* partial = reinterpret_cast<uint32_t*>(p); p += blockbin_partial_bytes((size_t)n);
    BlockMeta meta; meta.w = reinterpret_cast<uint32_t*>(p); meta.nbp = nbp; p += align128((2 * (kMaxBlocks + 1) + 2) * 4);
    uint32_t* tile_count = reinterpret_cast<uint32_t*>(p); p += align128((65536 + 1) * 4);
    uint32_t* tile_start = reinterpret_cast<uint32_t*>(p);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(bin_scratch);

    hipLaunchKernelGGL(coarse_count_kernel, dim3(chunks), dim3(kCoarse), 0, stream, n, sorted_depth, sorted_idx, rect_by_index,
                       nbx, nbp, rect_packed, table);
    GSR_LAUNCH_CHECK("coarse_count_kernel");
    hipLaunchKernelGGL(blockscan_reduce_kernel, dim3(groups), dim3(nbp), 0, stream, table, chunks, nbp, partial);
    GSR_LAUNCH_CHECK("blockscan_reduce_kernel");
    hipLaunchKernelGGL(blockscan_partials_kernel, dim3(1), dim3(nbp), 0, stream, partial, groups, nb, meta);
    GSR_LAUNCH_CHECK("blockscan_partials_kernel");
    hipLaunchKernelGGL(blockscan_apply_kernel, dim3(groups), dim3(nbp), 0, stream, table, chunks, nbp, partial, meta.list_start());
    GSR_LAUNCH_CHECK("blockscan_apply_kernel");
    const size_t mask_bytes = (size_t)nb * 32 * 4 * 2;
    if (mask_bytes > 48 * 1024)
        GSR_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(coarse_emit_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)mask_bytes));
    hipLaunchKernelGGL(coarse_emit_kernel, dim3(chunks), dim3(kCoarse), mask_bytes, stream, n, sorted_depth,
                       sorted_idx, rect_packed, table, nbx, nb, nbp, ent_rd, ent_idx);
    GSR_LAUNCH_CHECK("coarse_emit_kernel");
    if (ev_coarse_end) GSR_HIP_TRY(hipEventRecord(ev_coarse_end, stream));

    // persistent grids: enough waves to fill the chip, never more than there can be units
    const uint32_t max_units = r_total / kUnit + (uint32_t)nb + 1u;
    const uint32_t count_wgs = std::min<uint32_t>((max_units + 3) / 4, 256u * 8u);
    hipLaunchKernelGGL(unit_count_kernel, dim3(count_wgs), dim3(256), 0, stream, meta, nb, nbx, ent_rd, cnt);
    GSR_LAUNCH_CHECK("unit_count_kernel");
    GSR_HIP_TRY(hipMemsetAsync(tile_count, 0, (size_t)tiles * 4, stream));
    hipLaunchKernelGGL(block_prefix_kernel, dim3(nb), dim3(256), 0, stream, meta, nbx, grid_x, grid_y, cnt, tile_count);
    GSR_LAUNCH_CHECK("block_prefix_kernel");
    hipLaunchKernelGGL(tile_start_kernel, dim3(1), dim3(1024), 0, stream, tile_count, tiles, tile_start);
    GSR_LAUNCH_CHECK("tile_start_kernel");
    if (ev_prefix_end) GSR_HIP_TRY(hipEventRecord(ev_prefix_end, stream));

    const uint32_t emit_wgs = std::min<uint32_t>(max_units, 256u * 9u);
    hipLaunchKernelGGL(block_emit_kernel, dim3(emit_wgs), dim3(kWave), 0, stream, meta, nb, nbx, grid_x, grid_y, ent_rd, ent_idx,
                       cnt, tile_start, keys, values);
    GSR_LAUNCH_CHECK("block_emit_kernel");
    return GSR_OK;
}

}  // namespace gsr
