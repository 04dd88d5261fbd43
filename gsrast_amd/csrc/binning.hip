// Key emission and tile-range identification.
//
//  duplicate_kernel  — reference apps/gsrast/gscuda/GSCuda.cu:422-475 (duplicateWithKeys):
//      one (tile << 32 | depth bits, gaussian idx) pair per tile a Gaussian's rectangle
//      covers, rows outer / columns inner. Two differences from the reference, neither
//      visible in the sorted list:
//      (1) Gaussians are walked in DEPTH order (stable, ties by index — the low 32 key bits
//          were sorted once per Gaussian, before duplication, instead of once per key after
//          it), so the emitted list is already ordered by the key's depth half and only the
//          tile half remains to be sorted. The "unsorted" arrays therefore hold the same
//          multiset of pairs as the reference's, in depth order rather than index order.
//      (2) A Gaussian covering more than kOwnLaneMax tiles is expanded by its whole wave, so
//          a full-height rectangle becomes coalesced 512-byte key stores.
//  tile_ranges_kernel — reference GSCuda.cu:504-538 (identifyTileRanges), including the
//      placement of the "last element closes its tile" test inside the else branch.
#include <stdlib.h>

#include "gsr_common.hpp"

namespace gsr {
namespace {

constexpr int kOwnLaneMax = 4;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(hi, max(lo, v)); }

__device__ __forceinline__ void emit(uint64_t* __restrict__ keys, uint32_t* __restrict__ values, uint32_t pos,
                                     uint32_t tile, uint32_t depth_bits, uint32_t idx) {
#ifndef GSR_EMIT_NO_KEYS
    keys[pos] = ((uint64_t)tile << 32) | (uint64_t)depth_bits;
#endif
#ifndef GSR_EMIT_NO_VALUES
    values[pos] = idx;
#endif
}

// Two consecutive rows of one column run in one go: a 16-byte key store and an 8-byte value store.
// The destination is only 8-byte (keys) / 4-byte (values) aligned; gfx950 under HSA runs with
// unaligned vector-memory access enabled, so dword-aligned wide stores are legal.
struct __attribute__((packed, aligned(8))) KeyPair { uint64_t a, b; };
struct __attribute__((packed, aligned(4))) ValPair { uint32_t a, b; };
__device__ __forceinline__ void emit2(uint64_t* __restrict__ keys, uint32_t* __restrict__ values, uint32_t pos,
                                      uint32_t tile, uint32_t tile_step, uint32_t depth_bits, uint32_t idx) {
    KeyPair k;
    k.a = ((uint64_t)tile << 32) | (uint64_t)depth_bits;
    k.b = ((uint64_t)(tile + tile_step) << 32) | (uint64_t)depth_bits;
    *reinterpret_cast<KeyPair*>(keys + pos) = k;
    ValPair v;
    v.a = idx;
    v.b = idx;
    *reinterpret_cast<ValPair*>(values + pos) = v;
}

// Tiles covered by each depth-ordered Gaussian (0 for culled ones, whose depth key is ~0).
__global__ __launch_bounds__(256) void gather_counts_kernel(int n, const uint32_t* __restrict__ sorted_depth,
                                                            const uint32_t* __restrict__ sorted_idx,
                                                            const uint32_t* __restrict__ tiles_touched,
                                                            uint32_t* __restrict__ counts) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    counts[r] = (sorted_depth[r] == 0xFFFFFFFFu) ? 0u : tiles_touched[sorted_idx[r]];
}

// Lane r handles the Gaussian of depth rank r. emit_end is the inclusive scan of the
// depth-ordered tile counts. hist_x / hist_y (may be null) receive, per tile column / tile
// row, the number of keys emitted there: a w x h rectangle adds h to each of its w columns
// and w to each of its h rows, so the digit histograms of the two tile passes cost w + h
// LDS atomics per Gaussian instead of a pass over the w*h keys.
__global__ __launch_bounds__(256) void duplicate_kernel(int n, const uint32_t* __restrict__ sorted_depth,
                                                        const uint32_t* __restrict__ sorted_idx,
                                                        const uint32_t* __restrict__ emit_end,
                                                        const float2* __restrict__ means2D,
                                                        const int32_t* __restrict__ radii,
                                                        const int2* __restrict__ rects, FrameDims d,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ values,
                                                        uint32_t* __restrict__ hist_x, uint32_t* __restrict__ hist_y) {
    __shared__ uint32_t lds_hx[256], lds_hy[256];
    const bool want_hist = hist_x != nullptr;
    if (want_hist) {
        lds_hx[threadIdx.x] = 0;
        lds_hy[threadIdx.x] = 0;
        __syncthreads();
    }
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & (kWave - 1);
    int x0 = 0, y0 = 0, w = 0, cnt = 0;
    uint32_t off = 0, depth_bits = 0, idx = 0;
    if (r < n) {
        depth_bits = sorted_depth[r];
        if (depth_bits != 0xFFFFFFFFu) {
            idx = sorted_idx[r];
            const int rad = radii[idx];
            const float2 p = means2D[idx];
            int ex = rad, ey = rad;
            if (rects) { const int2 e = rects[idx]; ex = e.x; ey = e.y; }
            x0 = clampi((int)((p.x - (float)ex) / 16.0f), 0, d.grid_x);
            y0 = clampi((int)((p.y - (float)ey) / 16.0f), 0, d.grid_y);
            const int x1 = clampi((int)((((p.x + (float)ex) + 16.0f) - 1.0f) / 16.0f), 0, d.grid_x);
            int y1 = clampi((int)((((p.y + (float)ey) + 16.0f) - 1.0f) / 16.0f), 0, d.grid_y);
            y0 = clampi(y0, d.row_begin, d.row_end);
            y1 = clampi(y1, d.row_begin, d.row_end);
            w = x1 - x0;
            cnt = w * (y1 - y0);
            off = emit_end[r] - (uint32_t)cnt;
            if (want_hist && cnt > 0) {
                const uint32_t h = (uint32_t)(y1 - y0);
                for (int x = x0; x < x1; ++x) atomicAdd(&lds_hx[x], h);
                for (int y = y0; y < y1; ++y) atomicAdd(&lds_hy[y], (uint32_t)w);
            }
        }
    }
    // Small rectangles: the owning lane writes them itself.
    if (cnt > 0 && cnt <= kOwnLaneMax) {
        int x = x0, y = y0;
        for (int t = 0; t < cnt; ++t) {
            emit(keys, values, off + (uint32_t)t, (uint32_t)(y * d.grid_x + x), depth_bits, idx);
            if (++x == x0 + w) { x = x0; ++y; }
        }
    }
    // Large rectangles: one at a time, expanded by all 64 lanes of the wave.
    unsigned long long big = __ballot(cnt > kOwnLaneMax);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const int sx0 = __shfl(x0, src, kWave), sy0 = __shfl(y0, src, kWave);
        const int sw = __shfl(w, src, kWave), scnt = __shfl(cnt, src, kWave);
        const uint32_t soff = __shfl(off, src, kWave), sdepth = __shfl(depth_bits, src, kWave);
        const uint32_t sidx = __shfl(idx, src, kWave);
        const float inv_w = 1.0f / (float)sw;
        for (int t = lane; t < scnt; t += kWave) {
            int q = (int)((float)t * inv_w);          // t < 2^24: off by at most one
            int rem = t - q * sw;
            if (rem < 0) { --q; rem += sw; }
            if (rem >= sw) { ++q; rem -= sw; }
            emit(keys, values, soff + (uint32_t)t, (uint32_t)((sy0 + q) * d.grid_x + sx0 + rem), sdepth, sidx);
        }
    }
    if (want_hist) {
        __syncthreads();
        if (lds_hx[threadIdx.x]) atomicAdd(&hist_x[threadIdx.x], lds_hx[threadIdx.x]);
        if (lds_hy[threadIdx.x]) atomicAdd(&hist_y[threadIdx.x], lds_hy[threadIdx.x]);
    }
}

// ---- column-major emission (tile grids up to 255 x 255) -----------------------------------
// The first of the two tile passes (stable sort on the tile column x) is not run as a sort at
// all: its result is written directly. In depth order, the keys of column x are, Gaussian after
// Gaussian, the h rows of the rectangle — so key (g, x, y) lands at
//     start[x] + sum of h over earlier Gaussians covering x + (y - y0).
// column_count_kernel gets the per-workgroup (256 depth-consecutive Gaussians) column sums, one
// scan over the column-major table turns them into start offsets, emit_columns_kernel resolves
// the order inside a workgroup in LDS and writes every (Gaussian, column) run as one burst.
// What reaches the remaining pass (stable on the tile row y) is exactly what a stable x pass
// over the depth-ordered list would have produced.

// rect packed as x0 | w << 8 | y0 << 16 | h << 24 (all < 256); 0 = culled / empty
__global__ __launch_bounds__(256) void column_count_kernel(int n, const uint32_t* __restrict__ sorted_depth,
                                                           const uint32_t* __restrict__ sorted_idx,
                                                           const uint32_t* __restrict__ rect_by_index, FrameDims d,
                                                           uint32_t* __restrict__ rect_packed,
                                                           uint32_t* __restrict__ col_table, uint32_t num_blocks,
                                                           uint32_t* __restrict__ hist_y) {
    // Difference arrays: a w x h rectangle adds h at column x0 and takes it back at x0 + w (rows
    // likewise), four LDS atomics per Gaussian; one block-wide prefix sum per array then gives the
    // per-column / per-row key counts (u32 wrap-around keeps the sums exact).
    __shared__ uint32_t lds_hx[257], lds_hy[257];
    __shared__ uint32_t s_ws[2][4];
    lds_hx[threadIdx.x] = 0;
    lds_hy[threadIdx.x] = 0;
    if (threadIdx.x == 0) lds_hx[256] = lds_hy[256] = 0;
    __syncthreads();
    const int r = blockIdx.x * 256 + threadIdx.x;
    uint32_t packed = 0;
    // one 4-byte gather per Gaussian: preprocess left the band-clipped rectangle in index order
    if (r < n && sorted_depth[r] != 0xFFFFFFFFu) packed = rect_by_index[sorted_idx[r]];
    if (packed) {
        const uint32_t x0 = packed & 0xFFu, w = (packed >> 8) & 0xFFu, y0 = (packed >> 16) & 0xFFu, h = packed >> 24;
        atomicAdd(&lds_hx[x0], h);
        atomicSub(&lds_hx[x0 + w], h);
        atomicAdd(&lds_hy[y0], w);
        atomicSub(&lds_hy[y0 + h], w);
    }
    if (r < n) rect_packed[r] = packed;
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t ix = lds_hx[threadIdx.x], iy = lds_hy[threadIdx.x];
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t ox = __shfl_up(ix, off, kWave), oy = __shfl_up(iy, off, kWave);
        if (lane >= off) { ix += ox; iy += oy; }
    }
    if (lane == kWave - 1) { s_ws[0][wave] = ix; s_ws[1][wave] = iy; }
    __syncthreads();
    for (int ww = 0; ww < wave; ++ww) { ix += s_ws[0][ww]; iy += s_ws[1][ww]; }
    if ((int)threadIdx.x < d.grid_x) col_table[(size_t)threadIdx.x * num_blocks + blockIdx.x] = ix;
    if (iy) atomicAdd(&hist_y[threadIdx.x], iy);
}

#ifndef GSR_EMIT_COLS
#define GSR_EMIT_COLS 16
#endif
constexpr int kColsPerBlock = GSR_EMIT_COLS;  // tile columns one workgroup writes (its open output streams)
#ifndef GSR_EMIT_SMALL
#define GSR_EMIT_SMALL 16
#endif
constexpr int kSmallRect = GSR_EMIT_SMALL;     // rectangle parts up to this many tiles are written by one lane

// Workgroup (b, r): the 256 depth-consecutive Gaussians of chunk b, tile columns
// [16 r, 16 r + 16). One lane per Gaussian. For each of the 16 columns a block-wide exclusive
// prefix of "rows of the Gaussians covering it" gives every (Gaussian, column) run its place
// behind the chunk's start for that column; the runs are then written wave-cooperatively.
// Keeping a workgroup to 16 columns bounds the number of output streams it appends to (2 x 16
// partially written cache lines), which is what the L2 needs to turn the many short runs into
// full-line writes.
__global__ __launch_bounds__(256) void emit_columns_kernel(int n, const uint32_t* __restrict__ sorted_depth,
                                                           const uint32_t* __restrict__ sorted_idx,
                                                           const uint32_t* __restrict__ rect_packed,
                                                           const uint32_t* __restrict__ col_table_incl,
                                                           uint32_t num_blocks, int grid_x,
                                                           uint64_t* __restrict__ keys, uint32_t* __restrict__ values) {
    __shared__ uint32_t s_off[256][kColsPerBlock + 1];   // +1: odd stride, conflict-free column walks
    __shared__ uint32_t s_wsum[4][kColsPerBlock];
    __shared__ uint32_t s_colbase[kColsPerBlock];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int x_lo = blockIdx.y * kColsPerBlock;
    const uint32_t rect = (r < n) ? rect_packed[r] : 0u;
    const int x0 = (int)(rect & 0xFFu), w = (int)((rect >> 8) & 0xFFu);
    const uint32_t y0 = (rect >> 16) & 0xFFu, h = rect >> 24;
    // part of the rectangle inside this workgroup's columns
    const int cx0 = max(x0, x_lo), cx1 = min(x0 + w, min(x_lo + kColsPerBlock, grid_x));
    const int wr = max(0, cx1 - cx0);
    if (__syncthreads_or(wr > 0) == 0) return;           // nothing of this chunk in these columns
    if ((int)threadIdx.x < kColsPerBlock) {
        const int x = x_lo + (int)threadIdx.x;
        uint32_t base = 0;
        if (x < grid_x) {
            const size_t cell = (size_t)x * num_blocks + blockIdx.x;
            base = cell ? col_table_incl[cell - 1] : 0u;
        }
        s_colbase[threadIdx.x] = base;
    }
    // per column: exclusive prefix over the 256 Gaussians of the rows they contribute
    uint32_t pre[kColsPerBlock];
#pragma unroll
    for (int j = 0; j < kColsPerBlock; ++j) {
        const int x = x_lo + j;
        const uint32_t v = (x >= cx0 && x < cx1) ? h : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += o;
        }
        if (lane == kWave - 1) s_wsum[wave][j] = incl;
        pre[j] = incl - v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kColsPerBlock; ++j) {
        uint32_t wbase = s_colbase[j];
        for (int ww = 0; ww < wave; ++ww) wbase += s_wsum[ww][j];
        s_off[threadIdx.x][j] = wbase + pre[j];
    }
    // (s_off rows are read only by the wave that wrote them: no barrier needed)
    const uint32_t depth = (r < n) ? sorted_depth[r] : 0u;
    const uint32_t idx = (wr > 0) ? sorted_idx[r] : 0u;
    const uint32_t cnt = (uint32_t)wr * h;
    // small parts: the owning lane walks its few tiles itself
    if (cnt > 0 && cnt <= (uint32_t)kSmallRect) {
        uint32_t c = 0, yy = 0, base = s_off[threadIdx.x][cx0 - x_lo];
        for (uint32_t k = 0; k < cnt; ++k) {
            emit(keys, values, base + yy, __umul24(y0 + yy, (uint32_t)grid_x) + (uint32_t)cx0 + c, depth, idx);
            if (++yy == h) { yy = 0; ++c; if ((int)c < wr) base = s_off[threadIdx.x][cx0 - x_lo + (int)c]; }
        }
    }
    // large parts: one at a time, the 64 lanes walk it column-major (k -> column k / h, row k % h)
    unsigned long long big = __ballot(cnt > (uint32_t)kSmallRect);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t scx0 = (uint32_t)__shfl(cx0, src, kWave), sy0 = __shfl(y0, src, kWave), sh = __shfl(h, src, kWave);
        const uint32_t scnt = __shfl(cnt, src, kWave), sdepth = __shfl(depth, src, kWave), sidx = __shfl(idx, src, kWave);
        const uint32_t* off_row = &s_off[(wave << 6) + src][scx0 - (uint32_t)x_lo];
        // lanes take PAIRS of rows: pair k -> column k / hp, rows 2 (k % hp) and 2 (k % hp) + 1
        const uint32_t hp = (sh + 1u) >> 1;
        const uint32_t sw = scnt / sh, npairs = sw * hp;
        const float inv_hp = 1.0f / (float)hp;
        for (uint32_t k = (uint32_t)lane; k < npairs; k += kWave) {
            uint32_t c = (uint32_t)((float)k * inv_hp);             // k < 2^16: off by at most one
            int j = (int)k - (int)__umul24(c, hp);
            if (j < 0) { --c; j += (int)hp; }
            if (j >= (int)hp) { ++c; j -= (int)hp; }
            const uint32_t yy = 2u * (uint32_t)j;
            const uint32_t pos = off_row[c] + yy;
            const uint32_t tile = __umul24(sy0 + yy, (uint32_t)grid_x) + scx0 + c;
            if (yy + 1u < sh) emit2(keys, values, pos, tile, (uint32_t)grid_x, sdepth, sidx);
            else emit(keys, values, pos, tile, sdepth, sidx);
        }
    }
}

__global__ __launch_bounds__(256) void tile_ranges_kernel(const uint64_t* __restrict__ keys, size_t n,
                                                          uint2* __restrict__ ranges) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const uint32_t cur = (uint32_t)(keys[idx] >> 32);
    if (idx == 0) {
        ranges[cur].x = 0;
    } else {
        const uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
        if (prev != cur) {
            ranges[prev].y = (uint32_t)idx;
            ranges[cur].x = (uint32_t)idx;
        }
        if (idx == n - 1) ranges[cur].y = (uint32_t)n;
    }
}

// Same result as tile_ranges_kernel without streaming the R keys: thread t finds the first sorted
// key of tile t and of tile t + 1 by two interleaved binary searches (2 x ~log2 R dependent loads).
// Tiles that own no key keep (0, 0) and the R == 1 quirk (the lone tile is never closed) is kept.
__global__ __launch_bounds__(256) void tile_ranges_search_kernel(const uint64_t* __restrict__ keys, uint32_t n,
                                                                 uint2* __restrict__ ranges, uint32_t num_tiles) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= num_tiles) return;
    uint32_t lo_a = 0, hi_a = n, lo_b = 0, hi_b = n;      // lower bounds of tile t (a) and t + 1 (b)
    while (lo_a < hi_a || lo_b < hi_b) {
        const uint32_t mid_a = lo_a + ((hi_a - lo_a) >> 1), mid_b = lo_b + ((hi_b - lo_b) >> 1);
        const uint32_t ka = (lo_a < hi_a) ? (uint32_t)(keys[mid_a] >> 32) : 0u;
        const uint32_t kb = (lo_b < hi_b) ? (uint32_t)(keys[mid_b] >> 32) : 0u;
        if (lo_a < hi_a) { if (ka < t) lo_a = mid_a + 1; else hi_a = mid_a; }
        if (lo_b < hi_b) { if (kb < t + 1) lo_b = mid_b + 1; else hi_b = mid_b; }
    }
    uint2 r = make_uint2(0u, 0u);
    if (lo_b > lo_a && n > 1) r = make_uint2(lo_a, lo_b);
    ranges[t] = r;
}

}  // namespace

int launch_gather_counts(int n, const uint32_t* sorted_depth, const uint32_t* sorted_idx, const uint32_t* tiles_touched,
                         uint32_t* counts, hipStream_t stream) {
    hipLaunchKernelGGL(gather_counts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, sorted_depth,
                       sorted_idx, tiles_touched, counts);
    GSR_LAUNCH_CHECK("gather_counts_kernel");
    return GSR_OK;
}

int launch_duplicate(int n, const uint32_t* sorted_depth, const uint32_t* sorted_idx, const uint32_t* emit_end,
                     const gsr_geometry_state& g, const int32_t* radii, const int32_t* rects, const FrameDims& d,
                     uint64_t* keys, uint32_t* values, uint32_t* hist_x, uint32_t* hist_y, hipStream_t stream) {
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(duplicate_kernel, dim3(blocks), dim3(256), 0, stream, n, sorted_depth, sorted_idx, emit_end,
                       reinterpret_cast<const float2*>(g.means2D), radii, reinterpret_cast<const int2*>(rects), d, keys,
                       values, hist_x, hist_y);
    GSR_LAUNCH_CHECK("duplicate_kernel");
    return GSR_OK;
}

int launch_column_count(int n, const uint32_t* sorted_depth, const uint32_t* sorted_idx, const uint32_t* rect_by_index,
                        const FrameDims& d, uint32_t* rect_packed, uint32_t* col_table, uint32_t* hist_y, hipStream_t stream) {
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(column_count_kernel, dim3(blocks), dim3(256), 0, stream, n, sorted_depth, sorted_idx, rect_by_index, d,
                       rect_packed, col_table, blocks, hist_y);
    GSR_LAUNCH_CHECK("column_count_kernel");
    return GSR_OK;
}

int launch_emit_columns(int n, const uint32_t* sorted_depth, const uint32_t* sorted_idx, const uint32_t* rect_packed,
                        const uint32_t* col_table_incl, int grid_x, uint64_t* keys, uint32_t* values, hipStream_t stream) {
    const unsigned blocks = (unsigned)((n + 255) / 256);
    // Tuning aid: GSR_EMIT_DYN_LDS=<bytes> of unused dynamic LDS lowers the workgroups per CU.
    static const unsigned dyn_lds = [] { const char* e = getenv("GSR_EMIT_DYN_LDS"); return e ? (unsigned)atoi(e) : 0u; }();
    hipLaunchKernelGGL(emit_columns_kernel, dim3(blocks, (unsigned)((grid_x + kColsPerBlock - 1) / kColsPerBlock)), dim3(256), dyn_lds, stream, n, sorted_depth, sorted_idx, rect_packed,
                       col_table_incl, blocks, grid_x, keys, values);
    GSR_LAUNCH_CHECK("emit_columns_kernel");
    return GSR_OK;
}

int launch_tile_ranges(const uint64_t* keys, size_t n, uint32_t* ranges, int num_tiles, hipStream_t stream) {
    if (n == 0) {
        GSR_HIP_TRY(hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)num_tiles, stream));
        return GSR_OK;
    }
    // One thread per tile beats one thread per key as soon as there are more keys than a few per tile.
    if (n >= (size_t)num_tiles * 8 && n < 0xFFFFFFFFull) {
        hipLaunchKernelGGL(tile_ranges_search_kernel, dim3((unsigned)((num_tiles + 255) / 256)), dim3(256), 0, stream, keys,
                           (uint32_t)n, reinterpret_cast<uint2*>(ranges), (uint32_t)num_tiles);
        GSR_LAUNCH_CHECK("tile_ranges_search_kernel");
        return GSR_OK;
    }
    GSR_HIP_TRY(hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)num_tiles, stream));
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(blocks), dim3(256), 0, stream, keys, n,
                       reinterpret_cast<uint2*>(ranges));
    GSR_LAUNCH_CHECK("tile_ranges_kernel");
    return GSR_OK;
}

}  // namespace gsr
