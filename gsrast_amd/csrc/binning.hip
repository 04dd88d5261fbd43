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
//  tile_ranges_search_kernel — reference GSCuda.cu:504-538 (identifyTileRanges), including the
//      effect of the "last element closes its tile" test sitting inside the else branch.
#include <stdlib.h>

#include "gsr_common.hpp"

namespace gsr {
namespace {

constexpr int kOwnLaneMax = 4;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(hi, max(lo, v)); }

__device__ __forceinline__ void emit(uint64_t* __restrict__ keys, uint32_t* __restrict__ values, uint32_t pos,
                                     uint32_t tile, uint32_t depth_bits, uint32_t idx) {
    keys[pos] = ((uint64_t)tile << 32) | (uint64_t)depth_bits;
    values[pos] = idx;
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

// identifyTileRanges (GSCuda.cu:504-538) without streaming the R keys: thread t finds the first sorted
// key of tile t and of tile t + 1 by two interleaved binary searches (2 x ~log2 R dependent loads).
// Tiles that own no key get (0, 0) (the reference's memset) and the R == 1 quirk is kept: the lone tile is
// never closed there, because the "last element closes its tile" test sits inside the else branch.
// nonempty (may be null; zeroed by the caller's stream before this launch): += the tiles that got a list — the blend
// decides from it whether the frame's tiles can fill the chip with one wave each (blend.hip).
__global__ __launch_bounds__(256) void tile_ranges_search_kernel(const uint64_t* __restrict__ keys, uint32_t n,
                                                                 uint2* __restrict__ ranges, uint32_t num_tiles,
                                                                 bool close_single, uint32_t* __restrict__ nonempty) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const bool live = t < num_tiles;
    uint32_t lo_a = 0, hi_a = live ? n : 0u, lo_b = 0, hi_b = live ? n : 0u;      // lower bounds of tile t (a) and t + 1 (b)
    while (lo_a < hi_a || lo_b < hi_b) {
        const uint32_t mid_a = lo_a + ((hi_a - lo_a) >> 1), mid_b = lo_b + ((hi_b - lo_b) >> 1);
        const uint32_t ka = (lo_a < hi_a) ? (uint32_t)(keys[mid_a] >> 32) : 0u;
        const uint32_t kb = (lo_b < hi_b) ? (uint32_t)(keys[mid_b] >> 32) : 0u;
        if (lo_a < hi_a) { if (ka < t) lo_a = mid_a + 1; else hi_a = mid_a; }
        if (lo_b < hi_b) { if (kb < t + 1) lo_b = mid_b + 1; else hi_b = mid_b; }
    }
    uint2 r = make_uint2(0u, 0u);
    if (live && lo_b > lo_a && (n > 1 || close_single)) r = make_uint2(lo_a, lo_b);
    if (live) ranges[t] = r;
    if (nonempty) {
        const unsigned long long m = __ballot(r.y > r.x);
        if ((threadIdx.x & (kWave - 1)) == 0 && m) atomicAdd(nonempty, (uint32_t)__popcll(m));
    }
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

int launch_tile_ranges(const uint64_t* keys, size_t n, uint32_t* ranges, int num_tiles, bool close_single, hipStream_t stream,
                       uint32_t* nonempty) {
    if (n == 0) {
        GSR_HIP_TRY(hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)num_tiles, stream));
        return GSR_OK;
    }
    if (n >= 0xFFFFFFFFull) return GSR_ERR_TOO_LARGE;
    hipLaunchKernelGGL(tile_ranges_search_kernel, dim3((unsigned)((num_tiles + 255) / 256)), dim3(256), 0, stream, keys,
                       (uint32_t)n, reinterpret_cast<uint2*>(ranges), (uint32_t)num_tiles, close_single, nonempty);
    GSR_LAUNCH_CHECK("tile_ranges_search_kernel");
    return GSR_OK;
}

}  // namespace gsr
