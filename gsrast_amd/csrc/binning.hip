// Key emission and tile-range identification.
//
//  duplicate_kernel  — reference apps/gsrast/gscuda/GSCuda.cu:422-475 (duplicateWithKeys):
//      one (tile << 32 | depth bits, gaussian idx) pair per tile a Gaussian's rectangle
//      covers, rows outer / columns inner, at offsets taken from the inclusive scan.
//      The reference walks each rectangle with one thread; here a Gaussian that covers
//      more than kOwnLaneMax tiles is expanded by its whole wave, so a full-height
//      rectangle becomes coalesced 512-byte key stores instead of one lane's serial loop.
//  tile_ranges_kernel — reference GSCuda.cu:504-538 (identifyTileRanges), including the
//      placement of the "last element closes its tile" test inside the else branch.
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

__global__ __launch_bounds__(256) void duplicate_kernel(int n, const float2* __restrict__ means2D,
                                                        const float* __restrict__ depths,
                                                        const uint32_t* __restrict__ offsets,
                                                        const int32_t* __restrict__ radii,
                                                        const int2* __restrict__ rects, FrameDims d,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ values) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & (kWave - 1);
    int x0 = 0, y0 = 0, w = 0, cnt = 0;
    uint32_t off = 0, depth_bits = 0;
    if (idx < n) {
        const int r = radii[idx];
        if (r > 0) {
            const float2 p = means2D[idx];
            int ex = r, ey = r;
            if (rects) { const int2 e = rects[idx]; ex = e.x; ey = e.y; }
            x0 = clampi((int)((p.x - (float)ex) / 16.0f), 0, d.grid_x);
            y0 = clampi((int)((p.y - (float)ey) / 16.0f), 0, d.grid_y);
            const int x1 = clampi((int)((((p.x + (float)ex) + 16.0f) - 1.0f) / 16.0f), 0, d.grid_x);
            int y1 = clampi((int)((((p.y + (float)ey) + 16.0f) - 1.0f) / 16.0f), 0, d.grid_y);
            y0 = clampi(y0, d.row_begin, d.row_end);
            y1 = clampi(y1, d.row_begin, d.row_end);
            w = x1 - x0;
            cnt = w * (y1 - y0);
            off = (idx == 0) ? 0u : offsets[idx - 1];
            depth_bits = __float_as_uint(depths[idx]);
        }
    }
    // Small rectangles: the owning lane writes them itself.
    if (cnt > 0 && cnt <= kOwnLaneMax) {
        int x = x0, y = y0;
        for (int t = 0; t < cnt; ++t) {
            emit(keys, values, off + (uint32_t)t, (uint32_t)(y * d.grid_x + x), depth_bits, (uint32_t)idx);
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
        const uint32_t sidx = (uint32_t)(idx - lane + src);
        const float inv_w = 1.0f / (float)sw;
        for (int t = lane; t < scnt; t += kWave) {
            int q = (int)((float)t * inv_w);          // t < 2^24: off by at most one
            int r = t - q * sw;
            if (r < 0) { --q; r += sw; }
            if (r >= sw) { ++q; r -= sw; }
            emit(keys, values, soff + (uint32_t)t, (uint32_t)((sy0 + q) * d.grid_x + sx0 + r), sdepth, sidx);
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

}  // namespace

int launch_duplicate(int n, const gsr_geometry_state& g, const int32_t* radii, const int32_t* rects,
                     const FrameDims& d, uint64_t* keys, uint32_t* values, hipStream_t stream) {
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(duplicate_kernel, dim3(blocks), dim3(256), 0, stream, n,
                       reinterpret_cast<const float2*>(g.means2D), g.depths, g.point_offsets, radii,
                       reinterpret_cast<const int2*>(rects), d, keys, values);
    GSR_LAUNCH_CHECK("duplicate_kernel");
    return GSR_OK;
}

int launch_tile_ranges(const uint64_t* keys, size_t n, uint32_t* ranges, int num_tiles, hipStream_t stream) {
    GSR_HIP_TRY(hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)num_tiles, stream));
    if (n == 0) return GSR_OK;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(blocks), dim3(256), 0, stream, keys, n,
                       reinterpret_cast<uint2*>(ranges));
    GSR_LAUNCH_CHECK("tile_ranges_kernel");
    return GSR_OK;
}

}  // namespace gsr
