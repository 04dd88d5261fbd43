// Shared core of the per-tile compositing kernels (blend.hip: records come from the sorted point list;
// blockbin.hip: records come straight from the block lists). One wavefront per 16 x 16 tile, four pixels
// per lane: lane l owns pixels (x = l & 15, y = (l >> 4) + 4 k), k = 0..3.
//
// Semantics follow reference apps/gsrast/gscuda/GSCuda.cu:623-676 (the loop of renderCUDA); see blend.hip.
#pragma once
#include "gsr_common.hpp"

namespace gsr {

constexpr int kBatch = 256;

// exp(power) >= 1/255 needs power >= -ln(255) = -5.5413; anything below -5.56 fails the
// alpha >= 1/255 test for every opacity <= 1 with a 1.9 % margin, far outside rounding.
constexpr float kPowerFloor = -5.56f;

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct TileLanes {                 // per-lane state of the four pixels a lane owns
    int px, py0;
    float fx;
    f32x2 fy01, fy23;
    bool inside[4];
    uint32_t done[4];              // 0 / 1, kept in vector registers (as lane masks they cost scalar merges at every branch join)
    float T[4], cr[4], cg[4], cb[4];
    uint32_t last[4];
};

// only_strip >= 0: the wave composites just that 16 x 4 strip of the tile (the other three pixel slots start
// out finished and are never written) — four waves per tile, used when there are few tiles to go round.
__device__ __forceinline__ void tile_lanes_init(TileLanes& s, int tx, int ty, int lane, int width, int height,
                                                int only_strip = -1) {
    s.px = tx * kTile + (lane & 15);
    s.py0 = ty * kTile + (lane >> 4);
    s.fx = (float)s.px;
    s.fy01.x = (float)s.py0; s.fy01.y = (float)(s.py0 + 4); s.fy23.x = (float)(s.py0 + 8); s.fy23.y = (float)(s.py0 + 12);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        s.inside[k] = s.px < width && (s.py0 + 4 * k) < height && (only_strip < 0 || only_strip == k);
        s.done[k] = s.inside[k] ? 0u : 1u;
        s.T[k] = 1.0f; s.cr[k] = s.cg[k] = s.cb[k] = 0.0f; s.last[k] = 0;
    }
}

__device__ __forceinline__ bool tile_lanes_all_done(const TileLanes& s) {
    return __ballot((s.done[0] & s.done[1] & s.done[2] & s.done[3]) == 0u) == 0ull;
}

// Composites the `chunk` records staged in wave-private LDS (record j is the tile's contributor number
// first_contributor + j, 1-based). dx and the terms that only depend on it are computed once per record
// and lane; the per-row terms run as packed f32 pairs (v_pk_mul_f32 / v_pk_add_f32: two IEEE single
// operations per issue, same rounding as the scalar form, no fused multiply-add). Returns true as soon as
// every pixel of the tile is finished.
__device__ __forceinline__ bool composite_staged(TileLanes& s, const float2* s_xy, const float4* s_co, const float4* s_rgb,
                                                 uint32_t chunk, uint32_t first_contributor, float t_cutoff) {
    for (uint32_t j = 0; j < chunk; ++j) {
        const float2 xy = s_xy[j];
        const float4 co = s_co[j];
        const float dx = xy.x - s.fx;
        const float adx = co.x * dx;
        const float t1 = adx * dx;
        const float bdx = co.y * dx;
        f32x2 gy; gy.x = xy.y; gy.y = xy.y;
        const f32x2 dy01 = gy - s.fy01, dy23 = gy - s.fy23;
        const f32x2 pw01 = -0.5f * (t1 + (co.z * dy01) * dy01) - bdx * dy01;
        const f32x2 pw23 = -0.5f * (t1 + (co.z * dy23) * dy23) - bdx * dy23;
        const float power[4] = {pw01.x, pw01.y, pw23.x, pw23.y};
        const bool wide = co.w > 1.0f;       // wave-uniform: the floor only holds for opacity <= 1
        bool cand[4];
        bool any_cand = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            cand[k] = s.done[k] == 0u && !(power[k] > 0.0f) && (power[k] >= kPowerFloor || wide);
            any_cand = any_cand || cand[k];
        }
        if (__ballot(any_cand) == 0ull) continue;
        const float4 col = s_rgb[j];
        const uint32_t contributor = first_contributor + j;
        bool newly_done = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (__ballot(cand[k]) == 0ull) continue;      // nobody in this strip sees the record
            const float alpha = fminf(0.99f, co.w * __expf(power[k]));
            const bool live = cand[k] && !(alpha < 1.0f / 255.0f);
            const float test = s.T[k] * (1.0f - alpha);
            const bool stop = live && test < t_cutoff;
            if (live && !stop) {
                s.cr[k] += col.x * alpha * s.T[k];
                s.cg[k] += col.y * alpha * s.T[k];
                s.cb[k] += col.z * alpha * s.T[k];
                s.T[k] = test;
                s.last[k] = contributor;
            }
            s.done[k] |= stop ? 1u : 0u;
            newly_done = newly_done || stop;
        }
        if (__ballot(newly_done) != 0ull && tile_lanes_all_done(s)) return true;
    }
    return false;
}

__device__ __forceinline__ void tile_lanes_write(const TileLanes& s, int width, int height, const float* __restrict__ background,
                                                 float* __restrict__ final_t, uint32_t* __restrict__ n_contrib,
                                                 float* __restrict__ out_color) {
    const size_t plane = (size_t)width * (size_t)height;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (s.inside[k]) {
            const size_t pid = (size_t)(s.py0 + 4 * k) * (size_t)width + (size_t)s.px;
            final_t[pid] = s.T[k];
            n_contrib[pid] = s.last[k];
            out_color[pid] = s.cr[k] + s.T[k] * background[0];
            out_color[pid + plane] = s.cg[k] + s.T[k] * background[1];
            out_color[pid + 2 * plane] = s.cb[k] + s.T[k] * background[2];
        }
    }
}

// Blocks b and b+8 share an XCD (and its L2). Give every XCD one contiguous run of tiles so
// the Gaussians neighbouring tiles share are gathered through one L2.
__device__ __forceinline__ int xcd_tile_of_block(int b, int n) {
    const int q = n / 8, r = n % 8, x = b % 8, k = b / 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

}  // namespace gsr
