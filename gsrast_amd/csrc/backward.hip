// Backward pass of the forward rasterizer (BASELINE config 5): gradients of
// L = sum(dL_dout * out_color) w.r.t. the per-Gaussian 2-D quantities the blend loop reads, then
// through conic = inverse(cov2D) and cov2D = J W Sigma W^T J^T to the 3-D covariance, and through
// colour = 0.5 + 0.4 DC to the DC harmonics.
//
// The reference has no backward pass (apps/gsrast/gscuda has none; the upstream submodule that has
// one is empty), so this differentiates THIS library's forward, which restates the reference's:
//   blend loop          apps/gsrast/gscuda/GSCuda.cu:623-676
//   computeCov2D, conic GSCuda.cu:197-231, :329-335        colour  GSCuda.cu:362-366
// Parity unpinned; oracle/backward_np.py (float64, checked against finite differences) is the checker.
//
// render_backward_kernel: one wavefront per 16 x 16 tile, four pixels per lane (the layout of
// blend_wave_kernel). Every pixel walks its tile's list BACK TO FRONT from its own last contributor
// (nContrib) starting at its final transmittance, re-deriving T_i = T_{i+1} / (1 - alpha_i); the nine
// partial sums of a record are reduced over the wave with DPP row shifts / broadcasts and leave as
// nine float atomics per (record, tile) — only records that some pixel of the tile composited.
#include <hip/hip_runtime.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "blend_core.hpp"
#include "blockbin.hpp"

namespace gsr {
namespace {

struct RenderBackwardParams {
    const uint2* ranges;
    const uint32_t* point_list;      // the sorted list of Gaussian indices, or null: the tile lists come from `feed`
    BlockFeed feed;                  // (block plan with GSR_FLAG_NO_SORTED_LISTS: the lists as the forward blend read them)
    const float2* means2D;
    const float* colors;
    const float4* conic_opacity;
    const float* final_t;
    const uint32_t* n_contrib;
    const float* background;
    const float* dL_dout;
    float* dL_dmean2D;          // vec2[N]
    float* dL_dconic_opacity;   // vec4[N]: dA, dB, dC, dopacity
    float* dL_dcolors;          // vec3[N]
    float* dL_dcov2D;           // vec4[N]: (m00, m01, m11, 0), the gradient w.r.t. the full symmetric 2-D covariance; or null
    double* sums64;             // f64[12 N] or null: the twelve sums of every Gaussian, accumulated in double instead of in the four arrays above
    FrameDims dims;
    int num_tiles;
    const uint32_t* tile_order;  // the forward blend's workgroup order (slow tiles first), or null: patch order
};

// Sums of TWELVE per-lane values over the 64 lanes in about half the steps of twelve separate reductions: v_permlane32_swap
// and v_permlane16_swap (gfx950) fold the lanes while packing the values side by side — after the first fold a register
// holds value 2i in its lower 32 lanes and value 2i + 1 in its upper ones, after the second a row of 16 lanes per value —
// then, for the first eight, one select + row_ror:8 packs two registers into one and three quad / half-row steps finish; the
// last four (one register after their folds) take four steps inside their rows. On return lane 0 / 32 / 16 / 48 / 8 / 40 /
// 24 / 56 holds the total of value 0 / 1 / ... / 7 and lane 4 / 36 / 20 / 52 that of value 8 / 9 / 10 / 11.
__device__ __forceinline__ float wave_sum12(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7, float a8,
                                            float a9, float a10, float a11) {
    const int lane = threadIdx.x & (kWave - 1);
    auto swap32 = [](float& x, float& y) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
        x = __uint_as_float(r[0]); y = __uint_as_float(r[1]);
    };
    auto swap16 = [](float& x, float& y) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
        x = __uint_as_float(r[0]); y = __uint_as_float(r[1]);
    };
#define GSR_DPP(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, false))
    swap32(a0, a1); float p = a0 + a1;           // rows 0, 1: value 0 (lanes l and l + 32 added); rows 2, 3: value 1
    swap32(a2, a3); float q = a2 + a3;
    swap32(a4, a5); float r = a4 + a5;
    swap32(a6, a7); float t = a6 + a7;
    swap16(p, q); const float t0 = p + q;        // rows: value 0, 2, 1, 3, each added over the four 16-lane groups
    swap16(r, t); const float t1 = r + t;        // rows: value 4, 6, 5, 7
    const bool upper = (lane & 8) != 0;
    float u = (upper ? t1 : t0) + GSR_DPP(upper ? t0 : t1, 0x128);      // row_ror:8: lanes 0-7 of a row now carry t0, lanes 8-15 t1
    u += GSR_DPP(u, 0xB1);                       // quad_perm [1,0,3,2]
    u += GSR_DPP(u, 0x4E);                       // quad_perm [2,3,0,1]
    u += GSR_DPP(u, 0x141);                      // row_half_mirror
    swap32(a8, a9); float pp = a8 + a9;          // rows 0, 1: value 8; rows 2, 3: value 9
    swap32(a10, a11); float qq = a10 + a11;      // rows 0, 1: value 10; rows 2, 3: value 11
    swap16(pp, qq); float w = pp + qq;           // rows: value 8, 10, 9, 11
    w += GSR_DPP(w, 0x128);                      // row_ror:8
    w += GSR_DPP(w, 0xB1);
    w += GSR_DPP(w, 0x4E);
    w += GSR_DPP(w, 0x141);                      // every lane of a row: the row's total
#undef GSR_DPP
    return (lane & 7) == 4 ? w : u;
}

// Per-entry sums (block feed). A Gaussian that covers hundreds of tiles receives its nine sums from every one of them,
// and atomics to ONE address from all over the chip serialise: on the bench frame that was 0.31 of the kernel's 0.59 ms
// (timing builds: atomics spread over per-tile or per-block addresses 0.31 ms, no atomics 0.25 ms). With the block feed
// a record is an ENTRY of its block's list, shared by at most the 64 tiles of the block: the tiles add into nine floats
// per entry (scratch: the 8 R bytes of keysUnsorted, dead after the forward call), and flush_block_acc_kernel then adds every
// entry's sums to its Gaussian's — one lane per entry, different addresses in one instruction, and a Gaussian is hit
// once per block it touches instead of once per tile. Needs 48 bytes per entry: E <= R / 6, else the direct path.
// Decided per block: the sums of the part of the list that the forward blend looked into (BlockMeta::walked, whole
// units) are cleared before and flushed after the kernel, which pays where that part is short — at most kAccMaxUnits
// units; bench frame, 1-2 units per block: render backward 0.59 -> 0.38 ms. A block whose tiles walk deeper lists (the
// frame from outside the cloud: up to 10 000 entries per tile, small splats, little contention) keeps the direct
// atomics and, where the forward wrote them, the sorted lists (a plain array of indices is quicker to walk than the
// block lists): with per-entry sums for every block that frame went from 1.26 to 1.38 ms, with a limit of four units
// from 1.04 to 1.14 ms, with two it is unchanged.
constexpr uint32_t kAccMaxUnits = 2;
constexpr uint32_t kAccFloats = 12;     // sums per entry: mean2D (2), conic + opacity (4), colour (3), cov2D (3)
__device__ __forceinline__ bool block_acc_fits(const BlockFeed& f, uint32_t b) {
    return f.acc != nullptr && (unsigned long long)kAccFloats * (unsigned long long)f.meta.list_start()[f.meta.nbp] <= f.acc_floats &&
           f.meta.walked()[b] <= kAccMaxUnits;
}

// Both helpers: workgroup (b, part) of kAccParts takes a slice of the entries of block b that the forward blend
// looked into (whole units from the front of the block's list; a pixel's last contributor lies there).
constexpr uint32_t kAccParts = 4;
__device__ __forceinline__ void walked_slice(const BlockFeed& f, size_t& e0, size_t& e1) {
    const uint32_t b = blockIdx.x / kAccParts, part = blockIdx.x % kAccParts;
    e0 = e1 = 0;
    if (!block_acc_fits(f, b)) return;
    const size_t first = f.meta.list_start()[b], end = f.meta.list_start()[b + 1];
    const size_t upto = min(end, first + (size_t)f.meta.walked()[b] * kUnit);
    const size_t per = ((upto - first + kAccParts - 1) / kAccParts + 3) / 4 * 4;
    e0 = min(upto, first + part * per);
    e1 = min(upto, e0 + per);
}

__global__ __launch_bounds__(256) void zero_block_acc_kernel(const BlockFeed f) {
    size_t e0, e1;
    walked_slice(f, e0, e1);
    for (size_t i = kAccFloats * e0 + threadIdx.x; i < kAccFloats * e1; i += 256) f.acc[i] = 0.0f;
}

__global__ __launch_bounds__(256) void flush_block_acc_kernel(const BlockFeed f, float* __restrict__ dL_dmean2D,
                                                              float* __restrict__ dL_dconic_opacity, float* __restrict__ dL_dcolors,
                                                              float* __restrict__ dL_dcov2D, double* __restrict__ sums64) {
    size_t e0, e1;
    walked_slice(f, e0, e1);
    for (size_t e = e0 + threadIdx.x; e < e1; e += 256) {
        const float* a = f.acc + kAccFloats * e;
        float v[kAccFloats];
#pragma unroll
        for (int k = 0; k < (int)kAccFloats; ++k) v[k] = a[k];
        bool any = false;
#pragma unroll
        for (int k = 0; k < (int)kAccFloats; ++k) any = any || v[k] != 0.0f;
        if (!any) continue;
        const size_t id = f.ent_idx[e];
        if (sums64) {
#pragma unroll
            for (int k = 0; k < (int)kAccFloats; ++k)
                if (v[k] != 0.0f) unsafeAtomicAdd(sums64 + kAccFloats * id + k, (double)v[k]);
            continue;
        }
        unsafeAtomicAdd(dL_dmean2D + 2 * id, v[0]); unsafeAtomicAdd(dL_dmean2D + 2 * id + 1, v[1]);
        unsafeAtomicAdd(dL_dconic_opacity + 4 * id, v[2]); unsafeAtomicAdd(dL_dconic_opacity + 4 * id + 1, v[3]);
        unsafeAtomicAdd(dL_dconic_opacity + 4 * id + 2, v[4]); unsafeAtomicAdd(dL_dconic_opacity + 4 * id + 3, v[5]);
        unsafeAtomicAdd(dL_dcolors + 3 * id, v[6]); unsafeAtomicAdd(dL_dcolors + 3 * id + 1, v[7]); unsafeAtomicAdd(dL_dcolors + 3 * id + 2, v[8]);
        if (dL_dcov2D) {
            unsafeAtomicAdd(dL_dcov2D + 4 * id, v[9]); unsafeAtomicAdd(dL_dcov2D + 4 * id + 1, v[10]); unsafeAtomicAdd(dL_dcov2D + 4 * id + 2, v[11]);
        }
    }
}

__global__ __launch_bounds__(64) void render_backward_kernel(const RenderBackwardParams p) {
    __shared__ uint32_t s_id[kWave];
    __shared__ float2 s_xy[kWave];
    __shared__ float4 s_co[kWave];
    __shared__ float4 s_rgb[kWave];
    __shared__ float s_floor[kWave];
    __shared__ unsigned long long s_exp[32];
    exp_table_init(s_exp, (int)threadIdx.x);       // (wave-private LDS: ordered inside the wave)

    // (tiles dealt to the XCDs in patches, as in the forward blend: blend_core.hpp)
    const int tile_local = tile_of_workgroup(p.tile_order ? (int)p.tile_order[blockIdx.x] : (int)blockIdx.x, p.dims.grid_x, p.dims.row_end - p.dims.row_begin);
    if (tile_local < 0) return;
    const int tile = p.dims.row_begin * p.dims.grid_x + tile_local;
    const int tx = tile % p.dims.grid_x, ty = tile / p.dims.grid_x;
    const int lane = threadIdx.x;
    const int px = tx * kTile + (lane & 15), py0 = ty * kTile + (lane >> 4);
    const float fx = (float)px;
    const size_t plane = (size_t)p.dims.width * (size_t)p.dims.height;
    const float bg0 = p.background[0], bg1 = p.background[1], bg2 = p.background[2];

    uint32_t last[4];
    float T[4], S[4], g0[4], g1[4], g2[4], fy[4];
    uint32_t hi = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int py = py0 + 4 * k;
        fy[k] = (float)py;
        const bool inside = px < p.dims.width && py < p.dims.height;
        last[k] = 0; T[k] = 1.0f; g0[k] = g1[k] = g2[k] = 0.0f;
        if (inside) {
            const size_t pid = (size_t)py * (size_t)p.dims.width + (size_t)px;
            last[k] = p.n_contrib[pid];
            T[k] = p.final_t[pid];
            g0[k] = p.dL_dout[pid]; g1[k] = p.dL_dout[pid + plane]; g2[k] = p.dL_dout[pid + 2 * plane];
        }
        S[k] = T[k] * (bg0 * g0[k] + bg1 * g1[k] + bg2 * g2[k]);     // what lies behind, dotted with dL/dC
        hi = max(hi, last[k]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) hi = max(hi, (uint32_t)__shfl_xor((int)hi, off, kWave));
    if (hi == 0) return;
    const uint2 range = p.ranges[tile];
    const TileBox box = tile_box(tx, ty, p.dims.width, p.dims.height);

    // One batch of up to 64 list entries, lane l holding the entry at 0-based list position idx_l (descending batches,
    // ascending lanes): as in the forward blend (blend_core.hpp) most records of a tile's list cannot light any of its
    // pixels; they are dropped here, one lane per record, instead of being walked by the whole wave.
    // Which of a record's twelve sums this lane files (wave_sum12 leaves them in lanes 0, 8, ..., 56 and 4, 20, 36, 52), and
    // where: one atomic instruction with twelve lanes instead of twelve instructions with one.
    constexpr int kSumOfRow[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    constexpr int kSumOfRow4[4] = {8, 10, 9, 11};
    int my_sum = (lane & 7) == 0 ? kSumOfRow[lane >> 3] : ((lane & 15) == 4 ? kSumOfRow4[lane >> 4] : -1);
    if (my_sum >= 9 && !p.dL_dcov2D && !p.sums64) my_sum = -1;
    float* direct_base = my_sum < 2 ? p.dL_dmean2D + my_sum : (my_sum < 6 ? p.dL_dconic_opacity + (my_sum - 2) :
                         (my_sum < 9 ? p.dL_dcolors + (my_sum - 6) : p.dL_dcov2D + (my_sum - 9)));
    const uint32_t direct_stride = my_sum < 2 ? 2u : (my_sum < 6 ? 4u : (my_sum < 9 ? 3u : 4u));
    // `key` is what the record's sums are filed under: the Gaussian's index, or (block feed with room for per-entry
    // sums, see below) the record's entry number in the block lists.
    bool per_entry = false;
    auto walk_batch = [&](const bool present, const uint32_t id, const uint32_t idx_l, const uint32_t key, const float2 xy_l,
                          const float4 co_l) {
        const bool keep = present && !record_misses_tile(xy_l, co_l, box);
        const unsigned long long kept_mask = __ballot(keep);
        if (keep) {
            const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(kept_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)kept_mask, 0u));
            s_id[slot] = key;
            s_xy[slot] = xy_l;
            s_co[slot] = co_l;
            const float* col = p.colors + 3 * (size_t)id;
            s_rgb[slot] = make_float4(col[0], col[1], col[2], __uint_as_float(idx_l));
            // A pixel can pass alpha >= 1/255 only where power >= -ln(255 opacity) (less a margin for the rounding of the
            // logarithm, the exponential and the product; NaN opacity: the reference's min(0.99, NaN) is 0.99, it counts)
            const float op = co_l.w;
            s_floor[slot] = op != op ? -__builtin_inff() : (op <= 0.0f ? __builtin_inff() : -__logf(255.0f * op) - 1e-3f);
        }
        // wave-private LDS: the writes above and the reads below are ordered inside the wave
        for (int j = (int)__popcll(kept_mask) - 1; j >= 0; --j) {
            const uint32_t idx0 = __float_as_uint(s_rgb[j].w);   // 0-based position in the tile's list
            const float2 xy = s_xy[j];
            const float4 co = s_co[j];
            const float dx = xy.x - fx;
            float a_mx = 0.0f, a_my = 0.0f, a_A = 0.0f, a_B = 0.0f, a_C = 0.0f, a_op = 0.0f, a_r = 0.0f, a_g = 0.0f, a_b = 0.0f;
            float a_m00 = 0.0f, a_m01 = 0.0f, a_m11 = 0.0f;
            bool any = false;
            // The reference's three per-pixel tests (GSCuda.cu:636-655) decide, as in the forward, with the forward's own
            // arithmetic; what they guard runs for the whole strip without branches — a lane that fails them adds zeros
            // (w = 0, dpow = 0) and keeps its T — behind three wave-uniform skips ("no pixel of the strip has the record in
            // its list", "the record is too faint to count anywhere on the strip": before the exponential, "no pixel passes
            // the tests"). Per-lane branches cost this loop more than the work they skipped: twelve accumulators to merge at
            // every level of nesting (0.314 -> 0.238 ms on the bench frame, 1.36 -> 0.91 ms from (0,0,-30)). The sums take
            // explicit fused multiply-adds (no reference order to keep here), the transmittance its reciprocal in one
            // instruction (1 ulp).
            const float4 col = s_rgb[j];
            const float hxx = -0.5f * dx * dx, flo = s_floor[j];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // (the tests as lane masks in scalar registers, as in the forward blend)
                const unsigned long long inlist = __ballot(idx0 < last[k]);      // else: behind this pixel's last contributor
                if (inlist == 0ull) continue;
                const float dy = xy.y - fy[k];
                const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
                const unsigned long long cand = inlist & ~__ballot(power > 0.0f) & __ballot(power >= flo);
                if (cand == 0ull) continue;                      // the record does not reach this strip
                const float G = exp_ref(power, s_exp);           // the forward's exp: same contributing set
                const float raw = co.w * G;
                const float alpha = fminf(0.99f, raw);
                const unsigned long long act_mask = cand & ~__ballot(alpha < 1.0f / 255.0f);
                if (act_mask == 0ull) continue;
                const bool act = __builtin_amdgcn_inverse_ballot_w64(act_mask);
                any = true;
                const float inv = __builtin_amdgcn_rcpf(1.0f - alpha);
                const float Tn = T[k] * inv;                     // transmittance in front of this record
                T[k] = act ? Tn : T[k];
                const float cg = __builtin_fmaf(col.x, g0[k], __builtin_fmaf(col.y, g1[k], col.z * g2[k]));
                const float w = act ? alpha * Tn : 0.0f;
                a_r = __builtin_fmaf(w, g0[k], a_r); a_g = __builtin_fmaf(w, g1[k], a_g); a_b = __builtin_fmaf(w, g2[k], a_b);
                const float dL_dalpha = __builtin_fmaf(Tn, cg, -(S[k] * inv));
                S[k] = __builtin_fmaf(cg, w, S[k]);
                const bool live = __builtin_amdgcn_inverse_ballot_w64(act_mask & ~__ballot(raw > 0.99f));   // clamped: alpha does not move with the parameters
                const float gop = live ? G * dL_dalpha : 0.0f;
                const float dpow = live ? raw * dL_dalpha : 0.0f;
                a_op += gop;
                a_A = __builtin_fmaf(hxx, dpow, a_A);
                a_B = __builtin_fmaf(-(dx * dy), dpow, a_B);
                a_C = __builtin_fmaf(-0.5f * dy * dy, dpow, a_C);
                // u = K d is the gradient of the power w.r.t. the centre (up to sign) AND what the covariance sees: the pixel's
                // share of dL/dK is -0.5 dpow d d^T, so its share of dL/dcov2D = -K (dL/dK) K is 0.5 dpow u u^T — summed here,
                // pixel by pixel, instead of being derived from the summed dL/dK afterwards: that product loses cond(K)^2 of
                // the sums' digits, and the conic of a splat that fills the screen has a condition number of 1e6
                const float ux = __builtin_fmaf(co.x, dx, co.y * dy), uy = __builtin_fmaf(co.y, dx, co.z * dy);
                a_mx = __builtin_fmaf(-ux, dpow, a_mx);
                a_my = __builtin_fmaf(-uy, dpow, a_my);
                const float hux = 0.5f * dpow * ux, huy = 0.5f * dpow * uy;      // (the masked factor first: 0 x finite)
                a_m00 = __builtin_fmaf(hux, ux, a_m00); a_m01 = __builtin_fmaf(hux, uy, a_m01); a_m11 = __builtin_fmaf(huy, uy, a_m11);
            }
            if (!any) continue;
            const float total = wave_sum12(a_mx, a_my, a_A, a_B, a_C, a_op, a_r, a_g, a_b, a_m00, a_m01, a_m11);
            if (my_sum >= 0) {
                const size_t key = s_id[j];
                // (double sums: a record's twelve lie side by side, one 96-byte piece per atomic instruction; a Gaussian that
                // fills the screen gets its sums from 8 160 tiles with terms of either sign: in float the order of arrival
                // showed in the fourth digit of its gradients)
                if (!per_entry && p.sums64) unsafeAtomicAdd(p.sums64 + kAccFloats * key + my_sum, (double)total);
                else unsafeAtomicAdd(per_entry ? p.feed.acc + kAccFloats * key + my_sum : direct_base + direct_stride * key, total);
            }
        }
    };

    // The batches come from one of two iterators, back to front — the sorted list (a plain array), or the block lists
    // (BlockFeed, blockbin.hpp) — and pass two stations before they are walked, one loop iteration apart, as in the forward
    // blend: ids two batches ahead, records one batch ahead (loads unconditional: lanes without an entry read Gaussian 0).
    struct Batch {
        bool valid = false, present = false;
        uint32_t id = 0, idx_l = 0, key = 0;
        float2 xy = {0.0f, 0.0f};
        float4 co = {0.0f, 0.0f, 0.0f, 0.0f};
    };
    const BlockFeed& f = p.feed;
    const uint32_t b = (uint32_t)(ty / kBH) * (uint32_t)f.nbx + (uint32_t)(tx / kBW);
    per_entry = f.ent_idx != nullptr && block_acc_fits(f, b);
    const bool sorted = p.point_list && !per_entry;
    // -- sorted list: batch c = list positions [64 c, 64 c + 64) below hi
    int it_c = (int)((hi - 1) / kWave);
    // -- block lists: unit it_u of the block, its batches still to come in it_nz (highest first)
    const uint32_t col = (uint32_t)(tx % kBW), row = 8u + (uint32_t)(ty % kBH), t_in_block = (uint32_t)((ty % kBH) * kBW + tx % kBW);
    uint32_t u0 = 0, list0 = 0, it_u = 0, it_nz = 0, tm_lo = 0, tm_hi = 0, start_v = 0;
    if (!sorted) {
        u0 = f.meta.unit_start()[b];
        const uint32_t u1 = f.meta.unit_start()[b + 1];
        list0 = f.meta.list_start()[b];
        // the last unit whose first entry of this tile lies in front of position hi
        uint32_t below = 0;
        for (uint32_t k = u0; k < u1; k += kWave) {
            const uint32_t u = k + (uint32_t)lane;
            below += (uint32_t)__popcll(__ballot(u < u1 && f.prefix[(size_t)u * 64 + t_in_block] < hi));
        }
        it_u = u0 + below;                     // (units [u0, it_u) to come, last first)
    }
    auto next_batch = [&]() {
        Batch nb;
        if (sorted) {
            if (it_c < 0) return nb;
            const uint32_t first = (uint32_t)it_c * kWave;
            --it_c;
            nb.valid = true;
            nb.present = (uint32_t)lane < min((uint32_t)kWave, hi - first);
            nb.idx_l = first + (uint32_t)lane;
            nb.id = p.point_list[range.x + (nb.present ? nb.idx_l : 0u)];
            nb.key = nb.id;
            return nb;
        }
        while (it_nz == 0u) {
            if (it_u == u0) return nb;
            --it_u;
            const uint32_t base = f.prefix[(size_t)it_u * 64 + t_in_block];
            const uint2* um = f.unit_masks + (size_t)it_u * 16 * kBatches + (lane & (kBatches - 1));
            const uint2 xm = um[col * kBatches], ym = um[row * kBatches];
            tm_lo = (lane < kBatches) ? (xm.x & ym.x) : 0u;
            tm_hi = (lane < kBatches) ? (xm.y & ym.y) : 0u;
            const uint32_t c = (uint32_t)__popc(tm_lo) + (uint32_t)__popc(tm_hi);
            start_v = base + prefix32_inclusive(c) - c;                  // lane w < 32: list position of batch w's first entry
            it_nz = (uint32_t)__ballot(c != 0u && start_v < hi);          // batches with an entry in front of hi
        }
        const int w = 31 - __builtin_clz(it_nz);                          // (wave-uniform: it_nz comes from a ballot)
        it_nz &= ~(1u << w);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)tm_lo, w);
        const uint32_t hi_m = (uint32_t)__builtin_amdgcn_readlane((int)tm_hi, w);
        const uint32_t start = (uint32_t)__builtin_amdgcn_readlane((int)start_v, w);
        const unsigned long long mask = ((unsigned long long)hi_m << 32) | lo;
        nb.valid = true;
        nb.idx_l = start + __builtin_amdgcn_mbcnt_hi(hi_m, __builtin_amdgcn_mbcnt_lo(lo, 0u));
        nb.present = __builtin_amdgcn_inverse_ballot_w64(mask) && nb.idx_l < hi;
        const uint32_t e = list0 + (it_u - u0) * kUnit + (uint32_t)w * kWave + (uint32_t)lane;
        nb.id = f.ent_idx[nb.present ? e : list0];
        nb.key = per_entry ? e : nb.id;
        return nb;
    };
    auto fetch = [&](Batch& nb) {
        const uint32_t at = (nb.valid && nb.present) ? nb.id : 0u;
        nb.xy = p.means2D[at];
        nb.co = p.conic_opacity[at];
    };
    Batch b0 = next_batch();
    fetch(b0);
    Batch b1 = next_batch();
    while (b0.valid) {
        fetch(b1);
        Batch b2 = next_batch();
        walk_batch(b0.present, b0.id, b0.idx_l, b0.key, b0.xy, b0.co);
        b0 = b1;
        b1 = b2;
    }
}

// ---- per Gaussian: conic -> cov2D -> cov3D, colour -> DC harmonics -----------------------------------
struct PreprocessBackwardParams {
    int n;
    const float4* means3D;
    const float* cov3D;
    const int32_t* radii;
    const float* view;
    float tan_fovx, tan_fovy, focal;
    const float4* dL_dconic_opacity;
    const float4* dL_dcov2D;        // (m00, m01, m11, 0) summed by the render backward, or null: derived from dL_dconic_opacity
    // With sums64 (f64[12 N]) the render backward's sums arrive there instead: this kernel reads them, rounds them into the
    // four float arrays (every Gaussian: zeros for the ones without a tile), and leaves the doubles zero for the next call.
    double* sums64;
    float2* out_mean2D; float4* out_conic_opacity; float* out_colors; float4* out_cov2D;
    const float* dL_dcolors;
    int chain;              // run the covariance / colour chain (else: only round the double sums)
    float* dL_dcov3D;       // f32[6 N] or null
    float* dL_dshs;         // f32[48 N] or null; only the DC triple of every Gaussian is written
    // chain down to the inputs (each output optional)
    const float* proj;
    const float4* scales;
    const float4* rotations;
    float scale_modifier;
    int width, height;
    const float2* dL_dmean2D;
    float4* dL_dmeans3D;    // vec4[N] (x, y, z, 0): the layout of means3D
    float4* dL_dscales;     // vec4[N] (x, y, z, 0)
    float4* dL_drotations;  // vec4[N], w.r.t. the quaternion as given (not normalised)
    // the upstream (`inria`) semantics profile, csrc/preprocess_inria.hip: two focal lengths, w epsilon 1e-7, raw
    // quaternion, colour = max(0, 0.5 + SH(view direction)) on [16][3] coefficients
    int inria, sh_deg;
    float focal_x, focal_y, w_eps;
    const float* shs;
    const float* cam_pos;
    const uint8_t* clamped;   // GeometryState::clamped of the forward call: channel c of Gaussian i was clamped at zero
};

__constant__ float kShC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                               0.5462742152960396f};
__constant__ float kShC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// The 16 real SH basis values at the unit direction (x, y, z) and their gradients (oracle/backward_np.py: sh_basis).
__device__ __forceinline__ void sh_basis_grad(int deg, float x, float y, float z, float (&B)[16], float (&G)[16][3]) {
    constexpr float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { B[k] = 0.0f; G[k][0] = G[k][1] = G[k][2] = 0.0f; }
    B[0] = C0;
    if (deg > 0) {
        B[1] = -C1 * y; B[2] = C1 * z; B[3] = -C1 * x;
        G[1][1] = -C1; G[2][2] = C1; G[3][0] = -C1;
    }
    if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        B[4] = kShC2[0] * xy; B[5] = kShC2[1] * yz; B[6] = kShC2[2] * (2.0f * zz - xx - yy); B[7] = kShC2[3] * xz; B[8] = kShC2[4] * (xx - yy);
        G[4][0] = kShC2[0] * y; G[4][1] = kShC2[0] * x;
        G[5][1] = kShC2[1] * z; G[5][2] = kShC2[1] * y;
        G[6][0] = -2.0f * kShC2[2] * x; G[6][1] = -2.0f * kShC2[2] * y; G[6][2] = 4.0f * kShC2[2] * z;
        G[7][0] = kShC2[3] * z; G[7][2] = kShC2[3] * x;
        G[8][0] = 2.0f * kShC2[4] * x; G[8][1] = -2.0f * kShC2[4] * y;
        if (deg > 2) {
            B[9] = kShC3[0] * y * (3.0f * xx - yy); B[10] = kShC3[1] * xy * z; B[11] = kShC3[2] * y * (4.0f * zz - xx - yy);
            B[12] = kShC3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy); B[13] = kShC3[4] * x * (4.0f * zz - xx - yy);
            B[14] = kShC3[5] * z * (xx - yy); B[15] = kShC3[6] * x * (xx - 3.0f * yy);
            G[9][0] = kShC3[0] * 6.0f * xy; G[9][1] = kShC3[0] * (3.0f * xx - 3.0f * yy);
            G[10][0] = kShC3[1] * yz; G[10][1] = kShC3[1] * xz; G[10][2] = kShC3[1] * xy;
            G[11][0] = kShC3[2] * -2.0f * xy; G[11][1] = kShC3[2] * (4.0f * zz - xx - 3.0f * yy); G[11][2] = kShC3[2] * 8.0f * yz;
            G[12][0] = kShC3[3] * -6.0f * xz; G[12][1] = kShC3[3] * -6.0f * yz; G[12][2] = kShC3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy);
            G[13][0] = kShC3[4] * (4.0f * zz - 3.0f * xx - yy); G[13][1] = kShC3[4] * -2.0f * xy; G[13][2] = kShC3[4] * 8.0f * xz;
            G[14][0] = kShC3[5] * 2.0f * xz; G[14][1] = kShC3[5] * -2.0f * yz; G[14][2] = kShC3[5] * (xx - yy);
            G[15][0] = kShC3[6] * (3.0f * xx - 3.0f * yy); G[15][1] = kShC3[6] * -6.0f * xy;
        }
    }
}

// Memory schedule: every load of the Gaussian first, all stores last (a store may alias a later load as far as the
// compiler knows, and a load issued behind a store waits for the store's acknowledgement too: the round-1 kernel
// stored dL_dmeans3D before it fetched the scales and the quaternion). INRIA selects the upstream profile's chain at
// compile time: the 48 SH gradients of that profile do not cost the reference's profile their registers.
constexpr int kShRow = 49;      // LDS floats per SH record: odd, so that the 64 lanes' accesses to coefficient k spread over the banks
template <bool INRIA>
__global__ __launch_bounds__(256) void preprocess_backward_kernel(const PreprocessBackwardParams p) {
    // (upstream profile: the wave's 64 x 48 SH coefficients come in, and their gradients go out, through LDS — 16 bytes
    // per lane of consecutive memory per instruction instead of 64 records touched by each; lanes past n stay for that)
    __shared__ float s_sh[INRIA ? 4 * kWave * kShRow : 1];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const bool valid = idx < p.n;
    // (without dL_dcov3D the kernel only rounds the double sums into the float arrays: radii may then be missing)
    const bool has_tile = valid && (p.radii == nullptr || p.radii[idx] > 0);
    const bool visible = has_tile && p.chain != 0;
    // ---- loads ----
    float4 mean = make_float4(0.0f, 0.0f, 0.0f, 0.0f), g = mean, sc = mean, rot = mean;
    float2 g2 = make_float2(0.0f, 0.0f), c3a = g2, c3b = g2, c3c = g2;
    float gc[3] = {0.0f, 0.0f, 0.0f};
    double sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // (mean2D 2, conic 3 + opacity, colour 3, cov2D 3), as summed
    if (has_tile && p.sums64) {
        const double2* sp = reinterpret_cast<const double2*>(p.sums64 + 12 * (size_t)idx);
#pragma unroll
        for (int k = 0; k < 6; ++k) { const double2 v2 = sp[k]; sum[2 * k] = v2.x; sum[2 * k + 1] = v2.y; }
    }
    if (visible) {
        mean = p.means3D[idx];
        const float2* c3p = reinterpret_cast<const float2*>(p.cov3D + 6 * (size_t)idx);
        c3a = c3p[0]; c3b = c3p[1]; c3c = c3p[2];
        if (!p.sums64) {
            g = p.dL_dconic_opacity[idx];
            if (p.dL_dmeans3D) g2 = p.dL_dmean2D[idx];
            if (p.dL_dshs) {
                const float* gcp = p.dL_dcolors + 3 * (size_t)idx;
                gc[0] = gcp[0]; gc[1] = gcp[1]; gc[2] = gcp[2];
            }
            if (p.dL_dcov2D) { const float4 gcv = p.dL_dcov2D[idx]; sum[9] = gcv.x; sum[10] = gcv.y; sum[11] = gcv.z; }
            sum[0] = g2.x; sum[1] = g2.y; sum[2] = g.x; sum[3] = g.y; sum[4] = g.z; sum[5] = g.w; sum[6] = gc[0]; sum[7] = gc[1]; sum[8] = gc[2];
        }
        if (p.dL_dscales) { sc = p.scales[idx]; rot = p.rotations[idx]; }
    }
    if (p.sums64) { gc[0] = (float)sum[6]; gc[1] = (float)sum[7]; gc[2] = (float)sum[8]; }
    const bool have_cov2D = p.sums64 != nullptr || p.dL_dcov2D != nullptr;
    // The chain is N-sized and waits for memory: its arithmetic runs in double, so that what the render backward summed is
    // not degraded further (the rotation gradient of a nearly round splat is a difference of nearly equal numbers).
    typedef double R;
    R out[6] = {0, 0, 0, 0, 0, 0};
    R gmean[3] = {0, 0, 0};
    if (visible) {
        const float* vf = p.view;
        R v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (R)vf[i];
        // t, clamped as in computeCov2D (GSCuda.cu:201-210): the clamp DECISIONS are the forward's, in its float32 arithmetic
        const float txf = (vf[0] * mean.x + vf[4] * mean.y) + (vf[8] * mean.z + vf[12] * 1.0f);
        const float tyf = (vf[1] * mean.x + vf[5] * mean.y) + (vf[9] * mean.z + vf[13] * 1.0f);
        const float tzf = (vf[2] * mean.x + vf[6] * mean.y) + (vf[10] * mean.z + vf[14] * 1.0f);
        const float limx = 1.3f * p.tan_fovx, limy = 1.3f * p.tan_fovy;
        const float rx = txf / tzf, ry = tyf / tzf;
        const float cxf = fminf(limx, fmaxf(-limx, rx)), cyf = fminf(limy, fmaxf(-limy, ry));
        const bool clx = rx != cxf, cly = ry != cyf;          // clamped: t.x (t.y) no longer moves the entry, t.z does
        const R tz = (R)tzf, cx = (R)cxf, cy = (R)cyf;
        const R tx = cx * tz, ty = cy * tz;
        // P = J W (2 x 3): cov2D = P Sigma P^T
        const R fx = INRIA ? (R)p.focal_x : (R)p.focal, fy = INRIA ? (R)p.focal_y : (R)p.focal;
        const R j00 = fx / tz, j11 = fy / tz, j02 = -fx * tx / (tz * tz), j12 = -fy * ty / (tz * tz);
        R P[2][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            P[0][c] = j00 * v[4 * c + 0] + j02 * v[4 * c + 2];
            P[1][c] = j11 * v[4 * c + 1] + j12 * v[4 * c + 2];
        }
        const R c3[6] = {(R)c3a.x, (R)c3a.y, (R)c3b.x, (R)c3b.y, (R)c3c.x, (R)c3c.y};
        const R s[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
        R ps[2][3];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) ps[r][c] = P[r][0] * s[0][c] + P[r][1] * s[1][c] + P[r][2] * s[2][c];
        const R a = (ps[0][0] * P[0][0] + ps[0][1] * P[0][1] + ps[0][2] * P[0][2]) + 0.3;
        const R b = ps[0][0] * P[1][0] + ps[0][1] * P[1][1] + ps[0][2] * P[1][2];
        const R cc = (ps[1][0] * P[1][0] + ps[1][1] * P[1][1] + ps[1][2] * P[1][2]) + 0.3;
        const R det = a * cc - b * b;
        if (det != 0.0) {
            // gM: gradient w.r.t. the full symmetric cov2D. Summed directly by the render backward (dL_dcov2D), or, without that
            // array, -K gK K from the summed conic gradient (K = cov2D^-1: loses cond(K)^2 of the sums' digits)
            R m00 = sum[9], m01 = sum[10], m11 = sum[11];
            if (!have_cov2D) {
                const R inv = 1.0 / det;
                const R k00 = cc * inv, k01 = -b * inv, k11 = a * inv;
                const R q00 = sum[2], q01 = 0.5 * sum[3], q11 = sum[4];      // gradient w.r.t. the full symmetric K
                const R r00 = k00 * q00 + k01 * q01, r01 = k00 * q01 + k01 * q11;
                const R r10 = k01 * q00 + k11 * q01, r11 = k01 * q01 + k11 * q11;
                m00 = -(r00 * k00 + r01 * k01); m01 = -(r00 * k01 + r01 * k11);
                m11 = -(r10 * k01 + r11 * k11);
            }
            // gS = P^T gM P, stored entries: off-diagonals appear twice in Sigma
            R gp[2][3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                gp[0][c] = m00 * P[0][c] + m01 * P[1][c];
                gp[1][c] = m01 * P[0][c] + m11 * P[1][c];
            }
            auto gs = [&](int r, int c) { return P[0][r] * gp[0][c] + P[1][r] * gp[1][c]; };
            out[0] = gs(0, 0); out[1] = 2.0 * gs(0, 1); out[2] = 2.0 * gs(0, 2);
            out[3] = gs(1, 1); out[4] = 2.0 * gs(1, 2); out[5] = gs(2, 2);
            if (p.dL_dmeans3D) {
                // through the Jacobian J(t): cov2D = J Mw J^T with Mw = W Sigma W^T; gJ = 2 gcov J Mw
                R ws[3][3], mw[3][3];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) ws[r][c] = v[0 + r] * s[0][c] + v[4 + r] * s[1][c] + v[8 + r] * s[2][c];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) mw[r][c] = ws[r][0] * v[0 + c] + ws[r][1] * v[4 + c] + ws[r][2] * v[8 + c];
                const R J[2][3] = {{j00, 0.0, j02}, {0.0, j11, j12}};
                R jm[2][3], gJ[2][3];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) jm[r][c] = J[r][0] * mw[0][c] + J[r][1] * mw[1][c] + J[r][2] * mw[2][c];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    gJ[0][c] = 2.0 * (m00 * jm[0][c] + m01 * jm[1][c]);
                    gJ[1][c] = 2.0 * (m01 * jm[0][c] + m11 * jm[1][c]);
                }
                const R itz2 = 1.0 / (tz * tz);
                const R g_tx = -gJ[0][2] * fx * itz2, g_ty = -gJ[1][2] * fy * itz2;
                const R g_tz = -(gJ[0][0] * fx + gJ[1][1] * fy) * itz2 + (gJ[0][2] * fx * tx + gJ[1][2] * fy * ty) * (2.0 / (tz * tz * tz));
                const R gt0 = clx ? 0.0 : g_tx, gt1 = cly ? 0.0 : g_ty;
                const R gt2 = g_tz + (clx ? g_tx * cx : 0.0) + (cly ? g_ty * cy : 0.0);
#pragma unroll
                for (int j = 0; j < 3; ++j) gmean[j] = v[4 * j + 0] * gt0 + v[4 * j + 1] * gt1 + v[4 * j + 2] * gt2;
            }
        }
        if (p.dL_dmeans3D) {
            // pixel-space centre: pix = ((proj mean).x / ((proj mean).w + 0.001) * 0.5 + 0.5) * W (GSCuda.cu:302-305, :342)
            const float* pm = p.proj;
            const R mw_ = INRIA ? 1.0 : (R)mean.w;          // (the upstream projection takes the point as (x, y, z, 1))
            const R mx = (R)mean.x, my = (R)mean.y, mz = (R)mean.z;
            const R hx = ((R)pm[0] * mx + (R)pm[4] * my) + ((R)pm[8] * mz + (R)pm[12] * mw_);
            const R hy = ((R)pm[1] * mx + (R)pm[5] * my) + ((R)pm[9] * mz + (R)pm[13] * mw_);
            const R wp = (INRIA ? (R)p.w_eps : (R)0.001f) + (((R)pm[3] * mx + (R)pm[7] * my) + ((R)pm[11] * mz + (R)pm[15] * mw_));
            const R iw = 1.0 / wp, iw2 = iw * iw;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const R dx = 0.5 * (R)p.width * ((R)pm[4 * j + 0] * iw - hx * (R)pm[4 * j + 3] * iw2);
                const R dy = 0.5 * (R)p.height * ((R)pm[4 * j + 1] * iw - hy * (R)pm[4 * j + 3] * iw2);
                gmean[j] += dx * sum[0] + dy * sum[1];
            }
        }
    }
    [[maybe_unused]] const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    [[maybe_unused]] float* sh_rows = s_sh + (INRIA ? wave * kWave * kShRow : 0);
    [[maybe_unused]] const size_t sh_first = ((size_t)blockIdx.x * 256 + (size_t)wave * kWave) * 48, sh_limit = (size_t)p.n * 48;
    if constexpr (INRIA) {
        if (p.dL_dshs) {
            // colour_c = max(0, 0.5 + sum_k B_k(dir) sh[k][c]), dir = (mean - cam) / |mean - cam|: dL/dsh[k][c] = B_k g_c, and
            // the direction moves with the mean; a channel clamped at zero passes nothing (oracle: inria_color_backward)
            const int terms = (p.sh_deg + 1) * (p.sh_deg + 1);
            if (__ballot(visible) != 0ull) {
                const int pieces = p.sh_deg > 2 ? 12 : (p.sh_deg > 1 ? 7 : (p.sh_deg > 0 ? 3 : 1));   // 16-byte pieces of a record in use
                float4 piece[12];
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    const int f4 = q * kWave + lane;
                    const size_t at = sh_first + 4 * (size_t)f4;
                    piece[q] = ((f4 % 12) < pieces && at + 3 < sh_limit) ? *reinterpret_cast<const float4*>(p.shs + at)
                                                                        : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    const int f4 = q * kWave + lane;
                    float* dst = sh_rows + (f4 / 12) * kShRow + (f4 % 12) * 4;
                    dst[0] = piece[q].x; dst[1] = piece[q].y; dst[2] = piece[q].z; dst[3] = piece[q].w;
                }
            }
            // wave-private LDS: ordered inside the wave. From here a lane reads and then overwrites its own row only.
            float* row = sh_rows + lane * kShRow;
            float dx = 0.0f, dy = 0.0f, dz = 0.0f, il = 0.0f;
            float B[16], G[16][3];
            float g[3] = {0.0f, 0.0f, 0.0f};
            if (visible) {
                dx = mean.x - p.cam_pos[0]; dy = mean.y - p.cam_pos[1]; dz = mean.z - p.cam_pos[2];
                const float len = sqrtf(dx * dx + dy * dy + dz * dz);
                il = 1.0f / len;
                dx *= il; dy *= il; dz *= il;
#pragma unroll
                for (int c = 0; c < 3; ++c) g[c] = p.clamped[3 * (size_t)idx + c] ? 0.0f : gc[c];
            }
            sh_basis_grad(p.sh_deg, dx, dy, dz, B, G);
            float gd[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const bool on = visible && k < terms;
                if (on) {
                    const float w = row[3 * k] * g[0] + row[3 * k + 1] * g[1] + row[3 * k + 2] * g[2];
                    gd[0] += G[k][0] * w; gd[1] += G[k][1] * w; gd[2] += G[k][2] * w;
                }
                row[3 * k] = on ? B[k] * g[0] : 0.0f;
                row[3 * k + 1] = on ? B[k] * g[1] : 0.0f;
                row[3 * k + 2] = on ? B[k] * g[2] : 0.0f;
            }
            if (visible) {
                const float dot = dx * gd[0] + dy * gd[1] + dz * gd[2];
                gmean[0] += (gd[0] - dx * dot) * il; gmean[1] += (gd[1] - dy * dot) * il; gmean[2] += (gd[2] - dz * dot) * il;
            }
        }
    }
    R gsc[3] = {0, 0, 0}, gq[4] = {0, 0, 0, 0};
    if (p.dL_dscales) {
        // Sigma = M M^T, M = R diag(mod s): gM = 2 gSigma M (off-diagonal stored gradients split over both entries)
        const R gS[3][3] = {{out[0], 0.5 * out[1], 0.5 * out[2]}, {0.5 * out[1], out[3], 0.5 * out[4]}, {0.5 * out[2], 0.5 * out[4], out[5]}};
        const R mod = (R)p.scale_modifier;
        const R sv[3] = {mod * (R)sc.x, mod * (R)sc.y, mod * (R)sc.z};
        if (visible && INRIA) {
            // Sigma = M M^T, M = R(q) diag(mod s) with the RAW quaternion q = (r, x, y, z) (oracle: inria_cov3d_backward)
            const R r = (R)rot.x, x = (R)rot.y, y = (R)rot.z, z = (R)rot.w;
            const R Rm[3][3] = {{1.0 - 2.0 * (y * y + z * z), 2.0 * (x * y - r * z), 2.0 * (x * z + r * y)},
                                {2.0 * (x * y + r * z), 1.0 - 2.0 * (x * x + z * z), 2.0 * (y * z - r * x)},
                                {2.0 * (x * z - r * y), 2.0 * (y * z + r * x), 1.0 - 2.0 * (x * x + y * y)}};
            R gR[3][3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                R gm[3];
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) gm[rr] = 2.0 * (gS[rr][0] * Rm[0][c] + gS[rr][1] * Rm[1][c] + gS[rr][2] * Rm[2][c]) * sv[c];
                gsc[c] = mod * (Rm[0][c] * gm[0] + Rm[1][c] * gm[1] + Rm[2][c] * gm[2]);
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) gR[rr][c] = gm[rr] * sv[c];
            }
            gq[0] = 2.0 * (z * (gR[1][0] - gR[0][1]) + y * (gR[0][2] - gR[2][0]) + x * (gR[2][1] - gR[1][2]));
            gq[1] = 2.0 * (y * (gR[0][1] + gR[1][0]) + z * (gR[0][2] + gR[2][0]) + r * (gR[2][1] - gR[1][2])) - 4.0 * x * (gR[1][1] + gR[2][2]);
            gq[2] = 2.0 * (x * (gR[0][1] + gR[1][0]) + r * (gR[0][2] - gR[2][0]) + z * (gR[1][2] + gR[2][1])) - 4.0 * y * (gR[0][0] + gR[2][2]);
            gq[3] = 2.0 * (r * (gR[1][0] - gR[0][1]) + x * (gR[0][2] + gR[2][0]) + y * (gR[1][2] + gR[2][1])) - 4.0 * z * (gR[0][0] + gR[1][1]);
        } else if (visible) {
            const R qx = (R)rot.x, qy = (R)rot.y, qz = (R)rot.z, qw = (R)rot.w;
            const R nrm = sqrt((qx * qx + qy * qy) + (qz * qz + qw * qw));
            const R inv = 1.0 / nrm;
            const R x = qx * inv, y = qy * inv, z = qz * inv, w = qw * inv;
            const R Rm[3][3] = {{2.0 * (x * x + y * y) - 1.0, 2.0 * (y * z - x * w), 2.0 * (y * w + x * z)},
                                {2.0 * (y * z + x * w), 2.0 * (x * x + z * z) - 1.0, 2.0 * (z * w - x * y)},
                                {2.0 * (y * w - x * z), 2.0 * (z * w + x * y), 2.0 * (x * x + w * w) - 1.0}};
            R gM[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    gM[r][c] = 2.0 * (gS[r][0] * Rm[0][c] + gS[r][1] * Rm[1][c] + gS[r][2] * Rm[2][c]) * sv[c];
            R gR[3][3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                gsc[c] = mod * (Rm[0][c] * gM[0][c] + Rm[1][c] * gM[1][c] + Rm[2][c] * gM[2][c]);
#pragma unroll
                for (int r = 0; r < 3; ++r) gR[r][c] = gM[r][c] * sv[c];
            }
            // dR/dq for the normalised quaternion (x = real part), then through the normalisation
            const R gx = 4.0 * x * (gR[0][0] + gR[1][1] + gR[2][2]) + 2.0 * (w * (gR[1][0] - gR[0][1]) + z * (gR[0][2] - gR[2][0]) + y * (gR[2][1] - gR[1][2]));
            const R gy = 4.0 * y * gR[0][0] + 2.0 * (z * (gR[0][1] + gR[1][0]) + w * (gR[0][2] + gR[2][0]) + x * (gR[2][1] - gR[1][2]));
            const R gz = 4.0 * z * gR[1][1] + 2.0 * (y * (gR[0][1] + gR[1][0]) + w * (gR[1][2] + gR[2][1]) + x * (gR[0][2] - gR[2][0]));
            const R gw = 4.0 * w * gR[2][2] + 2.0 * (y * (gR[0][2] + gR[2][0]) + z * (gR[1][2] + gR[2][1]) + x * (gR[1][0] - gR[0][1]));
            const R dot = x * gx + y * gy + z * gz + w * gw;
            gq[0] = (gx - x * dot) * inv; gq[1] = (gy - y * dot) * inv; gq[2] = (gz - z * dot) * inv; gq[3] = (gw - w * dot) * inv;
        }
    }
    // ---- stores ----
    if constexpr (INRIA) {
        if (p.dL_dshs) {
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const int f4 = q * kWave + lane;
                const size_t at = sh_first + 4 * (size_t)f4;
                const float* src = sh_rows + (f4 / 12) * kShRow + (f4 % 12) * 4;
                if (at + 3 < sh_limit) *reinterpret_cast<float4*>(p.dL_dshs + at) = make_float4(src[0], src[1], src[2], src[3]);
            }
        }
    }
    if (valid && p.sums64) {
        // the sums, rounded once, for every Gaussian (what the memsets + float atomics leave in the other mode)
        if (p.out_mean2D) p.out_mean2D[idx] = make_float2((float)sum[0], (float)sum[1]);
        if (p.out_conic_opacity) p.out_conic_opacity[idx] = make_float4((float)sum[2], (float)sum[3], (float)sum[4], (float)sum[5]);
        if (p.out_colors) {
            float* oc = p.out_colors + 3 * (size_t)idx;
            oc[0] = (float)sum[6]; oc[1] = (float)sum[7]; oc[2] = (float)sum[8];
        }
        if (p.out_cov2D) p.out_cov2D[idx] = make_float4((float)sum[9], (float)sum[10], (float)sum[11], 0.0f);
        // (most Gaussians with a tile lie behind every pixel's last contributor and received nothing: no write for those)
        bool touched = false;
#pragma unroll
        for (int k = 0; k < 12; ++k) touched = touched || sum[k] != 0.0;
        if (touched) {
            double2* sp = reinterpret_cast<double2*>(p.sums64 + 12 * (size_t)idx);
#pragma unroll
            for (int k = 0; k < 6; ++k) sp[k] = make_double2(0.0, 0.0);
        }
    }
    if (valid && p.chain) {
        if (p.dL_dmeans3D) p.dL_dmeans3D[idx] = make_float4((float)gmean[0], (float)gmean[1], (float)gmean[2], 0.0f);
        if (p.dL_dscales) {
            p.dL_dscales[idx] = make_float4((float)gsc[0], (float)gsc[1], (float)gsc[2], 0.0f);
            if (p.dL_drotations) p.dL_drotations[idx] = make_float4((float)gq[0], (float)gq[1], (float)gq[2], (float)gq[3]);
        }
        if (p.dL_dcov3D) {
            float2* dst = reinterpret_cast<float2*>(p.dL_dcov3D + 6 * (size_t)idx);
            dst[0] = make_float2((float)out[0], (float)out[1]);
            dst[1] = make_float2((float)out[2], (float)out[3]);
            dst[2] = make_float2((float)out[4], (float)out[5]);
        }
    }
    // (lanes past n stay to the end: the SH gradients above and the DC gradient below are written by the wave together)
    if (p.dL_dshs && !INRIA) {
        // colour = 0.5 + 0.4 DC (GSCuda.cu:362-366). The triple goes out as the whole 64 bytes it lies in (the 13 floats
        // behind it are gradients of coefficients the colour does not depend on: zero). 12 bytes at a 192-byte stride
        // are a partial write per Gaussian, a read-modify-write in the memory: measured for 16 / 32 / 64 / 128 / 192
        // bytes per Gaussian the chain takes 0.47 / 0.48 / 0.38 / 0.57 / 0.68 ms on the bench frame (0.20 without dL_dshs).
        // Four lanes write one Gaussian's 64 bytes in one instruction (16 Gaussians per instruction), so that every
        // store leaves the CU as whole 64-byte pieces instead of a quarter of 64 different ones: 0.38 -> 0.36 ms; streaming: -> 0.33 ms.
        const float v0 = visible ? 0.4f * gc[0] : 0.0f, v1 = visible ? 0.4f * gc[1] : 0.0f, v2 = visible ? 0.4f * gc[2] : 0.0f;
        const int q = lane & 3;
        const size_t wave_first = (size_t)idx - (size_t)lane;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int src = it * 16 + (lane >> 2);
            const float a = __shfl(v0, src, kWave), b = __shfl(v1, src, kWave), c = __shfl(v2, src, kWave);
            const size_t gi = wave_first + (size_t)src;
            if (gi < (size_t)p.n)
            {
                // (a streaming store: each 128-byte line of dL_dshs gets this one 64-byte piece and nothing else; the other
                // outputs measured no better with streaming stores: 0.296 / 0.269 / 0.280 ms plain / this one / all)
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                const f32x4 val = {q == 0 ? a : 0.0f, q == 0 ? b : 0.0f, q == 0 ? c : 0.0f, 0.0f};
                __builtin_nontemporal_store(val, reinterpret_cast<f32x4*>(p.dL_dshs + 48 * gi) + q);
            }
        }
    }
}

// Events of one profiled call: created on the device that is current for THIS call, destroyed when it returns.
struct CallEvents {
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    int create() {
        for (auto& e : ev) GSR_HIP_TRY(hipEventCreate(&e));
        return GSR_OK;
    }
    ~CallEvents() {
        for (auto& e : ev)
            if (e) (void)hipEventDestroy(e);
    }
};

}  // namespace
}  // namespace gsr

using namespace gsr;

static int backward_impl(gsr_backward_args* a) {
    if (!a || a->struct_size != sizeof(gsr_backward_args)) return GSR_ERR_INVALID_ARG;
    a->stage_ms[0] = a->stage_ms[1] = 0.0f;
    const int n = a->num_gaussians;
    if (n <= 0 || a->width <= 0 || a->height <= 0 || !a->background || !a->means2D || !a->conic_opacity || !a->colors ||
        !a->ranges || !a->n_contrib || !a->final_t || !a->dL_dout_color)
        return GSR_ERR_INVALID_ARG;
    // (the float arrays of the sums are where the sums are kept unless sums_f64 keeps them: optional outputs then)
    const bool wide = a->sums_f64 != nullptr;
    if (!wide && (!a->dL_dmean2D || !a->dL_dconic_opacity || !a->dL_dcolors)) return GSR_ERR_INVALID_ARG;
    // the chain runs for whichever of its outputs is asked for; without sums_f64 it starts from dL_dcov3D's presence as before
    const bool chain = a->dL_dcov3D || (wide && (a->dL_dmeans3D || a->dL_dscales || a->dL_dshs));
    if (chain && (!a->cov3D || !a->means3D || !a->view_matrix || !a->radii)) return GSR_ERR_INVALID_ARG;
    if ((a->dL_dmeans3D || a->dL_dscales || a->dL_drotations) && !chain) return GSR_ERR_INVALID_ARG;
    if (a->dL_dmeans3D && !a->proj_matrix) return GSR_ERR_INVALID_ARG;
    if ((a->dL_dscales || a->dL_drotations) && (!a->scales || !a->rotations || !a->dL_dscales)) return GSR_ERR_INVALID_ARG;
    const bool inria = (a->flags & GSR_FLAG_SEMANTICS_INRIA) != 0;
    // the upstream profile's chain needs what its colour was computed from
    if (inria && a->dL_dshs && (!a->shs || !a->cam_pos || !a->clamped || !a->means3D)) return GSR_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)a->stream;
    const bool profile = (a->flags & GSR_FLAG_PROFILE) != 0;
    CallEvents g_bw;
    hipEvent_t* const g_bw_ev = g_bw.ev;
    if (profile) {
        const int rc = g_bw.create();
        if (rc != GSR_OK) return rc;
    }

    FrameDims d;
    d.width = a->width; d.height = a->height;
    d.grid_x = (a->width + kTile - 1) / kTile; d.grid_y = (a->height + kTile - 1) / kTile;
    d.row_begin = 0; d.row_end = d.grid_y;
    if (a->tile_row_begin != 0 || a->tile_row_end != 0) {
        if (a->tile_row_begin < 0 || a->tile_row_end > d.grid_y || a->tile_row_begin > a->tile_row_end) return GSR_ERR_INVALID_ARG;
        d.row_begin = a->tile_row_begin; d.row_end = a->tile_row_end;
    }
    // Which lists the forward call left (see gsr_backward_args.receipt). With a receipt: from the receipt and the chunk
    // layouts — where that call ran the block plan, its block lists serve the tiles of shallow blocks (per-entry gradient
    // sums, see block_acc_fits) and, if it left the sorted lists unwritten (GSR_FLAG_NO_SORTED_LISTS), all tiles. Without
    // one only the reference's contract can hold (sorted lists in point_list), and whether it does is read off
    // point_list[0]: a call that skipped the lists left GSR_LISTS_SKIPPED_STAMP there.
    BlockFeed feed = {};
    bool from_blocks = false, lists_written = true;
    const bool have_receipt = a->receipt.magic != 0;        // (any other magic than the library's: refused below)
    bool nothing_rendered = false;
    if (have_receipt) {
        const int rc = lists_of_receipt(a->receipt, n, a->width, a->height, d.row_begin, d.row_end, a->point_list, &feed,
                                        &from_blocks, &lists_written);
        if (rc != GSR_OK) return rc;
        nothing_rendered = a->receipt.num_rendered == 0;
    } else {
        if (!a->point_list) return GSR_ERR_INVALID_ARG;
        uint32_t first = 0;
        GSR_HIP_TRY(hipMemcpyAsync(&first, a->point_list, sizeof(first), hipMemcpyDeviceToHost, stream));
        GSR_HIP_TRY(hipStreamSynchronize(stream));
        if (first == GSR_LISTS_SKIPPED_STAMP) return GSR_ERR_INVALID_ARG;
    }
    // (with sums_f64 the float arrays are written once, by the kernel that rounds the double sums: no clearing)
    if (!wide || nothing_rendered) {
        if (a->dL_dmean2D) GSR_HIP_TRY(hipMemsetAsync(a->dL_dmean2D, 0, sizeof(float) * 2 * (size_t)n, stream));
        if (a->dL_dconic_opacity) GSR_HIP_TRY(hipMemsetAsync(a->dL_dconic_opacity, 0, sizeof(float) * 4 * (size_t)n, stream));
        if (a->dL_dcolors) GSR_HIP_TRY(hipMemsetAsync(a->dL_dcolors, 0, sizeof(float) * 3 * (size_t)n, stream));
        if (a->dL_dcov2D) GSR_HIP_TRY(hipMemsetAsync(a->dL_dcov2D, 0, sizeof(float) * 4 * (size_t)n, stream));
    }
    if (nothing_rendered) {
        // R == 0: no Gaussian reached a pixel (the forward call left even the tile ranges unwritten, GSCuda.cu:775-778)
        if (a->dL_dcov3D) GSR_HIP_TRY(hipMemsetAsync(a->dL_dcov3D, 0, sizeof(float) * 6 * (size_t)n, stream));
        if (a->dL_dshs) GSR_HIP_TRY(hipMemsetAsync(a->dL_dshs, 0, sizeof(float) * 48 * (size_t)n, stream));
        if (a->dL_dmeans3D) GSR_HIP_TRY(hipMemsetAsync(a->dL_dmeans3D, 0, sizeof(float) * 4 * (size_t)n, stream));
        if (a->dL_dscales) GSR_HIP_TRY(hipMemsetAsync(a->dL_dscales, 0, sizeof(float) * 4 * (size_t)n, stream));
        if (a->dL_drotations) GSR_HIP_TRY(hipMemsetAsync(a->dL_drotations, 0, sizeof(float) * 4 * (size_t)n, stream));
        return GSR_OK;
    }
    if (profile) GSR_HIP_TRY(hipEventRecord(g_bw_ev[0], stream));
    RenderBackwardParams r;
    r.ranges = reinterpret_cast<const uint2*>(a->ranges);
    r.point_list = lists_written ? a->point_list : nullptr;
    r.feed = feed;
    r.means2D = reinterpret_cast<const float2*>(a->means2D);
    r.colors = a->colors;
    r.conic_opacity = reinterpret_cast<const float4*>(a->conic_opacity);
    r.final_t = a->final_t;
    r.n_contrib = a->n_contrib;
    r.background = a->background;
    r.dL_dout = a->dL_dout_color;
    r.dL_dmean2D = a->dL_dmean2D;
    r.dL_dconic_opacity = a->dL_dconic_opacity;
    r.dL_dcolors = a->dL_dcolors;
    r.dL_dcov2D = a->dL_dcov2D;
    r.sums64 = a->sums_f64;
    r.dims = d;
    r.num_tiles = (d.row_end - d.row_begin) * d.grid_x;
    // (the tiles that were slow in the forward blend are the slow ones here: they go first, as they did there)
    r.tile_order = have_receipt ? tile_order_of_call(a->receipt, d.row_begin, d.row_end) : nullptr;
    // (all blocks of the frame, also in a sharded call: the blocks outside the band were not walked)
    const unsigned acc_wgs = (unsigned)(((d.grid_x + kBW - 1) / kBW) * ((d.grid_y + kBH - 1) / kBH)) * kAccParts;
    if (r.num_tiles > 0) {
        if (from_blocks && feed.acc) {
            hipLaunchKernelGGL(zero_block_acc_kernel, dim3(acc_wgs), dim3(256), 0, stream, feed);
            GSR_LAUNCH_CHECK("zero_block_acc_kernel");
        }
        hipLaunchKernelGGL(render_backward_kernel, dim3((unsigned)patch_workgroups(d.grid_x, d.row_end - d.row_begin)), dim3(kWave), 0, stream, r);
        GSR_LAUNCH_CHECK("render_backward_kernel");
        if (from_blocks && feed.acc) {
            hipLaunchKernelGGL(flush_block_acc_kernel, dim3(acc_wgs), dim3(256), 0, stream, feed, a->dL_dmean2D, a->dL_dconic_opacity,
                               a->dL_dcolors, a->dL_dcov2D, a->sums_f64);
            GSR_LAUNCH_CHECK("flush_block_acc_kernel");
        }
    }
    if (profile) GSR_HIP_TRY(hipEventRecord(g_bw_ev[1], stream));
    if (chain || wide) {
        PreprocessBackwardParams q;
        q.n = n;
        q.means3D = reinterpret_cast<const float4*>(a->means3D);
        q.cov3D = a->cov3D;
        q.radii = a->radii;
        q.view = a->view_matrix;
        q.tan_fovx = a->tan_fovx; q.tan_fovy = a->tan_fovy;
        q.focal = (float)a->height / (2.0f * a->tan_fovy);
        q.dL_dconic_opacity = reinterpret_cast<const float4*>(a->dL_dconic_opacity);
        q.dL_dcov2D = reinterpret_cast<const float4*>(a->dL_dcov2D);
        q.sums64 = a->sums_f64;
        q.out_mean2D = reinterpret_cast<float2*>(a->dL_dmean2D);
        q.out_conic_opacity = reinterpret_cast<float4*>(a->dL_dconic_opacity);
        q.out_colors = a->dL_dcolors;
        q.out_cov2D = reinterpret_cast<float4*>(a->dL_dcov2D);
        q.dL_dcolors = a->dL_dcolors;
        q.dL_dcov3D = a->dL_dcov3D;
        q.chain = chain ? 1 : 0;                               // (without the chain the kernel only rounds the double sums)
        q.dL_dshs = chain ? a->dL_dshs : nullptr;
        q.proj = a->proj_matrix;
        q.scales = reinterpret_cast<const float4*>(a->scales);
        q.rotations = reinterpret_cast<const float4*>(a->rotations);
        q.scale_modifier = a->scale_modifier;
        q.width = a->width; q.height = a->height;
        q.dL_dmean2D = reinterpret_cast<const float2*>(a->dL_dmean2D);
        q.dL_dmeans3D = reinterpret_cast<float4*>(a->dL_dmeans3D);
        q.dL_dscales = reinterpret_cast<float4*>(a->dL_dscales);
        q.dL_drotations = reinterpret_cast<float4*>(a->dL_drotations);
        q.inria = inria ? 1 : 0;
        q.sh_deg = a->sh_dims < 0 ? 0 : (a->sh_dims > 3 ? 3 : a->sh_dims);
        q.focal_x = (float)a->width / (2.0f * a->tan_fovx);
        q.focal_y = (float)a->height / (2.0f * a->tan_fovy);
        q.w_eps = 0.0000001f;
        q.shs = a->shs; q.cam_pos = a->cam_pos; q.clamped = a->clamped;
        if (inria) hipLaunchKernelGGL(preprocess_backward_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, q);
        else hipLaunchKernelGGL(preprocess_backward_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, q);
        GSR_LAUNCH_CHECK("preprocess_backward_kernel");
    }
    if (profile) {
        GSR_HIP_TRY(hipEventRecord(g_bw_ev[2], stream));
        GSR_HIP_TRY(hipStreamSynchronize(stream));
        GSR_HIP_TRY(hipEventElapsedTime(&a->stage_ms[0], g_bw_ev[0], g_bw_ev[1]));
        GSR_HIP_TRY(hipEventElapsedTime(&a->stage_ms[1], g_bw_ev[1], g_bw_ev[2]));
    }
    return GSR_OK;
}

extern "C" int gsr_backward(gsr_backward_args* a) { return record_error(backward_impl(a)); }
