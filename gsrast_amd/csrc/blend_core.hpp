// Shared core of the per-tile compositing kernels (blend.hip: records come from the sorted point list;
// blockbin.hip: records come straight from the block lists). One wavefront per 16 x 16 tile, four pixels
// per lane: lane l owns pixels (x = l & 15, y = (l >> 4) + 4 k), k = 0..3.
//
// Semantics follow reference apps/gsrast/gscuda/GSCuda.cu:623-676 (the loop of renderCUDA); see blend.hip.
#pragma once
#include "gsr_common.hpp"

namespace gsr {

constexpr int kBatch = 256;

// ---- records that cannot touch the tile ----------------------------------------------------------------------
// A tile's list holds every Gaussian whose RECTANGLE covers the tile (GSCuda.cu:449-474), and the rectangle is
// generous: its y extent is ceil(3 * cov.z) pixels (no square root, GSCuda.cu:352). On the frames measured here
// 73-93 % of the records staged by a tile light none of its pixels: every pixel fails `alpha < 1/255`
// (GSCuda.cu:646). Such a record changes nothing but the numbering of the records behind it, so it is dropped
// when the batch is staged (one lane per record), not walked by the compositing loop (one wave per record):
//   power(d) = -0.5 (A dx^2 + C dy^2) - B dx dy is concave for a positive definite conic, so its maximum over
//   the tile's pixel rectangle is 0 if the centre lies inside and otherwise sits on an edge facing the centre, where
//   it is a 1-D parabola with a closed-form clamped maximiser. A pixel can pass the alpha test only if
//   opacity * exp(power) >= 1/255, i.e. power >= -ln(255 * opacity).
// The bound is evaluated over the real rectangle (a superset of the integer pixel centres) with a margin for
// rounding (relative to the magnitude of the terms that cancel), so a dropped record provably contributes to no
// pixel; anything odd (conic not positive definite, NaN, opacity <= 0 or NaN) keeps the record.
struct TileBox { float x_lo, x_hi, y_lo, y_hi; };      // pixel-centre rectangle of the tile, clipped to the image

__device__ __forceinline__ TileBox tile_box(int tx, int ty, int width, int height) {
    TileBox b;
    b.x_lo = (float)(tx * kTile); b.x_hi = (float)min(tx * kTile + kTile - 1, width - 1);
    b.y_lo = (float)(ty * kTile); b.y_hi = (float)min(ty * kTile + kTile - 1, height - 1);
    return b;
}

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// upper bound of power on the segment {dx = e, dy in [lo, hi]} (A: coefficient of the fixed coordinate, C: of the free one)
__device__ __forceinline__ float edge_power_bound(float A, float B, float C, float inv_c, float e, float lo, float hi) {
    const float d = clampf(-(B * e) * inv_c, lo, hi);
    const float t0 = 0.5f * A * e * e, t1 = 0.5f * C * d * d, t2 = B * e * d;
    return -(t0 + t1) - t2 + 1e-5f * (t0 + t1 + fabsf(t2));      // + rounding of the three terms that cancel
}

__device__ __forceinline__ bool record_misses_tile(const float2 xy, const float4 co, const TileBox& box) {
    const float A = co.x, B = co.y, C = co.z;
    const bool pd = A > 0.0f && C > 0.0f && A * C - B * B > 0.0f && fabsf(xy.x) < 1e30f && fabsf(xy.y) < 1e30f;
    // d = centre - pixel (GSCuda.cu:626): d ranges over [centre - hi, centre - lo]
    const float dx_lo = xy.x - box.x_hi, dx_hi = xy.x - box.x_lo, dy_lo = xy.y - box.y_hi, dy_hi = xy.y - box.y_lo;
    // the point of the rectangle nearest to the centre (0 where the centre lies inside that range): the maximum over the
    // rectangle sits on an edge through it that faces the centre - the vertical one, the horizontal one, or either
    const float ex = clampf(0.0f, dx_lo, dx_hi), ey = clampf(0.0f, dy_lo, dy_hi);
    const bool inside = ex == 0.0f && ey == 0.0f;
    // v_rcp_f32 (1 ulp) only places the clamped maximiser, where the parabola is flat
    const float inv_a = __builtin_amdgcn_rcpf(A), inv_c = __builtin_amdgcn_rcpf(C);
    const float best = fmaxf(edge_power_bound(A, B, C, inv_c, ex, dy_lo, dy_hi), edge_power_bound(C, B, A, inv_a, ey, dx_lo, dx_hi));
    const float threshold = -__logf(255.0f * co.w);             // NaN for opacity <= 0 or NaN: the comparison below is then false
    return pd && !inside && (best + 2e-3f < threshold);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- the exponential of GSCuda.cu:645 -------------------------------------------------------------------------------
// alpha = min(0.99, opacity * exp(power)) feeds two hard tests (alpha >= 1/255, T (1 - alpha) >= cut-off) and, through T,
// every later record of the pixel: an exponential that is 3 ulp off (v_exp_f32 on a single-rounded power * log2 e: the
// product alone carries |power| * 2^-24 into the result) lets T drift by 1e-6 over a deep list and flips a threshold for a
// few pixels per 8 M (round 3: 6 pixels of the 4K parity frame beyond 1e-4, up to 1.3e-3). GSR_EXP_VARIANT:
//   2 (default) the float exponential as glibc computes it (sysdeps/ieee754/flt-32/e_expf.c since 2.27, from ARM's optimized
//     routines; error 0.502 ulp): double arithmetic, exp(x) = 2^(k/32) * p(r), k = round(32 x / ln 2), r = 32 x / ln 2 - k,
//     2^(i/32) from a 32-entry table, p of degree 3, rounded to float once. Restated here constant for constant (checked
//     against libm on 2e8 arguments on the host: no difference; on the device: tests/test_gpu_parity.py), so alpha, T and every
//     decision are bit for bit the CPU oracle's. The f64 pipe issues at the f32 rate on this part: 14 more issues per strip.
//   1 v_exp_f32 on a two-term product (hi = x * log2e_hi rounded, lo = the exact remainder + x * log2e_lo), e * (1 + lo ln 2):
//     ~1.2 ulp, 5 more issues.    0: __expf, as round 3 shipped.
#ifndef GSR_EXP_VARIANT
#define GSR_EXP_VARIANT 2
#endif
__constant__ const unsigned long long kExp2Table[32] = {       // asuint64(2^(i/32)) - (i << 47)
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
    0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
    0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
    0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
    0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
// tab: the table above in LDS (exp_table_init). Arguments below -104 give 0 (glibc: underflow); NaN gives NaN; arguments
// above 88.7 are not met here (the power is tested against 0 first, and nothing reads the lanes that fail).
__device__ __forceinline__ void exp_table_init(unsigned long long* tab, int lane) {
    if (lane < 32) tab[lane] = kExp2Table[lane];
}
__device__ __forceinline__ float exp_ref(float x, const unsigned long long* tab) {
#if GSR_EXP_VARIANT == 2
    constexpr double kN = 32.0, kInvLn2N = 0x1.71547652b82fep+0 * kN, kShift = 0x1.8p+52;
    constexpr double kC0 = 0x1.c6af84b912394p-5 / kN / kN / kN, kC1 = 0x1.ebfce50fac4f3p-3 / kN / kN, kC2 = 0x1.62e42ff0c52d6p-1 / kN;
    // (below -104 the result is 0, as glibc's underflow path returns: exp(-104) already rounds to it)
    float xc;
    asm("v_max_f32 %0, %1, %2" : "=v"(xc) : "v"(x), "s"(-104.0f));       // (fmaxf would canonicalise x first: one more issue)
    const double z = kInvLn2N * (double)xc;
    double kd = z + kShift;
    const uint32_t ki = (uint32_t)__double_as_longlong(kd);         // k: its low bits pick the table entry, the rest is the exponent
    kd -= kShift;
    const double r = z - kd;
    // 2^(k/32) = the table entry with k << 47 added: that only reaches the high word
    const unsigned long long t = tab[ki & 31u];
    const double sc = __hiloint2double((int)((uint32_t)(t >> 32) + (ki << 15)), (int)(uint32_t)t);
    const double zz = __builtin_fma(kC0, r, kC1);
    const double r2 = r * r;
    double y = __builtin_fma(kC2, r, 1.0);
    y = __builtin_fma(zz, r2, y);
    y = y * sc;
    return (float)y;
#elif GSR_EXP_VARIANT == 1
    (void)tab;
    constexpr float kHi = 1.44269502162933349609375f;          // log2(e) rounded to float
    constexpr float kLo = 1.925963033500011e-8f;               // log2(e) - kHi
    const float hi = x * kHi;
    const float lo = __builtin_fmaf(x, kHi, -hi) + x * kLo;
    const float e = __builtin_amdgcn_exp2f(hi);
    return __builtin_fmaf(e, lo * 0.693147182464599609375f, e);
#else
    (void)tab;
    return __expf(x);
#endif
}

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
    // a finished pixel's row coordinate is NaN: its power is NaN and fails the candidate range test without a test of its own
    const float qnan = __builtin_nanf("");
    if (s.done[0]) s.fy01.x = qnan;
    if (s.done[1]) s.fy01.y = qnan;
    if (s.done[2]) s.fy23.x = qnan;
    if (s.done[3]) s.fy23.y = qnan;
}

__device__ __forceinline__ bool tile_lanes_all_done(const TileLanes& s) {
    return __ballot((s.done[0] & s.done[1] & s.done[2] & s.done[3]) == 0u) == 0ull;
}

// Composites the `chunk` records staged in wave-private LDS (s_rgb[j].w carries, as bits, the record's 1-based
// position in the tile's list: the contributor number of GSCuda.cu:624 — records dropped at staging leave gaps).
// Returns true as soon as every pixel of the tile is finished.
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kFilterSlack = 0.25f;      // in units of the power: covers the rounding of terms up to 4e6 in magnitude; records that can exceed 1e6 are not filtered (stage_and_composite)
// A cheap FILTER in front of the reference's arithmetic. Of the records a tile stages, most reach only a part of its four
// 16 x 4 strips, and finding that out cost as much as compositing: the reference's power (GSCuda.cu:634) for all four
// pixels of a lane, 18 issues, and twelve compares. The filter evaluates the power a second way: the staged conic is
// pre-scaled (stage_and_composite), q = (-0.5 log2(e) A, -log2(e) B, -0.5 log2(e) C, the record's floor), so that
//   log2(e) * power = q.x dx^2 + dy (q.z dy + q.y dx)
// is four scalar operations for the dx terms and three packed ones per pair of rows (a subtract, two v_pk_fma_f32); a
// finished pixel has a NaN row coordinate (tile_lanes_init), so the range test "floor <= power <= 0" is the whole
// candidate test, two compares per pixel whose results stay lane masks in scalar registers. Its floor — the power below
// which THIS record's opacity cannot reach alpha = 1/255 — has margins, and a record fails it only if every lane is outside
// [floor, 0], so it decides nothing the reference's tests decide: for the pairs of strips that have a candidate the power is evaluated again in the
// reference's operation order (unfused; raw conic from s_raw), and alpha, the 1/255 test, T and the cut-off test follow
// from THAT — T and every decision are bit for bit what they were. (Feeding alpha from the fused evaluation was built
// first: 25 % faster still, but alpha then differs by 1e-7 relative, T drifts by 1e-6 to 1e-5 over a deep list, and a
// pixel whose T (1 - alpha) lands that close to the cut-off stops one record earlier or later: 59 to 109 pixels of the
// 8.3 M of the 4K parity frame beyond 1e-4 instead of fewer than 20, `gpurun_out/r3e`, `r3f`.) Only the colour sums
// take a fused form, three multiply-adds on alpha T — they feed no test, and move the pixel by a few 1e-8.
// (A record whose power terms are too large for the filter's slack is staged with a zero filter conic and no floor: every
// unfinished pixel is its candidate, stage_and_composite.)
// One staged record (slot j) against the wave's pixels; true: some pixel finished on it (the caller then looks whether all have).
__device__ __forceinline__ bool composite_record(TileLanes& s, const float2* s_xy, const float4* s_co, const float4* s_rgb,
                                                 const float4* s_raw, uint32_t j, float t_cutoff,
                                                 const unsigned long long* exp_tab) {
    const float qnan = __builtin_nanf("");
    constexpr float kAlphaMin = 1.0f / 255.0f;
    const float2 xy = s_xy[j];
    const float4 q = s_co[j];
    const float dx = xy.x - s.fx;
    const float h0 = (q.x * dx) * dx;
    const float g = q.y * dx;
    f32x2 gy; gy.x = xy.y; gy.y = xy.y;
    f32x2 gg; gg.x = g; gg.y = g;
    f32x2 hh; hh.x = h0; hh.y = h0;
    f32x2 cz; cz.x = q.z; cz.y = q.z;
    const f32x2 dy01 = gy - s.fy01, dy23 = gy - s.fy23;
    const f32x2 fw01 = __builtin_elementwise_fma(dy01, __builtin_elementwise_fma(cz, dy01, gg), hh);
    const f32x2 fw23 = __builtin_elementwise_fma(dy23, __builtin_elementwise_fma(cz, dy23, gg), hh);
    const float filter[4] = {fw01.x, fw01.y, fw23.x, fw23.y};
    // A pixel can pass alpha >= 1/255 only if power >= -ln(255 opacity): the record's own floor (q.w, set at staging with
    // its margins: a faint record's strips are candidates only where it is bright enough to count, and only candidates pay
    // for the exponential). The filter's window is wider than the reference's tests at both ends by kFilterSlack: the two
    // evaluations of the power differ by rounding, 6e-8 of the largest of three terms that may cancel — a record whose terms
    // can exceed what the slack covers has a zero filter conic — and a pixel the reference would accept must never be
    // filtered out. The reference's own "power > 0" and "alpha < 1/255" tests follow below, on its
    // own arithmetic.
    unsigned long long cand[4];       // lane masks (every lane of the wave is active here); false for a finished pixel: NaN
#pragma unroll
    for (int k = 0; k < 4; ++k)
        cand[k] = __ballot(filter[k] <= kFilterSlack * kLog2e) & __ballot(filter[k] >= q.w);
    if ((cand[0] | cand[1] | cand[2] | cand[3]) == 0ull) return false;
    const float4 col = s_rgb[j];
    const float4 raw = s_raw[j];
    const uint32_t contributor = __float_as_uint(col.w);
    // the reference's operation order from here on (this file is compiled without contraction)
    const float t1 = (raw.x * dx) * dx, bdx = raw.y * dx;
    unsigned long long newly_done = 0ull;
#pragma unroll
    for (int pair = 0; pair < 2; ++pair) {
        if ((cand[2 * pair] | cand[2 * pair + 1]) == 0ull) continue;
        // (finished pixels: their dy is NaN, and so is their power — they are no candidates)
        const f32x2 dy = pair == 0 ? dy01 : dy23;
        const f32x2 pw = -0.5f * (t1 + (raw.z * dy) * dy) - bdx * dy;
        // (the compiler sinks each into its strip's branch; forcing both up front, interleaved, measured slower on every
        // frame: the loop is bound by instruction issue, not by the latency of the double chain)
        const float e0 = exp_ref(pw.x, exp_tab), e1 = exp_ref(pw.y, exp_tab);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = 2 * pair + h;
            if (cand[k] == 0ull) continue;                // nobody in this strip sees the record
            const float power = h == 0 ? pw.x : pw.y;
            const float alpha = fminf(0.99f, raw.w * (h == 0 ? e0 : e1));
            const unsigned long long live = cand[k] & ~__ballot(power > 0.0f) & ~__ballot(alpha < kAlphaMin);
            const float test = s.T[k] * (1.0f - alpha);
            const unsigned long long stop = live & __ballot(test < t_cutoff);
            if (__builtin_amdgcn_inverse_ballot_w64(live & ~stop)) {
                const float w = alpha * s.T[k];
                s.cr[k] = __builtin_fmaf(col.x, w, s.cr[k]);
                s.cg[k] = __builtin_fmaf(col.y, w, s.cg[k]);
                s.cb[k] = __builtin_fmaf(col.z, w, s.cb[k]);
                s.T[k] = test;
                s.last[k] = contributor;
            }
            if (stop != 0ull) {
                if (__builtin_amdgcn_inverse_ballot_w64(stop)) {
                    s.done[k] = 1u;
                    if (k == 0) s.fy01.x = qnan;
                    if (k == 1) s.fy01.y = qnan;
                    if (k == 2) s.fy23.x = qnan;
                    if (k == 3) s.fy23.y = qnan;
                }
                newly_done |= stop;
            }
        }
    }
    return newly_done != 0ull;
}

__device__ __forceinline__ bool composite_staged(TileLanes& s, const float2* s_xy, const float4* s_co, const float4* s_rgb,
                                                 const float4* s_raw, uint32_t chunk, float t_cutoff,
                                                 const unsigned long long* exp_tab) {
    for (uint32_t j = 0; j < chunk; ++j)
        if (composite_record(s, s_xy, s_co, s_rgb, s_raw, j, t_cutoff, exp_tab) && tile_lanes_all_done(s)) return true;
    return false;
}

// Stages the records of one batch of up to 64 list entries and composites them. Lane l holds entry l of the batch
// (`present`: the lane has an entry; `id`: its Gaussian; `number`: its 1-based position in the tile's list).
// `first_number`..: the batch covers list positions [first_number - 1, first_number - 1 + count). The staged-record
// count keeps the reference's granularity: a batch of 256 list positions is "staged" when its first position is
// reached with some pixel unfinished (GSCuda.cu:595-609), whether or not its records survive the footprint test.
struct TileFeed {
    const float2* means2D;
    const float* colors;
    const float4* conic_opacity;
    TileBox box;
    uint32_t total;                 // records in the tile's list
    float t_cutoff;
    uint32_t dc_stride = 0;         // 0: `colors` are colours, 12 bytes apart. 48: `colors` is the caller's SH array and a record's
                                    // colour is 0.5 + 0.4 DC (GSCuda.cu:362-366, the preprocess's two operations) — geomState.rgb
                                    // is then being written BESIDE this blend (api.hip), for whoever reads the chunk afterwards
};

// One batch of up to 64 list entries on its way through the wave: lane l holds entry l. The loads of a batch are
// issued ahead of its use (ids two batches ahead, records one batch ahead: see the kernels), so the three dependent
// round trips id -> record -> colour are paid while earlier batches are being composited, not once per batch.
struct RecordBatch {
    unsigned long long mask = 0;   // lanes that hold an entry (wave-uniform)
    uint32_t pos = 0;              // list position of the batch's first entry (wave-uniform)
    bool valid = false;            // wave-uniform: false = the list is exhausted
    uint32_t id = 0;               // per lane: Gaussian index
    float2 xy;                     // per lane: its record
    float4 co;
    __device__ __forceinline__ uint32_t count() const { return (uint32_t)__popcll(mask); }
    __device__ __forceinline__ bool present() const { return __builtin_amdgcn_inverse_ballot_w64(mask); }
    __device__ __forceinline__ uint32_t rank() const {
        return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    }
};

__device__ __forceinline__ void fetch_records(RecordBatch& b, const TileFeed& f) {
    b.xy = make_float2(0.0f, 0.0f);
    b.co = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (b.valid && b.present()) { b.xy = f.means2D[b.id]; b.co = f.conic_opacity[b.id]; }
}

// One wave's staging area in LDS (wave-private: no barrier): the survivors of a batch, compacted.
struct StagedRecords {
    float2 xy[kWave];
    float4 co[kWave];              // conic as the compositing loop wants it (pre-scaled), opacity
    float4 rgb[kWave];             // colour, and (as bits) the record's 1-based position in the tile's list
    float4 raw[kWave];             // conic + opacity as fetched: the reference-order evaluation of the lanes near a threshold
    unsigned long long exp_tab[32];   // exp_ref's table (exp_table_init at the top of the kernel)
};

// Stages the survivors of one batch (the footprint test, one lane per record) compacted into `st`; returns their number.
// *before_boundary (optional): how many of them lie in front of the batch's first multiple-of-256 list position.
__device__ __forceinline__ uint32_t stage_batch(const TileFeed& f, StagedRecords& st, const RecordBatch& b, uint32_t* before_boundary = nullptr) {
    float2* const s_xy = st.xy;
    float4* const s_co = st.co;
    float4* const s_rgb = st.rgb;
    float4* const s_raw = st.raw;
    const bool present = b.present();
    const uint32_t rank = b.rank(), pos = b.pos;
    const bool keep = present && !record_misses_tile(b.xy, b.co, f.box);
    const unsigned long long m2 = __ballot(keep);
    if (keep) {
        const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0u));
        const float* c = f.colors + (f.dc_stride != 0u ? (size_t)f.dc_stride : (size_t)3) * (size_t)b.id;
        s_xy[slot] = b.xy;
        // The filter's evaluation of the power differs from the reference's by the rounding of three terms that may cancel:
        // a few 6e-8 of their magnitude, bounded here over the tile.
        const float dxm = fmaxf(fabsf(b.xy.x - f.box.x_lo), fabsf(b.xy.x - f.box.x_hi)), dym = fmaxf(fabsf(b.xy.y - f.box.y_lo), fabsf(b.xy.y - f.box.y_hi));
        const float terms = fabsf(b.co.x) * dxm * dxm + fabsf(b.co.z) * dym * dym + 2.0f * fabsf(b.co.y) * dxm * dym;
        // the filter's floor for this record, in units of log2: -ln(255 opacity) less the margins (1e-3 for the rounding of the
        // exponential and the product, and the filter's own rounding). Opacity <= 0: +inf, no
        // candidate (alpha <= 0 fails the 1/255 test); NaN opacity: -inf (the reference's min(0.99, NaN) is 0.99: it counts).
        const float p0 = -__logf(255.0f * b.co.w);
        const float floor2 = b.co.w != b.co.w ? -__builtin_inff() : (b.co.w <= 0.0f ? __builtin_inff() : (p0 - 1e-3f - 1e-6f * terms) * kLog2e);
        // The filter's slack covers the rounding of power terms up to 4e6 (kFilterSlack / 6e-8). A record whose terms can be
        // larger somewhere on the tile — a screen-filling needle seen along its axis at 4K: conic entries up to 3.3, |d| in the
        // thousands — gets a zero filter conic and no floor: its filter value is 0 for every unfinished pixel (NaN for the
        // finished ones, as always), all of them are candidates, and the reference-order evaluation decides alone.
        const bool filtered = terms < 1.0e6f;                                   // (NaN: not filtered)
        s_co[slot] = filtered ? make_float4((-0.5f * kLog2e) * b.co.x, -kLog2e * b.co.y, (-0.5f * kLog2e) * b.co.z, floor2)
                              : make_float4(0.0f, 0.0f, 0.0f, -__builtin_inff());
        s_raw[slot] = b.co;
        float c0 = c[0], c1 = c[1], c2 = c[2];
        if (f.dc_stride != 0u) { c0 = 0.5f + 0.4f * c0; c1 = 0.5f + 0.4f * c1; c2 = 0.5f + 0.4f * c2; }
        s_rgb[slot] = make_float4(c0, c1, c2, __uint_as_float(pos + rank + 1u));
    }
    if (before_boundary) {
        const uint32_t boundary = (pos + (uint32_t)kBatch - 1u) & ~((uint32_t)kBatch - 1u);   // first multiple of 256 >= pos
        *before_boundary = (uint32_t)__popcll(__ballot(keep && rank < boundary - pos));
    }
    return (uint32_t)__popcll(m2);
}

// exp_tab: exp_ref's table in LDS (null: the one in `st`)
__device__ __forceinline__ bool stage_and_composite(TileLanes& s, const TileFeed& f, StagedRecords& st,
                                                    const RecordBatch& b, unsigned long long& staged, const unsigned long long* exp_tab = nullptr) {
    if (!exp_tab) exp_tab = st.exp_tab;
    uint32_t before = 0;
    const uint32_t kept = stage_batch(f, st, b, &before);
    const uint32_t pos = b.pos, count = b.count();
    // wave-private LDS: the writes above and the reads of composite_staged are ordered inside the wave
    bool all_done = false;
    uint32_t first = 0;
    const uint32_t boundary = (pos + (uint32_t)kBatch - 1u) & ~((uint32_t)kBatch - 1u);   // first multiple of 256 >= pos
    if (boundary < pos + count) {
        // records in front of the boundary first; then the reference would test "whole tile done" and stage the next 256
        if (before) all_done = composite_staged(s, st.xy, st.co, st.rgb, st.raw, before, f.t_cutoff, exp_tab);
        if (all_done) return true;
        staged += min((uint32_t)kBatch, f.total - boundary);
        first = before;
    }
    if (kept > first) all_done = composite_staged(s, st.xy + first, st.co + first, st.rgb + first, st.raw + first, kept - first, f.t_cutoff, exp_tab);
    return all_done;
}

// ---- deep tiles: four waves per tile, one walk (blend.hip) -------------------------------------------------------------
// A round is 256 list entries; wave w fetches, culls and stages entries [64 w, 64 w + 64) into segment w, and composites ONE
// 16 x 4 strip — one pixel per lane — over the four segments in list order, taking the slots that can reach its strip.
struct DeepSegment {               // the survivors of a wave's 64 list entries, compacted
    float4 head[kWave];            // centre x, y | the filter's floor in units of log2 (-inf: none) | its 1-based list position (bits)
    float4 filt[kWave];            // the filter's conic (pre-scaled as StagedRecords::co; zeros: the record goes unfiltered), w unused
    float4 raw[kWave];             // conic + opacity as fetched
    float4 rgb[kWave];             // colour (w unused)
    uint32_t touch[kWave];         // strips the record can reach at all (bit k: strip k)
};
// Which strips a record can reach: strip k is out of reach when even the power's maximum over dx, -0.5 dy^2 det / A, stays
// below the record's floor for every row of the strip, |dy| > sqrt(2 (ln(255 opacity) + margin) A / det) — the margin of
// record_misses_tile plus the rounding of the power's terms over the tile, the determinant rounded down; anything odd: all four.
template <int STRIPS>
__device__ __forceinline__ uint32_t strips_in_reach(const float2 xy, const float4 co, float p0, float terms, bool filtered, float tile_y0, int height) {
    constexpr int kRows = kTile / STRIPS;          // rows of a strip: 4 (four waves per tile), 2 or 1 (eight, sixteen: see blend.hip)
    const float A = co.x, B = co.y, C = co.z;
    const float ac = A * C, det_lo = ac - B * B - 4e-7f * ac;
    const float reach2 = 2.0f * (2e-3f + 1e-5f * terms - p0) * A / det_lo * 1.0001f;     // (p0 = -ln(255 opacity))
    if (!(A > 0.0f && C > 0.0f && det_lo > 0.0f && reach2 >= 0.0f && reach2 < 1e12f && filtered)) return (1u << STRIPS) - 1u;
    const float reach = __builtin_sqrtf(reach2) + 0.01f;
    const float y_lo = xy.y - reach, y_hi = xy.y + reach;
    uint32_t bits = 0u;
#pragma unroll
    for (int k = 0; k < STRIPS; ++k) {
        const float r_lo = tile_y0 + (float)(kRows * k), r_hi = fminf(r_lo + (float)(kRows - 1), (float)(height - 1));
        if (y_lo <= r_hi && y_hi >= r_lo) bits |= 1u << k;
    }
    return bits;
}
// (the footprint test, the floor and the margins are stage_batch's)
template <int STRIPS>
__device__ __forceinline__ uint32_t stage_batch_deep(const TileFeed& f, DeepSegment& seg, const RecordBatch& b, int height) {
    const bool present = b.present();
    const uint32_t rank = b.rank(), pos = b.pos;
    const bool keep = present && !record_misses_tile(b.xy, b.co, f.box);
    const unsigned long long m2 = __ballot(keep);
    if (keep) {
        const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0u));
        const float* c = f.colors + (f.dc_stride != 0u ? (size_t)f.dc_stride : (size_t)3) * (size_t)b.id;
        const float dxm = fmaxf(fabsf(b.xy.x - f.box.x_lo), fabsf(b.xy.x - f.box.x_hi)), dym = fmaxf(fabsf(b.xy.y - f.box.y_lo), fabsf(b.xy.y - f.box.y_hi));
        const float terms = fabsf(b.co.x) * dxm * dxm + fabsf(b.co.z) * dym * dym + 2.0f * fabsf(b.co.y) * dxm * dym;
        const float p0 = -__logf(255.0f * b.co.w);
        const float floor2 = b.co.w != b.co.w ? -__builtin_inff() : (b.co.w <= 0.0f ? __builtin_inff() : (p0 - 1e-3f - 1e-6f * terms) * kLog2e);
        const bool filtered = terms < 1.0e6f;                                   // (NaN: not filtered)
        seg.head[slot] = make_float4(b.xy.x, b.xy.y, filtered ? floor2 : -__builtin_inff(), __uint_as_float(pos + rank + 1u));
        seg.filt[slot] = filtered ? make_float4((-0.5f * kLog2e) * b.co.x, -kLog2e * b.co.y, (-0.5f * kLog2e) * b.co.z, 0.0f)
                                  : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        seg.raw[slot] = b.co;
        float c0 = c[0], c1 = c[1], c2 = c[2];
        if (f.dc_stride != 0u) { c0 = 0.5f + 0.4f * c0; c1 = 0.5f + 0.4f * c1; c2 = 0.5f + 0.4f * c2; }
        seg.rgb[slot] = make_float4(c0, c1, c2, 0.0f);
        seg.touch[slot] = strips_in_reach<STRIPS>(b.xy, b.co, p0, terms, filtered, f.box.y_lo, height);
    }
    return (uint32_t)__popcll(m2);
}

struct StripLanes {                // one pixel per lane: lane l owns (x = l & 15, y = rows strip + (l >> 4)) of the tile (rows = 4; 2 or 1: the upper lanes idle)
    int px, py;
    float fx, fy;                  // (fy: NaN once the pixel is finished, as TileLanes' row coordinates)
    bool inside;
    float T, cr, cg, cb;
    uint32_t last;
};
__device__ __forceinline__ void strip_lanes_init(StripLanes& s, int tx, int ty, int strip, int lane, int width, int height, int rows = 4) {
    s.px = tx * kTile + (lane & 15);
    s.py = ty * kTile + rows * strip + (lane >> 4);
    s.fx = (float)s.px;
    s.inside = s.px < width && s.py < height && (lane >> 4) < rows;
    s.fy = s.inside ? (float)s.py : __builtin_nanf("");
    s.T = 1.0f; s.cr = s.cg = s.cb = 0.0f; s.last = 0u;
}
__device__ __forceinline__ bool strip_lanes_all_done(const StripLanes& s) { return __ballot(s.fy == s.fy) == 0ull; }
// composite_record for one strip, over the slots of `slots` (a lane mask over the segment's slots) in order: the same filter,
// the same reference-order evaluation, the same tests, per pixel the same floats in the same order (scalar single-precision
// operations round as the packed ones do; nothing here is contracted). One flat loop — the scalar unit of a CU serves its four
// SIMDs, and with eight of these waves on every SIMD the loop's scalar instructions (45 % of all it issues) are what it
// waits for: no call in it, one way through per record. *done_at: the list position of the record the strip's last
// pixel finished on.
__device__ __forceinline__ bool composite_strip(StripLanes& s, const DeepSegment& seg, unsigned long long slots, float t_cutoff,
                                                const unsigned long long* exp_tab, uint32_t* done_at) {
    constexpr float kAlphaMin = 1.0f / 255.0f;
    // (one exit: `finished` is a scalar the loop's condition reads — an early return from inside the masked region made the
    // compiler carry the way out in a vector register and compare it, a round trip through the scalar unit per record)
    bool finished = false;
    while (slots != 0ull && !finished) {
        const uint32_t j = (uint32_t)__builtin_ctzll(slots);
        slots &= slots - 1ull;
        const float4 hd = seg.head[j];
        const float4 q = seg.filt[j];
        const float dx = hd.x - s.fx, dy = hd.y - s.fy;
        // (an unfiltered record: zero conic, floor -inf — every unfinished pixel is its candidate, a finished one's dy is NaN)
        const float fw = __builtin_fmaf(dy, __builtin_fmaf(q.z, dy, q.y * dx), (q.x * dx) * dx);
        const unsigned long long cand = __ballot(fw <= kFilterSlack * kLog2e) & __ballot(fw >= hd.z);
        if (cand != 0ull) {
            const float4 raw = seg.raw[j];
            const float t1 = (raw.x * dx) * dx, bdx = raw.y * dx;
            const float power = -0.5f * (t1 + (raw.z * dy) * dy) - bdx * dy;
            const float alpha = fminf(0.99f, raw.w * exp_ref(power, exp_tab));
            const unsigned long long live = cand & ~(__ballot(power > 0.0f) | __ballot(alpha < kAlphaMin));
            const float test = s.T * (1.0f - alpha);
            const unsigned long long stop = live & __ballot(test < t_cutoff);
            if (__builtin_amdgcn_inverse_ballot_w64(live & ~stop)) {
                const float4 col = seg.rgb[j];
                const float w = alpha * s.T;
                s.cr = __builtin_fmaf(col.x, w, s.cr);
                s.cg = __builtin_fmaf(col.y, w, s.cg);
                s.cb = __builtin_fmaf(col.z, w, s.cb);
                s.T = test;
                s.last = __float_as_uint(hd.w);
            }
            if (stop != 0ull) {
                s.fy = __builtin_amdgcn_inverse_ballot_w64(stop) ? __builtin_nanf("") : s.fy;
                finished = __ballot(s.fy == s.fy) == 0ull;
                *done_at = __float_as_uint(hd.w);          // (read by the caller only if the strip is finished: the last write is the one)
            }
        }
    }
    return finished;
}
__device__ __forceinline__ void strip_lanes_write(const StripLanes& s, int width, int height, const float* __restrict__ background,
                                                  float* __restrict__ final_t, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color) {
    if (!s.inside) return;
    const size_t plane = (size_t)width * (size_t)height;
    const size_t pid = (size_t)s.py * (size_t)width + (size_t)s.px;
    final_t[pid] = s.T;
    n_contrib[pid] = s.last;
    out_color[pid] = s.cr + s.T * background[0];
    out_color[pid + plane] = s.cg + s.T * background[1];
    out_color[pid + 2 * plane] = s.cb + s.T * background[2];
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

// Issue priority of a tile's wave by what is LEFT of its list (re-evaluated batch by batch). A frame lasts as long as its
// deepest tile, every tile's wave is resident from the start (8 160 tiles, 8 per SIMD), and without this a deep tile gets an
// eighth of its SIMD's issue slots until the shallow ones are gone. Used by the blend from the sorted lists, one wave per
// tile, where the lists are long on average (8 192 entries per tile with a list) — the frames of small splats seen from
// outside, whose lists differ widely in length: 0.427 -> 0.416 ms from outside the cloud, 3.29 -> 2.84 ms with faint splats.
// (Shorter lists, eye (0,0,-30): 0.502 -> 0.52 ms with it — not used there.) On frames whose tiles all finish after a few hundred entries of 32 K-entry lists (the bench
// frame, block-fed) any difference in priority only delays somebody's chain of round trips (0.098 -> 0.101 ms), and where
// every tile walks a list of about the same length (bench frame with faint splats) it changes nothing: the block-fed
// blend and the four-waves-per-tile mode do not use it (`gpurun_out`, ab_extras runs after `r3n`).
__device__ __forceinline__ void set_tile_priority(uint32_t remaining) {
    if (remaining > 32768u) __builtin_amdgcn_s_setprio(3);
    else if (remaining > 8192u) __builtin_amdgcn_s_setprio(2);
    else if (remaining > 2048u) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

// Tile of a workgroup. Workgroup b runs on XCD b % 8, and each XCD has its own L2. Tiles dealt in tile order put every
// eighth tile on an XCD: perfectly level, but the eight neighbours of a tile — which gather largely the same Gaussians —
// are on the seven other L2s. Giving every XCD one contiguous eighth of the image measured 6-23 % SLOWER (the image's
// busy rows land on two or three XCDs while the others idle). In between: 64 consecutive workgroups are eight patches
// of 4 x 2 tiles, one patch per XCD. Measured against tile order (same box, `r02_blend_exp13-14.txt`): bench frame 0.117 ->
// 0.116 ms, from outside the cloud 0.668 -> 0.663 ms, the blend-bound frame 2.06 -> 1.95 ms. Larger patches trade the short
// frames for the long ones (4 x 4: 0.116 -> 0.121 ms and 2.03 -> 1.79 ms; 8 x 4: 0.117 -> 0.120 ms and 2.06 -> 1.73 ms).
constexpr int kPatchW = 4, kPatchH = 2, kPatchTiles = kPatchW * kPatchH;

// Longest tiles first. A frame's blend lasts as long as its slowest tiles, one wave each; which ones those are nobody knows
// in advance (not the length of the list: the pixels of an opaque centre saturate early) — except from the frame before: a
// camera moves little between two frames. Every tile's wave leaves how long it ran (`ticks`, quantised, library-owned memory
// that outlives the call); in the next call of the same size tile_order_kernel puts the workgroup numbers of the slow ones
// in front of the patch order above, longest first (`order`), beside the depth sort on a second stream, and the blend's
// workgroup i takes the tile of workgroup order[i]. The image does not depend on it (every tile is composited
// as before, only sooner or later); a first frame, a new size or stale ticks give some other valid order — all ticks equal:
// the patch order itself. Measured with the tiles ordered by their own frame's times (`profiles/r04_blend_tile_times.txt`):
// blend 0.470 -> 0.431 ms from (0,0,-14), 0.609 -> 0.532 from (0,0,-30), 0.212 -> 0.164 from (0,0,-9) on the sort plan,
// 3.24 -> 3.01 with faint splats; a random order costs 5-28 % (the neighbours' records in the XCD's L2).
struct TileOrder {
    const uint32_t* order = nullptr;   // [workgroups of one wave per tile] or null: patch order
    uint32_t* ticks = nullptr;         // [tiles of the frame] (units of 10 ns) or null: nothing recorded
};
// ~6 % steps (4 mantissa bits): tiles that took about as long keep their patch order among themselves
__device__ __forceinline__ uint32_t quantise_ticks(uint32_t t) {
    if (t < 16u) return t;
    const uint32_t e = 31u - (uint32_t)__clz((int)t);           // >= 4
    return min(511u, ((e - 3u) << 4) | ((t >> (e - 4u)) & 15u));
}
__device__ __forceinline__ uint32_t tile_clock() { return (uint32_t)wall_clock64(); }     // 100 MHz
// workgroups to launch for a band of `rows` tile rows (the patch grid is padded to whole groups of eight patches)
__host__ __device__ inline int patch_workgroups(int grid_x, int rows) {
    const int patches = ((grid_x + kPatchW - 1) / kPatchW) * ((rows + kPatchH - 1) / kPatchH);
    return (patches + 7) / 8 * 8 * kPatchTiles;
}
// tile (relative to the band) of workgroup b, or -1 if b falls into the padding
__device__ __forceinline__ int tile_of_workgroup(int b, int grid_x, int rows) {
    const int group = b / (8 * kPatchTiles), within = b % (8 * kPatchTiles);
    const int patch = group * 8 + within % 8, j = within / 8;
    const int ppr = (grid_x + kPatchW - 1) / kPatchW;
    const int tx = (patch % ppr) * kPatchW + j % kPatchW, ty = (patch / ppr) * kPatchH + j / kPatchW;
    return (tx < grid_x && ty < rows) ? ty * grid_x + tx : -1;
}

}  // namespace gsr
