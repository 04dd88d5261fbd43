// gsr_oracle.cpp — CPU restatement of the reference forward splat rasterizer.
//
// TEST INFRASTRUCTURE ONLY. This file is the parity checker for the HIP path and
// the timed, non-target CPU baseline. Nothing under gsrast_amd/ may link, import
// or call it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
//
// PARITY UNPINNED BY THE REFERENCE: 42yeah/GSRast ships no tests, golden vectors or
// fixtures for this path and its CUDA/glm/CUB sources cannot be
// built in this image (no nvcc, no glm). The oracle is therefore pinned only by
// (i) an independent numpy restatement (oracle/oracle_np.py), (ii) closed-form
// known-answer tests (tests/test_oracle_kat.py) and (iii) fixtures minted from (i).
//
// Every function cites the reference lines it follows, relative to
// /root/reference/apps/gsrast/gscuda/.  Arithmetic is scalar float32 in the
// reference's operation order; build with -ffp-contract=off so no FMA is formed.
// glm operation orders are restated from glm's generic (non-SIMD) templates:
//   mat4*vec4  : (m[0]*v.x + m[1]*v.y) + (m[2]*v.z + m[3]*v.w)
//   mat3*mat3  : R[c][r] = a[0][r]*b[c][0] + a[1][r]*b[c][1] + a[2][r]*b[c][2]
//   dot(vec4)  : (x*x + y*y) + (z*z + w*w);  normalize = v * (1/sqrt(dot))
//   min(a,b)   : (b < a) ? b : a ;  max(a,b) : (a < b) ? b : a
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <thread>
#include <vector>

namespace {

constexpr int kTile = 16;      // BLOCK_W / BLOCK_H, GSCuda.cu:20-21
constexpr int kBatch = 256;    // blockSize, GSCuda.cu:568

struct Mat3 { float m[3][3]; };   // m[col][row], glm layout

inline float gmin(float a, float b) { return (b < a) ? b : a; }
inline float gmax(float a, float b) { return (a < b) ? b : a; }
inline int imin(int a, int b) { return (b < a) ? b : a; }
inline int imax(int a, int b) { return (a < b) ? b : a; }

// float -> int as the device does it (round toward zero, saturating, NaN -> 0).
inline int f2i(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

inline Mat3 mul(const Mat3& a, const Mat3& b) {
    Mat3 r;
    for (int c = 0; c < 3; ++c)
        for (int row = 0; row < 3; ++row)
            r.m[c][row] = a.m[0][row] * b.m[c][0] + a.m[1][row] * b.m[c][1] + a.m[2][row] * b.m[c][2];
    return r;
}
inline Mat3 transpose(const Mat3& a) {
    Mat3 r;
    for (int c = 0; c < 3; ++c)
        for (int row = 0; row < 3; ++row) r.m[c][row] = a.m[row][c];
    return r;
}
// glm mat4 (column-major float[16]) times vec4
inline void mat4_mul_vec4(const float* m, const float v[4], float out[4]) {
    for (int r = 0; r < 4; ++r)
        out[r] = (m[0 + r] * v[0] + m[4 + r] * v[1]) + (m[8 + r] * v[2] + m[12 + r] * v[3]);
}

// GSCuda.cu:157-162 quatToMat. q = (x,y,z,w) as stored = (real, i, j, k).
// The 2.0 / 1.0 literals are double, so each entry is evaluated in double from a
// float sum and narrowed by glm::mat3's converting constructor.
inline Mat3 quat_to_mat(const float q[4]) {
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    Mat3 r;
    r.m[0][0] = (float)(2.0 * (double)(x * x + y * y) - 1.0);
    r.m[0][1] = (float)(2.0 * (double)(y * z + x * w));
    r.m[0][2] = (float)(2.0 * (double)(y * w - x * z));
    r.m[1][0] = (float)(2.0 * (double)(y * z - x * w));
    r.m[1][1] = (float)(2.0 * (double)(x * x + z * z) - 1.0);
    r.m[1][2] = (float)(2.0 * (double)(z * w + x * y));
    r.m[2][0] = (float)(2.0 * (double)(y * w + x * z));
    r.m[2][1] = (float)(2.0 * (double)(z * w - x * y));
    r.m[2][2] = (float)(2.0 * (double)(x * x + w * w) - 1.0);
    return r;
}

// GSCuda.cu:168-195 computeCov3D
inline void compute_cov3d(const float scale[3], float scale_modifier, const float rot[4], float* cov3d) {
    Mat3 s;
    std::memset(&s, 0, sizeof(s));
    for (int i = 0; i < 3; ++i) s.m[i][i] = scale_modifier * scale[i];
    const float d = (rot[0] * rot[0] + rot[1] * rot[1]) + (rot[2] * rot[2] + rot[3] * rot[3]);
    const float inv = 1.0f / std::sqrt(d);
    const float q[4] = {rot[0] * inv, rot[1] * inv, rot[2] * inv, rot[3] * inv};
    const Mat3 rm = quat_to_mat(q);
    const Mat3 rs = mul(rm, s);
    const Mat3 sigma = mul(rs, transpose(rs));
    cov3d[0] = sigma.m[0][0];
    cov3d[1] = sigma.m[1][0];
    cov3d[2] = sigma.m[2][0];
    cov3d[3] = sigma.m[1][1];
    cov3d[4] = sigma.m[2][1];
    cov3d[5] = sigma.m[2][2];
}

// GSCuda.cu:197-231 computeCov2D; returns (cov[0][0], cov[0][1], cov[1][1])
inline void compute_cov2d(const float mean[3], float focal, float tan_fovx, float tan_fovy,
                          const float* cov3d, const float* view, float out[3]) {
    const float mv[4] = {mean[0], mean[1], mean[2], 1.0f};
    float t[4];
    mat4_mul_vec4(view, mv, t);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = t[0] / t[2];
    const float tytz = t[1] / t[2];
    t[0] = gmin(limx, gmax(-limx, txtz)) * t[2];
    t[1] = gmin(limy, gmax(-limy, tytz)) * t[2];
    Mat3 j;
    j.m[0][0] = focal / t[2]; j.m[0][1] = 0.0f; j.m[0][2] = (-focal * t[0]) / (t[2] * t[2]);
    j.m[1][0] = 0.0f; j.m[1][1] = focal / t[2]; j.m[1][2] = (-focal * t[1]) / (t[2] * t[2]);
    j.m[2][0] = 0.0f; j.m[2][1] = 0.0f; j.m[2][2] = 0.0f;
    Mat3 w;  // mat3(transpose(view)): w[c][r] = view[r][c]
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) w.m[c][r] = view[4 * r + c];
    const Mat3 tm = mul(w, j);
    Mat3 vrk;
    vrk.m[0][0] = cov3d[0]; vrk.m[0][1] = cov3d[1]; vrk.m[0][2] = cov3d[2];
    vrk.m[1][0] = cov3d[1]; vrk.m[1][1] = cov3d[3]; vrk.m[1][2] = cov3d[4];
    vrk.m[2][0] = cov3d[2]; vrk.m[2][1] = cov3d[4]; vrk.m[2][2] = cov3d[5];
    Mat3 cov = mul(mul(transpose(tm), vrk), tm);
    cov.m[0][0] += 0.3f;
    cov.m[1][1] += 0.3f;
    out[0] = cov.m[0][0];
    out[1] = cov.m[0][1];
    out[2] = cov.m[1][1];
}

// GSCuda.cu:237-259 getRect (both overloads: ext is the radius twice, or the rect)
inline void get_rect(float px, float py, int ext_x, int ext_y, int grid_x, int grid_y,
                     uint32_t rmin[2], uint32_t rmax[2]) {
    rmin[0] = (uint32_t)imin(grid_x, imax(0, f2i((px - (float)ext_x) / (float)kTile)));
    rmin[1] = (uint32_t)imin(grid_y, imax(0, f2i((py - (float)ext_y) / (float)kTile)));
    rmax[0] = (uint32_t)imin(grid_x, imax(0, f2i((((px + (float)ext_x) + (float)kTile) - 1.0f) / (float)kTile)));
    rmax[1] = (uint32_t)imin(grid_y, imax(0, f2i((((py + (float)ext_y) + (float)kTile) - 1.0f) / (float)kTile)));
}

inline uint32_t f2bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

}  // namespace

// f(begin, end) over [0, n) on `threads` host threads (contiguous ranges; threads <= 1: one call). Only used where the
// iterations are independent, so the result does not depend on the thread count.
template <typename F>
static void parallel_ranges(uint64_t n, int threads, F f) {
    if (threads <= 1 || n < 4096) { f((uint64_t)0, n); return; }
    const uint64_t parts = std::min<uint64_t>((uint64_t)threads, n);
    std::vector<std::thread> pool;
    for (uint64_t p = 0; p < parts; ++p) pool.emplace_back([&, p] { f(n * p / parts, n * (p + 1) / parts); });
    for (auto& t : pool) t.join();
}

// Stable sort of `order` by `less` on `threads` host threads: the chunks are sorted independently, then merged pairwise
// (std::merge keeps the elements of the left range in front of equal ones of the right range: the result is the one
// std::stable_sort gives, whatever the thread count). Used by the timed CPU baseline; threads <= 1 is std::stable_sort.
template <typename Less>
static void parallel_stable_sort(std::vector<uint64_t>& order, int threads, Less less) {
    const size_t n = order.size();
    if (threads <= 1 || n < (size_t)1 << 16) {
        std::stable_sort(order.begin(), order.end(), less);
        return;
    }
    size_t parts = 1;
    while (parts * 2 <= (size_t)threads && parts < 1024) parts *= 2;
    std::vector<size_t> cut(parts + 1);
    for (size_t p = 0; p <= parts; ++p) cut[p] = n * p / parts;
    {
        std::vector<std::thread> pool;
        for (size_t p = 0; p < parts; ++p)
            pool.emplace_back([&, p] { std::stable_sort(order.begin() + cut[p], order.begin() + cut[p + 1], less); });
        for (auto& t : pool) t.join();
    }
    std::vector<uint64_t> tmp(n);
    std::vector<uint64_t>* src = &order;
    std::vector<uint64_t>* dst = &tmp;
    for (size_t width = 1; width < parts; width *= 2) {
        std::vector<std::thread> pool;
        for (size_t p = 0; p < parts; p += 2 * width)
            pool.emplace_back([&, p] {
                const size_t a = cut[p], m = cut[std::min(p + width, parts)], b = cut[std::min(p + 2 * width, parts)];
                std::merge(src->begin() + a, src->begin() + m, src->begin() + m, src->begin() + b, dst->begin() + a, less);
            });
        for (auto& t : pool) t.join();
        std::swap(src, dst);
    }
    if (src != &order) order.swap(tmp);
}

extern "C" {

// GSCuda.cu:481-502 getHigherMsb
uint32_t gsro_higher_msb(uint32_t n) {
    int msb = (int)sizeof(uint32_t) * 4;
    int step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return (uint32_t)msb;
}

// GSCuda.cu:261-375 preprocessCUDA + :771 inclusive scan. Arrays not written by the
// reference for a culled Gaussian are left untouched here too. `rects` may be null
// (radius-based rect) exactly as in the reference; `radii` is the caller's array.
// Returns numRendered (GSCuda.cu:772).
uint64_t gsro_preprocess_mt(int n, const float* means3d /*vec4*/, const float* scales /*vec4*/,
                            float scale_modifier, const float* rotations /*vec4*/,
                            const float* opacities, const float* shs /*48 per splat*/,
                            const float* cov3d_precomp, const float* colors_precomp,
                            const float* view, const float* proj, int width, int height,
                            float tan_fovx, float tan_fovy,
                            int* radii, float* means2d, float* depths, float* cov3ds, float* rgb,
                            float* conic_opacity, uint32_t* tiles_touched, int* rects,
                            uint32_t* point_offsets, int threads);

uint64_t gsro_preprocess(int n, const float* means3d /*vec4*/, const float* scales /*vec4*/,
                         float scale_modifier, const float* rotations /*vec4*/,
                         const float* opacities, const float* shs /*48 per splat*/,
                         const float* cov3d_precomp, const float* colors_precomp,
                         const float* view, const float* proj, int width, int height,
                         float tan_fovx, float tan_fovy,
                         int* radii, float* means2d, float* depths, float* cov3ds, float* rgb,
                         float* conic_opacity, uint32_t* tiles_touched, int* rects,
                         uint32_t* point_offsets) {
    return gsro_preprocess_mt(n, means3d, scales, scale_modifier, rotations, opacities, shs, cov3d_precomp, colors_precomp, view,
                              proj, width, height, tan_fovx, tan_fovy, radii, means2d, depths, cov3ds, rgb, conic_opacity,
                              tiles_touched, rects, point_offsets, 1);
}

// (threads > 1: the Gaussians are independent — one thread per grid thread in the reference — so ranges of them run on
// host threads; the scan stays one loop. Same outputs whatever the thread count.)
uint64_t gsro_preprocess_mt(int n, const float* means3d /*vec4*/, const float* scales /*vec4*/,
                            float scale_modifier, const float* rotations /*vec4*/,
                            const float* opacities, const float* shs /*48 per splat*/,
                            const float* cov3d_precomp, const float* colors_precomp,
                            const float* view, const float* proj, int width, int height,
                            float tan_fovx, float tan_fovy,
                            int* radii, float* means2d, float* depths, float* cov3ds, float* rgb,
                            float* conic_opacity, uint32_t* tiles_touched, int* rects,
                            uint32_t* point_offsets, int threads) {
    const float focal = (float)height / (2.0f * tan_fovy);          // GSCuda.cu:721
    const int grid_x = (width + kTile - 1) / kTile, grid_y = (height + kTile - 1) / kTile;
    parallel_ranges((uint64_t)(n < 0 ? 0 : n), threads, [&](uint64_t first, uint64_t last) {
    for (int idx = (int)first; idx < (int)last; ++idx) {
        radii[idx] = 0;
        tiles_touched[idx] = 0;
        float ph[4];
        mat4_mul_vec4(proj, &means3d[4 * idx], ph);
        const float one_over_w = 1.0f / (0.001f + ph[3]);
        const float pr[3] = {one_over_w * ph[0], one_over_w * ph[1], one_over_w * ph[2]};
        if (pr[2] < 0.0f || pr[2] > 1.0f || pr[0] < -1.3f || pr[0] > 1.3f || pr[1] < -1.3f || pr[1] > 1.3f)
            continue;
        const float* cov3d;
        if (cov3d_precomp) {
            cov3d = &cov3d_precomp[6 * idx];
        } else {
            compute_cov3d(&scales[4 * idx], scale_modifier, &rotations[4 * idx], &cov3ds[6 * idx]);
            cov3d = &cov3ds[6 * idx];
        }
        float cov[3];
        compute_cov2d(&means3d[4 * idx], focal, tan_fovx, tan_fovy, cov3d, view, cov);
        const float det = cov[0] * cov[2] - cov[1] * cov[1];
        if (det == 0.0f) continue;
        const float det_inv = 1.0f / det;
        const float conic[3] = {cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv};
        const float mid = 0.5f * (cov[0] + cov[2]);
        const float lambda1 = mid + std::sqrt(gmax(0.1f, mid * mid - det));
        const float lambda2 = mid - std::sqrt(gmax(0.1f, mid * mid - det));
        const float my_radius = std::ceil(3.0f * std::sqrt(gmax(lambda1, lambda2)));
        const float pix = (pr[0] * 0.5f + 0.5f) * (float)width;
        const float piy = (pr[1] * 0.5f + 0.5f) * (float)height;
        uint32_t rmin[2], rmax[2];
        if (rects == nullptr) {
            const int r = f2i(my_radius);
            get_rect(pix, piy, r, r, grid_x, grid_y, rmin, rmax);
        } else {
            const int ex = f2i(std::ceil(3.0f * std::sqrt(cov[0])));
            const int ey = f2i(std::ceil(3.0f * cov[2]));          // sic: no sqrt, GSCuda.cu:352
            rects[2 * idx] = ex;
            rects[2 * idx + 1] = ey;
            get_rect(pix, piy, ex, ey, grid_x, grid_y, rmin, rmax);
        }
        const uint32_t area = (rmax[0] - rmin[0]) * (rmax[1] - rmin[1]);
        if (area == 0) continue;
        if (!colors_precomp) {
            for (int c = 0; c < 3; ++c) rgb[3 * idx + c] = 0.5f + 0.4f * shs[48 * (size_t)idx + c];
        }
        depths[idx] = pr[2];
        radii[idx] = f2i(my_radius);
        means2d[2 * idx] = pix;
        means2d[2 * idx + 1] = piy;
        conic_opacity[4 * idx + 0] = conic[0];
        conic_opacity[4 * idx + 1] = conic[1];
        conic_opacity[4 * idx + 2] = conic[2];
        conic_opacity[4 * idx + 3] = opacities[idx];
        tiles_touched[idx] = area;
    }
    });
    uint32_t run = 0;                                              // GSCuda.cu:771 (u32 wrap as CUB)
    for (int i = 0; i < n; ++i) { run += tiles_touched[i]; point_offsets[i] = run; }
    return n > 0 ? point_offsets[n - 1] : 0;
}

// GSCuda.cu:422-475 duplicateWithKeys, :794-797 stable radix sort on the low
// (32 + getHigherMsb(tiles)) bits, :504-538 identifyTileRanges (with the R==1 quirk).
// `ranges` holds 2 u32 per tile and must be zeroed by the caller (:800 memset).
void gsro_bin_mt(int n, int width, int height, const int* radii, const float* means2d,
                 const float* depths, const uint32_t* point_offsets, const int* rects,
                 uint64_t num_rendered, uint64_t* keys_unsorted, uint32_t* values_unsorted,
                 uint64_t* keys, uint32_t* values, uint32_t* ranges, int threads);

void gsro_bin(int n, int width, int height, const int* radii, const float* means2d,
              const float* depths, const uint32_t* point_offsets, const int* rects,
              uint64_t num_rendered, uint64_t* keys_unsorted, uint32_t* values_unsorted,
              uint64_t* keys, uint32_t* values, uint32_t* ranges) {
    gsro_bin_mt(n, width, height, radii, means2d, depths, point_offsets, rects, num_rendered, keys_unsorted, values_unsorted,
                keys, values, ranges, 1);
}

void gsro_bin_mt(int n, int width, int height, const int* radii, const float* means2d,
                 const float* depths, const uint32_t* point_offsets, const int* rects,
                 uint64_t num_rendered, uint64_t* keys_unsorted, uint32_t* values_unsorted,
                 uint64_t* keys, uint32_t* values, uint32_t* ranges, int threads) {
    const int grid_x = (width + kTile - 1) / kTile, grid_y = (height + kTile - 1) / kTile;
    // (every Gaussian writes its own range of the arrays, GSCuda.cu:447: ranges of Gaussians run on host threads)
    parallel_ranges((uint64_t)(n < 0 ? 0 : n), threads, [&](uint64_t first, uint64_t last) {
    for (int idx = (int)first; idx < (int)last; ++idx) {
        if (radii[idx] <= 0) continue;
        uint32_t off = (idx == 0) ? 0u : point_offsets[idx - 1];
        uint32_t rmin[2], rmax[2];
        if (rects == nullptr)
            get_rect(means2d[2 * idx], means2d[2 * idx + 1], radii[idx], radii[idx], grid_x, grid_y, rmin, rmax);
        else
            get_rect(means2d[2 * idx], means2d[2 * idx + 1], rects[2 * idx], rects[2 * idx + 1], grid_x, grid_y, rmin, rmax);
        for (int y = (int)rmin[1]; y < (int)rmax[1]; ++y)
            for (int x = (int)rmin[0]; x < (int)rmax[0]; ++x) {
                uint64_t key = (uint64_t)(uint32_t)(y * grid_x + x);
                key <<= 32;
                key |= f2bits(depths[idx]);
                keys_unsorted[off] = key;
                values_unsorted[off] = (uint32_t)idx;
                ++off;
            }
    }
    });
    const uint32_t bits = 32 + gsro_higher_msb((uint32_t)(grid_x * grid_y));
    const uint64_t mask = (bits >= 64) ? ~0ull : ((1ull << bits) - 1ull);
    std::vector<uint64_t> order(num_rendered);
    std::iota(order.begin(), order.end(), 0ull);
    parallel_stable_sort(order, threads, [&](uint64_t a, uint64_t b) {
        return (keys_unsorted[a] & mask) < (keys_unsorted[b] & mask);
    });
    parallel_ranges(num_rendered, threads, [&](uint64_t first, uint64_t last) {
        for (uint64_t i = first; i < last; ++i) {
            keys[i] = keys_unsorted[order[i]];
            values[i] = values_unsorted[order[i]];
        }
    });
    for (uint64_t idx = 0; idx < num_rendered; ++idx) {
        const uint32_t cur = (uint32_t)(keys[idx] >> 32);
        if (idx == 0) {
            ranges[2 * cur] = 0;
        } else {
            const uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
            if (prev != cur) {
                ranges[2 * prev + 1] = (uint32_t)idx;
                ranges[2 * cur] = (uint32_t)idx;
            }
            if (idx == num_rendered - 1) ranges[2 * cur + 1] = (uint32_t)num_rendered;
        }
    }
}

// GSCuda.cu:543-677 renderCUDA for tile rows [ty0, ty1). One call = the blocks of
// those tile rows; the block's 256-record rounds, its all-done break (:595-599) and
// the per-pixel loop (:623-665) are restated as they stand. Returns the number of
// records staged into shared memory (R_f of SURVEY.md §8d) for those rows.
static uint64_t blend_rows(int ty0, int ty1, int width, int height, const uint32_t* ranges,
                           const uint32_t* point_list, const float* means2d, const float* colors,
                           const float* conic_opacity, const float* background, float* final_t,
                           uint32_t* n_contrib, float* out_color) {
    const int grid_x = (width + kTile - 1) / kTile;
    const size_t plane = (size_t)width * (size_t)height;
    uint64_t staged = 0;
    float cx[kBatch], cy[kBatch], cc[kBatch][4], crgb[kBatch][3];
    for (int ty = ty0; ty < ty1; ++ty)
        for (int tx = 0; tx < grid_x; ++tx) {
            const uint32_t r0 = ranges[2 * (ty * grid_x + tx)], r1 = ranges[2 * (ty * grid_x + tx) + 1];
            const int pmin_x = tx * kTile, pmin_y = ty * kTile;
            const int pmax_x = imin(pmin_x + kTile, width), pmax_y = imin(pmin_y + kTile, height);
            const int rounds = (int)((r1 - r0 + kBatch - 1) / kBatch);   // as the reference: u32 math then int
            int work = (int)(r1 - r0);
            float acc_t[kBatch], col[kBatch][3];
            uint32_t contributor[kBatch], last[kBatch];
            bool inside[kBatch], done[kBatch];
            for (int t = 0; t < kBatch; ++t) {
                const int px = pmin_x + (t % kTile), py = pmin_y + (t / kTile);
                inside[t] = px < pmax_x && py < pmax_y;
                done[t] = !inside[t];
                acc_t[t] = 1.0f; contributor[t] = 0; last[t] = 0;
                col[t][0] = col[t][1] = col[t][2] = 0.0f;
            }
            for (int i = 0; i < rounds; ++i, work -= kBatch) {
                int num_done = 0;
                for (int t = 0; t < kBatch; ++t) num_done += done[t] ? 1 : 0;
                if (num_done == kBatch) break;
                const int cnt = imin(kBatch, work);
                for (int k = 0; k < cnt; ++k) {
                    const uint32_t id = point_list[r0 + (uint32_t)(i * kBatch + k)];
                    cx[k] = means2d[2 * (size_t)id]; cy[k] = means2d[2 * (size_t)id + 1];
                    for (int c = 0; c < 4; ++c) cc[k][c] = conic_opacity[4 * (size_t)id + c];
                    for (int c = 0; c < 3; ++c) crgb[k][c] = colors[3 * (size_t)id + c];
                }
                staged += (uint64_t)cnt;
                for (int t = 0; t < kBatch; ++t) {
                    if (done[t]) continue;
                    const float fx = (float)(uint32_t)(pmin_x + (t % kTile));
                    const float fy = (float)(uint32_t)(pmin_y + (t / kTile));
                    for (int j = 0; !done[t] && j < cnt; ++j) {
                        contributor[t]++;
                        const float dx = cx[j] - fx, dy = cy[j] - fy;
                        const float power = -0.5f * (cc[j][0] * dx * dx + cc[j][2] * dy * dy) - cc[j][1] * dx * dy;
                        if (power > 0.0f) continue;
                        const float alpha = gmin(0.99f, cc[j][3] * std::exp(power));
                        if (alpha < 1.0f / 255.0f) continue;
                        const float test = acc_t[t] * (1.0f - alpha);
                        if (test < 0.001f) { done[t] = true; continue; }
                        for (int c = 0; c < 3; ++c) col[t][c] += crgb[j][c] * alpha * acc_t[t];
                        acc_t[t] = test;
                        last[t] = contributor[t];
                    }
                }
            }
            for (int t = 0; t < kBatch; ++t) {
                if (!inside[t]) continue;
                const size_t pid = (size_t)(pmin_y + t / kTile) * (size_t)width + (size_t)(pmin_x + t % kTile);
                final_t[pid] = acc_t[t];
                n_contrib[pid] = last[t];
                for (int c = 0; c < 3; ++c) out_color[pid + plane * c] = col[t][c] + acc_t[t] * background[c];
            }
        }
    return staged;
}

// threads <= 1: scalar single-thread loop. threads > 1: std::thread pool pulling tile
// rows from an atomic counter (the CPU baseline of SURVEY.md §8d). Returns R_f.
uint64_t gsro_blend(int width, int height, const uint32_t* ranges, const uint32_t* point_list,
                    const float* means2d, const float* colors, const float* conic_opacity,
                    const float* background, float* final_t, uint32_t* n_contrib, float* out_color,
                    int threads) {
    const int grid_y = (height + kTile - 1) / kTile;
    if (threads <= 1)
        return blend_rows(0, grid_y, width, height, ranges, point_list, means2d, colors, conic_opacity,
                          background, final_t, n_contrib, out_color);
    std::atomic<int> next{0};
    std::atomic<uint64_t> staged{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&] {
            uint64_t mine = 0;
            for (;;) {
                const int ty = next.fetch_add(1);
                if (ty >= grid_y) break;
                mine += blend_rows(ty, ty + 1, width, height, ranges, point_list, means2d, colors,
                                   conic_opacity, background, final_t, n_contrib, out_color);
            }
            staged += mine;
        });
    for (auto& th : pool) th.join();
    return staged.load();
}

unsigned gsro_hardware_concurrency(void) { return std::thread::hardware_concurrency(); }

// out[i] = exp(in[i]) exactly as the tile loop above computes it (GSCuda.cu:645: the float overload, libm's expf): the checker
// of the HIP blend's exponential (tests/test_gpu_parity.py).
void gsro_expf(long n, const float* in, float* out) {
    for (long i = 0; i < n; ++i) out[i] = std::exp(in[i]);
}

}  // extern "C"
