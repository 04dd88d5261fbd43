// Preprocess kernel of the `inria` semantics profile: the per-Gaussian stage of
// the UPSTREAM rasterizer (graphdeco-inria/diff-gaussian-rasterization) instead of the reference's
// gscuda variant. The upstream sources are not in this tree (empty submodule in the reference), so
// this follows the published algorithm; parity is unpinned and checked only against this repo's
// numpy restatement (oracle/inria_np.py). Differences from preprocess.hip, by the rows of the
// divergence table in SURVEY.md §8a:
//   D1/D2 colour  : real spherical harmonics up to degree 3 on [N][16][3] coefficients, + 0.5, clamp at 0
//   D3   depth    : view-space z is the sort key
//   D4   cull     : view-space z <= 0.2 only
//   D5   w eps    : 1e-7
//   D6   pixel    : ((ndc + 1) * S - 1) * 0.5, evaluated in double as upstream's literals make it
//   D7   rect     : square of the 3-sigma radius
//   D9   focal    : separate focal_x / focal_y
//   cov3D         : quaternion not re-normalised, Sigma = (S R)^T (S R)
// Inputs keep the reference app's buffers (means3D / scales vec4-strided).
#include "gsr_common.hpp"

namespace gsr {
namespace {

struct M3 { float m[3][3]; };   // m[col][row]

__device__ __forceinline__ M3 mul3(const M3& a, const M3& b) {
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int row = 0; row < 3; ++row)
            r.m[c][row] = a.m[0][row] * b.m[c][0] + a.m[1][row] * b.m[c][1] + a.m[2][row] * b.m[c][2];
    return r;
}
__device__ __forceinline__ M3 transpose3(const M3& a) {
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int row = 0; row < 3; ++row) r.m[c][row] = a.m[row][c];
    return r;
}
__device__ __forceinline__ float fminr(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float fmaxr(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(hi, max(lo, v)); }

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
__constant__ float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                               0.5462742152960396f};
__constant__ float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

struct InriaParams {
    int n, deg;
    const float4* means3D;
    const float4* scales;
    float scale_modifier;
    const float4* rotations;
    const float* opacities;
    const float* shs;              // [N][16][3]
    const float* cov3D_precomp;
    const float* colors_precomp;
    const float* view;
    const float* proj;
    const float* cam_pos;
    float tan_fovx, tan_fovy, focal_x, focal_y;
    int32_t* radii;
    float2* means2D;
    float* depths;
    float* cov3Ds;
    float* rgb;
    uint8_t* clamped;
    float4* conic_opacity;
    uint32_t* tiles_touched;
    uint32_t* depth_keys;
    uint32_t* rect_packed;
    uint4* wave_sums;              // (as in preprocess.hip)
    uint32_t big_from;
    FrameDims dims;
};

// Memory schedule of a wave, as in preprocess.hip: means; scale / rotation of the lanes in front of the camera; then,
// if any lane of the wave has a tile, the wave's 64 x 48 SH floats in ONE coalesced sweep (12 KB: twelve 1-KB loads)
// through wave-private LDS, each lane reading its own record from there; then every store. The first version read
// a lane's 48 coefficients with 48 scalar loads at a 192-byte stride, three of them at a time behind a byte store
// (`clamped`) the compiler had to assume they alias: 6.8 ms for 50 M Gaussians at degree 3.
constexpr int kShFloats = 48;
constexpr int kShStride = 49;            // LDS floats per record: odd, so the 64 lanes' reads of coefficient k hit 32 banks twice

__global__ __launch_bounds__(256) void preprocess_inria_kernel(const InriaParams p) {
    __shared__ float s_sh[4][kWave * kShStride];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const bool valid = idx < p.n;
    int32_t out_radius = 0;
    uint32_t out_tiles = 0, out_rect = 0;
    float view_z = 0.0f;
    float view[16], proj[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { view[i] = p.view[i]; proj[i] = p.proj[i]; }
    const float cam0 = p.cam_pos ? p.cam_pos[0] : 0.0f, cam1 = p.cam_pos ? p.cam_pos[1] : 0.0f, cam2 = p.cam_pos ? p.cam_pos[2] : 0.0f;
    const float4 mean = valid ? p.means3D[idx] : make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    const float* v = view;
    const float pvx = v[0] * mean.x + v[4] * mean.y + v[8] * mean.z + v[12];
    const float pvy = v[1] * mean.x + v[5] * mean.y + v[9] * mean.z + v[13];
    const float pvz = v[2] * mean.x + v[6] * mean.y + v[10] * mean.z + v[14];
    const bool in_front = valid && !(pvz <= 0.2f);
    float c3[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    float pix = 0.0f, piy = 0.0f, ca = 0.0f, cb = 0.0f, cc = 0.0f, det_inv = 0.0f;
    int ri = 0, x0 = 0, x1 = 0, fy0 = 0, fy1 = 0;
    bool has_tile = false;
    if (in_front) {
        const float* m = proj;
        const float phx = m[0] * mean.x + m[4] * mean.y + m[8] * mean.z + m[12];
        const float phy = m[1] * mean.x + m[5] * mean.y + m[9] * mean.z + m[13];
        const float phw = m[3] * mean.x + m[7] * mean.y + m[11] * mean.z + m[15];
        const float pw = 1.0f / (phw + 0.0000001f);
        const float prx = phx * pw, pry = phy * pw;
        if (p.cov3D_precomp) {
#pragma unroll
            for (int i = 0; i < 6; ++i) c3[i] = p.cov3D_precomp[6 * (size_t)idx + i];
        } else {
            const float4 sc = p.scales[idx];
            const float4 q = p.rotations[idx];
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            M3 rm;   // written row-wise into a column-major constructor, as upstream does
            rm.m[0][0] = 1.0f - 2.0f * (y * y + z * z); rm.m[0][1] = 2.0f * (x * y - r * z); rm.m[0][2] = 2.0f * (x * z + r * y);
            rm.m[1][0] = 2.0f * (x * y + r * z); rm.m[1][1] = 1.0f - 2.0f * (x * x + z * z); rm.m[1][2] = 2.0f * (y * z - r * x);
            rm.m[2][0] = 2.0f * (x * z - r * y); rm.m[2][1] = 2.0f * (y * z + r * x); rm.m[2][2] = 1.0f - 2.0f * (x * x + y * y);
            M3 s;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) s.m[c][rr] = 0.0f;
            s.m[0][0] = p.scale_modifier * sc.x; s.m[1][1] = p.scale_modifier * sc.y; s.m[2][2] = p.scale_modifier * sc.z;
            const M3 mm = mul3(s, rm);
            const M3 sigma = mul3(transpose3(mm), mm);
            c3[0] = sigma.m[0][0]; c3[1] = sigma.m[0][1]; c3[2] = sigma.m[0][2];
            c3[3] = sigma.m[1][1]; c3[4] = sigma.m[1][2]; c3[5] = sigma.m[2][2];
        }
        const float limx = 1.3f * p.tan_fovx, limy = 1.3f * p.tan_fovy;
        const float tz = pvz;
        const float tx = fminr(limx, fmaxr(-limx, pvx / tz)) * tz;
        const float ty = fminr(limy, fmaxr(-limy, pvy / tz)) * tz;
        M3 j;
        j.m[0][0] = p.focal_x / tz; j.m[0][1] = 0.0f; j.m[0][2] = -(p.focal_x * tx) / (tz * tz);
        j.m[1][0] = 0.0f; j.m[1][1] = p.focal_y / tz; j.m[1][2] = -(p.focal_y * ty) / (tz * tz);
        j.m[2][0] = 0.0f; j.m[2][1] = 0.0f; j.m[2][2] = 0.0f;
        M3 wv;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) wv.m[c][rr] = v[4 * rr + c];
        const M3 tm = mul3(wv, j);
        M3 vrk;
        vrk.m[0][0] = c3[0]; vrk.m[0][1] = c3[1]; vrk.m[0][2] = c3[2];
        vrk.m[1][0] = c3[1]; vrk.m[1][1] = c3[3]; vrk.m[1][2] = c3[4];
        vrk.m[2][0] = c3[2]; vrk.m[2][1] = c3[4]; vrk.m[2][2] = c3[5];
        const M3 cv = mul3(mul3(transpose3(tm), transpose3(vrk)), tm);
        ca = cv.m[0][0] + 0.3f; cb = cv.m[0][1]; cc = cv.m[1][1] + 0.3f;
        const float det = ca * cc - cb * cb;
        if (det != 0.0f) {
            det_inv = 1.0f / det;
            const float mid = 0.5f * (ca + cc);
            const float root = sqrtf(fmaxr(0.1f, mid * mid - det));
            const float my_radius = ceilf(3.0f * sqrtf(fmaxr(mid + root, mid - root)));
            pix = (float)((((double)prx + 1.0) * (double)p.dims.width - 1.0) * 0.5);
            piy = (float)((((double)pry + 1.0) * (double)p.dims.height - 1.0) * 0.5);
            ri = (int)my_radius;
            const float rf = (float)ri;
            x0 = clampi((int)((pix - rf) / 16.0f), 0, p.dims.grid_x);
            fy0 = clampi((int)((piy - rf) / 16.0f), 0, p.dims.grid_y);
            x1 = clampi((int)((((pix + rf) + 16.0f) - 1.0f) / 16.0f), 0, p.dims.grid_x);
            fy1 = clampi((int)((((piy + rf) + 16.0f) - 1.0f) / 16.0f), 0, p.dims.grid_y);
            has_tile = (uint32_t)(x1 - x0) * (uint32_t)(fy1 - fy0) != 0;
        }
    }

    // ---- colour of the lanes that have a tile: the wave's SH records, through LDS ----
    const bool want_sh = !p.colors_precomp;
    float res[3] = {0.0f, 0.0f, 0.0f};
    bool neg[3] = {false, false, false};
    const float opacity = has_tile ? p.opacities[idx] : 0.0f;
    if (want_sh && __ballot(has_tile) != 0ull) {
        float* w = s_sh[wave];
        // the wave's records: floats [first, first + 64 * 48) of shs, 16 bytes per lane and load; a 16-byte piece never
        // straddles two records (48 is a multiple of 4)
        const size_t first = ((size_t)blockIdx.x * 256 + (size_t)wave * kWave) * kShFloats;
        const size_t limit = (size_t)p.n * kShFloats;
        const int pieces = p.deg > 2 ? 12 : (p.deg > 1 ? 7 : (p.deg > 0 ? 3 : 1));      // 16-byte pieces of a record that hold coefficients 0 .. (deg + 1)^2 - 1
        float4 piece[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int f4 = q * kWave + lane;                      // index of the 16-byte piece inside the wave's block
            const size_t at = first + 4 * (size_t)f4;
            const bool needed = (f4 % 12) < pieces;               // pieces of a record beyond the degree in use are not read
            piece[q] = (needed && at + 3 < limit) ? *reinterpret_cast<const float4*>(p.shs + at) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int f4 = q * kWave + lane;
            const int rec = f4 / 12, k = (f4 % 12) * 4;
            float* dst = w + rec * kShStride + k;
            dst[0] = piece[q].x; dst[1] = piece[q].y; dst[2] = piece[q].z; dst[3] = piece[q].w;
        }
        // wave-private LDS: the writes above and the reads below are ordered inside the wave
        if (has_tile) {
            const float* sh = w + lane * kShStride;
            // real SH basis, view direction from the camera centre to the Gaussian
            float dx = mean.x - cam0, dy = mean.y - cam1, dz = mean.z - cam2;
            const float len = sqrtf(dx * dx + dy * dy + dz * dz);
            dx = dx / len; dy = dy / len; dz = dz / len;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float r = SH_C0 * sh[c];
                if (p.deg > 0) {
                    const float x = dx, y = dy, z = dz;
                    r = r - SH_C1 * y * sh[3 + c] + SH_C1 * z * sh[6 + c] - SH_C1 * x * sh[9 + c];
                    if (p.deg > 1) {
                        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        r = r + SH_C2[0] * xy * sh[12 + c] + SH_C2[1] * yz * sh[15 + c] +
                            SH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + c] + SH_C2[3] * xz * sh[21 + c] +
                            SH_C2[4] * (xx - yy) * sh[24 + c];
                        if (p.deg > 2) {
                            r = r + SH_C3[0] * y * (3.0f * xx - yy) * sh[27 + c] + SH_C3[1] * xy * z * sh[30 + c] +
                                SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + c] +
                                SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + c] +
                                SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + c] + SH_C3[5] * z * (xx - yy) * sh[42 + c] +
                                SH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + c];
                        }
                    }
                }
                r += 0.5f;
                neg[c] = r < 0.0f;
                res[c] = fmaxr(r, 0.0f);
            }
        }
    }
    if (!valid) return;

    // ---- stores from here on ---- every Gaussian stores every record, zeros where the upstream kernel stores nothing, as
    // streaming stores: whole lines, no partial writes (preprocess.hip explains; here 0.42 -> 0.33 ms on the bench frame)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    if (!p.cov3D_precomp) {
        f32x2* dst = reinterpret_cast<f32x2*>(p.cov3Ds + 6 * (size_t)idx);
        __builtin_nontemporal_store((f32x2){in_front ? c3[0] : 0.0f, in_front ? c3[1] : 0.0f}, dst);
        __builtin_nontemporal_store((f32x2){in_front ? c3[2] : 0.0f, in_front ? c3[3] : 0.0f}, dst + 1);
        __builtin_nontemporal_store((f32x2){in_front ? c3[4] : 0.0f, in_front ? c3[5] : 0.0f}, dst + 2);
    }
    if (want_sh) {
#pragma unroll
        for (int c = 0; c < 3; ++c) p.clamped[3 * (size_t)idx + c] = has_tile && neg[c];
        float* o = p.rgb + 3 * (size_t)idx;
        __builtin_nontemporal_store(has_tile ? res[0] : 0.0f, o);
        __builtin_nontemporal_store(has_tile ? res[1] : 0.0f, o + 1);
        __builtin_nontemporal_store(has_tile ? res[2] : 0.0f, o + 2);
    }
    __builtin_nontemporal_store(has_tile ? pvz : 0.0f, p.depths + idx);
    __builtin_nontemporal_store((f32x2){has_tile ? pix : 0.0f, has_tile ? piy : 0.0f}, reinterpret_cast<f32x2*>(p.means2D + idx));
    __builtin_nontemporal_store((f32x4){has_tile ? cc * det_inv : 0.0f, has_tile ? -cb * det_inv : 0.0f, has_tile ? ca * det_inv : 0.0f,
                                        has_tile ? opacity : 0.0f},
                                reinterpret_cast<f32x4*>(p.conic_opacity + idx));
    if (has_tile) {
        const int y0 = clampi(fy0, p.dims.row_begin, p.dims.row_end), y1 = clampi(fy1, p.dims.row_begin, p.dims.row_end);
        out_radius = ri;
        out_tiles = (uint32_t)(x1 - x0) * (uint32_t)(y1 - y0);
        view_z = pvz;
        if (out_tiles)
            out_rect = (uint32_t)x0 | ((uint32_t)(x1 - x0) << 8) | ((uint32_t)y0 << 16) | ((uint32_t)(y1 - y0) << 24);
    }
    p.radii[idx] = out_radius;
    p.tiles_touched[idx] = out_tiles;
    const uint32_t dkey = out_tiles ? __float_as_uint(view_z) : 0xFFFFFFFFu;
    p.depth_keys[idx] = dkey;
    if (p.rect_packed) p.rect_packed[idx] = out_rect;
    if (p.wave_sums) store_wave_sums(p.wave_sums, idx, out_tiles, (dkey >> 24) != kDepthMainTop, p.big_from);
}

}  // namespace

int launch_preprocess_inria(const gsr_forward_args& a, const gsr_geometry_state& g, int32_t* radii, uint32_t* depth_keys,
                            uint32_t* rect_packed, const FrameDims& d, hipStream_t stream, uint4* wave_sums, uint32_t big_from) {
    InriaParams p;
    p.n = a.num_gaussians;
    p.deg = a.sh_dims < 0 ? 0 : (a.sh_dims > 3 ? 3 : a.sh_dims);
    p.means3D = reinterpret_cast<const float4*>(a.means3D);
    p.scales = reinterpret_cast<const float4*>(a.scales);
    p.scale_modifier = a.scale_modifier;
    p.rotations = reinterpret_cast<const float4*>(a.rotations);
    p.opacities = a.opacities;
    p.shs = a.shs;
    p.cov3D_precomp = a.cov3D_precomp;
    p.colors_precomp = a.colors_precomp;
    p.view = a.view_matrix;
    p.proj = a.proj_matrix;
    p.cam_pos = a.cam_pos;
    p.tan_fovx = a.tan_fovx;
    p.tan_fovy = a.tan_fovy;
    p.focal_x = (float)a.width / (2.0f * a.tan_fovx);
    p.focal_y = (float)a.height / (2.0f * a.tan_fovy);
    p.radii = radii;
    p.means2D = reinterpret_cast<float2*>(g.means2D);
    p.depths = g.depths;
    p.cov3Ds = g.cov3D;
    p.rgb = g.rgb;
    p.clamped = g.clamped;
    p.conic_opacity = reinterpret_cast<float4*>(g.conic_opacity);
    p.tiles_touched = g.tiles_touched;
    p.depth_keys = depth_keys;
    p.rect_packed = rect_packed;
    p.wave_sums = wave_sums;
    p.big_from = big_from;
    p.dims = d;
    hipLaunchKernelGGL(preprocess_inria_kernel, dim3((unsigned)((a.num_gaussians + 255) / 256)), dim3(256), 0, stream, p);
    GSR_LAUNCH_CHECK("preprocess_inria_kernel");
    return GSR_OK;
}

}  // namespace gsr
