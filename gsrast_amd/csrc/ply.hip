// Scene loading: the .ply reader of the reference and its activations, as the next row after the
// hot path.
//
//   gsr_ply_parse_header — apps/gsrast/SplatData.cpp:114-145 (loadFromSplatsPly): three getline
//       calls, the vertex count is the THIRD token of the THIRD line, then lines are skipped up to
//       "end_header"; property names are never parsed. Host only.
//   gsr_ply_activate     — SplatData.cpp:28-66 (loadFromPly) + SplatData.hpp:17-25 (RichPoint):
//       62 little-endian floats per vertex (position 3, normal 3, SH 48, opacity 1, scale 3,
//       rotation 4) -> the SoA the rasterizer takes: position (x,y,z,1), scale exp(s) with w = e,
//       rotation normalised (real part first), opacity sigmoid, SH copied raw. Runs on the GPU over
//       the raw records already in HBM: one wave pulls 64 records (15.5 KB) through LDS with
//       coalesced 16-byte loads and writes every output array with coalesced stores.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "gsr_common.hpp"

namespace gsr {
namespace {

constexpr int kPlyFloats = 62;                 // sizeof(RichPoint) / 4
constexpr int kWaveFloats = kWave * kPlyFloats;  // 3968 floats = 992 float4 per wave

__global__ __launch_bounds__(256) void ply_activate_kernel(const float* __restrict__ raw, int n, float4* __restrict__ means3D,
                                                           float4* __restrict__ scales, float4* __restrict__ rotations,
                                                           float* __restrict__ opacities, float* __restrict__ shs, int sh_layout) {
    __shared__ float lds[4][kWaveFloats];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const long long first = ((long long)blockIdx.x * 4 + wave) * kWave;     // first record of this wave
    if (first >= n) return;
    const int count = (int)min((long long)kWave, (long long)n - first);
    const int nfloats = count * kPlyFloats;
    const float* src = raw + first * kPlyFloats;                             // 248-byte records: 8-byte aligned
    float* w = lds[wave];
    // A wave starts at a multiple of 64 records = 15872 bytes, so 16-byte vector loads are aligned
    // whenever the raw buffer is; the tail of a partial wave (an even number of floats) goes in pairs.
    const int bulk = nfloats & ~3;
    for (int f = 4 * lane; f < bulk; f += 4 * kWave) {
        const float4 v = *reinterpret_cast<const float4*>(src + f);
        w[f] = v.x; w[f + 1] = v.y; w[f + 2] = v.z; w[f + 3] = v.w;
    }
    if (lane == 0 && bulk < nfloats) {
        w[bulk] = src[bulk];
        w[bulk + 1] = src[bulk + 1];
    }
    // wave-private LDS region: no workgroup barrier needed
    if (lane < count) {
        const float* r = w + lane * kPlyFloats;
        const long long i = first + lane;
        means3D[i] = make_float4(r[0], r[1], r[2], 1.0f);
        scales[i] = make_float4(expf(r[55]), expf(r[56]), expf(r[57]), expf(1.0f));
        const float q0 = r[58], q1 = r[59], q2 = r[60], q3 = r[61];
        const float d = (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);           // glm::dot(vec4)
        const float inv = 1.0f / sqrtf(d);                                    // glm::inversesqrt
        rotations[i] = make_float4(q0 * inv, q1 * inv, q2 * inv, q3 * inv);
        opacities[i] = 1.0f / (1.0f + expf(-r[54]));                          // sigmoid, SplatData.cpp:8-11
    }
    // SH block of the wave's records: count * 48 contiguous output floats
    float* sh_out = shs + first * 48;
    for (int f = lane; f < count * 48; f += kWave) {
        const int s = f / 48, c = f - s * 48;
        int from = c;                                   // GSR_SH_LAYOUT_FILE: f_dc_0..2, f_rest_0..44 as they lie in the file
        if (sh_layout == GSR_SH_LAYOUT_COEFFICIENT_MAJOR && c >= 3) {
            const int k = c / 3, ch = c - 3 * k;        // coefficient 1..15, channel: f_rest is channel-major in the file
            from = 3 + ch * 15 + (k - 1);
        }
        sh_out[f] = w[s * kPlyFloats + 6 + from];
    }
}

}  // namespace
}  // namespace gsr

using namespace gsr;

extern "C" {

int gsr_ply_parse_header(const char* path, int* num_splats, long long* data_offset) {
    if (!path || !num_splats || !data_offset) return GSR_ERR_INVALID_ARG;
    FILE* f = fopen(path, "rb");
    if (!f) return GSR_ERR_INVALID_ARG;
    char line[4096];
    int count = 0;
    bool ok = true;
    for (int i = 0; i < 3 && ok; ++i) ok = fgets(line, sizeof(line), f) != nullptr;
    if (ok) {
        char a[256], b[256];
        if (sscanf(line, "%255s %255s %d", a, b, &count) != 3) count = 0;   // `ss >> dummy >> dummy >> numSplats`
        ok = false;
        while (fgets(line, sizeof(line), f)) {
            size_t len = strlen(line);
            while (len && (line[len - 1] == '\n')) line[--len] = 0;          // getline strips only '\n'
            if (strcmp(line, "end_header") == 0) { ok = true; break; }
        }
    }
    *num_splats = count;
    *data_offset = ok ? (long long)ftell(f) : -1;
    fclose(f);
    return ok ? GSR_OK : GSR_ERR_INVALID_ARG;
}

int gsr_ply_activate(const float* raw_device, int n, float* means3D, float* scales, float* rotations, float* opacities,
                     float* shs, void* stream) {
    return gsr_ply_activate_layout(raw_device, n, means3D, scales, rotations, opacities, shs, GSR_SH_LAYOUT_FILE, stream);
}

int gsr_ply_activate_layout(const float* raw_device, int n, float* means3D, float* scales, float* rotations, float* opacities,
                            float* shs, int sh_layout, void* stream) {
    if (n <= 0) return GSR_OK;
    if (!raw_device || !means3D || !scales || !rotations || !opacities || !shs) return GSR_ERR_INVALID_ARG;
    if (sh_layout != GSR_SH_LAYOUT_FILE && sh_layout != GSR_SH_LAYOUT_COEFFICIENT_MAJOR) return GSR_ERR_INVALID_ARG;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(ply_activate_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, raw_device, n,
                       reinterpret_cast<float4*>(means3D), reinterpret_cast<float4*>(scales),
                       reinterpret_cast<float4*>(rotations), opacities, shs, sh_layout);
    GSR_LAUNCH_CHECK("ply_activate_kernel");
    return GSR_OK;
}

}  // extern "C"
