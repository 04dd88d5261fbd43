// Point-splat path: every Gaussian centre projected to one pixel, nearest point wins.
//
// Follows the semantics of reference apps/gsrast/gscuda/GSCuda.cu:26-155 (clearColor,
// projectPoints, forwardPoints; the reference never calls it) with the pc:: chunk layout of
// AuxBuffer.cu:23-40. The reference resolves the depth test with an atomicMin on the depth and a
// colour write that follows it unordered, so with several points on one pixel its result depends
// on timing. Here the test is one 64-bit atomicMin on (depth bits << 32 | index) and a second
// kernel writes the winner's colour: always the nearest point, the lowest index among equals —
// one of the outcomes the reference can produce, the same one every run.
#include "gsr_common.hpp"

namespace gsr {
namespace {

constexpr unsigned long long kEmpty = 0xFFFFFFFFFFFFFFFFull;

__global__ __launch_bounds__(256) void points_clear_kernel(unsigned long long* __restrict__ winner, int pixels) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < pixels) winner[i] = kEmpty;
}

// means3D is read with a stride of THREE floats here (GSCuda.cu:65), unlike gsr_forward's vec4.
__global__ __launch_bounds__(256) void points_project_kernel(int n, const float* __restrict__ means3D,
                                                             const float* __restrict__ proj, int width, int height,
                                                             unsigned long long* __restrict__ winner) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const float x = means3D[3 * (size_t)idx], y = means3D[3 * (size_t)idx + 1], z = means3D[3 * (size_t)idx + 2];
    const float hx = (proj[0] * x + proj[4] * y) + (proj[8] * z + proj[12] * 1.0f);
    const float hy = (proj[1] * x + proj[5] * y) + (proj[9] * z + proj[13] * 1.0f);
    const float hz = (proj[2] * x + proj[6] * y) + (proj[10] * z + proj[14] * 1.0f);
    const float hw = (proj[3] * x + proj[7] * y) + (proj[11] * z + proj[15] * 1.0f);
    const float one_over_w = 1.0f / (hw + 0.001f);
    const float px = hx * one_over_w, py = hy * one_over_w, pz = hz * one_over_w;
    if (pz < 0.0f || pz > 1.0f || px < -1.0f || px > 1.0f || py < -1.0f || py > 1.0f) return;
    // (0.5f + 0.5 * projected.x) * width: the 0.5 literal makes this double arithmetic, narrowed to float (:78)
    const float ix = (float)((0.5 + 0.5 * (double)px) * (double)width);
    const float iy = (float)((0.5 + 0.5 * (double)py) * (double)height);
    const int cx = (int)roundf(ix), cy = (int)roundf(iy);
    if (cx < 0 || cx >= width || cy < 0 || cy >= height) return;
    const unsigned long long key = ((unsigned long long)__float_as_uint(pz) << 32) | (unsigned long long)(uint32_t)idx;
    atomicMin(winner + (size_t)cy * width + cx, key);
}

__global__ __launch_bounds__(256) void points_resolve_kernel(const unsigned long long* __restrict__ winner,
                                                             const float* __restrict__ shs, const float* __restrict__ background,
                                                             int pixels, float* __restrict__ depth, float* __restrict__ out_color) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= pixels) return;
    const unsigned long long key = winner[i];
    float r = background[0], g = background[1], b = background[2], d = 1.0f;
    if (key != kEmpty) {
        const size_t idx = (size_t)(key & 0xFFFFFFFFull);
        const float* sh = shs + 48 * idx;                       // 3 channels x 16 coefficients, DC first (:96-97)
        r = 0.4f * sh[0] + 0.5f; g = 0.4f * sh[1] + 0.5f; b = 0.4f * sh[2] + 0.5f;
        d = __uint_as_float((uint32_t)(key >> 32));
    }
    depth[i] = d;
    out_color[i] = r;
    out_color[i + (size_t)pixels] = g;
    out_color[i + 2 * (size_t)pixels] = b;
}

template <typename T>
inline void obtain(char*& chunk, T*& out, size_t bytes, size_t align = 128) {
    const size_t offset = reinterpret_cast<size_t>(chunk);
    const size_t aligned = align * ((offset + align - 1) / align);
    out = reinterpret_cast<T*>(aligned);
    chunk = reinterpret_cast<char*>(aligned + bytes);
}

}  // namespace
}  // namespace gsr

using namespace gsr;

extern "C" {

// pc::ImageState::fromChunk (AuxBuffer.cu:31-39) + this library's winner table behind it
char* gsr_points_image_from_chunk(char* chunk, int size, gsr_points_image_state* s) {
    const size_t P = (size_t)(size < 0 ? 0 : size);
    obtain(chunk, s->depth, sizeof(float) * P);
    obtain(chunk, s->out_color, sizeof(float) * P * 3);
    obtain(chunk, s->default_depth, sizeof(float));
    obtain(chunk, s->winner, sizeof(uint64_t) * P);
    return chunk;
}
size_t gsr_required_points_image(int size) {
    gsr_points_image_state s;
    return reinterpret_cast<size_t>(gsr_points_image_from_chunk(nullptr, size, &s));
}

static int forward_points_impl(gsr_forward_args* a) {
    if (!a || a->struct_size != sizeof(gsr_forward_args)) return GSR_ERR_INVALID_ARG;
    a->num_rendered = 0;
    const int n = a->num_gaussians;
    if (n <= 0 || a->width <= 0 || a->height <= 0 || !a->geometry_alloc || !a->image_alloc || !a->background || !a->means3D ||
        !a->shs || !a->proj_matrix || !a->out_color)
        return GSR_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)a->stream;
    // GSCuda.cu:128-134: the (empty) geometry state and the image state, each with 32 bytes of slack
    if (!a->geometry_alloc(a->geometry_user, 0 + 32)) return GSR_ERR_ALLOC;
    const int P = a->width * a->height;
    char* chunk = a->image_alloc(a->image_user, gsr_required_points_image(P) + 32);
    if (!chunk) return GSR_ERR_ALLOC;
    gsr_points_image_state im;
    gsr_points_image_from_chunk(chunk, P, &im);
    const float farthest = 1.0f;                                              // :140-141
    GSR_HIP_TRY(hipMemcpyAsync(im.default_depth, &farthest, sizeof(float), hipMemcpyHostToDevice, stream));
    unsigned long long* winner = reinterpret_cast<unsigned long long*>(im.winner);
    const unsigned pb = (unsigned)((P + 255) / 256);
    hipLaunchKernelGGL(points_clear_kernel, dim3(pb), dim3(256), 0, stream, winner, P);
    GSR_LAUNCH_CHECK("points_clear_kernel");
    hipLaunchKernelGGL(points_project_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, a->means3D,
                       a->proj_matrix, a->width, a->height, winner);
    GSR_LAUNCH_CHECK("points_project_kernel");
    hipLaunchKernelGGL(points_resolve_kernel, dim3(pb), dim3(256), 0, stream, winner, a->shs, a->background, P, im.depth,
                       im.out_color);
    GSR_LAUNCH_CHECK("points_resolve_kernel");
    // :154 — the temporary image is copied to the caller's outColor in one go
    GSR_HIP_TRY(hipMemcpyAsync(a->out_color, im.out_color, sizeof(float) * 3 * (size_t)P, hipMemcpyDeviceToDevice, stream));
    return GSR_OK;
}

int gsr_forward_points(gsr_forward_args* a) { return record_error(forward_points_impl(a)); }

}  // extern "C"
