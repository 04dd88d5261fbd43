// gscuda_shim.hpp — header-only C++ face of libgsrast_amd.so with the reference's exact
// signatures, so a caller written against apps/gsrast/gscuda/GSCuda.cuh and AuxBuffer.cuh
// compiles unchanged at the rasterizer symbol:
//
//   gscuda::forward(...)                        GSCuda.cuh:103-126 / GSCuda.cu:695-811
//   gscuda::required<T>(n)                      AuxBuffer.cuh:8-14
//   gscuda::gs::GeometryState::fromChunk etc.   AuxBuffer.cuh:38-76 / AuxBuffer.cu:44-89
//
// The three by-value std::function allocators become C callbacks through a trampoline; the
// void return + sticky error convention of the reference (its caller polls the runtime's last
// error after a device sync, apps/gsrast/CudaBuffer.hpp:8-12) is kept: forward() returns
// nothing and gscuda::lastError() reports the code of the last call.
//
// The state structs' pointer element types are glm's where glm is installed — the reference declares
// glm::vec2* / vec4* / vec3* / uvec2* (AuxBuffer.cuh:46-49,57), so a caller that assigns geomState.means2D to a
// glm::vec2* compiles unchanged — and layout-compatible PODs of the same names where it is not (this image has no glm).
// Define GSCUDA_SHIM_NO_GLM to get the PODs regardless.
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <utility>

#include "gsrast_amd.h"

#if !defined(GSCUDA_SHIM_NO_GLM) && defined(__has_include)
#if __has_include(<glm/glm.hpp>)
#include <glm/glm.hpp>
#define GSCUDA_SHIM_HAS_GLM 1
#endif
#endif

namespace gscuda {

#ifdef GSCUDA_SHIM_HAS_GLM
using vec2 = glm::vec2;
using vec3 = glm::vec3;
using vec4 = glm::vec4;
using uvec2 = glm::uvec2;
#else
struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct uvec2 { uint32_t x, y; };
#endif
// the strides the library writes (AuxBuffer.cu:52-59: vec2 8, vec3 12, vec4 16 bytes; a glm built with forced
// alignment of vec3 / vec4 would not fit)
static_assert(sizeof(vec2) == 8 && sizeof(vec3) == 12 && sizeof(vec4) == 16 && sizeof(uvec2) == 8,
              "gscuda_shim: vector types must be tightly packed");

namespace detail {
inline char* trampoline(void* user, size_t bytes) {
    return (*static_cast<std::function<char*(size_t)>*>(user))(bytes);
}
inline int& last_error_slot() {
    static thread_local int e = GSR_OK;
    return e;
}
inline gsr_forward_receipt& last_receipt_slot() {
    static thread_local gsr_forward_receipt r{};
    return r;
}
}  // namespace detail

inline int lastError() { return detail::last_error_slot(); }
// The receipt of the calling thread's last forward() (include/gsrast_amd.h): what gsr_backward and the poll below take.
inline const gsr_forward_receipt& lastReceipt() { return detail::last_receipt_slot(); }
// After the caller's device sync (the reference's CHECK_CUDA_ERROR, CudaBuffer.hpp:8-12): did a device-side wait of that
// forward() give up. GSR_OK also when the last forward() failed on the host (lastError() says how) or drew nothing.
inline int pollAsyncError() {
    const gsr_forward_receipt& r = detail::last_receipt_slot();
    return r.magic == GSR_RECEIPT_MAGIC ? gsr_poll_async_error(&r) : GSR_OK;
}

template <typename T>
size_t required(int num) {
    char* fake = nullptr;
    T::fromChunk(fake, num);
    return reinterpret_cast<size_t>(fake);
}

namespace gs {

struct GeometryState {
    uint32_t* tilesTouched;
    size_t scanSize;
    uint32_t numRendered;
    char* scanningSpace;
    float* depths;
    bool* clamped;
    int* internalRadii;
    vec2* means2D;
    float* cov3D;
    vec4* conicOpacity;
    vec3* rgb;
    uint32_t* pointOffsets;

    static GeometryState fromChunk(char*& chunk, int numGaussians) {
        gsr_geometry_state c;
        chunk = gsr_geometry_from_chunk(chunk, numGaussians, &c);
        GeometryState s;
        s.tilesTouched = c.tiles_touched;
        s.scanSize = c.scan_size;
        s.numRendered = c.num_rendered;
        s.scanningSpace = c.scanning_space;
        s.depths = c.depths;
        s.clamped = reinterpret_cast<bool*>(c.clamped);
        s.internalRadii = c.internal_radii;
        s.means2D = reinterpret_cast<vec2*>(c.means2D);
        s.cov3D = c.cov3D;
        s.conicOpacity = reinterpret_cast<vec4*>(c.conic_opacity);
        s.rgb = reinterpret_cast<vec3*>(c.rgb);
        s.pointOffsets = c.point_offsets;
        return s;
    }
};

struct ImageState {
    uvec2* ranges;
    uint32_t* nContrib;
    float* accumAlpha;

    static ImageState fromChunk(char*& chunk, int size) {
        gsr_image_state c;
        chunk = gsr_image_from_chunk(chunk, size, &c);
        ImageState s;
        s.ranges = reinterpret_cast<uvec2*>(c.ranges);
        s.nContrib = c.n_contrib;
        s.accumAlpha = c.accum_alpha;
        return s;
    }
};

struct BinningState {
    uint64_t* pointListKeysUnsorted;
    uint64_t* pointListKeys;
    uint32_t* pointListUnsorted;
    uint32_t* pointList;
    size_t sortingSize;
    char* listSortingSpace;

    static BinningState fromChunk(char*& chunk, int size) {
        gsr_binning_state c;
        chunk = gsr_binning_from_chunk(chunk, static_cast<size_t>(size < 0 ? 0 : size), &c);
        BinningState s;
        s.pointListKeysUnsorted = c.keys_unsorted;
        s.pointListKeys = c.keys;
        s.pointListUnsorted = c.values_unsorted;
        s.pointList = c.values;
        s.sortingSize = c.sorting_size;
        s.listSortingSpace = c.sorting_space;
        return s;
    }
};

}  // namespace gs

namespace detail {
// The reference's parameter list (GSCuda.cuh:103-126 = :19-42) filed into the C struct, field for field.
inline gsr_forward_args marshal(std::function<char*(size_t)>& geometryBuffer, std::function<char*(size_t)>& binningBuffer,
                                std::function<char*(size_t)>& imageBuffer, int numGaussians, int shDims, int M,
                                const float* background, int width, int height, const float* means3D, const float* shs,
                                const float* colorsPrecomp, const float* opacities, const float* scales, float scaleModifier,
                                const float* rotations, const float* cov3DPrecomp, const float* viewMatrix,
                                const float* projMatrix, const float* camPos, float tanFOVx, float tanFOVy, bool prefiltered,
                                float* outColor, int* radii, int* rects, float* boxMin, float* boxMax) {
    gsr_forward_args a{};
    a.struct_size = sizeof(a);
    a.geometry_alloc = trampoline; a.geometry_user = &geometryBuffer;
    a.binning_alloc = trampoline;  a.binning_user = &binningBuffer;
    a.image_alloc = trampoline;    a.image_user = &imageBuffer;
    a.num_gaussians = numGaussians; a.sh_dims = shDims; a.M = M;
    a.background = background;
    a.width = width; a.height = height;
    a.means3D = means3D; a.shs = shs; a.colors_precomp = colorsPrecomp;
    a.opacities = opacities; a.scales = scales; a.scale_modifier = scaleModifier;
    a.rotations = rotations; a.cov3D_precomp = cov3DPrecomp;
    a.view_matrix = viewMatrix; a.proj_matrix = projMatrix; a.cam_pos = camPos;
    a.tan_fovx = tanFOVx; a.tan_fovy = tanFOVy;
    a.prefiltered = prefiltered ? 1 : 0;
    a.out_color = outColor;
    a.radii = radii; a.rects = rects;
    a.box_min = boxMin; a.box_max = boxMax;
    a.stream = nullptr;                       // the reference runs on the default stream
    return a;
}
}  // namespace detail

// Same parameter list, order and meaning as the reference declaration.
inline void forward(std::function<char*(size_t)> geometryBuffer,
                    std::function<char*(size_t)> binningBuffer,
                    std::function<char*(size_t)> imageBuffer,
                    int numGaussians, int shDims, int M,
                    const float* background,
                    int width, int height,
                    const float* means3D,
                    const float* shs,
                    const float* colorsPrecomp,
                    const float* opacities,
                    const float* scales,
                    float scaleModifier,
                    const float* rotations,
                    const float* cov3DPrecomp,
                    const float* viewMatrix,
                    const float* projMatrix,
                    const float* camPos,
                    float tanFOVx, float tanFOVy,
                    bool prefiltered,
                    float* outColor,
                    int* radii,
                    int* rects,
                    float* boxMin,
                    float* boxMax) {
    gsr_forward_args a = detail::marshal(geometryBuffer, binningBuffer, imageBuffer, numGaussians, shDims, M, background, width,
                                         height, means3D, shs, colorsPrecomp, opacities, scales, scaleModifier, rotations,
                                         cov3DPrecomp, viewMatrix, projMatrix, camPos, tanFOVx, tanFOVy, prefiltered, outColor,
                                         radii, rects, boxMin, boxMax);
    detail::last_error_slot() = gsr_forward(&a);
    detail::last_receipt_slot() = a.receipt;
}

// gscuda::forwardPoints (GSCuda.cuh:19-42): same parameter list as forward; means3D has a stride of
// three floats on this path (GSCuda.cu:65).
inline void forwardPoints(std::function<char*(size_t)> geometryBuffer,
                          std::function<char*(size_t)> binningBuffer,
                          std::function<char*(size_t)> imageBuffer,
                          int numGaussians, int shDims, int M,
                          const float* background,
                          int width, int height,
                          const float* means3D,
                          const float* shs,
                          const float* colorsPrecomp,
                          const float* opacities,
                          const float* scales,
                          float scaleModifier,
                          const float* rotations,
                          const float* cov3DPrecomp,
                          const float* viewMatrix,
                          const float* projMatrix,
                          const float* camPos,
                          float tanFOVx, float tanFOVy,
                          bool prefiltered,
                          float* outColor,
                          int* radii,
                          int* rects,
                          float* boxMin,
                          float* boxMax) {
    gsr_forward_args a = detail::marshal(geometryBuffer, binningBuffer, imageBuffer, numGaussians, shDims, M, background, width,
                                         height, means3D, shs, colorsPrecomp, opacities, scales, scaleModifier, rotations,
                                         cov3DPrecomp, viewMatrix, projMatrix, camPos, tanFOVx, tanFOVy, prefiltered, outColor,
                                         radii, rects, boxMin, boxMax);
    detail::last_error_slot() = gsr_forward_points(&a);
}

namespace pc {
struct GeometryState {        // empty in the reference (AuxBuffer.cuh:21-24)
    static GeometryState fromChunk(char*&, int) { return GeometryState{}; }
};
struct ImageState {
    float* depth;
    float* outColor;
    float* defaultDepth;
    static ImageState fromChunk(char*& chunk, int size) {
        gsr_points_image_state c;
        chunk = gsr_points_image_from_chunk(chunk, size, &c);
        return ImageState{c.depth, c.out_color, c.default_depth};
    }
};
}  // namespace pc

}  // namespace gscuda

// The alternate spelling the reference can be switched to at compile time
// (apps/gsrast/GSGaussians.cpp:18-23): same parameter list.
namespace CudaRasterizer {
struct Rasterizer {
    template <typename... Args>
    static void forward(Args&&... args) { gscuda::forward(std::forward<Args>(args)...); }
};
}  // namespace CudaRasterizer
