// C-ABI entry points (include/gsrast_amd.h): chunk layout, gsr_forward orchestration,
// error reporting and the stage-level entry points the parity tests call.
//
// gsr_forward follows reference apps/gsrast/gscuda/GSCuda.cu:695-811 (gscuda::forward);
// the chunk carving follows AuxBuffer.cu:13-21 (obtain) and :44-89 (fromChunk).
#include <stdio.h>
#include <string.h>

#include "gsr_common.hpp"

namespace gsr {

static thread_local int g_last_error = GSR_OK;
static thread_local char g_hip_error[256] = "";

void set_hip_error(hipError_t e, const char* what) {
    snprintf(g_hip_error, sizeof(g_hip_error), "%s: %s", what, hipGetErrorString(e));
}

namespace {

// obtain(): AuxBuffer.cu:13-21 — align the running pointer up, hand out `bytes`.
template <typename T>
inline void obtain(char*& chunk, T*& out, size_t bytes, size_t align = 128) {
    const size_t offset = reinterpret_cast<size_t>(chunk);
    const size_t aligned = align * ((offset + align - 1) / align);
    out = reinterpret_cast<T*>(aligned);
    chunk = reinterpret_cast<char*>(aligned + bytes);
}

inline int fail(int code) { g_last_error = code; return code; }

// Pinned 4-byte landing zone for the numRendered read-back, one per host thread.
struct Readback {
    uint32_t* host = nullptr;
    unsigned long long* staged_dev = nullptr;
    unsigned long long* staged_host = nullptr;
    hipEvent_t ev[2 * GSR_NUM_STAGES] = {};   // [2s] start, [2s+1] end of stage s
    bool events = false;
    int ensure() {
        if (!host) {
            GSR_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&host), 64, hipHostMallocDefault));
            staged_host = reinterpret_cast<unsigned long long*>(host + 8);
        }
        return GSR_OK;
    }
    int ensure_staged() {
        if (!staged_dev) GSR_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&staged_dev), sizeof(unsigned long long)));
        return GSR_OK;
    }
    int ensure_events() {
        if (!events) {
            for (auto& e : ev) GSR_HIP_TRY(hipEventCreate(&e));
            events = true;
        }
        return GSR_OK;
    }
};
static thread_local Readback g_rb;

}  // namespace
}  // namespace gsr

using namespace gsr;

extern "C" {

char* gsr_geometry_from_chunk(char* chunk, int n, gsr_geometry_state* s) {
    const size_t N = (size_t)(n < 0 ? 0 : n);
    obtain(chunk, s->tiles_touched, sizeof(uint32_t) * N);
    s->scan_size = scan_temp_bytes(N);
    s->num_rendered = 0;
    obtain(chunk, s->scanning_space, s->scan_size);
    obtain(chunk, s->depths, sizeof(float) * N);
    obtain(chunk, s->clamped, sizeof(uint8_t) * N * 3);
    obtain(chunk, s->internal_radii, sizeof(int32_t) * N);
    obtain(chunk, s->means2D, sizeof(float) * 2 * N);
    obtain(chunk, s->cov3D, sizeof(float) * 6 * N);
    obtain(chunk, s->conic_opacity, sizeof(float) * 4 * N);
    obtain(chunk, s->rgb, sizeof(float) * 3 * N);
    obtain(chunk, s->point_offsets, sizeof(uint32_t) * N);
    return chunk;
}

char* gsr_image_from_chunk(char* chunk, int size, gsr_image_state* s) {
    const size_t P = (size_t)(size < 0 ? 0 : size);
    obtain(chunk, s->ranges, sizeof(uint32_t) * 2 * P);
    obtain(chunk, s->n_contrib, sizeof(uint32_t) * P);
    obtain(chunk, s->accum_alpha, sizeof(float) * P);
    return chunk;
}

char* gsr_binning_from_chunk(char* chunk, size_t size, gsr_binning_state* s) {
    obtain(chunk, s->keys_unsorted, sizeof(uint64_t) * size);
    obtain(chunk, s->keys, sizeof(uint64_t) * size);
    obtain(chunk, s->values_unsorted, sizeof(uint32_t) * size);
    obtain(chunk, s->values, sizeof(uint32_t) * size);
    s->sorting_size = sort_temp_bytes(size);
    obtain(chunk, s->sorting_space, s->sorting_size);
    return chunk;
}

size_t gsr_required_geometry(int n) { gsr_geometry_state s; return reinterpret_cast<size_t>(gsr_geometry_from_chunk(nullptr, n, &s)); }
size_t gsr_required_image(int size) { gsr_image_state s; return reinterpret_cast<size_t>(gsr_image_from_chunk(nullptr, size, &s)); }
size_t gsr_required_binning(size_t size) { gsr_binning_state s; return reinterpret_cast<size_t>(gsr_binning_from_chunk(nullptr, size, &s)); }

uint32_t gsr_higher_msb(uint32_t n) {   // GSCuda.cu:481-502
    int msb = (int)sizeof(uint32_t) * 4;
    int step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return (uint32_t)msb;
}

int gsr_last_error(void) { return g_last_error; }
const char* gsr_last_hip_error(void) { return g_hip_error; }
const char* gsr_error_string(int code) {
    switch (code) {
        case GSR_OK: return "ok";
        case GSR_ERR_INVALID_ARG: return "invalid argument";
        case GSR_ERR_ALLOC: return "chunk allocator returned NULL";
        case GSR_ERR_HIP: return "HIP runtime error";
        case GSR_ERR_NO_DEVICE: return "no HIP device";
        case GSR_ERR_TOO_LARGE: return "numRendered exceeds 32-bit offsets";
        default: return "unknown error";
    }
}

size_t gsr_scan_temp_bytes(size_t n) { return scan_temp_bytes(n); }
int gsr_inclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, char* temp, void* stream) {
    g_hip_error[0] = 0;
    if (n && (!in || !out || !temp)) return fail(GSR_ERR_INVALID_ARG);
    return fail(launch_inclusive_scan(in, out, n, temp, (hipStream_t)stream));
}
size_t gsr_sort_temp_bytes(size_t n) { return sort_temp_bytes(n); }
int gsr_sort_pairs_u64_u32(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* values_in,
                           uint32_t* values_out, size_t n, int end_bit, char* temp, void* stream) {
    g_hip_error[0] = 0;
    if (n && (!keys_in || !keys_out || !values_in || !values_out || !temp)) return fail(GSR_ERR_INVALID_ARG);
    return fail(launch_sort_pairs(keys_in, keys_out, values_in, values_out, n, end_bit, temp, (hipStream_t)stream));
}

int gsr_forward(gsr_forward_args* a) {
    g_hip_error[0] = 0;
    if (!a || a->struct_size != sizeof(gsr_forward_args)) return fail(GSR_ERR_INVALID_ARG);
    a->num_rendered = 0;
    a->records_staged = 0;
    memset(a->stage_ms, 0, sizeof(a->stage_ms));
    const int n = a->num_gaussians;
    if (n <= 0 || a->width <= 0 || a->height <= 0 || !a->geometry_alloc || !a->binning_alloc || !a->image_alloc ||
        !a->background || !a->means3D || !a->opacities || !a->view_matrix || !a->proj_matrix || !a->out_color ||
        (!a->shs && !a->colors_precomp) || (!a->cov3D_precomp && (!a->scales || !a->rotations)))
        return fail(GSR_ERR_INVALID_ARG);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(GSR_ERR_NO_DEVICE);

    hipStream_t stream = (hipStream_t)a->stream;
    const bool profile = (a->flags & GSR_FLAG_PROFILE) != 0;
    const bool count_staged = (a->flags & GSR_FLAG_COUNT_STAGED) != 0;
    int rc;
    if ((rc = g_rb.ensure()) != GSR_OK) return fail(rc);
    if (profile && (rc = g_rb.ensure_events()) != GSR_OK) return fail(rc);
    if (count_staged && (rc = g_rb.ensure_staged()) != GSR_OK) return fail(rc);

    FrameDims d;
    d.width = a->width;
    d.height = a->height;
    d.grid_x = (a->width + kTile - 1) / kTile;
    d.grid_y = (a->height + kTile - 1) / kTile;
    d.row_begin = 0;
    d.row_end = d.grid_y;
    if (a->tile_row_begin != 0 || a->tile_row_end != 0) {
        if (a->tile_row_begin < 0 || a->tile_row_end > d.grid_y || a->tile_row_begin > a->tile_row_end)
            return fail(GSR_ERR_INVALID_ARG);
        d.row_begin = a->tile_row_begin;
        d.row_end = a->tile_row_end;
    }
    const int num_tiles = d.grid_x * d.grid_y;

    // GSCuda.cu:723-729
    char* geo_chunk = a->geometry_alloc(a->geometry_user, gsr_required_geometry(n));
    if (!geo_chunk) return fail(GSR_ERR_ALLOC);
    gsr_geometry_state geom;
    gsr_geometry_from_chunk(geo_chunk, n, &geom);
    int32_t* radii = a->radii ? a->radii : geom.internal_radii;

    // GSCuda.cu:734-736
    const int P = a->width * a->height;
    char* img_chunk = a->image_alloc(a->image_user, gsr_required_image(P) + 128);
    if (!img_chunk) return fail(GSR_ERR_ALLOC);
    gsr_image_state img;
    gsr_image_from_chunk(img_chunk, P, &img);

#define GSR_BEGIN(s) do { if (profile) GSR_HIP_TRY(hipEventRecord(g_rb.ev[2 * (s)], stream)); } while (0)
#define GSR_END(s) do { if (profile) GSR_HIP_TRY(hipEventRecord(g_rb.ev[2 * (s) + 1], stream)); } while (0)
#define GSR_STEP(call) do { rc = (call); if (rc != GSR_OK) return fail(rc); } while (0)

    GSR_BEGIN(GSR_STAGE_PREPROCESS);
    GSR_STEP(launch_preprocess(*a, geom, radii, d, stream));                               // :744-768
    GSR_END(GSR_STAGE_PREPROCESS);
    GSR_BEGIN(GSR_STAGE_SCAN);
    GSR_STEP(launch_inclusive_scan(geom.tiles_touched, geom.point_offsets, (size_t)n,      // :771
                                   geom.scanning_space, stream));
    GSR_END(GSR_STAGE_SCAN);
    // :772 — the pipeline's one device->host sync: the binning chunk is sized by R.
    GSR_HIP_TRY(hipMemcpyAsync(g_rb.host, geom.point_offsets + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    GSR_HIP_TRY(hipStreamSynchronize(stream));
    const uint32_t R = *g_rb.host;
    a->num_rendered = R;
    if (R == 0) return fail(GSR_OK);                                                        // :775-778

    char* bin_chunk = a->binning_alloc(a->binning_user, gsr_required_binning(R) + 128);    // :782-784
    if (!bin_chunk) return fail(GSR_ERR_ALLOC);
    gsr_binning_state bin;
    gsr_binning_from_chunk(bin_chunk, R, &bin);

    GSR_BEGIN(GSR_STAGE_DUPLICATE);
    GSR_STEP(launch_duplicate(n, geom, radii, a->rects, d, bin.keys_unsorted, bin.values_unsorted, stream));   // :787
    GSR_END(GSR_STAGE_DUPLICATE);
    const int end_bit = 32 + (int)gsr_higher_msb((uint32_t)num_tiles);                     // :791
    GSR_BEGIN(GSR_STAGE_SORT);
    GSR_STEP(launch_sort_pairs(bin.keys_unsorted, bin.keys, bin.values_unsorted, bin.values, R, end_bit,
                               bin.sorting_space, stream));                                // :794-797
    GSR_END(GSR_STAGE_SORT);
    GSR_BEGIN(GSR_STAGE_RANGES);
    GSR_STEP(launch_tile_ranges(bin.keys, R, img.ranges, num_tiles, stream));              // :800-801
    GSR_END(GSR_STAGE_RANGES);
    if (count_staged) GSR_HIP_TRY(hipMemsetAsync(g_rb.staged_dev, 0, sizeof(unsigned long long), stream));
    const float* colors = a->colors_precomp ? a->colors_precomp : geom.rgb;                // :803
    GSR_BEGIN(GSR_STAGE_BLEND);
    GSR_STEP(launch_blend(d, img.ranges, bin.values, geom.means2D, colors, geom.conic_opacity, img.accum_alpha,
                          img.n_contrib, a->background, a->out_color, count_staged ? g_rb.staged_dev : nullptr,
                          stream));                                                        // :804-810
    GSR_END(GSR_STAGE_BLEND);

    if (profile || count_staged) {
        if (count_staged)
            GSR_HIP_TRY(hipMemcpyAsync(g_rb.staged_host, g_rb.staged_dev, sizeof(unsigned long long),
                                       hipMemcpyDeviceToHost, stream));
        GSR_HIP_TRY(hipStreamSynchronize(stream));
        if (count_staged) a->records_staged = *g_rb.staged_host;
        if (profile) {
            for (int s = 0; s < GSR_NUM_STAGES; ++s) {
                float ms = 0.0f;
                GSR_HIP_TRY(hipEventElapsedTime(&ms, g_rb.ev[2 * s], g_rb.ev[2 * s + 1]));
                a->stage_ms[s] = ms;
            }
        }
    }
#undef GSR_BEGIN
#undef GSR_END
#undef GSR_STEP
    return fail(GSR_OK);
}

}  // extern "C"
