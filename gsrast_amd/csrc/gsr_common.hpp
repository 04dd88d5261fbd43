// Shared declarations of the HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/gsrast_amd.h"

namespace gsr {

constexpr int kTile = 16;          // BLOCK_W / BLOCK_H, reference GSCuda.cu:20-21
constexpr int kWave = 64;          // gfx950 wavefront

// Records the failing HIP call for gsr_last_hip_error().
void set_hip_error(hipError_t e, const char* what);
// Sets the calling thread's sticky error (gsr_last_error) and returns `code`.
int record_error(int code);

#define GSR_HIP_TRY(expr)                                   \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) {                             \
            ::gsr::set_hip_error(e_, #expr);                \
            return GSR_ERR_HIP;                             \
        }                                                   \
    } while (0)

#define GSR_LAUNCH_CHECK(name)                              \
    do {                                                    \
        hipError_t e_ = hipGetLastError();                  \
        if (e_ != hipSuccess) {                             \
            ::gsr::set_hip_error(e_, name);                 \
            return GSR_ERR_HIP;                             \
        }                                                   \
    } while (0)

// What the scan wants to know of a preprocess wave's 64 Gaussians, left as ONE 16-byte record per wave so that its first
// launch reads 0.25 bytes per Gaussian instead of re-reading tilesTouched (50 M Gaussians: 80 -> 13 us): {sum of tilesTouched,
// Gaussians with a tile, the instances of those of big_from tiles and more, those with a tile whose depth key has another
// top byte than the main one}. Lanes past the last Gaussian have left the kernel: the last wave sums by readlane.
__device__ __forceinline__ void store_wave_sums(uint4* __restrict__ wave_sums, int idx, uint32_t tiles, bool other_top, uint32_t big_from) {
    const unsigned long long active = __ballot(true);
    const uint32_t z = (uint32_t)__popcll(__ballot(tiles != 0u)), o = (uint32_t)__popcll(__ballot(tiles != 0u && other_top));
    uint32_t s = tiles, b = tiles >= big_from ? tiles : 0u;
    if (active == ~0ull) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); b += __shfl_xor(b, off, 64); }
    } else {
        uint32_t ss = 0, bb = 0;
        for (unsigned long long m = active; m != 0ull; m &= m - 1ull) {
            const int l = __ffsll((long long)m) - 1;
            ss += (uint32_t)__builtin_amdgcn_readlane((int)s, l);
            bb += (uint32_t)__builtin_amdgcn_readlane((int)b, l);
        }
        s = ss; b = bb;
    }
    if ((threadIdx.x & 63) == 0) wave_sums[idx >> 6] = make_uint4(s, z, b, o);
}

struct FrameDims {
    int width, height;
    int grid_x, grid_y;            // tile grid of the whole image
    int row_begin, row_end;        // tile rows this call bins / sorts / blends
};

// ---- stage launchers (each asynchronous on `stream`) ----
int launch_preprocess(const gsr_forward_args& a, const gsr_geometry_state& g, int32_t* radii,
                      uint32_t* depth_keys, uint32_t* rect_packed, const FrameDims& d, hipStream_t stream,
                      uint4* wave_sums = nullptr, bool colors_elsewhere = false, uint32_t big_from = 0xFFFFFFFFu);   // wave_sums: store_wave_sums
// geomState.rgb for the Gaussians with a tile (zeros for the others), as preprocess_kernel writes it — for a second stream
int launch_colors_visible(int n, const uint32_t* tiles_touched, const float* shs, float* rgb, hipStream_t stream);

int launch_colors_from_dc(int n, const float* shs, float* colors, hipStream_t stream);

int launch_preprocess_inria(const gsr_forward_args& a, const gsr_geometry_state& g, int32_t* radii, uint32_t* depth_keys,
                            uint32_t* rect_packed, const FrameDims& d, hipStream_t stream, uint4* wave_sums = nullptr,
                            uint32_t big_from = 0xFFFFFFFFu);

// nonzero (u32 per 4096 elements) / nonzero_total: optional — exclusive prefix of the per-tile counts of non-zero
// elements and their total (the depth order's compaction offsets, radix_sort.hip)
// host_words (mapped host memory, optional; needs total64): [0] = the non-zero total, [2..3] = the 64-bit total, written by the
// scan itself. clear / clear_bytes (optional, 16-byte granules): device memory the first launch also zeroes.
// wave_sums / main_count / side_max / side_words (all or none; need nonzero and big): the preprocess's per-wave records
// (store_wave_sums) — the first launch then reads those instead of `in` — and the depth order's side list, see scan.hip.
int launch_inclusive_scan(const uint32_t* in, uint32_t* out, size_t n, char* temp, hipStream_t stream,
                          unsigned long long* total64 = nullptr, uint32_t* nonzero = nullptr, uint32_t* nonzero_total = nullptr,
                          uint32_t* host_words = nullptr, void* clear = nullptr, size_t clear_bytes = 0,
                          const uint4* wave_sums = nullptr, uint32_t* main_count = nullptr,
                          uint32_t side_max = 0, uint32_t* side_words = nullptr,
                          uint32_t* big = nullptr, uint32_t big_from = 0);     // big (u32 per 4096 elements, needs host_words): host_words[10..11] = the 64-bit sum of the elements >= big_from
size_t scan_temp_bytes(size_t n);

int launch_gather_counts(int n, const uint32_t* sorted_depth, const uint32_t* sorted_idx, const uint32_t* tiles_touched,
                         uint32_t* counts, hipStream_t stream);
int launch_duplicate(int n, const uint32_t* sorted_depth, const uint32_t* sorted_idx, const uint32_t* emit_end,
                     const gsr_geometry_state& g, const int32_t* radii, const int32_t* rects, const FrameDims& d,
                     uint64_t* keys, uint32_t* values, uint32_t* hist_x, uint32_t* hist_y, hipStream_t stream);

int launch_sort_pairs(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* values_in,
                      uint32_t* values_out, size_t n, int begin_bit, int end_bit, char* temp, hipStream_t stream, uint32_t* error_word = nullptr,
                      uint32_t error_value = 1u);
size_t sort_temp_bytes(size_t n);
struct SweepScratch;
// The depth order's side list (radix_sort.hip, depth_side_kernel): the visible keys whose top byte is not main_top, when
// the scan found few of them. words: [0] the side way is taken (decided by the scan), [1] keys on the list, [2] those of
// them below main_top. main_partial: the compaction's offsets counted for the main keys only.
constexpr uint32_t kDepthSideMax = 1024;
constexpr uint32_t kDepthMainTop = 0x3Fu;            // float bits of [0.5, 1)
struct DepthSide {
    uint32_t* words = nullptr;
    const uint32_t* main_partial = nullptr;
    uint32_t *keys = nullptr, *vals = nullptr, *rects = nullptr;
    uint32_t capacity = 0, main_top = kDepthMainTop;
};
int launch_depth_side(const DepthSide& side, uint32_t m, uint32_t m_lo, uint32_t main_count, uint32_t* out_k, uint32_t* out_v,
                      uint32_t* out_r, hipStream_t stream);
// Depth order (radix_sort.hip). sc4: one scratch area per pass, already zeroed by the caller (look-back words,
// tickets, error word, histograms); the error word and the digit histograms live in sc4[0].
size_t depth_compact_scratch_bytes(size_t n);
// offsets_ready: `partial` / info[1] already hold the compaction offsets per 4096 keys and the visible count (the scan of
// tilesTouched produced them on the way: a key is the sentinel exactly where tilesTouched is 0)
// host_top (optional, mapped host memory): gets info[0] too, for a host that reads it after an event.
// rect_by_index / out_r (both or neither): the visible Gaussians' packed rectangles, compacted with the pairs; passed on as
// second_in / a_s / b_s they travel through the passes with the indices and arrive in depth order (gathering them by
// index afterwards is a random 4-byte read per Gaussian).
int sort_u32_prepare(const uint32_t* keys_in, uint32_t n, uint32_t* out_k, uint32_t* out_v, uint32_t* partial,
                     const SweepScratch* sc4, uint32_t* info, hipStream_t stream, bool offsets_ready = false,
                     const uint32_t* rect_by_index = nullptr, uint32_t* out_r = nullptr, uint32_t* host_top = nullptr,
                     const DepthSide* side = nullptr);
int sort_u32_passes(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t n, uint32_t* a_k, uint32_t* a_v, uint32_t* b_k,
                    uint32_t* b_v, const SweepScratch* sc4, int first, int last, hipStream_t stream,
                    const uint32_t* n_dev = nullptr, const uint32_t* second_in = nullptr, uint32_t* a_s = nullptr,
                    uint32_t* b_s = nullptr, const DepthSide* drop_side = nullptr, uint32_t* rec_a = nullptr,
                    uint32_t* rec_b = nullptr);
// Scenes beyond 16 M Gaussians: no compaction — the digit counts only (and info[0], info[4], host_top as sort_u32_prepare);
// the first pass then reads the per-Gaussian keys itself, dropping the sentinels (sort_u32_passes, drop_side).
int sort_u32_prepare_counts(const uint32_t* keys_in, uint32_t n, const SweepScratch* sc4, uint32_t* info, hipStream_t stream,
                            uint32_t* host_top, const DepthSide* side);

// Column-major emission (emit.hip): count, column scan and emission. The two events (may be null)
// are recorded between the N-sized preparation and the emission kernel, for stage timing.
size_t emit_scratch_bytes(size_t n);
int launch_emit_columns(int n, const uint32_t* sorted_depth, const uint32_t* sorted_idx, const uint32_t* sorted_rect,
                        int grid_x, int grid_y, char* scratch, uint32_t* hist_y, uint64_t* keys,
                        uint32_t* values, hipStream_t stream, hipEvent_t mark_prep_end, hipEvent_t mark_emit_begin);

// Block binning (blockbin.hip): the sorted lists written directly by tile-block owners.
bool blockbin_supported(int grid_x, int grid_y);
size_t blockbin_geo_bytes(size_t n);
size_t blockbin_bin_bytes(size_t r);
int launch_block_binning(int n, const uint32_t* sorted_depth, const uint32_t* sorted_idx, const uint32_t* sorted_rect,
                         int grid_x, int grid_y, uint32_t r_total, char* geo_scratch, uint64_t* ent_rd,
                         uint32_t* ent_idx, char* bin_scratch, uint32_t* ranges, bool close_single, hipStream_t stream,
                         hipEvent_t ev_coarse_end, uint32_t* nonempty_tiles = nullptr, uint32_t* skipped_stamp = nullptr,
                         int cus = 256);
int launch_block_emit(int n, int grid_x, int grid_y, uint32_t r_total, char* geo_scratch, const uint64_t* ent_rd,
                      const uint32_t* ent_idx, char* bin_scratch, uint64_t* keys, uint32_t* values, hipStream_t stream,
                      bool beside_blend = false,           // (the blend runs on another stream meanwhile: leave it room)
                      int cus = 256);
int launch_blend_blocks(int n, const FrameDims& d, uint32_t r_total, char* geo_scratch, const uint32_t* ent_idx, char* bin_scratch,
                        const uint32_t* ranges, const float* means2D, const float* colors, const float* conic_opacity,
                        float* final_t, uint32_t* n_contrib, const float* background, float* out_color,
                        unsigned long long* staged_counter, float t_cutoff, hipStream_t stream,
                        const uint32_t* tile_order = nullptr, uint32_t* tile_ticks = nullptr, bool colors_are_shs = false);

// nonempty (may be null): device word, zero before the launch; receives the number of tiles that got a list
int launch_tile_ranges(const uint64_t* keys, size_t n, uint32_t* ranges, int num_tiles, bool close_single, hipStream_t stream,
                       uint32_t* nonempty = nullptr);

int launch_blend(const FrameDims& d, const uint32_t* ranges, const uint32_t* point_list,
                 const float* means2D, const float* colors, const float* conic_opacity,
                 float* final_t, uint32_t* n_contrib, const float* background, float* out_color,
                 unsigned long long* staged_counter, float t_cutoff, hipStream_t stream,
                 const uint32_t* nonempty_tiles = nullptr, uint32_t num_rendered = 0,       // (both: see blend.hip, four waves per tile)
                 const uint32_t* tile_order = nullptr, uint32_t* tile_ticks = nullptr,       // (longest tiles first: TileOrder, blend_core.hpp)
                 bool colors_are_shs = false,                                                // (`colors` = the SH array: TileFeed::dc_stride)
                 const uint32_t* deep_count = nullptr,                                       // (device word: the order's leading entries that get four waves, blend.hip)
                 bool deep_all = false,                                                      // (every tile gets four waves)
                 int deep_waves = 4);                                                        // (... or 8 or 16: frames whose work sits in few tiles)
// Longest tiles first: the order of this call's blend workgroups from the ticks the tiles of the call before left.
constexpr int kTileOrderMax = 32768;      // workgroups (one per tile, patch grid padded) up to which the order is kept: 128 KB of LDS for its sort
// What the launch heuristics need to know about the chip, derived from its CU count (hipDeviceAttributeMultiprocessorCount,
// read once per device: an MI355X in a partitioned mode shows fewer CUs per device, and every figure below follows).
struct DeviceShape {
    int cus;                                 // compute units of the device
    uint32_t blend_slots;                    // wave slots of the blend kernels: 4 SIMDs x 5 waves (96 VGPRs) per CU — 5 120 on 256 CUs
    uint32_t blend_slots_beside;             // ... beside the emission's persistent workgroups, which keep their registers: 3 per SIMD — 3 072
    unsigned long long light_frame_ticks;    // 250 us (in 10 ns) per blend wave slot, summed over the tiles: below it a frame counts as LIGHT
    uint32_t persistent_workgroups(uint32_t per_cu) const { return (uint32_t)cus * per_cu; }
};
DeviceShape device_shape_of(int cus);                      // (pure: include/gsrast_amd.h gsr_device_shape, tests/test_capi_cpu.py)
int current_device_shape(DeviceShape* out);                // the current device's, cached per device (api.hip)
int tile_order_workgroups(const FrameDims& d);
// ticks / ticks_before: the tile times of the history's last frame and of the one before it; *sorted = false (and nothing
// launched): this device has no room for the sort's LDS
// deep_count (device word, optional): receives how many leading entries of the order are DEEP tiles (blend.hip);
// wave_slots: the chip's wave slots for the blend kernels (device_shape)
int launch_tile_order(const FrameDims& d, const uint32_t* ticks, const uint32_t* ticks_before, uint32_t* order, uint32_t* stats,
                      hipStream_t stream, bool* sorted, uint32_t* deep_count, const DeviceShape& shape);

int launch_exp_test(int n, const float* in, float* out, hipStream_t stream);
int launch_footprint_test(int n, const float* xy, const float* conic_opacity, const int32_t* tile_xy, int width, int height,
                          uint8_t* misses, hipStream_t stream);

}  // namespace gsr
