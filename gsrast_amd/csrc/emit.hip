// Column-major key emission (tile grids up to 255 x 255): the first of the two tile passes of the
// sort — a stable pass on the tile column x — is not run as a sort, its result is written directly.
//
// Reference semantics: apps/gsrast/gscuda/GSCuda.cu:422-475 (duplicateWithKeys) emits, per
// Gaussian, one (tile << 32 | depth bits, index) pair per covered tile; :794-797 then sorts them.
// Here Gaussians arrive in depth order (sorted once per Gaussian, before duplication). In that
// order the keys of column x are, Gaussian after Gaussian, the h rows of its rectangle, so key
// (g, x, y) belongs at
//     start[x] + (rows of all earlier Gaussians covering x) + (y - y0).
//   column_count_kernel : per chunk of 256 depth-consecutive Gaussians, keys per column
//                         (difference arrays, 4 LDS atomics per Gaussian) -> table[chunk][x]
//   colscan_*           : exclusive prefix of the table down the chunks for every column, plus the
//                         column starts: table[chunk][x] becomes the output index of the chunk's
//                         first key in column x (three small launches, all row-coalesced)
//   emit_chunk_kernel   : resolves the order INSIDE a chunk with per-column bit masks of covering
//                         Gaussians (+ per-32-Gaussian row sums), then writes every
//                         (Gaussian, column) run as contiguous 16-byte key / 8-byte value stores.
// What reaches the remaining pass (stable on the tile row y) is exactly what a stable x pass over
// the depth-ordered list would have produced, so the sorted list is bit-identical.
#include <stdlib.h>

#include "gsr_common.hpp"

namespace gsr {
namespace {

constexpr int kChunk = 256;        // Gaussians per workgroup
constexpr int kRowsPerBlock = 256; // table rows per workgroup of the column scan
constexpr int kSmallRect = 8;   // rectangles up to this many tiles are written by their own lane

// Two consecutive rows of one column run in one go: a 16-byte key store and an 8-byte value store.
// The destination is only 8-byte (keys) / 4-byte (values) aligned; gfx950 under HSA runs with
// unaligned vector-memory access enabled, so dword-aligned wide stores are legal.
struct __attribute__((packed, aligned(8))) KeyPair { uint64_t a, b; };
struct __attribute__((packed, aligned(4))) ValPair { uint32_t a, b; };

__device__ __forceinline__ void emit1(uint64_t* __restrict__ keys, uint32_t* __restrict__ values, uint32_t pos,
                                      uint32_t tile, uint32_t depth_bits, uint32_t idx) {
    keys[pos] = ((uint64_t)tile << 32) | (uint64_t)depth_bits;
    values[pos] = idx;
}
__device__ __forceinline__ void emit2(uint64_t* __restrict__ keys, uint32_t* __restrict__ values, uint32_t pos,
                                      uint32_t tile, uint32_t tile_step, uint32_t depth_bits, uint32_t idx) {
    KeyPair k;
    k.a = ((uint64_t)tile << 32) | (uint64_t)depth_bits;
    k.b = ((uint64_t)(tile + tile_step) << 32) | (uint64_t)depth_bits;
    *reinterpret_cast<KeyPair*>(keys + pos) = k;
    ValPair v;
    v.a = idx;
    v.b = idx;
    *reinterpret_cast<ValPair*>(values + pos) = v;
}

// rect packed as x0 | w << 8 | y0 << 16 | h << 24 (all < 256); 0 = culled / empty
// rect_packed: the visible Gaussians' rectangles in depth order (they travel through the depth sort with the indices;
// gathering them by index here cost 0.93 ms on 50 M Gaussians)
__global__ __launch_bounds__(kChunk) void column_count_kernel(int n, const uint32_t* __restrict__ rect_packed, int stride_x,
                                                              int stride_y, uint32_t* __restrict__ table) {
    // Difference arrays: a w x h rectangle adds h at column x0 and takes it back at x0 + w (rows
    // likewise); one block-wide prefix sum per array then gives the per-column / per-row key counts.
    __shared__ uint32_t lds_hx[257], lds_hy[257];
    __shared__ uint32_t s_ws[2][4];
    lds_hx[threadIdx.x] = 0;
    lds_hy[threadIdx.x] = 0;
    if (threadIdx.x == 0) lds_hx[256] = lds_hy[256] = 0;
    __syncthreads();
    const int r = blockIdx.x * kChunk + threadIdx.x;
    const uint32_t packed = (r < n) ? rect_packed[r] : 0u;
    if (packed) {
        const uint32_t x0 = packed & 0xFFu, w = (packed >> 8) & 0xFFu, y0 = (packed >> 16) & 0xFFu, h = packed >> 24;
        atomicAdd(&lds_hx[x0], h);
        atomicSub(&lds_hx[x0 + w], h);
        atomicAdd(&lds_hy[y0], w);
        atomicSub(&lds_hy[y0 + h], w);
    }
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t ix = lds_hx[threadIdx.x], iy = lds_hy[threadIdx.x];
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t ox = __shfl_up(ix, off, kWave), oy = __shfl_up(iy, off, kWave);
        if (lane >= off) { ix += ox; iy += oy; }
    }
    if (lane == kWave - 1) { s_ws[0][wave] = ix; s_ws[1][wave] = iy; }
    __syncthreads();
    for (int ww = 0; ww < wave; ++ww) { ix += s_ws[0][ww]; iy += s_ws[1][ww]; }
    // one table row per chunk: [keys per tile column | keys per tile row]. The row counts are summed
    // down the chunks by the column scan below (its totals are the digit histogram of the tile-row
    // pass); adding them with global atomics instead would serialise every chunk on the same words.
    uint32_t* row = table + (size_t)blockIdx.x * (stride_x + stride_y);
    if ((int)threadIdx.x < stride_x) row[threadIdx.x] = ix;
    if ((int)threadIdx.x < stride_y) row[stride_x + threadIdx.x] = iy;
}

// ---- exclusive prefix of table[row][x] down the rows, for every column x ---------------------
__global__ __launch_bounds__(512) void colscan_reduce_kernel(const uint32_t* __restrict__ table, uint32_t rows, int stride,
                                                             uint32_t* __restrict__ partial) {
    const uint32_t r0 = blockIdx.x * kRowsPerBlock, r1 = min(rows, r0 + kRowsPerBlock);
    uint32_t s = 0;
#pragma unroll 8
    for (uint32_t r = r0; r < r1; ++r) s += table[(size_t)r * stride + threadIdx.x];
    partial[(size_t)blockIdx.x * stride + threadIdx.x] = s;
}

// One workgroup: per column the exclusive prefix of the block partials, then the exclusive prefix
// across columns of the column totals (= where column x starts in the output).
__global__ __launch_bounds__(512) void colscan_partials_kernel(uint32_t* __restrict__ partial, uint32_t blocks, int stride,
                                                               int stride_x, uint32_t* __restrict__ colbase,
                                                               uint32_t* __restrict__ hist_y) {
    __shared__ uint32_t s_ws[8];
    uint32_t running = 0;
#pragma unroll 8
    for (uint32_t b = 0; b < blocks; ++b) {
        const uint32_t v = partial[(size_t)b * stride + threadIdx.x];
        partial[(size_t)b * stride + threadIdx.x] = running;
        running += v;
    }
    // columns >= stride_x hold the per-tile-row counts: their totals are the row histogram
    if ((int)threadIdx.x >= stride_x) hist_y[(int)threadIdx.x - stride_x] = running;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t incl = ((int)threadIdx.x < stride_x) ? running : 0u;
    running = incl;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += o;
    }
    if (lane == kWave - 1) s_ws[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += s_ws[w];
    if ((int)threadIdx.x < stride_x) colbase[threadIdx.x] = base + incl - running;
}

// Only the tile-column part of a row is rewritten (blockDim.x = stride_x).
__global__ __launch_bounds__(256) void colscan_apply_kernel(uint32_t* __restrict__ table, uint32_t rows, int stride,
                                                            const uint32_t* __restrict__ partial,
                                                            const uint32_t* __restrict__ colbase) {
    const uint32_t r0 = blockIdx.x * kRowsPerBlock, r1 = min(rows, r0 + kRowsPerBlock);
    uint32_t running = partial[(size_t)blockIdx.x * stride + threadIdx.x] + colbase[threadIdx.x];
#pragma unroll 8
    for (uint32_t r = r0; r < r1; ++r) {
        const size_t cell = (size_t)r * stride + threadIdx.x;
        const uint32_t v = table[cell];
        table[cell] = running;
        running += v;
    }
}

// ---- emission ----------------------------------------------------------------------------------
__global__ __launch_bounds__(kChunk) void emit_chunk_kernel(int n, const uint32_t* __restrict__ sorted_depth,
                                                            const uint32_t* __restrict__ sorted_idx,
                                                            const uint32_t* __restrict__ rect_packed,
                                                            const uint32_t* __restrict__ table, int stride, int grid_x,
                                                            uint64_t* __restrict__ keys, uint32_t* __restrict__ values) {
    __shared__ uint32_t s_rect[kChunk];
    __shared__ uint32_t s_col[256];             // output index of the chunk's first key in column x
    __shared__ uint32_t s_mask[256][8];         // per column: which of the 256 Gaussians cover it
    __shared__ uint32_t s_wsum[256][8];         // per column and 32-Gaussian word: rows contributed
    __shared__ uint32_t s_tmp[4][256];          // per wave: run starts of the rectangle being expanded
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int g = threadIdx.x;
    // Which chunk: workgroup b runs on XCD b % 8. Consecutive chunks write adjacent pieces of every column's run (a few
    // keys each when the splats are small), so 32 consecutive chunks go to ONE XCD: the pieces of a line then meet in
    // that XCD's L2 and leave as whole lines instead of as one partial write per chunk.
    int chunk = (int)blockIdx.x;
    {
        constexpr int kPer = 32, kGroupChunks = 8 * kPer;          // (8 / 32 / 128 chunks in a row: 1.24 / 1.20 / 1.19 ms at 50 M, 1.48 before)
        const int group0 = chunk / kGroupChunks * kGroupChunks;
        if (group0 + kGroupChunks <= (int)gridDim.x) chunk = group0 + ((chunk - group0) % 8) * kPer + (chunk - group0) / 8;
    }
    const int r = chunk * kChunk + g;
    const uint32_t rect = (r < n) ? rect_packed[r] : 0u;
    if (__syncthreads_or(rect != 0u) == 0) return;              // (nothing visible in this chunk)
    s_rect[g] = rect;
    s_col[g] = (g < stride) ? table[(size_t)chunk * stride + g] : 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        s_mask[g][k] = 0;
        s_wsum[g][k] = 0;
    }
    __syncthreads();
    const uint32_t x0 = rect & 0xFFu, w = (rect >> 8) & 0xFFu, y0 = (rect >> 16) & 0xFFu, h = rect >> 24;
    for (uint32_t c = 0; c < w; ++c) {
        atomicOr(&s_mask[x0 + c][g >> 5], 1u << (g & 31));
        atomicAdd(&s_wsum[x0 + c][g >> 5], h);
    }
    __syncthreads();
    // rows of the chunk's earlier Gaussians in column x = whole 32-Gaussian words + the set bits
    // below g in its own word
    auto run_start = [&](uint32_t gg, uint32_t x) -> uint32_t {
        uint32_t s = s_col[x];
        const uint32_t word = gg >> 5;
        for (uint32_t k = 0; k < word; ++k) s += s_wsum[x][k];
        uint32_t m = s_mask[x][word] & ((1u << (gg & 31)) - 1u);
        while (m) {
            const uint32_t b = (uint32_t)__ffs((int)m) - 1u;
            m &= m - 1u;
            s += s_rect[(word << 5) + b] >> 24;
        }
        return s;
    };
    const uint32_t cnt = w * h;
    const uint32_t depth = cnt ? sorted_depth[r] : 0u, idx = cnt ? sorted_idx[r] : 0u;
    // small rectangles: the owning lane walks its few tiles itself
    if (cnt > 0 && cnt <= (uint32_t)kSmallRect) {
        for (uint32_t c = 0; c < w; ++c) {
            const uint32_t base = run_start((uint32_t)g, x0 + c);
            for (uint32_t yy = 0; yy < h; ++yy)
                emit1(keys, values, base + yy, __umul24(y0 + yy, (uint32_t)grid_x) + x0 + c, depth, idx);
        }
    }
    // large rectangles: one at a time; first the lanes resolve the run start of every column, then
    // the 64 lanes walk the rectangle in pairs of rows (pair k -> column k / hp, rows 2 (k % hp), +1)
    unsigned long long big = __ballot(cnt > (uint32_t)kSmallRect);
    uint32_t* tmp = s_tmp[wave];
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t sx0 = __shfl(x0, src, kWave), sw = __shfl(w, src, kWave), sy0 = __shfl(y0, src, kWave);
        const uint32_t sh = __shfl(h, src, kWave), sdepth = __shfl(depth, src, kWave), sidx = __shfl(idx, src, kWave);
        const uint32_t sg = (uint32_t)(wave << 6) + (uint32_t)src;
        for (uint32_t c = (uint32_t)lane; c < sw; c += kWave) tmp[c] = run_start(sg, sx0 + c);
        // (wave-private LDS: the writes above are ordered before the reads below inside the wave)
        const uint32_t hp = (sh + 1u) >> 1, npairs = sw * hp;
        const float inv_hp = 1.0f / (float)hp;
        for (uint32_t k = (uint32_t)lane; k < npairs; k += kWave) {
            uint32_t c = (uint32_t)((float)k * inv_hp);             // k < 2^16: off by at most one
            int j = (int)k - (int)__umul24(c, hp);
            if (j < 0) { --c; j += (int)hp; }
            if (j >= (int)hp) { ++c; j -= (int)hp; }
            const uint32_t yy = 2u * (uint32_t)j;
            const uint32_t pos = tmp[c] + yy;
            const uint32_t tile = __umul24(sy0 + yy, (uint32_t)grid_x) + sx0 + c;
            if (yy + 1u < sh) emit2(keys, values, pos, tile, (uint32_t)grid_x, sdepth, sidx);
            else emit1(keys, values, pos, tile, sdepth, sidx);
        }
    }
}

inline size_t align128(size_t v) { return (v + 127) / 128 * 128; }

}  // namespace

int emit_stride(int grid) { return (grid + 63) / 64 * 64; }

size_t emit_scratch_bytes(size_t n) {
    const size_t chunks = (n + kChunk - 1) / kChunk;
    const size_t blocks = (chunks + kRowsPerBlock - 1) / kRowsPerBlock;
    return align128(chunks * 512 * 4) + align128(blocks * 512 * 4) + align128(256 * 4);
}

// sorted_depth / sorted_idx / rect_packed: the n visible Gaussians in depth order. hist_y: out, 256 counters (keys per tile row).
int launch_emit_columns(int n, const uint32_t* sorted_depth, const uint32_t* sorted_idx, const uint32_t* rect_packed,
                        int grid_x, int grid_y, char* scratch, uint32_t* hist_y, uint64_t* keys,
                        uint32_t* values, hipStream_t stream, hipEvent_t mark_prep_end, hipEvent_t mark_emit_begin) {
    const int stride_x = emit_stride(grid_x), stride_y = emit_stride(grid_y), stride = stride_x + stride_y;
    const uint32_t chunks = (uint32_t)((n + kChunk - 1) / kChunk);
    const uint32_t blocks = (chunks + kRowsPerBlock - 1) / kRowsPerBlock;
    uint32_t* table = reinterpret_cast<uint32_t*>(scratch);
    uint32_t* partial = reinterpret_cast<uint32_t*>(scratch + align128((size_t)chunks * 512 * 4));
    uint32_t* colbase = reinterpret_cast<uint32_t*>(scratch + align128((size_t)chunks * 512 * 4) + align128((size_t)blocks * 512 * 4));
    GSR_HIP_TRY(hipMemsetAsync(hist_y, 0, 256 * sizeof(uint32_t), stream));
    hipLaunchKernelGGL(column_count_kernel, dim3(chunks), dim3(kChunk), 0, stream, n, rect_packed, stride_x, stride_y, table);
    GSR_LAUNCH_CHECK("column_count_kernel");
    hipLaunchKernelGGL(colscan_reduce_kernel, dim3(blocks), dim3(stride), 0, stream, table, chunks, stride, partial);
    GSR_LAUNCH_CHECK("colscan_reduce_kernel");
    hipLaunchKernelGGL(colscan_partials_kernel, dim3(1), dim3(stride), 0, stream, partial, blocks, stride, stride_x, colbase, hist_y);
    GSR_LAUNCH_CHECK("colscan_partials_kernel");
    hipLaunchKernelGGL(colscan_apply_kernel, dim3(blocks), dim3(stride_x), 0, stream, table, chunks, stride, partial, colbase);
    GSR_LAUNCH_CHECK("colscan_apply_kernel");
    if (mark_prep_end) GSR_HIP_TRY(hipEventRecord(mark_prep_end, stream));
    if (mark_emit_begin) GSR_HIP_TRY(hipEventRecord(mark_emit_begin, stream));
    hipLaunchKernelGGL(emit_chunk_kernel, dim3(chunks), dim3(kChunk), 0, stream, n, sorted_depth, sorted_idx, rect_packed, table,
                       stride, grid_x, keys, values);
    GSR_LAUNCH_CHECK("emit_chunk_kernel");
    return GSR_OK;
}

}  // namespace gsr
