// Column-major key emission (tile grids up to 255 x 255): the first of the two tile passes of the
// sort — a stable pass on the tile column x — is not run as a sort, its result is written directly.
//
// Reference semantics: apps/gsrast/gscuda/GSCuda.cu:422-475 (duplicateWithKeys) emits, per
// Gaussian, one (tile << 32 | depth bits, index) pair per covered tile; :794-797 then sorts them.
// Here Gaussians arrive in depth order (sorted once per Gaussian, before duplication). In that
// order the keys of column x are, Gaussian after Gaussian, the h rows of its rectangle, so key
// (g, x, y) belongs at
//     start[x] + (rows of all earlier Gaussians covering x) + (y - y0).
//   column_count_kernel : per chunk of 512 depth-consecutive Gaussians, keys per column
//                         (difference arrays, 4 LDS atomics per Gaussian) -> table[chunk][x]
//   colscan_*           : exclusive prefix of the table down the chunks for every column, plus the
//                         column starts: table[chunk][x] becomes the output index of the chunk's
//                         first key in column x (three small launches, all row-coalesced)
//   emit_chunk_kernel   : resolves the order INSIDE a chunk with per-column bit masks of covering
//                         Gaussians (+ per-32-Gaussian row sums), then writes the chunk's keys in
//                         (column, position) order, one key per lane: coalesced runs per column.
// What reaches the remaining pass (stable on the tile row y) is exactly what a stable x pass over
// the depth-ordered list would have produced, so the sorted list is bit-identical.
#include <stdlib.h>

#include "gsr_common.hpp"

namespace gsr {
namespace {

constexpr int kChunk = 512;        // Gaussians per workgroup
constexpr int kRowsPerBlock = 256; // table rows per workgroup of the column scan
constexpr int kCols = 256;         // tile columns / rows the tables are laid out for (grids up to 255 x 255)

// rect packed as x0 | w << 8 | y0 << 16 | h << 24 (all < 256); 0 = culled / empty
// rect_packed: the visible Gaussians' rectangles in depth order (they travel through the depth sort with the indices;
// gathering them by index here cost 0.93 ms on 50 M Gaussians)
constexpr int kCountChunks = 4;    // chunks per workgroup of column_count_kernel: one round trip to memory, two barriers and one
                                   // wave-wide scan per table row for four chunks (a chunk a workgroup, at 50 M Gaussians: 90 K
                                   // workgroups of 2.2 us, 0.20 ms in all; four chunks one after the other: 0.19)
static_assert(kCountChunks * 2 == kChunk / kWave, "one wave per (chunk, columns | rows) array");
__global__ __launch_bounds__(kChunk) void column_count_kernel(int n, const uint32_t* __restrict__ rect_packed, int stride_x,
                                                              int stride_y, uint32_t* __restrict__ table, uint32_t chunks) {
    // Difference arrays: a w x h rectangle adds h at column x0 and takes it back at x0 + w (rows likewise); one prefix sum
    // per array then gives the per-column / per-row key counts. Array (c, a) — chunk c of the workgroup's four, a = 0
    // columns / 1 rows — is summed by wave 2 c + a alone: four consecutive entries a lane, one wave-wide scan, no barrier.
    __shared__ __attribute__((aligned(16))) uint32_t lds_h[kCountChunks][2][kCols + 4];
    uint32_t packed_of[kCountChunks];
#pragma unroll
    for (int c = 0; c < kCountChunks; ++c) {
        const uint32_t chunk = blockIdx.x * kCountChunks + (uint32_t)c;
        const long long r = (long long)chunk * kChunk + threadIdx.x;
        packed_of[c] = (chunk < chunks && r < n) ? rect_packed[r] : 0u;
    }
    for (int i = threadIdx.x; i < kCountChunks * 2 * (kCols + 4); i += kChunk) (&lds_h[0][0][0])[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kCountChunks; ++c) {
        const uint32_t packed = packed_of[c];
        if (packed) {
            const uint32_t x0 = packed & 0xFFu, w = (packed >> 8) & 0xFFu, y0 = (packed >> 16) & 0xFFu, h = packed >> 24;
            atomicAdd(&lds_h[c][0][x0], h);
            atomicSub(&lds_h[c][0][x0 + w], h);
            atomicAdd(&lds_h[c][1][y0], w);
            atomicSub(&lds_h[c][1][y0 + h], w);
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int c = wave >> 1, a = wave & 1;
    uint4 v = *reinterpret_cast<const uint4*>(&lds_h[c][a][4 * lane]);
    v.y += v.x; v.z += v.y; v.w += v.z;
    uint32_t incl = v.w;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += o;
    }
    const uint32_t below = incl - v.w;
    v.x += below; v.y += below; v.z += below; v.w += below;
    // one table row per chunk: [keys per tile column | keys per tile row]. The row counts are summed down the chunks by the
    // column scan below (its totals are the digit histogram of the tile-row pass); adding them with global atomics
    // instead would serialise every chunk on the same words.
    const uint32_t chunk = blockIdx.x * kCountChunks + (uint32_t)c;
    if (chunk < chunks && 4 * lane < (a ? stride_y : stride_x))
        *reinterpret_cast<uint4*>(table + (size_t)chunk * (stride_x + stride_y) + (a ? stride_x : 0) + 4 * lane) = v;
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
    // (32 rows' loads are issued before the first of their stores — the compiler cannot know that the stores do not touch the
    // next row — : one workgroup walks 352 rows at 50 M Gaussians, 43 us a row at a time)
    for (uint32_t b0 = 0; b0 < blocks; b0 += 32u) {
        uint32_t v[32];
#pragma unroll
        for (uint32_t k = 0; k < 32u; ++k) v[k] = (b0 + k < blocks) ? partial[(size_t)(b0 + k) * stride + threadIdx.x] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 32u; ++k) {
            if (b0 + k < blocks) partial[(size_t)(b0 + k) * stride + threadIdx.x] = running;
            running += v[k];
        }
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
    // (32 rows' loads before the first of their stores, as in colscan_partials_kernel)
    for (uint32_t b0 = r0; b0 < r1; b0 += 32u) {
        uint32_t v[32];
#pragma unroll
        for (uint32_t k = 0; k < 32u; ++k) v[k] = (b0 + k < r1) ? table[(size_t)(b0 + k) * stride + threadIdx.x] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 32u; ++k) {
            if (b0 + k < r1) table[(size_t)(b0 + k) * stride + threadIdx.x] = running;
            running += v[k];
        }
    }
}

// ---- emission ----------------------------------------------------------------------------------
// The keys of a chunk leave in (column, position) order, one key per lane: consecutive lanes write consecutive keys of
// one column's run, so a wave's stores are a few whole lines. (One lane per Gaussian walking its own tiles and storing
// to memory — the first version — sent every key to a different place: two store requests per key, 250 M on the 50 M
// frame, which is what its 1.2 ms were made of.)
//   masks : per column, which of the chunk's Gaussians cover it (bit g of word g / 32) and, per word, the rows they
//           add there; prefix over the words -> where a word's keys start in the column's run of this chunk
// then one of two ways, chosen per chunk:
//   light chunks (at most kStage keys, no Gaussian with more than kLightKeys: the frames of small splats this plan is for)
//     place : every Gaussian works out where ITS keys go among the chunk's — per column of its rectangle: the column's
//             start, the rows of the words below, the rows of the set bits below its own in its word — and files them
//             (column, row, Gaussian: 25 bits) in an LDS image of the chunk's key sequence
//     keys  : lane j reads slot j and writes key and value
//   heavy chunks
//     keys  : key j of the chunk -> its column (search in the columns' starts), the word (search in the word prefix), the
//             Gaussian (walk over the word's set bits, subtracting heights) and the row inside its rectangle: 141 wave
//             instructions per 64 keys whatever the Gaussians' sizes (the placement is a loop over a lane's own keys: a
//             splat of a thousand tiles holds its wave for a thousand rounds — 0.15 -> 1.2 ms on a frame of such)
// At 50 M Gaussians (2.7 keys each) the search took 0.757 ms at 59 % vector-busy; the placement takes a third of the
// instructions.
constexpr int kStage = 2048;       // light chunks: keys at most (the LDS image)
constexpr uint32_t kLightKeys = 32;    // ... and keys per Gaussian at most

// SX: the grid's columns rounded up to 64 (the table's stride_x)
template <int SX>
__global__ __launch_bounds__(kChunk) void emit_chunk_kernel(int n, const uint32_t* __restrict__ sorted_depth,
                                                            const uint32_t* __restrict__ sorted_idx,
                                                            const uint32_t* __restrict__ rect_packed,
                                                            const uint32_t* __restrict__ table, int stride, int grid_x,
                                                            uint64_t* __restrict__ keys, uint32_t* __restrict__ values) {
    constexpr int W = kChunk / 32;                       // mask words per column
    constexpr int stride_x = SX;
    constexpr uint32_t sx = (uint32_t)SX;
    // Sized by the grid's columns: 33 KB in all up to 128 columns — four workgroups a CU.
    // [word][column]: the lanes of the build address one word of different columns, those of the prefix one column each
    __shared__ uint32_t s_dyn[(2 * W + 2) * SX];
    uint32_t* s_mask = s_dyn;                            // [W][sx]
    uint32_t* s_wpre = s_mask + W * sx;                  // [W + 1][sx] rows added by the words below; [W] = the column's keys in this chunk
    uint32_t* s_delta = s_wpre + (W + 1) * sx;           // [sx] output index of a key of the column minus its place among the chunk's
    __shared__ uint32_t s_cstart[kCols];                 // where the column's keys start among the chunk's (chunk_total past the grid's columns)
    __shared__ uint32_t s_rect[kChunk], s_depth[kChunk], s_idx[kChunk];
    __shared__ uint32_t s_stage[kStage];
    __shared__ uint32_t s_ws[kChunk / kWave];
    __shared__ uint32_t s_big;                           // some Gaussian of the chunk has more than kLightKeys keys
    const int g = threadIdx.x;
    // Which chunk: workgroup b runs on XCD b % 8. Consecutive chunks write adjacent pieces of every column's run, so 32
    // consecutive chunks go to ONE XCD: the pieces of a line then meet in that XCD's L2 and leave as whole lines.
    int chunk = (int)blockIdx.x;
    {
        constexpr int kPer = 32, kGroupChunks = 8 * kPer;
        const int group0 = chunk / kGroupChunks * kGroupChunks;
        if (group0 + kGroupChunks <= (int)gridDim.x) chunk = group0 + ((chunk - group0) % 8) * kPer + (chunk - group0) / 8;
    }
    const int r = chunk * kChunk + g;
    // (every load of the chunk is issued before the first wait: one round trip to memory per workgroup, not two)
    const uint32_t rect = (r < n) ? rect_packed[r] : 0u;
    const uint32_t depth_in = (r < n) ? sorted_depth[r] : 0u, idx_in = (r < n) ? sorted_idx[r] : 0u;
    const uint32_t col_in = (g < stride_x) ? table[(size_t)chunk * stride + g] : 0u;
    s_rect[g] = rect;
    s_depth[g] = depth_in;
    s_idx[g] = idx_in;
    if (g == 0) s_big = 0;
    // (s_mask and the first W rows of s_wpre are one run of 2 W sx words; 16 bytes per lane and store)
    for (uint32_t i = (uint32_t)g; i < 2u * W * sx / 4u; i += kChunk) reinterpret_cast<uint4*>(s_dyn)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    const uint32_t x0 = rect & 0xFFu, w = (rect >> 8) & 0xFFu, y0 = (rect >> 16) & 0xFFu, h = rect >> 24;
    const uint32_t word = (uint32_t)g >> 5, bit = 1u << (g & 31);
    for (uint32_t c = 0; c < w; ++c) {
        atomicOr(&s_mask[word * sx + x0 + c], bit);
        atomicAdd(&s_wpre[word * sx + x0 + c], h);
    }
    if (w * h > kLightKeys) s_big = 1;
    __syncthreads();
    // per column: the words' row counts -> exclusive prefix, total in [W]
    uint32_t col_total = 0;
    if (g < stride_x) {
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const uint32_t v = s_wpre[k * sx + g];
            s_wpre[k * sx + g] = col_total;
            col_total += v;
        }
        s_wpre[W * sx + g] = col_total;
    }
    uint32_t chunk_total;
    {
        // exclusive prefix of the columns' totals (non-decreasing; a column without keys shares its successor's start, and
        // everything behind the last key starts at chunk_total: never at or before a j < chunk_total)
        const int lane = g & (kWave - 1), wave = g / kWave;
        uint32_t incl = col_total;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += o;
        }
        if (lane == kWave - 1) s_ws[wave] = incl;
        __syncthreads();
        uint32_t base = 0, tot = 0;
        for (int ww = 0; ww < kChunk / kWave; ++ww) {
            if (ww < wave) base += s_ws[ww];
            tot += s_ws[ww];
        }
        chunk_total = tot;
        const uint32_t start = base + incl - col_total;  // (columns past stride_x: col_total = 0, start = chunk_total)
        if (g < kCols) s_cstart[g] = start;
        if (g < stride_x) s_delta[g] = col_in - start;   // (mod 2^32)
    }
    __syncthreads();
    if (chunk_total <= (uint32_t)kStage && s_big == 0) {
        // ---- light chunk: every Gaussian files its own keys
        for (uint32_t c = 0; c < w; ++c) {
            const uint32_t x = x0 + c;
            // the rows of the Gaussians below in the same word (set bits, their heights)
            uint32_t below = s_mask[word * sx + x] & (bit - 1u), at = s_cstart[x] + s_wpre[word * sx + x];
            while (below) {
                at += s_rect[(word << 5) + (uint32_t)__ffs((int)below) - 1u] >> 24;
                below &= below - 1u;
            }
            const uint32_t e = x | (y0 << 8) | ((uint32_t)g << 16);
            for (uint32_t row = 0; row < h; ++row) s_stage[at + row] = e + (row << 8);
        }
        __syncthreads();
        for (uint32_t j = (uint32_t)g; j < chunk_total; j += kChunk) {
            const uint32_t e = s_stage[j];
            const uint32_t x = e & 0xFFu, gg = e >> 16;
            const uint32_t tile = __umul24((e >> 8) & 0xFFu, (uint32_t)grid_x) + x;
            const uint32_t pos = j + s_delta[x];
            keys[pos] = ((uint64_t)tile << 32) | (uint64_t)s_depth[gg];
            values[pos] = s_idx[gg];
        }
        return;
    }
    // ---- heavy chunk: every key looks for its Gaussian
    const uint32_t top = stride_x > 128 ? 128u : (stride_x > 64 ? 64u : 32u);
    for (uint32_t j = (uint32_t)g; j < chunk_total; j += kChunk) {
        // the last column that starts at or before j (of columns sharing a start, the last one is the one with keys)
        uint32_t x = 0;
        for (uint32_t step = top; step >= 1; step >>= 1)
            if (s_cstart[x + step] <= j) x += step;
        const uint32_t t = j - s_cstart[x];
        uint32_t k = 0;
#pragma unroll
        for (uint32_t step = W / 2; step >= 1; step >>= 1)
            if (s_wpre[(k + step) * sx + x] <= t) k += step;
        // inside the word: Gaussian after Gaussian (set bits, ascending), each with the rows of its rectangle
        uint32_t rem = t - s_wpre[k * sx + x], m = s_mask[k * sx + x], gg = 0, gr = 0;
        for (;;) {
            gg = (k << 5) + (uint32_t)__ffs((int)m) - 1u;
            gr = s_rect[gg];
            const uint32_t gh = gr >> 24;
            m &= m - 1u;
            if (rem < gh || m == 0u) break;
            rem -= gh;
        }
        const uint32_t tile = __umul24(((gr >> 16) & 0xFFu) + rem, (uint32_t)grid_x) + x;
        const uint32_t pos = j + s_delta[x];
        keys[pos] = ((uint64_t)tile << 32) | (uint64_t)s_depth[gg];
        values[pos] = s_idx[gg];
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
    hipLaunchKernelGGL(column_count_kernel, dim3((chunks + kCountChunks - 1) / kCountChunks), dim3(kChunk), 0, stream, n, rect_packed, stride_x,
                       stride_y, table, chunks);
    GSR_LAUNCH_CHECK("column_count_kernel");
    hipLaunchKernelGGL(colscan_reduce_kernel, dim3(blocks), dim3(stride), 0, stream, table, chunks, stride, partial);
    GSR_LAUNCH_CHECK("colscan_reduce_kernel");
    hipLaunchKernelGGL(colscan_partials_kernel, dim3(1), dim3(stride), 0, stream, partial, blocks, stride, stride_x, colbase, hist_y);
    GSR_LAUNCH_CHECK("colscan_partials_kernel");
    hipLaunchKernelGGL(colscan_apply_kernel, dim3(blocks), dim3(stride_x), 0, stream, table, chunks, stride, partial, colbase);
    GSR_LAUNCH_CHECK("colscan_apply_kernel");
    if (mark_prep_end) GSR_HIP_TRY(hipEventRecord(mark_prep_end, stream));
    if (mark_emit_begin) GSR_HIP_TRY(hipEventRecord(mark_emit_begin, stream));
#define GSR_EMIT_CHUNK(SX)                                                                                                     \
    hipLaunchKernelGGL(emit_chunk_kernel<SX>, dim3(chunks), dim3(kChunk), 0, stream, n, sorted_depth, sorted_idx, rect_packed, table, \
                       stride, grid_x, keys, values)
    switch (stride_x) {
        case 64: GSR_EMIT_CHUNK(64); break;
        case 128: GSR_EMIT_CHUNK(128); break;
        case 192: GSR_EMIT_CHUNK(192); break;
        default: GSR_EMIT_CHUNK(256); break;
    }
#undef GSR_EMIT_CHUNK
    GSR_LAUNCH_CHECK("emit_chunk_kernel");
    return GSR_OK;
}

}  // namespace gsr
