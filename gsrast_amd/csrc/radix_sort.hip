// Stable LSD radix sort of (u64 key, u32 value) pairs on key bits [0, end_bit)
// (replaces cub::DeviceRadixSort::SortPairs at reference GSCuda.cu:794-797; no CUB /
// hipCUB / rocPRIM). 8-bit digits, 4096-key tiles, all ranking state in LDS.
//
// Per digit pass:
//   digit_histogram_kernel : tile -> 256-bin digit counts (LDS atomics) -> global, digit-major
//   (inclusive scan of the digit-major table: scan.hip)
//   scatter_kernel         : tile re-read; stable ranks from wave64 __ballot match groups +
//                            per-wave LDS counters; keys/values permuted through LDS so each
//                            digit run leaves as one contiguous burst; scattered to
//                            table[digit][tile] + position in run.
// Stability: a tile's keys are ranked in (wave, item, lane) order, which is their index
// order, and tiles are laid out in index order by the digit-major scan — so equal keys keep
// ascending input order (the tie rule SURVEY.md §8a row a9 requires).
// The input arrays are never written: pass 1 reads them and the remaining passes ping-pong
// between the output arrays and a scratch copy inside `temp`.
#include "gsr_common.hpp"

namespace gsr {
namespace {

constexpr int kBits = 8;
constexpr int kRadix = 1 << kBits;
constexpr int kThreads = 256;
constexpr int kWaves = kThreads / kWave;
constexpr int kItems = 16;                              // keys per lane
constexpr int kSortTile = kThreads * kItems;            // 4096 keys per workgroup
constexpr int kWaveSpan = kWave * kItems;               // 1024 consecutive keys per wave

__device__ __forceinline__ uint32_t digit_of(uint64_t key, int shift) {
    return (uint32_t)(key >> shift) & (uint32_t)(kRadix - 1);
}

__global__ __launch_bounds__(kThreads) void digit_histogram_kernel(const uint64_t* __restrict__ keys, size_t n, int shift,
                                                                    uint32_t* __restrict__ table, uint32_t num_tiles) {
    __shared__ uint32_t hist[kRadix];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * kSortTile;
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const size_t e = base + (size_t)i * kThreads + threadIdx.x;
        if (e < n) atomicAdd(&hist[digit_of(keys[e], shift)], 1u);
    }
    __syncthreads();
    table[(size_t)threadIdx.x * num_tiles + blockIdx.x] = hist[threadIdx.x];
}

__global__ __launch_bounds__(kThreads) void scatter_kernel(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                            uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                            size_t n, int shift, const uint32_t* __restrict__ table_incl,
                                                            uint32_t num_tiles) {
    __shared__ uint32_t wave_hist[kWaves][kRadix];
    __shared__ uint32_t run_start[kRadix];      // first position of digit d inside the sorted tile
    __shared__ uint32_t global_start[kRadix];   // first output index of this tile's digit-d run
    __shared__ uint32_t scan_ws[kWaves];
    __shared__ uint64_t stage_keys[kSortTile];
    __shared__ uint32_t stage_vals[kSortTile];

    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const size_t tile_base = (size_t)blockIdx.x * kSortTile;
    const uint32_t valid = (uint32_t)((n - tile_base < (size_t)kSortTile) ? (n - tile_base) : (size_t)kSortTile);

#pragma unroll
    for (int w = 0; w < kWaves; ++w) wave_hist[w][threadIdx.x] = 0;
    {   // exclusive global offsets of this tile's runs, from the inclusive digit-major scan
        const size_t cell = (size_t)threadIdx.x * num_tiles + blockIdx.x;
        global_start[threadIdx.x] = (cell == 0) ? 0u : table_incl[cell - 1];
    }

    uint64_t key[kItems];
    uint32_t rank[kItems];
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const uint32_t local = (uint32_t)(wave * kWaveSpan + i * kWave + lane);
        key[i] = (local < valid) ? keys_in[tile_base + local] : ~0ull;   // padding ranks last in the top digit
    }
    __syncthreads();

    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const uint32_t d = digit_of(key[i], shift);
        unsigned long long peers = ~0ull;
#pragma unroll
        for (int b = 0; b < kBits; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t below = (uint32_t)__popcll(peers & lt_mask);
        uint32_t prior = 0;
        if (below == 0) {                       // lowest lane of the match group owns the counter
            prior = wave_hist[wave][d];
            wave_hist[wave][d] = prior + (uint32_t)__popcll(peers);
        }
        prior = __shfl(prior, __ffsll((long long)peers) - 1, kWave);
        rank[i] = prior + below;
    }
    __syncthreads();

    {   // per digit: exclusive offsets across waves, then across digits
        const int d = threadIdx.x;
        uint32_t acc = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const uint32_t c = wave_hist[w][d];
            wave_hist[w][d] = acc;
            acc += c;
        }
        // block exclusive scan of acc over the 256 digits
        uint32_t incl = acc;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += o;
        }
        if (lane == kWave - 1) scan_ws[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w)
            if (w < wave) wbase += scan_ws[w];
        run_start[d] = wbase + incl - acc;
    }
    __syncthreads();

#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const uint32_t d = digit_of(key[i], shift);
        rank[i] += run_start[d] + wave_hist[wave][d];
        stage_keys[rank[i]] = key[i];
    }
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const uint32_t local = (uint32_t)(wave * kWaveSpan + i * kWave + lane);
        if (local < valid) stage_vals[rank[i]] = vals_in[tile_base + local];
    }
    __syncthreads();

#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const uint32_t p = (uint32_t)(i * kThreads) + threadIdx.x;
        if (p < valid) {
            const uint64_t k = stage_keys[p];
            const uint32_t d = digit_of(k, shift);
            const size_t dst = (size_t)global_start[d] + (p - run_start[d]);
            keys_out[dst] = k;
            vals_out[dst] = stage_vals[p];
        }
    }
}

struct SortTemp {
    uint64_t* keys;
    uint32_t* vals;
    uint32_t* table;
    char* scan_temp;
    size_t bytes;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

SortTemp carve_sort_temp(char* base, size_t n) {
    const size_t tiles = (n + kSortTile - 1) / kSortTile;
    size_t off = 0;
    SortTemp t;
    t.keys = reinterpret_cast<uint64_t*>(base + off); off = align_up(off + n * sizeof(uint64_t), 128);
    t.vals = reinterpret_cast<uint32_t*>(base + off); off = align_up(off + n * sizeof(uint32_t), 128);
    t.table = reinterpret_cast<uint32_t*>(base + off); off = align_up(off + tiles * kRadix * sizeof(uint32_t), 128);
    t.scan_temp = base + off; off = align_up(off + scan_temp_bytes(tiles * kRadix), 128);
    t.bytes = off;
    return t;
}

}  // namespace

size_t sort_temp_bytes(size_t n) { return carve_sort_temp(nullptr, n).bytes; }

int launch_sort_pairs(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* values_in,
                      uint32_t* values_out, size_t n, int end_bit, char* temp, hipStream_t stream) {
    if (n == 0) return GSR_OK;
    if (end_bit <= 0) end_bit = 1;
    if (end_bit > 64) end_bit = 64;
    const int passes = (end_bit + kBits - 1) / kBits;
    const SortTemp t = carve_sort_temp(temp, n);
    const uint32_t tiles = (uint32_t)((n + kSortTile - 1) / kSortTile);
    const uint64_t* src_k = keys_in;
    const uint32_t* src_v = values_in;
    for (int p = 0; p < passes; ++p) {
        // The last pass must land in the output arrays: odd distance from the end -> scratch.
        const bool to_out = ((passes - 1 - p) % 2) == 0;
        uint64_t* dst_k = to_out ? keys_out : t.keys;
        uint32_t* dst_v = to_out ? values_out : t.vals;
        const int shift = p * kBits;
        hipLaunchKernelGGL(digit_histogram_kernel, dim3(tiles), dim3(kThreads), 0, stream, src_k, n, shift, t.table, tiles);
        GSR_LAUNCH_CHECK("digit_histogram_kernel");
        const int rc = launch_inclusive_scan(t.table, t.table, (size_t)tiles * kRadix, t.scan_temp, stream);
        if (rc != GSR_OK) return rc;
        hipLaunchKernelGGL(scatter_kernel, dim3(tiles), dim3(kThreads), 0, stream, src_k, src_v, dst_k, dst_v, n, shift,
                           t.table, tiles);
        GSR_LAUNCH_CHECK("scatter_kernel");
        src_k = dst_k;
        src_v = dst_v;
    }
    return GSR_OK;
}

}  // namespace gsr
