// Inclusive u32 prefix sum over N elements (replaces cub::DeviceScan::InclusiveSum at
// reference GSCuda.cu:771). Reduce-then-scan in three launches; every launch reads or
// writes whole 16-byte vectors per lane, and the only scratch is one u32 per 4096-element
// tile, carved from GeometryState::scanningSpace.
//
//   1. tile_reduce : tile t (4096 elements) -> partial[t]
//   2. partial_scan: one workgroup turns partial[] into its exclusive prefix
//   3. tile_scan   : tile t re-reads its elements, scans them in registers + wave shuffles,
//                    adds partial[t] and stores
// Integer wrap-around is that of u32 addition, as in the reference; the un-wrapped 64-bit total is available
// on request (gsr_forward refuses frames whose instance count does not fit the u32 offsets).
#include "gsr_common.hpp"

namespace gsr {
namespace {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 16;                               // per lane: 4 x uint4
constexpr int kScanTile = kScanThreads * kScanItems;         // 4096
static_assert(kScanTile == 4096, "the depth order's compaction (radix_sort.hip) takes its per-chunk offsets from this scan's tiles");

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t o = __shfl_up(v, off, kWave);
        if ((int)(threadIdx.x & (kWave - 1)) >= off) v += o;
    }
    return v;
}

// Block-wide exclusive prefix of one value per thread; also returns the block total.
template <int THREADS>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds_wave_sums, uint32_t& total) {
    constexpr int kWaves = THREADS / kWave;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint32_t incl = wave_inclusive_scan(v);
    if (lane == kWave - 1) lds_wave_sums[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        const uint32_t s = lds_wave_sums[w];
        if (w < wave) wave_base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return wave_base + incl - v;
}

// Loads the 16 items of this lane: item k of lane l is element tile*4096 + (k/4)*1024 + 4*l + k%4,
// so each of the four uint4 loads of a wave covers 1 KiB contiguous.
__device__ __forceinline__ void load_items(const uint32_t* in, size_t n, size_t tile, uint32_t (&v)[kScanItems]) {
    const size_t base = tile * kScanTile;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const size_t e = base + (size_t)q * 1024 + 4 * (size_t)threadIdx.x;
        if (e + 3 < n) {
            const uint4 x = *reinterpret_cast<const uint4*>(in + e);
            v[4 * q + 0] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[4 * q + k] = (e + k < n) ? in[e + k] : 0u;
        }
    }
}

// big / big_from (big may be null): also the sum of the tile's elements that are >= big_from (a tile's own sum cannot wrap) — gsr_forward's
// choice of the binning plan wants to know how much of the frame's instances belongs to splats that cover hundreds of tiles.
// nonzero (may be null): also the number of non-zero elements of the tile — gsr_forward's depth order compacts the
// Gaussians with tilesTouched != 0 in chunks of the same 4096 elements, and this kernel has them in registers anyway.
// clear / clear_vecs (may be null / 0): 16-byte words this launch also zeroes, a slice per thread — gsr_forward's depth
// order wants its look-back words cleared before its first pass, and a memset of its own is one more 5 us stop on a chain
// of small launches.
__global__ __launch_bounds__(kScanThreads) void tile_reduce_kernel(const uint32_t* __restrict__ in, size_t n,
                                                                    uint32_t* __restrict__ partial, uint32_t* __restrict__ nonzero,
                                                                    uint4* __restrict__ clear, size_t clear_vecs,
                                                                    uint32_t* __restrict__ big, uint32_t big_from) {
    __shared__ uint32_t wave_sums[kScanThreads / kWave], wave_nz[kScanThreads / kWave], wave_big[kScanThreads / kWave];
    for (size_t i = (size_t)blockIdx.x * kScanThreads + threadIdx.x; i < clear_vecs; i += (size_t)gridDim.x * kScanThreads)
        clear[i] = make_uint4(0u, 0u, 0u, 0u);
    uint32_t v[kScanItems];
    load_items(in, n, blockIdx.x, v);
    uint32_t s = 0, z = 0, b = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) { s += v[k]; z += v[k] != 0u ? 1u : 0u; b += v[k] >= big_from ? v[k] : 0u; }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        s += __shfl_down(s, off, kWave); z += __shfl_down(z, off, kWave); b += __shfl_down(b, off, kWave);
    }
    if ((threadIdx.x & (kWave - 1)) == 0) { wave_sums[threadIdx.x / kWave] = s; wave_nz[threadIdx.x / kWave] = z; wave_big[threadIdx.x / kWave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0, tz = 0, tb = 0;
#pragma unroll
        for (int w = 0; w < kScanThreads / kWave; ++w) { t += wave_sums[w]; tz += wave_nz[w]; tb += wave_big[w]; }
        partial[blockIdx.x] = t;
        if (big) big[blockIdx.x] = tb;
        if (nonzero) nonzero[blockIdx.x] = tz;
    }
}

// The same from the preprocess's per-wave records (store_wave_sums, gsr_common.hpp: {sum, non-zero, big, others} per 64
// elements): a wave per tile — its 64 records, one per lane —, four tiles a workgroup; `in` is not read again.
__global__ __launch_bounds__(kScanThreads) void wave_sums_reduce_kernel(const uint4* __restrict__ wave_sums, size_t waves, size_t tiles,
                                                                         uint32_t* __restrict__ partial, uint32_t* __restrict__ nonzero,
                                                                         uint4* __restrict__ clear, size_t clear_vecs,
                                                                         uint32_t* __restrict__ main_count, uint32_t* __restrict__ big) {
    for (size_t i = (size_t)blockIdx.x * kScanThreads + threadIdx.x; i < clear_vecs; i += (size_t)gridDim.x * kScanThreads)
        clear[i] = make_uint4(0u, 0u, 0u, 0u);
    const int lane = threadIdx.x & (kWave - 1);
    const size_t tile = (size_t)blockIdx.x * (kScanThreads / kWave) + threadIdx.x / kWave;
    if (tile >= tiles) return;                               // (whole waves)
    const size_t w = tile * (kScanTile / kWave) + (size_t)lane;
    uint4 v = w < waves ? wave_sums[w] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        v.x += __shfl_down(v.x, off, kWave); v.y += __shfl_down(v.y, off, kWave);
        v.z += __shfl_down(v.z, off, kWave); v.w += __shfl_down(v.w, off, kWave);
    }
    if (lane == 0) {
        partial[tile] = v.x;
        nonzero[tile] = v.y;
        big[tile] = v.z;
        main_count[tile] = v.y - v.w;
    }
}

// total64 (may be null): the sum of all elements WITHOUT the u32 wrap-around, so that the caller can tell
// whether the scan's last element is the true total (a tile's own sum cannot wrap: 4096 elements of at most
// 2^20 tiles each).
// nonzero / nonzero_total (may be null): the per-tile non-zero counts become their exclusive prefix, the total goes out.
// host_words (may be null): mapped host memory; [0] also gets the non-zero total, [2..3] (8-byte aligned) the 64-bit total —
// the caller's host thread reads them after an event, without a copy command in between.
__global__ __launch_bounds__(1024) void partial_scan_kernel(uint32_t* __restrict__ partial, size_t tiles,
                                                            unsigned long long* __restrict__ total64,
                                                            uint32_t* __restrict__ nonzero, uint32_t* __restrict__ nonzero_total,
                                                            uint32_t* __restrict__ host_words, uint32_t* __restrict__ main_count,
                                                            uint32_t side_max, uint32_t* __restrict__ side_words,
                                                            const uint32_t* __restrict__ big) {
    __shared__ uint32_t wave_sums[1024 / kWave];
    __shared__ unsigned long long wide_sums[1024 / kWave], big_sums[1024 / kWave];
    // Thread t owns the `per` consecutive tiles [t per, (t + 1) per): its own sums first (every load issued before the first
    // is needed), ONE block-wide scan per array, then its tiles' prefixes. (A loop of 1 024 tiles a round, three block scans
    // each, took 39 us for the 12 208 tiles of 50 M Gaussians: twelve rounds of dependent round trips and barriers.)
    constexpr int kMaxPer = 16;                                  // tiles per thread kept in registers (16 K tiles = 67 M elements); beyond: re-read
    const size_t per = (tiles + 1023) / 1024;
    const size_t t0 = (size_t)threadIdx.x * per < tiles ? (size_t)threadIdx.x * per : tiles, t1 = t0 + per < tiles ? t0 + per : tiles;
    uint32_t v[kMaxPer], z[kMaxPer], m[kMaxPer];
    uint32_t sum_v = 0, sum_z = 0, sum_m = 0;
    unsigned long long wide = 0, wide_big = 0;
    if (per <= (size_t)kMaxPer) {
#pragma unroll
        for (int k = 0; k < kMaxPer; ++k) {
            const size_t i = t0 + (size_t)k;
            const bool in = i < t1;
            v[k] = in ? partial[i] : 0u;
            z[k] = (in && nonzero) ? nonzero[i] : 0u;
            m[k] = (in && main_count) ? main_count[i] : 0u;
            if (in && big) wide_big += big[i];
        }
#pragma unroll
        for (int k = 0; k < kMaxPer; ++k) { sum_v += v[k]; sum_z += z[k]; sum_m += m[k]; wide += v[k]; }
    } else {
        for (size_t i = t0; i < t1; ++i) {
            const uint32_t pv = partial[i];
            sum_v += pv; wide += pv;
            if (nonzero) sum_z += nonzero[i];
            if (main_count) sum_m += main_count[i];
            if (big) wide_big += big[i];
        }
    }
    uint32_t total_v, total_z = 0, total_m = 0;
    uint32_t run_v = block_exclusive_scan<1024>(sum_v, wave_sums, total_v);
    uint32_t run_z = nonzero ? block_exclusive_scan<1024>(sum_z, wave_sums, total_z) : 0u;
    uint32_t run_m = main_count ? block_exclusive_scan<1024>(sum_m, wave_sums, total_m) : 0u;
    if (per <= (size_t)kMaxPer) {
#pragma unroll
        for (int k = 0; k < kMaxPer; ++k) {
            const size_t i = t0 + (size_t)k;
            if (i < t1) {
                partial[i] = run_v; run_v += v[k];
                if (nonzero) { nonzero[i] = run_z; run_z += z[k]; }
                if (main_count) { main_count[i] = run_m; run_m += m[k]; }
            }
        }
    } else {
        for (size_t i = t0; i < t1; ++i) {
            const uint32_t pv = partial[i];
            partial[i] = run_v; run_v += pv;
            if (nonzero) { const uint32_t pz = nonzero[i]; nonzero[i] = run_z; run_z += pz; }
            if (main_count) { const uint32_t pm = main_count[i]; main_count[i] = run_m; run_m += pm; }
        }
    }
    const uint32_t carry_nz = total_z, carry_main = total_m;
    // main_count / side_words (depth order, radix_sort.hip): the keys whose top byte is not the main one go a side way when
    // there are few of them. side_words[0] = 1 if so, [1] and [2] = 0 (the side list's counters), and the count the sort
    // passes read (*nonzero_total) is then that of the main keys; host_words[0] keeps the count of ALL non-zero elements,
    // host_words[6] = the side flag, host_words[8] = how many keys go the side way.
    if (nonzero && threadIdx.x == 0) {
        uint32_t stream = carry_nz, side = 0u;
        if (main_count) {
            const uint32_t others = carry_nz - carry_main;
            side = (carry_main != 0u && others != 0u && others <= side_max) ? 1u : 0u;
            if (side) stream = carry_main;
            side_words[0] = side; side_words[1] = 0u; side_words[2] = 0u;
            if (host_words) { host_words[6] = side; host_words[8] = side ? others : 0u; }
        }
        *nonzero_total = stream;
        if (host_words) host_words[0] = carry_nz;
    }
    if (total64) {
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) { wide += __shfl_down(wide, off, kWave); wide_big += __shfl_down(wide_big, off, kWave); }
        if ((threadIdx.x & (kWave - 1)) == 0) { wide_sums[threadIdx.x / kWave] = wide; big_sums[threadIdx.x / kWave] = wide_big; }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t = 0, tb = 0;
#pragma unroll
            for (int w = 0; w < 1024 / kWave; ++w) { t += wide_sums[w]; tb += big_sums[w]; }
            *total64 = t;
            if (host_words) {
                *reinterpret_cast<unsigned long long*>(host_words + 2) = t;
                if (big) *reinterpret_cast<unsigned long long*>(host_words + 10) = tb;      // (the instances of splats of big_from tiles and more)
            }
        }
    }
}

__global__ __launch_bounds__(kScanThreads) void tile_scan_kernel(const uint32_t* in, uint32_t* out,
                                                                  size_t n, const uint32_t* __restrict__ partial) {
    __shared__ uint32_t wave_sums[kScanThreads / kWave];
    uint32_t v[kScanItems];
    load_items(in, n, blockIdx.x, v);
    // Order of elements inside the tile: chunk q (1024 elements) -> lane -> 4 items.
    uint32_t running = partial[blockIdx.x];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v[4 * q + 1] += v[4 * q + 0];
        v[4 * q + 2] += v[4 * q + 1];
        v[4 * q + 3] += v[4 * q + 2];
        uint32_t total;
        const uint32_t excl = block_exclusive_scan<kScanThreads>(v[4 * q + 3], wave_sums, total);
        const uint32_t add = running + excl;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[4 * q + k] += add;
        running += total;
    }
    const size_t base = (size_t)blockIdx.x * kScanTile;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const size_t e = base + (size_t)q * 1024 + 4 * (size_t)threadIdx.x;
        if (e + 3 < n) {
            *reinterpret_cast<uint4*>(out + e) = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (e + k < n) out[e + k] = v[4 * q + k];
        }
    }
}

}  // namespace

size_t scan_temp_bytes(size_t n) {
    const size_t tiles = (n + kScanTile - 1) / kScanTile;
    return ((tiles + 1) * sizeof(uint32_t) + 127) / 128 * 128;
}

int launch_inclusive_scan(const uint32_t* in, uint32_t* out, size_t n, char* temp, hipStream_t stream,
                          unsigned long long* total64, uint32_t* nonzero, uint32_t* nonzero_total, uint32_t* host_words,
                          void* clear, size_t clear_bytes, const uint4* wave_sums, uint32_t* main_count,
                          uint32_t side_max, uint32_t* side_words, uint32_t* big, uint32_t big_from) {
    if (n == 0) return GSR_OK;
    const size_t tiles = (n + kScanTile - 1) / kScanTile;
    uint32_t* partial = reinterpret_cast<uint32_t*>(temp);
    if (clear_bytes % 16 != 0 || (reinterpret_cast<uintptr_t>(clear) & 15) != 0 || (host_words && !total64)) return GSR_ERR_INVALID_ARG;
    if ((wave_sums != nullptr) != (main_count != nullptr) || (main_count && (!nonzero || !side_words || !big))) return GSR_ERR_INVALID_ARG;
    if (big && !host_words) return GSR_ERR_INVALID_ARG;
    if (wave_sums) {
        // (the preprocess has left the sums of every 64 elements: big_from is the one IT was given)
        constexpr unsigned kTilesPerGroup = kScanThreads / kWave;
        hipLaunchKernelGGL(wave_sums_reduce_kernel, dim3((unsigned)((tiles + kTilesPerGroup - 1) / kTilesPerGroup)), dim3(kScanThreads), 0, stream,
                           wave_sums, (n + kWave - 1) / kWave, tiles, partial, nonzero, reinterpret_cast<uint4*>(clear),
                           clear ? clear_bytes / 16 : (size_t)0, main_count, big);
        GSR_LAUNCH_CHECK("wave_sums_reduce_kernel");
    } else {
        hipLaunchKernelGGL(tile_reduce_kernel, dim3((unsigned)tiles), dim3(kScanThreads), 0, stream, in, n, partial, nonzero,
                           reinterpret_cast<uint4*>(clear), clear ? clear_bytes / 16 : (size_t)0, big, big_from);
        GSR_LAUNCH_CHECK("tile_reduce_kernel");
    }
    hipLaunchKernelGGL(partial_scan_kernel, dim3(1), dim3(1024), 0, stream, partial, tiles, total64, nonzero, nonzero_total, host_words,
                       main_count, side_max, side_words, big);
    GSR_LAUNCH_CHECK("partial_scan_kernel");
    hipLaunchKernelGGL(tile_scan_kernel, dim3((unsigned)tiles), dim3(kScanThreads), 0, stream, in, out, n, partial);
    GSR_LAUNCH_CHECK("tile_scan_kernel");
    return GSR_OK;
}

}  // namespace gsr
