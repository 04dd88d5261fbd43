// Depth order of the visible Gaussians in TWO memory passes (replaces the low 32 bits' share of
// cub::DeviceRadixSort::SortPairs at reference GSCuda.cu:794-797; the order produced is the one that stable sort
// gives: ascending depth bits, ties by ascending index, as GSCuda.cu:447,461-474 emits them).
//
// The LSD depth order (radix_sort.hip) moves every visible (key, index, rectangle) triple three or four times, and each
// pass is a chain of launch, load, rank, look-back, store that the 3 M keys of a frame cannot fill the chip with. Here:
//   samples   : one Gaussian in 256, picked by a hash of its group number, is looked at by the scan of tilesTouched
//               (scan.hip): its (depth bits << 32 | index) if it has a tile. No atomics: slot = group.
//   splitters : the samples are sorted by 128 workgroups (a coarse split by a sub-sample every workgroup ranks for
//               itself, then each workgroup ranks the samples of its own range) and every 8th becomes a splitter: buckets
//               of about 2048 Gaussians WHATEVER the distribution of the depths (they bunch near NDC z = 1, so fixed top
//               bits would not balance), and whatever the ties — the splitters are (key, index) pairs.
//   scatter   : every visible Gaussian finds its bucket (binary search in LDS), takes a slot of the bucket's region
//               (LDS counts per workgroup, one global atomic per workgroup and bucket) and writes one 16-byte entry.
//   sort      : one workgroup per bucket sorts it wholly in LDS by (key, index) — the order in which the entries arrived
//               does not matter — with 8-bit LSD passes over the bits that vary inside the bucket only, and writes its
//               piece of the three depth-ordered arrays. Its offset is the sum of the bucket counts before it.
// A bucket that outgrows its region (8192 entries: four times the mean; the sample makes that a 1e-7 event) raises a flag
// the host reads together with numRendered, and the frame falls back to the LSD passes.
#include <stdlib.h>

#include "gsr_common.hpp"
#include "depth_buckets.hpp"

namespace gsr {
namespace {

typedef unsigned long long u64;
constexpr u64 kNoSample = ~0ull;

constexpr int kSplitThreads = 1024;
constexpr int kSplitWGs = 128;                // workgroups of the splitter kernel = coarse ranges of the samples
constexpr int kSplitListCap = 4096;           // samples one workgroup may be handed

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, kWave);
    return v;
}
// sum over the workgroup (every thread gets it); s_tmp: one word per wave
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* s_tmp) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nw = blockDim.x / kWave;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) s_tmp[wave] = v;
    __syncthreads();
    uint32_t t = 0;
    for (int w = 0; w < nw; ++w) t += s_tmp[w];
    return t;
}

// ---- splitters ---------------------------------------------------------------------------------------------------
// words: [0] buckets B, [1] failure flag (this kernel: a coarse range took more samples than a workgroup holds; the
// scatter: a bucket outgrew its region), [2] valid samples. host_fail (mapped host memory): set to 1 with words[1].
constexpr int kSplitPerThread = 32;           // samples per thread: 32 768 samples at most (depth_buckets_supported)
__global__ __launch_bounds__(kSplitThreads) void depth_splitters_kernel(const u64* __restrict__ samples, uint32_t groups,
                                                                        const uint32_t* __restrict__ visible_dev, uint32_t n,
                                                                        uint32_t max_buckets, u64* __restrict__ splitters,
                                                                        uint32_t* __restrict__ words, uint32_t* host_fail) {
    __shared__ u64 s_sub[kSplitThreads];          // the sub-sample, then sorted
    __shared__ u64 s_sorted[kSplitThreads];
    __shared__ u64 s_list[kSplitListCap];
    __shared__ uint32_t s_n0, s_nw;
    __shared__ uint32_t s_tmp[kSplitThreads / kWave];
    const uint32_t t = threadIdx.x, w = blockIdx.x;
    if (t == 0) { s_n0 = 0; s_nw = 0; }
    // every sample this thread will look at, in one round trip (they are needed again below)
    u64 mine[kSplitPerThread];
#pragma unroll
    for (int k = 0; k < kSplitPerThread; ++k) {
        const uint32_t i = (uint32_t)k * kSplitThreads + t;
        mine[k] = (i < groups) ? samples[i] : kNoSample;
    }
    // (1) a sub-sample of about 2 x kSplitWGs valid samples, the same in every workgroup
    const uint32_t visible = *visible_dev;
    uint32_t probes = 2u * kSplitWGs;
    if (visible < n) {
        const float want = 2.0f * kSplitWGs * (float)n / (float)(visible ? visible : 1u);
        probes = want > (float)kSplitThreads ? (uint32_t)kSplitThreads : (uint32_t)want;
    }
    probes = min(probes, groups);
    u64 probe = kNoSample;
    if (t < probes) probe = samples[(u64)t * groups / probes];
    __syncthreads();
    if (probe != kNoSample) s_sub[atomicAdd(&s_n0, 1u)] = probe;
    __syncthreads();
    const uint32_t n0 = s_n0;
    if (t < n0) {
        const u64 me = s_sub[t];
        uint32_t r = 0;
#pragma unroll 8
        for (uint32_t j = 0; j < n0; ++j) r += s_sub[j] < me ? 1u : 0u;       // (samples are distinct: they carry their index)
        s_sorted[r] = me;
    }
    __syncthreads();
    // (2) this workgroup's range of the samples: [lo, hi)
    u64 lo = 0, hi = kNoSample;
    if (n0 != 0) {
        if (w != 0) lo = s_sorted[(u64)w * n0 / kSplitWGs];
        if (w != kSplitWGs - 1) hi = s_sorted[(u64)(w + 1) * n0 / kSplitWGs];
    } else if (w != 0) {
        hi = 0;                                       // (no sub-sample: workgroup 0 takes everything)
    }
    // (3) all samples: how many are valid, how many lie below the range, and the list of those inside it
    uint32_t valid = 0, below = 0;
#pragma unroll
    for (int k = 0; k < kSplitPerThread; ++k) {
        const u64 s = mine[k];
        const bool ok = s != kNoSample;
        valid += ok ? 1u : 0u;
        below += (ok && s < lo) ? 1u : 0u;
        if (ok && s >= lo && s < hi) {
            const uint32_t at = atomicAdd(&s_nw, 1u);
            if (at < (uint32_t)kSplitListCap) s_list[at] = s;
        }
    }
    valid = block_sum(valid, s_tmp);
    below = block_sum(below, s_tmp);
    __syncthreads();
    const uint32_t nw = s_nw;
    if (nw > (uint32_t)kSplitListCap) {
        if (t == 0) { words[1] = 1u; *reinterpret_cast<volatile uint32_t*>(host_fail) = 1u; }
        return;
    }
    // (4) every a-th sample in sorted order is a splitter: bucket j takes the Gaussians in [splitter j-1, splitter j)
    const uint32_t buckets = max(1u, min(max_buckets, valid / kDepthOversample));
    for (uint32_t e = t; e < nw; e += kSplitThreads) {
        const u64 me = s_list[e];
        uint32_t r = below;
#pragma unroll 8
        for (uint32_t j = 0; j < nw; ++j) r += s_list[j] < me ? 1u : 0u;
        // the splitter index j with floor(j * valid / buckets) == r, if there is one
        const u64 j0 = ((u64)r * buckets + valid - 1) / valid;
        if (j0 >= 1 && j0 < buckets && (j0 * valid) / buckets == r) splitters[j0 - 1] = me;
    }
    if (w == 0 && t == 0) { words[0] = buckets; words[2] = valid; }
}

// ---- scatter -----------------------------------------------------------------------------------------------------
// A workgroup takes up to 24 576 consecutive Gaussians (one workgroup per CU where the frame allows): the more of them per
// workgroup, the fewer global atomics (one per workgroup and bucket it has an entry for) and the longer the runs of entries a
// bucket receives from it. The entries leave in (bucket, rank) order, consecutive lanes writing consecutive entries of a
// bucket's region: written by the lanes that found them — 64 lanes, 64 regions per store instruction — the scatter took
// 85 us of the bench frame, bound by the number of memory requests.
constexpr int kScatThreads = 1024, kScatMaxRows = 24, kScatBatch = 4;
constexpr int kScatTable = (int)kDepthMaxBuckets;             // splitter keys in LDS, padded with ~0: a 12-step search
static_assert(kScatThreads * kScatMaxRows <= (1 << 15), "position inside the chunk is kept in 15 bits");
static_assert(kDepthMaxBuckets == kScatThreads * 4, "four buckets per thread in the scans and the global atomics");

__global__ __launch_bounds__(kScatThreads) void depth_scatter_kernel(const uint32_t* __restrict__ depth_key,
                                                                      const uint32_t* __restrict__ rect_by_index, uint32_t n,
                                                                      uint32_t rows, const u64* __restrict__ splitters,
                                                                      uint32_t* words, uint32_t* __restrict__ counts,
                                                                      uint4* __restrict__ regions, uint32_t bucket_cap,
                                                                      uint32_t* host_fail) {
    extern __shared__ uint32_t s_stage[];        // [rows * 1024] entry (position in the chunk | bucket << 15) in (bucket, rank) order
    __shared__ uint32_t s_key[kScatTable];       // the splitters' keys
    __shared__ uint32_t s_cnt[kScatTable];       // this chunk's entries per bucket, then: region slot - staging slot of the bucket's entries
    __shared__ uint32_t s_first[kScatTable];     // staging slot of the bucket's first entry
    __shared__ uint32_t s_ws[kScatThreads / kWave];
    const uint32_t t = threadIdx.x;
    const int lane = t & (kWave - 1), wave = t / kWave;
    const uint32_t base = blockIdx.x * rows * (uint32_t)kScatThreads;
    uint32_t key[kScatBatch], rect[kScatBatch];
#pragma unroll
    for (int r = 0; r < kScatBatch; ++r) {
        const uint32_t e = base + (uint32_t)r * kScatThreads + t;
        const bool in = (uint32_t)r < rows && e < n;
        rect[r] = in ? rect_by_index[e] : 0u;
        key[r] = in ? depth_key[e] : 0u;
    }
    const uint32_t buckets = words[0];
    if (words[1] != 0u || buckets == 0u) return;                 // (the splitter kernel gave up: the frame takes the LSD passes)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t j = (uint32_t)k * kScatThreads + t;
        s_key[j] = (j + 1u < buckets) ? (uint32_t)(splitters[j] >> 32) : 0xFFFFFFFFu;
        s_cnt[j] = 0u;
    }
    __syncthreads();
    uint32_t where[kScatMaxRows];                // bucket << 15 | rank inside (chunk, bucket); ~0: no tile
#pragma unroll
    for (int b = 0; b < kScatMaxRows / kScatBatch; ++b) {
        // the next batch's loads fly during this batch's search
        uint32_t key_n[kScatBatch], rect_n[kScatBatch];
        if (b + 1 < kScatMaxRows / kScatBatch) {
#pragma unroll
            for (int r = 0; r < kScatBatch; ++r) {
                const uint32_t row = (uint32_t)((b + 1) * kScatBatch + r);
                const uint32_t e = base + row * kScatThreads + t;
                const bool in = row < rows && e < n;
                rect_n[r] = in ? rect_by_index[e] : 0u;
                key_n[r] = in ? depth_key[e] : 0u;
            }
        }
        // the number of splitters below the key, twelve steps, the batch's searches side by side; then, where the key EQUALS
        // splitter keys (one Gaussian in 250), past those whose index is not above this one's: the splitters <= (key, index)
        uint32_t pos[kScatBatch];
#pragma unroll
        for (int r = 0; r < kScatBatch; ++r) pos[r] = 0;
#pragma unroll
        for (uint32_t step = kScatTable >> 1; step >= 1u; step >>= 1) {
#pragma unroll
            for (int r = 0; r < kScatBatch; ++r) pos[r] += (s_key[pos[r] + step - 1u] < key[r]) ? step : 0u;
        }
#pragma unroll
        for (int r = 0; r < kScatBatch; ++r) {
            const uint32_t e = base + (uint32_t)(b * kScatBatch + r) * kScatThreads + t;
            if (rect[r] != 0u) {
                while (pos[r] + 1u < buckets && s_key[pos[r]] == key[r] && (uint32_t)splitters[pos[r]] <= e) ++pos[r];
            }
            where[b * kScatBatch + r] = rect[r] != 0u ? ((pos[r] << 15) | atomicAdd(&s_cnt[pos[r]], 1u)) : 0xFFFFFFFFu;
        }
        if (b + 1 < kScatMaxRows / kScatBatch) {
#pragma unroll
            for (int r = 0; r < kScatBatch; ++r) { key[r] = key_n[r]; rect[r] = rect_n[r]; }
        }
    }
    __syncthreads();
    // staging slots: exclusive prefix of the chunk's counts over the buckets (thread t: buckets 4 t .. 4 t + 3); region slots:
    // one global atomic per bucket with an entry
    uint32_t total;
    {
        uint32_t c[4], g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = s_cnt[4u * t + (uint32_t)k];
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = c[k] ? atomicAdd(&counts[4u * t + (uint32_t)k], c[k]) : 0u;
        const uint32_t sum = (c[0] + c[1]) + (c[2] + c[3]);
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += o;
        }
        if (lane == kWave - 1) s_ws[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < kScatThreads / kWave; ++w) {
            const uint32_t v = s_ws[w];
            wbase += (w < wave) ? v : 0u;
            tot += v;
        }
        total = tot;
        uint32_t first = wbase + incl - sum;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s_first[4u * t + (uint32_t)k] = first;
            s_cnt[4u * t + (uint32_t)k] = g[k] - first;          // (u32 wrap-around: region slot = staging slot + this)
            first += c[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kScatMaxRows; ++r) {
        if (where[r] != 0xFFFFFFFFu) {
            const uint32_t b = where[r] >> 15;
            s_stage[s_first[b] + (where[r] & 0x7FFFu)] = ((uint32_t)r * kScatThreads + t) | (b << 15);
        }
    }
    __syncthreads();
    bool over = false;
    for (uint32_t q = t; q < total; q += kScatThreads) {
        const uint32_t st = s_stage[q];
        const uint32_t b = st >> 15, e = base + (st & 0x7FFFu);
        const uint32_t at = q + s_cnt[b];
        if (at < bucket_cap) regions[(size_t)b * kDepthBucketCap + at] = make_uint4(depth_key[e], e, rect_by_index[e], 0u);   // (second read: L2)
        else over = true;
    }
    if (over) { words[1] = 1u; *reinterpret_cast<volatile uint32_t*>(host_fail) = 1u; }
}

// ---- sort of one bucket ------------------------------------------------------------------------------------------
// One workgroup per bucket. A bucket of up to kBsLds entries (twice the mean) is sorted in LDS in one go: 8-bit LSD passes
// over the composite (key - smallest key of the run) << index bits | index, only as many as it has bits, the entries in
// registers between the passes. A larger one (one in 250; up to the region's 8192) is sorted run by run, the runs written
// back in place, and every entry then finds its rank by a binary search in each other run.
constexpr int kBsThreads = 512, kBsWaves = kBsThreads / kWave, kBsItems = 8, kBsLds = kBsThreads * kBsItems;   // 4096

struct BucketLds {
    uint32_t lo[kBsLds], hi[kBsLds], rect[kBsLds];
    uint16_t wave_hist[2][kBsWaves][256];
    uint32_t run_start[256];
    uint32_t tmp[kBsWaves];
    uint32_t kmin, kmax;
};

// Sorts the `count` (<= lds_cap <= kBsLds) entries at src by (key, index); leaves composite and rectangle in s.lo / s.hi /
// s.rect at their sorted positions and returns the smallest key (composite = (key - kmin) << idx_bits | index; s.hi is
// valid only if *wide).
__device__ __forceinline__ uint32_t sort_run(BucketLds& s, const uint4* __restrict__ src, uint32_t count, uint32_t idx_bits,
                                             bool* wide_out) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint32_t nit = (count + kBsThreads - 1) / kBsThreads;            // items per lane this run needs (1..8)
    const uint32_t span = nit * kWave;
    // current order: position = wave * span + item * 64 + lane; positions >= count are padding (composite ~0: last, always)
    uint32_t key[kBsItems], idx[kBsItems], rect[kBsItems];
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    if (threadIdx.x == 0) { s.kmin = 0xFFFFFFFFu; s.kmax = 0u; }
#pragma unroll
    for (int i = 0; i < kBsItems; ++i) {
        key[i] = idx[i] = 0xFFFFFFFFu; rect[i] = 0u;
        if ((uint32_t)i < nit) {
            const uint32_t p = (uint32_t)wave * span + (uint32_t)i * kWave + (uint32_t)lane;
            if (p < count) {
                const uint4 e = src[p];
                key[i] = e.x; idx[i] = e.y; rect[i] = e.z;
                kmin = min(kmin, e.x); kmax = max(kmax, e.x);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, kWave));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, kWave));
    }
    __syncthreads();
    if (lane == 0) { atomicMin(&s.kmin, kmin); atomicMax(&s.kmax, kmax); }
    for (int i = threadIdx.x; i < kBsWaves * 256; i += kBsThreads) (&s.wave_hist[0][0][0])[i] = 0;
    __syncthreads();
    kmin = s.kmin; kmax = s.kmax;
    const uint32_t key_bits = kmax > kmin ? 32u - (uint32_t)__clz((int)(kmax - kmin)) : 0u;
    const uint32_t total_bits = key_bits + idx_bits;
    const bool wide = total_bits > 32u;
    const uint32_t passes = (total_bits + 7u) / 8u;           // >= 1: the index alone has a bit
    u64 c[kBsItems];
#pragma unroll
    for (int i = 0; i < kBsItems; ++i) c[i] = idx[i] == 0xFFFFFFFFu ? ~0ull : (((u64)(key[i] - kmin) << idx_bits) | idx[i]);

    for (uint32_t pass = 0; pass < passes; ++pass) {
        const uint32_t shift = 8u * pass;
        uint16_t* hist = &s.wave_hist[pass & 1u][wave][0];
        uint32_t rd[kBsItems];
#pragma unroll
        for (int i = 0; i < kBsItems; ++i) {
            rd[i] = 0;
            if ((uint32_t)i < nit) {
                const uint32_t d = (uint32_t)(c[i] >> shift) & 255u;
                uint32_t peers_lo = ~0u, peers_hi = ~0u;
#pragma unroll
                for (int bit = 0; bit < 8; ++bit) {
                    const int m = __builtin_amdgcn_sbfe((int)d, bit, 1);                 // 0 or -1
                    const u64 bal = __ballot(m != 0);
                    peers_lo &= ~((uint32_t)bal ^ (uint32_t)m);
                    peers_hi &= ~((uint32_t)(bal >> 32) ^ (uint32_t)m);
                }
                const uint32_t below = __builtin_amdgcn_mbcnt_hi(peers_hi, __builtin_amdgcn_mbcnt_lo(peers_lo, 0u));
                const uint32_t prior = hist[d];
                if (below == 0) hist[d] = (uint16_t)(prior + (uint32_t)__popc(peers_lo) + (uint32_t)__popc(peers_hi));
                rd[i] = (d << 16) | (prior + below);
            }
        }
        __syncthreads();
        // per digit: exclusive offsets across the waves, then across the digits; the other copy of the counters is zeroed
        // for the next pass on the way
        uint32_t acc = 0;
        if (threadIdx.x < 256) {
#pragma unroll
            for (int wv = 0; wv < kBsWaves; ++wv) {
                const uint32_t v = s.wave_hist[pass & 1u][wv][threadIdx.x];
                s.wave_hist[pass & 1u][wv][threadIdx.x] = (uint16_t)acc;
                s.wave_hist[(pass & 1u) ^ 1u][wv][threadIdx.x] = 0;
                acc += v;
            }
        }
        uint32_t incl = acc;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += o;
        }
        if (lane == kWave - 1 && wave < 4) s.tmp[wave] = incl;
        __syncthreads();
        if (threadIdx.x < 256) {
            uint32_t wbase = 0;
            for (int wv = 0; wv < wave; ++wv) wbase += s.tmp[wv];
            s.run_start[threadIdx.x] = wbase + incl - acc;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kBsItems; ++i) {
            if ((uint32_t)i < nit) {
                const uint32_t d = rd[i] >> 16;
                const uint32_t slot = (rd[i] & 0xFFFFu) + hist[d] + s.run_start[d];
                s.lo[slot] = (uint32_t)c[i];
                if (wide) s.hi[slot] = (uint32_t)(c[i] >> 32);
                s.rect[slot] = rect[i];
            }
        }
        __syncthreads();
        if (pass + 1u < passes) {
#pragma unroll
            for (int i = 0; i < kBsItems; ++i) {
                if ((uint32_t)i < nit) {
                    const uint32_t p = (uint32_t)wave * span + (uint32_t)i * kWave + (uint32_t)lane;
                    c[i] = wide ? (((u64)s.hi[p] << 32) | s.lo[p]) : (u64)s.lo[p];
                    rect[i] = s.rect[p];
                }
            }
            // (no barrier here: the next pass writes lo / hi / rect two barriers further on)
        }
    }
    *wide_out = wide;
    return kmin;
}

__global__ __launch_bounds__(kBsThreads) void depth_bucket_sort_kernel(uint4* regions, const uint32_t* __restrict__ counts,
                                                                        const uint32_t* __restrict__ words, uint32_t idx_bits,
                                                                        uint32_t bucket_cap, uint32_t lds_cap,
                                                                        uint32_t* __restrict__ out_k, uint32_t* __restrict__ out_v,
                                                                        uint32_t* __restrict__ out_r) {
    __shared__ BucketLds s;
    const uint32_t b = blockIdx.x;
    if (words[1] != 0u || b >= words[0]) return;
    const uint32_t count = counts[b];
    if (count == 0u || count > bucket_cap) return;
    // where this bucket's piece starts: the entries of the buckets before it
    uint32_t before = 0;
    for (uint32_t j = threadIdx.x; j < b; j += kBsThreads) before += counts[j];
    before = block_sum(before, s.tmp);
    uint4* region = regions + (size_t)b * kDepthBucketCap;
    const u64 idx_mask = (1ull << idx_bits) - 1ull;
    const uint32_t runs = (count + lds_cap - 1u) / lds_cap;
    for (uint32_t run = 0; run < runs; ++run) {
        const uint32_t off = run * lds_cap, cnt = min(lds_cap, count - off);
        bool wide;
        const uint32_t kmin = sort_run(s, region + off, cnt, idx_bits, &wide);
        for (uint32_t p = threadIdx.x; p < cnt; p += kBsThreads) {
            const u64 v = wide ? (((u64)s.hi[p] << 32) | s.lo[p]) : (u64)s.lo[p];
            const uint32_t key = kmin + (uint32_t)(v >> idx_bits), idx = (uint32_t)(v & idx_mask);
            if (runs == 1u) {
                out_k[before + p] = key;
                out_v[before + p] = idx;
                out_r[before + p] = s.rect[p];
            } else {
                region[off + p] = make_uint4(key, idx, s.rect[p], 0u);     // (every thread has read its entries of this run)
            }
        }
        __syncthreads();
    }
    if (runs == 1u) return;
    // The runs, each in order, lie in the region. An entry's place among all of them: its place in its own run + the entries
    // of every other run below it — a binary search in LDS, run after run (searching the region itself: twelve dependent
    // round trips to memory per entry and run, 100 us for the one bucket in a hundred that comes here).
    __threadfence();
    __syncthreads();
    constexpr int kMaxMine = (int)(kDepthBucketCap / kBsThreads);      // entries per thread: 16
    uint32_t rank[kMaxMine];
#pragma unroll
    for (int k = 0; k < kMaxMine; ++k) {
        const uint32_t p = (uint32_t)k * kBsThreads + threadIdx.x;
        rank[k] = p % lds_cap;
    }
    for (uint32_t q = 0; q < runs; ++q) {
        const uint32_t off = q * lds_cap, cnt = min(lds_cap, count - off);
        for (uint32_t p = threadIdx.x; p < cnt; p += kBsThreads) {
            const uint4 e = region[off + p];
            s.hi[p] = e.x; s.lo[p] = e.y;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kMaxMine; ++k) {
            const uint32_t p = (uint32_t)k * kBsThreads + threadIdx.x;
            if (p < count && p / lds_cap != q) {
                const uint4 e = region[p];
                uint32_t lo = 0, hi = cnt;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    const bool less = s.hi[mid] < e.x || (s.hi[mid] == e.x && s.lo[mid] < e.y);
                    if (less) lo = mid + 1; else hi = mid;
                }
                rank[k] += lo;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < kMaxMine; ++k) {
        const uint32_t p = (uint32_t)k * kBsThreads + threadIdx.x;
        if (p < count) {
            const uint4 e = region[p];
            out_k[before + rank[k]] = e.x;
            out_v[before + rank[k]] = e.y;
            out_r[before + rank[k]] = e.z;
        }
    }
}

inline size_t align128(size_t v) { return (v + 127) / 128 * 128; }
inline uint32_t sample_groups(size_t n) { return (uint32_t)(((n + 4095) / 4096) * (4096 / kDepthSampleStride)); }
inline uint32_t max_buckets_for(size_t n) { return std::min<uint32_t>(kDepthMaxBuckets, sample_groups(n) / kDepthOversample + 1u); }

}  // namespace

bool depth_buckets_supported(size_t n) {
    // the splitter kernel ranks the samples of a coarse range by counting: beyond 32 K samples (8.4 M Gaussians) a range
    // holds too many of them
    return n >= 2 && sample_groups(n) <= 32768u;
}

size_t depth_buckets_cleared_bytes() { return 128 + align128(kDepthMaxBuckets * sizeof(uint32_t)); }

size_t depth_buckets_scratch_bytes(size_t n) {
    if (!depth_buckets_supported(n)) return depth_buckets_cleared_bytes();
    return depth_buckets_cleared_bytes() + align128(sample_groups(n) * sizeof(u64)) + align128(kDepthMaxBuckets * sizeof(u64)) +
           (size_t)max_buckets_for(n) * kDepthBucketCap * sizeof(uint4);
}

DepthBuckets carve_depth_buckets(char* base, size_t n) {
    DepthBuckets d;
    size_t off = 0;
    d.words = reinterpret_cast<uint32_t*>(base + off); off += 128;
    d.counts = reinterpret_cast<uint32_t*>(base + off); off += align128(kDepthMaxBuckets * sizeof(uint32_t));
    d.groups = sample_groups(n);
    d.max_buckets = max_buckets_for(n);
    d.samples = nullptr; d.splitters = nullptr; d.regions = nullptr;
    if (depth_buckets_supported(n)) {
        d.samples = reinterpret_cast<u64*>(base + off); off += align128(d.groups * sizeof(u64));
        d.splitters = reinterpret_cast<u64*>(base + off); off += align128(kDepthMaxBuckets * sizeof(u64));
        d.regions = base + off; off += (size_t)d.max_buckets * kDepthBucketCap * sizeof(uint4);
    }
    d.bytes = off;
    return d;
}

// Test knobs (environment, read once): GSR_DEBUG_BUCKET_CAP (entries a bucket's region takes before the frame falls back to
// the LSD passes; default and maximum kDepthBucketCap), GSR_DEBUG_BUCKET_LDS (entries the bucket sort takes in one run;
// default and maximum 4096) — small values drive small test frames through the fallback and through the run merge.
static uint32_t debug_knob(const char* name, uint32_t def) {
    const char* v = getenv(name);
    if (!v || !*v) return def;
    const long x = atol(v);
    return (x >= 1 && (uint32_t)x <= def) ? (uint32_t)x : def;
}

// visible_dev: the number of Gaussians with a tile (device word, written by the scan). host_fail: mapped host word, zero
// before the call; set if the frame has to take the LSD passes instead.
int launch_depth_bucket_scatter(const DepthBuckets& d, const uint32_t* depth_key, const uint32_t* rect_by_index, uint32_t n,
                                const uint32_t* visible_dev, uint32_t* host_fail, hipStream_t stream) {
    if (!d.samples || !rect_by_index) return GSR_ERR_INVALID_ARG;
    static const uint32_t bucket_cap = debug_knob("GSR_DEBUG_BUCKET_CAP", kDepthBucketCap);
    hipLaunchKernelGGL(depth_splitters_kernel, dim3(kSplitWGs), dim3(kSplitThreads), 0, stream,
                       reinterpret_cast<const u64*>(d.samples), d.groups, visible_dev, n, d.max_buckets,
                       reinterpret_cast<u64*>(d.splitters), d.words, host_fail);
    GSR_LAUNCH_CHECK("depth_splitters_kernel");
    // rows of 1024 Gaussians per workgroup: a workgroup per CU if that takes no more than 24 rows
    const uint32_t rows = std::min<uint32_t>(kScatMaxRows, std::max<uint32_t>(1u, (n + 256u * kScatThreads - 1u) / (256u * kScatThreads)));
    const uint32_t chunks = (n + rows * kScatThreads - 1) / (rows * kScatThreads);
    const size_t lds = (size_t)rows * kScatThreads * sizeof(uint32_t);
    GSR_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(depth_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kScatMaxRows * kScatThreads * (int)sizeof(uint32_t)));
    hipLaunchKernelGGL(depth_scatter_kernel, dim3(chunks), dim3(kScatThreads), lds, stream, depth_key, rect_by_index, n, rows,
                       reinterpret_cast<const u64*>(d.splitters), d.words, d.counts, reinterpret_cast<uint4*>(d.regions),
                       bucket_cap, host_fail);
    GSR_LAUNCH_CHECK("depth_scatter_kernel");
    return GSR_OK;
}

int launch_depth_bucket_sort(const DepthBuckets& d, uint32_t n, uint32_t* out_k, uint32_t* out_v, uint32_t* out_r,
                             hipStream_t stream) {
    if (!d.samples) return GSR_ERR_INVALID_ARG;
    static const uint32_t bucket_cap = debug_knob("GSR_DEBUG_BUCKET_CAP", kDepthBucketCap);
    static const uint32_t lds_cap = debug_knob("GSR_DEBUG_BUCKET_LDS", kBsLds);
    uint32_t idx_bits = 1;
    while (idx_bits < 32u && ((uint64_t)1 << idx_bits) < (uint64_t)n) ++idx_bits;
    hipLaunchKernelGGL(depth_bucket_sort_kernel, dim3(d.max_buckets), dim3(kBsThreads), 0, stream,
                       reinterpret_cast<uint4*>(d.regions), d.counts, d.words, idx_bits, bucket_cap, lds_cap, out_k, out_v, out_r);
    GSR_LAUNCH_CHECK("depth_bucket_sort_kernel");
    return GSR_OK;
}

}  // namespace gsr
