// Micro-benchmark (not part of the product): the WRITE side of one 8-bit onesweep pass over n (key, index, rectangle)
// records, for different record layouts and tile sizes. A workgroup owns a tile of T records already in digit order
// (run d of the tile = L[t][d] records, random around T / 256) and writes run d at off[d][t] — adjacent to the run of tile
// t - 1 for the same digit, exactly as the radix scatter does. Layouts: three u32 streams (round 4), one 12-byte record,
// one 16-byte record, (u64, u32). Optionally the same bytes are first READ coalesced (a pass reads what it writes).
// hipcc --offload-arch=gfx950 -O3 scripts/micro/scatter_records.hip -o scatter_records && ./scatter_records [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <random>
typedef uint32_t u3 __attribute__((ext_vector_type(3)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

// prefix: [tile][257] positions of the runs inside the tile; off: [256][tiles] global start of run d of tile t
template <int LAYOUT, int ITEMS, bool READ>
__global__ __launch_bounds__(512) void scatter(const uint32_t* __restrict__ prefix, const uint32_t* __restrict__ off, uint32_t tiles, int map,
                                               const uint32_t* __restrict__ in, uint32_t* __restrict__ a, uint32_t* __restrict__ b, uint32_t* __restrict__ c) {
    __shared__ uint32_t s_pre[257], s_delta[256];
    uint32_t t = blockIdx.x;
    if (map == 1) { const uint32_t x = t % 8, j = (t / 8) % 8, base = t / 64 * 64; if (base + 64 <= tiles) t = base + x * 8 + j; }
    if (threadIdx.x <= 256) s_pre[threadIdx.x] = prefix[(size_t)t * 257 + threadIdx.x];
    __syncthreads();
    if (threadIdx.x < 256) s_delta[threadIdx.x] = off[(size_t)threadIdx.x * tiles + t] - s_pre[threadIdx.x];
    constexpr uint32_t T = 512 * ITEMS;
    const uint32_t valid = prefix[(size_t)t * 257 + 256];
    uint32_t acc = 0;
    if (READ) {
        // the tile's records as a pass reads them: three coalesced dword streams (round-4 layout) from `in`
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const size_t e = (size_t)t * T + (size_t)i * 512 + threadIdx.x;
            acc += in[e] + in[e + (size_t)tiles * T] + in[e + 2 * (size_t)tiles * T];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t p = (uint32_t)i * 512 + threadIdx.x;
        if (p >= valid) continue;
        uint32_t d = 0;
#pragma unroll
        for (uint32_t step = 128; step >= 1; step >>= 1) if (s_pre[d + step] <= p) d += step;
        const size_t dst = (size_t)p + s_delta[d];
        const uint32_t k = p + acc, v = t, r = d;
        if (LAYOUT == 0) { a[dst] = k; b[dst] = v; c[dst] = r; }
        if (LAYOUT == 1) { *reinterpret_cast<u3*>(a + 3 * dst) = (u3){k, v, r}; }
        if (LAYOUT == 2) { *reinterpret_cast<u4*>(a + 4 * dst) = (u4){k, v, r, 0u}; }
        if (LAYOUT == 3) { *reinterpret_cast<u2*>(a + 2 * dst) = (u2){k, v}; c[dst] = r; }
    }
}
int main(int argc, char** argv) {
    const size_t n_want = argc > 1 ? strtoull(argv[1], 0, 10) : 46000000ull;
    hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
    for (int items : {16, 32}) {
        const uint32_t T = 512u * items, tiles = (uint32_t)(n_want / T) / 64 * 64;
        std::mt19937 rng(7);
        std::vector<uint32_t> prefix((size_t)tiles * 257), off((size_t)256 * tiles);
        std::vector<uint32_t> len((size_t)tiles * 256);
        // run lengths: multinomial-ish — every record of the tile draws a digit
        for (uint32_t t = 0; t < tiles; ++t) {
            uint32_t* L = &len[(size_t)t * 256];
            for (uint32_t i = 0; i < T; ++i) L[rng() & 255]++;
            uint32_t p = 0;
            for (int d = 0; d < 256; ++d) { prefix[(size_t)t * 257 + d] = p; p += L[d]; }
            prefix[(size_t)t * 257 + 256] = p;
        }
        size_t pos = 0;
        for (int d = 0; d < 256; ++d) for (uint32_t t = 0; t < tiles; ++t) { off[(size_t)d * tiles + t] = (uint32_t)pos; pos += len[(size_t)t * 256 + d]; }
        const size_t n = pos;
        uint32_t *dpre, *doff, *in, *a, *b, *c;
        hipMalloc(&dpre, prefix.size() * 4); hipMalloc(&doff, off.size() * 4);
        hipMalloc(&in, n * 12 + 4096); hipMalloc(&a, n * 16 + 4096); hipMalloc(&b, n * 4 + 4096); hipMalloc(&c, n * 4 + 4096);
        hipMemset(in, 1, n * 12);
        hipMemcpy(dpre, prefix.data(), prefix.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(doff, off.data(), off.size() * 4, hipMemcpyHostToDevice);
        printf("tile = %u records, %u tiles, n = %zu, mean run %u records\n", T, tiles, n, T / 256);
        auto run = [&](const char* name, double bytes, auto launch) {
            for (int map = 0; map < 2; ++map) {
                launch(map); hipDeviceSynchronize();
                hipEventRecord(ea); for (int i = 0; i < 5; ++i) launch(map); hipEventRecord(eb); hipEventSynchronize(eb);
                float ms; hipEventElapsedTime(&ms, ea, eb); ms /= 5;
                printf("  %-44s map %d  %7.3f ms  %5.0f GB/s\n", name, map, ms, bytes / ms / 1e6);
            }
        };
#define RUN(L, R, NAME, BYTES) \
        if (items == 16) run(NAME, BYTES, [&](int map) { scatter<L, 16, R><<<tiles, 512>>>(dpre, doff, tiles, map, in, a, b, c); }); \
        else run(NAME, BYTES, [&](int map) { scatter<L, 32, R><<<tiles, 512>>>(dpre, doff, tiles, map, in, a, b, c); });
        RUN(0, false, "write only: three u32 streams", n * 12.0)
        RUN(3, false, "write only: (u64 pair) + u32", n * 12.0)
        RUN(1, false, "write only: 12-byte records", n * 12.0)
        RUN(2, false, "write only: 16-byte records", n * 16.0)
        RUN(0, true, "read + write: three u32 streams", n * 24.0)
        RUN(1, true, "read + write: 12-byte records", n * 24.0)
        RUN(2, true, "read + write: 16-byte records (12 read)", n * 28.0)
        hipFree(dpre); hipFree(doff); hipFree(in); hipFree(a); hipFree(b); hipFree(c);
    }
    return 0;
}
