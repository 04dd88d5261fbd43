// Does it matter which XCD writes the two halves of a shared cache line? Tiles write one run per
// digit region (radix-scatter shape); run boundaries fall inside cache lines. map 0: tile = block
// (adjacent tiles on different XCDs); map 1: 8 consecutive tiles share blockIdx % 8 (same XCD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
__global__ void scatter(uint64_t* k, uint32_t* v, const uint32_t* off, int D, uint32_t tiles, int map) {
    uint32_t b = blockIdx.x, t = b;
    if (map == 1) { uint32_t x = b % 8, j = (b / 8) % 8, base = b / 64; t = base * 64 + x * 8 + j; }
    if (t >= tiles) return;
    for (int d = 0; d < D; ++d) {
        const uint32_t s = off[(size_t)d * (tiles + 1) + t], e = off[(size_t)d * (tiles + 1) + t + 1];
        for (uint32_t i = s + threadIdx.x; i < e; i += blockDim.x) { k[i] = i; v[i] = i; }
    }
}
int main() {
    const int D = 68; const uint32_t tiles = (32768u * 120u / MEAN) / 64 * 64; const uint32_t mean = MEAN;
    std::mt19937 rng(1); std::vector<uint32_t> off((size_t)D * (tiles + 1));
    uint64_t pos = 0;
    for (int d = 0; d < D; ++d) { for (uint32_t t = 0; t <= tiles; ++t) { off[(size_t)d * (tiles + 1) + t] = (uint32_t)pos; if (t < tiles) pos += mean / 2 + rng() % mean; } }
    uint64_t n = pos; printf("n = %llu keys\n", (unsigned long long)n);
    uint64_t* k; uint32_t* v; uint32_t* doff;
    hipMalloc(&k, n * 8); hipMalloc(&v, n * 4); hipMalloc(&doff, off.size() * 4);
    hipMemcpy(doff, off.data(), off.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int map = 0; map < 2; ++map) for (int rep = 0; rep < 2; ++rep) {
        scatter<<<tiles, 512>>>(k, v, doff, D, tiles, map); hipDeviceSynchronize();
        hipEventRecord(a); for (int i = 0; i < 5; ++i) scatter<<<tiles, 512>>>(k, v, doff, D, tiles, map); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("map %d: %.3f ms  %.0f GB/s\n", map, ms, n * 12.0 / ms / 1e6);
    }
    return 0;
}
