// Micro-benchmark (not part of the product): what does reading the DC triple — 12 bytes at the head of every 192-byte SH
// record — cost the memory? The same records read as 12 / 16 / 64 / 128 / 192 bytes each, one lane per record with U
// records in flight per lane, against the shape colors_visible_kernel had in round 4 (one lane per FLOAT, one load in flight).
// If 12 bytes cost what 64 cost and half of what 128 cost, the memory moves 64-byte sectors and the kernel is request-bound.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/gather_dc.hip -o gather_dc && ./gather_dc [records]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void per_float(size_t n, const float* __restrict__ shs, float* __restrict__ out) {
    const size_t f = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= 3 * n) return;
    const size_t g = f / 3;
    __builtin_nontemporal_store(0.5f + 0.4f * __builtin_nontemporal_load(shs + 48 * g + (f - 3 * g)), out + f);
}
// one lane per record, U records in flight per lane (records of one lane are 64 apart: a wave's loads of one round cover 64 consecutive records)
template <int U, bool NT>
__global__ __launch_bounds__(256) void per_record(size_t n, const float* __restrict__ shs, float* __restrict__ out) {
    const size_t base = ((size_t)blockIdx.x * 256 + (threadIdx.x & ~63u)) * U + (threadIdx.x & 63u);
    f3 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t g = base + (size_t)u * 64;
        if (g < n) v[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f3*>(shs + 48 * g)) : *reinterpret_cast<const f3*>(shs + 48 * g);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t g = base + (size_t)u * 64;
        if (g < n) { f3 o = 0.5f + 0.4f * v[u]; __builtin_nontemporal_store(o, reinterpret_cast<f3*>(out + 3 * g)); }
    }
}
// LANES lanes x 16 bytes of every record (LANES = 1, 4, 8, 12): the first LANES * 16 bytes
template <int LANES>
__global__ __launch_bounds__(256) void per_piece(size_t n, const float* __restrict__ shs, float* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t g = t / LANES, q = t % LANES;
    if (g >= n) return;
    const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(shs + 48 * g + 4 * q));
    if (q == 0) { f3 o = {0.5f + 0.4f * v.x, 0.5f + 0.4f * v.y, 0.5f + 0.4f * v.z}; __builtin_nontemporal_store(o, reinterpret_cast<f3*>(out + 3 * g)); }
    else if (v.x == 123456.0f) out[0] = v.y;      // (keeps the load)
}
int main(int argc, char** argv) {
    const size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 50000000ull;
    float *shs, *out;
    if (hipMalloc(&shs, n * 192) != hipSuccess || hipMalloc(&out, n * 12) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(shs, 0, n * 192);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(a); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-46s %8.3f ms  %6.1f Grecords/s  as 64-B sectors %5.0f GB/s, as 128-B lines %5.0f GB/s (+ %4.0f GB/s written)\n", name, ms,
               n / ms / 1e6, n * 64.0 / ms / 1e6, n * 128.0 / ms / 1e6, n * 12.0 / ms / 1e6);
    };
    printf("records = %zu (%.2f GB of SH array)\n", n, n * 192.0 / 1e9);
    run("lane per float, 1 load in flight (round 4)", [&] { per_float<<<(unsigned)((3 * n + 255) / 256), 256>>>(n, shs, out); });
#define PR(U) run("lane per record, dwordx3 nt, " #U " in flight", [&] { per_record<U, true><<<(unsigned)((n + 256 * U - 1) / (256 * U)), 256>>>(n, shs, out); })
    PR(1); PR(2); PR(4); PR(8);
    run("lane per record, dwordx3 plain, 4 in flight", [&] { per_record<4, false><<<(unsigned)((n + 1023) / 1024), 256>>>(n, shs, out); });
    run("16 B of every record", [&] { per_piece<1><<<(unsigned)((n + 255) / 256), 256>>>(n, shs, out); });
    run("64 B of every record (4 lanes x 16 B)", [&] { per_piece<4><<<(unsigned)((4 * n + 255) / 256), 256>>>(n, shs, out); });
    run("128 B of every record (8 lanes x 16 B)", [&] { per_piece<8><<<(unsigned)((8 * n + 255) / 256), 256>>>(n, shs, out); });
    run("192 B of every record (12 lanes x 16 B)", [&] { per_piece<12><<<(unsigned)((12 * n + 255) / 256), 256>>>(n, shs, out); });
    return 0;
}
