// micro-benchmark (not part of the product): the write rate of block_emit_kernel's store shape alone — every wave writes
// 128-key groups (16 B of keys + 8 B of values per lane, streaming stores) at scattered group positions — as a function
// of the number of waves per CU, the store policy and the bytes per lane. hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int NT, int WIDE, int RUN>
__global__ __launch_bounds__(256) void wr(uint64_t* k, uint32_t* v, size_t groups) {
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) / 64, waves = (size_t)gridDim.x * 4;
    const uint32_t lane = threadIdx.x & 63;
    // a wave writes RUN consecutive groups (one (unit, tile) run), then jumps
    for (size_t r = wave; r * RUN < groups; r += waves) {
        const size_t run = (r * 2654435761ull) % (groups / RUN);
        for (int j = 0; j < RUN; ++j) {
            const size_t a = (run * RUN + j) * 128;
            if (WIDE) {
                const u32x4 kk = {(uint32_t)a, lane, (uint32_t)a + 1, lane};
                const u32x2 vv = {lane, lane + 1};
                if (NT) { __builtin_nontemporal_store(kk, (u32x4*)(k + a + 2 * lane)); __builtin_nontemporal_store(vv, (u32x2*)(v + a + 2 * lane)); }
                else { *(u32x4*)(k + a + 2 * lane) = kk; *(u32x2*)(v + a + 2 * lane) = vv; }
            } else {
                for (int h = 0; h < 2; ++h) {
                    const u32x2 kk = {(uint32_t)a, lane};
                    if (NT) { __builtin_nontemporal_store(kk, (u32x2*)(k + a + 64 * h + lane)); __builtin_nontemporal_store(lane, v + a + 64 * h + lane); }
                    else { *(u32x2*)(k + a + 64 * h + lane) = kk; v[a + 64 * h + lane] = lane; }
                }
            }
        }
    }
}
int main() {
    size_t n = 267476934 / 128 * 128; uint64_t* k; uint32_t* v;
    hipMalloc(&k, n * 8); hipMalloc(&v, n * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize(); hipEventRecord(a); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5; printf("%-64s %.3f ms  %.0f GB/s\n", name, ms, n * 12.0 / ms / 1e6);
    };
    const size_t groups = n / 128;
    for (int wgs : {512, 1024, 2048, 8192}) {
        char nm[96];
        snprintf(nm, 96, "%5d WGs x 4 waves, nt, 16+8 B/lane, runs of 11 groups", wgs); run(nm, [&] { wr<1, 1, 11><<<wgs, 256>>>(k, v, groups); });
        snprintf(nm, 96, "%5d WGs x 4 waves, nt, 16+8 B/lane, single groups", wgs);     run(nm, [&] { wr<1, 1, 1><<<wgs, 256>>>(k, v, groups); });
        snprintf(nm, 96, "%5d WGs x 4 waves, plain, 16+8 B/lane, runs of 11", wgs);     run(nm, [&] { wr<0, 1, 11><<<wgs, 256>>>(k, v, groups); });
        snprintf(nm, 96, "%5d WGs x 4 waves, nt, 8+4 B/lane x2, runs of 11", wgs);      run(nm, [&] { wr<1, 0, 11><<<wgs, 256>>>(k, v, groups); });
        snprintf(nm, 96, "%5d WGs x 4 waves, plain, 8+4 B/lane x2, runs of 11", wgs);   run(nm, [&] { wr<0, 0, 11><<<wgs, 256>>>(k, v, groups); });
    }
    return 0;
}
