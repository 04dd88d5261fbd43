// micro-benchmark: achievable HBM write rate for different store shapes (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void wr(uint64_t* k, uint32_t* v, size_t n, int run) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    if (MODE == 0) { for (; i < n; i += stride) { k[i] = i; v[i] = (uint32_t)i; } }               // 8B + 4B per lane, sequential
    if (MODE == 1) { ulonglong2* k2 = (ulonglong2*)k; uint2* v2 = (uint2*)v; for (; i < n / 2; i += stride) { k2[i] = make_ulonglong2(i, i); v2[i] = make_uint2(i, i); } }
    if (MODE == 2) { // runs of `run` elements placed at scattered (hashed) run slots
        size_t nruns = n / run;
        for (; i < n; i += stride) { size_t r = i / run, o = i % run; size_t rr = (r * 2654435761ull) % nruns; k[rr * run + o] = i; v[rr * run + o] = (uint32_t)i; }
    }
    if (MODE == 4) { // unaligned runs: run r of length `run` starts at r*(run+3) (8-byte aligned only), lanes = consecutive elements
        for (; i < n; i += stride) { size_t r = i / run, o = i % run; size_t nr = n / (run + 3); size_t rr = (r * 2654435761ull) % nr; k[rr * (run + 3) + o] = i; v[rr * (run + 3) + o] = (uint32_t)i; }
    }
    if (MODE == 5) { // like 4 but WITHOUT gaps: run r starts at r*run, runs visited in hashed order, element o of a run by lane o
        size_t nr = n / run;
        for (; i < n; i += stride) { size_t r = i / run, o = i % run; size_t rr = (r * 2654435761ull) % nr; k[rr * run + o] = i; v[rr * run + o] = (uint32_t)i; }
    }
    if (MODE == 3) { for (; i < n; i += stride) { k[i] = i; } }                                          // keys only
}
int main() {
    size_t n = 268435456; uint64_t* k; uint32_t* v;
    hipMalloc(&k, n * 8); hipMalloc(&v, n * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, auto launch, double bytes) {
        launch(); hipDeviceSynchronize(); hipEventRecord(a); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5; printf("%-40s %.3f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6);
    };
    for (int blocks : {8192}) {
        printf("blocks=%d\n", blocks);
        run("8B+4B per lane sequential", [&] { wr<0><<<blocks, 256>>>(k, v, n, 0); }, n * 12.0);
        run("16B+8B per lane sequential", [&] { wr<1><<<blocks, 256>>>(k, v, n, 0); }, n * 12.0);
        run("keys only 8B sequential", [&] { wr<3><<<blocks, 256>>>(k, v, n, 0); }, n * 8.0);
        for (int r : {8, 32, 128, 512}) { char nm[64]; snprintf(nm, 64, "scattered runs of %d", r); run(nm, [&] { wr<2><<<blocks, 256>>>(k, v, n, r); }, n * 12.0); }
        for (int r : {5, 13, 29, 61, 125}) { char nm[64]; snprintf(nm, 64, "UNALIGNED scattered runs of %d", r); run(nm, [&] { wr<4><<<blocks, 256>>>(k, v, n, r); }, n * 12.0); }
        for (int r : {5, 13, 29, 61, 125}) { char nm[64]; snprintf(nm, 64, "gapless odd runs of %d (hashed order)", r); run(nm, [&] { wr<5><<<blocks, 256>>>(k, v, n, r); }, n * 12.0); }
    }
    return 0;
}
