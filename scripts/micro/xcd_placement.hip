// Diagnostic (not part of the product): which XCD does workgroup b run on, and in what order do workgroups start?
// The onesweep passes deal tiles by residue class blockIdx % 8 on the observation that blocks b and b + 8 share an XCD.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/xcd_placement.hip -o xcd_placement && ./xcd_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ __launch_bounds__(512) void probe(uint32_t* xcc, uint32_t* order, uint32_t* ticket, int spin) {
    __shared__ uint32_t lds[14000];                 // ~56 KB: two workgroups per CU, as the sort pass
    uint32_t id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(id));
    if (threadIdx.x == 0) { xcc[blockIdx.x] = id; order[blockIdx.x] = atomicAdd(ticket, 1u); }
    uint32_t v = threadIdx.x;
    for (int i = 0; i < spin; ++i) { lds[(v * 7 + i) % 14000] = v; v = v * 1664525u + 1013904223u + lds[(v >> 8) % 14000]; }
    if (v == 0xdeadbeef) xcc[0] = v;
}
int main() {
    const int blocks = 5630;
    uint32_t *xcc, *order, *ticket;
    hipMalloc(&xcc, blocks * 4); hipMalloc(&order, blocks * 4); hipMalloc(&ticket, 4);
    for (int spin : {200, 4000}) {
        hipMemset(ticket, 0, 4);
        probe<<<blocks, 512>>>(xcc, order, ticket, spin); hipDeviceSynchronize();
        std::vector<uint32_t> hx(blocks), ho(blocks);
        hipMemcpy(hx.data(), xcc, blocks * 4, hipMemcpyDeviceToHost); hipMemcpy(ho.data(), order, blocks * 4, hipMemcpyDeviceToHost);
        int same8 = 0, rr = 0; long disp = 0, maxdisp = 0; int cnt[8] = {0};
        for (int b = 0; b < blocks; ++b) {
            cnt[hx[b] & 7]++;
            if (b >= 8 && hx[b] == hx[b - 8]) ++same8;
            if (b >= 1 && hx[b] == ((hx[b - 1] + 1) & 7)) ++rr;
            long d = (long)ho[b] - b; if (d < 0) d = -d; disp += d; if (d > maxdisp) maxdisp = d;
        }
        printf("spin %d: xcc(b) == xcc(b-8) for %d of %d; xcc(b) == xcc(b-1)+1 for %d; first 16 xcc:", spin, same8, blocks - 8, rr);
        for (int b = 0; b < 16; ++b) printf(" %u", hx[b]);
        printf("\n  per XCD:"); for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
        printf("\n  start order vs blockIdx: mean |ticket - b| = %.1f, max %ld\n", (double)disp / blocks, maxdisp);
        // within a residue class: is start order = block order?
        int inv = 0, tot = 0;
        for (int r = 0; r < 8; ++r) for (int b = r + 8; b < blocks; b += 8) { ++tot; if (ho[b] < ho[b - 8]) ++inv; }
        printf("  within a residue class, block b started before block b - 8: %d of %d\n", inv, tot);
    }
    return 0;
}
