// Where do the waves of co-resident 8-wave blocks land? (round 5, E7.8) One record per wave: XCC, SE, CU, SIMD, threadgroup slot.
// hipcc --offload-arch=gfx950 -O2 -o simd_placement_probe tools/simd_placement_probe.hip && ./simd_placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void __launch_bounds__(512) probe(unsigned* out, int spin, int waves)
{
    extern __shared__ unsigned char lds[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);        // HW_REG_HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);       // HW_REG_XCC_ID
    volatile unsigned* l = (volatile unsigned*)lds;
    unsigned acc = 0;
    for (int i = 0; i < spin; ++i) acc += l[(threadIdx.x + i) & 1023];               // stay resident for a while
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * waves + (threadIdx.x >> 6);
        out[2 * w] = hw;
        out[2 * w + 1] = xcc + (acc & 0u);
    }
}
int main(int argc, char** argv)
{
    // usage: simd_placement_probe [blocks [waves per block [LDS bytes per block]]]
    const int blocks = argc > 1 ? atoi(argv[1]) : 768;
    const int waves = argc > 2 ? atoi(argv[2]) : 8;
    const int lds = argc > 3 ? atoi(argv[3]) : 51336;
    unsigned* d;
    hipMalloc(&d, blocks * 8 * 2 * sizeof(unsigned));
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(64 * waves), lds, 0, d, 20000, waves);
    std::vector<unsigned> h(blocks * 8 * 2);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    // per CU (xcc, se, cu): which SIMD holds the last wave of each resident block
    std::map<unsigned, std::vector<int>> simd_of_w7;
    int hist[8][4] = {};
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) {
            const unsigned hw = h[2 * (b * waves + w)], xcc = h[2 * (b * waves + w) + 1] & 15;
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            hist[w][simd]++;
            if (w == waves - 1) simd_of_w7[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back((int)simd);
        }
    for (int w = 0; w < waves; ++w) printf("wave %d of a block: SIMD 0/1/2/3 = %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    int cus = 0, same = 0, counts[8] = {};
    for (auto& kv : simd_of_w7) {
        ++cus;
        counts[kv.second.size() < 7 ? kv.second.size() : 7]++;   // (blocks that ran AFTER others left show up as extra entries: use blocks = what is resident)
        bool all_same = true;
        for (int s : kv.second) all_same = all_same && s == kv.second[0];
        if (kv.second.size() > 1 && all_same) ++same;
    }
    printf("compute units seen: %d; blocks per CU histogram (1..6):", cus);
    for (int i = 1; i < 7; ++i) printf(" %d", counts[i]);
    printf("\nCUs whose resident blocks ALL have the last wave on the same SIMD: %d\n", same);
    int shown = 0;
    for (auto& kv : simd_of_w7) {
        if (shown++ >= 6) break;
        printf("  cu key %05x: SIMD of the last wave per resident block:", kv.first);
        for (int s : kv.second) printf(" %d", s);
        printf("\n");
    }
    return 0;
}
