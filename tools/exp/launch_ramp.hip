// launch_ramp.hip -- experiment: how long after the first wave of a grid does the LAST workgroup start, as a function of
// workgroup size, dynamic LDS and register budget?  (tools/exp, not shipped.)  Every wave stamps wall_clock64() on entry.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/launch_ramp.hip -o tools/exp/launch_ramp.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__device__ long long g_t[8192 * 16];

template <int THREADS, int MINB>
__global__ __launch_bounds__(THREADS, MINB) void stamp(int spin) {
    extern __shared__ char lds[];
    const long long t = wall_clock64();
    if ((threadIdx.x & 63) == 0) g_t[blockIdx.x * 16 + (threadIdx.x >> 6)] = t;
    // stay resident for a while so that later workgroups cannot reuse this slot
    while (wall_clock64() - t < spin) { if (lds[0] == 77 && threadIdx.x == 9999) g_t[0] = 0; }
}

template <int THREADS, int MINB>
void run(const char* name, int grid, int ldsb, int spin) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(stamp<THREADS, MINB>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    std::vector<long long> h(8192 * 16, 0);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_t), h.data(), h.size() * 8);
        hipDeviceSynchronize();
        stamp<THREADS, MINB><<<grid, THREADS, ldsb>>>(spin);
        hipDeviceSynchronize();
    }
    std::vector<long long> t(8192 * 16);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_t), t.size() * 8);
    std::vector<long long> v;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < THREADS / 64; ++w) if (t[b * 16 + w]) v.push_back(t[b * 16 + w]);
    std::sort(v.begin(), v.end());
    double mean = 0; for (auto x : v) mean += x - v[0];
    printf("%-34s grid %5d lds %6d: waves %5zu  mean %6.2f us  median %6.2f  90%% %6.2f  last %6.2f\n", name, grid, ldsb, v.size(),
           mean / v.size() / 100.0, (v[v.size() / 2] - v[0]) / 100.0, (v[v.size() * 9 / 10] - v[0]) / 100.0, (v.back() - v[0]) / 100.0);
}

int main() {
    const int spin = 3000;      // 30 us resident
    run<512, 1>("512 thr, bounds(512,1)", 256, 0, spin);
    run<512, 1>("512 thr, bounds(512,1)", 256, 86528, spin);
    run<512, 1>("512 thr, bounds(512,1)", 256, 160 * 1024, spin);
    run<256, 1>("256 thr", 256, 0, spin);
    run<256, 1>("256 thr", 512, 0, spin);
    run<256, 1>("256 thr", 512, 66 * 1024, spin);
    run<256, 1>("256 thr", 1024, 0, spin);
    run<64, 1>("64 thr", 2048, 0, spin);
    run<64, 1>("64 thr", 1024, 0, spin);
    run<1024, 1>("1024 thr", 256, 0, spin);
    return 0;
}
