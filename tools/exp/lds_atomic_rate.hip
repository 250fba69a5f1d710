// LDS atomic throughput on gfx950: float add vs 32-bit / 64-bit integer add vs plain read-modify-write, random cells of a
// 76x76 plane (the ROI pooling / ROIAlign backward pattern).  hipcc --offload-arch=gfx950 -O3 lds_atomic_rate.hip -o lds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(1024) void k(const int* __restrict__ idx, int n, int reps, float* out) {
    __shared__ long long cells[5776];
    float* f = reinterpret_cast<float*>(cells);
    unsigned* u = reinterpret_cast<unsigned*>(cells);
    for (int i = threadIdx.x; i < 5776; i += blockDim.x) cells[i] = 0;
    __syncthreads();
    for (int r = 0; r < reps; ++r)
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int a = idx[i];
            if (MODE == 0) atomicAdd(&f[a], 1.0f);
            else if (MODE == 1) atomicAdd(&u[a], 3u);
            else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long*>(&cells[a]), 3ull);
            else f[a] += 1.0f;                                  // racy read-modify-write: the cost of NOT being atomic
        }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = f[0] + (float)u[1];
}

int main() {
    const int n = 98000, reps = 8;
    int* h = (int*)malloc(n * sizeof(int));
    for (int pat = 0; pat < 2; ++pat) {
        for (int i = 0; i < n; ++i) h[i] = pat == 0 ? rand() % 5776 : ((i / 49) * 37 + (i % 7) * 4 + ((i % 49) / 7) * 76 * 4) % 5776;
        int* d; float* o;
        hipMalloc(&d, n * sizeof(int)); hipMalloc(&o, 4096);
        hipMemcpy(d, h, n * sizeof(int), hipMemcpyHostToDevice);
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
            for (int it = 0; it < 2; ++it) {
                hipEventRecord(s);
                if (mode == 0) k<0><<<512, 1024>>>(d, n, reps, o);
                if (mode == 1) k<1><<<512, 1024>>>(d, n, reps, o);
                if (mode == 2) k<2><<<512, 1024>>>(d, n, reps, o);
                if (mode == 3) k<3><<<512, 1024>>>(d, n, reps, o);
                hipEventRecord(e); hipEventSynchronize(e);
            }
            float ms; hipEventElapsedTime(&ms, s, e);
            const char* nm[4] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "plain rmw"};
            printf("pattern %d %-12s %8.3f ms  %7.3f T updates/s chip-wide\n", pat, nm[mode], ms, 512.0 * n * reps / ms / 1e9);
        }
    }
    return 0;
}
