// store_pattern.hip -- experiment: how fast can a P x P fp32 matrix be WRITTEN when the stores arrive in the shapes a
// tiled E E^T kernel produces?  (tools/exp, not part of the library.)  Durations are read from a rocprofv3 kernel trace:
//   hipcc --offload-arch=gfx950 -O3 tools/exp/store_pattern.hip -o tools/exp/store_pattern.bin
//   rocprofv3 --kernel-trace --stats -- tools/exp/store_pattern.bin 4000
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void fill_f4(float* __restrict__ S, size_t n4) {
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<float4*>(S)[i] = v;
}

// upper-triangular 32 x 32 tiles, each written twice (direct: 16 x 4-byte stores = two 128-byte row segments per
// instruction; mirror: 4 x float4 = eight 128-byte row segments per instruction).  WAVES waves per workgroup, tiles dealt
// so that the waves of a workgroup own adjacent row bands and walk the column blocks together (the panel kernel's order).
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void tri_mirror(float* __restrict__ S, int P) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int nb = P / 32, npanel = (nb + WAVES - 1) / WAVES;
    long long items = 0;
    for (int p = 0; p < npanel; ++p) items += nb - p * WAVES;
    const long long lo = items * blockIdx.x / gridDim.x, hi = items * (blockIdx.x + 1) / gridDim.x;
    int panel = 0; long long rest = lo;
    while (panel < npanel && rest >= nb - panel * WAVES) { rest -= nb - panel * WAVES; ++panel; }
    int blk = panel * WAVES + (int)rest;
    for (long long it = lo; it < hi; ++it) {
        const int r0 = (panel * WAVES + wave) * 32, c0 = blk * 32;
        if (c0 >= r0 && r0 < P) {
            const unsigned dbase = (unsigned)(r0 + 4 * half) * (unsigned)P + (unsigned)(c0 + l31);
#pragma unroll
            for (int k = 0; k < 16; ++k) S[dbase + (unsigned)((k & 3) + 8 * (k >> 2)) * (unsigned)P] = (float)k;
            if (c0 != r0) {
                const unsigned mbase = (unsigned)(c0 + (lane >> 3)) * (unsigned)P + (unsigned)(r0 + (lane & 7) * 4);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    *reinterpret_cast<float4*>(S + (mbase + (unsigned)(8 * t) * (unsigned)P)) = make_float4(1.f, 2.f, 3.f, (float)t);
            }
        }
        if (++blk >= nb) { ++panel; blk = panel * WAVES; }
    }
}

// the FULL matrix, no mirror: a wave owns a band of 32 rows and a run of W-column pieces of it; each store instruction
// is a float4 per lane covering (256 / W) rows x 4 W bytes.  W = 32: 8 rows x 128 B ... W = 256: 1 row x 1 KB.
template <int W>
__global__ __launch_bounds__(512) void full_rows(float* __restrict__ S, int P, int chunk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nband = P / 32, npiece = P / W;                 // P a multiple of 256 here
    const long long items = (long long)nband * npiece / chunk; // an item = `chunk` consecutive pieces of one band
    const int lanes_per_row = W / 4, rows_per_inst = 64 / lanes_per_row;
    const int lr = lane / lanes_per_row, lc = (lane % lanes_per_row) * 4;
    for (long long it = (long long)blockIdx.x * 8 + wave; it < items; it += (long long)gridDim.x * 8) {
        const int per_band = npiece / chunk;
        const int band = (int)(it / per_band), first = (int)(it % per_band) * chunk;
        for (int c = 0; c < chunk; ++c) {
            const unsigned base = (unsigned)(band * 32 + lr) * (unsigned)P + (unsigned)((first + c) * W + lc);
#pragma unroll
            for (int t = 0; t < 32 / rows_per_inst; ++t)
                *reinterpret_cast<float4*>(S + (base + (unsigned)(t * rows_per_inst) * (unsigned)P)) = make_float4(1.f, 2.f, 3.f, (float)t);
        }
    }
}

int main(int argc, char** argv) {
    const int P = argc > 1 ? atoi(argv[1]) : 4096;
    const int reps = 30;
    float* S;
    CK(hipMalloc(&S, (size_t)P * P * 4));
    CK(hipMemset(S, 0, (size_t)P * P * 4));
    const size_t n4 = (size_t)P * P / 4;
    for (int r = 0; r < reps; ++r) {
        fill_f4<<<2048, 256>>>(S, n4);
        tri_mirror<7><<<256, 448>>>(S, P);
        tri_mirror<8><<<256, 512>>>(S, P);
        tri_mirror<4><<<512, 256>>>(S, P);
        full_rows<32><<<256, 512>>>(S, P, 1);
        full_rows<32><<<256, 512>>>(S, P, 4);
        full_rows<64><<<256, 512>>>(S, P, 2);
        full_rows<128><<<256, 512>>>(S, P, 1);
        full_rows<256><<<256, 512>>>(S, P, 1);
        full_rows<32><<<512, 512>>>(S, P, 4);
        full_rows<128><<<512, 512>>>(S, P, 1);
    }
    CK(hipDeviceSynchronize());
    printf("done P=%d bytes=%zu\n", P, (size_t)P * P * 4);
    return 0;
}
