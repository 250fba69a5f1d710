// dma_rate.hip -- experiment: what a CU sustains when it streams operand tiles into LDS, as a function of the bytes it keeps
// in flight and of the path: global_load_lds_dwordx4 (the LDS-DMA every GEMM / convolution kernel of csrc/ uses) against
// global_load_dwordx4 -> VGPR -> ds_write_b128.  One workgroup of 8 waves per CU, no arithmetic; the source is either one
// small region per workgroup that stays in L2 or a large one that streams from HBM.  (tools/exp, not shipped.)
//   hipcc --offload-arch=gfx950 -O3 tools/exp/dma_rate.hip -o tools/exp/dma_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// DEPTH = 1 KB pieces a wave keeps in flight (each piece: 64 lanes x 16 B = 8 rows of 128 B, like dma_rows)
// PF > 0: every 8 pieces (its next 64 lines) the wave first touches the 64 lines it will fetch PF pieces later with ONE
// global_load_dword (lane -> line): a software prefetch into L2 whose depth is not bounded by LDS space.
template <int DEPTH, int PF = 0>
__global__ __launch_bounds__(512, 2) void stream_dma(const char* __restrict__ src, size_t region, size_t wrap, int pieces, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    char* my = lds + wave * DEPTH * 1024;                       // the wave's private ring of DEPTH slots
    const char* base = src + (size_t)blockIdx.x * region;
    size_t off = (size_t)wave * 1024;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(base + off + lane * 16), (lds_void_t*)(my + d * 1024), 16, 0, 0);
        off += 8 * 1024; if (off >= wrap) off -= wrap;
    }
    unsigned pf = 0;
    for (int p = DEPTH; p < pieces; p += DEPTH) {
        if (PF > 0 && (p % 8) == 0) {
            size_t o = off + (size_t)PF * 8 * 1024 + (size_t)(lane >> 3) * 8 * 1024 + (size_t)(lane & 7) * 128;
            o = o % wrap;
            const char* a = base + o;
            asm volatile("global_load_dword %0, %1, off" : "+v"(pf) : "v"(a) : "memory");
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            // the oldest piece has landed when at most DEPTH - 1 are outstanding (+ the prefetch loads issued since)
            if (PF > 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH - 1 + (DEPTH >= 8 ? DEPTH / 8 : 1)) : "memory");
            else if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (DEPTH == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else if (DEPTH == 12) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(base + off + lane * 16), (lds_void_t*)(my + d * 1024), 16, 0, 0);
            off += 8 * 1024; if (off >= wrap) off -= wrap;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((my[lane] == 123 && lane == 77) || pf == 0x12345678u) *sink = 1;
}

// The GEMM's sharing pattern: workgroup b runs on XCD b % 8 with local index j = b / 8; it streams "A" tile row j / 8 of its
// XCD (shared by 8 workgroups) and "B" tile column j % 8 (shared by 4), two pieces of A for one of B, 12 pieces in flight.
// LOCK: a barrier every 6 pieces per wave (= one K step of 48 KB), like the kernels.
// PFD > 0: waves 0-3 touch the 256 lines of the A tile PFD steps ahead, waves 4-5 the 128 lines of the B tile, with one
// global_load_dword each (lane -> line), issued BEHIND the step's DMA so that only later DMA waits behind it in the queue.
template <bool LOCK, int PFD = 0>
__global__ __launch_bounds__(512, 2) void stream_gemm(const char* __restrict__ src, size_t stream_bytes, int steps, int* sink,
                                                      int stagger_ns = 0) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    char* my = lds + wave * 18 * 1024;                          // 3 stages x 6 pieces
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const char* A = src + (size_t)(xcd * 4 + (j >> 3)) * stream_bytes;
    const char* Bp = src + (size_t)(32 + xcd * 8 + (j & 7)) * stream_bytes;
    size_t oa = (size_t)wave * 4096, ob = (size_t)wave * 2048;   // per step: A 32 KB (4 KB per wave), B 16 KB (2 KB per wave)
    if (stagger_ns > 0) {
        // start the workgroups that share a stream one after the other: rank ((c - 2 r) mod 8) in units of stagger_ns -- every A
        // stream (fixed r) and every B stream (fixed c) then has ONE leader whose misses the others find in L2
        const int r = j >> 3, c = j & 7;
        const long long wait = (long long)(((c - 2 * r) & 7)) * stagger_ns / 10;      // wall_clock64: 100 MHz
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
    auto issue = [&](int stage) {
        char* dst = my + stage * 6 * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(A + oa + i * 1024 + lane * 16), (lds_void_t*)(dst + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(Bp + ob + i * 1024 + lane * 16), (lds_void_t*)(dst + (4 + i) * 1024), 16, 0, 0);
        oa += 32 * 1024; ob += 16 * 1024;
    };
    issue(0); issue(1);
    unsigned pf = 0;
    const bool pfw = PFD > 0 && wave < 6;
    // this wave's prefetch target inside a step's tiles: waves 0-3 -> A lines [64 w, 64 w + 64), waves 4-5 -> B lines
    const char* pbase = wave < 4 ? A + (size_t)wave * 8192 + (size_t)lane * 128 : Bp + (size_t)(wave - 4) * 8192 + (size_t)lane * 128;
    const size_t pstep = wave < 4 ? 32 * 1024 : 16 * 1024;
    for (int s = 0; s < steps; ++s) {
        if (pfw) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        if (LOCK) __builtin_amdgcn_s_barrier();
        issue((s + 2) % 3);
        if (pfw) {
            const char* a = pbase + (size_t)(s + PFD) * pstep;
            asm volatile("global_load_dword %0, %1, off" : "+v"(pf) : "v"(a) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((my[lane] == 123 && lane == 77) || pf == 0x12345678u) *sink = 1;
}

// The same pattern with the kernels' REAL addressing: a tile is 256 (A) / 128 (B) rows of a row-major matrix with `pitch` bytes
// per row, a K step takes 128 bytes of every row, one DMA instruction = 8 rows x 128 B.  Do the rows of a tile spread over the
// L2 channels?  (fc6's cell-major planes: pitch = 2 x 25088 x 2 = 100352 B = 392 x 256.)
__global__ __launch_bounds__(512, 2) void stream_rows(const char* __restrict__ src, size_t pitch, int steps, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    char* my = lds + wave * 18 * 1024;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const char* A = src + (size_t)(xcd * 4 + (j >> 3)) * 256 * pitch;                  // A tile rows
    const char* Bp = src + (size_t)(32 * 256) * pitch + (size_t)(xcd * 8 + (j & 7)) * 128 * pitch;
    const char* la = A + (size_t)(wave * 32 + (lane >> 3)) * pitch + (lane & 7) * 16;
    const char* lb = Bp + (size_t)(wave * 16 + (lane >> 3)) * pitch + (lane & 7) * 16;
    size_t k = 0;
    auto issue = [&](int stage) {
        char* dst = my + stage * 6 * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(la + (size_t)i * 8 * pitch + k), (lds_void_t*)(dst + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(lb + (size_t)i * 8 * pitch + k), (lds_void_t*)(dst + (4 + i) * 1024), 16, 0, 0);
        k += 128;
    };
    issue(0); issue(1);
    for (int s = 0; s < steps; ++s) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue((s + 2) % 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (my[lane] == 123 && lane == 77) *sink = 1;
}

template <int DEPTH>
__global__ __launch_bounds__(512, 2) void stream_reg(const char* __restrict__ src, size_t region, size_t wrap, int pieces, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    char* my = lds + wave * DEPTH * 1024;
    const char* base = src + (size_t)blockIdx.x * region;
    size_t off = (size_t)wave * 1024;
    uint4 r[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        r[d] = *reinterpret_cast<const uint4*>(base + off + lane * 16);
        off += 8 * 1024; if (off >= wrap) off -= wrap;
    }
    for (int p = DEPTH; p < pieces; p += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            *reinterpret_cast<uint4*>(my + d * 1024 + lane * 16) = r[d];           // (waits for r[d] only: loads return in order)
            r[d] = *reinterpret_cast<const uint4*>(base + off + lane * 16);
            off += 8 * 1024; if (off >= wrap) off -= wrap;
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) *reinterpret_cast<uint4*>(my + d * 1024 + lane * 16) = r[d];
    __syncthreads();
    if (my[lane] == 123 && lane == 77) *sink = 1;
}

template <int DEPTH>
void run(const char* src, int* sink, size_t region, size_t wrap, const char* what, int grid) {
    const int pieces = 16384;                                    // per wave: 16 MB; per workgroup 128 MB
    const int ldsb = 8 * DEPTH * 1024;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) hipFuncSetAttribute(reinterpret_cast<const void*>(stream_dma<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
        else hipFuncSetAttribute(reinterpret_cast<const void*>(stream_reg<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            if (mode == 0) stream_dma<DEPTH><<<grid, 512, ldsb>>>(src, region, wrap, pieces, sink);
            else stream_reg<DEPTH><<<grid, 512, ldsb>>>(src, region, wrap, pieces, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
        }
        const double bytes = (double)grid * 8 * pieces * 1024.0;
        printf("%-10s %-22s in flight %3d KB/CU: %7.1f GB/s per CU  (%6.2f TB/s chip, %.2f ms)\n", what,
               mode == 0 ? "global_load_lds x4" : "global_load x4 + ds_write", 8 * DEPTH, bytes / grid / (best * 1e-3) / 1e9,
               bytes / (best * 1e-3) / 1e12, best);
    }
}

template <int DEPTH, int PF>
void run_pf(const char* src, int* sink, size_t region, size_t wrap, const char* what, int grid) {
    const int pieces = 16384, ldsb = 8 * DEPTH * 1024;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute(reinterpret_cast<const void*>(stream_dma<DEPTH, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        stream_dma<DEPTH, PF><<<grid, 512, ldsb>>>(src, region, wrap, pieces, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
    }
    const double bytes = (double)grid * 8 * pieces * 1024.0;
    printf("%-10s global_load_lds x4 + L2 prefetch %3d pieces ahead, in flight %3d KB/CU: %7.1f GB/s per CU (%.2f ms)\n", what, PF, 8 * DEPTH,
           bytes / grid / (best * 1e-3) / 1e9, best);
}

int main() {
    const int grid = 256;
    const size_t big = (size_t)128 << 20;                        // per-workgroup region of the HBM case
    char* src; int* sink;
    hipMalloc(&src, big * grid); hipMalloc(&sink, 4);
    hipMemset(src, 1, big * grid);
    // L2-resident: every workgroup cycles through its own 256 KB (64 MB for the chip: MALL / L2)
    run<1>(src, sink, big, (size_t)256 << 10, "L2/MALL", grid);
    run<2>(src, sink, big, (size_t)256 << 10, "L2/MALL", grid);
    run<4>(src, sink, big, (size_t)256 << 10, "L2/MALL", grid);
    run<8>(src, sink, big, (size_t)256 << 10, "L2/MALL", grid);
    run<12>(src, sink, big, (size_t)256 << 10, "L2/MALL", grid);
    run<16>(src, sink, big, (size_t)256 << 10, "L2/MALL", grid);
    // all workgroups of an XCD read the SAME 64 KB (pure L2 hits)
    run<4>(src, sink, 0, (size_t)64 << 10, "L2 shared", grid);
    run<12>(src, sink, 0, (size_t)64 << 10, "L2 shared", grid);
    // HBM: distinct 128 MB per workgroup
    run<4>(src, sink, big, big, "HBM", grid);
    run<12>(src, sink, big, big, "HBM", grid);
    run<16>(src, sink, big, big, "HBM", grid);
    // the GEMM pattern: every workgroup streams the SAME 128 MB (first touch misses, everybody else hits on the miss)
    run<12>(src, sink, 0, big, "shared stream", grid);
    run_pf<12, 32>(src, sink, 0, big, "shared stream", grid);
    run_pf<12, 64>(src, sink, 0, big, "shared stream", grid);
    run_pf<12, 128>(src, sink, 0, big, "shared stream", grid);
    {
        const int steps = 1176;                                  // the fc6 sweep: 1176 K steps of 48 KB per workgroup
        const size_t sb = (size_t)40 << 20;                      // >= 1178 x 32 KB per stream
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int lock = 0; lock < 9; ++lock) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(stream_gemm<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(stream_gemm<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(stream_gemm<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(stream_gemm<true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(stream_gemm<true, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a);
                if (lock >= 5) stream_gemm<true><<<grid, 512, 144 * 1024>>>(src, sb, steps, sink, lock == 5 ? 500 : lock == 6 ? 1000 : lock == 7 ? 2000 : 4000);
                else if (lock == 1) stream_gemm<true><<<grid, 512, 144 * 1024>>>(src, sb, steps, sink);
                else if (lock == 2) stream_gemm<true, 4><<<grid, 512, 144 * 1024>>>(src, sb, steps, sink);
                else if (lock == 3) stream_gemm<true, 8><<<grid, 512, 144 * 1024>>>(src, sb, steps, sink);
                else if (lock == 4) stream_gemm<true, 16><<<grid, 512, 144 * 1024>>>(src, sb, steps, sink);
                else stream_gemm<false><<<grid, 512, 144 * 1024>>>(src, sb, steps, sink);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
            }
            printf("GEMM pattern (4 A + 8 B streams per XCD, 48 KB per step, 96 KB in flight)%s: %7.1f GB/s per CU, %.3f ms for %d steps (%.2f us per step)\n",
                   lock == 0 ? "" : lock == 1 ? " + barrier per step" : lock == 2 ? " + barrier + L2 prefetch 4 steps ahead" : lock == 3 ? " + barrier + L2 prefetch 8 ahead" : lock == 4 ? " + barrier + L2 prefetch 16 ahead" : lock == 5 ? " + barrier + sharers staggered 0.5 us" : lock == 6 ? " + barrier + staggered 1 us" : lock == 7 ? " + barrier + staggered 2 us" : " + barrier + staggered 4 us", 48.0 * 1024 * steps / (best * 1e-3) / 1e9, best, steps, best * 1e3 / steps);
        }
    }
    {
        const int steps = 784;                                   // 784 x 128 B = the 100352-byte rows of fc6's planes
        hipFuncSetAttribute(reinterpret_cast<const void*>(stream_rows), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        const size_t pitches[] = {100352, 100352 + 128, 100352 + 256, 100352 + 512, 100352 + 1024, 100352 + 2048 + 256, 131072, 131072 + 256};
        for (size_t pitch : pitches) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a);
                stream_rows<<<grid, 512, 144 * 1024>>>(src, pitch, steps, sink);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
            }
            printf("row tiles, pitch %7zu B (%5zu x 256 + %3zu): %7.1f GB/s per CU, %.2f us per step\n", pitch, pitch / 256, pitch % 256,
                   48.0 * 1024 * steps / (best * 1e-3) / 1e9, best * 1e3 / steps);
        }
    }
    return 0;
}
