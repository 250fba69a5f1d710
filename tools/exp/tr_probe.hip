// Probe of ds_read_b64_tr_b16 (gfx950): every 16-bit LDS element holds its own index; lane l reads 8 bytes at a
// per-lane address; prints, per lane, the four 16-bit values it receives.  Mode 0: address = 8 * lane (contiguous);
// mode 1: lane i of each 16-lane group g points at row (i >> 2) of a [4][pitch] block, columns 4 (i & 3) .. + 3,
// block column base 16 g (the layout a K-major MFMA operand tile would use), pitch = 64 elements.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s lds_v4s;

__global__ void probe(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int elem;
    if (mode == 0) elem = 4 * l;
    else { const int g = l >> 4, i = l & 15; elem = (i >> 2) * 64 + 16 * g + 4 * (i & 3); }
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(lds + elem));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)r[j];
}

int main() {
    unsigned short* d; unsigned short h[256];
    hipMalloc(&d, sizeof(h));
    for (int mode = 0; mode < 2; ++mode) {
        probe<<<1, 64>>>(d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
