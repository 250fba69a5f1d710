// odw_planes.h -- an fp32 value as bf16 planes hi + mid (+ lo), for kernels that WRITE a split-precision operand
// themselves instead of handing an fp32 tensor to split_rows_kernel (csrc/split.hip explains the scheme).
#pragma once
#include <hip/hip_runtime.h>

namespace odwpl {

constexpr int kMaxTerms = 8;
struct Pattern { int T; int p[kMaxTerms]; };       // plane codes along the reduction axis: 0 hi, 1 mid, 2 lo, 3 zeros

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// two values -> one packed pair of bf16 (v_cvt_pk_bf16_f32: round to nearest even)
__device__ __forceinline__ unsigned pk(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float rest(float x, float r) {          // inf / nan live in the hi plane alone
    return (__float_as_uint(x) & 0x7f800000u) == 0x7f800000u ? 0.0f : r;
}

// planes of two neighbouring values, packed like pk(); both subtractions are exact in fp32
__device__ __forceinline__ void split2(float a, float b, bool need_lo, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = pk(a, b);
    const float ra = rest(a, a - __uint_as_float(hi << 16)), rb = rest(b, b - __uint_as_float(hi & 0xffff0000u));
    mid = pk(ra, rb);
    lo = need_lo ? pk(ra - __uint_as_float(mid << 16), rb - __uint_as_float(mid & 0xffff0000u)) : 0u;
}

inline bool pattern_ok(const int* pattern, int T, Pattern& pat) {
    if (!pattern || T < 1 || T > kMaxTerms) return false;
    pat.T = T;
    for (int t = 0; t < kMaxTerms; ++t) pat.p[t] = 3;
    for (int t = 0; t < T; ++t) {
        if (pattern[t] < 0 || pattern[t] > 3) return false;
        pat.p[t] = pattern[t];
    }
    return true;
}

}  // namespace odwpl
