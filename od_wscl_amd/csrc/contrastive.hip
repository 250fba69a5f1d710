// contrastive.hip -- proposal x proposal similarity and the SupConLossV2
// (NT-Xent style) contrastive loss, forward + backward, for gfx950.
//
// Reference behaviour:
//   sim_mat = E E^T                      roi_heads/weak_head/loss.py:319
//   SupConLossV2.forward                 roi_heads/sim_head/sim_loss.py:49-80
//   (its backward is torch autograd; the closed form is in SURVEY.md s8a)
//
// MI355X structure.  Everything is built on ONE wave-level primitive: a 32x32
// tile of X Y^T for 128-d rows on the exact-fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32, bit-identical to an fmaf chain).  Because a dot
// product does not care in which order k is visited, lanes 0-31 walk
// k = 0..63 and lanes 32-63 walk k = 64..127: every lane reads ONE contiguous
// 256-byte half-row with 16-byte loads straight into MFMA operand registers --
// no LDS staging, no transposes.
//
//   * pairwise_sim: upper-triangular 64x64 block tiles only; the mirrored
//     tile is written through a padded LDS transpose so both stores are
//     128-byte coalesced.  HBM traffic = E once + S once.
//   * supcon: "flash" form -- S is never materialised.  The tile is computed
//     TRANSPOSED (rows j in registers, column i = lane) so the row-i softmax
//     statistics (running max, A_i = sum over same-label j, B_i = sum over all
//     j != i) are lane-local; the two half-waves are merged once at the end.
//     Backward recomputes the tile, forms H = G + G^T in registers and feeds
//     it straight back into the matrix pipe as the A operand of H F (the C
//     layout of one MFMA is the A layout of the next when k is walked in the
//     same permuted order), accumulating dF in 64 accumulator registers.
//     The j range is split across waves (grid.y) for occupancy; partials are
//     merged by tiny deterministic combine kernels -- no atomics anywhere.
#include "odw_common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kD = 128;      // embedding width served by the MFMA path (Sim_Net output, sim_net.py:14)
constexpr int kHalf = 64;    // floats per lane per row

// row index (0..31) held in accumulator register k of a 32x32 MFMA C tile
__device__ __forceinline__ int crow(int k, int half) { return (k & 3) + 8 * (k >> 2) + 4 * half; }

// this lane's 64-float half of row `row` of X (zeros past nrows)
__device__ __forceinline__ void load_half_row(const float* __restrict__ X, int row, int nrows, int half,
                                              float (&v)[kHalf]) {
    if (row < nrows) {
        const float4* p = reinterpret_cast<const float4*>(X + (size_t)row * kD + half * kHalf);
#pragma unroll
        for (int q = 0; q < kHalf / 4; ++q) {
            float4 t = p[q];
            v[4 * q + 0] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < kHalf; ++q) v[q] = 0.0f;
    }
}

// acc[m][n] = sum_k A[m][k] B[n][k] with lane (l&31) supplying row m of A and row n of B
__device__ __forceinline__ f32x16 tile_dot(const float (&a)[kHalf], const float (&b)[kHalf]) {
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < kHalf; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
    return acc;
}

// ------------------------------------------------------------------ pairwise
__global__ __launch_bounds__(256) void pairwise_sim_kernel(const float* __restrict__ E, int P,
                                                           float* __restrict__ S) {
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;
    __shared__ float tr[4][32 * 33];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, c = lane & 31;
    const int I = bi * 64 + (wave >> 1) * 32, J = bj * 64 + (wave & 1) * 32;
    float a[kHalf], b[kHalf];
    load_half_row(E, I + c, P, half, a);
    load_half_row(E, J + c, P, half, b);
    const f32x16 acc = tile_dot(a, b);   // rows: I + crow, col: J + c
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = I + crow(k, half);
        if (r < P && J + c < P) S[(size_t)r * P + J + c] = acc[k];
    }
    if (bi != bj) {  // block-uniform
        float* t = tr[wave];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[crow(k, half) * 33 + c] = acc[k];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int cc = 2 * it + half;  // column of the direct tile = row of the mirror
            if (J + cc < P && I + c < P) S[(size_t)(J + cc) * P + I + c] = t[c * 33 + cc];
        }
    }
}

// ------------------------------------------------------------------ pairwise, split-bf16 form (round 3)
// The same E E^T on the bf16 matrix cores at fp32 grade: every fp32 value is carried as three bf16 planes
// hi + mid + lo (csrc/split.hip) and a product as the six plane products of order <= 2 -- 6 x 2.5 PF-class MFMAs
// instead of one 157 TF-class fp32 MFMA chain.  The kernel is then what its roofline says it is: a 4 P^2-byte WRITE
// of S (64 MB at P = 4000), and its structure exists to keep that write stream busy from the first microsecond:
//   * ONE launch.  A workgroup (8 waves) owns a 256-row panel of S and a run of 32-column blocks of it (upper
//     triangle only; the runs are cut so that the 256 workgroups carry equal numbers of blocks).  Each wave keeps
//     the three planes of ITS 32 rows of E for the whole K = 128 in registers (96 VGPRs): the A operand never
//     touches LDS again.
//   * The 32 rows of E of a column block are fetched as fp32 (512 coalesced bytes per row), split into planes in
//     registers by the thread that fetched them (no pre-pass, no second launch: the split of round 2's pre-pass is
//     ~14 VALU instructions per thread and block here) and parked in LDS, double buffered: 0.5 KB of LDS reads per
//     MFMA, one barrier per block, the next block's loads in flight under the current block's 48 MFMAs per wave.
//   * Every block ends with its stores -- the direct tile as 128-byte row segments, the mirrored tile through a
//     wave-private padded LDS transpose so that it, too, leaves as full 128-byte lines -- and the wave moves on:
//     stores of block b drain under the MFMAs of block b + 1, all through the kernel, instead of one burst at its end
//     (round 2's one-round grid ran load -> MFMA -> store in lock-step across the chip: 25 + 18 us ~ the 38 us
//     measured at P = 4000).
//   * Diagonal 32x32 tiles write their upper triangle to both places, so S == S^T bit for bit.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ps_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ps_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned ps_pk(float a, float b) {
    const ps_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ps_bf16x2));      // v_cvt_pk_bf16_f32 (RNE)
}

// 8 fp32 -> hi / mid / lo planes, 8 bf16 (16 bytes) each; both subtractions are exact in fp32
__device__ __forceinline__ void ps_split8(const float4 a, const float4 b, uint4& hi, uint4& mid, uint4& lo) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        h[q] = ps_pk(x[2 * q], x[2 * q + 1]);
        const float r0 = x[2 * q] - __uint_as_float(h[q] << 16), r1 = x[2 * q + 1] - __uint_as_float(h[q] & 0xffff0000u);
        m[q] = ps_pk(r0, r1);
        l[q] = ps_pk(r0 - __uint_as_float(m[q] << 16), r1 - __uint_as_float(m[q] & 0xffff0000u));
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    mid = make_uint4(m[0], m[1], m[2], m[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

constexpr int kPwWaves = 8;                          // 7 compute waves + 1 loader wave
constexpr int kPwCompute = kPwWaves - 1;
constexpr int kPwPanel = 32 * kPwCompute;            // rows of S per workgroup panel (224)
constexpr int kPwPitch = 16 * 16 + 16;               // bytes per staged plane row: 128 k x 2 B + 16 B pad (bank spread)
constexpr int kPwSlot = 3 * 32 * kPwPitch;           // one column block: 3 planes x 32 rows
constexpr int kPwTrPitch = 36;                       // floats per row of a wave's 32 x 32 transpose scratch
constexpr int kPwScratch = kPwCompute * 32 * kPwTrPitch * 4;                // the compute waves' transpose scratch
constexpr int kPwStage = 4;                          // staging areas of the panel stage: the two block slots + two more behind the scratch
constexpr int kPwLds = 2 * kPwSlot + kPwScratch + (kPwStage - 2) * kPwSlot;        // 52224 + 32256 + 52224 bytes

// number of 32-column blocks panel p works on: from its first row's block to the last block of S
__device__ __host__ inline int pw_blocks_of(int p, int nblk32) { const int b = nblk32 - p * (kPwPanel / 32); return b > 0 ? b : 0; }

#ifdef ODW_EXPERIMENTS          // timing studies (WRONG results): 1 no direct stores, 2 no mirror stores, 4 no MFMAs
#define PW_DBG(bit) (dbg & (bit))
#else
#define PW_DBG(bit) false
#endif
// tools/exp/pairwise_timeline.hip compiles this file with ODW_PW_TIMELINE: lane 0 of every wave stamps wall_clock64()
// (100 MHz) at the phase boundaries of its first iterations
#ifdef ODW_PW_TIMELINE
__device__ long long g_pw_tl[1024 * 8 * 32];
#define PW_T(i) do { if (lane == 0 && (i) < 32) g_pw_tl[(blockIdx.x * 8 + wave) * 32 + (i)] = wall_clock64(); } while (0)
#else
#define PW_T(i) do { } while (0)
#endif
// How this structure was arrived at (P = 4000, rocprofv3 + PMC, tools/exp/pairwise_dbg.sh, tools/pmc_pairwise.sh).  The
// first form -- eight symmetric waves, each fetching its share of the next column block, running its MFMAs and storing
// its tile -- took 26-28 us whatever the order of its instructions: without stores 20, without MFMAs 20, without both
// 13, i.e. MFMA time (7 us), store time (7 us) and the fetch / barrier skeleton simply ADDED UP, and
// SQ_WAIT_INST_ANY showed every wave spending 8 us in s_waitcnt.  gfx950 counts loads and stores in ONE counter
// (vmcnt) and the two kinds may retire out of order, so a wave that has stores in flight can only consume a fetched
// row after vmcnt(0): every block waited for the previous tile's stores to be acknowledged (>= 1.5 us under a chip-wide
// store burst) before the next barrier could be reached.  (__syncthreads() has the same wait built in: pw_barrier.)
// Hence the roles: ONE loader wave per workgroup fetches, splits and parks the next column block and never stores;
// SEVEN compute waves read fragments, run MFMAs and store, and never wait on vmcnt at all -- their stores drain under
// the next tile's MFMAs.
struct PwRows { float4 a0, a1; };

// Workgroup barrier that orders LDS traffic ONLY (see above).  The stores of S are never read back inside the kernel.
__device__ __forceinline__ void pw_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(kPwWaves * 64, 1) void pairwise_sim_panel_kernel(const float* __restrict__ E, int P,
                                                                              float* __restrict__ S, int total_items, int dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave == kPwCompute;
    const int nblk32 = (P + 31) / 32, npanel = (P + kPwPanel - 1) / kPwPanel;
    // this workgroup's run of (panel, block) items in panel-major order
    const long long lo_it = (long long)total_items * blockIdx.x / gridDim.x, hi_it = (long long)total_items * (blockIdx.x + 1) / gridDim.x;
    if (lo_it >= hi_it) return;
    PW_T(0);
    int panel = 0, blk;
    {
        long long rest = lo_it;
        while (panel < npanel && rest >= pw_blocks_of(panel, nblk32)) { rest -= pw_blocks_of(panel, nblk32); ++panel; }
        blk = panel * (kPwPanel / 32) + (int)rest;
    }
    float* const scratch = reinterpret_cast<float*>(lds + 2 * kPwSlot) + (loader ? 0 : wave) * 32 * kPwTrPitch;
    const bool vec = (P & 3) == 0;
    const bool small = (unsigned long long)P * (unsigned long long)P < (1ull << 30);      // 32-bit element offsets

    // (row, 8-k chunk) items of a 32-row block: rows past the end are clamped to the last row (their products are
    // never stored), so every load is unconditional
    auto fetch_item = [&](int row, int chunk) {
        PwRows r;
        const float4* src = reinterpret_cast<const float4*>(E + (size_t)min(row, P - 1) * kD + chunk * 8);
        r.a0 = src[0]; r.a1 = src[1];
        return r;
    };
    auto park_item = [&](const PwRows& r, int slot, int row, int chunk) {
        uint4 h, m, l;
        unsigned char* base = lds + (slot < 2 ? slot * kPwSlot : kPwScratch + slot * kPwSlot) + row * kPwPitch + chunk * 16;
        ps_split8(r.a0, r.a1, h, m, l);
        *reinterpret_cast<uint4*>(base) = h;
        *reinterpret_cast<uint4*>(base + 32 * kPwPitch) = m;
        *reinterpret_cast<uint4*>(base + 64 * kPwPitch) = l;
    };
    auto advance = [&](int& pn, int& b) {                // (panel, block) after (pn, b)
        if (++b >= nblk32) { ++pn; b = pn * (kPwPanel / 32); }
    };
    const int srow = tid >> 4, schunk = tid & 15;        // all 512 threads: one item each of a 32-row block
    const int lrow = lane >> 4, lchunk = lane & 15;      // loader wave: items (4 q + lrow, lchunk), q = 0..7

    bf16x8 aH[8], aM[8], aL[8];
    int have_panel = -1, slot = 0;
    int p1 = panel, b1 = blk;
    advance(p1, b1);
    bool parked = false;

    for (long long it = lo_it; it < hi_it; ++it) {
        if (panel != have_panel) {
            // ---- this workgroup's 224 rows of E: seven 32-row pieces fetched with coalesced rows by all eight waves (all
            // fetches issued before the first wait), split, parked in four staging areas at a time; compute wave w
            // takes piece w's planes for the whole K into registers (lane: row l31, 8 k at 16 kk + 8 half)
            if (parked) pw_barrier();                    // every wave is done reading the slots
            PwRows pr[kPwCompute];
#pragma unroll
            for (int q = 0; q < kPwCompute; ++q) pr[q] = fetch_item(panel * kPwPanel + q * 32 + srow, schunk);
            const PwRows first = fetch_item(blk * 32 + srow, schunk);
            // (four pieces per round -- the two block slots and two more areas behind the scratch -- : two barrier-separated
            // rounds instead of the four the two slots alone allowed, ~1.1 us each before the first MFMA of a run)
#pragma unroll
            for (int hp = 0; hp < (kPwCompute + kPwStage - 1) / kPwStage; ++hp) {
#pragma unroll
                for (int j = 0; j < kPwStage; ++j)
                    if (kPwStage * hp + j < kPwCompute) park_item(pr[kPwStage * hp + j < kPwCompute ? kPwStage * hp + j : 0], j, srow, schunk);
                pw_barrier();
                if (!loader && wave / kPwStage == hp) {
                    const int sj = wave % kPwStage;
                    const unsigned char* sa = lds + (sj < 2 ? sj * kPwSlot : kPwScratch + sj * kPwSlot) + l31 * kPwPitch + half * 16;
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        aH[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sa + kk * 32));
                        aM[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sa + 32 * kPwPitch + kk * 32));
                        aL[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sa + 64 * kPwPitch + kk * 32));
                    }
                }
                pw_barrier();
            }
            park_item(first, 0, srow, schunk);
            slot = 0;
            have_panel = panel;
            parked = true;
            PW_T(1);
        }
        pw_barrier();                                    // block `blk` is parked in `slot`; the other slot is free
        PW_T(2 + 3 * (int)(it - lo_it));
        if (loader) {
            // ---- the loader wave: the next block's 32 rows (8 items per lane, 16 loads in flight), split, parked in the
            // other slot.  Unconditional (past the end of the run the slot receives rows nobody reads).  Its vmcnt
            // only ever counts loads.  (Fetching two blocks ahead -- the rows parked here loaded during the previous
            // iteration -- measured no faster: in steady state a block lasts as long as the chip needs to WRITE the
            // 14.7 MB that 256 x 7 tiles produce, ~3 us, not as long as this wave's load latency.)
            const int nb = min(b1, nblk32 - 1) * 32;
            PwRows q8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) q8[q] = fetch_item(nb + 4 * q + lrow, lchunk);
#ifdef ODW_PW_TIMELINE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PW_T(3 + 3 * (int)(it - lo_it));             // (loader wave: its fetches have arrived)
#endif
#pragma unroll
            for (int q = 0; q < 8; ++q) park_item(q8[q], slot ^ 1, 4 * q + lrow, lchunk);
        } else {
            const int r0 = panel * kPwPanel + wave * 32, c0 = blk * 32;
            if (c0 >= r0 && r0 < P) {                    // this wave's tile lies on or above the diagonal
                const unsigned char* sb = lds + slot * kPwSlot + l31 * kPwPitch + half * 16;
                f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc;
                if (!PW_DBG(4))
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const bf16x8 bH = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + kk * 32));
                    const bf16x8 bM = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + 32 * kPwPitch + kk * 32));
                    const bf16x8 bL = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + 64 * kPwPitch + kk * 32));
                    // two accumulators, alternated (shorter dependent chains); per accumulator the smaller terms first
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aL[kk], bH, acc, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aM[kk], bH, acc1, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aH[kk], bL, acc, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aH[kk], bM, acc1, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aM[kk], bM, acc, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aH[kk], bH, acc1, 0, 0, 0);
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] += acc1[k];
                PW_T(3 + 3 * (int)(it - lo_it));
                const int col = c0 + l31;
                const bool full = small && vec && r0 + 32 <= P && c0 + 32 <= P;      // wave-uniform: no per-element predicates
                if (c0 == r0) {
                    // diagonal tile: (r, c) and (c, r) were accumulated in different term orders -- the upper triangle
                    // goes to both places so that S is exactly symmetric
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int r = crow(k, half);
                        if (r <= l31 && col < P) {
                            S[(size_t)(r0 + r) * P + col] = acc[k];
                            S[(size_t)col * P + r0 + r] = acc[k];
                        }
                    }
                } else {
                    // mirror S[c0 + n][r0 + m]: the lane holds row n = l31 as 4 runs of 4 consecutive m -- parked in the
                    // wave's scratch as rows of 32 floats, read back 8 lanes per row, stored as full 128-byte lines
                    // (the wave's own LDS traffic is ordered: no barrier)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(scratch + l31 * kPwTrPitch + 8 * q + 4 * half) =
                            make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                    if (full) {
                        // interior tile: 32-bit element offsets from the kernel-argument base, no predicates
                        const unsigned dbase = (unsigned)(r0 + 4 * half) * (unsigned)P + (unsigned)col;
                        if (!PW_DBG(1))
#pragma unroll
                        for (int k = 0; k < 16; ++k)      // direct tile: one 128-byte row segment per half-wave
                            S[dbase + (unsigned)((k & 3) + 8 * (k >> 2)) * (unsigned)P] = acc[k];
                        const unsigned mbase = (unsigned)(c0 + (lane >> 3)) * (unsigned)P + (unsigned)(r0 + (lane & 7) * 4);
                        if (!PW_DBG(2))
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            *reinterpret_cast<float4*>(S + (mbase + (unsigned)(8 * t) * (unsigned)P)) =
                                *reinterpret_cast<const float4*>(scratch + (t * 8 + (lane >> 3)) * kPwTrPitch + (lane & 7) * 4);
                    } else {
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            const int r = r0 + crow(k, half);
                            if (r < P && col < P) S[(size_t)r * P + col] = acc[k];
                        }
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int n = t * 8 + (lane >> 3), m4 = (lane & 7) * 4;
                            const float4 v = *reinterpret_cast<const float4*>(scratch + n * kPwTrPitch + m4);
                            const int row = c0 + n, cc = r0 + m4;
                            if (row < P) {
                                float* dst = S + (size_t)row * P + cc;
                                if (vec && cc + 3 < P) *reinterpret_cast<float4*>(dst) = v;
                                else {
                                    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                                    for (int u = 0; u < 4; ++u) if (cc + u < P) dst[u] = e[u];
                                }
                            }
                        }
                    }
                }
            }
        }
        PW_T(4 + 3 * (int)(it - lo_it));
        slot ^= 1;
        blk = b1; panel = p1;
        advance(p1, b1);
    }
    PW_T(31);
}

// any D (multiple of 4): plain wave-per-row kernel, used when D != 128
__global__ void pairwise_sim_generic(const float* __restrict__ E, int P, int D, float* __restrict__ S) {
    const size_t total = (size_t)P * P;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (size_t)gridDim.x * blockDim.x) {
        int i = (int)(t / P), j = (int)(t - (size_t)i * P);
        const float4* x = reinterpret_cast<const float4*>(E + (size_t)i * D);
        const float4* y = reinterpret_cast<const float4*>(E + (size_t)j * D);
        float acc = 0.0f;
        for (int q = 0; q < D / 4; ++q) {
            float4 u = x[q], v = y[q];
            acc = fmaf(u.x, v.x, acc); acc = fmaf(u.y, v.y, acc);
            acc = fmaf(u.z, v.z, acc); acc = fmaf(u.w, v.w, acc);
        }
        S[t] = acc;
    }
}

// -------------------------------------------------------------------- supcon
// partial statistics of rows i over the j blocks of one split.
// part layout: [nsplit][3][Npad], Npad = 32*ceil(N/32)
__global__ __launch_bounds__(64) void supcon_stats_kernel(const float* __restrict__ F,
                                                          const int* __restrict__ labels, int N,
                                                          float inv_tau, float* __restrict__ part) {
    const int lane = threadIdx.x, half = lane >> 5, c = lane & 31;
    const int nblk = (N + 31) / 32, npad = nblk * 32;
    const int I = blockIdx.x * 32, i = I + c;
    const int sp = blockIdx.y, nsplit = gridDim.y;
    float b[kHalf];
    load_half_row(F, i, N, half, b);
    const int yi = i < N ? labels[i] : -1;
    float m = -__builtin_inff(), A = 0.0f, Bs = 0.0f;
    for (int jb = sp; jb < nblk; jb += nsplit) {
        const int J = jb * 32;
        float a[kHalf];
        load_half_row(F, J + c, N, half, a);
        const f32x16 acc = tile_dot(a, b);  // acc[k] = f_{J+crow(k)} . f_i
        float tmax = -__builtin_inff();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int j = J + crow(k, half);
            if (j < N) tmax = fmaxf(tmax, acc[k] * inv_tau);
        }
        if (tmax > m) {
            const float sc = expf(m - tmax);  // m = -inf -> 0
            A *= sc; Bs *= sc; m = tmax;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int j = J + crow(k, half);
            if (j < N && j != i) {
                const float e = expf(acc[k] * inv_tau - m);
                Bs += e;
                if (labels[j] == yi) A += e;
            }
        }
    }
    // merge the two half-waves (same i, disjoint j sets)
    const float m2 = __shfl_xor(m, 32), A2 = __shfl_xor(A, 32), B2 = __shfl_xor(Bs, 32);
    const float M = fmaxf(m, m2);
    const float s1 = (m == -__builtin_inff()) ? 0.0f : expf(m - M);
    const float s2 = (m2 == -__builtin_inff()) ? 0.0f : expf(m2 - M);
    if (half == 0 && i < N) {
        float* p = part + (size_t)sp * 3 * npad;
        p[i] = M;
        p[npad + i] = A * s1 + A2 * s2;
        p[2 * npad + i] = Bs * s1 + B2 * s2;
    }
}

// merge splits -> stats[3][Npad] (m, A, B), per-row loss, mean loss
__global__ __launch_bounds__(256) void supcon_combine_kernel(const float* __restrict__ part, int nsplit,
                                                             const float* __restrict__ w, int N,
                                                             float* __restrict__ stats,
                                                             float* __restrict__ loss) {
    const int npad = ((N + 31) / 32) * 32;
    __shared__ float red[256];
    float local = 0.0f;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        float M = -__builtin_inff();
        for (int s = 0; s < nsplit; ++s) M = fmaxf(M, part[(size_t)s * 3 * npad + i]);
        float A = 0.0f, B = 0.0f;
        for (int s = 0; s < nsplit; ++s) {
            const float* p = part + (size_t)s * 3 * npad;
            const float ms = p[i];
            const float sc = (ms == -__builtin_inff()) ? 0.0f : expf(ms - M);
            A += p[npad + i] * sc;
            B += p[2 * npad + i] * sc;
        }
        stats[i] = M; stats[npad + i] = A; stats[2 * npad + i] = B;
        local += -logf(A / B) * w[i];   // sim_loss.py:76-78
    }
    red[threadIdx.x] = local;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0] / (float)N;
}

// partial dF of rows i over the j blocks of one split.  dpart: [nsplit][Npad][128]
__global__ __launch_bounds__(64) void supcon_grad_kernel(const float* __restrict__ F,
                                                         const int* __restrict__ labels,
                                                         const float* __restrict__ w,
                                                         const float* __restrict__ stats, int N,
                                                         float inv_tau, float out_scale,
                                                         float* __restrict__ dpart) {
    const int lane = threadIdx.x, half = lane >> 5, c = lane & 31;
    const int nblk = (N + 31) / 32, npad = nblk * 32;
    const int I = blockIdx.x * 32, i = I + c;
    const int sp = blockIdx.y, nsplit = gridDim.y;
    const float invN = 1.0f / (float)N;
    float b[kHalf];
    load_half_row(F, i, N, half, b);
    const bool vi = i < N;
    const int yi = vi ? labels[i] : -1;
    const float mi = vi ? stats[i] : 0.0f;
    const float wi = vi ? w[i] * invN : 0.0f;
    const float ciA = vi ? wi / stats[npad + i] : 0.0f;
    const float ciB = vi ? wi / stats[2 * npad + i] : 0.0f;
    f32x16 out[4];
#pragma unroll
    for (int dc = 0; dc < 4; ++dc) out[dc] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int jb = sp; jb < nblk; jb += nsplit) {
        const int J = jb * 32;
        float a[kHalf];
        load_half_row(F, J + c, N, half, a);
        const f32x16 acc = tile_dot(a, b);
        float H[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int j = J + crow(k, half);
            float h = 0.0f;
            if (vi && j < N && j != i) {
                const float s = acc[k] * inv_tau;
                const bool same = labels[j] == yi;
                const float wj = w[j] * invN;
                // G_ij + G_ji: e_ij (w_i/N)(1/B_i - [same]/A_i) + e_ji (w_j/N)(1/B_j - [same]/A_j)
                const float g1 = expf(s - mi) * (ciB - (same ? ciA : 0.0f));
                const float g2 = expf(s - stats[j]) *
                                 (wj / stats[2 * npad + j] - (same ? wj / stats[npad + j] : 0.0f));
                h = g1 + g2;
            }
            H[k] = h;
        }
        // dF_i += sum_j H_ij f_j : A operand = H (already in A layout), B operand = rows of F
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int j = J + crow(k, half);
            const float* fj = F + (size_t)j * kD + c;
#pragma unroll
            for (int dc = 0; dc < 4; ++dc) {
                const float bv = j < N ? fj[dc * 32] : 0.0f;
                out[dc] = __builtin_amdgcn_mfma_f32_32x32x2f32(H[k], bv, out[dc], 0, 0, 0);
            }
        }
    }
    float* dp = dpart + (size_t)sp * npad * kD;
#pragma unroll
    for (int dc = 0; dc < 4; ++dc)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int r = I + crow(k, half);
            if (r < N) dp[(size_t)r * kD + dc * 32 + c] = out[dc][k] * out_scale;
        }
}

__global__ void supcon_grad_combine(const float* __restrict__ dpart, int nsplit, int N, int npad,
                                    float* __restrict__ dF) {
    const int total = N * kD / 4;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        float4 acc = reinterpret_cast<const float4*>(dpart)[t];
        for (int s = 1; s < nsplit; ++s) {
            float4 v = reinterpret_cast<const float4*>(dpart + (size_t)s * npad * kD)[t];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        reinterpret_cast<float4*>(dF)[t] = acc;
    }
}

int supcon_nsplit(int N) {
    const int nblk = (N + 31) / 32;
    int s = 1024 / (nblk > 0 ? nblk : 1);
    if (s > 16) s = 16;
    if (s > nblk) s = nblk;
    if (s < 1) s = 1;
    return s;
}

}  // namespace

ODW_EXPORT int64_t odw_pairwise_sim_workspace(int P, int D) {
    return D == kD && P > 0 ? odw_align_up((int64_t)P * 384 * 2, 256) : 0;
}

ODW_EXPORT int odw_pairwise_sim_ws(const float* E, int P, int D, float* S, void* workspace, int64_t workspace_bytes,
                                   void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(P >= 0 && D > 0 && D % 4 == 0, "pairwise_sim: bad dims P=%d D=%d", P, D);
    if (P == 0) return ODW_OK;
    ODW_REQUIRE(E && S, "pairwise_sim: null pointer");
    ODW_REQUIRE((((uintptr_t)E) & 15) == 0, "pairwise_sim: E must be 16-byte aligned");
    static const bool fp32_chain = getenv("ODW_PAIRWISE_FP32") != nullptr;      // force the exact-fp32 MFMA form (comparison)
    if (D == kD && !fp32_chain && (((uintptr_t)S) & 15) == 0) {
        // split-bf16 panel form: one launch, no workspace (the planes are made in registers)
        (void)workspace; (void)workspace_bytes;
        const int nblk32 = (P + 31) / 32, npanel = (P + kPwPanel - 1) / kPwPanel;
        long long items = 0;
        for (int pnl = 0; pnl < npanel; ++pnl) items += pw_blocks_of(pnl, nblk32);
        const int grid = (int)(items < ODW_NUM_CU ? items : ODW_NUM_CU);
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(pairwise_sim_panel_kernel),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, kPwLds);      // once
        ODW_CHECK_HIP(attr, "pairwise attr");
#ifdef ODW_EXPERIMENTS
        static const int dbg = getenv("ODW_PAIRWISE_DBG") ? atoi(getenv("ODW_PAIRWISE_DBG")) : 0;
#else
        const int dbg = 0;
#endif
        pairwise_sim_panel_kernel<<<grid, kPwWaves * 64, kPwLds, stream>>>(E, P, S, (int)items, dbg);
        ODW_CHECK_LAUNCH("pairwise_sim_panel_kernel");
    } else if (D == kD) {
        const int nb = (P + 63) / 64;
        pairwise_sim_kernel<<<dim3(nb, nb), 256, 0, stream>>>(E, P, S);
        ODW_CHECK_LAUNCH("pairwise_sim_kernel");
    } else {
        size_t total = (size_t)P * P;
        int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        pairwise_sim_generic<<<grid, 256, 0, stream>>>(E, P, D, S);
        ODW_CHECK_LAUNCH("pairwise_sim_generic");
    }
    return ODW_OK;
}

ODW_EXPORT int odw_pairwise_sim(const float* E, int P, int D, float* S, void* stream_) {
    return odw_pairwise_sim_ws(E, P, D, S, nullptr, 0, stream_);          // (the panel kernel needs no workspace)
}

ODW_EXPORT int64_t odw_supcon_workspace(int N) {
    if (N < 1) N = 1;
    const int64_t npad = ((N + 31) / 32) * 32;
    const int64_t ns = supcon_nsplit(N);
    return odw_align_up(3 * npad * 4, 256) + odw_align_up(ns * 3 * npad * 4, 256) +
           odw_align_up(ns * npad * kD * 4, 256);
}

ODW_EXPORT int odw_supcon_v2(const float* F, const int32_t* labels, const float* w, int N, int D, float tau,
                             float grad_scale, float* loss, float* dF, void* workspace,
                             int64_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(N >= 1, "supcon_v2: N=%d (the reference would take the mean of an empty tensor)", N);
    ODW_REQUIRE(D == kD, "supcon_v2: D=%d unsupported (Sim_Net emits 128-d embeddings)", D);
    ODW_REQUIRE(tau > 0.0f, "supcon_v2: temperature must be > 0");
    ODW_REQUIRE(F && labels && w && loss, "supcon_v2: null pointer");
    ODW_REQUIRE((((uintptr_t)F) & 15) == 0 && (dF == nullptr || (((uintptr_t)dF) & 15) == 0),
                "supcon_v2: F/dF must be 16-byte aligned");
    if (!workspace || workspace_bytes < odw_supcon_workspace(N)) {
        odw_set_error("supcon_v2: workspace %lld < %lld bytes", (long long)workspace_bytes,
                      (long long)odw_supcon_workspace(N));
        return ODW_EWORKSPACE;
    }
    const int nblk = (N + 31) / 32, npad = nblk * 32;
    const int ns = supcon_nsplit(N);
    unsigned char* p = (unsigned char*)workspace;
    float* stats = (float*)p; p += odw_align_up((int64_t)3 * npad * 4, 256);
    float* part = (float*)p;  p += odw_align_up((int64_t)ns * 3 * npad * 4, 256);
    float* dpart = (float*)p;
    const float inv_tau = 1.0f / tau;
    supcon_stats_kernel<<<dim3(nblk, ns), 64, 0, stream>>>(F, labels, N, inv_tau, part);
    ODW_CHECK_LAUNCH("supcon_stats_kernel");
    supcon_combine_kernel<<<1, 256, 0, stream>>>(part, ns, w, N, stats, loss);
    ODW_CHECK_LAUNCH("supcon_combine_kernel");
    if (dF) {
        supcon_grad_kernel<<<dim3(nblk, ns), 64, 0, stream>>>(F, labels, w, stats, N, inv_tau,
                                                             grad_scale * inv_tau, dpart);
        ODW_CHECK_LAUNCH("supcon_grad_kernel");
        int total = N * kD / 4;
        supcon_grad_combine<<<(total + 255) / 256, 256, 0, stream>>>(dpart, ns, N, npad, dF);
        ODW_CHECK_LAUNCH("supcon_grad_combine");
    }
    return ODW_OK;
}
