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

// Work distribution.  Panel p owns pw_blocks_of(p) column blocks and gets ceil(blocks / max_run) workgroups, which cut its
// blocks into equal runs: no workgroup crosses into another panel.  (Round 3 cut the panel-major list of all blocks into
// equal runs: every run that straddled two panels -- 18 workgroups at P = 4000, 36 at P = 8000 -- staged a second panel,
// ~5 us in the middle of its run, and finished that much after everybody else: the kernel's tail.)  pw_max_run picks the
// smallest run length whose workgroup count fits the chip.
__device__ __host__ inline int pw_groups(int max_run, int nblk32, int npanel) {
    int n = 0;
    for (int p = 0; p < npanel; ++p) n += (pw_blocks_of(p, nblk32) + max_run - 1) / max_run;
    return n;
}
__host__ inline int pw_max_run(int nblk32, int npanel, int max_groups) {
    // every panel contributes at least one group: with more panels than `max_groups` (P > 224 * 256 rows) no run length
    // fits the chip, and a run of all nblk32 blocks -- one workgroup per panel -- is the fewest groups there are
    if (max_groups < npanel) max_groups = npanel;
    int m = 1;
    while (m < nblk32 && pw_groups(m, nblk32, npanel) > max_groups) ++m;
    return m;
}
__device__ inline void pw_run_of(int g, int max_run, int nblk32, int npanel, int& panel, int& blk, int& run_len) {
    panel = 0; blk = 0; run_len = 0;
    for (int p = 0; p < npanel; ++p) {
        const int nb = pw_blocks_of(p, nblk32), n = (nb + max_run - 1) / max_run;
        if (g < n) {
            const int lo = (int)((long long)nb * g / n), hi = (int)((long long)nb * (g + 1) / n);
            panel = p; blk = p * (kPwPanel / 32) + lo; run_len = hi - lo;
            return;
        }
        g -= n;
    }
}

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
// store flavours of the interior tiles (dbg bits 8 / 16, ODW_PAIRWISE_ST: measurement of what a kernel that leaves no dirty
// lines behind gains at its boundary): plain, nontemporal, or sc1 (write-through, dropped from L2)
typedef float pw_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pw_st1(float* p, float v, int dbg) {
    if (dbg & 8) __builtin_nontemporal_store(v, p);
    else if (dbg & 16) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else *p = v;
}
__device__ __forceinline__ void pw_st4(float* p, const float4 v, int dbg) {
    const pw_f4 x = {v.x, v.y, v.z, v.w};
    if (dbg & 8) __builtin_nontemporal_store(x, reinterpret_cast<pw_f4*>(p));
    else if (dbg & 16) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(x) : "memory");
    else *reinterpret_cast<pw_f4*>(p) = x;
}

// The stores of one finished 32 x 32 tile (rows r0.., columns c0.. of S, c0 >= r0): the direct tile as 128-byte row segments,
// the mirrored tile through the wave's padded LDS scratch so that it leaves as full 128-byte lines too; a diagonal tile
// writes its upper triangle to both places.  Shared by the panel and the DMA kernel.
__device__ __forceinline__ void pw_store_tile(const f32x16& acc, float* __restrict__ S, int P, int ld, int r0, int c0, int lane, int half,
                                              int l31, float* scratch, bool small, bool vec, int dbg = 0) {
    const int col = c0 + l31;
    const bool full = small && vec && r0 + 32 <= P && c0 + 32 <= P;      // wave-uniform: no per-element predicates
    if (c0 == r0) {
        // diagonal tile: (r, c) and (c, r) were accumulated in different term orders -- the upper triangle
        // goes to both places so that S is exactly symmetric
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int r = crow(k, half);
            if (r <= l31 && col < P) {
                S[(size_t)(r0 + r) * ld + col] = acc[k];
                S[(size_t)col * ld + r0 + r] = acc[k];
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
            const unsigned dbase = (unsigned)(r0 + 4 * half) * (unsigned)ld + (unsigned)col;
            if (!PW_DBG(1))
#pragma unroll
            for (int k = 0; k < 16; ++k)      // direct tile: one 128-byte row segment per half-wave
                pw_st1(S + (dbase + (unsigned)((k & 3) + 8 * (k >> 2)) * (unsigned)ld), acc[k], dbg);
            const unsigned mbase = (unsigned)(c0 + (lane >> 3)) * (unsigned)ld + (unsigned)(r0 + (lane & 7) * 4);
            if (!PW_DBG(2))
#pragma unroll
            for (int t = 0; t < 4; ++t)
                pw_st4(S + (mbase + (unsigned)(8 * t) * (unsigned)ld),
                       *reinterpret_cast<const float4*>(scratch + (t * 8 + (lane >> 3)) * kPwTrPitch + (lane & 7) * 4), dbg);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int r = r0 + crow(k, half);
                if (r < P && col < P) S[(size_t)r * ld + col] = acc[k];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int n = t * 8 + (lane >> 3), m4 = (lane & 7) * 4;
                const float4 v = *reinterpret_cast<const float4*>(scratch + n * kPwTrPitch + m4);
                const int row = c0 + n, cc = r0 + m4;
                if (row < P) {
                    float* dst = S + (size_t)row * ld + cc;
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

struct PwRows { float4 a0, a1; };

// Workgroup barrier that orders LDS traffic ONLY (see above).  The stores of S are never read back inside the kernel.
__device__ __forceinline__ void pw_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(kPwWaves * 64, 1) void pairwise_sim_panel_kernel(const float* __restrict__ E, int P,
                                                                              float* __restrict__ S, int ld, int max_run, int dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave == kPwCompute;
    const int nblk32 = (P + 31) / 32, npanel = (P + kPwPanel - 1) / kPwPanel;
    // this workgroup's run of column blocks INSIDE ONE PANEL (pw_run_of: no run straddles two panels)
    int panel, blk, run_len;
    pw_run_of(blockIdx.x, max_run, nblk32, npanel, panel, blk, run_len);
    const long long lo_it = 0, hi_it = run_len;
    if (run_len <= 0) return;
    PW_T(0);
    float* const scratch = reinterpret_cast<float*>(lds + 2 * kPwSlot) + (loader ? 0 : wave) * 32 * kPwTrPitch;
    const bool vec = (ld & 3) == 0;
    const bool small = (unsigned long long)P * (unsigned long long)ld < (1ull << 30);      // 32-bit element offsets

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
    // (Measured and rejected in round 4, tools/pairwise_forms.py with ODW_PAIRWISE_ST: waves 4-6 -- the ones that share a
    // SIMD's matrix pipe with waves 0-2 -- storing their PREVIOUS tile first and running the current tile's MFMAs afterwards,
    // so that one wave of a SIMD owns the pipe while the other sits in its stores: 13.7 / 26.7 / 83-85 us -> 14.6 / 27.4-28.9 /
    // 83 us at P = 2000 / 4000 / 8000.  The block period is set by how fast the chip takes the stores, not by the pipe.)

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
                pw_store_tile(acc, S, P, ld, r0, c0, lane, half, l31, scratch, small, vec, dbg);
            }
        }
        PW_T(4 + 3 * (int)(it - lo_it));
        slot ^= 1;
        blk = b1; panel = p1;
        advance(p1, b1);
    }
    PW_T(31);
}

// ------------------------------------------------------------------ pairwise, planes + LDS-DMA form (round 4)
// The panel kernel above spends its edges and part of its steady state on the fp32 -> 3-plane split: every workgroup
// splits its 224 panel rows (two barrier-separated rounds through LDS, ~4.6 us before its first MFMA) and its loader
// wave splits every 32-row column block again (1.5 us per block, ~9 workgroups splitting the same block) -- the
// timeline of round 3 (profiles/r03/pairwise_timeline_4000.txt).  Here the planes exist ONCE, in HBM/L2 -- written by
// pw_split_planes_kernel (P x 128 values: ~1 us) or handed in by the caller (odw_pairwise_sim_planes: the producer of E
// can emit them) -- as [3 planes][Ppad rows][128] bf16, rows >= P zero, and reach LDS by DMA
// (global_load_lds_dwordx4: no VGPR round trip, no VALU):
//   * the loader wave's work per column block is 24 DMA instructions (3 planes x 32 rows x 256 B) instead of 16 loads +
//     ~350 VALU + 24 LDS stores per lane;
//   * the panel stage is DMA too: round 1 lands four 32-row pieces (and the first column block) while nothing else runs,
//     round 2 the other three -- no split, no ds_write pass;
//   * rows are unpadded 256-byte lines in LDS; the 16-byte chunk c of row r sits at position c ^ (r & 15), applied to the
//     DMA's SOURCE address (the LDS image of a DMA instruction is lane-linear), so that the 16 lanes ds_read_b128 serves
//     together (rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}) hit 16 distinct 4-bank groups.
// Everything after the fragments -- the six plane products, the direct and the mirrored tile, the diagonal rule -- is
// the panel kernel's, so the numbers are bit-identical to it.
// MEASURED (tools/pairwise_forms.py, tools/exp/pairwise_timeline.bin <P> dma; profiles/r04/pairwise_forms.txt): NOT faster.
// The block period is the same 2.9-3.0 us (it is the compute waves' own chain -- 1.43 us of MFMAs on a pipe two waves share,
// 0.95 us issuing the tile's stores into a store-bound chip, 0.5 us at the barrier -- and never was the loader), the loader's 24
// DMA instructions take 1.6 us to land (~15 GB/s for one wave), and the DMA panel stage takes 6.6 us against 4.6 (192 KB per
// workgroup at the ~30-60 GB/s per CU LDS-DMA sustains).  With the split kernel and its launch boundary in front
// (odw_pairwise_sim_ws) it is 5-7 us slower than the one-launch form at every size; on caller-provided planes it ties at
// P <= 6000 and is within box-to-box noise at P = 8000.  It stays as an explicit entry point (odw_pairwise_sim_planes) for a
// caller that already has the planes; odw_pairwise_sim_ws keeps the one-launch panel kernel (ODW_PAIRWISE_PLANES_MIN=0 forces
// this form for comparison).
// Round 6 re-measured both forms THROUGH odw_pairwise_sim_ws with the threshold forced either way (tools/exp/pairwise_min_ab.py,
// profiles/r06/pairwise_min_ab.txt; live HIP events over graph-replayed launches, as bench.py's roofline.kernels): panel / split + DMA
// = 12.1 / 18.5 us at P = 2000, 22.6 / 31.0 at 4000, 41.0 / 48.8 at 5600, 44.6 / 52.7 at 6000, 79.7 / 86.0 at 8000 -- the one-launch
// panel kernel is ahead at EVERY size.  Round 5 had switched the default to the pair from P = 5600 on figures taken from a table
// (profiles/r04/pairwise_forms.txt) whose `ws_us` column never ran this form; the driver's P = 8000 went 80.7 -> 89.5 us.  Back to
// never (ODW_PAIRWISE_PLANES_MIN=<P> still forces it for comparison).
constexpr int kPwPlanesMinP = 1 << 30;
constexpr int kPdSlot = 3 * 32 * 256;                // one 32-row block: 3 planes x 32 rows x 256 B (no padding)
constexpr int kPdAreas = 3;                          // panel staging areas besides column-block slot 1
constexpr int kPdLds = 2 * kPdSlot + kPdAreas * kPdSlot;           // 49152 + 73728 (the transpose scratch overlaps area 0/1)
static_assert(kPwScratch <= 2 * kPdSlot, "the compute waves' transpose scratch must fit in the first two staging areas");

__device__ __forceinline__ int pd_pad(int P) { return (P + 31) / 32 * 32; }

// E (P x 128 fp32) -> planes [3][Ppad][128] bf16 (rows P .. Ppad-1 zero)
__global__ __launch_bounds__(256) void pw_split_planes_kernel(const float* __restrict__ E, int P, int Ppad,
                                                              unsigned char* __restrict__ planes) {
    const size_t plane_bytes = (size_t)Ppad * 256;
    const int items = Ppad * 16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < items; i += gridDim.x * blockDim.x) {
        const int row = i >> 4, chunk = i & 15;
        uint4 h = make_uint4(0, 0, 0, 0), m = h, l = h;
        if (row < P) {
            const float4* src = reinterpret_cast<const float4*>(E + (size_t)row * kD + chunk * 8);
            ps_split8(src[0], src[1], h, m, l);
        }
        unsigned char* dst = planes + (size_t)row * 256 + chunk * 16;
        *reinterpret_cast<uint4*>(dst) = h;
        *reinterpret_cast<uint4*>(dst + plane_bytes) = m;
        *reinterpret_cast<uint4*>(dst + 2 * plane_bytes) = l;
    }
}

typedef __attribute__((address_space(3))) void pd_lds_t;
typedef __attribute__((address_space(1))) const void pd_gbl_t;

// DMA instruction `inst` (0..23) of the 32-row piece starting at plane row `row0` (a multiple of 32, clamped as a whole to
// the last piece: its rows are then other rows' values, which is fine -- products of rows past P are never stored): plane
// inst / 8, rows 4 (inst % 8) .. +3.  Lane (lr = lane >> 4, c' = lane & 15) fills LDS position c' of row r = 4 (inst % 8) + lr
// with logical chunk c' ^ (r & 15) = c' ^ lr ^ 4 (inst & 3): the per-lane part of the source address takes four values
// (voff[inst & 3]), everything else is wave-uniform.
__device__ __forceinline__ void pd_dma(const unsigned char* __restrict__ planes, size_t plane_bytes, int row0, int last_piece,
                                       unsigned char* area, int inst, const unsigned (&voff)[4]) {
    const int plane = inst >> 3, sub = inst & 7;
    const int r0 = row0 < last_piece ? row0 : last_piece;
    const unsigned char* base = planes + plane * plane_bytes + (size_t)(r0 + 4 * sub) * 256;
    __builtin_amdgcn_global_load_lds((pd_gbl_t*)(base + voff[inst & 3]), (pd_lds_t*)(area + plane * (32 * 256) + sub * 1024), 16, 0, 0);
}

__global__ __launch_bounds__(kPwWaves * 64, 1) void pairwise_sim_dma_kernel(const unsigned char* __restrict__ planes, int P,
                                                                            float* __restrict__ S, int max_run) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave == kPwCompute;
    const int nblk32 = (P + 31) / 32, npanel = (P + kPwPanel - 1) / kPwPanel;
    const int Ppad = nblk32 * 32;
    const size_t plane_bytes = (size_t)Ppad * 256;
    // this workgroup's run of column blocks INSIDE ONE PANEL (pw_run_of: no run straddles two panels)
    int panel, blk, run_len;
    pw_run_of(blockIdx.x, max_run, nblk32, npanel, panel, blk, run_len);
    const long long lo_it = 0, hi_it = run_len;
    if (run_len <= 0) return;
    PW_T(0);
    unsigned char* const slot0 = lds;
    unsigned char* const slot1 = lds + kPdSlot;
    unsigned char* const areas = lds + 2 * kPdSlot;
    float* const scratch = reinterpret_cast<float*>(areas) + (loader ? 0 : wave) * 32 * kPwTrPitch;
    const bool vec = (P & 3) == 0;
    const bool small = (unsigned long long)P * (unsigned long long)P < (1ull << 30);
    // fragment position of this lane inside a piece: row l31, logical chunk 2 kk + half at position (2 kk + half) ^ (l31 & 15)
    const int frag_row = l31 * 256, frag_x = l31 & 15;
    unsigned voff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) voff[q] = (unsigned)((lane >> 4) * 256 + (((lane & 15) ^ (lane >> 4) ^ (4 * q)) * 16));
    const int last_piece = Ppad - 32;
    auto advance = [&](int& pn, int& b) { if (++b >= nblk32) { ++pn; b = pn * (kPwPanel / 32); } };
    // staging area j of the panel stage: 0 .. kPdAreas-1 behind the slots, kPdAreas = column-block slot 1
    auto area_of = [&](int j) { return j < kPdAreas ? areas + j * kPdSlot : slot1; };
    constexpr int kRound = kPdAreas + 1;                 // pieces per round of the panel stage

    bf16x8 aH[8], aM[8], aL[8];
    int have_panel = -1, slot = 0;
    int p1 = panel, b1 = blk;
    advance(p1, b1);
    bool parked = false;

    for (long long it = lo_it; it < hi_it; ++it) {
        if (panel != have_panel) {
            if (parked) pw_barrier();                    // every wave is done reading the slots and its scratch
            // ---- panel stage: this workgroup's 224 rows as seven 32-row pieces, kRound per round, DMA'd by all eight
            // waves; the first column block goes to slot 0 with round 1.  (No stores are in flight in a wave that gets
            // here for the first time; a wave that changes panels waits for its tile stores once per panel.)
#pragma unroll 1
            for (int hp = 0; hp < (kPwCompute + kRound - 1) / kRound; ++hp) {
                const int npieces = min(kRound, kPwCompute - hp * kRound);
                const int njobs = (npieces + (hp == 0 ? 1 : 0)) * 24;
                for (int j = wave; j < njobs; j += kPwWaves) {
                    const int pc = j / 24, inst = j - pc * 24;
                    if (pc < npieces) pd_dma(planes, plane_bytes, panel * kPwPanel + (hp * kRound + pc) * 32, last_piece, area_of(pc), inst, voff);
                    else pd_dma(planes, plane_bytes, blk * 32, last_piece, slot0, inst, voff);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                pw_barrier();
                if (!loader && wave / kRound == hp) {
                    const unsigned char* sa = area_of(wave % kRound) + frag_row;
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        const int pos = ((2 * kk + half) ^ frag_x) * 16;
                        aH[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sa + pos));
                        aM[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sa + 32 * 256 + pos));
                        aL[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sa + 64 * 256 + pos));
                    }
                }
                pw_barrier();
            }
            slot = 0;
            have_panel = panel;
            parked = true;
            PW_T(1);
        } else {
            pw_barrier();                                // block `blk` has landed in `slot`; the other slot is free
        }
        PW_T(2 + 3 * (int)(it - lo_it));
        unsigned char* const cur = slot ? slot1 : slot0;
        unsigned char* const nxt = slot ? slot0 : slot1;
        if (loader) {
            // ---- the loader wave: the next block's 24 KB by DMA into the other slot (unconditional: past the end of the
            // run the slot receives rows nobody reads); it waits for them to land before the barrier that publishes them
            const int nb = min(b1, nblk32 - 1) * 32;
#pragma unroll 4
            for (int inst = 0; inst < 24; ++inst) pd_dma(planes, plane_bytes, nb, last_piece, nxt, inst, voff);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PW_T(3 + 3 * (int)(it - lo_it));
        } else {
            const int r0 = panel * kPwPanel + wave * 32, c0 = blk * 32;
            if (c0 >= r0 && r0 < P) {                    // this wave's tile lies on or above the diagonal
                const unsigned char* sb = cur + frag_row;
                f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const int pos = ((2 * kk + half) ^ frag_x) * 16;
                    const bf16x8 bH = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + pos));
                    const bf16x8 bM = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + 32 * 256 + pos));
                    const bf16x8 bL = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + 64 * 256 + pos));
                    // (the same term order as the panel kernel: bit-identical results)
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
                pw_store_tile(acc, S, P, P, r0, c0, lane, half, l31, scratch, small, vec);
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
__host__ __device__ inline int supcon_nsplit(int N) {
    const int nblk = (N + 31) / 32;
    int s = 1024 / (nblk > 0 ? nblk : 1);
    if (s > 16) s = 16;
    if (s > nblk) s = nblk;
    if (s < 1) s = 1;
    return s;
}
// Device-resident N (round 6, loss_lists.hip): the launch is sized for the capacity, N and the split count derived from it
// are read on the device -- the same values, the same summation order as the static launch of that N.
#define ODW_SUPCON_DYN_N()                                                  \
    int nsplit = gridDim.y;                                                 \
    if (n_dev) {                                                            \
        const int nd_ = *n_dev;                                             \
        N = nd_ < N ? nd_ : N;                                              \
        nsplit = supcon_nsplit(N);                                          \
        if ((int)blockIdx.y >= nsplit || (int)blockIdx.x * 32 >= N) return; \
    }
// partial statistics of rows i over the j blocks of one split.
// part layout: [nsplit][3][Npad], Npad = 32*ceil(N/32)
__global__ __launch_bounds__(64) void supcon_stats_kernel(const float* __restrict__ F,
                                                          const int* __restrict__ labels, int N,
                                                          float inv_tau, float* __restrict__ part,
                                                          const int* __restrict__ n_dev) {
    ODW_SUPCON_DYN_N();
    const int lane = threadIdx.x, half = lane >> 5, c = lane & 31;
    const int nblk = (N + 31) / 32, npad = nblk * 32;
    const int I = blockIdx.x * 32, i = I + c;
    const int sp = blockIdx.y;
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
// One workgroup (the mean is one ordered sum) of 1024 threads: a row's 3 x nsplit partials are loaded together before any of
// them is used -- with 256 threads and a load-use chain per split the kernel was 51 us of pure L2 latency at N ~ 1500.
constexpr int kCombineThreads = 1024, kMaxSplit = 16;
__global__ __launch_bounds__(kCombineThreads) void supcon_combine_kernel(const float* __restrict__ part, int nsplit,
                                                                         const float* __restrict__ w, int N,
                                                                         float* __restrict__ stats,
                                                                         float* __restrict__ loss,
                                                                         const int* __restrict__ n_dev) {
    if (n_dev) { const int nd_ = *n_dev; N = nd_ < N ? nd_ : N; nsplit = supcon_nsplit(N); }
    const int npad = ((N + 31) / 32) * 32;
    __shared__ float red[kCombineThreads];
    float local = 0.0f;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        float pm[kMaxSplit], pa[kMaxSplit], pb[kMaxSplit];
#pragma unroll
        for (int s = 0; s < kMaxSplit; ++s) {
            const float* p = part + (size_t)(s < nsplit ? s : 0) * 3 * npad;
            pm[s] = p[i]; pa[s] = p[npad + i]; pb[s] = p[2 * npad + i];
        }
        const float wi = w[i];
        float M = -__builtin_inff();
#pragma unroll
        for (int s = 0; s < kMaxSplit; ++s) if (s < nsplit) M = fmaxf(M, pm[s]);
        float A = 0.0f, B = 0.0f;
#pragma unroll
        for (int s = 0; s < kMaxSplit; ++s) {
            if (s < nsplit) {
                const float sc = (pm[s] == -__builtin_inff()) ? 0.0f : expf(pm[s] - M);
                A += pa[s] * sc;
                B += pb[s] * sc;
            }
        }
        stats[i] = M; stats[npad + i] = A; stats[2 * npad + i] = B;
        local += -logf(A / B) * wi;   // sim_loss.py:76-78
    }
    red[threadIdx.x] = local;
    __syncthreads();
    for (int off = kCombineThreads / 2; off > 0; off >>= 1) {
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
                                                         float* __restrict__ dpart, const int* __restrict__ n_dev) {
    ODW_SUPCON_DYN_N();
    const int lane = threadIdx.x, half = lane >> 5, c = lane & 31;
    const int nblk = (N + 31) / 32, npad = nblk * 32;
    const int I = blockIdx.x * 32, i = I + c;
    const int sp = blockIdx.y;
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
                                    float* __restrict__ dF, const int* __restrict__ n_dev) {
    if (n_dev) { const int nd_ = *n_dev; N = nd_ < N ? nd_ : N; nsplit = supcon_nsplit(N); npad = ((N + 31) / 32) * 32; }
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


}  // namespace

// The smallest P at which odw_pairwise_sim_ws uses its workspace (below it the one-launch panel kernel runs and the
// workspace may be NULL): callers allocate odw_pairwise_sim_workspace(P, D) bytes only from here on.
ODW_EXPORT int odw_pairwise_sim_planes_min(void) {
    return getenv("ODW_PAIRWISE_PLANES_MIN") ? atoi(getenv("ODW_PAIRWISE_PLANES_MIN")) : kPwPlanesMinP;
}

ODW_EXPORT int64_t odw_pairwise_sim_workspace(int P, int D) {
    // the three bf16 planes of E, rows padded to a multiple of 32 (zero rows): [3][Ppad][128]
    return D == kD && P > 0 ? odw_align_up((int64_t)((P + 31) / 32 * 32) * 384 * 2, 256) : 0;
}


// E (P x 128 fp32, 16-byte aligned) -> the plane form odw_pairwise_sim_planes reads: [3 planes][Ppad][128] bf16
// (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): hi + mid + lo == x exactly), Ppad = P rounded up to 32, the
// padding rows zero.  `planes` holds odw_pairwise_sim_workspace(P, 128) bytes.
ODW_EXPORT int odw_pairwise_split_planes(const float* E, int P, void* planes, void* stream_) {
    ODW_REQUIRE(P >= 0, "pairwise_split_planes: bad P=%d", P);
    if (P == 0) return ODW_OK;
    ODW_REQUIRE(E && planes, "pairwise_split_planes: null pointer");
    ODW_REQUIRE((((uintptr_t)E) & 15) == 0 && (((uintptr_t)planes) & 15) == 0, "pairwise_split_planes: buffers must be 16-byte aligned");
    const int Ppad = (P + 31) / 32 * 32;
    const int blocks = (Ppad * 16 + 255) / 256;
    pw_split_planes_kernel<<<blocks < 2048 ? blocks : 2048, 256, 0, (hipStream_t)stream_>>>(E, P, Ppad, (unsigned char*)planes);
    ODW_CHECK_LAUNCH("pw_split_planes_kernel");
    return ODW_OK;
}

// S = E E^T (P x P fp32, fp32-grade: six bf16 plane products of order <= 2 per element) from the PLANES of E -- the form for
// a caller whose producer of E emits them (the split then costs no launch and no pass over E).  Reference:
// roi_heads/weak_head/loss.py:319 (sim_mat = torch.mm(sim_feature, sim_feature.T)).  S must be 16-byte aligned.
ODW_EXPORT int odw_pairwise_sim_planes(const void* planes, int P, float* S, void* stream_) {
    ODW_REQUIRE(P >= 0, "pairwise_sim_planes: bad P=%d", P);
    if (P == 0) return ODW_OK;
    ODW_REQUIRE(planes && S, "pairwise_sim_planes: null pointer");
    ODW_REQUIRE((((uintptr_t)planes) & 15) == 0 && (((uintptr_t)S) & 15) == 0, "pairwise_sim_planes: buffers must be 16-byte aligned");
    const int nblk32 = (P + 31) / 32, npanel = (P + kPwPanel - 1) / kPwPanel;
    const int max_run = pw_max_run(nblk32, npanel, ODW_NUM_CU), grid = pw_groups(max_run, nblk32, npanel);
    const hipError_t attr = odw_set_max_lds(reinterpret_cast<const void*>(pairwise_sim_dma_kernel),
                                                       kPdLds);      // once
    ODW_CHECK_HIP(attr, "pairwise dma attr");
    pairwise_sim_dma_kernel<<<grid, kPwWaves * 64, kPdLds, (hipStream_t)stream_>>>((const unsigned char*)planes, P, S, max_run);
    ODW_CHECK_LAUNCH("pairwise_sim_dma_kernel");
    return ODW_OK;
}

// S with a row pitch of ldS elements (>= P).  A pitch that is not a multiple of 16 floats makes every 128-byte row segment
// of a tile straddle two cache lines: the stores then have to merge in L2 (plain stores, below) and the kernel is ~15%
// slower than with whole-line nontemporal stores (P = 5000: 33.9 us dense, 29.5 us with ldS = 5024;
// tools/exp/pairwise_psweep.py): callers that may choose the layout pass ldS = P rounded up to 32.
static int pairwise_sim_impl(const float* E, int P, int D, float* S, int64_t ldS, void* workspace, int64_t workspace_bytes,
                             void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(P >= 0 && D > 0 && D % 4 == 0, "pairwise_sim: bad dims P=%d D=%d", P, D);
    ODW_REQUIRE(ldS >= P && ldS <= INT32_MAX, "pairwise_sim: row pitch %lld < P=%d", (long long)ldS, P);
    ODW_REQUIRE(ldS == P || (D == kD && (((uintptr_t)S) & 15) == 0), "pairwise_sim: a padded row pitch needs D=%d and a 16-byte aligned S", kD);
    if (P == 0) return ODW_OK;
    ODW_REQUIRE(E && S, "pairwise_sim: null pointer");
    ODW_REQUIRE((((uintptr_t)E) & 15) == 0, "pairwise_sim: E must be 16-byte aligned");
    static const bool fp32_chain = getenv("ODW_PAIRWISE_FP32") != nullptr;      // force the exact-fp32 MFMA form (comparison)
    // planes + DMA form from this many rows on, when the caller gave the workspace for the planes (ODW_PAIRWISE_PLANES_MIN:
    // comparison runs; a huge value = always the one-launch panel kernel)
    static const int planes_min = getenv("ODW_PAIRWISE_PLANES_MIN") ? atoi(getenv("ODW_PAIRWISE_PLANES_MIN")) : kPwPlanesMinP;
    if (D == kD && !fp32_chain && ldS == P && (((uintptr_t)S) & 15) == 0 && P >= planes_min && workspace &&
        (((uintptr_t)workspace) & 15) == 0 && workspace_bytes >= odw_pairwise_sim_workspace(P, D)) {
        const int rc = odw_pairwise_split_planes(E, P, workspace, stream_);
        if (rc != ODW_OK) return rc;
        return odw_pairwise_sim_planes(workspace, P, S, stream_);
    }
    if (D == kD && (ldS != P || (!fp32_chain && (((uintptr_t)S) & 15) == 0))) {
        // split-bf16 panel form: one launch, no workspace (the planes are made in registers)
        const int nblk32 = (P + 31) / 32, npanel = (P + kPwPanel - 1) / kPwPanel;
        const int max_run = pw_max_run(nblk32, npanel, ODW_NUM_CU), grid = pw_groups(max_run, nblk32, npanel);
        const hipError_t attr = odw_set_max_lds(reinterpret_cast<const void*>(pairwise_sim_panel_kernel),
                                                           kPwLds);      // once
        ODW_CHECK_HIP(attr, "pairwise attr");
#ifdef ODW_EXPERIMENTS
        static const int dbg = getenv("ODW_PAIRWISE_DBG") ? atoi(getenv("ODW_PAIRWISE_DBG")) : 0;
#else
        // interior tiles leave as NONTEMPORAL stores (bit 8; ODW_PAIRWISE_ST=0: plain, 16: sc1 write-through): S is written once
        // and not read back by this kernel -- 27.5 -> 24.5 us at P = 4000, 13.7 -> 12.9 at P = 2000 on the same box
        // (tools/pairwise_forms.py, alternating runs)
        // Nontemporal stores do not merge in L2: with a row pitch that is not a multiple of 16 floats every 128-byte segment
        // straddles two lines, the halves go out as partial-line writes and the kernel takes TWICE as long (P = 5000 dense:
        // 58.3 us nontemporal, 33.9 us plain; rows padded to 5024 floats: 29.5 us nontemporal) -- plain stores there.
        static const int st = getenv("ODW_PAIRWISE_ST") ? (atoi(getenv("ODW_PAIRWISE_ST")) & 24) : -1;
        const int dbg = st >= 0 ? st : (ldS % 16 == 0 ? 8 : 0);
#endif
        pairwise_sim_panel_kernel<<<grid, kPwWaves * 64, kPwLds, stream>>>(E, P, S, (int)ldS, max_run, dbg);
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

ODW_EXPORT int odw_pairwise_sim_ws(const float* E, int P, int D, float* S, void* workspace, int64_t workspace_bytes,
                                   void* stream_) {
    return pairwise_sim_impl(E, P, D, S, P, workspace, workspace_bytes, stream_);
}

ODW_EXPORT int odw_pairwise_sim(const float* E, int P, int D, float* S, void* stream_) {
    return pairwise_sim_impl(E, P, D, S, P, nullptr, 0, stream_);          // (the panel kernel needs no workspace)
}

ODW_EXPORT int odw_pairwise_sim_ld(const float* E, int P, int D, float* S, int64_t ldS, void* stream_) {
    return pairwise_sim_impl(E, P, D, S, ldS, nullptr, 0, stream_);
}

ODW_EXPORT int64_t odw_supcon_workspace(int N) {
    if (N < 1) N = 1;
    const int64_t npad = ((N + 31) / 32) * 32;
    const int64_t ns = supcon_nsplit(N);
    return odw_align_up(3 * npad * 4, 256) + odw_align_up(ns * 3 * npad * 4, 256) +
           odw_align_up(ns * npad * kD * 4, 256);
}

static int supcon_launch(const float* F, const int32_t* labels, const float* w, int N, int D, float tau,
                         float grad_scale, float* loss, float* dF, void* workspace,
                         int64_t workspace_bytes, void* stream_, const int* n_dev);

ODW_EXPORT int odw_supcon_v2(const float* F, const int32_t* labels, const float* w, int N, int D, float tau,
                             float grad_scale, float* loss, float* dF, void* workspace,
                             int64_t workspace_bytes, void* stream_) {
    return supcon_launch(F, labels, w, N, D, tau, grad_scale, loss, dF, workspace, workspace_bytes, stream_, nullptr);
}

// SupConLossV2 over the first *n_dev rows of (F, labels, w); N_cap rows exist and size the workspace
// (odw_supcon_workspace(N_cap)) and the launch.  Same arithmetic, same order of sums as odw_supcon_v2 at N = *n_dev.
ODW_EXPORT int64_t odw_supcon_dyn_workspace(int N_cap) {
    if (N_cap < 1) N_cap = 1;
    const int64_t npad = ((N_cap + 31) / 32) * 32;
    return odw_align_up(3 * npad * 4, 256) + odw_align_up(16 * 3 * npad * 4, 256) + odw_align_up(16 * npad * kD * 4, 256);
}

ODW_EXPORT int odw_supcon_v2_dyn(const float* F, const int32_t* labels, const float* w, int N_cap, int D, float tau,
                                 float grad_scale, float* loss, float* dF, const int* n_dev, void* workspace,
                                 int64_t workspace_bytes, void* stream_) {
    ODW_REQUIRE(n_dev, "supcon_v2_dyn: n_dev is null");
    return supcon_launch(F, labels, w, N_cap, D, tau, grad_scale, loss, dF, workspace, workspace_bytes, stream_, n_dev);
}

static int supcon_launch(const float* F, const int32_t* labels, const float* w, int N, int D, float tau,
                         float grad_scale, float* loss, float* dF, void* workspace,
                         int64_t workspace_bytes, void* stream_, const int* n_dev) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(N >= 1, "supcon_v2: N=%d (the reference would take the mean of an empty tensor)", N);
    ODW_REQUIRE(D == kD, "supcon_v2: D=%d unsupported (Sim_Net emits 128-d embeddings)", D);
    ODW_REQUIRE(tau > 0.0f, "supcon_v2: temperature must be > 0");
    ODW_REQUIRE(F && labels && w && loss, "supcon_v2: null pointer");
    ODW_REQUIRE((((uintptr_t)F) & 15) == 0 && (dF == nullptr || (((uintptr_t)dF) & 15) == 0),
                "supcon_v2: F/dF must be 16-byte aligned");
    const int64_t need = n_dev ? odw_supcon_dyn_workspace(N) : odw_supcon_workspace(N);
    if (!workspace || workspace_bytes < need) {
        odw_set_error("supcon_v2: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
        return ODW_EWORKSPACE;
    }
    const int nblk = (N + 31) / 32, npad = nblk * 32;
    // (device-resident N: the regions are carved for the capacity and 16 splits; the kernels index them with the npad and
    // the split count of the live N -- they all derive both from *n_dev the same way)
    const int ns = n_dev ? 16 : supcon_nsplit(N);
    unsigned char* p = (unsigned char*)workspace;
    float* stats = (float*)p; p += odw_align_up((int64_t)3 * npad * 4, 256);
    float* part = (float*)p;  p += odw_align_up((int64_t)ns * 3 * npad * 4, 256);
    float* dpart = (float*)p;
    const float inv_tau = 1.0f / tau;
    supcon_stats_kernel<<<dim3(nblk, ns), 64, 0, stream>>>(F, labels, N, inv_tau, part, n_dev);
    ODW_CHECK_LAUNCH("supcon_stats_kernel");
    static_assert(kMaxSplit == 16, "supcon_nsplit caps the splits at 16");
    supcon_combine_kernel<<<1, kCombineThreads, 0, stream>>>(part, ns, w, N, stats, loss, n_dev);
    ODW_CHECK_LAUNCH("supcon_combine_kernel");
    if (dF) {
        supcon_grad_kernel<<<dim3(nblk, ns), 64, 0, stream>>>(F, labels, w, stats, N, inv_tau,
                                                             grad_scale * inv_tau, dpart, n_dev);
        ODW_CHECK_LAUNCH("supcon_grad_kernel");
        int total = N * kD / 4;
        supcon_grad_combine<<<(total + 255) / 256, 256, 0, stream>>>(dpart, ns, N, npad, dF, n_dev);
        ODW_CHECK_LAUNCH("supcon_grad_combine");
    }
    return ODW_OK;
}
