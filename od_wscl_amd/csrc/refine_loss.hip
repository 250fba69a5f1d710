// refine_loss.hip -- the dense part of the OD-WSCL loss as two launches per step, forward AND backward:
//   wsddn_scores : final = softmax_row(cls) * softmax_col-per-image(det), softmax(ref1), softmax(ref2)
//                  (roi_heads/weak_head/loss.py:234-247,357) -- the score matrices the selection kernels read
//   refine_losses: MIL image loss (BCE of the clamped column sums, loss.py:353-354), 3 x weighted
//                  cross-entropy (loss.py:375-377), 3 x smooth-L1 box regression on the pseudo-positive
//                  rows (loss.py:380-394), the 4 top-k accuracies (loss.py:25-33,396-400), and the
//                  gradient of the sum of the 7 losses w.r.t. the fused predictor output Y (P x 5C+12C).
// The reference runs ~80 tiny kernels forward and ~80 backward through autograd for this; here the rows are
// spread over many workgroups (a 32-lane group per proposal, classes across lanes -> coalesced rows,
// shuffle reductions) with per-workgroup partials combined by a tiny finishing kernel (deterministic).  Closed-form gradients:
//   g_c = dL/dphi_c (0 where phi was clamped);  ddet_rc = ds_rc g_c (cs_rc - colsum_c);
//   dcls_rc = cs_rc (g_c ds_rc - sum_c' g_c' final_rc');  dref = lam w (softmax - onehot)/P;
//   dbbox = lam w clamp(diff,-1,1)/P on the 4 columns of the pseudo label.
#include "odw_common.h"

namespace {

constexpr int kMaxC = 128;
constexpr int kVPL = kMaxC / 32;            // classes per lane in a 32-lane row group
constexpr int kRowThreads = 256;            // 8 row groups per workgroup
constexpr int kGroups = kRowThreads / 32;
constexpr int kMaxRowBlocks = 64;           // workgroups per image in the row-parallel kernels

struct Heads {        // column offsets inside one row of Y (predictor order: roi_weak_predictors.py:158-165)
    int cls, det, ref[3], box[3];
    int C, ldy;
};

__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 32));
    return v;
}
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 32);
    return v;
}

// softmax of one row held as x[v] = logits[lane + 32 v] by a 32-lane group -> p[v]; returns (max, log-sum-exp)
__device__ __forceinline__ void group_softmax(const float (&x)[kVPL], int C, int lane, float (&p)[kVPL], float& m,
                                              float& lse) {
    m = -__builtin_inff();
#pragma unroll
    for (int v = 0; v < kVPL; ++v)
        if (lane + 32 * v < C) m = fmaxf(m, x[v]);
    m = group_max(m);
    float s = 0.0f;
#pragma unroll
    for (int v = 0; v < kVPL; ++v) {
        p[v] = (lane + 32 * v < C) ? expf(x[v] - m) : 0.0f;
        s += p[v];
    }
    s = group_sum(s);
    lse = logf(s);
#pragma unroll
    for (int v = 0; v < kVPL; ++v) p[v] /= s;
}

// ---- det column statistics: one workgroup per image, lanes over columns (coalesced rows)
__global__ __launch_bounds__(1024) void det_colstats_kernel(const float* __restrict__ Y, Heads h,
                                                            const int* __restrict__ img_off, float* __restrict__ colstat) {
    __shared__ float part[1024];
    __shared__ float cmax[kMaxC];
    const int img = blockIdx.x, base = img_off[img], P = img_off[img + 1] - base, C = h.C;
    const int CL = C <= 32 ? 32 : (C <= 64 ? 64 : 128);
    const int c = threadIdx.x % CL, sl = threadIdx.x / CL, S = 1024 / CL;
    const float* y = Y + (size_t)base * h.ldy + h.det;
    // (eight rows of loads in flight per thread: a row per iteration was 2 x 63 dependent L2 round trips = 27 us for one
    //  workgroup; the running maximum / sum still take the rows in ascending order)
    constexpr int kFly = 8;
    float m = -__builtin_inff();
    if (c < C) {
        int r = sl;
        for (; r + (kFly - 1) * S < P; r += kFly * S) {
            float v[kFly];
#pragma unroll
            for (int u = 0; u < kFly; ++u) v[u] = y[(size_t)(r + u * S) * h.ldy + c];
#pragma unroll
            for (int u = 0; u < kFly; ++u) m = fmaxf(m, v[u]);
        }
        for (; r < P; r += S) m = fmaxf(m, y[(size_t)r * h.ldy + c]);
    }
    part[threadIdx.x] = m;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float a = -__builtin_inff();
        for (int k = 0; k < S; ++k) a = fmaxf(a, part[k * CL + threadIdx.x]);
        cmax[threadIdx.x] = a;
    }
    __syncthreads();
    float sum = 0.0f;
    if (c < C) {
        const float cm = cmax[c];
        int r = sl;
        for (; r + (kFly - 1) * S < P; r += kFly * S) {
            float v[kFly];
#pragma unroll
            for (int u = 0; u < kFly; ++u) v[u] = y[(size_t)(r + u * S) * h.ldy + c];
#pragma unroll
            for (int u = 0; u < kFly; ++u) sum += expf(v[u] - cm);
        }
        for (; r < P; r += S) sum += expf(y[(size_t)r * h.ldy + c] - cm);
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float a = 0.0f;
        for (int k = 0; k < S; ++k) a += part[k * CL + threadIdx.x];
        float* cs = colstat + (size_t)img * 3 * kMaxC;
        cs[threadIdx.x] = cmax[threadIdx.x];
        cs[kMaxC + threadIdx.x] = a;
    }
}

// ---- scores, row parallel: grid (blocks per image, n_img); a 32-lane group per proposal
__global__ __launch_bounds__(kRowThreads) void scores_rows_kernel(const float* __restrict__ Y, Heads h,
                                                                  const int* __restrict__ img_off,
                                                                  const float* __restrict__ colstat,
                                                                  float* __restrict__ final_s, float* __restrict__ src1,
                                                                  float* __restrict__ src2, float* __restrict__ fpart) {
    __shared__ float red[kGroups][kMaxC];
    const int img = blockIdx.y, base = img_off[img], P = img_off[img + 1] - base, C = h.C;
    const int grp = threadIdx.x >> 5, lane = threadIdx.x & 31;
    const float* cst = colstat + (size_t)img * 3 * kMaxC;
    float colacc[kVPL] = {0, 0, 0, 0};
    for (int r = blockIdx.x * kGroups + grp; r < P; r += gridDim.x * kGroups) {
        const float* row = Y + (size_t)(base + r) * h.ldy;
        float x[kVPL], p[kVPL], m, lse;
#pragma unroll
        for (int v = 0; v < kVPL; ++v) x[v] = (lane + 32 * v < C) ? row[h.cls + lane + 32 * v] : 0.0f;
        group_softmax(x, C, lane, p, m, lse);
#pragma unroll
        for (int v = 0; v < kVPL; ++v) {
            const int c = lane + 32 * v;
            if (c < C) {
                const float ds = expf(row[h.det + c] - cst[c]) / cst[kMaxC + c];
                const float f = p[v] * ds;
                final_s[(size_t)(base + r) * C + c] = f;
                colacc[v] += f;
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int v = 0; v < kVPL; ++v) x[v] = (lane + 32 * v < C) ? row[h.ref[k] + lane + 32 * v] : 0.0f;
            group_softmax(x, C, lane, p, m, lse);
            float* dst = (k == 0 ? src1 : src2) + (size_t)(base + r) * C;
#pragma unroll
            for (int v = 0; v < kVPL; ++v)
                if (lane + 32 * v < C) dst[lane + 32 * v] = p[v];
        }
    }
#pragma unroll
    for (int v = 0; v < kVPL; ++v) red[grp][lane + 32 * v] = colacc[v];
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float a = 0.0f;
        for (int g = 0; g < kGroups; ++g) a += red[g][threadIdx.x];
        fpart[((size_t)img * gridDim.x + blockIdx.x) * kMaxC + threadIdx.x] = a;
    }
}

// sum of n values `stride` floats apart, added in ascending order (the bits of the plain loop), eight loads in flight:
// the finish kernels are one small workgroup per image walking <= 64 partials -- a load-use chain per partial otherwise
__device__ __forceinline__ float ordered_sum(const float* __restrict__ p, int n, size_t stride) {
    float a = 0.0f;
    int b = 0;
    for (; b + 8 <= n; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(b + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; b < n; ++b) a += p[(size_t)b * stride];
    return a;
}

__global__ void scores_finish_kernel(const float* __restrict__ fpart, int nblocks, int C, float* __restrict__ colstat) {
    const int img = blockIdx.x, c = threadIdx.x;
    if (c >= C) return;
    const float a = ordered_sum(fpart + (size_t)img * nblocks * kMaxC + c, nblocks, kMaxC);
    colstat[(size_t)img * 3 * kMaxC + 2 * kMaxC + c] = a;       // column sums of final_score (unclamped)
}

// top-k accuracy (loss.py:25-33): mean of labels at the k largest scores (first maximum on ties)
__device__ float topk_acc(const float* score, const float* label, int n, int k) {
    unsigned long long taken[2] = {0, 0};
    float hit = 0.0f;
    for (int t = 0; t < k; ++t) {
        int bi = -1;
        float bv = -__builtin_inff();
        for (int c = 0; c < n; ++c)
            if (!((taken[c >> 6] >> (c & 63)) & 1ull) && score[c] > bv) { bv = score[c]; bi = c; }
        if (bi < 0) break;
        taken[bi >> 6] |= 1ull << (bi & 63);
        hit += label[bi];
    }
    return hit / (float)k;
}

__device__ __forceinline__ float phi_grad(float s, float y, float eps, int C, float inv_img) {
    const float phi = fminf(fmaxf(s, eps), 1.0f - eps);
    const bool inside = s >= eps && s <= 1.0f - eps;           // clamp passes the gradient inside [min,max]
    return inside ? (-(y / phi) + (1.0f - y) / (1.0f - phi)) / (float)C * inv_img : 0.0f;
}

// ---- losses + gradient, row parallel.  part: [n_img][blocks][8 + 3*kMaxC]: 6 loss sums, then ref column sums
__global__ __launch_bounds__(kRowThreads) void refine_rows_kernel(
    const float* __restrict__ Y, Heads h, const int* __restrict__ img_off, const float* __restrict__ final_s,
    const float* __restrict__ colstat, const float* __restrict__ lab, const long long* __restrict__ pseudo,
    const float* __restrict__ wts, const float* __restrict__ tgt, int sum_p, int n_img, float eps,
    float* __restrict__ part, float* __restrict__ dY) {
    __shared__ float red[kGroups][8 + 3 * kMaxC];
    const int img = blockIdx.y, base = img_off[img], P = img_off[img + 1] - base, C = h.C;
    const int grp = threadIdx.x >> 5, lane = threadIdx.x & 31;
    const float* cst = colstat + (size_t)img * 3 * kMaxC;
    const float* lv = lab + (size_t)img * C;
    const float inv_img = 1.0f / (float)n_img;
    float gcol[kVPL], cmaxv[kVPL], csumv[kVPL], fsumv[kVPL];
#pragma unroll
    for (int v = 0; v < kVPL; ++v) {
        const int c = lane + 32 * v;
        const bool ok = c < C;
        cmaxv[v] = ok ? cst[c] : 0.0f;
        csumv[v] = ok ? cst[kMaxC + c] : 1.0f;
        fsumv[v] = ok ? cst[2 * kMaxC + c] : 0.0f;
        gcol[v] = ok ? phi_grad(fsumv[v], lv[c], eps, C, inv_img) : 0.0f;
    }
    float lsum[6] = {0, 0, 0, 0, 0, 0};
    float rcol[3][kVPL];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int v = 0; v < kVPL; ++v) rcol[i][v] = 0.0f;

    for (int r = blockIdx.x * kGroups + grp; r < P; r += gridDim.x * kGroups) {
        const float* row = Y + (size_t)(base + r) * h.ldy;
        float* drow = dY + (size_t)(base + r) * h.ldy;
        float x[kVPL], p[kVPL], m, lse;
        // image loss -> cls / det
#pragma unroll
        for (int v = 0; v < kVPL; ++v) x[v] = (lane + 32 * v < C) ? row[h.cls + lane + 32 * v] : 0.0f;
        group_softmax(x, C, lane, p, m, lse);
        float inner = 0.0f;
#pragma unroll
        for (int v = 0; v < kVPL; ++v)
            if (lane + 32 * v < C) inner += gcol[v] * final_s[(size_t)(base + r) * C + lane + 32 * v];
        inner = group_sum(inner);
#pragma unroll
        for (int v = 0; v < kVPL; ++v) {
            const int c = lane + 32 * v;
            if (c < C) {
                const float ds = expf(row[h.det + c] - cmaxv[v]) / csumv[v];
                drow[h.cls + c] = p[v] * (gcol[v] * ds - inner);
                drow[h.det + c] = ds * gcol[v] * (p[v] - fsumv[v]);
            }
        }
        // refinement branches
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float lam = i == 0 ? 3.0f : 1.0f;
            const size_t pr = (size_t)i * sum_p + base + r;
            const int yl = (int)pseudo[pr];
            const float w = wts[pr];
            const float gscale = lam * inv_img / (float)P;
#pragma unroll
            for (int v = 0; v < kVPL; ++v) {
                x[v] = (lane + 32 * v < C) ? row[h.ref[i] + lane + 32 * v] : 0.0f;
                rcol[i][v] += (lane + 32 * v < C) ? x[v] : 0.0f;
            }
            group_softmax(x, C, lane, p, m, lse);
#pragma unroll
            for (int v = 0; v < kVPL; ++v) {
                const int c = lane + 32 * v;
                if (c < C) {
                    drow[h.ref[i] + c] = gscale * w * (p[v] - (c == yl ? 1.0f : 0.0f));
                    if (c == yl) lsum[2 * i] += (-(x[v] - m - lse)) * w;
                }
            }
            for (int c = lane; c < 4 * C; c += 32) drow[h.box[i] + c] = 0.0f;
            if (yl > 0 && lane < 4) {        // loss term of target coordinate k = lane
                const float d = row[h.box[i] + 4 * yl + lane] - tgt[pr * 4 + lane];
                const float n = fabsf(d);
                lsum[2 * i + 1] += (n < 1.0f ? 0.5f * n * n : n - 0.5f) * w;      // smooth_l1_loss.py:4-16, beta = 1
            }
        }
        // box gradient: written after the zero fill, by the SAME lane that zeroed the column (program order)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const size_t pr = (size_t)i * sum_p + base + r;
            const int yl = (int)pseudo[pr];
            if (yl > 0) {
                const float lam = i == 0 ? 3.0f : 1.0f;
                const float gscale = lam * inv_img / (float)P * wts[pr];
                for (int c = lane; c < 4 * C; c += 32) {
                    const int k = c - 4 * yl;
                    if (k >= 0 && k < 4) {
                        const float d = row[h.box[i] + c] - tgt[pr * 4 + k];
                        drow[h.box[i] + c] = gscale * (fabsf(d) < 1.0f ? d : (d > 0.0f ? 1.0f : -1.0f));
                    }
                }
            }
        }
    }
    // block partials
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const float s = group_sum(lsum[q]);
        if (lane == 0) red[grp][q] = s;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int v = 0; v < kVPL; ++v) red[grp][8 + i * kMaxC + lane + 32 * v] = rcol[i][v];
    __syncthreads();
    float* out = part + ((size_t)img * gridDim.x + blockIdx.x) * (8 + 3 * kMaxC);
    for (int q = threadIdx.x; q < 8 + 3 * kMaxC; q += kRowThreads) {
        if (q == 6 || q == 7) continue;
        float a = 0.0f;
        for (int g = 0; g < kGroups; ++g) a += red[g][q];
        out[q] = a;
    }
}

__global__ __launch_bounds__(512) void refine_finish_kernel(const float* __restrict__ part, int nblocks, Heads h,
                                                            const int* __restrict__ img_off,
                                                            const float* __restrict__ colstat,
                                                            const float* __restrict__ lab, const int* __restrict__ n_pos,
                                                            int n_img, float eps, float* __restrict__ out) {
    __shared__ float phi[kMaxC], bce[kMaxC], rsum[3][kMaxC], ls[6];
    const int img = blockIdx.x, C = h.C, P = img_off[img + 1] - img_off[img];
    const float* cst = colstat + (size_t)img * 3 * kMaxC;
    const float* lv = lab + (size_t)img * C;
    const float inv_img = 1.0f / (float)n_img;
    // 4 x 128 threads: group i < 3 sums column c of refinement branch i over the row blocks, group 3 the six loss
    // partials -- every sum keeps its block order (the same bits as one thread doing all four in sequence, a quarter
    // of the dependent-load chain)
    const int c = threadIdx.x & 127, grp = threadIdx.x >> 7;
    const size_t stride = 8 + 3 * kMaxC;
    if (grp == 0 && c < C) {
        const float s = cst[2 * kMaxC + c];
        const float ph = fminf(fmaxf(s, eps), 1.0f - eps);
        const float lp = fmaxf(logf(ph), -100.0f), lq = fmaxf(logf(1.0f - ph), -100.0f);   // torch BCE clamps logs
        phi[c] = ph;
        bce[c] = -(lv[c] * lp + (1.0f - lv[c]) * lq) / (float)C;
    }
    if (grp < 3 && c < C) {
        rsum[grp][c] = ordered_sum(part + (size_t)img * nblocks * stride + 8 + grp * kMaxC + c, nblocks, stride);
    }
    if (grp == 3 && c < 6) {
        ls[c] = ordered_sum(part + (size_t)img * nblocks * stride + c, nblocks, stride);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = out + (size_t)img * 16;
        float b = 0.0f;
        for (int k = 0; k < C; ++k) b += bce[k];
        o[0] = b * inv_img;
        for (int i = 0; i < 3; ++i) {
            const float lam = i == 0 ? 3.0f : 1.0f;
            o[1 + 2 * i] = lam * ls[2 * i] / (float)P * inv_img;
            o[2 + 2 * i] = lam * ls[2 * i + 1] / (float)P * inv_img;
        }
        const int k = max(n_pos[img], 1);
        o[7] = topk_acc(phi, lv, C, k) * inv_img;
        for (int i = 0; i < 3; ++i) o[8 + i] = topk_acc(rsum[i] + 1, lv + 1, C - 1, k) * inv_img;   // loss.py:398-400
    }
}

}  // namespace

static Heads make_heads(const int* offs, int C, int ldy) {
    Heads h;
    h.cls = offs[0]; h.det = offs[1];
    h.ref[0] = offs[2]; h.box[0] = offs[3]; h.ref[1] = offs[4]; h.box[1] = offs[5]; h.ref[2] = offs[6]; h.box[2] = offs[7];
    h.C = C; h.ldy = ldy;
    return h;
}

static int row_blocks(int max_p) {
    int b = (max_p + kGroups - 1) / kGroups;
    return b < 1 ? 1 : (b > kMaxRowBlocks ? kMaxRowBlocks : b);
}

ODW_EXPORT int64_t odw_refine_workspace(int n_img) {
    return (int64_t)(n_img > 0 ? n_img : 1) * kMaxRowBlocks * (8 + 3 * kMaxC) * sizeof(float);
}

ODW_EXPORT int odw_wsddn_scores(const float* Y, int ldy, const int* head_offsets, int C, const int* img_off, int n_img,
                                int max_p, float* final_s, float* src1, float* src2, float* colstat, void* workspace,
                                int64_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(n_img >= 0 && C >= 2 && C <= kMaxC && ldy > 0 && max_p >= 1, "wsddn_scores: bad dims (C <= %d)", kMaxC);
    if (n_img == 0) return ODW_OK;
    ODW_REQUIRE(Y && head_offsets && img_off && final_s && src1 && src2 && colstat && workspace,
                "wsddn_scores: null pointer");
    ODW_REQUIRE(workspace_bytes >= odw_refine_workspace(n_img), "wsddn_scores: workspace too small");
    Heads h = make_heads(head_offsets, C, ldy);
    const int nb = row_blocks(max_p);
    det_colstats_kernel<<<n_img, 1024, 0, stream>>>(Y, h, img_off, colstat);
    ODW_CHECK_LAUNCH("det_colstats_kernel");
    scores_rows_kernel<<<dim3(nb, n_img), kRowThreads, 0, stream>>>(Y, h, img_off, colstat, final_s, src1, src2,
                                                                   (float*)workspace);
    ODW_CHECK_LAUNCH("scores_rows_kernel");
    scores_finish_kernel<<<n_img, kMaxC, 0, stream>>>((const float*)workspace, nb, C, colstat);
    ODW_CHECK_LAUNCH("scores_finish_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_refine_losses(const float* Y, int ldy, const int* head_offsets, int C, const int* img_off, int n_img,
                                 int sum_p, int max_p, const float* final_s, const float* colstat, const float* lab,
                                 const int64_t* pseudo, const float* weights, const float* targets, const int* n_pos,
                                 float eps, float* out, float* dY, void* workspace, int64_t workspace_bytes,
                                 void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(n_img >= 0 && C >= 2 && C <= kMaxC && ldy > 0 && sum_p >= 0 && max_p >= 1, "refine_losses: bad dims");
    if (n_img == 0) return ODW_OK;
    ODW_REQUIRE(Y && head_offsets && img_off && final_s && colstat && lab && pseudo && weights && targets && n_pos &&
                    out && dY && workspace, "refine_losses: null pointer");
    ODW_REQUIRE(workspace_bytes >= odw_refine_workspace(n_img), "refine_losses: workspace too small");
    Heads h = make_heads(head_offsets, C, ldy);
    const int nb = row_blocks(max_p);
    refine_rows_kernel<<<dim3(nb, n_img), kRowThreads, 0, stream>>>(Y, h, img_off, final_s, colstat, lab,
                                                                   (const long long*)pseudo, weights, targets, sum_p,
                                                                   n_img, eps, (float*)workspace, dY);
    ODW_CHECK_LAUNCH("refine_rows_kernel");
    refine_finish_kernel<<<n_img, 512, 0, stream>>>((const float*)workspace, nb, h, img_off, colstat, lab, n_pos, n_img,
                                                    eps, out);
    ODW_CHECK_LAUNCH("refine_finish_kernel");
    return ODW_OK;
}
