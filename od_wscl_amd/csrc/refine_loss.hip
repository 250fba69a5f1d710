// refine_loss.hip -- the dense part of the OD-WSCL loss as two launches per step, forward AND backward:
//   wsddn_scores : final = softmax_row(cls) * softmax_col-per-image(det), softmax(ref1), softmax(ref2)
//                  (roi_heads/weak_head/loss.py:234-247,357) -- the score matrices the selection kernels read
//   refine_losses: MIL image loss (BCE of the clamped column sums, loss.py:353-354), 3 x weighted
//                  cross-entropy (loss.py:375-377), 3 x smooth-L1 box regression on the pseudo-positive
//                  rows (loss.py:380-394), the 4 top-k accuracies (loss.py:25-33,396-400), and the
//                  gradient of the sum of the 7 losses w.r.t. the fused predictor output Y (P x 5C+12C).
// The reference runs ~80 tiny kernels forward and ~80 backward through autograd for this; here one
// workgroup per image does it with deterministic tree reductions.  Closed-form gradients:
//   g_c = dL/dphi_c (0 where phi was clamped);  ddet_rc = ds_rc g_c (cs_rc - colsum_c);
//   dcls_rc = cs_rc (g_c ds_rc - sum_c' g_c' final_rc');  dref = lam w (softmax - onehot)/P;
//   dbbox = lam w clamp(diff,-1,1)/P on the 4 columns of the pseudo label.
#include "odw_common.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kMaxC = 128;

struct Heads {        // column offsets inside one row of Y (predictor order: roi_weak_predictors.py:158-165)
    int cls, det, ref[3], box[3];
    int C, ldy;
};

// sum over the block, result broadcast; red = LDS float[kThreads]
__device__ float block_sum(float v, float* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int off = kThreads / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    float out = red[0];
    __syncthreads();
    return out;
}

// per-column reduction over rows: thread = (column c, slice s); partial[s][c] in LDS -> result[c]
template <typename F, typename R>
__device__ void column_reduce(int P, int C, F value, R combine, float init, float* part, float* result) {
    const int S = kThreads / C;
    const int c = threadIdx.x % C, s = threadIdx.x / C;
    float acc = init;
    if (s < S)
        for (int r = s; r < P; r += S) acc = combine(acc, value(r, c));
    if (s < S) part[s * C + c] = acc;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float a = init;
        for (int k = 0; k < S; ++k) a = combine(a, part[k * C + threadIdx.x]);
        result[threadIdx.x] = a;
    }
    __syncthreads();
}

__global__ __launch_bounds__(kThreads) void wsddn_scores_kernel(const float* __restrict__ Y, Heads h,
                                                                const int* __restrict__ img_off,
                                                                float* __restrict__ final_s, float* __restrict__ src1,
                                                                float* __restrict__ src2, float* __restrict__ colstat) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* part = sm;                       // kThreads
    float* cmax = sm + kThreads;            // C
    float* csum = cmax + kMaxC;             // C
    float* fsum = csum + kMaxC;             // C
    const int img = blockIdx.x, base = img_off[img], P = img_off[img + 1] - base, C = h.C;
    const float* y = Y + (size_t)base * h.ldy;
    column_reduce(P, C, [&](int r, int c) { return y[(size_t)r * h.ldy + h.det + c]; },
                  [](float a, float b) { return fmaxf(a, b); }, -__builtin_inff(), part, cmax);
    column_reduce(P, C, [&](int r, int c) { return expf(y[(size_t)r * h.ldy + h.det + c] - cmax[c]); },
                  [](float a, float b) { return a + b; }, 0.0f, part, csum);
    // rows
    for (int r = threadIdx.x; r < P; r += kThreads) {
        const float* row = y + (size_t)r * h.ldy;
        float m = -__builtin_inff();
        for (int c = 0; c < C; ++c) m = fmaxf(m, row[h.cls + c]);
        float s = 0.0f;
        for (int c = 0; c < C; ++c) s += expf(row[h.cls + c] - m);
        for (int c = 0; c < C; ++c) {
            const float cs = expf(row[h.cls + c] - m) / s;
            const float ds = expf(row[h.det + c] - cmax[c]) / csum[c];
            final_s[(size_t)(base + r) * C + c] = cs * ds;
        }
        for (int k = 0; k < 2; ++k) {
            float* dst = (k == 0 ? src1 : src2) + (size_t)(base + r) * C;
            const float* x = row + h.ref[k];
            float mm = -__builtin_inff();
            for (int c = 0; c < C; ++c) mm = fmaxf(mm, x[c]);
            float ss = 0.0f;
            for (int c = 0; c < C; ++c) ss += expf(x[c] - mm);
            for (int c = 0; c < C; ++c) dst[c] = expf(x[c] - mm) / ss;
        }
    }
    __syncthreads();
    column_reduce(P, C, [&](int r, int c) { return final_s[(size_t)(base + r) * C + c]; },
                  [](float a, float b) { return a + b; }, 0.0f, part, fsum);
    if ((int)threadIdx.x < C) {
        float* cs = colstat + (size_t)img * 3 * kMaxC;
        cs[threadIdx.x] = cmax[threadIdx.x];
        cs[kMaxC + threadIdx.x] = csum[threadIdx.x];
        cs[2 * kMaxC + threadIdx.x] = fsum[threadIdx.x];           // column sums of final_score (unclamped)
    }
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

__global__ __launch_bounds__(kThreads) void refine_losses_kernel(
    const float* __restrict__ Y, Heads h, const int* __restrict__ img_off, const float* __restrict__ final_s,
    const float* __restrict__ colstat, const float* __restrict__ lab, const long long* __restrict__ pseudo,
    const float* __restrict__ wts, const float* __restrict__ tgt, int sum_p, int n_img, const int* __restrict__ n_pos,
    float eps, float* __restrict__ out, float* __restrict__ dY) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* red = sm;                        // kThreads
    float* part = red + kThreads;           // kThreads
    float* gcol = part + kThreads;          // C : dL_img/dphi_c
    float* colv = gcol + kMaxC;             // C scratch (ref column sums / phi)
    const int img = blockIdx.x, base = img_off[img], P = img_off[img + 1] - base, C = h.C;
    const float* y = Y + (size_t)base * h.ldy;
    float* dy = dY + (size_t)base * h.ldy;
    const float* cst = colstat + (size_t)img * 3 * kMaxC;
    const float* lv = lab + (size_t)img * C;
    const float inv_img = 1.0f / (float)n_img;
    float* o = out + (size_t)img * 16;      // per-image partial outputs: 7 losses + 4 accuracies

    // ---- MIL image loss
    float bce = 0.0f;
    if ((int)threadIdx.x < C) {
        const int c = threadIdx.x;
        const float s = cst[2 * kMaxC + c];
        const float phi = fminf(fmaxf(s, eps), 1.0f - eps);
        // torch BCE clamps the logs at -100
        const float lp = fmaxf(logf(phi), -100.0f), lq = fmaxf(logf(1.0f - phi), -100.0f);
        bce = -(lv[c] * lp + (1.0f - lv[c]) * lq) / (float)C;
        const bool inside = s >= eps && s <= 1.0f - eps;             // clamp passes the gradient inside [min,max]
        gcol[c] = inside ? (-(lv[c] / phi) + (1.0f - lv[c]) / (1.0f - phi)) / (float)C * inv_img : 0.0f;
        colv[c] = phi;
    }
    const float loss_img = block_sum(bce, red) * inv_img;
    if (threadIdx.x == 0) {
        o[0] = loss_img;
        o[7] = topk_acc(colv, lv, C, max(n_pos[img], 1)) * inv_img;
    }
    __syncthreads();
    // ---- gradient of the image loss w.r.t. cls / det logits
    for (int r = threadIdx.x; r < P; r += kThreads) {
        const float* row = y + (size_t)r * h.ldy;
        const float* fr = final_s + (size_t)(base + r) * C;
        float inner = 0.0f;
        for (int c = 0; c < C; ++c) inner += gcol[c] * fr[c];
        float m = -__builtin_inff();
        for (int c = 0; c < C; ++c) m = fmaxf(m, row[h.cls + c]);
        float s = 0.0f;
        for (int c = 0; c < C; ++c) s += expf(row[h.cls + c] - m);
        for (int c = 0; c < C; ++c) {
            const float cs = expf(row[h.cls + c] - m) / s;
            const float ds = expf(row[h.det + c] - cst[c]) / cst[kMaxC + c];
            dy[(size_t)r * h.ldy + h.cls + c] = cs * (gcol[c] * ds - inner);
            dy[(size_t)r * h.ldy + h.det + c] = ds * gcol[c] * (cs - cst[2 * kMaxC + c]);
        }
    }
    // ---- refinement branches
    for (int i = 0; i < 3; ++i) {
        const float lam = i == 0 ? 3.0f : 1.0f;
        const long long* ps = pseudo + (size_t)i * sum_p + base;
        const float* w = wts + (size_t)i * sum_p + base;
        const float* t = tgt + ((size_t)i * sum_p + base) * 4;
        const float gscale = lam * inv_img / (float)P;
        float ce = 0.0f, reg = 0.0f;
        for (int r = threadIdx.x; r < P; r += kThreads) {
            const float* x = y + (size_t)r * h.ldy + h.ref[i];
            float* dx = dy + (size_t)r * h.ldy + h.ref[i];
            const int yl = (int)ps[r];
            float m = -__builtin_inff();
            for (int c = 0; c < C; ++c) m = fmaxf(m, x[c]);
            float s = 0.0f;
            for (int c = 0; c < C; ++c) s += expf(x[c] - m);
            const float lse = logf(s);
            ce += (-(x[yl] - m - lse)) * w[r];
            for (int c = 0; c < C; ++c) dx[c] = gscale * w[r] * (expf(x[c] - m) / s - (c == yl ? 1.0f : 0.0f));
            const float* b = y + (size_t)r * h.ldy + h.box[i];
            float* db = dy + (size_t)r * h.ldy + h.box[i];
            for (int c = 0; c < 4 * C; ++c) db[c] = 0.0f;
            if (yl > 0) {
                for (int k = 0; k < 4; ++k) {
                    const float d = b[4 * yl + k] - t[(size_t)r * 4 + k];
                    const float n = fabsf(d);
                    reg += (n < 1.0f ? 0.5f * n * n : n - 0.5f) * w[r];     // smooth_l1_loss.py:4-16, beta = 1
                    db[4 * yl + k] = gscale * w[r] * (n < 1.0f ? d : (d > 0.0f ? 1.0f : -1.0f));
                }
            }
        }
        const float ce_sum = block_sum(ce, red);
        const float reg_sum = block_sum(reg, red);
        // acc_ref: column sums of the raw branch logits, classes 1..C-1 (loss.py:398-400)
        column_reduce(P, C, [&](int r, int c) { return y[(size_t)r * h.ldy + h.ref[i] + c]; },
                      [](float a, float b) { return a + b; }, 0.0f, part, colv);
        if (threadIdx.x == 0) {
            o[1 + 2 * i] = lam * ce_sum / (float)P * inv_img;
            o[2 + 2 * i] = lam * reg_sum / (float)P * inv_img;
            o[8 + i] = topk_acc(colv + 1, lv + 1, C - 1, max(n_pos[img], 1)) * inv_img;
        }
        __syncthreads();
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

ODW_EXPORT int odw_wsddn_scores(const float* Y, int ldy, const int* head_offsets, int C, const int* img_off, int n_img,
                                float* final_s, float* src1, float* src2, float* colstat, void* stream_) {
    ODW_REQUIRE(n_img >= 0 && C >= 2 && C <= kMaxC && ldy > 0, "wsddn_scores: bad dims (C <= %d)", kMaxC);
    if (n_img == 0) return ODW_OK;
    ODW_REQUIRE(Y && head_offsets && img_off && final_s && src1 && src2 && colstat, "wsddn_scores: null pointer");
    Heads h = make_heads(head_offsets, C, ldy);
    size_t lds = (kThreads + 3 * kMaxC) * sizeof(float);
    wsddn_scores_kernel<<<n_img, kThreads, lds, (hipStream_t)stream_>>>(Y, h, img_off, final_s, src1, src2, colstat);
    ODW_CHECK_LAUNCH("wsddn_scores_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_refine_losses(const float* Y, int ldy, const int* head_offsets, int C, const int* img_off, int n_img,
                                 int sum_p, const float* final_s, const float* colstat, const float* lab,
                                 const int64_t* pseudo, const float* weights, const float* targets, const int* n_pos,
                                 float eps, float* out, float* dY, void* stream_) {
    ODW_REQUIRE(n_img >= 0 && C >= 2 && C <= kMaxC && ldy > 0 && sum_p >= 0, "refine_losses: bad dims");
    if (n_img == 0) return ODW_OK;
    ODW_REQUIRE(Y && head_offsets && img_off && final_s && colstat && lab && pseudo && weights && targets && n_pos &&
                    out && dY, "refine_losses: null pointer");
    Heads h = make_heads(head_offsets, C, ldy);
    size_t lds = (2 * kThreads + 2 * kMaxC) * sizeof(float);
    refine_losses_kernel<<<n_img, kThreads, lds, (hipStream_t)stream_>>>(Y, h, img_off, final_s, colstat, lab,
                                                                         (const long long*)pseudo, weights, targets,
                                                                         sum_p, n_img, n_pos, eps, out, dY);
    ODW_CHECK_LAUNCH("refine_losses_kernel");
    return ODW_OK;
}
