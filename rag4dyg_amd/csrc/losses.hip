// The retriever's two contrastive losses and their gradients on the [5, B, d] mean-pooled embeddings of a training step
// (anchor, positive, hard negative, two augmented views) -- forward AND backward in three small launches, so that the
// step has no framework math between the encoder's forward and backward passes:
//
//   CLtime_loss   train/train_retriever.py:40-72   cross entropy over [ cos(a_i, p_j) e^{-l|ta_i - tp_j|} | cos(a_i, a_j) e^{-l|ta_i - ta_j|}
//                                                  (0 on the diagonal) | cos(a_i, n_j) e^{-l|ta_i - tn_j|} ] / tau with label i
//   info_nce      train/train_retriever.py:84-98   z = [s1; s2], sim = z z^T / tau; row r: cross entropy over
//                                                  [ sim[r, partner(r)] | sim[r, c] for c not in {r, partner(r)} ] with label 0,
//                                                  partner(r) = r +- B  (== logsumexp over c != r minus sim[r, partner])
//   loss = CLtime + alpha * info_nce              :196
//
// All sums run in a fixed order (wave butterflies, then sequential over waves / rows): the same bits on every launch and rank.
// The tables are tiny (B = 64: 64 x 192 and 128 x 128), the launches are latency-bound by design.
#include <math.h>
#include "common.h"

namespace r4d {

__device__ __forceinline__ float wave_sum_l(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max_l(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

struct LossWs {                      // float offsets into the workspace
    float *T1, *cosv, *Gc, *T2, *D, *norms, *terms;
};
__host__ __device__ static inline LossWs loss_ws(float* ws, int B) {
    LossWs w;
    const size_t b2 = (size_t)B * B;
    w.T1 = ws; w.cosv = w.T1 + 3 * b2; w.Gc = w.cosv + 3 * b2; w.T2 = w.Gc + 3 * b2; w.D = w.T2 + 4 * b2;
    w.norms = w.D + 4 * b2; w.terms = w.norms + 3 * B;
    return w;
}
static size_t loss_ws_floats(int B) { return 17 * (size_t)B * B + 6 * (size_t)B + 8; }

// emb [5][B][d]: 0 anchors, 1 positives, 2 hard negatives, 3 / 4 the augmented views.
// grid 5B: row < B: anchor i -> T1[i][0..3B) (dots with p_j | a_j | n_j) and norm; B <= row < 3B: z_r -> T2[r-B][0..2B);
// 3B <= row < 5B: positive / negative vector -> its norm.
__global__ __launch_bounds__(256) void loss_dots_kernel(const float* __restrict__ emb, int B, int d, float* __restrict__ ws) {
    extern __shared__ float own[];                                    // the workgroup's own vector
    const LossWs w = loss_ws(ws, B);
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* src;
    if (row < B) src = emb + (size_t)row * d;                          // anchor
    else if (row < 3 * B) src = emb + (size_t)(3 * B + (row - B)) * d; // z = [s1; s2] = emb[3], emb[4]
    else src = emb + (size_t)(B + (row - 3 * B)) * d;                  // positives then negatives
    for (int c = tid; c < d; c += 256) own[c] = src[c];
    __syncthreads();
    if (row >= 3 * B || row < B) {                                     // norm of an anchor / positive / negative (wave 0)
        if (wv == 0) {
            float s = 0.f;
            for (int c = lane; c < d; c += 64) s += own[c] * own[c];
            s = wave_sum_l(s);
            const int vi = row < B ? row : B + (row - 3 * B);          // norms: [a | p | n]
            if (lane == 0) w.norms[vi] = sqrtf(s);
        }
        if (row >= 3 * B) return;
    }
    const int ncol = row < B ? 3 * B : 2 * B;
    for (int c = wv; c < ncol; c += 4) {                               // a wavefront per column: fixed summation order
        const float* other;
        if (row < B) other = c < B ? emb + (size_t)(B + c) * d : (c < 2 * B ? emb + (size_t)(c - B) * d : emb + (size_t)(2 * B + (c - 2 * B)) * d);
        else other = emb + (size_t)(3 * B + c) * d;
        float s = 0.f;
        for (int k = lane; k < d; k += 64) s += own[k] * other[k];
        s = wave_sum_l(s);
        if (lane == 0) {
            if (row < B) w.T1[(size_t)row * 3 * B + c] = s; else w.T2[(size_t)(row - B) * 2 * B + c] = s;
        }
    }
}

// one workgroup; wavefront per row.  Writes cos / Gc (CLtime) and D (info_nce), the per-row loss terms, then the three losses.
__global__ __launch_bounds__(1024) void loss_softmax_kernel(const float* __restrict__ ta, const float* __restrict__ tp,
                                                            const float* __restrict__ tn, int B, float tau, float lam, float alpha,
                                                            float gscale, float* __restrict__ ws, float* __restrict__ losses) {
    const LossWs w = loss_ws(ws, B);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
    const float eps = 1e-8f;                                           // F.cosine_similarity's clamp on each norm
    for (int i = wv; i < B; i += nw) {                                 // ---- CLtime rows
        const float na = fmaxf(w.norms[i], eps), tai = ta[i];
        float mx = -INFINITY;
        for (int c = lane; c < 3 * B; c += 64) {
            const int j = c % B, blk = c / B;
            const float nv = fmaxf(w.norms[blk == 0 ? B + j : (blk == 1 ? j : 2 * B + j)], eps);
            const float tj = blk == 0 ? tp[j] : (blk == 1 ? ta[j] : tn[j]);
            float dec = expf(-lam * fabsf(tai - tj));
            if (blk == 1 && j == i) dec = 0.f;                         // decay_factor_neg.fill_diagonal_(0)
            const float cs = w.T1[(size_t)i * 3 * B + c] / (na * nv);
            w.cosv[(size_t)i * 3 * B + c] = cs;
            w.Gc[(size_t)i * 3 * B + c] = dec;                         // decay for now, the coefficient below
            mx = fmaxf(mx, cs * dec / tau);
        }
        mx = wave_max_l(mx);
        float se = 0.f;
        for (int c = lane; c < 3 * B; c += 64) se += expf(w.cosv[(size_t)i * 3 * B + c] * w.Gc[(size_t)i * 3 * B + c] / tau - mx);
        se = wave_sum_l(se);
        // softmax as exp(logit - max) / sum, the loss term as log(sum) - (logit - max): every quantity stays O(1).  (Forming
        // lse = max + log(sum) first rounds at the magnitude of the logits -- ~100 for info_nce at tau = 0.07 -- and that ONE
        // error scales all probabilities of the row alike: their sum misses 1 by ~4e-6 and sums of gradients that cancel
        // exactly in theory (ln_f.bias) came out 40x less accurate than the reference's fp32 autograd.)
        const float lg_se = logf(se);
        for (int c = lane; c < 3 * B; c += 64) {
            const size_t o = (size_t)i * 3 * B + c;
            const float dec = w.Gc[o], sh = w.cosv[o] * dec / tau - mx;
            const float p = expf(sh) / se;
            w.Gc[o] = (p - (c == i ? 1.f : 0.f)) / (float)B * dec / tau * gscale;       // dLoss / dcos[i, c]
            if (c == i) w.terms[i] = lg_se - sh;
        }
    }
    for (int r = wv; r < 2 * B; r += nw) {                             // ---- info_nce rows
        const int partner = r < B ? r + B : r - B;
        float mx = -INFINITY;
        for (int c = lane; c < 2 * B; c += 64) if (c != r) mx = fmaxf(mx, w.T2[(size_t)r * 2 * B + c] / tau);
        mx = wave_max_l(mx);
        float se = 0.f;
        for (int c = lane; c < 2 * B; c += 64) if (c != r) se += expf(w.T2[(size_t)r * 2 * B + c] / tau - mx);
        se = wave_sum_l(se);
        const float lg_se = logf(se);
        for (int c = lane; c < 2 * B; c += 64) {
            const size_t o = (size_t)r * 2 * B + c;
            const float sh = w.T2[o] / tau - mx;
            w.D[o] = c == r ? 0.f : (expf(sh) / se - (c == partner ? 1.f : 0.f)) / (float)(2 * B) / tau * alpha * gscale;   // dLoss / d(z_r . z_c)
            if (c == partner) w.terms[B + r] = lg_se - sh;
        }
    }
    __syncthreads();
    if (tid == 0) {                                                    // sequential, fixed order
        float cl = 0.f, nce = 0.f;
        for (int i = 0; i < B; ++i) cl += w.terms[i];
        for (int r = 0; r < 2 * B; ++r) nce += w.terms[B + r];
        cl /= (float)B;
        nce = alpha * nce / (float)(2 * B);
        losses[0] = cl; losses[1] = nce; losses[2] = cl + nce;
    }
}

// grid 5B: the gradient of one embedding row; thread t owns columns t, t + 256, ...
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ emb, int B, int d, const float* __restrict__ ws_c,
                                                        float* __restrict__ demb) {
    const LossWs w = loss_ws(const_cast<float*>(ws_c), B);
    const int row = blockIdx.x, tid = threadIdx.x;
    const float eps = 1e-8f;
    const int grp = row / B, i = row % B;
    const float* A = emb;                    // anchors
    const float* P = emb + (size_t)B * d;
    const float* Nn = emb + (size_t)2 * B * d;
    const float* Z = emb + (size_t)3 * B * d;
    for (int c0 = tid; c0 < d; c0 += 256) {
        float g = 0.f;
        if (grp == 0) {                                                // anchor i: as the row of all three blocks + as column j = i of the a-a block
            float acc = 0.f, self = 0.f;
            for (int c = 0; c < 3 * B; ++c) {
                const int j = c % B, blk = c / B;
                const float* v = blk == 0 ? P : (blk == 1 ? A : Nn);
                const float nv = fmaxf(w.norms[blk == 0 ? B + j : (blk == 1 ? j : 2 * B + j)], eps);
                const float gc = w.Gc[(size_t)i * 3 * B + c];
                acc += gc * v[(size_t)j * d + c0] / nv;
                self += gc * w.cosv[(size_t)i * 3 * B + c];
            }
            for (int j = 0; j < B; ++j) {
                const float gc = w.Gc[(size_t)j * 3 * B + B + i];
                acc += gc * A[(size_t)j * d + c0] / fmaxf(w.norms[j], eps);
                self += gc * w.cosv[(size_t)j * 3 * B + B + i];
            }
            const float na = fmaxf(w.norms[i], eps);
            g = acc / na - A[(size_t)i * d + c0] * self / (na * na);
        } else if (grp == 1 || grp == 2) {                             // positive / hard negative j = i: a column of its block
            const int blk = grp == 1 ? 0 : 2;
            const float* v = grp == 1 ? P : Nn;
            const float nv = fmaxf(w.norms[(grp == 1 ? B : 2 * B) + i], eps);
            float acc = 0.f, self = 0.f;
            for (int a = 0; a < B; ++a) {
                const float gc = w.Gc[(size_t)a * 3 * B + blk * B + i];
                acc += gc * A[(size_t)a * d + c0] / fmaxf(w.norms[a], eps);
                self += gc * w.cosv[(size_t)a * 3 * B + blk * B + i];
            }
            g = acc / nv - v[(size_t)i * d + c0] * self / (nv * nv);
        } else {                                                       // z_r, r = (grp - 3) * B + i
            const int r = (grp - 3) * B + i;
            float acc = 0.f;
            for (int c = 0; c < 2 * B; ++c) acc += (w.D[(size_t)r * 2 * B + c] + w.D[(size_t)c * 2 * B + r]) * Z[(size_t)c * d + c0];
            g = acc;
        }
        demb[(size_t)row * d + c0] = g;
    }
}

}  // namespace r4d

using namespace r4d;

extern "C" {

size_t r4d_retriever_losses_workspace_bytes(int32_t B) { return B > 0 ? loss_ws_floats(B) * sizeof(float) : 0; }

int r4d_retriever_losses_f32(const float* emb_d, const float* t_anchor_d, const float* t_pos_d, const float* t_neg_d, int32_t B, int32_t d,
                             float temperature, float lambda_decay, float alpha, float grad_scale, float* losses_d, float* d_emb_d,
                             void* workspace_d, size_t workspace_bytes, void* stream) {
    R4D_REQUIRE(emb_d && t_anchor_d && t_pos_d && t_neg_d && losses_d, "retriever_losses: null pointer");
    R4D_REQUIRE(B >= 1 && B <= 4096 && d >= 1 && d <= 8192, "retriever_losses: B=%d d=%d out of range", B, d);
    R4D_REQUIRE(temperature > 0.f, "retriever_losses: temperature must be positive");
    if (!workspace_d || workspace_bytes < r4d_retriever_losses_workspace_bytes(B)) {
        set_error("retriever_losses: workspace too small");
        return R4D_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace_d;
    hipLaunchKernelGGL(loss_dots_kernel, dim3(5 * B), dim3(256), (size_t)d * sizeof(float), s, emb_d, B, d, ws);
    R4D_CHECK_LAUNCH("loss_dots");
    hipLaunchKernelGGL(loss_softmax_kernel, dim3(1), dim3(1024), 0, s, t_anchor_d, t_pos_d, t_neg_d, B, temperature, lambda_decay, alpha,
                       grad_scale, ws, losses_d);
    R4D_CHECK_LAUNCH("loss_softmax");
    if (d_emb_d) {
        hipLaunchKernelGGL(loss_grad_kernel, dim3(5 * B), dim3(256), 0, s, emb_d, B, d, ws, d_emb_d);
        R4D_CHECK_LAUNCH("loss_grad");
    }
    return R4D_OK;
}

}  // extern "C"
