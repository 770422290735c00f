/*
 * r4d.h -- C ABI of librag4dyg_hip.so: the MI355X (gfx950) encode-and-retrieve hot path of RAG4DyG.
 *
 * The reference (YuxiaWu/RAG4DyG) is 100 % Python and has no FFI/plugin layer;
 * each entry point below cites the reference Python call surface it replaces.  Conventions:
 *   - extern "C", plain pointers and sizes, no torch / C++ types;
 *   - every pointer named *_d / documented "device" is a caller-owned HBM pointer (e.g. torch
 *     tensor.data_ptr()); the library never allocates device memory: scratch comes from an explicit
 *     caller-provided workspace whose size is returned by the matching *_workspace_bytes query;
 *   - kernels are enqueued on the caller's stream (a hipStream_t passed as void*; NULL = default
 *     stream) and the call returns without synchronising (graph-capturable);
 *   - return value: 0 = R4D_OK, negative = error; the message is available from r4d_last_error()
 *     (thread-local).  No exceptions cross the ABI;
 *   - all floating-point data is fp32 (the reference computes in fp32), Jaccard ratios are f64
 *     (python floats), indices are int32 / int64 as stated;
 *   - PROCESS MODEL: one process drives ONE device (one process per GPU, torch.distributed over RCCL between them).  A few
 *     launch parameters are latched per process the first time a kernel family runs (occupancy of the scan kernels, the
 *     > 64 KB dynamic-LDS attribute of the LDS-DMA scan forms, the persistent GEMM's ticket slots), i.e. for the device that was
 *     current then; a process that switched devices afterwards would get R4D_ERR_HIP from the first such launch on the second
 *     device (loud, never a wrong result).  The arithmetic switches (r4d_set_gemm_split3, r4d_set_attention_fused) are
 *     process-wide too: the ranks of a sharded job must select the same Conv1D arithmetic, or the merged per-shard top-k differs
 *     from the one-GPU result in the last bits of the scores.
 */
#ifndef R4D_H
#define R4D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define R4D_OK 0
#define R4D_ERR_INVALID (-1)      /* bad argument / unsupported shape */
#define R4D_ERR_HIP (-2)          /* a HIP runtime call or kernel launch failed */
#define R4D_ERR_WORKSPACE (-3)    /* workspace too small */

#define R4D_ABI_VERSION 6

/* ABI version of the loaded library. */
int r4d_abi_version(void);
/* 0 for a product build.  Non-zero when the library was compiled with one of the kernel-ablation macros of
 * tools/kc_ablate.sh (bit 0 KC_DBG, bit 1 ATT_DBG, bit 2 SK_DBG, bit 3 JAC_DBG, bit 4 SCAN_DBG, bit 5 S3_DBG, bit 6 H2_DBG): such a build
 * computes WRONG results by construction and the Python binding refuses to load it outside tools/. */
int r4d_build_flags(void);
/* Message of the last failing call on this thread ("" if none). */
const char* r4d_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * GPT-2 encoder (SimpleDyG backbone).
 * Replaces GPT2Model.forward  models/modeling_gpt2.py:357-509 (== models/modeling_rag.py:412-564)
 * and its callees Block :224-235, Attention :140-197, MLP :209-212, Conv1D modeling_utils.py:1267-1271,
 * plus the retriever's mean-pool  train/train_retriever.py:419-420,430-432.
 * Weight pointers use the reference checkpoint layout (Conv1D weights are [in,out]).
 * ---------------------------------------------------------------------------------------------- */
typedef struct r4d_gpt2_config {
    int32_t n_layer;
    int32_t n_head;
    int32_t n_embd;        /* d; d % 64 == 0, (d / n_head) % 16 == 0 */
    int32_t vocab;         /* rows of wte (len(tokenizer) after resize_token_embeddings) */
    int32_t n_positions;   /* rows of wpe (1024) */
    float   ln_eps;        /* layer_norm_epsilon (1e-5) */
} r4d_gpt2_config;

typedef struct r4d_gpt2_layer {   /* device pointers, one struct per transformer.h.<i> */
    const float* ln_1_w;      const float* ln_1_b;        /* [d] */
    const float* c_attn_w;    const float* c_attn_b;      /* [d,3d], [3d] */
    const float* attn_proj_w; const float* attn_proj_b;   /* [d,d],  [d]  */
    const float* ln_2_w;      const float* ln_2_b;        /* [d] */
    const float* c_fc_w;      const float* c_fc_b;        /* [d,4d], [4d] */
    const float* mlp_proj_w;  const float* mlp_proj_b;    /* [4d,d], [d]  */
    /* OPTIONAL transposed copies of the four static Conv1D weights ([out,in] row-major, i.e. K contiguous), made
     * once by the host when the checkpoint is loaded; NULL = not provided.  With them every GEMM operand is
     * k-contiguous and the faster b128-LDS kernel runs (same values up to fp32 summation order). */
    const float* c_attn_wT;   const float* attn_proj_wT;  /* [3d,d], [d,d]  */
    const float* c_fc_wT;     const float* mlp_proj_wT;   /* [4d,d], [d,4d] */
    /* OPTIONAL bf16x3 planes of the same four weights (r4d_split3_planes_bf16: uint16 [3][out][in], hi / mid / lo), made
     * once per checkpoint; NULL = not provided.  With them (and r4d_set_gemm_split3 != 0, the default) the four Conv1D
     * GEMMs of a block run on the bf16 matrix cores at fp32 accuracy (r4d_conv1d_s3_f32) in the encoder calls (the cached
     * decode step keeps its weight-stream kernels); everything else about the call is unchanged. */
    const uint16_t* c_attn_w3;  const uint16_t* attn_proj_w3;
    const uint16_t* c_fc_w3;    const uint16_t* mlp_proj_w3;
    /* TRAINING only, optional: the planes of the same weights as the operand of the data-gradient GEMMs dx = dy . W^T
     * (r4d_split3_planes_bf16(w, K = out, N = in, transposed = 1): uint16 [3][in][out]); NULL = exact-f32 kernel.  Like the
     * wT copies, every plane set must be refreshed by the caller after each optimizer step. */
    const uint16_t* c_attn_w3t; const uint16_t* attn_proj_w3t;
    const uint16_t* c_fc_w3t;   const uint16_t* mlp_proj_w3t;
    /* DECODE only, optional (ABI v4): LayerNorm pre-folded into the two projections that read it (r4d_fold_layernorm_f32 on
     * c_attn_wT with ln_1, on c_fc_wT with ln_2): *_wTg [out,in] = gain[in] * W^T, *_lnc [2][out] = (sum_in gain W, sum_in shift W).
     * With them the cached decode step (d in {512, 768}, batch <= 32) evaluates LN(x) W as rstd (x W'^T - mean c1) + c2 without
     * loading the gain / shift or forming c1 / c2 per launch; NULL = it forms them from ln_1 / ln_2 and *_wT as before. */
    const float* c_attn_wTg;    const float* c_attn_lnc;
    const float* c_fc_wTg;      const float* c_fc_lnc;
    /* OPTIONAL f16x2 planes of the four weights (ABI v5; r4d_split2_planes_f16: uint16 [out][in/32][2][32], hi and 2^11-scaled lo
     * as fp16), made once per checkpoint; NULL = not provided.  With them and r4d_set_gemm_split3(2) the four Conv1D GEMMs of a
     * block run on the fp16 matrix cores with THREE products per fp32 product at fp32 accuracy (r4d_conv1d_h2_f32; csrc/gemm_h2.hip
     * for the error analysis and the operand range: |activation| < 2^18, |weight| < 6e4). */
    const uint16_t* c_attn_h2;  const uint16_t* attn_proj_h2;
    const uint16_t* c_fc_h2;    const uint16_t* mlp_proj_h2;
    /* (ABI v6: r4d_gpt2_train_forward_f32 uses the *_h2 planes too in mode 2 -- the FORWARD GEMMs of a training step on the fp16
     * matrix cores; every gradient GEMM keeps bf16x3: a gradient operand needs fp32's exponent range.  Refreshed by the caller
     * after each optimizer step, like every plane set.) */
} r4d_gpt2_layer;

typedef struct r4d_gpt2_weights {
    const float* wte;                /* device [vocab,d]  transformer.wte.weight */
    const float* wpe;                /* device [n_positions,d] */
    const float* ln_f_w;             /* device [d] */
    const float* ln_f_b;
    const r4d_gpt2_layer* layers;    /* HOST array [n_layer] of device-pointer structs */
    const float* lm_head;            /* device [vocab,d] lm_head.weight, or NULL when it is tied to wte
                                      * (modeling_utils.py:155-181).  The reference UNTIES the two whenever it replaces
                                      * transformer.wte or model.transformer without re-tying (hepth node-feature injection,
                                      * utils/tokenizer.py:56-66; load_and_freeze_params, utils/model.py:71-78), so trained
                                      * generator checkpoints carry two different tensors.  Read by the greedy step only. */
} r4d_gpt2_weights;

/* Scratch bytes r4d_gpt2_encode_f32 needs for a [B,T] batch. */
size_t r4d_gpt2_workspace_bytes(const r4d_gpt2_config* cfg, int32_t B, int32_t T);

/*
 * One encoder forward over a right-padded batch.
 *   ids_d           device int64 [B,T] token ids, or NULL when inputs_embeds_d is given
 *   inputs_embeds_d device f32 [B,T,d] (generator-style call, modeling_rag.py:460-461,517-524) or NULL
 *   out_hidden_d    device f32 [B,T,d]  ln_f output ("hidden_states"), or NULL
 *   out_meanpool_d  device f32 [B,d]    mean over ALL T padded positions (train_retriever.py:420), or NULL
 *   out_layers_d    device f32 [n_layer,B,T,d] residual stream ENTERING each block (debug/parity), or NULL
 *   out_qkv_d       device f32 [n_layer,B,T,3d] c_attn outputs ("presents" source, modeling_gpt2.py:187), or NULL
 * Position ids are 0..T-1 (modeling_gpt2.py:420-423); dropout is the identity (eval).
 */
int r4d_gpt2_encode_f32(const r4d_gpt2_config* cfg, const r4d_gpt2_weights* w,
                        const int64_t* ids_d, const float* inputs_embeds_d, int32_t B, int32_t T,
                        float* out_hidden_d, float* out_meanpool_d, float* out_layers_d, float* out_qkv_d,
                        void* workspace_d, size_t workspace_bytes, void* stream);

/*
 * Several right-padded batches in ONE launch sequence (throughput form of the call above; same values).
 * The reference encodes one 32-sequence batch per model call (train_retriever.py:414-432); a batch's
 * embeddings depend on its own padding (mean over padded positions, :420), so batches are NOT merged -- but
 * the row-wise work (LayerNorm, Conv1D GEMMs) of all batches runs over their concatenated rows, while the
 * embedding/positions, attention and mean-pool stay per batch.
 *   ids_d  HOST array [n_groups] of device int64 [Bs[g], Ts[g]] pointers;  Bs, Ts HOST arrays
 *   out_meanpool_d device f32 [sum(Bs), d], batches in order
 */
size_t r4d_gpt2_groups_workspace_bytes(const r4d_gpt2_config* cfg, int32_t n_groups, const int32_t* Bs,
                                       const int32_t* Ts);
int r4d_gpt2_encode_groups_f32(const r4d_gpt2_config* cfg, const r4d_gpt2_weights* w, int32_t n_groups,
                               const int64_t* const* ids_d, const int32_t* Bs, const int32_t* Ts,
                               float* out_meanpool_d, void* workspace_d, size_t workspace_bytes, void* stream);

/* The same launch sequence with every output of r4d_gpt2_encode_f32: batches given as ids (ids_d) OR as embeddings
 * (embeds_d: HOST array of device f32 [Bs[g], Ts[g], d] pointers), exactly one of the two non-NULL.  Rows of all
 * batches are concatenated in order (batch g starts at row sum_{j<g} Bs[j]*Ts[j]):
 *   out_hidden_d   f32 [rows, d] or NULL     out_meanpool_d f32 [sum(Bs), d] or NULL (at least one of the two)
 *   out_qkv_d      f32 [n_layer, rows, 3d] or NULL (c_attn output per layer: the K / V rows a decode cache starts from)
 * Used to prefill a RAGGED set of prompts grouped by length: a causal model never lets right-padding reach a real
 * position, so each group only needs padding to its own longest member. */
int r4d_gpt2_encode_groups_ex_f32(const r4d_gpt2_config* cfg, const r4d_gpt2_weights* w, int32_t n_groups,
                                  const int64_t* const* ids_d, const float* const* embeds_d, const int32_t* Bs,
                                  const int32_t* Ts, float* out_hidden_d, float* out_meanpool_d, float* out_qkv_d,
                                  void* workspace_d, size_t workspace_bytes, void* stream);

/*
 * Incremental decode with a key/value cache -- the model's `past` mechanism (`layer_past` concatenation,
 * modeling_gpt2.py:177-197; `past` / `presents`, :400-509), which the reference's greedy loops never use
 * (utils/Evaluation_generator.py:153-167 and Evaluation_SimpleDyG.py:126-134 re-run the FULL forward per token).
 * One call = one new position for each of B independent sequences:
 *   ids_d            device int64 [B] new token ids, or NULL when inputs_embeds_d is given
 *   inputs_embeds_d  device f32 [B,d] or NULL
 *   pos_d            device int32 [B]: position of the new token = number of positions already cached for that
 *                    sequence (position embedding wpe[pos]; must be < t_cap and < n_positions, else the row is NaN)
 *   kv_cache_d       device f32 [n_layer, B, t_cap, 2*d]: per position the K row then the V row of every head.
 *                    Rows [0, pos) of each sequence must hold its history (fill them from r4d_gpt2_encode_f32's
 *                    out_qkv_d: columns d..3d of c_attn); row pos is WRITTEN by this call.
 *   out_hidden_d     device f32 [B,d]: ln_f output of the new position (feed r4d_lm_logits_f32)
 * Same values as the last row of a full forward over the extended sequence, up to fp32 summation order.
 */
size_t r4d_gpt2_decode_workspace_bytes(const r4d_gpt2_config* cfg, int32_t B);
int r4d_gpt2_decode_step_f32(const r4d_gpt2_config* cfg, const r4d_gpt2_weights* w, const int64_t* ids_d,
                             const float* inputs_embeds_d, const int32_t* pos_d, float* kv_cache_d, int32_t B,
                             int32_t t_cap, float* out_hidden_d, void* workspace_d, size_t workspace_bytes,
                             void* stream);

/*
 * Greedy decoding with the loop state ON THE DEVICE -- the reference's greedy loops (utils/Evaluation_SimpleDyG.py:126-145,
 * utils/Evaluation_generator.py:153-175: argmax of the last position's logits, append, stop rules) as one launch
 * sequence per token with no host round trip, so that a step can be captured once and replayed as a HIP graph
 * (a cached step is launch-latency-bound: ~70 small kernels).  One step =
 *     logits = last . wte^T  ->  per sequence: v = argmax (lowest index among equal maxima);
 *     if active: out_tokens[b][gen_len++] = v;  active &= !(gen_len >= max_gen || v in eos || gen_len >= out_cap
 *                                                          || lens + 1 >= min(len_limit, t_cap))
 *     next = v;  pos = active ? lens : 0;  lens += active        (finished sequences keep their slot and are ignored)
 *     last = r4d_gpt2_decode_step_f32(ids = next, pos)
 * All arrays are device memory owned by the caller and must stay valid (and in place) for the life of a graph:
 *   last_d [B,d] f32 in/out (ln_f row of the newest position)   logits_d [B,V] f32 scratch (the step's logits)
 *   next_d [B] i64   lens_d [B] i32 (positions cached; in/out)   pos_d [B] i32   active_d [B] i32 (in/out, 1 = running)
 *   gen_len_d [B] i32 (in/out)   out_tokens_d [B,out_cap] i32   params_d int32[8] (below)
 */
typedef struct r4d_greedy_state {
    float* last_d;
    float* logits_d;
    int64_t* next_d;
    int32_t* lens_d;
    int32_t* pos_d;
    int32_t* active_d;
    int32_t* gen_len_d;
    int32_t* out_tokens_d;
    const int32_t* params_d; /* device int32[8], read every step (so one captured graph serves every batch):
                              *   [0] max_gen    stop after this many generated tokens
                              *   [1] len_limit  stop once lens + 1 >= min(len_limit, t_cap) (the cache row written next is lens)
                              *   [2] n_eos 0..4, [3..6] eos ids: stop after generating any of them;  [7] reserved */
    int32_t out_cap;         /* row length of out_tokens_d */
} r4d_greedy_state;
typedef struct r4d_decode_graph r4d_decode_graph;      /* opaque: a captured, instantiated greedy step */

size_t r4d_gpt2_greedy_workspace_bytes(const r4d_gpt2_config* cfg, int32_t B);
/* one step, launched kernel by kernel on `stream` */
int r4d_gpt2_greedy_step_f32(const r4d_gpt2_config* cfg, const r4d_gpt2_weights* w, const r4d_greedy_state* st,
                             float* kv_cache_d, int32_t B, int32_t t_cap, void* workspace_d, size_t workspace_bytes,
                             void* stream);
/* capture the same step into a HIP graph (nothing runs); every pointer reachable from the arguments is baked in */
int r4d_gpt2_greedy_graph_create(const r4d_gpt2_config* cfg, const r4d_gpt2_weights* w, const r4d_greedy_state* st,
                                 float* kv_cache_d, int32_t B, int32_t t_cap, void* workspace_d, size_t workspace_bytes,
                                 r4d_decode_graph** out_graph);
/* replay it n_steps times on `stream` (asynchronous) */
int r4d_decode_graph_launch(r4d_decode_graph* graph, int32_t n_steps, void* stream);
void r4d_decode_graph_destroy(r4d_decode_graph* graph);

/* lm_logits = hidden @ wte^T  (tied lm_head, modeling_gpt2.py:585; modeling_rag.py:675).
 * hidden_d [M,d], wte_d [V,d] -> logits_d [M,V]. */
int r4d_lm_logits_f32(const float* hidden_d, const float* wte_d, int32_t M, int32_t V, int32_t d,
                      float* logits_d, void* stream);

/* Attention implementation switch: -1 = auto (default) / 1 = fused flash-style kernels wherever instantiated
 * (head_dim in {32,64}: key-split kernel; {96,128,256}: column-split kernel; scores never leave the CU),
 * 0 = three launches (batched Q.K^T GEMM, causal softmax, P.V GEMM; also the fallback for other head dims),
 * 2 = fused with the key-split kernel forced at head_dim 96/128/256 (A/B tuning).  Same results to fp32 rounding. */
int r4d_set_attention_fused(int32_t mode);

/* f16x2 attention (ABI v5, additive): in gemm mode 2 (f16x2) r4d_gpt2_encode_* runs head_dim 128 / 256 attention on the fp16
 * matrix cores (csrc/attention_h2.hip) -- the c_attn GEMM then writes its result as "h2 words" (uint32 per element:
 * fp16 hi | fp16 lo' << 16 of value / 4, value = hi + 2^-11 lo' to 2^-22 relative; csrc/h2.h) instead of fp32, and the attention
 * kernel consumes them without conversion.  Not used when the caller asks for the qkv tensor (out_qkv_d) and for other head
 * dims; |q|, |k|, |v| < 2^18.  r4d_set_attention_h2(0) keeps the exact-f32 attention kernels in every mode (returns the previous
 * setting; process-wide like r4d_set_gemm_split3: ranks must agree).
 * r4d_pack_h2_words_f32: x_d fp32 [n] -> words_d uint32 [n] (what the GEMM epilogue writes; for tests and external producers).
 * r4d_attention_h2_f32: r4d_attention_f32 on words: qkv_words_d [B*T, 3d] -> a_d fp32 [B*T, d]; head_dim 128 / 256 only. */
int r4d_set_attention_h2(int32_t on);
/* KEY-BLOCKED K (round 5): row-major K costs every K load instruction of the attention 32 cache lines (each lane
 * its own key row, 32 bytes of each line used); in the key-blocked image -- [ceil(rows / 32)][n_head][head_dim / 8][2][32][4] uint32:
 * for every 32 consecutive token rows, head and 8-element step one contiguous 1 KB chunk [half][row & 31][4 words] -- the same
 * instruction reads 8-10 whole lines (a fifth off the head_dim-256 launch in the bench step; every head_dim the f16x2 attention serves).  Inside the encoder the LDS-DMA c_attn GEMM writes the
 * image directly (r4d_set_attention_kblk(0) keeps row-major K; returns the previous setting; env R4D_ATT_KBLK=0 does the same).
 * r4d_pack_kblk_words builds it from row-major qkv words (tests, external producers); r4d_attention_h2_kblk_f32 is
 * r4d_attention_h2_f32 reading K from it (the K columns of qkv_words_d are not read) -- bit-identical results. */
int r4d_set_attention_kblk(int32_t on);
int r4d_pack_kblk_words(const uint32_t* qkv_words_d, int64_t rows, int32_t n_head, int32_t d, uint32_t* kblk_d, void* stream);
int r4d_attention_h2_kblk_f32(const uint32_t* qkv_words_d, const uint32_t* kblk_d, int32_t B, int32_t T, int32_t n_head, int32_t d,
                              float* a_d, void* stream);
int r4d_pack_h2_words_f32(const float* x_d, int64_t n, uint32_t* words_d, void* stream);
int r4d_attention_h2_f32(const uint32_t* qkv_words_d, int32_t B, int32_t T, int32_t n_head, int32_t d, float* a_d, void* stream);

/* Pre-split activations (ABI v6, additive; csrc/gemm_h2p.hip).  In gemm mode 2 (f16x2) r4d_gpt2_encode_* keeps the INPUT rows of c_attn,
 * c_fc and mlp.c_proj as "f16x2 lines" -- uint16 [rows][K/32][2][32]: per row and 32 consecutive k one 128-byte line of 32 hi = RN16(x/4)
 * then 32 lo' = RN16((x/4 - hi) 2^11), the layout r4d_split2_planes_f16 gives the weights, 4 bytes per element like fp32 -- written by
 * their producers (the LayerNorm kernels, the c_fc GELU epilogue), and both GEMM operands travel global -> LDS by DMA.  The results are
 * those of the register-staged f16x2 GEMM bit for bit; r4d_set_gemm_h2p(0) selects that one again (returns the previous setting;
 * process-wide like r4d_set_gemm_split3).  n_embd % 256 == 0, else the register-staged path.
 * r4d_split2_lines_f16: x_d fp32 [rows,K] -> lines (tests, external producers).
 * r4d_conv1d_h2p_f32: r4d_conv1d_h2_f32 with x given as lines; out_lines != 0 (epilogue 1 = gelu only): y_d receives lines [M][N/32][2][32].
 * r4d_layernorm_lines_f32: r4d_layernorm_f32 whose output rows are lines (d % 256 == 0; what the encoder launches in this mode). */
int r4d_set_gemm_h2p(int32_t on);
int r4d_layernorm_lines_f32(const float* x_d, const float* w_d, const float* b_d, int32_t rows, int32_t d, float eps,
                            uint16_t* y_lines_d, void* stream);
int r4d_split2_lines_f16(const float* x_d, int64_t rows, int32_t K, uint16_t* lines_d, void* stream);
int r4d_conv1d_h2p_f32(const uint16_t* x_lines_d, const uint16_t* planes_d, const float* bias_d, const float* residual_d, int32_t M,
                       int32_t K, int32_t N, int32_t epilogue, int32_t out_lines, void* y_d, void* stream);

/* Range guard (ABI v6, additive).  The f16x2 arithmetic has fp16's exponent range: an activation, q, k or v of magnitude >= 2^18
 * becomes inf in its first fp16 term and every hidden state downstream of it NaN -- and NaN scores are ordered below everything by
 * the selection, i.e. an ordinary-looking top-k of garbage.  r4d_set_range_flag registers ONE caller-owned
 * device word for the process (one process per device; NULL unregisters; the library never clears it).  While registered:
 *   bit 0 (R4D_RANGE_NONFINITE_HIDDEN) is OR-ed in by the ln_f / mean-pool kernel of r4d_gpt2_encode_* for every row of the
 *         final residual stream that is non-finite or whose variance overflowed (any arithmetic; in f16x2 mode this is how an
 *         out-of-range activation surfaces);
 *   bit 1 (R4D_RANGE_BAD_NORM) by r4d_normalize_rows_f32 for every row whose norm is NaN, inf or zero.
 * The calls stay asynchronous: the caller reads the word at its own synchronisation points.  What the Python host mirror does
 * with it (rag4dyg_amd/ops.py, retrieval.py, gpt2.py): retrieval.encode_batches and GPT2Model.forward re-run an f16x2 call whose
 * bit 0 came up ONCE under bf16x3 (fp32's exponent range) with a warning and raise R4DError if the bit comes up again;
 * PoolIndex and search raise R4DError on either bit before any top-k result is handed out; bench.py reports the word of the
 * whole timed region.  encode_* called directly leave the check to the caller (ops.take_range_flag). */
#define R4D_RANGE_NONFINITE_HIDDEN 1u
#define R4D_RANGE_BAD_NORM 2u
int r4d_set_range_flag(uint32_t* flag_d);

/* --- single ops, exported for per-op parity tests (same kernels the encoder launches) --- */
/* y = LayerNorm(x) over the last dim.  nn.LayerNorm, modeling_gpt2.py:219,221,339. */
int r4d_layernorm_f32(const float* x_d, const float* w_d, const float* b_d, int32_t rows, int32_t d,
                      float eps, float* y_d, void* stream);
/* y = epilogue(x[M,K] @ W[K,N] + bias[N]); epilogue: 0 none, 1 gelu_new (modeling_gpt2.py:206),
 * 2 add residual_d[M,N].  Conv1D.forward, modeling_utils.py:1267-1271. */
int r4d_conv1d_f32(const float* x_d, const float* w_d, const float* w_t_d /* optional [N,K] copy or NULL */,
                   const float* bias_d, const float* residual_d,
                   int32_t M, int32_t K, int32_t N, int32_t epilogue, float* y_d, void* stream);
/*
 * The same Conv1D on the bf16 matrix cores AT FP32 ACCURACY ("bf16x3" split): every fp32 operand is the exact sum of three
 * bf16 numbers (hi + mid + lo) and a product is evaluated as its six leading partial products
 * (v_mfma_f32_32x32x16_bf16, fp32 accumulation) -- 2.67x fewer matrix-pipe cycles than the exact-f32 MFMA at an error
 * against float64 that is no larger (shorter fp32 accumulation chains; acceptance table in profiles/).
 * r4d_split3_planes_bf16: the STATIC operand, once per checkpoint: w_d fp32 [K,N] (reference Conv1D layout; transposed != 0:
 * an [N,K] copy) -> planes_d bf16 [3][N][K] (hi, mid, lo planes, K contiguous; 6 * N * K bytes).
 * r4d_conv1d_s3_f32: y = epilogue(x[M,K] @ W + bias) with W given as those planes; x stays fp32 and is split on the fly.
 * K % 32 == 0; epilogue as r4d_conv1d_f32.
 */
int r4d_split3_planes_bf16(const float* w_d, int32_t K, int32_t N, int32_t transposed, uint16_t* planes_d, void* stream);
int r4d_conv1d_s3_f32(const float* x_d, const uint16_t* planes_d, const float* bias_d, const float* residual_d,
                      int32_t M, int32_t K, int32_t N, int32_t epilogue, float* y_d, void* stream);
/*
 * The same Conv1D on the FP16 matrix cores at fp32 accuracy with THREE products per fp32 product ("f16x2", ABI v5;
 * csrc/gemm_h2.hip): x = hi + 2^-11 lo' with hi = RN16(x), lo' = RN16((x - hi) 2^11) (the second term stays a normal fp16
 * number over the whole range), a.b = hi.hi + 2^-11 (lo'.hi + hi.lo') in two fp32 accumulator sets joined once; half the
 * matrix-pipe cycles of bf16x3 at an error against float64 no larger than bf16x3's or the exact-f32 MFMA's (acceptance table
 * in profiles/).  Operand range: |x| < 2^18 for the activation (pre-scaled by 2^-2 inside the kernel; full precision for elements >= 2.4e-4,
 * an absolute floor of 6e-11 below), |w| < 6e4 for the
 * weight; beyond it the result is inf / NaN, not a silently wrong number.
 * r4d_split2_planes_f16: w_d fp32 [K,N] (transposed != 0: [N,K]) -> planes_d fp16 [N][K/32][2][32] (4 * N * K bytes: per row and 32 consecutive k one 128-byte line of 32 hi then 32 lo'
 * values; K % 32 == 0), once per checkpoint.
 * r4d_conv1d_h2_f32: y = epilogue(x[M,K] @ W + bias), K % 32 == 0; epilogue as r4d_conv1d_f32.
 */
int r4d_split2_planes_f16(const float* w_d, int32_t K, int32_t N, int32_t transposed, uint16_t* planes_d, void* stream);
int r4d_conv1d_h2_f32(const float* x_d, const uint16_t* planes_d, const float* bias_d, const float* residual_d,
                      int32_t M, int32_t K, int32_t N, int32_t epilogue, float* y_d, void* stream);
/* Conv1D arithmetic of the encoder / training-forward GEMMs.  1 (default): bf16x3 where a layer carries its planes; 2: f16x2
 * where it carries those planes (bf16x3 where it only has the bf16 ones; the training GEMMs stay bf16x3); 0: exact-f32 MFMA
 * kernels everywhere (planes ignored).  PROCESS-WIDE (one process drives one device; ranks of a sharded job must agree, or the
 * merged shards differ from the one-GPU result in the last bits), not thread-safe against concurrent calls. */
/* Decode-only fold of a LayerNorm into the projection that reads it (ABI v4): wT_d [N,K] (k-contiguous weight), ln_w_d / ln_b_d [K]
 * -> wTg_d [N,K] = ln_w[k] * wT[n][k] and lnc_d [2][N] = (sum_k ln_w[k] wT[n][k], sum_k ln_b[k] wT[n][k]); the operands of
 * r4d_gpt2_layer's c_attn_wTg / c_attn_lnc (with ln_1) and c_fc_wTg / c_fc_lnc (with ln_2).  Once per checkpoint. */
int r4d_fold_layernorm_f32(const float* wT_d, const float* ln_w_d, const float* ln_b_d, int32_t N, int32_t K, float* wTg_d,
                           float* lnc_d, void* stream);
int r4d_set_gemm_split3(int32_t mode);
int r4d_get_gemm_split3(void);
/* Causal multi-head attention on packed c_attn output qkv_d [B,T,3d] -> a_d [B,T,d] (heads merged).
 * Attention._attn + split/merge_heads, modeling_gpt2.py:140-175; scale = division by sqrt(hd) (:143).
 * scores_ws_d: device scratch of r4d_attention_workspace_bytes(B,H,T). */
size_t r4d_attention_workspace_bytes(int32_t B, int32_t n_head, int32_t T);
int r4d_attention_f32(const float* qkv_d, int32_t B, int32_t T, int32_t n_head, int32_t d,
                      float* a_d, void* scores_ws_d, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Retrieval scoring.  Replaces train/train_retriever.py:433-438 (normalise, dot, (x+1)/2) and the
 * ranking of save_index_score :357-358 / the hit loop :461-467.
 * ---------------------------------------------------------------------------------------------- */
/* out[i,:] = x[i,:] / ||x[i,:]||_2 (no eps, train_retriever.py:433,436). */
int r4d_normalize_rows_f32(const float* x_d, int32_t n, int32_t d, float* out_d, void* stream);

size_t r4d_score_topk_workspace_bytes(int32_t Q, int32_t N, int32_t k);
/*
 * S = (q_hat @ pool_hat^T + 1) / 2 over one pool shard, then per-row top-k with the canonical order
 * (score descending, pool index ascending).
 *   q_hat_d [Q,d], pool_hat_d [N,d] row-normalised;  index_offset = global index of shard row 0
 *   out_val_d f32 [Q,k], out_idx_d int64 [Q,k] (global indices)        (k <= 64, k <= N)
 *   out_scores_d f32 [Q,N] full score rows (file-compat mode) or NULL
 */
int r4d_score_topk_f32(const float* q_hat_d, const float* pool_hat_d, int32_t Q, int32_t N, int32_t d,
                       int32_t k, int64_t index_offset, float* out_val_d, int64_t* out_idx_d,
                       float* out_scores_d, void* workspace_d, size_t workspace_bytes, void* stream);
/* Per-row top-k of an f32 matrix with the same canonical order (e.g. the ground-truth top-3 of the float32
 * Jaccard rows, train_retriever.py:461-462).  rows <= 65535.  One launch for n <= 16384 * 1024 / k. */
size_t r4d_topk_f32_workspace_bytes(int32_t rows, int32_t n, int32_t k);
int r4d_topk_f32(const float* m_d, int32_t rows, int32_t n, int32_t k, float* out_val_d, int64_t* out_idx_d,
                 void* workspace_d, size_t workspace_bytes, void* stream);
/* Merge G per-shard candidate lists (after the RCCL all-gather): vals_d [G,Q,k], idx_d [G,Q,k]
 * -> out [Q,k], same canonical order; result == single-GPU top-k by construction.  Shards in ascending order of
 * their index ranges (rank order), every list sorted by (value desc, index asc) as r4d_score_topk_f32 emits them. */
size_t r4d_merge_topk_workspace_bytes(int32_t G, int32_t Q, int32_t k);
int r4d_merge_topk_f32(const float* vals_d, const int64_t* idx_d, int32_t G, int32_t Q, int32_t k,
                       float* out_val_d, int64_t* out_idx_d, void* workspace_d, size_t workspace_bytes, void* stream);
/* Full-row ranking (file-compat mode: the reference writes the whole permutation of the pool per query,
 * train_retriever.py:357-362; retrieval_data_annotation.py:88-93): perm_d int32 [rows,n] = stable argsort of -scores
 * (ties by ascending index) == np.argsort(-S, axis=1, kind='stable'); NaN last, -0.0 == +0.0.  Any n < 2^31 / 16:
 * chunks of 2048 (key, index) pairs are sorted in LDS, an element's rank is the sum of its binary-search ranks in the
 * sorted chunks of its row (O(n * n/2048 * 11) instead of O(n^2) compares).  elem_bytes = 4 (f32) or 8 (f64). */
size_t r4d_argsort_workspace_bytes(int32_t rows, int32_t n, int32_t elem_bytes);
int r4d_argsort_desc_f32(const float* scores_d, int32_t rows, int32_t n, int32_t* perm_d, void* workspace_d,
                         size_t workspace_bytes, void* stream);
int r4d_argsort_desc_f64(const double* scores_d, int32_t rows, int32_t n, int32_t* perm_d, void* workspace_d,
                         size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Jaccard pool annotation.  Replaces occurrence_matrix / co_occurrence_ratio,
 * retrieval_data_annotation.py:36-41 / :5-15, and np.fill_diagonal :172-173.
 * Sets are CSR: ptr int32 [n+1], idx int32 sorted-unique token ids in [0,vocab); *_nnz = ptr[n] = number of
 * idx elements (the idx buffers must hold at least one element even when every set is empty).
 * ---------------------------------------------------------------------------------------------- */
/* out_d f64 [na,nb] row-major: |A_i & B_j| / |A_i | B_j|, 0.0 when either set is empty;
 * zero_diag != 0 additionally writes 0.0 at i == j. */
int r4d_jaccard_f64(const int32_t* a_ptr_d, const int32_t* a_idx_d, int32_t na, int32_t a_nnz,
                    const int32_t* b_ptr_d, const int32_t* b_idx_d, int32_t nb, int32_t b_nnz,
                    int32_t vocab, int32_t zero_diag, double* out_d, void* stream);
/* The same matrix with the A rows VISITED in the order a_order_d (int32 [na], a permutation of 0..na-1; NULL = file
 * order).  Results land in their own rows; only the schedule changes: a wavefront walks the tokens of four A rows
 * jointly, to the longest of the four, so visiting rows of similar length together (longest first) saves the padding
 * steps -- 1.7x fewer token steps on log-normal input-set lengths.  Values are identical for every order.
 * DENSE TOKENS (a_dense_d / b_dense_d, both or neither; NULL = none): uint32 [na] / [nb], bit t of a set's word = "the
 * set contains dense token t".  The caller picks up to 32 tokens (the frequent ones: the <|timeK|> tokens that
 * get_input_seq keeps, retrieval_data_annotation.py:17-20) and REMOVES them from the CSR lists; the kernel adds
 * popcount(a_dense & b_dense) to the intersection and the popcounts to the set sizes: the same integers as the plain
 * CSR form, for two vector instructions per row and 64 pairs instead of 4.3 per token. */
int r4d_jaccard_ordered_f64(const int32_t* a_ptr_d, const int32_t* a_idx_d, int32_t na, int32_t a_nnz,
                            const int32_t* b_ptr_d, const int32_t* b_idx_d, int32_t nb, int32_t b_nnz,
                            int32_t vocab, int32_t zero_diag, const int32_t* a_order_d,
                            const uint32_t* a_dense_d, const uint32_t* b_dense_d, double* out_d, void* stream);
/* The same matrix with the schedule PREPARED ON THE DEVICE (round 5): the caller hands over the plain CSR sets and says
 * which of the two schedule aids it wants; nothing is computed on the host and no torch op runs.
 *   dense_split != 0: the 32 most frequent tokens -- counted over a strided sample of at most 2 x 512 sets, ties to the
 *     smaller token id -- leave the lists for one membership word per set (the a_dense_d / b_dense_d of the entry above);
 *     the remaining tokens are squeezed to the front of each set's own segment in a workspace copy of idx.
 *   sort_rows != 0: the A rows are visited longest (remaining) list first (a_order_d above; ties in no particular order).
 * Neither changes a value: out_d holds the bits r4d_jaccard_f64 writes.  The sets hold DISTINCT tokens (as
 * get_input_seq / get_target_seq build them, retrieval_data_annotation.py:17-34).  workspace_d from
 * r4d_jaccard_prepared_workspace_bytes (may be NULL when both flags are 0). */
size_t r4d_jaccard_prepared_workspace_bytes(int32_t na, int32_t a_nnz, int32_t nb, int32_t b_nnz, int32_t vocab);
int r4d_jaccard_prepared_f64(const int32_t* a_ptr_d, const int32_t* a_idx_d, int32_t na, int32_t a_nnz,
                             const int32_t* b_ptr_d, const int32_t* b_idx_d, int32_t nb, int32_t b_nnz,
                             int32_t vocab, int32_t zero_diag, int32_t dense_split, int32_t sort_rows,
                             double* out_d, void* workspace_d, size_t workspace_bytes, void* stream);
/* Per-row top-k (value descending, index ascending) of an f64 matrix: save_score_file_train,
 * retrieval_data_annotation.py:97-103 (topk=10).  ws from r4d_topk_f64_workspace_bytes. */
size_t r4d_topk_f64_workspace_bytes(int32_t rows, int32_t n, int32_t k);
int r4d_topk_f64(const double* m_d, int32_t rows, int32_t n, int32_t k, double* out_val_d,
                 int32_t* out_idx_d, void* workspace_d, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Retriever TRAINING step.  Replaces, for the encoder, what torch autograd does for the reference in
 * train/train_retriever.py:177-214: five forwards (anchor, positive, hard negative, two augmented views), loss.backward(),
 * clip_grad_norm_, AdamW.  The contrastive losses ([B,3B] / [2B,2B] similarity tables, :40-98) and their gradient on the
 * mean-pooled embeddings are r4d_retriever_losses_f32 below (round 3; a caller may still substitute its own).
 * ---------------------------------------------------------------------------------------------- */
typedef struct r4d_gpt2_layer_grads {   /* device pointers, same shapes as r4d_gpt2_layer's reference-layout tensors */
    float* ln_1_w;      float* ln_1_b;
    float* c_attn_w;    float* c_attn_b;      /* [d,3d], [3d] */
    float* attn_proj_w; float* attn_proj_b;   /* [d,d],  [d]  */
    float* ln_2_w;      float* ln_2_b;
    float* c_fc_w;      float* c_fc_b;        /* [d,4d], [4d] */
    float* mlp_proj_w;  float* mlp_proj_b;    /* [4d,d], [d]  */
} r4d_gpt2_layer_grads;
typedef struct r4d_gpt2_grads {
    float* wte;                           /* [vocab,d]        (zeroed, then scatter-added) */
    float* wpe;                           /* [n_positions,d]  (zeroed, then scatter-added) */
    float* ln_f_w;
    float* ln_f_b;
    const r4d_gpt2_layer_grads* layers;   /* HOST array [n_layer] */
} r4d_gpt2_grads;

/* nn.Dropout of the training forward (model.train(), train_retriever.py:161): embd_p after wte + wpe (modeling_gpt2.py:337,427),
 * attn_p on the attention probabilities (:153), resid_p on the two residual branches (:194, :212).  Masks come from a
 * counter-based generator (Philox-4x32-10 keyed by `seed`, counter = (element index / 4, site, step)): the SAME struct must be
 * passed to the forward and the backward call of a step; change `step` every step.  Sites: 4 * layer + {0: attention
 * probabilities, 1: attention branch, 2: MLP branch}, R4D_DROPOUT_SITE_EMBD for the embeddings; element index = offset in the
 * concatenated [rows, d] activations (rows of all batches in call order), for the probabilities the offset in the step's
 * [B*H, T, ceil128(T)] blocks laid end to end in batch order.  A null pointer (or all p == 0) is the identity (model.eval()). */
typedef struct {
    float embd_p, attn_p, resid_p;
    uint64_t seed, step;
} r4d_train_dropout;
#define R4D_DROPOUT_SITE_EMBD 65535u
/* out = (resid_d ? resid_d : 0) + dropout_p(x_d) for one site (per-op tests; the kernel the step launches).  n and
 * index_base multiples of 4; out_d may be x_d or resid_d. */
int r4d_dropout_f32(const float* x_d, const float* resid_d, int64_t n, float* out_d, float p, uint64_t seed, uint64_t step,
                    uint32_t site, uint64_t index_base, void* stream);

/* Scratch of one step: the activations the backward pass needs (16 * rows * d floats per layer + the attention
 * probabilities) and the backward temporaries.  The SAME buffer goes to the forward and to the backward call. */
size_t r4d_gpt2_train_workspace_bytes(const r4d_gpt2_config* cfg, int32_t n_groups, const int32_t* Bs, const int32_t* Ts);
/* Forward over up to 16 right-padded id batches (one launch sequence over their concatenated rows) that keeps every
 * activation the backward pass reads.  The optional wT copies of the layers are USED when present (faster forward GEMMs): the
 * caller must refresh them after every optimizer step; the backward pass reads the reference layout only.  out_meanpool_d f32 [sum(Bs), d] = torch.mean(h, dim=1) per sequence. */
int r4d_gpt2_train_forward_f32(const r4d_gpt2_config* cfg, const r4d_gpt2_weights* w, int32_t n_groups,
                               const int64_t* const* ids_d, const int32_t* Bs, const int32_t* Ts, float* out_meanpool_d,
                               const r4d_train_dropout* dropout, void* workspace_d, size_t workspace_bytes, void* stream);
/* Backward of the call above (same arguments, same workspace contents): d_meanpool_d f32 [sum(Bs), d] = dLoss / d(embeddings)
 * -> every gradient of `grads` is OVERWRITTEN with dLoss / d(parameter) (lm_head is not on this path: the retriever discards
 * the logits, train_retriever.py:177-179). */
int r4d_gpt2_train_backward_f32(const r4d_gpt2_config* cfg, const r4d_gpt2_weights* w, const r4d_gpt2_grads* grads,
                                int32_t n_groups, const int64_t* const* ids_d, const int32_t* Bs, const int32_t* Ts,
                                const float* d_meanpool_d, const r4d_train_dropout* dropout,
                                void* workspace_d, size_t workspace_bytes, void* stream);
/* The step's loss head on the device (train/train_retriever.py:40-98,196): emb_d f32 [5,B,d] = the mean-pooled embeddings of
 * anchors, positives, hard negatives and the two augmented views (r4d_gpt2_train_forward_f32's output); t_*_d f32 [B] the query
 * times of the three sequence families.  losses_d f32[3] <- { CLtime_loss, alpha * info_nce, their sum };  d_emb_d (nullable)
 * f32 [5,B,d] <- grad_scale * d(sum)/d(emb): what r4d_gpt2_train_backward_f32 takes as d_meanpool_d.  Fixed summation order
 * (the same bits on every launch and rank).  cosine_similarity clamps each norm at 1e-8 like torch. */
size_t r4d_retriever_losses_workspace_bytes(int32_t B);
int r4d_retriever_losses_f32(const float* emb_d, const float* t_anchor_d, const float* t_pos_d, const float* t_neg_d, int32_t B,
                             int32_t d, float temperature, float lambda_decay, float alpha, float grad_scale, float* losses_d,
                             float* d_emb_d, void* workspace_d, size_t workspace_bytes, void* stream);
/* Single backward ops, exported for per-op parity tests (the kernels the step launches). */
/* Conv1D parameter gradients (modeling_utils.py:1267-1271 under autograd): dw_d [in,out] = x^T . dy, db_d [out] (nullable) =
 * column sums of dy, for x_d [rows,in] and dy_d [rows,out]; in / out multiples of 4. */
size_t r4d_weight_grad_workspace_bytes(int32_t rows, int32_t in_features, int32_t out_features);
int r4d_weight_grad_f32(const float* x_d, const float* dy_d, int32_t rows, int32_t in_features, int32_t out_features, float* dw_d,
                        float* db_d, void* workspace_d, size_t workspace_bytes, void* stream);
size_t r4d_layernorm_bwd_workspace_bytes(int32_t rows, int32_t d);
/* dx = dLayerNorm/dx (+ add_d when given; add_d may be dx_d), dw / db = gains' and shifts' gradients (overwritten). */
int r4d_layernorm_bwd_f32(const float* x_d, const float* w_d, const float* dy_d, const float* add_d, int32_t rows, int32_t d,
                          float eps, float* dx_d, float* dw_d, float* db_d, void* workspace_d, size_t workspace_bytes,
                          void* stream);
int r4d_gelu_new_f32(const float* pre_d, int64_t n, float* y_d, void* stream);                     /* modeling_gpt2.py:25 */
int r4d_gelu_new_bwd_f32(const float* pre_d, const float* dy_d, int64_t n, float* dx_d, void* stream);   /* dx may be dy */
/* p_d [nbh,T,ld] causal probabilities (zero right of the diagonal), dp_d [nbh,T,ld] = dLoss/dP on entry and
 * dLoss/d(raw Q.K^T logits) on return (the logits were divided by scale_div before the softmax, modeling_gpt2.py:143);
 * columns right of the diagonal are written as zero. */
int r4d_causal_softmax_bwd_f32(const float* p_d, float* dp_d, int32_t nbh, int32_t T, int32_t ld, float scale_div, void* stream);
/* accum_d[0] += sum x^2 (total gradient norm of clip_grad_norm_, train_retriever.py:210; zero accum_d[0] first).  accum_d is
 * f32[R4D_SUMSQ_FLOATS]: element 0 the running total, the rest scratch of the two-stage sum (no atomics: the same bits on
 * every data-parallel rank). */
#define R4D_SUMSQ_FLOATS 1025
int r4d_sumsq_accumulate_f32(const float* x_d, int64_t n, float* accum_d, void* stream);
/* One transformers.AdamW update of a flat tensor (utils/model.py:80-93; decoupled weight decay applied after the step,
 * bias correction on): step >= 1 is the update count; grad_sumsq_d (nullable) with max_grad_norm > 0 clips the gradient
 * by min(1, max_grad_norm / (sqrt(sum) + 1e-6)) on the fly (torch.nn.utils.clip_grad_norm_). */
int r4d_adamw_step_f32(float* p_d, const float* g_d, float* m_d, float* v_d, int64_t n, double lr, double beta1, double beta2,
                       double eps, double weight_decay, int32_t step, const float* grad_sumsq_d, float max_grad_norm,
                       void* stream);

/* ------------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py): when enabled, every kernel launch is bracketed by HIP events on the
 * launch stream and accumulated per kernel class together with its ALGORITHMIC work (flop for the
 * MFMA-bound classes, bytes for the HBM-bound ones; definitions in DESIGN.md).  Off by default;
 * not graph-capturable while on.
 * ---------------------------------------------------------------------------------------------- */
int r4d_profile_enable(int32_t on);          /* also clears the accumulated records */
int r4d_profile_num_classes(void);
const char* r4d_profile_class_name(int32_t cls);
/* Synchronises the recorded events; total_ms / launches / work summed over the launches of `cls`. */
int r4d_profile_read(int32_t cls, double* total_ms, int64_t* launches, double* work);

/* Dispatcher-branch coverage: every host-side decision that selects a kernel variant (tile shape, template instantiation,
 * split-K form, fallback) counts its launches under a name.  Tests enumerate the table and assert that each branch ran
 * (names starting with "tuning:" are reachable through environment switches only). */
int r4d_dispatch_num_branches(void);
const char* r4d_dispatch_branch_name(int32_t i);
int64_t r4d_dispatch_branch_hits(int32_t i);      /* launches through branch i since load / the last reset */
int r4d_dispatch_reset(void);

#ifdef __cplusplus
}
#endif
#endif /* R4D_H */
