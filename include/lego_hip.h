/* lego_hip.h -- C ABI of liblego_hip.so: the MI355X (gfx950) kernels for the Legommenders
 * two-tower training hot path (NAML / NRMS `Legommender.forward` + backward + Adam).
 *
 * The reference (Jyonn/Legommenders) is pure Python/PyTorch and has no native layer; each
 * entry point below names the reference call site (file:line, relative to the reference root)
 * whose arithmetic it replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; tensors are dense,
 *     row-major, fp32 / int32; `ld*` are row strides in elements (multiples of 4, 16-B aligned rows)
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue
 *     work (no allocation, no synchronisation) so a caller may capture them into a hipGraph
 *   - "dyn" arguments are device int32 scalars read by the kernels (ragged row counts produced
 *     by lego_plan_batch), so the host never synchronises on a batch's raggedness
 *   - return value 0 = ok; non-zero = error, message via lego_last_error() (thread-local)
 *   - thread-safety: no global mutable state besides the thread-local error string and the process-wide
 *     product mode (lego_set_product_mode; set it before launching, not concurrently with launches)
 *
 * Row spaces of one batch (built by lego_plan_batch / lego_plan_dense)
 *   token rows  r in [0,R)      one per live title token; rows of one item instance are contiguous
 *   Y rows      [0,R) token rows followed by [R,R+NI) one category row per item instance
 *   rowinfo[r]  bit0 has-left-neighbour, bit1 has-right-neighbour, bit2 live, bits 8.. instance
 *   counters[]  [0]=R  [1]=NI  [2]=R+NI  [3]=number of history instances (NI - B*C)
 */
#ifndef LEGO_HIP_H
#define LEGO_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LEGO_ABI_VERSION 8
#define LEGO_COUNTERS 8

const char* lego_last_error(void);
int lego_abi_version(void);

/* Product mode of the LARGE dense products (lego_linear_*, lego_conv3_fwd / _bwd_data / _bwd_weight with >= 2048 rows).
 *   0 (default)  exact f32 on the fp32 matrix instructions: the mode every parity statement of this library is made in
 *                (the reference computes in torch fp32: model/operators/, model/common/attention.py)
 *   1            split-bf16: every operand element x = hi + lo (two bf16), a.b = lo_a hi_b + hi_a lo_b + hi_a hi_b on the bf16
 *                matrix instructions with fp32 accumulation.  OPT-IN throughput mode (about 2x the f32 rate), relative error
 *                ~4e-6 of the largest output per product: NOT bit-compatible with the exact mode; the Winograd conv entry points
 *                do not take part (use the direct lego_conv3_* ones).  LEGO_SPLIT_BF16=1 in the environment selects it at load.
 * lego_set_product_mode returns 0 / an error for an unknown mode; lego_get_product_mode returns the mode. */
int lego_set_product_mode(int mode);
int lego_get_product_mode(void);

typedef struct {
    float p;             /* drop probability, 0 = off (eval) */
    uint64_t seed;       /* per-run seed */
    uint32_t site;       /* stream id: site index + 16 * step, so every step draws fresh masks */
    const uint8_t* mask; /* optional (NULL = draw in the kernel): the keep bits of this (seed, site) made ahead of time by
                          * lego_dropout_mask over the SAME column count: byte [(row / 4) * cols + col], bit i = row % 4 */
} lego_dropout;

/* The reference's nn.Dropout sites on the path -- Transformation.dropout (loader/embedding_hub.py:73-96), CNNOperator.dropout
 * (model/operators/cnn_operator.py:44,57), MultiheadAttention's attention dropout (model/operators/attention_operator.py:36-41)
 * -- are epilogues of the producing kernels here (argument `drop`), regenerated from (seed, site) in backward instead of
 * storing a mask.  lego_dropout_mask: keep bits of one site for rows [0, rows) x cols columns made ahead of time (see
 * lego_dropout.mask); drop->mask is ignored. */
int lego_dropout_mask(const lego_dropout* drop, int rows_cap, const int32_t* rows_dyn, int cols, uint8_t* mask, void* stream);

/* ---- a11 / a2: ragged batch plan.  Replaces Resampler.rebuild_clicks padding + the dense
 * [B,C|S,T] stacking (loader/resampler.py:191-193,213-259) and Shaper.transform
 * (utils/shaper.py:92-106): history pads are never materialised, pad tokens never become rows. */
int lego_plan_batch(const int32_t* cand /*[B,C]*/, const int32_t* hist /*[B,S]*/, const int32_t* hist_len /*[B]*/,
                    int B, int C, int S,
                    const int32_t* title_tok /*[n_items,T], -1 pad*/, const int32_t* title_len /*[n_items]*/, int T,
                    int32_t* counters /*[LEGO_COUNTERS]*/, int32_t* inst_item /*[B*(C+S)]*/,
                    int32_t* seg_off /*[B*(C+S)+1]*/, int32_t* hist_off /*[B+1]*/,
                    int32_t* rowinfo /*[B*(C+S)*T]*/, int32_t* row_tok /*[B*(C+S)*T]*/, void* stream);

/* Dense plan for the operator-level API: n sequences of L positions, mask[n,L] (int32 0/1). */
int lego_plan_dense(const int32_t* mask /*[n,L] or NULL = all live*/, int n, int L,
                    int32_t* counters, int32_t* seg_off /*[n+1]*/, int32_t* rowinfo /*[n*L]*/, void* stream);

/* ---- a4: embedding row gather.  nn.Embedding look-up of EmbeddingHub / Transformation
 * (loader/embedding_hub.py:95-96,378-385); idx < 0 (pad) yields a zero row.  HBM-bound. */
int lego_gather_rows(const float* table, int ld_table, int width, const int32_t* idx, int rows_cap,
                     const int32_t* rows_dyn /*nullable*/, float* out, int ld_out,
                     int accumulate /*1: out[r] += row where idx >= 0 (ConcatInputer's summed look-ups)*/, void* stream);
/* ---- a4, de-duplicated: the projection Linear(glove[tok]) of Transformation (loader/embedding_hub.py:73-96) depends on the
 * token id alone, and a batch repeats tokens (Zipf).  lego_unique_tokens builds, from the plan's row_tok[R]: uniq[U] (distinct
 * ids, ascending), inv[R] (row -> index into uniq), perm[R] (rows grouped by inv), start[U] (first position of each group in
 * perm) and *n_uniq = U.  Workspaces: stamp[V] (zero-initialised ONCE, never cleared; `epoch` must differ from every earlier
 * call's and from 0), rank[V], bsum[ceil(V/1024)], cnt[min(R_cap,V)].  perm / start / cnt may be NULL (no grouping); sort_keys
 * (nullable, [R_cap]) receives inv with the rows past R set to INT32_MAX: the key array lego_sort_rows groups the rows by. */
int lego_unique_tokens(const int32_t* row_tok, int R_cap, const int32_t* R_dyn, int V, uint32_t* stamp, uint32_t epoch,
                       int32_t* rank, int32_t* bsum, int32_t* uniq, int32_t* inv, int32_t* cnt, int32_t* start,
                       int32_t* perm, int32_t* sort_keys, int32_t* n_uniq, void* stream);
/* perm = the row indices 0 .. n-1 stably sorted by keys[r] (radix sort over `key_bits` bits + the sentinel's top bit; rocPRIM
 * through hipCUB: a utility on the prefetch stream, not a hot kernel).  temp: lego_sort_rows_temp_bytes(n) bytes. */
int lego_sort_rows(const int32_t* keys, int n, int32_t* keys_sorted, int32_t* perm, void* temp, int64_t temp_bytes, void* stream);
int64_t lego_sort_rows_temp_bytes(int n);
/* out[r,:] = live_r * dropout_r(src[inv[r],:]) for r < rows: expands per-token projections to token rows, the site's Dropout
 * applied per ROW (embedding_hub.py:95-96: Dropout(Linear(.))); rows whose live bit is 0 in rowinfo are written as zeros (the
 * [SEP] / category positions of a ConcatInputer sequence, concat_inputer.py:96-114: their token look-up is masked) */
/* add_a / idx_a, add_b / idx_b (nullable pairs): out[r,:] += add_a[idx_a[r],:] where idx_a[r] >= 0, the same for b -- the special-id and
 * category look-ups ConcatInputer.get_embeddings sums with the token look-up (concat_inputer.py:96-114) */
int lego_expand_rows(const float* src, int ld_src, const int32_t* inv, int rows_cap, const int32_t* rows_dyn, int width,
                     const lego_dropout* drop, const int32_t* rowinfo /*nullable*/, const float* add_a, int ld_a, const int32_t* idx_a,
                     const float* add_b, int ld_b, const int32_t* idx_b, float* out, int ld_out, void* stream);
/* NRMS, GloVe projection (round 5, csrc/dropcorr_ops.hip): the attention in-projection of nn.MultiheadAttention
 * (attention_operator.py:49-55) once per DISTINCT key with an exact sparse correction for the Dropout that embedding_hub.py:95-96
 * puts in front of it.  qkvu[k,:] = Eu[k,:] W^T (no bias) over the distinct keys; wt = W^T, [D][N] row-major.  For a token row r
 * (live bit of rowinfo[r]) of key k = inv[r]:  out[r,:] = s (qkvu[k,:] - sum over the coordinates c the site DROPPED in row r of
 * eu[k,c] wt[c,:]) + bias,  s = 1 / (1 - p); for the other rows ([SEP] / category positions, no Dropout): out[r,:] = qkvu[k,:] + bias.
 * drop NULL or p == 0: plain expansion.  The keep bits must be precomputed (lego_dropout_mask over [rows, D]).  D <= 256.
 * With Dropout the call is two launches on `stream` (round 6): the rows' dropped coordinates as (offset, value) pairs into a buffer
 * the library keeps per (device, stream) -- rows_cap x roundup(D, 8) x 8 bytes (216 MB at rows_cap = 105 600, D = 256), grown on
 * demand -- then the expansion, which reads them through the scalar unit. */
int lego_qkv_expand_dropcorr(const float* qkvu, int ldq, const float* eu, int lde, const float* wt, int ldw, const float* bias /*nullable*/,
                             const int32_t* inv, const int32_t* rowinfo, const lego_dropout* drop /*nullable*/, int rows_cap,
                             const int32_t* rows_dyn, int D, int N, float* out, int ldo, void* stream);
/* out[u,:] = sum over rows r with inv[r] == u of g[r,:] (u < U; perm = the rows grouped by inv): the per-token sums the
 * projection's weight gradient is formed from.  out[0:U] must be zero on entry: zero_first = 1 clears it here, 0 = the caller
 * has (lego_zero_rows, e.g. on another stream ahead of time). */
int lego_segment_sum_rows(const float* g, int ld_g, int width, const int32_t* perm, const int32_t* inv, int R_cap,
                          const int32_t* sorted_keys /*nullable: inv[perm[p]] per position, lego_sort_rows' key output*/,
                          const int32_t* R_dyn, float* out, int ld_out, int U_cap, const int32_t* U_dyn, int zero_first,
                          const lego_dropout* drop /*nullable; must carry lego_dropout_mask's keep bits: g[r,:] is multiplied by
                                                     keep(r,:) / (1 - p) as it is read -- the Dropout backward of the rows*/,
                          const int32_t* rowinfo /*nullable: rows whose live bit is 0 add nothing*/, void* stream);
int lego_zero_rows(float* out, int ld_out, int width, int rows_cap, const int32_t* rows_dyn /*nullable*/, void* stream);
/* backward of a TRAINABLE table (embed/null.yaml): grad_table[idx[r]] += g[r] (dense grad semantics) */
int lego_scatter_add_rows(float* grad_table, int ld_table, int width, int table_rows /*<= 32: LDS pre-reduction*/,
                          const int32_t* idx, int rows_cap, const int32_t* rows_dyn, const float* g, int ld_g, void* stream);
/* the same for destination rows in [row_lo, row_hi) only: the 410 MB table gradient of embed/null is produced bucket by bucket
 * under data parallelism, so that bucket k is all-reduced while bucket k+1 is scattered (train_step.TrainStep.exchange hooks) */
int lego_scatter_add_rows_range(float* grad_table, int ld_table, int width, const int32_t* idx, int rows_cap,
                                const int32_t* rows_dyn, const float* g, int ld_g, int row_lo, int row_hi, void* stream);

/* ---- a4/a5/a6/a8: Linear layers.  out[M,N] = act(x[M,K] . W[N,K]^T + bias) * live * dropout.
 * act: 0 none (nn.Linear: embedding_hub.py:95, cnn_operator.py:59, attention_operator.py:56),
 *      2 tanh (AdditiveAttention.encoder[0..1], model/common/attention.py:17-21).
 * x_row_off_dyn / out_row_off_dyn: optional device row offsets (category rows live at Y rows R..). */
int lego_linear_fwd(const float* x, int ldx, const float* W, int ldw, const float* bias,
                    float* out, int ldo, int M_cap, const int32_t* M_dyn, int N, int K, int act,
                    const int32_t* rowinfo /*nullable: zero rows whose live bit is 0*/,
                    const lego_dropout* drop /*nullable*/,
                    const int32_t* x_row_off_dyn, const int32_t* out_row_off_dyn, void* stream);
/* dx[M,K] (+)= g[M,N] . W[N,K];  optional fused ReLU-backward (dx = ref>0 ? dx*scale : 0),
 * dropout/live re-masking and column sums of the result (bias gradient of the producer layer). */
int lego_linear_bwd_data(const float* g, int ldg, const float* W, int ldw, float* dx, int lddx,
                         int M_cap, const int32_t* M_dyn, int N, int K, int accumulate,
                         const float* relu_ref, int ld_ref, float relu_scale,
                         const int32_t* rowinfo, const lego_dropout* drop, float* colsum /*[K] nullable, += */,
                         const int32_t* g_row_off_dyn, const int32_t* dx_row_off_dyn, void* stream);
/* dW[N,K] += g[M,N]^T . x[M,K]   (split-K over the ragged row count, fp32 atomics) */
int lego_linear_bwd_weight(const float* g, int ldg, const float* x, int ldx, float* dW, int lddw,
                           int M_cap, const int32_t* M_dyn, int N, int K,
                           const int32_t* g_row_off_dyn, const int32_t* x_row_off_dyn, void* stream);
/* out[c] += sum_r x[r,c] over rows [off, off+M)  (bias gradients not fused elsewhere) */
int lego_colsum(const float* x, int ldx, int M_cap, const int32_t* M_dyn, const int32_t* row_off_dyn,
                int N, float* out, void* stream);

/* ---- a5: CNNOperator title branch (model/operators/cnn_operator.py:54-57): Conv1d(k=3,'same')
 * -> ReLU -> *mask -> Dropout as an implicit GEMM over token rows (taps = rows r-1,r,r+1 of the
 * same item, from rowinfo).  Wt is the tap-major copy [3][Dout][Din] made by lego_conv3_pack. */
int lego_conv3_pack(const float* w /*[Dout,Din,3]*/, float* wt /*[3,Dout,Din]*/, int Dout, int Din, void* stream);
int lego_conv3_unpack_add(float* dwt /*[3,Dout,Din], cleared on return*/, float* dw /*[Dout,Din,3], +=*/, int Dout, int Din, void* stream);
/* Winograd F(2,3) form of the same conv over ROW PAIRS (two thirds of the direct form's MFMA work; same result up to
 * fp32 rounding).  lego_plan_pairs derives the pairs from seg_off: pair_info[p] = first_row << 3 | has_second |
 * has_left << 1 | has_right2 << 2, *n_pairs_out = P.  u / du are the transformed weights [4][Dout][Din]
 * (lego_conv3_wino_pack).  The weight gradient writes its partial results as S = lego_conv3_wino_du_slabs(Dout, Din, P_cap)
 * slabs du[S][4][Dout][Din]: S > 1 (long reductions) = one slab per k split, written whole with plain stores, no fp32
 * atomics and nothing to clear; S == 1 = an atomics accumulator that must be zero on entry.  lego_conv3_wino_unpack_add sums
 * the S slabs, adds the transposed transform into dw and, for S == 1, hands du back cleared.
 * Every planned row must be live (ragged plans).  Din, Dout multiples of 32, <= 256.  Same reference lines as
 * lego_conv3_*: nn.Conv1d(k=3, padding='same') -> ReLU -> Dropout of model/operators/cnn_operator.py:33-38,54-57 and its
 * autograd backward. */
int lego_plan_pairs(const int32_t* seg_off, int n_cap, const int32_t* n_dyn, int32_t* pair_info, int32_t* n_pairs_out,
                    void* stream);
int lego_conv3_wino_pack(const float* w /*[Dout,Din,3]*/, float* u /*[4,Dout,Din]*/, float* ut /*nullable [4,Din,Dout]: u transposed*/,
                         int Dout, int Din, void* stream);
int lego_conv3_wino_du_slabs(int Dout, int Din, int P_cap);
int lego_conv3_wino_unpack_add(float* du /*[S,4,Dout,Din]*/, int n_slabs, float* dw /*[Dout,Din,3], +=*/, int Dout, int Din, void* stream);
int lego_conv3_wino_fwd(const float* h, int ldh, const float* u, const float* bias, const int32_t* pair_info,
                        int P_cap, const int32_t* P_dyn, float* y, int ldy, int Dout, int Din,
                        const lego_dropout* drop, void* stream);
int lego_conv3_wino_bwd_data(const float* gy, int ldg, const float* u, const float* ut /*required (ABI 8): the transposed sets*/,
                             const int32_t* pair_info,
                             int P_cap, const int32_t* P_dyn, float* dh, int lddh, int Dout, int Din,
                             const lego_dropout* drop_in, float* colsum, void* stream);
/* n_slabs = the S the caller sized du for (lego_conv3_wino_du_slabs at that time): a launch that would write another number of slabs
 * -- the process-wide product mode changed in between -- is refused (ABI 8) */
int lego_conv3_wino_bwd_weight(const float* gy, int ldg, const float* h, int ldh, const int32_t* pair_info,
                               int P_cap, const int32_t* P_dyn, float* du /*[S,4,Dout,Din]*/, int n_slabs, int Dout, int Din, void* stream);
int lego_conv3_fwd(const float* h, int ldh, const float* wt, const float* bias, const int32_t* rowinfo,
                   float* y, int ldy, int R_cap, const int32_t* R_dyn, int Dout, int Din,
                   const lego_dropout* drop, int mask_rows /*0: every row is live (ragged plan), skip the live-bit loads*/,
                   void* stream);
/* dh = live*dropout_in * sum_tap gy[r-(tap-1)] . Wt[tap];  colsum(dh) -> bias grad of the input projection */
int lego_conv3_bwd_data(const float* gy, int ldg, const float* wt, const int32_t* rowinfo,
                        float* dh, int lddh, int R_cap, const int32_t* R_dyn, int Dout, int Din,
                        const lego_dropout* drop_in, float* colsum, int mask_rows, void* stream);
/* dwt[tap] += gy^T . h[r+tap-1] */
int lego_conv3_bwd_weight(const float* gy, int ldg, const float* h, int ldh, const int32_t* rowinfo,
                          float* dwt, int R_cap, const int32_t* R_dyn, int Dout, int Din, void* stream);

/* ---- a6/a7: AdditiveAttention pooling (model/common/attention.py:31-38) over ragged segments.
 * t = tanh(W1 x + b1) rows come from lego_linear_fwd(act=2).  Per segment i (rows seg_off[i]..
 * seg_off[i+1]) plus optionally one extra row (extra_off_dyn + i: the category row):
 *   a = t.w2; e = exp(a)*live (no max-subtraction); w = e/(sum e + 2^-23); out = sum w x. */
int lego_additive_pool_fwd(const float* t, int ldt, const float* x, int ldx, const float* w2,
                           const int32_t* seg_off, const int32_t* rowinfo /*nullable*/,
                           const int32_t* extra_off_dyn /*nullable*/, int n_cap, const int32_t* n_dyn,
                           int D, int A, float* out, int ldo, float* wrow /*[rows] saved weights*/, void* stream);
/* given gout[n,D]: dx rows = w*gout (written, not accumulated); t is overwritten in place with
 * dpre = da*w2*(1-t^2); gw2[A] += sum da*t ; gb1[A] += sum dpre.
 * scratch (nullable): LEGO_POOL_SCRATCH(A) zero-initialised floats owned by the caller, one per stream that calls this:
 * the per-workgroup partials of gw2 / gb1 are spread over 32 copies there instead of ~1000 workgroups adding to the same
 * 2*A words; lego_additive_pool_bwd_fold then adds the copies into gw2 / gb1 and hands the scratch back zeroed -- a
 * separate call so that it can run on another stream, off the critical chain (it must be ordered after the pool
 * backward and before the scratch is used again). */
#define LEGO_POOL_SCRATCH(A) (32 * 2 * (A))
int lego_additive_pool_bwd(float* t_dpre, int ldt, const float* x, int ldx, const float* w2,
                           const int32_t* seg_off, const int32_t* extra_off_dyn, int n_cap, const int32_t* n_dyn,
                           int D, int A, const float* gout, int ldgo, const float* wrow,
                           float* dx, int lddx, float* gw2, float* gb1, float* scratch, void* stream);
int lego_additive_pool_bwd_fold(float* scratch, int A, float* gw2, float* gb1, void* stream);

/* ---- a9/a10: DotPredictor + CrossEntropy(label 0) (model/predictors/dot_predictor.py:7-10,
 * model/operators/base_operator.py:65-69, model/legommender.py:254,263,268-283). */
int lego_dot_ce_fwd(const float* user /*[B,D]*/, int ldu, const float* items /*[B*C,D]*/, int ldi,
                    int B, int C, int D, float* scores /*[B,C]*/, float* loss /*[1], +=mean*/, void* stream);
int lego_dot_ce_bwd(const float* user, int ldu, const float* items, int ldi, const float* scores,
                    int B, int C, int D, float gscale /*dloss * 1/B*/,
                    const float* gscale_dev /*nullable [1]: multiplied into gscale ON THE DEVICE -- autograd's upstream gradient of the loss
                    without a host read*/, float* guser, int ldgu, float* gitems, int ldgi, void* stream);

/* ---- a7 + a9 + a10 fused for TRAINING: AdaOperator pool over each user's clicked-item vectors (rows
 * items[B*C + hist_off[b] ..]), dot scores against the user's C candidates (rows items[b*C ..]), CE(label 0), the
 * gradient of the scores, and the backward of the pool -- one workgroup per impression instead of five dependent
 * launches (ada_operator.py:31-34, dot_predictor.py:7-10, legommender.py:254,263 and their autograd).
 * t_dpre rows (tanh hidden of the history rows) are overwritten with dpre; d_items gets g_c*u for candidates and
 * w_l*du for history rows (the dpre.W1 term is added by lego_linear_bwd_data); gw2/gb1 accumulate. */
int lego_user_tower_train(float* t_dpre, int ldt, const float* items, int ldi, const float* w2,
                          const int32_t* hist_off, int B, int C, int S, int D, int A, float gscale /* dloss / B */,
                          float* user /*[B,D]*/, float* scores /*[B,C]*/, float* loss /*[1] += mean*/,
                          float* d_items, int lddi, float* gw2, float* gb1, void* stream);

/* DotPredictor.predict on already-expanded (user, item) pairs, as the plug-in interface passes them
 * (model/legommender.py:275-283): out[r] = sum_d u[r,d]*it[r,d], and its backward. */
int lego_rowdot_fwd(const float* u, int ldu, const float* it, int ldi, int n, int D, float* out, void* stream);
int lego_rowdot_bwd(const float* u, int ldu, const float* it, int ldi, const float* g, int n, int D,
                    float* gu, int ldgu, float* gi, int ldgi, void* stream);
/* g = ref > 0 ? g*scale : 0 -- backward of ReLU/mask/Dropout of CNNOperator (cnn_operator.py:55-57) given the
 * stored output `ref`, for callers outside the fused GEMM epilogue (operator-level API) */
int lego_relu_bwd(float* g, int ldg, const float* ref, int ldr, int rows, int width, float scale, void* stream);

/* ---- a8: nn.MultiheadAttention core of AttentionOperator (model/operators/attention_operator.py:49-55)
 * over ragged segments: per (segment, head) softmax(q k^T / sqrt(hd)) v with all keys of the segment
 * live (pads are not rows).  qkv rows are [q | k | v] (3*D) from lego_linear_fwd with in_proj. */
/* `part`: which segments a call handles -- all, only those of <= 32 rows, or only the longer ones.  The two groups run as separate
 * launches that touch disjoint rows; a caller may put them on two streams (the long-segment launch is latency-bound and hides
 * behind the HBM-bound short-segment one). */
#define LEGO_MHSA_ALL 0
#define LEGO_MHSA_SHORT 1
#define LEGO_MHSA_LONG 2
#define LEGO_MHSA_ALL_LONG 3 /* every segment through the two-wave (<= 64 rows) instantiation: one launch; for a few hundred pairs (the user side) */
/* long_list / long_count (nullable, together): the segments of more than 32 rows and their number, from lego_mhsa_long_segments on
 * the same seg_off -- the long-segment launch then gives every (segment, head) pair its own workgroup instead of searching for them */
int lego_mhsa_long_segments(const int32_t* seg_off, int n_cap, const int32_t* n_dyn, int32_t* list /*[n_cap]*/, int32_t* count /*[1]*/,
                            void* stream);
/* ABI v5: two ways to hand the softmax to the backward pass, chosen by which pointer is given (both may be; the backward pass then
 * uses `probs`):
 *   lse   [rows, heads]        log-sum-exp of every query row's scaled scores; the backward pass recomputes S = Q K^T / sqrt(hd),
 *                              p = exp(s - lse) and redraws the dropout keep bits from `drop` (the SAME p / seed / site as the forward call)
 *   probs [rows, heads, Lmax]  the probabilities themselves: tile of (segment, head) at ((beg * heads + h * L) * Lmax),
 *                              [key j][query i], sign bit set = dropped; read back by the backward pass (no random numbers there). */
int lego_mhsa_core_fwd(const float* qkv, int ldq, const int32_t* seg_off, int n_cap, const int32_t* n_dyn,
                       int D, int heads, float* out, int ldo, float* lse /*nullable*/, float* probs /*nullable*/,
                       int Lmax, const lego_dropout* drop, int rows_cap, int part, const int32_t* long_list, const int32_t* long_count,
                       void* stream);
int lego_mhsa_core_bwd(const float* qkv, int ldq, const int32_t* seg_off, int n_cap, const int32_t* n_dyn,
                       int D, int heads, const float* gout, int ldgo, const float* lse /*nullable*/, const float* probs /*nullable*/, int Lmax,
                       const lego_dropout* drop, int rows_cap, float* gqkv, int ldgq,
                       float* colsum /*nullable [3*D]: += column sums of gqkv = the in_proj_bias gradient*/, int part,
                       const int32_t* long_list, const int32_t* long_count, void* stream);

/* ---- a8 (engine route): the two affine layers behind the attention core -- nn.MultiheadAttention's out_proj and
 * AttentionOperator.linear (attention_operator.py:49-56, nothing between them) -- and the hidden layer of the additive attention
 * that reads their output (model/common/attention.py:31-34) folded into ONE weight each:
 *   Wc = Wl Wo [D,D], bc = Wl bo + bl [D];   W2 = W1 Wc [A,D], b2 = W1 bc + b1 [A]   (W2 / b2 / W1 / b1 nullable: first fold only).
 * lego_attn_fold_grads is the backward of that parameter map, given
 *   Tp = dpre^T o [A,D], sp = colsum(dpre) [A]           (W1 nullable: skipped)
 *   T  = d(loss)/dWc so far (d_out^T pooled, or d_lin^T o for the first fold) [D,D],  s = d(loss)/dbc so far [D]:
 *   gW1 += Tp Wc^T + sp (x) bc, gb1 += sp, T += W1^T Tp, s += sp W1;
 *   gWl += T Wo^T + s (x) bo, gbl += s, gWo += Wl^T T, gbo += s Wl.      T and s are updated in place.
 * One launch per dependency level (two each): one wave per 16 x 16 output tile, operands straight from L2. */
int lego_attn_fold_prepare(const float* Wo, const float* bo, const float* Wl, const float* bl, const float* W1,
                           const float* b1, float* Wc, float* bc, float* W2, float* b2, int D, int A, void* stream);
int lego_attn_fold_grads(const float* Wo, const float* bo, const float* Wl, const float* W1, const float* Wc,
                         const float* bc, const float* Tp, const float* sp, float* T, float* s,
                         float* gWo, float* gbo, float* gWl, float* gbl, float* gW1, float* gb1,
                         int D, int A, void* stream);

/* NRMS user head of a TRAINING step in one launch (engine route, folded attention block): user vector from the pooled attention
 * output (u = Wc p + bc: attention_operator.py:52-56 with the two affine layers folded), DotPredictor over the C candidates
 * (dot_predictor.py:7-10), CrossEntropy with label 0 (legommender.py:254,263), and their backward: loss += mean CE,
 * d_user[b] = sum_c g_c item_c, d_items[b*C + c] = g_c u_b (written), d_pooled = Wc^T d_user, g = (softmax - e_0) * gscale. */
int lego_nrms_user_head_train(const float* pooled, int ldp, const float* Wc, const float* bc, const float* items, int ldi,
                              int B, int C, int D, float gscale /* dloss / B */, float* user, int ldu, float* scores /*[B,C]*/,
                              float* loss /*[1] += mean, nullable*/, float* d_user, int lddu, float* d_items, int lddi,
                              float* d_pooled, int lddp,
                              const int32_t* seg_off /*nullable [B+1]: a user whose segment is EMPTY gets u = 0 and d_user = 0 (what the
                              un-folded operator yields for it) instead of u = bc*/, void* stream);

/* ---- a13: torch.optim.Adam (defaults, base_lego.py:201-204) over one flat fp32 buffer;
 * grad is multiplied by grad_scale first (1/world after the RCCL all-reduce). step is 1-based.
 * zero_grad != 0: g is cleared as it is consumed (optimizer.zero_grad() of the next step, trainer.py:199). */
int lego_adam_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                   float eps, int step, float grad_scale, int zero_grad, void* stream);
/* The same update over the rows of a trainable [rows, width] embedding table (config/embed/null.yaml: nn.Embedding(400k, D),
 * loader/embedding_hub.py:325-335 -- dense gradient, dense Adam), skipping rows whose byte in `touched` is 0.  A row that
 * has never received a gradient has g = m = v = 0, where Adam's update is exactly 0: skipping it is bit-identical to the
 * dense rule, and once touched a row is updated every step.  lego_mark_rows sets the bytes of idx[0..n) (ids < 0 ignored);
 * data-parallel ranks merge their flags (all-reduce MAX) with the gradient all-reduce. */
int lego_adam_step_rows(float* p, float* g, float* m, float* v, int rows, int width, const uint8_t* touched /*[rows]*/,
                        float lr, float beta1, float beta2, float eps, int step, float grad_scale, int zero_grad, void* stream);
int lego_mark_rows(const int32_t* idx, int n_cap, const int32_t* n_dyn, int rows, uint8_t* touched, void* stream);

/* ---- a11: negative sampling of Resampler.rebuild_candidates (loader/resampler.py:159-171) on
 * device: cand[b,0] = positive; min(K,len) distinct draws from the user's true-negative list,
 * the rest uniform item ids in [0,n_items).  The Philox stream of row b is keyed on
 * (seed, step, row_base + b * row_stride): data-parallel rank r of W passes (r, W) -- the row's position in
 * the global batch of the step -- so W ranks x B rows draw what one device with batch W*B draws (0, 1).
 * row_pos != NULL: the position of row b is row_pos[b] instead (ranks that deal the rows of a global batch by their live-row
 * cost rather than r::W, train_step.DeviceData(balance=B)). */
int lego_sample_negatives(const int32_t* row_user /*[B]*/, const int32_t* row_item /*[B]*/,
                          const int32_t* neg_list /*[n_users,neg_cap]*/, const int32_t* neg_len, int neg_cap,
                          int B, int K, int n_items, uint64_t seed, uint32_t step, uint32_t row_base,
                          uint32_t row_stride, const int32_t* row_pos /*[B] or NULL*/, int32_t* cand /*[B,K+1]*/,
                          void* stream);
/* hist[b,:] / hist_len[b] = user tables rows of row_user[b]  (the DataSet row copy, data_set.py:61-85) */
int lego_gather_history(const int32_t* row_user, const int32_t* user_hist /*[n_users,S]*/,
                        const int32_t* user_hist_len, int B, int S, int32_t* hist, int32_t* hist_len, void* stream);

/* ---- a3': ConcatInputer layout of NRMS (model/inputer/concat_inputer.py:58-114).  The item sequence
 * [title..., SEP, category, SEP] is planned with lego_plan_batch over an encoded per-item sequence table
 * (token id >= 0, SEP = -2, category = -(3+cat)); this splits the plan's row words into the three
 * look-up index columns (-1 = column absent at that position) and a live-bit word for token rows. */
int lego_nrms_decode_rows(const int32_t* row_tok, int R_cap, const int32_t* R_dyn, int32_t* idx_tok,
                          int32_t* idx_special, int32_t* idx_cat, int32_t* tokinfo, void* stream);
/* (round 5) key space of the per-key in-projection: row_key[r] = row_tok[r] for tokens, V for [SEP] (-2), V + 1 + c for category c
 * (-(3 + c)) -- every position of a ConcatInputer sequence (concat_inputer.py:96-114) is a function of that one id; and, for the
 * DISTINCT keys lego_unique_tokens finds in that space, the row index in each of the three tables (-1 = not this table's; [SEP] is
 * row 2 of the special-id table) + RI_LIVE for token keys. */
int lego_nrms_key_rows(const int32_t* row_tok, int R_cap, const int32_t* R_dyn, int V, int32_t* row_key, void* stream);
int lego_nrms_decode_keys(const int32_t* uniq, int U_cap, const int32_t* U_dyn, int V, int32_t* idx_tok, int32_t* idx_special,
                          int32_t* idx_cat, int32_t* keyinfo, void* stream);
/* Backward of ConcatInputer's two small look-ups (concat_inputer.py:96-114: token + special-id + category embeddings are summed
 * per position): an item's sequence is [title..., SEP, category, SEP], so rows L-3 and L-1 of each segment add into the [SEP]
 * row g_sep[width] of the special-id table and row L-2 into row idx_cat[beg+L-2] of g_cat. */
int lego_nrms_special_grads(const int32_t* seg_off, int n_cap, const int32_t* n_dyn, const int32_t* idx_cat, const float* g,
                            int ld, int width, float* g_sep /*row SEP of the special table*/, float* g_cat, int ld_cat, int n_cat,
                            void* stream);
/* x[r,:] *= live(rowinfo[r]) * dropout-scale: backward of `Transformation`'s Dropout + the inputer mask
 * (loader/embedding_hub.py:96, concat_inputer.py:111) when the producer is not a fused GEMM epilogue.
 * colsum (nullable, [width], +=): column sums of the masked result -- the bias gradient of the projection
 * (`Transformation.linear.bias`), as lego_colsum on the masked rows. */
int lego_mask_dropout_rows(float* x, int ld, int R_cap, const int32_t* R_dyn, int width, const int32_t* rowinfo,
                           const lego_dropout* drop, float* colsum, void* stream);

/* ---- a14: grouped ranking metrics of the evaluation path -- MetricPool.calculate's per-group loop
 * (utils/metrics.py:313-369; pandas groupby + Pool(5) of Python metric calls) as one launch.
 * Rows are sorted by group (stable): group g owns rows [group_off[g], group_off[g+1]).  `ks` is a HOST array of n_k
 * cut-offs (<= LEGO_METRIC_MAX_K).  out is [4 + 3*n_k][n_groups] fp32, per group:
 *   row 0 GAUC term  roc_auc_score          (utils/metrics.py:95-96,98-107; NaN when one class only -- sklearn raises)
 *   row 1 MRR        mean over positives of 1/rank   (:147-160; NaN without positives -- the reference divides by 0)
 *   row 2 MRR0       1/rank of the first positive, 0 without one  (:125-140)
 *   row 3 LRAP       label_ranking_average_precision_score        (:108-117)
 *   row 4+3q NDCG@k  sklearn ndcg_score(k), ties averaged          (:216-229)
 *   row 5+3q HitRatio@k (:183-197)      row 6+3q Recall@k (:199-214; NaN without positives)
 * rank = position after a stable descending sort of the group's scores (Python `sorted(..., reverse=True)`).
 * The caller takes the fp32 mean over groups (utils/metrics.py:367). */
#define LEGO_METRIC_MAX_K 8
int lego_grouped_metrics(const float* scores, const int32_t* labels, const int32_t* group_off, int n_groups,
                         const int32_t* ks /*host*/, int n_k, float* out, void* stream);

/* ---- 8f-2 (config 5): the row-wise pieces of a BERT block around the path's products and attention core -- the `transformers`
 * BertSelfOutput / BertOutput tail `LayerNorm(Dropout(dense(x)) + residual)`, BertEmbeddings' `Dropout(LayerNorm(sum of embeddings))`
 * and BertIntermediate's exact GELU (reference call sites: model/operators/once_operator.py:156-193, bert_operator.py:10-52).
 *   out = drop_post(LayerNorm(drop_pre(y) + resid) * gamma + beta);   mean / rstd [rows] are saved for the backward pass, which
 *   forms v = drop_pre(y) + resid again from y, resid and the redrawn keep bits.  resid, drop_pre, drop_post nullable.
 *   backward: dy (nullable) = d(loss)/dy, dresid (nullable) = d(loss)/dresid, dgamma / dbeta (nullable) += their gradients. */
int lego_dropout_add_layernorm_fwd(const float* y, int ldy, const float* resid, int ldr, const float* gamma, const float* beta, float eps,
                                   const lego_dropout* drop_pre, const lego_dropout* drop_post, float* out, int ldo,
                                   float* mean /*[rows]*/, float* rstd /*[rows]*/, int rows, int width, void* stream);
int lego_dropout_add_layernorm_bwd(const float* dout, int lddo, const float* y, int ldy, const float* resid, int ldr, const float* gamma,
                                   const float* mean, const float* rstd, const lego_dropout* drop_pre, const lego_dropout* drop_post,
                                   float* dy, int lddy, float* dresid, int lddr, float* dgamma, float* dbeta,
                                   float* dybias /*nullable [width], += column sums of dy: the bias gradient of the layer that produced y*/,
                                   int rows, int width, void* stream);
/* g = z Phi(z) (erf form) over n contiguous floats; dz = dg (Phi(z) + z phi(z)), dz may alias dg */
int lego_gelu_fwd(const float* z, float* g, int64_t n, void* stream);
/* (ABI 8) BertIntermediate / BertOutput.dense of the BERT news encoder (reference model/operators/bert_operator.py:16 ->
 * transformers modeling_bert.py BertIntermediate.forward, BertOutput.forward) with the GELU inside the product's epilogue:
 *   lego_linear_gelu_fwd       z[M,N] = x[M,K] W[N,K]^T + bias  (kept for the backward pass),  g[M,N] = gelu(z)
 *   lego_linear_bwd_data_gelu  dz[M,K] = (dy[M,N] W[N,K]) * gelu'(z[M,K])
 * -- the same numbers as lego_linear_fwd + lego_gelu_fwd and lego_linear_bwd_data + lego_gelu_bwd (one rounding sequence), without the
 * second pass over the [M, 3072] tensors. */
int lego_linear_gelu_fwd(const float* x, int ldx, const float* W, int ldw, const float* bias /*nullable*/, float* z, int ldz, float* g, int ldg,
                         int M, int N, int K, void* stream);
int lego_linear_bwd_data_gelu(const float* dy, int ldy, const float* W, int ldw, const float* z, int ldz, float* dz, int lddz,
                              int M, int N, int K, void* stream);
int lego_gelu_bwd(const float* dg, const float* z, float* dz, int64_t n, void* stream);

/* small utilities used by the host side */
int lego_gather_i32(const int32_t* table, const int32_t* idx, int n_cap, const int32_t* n_dyn, int32_t* out, void* stream);
/* rowinfo[i] = live bit (4) iff segment i of seg_off has rows: with it the live-mask epilogue of lego_linear_fwd zeroes the output
 * rows of EMPTY segments (folded NRMS user vector of a user without clicked items: 0, as the un-folded operator, not the bias) */
int lego_segment_live(const int32_t* seg_off, int n_cap, const int32_t* n_dyn, int32_t* rowinfo /*[n_cap]*/, void* stream);

#ifdef __cplusplus
}
#endif
#endif
