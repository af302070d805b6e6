/*
 * slu_hip.h — C ABI of libslu_hip.so: hand-written HIP kernels (gfx950 / CDNA4, MI355X) for the
 * SincNet-conv + stacked-biGRU speech-encoder hot path of lorenlugosch/end-to-end-SLU.
 *
 * The reference has no FFI of its own: its "kernel launches" are the PyTorch operator call
 * sites in models.py (F.conv1d :108, nn.Conv1d :190/:200, nn.MaxPool1d :205, nn.LeakyReLU :211,
 * nn.GRU :232/:262/:686, nn.Dropout :246/:276/:700, F.avg_pool1d/max_pool1d :44/:46).  Each entry
 * point below names the call site(s) it replaces.  A maintainer of the reference binds them with
 * ctypes exactly as end-to-end-slu_amd/slu_hip/lib.py does (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns 0 on success or a negative slu_status; slu_last_error() returns a
 *     thread-local message for the last failure on the calling thread;
 *   - all data pointers are DEVICE pointers to caller-owned buffers (16-byte aligned, as
 *     torch's allocator guarantees); nothing is retained after the call returns;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it, no
 *     entry point synchronises the device, allocates or frees (safe under hipGraph capture);
 *   - scratch memory is supplied by the caller: query with the matching *_workspace_bytes;
 *   - sizes are int64_t, strides are in ELEMENTS;
 *   - activation layouts: waveform (B,T); CNN activations channels-last (B,L,C) or, through the
 *     output strides, time-major (L,B,C); RNN activations TIME-MAJOR (T,B,C).  The host mirror
 *     presents them to callers in the reference's (B,C,L)/(B,T,C) shapes as strided views.
 */
#ifndef SLU_HIP_H
#define SLU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  SLU_OK = 0,
  SLU_ERR_INVALID_ARG = -1,   /* bad size / null pointer / unsupported combination        */
  SLU_ERR_UNSUPPORTED = -2,   /* shape outside what the gfx950 kernels are instantiated for */
  SLU_ERR_HIP = -3,           /* a HIP runtime call failed (message has hipGetErrorString)  */
  SLU_ERR_WORKSPACE = -4,     /* workspace too small                                         */
  SLU_ERR_DEVICE = -5         /* current device is not gfx950                                */
} slu_status;

#define SLU_ABI_VERSION 9

/* -------- library ------------------------------------------------------------------------- */
int slu_version(void);                    /* returns SLU_ABI_VERSION                           */
const char* slu_last_error(void);         /* thread-local, never NULL                          */
int slu_device_check(void);               /* 0 iff the current HIP device is gfx950            */
const char* slu_device_arch(void);        /* gcnArchName of the current device ("" on failure) */
/* HIP stream restricted to CUs [first_cu, first_cu + n_cus) (mask bit i lands on XCD i mod 8, so a
 * contiguous range is spread over all XCDs); look-ahead pipeline of training.Trainer. Never freed. */
int slu_stream_create_cu_range(int64_t first_cu, int64_t n_cus, void** stream_out);

/* dst[0..count) = values[0..count) (count <= 64 (ABI 9; 32 before); `values` is a HOST array travelling in the kernel arguments): the
 * row-pointer table (slu_wconv_fwd_bf16 in_table) and the dropout-stream offset of a captured super-batch.          */
int slu_store_u64(uint64_t* dst, const uint64_t* values, int64_t count, void* stream);
/* One launch for the per-step input refresh of a captured step: up to 4 strided 2-D copies (rows x row_bytes
 * from src + r * src_stride_bytes to a dense dst) and *set_ptr = set_value (set_ptr may be NULL).  Pointer
 * arrays are HOST arrays of device pointers.                                                                   */
int slu_stage_inputs(const void* const* src, void* const* dst, const int64_t* rows, const int64_t* row_bytes,
                     const int64_t* src_stride_bytes, int64_t count, int64_t* set_ptr, int64_t set_value,
                     void* stream);

/* Up to slu_multi_max() small device-to-device copies in ONE launch (HOST arrays of device pointers and byte counts,
 * travelling in the kernel arguments; every segment a whole number of 4-byte words): packing the fresh gradients of a
 * step into the flat all-reduce bucket.  slu_scale_multi: x_k[i] *= *scale_dev for up to slu_multi_max() fp32 tensors
 * (an upstream scalar applied to the few gradients of a loss head).                                                    */
int slu_multi_max(void);
int slu_copy_multi(const void* const* src, void* const* dst, const int64_t* nbytes, int64_t count, void* stream);
int slu_scale_multi(float* const* ptrs, const int64_t* numel, int64_t count, const float* scale_dev, void* stream);
/* Range words (ABI 5): words[k] = max(words[k], IEEE bit pattern of max |x| over tensor k) for up to slu_multi_max() fp32
 * tensors in one launch — an unsigned integer maximum of the sign-stripped patterns, so NaN / infinity rank above every
 * finite value.  The guard of the f16x2 split scheme (values must stay below 65504 = pattern 0x477FE000) checks a model's
 * frozen weights with it once per weight version; the same word format is written by slu_wconv_fwd_bf16(absmax_word).  The
 * reference needs no counterpart: its fp32 ATen kernels (models.py:108, :200, :232) have fp32's range.                   */
int slu_absmax_multi(const float* const* ptrs, const int64_t* numel, int64_t count, uint32_t* words, void* stream);
/* out[i] = in[i] * scale (ABI 5): PCM16 samples to fp32 for a first block on the exact fp32 kernels (a trainable Sinc
 * layer); the split-precision first block reads the samples itself (slu_wconv_fwd_bf16 in_pcm16).                      */
int slu_pcm16_to_f32(const int16_t* in, float* out, int64_t n, float scale, void* stream);

/* -------- Sinc filterbank: models.py:79-106 (SincLayer.forward up to the conv), :7-24 ------- */
/* filters[n_filt][filt_dim] (float32) from the two float64 parameters, filt_dim odd.            */
int slu_sinc_filters_fwd(const double* filt_b1, const double* filt_band, float* filters,
                         int64_t n_filt, int64_t filt_dim, double fs, void* stream);
/* d(filt_b1), d(filt_band) (float64, what torch autograd leaves in .grad of the f64 params)
 * from d(filters).                                                                               */
int slu_sinc_filters_bwd(const double* filt_b1, const double* filt_band, const float* d_filters,
                         double* d_filt_b1, double* d_filt_band,
                         int64_t n_filt, int64_t filt_dim, double fs, void* stream);

/* -------- windowed convolution block --------------------------------------------------------
 * One fused launch for [conv -> (+bias) -> (abs) -> MaxPool1d(pool, ceil_mode) -> LeakyReLU(slope)]
 * i.e. models.py:108 + :163-168 + :205 + :211 for the Sinc layer (c_in = 1, weights = the
 * filterbank, do_abs = 1, pool = 2) and models.py:200 + :205 + :211 for the dense Conv1d layers.
 *   in      (B, l_in, c_in) channels-last, contiguous (the waveform is l_in = T, c_in = 1)
 *   weight  (c_out, c_in, k_t) — torch Conv1d layout; padding = k_t / 2 (models.py:186,200)
 *   bias    (c_out) or NULL
 *   out     element (b, l, c) at out[b*out_sb + l*out_sl + c]; l_out = ceil(l_conv / pool),
 *           l_conv = (l_in + 2*(k_t/2) - k_t) / stride_t + 1
 *   route   NULL or uint8 (B, l_out, c_out) contiguous: bit0 = index of the max inside the pool
 *           window, bit1 = conv value was negative before abs (needed by slu_wconv_bwd_act)
 *   slope   LeakyReLU negative slope (0.2), 0.0 for ReLU, 1.0 for "no activation"
 * pool must be 1 or 2, c_out <= 128.                                                            */
size_t slu_wconv_workspace_bytes(int64_t c_out, int64_t c_in, int64_t k_t);
int slu_wconv_fwd(const float* in, const float* weight, const float* bias, float* out,
                  uint8_t* route, int64_t B, int64_t l_in, int64_t c_in, int64_t c_out,
                  int64_t k_t, int64_t stride_t, int do_abs, int pool, float slope,
                  int64_t out_sb, int64_t out_sl, void* workspace, size_t workspace_bytes,
                  void* stream);
/* d(conv output) (B, l_conv, c_out) contiguous from d(out): undoes activation, pooling, abs.
 *   dy / y  element (b,l,c) at [b*sb + l*sl + c] (same strides for both); route as written by fwd
 *           (may be NULL when pool == 1 and do_abs == 0).                                        */
int slu_wconv_bwd_act(const float* dy, const float* y, const uint8_t* route, float* d_conv,
                      int64_t B, int64_t l_conv, int64_t c_out, int do_abs, int pool, float slope,
                      int64_t sb, int64_t sl, void* stream);
/* d(in) (B, l_in, c_in) from d_conv; stride_t must be 1 (the only strided layer of the
 * reference architecture is the first one, whose input needs no gradient).                      */
int slu_wconv_bwd_data(const float* d_conv, const float* weight, float* d_in,
                       int64_t B, int64_t l_in, int64_t c_in, int64_t c_out, int64_t k_t,
                       void* workspace, size_t workspace_bytes, void* stream);
/* d(weight) (c_out, c_in, k_t) and d(bias) (c_out; may be NULL) from d_conv and the layer input. */
size_t slu_wconv_bwd_weight_workspace_bytes(int64_t B, int64_t l_in, int64_t c_in, int64_t c_out,
                                            int64_t k_t, int64_t stride_t);
int slu_wconv_bwd_weight(const float* d_conv, const float* in, float* d_weight, float* d_bias,
                         int64_t B, int64_t l_in, int64_t c_in, int64_t c_out, int64_t k_t,
                         int64_t stride_t, void* workspace, size_t workspace_bytes, void* stream);

/* -------- fp32 MFMA GEMM (GRU input projections and their gradients; nn.GRU's x @ W_ih^T) ----
 * C(m,n) = [C(m,n) if accumulate] + sum_k A(m,k) * B(k,n) + (bias_n ? bias_n[n] : 0)
 * with A(m,k) = A[m*a_rs + k*a_cs], B(k,n) = B[k*b_rs + n*b_cs], C(m,n) = C[m*c_rs + n*c_cs].
 * Exact fp32 (v_mfma_f32_16x16x4_f32), split-K through the workspace when M*N is small.          */
size_t slu_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);
int slu_gemm_f32(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs,
                 int64_t b_cs, float* C, int64_t c_rs, int64_t c_cs, const float* bias_n,
                 int64_t M, int64_t N, int64_t K, int accumulate, void* workspace,
                 size_t workspace_bytes, void* stream);
/* Up to four small weight-gradient GEMMs in ONE launch, no workspace, deterministic:
 * C_q (M_q x N_q, row stride ldc) = A_q^T B_q with A_q (K_q x M_q, row stride lda), B_q (K_q x N_q, row stride ldb)
 * — dW_ih = d_gx^T x and dW_hh = d_gh^T h_prev (per direction) of one GRU layer when T*B is a few thousand rows
 * (the generic kernel would need split-K and a reduce launch per matrix).  Every M a multiple of 3, or every M and lda a
 * multiple of 4 with A 16-byte aligned (48- or 64-row output tiles); N, ldb even, B 8-byte aligned.  Pointer / size
 * arrays are HOST arrays.  Optional extra job of the same launch (rowsum_src != NULL):
 * rowsum_dst[c] = sum_r rowsum_src[r*rowsum_cols + c], rows added in order — the layer's bias gradients from the
 * per-tile partial sums slu_gru_seq_bwd leaves (d_bias_part), instead of a reduce launch.                          */
int slu_gemm_tn_batched(const float* const* A, const int64_t* lda, const float* const* B, const int64_t* ldb,
                        float* const* C, const int64_t* ldc, const int64_t* M, const int64_t* N, const int64_t* K,
                        int64_t count, const float* rowsum_src, int64_t rowsum_rows, int64_t rowsum_cols,
                        float* rowsum_dst, void* stream);
/* The same contract for LONG k ranges (K = T * B of several thousand rows: the weight gradients of the phoneme / word GRU
 * layers when they train — reference models.py:232 / :262 through autograd): 64 x 64 output tiles whose k range is also
 * split over workgroups; a tile's last workgroup to arrive adds the partial tiles in a fixed order (deterministic, no
 * reduce launch).  Every M, lda, N, ldb a multiple of 4, A and B 16-byte aligned.  workspace: at least
 * slu_gemm_tn_splitk_workspace_bytes(M, N, K, count) bytes (uninitialised); tickets: one 32-bit word per 64 x 64 output
 * tile of the call (sum over problems of ceil(M / 64) * ceil(N / 64)), ZERO at entry and left zero — a buffer the caller
 * zeroes once and reuses for launches on ONE stream.                                                                 */
size_t slu_gemm_tn_splitk_workspace_bytes(const int64_t* M, const int64_t* N, const int64_t* K, int64_t count);
int slu_gemm_tn_batched_splitk(const float* const* A, const int64_t* lda, const float* const* B, const int64_t* ldb,
                               float* const* C, const int64_t* ldc, const int64_t* M, const int64_t* N, const int64_t* K,
                               int64_t count, const float* rowsum_src, int64_t rowsum_rows, int64_t rowsum_cols,
                               float* rowsum_dst, void* workspace, size_t workspace_bytes, uint32_t* tickets,
                               int64_t n_tickets, void* stream);
/* The same with a workgroup BUDGET (ABI 8): max_workgroups > 0 caps tiles x splits (default 512 = one round at two per CU).
 * For a launch that runs on a graph branch of its own beside a latency-bound recurrence (the weight gradients of GRU layer
 * l beside the BPTT of layer l - 1): 192 - 216 workgroups spread one per CU and leave whole CUs empty for the recurrence's
 * 32 workgroups.  The split count decides the summation order: the same budget gives the same bits.                      */
size_t slu_gemm_tn_splitk_workspace_bytes_wg(const int64_t* M, const int64_t* N, const int64_t* K, int64_t count,
                                             int64_t max_workgroups);
int slu_gemm_tn_batched_splitk_wg(const float* const* A, const int64_t* lda, const float* const* B, const int64_t* ldb,
                                  float* const* C, const int64_t* ldc, const int64_t* M, const int64_t* N,
                                  const int64_t* K, int64_t count, const float* rowsum_src, int64_t rowsum_rows,
                                  int64_t rowsum_cols, float* rowsum_dst, void* workspace, size_t workspace_bytes,
                                  uint32_t* tickets, int64_t n_tickets, int64_t max_workgroups, void* stream);
/* out[n] = [out[n] if accumulate] + sum_m X[m*x_rs + n]   (bias gradients)                       */
/* Up to four independent SMALL-M products in one launch (latency-bound shapes: the seq2seq decoder's per-step Linear /
 * GRUCell products and their data gradients, models.py:427-485): C_q (M x N) = [C_q +] A_q (M x K, row stride lda, k fast)
 * * op(B_q) + bias_q; mode 0: B is (N x K) with row stride ldb (a weight as stored, C = A B^T), mode 1: B is (K x N) with
 * row stride ldb (C = A B).  Exact fp32 MFMA, eight waves split K, deterministic.  K % 4 == 0, lda % 4 == 0, A 16-byte
 * aligned (also B and ldb for mode 0).  Arrays are HOST arrays of `count` entries; bias[q] may be NULL.              */
int slu_gemm_small_batched(const float* const* A, const int64_t* lda, const float* const* B, const int64_t* ldb,
                           const int* mode, float* const* C, const int64_t* ldc, const float* const* bias,
                           const int* accumulate, const int64_t* M, const int64_t* N, const int64_t* K, int64_t count,
                           void* stream);
int slu_colsum_f32(const float* X, int64_t x_rs, float* out, int64_t M, int64_t N,
                   int accumulate, void* stream);

/* -------- split-precision (16-bit MFMA) path of the FROZEN encoder stages --------------------------------
 * `nsplit` selects the scheme (csrc/slu_bf16.h):
 *   3  "bf16x3": x = x1 + x2 + x3 (three bf16 terms, exact); a product keeps the six bf16 x bf16 terms above
 *      2^-24 |a b| (v_mfma_f32_16x16x32_bf16, fp32 accumulation): fp32-class at 6/16 of the fp32-MFMA cycles, no
 *      range limit;
 *   2  "f16x2": x = hi + 2^-11 lo (two fp16 terms, 22-bit significand); hi hi on one fp32 accumulator, hi lo + lo hi
 *      on a second one, result acc0 + 2^-11 acc1 (v_mfma_f32_16x16x32_f16): fp32-class (<= 3 * 2^-22 per product
 *      worst case, measured 1e-7 of sum |a b| against float64: an fp32 fmaf chain's deviation) at 3/16 of the fp32-MFMA cycles; |x| < 65504, values
 *      below 6.1e-5 carry 11 bits (absolute error <= 1.5e-8);
 *   1  plain bf16 (BASELINE configs[4]).
 * Activations between the stages are `nsplit` planes of 16-bit terms (rows x ld, ld = round_up(K, 32), zero padded),
 * plane p at planes + p * plane_stride (in 16-bit elements).
 *   slu_split_bf16     fp32 (rows x K, row stride ldx) -> planes (entry into the format)
 *   slu_gemm_bf16_pack W (N x K) fp32 -> packed planes in MFMA B-fragment order (once per weight)
 *   slu_gemm_bf16      C (M x N fp32, row stride ldc) = A W^T + bias: the input projection x W_ih^T + b_ih of
 *                      nn.GRU (models.py:232/:262) for frozen layers; N must be a multiple of 64           */
int slu_split_bf16(const float* x, int64_t ldx, void* planes, int64_t plane_stride, int64_t rows, int64_t K,
                   int nsplit, void* stream);
size_t slu_gemm_bf16_pack_bytes(int64_t N, int64_t K, int nsplit);
int slu_gemm_bf16_pack(const float* W, int64_t ldw, int64_t w_cs, void* packed, int64_t N, int64_t K, int nsplit,
                       void* stream);      /* W[n][k] at W + n * ldw + k * w_cs: a weight or its transpose, in place */
int slu_gemm_bf16(const void* A_planes, int64_t a_plane_stride, int64_t lda, const void* w_packed,
                  const float* bias, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int nsplit,
                  void* stream);
/* The same product with A given as plain fp32 (M x K, k fast, row stride lda) and split on the fly inside the kernel —
 * the GEMMs of TRAINABLE layers, whose operands are produced by exact-fp32 kernels a moment earlier: the forward
 * projection x W_ih^T + b_ih and the data gradient d_gx W_ih (pack W_ih^T in place with w_cs / ldw swapped).  N, K, lda,
 * ldc multiples of 4, A and C 16-byte aligned; N need not be a multiple of 64.                                          */
int slu_gemm_bf16_a32(const float* A, int64_t lda, const void* w_packed, const float* bias, float* C, int64_t ldc,
                      int64_t M, int64_t N, int64_t K, int nsplit, void* stream);

/* C (M x N, row stride ldc) = A^T B on split-precision operands (fp32 accumulation): A (K x M, row stride lda), B (K x N,
 * ldb) fp32 with k as the slow index of both — the weight gradients d_gx^T x / d_gh^T h_prev of a GRU layer (the
 * reference's autograd GEMMs of nn.GRU, models.py:232/:262/:686): nsplit = 2 (f16x2, fp32-class: the default of trainable
 * layers, SLU_TRAIN_MATH), 1 (bf16: SLU_DTYPE=bf16, BASELINE configs[4]) or 3.  The operands are split / rounded while they
 * are staged (transposed) in LDS.  M and N multiples of 4, lda / ldb multiples of 4, 16-byte aligned
 * bases; deterministic split-K through the caller's workspace (slu_gemm_tn_bf16_workspace_bytes, 0 = none needed).   */
size_t slu_gemm_tn_bf16_workspace_bytes(int64_t M, int64_t N, int64_t K);
int slu_gemm_tn_bf16(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                     int64_t N, int64_t K, int nsplit, void* workspace, size_t workspace_bytes, void* stream);

/* slu_wconv_fwd for a FROZEN CNN block on the split-precision MFMA path (no `route`: forward only).  Needs
 * stride_t * c_in % 8 == 0 for c_in == 1, stride_t == 1 otherwise (channels are padded to a multiple of 8).
 * out_planes != NULL (pool == 1): the result goes straight into the split-precision activation format instead of
 * `out` — nsplit 16-bit planes (plane stride out_plane_stride elements) of (l_out * B) x round_up(c_out, 32), rows in
 * time-major order l * B + b, zero padded columns — which slu_gemm_bf16 reads (no fp32 round trip, no slu_split_bf16).
 * in_table != NULL: a DEVICE array of ceil(B / table_rows) base pointers; batch row b is read from
 * in_table[b / table_rows] + (b % table_rows) * l_in * c_in instead of in + b * l_in * c_in — a look-ahead super-batch
 * reads its batches where they lie instead of a concatenated copy (`in` is then ignored).
 * route (NULL, or as slu_wconv_fwd's, with the fp32 `out`): the block is TRAINABLE and runs its forward on bf16
 * operands (nsplit = 1, BASELINE configs[4]); the backward is slu_wconv_bwd_act / _bwd_weight / _bwd_data as usual.
 * packed_valid != 0: `workspace` still holds the filter pack a previous call built from these very weights (a frozen
 * block: the caller keeps the workspace per weight version) — the pack launch is skipped.
 * absmax_word (ABI 5; NULL or a device uint32, used by nsplit = 2 only): the launch raises it (atomic maximum) to the IEEE
 * bit pattern of the largest |v| among the values it splits into fp16 pairs — its input window and, with out_planes, its
 * output.  f16x2 operands must stay below 65504; the caller zeroes the word, reads it back after the launch and re-runs
 * the stage with nsplit = 3 (bf16x3: fp32's range) when the pattern is >= 0x477FE000 (65504.0f; NaN / inf rank higher).
 * in_pcm16 != 0 (ABI 5; c_in == 1, no route): `in` / the table's pointers address int16 samples and the block computes on
 * sample * in_scale (1 / 32768: what the reference's loaders hand the model, data.py:273-293 — exact in fp32, so the result
 * equals the call on the converted fp32 waveform bit for bit): PCM16 audio crosses PCIe at half the bytes.               */
size_t slu_wconv_bf16_workspace_bytes(int64_t c_out, int64_t c_in, int64_t k_t, int nsplit);
int slu_wconv_fwd_bf16(const float* in, const float* const* in_table, int64_t table_rows, const float* weight,
                       const float* bias, float* out, uint8_t* route, int64_t B,
                       int64_t l_in, int64_t c_in, int64_t c_out, int64_t k_t, int64_t stride_t, int do_abs,
                       int pool, float slope, int64_t out_sb, int64_t out_sl, void* out_planes,
                       int64_t out_plane_stride, void* workspace, size_t workspace_bytes, int packed_valid, int nsplit,
                       uint32_t* absmax_word, int in_pcm16, float in_scale, void* stream);
/* Persistent GRU recurrence on the split-precision MFMA path; arguments as slu_gru_seq_fwd, 16-sequence tiles,
 * W_hh (3 gates x nsplit 16-bit planes) resident in VGPRs, H = 64 / 128.  `reserve` (NULL for frozen layers) takes
 * the saved gates in the 16-sequence layout of slu_gru_reserve_bytes, so that slu_gru_seq_bwd (exact fp32 BPTT)
 * back-propagates through a bf16 forward (BASELINE configs[4]: bf16 forward contractions, fp32 gradients).
 * Fused input projection (x_planes != NULL, then gx = NULL): for a layer with K <= 64 input channels (the first GRU
 * layer: K = 60) the kernel computes x_t W_ih^T + b_ih itself — x as nsplit planes of (T*B) x round_up(K, 32) (the
 * previous stage's plane output), W_ih (D*3H x K, both directions stacked) packed by slu_gemm_bf16_pack, b_ih (D*3H) —
 * with the GEMM's accumulation order, i.e. the result equals slu_gemm_bf16 + this call bit for bit, without the
 * projection launch and its fp32 gx round trip.  nsplit = 2, H = 128, reserve = NULL only.                          */
int slu_gru_seq_fwd_bf16(const float* gx, const float* w_hh_fwd, const float* w_hh_rev, const float* b_hh_fwd,
                         const float* b_hh_rev, float* out, float* reserve, const void* x_planes,
                         int64_t x_plane_stride, int64_t K, const void* w_ih_packed, const float* b_ih, int64_t T,
                         int64_t B, int64_t H, int64_t D, int nsplit, int seq_tiles, void* stream);
/* The same recurrence of a FROZEN layer with the layer's Dropout(p) + Downsample("avg", 2) (models.py:246-251 / :276-281,
 * :26-46) applied in its epilogue (ABI 5): the lane that owns four consecutive hidden units of a sequence keeps the masked
 * h of a pooling window's first frame in registers and writes the average when the second frame arrives — the fp32
 * (T, B, D*H) output, its re-read and the slu_dropout_pool_fwd launch disappear.  Exactly one output:
 *   out_pooled  (ceil(T/2), B, D*H) fp32, or
 *   out_planes  nsplit 16-bit planes of (ceil(T/2)*B) x (D*H) (plane stride out_plane_stride elements): what
 *               slu_dropout_pool_fwd_planes writes, read by the next frozen layer's slu_gemm_bf16.
 * keep_bits (NULL iff p_drop == 0): the mask from slu_dropout_bits, one bit per element.  The result equals
 * slu_gru_seq_fwd_bf16 + slu_dropout_pool_fwd[_planes](method 1, factor 2) bit for bit.  D*H % 32 == 0.                  */
int slu_gru_seq_fwd_pool_bf16(const float* gx, const float* w_hh_fwd, const float* w_hh_rev, const float* b_hh_fwd,
                              const float* b_hh_rev, float* out_pooled, void* out_planes, int64_t out_plane_stride,
                              const uint32_t* keep_bits, float p_drop, const void* x_planes, int64_t x_plane_stride,
                              int64_t K, const void* w_ih_packed, const float* b_ih, int64_t T, int64_t B, int64_t H,
                              int64_t D, int nsplit, int seq_tiles, void* stream);

/* -------- GRU recurrence: torch.nn.GRU (models.py:232, :262, :686), h0 = 0, gates [r; z; n] ----
 *   gx      (T, B, D*3H): x_t @ W_ih^T + b_ih for direction d in columns [d*3H, (d+1)*3H)
 *   w_hh[d] (3H, H), b_hh[d] (3H): weight_hh_l0 / bias_hh_l0 (d = 0) and *_reverse (d = 1)
 *   out     (T, B, D*H): hidden state of direction d at time t in columns [d*H, (d+1)*H);
 *           direction 1 scans t = T-1 .. 0 (bidirectional GRU output, RNNSelect models.py:138-149)
 *   reserve NULL (inference / frozen layer) or slu_gru_reserve_bytes(): per step r, z, n,
 *           W_hn h + b_hn and h_{t-1}, in the lane order of the backward kernel
 * H = 16 / 32 / 64 / 128: one persistent workgroup per (direction, sequence tile) runs the whole time
 * loop with its slice of W_hh resident in VGPRs; tiles hold 16 sequences (v_mfma_f32_16x16x4_f32), or
 * 4 sequences (v_mfma_f32_4x4x1_16b_f32, H = 64 / 128) while the 16-sequence grid would leave CUs idle.
 * Any other H in [1, 8192] (W_hh no longer fits a CU's registers): one launch per time step enqueued by
 * this call, (16 units) x (16 sequences) x D workgroups each, W_hh from L2 (the reserve then has a
 * different, private layout: always pair slu_gru_seq_fwd / _bwd of the same H).  D is 1 or 2.        */
size_t slu_gru_reserve_bytes(int64_t T, int64_t B, int64_t H, int64_t D);
int slu_gru_seq_fwd(const float* gx, const float* w_hh_fwd, const float* w_hh_rev,
                    const float* b_hh_fwd, const float* b_hh_rev, float* out, float* reserve,
                    int64_t T, int64_t B, int64_t H, int64_t D, void* stream);
/* Input projection + recurrence of a TRAINABLE layer in ONE launch (exact fp32; reference models.py:232 / :262: nn.GRU's
 * x W_ih^T + b_ih followed by its T-step loop): workgroups of the same grid compute gx = x W_ih^T + b_ih tile by tile in
 * time order and run slu_gru_seq_fwd's 4-sequence recurrence, which picks up each step's rows as soon as their tiles are
 * published (write-through stores + a per-row-tile counter; see csrc/slu_gru_proj.hip).  Results are bit-identical to
 * slu_gemm_f32 followed by slu_gru_seq_fwd.  x: (T*B, I) rows of stride x_rs; w_ih: direction-stacked (D*3H, I), row
 * stride w_rs; b_ih (D*3H) or NULL; gx_scratch: T*B*D*3H floats (contents undefined afterwards); out / reserve as
 * slu_gru_seq_fwd.  state: slu_gru_proj_state_words(T, B) 32-bit words, ZERO before the first launch, then owned by the
 * launches of ONE (shape, stream) pair (counters accumulate across launches).  slu_gru_proj_supported: 1 if the shape is
 * taken (H = 128, B % 4 == 0, the 4-sequence geometry), else use the two calls.                                     */
int slu_gru_proj_supported(int64_t T, int64_t B, int64_t I, int64_t H, int64_t D);
int64_t slu_gru_proj_state_words(int64_t T, int64_t B);
int slu_gru_proj_seq_fwd(const float* x, int64_t x_rs, const float* w_ih, int64_t w_rs, const float* b_ih,
                         float* gx_scratch, const float* w_hh_fwd, const float* w_hh_rev,
                         const float* b_hh_fwd, const float* b_hh_rev, float* out, float* reserve,
                         int64_t T, int64_t B, int64_t I, int64_t H, int64_t D,
                         int32_t* state, int64_t state_words, void* stream);
/* Back-propagation through time.
 *   d_out   (T, B, D*H)  gradient w.r.t. `out`
 *   d_gx    (T, B, D*3H) gradient w.r.t. gx = x W_ih^T + b_ih          [dr_pre, dz_pre, dn_pre]
 *   d_gh    (T, B, D*3H) gradient w.r.t. h_{t-1} W_hh^T + b_hh          [dr_pre, dz_pre, dq],
 *           dq = dn_pre * r; the caller forms d(W_ih) = d_gx^T x, d(x) = d_gx W_ih and
 *           d(W_hh) = d_gh^T h_{t-1} with slu_gemm_f32 (h_{t-1} = `out` shifted by one step).
 *   d_bias_part NULL or (slu_gru_bias_tiles(T,B,H,D), D, 6H): partial sums (per sequence tile, or per
 *           block of (t, b) rows on the step-wise path) of [d_gx (3H) | d_gh (3H)]; summing over the first axis gives
 *           d(b_ih) = [0:3H) and d(b_hh) = [3H:6H).  (No atomics: deterministic.)                  */
int64_t slu_gru_bias_tiles(int64_t T, int64_t B, int64_t H, int64_t D);
int slu_gru_seq_bwd(const float* d_out, const float* reserve, const float* w_hh_fwd,
                    const float* w_hh_rev, float* d_gx, float* d_gh, float* d_bias_part,
                    int64_t T, int64_t B, int64_t H, int64_t D, void* stream);

/* -------- Dropout + Downsample: nn.Dropout (models.py:246,276,700) + Downsample (:26-46) -------
 *   x (T,B,C) time-major -> y (T_out,B,C); method 0 "none" (x[::factor]), 1 "avg", 2 "max" with
 *   ceil_mode (partial last window uses the frames that exist); T_out = ceil(T / factor).
 *   Dropout keep-mask, applied BEFORE pooling, scaled by 1/(1-p):
 *     mask != NULL : float {0,1} values, element (t,b,c) at mask[t*m_st + b*m_sb + c]
 *     mask == NULL and p > 0 : Philox4x32-10 keyed by (seed, offset), one draw per element index;
 *                              offset_dev (NULL or a device uint64) is added to `offset` at run
 *                              time, so a captured hipGraph can be replayed with fresh masks;
 *                              sub_batch > 0 treats the B sequences as B/sub_batch independent
 *                              batches of sub_batch: element indices are local to a sub-batch and
 *                              sub-batch k draws from offset + k*sub_stride (several training
 *                              steps' frozen stages evaluated in one launch, each with the masks
 *                              its own step would have drawn)
 *     keep_bits != NULL (ABI 5, forward only; mask must be NULL, C % 32 == 0): the 1-bit mask of slu_dropout_bits —
                              word (t*B + b) * C/32 + c/32, bit c % 32 = element (t,b,c) is kept: the mask of a FROZEN layer
     p == 0 : no dropout (eval mode)                                                            */
int slu_dropout_pool_fwd(const float* x, const float* mask, int64_t m_st, int64_t m_sb, const uint32_t* keep_bits, float p,
                         uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                         int64_t sub_batch, uint64_t sub_stride, int method, int64_t factor, float* y,
                         int64_t T, int64_t B, int64_t C, void* stream);
/* The same, writing the result straight into the split-precision activation format (nsplit 16-bit planes of
 * (T_out*B) x C, see slu_split_bf16) read by the next frozen layer's slu_gemm_bf16.  C % 32 == 0.             */
int slu_dropout_pool_fwd_planes(const float* x, const float* mask, int64_t m_st, int64_t m_sb, const uint32_t* keep_bits,
                                float p, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int64_t sub_batch,
                                uint64_t sub_stride, int method, int64_t factor, void* planes,
                                int64_t plane_stride, int nsplit, int64_t T, int64_t B, int64_t C, void* stream);
/* The dropout mask of a FROZEN layer as a bit stream (ABI 5): bits[(t*B + b) * C/32 + c/32] bit c%32 = element (t,b,c)
 * is kept — consumed by slu_gru_seq_fwd_pool_bf16 and by slu_dropout_pool_fwd[_planes](keep_bits), so that a frozen layer
 * has one mask whichever kernel applies it.  Philox4x32-10 keyed by (seed, offset [+ *offset_dev]); sub_batch / sub_stride
 * as above (row = t*B + b, or t*sub_batch + b % sub_batch on the stream of sub-batch b / sub_batch).  Random bits are used
 * economically: for p == 0.5 and C % 128 == 0 (every layer of the reference cfgs) the four words of block
 * (row * C/128 + c/128) ARE the 128 keep bits of channels [128 (c/128), +128); otherwise element e = row*C + c is kept iff
 * the 16-bit draw e%2 of word (e/2)%4 of block e/8 is below round((1-p) 2^16).  (The trainable layers' kernels above draw one
 * 24-bit uniform per element; the two streams are unrelated.)  C % 32 == 0, T <= 65535, bits 16-byte aligned.           */
int slu_dropout_bits(uint32_t* bits, float p, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                     int64_t sub_batch, uint64_t sub_stride, int64_t T, int64_t B, int64_t C, void* stream);
/* dx (T,B,C) from dy (T_out,B,C); x and y (forward input/output) are needed for method 2 only. */
int slu_dropout_pool_bwd(const float* dy, const float* x, const float* y, const float* mask,
                         int64_t m_st, int64_t m_sb, float p, uint64_t seed, uint64_t offset,
                         const uint64_t* offset_dev, int64_t sub_batch, uint64_t sub_stride,
                         int method, int64_t factor, float* dx, int64_t T, int64_t B, int64_t C,
                         void* stream);

/* [Abs ->] MaxPool1d(pool, ceil_mode=True) -> LeakyReLU(slope) / ReLU (slope 0) for pool widths the convolution's
 * epilogue does not fuse (any cnn_max_pool_len > 2: models.py:163-168, :205, :211-213).  x channels-last (B, L, C);
 * y[b, lo, c] at b * out_sb + lo * out_sl + c (channels-last or time-major); route (B, ceil(L / pool), C) bytes =
 * arg-max offset inside the window | (input was negative, abs) << 7, NULL when no backward follows; pool <= 127.
 * Backward: dx (B, L, C) fully written (zeros off the arg-max).                                                    */
int slu_pool_act_fwd(const float* x, float* y, uint8_t* route, int64_t B, int64_t L, int64_t C, int64_t pool,
                     int do_abs, float slope, int64_t out_sb, int64_t out_sl, void* stream);
int slu_pool_act_bwd(const float* dy, const float* y, const uint8_t* route, float* dx, int64_t B, int64_t L,
                     int64_t C, int64_t pool, float slope, int64_t out_sb, int64_t out_sl, void* stream);

/* -------- intent head: Linear "final_classifier" (models.py:709) -> FinalPool max over time
 * (:112-123) -> per-slot cross-entropy, accuracy and arg-max (:811-821, :839-844) ------------------
 *   h (T,B,C) time-major intent-GRU output; weight (V,C), bias (V), V = sum(values_per_slot);
 *   y (B,S) int64 labels or NULL (inference: logits / pred only);
 *   values_per_slot: HOST array of S entries (S <= 8) — the only host pointer of this ABI;
 *   logits (B,V) = max over t of h_t W^T + b;  argmax_t (B,V) int32;  pred (B,S) int64;
 *   d_logits (B,V) or NULL: d loss / d logits (softmax - onehot)/B per slot;
 *   row_stats (B,2) scratch;  loss_acc (2): loss = sum_slots mean_b CE, acc = mean_b [all slots right];
 *   epoch_sums (2 doubles) or NULL: += B * (loss, acc) — the running epoch statistics the reference accumulates on
 *   the host after every step (training.py:100-104: loss.item() * batch_size), kept on the device instead;
 *   ticket: NULL, or a zero-initialised device uint32 of the caller's (zero again afterwards; one per stream
 *   that may run this call concurrently): the launch's last workgroup then reduces row_stats to loss_acc itself
 *   instead of a second one-workgroup launch (same summation, same bits).
 *   drop_p > 0: the nn.Dropout between the last intent GRU and the classifier (models.py:700) is applied HERE — h is
 *   the GRU's raw output, element (t,b,c) is multiplied by the keep factor of element (t*B + b)*C + c of the Philox
 *   stream (drop_seed, drop_offset + *drop_offset_dev), i.e. the mask slu_dropout_pool_fwd draws for the same
 *   arguments, and the dropped activations are written to h_drop (T,B,C) for the backward pass (no separate dropout
 *   launches in the step).  Needs C % 4 == 0 and 16-byte aligned h / weight / h_drop.
 * Backward: d_h (T,B,C), d_weight (V,C), d_bias (V), all scaled by the device scalar *grad_scale;
 * d_h or the (d_weight, d_bias) pair may be NULL.  h = the activations the classifier saw (h_drop of a fused
 * forward); with drop_p > 0 (the forward's arguments) d_h is the gradient w.r.t. the GRU's RAW output.               */
int slu_cls_maxpool_ce_fwd(const float* h, const float* weight, const float* bias, const int64_t* y,
                           const int64_t* values_per_slot, int64_t num_slots, float* logits,
                           int32_t* argmax_t, int64_t* pred, float* d_logits, float* row_stats,
                           float* loss_acc, double* epoch_sums, uint32_t* ticket, float drop_p, uint64_t drop_seed,
                           uint64_t drop_offset, const uint64_t* drop_offset_dev, float* h_drop, int64_t T, int64_t B,
                           int64_t C, void* stream);
int slu_cls_maxpool_ce_bwd(const float* d_logits, const int32_t* argmax_t, const float* h,
                           const float* weight, const float* grad_scale, float* d_h,
                           float* d_weight, float* d_bias, float drop_p, uint64_t drop_seed, uint64_t drop_offset,
                           const uint64_t* drop_offset_dev, int64_t T, int64_t B, int64_t C, int64_t V, void* stream);

/* -------- ASR pre-training heads: F.cross_entropy(logits, y, ignore_index) + frame accuracy ----------
 * (PretrainedModel.forward, models.py:291-331; the Linear layers are slu_gemm_f32 calls).
 *   logits (N, V) row-major; y (N) int64, rows with y == ignore_index carry no label
 *   out3 = [mean CE over labelled rows, accuracy over labelled rows (first arg-max == y), #labelled]
 *   write_grad != 0: logits is overwritten IN PLACE with d(out3[0]) / d(logits)
 *   row_stats: workspace of 2 N floats.  Targets must lie in [0, V) or equal ignore_index.            */
int slu_frame_ce_fwd(float* logits, const int64_t* y, int64_t N, int64_t V, int64_t ignore_index,
                     int write_grad, float* row_stats, float* out3, void* stream);

/* -------- seq2seq intent decoder (models.py:418-557: Attention, DecoderRNN, Seq2SeqDecoder.forward) ------------------
 * The decoder's Linear layers and the GRUCell projections are slu_gemm_f32 calls; these are the step's other pieces,
 * forward and backward.  All (B, n) operands are row-major with the stated row strides (in elements).
 *   slu_gru_cell_fwd   torch.nn.GRUCell gate math: gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh (B, 3H, gates [r; z; n])
 *                      -> h_out = (1 - z) n + z h_prev; save (NULL or (4, B, H)) = r, z, n, gh_n for the backward;
 *                      drop_out (NULL or (B, H)) = nn.Dropout(p) of h_out (models.py:455: the next layer's input):
 *                      keep-mask from `mask` ((B, H) {0,1} floats) or Philox(seed, offset [+ *offset_dev]) at element
 *                      index idx_base + b * H + j (one index space for all steps and layers of a forward)
 *   slu_gru_cell_bwd   d_h (B, H; NULL = 0) + d_drop (NULL or (B, H), gradient of drop_out) -> d_gi, d_gh (B, 3H) and
 *                      d_h_prev = dh * z (the recurrent part through W_hh is the caller's GEMM); d_h_prev may alias d_h
 *   slu_attention_fwd  scores_t = <keys[b,t], query[b]> * inv_scale, weights = softmax_t, ctx[b] = sum_t weights_t
 *                      values[b,t]; keys[b,t,:] at keys + t * k_st + b * k_sb (time- or batch-major), values likewise
 *   slu_attention_bwd  d_keys / d_values are ACCUMULATED into (+=, same addressing: the encoder states are shared by
 *                      all decoding steps), d_query (B, Kd) is overwritten
 *   slu_logsoftmax_dot_fwd  logp_acc[b] += sum_v log_softmax(logits[b])_v * y[b, v]; lse[b] (NULL or B) = logsumexp
 *   slu_logsoftmax_dot_bwd  d_logits[b, v] = g[b * g_stride] * (y[b, v] - softmax_v * sum_v' y[b, v'])
 *   slu_neg_mean_f32   out[0] = -mean(x[0..n))  (loss = -log_probs.mean(), models.py:825)
 *   slu_fill_scaled_f32 dst[0..n) = g[0] * scale;  slu_broadcast_rows_f32 dst[b, 0..n) = src[0..n) for b < rows        */
int slu_gru_cell_fwd(const float* gi, const float* gh, const float* h_prev, int64_t ld_prev, float* h_out,
                     int64_t ld_out, float* save, float* drop_out, const float* mask, float p, uint64_t seed,
                     uint64_t offset, const uint64_t* offset_dev, uint64_t idx_base, int64_t B, int64_t H, void* stream);
int slu_gru_cell_bwd(const float* d_h, int64_t ld_dh, const float* d_drop, const float* save, const float* h_prev,
                     int64_t ld_prev, float* d_gi, float* d_gh, float* d_h_prev, int64_t ld_dprev, const float* mask,
                     float p, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, uint64_t idx_base, int64_t B,
                     int64_t H, void* stream);
int slu_attention_fwd(const float* keys, int64_t k_st, int64_t k_sb, const float* values, int64_t v_st, int64_t v_sb,
                      const float* query, int64_t ld_q, float* ctx, int64_t ld_ctx, float* weights, float inv_scale,
                      int64_t B, int64_t T, int64_t Kd, int64_t Vd, void* stream);
int slu_attention_bwd(const float* keys, int64_t k_st, int64_t k_sb, const float* values, int64_t v_st, int64_t v_sb,
                      const float* query, int64_t ld_q, const float* d_ctx, int64_t ld_dctx, const float* weights,
                      float* d_keys, float* d_values, float* d_query, int64_t ld_dq, float inv_scale, int64_t B,
                      int64_t T, int64_t Kd, int64_t Vd, void* stream);
int slu_logsoftmax_dot_fwd(const float* logits, const float* y, int64_t ld_y, float* logp_acc, float* lse, int64_t B,
                           int64_t V, void* stream);
int slu_logsoftmax_dot_bwd(const float* logits, const float* y, int64_t ld_y, const float* lse, const float* g,
                           int64_t g_stride, float* d_logits, int64_t B, int64_t V, void* stream);
int slu_neg_mean_f32(const float* x, float* out, int64_t n, void* stream);
int slu_fill_scaled_f32(float* dst, int64_t n, const float* g, float scale, void* stream);
int slu_broadcast_rows_f32(const float* src, float* dst, int64_t ld_dst, int64_t rows, int64_t n, void* stream);

/* -------- Adam: torch.optim.Adam(model.parameters(), lr) (training.py:19, default betas / eps) ------
 * One launch updates up to slu_adam_max_tensors() tensors of one dtype (elem_bytes 4 / 8); the pointer
 * arrays are HOST arrays of device pointers (they travel in the kernel arguments: hipGraph-safe).
 * *step_dev (int64, device) = number of updates these tensors have received so far.  With `ticket` (a
 * zero-initialised device uint32 of the caller's; zero again when the launch is done) the launch advances
 * *step_dev itself — its last workgroup writes it, after every workgroup has read it: pass the ticket with
 * the LAST tensor list that uses this counter in an optimisation step.  With ticket = NULL the counter is only
 * read, and slu_adam_advance_step adds 1 to the counters step_dev[i] whose bit i is set in cohort_mask, once per
 * optimisation step (only tensors that received a gradient advance, as in torch.optim.Adam).
 * grad_div: the gradients are divided by it before use (the world size under data parallelism, where
 * the RCCL all-reduce delivers the SUM over ranks; 1.0 otherwise).
 *   g /= grad_div;  m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g^2;
 *   p -= lr/(1 - b1^t) * m / (sqrt(v)/sqrt(1 - b2^t) + eps) */
int slu_adam_max_tensors(void);
int slu_adam_multi(void* const* params, const void* const* grads, void* const* exp_avg,
                   void* const* exp_avg_sq, const int64_t* numel, int64_t count, int elem_bytes,
                   int64_t* step_dev, double lr, double beta1, double beta2, double eps,
                   double grad_div, uint32_t* ticket, void* stream);
int slu_adam_advance_step(int64_t* step_dev, uint64_t cohort_mask, void* stream);

/* -------- data parallelism: the gradient all-reduce over the GPUs of a node (RCCL over xGMI) -----------------
 * The reference has no distributed code; these are thin wrappers over the RCCL instance the process already
 * holds (resolved at run time, not linked), so that the one collective of a training step can be enqueued on
 * the training stream between the captured backward and Adam graphs.  Protocol: rank 0 calls
 * slu_comm_unique_id and broadcasts the 128 bytes (any side channel: torch.distributed, a file, MPI); every
 * rank calls slu_comm_init with its rank (collective, like ncclCommInitRank); all-reduces are in place, SUM,
 * asynchronous on `stream`; the mean's 1/N lives in slu_adam_multi's grad_div.  slu_comm_version() = the RCCL
 * version code (0 when no RCCL library is mapped).                                                             */
int slu_comm_version(void);
int slu_comm_unique_id(void* id128);
int slu_comm_init(void** comm_out, const void* id128, int64_t nranks, int64_t rank);
int slu_comm_allreduce_f32(void* comm, float* buf, int64_t count, void* stream);
int slu_comm_allreduce_f64(void* comm, double* buf, int64_t count, void* stream);
int slu_comm_destroy(void* comm);
/* Both gradient buckets of a step as ONE grouped RCCL operation (ncclGroupStart / End around the fp32 and the float64
 * all-reduce): one launch per step whatever the trainable set.  Either count may be 0.  (ABI 8)                       */
int slu_comm_allreduce_group(void* comm, float* f32, int64_t n32, double* f64, int64_t n64, void* stream);

/* -------- hand-written all-reduce over peer-mapped device memory (xGMI within a node) — ABI 8 ------------------
 * the collective of a data-parallel step is latency-bound (1.2 - 5.5 MB per 0.15 - 2.7 ms
 * step), and xGMI is point to point: a two-shot all-reduce written as ONE kernel uses all links at once with two
 * flag hand-offs where a ring pays 2 (N - 1) hops (csrc/slu_comm_ipc.hip has the protocol).  No library collective
 * is involved: every rank creates a WINDOW (fine-grained device memory; slu_comm_ipc_window_bytes(payload) bytes),
 * sends its 64-byte hipIpcMemHandle to the peers over any side channel (torch.distributed / gloo here), maps theirs,
 * and calls slu_comm_allreduce_ipc with the N window pointers in rank order (windows[rank] = its own).  The fp32
 * bucket and the float64 bucket are typed segments of one payload: ONE launch per step, in place, SUM, added in rank
 * order by one rank per element (replicas receive bit-identical sums); asynchronous on `stream`, no host argument per
 * call (flags carry a device-resident epoch), so the launch replays as a node of the step's hipGraph.  All ranks must
 * call it the same number of times with the same sizes.  Waits are bounded (~1 min): slu_comm_ipc_status (synchronises
 * the device) returns 0, or 1 + q when a wait for rank q timed out.  window_create / open / close / destroy allocate,
 * map and release (never under capture).                                                                          */
int64_t slu_comm_ipc_window_bytes(int64_t payload_bytes);
int slu_comm_ipc_window_create(int64_t window_bytes, int64_t fine_grained, void** window_out, void* handle64);
int slu_comm_ipc_window_open(const void* handle64, void** window_out);
int slu_comm_ipc_window_close(void* peer_window);
int slu_comm_ipc_window_destroy(void* own_window);
int slu_comm_allreduce_ipc(void* const* windows, int64_t rank, int64_t nranks, int64_t window_bytes,
                           float* f32, int64_t n32, double* f64, int64_t n64, void* stream);
int slu_comm_ipc_status(void* own_window, int64_t* status_out);
/* Bound of every later wait of this rank's all-reduce launches, in polls (0 = the start-up default, 2^26 ~ minutes): set
 * after start-up, when waits are microseconds and a dead peer should be noticed in seconds (ABI 9).  Synchronises.      */
int slu_comm_ipc_set_spin_limit(void* own_window, int64_t polls);
/* Workgroups of the all-reduce kernel that can be resident together on `cus` compute units, and the number one call
 * launches (64): they spin on flags raised by the launch's own last workgroup, so launched <= resident is REQUIRED on the
 * partition the training stream is confined to (checked by the Python communicator at start-up; ABI 9).                 */
int slu_comm_ipc_resident_workgroups(int64_t cus, int64_t* resident_out, int64_t* launched_out);
int slu_comm_ipc_max_wait(void* own_window, int64_t* polls_out);   /* longest wait so far in ~1 us polls (diagnostics)   */
/* One load per 4 KiB page of every window, no flags, no waits: peer windows are mapped lazily, and a first touch inside
 * the all-reduce can stall a rank past its peers' bounded waits.  Call once after every window is open (then synchronise
 * and barrier) — slu_hip/dp.IpcComm does.                                                                             */
int slu_comm_ipc_window_touch(void* const* windows, int64_t rank, int64_t nranks, int64_t window_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SLU_HIP_H */
