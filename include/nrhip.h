/* libnrhip — C ABI of the MI355X (gfx950) NRMS / NAML encoder + scorer hot path.
 *
 * The reference (patngnw/NewsRecommendation) is pure Python on ATen ops and has no FFI of
 * its own; each entry point below replaces one ATen op sequence of the reference's
 * src/model/{NRMS,NAML,model_utils}.py and is what a `ctypes` binding inside those modules would call
 * (INTEGRATION.md shows the stub).  Citations are paths under /root/reference/.
 *
 * Conventions
 *  - All pointers are DEVICE pointers unless a comment says host.  Row-major, contiguous
 *    unless a leading dimension (`ld*`, in elements) is given.
 *  - `dtype` selects the storage type of activations / packed weights AND the MFMA operand
 *    type: NR_F32 (v_mfma_f32_16x16x4_f32, exact fp32) or NR_BF16 (v_mfma_f32_16x16x32_bf16,
 *    fp32 accumulate).  Parameters, gradients of parameters, pooled vectors, scores and the
 *    loss are always fp32.
 *  - The caller owns every buffer (outputs, saved activations, workspaces).  The library
 *    never allocates or frees device memory and never synchronises the stream; every call
 *    only enqueues kernels on `stream` (a hipStream_t).
 *  - Returns NR_OK (0) or an NR_ERR_* code; nr_last_error() gives the message.  Nothing
 *    throws across the ABI.
 *  - Leading dimensions of dtype buffers used as GEMM operands must be multiples of one
 *    16-byte chunk (4 fp32 / 8 bf16) and the buffers 16-byte aligned; nr_cast_pad produces
 *    such buffers (zero padded).
 *  - "accumulate" outputs (dw*, db*, dtable, dpad) are ADDED to: zero them first.
 *  - Dropout: Bernoulli keep decisions are a pure function of (seed, element index); the
 *    backward call must receive the same seed.  p == 0 disables it (eval mode).
 */
#ifndef NRHIP_H
#define NRHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NR_F32 0
#define NR_BF16 1

#define NR_OK 0
#define NR_ERR_ARG 1
#define NR_ERR_HIP 2

#define NR_SRC_DENSE 0  /* x is [M, ldx] of dtype                                   */
#define NR_SRC_GATHER 1 /* x is a table [V, ldx] of dtype, row m = table[ids[m*ids_stride]] */

typedef void* nr_stream_t; /* hipStream_t */

int nr_version(void);
/* Copies the calling thread's last error message (NUL terminated) into buf; returns its length. */
int nr_last_error(char* buf, size_t n);
/* Integer switches of the library (kernel-family selection, debugging): name without the NR_ prefix, e.g.
 * "NO_SLABS", "SIDE_STREAM", "TN3_ROUNDS" (full list: csrc/nr_common.h, enum NrOpt).  Defaults are the production
 * configuration; the environment variable NR_<NAME> presets an option once per process.  nr_get_option returns
 * -1 for an unknown name.  Not thread-synchronised with calls in flight: set options between calls.              */
/* sizeof of the descriptor structs as this library was compiled: out[0..3] = nr_mhsa_desc, nr_conv_desc, nr_pool_desc,
 * nr_linear_desc, and with n >= 6 also out[4..5] = nr_cast_job, nr_pack_job.  A binding compares them with its own layout at
 * load time (ABI drift -> refuse to run).                                                                              */
int nr_abi_sizes(size_t* out, int n);
/* Deterministic mode.  Outputs that several workgroups add into (dW, db, dtable, dpad) are accumulated with fp32 atomics
 * by default, so their last bits depend on the arrival order.  With a scratch buffer registered here they are accumulated
 * in 2^-36 fixed point with 64-bit integer atomics (order independent) and flushed onto the fp32 output at the end of
 * each call: gradients are bit-reproducible from run to run.  scratch: DEVICE memory, ZERO filled, 8 bytes per element of
 * the largest set of accumulated outputs of one call (e.g. 3N*(d_model+1) + table_rows*d_model for nr_mhsa_bwd); the
 * library leaves it zero filled after every call.  scratch == NULL switches the mode off.  One stream at a time.
 * Gather sources need `table_rows` in their descriptor in this mode.                                               */
int nr_set_deterministic(void* scratch, size_t bytes);
int nr_set_option(const char* name, int value);
int nr_get_option(const char* name);

/* Device: every entry point that takes a stream runs on the device that owns that stream (for the NULL stream: the
 * device that owns its first pointer argument) -- the library switches the calling thread to it for the duration
 * of the call (forward runs on the rank's main thread, backward on the autograd engine thread).                   */

/* ---------------------------------------------------------------------------------------
 * Packing: fp32 master parameter -> dtype operand with zero padded leading dimension.
 * transpose == 0: dst[r, c] = src[r, c]      dst is [rows, ld_dst]
 * transpose == 1: dst[c, r] = src[r, c]      dst is [cols, ld_dst]
 * Replaces the implicit `.float()` / `.cuda()` parameter materialisation of
 * src/model/NRMS.py:70-73 and src/main.py:79.                                              */
int nr_cast_pad(const float* src, int rows, int cols, int ld_src, void* dst, int ld_dst, int dtype,
                int transpose, nr_stream_t stream);
/* The same for up to NR_CAST_BATCH_MAX operands in ONE launch (the weights a training step re-packs after each optimizer
 * step are small: one 6-us launch each otherwise).  `jobs` is host memory; all destinations share `dtype`.            */
#define NR_CAST_BATCH_MAX 16
typedef struct {
  const float* src;
  void* dst;
  int rows, cols, ld_src, ld_dst, transpose;
} nr_cast_job;
int nr_cast_pad_batch(const nr_cast_job* jobs, int n, int dtype, nr_stream_t stream);
/* Conv1d weight [N, D, 3] (src/model/NAML.py:27-32) -> tap-major GEMM operand [N, 3*Dp]
 * (dst[n, tap*Dp + d] = w[n, d, tap]); unpack is the inverse on an fp32 gradient
 * (accumulate != 0: dw += ..., else dw = ...).                                               */
int nr_pack_conv_w(const float* w, int N, int D, void* dst, int Dp, int dtype, nr_stream_t stream);
int nr_unpack_conv_dw(const float* dw_pack, int N, int D, int Dp, float* dw, int accumulate, nr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K1  embedding row gather — nn.Embedding(padding_idx=0) lookup,
 * src/model/NRMS.py:28, src/model/NAML.py:48-50, and the eval-time news-vector gather
 * src/dataset.py:68,72.  out[m, 0:cols] = table[ids[m*ids_stride], 0:cols] (fp32 out).
 * bwd: dtable[ids[m], :] += dout[m, :] for ids[m] != 0 (padding_idx row gets no gradient). */
int nr_embed_gather_fwd(const void* table, int ld_table, int dtype, const int32_t* ids, int n_ids,
                        int ids_stride, int cols, float* out, int ld_out, nr_stream_t stream);
int nr_embed_gather_bwd(const float* dout, int ld_dout, const int32_t* ids, int n_ids, int ids_stride,
                        int cols, float* dtable, int ld_dtable, nr_stream_t stream);
/* Eval-time gather of news vectors straight into the compute dtype (src/dataset.py:68 `news_scoring[click_docs]` + the
 * cast the user encoder would do next): out[m, 0:cols] = (out_dtype) table[ids[m], 0:cols], table fp32.              */
int nr_gather_cast_fwd(const float* table, int ld_table, const int32_t* ids, int n_ids, int cols, void* out, int ld_out,
                       int out_dtype, nr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K3 (+K1,K2 fused)  multi-head self-attention — MultiHeadSelfAttention.forward +
 * ScaledDotProductAttention.forward, src/model/model_utils.py:39-55,78-95, with the
 * embedding lookup and the two F.dropout calls of src/model/NRMS.py:28-34 folded in.
 *   X      = dropout_in(rows of x)                         [n*L, d_model]
 *   Q|K|V  = X W_{Q|K|V}^T + b                             [n*L, 3N], N = heads*d_head
 *   per head: A = exp(QK^T/sqrt(d)) * mask_j / (sum_j + 1e-8);  ctx = A V
 *   y      = dropout_out(concat_heads ctx)                 [n*L, N]
 * No max-subtraction in the reference; the kernels use the algebraically identical stable
 * form exp(s-m)/(sum exp(s-m) + 1e-8 exp(-m)).                                             */
typedef struct {
  int n, L, d_model, heads, d_head, dtype;
  int src_kind;       /* NR_SRC_DENSE | NR_SRC_GATHER */
  const void* x;      /* dense: [n*L, ldx] dtype; gather: table [V, ldx] dtype */
  int ldx;
  const int32_t* ids; /* gather: [n*L] token ids (int32, as src/preprocess.py:52-54)        */
  float p_in;         /* dropout on X (src/model/NRMS.py:28-30); element index m*d_model + k */
  uint32_t seed_in;
  float p_out;        /* dropout on y (src/model/NRMS.py:32-34); element index m*N + c       */
  uint32_t seed_out;
  const float* mask;  /* [n, L] key-side 0/1 mask or NULL (src/model/model_utils.py:50-51)   */
  const void* w_qkv;  /* [3N, ldw] dtype: rows W_Q | W_K | W_V (nr_cast_pad of each)         */
  int ldw;
  const float* b_qkv; /* [3N] fp32 */
  void* x_rows;       /* optional [n*L, ld_rows] dtype (ld_rows >= d_model rounded up to a chunk): with a gather
                         source nr_mhsa_fwd stores the gathered + dropped-out rows X here and nr_mhsa_bwd reads
                         them back as a dense operand of the weight-gradient GEMM (no second gather / RNG pass);
                         NULL: backward regenerates X from (table, ids, seed_in).                            */
  int ld_rows;
  int32_t* row_ws;    /* optional scratch of nr_mhsa_workspace_bytes(d) bytes (bf16 gather source only).  Padding tokens (id 0) gather the
                         zero row of the table: nr_mhsa_fwd (when x_rows is given too) compacts the other rows on the
                         device, projects those alone and writes the bias into the rest (if table row 0 is not zero every
                         row is kept); nr_mhsa_bwd compacts again and runs the dX GEMM over the rows that reach the
                         table gradient; a pass over dy flags the sequences with a non-zero upstream gradient and the
                         weight-gradient GEMM contracts only the 32-row slabs that touch one (the rest of dQ|dK|dV is exactly
                         zero); all-padding sequences that no live slab comes near are left out of the backward attention altogether
                         (their dQ|dK|dV rows stay unwritten and are never read).  No host synchronisation.  NULL: every row
                         goes through the GEMMs.                    */
  size_t row_ws_bytes; /* size of row_ws in bytes: must be >= nr_mhsa_workspace_bytes(d) when row_ws != NULL (checked) */
  int table_rows;     /* gather source: rows V of the table (needed in deterministic mode only, to size dtable's shadow; else 0) */
  const void* proj_table; /* nr_mhsa_fwd only, optional [V, 3N] dtype = table . W_qkv^T + b_qkv (one nr_gemm_nt over the table):
                         eval mode (p_in == 0, bf16 gather source, qkv == NULL, no backward) gathers the projections of a token
                         from here instead of projecting every occurrence -- same values, ~V/(n*L) of the GEMM work */
  const int32_t* seq_needed; /* optional [n]: 0 = the caller will not use this sequence's output (it reaches the
                         loss through a factor 0, e.g. a masked history slot, src/model/NRMS.py:59-60, model_utils.py:28,51): its y rows
                         are written as exact zeros without being computed.  NULL: every sequence is computed.  With row_ws the
                         forward also leaves the x_rows of all-padding sequences farther than 32 / L + 2 sequences from every
                         needed one unwritten (nothing reads them): pass the SAME flags to nr_mhsa_bwd.                      */
  const int32_t* seq_nz; /* nr_mhsa_bwd only, optional [n]: 0 = the upstream gradient dy of this sequence is exactly zero (the flags
                         nr_additive_pool_bwd leaves in its workspace, see nr_pool_seq_flags); NULL: the library scans dy itself */
  int row_ws_ready;   /* nr_mhsa_bwd only: nonzero = row_ws still holds what nr_mhsa_fwd wrote for these ids (reused as is).
                         REQUIRED when nr_mhsa_fwd was given row_ws: on the bf16 title-level path the forward then leaves
                         the qkv rows of all-padding sequences unwritten (the attention kernels substitute the bias), and the
                         backward attention has to do the same.                                                   */
  int bwd_phase;      /* nr_mhsa_bwd only.  0: the whole backward.  1: everything but the weight / bias gradients (flags, attention
                         backward, dx / table gradient) ; 2: only dw_qkv / db_qkv, after a phase-1 call with the same
                         descriptor and row_ws.  The split lets a data-parallel host start the all-reduce of the (large)
                         table gradient while the weight-gradient GEMM still runs.  Not in deterministic mode.      */
  int y_far_unwritten; /* nr_mhsa_fwd only, with seq_needed: nonzero = the y rows of an unneeded sequence need to be ZEROS only when
                         a needed sequence lies within 32 / L + 2 sequences of it (a 32-row slab of a weight-gradient GEMM can then
                         reach them); farther ones may stay UNWRITTEN.  For callers whose only consumer of y is
                         nr_additive_pool_fwd / _bwd with the same flags on a shape for which nr_pool_contracts_slabs() says 1
                         (it never reads those rows); 0.3 GB of zero stores per step at the news level otherwise.          */
  int dy_far_unwritten; /* nr_mhsa_bwd only: nonzero = the dy rows of a sequence whose seq_nz flag is 0 are zeros only within 32 / L + 2
                         sequences of a flagged one, UNWRITTEN memory farther away (nr_pool_desc.dx_far_unwritten): the backward
                         then takes dy = 0 for every unflagged sequence without reading it.  Needs seq_nz and a descriptor for
                         which nr_mhsa_compact_rows() says 1 (refused otherwise).                                            */
} nr_mhsa_desc;

/* qkv: [n*L, 3N] dtype (saved for backward); y: [n*L, N] dtype.
 * Title-level bf16 gather sources (L <= 32, 3*d_head <= 64, 288 < d_model <= 320) run as ONE fused kernel that keeps
 * Q|K|V on chip; nr_mhsa_fwd_fused(d) returns 1 for such a descriptor, and then qkv may be NULL (inference: the
 * projections are never written to HBM).  On every other path qkv is required.                              */
int nr_mhsa_fwd_fused(const nr_mhsa_desc* d);
/* 1 when nr_mhsa_fwd / nr_mhsa_bwd keep x_rows and dqkv in compact row storage for this descriptor (bf16 gather source with
 * x_rows and row_ws, title-level shapes): only then may dy_far_unwritten be used.                                  */
int nr_mhsa_compact_rows(const nr_mhsa_desc* d);
/* Bytes of row_ws that nr_mhsa_fwd / nr_mhsa_bwd use for this descriptor (depends on n and L only). */
size_t nr_mhsa_workspace_bytes(const nr_mhsa_desc* d);
int nr_mhsa_fwd(const nr_mhsa_desc* d, void* qkv, void* y, nr_stream_t stream);
/* dy [n*L, N] dtype.  dqkv: workspace [n*L, 3N] dtype.  w_qkv_t: [Kp, ldwt] dtype = w_qkv^T
 * (nr_cast_pad transpose=1; Kp = d_model rounded up to a chunk; needed only if dx/dtable).
 * dw_qkv [3N, d_model], db_qkv [3N]: fp32, accumulated.  dx: dense source -> [n*L, ldx] dtype
 * or NULL; dtable: gather source -> [V, d_model] fp32 accumulated (skips id 0) or NULL.      */
int nr_mhsa_bwd(const nr_mhsa_desc* d, const void* qkv, const void* dy, void* dqkv, const void* w_qkv_t,
                int ldwt, float* dw_qkv, float* db_qkv, void* dx, float* dtable, nr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * The attention core alone -- ScaledDotProductAttention.forward, src/model/model_utils.py:39-55 -- on given
 * projections.  qkv: [n*L, 3N] dtype, token-major, columns Q | K | V with head h at [h*d_head, (h+1)*d_head) of each
 * (the layout nr_mhsa_fwd produces; the Python surface class packs the reference's [n, h, L, d] views into it).
 * mask: [n, L] key-side 0/1 or NULL.  y: [n*L, N] dtype, heads concatenated (src/model/model_utils.py:94).
 * p_out / seed_out: optional dropout on y (element index m*N + c), 0 = none.
 * bwd: dqkv [n*L, 3N] dtype = gradient of the three projections given dy [n*L, N].                          */
int nr_sdpa_fwd(const void* qkv, const float* mask, void* y, int n, int L, int heads, int d_head, int dtype, float p_out,
                uint32_t seed_out, nr_stream_t stream);
int nr_sdpa_bwd(const void* qkv, const float* mask, const void* dy, void* dqkv, int n, int L, int heads, int d_head,
                int dtype, float p_out, uint32_t seed_out, nr_stream_t stream);

/* Index validation (what torch's embedding / cross_entropy raise IndexError for; the kernels themselves trust their
 * indices).  Counts the entries of ids[i*stride], i < count, outside [0, rows) into *bad (DEVICE int32, accumulated:
 * zero it first).  The Python layer calls it when ops.CHECK_INDICES is on and raises IndexError.                */
int nr_check_ids(const int32_t* ids, int count, int stride, int rows, int32_t* bad, nr_stream_t stream);
int nr_check_labels(const int64_t* label, int count, int classes, int32_t* bad, nr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K4 (+K1,K2 fused)  Conv1d(D, N, kernel_size=3, padding=1) over title tokens,
 * src/model/NAML.py:27-32,47-54: row gather of a [T*D] title-embedding row per news id,
 * dropout, im2col-free GEMM against the tap-major packed weight.  y [n*T, N] dtype.         */
typedef struct {
  int n, T, D, Dp, N, dtype;
  const void* table;  /* [V, T*Dp] dtype (nr_cast_pad of the [V*T, D] view with ld_dst = Dp) */
  const int32_t* ids; /* news ids, id of row i = ids[i*ids_stride] (src/model/NAML.py:47)    */
  int ids_stride;
  float p_in;         /* dropout on the gathered [T, D] block (src/model/NAML.py:51-53); element index (i*T+t)*D + d */
  uint32_t seed_in;
  const void* w_pack; /* [N, 3*Dp] dtype (nr_pack_conv_w) */
  const float* bias;  /* [N] */
  void* x_rows;       /* optional scratch [n*(T+1)+1, Dp] dtype: the forward stores the token rows (gather + dropout) here
                         once -- token t of title i at row i*(T+1)+1+t, zero rows in between -- so that the im2col row of
                         a token is 3*Dp CONTIGUOUS elements starting one row above it, and runs the LDS-DMA GEMM on those
                         overlapping rows (no 3x im2col copy); a backward given the same buffer reuses them.
                         NULL: the operand is gathered on the fly inside the GEMMs.                                   */
  int ld_rows;        /* == Dp */
  int32_t* bwd_ws;    /* optional backward scratch of nr_conv_workspace_bytes(d) bytes (bf16, with x_rows): nr_conv1d_k3_bwd flags the
                         titles whose upstream gradient dy is not all zero and contracts only the 32-row slabs that touch
                         one (masked history slots have an exactly zero dy).  NULL: every row is contracted.          */
  size_t bwd_ws_bytes; /* size of bwd_ws in bytes, >= nr_conv_workspace_bytes(d) when bwd_ws != NULL (checked)              */
  const int32_t* seq_nz; /* nr_conv1d_k3_bwd only, optional [n]: 0 = dy of this title is exactly zero (see nr_pool_seq_flags)  */
  const int32_t* seq_needed; /* optional [n]: 0 = the caller will not use this title's output (see nr_mhsa_desc).  nr_conv1d_k3_fwd
                         does not compute row tiles made of such titles only -- their y rows stay UNWRITTEN, the consumer
                         (nr_additive_pool_fwd / _bwd with the same flags) never reads them -- and, on shapes whose backward
                         contracts live slabs only, does not store the x_rows of titles far from every needed one;
                         nr_conv1d_k3_bwd must then be given the same flags and bwd_ws                                         */
} nr_conv_desc;
size_t nr_conv_workspace_bytes(const nr_conv_desc* d);
int nr_conv1d_k3_fwd(const nr_conv_desc* d, void* y, nr_stream_t stream);
/* dw_pack [N, 3*Dp] fp32 accumulated (nr_unpack_conv_dw -> [N, D, 3]); db [N] accumulated.
 * The title-embedding table is frozen on this path (src/demo.sh:12): no dtable.             */
int nr_conv1d_k3_bwd(const nr_conv_desc* d, const void* dy, float* dw_pack, float* db, nr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K5  additive attention pooling — AttentionPooling.forward, src/model/model_utils.py:13-31
 *   e = tanh(x W1^T + b1); a = exp(e w2 + b2) * mask; a /= sum_L a + 1e-8; out = sum_L a x   */
typedef struct {
  int n, L, N, q, dtype;
  const void* x;     /* [n*L, N] dtype */
  const float* mask; /* [n, L] or NULL */
  const void* w1;    /* [q, ldw1] dtype */
  int ldw1;
  const float* b1;   /* [q] */
  const float* w2;   /* [q] (att_fc2.weight [1, q]) */
  const float* b2;   /* [1] */
  size_t partial_bytes; /* nr_additive_pool_bwd: size of `partial` in bytes, >= nr_pool_workspace_bytes(d) (checked) */
  const int32_t* seq_needed; /* optional [n]: 0 = this sequence's output is not used by the caller.  nr_additive_pool_fwd writes its
                         out row and alpha as zeros and may leave its e rows unwritten; nr_additive_pool_bwd must be given the same
                         flags: the pooled gradient of such a sequence is zero by contract, its dpre / dx rows are written as zeros
                         and its e rows are never read                                                                       */
  int dx_far_unwritten; /* nr_additive_pool_bwd: nonzero = the dx rows of a sequence with a zero pooled gradient need to be zeros only
                         within 32 / L + 2 sequences of one with a gradient; farther ones MAY stay unwritten (0.2 GB of zero stores
                         per step at the news level).  For a caller that hands dx, together with the flags of nr_pool_seq_flags,
                         to nr_mhsa_bwd with dy_far_unwritten set -- nothing else may read those rows.                       */
} nr_pool_desc;
/* 1 when nr_additive_pool_bwd contracts only the 32-row slabs that touch a sequence with a non-zero pooled gradient for this
 * descriptor (n, L, N, q, dtype are read): it then never reads x rows farther than 32 / L + 2 sequences from such a sequence
 * (see nr_mhsa_desc.y_far_unwritten).  0: its weight-gradient GEMM reads every row of x.                                  */
int nr_pool_contracts_slabs(const nr_pool_desc* d);
/* Bytes of the `partial` workspace of nr_additive_pool_bwd. */
size_t nr_pool_workspace_bytes(const nr_pool_desc* d);
/* After nr_additive_pool_bwd (bf16, L <= 32, rows a multiple of 32): the [n] int32 flags inside `partial` that say which
 * sequences had a non-zero pooled gradient g -- exactly the sequences whose dx rows are non-zero.  The producer of x can
 * take them as its `seq_nz` instead of scanning dx.  Returns NULL when this descriptor does not produce flags.        */
const int32_t* nr_pool_seq_flags(const nr_pool_desc* d, const float* partial);
/* e: [n*L, q] dtype (tanh output, saved); alpha: [n*L] fp32 (saved); out: fp32, row i at out + i*ld_out. */
int nr_additive_pool_fwd(const nr_pool_desc* d, void* e, float* alpha, float* out, int ld_out, nr_stream_t stream);
/* g: fp32 d(out), row i at g + i*ld_g.  w1_t [N, ldw1t] dtype = w1^T.  dpre: workspace [n*L, q] dtype.
 * partial: workspace of nr_pool_workspace_bytes(d) bytes, fp32 (one row per workgroup; small n uses one sequence per workgroup; at the news level the
 *          unused tail holds the int32 flags / live-slab list of the sequences whose pooled gradient g is not all zero --
 *          sequences with g == 0 get exact zeros in dpre / dx and are skipped by the att_fc1 weight gradient).
 * dw1 [q,N], db1 [q], dw2 [q], db2 [1]: accumulated.
 * dx: [n*L, N] dtype (overwritten) or NULL.                                                  */
int nr_additive_pool_bwd(const nr_pool_desc* d, const void* e, const float* alpha, const float* g, int ld_g,
                         const void* w1_t, int ldw1t, void* dpre, float* partial, float* dw1, float* db1,
                         float* dw2, float* db2, void* dx, nr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K6  pad-doc blend — src/model/NRMS.py:59-60, src/model/NAML.py:94-95:
 *   out = x*m + pad*(1-m), written as dtype (the GEMM operand of the user-level ops).
 * mask == NULL: plain fp32 -> dtype cast (user_log_mask=True path).
 * bwd: dx = dout*m (fp32); dpad[c] += sum (1-m) dout (accumulated).                          */
int nr_pad_blend_fwd(const float* x, const float* mask, const float* pad, void* out, int n, int L, int N,
                     int dtype, nr_stream_t stream);
int nr_pad_blend_bwd(const void* dout, const float* mask, float* dx, float* dpad, int n, int L, int N,
                     int dtype, nr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K7  gather + Linear — category / subcategory view, src/model/NAML.py:19-24,60-68:
 *   out[m, :] = rows(x)[m, :] W^T + b      (fp32 out, row m at out + m*ld_out)               */
typedef struct {
  int M, K, N, dtype;
  int src_kind;
  const void* x; /* dense [M, ldx] dtype or table [V, ldx] dtype */
  int ldx;
  const int32_t* ids;
  int ids_stride;
  const void* w; /* [N, ldw] dtype */
  int ldw;
  const float* bias; /* [N] or NULL */
  const void* w_t;   /* [K, ldwt] dtype = w^T; only read by nr_linear_bwd when dtable != NULL */
  int ldwt;
  size_t dout_ws_bytes; /* nr_linear_bwd: size of dout_ws in bytes, >= nr_linear_workspace_bytes(d) (checked) */
  int table_rows;    /* gather source: rows V of the table (deterministic mode only, see nr_set_deterministic; else 0) */
} nr_linear_desc;
size_t nr_linear_workspace_bytes(const nr_linear_desc* d);
int nr_linear_fwd(const nr_linear_desc* d, float* out, int ld_out, nr_stream_t stream);
/* dout fp32 (row stride ld_dout); dout_ws: workspace of nr_linear_workspace_bytes(d) bytes ([M, Nc] dtype, Nc = N rounded
 * up to a chunk).
 * dw [N, K], db [N] accumulated; dtable [V, K] fp32 accumulated (gather source, skips id 0) or NULL. */
int nr_linear_bwd(const nr_linear_desc* d, const float* dout, int ld_dout, void* dout_ws, float* dw,
                  float* db, float* dtable, nr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K8  candidate scorer + cross-entropy — torch.bmm + nn.CrossEntropyLoss,
 * src/model/NRMS.py:77,93-94, src/model/NAML.py:111,128-129.
 *   score[b, j] = <cand[b, j, :], user[b, :]>;  loss = mean_b -log softmax(score[b])[label[b]]
 * cand row (b, j) at cand + (b*C + j)*ld_cand.  lossvec: workspace fp32 [B].                 */
int nr_score_ce_fwd(const float* cand, int ld_cand, const float* user, const int64_t* label, float* score,
                    float* loss, float* lossvec, int B, int C, int N, nr_stream_t stream);
/* gloss: DEVICE pointer to the scalar d(loss) (no host sync); gscore: optional [B, C] d(score) added to
 * the cross-entropy term, or NULL.  dcand row (b, j) at dcand + (b*C + j)*ld_dcand; duser [B, N].        */
int nr_score_ce_bwd(const float* cand, int ld_cand, const float* user, const int64_t* label,
                    const float* score, const float* gloss, const float* gscore, float* dcand, int ld_dcand,
                    float* duser, int B, int C, int N, nr_stream_t stream);
/* Eval-time scorer (src/main.py:253): variable-length candidate lists.
 * score[i] = <news_vecs[cand_ids[i]], user[imp_of[i]]>, i in [0, n_cand).                     */
int nr_score_eval(const float* news_vecs, int ld_news, const int32_t* cand_ids, const int32_t* imp_of,
                  const float* user, int ld_user, float* score, int n_cand, int N, nr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * f1  device-side batch assembly -- DatasetTrain.line_mapper's `news_combined[...]` gathers + the positive splice +
 * the DataLoader collate, src/dataset.py:44-49, src/main.py:89-103.  Only news INDICES come from the host (once per
 * epoch); the feature rows are gathered here.
 *   news_combined [n_rows, F] int32 (src/preprocess.py:50-72: title token ids, or [news, category, subcategory] ids)
 *   hist_idx [B, H] news indices (front padded with 0, src/dataset.py:17-24); pos_idx [B]; neg_idx [B, K];
 *   label [B] int64 = position of the positive (random.randint(0, K), src/dataset.py:45)
 *   history [B, H, F] = news_combined[hist_idx]
 *   candidate [B, 1+K, F]: slot j holds neg[j] for j < label, pos for j == label, neg[j-1] beyond (src/dataset.py:46)
 * bad (optional, DEVICE int32, accumulated): number of out-of-range indices / labels met (row 0 / label 0 is used
 * instead; numpy / torch would raise IndexError).                                                               */
int nr_assemble_batch(const int32_t* news_combined, int n_rows, int F, const int32_t* hist_idx, const int32_t* pos_idx,
                      const int32_t* neg_idx, const int64_t* label, int B, int H, int K, int32_t* history, int32_t* candidate,
                      int32_t* bad, nr_stream_t stream);

/* One encoder batch, Model.forward's `torch.cat([candidate titles, history titles])` (the reference encodes the two in two
 * passes, src/model/NRMS.py:87,90 / src/model/NAML.py forward; one pass here) plus the "is this title's vector used at all"
 * flags of the rows: out [rows_a + rows_b, F] = [a ; b], flags [rows_a + rows_b] = [1 .. 1 ; mask_b != 0] (flags and mask_b
 * optional; a NULL mask_b flags every row).                                                                          */
int nr_stack_rows(const int32_t* a, int rows_a, const int32_t* b, int rows_b, int F, const float* mask_b, int32_t* out,
                  int32_t* flags, nr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * f2  on-device ranking metrics -- src/metrics.py:5-23 (dcg/ndcg/mrr), sklearn roc_auc_score (src/metrics.py:1),
 * accumulated as src/main.py:249-263 does.  Impression i owns score / label entries [offsets[i], offsets[i+1]).
 * Impressions whose labels are all 0 or all 1 are skipped (src/main.py:250).  Computed in fp64.
 *   per_imp: workspace of nr_eval_metrics_workspace_bytes(n_imp) bytes; afterwards row i = [AUC, MRR, nDCG@5, nDCG@10]
 *            of impression i (AUC = -1 marks a skipped impression)
 *   sums[5] (fp64, overwritten) = [scored impressions, sum AUC, sum MRR, sum nDCG@5, sum nDCG@10], reduced in a
 *            fixed order (bit-reproducible)
 * max_cand: largest candidate count of an impression (host knowledge of `offsets`); at most 4096.
 * Ties between scores: descending order with the LATER candidate first (np.argsort(kind="stable")[::-1]); numpy's
 * default sort leaves the order of tied scores unspecified.                                                    */
size_t nr_eval_metrics_workspace_bytes(int n_imp);
int nr_eval_metrics(const float* score, const int32_t* label, const int32_t* offsets, int n_imp, int max_cand, double* per_imp,
                    size_t per_imp_bytes, double* sums, nr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * f4  Adam (torch.optim.Adam defaults of src/main.py:76: no weight decay, no amsgrad) over ONE flat fp32 bucket:
 *   g' = grad * grad_scale (the 1/world of the gradient all-reduce-mean, src/main.py:82 semantics)
 *   m = lerp(m, g', 1-beta1); v = beta2 v + (1-beta2) g'^2;
 *   param -= lr / (1-beta1^step) * m / (sqrt(v) / sqrt(1-beta2^step) + eps)
 * zero_grad != 0: grad is cleared in the same pass (the next step's optimizer.zero_grad(), src/main.py:108).
 * step >= 1 is the 1-based update count.  Buffers 16-byte aligned.                                               */
int nr_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2, float eps,
                 int step, float grad_scale, int zero_grad, nr_stream_t stream);
/* The same update, plus the packed bf16 GEMM operands of parameter matrices that live in the bucket (what nr_cast_pad would
 * make of the new values -- bit-identical -- written in the same pass).  Job j: the matrix at elements [first, first + count)
 * of the bucket, `cols` columns per row (a multiple of 4, as is `first`), goes to dst [count / cols, ld_dst] bf16; padding
 * columns of dst are not touched (nr_cast_pad zeroed them when the operand was first made).  At most NR_ADAM_PACK_MAX jobs. */
#define NR_ADAM_PACK_MAX 4
typedef struct {
  size_t first, count;
  int cols, ld_dst;
  void* dst;
} nr_pack_job;
int nr_adam_step_packed(float* param, float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2, float eps,
                        int step, float grad_scale, int zero_grad, const nr_pack_job* jobs, int n_jobs, nr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Per-kernel timing (measurement only).  While enabled, every kernel launch inside the library is
 * bracketed by hipEventRecord on the launch stream.  nr_prof_collect waits for the recorded events,
 * writes one line per label "label<TAB>launches<TAB>total_ms\n" into buf (host), clears the log and
 * returns the number of bytes written (or a negative NR_ERR_* code).                          */
/* on = 1: every launch; on = 2: only launches that work on >= 65 536 rows (a pair of event records costs ~3 us of
 * stream time, which adds up over the ~60 small launches of a step); on = 3: only launches whose label starts with the
 * prefix given to nr_prof_filter (one kernel: the timed region of a benchmark pays for two event records per step);
 * on = 0: off.                                                                                  */
int nr_prof_enable(int on);
int nr_prof_filter(const char* label_prefix);
int nr_prof_collect(char* buf, size_t n);
/* Phase stamps of the tiled LDS-DMA NT GEMM (measurement only; option NT_ABLATE bit 64, tools/nt_trace.py): copies up to
 * n (<= 8 * 4096) 64-bit words -- per workgroup 7 s_memrealtime stamps and the HW_ID register -- after a device sync. */
int nr_debug_nt_trace(unsigned long long* out, int n);

/* ---------------------------------------------------------------------------------------
 * The two MFMA GEMM building blocks on dense operands (used by every op above; exported for unit tests
 * and kernel-level measurement).  A [M, lda], B [N, ldb] of dtype; C [M, ldc] of out_dtype:
 *   nr_gemm_nt:  C = A . B^T (+ bias[N]) (tanh if act_tanh)
 *   nr_gemm_tn:  dW[N, ldw] (fp32) += dC[M, ldc]^T . A[M, lda] ; db[N] += column sums of dC (db may be NULL) */
int nr_gemm_nt(int dtype, const void* A, int lda, const void* B, int ldb, const float* bias, int act_tanh, void* C,
               int ldc, int out_dtype, int M, int N, int K, nr_stream_t stream);
int nr_gemm_tn(int dtype, const void* dC, int ldc, const void* A, int lda, float* dW, int ldw, float* db, int M, int N,
               int K, nr_stream_t stream);

/* Test hook: materialise the dropout keep mask (1.0 / 0.0) for `count` element indices. */
int nr_dropout_mask(float* out, uint32_t count, float p, uint32_t seed, nr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NRHIP_H */
