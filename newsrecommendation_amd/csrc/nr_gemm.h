// Internal GEMM interface of libnrhip: MFMA tiles (16x16x32 bf16 / 16x16x4 f32) on gfx950.
//   NT :  C[M,N]  = epi( rows(A)[M,K] . B[N,K]^T )           activations x packed weights
//   TN :  dW[N,K] += dC[M,N]^T . rows(A)[M,K]  (+ db = colsum dC)   weight gradients, split over M
// rows(A) is a "row source": dense rows, rows gathered from a table by id (embedding
// lookup fused into the operand load), or the im2col view of a k=3 convolution over a
// gathered [T, D] block; optional counter-based dropout is applied while staging.
#pragma once
#include "nr_common.h"

enum { ROWS_DENSE = 0, ROWS_GATHER = 1, ROWS_IM2COL3 = 2 };

struct RowSrc {
  const void* base;    // dense: [M, ld]; gather: table [V, ld]; im2col3: table [V, Tlen*ld]
  const int32_t* ids;  // gather: id of row m = ids[m*ids_stride]; im2col3: id of block s = ids[s*ids_stride]
  int ld;              // elements between consecutive rows (im2col3: between tokens, = Dp)
  int ids_stride;
  int kind;
  int Tlen;            // im2col3: tokens per block
  int Dtrue;           // dropout: elements per logical row (index = row*Dtrue + col, applied for col < Dtrue)
  DropCfg drop;
  int gap;             // dense, > 0: logical row m starts at buffer row m + m / gap (one extra row after every `gap` rows) and may
                       // run on into the following buffer rows (K > ld): the k = 3 convolution's token rows with a zero row
                       // between titles (nr_launch_conv_rows)
};

enum {
  EPI_STORE = 0,    // C = acc (+ bias[n]) (tanh)
  EPI_POOLBWD = 1,  // C = acc + rowscale[m] * G[(m / L) * ldg + n]
  EPI_SCATTER = 2,  // dtable[ids[m]*ldc + n] += keep(m*Dtrue+n) ? acc*scale : 0   (ids[m] != 0, n < Dtrue)
  EPI_STORE_TANH = 3,  // C = tanh(acc + bias[n]) -- its own instantiation: tanhf costs ~50 VGPRs in the epilogue
};

struct EpiArgs {
  void* C;
  int ldc;
  int out_dtype;          // NR_F32 / NR_BF16 for EPI_STORE / EPI_POOLBWD
  const float* bias;      // [N] or null
  int act_tanh;
  const float* rowscale;  // POOLBWD: alpha [M]
  const float* G;         // POOLBWD: [n, ldg] fp32
  int ldg, L;
  const int32_t* ids;     // SCATTER
  int ids_stride;
  int Dtrue;
  DropCfg drop;           // SCATTER: the forward's input dropout
  const int32_t* row_count;  // SCATTER / STORE (bf16 DMA kernel), optional compaction (nr_launch_compact_rows): device count of live rows,
  const int32_t* row_idx;    //   their original row numbers (A row, dropout element index) and
  const int32_t* row_ids;    //   their token ids, both in compacted order
  const int32_t* seq_nz;     // POOLBWD, optional [M / L]: 0 = this sequence's rowscale-term operand g and its A rows are all zero;
                             // STORE / STORE_TANH (dense rows), optional [M / L]: 0 = nobody needs this sequence's output rows: row
                             // tiles made of such sequences only are skipped (their rows stay unwritten)
  void* rows_out;         // optional: the staged A rows (after gather / dropout), dtype, [M, ld_rows_out]
  int ld_rows_out;
  int a_gap;              // LDS-DMA kernel: RowSrc::gap of the A operand (set by the launcher)
  const int32_t* a_idx;   // with row_count: the A row of compacted row m is a_idx[m] (A itself stored in live-list order) instead of row_idx[m]
  int a_dense;            // with row_count (weights-in-registers kernel): A row of compacted row m is m itself; row_idx only scatters the output
};

// C-level launchers (enqueue only).  dtype selects T.
int nr_launch_gemm_nt(int dtype, const RowSrc& A, const void* B, int ldb, int M, int N, int K, int epi,
                      const EpiArgs& ep, hipStream_t stream);
// The k = 3 convolution's operand without an im2col buffer: out = [n * (T + 1) + 1, Dp] token rows (gather + dropout once),
// title blk's token t at row blk * (T + 1) + 1 + t, zero rows in between.  The im2col row of token m = (blk, t) is then
// the 3 * Dp CONTIGUOUS elements starting at row m + blk = m + m / T: a dense operand with ld = Dp, K = 3 Dp, gap = T.
// needed (optional [n]) / margin: titles farther than margin titles from every needed one are not written.
int nr_launch_conv_rows(int dtype, const RowSrc& A, void* out, int n, hipStream_t stream, const int32_t* needed = nullptr, int margin = 0);
// needed (optional, im2col rows): [M / Tlen] flags; titles farther than `margin` titles from every needed one are not written
// (gather rows: + L tokens per sequence and keep_all = device flag "table row 0 is not zero": all-padding sequences far
//  from every needed one are skipped)
int nr_launch_rows_materialize(int dtype, const RowSrc& A, void* out, int ldo, int M, int K, hipStream_t stream,
                               const int32_t* needed = nullptr, int margin = 0, int L = 0, const int32_t* keep_all = nullptr);
// ws: int32 [2*M + 4] -> ws[0] = number of rows with ids[m*stride] != 0, ws[4 ..] their row numbers, ws[4 + M ..] their ids
int nr_launch_compact_rows(const int32_t* ids, int ids_stride, int M, int32_t* ws, hipStream_t stream);
// live rows (count on the device, their row numbers and token ids) -> the same rows grouped by token id (counting sort:
// hist int32 [table_rows + 8] scratch).  Order inside a group is arbitrary.
// k_out (optional): position of each sorted row in the unsorted live list
int nr_launch_sort_rows_by_id(const int32_t* count, const int32_t* rows, const int32_t* ids, int Mmax, int table_rows, int32_t* hist,
                              int32_t* rows_out, int32_t* ids_out, hipStream_t stream, int32_t* k_out = nullptr,
                              int32_t* counted_cursor = nullptr);   // counted_cursor [table_rows]: hist is already counted (and stays intact)
// forward flavour: ws int32 [3*M + n + 4] (adds ws[1] = dead count, ws[2] = "table row 0 is not zero", ws[4+2M ..] dead rows,
// ws[4+3M ..] per-sequence live-token bit masks when L <= 32)
// posmap (optional [M]): position of every row in the live list (-1: dead)
int nr_launch_compact_rows_fwd(const int32_t* ids, int M, int n, int L, const void* table_row0, int cols, int32_t* ws,
                               hipStream_t stream, int32_t* posmap = nullptr, int32_t* zero4 = nullptr, int32_t* zhist = nullptr,
                               int nhist = 0);   // zero4 / zhist [nhist]: more counters to clear
// tmask (optional, with L): rows of sequences whose live-token mask is 0 are skipped (the attention kernels cover them)
int nr_launch_bias_rows(void* C, int ldc, int N, const float* bias, const int32_t* rows, const int32_t* count, int max_rows,
                        const uint32_t* tmask, int L, hipStream_t stream);
// weight gradient over the live 32-row slabs only (bf16 dense operands, tn3 shapes): see nr_launch_live_slabs
int nr_launch_title_flags(const void* dy, int n, int L, int N, int32_t* title_nz, hipStream_t stream);
int nr_launch_row_flags_f32(const float* g, int ld, int N, int n, int32_t* nz, hipStream_t stream);
int nr_launch_live_slabs(int32_t* ws, int n, int L, hipStream_t stream);
// reach >= 0: the y rows of an unneeded sequence are zero-filled only when a needed one lies within `reach` sequences
int nr_launch_needed_list(const int32_t* flags, int n, int32_t* out, void* y, size_t seq_bytes, hipStream_t stream, int reach = -1,
                          bool zeroed = false);   // zeroed: out[0..4) was cleared by an earlier kernel of the call
// reach: an all-padding sequence is left out when its own and `reach` neighbours' gradients are zero (-1: 32 / L + 2, the span of a slab)
int nr_launch_seq_list(const int32_t* title_nz, const uint32_t* tmask, int n, int L, int32_t* out, hipStream_t stream, int reach = -1,
                       bool zeroed = false);
// za [na] / zb [nb] (optional int32 regions): cleared by the same launch
int nr_launch_zero_tail_rows(void* buf, int ld, const int32_t* count, int Mmax, hipStream_t stream, int32_t* za = nullptr, int na = 0,
                             int32_t* zb = nullptr, int nb = 0);
// scratch (optional, nr_gemm_tn_scratch_floats(M, N, K) floats, 16-byte aligned): the splits store their partial tiles there
// and a reduce pass adds them up in order, instead of every split adding its tile into dW with fp32 atomics
int nr_launch_gemm_tn_slabs(const void* dC, int ldc, const void* X, int ldx, float* dW, int ldw, float* db, int M, int N, int K,
                            int Nstore, int Kstore, const int32_t* slab_list, const int32_t* slab_count, hipStream_t stream, int xgap = 0,
                            float* scratch = nullptr, size_t scratch_floats = 0);
size_t nr_gemm_tn_scratch_floats(int M, int N, int K);
bool nr_gemm_tn_slabs_ok(int ldc, int ldx, int M, int N, int K);
// fused additive-pooling forward (fc1 + tanh + fc2 + softmax + weighted sum), title-level shapes; see pool_fused_fwd_kernel
int nr_pool_fused_fwd_ok(int dtype, int n, int L, int N, int q, int ldw1);
int nr_launch_pool_fused_fwd(const void* x, int ldx, const void* w1, int ldw1, const float* b1, const float* w2, const float* b2,
                             const float* mask, void* e, int lde, float* alpha, float* out, int ld_out, int n, int L, int N, int q,
                             const int32_t* needed, hipStream_t stream);
int nr_launch_gemm_tn_counted(const void* dC, int ldc, const void* X, int ldx, float* dW, int ldw, int Mmax, int N, int K, int Nstore,
                              int Kstore, const int32_t* row_count, hipStream_t stream, float* scratch = nullptr, size_t scratch_floats = 0);
// fused additive-pooling backward core (dA, ds, dpre, dX, dw2 / db2 partials); see pool_fused_bwd_kernel
int nr_pool_fused_bwd_ok(int dtype, int n, int L, int N, int q, int ldw1t);
int nr_launch_pool_fused_bwd(const void* x, int ldx, const void* e, int lde, const float* alpha, const float* g, int ldg, const float* w2,
                             const void* w1t, int ldw1t, void* dpre, int ldp, void* dx, int lddx, float* partial, int partial_rows,
                             const int32_t* nz, int n, int L, int N, int q, hipStream_t stream, int* grid_out, int dx_far_unwritten = 0);
// live rows only, in live-list order (count / rows / ids of nr_launch_compact_rows_fwd), zero-filled to a multiple of 32 rows
int nr_launch_gather_live_rows(int dtype, const RowSrc& A, void* out, int ldo, int Mmax, int K, const int32_t* count, const int32_t* rows,
                               const int32_t* ids, hipStream_t stream, int32_t* hist = nullptr, int V = 0);   // hist [V]: occurrences per id, counted on the way
int nr_launch_gemm_tn(int dtype, const void* dC, int ldc, const RowSrc& A, float* dW, int ldw, float* db,
                      int M, int N, int K, int Nstore, int Kstore, hipStream_t stream);
