// extern "C" entry points of libnrhip that compose several launches (see include/nrhip.h).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nr_gemm.h"

// launchers defined in nr_attn.hip / nr_pool.hip
int nr_launch_attn(bool bwd, int dtype, const void* qkv, const float* mask, void* y, const void* dy, void* dqkv, int n, int L,
                   int heads, int d_head, const DropCfg& drop, hipStream_t stream, const uint32_t* tmask = nullptr,
                   const float* bias = nullptr, const int32_t* seq_list = nullptr, const int32_t* seq_count = nullptr,
                   const int32_t* needed = nullptr);
bool nr_attn_pad_ok(int dtype, int L, int d_head, const void* p0, const void* p1);
bool nr_attn_rowsub_ok(int dtype, int L, int d_head, int heads);
bool nr_attn_compact_ok(int dtype, int L, int d_head, int heads);
int nr_launch_attn_bwd_compact(const void* qkv, const float* mask, const void* dy, void* dqkv, int n, int L, int heads, int d_head,
                               const DropCfg& drop, hipStream_t stream, const uint32_t* tmask, const float* bias, const int32_t* seq_list,
                               const int32_t* seq_count, const int32_t* pos, void* dump, float* db, const int32_t* nzf);
int nr_launch_attn_gather_fwd(const void* proj_table, const int32_t* ids, const float* mask, void* y, int n, int L, int heads,
                              int d_head, const DropCfg& drop, hipStream_t stream);
int nr_launch_pool_core_fwd(int dtype, const void* x, const void* e, const float* w2, const float* b2, const float* mask,
                            float* alpha, float* out, int ld_out, int n, int L, int N, int q, hipStream_t s,
                            const int32_t* needed = nullptr);
int nr_launch_pool_core_bwd(int dtype, const void* x, const void* e, const float* w2, const float* alpha, const float* g,
                            int ld_g, void* dpre, float* partial, float* dw2, float* db2, int n, int L, int N, int q,
                            hipStream_t s, const int32_t* seq_nz = nullptr);
int nr_launch_cast_rows(int dtype, const float* src, int ld_src, void* dst, int ld_dst, int rows, int cols, hipStream_t s);
int nr_launch_colsum_split(const float* in, int rows, int cols, int ld, float* out, int split, float* out2, hipStream_t s);
int nr_pool_partial_rows(int n);
bool nr_mhsa_fused_shape_ok(int L, int heads, int d_head, int d_model, int ldt, int ldw);
int nr_launch_mhsa_fused_fwd(const void* table, int ldt, const int32_t* ids, const void* w, int ldw, const float* bias,
                             const float* mask, void* qkv, void* xrows, int ldxr, void* y, int n, int L, int heads, int d_head,
                             int d_model, const DropCfg& drop_in, const DropCfg& drop_out, hipStream_t stream);

#include <map>
#include <mutex>
#include <string>
#include <vector>

static thread_local char g_err[512] = "";

// ---- library options (nr_common.h: NrOpt) ------------------------------------------------------
#include <atomic>
namespace {
struct OptDef { const char* name; int def; };
const OptDef g_opt_defs[NR_OPT_COUNT] = {
    {"NO_SLABS", 0},   {"NO_ATTN_SKIP", 0}, {"SIDE_STREAM", 0}, {"ATTN_OLD", 0},  {"ATTN_VALU", 0},   {"NO_PAD_SUB", 0},
    {"NO_FUSED_FWD", 0}, {"NO_TN3", 0},     {"TN_V1", 0},       {"TN3_ROUNDS", 0}, {"TN3_WK", 0},      {"TN3_NI", 0},
    {"NT_NOWIDE", 0},  {"NT_NODMA", 0},     {"DMA_MIN_K", 192}, {"DMA_WM2_ALL", 0}, {"ATTN_PRED", 0}, {"ATTN_GENERIC", 0}, {"NO_ROW_SUB", 0}, {"ATTN_BWD_OCC4", 0}, {"NT_ABLATE", 0}, {"NT_WREG", 1}, {"NO_SCATTER_SORT", 0}, {"TN3_MIN_M", 16384}, {"NO_COMPACT_ROWS", 0}, {"NO_POOL_FUSED", 0}, {"ATTN_BWD_GRID", 0}, {"TN3_ATOMIC", 0}, {"TN3_ABLATE", 0}, {"POOL_ABLATE", 0}};
std::atomic<int> g_opt[NR_OPT_COUNT];
std::once_flag g_opt_once;
void opt_init() {
  for (int i = 0; i < NR_OPT_COUNT; ++i) {
    int v = g_opt_defs[i].def;
    char env[64];
    snprintf(env, sizeof(env), "NR_%s", g_opt_defs[i].name);
    const char* e = getenv(env);               // the ONLY environment lookup of the library, once per process
    if (e != nullptr) v = (e[0] >= '0' && e[0] <= '9') || e[0] == '-' ? atoi(e) : 1;
    g_opt[i].store(v, std::memory_order_relaxed);
  }
}
}  // namespace
int nr_opt(int which) {
  std::call_once(g_opt_once, opt_init);
  return (which >= 0 && which < NR_OPT_COUNT) ? g_opt[which].load(std::memory_order_relaxed) : 0;
}


// ---- per-kernel timing ---------------------------------------------------------------------
bool g_nr_prof_on = false;
namespace {
struct ProfEntry { std::string label; hipEvent_t a, b; };
std::mutex g_prof_mu;
std::vector<ProfEntry> g_prof_log;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;
int g_prof_mode = 1;   // 1: every launch; 2: only launches over >= 65 536 rows (the label carries M= / Mmax= / n=,L=)
// rows a launch works on, read off its label; -1 when the label has no such field
long prof_label_rows(const char* label) {
  const char* p;
  if ((p = strstr(label, "Mmax=")) != nullptr) return atol(p + 5);
  if ((p = strstr(label, "M=")) != nullptr) return atol(p + 2);
  if ((p = strstr(label, "max=")) != nullptr) return atol(p + 4);
  if ((p = strstr(label, "n=")) != nullptr) {
    const long n = atol(p + 2);
    const char* q = strstr(p, "L=");
    return q != nullptr ? n * atol(q + 2) : n;
  }
  return -1;
}
std::string g_prof_filter;   // mode 3: only labels that start with this
int prof_begin(const char* label, hipStream_t s) {
  if (g_prof_mode == 2 && prof_label_rows(label) < 65536) return -1;   // a pair of event records costs ~3 us of stream time
  if (g_prof_mode == 3 && strncmp(label, g_prof_filter.c_str(), g_prof_filter.size()) != 0) return -1;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfEntry e;
  e.label = label;
  if (!g_prof_pool.empty()) {
    e.a = g_prof_pool.back().first; e.b = g_prof_pool.back().second;
    g_prof_pool.pop_back();
  } else {
    if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return -1;
  }
  (void)hipEventRecord(e.a, s);
  g_prof_log.push_back(e);
  return (int)g_prof_log.size() - 1;
}
}  // namespace
NrProfScope::NrProfScope(const char* label, hipStream_t s) : idx(-1), stream(s) {
  if (g_nr_prof_on) idx = prof_begin(label, s);
}
NrProfScope::NrProfScope(hipStream_t s, const char* fmt, ...) : idx(-1), stream(s) {
  if (!g_nr_prof_on) return;
  char buf[192];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  idx = prof_begin(buf, s);
}
NrProfScope::~NrProfScope() {
  if (idx < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (idx < (int)g_prof_log.size()) (void)hipEventRecord(g_prof_log[idx].b, stream);
}

void nr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---- deterministic mode (nr_set_deterministic, nr_common.h: NrFixTable) ------------------------------------------
namespace {
std::atomic<long long*> g_det_ws{nullptr};
std::atomic<size_t> g_det_elems{0};
// Host copies of the range table must outlive the asynchronous symbol copies: a small per-thread ring.
thread_local NrFixTable g_fix_ring[64];
thread_local int g_fix_pos = 0;
struct DetScope {
  NrFixTable* t = nullptr;
  hipStream_t s;
  size_t used = 0;
  bool gemm = false, pool = false, begun = false;
  int rc = NR_OK;
  explicit DetScope(hipStream_t stream) : s(stream) {
    if (g_det_ws.load() != nullptr) {
      t = &g_fix_ring[g_fix_pos];
      g_fix_pos = (g_fix_pos + 1) & 63;
      memset(t, 0, sizeof(*t));
    }
  }
  bool on() const { return t != nullptr; }
  void add(float* base, size_t n) {
    if (!on() || base == nullptr || n == 0) return;
    if (t->count >= NR_FIX_RANGES || used + n > g_det_elems.load()) {
      nr_set_error("deterministic mode: the registered scratch holds %zu elements, this call needs more than %zu", g_det_elems.load(), used + n);
      rc = NR_ERR_ARG;
      return;
    }
    t->base[t->count] = base; t->fix[t->count] = g_det_ws.load() + used; t->n[t->count] = n;
    t->count++;
    used += n;
  }
  int begin(bool in_gemm, bool in_pool) {
    if (!on() || rc) return rc;
    gemm = in_gemm; pool = in_pool; begun = true;
    if (gemm && (rc = nr_fix_set_gemm(t, s))) return rc;
    if (pool && (rc = nr_fix_set_pool(t, s))) return rc;
    return NR_OK;
  }
  int end() {
    if (!begun) return NR_OK;
    begun = false;
    int r = nr_fix_flush(t, s);
    NrFixTable* z = &g_fix_ring[g_fix_pos];
    g_fix_pos = (g_fix_pos + 1) & 63;
    memset(z, 0, sizeof(*z));
    if (gemm) { const int r2 = nr_fix_set_gemm(z, s); if (!r) r = r2; }
    if (pool) { const int r2 = nr_fix_set_pool(z, s); if (!r) r = r2; }
    return r;
  }
  ~DetScope() { (void)end(); }
};
}  // namespace

namespace {
thread_local DetScope* g_det_open[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
}
int nr_det_open(hipStream_t s, float* base0, size_t n0, float* base1, size_t n1, bool gemm, bool pool, int* rc) {
  *rc = NR_OK;
  if (g_det_ws.load() == nullptr) return 0;
  for (int h = 0; h < 8; ++h)
    if (g_det_open[h] == nullptr) {
      DetScope* d = new DetScope(s);
      d->add(base0, n0);
      d->add(base1, n1);
      *rc = d->begin(gemm, pool);
      if (*rc) { delete d; return 0; }
      g_det_open[h] = d;
      return h + 1;
    }
  nr_set_error("deterministic mode: too many nested scopes");
  *rc = NR_ERR_ARG;
  return 0;
}
int nr_det_close(int handle) {
  if (handle <= 0 || handle > 8 || g_det_open[handle - 1] == nullptr) return NR_OK;
  DetScope* d = g_det_open[handle - 1];
  g_det_open[handle - 1] = nullptr;
  const int rc = d->end();
  delete d;
  return rc;
}

// ---- side stream: two independent GEMMs of one backward composite run concurrently -----------------------------
// The weight-gradient GEMM (LDS/MFMA bound) and the input-gradient GEMM (bound by L2 atomics or output stores)
// consume the same upstream gradient and do not depend on each other.  fork: the side stream waits for everything
// enqueued so far on the caller's stream; join: the caller's stream waits for the side stream.  Both are event
// waits on the device -- nothing blocks on the host, and memory the caller frees after the call is only reused
// by work ordered behind the join.
namespace {
struct SideStream { hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; int dev = -1; };
thread_local SideStream g_side;
// Opt-in (NR_SIDE_STREAM=1): measured 9.66 -> 9.43 ms/step at the bench shape; off by default because overlapped
// launches no longer have a per-kernel duration of their own, which the roofline accounting of bench.py relies on.
bool side_enabled() {
  return nr_opt(NR_OPT_SIDE_STREAM) != 0;
}
int side_fork(hipStream_t main, hipStream_t* out) {
  int dev = 0;
  NR_CHECK_HIP(hipGetDevice(&dev));
  if (g_side.s == nullptr || g_side.dev != dev) {
    NR_CHECK_HIP(hipStreamCreateWithFlags(&g_side.s, hipStreamNonBlocking));
    NR_CHECK_HIP(hipEventCreateWithFlags(&g_side.fork, hipEventDisableTiming));
    NR_CHECK_HIP(hipEventCreateWithFlags(&g_side.join, hipEventDisableTiming));
    g_side.dev = dev;
  }
  NR_CHECK_HIP(hipEventRecord(g_side.fork, main));
  NR_CHECK_HIP(hipStreamWaitEvent(g_side.s, g_side.fork, 0));
  *out = g_side.s;
  return NR_OK;
}
int side_join(hipStream_t main) {
  NR_CHECK_HIP(hipEventRecord(g_side.join, g_side.s));
  NR_CHECK_HIP(hipStreamWaitEvent(main, g_side.join, 0));
  return NR_OK;
}
}  // namespace
static inline bool dtype_ok(int dt) { return dt == NR_F32 || dt == NR_BF16; }

static RowSrc dense_rows(const void* base, int ld, int cols) {
  RowSrc s;
  memset(&s, 0, sizeof(s));
  s.base = base; s.ld = ld; s.kind = ROWS_DENSE; s.ids_stride = 1; s.Tlen = 1; s.Dtrue = cols;
  s.drop = nr_make_drop(0.f, 0);
  return s;
}

static EpiArgs store_epi(void* C, int ldc, int out_dtype, const float* bias, int act_tanh) {
  EpiArgs e;
  memset(&e, 0, sizeof(e));
  e.C = C; e.ldc = ldc; e.out_dtype = out_dtype; e.bias = bias; e.act_tanh = act_tanh;
  e.drop = nr_make_drop(0.f, 0);
  return e;
}

static int mhsa_rows(const nr_mhsa_desc* d, RowSrc* out) {
  RowSrc s = dense_rows(d->x, d->ldx, d->d_model);
  if (d->src_kind == NR_SRC_GATHER) {
    NR_CHECK_ARG(d->ids != nullptr, "mhsa: gather source without ids");
    s.kind = ROWS_GATHER;
    s.ids = d->ids;
  } else {
    NR_CHECK_ARG(d->src_kind == NR_SRC_DENSE, "mhsa: bad src_kind %d", d->src_kind);
  }
  s.drop = nr_make_drop(d->p_in, d->seed_in);
  *out = s;
  return NR_OK;
}

// ---- workspace layouts: the ONE place the formulas live (exported as nr_*_workspace_bytes) --------------------
// row_ws (int32): [0,4) counters | M live rows | M their ids | M padding rows | n per-sequence live-token masks |
//                 slab scratch: n title flags, 4 counters, M/32 slab ids, 4 pad | sequence list: 4 counters, n entries
struct MhsaWs {
  size_t live_idx, live_ids, dead_idx, tmask, slab, seq, sort_idx, sort_ids, hist, cursor, pos, sort_k, dump, tn_scratch, tn_floats, total;   // offsets in int32 elements
};
// table_rows > 0 (gather source): room for the live rows sorted by token id (table-gradient scatter) and its histogram
static MhsaWs mhsa_ws_layout(int n, int L, int table_rows, int N3 = 0, int Kp = 0) {
  const size_t M = (size_t)n * L;
  MhsaWs w;
  w.live_idx = 4; w.live_ids = 4 + M; w.dead_idx = 4 + 2 * M; w.tmask = 4 + 3 * M;
  w.slab = w.tmask + n;
  w.seq = w.slab + n + 4 + M / 32 + 4;
  w.sort_idx = (w.seq + 4 + n + 4 + 3) / 4 * 4;                      // 16-byte aligned: the GEMM stages these lists by DMA
  w.sort_ids = w.sort_idx + (table_rows > 0 ? (M + 3) / 4 * 4 : 0);
  w.hist = w.sort_ids + (table_rows > 0 ? (M + 3) / 4 * 4 : 0);
  // compact row storage (gather source): position of every row in the live list | the id-sorted rows' positions in it | one
  // dump row of 3N <= 2048 elements (what an all-padding sequence's gradient stores hit)
  w.cursor = w.hist + (table_rows > 0 ? (size_t)table_rows + 8 : 0);      // the scanned histogram (compact row storage: hist itself stays)
  w.pos = (w.cursor + (table_rows > 0 ? (size_t)table_rows + 8 : 0) + 3) / 4 * 4;
  w.sort_k = w.pos + (table_rows > 0 ? (M + 3) / 4 * 4 : 0);
  w.dump = w.sort_k + (table_rows > 0 ? (M + 3) / 4 * 4 : 0);
  // partial tiles of the weight-gradient GEMM's splits (fp32; see nr_launch_gemm_tn_slabs): store + reduce instead of atomics
  w.tn_scratch = (w.dump + (table_rows > 0 ? 1024 : 0) + 3) / 4 * 4;
  w.tn_floats = (table_rows > 0 && N3 >= 8 && Kp >= 8 && M % 32 == 0) ? nr_gemm_tn_scratch_floats((int)M, N3, Kp) : 0;
  w.total = w.tn_scratch + w.tn_floats;
  return w;
}
static MhsaWs mhsa_ws_of(const nr_mhsa_desc* d) {
  const bool gather = d->src_kind == NR_SRC_GATHER;
  const int ch = dtype_ok(d->dtype) ? nr_chunk(d->dtype) : 4;
  return mhsa_ws_layout(d->n, d->L, gather ? d->table_rows : 0, gather && d->dtype == NR_BF16 ? 3 * d->heads * d->d_head : 0,
                        gather && d->dtype == NR_BF16 ? round_up(d->d_model, ch) : 0);
}
// bwd_ws of the convolution (int32): n title flags | 4 counters | M/32 slab ids | pad
static size_t conv_ws_elems(int n, int T) { return (size_t)n + 4 + ((size_t)n * T) / 32 + 12; }
int nr_pool_partial_rows(int n);
// `partial` of the pooling backward (fp32): nr_pool_partial_rows(n) rows of (q+1) | int32 scratch: n flags, 4 counters, M/32 slabs
static size_t pool_ws_used(int n, int q) { return ((size_t)nr_pool_partial_rows(n) * (q + 1) + 3) / 4 * 4; }
static size_t pool_ws_ints(int n, int L, int q) { return (pool_ws_used(n, q) + (size_t)n + 8 + ((size_t)n * L) / 32 + 4 + 3) / 4 * 4; }
// ... | fp32 partial tiles of the att_fc1 weight-gradient GEMM's splits (store + reduce epilogue, nr_launch_gemm_tn_slabs)
static size_t pool_tn_floats(int n, int L, int q, int N, int dtype) {
  const size_t M = (size_t)n * L;
  return (dtype == NR_BF16 && q >= 8 && N >= 8 && M % 32 == 0 && M > 0) ? nr_gemm_tn_scratch_floats((int)M, q, N) : 0;
}
static size_t pool_ws_elems(int n, int L, int q, int N = 0, int dtype = NR_F32) { return pool_ws_ints(n, L, q) + pool_tn_floats(n, L, q, N, dtype); }

static bool pool_has_flags(const nr_pool_desc* d) {
  const int M = d->n * d->L;
  return !nr_opt(NR_OPT_NO_SLABS) && d->dtype == NR_BF16 && M % 32 == 0 && d->L <= 32 && nr_gemm_tn_slabs_ok(d->q, d->N, M, d->q, d->N);
}

// Shapes whose conv backward contracts live 32-row slabs only (bf16, rows a multiple of 32, tn3-eligible): only then may the
// forward leave the im2col rows of far-from-needed titles unwritten
static bool conv_slab_shape(const nr_conv_desc* d) {
  const int M = d->n * d->T;
  return !nr_opt(NR_OPT_NO_SLABS) && d->dtype == NR_BF16 && d->x_rows != nullptr && M % 32 == 0 && d->T <= 32 && d->N % 8 == 0 &&
         nr_gemm_tn_slabs_ok(d->N, d->ld_rows, M, d->N, 3 * d->Dp);
}

// Compact row storage of the news-level training path (bf16, gather source, x_rows + row_ws given): x_rows and dqkv hold ONLY
// the live rows (non-padding tokens), in live-list order -- the padding rows of both are never needed: a padding token
// gathers the zero row (the projection substitutes the bias, the weight gradient would contract a zero row, padding_idx
// gets no table gradient), and the one thing its dQ|dK|dV row feeds, the bias gradient, comes out of the attention backward
// itself.  Halves the gradient stores of the attention backward and lets the weight-gradient GEMM contract 0.33 instead of
// 0.57 of the rows of a MIND-shaped batch.  The forward and the backward call evaluate this same predicate.
static bool mhsa_compact_rows(const nr_mhsa_desc* d) {
  const int N = d->heads * d->d_head, M = d->n * d->L, Kp = round_up(d->d_model, nr_chunk(d->dtype));
  return !nr_opt(NR_OPT_NO_COMPACT_ROWS) && !nr_opt(NR_OPT_NO_SLABS) && !nr_opt(NR_OPT_NO_ATTN_SKIP) && !nr_opt(NR_OPT_NO_SCATTER_SORT) &&
         g_det_elems.load() == 0 && d->dtype == NR_BF16 && d->src_kind == NR_SRC_GATHER && d->x_rows != nullptr && d->row_ws != nullptr &&
         d->b_qkv != nullptr && d->table_rows > 0 && M >= 4096 && M % 32 == 0 && (3 * N) % 8 == 0 && 3 * N <= 2048 && d->d_model % 4 == 0 &&
         d->ld_rows >= Kp && nr_attn_compact_ok(d->dtype, d->L, d->d_head, d->heads) && nr_gemm_tn_slabs_ok(3 * N, d->ld_rows, M, 3 * N, Kp);
}

static int mhsa_check(const nr_mhsa_desc* d) {
  NR_CHECK_ARG(d != nullptr, "mhsa: null descriptor");
  NR_CHECK_ARG(dtype_ok(d->dtype), "mhsa: bad dtype %d", d->dtype);
  NR_CHECK_ARG(d->n >= 0 && d->L >= 1 && d->L <= 64 && d->d_model >= 1 && d->heads >= 1 && d->d_head >= 1,
               "mhsa: bad shape n=%d L=%d d_model=%d heads=%d d_head=%d", d->n, d->L, d->d_model, d->heads, d->d_head);
  const int ch = nr_chunk(d->dtype);
  NR_CHECK_ARG((3 * d->heads * d->d_head) % ch == 0, "mhsa: 3*news_dim=%d must be a multiple of %d for this dtype",
               3 * d->heads * d->d_head, ch);
  NR_CHECK_ARG(d->ldx >= round_up(d->d_model, ch) && d->ldw >= round_up(d->d_model, ch),
               "mhsa: ldx=%d / ldw=%d must cover d_model=%d rounded up to %d", d->ldx, d->ldw, d->d_model, ch);
  NR_CHECK_ARG(d->n == 0 || (d->x && d->w_qkv && d->b_qkv), "mhsa: null operand");
  NR_CHECK_ARG(d->p_in >= 0.f && d->p_in < 1.f && d->p_out >= 0.f && d->p_out < 1.f, "mhsa: dropout p out of range");
  NR_CHECK_ARG((uint64_t)d->n * d->L * (uint64_t)(3 * d->heads * d->d_head) < 0xffffffffull, "mhsa: problem too large for 32-bit element counters");
  NR_CHECK_ARG(d->row_ws == nullptr || d->row_ws_bytes >= mhsa_ws_of(d).total * sizeof(int32_t),
               "mhsa: row_ws holds %zu bytes, nr_mhsa_workspace_bytes() asks for %zu", d->row_ws_bytes,
               mhsa_ws_of(d).total * sizeof(int32_t));
  return NR_OK;
}

extern "C" {

int nr_version(void) { return 200; }

int nr_prof_enable(int on) {
  g_nr_prof_on = on != 0;
  g_prof_mode = on == 2 ? 2 : (on == 3 ? 3 : 1);
  return NR_OK;
}

int nr_prof_filter(const char* label_prefix) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_filter = label_prefix ? label_prefix : "";
  return NR_OK;
}

int nr_prof_collect(char* buf, size_t n) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  std::map<std::string, std::pair<int, double>> agg;
  std::vector<std::string> order;
  for (auto& e : g_prof_log) {
    float ms = 0.f;
    if (hipEventSynchronize(e.b) != hipSuccess || hipEventElapsedTime(&ms, e.a, e.b) != hipSuccess) {
      nr_set_error("prof_collect: event query failed for %s", e.label.c_str());
      return -NR_ERR_HIP;
    }
    if (!agg.count(e.label)) order.push_back(e.label);
    auto& a = agg[e.label];
    a.first += 1;
    a.second += ms;
    g_prof_pool.emplace_back(e.a, e.b);
  }
  g_prof_log.clear();
  size_t off = 0;
  for (auto& l : order) {
    char line[256];
    int w = snprintf(line, sizeof(line), "%s\t%d\t%.6f\n", l.c_str(), agg[l].first, agg[l].second);
    if (buf && off + (size_t)w < n) {
      memcpy(buf + off, line, (size_t)w);
      off += (size_t)w;
    }
  }
  if (buf && n) buf[off < n ? off : n - 1] = 0;
  return (int)off;
}

int nr_last_error(char* buf, size_t n) {
  if (buf && n) {
    strncpy(buf, g_err, n - 1);
    buf[n - 1] = 0;
  }
  return (int)strlen(g_err);
}

int nr_abi_sizes(size_t* out, int n) {
  NR_CHECK_ARG(out != nullptr && n >= 4, "abi_sizes: need room for 4 entries");
  out[0] = sizeof(nr_mhsa_desc); out[1] = sizeof(nr_conv_desc); out[2] = sizeof(nr_pool_desc); out[3] = sizeof(nr_linear_desc);
  if (n >= 6) { out[4] = sizeof(nr_cast_job); out[5] = sizeof(nr_pack_job); }
  return NR_OK;
}

int nr_set_deterministic(void* scratch, size_t bytes) {
  NR_CHECK_ARG(scratch == nullptr || (bytes >= 8 && (((uintptr_t)scratch) & 7) == 0), "set_deterministic: scratch must be 8-byte aligned and non-empty");
  g_det_elems.store(scratch ? bytes / 8 : 0);
  g_det_ws.store(reinterpret_cast<long long*>(scratch));
  return NR_OK;
}

int nr_set_option(const char* name, int value) {
  NR_CHECK_ARG(name != nullptr, "set_option: null name");
  (void)nr_opt(0);                                 // environment presets first, so that this call wins
  if (strncmp(name, "NR_", 3) == 0) name += 3;
  for (int i = 0; i < NR_OPT_COUNT; ++i)
    if (strcmp(name, g_opt_defs[i].name) == 0) {
      g_opt[i].store(value, std::memory_order_relaxed);
      return NR_OK;
    }
  nr_set_error("set_option: unknown option %s", name);
  return NR_ERR_ARG;
}

int nr_get_option(const char* name) {
  if (name == nullptr) return -1;
  if (strncmp(name, "NR_", 3) == 0) name += 3;
  for (int i = 0; i < NR_OPT_COUNT; ++i)
    if (strcmp(name, g_opt_defs[i].name) == 0) return nr_opt(i);
  return -1;
}

size_t nr_mhsa_workspace_bytes(const nr_mhsa_desc* d) {
  return (d == nullptr || d->n < 0 || d->L < 1) ? 0 : mhsa_ws_of(d).total * sizeof(int32_t);
}
size_t nr_conv_workspace_bytes(const nr_conv_desc* d) {
  return (d == nullptr || d->n < 0 || d->T < 1) ? 0 : conv_ws_elems(d->n, d->T) * sizeof(int32_t);
}
int nr_pool_contracts_slabs(const nr_pool_desc* d) {
  return (d != nullptr && d->n > 0 && d->L >= 1 && d->q >= 1 && dtype_ok(d->dtype) && pool_has_flags(d)) ? 1 : 0;
}
size_t nr_pool_workspace_bytes(const nr_pool_desc* d) {
  return (d == nullptr || d->n < 0 || d->L < 1 || d->q < 1) ? 0 : pool_ws_elems(d->n, d->L, d->q, d->N, d->dtype) * sizeof(float);
}
const int32_t* nr_pool_seq_flags(const nr_pool_desc* d, const float* partial) {
  if (d == nullptr || partial == nullptr || d->n <= 0 || d->L < 1 || d->q < 1 || !dtype_ok(d->dtype) || !pool_has_flags(d)) return nullptr;
  return reinterpret_cast<const int32_t*>(partial + pool_ws_used(d->n, d->q));
}
size_t nr_linear_workspace_bytes(const nr_linear_desc* d) {
  if (d == nullptr || d->M < 0 || d->N < 1 || !dtype_ok(d->dtype)) return 0;
  return (size_t)d->M * round_up(d->N, nr_chunk(d->dtype)) * nr_elt_size(d->dtype);
}

// ---------------------------------------------------------------------------------------- index validation
namespace {
__global__ __launch_bounds__(256) void check_ids_kernel(const int32_t* __restrict__ ids, int count, int stride, int rows,
                                                        int32_t* __restrict__ bad) {
  int nbad = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) {
    const int v = ids[i * stride];
    nbad += (v < 0 || v >= rows) ? 1 : 0;
  }
  nbad = (int)wave_sum((float)nbad);
  if ((threadIdx.x & 63) == 0 && nbad) atomicAdd(bad, nbad);
}
__global__ __launch_bounds__(256) void check_labels_kernel(const int64_t* __restrict__ label, int count, int classes,
                                                           int32_t* __restrict__ bad) {
  int nbad = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) {
    const int64_t v = label[i];
    nbad += (v < 0 || v >= classes) ? 1 : 0;
  }
  nbad = (int)wave_sum((float)nbad);
  if ((threadIdx.x & 63) == 0 && nbad) atomicAdd(bad, nbad);
}
}  // namespace

int nr_check_ids(const int32_t* ids, int count, int stride, int rows, int32_t* bad, nr_stream_t stream) {
  NR_CHECK_ARG(count >= 0 && stride >= 1 && rows >= 0 && bad != nullptr && (count == 0 || ids != nullptr), "check_ids: bad arguments");
  if (count == 0) return NR_OK;
  NR_DEVICE_GUARD(stream, bad);
  const int blocks = count / 256 + 1 > 1024 ? 1024 : count / 256 + 1;
  hipLaunchKernelGGL(check_ids_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ids, count, stride, rows, bad);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_check_labels(const int64_t* label, int count, int classes, int32_t* bad, nr_stream_t stream) {
  NR_CHECK_ARG(count >= 0 && classes >= 0 && bad != nullptr && (count == 0 || label != nullptr), "check_labels: bad arguments");
  if (count == 0) return NR_OK;
  NR_DEVICE_GUARD(stream, bad);
  const int blocks = count / 256 + 1 > 1024 ? 1024 : count / 256 + 1;
  hipLaunchKernelGGL(check_labels_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, label, count, classes, bad);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

// ---------------------------------------------------------------------------------------- attention core alone
static int sdpa_check(const void* qkv, int n, int L, int heads, int d_head, int dtype, float p_out) {
  NR_CHECK_ARG(dtype_ok(dtype), "sdpa: bad dtype %d", dtype);
  NR_CHECK_ARG(n >= 0 && L >= 1 && L <= 64 && heads >= 1 && d_head >= 1, "sdpa: bad shape n=%d L=%d heads=%d d_head=%d", n, L, heads, d_head);
  NR_CHECK_ARG(n == 0 || qkv != nullptr, "sdpa: null operand");
  NR_CHECK_ARG(p_out >= 0.f && p_out < 1.f, "sdpa: dropout p out of range");
  NR_CHECK_ARG((uint64_t)n * L * (uint64_t)(3 * heads * d_head) < 0xffffffffull, "sdpa: problem too large for 32-bit element counters");
  return NR_OK;
}

int nr_sdpa_fwd(const void* qkv, const float* mask, void* y, int n, int L, int heads, int d_head, int dtype, float p_out,
                uint32_t seed_out, nr_stream_t stream) {
  int rc = sdpa_check(qkv, n, L, heads, d_head, dtype, p_out);
  if (rc) return rc;
  if (n == 0) return NR_OK;
  NR_CHECK_ARG(y != nullptr, "sdpa_fwd: null output");
  NR_DEVICE_GUARD(stream, y);
  return nr_launch_attn(false, dtype, qkv, mask, y, nullptr, nullptr, n, L, heads, d_head, nr_make_drop(p_out, seed_out), (hipStream_t)stream);
}

int nr_sdpa_bwd(const void* qkv, const float* mask, const void* dy, void* dqkv, int n, int L, int heads, int d_head, int dtype,
                float p_out, uint32_t seed_out, nr_stream_t stream) {
  int rc = sdpa_check(qkv, n, L, heads, d_head, dtype, p_out);
  if (rc) return rc;
  if (n == 0) return NR_OK;
  NR_CHECK_ARG(dy != nullptr && dqkv != nullptr, "sdpa_bwd: null operand");
  NR_DEVICE_GUARD(stream, dqkv);
  return nr_launch_attn(true, dtype, qkv, mask, nullptr, dy, dqkv, n, L, heads, d_head, nr_make_drop(p_out, seed_out), (hipStream_t)stream);
}

int nr_gemm_nt(int dtype, const void* A, int lda, const void* B, int ldb, const float* bias, int act_tanh, void* C, int ldc,
               int out_dtype, int M, int N, int K, nr_stream_t stream) {
  NR_CHECK_ARG(dtype_ok(dtype) && dtype_ok(out_dtype) && A && B && C, "gemm_nt: bad dtype / null operand");
  NR_DEVICE_GUARD(stream, C);
  RowSrc a = dense_rows(A, lda, K);
  EpiArgs ep = store_epi(C, ldc, out_dtype, bias, act_tanh);
  return nr_launch_gemm_nt(dtype, a, B, ldb, M, N, K, EPI_STORE, ep, (hipStream_t)stream);
}

int nr_gemm_tn(int dtype, const void* dC, int ldc, const void* A, int lda, float* dW, int ldw, float* db, int M, int N, int K,
               nr_stream_t stream) {
  NR_CHECK_ARG(dtype_ok(dtype) && dC && A && dW, "gemm_tn: bad dtype / null operand");
  NR_DEVICE_GUARD(stream, dW);
  RowSrc a = dense_rows(A, lda, K);
  return nr_launch_gemm_tn(dtype, dC, ldc, a, dW, ldw, db, M, N, K, N, K, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------- MHSA
int nr_mhsa_fwd_fused(const nr_mhsa_desc* d) {
  if (d == nullptr) return 0;
  return d->dtype == NR_BF16 && d->src_kind == NR_SRC_GATHER && (((uintptr_t)d->x | (uintptr_t)d->w_qkv) & 15) == 0 &&
                 nr_mhsa_fused_shape_ok(d->L, d->heads, d->d_head, d->d_model, d->ldx, d->ldw)
             ? 1
             : 0;
}

int nr_mhsa_compact_rows(const nr_mhsa_desc* d) { return (d != nullptr && mhsa_check(d) == NR_OK && d->n > 0 && mhsa_compact_rows(d)) ? 1 : 0; }

int nr_mhsa_fwd(const nr_mhsa_desc* d, void* qkv, void* y, nr_stream_t stream) {
  int rc = mhsa_check(d);
  if (rc) return rc;
  if (d->n == 0) return NR_OK;
  NR_CHECK_ARG(y != nullptr, "mhsa_fwd: null output");
  hipStream_t s = (hipStream_t)stream;
  NR_DEVICE_GUARD(stream, y);
  const int N = d->heads * d->d_head, M = d->n * d->L, Kp = round_up(d->d_model, nr_chunk(d->dtype));
  const MhsaWs W = mhsa_ws_of(d);
  RowSrc A;
  if ((rc = mhsa_rows(d, &A))) return rc;
  if (d->proj_table != nullptr) {
    // Eval mode (no input dropout): Q|K|V of a token are W x + b of its table row -- a function of the token id alone.  The
    // caller projected the whole table once (nr_gemm_nt over [V, d_model]); the attention kernel gathers projected rows.
    NR_CHECK_ARG(d->src_kind == NR_SRC_GATHER && d->p_in == 0.f && d->dtype == NR_BF16 && d->ids != nullptr && qkv == nullptr,
                 "mhsa_fwd: proj_table needs a bf16 gather source without input dropout and no qkv buffer (no backward)");
    rc = nr_launch_attn_gather_fwd(d->proj_table, d->ids, d->mask, y, d->n, d->L, d->heads, d->d_head, nr_make_drop(d->p_out, d->seed_out), s);
    NR_CHECK_ARG(rc >= 0, "mhsa_fwd: proj_table is only supported for L <= 64, d_head %% 4 == 0 (<= 32), 8-byte aligned tensors");
    return rc;
  }
  if (qkv == nullptr && d->dtype == NR_BF16 && d->src_kind == NR_SRC_GATHER) {
    // Inference (no backward -> the caller passes no qkv buffer): one fused kernel, gather + projection + attention with
    // Q|K|V kept on chip.  Measured on MI355X it ties the unfused chain in time (2.8 ms per 28 160 titles) and saves
    // the 2 GB Q|K|V round trip; with a backward pending the unfused chain is used because it has to write Q|K|V anyway.
    rc = nr_launch_mhsa_fused_fwd(d->x, d->ldx, d->ids, d->w_qkv, d->ldw, d->b_qkv, d->mask, nullptr, d->x_rows, d->ld_rows, y, d->n,
                                  d->L, d->heads, d->d_head, d->d_model, nr_make_drop(d->p_in, d->seed_in),
                                  nr_make_drop(d->p_out, d->seed_out), s);
    if (rc >= 0) return rc;
  }
  NR_CHECK_ARG(qkv != nullptr, "mhsa_fwd: the unfused path needs the qkv buffer (see nr_mhsa_fwd_fused)");
  EpiArgs ep = store_epi(qkv, 3 * N, d->dtype, d->b_qkv, 0);
  const uint32_t* tmask = nullptr;
  bool seq_hdr_zeroed = false;
  if (d->x_rows != nullptr && d->src_kind == NR_SRC_GATHER) {
    // gather + dropout once into x_rows (kept for the backward), then a plain dense projection GEMM
    NR_CHECK_ARG(d->ld_rows >= Kp && d->ld_rows % nr_chunk(d->dtype) == 0, "mhsa_fwd: ld_rows=%d must cover %d", d->ld_rows, Kp);
    const bool compacting = d->row_ws != nullptr && d->dtype == NR_BF16 && M >= 4096 && (3 * N) % 8 == 0;
    const bool cstore = compacting && mhsa_compact_rows(d);     // x_rows holds the live rows only, in live-list order
    // Padding tokens (id 0) gather the zero row of the table: their projection is the bias.  Project the live rows
    // only (compacted on the device) and write the bias into the others.  If table row 0 is not zero, the
    // compaction keeps every row and nothing changes.
    // (the compaction kernel also clears the counters of the "needed" list built further down: one memset less in the chain)
    if (compacting && (rc = nr_launch_compact_rows_fwd(d->ids, M, d->n, d->L, d->x, d->d_model, d->row_ws, s, cstore ? d->row_ws + W.pos : nullptr,
                                                       d->row_ws + W.seq, cstore ? d->row_ws + W.hist : nullptr, d->table_rows)))
      return rc;
    seq_hdr_zeroed = compacting;
    if (cstore) {
      NR_CHECK_ARG(nr_attn_pad_ok(d->dtype, d->L, d->d_head, qkv, y), "mhsa_fwd: qkv / y must be 8-byte aligned");
      if ((rc = nr_launch_gather_live_rows(d->dtype, A, d->x_rows, d->ld_rows, M, Kp, d->row_ws, d->row_ws + W.live_idx, d->row_ws + W.live_ids, s,
                                           d->row_ws + W.hist, d->table_rows)))       // (+ the id histogram for the backward's sort)
        return rc;
      A = dense_rows(d->x_rows, d->ld_rows, d->d_model);
      ep.row_count = d->row_ws; ep.row_idx = d->row_ws + W.live_idx; ep.row_ids = d->row_ws + W.live_ids;
      ep.a_dense = 1;                                  // A row k = live row k; row_idx scatters the output rows
      tmask = reinterpret_cast<const uint32_t*>(d->row_ws + W.tmask);      // per-row substitution: no bias rows are written
    } else {
    // under "needed" flags the rows of all-padding sequences no weight-gradient slab can reach are not even written
    // (only where the backward contracts live slabs: a dense contraction would read every row)
    const bool skip_far = compacting && d->seq_needed != nullptr && M % 32 == 0 && nr_gemm_tn_slabs_ok(3 * N, d->ld_rows, M, 3 * N, Kp) &&
                          !nr_opt(NR_OPT_NO_SLABS);
    if ((rc = nr_launch_rows_materialize(d->dtype, A, d->x_rows, d->ld_rows, M, Kp, s, skip_far ? d->seq_needed : nullptr, 32 / d->L + 2,
                                         d->L, skip_far ? d->row_ws + 2 : nullptr)))
      return rc;
    A = dense_rows(d->x_rows, d->ld_rows, d->d_model);
    if (compacting) {
      ep.row_count = d->row_ws; ep.row_idx = d->row_ws + W.live_idx; ep.row_ids = d->row_ws + W.live_ids;
      // Sequences made of padding tokens only (empty history slots, ~45 % of the titles of a MIND-shaped batch): the
      // attention kernels take their Q|K|V from the bias themselves (per-sequence live mask == 0), so those qkv rows are
      // neither written here nor read there.  The bias goes into the padding rows of the other sequences.
      if (d->b_qkv != nullptr && nr_attn_pad_ok(d->dtype, d->L, d->d_head, qkv, y))
        tmask = reinterpret_cast<const uint32_t*>(d->row_ws + W.tmask);
      // ... unless the attention kernels substitute per row (then no padding row of qkv is ever written or read)
      if (!(tmask != nullptr && nr_attn_rowsub_ok(d->dtype, d->L, d->d_head, d->heads)) &&
          (rc = nr_launch_bias_rows(qkv, 3 * N, 3 * N, d->b_qkv, d->row_ws + W.dead_idx, d->row_ws + 1, M, tmask, d->L, s)))
        return rc;
    }
    }
  }
  if ((rc = nr_launch_gemm_nt(d->dtype, A, d->w_qkv, d->ldw, M, 3 * N, Kp, EPI_STORE, ep, s))) return rc;
  // "output not needed" flags: with the compaction scratch at hand the attention kernel walks a device-side list of the
  // needed sequences and a store-only kernel zero-fills the y rows of the others; without it the kernel skips in place
  const int32_t* fwd_list = nullptr;
  if (d->seq_needed != nullptr && tmask != nullptr && ((size_t)d->L * N * nr_elt_size(d->dtype)) % 16 == 0 && (((uintptr_t)y) & 15) == 0) {
    int32_t* lw = d->row_ws + W.seq;
    if ((rc = nr_launch_needed_list(d->seq_needed, d->n, lw, y, (size_t)d->L * N * nr_elt_size(d->dtype), s,
                                    d->y_far_unwritten ? 32 / d->L + 2 : -1, seq_hdr_zeroed)))
      return rc;
    fwd_list = lw;
  }
  return nr_launch_attn(false, d->dtype, qkv, d->mask, y, nullptr, nullptr, d->n, d->L, d->heads, d->d_head,
                        nr_make_drop(d->p_out, d->seed_out), s, tmask, tmask ? d->b_qkv : nullptr, fwd_list ? fwd_list + 4 : nullptr, fwd_list,
                        fwd_list ? nullptr : d->seq_needed);
}

int nr_mhsa_bwd(const nr_mhsa_desc* d, const void* qkv, const void* dy, void* dqkv, const void* w_qkv_t, int ldwt,
                float* dw_qkv, float* db_qkv, void* dx, float* dtable, nr_stream_t stream) {
  int rc = mhsa_check(d);
  if (rc) return rc;
  if (d->n == 0) return NR_OK;
  NR_CHECK_ARG(qkv && dy && dqkv && dw_qkv && db_qkv, "mhsa_bwd: null operand");
  hipStream_t s = (hipStream_t)stream;
  NR_DEVICE_GUARD(stream, dqkv);
  const int N = d->heads * d->d_head, M = d->n * d->L, ch = nr_chunk(d->dtype), Kp = round_up(d->d_model, ch);
  const MhsaWs W = mhsa_ws_of(d);
  RowSrc A;
  if ((rc = mhsa_rows(d, &A))) return rc;
  NR_CHECK_ARG(d->bwd_phase >= 0 && d->bwd_phase <= 2, "mhsa_bwd: bwd_phase=%d", d->bwd_phase);
  const bool ph_main = d->bwd_phase != 2;          // flags, attention backward, dx / table gradient
  const bool ph_dw = d->bwd_phase != 1;            // weight / bias gradients
  // row_ws_ready: the forward compacted the rows and (when the attention kernels support it) left the qkv rows of
  // padding tokens unwritten -- the backward attention must substitute the bias exactly as the forward one did
  const uint32_t* tmask = nullptr;
  if (d->row_ws != nullptr && d->row_ws_ready && d->src_kind == NR_SRC_GATHER && d->dtype == NR_BF16 && M >= 4096 &&
      (3 * N) % 8 == 0 && d->b_qkv != nullptr && d->x_rows != nullptr && nr_attn_pad_ok(d->dtype, d->L, d->d_head, nullptr, nullptr)) {
    // shape-wise the forward substituted; whether it really did also hung on the alignment of its qkv / y
    NR_CHECK_ARG(nr_attn_pad_ok(d->dtype, d->L, d->d_head, qkv, dqkv) && (((uintptr_t)dy) & 7) == 0,
                 "mhsa_bwd: the forward left padding rows of qkv unwritten; qkv / dy / dqkv must be 8-byte aligned");
    tmask = reinterpret_cast<const uint32_t*>(d->row_ws + W.tmask);
  }
  if (d->row_ws_ready && mhsa_compact_rows(d)) {
    // ---- compact row storage (see mhsa_compact_rows): x_rows holds the live rows in live-list order (forward), dqkv gets them
    // in the same order; dW = dqkv_c^T . x_c over `count` dense rows, db from the attention kernel, the table gradient reads
    // dqkv_c through the id-sorted positions
    NR_CHECK_ARG(tmask != nullptr, "mhsa_bwd: the forward stored x_rows compactly; qkv / dy / dqkv must be 8-byte aligned");
    NR_CHECK_ARG(!d->dy_far_unwritten || d->seq_nz != nullptr, "mhsa_bwd: dy_far_unwritten needs the seq_nz flags");
    NR_CHECK_ARG(dtable != nullptr && dx == nullptr && w_qkv_t != nullptr && ldwt >= 3 * N, "mhsa_bwd: gather source takes dtable (and w_qkv_t [d_model, >=3N])");
    int32_t* ws = d->row_ws;
    if (ph_main) {
      // n sequence flags: which sequences got a non-zero upstream gradient (the caller's, read in place, or made here)
      const int32_t* slab_ws = d->seq_nz;
      if (slab_ws == nullptr) {
        if ((rc = nr_launch_title_flags(dy, d->n, d->L, N, ws + W.slab, s))) return rc;
        slab_ws = ws + W.slab;
      }
      // rows count .. roundup32(count) of dqkv: the weight-gradient GEMM contracts whole 32-row slabs (x_c is zero there);
      // the same launch clears the counters of the sequence list that follows
      int32_t* seq_ws = ws + W.seq;
      if ((rc = nr_launch_zero_tail_rows(dqkv, 3 * N, ws, M, s, seq_ws, 4))) return rc;
      // the walk leaves out the all-padding sequences with a zero gradient (nothing to store, nothing to add to db); slab
      // distances do not matter any more -- no slab is contracted -- so the list kernel runs with a zero reach
      if ((rc = nr_launch_seq_list(slab_ws, tmask, d->n, d->L, seq_ws, s, /*reach=*/0, /*zeroed=*/true))) return rc;
      if ((rc = nr_launch_attn_bwd_compact(qkv, d->mask, dy, dqkv, d->n, d->L, d->heads, d->d_head, nr_make_drop(d->p_out, d->seed_out), s, tmask,
                                           d->b_qkv, seq_ws + 4, seq_ws, ws + W.pos, ws + W.dump, db_qkv, d->dy_far_unwritten ? slab_ws : nullptr)))
        return rc;
      // table gradient: rows in token-id order, A rows through their live-list positions
      if ((rc = nr_launch_sort_rows_by_id(ws, ws + W.live_idx, ws + W.live_ids, M, d->table_rows, ws + W.hist, ws + W.sort_idx, ws + W.sort_ids, s,
                                          ws + W.sort_k, /*histogram counted by the forward:*/ws + W.cursor)))
        return rc;
      EpiArgs ep = store_epi(dtable, d->d_model, NR_F32, nullptr, 0);
      ep.ids = d->ids; ep.ids_stride = 1; ep.Dtrue = d->d_model; ep.drop = nr_make_drop(d->p_in, d->seed_in);
      ep.row_count = ws; ep.row_idx = ws + W.sort_idx; ep.row_ids = ws + W.sort_ids; ep.a_idx = ws + W.sort_k;
      RowSrc G = dense_rows(dqkv, 3 * N, 3 * N);
      if ((rc = nr_launch_gemm_nt(d->dtype, G, w_qkv_t, ldwt, M, d->d_model, 3 * N, EPI_SCATTER, ep, s))) return rc;
    }
    if (ph_dw && (rc = nr_launch_gemm_tn_counted(dqkv, 3 * N, d->x_rows, d->ld_rows, dw_qkv, d->d_model, M, 3 * N, Kp, 3 * N, d->d_model, ws, s,
                                                 W.tn_floats ? reinterpret_cast<float*>(ws + W.tn_scratch) : nullptr, W.tn_floats)))
      return rc;
    return NR_OK;
  }
  NR_CHECK_ARG(!d->dy_far_unwritten, "mhsa_bwd: dy_far_unwritten is only honoured with compact row storage (nr_mhsa_compact_rows)");
  // Sequences whose upstream gradient dy is exactly zero (history slots the user encoder masks out) get exact zeros in
  // dQ|dK|dV (dP = dy.V^T = 0, so dS = 0): a pass over dy flags the others, and the weight-gradient GEMM contracts only
  // the 32-row slabs that touch a flagged sequence.  Scratch: the tail of row_ws (n flags, count, M/32 slab ids).
  const bool no_slabs = nr_opt(NR_OPT_NO_SLABS) != 0;
  int32_t* slab_ws = nullptr;
  int32_t* seq_ws = nullptr;
  if (!no_slabs && d->row_ws != nullptr && d->dtype == NR_BF16 && d->src_kind == NR_SRC_GATHER && d->x_rows != nullptr && M % 32 == 0 &&
      M >= 4096 && nr_attn_pad_ok(d->dtype, d->L, d->d_head, qkv, dqkv) && (((uintptr_t)dy) & 7) == 0 &&
      nr_gemm_tn_slabs_ok(3 * N, d->ld_rows, M, 3 * N, Kp)) {
    slab_ws = d->row_ws + W.slab;
    if (ph_main) {                                 // (phase 2 finds the lists of its phase-1 call in row_ws)
      if (d->seq_nz != nullptr) {                  // the consumer of y already knows which sequences got a gradient
        NR_CHECK_HIP(hipMemcpyAsync(slab_ws, d->seq_nz, (size_t)d->n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
      } else if ((rc = nr_launch_title_flags(dy, d->n, d->L, N, slab_ws, s))) {      // one pass over dy (bf16 [M, N])
        return rc;
      }
      if ((rc = nr_launch_live_slabs(slab_ws, d->n, d->L, s))) return rc;
    }
    const bool no_skip = nr_opt(NR_OPT_NO_ATTN_SKIP) != 0;
    if (tmask != nullptr && !no_skip) {
      // the attention backward walks a list that leaves out the all-padding sequences no live slab comes near
      seq_ws = d->row_ws + W.seq;
      if (ph_main && (rc = nr_launch_seq_list(slab_ws, tmask, d->n, d->L, seq_ws, s))) return rc;
    }
  }
  // x_rows written under "needed" flags (nr_mhsa_fwd: skip_far) are complete only where a live slab can reach: the slab
  // path is then mandatory (a dense contraction would multiply unwritten rows by their zero gradient)
  NR_CHECK_ARG(!(d->row_ws_ready && d->seq_needed != nullptr && d->row_ws != nullptr && d->dtype == NR_BF16 && d->src_kind == NR_SRC_GATHER &&
                 d->x_rows != nullptr && M >= 4096 && (3 * N) % 8 == 0 && M % 32 == 0 && nr_gemm_tn_slabs_ok(3 * N, d->ld_rows, M, 3 * N, Kp) &&
                 !no_slabs) || slab_ws != nullptr,
               "mhsa_bwd: x_rows were materialised under seq_needed: qkv / dqkv / dy must be 8-byte aligned so that the live-slab path runs");
  if (ph_main && (rc = nr_launch_attn(true, d->dtype, qkv, d->mask, nullptr, dy, dqkv, d->n, d->L, d->heads, d->d_head,
                                      nr_make_drop(d->p_out, d->seed_out), s, tmask, tmask ? d->b_qkv : nullptr,
                                      seq_ws ? seq_ws + 4 : nullptr, seq_ws)))
    return rc;
  // dW_qkv[3N, d_model] += dQKV^T . X ; db += colsum(dQKV).  X: the rows saved by the forward when present.
  RowSrc Xs = A;
  if (d->x_rows != nullptr && d->src_kind == NR_SRC_GATHER) {
    NR_CHECK_ARG(d->ld_rows >= Kp, "mhsa_bwd: ld_rows=%d must cover %d", d->ld_rows, Kp);
    Xs = dense_rows(d->x_rows, d->ld_rows, d->d_model);
  }
  const bool want_dx = dx != nullptr || dtable != nullptr;
  DetScope det(s);
  NR_CHECK_ARG(!det.on() || d->bwd_phase == 0, "mhsa_bwd: deterministic mode runs the backward in one call (bwd_phase = 0)");
  if (det.on()) {
    NR_CHECK_ARG(dtable == nullptr || d->table_rows > 0, "mhsa_bwd: deterministic mode needs table_rows in the descriptor");
    det.add(dw_qkv, (size_t)3 * N * d->d_model);
    det.add(db_qkv, (size_t)3 * N);
    if (dtable != nullptr) det.add(dtable, (size_t)d->table_rows * d->d_model);
    if ((rc = det.begin(true, false))) return rc;
  }
  const bool fork = want_dx && side_enabled() && M >= 65536 && !det.on() && d->bwd_phase == 0;
  hipStream_t s2 = s;
  if (fork && (rc = side_fork(s, &s2))) return rc;
  if (want_dx && ph_main) {
    hipStream_t s = s2;   // the input-gradient GEMM goes to the side stream
    NR_CHECK_ARG(w_qkv_t != nullptr && ldwt >= 3 * N, "mhsa_bwd: w_qkv_t [d_model, >=3N] needed for dx / dtable");
    RowSrc G = dense_rows(dqkv, 3 * N, 3 * N);
    if (d->src_kind == NR_SRC_GATHER) {
      NR_CHECK_ARG(dtable != nullptr && dx == nullptr, "mhsa_bwd: gather source takes dtable, not dx");
      NR_CHECK_ARG(d->d_model % 4 == 0, "mhsa_bwd: d_model=%d must be a multiple of 4", d->d_model);
      EpiArgs ep = store_epi(dtable, d->d_model, NR_F32, nullptr, 0);
      ep.ids = d->ids; ep.ids_stride = 1; ep.Dtrue = d->d_model; ep.drop = nr_make_drop(d->p_in, d->seed_in);
      if (d->row_ws != nullptr && d->dtype == NR_BF16 && M >= 4096) {
        // only rows with a non-padding token id reach the table gradient: compact them, GEMM over those alone
        if (!d->row_ws_ready && (rc = nr_launch_compact_rows(d->ids, 1, M, d->row_ws, s))) return rc;
        ep.row_count = d->row_ws; ep.row_idx = d->row_ws + W.live_idx; ep.row_ids = d->row_ws + W.live_ids;
        if (d->table_rows > 0 && !nr_opt(NR_OPT_NO_SCATTER_SORT)) {
          // ... in token-id order: occurrences of one word become neighbours, the scatter epilogue adds them up in
          // registers and issues one atomic row per run instead of one per occurrence (memory-side fp32 atomics run at
          // ~1.3 TB/s chip-wide; a MIND-shaped batch repeats each word ~9 times)
          if ((rc = nr_launch_sort_rows_by_id(d->row_ws, ep.row_idx, ep.row_ids, M, d->table_rows, d->row_ws + W.hist,
                                              d->row_ws + W.sort_idx, d->row_ws + W.sort_ids, s)))
            return rc;
          ep.row_idx = d->row_ws + W.sort_idx; ep.row_ids = d->row_ws + W.sort_ids;
        }
      }
      rc = nr_launch_gemm_nt(d->dtype, G, w_qkv_t, ldwt, M, d->d_model, 3 * N, EPI_SCATTER, ep, s);
    } else {
      NR_CHECK_ARG(dx != nullptr && dtable == nullptr, "mhsa_bwd: dense source takes dx, not dtable");
      NR_CHECK_ARG(d->p_in == 0.f, "mhsa_bwd: input dropout on a dense source is not supported");
      NR_CHECK_ARG(d->d_model % 4 == 0, "mhsa_bwd: d_model=%d must be a multiple of 4", d->d_model);
      EpiArgs ep = store_epi(dx, d->ldx, d->dtype, nullptr, 0);
      rc = nr_launch_gemm_nt(d->dtype, G, w_qkv_t, ldwt, M, d->d_model, 3 * N, EPI_STORE, ep, s);
    }
  }
  if (rc) return rc;
  // (after the table gradient: in a phased call the host's all-reduce of that gradient overlaps this GEMM)
  if (ph_dw) {
    if (slab_ws != nullptr) {
      if ((rc = nr_launch_gemm_tn_slabs(dqkv, 3 * N, d->x_rows, d->ld_rows, dw_qkv, d->d_model, db_qkv, M, 3 * N, Kp, 3 * N, d->d_model,
                                        slab_ws + d->n + 4, slab_ws + d->n, s, 0,
                                        W.tn_floats ? reinterpret_cast<float*>(d->row_ws + W.tn_scratch) : nullptr, W.tn_floats)))
        return rc;
    } else if ((rc = nr_launch_gemm_tn(d->dtype, dqkv, 3 * N, Xs, dw_qkv, d->d_model, db_qkv, M, 3 * N, Kp, 3 * N, d->d_model, s))) {
      return rc;
    }
  }
  if (fork) {
    const int rj = side_join(s);
    if (rc == NR_OK) rc = rj;
  }
  const int rd = det.end();
  return rc ? rc : rd;
}

// ---------------------------------------------------------------------------------------- Conv1d k=3
static int conv_rows(const nr_conv_desc* d, RowSrc* out) {
  NR_CHECK_ARG(d != nullptr, "conv1d: null descriptor");
  NR_CHECK_ARG(dtype_ok(d->dtype), "conv1d: bad dtype %d", d->dtype);
  const int ch = nr_chunk(d->dtype);
  NR_CHECK_ARG(d->n >= 0 && d->T >= 1 && d->D >= 1 && d->N >= 1 && d->Dp >= d->D && d->Dp % ch == 0,
               "conv1d: bad shape n=%d T=%d D=%d Dp=%d N=%d", d->n, d->T, d->D, d->Dp, d->N);
  NR_CHECK_ARG(d->N % ch == 0, "conv1d: N=%d must be a multiple of %d", d->N, ch);
  NR_CHECK_ARG(d->n == 0 || (d->table && d->ids && d->w_pack && d->bias && d->ids_stride >= 1), "conv1d: null operand");
  NR_CHECK_ARG((uint64_t)d->n * d->T * (uint64_t)d->D < 0xffffffffull, "conv1d: problem too large for 32-bit element counters");
  RowSrc s = dense_rows(d->table, d->Dp, d->D);
  s.kind = ROWS_IM2COL3;
  s.ids = d->ids; s.ids_stride = d->ids_stride; s.Tlen = d->T;
  s.drop = nr_make_drop(d->p_in, d->seed_in);
  *out = s;
  return NR_OK;
}

int nr_conv1d_k3_fwd(const nr_conv_desc* d, void* y, nr_stream_t stream) {
  RowSrc A;
  int rc = conv_rows(d, &A);
  if (rc) return rc;
  if (d->n == 0) return NR_OK;
  NR_CHECK_ARG(y != nullptr, "conv1d_fwd: null output");
  NR_DEVICE_GUARD(stream, y);
  EpiArgs ep = store_epi(y, d->N, d->dtype, d->bias, 0);
  // Row tiles made of unneeded titles only are not computed and their y rows stay UNWRITTEN -- but only on shapes whose
  // consumers contract live slabs (the pooling backward's dW1 = dpre^T . y): a dense contraction (small shapes) multiplies every
  // row by its zero gradient, and 0 x whatever the allocator left there is NaN (seen once the title table stopped being the
  // first big allocation: tests/test_gpu_model_parity.py, naml_mind_3view)
  ep.seq_nz = conv_slab_shape(d) ? d->seq_needed : nullptr; ep.L = d->T;
  hipStream_t s = (hipStream_t)stream;
  const int M = d->n * d->T, K = 3 * d->Dp;
  if (d->x_rows != nullptr) {
    // token rows (gather + dropout hashed once) with a zero row between titles: the im2col row of a token is then 3 * Dp
    // CONTIGUOUS elements of that buffer -- a dense operand with overlapping rows for the LDS-DMA GEMM, no 3x copy
    NR_CHECK_ARG(d->ld_rows == d->Dp, "conv1d_fwd: x_rows is [n * (T + 1) + 1, Dp] (ld_rows=%d, Dp=%d)", d->ld_rows, d->Dp);
    // with "needed" flags only the titles near a needed one are materialised: a 256-row GEMM tile / a 32-row slab spans
    // at most 256 / T + 2 titles
    // (only on shapes whose backward reads x_rows through the live-slab list -- a dense contraction would read every row)
    if ((rc = nr_launch_conv_rows(d->dtype, A, d->x_rows, d->n, s, conv_slab_shape(d) ? d->seq_needed : nullptr, 256 / d->T + 2))) return rc;
    A = dense_rows(d->x_rows, d->Dp, K);
    A.gap = d->T;
  }
  return nr_launch_gemm_nt(d->dtype, A, d->w_pack, K, M, d->N, K, EPI_STORE, ep, s);
}

int nr_conv1d_k3_bwd(const nr_conv_desc* d, const void* dy, float* dw_pack, float* db, nr_stream_t stream) {
  RowSrc A;
  int rc = conv_rows(d, &A);
  if (rc) return rc;
  if (d->n == 0) return NR_OK;
  NR_CHECK_ARG(dy && dw_pack && db, "conv1d_bwd: null operand");
  NR_DEVICE_GUARD(stream, dw_pack);
  NR_CHECK_ARG(d->bwd_ws == nullptr || d->bwd_ws_bytes >= conv_ws_elems(d->n, d->T) * sizeof(int32_t),
               "conv1d_bwd: bwd_ws holds %zu bytes, nr_conv_workspace_bytes() asks for %zu", d->bwd_ws_bytes,
               conv_ws_elems(d->n, d->T) * sizeof(int32_t));
  DetScope det((hipStream_t)stream);                 // flushes when the function returns
  det.add(dw_pack, (size_t)d->N * 3 * d->Dp);
  det.add(db, (size_t)d->N);
  if ((rc = det.begin(true, false))) return rc;
  if (d->x_rows != nullptr) {   // the rows the forward stored
    NR_CHECK_ARG(d->ld_rows == d->Dp, "conv1d_bwd: x_rows is [n * (T + 1) + 1, Dp] (ld_rows=%d, Dp=%d)", d->ld_rows, d->Dp);
    A = dense_rows(d->x_rows, d->Dp, 3 * d->Dp);
    A.gap = d->T;
    // titles with an exactly zero upstream gradient (masked history slots) add nothing to dW / db: live slabs only
    const bool no_slabs = nr_opt(NR_OPT_NO_SLABS) != 0;
    const int M = d->n * d->T;
    // x_rows written under "needed" flags are complete only where a live slab can reach: the slab path is then mandatory
    NR_CHECK_ARG(!(d->seq_needed != nullptr && conv_slab_shape(d)) || (d->bwd_ws != nullptr && (((uintptr_t)dy) & 15) == 0),
                 "conv1d_bwd: x_rows were materialised under seq_needed: bwd_ws (and a 16-byte aligned dy) are required");
    if (!no_slabs && d->bwd_ws != nullptr && d->dtype == NR_BF16 && M % 32 == 0 && d->T <= 32 && d->N % 8 == 0 &&
        (((uintptr_t)dy) & 15) == 0 && nr_gemm_tn_slabs_ok(d->N, d->ld_rows, M, d->N, 3 * d->Dp)) {
      hipStream_t s = (hipStream_t)stream;
      if (d->seq_nz != nullptr) {
        NR_CHECK_HIP(hipMemcpyAsync(d->bwd_ws, d->seq_nz, (size_t)d->n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
      } else if ((rc = nr_launch_title_flags(dy, d->n, d->T, d->N, d->bwd_ws, s))) {
        return rc;
      }
      if ((rc = nr_launch_live_slabs(d->bwd_ws, d->n, d->T, s))) return rc;
      return nr_launch_gemm_tn_slabs(dy, d->N, d->x_rows, d->Dp, dw_pack, 3 * d->Dp, db, M, d->N, 3 * d->Dp, d->N, 3 * d->Dp,
                                     d->bwd_ws + d->n + 4, d->bwd_ws + d->n, s, d->T);
    }
  }
  return nr_launch_gemm_tn(d->dtype, dy, d->N, A, dw_pack, 3 * d->Dp, db, d->n * d->T, d->N, 3 * d->Dp, d->N, 3 * d->Dp,
                           (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------- additive pooling
static int pool_check(const nr_pool_desc* d) {
  NR_CHECK_ARG(d != nullptr, "additive_pool: null descriptor");
  NR_CHECK_ARG(dtype_ok(d->dtype), "additive_pool: bad dtype %d", d->dtype);
  const int ch = nr_chunk(d->dtype);
  NR_CHECK_ARG(d->n >= 0 && d->L >= 1 && d->L <= 64 && d->N >= 1 && d->q >= 1, "additive_pool: bad shape n=%d L=%d N=%d q=%d", d->n, d->L,
               d->N, d->q);
  NR_CHECK_ARG(d->N % ch == 0 && d->q % ch == 0, "additive_pool: N=%d and q=%d must be multiples of %d for this dtype", d->N, d->q, ch);
  NR_CHECK_ARG(d->ldw1 >= d->N, "additive_pool: ldw1=%d < N=%d", d->ldw1, d->N);
  NR_CHECK_ARG(d->n == 0 || (d->x && d->w1 && d->b1 && d->w2 && d->b2), "additive_pool: null operand");
  return NR_OK;
}

int nr_additive_pool_fwd(const nr_pool_desc* d, void* e, float* alpha, float* out, int ld_out, nr_stream_t stream) {
  int rc = pool_check(d);
  if (rc) return rc;
  if (d->n == 0) return NR_OK;
  NR_CHECK_ARG(e && alpha && out && ld_out >= d->N, "additive_pool_fwd: null output / ld_out");
  NR_DEVICE_GUARD(stream, out);
  hipStream_t s = (hipStream_t)stream;
  // title-level shapes (L <= 32 tokens, N ~ 400, q <= 256, bf16): fc1 + tanh + fc2 + softmax + weighted sum in one pass over x
  if (nr_pool_fused_fwd_ok(d->dtype, d->n, d->L, d->N, d->q, d->ldw1) && ((((uintptr_t)d->x) | ((uintptr_t)d->w1) | ((uintptr_t)e)) & 15) == 0 &&
      d->N % 8 == 0 && d->mask == nullptr)
    return nr_launch_pool_fused_fwd(d->x, d->N, d->w1, d->ldw1, d->b1, d->w2, d->b2, d->mask, e, d->q, alpha, out, ld_out, d->n, d->L, d->N, d->q,
                                    d->seq_needed, s);
  RowSrc A = dense_rows(d->x, d->N, d->N);
  EpiArgs ep = store_epi(e, d->q, d->dtype, d->b1, 1);
  ep.seq_nz = d->seq_needed; ep.L = d->L;          // row tiles made of unneeded sequences only are not computed (e stays unwritten)
  if ((rc = nr_launch_gemm_nt(d->dtype, A, d->w1, d->ldw1, d->n * d->L, d->q, d->N, EPI_STORE, ep, s))) return rc;
  return nr_launch_pool_core_fwd(d->dtype, d->x, e, d->w2, d->b2, d->mask, alpha, out, ld_out, d->n, d->L, d->N, d->q, s, d->seq_needed);
}

int nr_additive_pool_bwd(const nr_pool_desc* d, const void* e, const float* alpha, const float* g, int ld_g, const void* w1_t,
                         int ldw1t, void* dpre, float* partial, float* dw1, float* db1, float* dw2, float* db2, void* dx,
                         nr_stream_t stream) {
  int rc = pool_check(d);
  if (rc) return rc;
  if (d->n == 0) return NR_OK;
  NR_CHECK_ARG(e && alpha && g && dpre && partial && dw1 && db1 && dw2 && db2, "additive_pool_bwd: null operand");
  NR_DEVICE_GUARD(stream, dpre);
  NR_CHECK_ARG(d->partial_bytes >= pool_ws_elems(d->n, d->L, d->q, d->N, d->dtype) * sizeof(float),
               "additive_pool_bwd: partial holds %zu bytes, nr_pool_workspace_bytes() asks for %zu", d->partial_bytes,
               pool_ws_elems(d->n, d->L, d->q, d->N, d->dtype) * sizeof(float));
  float* tn_scratch = partial + pool_ws_ints(d->n, d->L, d->q);
  const size_t tn_floats = pool_tn_floats(d->n, d->L, d->q, d->N, d->dtype);
  hipStream_t s = (hipStream_t)stream;
  const int M = d->n * d->L;
  DetScope det(s);                                   // dw1 / db1 come from the GEMM kernels, dw2 / db2 from the column sums
  det.add(dw1, (size_t)d->q * d->N);
  det.add(db1, (size_t)d->q);
  det.add(dw2, (size_t)d->q);
  det.add(db2, 1);
  if ((rc = det.begin(true, true))) return rc;
  // A sequence whose pooled gradient g is exactly zero (a history slot the user encoder masks out) has dA = <g, x> = 0,
  // so ds = 0 and its dpre rows are exact zeros: the core kernel only writes those zeros for it, and the att_fc1 weight
  // gradient contracts only the 32-row slabs that touch a sequence with g != 0.  The int scratch (n flags, count, M/32
  // slab ids) lives in the unused tail of `partial` (its first nr_pool_partial_rows(n) rows are taken).
  int32_t* ws = nullptr;
  if (pool_has_flags(d)) {
    ws = reinterpret_cast<int32_t*>(partial + pool_ws_used(d->n, d->q));   // the int scratch behind the partial rows
    if ((rc = nr_launch_row_flags_f32(g, ld_g, d->N, d->n, ws, s))) return rc;
  }
  // Without the flags derived from g (small / fp32 shapes) the caller's "output not needed" flags serve: such a sequence has a
  // zero pooled gradient by contract, and its e rows may never have been written by the forward.
  const int32_t* zero_flags = ws != nullptr ? ws : d->seq_needed;
  // title-level shapes with slab flags and an input gradient asked for: dA, ds, dpre, dX and the dw2 / db2 partial sums in ONE
  // kernel (x and e read once, dpre never read back for dX); the weight gradient dW1 = dpre^T . x below is unchanged
  // (not in deterministic mode: the kernel adds its threads' dw2 partial sums with LDS float atomics)
  if (ws != nullptr && dx != nullptr && w1_t != nullptr && !det.on() && nr_pool_fused_bwd_ok(d->dtype, d->n, d->L, d->N, d->q, ldw1t) && ld_g % 4 == 0 &&
      ((((uintptr_t)d->x) | ((uintptr_t)e) | ((uintptr_t)w1_t) | ((uintptr_t)dpre) | ((uintptr_t)dx) | ((uintptr_t)g)) & 15) == 0 && d->N % 8 == 0) {
    int rows = 0;
    if ((rc = nr_launch_pool_fused_bwd(d->x, d->N, e, d->q, alpha, g, ld_g, d->w2, w1_t, ldw1t, dpre, d->q, dx, d->N, partial,
                                       nr_pool_partial_rows(d->n), ws, d->n, d->L, d->N, d->q, s, &rows, d->dx_far_unwritten)))
      return rc;
    if ((rc = nr_launch_colsum_split(partial, rows, d->q + 1, d->q + 1, dw2, d->q, db2, s))) return rc;
    if ((rc = nr_launch_live_slabs(ws, d->n, d->L, s))) return rc;
    return nr_launch_gemm_tn_slabs(dpre, d->q, d->x, d->N, dw1, d->N, db1, M, d->q, d->N, d->q, d->N, ws + d->n + 4, ws + d->n, s, 0,
                                   tn_floats ? tn_scratch : nullptr, tn_floats);
  }
  if ((rc = nr_launch_pool_core_bwd(d->dtype, d->x, e, d->w2, alpha, g, ld_g, dpre, partial, dw2, db2, d->n, d->L, d->N, d->q, s, zero_flags)))
    return rc;
  RowSrc X = dense_rows(d->x, d->N, d->N);
  const bool fork = dx != nullptr && side_enabled() && M >= 65536 && !det.on();
  hipStream_t s2 = s;
  if (fork && (rc = side_fork(s, &s2))) return rc;
  if (ws != nullptr) {
    if ((rc = nr_launch_live_slabs(ws, d->n, d->L, s))) return rc;
    if ((rc = nr_launch_gemm_tn_slabs(dpre, d->q, d->x, d->N, dw1, d->N, db1, M, d->q, d->N, d->q, d->N, ws + d->n + 4, ws + d->n, s, 0,
                                      tn_floats ? tn_scratch : nullptr, tn_floats)))
      return rc;
  } else if ((rc = nr_launch_gemm_tn(d->dtype, dpre, d->q, X, dw1, d->N, db1, M, d->q, d->N, d->q, d->N, s))) {
    return rc;
  }
  if (dx != nullptr) {
    NR_CHECK_ARG(w1_t != nullptr && ldw1t >= d->q, "additive_pool_bwd: w1_t [N, >=q] needed for dx");
    RowSrc P = dense_rows(dpre, d->q, d->q);
    EpiArgs ep = store_epi(dx, d->N, d->dtype, nullptr, 0);
    ep.rowscale = alpha; ep.G = g; ep.ldg = ld_g; ep.L = d->L;
    ep.seq_nz = zero_flags;                          // tiles made of zero-gradient sequences only just write zeros
    rc = nr_launch_gemm_nt(d->dtype, P, w1_t, ldw1t, M, d->N, d->q, EPI_POOLBWD, ep, s2);
  }
  if (fork) {
    const int rj = side_join(s);
    if (rc == NR_OK) rc = rj;
  }
  return rc;
}

// ---------------------------------------------------------------------------------------- gather + Linear
static int linear_rows(const nr_linear_desc* d, RowSrc* out) {
  NR_CHECK_ARG(d != nullptr, "linear: null descriptor");
  NR_CHECK_ARG(dtype_ok(d->dtype), "linear: bad dtype %d", d->dtype);
  const int ch = nr_chunk(d->dtype);
  NR_CHECK_ARG(d->M >= 0 && d->K >= 1 && d->N >= 1, "linear: bad shape M=%d K=%d N=%d", d->M, d->K, d->N);
  NR_CHECK_ARG(d->ldx >= round_up(d->K, ch) && d->ldw >= round_up(d->K, ch), "linear: ldx=%d / ldw=%d must cover K=%d rounded up to %d",
               d->ldx, d->ldw, d->K, ch);
  NR_CHECK_ARG(d->M == 0 || (d->x && d->w), "linear: null operand");
  RowSrc s = dense_rows(d->x, d->ldx, d->K);
  if (d->src_kind == NR_SRC_GATHER) {
    NR_CHECK_ARG(d->ids != nullptr && d->ids_stride >= 1, "linear: gather source without ids");
    s.kind = ROWS_GATHER; s.ids = d->ids; s.ids_stride = d->ids_stride;
  } else {
    NR_CHECK_ARG(d->src_kind == NR_SRC_DENSE, "linear: bad src_kind %d", d->src_kind);
  }
  *out = s;
  return NR_OK;
}

int nr_linear_fwd(const nr_linear_desc* d, float* out, int ld_out, nr_stream_t stream) {
  RowSrc A;
  int rc = linear_rows(d, &A);
  if (rc) return rc;
  if (d->M == 0) return NR_OK;
  NR_CHECK_ARG(out != nullptr && ld_out >= d->N, "linear_fwd: null output / ld_out");
  NR_DEVICE_GUARD(stream, out);
  EpiArgs ep = store_epi(out, ld_out, NR_F32, d->bias, 0);
  return nr_launch_gemm_nt(d->dtype, A, d->w, d->ldw, d->M, d->N, round_up(d->K, nr_chunk(d->dtype)), EPI_STORE, ep, (hipStream_t)stream);
}

int nr_linear_bwd(const nr_linear_desc* d, const float* dout, int ld_dout, void* dout_ws, float* dw, float* db, float* dtable,
                  nr_stream_t stream) {
  RowSrc A;
  int rc = linear_rows(d, &A);
  if (rc) return rc;
  if (d->M == 0) return NR_OK;
  NR_CHECK_ARG(dout && dout_ws && dw && db, "linear_bwd: null operand");
  NR_DEVICE_GUARD(stream, dw);
  NR_CHECK_ARG(d->dout_ws_bytes >= nr_linear_workspace_bytes(d), "linear_bwd: dout_ws holds %zu bytes, nr_linear_workspace_bytes() asks for %zu",
               d->dout_ws_bytes, nr_linear_workspace_bytes(d));
  DetScope det((hipStream_t)stream);
  if (det.on()) {
    NR_CHECK_ARG(dtable == nullptr || d->table_rows > 0, "linear_bwd: deterministic mode needs table_rows in the descriptor");
    det.add(dw, (size_t)d->N * d->K);
    det.add(db, (size_t)d->N);
    if (dtable != nullptr) det.add(dtable, (size_t)d->table_rows * d->K);
    if ((rc = det.begin(true, false))) return rc;
  }
  hipStream_t s = (hipStream_t)stream;
  const int ch = nr_chunk(d->dtype), Nc = round_up(d->N, ch), Kp = round_up(d->K, ch);
  if ((rc = nr_launch_cast_rows(d->dtype, dout, ld_dout, dout_ws, Nc, d->M, d->N, s))) return rc;
  if ((rc = nr_launch_gemm_tn(d->dtype, dout_ws, Nc, A, dw, d->K, db, d->M, Nc, Kp, d->N, d->K, s))) return rc;
  if (dtable != nullptr) {
    NR_CHECK_ARG(d->src_kind == NR_SRC_GATHER, "linear_bwd: dtable needs a gather source");
    NR_CHECK_ARG(d->w_t != nullptr && d->ldwt >= Nc, "linear_bwd: w_t [K, >=N rounded up] needed for dtable");
    RowSrc G = dense_rows(dout_ws, Nc, Nc);
    EpiArgs ep = store_epi(dtable, d->K, NR_F32, nullptr, 0);
    ep.ids = d->ids; ep.ids_stride = d->ids_stride; ep.Dtrue = d->K;
    rc = nr_launch_gemm_nt(d->dtype, G, d->w_t, d->ldwt, d->M, d->K, Nc, EPI_SCATTER, ep, s);
  }
  return rc;
}

}  // extern "C"
