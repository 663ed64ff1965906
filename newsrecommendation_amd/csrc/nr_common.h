// Shared device/host helpers for libnrhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/nrhip.h"

typedef __bf16 bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define NR_WAVE 64

// ---- error plumbing (host) -------------------------------------------------------------
void nr_set_error(const char* fmt, ...);
#define NR_CHECK_ARG(cond, ...)                    \
  do {                                             \
    if (!(cond)) {                                 \
      nr_set_error(__VA_ARGS__);                   \
      return NR_ERR_ARG;                           \
    }                                              \
  } while (0)
#define NR_CHECK_HIP(expr)                                                        \
  do {                                                                            \
    hipError_t e__ = (expr);                                                      \
    if (e__ != hipSuccess) {                                                      \
      nr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return NR_ERR_HIP;                                                          \
    }                                                                             \
  } while (0)
#define NR_CHECK_LAUNCH() NR_CHECK_HIP(hipGetLastError())

// ---- device guard -------------------------------------------------------------------------------
// Every compute entry point runs on the device that owns `stream` (or, for the NULL stream, the device that owns the
// given pointer): forward is called from the rank's main thread, backward from the autograd engine thread, and
// hipFuncSetAttribute / hipEventCreate / kernel launches act on the calling thread's CURRENT device.  The guard
// switches to the right device for the duration of the call and restores the previous one.
struct DeviceGuard {
  int prev = -1, want = -1;
  int rc = NR_OK;
  DeviceGuard(hipStream_t s, const void* p) {
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; return; }
    if (s != nullptr) {
      hipDevice_t d;
      if (hipStreamGetDevice(s, &d) == hipSuccess) want = (int)d;
    } else if (p != nullptr) {
      hipPointerAttribute_t at;
      if (hipPointerGetAttributes(&at, p) == hipSuccess) want = at.device;
      else (void)hipGetLastError();              // not a device pointer the runtime knows: leave the device alone
    }
    if (want >= 0 && want != prev) {
      if (hipSetDevice(want) != hipSuccess) { nr_set_error("cannot switch to device %d", want); rc = NR_ERR_HIP; }
    } else {
      prev = -1;                                 // nothing to restore
    }
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
#define NR_DEVICE_GUARD(stream, p)              \
  DeviceGuard nr_guard__((hipStream_t)(stream), (p)); \
  if (nr_guard__.rc) return nr_guard__.rc

// ---- library options (host) -----------------------------------------------------------------
// One table of integer switches instead of getenv() calls scattered over the launchers.  Defaults are the
// production configuration; each entry can be preset from the environment variable NR_<NAME> (read ONCE, when the
// first option is looked up) or changed at run time through nr_set_option() (include/nrhip.h) -- the tests use the
// latter to compare a path with its own switched-off variant inside one process.
enum NrOpt {
  NR_OPT_NO_SLABS = 0,   // 1: weight-gradient GEMMs contract every row (no live-slab lists)
  NR_OPT_NO_ATTN_SKIP,   // 1: the attention backward walks every sequence (no device-side sequence list)
  NR_OPT_SIDE_STREAM,    // 1: weight- and input-gradient GEMMs of one backward run on two streams
  NR_OPT_ATTN_OLD,       // 1: per-wave staging attention kernels instead of the bf16 panel kernels
  NR_OPT_ATTN_VALU,      // 1: LDS/VALU attention kernels
  NR_OPT_NO_PAD_SUB,     // 1: no bias substitution for all-padding sequences
  NR_OPT_NO_FUSED_FWD,   // 1: never use the fused title-level forward kernels
  NR_OPT_NO_TN3,         // 1: weight gradients through tn2 / v1 kernels
  NR_OPT_TN_V1,          // 1: first-generation TN kernel
  NR_OPT_TN3_ROUNDS,     // >0: force the number of row-split rounds of tn3
  NR_OPT_TN3_WK,         // 2 / 4: force the tn3 tile family
  NR_OPT_TN3_NI,         // 4 / 8: force the tn3 n-tile
  NR_OPT_NT_NOWIDE,      // 1: no "wide" NT kernel
  NR_OPT_NT_NODMA,       // 1: no LDS-DMA NT kernel
  NR_OPT_DMA_MIN_K,      // smallest K that takes the LDS-DMA NT kernel (default 192)
  NR_OPT_DMA_WM2_ALL,    // 1: 128-row DMA tiles for every N
  NR_OPT_ATTN_PRED,      // 1: predicated (pre-"FULL") memory instructions in the bf16 panel attention kernels
  NR_OPT_ATTN_GENERIC,   // 1: no shape-specialised (L=30, 20 heads of 20) instantiation of the panel attention kernels
  NR_OPT_NO_ROW_SUB,     // 1: the projection writes the bias into padding rows (bias_rows) instead of per-row substitution
  NR_OPT_ATTN_BWD_OCC4,  // 1: the specialised attention backward built for 4 waves per SIMD (128 VGPRs, a few spills) instead of 3
  NR_OPT_NT_ABLATE,      // measurement only (results are WRONG): tiled LDS-DMA NT kernel without 1: output stores, 2: MFMAs, 4: operand DMA, 8: epilogue; 64: phase stamps (nr_debug_nt_trace)
  NR_OPT_NT_WREG,        // 1 (default): skinny-K bf16 NT GEMMs with the weights held in registers (persistent, LDS ring of activation rows); 3: the same with 16-row instead of 32-row stages for the QKV shape; 0: tile kernels
  NR_OPT_NO_SCATTER_SORT, // 1: the table-gradient GEMM walks the live rows in batch order instead of token-id order
  NR_OPT_TN3_MIN_M,      // smallest row count that takes the LDS-DMA weight-gradient kernel (default 16384)
  NR_OPT_NO_COMPACT_ROWS, // 1: x_rows / dqkv of the news-level training path keep one row per token (no compact row storage)
  NR_OPT_NO_POOL_FUSED,  // 1: additive pooling forward as fc1 GEMM + pool_core_fwd instead of the fused kernel
  NR_OPT_ATTN_BWD_GRID,  // >0: workgroups of the compact-row attention backward (default 4096 = 16 per CU, grid-stride over the walk)
  NR_OPT_TN3_ATOMIC,     // 1: the LDS-DMA weight-gradient kernel adds its tiles into dW with fp32 atomics even when the caller brought scratch
  NR_OPT_TN3_ABLATE,     // measurement only (results are WRONG): 1 = the LDS-DMA weight-gradient kernel skips its fp32 atomic epilogue
  NR_OPT_POOL_ABLATE,    // measurement only (results are WRONG): fused pooling forward without 1: weighted sum, 2: softmax, 4: out reduction, 8: MFMAs, 16: tanh / logit epilogue
  NR_OPT_COUNT
};
int nr_opt(int which);

// ---- per-kernel timing scope (host); no-op unless nr_prof_enable(1) ----------------------
struct NrProfScope {
  int idx;
  hipStream_t stream;
  NrProfScope(const char* label, hipStream_t s);
  NrProfScope(hipStream_t s, const char* fmt, ...);
  ~NrProfScope();
};
extern bool g_nr_prof_on;

static inline int nr_elt_size(int dtype) { return dtype == NR_BF16 ? 2 : 4; }
// K granule of one 16-byte chunk in elements
static inline int nr_chunk(int dtype) { return dtype == NR_BF16 ? 8 : 4; }

// ---- counter-based dropout RNG ---------------------------------------------------------
// keep(seed, idx) is a pure function of (seed, element counter): forward and backward
// regenerate the same Bernoulli draw without storing a mask.
__host__ __device__ __forceinline__ uint32_t nr_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t nr_drop_key(uint32_t seed) { return nr_mix32(seed * 0x9e3779b9u + 0x7f4a7c15u); }
// One 32-bit hash serves TWO consecutive elements (16 bits each): the Bernoulli threshold has 2^-16 resolution
// (p = 0.2 -> 13107/65536) and the kernels, which handle 4 or 8 consecutive elements per lane, hash half as often.
__host__ __device__ __forceinline__ uint32_t nr_drop_thresh(float p) {
  double t = (double)p * 65536.0 + 0.5;
  return t >= 65535.0 ? 65535u : (uint32_t)t;
}
__host__ __device__ __forceinline__ uint32_t nr_pair_hash(uint32_t key, uint32_t idx) { return nr_mix32((idx >> 1) ^ key); }
// returns 1 if element idx is kept
__host__ __device__ __forceinline__ bool nr_keep(uint32_t key, uint32_t idx, uint32_t thresh) {
  const uint32_t h = nr_pair_hash(key, idx);
  return ((idx & 1u) ? (h >> 16) : (h & 0xffffu)) >= thresh;
}
// keep bits of the 4 consecutive elements e0 .. e0+3 (bit e set = kept); e0 must be EVEN
__host__ __device__ __forceinline__ uint32_t nr_keep4(uint32_t key, uint32_t e0, uint32_t thresh) {
  const uint32_t h0 = nr_mix32((e0 >> 1) ^ key), h1 = nr_mix32(((e0 >> 1) + 1u) ^ key);
  return ((h0 & 0xffffu) >= thresh ? 1u : 0u) | ((h0 >> 16) >= thresh ? 2u : 0u) | ((h1 & 0xffffu) >= thresh ? 4u : 0u) |
         ((h1 >> 16) >= thresh ? 8u : 0u);
}

struct DropCfg {
  uint32_t key;     // nr_drop_key(seed)
  uint32_t thresh;  // drop if the element's 16 hash bits < thresh ; 0 = dropout off
  float scale;      // 1/(1-p)
};
static inline DropCfg nr_make_drop(float p, uint32_t seed) {
  DropCfg d;
  d.key = nr_drop_key(seed);
  d.thresh = p > 0.f ? nr_drop_thresh(p) : 0u;
  d.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  return d;
}

// ---- order-independent accumulation (deterministic mode, nr_set_deterministic) ---------------------------
// Gradient outputs that several workgroups add into (dW / db / dtable / dpad) are accumulated with fp32 atomics: the
// result depends on the arrival order in the last bits.  In deterministic mode every contribution is rounded to 2^-36
// fixed point and added with a 64-bit INTEGER atomic into a shadow buffer -- integer addition is associative, so the sum
// is the same whatever the order -- and a flush pass adds the shadow (converted back) onto the fp32 output and zeroes it.
// The ranges of the outputs of the call in flight live in a per-translation-unit device table that the entry point sets
// (stream ordered) before its kernels and clears after them; an empty table (the default) means plain fp32 atomics.
#define NR_FIX_RANGES 4
struct NrFixTable {
  int count;
  int pad;
  const float* base[NR_FIX_RANGES];
  long long* fix[NR_FIX_RANGES];
  unsigned long long n[NR_FIX_RANGES];
};
#ifdef __HIPCC__
static __device__ NrFixTable g_nr_fix;      // one copy per translation unit (no relocatable device code)
// `det` = nr_fix_on(), read ONCE per kernel (a per-element table lookup in the default mode cost the scatter epilogue 30 %)
__device__ __forceinline__ bool nr_fix_on() { return g_nr_fix.count != 0; }
__device__ __forceinline__ void nr_accum(float* p, float x, bool det) {
  if (det) {
    const int cnt = g_nr_fix.count;
    for (int i = 0; i < cnt; ++i) {
      const unsigned long long idx = (unsigned long long)(p - g_nr_fix.base[i]);
      if (p >= g_nr_fix.base[i] && idx < g_nr_fix.n[i]) {
        atomicAdd(reinterpret_cast<unsigned long long*>(g_nr_fix.fix[i] + idx), (unsigned long long)__float2ll_rn(x * 68719476736.0f));
        return;
      }
    }
  }
  atomicAdd(p, x);
}
#endif
int nr_fix_set_gemm(const NrFixTable* t, hipStream_t s);   // nr_gemm.hip's copy of the table
int nr_fix_set_pool(const NrFixTable* t, hipStream_t s);   // nr_pool.hip's copy
int nr_fix_flush(const NrFixTable* t, hipStream_t s);      // out[i] += fix[i] * 2^-36 ; fix[i] = 0   for every range
// One accumulation scope of an entry point outside nr_api.hip (which has its own RAII form): returns 0 when the mode is
// off; otherwise registers up to two outputs, sets the tables, and nr_det_close flushes and clears them.
int nr_det_open(hipStream_t s, float* base0, size_t n0, float* base1, size_t n1, bool gemm, bool pool, int* rc);
int nr_det_close(int handle);

#ifdef __HIPCC__
// pieces of nr_accum for callers that sum several contributions before the atomic: the sum must then be taken in fixed
// point too (integer addition is associative, fp32 addition is not)
__device__ __forceinline__ long long nr_to_fix(float x) { return __float2ll_rn(x * 68719476736.0f); }
__device__ __forceinline__ void nr_accum_fix(float* p, long long q) {       // deterministic mode only
  const int cnt = g_nr_fix.count;
  for (int i = 0; i < cnt; ++i) {
    const unsigned long long idx = (unsigned long long)(p - g_nr_fix.base[i]);
    if (p >= g_nr_fix.base[i] && idx < g_nr_fix.n[i]) {
      atomicAdd(reinterpret_cast<unsigned long long*>(g_nr_fix.fix[i] + idx), (unsigned long long)q);
      return;
    }
  }
  atomicAdd(p, (float)q * (1.0f / 68719476736.0f));     // not a registered output (does not happen: DetScope lists them all)
}
#endif
// ---- small device helpers ----------------------------------------------------------------
template <typename T> struct EltTraits;
template <> struct EltTraits<float> { static constexpr int CH = 4; static constexpr int DT = NR_F32; };
template <> struct EltTraits<bf16_t> { static constexpr int CH = 8; static constexpr int DT = NR_BF16; };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v) { return (T)v; }
