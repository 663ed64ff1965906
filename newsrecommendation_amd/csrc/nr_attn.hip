// Per-(sequence, head) scaled-dot-product attention core on a materialised Q|K|V buffer.
// Restates src/model/model_utils.py:47-54 (forward) and its analytic gradient, including the
// reference's non-standard softmax: exp without max-subtraction, key mask applied AFTER exp,
// denominator sum + 1e-8.  We evaluate the algebraically identical stable form
//     a_ij = exp(s_ij - m_i) mask_j / (sum_j exp(s_ij - m_i) mask_j + 1e-8 exp(-m_i)).
//
// One workgroup = one sequence (L <= 64 tokens) x a group of HG heads; K and V of the group
// are staged in LDS as fp32; thread (i, head) owns query row i.  Backward is two phases
// (row-wise: dQ + row statistics; column-wise: dK, dV) so no atomics are needed and the
// result is deterministic.
#include <stdlib.h>

#include "nr_common.h"

namespace {

constexpr int ATT_THREADS = 256;

struct AttnArgs {
  const void* qkv;    // [n*L, 3N]
  const float* mask;  // [n, L] or null
  void* y;            // fwd out [n*L, N]
  const void* dy;     // bwd in  [n*L, N]
  void* dqkv;         // bwd out [n*L, 3N]
  int n, L, heads, N, HG;
  float scale;
  DropCfg drop;       // output dropout (element index m*N + c)
};

template <int DH>
__device__ __forceinline__ float dot_lds(const float (&q)[DH], const float* __restrict__ r) {
  float d = 0.f;
#pragma unroll
  for (int c = 0; c < DH; ++c) d = fmaf(q[c], r[c], d);
  return d;
}

template <typename T, int DH>
__global__ __launch_bounds__(ATT_THREADS) void attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int seq = blockIdx.x, h0 = blockIdx.y * a.HG;
  const int hg = min(a.HG, a.heads - h0);
  const int W = hg * DH;  // staged width
  float* sK = reinterpret_cast<float*>(smem);
  float* sV = sK + a.L * a.HG * DH;
  float* sMask = sV + a.L * a.HG * DH;
  const T* qkv = reinterpret_cast<const T*>(a.qkv);
  const int N = a.N, L = a.L, tid = threadIdx.x;
  const size_t row0 = (size_t)seq * L;

  for (int idx = tid; idx < L * W; idx += ATT_THREADS) {
    const int j = idx / W, r = idx - j * W;
    const T* src = qkv + (row0 + j) * 3 * N + h0 * DH + r;
    sK[j * W + r] = (float)src[N];
    sV[j * W + r] = (float)src[2 * N];
  }
  for (int j = tid; j < L; j += ATT_THREADS) sMask[j] = a.mask ? a.mask[(size_t)seq * L + j] : 1.f;
  __syncthreads();

  for (int p = tid; p < L * hg; p += ATT_THREADS) {
    const int hh = p / L, i = p - hh * L;
    const int h = h0 + hh;
    float q[DH];
    const T* qp = qkv + (row0 + i) * 3 * N + h * DH;
#pragma unroll
    for (int c = 0; c < DH; ++c) q[c] = (float)qp[c];
    // pass 1: row max; pass 2: weights and context (dot products recomputed, nothing spilled)
    float m = -INFINITY;
#pragma unroll 2
    for (int j = 0; j < L; ++j) m = fmaxf(m, dot_lds<DH>(q, sK + j * W + hh * DH) * a.scale);
    float sum = 0.f;
    float ctx[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) ctx[c] = 0.f;
#pragma unroll 2
    for (int j = 0; j < L; ++j) {
      const float e = __expf(dot_lds<DH>(q, sK + j * W + hh * DH) * a.scale - m) * sMask[j];
      sum += e;
      const float* vr = sV + j * W + hh * DH;
#pragma unroll
      for (int c = 0; c < DH; ++c) ctx[c] = fmaf(e, vr[c], ctx[c]);
    }
    const float inv = 1.f / (sum + 1e-8f * __expf(-m));
    T* yp = reinterpret_cast<T*>(a.y) + (row0 + i) * N + h * DH;
    const uint32_t e0 = (uint32_t)(row0 + i) * (uint32_t)N + (uint32_t)(h * DH);
#pragma unroll
    for (int c = 0; c < DH; ++c) {
      float v = ctx[c] * inv;
      if (a.drop.thresh) v = nr_keep(a.drop.key, e0 + c, a.drop.thresh) ? v * a.drop.scale : 0.f;
      yp[c] = (T)v;
    }
  }
}

template <typename T, int DH>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int seq = blockIdx.x, h0 = blockIdx.y * a.HG;
  const int hg = min(a.HG, a.heads - h0);
  const int W = hg * DH;
  const int LW = a.L * a.HG * DH;
  float* sQ = reinterpret_cast<float*>(smem);
  float* sK = sQ + LW;
  float* sV = sK + LW;
  float* sG = sV + LW;                 // d(ctx) after undoing the output dropout
  float* sMask = sG + LW;              // [L]
  float* sM = sMask + 64;              // [HG][64] row max
  float* sInv = sM + 64 * a.HG;        // [HG][64] 1/(sum + eps')
  float* sRd = sInv + 64 * a.HG;       // [HG][64] sum_j a_ij dA_ij
  const T* qkv = reinterpret_cast<const T*>(a.qkv);
  const T* dy = reinterpret_cast<const T*>(a.dy);
  T* dqkv = reinterpret_cast<T*>(a.dqkv);
  const int N = a.N, L = a.L, tid = threadIdx.x;
  const size_t row0 = (size_t)seq * L;

  for (int idx = tid; idx < L * W; idx += ATT_THREADS) {
    const int j = idx / W, r = idx - j * W;
    const T* src = qkv + (row0 + j) * 3 * N + h0 * DH + r;
    sQ[j * W + r] = (float)src[0];
    sK[j * W + r] = (float)src[N];
    sV[j * W + r] = (float)src[2 * N];
    float g = (float)dy[(row0 + j) * N + h0 * DH + r];
    if (a.drop.thresh) {
      const uint32_t e = (uint32_t)(row0 + j) * (uint32_t)N + (uint32_t)(h0 * DH + r);
      g = nr_keep(a.drop.key, e, a.drop.thresh) ? g * a.drop.scale : 0.f;
    }
    sG[j * W + r] = g;
  }
  for (int j = tid; j < L; j += ATT_THREADS) sMask[j] = a.mask ? a.mask[(size_t)seq * L + j] : 1.f;
  __syncthreads();

  // phase 1: query rows -> dQ and the row statistics (m_i, 1/Z_i, sum_j a_ij dA_ij)
  for (int p = tid; p < L * hg; p += ATT_THREADS) {
    const int hh = p / L, i = p - hh * L;
    float q[DH], g[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) {
      q[c] = sQ[i * W + hh * DH + c];
      g[c] = sG[i * W + hh * DH + c];
    }
    float m = -INFINITY;
#pragma unroll 2
    for (int j = 0; j < L; ++j) m = fmaxf(m, dot_lds<DH>(q, sK + j * W + hh * DH) * a.scale);
    float sum = 0.f, rdu = 0.f;  // rdu = sum_j e_ij dA_ij (unnormalised)
#pragma unroll 2
    for (int j = 0; j < L; ++j) {
      const float e = __expf(dot_lds<DH>(q, sK + j * W + hh * DH) * a.scale - m) * sMask[j];
      sum += e;
      rdu = fmaf(e, dot_lds<DH>(g, sV + j * W + hh * DH), rdu);
    }
    const float inv = 1.f / (sum + 1e-8f * __expf(-m));
    const float rd = rdu * inv;
    float dq[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) dq[c] = 0.f;
#pragma unroll 2
    for (int j = 0; j < L; ++j) {
      const float* kr = sK + j * W + hh * DH;
      const float aij = __expf(dot_lds<DH>(q, kr) * a.scale - m) * sMask[j] * inv;
      const float ds = aij * (dot_lds<DH>(g, sV + j * W + hh * DH) - rd);
#pragma unroll
      for (int c = 0; c < DH; ++c) dq[c] = fmaf(ds, kr[c], dq[c]);
    }
    T* op = dqkv + (row0 + i) * 3 * N + (h0 + hh) * DH;
#pragma unroll
    for (int c = 0; c < DH; ++c) op[c] = (T)(dq[c] * a.scale);
    sM[hh * 64 + i] = m;
    sInv[hh * 64 + i] = inv;
    sRd[hh * 64 + i] = rd;
  }
  __syncthreads();

  // phase 2: key / value columns -> dK, dV
  for (int p = tid; p < L * hg; p += ATT_THREADS) {
    const int hh = p / L, j = p - hh * L;
    float k[DH], v[DH], dk[DH], dv[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) {
      k[c] = sK[j * W + hh * DH + c];
      v[c] = sV[j * W + hh * DH + c];
      dk[c] = 0.f;
      dv[c] = 0.f;
    }
    const float mj = sMask[j];
#pragma unroll 2
    for (int i = 0; i < L; ++i) {
      const float* qr = sQ + i * W + hh * DH;
      const float* gr = sG + i * W + hh * DH;
      const float aij = __expf(dot_lds<DH>(k, qr) * a.scale - sM[hh * 64 + i]) * mj * sInv[hh * 64 + i];
      const float ds = aij * (dot_lds<DH>(v, gr) - sRd[hh * 64 + i]) * a.scale;
#pragma unroll
      for (int c = 0; c < DH; ++c) {
        dk[c] = fmaf(ds, qr[c], dk[c]);
        dv[c] = fmaf(aij, gr[c], dv[c]);
      }
    }
    T* op = dqkv + (row0 + j) * 3 * N + (h0 + hh) * DH;
#pragma unroll
    for (int c = 0; c < DH; ++c) {
      op[N + c] = (T)dk[c];
      op[2 * N + c] = (T)dv[c];
    }
  }
}

template <typename T, int DH>
int launch_attn(bool bwd, const AttnArgs& a, hipStream_t stream) {
  const int groups = (a.heads + a.HG - 1) / a.HG;
  const size_t lw = (size_t)a.L * a.HG * DH * sizeof(float);
  if (!bwd) {
    const size_t smem = 2 * lw + 64 * sizeof(float);
    auto k = attn_fwd_kernel<T, DH>;
    NR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(k, dim3(a.n, groups), dim3(ATT_THREADS), smem, stream, a);
  } else {
    const size_t smem = 4 * lw + (64 + 3 * 64 * a.HG) * sizeof(float);
    auto k = attn_bwd_kernel<T, DH>;
    NR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(k, dim3(a.n, groups), dim3(ATT_THREADS), smem, stream, a);
  }
  NR_CHECK_LAUNCH();
  return NR_OK;
}

template <typename T>
int launch_attn_d(bool bwd, int DH, const AttnArgs& a, hipStream_t s) {
  switch (DH) {
    case 4: return launch_attn<T, 4>(bwd, a, s);
    case 8: return launch_attn<T, 8>(bwd, a, s);
    case 16: return launch_attn<T, 16>(bwd, a, s);
    case 20: return launch_attn<T, 20>(bwd, a, s);
    case 32: return launch_attn<T, 32>(bwd, a, s);
  }
  nr_set_error("attention: d_head=%d not instantiated (4, 8, 16, 20, 32)", DH);
  return NR_ERR_ARG;
}

}  // namespace

bool nr_attn_mfma_supported(int L, int d_head);
int nr_launch_attn_mfma(bool bwd, int dtype, const void* qkv, const float* mask, void* y, const void* dy, void* dqkv, int n,
                        int L, int heads, int d_head, const DropCfg& drop, hipStream_t stream, const uint32_t* tmask,
                        const float* bias, const int32_t* seq_list, const int32_t* seq_count, const int32_t* needed);

// qkv [n*L, 3N] -> y [n*L, N] (fwd) ; (qkv, dy) -> dqkv (bwd)
int nr_launch_attn(bool bwd, int dtype, const void* qkv, const float* mask, void* y, const void* dy, void* dqkv, int n, int L,
                   int heads, int d_head, const DropCfg& drop, hipStream_t stream, const uint32_t* tmask, const float* bias,
                   const int32_t* seq_list, const int32_t* seq_count, const int32_t* needed) {
  NR_CHECK_ARG(L >= 1 && L <= 64, "attention: L=%d must be in [1, 64]", L);
  NR_CHECK_ARG(n >= 1 && heads >= 1, "attention: empty problem");
  // L <= 32: one wave per (sequence, head) on the matrix cores; longer sequences: LDS/VALU kernels below.
  const bool force_valu = nr_opt(NR_OPT_ATTN_VALU) != 0;
  if (!force_valu && nr_attn_mfma_supported(L, d_head)) {
    const int rc = nr_launch_attn_mfma(bwd, dtype, qkv, mask, y, dy, dqkv, n, L, heads, d_head, drop, stream, tmask, bias, seq_list,
                                       seq_count, needed);
    if (rc >= 0) return rc;   // -1: this shape / dtype has no MFMA kernel, use the LDS/VALU kernels below
  }
  NR_CHECK_ARG(tmask == nullptr, "attention: padding-token substitution is not available on the LDS/VALU kernels");
  AttnArgs a;
  a.qkv = qkv; a.mask = mask; a.y = y; a.dy = dy; a.dqkv = dqkv;
  a.n = n; a.L = L; a.heads = heads; a.N = heads * d_head;
  int hg = ATT_THREADS / L;
  if (hg < 1) hg = 1;
  if (hg > heads) hg = heads;
  // keep the backward's 4 staged operands within ~96 KB of LDS
  while (hg > 1 && (size_t)4 * L * hg * d_head * sizeof(float) > 96 * 1024) --hg;
  a.HG = hg;
  a.scale = 1.0f / sqrtf((float)d_head);
  a.drop = drop;
  NrProfScope ps(stream, "attn_%s[%s,n=%d,L=%d,h=%d,d=%d]", bwd ? "bwd" : "fwd", dtype == NR_BF16 ? "bf16" : "f32", n, L, heads, d_head);
  if (dtype == NR_BF16) return launch_attn_d<bf16_t>(bwd, d_head, a, stream);
  return launch_attn_d<float>(bwd, d_head, a, stream);
}
