// MFMA attention core for sequences of L <= 32 tokens (the title level: 98 % of the attention work).
//
// One wave owns one (sequence, head) pair: a 32 x 32 score tile on v_mfma_f32_32x32x16_bf16
// (or v_mfma_f32_32x32x2_f32 in the exact-fp32 mode).  The wave stages its head's Q, K, V (and
// d(ctx)) rows in a private LDS region, zero padded to 32 x 32, in two images:
//   row-major  [token][c]   -> operands of products that contract over the head dim c
//   transposed [c][token]   -> B operands of products that contract over tokens
// Products contracting over tokens take their A operand straight from the accumulators of the
// previous product (column on the lane, rows in the 16 registers; the k order inside a step is
// the accumulator's row order, and the transposed LDS image is read in the same order).
// Scores are computed in BOTH orientations in the backward pass so that the softmax statistics
// (per query row) are lane-local and dK / dV (sums over query rows) need no cross-lane traffic:
//   S^T = K Q^T, dP^T = V G^T (lane = query i)  -> m_i, 1/Z_i, rd_i, dQ = dS K
//   S   = Q K^T, dP   = G V^T (lane = key j)    -> dK = dS^T Q, dV = P^T G
// Softmax is the reference's (src/model/model_utils.py:47-53): exp, key mask after exp, denominator
// sum + 1e-8, evaluated in the stable form with the row max factored out.
#include <stdlib.h>
#include <type_traits>

#include "nr_common.h"

namespace {

constexpr int AW = 4;  // waves per workgroup

template <typename T> struct AT;
template <> struct AT<bf16_t> { static constexpr int SW = 40; };  // LDS row stride (elements): 80 B
template <> struct AT<float> { static constexpr int SW = 36; };   // 144 B

struct AttnMArgs {
  const void* qkv;    // [n*L, 3N]
  const float* mask;  // [n, L] or null
  void* y;            // fwd out [n*L, N]
  const void* dy;     // bwd in
  void* dqkv;         // bwd out [n*L, 3N]
  int n, L, heads, d, N;
  int vec;            // 1: d % 4 == 0 and all head slices are 4-element aligned -> vector staging / stores
  float scale;
  DropCfg drop;
  const int32_t* seq_list;   // optional (backward): the kernel walks seq_list[0 .. *seq_count) instead of 0 .. n-1 -- sequences left
  const int32_t* seq_count;  //   out have exactly zero dQ|dK|dV rows that nothing downstream reads
  const uint32_t* tmask;  // optional [n]: live-token bit mask of each sequence.  0 = padding tokens only: every Q|K|V row
  const float* bias;      //   of that sequence is this bias [3N]; its qkv rows are never written and never read
  const int32_t* needed;  // optional (forward, bf16 panel kernel) [n]: 0 = nobody uses this sequence's output: zeros are stored
  const int32_t* ids;     // optional (forward, bf16 panel kernel) [n*L]: row m of Q|K|V is qkv[ids[m]] -- `qkv` is then a
                          //   per-token-id table of projections (eval mode: W x + b depends on the token id only)
  // Compact row storage (backward, bf16 panel kernel with per-row padding substitution, L <= 31):
  const int32_t* pos;     // [n*L]: dQ|dK|dV of token row m is stored at dqkv row pos[m]; pos[m] < 0 (a padding token: nothing
                          //   downstream reads its gradient row) is not stored at all
  void* dump;             // 3N elements of scratch: where the (unpredicated) stores of an all-padding sequence land
  float* db;              // [3N] fp32, ACCUMULATED: column sums of dQ | dK | dV over every row of every sequence walked -- the bias
                          //   gradient, produced here because the padding rows that carry part of it are no longer stored
  const int32_t* nzf;     // optional [n]: 0 = this sequence's dy is zero BY CONTRACT and its rows may be unwritten memory: the kernel
                          //   puts zeros into the G image instead of what it loaded (nr_mhsa_desc.dy_far_unwritten)
};

__device__ __forceinline__ int rowof(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// acc[row a][col b] += sum_c A[a][c] * B[b][c]; A, B row-major LDS images [32][SW]
__device__ __forceinline__ void mm_rr(f32x16& acc, const bf16_t* A, const bf16_t* B, int lane) {
  constexpr int SW = AT<bf16_t>::SW;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(A + r * SW + 16 * s + 8 * h);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(B + r * SW + 16 * s + 8 * h);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
}
__device__ __forceinline__ void mm_rr(f32x16& acc, const float* A, const float* B, int lane) {
  constexpr int SW = AT<float>::SW;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int s = 0; s < 16; ++s)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * SW + 2 * s + h], B[r * SW + 2 * s + h], acc, 0, 0, 0);
}

// acc[row i][col c] += sum_j X[j][i] * Bt[c][j]; X = accumulator tile (col i on the lane, rows j in
// the registers), Bt = transposed LDS image [32 c][SW] with the token index contiguous.
__device__ __forceinline__ void mm_xt(f32x16& acc, const f32x16& x, const bf16_t* Bt, int lane) {
  constexpr int SW = AT<bf16_t>::SW;
  const int c = lane & 31, h = lane >> 5;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 a;
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = (bf16_t)x[8 * s + e];
    const bf16_t* bp = Bt + c * SW + 16 * s + 4 * h;
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(bp);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(bp + 8);
    const bf16x8 b = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
}
__device__ __forceinline__ void mm_xt(f32x16& acc, const f32x16& x, const float* Bt, int lane) {
  constexpr int SW = AT<float>::SW;
  const int c = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x[t], Bt[c * SW + rowof(t, h)], acc, 0, 0, 0);
}

// Stage one [L, d] head slice (row stride ld in global memory) as zero padded LDS images.
// lane = (row r = lane & 31, part = lane >> 5); part p owns columns [16p, 16p + 16).
template <typename T, bool ROWMAJOR, bool TRANSPOSED, bool DROP>
__device__ __forceinline__ void stage_head(const T* __restrict__ src, size_t ld, int L, int d, T* sR, T* sT, int lane,
                                           const DropCfg& drop, uint32_t eidx0, uint32_t erow) {
  constexpr int SW = AT<T>::SW;
  const int r = lane & 31, c0 = (lane >> 5) * 16;
  T v[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int c = c0 + e;
    float f = 0.f;
    if (r < L && c < d) {
      f = (float)src[(size_t)r * ld + c];
      if (DROP && drop.thresh) f = nr_keep(drop.key, eidx0 + (uint32_t)r * erow + (uint32_t)c, drop.thresh) ? f * drop.scale : 0.f;
    }
    v[e] = (T)f;
  }
  if (ROWMAJOR) {
#pragma unroll
    for (int e = 0; e < 16; ++e) sR[r * SW + c0 + e] = v[e];
  }
  if (TRANSPOSED) {
#pragma unroll
    for (int e = 0; e < 16; ++e) sT[(c0 + e) * SW + r] = v[e];
  }
}


// 4-element vectors (8 B bf16 / 16 B f32) for the staging / store paths
template <typename T> struct V4;
template <> struct V4<bf16_t> { using type = bf16x4; };
template <> struct V4<float> { using type = f32x4; };

// Vector form of stage_head: a head slice row is d/4 chunks of 4 elements; lane (r, part) owns chunks
// part, part+2, part+4, part+6.  Needs d % 4 == 0 and a 4-element aligned slice (checked by the caller).
template <typename T, bool ROWMAJOR, bool TRANSPOSED, bool DROP>
__device__ __forceinline__ void stage_head_v(const T* __restrict__ src, size_t ld, int L, int d, T* sR, T* sT, int lane,
                                             const DropCfg& drop, uint32_t eidx0, uint32_t erow) {
  constexpr int SW = AT<T>::SW;
  using VT = typename V4<T>::type;
  const int r = lane & 31, part = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 4 * (2 * q + part);
    VT v = {(T)0.f, (T)0.f, (T)0.f, (T)0.f};
    if (r < L && c < d) {
      v = *reinterpret_cast<const VT*>(src + (size_t)r * ld + c);
      if (DROP && drop.thresh) {
        const uint32_t e0 = eidx0 + (uint32_t)r * erow + (uint32_t)c;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = nr_keep(drop.key, e0 + e, drop.thresh) ? (T)((float)v[e] * drop.scale) : (T)0.f;
      }
    }
    if (ROWMAJOR) *reinterpret_cast<VT*>(sR + r * SW + c) = v;
    if (TRANSPOSED) {
#pragma unroll
      for (int e = 0; e < 4; ++e) sT[(c + e) * SW + r] = v[e];
    }
  }
}

// accumulator tile (col c on the lane, rows in the registers) -> LDS image [row][c] of type T
template <typename T>
__device__ __forceinline__ void acc_to_lds(const f32x16& acc, float mul, T* sO, int lane) {
  constexpr int SW = AT<T>::SW;
  const int c = lane & 31, h = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) sO[rowof(r, h) * SW + c] = (T)(acc[r] * mul);
}

// LDS image [row][c] -> global rows (stride ld), 4 elements at a time; optional dropout by element index
template <typename T, bool DROP>
__device__ __forceinline__ void lds_to_global_v(const T* sO, T* __restrict__ dst, size_t ld, int L, int d, int lane,
                                                const DropCfg& drop, uint32_t eidx0, uint32_t erow) {
  constexpr int SW = AT<T>::SW;
  using VT = typename V4<T>::type;
  const int r = lane & 31, part = lane >> 5;
  if (r >= L) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 4 * (2 * q + part);
    if (c < d) {
      VT v = *reinterpret_cast<const VT*>(sO + r * SW + c);
      if (DROP && drop.thresh) {
        const uint32_t e0 = eidx0 + (uint32_t)r * erow + (uint32_t)c;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = nr_keep(drop.key, e0 + e, drop.thresh) ? (T)((float)v[e] * drop.scale) : (T)0.f;
      }
      *reinterpret_cast<VT*>(dst + (size_t)r * ld + c) = v;
    }
  }
}

// per-wave LDS carve
template <typename T> struct WaveLds {
  static constexpr int IMG = 32 * AT<T>::SW;  // elements of one image
};

// ------------------------------------------------------------------------------------------ forward
template <typename T>
__global__ __launch_bounds__(AW * 64) void attn_mfma_fwd_kernel(AttnMArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int IMG = WaveLds<T>::IMG;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  T* base = reinterpret_cast<T*>(smem) + (size_t)wid * 3 * IMG;
  T* sQ = base;
  T* sK = base + IMG;
  T* sVt = base + 2 * IMG;
  float* sMask = reinterpret_cast<float*>(reinterpret_cast<T*>(smem) + (size_t)AW * 3 * IMG) + wid * 32;
  const T* qkv = reinterpret_cast<const T*>(a.qkv);
  T* y = reinterpret_cast<T*>(a.y);
  const int N = a.N, L = a.L, d = a.d;
  const long total = (long)a.n * a.heads;
  const int h2 = lane >> 5, li = lane & 31;
  const bool vec = a.vec != 0;

  for (long p0 = (long)blockIdx.x * AW; p0 < total; p0 += (long)gridDim.x * AW) {
    const long p = p0 + wid;
    const bool active = p < total;
    const int seq = active ? (int)(p / a.heads) : 0, head = active ? (int)(p % a.heads) : 0;
    const size_t row0 = (size_t)seq * L;
    if (active) {
      const T* src = qkv + row0 * 3 * N + head * d;
      if (vec) {
        stage_head_v<T, true, false, false>(src, 3 * N, L, d, sQ, nullptr, lane, a.drop, 0, 0);
        stage_head_v<T, true, false, false>(src + N, 3 * N, L, d, sK, nullptr, lane, a.drop, 0, 0);
        stage_head_v<T, false, true, false>(src + 2 * N, 3 * N, L, d, nullptr, sVt, lane, a.drop, 0, 0);
      } else {
        stage_head<T, true, false, false>(src, 3 * N, L, d, sQ, nullptr, lane, a.drop, 0, 0);
        stage_head<T, true, false, false>(src + N, 3 * N, L, d, sK, nullptr, lane, a.drop, 0, 0);
        stage_head<T, false, true, false>(src + 2 * N, 3 * N, L, d, nullptr, sVt, lane, a.drop, 0, 0);
      }
      if (lane < 32) sMask[lane] = (lane < L) ? (a.mask ? a.mask[row0 + lane] : 1.f) : 0.f;
    }
    __syncthreads();
    if (active) {
      f32x16 st;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
      mm_rr(st, sK, sQ, lane);  // S^T[j][i]: rows j (registers), col i (lane)
      float m = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[r] *= a.scale;
        if (rowof(r, h2) < L) m = fmaxf(m, st[r]);
      }
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = rowof(r, h2);
        const float e = (j < L) ? __expf(st[r] - m) * sMask[j] : 0.f;
        st[r] = e;
        sum += e;
      }
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.f / (sum + 1e-8f * __expf(-m));
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] *= inv;
      f32x16 ctx;
#pragma unroll
      for (int r = 0; r < 16; ++r) ctx[r] = 0.f;
      mm_xt(ctx, st, sVt, lane);  // ctx[i][c]: rows i (registers), col c (lane)
      if (!vec) {
        if (li < d) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = rowof(r, h2);
            if (i < L) {
              float v = ctx[r];
              const size_t off = (row0 + i) * N + head * d + li;
              if (a.drop.thresh) v = nr_keep(a.drop.key, (uint32_t)off, a.drop.thresh) ? v * a.drop.scale : 0.f;
              y[off] = (T)v;
            }
          }
        }
      } else {
        acc_to_lds<T>(ctx, 1.f, sQ, lane);   // sQ is dead after S^T (wave-private region)
      }
    }
    __syncthreads();
    if (active && vec)
      lds_to_global_v<T, true>(sQ, y + row0 * N + head * d, N, L, d, lane, a.drop, (uint32_t)(row0 * N + head * d), (uint32_t)N);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------ backward
template <typename T>
__global__ __launch_bounds__(AW * 64) void attn_mfma_bwd_kernel(AttnMArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int IMG = WaveLds<T>::IMG;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  T* base = reinterpret_cast<T*>(smem) + (size_t)wid * 7 * IMG;
  T* sQ = base;
  T* sK = base + IMG;
  T* sV = base + 2 * IMG;
  T* sG = base + 3 * IMG;
  T* sQt = base + 4 * IMG;
  T* sKt = base + 5 * IMG;
  T* sGt = base + 6 * IMG;
  float* sF = reinterpret_cast<float*>(reinterpret_cast<T*>(smem) + (size_t)AW * 7 * IMG) + wid * 128;
  float* sMask = sF;        // [32]
  float* sM = sF + 32;      // [32] row max
  float* sInv = sF + 64;    // [32] 1/(Z)
  float* sRd = sF + 96;     // [32] sum_j a_ij dA_ij
  const T* qkv = reinterpret_cast<const T*>(a.qkv);
  const T* dy = reinterpret_cast<const T*>(a.dy);
  T* dqkv = reinterpret_cast<T*>(a.dqkv);
  const int N = a.N, L = a.L, d = a.d;
  const long total = (long)a.n * a.heads;
  const int h2 = lane >> 5, li = lane & 31;
  const bool vec = a.vec != 0;
  DropCfg nodrop;
  nodrop.key = 0; nodrop.thresh = 0; nodrop.scale = 1.f;

  for (long p0 = (long)blockIdx.x * AW; p0 < total; p0 += (long)gridDim.x * AW) {
    const long p = p0 + wid;
    const bool active = p < total;
    const int seq = active ? (int)(p / a.heads) : 0, head = active ? (int)(p % a.heads) : 0;
    const size_t row0 = (size_t)seq * L;
    if (active) {
      const T* src = qkv + row0 * 3 * N + head * d;
      if (vec) {
        stage_head_v<T, true, true, false>(src, 3 * N, L, d, sQ, sQt, lane, nodrop, 0, 0);
        stage_head_v<T, true, true, false>(src + N, 3 * N, L, d, sK, sKt, lane, nodrop, 0, 0);
        stage_head_v<T, true, false, false>(src + 2 * N, 3 * N, L, d, sV, nullptr, lane, nodrop, 0, 0);
        stage_head_v<T, true, true, true>(dy + row0 * N + head * d, N, L, d, sG, sGt, lane, a.drop,
                                          (uint32_t)(row0 * N + head * d), (uint32_t)N);
      } else {
        stage_head<T, true, true, false>(src, 3 * N, L, d, sQ, sQt, lane, nodrop, 0, 0);
        stage_head<T, true, true, false>(src + N, 3 * N, L, d, sK, sKt, lane, nodrop, 0, 0);
        stage_head<T, true, false, false>(src + 2 * N, 3 * N, L, d, sV, nullptr, lane, nodrop, 0, 0);
        stage_head<T, true, true, true>(dy + row0 * N + head * d, N, L, d, sG, sGt, lane, a.drop,
                                        (uint32_t)(row0 * N + head * d), (uint32_t)N);
      }
      if (lane < 32) sMask[lane] = (lane < L) ? (a.mask ? a.mask[row0 + lane] : 1.f) : 0.f;
    }
    __syncthreads();
    f32x16 dst;  // dS^T (lane = query i)
    if (active) {
      f32x16 st, dpt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
      mm_rr(st, sK, sQ, lane);   // S^T[j][i]
      mm_rr(dpt, sV, sG, lane);  // dP^T[j][i] = <V_j, G_i>
      float m = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[r] *= a.scale;
        if (rowof(r, h2) < L) m = fmaxf(m, st[r]);
      }
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float sum = 0.f, rdu = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = rowof(r, h2);
        const float e = (j < L) ? __expf(st[r] - m) * sMask[j] : 0.f;
        st[r] = e;
        sum += e;
        rdu = fmaf(e, dpt[r], rdu);
      }
      sum += __shfl_xor(sum, 32, 64);
      rdu += __shfl_xor(rdu, 32, 64);
      const float inv = 1.f / (sum + 1e-8f * __expf(-m));
      const float rd = rdu * inv;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[r] = st[r] * inv * (dpt[r] - rd);
      if (lane < 32) {
        sM[lane] = m;
        sInv[lane] = inv;
        sRd[lane] = rd;
      }
    }
    __syncthreads();
    f32x16 dq, dk, dv;
    if (active) {
      // dQ[i][c] = scale * sum_j dS[i][j] K[j][c]
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[r] = 0.f;
      mm_xt(dq, dst, sKt, lane);
      // second orientation: lane = key j, registers = query rows i
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      mm_rr(s, sQ, sK, lane);   // S[i][j]
      mm_rr(dp, sG, sV, lane);  // dP[i][j]
      const float mj = sMask[li];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = rowof(r, h2);
        const float pij = __expf(s[r] * a.scale - sM[i]) * mj * sInv[i];  // rows i >= L: multiplied by zero operands below
        s[r] = pij;                                                      // P[i][j]
        dp[r] = pij * (dp[r] - sRd[i]);                                  // dS[i][j]
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
      mm_xt(dk, dp, sQt, lane);  // dK[j][c] = sum_i dS[i][j] Q[i][c]   (Q rows >= L are zero)
      mm_xt(dv, s, sGt, lane);   // dV[j][c] = sum_i P[i][j] G[i][c]    (G rows >= L are zero)
      if (!vec && li < d) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int t = rowof(r, h2);
          if (t < L) {
            T* op = dqkv + (row0 + t) * 3 * N + head * d + li;
            op[0] = (T)(dq[r] * a.scale);
            op[N] = (T)(dk[r] * a.scale);
            op[2 * N] = (T)dv[r];
          }
        }
      }
    }
    if (vec) {
      __syncthreads();   // every product has consumed the wave's images: reuse three of them as output tiles
      if (active) {
        acc_to_lds<T>(dq, a.scale, sQ, lane);
        acc_to_lds<T>(dk, a.scale, sK, lane);
        acc_to_lds<T>(dv, 1.f, sV, lane);
      }
      __syncthreads();
      if (active) {
        T* op = dqkv + row0 * 3 * N + head * d;
        lds_to_global_v<T, false>(sQ, op, 3 * N, L, d, lane, nodrop, 0, 0);
        lds_to_global_v<T, false>(sK, op + N, 3 * N, L, d, lane, nodrop, 0, 0);
        lds_to_global_v<T, false>(sV, op + 2 * N, 3 * N, L, d, lane, nodrop, 0, 0);
      }
    }
    __syncthreads();
  }
}


// ==========================================================================================
// bf16 fast path (d % 4 == 0, aligned slices): every operand lives in ONE row-major 32 x 32 LDS
// image per matrix (64-byte rows, no padding, 16-byte chunk index XOR-swizzled by (row >> 2) & 3 so
// that both the ds_read_b128 row fragments and the transposed reads are bank-conflict free).
// Products that contract over tokens fetch their B operand with ds_read_b64_tr_b16 (a 4-token x
// 16-column block delivered column-major) straight from the row-major image, and the outputs go
// back to global memory through a transposed image written 8 bytes at a time and read with the
// same instruction -- no element-wise transposing LDS traffic at all.
// ==========================================================================================
namespace b16 {
constexpr int IMG = 32 * 32;
// Panel kernels (L <= 32): the AW head images of a group used to sit a whole number of KB apart, so the 8-byte pieces of a
// panel row (4 heads x d/4 pieces, scattered by one ds_write_b64 per 16 lanes) met on the same banks -- 3-4 way conflicts,
// 41-50 % of the LDS-active cycles (profiles/r02_final_sq_counters.txt).  Every head slot now carries HPAD elements of slack
// and starts hbank(h) elements into it: 0 / 64 / 32 / 96 bytes modulo the 128-byte bank window, the best a 16-byte granule
// allows (the b128 fragment reads need that alignment): model 153 -> 96 LDS cycles per 12 piece writes (48 = conflict free).
constexpr int HPAD = 64;
__device__ __forceinline__ constexpr int hbank(int h) { return 16 * (((h & 1) << 1) | ((h >> 1) & 1)); }
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ int ioff(int row, int col) { return row * 32 + ((((col >> 3) ^ ((row >> 2) & 3)) << 3) | (col & 7)); }
__device__ __forceinline__ bf16x4 trd(const bf16_t* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p)); }

// A head slice [L, d] travels global -> 4 x 8-byte registers per lane -> swizzled LDS image.  The two halves are
// separate so the NEXT (sequence, head) slice can be in flight while the current one is being computed.
struct Slice { bf16x4 v[4]; };
__device__ __forceinline__ void slice_load(Slice& s, const bf16_t* __restrict__ src, int ld, int L, int d, int lane) {
  const int r = lane & 31, part = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 4 * (2 * q + part);
    s.v[q] = (bf16x4){(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    if (r < L && c < d) s.v[q] = *reinterpret_cast<const bf16x4*>(src + (uint32_t)(r * ld + c));
  }
}
// row-indirect: token row r lives at table row ids[r] (ids points at the first token of this 32-row block)
__device__ __forceinline__ void slice_load_g(Slice& s, const bf16_t* __restrict__ table, int ld, const int32_t* __restrict__ ids, int coff,
                                             int L, int d, int lane) {
  const int r = lane & 31, part = lane >> 5;
  const bf16_t* rowp = table + (size_t)(r < L ? ids[r] : 0) * ld + coff;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 4 * (2 * q + part);
    s.v[q] = (bf16x4){(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    if (r < L && c < d) s.v[q] = *reinterpret_cast<const bf16x4*>(rowp + c);
  }
}
template <bool DROP>
__device__ __forceinline__ void slice_put(const Slice& s, int L, int d, bf16_t* img, int lane, const DropCfg& drop, uint32_t eidx0,
                                          uint32_t erow) {
  const int r = lane & 31, part = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 4 * (2 * q + part);
    bf16x4 v = s.v[q];
    if (DROP && drop.thresh && r < L && c < d) {
      const uint32_t kb = nr_keep4(drop.key, eidx0 + (uint32_t)r * erow + (uint32_t)c, drop.thresh);   // index is even (d % 4 == 0)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ((kb >> e) & 1u) ? (bf16_t)((float)v[e] * drop.scale) : (bf16_t)0.f;
    }
    *reinterpret_cast<bf16x4*>(img + ioff(r, c)) = v;
  }
}

// acc[row a][col b] += sum_c A[a][c] B[b][c]
__device__ __forceinline__ void mm_rr(f32x16& acc, const bf16_t* A, const bf16_t* B, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(A + ioff(r, 16 * s + 8 * h));
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(B + ioff(r, 16 * s + 8 * h));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
}

// acc[row i][col c] += sum_j X[j][i] Bm[j][c]; X = accumulator tile, Bm = ROW-MAJOR image [token j][c]
__device__ __forceinline__ void mm_xt(f32x16& acc, const f32x16& x, const bf16_t* Bm, int lane) {
  const int g16 = lane >> 4, h = g16 >> 1, cb = 16 * (g16 & 1), qq = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 a;
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = (bf16_t)x[8 * s + e];
    const bf16x4 lo = trd(Bm + ioff(16 * s + 4 * h + qq, cb + 4 * pp));
    const bf16x4 hi = trd(Bm + ioff(16 * s + 8 + 4 * h + qq, cb + 4 * pp));
    const bf16x8 b = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
}

// accumulator tile (col c on the lane, rows in registers) -> TRANSPOSED image Ot[c][row], 8 bytes per write
__device__ __forceinline__ void acc_to_img_t(const f32x16& acc, float mul, bf16_t* Ot, int lane) {
  const int c = lane & 31, h = lane >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bf16x4 v = {(bf16_t)(acc[4 * k] * mul), (bf16_t)(acc[4 * k + 1] * mul), (bf16_t)(acc[4 * k + 2] * mul),
                      (bf16_t)(acc[4 * k + 3] * mul)};
    *reinterpret_cast<bf16x4*>(Ot + ioff(c, 8 * k + 4 * h)) = v;
  }
}

// transposed image Ot[c][row] -> global rows, 4 columns (8 bytes) per lane and store
template <bool DROP>
__device__ __forceinline__ void img_t_to_global(const bf16_t* Ot, bf16_t* __restrict__ dst, int ld, int L, int d, int lane,
                                                const DropCfg& drop, uint32_t eidx0, uint32_t erow) {
  const int g16 = lane >> 4, part = g16 >> 1, rb = 16 * (g16 & 1), qq = (lane & 15) >> 2, pp = lane & 3, r = lane & 31;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 4 * (2 * q + part);
    bf16x4 v = trd(Ot + ioff(c + qq, rb + 4 * pp));   // executed by every lane (EXEC must be full)
    if (r < L && c < d) {
      if (DROP && drop.thresh) {
        const uint32_t kb = nr_keep4(drop.key, eidx0 + (uint32_t)r * erow + (uint32_t)c, drop.thresh);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ((kb >> e) & 1u) ? (bf16_t)((float)v[e] * drop.scale) : (bf16_t)0.f;
      }
      *reinterpret_cast<bf16x4*>(dst + (uint32_t)(r * ld + c)) = v;
    }
  }
}

// acc[col c][row... transposed product: acc[c][i] += sum_j Bm[j][c] X[j][i] -- the operands of mm_xt with their roles swapped, so
// the lane owns query/token i and its registers hold 4 CONSECUTIVE columns c = 8k + 4h .. +3 per group k: the result
// goes to global memory as 8-byte pieces straight from the accumulators (no LDS transposition, images stay clean).
__device__ __forceinline__ void mm_xt_T(f32x16& acc, const f32x16& x, const bf16_t* Bm, int lane) {
  const int g16 = lane >> 4, h = g16 >> 1, cb = 16 * (g16 & 1), qq = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 a;
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = (bf16_t)x[8 * s + e];
    const bf16x4 lo = trd(Bm + ioff(16 * s + 4 * h + qq, cb + 4 * pp));
    const bf16x4 hi = trd(Bm + ioff(16 * s + 8 + 4 * h + qq, cb + 4 * pp));
    const bf16x8 b = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0);
  }
}
// acc[c][j] += sum_i Bm[i][c] Xi[i][j]: mm_xt_T with the accumulator-tile operand replaced by a ROW-MAJOR image Xi[token i][j]
// (written by acc_to_img_t from the tile whose lanes own i): both operands contract over tokens through transposed reads
__device__ __forceinline__ void mm_tt_T(f32x16& acc, const bf16_t* Xi, const bf16_t* Bm, int lane) {
  const int g16 = lane >> 4, h = g16 >> 1, cb = 16 * (g16 & 1), qq = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int o0 = ioff(16 * s + 4 * h + qq, cb + 4 * pp), o1 = ioff(16 * s + 8 + 4 * h + qq, cb + 4 * pp);
    const bf16x4 lo = trd(Bm + o0), hi = trd(Bm + o1), xl = trd(Xi + o0), xh = trd(Xi + o1);
    const bf16x8 b = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    const bf16x8 x = {xl[0], xl[1], xl[2], xl[3], xh[0], xh[1], xh[2], xh[3]};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, x, acc, 0, 0, 0);
  }
}
template <bool DROP>
__device__ __forceinline__ void acc_t_to_global(const f32x16& acc, bf16_t* __restrict__ dst, int ld, int L, int d, int lane,
                                                const DropCfg& drop, uint32_t eidx0, uint32_t erow) {
  const int i = lane & 31, h = lane >> 5;
  if (i >= L) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = 8 * k + 4 * h;
    if (c < d) {
      float f[4] = {acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]};
      if (DROP && drop.thresh) {
        const uint32_t kb = nr_keep4(drop.key, eidx0 + (uint32_t)i * erow + (uint32_t)c, drop.thresh);
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = ((kb >> e) & 1u) ? f[e] * drop.scale : 0.f;
      }
      const bf16x4 v = {(bf16_t)f[0], (bf16_t)f[1], (bf16_t)f[2], (bf16_t)f[3]};
      *reinterpret_cast<bf16x4*>(dst + (uint32_t)(i * ld + c)) = v;
    }
  }
}

// Work items of a workgroup: sequences blockIdx.x, +gridDim.x, ... and, inside a sequence, AW heads at a time.
// Plain int counters (no 64-bit division in the loop).
struct ItemIter {
  int sb, hg;
  __device__ __forceinline__ void next(int hgroups, int stride) {
    if (++hg == hgroups) { hg = 0; sb += stride; }
  }
};

constexpr float LOG2E = 1.4426950408889634f;

// Cooperative staging.  The AW heads a workgroup works on are ADJACENT in memory: per token row the AW x d columns
// of Q (or K, V, dy) are one contiguous AW*2d-byte run.  Instead of every wave fetching its own 2d-byte sliver of
// 32 rows (32 cache lines touched per load instruction), the workgroup loads the [L, AW*d] panel as consecutive
// 8-byte pieces -- a wave instruction covers ~3 whole rows -- and scatters the pieces into the per-head LDS images.
// The piece -> (row, head, column) map is the same for every item, so it is computed once per thread.
// Pad rows/columns of the images are zeroed once: nothing ever overwrites them (outputs leave from registers).
// PT = pieces per thread and matrix: needs L * AW * d / 4 <= PT * AW * 64  (PT = 3 up to L * d = 768, else 4)
// two packed words per piece (the backward kernel sits at its register limit): token row | head slot << 8, and
// offset in the image block | first column inside the panel row << 16
template <int PT> struct Pieces {
  int rh[PT], lq[PT];
  __device__ __forceinline__ int row(int t) const { return rh[t] & 0xff; }
  __device__ __forceinline__ int hh(int t) const { return rh[t] >> 8; }
  __device__ __forceinline__ int loff(int t) const { return lq[t] & 0xffff; }
  __device__ __forceinline__ int q4(int t) const { return lq[t] >> 16; }
};
// A wave-uniform 32-bit word through the scalar data cache: s_load + lgkmcnt instead of a vector load + vmcnt(0), which
// would also wait for every store the wave still has in flight (loads and stores share vmcnt on gfx9).  Only for data
// written by EARLIER kernels (the scalar cache is not coherent with this kernel's own vector stores).
__device__ __forceinline__ uint32_t sload_u32(const void* p) {
  uint32_t v;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}

// FULL: every head slot of a group is a real head (heads % AW == 0).  The slots beyond the L*AW*d/4 pieces of a panel then
// WRAP onto real pieces (duplicates load / store the same bytes), so every thread issues the same number of memory
// instructions per item and the compiler can count them (s_waitcnt vmcnt(N) instead of vmcnt(0)).
template <int PT, bool FULL = false> __device__ __forceinline__ Pieces<PT> make_pieces(int tid, int L, int d, int nimg, bool slack = true) {
  Pieces<PT> pc;
  const int pph = d >> 2, ppr = AW * pph;            // pieces per head row / per panel row
  const uint32_t inv_ppr = (65536 + ppr - 1) / ppr, inv_pph = (65536 + pph - 1) / pph;   // exact for p < 1024, divisors <= 40
  const int npieces = L * ppr;
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    int p = tid + t * AW * 64;
    if (FULL) p %= npieces;                            // once per thread
    const int row = (int)(((uint32_t)p * inv_ppr) >> 16), q = p - row * ppr;
    const int h = (int)(((uint32_t)q * inv_pph) >> 16), c = 4 * (q - h * pph);
    const bool ok = row < L;
    pc.rh[t] = (row & 31) | ((ok ? h : AW) << 8);     // head slot AW = never active
    pc.lq[t] = ((slack ? h * (nimg * IMG + HPAD) + hbank(h) : h * nimg * IMG) + ioff(row & 31, c)) | ((4 * q) << 16);
  }
  return pc;
}
template <int PT> struct Panel { bf16x4 v[PT]; };
// src: first element of the panel (row 0, first column of the head group) in a row-major tensor with row stride ld.
template <int PT>
__device__ __forceinline__ void panel_load(Panel<PT>& r, const bf16_t* __restrict__ src, int ld, const Pieces<PT>& pc, int hcount) {
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    r.v[t] = (bf16x4){(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    if (pc.hh(t) < hcount) r.v[t] = *reinterpret_cast<const bf16x4*>(src + (uint32_t)(pc.row(t) * ld + pc.q4(t)));
  }
}
// FULL variants: no predicates (see make_pieces)
template <int PT>
__device__ __forceinline__ void panel_load_all(Panel<PT>& r, const bf16_t* __restrict__ src, int ld, const Pieces<PT>& pc) {
#pragma unroll
  for (int t = 0; t < PT; ++t) r.v[t] = *reinterpret_cast<const bf16x4*>(src + (uint32_t)(pc.row(t) * ld + pc.q4(t)));
}
template <int PT>
__device__ __forceinline__ void panel_store_all(const bf16_t* panel, int ops, bf16_t* __restrict__ dst, int ld, const Pieces<PT>& pc) {
#pragma unroll
  for (int t = 0; t < PT; ++t)
    *reinterpret_cast<bf16x4*>(dst + (uint32_t)(pc.row(t) * ld + pc.q4(t))) = *reinterpret_cast<const bf16x4*>(panel + pc.row(t) * ops + pc.q4(t));
}
// Per-ROW padding substitution (FULL kernels with a live-token mask tm of the sequence): a padding token's Q|K|V row is the
// bias, so its row of the qkv buffer is never written by the projection and never read here -- the load goes to row 0 of
// the buffer instead (an L2 hit, value ignored) and the LDS image gets the bias piece.
template <int PT>
__device__ __forceinline__ void panel_load_sub(Panel<PT>& r, const bf16_t* __restrict__ buf, uint32_t seq_off, int coff, int ld,
                                               const Pieces<PT>& pc, uint32_t tm) {
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const uint32_t row = (uint32_t)pc.row(t);
    const uint32_t off = (((tm >> row) & 1u) ? seq_off + row * (uint32_t)ld : 0u) + (uint32_t)(coff + pc.q4(t));
    r.v[t] = *reinterpret_cast<const bf16x4*>(buf + (size_t)off);
  }
}
template <int PT>
__device__ __forceinline__ void panel_put_sub(const Panel<PT>& r, bf16_t* img0, const Pieces<PT>& pc, const bf16_t* sbias, uint32_t tm) {
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const bf16x4 b = *reinterpret_cast<const bf16x4*>(sbias + pc.q4(t));
    const bf16x4 v = ((tm >> (uint32_t)pc.row(t)) & 1u) ? r.v[t] : b;
    *reinterpret_cast<bf16x4*>(img0 + pc.loff(t)) = v;
  }
}
// row-indirect variant: token row r of the sequence lives at table row ids[r] (ids points at the sequence's first token)
template <int PT>
__device__ __forceinline__ void panel_load_g(Panel<PT>& r, const bf16_t* __restrict__ table, int ld, const int32_t* __restrict__ ids,
                                             int coff, const Pieces<PT>& pc, int hcount) {
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    r.v[t] = (bf16x4){(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    if (pc.hh(t) < hcount) r.v[t] = *reinterpret_cast<const bf16x4*>(table + (size_t)ids[pc.row(t)] * ld + coff + pc.q4(t));
  }
}
// ... FULL flavour: every slot is a real piece (see make_pieces), no predicates
template <int PT>
__device__ __forceinline__ void panel_load_g_all(Panel<PT>& r, const bf16_t* __restrict__ table, int ld, const int32_t* __restrict__ ids,
                                                 int coff, const Pieces<PT>& pc) {
#pragma unroll
  for (int t = 0; t < PT; ++t) r.v[t] = *reinterpret_cast<const bf16x4*>(table + (size_t)ids[pc.row(t)] * ld + coff + pc.q4(t));
}
template <bool DROP, int PT, bool FULL = false>
__device__ __forceinline__ void panel_put(const Panel<PT>& r, bf16_t* img0, const Pieces<PT>& pc, const DropCfg& drop, uint32_t eidx0,
                                          int erow) {
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    if (FULL || pc.hh(t) < AW) {
      bf16x4 v = r.v[t];
      if (DROP && drop.thresh) {
        const uint32_t kb = nr_keep4(drop.key, eidx0 + (uint32_t)(pc.row(t) * erow + pc.q4(t)), drop.thresh);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ((kb >> e) & 1u) ? (bf16_t)((float)v[e] * drop.scale) : (bf16_t)0.f;
      }
      *reinterpret_cast<bf16x4*>(img0 + pc.loff(t)) = v;
    }
  }
}
// Q / K / V images of a sequence made of padding tokens only: every row is the bias (table in LDS), nothing was loaded
template <int PT, bool FULL = false>
__device__ __forceinline__ void panel_put_bias(bf16_t* img0, const Pieces<PT>& pc, const bf16_t* sbias, int hcount) {
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    if (FULL || pc.hh(t) < AW) {
      bf16x4 v = (bf16x4){(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
      if (FULL || pc.hh(t) < hcount) v = *reinterpret_cast<const bf16x4*>(sbias + pc.q4(t));
      *reinterpret_cast<bf16x4*>(img0 + pc.loff(t)) = v;
    }
  }
}
__device__ __forceinline__ void zero_images(bf16_t* imgs, int elems, int tid) {      // elems: a multiple of 8
  for (int i = tid; i < elems / 8; i += AW * 64) reinterpret_cast<uint4*>(imgs)[i] = make_uint4(0, 0, 0, 0);
}

// Outputs leave the same way they came: each wave drops its head's [L, d] result (lane = token, registers = 4-column
// pieces, see mm_xt_T) into a row-major [L][AW*d (+8 pad)] LDS panel, and after the barrier the workgroup stores the
// panel as consecutive 8-byte pieces.
template <bool DROP>
__device__ __forceinline__ void acc_t_to_panel(const f32x16& acc, bf16_t* panel, int ops, int wid, int L, int d, int lane,
                                               const DropCfg& drop, uint32_t eidx0, uint32_t erow) {
  const int i = lane & 31, h = lane >> 5;
  if (i >= L) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = 8 * k + 4 * h;
    if (c < d) {
      float f[4] = {acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]};
      if (DROP && drop.thresh) {
        const uint32_t kb = nr_keep4(drop.key, eidx0 + (uint32_t)i * erow + (uint32_t)c, drop.thresh);
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = ((kb >> e) & 1u) ? f[e] * drop.scale : 0.f;
      }
      const bf16x4 v = {(bf16_t)f[0], (bf16_t)f[1], (bf16_t)f[2], (bf16_t)f[3]};
      *reinterpret_cast<bf16x4*>(panel + i * ops + wid * d + c) = v;
    }
  }
}
template <int PT>
__device__ __forceinline__ void panel_store(const bf16_t* panel, int ops, bf16_t* __restrict__ dst, int ld, const Pieces<PT>& pc,
                                            int hcount) {
#pragma unroll
  for (int t = 0; t < PT; ++t)
    if (pc.hh(t) < hcount)
      *reinterpret_cast<bf16x4*>(dst + (uint32_t)(pc.row(t) * ld + pc.q4(t))) =
          *reinterpret_cast<const bf16x4*>(panel + pc.row(t) * ops + pc.q4(t));
}

// softmax scale and log2(e) are folded into one fma feeding v_exp_f32; the mask multiply only exists when a mask is
// given; one dropout hash serves two elements; all lane offsets are 32-bit.
// LC / DC / HC: compile-time sequence length, head width and head count (0 = taken from the arguments).  The title-level
// shape of the reference's defaults (30 tokens, 20 heads of 20) is instantiated with constants: both kernels are bound by
// instruction issue (~420 / ~560 VALU instructions per item and wave around 4 / 14 MFMAs), and constants fold the row /
// column predicates and the address arithmetic.
template <bool HAS_MASK, int PT, bool SUB, bool GATHER = false, bool FULL = false, int LC = 0, int DC = 0, int HC = 0>
// (5 waves per SIMD stated outright: left to itself the allocator parks the MFMA results in 16 AGPRs next to 80 VGPRs -- the
//  same 96 registers of the unified file, plus 64 v_accvgpr moves per item in a kernel bound by VALU issue)
__global__ __launch_bounds__(AW * 64) __attribute__((amdgpu_waves_per_eu(5, 5))) void fwd_kernel(AttnMArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> SGPR address math
  bf16_t* img0 = reinterpret_cast<bf16_t*>(smem);
  constexpr int HS = 3 * IMG + HPAD;                     // head slot: Q | K | V images + bank-offset slack (see hbank)
  bf16_t* base = img0 + (size_t)wid * HS + hbank(wid);
  bf16_t *sQ = base, *sK = base + IMG, *sV = base + 2 * IMG;
  float* sMask = reinterpret_cast<float*>(img0 + (size_t)AW * HS) + wid * 32;
  bf16_t* sOut = reinterpret_cast<bf16_t*>(reinterpret_cast<float*>(img0 + (size_t)AW * HS) + AW * 32);   // [32][ops]
  const bf16_t* qkv = reinterpret_cast<const bf16_t*>(a.qkv);
  bf16_t* y = reinterpret_cast<bf16_t*>(a.y);
  const int L = LC ? LC : a.L, d = DC ? DC : a.d, heads = HC ? HC : a.heads;
  const int N = heads * d, N3 = 3 * N, ops = AW * d + 4;     // ops / 4 odd (d even): the 8-byte row writes of 16 lanes hit 16 distinct slots
  const int h2 = lane >> 5;
  const float c1 = a.scale * LOG2E;   // scale > 0: the row maximum can be taken on the raw scores
  const Pieces<PT> pc = make_pieces<PT, FULL>(tid, L, d, 3);
  zero_images(img0, AW * HS, tid);
  bf16_t* sBias = sOut + 32 * ops;                     // SUB: bias [3N] as bf16 (what the projection of a zero row is)
  if (SUB)
    for (int i = tid; i < N3; i += AW * 64) sBias[i] = (bf16_t)a.bias[i];

  // One workgroup walks ALL heads of a sequence (AW heads at a time) before moving on; the panels of the NEXT item
  // are loaded into registers while the current item is computed.
  // FULL: every thread issues the same loads / stores for every item (duplicate pieces, dead sequences read sequence 0's
  // rows and ignore them), the per-sequence flag comes through the scalar cache -- nothing in the loop waits on vmcnt(0),
  // so the stores of item i drain while item i+1 is computed.
  const int hgroups = (heads + AW - 1) / AW, stride = gridDim.x;
  ItemIter it{(int)blockIdx.x, 0}, nx{(int)blockIdx.x, 0};
  const bool listed = a.seq_list != nullptr;             // walk a device-side list of sequences (the others were zero-filled)
  const int nseq = listed ? *a.seq_count : a.n;
  int sq_next = 0;                                       // the sequence sitting in the prefetch registers
  Panel<PT> rq, rk, rv;
  bool dead_next = false;                                // the item sitting in rq / rk / rv is all padding
  bool dead_seq = false;                                 // ... of the sequence the prefetcher is in
  constexpr bool ROWSUB = FULL && SUB && !GATHER;        // per-row padding substitution (see panel_load_sub)
  uint32_t tm_seq = 0, tm_item = 0;                      // live-token mask of the prefetcher's sequence / of the prefetched item
  const bool has_needed = a.needed != nullptr;
  bool skip_seq = false, skip_next = false, skip_cur = false;   // nobody uses the sequence's output: zeros are stored, nothing is computed
  auto prefetch = [&](const ItemIter& t) {
    const int hoff = t.hg * AW * d;
    const int hcount = min(AW, heads - t.hg * AW);
    if (t.hg == 0 || !FULL) {
      const int sbu = __builtin_amdgcn_readfirstlane(t.sb);
      sq_next = !listed ? sbu : (FULL ? (int)sload_u32(a.seq_list + sbu) : a.seq_list[sbu]);
      if (SUB) {                                         // a sequence of padding tokens only: Q|K|V = bias, nothing to load
        tm_seq = FULL ? sload_u32(a.tmask + sq_next) : a.tmask[sq_next];
        dead_seq = tm_seq == 0;
      }
      if (has_needed) skip_seq = (FULL ? sload_u32(a.needed + sq_next) : (uint32_t)a.needed[sq_next]) == 0;
    }
    skip_next = has_needed && skip_seq;
    dead_next = (SUB && dead_seq) || (FULL && skip_next);   // FULL: a skipped sequence loads like a dead one (rows ignored)
    if (GATHER) {
      const int32_t* idp = a.ids + (size_t)sq_next * L;
      if (FULL) {
        panel_load_g_all(rq, qkv, N3, idp, hoff, pc);
        panel_load_g_all(rk, qkv, N3, idp, N + hoff, pc);
        panel_load_g_all(rv, qkv, N3, idp, 2 * N + hoff, pc);
      } else {
        panel_load_g(rq, qkv, N3, idp, hoff, pc, hcount);
        panel_load_g(rk, qkv, N3, idp, N + hoff, pc, hcount);
        panel_load_g(rv, qkv, N3, idp, 2 * N + hoff, pc, hcount);
      }
    } else if (ROWSUB) {
      tm_item = skip_next ? 0u : tm_seq;                 // a skipped sequence loads nothing real either
      const uint32_t so = (uint32_t)sq_next * (uint32_t)(L * N3);
      panel_load_sub(rq, qkv, so, hoff, N3, pc, tm_item);
      panel_load_sub(rk, qkv, so, N + hoff, N3, pc, tm_item);
      panel_load_sub(rv, qkv, so, 2 * N + hoff, N3, pc, tm_item);
    } else if (FULL) {
      const bf16_t* src = qkv + (dead_next ? (size_t)0 : (size_t)sq_next * L * N3) + hoff;   // dead: any valid rows (L2 hits), ignored
      panel_load_all(rq, src, N3, pc);
      panel_load_all(rk, src + N, N3, pc);
      panel_load_all(rv, src + 2 * N, N3, pc);
    } else if (!dead_next) {
      const bf16_t* src = qkv + (size_t)sq_next * L * N3 + hoff;
      panel_load(rq, src, N3, pc, hcount);
      panel_load(rk, src + N, N3, pc, hcount);
      panel_load(rv, src + 2 * N, N3, pc, hcount);
    }
  };
  DropCfg nodrop;
  nodrop.key = 0; nodrop.thresh = 0; nodrop.scale = 1.f;
  // put: the item sitting in the prefetch registers goes into the LDS images (waits for its loads)
  auto put = [&](const ItemIter& t) {
    if (ROWSUB) {
      const int hoff = t.hg * AW * d;
      panel_put_sub(rq, img0, pc, sBias + hoff, tm_item);
      panel_put_sub(rk, img0 + IMG, pc, sBias + N + hoff, tm_item);
      panel_put_sub(rv, img0 + 2 * IMG, pc, sBias + 2 * N + hoff, tm_item);
    } else if (SUB && dead_next) {
      const int hoff = t.hg * AW * d, hcount = min(AW, heads - t.hg * AW);
      panel_put_bias<PT, FULL>(img0, pc, sBias + hoff, hcount);
      panel_put_bias<PT, FULL>(img0 + IMG, pc, sBias + N + hoff, hcount);
      panel_put_bias<PT, FULL>(img0 + 2 * IMG, pc, sBias + 2 * N + hoff, hcount);
    } else {
      panel_put<false, PT, FULL>(rq, img0, pc, nodrop, 0, 0);
      panel_put<false, PT, FULL>(rk, img0 + IMG, pc, nodrop, 0, 0);
      panel_put<false, PT, FULL>(rv, img0 + 2 * IMG, pc, nodrop, 0, 0);
    }
    if (HAS_MASK && lane < 32) sMask[lane] = (lane < L && t.hg * AW + wid < heads) ? a.mask[(size_t)sq_next * L + lane] : 0.f;
    skip_cur = skip_next;
  };
  __syncthreads();                                       // images zeroed, bias table in LDS
  if (nseq <= (int)blockIdx.x) return;                   // uniform: nothing for this workgroup
  prefetch(nx);
  put(nx);
  __syncthreads();
  // Rotated software pipeline: [issue the loads of item i+1] [compute item i] [store item i] [item i+1 -> LDS].  The loads
  // are older than the stores in the in-order vmcnt queue and both sit in one straight-line stretch of the loop body, so
  // the wait before the LDS writes is a COUNTED one (vmcnt = number of stores): the stores of item i drain while item i+1
  // is computed instead of stalling the wave at the top of the next iteration.
  for (; it.sb < nseq; it.next(hgroups, stride)) {
    const int head = it.hg * AW + wid;
    const bool active = head < heads;
    const size_t row0 = (size_t)sq_next * L;             // the current item's sequence: taken before the prefetch overwrites it
    const int Ls = active ? L : 0;                       // inactive waves store nothing
    nx.next(hgroups, stride);
    const ItemIter pf = nx.sb < nseq ? nx : it;          // the last item loads itself again: same instruction stream everywhere
    prefetch(pf);
    f32x16 ctx;
#pragma unroll
    for (int r = 0; r < 16; ++r) ctx[r] = 0.f;
    if (!skip_cur) {                                     // wave-uniform
      f32x16 st;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
      mm_rr(st, sK, sQ, lane);  // S^T[j][i] (unscaled)
      float m = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (rowof(r, h2) < L) m = fmaxf(m, st[r]);
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      const float mc = m * c1;
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = rowof(r, h2);
        float e = (j < L) ? __builtin_amdgcn_exp2f(fmaf(st[r], c1, -mc)) : 0.f;
        if (HAS_MASK) e *= sMask[j];
        st[r] = e;
        sum += e;
      }
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.f / (sum + 1e-8f * __builtin_amdgcn_exp2f(-mc));
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] *= inv;
      mm_xt_T(ctx, st, sV, lane);  // ctx^T[c][i]
    }
    const uint32_t e0 = (uint32_t)(row0 * N) + (uint32_t)((active ? head : 0) * d);
    if (skip_cur) acc_t_to_panel<false>(ctx, sOut, ops, wid, Ls, d, lane, nodrop, 0, 0);   // zeros, no dropout hashing
    else acc_t_to_panel<true>(ctx, sOut, ops, wid, Ls, d, lane, a.drop, e0, (uint32_t)N);
    __syncthreads();   // output panel complete; every wave is done with its images
    if (FULL) panel_store_all<PT>(sOut, ops, y + row0 * N + it.hg * AW * d, N, pc);
    else panel_store<PT>(sOut, ops, y + row0 * N + it.hg * AW * d, N, pc, min(AW, heads - it.hg * AW));
    put(pf);           // next item -> images
    __syncthreads();   // images complete; every wave has read the output panel
  }
}

template <bool HAS_MASK, int PT, bool SUB, bool FULL = false, int LC = 0, int DC = 0, int HC = 0, int OCC = 3, bool CPT = false>
// CPT (needs FULL && SUB, L <= 31): compact row storage + bias gradient, see AttnMArgs::pos / dump / db
// OCC = waves per SIMD the register allocation aims at.  The generic instantiations need ~160 VGPRs and spill heavily under a
// 128 cap (1.6 - 2.3 ms instead of 0.75); the shape-specialised one fitted 128 before the per-row padding substitution and
// spills a little with it -- 3 waves without spills win (NR_ATTN_BWD_OCC4 keeps the other build selectable)
__global__ __launch_bounds__(AW * 64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void bwd_kernel(AttnMArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> SGPR address math
  // Image block, MATRIX-major: [Q x AW waves | V x AW | K x AW | G x AW].  The output panels of an item are written over
  // the Q and V images (see the end of the loop) and dirty their zero padding for good; K and G stay clean, and every
  // product that contracts over the padded head dimension pairs a dirty operand with a clean one (K.Q^T, V.G^T), while
  // the token-contracting products meet exact zeros of P / dS / G in the padded rows.
  bf16_t* img0 = reinterpret_cast<bf16_t*>(smem);
  // SLIM (compact storage at 4 waves per SIMD): the workgroup must fit 40 KB of LDS -- no bank-offset slack between the head
  // images (1-2 % of time, see HPAD), no softmax-statistics block (nothing reads it any more), 20-float bias-gradient rows
  constexpr bool SLIM = CPT && OCC == 4;
  constexpr int HS = IMG + (SLIM ? 0 : HPAD), MS = AW * HS;   // head slot (image + bank-offset slack, see hoff) / matrix block
  constexpr int FW = SLIM ? (HAS_MASK ? 32 : 0) : 128;        // floats per wave behind the images: the key mask (+ legacy room)
  bf16_t *imQ = img0, *imV = img0 + MS, *imK = img0 + 2 * MS, *imG = img0 + 3 * MS;
  const int wo = wid * HS + (SLIM ? 0 : hbank(wid));
  bf16_t *sQ = imQ + wo, *sK = imK + wo, *sV = imV + wo, *sG = imG + wo;
  float* sF = reinterpret_cast<float*>(img0 + (size_t)4 * MS) + wid * FW;
  float* sMask = sF;
  static_assert(!CPT || (FULL && SUB), "compact row storage rides on the per-row padding substitution");
  const int L = LC ? LC : a.L, d = DC ? DC : a.d, heads = HC ? HC : a.heads;
  const int N = heads * d, N3 = 3 * N;
  const bf16_t* qkv = reinterpret_cast<const bf16_t*>(a.qkv);
  const bf16_t* dy = reinterpret_cast<const bf16_t*>(a.dy);
  bf16_t* dqkv = reinterpret_cast<bf16_t*>(a.dqkv);
  const int h2 = lane >> 5, li = lane & 31;
  const float c1 = a.scale * LOG2E;
  const Pieces<PT> pc = make_pieces<PT, FULL>(tid, L, d, 1, !SLIM);    // per matrix the AW head images are adjacent
  zero_images(img0, 4 * MS, tid);
  bf16_t* sBias = reinterpret_cast<bf16_t*>(reinterpret_cast<float*>(img0 + (size_t)4 * MS) + AW * FW);   // SUB: bias [3N] as bf16
  if (SUB)
    for (int i = tid; i < N3; i += AW * 64) sBias[i] = (bf16_t)a.bias[i];

  const int hgroups = (heads + AW - 1) / AW, stride = gridDim.x;
  // CPT: behind the bias table -- [32] positions of the current item's rows | per wave [32] column sums of dS | the
  // bias-gradient accumulators [hgroups][AW][3][DBS] (column c of head slot w of group hg of matrix Q / K / V)
  constexpr int CSW = SLIM ? 32 : 96, DBS = SLIM ? 20 : 32;
  int* sPos = reinterpret_cast<int*>(sBias + ((N3 + 7) / 8) * 8);
  float* sCs = reinterpret_cast<float*>(sPos + 32) + wid * CSW;
  float* sDb = reinterpret_cast<float*>(sPos + 32) + AW * CSW;
  if (CPT)
    for (int i = tid; i < hgroups * AW * 3 * DBS; i += AW * 64) sDb[i] = 0.f;
  int pos_next = 0;                                      // CPT: dqkv row of token (tid & 31) of the prefetched sequence
  bool gz_seq = false;                                   // CPT: the prefetcher's sequence has dy = 0 by contract (a.nzf): its rows are not trusted
  uint32_t tm_cur = 0;                                   // CPT: live-token mask of the CURRENT item's sequence
  ItemIter it{(int)blockIdx.x, 0}, nx{(int)blockIdx.x, 0};
  Panel<PT> rq, rk, rv, rg;
  bool dead_next = false;
  const bool listed = SUB && a.seq_list != nullptr;
  const int nseq = listed ? *a.seq_count : a.n;          // sequences to walk
  int sq_next = 0;                                       // the sequence sitting in the prefetch registers
  // FULL (see fwd_kernel): the sequence number and its padding flag come through the scalar cache once per sequence, every
  // thread issues the same 4 * PT loads and 3 * PT stores per item, so no wait in the loop is a vmcnt(0)
  bool dead_seq = false;
  constexpr bool ROWSUB = FULL && SUB;                   // per-row padding substitution (see panel_load_sub)
  uint32_t tm_seq = 0;                                   // live-token mask of the prefetcher's sequence
  auto prefetch = [&](const ItemIter& t) {
    if (FULL) {
      if (t.hg == 0) {
        const int sbu = __builtin_amdgcn_readfirstlane(t.sb);
        sq_next = listed ? (int)sload_u32(a.seq_list + sbu) : sbu;
        if (SUB) {
          tm_seq = sload_u32(a.tmask + sq_next);
          dead_seq = tm_seq == 0;
        }
        if (CPT) gz_seq = a.nzf != nullptr && sload_u32(a.nzf + sq_next) == 0;
      }
    } else {
      sq_next = listed ? a.seq_list[t.sb] : t.sb;
      dead_seq = SUB && a.tmask[sq_next] == 0;
    }
    const int sq = sq_next;
    const size_t r0 = (size_t)sq * L;
    const int hd = t.hg * AW * d;
    const int hcount = min(AW, heads - t.hg * AW);
    dead_next = SUB && dead_seq;
    if (ROWSUB) {
      const uint32_t so = (uint32_t)sq * (uint32_t)(L * N3);
      panel_load_sub(rq, qkv, so, hd, N3, pc, tm_seq);
      panel_load_sub(rk, qkv, so, N + hd, N3, pc, tm_seq);
      panel_load_sub(rv, qkv, so, 2 * N + hd, N3, pc, tm_seq);
      panel_load_all(rg, dy + r0 * N + hd, N, pc);
      if (CPT) pos_next = a.pos[(uint32_t)r0 + (uint32_t)min(tid & 31, L - 1)];     // every thread, every item: same instruction count
    } else if (FULL) {
      const bf16_t* src = qkv + (dead_next ? (size_t)0 : r0 * N3) + hd;   // dead: any valid rows (L2 hits), ignored
      panel_load_all(rq, src, N3, pc);
      panel_load_all(rk, src + N, N3, pc);
      panel_load_all(rv, src + 2 * N, N3, pc);
      panel_load_all(rg, dy + r0 * N + hd, N, pc);
    } else {
      const bf16_t* src = qkv + r0 * N3 + hd;
      if (!dead_next) {
        panel_load(rq, src, N3, pc, hcount);
        panel_load(rk, src + N, N3, pc, hcount);
        panel_load(rv, src + 2 * N, N3, pc, hcount);
      }
      panel_load(rg, dy + r0 * N + hd, N, pc, hcount);      // the upstream gradient has no padding rows
    }
  };
  DropCfg nodrop;
  nodrop.key = 0; nodrop.thresh = 0; nodrop.scale = 1.f;
  // put: the item sitting in the prefetch registers (sequence sq_next) goes into the LDS images
  auto put = [&](const ItemIter& t) {
    const size_t r0 = (size_t)sq_next * L;
    if (ROWSUB) {                                        // tm_seq still belongs to the prefetched item: put follows its prefetch
      const int hoff = t.hg * AW * d;
      panel_put_sub(rq, imQ, pc, sBias + hoff, tm_seq);
      panel_put_sub(rk, imK, pc, sBias + N + hoff, tm_seq);
      panel_put_sub(rv, imV, pc, sBias + 2 * N + hoff, tm_seq);
    } else if (SUB && dead_next) {
      const int hoff = t.hg * AW * d, hcount = min(AW, heads - t.hg * AW);
      panel_put_bias<PT, FULL>(imQ, pc, sBias + hoff, hcount);
      panel_put_bias<PT, FULL>(imK, pc, sBias + N + hoff, hcount);
      panel_put_bias<PT, FULL>(imV, pc, sBias + 2 * N + hoff, hcount);
    } else {
      panel_put<false, PT, FULL>(rq, imQ, pc, nodrop, 0, 0);
      panel_put<false, PT, FULL>(rk, imK, pc, nodrop, 0, 0);
      panel_put<false, PT, FULL>(rv, imV, pc, nodrop, 0, 0);
    }
    if (CPT && gz_seq) {                                 // (workgroup-uniform) dy = 0 by contract: what was loaded may be anything
#pragma unroll
      for (int t2 = 0; t2 < PT; ++t2) *reinterpret_cast<bf16x4*>(imG + pc.loff(t2)) = (bf16x4){(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    } else {
      panel_put<true, PT, FULL>(rg, imG, pc, a.drop, (uint32_t)(r0 * N) + (uint32_t)(t.hg * AW * d), N);
    }
    if (HAS_MASK && lane < 32) sMask[lane] = (lane < L && t.hg * AW + wid < heads) ? a.mask[r0 + lane] : 0.f;
    if (CPT) {
      if (tid < 32) sPos[tid] = pos_next;
      tm_cur = tm_seq;                                   // (put follows the item's own prefetch: tm_seq is still its mask)
    }
  };
  __syncthreads();                                       // images zeroed, bias table in LDS
  if (nseq <= (int)blockIdx.x) return;                   // uniform: nothing for this workgroup
  prefetch(nx);
  put(nx);
  __syncthreads();
  // rotated software pipeline, see fwd_kernel: loads(i+1) | compute(i) | stores(i) | item i+1 -> LDS with a counted wait
  for (; it.sb < nseq; it.next(hgroups, stride)) {
    const int head = it.hg * AW + wid;
    const bool active = head < heads;
    const size_t row0 = (size_t)sq_next * L;             // the current item's sequence: taken before the prefetch overwrites it
    const int Ls = active ? L : 0;
    nx.next(hgroups, stride);
    const ItemIter pf = nx.sb < nseq ? nx : it;
    prefetch(pf);
    // ONE orientation (S^T: the lane owns query i, its registers the keys j, softmax statistics lane-local).  The two
    // products that contract over QUERIES (dK, dV) used to recompute S, P, dP and dS in the other orientation -- 4 more
    // MFMAs, 16 more v_exp and ~110 more VALU instructions per item in a kernel that is VALU-issue bound (SQ counters,
    // profiles/r03a: VALU 3 x 21 % of the SIMD, MFMA 15 %).  Now P and dS go as row-major bf16 images [i][j] into the
    // wave's V image (dead once dP^T is formed) and come back through the same transposed reads that deliver Q^T and G^T.
    f32x16 dst;  // dS^T (lane = query i), carries the 1/sqrt(d) factor of dQ and dK
    f32x16 dq, dk, dv;
    {
      f32x16 st, dpt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
      mm_rr(st, sK, sQ, lane);   // S^T[j][i] (unscaled)
      mm_rr(dpt, sV, sG, lane);  // dP^T[j][i]
      float m = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (rowof(r, h2) < L) m = fmaxf(m, st[r]);
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      const float mc = m * c1;
      float sum = 0.f, rdu = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = rowof(r, h2);
        float e = (j < L) ? __builtin_amdgcn_exp2f(fmaf(st[r], c1, -mc)) : 0.f;
        if (HAS_MASK) e *= sMask[j];
        st[r] = e;
        sum += e;
        rdu = fmaf(e, dpt[r], rdu);
      }
      sum += __shfl_xor(sum, 32, 64);
      rdu += __shfl_xor(rdu, 32, 64);
      const float eps = 1e-8f * __builtin_amdgcn_exp2f(-mc);
      const float inv = 1.f / (sum + eps);
      const float rd = rdu * inv;
      const float invs = inv * a.scale;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dst[r] = st[r] * invs * (dpt[r] - rd);
        st[r] *= inv;                                    // P^T[j][i]
      }
      if (CPT) {
        // Bias gradient without storing the padding rows: db_q = sum_j (sum_i dS_ij) K_j, db_k = sum_i (sum_j dS_ij) Q_i,
        // db_v = sum_i (sum_j P_ij) G_i.  The three row / column sums travel through the products below as ONE EXTRA token
        // -- index 31 of the 32 x 32 tiles, free because L <= 31 -- and come out in lanes 31 / 63 of dq / dk / dv:
        //   dS[i][31] <- row sum of dS = scale * rd_i * (1 - rho_i), P[i][31] <- rho_i = sum_j P_ij = 1 - eps_i inv_i: the
        //     lane's own statistics (register 15 of the upper half-wave is key 31; K row 31 is zero, so dQ does not see it)
        //   Q[i][31] <- 1: row 31 of dK^T then holds the COLUMN sums of dS; they go through LDS into the registers of lanes
        //     31 / 63 (query 31) before the dQ product, which is therefore computed last
        const float er = eps * inv;
        const bool up = h2 != 0;
        dst[15] = up ? (li < L ? a.scale * rd * er : 0.f) : dst[15];
        st[15] = up ? (li < L ? 1.f - er : 0.f) : st[15];
        if (lane < 32) sQ[ioff(lane, 31)] = (bf16_t)1.f;
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[r] = 0.f;
        mm_xt_T(dq, dst, sK, lane); // dQ^T[c][i] = sum_j K[j][c] dS[i][j] / sqrt(d)
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
      acc_to_img_t(st, 1.f, sV, lane);   // P[i][j]
      mm_tt_T(dv, sV, sG, lane);         // dV^T[c][j] = sum_i G[i][c] P[i][j]
      acc_to_img_t(dst, 1.f, sV, lane);  // dS[i][j] (the wave's LDS operations execute in order: the reads above are done)
      mm_tt_T(dk, sV, sQ, lane);         // dK^T[c][j] = sum_i Q[i][c] dS[i][j] / sqrt(d)
    }
    if (CPT) {
      {
        if (h2) sCs[li] = dk[15];                          // dK^T[31][j]: column sums of dS
        const bool x31 = li == 31;
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {                      // registers 4k .. 4k+3 = keys 8k + 4 h2 .. + 3 (the wave's own LDS words)
          const f32x4 c4 = *reinterpret_cast<const f32x4*>(sCs + 8 * k + 4 * h2);
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[4 * k + e] = x31 ? c4[e] : dst[4 * k + e];
        }
        mm_xt_T(dq, dst, sK, lane);
      }
      if (CPT && li == 31) {
        // lanes 31 / 63: the item's bias-gradient contribution, columns c = 8k + 4 h2 + e of this wave's head; only these two
        // lanes ever touch the wave's accumulator rows, so a plain read-modify-write is enough
        float* acc = sDb + ((it.hg * AW + wid) * 3) * DBS + 4 * h2;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          if (8 * k + 4 * h2 < d) {
            f32x4 q4 = *reinterpret_cast<f32x4*>(acc + 8 * k), k4 = *reinterpret_cast<f32x4*>(acc + DBS + 8 * k),
                  v4 = *reinterpret_cast<f32x4*>(acc + 2 * DBS + 8 * k);
#pragma unroll
            for (int e = 0; e < 4; ++e) { q4[e] += dq[4 * k + e]; k4[e] += dk[4 * k + e]; v4[e] += dv[4 * k + e]; }
            *reinterpret_cast<f32x4*>(acc + 8 * k) = q4;
            *reinterpret_cast<f32x4*>(acc + DBS + 8 * k) = k4;
            *reinterpret_cast<f32x4*>(acc + 2 * DBS + 8 * k) = v4;
          }
        }
      }
    }
    // stored straight from the registers: routing dQ/dK/dV through LDS panels like the forward output measured
    // slower here (1.38 vs 1.18 ms)
    if (FULL || 3 * 32 * AW * d <= 2 * AW * IMG) {        // d <= 21 (FULL is only launched for such shapes)
      // dQ|dK|dV leave through three [32][AW*d + 4] panels that ALIAS the (now dead) Q and V images: the workgroup then
      // stores consecutive 8-byte pieces, 160-byte runs per row and matrix instead of 40-byte head slivers.  Row stride
      // AW*d + 4: an odd number of 8-byte slots, so the 16 lanes of a write group hit 16 distinct slots (AW*d = 80 gave
      // 4-way conflicts); 3 * 32 * (AW*d + 4) <= 2 * MS holds for every d this branch admits.
      const int ops = AW * d + 4;
      __syncthreads();                                   // every wave is done reading the images
      acc_t_to_panel<false>(dq, img0, ops, wid, Ls, d, lane, nodrop, 0, 0);
      acc_t_to_panel<false>(dk, img0 + 32 * ops, ops, wid, Ls, d, lane, nodrop, 0, 0);
      acc_t_to_panel<false>(dv, img0 + 64 * ops, ops, wid, Ls, d, lane, nodrop, 0, 0);
      __syncthreads();
      bf16_t* op = dqkv + row0 * N3 + it.hg * AW * d;
      const int hcount = min(AW, heads - it.hg * AW);
      if (CPT) {
        // compact storage: token row r goes to dqkv row sPos[r]; a padding row (bit r of tm_cur clear) is not stored -- its
        // slot re-stores the sequence's first live row instead (same bytes twice: the instruction count stays uniform); a
        // sequence without any live row stores into the dump row
        const bool dead = tm_cur == 0;                     // wave-uniform
        const uint32_t r0l = dead ? 0u : (uint32_t)__builtin_ctz(tm_cur);
        bf16_t* base = dead ? reinterpret_cast<bf16_t*>(a.dump) : dqkv;
        const uint32_t hcol = (uint32_t)(it.hg * AW * d);
#pragma unroll
        for (int t = 0; t < PT; ++t) {
          const uint32_t row = (uint32_t)pc.row(t), re = ((tm_cur >> row) & 1u) ? row : r0l;
          const uint32_t off = (dead ? 0u : (uint32_t)sPos[re] * (uint32_t)N3) + hcol + (uint32_t)pc.q4(t);
          const bf16_t* src = img0 + re * ops + pc.q4(t);
          *reinterpret_cast<bf16x4*>(base + off) = *reinterpret_cast<const bf16x4*>(src);
          *reinterpret_cast<bf16x4*>(base + off + N) = *reinterpret_cast<const bf16x4*>(src + 32 * ops);
          *reinterpret_cast<bf16x4*>(base + off + 2 * N) = *reinterpret_cast<const bf16x4*>(src + 64 * ops);
        }
      } else if (FULL) {
        panel_store_all<PT>(img0, ops, op, N3, pc);
        panel_store_all<PT>(img0 + 32 * ops, ops, op + N, N3, pc);
        panel_store_all<PT>(img0 + 64 * ops, ops, op + 2 * N, N3, pc);
      } else {
        panel_store<PT>(img0, ops, op, N3, pc, hcount);
        panel_store<PT>(img0 + 32 * ops, ops, op + N, N3, pc, hcount);
        panel_store<PT>(img0 + 64 * ops, ops, op + 2 * N, N3, pc, hcount);
      }
    } else {                                              // wider heads: the panels would not fit into two image groups
      bf16_t* op = dqkv + row0 * N3 + (active ? head : 0) * d;
      acc_t_to_global<false>(dq, op, N3, Ls, d, lane, nodrop, 0, 0);
      acc_t_to_global<false>(dk, op + N, N3, Ls, d, lane, nodrop, 0, 0);
      acc_t_to_global<false>(dv, op + 2 * N, N3, Ls, d, lane, nodrop, 0, 0);
    }
    __syncthreads();   // every wave has read the output panels (they alias the Q / V images)
    put(pf);           // next item -> images
    __syncthreads();
  }
  if (CPT) {
    // this workgroup's share of the bias gradient: 3N fp32 atomics (4096 workgroups x 4.8 KB: ~15 us of the chip's atomic rate)
    // (plain fp32 atomics: the caller does not use compact storage in deterministic mode)
    for (int i = tid; i < hgroups * AW * 3 * DBS; i += AW * 64) {
      const int c = i % DBS, mat = (i / DBS) % 3, head = i / (3 * DBS);          // accumulator slot hg * AW + w = head
      if (c < d && head < heads) atomicAdd(a.db + mat * N + head * d + c, sDb[i]);
    }
  }
}


// ---- 32 < L <= 64 (the clicked-history level): the 64 x 64 score matrix as 2 x 2 tiles of 32 x 32 -------------
// Every matrix is two row-block images; queries are processed one 32-row block at a time against both key blocks.
__device__ __forceinline__ int clampL(int L, int rb) { return min(32, max(0, L - 32 * rb)); }

// LC / DC / HC: compile-time sequence length, head width and head count (0 = from the arguments), as for fwd_kernel: the
// reference's user level (50 clicks, 20 heads of 20) gets its own instantiation
template <bool GATHER, int LC = 0, int DC = 0, int HC = 0>
__global__ __launch_bounds__(AW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void fwd64_kernel(AttnMArgs a) {   // (no AGPR parking, see fwd_kernel)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf16_t* base = reinterpret_cast<bf16_t*>(smem) + (size_t)wid * 7 * IMG;
  bf16_t *sQ = base, *sK = base + 2 * IMG, *sV = base + 4 * IMG, *sO = base + 6 * IMG;
  float* sMask = reinterpret_cast<float*>(reinterpret_cast<bf16_t*>(smem) + (size_t)AW * 7 * IMG) + wid * 64;
  const bf16_t* qkv = reinterpret_cast<const bf16_t*>(a.qkv);
  bf16_t* y = reinterpret_cast<bf16_t*>(a.y);
  const int L = LC ? LC : a.L, d = DC ? DC : a.d, heads = HC ? HC : a.heads, N = heads * d, h2 = lane >> 5;
  DropCfg nodrop;
  nodrop.key = 0; nodrop.thresh = 0; nodrop.scale = 1.f;
  const int hgroups = (heads + AW - 1) / AW;
  // Items = (sequence, head group), all head groups of a sequence back to back in one workgroup (its table rows / qkv rows
  // stay hot in L1/L2).  The six slices (2 row blocks x Q, K, V) of the NEXT item are requested before this item is computed
  // -- a load -> wait -> compute chain per item was most of this kernel's time (the id -> row -> slice chain of the gather
  // source twice over): gather ids are fetched one sequence ahead, slices one item ahead.
  Slice t[2][3];
  int idr[2] = {0, 0};                                   // GATHER: table rows of this lane's tokens (row blocks 0 / 1) of the sequence being loaded
  // GATHER with a key mask (eval: the clicked history is front-padded, src/dataset.py:17-24, and masked, user_log_mask): when the
  // unmasked positions of a sequence are ONE run [shift, shift + len) the sequence is treated as that run alone -- tokens
  // shift.. land in rows 0.., every key is valid, the output rows of the masked positions are zeros (the pooling that follows
  // gives them weight 0, src/model/model_utils.py:28) -- and a run of <= 32 tokens needs ONE 32 x 32 tile where the padded
  // sequence needed four.  Mathematically the same attention (the softmax maximum is taken over the valid keys only, the
  // reference's + 1e-8 travels with it); any other mask pattern takes the general path.  *_n: the sequence being loaded.
  int sh_n = 0, ln_n = L, sh_c = 0, ln_c = L;
  bool gen_n = false, gen_c = false;
  auto load_ids = [&](long sb) {
    if (GATHER) {
      const size_t row0 = (size_t)sb * L;
      sh_n = 0; ln_n = L; gen_n = false;
      if (a.mask != nullptr) {
        const float mv = lane < L ? a.mask[row0 + lane] : 0.f;
        const unsigned long long bal = __ballot(mv != 0.f);
        if (bal == 0) {
          sh_n = L; ln_n = 0;                            // nothing valid: every output row is zero
        } else {
          const int s0 = __builtin_ctzll(bal), hi = 64 - __builtin_clzll(bal);
          if (__popcll(bal) == hi - s0) { sh_n = s0; ln_n = hi - s0; } else gen_n = true;
        }
      }
      const int r = lane & 31;
      idr[0] = r < ln_n ? a.ids[row0 + sh_n + r] : 0;
      idr[1] = 32 + r < ln_n ? a.ids[row0 + sh_n + 32 + r] : 0;
    }
  };
  auto load = [&](long sb, int hgi) {
    const int hraw = hgi * AW + wid;
    const bool active = hraw < heads;
    const int head = active ? hraw : 0, Lw = active ? (GATHER ? ln_n : L) : 0;
    const size_t row0 = (size_t)sb * L;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int Lr = clampL(Lw, rb);
      if (GATHER) {
        const int r = lane & 31, part = lane >> 5;
        const bf16_t* rowp = qkv + (size_t)idr[rb] * 3 * N + head * d;
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c = 4 * (2 * q + part);
            t[rb][w].v[q] = (bf16x4){(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
            if (r < Lr && c < d) t[rb][w].v[q] = *reinterpret_cast<const bf16x4*>(rowp + w * N + c);
          }
      } else {
        const bf16_t* src = qkv + row0 * 3 * N + head * d;
#pragma unroll
        for (int w = 0; w < 3; ++w) slice_load(t[rb][w], src + (size_t)32 * rb * 3 * N + w * N, 3 * N, Lr, d, lane);
      }
    }
  };
  long sb = blockIdx.x;
  if (sb >= a.n) return;
  load_ids(sb);
  load(sb, 0);
  for (; sb < a.n; sb += gridDim.x)
    for (int hgi = 0; hgi < hgroups; ++hgi) {
      const int hraw = hgi * AW + wid;
      const bool active = hraw < heads;
      if (GATHER && hgi == 0) { sh_c = sh_n; ln_c = ln_n; gen_c = gen_n; }     // (the prefetch below moves *_n on to the next sequence)
      const int Lc = GATHER ? ln_c : L;                  // tokens of this sequence that take part
      const bool two = Lc > 32;                          // wave- and workgroup-uniform: the second row block is in use
      const int head = active ? hraw : 0, Lw = active ? Lc : 0;
      const size_t row0 = (size_t)sb * L;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const int Lr = clampL(Lw, rb);
        slice_put<false>(t[rb][0], Lr, d, sQ + rb * IMG, lane, nodrop, 0, 0);
        slice_put<false>(t[rb][1], Lr, d, sK + rb * IMG, lane, nodrop, 0, 0);
        slice_put<false>(t[rb][2], Lr, d, sV + rb * IMG, lane, nodrop, 0, 0);
      }
      sMask[lane] = (lane < Lw) ? ((a.mask && (!GATHER || gen_c)) ? a.mask[row0 + lane] : 1.f) : 0.f;
      if (GATHER && Lc < L) {
        // output rows of the masked positions (in front of and behind the run): zeros, this head group's columns
        const int hcols = min(AW, heads - hgi * AW) * d, ppr = hcols >> 2, nz = (L - Lc) * ppr;
        for (int u = threadIdx.x; u < nz; u += AW * 64) {
          const int i = u / ppr, c = (u - i * ppr) * 4, row = i < sh_c ? i : i + Lc;
          *reinterpret_cast<bf16x4*>(y + (row0 + row) * N + hgi * AW * d + c) = (bf16x4){(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        }
      }
      __syncthreads();
      // this lane's 16 + 16 key-mask values (0 beyond L), once per item instead of once per query block and element
      float mk0[16], mk1[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) { mk0[r] = sMask[rowof(r, h2)]; mk1[r] = sMask[32 + rowof(r, h2)]; }
      // next item -> registers (in flight during the MFMAs below)
      if (hgi + 1 < hgroups) {
        load(sb, hgi + 1);
      } else if (sb + gridDim.x < a.n) {
        load_ids(sb + gridDim.x);
        load(sb + gridDim.x, 0);
      }
#pragma unroll 1
      for (int qb = 0; qb < (two ? 2 : 1); ++qb) {
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
        mm_rr(s0, sK, sQ + qb * IMG, lane);         // keys 0..31   x queries of block qb
        if (two) mm_rr(s1, sK + IMG, sQ + qb * IMG, lane);   // keys 32..63
        // softmax over the keys of one query (a lane pair): exp(scale*s - m) = exp2(s*c - m*c), c = scale*log2(e): the raw
        // scores go through one fma + v_exp_f32; keys >= L have mask 0 (the key block 0 is always complete: L > 32)
        const float c2 = a.scale * 1.44269504088896341f;
        float mr = -INFINITY;                          // max of the RAW scores over the valid keys (scale > 0)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          mr = fmaxf(mr, s0[r]);
          // GATHER: the run length is a run-time value; key rows behind it are ZERO rows of the images (score exactly 0), which
          // may take part in the maximum -- any finite m gives the same softmax -- so no per-row predicate is needed
          if (GATHER ? two : (32 + rowof(r, h2) < L)) mr = fmaxf(mr, s1[r]);
        }
        mr = fmaxf(mr, __shfl_xor(mr, 32, 64));
        const float m = mr * a.scale, mc = -mr * c2;
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e0 = __builtin_amdgcn_exp2f(fmaf(s0[r], c2, mc)) * mk0[r];
          s0[r] = e0;
          sum += e0;
        }
        if (two) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float e1 = __builtin_amdgcn_exp2f(fmaf(s1[r], c2, mc)) * mk1[r];
            s1[r] = e1;
            sum += e1;
          }
        }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / (sum + 1e-8f * __expf(-m));
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] *= inv; s1[r] *= inv; }
        f32x16 ctx;
#pragma unroll
        for (int r = 0; r < 16; ++r) ctx[r] = 0.f;
        mm_xt(ctx, s0, sV, lane);
        if (two) mm_xt(ctx, s1, sV + IMG, lane);
        __syncthreads();
        acc_to_img_t(ctx, 1.f, sO, lane);
        __syncthreads();
        const size_t r0 = row0 + (GATHER ? sh_c : 0) + 32 * qb;
        img_t_to_global<true>(sO, y + r0 * N + head * d, N, clampL(Lw, qb), d, lane, a.drop, (uint32_t)(r0 * N + head * d), (uint32_t)N);
      }
      __syncthreads();
    }
}

template <int LC = 0, int DC = 0, int HC = 0>
__global__ __launch_bounds__(AW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void bwd64_kernel(AttnMArgs a) {   // (no AGPR parking, see fwd_kernel)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf16_t* base = reinterpret_cast<bf16_t*>(smem) + (size_t)wid * 9 * IMG;
  bf16_t *sQ = base, *sK = base + 2 * IMG, *sV = base + 4 * IMG, *sG = base + 6 * IMG, *sO = base + 8 * IMG;
  float* sF = reinterpret_cast<float*>(reinterpret_cast<bf16_t*>(smem) + (size_t)AW * 9 * IMG) + wid * 256;
  float *sMask = sF, *sM = sF + 64, *sInv = sF + 128, *sRd = sF + 192;
  const bf16_t* qkv = reinterpret_cast<const bf16_t*>(a.qkv);
  const bf16_t* dy = reinterpret_cast<const bf16_t*>(a.dy);
  bf16_t* dqkv = reinterpret_cast<bf16_t*>(a.dqkv);
  const int L = LC ? LC : a.L, d = DC ? DC : a.d, heads = HC ? HC : a.heads, N = heads * d, h2 = lane >> 5, li = lane & 31;
  DropCfg nodrop;
  nodrop.key = 0; nodrop.thresh = 0; nodrop.scale = 1.f;
  const int hgroups = (heads + AW - 1) / AW;
  for (long sb = blockIdx.x; sb < a.n; sb += gridDim.x)
    for (int hgi = 0; hgi < hgroups; ++hgi) {
      const int hraw = hgi * AW + wid;
      const bool active = hraw < heads;
      const int head = active ? hraw : 0, Lw = active ? L : 0;
      const size_t row0 = (size_t)sb * L;
      const bf16_t* src = qkv + row0 * 3 * N + head * d;
      Slice t[2][4];                                   // all eight slices requested before the first LDS write (see fwd64_kernel)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const int Lr = clampL(Lw, rb);
        const size_t ro = (size_t)32 * rb;
#pragma unroll
        for (int w = 0; w < 3; ++w) slice_load(t[rb][w], src + ro * 3 * N + w * N, 3 * N, Lr, d, lane);
        slice_load(t[rb][3], dy + (row0 + ro) * N + head * d, N, Lr, d, lane);
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const int Lr = clampL(Lw, rb);
        const size_t ro = (size_t)32 * rb;
        slice_put<false>(t[rb][0], Lr, d, sQ + rb * IMG, lane, nodrop, 0, 0);
        slice_put<false>(t[rb][1], Lr, d, sK + rb * IMG, lane, nodrop, 0, 0);
        slice_put<false>(t[rb][2], Lr, d, sV + rb * IMG, lane, nodrop, 0, 0);
        slice_put<true>(t[rb][3], Lr, d, sG + rb * IMG, lane, a.drop, (uint32_t)((row0 + ro) * N + head * d), (uint32_t)N);
      }
      sMask[lane] = (lane < Lw) ? (a.mask ? a.mask[row0 + lane] : 1.f) : 0.f;
      __syncthreads();
      bf16_t* op = dqkv + row0 * 3 * N + head * d;
      // phase A: per query block -> row statistics and dQ
#pragma unroll 1
      for (int qb = 0; qb < 2; ++qb) {
        f32x16 s0, s1, p0, p1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; p0[r] = 0.f; p1[r] = 0.f; }
        mm_rr(s0, sK, sQ + qb * IMG, lane);
        mm_rr(s1, sK + IMG, sQ + qb * IMG, lane);
        mm_rr(p0, sV, sG + qb * IMG, lane);         // dP^T
        mm_rr(p1, sV + IMG, sG + qb * IMG, lane);
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s0[r] *= a.scale; s1[r] *= a.scale;
          if (rowof(r, h2) < L) m = fmaxf(m, s0[r]);
          if (32 + rowof(r, h2) < L) m = fmaxf(m, s1[r]);
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f, rdu = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = rowof(r, h2);
          const float e0 = (j < L) ? __expf(s0[r] - m) * sMask[j] : 0.f;
          const float e1 = (32 + j < L) ? __expf(s1[r] - m) * sMask[32 + j] : 0.f;
          s0[r] = e0; s1[r] = e1;
          sum += e0 + e1;
          rdu = fmaf(e0, p0[r], fmaf(e1, p1[r], rdu));
        }
        sum += __shfl_xor(sum, 32, 64);
        rdu += __shfl_xor(rdu, 32, 64);
        const float inv = 1.f / (sum + 1e-8f * __expf(-m));
        const float rd = rdu * inv;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s0[r] = s0[r] * inv * (p0[r] - rd);       // dS^T
          s1[r] = s1[r] * inv * (p1[r] - rd);
        }
        if (lane < 32) {
          sM[32 * qb + lane] = m;
          sInv[32 * qb + lane] = inv;
          sRd[32 * qb + lane] = rd;
        }
        f32x16 dq;
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[r] = 0.f;
        mm_xt(dq, s0, sK, lane);
        mm_xt(dq, s1, sK + IMG, lane);
        __syncthreads();
        acc_to_img_t(dq, a.scale, sO, lane);
        __syncthreads();
        img_t_to_global<false>(sO, op + (size_t)32 * qb * 3 * N, 3 * N, clampL(Lw, qb), d, lane, nodrop, 0, 0);
      }
      __syncthreads();
      // phase B: per key block -> dK, dV (lane = key j, registers = query rows)
#pragma unroll 1
      for (int kb = 0; kb < 2; ++kb) {
        f32x16 dk, dv;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
        const float mj = sMask[32 * kb + li];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          f32x16 s, dp;
#pragma unroll
          for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
          mm_rr(s, sQ + qb * IMG, sK + kb * IMG, lane);
          mm_rr(dp, sG + qb * IMG, sV + kb * IMG, lane);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = 32 * qb + rowof(r, h2);
            const float pij = __expf(s[r] * a.scale - sM[i]) * mj * sInv[i];
            s[r] = pij;
            dp[r] = pij * (dp[r] - sRd[i]);
          }
          mm_xt(dk, dp, sQ + qb * IMG, lane);
          mm_xt(dv, s, sG + qb * IMG, lane);
        }
        const int Lk = clampL(Lw, kb);
        __syncthreads();
        acc_to_img_t(dk, a.scale, sO, lane);
        __syncthreads();
        img_t_to_global<false>(sO, op + (size_t)32 * kb * 3 * N + N, 3 * N, Lk, d, lane, nodrop, 0, 0);
        __syncthreads();
        acc_to_img_t(dv, 1.f, sO, lane);
        __syncthreads();
        img_t_to_global<false>(sO, op + (size_t)32 * kb * 3 * N + 2 * N, 3 * N, Lk, d, lane, nodrop, 0, 0);
      }
      __syncthreads();
    }
}


// ==========================================================================================
// Fused forward of the title level (K1 + K2 + K3): embedding gather -> dropout -> Q|K|V projection ->
// per-head attention -> dropout, one WAVE per title, the title's [32 x 320] input rows held in registers as
// MFMA operands for the whole kernel.  Per head the workgroup (4 waves = 4 titles) stages the head's
// [Q_h | K_h | V_h] weight rows (64 x 320, from L2) in LDS, double buffered; each wave then
//   1. projects its title: 80 x v_mfma_f32_16x16x32_bf16 with the weights as the MFMA row operand, so a lane
//      owns 4 consecutive output columns of one token (bias added in registers),
//   2. drops the 60 columns as 8-byte writes into its three private swizzled 32 x 32 images (and, when the
//      backward will need them, into the token-major Q|K|V buffer),
//   3. runs the 32 x 32 attention tile on 32x32x16 MFMA exactly as fwd_kernel above and stores y.
// Q|K|V never make a round trip through HBM on this path; with qkv == nullptr (inference) they are not
// written at all.
// ==========================================================================================
struct FusedArgs {
  const bf16_t* table; int ldt;
  const int32_t* ids;
  const bf16_t* w; int ldw;
  const float* bias;
  const float* mask;
  bf16_t* qkv;
  bf16_t* xrows; int ldxr;
  bf16_t* y;
  int n, L, heads, d, N, d_model;
  float scale;
  DropCfg drop_in, drop_out;
};

template <int KS>
__global__ __launch_bounds__(AW * 64) void fused_fwd_kernel(FusedArgs a) {
  constexpr int WS = KS * 32 + 8;                 // LDS row stride of the weight chunk (elements)
  constexpr int WCH = KS * 4;                     // 16-byte chunks per weight row
  constexpr int NWL = 64 * WCH / (AW * 64);       // weight chunks staged per thread per head
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* sW = reinterpret_cast<bf16_t*>(smem);                   // [2][64][WS]
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> SGPR address math
  bf16_t* img = sW + 2 * 64 * WS + (size_t)wid * 4 * IMG;
  bf16_t *sQ = img, *sK = img + IMG, *sV = img + 2 * IMG, *sO = img + 3 * IMG;
  float* sMask = reinterpret_cast<float*>(sW + 2 * 64 * WS + (size_t)AW * 4 * IMG) + wid * 32;
  const int N = a.N, L = a.L, d = a.d, h2 = lane >> 5;
  const int fr = lane & 15, g = lane >> 4;
  const long title = (long)blockIdx.x * AW + wid;
  const bool active = title < a.n;
  const size_t row0 = (size_t)(active ? title : 0) * L;

  // zero the private images once: pad columns (>= d) are never written again
  for (int e = lane; e < 4 * IMG / 8; e += 64) reinterpret_cast<uint4*>(img)[e] = make_uint4(0, 0, 0, 0);
  if (lane < 32) sMask[lane] = (active && lane < L) ? (a.mask ? a.mask[row0 + lane] : 1.f) : 0.f;

  // the title's input rows as MFMA operands: xf[i][s] = X[16 i + fr][32 s + 8 g .. + 8)
  bf16x8 xf[2][KS];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 16 * i + fr;
    const bool valid = active && row < L;
    const int id = valid ? a.ids[row0 + row] : 0;
    const bf16_t* src = a.table + (size_t)id * a.ldt + 8 * g;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bf16x8 v = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
      if (valid) v = *reinterpret_cast<const bf16x8*>(src + 32 * s);
      const int col = 32 * s + 8 * g;
      if (a.drop_in.thresh && valid) {
        const uint32_t e0 = (uint32_t)(row0 + row) * (uint32_t)a.d_model + (uint32_t)col;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (col + e < a.d_model) v[e] = nr_keep(a.drop_in.key, e0 + e, a.drop_in.thresh) ? (bf16_t)((float)v[e] * a.drop_in.scale) : (bf16_t)0.f;
      }
      xf[i][s] = v;
      if (a.xrows != nullptr && valid && col < a.ldxr) *reinterpret_cast<bf16x8*>(a.xrows + (row0 + row) * a.ldxr + col) = v;
    }
  }

  // weight chunk of head h: rows [Q_h (d) | K_h (d) | V_h (d) | zeros], staged global -> registers -> LDS
  uint4 wr[NWL];
  auto wload = [&](int h) {
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int c = tid + i * AW * 64, r = c / WCH, ch = c - r * WCH;
      uint4 z = make_uint4(0, 0, 0, 0);
      if (r < 3 * d) {
        const int which = r / d, rr = r - which * d;
        z = *reinterpret_cast<const uint4*>(a.w + (size_t)(which * N + h * d + rr) * a.ldw + ch * 8);
      }
      wr[i] = z;
    }
  };
  auto wstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int c = tid + i * AW * 64, r = c / WCH, ch = c - r * WCH;
      *reinterpret_cast<uint4*>(sW + (size_t)buf * 64 * WS + r * WS + ch * 8) = wr[i];
    }
  };
  wload(0);
  wstore(0);
  __syncthreads();

  for (int h = 0; h < a.heads; ++h) {
    const int cur = h & 1;
    if (h + 1 < a.heads) wload(h + 1);
    // ---- projection: C[token][col] for 32 tokens x 64 columns; a lane owns cols 16 j + 4 g .. + 4 of token 16 i + fr
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16_t* wc = sW + (size_t)cur * 64 * WS;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bf16x8 wf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(wc + (16 * j + fr) * WS + 32 * s + 8 * g);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i][s], acc[i][j], 0, 0, 0);
    }
    // ---- bias, bf16, images (+ token-major Q|K|V for the backward)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 16 * j + 4 * g;                 // first of the lane's 4 chunk columns
      if (c < 3 * d) {
        const int which = c / d, cc = c - which * d;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + which * N + h * d + cc);
        bf16_t* im = which == 0 ? sQ : (which == 1 ? sK : sV);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = 16 * i + fr;
          const f32x4 v = acc[i][j] + bv;
          const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          *reinterpret_cast<bf16x4*>(im + ioff(row, cc)) = o;
          if (a.qkv != nullptr && active && row < L)
            *reinterpret_cast<bf16x4*>(a.qkv + (row0 + row) * 3 * N + which * N + h * d + cc) = o;
        }
      }
    }
    if (h + 1 < a.heads) wstore(cur ^ 1);
    __syncthreads();
    // ---- attention tile (as fwd_kernel)
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
    mm_rr(st, sK, sQ, lane);
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] *= a.scale;
      if (rowof(r, h2) < L) m = fmaxf(m, st[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jj = rowof(r, h2);
      const float e = (jj < L) ? __expf(st[r] - m) * sMask[jj] : 0.f;
      st[r] = e;
      sum += e;
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / (sum + 1e-8f * __expf(-m));
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] *= inv;
    f32x16 ctx;
#pragma unroll
    for (int r = 0; r < 16; ++r) ctx[r] = 0.f;
    mm_xt(ctx, st, sV, lane);
    acc_to_img_t(ctx, 1.f, sO, lane);
    __syncthreads();
    img_t_to_global<true>(sO, a.y + row0 * N + h * d, N, active ? L : 0, d, lane, a.drop_out, (uint32_t)(row0 * N + h * d), (uint32_t)N);
    // next head's barrier (after its projection) orders these image reads before the next image writes:
    // the projection phase touches only registers and the weight buffer
  }
}

int launch_fused_fwd(const FusedArgs& a, hipStream_t stream) {
  constexpr int KS = 10;
  constexpr size_t smem = (size_t)2 * 64 * (KS * 32 + 8) * sizeof(bf16_t) + (size_t)AW * 4 * IMG * sizeof(bf16_t) + AW * 32 * sizeof(float);
  auto k = fused_fwd_kernel<KS>;
  NR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(k, dim3((a.n + AW - 1) / AW), dim3(AW * 64), smem, stream, a);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int launch(bool bwd, const AttnMArgs& a, hipStream_t stream) {
  long blocks = a.n;   // one workgroup per sequence (grid-stride beyond the cap), all heads inside
  const long cap = 256L * 4 * 4;
  if (blocks > cap) blocks = cap;
  if (a.L > 32) {
    const size_t smem64 = bwd ? AW * (9 * IMG * sizeof(bf16_t) + 256 * sizeof(float)) : AW * (7 * IMG * sizeof(bf16_t) + 64 * sizeof(float));
    if (bwd) {
      if (a.L == 50 && a.d == 20 && a.heads == 20 && !nr_opt(NR_OPT_ATTN_GENERIC)) {
        NR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bwd64_kernel<50, 20, 20>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem64));
        hipLaunchKernelGGL((bwd64_kernel<50, 20, 20>), dim3((unsigned)blocks), dim3(AW * 64), smem64, stream, a);
      } else {
        NR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bwd64_kernel<>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem64));
        hipLaunchKernelGGL(bwd64_kernel<>, dim3((unsigned)blocks), dim3(AW * 64), smem64, stream, a);
      }
    } else {
      const bool user50 = a.L == 50 && a.d == 20 && a.heads == 20 && !nr_opt(NR_OPT_ATTN_GENERIC);     // the reference's defaults
      if (a.ids != nullptr) {
        if (user50) hipLaunchKernelGGL((fwd64_kernel<true, 50, 20, 20>), dim3((unsigned)blocks), dim3(AW * 64), smem64, stream, a);
        else hipLaunchKernelGGL(fwd64_kernel<true>, dim3((unsigned)blocks), dim3(AW * 64), smem64, stream, a);
      } else if (user50) {
        hipLaunchKernelGGL((fwd64_kernel<false, 50, 20, 20>), dim3((unsigned)blocks), dim3(AW * 64), smem64, stream, a);
      } else {
        hipLaunchKernelGGL(fwd64_kernel<false>, dim3((unsigned)blocks), dim3(AW * 64), smem64, stream, a);
      }
    }
    NR_CHECK_LAUNCH();
    return NR_OK;
  }
  const size_t panel = (size_t)32 * (AW * a.d + 4) * sizeof(bf16_t);
  const size_t smem = bwd ? AW * (4 * (IMG + HPAD) * sizeof(bf16_t) + 128 * sizeof(float))
                          : AW * ((3 * IMG + HPAD) * sizeof(bf16_t) + 32 * sizeof(float)) + panel;
  const bool p3 = a.L * a.d <= 768, sub = a.tmask != nullptr;
  const bool cpt = bwd && a.pos != nullptr;              // compact row storage + bias gradient (see bwd_kernel, CPT)
  const int hgroups = (a.heads + AW - 1) / AW;
  size_t smem_s = smem + (sub ? (size_t)((3 * a.N + 7) / 8) * 8 * sizeof(bf16_t) : 0) +
                  (cpt ? (size_t)(32 + AW * 96 + hgroups * AW * 96) * sizeof(float) : 0);
  // the compact-storage kernel of the reference's title shape runs 4 workgroups per CU on a slimmer layout (bwd_kernel, SLIM)
  const bool slim = bwd && cpt && a.heads % AW == 0 && a.L == 30 && a.d == 20 && a.heads == 20 && p3 && !nr_opt(NR_OPT_ATTN_GENERIC) &&
                    !nr_opt(NR_OPT_ATTN_PRED) && !nr_opt(NR_OPT_ATTN_BWD_OCC4);
  if (slim)
    smem_s = (size_t)AW * (4 * IMG * sizeof(bf16_t) + (a.mask ? 32 : 0) * sizeof(float)) + (size_t)((3 * a.N + 7) / 8) * 8 * sizeof(bf16_t) +
             (size_t)(32 + AW * 32 + hgroups * AW * 3 * 20) * sizeof(float);
  if (cpt && nr_opt(NR_OPT_ATTN_BWD_GRID) > 0) blocks = std::min<long>(blocks, nr_opt(NR_OPT_ATTN_BWD_GRID));
  auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(AW * 64), smem_s, stream, a); };
  // FULL: all head slots real and the store panels alias the images (d <= 21 in the backward): unpredicated memory
  // instructions, counted waits (see fwd_kernel)
  // (the forward takes the backward's shape condition too: both directions must agree on who supplies padding rows)
  const bool full = a.heads % AW == 0 && a.L * a.d >= 64 && 3 * 32 * AW * a.d <= 2 * AW * IMG && !nr_opt(NR_OPT_ATTN_PRED);
  const bool title30 = full && p3 && a.L == 30 && a.d == 20 && a.heads == 20 && !nr_opt(NR_OPT_ATTN_GENERIC);   // the reference's defaults
  auto pick = [&](auto tag_mask, auto tag_sub) {
    constexpr bool HM = decltype(tag_mask)::value, SB = decltype(tag_sub)::value;
    if (bwd && cpt) {
      // (launcher contract: FULL shape with per-row substitution, L <= 31)
      if constexpr (SB) {
        if (title30 && slim) go(bwd_kernel<HM, 3, true, true, 30, 20, 20, 4, true>);          // (NR_ATTN_BWD_OCC4=1: the 3-wave build)
        else if (title30) go(bwd_kernel<HM, 3, true, true, 30, 20, 20, 3, true>);
        else p3 ? go(bwd_kernel<HM, 3, true, true, 0, 0, 0, 3, true>) : go(bwd_kernel<HM, 4, true, true, 0, 0, 0, 3, true>);
      }
    } else if (bwd) {
      // 3 waves per SIMD without spills beat 4 with the 11 scratch accesses per item the row substitution pushes the
      // 128-VGPR build into (same box: 0.74 vs 0.91 ms)
      if (title30) nr_opt(NR_OPT_ATTN_BWD_OCC4) ? go(bwd_kernel<HM, 3, SB, true, 30, 20, 20, 4>) : go(bwd_kernel<HM, 3, SB, true, 30, 20, 20, 3>);
      else if (full) p3 ? go(bwd_kernel<HM, 3, SB, true>) : go(bwd_kernel<HM, 4, SB, true>);
      else p3 ? go(bwd_kernel<HM, 3, SB>) : go(bwd_kernel<HM, 4, SB>);
    } else if (!SB && a.ids != nullptr) {
      // eval's title-level gather: the specialised, unpredicated instantiation for the reference's default shape too
      // (32 x 20 x 20: eval's short histories -- the last 32 clicks of a front-padded history, train.score_shard)
      const bool user32 = full && p3 && a.L == 32 && a.d == 20 && a.heads == 20 && !nr_opt(NR_OPT_ATTN_GENERIC);
      if (title30) go(fwd_kernel<HM, 3, false, true, true, 30, 20, 20>);
      else if (user32) go(fwd_kernel<HM, 3, false, true, true, 32, 20, 20>);
      else p3 ? go(fwd_kernel<HM, 3, false, true>) : go(fwd_kernel<HM, 4, false, true>);
    } else if (title30) {
      go(fwd_kernel<HM, 3, SB, false, true, 30, 20, 20>);
    } else if (full) {
      p3 ? go(fwd_kernel<HM, 3, SB, false, true>) : go(fwd_kernel<HM, 4, SB, false, true>);
    } else {
      p3 ? go(fwd_kernel<HM, 3, SB>) : go(fwd_kernel<HM, 4, SB>);
    }
  };
  if (a.mask) sub ? pick(std::true_type{}, std::true_type{}) : pick(std::true_type{}, std::false_type{});
  else sub ? pick(std::false_type{}, std::true_type{}) : pick(std::false_type{}, std::false_type{});
  NR_CHECK_LAUNCH();
  return NR_OK;
}
}  // namespace b16

template <typename T>
int launch_t(bool bwd, const AttnMArgs& a, hipStream_t stream) {
  const long total = (long)a.n * a.heads;
  long blocks = (total + AW - 1) / AW;
  const size_t img = (size_t)WaveLds<T>::IMG * sizeof(T);
  const size_t smem = bwd ? AW * (7 * img + 128 * sizeof(float)) : AW * (3 * img + 32 * sizeof(float));
  const long cap = 256L * (bwd ? 2 : 8) * 2;
  if (blocks > cap) blocks = cap;
  if (bwd) {
    auto k = attn_mfma_bwd_kernel<T>;
    NR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(AW * 64), smem, stream, a);
  } else {
    auto k = attn_mfma_fwd_kernel<T>;
    NR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(AW * 64), smem, stream, a);
  }
  NR_CHECK_LAUNCH();
  return NR_OK;
}

}  // namespace

// L <= 32: every dtype; 32 < L <= 64: the bf16 fast path only (vector-aligned slices); else the LDS/VALU kernels
bool nr_attn_mfma_supported(int L, int d_head) { return L >= 1 && L <= 64 && d_head >= 1 && d_head <= 32; }

// Padding-token substitution (tmask / bias) exists on the bf16 panel kernels only: L <= 32, d_head % 4 == 0, 8-byte
// aligned tensors.  nr_attn_pad_ok tells the caller beforehand; a launch that asks for it elsewhere is an error.
bool nr_attn_pad_ok(int dtype, int L, int d_head, const void* p0, const void* p1) {
  const bool old_path = nr_opt(NR_OPT_ATTN_OLD) || nr_opt(NR_OPT_ATTN_VALU) || nr_opt(NR_OPT_NO_PAD_SUB);
  return !old_path && dtype == NR_BF16 && L >= 1 && L <= 32 && d_head >= 4 && d_head <= 32 && d_head % 4 == 0 &&
         (((uintptr_t)p0 | (uintptr_t)p1) & 7) == 0;
}

int nr_launch_attn_mfma(bool bwd, int dtype, const void* qkv, const float* mask, void* y, const void* dy, void* dqkv, int n,
                        int L, int heads, int d_head, const DropCfg& drop, hipStream_t stream, const uint32_t* tmask,
                        const float* bias, const int32_t* seq_list, const int32_t* seq_count, const int32_t* needed) {
  if (!nr_attn_mfma_supported(L, d_head)) return -1;
  AttnMArgs a;
  a.tmask = nullptr; a.bias = nullptr; a.seq_list = nullptr; a.seq_count = nullptr; a.ids = nullptr; a.needed = needed;
  a.pos = nullptr; a.dump = nullptr; a.db = nullptr; a.nzf = nullptr;
  a.qkv = qkv; a.mask = mask; a.y = y; a.dy = dy; a.dqkv = dqkv;
  a.n = n; a.L = L; a.heads = heads; a.d = d_head; a.N = heads * d_head;
  a.scale = 1.0f / sqrtf((float)d_head);
  a.drop = drop;
  const uintptr_t al = dtype == NR_BF16 ? 7 : 15;
  a.vec = (d_head % 4 == 0) && (((uintptr_t)qkv | (uintptr_t)y | (uintptr_t)dy | (uintptr_t)dqkv) & al) == 0;
  const bool old_path = nr_opt(NR_OPT_ATTN_OLD) != 0;
  const bool fast = dtype == NR_BF16 && a.vec && !old_path;
  if (L > 32 && !fast) return -1;   // caller falls back to the LDS/VALU kernels
  // "_live": the backward walks a device-side list of sequences (n is then an upper bound)
  NrProfScope ps(stream, "attn_mfma_%s[%s,n=%d,L=%d,h=%d,d=%d]", bwd ? (seq_list ? "bwd_live" : "bwd") : (seq_list ? "fwd_live" : "fwd"),
                 dtype == NR_BF16 ? "bf16" : "f32", n, L, heads, d_head);
  if (tmask != nullptr) {
    if (!(fast && L <= 32 && bias != nullptr)) {
      nr_set_error("attention: padding-token substitution needs the bf16 panel kernels (L <= 32, aligned tensors)");
      return NR_ERR_ARG;
    }
    a.tmask = tmask; a.bias = bias;
    if (bwd && seq_list != nullptr) { a.seq_list = seq_list; a.seq_count = seq_count; }
  }
  if (!bwd && fast && L <= 32 && seq_list != nullptr) { a.seq_list = seq_list; a.seq_count = seq_count; }   // forward: needed sequences only
  if (fast) return b16::launch(bwd, a, stream);
  return dtype == NR_BF16 ? launch_t<bf16_t>(bwd, a, stream) : launch_t<float>(bwd, a, stream);
}


bool nr_attn_rowsub_ok(int dtype, int L, int d_head, int heads);
// Backward with compact row storage (see AttnMArgs::pos / dump / db): the caller checked nr_attn_compact_ok.
bool nr_attn_compact_ok(int dtype, int L, int d_head, int heads) { return nr_attn_rowsub_ok(dtype, L, d_head, heads) && L <= 31; }

int nr_launch_attn_bwd_compact(const void* qkv, const float* mask, const void* dy, void* dqkv, int n, int L, int heads, int d_head,
                               const DropCfg& drop, hipStream_t stream, const uint32_t* tmask, const float* bias, const int32_t* seq_list,
                               const int32_t* seq_count, const int32_t* pos, void* dump, float* db, const int32_t* nzf) {
  NR_CHECK_ARG(nr_attn_compact_ok(NR_BF16, L, d_head, heads) && tmask && bias && pos && dump && db && qkv && dy && dqkv,
               "attention backward (compact rows): shape or operands not eligible");
  NR_CHECK_ARG(((((uintptr_t)qkv) | ((uintptr_t)dy) | ((uintptr_t)dqkv) | ((uintptr_t)dump)) & 7) == 0, "attention backward (compact rows): 8-byte alignment");
  AttnMArgs a;
  a.ids = nullptr; a.needed = nullptr; a.y = nullptr;
  a.qkv = qkv; a.mask = mask; a.dy = dy; a.dqkv = dqkv;
  a.n = n; a.L = L; a.heads = heads; a.d = d_head; a.N = heads * d_head;
  a.scale = 1.0f / sqrtf((float)d_head);
  a.drop = drop;
  a.vec = 1;
  a.tmask = tmask; a.bias = bias; a.seq_list = seq_list; a.seq_count = seq_list ? seq_count : nullptr;
  a.pos = pos; a.dump = dump; a.db = db; a.nzf = nzf;
  NrProfScope ps(stream, "attn_mfma_bwd_rows[bf16,n=%d,L=%d,h=%d,d=%d]", n, L, heads, d_head);
  return b16::launch(true, a, stream);
}

// True when the bf16 panel kernels substitute the bias for padding ROWS themselves (FULL + live-token masks): the projection
// then need not write the bias into the padding rows of partly live sequences (nr_launch_bias_rows), nobody reads them.
bool nr_attn_rowsub_ok(int dtype, int L, int d_head, int heads) {
  return dtype == NR_BF16 && nr_attn_pad_ok(dtype, L, d_head, nullptr, nullptr) && heads % AW == 0 && L * d_head >= 64 &&
         3 * 32 * AW * d_head <= 2 * AW * b16::IMG && !nr_opt(NR_OPT_ATTN_PRED) && !nr_opt(NR_OPT_NO_ROW_SUB);
}

// Forward attention whose Q|K|V rows are gathered from a per-token-id table of projections [V, 3N] (bf16 panel kernel:
// L <= 32, d_head % 4 == 0, 8-byte aligned).  Returns -1 when the shape has no such kernel.
int nr_launch_attn_gather_fwd(const void* proj_table, const int32_t* ids, const float* mask, void* y, int n, int L, int heads,
                              int d_head, const DropCfg& drop, hipStream_t stream) {
  const bool long_seq = L > 32 && L <= 64 && d_head >= 4 && d_head <= 32 && d_head % 4 == 0 && ((((uintptr_t)proj_table) | ((uintptr_t)y)) & 7) == 0;
  if ((!nr_attn_pad_ok(NR_BF16, L, d_head, proj_table, y) && !long_seq) || ids == nullptr) return -1;
  AttnMArgs a;
  a.tmask = nullptr; a.bias = nullptr; a.seq_list = nullptr; a.seq_count = nullptr; a.needed = nullptr;
  a.pos = nullptr; a.dump = nullptr; a.db = nullptr; a.nzf = nullptr;
  a.ids = ids;
  a.qkv = proj_table; a.mask = mask; a.y = y; a.dy = nullptr; a.dqkv = nullptr;
  a.n = n; a.L = L; a.heads = heads; a.d = d_head; a.N = heads * d_head;
  a.scale = 1.0f / sqrtf((float)d_head);
  a.drop = drop;
  a.vec = 1;
  NrProfScope ps(stream, "attn_mfma_fwd_gather[bf16,n=%d,L=%d,h=%d,d=%d]", n, L, heads, d_head);
  return b16::launch(false, a, stream);
}

// Fused title-level forward (gather + dropout + Q|K|V projection + attention + dropout), bf16 only.
// Returns -1 when the shape is outside what the fused kernel covers (the caller then runs the unfused path).
bool nr_mhsa_fused_shape_ok(int L, int heads, int d_head, int d_model, int ldt, int ldw) {
  const bool off = nr_opt(NR_OPT_NO_FUSED_FWD) != 0;
  if (off) return false;
  return L >= 1 && L <= 32 && d_head % 4 == 0 && 3 * d_head <= 64 && d_model <= 320 && d_model > 288 && ldt >= 320 && ldw >= 320 &&
         (heads * d_head) % 4 == 0;
}

int nr_launch_mhsa_fused_fwd(const void* table, int ldt, const int32_t* ids, const void* w, int ldw, const float* bias,
                             const float* mask, void* qkv, void* xrows, int ldxr, void* y, int n, int L, int heads, int d_head,
                             int d_model, const DropCfg& drop_in, const DropCfg& drop_out, hipStream_t stream) {
  if (!nr_mhsa_fused_shape_ok(L, heads, d_head, d_model, ldt, ldw)) return -1;
  if ((heads * d_head) % 4 != 0 || (((uintptr_t)table | (uintptr_t)w | (uintptr_t)y | (uintptr_t)qkv | (uintptr_t)xrows) & 15) != 0) return -1;
  if (xrows != nullptr && (ldxr % 8 != 0)) return -1;
  b16::FusedArgs a;
  a.table = (const bf16_t*)table; a.ldt = ldt; a.ids = ids; a.w = (const bf16_t*)w; a.ldw = ldw; a.bias = bias; a.mask = mask;
  a.qkv = (bf16_t*)qkv; a.xrows = (bf16_t*)xrows; a.ldxr = ldxr; a.y = (bf16_t*)y;
  a.n = n; a.L = L; a.heads = heads; a.d = d_head; a.N = heads * d_head; a.d_model = d_model;
  a.scale = 1.0f / sqrtf((float)d_head);
  a.drop_in = drop_in; a.drop_out = drop_out;
  NrProfScope ps(stream, "mhsa_fused_fwd[bf16,n=%d,L=%d,h=%d,d=%d,D=%d,save_qkv=%d]", n, L, heads, d_head, d_model, qkv != nullptr);
  return b16::launch_fused_fwd(a, stream);
}
