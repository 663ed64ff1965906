// MFMA GEMMs of libnrhip (gfx950).  See nr_gemm.h for the operator definitions.
//
// Tile: 128 x 128 x 32 per workgroup of 4 waves (2 x 2), each wave 64 x 64 = 4 x 4 MFMA
// tiles of 16 x 16.  Both operands are staged into LDS k-contiguous, so one lane's
// fragment for a 32-deep k-step is 8 contiguous elements [row][8*(lane>>4) .. +8):
//   bf16: one v_mfma_f32_16x16x32_bf16 per (m,n) tile per k-step
//   f32 : eight v_mfma_f32_16x16x4_f32 (sub-step e takes k = 8*(lane>>4)+e from both operands)
// Global -> register -> LDS double buffering, one barrier per k-step.  The epilogue goes
// through an fp32 LDS tile so that global stores / atomics are 16-byte, row-contiguous.
#include <stdlib.h>

#include "nr_gemm.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, NTHR = 256;
constexpr int SC = BN + 4;  // fp32 epilogue tile row stride

template <typename T> struct TileCfg {
  static constexpr int CH = 16 / (int)sizeof(T);  // elements per 16-byte chunk (bf16 8, f32 4)
  static constexpr int SK = BK + CH;              // LDS row stride in elements (80 B / 144 B)
  static constexpr int CPR = BK / CH;             // chunks per tile row
  static constexpr int NCH = BM * CPR / NTHR;     // chunks per thread per operand (2 / 4)
  static constexpr int RPC = BM / CH;             // rows per permutation class (TN staging)
  static constexpr size_t STAGE_BYTES = (size_t)2 * 2 * BM * SK * sizeof(T);
};
constexpr size_t EPI_BYTES = (size_t)64 * SC * sizeof(float);   // epilogue tile: 64 rows at a time

template <typename T> constexpr size_t smem_bytes() {
  return TileCfg<T>::STAGE_BYTES > EPI_BYTES ? TileCfg<T>::STAGE_BYTES : EPI_BYTES;
}

union Chunk {
  uint4 u;
  float f[4];
  bf16_t h[8];
  uint32_t w[4];
};

// ---- operand row sources ---------------------------------------------------------------
template <typename T> __device__ __forceinline__ void drop_chunk(Chunk& c, const DropCfg& dr, uint32_t eidx, int col, int Dtrue);
template <> __device__ __forceinline__ void drop_chunk<float>(Chunk& c, const DropCfg& dr, uint32_t eidx, int col, int Dtrue) {
  if ((eidx & 1u) == 0) {   // usual case (even row length): two hashes for the four elements
    const uint32_t kb = nr_keep4(dr.key, eidx, dr.thresh);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (col + e < Dtrue) c.f[e] = ((kb >> e) & 1u) ? c.f[e] * dr.scale : 0.f;
    return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (col + e < Dtrue) c.f[e] = nr_keep(dr.key, eidx + e, dr.thresh) ? c.f[e] * dr.scale : 0.f;
}
template <> __device__ __forceinline__ void drop_chunk<bf16_t>(Chunk& c, const DropCfg& dr, uint32_t eidx, int col, int Dtrue) {
  if ((eidx & 1u) == 0) {
    const uint32_t kb = nr_keep4(dr.key, eidx, dr.thresh) | (nr_keep4(dr.key, eidx + 4, dr.thresh) << 4);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (col + e < Dtrue) c.h[e] = ((kb >> e) & 1u) ? (bf16_t)((float)c.h[e] * dr.scale) : (bf16_t)0.f;
    return;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (col + e < Dtrue) c.h[e] = nr_keep(dr.key, eidx + e, dr.thresh) ? (bf16_t)((float)c.h[e] * dr.scale) : (bf16_t)0.f;
}

// 16-byte chunk [m][k .. k+CH) of the logical [M, K] operand; zeros outside.
template <typename T, int KIND>
__device__ __forceinline__ uint4 load_rows_chunk(const RowSrc& s, int m, int k, int M, int K) {
  Chunk c;
  c.u = make_uint4(0, 0, 0, 0);
  if (m >= M || k >= K) return c.u;
  const T* p;
  uint32_t eidx;
  int col = k;
  if (KIND == ROWS_DENSE) {
    p = (const T*)s.base + (size_t)(s.gap > 0 ? m + m / s.gap : m) * s.ld + k;
    eidx = (uint32_t)m * (uint32_t)s.Dtrue + (uint32_t)k;
  } else if (KIND == ROWS_GATHER) {
    const int id = s.ids[(size_t)m * s.ids_stride];
    p = (const T*)s.base + (size_t)id * s.ld + k;
    eidx = (uint32_t)m * (uint32_t)s.Dtrue + (uint32_t)k;
  } else {  // ROWS_IM2COL3: k = tap*Dp + d ; source token t-1+tap of the block
    const int blk = m / s.Tlen, t = m - blk * s.Tlen;
    const int tap = k / s.ld, d = k - tap * s.ld;
    const int tt = t - 1 + tap;
    if (tt < 0 || tt >= s.Tlen) return c.u;
    const int id = s.ids[(size_t)blk * s.ids_stride];
    p = (const T*)s.base + ((size_t)id * s.Tlen + tt) * s.ld + d;
    eidx = (uint32_t)(blk * s.Tlen + tt) * (uint32_t)s.Dtrue + (uint32_t)d;
    col = d;
  }
  c.u = *reinterpret_cast<const uint4*>(p);
  // an all-zero chunk (padding token) stays zero whatever the mask says: no hashing
  if (s.drop.thresh && (c.u.x | c.u.y | c.u.z | c.u.w) != 0u) drop_chunk<T>(c, s.drop, eidx, col, s.Dtrue);
  return c.u;
}

// Per-(thread, chunk slot) row context for the NT kernel: the row a staging slot reads is the same
// for every k-step, so the id lookup (a dependent global load) is done once, not per step.
template <typename T> struct RowCtx {
  const T* p;     // dense / gather: row base ; im2col3: base of the gathered [Tlen, ld] block
  uint32_t e0;    // dropout element index of column 0 of this row (im2col3: of the block)
  int t;          // im2col3: token index inside the block
  bool valid;
};

template <typename T, int KIND>
__device__ __forceinline__ RowCtx<T> make_row_ctx(const RowSrc& s, int m, int M) {
  RowCtx<T> c;
  c.valid = m < M;
  c.p = nullptr; c.e0 = 0; c.t = 0;
  if (!c.valid) return c;
  if (KIND == ROWS_DENSE) {
    c.p = (const T*)s.base + (size_t)(s.gap > 0 ? m + m / s.gap : m) * s.ld;
    c.e0 = (uint32_t)m * (uint32_t)s.Dtrue;
  } else if (KIND == ROWS_GATHER) {
    const int id = s.ids[(size_t)m * s.ids_stride];
    c.p = (const T*)s.base + (size_t)id * s.ld;
    c.e0 = (uint32_t)m * (uint32_t)s.Dtrue;
  } else {
    const int blk = m / s.Tlen;
    c.t = m - blk * s.Tlen;
    const int id = s.ids[(size_t)blk * s.ids_stride];
    c.p = (const T*)s.base + (size_t)id * s.Tlen * s.ld;
    c.e0 = (uint32_t)(blk * s.Tlen) * (uint32_t)s.Dtrue;
  }
  return c;
}

template <typename T, int KIND>
__device__ __forceinline__ uint4 load_ctx_chunk(const RowSrc& s, const RowCtx<T>& c, int k, int K) {
  Chunk ch;
  ch.u = make_uint4(0, 0, 0, 0);
  if (!c.valid || k >= K) return ch.u;
  const T* p;
  uint32_t eidx;
  int col = k;
  if (KIND == ROWS_IM2COL3) {
    const int tap = k / s.ld, d = k - tap * s.ld;
    const int tt = c.t - 1 + tap;
    if (tt < 0 || tt >= s.Tlen) return ch.u;
    p = c.p + (size_t)tt * s.ld + d;
    eidx = c.e0 + (uint32_t)tt * (uint32_t)s.Dtrue + (uint32_t)d;
    col = d;
  } else {
    p = c.p + k;
    eidx = c.e0 + (uint32_t)k;
  }
  ch.u = *reinterpret_cast<const uint4*>(p);
  if (s.drop.thresh && (ch.u.x | ch.u.y | ch.u.z | ch.u.w) != 0u) drop_chunk<T>(ch, s.drop, eidx, col, s.Dtrue);
  return ch.u;
}

// Materialise rows(A) (gather + dropout) as a dense [M, K] operand: every element is hashed once here
// instead of once per 128-column tile inside the GEMM.
template <typename T, int KIND>
__global__ __launch_bounds__(256) void rows_materialize_kernel(RowSrc A, T* __restrict__ out, int ldo, int M, int K) {
  constexpr int CH = 16 / (int)sizeof(T);
  const int cpr = K / CH;
  const size_t total = (size_t)M * cpr;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(t / cpr), k = (int)(t - (size_t)m * cpr) * CH;
    *reinterpret_cast<uint4*>(out + (size_t)m * ldo + k) = load_rows_chunk<T, KIND>(A, m, k, M, K);
  }
}

// ---- MFMA fragments --------------------------------------------------------------------
struct F8 { float v[8]; };
template <typename T> struct FragOf;
template <> struct FragOf<bf16_t> { using type = bf16x8; };
template <> struct FragOf<float> { using type = F8; };

__device__ __forceinline__ void read_frag(bf16x8& f, const bf16_t* p) { f = *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void read_frag(F8& f, const float* p) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p);
  const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
  f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
  f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
}
__device__ __forceinline__ void mma(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma(f32x4& acc, const F8& a, const F8& b) {
#pragma unroll
  for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[e], b.v[e], acc, 0, 0, 0);
}

// SWAP = false: acc[i][j][r] = C[row i*16 + 4*(lane>>4) + r][col j*16 + (lane&15)]   (A rows on the MFMA row side)
// SWAP = true : acc[i][j][r] = C[row i*16 + (lane&15)][col j*16 + 4*(lane>>4) + r]   (a lane owns 4 consecutive columns)
template <typename T, bool SWAP>
__device__ __forceinline__ void mma_tile_step(f32x4 (&acc)[4][4], const T* sA, const T* sB, int wm, int wn, int lane) {
  using TL = TileCfg<T>;
  typename FragOf<T>::type a[4], b[4];
  const int fr = lane & 15, fk = (lane >> 4) * 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) read_frag(a[i], sA + (wm * 64 + i * 16 + fr) * TL::SK + fk);
#pragma unroll
  for (int j = 0; j < 4; ++j) read_frag(b[j], sB + (wn * 64 + j * 16 + fr) * TL::SK + fk);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (SWAP) mma(acc[i][j], b[j], a[i]);
      else mma(acc[i][j], a[i], b[j]);
    }
}

// ---- epilogue emit: 4 consecutive columns of one row ------------------------------------
template <int EPI>
__device__ __forceinline__ void emit4(const EpiArgs& ep, int m, int n, int N, f32x4 v) {
  if (EPI == EPI_POOLBWD) {
    const float rs = ep.rowscale[m];
    const float* g = ep.G + (size_t)(m / ep.L) * ep.ldg + n;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (n + e < N) v[e] += rs * g[e];
  }
  if (EPI == EPI_SCATTER) {
    const int id = ep.ids[(size_t)m * ep.ids_stride];
    if (id == 0) return;  // padding_idx row receives no gradient
    float* dst = (float*)ep.C + (size_t)id * ep.ldc + n;
    const uint32_t eidx = (uint32_t)m * (uint32_t)ep.Dtrue + (uint32_t)n;
    uint32_t kb = 0xfu;
    if (ep.drop.thresh) {
      if ((eidx & 1u) == 0) {
        kb = nr_keep4(ep.drop.key, eidx, ep.drop.thresh);
      } else {
        kb = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) kb |= nr_keep(ep.drop.key, eidx + e, ep.drop.thresh) ? (1u << e) : 0u;
      }
    }
    const bool det = nr_fix_on();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (n + e >= ep.Dtrue) break;
      const float x = ((kb >> e) & 1u) ? v[e] * ep.drop.scale : 0.f;
      if (x != 0.f) nr_accum(dst + e, x, det);
    }
    return;
  }
  if (ep.out_dtype == NR_F32) {
    float* dst = (float*)ep.C + (size_t)m * ep.ldc + n;
    if (n + 4 <= N) {
      *reinterpret_cast<f32x4*>(dst) = v;
    } else {
      for (int e = 0; e < 4 && n + e < N; ++e) dst[e] = v[e];
    }
  } else {
    bf16_t* dst = (bf16_t*)ep.C + (size_t)m * ep.ldc + n;
    if (n + 4 <= N) {
      bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      *reinterpret_cast<bf16x4*>(dst) = o;
    } else {
      for (int e = 0; e < 4 && n + e < N; ++e) dst[e] = (bf16_t)v[e];
    }
  }
}

// =========================================================================================
// NT
// =========================================================================================
template <typename T, int KIND, int EPI>
__global__ __launch_bounds__(NTHR) void gemm_nt_kernel(RowSrc A, const T* __restrict__ B, int ldb, int M, int N, int K,
                                                       EpiArgs ep, int tilesM, int tilesN) {
  using TL = TileCfg<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sA = reinterpret_cast<T*>(smem);        // [2][BM*SK]
  T* sB = sA + 2 * BM * TL::SK;              // [2][BN*SK]
  float* sC = reinterpret_cast<float*>(smem);

  // XCD-aware mapping: blocks b, b+8, ... share an XCD (and its L2); give one XCD all the
  // N-tiles of an M-tile back to back so the gathered A rows are re-read from that L2.
  const int b = blockIdx.x, xcd = b & 7, local = b >> 3;
  const int tm = (local / tilesN) * 8 + xcd, tn = local % tilesN;
  if (tm >= tilesM) return;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wid >> 1, wn = wid & 1;
  // bf16 outputs: operand roles are swapped so that a lane owns 4 consecutive output columns; the tile goes
  // through a bf16 LDS image (one 8-byte write per MFMA tile) and leaves as 16-byte row-contiguous stores.
  constexpr bool DIRECT = (sizeof(T) == 2) && (EPI != EPI_SCATTER);

  uint4 ra[TL::NCH], rb[TL::NCH];
  RowCtx<T> rctx[TL::NCH];
#pragma unroll
  for (int i = 0; i < TL::NCH; ++i) rctx[i] = make_row_ctx<T, KIND>(A, m0 + (tid + i * NTHR) / TL::CPR, M);
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < TL::NCH; ++i) {
      const int c = tid + i * NTHR, row = c / TL::CPR, kc = (c % TL::CPR) * TL::CH;
      ra[i] = load_ctx_chunk<T, KIND>(A, rctx[i], k0 + kc, K);
      uint4 z = make_uint4(0, 0, 0, 0);
      if (n0 + row < N && k0 + kc < K) z = *reinterpret_cast<const uint4*>(B + (size_t)(n0 + row) * ldb + k0 + kc);
      rb[i] = z;
    }
  };
  const bool save_rows = (ep.rows_out != nullptr) && (tn == 0);
  auto swrite = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < TL::NCH; ++i) {
      const int c = tid + i * NTHR, row = c / TL::CPR, kc = (c % TL::CPR) * TL::CH;
      *reinterpret_cast<uint4*>(sA + buf * BM * TL::SK + row * TL::SK + kc) = ra[i];
      *reinterpret_cast<uint4*>(sB + buf * BN * TL::SK + row * TL::SK + kc) = rb[i];
      if (save_rows && m0 + row < M && k0 + kc < K)
        *reinterpret_cast<uint4*>((T*)ep.rows_out + (size_t)(m0 + row) * ep.ld_rows_out + k0 + kc) = ra[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (K + BK - 1) / BK;
  gload(0);
  swrite(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BK);
    mma_tile_step<T, DIRECT>(acc, sA + cur * BM * TL::SK, sB + cur * BN * TL::SK, wm, wn, lane);
    if (kt + 1 < nk) swrite(cur ^ 1, (kt + 1) * BK);
    __syncthreads();
  }

  if (DIRECT) {
    bf16_t* sCb = reinterpret_cast<bf16_t*>(smem);
    constexpr int SCB = BN + 8;   // 272-byte rows
    const bool out_bf16 = ep.out_dtype == NR_BF16;
    // column tile outermost: the lane's 4 bias values are live for one tile column only (register pressure)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nl = wn * 64 + j * 16 + 4 * (lane >> 4), n = n0 + nl;
      f32x4 bvec = (f32x4){0.f, 0.f, 0.f, 0.f};
      if ((EPI == EPI_STORE || EPI == EPI_STORE_TANH) && ep.bias != nullptr) {
        if (n + 4 <= N) bvec = *reinterpret_cast<const f32x4*>(ep.bias + n);
        else
          for (int r = 0; r < 4; ++r)
            if (n + r < N) bvec[r] = ep.bias[n + r];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ml = wm * 64 + i * 16 + (lane & 15), m = m0 + ml;
        f32x4 v = acc[i][j] + bvec;
        if (EPI == EPI_STORE_TANH) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
        }
        if (EPI == EPI_POOLBWD && m < M && n < N) {
          const float rs = ep.rowscale[m];
          const float* grow = ep.G + (size_t)(m / ep.L) * ep.ldg;
          if (n + 4 <= N && (ep.ldg & 3) == 0) {
            v += rs * *reinterpret_cast<const f32x4*>(grow + n);
          } else {
            for (int r = 0; r < 4; ++r)
              if (n + r < N) v[r] += rs * grow[n + r];
          }
        }
        if (out_bf16) {
          const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          *reinterpret_cast<bf16x4*>(sCb + ml * SCB + nl) = o;
        } else if (m < M && n < N) {          // fp32 output from bf16 operands: direct 16-byte stores
          float* dst = (float*)ep.C + (size_t)m * ep.ldc + n;
          if (n + 4 <= N) *reinterpret_cast<f32x4*>(dst) = v;
          else
            for (int r = 0; r < 4 && n + r < N; ++r) dst[r] = v[r];
        }
      }
    }
    if (!out_bf16) return;
    __syncthreads();
    for (int u = tid; u < BM * (BN / 8); u += NTHR) {
      const int row = u / (BN / 8), c0 = (u % (BN / 8)) * 8;
      const int m = m0 + row, n = n0 + c0;
      if (m < M && n < N) {
        bf16_t* dst = (bf16_t*)ep.C + (size_t)m * ep.ldc + n;
        if (n + 8 <= N && (ep.ldc % 8) == 0) {
          *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(sCb + row * SCB + c0);
        } else {
          for (int e = 0; e < 8 && n + e < N; ++e) dst[e] = sCb[row * SCB + c0 + e];
        }
      }
    }
    return;
  }

  // accumulators -> fp32 LDS tile, 64 rows at a time (bias / tanh applied here, column is lane-constant),
  // then row-contiguous 16-byte stores / atomics.
  float bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = wn * 64 + j * 16 + (lane & 15);
    bv[j] = ((EPI == EPI_STORE || EPI == EPI_STORE_TANH) && ep.bias != nullptr && n0 + col < N) ? ep.bias[n0 + col] : 0.f;
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();
    if (wm == half) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = wn * 64 + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[i][j][r] + bv[j];
            if (EPI == EPI_STORE_TANH) v = tanhf(v);
            sC[(i * 16 + (lane >> 4) * 4 + r) * SC + col] = v;
          }
      }
    }
    __syncthreads();
    for (int u = tid; u < 64 * (BN / 4); u += NTHR) {
      const int row = u / (BN / 4), c0 = (u % (BN / 4)) * 4;
      const int m = m0 + half * 64 + row, n = n0 + c0;
      if (m < M && n < N) emit4<EPI>(ep, m, n, N, *reinterpret_cast<const f32x4*>(sC + row * SC + c0));
    }
  }
}

// =========================================================================================
// TN : dW[N,K] += sum_m dC[m,N]^T rows(A)[m,K]   (contraction over rows, split across blocks)
// Both operands arrive contraction-major in memory, so they are transposed while staging.
// A thread's 16-byte chunk holds CH consecutive tile rows; element e of chunk nc is written
// to LDS row rho = e*(128/CH) + nc so that simultaneous writes land on consecutive LDS rows
// (<= 2-way bank conflict for bf16).  The MFMA tile is therefore computed in permuted row /
// column order and un-permuted when the accumulators are written to the epilogue tile.
// =========================================================================================
template <typename T> __device__ __forceinline__ int unperm(int r) {
  using TL = TileCfg<T>;
  return (r % TL::RPC) * TL::CH + r / TL::RPC;
}

template <typename T, int KIND>
__global__ __launch_bounds__(NTHR) void gemm_tn_kernel(const T* __restrict__ dC, int ldc, RowSrc A, float* __restrict__ dW,
                                                       int ldw, float* __restrict__ db, int M, int N, int K, int Nstore, int Kstore,
                                                       int tilesK, int rows_per_split) {
  using TL = TileCfg<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bool det = nr_fix_on();          // deterministic mode: fixed-point accumulation of the outputs (nr_common.h)
  T* sA = reinterpret_cast<T*>(smem);
  T* sB = sA + 2 * BM * TL::SK;
  float* sC = reinterpret_cast<float*>(smem);
  float* sRed = reinterpret_cast<float*>(smem);   // aliases the staging buffers after the main loop

  const int tn = blockIdx.x / tilesK, tk = blockIdx.x % tilesK;
  const int n0 = tn * BM, k0 = tk * BN;
  const int mbeg = blockIdx.y * rows_per_split;
  const int mend = min(M, mbeg + rows_per_split);
  if (mbeg >= mend) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wid >> 1, wn = wid & 1;
  const bool do_db = (db != nullptr) && (tk == 0);

  constexpr bool IS_BF16 = (sizeof(T) == 2);
  // staging units: bf16: (nc = tid&15, row pair mp = tid>>4); f32: 4 units (nc = tid&31, m = (tid>>5) + 8*i)
  constexpr int NU = IS_BF16 ? 2 : 4;
  uint4 ra[NU], rb[NU];
  float colsum[TL::CH];
#pragma unroll
  for (int e = 0; e < TL::CH; ++e) colsum[e] = 0.f;

  auto gload = [&](int mt) {
    if (IS_BF16) {
      const int nc = tid & 15, mp = tid >> 4;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int m = mt + 2 * mp + h;
        uint4 z = make_uint4(0, 0, 0, 0);
        if (m < mend && n0 + nc * 8 < N) z = *reinterpret_cast<const uint4*>(dC + (size_t)m * ldc + n0 + nc * 8);
        ra[h] = z;
        rb[h] = load_rows_chunk<T, KIND>(A, m, k0 + nc * 8, mend, K);
      }
    } else {
      const int nc = tid & 31;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = mt + (tid >> 5) + 8 * i;
        uint4 z = make_uint4(0, 0, 0, 0);
        if (m < mend && n0 + nc * 4 < N) z = *reinterpret_cast<const uint4*>(dC + (size_t)m * ldc + n0 + nc * 4);
        ra[i] = z;
        rb[i] = load_rows_chunk<T, KIND>(A, m, k0 + nc * 4, mend, K);
      }
    }
  };
  auto swrite = [&](int buf) {
    T* a = sA + buf * BM * TL::SK;
    T* bq = sB + buf * BN * TL::SK;
    if (IS_BF16) {
      const int nc = tid & 15, mp = tid >> 4;
      Chunk a0, a1, b0, b1;
      a0.u = ra[0]; a1.u = ra[1]; b0.u = rb[0]; b1.u = rb[1];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int rho = e * 16 + nc;
        bf16x2 pa = {a0.h[e], a1.h[e]};
        bf16x2 pb = {b0.h[e], b1.h[e]};
        *reinterpret_cast<bf16x2*>(reinterpret_cast<bf16_t*>(a) + rho * TL::SK + 2 * mp) = pa;
        *reinterpret_cast<bf16x2*>(reinterpret_cast<bf16_t*>(bq) + rho * TL::SK + 2 * mp) = pb;
        if (do_db) colsum[e] += (float)a0.h[e] + (float)a1.h[e];
      }
    } else {
      const int nc = tid & 31;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ml = (tid >> 5) + 8 * i;
        Chunk ca, cb;
        ca.u = ra[i]; cb.u = rb[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int rho = e * 32 + nc;
          reinterpret_cast<float*>(a)[rho * TL::SK + ml] = ca.f[e];
          reinterpret_cast<float*>(bq)[rho * TL::SK + ml] = cb.f[e];
          if (do_db) colsum[e] += ca.f[e];
        }
      }
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (mend - mbeg + BK - 1) / BK;
  gload(mbeg);
  swrite(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload(mbeg + (kt + 1) * BK);
    mma_tile_step<T, false>(acc, sA + cur * BM * TL::SK, sB + cur * BN * TL::SK, wm, wn, lane);
    if (kt + 1 < nk) swrite(cur ^ 1);
    __syncthreads();
  }

  // bias gradient: column sums of dC over this block's rows (sRed aliases the drained staging buffers)
  if (do_db) {
    if (IS_BF16) {
      const int nc = tid & 15, mp = tid >> 4;
#pragma unroll
      for (int e = 0; e < TL::CH; ++e) sRed[mp * 128 + nc * 8 + e] = colsum[e];
    } else {
      const int nc = tid & 31, g = tid >> 5;
#pragma unroll
      for (int e = 0; e < TL::CH; ++e) sRed[g * 128 + nc * 4 + e] = colsum[e];
    }
    __syncthreads();
    if (tid < 128 && n0 + tid < Nstore) {
      float sacc = 0.f;
      constexpr int G = IS_BF16 ? 16 : 8;
#pragma unroll
      for (int g = 0; g < G; ++g) sacc += sRed[g * 128 + tid];
      nr_accum(db + n0 + tid, sacc, det);
    }
    __syncthreads();
  }
  // accumulators -> epilogue tile in natural (n, k) order, 64 output rows (n) at a time.  A wave's
  // 64 permuted rows unpermute to rows spread over the whole tile, so every wave writes in both halves.
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kl = unperm<T>(wn * 64 + j * 16 + (lane & 15));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int nl = unperm<T>(wm * 64 + i * 16 + (lane >> 4) * 4 + r);
          if ((nl >> 6) == half) sC[(nl & 63) * SC + kl] = acc[i][j][r];
        }
    }
    __syncthreads();
    for (int u = tid; u < 64 * (BN / 4); u += NTHR) {
      const int row = u / (BN / 4), c0 = (u % (BN / 4)) * 4;
      const int n = n0 + half * 64 + row, k = k0 + c0;
      if (n < Nstore && k < Kstore) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sC + row * SC + c0);
        float* dst = dW + (size_t)n * ldw + k;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k + e < Kstore) nr_accum(dst + e, v[e], det);
      }
    }
  }
}


// =========================================================================================
// TN v2 (bf16, dense operands): dW[N,K] += dC[M,N]^T X[M,K] without transposing writes.
// Both operands are staged in LDS exactly as they lie in memory ([m][n] / [m][k], 16-byte
// chunks, XOR-swizzled per row) and the MFMA fragments, which need 8 consecutive m for a fixed
// column, are fetched with ds_read_b64_tr_b16 (a 4-row x 16-column block delivered column-major
// to a 16-lane group).  Output tile 128 (n) x 160 (k) per workgroup, 32 rows of m per step.
// The per-row chunk swizzles make every transposed read conflict-free (a 32-lane half covers
// 8 rows x 32 B = all 64 banks once):
//   dC tile, 256-B rows : chunk ^= 2*(row & 3) + 8*((row >> 3) & 1)
//   X  tile, 320-B rows : chunk ^= 2*((row >> 3) & 1)   (rows are already skewed by 64 B)
// =========================================================================================
namespace tn2 {
constexpr int TBN = 128, TBK = 160, TBM = 64;   // 64 rows of m (two 32-deep MFMA k-steps) per barrier
constexpr int SCW = TBK + 4;
constexpr size_t STAGE = (size_t)2 * TBM * (TBN + TBK) * sizeof(bf16_t);   // 36,864 B
constexpr size_t EPI = (size_t)64 * SCW * sizeof(float);                     // 41,984 B
constexpr size_t SMEM = STAGE > EPI ? STAGE : EPI;

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ int swz_a(int row, int chunk) { return chunk ^ (2 * (row & 3) + 8 * ((row >> 3) & 1)); }
__device__ __forceinline__ int swz_b(int row, int chunk) { return chunk ^ (2 * ((row >> 3) & 1)); }

__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* p0, const bf16_t* p1) {
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p1));
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ __launch_bounds__(NTHR) void gemm_tn2_kernel(const bf16_t* __restrict__ dC, int ldc, const bf16_t* __restrict__ X,
                                                        int ldx, float* __restrict__ dW, int ldw, float* __restrict__ db, int M,
                                                        int N, int K, int Nstore, int Kstore, int tilesK, int ntile, int nsplit,
                                                        int rps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bool det = nr_fix_on();          // deterministic mode: fixed-point accumulation of the outputs (nr_common.h)
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem);           // [2][32][128]
  bf16_t* sB = sA + 2 * TBM * TBN;                        // [2][32][160]
  float* sC = reinterpret_cast<float*>(smem);
  float* sRed = reinterpret_cast<float*>(smem);

  // XCD-aware: the ntile workgroups of one split (same rows of dC / X) sit on one XCD back to back
  const int b = blockIdx.x, xcd = b & 7, local = b >> 3;
  const int split = (local / ntile) * 8 + xcd, tile = local % ntile;
  if (split >= nsplit) return;
  const int tn = tile / tilesK, tk = tile % tilesK;
  const int n0 = tn * TBN, k0 = tk * TBK;
  const int mbeg = split * rps, mend = min(M, mbeg + rps);
  if (mbeg >= mend) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wid >> 1, wn = wid & 1;
  const bool do_db = (db != nullptr) && (tk == 0);

  constexpr int NA = TBM * 16 / NTHR, NB = (TBM * 20 + NTHR - 1) / NTHR;
  uint4 ra[NA], rb[NB];
  float colsum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) colsum[e] = 0.f;

  auto gload = [&](int mt) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int c = tid + i * NTHR, row = c >> 4, ch = c & 15;
      uint4 z = make_uint4(0, 0, 0, 0);
      if (mt + row < mend && n0 + ch * 8 < N) z = *reinterpret_cast<const uint4*>(dC + (size_t)(mt + row) * ldc + n0 + ch * 8);
      ra[i] = z;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int c = tid + i * NTHR, row = c / 20, ch = c - row * 20;
      uint4 z = make_uint4(0, 0, 0, 0);
      if (c < TBM * 20 && mt + row < mend && k0 + ch * 8 < K) z = *reinterpret_cast<const uint4*>(X + (size_t)(mt + row) * ldx + k0 + ch * 8);
      rb[i] = z;
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int c = tid + i * NTHR, row = c >> 4, ch = c & 15;
      *reinterpret_cast<uint4*>(sA + buf * TBM * TBN + row * TBN + swz_a(row, ch) * 8) = ra[i];
      if (do_db) {
        Chunk cc;
        cc.u = ra[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) colsum[e] += (float)cc.h[e];
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int c = tid + i * NTHR, row = c / 20, ch = c - row * 20;
      if (c < TBM * 20) *reinterpret_cast<uint4*>(sB + buf * TBM * TBK + row * TBK + swz_b(row, ch) * 8) = rb[i];
    }
  };

  f32x4 acc[4][5];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transposed-read lane geometry: 16-lane group g holds m rows 8g..8g+7; lane 4q+p of the group
  // addresses row q, columns 4p..4p+3 of a 4 x 16 block
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int r0 = 8 * g + q, r1 = r0 + 4;

  const int nk = (mend - mbeg + TBM - 1) / TBM;
  gload(mbeg);
  swrite(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload(mbeg + (kt + 1) * TBM);
#pragma unroll
    for (int kh = 0; kh < TBM / 32; ++kh) {
      const bf16_t* a = sA + cur * TBM * TBN;
      const bf16_t* bq = sB + cur * TBM * TBK;
      const int ra0 = 32 * kh + r0, ra1 = 32 * kh + r1;
      bf16x8 af[4], bf[5];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int chunk = (wm * 64 + i * 16) / 8 + (pp >> 1), off = 4 * (pp & 1);
        af[i] = tr_frag(a + ra0 * TBN + swz_a(ra0, chunk) * 8 + off, a + ra1 * TBN + swz_a(ra1, chunk) * 8 + off);
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int chunk = (wn * 80 + j * 16) / 8 + (pp >> 1), off = 4 * (pp & 1);
        bf[j] = tr_frag(bq + ra0 * TBK + swz_b(ra0, chunk) * 8 + off, bq + ra1 * TBK + swz_b(ra1, chunk) * 8 + off);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) swrite(cur ^ 1);
    __syncthreads();
  }

  if (do_db) {
    const int ch = tid & 15, t = tid >> 4;
#pragma unroll
    for (int e = 0; e < 8; ++e) sRed[t * 128 + ch * 8 + e] = colsum[e];
    __syncthreads();
    if (tid < 128 && n0 + tid < Nstore) {
      float sacc = 0.f;
#pragma unroll
      for (int t2 = 0; t2 < 16; ++t2) sacc += sRed[t2 * 128 + tid];
      nr_accum(db + n0 + tid, sacc, det);
    }
    __syncthreads();
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();
    if (wm == half) {
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int col = wn * 80 + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) sC[(i * 16 + (lane >> 4) * 4 + r) * SCW + col] = acc[i][j][r];
      }
    }
    __syncthreads();
    for (int u = tid; u < 64 * TBK; u += NTHR) {   // lanes on consecutive floats (whole lines per atomic instruction)
      const int row = u / TBK, c = u - row * TBK;
      const int n = n0 + half * 64 + row, k = k0 + c;
      if (n < Nstore && k < Kstore) nr_accum(dW + (size_t)n * ldw + k, sC[row * SCW + c], det);
    }
  }
}

int launch(const void* dC, int ldc, const void* X, int ldx, float* dW, int ldw, float* db, int M, int N, int K, int Nstore,
           int Kstore, hipStream_t stream) {
  const int tilesN = (N + TBN - 1) / TBN, tilesK = (K + TBK - 1) / TBK, ntile = tilesN * tilesK;
  int nsplit = (256 * 6 + ntile - 1) / ntile;
  int rps = (M + nsplit - 1) / nsplit;
  rps = ((rps + TBM - 1) / TBM) * TBM;
  if (rps < 16 * TBM) rps = 16 * TBM;
  nsplit = (M + rps - 1) / rps;
  const int grid = ((nsplit + 7) / 8) * 8 * ntile;
  NR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM));
  hipLaunchKernelGGL(gemm_tn2_kernel, dim3(grid), dim3(NTHR), SMEM, stream, (const bf16_t*)dC, ldc, (const bf16_t*)X, ldx, dW, ldw,
                     db, M, N, K, Nstore, Kstore, tilesK, ntile, nsplit, rps);
  NR_CHECK_LAUNCH();
  return NR_OK;
}
}  // namespace tn2

// ---- host side ---------------------------------------------------------------------------
template <typename K> int set_smem(K kernel, size_t bytes) {
  NR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return NR_OK;
}


// =========================================================================================
// NT "wide" (bf16, dense A): one workgroup computes ALL N <= 16*NT16 output columns of a 128-row
// M-tile, so the activation operand A [M, K] is streamed from HBM exactly once (the tiled kernel
// re-reads it once per 128-column tile).  8 waves = 4 (M) x 2 (N halves); a wave owns 32 rows x
// NT16/2 column tiles (2 x NT16/2 MFMA tiles, <= 104 accumulator registers -> 2 waves per SIMD);
// the weight tile [N, 32] per k-step comes from L2.  Epilogue through a 32-row fp32 LDS tile
// (row-contiguous 8/16-byte stores, 256-byte atomic segments).
// Used for att_fc1 (N=200), the pooling backward dX (N=400) and the embedding-gradient GEMM (N=300).
// =========================================================================================
constexpr int WTHR = 512;

template <int EPI, int NT16>
__global__ __launch_bounds__(WTHR) void gemm_nt_wide_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B,
                                                            int ldb, int M, int N, int K, EpiArgs ep) {
  constexpr int SK = 40, WBN = NT16 * 16, HN = (NT16 + 1) / 2;   // HN: column tiles per wave
  constexpr int NCB = (WBN * 4 + WTHR - 1) / WTHR;
  constexpr int SCW = WBN + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem);   // [2][128][SK]
  bf16_t* sB = sA + 2 * BM * SK;                  // [2][WBN][SK]
  float* sC = reinterpret_cast<float*>(smem);     // epilogue: [32][SCW]
  const int m0 = blockIdx.x * BM;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wid >> 1, wn = wid & 1;

  uint4 ra, rb[NCB];
  auto gload = [&](int k0) {
    {
      const int row = tid >> 2, kc = (tid & 3) * 8;
      uint4 z = make_uint4(0, 0, 0, 0);
      if (m0 + row < M && k0 + kc < K) z = *reinterpret_cast<const uint4*>(A + (size_t)(m0 + row) * lda + k0 + kc);
      ra = z;
    }
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const int c = tid + i * WTHR, row = c >> 2, kc = (c & 3) * 8;
      uint4 z = make_uint4(0, 0, 0, 0);
      if (row < WBN && row < N && k0 + kc < K) z = *reinterpret_cast<const uint4*>(B + (size_t)row * ldb + k0 + kc);
      rb[i] = z;
    }
  };
  auto swrite = [&](int buf) {
    {
      const int row = tid >> 2, kc = (tid & 3) * 8;
      *reinterpret_cast<uint4*>(sA + buf * BM * SK + row * SK + kc) = ra;
    }
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const int c = tid + i * WTHR, row = c >> 2, kc = (c & 3) * 8;
      if (row < WBN) *reinterpret_cast<uint4*>(sB + buf * WBN * SK + row * SK + kc) = rb[i];
    }
  };

  f32x4 acc[2][HN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < HN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fk = (lane >> 4) * 8;
  const int nk = (K + BK - 1) / BK;
  gload(0);
  swrite(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BK);
    {
      const bf16_t* a = sA + cur * BM * SK;
      const bf16_t* bq = sB + cur * WBN * SK;
      bf16x8 af[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(a + (wm * 32 + i * 16 + fr) * SK + fk);
#pragma unroll
      for (int j = 0; j < HN; ++j) {
        const int jt = wn * HN + j;
        if (jt < NT16) {
          const bf16x8 bf = *reinterpret_cast<const bf16x8*>(bq + (jt * 16 + fr) * SK + fk);
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[i][j], 0, 0, 0);
        }
      }
    }
    if (kt + 1 < nk) swrite(cur ^ 1);
    __syncthreads();
  }
  // epilogue: 32 rows (one wm group) at a time through the fp32 LDS tile
#pragma unroll 1
  for (int pass = 0; pass < 4; ++pass) {
    if (pass) __syncthreads();
    if (wm == pass) {
#pragma unroll
      for (int j = 0; j < HN; ++j) {
        const int jt = wn * HN + j;
        if (jt < NT16) {
          const int col = jt * 16 + (lane & 15);
          float bv = 0.f;
          if ((EPI == EPI_STORE || EPI == EPI_STORE_TANH) && ep.bias != nullptr && col < N) bv = ep.bias[col];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = acc[i][j][r] + bv;
              if (EPI == EPI_STORE_TANH) v = tanhf(v);
              sC[(i * 16 + (lane >> 4) * 4 + r) * SCW + col] = v;
            }
        }
      }
    }
    __syncthreads();
    for (int u = tid; u < 32 * (WBN / 4); u += WTHR) {
      const int row = u / (WBN / 4), c0 = (u % (WBN / 4)) * 4;
      const int m = m0 + pass * 32 + row;
      if (m < M && c0 < N) emit4<EPI>(ep, m, c0, N, *reinterpret_cast<const f32x4*>(sC + row * SCW + c0));
    }
  }
}

template <int EPI, int NT16>
int launch_nt_wide_t(const RowSrc& A, const void* B, int ldb, int M, int N, int K, const EpiArgs& ep, hipStream_t stream) {
  constexpr size_t stage = (size_t)2 * (BM + NT16 * 16) * 40 * sizeof(bf16_t);
  constexpr size_t epi = (size_t)32 * (NT16 * 16 + 4) * sizeof(float);
  constexpr size_t smem = stage > epi ? stage : epi;
  auto kern = gemm_nt_wide_kernel<EPI, NT16>;
  NR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(kern, dim3((M + BM - 1) / BM), dim3(WTHR), smem, stream, (const bf16_t*)A.base, A.ld, (const bf16_t*)B, ldb, M,
                     N, K, ep);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

template <int NT16>
int launch_nt_wide_e(const RowSrc& A, const void* B, int ldb, int M, int N, int K, int epi, const EpiArgs& ep, hipStream_t s) {
  switch (epi) {
    case EPI_STORE: return launch_nt_wide_t<EPI_STORE, NT16>(A, B, ldb, M, N, K, ep, s);
    case EPI_STORE_TANH: return launch_nt_wide_t<EPI_STORE_TANH, NT16>(A, B, ldb, M, N, K, ep, s);
    case EPI_POOLBWD: return launch_nt_wide_t<EPI_POOLBWD, NT16>(A, B, ldb, M, N, K, ep, s);
    case EPI_SCATTER: return launch_nt_wide_t<EPI_SCATTER, NT16>(A, B, ldb, M, N, K, ep, s);
  }
  nr_set_error("gemm_nt_wide: bad epilogue %d", epi);
  return NR_ERR_ARG;
}


// =========================================================================================
// NT "wide" with an LDS-DMA ring (bf16, dense A; B zero-padded to a multiple of 32 in K).
// Same tile shape as gemm_nt_wide_kernel, but the operands are moved HBM/L2 -> LDS by
// global_load_lds_dwordx4 (no staging registers, no ds_write) into a ring of NS stages with
// NS-1 stages in flight (counted s_waitcnt vmcnt, one raw s_barrier per k-step), so the HBM
// latency of the activation stream is covered by three k-steps of MFMA work.
//   stage = [A tile 128 x 32][B tile WBN x 32][scratch]  : unpadded 64-byte rows
//   a DMA wave-instruction fills 1 KB = 16 rows; the 16-byte chunk a lane FETCHES is XOR-permuted
//   (chunk ^ P[(row >> 2) & 3], P = {0,2,3,1}) so that the later ds_read_b128 fragment reads of the
//   linear LDS image are bank-conflict free (swizzle on the source side, LDS stays lane-linear).
// Row overruns are clamped (the rows are never stored); the K tail relies on B's zero padding.
// =========================================================================================
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst))
               : "memory");
}
__device__ __forceinline__ int swzP(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

// ring depth of the DMA kernel: 4 stages when they fit (150 KB for the one-per-CU tile, 78 KB when two workgroups share a CU)
constexpr int dma_ring_stages(int stage_bytes, int wm) { return 4 * stage_bytes <= (wm == 4 ? 150 : 78) * 1024 ? 4 : 3; }

// phase stamps of the LDS-DMA NT kernel (NR_NT_ABLATE bit 64; measurement only): 8 x s_memrealtime (100 MHz) per workgroup
__device__ unsigned long long g_nt_trace[8 * 4096];
__device__ __forceinline__ void nt_stamp(int abl, int slot) {
  if ((abl & 64) && threadIdx.x == 0 && blockIdx.x < 4096) g_nt_trace[blockIdx.x * 8 + slot] = wall_clock64();
}

template <int EPI, int NT16, bool PK, int WM>
__global__ __launch_bounds__(128 * WM) void gemm_nt_dma_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B,
                                                           int ldb, int M, int Ntot, int K, EpiArgs ep, int nchunks_abl) {
  const int nchunks = nchunks_abl & 0xffff, abl = nchunks_abl >> 16;   // abl: NR_NT_ABLATE phase ablation (measurement only)
  nt_stamp(abl, 0);
  if ((abl & 64) && threadIdx.x == 0 && blockIdx.x < 4096) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    g_nt_trace[blockIdx.x * 8 + 7] = hwid;
  }
  // WM row groups of 64 rows x 2 column halves = 2*WM waves.  WM = 4: 256-row tile, one workgroup per CU.
  // WM = 2: 128-row tile whose ring fits twice into the CU's LDS, so one workgroup's pipeline fill and epilogue
  // overlap the other's MFMA loop.
  constexpr int DBM = 64 * WM, NW = 2 * WM, WTHR = 128 * WM;
  constexpr int WBN = NT16 * 16;
  constexpr int NP = DBM / 16 + NT16;            // 1-KB DMA pieces per stage (A rows, then B rows)
  constexpr int PB = NP / NW, PX = NP % NW;      // a wave issues PB (+1 if wid < PX) pieces per stage
  constexpr int STAGE = NP * 1024;               // bytes per ring stage
  constexpr int NS = dma_ring_stages(STAGE, WM);
  constexpr int SCW = WBN + 4;
  constexpr int AB = DBM * 64;                   // byte offset of the B tile inside a stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bool det = nr_fix_on();          // deterministic mode: fixed-point accumulation of the outputs (nr_common.h)
  float* sC = reinterpret_cast<float*>(smem);
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int bid = blockIdx.x, per = 8 * nchunks;
  const int mt = (bid / per) * 8 + (bid & 7), nch = (bid >> 3) % nchunks;
  const int m0 = mt * DBM, nbase = nch * WBN;
  if (m0 >= M) return;
  const int N = min(WBN, Ntot - nbase);
  constexpr int HN = (NT16 + 1) / 2;             // column tiles per wave
  const int tid = threadIdx.x, lane = tid & 63;
  if (EPI == EPI_POOLBWD && PK && ep.seq_nz != nullptr) {
    // every sequence of this tile has a zero pooled gradient (and with it zero dpre rows): the output rows are zeros
    const int t_first = m0 / ep.L, t_last = (min(m0 + DBM, M) - 1) / ep.L;
    bool live = false;
    for (int t = t_first + tid; t <= t_last; t += WTHR) live |= ep.seq_nz[t] != 0;
    if (!__syncthreads_or(live)) {
      const int cpr = N / 8;                                    // N % 8 == 0 on the packed path
      if ((N & 7) == 0 && (ep.ldc & 7) == 0 && (nbase & 7) == 0) {
        for (int u = tid; u < DBM * cpr; u += WTHR) {
          const int row = u / cpr, c = (u - row * cpr) * 8;
          if (m0 + row < M) *reinterpret_cast<uint4*>((bf16_t*)ep.C + (size_t)(m0 + row) * ep.ldc + nbase + c) = make_uint4(0, 0, 0, 0);
        }
        return;
      }
    }
  }
  if ((EPI == EPI_STORE || EPI == EPI_STORE_TANH) && ep.seq_nz != nullptr && ep.row_count == nullptr) {
    // every sequence of this row tile is flagged "output not needed": nothing is computed, the rows stay unwritten
    const int t_first = m0 / ep.L, t_last = (min(m0 + DBM, M) - 1) / ep.L;
    bool live = false;
    for (int t = t_first + tid; t <= t_last; t += WTHR) live |= ep.seq_nz[t] != 0;
    if (!__syncthreads_or(live)) return;
  }
  const bool compact = (EPI == EPI_SCATTER || EPI == EPI_STORE) && ep.row_count != nullptr;
  if (compact) {
    M = *ep.row_count;                             // rows that survive the compaction (device side, no host sync)
    if (m0 >= M) return;
  } else if (EPI == EPI_SCATTER) {
    // A tile whose token ids are all 0 (padding: zero-padded title tails, empty history slots) scatters nothing
    // (padding_idx row, src/model/NRMS.py:71): leave before loading anything.  ~30 % of the tiles of a MIND-shaped batch.
    bool live = false;
    for (int r = tid; r < DBM && m0 + r < M; r += WTHR) live |= ep.ids[(size_t)(m0 + r) * ep.ids_stride] != 0;
    if (!__syncthreads_or(live)) return;
  }
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wid >> 1, wn = wid & 1;
  const bool extra = wid < PX;                   // wave-uniform
  const int pfirst = wid * PB + min(wid, PX);    // first piece of this wave

  // this wave's pieces: per-lane source row pointer; A pieces clamp the K tail, B relies on zero padding
  const bf16_t* src[PB + 1];
  bool isA[PB + 1];
  const int prow = lane >> 2, c8 = ((lane & 3) ^ swzP(prow)) * 8;   // (16p + prow) >> 2 & 3 == prow >> 2 & 3
#pragma unroll
  for (int t = 0; t < PB + 1; ++t) {
    const int p = pfirst + t;
    if (p < DBM / 16) {
      int arow = min(m0 + 16 * p + prow, M - 1);
      if (compact) arow = (ep.a_idx != nullptr ? ep.a_idx : ep.row_idx)[arow];   // compacted row -> row of A (a_idx: A is stored compactly)
      if (ep.a_gap > 0) arow += arow / ep.a_gap;   // token rows with a zero row between titles (RowSrc::gap)
      src[t] = A + (size_t)arow * lda + c8;
      isA[t] = true;
    } else {
      src[t] = B + (size_t)min(nbase + 16 * (p - DBM / 16) + prow, Ntot - 1) * ldb + c8;
      isA[t] = false;
    }
  }
  auto issue = [&](int stage, int k0) {
#pragma unroll
    for (int t = 0; t < PB + 1; ++t) {
      if (t < PB || extra) {
        const int kk = isA[t] ? min(k0, K - 8 - c8) : k0;   // K tail of A: re-read a valid chunk (B is zero there)
        if (!(abl & 4)) dma16(src[t] + kk, lds0 + stage * STAGE + (pfirst + t) * 1024);
      }
    }
  };

  // wave tile = 64 rows x HN*16 columns: 4 A fragments are reused by every B fragment (4 MFMAs per LDS read)
  f32x4 acc[4][HN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < HN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment byte offsets inside a stage (loop invariant)
  const int fr = lane & 15, g = lane >> 4;
  int offA[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wm * 64 + i * 16 + fr;
    offA[i] = r * 64 + ((g ^ swzP(r)) << 4);
  }
  // B tile row (wn*HN + j)*16 + fr : + j*1024 per column tile (16 rows x 64 B; the swizzle has period 16 rows)
  const int offB = AB + wn * HN * 1024 + fr * 64 + ((g ^ swzP(fr)) << 4);

  // POOLBWD: the pooled-gradient rows g[title] and alpha of this tile are DMA'd into LDS behind the ring before the
  // main loop starts (they are the oldest loads, so every later counted wait covers them); the epilogue then reads them
  // from LDS instead of issuing ~30 dependent global loads per lane with nothing left to hide their latency.
  constexpr int GT = 12, GBYTES = ((GT * WBN * 4 + 1023) / 1024) * 1024;        // up to 12 titles per tile (L >= 24)
  const bool g_lds = (EPI == EPI_POOLBWD) && PK && ep.L >= DBM / (GT - 2) && (ep.ldg % 4 == 0) && (nbase % 4 == 0) &&
                     (((uintptr_t)ep.G & 15) == 0) && (((uintptr_t)ep.rowscale & 15) == 0) && M >= 4;
  const int t0 = (EPI == EPI_POOLBWD) ? m0 / max(ep.L, 1) : 0;
  // STORE / STORE_TANH: the bias of this column chunk, same idea (one DMA piece)
  // (one piece holds 256 floats: the 320-column chunk reads its bias from global memory -- found by tests/test_gpu_gemm_wreg.py)
  const bool b_lds = (EPI == EPI_STORE || EPI == EPI_STORE_TANH) && PK && WBN <= 256 && ep.bias != nullptr && (nbase % 4 == 0) && N >= 4 &&
                     (((uintptr_t)ep.bias & 15) == 0);
  if (b_lds && wid == NW - 1) dma16(ep.bias + nbase + min(4 * lane, ((N - 4) / 4) * 4), lds0 + NS * STAGE);
  // compacted STORE: the original row numbers of this tile (output rows are scattered back)
  const bool r_lds = (EPI == EPI_STORE) && PK && compact && M >= 4 && (((uintptr_t)ep.row_idx & 15) == 0) && m0 + 255 < M;
  if (r_lds && wid == NW - 2) dma16(ep.row_idx + m0 + 4 * lane, lds0 + NS * STAGE + 1024);
  // SCATTER: the 256 token ids of this tile
  const int32_t* tile_ids = compact ? ep.row_ids : ep.ids;
  const bool i_lds = (EPI == EPI_SCATTER) && !PK && (compact || ep.ids_stride == 1) && M >= 4 && (((uintptr_t)tile_ids & 15) == 0) &&
                     (!compact || (((uintptr_t)ep.row_idx & 15) == 0));
  if (i_lds && wid == NW - 1) {
    dma16(tile_ids + min(m0 + 4 * lane, M - 4), lds0 + NS * STAGE);
    if (compact) dma16(ep.row_idx + min(m0 + 4 * lane, M - 4), lds0 + NS * STAGE + 1024);
  }
  if (g_lds) {
    const int tlast = (M - 1) / ep.L;
    for (int p = wid; p < GBYTES / 1024; p += NW) {
      const int u = 64 * p + lane, t = u / (WBN / 4), c4 = u - t * (WBN / 4);
      const float* gsrc = ep.G + (size_t)min(t0 + t, tlast) * ep.ldg + nbase + min(4 * c4, max(N - 4, 0));
      dma16(gsrc, lds0 + NS * STAGE + p * 1024);
    }
    if (wid == NW - 1) dma16(ep.rowscale + min(m0 + 4 * lane, M - 4), lds0 + NS * STAGE + GBYTES);   // alpha of rows m0 .. m0+255
  }
  const int nk = (K + BK - 1) / BK;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) issue(s, s * BK);
  // wait until at most `stages` of this wave's most recent stage issues are still in flight
  auto wait_stages = [&](int stages) {
    if (extra) {
      if (stages >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (PB + 1)) : "memory");
      else if (stages == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PB + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (stages >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PB) : "memory");
      else if (stages == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  constexpr bool PF = (NS == 4) && (HN <= 7);    // fragment double buffering fits the register file
  if constexpr (PF) {
    // The fragments of k-step kt+1 are read from LDS while the MFMAs of k-step kt run, so the LDS latency and the
    // burst of 8 waves reading at once no longer sit between the barrier and the first MFMA.  Costs one stage of
    // DMA lead (stage kt+1 must have landed at step kt).
    bf16x8 fa[2][4], fb[2][HN];
    auto load_frags = [&](int buf, int stage) {
      const char* st = smem + stage * STAGE;
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[buf][i] = *reinterpret_cast<const bf16x8*>(st + offA[i]);
#pragma unroll
      for (int j = 0; j < HN; ++j)
        if (wn * HN + j < NT16) fb[buf][j] = *reinterpret_cast<const bf16x8*>(st + offB + j * 1024);
    };
    auto mfmas = [&](int buf) {
      if (abl & 2) return;
#pragma unroll
      for (int j = 0; j < HN; ++j) {
        if (wn * HN + j < NT16) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            // (weights on the row side: a lane then owns 4 CONSECUTIVE output columns of one row, see the epilogues)
            if (PK || EPI == EPI_SCATTER) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[buf][j], fa[buf][i], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[buf][i], fb[buf][j], acc[i][j], 0, 0, 0);
          }
        }
      }
    };
    auto step = [&](int kt, int cur) {
      // The fragment reads of the previous step are waited for HERE, as a builtin the compiler's wait-count pass can see:
      // otherwise it has to assume they are still pending when the MFMAs below start and puts an lgkmcnt(0) in front of
      // them -- behind the reads this step has just issued, which then are not hidden by the MFMAs at all.
      __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0)
      if (kt + 1 < nk) {
        wait_stages(min(NS - 3, nk - 2 - kt));   // stage kt+1 landed (stages kt+2.. may be in flight)
        __builtin_amdgcn_s_barrier();
        if (kt + NS - 1 < nk) issue((kt + NS - 1) % NS, (kt + NS - 1) * BK);
        load_frags(cur ^ 1, (kt + 1) % NS);
      }
      mfmas(cur);
    };
    nt_stamp(abl, 1);
    wait_stages(min(NS - 2, nk - 1));
    __builtin_amdgcn_s_barrier();
    nt_stamp(abl, 2);
    load_frags(0, 0);
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      step(kt, 0);
      step(kt + 1, 1);
    }
    if (kt < nk) step(kt, 0);
  } else {
  for (int kt = 0; kt < nk; ++kt) {
    const int rem = min(NS - 2, nk - 1 - kt);    // stages issued after stage kt that may stay in flight
    wait_stages(rem);
    __builtin_amdgcn_s_barrier();
    if (kt + NS - 1 < nk) issue((kt + NS - 1) % NS, (kt + NS - 1) * BK);
    const char* st = smem + (kt % NS) * STAGE;
    bf16x8 af[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st + offA[i]);
#pragma unroll
    for (int j = 0; j < HN; ++j) {
      if (wn * HN + j < NT16) {
        const bf16x8 bf = *reinterpret_cast<const bf16x8*>(st + offB + j * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // PK: weights on the MFMA row side -> a lane owns 4 consecutive output columns of one row
          if (PK || EPI == EPI_SCATTER) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, af[i], acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[i][j], 0, 0, 0);
        }
      }
    }
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  nt_stamp(abl, 3);
  if (abl & 8) {
    if (acc[0][0][0] == 12345.678f) ((float*)ep.C)[0] = acc[1][0][1];   // keeps the loop alive
    return;
  }
  if (PK) {
    // packed bf16 epilogue: 128 rows at a time through a bf16 LDS image (8-byte writes), 16-byte row stores
    bf16_t* sCb = reinterpret_cast<bf16_t*>(smem);
    constexpr int SCB = WBN + 8;
#pragma unroll 1
    for (int pass = 0; pass < WM / 2; ++pass) {
      if (pass) __syncthreads();
      if ((wm >> 1) == pass) {
        // POOLBWD: the lane's 4 rows, their alpha and pooled-gradient rows, looked up once (not per column tile)
        float rs4[4] = {0.f, 0.f, 0.f, 0.f};
        const float* grow4[4] = {nullptr, nullptr, nullptr, nullptr};
        const float* sG = reinterpret_cast<const float*>(smem + NS * STAGE);
        const float* sAl = reinterpret_cast<const float*>(smem + NS * STAGE + GBYTES);
        if (EPI == EPI_POOLBWD) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int mloc = pass * 128 + (wm & 1) * 64 + i * 16 + (lane & 15), m = m0 + mloc;
            if (m < M) {
              if (g_lds && m0 + 255 < M) rs4[i] = sAl[mloc]; else rs4[i] = ep.rowscale[m];
              grow4[i] = g_lds ? sG + (m / ep.L - t0) * WBN : ep.G + (size_t)(m / ep.L) * ep.ldg + nbase;
            }
          }
        }
        const bool gvec = g_lds || ((EPI == EPI_POOLBWD) && (ep.ldg % 4 == 0) && (nbase % 4 == 0) && (((uintptr_t)ep.G & 15) == 0));
#pragma unroll
        for (int j = 0; j < HN; ++j) {
          const int jt = wn * HN + j;
          if (jt < NT16) {
            const int nl = jt * 16 + 4 * (lane >> 4);
            f32x4 bvec = (f32x4){0.f, 0.f, 0.f, 0.f};
            if ((EPI == EPI_STORE || EPI == EPI_STORE_TANH) && ep.bias != nullptr) {
              if (b_lds && nl + 4 <= N) {
                bvec = *reinterpret_cast<const f32x4*>(sG + nl);
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  if (nl + r < N) bvec[r] = ep.bias[nbase + nl + r];
              }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int ml = (wm & 1) * 64 + i * 16 + (lane & 15), m = m0 + pass * 128 + ml;
              f32x4 v = acc[i][j] + bvec;
              if (EPI == EPI_STORE_TANH) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
              }
              if (EPI == EPI_POOLBWD && grow4[i] != nullptr) {
                if (gvec && nl + 4 <= N) {
                  const f32x4 gq = *reinterpret_cast<const f32x4*>(grow4[i] + nl);
#pragma unroll
                  for (int r = 0; r < 4; ++r) v[r] += rs4[i] * gq[r];
                } else {
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                    if (nl + r < N) v[r] += rs4[i] * grow4[i][nl + r];
                }
              }
              const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
              *reinterpret_cast<bf16x4*>(sCb + ml * SCB + nl) = o;
            }
          }
        }
      }
      __syncthreads();
      nt_stamp(abl, 4 + pass);
      for (int u = tid; u < 128 * (WBN / 8); u += WTHR) {
        const int row = u / (WBN / 8), c0 = (u % (WBN / 8)) * 8;
        const int m = m0 + pass * 128 + row;
        if (m < M && c0 < N) {
          const int mout = !compact ? m : (r_lds ? reinterpret_cast<const int*>(smem + NS * STAGE + 1024)[pass * 128 + row] : ep.row_idx[m]);
          bf16_t* dst = (bf16_t*)ep.C + (size_t)mout * ep.ldc + nbase + c0;
          if (abl & 1) {
            const uint4 q = *reinterpret_cast<const uint4*>(sCb + row * SCB + c0);
            if (q.x == 0x12345678u && q.y == 0x9abcdef0u) *reinterpret_cast<uint4*>(dst) = q;
          } else if (c0 + 8 <= N && (ep.ldc % 8) == 0 && (nbase % 8) == 0) {
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(sCb + row * SCB + c0);
          } else {
            for (int e = 0; e < 8 && c0 + e < N; ++e) dst[e] = sCb[row * SCB + c0 + e];
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    nt_stamp(abl, 6);
    return;
  }
  if constexpr (EPI == EPI_SCATTER) {
    // Table-gradient scatter, WAVE-LOCAL: every wave stages 16 of its 64 rows x its HN*16 columns in its own block of the
    // (drained) ring -- 16-byte writes, a lane owns 4 consecutive columns of a row -- and walks those rows itself, lanes on
    // adjacent column pairs.  No workgroup barrier in the whole epilogue and all 8 waves busy (the shared 64-row tile had 2
    // waves writing 160 scalar words per lane while 6 waited, 4 times per tile: 0.115 of this kernel's 0.43 ms).  Rows that
    // follow each other with the SAME token id -- the caller hands the rows over sorted by id -- are summed in registers,
    // now across all 64 rows of the wave, and leave as one atomic row (memory-side float atomics: ~1.3 TB/s chip-wide).
    // One dropout hash serves the lane's two columns (the pair hash of nr_keep) when the pair starts on an even element.
    constexpr int WC = HN * 16, SW = WC + 4, NC2 = (WC + 127) / 128;
    static_assert((size_t)NW * 16 * SW * sizeof(float) <= (size_t)NS * STAGE, "scatter blocks must fit the ring");
    float* sW = sC + wid * 16 * SW;
    const int cw0 = wn * WC;                           // first column of this wave inside the chunk
    auto scatter = [&](auto det_tag) {
    constexpr bool DET = decltype(det_tag)::value;     // two bodies: the fixed-point state of the deterministic mode costs the default one its registers
    float run[NC2][2];
    long long runq[NC2][2];                            // deterministic mode: the run is summed in fixed point (order independent)
#pragma unroll
    for (int u = 0; u < NC2; ++u) { run[u][0] = run[u][1] = 0.f; runq[u][0] = runq[u][1] = 0; }
    int run_id = 0;
    auto flush = [&]() {
      if (run_id != 0) {                               // padding_idx row receives no gradient
        float* dst = (float*)ep.C + (size_t)run_id * ep.ldc + nbase + cw0;
#pragma unroll
        for (int u = 0; u < NC2; ++u)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int c = 128 * u + 2 * lane + e;
            if (c < WC && cw0 + c < N && nbase + cw0 + c < ep.Dtrue) {
              if (DET) { if (runq[u][e] != 0) nr_accum_fix(dst + c, runq[u][e]); }
              else if (run[u][e] != 0.f && !((abl & 1) && run[u][e] != 1234.5f)) atomicAdd(dst + c, run[u][e]);
            }
          }
      }
#pragma unroll
      for (int u = 0; u < NC2; ++u) { run[u][0] = run[u][1] = 0.f; runq[u][0] = runq[u][1] = 0; }
    };
    const bool from_lds = i_lds && m0 + DBM - 1 < M;
    const bool pair_ok = ((ep.Dtrue | nbase | cw0) & 1) == 0;     // every (row, even column) of this wave is an even element index
    const bool dropping = ep.drop.thresh != 0 && !(abl & 16);
    constexpr int RB = DET ? 2 : 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
      for (int j = 0; j < HN; ++j)
        if (wn * HN + j < NT16) *reinterpret_cast<f32x4*>(sW + (lane & 15) * SW + j * 16 + (lane >> 4) * 4) = acc[p][j];
      // (the wave's own LDS operations execute in order: its reads below see these writes, the next pass's writes come after)
      // token id and original row number of the pass's 16 rows: lane q holds row q's, the row loop reads them back as scalars
      const int mb = wm * 64 + p * 16, ml = mb + (lane & 15), mg = min(m0 + ml, M - 1);
      int idv = from_lds ? reinterpret_cast<const int*>(smem + NS * STAGE)[ml] : (compact ? ep.row_ids[mg] : ep.ids[(size_t)mg * ep.ids_stride]);
      const int mov = !compact ? mg : (from_lds ? reinterpret_cast<const int*>(smem + NS * STAGE + 1024)[ml] : ep.row_idx[mg]);
      if (m0 + ml >= M) idv = 0;                       // rows behind the end behave like padding rows
      // RB rows at a time: their RB x NC2 LDS reads and dropout hashes are independent of each other and of the run logic
      // (one row after the other, each waiting for its own id, values and hash, was a chain of latencies 64 rows long)
#pragma unroll 1
      for (int q0 = 0; q0 < 16; q0 += RB) {
        float xs[RB][NC2][2];
#pragma unroll
        for (int t = 0; t < RB; ++t)
#pragma unroll
          for (int u = 0; u < NC2; ++u) {
            const int c = 128 * u + 2 * lane;
            float2 xv = make_float2(0.f, 0.f);
            if (c < WC) xv = *reinterpret_cast<const float2*>(sW + (q0 + t) * SW + c);
            xs[t][u][0] = xv.x; xs[t][u][1] = xv.y;
          }
        if (dropping) {
#pragma unroll
          for (int t = 0; t < RB; ++t) {
            const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane(mov, q0 + t) * (uint32_t)ep.Dtrue + (uint32_t)(nbase + cw0);
#pragma unroll
            for (int u = 0; u < NC2; ++u) {
              const int c = 128 * u + 2 * lane;
              if (pair_ok) {
                const uint32_t h = nr_pair_hash(ep.drop.key, e0 + c);
                xs[t][u][0] = (h & 0xffffu) >= ep.drop.thresh ? xs[t][u][0] * ep.drop.scale : 0.f;
                xs[t][u][1] = (h >> 16) >= ep.drop.thresh ? xs[t][u][1] * ep.drop.scale : 0.f;
              } else {
                xs[t][u][0] = nr_keep(ep.drop.key, e0 + c, ep.drop.thresh) ? xs[t][u][0] * ep.drop.scale : 0.f;
                xs[t][u][1] = nr_keep(ep.drop.key, e0 + c + 1, ep.drop.thresh) ? xs[t][u][1] * ep.drop.scale : 0.f;
              }
            }
          }
        }
#pragma unroll
        for (int t = 0; t < RB; ++t) {
          const int id = __builtin_amdgcn_readlane(idv, q0 + t);       // wave-uniform: scalar compare and branch
          if (id != run_id) {
            flush();
            run_id = id;
          }
          if (id != 0) {
#pragma unroll
            for (int u = 0; u < NC2; ++u) {
              if (DET) { runq[u][0] += nr_to_fix(xs[t][u][0]); runq[u][1] += nr_to_fix(xs[t][u][1]); }
              else { run[u][0] += xs[t][u][0]; run[u][1] += xs[t][u][1]; }
            }
          }
        }
      }
    }
    flush();
    };
    if (det) scatter(std::true_type{}); else scatter(std::false_type{});
    return;
  }
  // epilogue: 32 rows at a time through the fp32 LDS tile (pass p = rows 32p..32p+31 = wave row wm = p>>1, tiles 2(p&1), +1)
#pragma unroll 1
  for (int pass = 0; pass < 2 * WM; ++pass) {
    if (pass) __syncthreads();
    if (wm == (pass >> 1)) {
#pragma unroll
      for (int j = 0; j < HN; ++j) {
        const int jt = wn * HN + j;
        if (jt < NT16) {
          const int col = jt * 16 + (lane & 15);
          float bv = 0.f;
          if ((EPI == EPI_STORE || EPI == EPI_STORE_TANH) && ep.bias != nullptr && col < N) bv = ep.bias[nbase + col];
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              // static accumulator index: both candidate tiles are written from a select
              float v = ((pass & 1) ? acc[2 + i2][j][r] : acc[i2][j][r]) + bv;
              if (EPI == EPI_STORE_TANH) v = tanhf(v);
              sC[(i2 * 16 + (lane >> 4) * 4 + r) * SCW + col] = v;
            }
        }
      }
    }
    __syncthreads();
    if (EPI == EPI_SCATTER) {
      // table-gradient scatter: one wave per row, lanes on CONSECUTIVE floats, so an atomic instruction covers whole
      // 128-byte lines of the destination row (float4-per-lane ownership spread every instruction over 8 lines)
      for (int rr = wid; rr < 32; rr += NW) {
        const int m = m0 + pass * 32 + rr;
        if (m >= M) continue;
        const int mloc = pass * 32 + rr;
        const bool from_lds = i_lds && m0 + 255 < M;
        const int id = from_lds ? reinterpret_cast<const int*>(smem + NS * STAGE)[mloc]
                                : (compact ? ep.row_ids[m] : ep.ids[(size_t)m * ep.ids_stride]);
        if (id == 0) continue;                       // padding_idx row receives no gradient
        float* dst = (float*)ep.C + (size_t)id * ep.ldc + nbase;
        // the dropout counter follows the ORIGINAL row
        const int morig = !compact ? m : (from_lds ? reinterpret_cast<const int*>(smem + NS * STAGE + 1024)[mloc] : ep.row_idx[m]);
        const uint32_t e0 = (uint32_t)morig * (uint32_t)ep.Dtrue + (uint32_t)nbase;
        for (int c = lane; c < N; c += 64) {
          float x = sC[rr * SCW + c];
          if (ep.drop.thresh) x = nr_keep(ep.drop.key, e0 + c, ep.drop.thresh) ? x * ep.drop.scale : 0.f;
          if (x != 0.f && nbase + c < ep.Dtrue) nr_accum(dst + c, x, det);
        }
      }
    } else {
      for (int u = tid; u < 32 * (WBN / 4); u += WTHR) {
        const int row = u / (WBN / 4), c0 = (u % (WBN / 4)) * 4;
        const int m = m0 + pass * 32 + row;
        if (m < M && c0 < N) emit4<EPI>(ep, m, nbase + c0, Ntot, *reinterpret_cast<const f32x4*>(sC + row * SCW + c0));
      }
    }
  }
}

}  // namespace
extern "C" int nr_debug_nt_trace(unsigned long long* out, int n) {
  if (out == nullptr || n <= 0) return NR_ERR_ARG;
  NR_CHECK_HIP(hipDeviceSynchronize());
  NR_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nt_trace), sizeof(unsigned long long) * (size_t)(n < 8 * 4096 ? n : 8 * 4096)));
  return NR_OK;
}
namespace {
// =========================================================================================
// NT with the WEIGHTS IN REGISTERS (bf16 in, bf16 out; K <= 32 * KS; EPI_STORE / EPI_STORE_TANH).
// For the skinny-K projections (QKV: N = 1200, K = 304) the tile kernels above spend their time
// on everything but MFMAs: per 256 x 208 tile 10 k-steps of barrier + DMA issue + 11 fragment
// reads in front of 28 MFMAs, then a fill and a drain that overlap nothing (tools/nt_trace.py:
// 20 us per tile, 4.5 us of it MFMA).  Here a wave keeps its slice of the weight matrix --
// TPW column tiles x the whole K -- in VGPRs for the life of the (persistent) workgroup and only
// the activation rows move: 16-row stages through an LDS ring filled by LDS-DMA, ONE barrier per
// stage, per stage and wave KS fragment reads for KS * TPW MFMAs, and an epilogue that goes from
// the accumulators straight to global memory (v_permlane16_swap pairs two column tiles into
// 16-byte stores) while the partner wave of the SIMD keeps the MFMA pipe busy.
//   grid      8 XCDs x (ngroups column groups x Q) workgroups of 8 waves; group g owns columns
//             [g * 128 * TPW, ...), wave w of it tiles (g * 8 + w) * TPW ...
//   rows      workgroup (xcd, q, g) walks the 16-row blocks (q * 8 + xcd) + k * 8 * Q; the ngroups
//             workgroups of one (xcd, q) walk the same blocks at the same pace, so A comes from HBM once
//   COMPACT   rows are the compacted live rows (ep.row_count / ep.row_idx): the row numbers of a
//             block ride along with its stage as one more DMA piece (own rows for the output, the
//             rows of the block NS-1 steps ahead for the DMA issue)
// Vector-memory bookkeeping is static: every wave issues exactly PW LDS-DMAs and S stores per step
// (surplus DMAs and the stores of masked lanes go to dump areas), so "stage k has landed" is the
// immediate s_waitcnt vmcnt((NS-2) * PW + (NS-1) * S).
// =========================================================================================
__device__ uint4 g_nt_dump[512];                 // masked lanes of the always-issued epilogue stores land here

__device__ __forceinline__ void wait_vmcnt_le(int n) {   // n is wave-uniform
  switch (n) {
#define NR_VMC(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
    NR_VMC(0) NR_VMC(1) NR_VMC(2) NR_VMC(3) NR_VMC(4) NR_VMC(5) NR_VMC(6) NR_VMC(7) NR_VMC(8) NR_VMC(9)
    NR_VMC(10) NR_VMC(11) NR_VMC(12) NR_VMC(13) NR_VMC(14) NR_VMC(15) NR_VMC(16) NR_VMC(17) NR_VMC(18) NR_VMC(19)
    NR_VMC(20) NR_VMC(21) NR_VMC(22) NR_VMC(23) NR_VMC(24) NR_VMC(25) NR_VMC(26) NR_VMC(27) NR_VMC(28) NR_VMC(29)
    NR_VMC(30) NR_VMC(31) NR_VMC(32) NR_VMC(33) NR_VMC(34) NR_VMC(35) NR_VMC(36) NR_VMC(37) NR_VMC(38) NR_VMC(39)
    NR_VMC(40) NR_VMC(41) NR_VMC(42) NR_VMC(43) NR_VMC(44) NR_VMC(45) NR_VMC(46) NR_VMC(47) NR_VMC(48) NR_VMC(49)
    NR_VMC(50) NR_VMC(51) NR_VMC(52) NR_VMC(53) NR_VMC(54) NR_VMC(55) NR_VMC(56) NR_VMC(57) NR_VMC(58) NR_VMC(59)
    NR_VMC(60) NR_VMC(61) NR_VMC(62)
#undef NR_VMC
    default: break;                               // 63 and more: the counter cannot exceed 63, nothing to wait for
  }
}

// MODE of the weights-in-registers kernel: how a workgroup finds its row blocks
enum { WREG_DENSE = 0,    // blocks b0 + k * stride of the dense rows
       WREG_COMPACT = 1,  // the same over the compacted live rows (ep.row_count / ep.row_idx); row numbers ride with the stages
       WREG_LIST = 2,     // dense rows, but only blocks that touch a sequence flagged in ep.seq_nz (list built in the prologue)
       WREG_COMPACT_DENSE = 3 };  // COMPACT whose A operand is itself stored in live-list order (A row k = live row k): the row
                                  // numbers only scatter the output rows

template <int EPI, int TPW, int KS, int RT, int MODE, bool SEQ = false>
struct WregCfg {
  static constexpr int NW = 8, R = 16 * RT;
  static constexpr int GCOLS = NW * TPW * 16;                         // columns of a group
  static constexpr int APIECES = RT * KS;                             // 1-KB pieces: 16 rows x 64 B per (row tile, k-step)
  static constexpr int RIDER = (MODE == WREG_COMPACT || MODE == WREG_COMPACT_DENSE) ? 1 : 0;   // row numbers
  static constexpr int GT = 3, GP = (GCOLS * 4 + 1023) / 1024;        // POOLBWD: pooled-gradient rows of up to GT sequences
  static constexpr int PBW = EPI == EPI_POOLBWD ? 1 + GT * GP : 0;    //          alpha of the block's rows + those G rows
  static constexpr int PIECES = APIECES + RIDER + PBW;
  static constexpr int PW = (PIECES + NW - 1) / NW;                   // DMAs per wave and stage
  static constexpr int STAGE = PIECES * 1024;
  static constexpr int NS0 = (132 * 1024) / STAGE;
  static constexpr int NS = NS0 > 8 ? 8 : NS0;                        // ring depth
  static constexpr int S = RT * ((TPW + 1) / 2);                      // stores per wave and step
  static constexpr int DUMP = NS * STAGE, BIAS = DUMP + NW * 1024, LIST = BIAS + GCOLS * 4;
  static_assert(NS >= 3, "stage too large for the LDS ring");
  static_assert((NS - 2) * PW + (NS - 1) * S <= 62, "vmcnt is a 6-bit counter");
};

// SEQ: the RT row tiles of a stage are computed one after the other through ONE set of accumulators (MFMAs, epilogue, MFMAs,
// epilogue): a 32-row stage -- half the barriers and stage issues per row -- without the 20 extra accumulator registers
// that the QKV instantiation (200 registers of weights) does not have.
template <int EPI, int TPW, int KS, int RT, int MODE, bool SEQ = false>
__global__ __launch_bounds__(512) void gemm_nt_wreg_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
                                                           int Mmax, int Ntot, int K, EpiArgs ep, int ngroups) {
  using Cfg = WregCfg<EPI, TPW, KS, RT, MODE, SEQ>;
  constexpr int NW = Cfg::NW, R = Cfg::R, NS = Cfg::NS, PW = Cfg::PW, STAGE = Cfg::STAGE, S = Cfg::S, GCOLS = Cfg::GCOLS;
  constexpr int APIECES = Cfg::APIECES, GT = Cfg::GT, GP = Cfg::GP;
  constexpr bool ADENSE = MODE == WREG_COMPACT_DENSE, COMPACT = MODE == WREG_COMPACT || ADENSE, LISTED = MODE == WREG_LIST, PB = EPI == EPI_POOLBWD;
  constexpr int P_RIDER = APIECES;                       // piece index of the row-number rider (COMPACT)
  constexpr int P_ALPHA = APIECES + Cfg::RIDER;          // POOLBWD: alpha piece, then GT * GP pieces of G rows
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int prow = lane >> 2, c8 = ((lane & 3) ^ swzP(prow)) * 8;

  int M = Mmax;
  if (COMPACT) M = min(*ep.row_count, Mmax);
  const int nblk = (M + R - 1) / R;
  const int x = blockIdx.x & 7, j = blockIdx.x >> 3, Q = (gridDim.x >> 3) / ngroups;
  const int grp = j % ngroups, q = j / ngroups;
  const int b0 = q * 8 + x, bstride = 8 * Q;
  int nsteps = b0 < nblk ? (nblk - b0 + bstride - 1) / bstride : 0;
  if (nsteps == 0) return;
  const int gcol0 = grp * GCOLS, wcol0 = gcol0 + wid * TPW * 16;
  int* sList = reinterpret_cast<int*>(smem + Cfg::LIST);
  if (LISTED) {
    // ordered list of this workgroup's blocks that touch a flagged sequence
    __shared__ int sCnt[NW + 1];
    const int cand = nsteps;
    int run = 0;
    for (int base = 0; base < cand; base += 512) {
      const int c = base + tid, b = b0 + c * bstride;
      bool live = false;
      if (c < cand) {
        const int t0 = (b * R) / ep.L, t1 = (min(b * R + R, M) - 1) / ep.L;
        for (int t = t0; t <= t1; ++t) live |= ep.seq_nz[t] != 0;
      }
      const uint64_t bal = __ballot(live);
      if (lane == 0) sCnt[wid] = __popcll(bal);
      __syncthreads();
      int before = 0, total = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const int cw = sCnt[w];
        before += w < wid ? cw : 0;
        total += cw;
      }
      if (live) sList[run + before + __popcll(bal & ((1ull << lane) - 1ull))] = b;
      run += total;
      __syncthreads();
    }
    nsteps = run;
    if (nsteps == 0) return;                        // (POOLBWD: the zeros of such blocks come from zero_dead_blocks_kernel)
  }
  auto blk = [&](int k) {
    if (LISTED) return k < nsteps ? sList[k] : 0;
    return b0 + k * bstride;
  };

  // ---- this wave's slice of the weights: TPW column tiles x K, resident in registers
  bf16x8 bfr[TPW][KS];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int n = min(wcol0 + t * 16 + fr, Ntot - 1);
    const bf16_t* brow = B + (size_t)n * ldb + 8 * g;
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) bfr[t][s2] = *reinterpret_cast<const bf16x8*>(brow + 32 * s2);
  }
  // bias of the group's columns (zeros without one): the accumulators START from it
  float* sBias = reinterpret_cast<float*>(smem + Cfg::BIAS);
  for (int c = tid; c < GCOLS; c += 512) sBias[c] = (!PB && ep.bias != nullptr && gcol0 + c < Ntot) ? ep.bias[gcol0 + c] : 0.f;
  // the compiler must not carry "weights still loading" into the loop (it would wait with vmcnt(0) there every step)
  __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
  if (LISTED) __syncthreads();                   // the list is complete

  // ---- stage issue: every wave exactly PW DMAs (surplus ones into its dump piece)
  const uint32_t dump_lds = lds0 + Cfg::DUMP + wid * 1024;
  auto issue_stage = [&](int kk, const int (&arow)[RT]) {
    const uint32_t slot = lds0 + (kk % NS) * STAGE;
    const bool real = kk < nsteps;
    const int bk = real ? blk(kk) : 0;
    const bf16_t* rowp[RT];                        // one 64-bit row address per row tile and stage, not per DMA
#pragma unroll
    for (int i = 0; i < RT; ++i) rowp[i] = A + (size_t)arow[i] * lda + c8;
#pragma unroll
    for (int u = 0; u < PW; ++u) {
      const int p = wid + NW * u;                  // wave-uniform
      if (real && p < APIECES) {
        const int i = p / KS, s2 = p - i * KS;
        const bf16_t* rp = rowp[0];
#pragma unroll
        for (int ii = 1; ii < RT; ++ii) rp = i == ii ? rowp[ii] : rp;
        // (K tail: the last k-step may reach past K; the chunk is clamped back -- B is zero there, see nr_launch_gemm_nt)
        dma16(rp + min(32 * s2, K - 8 - c8), slot + p * 1024);
      } else if (COMPACT && real && p == P_RIDER) {
        // rider: the row numbers of this block (R ints), then those of the block NS-1 steps ahead (R ints)
        const int half = lane / (R / 4), bb = half == 0 ? bk : blk(kk + NS - 1);
        const int e = min(bb * R + 4 * (lane % (R / 4)), Mmax - 4);
        dma16(ep.row_idx + (lane < R / 2 ? e : 0), slot + p * 1024);
      } else if (PB && real && p == P_ALPHA) {
        dma16(ep.rowscale + min(bk * R + 4 * (lane % (R / 4)), Mmax - 4), slot + p * 1024);
      } else if (PB && real && p > P_ALPHA && p < P_ALPHA + 1 + GT * GP) {
        // pooled-gradient row of sequence t0 + tt, columns of this group: GP pieces of 256 floats
        const int u2 = p - P_ALPHA - 1, tt = u2 / GP, part = u2 - tt * GP;
        const int t = min((bk * R) / ep.L + tt, (Mmax - 1) / ep.L);
        const int col = min(gcol0 + part * 256 + 4 * lane, Ntot - 4);
        dma16(ep.G + (size_t)t * ep.ldg + col, slot + p * 1024);
      } else {
        dma16(A + c8, dump_lds);
      }
    }
  };
  // A rows this lane fetches for block kk (prologue: straight from global memory; steady state: from the rider)
  auto arows_global = [&](int kk, int (&arow)[RT]) {
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int row = blk(kk) * R + 16 * i + prow;
      arow[i] = (kk >= nsteps || row >= M) ? 0 : ((COMPACT && !ADENSE) ? ep.row_idx[row] : row);
    }
  };
#pragma unroll 1
  for (int kk = 0; kk < NS - 1; ++kk) {
    int ar[RT];
    arows_global(kk, ar);
    issue_stage(kk, ar);
  }

  const int offA = fr * 64 + ((g ^ swzP(fr)) << 4);
  bf16_t* Cb = (bf16_t*)ep.C;
  uint4* dump_g = g_nt_dump + wid * 64 + lane;

  // Tried and dropped (round 3): running the last pass's epilogue of waves 4..7 after the NEXT barrier (accumulators live across
  // it, one extra register) so that a SIMD's two waves alternate MFMA and epilogue phases instead of marching in lockstep --
  // same box A/B at the QKV shape: 0.242 -> 0.280 ms.  The late waves issue their DMAs ~500 cycles after the barrier then.
  constexpr int RP = SEQ ? 1 : RT, NPASS = RT / RP;     // row tiles per pass through the accumulators
  f32x4 acc[RP][TPW];

  // epilogue of pass q of the stage that block b / output rows mo belong to: exactly S / NPASS stores
  auto epilogue = [&](int q, int b, const int (&mo)[RT], const char* st) {
#pragma unroll
    for (int ii = 0; ii < RP; ++ii) {
      const int i = q * RP + ii;
      const int m = b * R + 16 * i + fr;
      const bool rok = m < M;
      bf16_t* crow = Cb + (size_t)mo[i] * ep.ldc;
      float al = 0.f;
      const float* grow = nullptr;
      if (PB) {
        // C = acc + alpha[m] * G[m / L][col]: alpha and the G rows of the block's (at most GT) sequences came with the stage
        const int t0 = (b * R) / ep.L, r0 = m - t0 * ep.L;
        const int tt = (r0 >= ep.L ? 1 : 0) + (r0 >= 2 * ep.L ? 1 : 0);
        al = reinterpret_cast<const float*>(st + P_ALPHA * 1024)[16 * i + fr];
        grow = reinterpret_cast<const float*>(st + (P_ALPHA + 1 + tt * GP) * 1024);
      }
      auto pack4 = [&](f32x4 v, int lcol) {
        if (EPI == EPI_STORE_TANH) {
          // tanh x = 1 - 2 / (e^2x + 1) on v_exp_f32 / v_rcp_f32 (abs. error ~1e-7, the result is rounded to bf16 next):
          // libm's tanhf is ~35 instructions per value, 16 values per lane and step -- it, not memory, bound this kernel
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * v[r]) + 1.f);
        }
        if (PB) {
          const f32x4 gq = *reinterpret_cast<const f32x4*>(grow + lcol);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += al * gq[r];
        }
        const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        return __builtin_bit_cast(uint2, o);
      };
#pragma unroll
      for (int t = 0; t < TPW; t += 2) {
        const int lc = (wid * TPW + t) * 16 + 4 * g;         // this lane's first column inside the group, tile t
        if (t + 1 < TPW) {
          const uint2 xv = pack4(acc[ii][t], lc), yv = pack4(acc[ii][t + 1 < TPW ? t + 1 : t], lc + 16);
          const auto s0 = __builtin_amdgcn_permlane16_swap(xv.x, yv.x, false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(xv.y, yv.y, false, false);
          const int col = wcol0 + (t + (g & 1)) * 16 + (g >> 1) * 8;
          uint4* dst = (rok && col + 8 <= Ntot) ? reinterpret_cast<uint4*>(crow + col) : dump_g;
          *dst = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        } else {
          const uint2 xv = pack4(acc[ii][t], lc);
          const int col = wcol0 + t * 16 + 4 * g;
          uint2* dst = (rok && col + 4 <= Ntot) ? reinterpret_cast<uint2*>(crow + col) : reinterpret_cast<uint2*>(dump_g);
          *dst = xv;
        }
      }
    }
  };

#pragma unroll 1
  for (int k = 0; k < nsteps; ++k) {
    if (k >= NS - 1) {                             // steady state: one immediate
      constexpr int STEADY = (NS - 2) * PW + (NS - 1) * S;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STEADY) : "memory");
    } else {
      wait_vmcnt_le((NS - 2) * PW + k * S);
    }
    __builtin_amdgcn_s_barrier();                  // stage k is in LDS for everyone; everyone is done with stage k-1
    const char* st = smem + (k % NS) * STAGE;
    const int b = blk(k);
    int ar[RT], mout[RT];
    {
      const int bn = blk(k + NS - 1);
      const int* rid = reinterpret_cast<const int*>(st + P_RIDER * 1024);
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const int rown = bn * R + 16 * i + prow;
        const bool ok = k + NS - 1 < nsteps && rown < M;
        ar[i] = !ok ? 0 : ((COMPACT && !ADENSE) ? rid[R + 16 * i + prow] : rown);
        mout[i] = COMPACT ? rid[16 * i + fr] : b * R + 16 * i + fr;
      }
    }
    issue_stage(k + NS - 1, ar);

#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(sBias + (wid * TPW + t) * 16 + 4 * g);
#pragma unroll
        for (int i = 0; i < RP; ++i) acc[i][t] = bv;
      }
      if (wcol0 < Ntot) {                          // wave-uniform: a wave whose tiles are all beyond N only keeps step with the others
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) {
#pragma unroll
          for (int ii = 0; ii < RP; ++ii) {
            const int i = q * RP + ii;
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(st + (i * KS + s2) * 1024 + offA);
#pragma unroll
            for (int t = 0; t < TPW; ++t) acc[ii][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[t][s2], af, acc[ii][t], 0, 0, 0);
          }
        }
      }
      // ---- epilogue: exactly S stores per wave and stage
      epilogue(q, b, mout, st);
    }
  }
}

// POOLBWD with sequence flags: rows of blocks no flagged sequence touches are zeros -- written by this store-only kernel
__global__ __launch_bounds__(256) void zero_dead_blocks_kernel(const int32_t* __restrict__ seq_nz, int L, int M, int R, uint4* __restrict__ C,
                                                               int row_chunks) {
  const int b = blockIdx.x;
  const int t0 = (b * R) / L, t1 = (min(b * R + R, M) - 1) / L;
  bool live = false;
  for (int t = t0; t <= t1; ++t) live |= seq_nz[t] != 0;     // uniform
  if (live) return;
  const int rows = min(R, M - b * R);
  uint4* dst = C + (size_t)b * R * row_chunks;
  for (int u = threadIdx.x; u < rows * row_chunks; u += 256) dst[u] = make_uint4(0, 0, 0, 0);
}

template <int EPI, int TPW, int KS, int RT, int MODE, bool SEQ = false>
int launch_nt_wreg_m(const RowSrc& A, const void* B, int ldb, int M, int N, int K, const EpiArgs& ep, hipStream_t stream) {
  using Cfg = WregCfg<EPI, TPW, KS, RT, MODE, SEQ>;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    NR_CHECK_HIP(hipGetDevice(&dev));
    NR_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cus = n >= 8 ? n : 256;
  }
  const int ngroups = (N + Cfg::GCOLS - 1) / Cfg::GCOLS;
  int Q = (cus / 8) / ngroups;                   // workgroups per XCD and group: one workgroup per CU
  if (Q < 1) Q = 1;
  const int nblk = (M + Cfg::R - 1) / Cfg::R, qmax = (nblk + 7) / 8;
  if (Q > qmax) Q = qmax;
  const int max_steps = (nblk + 8 * Q - 1) / (8 * Q);
  const size_t smem = (size_t)Cfg::LIST + (MODE == WREG_LIST ? (size_t)(max_steps + 8) * sizeof(int) : 0);
  NR_CHECK_ARG(smem <= 160 * 1024, "gemm_nt_wreg: %d row blocks per workgroup do not fit the LDS list", max_steps);
  auto kern = gemm_nt_wreg_kernel<EPI, TPW, KS, RT, MODE, SEQ>;
  NR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  if (EPI == EPI_POOLBWD && MODE == WREG_LIST) {
    hipLaunchKernelGGL(zero_dead_blocks_kernel, dim3(nblk), dim3(256), 0, stream, ep.seq_nz, ep.L, M, Cfg::R, (uint4*)ep.C, ep.ldc / 8);
  }
  hipLaunchKernelGGL(kern, dim3(8 * ngroups * Q), dim3(512), smem, stream, (const bf16_t*)A.base, A.ld, (const bf16_t*)B, ldb, M, N, K, ep,
                     ngroups);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

// =========================================================================================
// Fused additive-attention pooling FORWARD (src/model/model_utils.py:21-30) for sequences of L <= 32 tokens:
//   e = tanh(x W1^T + b1) -> a = exp(e w2 + b2) (* mask) / (sum + 1e-8) -> out = sum_l a_l x_l
// in ONE pass over x.  The unfused path ran the weights-in-registers fc1 GEMM (x in, e out) and then pool_core_fwd
// (x and e in again): 0.14 + 0.16 ms and 0.57 GB of re-reads per step at the news level.
//
// Built on the weights-in-registers GEMM above (8 waves, W1 resident in registers, 2 column tiles per wave, activation rows
// through an LDS ring filled by LDS-DMA), with a stage = ONE SEQUENCE (32 rows from row seq * L: the L real ones and 32 - L
// rows of the next sequence that only ride along), and a software pipeline with one barrier per step k:
//   S0  issue the DMAs of sequence k + NS - 2                              (slot (k - 2) % NS, last read in step k - 1)
//   S1  sequence k - 2: out = sum over the 8 row groups of sRed[(k - 2) % 2]                       -> global (fp32)
//   S2  sequence k - 1: logits = sum of the 8 waves' partial dots sPart[(k - 1) % 2] + b2 -> softmax weights (every wave,
//       32 lanes) -> alpha (global) -> this wave's 4 rows of the weighted sum over the x rows STILL IN the ring -> sRed
//   S3  sequence k    : MFMAs -> tanh -> e (global, bf16: the backward needs it) -> partial dots with w2 -> sPart[k % 2]
// Vector-memory bookkeeping is static as above: every wave issues PW DMAs + S + 2 stores per step.
// Sequences nobody needs (flags == 0) get zeros in out / alpha from the prologue and are left out of the walk.
// =========================================================================================
struct PoolFusedArgs {
  const bf16_t* x; int ldx;          // [n * L, ldx]
  const bf16_t* w1; int ldw1;        // [q, ldw1] packed (zero beyond N up to a multiple of 32)
  const float* b1; const float* w2; const float* b2;
  bf16_t* e; int lde;                // [n * L, lde] out
  float* alpha;                      // [n * L] out
  float* out; int ld_out;            // [n, ld_out] out
  const int32_t* needed;             // [n] or null
  int n, L, N, q;
  int ablate;                        // NR_OPT_POOL_ABLATE (measurement only)
};

template <int KS>
struct PoolFusedCfg {
  static constexpr int NW = 8, TPW = 2, RT = 2, R = 32;
  static constexpr int APIECES = RT * KS, PW = (APIECES + NW - 1) / NW, STAGE = APIECES * 1024;
  static constexpr int NS = 4;
  static constexpr int S = RT;                                         // e stores per wave and step (two tiles paired per store)
  static constexpr int NCOL = KS * 32;                                 // columns of x the ring holds (>= N)
  static constexpr int DUMP = NS * STAGE, BIAS = DUMP + NW * 1024;     // per-wave dump piece | bias [256] fp32
  static constexpr int PART = BIAS + 1024;                             // sPart [2][NW][32] fp32
  static constexpr int RED = PART + 2 * NW * 32 * 4;                   // sRed [2][NW][NCOL] fp32
  static constexpr int LIST = RED + 2 * NW * NCOL * 4;                 // the walk: sequence numbers
  static constexpr int STEADY = PW + 2 * (S + 2);                      // ops younger than stage k's DMAs at the top of step k
  static_assert(STEADY <= 62, "vmcnt is a 6-bit counter");
};

template <int KS>
__global__ __launch_bounds__(512) void pool_fused_fwd_kernel(PoolFusedArgs a, int max_steps) {
  using Cfg = PoolFusedCfg<KS>;
  constexpr int NW = Cfg::NW, TPW = Cfg::TPW, RT = Cfg::RT, R = Cfg::R, NS = Cfg::NS, PW = Cfg::PW, STAGE = Cfg::STAGE, NCOL = Cfg::NCOL;
  constexpr int APIECES = Cfg::APIECES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int prow = lane >> 2, c8 = ((lane & 3) ^ swzP(prow)) * 8;
  const int L = a.L, N = a.N, q = a.q, M = a.n * a.L;
  float* sBias = reinterpret_cast<float*>(smem + Cfg::BIAS);
  float* sPart = reinterpret_cast<float*>(smem + Cfg::PART);
  float* sRed = reinterpret_cast<float*>(smem + Cfg::RED);
  int* sList = reinterpret_cast<int*>(smem + Cfg::LIST);

  // ---- the walk: this workgroup's needed sequences, in order; the others get their zeros here
  const int G = gridDim.x, b0 = blockIdx.x;
  int nsteps = b0 < a.n ? (a.n - b0 + G - 1) / G : 0;
  if (a.needed != nullptr) {
    __shared__ int sCnt[NW + 1];
    const int cand = nsteps;
    int run = 0;
    for (int base = 0; base < cand; base += 512) {
      const int c = base + tid, sq = b0 + c * G;
      const bool live = c < cand && a.needed[sq] != 0;
      const uint64_t bal = __ballot(live);
      if (lane == 0) sCnt[wid] = __popcll(bal);
      __syncthreads();
      int before = 0, total = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const int cw = sCnt[w];
        before += w < wid ? cw : 0;
        total += cw;
      }
      if (live) sList[run + before + __popcll(bal & ((1ull << lane) - 1ull))] = sq;
      run += total;
      __syncthreads();
    }
    // zeros for the sequences left out: a wave per sequence
    for (int c = wid; c < cand; c += NW) {
      const int sq = b0 + c * G;
      if (a.needed[sq] != 0) continue;                   // wave-uniform
      if (lane < L) a.alpha[(size_t)sq * L + lane] = 0.f;
      for (int col = lane; col < N; col += 64) a.out[(size_t)sq * a.ld_out + col] = 0.f;
    }
    nsteps = run;
  } else {
    for (int c = tid; c < nsteps; c += 512) sList[c] = b0 + c * G;
  }
  if (nsteps == 0) return;

  // ---- W1 slice of this wave (2 column tiles x K), w2 of the lane's 8 columns, bias of all columns
  const int wcol0 = wid * TPW * 16;
  bf16x8 bfr[TPW][KS];
  float w2r[TPW][4];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int nrow = min(wcol0 + t * 16 + fr, q - 1);
    const bf16_t* brow = a.w1 + (size_t)nrow * a.ldw1 + 8 * g;
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) bfr[t][s2] = *reinterpret_cast<const bf16x8*>(brow + 32 * s2);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = wcol0 + t * 16 + 4 * g + r;
      w2r[t][r] = col < q ? a.w2[col] : 0.f;             // columns beyond q add nothing to the logit
    }
  }
  for (int c = tid; c < NW * TPW * 16; c += 512) sBias[c] = c < q ? a.b1[c] : 0.f;
  const float b2 = a.b2[0];
  __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): nothing of the prologue is carried into the loop
  __syncthreads();                                       // list + bias complete

  const uint32_t dump_lds = lds0 + Cfg::DUMP + wid * 1024;
  // stage issue: every wave exactly PW DMAs (surplus ones and the steps past the end into its dump piece)
  auto issue_stage = [&](int kk) {
    const uint32_t slot = lds0 + (kk % NS) * STAGE;
    const bool real = kk < nsteps;
    const int row0 = real ? sList[kk] * L : 0;
    const bf16_t* rowp[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) rowp[i] = a.x + (size_t)min(row0 + 16 * i + prow, M - 1) * a.ldx + c8;
#pragma unroll
    for (int u = 0; u < PW; ++u) {
      const int p = wid + NW * u;                        // wave-uniform
      if (real && p < APIECES) {
        const int i = p / KS, s2 = p - i * KS;
        const bf16_t* rp = i == 0 ? rowp[0] : rowp[1];
        dma16(rp + min(32 * s2, N - 8 - c8), slot + p * 1024);   // (K tail: re-read a valid chunk; W1 is zero there)
      } else {
        dma16(a.x + c8, dump_lds);
      }
    }
  };
  uint4* dump_g = g_nt_dump + wid * 64 + lane;
  const int offA = fr * 64 + ((g ^ swzP(fr)) << 4);
  // weighted-sum geometry (S2): lane = 16-byte column chunk (0 .. N/8), wave = row group (rows wid, wid + 8, ...)
  const int cc = lane, ccp = cc >> 2, ccs = cc & 3;
  const bool cc_ok = cc * 8 < N;

#pragma unroll 1
  for (int kk = 0; kk < NS - 2; ++kk) issue_stage(kk);

  // 32-lane all-reduce of a value held identically by lanes l and l + 32: four DPP row rotations + one cross-row exchange
  auto sum32 = [&](float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
    return v + __shfl_xor(v, 16, 64);
  };
  // S1: sequence k - 2 -> out
  auto phase_out = [&](int k) {
    const int kf = k - 2;
    float* dst = reinterpret_cast<float*>(dump_g);
    if (kf >= 0 && kf < nsteps && tid < N && !(a.ablate & 4)) {
      const float* red = sRed + (kf & 1) * NW * NCOL + tid;
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < NW; ++r) v += red[r * NCOL];
      *(a.out + (size_t)sList[kf] * a.ld_out + tid) = v;
    } else {
      *dst = 0.f;                                                              // same store count in every wave
    }
  };
  // S2: sequence k - 1 -> alpha, this wave's rows of the weighted sum
  auto phase_alpha = [&](int k) {
    const int kp = k - 1;
    const bool real = kp >= 0 && kp < nsteps;
    const int sq = real ? sList[kp] : 0;
    const float* part = sPart + (kp & 1) * NW * 32 + (lane & 31);
    float al = 1.f;
    if (!(a.ablate & 2)) {
      // a_l = exp(s_l) / (sum + 1e-8), exactly as src/model/model_utils.py:23-30 writes it (no maximum is subtracted there
      // either; |s| <= |w2|_1 + |b2| because |e| <= 1)
      float sl = b2;
#pragma unroll
      for (int w = 0; w < NW; ++w) sl += part[w * 32];
      const float ex = (lane & 31) < L ? __expf(sl) : 0.f;
      al = ex / (sum32(ex) + 1e-8f);
    }
    float* adst = (real && wid == 0 && lane < L) ? a.alpha + (size_t)sq * L + lane : reinterpret_cast<float*>(dump_g);
    *adst = al;
    if (real && !(a.ablate & 1)) {
      const char* st = smem + (kp % NS) * STAGE;
      // lane = 16-byte column chunk (clamped: the surplus lanes read chunk 0 and drop it), wave = rows wid, wid + 8, ...
      const int ccr = cc_ok ? cc : 0;
      bf16x8 xv[R / NW];
#pragma unroll
      for (int j = 0; j < R / NW; ++j) {
        const int m = min(wid + NW * j, L - 1);            // rows L .. 31 belong to the NEXT sequence (maybe unwritten memory): never read
        xv[j] = *reinterpret_cast<const bf16x8*>(st + ((m >> 4) * KS + (ccr >> 2)) * 1024 + (m & 15) * 64 + (((ccr & 3) ^ swzP(m & 15)) << 4));
      }
      float acc[8];
#pragma unroll
      for (int e2 = 0; e2 < 8; ++e2) acc[e2] = 0.f;
#pragma unroll
      for (int j = 0; j < R / NW; ++j) {
        const int m = wid + NW * j;                          // wave-uniform
        const float am = m < L ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, al), m < L ? m : 0)) : 0.f;
#pragma unroll
        for (int e2 = 0; e2 < 8; ++e2) acc[e2] = fmaf(am, (float)xv[j][e2], acc[e2]);
      }
      if (cc_ok) {
        float* rd = sRed + ((kp & 1) * NW + wid) * NCOL + cc * 8;
        *reinterpret_cast<f32x4*>(rd) = (f32x4){acc[0], acc[1], acc[2], acc[3]};
        *reinterpret_cast<f32x4*>(rd + 4) = (f32x4){acc[4], acc[5], acc[6], acc[7]};
      }
    }
  };
  // S3: sequence k -> e, partial logits
  auto phase_gemm = [&](int k) {
    const bool real = k < nsteps;
    const int row0 = real ? sList[k] * L : 0;
    const char* st = smem + (k % NS) * STAGE;
    f32x4 acc[RT][TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(sBias + (wid * TPW + t) * 16 + 4 * g);
#pragma unroll
      for (int i = 0; i < RT; ++i) acc[i][t] = bv;
    }
    if (real && wcol0 < q && !(a.ablate & 8)) {                              // wave-uniform
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(st + (i * KS + s2) * 1024 + offA);
#pragma unroll
          for (int t = 0; t < TPW; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[t][s2], af, acc[i][t], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int ml = 16 * i + fr;                                            // row inside the sequence
      const bool rok = real && ml < L;
      bf16_t* crow = a.e + (size_t)(row0 + ml) * a.lde;
      float pd = 0.f;
      uint2 pk[TPW];
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float th = (a.ablate & 16) ? acc[i][t][r] : 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * acc[i][t][r]) + 1.f);
          o[r] = (bf16_t)th;
          pd = fmaf((float)o[r], w2r[t][r], pd);                              // the logit sees e as the backward will (bf16)
        }
        pk[t] = __builtin_bit_cast(uint2, o);
      }
      const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
      const int col = wcol0 + (g & 1) * 16 + (g >> 1) * 8;
      uint4* dst = (rok && col + 8 <= q) ? reinterpret_cast<uint4*>(crow + col) : dump_g;
      *dst = make_uint4(s0[0], s1[0], s0[1], s1[1]);
      pd += __shfl_xor(pd, 16, 64);
      pd += __shfl_xor(pd, 32, 64);
      if (g == 0) sPart[((k & 1) * NW + wid) * 32 + ml] = pd;
    }
  };

#pragma unroll 1
  for (int k = 0; k < nsteps + 2; ++k) {
    // stage k landed: younger than its DMAs are the stores of steps k-2, k-1 and the DMAs of step k-1
    if (k >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cfg::STEADY) : "memory");
    else wait_vmcnt_le(k == 0 ? (NS - 3) * PW : PW + (Cfg::S + 2));
    __builtin_amdgcn_s_barrier();
    issue_stage(k + NS - 2);                                                  // S0
    // The three phases of a step touch three different sequences and disjoint LDS buffers: any order is legal.  The two waves
    // of a SIMD (w and w + 4) take opposite orders, so one runs its MFMAs while the other is in its VALU / LDS phases
    // (MI355X_MICROARCH.md, "Two waves that run the SAME program with one barrier per block: try a stagger")
    if (wid < NW / 2) {
      phase_gemm(k);
      phase_out(k);
      phase_alpha(k);
    } else {
      phase_out(k);
      phase_alpha(k);
      phase_gemm(k);
    }
  }
}

// =========================================================================================
// Fused additive-attention pooling BACKWARD core for sequences of L <= 32 tokens (src/model/model_utils.py:21-30 differentiated):
//   dA_l = <g, x_l> ; ds_l = a_l (dA_l - sum_u a_u dA_u) ; dpre_l = ds_l w2 (1 - e_l^2) ; dX_l = dpre_l W1 + a_l g
//   dw2 += sum_l ds_l e_l ; db2 += sum_l ds_l                       (dW1 = dpre^T x stays with the weight-gradient GEMM)
// The unfused path ran pool_core_bwd (x, e in; dpre out) and the weights-in-registers dX GEMM (dpre in again; dX out): 0.25 +
// 0.19 ms at the news level.  Here a stage = ONE SEQUENCE, three sequences are in flight, one barrier per step k:
//   S0  DMA e | g of sequence k + 2 into the ring (the e pieces in the MFMA fragment layout)
//   A   sequence k    : dA partials -- this wave's k-steps of x . g on the matrix cores (g as bf16 hi + lo; the x fragments
//       come straight from global memory into registers, loaded one step ahead: x never goes through LDS)      -> sPartA
//   B   sequence k - 1: dA, ds (every wave, 32 lanes); the workgroup's 8-column chunks of dpre = ds w2 (1 - e^2) written
//       IN PLACE over e in the ring and to global memory; dw2 / db2 partial sums in registers
//   C   sequence k - 2: dX = dpre . W1 (W1^T resident in registers, A fragments from the ring) + alpha g    -> global (bf16)
// Sequences with a zero pooled gradient (flags) are left out of the walk: their dX rows are zero-filled in the prologue, their
// dpre rows only where a 32-row slab of the weight-gradient GEMM can reach them (within `reach` sequences of a live one).
// Vector-memory bookkeeping is static: every wave issues PW DMAs, 5 loads and 6 stores per step, in that order.
// =========================================================================================
struct PoolFusedBwdArgs {
  const bf16_t* x; int ldx;          // [n * L, ldx]
  const bf16_t* e; int lde;          // [n * L, lde]
  const float* alpha;                // [n * L]
  const float* g; int ldg;           // [n, ldg] pooled gradient
  const float* w2;                   // [q]
  const bf16_t* w1t; int ldw1t;      // [N, ldw1t] = W1^T packed (zero beyond q up to a multiple of 32)
  bf16_t* dpre; int ldp;             // [n * L, ldp] out
  bf16_t* dx; int lddx;              // [n * L, lddx] out
  float* partial;                    // [gridDim.x, q + 1] out: this workgroup's dw2 | db2
  const int32_t* nz;                 // [n] or null: 0 = zero pooled gradient
  int n, L, N, q;
  int dx_far_unwritten;              // dX rows of zero-gradient sequences far from every live one stay unwritten (nr_pool_desc)
};

template <int KS, int KS2>
struct PoolFusedBwdCfg {
  static constexpr int NW = 8, TPW = 4, RT = 2, R = 32;
  static constexpr int EPIECES = RT * KS2, GPIECES = 2, PIECES = EPIECES + GPIECES;
  static constexpr int PW = (PIECES + NW - 1) / NW, STAGE = PIECES * 1024, NS = 5;
  static constexpr int LOADS = 5;                                       // 2 row tiles x 2 k-steps of x + alpha
  static constexpr int OPS = PW + LOADS + 2 + RT * 2;                   // + dpre stores + dX stores
  static constexpr int STEADY = 2 * OPS - PW;                           // ops younger than stage k's DMAs at the top of step k
  static constexpr int DUMP = NS * STAGE;                               // per-wave dump piece
  static constexpr int PARTA = DUMP + NW * 1024;                        // sPartA [2][NW][32] fp32
  static constexpr int ALPHA = PARTA + 2 * NW * 32 * 4;                 // sAlpha [4][32] fp32
  static constexpr int W2 = ALPHA + 4 * 32 * 4;                         // sW2 [256] fp32 | sRedW [256] fp32
  static constexpr int ACCW = W2 + 2 * 256 * 4;                         // per-thread dw2 partial sums [512][16] fp32 (registers are full)
  static constexpr int LIST = ACCW + 512 * 16 * 4;
  static_assert(STEADY <= 62, "vmcnt is a 6-bit counter");
};

template <int KS, int KS2>
__global__ __launch_bounds__(512) void pool_fused_bwd_kernel(PoolFusedBwdArgs a, int reach) {
  using Cfg = PoolFusedBwdCfg<KS, KS2>;
  constexpr int NW = Cfg::NW, TPW = Cfg::TPW, RT = Cfg::RT, NS = Cfg::NS, PW = Cfg::PW, STAGE = Cfg::STAGE, EPIECES = Cfg::EPIECES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int prow = lane >> 2, c8 = ((lane & 3) ^ swzP(prow)) * 8;
  const int L = a.L, N = a.N, q = a.q, M = a.n * a.L, qc = q >> 3;
  float* sPartA = reinterpret_cast<float*>(smem + Cfg::PARTA);
  float* sAlpha = reinterpret_cast<float*>(smem + Cfg::ALPHA);
  float* sW2 = reinterpret_cast<float*>(smem + Cfg::W2);
  float* sRedW = sW2 + 256;
  int* sList = reinterpret_cast<int*>(smem + Cfg::LIST);

  // ---- the walk: this workgroup's sequences with a non-zero pooled gradient; zeros for the others
  const int G = gridDim.x, b0 = blockIdx.x;
  int nsteps = b0 < a.n ? (a.n - b0 + G - 1) / G : 0;
  if (a.nz != nullptr) {
    __shared__ int sCnt[NW + 1];
    const int cand = nsteps;
    int run = 0;
    for (int base = 0; base < cand; base += 512) {
      const int c = base + tid, sq = b0 + c * G;
      const bool live = c < cand && a.nz[sq] != 0;
      const uint64_t bal = __ballot(live);
      if (lane == 0) sCnt[wid] = __popcll(bal);
      __syncthreads();
      int before = 0, total = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const int cw = sCnt[w];
        before += w < wid ? cw : 0;
        total += cw;
      }
      if (live) sList[run + before + __popcll(bal & ((1ull << lane) - 1ull))] = sq;
      run += total;
      __syncthreads();
    }
    for (int c = wid; c < cand; c += NW) {                 // a wave per zero-gradient sequence
      const int sq = b0 + c * G;
      if (a.nz[sq] != 0) continue;                         // wave-uniform
      bool near = false;
      for (int t = max(0, sq - reach) + lane; t <= min(a.n - 1, sq + reach); t += 64) near |= a.nz[t] != 0;
      near = __ballot(near) != 0ull;
      const size_t r0 = (size_t)sq * L;
      const int xc = N >> 3;
      if (near || !a.dx_far_unwritten)
        for (int u = lane; u < L * xc; u += 64) {
          const int row = u / xc, ch = u - row * xc;
          *reinterpret_cast<uint4*>(a.dx + (r0 + row) * a.lddx + ch * 8) = make_uint4(0, 0, 0, 0);
        }
      if (near)
        for (int u = lane; u < L * qc; u += 64) {
          const int row = u / qc, ch = u - row * qc;
          *reinterpret_cast<uint4*>(a.dpre + (r0 + row) * a.ldp + ch * 8) = make_uint4(0, 0, 0, 0);
        }
    }
    nsteps = run;
  } else {
    for (int c = tid; c < nsteps; c += 512) sList[c] = b0 + c * G;
  }
  for (int c = tid; c < 256; c += 512) {
    sW2[c] = c < q ? a.w2[c] : 0.f;
    sRedW[c] = 0.f;
  }
  // ---- W1^T slice of this wave: TPW column tiles x KS2 k-steps
  const int wcol0 = wid * TPW * 16;
  bf16x8 bfr[TPW][KS2];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const bf16_t* brow = a.w1t + (size_t)min(wcol0 + t * 16 + fr, N - 1) * a.ldw1t + 8 * g;
#pragma unroll
    for (int s2 = 0; s2 < KS2; ++s2) bfr[t][s2] = *reinterpret_cast<const bf16x8*>(brow + 32 * s2);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): nothing of the prologue is carried into the loop
  __syncthreads();
  float* prow_out = a.partial + (size_t)blockIdx.x * (q + 1);
  if (nsteps == 0) {                                       // (the caller sums every workgroup's partial row)
    for (int c = tid; c <= q; c += 512) prow_out[c] = 0.f;
    return;
  }

  const uint32_t dump_lds = lds0 + Cfg::DUMP + wid * 1024;
  uint4* dump_g = g_nt_dump + wid * 64 + lane;
  auto issue_stage = [&](int kk) {
    const uint32_t slot = lds0 + (kk % NS) * STAGE;
    const bool real = kk < nsteps;
    const int sq = real ? sList[kk] : 0;
    const int row0 = sq * L;
#pragma unroll
    for (int u = 0; u < PW; ++u) {
      const int p = wid + NW * u;                          // wave-uniform
      if (real && p < EPIECES) {
        const int i = p / KS2, s2 = p - i * KS2;
        const bf16_t* rp = a.e + (size_t)min(row0 + 16 * i + prow, M - 1) * a.lde + c8;
        dma16(rp + min(32 * s2, q - 8 - c8), slot + p * 1024);       // (K tail: a valid chunk again; W1^T is zero there)
      } else if (real && p < EPIECES + Cfg::GPIECES) {
        const int j = p - EPIECES;
        dma16(a.g + (size_t)sq * a.ldg + min(256 * j + 4 * lane, N - 4), slot + p * 1024);
      } else {
        dma16(a.e + c8, dump_lds);
      }
    }
  };
  // x fragments of this wave for phase A: k-steps wid and wid + 8 (the second one empty for waves 5..7), both row tiles;
  // alpha of the same sequence (lane = row), for its phase B one step later
  const int ks0 = wid, ks1 = wid + NW;
  // (step k loads the x fragments of sequence k + 1 -- its phase A runs at step k + 1 -- and the alpha of sequence k, whose
  //  phase B runs at step k + 1 too)
  auto load_seq = [&](int kk, bf16x8 (&xf)[RT][2], float& al) {
    const int row0 = sList[kk < nsteps ? kk : 0] * L;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const bf16_t* rp = a.x + (size_t)min(row0 + 16 * i + fr, M - 1) * a.ldx;
      xf[i][0] = *reinterpret_cast<const bf16x8*>(rp + min(32 * ks0 + 8 * g, N - 8));
      xf[i][1] = *reinterpret_cast<const bf16x8*>(rp + min(32 * ks1 + 8 * g, N - 8));
    }
    const int rowa = sList[(kk >= 1 && kk - 1 < nsteps) ? kk - 1 : 0] * L;
    al = a.alpha[min(rowa + (lane & 31), M - 1)];
  };
  // g as the MFMA A operand of k-step ks: elements 32 ks + 8 g .. + 7 of the pooled-gradient row, split bf16 hi + lo
  auto g_frag = [&](const float* grow, int ks, bf16x8& hi, bf16x8& lo) {
    const int k0 = 32 * ks + 8 * g;
    const bool ok = ks < KS && k0 < N;                      // (the clamped x chunks beyond N meet zeros here)
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(grow + min(k0, N - 8));
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(grow + min(k0, N - 8) + 4);
#pragma unroll
    for (int e2 = 0; e2 < 8; ++e2) {
      const float v = ok ? (e2 < 4 ? v0[e2] : v1[e2 - 4]) : 0.f;
      const bf16_t h = (bf16_t)v;
      hi[e2] = h;
      lo[e2] = (bf16_t)(v - (float)h);
    }
  };
  auto sum32 = [&](float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v + __shfl_xor(v, 16, 64);
  };
  const int offA = fr * 64 + ((g ^ swzP(fr)) << 4);
  const uint32_t inv_qc = (uint32_t)((0x100000000ull + (uint32_t)qc - 1) / (uint32_t)qc);
  // phase B geometry: this thread's two 8-column chunks (the same columns every step)
  int brow[2], bcc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const uint32_t u = (uint32_t)(tid + 512 * j);
    uint32_t r = __umulhi(u, inv_qc);
    if (r * (uint32_t)qc > u) --r;
    brow[j] = (int)r;
    bcc[j] = (int)(u - r * (uint32_t)qc);
  }

  bf16x8 xf[RT][2];
  float al_nb, al_cur = 0.f;
  // dw2 partial sums of this thread's two chunks: a private strip of LDS, [j][thread] x 8 floats (conflict-free b128 accesses)
  float* myacc = reinterpret_cast<float*>(smem + Cfg::ACCW) + tid * 8;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    *reinterpret_cast<f32x4*>(myacc + j * 4096) = (f32x4){0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(myacc + j * 4096 + 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float accb = 0.f;

#pragma unroll 1
  for (int kk = 0; kk < 2; ++kk) issue_stage(kk);
  load_seq(0, xf, al_nb);

#pragma unroll 1
  for (int k = 0; k < nsteps + 2; ++k) {
    // stage k landed: younger than its DMAs are all ops of step k-1 and those of step k-2 behind its DMA issue
    if (k >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cfg::STEADY) : "memory");
    else wait_vmcnt_le(k == 0 ? PW + Cfg::LOADS : Cfg::OPS + Cfg::LOADS);
    __builtin_amdgcn_s_barrier();
    issue_stage(k + 2);                                                       // S0
    // ---- A: sequence k -> partial dA from the x fragments loaded one step ago; then fetch those of sequence k + 1
    auto phase_a = [&]()     {
      const bool real = k < nsteps;
      const float* grow = reinterpret_cast<const float*>(smem + (k % NS) * STAGE + EPIECES * 1024);
      f32x4 da[RT];
#pragma unroll
      for (int i = 0; i < RT; ++i) da[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (real) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bf16x8 hi, lo;
          g_frag(grow, j == 0 ? ks0 : ks1, hi, lo);
#pragma unroll
          for (int i = 0; i < RT; ++i) {
            da[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, xf[i][j], da[i], 0, 0, 0);
            da[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo, xf[i][j], da[i], 0, 0, 0);
          }
        }
      }
      if (g == 0) {
#pragma unroll
        for (int i = 0; i < RT; ++i) sPartA[((k & 1) * NW + wid) * 32 + 16 * i + fr] = da[i][0];
      }
      al_cur = al_nb;                                      // alpha of sequence k - 1 (loaded one step ago) for phase B below
      load_seq(k + 1, xf, al_nb);
    };
    // ---- B: sequence k - 1 -> ds, dpre (in place over e, and to global), dw2 / db2 partial sums
    auto phase_b = [&]()     {
      const int kp = k - 1;
      const bool real = kp >= 0 && kp < nsteps;
      const int row0 = real ? sList[kp] * L : 0;
      const bool in = (lane & 31) < L;
      const float alv = in ? al_cur : 0.f;
      const float* part = sPartA + ((kp & 1) * NW) * 32 + (lane & 31);
      float dA = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) dA += part[w * 32];
      dA = in ? dA : 0.f;                                  // (rows L .. 31 belong to the next sequence: whatever came out of them)
      const float tsum = sum32(alv * dA);
      const float ds = alv * (dA - tsum);
      if (wid == 0 && lane < 32) sAlpha[(kp & 3) * 32 + lane] = alv;
      if (real) accb += sum32(ds);
      char* st = smem + ((kp + NS) % NS) * STAGE;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = brow[j], cc = bcc[j];
        const bool valid = real && row < L;
        const int rc = min(row, L - 1);
        const float dsr = __shfl(ds, rc, 64);
        const int eoff = ((rc >> 4) * KS2 + (cc >> 2)) * 1024 + (rc & 15) * 64 + (((cc & 3) ^ swzP(rc & 15)) << 4);
        const bf16x8 ev = *reinterpret_cast<const bf16x8*>(st + eoff);
        const f32x4 wa = *reinterpret_cast<const f32x4*>(sW2 + cc * 8), wb = *reinterpret_cast<const f32x4*>(sW2 + cc * 8 + 4);
        bf16x8 o;
        f32x4 a0 = *reinterpret_cast<const f32x4*>(myacc + j * 4096), a1 = *reinterpret_cast<const f32x4*>(myacc + j * 4096 + 4);
#pragma unroll
        for (int e2 = 0; e2 < 8; ++e2) {
          const float f = (float)ev[e2], wv = e2 < 4 ? wa[e2] : wb[e2 - 4];
          o[e2] = (bf16_t)(dsr * wv * (1.f - f * f));
          const float p = valid ? dsr * f : 0.f;             // (a step without a sequence reads whatever the ring slot holds)
          if (e2 < 4) a0[e2] += p;
          else a1[e2 - 4] += p;
        }
        *reinterpret_cast<f32x4*>(myacc + j * 4096) = a0;
        *reinterpret_cast<f32x4*>(myacc + j * 4096 + 4) = a1;
        if (valid) *reinterpret_cast<bf16x8*>(st + eoff) = o;
        uint4* dst = valid ? reinterpret_cast<uint4*>(a.dpre + (size_t)(row0 + row) * a.ldp + cc * 8) : dump_g;
        *dst = __builtin_bit_cast(uint4, o);
      }
    };
    // ---- C: sequence k - 2 -> dX
    auto phase_c = [&]()     {
      const int kc = k - 2;
      const bool real = kc >= 0 && kc < nsteps;
      const int row0 = real ? sList[kc] * L : 0;
      const char* st = smem + ((kc + NS) % NS) * STAGE;
      const float* grow = reinterpret_cast<const float*>(st + EPIECES * 1024);
      // the two row tiles one after the other through ONE set of accumulators (16 registers less: the kernel sits at 256)
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        f32x4 acc[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (real && wcol0 < N) {                           // wave-uniform
#pragma unroll
          for (int s2 = 0; s2 < KS2; ++s2) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(st + (i * KS2 + s2) * 1024 + offA);
#pragma unroll
            for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[t][s2], af, acc[t], 0, 0, 0);
          }
        }
        const int ml = 16 * i + fr;
        const bool rok = real && ml < L;
        const float al = sAlpha[((kc + 4) & 3) * 32 + ml];
        bf16_t* crow = a.dx + (size_t)(row0 + ml) * a.lddx;
#pragma unroll
        for (int t = 0; t < TPW; t += 2) {
          uint2 pk[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int lc = wcol0 + (t + h) * 16 + 4 * g;     // this lane's first column of tile t + h
            const f32x4 gq = *reinterpret_cast<const f32x4*>(grow + lc);
            const f32x4 v = acc[t + h];
            const bf16x4 o = {(bf16_t)(v[0] + al * gq[0]), (bf16_t)(v[1] + al * gq[1]), (bf16_t)(v[2] + al * gq[2]), (bf16_t)(v[3] + al * gq[3])};
            pk[h] = __builtin_bit_cast(uint2, o);
          }
          const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
          const int col = wcol0 + (t + (g & 1)) * 16 + (g >> 1) * 8;
          uint4* dst = (rok && col + 8 <= N) ? reinterpret_cast<uint4*>(crow + col) : dump_g;
          *dst = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
      }
    };
    // The three phases touch three different sequences and disjoint buffers: any order is legal, and the static vmcnt
    // bookkeeping only counts operations per step.  The two waves of a SIMD (w and w + 4) take opposite orders, so one runs the
    // dX MFMAs while the other is in the VALU / LDS phases.
    if (wid < NW / 2) {
      phase_a();
      phase_b();
      phase_c();
    } else {
      phase_c();
      phase_a();
      phase_b();
    }
  }
  // ---- this workgroup's dw2 | db2: threads that own the same column chunk meet in LDS
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e2 = 0; e2 < 8; ++e2) atomicAdd(sRedW + bcc[j] * 8 + e2, myacc[j * 4096 + e2]);
  __syncthreads();
  for (int c = tid; c < q; c += 512) prow_out[c] = sRedW[c];
  if (tid == 0) prow_out[q] = accb;
}

template <int EPI, int NT16, bool PK, int WM>
int launch_nt_dma_w(const RowSrc& A, const void* B, int ldb, int M, int N, int K, const EpiArgs& ep, hipStream_t stream) {
  constexpr int DBM = 64 * WM, NP = DBM / 16 + NT16, STAGE = NP * 1024, NS = dma_ring_stages(STAGE, WM);
  constexpr size_t ring = (size_t)NS * STAGE, epi = (size_t)32 * (NT16 * 16 + 4) * sizeof(float);
  constexpr size_t epk = PK ? (size_t)128 * (NT16 * 16 + 8) * sizeof(bf16_t) : 0;
  constexpr size_t gtile = EPI == EPI_POOLBWD && PK ? (size_t)((12 * NT16 * 16 * 4 + 1023) / 1024) * 1024 + 1024
                                                    : ((EPI == EPI_SCATTER || EPI == EPI_STORE) ? 2048 : 1024);
  constexpr size_t smem0 = ring > epi ? ring : epi, smem1 = smem0 > epk ? smem0 : epk, smem = smem1 + gtile;
  auto kern = gemm_nt_dma_kernel<EPI, NT16, PK, WM>;
  NR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const int tilesM = (M + DBM - 1) / DBM, nchunks = (N + NT16 * 16 - 1) / (NT16 * 16);
  const int grid = ((tilesM + 7) / 8) * 8 * nchunks;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * WM), smem, stream, (const bf16_t*)A.base, A.ld, (const bf16_t*)B, ldb, M, N, K, ep,
                     nchunks | (nr_opt(NR_OPT_NT_ABLATE) << 16));
  NR_CHECK_LAUNCH();
  return NR_OK;
}

template <int EPI, int NT16, bool PK>
int launch_nt_dma_p(const RowSrc& A, const void* B, int ldb, int M, int N, int K, const EpiArgs& ep, hipStream_t stream) {
  // two 128-row workgroups per CU when three ring stages of the small tile fit twice into LDS and the whole N is one
  // column chunk (att_fc1: 0.47 -> 0.41 ms); with several chunks the smaller tile re-reads A more often and loses
  // (QKV projection: 1.45 -> 1.57 ms)
  constexpr bool fits2 = 3 * (8 + NT16) * 1024 <= 78 * 1024;
  const bool wm2_all = nr_opt(NR_OPT_DMA_WM2_ALL) != 0;
  // short contractions (K <= 224: at most 7 k-steps) are all pipeline fill and epilogue: two workgroups per CU overlap
  // them even with two column chunks (pooling dX, N = 400, K = 200: 0.445 -> 0.372 ms)
  // ... and up to two column chunks the second read of A is cheaper than the idle fill / epilogue phases of a lone
  // workgroup (NAML conv, N = 400, K = 960: 0.738 -> 0.694 ms); six chunks (QKV projection) still lose (0.51 -> 0.57)
  if (fits2 && (N <= 2 * NT16 * 16 || K <= 224 || wm2_all)) return launch_nt_dma_w<EPI, NT16, PK, 2>(A, B, ldb, M, N, K, ep, stream);
  return launch_nt_dma_w<EPI, NT16, PK, 4>(A, B, ldb, M, N, K, ep, stream);
}

template <int EPI, int NT16>
int launch_nt_dma_t(const RowSrc& A, const void* B, int ldb, int M, int N, int K, const EpiArgs& ep, hipStream_t stream) {
  if (EPI != EPI_SCATTER && ep.out_dtype == NR_BF16) return launch_nt_dma_p<EPI, NT16, true>(A, B, ldb, M, N, K, ep, stream);
  return launch_nt_dma_p<EPI, NT16, false>(A, B, ldb, M, N, K, ep, stream);
}

template <int NT16>
int launch_nt_dma_e(const RowSrc& A, const void* B, int ldb, int M, int N, int K, int epi, const EpiArgs& ep, hipStream_t s) {
  switch (epi) {
    case EPI_STORE: return launch_nt_dma_t<EPI_STORE, NT16>(A, B, ldb, M, N, K, ep, s);
    case EPI_STORE_TANH: return launch_nt_dma_t<EPI_STORE_TANH, NT16>(A, B, ldb, M, N, K, ep, s);
    case EPI_POOLBWD: return launch_nt_dma_t<EPI_POOLBWD, NT16>(A, B, ldb, M, N, K, ep, s);
    case EPI_SCATTER: return launch_nt_dma_t<EPI_SCATTER, NT16>(A, B, ldb, M, N, K, ep, s);
  }
  nr_set_error("gemm_nt_dma: bad epilogue %d", epi);
  return NR_ERR_ARG;
}

// =========================================================================================
// TN on the LDS-DMA ring (bf16, dense operands, rows a multiple of 32): dW[n][k] += sum_m dC[m][n] X[m][k].
// Same math and LDS images as tn2 (operands stay row-major as they lie in memory, fragments come out of
// ds_read_b64_tr_b16, swizzles on the 16-byte chunk index), but the tiles travel global -> LDS with
// global_load_lds_dwordx4 into a 4-stage ring of 32-row slabs, 3 stages in flight, no staging registers: the
// contraction over M = 844 800 rows is all main loop, and the register-staged version kept one slab in flight.
// Block tile 128 (n) x 80*WK (k): WK = 4 covers K <= 320 in one tile (dC, the 2 GB operand of the QKV weight
// gradient, is read once), 8 waves, one workgroup per CU; WK = 2: 4 waves, two workgroups per CU.
// Column tails are clamped to valid chunks (they only feed outputs that are never stored).
// db: two waves (on different SIMDs) run one extra MFMA per A fragment against a vector of ones.
// =========================================================================================
namespace tn3 {
using tn2::swz_a;
using tn2::tr_frag;
constexpr int TBM = 32, NS = 4;
// NI = 16-row A fragments per wave (2 n-waves): NI = 4 -> 128-row n-tile, NI = 8 -> 256-row n-tile (X is re-read
// half as often and a wave runs 40 MFMAs per 13 fragment loads instead of 20 per 9: LDS traffic stops binding).
template <int WK, int NI> struct Geo {
  static constexpr int TBN = 32 * NI, TBK = 80 * WK, NT = 128 * WK, NW = 2 * WK, CHA = TBN / 8, CH = TBK / 8, SCW = TBK + 4;
  static constexpr int PA = TBM * CHA / 64, PBP = TBM * CH / 64, NP = PA + PBP, STAGE = NP * 1024;
  static constexpr size_t RING = (size_t)NS * STAGE, EPI = (size_t)64 * SCW * sizeof(float);
  static constexpr size_t SMEM = RING > EPI ? RING : EPI;
};
// X tile rows are 320 B (WK=2) or 640 B (WK=4) apart: the 4 rows of a 16-lane group and the odd/even 8-row groups
// must land on different 32-byte segments of a 256-byte bank window
template <int WK> __device__ __forceinline__ int swz_b(int row, int chunk) {
  return WK == 2 ? chunk ^ (2 * ((row >> 3) & 1)) : chunk ^ (4 * ((row >> 1) & 1) + 2 * ((row >> 3) & 1));
}

template <int WK, int NI>
__global__ __launch_bounds__(128 * WK) void gemm_tn3_kernel(const bf16_t* __restrict__ dC, int ldc, const bf16_t* __restrict__ X,
                                                            int ldx, float* __restrict__ dW, int ldw, float* __restrict__ db,
                                                            int M, int N, int K, int Nstore, int Kstore, int tilesK, int ntile,
                                                            int nsplit, int rps, const int32_t* __restrict__ slab_list,
                                                            const int32_t* __restrict__ slab_count, int xgap, int ablate,
                                                            float* __restrict__ scratch) {
  using G = Geo<WK, NI>;
  constexpr int TBN = G::TBN, TBK = G::TBK, NT = G::NT, NW = G::NW, CHA = G::CHA, CH = G::CH, SCW = G::SCW, PA = G::PA, NP = G::NP,
                STAGE = G::STAGE;
  constexpr int PB = NP / NW, PX = NP % NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bool det = nr_fix_on();          // deterministic mode: fixed-point accumulation of the outputs (nr_common.h)
  float* sC = reinterpret_cast<float*>(smem);
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;

  // XCD-aware: the ntile workgroups of one split (same rows of dC / X) sit on one XCD back to back
  const int b = blockIdx.x, xcd = b & 7, local = b >> 3;
  const int split = (local / ntile) * 8 + xcd, tile = local % ntile;
  if (split >= nsplit) return;
  const int tn = tile / tilesK, tk = tile % tilesK;
  const int n0 = tn * TBN, k0 = tk * TBK;
  // Slab mode: only the 32-row slabs listed on the device are contracted (the others are known to be all zero in dC);
  // the list is divided evenly over the splits and this split's part is staged in LDS.
  // Counted mode (slab_list == nullptr, slab_count != nullptr): the operands are dense but only their first *slab_count
  // ROWS exist (compactly stored live rows, zero-filled up to the next multiple of 32): slabs 0 .. ceil(count / 32).
  __shared__ int sSlab[1024];
  const bool slabs = slab_list != nullptr, counted = !slabs && slab_count != nullptr;
  int mbeg = split * rps, nk, kbeg = 0;
  if (slabs || counted) {
    const int total = counted ? (*slab_count + TBM - 1) / TBM : *slab_count, per = (total + nsplit - 1) / nsplit;
    kbeg = split * per;
    nk = min(per, total - kbeg);
    if (nk <= 0) return;
    if (slabs) {
      for (int i = threadIdx.x; i < nk; i += blockDim.x) sSlab[i] = slab_list[kbeg + i];
      __syncthreads();
    }
    mbeg = 0;
  } else {
    const int mend = min(M, mbeg + rps);
    if (mbeg >= mend) return;
    nk = (mend - mbeg) / TBM;                     // exact: M and rps are multiples of TBM (launcher)
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wid / WK, wn = wid % WK;
  const bool extra = wid < PX;
  const int pfirst = wid * PB + min(wid, PX);
  const bool db_wave = (db != nullptr) && (tk == 0) && (wn == wm);

  const bf16_t* src[PB + 1];
  size_t adv[PB + 1];
  int xrow[PB + 1];                                 // X pieces: the lane's row (xgap > 0: logical row m sits at X row m + m / xgap)
#pragma unroll
  for (int t = 0; t < PB + 1; ++t) {
    const int p = pfirst + t;
    xrow[t] = -1;
    if (p < PA) {
      const int P = 64 * p + lane, row = P / CHA, cpos = P - row * CHA;
      const int col = min(n0 + swz_a(row, cpos) * 8, N - 8);
      src[t] = dC + (size_t)(mbeg + row) * ldc + col;
      adv[t] = (size_t)TBM * ldc;
    } else {
      const int P = 64 * (p - PA) + lane, row = min(P / CH, TBM - 1), cpos = P - (P / CH) * CH;
      const int col = min(k0 + swz_b<WK>(row, cpos) * 8, K - 8);
      src[t] = X + (size_t)(mbeg + row) * ldx + col;
      adv[t] = (size_t)TBM * ldx;
      xrow[t] = mbeg + row;
    }
  }
  const uint32_t ginv = xgap > 0 ? (uint32_t)((0x100000000ull + (uint32_t)xgap - 1) / (uint32_t)xgap) : 0u;   // ceil(2^32 / xgap)
  auto issue = [&](int stage, int kt) {
    const size_t sl = slabs ? (size_t)sSlab[kt] : (size_t)(kbeg + kt);      // (kbeg = 0 unless counted)
#pragma unroll
    for (int t = 0; t < PB + 1; ++t)
      if (t < PB || extra) {
        size_t off = sl * adv[t];
        if (xgap > 0 && xrow[t] >= 0) {
          const uint32_t m = (uint32_t)sl * TBM + (uint32_t)xrow[t];
          uint32_t q = __umulhi(m, ginv);            // m / xgap, exact after one correction for m < 2^31
          if (q * (uint32_t)xgap > m) --q;
          off += (size_t)q * ldx;
        }
        dma16(src[t] + off, lds0 + stage * STAGE + (pfirst + t) * 1024);
      }
  };
  auto wait_stages = [&](int stages) {
    if (extra) {
      if (stages >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (PB + 1)) : "memory");
      else if (stages == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PB + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (stages >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PB) : "memory");
      else if (stages == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };

  f32x4 acc[NI][5];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // db: ONE accumulator for all NI row tiles.  Tile i is multiplied by a selector matrix whose row i is all ones,
  // so its column sums land in accumulator row i: accdb[row = tile][col = column inside the tile].
  f32x4 accdb = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bf16_t one = (bf16_t)1.f, zero = (bf16_t)0.f;

  // transposed-read lane geometry (see tn2): 16-lane group g holds m rows 8g..8g+7
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int r0 = 8 * g + q, r1 = r0 + 4, off = 4 * (pp & 1);
  const int a0 = r0 * TBN + off, a1 = r1 * TBN + off, ca = wm * 2 * NI + (pp >> 1);   // A chunk of tile i: ca + 2i
  int ob0[5], ob1[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int chunk = (wn * 80 + j * 16) / 8 + (pp >> 1);
    ob0[j] = TBM * TBN + r0 * TBK + swz_b<WK>(r0, chunk) * 8 + off;
    ob1[j] = TBM * TBN + r1 * TBK + swz_b<WK>(r1, chunk) * 8 + off;
  }

#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) issue(s, s);
  for (int kt = 0; kt < nk; ++kt) {
    wait_stages(min(NS - 2, nk - 1 - kt));
    __builtin_amdgcn_s_barrier();
    if (kt + NS - 1 < nk) issue((kt + NS - 1) % NS, kt + NS - 1);
    const bf16_t* st = reinterpret_cast<const bf16_t*>(smem + (kt % NS) * STAGE);
    bf16x8 bf[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) bf[j] = tr_frag(st + ob0[j], st + ob1[j]);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const bf16x8 af = tr_frag(st + a0 + swz_a(r0, ca + 2 * i) * 8, st + a1 + swz_a(r1, ca + 2 * i) * 8);
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[j], acc[i][j], 0, 0, 0);
      if (db_wave) {
        const bf16_t v = (lane & 15) == i ? one : zero;
        const bf16x8 sel = {v, v, v, v, v, v, v, v};
        accdb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel, af, accdb, 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  if (db_wave) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ti = 4 * (lane >> 4) + e, n = n0 + wm * 16 * NI + ti * 16 + (lane & 15);
      if (ti < NI && n < Nstore) nr_accum(db + n, accdb[e], det);
    }
  }
#pragma unroll
  for (int pass = 0; pass < TBN / 64; ++pass) {
    if (pass) __syncthreads();
    if (wm == pass / (NI / 4)) {
      const int ib = (pass % (NI / 4)) * 4;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int col = wn * 80 + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) sC[(i * 16 + (lane >> 4) * 4 + r) * SCW + col] = acc[ib + i][j][r];
      }
    }
    __syncthreads();
    if (scratch != nullptr) {
      // partial tile of this split -> caller scratch [split][tile][TBN][TBK] with plain 16-byte stores; tn3_reduce_kernel adds
      // the splits up in a fixed order.  (48 splits x 5 tiles of fp32 atomics were 75 MB at the chip's ~1.3 TB/s atomic rate:
      // 58 us of the 0.28 ms QKV weight gradient, half of the user-level launches.)
      float* base = scratch + ((size_t)(split * ntile + tile) * TBN + pass * 64) * TBK;
      for (int u = tid; u < 64 * (TBK / 4); u += NT) {
        const int row = u / (TBK / 4), c4 = (u - row * (TBK / 4)) * 4;
        *reinterpret_cast<f32x4*>(base + (size_t)row * TBK + c4) = *reinterpret_cast<const f32x4*>(sC + row * SCW + c4);
      }
      continue;
    }
    // lanes on consecutive floats: one atomic instruction covers whole 128-byte lines of a dW row
    if (!ablate)
    for (int u = tid; u < 64 * TBK; u += NT) {
      const int row = u / TBK, c = u - row * TBK;
      const int n = n0 + pass * 64 + row, k = k0 + c;
      if (n < Nstore && k < Kstore) nr_accum(dW + (size_t)n * ldw + k, sC[row * SCW + c], det);
    }
  }
}

// dW[n][k] += sum over the splits that ran of scratch[split][tile(n, k)][n % TBN][k % TBK], splits in ascending order (plain
// read-modify-write: every element has one owner, so this half of the weight gradient is bit-reproducible as it stands).
// mode 0: every split ran; 1: slab list (count = slabs); 2: counted rows (count = rows) -- the kernel's own arithmetic.
__global__ __launch_bounds__(256) void tn3_reduce_kernel(const float* __restrict__ scratch, float* __restrict__ dW, int ldw, int Nstore, int Kstore,
                                                         int TBN, int TBK, int tilesK, int ntile, int nsplit, int mode,
                                                         const int32_t* __restrict__ count) {
  __shared__ f32x4 sAcc[4][64];
  int active = nsplit;
  if (mode != 0) {
    const int total = mode == 2 ? (*count + TBM - 1) / TBM : *count, per = (total + nsplit - 1) / nsplit;
    active = per > 0 ? (total + per - 1) / per : 0;
  }
  // a workgroup = 64 output float4s x 4 interleaved quarters of the splits (independent loads in flight; the quarters meet in
  // LDS and are added in a fixed order)
  const int k4n = (Kstore + 3) / 4, o = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const size_t sstride = (size_t)ntile * TBN * TBK;
  for (int u0 = blockIdx.x * 64; u0 < Nstore * k4n; u0 += gridDim.x * 64) {
    const int u = u0 + o;
    const bool ok = u < Nstore * k4n;
    const int n = ok ? u / k4n : 0, k = ok ? (u - n * k4n) * 4 : 0;
    const int tile = (n / TBN) * tilesK + k / TBK;
    const float* p = scratch + ((size_t)tile * TBN + n % TBN) * TBK + k % TBK;
    f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0;
    int sp = sg;
    for (; sp + 4 < active; sp += 8) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (size_t)sp * sstride);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (size_t)(sp + 4) * sstride);
      a0[0] += v0[0]; a0[1] += v0[1]; a0[2] += v0[2]; a0[3] += v0[3];
      a1[0] += v1[0]; a1[1] += v1[1]; a1[2] += v1[2]; a1[3] += v1[3];
    }
    if (sp < active) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (size_t)sp * sstride);
      a0[0] += v0[0]; a0[1] += v0[1]; a0[2] += v0[2]; a0[3] += v0[3];
    }
    sAcc[sg][o] = (f32x4){a0[0] + a1[0], a0[1] + a1[1], a0[2] + a1[2], a0[3] + a1[3]};
    __syncthreads();
    if (sg == 0 && ok) {
      const f32x4 b0 = sAcc[0][o], b1 = sAcc[1][o], b2 = sAcc[2][o], b3 = sAcc[3][o];
      float* d = dW + (size_t)n * ldw + k;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k + e < Kstore) d[e] += (b0[e] + b1[e]) + (b2[e] + b3[e]);
    }
    __syncthreads();
  }
}

// floats of caller scratch the store + reduce epilogue of this geometry needs (0: shape not eligible)
template <int WK, int NI>
size_t scratch_floats_t(int M, int N, int K) {
  using G = Geo<WK, NI>;
  const int tilesN = (N + G::TBN - 1) / G::TBN, tilesK = (K + G::TBK - 1) / G::TBK, ntile = tilesN * tilesK;
  const int resident = 256 * (WK == 4 ? 1 : 2);
  int nsplit = (resident / ntile / 8) * 8;
  if (nsplit < 8) nsplit = 8;
  return (size_t)(nsplit + 8) * ntile * G::TBN * G::TBK;
}

template <int WK, int NI>
int launch_t(const void* dC, int ldc, const void* X, int ldx, float* dW, int ldw, float* db, int M, int N, int K, int Nstore,
             int Kstore, hipStream_t stream, const int32_t* slab_list = nullptr, const int32_t* slab_count = nullptr, int xgap = 0,
             float* scratch = nullptr, size_t scratch_floats = 0) {
  using G = Geo<WK, NI>;
  const int tilesN = (N + G::TBN - 1) / G::TBN, tilesK = (K + G::TBK - 1) / G::TBK, ntile = tilesN * tilesK;
  // ONE round of resident workgroups (256 CUs x 1 or 2), splits a multiple of the 8 XCDs, >= 16 slabs per split.
  // Measured at N=1200, K=304: 1 / 2 / 3 / 4 rounds = 1.03 / 1.14 / 1.25 / 1.34 ms -- every split pays a ring fill and
  // an fp32 atomic epilogue over the whole [N, K] tile, so fewer, longer splits win.
  const int force_rounds = nr_opt(NR_OPT_TN3_ROUNDS);
  const int resident = 256 * (WK == 4 ? 1 : 2), rounds = force_rounds ? force_rounds : 1;
  int nsplit = (resident * rounds / ntile / 8) * 8;     // rounded down: never a few workgroups left for an extra round
  if (nsplit < 8) nsplit = 8;
  int rps = (M + nsplit - 1) / nsplit;
  rps = ((rps + TBM - 1) / TBM) * TBM;
  if (rps < 16 * TBM) rps = 16 * TBM;
  nsplit = (M + rps - 1) / rps;
  const int grid = ((nsplit + 7) / 8) * 8 * ntile;
  auto kern = gemm_tn3_kernel<WK, NI>;
  NR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::SMEM));
  const bool counted = slab_list == nullptr && slab_count != nullptr;           // dense rows, device-side row count
  if (slab_list != nullptr && (M / TBM + nsplit - 1) / nsplit > 1024) slab_list = nullptr;   // a split's list must fit its LDS stage
  // store + reduce epilogue when the caller brought enough scratch (not with forced extra rounds).  Deterministic mode takes it
  // too: the splits are added in a fixed order by one owner per element, the fixed-point table then only carries db.
  const bool use_scratch = scratch != nullptr && scratch_floats >= (size_t)nsplit * ntile * G::TBN * G::TBK && !force_rounds &&
                           !nr_opt(NR_OPT_TN3_ATOMIC) && (((uintptr_t)scratch) & 15) == 0;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(G::NT), G::SMEM, stream, (const bf16_t*)dC, ldc, (const bf16_t*)X, ldx, dW, ldw, db, M, N,
                     K, Nstore, Kstore, tilesK, ntile, nsplit, rps, slab_list, (slab_list || counted) ? slab_count : nullptr, xgap,
                     nr_opt(NR_OPT_TN3_ABLATE), use_scratch ? scratch : nullptr);
  if (use_scratch) {
    const int mode = slab_list != nullptr ? 1 : (counted ? 2 : 0);
    int rg = (Nstore * ((Kstore + 3) / 4) + 63) / 64;
    if (rg > 4096) rg = 4096;
    hipLaunchKernelGGL(tn3_reduce_kernel, dim3(rg), dim3(256), 0, stream, (const float*)scratch, dW, ldw, Nstore, Kstore, G::TBN, G::TBK, tilesK,
                       ntile, nsplit, mode, mode ? slab_count : nullptr);
  }
  NR_CHECK_LAUNCH();
  return NR_OK;
}

// rows a multiple of the 32-row slab, at least 8 columns on both sides (tail clamping), 16-byte aligned rows
bool eligible(int ldc, int ldx, int M, int N, int K) {
  const bool off = nr_opt(NR_OPT_NO_TN3) != 0;
  // (round 1 kept the user level, M = 25 600, on tn2: 0.31 vs 0.18 ms then; with one round of splits and >= 16 slabs per
  //  split tn3 now wins there too: 0.067 vs 0.089 ms for [25 600, 1200] x [25 600, 400])
  return !off && M % TBM == 0 && M >= nr_opt(NR_OPT_TN3_MIN_M) && N >= 8 && K >= 8 && N % 8 == 0 && K % 8 == 0 && ldc % 8 == 0 && ldx % 8 == 0;
}
int launch(const void* dC, int ldc, const void* X, int ldx, float* dW, int ldw, float* db, int M, int N, int K, int Nstore,
           int Kstore, hipStream_t stream, const int32_t* slab_list = nullptr, const int32_t* slab_count = nullptr, int xgap = 0,
           float* scratch = nullptr, size_t sf = 0) {
  const int force = nr_opt(NR_OPT_TN3_WK);
  const int force_ni = nr_opt(NR_OPT_TN3_NI);
  const bool wide = force ? force == 4 : (K > 160 && ((K + 319) / 320) * 320 * 100 <= K * 115);
  if (wide) {
    const bool big = force_ni ? force_ni == 8 : N > 256;
    if (big) return launch_t<4, 8>(dC, ldc, X, ldx, dW, ldw, db, M, N, K, Nstore, Kstore, stream, slab_list, slab_count, xgap, scratch, sf);
    return launch_t<4, 4>(dC, ldc, X, ldx, dW, ldw, db, M, N, K, Nstore, Kstore, stream, slab_list, slab_count, xgap, scratch, sf);
  }
  return launch_t<2, 4>(dC, ldc, X, ldx, dW, ldw, db, M, N, K, Nstore, Kstore, stream, slab_list, slab_count, xgap, scratch, sf);
}
size_t scratch_floats(int M, int N, int K) {
  if (N < 8 || K < 8) return 0;
  const int force = nr_opt(NR_OPT_TN3_WK);
  const int force_ni = nr_opt(NR_OPT_TN3_NI);
  const bool wide = force ? force == 4 : (K > 160 && ((K + 319) / 320) * 320 * 100 <= K * 115);
  if (wide) return ((force_ni ? force_ni == 8 : N > 256)) ? scratch_floats_t<4, 8>(M, N, K) : scratch_floats_t<4, 4>(M, N, K);
  return scratch_floats_t<2, 4>(M, N, K);
}
}  // namespace tn3

template <typename T, int KIND, int EPI>
int launch_nt_t(const RowSrc& A, const void* B, int ldb, int M, int N, int K, const EpiArgs& ep, hipStream_t stream) {
  const int tilesM = (M + BM - 1) / BM, tilesN = (N + BN - 1) / BN;
  const int grid = ((tilesM + 7) / 8) * 8 * tilesN;
  auto kern = gemm_nt_kernel<T, KIND, EPI>;
  const size_t smem = smem_bytes<T>();
  int rc = set_smem(kern, smem);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), smem, stream, A, (const T*)B, ldb, M, N, K, ep, tilesM, tilesN);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

template <typename T, int KIND>
int launch_nt_k(const RowSrc& A, const void* B, int ldb, int M, int N, int K, int epi, const EpiArgs& ep, hipStream_t s) {
  switch (epi) {
    case EPI_STORE: return launch_nt_t<T, KIND, EPI_STORE>(A, B, ldb, M, N, K, ep, s);
    case EPI_STORE_TANH: return launch_nt_t<T, KIND, EPI_STORE_TANH>(A, B, ldb, M, N, K, ep, s);
    case EPI_POOLBWD: return launch_nt_t<T, KIND, EPI_POOLBWD>(A, B, ldb, M, N, K, ep, s);
    case EPI_SCATTER: return launch_nt_t<T, KIND, EPI_SCATTER>(A, B, ldb, M, N, K, ep, s);
  }
  nr_set_error("gemm_nt: bad epilogue %d", epi);
  return NR_ERR_ARG;
}

template <typename T>
int launch_nt_d(const RowSrc& A, const void* B, int ldb, int M, int N, int K, int epi, const EpiArgs& ep, hipStream_t s) {
  switch (A.kind) {
    case ROWS_DENSE: return launch_nt_k<T, ROWS_DENSE>(A, B, ldb, M, N, K, epi, ep, s);
    case ROWS_GATHER: return launch_nt_k<T, ROWS_GATHER>(A, B, ldb, M, N, K, epi, ep, s);
    case ROWS_IM2COL3:
      if (epi == EPI_STORE) return launch_nt_t<T, ROWS_IM2COL3, EPI_STORE>(A, B, ldb, M, N, K, ep, s);
      if (epi == EPI_STORE_TANH) return launch_nt_t<T, ROWS_IM2COL3, EPI_STORE_TANH>(A, B, ldb, M, N, K, ep, s);
      break;
  }
  nr_set_error("gemm_nt: unsupported row source %d / epilogue %d", A.kind, epi);
  return NR_ERR_ARG;
}

template <typename T, int KIND>
int launch_tn_t(const void* dC, int ldc, const RowSrc& A, float* dW, int ldw, float* db, int M, int N, int K, int Nstore,
                int Kstore, hipStream_t stream) {
  const int tilesN = (N + BM - 1) / BM, tilesK = (K + BN - 1) / BN;
  // split the contraction so that the grid has ~8 blocks per CU worth of work at most
  int splits = (256 * 4 + tilesN * tilesK - 1) / (tilesN * tilesK);
  int rps = (M + splits - 1) / splits;
  rps = ((rps + BK - 1) / BK) * BK;
  if (rps < 8 * BK) rps = 8 * BK;
  splits = (M + rps - 1) / rps;
  auto kern = gemm_tn_kernel<T, KIND>;
  const size_t smem = smem_bytes<T>();
  int rc = set_smem(kern, smem);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3(tilesN * tilesK, splits), dim3(NTHR), smem, stream, (const T*)dC, ldc, A, dW, ldw, db, M, N,
                     K, Nstore, Kstore, tilesK, rps);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

template <typename T>
int launch_tn_d(const void* dC, int ldc, const RowSrc& A, float* dW, int ldw, float* db, int M, int N, int K, int Nstore,
                int Kstore, hipStream_t s) {
  switch (A.kind) {
    case ROWS_DENSE: return launch_tn_t<T, ROWS_DENSE>(dC, ldc, A, dW, ldw, db, M, N, K, Nstore, Kstore, s);
    case ROWS_GATHER: return launch_tn_t<T, ROWS_GATHER>(dC, ldc, A, dW, ldw, db, M, N, K, Nstore, Kstore, s);
    case ROWS_IM2COL3: return launch_tn_t<T, ROWS_IM2COL3>(dC, ldc, A, dW, ldw, db, M, N, K, Nstore, Kstore, s);
  }
  nr_set_error("gemm_tn: bad row source %d", A.kind);
  return NR_ERR_ARG;
}

}  // namespace

// im2col rows of the k=3 convolution, [M, 3*Dp]: row m = [x[m-1] | x[m] | x[m+1]] inside its title (zeros across the
// title borders).  One thread reads and hashes a source chunk x[m][d..] ONCE and stores it into the (up to) three rows
// that contain it; the border zeros are written by the token that sits at the border.
template <typename T>
__global__ __launch_bounds__(256) void im2col3_materialize_kernel(RowSrc A, T* __restrict__ out, int ldo, int M) {
  constexpr int CH = 16 / (int)sizeof(T);
  const int Dp = A.ld, cpr = Dp / CH, T_ = A.Tlen;
  const size_t total = (size_t)M * cpr;
  const uint4 zero = make_uint4(0, 0, 0, 0);
  for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(u / cpr), d = (int)(u - (size_t)m * cpr) * CH;
    const int blk = m / T_, t = m - blk * T_;
    // centre tap (k = Dp + d) of row m is x[m][d] itself
    const uint4 v = load_rows_chunk<T, ROWS_IM2COL3>(A, m, Dp + d, M, 3 * Dp);
    T* o = out + (size_t)m * ldo + d;
    *reinterpret_cast<uint4*>(o + Dp) = v;                                       // row m, tap 1
    if (t + 1 < T_) *reinterpret_cast<uint4*>(o + ldo) = v;                      // row m+1, tap 0
    else *reinterpret_cast<uint4*>(o + 2 * Dp) = zero;                           // last token: its own tap 2 is empty
    if (t > 0) *reinterpret_cast<uint4*>(o - ldo + 2 * Dp) = v;                  // row m-1, tap 2
    else *reinterpret_cast<uint4*>(o) = zero;                                    // first token: its own tap 0 is empty
  }
}

// The same, one workgroup per title, for callers that say which titles anybody needs: a title farther than `margin` titles
// from every needed one lies in no row tile the projection computes and in no 32-row slab the weight gradient contracts
// (both cover at most margin titles), so its im2col rows are never read and are not written.
template <typename T>
__global__ __launch_bounds__(256) void im2col3_title_kernel(RowSrc A, T* __restrict__ out, int ldo, int M, const int32_t* __restrict__ needed,
                                                            int n, int margin) {
  constexpr int CH = 16 / (int)sizeof(T);
  const int Dp = A.ld, cpr = Dp / CH, T_ = A.Tlen, blk = blockIdx.x;
  __shared__ int near;
  if (threadIdx.x == 0) near = 0;
  __syncthreads();
  for (int t = max(0, blk - margin) + (int)threadIdx.x; t <= min(n - 1, blk + margin); t += 256)
    if (needed[t] != 0) near = 1;                        // benign race: every writer stores 1
  __syncthreads();
  if (!near) return;
  const uint4 zero = make_uint4(0, 0, 0, 0);
  for (int u = threadIdx.x; u < T_ * cpr; u += 256) {
    const int t = u / cpr, d = (u - t * cpr) * CH, m = blk * T_ + t;
    if (m >= M) break;
    const uint4 v = load_rows_chunk<T, ROWS_IM2COL3>(A, m, Dp + d, M, 3 * Dp);
    T* o = out + (size_t)m * ldo + d;
    *reinterpret_cast<uint4*>(o + Dp) = v;
    if (t + 1 < T_) *reinterpret_cast<uint4*>(o + ldo) = v;
    else *reinterpret_cast<uint4*>(o + 2 * Dp) = zero;
    if (t > 0) *reinterpret_cast<uint4*>(o - ldo + 2 * Dp) = v;
    else *reinterpret_cast<uint4*>(o) = zero;
  }
}

// Token rows of the k = 3 convolution (nr_launch_conv_rows): one workgroup per title writes its T gathered (+ dropped-out)
// rows and BOTH zero rows around them (a neighbour that is skipped would not write the shared one).
template <typename T>
__global__ __launch_bounds__(256) void conv_rows_kernel(RowSrc A, T* __restrict__ out, int n, const int32_t* __restrict__ needed, int margin) {
  constexpr int CH = 16 / (int)sizeof(T);
  const int Dp = A.ld, cpr = Dp / CH, T_ = A.Tlen, blk = blockIdx.x;
  if (needed != nullptr) {
    __shared__ int near;
    if (threadIdx.x == 0) near = 0;
    __syncthreads();
    for (int t = max(0, blk - margin) + (int)threadIdx.x; t <= min(n - 1, blk + margin); t += 256)
      if (needed[t] != 0) near = 1;                      // benign race: every writer stores 1
    __syncthreads();
    if (!near) return;
  }
  const uint4 zero = make_uint4(0, 0, 0, 0);
  T* o0 = out + (size_t)blk * (T_ + 1) * Dp;               // the zero row in front of this title
  for (int u = threadIdx.x; u < (T_ + 2) * cpr; u += 256) {
    const int r = u / cpr, d = (u - r * cpr) * CH;         // r = 0: leading zero row, 1..T: tokens, T+1: trailing zero row
    uint4 v = zero;
    if (r >= 1 && r <= T_) v = load_rows_chunk<T, ROWS_IM2COL3>(A, blk * T_ + r - 1, Dp + d, n * T_, 3 * Dp);   // centre tap = the token itself
    *reinterpret_cast<uint4*>(o0 + (size_t)r * Dp + d) = v;
  }
}

// Gathered rows, one workgroup per sequence of L tokens, under "needed" flags: an all-padding sequence (every id 0, table
// row 0 zero: *keep_all == 0) has no live row for the projection to read, and if no needed sequence lies within `margin`
// sequences no 32-row slab of the weight-gradient GEMM reaches it either -- its rows are not written.
template <typename T>
__global__ __launch_bounds__(256) void gather_title_kernel(RowSrc A, T* __restrict__ out, int ldo, int M, int K, int L,
                                                           const int32_t* __restrict__ needed, int n, int margin,
                                                           const int32_t* __restrict__ keep_all) {
  constexpr int CH = 16 / (int)sizeof(T);
  const int cpr = K / CH, blk = blockIdx.x;
  __shared__ int near;
  if (threadIdx.x == 0) near = *keep_all != 0 ? 1 : 0;
  __syncthreads();
  for (int t = max(0, blk - margin) + (int)threadIdx.x; t <= min(n - 1, blk + margin); t += 256)
    if (needed[t] != 0) near = 1;                        // benign race: every writer stores 1
  for (int t = threadIdx.x; t < L; t += 256)
    if (blk * L + t < M && A.ids[(size_t)(blk * L + t) * A.ids_stride] != 0) near = 1;
  __syncthreads();
  if (!near) return;
  for (int u = threadIdx.x; u < L * cpr; u += 256) {
    const int t = u / cpr, k = (u - t * cpr) * CH, m = blk * L + t;
    if (m >= M) break;
    *reinterpret_cast<uint4*>(out + (size_t)m * ldo + k) = load_rows_chunk<T, ROWS_GATHER>(A, m, k, M, K);
  }
}

// Compact row storage (news level, training): ONLY the live rows of the gathered + dropped-out operand are materialised,
// in live-list order: out[k] = dropout(table[ids[k]]) with the dropout counter of the ORIGINAL row rows[k] (the draws are
// those of the dense computation).  A padding token gathers the zero row and no kernel downstream needs it: the
// projection substitutes the bias, the weight gradient would multiply a zero row.  Rows count .. roundup32(count) are
// zero-filled (the weight-gradient GEMM contracts whole 32-row slabs).  Four independent (index -> table -> store) chains
// per thread: the gather is latency bound, not bandwidth bound, with one (rows_materialize ran at 2.9 TB/s).
template <typename T>
// hist (optional, [V], cleared by the compaction kernel): occurrences per token id, counted by the thread that holds a row's
// first chunk -- the backward's id sort then starts at its scan (the histogram pass was 14 us of dependent launch there; here
// its atomics disappear under the row traffic)
__global__ __launch_bounds__(256) void gather_live_rows_kernel(RowSrc A, T* __restrict__ out, int ldo, int K, const int32_t* __restrict__ count,
                                                               const int32_t* __restrict__ rows, const int32_t* __restrict__ ids,
                                                               int32_t* __restrict__ hist, int V) {
  constexpr int CH = 16 / (int)sizeof(T), U = 4;
  const int cpr = K / CH, n = *count, nz = (n + 31) / 32 * 32;
  const uint32_t total = (uint32_t)nz * (uint32_t)cpr, stride = gridDim.x * 256u;
  const uint32_t inv_cpr = (uint32_t)((0x100000000ull + (uint32_t)cpr - 1) / (uint32_t)cpr);
  for (uint32_t u0 = blockIdx.x * 256u + threadIdx.x; u0 < total; u0 += U * stride) {
    int k[U], c[U], row[U], id[U];
    Chunk v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const uint32_t u = u0 + (uint32_t)j * stride;
      uint32_t q = __umulhi(u, inv_cpr);                 // u / cpr (exact after one correction, u < 2^31)
      if (q * (uint32_t)cpr > u) --q;
      k[j] = (int)q;
      c[j] = (int)(u - q * (uint32_t)cpr) * CH;
      const bool ok = u < total && k[j] < n;
      row[j] = ok ? rows[k[j]] : -1;
      id[j] = ok ? ids[k[j]] : 0;
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      v[j].u = make_uint4(0, 0, 0, 0);
      if (row[j] >= 0) v[j].u = *reinterpret_cast<const uint4*>((const T*)A.base + (size_t)id[j] * A.ld + c[j]);
      if (hist != nullptr && row[j] >= 0 && c[j] == 0 && (uint32_t)id[j] < (uint32_t)V) atomicAdd(hist + id[j], 1);
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const uint32_t u = u0 + (uint32_t)j * stride;
      if (u >= total) continue;
      if (row[j] >= 0 && A.drop.thresh && (v[j].u.x | v[j].u.y | v[j].u.z | v[j].u.w) != 0u)
        drop_chunk<T>(v[j], A.drop, (uint32_t)row[j] * (uint32_t)A.Dtrue + (uint32_t)c[j], c[j], A.Dtrue);
      *reinterpret_cast<uint4*>(out + (size_t)k[j] * ldo + c[j]) = v[j].u;
    }
  }
}

// Row compaction for the table-gradient GEMM: rows with token id 0 add nothing (padding_idx), and in a MIND-shaped
// batch they are ~70 % of all rows (zero-padded title tails, empty history slots).  One pass, no host round trip:
// 256 rows per workgroup, order kept inside a workgroup, workgroups append through one atomic counter.
namespace {
// ws: [0] live count, [1] dead count, [4 .. 4+M) live rows, [4+M .. 4+2M) their ids, [4+2M .. 4+3M) dead rows (if DEAD).
// all_live (device flag, may be null): when set every row counts as live (table row 0 is not zero, so a padding token
// does not gather a zero row and the forward may not treat it as one).
// posmap (optional, [M]): position of row m in the live list, -1 for a dead row (the inverse of ws[4 ..]).
template <bool DEAD>
__global__ __launch_bounds__(256) void compact_rows_kernel(const int32_t* __restrict__ ids, int stride, int M, int32_t* __restrict__ ws,
                                                           const int32_t* __restrict__ all_live, int32_t* __restrict__ posmap) {
  constexpr int RPT = 4;                                  // 1024 rows per workgroup: 4 batches of 256, order preserved
  __shared__ int wave_cnt[RPT][4];
  __shared__ int base, dbase;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * 256 * RPT;
  const bool keep_all = all_live != nullptr && *all_live != 0;
  int id[RPT];
  uint64_t bal[RPT];
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int m = row0 + j * 256 + tid;
    id[j] = m < M ? ids[(size_t)m * stride] : 0;
    bal[j] = __ballot(m < M && (id[j] != 0 || keep_all));
    if (lane == 0) wave_cnt[j][wid] = __popcll(bal[j]);
  }
  __syncthreads();
  if (tid == 0) {
    int nl = 0;
    for (int j = 0; j < RPT; ++j) nl += wave_cnt[j][0] + wave_cnt[j][1] + wave_cnt[j][2] + wave_cnt[j][3];
    base = atomicAdd(ws, nl);
    if (DEAD) dbase = atomicAdd(ws + 1, min(256 * RPT, M - row0) - nl);
  }
  __syncthreads();
  int before = 0;                                         // live rows of this workgroup in front of (j, tid)
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int m = row0 + j * 256 + tid;
    int pos = before + __popcll(bal[j] & ((1ull << lane) - 1ull));
    for (int w = 0; w < wid; ++w) pos += wave_cnt[j][w];
    const bool live = (bal[j] >> lane) & 1ull;
    if (live) {
      ws[4 + base + pos] = m;
      ws[4 + M + base + pos] = id[j];
    } else if (DEAD && m < M) {
      ws[4 + 2 * M + dbase + (j * 256 + tid - pos)] = m;
    }
    if (posmap != nullptr && m < M) posmap[m] = live ? base + pos : -1;
    before += wave_cnt[j][0] + wave_cnt[j][1] + wave_cnt[j][2] + wave_cnt[j][3];
  }
}

// qkv rows of padding tokens: x = 0 there, so the projection is the bias itself.  A thread owns one 16-byte column chunk
// (its bias values are converted once); a workgroup walks batches of 64 row numbers staged through LDS.
__global__ __launch_bounds__(256) void bias_rows_kernel(bf16_t* __restrict__ C, int ldc, int N, const float* __restrict__ bias,
                                                        const int32_t* __restrict__ rows, const int32_t* __restrict__ count,
                                                        const uint32_t* __restrict__ tmask, int L) {
  __shared__ int srow[64];
  const int cpr = N / 8, n = *count, tid = threadIdx.x;
  for (int c0 = 0; c0 < cpr; c0 += 256) {
    const int c = c0 + tid;
    Chunk o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.h[e] = (bf16_t)((bias && c < cpr) ? bias[c * 8 + e] : 0.f);
    for (int r0 = blockIdx.x * 64; r0 < n; r0 += gridDim.x * 64) {
      __syncthreads();
      if (tid < 64) {
        int r = r0 + tid < n ? rows[r0 + tid] : -1;
        // sequences made of padding tokens only are handled inside the attention kernels: their rows stay unwritten
        if (r >= 0 && tmask != nullptr && tmask[r / L] == 0) r = -1;
        srow[tid] = r;
      }
      __syncthreads();
      if (c < cpr) {
#pragma unroll 8
        for (int i = 0; i < 64; ++i) {
          const int r = srow[i];
          if (r >= 0) *reinterpret_cast<uint4*>(C + (size_t)r * ldc + c * 8) = o.u;
        }
      }
    }
  }
}

// Forward flavour of compact_rows_kernel in ONE launch: 4080 .. 4096 rows (a whole number of sequences) per workgroup -- a
// fifth of the same-address atomics on the live counter that paced the 1024-row version (24 us for 3.4 MB of ids) -- with the
// "table row 0 is not zero" test done by every workgroup for itself (600 bytes) and the per-sequence live-token masks built
// from the flags it has in LDS anyway (both used to be launches of their own).
__global__ __launch_bounds__(256) void compact_rows_fwd_kernel(const int32_t* __restrict__ ids, int M, int n, int L,
                                                               const bf16_t* __restrict__ row0, int cols, int32_t* __restrict__ ws,
                                                               int32_t* __restrict__ posmap, uint32_t* __restrict__ tmask, int rows_per_wg,
                                                               int32_t* __restrict__ zero4, int32_t* __restrict__ zhist, int nhist) {
  constexpr int RPT = 16;
  __shared__ int wave_cnt[RPT][4];
  __shared__ int base, dbase;
  __shared__ unsigned char sLive[256 * RPT];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  bool nz = false;
  for (int c = tid; c < cols; c += 256) nz |= (float)row0[c] != 0.f;
  const bool keep_all = __syncthreads_or(nz) != 0;
  if (blockIdx.x == 0 && tid == 0) ws[2] = keep_all ? 1 : 0;
  if (zero4 != nullptr && blockIdx.x == 0 && tid < 4) zero4[tid] = 0;       // the counters of a list a LATER kernel of the call builds
  for (int i = blockIdx.x * 256 + tid; i < nhist; i += gridDim.x * 256) zhist[i] = 0;   // ... and the id histogram the row gather fills
  const int r0 = blockIdx.x * rows_per_wg, r1 = min(M, r0 + rows_per_wg);
  int id[RPT];
  uint64_t bal[RPT];
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int m = r0 + j * 256 + tid;
    id[j] = m < r1 ? ids[m] : 0;
  }
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int m = r0 + j * 256 + tid;
    const bool live = m < r1 && (id[j] != 0 || keep_all);
    bal[j] = __ballot(live);
    if (lane == 0) wave_cnt[j][wid] = __popcll(bal[j]);
    sLive[j * 256 + tid] = live ? 1 : 0;
  }
  __syncthreads();
  if (tid == 0) {
    int nl = 0;
    for (int j = 0; j < RPT; ++j) nl += wave_cnt[j][0] + wave_cnt[j][1] + wave_cnt[j][2] + wave_cnt[j][3];
    base = atomicAdd(ws, nl);
    dbase = atomicAdd(ws + 1, max(r1 - r0, 0) - nl);
  }
  __syncthreads();
  int before = 0;                                         // live rows of this workgroup in front of (j, tid)
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int m = r0 + j * 256 + tid;
    int pos = before + __popcll(bal[j] & ((1ull << lane) - 1ull));
    for (int w = 0; w < wid; ++w) pos += wave_cnt[j][w];
    const bool live = (bal[j] >> lane) & 1ull;
    if (live) {
      ws[4 + base + pos] = m;
      ws[4 + M + base + pos] = id[j];
    } else if (m < r1) {
      ws[4 + 2 * M + dbase + (j * 256 + tid - pos)] = m;
    }
    if (posmap != nullptr && m < r1) posmap[m] = live ? base + pos : -1;
    before += wave_cnt[j][0] + wave_cnt[j][1] + wave_cnt[j][2] + wave_cnt[j][3];
  }
  if (tmask != nullptr) {                                 // (rows_per_wg is a multiple of L then)
    const int s0 = r0 / L;
    for (int sq = tid; sq < rows_per_wg / L && s0 + sq < n; sq += 256) {
      uint32_t mk = 0;
      for (int t = 0; t < L; ++t) mk |= (uint32_t)sLive[sq * L + t] << t;
      tmask[s0 + sq] = keep_all ? 0xffffffffu : mk;
    }
  }
}

__global__ __launch_bounds__(256) void title_mask_kernel(const int32_t* __restrict__ ids, int n, int L, const int32_t* __restrict__ all_live,
                                                         uint32_t* __restrict__ tmask) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t m = 0;
  if (*all_live != 0) {
    m = 0xffffffffu;
  } else {
    for (int t = 0; t < L; ++t) m |= (ids[(size_t)i * L + t] != 0 ? 1u : 0u) << t;
  }
  tmask[i] = m;
}

__global__ void row0_flag_kernel(const bf16_t* __restrict__ row0, int cols, int32_t* __restrict__ flag) {
  bool nz = false;
  for (int c = threadIdx.x; c < cols; c += blockDim.x) nz |= (float)row0[c] != 0.f;
  const int any = __syncthreads_or(nz);
  if (threadIdx.x == 0) *flag = any;
}
}  // namespace

int nr_fix_set_gemm(const NrFixTable* t, hipStream_t s) {
  NR_CHECK_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_nr_fix), t, sizeof(NrFixTable), 0, hipMemcpyHostToDevice, s));
  return NR_OK;
}

int nr_launch_compact_rows(const int32_t* ids, int ids_stride, int M, int32_t* ws, hipStream_t stream) {
  NR_CHECK_ARG(ids != nullptr && ws != nullptr && M > 0 && ids_stride >= 1, "compact_rows: bad arguments");
  NR_CHECK_HIP(hipMemsetAsync(ws, 0, 4 * sizeof(int32_t), stream));
  NrProfScope ps(stream, "compact_rows[M=%d]", M);
  hipLaunchKernelGGL(compact_rows_kernel<false>, dim3((M + 1023) / 1024), dim3(256), 0, stream, ids, ids_stride, M, ws, (const int32_t*)nullptr,
                     (int32_t*)nullptr);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

// Live rows grouped by token id (counting sort on the device).
namespace {
__global__ __launch_bounds__(256) void id_hist_kernel(const int32_t* __restrict__ count, const int32_t* __restrict__ ids, int V,
                                                      int32_t* __restrict__ hist) {
  const int n = *count;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int id = ids[i];
    if (id >= 0 && id < V) atomicAdd(hist + id, 1);
  }
}
// exclusive scan of hist[0 .. V) in place, by one workgroup (V is a vocabulary size: tens of thousands).  in_lds: the
// whole histogram is staged in LDS by coalesced, independent loads (the strided per-thread runs over global memory were a
// chain of dependent-latency loads: 30 us of the sort's 80)
__global__ __launch_bounds__(1024) void id_scan_kernel(const int32_t* hist, int32_t* out, int V, int in_lds) {   // out may be hist itself
  __shared__ int sSum[1024];
  extern __shared__ __attribute__((aligned(16))) int sH[];
  const int tid = threadIdx.x, per = (V + 1023) / 1024, v0 = tid * per, v1 = min(V, v0 + per);
  if (in_lds) {
    for (int base = 0; base < V; base += 8 * 1024) {
      int x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int v = base + u * 1024 + tid;
        x[u] = v < V ? hist[v] : 0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int v = base + u * 1024 + tid;
        if (v < V) sH[v] = x[u];
      }
    }
    __syncthreads();
  }
  int c = 0;
  for (int v = v0; v < v1; ++v) c += in_lds ? sH[v] : hist[v];
  sSum[tid] = c;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int x = tid >= o ? sSum[tid - o] : 0;
    __syncthreads();
    sSum[tid] += x;
    __syncthreads();
  }
  int run = sSum[tid] - c;
  if (in_lds) {
    for (int v = v0; v < v1; ++v) {
      const int h = sH[v];
      sH[v] = run;
      run += h;
    }
    __syncthreads();
    for (int v = tid; v < V; v += 1024) out[v] = sH[v];
  } else {
    for (int v = v0; v < v1; ++v) {
      const int h = hist[v];
      out[v] = run;
      run += h;
    }
  }
}
__global__ __launch_bounds__(256) void id_scatter_kernel(const int32_t* __restrict__ count, const int32_t* __restrict__ rows,
                                                         const int32_t* __restrict__ ids, int V, int32_t* __restrict__ cursor,
                                                         int32_t* __restrict__ rows_out, int32_t* __restrict__ ids_out,
                                                         int32_t* __restrict__ k_out) {
  const int n = *count;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int id = ids[i];
    if (id >= 0 && id < V) {
      const int pos = atomicAdd(cursor + id, 1);
      rows_out[pos] = rows[i];
      ids_out[pos] = id;
      if (k_out != nullptr) k_out[pos] = i;          // position in the (unsorted) live list: the row of a compactly stored operand
    }
  }
}
}  // namespace

int nr_launch_sort_rows_by_id(const int32_t* count, const int32_t* rows, const int32_t* ids, int Mmax, int table_rows, int32_t* hist,
                              int32_t* rows_out, int32_t* ids_out, hipStream_t stream, int32_t* k_out, int32_t* counted_cursor) {
  NR_CHECK_ARG(count && rows && ids && hist && rows_out && ids_out && Mmax > 0 && table_rows > 0, "sort_rows_by_id: bad arguments");
  NrProfScope ps(stream, "sort_rows_by_id[Mmax=%d,V=%d]", Mmax, table_rows);
  int grid = (Mmax + 255) / 256;
  if (grid > 2048) grid = 2048;
  // counted_cursor: `hist` already holds the histogram of exactly these ids (the forward's row gather counted them); it is left
  // intact -- the call may be repeated -- and the scan goes into counted_cursor [table_rows]
  int32_t* cursor = counted_cursor != nullptr ? counted_cursor : hist;
  if (counted_cursor == nullptr) {
    NR_CHECK_HIP(hipMemsetAsync(hist, 0, (size_t)table_rows * sizeof(int32_t), stream));
    hipLaunchKernelGGL(id_hist_kernel, dim3(grid), dim3(256), 0, stream, count, ids, table_rows, hist);
  }
  const int in_lds = table_rows <= 36 * 1024 ? 1 : 0;
  if (in_lds) {
    NR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(id_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 36 * 1024 * 4));
  }
  hipLaunchKernelGGL(id_scan_kernel, dim3(1), dim3(1024), in_lds ? (size_t)((table_rows + 3) / 4) * 16 : 0, stream, (const int32_t*)hist, cursor,
                     table_rows, in_lds);
  hipLaunchKernelGGL(id_scatter_kernel, dim3(grid), dim3(256), 0, stream, count, rows, ids, table_rows, cursor, rows_out, ids_out, k_out);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

// Forward flavour: ws int32 [3*M + n + 4]; ws[2] = 1 when row 0 of the (bf16) table is not all zero -> every row live;
// ws[4 + 3M + i] = bit mask of sequence i (L <= 32): bit t set = token t is live.
int nr_launch_compact_rows_fwd(const int32_t* ids, int M, int n, int L, const void* table_row0, int cols, int32_t* ws,
                               hipStream_t stream, int32_t* posmap, int32_t* zero4, int32_t* zhist, int nhist) {
  NR_CHECK_ARG(ids != nullptr && ws != nullptr && M > 0 && table_row0 != nullptr && n * L == M, "compact_rows_fwd: bad arguments");
  NR_CHECK_HIP(hipMemsetAsync(ws, 0, 4 * sizeof(int32_t), stream));
  NrProfScope ps(stream, "compact_rows[M=%d]", M);
  const int rpw = L <= 32 ? (4096 / L) * L : 4096;       // whole sequences per workgroup when their token masks are wanted
  hipLaunchKernelGGL(compact_rows_fwd_kernel, dim3((M + rpw - 1) / rpw), dim3(256), 0, stream, ids, M, n, L, (const bf16_t*)table_row0, cols, ws,
                     posmap, L <= 32 ? reinterpret_cast<uint32_t*>(ws + 4 + 3 * (size_t)M) : nullptr, rpw, zero4, zhist, zhist ? nhist : 0);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_launch_bias_rows(void* C, int ldc, int N, const float* bias, const int32_t* rows, const int32_t* count, int max_rows,
                        const uint32_t* tmask, int L, hipStream_t stream) {
  NR_CHECK_ARG(N % 8 == 0 && ldc % 8 == 0, "bias_rows: N=%d / ldc=%d must be multiples of 8", N, ldc);
  NrProfScope ps(stream, "bias_rows[max=%d,N=%d]", max_rows, N);
  size_t grid = ((size_t)max_rows + 63) / 64;
  if (grid > 256 * 8) grid = 256 * 8;
  hipLaunchKernelGGL(bias_rows_kernel, dim3((unsigned)grid), dim3(256), 0, stream, (bf16_t*)C, ldc, N, bias, rows, count, tmask, L);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_launch_conv_rows(int dtype, const RowSrc& A, void* out, int n, hipStream_t stream, const int32_t* needed, int margin) {
  NR_CHECK_ARG(A.kind == ROWS_IM2COL3 && out != nullptr && n > 0 && A.ld % nr_chunk(dtype) == 0, "conv_rows: bad arguments");
  NrProfScope ps(stream, needed ? "conv_rows_needed[%s,n=%d,T=%d,Dp=%d]" : "conv_rows[%s,n=%d,T=%d,Dp=%d]", dtype == NR_BF16 ? "bf16" : "f32", n,
                 A.Tlen, A.ld);
  if (dtype == NR_BF16) hipLaunchKernelGGL(conv_rows_kernel<bf16_t>, dim3(n), dim3(256), 0, stream, A, (bf16_t*)out, n, needed, margin);
  else hipLaunchKernelGGL(conv_rows_kernel<float>, dim3(n), dim3(256), 0, stream, A, (float*)out, n, needed, margin);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_launch_rows_materialize(int dtype, const RowSrc& A, void* out, int ldo, int M, int K, hipStream_t stream, const int32_t* needed,
                               int margin, int L, const int32_t* keep_all) {
  const int ch = nr_chunk(dtype);
  NR_CHECK_ARG((A.kind == ROWS_GATHER || A.kind == ROWS_IM2COL3) && K % ch == 0 && ldo % ch == 0 && ldo >= K, "rows_materialize: bad arguments");
  if (needed != nullptr && A.kind == ROWS_GATHER && L > 0 && M % L == 0 && keep_all != nullptr) {
    NrProfScope ps(stream, "rows_materialize_needed[%s,Mmax=%d,K=%d]", dtype == NR_BF16 ? "bf16" : "f32", M, K);
    const int n = M / L;
    if (dtype == NR_BF16)
      hipLaunchKernelGGL(gather_title_kernel<bf16_t>, dim3(n), dim3(256), 0, stream, A, (bf16_t*)out, ldo, M, K, L, needed, n, margin, keep_all);
    else
      hipLaunchKernelGGL(gather_title_kernel<float>, dim3(n), dim3(256), 0, stream, A, (float*)out, ldo, M, K, L, needed, n, margin, keep_all);
    NR_CHECK_LAUNCH();
    return NR_OK;
  }
  if (needed != nullptr && A.kind == ROWS_IM2COL3 && K == 3 * A.ld && M % A.Tlen == 0) {
    NrProfScope ps(stream, "rows_materialize_needed[%s,Mmax=%d,K=%d]", dtype == NR_BF16 ? "bf16" : "f32", M, K);
    const int n = M / A.Tlen;
    if (dtype == NR_BF16)
      hipLaunchKernelGGL(im2col3_title_kernel<bf16_t>, dim3(n), dim3(256), 0, stream, A, (bf16_t*)out, ldo, M, needed, n, margin);
    else
      hipLaunchKernelGGL(im2col3_title_kernel<float>, dim3(n), dim3(256), 0, stream, A, (float*)out, ldo, M, needed, n, margin);
    NR_CHECK_LAUNCH();
    return NR_OK;
  }
  NrProfScope ps(stream, "rows_materialize[%s,M=%d,K=%d]", dtype == NR_BF16 ? "bf16" : "f32", M, K);
  const size_t total = (size_t)M * (K / ch);
  size_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  if (A.kind == ROWS_IM2COL3) {
    NR_CHECK_ARG(K == 3 * A.ld && A.ld % ch == 0, "rows_materialize: im2col rows are [M, 3*Dp]");
    size_t g3 = ((size_t)M * (A.ld / ch) + 255) / 256;
    if (g3 > 256 * 32) g3 = 256 * 32;
    if (dtype == NR_BF16)
      hipLaunchKernelGGL(im2col3_materialize_kernel<bf16_t>, dim3((unsigned)g3), dim3(256), 0, stream, A, (bf16_t*)out, ldo, M);
    else
      hipLaunchKernelGGL(im2col3_materialize_kernel<float>, dim3((unsigned)g3), dim3(256), 0, stream, A, (float*)out, ldo, M);
  } else if (dtype == NR_BF16)
    hipLaunchKernelGGL((rows_materialize_kernel<bf16_t, ROWS_GATHER>), dim3((unsigned)grid), dim3(256), 0, stream, A, (bf16_t*)out, ldo, M, K);
  else
    hipLaunchKernelGGL((rows_materialize_kernel<float, ROWS_GATHER>), dim3((unsigned)grid), dim3(256), 0, stream, A, (float*)out, ldo, M, K);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_pool_fused_fwd_ok(int dtype, int n, int L, int N, int q, int ldw1) {
  return !nr_opt(NR_OPT_NO_POOL_FUSED) && dtype == NR_BF16 && L >= 24 && L <= 32 && N > 384 && N <= 416 && N % 8 == 0 && q >= 8 && q <= 256 &&
         q % 8 == 0 && ldw1 >= 416 && (long)n * L >= 4096;
}

int nr_launch_pool_fused_fwd(const void* x, int ldx, const void* w1, int ldw1, const float* b1, const float* w2, const float* b2,
                             const float* mask, void* e, int lde, float* alpha, float* out, int ld_out, int n, int L, int N, int q,
                             const int32_t* needed, hipStream_t stream) {
  constexpr int KS = 13;
  using Cfg = PoolFusedCfg<KS>;
  NR_CHECK_ARG(nr_pool_fused_fwd_ok(NR_BF16, n, L, N, q, ldw1) && ldx >= N && ldx % 8 == 0 && lde >= q && lde % 8 == 0 &&
                   ((((uintptr_t)x) | ((uintptr_t)w1) | ((uintptr_t)e)) & 15) == 0 && mask == nullptr,
               "pool_fused_fwd: shape / alignment not eligible (and no mask on this path)");
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, c = 0;
    NR_CHECK_HIP(hipGetDevice(&dev));
    NR_CHECK_HIP(hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev));
    cus = c >= 8 ? c : 256;
  }
  const int grid = n < cus ? n : cus;                    // one persistent workgroup per CU
  const int max_steps = (n + grid - 1) / grid;
  const size_t smem = (size_t)Cfg::LIST + (size_t)(max_steps + 8) * sizeof(int);
  NR_CHECK_ARG(smem <= 160 * 1024, "pool_fused_fwd: %d sequences per workgroup do not fit the LDS list", max_steps);
  PoolFusedArgs a;
  a.x = (const bf16_t*)x; a.ldx = ldx; a.w1 = (const bf16_t*)w1; a.ldw1 = ldw1; a.b1 = b1; a.w2 = w2; a.b2 = b2;
  a.e = (bf16_t*)e; a.lde = lde; a.alpha = alpha; a.out = out; a.ld_out = ld_out; a.needed = needed; a.n = n; a.L = L; a.N = N; a.q = q;
  a.ablate = nr_opt(NR_OPT_POOL_ABLATE);
  auto kern = pool_fused_fwd_kernel<KS>;
  NR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  NrProfScope ps(stream, needed ? "pool_fused_fwd_needed[bf16,n=%d,L=%d,N=%d,q=%d]" : "pool_fused_fwd[bf16,n=%d,L=%d,N=%d,q=%d]", n, L, N, q);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, stream, a, max_steps);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_pool_fused_bwd_ok(int dtype, int n, int L, int N, int q, int ldw1t) {
  return !nr_opt(NR_OPT_NO_POOL_FUSED) && dtype == NR_BF16 && L >= 24 && L <= 32 && N > 384 && N <= 416 && N % 8 == 0 && q > 192 && q <= 224 &&
         q % 8 == 0 && ldw1t >= 224 && (long)n * L >= 4096;
}

// partial: [>= grid rows][q + 1]; *grid_out = rows written (the caller sums them)
int nr_launch_pool_fused_bwd(const void* x, int ldx, const void* e, int lde, const float* alpha, const float* g, int ldg, const float* w2,
                             const void* w1t, int ldw1t, void* dpre, int ldp, void* dx, int lddx, float* partial, int partial_rows,
                             const int32_t* nz, int n, int L, int N, int q, hipStream_t stream, int* grid_out, int dx_far_unwritten) {
  constexpr int KS = 13, KS2 = 7;
  using Cfg = PoolFusedBwdCfg<KS, KS2>;
  NR_CHECK_ARG(nr_pool_fused_bwd_ok(NR_BF16, n, L, N, q, ldw1t) && ldx >= N && ldx % 8 == 0 && lde >= q && lde % 8 == 0 && ldp >= q && ldp % 8 == 0 &&
                   lddx >= N && lddx % 8 == 0 && ldg >= N && ldg % 4 == 0 &&
                   ((((uintptr_t)x) | ((uintptr_t)e) | ((uintptr_t)w1t) | ((uintptr_t)dpre) | ((uintptr_t)dx) | ((uintptr_t)g)) & 15) == 0,
               "pool_fused_bwd: shape / alignment not eligible");
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, c = 0;
    NR_CHECK_HIP(hipGetDevice(&dev));
    NR_CHECK_HIP(hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev));
    cus = c >= 8 ? c : 256;
  }
  int grid = n < cus ? n : cus;                          // one persistent workgroup per CU
  if (grid > partial_rows) grid = partial_rows;
  const int max_steps = (n + grid - 1) / grid;
  const size_t smem = (size_t)Cfg::LIST + (size_t)(max_steps + 8) * sizeof(int);
  NR_CHECK_ARG(smem <= 160 * 1024, "pool_fused_bwd: %d sequences per workgroup do not fit the LDS list", max_steps);
  PoolFusedBwdArgs a;
  a.x = (const bf16_t*)x; a.ldx = ldx; a.e = (const bf16_t*)e; a.lde = lde; a.alpha = alpha; a.g = g; a.ldg = ldg; a.w2 = w2;
  a.w1t = (const bf16_t*)w1t; a.ldw1t = ldw1t; a.dpre = (bf16_t*)dpre; a.ldp = ldp; a.dx = (bf16_t*)dx; a.lddx = lddx; a.partial = partial;
  a.nz = nz; a.n = n; a.L = L; a.N = N; a.q = q; a.dx_far_unwritten = (nz != nullptr && dx_far_unwritten) ? 1 : 0;
  auto kern = pool_fused_bwd_kernel<KS, KS2>;
  NR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  NrProfScope ps(stream, "pool_fused_bwd[bf16,n=%d,L=%d,N=%d,q=%d]", n, L, N, q);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, stream, a, 32 / L + 2);
  NR_CHECK_LAUNCH();
  *grid_out = grid;
  return NR_OK;
}

int nr_launch_gather_live_rows(int dtype, const RowSrc& A, void* out, int ldo, int Mmax, int K, const int32_t* count, const int32_t* rows,
                               const int32_t* ids, hipStream_t stream, int32_t* hist, int V) {
  const int ch = nr_chunk(dtype);
  NR_CHECK_ARG(A.kind == ROWS_GATHER && K % ch == 0 && ldo % ch == 0 && ldo >= K && A.ld >= K && count && rows && ids && out,
               "gather_live_rows: bad arguments");
  NrProfScope ps(stream, "rows_materialize_live[%s,Mmax=%d,K=%d]", dtype == NR_BF16 ? "bf16" : "f32", Mmax, K);
  const size_t total = (size_t)Mmax * (K / ch);
  size_t grid = (total + 256 * 4 - 1) / (256 * 4);
  if (grid > 256 * 16) grid = 256 * 16;              // 16 workgroups of 4 waves per CU: 64 row chains in flight per CU
  if (grid < 1) grid = 1;
  if (dtype == NR_BF16)
    hipLaunchKernelGGL(gather_live_rows_kernel<bf16_t>, dim3((unsigned)grid), dim3(256), 0, stream, A, (bf16_t*)out, ldo, K, count, rows, ids, hist, V);
  else
    hipLaunchKernelGGL(gather_live_rows_kernel<float>, dim3((unsigned)grid), dim3(256), 0, stream, A, (float*)out, ldo, K, count, rows, ids, hist, V);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_launch_gemm_nt(int dtype, const RowSrc& A, const void* B, int ldb, int M, int N, int K, int epi, const EpiArgs& ep,
                      hipStream_t stream) {
  const int ch = nr_chunk(dtype);
  NR_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
  NR_CHECK_ARG(K % ch == 0 && ldb % ch == 0 && A.ld % ch == 0, "gemm_nt: K=%d ldb=%d lda=%d must be multiples of %d", K, ldb, A.ld, ch);
  NR_CHECK_ARG(((uintptr_t)A.base & 15) == 0 && ((uintptr_t)B & 15) == 0, "gemm_nt: operands must be 16-byte aligned");
  if (epi != EPI_SCATTER) NR_CHECK_ARG(ep.ldc % 4 == 0 && ((uintptr_t)ep.C & 15) == 0, "gemm_nt: output ld %d / alignment", ep.ldc);
  if (epi == EPI_STORE && ep.act_tanh) epi = EPI_STORE_TANH;
  // Kernel choice (bf16, dense A), from per-shape measurements on MI355X (tools/gemm_probe.py):
  //   K >= 256            : LDS-DMA ring kernel, 256-row tiles x column chunks of 208 / 320 (QKV projection K=304:
  //                         1.42 ms vs 1.92 ms tiled; att_fc1 K=400; dX K=1200), packed bf16 epilogue
  //   N <= 208, small K   : "wide" kernel, all N columns per workgroup (A read once)
  //   otherwise           : 128 x 128 tiled kernel with the packed bf16 epilogue (pooling dX, K=200: 0.77 vs 1.02 ms)
  const bool no_wide = nr_opt(NR_OPT_NT_NOWIDE) != 0;
  const bool no_dma = nr_opt(NR_OPT_NT_NODMA) != 0;
  const int kr32 = (K + 31) / 32 * 32;
  const bool dense_bf16 = dtype == NR_BF16 && A.kind == ROWS_DENSE && A.drop.thresh == 0 && ep.rows_out == nullptr;
  const int dma_min_k = nr_opt(NR_OPT_DMA_MIN_K);
  if (dense_bf16 && !no_dma && K >= dma_min_k && ldb >= kr32 && (A.ld >= K || A.gap > 0)) {
    EpiArgs epg = ep;
    epg.a_gap = A.gap;
    const EpiArgs& ep = epg;                       // (shadows the parameter: the kernels below see the operand's row gap)
    // B must be zero beyond K up to the next multiple of 32 (nr_cast_pad with such an ld guarantees it)
    // with row compaction only the live rows (count on the device) are multiplied: M is then an upper bound
    // "_live": rows compacted on the device; "_needed": row tiles of unneeded sequences are skipped (M is an upper bound in both)
    const bool tile_skip = (epi == EPI_STORE || epi == EPI_STORE_TANH) && ep.seq_nz != nullptr && ep.row_count == nullptr;
    // skinny K, bf16 out: weights-in-registers kernels (see gemm_nt_wreg_kernel); shapes with an instantiation:
    //   QKV projection    STORE       K in (288, 320], N >= 320        5 column tiles per wave, 16-row steps
    //   user-level QKV    STORE       K in (384, 416], N >= 320        3 column tiles per wave, 32-row stages (two passes)
    //   pooling fc1       STORE_TANH  K in (384, 416], N <= 256        2 column tiles per wave, 32-row steps
    //   pooling dX        POOLBWD     K in (192, 224], N <= 512        4 column tiles per wave, 32-row steps
    // ("_needed" there: only 32-row blocks that touch a flagged sequence are computed)
    if (nr_opt(NR_OPT_NT_WREG) && A.gap == 0 && ep.out_dtype == NR_BF16 && N % 8 == 0 && ep.ldc % 8 == 0 && M >= 64 && ldb >= kr32 &&
        (ep.row_count == nullptr || (((uintptr_t)ep.row_idx & 15) == 0 && M % 4 == 0))) {
      const bool flags = ep.seq_nz != nullptr && ep.row_count == nullptr && ep.L >= 16;
      const char* lbl = ep.row_count ? "gemm_nt_wreg_live[bf16,epi=%d,Mmax=%d,N=%d,K=%d]"
                                     : (ep.seq_nz ? "gemm_nt_wreg_needed[bf16,epi=%d,Mmax=%d,N=%d,K=%d]" : "gemm_nt_wreg[bf16,epi=%d,M=%d,N=%d,K=%d]");
      if (epi == EPI_STORE && !tile_skip && K > 288 && K <= 320 && N >= 320) {
        NrProfScope ps(stream, lbl, epi, M, N, K);
        if (ep.row_count && ep.a_dense) return launch_nt_wreg_m<EPI_STORE, 5, 10, 2, WREG_COMPACT_DENSE, true>(A, B, ldb, M, N, K, ep, stream);
        if (nr_opt(NR_OPT_NT_WREG) == 3)           // 16-row stages (one row tile per barrier)
          return ep.row_count ? launch_nt_wreg_m<EPI_STORE, 5, 10, 1, WREG_COMPACT>(A, B, ldb, M, N, K, ep, stream)
                              : launch_nt_wreg_m<EPI_STORE, 5, 10, 1, WREG_DENSE>(A, B, ldb, M, N, K, ep, stream);
        return ep.row_count ? launch_nt_wreg_m<EPI_STORE, 5, 10, 2, WREG_COMPACT, true>(A, B, ldb, M, N, K, ep, stream)
                            : launch_nt_wreg_m<EPI_STORE, 5, 10, 2, WREG_DENSE, true>(A, B, ldb, M, N, K, ep, stream);
      }
      if (epi == EPI_STORE && !tile_skip && ep.row_count == nullptr && K > 384 && K <= 416 && N >= 320) {   // user-level QKV (d_model = 400)
        NrProfScope ps(stream, lbl, epi, M, N, K);
        return launch_nt_wreg_m<EPI_STORE, 3, 13, 2, WREG_DENSE, true>(A, B, ldb, M, N, K, ep, stream);
      }
      if (epi == EPI_STORE_TANH && ep.row_count == nullptr && K > 384 && K <= 416 && N <= 256 && (!tile_skip || flags)) {
        NrProfScope ps(stream, lbl, epi, M, N, K);
        return tile_skip ? launch_nt_wreg_m<EPI_STORE_TANH, 2, 13, 2, WREG_LIST>(A, B, ldb, M, N, K, ep, stream)
                         : launch_nt_wreg_m<EPI_STORE_TANH, 2, 13, 2, WREG_DENSE>(A, B, ldb, M, N, K, ep, stream);
      }
      if (epi == EPI_POOLBWD && ep.row_count == nullptr && K > 192 && K <= 224 && N <= 512 && ep.L >= 16 && ep.ldg % 4 == 0 &&
          (((uintptr_t)ep.G | (uintptr_t)ep.rowscale) & 15) == 0 && M % 4 == 0 && (ep.seq_nz == nullptr || flags)) {
        NrProfScope ps(stream, lbl, epi, M, N, K);
        return ep.seq_nz ? launch_nt_wreg_m<EPI_POOLBWD, 4, 7, 2, WREG_LIST>(A, B, ldb, M, N, K, ep, stream)
                         : launch_nt_wreg_m<EPI_POOLBWD, 4, 7, 2, WREG_DENSE>(A, B, ldb, M, N, K, ep, stream);
      }
    }
    // (gap=T: the operand's rows overlap -- the conv's im2col-free operand, K/3 fresh columns per row: bench.py prices its bytes so)
    NrProfScope ps(stream, ep.row_count ? "gemm_nt_dma_live[bf16,epi=%d,Mmax=%d,N=%d,K=%d,gap=%d]"
                                        : (tile_skip ? "gemm_nt_dma_needed[bf16,epi=%d,Mmax=%d,N=%d,K=%d,gap=%d]" : "gemm_nt_dma[bf16,epi=%d,M=%d,N=%d,K=%d,gap=%d]"),
                   epi, M, N, K, A.gap);
    const int c13 = ((N + 207) / 208) * 13, c20 = ((N + 319) / 320) * 20;   // fewer padded column tiles wins
    if (c13 < c20) return launch_nt_dma_e<13>(A, B, ldb, M, N, K, epi, ep, stream);
    return launch_nt_dma_e<20>(A, B, ldb, M, N, K, epi, ep, stream);
  }
  if (dense_bf16 && !no_wide && N <= 208 && A.gap == 0) {
    NrProfScope ps(stream, "gemm_nt_wide[bf16,epi=%d,M=%d,N=%d,K=%d]", epi, M, N, K);
    return launch_nt_wide_e<13>(A, B, ldb, M, N, K, epi, ep, stream);
  }
  NrProfScope ps(stream, "gemm_nt[%s,rows=%d,epi=%d,M=%d,N=%d,K=%d]", dtype == NR_BF16 ? "bf16" : "f32", A.kind, epi, M, N, K);
  return dtype == NR_BF16 ? launch_nt_d<bf16_t>(A, B, ldb, M, N, K, epi, ep, stream)
                          : launch_nt_d<float>(A, B, ldb, M, N, K, epi, ep, stream);
}

namespace {
// 32-row slabs of a [n*L, *] tensor that touch at least one sequence flagged in title_nz, in order; ws[0] = their number
// ORDERED compaction by one workgroup (nslab is ~26 000 at the bench shape): every thread owns a contiguous run of slabs,
// counts its live ones, the block scans the counts, and the runs are written back to back -- the list is ascending, so the
// order in which a split of the weight-gradient GEMM contracts its slabs (and with it every fp32 rounding) is the same
// from run to run.
__global__ __launch_bounds__(1024) void live_slabs_kernel(const int32_t* __restrict__ title_nz, int n, int L, int nslab,
                                                          int32_t* __restrict__ count, int32_t* __restrict__ list, int in_lds) {
  __shared__ int sCnt[1024];
  extern __shared__ __attribute__((aligned(16))) unsigned char sFlag[];   // in_lds: the n sequence flags as bytes
  const int tid = threadIdx.x;
  if (in_lds) {
    // one coalesced sweep with independent loads (8 in flight per thread); the per-slab lookups below (2-3 flags per slab,
    // a dependent chain of ~26 slabs per thread) then hit LDS instead of paying a global-memory latency each
    for (int base = 0; base < n; base += 8 * 1024) {
      int v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = base + u * 1024 + tid;
        v[u] = t < n ? title_nz[t] : 0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = base + u * 1024 + tid;
        if (t < n) sFlag[t] = v[u] != 0 ? 1 : 0;
      }
    }
    __syncthreads();
  }
  const int per = (nslab + 1023) / 1024, s0 = tid * per, s1 = min(nslab, s0 + per);
  // sequence of row 32 s and its offset inside it are carried along the thread's run of slabs (one integer division per
  // thread: two per slab, ~40 instructions each, were most of this kernel's 38 us)
  const int tq = s0 < nslab ? (32 * s0) / L : 0, rq = 32 * s0 - tq * L;
  auto flag = [&](int t) { return in_lds ? (int)sFlag[t] : (title_nz[t] != 0 ? 1 : 0); };
  auto walk = [&](auto&& emit) {
    int t = tq, rem = rq;
    for (int s = s0; s < s1; ++s) {
      int lv = flag(min(t, n - 1)), e = rem + 31, te = t;
      while (e >= L) {                              // further sequences the slab's 32 rows reach into
        e -= L; ++te;
        if (te < n) lv |= flag(te);
      }
      emit(s, lv != 0);
      rem += 32;
      while (rem >= L) { rem -= L; ++t; }
    }
  };
  int c = 0;
  walk([&](int, bool lv) { c += lv ? 1 : 0; });
  // block scan: inside each wave by shuffles, then the 16 wave totals by the first wave (2 barriers instead of the 20 of a
  // Hillis-Steele scan over LDS)
  const int lane = tid & 63, wv = tid >> 6;
  int v = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  if (lane == 63) sCnt[wv] = v;
  __syncthreads();
  if (wv == 0) {
    int w = lane < 16 ? sCnt[lane] : 0;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int t = __shfl_up(w, o, 64);
      if (lane >= o) w += t;
    }
    if (lane < 16) sCnt[16 + lane] = w;                   // inclusive totals of waves 0 .. lane
  }
  __syncthreads();
  int pos = v - c + (wv > 0 ? sCnt[16 + wv - 1] : 0);
  walk([&](int s, bool lv) { if (lv) list[pos++] = s; });
  if (tid == 1023) sCnt[1023] = sCnt[16 + 15];            // (the line below reports the grand total)
  if (tid == 1023) *count = sCnt[1023];
}
}  // namespace

// The same in two launches, which is what runs for up to 128 K slabs: the per-slab test (two integer divisions, 2-3 flag
// loads) over the whole chip, then ONE workgroup that only counts bytes, scans and writes the ordered list.  The single
// workgroup doing the per-slab work for all 26 400 slabs was instruction bound on its one CU: 30 us a call, twice a step.
namespace {
__global__ __launch_bounds__(256) void slab_flags_kernel(const int32_t* __restrict__ title_nz, int n, int L, int nslab,
                                                         unsigned char* __restrict__ flags) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= nslab) return;
  const int t0 = (32 * s) / L, t1 = min(n - 1, (32 * s + 31) / L);
  bool lv = false;
  for (int t = t0; t <= t1; ++t) lv |= title_nz[t] != 0;
  flags[s] = lv ? 1 : 0;
}
// flags: nslab bytes that may overlap the END of `list` (they are staged into LDS before the first list entry is written)
__global__ __launch_bounds__(1024) void slab_list_kernel(const unsigned char* __restrict__ flags, int nslab, int32_t* __restrict__ count,
                                                         int32_t* __restrict__ list) {
  __shared__ int sCnt[32];
  extern __shared__ __attribute__((aligned(16))) unsigned char sFlag[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int words = (nslab + 3) / 4;
  const uint32_t* fw = reinterpret_cast<const uint32_t*>(flags);
  for (int base = 0; base < words; base += 8 * 1024) {
    uint32_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int w = base + u * 1024 + tid;
      v[u] = w < words ? fw[w] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int w = base + u * 1024 + tid;
      if (w < words) reinterpret_cast<uint32_t*>(sFlag)[w] = v[u];
    }
  }
  __syncthreads();
  const int per = (nslab + 1023) / 1024, s0 = tid * per, s1 = min(nslab, s0 + per);
  int c = 0;
  for (int s2 = s0; s2 < s1; ++s2) c += sFlag[s2];
  int v = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  if (lane == 63) sCnt[wv] = v;
  __syncthreads();
  if (wv == 0) {
    int w = lane < 16 ? sCnt[lane] : 0;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int t = __shfl_up(w, o, 64);
      if (lane >= o) w += t;
    }
    if (lane < 16) sCnt[16 + lane] = w;
  }
  __syncthreads();
  int pos = v - c + (wv > 0 ? sCnt[16 + wv - 1] : 0);
  for (int s2 = s0; s2 < s1; ++s2)
    if (sFlag[s2]) list[pos++] = s2;
  if (tid == 0) *count = sCnt[16 + 15];
}
}  // namespace

// title_nz[s] = 1 when any element of the L rows of sequence s in dy [n*L, N] (bf16) is non-zero (-0 counts as zero)
namespace {
__global__ __launch_bounds__(256) void title_flags_kernel(const bf16_t* __restrict__ dy, int L, int N, int32_t* __restrict__ title_nz) {
  const uint4* p = reinterpret_cast<const uint4*>(dy + (size_t)blockIdx.x * L * N);
  const int chunks = L * N / 8, first = min(chunks, 256);
  __shared__ int flag;
  if (threadIdx.x == 0) flag = 0;
  __syncthreads();
  // a sequence with a gradient shows it in its first 4 KB almost always: look there first, scan the rest only if that is zero
  if ((int)threadIdx.x < first) {
    const uint4 v = p[threadIdx.x];
    if (((v.x | v.y | v.z | v.w) & 0x7fff7fffu) != 0u) flag = 1;     // benign race: every writer stores the same value
  }
  __syncthreads();
  if (flag == 0) {                                                   // uniform: read after the barrier
    uint32_t any = 0;
    for (int c = first + threadIdx.x; c < chunks; c += 256) {
      const uint4 v = p[c];
      any |= (v.x | v.y | v.z | v.w) & 0x7fff7fffu;
    }
    __syncthreads();                                                 // everyone has read flag == 0 before anyone sets it
    if (any != 0u) flag = 1;
    __syncthreads();
  }
  if (threadIdx.x == 0) title_nz[blockIdx.x] = flag;
}
}  // namespace
// the same for fp32 rows g [n, ld] (one row per sequence)
namespace {
__global__ __launch_bounds__(256) void row_flags_f32_kernel(const float* __restrict__ g, int ld, int N, int n, int32_t* __restrict__ nz) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  const uint32_t* p = reinterpret_cast<const uint32_t*>(g + (size_t)row * ld);
  uint32_t any = 0;
  for (int c = lane; c < N; c += 64) any |= p[c] & 0x7fffffffu;
  const uint64_t b = __ballot(any != 0u);
  if (lane == 0) nz[row] = b != 0ull ? 1 : 0;
}
}  // namespace
int nr_launch_row_flags_f32(const float* g, int ld, int N, int n, int32_t* nz, hipStream_t stream) {
  NR_CHECK_ARG(g != nullptr && nz != nullptr && n > 0, "row_flags: bad arguments");
  hipLaunchKernelGGL(row_flags_f32_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, g, ld, N, n, nz);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_launch_title_flags(const void* dy, int n, int L, int N, int32_t* title_nz, hipStream_t stream) {
  NR_CHECK_ARG(dy != nullptr && title_nz != nullptr && (L * N) % 8 == 0 && (((uintptr_t)dy) & 15) == 0 && N % 8 == 0,
               "title_flags: bad arguments");
  NrProfScope ps(stream, "title_flags[n=%d,L=%d,N=%d]", n, L, N);
  hipLaunchKernelGGL(title_flags_kernel, dim3(n), dim3(256), 0, stream, (const bf16_t*)dy, L, N, title_nz);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

// Sequences the attention backward has to process, in order: everything except the all-padding sequences (tmask == 0)
// whose own and `margin` neighbours' upstream gradients are zero -- no live 32-row slab overlaps their rows and the
// table-gradient GEMM never gathers them, so their (exactly zero) dQ|dK|dV rows need not even be written.
namespace {
__global__ __launch_bounds__(256) void seq_list_kernel(const int32_t* __restrict__ title_nz, const uint32_t* __restrict__ tmask, int n,
                                                       int margin, int32_t* __restrict__ count, int32_t* __restrict__ list) {
  __shared__ int wave_cnt[4];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i = blockIdx.x * 256 + tid;
  bool keep = false;
  if (i < n) {
    bool skip = tmask[i] == 0;
    for (int j = max(0, i - margin); skip && j <= min(n - 1, i + margin); ++j) skip = title_nz[j] == 0;
    keep = !skip;
  }
  const uint64_t bal = __ballot(keep);
  if (lane == 0) wave_cnt[wid] = __popcll(bal);
  __syncthreads();
  if (tid == 0) base = atomicAdd(count, wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3]);
  __syncthreads();
  if (keep) {
    int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wid; ++w) pos += wave_cnt[w];
    list[pos] = i;
  }
}
}  // namespace
// Forward: the sequences whose output somebody needs (flags[i] != 0), and exact zeros in the y rows of the others.
namespace {
__global__ __launch_bounds__(256) void needed_list_kernel(const int32_t* __restrict__ flags, int n, int32_t* __restrict__ count,
                                                          int32_t* __restrict__ list) {
  __shared__ int wave_cnt[4];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i = blockIdx.x * 256 + tid;
  const bool keep = i < n && flags[i] != 0;
  const uint64_t bal = __ballot(keep);
  if (lane == 0) wave_cnt[wid] = __popcll(bal);
  __syncthreads();
  if (tid == 0) base = atomicAdd(count, wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3]);
  __syncthreads();
  if (keep) {
    int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wid; ++w) pos += wave_cnt[w];
    list[pos] = i;
  }
}
// one workgroup per unneeded sequence: its rows (chunks16 16-byte pieces) become zeros
// reach >= 0: only when a needed sequence lies within `reach` sequences (the others stay unwritten: nobody reads them)
__global__ __launch_bounds__(256) void zero_unneeded_kernel(const int32_t* __restrict__ flags, uint4* __restrict__ y, int chunks16, int n,
                                                            int reach) {
  const int seq = blockIdx.x;
  if (flags[seq] != 0) return;
  if (reach >= 0) {
    bool near = false;
    for (int t = max(0, seq - reach) + (int)threadIdx.x; t <= min(n - 1, seq + reach); t += 256) near |= flags[t] != 0;
    if (!__syncthreads_or(near)) return;
  }
  uint4* p = y + (size_t)seq * chunks16;
  for (int c = threadIdx.x; c < chunks16; c += 256) p[c] = make_uint4(0, 0, 0, 0);
}
}  // namespace
// out: int32 [4 + n]: out[0] = count, out[4 ..] = the needed sequences; y rows of the others are zero-filled (row_bytes % 16 == 0)
int nr_launch_needed_list(const int32_t* flags, int n, int32_t* out, void* y, size_t seq_bytes, hipStream_t stream, int reach, bool zeroed) {
  NR_CHECK_ARG(flags != nullptr && out != nullptr && y != nullptr && seq_bytes % 16 == 0 && (((uintptr_t)y) & 15) == 0, "needed_list: bad arguments");
  if (!zeroed) NR_CHECK_HIP(hipMemsetAsync(out, 0, 4 * sizeof(int32_t), stream));     // (zeroed: an earlier kernel of the call cleared out[0..4))
  NrProfScope ps(stream, "needed_list[n=%d]", n);
  hipLaunchKernelGGL(needed_list_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, flags, n, out, out + 4);
  hipLaunchKernelGGL(zero_unneeded_kernel, dim3(n), dim3(256), 0, stream, flags, reinterpret_cast<uint4*>(y), (int)(seq_bytes / 16), n, reach);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

// out: int32 [4 + n]: out[0] = count, out[4 ..] = sequence numbers
int nr_launch_seq_list(const int32_t* title_nz, const uint32_t* tmask, int n, int L, int32_t* out, hipStream_t stream, int reach, bool zeroed) {
  if (!zeroed) NR_CHECK_HIP(hipMemsetAsync(out, 0, 4 * sizeof(int32_t), stream));
  hipLaunchKernelGGL(seq_list_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, title_nz, tmask, n, reach >= 0 ? reach : 32 / L + 2, out,
                     out + 4);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

// rows *count .. roundup32(*count) of a [Mmax, ld] bf16 buffer become zeros (compact row storage: the weight-gradient GEMM
// contracts whole 32-row slabs of the live rows)
namespace {
// za / zb (optional): int32 regions cleared on the way -- counters and histograms of kernels that follow in the same call
// (each used to be a hipMemsetAsync of its own, ~5 us apiece in a chain of dependent launches)
__global__ __launch_bounds__(256) void zero_tail_rows_kernel(uint4* __restrict__ buf, int chunks, const int32_t* __restrict__ count, int Mmax,
                                                             int32_t* __restrict__ za, int na, int32_t* __restrict__ zb, int nb) {
  const int t = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
  for (int i = t; i < na; i += stride) za[i] = 0;
  for (int i = t; i < nb; i += stride) zb[i] = 0;
  if (blockIdx.x != 0) return;
  const int n = *count, end = min(Mmax, (n + 31) / 32 * 32);
  for (int u = threadIdx.x; u < (end - n) * chunks; u += 256) buf[(size_t)n * chunks + u] = make_uint4(0, 0, 0, 0);
}
}  // namespace
int nr_launch_zero_tail_rows(void* buf, int ld, const int32_t* count, int Mmax, hipStream_t stream, int32_t* za, int na, int32_t* zb, int nb) {
  NR_CHECK_ARG(buf != nullptr && count != nullptr && ld % 8 == 0 && (((uintptr_t)buf) & 15) == 0, "zero_tail_rows: bad arguments");
  if (za == nullptr) na = 0;
  if (zb == nullptr) nb = 0;
  const int most = na > nb ? na : nb, grid = most > 4096 ? 16 : 1;
  hipLaunchKernelGGL(zero_tail_rows_kernel, dim3(grid), dim3(256), 0, stream, (uint4*)buf, ld / 8, count, Mmax, za, na, zb, nb);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

// ws: int32 [n + 4 + M/32]: ws[0..n) = title_nz (nr_launch_title_flags), ws[n] = slab count, ws[n+4 ..] slab list.
// The list is ascending (see live_slabs_kernel).
int nr_launch_live_slabs(int32_t* ws, int n, int L, hipStream_t stream) {
  const int M = n * L, nslab = M / 32;
  NR_CHECK_ARG(ws != nullptr && M % 32 == 0, "live_slabs: bad arguments");
  int32_t* list = ws + n + 4;
  if (nslab >= 4 && nslab <= 128 * 1024) {
    // the flag bytes live in the last nslab bytes of the list array (int aligned) until the list kernel has staged them
    unsigned char* flags = reinterpret_cast<unsigned char*>(list + nslab - (nslab + 3) / 4);
    hipLaunchKernelGGL(slab_flags_kernel, dim3((nslab + 255) / 256), dim3(256), 0, stream, ws, n, L, nslab, flags);
    hipLaunchKernelGGL(slab_list_kernel, dim3(1), dim3(1024), (size_t)((nslab + 15) / 16) * 16, stream, flags, nslab, ws + n, list);
    NR_CHECK_LAUNCH();
    return NR_OK;
  }
  const int in_lds = n <= 128 * 1024 ? 1 : 0;
  hipLaunchKernelGGL(live_slabs_kernel, dim3(1), dim3(1024), in_lds ? (size_t)((n + 15) / 16) * 16 : 0, stream, ws, n, L, nslab, ws + n,
                     list, in_lds);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_launch_gemm_tn_slabs(const void* dC, int ldc, const void* X, int ldx, float* dW, int ldw, float* db, int M, int N, int K,
                            int Nstore, int Kstore, const int32_t* slab_list, const int32_t* slab_count, hipStream_t stream, int xgap,
                            float* scratch, size_t scratch_floats) {
  NR_CHECK_ARG(tn3::eligible(ldc, ldx, M, N, K), "gemm_tn_slabs: shape not eligible");
  NrProfScope ps(stream, "gemm_tn3_live[bf16,Mmax=%d,N=%d,K=%d,gap=%d]", M, N, K, xgap);
  return tn3::launch(dC, ldc, X, ldx, dW, ldw, db, M, N, K, Nstore, Kstore, stream, slab_list, slab_count, xgap, scratch, scratch_floats);
}
size_t nr_gemm_tn_scratch_floats(int M, int N, int K) { return tn3::scratch_floats(M, N, K); }
bool nr_gemm_tn_slabs_ok(int ldc, int ldx, int M, int N, int K) { return tn3::eligible(ldc, ldx, M, N, K); }
// dW += dC^T . X over the first *row_count rows of two dense operands that hold at most Mmax rows (compact row storage: rows
// row_count .. roundup32(row_count) must be zero in X and finite in dC).  No bias gradient: its rows are not all here.
int nr_launch_gemm_tn_counted(const void* dC, int ldc, const void* X, int ldx, float* dW, int ldw, int Mmax, int N, int K, int Nstore,
                              int Kstore, const int32_t* row_count, hipStream_t stream, float* scratch, size_t scratch_floats) {
  NR_CHECK_ARG(tn3::eligible(ldc, ldx, Mmax, N, K) && row_count != nullptr, "gemm_tn_counted: shape not eligible");
  NrProfScope ps(stream, "gemm_tn3_rows[bf16,Mmax=%d,N=%d,K=%d,gap=0]", Mmax, N, K);
  return tn3::launch(dC, ldc, X, ldx, dW, ldw, nullptr, Mmax, N, K, Nstore, Kstore, stream, nullptr, row_count, 0, scratch, scratch_floats);
}

int nr_launch_gemm_tn(int dtype, const void* dC, int ldc, const RowSrc& A, float* dW, int ldw, float* db, int M, int N,
                      int K, int Nstore, int Kstore, hipStream_t stream) {
  const int ch = nr_chunk(dtype);
  NR_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm_tn: empty problem M=%d N=%d K=%d", M, N, K);
  NR_CHECK_ARG(N % ch == 0 && K % ch == 0 && ldc % ch == 0 && A.ld % ch == 0, "gemm_tn: N=%d K=%d ldc=%d lda=%d must be multiples of %d",
               N, K, ldc, A.ld, ch);
  NR_CHECK_ARG(((uintptr_t)A.base & 15) == 0 && ((uintptr_t)dC & 15) == 0, "gemm_tn: operands must be 16-byte aligned");
  const bool tn_v1 = nr_opt(NR_OPT_TN_V1) != 0;
  if (!tn_v1 && dtype == NR_BF16 && A.kind == ROWS_DENSE && A.drop.thresh == 0) {
    if (tn3::eligible(ldc, A.ld, M, N, K)) {
      NrProfScope ps(stream, "gemm_tn3[bf16,M=%d,N=%d,K=%d,gap=%d]", M, N, K, A.gap);
      return tn3::launch(dC, ldc, A.base, A.ld, dW, ldw, db, M, N, K, Nstore, Kstore, stream, nullptr, nullptr, A.gap);
    }
    if (A.gap == 0) {                              // (gapped rows: the generic kernel below follows RowSrc::gap)
      NrProfScope ps(stream, "gemm_tn2[bf16,M=%d,N=%d,K=%d]", M, N, K);
      return tn2::launch(dC, ldc, A.base, A.ld, dW, ldw, db, M, N, K, Nstore, Kstore, stream);
    }
  }
  NrProfScope ps(stream, "gemm_tn[%s,rows=%d,M=%d,N=%d,K=%d]", dtype == NR_BF16 ? "bf16" : "f32", A.kind, M, N, K);
  return dtype == NR_BF16 ? launch_tn_d<bf16_t>(dC, ldc, A, dW, ldw, db, M, N, K, Nstore, Kstore, stream)
                          : launch_tn_d<float>(dC, ldc, A, dW, ldw, db, M, N, K, Nstore, Kstore, stream);
}
