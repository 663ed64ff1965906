// HBM-bound kernels of libnrhip: additive-attention pooling core (K5), pad-doc blend (K6),
// scorer + cross-entropy (K8), embedding row gather / scatter-add (K1), parameter packing.
#include "nr_common.h"

namespace {

// ------------------------------------------------------------------------------------------
// K5 forward core.  e = tanh(fc1(x)) comes from the GEMM epilogue; here:
//   s_l = <e_l, w2> + b2 ; a_l = exp(s_l) mask_l / (sum + 1e-8) ; out = sum_l a_l x_l
// (src/model/model_utils.py:23-30), stable form with the row max factored out.
// One workgroup per sequence; a wave per token for the dot products, a thread per column for
// the weighted sum (coalesced over the [L, N] tile).
// ------------------------------------------------------------------------------------------
// 16-byte chunk of T as floats
template <typename T> struct ChunkOf { static constexpr int CH = 16 / (int)sizeof(T); };
template <typename T> __device__ __forceinline__ void load_chunk_f(const T* p, float (&f)[ChunkOf<T>::CH]);
template <> __device__ __forceinline__ void load_chunk_f<float>(const float* p, float (&f)[4]) {
  const f32x4 v = *reinterpret_cast<const f32x4*>(p);
  f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
}
template <> __device__ __forceinline__ void load_chunk_f<bf16_t>(const bf16_t* p, float (&f)[8]) {
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
}
template <typename T> __device__ __forceinline__ void store_chunk_f(T* p, const float (&f)[ChunkOf<T>::CH]);
template <> __device__ __forceinline__ void store_chunk_f<float>(float* p, const float (&f)[4]) {
  *reinterpret_cast<f32x4*>(p) = (f32x4){f[0], f[1], f[2], f[3]};
}
template <> __device__ __forceinline__ void store_chunk_f<bf16_t>(bf16_t* p, const float (&f)[8]) {
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (bf16_t)f[e];
  *reinterpret_cast<bf16x8*>(p) = v;
}

// <row (nchunks chunks of T), vec (fp32)> with one 16-byte load per lane, summed over the wave
template <typename T>
__device__ __forceinline__ float wave_row_dot(const T* __restrict__ row, const float* __restrict__ vec, int nchunks, int lane) {
  constexpr int CH = ChunkOf<T>::CH;
  float p = 0.f;
  for (int c = lane; c < nchunks; c += 64) {
    float f[CH];
    load_chunk_f<T>(row + c * CH, f);
#pragma unroll
    for (int e = 0; e < CH; ++e) p = fmaf(f[e], vec[c * CH + e], p);
  }
  return wave_sum(p);
}

constexpr int POOL_RED_FLOATS = 256 * 8;   // LDS reduction scratch: [row groups][columns] <= 256 chunks of <= 8

template <typename T>
__global__ __launch_bounds__(256) void pool_fwd_kernel(const T* __restrict__ x, const T* __restrict__ e,
                                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                                       const float* __restrict__ mask, float* __restrict__ alpha,
                                                       float* __restrict__ out, int ld_out, int L, int N, int q,
                                                       const int32_t* __restrict__ needed) {
  constexpr int CH = ChunkOf<T>::CH;
  __shared__ float sS[64];
  __shared__ float sRed[POOL_RED_FLOATS];
  const int seq = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t row0 = (size_t)seq * L;
  if (needed != nullptr && needed[seq] == 0) {             // nobody uses this sequence's vector: zeros, x / e are not read
    for (int l = tid; l < L; l += 256) alpha[row0 + l] = 0.f;
    for (int c = tid; c < N; c += 256) out[(size_t)seq * ld_out + c] = 0.f;
    return;
  }
  for (int l = wid; l < L; l += 4) {
    const float p = wave_row_dot<T>(e + (row0 + l) * q, w2, q / CH, lane);
    if (lane == 0) sS[l] = p + b2[0];
  }
  __syncthreads();
  if (wid == 0) {
    const float s = lane < L ? sS[lane] : -INFINITY;
    const float m = wave_max(s);
    float ex = 0.f;
    if (lane < L) ex = __expf(s - m) * (mask ? mask[row0 + lane] : 1.f);
    const float sum = wave_sum(ex);
    const float a = ex / (sum + 1e-8f * __expf(-m));
    if (lane < L) {
      sS[lane] = a;
      alpha[row0 + lane] = a;
    }
  }
  __syncthreads();
  // out[c] = sum_l a_l x[l][c]: thread = (column chunk, row group); row groups reduced through LDS
  const int nc = N / CH;
  if (nc <= 256) {
    const int nrg = 256 / nc, cx = tid % nc, rg = tid / nc;
    if (rg < nrg) {
      float acc[CH];
#pragma unroll
      for (int k = 0; k < CH; ++k) acc[k] = 0.f;
      for (int l = rg; l < L; l += nrg) {
        float f[CH];
        load_chunk_f<T>(x + (row0 + l) * N + cx * CH, f);
        const float a = sS[l];
#pragma unroll
        for (int k = 0; k < CH; ++k) acc[k] = fmaf(a, f[k], acc[k]);
      }
#pragma unroll
      for (int k = 0; k < CH; ++k) sRed[rg * N + cx * CH + k] = acc[k];
    }
    __syncthreads();
    for (int c = tid; c < N; c += 256) {
      float v = 0.f;
      for (int r = 0; r < nrg; ++r) v += sRed[r * N + c];
      out[(size_t)seq * ld_out + c] = v;
    }
  } else {
    for (int c = tid; c < N; c += 256) {
      float acc = 0.f;
      for (int l = 0; l < L; ++l) acc = fmaf(sS[l], (float)x[(row0 + l) * N + c], acc);
      out[(size_t)seq * ld_out + c] = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------
// K5 backward core.  With a = e^/Z, Z = sum e^ + 1e-8:
//   dA_l = <g, x_l> ; ds_l = a_l (dA_l - sum_u a_u dA_u) ; dpre_l = ds_l w2 (1 - e_l^2)
//   dw2 += sum_l ds_l e_l ; db2 += sum_l ds_l
// The direct term a_l g of dx and dpre.W1 are produced together by the GEMM epilogue
// (EPI_POOLBWD).  Each workgroup handles POOL_SPB sequences and writes one partial row
// of (dw2 | db2); colsum_kernel reduces the partial rows deterministically.
// All tile accesses are 16-byte chunks.
// ------------------------------------------------------------------------------------------
constexpr int POOL_SPB = 8;

template <typename T>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ e,
                                                       const float* __restrict__ w2, const float* __restrict__ alpha,
                                                       const float* __restrict__ g, int ld_g, T* __restrict__ dpre,
                                                       float* __restrict__ partial, int n, int L, int N, int q, int spb,
                                                       const int32_t* __restrict__ seq_nz) {
  constexpr int CH = ChunkOf<T>::CH;
  __shared__ float sDA[64];
  __shared__ float sDS[64];
  __shared__ float sRed[POOL_RED_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qc = q / CH;                       // chunks per e row (<= 256: q <= 1024)
  const int nrg = 256 / qc, cx = tid % qc, rg = tid / qc;
  float accw[4] = {0.f, 0.f, 0.f, 0.f};        // dw2 columns tid, tid+256, ...
  float accb = 0.f;
  float w2c[CH];
#pragma unroll
  for (int k = 0; k < CH; ++k) w2c[k] = w2[cx * CH + k];
  for (int s = 0; s < spb; ++s) {
    const int seq = blockIdx.x * spb + s;
    if (seq >= n) break;
    const size_t row0 = (size_t)seq * L;
    const float* gr = g + (size_t)seq * ld_g;
    if (seq_nz != nullptr && seq_nz[seq] == 0) {
      // g == 0 for this sequence (a masked history slot): dA = 0, ds = 0, dpre = exact zeros, nothing for dw2 / db2
      if (rg < nrg)
        for (int l = rg; l < L; l += nrg) {
          float o[CH];
#pragma unroll
          for (int k = 0; k < CH; ++k) o[k] = 0.f;
          store_chunk_f<T>(dpre + (row0 + l) * q + cx * CH, o);
        }
      continue;
    }
    for (int l = wid; l < L; l += 4) {
      const float p = wave_row_dot<T>(x + (row0 + l) * N, gr, N / CH, lane);
      if (lane == 0) sDA[l] = p;
    }
    __syncthreads();
    if (wid == 0) {
      const float a = lane < L ? alpha[row0 + lane] : 0.f;
      const float dA = lane < L ? sDA[lane] : 0.f;
      const float rd = wave_sum(a * dA);
      if (lane < L) sDS[lane] = a * (dA - rd);
    }
    __syncthreads();
    float wacc[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) wacc[k] = 0.f;
    if (rg < nrg) {
      for (int l = rg; l < L; l += nrg) {
        float f[CH], o[CH];
        load_chunk_f<T>(e + (row0 + l) * q + cx * CH, f);
        const float ds = sDS[l];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          o[k] = ds * w2c[k] * (1.f - f[k] * f[k]);
          wacc[k] = fmaf(ds, f[k], wacc[k]);
        }
        store_chunk_f<T>(dpre + (row0 + l) * q + cx * CH, o);
      }
#pragma unroll
      for (int k = 0; k < CH; ++k) sRed[rg * q + cx * CH + k] = wacc[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = tid + k * 256;
      if (c < q) {
        float t = 0.f;
        for (int r = 0; r < nrg; ++r) t += sRed[r * q + c];
        accw[k] += t;
      }
    }
    if (tid == 0) {
      float t = 0.f;
      for (int l = 0; l < L; ++l) t += sDS[l];
      accb += t;
    }
    __syncthreads();
  }
  float* pr = partial + (size_t)blockIdx.x * (q + 1);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = tid + k * 256;
    if (c < q) pr[c] = accw[k];
  }
  if (tid == 0) pr[q] = accb;
}

// out[c] += sum_r in[r*ld + c]: workgroup (x, y) sums rows y, y + gridDim.y, ... of 64 columns and adds its partial
// with one atomic per column (gridDim.y <= 32 partials per column).
// Columns [0, split) go to out, columns [split, cols) to out2 (the pooling backward's dw2 | db2 in one launch).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, int rows, int cols, int ld,
                                                     float* __restrict__ out, int split, float* __restrict__ out2) {
  __shared__ float red[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  float s = 0.f;
  if (c < cols)
    for (int r = blockIdx.y * 4 + ry; r < rows; r += 4 * gridDim.y) s += in[(size_t)r * ld + c];
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && c < cols)
    nr_accum(c < split ? out + c : out2 + (c - split), red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx], nr_fix_on());
}

// ------------------------------------------------------------------------------------------
// K6 pad-doc blend
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void blend_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                 const float* __restrict__ pad, T* __restrict__ out, size_t rows, int N) {
  const size_t total = rows * (size_t)N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / N;
    const int c = (int)(i - r * N);
    float v = x[i];
    if (mask) {
      const float m = mask[r];
      v = v * m + pad[c] * (1.f - m);
    }
    out[i] = (T)v;
  }
}

// 4 consecutive columns per thread (N % 4 == 0, 16-byte aligned rows): one 16-byte load, one mask word and one 8- / 16-byte
// store per thread instead of four of each and a 64-bit division per element (20 -> ~10 us for the 512 x 50 history rows)
template <typename T>
__global__ __launch_bounds__(256) void blend_fwd4_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                         const float* __restrict__ pad, T* __restrict__ out, uint32_t rows, uint32_t N4) {
  const uint32_t total = rows * N4;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t r = i / N4, c = (i - r * N4) * 4u;
    f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)i * 4);
    if (mask) {
      const float m = mask[r];
      const f32x4 pv = *reinterpret_cast<const f32x4*>(pad + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] * m + pv[e] * (1.f - m);
    }
    if (sizeof(T) == 2) {
      const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      *reinterpret_cast<bf16x4*>(out + (size_t)i * 4) = o;
    } else {
      *reinterpret_cast<f32x4*>(out + (size_t)i * 4) = v;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void blend_bwd_kernel(const T* __restrict__ dout, const float* __restrict__ mask,
                                                        float* __restrict__ dx, float* __restrict__ dpad, int rows,
                                                        int N, int rows_per_block) {
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  for (int c = threadIdx.x; c < N; c += 256) {
    float acc = 0.f;
#pragma unroll 8
    for (int r = r0; r < r1; ++r) {
      const float d = (float)dout[(size_t)r * N + c];
      const float m = mask ? mask[r] : 1.f;
      dx[(size_t)r * N + c] = d * m;
      acc = fmaf(1.f - m, d, acc);
    }
    if (mask && dpad) nr_accum(dpad + c, acc, nr_fix_on());
  }
}

// N % 4 == 0: a thread owns 4 consecutive columns (8-byte bf16 loads, 16-byte fp32 stores; the scalar version above moved
// 2 / 4 bytes per lane: 48 us for 61 MB), 256 / (N / 4) rows in flight per workgroup, dpad reduced in LDS before the atomics
template <typename T>
__global__ __launch_bounds__(256) void blend_bwd4_kernel(const T* __restrict__ dout, const float* __restrict__ mask,
                                                         float* __restrict__ dx, float* __restrict__ dpad, int rows,
                                                         int N, int rows_per_block) {
  extern __shared__ float sAcc[];                                  // [rl][N]
  const int cg = N / 4, rl_n = 256 / cg;                           // column groups, rows handled side by side
  const int tid = threadIdx.x, rl = tid / cg, c = (tid - rl * cg) * 4;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (rl < rl_n) {
    for (int r = r0 + rl; r < r1; r += rl_n) {
      float d[4];
      if (sizeof(T) == 2) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(dout + (size_t)r * N + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = (float)v[e];
      } else {
        const f32x4 v = *reinterpret_cast<const f32x4*>(dout + (size_t)r * N + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = v[e];
      }
      const float m = mask ? mask[r] : 1.f;
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = d[e] * m; acc[e] = fmaf(1.f - m, d[e], acc[e]); }
      *reinterpret_cast<f32x4*>(dx + (size_t)r * N + c) = o;
    }
  }
  if (mask && dpad) {
    if (rl < rl_n) {
#pragma unroll
      for (int e = 0; e < 4; ++e) sAcc[rl * N + c + e] = acc[e];
    }
    __syncthreads();
    const bool det = nr_fix_on();
    for (int k = tid; k < N; k += 256) {
      float t = 0.f;
      for (int j = 0; j < rl_n; ++j) t += sAcc[j * N + k];
      nr_accum(dpad + k, t, det);
    }
  }
}

// ------------------------------------------------------------------------------------------
// K8 scorer + cross entropy.  One wave per impression.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void score_ce_fwd_kernel(const float* __restrict__ cand, int ld_cand,
                                                          const float* __restrict__ user,
                                                          const int64_t* __restrict__ label, float* __restrict__ score,
                                                          float* __restrict__ lossvec, int C, int N) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* u = user + (size_t)b * N;
  float my = -INFINITY;  // lane j keeps score j (C <= 64)
  // 8 candidates at a time: their loads are independent (one candidate after the other was a chain of C load latencies)
  for (int j0 = 0; j0 < C; j0 += 8) {
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = lane; c < N; c += 64) {
      const float uc = u[c];
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (j0 + t < C) p[t] = fmaf(cand[((size_t)b * C + j0 + t) * ld_cand + c], uc, p[t]);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float q = wave_sum(p[t]);
      if (lane == j0 + t) my = q;
    }
  }
  if (lane < C) score[(size_t)b * C + lane] = my;
  const float m = wave_max(my);
  const float ex = lane < C ? expf(my - m) : 0.f;
  const float lse = m + logf(wave_sum(ex));
  const int lab = (int)label[b];
  const float sl = __shfl(my, lab, 64);
  if (lane == 0) lossvec[b] = lse - sl;
}

__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] / (float)n;
}

__global__ __launch_bounds__(256) void score_ce_bwd_kernel(const float* __restrict__ cand, int ld_cand,
                                                           const float* __restrict__ user,
                                                           const int64_t* __restrict__ label,
                                                           const float* __restrict__ score,
                                                           const float* __restrict__ gloss,
                                                           const float* __restrict__ gscore, float inv_b,
                                                           float* __restrict__ dcand, int ld_dcand,
                                                           float* __restrict__ duser, int C, int N) {
  __shared__ float sD[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < 64) {
    const float s = tid < C ? score[(size_t)b * C + tid] : -INFINITY;
    const float m = wave_max(s);
    const float ex = tid < C ? expf(s - m) : 0.f;
    const float sum = wave_sum(ex);
    if (tid < C) {
      float d = (ex / sum - (tid == (int)label[b] ? 1.f : 0.f)) * (gloss ? gloss[0] * inv_b : 0.f);
      if (gscore) d += gscore[(size_t)b * C + tid];
      sD[tid] = d;
    }
  }
  __syncthreads();
  for (int c = tid; c < N; c += 256) {
    const float u = user[(size_t)b * N + c];
    float du = 0.f;
    for (int j = 0; j < C; ++j) {
      const float d = sD[j];
      dcand[((size_t)b * C + j) * ld_dcand + c] = d * u;
      du = fmaf(d, cand[((size_t)b * C + j) * ld_cand + c], du);
    }
    duser[(size_t)b * N + c] = du;
  }
}

__global__ __launch_bounds__(256) void score_eval_kernel(const float* __restrict__ news, int ld_news,
                                                         const int32_t* __restrict__ cand_ids,
                                                         const int32_t* __restrict__ imp_of,
                                                         const float* __restrict__ user, int ld_user,
                                                         float* __restrict__ score, int n_cand, int N) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n_cand) return;
  const float* nr = news + (size_t)cand_ids[i] * ld_news;
  const float* u = user + (size_t)imp_of[i] * ld_user;
  float p = 0.f;
  // 16-byte loads where the rows allow it (400 floats: 2 wave instructions per operand instead of 7 dword ones)
  if (((N | ld_news | ld_user) & 3) == 0 && ((((uintptr_t)news) | ((uintptr_t)user)) & 15) == 0) {
    for (int c = 4 * lane; c < N; c += 256) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(nr + c), b = *reinterpret_cast<const f32x4*>(u + c);
      p = fmaf(a[0], b[0], fmaf(a[1], b[1], fmaf(a[2], b[2], fmaf(a[3], b[3], p))));
    }
  } else {
    for (int c = lane; c < N; c += 64) p = fmaf(nr[c], u[c], p);
  }
  p = wave_sum(p);
  if (lane == 0) score[i] = p;
}

// ------------------------------------------------------------------------------------------
// K1 standalone row gather / scatter-add
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gather_fwd_kernel(const T* __restrict__ table, int ld_table,
                                                         const int32_t* __restrict__ ids, int n_ids, int ids_stride,
                                                         int cols, float* __restrict__ out, int ld_out) {
  // one wave per row, 4 rows per workgroup; lanes stride the row 16 bytes at a time
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= n_ids) return;
  const T* src = table + (size_t)ids[(size_t)r * ids_stride] * ld_table;
  float* dst = out + (size_t)r * ld_out;
  constexpr int CH = 16 / (int)sizeof(T);
  const bool vec = (cols % CH == 0) && (ld_table % CH == 0) && (ld_out % 4 == 0) &&
                   (((uintptr_t)table | (uintptr_t)out) & 15) == 0;
  if (vec) {
    for (int c = lane * CH; c < cols; c += 64 * CH) {
      const uint4 raw = *reinterpret_cast<const uint4*>(src + c);
      if (sizeof(T) == 4) {
        *reinterpret_cast<uint4*>(dst + c) = raw;
      } else {
        const bf16_t* h = reinterpret_cast<const bf16_t*>(&raw);
        f32x4 lo = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        f32x4 hi = {(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
        *reinterpret_cast<f32x4*>(dst + c) = lo;
        *reinterpret_cast<f32x4*>(dst + c + 4) = hi;
      }
    }
  } else {
    for (int c = lane; c < cols; c += 64) dst[c] = (float)src[c];
  }
}

__global__ __launch_bounds__(256) void gather_bwd_kernel(const float* __restrict__ dout, int ld_dout,
                                                         const int32_t* __restrict__ ids, int n_ids, int ids_stride,
                                                         int cols, float* __restrict__ dtable, int ld_dtable) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= n_ids) return;
  const int id = ids[(size_t)r * ids_stride];
  if (id == 0) return;  // padding_idx
  const bool det = nr_fix_on();
  for (int c = lane; c < cols; c += 64) nr_accum(dtable + (size_t)id * ld_dtable + c, dout[(size_t)r * ld_dout + c], det);
}

// ------------------------------------------------------------------------------------------
// parameter packing
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void cast_pad_kernel(const float* __restrict__ src, int rows, int cols, int ld_src, T* __restrict__ dst,
                                int ld_dst, int transpose) {
  const int drows = transpose ? cols : rows;
  const size_t total = (size_t)drows * ld_dst;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / ld_dst), c = (int)(i - (size_t)r * ld_dst);
    float v = 0.f;
    if (!transpose) {
      if (c < cols) v = src[(size_t)r * ld_src + c];
    } else {
      if (c < rows) v = src[(size_t)c * ld_src + r];
    }
    dst[i] = (T)v;
  }
}

// several cast_pad jobs in one launch (the weights a training step re-packs after every optimizer step are tiny: nine
// 6-us launches).  blk0[j] = first block of job j; a block works on 1024 consecutive destination elements of its job.
struct CastJobs {
  const float* src[NR_CAST_BATCH_MAX];
  void* dst[NR_CAST_BATCH_MAX];
  int rows[NR_CAST_BATCH_MAX], cols[NR_CAST_BATCH_MAX], ld_src[NR_CAST_BATCH_MAX], ld_dst[NR_CAST_BATCH_MAX], transpose[NR_CAST_BATCH_MAX];
  int blk0[NR_CAST_BATCH_MAX + 1];
  int n;
};
template <typename T>
__global__ __launch_bounds__(256) void cast_pad_batch_kernel(CastJobs J) {
  int j = 0;
  while (j + 1 < J.n && (int)blockIdx.x >= J.blk0[j + 1]) ++j;        // uniform
  const int rows = J.rows[j], cols = J.cols[j], ld_src = J.ld_src[j], ld_dst = J.ld_dst[j], transpose = J.transpose[j];
  const float* __restrict__ src = J.src[j];
  T* __restrict__ dst = reinterpret_cast<T*>(J.dst[j]);
  const size_t total = (size_t)(transpose ? cols : rows) * ld_dst;
  const size_t i0 = (size_t)((int)blockIdx.x - J.blk0[j]) * 1024;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const size_t i = i0 + u * 256 + threadIdx.x;
    if (i >= total) break;
    const int r = (int)(i / ld_dst), c = (int)(i - (size_t)r * ld_dst);
    float v = 0.f;
    if (!transpose) {
      if (c < cols) v = src[(size_t)r * ld_src + c];
    } else {
      if (c < rows) v = src[(size_t)c * ld_src + r];
    }
    dst[i] = (T)v;
  }
}

template <typename T>
__global__ void pack_conv_w_kernel(const float* __restrict__ w, int N, int D, T* __restrict__ dst, int Dp) {
  const size_t total = (size_t)N * 3 * Dp;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / (3 * Dp)), r = (int)(i - (size_t)n * 3 * Dp), tap = r / Dp, d = r - tap * Dp;
    dst[i] = (T)(d < D ? w[((size_t)n * D + d) * 3 + tap] : 0.f);
  }
}

__global__ void unpack_conv_dw_kernel(const float* __restrict__ dwp, int N, int D, int Dp, float* __restrict__ dw, int accumulate) {
  const size_t total = (size_t)N * D * 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / (3 * D)), r = (int)(i - (size_t)n * 3 * D), d = r / 3, tap = r - d * 3;
    const float v = dwp[(size_t)n * 3 * Dp + tap * Dp + d];
    dw[i] = accumulate ? dw[i] + v : v;
  }
}

// fp32 rows (stride ld_src) -> dtype rows (stride ld_dst, zero padded)
template <typename T>
__global__ void cast_rows_kernel(const float* __restrict__ src, int ld_src, T* __restrict__ dst, int ld_dst, int rows,
                                 int cols) {
  const size_t total = (size_t)rows * ld_dst;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / ld_dst;
    const int c = (int)(i - r * ld_dst);
    dst[i] = (T)(c < cols ? src[r * ld_src + c] : 0.f);
  }
}

__global__ void dropout_mask_kernel(float* __restrict__ out, uint32_t count, uint32_t key, uint32_t thresh) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x)
    out[i] = (thresh == 0 || nr_keep(key, i, thresh)) ? 1.f : 0.f;
}

inline int grid_for(size_t total, int block = 256, int cap = 256 * 16) {
  size_t g = (total + block - 1) / block;
  if (g < 1) g = 1;
  if (g > (size_t)cap) g = cap;
  return (int)g;
}

}  // namespace

// ---- launchers used by nr_api.hip ---------------------------------------------------------
int nr_launch_pool_core_fwd(int dtype, const void* x, const void* e, const float* w2, const float* b2, const float* mask,
                            float* alpha, float* out, int ld_out, int n, int L, int N, int q, hipStream_t s, const int32_t* needed) {
  NR_CHECK_ARG(L >= 1 && L <= 64, "additive_pool: L=%d must be in [1, 64]", L);
  NR_CHECK_ARG((((uintptr_t)x | (uintptr_t)e) & 15) == 0 && q <= 1024, "additive_pool: x / e must be 16-byte aligned, q <= 1024");
  NrProfScope ps(s, "pool_core_fwd[n=%d,L=%d,N=%d,q=%d]", n, L, N, q);
  if (dtype == NR_BF16)
    hipLaunchKernelGGL(pool_fwd_kernel<bf16_t>, dim3(n), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)e, w2, b2, mask, alpha, out, ld_out, L, N, q, needed);
  else
    hipLaunchKernelGGL(pool_fwd_kernel<float>, dim3(n), dim3(256), 0, s, (const float*)x, (const float*)e, w2, b2, mask, alpha, out, ld_out, L, N, q, needed);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

// sequences per workgroup: 8 at the news level; fewer when there are few sequences (user level, n = batch), so that the
// grid still covers the chip.  The caller's partial buffer holds ceil(n / 8) rows at least and n rows at most.
static int pool_spb(int n) { return n >= 8192 ? POOL_SPB : (n >= 2048 ? 2 : 1); }
int nr_pool_partial_rows(int n) { const int spb = pool_spb(n); return (n + spb - 1) / spb; }

int nr_launch_pool_core_bwd(int dtype, const void* x, const void* e, const float* w2, const float* alpha, const float* g,
                            int ld_g, void* dpre, float* partial, float* dw2, float* db2, int n, int L, int N, int q,
                            hipStream_t s, const int32_t* seq_nz) {
  NR_CHECK_ARG(L >= 1 && L <= 64, "additive_pool: L=%d must be in [1, 64]", L);
  NR_CHECK_ARG(q <= 1024, "additive_pool: q=%d must be <= 1024", q);
  NR_CHECK_ARG((((uintptr_t)x | (uintptr_t)e | (uintptr_t)dpre | (uintptr_t)g) & 15) == 0 && ld_g % 4 == 0,
               "additive_pool_bwd: x / e / dpre / g must be 16-byte aligned");
  const int nb = nr_pool_partial_rows(n);
  NrProfScope ps(s, "pool_core_bwd[n=%d,L=%d,N=%d,q=%d]", n, L, N, q);
  if (dtype == NR_BF16)
    hipLaunchKernelGGL(pool_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)e, w2, alpha, g, ld_g, (bf16_t*)dpre, partial, n, L, N, q, pool_spb(n), seq_nz);
  else
    hipLaunchKernelGGL(pool_bwd_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)x, (const float*)e, w2, alpha, g, ld_g, (float*)dpre, partial, n, L, N, q, pool_spb(n), seq_nz);
  NR_CHECK_LAUNCH();
  const int ysplit = nb >= 512 ? 32 : (nb >= 64 ? 8 : 1);
  hipLaunchKernelGGL(colsum_kernel, dim3((q + 1 + 63) / 64, ysplit), dim3(256), 0, s, partial, nb, q + 1, q + 1, dw2, q, db2);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

// out[c] += column sums of in[rows, cols] for c < split, out2[c - split] for the rest (the pooling backward's dw2 | db2)
int nr_launch_colsum_split(const float* in, int rows, int cols, int ld, float* out, int split, float* out2, hipStream_t s) {
  const int ysplit = rows >= 512 ? 32 : (rows >= 64 ? 8 : 1);
  hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, ysplit), dim3(256), 0, s, in, rows, cols, ld, out, split, out2);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

namespace {
__global__ __launch_bounds__(256) void fix_flush_kernel(float* __restrict__ out, long long* __restrict__ fix, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const long long v = fix[i];
    if (v != 0) {
      out[i] += (float)((double)v * (1.0 / 68719476736.0));
      fix[i] = 0;
    }
  }
}
}  // namespace
int nr_fix_set_pool(const NrFixTable* t, hipStream_t s) {
  NR_CHECK_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_nr_fix), t, sizeof(NrFixTable), 0, hipMemcpyHostToDevice, s));
  return NR_OK;
}
int nr_fix_flush(const NrFixTable* t, hipStream_t s) {
  for (int i = 0; i < t->count; ++i) {
    const size_t n = (size_t)t->n[i];
    if (n == 0) continue;
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(fix_flush_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, s, const_cast<float*>(t->base[i]), t->fix[i], n);
  }
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_launch_cast_rows(int dtype, const float* src, int ld_src, void* dst, int ld_dst, int rows, int cols, hipStream_t s) {
  const size_t total = (size_t)rows * ld_dst;
  if (dtype == NR_BF16)
    hipLaunchKernelGGL(cast_rows_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, src, ld_src, (bf16_t*)dst, ld_dst, rows, cols);
  else
    hipLaunchKernelGGL(cast_rows_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, src, ld_src, (float*)dst, ld_dst, rows, cols);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

extern "C" {

int nr_cast_pad(const float* src, int rows, int cols, int ld_src, void* dst, int ld_dst, int dtype, int transpose,
                nr_stream_t stream) {
  NR_CHECK_ARG(src && dst && rows > 0 && cols > 0, "cast_pad: null/empty");
  NR_CHECK_ARG(ld_dst >= (transpose ? rows : cols) && ld_src >= cols, "cast_pad: leading dimensions too small");
  NR_DEVICE_GUARD(stream, dst);
  const size_t total = (size_t)(transpose ? cols : rows) * ld_dst;
  hipStream_t s = (hipStream_t)stream;
  NrProfScope ps(s, "cast_pad[rows=%d,cols=%d]", rows, cols);
  if (dtype == NR_BF16)
    hipLaunchKernelGGL(cast_pad_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, src, rows, cols, ld_src, (bf16_t*)dst, ld_dst, transpose);
  else
    hipLaunchKernelGGL(cast_pad_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, src, rows, cols, ld_src, (float*)dst, ld_dst, transpose);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_cast_pad_batch(const nr_cast_job* jobs, int n, int dtype, nr_stream_t stream) {
  NR_CHECK_ARG(jobs != nullptr && n >= 1 && n <= NR_CAST_BATCH_MAX, "cast_pad_batch: 1..%d jobs", NR_CAST_BATCH_MAX);
  CastJobs J;
  J = CastJobs();
  J.n = n;
  int blocks = 0;
  for (int j = 0; j < n; ++j) {
    const nr_cast_job& q = jobs[j];
    NR_CHECK_ARG(q.src && q.dst && q.rows > 0 && q.cols > 0, "cast_pad_batch: job %d null/empty", j);
    NR_CHECK_ARG(q.ld_dst >= (q.transpose ? q.rows : q.cols) && q.ld_src >= q.cols, "cast_pad_batch: job %d leading dimensions too small", j);
    J.src[j] = q.src; J.dst[j] = q.dst; J.rows[j] = q.rows; J.cols[j] = q.cols; J.ld_src[j] = q.ld_src; J.ld_dst[j] = q.ld_dst;
    J.transpose[j] = q.transpose;
    J.blk0[j] = blocks;
    const size_t total = (size_t)(q.transpose ? q.cols : q.rows) * q.ld_dst;
    NR_CHECK_ARG(total < ((size_t)1 << 30), "cast_pad_batch: job %d too large for the batched kernel", j);
    blocks += (int)((total + 1023) / 1024);
  }
  J.blk0[n] = blocks;
  NR_DEVICE_GUARD(stream, jobs[0].dst);
  hipStream_t s = (hipStream_t)stream;
  NrProfScope ps(s, "cast_pad_batch[jobs=%d]", n);
  if (dtype == NR_BF16) hipLaunchKernelGGL(cast_pad_batch_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, J);
  else hipLaunchKernelGGL(cast_pad_batch_kernel<float>, dim3(blocks), dim3(256), 0, s, J);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_pack_conv_w(const float* w, int N, int D, void* dst, int Dp, int dtype, nr_stream_t stream) {
  NR_CHECK_ARG(w && dst && N > 0 && D > 0 && Dp >= D, "pack_conv_w: bad arguments");
  NR_DEVICE_GUARD(stream, dst);
  const size_t total = (size_t)N * 3 * Dp;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == NR_BF16)
    hipLaunchKernelGGL(pack_conv_w_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, w, N, D, (bf16_t*)dst, Dp);
  else
    hipLaunchKernelGGL(pack_conv_w_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, w, N, D, (float*)dst, Dp);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_unpack_conv_dw(const float* dw_pack, int N, int D, int Dp, float* dw, int accumulate, nr_stream_t stream) {
  NR_CHECK_ARG(dw_pack && dw && N > 0 && D > 0 && Dp >= D, "unpack_conv_dw: bad arguments");
  NR_DEVICE_GUARD(stream, dw);
  hipLaunchKernelGGL(unpack_conv_dw_kernel, dim3(grid_for((size_t)N * D * 3)), dim3(256), 0, (hipStream_t)stream, dw_pack, N, D, Dp, dw, accumulate);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_embed_gather_fwd(const void* table, int ld_table, int dtype, const int32_t* ids, int n_ids, int ids_stride,
                        int cols, float* out, int ld_out, nr_stream_t stream) {
  NR_CHECK_ARG(table && ids && out, "embed_gather_fwd: null pointer");
  NR_DEVICE_GUARD(stream, out);
  if (n_ids == 0) return NR_OK;
  NR_CHECK_ARG(n_ids > 0 && cols > 0 && ids_stride >= 1, "embed_gather_fwd: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == NR_BF16)
    hipLaunchKernelGGL(gather_fwd_kernel<bf16_t>, dim3((n_ids + 3) / 4), dim3(256), 0, s, (const bf16_t*)table, ld_table, ids, n_ids, ids_stride, cols, out, ld_out);
  else
    hipLaunchKernelGGL(gather_fwd_kernel<float>, dim3((n_ids + 3) / 4), dim3(256), 0, s, (const float*)table, ld_table, ids, n_ids, ids_stride, cols, out, ld_out);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

namespace {
// fp32 table rows -> bf16 rows (gather + cast in one pass): one wave per row, 8 elements (32 B in, 16 B out) per lane step
__global__ __launch_bounds__(256) void gather_cast_kernel(const float* __restrict__ table, int ld_table, const int32_t* __restrict__ ids,
                                                          int n_ids, int cols, bf16_t* __restrict__ out, int ld_out) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= n_ids) return;
  const float* src = table + (size_t)ids[r] * ld_table;
  bf16_t* dst = out + (size_t)r * ld_out;
  const bool vec = (cols % 8 == 0) && (ld_table % 4 == 0) && (ld_out % 8 == 0) && ((((uintptr_t)table) | ((uintptr_t)out)) & 15) == 0;
  if (vec) {
    for (int c = lane * 8; c < cols; c += 64 * 8) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src + c), b = *reinterpret_cast<const f32x4*>(src + c + 4);
      const bf16x8 o = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
      *reinterpret_cast<bf16x8*>(dst + c) = o;
    }
  } else {
    for (int c = lane; c < cols; c += 64) dst[c] = (bf16_t)src[c];
  }
}
}  // namespace

int nr_gather_cast_fwd(const float* table, int ld_table, const int32_t* ids, int n_ids, int cols, void* out, int ld_out, int out_dtype,
                       nr_stream_t stream) {
  NR_CHECK_ARG(table && ids && out, "gather_cast_fwd: null pointer");
  NR_CHECK_ARG(out_dtype == NR_F32 || out_dtype == NR_BF16, "gather_cast_fwd: bad dtype %d", out_dtype);
  if (out_dtype == NR_F32) return nr_embed_gather_fwd(table, ld_table, NR_F32, ids, n_ids, 1, cols, (float*)out, ld_out, stream);
  NR_DEVICE_GUARD(stream, out);
  if (n_ids == 0) return NR_OK;
  NR_CHECK_ARG(n_ids > 0 && cols > 0 && ld_out >= cols && ld_table >= cols, "gather_cast_fwd: bad sizes");
  NrProfScope ps((hipStream_t)stream, "gather_cast[n=%d,cols=%d]", n_ids, cols);
  hipLaunchKernelGGL(gather_cast_kernel, dim3((n_ids + 3) / 4), dim3(256), 0, (hipStream_t)stream, table, ld_table, ids, n_ids, cols, (bf16_t*)out, ld_out);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_embed_gather_bwd(const float* dout, int ld_dout, const int32_t* ids, int n_ids, int ids_stride, int cols,
                        float* dtable, int ld_dtable, nr_stream_t stream) {
  NR_CHECK_ARG(dout && ids && dtable, "embed_gather_bwd: null pointer");
  NR_DEVICE_GUARD(stream, dtable);
  if (n_ids == 0) return NR_OK;
  hipLaunchKernelGGL(gather_bwd_kernel, dim3((n_ids + 3) / 4), dim3(256), 0, (hipStream_t)stream, dout, ld_dout, ids, n_ids, ids_stride, cols, dtable, ld_dtable);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_pad_blend_fwd(const float* x, const float* mask, const float* pad, void* out, int n, int L, int N, int dtype,
                     nr_stream_t stream) {
  NR_CHECK_ARG(x && out && n > 0 && L > 0 && N > 0, "pad_blend_fwd: null/empty");
  NR_CHECK_ARG(mask == nullptr || pad != nullptr, "pad_blend_fwd: mask without pad_doc");
  NR_DEVICE_GUARD(stream, out);
  const size_t rows = (size_t)n * L;
  hipStream_t s = (hipStream_t)stream;
  NrProfScope ps(s, "pad_blend_fwd[n=%d,L=%d,N=%d]", n, L, N);
  if (N % 4 == 0 && rows * (size_t)(N / 4) < (1ull << 31) && (((uintptr_t)x | (uintptr_t)out | (uintptr_t)pad) & 15) == 0) {
    const uint32_t N4 = (uint32_t)(N / 4);
    const dim3 grid(grid_for(rows * N4));
    if (dtype == NR_BF16) hipLaunchKernelGGL(blend_fwd4_kernel<bf16_t>, grid, dim3(256), 0, s, x, mask, pad, (bf16_t*)out, (uint32_t)rows, N4);
    else hipLaunchKernelGGL(blend_fwd4_kernel<float>, grid, dim3(256), 0, s, x, mask, pad, (float*)out, (uint32_t)rows, N4);
    NR_CHECK_LAUNCH();
    return NR_OK;
  }
  if (dtype == NR_BF16)
    hipLaunchKernelGGL(blend_fwd_kernel<bf16_t>, dim3(grid_for(rows * N)), dim3(256), 0, s, x, mask, pad, (bf16_t*)out, rows, N);
  else
    hipLaunchKernelGGL(blend_fwd_kernel<float>, dim3(grid_for(rows * N)), dim3(256), 0, s, x, mask, pad, (float*)out, rows, N);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_pad_blend_bwd(const void* dout, const float* mask, float* dx, float* dpad, int n, int L, int N, int dtype,
                     nr_stream_t stream) {
  NR_CHECK_ARG(dout && dx && n > 0 && L > 0 && N > 0, "pad_blend_bwd: null/empty");
  NR_DEVICE_GUARD(stream, dx);
  const int rows = n * L, rpb = 32;
  hipStream_t s = (hipStream_t)stream;
  int det_rc = NR_OK;
  const int det = (mask != nullptr && dpad != nullptr) ? nr_det_open(s, dpad, (size_t)N, nullptr, 0, false, true, &det_rc) : 0;
  if (det_rc) return det_rc;
  NrProfScope ps(s, "pad_blend_bwd[n=%d,L=%d,N=%d]", n, L, N);
  if (N % 4 == 0 && N / 4 <= 256 && (((uintptr_t)dout | (uintptr_t)dx) & 15) == 0) {
    const int rl_n = 256 / (N / 4), rpb4 = 64;
    const size_t smem = (size_t)rl_n * N * sizeof(float);
    if (dtype == NR_BF16)
      hipLaunchKernelGGL(blend_bwd4_kernel<bf16_t>, dim3((rows + rpb4 - 1) / rpb4), dim3(256), smem, s, (const bf16_t*)dout, mask, dx, dpad, rows, N, rpb4);
    else
      hipLaunchKernelGGL(blend_bwd4_kernel<float>, dim3((rows + rpb4 - 1) / rpb4), dim3(256), smem, s, (const float*)dout, mask, dx, dpad, rows, N, rpb4);
    NR_CHECK_LAUNCH();
    return nr_det_close(det);
  }
  if (dtype == NR_BF16)
    hipLaunchKernelGGL(blend_bwd_kernel<bf16_t>, dim3((rows + rpb - 1) / rpb), dim3(256), 0, s, (const bf16_t*)dout, mask, dx, dpad, rows, N, rpb);
  else
    hipLaunchKernelGGL(blend_bwd_kernel<float>, dim3((rows + rpb - 1) / rpb), dim3(256), 0, s, (const float*)dout, mask, dx, dpad, rows, N, rpb);
  NR_CHECK_LAUNCH();
  return nr_det_close(det);
}

int nr_score_ce_fwd(const float* cand, int ld_cand, const float* user, const int64_t* label, float* score, float* loss,
                    float* lossvec, int B, int C, int N, nr_stream_t stream) {
  NR_CHECK_ARG(cand && user && label && score && loss && lossvec, "score_ce_fwd: null pointer");
  NR_CHECK_ARG(B > 0 && C >= 1 && C <= 64 && N > 0, "score_ce_fwd: B=%d C=%d (1..64) N=%d", B, C, N);
  NR_DEVICE_GUARD(stream, score);
  hipStream_t s = (hipStream_t)stream;
  NrProfScope ps(s, "score_ce_fwd[B=%d,C=%d,N=%d]", B, C, N);
  hipLaunchKernelGGL(score_ce_fwd_kernel, dim3(B), dim3(64), 0, s, cand, ld_cand, user, label, score, lossvec, C, N);
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, s, lossvec, B, loss);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_score_ce_bwd(const float* cand, int ld_cand, const float* user, const int64_t* label, const float* score,
                    const float* gloss, const float* gscore, float* dcand, int ld_dcand, float* duser, int B, int C, int N,
                    nr_stream_t stream) {
  NR_CHECK_ARG(cand && user && label && score && dcand && duser, "score_ce_bwd: null pointer");
  NR_CHECK_ARG(B > 0 && C >= 1 && C <= 64 && N > 0, "score_ce_bwd: B=%d C=%d (1..64) N=%d", B, C, N);
  NR_DEVICE_GUARD(stream, dcand);
  NrProfScope ps((hipStream_t)stream, "score_ce_bwd[B=%d,C=%d,N=%d]", B, C, N);
  hipLaunchKernelGGL(score_ce_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, cand, ld_cand, user, label, score, gloss, gscore, 1.0f / (float)B, dcand, ld_dcand, duser, C, N);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_score_eval(const float* news_vecs, int ld_news, const int32_t* cand_ids, const int32_t* imp_of, const float* user,
                  int ld_user, float* score, int n_cand, int N, nr_stream_t stream) {
  NR_CHECK_ARG(news_vecs && cand_ids && imp_of && user && score, "score_eval: null pointer");
  NR_DEVICE_GUARD(stream, score);
  if (n_cand == 0) return NR_OK;
  hipLaunchKernelGGL(score_eval_kernel, dim3((n_cand + 3) / 4), dim3(256), 0, (hipStream_t)stream, news_vecs, ld_news, cand_ids, imp_of, user, ld_user, score, n_cand, N);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_dropout_mask(float* out, uint32_t count, float p, uint32_t seed, nr_stream_t stream) {
  NR_CHECK_ARG(out != nullptr, "dropout_mask: null pointer");
  NR_DEVICE_GUARD(stream, out);
  if (count == 0) return NR_OK;
  const DropCfg d = nr_make_drop(p, seed);
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, out, count, d.key, d.thresh);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

}  // extern "C"
