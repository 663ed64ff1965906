// Callers either side of the encoder path, on the device:
//   nr_assemble_batch  - DatasetTrain.line_mapper's news_combined gather + positive splice + DataLoader collate
//                        (src/dataset.py:26-49, src/main.py:89-103): only news INDICES cross PCIe, once per epoch
//   nr_eval_metrics    - per-impression AUC / MRR / nDCG@5 / nDCG@10 over variable-length candidate lists
//                        (src/metrics.py:5-23, sklearn roc_auc_score, src/main.py:249-263) and their sums
//   nr_adam_step       - torch.optim.Adam's update rule (src/main.py:76,110) over ONE flat fp32 bucket, with the
//                        all-reduce averaging factor and the next step's zero_grad folded in
// All of it is HBM-bound integer / elementwise work: coalesced 16-byte accesses, no MFMA.
#include "nr_common.h"

namespace {

// ------------------------------------------------------------------------------------------ batch assembly
// One thread per output element.  Slot s of impression b: s < H -> history slot, else candidate slot j = s - H with
// news index  j < label ? neg[j] : (j == label ? pos : neg[j - 1])   (sample_news = neg[:label] + [pos] + neg[label:]).
__global__ __launch_bounds__(256) void assemble_kernel(const int32_t* __restrict__ comb, int n_rows, int F, const int32_t* __restrict__ hist_idx,
                                                        const int32_t* __restrict__ pos_idx, const int32_t* __restrict__ neg_idx,
                                                        const int64_t* __restrict__ label, int B, int H, int K, int32_t* __restrict__ history,
                                                        int32_t* __restrict__ candidate, int32_t* __restrict__ bad) {
  const int S = H + 1 + K;
  const long total = (long)B * S * F;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int f = (int)(e % F);
    const long bs = e / F;
    const int s = (int)(bs % S), b = (int)(bs / S);
    int idx;
    int32_t* dst;
    if (s < H) {
      idx = hist_idx[(size_t)b * H + s];
      dst = history + ((size_t)b * H + s) * F + f;
    } else {
      const int j = s - H;
      long lab = label[b];
      if (lab < 0 || lab > K) {                         // torch would raise on an out-of-range class index later on
        if (f == 0 && j == 0 && bad != nullptr) atomicAdd(bad, 1);
        lab = 0;
      }
      idx = j < lab ? neg_idx[(size_t)b * K + j] : (j == lab ? pos_idx[b] : neg_idx[(size_t)b * K + j - 1]);
      dst = candidate + ((size_t)b * (1 + K) + j) * F + f;
    }
    if (idx < 0 || idx >= n_rows) {                     // numpy fancy indexing would raise IndexError (src/dataset.py:48)
      if (f == 0 && bad != nullptr) atomicAdd(bad, 1);
      idx = 0;
    }
    *dst = comb[(size_t)idx * F + f];
  }
}

// ------------------------------------------------------------------------------------------ ranking metrics
constexpr int MET_MAXC = 4096;   // candidates of one impression held in LDS

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// One wave per impression.  Descending rank of candidate i with ties broken towards the LATER index first -- the order of
// np.argsort(score, kind="stable")[::-1]; numpy's default sort (src/metrics.py:6,20) leaves the order of tied scores
// unspecified, any of them is "the reference's".  AUC is the Mann-Whitney statistic with half credit for ties, which is
// what sklearn's roc_auc_score (trapezoidal ROC area) evaluates to.
__global__ __launch_bounds__(256) void metrics_kernel(const float* __restrict__ score, const int32_t* __restrict__ label,
                                                       const int32_t* __restrict__ offsets, int n_imp, double* __restrict__ per_imp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  float* sS = reinterpret_cast<float*>(smem) + (size_t)wid * MET_MAXC;
  unsigned char* sL = reinterpret_cast<unsigned char*>(reinterpret_cast<float*>(smem) + 4 * (size_t)MET_MAXC) + (size_t)wid * MET_MAXC;
  for (int imp = blockIdx.x * 4 + wid; imp < n_imp; imp += gridDim.x * 4) {
    const int o0 = offsets[imp], c = offsets[imp + 1] - o0;
    int npos = 0;
    for (int i = lane; i < c; i += 64) {
      sS[i] = score[o0 + i];
      const int l = label[o0 + i] != 0;
      sL[i] = (unsigned char)l;
      npos += l;
    }
    npos = (int)wave_sum_d((double)npos);
    const int nneg = c - npos;
    double auc = 0.0, rr = 0.0, d5 = 0.0, d10 = 0.0;
    __builtin_amdgcn_s_waitcnt(0);                     // wave-private LDS region: writes above are visible to the wave
    __builtin_amdgcn_wave_barrier();
    if (npos > 0 && nneg > 0) {
      for (int i = lane; i < c; i += 64) {
        if (!sL[i]) continue;
        const float si = sS[i];
        int greater = 0, eq_after = 0, less_neg = 0, eq_neg = 0;
        for (int j = 0; j < c; ++j) {
          const float sj = sS[j];
          const int lj = sL[j];
          greater += sj > si;
          eq_after += (sj == si) & (j > i);
          less_neg += (sj < si) & !lj;
          eq_neg += (sj == si) & !lj;
        }
        const int r = greater + eq_after;              // 0-based position in the descending order
        auc += (double)less_neg + 0.5 * (double)eq_neg;
        rr += 1.0 / (double)(r + 1);
        const double g = 1.0 / log2((double)(r + 2));
        if (r < 5) d5 += g;
        if (r < 10) d10 += g;
      }
    }
    auc = wave_sum_d(auc); rr = wave_sum_d(rr); d5 = wave_sum_d(d5); d10 = wave_sum_d(d10);
    if (lane == 0) {
      double* o = per_imp + (size_t)imp * 4;
      if (npos > 0 && nneg > 0) {
        double b5 = 0.0, b10 = 0.0;                    // ideal DCG: the positives first
        for (int r = 0; r < min(npos, 10); ++r) {
          const double g = 1.0 / log2((double)(r + 2));
          if (r < 5) b5 += g;
          b10 += g;
        }
        o[0] = auc / ((double)npos * (double)nneg);
        o[1] = rr / (double)npos;
        o[2] = d5 / b5;
        o[3] = d10 / b10;
      } else {                                         // src/main.py:250: impressions with one class only are skipped
        o[0] = -1.0; o[1] = 0.0; o[2] = 0.0; o[3] = 0.0;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// fixed-order reduction of the per-impression values: sums[0] = scored impressions, sums[1..4] = metric sums
__global__ __launch_bounds__(1024) void metrics_reduce_kernel(const double* __restrict__ per_imp, int n_imp, double* __restrict__ sums) {
  __shared__ double sh[5][16];
  double a[5] = {0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < n_imp; i += 1024) {
    const double* o = per_imp + (size_t)i * 4;
    if (o[0] >= 0.0) {
      a[0] += 1.0; a[1] += o[0]; a[2] += o[1]; a[3] += o[2]; a[4] += o[3];
    }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const double v = wave_sum_d(a[k]);
    if (lane == 0) sh[k][wid] = v;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    double v = 0.0;
    for (int w = 0; w < 16; ++w) v += sh[threadIdx.x][w];
    sums[threadIdx.x] = v;
  }
}

// ------------------------------------------------------------------------------------------ Adam over a flat bucket
struct AdamCfg {
  float beta1, beta2, eps, step_size, bc2_sqrt, grad_scale;
  int zero_grad;
};

__device__ __forceinline__ void adam1(float& p, float& g, float& m, float& v, const AdamCfg& c) {
  const float gr = g * c.grad_scale;
  m = fmaf(1.f - c.beta1, gr - m, m);                             // exp_avg.lerp_(grad, 1 - beta1)
  v = fmaf(c.beta2, v, (1.f - c.beta2) * gr * gr);                // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;              // sqrt(v) / sqrt(1 - beta2^t) + eps
  p -= c.step_size * (m / denom);                                 // step_size = lr / (1 - beta1^t)
  if (c.zero_grad) g = 0.f;
}

// Packed (bf16, row-padded) copies of parameter matrices that live in the bucket: written in the same pass as the update
// itself, bit-identical to what nr_cast_pad makes of the new values (the separate re-pack of a 30 000 x 300 word table was
// 15 us per step).  first / count in elements, both multiples of 4, cols a multiple of 4: a thread's 4 values share a row.
struct AdamPacks {
  int n;
  unsigned long long first[NR_ADAM_PACK_MAX], count[NR_ADAM_PACK_MAX];
  int cols[NR_ADAM_PACK_MAX], ld[NR_ADAM_PACK_MAX];
  bf16_t* dst[NR_ADAM_PACK_MAX];
};

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    size_t n, AdamCfg c, AdamPacks pk) {
  const size_t n4 = n / 4;
  f32x4* p4 = reinterpret_cast<f32x4*>(p);
  f32x4* g4 = reinterpret_cast<f32x4*>(g);
  f32x4* m4 = reinterpret_cast<f32x4*>(m);
  f32x4* v4 = reinterpret_cast<f32x4*>(v);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 pv = p4[i], gv = g4[i], mv = m4[i], vv = v4[i];
    float pp[4] = {pv[0], pv[1], pv[2], pv[3]}, gg[4] = {gv[0], gv[1], gv[2], gv[3]}, mm[4] = {mv[0], mv[1], mv[2], mv[3]},
          ww[4] = {vv[0], vv[1], vv[2], vv[3]};
#pragma unroll
    for (int e = 0; e < 4; ++e) adam1(pp[e], gg[e], mm[e], ww[e], c);
    p4[i] = (f32x4){pp[0], pp[1], pp[2], pp[3]};
    for (int j = 0; j < pk.n; ++j) {
      const unsigned long long e = (unsigned long long)i * 4 - pk.first[j];       // (wraps to a huge value in front of the range)
      if (e < pk.count[j]) {
        const uint32_t r = (uint32_t)e / (uint32_t)pk.cols[j], col = (uint32_t)e - r * (uint32_t)pk.cols[j];
        const bf16x4 o = {(bf16_t)pp[0], (bf16_t)pp[1], (bf16_t)pp[2], (bf16_t)pp[3]};
        *reinterpret_cast<bf16x4*>(pk.dst[j] + (size_t)r * pk.ld[j] + col) = o;
      }
    }
    m4[i] = (f32x4){mm[0], mm[1], mm[2], mm[3]};
    v4[i] = (f32x4){ww[0], ww[1], ww[2], ww[3]};
    if (c.zero_grad) g4[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const size_t tail = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x;
  if (tail < n) adam1(p[tail], g[tail], m[tail], v[tail], c);
}



// ------------------------------------------------------------------------------------------ one encoder batch
// out = [a ; b] (rows of F int32 ids), flags = [1 ... 1 ; mask_b != 0]: what Model.forward hands the news encoder -- the
// candidate titles followed by the history titles and the "is this title's vector used at all" flags -- in one launch
// (torch needed a fill, two concatenations, a compare and a cast: five launches of ~5 us each).
__global__ __launch_bounds__(256) void stack_rows_kernel(const int32_t* __restrict__ a, long na, const int32_t* __restrict__ b, long nb, int F,
                                                         const float* __restrict__ mask_b, int32_t* __restrict__ out,
                                                         int32_t* __restrict__ flags) {
  const long ea = na * F, total = (na + nb) * F;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) out[e] = e < ea ? a[e] : b[e - ea];
  if (flags != nullptr)
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < na + nb; r += (long)gridDim.x * 256)
      flags[r] = r < na ? 1 : (mask_b == nullptr || mask_b[r - na] != 0.f ? 1 : 0);
}

}  // namespace

extern "C" {

int nr_stack_rows(const int32_t* a, int rows_a, const int32_t* b, int rows_b, int F, const float* mask_b, int32_t* out, int32_t* flags,
                  nr_stream_t stream) {
  NR_CHECK_ARG(rows_a >= 0 && rows_b >= 0 && F >= 1, "stack_rows: bad sizes %d + %d rows of %d", rows_a, rows_b, F);
  if (rows_a + rows_b == 0) return NR_OK;
  NR_CHECK_ARG((rows_a == 0 || a) && (rows_b == 0 || b) && out, "stack_rows: null pointer");
  NR_DEVICE_GUARD(stream, out);
  const long total = ((long)rows_a + rows_b) * F, blocks = (total + 255) / 256;
  NrProfScope ps((hipStream_t)stream, "stack_rows[%d+%d,F=%d]", rows_a, rows_b, F);
  hipLaunchKernelGGL(stack_rows_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, a, (long)rows_a, b,
                     (long)rows_b, F, mask_b, out, flags);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_assemble_batch(const int32_t* news_combined, int n_rows, int F, const int32_t* hist_idx, const int32_t* pos_idx,
                      const int32_t* neg_idx, const int64_t* label, int B, int H, int K, int32_t* history, int32_t* candidate,
                      int32_t* bad, nr_stream_t stream) {
  NR_CHECK_ARG(B >= 0 && H >= 0 && K >= 0 && F >= 1 && n_rows >= 1, "assemble_batch: bad sizes B=%d H=%d K=%d F=%d rows=%d", B, H, K, F, n_rows);
  if (B == 0) return NR_OK;
  NR_CHECK_ARG(news_combined && pos_idx && label && candidate && (H == 0 || (hist_idx && history)) && (K == 0 || neg_idx),
               "assemble_batch: null pointer");
  NR_DEVICE_GUARD(stream, candidate);
  const long total = (long)B * (H + 1 + K) * F;
  const long blocks = (total + 255) / 256;
  NrProfScope ps((hipStream_t)stream, "assemble_batch[B=%d,H=%d,K=%d,F=%d]", B, H, K, F);
  hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)(blocks > 65536 ? 65536 : blocks)), dim3(256), 0, (hipStream_t)stream, news_combined, n_rows,
                     F, hist_idx, pos_idx, neg_idx, label, B, H, K, history, candidate, bad);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

size_t nr_eval_metrics_workspace_bytes(int n_imp) { return n_imp < 0 ? 0 : (size_t)n_imp * 4 * sizeof(double); }

int nr_eval_metrics(const float* score, const int32_t* label, const int32_t* offsets, int n_imp, int max_cand, double* per_imp,
                    size_t per_imp_bytes, double* sums, nr_stream_t stream) {
  NR_CHECK_ARG(n_imp >= 0 && sums != nullptr, "eval_metrics: bad arguments");
  NR_CHECK_ARG(max_cand >= 0 && max_cand <= MET_MAXC, "eval_metrics: an impression with %d candidates exceeds the limit of %d", max_cand, MET_MAXC);
  NR_CHECK_ARG(per_imp_bytes >= nr_eval_metrics_workspace_bytes(n_imp), "eval_metrics: per_imp holds %zu bytes, nr_eval_metrics_workspace_bytes() asks for %zu",
               per_imp_bytes, nr_eval_metrics_workspace_bytes(n_imp));
  NR_DEVICE_GUARD(stream, sums);
  hipStream_t s = (hipStream_t)stream;
  if (n_imp == 0) {
    NR_CHECK_HIP(hipMemsetAsync(sums, 0, 5 * sizeof(double), s));
    return NR_OK;
  }
  NR_CHECK_ARG(score && label && offsets && per_imp, "eval_metrics: null pointer");
  const size_t smem = (size_t)4 * MET_MAXC * (sizeof(float) + 1);
  static_assert((size_t)4 * MET_MAXC * (sizeof(float) + 1) <= 160 * 1024, "metrics LDS");
  NR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(metrics_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const int blocks = (n_imp + 3) / 4 > 4096 ? 4096 : (n_imp + 3) / 4;
  {
    NrProfScope ps(s, "eval_metrics[n=%d]", n_imp);
    hipLaunchKernelGGL(metrics_kernel, dim3(blocks), dim3(256), smem, s, score, label, offsets, n_imp, per_imp);
    hipLaunchKernelGGL(metrics_reduce_kernel, dim3(1), dim3(1024), 0, s, (const double*)per_imp, n_imp, sums);
  }
  NR_CHECK_LAUNCH();
  return NR_OK;
}

int nr_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2, float eps,
                 int step, float grad_scale, int zero_grad, nr_stream_t stream) {
  return nr_adam_step_packed(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, grad_scale, zero_grad, nullptr, 0, stream);
}

int nr_adam_step_packed(float* param, float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2, float eps,
                        int step, float grad_scale, int zero_grad, const nr_pack_job* jobs, int n_jobs, nr_stream_t stream) {
  NR_CHECK_ARG(n_jobs >= 0 && n_jobs <= NR_ADAM_PACK_MAX && (n_jobs == 0 || jobs != nullptr), "adam_step: %d pack jobs (at most %d)", n_jobs,
               NR_ADAM_PACK_MAX);
  AdamPacks pk;
  pk.n = n_jobs;
  for (int j = 0; j < n_jobs; ++j) {
    const nr_pack_job& q = jobs[j];
    NR_CHECK_ARG(q.dst != nullptr && q.cols >= 4 && q.cols % 4 == 0 && q.ld_dst >= q.cols && q.ld_dst % 4 == 0 && q.first % 4 == 0 &&
                     q.count % (size_t)q.cols == 0 && q.count < (1ull << 32) && q.first + q.count <= n && (((uintptr_t)q.dst) & 7) == 0,
                 "adam_step: pack job %d (first %zu, count %zu, cols %d, ld %d) must cover whole rows of a multiple of 4 columns inside the bucket",
                 j, q.first, q.count, q.cols, q.ld_dst);
    pk.first[j] = q.first; pk.count[j] = q.count; pk.cols[j] = q.cols; pk.ld[j] = q.ld_dst; pk.dst[j] = reinterpret_cast<bf16_t*>(q.dst);
  }
  NR_CHECK_ARG(step >= 1 && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "adam_step: bad hyper-parameters");
  if (n == 0) return NR_OK;
  NR_CHECK_ARG(param && grad && exp_avg && exp_avg_sq, "adam_step: null pointer");
  NR_CHECK_ARG((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0, "adam_step: buffers must be 16-byte aligned");
  NR_DEVICE_GUARD(stream, param);
  AdamCfg c;
  c.beta1 = beta1; c.beta2 = beta2; c.eps = eps; c.grad_scale = grad_scale; c.zero_grad = zero_grad;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  c.step_size = (float)((double)lr / bc1);
  c.bc2_sqrt = (float)sqrt(bc2);
  const size_t n4 = n / 4 + 1;
  const size_t blocks = (n4 + 255) / 256;
  NrProfScope ps((hipStream_t)stream, "adam_step[n=%zu]", n);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                     n, c, pk);
  NR_CHECK_LAUNCH();
  return NR_OK;
}

}  // extern "C"
