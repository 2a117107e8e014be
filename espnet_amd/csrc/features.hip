// Feature-side layers that sit between the fbank batch and the encoder inside ESPnetASRModel.encode
// (SURVEY.md section 8f rank 1): SpecAugment (time warp, frequency masks, time masks) and the two feature
// normalisations.  All of it is one or two sweeps over a [B, T, F] batch (10 MB at config 2): HBM-bound, lanes
// along the feature dim.  The random draws (warp centre, mask positions / widths) are host work exactly as in the
// reference (a few torch.randint calls); the kernels take them as small device arrays.
// reference: espnet2/asr/specaug/specaug.py:19-84, espnet2/layers/time_warp.py:15-94,
//            espnet2/layers/mask_along_axis.py:7-62, espnet2/layers/global_mvn.py:62-98,
//            espnet2/layers/utterance_mvn.py:42-88, espnet2/asr/espnet_model.py:187-197.
#include "common.h"
#include "../../include/espnet_amd.h"

namespace {

inline int grid_for(long n) {
  long g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 65535 ? 65535 : g));
}

// torch.nn.functional.interpolate(mode="bicubic", align_corners=False) coefficients (A = -0.75)
__device__ __forceinline__ float cc1(float x) { return ((-0.75f + 2.f) * x - (-0.75f + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cc2(float x) { return ((-0.75f * x - 5.f * -0.75f) * x + 8.f * -0.75f) * x - 4.f * -0.75f; }

// y[b,t,f]: time-warped x (bicubic along time inside the two segments [0,center) -> [0,warped) and
// [center,len) -> [warped,len); the frequency axis keeps its size, for which the bicubic kernel is the identity),
// zero for t >= len[b] (pad_list of the per-utterance path), then zero inside any frequency / time mask.
// center[b] < 0: no warp for that utterance.  Masks: pos/len arrays [B, nmask], mask = pos <= i < pos + len.
__global__ void specaug_kernel(const float* __restrict__ x, float* __restrict__ y, const int* __restrict__ lens,
                               const int* __restrict__ center, const int* __restrict__ warped,
                               const int* __restrict__ fpos, const int* __restrict__ flen, int nf,
                               const int* __restrict__ tpos, const int* __restrict__ tlen, int nt, int B, int T, int F) {
  const long n = (long)B * T * F;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int f = i % F; long p = i / F;
    const int t = p % T; const int b = p / T;
    const int len = lens ? lens[b] : T;
    float v = 0.f;
    if (t < len) {
      const int c = center ? center[b] : -1;
      if (c < 0) {
        v = x[i];
      } else {
        const int w = warped[b];
        int in0, in_size, out_size, o;
        if (t < w) { in0 = 0; in_size = c; out_size = w; o = t; }
        else       { in0 = c; in_size = len - c; out_size = len - w; o = t - w; }
        const float scale = (float)in_size / (float)out_size;
        const float src = scale * (o + 0.5f) - 0.5f;
        const float fl = floorf(src);
        const int i0 = (int)fl;
        const float tt = src - fl;
        const float w0 = cc2(tt + 1.f), w1 = cc1(tt), w2 = cc1(1.f - tt), w3 = cc2(2.f - tt);
        const float* xb = x + ((long)b * T + in0) * F + f;
        const int hi = in_size - 1;
        const float x0 = xb[(long)min(max(i0 - 1, 0), hi) * F], x1 = xb[(long)min(max(i0, 0), hi) * F];
        const float x2 = xb[(long)min(max(i0 + 1, 0), hi) * F], x3 = xb[(long)min(max(i0 + 2, 0), hi) * F];
        v = x0 * w0 + x1 * w1 + x2 * w2 + x3 * w3;
      }
    }
    bool masked = false;
    for (int k = 0; k < nf; ++k) { const int q = fpos[b * nf + k]; masked |= (q <= f && f < q + flen[b * nf + k]); }
    for (int k = 0; k < nt; ++k) { const int q = tpos[b * nt + k]; masked |= (q <= t && t < q + tlen[b * nt + k]); }
    y[i] = masked ? 0.f : v;
  }
}

// GlobalMVN: y = ((x - mean[f]) * keep(t < len)) / std[f]   (either half optional)
__global__ void global_mvn_kernel(const float* __restrict__ x, float* __restrict__ y, const int* __restrict__ lens,
                                  const float* __restrict__ mean, const float* __restrict__ stdv, int B, int T, int F) {
  const long n = (long)B * T * F;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int f = i % F; long p = i / F;
    const int t = p % T; const int b = p / T;
    float v = x[i];
    if (mean) v -= mean[f];
    if (lens && t >= lens[b]) v = 0.f;
    if (stdv) v /= stdv[f];
    y[i] = v;
  }
}

// UtteranceMVN statistics: one workgroup per (utterance, 64 features); sum and (optionally centred) sum of squares
// over the valid frames.  mode 0: mean only; 1: mean, then sum (x-mean)^2 (two sweeps over the utterance)
__global__ __launch_bounds__(256) void utt_stats_kernel(const float* __restrict__ x, const int* __restrict__ lens,
                                                        float* __restrict__ mean, float* __restrict__ var, int T, int F,
                                                        int centred_pad) {
  __shared__ float red[4][64];
  const int b = blockIdx.x, lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int f = blockIdx.y * 64 + lane;
  const int len = lens ? lens[b] : T;
  float s = 0.f;
  if (f < F) for (int t = sub; t < len; t += 4) s += x[((long)b * T + t) * F + f];
  red[sub][lane] = s;
  __syncthreads();
  const float m = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) / (float)len;
  __syncthreads();
  if (sub == 0 && f < F) mean[(long)b * F + f] = m;
  if (!var) return;
  float q = 0.f;
  if (f < F) for (int t = sub; t < len; t += 4) { const float d = x[((long)b * T + t) * F + f] - m; q += d * d; }
  red[sub][lane] = q;
  __syncthreads();
  // utterance_mvn.py:70-72: with norm_means the sum runs over ALL T frames of x - mean, and the padding already
  // holds -mean there
  if (sub == 0 && f < F)
    var[(long)b * F + f] = (((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) +
                            (centred_pad ? (float)(T - len) * m * m : 0.f)) / (float)len;
}
// UtteranceMVN apply, following utterance_mvn.py:62-88 to the letter:
//   norm_means: x (padding zeroed) - mean everywhere (padding becomes -mean), then / sqrt(clamp(sqrt(var), eps)) if norm_vars
//   !norm_means && norm_vars: x (padding zeroed) / clamp(sqrt(var), eps)
__global__ void utt_apply_kernel(const float* __restrict__ x, float* __restrict__ y, const int* __restrict__ lens,
                                 const float* __restrict__ mean, const float* __restrict__ var, int norm_means,
                                 int norm_vars, float eps, int B, int T, int F) {
  const long n = (long)B * T * F;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int f = i % F; long p = i / F;
    const int t = p % T; const int b = p / T;
    float v = (lens && t >= lens[b]) ? 0.f : x[i];
    if (norm_means) {
      v -= mean[(long)b * F + f];
      if (norm_vars) v /= sqrtf(fmaxf(sqrtf(var[(long)b * F + f]), eps));
    } else if (norm_vars) {
      v /= fmaxf(sqrtf(var[(long)b * F + f]), eps);
    }
    y[i] = v;
  }
}

// ---- log-mel frontend ---------------------------------------------------------------------------------
// torch.stft(center=True, pad_mode="reflect") pads the [B, L] batch by n_fft/2 reflected samples on both sides
// (stft.py:82-91).  y rows have stride ldy (a multiple of the hop, so that frame t of utterance b is the GEMM
// operand row b*ldy/hop + t with leading dimension hop); the tail [L + 2 pad, ldy) is zero.
__global__ void reflect_pad_kernel(const float* __restrict__ x, long ldx, float* __restrict__ y, long ldy, int B, int L,
                                   int pad) {
  const long n = (long)B * ldy;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long b = i / ldy; const long j = i % ldy;
    float v = 0.f;
    if (j < L + 2L * pad) {
      long s = j - pad;
      if (s < 0) s = -s;
      if (s >= L) s = 2L * (L - 1) - s;
      v = x[b * ldx + s];
    }
    y[i] = v;
  }
}

// One wave per frame: |X|^2 of the F = n_fft/2+1 interleaved (re, im) bins into LDS, then lane m sums its mel
// filter's bin range [lo[m], hi[m]) (Slaney triangles are short), clamp 1e-10, log; frames >= flens[b] -> 0.
// reference: frontend/default.py:121-124 (power spectrum), log_mel.py:55-75
__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ spec, long ld, long rows_per_utt,
                                                     const float* __restrict__ melmat, const int* __restrict__ lo,
                                                     const int* __restrict__ hi, const int* __restrict__ flens,
                                                     float* __restrict__ out, int B, int T, int F, int M,
                                                     float log_scale, int power_input) {
  extern __shared__ float pw[];   // [4][F]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long fr = (long)blockIdx.x * 4 + wave;
  if (fr >= (long)B * T) return;
  const int b = fr / T, t = fr % T;
  float* o = out + fr * M;
  if (flens && t >= flens[b]) {
    for (int m = lane; m < M; m += 64) o[m] = 0.f;
    return;
  }
  const float* row1 = spec + ((long)b * rows_per_utt + t) * ld;
  const float2* row = reinterpret_cast<const float2*>(row1);
  float* p = pw + wave * F;
  if (power_input) {
    for (int f = lane; f < F; f += 64) p[f] = row1[f];
  } else {
    for (int f = lane; f < F; f += 64) { const float2 c = row[f]; p[f] = c.x * c.x + c.y * c.y; }
  }
  __builtin_amdgcn_wave_barrier();
  for (int m = lane; m < M; m += 64) {
    float acc = 0.f;
    for (int f = lo[m]; f < hi[m]; ++f) acc += p[f] * melmat[(long)f * M + m];
    o[m] = logf(fmaxf(acc, 1e-10f)) * log_scale;
  }
}

}  // namespace

extern "C" {

int eamd_specaug(const float* x, float* y, const int32_t* lens, const int32_t* center, const int32_t* warped,
                 const int32_t* fpos, const int32_t* flen, int nf, const int32_t* tpos, const int32_t* tlen, int nt, int B,
                 int T, int F, void* stream) {
  if (!x || !y || x == y || B <= 0 || T <= 0 || F <= 0 || nf < 0 || nt < 0) return EAMD_EINVAL;
  if ((center && !warped) || (nf > 0 && (!fpos || !flen)) || (nt > 0 && (!tpos || !tlen))) return EAMD_EINVAL;
  hipLaunchKernelGGL(specaug_kernel, dim3(grid_for((long)B * T * F)), dim3(256), 0, (hipStream_t)stream, x, y, lens, center,
                     warped, fpos, flen, nf, tpos, tlen, nt, B, T, F);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_global_mvn(const float* x, float* y, const int32_t* lens, const float* mean, const float* stdv, int B, int T,
                    int F, void* stream) {
  if (!x || !y || B <= 0 || T <= 0 || F <= 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(global_mvn_kernel, dim3(grid_for((long)B * T * F)), dim3(256), 0, (hipStream_t)stream, x, y, lens, mean,
                     stdv, B, T, F);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

/* workspace: 2 * B * F floats (per-utterance mean and variance) */
int eamd_utterance_mvn(const float* x, float* y, const int32_t* lens, float* workspace, int norm_means, int norm_vars,
                       float eps, int B, int T, int F, void* stream) {
  if (!x || !y || !workspace || B <= 0 || T <= 0 || F <= 0) return EAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  float* mean = workspace;
  float* var = workspace + (long)B * F;
  hipLaunchKernelGGL(utt_stats_kernel, dim3(B, (F + 63) / 64), dim3(256), 0, s, x, lens, mean, norm_vars ? var : nullptr, T, F, norm_means);
  EAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(utt_apply_kernel, dim3(grid_for((long)B * T * F)), dim3(256), 0, s, x, y, lens, mean, var, norm_means,
                     norm_vars, eps, B, T, F);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_reflect_pad(const float* x, int64_t ldx, float* y, int64_t ldy, int B, int L, int pad, void* stream) {
  if (!x || !y || B <= 0 || L <= 1 || pad < 0 || pad >= L || ldx < L || ldy < L + 2L * pad) return EAMD_EINVAL;
  hipLaunchKernelGGL(reflect_pad_kernel, dim3(grid_for((long)B * ldy)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, y,
                     (long)ldy, B, L, pad);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_logmel(const float* spec, int64_t ld, int64_t rows_per_utt, const float* melmat, const int32_t* lo,
                const int32_t* hi, const int32_t* flens, float* out, int B, int T, int F, int M, float log_scale,
                int power_input, void* stream) {
  if (!spec || !melmat || !lo || !hi || !out || B <= 0 || T <= 0 || F <= 0 || M <= 0 || rows_per_utt < T) return EAMD_EINVAL;
  if (power_input ? ld < F : (ld < 2L * F || (ld & 1) || ((uintptr_t)spec & 7))) return EAMD_EINVAL;
  const size_t sm = (size_t)4 * F * sizeof(float);
  if (sm > 64 * 1024) return EAMD_EUNSUPPORTED;
  hipLaunchKernelGGL(logmel_kernel, dim3((unsigned)(((long)B * T + 3) / 4)), dim3(256), sm, (hipStream_t)stream, spec,
                     (long)ld, (long)rows_per_utt, melmat, lo, hi, flens, out, B, T, F, M, log_scale, power_input);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

}  // extern "C"
