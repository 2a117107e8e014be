// Flat-buffer optimizer kernels: all parameters / gradients / Adam moments of the model live in
// four contiguous fp32 arenas (sized for HBM3E, one launch per step instead of one per tensor).
// reference: transformer/optimizer.py:12-75 (NoamOpt + Adam(betas=(0.9,0.98), eps=1e-9)),
// espnet2/schedulers/warmup_lr.py:10-53, trainer.py:430-467 (clip_grad_norm_, non-finite => skip),
// asr.py:228-240 (espnet1 clip + NaN guard).
#include "common.h"
#include "../../include/espnet_amd.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n,
                                                            float* __restrict__ part) {
  __shared__ float red[16];
  float s = 0.f;
  const long stride = (long)gridDim.x * blockDim.x;
  const long n4 = n / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 v = reinterpret_cast<const float4*>(g)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += g[i] * g[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// out[0] = sqrt(sum(part)), fixed order => bitwise reproducible norm
__global__ __launch_bounds__(1024) void sumsq_final_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += part[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = sqrtf(s);
}

// state (device, 8 floats): [0]=step (float count), [1]=lr, [2]=bias_corr1, [3]=bias_corr2,
//                           [4]=last grad norm, [5]=skipped-steps counter, [6]=clip coef, [7]=Adadelta eps (host-set)
// Noam / WarmupLR schedule evaluated on device so the whole training step stays capturable.
//   mode 0: lr = base           (constant)
//   mode 1: lr = factor * d^-0.5 * min(step^-0.5, step*warmup^-1.5)         (NoamOpt)
//   mode 2: lr = base * warmup^0.5 * min(step^-0.5, step*warmup^-1.5)       (WarmupLR)
__global__ void sched_kernel(float* __restrict__ st, const float* __restrict__ gnorm, int mode, float base,
                             float factor, float dmodel, float warmup, float beta1, float beta2, float max_norm) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float norm = gnorm ? gnorm[0] : 0.f;
  st[4] = norm;
  bool finite = isfinite(norm);
  if (!finite) { st[5] += 1.f; st[6] = 0.f; return; }  // reference skips optimizer.step() entirely
  float step = st[0] + 1.f;
  st[0] = step;
  float lr = base;
  if (mode == 1) lr = factor * rsqrtf(dmodel) * fminf(rsqrtf(step), step * powf(warmup, -1.5f));
  else if (mode == 2) lr = base * sqrtf(warmup) * fminf(rsqrtf(step), step * powf(warmup, -1.5f));
  st[1] = lr;
  st[2] = 1.f - powf(beta1, step);
  st[3] = 1.f - powf(beta2, step);
  float coef = 1.f;
  if (max_norm > 0.f) { coef = max_norm / (norm + 1e-6f); if (coef > 1.f) coef = 1.f; }
  st[6] = coef;
}

// torch.optim.Adam semantics (L2 weight decay added to the gradient), gradient pre-scaled by the
// clip coefficient; a zero coefficient together with a non-finite norm means "skip this step".
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   unsigned short* __restrict__ p16, long n,
                                                   const float* __restrict__ st, float beta1, float beta2,
                                                   float eps, float weight_decay) {
  const float norm = st[4];
  if (!isfinite(norm)) return;
  const float lr = st[1], bc1 = st[2], bc2 = st[3], coef = st[6];
  const float step_size = lr / bc1;
  const float inv_sqrt_bc2 = rsqrtf(bc2);
  const long stride = (long)gridDim.x * blockDim.x;
  const long n4 = n / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* pp = &pv.x; float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gg = gp[k] * coef + weight_decay * pp[k];
      mp[k] = beta1 * mp[k] + (1.f - beta1) * gg;
      vp[k] = beta2 * vp[k] + (1.f - beta2) * gg * gg;
      float denom = sqrtf(vp[k]) * inv_sqrt_bc2 + eps;
      pp[k] -= step_size * mp[k] / denom;
    }
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (p16) {   // bf16 shadow of the updated weights for the bf16-operand GEMMs (no extra read pass)
      uint2 h;
      h.x = eamd_f2bf(pv.x) | ((unsigned)eamd_f2bf(pv.y) << 16);
      h.y = eamd_f2bf(pv.z) | ((unsigned)eamd_f2bf(pv.w) << 16);
      reinterpret_cast<uint2*>(p16)[i] = h;
    }
  }
  for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float gg = g[i] * coef + weight_decay * p[i];
    m[i] = beta1 * m[i] + (1.f - beta1) * gg;
    v[i] = beta2 * v[i] + (1.f - beta2) * gg * gg;
    float denom = sqrtf(v[i]) * inv_sqrt_bc2 + eps;
    p[i] -= step_size * m[i] / denom;
    if (p16) p16[i] = eamd_f2bf(p[i]);
  }
}

// torch.optim.Adadelta semantics (the optimizer of the RNN recipes: espnet/asr/pytorch_backend/asr.py:505-508,
// rho 0.95, eps from --eps, decayed by the trainer: asr.py:798-830 multiplies param_group["eps"] by --eps-decay when the
// validation criterion stops improving).  eps lives in state[7] so that a captured step follows its decay; the
// gradient is pre-scaled by the clip coefficient state[6]; a non-finite gradient norm skips the step (asr.py:228-240).
__global__ __launch_bounds__(256) void adadelta_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ sq, float* __restrict__ acc,
                                                       unsigned short* __restrict__ p16, long n,
                                                       const float* __restrict__ st, float rho, float weight_decay) {
  if (!isfinite(st[4])) return;
  const float lr = st[1], coef = st[6], eps = st[7];
  const long stride = (long)gridDim.x * blockDim.x;
  const long n4 = n / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 sv = reinterpret_cast<float4*>(sq)[i];
    float4 av = reinterpret_cast<float4*>(acc)[i];
    float* pp = &pv.x; const float* gp = &gv.x; float* sp = &sv.x; float* ap = &av.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = gp[k] * coef + weight_decay * pp[k];
      sp[k] = sp[k] * rho + (1.f - rho) * gg * gg;
      const float delta = sqrtf(ap[k] + eps) / sqrtf(sp[k] + eps) * gg;
      ap[k] = ap[k] * rho + (1.f - rho) * delta * delta;
      pp[k] -= lr * delta;
    }
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(sq)[i] = sv;
    reinterpret_cast<float4*>(acc)[i] = av;
    if (p16) {
      uint2 h;
      h.x = eamd_f2bf(pv.x) | ((unsigned)eamd_f2bf(pv.y) << 16);
      h.y = eamd_f2bf(pv.z) | ((unsigned)eamd_f2bf(pv.w) << 16);
      reinterpret_cast<uint2*>(p16)[i] = h;
    }
  }
  for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gg = g[i] * coef + weight_decay * p[i];
    sq[i] = sq[i] * rho + (1.f - rho) * gg * gg;
    const float delta = sqrtf(acc[i] + eps) / sqrtf(sq[i] + eps) * gg;
    acc[i] = acc[i] * rho + (1.f - rho) * delta * delta;
    p[i] -= lr * delta;
    if (p16) p16[i] = eamd_f2bf(p[i]);
  }
}

// g[i] += sigma * N(0, 1): counter-based normals (Box-Muller over two 32-bit hashes of the element-pair index, the
// dropout generator's seed derivation), so that the launch is graph-replayable: the device step counter advances the
// stream of draws.  reference: espnet2/torch_utils/add_gradient_noise.py:4-31 (param.grad += sigma * randn).
__global__ __launch_bounds__(256) void grad_noise_kernel(float* __restrict__ g, long n, float sigma,
                                                         const unsigned long long* __restrict__ step,
                                                         unsigned long long salt) {
  const unsigned s0 = eamd_drop_seed(step, salt), s1 = eamd_drop_seed(step, salt ^ 0x5bd1e995ULL);
  const long stride = (long)gridDim.x * blockDim.x;
  const long np = (n + 1) / 2;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < np; q += stride) {
    const unsigned h0 = eamd_drop_pair(s0, (unsigned long long)q), h1 = eamd_drop_pair(s1, (unsigned long long)q);
    const float u0 = ((float)h0 + 1.0f) * 2.3283064365386963e-10f;      // (0, 1]
    const float u1 = (float)h1 * 2.3283064365386963e-10f;               // [0, 1)
    const float r = sigma * sqrtf(-2.0f * __logf(u0));
    float sn, cs;
    __sincosf(6.283185307179586f * u1, &sn, &cs);
    g[2 * q] += r * cs;
    if (2 * q + 1 < n) g[2 * q + 1] += r * sn;
  }
}

}  // namespace

extern "C" {

int eamd_add_gradient_noise(float* g, int64_t n, float sigma, const uint64_t* step_dev, uint64_t salt, void* stream) {
  if (!g || n <= 0 || !(sigma >= 0.f)) return EAMD_EINVAL;
  long want = ((n + 1) / 2 + 255) / 256;
  int nblk = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
  hipLaunchKernelGGL(grad_noise_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, g, (long)n, sigma,
                     (const unsigned long long*)step_dev, (unsigned long long)salt);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

/* gnorm_out[0] = ||g||_2 ; workspace >= 1024 floats */
int eamd_grad_norm(const float* g, int64_t n, float* workspace, float* gnorm_out, void* stream) {
  if (!g || !workspace || !gnorm_out || n <= 0) return EAMD_EINVAL;
  if ((uintptr_t)g & 15) return EAMD_EINVAL;
  long want = (n / 4 + 255) / 256;
  int nblk = (int)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nblk), dim3(256), 0, s, g, (long)n, workspace);
  EAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(1024), 0, s, workspace, nblk, gnorm_out);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_sched_step(float* state, const float* gnorm, int mode, float base_lr, float factor, float dmodel,
                    float warmup, float beta1, float beta2, float max_norm, void* stream) {
  if (!state || mode < 0 || mode > 2) return EAMD_EINVAL;
  hipLaunchKernelGGL(sched_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, gnorm, mode, base_lr, factor,
                     dmodel, warmup, beta1, beta2, max_norm);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, const float* state,
                   float beta1, float beta2, float eps, float weight_decay, void* stream) {
  if (!p || !g || !m || !v || !state || n <= 0) return EAMD_EINVAL;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return EAMD_EINVAL;
  long want = (n / 4 + 255) / 256;
  int nblk = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL(adam_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (unsigned short*)p_bf16,
                     (long)n, state, beta1, beta2, eps, weight_decay);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_adadelta_step(float* p, const float* g, float* square_avg, float* acc_delta, void* p_bf16, int64_t n,
                       const float* state, float rho, float weight_decay, void* stream) {
  if (!p || !g || !square_avg || !acc_delta || !state || n <= 0 || !(rho >= 0.f && rho <= 1.f)) return EAMD_EINVAL;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)square_avg | (uintptr_t)acc_delta) & 15) return EAMD_EINVAL;
  long want = (n / 4 + 255) / 256;
  int nblk = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL(adadelta_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, p, g, square_avg, acc_delta,
                     (unsigned short*)p_bf16, (long)n, state, rho, weight_decay);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}
}  // extern "C"
