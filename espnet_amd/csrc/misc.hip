// Integer / layout helpers of the ASR hot path (bit-exact work, no floating point rounding):
// <sos>/<eos> insertion, greedy CTC collapse, Conv2d weight re-layout for the implicit GEMMs.
#include "common.h"
#include "../../include/espnet_amd.h"

namespace {

// reference: transformer/add_sos_eos.py:12-31 (+ nets_utils.py:34-61 pad_list).
// ys_in  = [sos, y...]  padded with eos ; ys_out = [y..., eos] padded with ignore_id ; olen = len(y)
__global__ void add_sos_eos_kernel(const long long* __restrict__ ys, long long* __restrict__ ys_in,
                                   long long* __restrict__ ys_out, int* __restrict__ olen, int B, int L, int sos,
                                   int eos, int ignore_id) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int U = L + 1;
  int n = 0;
  ys_in[(long)b * U] = sos;
  for (int i = 0; i < L; ++i) {
    long long y = ys[(long)b * L + i];
    if (y != ignore_id) {
      ys_in[(long)b * U + 1 + n] = y;
      ys_out[(long)b * U + n] = y;
      ++n;
    }
  }
  ys_out[(long)b * U + n] = eos;
  for (int i = n + 1; i < U; ++i) { ys_in[(long)b * U + i] = eos; ys_out[(long)b * U + i] = ignore_id; }
  if (olen) olen[b] = n;
}

// reference: e2e_asr_transformer.py:274-284 (argmax -> groupby -> drop blank).
// ids [B, T] int32 (per-frame argmax), hlens [B] valid frames; out [B, T] int32 padded with -1.
__global__ void ctc_collapse_kernel(const int* __restrict__ ids, const int* __restrict__ hlens,
                                    int* __restrict__ out, int* __restrict__ outlen, int B, int T, int blank) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int Tb = hlens ? min(hlens[b], T) : T;
  int n = 0, prev = -1;
  for (int t = 0; t < Tb; ++t) {
    int v = ids[(long)b * T + t];
    if (v != prev) { if (v != blank) out[(long)b * T + n++] = v; prev = v; }
  }
  for (int i = n; i < T; ++i) out[(long)b * T + i] = -1;
  outlen[b] = n;
}

// Conv2d(C,C,3,2) weight [Co][Ci][3][3] ->
//   wf [tap=kh*3+kw][Ci][Co]   (forward / weight-gradient operand, transB layout [K][N])
//   wd [q][Co][Ci]             (input-gradient operand; taps in stride-parity class order, see below)
// Class order q -> (kh,kw): (0,0)(0,2)(2,0)(2,2) | (0,1)(2,1) | (1,0)(1,2) | (1,1)
__constant__ int kTapOrder[9] = {0, 2, 6, 8, 1, 7, 3, 5, 4};
__global__ void conv2_weight_prep_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wd,
                                         int Co, int Ci, int bf16) {
  const long n = (long)9 * Co * Ci;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    // i enumerates wf: [tap][ci][co]
    int co = i % Co; long t = i / Co; int ci = t % Ci; int tap = t / Ci;
    const float v = w[((long)co * Ci + ci) * 9 + tap];
    if (bf16) reinterpret_cast<unsigned short*>(wf)[i] = eamd_f2bf(v); else wf[i] = v;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    // i enumerates wd: [q][co][ci]
    int ci = i % Ci; long t = i / Ci; int co = t % Co; int q = t / Co;
    const float v = w[((long)co * Ci + ci) * 9 + kTapOrder[q]];
    if (bf16) reinterpret_cast<unsigned short*>(wd)[i] = eamd_f2bf(v); else wd[i] = v;
  }
}
// dw[Co][Ci][3][3] += dwf[tap][Ci][Co]
__global__ void conv2_weight_grad_kernel(const float* __restrict__ dwf, float* __restrict__ dw, int Co, int Ci) {
  const long n = (long)9 * Co * Ci;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    // i enumerates dw: [co][ci][tap]
    int tap = i % 9; long t = i / 9; int ci = t % Ci; int co = t / Ci;
    dw[i] += dwf[((long)tap * Ci + ci) * Co + co];
  }
}

}  // namespace

extern "C" {

int eamd_add_sos_eos(const int64_t* ys_pad, int64_t* ys_in, int64_t* ys_out, int32_t* olen, int B, int L, int sos,
                     int eos, int ignore_id, void* stream) {
  if (!ys_pad || !ys_in || !ys_out || B <= 0 || L < 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(add_sos_eos_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                     (const long long*)ys_pad, (long long*)ys_in, (long long*)ys_out, olen, B, L, sos, eos, ignore_id);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_ctc_collapse(const int32_t* ids, const int32_t* hlens, int32_t* out, int32_t* outlen, int B, int T,
                      int blank, void* stream) {
  if (!ids || !out || !outlen || B <= 0 || T <= 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(ctc_collapse_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, ids, hlens, out,
                     outlen, B, T, blank);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_conv2_weight_prep(const float* w, void* wf, void* wd, int Co, int Ci, int out_bf16, void* stream) {
  if (!w || !wf || !wd || Co <= 0 || Ci <= 0) return EAMD_EINVAL;
  long n = (long)9 * Co * Ci;
  int g = (int)((n + 255) / 256); if (g > 2048) g = 2048;
  hipLaunchKernelGGL(conv2_weight_prep_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, w, (float*)wf, (float*)wd,
                     Co, Ci, out_bf16);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_conv2_weight_grad(const float* dwf, float* dw, int Co, int Ci, void* stream) {
  if (!dwf || !dw || Co <= 0 || Ci <= 0) return EAMD_EINVAL;
  long n = (long)9 * Co * Ci;
  int g = (int)((n + 255) / 256); if (g > 2048) g = 2048;
  hipLaunchKernelGGL(conv2_weight_grad_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, dwf, dw, Co, Ci);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

}  // extern "C"
