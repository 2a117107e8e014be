// Recurrent building blocks of the RNN paths (VGG-BLSTM(P) encoder, attention / transducer LSTM
// decoders).  The recurrent matrix products run through eamd_gemm; these kernels are the per-step
// pointwise parts, written so that one launch covers a whole [B, H] step (lanes along H: coalesced).
// reference: torch.nn.LSTM / LSTMCell as used by rnn/encoders.py:15-162, rnn/decoders.py:88-101,120-134,
//            transducer/rnn_decoder.py:47-57,106-138  (gate order i, f, g, o).
#include "common.h"
#include "../../include/espnet_amd.h"

namespace {

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// g: [B, 4H] gate pre-activations (x W_ih^T + b_ih + h W_hh^T + b_hh)
// live[b] == 0 (packed-sequence padding): state is carried through unchanged and the output row is 0,
// which is exactly what pack_padded_sequence / pad_packed_sequence produce (encoders.py:62-71).
__global__ __launch_bounds__(256) void lstm_fwd_kernel(const float* __restrict__ g, const float* __restrict__ c_prev,
                                                       const float* __restrict__ h_prev,
                                                       const unsigned char* __restrict__ live,
                                                       float* __restrict__ h, float* __restrict__ c,
                                                       float* __restrict__ y, float* __restrict__ acts, int B,
                                                       int H) {
  const long n = (long)B * H;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const int b = idx / H, j = idx % H;
    const float* gb = g + (long)b * 4 * H;
    const float i = sigm(gb[j]), f = sigm(gb[H + j]), gg = tanhf(gb[2 * H + j]), o = sigm(gb[3 * H + j]);
    const float cp = c_prev[idx];
    float cn = f * cp + i * gg;
    float hn = o * tanhf(cn);
    float yo = hn;
    if (live && !live[b]) { cn = cp; hn = h_prev[idx]; yo = 0.f; }
    c[idx] = cn; h[idx] = hn;
    if (y) y[idx] = yo;
    float* ab = acts + (long)b * 4 * H;
    ab[j] = i; ab[H + j] = f; ab[2 * H + j] = gg; ab[3 * H + j] = o;
  }
}

// dh: gradient wrt h (sum of the output-path and the recurrent-path gradients), dc: gradient wrt c.
// Outputs dgates [B,4H] (pre-activation gradients), dc_prev, and dh_pass = the part of dh that goes
// straight to h_prev (masked rows only; 0 elsewhere).
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ dc,
                                                       const float* __restrict__ acts,
                                                       const float* __restrict__ c_prev, const float* __restrict__ c,
                                                       const unsigned char* __restrict__ live,
                                                       float* __restrict__ dgates, float* __restrict__ dc_prev,
                                                       float* __restrict__ dh_pass, int B, int H) {
  const long n = (long)B * H;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const int b = idx / H, j = idx % H;
    float* db = dgates + (long)b * 4 * H;
    const float dhv = dh[idx], dcv = dc ? dc[idx] : 0.f;
    if (live && !live[b]) {
      db[j] = 0.f; db[H + j] = 0.f; db[2 * H + j] = 0.f; db[3 * H + j] = 0.f;
      dc_prev[idx] = dcv;
      if (dh_pass) dh_pass[idx] = dhv;
      continue;
    }
    const float* ab = acts + (long)b * 4 * H;
    const float i = ab[j], f = ab[H + j], gg = ab[2 * H + j], o = ab[3 * H + j];
    const float tc = tanhf(c[idx]);
    const float dct = dcv + dhv * o * (1.f - tc * tc);
    db[j] = dct * gg * i * (1.f - i);
    db[H + j] = dct * c_prev[idx] * f * (1.f - f);
    db[2 * H + j] = dct * i * (1.f - gg * gg);
    db[3 * H + j] = dhv * tc * o * (1.f - o);
    dc_prev[idx] = dct * f;
    if (dh_pass) dh_pass[idx] = 0.f;
  }
}

// ---- 2x2 / stride-2 max pooling with ceil_mode on NHWC activations (VGG2L, encoders.py:205,208) ----
// x [B, H, W, C] -> y [B, ceil(H/2), ceil(W/2), C]; idx keeps the winning input offset (0..3) for backward
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx,
                                   int B, int H, int W, int C, int Ho, int Wo) {
  const long n = (long)B * Ho * Wo * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = i % C; long p = i / C;
    const int wo = p % Wo; p /= Wo;
    const int ho = p % Ho; const int b = p / Ho;
    float best = -INFINITY; int bi = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int hh = 2 * ho + (k >> 1), ww = 2 * wo + (k & 1);
      if (hh < H && ww < W) {
        const float v = x[(((long)b * H + hh) * W + ww) * C + c];
        if (v > best) { best = v; bi = k; }      // first maximum wins, as in ATen's max_pool2d
      }
    }
    y[i] = best; idx[i] = (unsigned char)bi;
  }
}
__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx,
                                   float* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo) {
  const long n = (long)B * H * W * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = i % C; long p = i / C;
    const int ww = p % W; p /= W;
    const int hh = p % H; const int b = p / H;
    const long o = (((long)b * Ho + (hh >> 1)) * Wo + (ww >> 1)) * C + c;
    const int k = ((hh & 1) << 1) | (ww & 1);
    dx[i] = idx[o] == k ? dy[o] : 0.f;
  }
}

inline int grid_for(long n) {
  long g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 65535 ? 65535 : g));
}

}  // namespace

extern "C" {

int eamd_lstm_cell_fwd(const float* gates, const float* c_prev, const float* h_prev, const uint8_t* live, float* h,
                       float* c, float* y, float* acts, int B, int H, void* stream) {
  if (!gates || !c_prev || !h || !c || !acts || B <= 0 || H <= 0) return EAMD_EINVAL;
  if (live && !h_prev) return EAMD_EINVAL;
  hipLaunchKernelGGL(lstm_fwd_kernel, dim3(grid_for((long)B * H)), dim3(256), 0, (hipStream_t)stream, gates, c_prev,
                     h_prev, live, h, c, y, acts, B, H);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_lstm_cell_bwd(const float* dh, const float* dc, const float* acts, const float* c_prev, const float* c,
                       const uint8_t* live, float* dgates, float* dc_prev, float* dh_pass, int B, int H,
                       void* stream) {
  if (!dh || !acts || !c_prev || !c || !dgates || !dc_prev || B <= 0 || H <= 0) return EAMD_EINVAL;
  if (live && !dh_pass) return EAMD_EINVAL;
  hipLaunchKernelGGL(lstm_bwd_kernel, dim3(grid_for((long)B * H)), dim3(256), 0, (hipStream_t)stream, dh, dc, acts,
                     c_prev, c, live, dgates, dc_prev, dh_pass, B, H);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_maxpool2x2_fwd(const float* x, float* y, uint8_t* idx, int B, int H, int W, int C, void* stream) {
  if (!x || !y || !idx || B <= 0 || H <= 0 || W <= 0 || C <= 0) return EAMD_EINVAL;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((long)B * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, x, y,
                     idx, B, H, W, C, Ho, Wo);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_maxpool2x2_bwd(const float* dy, const uint8_t* idx, float* dx, int B, int H, int W, int C, void* stream) {
  if (!dy || !idx || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0) return EAMD_EINVAL;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((long)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, dy, idx,
                     dx, B, H, W, C, Ho, Wo);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

}  // extern "C"
