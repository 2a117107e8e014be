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

// dy: gradient wrt the step output y, dh: gradient wrt h arriving from the next step (recurrent path),
// dc: gradient wrt c (each may be NULL = 0).  Outputs dgates [B,4H] (pre-activation gradients), dc_prev,
// and dh_pass = the part that goes straight to h_prev (masked rows only; 0 elsewhere).
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dh,
                                                       const float* __restrict__ dc,
                                                       const float* __restrict__ acts,
                                                       const float* __restrict__ c_prev, const float* __restrict__ c,
                                                       const unsigned char* __restrict__ live,
                                                       float* __restrict__ dgates, float* __restrict__ dc_prev,
                                                       float* __restrict__ dh_pass, int B, int H) {
  const long n = (long)B * H;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const int b = idx / H, j = idx % H;
    float* db = dgates + (long)b * 4 * H;
    const float dhr = dh ? dh[idx] : 0.f, dcv = dc ? dc[idx] : 0.f;
    if (live && !live[b]) {      // y was 0 there: only the recurrent-path gradient passes through
      db[j] = 0.f; db[H + j] = 0.f; db[2 * H + j] = 0.f; db[3 * H + j] = 0.f;
      dc_prev[idx] = dcv;
      if (dh_pass) dh_pass[idx] = dhr;
      continue;
    }
    const float dhv = dhr + (dy ? dy[idx] : 0.f);
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

// ---- GRU step (torch.nn.GRU / GRUCell, gate order r, z, n) ------------------------------------------------
// gx = x W_ih^T + b_ih [B,3H], gh = h W_hh^T + b_hh [B,3H]:
//   r = sig(gx_r + gh_r), z = sig(gx_z + gh_z), n = tanh(gx_n + r * gh_n), h' = (1 - z) * n + z * h
// acts [B,4H] keeps r, z, n, gh_n for the backward.  live as in the LSTM step.
__global__ __launch_bounds__(256) void gru_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ gh,
                                                      const float* __restrict__ h_prev,
                                                      const unsigned char* __restrict__ live, float* __restrict__ h,
                                                      float* __restrict__ y, float* __restrict__ acts, int B, int H) {
  const long n_ = (long)B * H;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n_; idx += (long)gridDim.x * blockDim.x) {
    const int b = idx / H, j = idx % H;
    const float* xb = gx + (long)b * 3 * H;
    const float* hb = gh + (long)b * 3 * H;
    const float r = sigm(xb[j] + hb[j]), z = sigm(xb[H + j] + hb[H + j]);
    const float ghn = hb[2 * H + j];
    const float nn = tanhf(xb[2 * H + j] + r * ghn);
    const float hp = h_prev[idx];
    float hn = (1.f - z) * nn + z * hp;
    float yo = hn;
    if (live && !live[b]) { hn = hp; yo = 0.f; }
    h[idx] = hn;
    if (y) y[idx] = yo;
    float* ab = acts + (long)b * 4 * H;
    ab[j] = r; ab[H + j] = z; ab[2 * H + j] = nn; ab[3 * H + j] = ghn;
  }
}
// dy / dh as in lstm_bwd_kernel.  dgx, dgh [B,3H]; dh_direct [B,H] = share of the gradient that reaches h_prev
// without passing through W_hh (z * dh', or everything for a masked row)
__global__ __launch_bounds__(256) void gru_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dh,
                                                      const float* __restrict__ acts, const float* __restrict__ h_prev,
                                                      const unsigned char* __restrict__ live, float* __restrict__ dgx,
                                                      float* __restrict__ dgh, float* __restrict__ dh_direct, int B,
                                                      int H) {
  const long n_ = (long)B * H;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n_; idx += (long)gridDim.x * blockDim.x) {
    const int b = idx / H, j = idx % H;
    float* xb = dgx + (long)b * 3 * H;
    float* hb = dgh + (long)b * 3 * H;
    const float dhr = dh ? dh[idx] : 0.f;
    if (live && !live[b]) {
      xb[j] = 0.f; xb[H + j] = 0.f; xb[2 * H + j] = 0.f; hb[j] = 0.f; hb[H + j] = 0.f; hb[2 * H + j] = 0.f;
      dh_direct[idx] = dhr;
      continue;
    }
    const float d = dhr + (dy ? dy[idx] : 0.f);
    const float* ab = acts + (long)b * 4 * H;
    const float r = ab[j], z = ab[H + j], nn = ab[2 * H + j], ghn = ab[3 * H + j];
    const float dn_pre = d * (1.f - z) * (1.f - nn * nn);
    const float dz_pre = d * (h_prev[idx] - nn) * z * (1.f - z);
    const float dr_pre = dn_pre * ghn * r * (1.f - r);
    xb[j] = dr_pre; xb[H + j] = dz_pre; xb[2 * H + j] = dn_pre;
    hb[j] = dr_pre; hb[H + j] = dz_pre; hb[2 * H + j] = dn_pre * r;
    dh_direct[idx] = d * z;
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

__device__ __forceinline__ float row16_sum(float v) {     // sum over the 16 lanes of a DPP row (no LDS crossbar trip)
  v += eamd_dpp<0xB1>(v);
  v += eamd_dpp<0x4E>(v);
  v += eamd_dpp<0x141>(v);
  v += eamd_dpp<0x140>(v);
  return v;
}

// ---- location-aware attention (AttLoc), one decoder step -----------------------------------------
// reference: rnn/attentions.py:300-380
//   conv[b,t,c]  = sum_k conv_w[c,k] * att_prev[b, t + k - F]                 (Conv2d(1,C,(1,2F+1)), no bias)
//   e[b,t]       = gvec . tanh(W_att conv[b,t] + pre_enc[b,t] + dec_proj[b]) + gb ; -inf for t >= len_b
//   w[b,:]       = softmax(scaling * e[b,:]) ;  ctx[b,:] = sum_t w[b,t] * enc_h[b,t,:]
// one workgroup per (b,t); the C location features live in LDS, threads run along the attention dim
__global__ __launch_bounds__(128) void attloc_energy_fwd_kernel(
    const float* __restrict__ att_prev, const float* __restrict__ conv_w, const float* __restrict__ w_att,
    const float* __restrict__ pre_enc, const float* __restrict__ dec_proj, const float* __restrict__ gvec,
    const float* __restrict__ gb, const int* __restrict__ lens, float* __restrict__ e, float* __restrict__ th,
    float* __restrict__ conv, int T, int A, int C, int K, int R) {
  __shared__ float convl[64];
  __shared__ float red[16];
  const int bt = blockIdx.x, b = bt / T, t = bt % T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int F = (K - 1) / 2;
  const float* pv = att_prev + (long)b * R * T;      // R rows of history (AttLoc2D window), R = 1: the previous weights
  for (int c = wave; c < C; c += nw) {
    float sacc = 0.f;
    for (int q = lane; q < R * K; q += 64) {
      const int r = q / K, k = q % K;
      const int ts = t + k - F;
      if (ts >= 0 && ts < T) sacc += conv_w[((long)c * R + r) * K + k] * pv[(long)r * T + ts];
    }
    sacc = wave_sum(sacc);
    if (lane == 0) { convl[c] = sacc; conv[(long)bt * C + c] = sacc; }
  }
  __syncthreads();
  float acc = 0.f;
  for (int a = threadIdx.x; a < A; a += blockDim.x) {
    float f = pre_enc[(long)bt * A + a] + dec_proj[(long)b * A + a];
    for (int c = 0; c < C; ++c) f += w_att[a * C + c] * convl[c];
    const float v = tanhf(f);
    th[(long)bt * A + a] = v;
    acc += gvec[a] * v;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) e[bt] = t < lens[b] ? acc + gb[0] : -INFINITY;
}

// The same for A <= 1024, A % 4 == 0, C <= 16, one row of history (R = 1), K <= 255 taps: a workgroup owns eight frames of one utterance and a thread four adjacent
// attention units, so mlp_att's weight rows, dec_proj and gvec are read once per eight frames into registers (the
// per-frame kernel above re-reads the [A,C] weight with a 4C-byte lane stride for every frame: 53 us per step at
// config 4, against 16 us for its 65 MB of pre_enc / th traffic).
constexpr int ATTE_TCH = 8, ATTE_MAXK = 255;
template <int CT>
__global__ __launch_bounds__(256) void attloc_energy_fwd8_kernel(
    const float* __restrict__ att_prev, const float* __restrict__ conv_w, const float* __restrict__ w_att,
    const float* __restrict__ pre_enc, const float* __restrict__ dec_proj, const float* __restrict__ gvec,
    const float* __restrict__ gb, const int* __restrict__ lens, float* __restrict__ e, float* __restrict__ th,
    float* __restrict__ conv, int T, int A, int C, int K, int R) {
  __shared__ float convl[ATTE_TCH][CT];
  __shared__ float red[ATTE_TCH][16];
  __shared__ float wl[CT * ATTE_MAXK];
  __shared__ float win[ATTE_TCH + ATTE_MAXK];
  const int b = blockIdx.x, t0 = blockIdx.y * ATTE_TCH, nt = min(ATTE_TCH, T - t0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int F = (K - 1) / 2;
  // location features of the eight frames (R = 1): filter bank and the window of previous weights through LDS, one thread
  // per (frame, channel) - as one wave per output the 80 short reductions of a workgroup ran back to back (30 us)
  const float* pv = att_prev + (long)b * T;
  for (int i = tid; i < C * K; i += 256) wl[i] = conv_w[i];
  for (int i = tid; i < nt + K - 1; i += 256) {
    const int ts = t0 - F + i;
    win[i] = (ts >= 0 && ts < T) ? pv[ts] : 0.f;
  }
  __syncthreads();
  if (tid < nt * CT) {
    const int tl = tid / CT, c = tid - tl * CT;
    float sacc = 0.f;
    if (c < C) {
      const float* wr = wl + c * K;
      const float* xr = win + tl;
      for (int k = 0; k < K; ++k) sacc += wr[k] * xr[k];
      conv[((long)b * T + t0 + tl) * C + c] = sacc;
    }
    convl[tl][c] = sacc;
  }
  const int a = 4 * tid;
  const bool live = a < A;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float W[CT][4];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int k = 0; k < 4; ++k) W[c][k] = (live && c < C) ? w_att[(long)(a + k) * C + min(c, C - 1)] : 0.f;
  const float4 dp = live ? *reinterpret_cast<const float4*>(dec_proj + (long)b * A + a) : z4;
  const float4 g = live ? *reinterpret_cast<const float4*>(gvec + a) : z4;
  float4 pe[ATTE_TCH];
#pragma unroll
  for (int u = 0; u < ATTE_TCH; ++u)
    pe[u] = (live && u < nt) ? *reinterpret_cast<const float4*>(pre_enc + ((long)b * T + t0 + u) * A + a) : z4;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < ATTE_TCH; ++u) {
    float4 f = make_float4(pe[u].x + dp.x, pe[u].y + dp.y, pe[u].z + dp.z, pe[u].w + dp.w);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const float cv = convl[u < nt ? u : 0][c];
      f.x += W[c][0] * cv; f.y += W[c][1] * cv; f.z += W[c][2] * cv; f.w += W[c][3] * cv;
    }
    const float4 v = make_float4(tanhf(f.x), tanhf(f.y), tanhf(f.z), tanhf(f.w));
    if (live && u < nt) *reinterpret_cast<float4*>(th + ((long)b * T + t0 + u) * A + a) = v;
    const float s = row16_sum(live ? g.x * v.x + g.y * v.y + g.z * v.z + g.w * v.w : 0.f);
    if ((lane & 15) == 0) red[u][tid >> 4] = s;
  }
  __syncthreads();
  if (tid < nt) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += red[tid][r];
    e[(long)b * T + t0 + tid] = (t0 + tid) < lens[b] ? s + gb[0] : -INFINITY;
  }
}

// grid (B, ceil(E/64)); 256 threads = 4 frame-subgroups x 64 feature lanes; dynamic LDS: T floats
__global__ __launch_bounds__(256) void attloc_ctx_fwd_kernel(const float* __restrict__ e,
                                                             const float* __restrict__ enc_h, float scaling,
                                                             float* __restrict__ w, float* __restrict__ ctx, int T,
                                                             int E) {
  extern __shared__ float wl[];
  __shared__ float red[16];
  __shared__ float part[4][64];
  const int b = blockIdx.x;
  const float* eb = e + (long)b * T;
  float m = -INFINITY;
  for (int t = threadIdx.x; t < T; t += blockDim.x) m = fmaxf(m, scaling * eb[t]);
  m = block_max(m, red);
  float sum = 0.f;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const float v = expf(scaling * eb[t] - m);
    wl[t] = v; sum += v;
  }
  sum = block_sum(sum, red);     // (barriers inside also publish wl)
  const float inv = 1.f / sum;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const float v = wl[t] * inv;
    wl[t] = v;
    if (blockIdx.y == 0) w[(long)b * T + t] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + lane;
  float acc = 0.f;
  if (col < E)
    for (int t = sub; t < T; t += 4) acc += wl[t] * enc_h[((long)b * T + t) * E + col];
  part[sub][lane] = acc;
  __syncthreads();
  if (sub == 0 && col < E) ctx[(long)b * E + col] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// backward of softmax + context, two launches sized for the machine (one workgroup per utterance starved it):
//  (1) one wave per frame (b,t): dw[b,t] = dw_ext[b,t] + enc_h[b,t,:] . dctx[b,:]  and  d_enc_h[b,t,:] = w[b,t] * dctx[b,:]
//  (2) one workgroup per utterance: de[b,t] = scaling * w * (dw - sum_t w dw) in place over dw; dgb += sum_t de
__global__ __launch_bounds__(256) void attloc_ctx_bwd_rows_kernel(const float* __restrict__ dctx,
                                                                  const float* __restrict__ dw_ext,
                                                                  const float* __restrict__ w,
                                                                  const float* __restrict__ enc_h,
                                                                  float* __restrict__ dwv, float* __restrict__ d_enc_h,
                                                                  int BT, int T, int E, int accum = 0) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= BT) return;
  const int b = row / T;
  const float* dc = dctx + (long)b * E;
  const float* h = enc_h + (long)row * E;
  float* o = d_enc_h + (long)row * E;
  const float wv = w[row];
  float d = 0.f;
  for (int x = lane; x < E; x += 64) {
    const float c = dc[x];
    d += c * h[x];
    o[x] = accum ? o[x] + wv * c : wv * c;       // accum: the running sum over the decoder steps (see eamd_attloc_bwd_energy_conv)
  }
  d = wave_sum(d);
  if (lane == 0) dwv[row] = d + (dw_ext ? dw_ext[row] : 0.f);
}
__global__ __launch_bounds__(256) void attloc_softmax_bwd_kernel(const float* __restrict__ w, float scaling,
                                                                 float* __restrict__ de, float* __restrict__ dgb, int T) {
  __shared__ float red[16];
  const int b = blockIdx.x;
  float s = 0.f;
  for (int t = threadIdx.x; t < T; t += blockDim.x) s += w[(long)b * T + t] * de[(long)b * T + t];
  s = block_sum(s, red);
  float tot = 0.f;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const float v = scaling * w[(long)b * T + t] * (de[(long)b * T + t] - s);
    de[(long)b * T + t] = v;
    tot += v;
  }
  tot = block_sum(tot, red);
  if (threadIdx.x == 0) atomicAdd(dgb, tot);
}

// df[b,t,a] = de[b,t] * gvec[a] * (1 - th^2); dgvec[a] += sum de*th; d_dec_proj[b,a] += sum_t df
// grid (B, ceil(T/TCH)), threads along a
constexpr int ATT_TCH = 8;
__global__ __launch_bounds__(256) void attloc_energy_bwd_kernel(const float* __restrict__ de,
                                                                const float* __restrict__ th,
                                                                const float* __restrict__ gvec, float* __restrict__ df,
                                                                float* __restrict__ dgvec, float* __restrict__ ddec,
                                                                int T, int A) {
  const int b = blockIdx.x;
  const int t0 = blockIdx.y * ATT_TCH, t1 = min(T, t0 + ATT_TCH);
  for (int a = threadIdx.x; a < A; a += blockDim.x) {
    const float g = gvec[a];
    float pg = 0.f, pd = 0.f;
    for (int t = t0; t < t1; ++t) {
      const long i = ((long)b * T + t) * A + a;
      const float d = de[(long)b * T + t], v = th[i];
      const float f = d * g * (1.f - v * v);
      df[i] = f;
      pg += d * v; pd += f;
    }
    atomicAdd(&dgvec[a], pg);
    atomicAdd(&ddec[(long)b * A + a], pd);
  }
}

// The energy backward with the two products over W_att folded in (location-aware attention, one decoder step):
//   df[b,t,a] = de[b,t] * gvec[a] * (1 - th^2)            dconv[b,t,c] = sum_a df[b,t,a] * W_att[a,c]
//   dW_att[a,c] += sum_{b,t} df[b,t,a] * conv[b,t,c]      dgvec[a] += sum de * th      d_dec_proj[b,a] += sum_t df
// As GEMMs the two products have N = C = 10 columns (56 + 63 us per step at config 4, each re-reading the 33 MB df);
// here a workgroup owns (b, 32 frames), a thread 4 adjacent a: df is formed once, every reduction over (b, t) is kept
// in registers and leaves as a per-workgroup partial that attloc_bwd_reduce_kernel sums (no atomics), the reduction
// over a for dconv goes through the waves and LDS.  reference: autograd of rnn/attentions.py:329-365 (AttLoc.forward).
constexpr int ATTF_TCH = 32, ATTF_MAXC = 16;
template <int CT>       // channels the code is unrolled for: C <= CT, the excess ones carry zeros
__global__ __launch_bounds__(256) void attloc_energy_bwd_fused_kernel(const float* __restrict__ de,
                                                                      const float* __restrict__ th,
                                                                      const float* __restrict__ gvec,
                                                                      const float* __restrict__ conv,
                                                                      const float* __restrict__ w_att,
                                                                      float* __restrict__ df, float* __restrict__ dconv,
                                                                      float* __restrict__ part, int T, int A, int C, int accum) {
  __shared__ float red[ATTF_TCH][16][ATTF_MAXC];         // [frame][16-lane row of the workgroup][channel]
  const int b = blockIdx.x, ch = blockIdx.y;
  const int t0 = ch * ATTF_TCH, t1 = min(T, t0 + ATTF_TCH);
  const int tid = threadIdx.x, lane = tid & 63;
  const int a = 4 * tid;
  const bool live = a < A;                               // A % 4 == 0, A <= 1024 (host-checked)
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 g = live ? *reinterpret_cast<const float4*>(gvec + a) : z4;
  float W[CT][4], acc[CT][4];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      W[c][k] = (live && c < C) ? w_att[(long)(a + k) * C + min(c, C - 1)] : 0.f;
      acc[c][k] = 0.f;
    }
  float4 pg = z4, pd = z4;
  // frames in groups of four, the next group's loads in flight under the arithmetic of this one (one wave per SIMD:
  // nothing else hides the HBM round trip)
  float4 vb[4];
  float db[4];
  auto load_group = [&](int tg) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = tg + u;
      const long row = (long)b * T + min(t, t1 - 1);
      db[u] = t < t1 ? de[row] : 0.f;
      vb[u] = live ? *reinterpret_cast<const float4*>(th + row * A + a) : z4;
    }
  };
  load_group(t0);
  for (int tg = t0; tg < t1; tg += 4) {
    float4 vc[4];
    float dc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { vc[u] = vb[u]; dc[u] = db[u]; }
    if (tg + 4 < t1) load_group(tg + 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = min(tg + u, t1 - 1);                   // a frame past the chunk repeats the last one with d = 0
      const long row = (long)b * T + t;
      const float d = dc[u];
      const float4 v = vc[u];
      const float* cvr = conv + row * C;                   // workgroup-uniform: scalar loads, all CT requested together
      float cv[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) cv[c] = cvr[CT == ATTF_MAXC ? min(c, C - 1) : c];
      float4 f;
      f.x = d * g.x * (1.f - v.x * v.x); f.y = d * g.y * (1.f - v.y * v.y);
      f.z = d * g.z * (1.f - v.z * v.z); f.w = d * g.w * (1.f - v.w * v.w);
      if (live && tg + u < t1) {
        float4* dst = reinterpret_cast<float4*>(df + row * A + a);
        if (accum) { const float4 o = *dst; *dst = make_float4(o.x + f.x, o.y + f.y, o.z + f.z, o.w + f.w); }
        else *dst = f;
      }
      pg.x += d * v.x; pg.y += d * v.y; pg.z += d * v.z; pg.w += d * v.w;
      pd.x += f.x; pd.y += f.y; pd.z += f.z; pd.w += f.w;
      float sc[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        acc[c][0] += f.x * cv[c]; acc[c][1] += f.y * cv[c]; acc[c][2] += f.z * cv[c]; acc[c][3] += f.w * cv[c];
        sc[c] = row16_sum(f.x * W[c][0] + f.y * W[c][1] + f.z * W[c][2] + f.w * W[c][3]);
      }
      if ((lane & 15) == 0 && tg + u < t1) {
#pragma unroll
        for (int c = 0; c < CT; ++c) red[t - t0][tid >> 4][c] = sc[c];
      }
    }
  }
  __syncthreads();
  for (int q = tid; q < (t1 - t0) * C; q += 256) {
    const int tl = q / C, c = q - tl * C;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += red[tl][r][c];
    dconv[((long)b * T + t0 + tl) * C + c] = s;
  }
  if (live) {
    float* ps = part + ((long)b * gridDim.y + ch) * A * (C + 2);
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < C) {
#pragma unroll
        for (int k = 0; k < 4; ++k) ps[(long)(a + k) * C + c] = acc[c][k];
      }
    *reinterpret_cast<float4*>(ps + (long)A * C + a) = pg;
    *reinterpret_cast<float4*>(ps + (long)A * (C + 1) + a) = pd;
  }
}
// dW_att / dgvec += the sum of the workgroup partials over all (b, chunk); d_dec_proj[b,:] += the sum over b's chunks.
// A workgroup owns 64 adjacent outputs; its four waves split the slabs (eight loads in flight each) and meet in LDS.
__global__ __launch_bounds__(256) void attloc_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw_att,
                                                                float* __restrict__ dgvec, float* __restrict__ ddec, int B,
                                                                int NCH, int A, int C) {
  __shared__ float red[4][64];
  const long stride = (long)A * (C + 2);
  const long n1 = (long)A * (C + 1), n2 = (long)B * A;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + lane;
  float s = 0.f;
  if (i < n1) {
    const int ns = B * NCH;
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int sl = wave;
    for (; sl + 28 < ns; sl += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s8[u] += part[(long)(sl + 4 * u) * stride + i];
    }
    for (; sl < ns; sl += 4) s8[0] += part[(long)sl * stride + i];
    s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  } else if (i < n1 + n2) {
    const long j = i - n1;
    const int b = (int)(j / A), a = (int)(j - (long)b * A);
    for (int chn = wave; chn < NCH; chn += 4) s += part[((long)b * NCH + chn) * stride + n1 + a];
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0) {
    s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (i < (long)A * C) dw_att[i] += s;
    else if (i < n1) dgvec[i - (long)A * C] += s;
    else if (i < n1 + n2) ddec[i - n1] += s;
  }
}
// d_prev[b,r,s] = sum_{c,k} dconv[b, s - k + F, c] * conv_w[c,r,k]      one wave per output (b,r,s): lanes over k
__global__ __launch_bounds__(256) void attloc_conv_bwd_prev_kernel(const float* __restrict__ dconv,
                                                                   const float* __restrict__ conv_w,
                                                                   float* __restrict__ d_prev, int BRT, int T, int C,
                                                                   int K, int R) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= BRT) return;
  const int s = row % T, r = (row / T) % R, b = row / (T * R);
  const int F = (K - 1) / 2;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) {
    const int t = s - k + F;
    if (t < 0 || t >= T) continue;
    const float* dcv = dconv + ((long)b * T + t) * C;
    for (int c = 0; c < C; ++c) acc += dcv[c] * conv_w[((long)c * R + r) * K + k];
  }
  acc = wave_sum(acc);
  if (lane == 0) d_prev[row] = acc;
}
// dconv_w[c,r,k] += sum_{b,t} dconv[b,t,c] * att_prev[b, r, t + k - F]       grid (C, B), threads along (r, k)
__global__ __launch_bounds__(256) void attloc_conv_bwd_w_kernel(const float* __restrict__ dconv,
                                                                const float* __restrict__ att_prev,
                                                                float* __restrict__ dconv_w, int T, int C, int K, int R) {
  const int c = blockIdx.x, b = blockIdx.y;
  const int F = (K - 1) / 2;
  for (int q = threadIdx.x; q < R * K; q += blockDim.x) {
    const int r = q / K, k = q % K;
    // frames t with 0 <= t + k - F < T, eight independent products per trip (the loop is a chain of L2 round trips otherwise)
    const int ta = max(0, F - k), tb = min(T, T + F - k);
    const float* dcv = dconv + (long)b * T * C + c;
    const float* pv = att_prev + ((long)b * R + r) * T + (k - F);
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int t = ta;
    for (; t + 8 <= tb; t += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a8[u] += dcv[(long)(t + u) * C] * pv[t + u];
    }
    for (; t < tb; ++t) a8[0] += dcv[(long)t * C] * pv[t];
    const float acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    atomicAdd(&dconv_w[((long)c * R + r) * K + k], acc);
  }
}

// ---- AttLocRec front end (attentions.py:690-696): pooled[b,c] = max_t relu(conv(att_prev)[b,c,t]), idx = its frame
// one wave per (b,c): lanes over frames, the K-tap window per frame read from global (att_prev rows are tiny)
__global__ __launch_bounds__(256) void attloc_convmax_fwd_kernel(const float* __restrict__ att_prev,
                                                                 const float* __restrict__ conv_w,
                                                                 float* __restrict__ pooled, int* __restrict__ idx, int BC,
                                                                 int T, int C, int K) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= BC) return;
  const int b = row / C, c = row % C;
  const int F = (K - 1) / 2;
  const float* pv = att_prev + (long)b * T;
  float best = -INFINITY; int bi = 0;
  for (int t = lane; t < T; t += 64) {
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
      const int ts = t + k - F;
      if (ts >= 0 && ts < T) acc += conv_w[c * K + k] * pv[ts];
    }
    if (acc > best) { best = acc; bi = t; }
  }
  // wave arg-max, ties to the earliest frame (what max_pool2d's backward picks)
  for (int off = 32; off > 0; off >>= 1) {
    const float ob = __shfl_xor(best, off); const int oi = __shfl_xor(bi, off);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) { pooled[row] = fmaxf(best, 0.f); idx[row] = bi; }
}
// backward: only the arg-max frame of each (b,c) carries gradient, and only where the ReLU was active
// d_prev[b, idx+k-F] += dpool[b,c] * conv_w[c,k] ; dconv_w[c,k] += dpool[b,c] * att_prev[b, idx+k-F]   thread = (b,c,k)
__global__ void attloc_convmax_bwd_kernel(const float* __restrict__ dpool, const float* __restrict__ pooled,
                                          const int* __restrict__ idx, const float* __restrict__ att_prev,
                                          const float* __restrict__ conv_w, float* __restrict__ d_prev,
                                          float* __restrict__ dconv_w, int B, int T, int C, int K) {
  const long n = (long)B * C * K;
  const int F = (K - 1) / 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int k = i % K; const long bc = i / K; const int c = bc % C; const int b = bc / C;
    if (pooled[bc] <= 0.f) continue;
    const int ts = idx[bc] + k - F;
    if (ts < 0 || ts >= T) continue;
    const float g = dpool[bc];
    atomicAdd(&d_prev[(long)b * T + ts], g * conv_w[c * K + k]);
    atomicAdd(&dconv_w[c * K + k], g * att_prev[(long)b * T + ts]);
  }
}

// ---- dot-product attention energies (AttDot :91-164, AttMultiHeadDot :845-990) ---------------------------
// e[b,t] = sum_a k[b,t,a] * q[b,a]  (k = tanh(mlp_k h), q = tanh(mlp_q z), both already activated); -inf for t >= len
__global__ __launch_bounds__(64) void att_dot_energy_fwd_kernel(const float* __restrict__ k, const float* __restrict__ q,
                                                               const int* __restrict__ lens, float* __restrict__ e,
                                                               int T, int A) {
  const int bt = blockIdx.x, b = bt / T, t = bt % T;
  float acc = 0.f;
  for (int a = threadIdx.x; a < A; a += 64) acc += k[(long)bt * A + a] * q[(long)b * A + a];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) e[bt] = t < lens[b] ? acc : -INFINITY;
}
// dk[b,t,a] = de[b,t] * q[b,a];  dq[b,a] = sum_t de[b,t] * k[b,t,a]      grid (B, ceil(T/TCH)), threads along a
__global__ __launch_bounds__(256) void att_dot_energy_bwd_kernel(const float* __restrict__ de, const float* __restrict__ k,
                                                                 const float* __restrict__ q, float* __restrict__ dk,
                                                                 float* __restrict__ dq, int T, int A) {
  const int b = blockIdx.x;
  const int t0 = blockIdx.y * ATT_TCH, t1 = min(T, t0 + ATT_TCH);
  for (int a = threadIdx.x; a < A; a += blockDim.x) {
    const float qv = q[(long)b * A + a];
    float acc = 0.f;
    for (int t = t0; t < t1; ++t) {
      const long i = ((long)b * T + t) * A + a;
      const float d = de[(long)b * T + t];
      dk[i] = d * qv;
      acc += d * k[i];
    }
    atomicAdd(&dq[(long)b * A + a], acc);
  }
}

inline int grid_for(long n) {
  long g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 65535 ? 65535 : g));
}


// ---- one LSTM time step as ONE launch --------------------------------------------------------------------------------------
// gates = gx_t + b_hh + h_prev W_hh^T and the cell update, for all B utterances.  A workgroup owns U = 4 hidden units = the
// 16 gate rows (i, f, g, o of each): a full 16-column MFMA tile.  Its 4 waves split the reduction over the H inputs; both
// operands are read straight from global memory / L2 as 16-byte fragments (v_mfma_f32_16x16x4_f32: lane (row = lane % 16,
// quad = lane / 16) holds 4 consecutive k of its row; MFMA e of a quad-step contracts k = k0 + 4*quad + e on both sides),
// the four partial tiles are summed through LDS and 32 x 4 threads finish the cell.  Compared with the split-K GEMM
// (zero fill + 256 atomically accumulated tiles) + cell kernel it replaces: no atomics, no [B,4H] round trip, 1 launch
// instead of 3; the kernel boundary is the only inter-step synchronisation (a persistent kernel would pay a grid
// barrier of several microseconds per step for the same all-to-all exchange of h).
// reference: torch.nn.LSTM / LSTMCell steps inside rnn/encoders.py:36-117, rnn/decoders.py:120-134.
constexpr int LS_U = 4;           // hidden units per workgroup in the forward step
template <int MT>                 // batch tiles of 16 rows (B <= 16 * MT)
__global__ __launch_bounds__(1024) void lstm_step_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ w_hh,
                                                            const float* __restrict__ b_hh, const float* __restrict__ h_prev,
                                                            const float* __restrict__ c_prev,
                                                            const unsigned char* __restrict__ live, float* __restrict__ h,
                                                            float* __restrict__ c, float* __restrict__ y,
                                                            float* __restrict__ acts, int B, int H) {
  // the reduction is split over the workgroup's NW waves (up to 16: the step is latency-bound, what counts is how
  // many 16-byte loads the CU has in flight); part[NW][MT*16][17]
  extern __shared__ float part_raw[];
  float (*part)[MT * 16][17] = reinterpret_cast<float (*)[MT * 16][17]>(part_raw);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fr = lane & 15, fq = lane >> 4;
  const int NW = blockDim.x >> 6;
  const int u0 = blockIdx.x * LS_U;
  // B-operand row n = gate (n / 4) of unit u0 + n % 4
  const float* wrow = w_hh + ((long)(fr >> 2) * H + u0 + (fr & 3)) * H;
  const int kw = H / NW, k_begin = wave * kw;
  f32x4 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* arow[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) arow[i] = h_prev + (long)min(i * 16 + fr, B - 1) * H;
  for (int k0 = k_begin; k0 < k_begin + kw; k0 += 64) {       // 4 quad-steps of 16 k in flight
    float4 bv[4], av[MT][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = min(k0 + q * 16, k_begin + kw - 16) + fq * 4;
      bv[q] = *reinterpret_cast<const float4*>(wrow + k);
#pragma unroll
      for (int i = 0; i < MT; ++i) av[i][q] = *reinterpret_cast<const float4*>(arow[i] + k);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (k0 + q * 16 < k_begin + kw) {        // wave-uniform (the clamped tail steps re-read the last valid one)
        const float b4[4] = {bv[q].x, bv[q].y, bv[q].z, bv[q].w};
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const float a4[4] = {av[i][q].x, av[i][q].y, av[i][q].z, av[i][q].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], b4[e], acc[i], 0, 0, 0);
        }
      }
    }
  }
  // accumulator layout: lane (fr = column n, fq) holds rows 4*fq + r of the tile
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][i * 16 + fq * 4 + r][fr] = acc[i][r];
  __syncthreads();
  for (int p = t; p < B * LS_U; p += blockDim.x) {
    const int b = p / LS_U, j = p % LS_U, u = u0 + j;
    float g4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = g * 4 + j;
      float sacc = 0.f;
      for (int w = 0; w < NW; ++w) sacc += part[w][b][n];
      g4[g] = sacc + gx[(long)b * 4 * H + g * H + u] + (b_hh ? b_hh[g * H + u] : 0.f);
    }
    const float ig = sigm(g4[0]), fg = sigm(g4[1]), gg = tanhf(g4[2]), og = sigm(g4[3]);
    const long idx = (long)b * H + u;
    const float cp = c_prev[idx];
    float cn = fg * cp + ig * gg;
    float hn = og * tanhf(cn);
    float yo = hn;
    if (live && !live[b]) { cn = cp; hn = h_prev[idx]; yo = 0.f; }
    c[idx] = cn; h[idx] = hn;
    if (y) y[idx] = yo;
    float* ab = acts + (long)b * 4 * H;
    ab[u] = ig; ab[H + u] = fg; ab[2 * H + u] = gg; ab[3 * H + u] = og;
  }
}

// Backward twin.  A workgroup owns U = 16 hidden units (a full MFMA tile of the recurrent input gradient):
//   dh[b, u] = dh_pass[b, u] + sum_r dgates_next[b, r] * W_hh[r, u]      (r over the 4H gate rows; w_t = W_hh^T [H, 4H])
// with the reduction split over its 4 waves, then the cell backward of its units writes dgates[b, {i,f,g,o}, u], dc_prev
// and the masked pass-through.  dgates_next == NULL: the first backward step (no recurrent gradient yet).
constexpr int LB_U = 16;
template <int MT>
__global__ __launch_bounds__(1024) void lstm_step_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dgates_next,
                                                            const float* __restrict__ w_t, const float* __restrict__ dh_pass_in,
                                                            const float* __restrict__ dc, const float* __restrict__ acts,
                                                            const float* __restrict__ c_prev, const float* __restrict__ c,
                                                            const unsigned char* __restrict__ live,
                                                            float* __restrict__ dgates, float* __restrict__ dc_prev,
                                                            float* __restrict__ dh_pass, int B, int H) {
  extern __shared__ float part_raw[];
  float (*part)[MT * 16][17] = reinterpret_cast<float (*)[MT * 16][17]>(part_raw);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fr = lane & 15, fq = lane >> 4;
  const int NW = blockDim.x >> 6;
  const int u0 = blockIdx.x * LB_U;
  const int K = 4 * H;
  if (dgates_next) {
    const float* wrow = w_t + (long)(u0 + fr) * K;
    const int kw = K / NW, k_begin = wave * kw;
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* arow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) arow[i] = dgates_next + (long)min(i * 16 + fr, B - 1) * K;
    for (int k0 = k_begin; k0 < k_begin + kw; k0 += 64) {
      float4 bv[4], av[MT][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = min(k0 + q * 16, k_begin + kw - 16) + fq * 4;
        bv[q] = *reinterpret_cast<const float4*>(wrow + k);
#pragma unroll
        for (int i = 0; i < MT; ++i) av[i][q] = *reinterpret_cast<const float4*>(arow[i] + k);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (k0 + q * 16 < k_begin + kw) {
          const float b4[4] = {bv[q].x, bv[q].y, bv[q].z, bv[q].w};
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const float a4[4] = {av[i][q].x, av[i][q].y, av[i][q].z, av[i][q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], b4[e], acc[i], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[wave][i * 16 + fq * 4 + r][fr] = acc[i][r];
  }
  __syncthreads();
  for (int p = t; p < B * LB_U; p += blockDim.x) {
    const int b = p / LB_U, j = p % LB_U, u = u0 + j;
    const long idx = (long)b * H + u;
    float dhr = dh_pass_in ? dh_pass_in[idx] : 0.f;
    if (dgates_next)
      for (int w = 0; w < NW; ++w) dhr += part[w][b][j];
    const float dcv = dc ? dc[idx] : 0.f;
    float* db = dgates + (long)b * 4 * H;
    if (live && !live[b]) {      // y was 0 there: only the recurrent-path gradient passes through
      db[u] = 0.f; db[H + u] = 0.f; db[2 * H + u] = 0.f; db[3 * H + u] = 0.f;
      dc_prev[idx] = dcv;
      dh_pass[idx] = dhr;
      continue;
    }
    const float dhv = dhr + (dy ? dy[idx] : 0.f);
    const float* ab = acts + (long)b * 4 * H;
    const float ig = ab[u], fg = ab[H + u], gg = ab[2 * H + u], og = ab[3 * H + u];
    const float tc = tanhf(c[idx]);
    const float dct = dcv + dhv * og * (1.f - tc * tc);
    db[u] = dct * gg * ig * (1.f - ig);
    db[H + u] = dct * c_prev[idx] * fg * (1.f - fg);
    db[2 * H + u] = dct * ig * (1.f - gg * gg);
    db[3 * H + u] = dhv * tc * og * (1.f - og);
    dc_prev[idx] = dct * fg;
    dh_pass[idx] = 0.f;
  }
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

int eamd_lstm_cell_bwd(const float* dy, const float* dh, const float* dc, const float* acts, const float* c_prev,
                       const float* c, const uint8_t* live, float* dgates, float* dc_prev, float* dh_pass, int B, int H,
                       void* stream) {
  if ((!dy && !dh && !dc) || !acts || !c_prev || !c || !dgates || !dc_prev || B <= 0 || H <= 0) return EAMD_EINVAL;
  if (live && !dh_pass) return EAMD_EINVAL;
  hipLaunchKernelGGL(lstm_bwd_kernel, dim3(grid_for((long)B * H)), dim3(256), 0, (hipStream_t)stream, dy, dh, dc, acts,
                     c_prev, c, live, dgates, dc_prev, dh_pass, B, H);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

/* waves per workgroup of the step kernels: the largest count <= 16 that leaves every wave a multiple of 16 reduction steps */
static int lstm_step_waves(int K) {
  const int q = K / 16;
  for (int d = 16; d > 1; --d)
    if (q % d == 0) return d;
  return 1;
}

/* one whole LSTM time step in one launch (recurrent product + cell); shapes it takes: H % 64 == 0, B <= 64, 16-byte
 * aligned operands - EAMD_EUNSUPPORTED otherwise (the caller then runs eamd_gemm + eamd_lstm_cell_fwd) */
int eamd_lstm_step_fwd(const float* gx, const float* w_hh, const float* b_hh, const float* h_prev, const float* c_prev,
                       const uint8_t* live, float* h, float* c, float* y, float* acts, int B, int H, void* stream) {
  if (!gx || !w_hh || !h_prev || !c_prev || !h || !c || !acts || B <= 0 || H <= 0) return EAMD_EINVAL;
  if (H % 64 != 0 || B > 64 || (((uintptr_t)gx | (uintptr_t)w_hh | (uintptr_t)h_prev) & 15)) return EAMD_EUNSUPPORTED;
  const int nw = lstm_step_waves(H);
  const int mt = (B + 15) / 16;
  const dim3 grid(H / LS_U), block(64 * nw);
  const size_t lds = (size_t)nw * mt * 16 * 17 * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define EAMD_LSF(MT_) hipLaunchKernelGGL(lstm_step_fwd_kernel<MT_>, grid, block, lds, s, gx, w_hh, b_hh, h_prev, c_prev, live, h, c, y, acts, B, H)
  if (B <= 16) EAMD_LSF(1); else if (B <= 32) EAMD_LSF(2); else if (B <= 48) EAMD_LSF(3); else EAMD_LSF(4);
#undef EAMD_LSF
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

/* one backward step: dh = dh_pass_in + dgates_next W_hh (w_t = W_hh^T [H, 4H]; dgates_next / dh_pass_in / dc may be NULL
 * on the first step), then the cell backward -> dgates [B,4H], dc_prev [B,H], dh_pass [B,H] (masked rows' pass-through) */
int eamd_lstm_step_bwd(const float* dy, const float* dgates_next, const float* w_t, const float* dh_pass_in, const float* dc,
                       const float* acts, const float* c_prev, const float* c, const uint8_t* live, float* dgates,
                       float* dc_prev, float* dh_pass, int B, int H, void* stream) {
  if (!acts || !c_prev || !c || !dgates || !dc_prev || !dh_pass || B <= 0 || H <= 0 || (dgates_next && !w_t)) return EAMD_EINVAL;
  if (H % 64 != 0 || B > 64 || (((uintptr_t)dgates_next | (uintptr_t)w_t) & 15)) return EAMD_EUNSUPPORTED;
  const int nw = lstm_step_waves(4 * H);
  const int mt = (B + 15) / 16;
  const dim3 grid(H / LB_U), block(64 * nw);
  const size_t lds = (size_t)nw * mt * 16 * 17 * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define EAMD_LSB(MT_) hipLaunchKernelGGL(lstm_step_bwd_kernel<MT_>, grid, block, lds, s, dy, dgates_next, w_t, dh_pass_in, dc, acts, c_prev, c, live, dgates, dc_prev, dh_pass, B, H)
  if (B <= 16) EAMD_LSB(1); else if (B <= 32) EAMD_LSB(2); else if (B <= 48) EAMD_LSB(3); else EAMD_LSB(4);
#undef EAMD_LSB
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_gru_cell_fwd(const float* gx, const float* gh, const float* h_prev, const uint8_t* live, float* h, float* y,
                      float* acts, int B, int H, void* stream) {
  if (!gx || !gh || !h_prev || !h || !acts || B <= 0 || H <= 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(gru_fwd_kernel, dim3(grid_for((long)B * H)), dim3(256), 0, (hipStream_t)stream, gx, gh, h_prev, live, h,
                     y, acts, B, H);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_gru_cell_bwd(const float* dy, const float* dh, const float* acts, const float* h_prev, const uint8_t* live,
                      float* dgx, float* dgh, float* dh_direct, int B, int H, void* stream) {
  if ((!dy && !dh) || !acts || !h_prev || !dgx || !dgh || !dh_direct || B <= 0 || H <= 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(gru_bwd_kernel, dim3(grid_for((long)B * H)), dim3(256), 0, (hipStream_t)stream, dy, dh, acts, h_prev,
                     live, dgx, dgh, dh_direct, B, H);
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

int eamd_attloc_fwd(const float* att_prev, const float* conv_w, const float* w_att, const float* pre_enc,
                    const float* dec_proj, const float* gvec, const float* gb, const int32_t* lens, const float* enc_h,
                    float scaling, float* e, float* th, float* conv, float* w, float* ctx, int B, int T, int A, int C,
                    int K, int R, int E, void* stream) {
  if (!pre_enc || !dec_proj || !gvec || !gb || !lens || !enc_h || !e || !th || !w || !ctx || B <= 0 || T <= 0 ||
      A <= 0 || C < 0 || E <= 0)
    return EAMD_EINVAL;
  if (C > 0 && (!att_prev || !conv_w || !w_att || !conv || K <= 0 || R <= 0)) return EAMD_EINVAL;   // C = 0: additive attention
  if (C > 64 || (C > 0 && (K & 1) == 0) || (size_t)T * sizeof(float) > 60 * 1024) return EAMD_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const bool al = (((uintptr_t)pre_enc | (uintptr_t)th | (uintptr_t)dec_proj | (uintptr_t)gvec) & 15) == 0;
  if (C > 0 && C <= 16 && R == 1 && K <= ATTE_MAXK && A % 4 == 0 && A <= 1024 && al) {
    const dim3 grid(B, (T + ATTE_TCH - 1) / ATTE_TCH);
    if (C == 10)
      hipLaunchKernelGGL(attloc_energy_fwd8_kernel<10>, grid, dim3(256), 0, s, att_prev, conv_w, w_att, pre_enc, dec_proj, gvec,
                         gb, lens, e, th, conv, T, A, C, K, R);
    else
      hipLaunchKernelGGL(attloc_energy_fwd8_kernel<16>, grid, dim3(256), 0, s, att_prev, conv_w, w_att, pre_enc, dec_proj, gvec,
                         gb, lens, e, th, conv, T, A, C, K, R);
  } else {
    hipLaunchKernelGGL(attloc_energy_fwd_kernel, dim3(B * T), dim3(128), 0, s, att_prev, conv_w, w_att, pre_enc, dec_proj,
                       gvec, gb, lens, e, th, conv, T, A, C, K, C > 0 ? R : 1);
  }
  EAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(attloc_ctx_fwd_kernel, dim3(B, (E + 63) / 64), dim3(256), T * sizeof(float), s, e, enc_h, scaling,
                     w, ctx, T, E);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

/* stage 1 of the backward: de, d_enc_h, dgb, then df / dgvec / d_dec_proj (d_dec_proj, dgvec, dgb accumulate) */
int eamd_attloc_bwd_energy(const float* dctx, const float* dw_ext, const float* w, const float* enc_h, const float* th,
                           const float* gvec, float scaling, float* de, float* d_enc_h, float* df, float* dgvec,
                           float* dgb, float* d_dec_proj, int B, int T, int A, int E, void* stream) {
  if (!dctx || !w || !enc_h || !th || !gvec || !de || !d_enc_h || !df || !dgvec || !dgb || !d_dec_proj || B <= 0 ||
      T <= 0 || A <= 0 || E <= 0)
    return EAMD_EINVAL;
  if ((size_t)T * sizeof(float) > 60 * 1024) return EAMD_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(attloc_ctx_bwd_rows_kernel, dim3((B * T + 3) / 4), dim3(256), 0, s, dctx, dw_ext, w, enc_h, de, d_enc_h,
                     B * T, T, E);
  EAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(attloc_softmax_bwd_kernel, dim3(B), dim3(256), 0, s, w, scaling, de, dgb, T);
  EAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(attloc_energy_bwd_kernel, dim3(B, (T + ATT_TCH - 1) / ATT_TCH), dim3(256), 0, s, de, th, gvec, df,
                     dgvec, d_dec_proj, T, A);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

/* stage 1 with the W_att products folded in (see attloc_energy_bwd_fused_kernel): additionally dconv [B,T,C] = df @ W_att is
 * written and dw_att [A,C] += df^T conv accumulated.  workspace: eamd_attloc_bwd_workspace(B, T, A, C) bytes. */
int64_t eamd_attloc_bwd_workspace(int B, int T, int A, int C) {
  return (int64_t)B * ((T + ATTF_TCH - 1) / ATTF_TCH) * A * (C + 2) * (int64_t)sizeof(float);
}
int eamd_attloc_bwd_energy_conv(const float* dctx, const float* dw_ext, const float* w, const float* enc_h, const float* th,
                                const float* gvec, float scaling, const float* conv, const float* w_att, float* de,
                                float* d_enc_h, float* df, float* dconv, float* dgvec, float* dgb, float* d_dec_proj,
                                float* dw_att, float* workspace, int accumulate, int B, int T, int A, int C, int E,
                                void* stream) {
  if (!dctx || !w || !enc_h || !th || !gvec || !conv || !w_att || !de || !d_enc_h || !df || !dconv || !dgvec || !dgb ||
      !d_dec_proj || !dw_att || !workspace || B <= 0 || T <= 0 || A <= 0 || C <= 0 || E <= 0)
    return EAMD_EINVAL;
  if ((size_t)T * sizeof(float) > 60 * 1024 || C > ATTF_MAXC || A % 4 || A > 1024) return EAMD_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(th) | reinterpret_cast<uintptr_t>(df) | reinterpret_cast<uintptr_t>(gvec) |
       reinterpret_cast<uintptr_t>(workspace)) & 15)
    return EAMD_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int nch = (T + ATTF_TCH - 1) / ATTF_TCH;
  hipLaunchKernelGGL(attloc_ctx_bwd_rows_kernel, dim3((B * T + 3) / 4), dim3(256), 0, s, dctx, dw_ext, w, enc_h, de, d_enc_h,
                     B * T, T, E, accumulate);
  EAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(attloc_softmax_bwd_kernel, dim3(B), dim3(256), 0, s, w, scaling, de, dgb, T);
  EAMD_LAUNCH_CHECK();
  // exact unrolls for the usual channel counts (aconv_chans = 10 in the recipes), the padded 16-channel form otherwise
  if (C == 10)
    hipLaunchKernelGGL(attloc_energy_bwd_fused_kernel<10>, dim3(B, nch), dim3(256), 0, s, de, th, gvec, conv, w_att, df, dconv,
                       workspace, T, A, C, accumulate);
  else if (C == 4)
    hipLaunchKernelGGL(attloc_energy_bwd_fused_kernel<4>, dim3(B, nch), dim3(256), 0, s, de, th, gvec, conv, w_att, df, dconv,
                       workspace, T, A, C, accumulate);
  else
    hipLaunchKernelGGL(attloc_energy_bwd_fused_kernel<ATTF_MAXC>, dim3(B, nch), dim3(256), 0, s, de, th, gvec, conv, w_att, df,
                       dconv, workspace, T, A, C, accumulate);
  EAMD_LAUNCH_CHECK();
  const long n = (long)A * (C + 1) + (long)B * A;
  hipLaunchKernelGGL(attloc_bwd_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, workspace, dw_att, dgvec,
                     d_dec_proj, B, nch, A, C);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

/* stage 2 (after dconv = df @ W_att has been formed by eamd_gemm): gradients of the location convolution */
int eamd_attloc_bwd_conv(const float* dconv, const float* conv_w, const float* att_prev, float* d_prev, float* dconv_w,
                         int B, int T, int C, int K, int R, void* stream) {
  if (!dconv || !conv_w || !att_prev || !d_prev || !dconv_w || B <= 0 || T <= 0 || C <= 0 || K <= 0 || (K & 1) == 0 ||
      R <= 0)
    return EAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(attloc_conv_bwd_prev_kernel, dim3((B * R * T + 3) / 4), dim3(256), 0, s, dconv, conv_w, d_prev,
                     B * R * T, T, C, K, R);
  EAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(attloc_conv_bwd_w_kernel, dim3(C, B), dim3(256), 0, s, dconv, att_prev, dconv_w, T, C, K, R);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_attloc_convmax_fwd(const float* att_prev, const float* conv_w, float* pooled, int32_t* idx, int B, int T, int C,
                            int K, void* stream) {
  if (!att_prev || !conv_w || !pooled || !idx || B <= 0 || T <= 0 || C <= 0 || K <= 0 || (K & 1) == 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(attloc_convmax_fwd_kernel, dim3((B * C + 3) / 4), dim3(256), 0, (hipStream_t)stream, att_prev, conv_w,
                     pooled, idx, B * C, T, C, K);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

/* d_prev [B,T] and dconv_w [C,K] are ACCUMULATED (zero d_prev first) */
int eamd_attloc_convmax_bwd(const float* dpool, const float* pooled, const int32_t* idx, const float* att_prev,
                            const float* conv_w, float* d_prev, float* dconv_w, int B, int T, int C, int K, void* stream) {
  if (!dpool || !pooled || !idx || !att_prev || !conv_w || !d_prev || !dconv_w || B <= 0 || T <= 0 || C <= 0 || K <= 0 ||
      (K & 1) == 0)
    return EAMD_EINVAL;
  hipLaunchKernelGGL(attloc_convmax_bwd_kernel, dim3(grid_for((long)B * C * K)), dim3(256), 0, (hipStream_t)stream, dpool,
                     pooled, idx, att_prev, conv_w, d_prev, dconv_w, B, T, C, K);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

/* softmax(scaling * e) over the frames + context vector, for energies produced elsewhere (dot-product attention) */
int eamd_att_ctx_fwd(const float* e, const float* v, float scaling, float* w, float* ctx, int B, int T, int E,
                     void* stream) {
  if (!e || !v || !w || !ctx || B <= 0 || T <= 0 || E <= 0) return EAMD_EINVAL;
  if ((size_t)T * sizeof(float) > 60 * 1024) return EAMD_EUNSUPPORTED;
  hipLaunchKernelGGL(attloc_ctx_fwd_kernel, dim3(B, (E + 63) / 64), dim3(256), T * sizeof(float), (hipStream_t)stream, e, v,
                     scaling, w, ctx, T, E);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}
int eamd_att_ctx_bwd(const float* dctx, const float* dw_ext, const float* w, const float* v, float scaling, float* de,
                     float* d_v, float* dsum, int B, int T, int E, void* stream) {
  if (!dctx || !w || !v || !de || !d_v || !dsum || B <= 0 || T <= 0 || E <= 0) return EAMD_EINVAL;
  if ((size_t)T * sizeof(float) > 60 * 1024) return EAMD_EUNSUPPORTED;
  hipLaunchKernelGGL(attloc_ctx_bwd_rows_kernel, dim3((B * T + 3) / 4), dim3(256), 0, (hipStream_t)stream, dctx, dw_ext, w, v,
                     de, d_v, B * T, T, E);
  EAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(attloc_softmax_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, w, scaling, de, dsum, T);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}
int eamd_att_dot_energy_fwd(const float* k, const float* q, const int32_t* lens, float* e, int B, int T, int A,
                            void* stream) {
  if (!k || !q || !lens || !e || B <= 0 || T <= 0 || A <= 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(att_dot_energy_fwd_kernel, dim3(B * T), dim3(64), 0, (hipStream_t)stream, k, q, lens, e, T, A);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}
/* dk [B,T,A] written, dq [B,A] ACCUMULATED */
int eamd_att_dot_energy_bwd(const float* de, const float* k, const float* q, float* dk, float* dq, int B, int T, int A,
                            void* stream) {
  if (!de || !k || !q || !dk || !dq || B <= 0 || T <= 0 || A <= 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(att_dot_energy_bwd_kernel, dim3(B, (T + ATT_TCH - 1) / ATT_TCH), dim3(256), 0, (hipStream_t)stream, de,
                     k, q, dk, dq, T, A);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

}  // extern "C"
