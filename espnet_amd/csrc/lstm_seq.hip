// Persistent LSTM sequence kernels for gfx950: ALL time steps of up to two independent recurrences (the two directions of
// a BLSTM layer) in ONE launch.
//
// reference: torch.nn.LSTM as rnn/encoders.py:36-39,110-117 runs it (bidirectional, packed sequences), gate order
//            i, f, g, o; backward = autograd of the same.
//
// Why: as one launch per time step (rnn.hip, eamd_lstm_step_fwd / _bwd) a step re-reads its 16 MB recurrent matrix from
// L2, pays a kernel boundary, and the two directions of a layer queue behind each other: 3000 launches of 18 us at
// BASELINE config 4 (B = 32, H = 1024, T' = 250, 3 layers x 2 directions), half of that model's step.  Here
//   * a workgroup owns a slice of hidden units for the whole sequence and keeps its rows of W_hh (forward) / W_hh^T
//     (backward) in REGISTERS as MFMA operand fragments, loaded once: 32 resp. 64 VGPRs per lane at H = 1024;
//   * its cell state (c, and h for masked frames; dc and the masked pass-through in backward) lives in registers too;
//   * the only thing exchanged between workgroups per step is h_t (forward, B x H) resp. dgates_t (backward, B x 4H),
//     which are outputs anyway: stored write-through (sc1), published with ONE flag per workgroup, and read back by
//     every consumer with sc1 loads after ONE wave has polled the flags of the producers it depends on (the
//     placement-independent hand-off of the CDNA programming guide, Guideline 16 / MI355X visibility table, first row:
//     one lane of each storing workgroup signals for all its stores after every storing wave's vmcnt(0) wait);
//   * both directions of a layer run side by side on disjoint workgroups of the same launch (grid <= number of CUs, one
//     workgroup per CU, so every workgroup is resident and every spin is bounded by a wall-clock limit that sets a
//     status word, writes NaN into the step's outputs and lets the whole grid drain).  Residency is the caller's side of
//     the contract: two such launches running CONCURRENTLY on one device (two streams, or two processes sharing a GPU)
//     can each hold half the CUs while waiting for workgroups that cannot start - they then time out, they do not hang.
// The reduction over the recurrent inputs is split over the waves of a workgroup exactly as in the per-step kernels
// (results agree with theirs to the last bit or two: only the compiler's fma contraction of the cell update differs).
#include <stdlib.h>
#include "common.h"
#include "../../include/espnet_amd.h"

namespace {

__device__ __forceinline__ float sigm_(float x) { return 1.0f / (1.0f + expf(-x)); }

// 16 bytes of handed-off data as two 8-byte agent-scope relaxed loads (global_load_dwordx2 sc1: served by L2, never by
// this CU's L1).
__device__ __forceinline__ f32x4 ld_sc1_x4(const float* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (f32x4){__builtin_bit_cast(float, (unsigned)a), __builtin_bit_cast(float, (unsigned)(a >> 32)),
                 __builtin_bit_cast(float, (unsigned)b), __builtin_bit_cast(float, (unsigned)(b >> 32))};
}

constexpr int SYNC_TMO_WORD = 0;        // status word: 0 = fine, else the code of the wait that gave up
constexpr int SYNC_FLAG0 = 64;          // flags[blockIdx.x] start 256 bytes in
constexpr long long SPIN_LIMIT = 300000000LL;   // 3 s of the 100 MHz wall clock

__device__ __forceinline__ unsigned ld_flag(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ONE wave waits until flags[first + i * stride] >= epoch for i < n (n <= 256); returns false when it gave up (or another
// workgroup already had): the caller then leaves its time loop.
__device__ __forceinline__ bool wait_flags(unsigned* ws, int first, int n, unsigned epoch, int lane, unsigned code) {
  const long long t0 = wall_clock64();
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = lane + 64 * i;
      if (k < n) ok &= ld_flag(ws + SYNC_FLAG0 + first + k) >= epoch;
    }
    const unsigned dead = ld_flag(ws + SYNC_TMO_WORD);
    if (__all(ok)) return dead == 0;
    if (dead != 0) return false;
    if (wall_clock64() - t0 > SPIN_LIMIT) {
      if (lane == 0) __hip_atomic_store(ws + SYNC_TMO_WORD, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

// timing probes of the diagnostic build (-DLSTM_SEQ_PROBE, EAMD_LSTM_PROBE bits; results are wrong by design):
// 1 = no MFMAs, 2 = no loads of the handed-over rows, 4 = no waiting for the flags, 8 = no cell arithmetic / stores,
// 16 = workgroup 0 never publishes its flag (exercises the give-up path: status word, NaN outputs, drain)
#ifdef LSTM_SEQ_PROBE
#define PROBE(bit) ((a.probe & (bit)) != 0)
#else
#define PROBE(bit) false
#endif

// 16 bytes of hand-off payload as two write-through (sc1) 8-byte stores
__device__ __forceinline__ void st_sc1_x4(float* p, f32x4 v) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
  // (scalar copies first: hipcc 7.2 folds __builtin_bit_cast(unsigned, v[i]) of an ext_vector ELEMENT to element 0)
  const float v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
  const unsigned long long a = (unsigned long long)__float_as_uint(v0) | ((unsigned long long)__float_as_uint(v1) << 32);
  const unsigned long long b = (unsigned long long)__float_as_uint(v2) | ((unsigned long long)__float_as_uint(v3) << 32);
  __hip_atomic_store(q, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the values of lanes l .. l + 3 (l % 4 == 0 uses the result): four adjacent hidden units of one row leave as one 16-byte store
__device__ __forceinline__ f32x4 quad_gather(float x) {
  return (f32x4){x, __shfl_down(x, 1), __shfl_down(x, 2), __shfl_down(x, 3)};
}

// handed-over rows: plain 16-byte loads behind the polling wave's agent-scope acquire (acq mode 1, the default: the 32
// workgroups of an XCD then share the rows through its L2 - 11.5 / 12.3 us per step forward / backward at config 4), or
// sc1 loads with no acquire (mode 0: every workgroup pulls its own copy over the fabric, 12.4 / 16.8 us)
#define LD_X4(p) (a.acq == 1 ? *reinterpret_cast<const f32x4*>(p) : ld_sc1_x4(p))

struct FwdJob {
  const float* gx; const float* w_hh; const float* b_hh; const unsigned char* live;
  float* h_out; float* c_out; float* y; float* acts;
  int reverse;
};
struct FwdArgs { FwdJob job[2]; int T, B, H, nut, nmt; unsigned* ws; int acq, probe; };

// Forward.  Workgroup = (job, 16 * MT batch rows, U = 4 * NT hidden units): B-operand tile nt row n = gate (n / 4) of
// unit u0 + 4 nt + n % 4.  NW waves split the reduction over the H recurrent inputs, KQ quad-steps of 16 each.  It
// waits only for the workgroups of its own batch-row tile (the rows of h it multiplies).
template <int MT, int NT, int KQ>
__global__ __launch_bounds__(1024) void lstm_seq_fwd_kernel(const FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int PW = NT * 16 + 1;
  float (*part)[MT * 16][PW] = reinterpret_cast<float (*)[MT * 16][PW]>(smem);
  __shared__ int go;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), fr = lane & 15, fq = lane >> 4;
  const int NW = blockDim.x >> 6;
  const int per_job = a.nut * a.nmt;
  const int jb = blockIdx.x / per_job, rem = blockIdx.x % per_job;
  const int mt = rem / a.nut, wg = rem % a.nut;       // the workgroups of one batch-row tile are neighbours in the flag array
  const FwdJob J = a.job[jb];
  const int B = a.B, H = a.H, T = a.T;
  constexpr int U = 4 * NT;
  const int u0 = wg * U, b0 = mt * 16 * MT;
  const int kbeg = wave * (KQ * 16);
  // recurrent weights: registers for the whole sequence
  f32x4 bv[NT][KQ];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const float* wrow = J.w_hh + ((long)(fr >> 2) * H + u0 + 4 * nt + (fr & 3)) * H;
#pragma unroll
    for (int q = 0; q < KQ; ++q) bv[nt][q] = *reinterpret_cast<const f32x4*>(wrow + kbeg + q * 16 + fq * 4);
  }
  // the cell of (row r, unit j) of the tile belongs to thread r * U + j for all steps
  const int cr = t / U, cb = b0 + cr, cj = t % U, cu = u0 + cj;
  const bool cell = t < 16 * MT * U && cb < B;
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if (cell && J.b_hh)
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = J.b_hh[g * H + cu];
  float c_reg = 0.f, h_reg = 0.f;
  if (a.acq == 2 && wave == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  int aoff[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) aoff[i] = min(b0 + i * 16 + fr, B - 1) * H + kbeg + fq * 4;

  for (int s = 0; s < T; ++s) {
    const int tt = J.reverse ? T - 1 - s : s;
    const int tp = J.reverse ? tt + 1 : tt - 1;
    // this step's input-side gate rows do not depend on the exchange: request them first
    float gxv[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned char lv = 1;
    if (cell) {
      const float* gp = J.gx + ((long)tt * B + cb) * 4 * H + cu;
#pragma unroll
      for (int g = 0; g < 4; ++g) gxv[g] = gp[g * H];
      if (J.live) lv = J.live[(long)tt * B + cb];
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[i][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      if (wave == 0) {
        const bool ok = PROBE(4) ? true : wait_flags(a.ws, jb * per_job + mt * a.nut, a.nut, (unsigned)s, lane, 0x100u + jb);
        if (lane == 0) go = ok ? 1 : 0;
        if (a.acq == 1) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      __syncthreads();
      if (!go) {                                    // workgroup-uniform: a wait gave up somewhere - poison this and every later
        if (cell)                                   // frame of the outputs (the loss turns NaN whichever frames reach it, the
          for (int s2 = s; s2 < T; ++s2) {          // optimizer's non-finite guard skips the step) and drain
            const long idx = ((long)(J.reverse ? T - 1 - s2 : s2) * B + cb) * H + cu;
            J.h_out[idx] = __builtin_nanf("");
            if (J.y) J.y[idx] = __builtin_nanf("");
          }
        break;
      }
      const float* hp = J.h_out + (long)tp * B * H;
      f32x4 av[MT][KQ];
#pragma unroll
      for (int q = 0; q < KQ; ++q)
#pragma unroll
        for (int i = 0; i < MT; ++i) av[i][q] = PROBE(2) ? (f32x4){0.f, 0.f, 0.f, 0.f} : LD_X4(hp + aoff[i] + q * 16);
#pragma unroll
      for (int q = 0; q < KQ; ++q)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float ae = av[i][q][e];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              if (!PROBE(1)) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae, bv[nt][q][e], acc[i][nt], 0, 0, 0);
          }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][i * 16 + fq * 4 + r][nt * 16 + fr] = acc[i][nt][r];
    __syncthreads();
    if (cell && !PROBE(8)) {
      float g4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = (cj >> 2) * 16 + g * 4 + (cj & 3);
        float sacc = 0.f;
        for (int w = 0; w < NW; ++w) sacc += part[w][cr][n];
        g4[g] = sacc + gxv[g] + bias4[g];
      }
      const float ig = sigm_(g4[0]), fg = sigm_(g4[1]), gg = tanhf(g4[2]), og = sigm_(g4[3]);
      float cn = fg * c_reg + ig * gg;
      float hn = og * tanhf(cn);
      float yo = hn;
      if (!lv) { cn = c_reg; hn = h_reg; yo = 0.f; }
      c_reg = cn; h_reg = hn;
      const long idx = ((long)tt * B + cb) * H + cu;
      const f32x4 hv = quad_gather(hn), cv = quad_gather(cn), yv = quad_gather(yo);
      const f32x4 iv = quad_gather(ig), fv = quad_gather(fg), gv = quad_gather(gg), ov = quad_gather(og);
      if ((cj & 3) == 0) {
        st_sc1_x4(J.h_out + idx, hv);                       // sc1: the payload of the hand-off
        *reinterpret_cast<f32x4*>(J.c_out + idx) = cv;
        if (J.y) *reinterpret_cast<f32x4*>(J.y + idx) = yv;
        float* ab = J.acts + ((long)tt * B + cb) * 4 * H + cu;
        *reinterpret_cast<f32x4*>(ab) = iv;
        *reinterpret_cast<f32x4*>(ab + H) = fv;
        *reinterpret_cast<f32x4*>(ab + 2 * H) = gv;
        *reinterpret_cast<f32x4*>(ab + 3 * H) = ov;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains before the barrier
    __syncthreads();                                         // also: part[] free for the next step
    if (t == 0 && !(PROBE(16) && blockIdx.x == 0))
      __hip_atomic_store(a.ws + SYNC_FLAG0 + blockIdx.x, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

struct BwdJob {
  const float* dy; const float* w_t; const float* acts; const float* c_out; const unsigned char* live;
  float* dgates;
  int reverse;
};
struct BwdArgs { BwdJob job[2]; int T, B, H, nut, nmt; unsigned* ws; int acq, probe; };

// Backward.  Workgroup = (job, 16 hidden units, 16 batch rows):
//   dh[b, u] = pass[b, u] + sum_r dgates_next[b, r] W_hh[r, u]   (r over the 4H gate rows, split over NW waves x KQ quad-steps),
// then the cell backward of its 256 (b, u) pairs.  It waits only for the workgroups of its own batch-row tile.
template <int KQ>
__global__ __launch_bounds__(1024) void lstm_seq_bwd_kernel(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float (*part)[16][17] = reinterpret_cast<float (*)[16][17]>(smem);
  __shared__ int go;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), fr = lane & 15, fq = lane >> 4;
  const int NW = blockDim.x >> 6;
  const int per_job = a.nut * a.nmt;
  const int jb = blockIdx.x / per_job, rem = blockIdx.x % per_job;
  const int mt = rem / a.nut, ut = rem % a.nut;       // the workgroups of one batch-row tile are neighbours in the flag array
  const BwdJob J = a.job[jb];
  const int B = a.B, H = a.H, T = a.T, K = 4 * H;
  const int u0 = ut * 16, b0 = mt * 16;
  const int kbeg = wave * (KQ * 16);
  f32x4 bv[KQ];
  {
    const float* wrow = J.w_t + (long)(u0 + fr) * K + kbeg + fq * 4;
#pragma unroll
    for (int q = 0; q < KQ; ++q) bv[q] = *reinterpret_cast<const f32x4*>(wrow + q * 16);
  }
  const int cb = b0 + (t >> 4), cu = u0 + (t & 15);
  const bool cell = t < 256 && cb < B;
  float dc_reg = 0.f, pass_reg = 0.f;
  if (a.acq == 2 && wave == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const int aoff = min(b0 + fr, B - 1) * K + kbeg + fq * 4;
  constexpr int CH = KQ < 4 ? KQ : 4;               // quad-steps per batch of loads

  for (int s = 0; s < T; ++s) {
    const int tt = J.reverse ? s : T - 1 - s;            // backward walks the sequence against the forward direction
    const int tn = J.reverse ? tt - 1 : tt + 1;          // frame handled one backward step earlier
    const int pt = J.reverse ? tt + 1 : tt - 1;          // frame whose state fed this one in forward
    float av4[4] = {0.f, 0.f, 0.f, 0.f}, cv = 0.f, cpv = 0.f, dyv = 0.f;
    unsigned char lv = 1;
    if (cell) {
      const float* ab = J.acts + ((long)tt * B + cb) * 4 * H + cu;
#pragma unroll
      for (int g = 0; g < 4; ++g) av4[g] = ab[g * H];
      const long idx = ((long)tt * B + cb) * H + cu;
      cv = J.c_out[idx];
      if (pt >= 0 && pt < T) cpv = J.c_out[((long)pt * B + cb) * H + cu];
      if (J.dy) dyv = J.dy[idx];
      if (J.live) lv = J.live[(long)tt * B + cb];
    }
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      if (wave == 0) {
        const bool ok = PROBE(4) ? true : wait_flags(a.ws, jb * per_job + mt * a.nut, a.nut, (unsigned)s, lane, 0x200u + jb);
        if (lane == 0) go = ok ? 1 : 0;
        if (a.acq == 1) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      __syncthreads();
      if (!go) {                                    // as in the forward: NaN into this and every later step's gradients, then drain
        if (cell)
          for (int s2 = s; s2 < T; ++s2) J.dgates[((long)(J.reverse ? s2 : T - 1 - s2) * B + cb) * K + cu] = __builtin_nanf("");
        break;
      }
      const float* dg = J.dgates + (long)tn * B * K;
      f32x4 av[2][CH];
#pragma unroll
      for (int q = 0; q < CH; ++q) av[0][q] = PROBE(2) ? (f32x4){0.f, 0.f, 0.f, 0.f} : LD_X4(dg + aoff + q * 16);
#pragma unroll
      for (int c = 0; c < KQ / CH; ++c) {
        if (c + 1 < KQ / CH)
#pragma unroll
          for (int q = 0; q < CH; ++q)
            av[(c + 1) & 1][q] = PROBE(2) ? (f32x4){0.f, 0.f, 0.f, 0.f} : LD_X4(dg + aoff + ((c + 1) * CH + q) * 16);
#pragma unroll
        for (int q = 0; q < CH; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (!PROBE(1)) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c & 1][q][e], bv[c * CH + q][e], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][fq * 4 + r][fr] = acc[r];
    __syncthreads();
    if (cell && !PROBE(8)) {
      float dhr = pass_reg;
      if (s > 0)
        for (int w = 0; w < NW; ++w) dhr += part[w][t >> 4][t & 15];
      float* db = J.dgates + ((long)tt * B + cb) * K + cu;
      float d4[4];
      if (!lv) {           // y was 0 there: only the recurrent-path gradient passes through
        d4[0] = d4[1] = d4[2] = d4[3] = 0.f;
        pass_reg = dhr;    // dc_reg unchanged
      } else {
        const float dhv = dhr + dyv;
        const float ig = av4[0], fg = av4[1], gg = av4[2], og = av4[3];
        const float tc = tanhf(cv);
        const float dct = dc_reg + dhv * og * (1.f - tc * tc);
        d4[0] = dct * gg * ig * (1.f - ig);
        d4[1] = dct * cpv * fg * (1.f - fg);
        d4[2] = dct * ig * (1.f - gg * gg);
        d4[3] = dhv * tc * og * (1.f - og);
        dc_reg = dct * fg;
        pass_reg = 0.f;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 dv = quad_gather(d4[g]);
        if ((t & 3) == 0) st_sc1_x4(db + g * H, dv);        // sc1: the payload of the hand-off
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0 && !(PROBE(16) && blockIdx.x == 0))
      __hip_atomic_store(a.ws + SYNC_FLAG0 + blockIdx.x, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int device_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    cus = prop.multiProcessorCount;
  }
  return cus;
}

int acq_mode() {
  static int m = -1;
  if (m < 0) {
    const char* e = getenv("EAMD_LSTM_SEQ_ACQ");
    m = e ? atoi(e) : 1;
  }
  return m;
}

int probe_bits() {
#ifdef LSTM_SEQ_PROBE
  const char* e = getenv("EAMD_LSTM_PROBE");
  return e ? atoi(e) : 0;
#else
  return 0;
#endif
}

constexpr int64_t SYNC_BYTES = 4096;     // status word + 256 bytes of padding + up to 960 flags

}  // namespace

extern "C" {

int64_t eamd_lstm_seq_sync_bytes(void) { return SYNC_BYTES; }

int eamd_lstm_seq_fwd(const eamd_lstm_seq_fwd_t* jobs, int njobs, int T, int B, int H, void* sync_ws, void* stream) {
  if (!jobs || njobs < 1 || njobs > 2 || T <= 0 || B <= 0 || H <= 0 || !sync_ws) return EAMD_EINVAL;
  for (int j = 0; j < njobs; ++j) {
    const eamd_lstm_seq_fwd_t& q = jobs[j];
    if (!q.gx || !q.w_hh || !q.h_out || !q.c_out || !q.acts) return EAMD_EINVAL;
    if (((uintptr_t)q.gx | (uintptr_t)q.w_hh | (uintptr_t)q.h_out) & 15) return EAMD_EUNSUPPORTED;
  }
  if (H % 64 != 0 || B > 64 || (int64_t)B * H * 4 >= (1LL << 31)) return EAMD_EUNSUPPORTED;
  // waves: KQ quad-steps of 16 recurrent inputs each, KQ in {1, 2, 4}, at most 16 waves (the per-step kernels' split)
  int kq = 0;
  for (int c : {1, 2, 4})
    if (H % (16 * c) == 0 && H / (16 * c) <= 16) { kq = c; break; }
  if (!kq) return EAMD_EUNSUPPORTED;
  const int nw = H / (16 * kq);
  const int cus = device_cus();
  // (rows, units) per workgroup: as many workgroups as there are CUs (the MFMA work spreads), then as few rows as
  // possible (a workgroup reads its rows of h every step: 64 KB for 16 rows at H = 1024), then as many units
  int mt = 0, nt = 0, best = 0;
  for (int m : {1, 2, 4})
    for (int n : {4, 2, 1}) {
      if (H % (4 * n) != 0 || 16 * m * 4 * n > 64 * nw) continue;
      if (m > 1 && 16 * (m / 2) >= B) continue;                     // tiles of rows that do not exist
      const int wgs = njobs * (H / (4 * n)) * ((B + 16 * m - 1) / (16 * m));
      if (wgs > cus || wgs > 960 || H / (4 * n) > 256) continue;
      if (wgs > best) { best = wgs; mt = m; nt = n; }
    }
  if (!best) return EAMD_EUNSUPPORTED;
  const int nut = H / (4 * nt), nmt = (B + 16 * mt - 1) / (16 * mt);
  hipStream_t s = (hipStream_t)stream;
  if (eamd_zero_async(sync_ws, SYNC_BYTES, s) != EAMD_OK) return EAMD_EINVAL;
  FwdArgs a;
  for (int j = 0; j < 2; ++j) {
    const eamd_lstm_seq_fwd_t& q = jobs[j < njobs ? j : 0];
    a.job[j] = FwdJob{q.gx, q.w_hh, q.b_hh, q.live, q.h_out, q.c_out, q.y, q.acts, q.reverse};
  }
  a.T = T; a.B = B; a.H = H; a.nut = nut; a.nmt = nmt; a.ws = (unsigned*)sync_ws; a.acq = acq_mode(); a.probe = probe_bits();
  const dim3 grid(njobs * nut * nmt), block(64 * nw);
  const size_t lds = (size_t)nw * mt * 16 * (nt * 16 + 1) * sizeof(float);
#define EAMD_LQF(MT_, NT_, KQ_)                                                                                          \
  do {                                                                                                                   \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)lstm_seq_fwd_kernel<MT_, NT_, KQ_>,                             \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
    hipLaunchKernelGGL((lstm_seq_fwd_kernel<MT_, NT_, KQ_>), grid, block, lds, s, a);                                     \
  } while (0)
#define EAMD_LQF_K(MT_, NT_)                                                                                             \
  do {                                                                                                                   \
    if (kq == 1) EAMD_LQF(MT_, NT_, 1); else if (kq == 2) EAMD_LQF(MT_, NT_, 2); else EAMD_LQF(MT_, NT_, 4);             \
  } while (0)
#define EAMD_LQF_N(MT_)                                                                                                  \
  do {                                                                                                                   \
    if (nt == 1) EAMD_LQF_K(MT_, 1); else if (nt == 2) EAMD_LQF_K(MT_, 2); else EAMD_LQF_K(MT_, 4);                      \
  } while (0)
  if (mt == 1) EAMD_LQF_N(1); else if (mt == 2) EAMD_LQF_N(2); else EAMD_LQF_N(4);
#undef EAMD_LQF_N
#undef EAMD_LQF_K
#undef EAMD_LQF
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_lstm_seq_bwd(const eamd_lstm_seq_bwd_t* jobs, int njobs, int T, int B, int H, void* sync_ws, void* stream) {
  if (!jobs || njobs < 1 || njobs > 2 || T <= 0 || B <= 0 || H <= 0 || !sync_ws) return EAMD_EINVAL;
  for (int j = 0; j < njobs; ++j) {
    const eamd_lstm_seq_bwd_t& q = jobs[j];
    if (!q.w_t || !q.acts || !q.c_out || !q.dgates) return EAMD_EINVAL;
    if (((uintptr_t)q.w_t | (uintptr_t)q.dgates) & 15) return EAMD_EUNSUPPORTED;
  }
  if (H % 16 != 0 || H % 64 != 0 || B > 64 || (int64_t)B * H * 16 >= (1LL << 31)) return EAMD_EUNSUPPORTED;
  const int K = 4 * H;
  int kq = 0;
  for (int c : {1, 2, 4, 8, 16})
    if (K % (16 * c) == 0 && K / (16 * c) <= 16) { kq = c; break; }
  if (!kq) return EAMD_EUNSUPPORTED;
  const int nw = K / (16 * kq);
  if (nw < 4) return EAMD_EUNSUPPORTED;              // the 256 cell threads
  const int nut = H / 16, nmt = (B + 15) / 16;
  if (njobs * nut * nmt > device_cus() || njobs * nut * nmt > 960 || nut > 256) return EAMD_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (eamd_zero_async(sync_ws, SYNC_BYTES, s) != EAMD_OK) return EAMD_EINVAL;
  BwdArgs a;
  for (int j = 0; j < 2; ++j) {
    const eamd_lstm_seq_bwd_t& q = jobs[j < njobs ? j : 0];
    a.job[j] = BwdJob{q.dy, q.w_t, q.acts, q.c_out, q.live, q.dgates, q.reverse};
  }
  a.T = T; a.B = B; a.H = H; a.nut = nut; a.nmt = nmt; a.ws = (unsigned*)sync_ws; a.acq = acq_mode(); a.probe = probe_bits();
  const dim3 grid(njobs * nut * nmt), block(64 * nw);
  const size_t lds = (size_t)nw * 16 * 17 * sizeof(float);
#define EAMD_LQB(KQ_) hipLaunchKernelGGL(lstm_seq_bwd_kernel<KQ_>, grid, block, lds, s, a)
  switch (kq) {
    case 1: EAMD_LQB(1); break;
    case 2: EAMD_LQB(2); break;
    case 4: EAMD_LQB(4); break;
    case 8: EAMD_LQB(8); break;
    default: EAMD_LQB(16); break;
  }
#undef EAMD_LQB
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

namespace {
__global__ void lstm_seq_status_merge_kernel(const unsigned* __restrict__ ws, unsigned* __restrict__ sticky) {
  const unsigned v = ws[SYNC_TMO_WORD];
  if (v != 0 && sticky[0] == 0) sticky[0] = v;      // the first give-up since the caller cleared the word
}
}  // namespace

/* folds the status word of the launch that used sync_ws into a caller-owned int32 device word (kept non-zero until the caller
 * clears it): call it behind every eamd_lstm_seq_fwd / _bwd on the same stream and read the word where the host synchronises
 * anyway (once per epoch / step log) - a launch that gave up must not only show as a skipped optimizer step */
int eamd_lstm_seq_status_merge(const void* sync_ws, void* sticky, void* stream) {
  if (!sync_ws || !sticky) return EAMD_EINVAL;
  hipLaunchKernelGGL(lstm_seq_status_merge_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const unsigned*)sync_ws,
                     (unsigned*)sticky);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

/* the status word of the last launch that used sync_ws, copied to the host (synchronises the stream): 0 = every wait
 * completed; otherwise the code of the first wait that gave up (the launch drained, its outputs are unusable) */
int eamd_lstm_seq_status(const void* sync_ws, void* stream) {
  if (!sync_ws) return EAMD_EINVAL;
  unsigned v = 0;
  if (hipMemcpyAsync(&v, sync_ws, 4, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return EAMD_EINVAL;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return EAMD_EINVAL;
  return (int)v;
}

}  // extern "C"
