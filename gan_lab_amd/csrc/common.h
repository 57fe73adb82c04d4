// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  Wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ganlab_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// hipGetLastError() reports the last error of ANY runtime call on this thread (e.g. a
// hipErrorNotReady left behind by an event query in the caller's allocator), so it is cleared right
// before each launch and sampled right after it; the per-thread status is collected by
// GL_CHECK_LAUNCH() at the end of the entry point.
static thread_local int gl_launch_status = GANLAB_OK;
// Diagnostic record of this thread's most recent launch (symbol + workgroup count), read back by
// ganlab_last_launch(): bench.py names the kernel it prices from what was actually dispatched.
extern thread_local const void* gl_last_kernel_fn;
extern thread_local unsigned gl_last_grid;
// ... and of the GL_LAUNCH_RING launches before it (ganlab_launch_count / ganlab_launch_history): an entry point may
// launch several kernels (a weight gradient and its slot reduction), a test that asserts WHICH kernels a step dispatched
// reads every launch between two counter values
#define GL_LAUNCH_RING 16
extern thread_local const void* gl_ring_fn[GL_LAUNCH_RING];
extern thread_local unsigned gl_ring_grid[GL_LAUNCH_RING];
extern thread_local unsigned long long gl_launch_count;
#define GL_LAUNCH(k, grid, ...)                                                  \
  do {                                                                           \
    (void)hipGetLastError();                                                     \
    const dim3 gl_grid_ = (grid);                                                \
    gl_last_kernel_fn = reinterpret_cast<const void*>(k);                        \
    gl_last_grid = gl_grid_.x * gl_grid_.y * gl_grid_.z;                         \
    gl_ring_fn[gl_launch_count % GL_LAUNCH_RING] = gl_last_kernel_fn;            \
    gl_ring_grid[gl_launch_count % GL_LAUNCH_RING] = gl_last_grid;               \
    ++gl_launch_count;                                                           \
    hipLaunchKernelGGL(k, gl_grid_, __VA_ARGS__);                                \
    if (hipGetLastError() != hipSuccess) gl_launch_status = GANLAB_ELAUNCH;      \
  } while (0)
static inline int gl_take_status() {
  const int s = gl_launch_status;
  gl_launch_status = GANLAB_OK;
  return s;
}
#define GL_CHECK_LAUNCH() gl_take_status()

// Environment knobs (A/B switches and tuning knobs, DESIGN.md "knobs") are read ONCE per process, at their first use - the
// entry points keep no other global mutable state.  gl_env_str: the string (or nullptr), cached per call site.
#include <cstdlib>
#define GL_ENV_ONCE(name) ([]() -> const char* { static const char* const v = getenv(name); return v; }())

static inline hipStream_t gl_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Blocks b and b+8 share an XCD (round-robin dispatch, MI355X_MICROARCH.md "Workgroup dispatch"):
// hand each XCD a contiguous chunk of the logical tile space so tiles that share operand panels hit
// the same per-XCD L2.  Bijective for any grid size (cdna_hip_programming.md T1).  Speed only.
__device__ __forceinline__ int gl_xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7, xcd = b & 7, i = b >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + i;
}

__device__ __forceinline__ float gl_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum over a 256-thread block; result valid in every thread.  `red` = 4 floats of LDS.
__device__ __forceinline__ float gl_block_sum_256(float v, float* red) {
  v = gl_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// fp64 variant (`red` = 4 doubles): long per-channel sums - bias / noise-weight gradients are sums over N*H*W terms that
// largely cancel - keep their partials in double from the thread accumulator to the fixed-order finish
__device__ __forceinline__ double gl_block_sum_256d(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// Accumulation chains.  An fp32 MFMA accumulator is ONE k-ordered fmaf chain per output (bit for bit): over a 4608-term
// contraction its rounding error grows like sqrt(K) and the thick kernels were 2-2.5x further from float64 than ATen's
// blocked sums (tools/op_error_probe.py).  The kernels with more than one K-chunk therefore keep a SECOND accumulator set:
// every GL_ACC_DUMP_TERMS products the running chain is added into it and restarts from zero - chains of <= 144 terms
// summed by a second chain of K / 144 partials, the error of a single chain of ~(144 + K / 144) terms.  -DGL_ACC_DUMP=0
// builds the single-chain kernels (A/B: tools/acc_dump_ab.sh).
#ifndef GL_ACC_DUMP
#define GL_ACC_DUMP 1
#endif
#ifndef GL_ACC_DUMP_TERMS_V
#define GL_ACC_DUMP_TERMS_V 144
#endif
constexpr int GL_ACC_DUMP_TERMS = GL_ACC_DUMP_TERMS_V;

__device__ __forceinline__ float gl_lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// One output channel `co` of a 1x1 convolution with <= 4 input channels (fromRGB forward / toRGB input gradient), 4 pixels
// at a time: start + sum_ci wp[ci * Cout_p + co] * xv[ci], channels in order.  Shared by pw_few_to_many_kernel (conv.hip)
// and the InstanceNorm backward kernels that recompute toRGB's input gradient instead of reading it (pointwise.hip), so
// both produce the same bits.
__device__ __forceinline__ float4 gl_few_dot(const float4 (&xv)[4], const float* __restrict__ wp, int Cin, int Cout_p, int co,
                                             float start) {
  float4 a = float4{start, start, start, start};
#pragma unroll
  for (int ci = 0; ci < 4; ++ci) {
    const float w = ci < Cin ? wp[(long long)ci * Cout_p + co] : 0.f;
    a.x += w * xv[ci].x; a.y += w * xv[ci].y; a.z += w * xv[ci].z; a.w += w * xv[ci].w;
  }
  return a;
}

// ---- conv_x3.hip packed weights: [co tile 64][k-step][plane 3][k-group 4][co 64][8] bf16, three planes that sum to the fp32
// weight exactly.  k-step s of a 32-channel chunk: lane groups 0,1 = (tap_lo(s), half_lo(s)), 2,3 = (tap_hi(s), half_hi(s)) with
//   steps 0-3: taps (0,1) (2,3) (4,5) (6,7) of half 0 | step 4: tap 8 of half 0 and of half 1 | steps 5-8: half 1.
// gl_x3_pack_position writes the 9 taps x 3 planes of GEMM position (row ci, column co) - shared by the single-weight
// kernel (conv_x3.hip) and the batched re-pack (pack.hip), so both produce the same bits.  w9: the position's 9 taps in
// GEMM order (already flipped for the input gradient), scaled here.
__device__ __forceinline__ void gl_x3_pack_position(const float* __restrict__ w9, bool flip, float scale, __bf16* __restrict__ out,
                                                    int CI, int ci, int co) {
  const int steps = CI / 32 * 9;
  const int ct = co >> 6, col = co & 63, c = ci >> 5, half = (ci >> 4) & 1, g = (ci >> 3) & 1, j = ci & 7;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int s = t < 8 ? (half ? 5 : 0) + (t >> 1) : 4;
    const int hi = t < 8 ? (t & 1) : half;
    const int kg = hi * 2 + g;
    float v = w9[flip ? 8 - t : t] * scale;
    asm volatile("" : "+v"(v));      // the ROUNDED product is what is split: no contraction of this multiply into v - h below
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    __bf16* base = out + (((long long)ct * steps + c * 9 + s) * (3 * 4 * 64)) * 8;
    base[((0 * 4 + kg) * 64 + col) * 8 + j] = h;
    base[((1 * 4 + kg) * 64 + col) * 8 + j] = m;
    base[((2 * 4 + kg) * 64 + col) * 8 + j] = l;
  }
}

// the transposed 4x4 stride-2 form (conv_x3_up.hip): [co tile 64][row parity py][k-step = ci / 8][plane 3][tap kg = ty * 2 + tx][column
// parity px][co 64][8]; tap (ty, tx) of parity (py, px) sums the 3x3 weights of rows R(py, ty) x columns R(px, tx): up forward
// R(0,0) = {0}, R(0,1) = {1,2}, R(1,0) = {0,1}, R(1,1) = {2}; the input gradient of the pooled conv takes the flipped sets
// (k -> 2 - k) and 1/2 per axis.  `up`: 1 = forward of the up layer (GEMM rows = Cin), 0 = input gradient of the pooled layer.
// An output-channel count that is not a multiple of 64 (32: the 64 -> 32 layers) takes the 32-channel layout: a workgroup covers
// BOTH row parities of 32 channels and an image's 64 columns are [py][co 32] - [co tile 32][k-step][plane][tap][px][py * 32 + co][8].
__device__ __forceinline__ void gl_x3_up_pack_position(const float* __restrict__ w9, int up, float scale, __bf16* __restrict__ out,
                                                       int CI, int CO, int ci, int co) {
  const int nsteps = CI / 8;
  const bool c32 = (CO & 63) != 0;
  const int ct = c32 ? co >> 5 : co >> 6, j = ci & 7;
#pragma unroll
  for (int py = 0; py < 2; ++py)
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int ty = 0; ty < 2; ++ty)
#pragma unroll
        for (int tx = 0; tx < 2; ++tx) {
          float v = 0.f;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const int ry = up ? ky : 2 - ky, rx = up ? kx : 2 - kx;       // position in the up-forward row / column sets
              const bool iny = py == 0 ? (ty == 0 ? ry == 0 : ry >= 1) : (ty == 0 ? ry <= 1 : ry == 2);
              const bool inx = px == 0 ? (tx == 0 ? rx == 0 : rx >= 1) : (tx == 0 ? rx <= 1 : rx == 2);
              if (iny && inx) v += w9[ky * 3 + kx];
            }
          v = v * (up ? scale : 0.25f * scale);
          asm volatile("" : "+v"(v));
          const __bf16 h = (__bf16)v;
          const float r1 = v - (float)h;
          const __bf16 mm = (__bf16)r1;
          const __bf16 l = (__bf16)(r1 - (float)mm);
          __bf16* base = out + (((c32 ? (long long)ct : (long long)ct * 2 + py) * nsteps + (ci >> 3)) * (3 * 4 * 128)) * 8;
          const int col = c32 ? py * 32 + (co & 31) : co & 63;
          const int kgq = ty * 2 + tx;
          base[((0 * 4 + kgq) * 128 + px * 64 + col) * 8 + j] = h;
          base[((1 * 4 + kgq) * 128 + px * 64 + col) * 8 + j] = mm;
          base[((2 * 4 + kgq) * 128 + px * 64 + col) * 8 + j] = l;
        }
}

// the 4x3 combination matrices of the stride-2 fused layers (conv_s2.hip): K4 = M W M^T
__device__ __forceinline__ float gl_comb_s2(int up, int a, int k) {
  if (up) {
    const int lo = (a == 0) ? 2 : (a == 1 ? 1 : 0), hi = (a == 0) ? 2 : (a == 1 ? 2 : (a == 2 ? 1 : 0));
    return (k >= lo && k <= hi) ? 1.f : 0.f;
  }
  const int lo = (a <= 1) ? 0 : (a == 2 ? 1 : 2), hi = (a == 0) ? 0 : (a == 1 ? 1 : 2);
  return (k >= lo && k <= hi) ? 0.5f : 0.f;
}
// the strided 4x4 stride-2 form of conv_x3_down.hip: [co tile 128][k-step = (ci / 8) * 4 + a][plane 3][tap column b][co 128][8];
// K4[a][b] formed in fp32 exactly as pack.hip's stride-2 branch forms it (same operands as the exact-fp32 S kernel)
__device__ __forceinline__ void gl_x3_down_pack_position(const float* __restrict__ w9, int up, float scale, __bf16* __restrict__ out,
                                                         int CI, int ci, int co) {
  const int nsteps = CI / 8 * 4;
  const int ct = co >> 7, col = co & 127, q = ci >> 3, j = ci & 7;
  float k9[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) k9[t] = w9[t];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float v = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) v += gl_comb_s2(up, a, ky) * gl_comb_s2(up, b, kx) * k9[ky * 3 + kx];
      v = v * scale;
      asm volatile("" : "+v"(v));
      const __bf16 h = (__bf16)v;
      const float r1 = v - (float)h;
      const __bf16 mm = (__bf16)r1;
      const __bf16 l = (__bf16)(r1 - (float)mm);
      __bf16* base = out + (((long long)ct * nsteps + q * 4 + a) * (3 * 4 * 128)) * 8;
      base[((0 * 4 + b) * 128 + col) * 8 + j] = h;
      base[((1 * 4 + b) * 128 + col) * 8 + j] = mm;
      base[((2 * 4 + b) * 128 + col) * 8 + j] = l;
    }
}

// ---- conv.hip: mean / rstd of every (n, c) plane from per-tile sums of y and y^2 (fixed order, fp64) ---------------------------
extern "C" int gl_tail_stats_finish(const double* spart, float* mean, float* rstd, long long planes, int chunks, double inv_hw, float eps,
                         hipStream_t st);

// ---- wgrad_roll.hip: rolling-window weight gradient of the thin 3x3 layers (used by ganlab_conv_wgrad_f32) ----------
bool gl_wgrad_roll_supported(int N, int Cin, int Cout, int H, int W, int ks, int pad, int up, const void* x,
                             const void* gy);
int gl_wgrad_roll_slots(int N, int Cin, int Cout, int H, int W);
int gl_wgrad_roll_launch(const float* x, const float* gy, float* part, int N, int Cin, int Cout, int H, int W,
                         hipStream_t st, const float* aff_s = nullptr, const float* aff_t = nullptr);
bool gl_wgrad_s2_roll_supported(int N, int Cl, int Ch, int Hl, int Wl, const void* low, const void* high);
int gl_wgrad_s2_roll_slots(int N, int Cl, int Ch, int Hl, int Wl);
int gl_wgrad_s2_roll_launch(const float* low, const float* high, float* part, int N, int Cl, int Ch, int Hl, int Wl,
                            hipStream_t st, const float* aff_s = nullptr, const float* aff_t = nullptr);

// ---- conv_s2_roll.hip: rolling-window S / T kernels of the thin (16 <-> 32 channel) stride-2 fused layers ----------------
bool gl_s2_roll_supported(int is_T, int N, int Cin, int Cout, int Hl, int Wl, const void* x, const void* y);
int gl_s2_roll_launch(int is_T, const float* x, const float* wp, const float* bias, float* y, int N, int Cin, int Cout,
                      int Hl, int Wl, int Cin_p, int Cout_p, float bias_scale, int act, float slope, hipStream_t st,
                      const float* aff_s = nullptr, const float* aff_t = nullptr);

// ---- conv_roll_blur.hip: the thin rolling 3x3 kernel in the wave-owns-a-column-block layout; blur = 1: conv + bias + LeakyReLU
//      + binomial blur (+ sign bits), blur = 0: conv + bias + act ------------------------------------------------------------
bool gl_roll_blur_supported(int N, int Cin, int Cout, int H, int W, const void* x, const void* y);
int gl_roll_blur_launch(const float* x, const float* wp, const float* bias, float* y, void* bits, int N, int Cin, int Cout,
                        int H, int W, int Cin_p, int Cout_p, float bias_scale, float slope, hipStream_t st, int blur, int act);
