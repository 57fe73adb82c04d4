// ONE launch for every weight re-layout of a network (launch count): after an optimiser step rewrites a
// parameter arena, all packed forms the training step uses - [tap][ci][co] forward / input-gradient layouts of conv.hip,
// the 16-tap K4 layouts of conv_s2.hip, the bf16 layouts of conv_bf16.hip; 60-190 small launches per step before - are
// rebuilt from a device-resident descriptor table: block b finds its descriptor by binary search over the block
// offsets and converts 256 consecutive output elements with the SAME element formula as the single-weight kernels
// (pack_kernel, pack_s2_kernel, pack_bf16_kernel), so the results are bit-identical.  The table is built once per set of
// weights (the pointers are stable: parameters live in flat arenas) by gan_lab_amd/ops.py.
#include "common.h"

namespace {

constexpr int round_up_c(int v, int m) { return (v + m - 1) / m * m; }

__device__ __forceinline__ float comb_s2(int up, int a, int k) {          // conv_s2.hip: the 4x3 combination matrices
  if (up) {
    const int lo = (a == 0) ? 2 : (a == 1 ? 1 : 0), hi = (a == 0) ? 2 : (a == 1 ? 2 : (a == 2 ? 1 : 0));
    return (k >= lo && k <= hi) ? 1.f : 0.f;
  }
  const int lo = (a <= 1) ? 0 : (a == 2 ? 1 : 2), hi = (a == 0) ? 0 : (a == 1 ? 1 : 2);
  return (k >= lo && k <= hi) ? 0.5f : 0.f;
}

__global__ __launch_bounds__(256) void pack_many_kernel(const ganlab_pack_desc* __restrict__ descs, int n_desc) {
  // binary search: last descriptor whose first block is <= blockIdx.x
  int lo = 0, hi = n_desc - 1;
  const long long b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].block0 <= b) lo = mid; else hi = mid - 1;
  }
  const ganlab_pack_desc d = descs[lo];
  // One thread per WEIGHT POSITION (all of its taps), not per output element: the taps of (co, ci) are 36 contiguous bytes
  // of the source, read once here instead of by 9 (plain), 16 (stride-2) or 9 (bf16) threads of far-apart workgroups; the
  // stores of a tap stay coalesced across the threads (adjacent threads = adjacent output columns).
  const long long e = (b - d.block0) * 256 + threadIdx.x;
  const float* w = d.src;
  if (d.kind == GANLAB_PACKKIND_PLAIN) {
    const int KK = d.ks * d.ks, dgrad = d.mode == GANLAB_PACK_DGRAD;
    const int rows = dgrad ? d.Cout : d.Cin, cols = dgrad ? d.Cin : d.Cout;
    const int rows_p = round_up_c(rows, d.ks == 1 ? 32 : 16), cols_p = round_up_c(cols, 64);
    const long long plane = (long long)rows_p * cols_p;
    if (e >= plane) return;
    const int col = (int)(e % cols_p), row = (int)(e / cols_p);
    const bool in = row < rows && col < cols;
    const int co = dgrad ? row : col, ci = dgrad ? col : row;
    const float* ws = w + ((long long)co * d.Cin + ci) * KK;
    float* out = reinterpret_cast<float*>(d.dst) + e;
    for (int tap = 0; tap < KK; ++tap)
      out[tap * plane] = in ? d.scale * ws[dgrad ? (KK - 1 - tap) : tap] : 0.f;
  } else if (d.kind == GANLAB_PACKKIND_S2) {
    const int transpose = d.mode;
    const int rows = transpose ? d.Cout : d.Cin, cols = transpose ? d.Cin : d.Cout;
    const int rows_p = round_up_c(rows, 16), cols_p = round_up_c(cols, 64);
    const long long plane = (long long)rows_p * cols_p;
    if (e >= plane) return;
    const int col = (int)(e % cols_p), row = (int)(e / cols_p);
    const bool in = row < rows && col < cols;
    const int co = transpose ? row : col, ci = transpose ? col : row;
    float k9[9];
    const float* ws = w + ((long long)co * d.Cin + ci) * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) k9[t] = in ? ws[t] : 0.f;
    float* out = reinterpret_cast<float*>(d.dst) + e;
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) {
      const int a = tap >> 2, bb = tap & 3;
      float v = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) v += comb_s2(d.up, a, ky) * comb_s2(d.up, bb, kx) * k9[ky * 3 + kx];
      out[tap * plane] = v * d.scale;
    }
  } else if (d.kind == GANLAB_PACKKIND_X3 && d.ks == 4) {      // conv_x3.hip, transposed stride-2 form (d.up: 1 = up layer's
    const int up = d.up;                                         // forward weights, 0 = pooled layer's input-gradient weights)
    const int CO = up ? d.Cout : d.Cin, CI = up ? d.Cin : d.Cout;
    if (e >= (long long)CO * CI) return;
    int ci, co;
    if ((CO & 63) == 0) {
      const int col = (int)(e & 63);
      const long long t = e >> 6;
      ci = (int)(t % CI);
      co = (int)(t / CI) * 64 + col;
    } else {
      co = (int)(e % CO);
      ci = (int)(e / CO);
    }
    const float* w9 = up ? w + ((long long)co * d.Cin + ci) * 9 : w + ((long long)ci * d.Cin + co) * 9;
    gl_x3_up_pack_position(w9, up, d.scale, reinterpret_cast<__bf16*>(d.dst), CI, CO, ci, co);
  } else if (d.kind == GANLAB_PACKKIND_X3 && d.ks == 5) {      // conv_x3_down.hip, strided stride-2 form (d.up: 0 = pooled layer's
    const int up = d.up;                                         // forward weights, 1 = up layer's input-gradient weights)
    const int CO = up ? d.Cin : d.Cout, CI = up ? d.Cout : d.Cin;
    if (e >= (long long)CO * CI) return;
    const int col = (int)(e & 127);
    const long long t = e >> 7;
    const int ci = (int)(t % CI), ct = (int)(t / CI);
    const int co = ct * 128 + col;
    const float* w9 = up ? w + ((long long)ci * d.Cin + co) * 9 : w + ((long long)co * d.Cin + ci) * 9;
    gl_x3_down_pack_position(w9, up, d.scale, reinterpret_cast<__bf16*>(d.dst), CI, ci, co);
  } else if (d.kind == GANLAB_PACKKIND_X3) {      // conv_x3.hip: three bf16 planes per weight, k-step images
    const bool dg = d.mode == GANLAB_PACK_DGRAD;
    const int CO = dg ? d.Cin : d.Cout, CI = dg ? d.Cout : d.Cin;
    if (e >= (long long)CO * CI) return;
    const int col = (int)(e & 63);
    const long long t = e >> 6;
    const int ci = (int)(t % CI), ct = (int)(t / CI);
    const int co = ct * 64 + col;
    const float* w9 = dg ? w + ((long long)ci * d.Cin + co) * 9 : w + ((long long)co * d.Cin + ci) * 9;
    gl_x3_pack_position(w9, dg, d.scale, reinterpret_cast<__bf16*>(d.dst), CI, ci, co);
  } else {      // GANLAB_PACKKIND_BF16: [chunk = ci/32][tap][kg = (ci%32)/8][CO][ci%8]
    const int dg = d.mode == GANLAB_PACK_DGRAD;
    const int CO = dg ? d.Cin : d.Cout;
    if (e >= d.total / 9) return;
    const int j = (int)(e & 7);
    long long t = e >> 3;
    const int co = (int)(t % CO);
    t /= CO;
    const int kg = (int)(t & 3);
    const int chunk = (int)(t >> 2);
    const int ci = chunk * 32 + kg * 8 + j;
    const float* ws = dg ? w + ((long long)ci * d.Cin + co) * 9 : w + ((long long)co * d.Cin + ci) * 9;
    __bf16* out = reinterpret_cast<__bf16*>(d.dst);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
      out[((((long long)chunk * 9 + tap) * 4 + kg) * CO + co) * 8 + j] = (__bf16)(ws[dg ? 8 - tap : tap] * d.scale);
  }
}

}  // namespace

extern "C" {

int ganlab_pack_desc_size(void) { return (int)sizeof(ganlab_pack_desc); }

int ganlab_pack_many(const ganlab_pack_desc* descs_device, int n_desc, long long total_blocks, void* stream) {
  if (!descs_device || n_desc <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffLL) return GANLAB_EINVAL;
  GL_LAUNCH(pack_many_kernel, dim3((unsigned)total_blocks), dim3(256), 0, gl_stream(stream), descs_device, n_desc);
  return GL_CHECK_LAUNCH();
}

}  // extern "C"
