// fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32), NCHW.
//
// One kernel family serves Conv2dEx / LinearEx forward, their input gradient (same kernel, flipped
// + transposed packed weights) and their weight gradient (separate kernel, split over pixels):
// reference math = F.conv2d(x * wscale, W, padding) at utils/custom_layers.py:202-211 and
// F.linear at :282-291, plus the nearest 2x upsample in front of generator convs
// (stylegan/architectures.py:292-334) which is folded into the tile staging (never materialised).
//
// GEMM view (forward):  D[co][p] = sum_{tap,ci} Wp[tap][ci][co] * Xv[n(p), ci, oy(p)+ky-pad, ox(p)+kx-pad]
//   MFMA 16x16x4 f32:  A[i=co][k=ci]  (lane l holds A[l&15][l>>4])
//                      B[k=ci][j=px]  (lane l holds B[l>>4][l&15])
//                      D[i][j]        (lane l holds rows (l>>4)*4+r, r=0..3, column l&15)
//   so every lane ends up with 4 consecutive output channels of ONE pixel and the 16 lanes of a
//   quarter-wave cover 16 consecutive pixels -> 64-byte coalesced NCHW stores.
// A workgroup (256 threads = 4 waves, one per SIMD) owns CO_T channels x PX_T pixels (an NI x TH x TW
// patch).  Per K-chunk of CI_T input channels the halo'd activation patch and the [tap][ci][co]
// weight slab go global -> registers -> LDS; the registers of chunk k+1 are loaded BEFORE the MFMA
// phase of chunk k (software prefetch: HBM/L2 latency hides under 72..288 MFMAs per wave) and
// every staging access is 16 bytes wide where the shape allows:
//   XMODE 1 (VEC)   : rows [ox0-4, ox0+TW+4) as aligned float4 (W % 4 == 0, stride-1 source)
//   XMODE 2 (VECUP) : low-res rows as aligned float4, duplicated 2x2 while being written to LDS
//   XMODE 0         : per-element path for ragged / tiny shapes (4x4, 8x8, linears, odd sizes), same register
//                     prefetch (one float per item)
// LDS strides are padded so the two k-halves of a 32-lane read group land on disjoint banks.
#include "common.h"

#include <cxxabi.h>
#include <stdlib.h>
#include <string.h>

thread_local const void* gl_last_kernel_fn = nullptr;
thread_local unsigned gl_last_grid = 0;
thread_local const void* gl_ring_fn[GL_LAUNCH_RING] = {};
thread_local unsigned gl_ring_grid[GL_LAUNCH_RING] = {};
thread_local unsigned long long gl_launch_count = 0;

namespace {

constexpr int round_up_c(int v, int m) { return (v + m - 1) / m * m; }
constexpr int ceil_div_c(int a, int b) { return (a + b - 1) / b; }
// smallest s >= v with s % 32 == r
constexpr int pad_mod32(int v, int r) { return v + ((r - (v % 32)) + 32) % 32; }

enum { XSCALAR = 0, XVEC = 1, XVECUP = 2 };

// ---- activation-patch geometry shared by the forward and the weight-gradient kernels ----------------
template <int KS_, int TWL_, int THL_, int NIL_, int XMODE_>
struct PatchGeo {
  static constexpr int KS = KS_, KK = KS_ * KS_, XMODE = XMODE_;
  static constexpr int TWL = TWL_, THL = THL_, NIL = NIL_;
  static constexpr int TW = 1 << TWL_, TH = 1 << THL_, NI = 1 << NIL_;
  static constexpr int PX_T = TW * TH * NI;
  static constexpr int PADC = (KS_ - 1) / 2;                       // "same" padding (vector modes)
  static constexpr int LP = XMODE_ == XVECUP ? 8 : (XMODE_ == XVEC ? (KS_ == 3 ? 4 : 0) : 0);
  static constexpr int R = TH + KS_ - 1;
  static constexpr int RP = XMODE_ == XSCALAR ? TW + KS_ - 1 : TW + 2 * LP;   // LDS row pitch
  static constexpr int XOFF = XMODE_ == XSCALAR ? 0 : LP - PADC;   // column of tap kx=0 for tx=0
  static constexpr int IMG = R * RP;
  // vector staging items (float4) per input channel
  static constexpr int ROW4 = RP / 4;                // XVEC: float4 per patch row
  static constexpr int LR = TH / 2 + 2;              // XVECUP: low-res rows per patch
  static constexpr int LROW4 = RP / 8;               // XVECUP: float4 per low-res row
  static constexpr int ITEMS_PER_CI =
      XMODE_ == XSCALAR ? NI * IMG : (XMODE_ == XVEC ? NI * R * ROW4 : NI * LR * LROW4);
  static_assert(XMODE_ == XSCALAR || (TW >= 8 && NI == 1), "vector staging needs TW >= 8, one image per tile");
};

struct PatchArgs {
  const float* x;
  int N, Cin, Hi, Wi, Hv, Wv, pad, up;
  // "deferred InstanceNorm" (ops.Deferred): the tensor this patch is cut from is a generator layer's activated output a
  // whose InstanceNorm + style, b = a * s[n,ci] + t[n,ci], has not been written out - the AFF kernels apply it while the
  // patch goes from the prefetch registers to LDS (elements in the zero padding of b stay zero).  [N][Cin] each, or null.
  const float* aff_s;
  const float* aff_t;
};

// Descriptor of one staging item: LDS offset + channel, and the global offset relative to
// x + n0*Cin*Hi*Wi (fits an int: NI*Cin*Hi*Wi < 2^31), or -1 when it is padding.
template <class G, int CI_T, int PLANE>
struct XStage {
  static constexpr int NITEMS = CI_T * G::ITEMS_PER_CI;
  static constexpr int PT = ceil_div_c(NITEMS, 256);
  int goff[PT];
  int loff[PT];  // LDS offset | (ci << 20); for XVECUP bit 30/31 = write row 2lr-1 / 2lr
  int n0_, oy0_, ox0_;  // scalar mode of the weight-gradient kernel: tile origin (items re-derived while staging)

  __device__ __forceinline__ void init(const PatchArgs& p, int tid, int n0, int oy0, int ox0) {
    const int plane = p.Hi * p.Wi;
    n0_ = n0; oy0_ = oy0; ox0_ = ox0;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int e = tid + i * 256;
      int g = -1, l = 0;
      if (G::XMODE == XSCALAR) {   // one element per item: (ci, image, row, column) of the halo'd patch
        const int c = e % G::RP;
        int t = e / G::RP;
        const int r = t % G::R;
        t /= G::R;
        const int ni = t % G::NI, ci = t / G::NI;
        const int vy = oy0 + r - p.pad, vx = ox0 + c - p.pad;
        l = (ci * PLANE + ni * G::IMG + r * G::RP + c) | (ci << 20);
        if (e < NITEMS && n0 + ni < p.N && (unsigned)vy < (unsigned)p.Hv && (unsigned)vx < (unsigned)p.Wv)
          g = (ni * p.Cin + ci) * plane + (p.up ? (vy >> 1) : vy) * p.Wi + (p.up ? (vx >> 1) : vx);
      } else if (G::XMODE == XVEC) {
        const int q = e % G::ROW4;
        int t = e / G::ROW4;
        const int r = t % G::R, ci = t / G::R;
        const int vy = oy0 + r - G::PADC, vx = ox0 - G::LP + 4 * q;
        l = (ci * PLANE + r * G::RP + 4 * q) | (ci << 20);
        if (e < NITEMS && n0 < p.N && (unsigned)vy < (unsigned)p.Hi && (unsigned)vx < (unsigned)p.Wi)
          g = ci * plane + vy * p.Wi + vx;
      } else {  // XVECUP: item = one low-res float4; expands to virtual rows 2lr-1, 2lr and 8 columns
        const int q = e % G::LROW4;
        int t = e / G::LROW4;
        const int lr = t % G::LR, ci = t / G::LR;
        const int ly = (oy0 >> 1) - 1 + lr, lx = ((ox0 - G::LP) >> 1) + 4 * q;
        l = (ci * PLANE + (2 * lr) * G::RP + 8 * q) | (ci << 20);   // row 2lr; row 2lr-1 is one pitch above
        if (lr > 0) l |= (1 << 30);
        if (2 * lr < G::R) l |= (1 << 31);
        if (e < NITEMS && n0 < p.N && (unsigned)ly < (unsigned)p.Hi && (unsigned)lx < (unsigned)p.Wi)
          g = ci * plane + ly * p.Wi + lx;
        else if (e >= NITEMS)
          l &= ~((1 << 30) | (1 << 31));
      }
      goff[i] = g;
      loff[i] = l;
    }
  }

  // Strip variant (forward kernel walks several tiles along x): goff holds ci*plane + vy*Wi (or -1 when the
  // item is padding in y / n / past the item count) and xq the item's column relative to the tile origin
  // (in units of the source: low-res columns for XVECUP); the x coordinate and its bounds test are applied
  // per tile in x_load_strip.
  __device__ __forceinline__ void init_strip(const PatchArgs& p, int tid, int n0, int oy0) {
    const int plane = p.Hi * p.Wi;
    n0_ = n0; oy0_ = oy0; ox0_ = 0;
    if (G::XMODE == XSCALAR) return;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int e = tid + i * 256;
      int g = -1, l = 0;
      if (G::XMODE == XVEC) {
        const int q = e % G::ROW4;
        int t = e / G::ROW4;
        const int r = t % G::R, ci = t / G::R;
        const int vy = oy0 + r - G::PADC;
        l = (ci * PLANE + r * G::RP + 4 * q) | (ci << 20);
        if (e < NITEMS && n0 < p.N && (unsigned)vy < (unsigned)p.Hi) g = ci * plane + vy * p.Wi;
      } else {
        const int q = e % G::LROW4;
        int t = e / G::LROW4;
        const int lr = t % G::LR, ci = t / G::LR;
        const int ly = (oy0 >> 1) - 1 + lr;
        l = (ci * PLANE + (2 * lr) * G::RP + 8 * q) | (ci << 20);
        if (lr > 0) l |= (1 << 30);
        if (2 * lr < G::R) l |= (1 << 31);
        if (e < NITEMS && n0 < p.N && (unsigned)ly < (unsigned)p.Hi) g = ci * plane + ly * p.Wi;
        else if (e >= NITEMS) l &= ~((1 << 30) | (1 << 31));
      }
      goff[i] = g;
      loff[i] = l;
    }
  }
  // column of item i relative to the tile origin, in source units (re-derived: cheaper than 7 live registers)
  static __device__ __forceinline__ int xq_of(int tid, int i) {
    const int e = tid + i * 256;
    return G::XMODE == XVEC ? 4 * (e % G::ROW4) - G::LP : 4 * (e % G::LROW4) - G::LP / 2;
  }
};

// registers holding one staged chunk of the activation patch
template <int XMODE>
struct XElem { typedef float4 type; };
template <>
struct XElem<XSCALAR> { typedef float type; };
template <class G, int PT>
struct XRegs {
  typename XElem<G::XMODE>::type v[PT];
};
__device__ __forceinline__ void x_zero(float4& v) { v = float4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ void x_zero(float& v) { v = 0.f; }
__device__ __forceinline__ void x_ld(float4& v, const float* p) { v = *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void x_ld(float& v, const float* p) { v = *p; }

template <class G, int CI_T, int PLANE>
__device__ __forceinline__ void x_load(XRegs<G, XStage<G, CI_T, PLANE>::PT>& r, const XStage<G, CI_T, PLANE>& st,
                                       const float* xb, int ci0, int Cin, int plane) {
  constexpr int PT = XStage<G, CI_T, PLANE>::PT;
  const float* src = xb + (long long)ci0 * plane;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int ci = (st.loff[i] >> 20) & 0x3ff;
    const bool ok = st.goff[i] >= 0 && ci0 + ci < Cin;
    x_zero(r.v[i]);
    if (ok) x_ld(r.v[i], src + st.goff[i]);
  }
}

// strip variant: `xbase` = tile origin column in source units (ox0, or ox0/2 for the folded upsample)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// the same through a buffer descriptor over ONE image's Cin planes (vector mode, one image per tile): invalid items
// read as zero by the hardware's bounds check (offset 0x80000000) - no exec-mask branches around the loads
template <class G, int CI_T, int PLANE>
__device__ __forceinline__ void x_load_desc(XRegs<G, XStage<G, CI_T, PLANE>::PT>& r,
                                            const XStage<G, CI_T, PLANE>& st, __amdgpu_buffer_rsrc_t rsrc, int ci0,
                                            int Cin, int plane) {
  constexpr int PT = XStage<G, CI_T, PLANE>::PT;
  static_assert(G::XMODE == XVEC && G::NI == 1, "descriptor loads: plain vector staging, one image per tile");
  const int soff = ci0 * plane * 4;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int ci = (st.loff[i] >> 20) & 0x3ff;
    const int off = (st.goff[i] >= 0 && ci0 + ci < Cin) ? st.goff[i] * 4 : (int)0x80000000;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, soff, 0);
    r.v[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
  }
}
template <class G, int CI_T, int PLANE>
__device__ __forceinline__ void x_load_strip(XRegs<G, XStage<G, CI_T, PLANE>::PT>& r,
                                             const XStage<G, CI_T, PLANE>& st, const float* xb, int ci0, int Cin,
                                             int plane, int xbase, int Wi, int tid) {
  constexpr int PT = XStage<G, CI_T, PLANE>::PT;
  if (G::XMODE == XSCALAR) return;
  const float* src = xb + (long long)ci0 * plane;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int ci = (st.loff[i] >> 20) & 0x3ff;
    const int vx = xbase + XStage<G, CI_T, PLANE>::xq_of(tid, i);
    const bool ok = st.goff[i] >= 0 && ci0 + ci < Cin && (unsigned)vx < (unsigned)Wi;
    r.v[i] = ok ? *reinterpret_cast<const float4*>(src + st.goff[i] + vx) : float4{0.f, 0.f, 0.f, 0.f};
  }
}

// Buffer-descriptor variant: `rsrc` spans the Cin*plane floats of this workgroup's image, so channel padding
// (ci >= Cin) falls outside the descriptor and reads as 0 in hardware; rows / columns in the zero padding get an
// out-of-range offset.  No exec-mask branches: ~6 instructions per float4 instead of ~20.
template <class G, int CI_T, int PLANE>
__device__ __forceinline__ void x_load_buf(XRegs<G, XStage<G, CI_T, PLANE>::PT>& r,
                                           const XStage<G, CI_T, PLANE>& st, __amdgpu_buffer_rsrc_t rsrc, int ci0,
                                           int plane, int xbase, int Wi, int tid) {
  constexpr int PT = XStage<G, CI_T, PLANE>::PT;
  if (G::XMODE == XSCALAR) return;
  const int soff = ci0 * plane * 4;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int vx = xbase + XStage<G, CI_T, PLANE>::xq_of(tid, i);
    const bool ok = st.goff[i] >= 0 && (unsigned)vx < (unsigned)Wi;
    const int off = ok ? (st.goff[i] + vx) * 4 : (int)0x80000000;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, soff, 0);
    r.v[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
  }
}

// Per-element staging WITHOUT register prefetch, global -> LDS in batches of 8 independent loads: the weight-gradient
// kernel's ragged / tiny geometries (their 32-channel patches would not fit a register-prefetch copy).
template <class G, int CI_T, int PLANE>
__device__ __forceinline__ void x_stage_scalar(const XStage<G, CI_T, PLANE>& st, const PatchArgs& p,
                                               const float* xb, int ci0, float* Xs, int tid) {
  constexpr int NITEMS = CI_T * G::ITEMS_PER_CI;
  constexpr int BATCH = 8;
  const int plane = p.Hi * p.Wi;
  for (int base = 0; base < NITEMS; base += 256 * BATCH) {
    float v[BATCH];
    int l[BATCH];
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      const int e = base + tid + i * 256;
      const int c = e % G::RP;
      int t = e / G::RP;
      const int r = t % G::R;
      t /= G::R;
      const int ni = t % G::NI, ci = t / G::NI;
      const int vy = st.oy0_ + r - p.pad, vx = st.ox0_ + c - p.pad, n = st.n0_ + ni, cig = ci0 + ci;
      l[i] = e < NITEMS ? ci * PLANE + ni * G::IMG + r * G::RP + c : -1;
      v[i] = 0.f;
      if (e < NITEMS && cig < p.Cin && n < p.N && (unsigned)vy < (unsigned)p.Hv && (unsigned)vx < (unsigned)p.Wv) {
        const int iy = p.up ? (vy >> 1) : vy, ix = p.up ? (vx >> 1) : vx;
        v[i] = xb[((long long)ni * p.Cin + cig) * plane + iy * p.Wi + ix];
      }
    }
#pragma unroll
    for (int i = 0; i < BATCH; ++i)
      if (l[i] >= 0) Xs[l[i]] = v[i];
  }
}

// 16-byte LDS store, or two 8-byte ones when the plane stride is only 8-byte aligned (wgrad)
template <bool A16>
__device__ __forceinline__ void lds_store4(float* dst, float4 v) {
  if (A16) {
    *reinterpret_cast<float4*>(dst) = v;
  } else {
    *reinterpret_cast<float2*>(dst) = float2{v.x, v.y};
    *reinterpret_cast<float2*>(dst + 2) = float2{v.z, v.w};
  }
}

template <class G, int CI_T, int PLANE, bool A16, bool AFF = false>
__device__ __forceinline__ void x_store(const XRegs<G, XStage<G, CI_T, PLANE>::PT>& r,
                                        const XStage<G, CI_T, PLANE>& st, float* Xs, int tid,
                                        const float* aff_tab = nullptr, int tab_ci0 = 0, int tab_t = 0, int ci_left = 0) {
  constexpr int PT = XStage<G, CI_T, PLANE>::PT;
  constexpr int NITEMS = XStage<G, CI_T, PLANE>::NITEMS;
  static_assert(!AFF || G::XMODE == XVEC, "affine-on-load: plain vector staging only");
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int l = st.loff[i] & 0xfffff;
    if constexpr (G::XMODE == XSCALAR) {
      if (tid + i * 256 < NITEMS) Xs[l] = r.v[i];
    } else if constexpr (G::XMODE == XVEC) {
      if constexpr (AFF) {
        // b = a * s + t for the elements INSIDE the image (a whole float4 is inside or outside: W % 4 == 0); the loads
        // returned zeros for the others, which is what the zero padding of b holds.  aff_tab (LDS): s at [tab_ci0 + ci],
        // t at [tab_t + tab_ci0 + ci]; ci_left = channels left from this chunk's first one
        const int ci = (st.loff[i] >> 20) & 0x3ff;
        const bool ok = st.goff[i] >= 0 && ci < ci_left;
        // (not-ok lanes must not touch the table: uninitialised LDS may hold a NaN, and 0 * NaN is a NaN)
        const float sv = ok ? aff_tab[tab_ci0 + ci] : 0.f, tv = ok ? aff_tab[tab_t + tab_ci0 + ci] : 0.f;
        float4 v = r.v[i];
        v.x = fmaf(v.x, sv, tv); v.y = fmaf(v.y, sv, tv); v.z = fmaf(v.z, sv, tv); v.w = fmaf(v.w, sv, tv);
        if (tid + i * 256 < NITEMS) lds_store4<A16>(Xs + l, v);
      } else {
        if (tid + i * 256 < NITEMS) lds_store4<A16>(Xs + l, r.v[i]);
      }
    } else {
      const float4 v = r.v[i];
      const float4 a = float4{v.x, v.x, v.y, v.y}, b = float4{v.z, v.z, v.w, v.w};
      if (st.loff[i] & (1 << 30)) {   // virtual row 2lr-1
        lds_store4<A16>(Xs + l - G::RP, a);
        lds_store4<A16>(Xs + l - G::RP + 4, b);
      }
      if (st.loff[i] & (1 << 31)) {   // virtual row 2lr
        lds_store4<A16>(Xs + l, a);
        lds_store4<A16>(Xs + l + 4, b);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward / input-gradient kernel
// ------------------------------------------------------------------------------------------------
struct ConvArgs {
  PatchArgs in;
  const float* wp;
  const float* bias;
  float* y;
  int Cout, Ho, Wo;
  int Cin_p, Cout_p;   // packed-weight dims: wp[tap][Cin_p][Cout_p]
  int tiles_x, tiles_y, tiles_n, tiles_co;
  int strip, strips_x;   // each workgroup walks `strip` consecutive tiles along x
  float bias_scale, slope;
  int act;
  // output mask (general kernel only): y *= (mask > 0 ? 1 : mslope), mask shaped like y.  The input gradient of a conv
  // whose input is a LeakyReLU output comes out already multiplied by that LeakyReLU's derivative.
  const float* mask;
  float mslope;
  // split-K (scalar-staged small geometries only): workgroup (tile, ks) contracts input channels
  // [ks * ksplit_ci, (ks + 1) * ksplit_ci) and writes its raw partial sums to y + ks * ksplit_stride
  int ksplit, ksplit_ci;
  long long ksplit_stride;
  // TAIL (AFF tile kernels as a whole generator layer, ganlab_conv_fwd_aff_tail_f32): y = act(conv + noise_w[c] * noise[n,hw]
  // + bias) and the partial sums of y and y^2 of this workgroup's tile per (n, c): spart[(n*Cout + c)*chunks + tile][2]
  const float* noise;
  const float* noise_w;
  double* spart;
};

template <int KS_, int MB_, int TWL_, int THL_, int NIL_, int XMODE_>
struct FwdCfg {
  using G = PatchGeo<KS_, TWL_, THL_, NIL_, XMODE_>;
  static constexpr int KS = KS_, KK = KS_ * KS_, MB = MB_;
  static constexpr int WN = 4;  // 4 waves side by side along the pixel dim
  static constexpr int NB = G::PX_T / (16 * WN);
  static constexpr int CO_T = 16 * MB_;
  // scalar staging keeps 3 registers per patch element in flight (global offset, LDS offset, value): 8-channel
  // chunks bound that at ~55 registers for the 4x4x16-image geometry
#ifndef GL_THICK_CI_T       // 16-channel chunks for the two-workgroups-per-CU thick kernel (half the barriers, 72 KB of LDS):
#define GL_THICK_CI_T 8     // measured slower in round 4 (plain 64..512-channel layers: 5.90 against 5.79 ms), stays 8
#endif
  static constexpr int CI_T = (KS_ == 1) ? 32 : ((MB_ == 1 && XMODE_ != XSCALAR) ? 16 : ((MB_ == 4 && XMODE_ == XVEC && GL_ACC_DUMP) ? GL_THICK_CI_T : 8));
  static constexpr int PLANE = pad_mod32(G::NI * G::IMG, 16);
  static constexpr int COP = pad_mod32(CO_T, 16);
  static constexpr int XS = CI_T * PLANE, WS = KK * CI_T * COP;
  static constexpr int NWI = KK * CI_T * CO_T / 4;  // float4 weight items per chunk
  static constexpr int WPT = ceil_div_c(NWI, 256);
  static constexpr bool STRIP = MB_ == 1 && XMODE_ != XSCALAR;
  // chunks per accumulator dump (0: single chain - the 1x1 / linear layers contract <= 512 terms, split-K'd further)
  static constexpr int DUMP = (GL_ACC_DUMP && KS_ == 3) ? (GL_ACC_DUMP_TERMS / (KK * CI_T) > 0 ? GL_ACC_DUMP_TERMS / (KK * CI_T) : 1) : 0;
  // operand fragments of the next K-step requested before the MFMAs of this one (conv_fwd_kernel)
#ifndef GL_FRAG_PREFETCH
#define GL_FRAG_PREFETCH 1
#endif
  static_assert(NB >= 1, "pixel tile too small");
};

// The 64-channel tile of the vector-staged 3x3 layers (64 .. 512 channels at >= 16^2).  With the second accumulator set
// (GL_ACC_DUMP) a 64-channel x 256-pixel workgroup tile needs ~205 registers: two waves per SIMD.  Over 128 pixels (16 x 8,
// two accumulator tiles per channel block and wave) it needs 125 and four workgroups share a CU: measured in round 4 on the
// plain layers at batch 32, 64 -> 64 @256^2 1.184 -> 1.152 ms, 128 @128^2 1.136 -> 1.113, 256 @64^2 1.115 -> 1.099,
// 512 @32^2 1.115 -> 1.105 (-DGL_THICK_NB2=0 builds the 256-pixel tiles).
#ifndef GL_THICK_NB2
#define GL_THICK_NB2 1
#endif
using ThickCfg = FwdCfg<3, 4, GL_THICK_NB2 ? 4 : 5, 3, 0, XVEC>;

#ifdef GL_PHASES  // tools/phase_probe.py: per-workgroup phase timestamps (debug builds only)
__device__ unsigned long long* gl_phase_buf;
#define GL_T(i) \
  if (threadIdx.x == 0) gl_phase_buf[(long long)blockIdx.x * 8 + (i)] = wall_clock64();
// accumulated phase times over the tiles of a strip (slots 4..7)
#define GL_ACC_DECL unsigned long long gl_t_prev = wall_clock64(), gl_acc[4] = {0, 0, 0, 0};
#define GL_ACC(i)                                   \
  {                                                 \
    const unsigned long long gl_now = wall_clock64(); \
    gl_acc[i] += gl_now - gl_t_prev;                \
    gl_t_prev = gl_now;                             \
  }
#define GL_ACC_FLUSH \
  if (threadIdx.x == 0)                            \
    for (int gl_i = 0; gl_i < 4; ++gl_i) gl_phase_buf[(long long)blockIdx.x * 8 + 4 + gl_i] = gl_acc[gl_i];
#else
#define GL_T(i)
#define GL_ACC_DECL
#define GL_ACC(i)
#define GL_ACC_FLUSH
#endif

// Strip kernel for thin layers (one 16-channel output block, ONE K-chunk: Cin_p == CI_T, full tiles - the host
// checks): a workgroup walks `strip` tiles of one tile row.  Per tile and wave exactly PT buffer loads (the next
// tile's patch, hardware zero-fill at the borders) and NB 16-byte buffer stores are issued, unconditionally, so the
// compiler's vmcnt bookkeeping stays exact across the loop: the wait for the prefetched patch does not drain the
// stores of the tile that was just written.  The weight slab is staged in LDS once per strip.
template <class Cfg>
__global__ __launch_bounds__(256, 3) void conv_fwd_strip_kernel(ConvArgs p) {
  GL_T(0)
  using G = typename Cfg::G;
  constexpr int KS = Cfg::KS, KK = Cfg::KK, MB = Cfg::MB, NB = Cfg::NB, CI_T = Cfg::CI_T;
  constexpr int RP = G::RP, IMG = G::IMG, PLANE = Cfg::PLANE, COP = Cfg::COP;
  constexpr int TW = G::TW, TH = G::TH, CO_T = Cfg::CO_T, WPT = Cfg::WPT, NWI = Cfg::NWI;
  static_assert(MB == 1 && G::NI == 1 && G::XMODE != XSCALAR, "strip kernel: thin vector-staged layers only");
  using XS_t = XStage<G, CI_T, PLANE>;
  // LDS is padded to > 160 KB / 4 so that at most 3 workgroups share a CU (a 4th only adds DRAM-page and L2 churn)
  constexpr int SMEM = (Cfg::XS + Cfg::WS) > 10496 ? (Cfg::XS + Cfg::WS) : 10496;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];
  float* Xs = smem;
  float* Ws = smem + Cfg::XS;

  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  int bid = gl_xcd_remap(blockIdx.x, gridDim.x);
  const int co_t = bid % p.tiles_co;
  bid /= p.tiles_co;
  const int sxi = bid % p.strips_x;
  bid /= p.strips_x;
  const int tyi = bid % p.tiles_y;
  const int n0 = bid / p.tiles_y;
  const int tx0 = sxi * p.strip;
  const int ntiles = min(p.strip, p.tiles_x - tx0);
  const int co0 = co_t * CO_T, oy0 = tyi * TH;
  const int plane = p.in.Hi * p.in.Wi;
  const float* xb = p.in.x + (long long)n0 * p.in.Cin * plane;

  XS_t xst;
  xst.init_strip(p.in, tid, n0, oy0);
  int boff[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int j = wn * (16 * NB) + nb * 16 + (lane & 15);
    const int ty = (j >> G::TWL) & (TH - 1), tx = j & (TW - 1);
    boff[nb] = ty * RP + tx + G::XOFF + (lane >> 4) * PLANE;
  }
  const int aoff = (lane >> 4) * COP + (lane & 15);
  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // The MFMA below takes the activation patch as A and the weight slab as B, so D comes out [pixel][channel]: each
  // lane ends up with 4 CONSECUTIVE PIXELS (rows 4(l>>4)+r of the 16-pixel block) of ONE output channel (l&15) and
  // the epilogue is one 16-byte store per block instead of four 4-byte ones (store-issue bound otherwise).
  const int co_lane = co0 + (lane & 15);
  const float bv = (p.bias != nullptr && co_lane < p.Cout) ? p.bias[co_lane] * p.bias_scale : 0.f;
  // weight slab -> LDS (once)
#pragma unroll
  for (int i = 0; i < WPT; ++i) {
    const int e = tid + i * 256;
    const int c4 = e % (CO_T / 4);
    const int t = e / (CO_T / 4);
    const int ci = t % CI_T, tap = t / CI_T;
    if (e < NWI)
      *reinterpret_cast<float4*>(Ws + (tap * CI_T + ci) * COP + 4 * c4) =
          *reinterpret_cast<const float4*>(p.wp + (long long)(tap * p.Cin_p + ci) * p.Cout_p + co0 + 4 * c4);
  }
  // buffer descriptors (wave-uniform by construction: kernel arguments and blockIdx only)
  const long long out_plane = (long long)p.Ho * p.Wo;
  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (unsigned)(p.in.Cin * plane * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      p.y + (long long)n0 * p.Cout * out_plane, 0, (unsigned)(p.Cout * out_plane * 4), 0x00020000);
  int vo_lane[NB];   // byte offset of this lane's 4-pixel group (tile at x = 0) in channel co0 + (lane & 15)
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int j = wn * (16 * NB) + nb * 16 + (lane >> 4) * 4;
    const int ty = (j >> G::TWL) & (TH - 1), tx = j & (TW - 1);
    vo_lane[nb] = (int)(((long long)co_lane * out_plane + (long long)(oy0 + ty) * p.Wo + tx) * 4);
  }
  auto xbase_of = [&](int ox) { return G::XMODE == XVECUP ? (ox >> 1) : ox; };

  XRegs<G, XS_t::PT> xr;
  int ox0 = tx0 * TW;
  x_load_buf<G, CI_T, PLANE>(xr, xst, rs_in, 0, plane, xbase_of(ox0), p.in.Wi, tid);
  x_store<G, CI_T, PLANE, true>(xr, xst, Xs, tid);
  GL_ACC_DECL
  for (int t = 0; t < ntiles; ++t, ox0 += TW) {
    __syncthreads();  // the patch of tile t (and, first time round, the weight slab) is in LDS
    GL_T(1)
    GL_ACC(1)
    // next tile's patch (past the end of the strip / row the offsets fall outside the descriptor: zeros, unused)
    x_load_buf<G, CI_T, PLANE>(xr, xst, rs_in, 0, plane, t + 1 < ntiles ? xbase_of(ox0 + TW) : (1 << 28), p.in.Wi,
                               tid);
    // K-steps = (ky, kx, 4-channel group), fully unrolled with immediate LDS offsets.  With one 16-channel
    // output block per workgroup every MFMA needs 1.25 LDS reads, so the loop is bound by LDS latency unless
    // the operand reads run well ahead: a ring of PD+1 register sets keeps PD steps of reads in flight.
    {
      constexpr int C4N = CI_T / 4, NSTEP = KK * C4N, PD = 3;
      float ra[PD + 1], rb[PD + 1][NB];
      auto fetch = [&](int st, int slot) {
        const int ky = st / (KS * C4N), kx = (st / C4N) % KS, c4 = st % C4N;
        ra[slot] = Ws[ky * (KS * CI_T * COP) + aoff + (kx * CI_T + c4 * 4) * COP];
        const float* xrow = Xs + ky * RP + c4 * 4 * PLANE + kx;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) rb[slot][nb] = xrow[boff[nb]];
      };
#pragma unroll
      for (int st = 0; st < PD && st < NSTEP; ++st) fetch(st, st % (PD + 1));
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        if (st + PD < NSTEP) fetch(st + PD, (st + PD) % (PD + 1));
        const int slot = st % (PD + 1);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[slot][nb], ra[slot], acc[nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);   // keep "reads PD steps ahead, then this step's MFMAs" as written
      }
    }
    GL_T(2)
    GL_ACC(2)
    // epilogue: + bias, activation; the per-lane part of the address is fixed for the strip, the rest is an SGPR
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      u32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[nb][r] + bv;
        if (p.act == GANLAB_ACT_LRELU) v = gl_lrelu(v, p.slope);
        o[r] = __float_as_uint(v);
      }
      // the x offset goes into the VGPR offset, not the SGPR soffset: with a register soffset the compiler's hazard
      // recogniser assumes a >64-bit buffer store needs no wait state before its data VGPRs are rewritten, and on
      // gfx950 the next block's epilogue then overwrites element 0 for the last lanes of each row (seen on MI355X)
      __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, vo_lane[nb] + ox0 * 4, 0, 0);
      acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    GL_ACC(3)
    __syncthreads();  // every wave is done reading tile t's patch
    GL_ACC(0)
    // registers -> LDS for tile t+1.  Loads and stores above sit in this same straight-line block, so the wait
    // emitted here is vmcnt(#stores): the prefetched patch, not the stores that were just issued.
    x_store<G, CI_T, PLANE, true>(xr, xst, Xs, tid);
  }
  GL_ACC_FLUSH
  GL_T(3)
}

// Double-buffered VERTICAL strip kernel for the thinnest 3x3 layers (16 -> 16 channels: the whole weight set is 36
// values per lane, ONE K-chunk).  Differences from conv_fwd_strip_kernel:
//   * the weight slab lives in REGISTERS (lane l keeps W[tap][ci = 4 c4 + (l>>4)][co = l&15] for the 36 k-steps),
//     so the loop issues one LDS read per MFMA instead of 1.25 and no LDS is spent on weights;
//   * the patch is double-buffered in LDS (2 x 25.6 KB -> still 3 workgroups per CU): tile t+1 is written into the
//     other buffer right after tile t's MFMA loop, so there is ONE barrier per tile;
//   * a workgroup walks DOWN a column of tiles and neighbouring workgroups (consecutive block ids, same XCD) own
//     neighbouring columns: at any time the workgroups in flight read and write the SAME image rows, i.e. whole
//     4 KB rows of HBM pages, instead of 160-byte pieces of rows 4 KB apart.  Measured on the north-star conv
//     (tools/strip_ablate.py): MFMA + LDS alone 1.16 ms, with the scattered reads and writes 1.77 ms.
// (16x16 tiles carry a 18x24 patch per channel through the prefetch registers: two workgroups per CU is what they fit)
template <class Cfg>
__global__ __launch_bounds__(256, (Cfg::G::TW == 16 ? 2 : 3)) void conv_fwd_strip2_kernel(ConvArgs p) {
  GL_T(0)
  using G = typename Cfg::G;
  constexpr int KS = Cfg::KS, NB = Cfg::NB, CI_T = Cfg::CI_T;
  constexpr int RP = G::RP, PLANE = Cfg::PLANE, TW = G::TW, TH = G::TH, CO_T = Cfg::CO_T, XS = Cfg::XS;
  constexpr int C4N = CI_T / 4, NSTEP = KS * KS * C4N, PD = 3;
  constexpr int NITEMS = CI_T * G::R * G::ROW4, PT = ceil_div_c(NITEMS, 256);
  static_assert(KS == 3 && Cfg::MB == 1 && G::NI == 1 && G::XMODE == XVEC && CI_T == 16, "16-channel 3x3 vector strips");
  __shared__ __attribute__((aligned(16))) float smem[2 * XS];

  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  int bid = gl_xcd_remap(blockIdx.x, gridDim.x);
  const int co_t = bid % p.tiles_co;
  bid /= p.tiles_co;
  const int txi = bid % p.tiles_x;        // x fastest: neighbouring workgroups share image rows
  bid /= p.tiles_x;
  const int syi = bid % p.strips_x;       // (strips_x / strip count the strips along y here)
  const int n0 = bid / p.strips_x;
  const int ty0 = syi * p.strip;
  const int ntiles = min(p.strip, p.tiles_y - ty0);
  const int co0 = co_t * CO_T, ox0 = txi * TW, oy_first = ty0 * TH;
  const int plane = p.in.Hi * p.in.Wi;
  const float* xb = p.in.x + (long long)n0 * p.in.Cin * plane;

  // staging items (ci, patch row r, float4 column q): byte offset without the row term (or the out-of-range marker
  // when the column group lies in the zero padding / the item does not exist) and LDS offset | r << 20
  int gbase[PT], lo[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int e = tid + i * 256;
    const int q = e % G::ROW4;
    const int t = e / G::ROW4;
    const int r = t % G::R, ci = t / G::R;
    const int vx = ox0 - G::LP + 4 * q;
    gbase[i] = (e < NITEMS && (unsigned)vx < (unsigned)p.in.Wi) ? (ci * plane + vx) * 4 : (int)0x80000000;
    lo[i] = (ci * PLANE + r * RP + 4 * q) | (r << 20);
  }
  int boff[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int j = wn * (16 * NB) + nb * 16 + (lane & 15);
    const int ty = (j >> G::TWL) & (TH - 1), tx = j & (TW - 1);
    boff[nb] = ty * RP + tx + G::XOFF + (lane >> 4) * PLANE;
  }
  // weights -> registers (k-step st = tap * 4 + c4)
  float wreg[NSTEP];
#pragma unroll
  for (int st = 0; st < NSTEP; ++st)
    wreg[st] = p.wp[(long long)((st / C4N) * p.Cin_p + (st % C4N) * 4 + (lane >> 4)) * p.Cout_p + co0 + (lane & 15)];
  const int co_lane = co0 + (lane & 15);
  const float bv = (p.bias != nullptr && co_lane < p.Cout) ? p.bias[co_lane] * p.bias_scale : 0.f;

  const long long out_plane = (long long)p.Ho * p.Wo;
  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (unsigned)(p.in.Cin * plane * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      p.y + (long long)n0 * p.Cout * out_plane, 0, (unsigned)(p.Cout * out_plane * 4), 0x00020000);
  int vo_lane[NB];   // byte offset of this lane's 4-pixel group (tile row 0 of the image) in channel co0 + (lane & 15)
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int j = wn * (16 * NB) + nb * 16 + (lane >> 4) * 4;
    const int ty = (j >> G::TWL) & (TH - 1), tx = j & (TW - 1);
    vo_lane[nb] = (int)(((long long)co_lane * out_plane + (long long)ty * p.Wo + ox0 + tx) * 4);
  }
  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  float4 xr[PT];
  // patch of the tile whose first output row is `oy` (rows oy-1 .. oy+TH); oy >= Ho: every offset out of range
  auto load_patch = [&](int oy) {
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int vy = oy - G::PADC + (lo[i] >> 20);
      const bool ok = gbase[i] != (int)0x80000000 && (unsigned)vy < (unsigned)p.in.Hi;
      const int off = ok ? gbase[i] + (int)((unsigned)vy * (unsigned)(p.in.Wi * 4)) : (int)0x80000000;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
      xr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    }
  };
  auto store_patch = [&](float* Xd) {
#pragma unroll
    for (int i = 0; i < PT; ++i)
      if (tid + i * 256 < NITEMS) *reinterpret_cast<float4*>(Xd + (lo[i] & 0xfffff)) = xr[i];
  };

  int oy0 = oy_first;
  load_patch(oy0);
  store_patch(smem);
  __syncthreads();
  GL_T(1)
  GL_T(2)
  GL_ACC_DECL
  for (int t = 0; t < ntiles; ++t, oy0 += TH) {
    const float* Xb = smem + (t & 1) * XS;
    // next tile's patch -> registers (past the end of the strip: out-of-range offsets, zeros, written but never read)
#ifdef GL_ABL_NOLOAD   // tools/strip_ablate.py: every "next tile" load falls outside the descriptor (no HBM reads)
    load_patch(1 << 28);
#else
    load_patch(t + 1 < ntiles ? oy0 + TH : (1 << 28));
#endif
#ifndef GL_ABL_NOMFMA
    {
      float rb[PD + 1][NB];
      auto fetch = [&](int st, int slot) {
        const int ky = st / (KS * C4N), kx = (st / C4N) % KS, c4 = st % C4N;
        const float* xrow = Xb + ky * RP + c4 * 4 * PLANE + kx;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) rb[slot][nb] = xrow[boff[nb]];
      };
#pragma unroll
      for (int st = 0; st < PD; ++st) fetch(st, st % (PD + 1));
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        if (st + PD < NSTEP) fetch(st + PD, (st + PD) % (PD + 1));
        const int slot = st % (PD + 1);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[slot][nb], wreg[st], acc[nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#endif
    GL_ACC(2)
    // tile t+1 into the other buffer: its last readers (tile t-1) are behind the previous barrier
    store_patch(smem + ((t + 1) & 1) * XS);
    GL_ACC(1)
    const int orow = oy0 * p.Wo * 4;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      u32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[nb][r] + bv;
        if (p.act == GANLAB_ACT_LRELU) v = gl_lrelu(v, p.slope);
        o[r] = __float_as_uint(v);
      }
      // the row offset goes into the VGPR offset, not the SGPR soffset: with a register soffset the compiler's
      // hazard recogniser assumes a >64-bit buffer store needs no wait state before its data VGPRs are rewritten,
      // and on gfx950 the next block's epilogue then overwrites element 0 for the last lanes of each row
#ifdef GL_ABL_NOSTORE   // only the last tile of the strip is written; the accumulators run on so every MFMA stays live
      if (t + 1 == ntiles)
#endif
      __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, vo_lane[nb] + orow, 0, 0);
#ifndef GL_ABL_NOSTORE
      acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
    }
    GL_ACC(3)
    __syncthreads();   // everyone is done reading buffer t&1 and buffer (t+1)&1 is complete
    GL_ACC(0)
  }
  GL_ACC_FLUSH
  GL_T(3)
}

// Rolling-window kernel for the thinnest 3x3 layers (<= 16 -> 16 channels, W % 64 == 0, H % 4 == 0).  A workgroup
// owns a 64-pixel-wide column strip and walks DOWN it 4 output rows at a time; the input rows live in an LDS ring of 10
// row slots ([ci 16][80 floats] each): a step needs rows 4t .. 4t+5 of the strip, rows 4t+6 .. 4t+9 are prefetched into
// registers during its MFMA loop and written into the four free slots afterwards, so
//   * every input element is fetched ONCE per workgroup (plus the 8-of-72 column halo): 288-byte contiguous runs, 4
//     cache lines per (channel, row) and 4 output rows, instead of 3 lines per (channel, row) and 0.8 output rows of
//     the tile-based kernels - the thin layers are bound by the number of line requests, not by bandwidth;
//   * one barrier per step (144 MFMAs per wave), weights in registers, 51 KB of LDS: 3 workgroups per CU.
// Wave w computes output row 4t + w (64 pixels = 4 MFMA column blocks).
#ifndef RW_NSLOTS
#define RW_NSLOTS 6
#endif
// RW_SLOTS = 10: the four prefetched rows go into slots nobody reads, one barrier per step (51 KB: 3 workgroups per CU).
// RW_SLOTS = 6: they overwrite rows 4t .. 4t+3 after a barrier of their own (31 KB: 4 workgroups per CU, the kernel
// needs 127 VGPRs) - two barriers per step, the second one right behind five LDS stores.
constexpr int RW_TW = 64, RW_ROWS = 4, RW_SLOTS = RW_NSLOTS, RW_RP = 80, RW_SLOT = 16 * RW_RP, RW_Q = 18;
static_assert(RW_SLOTS == 6 || RW_SLOTS == 10, "ring of 6 (two barriers) or 10 (one barrier) row slots");
#ifndef RW_LOAD_AT
#define RW_LOAD_AT 6
#endif
constexpr int RW_ITEMS = RW_ROWS * 16 * RW_Q;            // float4 items of one 4-row prefetch: 1152
constexpr int RW_PT = (RW_ITEMS + 255) / 256;            // 5

__global__ __launch_bounds__(256, (RW_NSLOTS == 6 ? 4 : 3)) void conv_fwd_roll_kernel(ConvArgs p) {
  __shared__ __attribute__((aligned(16))) float ring[RW_SLOTS * RW_SLOT];
  constexpr int C4N = 4, NSTEP = 36, PD = 3, NB = 4;
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  int bid = gl_xcd_remap(blockIdx.x, gridDim.x);
  const int txi = bid % p.tiles_x;        // x fastest: neighbouring workgroups share image rows
  bid /= p.tiles_x;
  const int syi = bid % p.strips_x;
  const int n0 = bid / p.strips_x;
  const int step0 = syi * p.strip;
  const int nsteps = min(p.strip, p.tiles_y - step0);     // tiles_y = H / 4 steps per column
  const int ox0 = txi * RW_TW, oy_first = step0 * RW_ROWS;
  const int plane = p.in.Hi * p.in.Wi;
  const float* xb = p.in.x + (long long)n0 * p.in.Cin * plane;

  // staging items of a 4-row group: (k = row in group, ci, q = float4 column): byte offset without the row term (or the
  // out-of-range marker) and LDS offset within a slot | k << 20
  int gbase[RW_PT], lo[RW_PT];
#pragma unroll
  for (int i = 0; i < RW_PT; ++i) {
    const int e = tid + i * 256;
    const int q = e % RW_Q;
    const int t = e / RW_Q;
    const int ci = t & 15, k = t >> 4;
    const int vx = ox0 - 4 + 4 * q;
    gbase[i] = (e < RW_ITEMS && ci < p.in.Cin && (unsigned)vx < (unsigned)p.in.Wi) ? (ci * plane + vx) * 4
                                                                                  : (int)0x80000000;
    lo[i] = (ci * RW_RP + 4 * q) | (k << 20);
  }
  // weights -> registers (k-step st = tap * 4 + c4)
  float wreg[NSTEP];
#pragma unroll
  for (int st = 0; st < NSTEP; ++st)
    wreg[st] = p.wp[(long long)((st / C4N) * p.Cin_p + (st % C4N) * 4 + (lane >> 4)) * p.Cout_p + (lane & 15)];
  const int co_lane = lane & 15;
  const float bv = (p.bias != nullptr && co_lane < p.Cout) ? p.bias[co_lane] * p.bias_scale : 0.f;
  int boff[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) boff[nb] = (lane >> 4) * RW_RP + nb * 16 + (lane & 15) + 3;   // + LP(4) - pad(1)

  const long long out_plane = (long long)p.Ho * p.Wo;
  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (unsigned)(p.in.Cin * plane * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      p.y + (long long)n0 * p.Cout * out_plane, 0, (unsigned)(p.Cout * out_plane * 4), 0x00020000);
  int vo_lane[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
    vo_lane[nb] = co_lane < p.Cout
                      ? (int)(((long long)co_lane * out_plane + ox0 + nb * 16 + (lane >> 4) * 4) * 4)
                      : (int)0x80000000;          // channel padding: the store falls outside the descriptor
  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  float4 xr[RW_PT];
  // rows rel0 .. rel0+nrows-1 of the strip (row 0 = input row oy_first - 1) -> registers; rows outside the image and
  // k >= nrows read as zeros (out-of-range offsets)
  auto load_rows = [&](int rel0, int nrows) {
#pragma unroll
    for (int i = 0; i < RW_PT; ++i) {
      const int k = lo[i] >> 20;
      const int vy = oy_first - 1 + rel0 + k;
      const bool ok = gbase[i] != (int)0x80000000 && k < nrows && (unsigned)vy < (unsigned)p.in.Hi;
      const int off = ok ? gbase[i] + (int)((unsigned)vy * (unsigned)(p.in.Wi * 4)) : (int)0x80000000;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
      xr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    }
  };
  auto store_rows = [&](int rel0, int nrows) {
#pragma unroll
    for (int i = 0; i < RW_PT; ++i) {
      const int k = lo[i] >> 20;
      if (tid + i * 256 < RW_ITEMS && k < nrows)
        *reinterpret_cast<float4*>(ring + ((rel0 + k) % RW_SLOTS) * RW_SLOT + (lo[i] & 0xfffff)) = xr[i];
    }
  };
  // the steady-state form: all four rows, slot of the first row given (a scalar), no per-item modulo
  int lo_k[RW_PT], lo_off[RW_PT];
#pragma unroll
  for (int i = 0; i < RW_PT; ++i) {
    lo_k[i] = lo[i] >> 20;
    lo_off[i] = (lo[i] & 0xfffff) + lo_k[i] * RW_SLOT;
  }
  auto store_rows4 = [&](int slot0) {      // slot0 = (rel0 % RW_SLOTS), wave-uniform
#pragma unroll
    for (int i = 0; i < RW_PT; ++i) {
      const int wrap = (slot0 + lo_k[i] >= RW_SLOTS) ? RW_SLOTS * RW_SLOT : 0;
      if (i + 1 < RW_PT || tid + i * 256 < RW_ITEMS)
        *reinterpret_cast<float4*>(ring + slot0 * RW_SLOT + lo_off[i] - wrap) = xr[i];
    }
  };

  load_rows(0, 4);
  store_rows(0, 4);
  load_rows(4, 2);
  store_rows(4, 2);
  __syncthreads();
  for (int t = 0; t < nsteps; ++t) {
    {
      int sbase[3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) sbase[ky] = ((4 * t + wn + ky) % RW_SLOTS) * RW_SLOT;
      float rb[PD + 1][NB];
      auto fetch = [&](int st, int slot) {
        const int ky = st / (3 * C4N), kx = (st / C4N) % 3, c4 = st % C4N;
        const float* xrow = ring + sbase[ky] + c4 * 4 * RW_RP + kx;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) rb[slot][nb] = xrow[boff[nb]];
      };
#pragma unroll
      for (int st = 0; st < PD; ++st) fetch(st, st % (PD + 1));
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        if (st + PD < NSTEP) fetch(st + PD, (st + PD) % (PD + 1));
        const int slot = st % (PD + 1);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[slot][nb], wreg[st], acc[nb], 0, 0, 0);
        // rows 4t+6 .. 4t+9 for the next step (none after the last step: every offset out of range).  Issued a few
        // k-steps INTO the loop: the loads reuse the registers the previous step's output stores read their data from, so
        // the compiler waits for those stores first - with MFMAs already queued that wait costs nothing
        if (st == RW_LOAD_AT) load_rows(4 * t + 6, t + 1 < nsteps ? 4 : 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (RW_SLOTS == 10) store_rows4((4 * t + 6) % RW_SLOTS);       // the four slots not read by this step
    const int orow = (oy_first + 4 * t + wn) * p.Wo * 4;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      u32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[nb][r] + bv;
        if (p.act == GANLAB_ACT_LRELU) v = gl_lrelu(v, p.slope);
        o[r] = __float_as_uint(v);
      }
      // offset in the VGPR, soffset 0: see the store-data hazard note in conv_fwd_strip2_kernel
      __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, vo_lane[nb] == (int)0x80000000 ? vo_lane[nb] : vo_lane[nb] + orow,
                                             0, 0);
      acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();   // rows 4t .. 4t+5 are no longer read; (10 slots:) rows 4t+6 .. 4t+9 are complete
    if constexpr (RW_SLOTS == 6) {
      store_rows4((4 * t + 6) % RW_SLOTS);     // = the slots of rows 4t .. 4t+3
      __syncthreads();
    }
  }
}

// One tile per workgroup (thick layers: dozens of K-chunks per tile amortise the set-up, and the register budget
// has no room for the strip bookkeeping).
constexpr int AFF_MAXC = 512;     // channels of the per-image (s, t) table the AFF kernels keep in LDS
#ifndef GL_FWD_PIPE_WG
#define GL_FWD_PIPE_WG 4
#endif
#ifndef GL_FWD_MB2_WG
#define GL_FWD_MB2_WG 4
#endif
template <class Cfg, bool MASK = false, bool SPLITK = false, bool AFF = false, bool TAIL = false>
__global__ __launch_bounds__(256, (Cfg::G::XMODE == XSCALAR || (Cfg::DUMP > 0 && Cfg::MB >= 4 && Cfg::NB >= 4) ? 2 :
                                   (Cfg::KS == 3 && Cfg::MB == 4 && Cfg::NB == 2 && Cfg::G::XMODE == XVEC && !SPLITK) ? GL_FWD_PIPE_WG :
                                   (Cfg::KS == 3 && Cfg::MB == 2 && Cfg::NB == 4 && Cfg::G::XMODE == XVEC && !SPLITK) ? GL_FWD_MB2_WG : 3)) void conv_fwd_kernel(ConvArgs p) {
  using G = typename Cfg::G;
  constexpr int KS = Cfg::KS, KK = Cfg::KK, MB = Cfg::MB, NB = Cfg::NB, CI_T = Cfg::CI_T;
  constexpr int RP = G::RP, IMG = G::IMG, PLANE = Cfg::PLANE, COP = Cfg::COP;
  constexpr int TW = G::TW, TH = G::TH, NI = G::NI, CO_T = Cfg::CO_T, WPT = Cfg::WPT, NWI = Cfg::NWI;
  using XS_t = XStage<G, CI_T, PLANE>;
  __shared__ __attribute__((aligned(16))) float smem[Cfg::XS + Cfg::WS + (AFF ? 2 * AFF_MAXC : 0)];
  float* Xs = smem;
  float* Ws = smem + Cfg::XS;
  [[maybe_unused]] float* afftab = smem + Cfg::XS + Cfg::WS;     // AFF: s[0 .. Cin) | t at + AFF_MAXC of THIS image

  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  int bid = gl_xcd_remap(blockIdx.x, gridDim.x);
  int c_begin = 0, c_end = p.Cin_p;
  if constexpr (SPLITK) {
    const int ks = bid % p.ksplit;
    bid /= p.ksplit;
    c_begin = ks * p.ksplit_ci;
    c_end = min(p.Cin_p, c_begin + p.ksplit_ci);
    p.y += ks * p.ksplit_stride;
  }
  const int co_t = bid % p.tiles_co;
  bid /= p.tiles_co;
  const int txi = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int tyi = bid % p.tiles_y;
  const int tni = bid / p.tiles_y;
  const int co0 = co_t * CO_T, ox0 = txi * TW, oy0 = tyi * TH, n0 = tni * NI;
  const int plane = p.in.Hi * p.in.Wi;
  const float* xb = p.in.x + (long long)n0 * p.in.Cin * plane;

  // ---- staging descriptors (independent of the K-chunk) ----
  XS_t xst;
  xst.init(p.in, tid, n0, oy0, ox0);
  // (one tap per 256 threads when a tap's slab is exactly 256 float4 items: the descriptors are affine in i and need no
  // registers of their own)
  constexpr bool WAFF = CI_T * (CO_T / 4) == 256;
  int wg[WAFF ? 1 : WPT], wl[WAFF ? 1 : WPT];
  [[maybe_unused]] const int wg_step = p.Cin_p * p.Cout_p;
#pragma unroll
  for (int i = 0; i < WPT; ++i) {
    const int e = tid + i * 256;
    const int c4 = e % (CO_T / 4);
    const int t = e / (CO_T / 4);
    const int ci = t % CI_T, tap = t / CI_T;
    if (WAFF && i > 0) break;
    wl[i] = (tap * CI_T + ci) * COP + 4 * c4;
    wg[i] = e < NWI ? (tap * p.Cin_p + ci) * p.Cout_p + co0 + 4 * c4 : -1;
  }
  // ---- per-lane MFMA operand offsets ----
  int boff[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int j = wn * (16 * NB) + nb * 16 + (lane & 15);
    const int ni = j >> (G::TWL + G::THL), ty = (j >> G::TWL) & (TH - 1), tx = j & (TW - 1);
    boff[nb] = ni * IMG + ty * RP + tx + G::XOFF + (lane >> 4) * PLANE;
  }
  const int aoff = (lane >> 4) * COP + (lane & 15);
  f32x4 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int DUMP = Cfg::DUMP;
  [[maybe_unused]] f32x4 acc2[DUMP ? MB : 1][DUMP ? NB : 1];
  [[maybe_unused]] int since_dump = 0;
  if constexpr (DUMP > 0) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc2[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- PIPE: the 64-channel x 128-pixel tile of the thick 3x3 layers by HALF chunks ------------------------------------------
  // The chunk loop below is [barrier][registers -> LDS][barrier][loads][144 MFMAs per wave]: a build without the LDS stores runs
  // 256 -> 256 @64^2 in 1.035 instead of 1.101 ms (0.95 of the peak) - nothing of this workgroup covers them.  Here the 8-channel
  // chunk's LDS space holds TWO images of a 4-channel half chunk: the stores of half chunk h+1 go to the image that died at the
  // previous barrier, in pieces behind the K-steps (taps) of half chunk h's MFMAs, its loads were issued two half chunks earlier
  // (two register sets), and ONE barrier per half chunk (72 MFMAs per wave) publishes them - the same barriers per MFMA as before.
#ifndef GL_FWD_PIPE
#define GL_FWD_PIPE 1
#endif
#ifndef GL_FWD_PIPE_MB2         // the 32-channel x 256-pixel tile the same way, with ONE register set (two sets: 138 instead of 127 VGPRs =
#define GL_FWD_PIPE_MB2 1       // three workgroups per CU instead of four, 32 -> 32 @512^2 x32 1.245 -> 1.260 ms; forced to 128 it spills: 1.376)
#endif
  constexpr bool PIPE = GL_FWD_PIPE && KS == 3 && ((MB == 4 && NB == 2) || (GL_FWD_PIPE_MB2 && MB == 2 && NB == 4)) && G::XMODE == XVEC &&
                        G::NI == 1 && !SPLITK && CI_T == 8 && DUMP > 0;
  if constexpr (PIPE) {
    constexpr int HC = 4, XH = HC * PLANE, WH = KK * HC * COP;
    constexpr int XI = HC * G::R * G::ROW4, WI = KK * HC * (CO_T / 4);
    constexpr int XPH = (XI + 255) / 256, WPH = (WI + 255) / 256;          // 1 + 3 (64 channels x 128 pixels), 2 + 2 (32 x 256)
    static_assert(!PIPE || (2 * XH == Cfg::XS && 2 * WH == Cfg::WS && 2 * (XPH + WPH) <= KK), "half-chunk images / pieces");
    constexpr int NOITEM = (int)0x80000000;
    // staging descriptors of a half chunk (tile-fixed): byte offset inside the image (or NOITEM) and LDS offset | ci << 20
    int xg[XPH], xl[XPH];
#pragma unroll
    for (int i = 0; i < XPH; ++i) {
      const int e = tid + i * 256;
      const int q = e % G::ROW4, t = e / G::ROW4;
      const int r = t % G::R, ci = t / G::R;
      const int vy = oy0 + r - G::PADC, vx = ox0 - G::LP + 4 * q;
      xl[i] = e < XI ? ((ci * PLANE + r * RP + 4 * q) | (ci << 20)) : -1;
      xg[i] = (e < XI && n0 < p.in.N && (unsigned)vy < (unsigned)p.in.Hi && (unsigned)vx < (unsigned)p.in.Wi)
                  ? (ci * plane + vy * p.in.Wi + vx) * 4 : NOITEM;
    }
    int wgo[WPH];                                       // (the LDS offset (tap * HC + ci) * COP + 4 c4 is recomputed where it is used:
#pragma unroll                                          //  three instructions against a register across the loop)
    for (int i = 0; i < WPH; ++i) {
      const int e = tid + i * 256;
      const int c4 = e % (CO_T / 4), t = e / (CO_T / 4);
      const int ci = t % HC, tap = t / HC;
      wgo[i] = e < WI ? ((tap * p.Cin_p + ci) * p.Cout_p + co0 + 4 * c4) * 4 : NOITEM;
    }
    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (unsigned)(p.in.Cin * plane * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.wp), 0, (unsigned)(KK * p.Cin_p * p.Cout_p * 4), 0x00020000);
    const int nh = (c_end - c_begin) / HC;            // half chunks (even: Cin_p is a multiple of 8)
    // register sets: two (loads two half chunks ahead) for the 64-channel tile; ONE for the 32-channel tile (loads one half
    // chunk ahead - its operands come from L2 - and 16 registers less: the fourth workgroup per CU)
#ifndef GL_FWD_PIPE_NSET4
#define GL_FWD_PIPE_NSET4 2
#endif
    constexpr int NSET = MB == 4 ? GL_FWD_PIPE_NSET4 : 1;
    float4 xr2[NSET][XPH], wr2[NSET][WPH];
    auto load_x2 = [&](int h, int set, int i) {
      const int c0 = c_begin + h * HC, ci = (xl[i] >> 20) & 0x3ff;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (h < nh && c0 + ci < p.in.Cin) ? xg[i] : NOITEM, c0 * plane * 4, 0);
      xr2[set][i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    };
    auto load_w2 = [&](int h, int set, int i) {
      const int c0 = c_begin + h * HC;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, h < nh ? wgo[i] : NOITEM, c0 * p.Cout_p * 4, 0);
      wr2[set][i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    };
    auto store_x2 = [&](int h, int set, int i) {        // half chunk h -> image h & 1
      float4 v = xr2[set][i];
      if constexpr (AFF) {
        const int c = c_begin + h * HC + ((xl[i] >> 20) & 0x3ff);
        const bool ok = xl[i] != -1 && xg[i] != NOITEM && h < nh && c < p.in.Cin;
        const float sv = ok ? afftab[ok ? c : 0] : 0.f, tv = ok ? afftab[ok ? AFF_MAXC + c : 0] : 0.f;
        v.x = fmaf(v.x, sv, tv); v.y = fmaf(v.y, sv, tv); v.z = fmaf(v.z, sv, tv); v.w = fmaf(v.w, sv, tv);
      }
      if (xl[i] != -1) *reinterpret_cast<float4*>(Xs + (h & 1) * XH + (xl[i] & 0xfffff)) = v;
    };
    auto store_w2 = [&](int h, int set, int i) {
      const int e = tid + i * 256;
      if (e < WI) *reinterpret_cast<float4*>(Ws + (h & 1) * WH + (e / (CO_T / 4)) * COP + 4 * (e % (CO_T / 4))) = wr2[set][i];
    };
    auto load_half = [&](int h, int set) {
#pragma unroll
      for (int i = 0; i < XPH; ++i) load_x2(h, set, i);
#pragma unroll
      for (int i = 0; i < WPH; ++i) load_w2(h, set, i);
    };
    load_half(0, 0);
    if constexpr (NSET == 2) load_half(1, 1);
    if constexpr (AFF) {
      for (int c = tid; c < p.in.Cin; c += 256) {
        afftab[c] = p.in.aff_s[(long long)n0 * p.in.Cin + c];
        afftab[AFF_MAXC + c] = p.in.aff_t[(long long)n0 * p.in.Cin + c];
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < XPH; ++i) store_x2(0, 0, i);
#pragma unroll
    for (int i = 0; i < WPH; ++i) store_w2(0, 0, i);
    load_half(NSET == 2 ? 2 : 1, 0);
    __syncthreads();
    constexpr int DUMPH = 2 * DUMP;                     // half chunks per accumulator dump
    auto half_chunk = [&](int h, int par) {             // `par` = h & 1, a literal at both call sites
      const float* xs = Xs + par * XH;
      const float* ws = Ws + par * WH + aoff;
      float a[2][MB], b[2][NB];
      auto fetch = [&](int k, int st) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) a[st][mb] = ws[k * HC * COP + mb * 16];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[st][nb] = xs[(k / KS) * RP + (k % KS) + boff[nb]];
      };
      // staging pieces behind the taps: half chunk h+1 (register set par ^ 1) goes to the other image, then that set is
      // refilled with half chunk h+3
      auto piece = [&](int k) {
        constexpr int AHEAD = NSET == 2 ? 3 : 2;
        const int set = NSET == 2 ? (par ^ 1) : 0;
        if (k < XPH) store_x2(h + 1, set, k);
        else if (k < XPH + WPH) store_w2(h + 1, set, k - XPH);
        else if (k < 2 * XPH + WPH) load_x2(h + AHEAD, set, k - XPH - WPH);
        else if (k < 2 * (XPH + WPH)) load_w2(h + AHEAD, set, k - 2 * XPH - WPH);
      };
      fetch(0, 0);
#pragma unroll
      for (int k = 0; k < KK; ++k) {
        if (k + 1 < KK) fetch(k + 1, (k + 1) & 1);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[k & 1][nb], a[k & 1][mb], acc[mb][nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        piece(k);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (++since_dump == DUMPH && h + 1 < nh) {
        since_dump = 0;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            acc2[mb][nb] += acc[mb][nb];
            acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
      }
      __syncthreads();
    };
    for (int h = 0; h < nh; h += 2) {
      half_chunk(h, 0);
      half_chunk(h + 1, 1);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[mb][nb] += acc2[mb][nb];
  } else {
  XRegs<G, XS_t::PT> xr;
  float4 wr[WPT];
  // plain vector staging (the thick layers): patch and weight prefetch through buffer descriptors - out-of-range items
  // come back as zeros from the bounds check, so the ~20 loads of a chunk are straight-line code between the barrier
  // and the MFMA loop instead of 20 exec-mask branches
  constexpr bool DESC = G::XMODE == XVEC && G::NI == 1 && !SPLITK;
  __amdgpu_buffer_rsrc_t rs_x, rs_w;
  if constexpr (DESC) {
    rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (unsigned)(p.in.Cin * plane * 4), 0x00020000);
    rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), 0, (unsigned)(KK * p.Cin_p * p.Cout_p * 4),
                                             0x00020000);
  }
  auto load_w = [&](int ci0) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      if constexpr (DESC) {
        const int wgi = WAFF ? wg[0] + i * wg_step : wg[i];
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wgi >= 0 ? wgi * 4 : (int)0x80000000,
                                                              ci0 * p.Cout_p * 4, 0);
        wr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
      } else {
        const int wgi = WAFF ? wg[0] + i * wg_step : wg[i];
        wr[i] = wgi >= 0 ? *reinterpret_cast<const float4*>(p.wp + (long long)ci0 * p.Cout_p + wgi)
                           : float4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto load_x = [&](int ci0) {
    if constexpr (DESC) x_load_desc<G, CI_T, PLANE>(xr, xst, rs_x, ci0, p.in.Cin, plane);
    else x_load<G, CI_T, PLANE>(xr, xst, xb, ci0, p.in.Cin, plane);
  };
  load_x(c_begin);
  load_w(c_begin);
  if constexpr (AFF) {      // (visible to every wave after the first barrier of the chunk loop)
    static_assert(!AFF || (G::XMODE == XVEC && G::NI == 1 && !SPLITK), "AFF: vector staging, one image per tile");
    for (int c = tid; c < p.in.Cin; c += 256) {
      afftab[c] = p.in.aff_s[(long long)n0 * p.in.Cin + c];
      afftab[AFF_MAXC + c] = p.in.aff_t[(long long)n0 * p.in.Cin + c];
    }
  }
  auto stage = [&](float* xs, float* ws, int ci0) {       // prefetch registers -> LDS
    if constexpr (AFF) x_store<G, CI_T, PLANE, true, true>(xr, xst, xs, tid, afftab, ci0, AFF_MAXC, p.in.Cin - ci0);
    else x_store<G, CI_T, PLANE, true>(xr, xst, xs, tid);
#pragma unroll
    for (int i = 0; i < WPT; ++i)
      if (tid + i * 256 < NWI) *reinterpret_cast<float4*>(ws + (WAFF ? wl[0] + i * CI_T * COP : wl[i])) = wr[i];
  };
  auto dump = [&](bool more) {        // the chain of the last DUMP chunks joins the second-level sum and restarts
    if constexpr (DUMP > 0) {
      if (++since_dump == DUMP && more) {
        since_dump = 0;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            acc2[mb][nb] += acc[mb][nb];
            acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
      }
    }
  };
  // taps: ky is a real loop (bounds the compiler's hoisting of LDS reads, i.e. register pressure),
  // kx and the 4-channel K-steps are unrolled with immediate LDS offsets
  auto mfma_row = [&](const float* xs, const float* ws, int ky) {
    const float* wrow = ws + ky * (KS * CI_T * COP) + aoff;
    const float* xrow = xs + ky * RP;
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
      for (int c4 = 0; c4 < CI_T / 4; ++c4) {
        float a[MB], b[NB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) a[mb] = wrow[(kx * CI_T + c4 * 4) * COP + mb * 16];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[nb] = xrow[c4 * 4 * PLANE + boff[nb] + kx];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[nb], a[mb], acc[mb][nb], 0, 0, 0);
      }
    }
  };

  // The same K-steps with the operand fragments of step k+1 requested BEFORE the MFMAs of step k (two fragment sets, the
  // chunk fully unrolled, order pinned): left to itself hipcc emits "ds_read, s_waitcnt lgkmcnt(0), 8 MFMAs" groups, one
  // exposed LDS latency per 256 MFMA cycles that only the other waves of the SIMD cover - which the kernels with a second
  // accumulator set (two waves per SIMD) have too few of.
  auto mfma_chunk = [&](const float* xs, const float* ws) {
    constexpr int NK = KK * (CI_T / 4);
    float a[2][MB], b[2][NB];
    auto fetch = [&](int k, int st) {
      const int tap = k / (CI_T / 4), c4 = k % (CI_T / 4), ky = tap / KS, kx = tap % KS;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) a[st][mb] = ws[aoff + (tap * CI_T + c4 * 4) * COP + mb * 16];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) b[st][nb] = xs[ky * RP + c4 * 4 * PLANE + boff[nb] + kx];
    };
    fetch(0, 0);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      if (k + 1 < NK) fetch(k + 1, (k + 1) & 1);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[k & 1][nb], a[k & 1][mb], acc[mb][nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // (every 3x3 instantiation: the element-wise staged small-map tiles gain the most - 512 -> 512 at 8^2 87 -> 96 TFLOP/s, at 4^2
  // 52 -> 59, at 16^2 and batch 8 91 -> 101; GL_FRAG_PREFETCH_K1 extends it to the 1x1 / linear instantiations)
#ifndef GL_FRAG_PREFETCH_K1
#define GL_FRAG_PREFETCH_K1 0
#endif
  constexpr bool FRAGPF = GL_FRAG_PREFETCH && (KS == 3 || GL_FRAG_PREFETCH_K1);

  for (int ci0 = c_begin; ci0 < c_end; ci0 += CI_T) {
    __syncthreads();  // every wave is done reading the previous chunk
    stage(Xs, Ws, ci0);   // every staging mode goes through the register prefetch
    __syncthreads();
    if (ci0 + CI_T < c_end) {  // prefetch the next chunk: in flight during the MFMA phase below
      load_x(ci0 + CI_T);
      load_w(ci0 + CI_T);
    }
    if constexpr (FRAGPF) {
      mfma_chunk(Xs, Ws);
    } else {
#pragma unroll 1
      for (int ky = 0; ky < KS; ++ky) mfma_row(Xs, Ws, ky);
    }
    dump(ci0 + CI_T < c_end);
  }
  if constexpr (DUMP > 0) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[mb][nb] += acc2[mb][nb];
  }
  }      // (!PIPE)
  // ---- epilogue: + bias, activation, NCHW store ----
  // The patch is the MFMA's A operand and the weights its B operand, so D is [pixel][channel]: a lane holds pixels
  // 4(l>>4)+r (r = 0..3, consecutive along x in every tile geometry) of output channel l&15 -> one 16-byte store
  // per accumulator tile when the row is 4-aligned (vector staging modes), four 4-byte ones for ragged shapes.
  const long long out_plane = (long long)p.Ho * p.Wo;
  float bv[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int co = co0 + mb * 16 + (lane & 15);
    bv[mb] = (p.bias != nullptr && co < p.Cout) ? p.bias[co] * p.bias_scale : 0.f;
  }
  if constexpr (MASK && G::XMODE != XSCALAR) {   // (its own instantiation: the plain kernel's registers stay as they were)
    {                              // all mask loads in flight together, then the stores
      float4 mk[NB][MB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int j0 = wn * (16 * NB) + nb * 16 + (lane >> 4) * 4;
        const int oy = oy0 + ((j0 >> G::TWL) & (TH - 1)), ox = ox0 + (j0 & (TW - 1));
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const int co = co0 + mb * 16 + (lane & 15);
          // out-of-range outputs are never stored: they read mask[0..3] (always there) instead of being predicated,
          // which keeps eight saved exec masks out of the scalar registers
          const bool ok = co < p.Cout && n0 < p.in.N && oy < p.Ho && ox < p.Wo;
          const long long off = ok ? ((long long)n0 * p.Cout + co) * out_plane + (long long)oy * p.Wo + ox : 0LL;
          mk[nb][mb] = *reinterpret_cast<const float4*>(p.mask + off);
        }
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const float4 m = mk[nb][mb];
          if (!(m.x > 0.f)) acc[mb][nb][0] *= p.mslope;
          if (!(m.y > 0.f)) acc[mb][nb][1] *= p.mslope;
          if (!(m.z > 0.f)) acc[mb][nb][2] *= p.mslope;
          if (!(m.w > 0.f)) acc[mb][nb][3] *= p.mslope;
        }
    }
  }
  if constexpr (TAIL) {
    // the rest of a generator layer in the epilogue (stylegan/architectures.py:497-526): + noise_w * noise + bias,
    // LeakyReLU, store, and the tile's share of the InstanceNorm statistics of the result (fp32 over this lane's <= 4*NB
    // pixels, fp64 from there on: lane groups, waves, then the fixed-order finish over the tiles)
    static_assert(!TAIL || (G::XMODE == XVEC && G::NI == 1), "TAIL: vector staging, one image per tile");
    float nwv[MB], s1[MB], s2[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int co = co0 + mb * 16 + (lane & 15);
      nwv[mb] = (p.noise != nullptr && co < p.Cout) ? p.noise_w[co] : 0.f;
      s1[mb] = 0.f;
      s2[mb] = 0.f;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int j0 = wn * (16 * NB) + nb * 16 + (lane >> 4) * 4;
      const int oy = oy0 + ((j0 >> G::TWL) & (TH - 1)), ox = ox0 + (j0 & (TW - 1));
      const bool in = oy < p.Ho && ox < p.Wo;
      float4 nz = float4{0.f, 0.f, 0.f, 0.f};
      if (p.noise != nullptr && in) nz = *reinterpret_cast<const float4*>(p.noise + ((long long)n0 * p.Ho + oy) * p.Wo + ox);
      const float nzv[4] = {nz.x, nz.y, nz.z, nz.w};
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int co = co0 + mb * 16 + (lane & 15);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = fmaf(nzv[r], nwv[mb], acc[mb][nb][r] + bv[mb]);
          if (p.act == GANLAB_ACT_LRELU) v[r] = gl_lrelu(v[r], p.slope);
        }
        if (in && co < p.Cout) {
          *reinterpret_cast<float4*>(p.y + ((long long)n0 * p.Cout + co) * out_plane + (long long)oy * p.Wo + ox) =
              float4{v[0], v[1], v[2], v[3]};
          s1[mb] += (v[0] + v[1]) + (v[2] + v[3]);
          s2[mb] += fmaf(v[0], v[0], v[1] * v[1]) + fmaf(v[2], v[2], v[3] * v[3]);
        }
      }
    }
    __syncthreads();                         // the operand tiles are dead: their memory takes the waves' partial sums
    double* red = reinterpret_cast<double*>(smem);       // [wave 4][MB * 16][2]
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      double a = (double)s1[mb], b = (double)s2[mb];
      a += __shfl_xor(a, 16, 64);
      a += __shfl_xor(a, 32, 64);
      b += __shfl_xor(b, 16, 64);
      b += __shfl_xor(b, 32, 64);
      if ((lane >> 4) == 0) {
        red[((wn * MB + mb) * 16 + lane) * 2] = a;
        red[((wn * MB + mb) * 16 + lane) * 2 + 1] = b;
      }
    }
    __syncthreads();
    if (tid < MB * 16) {
      const int co = co0 + tid;
      if (co < p.Cout && p.spart != nullptr) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) {
          a += red[((w4 * MB) * 16 + tid) * 2];
          b += red[((w4 * MB) * 16 + tid) * 2 + 1];
        }
        const long long chunks = (long long)p.tiles_x * p.tiles_y, tile = (long long)tyi * p.tiles_x + txi;
        double* dst = p.spart + (((long long)n0 * p.Cout + co) * chunks + tile) * 2;
        dst[0] = a;
        dst[1] = b;
      }
    }
    return;
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int j0 = wn * (16 * NB) + nb * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int co = co0 + mb * 16 + (lane & 15);
      if (co >= p.Cout) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[mb][nb][r] + bv[mb];
        if (p.act == GANLAB_ACT_LRELU) v[r] = gl_lrelu(v[r], p.slope);
      }
      if (G::XMODE != XSCALAR) {   // TW >= 8, Wo % 4 == 0, ox % 4 == 0: the 4 pixels share a row and are in range together
        const int ty = (j0 >> G::TWL) & (TH - 1), tx = j0 & (TW - 1);
        const int oy = oy0 + ty, ox = ox0 + tx;
        if (n0 < p.in.N && oy < p.Ho && ox < p.Wo)
          *reinterpret_cast<float4*>(p.y + ((long long)n0 * p.Cout + co) * out_plane + (long long)oy * p.Wo + ox) =
              float4{v[0], v[1], v[2], v[3]};
      } else {                     // ragged / tiny geometries (down to 1x1 "pixels = samples"): element by element
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = j0 + r;
          const int ni = j >> (G::TWL + G::THL), ty = (j >> G::TWL) & (TH - 1), tx = j & (TW - 1);
          const int n = n0 + ni, oy = oy0 + ty, ox = ox0 + tx;
          if (n < p.in.N && oy < p.Ho && ox < p.Wo) {
            p.y[((long long)n * p.Cout + co) * out_plane + (long long)oy * p.Wo + ox] = v[r];
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight re-layout: OIHW -> [tap][rows_p][cols_p] (rows = GEMM-K channel, cols = GEMM-M channel)
// ------------------------------------------------------------------------------------------------
__global__ void pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin,
                            int KK, int rows, int cols, int rows_p, int cols_p, int dgrad, float scale) {
  const long long total = (long long)KK * rows_p * cols_p;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(e % cols_p);
    const long long t = e / cols_p;
    const int row = (int)(t % rows_p), tap = (int)(t / rows_p);
    float v = 0.f;
    if (row < rows && col < cols) {
      // fwd:   row = ci, col = co, src tap = tap
      // dgrad: row = co, col = ci, src tap = KK-1-tap
      const int co = dgrad ? row : col, ci = dgrad ? col : row, st = dgrad ? (KK - 1 - tap) : tap;
      v = scale * w[((long long)co * Cin + ci) * KK + st];
    }
    out[e] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient: D[co][ci] (per tap) = sum_px gy[co][px] * Xv[ci][px shifted by tap]
//   MFMA 16x16x4:  A[i=co][k=px] from a [co][px] LDS tile of gy, B[k=px][j=ci] from the halo'd patch.
//   Waves are laid out WM (over co blocks) x WK (over the tile's 4-pixel K-steps); each workgroup
//   walks pixel tiles `split, split+S, ...` (next tile prefetched into registers during the MFMA
//   phase) and finally dumps its accumulators to a private slot of the workspace;
//   wgrad_reduce_kernel sums the slots in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
  PatchArgs in;
  const float* gy;
  float* part;  // [slots][Cout][Cin][KK]
  int Cout, Ho, Wo;
  int tiles_x, tiles_y, tiles_n, tiles_co, tiles_ci, S, xcd;
};

template <int KS_, int NBC_, int WM_, int WK_, int TWL_, int THL_, int NIL_, int XMODE_>
struct WgCfg {
  using G = PatchGeo<KS_, TWL_, THL_, NIL_, XMODE_>;
  static constexpr int KS = KS_, KK = KS_ * KS_, WM = WM_, WK = WK_, NBC = NBC_;
  static constexpr int CO_T = 16 * WM_, CI_T = 16 * NBC_;
  static constexpr int PLANE = pad_mod32(G::NI * G::IMG, 2);
  static constexpr int GP = pad_mod32(G::PX_T, 2);
  static constexpr int GS = CO_T * GP, XS = CI_T * PLANE;
  static constexpr int GVEC = XMODE_ != XSCALAR;
  static constexpr int NGI = GVEC ? CO_T * G::PX_T / 4 : CO_T * G::PX_T;
  static constexpr int GPT = ceil_div_c(NGI, 256);
  static_assert(WM_ * WK_ == 4, "4 waves");
  static_assert(G::PX_T % (4 * WK_) == 0, "K-steps must split evenly over waves");
};

#ifndef GL_WGRAD_MINW
#define GL_WGRAD_MINW 1
#endif
#ifndef GL_WGRAD_THICK_MINW
#define GL_WGRAD_THICK_MINW 1
#endif
template <class Cfg, bool AFF = false>
__global__ __launch_bounds__(256, (GL_WGRAD_THICK_MINW > 1 && Cfg::WM == 4 && Cfg::G::XMODE == XVEC && Cfg::KS == 3 && !AFF) ? GL_WGRAD_THICK_MINW : GL_WGRAD_MINW) void conv_wgrad_kernel(WgradArgs p) {
  using G = typename Cfg::G;
  constexpr int KS = Cfg::KS, KK = Cfg::KK, NBC = Cfg::NBC, WK = Cfg::WK;
  constexpr int RP = G::RP, IMG = G::IMG, PLANE = Cfg::PLANE, GP = Cfg::GP;
  constexpr int TW = G::TW, TH = G::TH, NI = G::NI, CO_T = Cfg::CO_T, CI_T = Cfg::CI_T, PX_T = G::PX_T;
  constexpr int GPT = Cfg::GPT, NGI = Cfg::NGI;
  using XS_t = XStage<G, CI_T, PLANE>;
  __shared__ __attribute__((aligned(16))) float smem[Cfg::GS + Cfg::XS + (AFF ? 2 * CI_T : 0)];
  float* Gs = smem;
  float* Xs = smem + Cfg::GS;
  // AFF (deferred InstanceNorm, see PatchArgs): s | t of this workgroup's CI_T input channels for the CURRENT tile's image
  [[maybe_unused]] float* afftab = smem + Cfg::GS + Cfg::XS;
  [[maybe_unused]] float aff_reg = 0.f;
  static_assert(!AFF || (G::XMODE == XVEC && G::NI == 1 && 2 * CI_T <= 256), "AFF: vector staging, one image per tile");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WK, wk = wave % WK;
  // Workgroups of one split walk the same pixel tiles, each with its own (co, ci) tile pair: gy is shared by the pairs of
  // a co tile, x by those of a ci tile.  Consecutive logical ids = the pairs of one split, and gl_xcd_remap keeps them on
  // one XCD, so those re-reads hit its L2 (with the split index fastest they were 5.9x the operands from HBM).
  int bid = p.xcd ? gl_xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  int split, ci_t, co_t;
  if (p.xcd) {
    const int pairs = p.tiles_ci * p.tiles_co;
    split = bid / pairs;
    bid %= pairs;
    ci_t = bid % p.tiles_ci;
    co_t = bid / p.tiles_ci;
  } else {
    split = bid % p.S;
    bid /= p.S;
    ci_t = bid % p.tiles_ci;
    co_t = bid / p.tiles_ci;
  }
  // (workgroup-uniform; said explicitly so that buffer descriptors and scalar offsets built from them stay in SGPRs - without it
  // every descriptor load of the tile sits in a waterfall loop over the "divergent" channel offset)
  const int co0 = __builtin_amdgcn_readfirstlane(co_t) * CO_T, ci0 = __builtin_amdgcn_readfirstlane(ci_t) * CI_T;
  split = __builtin_amdgcn_readfirstlane(split);
  const int plane = p.in.Hi * p.in.Wi, oplane = p.Ho * p.Wo;

  f32x4 acc[KK][NBC];
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int nb = 0; nb < NBC; ++nb) acc[t][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // tile-independent part of the gy staging descriptors
  int gl_[GPT], gj[GPT], gco[GPT];
#pragma unroll
  for (int i = 0; i < GPT; ++i) {
    const int e = tid + i * 256;
    const int per = Cfg::GVEC ? PX_T / 4 : PX_T;
    const int j = (e % per) * (Cfg::GVEC ? 4 : 1), co = e / per;
    gj[i] = e < NGI ? j : -1;
    gco[i] = co;
    gl_[i] = co * GP + j;
  }

  XS_t xst;
  XRegs<G, XS_t::PT> xr;
  float4 gv[Cfg::GVEC ? GPT : 1];
  float gs[Cfg::GVEC ? 1 : GPT];

  const int n_tiles = p.tiles_n * p.tiles_y * p.tiles_x;
#ifndef GL_WGRAD_DESC
#define GL_WGRAD_DESC 1
#endif
  // Vector staging, one image per tile (the >= 8-wide maps): loads through buffer descriptors over this image's planes - no
  // 64-bit addresses, no exec-mask branches - with everything that does not depend on the tile hoisted out of the tile loop
  // (the kernel is bound by what a wave issues per 64-pixel tile NEXT TO its 288 MFMAs, at two workgroups per CU: descriptor
  // loads alone took 64 -> 64 @256^2 x32 from 1.238 to 1.193 ms, round 4).
  constexpr bool DESCW = GL_WGRAD_DESC && G::XMODE == XVEC && G::NI == 1 && Cfg::GVEC;
  constexpr int XPT = XS_t::PT;
  [[maybe_unused]] int xrel[DESCW ? XPT : 1], xyx[DESCW ? XPT : 1];      // ci*plane + (r-1)*Wi + 4q - LP (or INT_MIN) ; (r-1) | (4q-LP) << 16
  [[maybe_unused]] int grel[DESCW ? GPT : 1], gyx[DESCW ? GPT : 1];      // co*oplane + ty*Wo + tx (or INT_MIN) ; ty | tx << 16
  constexpr int NOITEM = (int)0x80000000;
  if constexpr (DESCW) {
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int e = tid + i * 256;
      const int q = e % G::ROW4;
      const int t = e / G::ROW4;
      const int r = t % G::R, ci = t / G::R;
      xst.loff[i] = (ci * PLANE + r * G::RP + 4 * q) | (ci << 20);
      const int dy = r - G::PADC, dx = 4 * q - G::LP;
      xrel[i] = (e < XS_t::NITEMS && ci0 + ci < p.in.Cin) ? ci * plane + dy * p.in.Wi + dx : NOITEM;
      xyx[i] = (dy & 0xffff) | (dx << 16);
    }
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
      const int j = gj[i];
      const int ty = (j >> G::TWL) & (TH - 1), tx = j & (TW - 1), cog = co0 + gco[i];
      grel[i] = (j >= 0 && cog < p.Cout) ? cog * oplane + ty * p.Wo + tx : NOITEM;
      gyx[i] = ty | (tx << 16);
    }
  }
  auto load_tile_at = [&](int txi, int tyi, int tni) {
    if constexpr (DESCW) {
    const int ox0 = txi * TW, oy0 = tyi * TH, n0 = tni;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in.x + (long long)n0 * p.in.Cin * plane), 0, (unsigned)(p.in.Cin * plane * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.gy + (long long)n0 * p.Cout * oplane), 0, (unsigned)(p.Cout * oplane * 4), 0x00020000);
    const int xbase = oy0 * p.in.Wi + ox0, gbase = oy0 * p.Wo + ox0, soff = ci0 * plane * 4;
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int vy = oy0 + (int)(short)(xyx[i] & 0xffff), vx = ox0 + (xyx[i] >> 16);
      const bool ok = xrel[i] != NOITEM && (unsigned)vy < (unsigned)p.in.Hi && (unsigned)vx < (unsigned)p.in.Wi;
      xst.goff[i] = ok ? xrel[i] + xbase : -1;        // (x_store's affine form asks "inside the image?")
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? (xrel[i] + xbase) * 4 : NOITEM, soff, 0);
      xr.v[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    }
    if constexpr (AFF) {
      if (tid < 2 * CI_T) {
        const int c = ci0 + (tid % CI_T);
        const float* src = tid < CI_T ? p.in.aff_s : p.in.aff_t;
        aff_reg = c < p.in.Cin ? src[(long long)n0 * p.in.Cin + c] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
      const bool ok = grel[i] != NOITEM && oy0 + (gyx[i] & 0xffff) < p.Ho && ox0 + (gyx[i] >> 16) < p.Wo;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_g, ok ? (grel[i] + gbase) * 4 : NOITEM, 0, 0);
      gv[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    }
    }
  };
  auto load_tile = [&](int tile) {
    const int txi = tile % p.tiles_x;
    const int t2 = tile / p.tiles_x;
    const int tyi = t2 % p.tiles_y, tni = t2 / p.tiles_y;
    if constexpr (DESCW) {
      load_tile_at(txi, tyi, tni);
      return;
    }
    const int ox0 = txi * TW, oy0 = tyi * TH, n0 = tni * NI;
    xst.init(p.in, tid, n0, oy0, ox0);
    if constexpr (G::XMODE != XSCALAR)
      x_load<G, CI_T, PLANE>(xr, xst, p.in.x + (long long)n0 * p.in.Cin * plane, ci0, p.in.Cin, plane);
    if constexpr (AFF) {
      if (tid < 2 * CI_T) {
        const int c = ci0 + (tid % CI_T);
        const float* src = tid < CI_T ? p.in.aff_s : p.in.aff_t;
        aff_reg = (c < p.in.Cin && n0 < p.in.N) ? src[(long long)n0 * p.in.Cin + c] : 0.f;
      }
    }
    const float* gb = p.gy + (long long)n0 * p.Cout * oplane;
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
      const int j = gj[i];
      const int ni = j >> (G::TWL + G::THL), ty = (j >> G::TWL) & (TH - 1), tx = j & (TW - 1);
      const int n = n0 + ni, oy = oy0 + ty, ox = ox0 + tx, cog = co0 + gco[i];
      const bool ok = j >= 0 && cog < p.Cout && n < p.in.N && oy < p.Ho && ox < p.Wo;
      const long long off = ((long long)ni * p.Cout + cog) * oplane + oy * p.Wo + ox;
      if (Cfg::GVEC) gv[i] = ok ? *reinterpret_cast<const float4*>(gb + off) : float4{0.f, 0.f, 0.f, 0.f};
      else gs[i] = ok ? gb[off] : 0.f;
    }
  };

  int tile = split;
  // (DESCW) the tile coordinates advance by S's own (x, y, image) digits with carries instead of two divisions per tile
  [[maybe_unused]] int txi = tile % p.tiles_x, tyi = (tile / p.tiles_x) % p.tiles_y, tni = tile / (p.tiles_x * p.tiles_y);
  [[maybe_unused]] const int sdx = p.S % p.tiles_x, sdy = (p.S / p.tiles_x) % p.tiles_y, sdn = p.S / (p.tiles_x * p.tiles_y);
  if (tile < n_tiles) load_tile(tile);
  while (tile < n_tiles) {
    if constexpr (AFF) {      // nobody reads the table between the previous tile's second barrier and this one
      if (tid < 2 * CI_T) afftab[tid] = aff_reg;
    }
    __syncthreads();
    if constexpr (G::XMODE == XSCALAR)
      x_stage_scalar<G, CI_T, PLANE>(xst, p.in, p.in.x + (long long)xst.n0_ * p.in.Cin * plane, ci0, Xs, tid);
    else if constexpr (AFF)
      x_store<G, CI_T, PLANE, false, true>(xr, xst, Xs, tid, afftab, 0, CI_T, p.in.Cin - ci0);
    else
      x_store<G, CI_T, PLANE, false>(xr, xst, Xs, tid);
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
      if (tid + i * 256 < NGI) {
        if (Cfg::GVEC) {  // GP is even but not a multiple of 4: two 8-byte stores
          *reinterpret_cast<float2*>(Gs + gl_[i]) = float2{gv[i].x, gv[i].y};
          *reinterpret_cast<float2*>(Gs + gl_[i] + 2) = float2{gv[i].z, gv[i].w};
        } else {
          Gs[gl_[i]] = gs[i];
        }
      }
    }
    __syncthreads();
    const int next = tile + p.S;
    if constexpr (DESCW) {
      txi += sdx;
      int carry = txi >= p.tiles_x ? 1 : 0;
      txi -= carry * p.tiles_x;
      tyi += sdy + carry;
      carry = tyi >= p.tiles_y ? 1 : 0;
      tyi -= carry * p.tiles_y;
      tni += sdn + carry;
      if (next < n_tiles) load_tile_at(txi, tyi, tni);  // in flight during the MFMA phase
    } else {
      if (next < n_tiles) load_tile(next);
    }
    for (int q = wk; q < PX_T / 4; q += WK) {
      const int j = 4 * q + (lane >> 4);
      const int ni = j >> (G::TWL + G::THL), ty = (j >> G::TWL) & (TH - 1), tx = j & (TW - 1);
      const int poff = ni * IMG + ty * RP + tx + G::XOFF + (lane & 15) * PLANE;
      const float a = Gs[(wm * 16 + (lane & 15)) * GP + j];
#pragma unroll
      for (int tap = 0; tap < KK; ++tap) {
        const int toff = (tap / KS) * RP + (tap % KS);
#pragma unroll
        for (int nb = 0; nb < NBC; ++nb) {
          const float b = Xs[nb * 16 * PLANE + poff + toff];
          acc[tap][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[tap][nb], 0, 0, 0);
        }
      }
    }
    tile = next;
  }
  // dump: slot = split*WK + wk ; D rows = co (lane>>4)*4+r, col = ci (lane&15)
  const int slot = split * WK + wk;
  float* dst = p.part + (long long)slot * p.Cout * p.in.Cin * KK;
#pragma unroll
  for (int tap = 0; tap < KK; ++tap)
#pragma unroll
    for (int nb = 0; nb < NBC; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wm * 16 + (lane >> 4) * 4 + r;
        const int ci = ci0 + nb * 16 + (lane & 15);
        if (co < p.Cout && ci < p.in.Cin) dst[((long long)co * p.in.Cin + ci) * KK + tap] = acc[tap][nb][r];
      }
}

// Deterministic slot reduction.  grid = (ceil(n/256), groups): block (bx, g) sums slots
// g, g+groups, ... of outputs [256*bx, 256*bx+256) into out[g][.]; with groups == 1 the result is
// final (scaled), otherwise a second launch with slots = groups finishes it.
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long long n,
                                    int slots, int groups, float scale) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int g = blockIdx.y;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = g;
  for (; k + 3 * groups < slots; k += 4 * groups) {
    s0 += part[(long long)k * n + i];
    s1 += part[(long long)(k + groups) * n + i];
    s2 += part[(long long)(k + 2 * groups) * n + i];
    s3 += part[(long long)(k + 3 * groups) * n + i];
  }
  for (; k < slots; k += groups) s0 += part[(long long)k * n + i];
  out[(long long)g * n + i] = ((s0 + s1) + (s2 + s3)) * scale;
}

// ------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------
inline int cin_pad(int ks) { return ks == 1 ? 32 : 16; }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

PatchArgs make_patch(const float* x, int N, int Cin, int Hi, int Wi, int pad, int up) {
  PatchArgs a{};
  a.x = x; a.N = N; a.Cin = Cin; a.Hi = Hi; a.Wi = Wi;
  a.Hv = up ? 2 * Hi : Hi; a.Wv = up ? 2 * Wi : Wi; a.pad = pad; a.up = up;
  return a;
}

// staging mode for a patch source: vector paths need 16-byte aligned rows and "same" padding
int pick_xmode(const PatchArgs& in, int ks, int out_w) {
  if (out_w < 16 || (in.Wi & 3) || !aligned16(in.x) || in.pad != (ks - 1) / 2) return XSCALAR;
  if (in.up) return ks == 3 ? XVECUP : XSCALAR;
  return XVEC;
}

inline bool roll_col_enabled() {
  static const bool on = [] { const char* e = getenv("GANLAB_ROLL_COL"); return !(e && e[0] == '0'); }();
  return on;
}

template <class Cfg>
int launch_fwd(ConvArgs a, hipStream_t st) {
  using G = typename Cfg::G;
  a.tiles_x = ceil_div(a.Wo, G::TW);
  a.tiles_y = ceil_div(a.Ho, G::TH);
  a.tiles_n = ceil_div(a.in.N, G::NI);
  a.tiles_co = ceil_div(a.Cout, Cfg::CO_T);
  // Strips: a workgroup walks `strip` consecutive tiles of one tile row, prefetching the next tile while the
  // MFMAs of the current one run.  Only where there are plenty of tiles (thin, large layers): keep >= ~8
  // rounds of 3 workgroups per CU in the grid, and split the row evenly.
  const long long tiles = (long long)a.tiles_x * a.tiles_y * a.tiles_n * a.tiles_co;
  if constexpr (Cfg::STRIP) {
    // the strip kernel takes full tiles, one K-chunk, whole 16-channel blocks and 31-bit byte offsets
    const bool ok = a.Cin_p == Cfg::CI_T && a.Wo % G::TW == 0 && a.Ho % G::TH == 0 && a.Cout % Cfg::CO_T == 0 &&
                    (long long)a.in.Cin * a.in.Hi * a.in.Wi * 4 < 0x7fffffffLL &&
                    (long long)a.Cout * a.Ho * a.Wo * 4 < 0x7fffffffLL;
    if (ok && a.mask == nullptr) {             // (the strip / rolling kernels have no mask epilogue)
      int k = 1;                               // strips per tile row
      while (k < a.tiles_x && tiles / ceil_div(a.tiles_x, k) < 6144) ++k;
      a.strip = ceil_div(a.tiles_x, k);
      a.strips_x = ceil_div(a.tiles_x, a.strip);
      const long long sgrid = (long long)a.strips_x * a.tiles_y * a.tiles_n * a.tiles_co;
      if (sgrid <= 0 || sgrid > 0x7fffffffLL) return GANLAB_EINVAL;
      if constexpr (Cfg::KS == 3 && G::XMODE == XVEC) {   // weights in registers, double-buffered patch
#ifdef GL_ABL_OLDSTRIP   // ablation builds (tools/strip_ablate.py): horizontal strips, weights in LDS
        if (true) {
          GL_LAUNCH(conv_fwd_strip_kernel<Cfg>, dim3((unsigned)sgrid), dim3(256), 0, st, a);
        } else
#endif
        if (a.Wo % RW_TW == 0 && a.Ho % RW_ROWS == 0 && a.Cout <= 16 && a.tiles_co == 1 && roll_col_enabled() &&
            a.in.up == 0 && a.in.pad == 1 && a.in.aff_s == nullptr) {
          // rolling window, wave = column block (conv_roll_blur.hip, BLUR = false): half the LDS operand reads of the
          // wave = row kernel below - 1.336 against 1.403 ms on the 16 -> 16 layer at 1024^2 x 32
          return gl_roll_blur_launch(a.in.x, a.wp, a.bias, a.y, nullptr, a.in.N, a.in.Cin, a.Cout, a.in.Hi, a.in.Wi, a.Cin_p,
                                     a.Cout_p, a.bias_scale, a.slope, st, 0, a.act);
        } else if (a.Wo % RW_TW == 0 && a.Ho % RW_ROWS == 0 && a.Cout <= 16 && a.tiles_co == 1) {
          // rolling window: 64-pixel column strips, 4 output rows per step
          ConvArgs r = a;
          r.tiles_x = a.Wo / RW_TW;
          r.tiles_y = a.Ho / RW_ROWS;                      // steps per column
          const long long cols = (long long)r.tiles_x * a.tiles_n;
          int kk = 1;                                      // strips per column: >= ~4096 workgroups, >= 8 steps each
          while (kk < r.tiles_y && cols * kk < 4096 && ceil_div(r.tiles_y, kk + 1) >= 8) ++kk;
          r.strip = ceil_div(r.tiles_y, kk);
          r.strips_x = ceil_div(r.tiles_y, r.strip);
          const long long rgrid = cols * r.strips_x;
          if (rgrid <= 0 || rgrid > 0x7fffffffLL) return GANLAB_EINVAL;
          GL_LAUNCH(conv_fwd_roll_kernel, dim3((unsigned)rgrid), dim3(256), 0, st, r);
        } else {   // vertical strips: `strip` tiles down a column, `strips_x` strips per column
          int ky = 1;
          while (ky < a.tiles_y && tiles / ceil_div(a.tiles_y, ky) < 6144) ++ky;
          a.strip = ceil_div(a.tiles_y, ky);
          a.strips_x = ceil_div(a.tiles_y, a.strip);
          const long long vgrid = (long long)a.tiles_x * a.strips_x * a.tiles_n * a.tiles_co;
          if (vgrid <= 0 || vgrid > 0x7fffffffLL) return GANLAB_EINVAL;
          GL_LAUNCH(conv_fwd_strip2_kernel<Cfg>, dim3((unsigned)vgrid), dim3(256), 0, st, a);
        }
      } else {
        GL_LAUNCH(conv_fwd_strip_kernel<Cfg>, dim3((unsigned)sgrid), dim3(256), 0, st, a);
      }
      return GL_CHECK_LAUNCH();
    }
  }
  if (tiles <= 0 || tiles > 0x7fffffffLL) return GANLAB_EINVAL;
  if (a.ksplit > 1) {             // split-K: the small geometries only - scalar-staged, or 16x16 tiles (splitk_plan)
    if constexpr (G::XMODE == XSCALAR || (G::XMODE == XVEC && G::TW == 16)) {
      a.ksplit_ci = round_up_c(ceil_div(a.Cin_p / Cfg::CI_T, a.ksplit) * Cfg::CI_T, Cfg::CI_T);
      GL_LAUNCH((conv_fwd_kernel<Cfg, false, true>), dim3((unsigned)(tiles * a.ksplit)), dim3(256), 0, st, a);
      return GL_CHECK_LAUNCH();
    } else {
      return GANLAB_EUNSUPPORTED;
    }
  }
  if (a.mask != nullptr) {        // output mask: the 3x3 vector-staged kernels only (ganlab_conv_dgrad_mask_supported)
    if constexpr (Cfg::KS == 3 && G::XMODE == XVEC) {
      GL_LAUNCH((conv_fwd_kernel<Cfg, true>), dim3((unsigned)tiles), dim3(256), 0, st, a);
      return GL_CHECK_LAUNCH();
    } else {
      return GANLAB_EUNSUPPORTED;
    }
  }
  GL_LAUNCH(conv_fwd_kernel<Cfg>, dim3((unsigned)tiles), dim3(256), 0, st, a);
  return GL_CHECK_LAUNCH();
}

template <int KS, int MB>
int dispatch_geom(const ConvArgs& a, hipStream_t st) {
  if (a.Ho == 1 && a.Wo == 1) return launch_fwd<FwdCfg<KS, MB, 0, 0, 6, XSCALAR>>(a, st);
  const int mode = (aligned16(a.wp)) ? pick_xmode(a.in, KS, a.Wo) : XSCALAR;
  if constexpr (GL_THICK_NB2 && KS == 3 && MB == 4) {
    if (a.Wo >= 16 && mode == XVEC) return launch_fwd<ThickCfg>(a, st);
  }
#ifndef GL_MB2_NB2          // the same halving for the 32-channel tile (79 registers): measured slower, 32 -> 32 @512^2 1.237 -> 1.286 ms
#define GL_MB2_NB2 0
#endif
  if constexpr (GL_MB2_NB2 && KS == 3 && MB == 2) {
    if (a.Wo >= 16 && mode == XVEC) return launch_fwd<FwdCfg<3, 2, 4, 3, 0, XVEC>>(a, st);
  }
  if (a.Wo >= 32) {
    if (mode == XVEC) return launch_fwd<FwdCfg<KS, MB, 5, 3, 0, XVEC>>(a, st);
    if (KS == 3 && mode == XVECUP) return launch_fwd<FwdCfg<3, MB, 5, 3, 0, XVECUP>>(a, st);
    return launch_fwd<FwdCfg<KS, MB, 5, 3, 0, XSCALAR>>(a, st);
  }
  if (a.Wo >= 16) {
    if (mode == XVEC) return launch_fwd<FwdCfg<KS, MB, 4, 4, 0, XVEC>>(a, st);
    if (KS == 3 && mode == XVECUP) return launch_fwd<FwdCfg<3, MB, 4, 4, 0, XVECUP>>(a, st);
    return launch_fwd<FwdCfg<KS, MB, 4, 4, 0, XSCALAR>>(a, st);
  }
  // (the 64-channel tile with a second accumulator set does not fit the element-wise staged multi-image tiles'
  // registers: those take the 32-channel tile)
  constexpr int MBS = (GL_ACC_DUMP && KS == 3 && MB == 4) ? 2 : MB;
  if (a.Wo >= 8) return launch_fwd<FwdCfg<KS, MBS, 3, 3, 2, XSCALAR>>(a, st);
  return launch_fwd<FwdCfg<KS, MBS, 2, 2, 4, XSCALAR>>(a, st);
}

// pixel tiles of the 64-channel-tile kernel where they differ from the narrower kernels' (ThickCfg: 16 x 8 pixels)
inline long long px_tiles_mb4(long long px_tiles, int N, int Ho, int Wo, int ks) {
  if (GL_THICK_NB2 && ks == 3 && Wo >= 16) return (long long)ceil_div(Wo, 16) * ceil_div(Ho, 8) * N;
  return px_tiles;
}

template <int KS>
int dispatch_co(const ConvArgs& a, hipStream_t st) {
  if (a.Cout <= 16) return dispatch_geom<KS, 1>(a, st);
  if (a.Cout <= 32) return dispatch_geom<KS, 2>(a, st);
  // Small spatial extents (4x4 / 8x8 / linear layers at 512 channels) give only a handful of pixel
  // tiles: narrower channel tiles put more workgroups on the 256 CUs (the layer is latency-bound).
  long long px_tiles;
  if (a.Ho == 1 && a.Wo == 1) px_tiles = ceil_div(a.in.N, 64);
  else if (a.Wo >= 32) px_tiles = (long long)ceil_div(a.Wo, 32) * ceil_div(a.Ho, 8) * a.in.N;
  else if (a.Wo >= 16) px_tiles = (long long)ceil_div(a.Wo, 16) * ceil_div(a.Ho, 16) * a.in.N;
  else if (a.Wo >= 8) px_tiles = (long long)ceil_div(a.Wo, 8) * ceil_div(a.Ho, 8) * ceil_div(a.in.N, 4);
  else px_tiles = (long long)ceil_div(a.Wo, 4) * ceil_div(a.Ho, 4) * ceil_div(a.in.N, 16);
  if (px_tiles_mb4(px_tiles, a.in.N, a.Ho, a.Wo, KS) * ceil_div(a.Cout, 64) >= 256) return dispatch_geom<KS, 4>(a, st);
  if (px_tiles * ceil_div(a.Cout, 32) >= 256) return dispatch_geom<KS, 2>(a, st);
  return dispatch_geom<KS, 1>(a, st);
}

// ------------------------------------------------------------------------------------------------
// fromRGB / toRGB: 1x1 convolutions with 3 channels on one side (progan/architectures.py:286-292, :150-160;
// stylegan/architectures.py torgb).  48 multiply-adds per pixel against 64+ bytes of traffic: HBM-bound streaming
// work, not MFMA work - each thread owns 4 consecutive pixels, the 3 x C weights are wave-uniform scalars.
//   few_to_many: y[co] = act(sum_ci w[ci][co] * x[ci] + b[co]),  Cin <= 4 (fromRGB forward, toRGB input gradient)
//   many_to_few: the same with Cout <= 4                          (toRGB forward, fromRGB input gradient)
//   cross_sums:  part[blk][b][s] = sum_px big[b] * small[s]        (both weight gradients; finished by a fixed-order sum)
// wp is the packed [Cin_p][Cout_p] layout of pack_kernel (scale folded in).
// ------------------------------------------------------------------------------------------------
struct PwArgs {
  const float* x;
  const float* wp;
  const float* bias;
  float* y;
  int N, Cin, Cout, Cout_p;
  long long hw4;      // float4 per plane
  float bias_scale, slope;
  int act;
  // LeakyReLU derivative folded into the "many" side: factor 1 where mask > 0 else mslope.  few_to_many multiplies its
  // OUTPUT (mask: N x Cout planes), many_to_few its INPUT (mask: N x Cin planes) - the fromRGB activation's backward
  const float* mask;
  float mslope;
  // masks as sign BITS (bit e of the NCHW-linear element index e, 32 per word; planes of whole words: H*W % 32 == 0):
  // mask_bits: `mask` points to such words instead of floats; out_bits (few_to_many only): also write the sign bits of y
  int mask_bits;
  unsigned* out_bits;
};

__device__ __forceinline__ float4 pw_masked(float4 v, float4 m, float sl) {
  v.x = m.x > 0.f ? v.x : v.x * sl; v.y = m.y > 0.f ? v.y : v.y * sl;
  v.z = m.z > 0.f ? v.z : v.z * sl; v.w = m.w > 0.f ? v.w : v.w * sl;
  return v;
}

__device__ __forceinline__ float4 pw_masked_bits(float4 v, const unsigned* __restrict__ bits, long long e, float sl) {
  const unsigned nb = (bits[e >> 5] >> (unsigned)(e & 31)) & 0xFu;      // e % 4 == 0
  v.x = (nb & 1u) ? v.x : v.x * sl; v.y = (nb & 2u) ? v.y : v.y * sl;
  v.z = (nb & 4u) ? v.z : v.z * sl; v.w = (nb & 8u) ? v.w : v.w * sl;
  return v;
}

__global__ __launch_bounds__(256) void pw_few_to_many_kernel(PwArgs p) {
  const long long total = (long long)p.N * p.hw4;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / p.hw4, q = i - n * p.hw4;
    const float4* xb = reinterpret_cast<const float4*>(p.x) + n * p.Cin * p.hw4 + q;
    float4 xv[4];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) xv[ci] = ci < p.Cin ? xb[(long long)ci * p.hw4] : float4{0.f, 0.f, 0.f, 0.f};
    float4* yb = reinterpret_cast<float4*>(p.y) + n * p.Cout * p.hw4 + q;
    for (int co = 0; co < p.Cout; ++co) {
      const float b = p.bias != nullptr ? p.bias[co] * p.bias_scale : 0.f;
      float4 a = gl_few_dot(xv, p.wp, p.Cin, p.Cout_p, co, b);
      if (p.act == GANLAB_ACT_LRELU) {
        a.x = gl_lrelu(a.x, p.slope); a.y = gl_lrelu(a.y, p.slope); a.z = gl_lrelu(a.z, p.slope); a.w = gl_lrelu(a.w, p.slope);
      }
      if (p.mask != nullptr) {
        if (p.mask_bits)
          a = pw_masked_bits(a, reinterpret_cast<const unsigned*>(p.mask), ((n * p.Cout + co) * p.hw4 + q) * 4, p.mslope);
        else
          a = pw_masked(a, (reinterpret_cast<const float4*>(p.mask) + n * p.Cout * p.hw4 + q)[(long long)co * p.hw4], p.mslope);
      }
      yb[(long long)co * p.hw4] = a;
      if (p.out_bits != nullptr) {
        // sign bits of this output: 8 consecutive lanes hold 32 consecutive pixels of one plane (hw4 % 8 == 0, so a group of
        // 8 lanes is inside one image and active or inactive as a whole)
        unsigned wv = ((a.x > 0.f ? 1u : 0u) | (a.y > 0.f ? 2u : 0u) | (a.z > 0.f ? 4u : 0u) | (a.w > 0.f ? 8u : 0u))
                      << (4 * (threadIdx.x & 7));
        wv |= __shfl_xor(wv, 1, 64);
        wv |= __shfl_xor(wv, 2, 64);
        wv |= __shfl_xor(wv, 4, 64);
        if ((threadIdx.x & 7) == 0) p.out_bits[(((n * p.Cout + co) * p.hw4 + q) * 4) >> 5] = wv;
      }
    }
  }
}

__global__ __launch_bounds__(256) void pw_many_to_few_kernel(PwArgs p) {
  const long long total = (long long)p.N * p.hw4;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / p.hw4, q = i - n * p.hw4;
    const float4* xb = reinterpret_cast<const float4*>(p.x) + n * p.Cin * p.hw4 + q;
    float4 a[4];
#pragma unroll
    for (int co = 0; co < 4; ++co) {
      const float b = (p.bias != nullptr && co < p.Cout) ? p.bias[co] * p.bias_scale : 0.f;
      a[co] = float4{b, b, b, b};
    }
    const float4* mb = (p.mask != nullptr && !p.mask_bits) ? reinterpret_cast<const float4*>(p.mask) + n * p.Cin * p.hw4 + q
                                                           : nullptr;
    for (int ci = 0; ci < p.Cin; ++ci) {
      float4 v = xb[(long long)ci * p.hw4];
      if (p.mask != nullptr) {
        if (p.mask_bits)
          v = pw_masked_bits(v, reinterpret_cast<const unsigned*>(p.mask), ((n * p.Cin + ci) * p.hw4 + q) * 4, p.mslope);
        else
          v = pw_masked(v, mb[(long long)ci * p.hw4], p.mslope);
      }
#pragma unroll
      for (int co = 0; co < 4; ++co) {
        const float w = co < p.Cout ? p.wp[(long long)ci * p.Cout_p + co] : 0.f;
        a[co].x += w * v.x; a[co].y += w * v.y; a[co].z += w * v.z; a[co].w += w * v.w;
      }
    }
    float4* yb = reinterpret_cast<float4*>(p.y) + n * p.Cout * p.hw4 + q;
#pragma unroll
    for (int co = 0; co < 4; ++co) {
      if (co >= p.Cout) continue;
      float4 v = a[co];
      if (p.act == GANLAB_ACT_LRELU) {
        v.x = gl_lrelu(v.x, p.slope); v.y = gl_lrelu(v.y, p.slope); v.z = gl_lrelu(v.z, p.slope); v.w = gl_lrelu(v.w, p.slope);
      }
      yb[(long long)co * p.hw4] = v;
    }
  }
}

// part[blk][b][s]: b < B (<= 16 per launch: block of the "big" tensor's channels b0 .. b0+15), s < 4
// mask (shape of big): big is multiplied by the LeakyReLU derivative of mask on load; ones: column s = S of the small
// tensor is taken as 1, so acc[b][S] = sum_px big[b] (the bias gradient of fromRGB comes out of the same pass)
__global__ __launch_bounds__(256) void pw_cross_sums_kernel(const float* __restrict__ big, const float* __restrict__ small,
                                                            float* __restrict__ part, int N, int B, int b0, int S,
                                                            long long hw4, int Btot,
                                                            const float* __restrict__ mask = nullptr, float mslope = 1.f,
                                                            int ones = 0, int mask_bits = 0) {
  __shared__ float red[4][64];
  float acc[16][4];
#pragma unroll
  for (int b = 0; b < 16; ++b)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[b][s] = 0.f;
  const long long total = (long long)N * hw4;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / hw4, q = i - n * hw4;
    const float4* sb = reinterpret_cast<const float4*>(small) + n * S * hw4 + q;
    const float4* bb = reinterpret_cast<const float4*>(big) + (n * Btot + b0) * hw4 + q;
    float4 sv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float f = (ones && s == S) ? 1.f : 0.f;
      sv[s] = s < S ? sb[(long long)s * hw4] : float4{f, f, f, f};
    }
    float4 bv[16];        // all loads first (independent, in flight together), then the multiply-adds
#pragma unroll
    for (int b = 0; b < 16; ++b) bv[b] = b < B ? bb[(long long)b * hw4] : float4{0.f, 0.f, 0.f, 0.f};
    if (mask != nullptr && mask_bits) {
#pragma unroll
      for (int b = 0; b < 16; ++b)
        if (b < B)
          bv[b] = pw_masked_bits(bv[b], reinterpret_cast<const unsigned*>(mask), ((n * Btot + b0 + b) * hw4 + q) * 4, mslope);
    } else if (mask != nullptr) {
      const float4* mk = reinterpret_cast<const float4*>(mask) + (n * Btot + b0) * hw4 + q;
#pragma unroll
      for (int b = 0; b < 16; ++b)
        if (b < B) bv[b] = pw_masked(bv[b], mk[(long long)b * hw4], mslope);
    }
#pragma unroll
    for (int b = 0; b < 16; ++b)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc[b][s] += (bv[b].x * sv[s].x + bv[b].y * sv[s].y) + (bv[b].z * sv[s].z + bv[b].w * sv[s].w);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int b = 0; b < 16; ++b)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float v = gl_wave_sum(acc[b][s]);
      if (lane == 0) red[wv][b * 4 + s] = v;
    }
  __syncthreads();
  if (threadIdx.x < 64)
    part[(long long)blockIdx.x * 64 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// gw[co][ci] (OIHW, 1x1) = scale * sum_blk part[blk][b][s];  big_is_out: (b, s) = (co - b0, ci) else (ci - b0, co)
__global__ void pw_cross_finish_kernel(const float* __restrict__ part, float* __restrict__ gw, int blocks, int B, int b0,
                                       int S, int Cout, int Cin, int big_is_out, float scale,
                                       float* __restrict__ gb = nullptr, float bias_scale = 1.f) {
  __shared__ float red[16][64];
  const int t = threadIdx.x & 63, j = threadIdx.x >> 6;       // t = b * 4 + s; 16 groups of 64 threads split the blocks
  float s0 = 0.f, s1 = 0.f;
  int k = j;
  for (; k + 16 < blocks; k += 32) {
    s0 += part[(long long)k * 64 + t];
    s1 += part[(long long)(k + 16) * 64 + t];
  }
  if (k < blocks) s0 += part[(long long)k * 64 + t];
  red[j][t] = s0 + s1;
  __syncthreads();
  if (j != 0) return;
  const int b = t >> 2, s = t & 3;
  const bool is_bias = gb != nullptr && s == S;                // the "ones" column of pw_cross_sums_kernel
  if (b >= B || (s >= S && !is_bias)) return;
  float sum = 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g) sum += red[g][t];               // fixed order: deterministic
  if (is_bias) {
    gb[b0 + b] = sum * bias_scale;
    return;
  }
  const int co = big_is_out ? b0 + b : s, ci = big_is_out ? s : b0 + b;
  gw[(long long)co * Cin + ci] = sum * scale;
}

inline bool pw_small_ok(int Cin, int Cout, int ks, int pad, int up, int Hi, int Wi, const void* a, const void* b) {
  const long long hw = (long long)Hi * Wi;
  return ks == 1 && pad == 0 && !up && (hw & 3) == 0 && hw >= 4096 && ((Cin <= 4 && Cout <= 64) || (Cout <= 4 && Cin <= 64)) &&
         aligned16(a) && aligned16(b);
}

// y = act(sum_s part[s] + bias): finishes a split-K launch (fixed summation order)
__global__ void splitk_finish_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                     float* __restrict__ y, int S, long long total, int C, int HW, float bias_scale,
                                     int act, float slope) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  float v = part[i];
  for (int k = 1; k < S; ++k) v += part[(long long)k * total + i];
  if (bias != nullptr) v += bias[(int)((i / HW) % C)] * bias_scale;
  if (act == GANLAB_ACT_LRELU) v = gl_lrelu(v, slope);
  y[i] = v;
}

// Split-K factor for a conv with output (N, Cout, Ho, Wo) contracting Cin channels: only the small geometries that
// dispatch_geom runs on the scalar-staged kernels (Wo < 16 or 1x1), and only when the plain launch would put fewer than
// ~2 workgroups on a CU.  Mirrors dispatch_co's channel-tile choice.
int splitk_plan(int N, int Cin, int Cout, int Ho, int Wo, int ks) {
  long long px_tiles;
  if (Ho == 1 && Wo == 1) px_tiles = ceil_div(N, 64);
  else if (Wo >= 32) return 1;
  else if (Wo >= 16) px_tiles = (long long)ceil_div(Wo, 16) * ceil_div(Ho, 16) * N;
  else if (Wo >= 8) px_tiles = (long long)ceil_div(Wo, 8) * ceil_div(Ho, 8) * ceil_div(N, 4);
  else px_tiles = (long long)ceil_div(Wo, 4) * ceil_div(Ho, 4) * ceil_div(N, 16);
  int mb = 1;
  if (Cout <= 16) mb = 1;
  else if (Cout <= 32) mb = 2;
  else if (px_tiles_mb4(px_tiles, N, Ho, Wo, ks) * ceil_div(Cout, 64) >= 256) mb = 4;
  else if (px_tiles * ceil_div(Cout, 32) >= 256) mb = 2;
  if (GL_ACC_DUMP && ks == 3 && mb == 4 && Wo < 16) mb = 2;      // as dispatch_geom does
  if (mb == 4) px_tiles = px_tiles_mb4(px_tiles, N, Ho, Wo, ks);
  const long long wgs = px_tiles * ceil_div(Cout, 16 * mb);
  // K-chunk of the kernel that will run: 32 (1x1), 16 (vector-staged 16x16 tiles with <= 32 output channels per
  // workgroup), else 8; a scalar-staged fallback for unaligned pointers halves it, which keeps every split non-empty
  const int ci_t = ks == 1 ? 32 : ((Wo >= 16 && mb == 1) ? 16 : 8);
  const int chunks = round_up_c(Cin, cin_pad(ks)) / ci_t;
#ifndef GL_SPLITK_TARGET
#define GL_SPLITK_TARGET 512
#endif
  long long S = (GL_SPLITK_TARGET + wgs - 1) / wgs;
  const int cap = (Ho == 1 && Wo == 1) ? 8 : 4;
  if (S > cap) S = cap;
  if (S > chunks / 4) S = chunks / 4;      // at least four K-chunks per workgroup
  while (S > 1 && (S - 1) * ceil_div(chunks, (int)S) >= chunks) --S;   // no empty split
  return S < 2 ? 1 : (int)S;
}

int run_conv(const float* x, const float* wp, const float* bias, float* y, int N, int Cin, int Hi, int Wi,
             int Cout, int ks, int pad, int up, float bias_scale, int act, float slope, hipStream_t st,
             const float* pw_mask = nullptr, float pw_mslope = 1.f, const float* out_mask = nullptr,
             float* splitk_ws = nullptr, int ksplit = 1, int pw_mask_bits = 0, unsigned* pw_out_bits = nullptr) {
  if (!x || !wp || !y || N <= 0 || Cin <= 0 || Cout <= 0 || Hi <= 0 || Wi <= 0) return GANLAB_EINVAL;
  ConvArgs a{};
  a.in = make_patch(x, N, Cin, Hi, Wi, pad, up);
  a.wp = wp; a.bias = bias; a.y = y;
  a.Cout = Cout;
  a.Ho = a.in.Hv + 2 * pad - ks + 1; a.Wo = a.in.Wv + 2 * pad - ks + 1;
  if (a.Ho <= 0 || a.Wo <= 0) return GANLAB_EINVAL;
  if ((long long)Cin * a.in.Hi * a.in.Wi * 16 >= 0x7fffffffLL) return GANLAB_EINVAL;  // int offsets per tile
  a.Cin_p = round_up_c(Cin, cin_pad(ks)); a.Cout_p = round_up_c(Cout, 64);
  a.bias_scale = bias_scale; a.slope = slope; a.act = act;
  a.mask = out_mask; a.mslope = pw_mslope;
  if (out_mask == nullptr && pw_small_ok(Cin, Cout, ks, pad, up, Hi, Wi, x, y)) {      // fromRGB / toRGB: HBM-bound streaming kernels
    PwArgs q{};
    q.x = x; q.wp = wp; q.bias = bias; q.y = y;
    q.N = N; q.Cin = Cin; q.Cout = Cout; q.Cout_p = a.Cout_p;
    q.hw4 = (long long)Hi * Wi / 4;
    q.bias_scale = bias_scale; q.slope = slope; q.act = act;
    q.mask = pw_mask; q.mslope = pw_mslope;
    q.mask_bits = pw_mask_bits; q.out_bits = pw_out_bits;
    if ((pw_mask_bits || pw_out_bits != nullptr) && ((q.hw4 & 7) != 0 || (pw_out_bits != nullptr && Cin > 4)))
      return GANLAB_EUNSUPPORTED;
    const long long items = (long long)N * q.hw4;
    const unsigned blocks = (unsigned)((items + 255) / 256 < 256 * 16 ? (items + 255) / 256 : 256 * 16);
    if (Cin <= 4) GL_LAUNCH(pw_few_to_many_kernel, dim3(blocks), dim3(256), 0, st, q);
    else GL_LAUNCH(pw_many_to_few_kernel, dim3(blocks), dim3(256), 0, st, q);
    return GL_CHECK_LAUNCH();
  }
  if (pw_mask != nullptr || pw_out_bits != nullptr) return GANLAB_EUNSUPPORTED;   // only the streaming 1x1 kernels fold the mask in
  if (ksplit > 1) {
    if (splitk_ws == nullptr || out_mask != nullptr || up) return GANLAB_EINVAL;
    const long long total = (long long)N * Cout * a.Ho * a.Wo;
    a.y = splitk_ws; a.bias = nullptr; a.act = GANLAB_ACT_NONE;
    a.ksplit = ksplit; a.ksplit_stride = total;
    const int rc = ks == 1 ? dispatch_co<1>(a, st) : dispatch_co<3>(a, st);
    if (rc != GANLAB_OK) return rc;
    GL_LAUNCH(splitk_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float*)splitk_ws,
              bias, y, ksplit, total, Cout, a.Ho * a.Wo, bias_scale, act, slope);
    return GL_CHECK_LAUNCH();
  }
  if (ks == 1) return dispatch_co<1>(a, st);
  if (ks == 3) return dispatch_co<3>(a, st);
  return GANLAB_EINVAL;
}

// ---- weight gradient ----
enum WgGeom { WG_A, WG_B, WG_C, WG_D, WG_E };
struct WgPlan { int thin, nbc, geom, xmode; int tiles_x, tiles_y, tiles_n, tiles_co, tiles_ci, S, slots; };

inline WgGeom wg_geom(int Ho, int Wo, int thin) {
  if (Ho == 1 && Wo == 1) return WG_E;
  if (!thin) return (Wo >= 8) ? WG_A : WG_D;
  if (Wo >= 32) return WG_A;
  if (Wo >= 16) return WG_B;
  if (Wo >= 8) return WG_C;
  return WG_D;
}

template <class Cfg>
void fill_plan(WgPlan& pl, int N, int Cin, int Cout, int Ho, int Wo) {
  using G = typename Cfg::G;
  pl.tiles_x = ceil_div(Wo, G::TW); pl.tiles_y = ceil_div(Ho, G::TH); pl.tiles_n = ceil_div(N, G::NI);
  pl.tiles_co = ceil_div(Cout, Cfg::CO_T); pl.tiles_ci = ceil_div(Cin, Cfg::CI_T);
  const long long n_tiles = (long long)pl.tiles_x * pl.tiles_y * pl.tiles_n;
  const long long base = (long long)pl.tiles_co * pl.tiles_ci;
  // ONE full round of what fits on a CU - three workgroups of the 16-channel thin configuration (43 KB of LDS, 122
  // VGPRs), two of the others: no tail round, and the fewest slots to reduce afterwards (same-box A/B: 1024 -> 768 resp.
  // 512 workgroups, -1.3 and -1.2 ms per step)
  const long long target = (Cfg::WM == 1 && Cfg::NBC == 1) ? 768 : 512;
  long long S = (target + base - 1) / base;
  if (S > n_tiles) S = n_tiles;
  if (S < 1) S = 1;
  pl.S = (int)S;
  pl.slots = pl.S * Cfg::WK;
}

// thin : CO_T = 16, waves split the pixel K-steps (WM=1, WK=4), 256-pixel tiles, 16 or 32 input channels
// thick: CO_T = 64, one co block per wave (WM=4, WK=1), 64-pixel tiles, 32 input channels
template <int KS, int NBC, int XM> using WgThinA = WgCfg<KS, NBC, 1, 4, 5, 3, 0, XM>;    // 32x8
template <int KS, int NBC, int XM> using WgThinB = WgCfg<KS, NBC, 1, 4, 4, 4, 0, XM>;    // 16x16
template <int KS, int NBC> using WgThinC = WgCfg<KS, NBC, 1, 4, 3, 3, 2, XSCALAR>;       // 8x8 x4 images
template <int KS, int NBC> using WgThinD = WgCfg<KS, NBC, 1, 4, 2, 2, 4, XSCALAR>;       // 4x4 x16 images
template <int KS, int NBC> using WgThinE = WgCfg<KS, NBC, 1, 4, 0, 0, 6, XSCALAR>;       // 1x1 x64 samples
#ifndef GL_WGRAD_THICK_NBC
#define GL_WGRAD_THICK_NBC 2
#endif
// (16 input channels per workgroup - 158 registers, three workgroups per CU instead of two - measured slower in round 4:
// 64 -> 64 @256^2 1.227 -> 1.345 ms, 128 .. 512 channels 1.21 -> 1.29)
template <int KS, int XM> using WgThickA = WgCfg<KS, (KS == 3 && XM == XVEC) ? GL_WGRAD_THICK_NBC : 2, 4, 1, 3, 3, 0, XM>;   // 8x8
template <int KS> using WgThickD = WgCfg<KS, 2, 4, 1, 2, 2, 2, XSCALAR>;                 // 4x4 x4 images
template <int KS> using WgThickE = WgCfg<KS, 2, 4, 1, 0, 0, 6, XSCALAR>;                 // 1x1 x64 samples

// Calls F::template run<Cfg>() for the configuration selected by the plan's (thin, nbc, geom, xmode).
template <int KS, class F>
int wg_select(const WgPlan& pl, F&& f) {
  const int xm = pl.xmode;
  if (pl.thin) {
    if (pl.nbc == 1) {
      switch (pl.geom) {
        case WG_A: return xm == XVEC ? f.template run<WgThinA<KS, 1, XVEC>>()
                        : (KS == 3 && xm == XVECUP) ? f.template run<WgThinA<3, 1, XVECUP>>()
                                                    : f.template run<WgThinA<KS, 1, XSCALAR>>();
        case WG_B: return xm == XVEC ? f.template run<WgThinB<KS, 1, XVEC>>()
                        : (KS == 3 && xm == XVECUP) ? f.template run<WgThinB<3, 1, XVECUP>>()
                                                    : f.template run<WgThinB<KS, 1, XSCALAR>>();
        case WG_C: return f.template run<WgThinC<KS, 1>>();
        case WG_D: return f.template run<WgThinD<KS, 1>>();
        default: return f.template run<WgThinE<KS, 1>>();
      }
    }
    switch (pl.geom) {
      case WG_A: return xm == XVEC ? f.template run<WgThinA<KS, 2, XVEC>>()
                      : (KS == 3 && xm == XVECUP) ? f.template run<WgThinA<3, 2, XVECUP>>()
                                                  : f.template run<WgThinA<KS, 2, XSCALAR>>();
      case WG_B: return xm == XVEC ? f.template run<WgThinB<KS, 2, XVEC>>()
                      : (KS == 3 && xm == XVECUP) ? f.template run<WgThinB<3, 2, XVECUP>>()
                                                  : f.template run<WgThinB<KS, 2, XSCALAR>>();
      case WG_C: return f.template run<WgThinC<KS, 2>>();
      case WG_D: return f.template run<WgThinD<KS, 2>>();
      default: return f.template run<WgThinE<KS, 2>>();
    }
  }
  switch (pl.geom) {
    case WG_A: return xm == XVEC ? f.template run<WgThickA<KS, XVEC>>()
                    : (KS == 3 && xm == XVECUP) ? f.template run<WgThickA<3, XVECUP>>()
                                                : f.template run<WgThickA<KS, XSCALAR>>();
    case WG_D: return f.template run<WgThickD<KS>>();
    default: return f.template run<WgThickE<KS>>();
  }
}

struct PlanFn {
  WgPlan* pl; int N, Cin, Cout, Ho, Wo;
  template <class Cfg> int run() { fill_plan<Cfg>(*pl, N, Cin, Cout, Ho, Wo); return 0; }
};

struct LaunchFn {
  WgradArgs a; const WgPlan* pl; hipStream_t st;
  template <class Cfg> int run() {
    a.tiles_x = pl->tiles_x; a.tiles_y = pl->tiles_y; a.tiles_n = pl->tiles_n;
    a.tiles_co = pl->tiles_co; a.tiles_ci = pl->tiles_ci; a.S = pl->S;
    // XCD-aware block order (default; GANLAB_WGRAD_XCD=0 restores split-fastest): 64->64 @256^2 +4.5 %, 128->128 @128^2 +2 %,
    // the others +-0.5 % (tools/wgrad_thick_probe.py), HBM reads 5.9x -> see profiles/r02g_thick_wgrad_pmc.txt
    { static const int v = [] { const char* e = getenv("GANLAB_WGRAD_XCD"); return (e && e[0] == '0') ? 0 : 1; }(); a.xcd = v; }
    const long long grid = (long long)pl->tiles_co * pl->tiles_ci * pl->S;
    GL_LAUNCH(conv_wgrad_kernel<Cfg>, dim3((unsigned)grid), dim3(256), 0, st, a);
    return GL_CHECK_LAUNCH();
  }
};

// The plan must not depend on pointer alignment for the WORKSPACE SIZE (queried without pointers):
// slots only depend on (thin, geom) through TW/TH/NI/WK, which are the same for every xmode.
WgPlan plan_wgrad(const PatchArgs& in, const float* gy, int ks, int Cout, int Ho, int Wo) {
  WgPlan pl{};
  pl.thin = Cout <= 32;
  pl.nbc = (pl.thin && in.Cin <= 16) ? 1 : 2;
  pl.geom = wg_geom(Ho, Wo, pl.thin);
  int xm = XSCALAR;
  const int tw = pl.thin ? (pl.geom == WG_A ? 32 : (pl.geom == WG_B ? 16 : 0)) : (pl.geom == WG_A ? 8 : 0);
  if (tw >= 8 && in.x && gy && !(in.Wi & 3) && !(Wo & 3) && aligned16(in.x) && aligned16(gy) &&
      in.pad == (ks - 1) / 2)
    xm = in.up ? (ks == 3 ? XVECUP : XSCALAR) : XVEC;
  pl.xmode = xm;
  PlanFn f{&pl, in.N, in.Cin, Cout, Ho, Wo};
  if (ks == 1) wg_select<1>(pl, f); else wg_select<3>(pl, f);
  return pl;
}

bool geom_ok(const ganlab_conv_geom* g) {
  return g && g->N > 0 && g->Cin > 0 && g->Hin > 0 && g->Win > 0 && g->Cout > 0 &&
         (g->ks == 1 || g->ks == 3) && g->pad >= 0 && g->pad < g->ks && (g->up == 0 || g->up == 1) &&
         g->pool == 0;
}

}  // namespace

#ifdef GL_PHASES
extern "C" int ganlab_dbg_set_phase_buf(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(gl_phase_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif


extern "C" {

static int gl_symbol_of(const void* fn, char* name, int cap) {
  if (fn == nullptr) return 0;
  const char* mangled = hipKernelNameRefByPtr(fn, nullptr);
  if (mangled == nullptr) return 0;
  int status = 0;
  char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
  const char* s = (status == 0 && dem) ? dem : mangled;
  const int n = (int)strlen(s);
  if (name && cap > 0) {
    strncpy(name, s, (size_t)cap - 1);
    name[cap - 1] = 0;
  }
  free(dem);
  return n;
}

int ganlab_last_launch(char* name, int cap, unsigned* grid) {
  if (grid) *grid = gl_last_grid;
  return gl_symbol_of(gl_last_kernel_fn, name, cap);
}

unsigned long long ganlab_launch_count(void) { return gl_launch_count; }

int ganlab_launch_history(int back, char* name, int cap, unsigned* grid) {
  if (back < 0 || back >= GL_LAUNCH_RING || (unsigned long long)back >= gl_launch_count) return 0;
  const unsigned long long i = (gl_launch_count - 1 - (unsigned long long)back) % GL_LAUNCH_RING;
  if (grid) *grid = gl_ring_grid[i];
  return gl_symbol_of(gl_ring_fn[i], name, cap);
}

int ganlab_conv_geom_size(void) { return (int)sizeof(ganlab_conv_geom); }

int ganlab_conv_out_hw(const ganlab_conv_geom* g, int* Hout, int* Wout) {
  if (!geom_ok(g)) return GANLAB_EINVAL;
  const int hv = g->up ? 2 * g->Hin : g->Hin, wv = g->up ? 2 * g->Win : g->Win;
  const int ho = hv + 2 * g->pad - g->ks + 1, wo = wv + 2 * g->pad - g->ks + 1;
  if (ho <= 0 || wo <= 0) return GANLAB_EINVAL;
  if (Hout) *Hout = ho;
  if (Wout) *Wout = wo;
  return GANLAB_OK;
}

long long ganlab_conv_pack_f32(const float* w, float* out, int Cout, int Cin, int ks, int mode, float scale,
                               void* stream) {
  if (Cout <= 0 || Cin <= 0 || (ks != 1 && ks != 3) || (mode != GANLAB_PACK_FWD && mode != GANLAB_PACK_DGRAD))
    return GANLAB_EINVAL;
  const int rows = mode == GANLAB_PACK_DGRAD ? Cout : Cin, cols = mode == GANLAB_PACK_DGRAD ? Cin : Cout;
  const int rows_p = round_up_c(rows, cin_pad(ks)), cols_p = round_up_c(cols, 64);
  const long long total = (long long)ks * ks * rows_p * cols_p;
  if (!out) return total;
  if (!w) return GANLAB_EINVAL;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  GL_LAUNCH(pack_kernel, dim3((unsigned)blocks), dim3(256), 0, gl_stream(stream), w, out, Cout, Cin,
            ks * ks, rows, cols, rows_p, cols_p, mode == GANLAB_PACK_DGRAD ? 1 : 0, scale);
  return GL_CHECK_LAUNCH() == GANLAB_OK ? total : GANLAB_ELAUNCH;
}

int ganlab_conv_fwd_f32(const float* x, const float* wp, const float* bias, float* y, const ganlab_conv_geom* g,
                        float bias_scale, int act, float slope, void* stream) {
  if (!geom_ok(g)) return GANLAB_EINVAL;
  return run_conv(x, wp, bias, y, g->N, g->Cin, g->Hin, g->Win, g->Cout, g->ks, g->pad, g->up, bias_scale, act,
                  slope, gl_stream(stream));
}

int ganlab_conv_dgrad_f32(const float* gy, const float* wp, float* gx_virtual, const ganlab_conv_geom* g,
                          void* stream) {
  int ho, wo;
  if (ganlab_conv_out_hw(g, &ho, &wo) != GANLAB_OK) return GANLAB_EINVAL;
  // the transposed conv is a plain conv of gy (N, Cout, Ho, Wo) with flipped/transposed weights and
  // padding ks-1-pad, producing the gradient w.r.t. the virtual (possibly upsampled) input
  return run_conv(gy, wp, nullptr, gx_virtual, g->N, g->Cout, ho, wo, g->Cin, g->ks, g->ks - 1 - g->pad, 0, 0.f,
                  GANLAB_ACT_NONE, 0.f, gl_stream(stream));
}

/* gx = dgrad(gy, w) * lrelu'(x): x (the conv's INPUT, shaped like gx) is itself a LeakyReLU output, and its derivative
 * is applied in the dgrad epilogue instead of a separate pass in the producing layer's backward.  Plain (no upsample,
 * no pool) fp32 convs. */
/* Split-K for the small, channel-heavy layers (512-channel 4x4 / 8x8 maps, linear layers at small batch): 0 / 1 = run the
 * plain entry point; S >= 2 = ganlab_conv_{fwd,dgrad}_splitk_f32 with a workspace of S * (output elements) floats. */
int ganlab_conv_splitk_plan(const ganlab_conv_geom* g, int dgrad) {
  int ho, wo;
  if (ganlab_conv_out_hw(g, &ho, &wo) != GANLAB_OK || g->up) return 0;
  return dgrad ? splitk_plan(g->N, g->Cout, g->Cin, g->Hin, g->Win, g->ks)
               : splitk_plan(g->N, g->Cin, g->Cout, ho, wo, g->ks);
}

int ganlab_conv_fwd_splitk_f32(const float* x, const float* wp, const float* bias, float* y, const ganlab_conv_geom* g,
                               float bias_scale, int act, float slope, void* workspace, size_t workspace_bytes,
                               void* stream) {
  const int S = ganlab_conv_splitk_plan(g, 0);
  int ho, wo;
  if (S < 2 || ganlab_conv_out_hw(g, &ho, &wo) != GANLAB_OK) return GANLAB_EUNSUPPORTED;
  if (!workspace || workspace_bytes < (size_t)S * g->N * g->Cout * ho * wo * sizeof(float)) return GANLAB_EWORKSPACE;
  return run_conv(x, wp, bias, y, g->N, g->Cin, g->Hin, g->Win, g->Cout, g->ks, g->pad, 0, bias_scale, act, slope,
                  gl_stream(stream), nullptr, 1.f, nullptr, (float*)workspace, S);
}

int ganlab_conv_dgrad_splitk_f32(const float* gy, const float* wp, float* gx, const ganlab_conv_geom* g, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  const int S = ganlab_conv_splitk_plan(g, 1);
  int ho, wo;
  if (S < 2 || ganlab_conv_out_hw(g, &ho, &wo) != GANLAB_OK) return GANLAB_EUNSUPPORTED;
  if (!workspace || workspace_bytes < (size_t)S * g->N * g->Cin * g->Hin * g->Win * sizeof(float))
    return GANLAB_EWORKSPACE;
  return run_conv(gy, wp, nullptr, gx, g->N, g->Cout, ho, wo, g->Cin, g->ks, g->ks - 1 - g->pad, 0, 0.f,
                  GANLAB_ACT_NONE, 0.f, gl_stream(stream), nullptr, 1.f, nullptr, (float*)workspace, S);
}

int ganlab_conv_dgrad_mask_supported(const ganlab_conv_geom* g) {
  // 3x3 "same" convs on >= 16-wide, 4-aligned rows (the vector-staged kernels)
  return (geom_ok(g) && g->ks == 3 && g->pad == 1 && !g->up && g->Win >= 16 && (g->Win & 3) == 0) ? 1 : 0;
}

int ganlab_conv_dgrad_mask_f32(const float* gy, const float* wp, const float* x, float* gx, const ganlab_conv_geom* g,
                               float slope, void* stream) {
  int ho, wo;
  if (ganlab_conv_out_hw(g, &ho, &wo) != GANLAB_OK || !x) return GANLAB_EINVAL;
  if (!ganlab_conv_dgrad_mask_supported(g)) return GANLAB_EUNSUPPORTED;
  return run_conv(gy, wp, nullptr, gx, g->N, g->Cout, ho, wo, g->Cin, g->ks, g->ks - 1 - g->pad, 0, 0.f,
                  GANLAB_ACT_NONE, 0.f, gl_stream(stream), nullptr, slope, x);
}

/* 1 when the conv's LeakyReLU backward can be folded into its own dgrad / wgrad kernels (the HBM-streaming 1x1
 * few-channel kernels: fromRGB at >= 64x64) */
int ganlab_conv_act_bwd_fused_supported(const ganlab_conv_geom* g) {
  if (!geom_ok(g)) return 0;
  const void* al = reinterpret_cast<const void*>(uintptr_t(16));
  return (g->Cin <= 3 && pw_small_ok(g->Cin, g->Cout, g->ks, g->pad, g->up, g->Hin, g->Win, al, al)) ? 1 : 0;
}

/* gx = dgrad(gy * lrelu'(y), w): y is the conv's activated output */
int ganlab_conv_dgrad_act_f32(const float* gy, const float* y, const float* wp, float* gx, const ganlab_conv_geom* g,
                              float slope, void* stream) {
  if (!gy || !y || !wp || !gx) return GANLAB_EINVAL;
  if (!ganlab_conv_act_bwd_fused_supported(g) || !aligned16(gy) || !aligned16(y) || !aligned16(gx))
    return GANLAB_EUNSUPPORTED;
  return run_conv(gy, wp, nullptr, gx, g->N, g->Cout, g->Hin, g->Win, g->Cin, 1, 0, 0, 0.f, GANLAB_ACT_NONE, 0.f,
                  gl_stream(stream), y, slope);
}

/* out = conv(x, w) * lrelu'(y)   (the adjoint of ganlab_conv_dgrad_act_f32 w.r.t. gy: R1's second-order sweep) */
int ganlab_conv_fwd_mask_f32(const float* x, const float* wp, const float* y, float* out, const ganlab_conv_geom* g,
                             float slope, void* stream) {
  if (!x || !y || !wp || !out) return GANLAB_EINVAL;
  if (!ganlab_conv_act_bwd_fused_supported(g) || !aligned16(x) || !aligned16(y) || !aligned16(out))
    return GANLAB_EUNSUPPORTED;
  return run_conv(x, wp, nullptr, out, g->N, g->Cin, g->Hin, g->Win, g->Cout, 1, 0, 0, 0.f, GANLAB_ACT_NONE, 0.f,
                  gl_stream(stream), y, slope);
}

/* gw = scale * wgrad(gy * lrelu'(y), x),  gb = bias_scale * sum(gy * lrelu'(y)) (or NULL) in one pass;
 * workspace: ganlab_conv_wgrad_workspace(g) bytes */
int ganlab_conv_wgrad_act_f32(const float* gy, const float* y, const float* x, float* gw, float* gb,
                              const ganlab_conv_geom* g, float scale, float bias_scale, float slope, void* workspace,
                              size_t workspace_bytes, void* stream) {
  if (!gy || !y || !x || !gw) return GANLAB_EINVAL;
  if (!ganlab_conv_act_bwd_fused_supported(g) || !aligned16(gy) || !aligned16(y) || !aligned16(x))
    return GANLAB_EUNSUPPORTED;
  long long blocks = (long long)(workspace_bytes / (64 * sizeof(float)));
  if (blocks > 1024) blocks = 1024;
  if (!workspace || blocks < 64) return GANLAB_EWORKSPACE;
  const long long hw4 = (long long)g->Hin * g->Win / 4;
  hipStream_t st = gl_stream(stream);
  for (int b0 = 0; b0 < g->Cout; b0 += 16) {
    const int B = g->Cout - b0 < 16 ? g->Cout - b0 : 16;
    GL_LAUNCH(pw_cross_sums_kernel, dim3((unsigned)blocks), dim3(256), 0, st, gy, x, (float*)workspace, g->N, B, b0,
              g->Cin, hw4, g->Cout, y, slope, gb ? 1 : 0);
    GL_LAUNCH(pw_cross_finish_kernel, dim3(1), dim3(1024), 0, st, (const float*)workspace, gw, (int)blocks, B, b0,
              g->Cin, g->Cout, g->Cin, 1, scale, gb, bias_scale);
  }
  return GL_CHECK_LAUNCH();
}

/* The same four with the LeakyReLU mask as sign BITS (ganlab_mask_bits_supported: planes of whole 32-bit words): the forward
 * writes y AND its sign bits, the three gradient kernels read the bits instead of y (fromRGB at the top of the critic:
 * progan/architectures.py:286-292; 16 channels x 1024^2 x batch: 2 GiB of mask reads per pass become 64 MiB). */
int ganlab_conv_fwd_bits_f32(const float* x, const float* wp, const float* bias, float* y, unsigned* ybits,
                             const ganlab_conv_geom* g, float bias_scale, int act, float slope, void* stream) {
  if (!x || !wp || !y || !ybits) return GANLAB_EINVAL;
  if (!ganlab_conv_act_bwd_fused_supported(g) || ((long long)g->Hin * g->Win) % 32 != 0 || !aligned16(x) || !aligned16(y))
    return GANLAB_EUNSUPPORTED;
  return run_conv(x, wp, bias, y, g->N, g->Cin, g->Hin, g->Win, g->Cout, 1, 0, 0, bias_scale, act, slope, gl_stream(stream),
                  nullptr, 1.f, nullptr, nullptr, 1, 0, ybits);
}

int ganlab_conv_dgrad_act_bits_f32(const float* gy, const unsigned* ybits, const float* wp, float* gx,
                                   const ganlab_conv_geom* g, float slope, void* stream) {
  if (!gy || !ybits || !wp || !gx) return GANLAB_EINVAL;
  if (!ganlab_conv_act_bwd_fused_supported(g) || ((long long)g->Hin * g->Win) % 32 != 0 || !aligned16(gy) || !aligned16(gx))
    return GANLAB_EUNSUPPORTED;
  return run_conv(gy, wp, nullptr, gx, g->N, g->Cout, g->Hin, g->Win, g->Cin, 1, 0, 0, 0.f, GANLAB_ACT_NONE, 0.f,
                  gl_stream(stream), reinterpret_cast<const float*>(ybits), slope, nullptr, nullptr, 1, 1);
}

int ganlab_conv_fwd_mask_bits_f32(const float* x, const float* wp, const unsigned* ybits, float* out,
                                  const ganlab_conv_geom* g, float slope, void* stream) {
  if (!x || !ybits || !wp || !out) return GANLAB_EINVAL;
  if (!ganlab_conv_act_bwd_fused_supported(g) || ((long long)g->Hin * g->Win) % 32 != 0 || !aligned16(x) || !aligned16(out))
    return GANLAB_EUNSUPPORTED;
  return run_conv(x, wp, nullptr, out, g->N, g->Cin, g->Hin, g->Win, g->Cout, 1, 0, 0, 0.f, GANLAB_ACT_NONE, 0.f,
                  gl_stream(stream), reinterpret_cast<const float*>(ybits), slope, nullptr, nullptr, 1, 1);
}

int ganlab_conv_wgrad_act_bits_f32(const float* gy, const unsigned* ybits, const float* x, float* gw, float* gb,
                                   const ganlab_conv_geom* g, float scale, float bias_scale, float slope, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  if (!gy || !ybits || !x || !gw) return GANLAB_EINVAL;
  if (!ganlab_conv_act_bwd_fused_supported(g) || ((long long)g->Hin * g->Win) % 32 != 0 || !aligned16(gy) || !aligned16(x))
    return GANLAB_EUNSUPPORTED;
  long long blocks = (long long)(workspace_bytes / (64 * sizeof(float)));
  if (blocks > 1024) blocks = 1024;
  if (!workspace || blocks < 64) return GANLAB_EWORKSPACE;
  const long long hw4 = (long long)g->Hin * g->Win / 4;
  hipStream_t st = gl_stream(stream);
  for (int b0 = 0; b0 < g->Cout; b0 += 16) {
    const int B = g->Cout - b0 < 16 ? g->Cout - b0 : 16;
    GL_LAUNCH(pw_cross_sums_kernel, dim3((unsigned)blocks), dim3(256), 0, st, gy, x, (float*)workspace, g->N, B, b0,
              g->Cin, hw4, g->Cout, reinterpret_cast<const float*>(ybits), slope, gb ? 1 : 0, 1);
    GL_LAUNCH(pw_cross_finish_kernel, dim3(1), dim3(1024), 0, st, (const float*)workspace, gw, (int)blocks, B, b0,
              g->Cin, g->Cout, g->Cin, 1, scale, gb, bias_scale);
  }
  return GL_CHECK_LAUNCH();
}

// ---- deferred InstanceNorm: affine-on-load variants (see PatchArgs, mod.hip) ------------------------------------------
static bool aff_wgrad_roll_ok(const ganlab_conv_geom* g, const void* x, const void* gy) {
  const char* roll_env = GL_ENV_ONCE("GANLAB_WGRAD_ROLL");
  return !(roll_env && roll_env[0] == '0') &&
         gl_wgrad_roll_supported(g->N, g->Cin, g->Cout, g->Hin, g->Win, g->ks, g->pad, g->up, x, gy);
}

int ganlab_conv_aff_supported(const ganlab_conv_geom* g) {
  if (!geom_ok(g) || g->ks != 3 || g->pad != 1 || g->up != 0 || (g->Win & 3) || g->Win < 32 || g->Hin < 8) return 0;
  if ((long long)g->Cin * g->Hin * g->Win * 16 >= 0x7fffffffLL) return 0;
  int bits = 0;
  if (g->Cout > 16 && g->Cin <= AFF_MAXC) bits |= 1;
  if (aff_wgrad_roll_ok(g, nullptr, nullptr) || g->Cout > 32) bits |= 2;     // rolling window, or the thick 8x8-tile kernel
  return bits;
}

int ganlab_conv_fwd_aff_f32(const float* x, const float* wp, const float* aff_s, const float* aff_t, const float* bias,
                            float* y, const ganlab_conv_geom* g, float bias_scale, int act, float slope, void* stream) {
  if (!(ganlab_conv_aff_supported(g) & 1)) return GANLAB_EUNSUPPORTED;
  if (!x || !wp || !y || !aff_s || !aff_t || !aligned16(x) || !aligned16(wp) || !aligned16(y)) return GANLAB_EINVAL;
  ConvArgs a{};
  a.in = make_patch(x, g->N, g->Cin, g->Hin, g->Win, g->pad, 0);
  a.in.aff_s = aff_s; a.in.aff_t = aff_t;
  a.wp = wp; a.bias = bias; a.y = y;
  a.Cout = g->Cout; a.Ho = g->Hin; a.Wo = g->Win;
  a.Cin_p = round_up_c(g->Cin, cin_pad(3)); a.Cout_p = round_up_c(g->Cout, 64);
  a.bias_scale = bias_scale; a.slope = slope; a.act = act;
  a.ksplit = 1;
  a.tiles_x = ceil_div(a.Wo, 32); a.tiles_y = ceil_div(a.Ho, 8); a.tiles_n = g->N;
  hipStream_t st = gl_stream(stream);
  if (g->Cout <= 32) {
    using Cfg = FwdCfg<3, 2, 5, 3, 0, XVEC>;
    a.tiles_co = ceil_div(a.Cout, Cfg::CO_T);
    const long long tiles = (long long)a.tiles_x * a.tiles_y * a.tiles_n * a.tiles_co;
    if (tiles <= 0 || tiles > 0x7fffffffLL) return GANLAB_EINVAL;
    GL_LAUNCH((conv_fwd_kernel<Cfg, false, false, true>), dim3((unsigned)tiles), dim3(256), 0, st, a);
  } else {
    using Cfg = ThickCfg;
    a.tiles_x = ceil_div(a.Wo, Cfg::G::TW);
    a.tiles_co = ceil_div(a.Cout, Cfg::CO_T);
    const long long tiles = (long long)a.tiles_x * a.tiles_y * a.tiles_n * a.tiles_co;
    if (tiles <= 0 || tiles > 0x7fffffffLL) return GANLAB_EINVAL;
    GL_LAUNCH((conv_fwd_kernel<Cfg, false, false, true>), dim3((unsigned)tiles), dim3(256), 0, st, a);
  }
  return GL_CHECK_LAUNCH();
}

// mean / rstd of every plane from the tile partials (fixed order, fp64)
__global__ void conv_tail_stats_finish_kernel(const double* __restrict__ spart, float* __restrict__ mean,
                                              float* __restrict__ rstd, long long planes, int chunks, double inv_hw, float eps) {
  const long long pl = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (pl >= planes) return;
  double s = 0.0, ss = 0.0;
  for (int k = 0; k < chunks; ++k) {
    s += spart[(pl * chunks + k) * 2];
    ss += spart[(pl * chunks + k) * 2 + 1];
  }
  const double m = s * inv_hw;
  double var = ss * inv_hw - m * m;
  if (var < 0.0) var = 0.0;
  mean[pl] = (float)m;
  rstd[pl] = (float)(1.0 / sqrt(var + (double)eps));
}

/* conv_x3.hip's TAIL form finishes its statistics with the same fixed-order kernel (declared in common.h) */
int gl_tail_stats_finish(const double* spart, float* mean, float* rstd, long long planes, int chunks, double inv_hw, float eps,
                         hipStream_t st) {
  GL_LAUNCH(conv_tail_stats_finish_kernel, dim3((unsigned)((planes + 255) / 256)), dim3(256), 0, st, spart, mean, rstd, planes,
            chunks, inv_hw, eps);
  return GL_CHECK_LAUNCH();
}

/* statistics tiles per (n, c) plane ganlab_conv_fwd_aff_tail_f32 writes (workspace: N * Cout * tiles * 2 doubles); 0 = not
 * a geometry that form takes */
int ganlab_conv_fwd_aff_tail_chunks(const ganlab_conv_geom* g) {
  if (!(ganlab_conv_aff_supported(g) & 1) || (g->Win % 32) != 0 || (g->Hin % 8) != 0) return 0;
  const char* e = GL_ENV_ONCE("GANLAB_CONV_TAIL");
  if (e != nullptr && e[0] == '0') return 0;
  return (g->Win / (g->Cout <= 32 ? 32 : ThickCfg::G::TW)) * (g->Hin / 8);
}

/* A plain 3x3 generator layer with a deferred-InstanceNorm input in ONE pass over the activations, thicker than the rolling
 * ganlab_mod_conv_fwd_f32 takes (stylegan/architectures.py:497-526): y = act(conv(x * aff_s + aff_t, w) + noise_w * noise +
 * bias * bias_scale); mean / rstd: the InstanceNorm statistics of y, from the tiles' partial sums. */
int ganlab_conv_fwd_aff_tail_f32(const float* x, const float* wp, const float* aff_s, const float* aff_t, const float* bias,
                                 const float* noise, const float* noise_w, float* y, float* mean, float* rstd,
                                 const ganlab_conv_geom* g, float bias_scale, int act, float slope, float eps, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  const int chunks = ganlab_conv_fwd_aff_tail_chunks(g);
  if (chunks <= 0) return GANLAB_EUNSUPPORTED;
  if (!x || !wp || !y || !aff_s || !aff_t || !mean || !rstd || (noise && !noise_w) || !aligned16(x) || !aligned16(wp) ||
      !aligned16(y) || (noise && !aligned16(noise)))
    return GANLAB_EINVAL;
  const long long planes = (long long)g->N * g->Cout;
  if (!workspace || workspace_bytes < (size_t)planes * chunks * 2 * sizeof(double)) return GANLAB_EWORKSPACE;
  ConvArgs a{};
  a.in = make_patch(x, g->N, g->Cin, g->Hin, g->Win, g->pad, 0);
  a.in.aff_s = aff_s; a.in.aff_t = aff_t;
  a.wp = wp; a.bias = bias; a.y = y;
  a.Cout = g->Cout; a.Ho = g->Hin; a.Wo = g->Win;
  a.Cin_p = round_up_c(g->Cin, cin_pad(3)); a.Cout_p = round_up_c(g->Cout, 64);
  a.bias_scale = bias_scale; a.slope = slope; a.act = act;
  a.ksplit = 1;
  a.noise = noise; a.noise_w = noise_w; a.spart = reinterpret_cast<double*>(workspace);
  a.tiles_x = a.Wo / 32; a.tiles_y = a.Ho / 8; a.tiles_n = g->N;
  hipStream_t st = gl_stream(stream);
  if (g->Cout <= 32) {
    using Cfg = FwdCfg<3, 2, 5, 3, 0, XVEC>;
    a.tiles_co = ceil_div(a.Cout, Cfg::CO_T);
    const long long tiles = (long long)a.tiles_x * a.tiles_y * a.tiles_n * a.tiles_co;
    if (tiles <= 0 || tiles > 0x7fffffffLL) return GANLAB_EINVAL;
    GL_LAUNCH((conv_fwd_kernel<Cfg, false, false, true, true>), dim3((unsigned)tiles), dim3(256), 0, st, a);
  } else {
    using Cfg = ThickCfg;
    a.tiles_x = a.Wo / Cfg::G::TW;
    a.tiles_co = ceil_div(a.Cout, Cfg::CO_T);
    const long long tiles = (long long)a.tiles_x * a.tiles_y * a.tiles_n * a.tiles_co;
    if (tiles <= 0 || tiles > 0x7fffffffLL) return GANLAB_EINVAL;
    GL_LAUNCH((conv_fwd_kernel<Cfg, false, false, true, true>), dim3((unsigned)tiles), dim3(256), 0, st, a);
  }
  GL_LAUNCH(conv_tail_stats_finish_kernel, dim3((unsigned)((planes + 255) / 256)), dim3(256), 0, st, (const double*)a.spart,
            mean, rstd, planes, chunks, 1.0 / ((double)g->Hin * g->Win), eps);
  return GL_CHECK_LAUNCH();
}

int ganlab_conv_wgrad_aff_f32(const float* gy, const float* x, const float* aff_s, const float* aff_t, float* gw,
                              const ganlab_conv_geom* g, float scale, void* workspace, size_t workspace_bytes,
                              void* stream) {
  if (!(ganlab_conv_aff_supported(g) & 2)) return GANLAB_EUNSUPPORTED;
  if (!gy || !x || !gw || !aff_s || !aff_t || !aligned16(x) || !aligned16(gy)) return GANLAB_EINVAL;
  if ((long long)g->Cout * g->Hin * g->Win * 4 >= 0x7fffffffLL) return GANLAB_EINVAL;
  PatchArgs in = make_patch(x, g->N, g->Cin, g->Hin, g->Win, g->pad, 0);
  in.aff_s = aff_s; in.aff_t = aff_t;
  const int ho = g->Hin, wo = g->Win;
  const WgPlan pl = plan_wgrad(in, gy, 3, g->Cout, ho, wo);
  const long long nw = (long long)g->Cout * g->Cin * 9;
  if (!workspace || workspace_bytes < (size_t)(pl.slots + 32) * nw * sizeof(float)) return GANLAB_EWORKSPACE;
  hipStream_t st = gl_stream(stream);
  int slots = pl.slots;
  if (aff_wgrad_roll_ok(g, x, gy) &&
      workspace_bytes >= (size_t)(gl_wgrad_roll_slots(g->N, g->Cin, g->Cout, g->Hin, g->Win) + 32) * nw * sizeof(float)) {
    const int rc = gl_wgrad_roll_launch(x, gy, (float*)workspace, g->N, g->Cin, g->Cout, g->Hin, g->Win, st, aff_s, aff_t);
    if (rc != GANLAB_OK) return rc;
    slots = gl_wgrad_roll_slots(g->N, g->Cin, g->Cout, g->Hin, g->Win);
  } else {
    if (pl.thin || pl.geom != WG_A || pl.xmode != XVEC) return GANLAB_EUNSUPPORTED;
    using Cfg = WgThickA<3, XVEC>;
    WgradArgs a{};
    a.in = in; a.gy = gy; a.part = (float*)workspace;
    a.Cout = g->Cout; a.Ho = ho; a.Wo = wo;
    a.tiles_x = pl.tiles_x; a.tiles_y = pl.tiles_y; a.tiles_n = pl.tiles_n;
    a.tiles_co = pl.tiles_co; a.tiles_ci = pl.tiles_ci; a.S = pl.S;
    { static const int v = [] { const char* e = getenv("GANLAB_WGRAD_XCD"); return (e && e[0] == '0') ? 0 : 1; }(); a.xcd = v; }
    const long long grid = (long long)pl.tiles_co * pl.tiles_ci * pl.S;
    GL_LAUNCH((conv_wgrad_kernel<Cfg, true>), dim3((unsigned)grid), dim3(256), 0, st, a);
  }
  const unsigned nblk = (unsigned)((nw + 255) / 256);
  const int groups = slots >= 64 ? 32 : 1;
  float* ws = (float*)workspace;
  if (groups == 1) {
    GL_LAUNCH(wgrad_reduce_kernel, dim3(nblk), dim3(256), 0, st, (const float*)ws, gw, nw, slots, 1, scale);
  } else {
    float* stage2 = ws + (long long)slots * nw;
    GL_LAUNCH(wgrad_reduce_kernel, dim3(nblk, groups), dim3(256), 0, st, (const float*)ws, stage2, nw, slots, groups, 1.0f);
    GL_LAUNCH(wgrad_reduce_kernel, dim3(nblk), dim3(256), 0, st, (const float*)stage2, gw, nw, groups, 1, scale);
  }
  return GL_CHECK_LAUNCH();
}

size_t ganlab_conv_wgrad_workspace(const ganlab_conv_geom* g) {
  int ho, wo;
  if (ganlab_conv_out_hw(g, &ho, &wo) != GANLAB_OK) return 0;
  const PatchArgs in = make_patch(nullptr, g->N, g->Cin, g->Hin, g->Win, g->pad, g->up);
  const WgPlan pl = plan_wgrad(in, nullptr, g->ks, g->Cout, ho, wo);
  size_t bytes = (size_t)(pl.slots + 32) * g->Cout * g->Cin * g->ks * g->ks * sizeof(float);
  const void* al = reinterpret_cast<const void*>(uintptr_t(16));
  if (pw_small_ok(g->Cin, g->Cout, g->ks, g->pad, g->up, g->Hin, g->Win, al, al) && bytes < 2048 * 64 * sizeof(float))
    bytes = 2048 * 64 * sizeof(float);      // 2048 partial-sum blocks of the fromRGB / toRGB cross-sum kernel
  return bytes;
}

int ganlab_conv_wgrad_f32(const float* gy, const float* x, float* gw, const ganlab_conv_geom* g, float scale,
                          void* workspace, size_t workspace_bytes, void* stream) {
  int ho, wo;
  if (ganlab_conv_out_hw(g, &ho, &wo) != GANLAB_OK || !gy || !x || !gw) return GANLAB_EINVAL;
  if ((long long)g->Cin * g->Hin * g->Win * 64 >= 0x7fffffffLL && g->Hin * g->Win < 64) return GANLAB_EINVAL;
  // (byte offsets inside one image are 32-bit: buffer-descriptor loads of the vector-staged kernels)
  if ((long long)g->Cin * g->Hin * g->Win * 4 >= 0x7fffffffLL || (long long)g->Cout * ho * wo * 4 >= 0x7fffffffLL) return GANLAB_EINVAL;
  const PatchArgs in = make_patch(x, g->N, g->Cin, g->Hin, g->Win, g->pad, g->up);
  const WgPlan pl = plan_wgrad(in, gy, g->ks, g->Cout, ho, wo);
  const long long nw = (long long)g->Cout * g->Cin * g->ks * g->ks;
  if (!workspace || workspace_bytes < (size_t)(pl.slots + 32) * nw * sizeof(float)) return GANLAB_EWORKSPACE;
  if (pw_small_ok(g->Cin, g->Cout, g->ks, g->pad, g->up, g->Hin, g->Win, gy, x)) {
    // fromRGB / toRGB weight gradient: cross sums between the 3-channel tensor and blocks of 16 channels of the other
    const bool big_is_out = g->Cin <= 4;                 // fromRGB: gy is the wide tensor
    const float* big = big_is_out ? gy : x;
    const float* small = big_is_out ? x : gy;
    const int Btot = big_is_out ? g->Cout : g->Cin, S = big_is_out ? g->Cin : g->Cout;
    const long long hw4 = (long long)g->Hin * g->Win / 4;
    long long blocks = (long long)(workspace_bytes / (64 * sizeof(float)));
    if (blocks > 1024) blocks = 1024;
    if (blocks >= 64) {
      hipStream_t st = gl_stream(stream);
      for (int b0 = 0; b0 < Btot; b0 += 16) {
        const int B = Btot - b0 < 16 ? Btot - b0 : 16;
        GL_LAUNCH(pw_cross_sums_kernel, dim3((unsigned)blocks), dim3(256), 0, st, big, small, (float*)workspace, g->N, B,
                  b0, S, hw4, Btot);
        GL_LAUNCH(pw_cross_finish_kernel, dim3(1), dim3(1024), 0, st, (const float*)workspace, gw, (int)blocks, B, b0, S,
                  g->Cout, g->Cin, big_is_out ? 1 : 0, scale);
      }
      return GL_CHECK_LAUNCH();
    }
  }
  LaunchFn f{};
  f.a.in = in; f.a.gy = gy; f.a.part = (float*)workspace;
  f.a.Cout = g->Cout; f.a.Ho = ho; f.a.Wo = wo;
  f.pl = &pl; f.st = gl_stream(stream);
  int slots = pl.slots;
  // thin 3x3 layers on 64-pixel-aligned planes: the rolling-window kernel (wgrad_roll.hip), one slot per workgroup
  // (GANLAB_WGRAD_ROLL=0 keeps the tile kernel: same-box A/B measurements)
  const char* roll_env = GL_ENV_ONCE("GANLAB_WGRAD_ROLL");
  const bool roll_on = !(roll_env && roll_env[0] == '0');
  bool rolled = false;
  if (roll_on && gl_wgrad_roll_supported(g->N, g->Cin, g->Cout, g->Hin, g->Win, g->ks, g->pad, g->up, x, gy)) {
    const int rs = gl_wgrad_roll_slots(g->N, g->Cin, g->Cout, g->Hin, g->Win);
    if (workspace_bytes >= (size_t)(rs + 32) * nw * sizeof(float)) {
      const int rc = gl_wgrad_roll_launch(x, gy, (float*)workspace, g->N, g->Cin, g->Cout, g->Hin, g->Win, f.st);
      if (rc != GANLAB_OK) return rc;
      slots = rs;
      rolled = true;
    }
  }
  if (!rolled) {
    const int rc = g->ks == 1 ? wg_select<1>(pl, f) : wg_select<3>(pl, f);
    if (rc != GANLAB_OK) return rc;
  }
  // two-stage when there are many slots and few outputs (thin layers: 4096 slots x 2304 weights)
  const unsigned nblk = (unsigned)((nw + 255) / 256);
  const int groups = slots >= 64 ? 32 : 1;
  float* ws = (float*)workspace;
  if (groups == 1) {
    GL_LAUNCH(wgrad_reduce_kernel, dim3(nblk), dim3(256), 0, f.st, (const float*)ws, gw, nw, slots, 1, scale);
  } else {
    float* stage2 = ws + (long long)slots * nw;
    GL_LAUNCH(wgrad_reduce_kernel, dim3(nblk, groups), dim3(256), 0, f.st, (const float*)ws, stage2, nw, slots,
              groups, 1.0f);
    GL_LAUNCH(wgrad_reduce_kernel, dim3(nblk), dim3(256), 0, f.st, (const float*)stage2, gw, nw, groups, 1, scale);
  }
  return GL_CHECK_LAUNCH();
}

}  // extern "C"
