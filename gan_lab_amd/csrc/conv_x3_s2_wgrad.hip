// Weight gradient of the thick stride-2 3x3 layers (pooled: avgpool2(conv3(x)); up: conv3(nearest-up2(x))) as split products on
// the bf16 matrix cores (arithmetic: conv_x3.hip - three bf16 planes per operand that sum to it exactly, six
// v_mfma_f32_16x16x32_bf16 per fp32 product, the five small products in their own accumulator set S, hi*hi in a chain H).
//
// The BOX form.  With box(h)[Y][X] = h[Y][X] + h[Y][X+1] + h[Y+1][X] + h[Y+1][X+1] (h zero outside its plane):
//   pooled  (custom_layers.py:202-211 + the 2x2 average behind it, autograd):
//     gw[co][ci][ky][kx] = scale/4 * sum_{n,y,x} gy[n,co,y,x] * box(X)[n,ci,2y+ky-1,2x+kx-1]
//   up      (nearest upsample in front of the convolution):
//     gw[co][ci][ky][kx] = scale   * sum_{n,y,x} X[n,ci,y,x]  * box(gy)[n,co,2y+1-ky,2x+1-kx]
// i.e. ONE kernel  out[lc][bc][ty][tx] = sum L[lc][y][x] * box(Hh)[bc][2y+ty-1][2x+tx-1]  over a low-resolution operand L and
// the box sums of the high-resolution one: nine taps at LOW resolution - 9/16 of the matrix work of the 16-tap form the
// exact-fp32 kernel (wgrad_roll.hip) walks, a quarter of the direct form's.  The box sums are taken in fp32 while the rows are
// staged (vertical pair first, then horizontal: one fixed order), then split like every other operand; their rounding
// (<= 1.5 ulp per element, independent across elements) is below the accumulation error (tests/test_gpu_x3.py).
//
// GEMM per tap: D[lc][bc] += sum_k A[lc][k] B[k][bc], k = 32 consecutive low pixels x of ONE low row y (a k-step):
//   A = L[lc][y][x];   B = box row 2y+ty-1 at column 2x+tx-1:  rows  VP[y] = Hh[2y] + Hh[2y+1] (ty = 1),  VQ[y] = Hh[2y+1] +
//   Hh[2y+2] (ty = 2),  VQ[y-1] (ty = 0);  columns  P[x] = V[2x] + V[2x+1] (tx = 1),  Q[x] = V[2x+1] + V[2x+2] (tx = 2),
//   Q[x-1] (tx = 0: the Q fragment shifted by one pixel in registers, v_alignbit with the dword before it - a halo element in
//   front of every Q row holds Q[x0-1]).
// VQ[y] serves ty = 2 of row y and ty = 0 of row y+1: a k-step runs VP[f] x L[f], VQ[f] x L[f], VQ[f] x L[f+1]; at the last
// row of a strip the third group takes the NEXT strip's VQ[-1] (= its high row 0 alone, the SP image) instead.
// Workgroup: 512 threads, 64 low x 32 box channels x 9 taps; wave (wl = 0..3, wb = 0..1) owns 16 x 16 x 9: S and H in
// registers (72), the closed chains T in the workgroup's slot of the workspace - every 32 k-steps (1024 terms) a tap is read, added and
// written back (one tap per k-step), the reads issued a k-step's matrix work ahead of their use.
// Staging: threads 0..255 build VP, 256..511 VQ (item = one channel's 8 high pixels of two rows + a halo dword), all 512 one
// L item; loads two k-steps ahead of their stores (inline asm, hand-counted waits - extra memory operations only make a
// counted wait stricter).  One barrier per k-step.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int XS_CL = 64, XS_CB = 32;
// ablation builds: 1 no chain closing, 2 no matrix work, 4 no staging (loads only), 8 no global loads, 16 no fragment reads,
// 32 no barrier in the k-step, 64 no fragment shift
#ifndef XS_EXP
#define XS_EXP 0
#endif
#ifndef XS_PV
#define XS_PV 6
#endif
constexpr int XS_P = 5;                            // units (16 B) per channel of an L row image: 4 + 1
constexpr int XS_Q = XS_PV;                        // ... of a box row array: 4 + 1 (halo in front) + 1
constexpr int XS_LPL = XS_CL * XS_P;               // one bf16 plane of an L row image
constexpr int XS_LROW = 3 * XS_LPL;                // [plane][lc 64][5]
constexpr int XS_VPL = XS_CB * XS_Q;               // one plane of one array (P or Q) of a box row image
constexpr int XS_VARR = 3 * XS_VPL;                // [plane][bc 32][5]
constexpr int XS_VROW = 2 * XS_VARR;               // [P | Q]
constexpr int XS_VP0 = 2 * XS_LROW, XS_VQ0 = XS_VP0 + 2 * XS_VROW, XS_SP = XS_VQ0 + 4 * XS_VROW;      // L[2] | VP[2] | VQ[4] | SP
constexpr int XS_LDS = XS_SP + XS_VROW;            // 8640 units = 138,240 bytes
constexpr int XS_SLOT = XS_CL * XS_CB * 9;         // floats of one partial slot: [wave 8][tap 9][lane 64][4]
constexpr int XS_LOADS = 7;                        // vector-memory loads per thread and k-step

#define XS_MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define XS_ACC3(a, o) "+v"(a[o]), "+v"(a[o + 1]), "+v"(a[o + 2])
#define XS_ACC9(a) XS_ACC3(a, 0), XS_ACC3(a, 3), XS_ACC3(a, 6)
#define XS_MFMA_DRAIN9(a) asm volatile("s_nop 15\n\ts_nop 15" : XS_ACC9(a))
#define XS_VALU_SETTLE9(a) asm volatile("s_nop 7\n\ts_nop 7" : XS_ACC9(a))
#define XS_MFMA_DRAIN3(a, o) asm volatile("s_nop 15\n\ts_nop 15" : XS_ACC3(a, o))
#define XS_VALU_SETTLE3(a, o) asm volatile("s_nop 7\n\ts_nop 7" : XS_ACC3(a, o))

__device__ __forceinline__ u32x4 xs_rsrc(const void* base, unsigned bytes) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(base);
  return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)) & 0xffffu,
               (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}
__device__ __forceinline__ void xs_ld(f32x4& d, const u32x4& rs, int voff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(d) : "v"(voff), "s"(rs) : "memory");
}
__device__ __forceinline__ void xs_ld16(f32x4& d, const u32x4& rs, int voff) {       // ... the next 16 bytes
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(d) : "v"(voff), "s"(rs) : "memory");
}
__device__ __forceinline__ void xs_ld1(float& d, const u32x4& rs, int voff) {
  asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(d) : "v"(voff), "s"(rs) : "memory");
}

struct XSArgs {
  const float* low;         // [N][CL][Hl][Wl]
  const float* high;        // [N][CB][2 Hl][2 Wl]
  float* part;              // [pair][split][XS_SLOT]
  const float* aff_s;       // AFF: L = low * aff_s[n][lc] + aff_t[n][lc]
  const float* aff_t;
  int N, CL, CB, Hl, Wl;
  int hshift;               // Hl = 1 << hshift
  int strips_x, nstrips;    // Wl / 32, N * Wl / 32
  int tiles_b, pairs, splits, sps;   // box-channel tiles, channel-tile pairs, k-splits per pair, strips per split
};

struct XSSet {              // one k-step's loads of a thread
  f32x4 a0, a1, b0, b1, lv;
  float ha, hb;
};
template <int YOUNGER>
__device__ __forceinline__ void xs_wait(XSSet& s) {
  asm volatile("s_waitcnt vmcnt(%7)" : "+v"(s.a0), "+v"(s.a1), "+v"(s.b0), "+v"(s.b1), "+v"(s.lv), "+v"(s.ha), "+v"(s.hb) : "n"(YOUNGER));
}

// split four values into planes (8 bytes each): two values per v_cvt_pk_bf16_f32 (round to nearest even, like the scalar cast)
__device__ __forceinline__ unsigned xs_cvt_pk(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void xs_split4(const f32x4& v, u32x2& h, u32x2& m, u32x2& l) {
  float r[4], q[4];
  h[0] = xs_cvt_pk(v[0], v[1]); h[1] = xs_cvt_pk(v[2], v[3]);
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = v[j] - __uint_as_float((j & 1) ? (h[j >> 1] & 0xffff0000u) : (h[j >> 1] << 16));
  m[0] = xs_cvt_pk(r[0], r[1]); m[1] = xs_cvt_pk(r[2], r[3]);
#pragma unroll
  for (int j = 0; j < 4; ++j) q[j] = r[j] - __uint_as_float((j & 1) ? (m[j >> 1] & 0xffff0000u) : (m[j >> 1] << 16));
  l[0] = xs_cvt_pk(q[0], q[1]); l[1] = xs_cvt_pk(q[2], q[3]);
}

template <int U> struct xs_ic { static constexpr int value = U; };

template <bool AFF>
__global__ __launch_bounds__(512) void conv_x3_s2_wgrad_kernel(XSArgs p) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[XS_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wl = wv & 3, wb = wv >> 2;             // low-channel block (16) of the tile's 64, box-channel block of its 32
  const int l16 = lane & 15, kg = lane >> 4;

  // logical workgroup id: the pairs of one k-split are neighbours and (XCD chunks) share an L2
  const int b = gl_xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int pair = b % p.pairs, split = b / p.pairs;
  const int bt = pair % p.tiles_b, lt = pair / p.tiles_b;
  const int l0 = lt * XS_CL, b0 = bt * XS_CB;
  const int s_first = split * p.sps;
  const int s_end = min(s_first + p.sps, p.nstrips);
  const int F = (s_end - s_first) << p.hshift;     // k-steps of this workgroup (a multiple of 4)
  const int hmask = p.Hl - 1;
  const int lplane = p.Hl * p.Wl, W2 = 2 * p.Wl;

  const unsigned lbytes = (unsigned)((long long)p.N * p.CL * lplane * 4), hbytes = (unsigned)((long long)p.N * p.CB * lplane * 16);
  const u32x4 rs_l = xs_rsrc(p.low, lbytes);
  const u32x4 rs_h = xs_rsrc(p.high, hbytes);
  float* const twave = p.part + ((long long)pair * p.splits + split) * XS_SLOT + wv * 9 * 256;   // this wave's nine T tiles
  const u32x4 rs_t = xs_rsrc(twave, 9 * 1024);                                                   // tile t of this lane: t * 1024 + lane * 16
  auto t_store = [&](const f32x4& v, int t) {
    // (a store of more than 8 bytes reads its data registers over several cycles: the wait states behind it keep the next
    // instruction - the compiler does not see this store - from overwriting them; without: elements 2, 3 of lanes 12..15 of
    // every row were the NEXT value of the register)
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 2" :: "v"(v), "v"(t * 1024 + lane * 16), "s"(rs_t) : "memory");
  };

  // ---- staging items ---------------------------------------------------------------------------------------------------------
  // waves 0..3 (vq = 0) build the VP rows (high rows 2y, 2y+1), waves 4..7 (vq = 1) the VQ rows (2y+1, 2y+2)
  const int vq = __builtin_amdgcn_readfirstlane(tid >> 8);
  const int b_ch = (tid & 255) >> 3, b_q = tid & 7;            // box channel, 8-pixel item of the 64 high pixels
  const int l_c = tid >> 3, l_q = tid & 7;                     // low channel, 4-pixel item of the 32 low pixels
  const int l_dst = (l_c * XS_P + (l_q >> 1)) * 16 + (l_q & 1) * 8;                 // + plane * XS_LPL * 16
  const int p_dst = (b_ch * XS_Q + (b_q >> 1)) * 16 + (b_q & 1) * 8;                // + plane * XS_VPL * 16; Q: + (XS_VARR + 1) * 16
  constexpr int OOB = (int)0x80000000;      // (+ a row pitch stays out of range: pitches are < 2^30)
  // per-thread byte offsets (the workgroup's scalar cursor is added by the load's soffset)
  // (a 16-pixel-wide low map is ONE strip whose upper 16 pixels do not exist: those items load zeros - half of the k-step's
  // products are wasted, still 9/16 x 2 of the exact kernel's count at eight times its rate)
  const int thrL = 4 * l_q < p.Wl ? (l_c * lplane + 4 * l_q) * 4 : OOB;
  const int thrRow = (b_ch * 2 * p.Hl + vq) * W2 * 4;          // row A of this thread's channel (row B: + W2 * 4)
  const int thrA = 8 * b_q < W2 ? thrRow + 32 * b_q : OOB, thrB = thrA + W2 * 4;
  float a_s = 1.f, a_t = 0.f;

  // load cursor: k-step g of the next issue_loads (they are issued in order g = 0, 1, 2, ...)
  int c_y = 0, c_strip = s_first;
  unsigned c_sL = 0, c_sH = 0, c_sHr = 0;         // scalar byte offsets: low row; high row pair at column 2 x0; ... at column 0
  int c_thrH = OOB;                               // this thread's halo column (row A), OOB where there is none
  auto cursor_strip = [&]() {
    const int n = c_strip / p.strips_x, x0 = (c_strip - n * p.strips_x) * 32;
    c_sL = (unsigned)__builtin_amdgcn_readfirstlane((((n * p.CL + l0) * p.Hl) * p.Wl + x0) * 4);
    c_sHr = (unsigned)__builtin_amdgcn_readfirstlane(((n * p.CB + b0) * 2 * p.Hl) * W2 * 4);
    c_sH = c_sHr + 8 * x0;
    const bool left = x0 > 0, right = x0 + 32 < p.Wl;
    c_thrH = b_q == 0 ? (left ? thrRow + (2 * x0 - 1) * 4 : OOB) : (b_q == 7 ? (right ? thrRow + (2 * x0 + 64) * 4 : OOB) : OOB);
  };
  cursor_strip();
  auto issue_loads = [&](XSSet& s) {              // the rows of the cursor's k-step (past the end: zeros), then advance it
    const bool on = c_strip < s_end;
    u32x4 ra = rs_h, rb = rs_h, rl = rs_l;
    ra[2] = on ? hbytes : 0u;
    rb[2] = (on && !(vq && c_y == hmask)) ? hbytes : 0u;      // VQ[Hl-1]: the row below the plane is zero
    rl[2] = on ? lbytes : 0u;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(s.a0) : "v"(thrA), "s"(ra), "s"(c_sH) : "memory");
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(s.a1) : "v"(thrA), "s"(ra), "s"(c_sH) : "memory");
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(s.b0) : "v"(thrB), "s"(rb), "s"(c_sH) : "memory");
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(s.b1) : "v"(thrB), "s"(rb), "s"(c_sH) : "memory");
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(s.ha) : "v"(c_thrH), "s"(ra), "s"(c_sHr) : "memory");
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(s.hb) : "v"(c_thrH + W2 * 4), "s"(rb), "s"(c_sHr) : "memory");
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(s.lv) : "v"(thrL), "s"(rl), "s"(c_sL) : "memory");
    if (++c_y == p.Hl) {
      c_y = 0;
      ++c_strip;
      cursor_strip();
    } else {
      c_sL += p.Wl * 4; c_sH += 2 * W2 * 4; c_sHr += 2 * W2 * 4;
    }
  };
  // one box row image from the vertical sums v0 (columns 0..3 of the item), v1 (4..7) and the halo column's vh
  auto emit = [&](const f32x4& v0, const f32x4& v1, float vh, int base) {
    const float nx = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v0[0]), 0x101, 0xf, 0xf, true));   // lane + 1's column 0
    const float e8 = b_q == 7 ? vh : nx;
    const f32x4 pv{v0[0] + v0[1], v0[2] + v0[3], v1[0] + v1[1], v1[2] + v1[3]};
    const f32x4 qv{v0[1] + v0[2], v0[3] + v1[0], v1[1] + v1[2], v1[3] + e8};
    u32x2 h, m, l;
    unsigned char* d = reinterpret_cast<unsigned char*>(lds + base) + p_dst;
    xs_split4(pv, h, m, l);
    *reinterpret_cast<u32x2*>(d) = h;
    *reinterpret_cast<u32x2*>(d + XS_VPL * 16) = m;
    *reinterpret_cast<u32x2*>(d + 2 * XS_VPL * 16) = l;
    xs_split4(qv, h, m, l);
    d += (XS_VARR + 1) * 16;
    *reinterpret_cast<u32x2*>(d) = h;
    *reinterpret_cast<u32x2*>(d + XS_VPL * 16) = m;
    *reinterpret_cast<u32x2*>(d + 2 * XS_VPL * 16) = l;
    if (b_q == 0) {                               // Q[x0 - 1]: the last bf16 of the halo unit in front of the row
      const float qm = vh + v0[0];
      const __bf16 hh = (__bf16)qm;
      const float r1 = qm - (float)hh;
      const __bf16 mm = (__bf16)r1;
      const __bf16 ll = (__bf16)(r1 - (float)mm);
      __bf16* dh = reinterpret_cast<__bf16*>(d - 16) + 7;
      dh[0] = hh; dh[XS_VPL * 8] = mm; dh[2 * XS_VPL * 8] = ll;
    }
  };
  // k-step g's images: L -> buffer g & 1;  VP -> buffer g & 1 (and, first row of a strip, SP = high row 0 alone);  VQ -> slot g & 3
  auto store_images = [&](XSSet& s, int g, int u) {    // u = g & 3
    const int y = g & hmask;
    f32x4 v = s.lv;
    if constexpr (AFF) {
      if (y == 0 && g < F) {                      // a new strip: this channel's affine of its image
        const int n = (s_first + (g >> p.hshift)) / p.strips_x;
        a_s = p.aff_s[(long long)n * p.CL + l0 + l_c];
        a_t = p.aff_t[(long long)n * p.CL + l0 + l_c];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (g < F && thrL != OOB) ? fmaf(v[j], a_s, a_t) : 0.f;      // (pixels that do not exist stay zero)
    }
    u32x2 h, m, l;
    xs_split4(v, h, m, l);
    unsigned char* d = reinterpret_cast<unsigned char*>(lds + (u & 1) * XS_LROW) + l_dst;
    *reinterpret_cast<u32x2*>(d) = h;
    *reinterpret_cast<u32x2*>(d + XS_LPL * 16) = m;
    *reinterpret_cast<u32x2*>(d + 2 * XS_LPL * 16) = l;
    if (!vq && y == 0) emit(s.a0, s.a1, s.ha, XS_SP);
    emit(s.a0 + s.b0, s.a1 + s.b1, s.ha + s.hb, vq ? XS_VQ0 + u * XS_VROW : XS_VP0 + (u & 1) * XS_VROW);
  };

  f32x4 accS[9], accH[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) { accS[t] = f32x4{0.f, 0.f, 0.f, 0.f}; accH[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  XS_VALU_SETTLE9(accS);
  XS_VALU_SETTLE9(accH);

  // fragment addresses (units)
  const int laneL = (wl * 16 + l16) * XS_P + kg;
  const int laneB = (wb * 16 + l16) * XS_Q + kg;
  bf16x8 aL[2][3];          // [k-step parity][plane] L fragments
  bf16x8 bP[2][3], bQ[2][3];   // [set][plane] box fragments of a row image: P, Q
  unsigned bq_[2][3];       // the dword in front of the Q fragment
  auto l_frags = [&](int buf, int set) {
    if (XS_EXP & 16) return;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) aL[set][pl] = __builtin_bit_cast(bf16x8, lds[buf * XS_LROW + pl * XS_LPL + laneL]);
  };
  auto b_frags = [&](int base, int set) {
    if (XS_EXP & 16) return;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      bP[set][pl] = __builtin_bit_cast(bf16x8, lds[base + pl * XS_VPL + laneB]);
      const int u = base + XS_VARR + pl * XS_VPL + laneB + 1;
      bQ[set][pl] = __builtin_bit_cast(bf16x8, lds[u]);
      bq_[set][pl] = reinterpret_cast<const unsigned*>(lds + u - 1)[3];
    }
  };
  auto b_shift = [&](int set) {    // in place: bQ[j] := Q[j - 1] (behind the MFMAs that read the unshifted fragment)
    if (XS_EXP & 64) return;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const u32x4 c = __builtin_bit_cast(u32x4, bQ[set][pl]);
      u32x4 dn;
      dn[0] = __builtin_amdgcn_alignbit(c[0], bq_[set][pl], 16); dn[1] = __builtin_amdgcn_alignbit(c[1], c[0], 16);
      dn[2] = __builtin_amdgcn_alignbit(c[2], c[1], 16); dn[3] = __builtin_amdgcn_alignbit(c[3], c[2], 16);
      bQ[set][pl] = __builtin_bit_cast(bf16x8, dn);
    }
  };
// (alternating the P and Q taps' accumulators - no MFMA behind the one that wrote its accumulator, none of hipcc's s_nop between
// them - was measured 1 % SLOWER here and 4 % slower in conv_x3.hip: dependent MFMAs issue back to back at full rate)
#define XS_TAP(t, A, Bf)                                                                                      \
  XS_MFMA(accS[t], A[2], Bf[0]); XS_MFMA(accS[t], A[0], Bf[2]); XS_MFMA(accS[t], A[1], Bf[1]);                \
  XS_MFMA(accS[t], A[1], Bf[0]); XS_MFMA(accS[t], A[0], Bf[1]); XS_MFMA(accH[t], A[0], Bf[0])
#define XS_GROUP(ty, A, set) do {                                                  \
    XS_TAP(3 * (ty) + 1, A, bP[set]); XS_TAP(3 * (ty) + 2, A, bQ[set]);            \
    __builtin_amdgcn_sched_barrier(0);                                             \
    b_shift(set);                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                             \
    XS_TAP(3 * (ty), A, bQ[set]);                                                  \
    __builtin_amdgcn_sched_barrier(0);                                             \
  } while (0)

  // ---- prologue: T := 0; the images of k-step 0 in LDS, its L and VP fragments in registers; the loads of k-steps 1, 2 in flight
#pragma unroll
  for (int t = 0; t < 9; ++t) t_store(f32x4{0.f, 0.f, 0.f, 0.f}, t);
  XSSet s0, s1;
  issue_loads(s0);
  xs_wait<0>(s0);
  store_images(s0, 0, 0);
  issue_loads(s0);
  issue_loads(s1);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  l_frags(0, 0);
  b_frags(XS_VP0, 0);

  // One k-step f (u = f & 3; c = f & 1: the L fragments aL[c] and VP[f] in fragment set c were read behind the barrier of the
  // k-step before, under its last 18 MFMAs):
  //   [stage f+1]  VP[f] x L  ->  VQ[f] x L   | barrier |  read L[f+1], VP[f+1];  VQ[f-1] x L
  // (the matrix work stands ONCE in the k-step, outside any branch: accumulators that two paths write get copied.)  The LDS
  // reads of a group are issued a group ahead, the first group's across the barrier, so no wave starts a k-step waiting for
  // the LDS.  XS_STAG (A/B builds): 1 = waves 4..7 stage BEHIND their first two groups instead - the two waves of a SIMD half
  // a k-step out of phase, one's splitting under the other's MFMAs: 10 % slower (0.483 against 0.435 ms, 256 -> 512 low 32^2);
  // 2 = odd waves late: 0.450.
#ifndef XS_STAG
#define XS_STAG 0
#endif
  const bool early = XS_STAG == 0 ? true : (XS_STAG == 1 ? vq == 0 : (wv & 1) == 0);      // stages in front of its matrix work
  auto body = [&](auto U, int f) {
    constexpr int u = decltype(U)::value;          // = f & 3
    constexpr int c = u & 1;
    XSSet& s = c ? s1 : s0;
    const int y = f & hmask;
    const int ph = (XS_EXP & 1) ? -1 : (f & 31) - 23;      // >= 0: tap ph closes its hi*hi chain in this k-step
    f32x4 tq;
    if (ph >= 0) xs_ld(tq, rs_t, ph * 1024 + lane * 16);
    // staging: the images of k-step f + 1 (loads requested two k-steps ago; behind them the seven of the k-step before - and, in
    // a closing k-step, tq's: a stricter wait, nothing else), then the requests of k-step f + 3
    auto stage = [&]() {
      xs_wait<XS_LOADS>(s);
      if (!(XS_EXP & 4)) store_images(s, f + 1, (u + 1) & 3);
      if (!(XS_EXP & 8)) issue_loads(s);
      __builtin_amdgcn_sched_barrier(0);
    };
    if (early) stage();
    if (!(XS_EXP & 2)) {
      b_frags(XS_VQ0 + u * XS_VROW, c ^ 1);          // VQ[f]
      __builtin_amdgcn_sched_barrier(0);
      XS_GROUP(1, aL[c], c);                         // ty = 1: VP[f]
      b_frags(y == 0 ? XS_SP : XS_VQ0 + ((u + 3) & 3) * XS_VROW, c);   // VQ[f - 1]; first row of a strip: high row 0 alone
      __builtin_amdgcn_sched_barrier(0);
      XS_GROUP(2, aL[c], c ^ 1);                     // ty = 2: VQ[f]
    }
    if (!early) stage();
    if (!(XS_EXP & 32)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the images of k-step f + 1 are visible behind it
    __builtin_amdgcn_sched_barrier(0);
    if (!(XS_EXP & 2)) {
      l_frags(c ^ 1, c ^ 1);                         // L[f + 1]
      b_frags(XS_VP0 + (c ^ 1) * XS_VROW, c ^ 1);    // VP[f + 1]
      __builtin_amdgcn_sched_barrier(0);
      XS_GROUP(0, aL[c], c);                         // ty = 0: VQ[f - 1]
    }
    if (ph >= 0) {                                 // T += H for one tap; only this k-step's seven loads are younger than tq's
      asm volatile("s_waitcnt vmcnt(%1)" : "+v"(tq) : "n"(XS_LOADS));
      auto close1 = [&](auto O) {
        constexpr int o = decltype(O)::value;
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(accH[o]));
        tq += accH[o];
        accH[o] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(accH[o]));
        t_store(tq, o);
      };
      switch (ph) {
        case 0: close1(xs_ic<0>{}); break;
        case 1: close1(xs_ic<1>{}); break;
        case 2: close1(xs_ic<2>{}); break;
        case 3: close1(xs_ic<3>{}); break;
        case 4: close1(xs_ic<4>{}); break;
        case 5: close1(xs_ic<5>{}); break;
        case 6: close1(xs_ic<6>{}); break;
        case 7: close1(xs_ic<7>{}); break;
        default: close1(xs_ic<8>{}); break;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int f0 = 0; f0 < F; f0 += 4) {
    body(xs_ic<0>{}, f0);
    body(xs_ic<1>{}, f0 + 1);
    body(xs_ic<2>{}, f0 + 2);
    body(xs_ic<3>{}, f0 + 3);
  }

  // ---- this workgroup's partial sums, in place: T + H + S ---------------------------------------------------------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  XS_MFMA_DRAIN9(accS);
  XS_MFMA_DRAIN9(accH);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    f32x4 v;
    xs_ld(v, rs_t, t * 1024 + lane * 16);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v));
    t_store(v + accH[t] + accS[t], t);
  }
#undef XS_TAP
#undef XS_GROUP
}

// gw = factor * sum over the k-splits (fixed order), scattered from the waves' register layout to [co][ci][ky][kx]
__global__ void x3sw_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, int pairs, int splits, int tiles_b,
                                   int CL, int CB, int up, float factor) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pairs * XS_SLOT) return;
  const int pair = i / XS_SLOT, e = i - pair * XS_SLOT;
  const float* src = part + (long long)pair * splits * XS_SLOT + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
  for (; k + 3 < splits; k += 4) {
    s0 += src[(long long)k * XS_SLOT]; s1 += src[(long long)(k + 1) * XS_SLOT];
    s2 += src[(long long)(k + 2) * XS_SLOT]; s3 += src[(long long)(k + 3) * XS_SLOT];
  }
  for (; k < splits; ++k) s0 += src[(long long)k * XS_SLOT];
  const float v = ((s0 + s1) + (s2 + s3)) * factor;
  const int r = e & 3, lane = (e >> 2) & 63, q = e >> 8, t = q % 9, wv = q / 9;
  const int bt = pair % tiles_b, lt = pair / tiles_b;
  const int lc = lt * XS_CL + (wv & 3) * 16 + (lane >> 4) * 4 + r, bc = bt * XS_CB + (wv >> 2) * 16 + (lane & 15);
  const int ty = t / 3, tx = t - 3 * ty;
  if (up) gw[(((long long)bc * CL + lc) * 3 + (2 - ty)) * 3 + (2 - tx)] = v;
  else gw[(((long long)lc * CB + bc) * 3 + ty) * 3 + tx] = v;
}

struct XSGeom { int CL, CB, Hl, Wl; };
bool xs_geom(const ganlab_conv_geom* g, XSGeom& q) {
  if (g == nullptr || g->ks != 3 || g->pad != 1 || g->N <= 0 || (g->up != 0) == (g->pool != 0)) return false;
  if (g->pool) {
    if ((g->Hin & 1) || (g->Win & 1)) return false;
    q = XSGeom{g->Cout, g->Cin, g->Hin / 2, g->Win / 2};
  } else {
    q = XSGeom{g->Cin, g->Cout, g->Hin, g->Win};
  }
  if (q.Hl < 4 || (q.Hl & (q.Hl - 1)) != 0 || (q.Wl % 32 != 0 && q.Wl != 16)) return false;
  if (q.CL % XS_CL != 0 || q.CB % XS_CB != 0) return false;
  const long long hb = (long long)g->N * q.CB * q.Hl * q.Wl * 16, lb = (long long)g->N * q.CL * q.Hl * q.Wl * 4;
  return hb < 0x7fffffffLL && lb < 0x7fffffffLL;
}
struct XSPlan { int pairs, splits, sps; };
XSPlan xs_plan(const ganlab_conv_geom* g, const XSGeom& q) {
  const int pairs = (q.CL / XS_CL) * (q.CB / XS_CB);
  const int nstrips = g->N * ((q.Wl + 31) / 32);
  int splits = (512 + pairs - 1) / pairs;          // ~ two rounds of workgroups on the 256 CUs
  if (splits > nstrips) splits = nstrips;
  if (splits < 1) splits = 1;
  const int sps = (nstrips + splits - 1) / splits;
  splits = (nstrips + sps - 1) / sps;
  return XSPlan{pairs, splits, sps};
}

}  // namespace

extern "C" {

int ganlab_conv_s2_wgrad_x3_supported(const ganlab_conv_geom* g) {
  XSGeom q;
  return xs_geom(g, q) ? 1 : 0;
}

size_t ganlab_conv_s2_wgrad_x3_workspace(const ganlab_conv_geom* g) {
  XSGeom q;
  if (!xs_geom(g, q)) return 0;
  const XSPlan pl = xs_plan(g, q);
  return (size_t)pl.pairs * pl.splits * XS_SLOT * sizeof(float);
}

/* ganlab_conv_s2_wgrad_f32 / ganlab_conv_s2_wgrad_aff_f32 (aff_s, aff_t non-null, up layers only: the x operand is
 * x * aff_s[n][ci] + aff_t[n][ci]) */
int ganlab_conv_s2_wgrad_x3(const float* gy, const float* x, const float* aff_s, const float* aff_t, float* gw,
                            const ganlab_conv_geom* g, float scale, void* workspace, size_t workspace_bytes, void* stream) {
  XSGeom q;
  if (!xs_geom(g, q)) return GANLAB_EUNSUPPORTED;
  if (!gy || !x || !gw || (aff_s == nullptr) != (aff_t == nullptr) || (aff_s != nullptr && !g->up)) return GANLAB_EINVAL;
  const XSPlan pl = xs_plan(g, q);
  if (!workspace || workspace_bytes < (size_t)pl.pairs * pl.splits * XS_SLOT * sizeof(float)) return GANLAB_EWORKSPACE;
  XSArgs a{};
  a.low = g->up ? x : gy; a.high = g->up ? gy : x;
  a.part = reinterpret_cast<float*>(workspace); a.aff_s = aff_s; a.aff_t = aff_t;
  a.N = g->N; a.CL = q.CL; a.CB = q.CB; a.Hl = q.Hl; a.Wl = q.Wl;
  a.hshift = 0;
  while ((1 << a.hshift) < a.Hl) ++a.hshift;
  a.strips_x = (a.Wl + 31) / 32; a.nstrips = a.N * a.strips_x;
  a.tiles_b = a.CB / XS_CB; a.pairs = pl.pairs; a.splits = pl.splits; a.sps = pl.sps;
  const long long grid = (long long)pl.pairs * pl.splits;
  if (grid <= 0 || grid > 0x7fffffffLL) return GANLAB_EINVAL;
  hipStream_t st = gl_stream(stream);
  if (aff_s != nullptr) GL_LAUNCH(conv_x3_s2_wgrad_kernel<true>, dim3((unsigned)grid), dim3(512), 0, st, a);
  else GL_LAUNCH(conv_x3_s2_wgrad_kernel<false>, dim3((unsigned)grid), dim3(512), 0, st, a);
  const int n = pl.pairs * XS_SLOT;
  GL_LAUNCH(x3sw_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)a.part, gw, pl.pairs, pl.splits,
            a.tiles_b, a.CL, a.CB, g->up ? 1 : 0, g->up ? scale : scale * 0.25f);
  return GL_CHECK_LAUNCH();
}

}  // extern "C"
