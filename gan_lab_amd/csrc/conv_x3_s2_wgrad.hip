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
// registers (72), the closed chains T in the workgroup's slot of the workspace - every 32 k-steps (1024 terms) three taps per
// k-step are read, added and written back, the reads issued a k-step's matrix work ahead of their use.
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
constexpr int XS_P = 5;                            // units (16 B) per channel of a row image: 4 (+ 1 pad / halo in front)
constexpr int XS_LPL = XS_CL * XS_P;               // one bf16 plane of an L row image
constexpr int XS_LROW = 3 * XS_LPL;                // [plane][lc 64][5]
constexpr int XS_VPL = XS_CB * XS_P;               // one plane of one array (P or Q) of a box row image
constexpr int XS_VARR = 3 * XS_VPL;                // [plane][bc 32][5]
constexpr int XS_VROW = 2 * XS_VARR;               // [P | Q]
constexpr int XS_VP0 = 2 * XS_LROW, XS_VQ0 = XS_VP0 + 2 * XS_VROW, XS_SP = XS_VQ0 + 2 * XS_VROW;
constexpr int XS_LDS = XS_SP + XS_VROW;            // 6720 units = 107,520 bytes
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

__device__ __forceinline__ void xs_split4(const f32x4& v, u32x2& h, u32x2& m, u32x2& l) {
  bf16x4 hh, mm, ll;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hh[j] = (__bf16)v[j];
    const float r1 = v[j] - (float)hh[j];
    mm[j] = (__bf16)r1;
    ll[j] = (__bf16)(r1 - (float)mm[j]);
  }
  h = __builtin_bit_cast(u32x2, hh); m = __builtin_bit_cast(u32x2, mm); l = __builtin_bit_cast(u32x2, ll);
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
  const int s_count = min(p.sps, p.nstrips - s_first);
  const int F = s_count << p.hshift;               // k-steps of this workgroup (a multiple of 4)
  const int hmask = p.Hl - 1;
  const int lplane = p.Hl * p.Wl, hplane = 4 * lplane, W2 = 2 * p.Wl;

  const u32x4 rs_l = xs_rsrc(p.low, (unsigned)((long long)p.N * p.CL * lplane * 4));
  const u32x4 rs_h = xs_rsrc(p.high, (unsigned)((long long)p.N * p.CB * hplane * 4));
  float* const twave = p.part + ((long long)pair * p.splits + split) * XS_SLOT + wv * 9 * 256;   // this wave's nine T tiles
  float* const tbase = twave + lane * 4;                                                         // + t * 256
  const u32x4 rs_t = xs_rsrc(twave, 9 * 1024);

  // ---- staging items ---------------------------------------------------------------------------------------------------------
  const int vq = __builtin_amdgcn_readfirstlane(tid >> 8);     // 0: this thread builds VP rows (high rows 2y, 2y+1), 1: VQ rows (2y+1, 2y+2)
  const int b_ch = (tid & 255) >> 3, b_q = tid & 7;            // box channel, 8-pixel item of the 64 high pixels
  const int l_c = tid >> 3, l_q = tid & 7;                     // low channel, 4-pixel item of the 32 low pixels
  const int l_dst = (l_c * XS_P + (l_q >> 1)) * 16 + (l_q & 1) * 8;                 // + plane * XS_LPL * 16
  const int p_dst = (b_ch * XS_P + (b_q >> 1)) * 16 + (b_q & 1) * 8;                // + plane * XS_VPL * 16; Q: + (XS_VARR + 1) * 16
  float a_s = 1.f, a_t = 0.f;
  int aff_n = -1;

  auto flat_pos = [&](int f, int& n, int& y, int& x0) {
    const int strip = s_first + (f >> p.hshift);
    y = f & hmask;
    n = strip / p.strips_x;
    x0 = (strip - n * p.strips_x) * 32;
  };
  constexpr int OOB = (int)0x80000000;
  auto issue_loads = [&](XSSet& s, int g) {       // the rows of k-step g's images (past the end: zeros)
    int n, y, x0;
    flat_pos(g, n, y, x0);
    const bool on = g < F;
    const int loff = on ? (((n * p.CL + l0 + l_c) * p.Hl + y) * p.Wl + x0 + 4 * l_q) * 4 : OOB;
    const int cha = ((n * p.CB + b0 + b_ch) * 2 * p.Hl + 2 * y + vq) * W2;        // element offset of row A of this channel
    int chb = cha + W2, x0b = x0;
    bool onb = on;
    if (vq && y == hmask) {                       // VQ at the strip's last row: row B is the NEXT strip's high row 0 (its VQ[-1])
      int n2, y2;
      flat_pos(g + 1, n2, y2, x0b);
      onb = g + 1 < F;
      chb = ((n2 * p.CB + b0 + b_ch) * 2 * p.Hl) * W2;
    }
    const int hc = b_q == 0 ? -1 : (b_q == 7 ? 64 : -0x40000000);                  // halo column relative to 2 x0
    const int ca = 2 * x0 + hc, cb = 2 * x0b + hc;
    const int offa = on ? (cha + 2 * x0 + 8 * b_q) * 4 : OOB;
    const int offb = onb ? (chb + 2 * x0b + 8 * b_q) * 4 : OOB;
    const int offha = (on && ca >= 0 && ca < W2) ? (cha + ca) * 4 : OOB;
    const int offhb = (onb && cb >= 0 && cb < W2) ? (chb + cb) * 4 : OOB;
    xs_ld(s.a0, rs_h, offa); xs_ld16(s.a1, rs_h, offa);
    xs_ld(s.b0, rs_h, offb); xs_ld16(s.b1, rs_h, offb);
    xs_ld1(s.ha, rs_h, offha); xs_ld1(s.hb, rs_h, offhb);
    xs_ld(s.lv, rs_l, loff);
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
  auto store_images = [&](XSSet& s, int g) {      // k-step g's images: L -> buffer g & 1, VP / VQ -> buffer g & 1, (SP)
    int n, y, x0;
    flat_pos(g, n, y, x0);
    f32x4 v = s.lv;
    if constexpr (AFF) {
      if (n != aff_n && g < F) {                  // a new image: this channel's affine (once per strip at most)
        aff_n = n;
        a_s = p.aff_s[(long long)n * p.CL + l0 + l_c];
        a_t = p.aff_t[(long long)n * p.CL + l0 + l_c];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = g < F ? fmaf(v[j], a_s, a_t) : 0.f;
    }
    u32x2 h, m, l;
    xs_split4(v, h, m, l);
    unsigned char* d = reinterpret_cast<unsigned char*>(lds + (g & 1) * XS_LROW) + l_dst;
    *reinterpret_cast<u32x2*>(d) = h;
    *reinterpret_cast<u32x2*>(d + XS_LPL * 16) = m;
    *reinterpret_cast<u32x2*>(d + 2 * XS_LPL * 16) = l;
    const int vbase = (vq ? XS_VQ0 : XS_VP0) + (g & 1) * XS_VROW;
    if (vq && y == hmask) {
      emit(s.a0, s.a1, s.ha, vbase);              // VQ[Hl-1] = the last high row alone
      emit(s.b0, s.b1, s.hb, XS_SP);              // the next strip's VQ[-1] = its high row 0 alone
    } else {
      emit(s.a0 + s.b0, s.a1 + s.b1, s.ha + s.hb, vbase);
    }
  };

  f32x4 accS[9], accH[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) { accS[t] = f32x4{0.f, 0.f, 0.f, 0.f}; accH[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  XS_VALU_SETTLE9(accS);
  XS_VALU_SETTLE9(accH);

  // fragment addresses (units)
  const int laneL = (wl * 16 + l16) * XS_P + kg;
  const int laneB = (wb * 16 + l16) * XS_P + kg;
  bf16x8 aL[2][3];          // [set][plane] L fragments of k-steps f (set f & 1) and f + 1
  bf16x8 bP[3], bQ[3], bS[3];   // [plane] box fragments of a row image: P, Q and Q shifted by one pixel
  unsigned bq_[3];          // the dword in front of the Q fragment
  auto l_frags = [&](int buf, int set) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) aL[set][pl] = __builtin_bit_cast(bf16x8, lds[buf * XS_LROW + pl * XS_LPL + laneL]);
  };
  auto b_frags = [&](int base) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      bP[pl] = __builtin_bit_cast(bf16x8, lds[base + pl * XS_VPL + laneB]);
      const int u = base + XS_VARR + pl * XS_VPL + laneB + 1;
      bQ[pl] = __builtin_bit_cast(bf16x8, lds[u]);
      bq_[pl] = reinterpret_cast<const unsigned*>(lds + u - 1)[3];
    }
  };
  auto b_shift = [&]() {    // bS[j] = Q[j - 1]
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const u32x4 c = __builtin_bit_cast(u32x4, bQ[pl]);
      u32x4 dn;
      dn[0] = __builtin_amdgcn_alignbit(c[0], bq_[pl], 16); dn[1] = __builtin_amdgcn_alignbit(c[1], c[0], 16);
      dn[2] = __builtin_amdgcn_alignbit(c[2], c[1], 16); dn[3] = __builtin_amdgcn_alignbit(c[3], c[2], 16);
      bS[pl] = __builtin_bit_cast(bf16x8, dn);
    }
  };
#define XS_TAP(t, A, Bf)                                                                                      \
  XS_MFMA(accS[t], A[2], Bf[0]); XS_MFMA(accS[t], A[0], Bf[2]); XS_MFMA(accS[t], A[1], Bf[1]);                \
  XS_MFMA(accS[t], A[1], Bf[0]); XS_MFMA(accS[t], A[0], Bf[1]); XS_MFMA(accH[t], A[0], Bf[0])
#define XS_GROUP(ty, A) do { XS_TAP(3 * (ty) + 1, A, bP); XS_TAP(3 * (ty) + 2, A, bQ); XS_TAP(3 * (ty), A, bS); } while (0)

  // ---- prologue: T := 0; images of k-step 0 and the first strip's SP (= its high row 0 alone: row A of the VP threads) in LDS;
  //      the loads of k-steps 1 and 2 in flight ---------------------------------------------------------------------------------
#pragma unroll
  for (int t = 0; t < 9; ++t) *reinterpret_cast<f32x4*>(tbase + t * 256) = f32x4{0.f, 0.f, 0.f, 0.f};
  XSSet s0, s1;
  issue_loads(s0, 0);
  xs_wait<0>(s0);
  if (!vq) emit(s0.a0, s0.a1, s0.ha, XS_SP);
  store_images(s0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  issue_loads(s0, 1);
  issue_loads(s1, 2);
  l_frags(0, 0);
  b_frags(XS_SP);
  b_shift();
  __builtin_amdgcn_sched_barrier(0);
  XS_GROUP(0, aL[0]);       // ty = 0 of the first row: VQ[-1] x L[0]
  __builtin_amdgcn_sched_barrier(0);
  b_frags(XS_VP0);          // VP[0]

  auto body = [&](auto U, int f) {
    constexpr int u = decltype(U)::value;          // = f & 1: register set, buffers
    XSSet& s = u ? s1 : s0;
    const int y = f & hmask;
    const int ph = (f & 31) - 29;                   // >= 0: taps 3 ph .. 3 ph + 2 close their hi*hi chains in this k-step
    f32x4 tq[3];
    b_shift();
    __builtin_amdgcn_sched_barrier(0);
    XS_GROUP(1, aL[u]);                             // ty = 1: VP[f] x L[f]
    __builtin_amdgcn_sched_barrier(0);
    // ---- staging: the images of k-step f + 1 (loads requested two k-steps ago; behind them only the seven of the k-step before)
    xs_wait<XS_LOADS>(s);
    if (ph >= 0) {
#pragma unroll
      for (int j = 0; j < 3; ++j) xs_ld(tq[j], rs_t, (3 * ph + j) * 1024 + lane * 16);
    }
    store_images(s, f + 1);
    __builtin_amdgcn_sched_barrier(0);
    b_frags(XS_VQ0 + u * XS_VROW);                  // VQ[f]
    b_shift();
    __builtin_amdgcn_sched_barrier(0);
    XS_GROUP(2, aL[u]);                             // ty = 2: VQ[f] x L[f]
    __builtin_amdgcn_sched_barrier(0);
    issue_loads(s, f + 3);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the images stored above are visible behind it
    __builtin_amdgcn_sched_barrier(0);
    l_frags(u ^ 1, u ^ 1);                          // L[f + 1]
    if (y == hmask) {                               // the next strip's VQ[-1] instead of this strip's VQ[Hl-1]
      b_frags(XS_SP);
      b_shift();
    }
    __builtin_amdgcn_sched_barrier(0);
    XS_GROUP(0, aL[u ^ 1]);                         // ty = 0 of the next row: VQ[f] x L[f + 1]
    __builtin_amdgcn_sched_barrier(0);
    b_frags(XS_VP0 + (u ^ 1) * XS_VROW);            // VP[f + 1]
    if (ph >= 0) {                                  // T += H for three taps; only this k-step's seven loads are younger than tq's
      asm volatile("s_waitcnt vmcnt(%3)" : "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]) : "n"(XS_LOADS));
      auto close3 = [&](auto O) {
        constexpr int o = decltype(O)::value;
        XS_MFMA_DRAIN3(accH, o);
#pragma unroll
        for (int j = 0; j < 3; ++j) { tq[j] += accH[o + j]; accH[o + j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        XS_VALU_SETTLE3(accH, o);
#pragma unroll
        for (int j = 0; j < 3; ++j) *reinterpret_cast<f32x4*>(tbase + (o + j) * 256) = tq[j];
      };
      if (ph == 0) close3(xs_ic<0>{});
      else if (ph == 1) close3(xs_ic<3>{});
      else close3(xs_ic<6>{});
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int f0 = 0; f0 < F; f0 += 2) {
    body(xs_ic<0>{}, f0);
    body(xs_ic<1>{}, f0 + 1);
  }

  // ---- this workgroup's partial sums, in place: T + H + S ---------------------------------------------------------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  XS_MFMA_DRAIN9(accS);
  XS_MFMA_DRAIN9(accH);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    f32x4* q = reinterpret_cast<f32x4*>(tbase + t * 256);
    *q = *q + accH[t] + accS[t];
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
  if (q.Hl < 4 || (q.Hl & (q.Hl - 1)) != 0 || q.Wl % 32 != 0) return false;
  if (q.CL % XS_CL != 0 || q.CB % XS_CB != 0) return false;
  const long long hb = (long long)g->N * q.CB * q.Hl * q.Wl * 16, lb = (long long)g->N * q.CL * q.Hl * q.Wl * 4;
  return hb < 0x7fffffffLL && lb < 0x7fffffffLL;
}
struct XSPlan { int pairs, splits, sps; };
XSPlan xs_plan(const ganlab_conv_geom* g, const XSGeom& q) {
  const int pairs = (q.CL / XS_CL) * (q.CB / XS_CB);
  const int nstrips = g->N * (q.Wl / 32);
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
  a.strips_x = a.Wl / 32; a.nstrips = a.N * a.strips_x;
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
