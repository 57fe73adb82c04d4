// Feasibility probe for "fp32 products from three bf16 planes per operand" (VERDICT r04 item 1).  Standalone:
//   hipcc --offload-arch=gfx950 -O3 tools/x3_probe.hip -o tools/build/x3_probe && tools/build/x3_probe
// Part 1 (accuracy): C = A * B^T, K-term dot products, every variant against a float64 host reference.
//   mode 0  v_mfma_f32_16x16x4_f32, one chain                     (what the fp32 kernels did before the dump)
//   mode 1  fp32 MFMA, chain dumped every 144 terms into a second set (csrc/common.h GL_ACC_DUMP)
//   mode 2  3 x bf16 planes (round to nearest), 6 products, ONE accumulator, small terms first
//   mode 3  the same, hi*hi in its own accumulator
//   mode 4  9 products, one accumulator
//   mode 5  6 products, truncating split
// Part 2 (rate): 8 waves per CU, 64 x 64 register tile per wave, operand fragments re-read from LDS by ds_read_b128
//   at the rate the planned conv kernel needs (24 reads per 96 MFMAs), random operands; reports bf16 TFLOP/s, the
//   fp32-equivalent (/6) and the in-kernel clock.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float bf_hi(float x) { return (float)(__bf16)x; }
__device__ __forceinline__ float bf_tr(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

template <bool TRUNC>
__device__ __forceinline__ void split8(const float* p, bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = p[j];
    const float fh = TRUNC ? bf_tr(x) : bf_hi(x);
    const float r1 = x - fh;
    const float fm = TRUNC ? bf_tr(r1) : bf_hi(r1);
    const float r2 = r1 - fm;
    h[j] = (__bf16)fh; m[j] = (__bf16)fm; l[j] = (__bf16)r2;
  }
}

// one wave per 16 x 16 tile of C[M][N]; A[M][K], Bt[N][K]
__global__ void acc_kernel(const float* A, const float* Bt, float* C, int M, int N, int K, int mode) {
  const int lane = threadIdx.x & 63, tile = blockIdx.x;
  const int tn = N / 16, tr = tile / tn, tc = tile % tn;
  const int l16 = lane & 15, g = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
  const float* ar = A + (long long)(tr * 16 + l16) * K;
  const float* br = Bt + (long long)(tc * 16 + l16) * K;
  if (mode <= 1) {
    for (int k = 0; k < K; k += 4) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[k + g], br[k + g], acc, 0, 0, 0);
      if (mode == 1 && (k + 4) % 144 == 0) { acc2 += acc; acc = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    acc += acc2;
  } else {
    for (int k = 0; k < K; k += 32) {
      bf16x8 ah, am, al, bh, bm, bl;
      if (mode == 5) { split8<true>(ar + k + 8 * g, ah, am, al); split8<true>(br + k + 8 * g, bh, bm, bl); }
      else { split8<false>(ar + k + 8 * g, ah, am, al); split8<false>(br + k + 8 * g, bh, bm, bl); }
      if (mode == 4) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bl, acc, 0, 0, 0);
      }
      f32x4& small = mode == 3 ? acc2 : acc;
      small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, small, 0, 0, 0);
      small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, small, 0, 0, 0);
      small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, small, 0, 0, 0);
      small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, small, 0, 0, 0);
      small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, small, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
    }
    if (mode == 3) acc += acc2;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) C[(long long)(tr * 16 + 4 * g + r) * N + tc * 16 + l16] = acc[r];
}

// ---- rate ---------------------------------------------------------------------------------------------------------
// LDS: A image 3 planes x 4 kg x 256 px units (16 B), B image 3 planes x 4 kg x 128 co units; filled with random bf16.
constexpr int A_UNITS = 3 * 4 * 256, B_UNITS = 3 * 4 * 128;
template <int LDSREAD>
__global__ __launch_bounds__(512, 2) void rate_kernel(const u32x4* src, float* out, unsigned long long* clk, int steps) {
  __shared__ __attribute__((aligned(16))) u32x4 As[A_UNITS];
  __shared__ __attribute__((aligned(16))) u32x4 Bs[B_UNITS];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
  for (int i = tid; i < A_UNITS; i += 512) As[i] = src[i];
  for (int i = tid; i < B_UNITS; i += 512) Bs[i] = src[A_UNITS + i];
  __syncthreads();
  const int l16 = lane & 15, kg = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int abase = kg * 256 + wm * 64 + l16, bbase = kg * 128 + wn * 64 + l16;
  bf16x8 a[3][4], b[3][4];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      a[p][m] = __builtin_bit_cast(bf16x8, As[p * 1024 + abase + m * 16]);
      b[p][m] = __builtin_bit_cast(bf16x8, Bs[p * 512 + bbase + m * 16]);
    }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int s = 0; s < steps; ++s) {
    if (LDSREAD) {
      const int rot = (s & 3);   // keep the addresses moving so the reads cannot be hoisted
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          a[p][m] = __builtin_bit_cast(bf16x8, As[p * 1024 + ((abase + m * 16 + rot) & 1023)]);
          b[p][m] = __builtin_bit_cast(bf16x8, Bs[p * 512 + ((bbase + m * 16 + rot) & 511)]);
        }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        f32x4 c = acc[m][n];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2][m], b[0][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][m], b[2][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][m], b[1][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][m], b[0][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][m], b[1][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][m], b[0][n], c, 0, 0, 0);
        acc[m][n] = c;
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

static void split_host(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  union FU { float f; unsigned u; };
  auto rn = [](float v) { FU c; c.f = v; unsigned u = c.u; u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); };
  auto up = [](unsigned short b) { FU c; c.u = (unsigned)b << 16; return c.f; };
  h = rn(x); const float r1 = x - up(h); m = rn(r1); const float r2 = r1 - up(m); l = rn(r2);
}

int main(int argc, char** argv) {
  const int part = argc > 1 ? atoi(argv[1]) : 3;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::uniform_real_distribution<float> ud(-3.f, 3.f);
  if (part & 1) {
    const int M = 64, N = 64;
    for (int dist = 0; dist < 3; ++dist)
      for (int K : {288, 2304, 4608}) {
        std::vector<float> A((size_t)M * K), B((size_t)N * K);
        for (auto& v : A) v = dist == 0 ? nd(rng) : dist == 1 ? nd(rng) * powf(10.f, ud(rng)) : fabsf(nd(rng));
        for (auto& v : B) v = dist == 2 ? fabsf(nd(rng)) : nd(rng) / sqrtf((float)K);
        std::vector<double> ref((size_t)M * N);
        double rms = 0;
        for (int i = 0; i < M; ++i)
          for (int j = 0; j < N; ++j) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)A[(size_t)i * K + k] * (double)B[(size_t)j * K + k];
            ref[(size_t)i * N + j] = s; rms += s * s;
          }
        rms = sqrt(rms / (M * N));
        float *dA, *dB, *dC;
        CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        printf("dist %d K %4d:", dist, K);
        for (int mode = 0; mode < 6; ++mode) {
          hipLaunchKernelGGL(acc_kernel, dim3(M / 16 * N / 16), dim3(64), 0, 0, dA, dB, dC, M, N, K, mode);
          std::vector<float> C((size_t)M * N);
          CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
          double e = 0;
          for (size_t i = 0; i < C.size(); ++i) e += ((double)C[i] - ref[i]) * ((double)C[i] - ref[i]);
          printf("  m%d %.3e", mode, sqrt(e / C.size()) / rms);
        }
        printf("\n");
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
      }
  }
  if (part & 2) {
    const int units = A_UNITS + B_UNITS;
    std::vector<unsigned short> img((size_t)units * 8);
    for (size_t i = 0; i < img.size(); i += 3) {
      unsigned short h, m, l;
      split_host(nd(rng), h, m, l);
      img[i] = h; if (i + 1 < img.size()) img[i + 1] = m; if (i + 2 < img.size()) img[i + 2] = l;
    }
    // planes as the kernel reads them: plane 0 = hi-like magnitudes, 1 = mid, 2 = lo (magnitudes matter for power only)
    std::vector<unsigned short> pl((size_t)units * 8);
    for (int u = 0; u < units; ++u) {
      const bool isA = u < A_UNITS;
      const int plane = isA ? u / 1024 : (u - A_UNITS) / 512;
      for (int j = 0; j < 8; ++j) {
        unsigned short h, m, l;
        split_host(nd(rng), h, m, l);
        pl[(size_t)u * 8 + j] = plane == 0 ? h : plane == 1 ? m : l;
      }
    }
    u32x4* dsrc; float* dout; unsigned long long* dclk;
    const int grid = 512;
    CK(hipMalloc(&dsrc, pl.size() * 2)); CK(hipMalloc(&dout, (size_t)grid * 512 * 4)); CK(hipMalloc(&dclk, grid * 16));
    CK(hipMemcpy(dsrc, pl.data(), pl.size() * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 2; ++variant) {
      const int steps = 2000;
      auto launch = [&]() {
        if (variant == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(grid), dim3(512), 0, 0, dsrc, dout, dclk, steps);
        else hipLaunchKernelGGL(rate_kernel<1>, dim3(grid), dim3(512), 0, 0, dsrc, dout, dclk, steps);
      };
      for (int i = 0; i < 20; ++i) launch();
      CK(hipDeviceSynchronize());
      const int reps = 40;
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> clk(grid * 2);
      CK(hipMemcpy(clk.data(), dclk, grid * 16, hipMemcpyDeviceToHost));
      std::vector<double> ghz;
      for (int i = 0; i < grid; ++i) ghz.push_back((double)clk[2 * i] / (double)clk[2 * i + 1] * 0.1);
      std::sort(ghz.begin(), ghz.end());
      const double flop = (double)grid * 8 * steps * 96 * 16384.0 * reps;
      const double tf = flop / (ms * 1e-3) * 1e-12;
      printf("rate %s: %.3f ms per launch, %.1f bf16 TFLOP/s, fp32-equivalent %.1f TFLOP/s, in-kernel clock median %.2f GHz; "
             "cycles per MFMA per SIMD %.2f\n", variant ? "LDS reads " : "registers ", ms / reps, tf, tf / 6,
             ghz[grid / 2], (double)clk[grid] /* one block's cycles */ / (steps * 96.0 * 2));
    }
  }
  return 0;
}
