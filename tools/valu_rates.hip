// VALU instruction-rate microbenchmark for gfx950 (MI355X).
//
// Purpose: the blind-rotate kernel is bound by vector-ALU issue, not HBM, so the choice of
// modular arithmetic for the N=1024 negacyclic transform (FP64-FMA over a 51-bit prime,
// 32-bit Montgomery/Shoup RNS, or 64-bit Goldilocks via v_mad_u64_u32) must come from measured
// per-instruction issue cost on this chip. Output: cycles per wave64 instruction per SIMD
// (all 1024 SIMDs busy, W waves per SIMD), derived from wall time and the in-kernel clock.
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_rates.hip -o tools/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2000;
constexpr int UNROLL = 4;      // asm blocks per loop iteration
constexpr int PER_BLOCK = 8;   // instructions per asm block (8 independent chains)

// 8 independent 32-bit chains: "op d, d, s1" style bodies supplied by macro
#define BODY32(INS) \
  asm volatile( \
    INS(%0) INS(%1) INS(%2) INS(%3) INS(%4) INS(%5) INS(%6) INS(%7) \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
    : "v"(b), "v"(c))

#define KERNEL32(NAME, T, INS) \
__global__ void NAME(T* out, T b, T c, unsigned long long* clk) { \
  T a0 = (T)threadIdx.x, a1 = a0 + (T)1, a2 = a0 + (T)2, a3 = a0 + (T)3, a4 = a0 + (T)4, a5 = a0 + (T)5, a6 = a0 + (T)6, a7 = a0 + (T)7; \
  unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime(); \
  for (int it = 0; it < ITERS; ++it) { \
    BODY32(INS); BODY32(INS); BODY32(INS); BODY32(INS); \
  } \
  unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime(); \
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; } \
}

// ---- FP64
#define I_FMA64(R)  "v_fma_f64 " #R ", " #R ", %8, %9\n\t"
#define I_MUL64(R)  "v_mul_f64 " #R ", " #R ", %8\n\t"
#define I_ADD64(R)  "v_add_f64 " #R ", " #R ", %8\n\t"
#define I_RND64(R)  "v_rndne_f64 " #R ", " #R "\n\t"
KERNEL32(k_fma_f64, double, I_FMA64)
KERNEL32(k_mul_f64, double, I_MUL64)
KERNEL32(k_add_f64, double, I_ADD64)
KERNEL32(k_rndne_f64, double, I_RND64)
// ---- FP32
#define I_FMA32(R)  "v_fma_f32 " #R ", " #R ", %8, %9\n\t"
KERNEL32(k_fma_f32, float, I_FMA32)
#define I_PKFMA32(R)  "v_pk_fma_f32 " #R ", " #R ", %8, %9\n\t"
KERNEL32(k_pk_fma_f32, double, I_PKFMA32)   // register pairs
// ---- INT32
#define I_MULLO(R)  "v_mul_lo_u32 " #R ", " #R ", %8\n\t"
#define I_MULHI(R)  "v_mul_hi_u32 " #R ", " #R ", %8\n\t"
#define I_MUL24(R)  "v_mul_u32_u24 " #R ", " #R ", %8\n\t"
#define I_MULHI24(R) "v_mul_hi_u32_u24 " #R ", " #R ", %8\n\t"
#define I_MAD24(R)  "v_mad_u32_u24 " #R ", " #R ", %8, %9\n\t"
#define I_ADD32(R)  "v_add_u32 " #R ", " #R ", %8\n\t"
#define I_ADD3(R)   "v_add3_u32 " #R ", " #R ", %8, %9\n\t"
#define I_MIN32(R)  "v_min_u32 " #R ", " #R ", %8\n\t"
#define I_ALIGN(R)  "v_alignbit_b32 " #R ", " #R ", %8, 7\n\t"
#define I_DPPQ(R)   "v_mov_b32_dpp " #R ", " #R " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
#define I_DPPROW(R) "v_mov_b32_dpp " #R ", " #R " row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
#define I_ADDDPP(R) "v_add_u32_dpp " #R ", " #R ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define I_PERMLANE32(R) "v_permlane32_swap " #R ", " #R "\n\t"
KERNEL32(k_mul_lo_u32, uint32_t, I_MULLO)
KERNEL32(k_mul_hi_u32, uint32_t, I_MULHI)
KERNEL32(k_mul_u32_u24, uint32_t, I_MUL24)
KERNEL32(k_mul_hi_u32_u24, uint32_t, I_MULHI24)
KERNEL32(k_mad_u32_u24, uint32_t, I_MAD24)
KERNEL32(k_add_u32, uint32_t, I_ADD32)
KERNEL32(k_add3_u32, uint32_t, I_ADD3)
KERNEL32(k_min_u32, uint32_t, I_MIN32)
KERNEL32(k_alignbit, uint32_t, I_ALIGN)
KERNEL32(k_mov_dpp_quad, uint32_t, I_DPPQ)
KERNEL32(k_mov_dpp_rowshr, uint32_t, I_DPPROW)
KERNEL32(k_add_dpp_quad, uint32_t, I_ADDDPP)
// ---- gfx950 half-wave / row swaps: both operands are written (lane bit 5 / lane bit 4 <-> register transposition)
#define BODYSWAP(OP) \
  asm volatile( \
    OP " %0, %1\n\t" OP " %2, %3\n\t" OP " %4, %5\n\t" OP " %6, %7\n\t" OP " %1, %2\n\t" OP " %3, %4\n\t" OP " %5, %6\n\t" OP " %7, %0\n\t" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
#define KERNELSWAP(NAME, OP) \
__global__ void NAME(uint32_t* out, uint32_t b, uint32_t c, unsigned long long* clk) { \
  uint32_t a0 = threadIdx.x + b, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + c; \
  unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime(); \
  for (int it = 0; it < ITERS; ++it) { \
    BODYSWAP(OP); BODYSWAP(OP); BODYSWAP(OP); BODYSWAP(OP); \
  } \
  unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime(); \
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; } \
}
KERNELSWAP(k_permlane32_swap, "v_permlane32_swap_b32")
KERNELSWAP(k_permlane16_swap, "v_permlane16_swap_b32")
// ---- 64-bit integer
#define I_MAD64(R)  "v_mad_u64_u32 " #R ", vcc, %8, %9, " #R "\n\t"
#define I_LSHL64(R) "v_lshlrev_b64 " #R ", 5, " #R "\n\t"
#define I_CVTF64I32(R) "v_cvt_f64_i32 " #R ", %8\n\t"
__global__ void k_mad_u64_u32(uint64_t* out, uint32_t b, uint32_t c, unsigned long long* clk) {
  uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; ++it) {
#define B64 asm volatile(I_MAD64(%0) I_MAD64(%1) I_MAD64(%2) I_MAD64(%3) I_MAD64(%4) I_MAD64(%5) I_MAD64(%6) I_MAD64(%7) \
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc")
    B64; B64; B64; B64;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
KERNEL32(k_lshl_b64, uint64_t, I_LSHL64)
__global__ void k_add_u64(uint64_t* out, uint64_t b, uint64_t c, unsigned long long* clk) {
  uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      a0 += b; a1 += b; a2 += b; a3 += b; a4 += b; a5 += b; a6 += b; a7 += b;
      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// ---- composite candidates: one modular butterfly of each flavour, 4 independent butterflies/lane ----
// (a) FP64, 51-bit prime: t = w*y mod p ; (x+t, x-t)
__global__ void k_bfly_fp64(double* out, double w, double p, double pinv, unsigned long long* clk) {
  double x[4], y[4];
  for (int i = 0; i < 4; ++i) { x[i] = threadIdx.x + i; y[i] = threadIdx.x * 3 + i; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double h = w * y[i];
        double l = __builtin_fma(w, y[i], -h);
        double q = __builtin_rint(h * pinv);
        double r = __builtin_fma(-q, p, h) + l;
        double xn = x[i] + r, yn = x[i] - r;
        x[i] = xn; y[i] = yn;
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3] + y[0] + y[1] + y[2] + y[3];
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
// (a') same with the magic-constant rounding instead of v_rndne
__global__ void k_bfly_fp64_magic(double* out, double w, double p, double pinv, unsigned long long* clk) {
  double x[4], y[4];
  const double M = 6755399441055744.0;  // 1.5 * 2^52
  for (int i = 0; i < 4; ++i) { x[i] = threadIdx.x + i; y[i] = threadIdx.x * 3 + i; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double h = w * y[i];
        double l = __builtin_fma(w, y[i], -h);
        double qm = __builtin_fma(h, pinv, M);
        asm volatile("" : "+v"(qm));
        double q = qm - M;
        double r = __builtin_fma(-q, p, h) + l;
        double xn = x[i] + r, yn = x[i] - r;
        x[i] = xn; y[i] = yn;
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3] + y[0] + y[1] + y[2] + y[3];
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
// (b) 32-bit Shoup/Harvey lazy butterfly, one 31-bit prime (two of these = one 62-bit RNS butterfly)
__global__ void k_bfly_u32(uint32_t* out, uint32_t w, uint32_t wp, uint32_t p, unsigned long long* clk) {
  uint32_t x[4], y[4];
  for (int i = 0; i < 4; ++i) { x[i] = threadIdx.x + i; y[i] = threadIdx.x * 3 + i; }
  const uint32_t p2 = 2 * p;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint32_t xr = min(x[i], x[i] - p2);
        uint32_t q = __umulhi(wp, y[i]);
        uint32_t t = w * y[i] - q * p;
        x[i] = xr + t;
        y[i] = xr + (p2 - t);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3] + y[0] + y[1] + y[2] + y[3];
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
// (c) Goldilocks P = 2^64 - 2^32 + 1 generic butterfly (64x64 -> 128 -> reduce)
__device__ __forceinline__ uint64_t gl_reduce128(uint64_t lo, uint64_t hi) {
  // hi = hh*2^32 + hl ; x = lo + hl*(2^32-1) - hh  (mod P)
  uint64_t hh = hi >> 32, hl = hi & 0xffffffffull;
  uint64_t t0 = lo - hh; if (lo < hh) t0 -= 0xffffffffull;           // borrow: subtract (2^32-1) == add P
  uint64_t t1 = hl * 0xffffffffull;
  uint64_t r = t0 + t1; if (r < t1) r += 0xffffffffull;
  return r;
}
__device__ __forceinline__ uint64_t gl_mul(uint64_t a, uint64_t b) {
  unsigned __int128 z = (unsigned __int128)a * b;
  return gl_reduce128((uint64_t)z, (uint64_t)(z >> 64));
}
__global__ void k_bfly_goldilocks(uint64_t* out, uint64_t w, unsigned long long* clk) {
  const uint64_t P = 0xffffffff00000001ull;
  uint64_t x[4], y[4];
  for (int i = 0; i < 4; ++i) { x[i] = threadIdx.x + i; y[i] = threadIdx.x * 3 + i; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint64_t t = gl_mul(w, y[i]);
        uint64_t s = x[i] + t; if (s < t) s += 0xffffffffull; if (s >= P) s -= P;
        uint64_t d = x[i] - t; if (x[i] < t) d -= 0xffffffffull;
        x[i] = s; y[i] = d;
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3] + y[0] + y[1] + y[2] + y[3];
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <typename F>
static void run(const char* name, int waves_per_simd, double instr_per_thread, F launch, unsigned long long* d_clk,
                int n_cu) {
  int block = 256 * waves_per_simd;  // waves_per_simd waves on each of the 4 SIMDs (<= 1024 threads)
  int blocks_per_cu = 1;
  if (block > 1024) { blocks_per_cu = block / 1024; block = 1024; }
  int grid = n_cu * blocks_per_cu;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch(grid, block);  // warm-up
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  launch(grid, block);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> clk(2 * grid);
  CHECK(hipMemcpy(clk.data(), d_clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
  double cyc = 0, real = 0;
  for (int i = 0; i < grid; ++i) { cyc += clk[2 * i]; real += clk[2 * i + 1]; }
  cyc /= grid; real /= grid;
  double ghz = cyc / (real * 10.0);  // s_memrealtime ticks at 100 MHz
  // cycles per wave-instruction per SIMD: each SIMD hosts waves_per_simd waves, each issuing instr_per_thread
  double cyc_per_instr = cyc / (instr_per_thread * waves_per_simd);
  double total_wave_instr = (double)grid * (block / 64) * instr_per_thread;
  printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"cyc_per_wave_instr_per_simd\": %.2f, \"in_kernel_GHz\": %.3f, "
         "\"ms\": %.3f, \"G_wave_instr_per_s\": %.1f}\n",
         name, waves_per_simd, cyc_per_instr, ghz, ms, total_wave_instr / (ms * 1e6));
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int n_cu = prop.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", prop.gcnArchName, n_cu, prop.clockRate);
  void* d_out; unsigned long long* d_clk;
  CHECK(hipMalloc(&d_out, 8ull * 1024 * 8 * n_cu));
  CHECK(hipMalloc(&d_clk, sizeof(unsigned long long) * 2 * n_cu * 8));
  const double n_simple = (double)ITERS * UNROLL * PER_BLOCK;
  for (int w : {1, 2, 4}) {
#define RUN32(K, T, B, C) run(#K, w, n_simple, [&](int g, int b) { hipLaunchKernelGGL(K, dim3(g), dim3(b), 0, 0, (T*)d_out, (T)(B), (T)(C), d_clk); }, d_clk, n_cu)
    RUN32(k_fma_f64, double, 1.0000001, 1e-9);
    RUN32(k_mul_f64, double, 1.0000001, 0);
    RUN32(k_add_f64, double, 1.25, 0);
    RUN32(k_rndne_f64, double, 0, 0);
    RUN32(k_fma_f32, float, 1.0001f, 1e-6f);
    RUN32(k_pk_fma_f32, double, 1.0, 1.0);
    RUN32(k_mul_lo_u32, uint32_t, 3, 0);
    RUN32(k_mul_hi_u32, uint32_t, 0xdeadbeef, 0);
    RUN32(k_mul_u32_u24, uint32_t, 3, 0);
    RUN32(k_mul_hi_u32_u24, uint32_t, 0xbeef, 0);
    RUN32(k_mad_u32_u24, uint32_t, 3, 7);
    RUN32(k_add_u32, uint32_t, 3, 0);
    RUN32(k_add3_u32, uint32_t, 3, 5);
    RUN32(k_min_u32, uint32_t, 0x7fffffff, 0);
    RUN32(k_alignbit, uint32_t, 0x1234567, 0);
    RUN32(k_mov_dpp_quad, uint32_t, 0, 0);
    RUN32(k_mov_dpp_rowshr, uint32_t, 0, 0);
    RUN32(k_add_dpp_quad, uint32_t, 1, 0);
    RUN32(k_permlane32_swap, uint32_t, 0, 7);
    RUN32(k_permlane16_swap, uint32_t, 0, 7);
    RUN32(k_lshl_b64, uint64_t, 0, 0);
    run("k_mad_u64_u32", w, n_simple, [&](int g, int b) { hipLaunchKernelGGL(k_mad_u64_u32, dim3(g), dim3(b), 0, 0, (uint64_t*)d_out, 0x9e3779b9u, 0x7f4a7c15u, d_clk); }, d_clk, n_cu);
    run("k_add_u64(2 instr)", w, n_simple, [&](int g, int b) { hipLaunchKernelGGL(k_add_u64, dim3(g), dim3(b), 0, 0, (uint64_t*)d_out, 0x9e3779b97f4a7c15ull, 0ull, d_clk); }, d_clk, n_cu);
    // composite butterflies: count = butterflies per thread
    const double n_bfly = (double)ITERS * 2 * 4;
    run("bfly_fp64_rndne(per butterfly)", w, n_bfly, [&](int g, int b) { hipLaunchKernelGGL(k_bfly_fp64, dim3(g), dim3(b), 0, 0, (double*)d_out, 1234567890123.0, 2251799813160961.0, 1.0 / 2251799813160961.0, d_clk); }, d_clk, n_cu);
    run("bfly_fp64_magic(per butterfly)", w, n_bfly, [&](int g, int b) { hipLaunchKernelGGL(k_bfly_fp64_magic, dim3(g), dim3(b), 0, 0, (double*)d_out, 1234567890123.0, 2251799813160961.0, 1.0 / 2251799813160961.0, d_clk); }, d_clk, n_cu);
    run("bfly_u32_shoup(per butterfly, 1 prime)", w, n_bfly, [&](int g, int b) { hipLaunchKernelGGL(k_bfly_u32, dim3(g), dim3(b), 0, 0, (uint32_t*)d_out, 123456789u, 246913578u, 2147473409u, d_clk); }, d_clk, n_cu);
    run("bfly_goldilocks(per butterfly)", w, n_bfly, [&](int g, int b) { hipLaunchKernelGGL(k_bfly_goldilocks, dim3(g), dim3(b), 0, 0, (uint64_t*)d_out, 0x123456789abcdefull, d_clk); }, d_clk, n_cu);
  }
  CHECK(hipFree(d_out)); CHECK(hipFree(d_clk));
  return 0;
}
