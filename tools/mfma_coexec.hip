// Does the FP64 matrix pipe of gfx950 (MI355X) run BESIDE the FP64 vector ALU, or on it?
//
// The blind rotation is bound by FP64 vector issue. Stages 0-2 of every transform
// use wave-uniform twiddles, i.e. they are one dense 16x16 real matrix applied to every lane's
// column -- a shape v_mfma_f64_16x16x4_f64 could take over IF its execution overlaps vector FP64
// work. This tool times, per SIMD, loops of
//   V  : 16*K independent v_fma_f64                      (vector only)
//   M  : 1 v_mfma_f64_16x16x4_f64 per iteration          (matrix only; 2 accumulators alternate)
//   MV : 1 v_mfma_f64_16x16x4_f64 + K v_fma_f64          (K = 0, 4, 8, 12, 16, 24, 32)
// at 1, 2 and 4 waves per SIMD. Co-execution shows as t(MV) ~ max(t(M), t(V)); a shared pipe as
// t(MV) ~ t(M) + t(V).
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_coexec.hip -o tools/mfma_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef double dvec4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 4000;

template <int K, bool MFMA>
__global__ void k_mix(double* out, double b, double c, unsigned long long* clk) {
  double v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (double)threadIdx.x + i;
  dvec4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  const double ma = (double)(threadIdx.x & 3) * 0.25, mb = (double)(threadIdx.x & 15) * 0.125;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; it += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (MFMA) {
        if (h == 0) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc0, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc1, 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < K; ++k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[k & 7]) : "v"(b), "v"(c));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  s += acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int K, bool MFMA>
static void run(const char* name, int waves_per_simd, int cus, double* out, unsigned long long* clk) {
  const int threads = 64 * 4 * waves_per_simd;   // one workgroup per CU, waves spread over the 4 SIMDs
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_mix<K, MFMA>), dim3(cus), dim3(threads), 0, 0, out, 1.0000001, 1e-9, clk);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_mix<K, MFMA>), dim3(cus), dim3(threads), 0, 0, out, 1.0000001, 1e-9, clk);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[2];
  CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
  // s_memrealtime ticks at 100 MHz; s_memtime at the shader clock
  const double ghz = (double)h[0] / ((double)h[1] * 10.0);
  const double cyc_per_iter = (double)h[0] / ITERS;
  printf("{\"loop\": \"%s\", \"mfma\": %d, \"fma_per_iter\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"in_kernel_GHz\": %.3f, "
         "\"wave_cycles_per_iter\": %.1f, \"simd_cycles_per_iter_per_wave\": %.1f}\n",
         name, MFMA ? 1 : 0, K, waves_per_simd, ms, ghz, cyc_per_iter, cyc_per_iter / waves_per_simd);
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  double* out; unsigned long long* clk;
  CHECK(hipMalloc(&out, sizeof(double) * cus * 1024));
  CHECK(hipMalloc(&clk, sizeof(unsigned long long) * 2 * cus));
  printf("{\"device\": \"%s\", \"cus\": %d}\n", p.gcnArchName, cus);
  for (int w : {1, 2, 4}) {
    run<16, false>("V", w, cus, out, clk);
    run<32, false>("V", w, cus, out, clk);
    run<0, true>("M", w, cus, out, clk);
    run<4, true>("MV", w, cus, out, clk);
    run<8, true>("MV", w, cus, out, clk);
    run<12, true>("MV", w, cus, out, clk);
    run<16, true>("MV", w, cus, out, clk);
    run<24, true>("MV", w, cus, out, clk);
    run<32, true>("MV", w, cus, out, clk);
  }
  return 0;
}
