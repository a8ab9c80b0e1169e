// What an LDS instruction costs a wavefront whose vector ALU stream is FP64-bound (gfx950 / MI355X).
//
// The blind-rotate kernels alternate bursts of v_fma_f64 with the register <-> LDS exchanges of the FFT. A wavefront
// issues in order, so every ds_* instruction takes an issue slot of ITS wave; with two waves per SIMD the partner may
// issue vector instructions meanwhile. This program measures, for one and for two waves per SIMD and every CU busy,
// the cycles of one loop iteration = 96 v_fma_f64 (8 independent chains) + a pattern of LDS instructions, i.e. the
// MARGINAL cost of each pattern over the bare FMA loop. It decides between kernel forms.
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/lds_issue_bench.hip -o tools/lds_issue_bench ; run: tools/lds_issue_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int ITERS = 1500;

// 8 independent FP64 chains: operands %0..%7 read-write, %8 %9 the multiplier / addend, %10 the LDS byte address of the lane
#define F8 "v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t" \
           "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9\n\t"
#define F16 F8 F8
#define F24 F8 F8 F8
#define F48 F24 F24
#define F96 F48 F48
// stores take their data from the chains, loads land in v[100:131] (clobbered); 512 B (b64) / 1 KB (b128) per wave-instruction
#define W64(k, r) "ds_write_b64 %10, " #r " offset:" #k "\n\t"
#define R64(k, d) "ds_read_b64 v[" #d "], %10 offset:" #k "\n\t"
#define W128(k, d) "ds_write_b128 %11, v[" #d "] offset:" #k "\n\t"
#define R128(k, d) "ds_read_b128 v[" #d "], %11 offset:" #k "\n\t"
#define W32(k, d) "ds_write_b32 %10, v" #d " offset:" #k "\n\t"
#define WX8 W64(0, %0) W64(512, %1) W64(1024, %2) W64(1536, %3) W64(2048, %4) W64(2560, %5) W64(3072, %6) W64(3584, %7)
#define RX8 R64(0, 100:101) R64(512, 102:103) R64(1024, 104:105) R64(1536, 106:107) R64(2048, 108:109) R64(2560, 110:111) R64(3072, 112:113) R64(3584, 114:115)
#define WQ4 W128(0, 100:103) W128(1024, 104:107) W128(2048, 108:111) W128(3072, 112:115)
#define RQ4 R128(0, 116:119) R128(1024, 120:123) R128(2048, 124:127) R128(3072, 128:131)
#define RQ8 RQ4 R128(4096, 116:119) R128(5120, 120:123) R128(6144, 124:127) R128(7168, 128:131)
#define WAIT0 "s_waitcnt lgkmcnt(0)\n\t"

#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", \
             "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "memory"

#define BODY(STR) asm volatile(STR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                               : "v"(b), "v"(c), "v"(addr8), "v"(addr16) : CLOB)

template <int P>
__global__ __launch_bounds__(512) void k_pattern(double* out, double b, double c, unsigned long long* clk) {
  extern __shared__ char lds[];
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + wave * 8192u;
  const unsigned addr8 = base + lane * 8u, addr16 = base + lane * 16u;
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
    if constexpr (P == 0) BODY(F96);                                                   // FMA only
    if constexpr (P == 1) BODY(WX8 F96);                                               // + 8 ds_write_b64, one burst
    if constexpr (P == 2) BODY(RX8 F96 WAIT0);                                         // + 8 ds_read_b64, one burst
    if constexpr (P == 3) BODY(WX8 F48 RX8 F48 WAIT0);                                 // one planar exchange phase pair (store burst / load burst)
    if constexpr (P == 4) BODY(WQ4 F96);                                               // + 4 ds_write_b128 (same bytes as P1)
    if constexpr (P == 5) BODY(RQ4 F96 WAIT0);                                         // + 4 ds_read_b128
    if constexpr (P == 6) BODY(WQ4 F48 RQ4 F48 WAIT0);                                 // the exchange as 16-byte accesses
    if constexpr (P == 7) BODY(WX8 F24 RX8 F24 WX8 F24 RX8 F24 WAIT0);                 // a whole planar exchange of 8 complex values (re, im planes)
    if constexpr (P == 8) BODY(RQ8 F96 WAIT0);                                         // + 8 ds_read_b128 (a key half-row)
    if constexpr (P == 9) BODY(RX8 WAIT0 F96);                                         // loads waited for at once: exposed LDS round trip
    if constexpr (P == 10) BODY(WX8 RX8 WAIT0 F96);                                    // store burst, load burst, full drain, then the FMAs
    if constexpr (P == 11) BODY(W64(0, %0) F8 W64(512, %1) F8 W64(1024, %2) F8 W64(1536, %3) F8 W64(2048, %4) F8 W64(2560, %5) F8 W64(3072, %6) F8 W64(3584, %7) F8 F24 F8);  // stores spread one per 8 FMAs
    if constexpr (P == 12) BODY(WX8 WX8 F96);                                          // + 16 ds_write_b64
    if constexpr (P == 13) BODY(RX8 RX8 F96 WAIT0);                                    // + 16 ds_read_b64
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (lane == 0) clk[(size_t)blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

struct Row { const char* name; int lds_ops; };
static const Row kRows[] = {
    {"96 fma", 0}, {"96 fma + 8 ds_write_b64 (burst)", 8}, {"96 fma + 8 ds_read_b64 (burst)", 8}, {"8 w64 | 48 fma | 8 r64 | 48 fma", 16},
    {"96 fma + 4 ds_write_b128", 4}, {"96 fma + 4 ds_read_b128", 4}, {"4 w128 | 48 fma | 4 r128 | 48 fma", 8},
    {"planar exchange: (8 w64 | 24 fma | 8 r64 | 24 fma) x 2", 32}, {"96 fma + 8 ds_read_b128", 8}, {"8 r64, wait, 96 fma", 8},
    {"8 w64, 8 r64, wait, 96 fma", 16}, {"8 x (w64 + 8 fma) + 32 fma", 8}, {"96 fma + 16 ds_write_b64", 16}, {"96 fma + 16 ds_read_b64", 16}};

template <int P>
static void run(int waves_per_simd, int num_cus, double* out, unsigned long long* clk, std::vector<double>& cyc, std::vector<double>& us) {
  const int threads = 256 * waves_per_simd;
  const size_t lds = 96 * 1024;   // one workgroup per CU
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pattern<P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_pattern<P>, dim3(num_cus), dim3(threads), lds, 0, out, 1.0000001, 1e-9, clk);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_pattern<P>, dim3(num_cus), dim3(threads), lds, 0, out, 1.0000001, 1e-9, clk);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)num_cus * threads / 64);
  CHECK(hipMemcpy(h.data(), clk, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost));
  double s = 0;
  for (auto v : h) s += (double)v;
  cyc.push_back(s / h.size() / ITERS);
  us.push_back(1e3 * ms / ITERS);
}

template <int... Ps>
static void run_all(std::integer_sequence<int, Ps...>, int w, int cus, double* out, unsigned long long* clk, std::vector<double>& cyc, std::vector<double>& us) {
  (run<Ps>(w, cus, out, clk, cyc, us), ...);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  double* out; unsigned long long* clk;
  CHECK(hipMalloc(&out, (size_t)cus * 512 * sizeof(double)));
  CHECK(hipMalloc(&clk, (size_t)cus * 8 * sizeof(unsigned long long)));
  constexpr int NP = sizeof(kRows) / sizeof(kRows[0]);
  for (int w = 1; w <= 2; ++w) {
    std::vector<double> cyc, us;
    run_all(std::make_integer_sequence<int, NP>{}, w, cus, out, clk, cyc, us);
    for (int p = 0; p < NP; ++p) {
      // s_memtime ticks at a constant 100 MHz-derived rate on this chip? no: shader cycles (MI355X_MICROARCH.md); wall time beside it
      printf("{\"waves_per_simd\": %d, \"pattern\": \"%s\", \"lds_ops\": %d, \"cycles_per_iter\": %.1f, \"extra_cycles_over_fma\": %.1f, "
             "\"extra_per_lds_op\": %.2f, \"wall_ns_per_iter\": %.1f}\n",
             w, kRows[p].name, kRows[p].lds_ops, cyc[p], cyc[p] - cyc[0], kRows[p].lds_ops ? (cyc[p] - cyc[0]) / kRows[p].lds_ops : 0.0, 1e3 * us[p]);
    }
  }
  return 0;
}
