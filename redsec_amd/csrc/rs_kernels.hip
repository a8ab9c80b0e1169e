// rs_kernels.hip -- integer kernels of the gate-bootstrapping hot path (the transform-based ones
// live in rs_bootstrap.hip):
//   keyswitch_tiled_kernel / keyswitch_kernel   lweKeySwitch (N -> n)
//   lincomb / gather_rows / linear_fc / conv_ternary / sumpool   LWE word arithmetic of the layer
//                                                                linear stage and the max-pool gather
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "rs_host.h"
#include "rs_kernels.h"
#include "rs_ntt.h"

namespace rs {

// -------------------------------------------------------------------------------------------------
// Keyswitch: one workgroup per ciphertext, threads over output words. The KSK (83-104 MB) stays in
// the Infinity Cache; rows are gathered by digit. u = u0 (+ u1) (+ bconst on the b word): the sum
// form serves bootsMUX.
// -------------------------------------------------------------------------------------------------
constexpr int KS_THREADS = 256;
constexpr int KS_MAXR = 4;  // output words per thread: a workgroup covers 1024 words, blockIdx.y walks wider samples

__global__ __launch_bounds__(KS_THREADS) void keyswitch_kernel(KeyswitchArgs a) {
  const long ct = blockIdx.x;
  const int tid = threadIdx.x;
  const int N = a.N, w_base = (int)blockIdx.y * KS_THREADS * KS_MAXR;
  const int32_t* u0 = a.u0 + ct * (N + 1);
  const int32_t* u1 = a.u1 ? a.u1 + ct * (N + 1) : nullptr;
  uint32_t acc[KS_MAXR];
#pragma unroll
  for (int k = 0; k < KS_MAXR; ++k) acc[k] = 0;
  const int W = a.W, t = a.t, basebit = a.basebit;
  const uint32_t prec_offset = 1u << (32 - (1 + basebit * t));
  const uint32_t mask = (1u << basebit) - 1u;
  for (int i = 0; i < N; ++i) {
    uint32_t ai = (uint32_t)u0[i];
    if (u1) ai += (uint32_t)u1[i];
    const uint32_t aibar = ai + prec_offset;
    for (int j = 0; j < t; ++j) {
      const uint32_t dgt = (aibar >> (32 - (j + 1) * basebit)) & mask;
      if (dgt == 0) continue;
      const int32_t* row = a.ksk + ((((size_t)i * t + j) << basebit) + dgt) * (size_t)W;
#pragma unroll
      for (int k = 0; k < KS_MAXR; ++k) {
        const int w = w_base + tid + k * KS_THREADS;
        if (w < W) acc[k] += (uint32_t)row[w];
      }
    }
  }
  uint32_t bw = (uint32_t)u0[N];
  if (u1) bw += (uint32_t)u1[N];
  bw += (uint32_t)a.bconst;
  int32_t* out = a.out + ct * W;
#pragma unroll
  for (int k = 0; k < KS_MAXR; ++k) {
    const int w = w_base + tid + k * KS_THREADS;
    if (w < W) out[w] = (int32_t)((w == W - 1 ? bw : 0u) - acc[k]);
  }
}

// -------------------------------------------------------------------------------------------------
// Tiled keyswitch (the production path for the shipped parameter sets).
//
// A workgroup of 4 waves owns 256 ciphertexts x KS_CH output words; lane = ciphertext. For a group
// of KS_IG input coefficients the KSK slice [KS_IG][t][base][KS_CH] is staged in LDS once (row v=0
// is a resident zero row), then every lane extracts its own digits and subtracts the LDS row they
// select: one ds_read_b128 + four v_sub per four output words. Each KSK byte fetched from L2 thus
// serves 256 ciphertexts instead of one, and the extracted samples are read once per word chunk.
// Rows are padded by 16 B so that the <= 8 distinct rows a wave touches sit in distinct bank quads.
// -------------------------------------------------------------------------------------------------
constexpr int KS_CH = 32;            // output words per workgroup
constexpr int KS_CHP = KS_CH + 4;    // padded row (words)
constexpr int KS_TILE_THREADS = 256;

// What a tile workgroup does with a lane's KS_CH partial sums: the plain store of the throughput form, the slice's partials for
// the two-step sliced form, or integer atomics into a zeroed output (sliced, no scratch).
__device__ __forceinline__ void ks_tile_finish(const KeyswitchArgs& a, const uint32_t (&acc)[KS_CH], long ct, int w0, const int32_t* u0, const int32_t* u1) {
  const int W = a.W, N = a.N;
  uint32_t bw = 0;
  if (blockIdx.z == 0) {
    bw = (uint32_t)u0[N];
    if (u1) bw += (uint32_t)u1[N];
    bw += (uint32_t)a.bconst;
  }
  int32_t* out = a.out + ct * W + w0;
  if (gridDim.z == 1) {
#pragma unroll
    for (int k = 0; k < KS_CH; ++k) {
      const int w = w0 + k;
      if (w < W) out[k] = (int32_t)((w == W - 1 ? bw : 0u) - acc[k]);
    }
  } else if (a.scratch) {
    // sliced form: this slice's partial sums, [slice][word][ciphertext] -- consecutive lanes write consecutive words
    uint32_t* part = a.scratch + ((size_t)blockIdx.z * W + w0) * (size_t)a.B + ct;
#pragma unroll
    for (int k = 0; k < KS_CH; ++k) {
      if (w0 + k < W) part[(size_t)k * a.B] = acc[k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < KS_CH; ++k) {
      const int w = w0 + k;
      if (w < W) atomicAdd(reinterpret_cast<unsigned int*>(out + k), (w == W - 1 ? bw : 0u) - acc[k]);
    }
  }
}

// Tile order of the tiled keyswitch kernels: dispatch order (x fastest). An XCD-aware order (each XCD working through whole
// 32-word slices of the KSK) was measured in round 4 -- 16.8 instead of 27.6 GB of fabric traffic, 11.30 instead of 11.18 ms
// (profiles/r04/bm_*): the kernel is bound by its LDS reads, not by that traffic -- and removed.
template <int T, int BASEBIT, int KS_IG>  // KS_IG = input coefficients staged per round
__global__ __launch_bounds__(KS_TILE_THREADS) void keyswitch_tiled_kernel(KeyswitchArgs a) {
  constexpr int BASE = 1 << BASEBIT;
  constexpr int ROWS = KS_IG * T * BASE;
  __shared__ __attribute__((aligned(16))) int32_t s_ksk[2][ROWS * KS_CHP];
  const int tid = threadIdx.x;
  const long ct = (long)blockIdx.x * KS_TILE_THREADS + tid;
  const bool live = ct < a.B;
  const int w0 = (int)blockIdx.y * KS_CH;
  const int W = a.W, N = a.N;   // ring degree of the extracted samples: 1024 ... 8192
  const int32_t* u0 = a.u0 + (live ? ct : 0) * (size_t)(N + 1);
  const int32_t* u1 = a.u1 ? a.u1 + (live ? ct : 0) * (size_t)(N + 1) : nullptr;

  for (int e = tid; e < 2 * ROWS * KS_CHP; e += KS_TILE_THREADS) (&s_ksk[0][0])[e] = 0;

  uint32_t acc[KS_CH];
#pragma unroll
  for (int k = 0; k < KS_CH; ++k) acc[k] = 0;

  constexpr uint32_t prec_offset = 1u << (32 - (1 + BASEBIT * T));
  constexpr uint32_t mask = (1u << BASEBIT) - 1u;
  // staging: KS_IG * T * (BASE-1) row segments of KS_CH words; 8 threads x 16 B per segment
  constexpr int SEGS = KS_IG * T * (BASE - 1);
  // the next group's rows are requested into registers before the current group's lookups and stored to LDS after them: their L2
  // round trip runs behind the lookups (round 4; as in keyswitch_tiled_comb_kernel)
  constexpr int NST = (SEGS * 8 + KS_TILE_THREADS - 1) / KS_TILE_THREADS;
  int32_t st[NST][4];
  auto stage_load = [&](int i0) {
#pragma unroll
    for (int sl = 0; sl < NST; ++sl) {
      const int sidx = tid + sl * KS_TILE_THREADS;
      const int seg = sidx >> 3, part = sidx & 7;
      const int v = seg % (BASE - 1) + 1;
      const int ij = seg / (BASE - 1);          // ii * T + j
      const int wq = w0 + part * 4;
      // rows are only 4-byte aligned in general (W odd): assemble from scalar loads
      const int32_t* src = a.ksk + (((size_t)i0 * T + ij) * BASE + v) * (size_t)W + wq;
#pragma unroll
      for (int e = 0; e < 4; ++e) st[sl][e] = (sidx < SEGS * 8 && wq + e < W) ? src[e] : 0;
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int sl = 0; sl < NST; ++sl) {
      const int sidx = tid + sl * KS_TILE_THREADS;
      if (sidx < SEGS * 8) {
        const int seg = sidx >> 3, part = sidx & 7;
        const int v = seg % (BASE - 1) + 1;
        const int ij = seg / (BASE - 1);
        *reinterpret_cast<int4*>(&s_ksk[buf][(ij * BASE + v) * KS_CHP + part * 4]) = make_int4(st[sl][0], st[sl][1], st[sl][2], st[sl][3]);
      }
    }
  };

  // blockIdx.z selects a slice of the N input coefficients (latency form for small batches: the
  // slices add their partial sums into a zeroed output with integer atomics -- exact and
  // order-independent mod 2^32); gridDim.z == 1 is the plain-store throughput form.
  const int groups_per_split = (N / KS_IG) / (int)gridDim.z;
  const int g_begin = (int)blockIdx.z * groups_per_split, g_end = g_begin + groups_per_split;
  __syncthreads();
  stage_load(g_begin * KS_IG);
  stage_store(g_begin & 1);
  __syncthreads();
  for (int g = g_begin; g < g_end; ++g) {
    const int buf = g & 1;
    if (g + 1 < g_end) stage_load((g + 1) * KS_IG);
    const int i0 = g * KS_IG;
    uint32_t ai[KS_IG];
#pragma unroll
    for (int ii = 0; ii < KS_IG; ++ii) {
      uint32_t v = live ? (uint32_t)u0[i0 + ii] : 0u;
      if (u1 && live) v += (uint32_t)u1[i0 + ii];
      ai[ii] = live ? v + prec_offset : 0u;
    }
#pragma unroll 1
    for (int ii = 0; ii < KS_IG; ++ii) {
#pragma unroll
      for (int j = 0; j < T; ++j) {
        const uint32_t dgt = (ai[ii] >> (32 - (j + 1) * BASEBIT)) & mask;
        const int4* row = reinterpret_cast<const int4*>(&s_ksk[buf][((ii * T + j) * BASE + (int)dgt) * KS_CHP]);
#pragma unroll
        for (int q = 0; q < KS_CH / 4; ++q) {
          const int4 r = row[q];
          acc[4 * q + 0] += (uint32_t)r.x;
          acc[4 * q + 1] += (uint32_t)r.y;
          acc[4 * q + 2] += (uint32_t)r.z;
          acc[4 * q + 3] += (uint32_t)r.w;
        }
      }
    }
    if (g + 1 < g_end) stage_store(buf ^ 1);   // the other buffer: nobody reads it in this round
    __syncthreads();
  }
  if (!live) return;
  ks_tile_finish(a, acc, ct, w0, u0, u1);
}

// -------------------------------------------------------------------------------------------------
// Tiled keyswitch with COMBINED digits. The tiled kernel above spends one 16-byte LDS read and four adds per four output words
// and DIGIT, and both its LDS pipe and its vector pipe run at two thirds of their peaks. D adjacent digits of a coefficient
// select one of BASE^D sums of KSK rows; those sums are built once per workgroup -- 256 ciphertexts share them -- into a second
// LDS table, and every lane then does ONE read-and-add per D digits: default-128 (t = 8 digits of 2 bits) takes 4 lookups of 16
// rows instead of 8 of 4, the (18, 1) keys of the larger sets 5 lookups (4 x 16 rows + 1 x 4) instead of 18 of 2. Wrapping
// 32-bit sums are exact in any grouping, so the result is the tiled kernel's bit for bit. (The REDsec set's 9 digits of 3 bits
// would need 64-row groups: building them costs what they save; it stays on the kernel above.)
// Per group of KS_IG coefficients: base rows global -> s_base (requested before the previous group's lookups), barrier, sums
// s_base -> s_tab, barrier, lookups.
// -------------------------------------------------------------------------------------------------
template <int T, int BASEBIT, int KS_IG, int D>
__global__ __launch_bounds__(KS_TILE_THREADS) void keyswitch_tiled_comb_kernel(KeyswitchArgs a) {
  constexpr int BASE = 1 << BASEBIT;
  constexpr int NG = (T + D - 1) / D;              // lookups per coefficient
  constexpr int RG = 1 << (BASEBIT * D);           // rows of a full group
  constexpr int DL = T - (NG - 1) * D;             // digits of the last group (1 ... D)
  constexpr int BROWS = KS_IG * T * BASE, TROWS = KS_IG * NG * RG;
  __shared__ __attribute__((aligned(16))) int32_t s_base[BROWS * KS_CHP];
  __shared__ __attribute__((aligned(16))) int32_t s_tab[TROWS * KS_CHP];
  const int tid = threadIdx.x;
  const long ct = (long)blockIdx.x * KS_TILE_THREADS + tid;
  const bool live = ct < a.B;
  const int w0 = (int)blockIdx.y * KS_CH;
  const int W = a.W, N = a.N;
  const int32_t* u0 = a.u0 + (live ? ct : 0) * (size_t)(N + 1);
  const int32_t* u1 = a.u1 ? a.u1 + (live ? ct : 0) * (size_t)(N + 1) : nullptr;
  for (int e = tid; e < BROWS * KS_CHP; e += KS_TILE_THREADS) s_base[e] = 0;   // the v = 0 rows stay zero

  uint32_t acc[KS_CH];
#pragma unroll
  for (int k = 0; k < KS_CH; ++k) acc[k] = 0;
  constexpr uint32_t prec_offset = 1u << (32 - (1 + BASEBIT * T));
  constexpr int SEGS = KS_IG * T * (BASE - 1);
  // Base rows of the next group: requested into registers BEFORE the current group's lookups and put into s_base after them, so
  // that their L2 round trip runs behind the lookups (8 threads x 16 B per row segment; rows are only 4-byte aligned in general)
  constexpr int NST = (SEGS * 8 + KS_TILE_THREADS - 1) / KS_TILE_THREADS;
  int32_t st[NST][4];
  auto stage_load = [&](int i0) {
#pragma unroll
    for (int sl = 0; sl < NST; ++sl) {
      const int sidx = tid + sl * KS_TILE_THREADS;
      const int seg = sidx >> 3, part = sidx & 7;
      const int v = seg % (BASE - 1) + 1;
      const int ij = seg / (BASE - 1);          // ii * T + j
      const int wq = w0 + part * 4;
      const int32_t* src = a.ksk + (((size_t)i0 * T + ij) * BASE + v) * (size_t)W + wq;
#pragma unroll
      for (int e = 0; e < 4; ++e) st[sl][e] = (sidx < SEGS * 8 && wq + e < W) ? src[e] : 0;
    }
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int sl = 0; sl < NST; ++sl) {
      const int sidx = tid + sl * KS_TILE_THREADS;
      if (sidx < SEGS * 8) {
        const int seg = sidx >> 3, part = sidx & 7;
        const int v = seg % (BASE - 1) + 1;
        const int ij = seg / (BASE - 1);
        *reinterpret_cast<int4*>(&s_base[(ij * BASE + v) * KS_CHP + part * 4]) = make_int4(st[sl][0], st[sl][1], st[sl][2], st[sl][3]);
      }
    }
  };
  // row `comb` of group gq of coefficient ii = the sum over the group's digits k of base row (ii, gq D + k, digit k of comb),
  // digit 0 the most significant (as it sits in the coefficient)
  auto build = [&]() {
#pragma unroll 2
    for (int item = tid; item < TROWS * 8; item += KS_TILE_THREADS) {
      const int part = item & 7, r = item >> 3;
      const int comb = r % RG, igq = r / RG, gq = igq % NG, ii = igq / NG;
      const int dl = gq == NG - 1 ? DL : D;
      if (comb >> (BASEBIT * dl)) continue;       // rows a short last group never selects
      int4 sum = make_int4(0, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < D; ++k) {
        if (k < dl) {
          const int dgt = ks_comb_digit(comb, k, dl, BASEBIT);   // rs_host.h (host-tested)
          const int4 v = *reinterpret_cast<const int4*>(&s_base[((ii * T + gq * D + k) * BASE + dgt) * KS_CHP + part * 4]);
          sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
      }
      *reinterpret_cast<int4*>(&s_tab[r * KS_CHP + part * 4]) = sum;
    }
  };

  const int groups_per_split = (N / KS_IG) / (int)gridDim.z;
  const int g_begin = (int)blockIdx.z * groups_per_split, g_end = g_begin + groups_per_split;
  __syncthreads();
  stage_load(g_begin * KS_IG);
  stage_store();
  __syncthreads();
  build();
  __syncthreads();
  for (int g = g_begin; g < g_end; ++g) {
    if (g + 1 < g_end) stage_load((g + 1) * KS_IG);
    const int i0 = g * KS_IG;
    uint32_t ai[KS_IG];
#pragma unroll
    for (int ii = 0; ii < KS_IG; ++ii) {
      uint32_t v = live ? (uint32_t)u0[i0 + ii] : 0u;
      if (u1 && live) v += (uint32_t)u1[i0 + ii];
      ai[ii] = live ? v + prec_offset : 0u;
    }
#pragma unroll 1
    for (int ii = 0; ii < KS_IG; ++ii) {
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) {
        constexpr int dl_full = D;
        const int dl = gq == NG - 1 ? DL : dl_full;
        const uint32_t comb = ks_comb_index(ai[ii], gq, D, dl, BASEBIT);
        const int4* row = reinterpret_cast<const int4*>(&s_tab[((ii * NG + gq) * RG + (int)comb) * KS_CHP]);
#pragma unroll
        for (int q = 0; q < KS_CH / 4; ++q) {
          const int4 r = row[q];
          acc[4 * q + 0] += (uint32_t)r.x;
          acc[4 * q + 1] += (uint32_t)r.y;
          acc[4 * q + 2] += (uint32_t)r.z;
          acc[4 * q + 3] += (uint32_t)r.w;
        }
      }
    }
    if (g + 1 < g_end) stage_store();   // s_base is free: its sums were built before the last barrier
    __syncthreads();   // the next base rows are in s_base; every lane has finished with s_tab
    if (g + 1 < g_end) build();
    __syncthreads();   // the next sums are in s_tab; s_base is free
  }
  if (!live) return;
  ks_tile_finish(a, acc, ct, w0, u0, u1);
}

// Sliced keyswitch, second step: out[ct][w] = (w == n ? b word : 0) - sum over slices of scratch[slice][w][ct]. A workgroup
// owns a 16 x 16 tile: the partial sums are read with the ciphertext index fastest (as the slices wrote them), transposed
// through LDS and stored with the word index fastest. Exact in any order (wrapping 32-bit sums).
constexpr int KS_RT = 16;
__global__ __launch_bounds__(KS_RT * KS_RT) void keyswitch_reduce_kernel(KeyswitchArgs a, int slices) {
  __shared__ uint32_t tile[KS_RT][KS_RT + 1];
  const int tx = threadIdx.x % KS_RT, ty = threadIdx.x / KS_RT;
  const long ct0 = (long)blockIdx.x * KS_RT;
  const int w0 = (int)blockIdx.y * KS_RT, W = a.W, N = a.N;
  {
    const int w = w0 + ty;
    const long ct = ct0 + tx;
    uint32_t s = 0;
    if (w < W && ct < a.B) {
      const uint32_t* p = a.scratch + (size_t)w * (size_t)a.B + ct;
      const size_t stride = (size_t)W * (size_t)a.B;
      int z = 0;
      for (; z + 8 <= slices; z += 8) {
        uint32_t v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p[(size_t)(z + e) * stride];
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
      }
      for (; z < slices; ++z) s += p[(size_t)z * stride];
    }
    tile[ty][tx] = s;
  }
  __syncthreads();
  const long ct = ct0 + ty;
  const int w = w0 + tx;
  if (ct < a.B && w < W) {
    uint32_t v = 0u - tile[tx][ty];
    if (w == W - 1) {
      uint32_t bw = (uint32_t)a.u0[ct * (size_t)(N + 1) + N];
      if (a.u1) bw += (uint32_t)a.u1[ct * (size_t)(N + 1) + N];
      v += bw + (uint32_t)a.bconst;
    }
    a.out[ct * W + w] = (int32_t)v;
  }
}

// -------------------------------------------------------------------------------------------------
// LWE word arithmetic
// -------------------------------------------------------------------------------------------------
__global__ void lincomb_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ x, int32_t cx, const int32_t* __restrict__ y,
                               int32_t cy, int32_t bconst, int W, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    uint32_t v = (uint32_t)cx * (uint32_t)x[e];
    if (y) v += (uint32_t)cy * (uint32_t)y[e];
    if ((int)(e % W) == W - 1) v += (uint32_t)bconst;
    out[e] = (int32_t)v;
  }
}

__global__ void synthetic_words_kernel(int32_t* __restrict__ out, uint64_t seed, size_t total) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x)
    out[e] = (int32_t)synthetic_key_word(seed, (uint64_t)e);
}

__global__ void gather_rows_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ in, const int32_t* __restrict__ idx, int W,
                                   long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long row = e / W;
    const int w = (int)(e - row * W);
    const int src = idx[row];
    out[e] = src < 0 ? 0 : in[(size_t)src * W + w];
  }
}

// out[m][w] = sum_k s(k,m) in[k][w] (+ constants on the b word). A workgroup is 64 words x 4 k-lanes
// (one wave per k-lane, so the weight bytes are wave-uniform), grid (ceil(W/64), M, S): with few
// outputs (the 10 logits of a final layer: K = 1024 serial taps per thread took 0.41 ms) the K taps
// are also cut into S slices whose partial sums meet by integer atomics in a zeroed output -- exact
// and order-independent, like every other sum of this stage.
constexpr int FC_WORDS = 64, FC_KL = 4;
__global__ __launch_bounds__(FC_WORDS * FC_KL) void linear_fc_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ in,
                                                                      const uint8_t* __restrict__ sign, const uint8_t* __restrict__ zero,
                                                                      int K, int M, int W, int32_t zero_tap_b,
                                                                      const int32_t* __restrict__ bias_b, int bias_depth, int kslice) {
  __shared__ uint32_t s_acc[FC_KL][FC_WORDS];
  __shared__ uint32_t s_nz[FC_KL][FC_WORDS];
  const int wl = threadIdx.x & (FC_WORDS - 1);
  const int kl = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / FC_WORDS));
  const int w = blockIdx.x * FC_WORDS + wl;
  const int m = blockIdx.y;
  const int k0 = blockIdx.z * kslice;
  const int k1 = (k0 + kslice < K) ? k0 + kslice : K;
  uint32_t acc = 0;
  uint32_t nzero = 0;
  if (w < W) {
    for (int k = k0 + kl; k < k1; k += FC_KL) {
      const size_t fi = (size_t)k * M + m;
      if (zero && zero[fi]) { ++nzero; continue; }
      const uint32_t v = (uint32_t)in[(size_t)k * W + w];
      acc += sign[fi] ? v : (0u - v);
    }
  }
  s_acc[kl][wl] = acc;
  s_nz[kl][wl] = nzero;
  __syncthreads();
  if (kl != 0 || w >= W) return;
#pragma unroll
  for (int j = 1; j < FC_KL; ++j) { acc += s_acc[j][wl]; nzero += s_nz[j][wl]; }
  if (w == W - 1) {
    acc += nzero * (uint32_t)zero_tap_b;
    if (bias_b && blockIdx.z == 0) acc += (uint32_t)bias_b[m % bias_depth];
  }
  if (gridDim.z > 1) atomicAdd(reinterpret_cast<unsigned int*>(out) + (size_t)m * W + w, acc);
  else out[(size_t)m * W + w] = (int32_t)acc;
}

// out[oh][ow][od][w]; grid (ceil(W/256), Cout, Ho*Wo)
__global__ __launch_bounds__(256) void conv_ternary_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ in,
                                                           const uint8_t* __restrict__ sign, const uint8_t* __restrict__ zero,
                                                           ConvShape s, int W, int32_t zero_tap_b, int32_t pad_tap_b,
                                                           const int32_t* __restrict__ bias_b, int bias_depth) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  const int od = blockIdx.y;
  const int pix = blockIdx.z;
  if (w >= W) return;
  const int oh = pix / s.Wo, ow = pix % s.Wo;
  uint32_t acc = 0, nzero = 0, npad = 0;
  for (int fh = 0; fh < s.fh; ++fh) {
    const int ih = fh + oh * s.stride_h - s.off_h;
    for (int fw = 0; fw < s.fw; ++fw) {
      const int iw = fw + ow * s.stride_w - s.off_w;
      const bool oob = (unsigned)ih >= (unsigned)s.H || (unsigned)iw >= (unsigned)s.Wd;
      for (int di = 0; di < s.Cin; ++di) {
        const size_t fi = (((size_t)fh * s.fw + fw) * s.Cin + di) * s.Cout + od;
        if (!oob && !(zero && zero[fi])) {
          const uint32_t v = (uint32_t)in[(((size_t)ih * s.Wd + iw) * s.Cin + di) * W + w];
          acc += sign[fi] ? v : (0u - v);
        } else if (zero && zero[fi]) {
          ++nzero;  // reference order: ternary-zero test precedes the padding branch
        } else {
          ++npad;
        }
      }
    }
  }
  if (w == W - 1) {
    acc += nzero * (uint32_t)zero_tap_b + npad * (uint32_t)pad_tap_b;
    if (bias_b) acc += (uint32_t)bias_b[od % bias_depth];
  }
  out[(((size_t)oh * s.Wo + ow) * s.Cout + od) * W + w] = (int32_t)acc;
}

// ---- register-tiled ternary convolution (the throughput form; used when the constant taps are 0) ----
// The reference's ENCRYPTED convolution (lib/BinFunc.cpp:195-320) is a ternary (+1 / -1 / 0) "GEMM" over
// LWE words: out[pix][od][w] = sum_taps s(tap, od) * in[tap(pix)][w]. One thread owns one word index w of
// one output pixel and kConvTile output channels at once, so every input word is loaded ONCE per 32
// channels (the first kernel re-read it for every channel: L2-bound at 1.5 % of the integer-add rate).
// Weights are pre-expanded (conv_expand_kernel) to two masks per (tap, channel): nz = 0 for a ternary
// zero else ~0, ng = ~0 for -1 else 0; they are wave-uniform, live in scalar registers, and
//   acc += ((v & nz) ^ ng)          -- v_and_b32 + v_xad_u32, 2 vector ops per ternary MAC
// gives +v, ~v = -v - 1 or 0. The missing "+1" of every negative tap is the per-channel constant
// negtotal[od]; padding taps run with v = 0 and so stay consistent with it. Integer wrap-around adds are
// order-independent, hence bit-exact against any other summation order.
constexpr int kConvTile = 32;

// masks[k][od_padded][2], negtotal[od_padded]; grid over k*od_padded / 256
__global__ __launch_bounds__(256) void conv_expand_kernel(uint32_t* __restrict__ masks, const uint8_t* __restrict__ sign,
                                                          const uint8_t* __restrict__ zero, int K, int Cout, int Cpad) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)K * Cpad) return;
  const int k = (int)(i / Cpad), od = (int)(i % Cpad);
  uint32_t nz = 0, ng = 0;
  if (od < Cout) {
    const size_t fi = (size_t)k * Cout + od;
    const bool z = zero && zero[fi];
    nz = z ? 0u : 0xFFFFFFFFu;
    ng = (!z && !sign[fi]) ? 0xFFFFFFFFu : 0u;
  }
  masks[2 * i] = nz;
  masks[2 * i + 1] = ng;
}
__global__ __launch_bounds__(256) void conv_negtotal_kernel(uint32_t* __restrict__ negtotal, const uint32_t* __restrict__ masks, int K, int Cpad) {
  const int od = blockIdx.x * 256 + threadIdx.x;
  if (od >= Cpad) return;
  uint32_t n = 0;
  for (int k = 0; k < K; ++k) n += masks[2 * ((size_t)k * Cpad + od) + 1] & 1u;
  negtotal[od] = n;
}

// grid (ceil(W/128), Cpad/32, Ho*Wo)
__global__ __launch_bounds__(128) void conv_ternary_tiled_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ in,
                                                                 const uint32_t* __restrict__ masks, const uint32_t* __restrict__ negtotal,
                                                                 ConvShape s, int Cpad, int W, const int32_t* __restrict__ bias_b, int bias_depth) {
  const int w = blockIdx.x * 128 + threadIdx.x;
  const int od0 = blockIdx.y * kConvTile;
  const int pix = blockIdx.z;
  const int oh = pix / s.Wo, ow = pix % s.Wo;
  const int wc = w < W ? w : W - 1;   // lanes past the row compute a duplicate and do not store
  uint32_t acc[kConvTile];
#pragma unroll
  for (int c = 0; c < kConvTile; ++c) acc[c] = 0;
  const uint32_t* mrow = masks + 2 * (size_t)od0;
  for (int fh = 0; fh < s.fh; ++fh) {
    const int ih = fh + oh * s.stride_h - s.off_h;
    for (int fw = 0; fw < s.fw; ++fw) {
      const int iw = fw + ow * s.stride_w - s.off_w;
      const bool oob = (unsigned)ih >= (unsigned)s.H || (unsigned)iw >= (unsigned)s.Wd;
      const int32_t* src = in + (((size_t)(oob ? 0 : ih) * s.Wd + (oob ? 0 : iw)) * s.Cin) * W + wc;
      for (int di = 0; di < s.Cin; ++di) {
        const uint32_t v = oob ? 0u : (uint32_t)src[(size_t)di * W];
        const uint32_t* m = mrow + 2 * (size_t)((fh * s.fw + fw) * s.Cin + di) * Cpad;   // wave-uniform: scalar loads
#pragma unroll
        for (int c = 0; c < kConvTile; ++c) {
          const uint32_t t = v & m[2 * c];
          // acc = (t ^ ng) + acc in one instruction (hipcc emits v_xor + v_add for the C expression)
          asm("v_xad_u32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(t), "s"(m[2 * c + 1]));
        }
      }
    }
  }
  if (w >= W) return;
#pragma unroll
  for (int c = 0; c < kConvTile; ++c) {
    const int od = od0 + c;
    if (od >= s.Cout) break;
    uint32_t r = acc[c] + negtotal[od];
    if (w == W - 1 && bias_b) r += (uint32_t)bias_b[od % bias_depth];
    out[(((size_t)oh * s.Wo + ow) * s.Cout + od) * W + w] = (int32_t)r;
  }
}

// grid (ceil(W/256), C, Ho*Wo)
__global__ __launch_bounds__(256) void sumpool_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ in, PoolShape s, int W,
                                                      const int32_t* __restrict__ bias_b, int bias_depth) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  const int pix = blockIdx.z;
  if (w >= W) return;
  const int oh = pix / s.Wo, ow = pix % s.Wo;
  uint32_t acc = 0;
  for (int fh = 0; fh < s.win_h; ++fh) {
    const int ih = oh * s.stride_h - s.off_h + fh;
    if (ih < 0 || ih >= s.H) continue;
    for (int fw = 0; fw < s.win_w; ++fw) {
      const int iw = ow * s.stride_w - s.off_w + fw;
      if (iw < 0 || iw >= s.Wd) continue;
      acc += (uint32_t)in[(((size_t)ih * s.Wd + iw) * s.C + c) * W + w];
    }
  }
  if (w == W - 1 && bias_b) acc += (uint32_t)bias_b[c % bias_depth];
  out[(((size_t)oh * s.Wo + ow) * s.C + c) * W + w] = (int32_t)acc;
}

// -------------------------------------------------------------------------------------------------
// Launchers
// -------------------------------------------------------------------------------------------------
// Slices of the N input coefficients for a tiled launch of a.B ciphertexts (1 = plain-store throughput form): doubled until
// ~1024 workgroups exist, at most 64, every slice at least `min_per_slice` staging groups... coefficients long.
static bool ks_tiled_shape(const KeyswitchArgs& a) {
  return (a.t == 8 && a.basebit == 2) || (a.t == 9 && a.basebit == 3) || (a.t == 18 && a.basebit == 1);
}
static_assert(KS_TILE_THREADS == 256 && KS_CH == 32, "rs_host.h keyswitch_slices() restates this grid");
static unsigned ks_slices(const KeyswitchArgs& a) {
  if (!ks_tiled_shape(a) || a.B <= 0) return 1;
  return keyswitch_slices(a.B, a.W, a.N);   // host logic (rs_host.h), CPU-tested through the emulator library
}
size_t keyswitch_scratch_words(const KeyswitchArgs& a) {
  const unsigned split = ks_slices(a);
  return split > 1 ? (size_t)split * (size_t)a.W * (size_t)a.B : 0;
}

hipError_t launch_keyswitch(const KeyswitchArgs& a_in, hipStream_t st) {
  if (a_in.B <= 0) return hipSuccess;
  KeyswitchArgs a = a_in;
  // (basebit = 1, the (18, 1) keys of redsec_params_small / medium / large: a digit is one bit and the selected row is the same
  // for every lane, so a form that reads the rows with SCALAR loads and adds them under the execution mask -- no LDS at all --
  // was built and measured in round 3: bit-exact, and slower than the tiled kernel, 21.7 against 14.7 ms per 1,024 medium
  // ciphertexts, 44.7 against 29.1 ms per 512 large ones, 9.8 against 2.2 ms per 4,096 small ones: two 64-byte scalar loads per
  // (i, j) with ~100 scalar registers to keep three rows in flight is latency-bound. Removed again; profiles/r03/t_*.)
  dim3 grid((unsigned)((a.B + KS_TILE_THREADS - 1) / KS_TILE_THREADS), (unsigned)((a.W + KS_CH - 1) / KS_CH), 1);
  // tiled forms: the two shipped shapes and (18, 1) of redsec_params_small / medium / large; any power-of-two ring
  const bool tiled = ks_tiled_shape(a);
  // small batches: slice the input coefficients until enough workgroups exist (latency form). The slices leave their partial
  // sums in the scratch and a second kernel adds them up; without a scratch they add into a zeroed output with integer atomics.
  const unsigned split = ks_slices(a);
  grid.z = split;
  const bool two_step = split > 1 && a.scratch && a.scratch_words >= keyswitch_scratch_words(a);
  if (!two_step) a.scratch = nullptr;
  if (split > 1 && !two_step) {
    hipError_t e = hipMemsetAsync(a.out, 0, (size_t)a.B * a.W * sizeof(int32_t), st);
    if (e != hipSuccess) return e;
  }
  if (tiled && a.t == 8) {
    hipLaunchKernelGGL((keyswitch_tiled_comb_kernel<8, 2, 4, 2>), grid, dim3(KS_TILE_THREADS), 0, st, a);
  } else if (tiled && a.t == 9) {
    hipLaunchKernelGGL((keyswitch_tiled_kernel<9, 3, 2>), grid, dim3(KS_TILE_THREADS), 0, st, a);
  } else if (tiled) {
    hipLaunchKernelGGL((keyswitch_tiled_comb_kernel<18, 1, 4, 4>), grid, dim3(KS_TILE_THREADS), 0, st, a);
  } else {
    // generic gather form: any ring degree, any (t, basebit), any sample width
    const unsigned wy = (unsigned)((a.W + KS_THREADS * KS_MAXR - 1) / (KS_THREADS * KS_MAXR));
    hipLaunchKernelGGL(keyswitch_kernel, dim3((unsigned)a.B, wy), dim3(KS_THREADS), 0, st, a);
  }
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  if (two_step) {
    const dim3 rgrid((unsigned)((a.B + KS_RT - 1) / KS_RT), (unsigned)((a.W + KS_RT - 1) / KS_RT));
    hipLaunchKernelGGL(keyswitch_reduce_kernel, rgrid, dim3(KS_RT * KS_RT), 0, st, a, (int)split);
  }
  return hipGetLastError();
}

hipError_t launch_lincomb(int32_t* out, const int32_t* x, int32_t cx, const int32_t* y, int32_t cy, int32_t bconst, int W, long B,
                          hipStream_t st) {
  const long total = B * W;
  if (total <= 0) return hipSuccess;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(lincomb_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, x, cx, y, cy, bconst, W, total);
  return hipGetLastError();
}

hipError_t launch_synthetic_words(int32_t* out, uint64_t seed, size_t total, hipStream_t st) {
  if (total == 0) return hipSuccess;
  size_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(synthetic_words_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, seed, total);
  return hipGetLastError();
}

hipError_t launch_gather_rows(int32_t* out, const int32_t* in, const int32_t* idx, int W, long B, hipStream_t st) {
  const long total = B * W;
  if (total <= 0) return hipSuccess;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, in, idx, W, total);
  return hipGetLastError();
}

hipError_t launch_linear_fc(int32_t* out, const int32_t* in, const uint8_t* sign, const uint8_t* zero, int K, int M, int W,
                            int32_t zero_tap_b, const int32_t* bias_b, int bias_depth, hipStream_t st) {
  // enough workgroups to fill the chip: slice K when (words x outputs) alone gives fewer than ~1024 of them
  const int wblocks = (W + FC_WORDS - 1) / FC_WORDS;
  long S = (1024 + (long)wblocks * M - 1) / ((long)wblocks * M);
  const long max_s = (K + 4 * FC_KL - 1) / (4 * FC_KL);           // at least 4 taps per thread
  if (S > max_s) S = max_s;
  if (S < 1) S = 1;
  int kslice = (int)((K + S - 1) / S);
  kslice = (kslice + FC_KL - 1) / FC_KL * FC_KL;
  if (kslice < FC_KL) kslice = FC_KL;                              // K == 0: one (empty) slice writes the constants
  S = (K + kslice - 1) / kslice;
  if (S < 1) S = 1;
  if (S > 1) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(int32_t) * (size_t)M * W, st);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(linear_fc_kernel, dim3(wblocks, M, (unsigned)S), dim3(FC_WORDS * FC_KL), 0, st, out, in, sign, zero, K, M, W,
                     zero_tap_b, bias_b, bias_depth, kslice);
  return hipGetLastError();
}

hipError_t launch_conv_ternary(int32_t* out, const int32_t* in, const uint8_t* sign, const uint8_t* zero, const ConvShape& s, int W,
                               int32_t zero_tap_b, int32_t pad_tap_b, const int32_t* bias_b, int bias_depth, hipStream_t st) {
  hipLaunchKernelGGL(conv_ternary_kernel, dim3((W + 255) / 256, s.Cout, s.Ho * s.Wo), dim3(256), 0, st, out, in, sign, zero, s, W,
                     zero_tap_b, pad_tap_b, bias_b, bias_depth);
  return hipGetLastError();
}

// scratch: uint32[2 * K * Cpad + Cpad] with Cpad = Cout rounded up to kConvTile (conv_tiled_scratch_words)
size_t conv_tiled_scratch_words(const ConvShape& s) {
  const size_t K = (size_t)s.fh * s.fw * s.Cin, Cpad = (size_t)(s.Cout + kConvTile - 1) / kConvTile * kConvTile;
  return 2 * K * Cpad + Cpad;
}
hipError_t launch_conv_ternary_tiled(int32_t* out, const int32_t* in, const uint8_t* sign, const uint8_t* zero, const ConvShape& s, int W,
                                     const int32_t* bias_b, int bias_depth, uint32_t* scratch, hipStream_t st) {
  const int K = s.fh * s.fw * s.Cin, Cpad = (s.Cout + kConvTile - 1) / kConvTile * kConvTile;
  uint32_t* masks = scratch;
  uint32_t* negtotal = scratch + 2 * (size_t)K * Cpad;
  const long n = (long)K * Cpad;
  hipLaunchKernelGGL(conv_expand_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, masks, sign, zero, K, s.Cout, Cpad);
  hipLaunchKernelGGL(conv_negtotal_kernel, dim3((Cpad + 255) / 256), dim3(256), 0, st, negtotal, masks, K, Cpad);
  hipLaunchKernelGGL(conv_ternary_tiled_kernel, dim3((W + 127) / 128, Cpad / kConvTile, s.Ho * s.Wo), dim3(128), 0, st, out, in, masks,
                     negtotal, s, Cpad, W, bias_b, bias_depth);
  return hipGetLastError();
}

// Box calibration for bench.py: what the FP64 vector pipes of THIS device sustain right now. One workgroup of 8 waves per CU (two
// per SIMD, the occupancy of the blind-rotation kernels), 16 independent fused multiply-add chains per lane and nothing else in
// the loop. The boxes of a pool differ by several per cent in exactly this number (clocks under their power limit); a bench
// line carries it so that a figure from a slow box can be told from a slow kernel.
__global__ __launch_bounds__(512) void fp64_rate_kernel(double* out, double b, double c, int iters) {
  double x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = (double)threadIdx.x * 1e-9 + (double)k;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = __builtin_fma(x[k], b, c);
  }
  double sum = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) sum += x[k];
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = sum;
}
// lane_ops = the FP64 lane-operations one launch performs (an FMA counted once, as in bench.py's roofline_valu)
hipError_t launch_fp64_rate(double* out, int num_cus, int iters, double* lane_ops, hipStream_t st) {
  hipLaunchKernelGGL(fp64_rate_kernel, dim3((unsigned)num_cus), dim3(512), 0, st, out, 0.999999, 1e-7, iters);
  *lane_ops = (double)num_cus * 512.0 * 16.0 * (double)iters;
  return hipGetLastError();
}

hipError_t launch_sumpool(int32_t* out, const int32_t* in, const PoolShape& s, int W, const int32_t* bias_b, int bias_depth,
                          hipStream_t st) {
  hipLaunchKernelGGL(sumpool_kernel, dim3((W + 255) / 256, s.C, s.Ho * s.Wo), dim3(256), 0, st, out, in, s, W, bias_b, bias_depth);
  return hipGetLastError();
}

}  // namespace rs
