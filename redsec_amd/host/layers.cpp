// layers.cpp -- host side of the encrypted layer API on the MI355X backend.
//
// Mirrors, for the ENCRYPTED flavour, the reference's
//   lib/Layer.cpp            allocation helpers                      (Layer.cpp:27-194)
//   lib/BinOps_enc.cpp       per-ciphertext binary primitives        (BinOps_enc.cpp:27-305)
//   lib/IntOps_enc.cpp       per-ciphertext integer primitives       (IntOps_enc.cpp:20-84)
//   lib/BinLayer.cpp, lib/IntLayer.cpp   layer orchestration conv -> sumpool -> quantize -> maxpool
//                            (BinLayer.cpp:150-241, IntLayer.cpp:153-235) with the geometry of
//                            lib/BinFunc.cpp / lib/IntFunc.cpp prep() functions
// with the same names, argument meaning and ownership (each execute() frees its input and returns
// freshly allocated host ciphertext arrays), so nets/*/*/net.cpp and main.cpp link unchanged.
//
// What differs is HOW a stage runs: instead of an OpenMP loop of per-ciphertext TFHE calls
// (e.g. BinFunc.cpp:1056-1071) every stage is one batched launch through the C ABI
// (include/redsec_hip.h) on device-resident ciphertext slabs int32[count][n+1]. A layer's output is
// also kept on the device and keyed by the host pointer it returns, so the next layer's execute()
// finds its input already in HBM; only the image goes up and the logits come down.
#include <algorithm>
#include <cassert>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <iterator>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "lib/BinLayer.h"
#include "lib/BinOps_enc.h"
#include "lib/IntLayer.h"
#include "lib/IntOps_enc.h"
#include "lib/Layer.h"
#include "redsec_hip.h"

// =================================================================================================
// Layer.cpp helpers
// =================================================================================================
void print_status(const char* s) { printf("%s", s); }   // _PRINT_STATUS_ is on in the ENCRYPTED flavour (Layer.h:25-28)

uint64_t get_size(tRectangle* ws, uint16_t in_dep, uint16_t out_dep) { return (uint64_t)(ws->h) * (ws->w) * in_dep * out_dep; }

void netParamsCpy(tNetParams* dest, tNetParams* src) {
  dest->conv = src->conv; dest->pool = src->pool; dest->bnorm = src->bnorm;
  dest->e_bias = src->e_bias; dest->version = src->version;
}

tBit* bit_calloc(uint32_t len, TFheGateBootstrappingCloudKeySet* bk) { return new_gate_bootstrapping_ciphertext_array((int32_t)len, bk->params); }

// The drivers release result arrays with free() (nets/mnist/sign1024x1/main.cpp:87), so the struct
// arrays come from calloc.
tMultiBit* mbit_calloc(uint32_t len, uint8_t bits, TFheGateBootstrappingCloudKeySet* bk) {
  tMultiBit* ret = (tMultiBit*)calloc(len ? len : 1, sizeof(tMultiBit));
  for (uint32_t i = 0; i < len; ++i) {
    ret[i].size = bits;
    ret[i].ctxt = new_gate_bootstrapping_ciphertext_array(bits, bk->params);
  }
  return ret;
}
tFixedPoint* fixpt_calloc(uint32_t len, uint8_t bits, TFheGateBootstrappingCloudKeySet* bk) { return mbit_calloc(len, bits, bk); }  // samples start cleared

void bit_free(uint32_t len, tBit* to_free) { delete_gate_bootstrapping_ciphertext_array((int32_t)len, to_free); }
void mbit_free(uint32_t len, tMultiBit* to_free) {
  for (uint32_t i = 0; i < len; ++i) delete_gate_bootstrapping_ciphertext_array((int32_t)to_free[i].size, to_free[i].ctxt);
  free(to_free);
}
void fixpt_free(uint32_t len, tFixedPoint* to_free) { mbit_free(len, to_free); }

// =================================================================================================
// BinOps / IntOps: per-ciphertext primitives
// =================================================================================================
namespace BinOps {

void multiply(tBit* result, const tBit* a, const uint8_t b, TFheGateBootstrappingCloudKeySet* bk) {
  if (b == 0) bootsNOT(result, a, bk); else bootsCOPY(result, a, bk);   // XNOR with a plaintext bit
}
void multiply_pc_ints(LweSample* result, LweSample* in1, const uint32_t* multicand, uint8_t, uint8_t, TFheGateBootstrappingCloudKeySet* bk) {
  lweAddMulTo(result, (int32_t)*multicand, in1, bk->params->in_out_params);
}
void add_bit(tMultiBit* result, const tBit* a, const tBit* b, TFheGateBootstrappingCloudKeySet* bk) {
  result->ctxt = new_gate_bootstrapping_ciphertext_array(2, bk->params);
  result->size = 2;
  bootsXOR(&result->ctxt[0], a, b, bk);
  bootsAND(&result->ctxt[1], a, b, bk);
}
// Ripple-carry adder: per bit 2 XOR + 2 AND + 1 OR bootstraps. The reference initialises the carry
// with bootsCOPY(&carry, 0, bk) -- a null source (BinOps_enc.cpp:99); here carry_0 is the trivial 0.
void add(tMultiBit* result, const tMultiBit* a, const tMultiBit* b, uint8_t bits, TFheGateBootstrappingCloudKeySet* bk) {
  uint32_t sz = bits ? bits : (a->size > b->size ? a->size : b->size);
  if (sz != result->size) {
    delete_gate_bootstrapping_ciphertext_array((int32_t)result->size, result->ctxt);
    result->size = sz;
    result->ctxt = new_gate_bootstrapping_ciphertext_array((int32_t)sz, bk->params);
  }
  tBit* x = new_gate_bootstrapping_ciphertext_array((int32_t)sz, bk->params);
  tBit* y = new_gate_bootstrapping_ciphertext_array((int32_t)sz, bk->params);
  tBit* carry = new_gate_bootstrapping_ciphertext_array((int32_t)sz + 1, bk->params);
  tBit* t = new_gate_bootstrapping_ciphertext_array(3, bk->params);
  for (uint32_t i = 0; i < sz; ++i) {
    if (i >= a->size) bootsCONSTANT(&x[i], 0, bk); else bootsCOPY(&x[i], &a->ctxt[i], bk);
    if (i >= b->size) bootsCONSTANT(&y[i], 0, bk); else bootsCOPY(&y[i], &b->ctxt[i], bk);
  }
  bootsCONSTANT(&carry[0], 0, bk);
  for (uint32_t i = 0; i + 1 < sz; ++i) {
    bootsXOR(&t[0], &x[i], &y[i], bk);
    bootsXOR(&result->ctxt[i], &carry[i], &t[0], bk);
    bootsAND(&t[1], &carry[i], &t[0], bk);
    bootsAND(&t[2], &x[i], &y[i], bk);
    bootsOR(&carry[i + 1], &t[1], &t[2], bk);
  }
  bootsXOR(&t[0], &x[sz - 1], &y[sz - 1], bk);
  bootsXOR(&result->ctxt[sz - 1], &carry[sz - 1], &t[0], bk);
  delete_gate_bootstrapping_ciphertext_array((int32_t)sz, x);
  delete_gate_bootstrapping_ciphertext_array((int32_t)sz, y);
  delete_gate_bootstrapping_ciphertext_array((int32_t)sz + 1, carry);
  delete_gate_bootstrapping_ciphertext_array(3, t);
}
void add_int(LweSample* result, const LweSample* a, const LweSample* b, TFheGateBootstrappingCloudKeySet* bk) {
  const LweParams* p = bk->params->in_out_params;
  lweClear(result, p); lweAddTo(result, a, p); lweAddTo(result, b, p);
}
void add_int_inplace(LweSample* result, const LweSample* a, TFheGateBootstrappingCloudKeySet* bk) { lweAddTo(result, a, bk->params->in_out_params); }
void add_pc_ints(LweSample* result, LweSample* in1, const uint16_t* addend, uint8_t, TFheGateBootstrappingCloudKeySet* bk) {
  const LweParams* p = bk->params->in_out_params;
  LweSample* c = new_gate_bootstrapping_ciphertext_array(1, bk->params);
  lweNoiselessTrivial(c, modSwitchToTorus32((int32_t)(*addend & 0xFFFF), MULTIBIT_SPACE), p);
  lweAddTo(result, in1, p); lweAddTo(result, c, p);
  delete_gate_bootstrapping_ciphertext_array(1, c);
}
void inc(tMultiBit* result, const tMultiBit* a, const tBit* b, uint8_t, TFheGateBootstrappingCloudKeySet* bk) {
  tBit* carry = new_gate_bootstrapping_ciphertext_array((int32_t)a->size, bk->params);
  result->size = a->size;
  result->ctxt = new_gate_bootstrapping_ciphertext_array((int32_t)a->size, bk->params);
  bootsCOPY(&carry[0], b, bk);
  for (uint32_t i = 0; i + 1 < a->size; ++i) {
    bootsXOR(&result->ctxt[i], &carry[i], &a->ctxt[i], bk);
    bootsAND(&carry[i + 1], &carry[i], &a->ctxt[i], bk);
  }
  bootsXOR(&result->ctxt[a->size - 1], &carry[a->size - 1], &a->ctxt[a->size - 1], bk);
  delete_gate_bootstrapping_ciphertext_array((int32_t)a->size, carry);
}
void max(tBit* result, const tBit* a, const tBit* b, TFheGateBootstrappingCloudKeySet* bk) { bootsOR(result, a, b, bk); }
int pow_int(int base, int exponent) {
  int r = 1;
  for (; exponent > 0; exponent >>= 1, base *= base) if (exponent & 1) r *= base;
  return r;
}
void binarize_int(LweSample* result, const LweSample* a, const int, TFheGateBootstrappingCloudKeySet* bk) {
  tfhe_bootstrap_FFT(result, bk->bkFFT, modSwitchToTorus32(1, 4096), a);
}
void unbinarize_int(LweSample* result, const LweSample* a, TFheGateBootstrappingCloudKeySet* bk) {
  tfhe_bootstrap_FFT(result, bk->bkFFT, modSwitchToTorus32(1, MULTIBIT_SPACE), a);
}
void binarize(tBit* result, const tMultiBit* a, uint8_t, TFheGateBootstrappingCloudKeySet* bk) { bootsCOPY(result, &a->ctxt[a->size - 1], bk); }
void relu(tFixedPoint* result, tMultiBit* in1, uint8_t in_bits, TFheGateBootstrappingCloudKeySet* bk) {
  for (uint8_t i = 0; i + 1 < in_bits; ++i) bootsAND(&result->ctxt[i], &in1->ctxt[i], &in1->ctxt[in_bits - 1], bk);
}
void shift(tMultiBit* result, tMultiBit* in1, uint8_t in_bits, uint8_t shift_bits, TFheGateBootstrappingCloudKeySet* bk) {
  if (result->size != in_bits) {
    result->size = in_bits;
    result->ctxt = new_gate_bootstrapping_ciphertext_array(in_bits, bk->params);
  }
  assert(in_bits > 0 && shift_bits <= in_bits);
  for (int i = 0; i < in_bits; ++i) {
    const int src = (i + shift_bits > in_bits - 1) ? in_bits - 1 : i + shift_bits;   // sign extend
    bootsCOPY(&result->ctxt[i], &in1->ctxt[src], bk);
  }
}
void get_filters(FILE* fd_in, tBit* p_filt_b, uint32_t len, TFheGateBootstrappingCloudKeySet* bk) {
  std::vector<float> w(len);
  size_t got = fread(w.data(), sizeof(float), len, fd_in); (void)got;
  for (uint32_t i = 0; i < len; ++i) bootsCONSTANT(&p_filt_b[i], w[i] < 0 ? 0 : 1, bk);
}
// Weight records: u8 tag (1 BIN, 2 TERN, 3 UINT32, 4 INT32), then MSB-first bit-packed weights.
void get_ternfilters(FILE* fd_in, uint8_t* p_filt_b, uint8_t* p_tern, uint32_t len, float, TFheGateBootstrappingCloudKeySet*) {
  uint8_t tag = 0;
  size_t got = fread(&tag, 1, 1, fd_in); (void)got;
  assert(tag == 1 || tag == 2);
  const int nbits = tag == 1 ? 1 : 2;
  const size_t nbytes = ((size_t)len * nbits + 7) / 8;
  std::vector<uint8_t> pack(nbytes);
  got = fread(pack.data(), 1, nbytes, fd_in);
  for (uint32_t i = 0; i < len; ++i) {
    const size_t bit = (size_t)i * nbits;
    p_filt_b[i] = (pack[bit >> 3] >> (7 - (bit & 7))) & 1;
    if (p_tern) p_tern[i] = nbits == 2 ? (pack[(bit + 1) >> 3] >> (7 - ((bit + 1) & 7))) & 1 : 0;
  }
}
void get_intfilters(FILE* fd_in, tMultiBit* p_filt_mb, uint32_t len, TFheGateBootstrappingCloudKeySet* bk) {
  const LweParams* p = bk->params->in_out_params;
  uint8_t tag = 0;
  size_t got = fread(&tag, 1, 1, fd_in); (void)got;
  assert(tag == 3 || tag == 4);
  std::vector<int32_t> v(len);
  got = fread(v.data(), sizeof(int32_t), len, fd_in);
  for (uint32_t i = 0; i < len; ++i) {
    p_filt_mb[i].size = 1;
    p_filt_mb[i].ctxt = new_LweSample(p);
    lweNoiselessTrivial(&p_filt_mb[i].ctxt[0], modSwitchToTorus32(v[i], 4096), p);
  }
}
void get_intfilters_ptxt(FILE* fd_in, uint32_t* p_filt_mb, uint32_t len) {
  uint8_t tag = 0;
  size_t got = fread(&tag, 1, 1, fd_in); (void)got;
  assert(tag == 3 || tag == 4);
  got = fread(p_filt_mb, sizeof(uint32_t), len, fd_in);
}

}  // namespace BinOps

namespace IntOps {

void invert(tFixedPoint* result, const tFixedPoint* a, const uint8_t* b, uint8_t, TFheGateBootstrappingCloudKeySet* bk) {
  result->size = a->size;
  result->ctxt = new_gate_bootstrapping_ciphertext_array((int32_t)result->size, bk->params);
  for (uint32_t i = 0; i < result->size; ++i) {
    if (*b == 1) bootsCOPY(&result->ctxt[i], &a->ctxt[i], bk); else bootsNOT(&result->ctxt[i], &a->ctxt[i], bk);
  }
}
void add(tFixedPoint* result, const tFixedPoint* a, const tFixedPoint* b, uint8_t, TFheGateBootstrappingCloudKeySet* bk) {
  BinOps::add_int(&result->ctxt[0], &a->ctxt[0], &b->ctxt[0], bk);
}
void add_inplace(tFixedPoint* result, const tFixedPoint* a, uint8_t, TFheGateBootstrappingCloudKeySet* bk) {
  lweAddTo(&result->ctxt[0], &a->ctxt[0], bk->params->in_out_params);
}
void subtract(tFixedPoint* result, const tFixedPoint* a, const tFixedPoint* b, uint8_t, TFheGateBootstrappingCloudKeySet* bk) {
  const LweParams* p = bk->params->in_out_params;
  result->size = 1;
  result->ctxt = new_LweSample(p);
  lweCopy(result->ctxt, a->ctxt, p);
  lweSubTo(result->ctxt, b->ctxt, p);
}
void relu(tFixedPoint* result, tFixedPoint* in1, uint8_t in_bits, TFheGateBootstrappingCloudKeySet* bk) { BinOps::relu(result, in1, in_bits, bk); }
void shift(tFixedPoint* result, tFixedPoint* in1, uint8_t in_bits, uint8_t shift_bits, TFheGateBootstrappingCloudKeySet* bk) {
  BinOps::shift(result, in1, in_bits, shift_bits, bk);
}

}  // namespace IntOps

// =================================================================================================
// Batched layers
// =================================================================================================
namespace redsec_host {

#define RS_CHECK(call)                                                                  \
  do {                                                                                  \
    if ((call) != 0) { fprintf(stderr, "redsec layers: %s: %s\n", #call, rs_last_error()); abort(); } \
  } while (0)

// A ciphertext array on the device(s): one full replica int32[rows][W] per context of the fleet (one context =
// one GPU; a single GPU unless REDSEC_DEVICES lists several).
// Device buffers come from a per-context cache instead of hipMalloc / hipFree per stage (hipFree waits for the whole device;
// a layer chain allocates and frees a few hundred MB a dozen times per image, always the same few sizes). Blocks are handed
// out again in HOST order without waiting for anything, which is safe because every use of a context's buffers is ordered
// against its device's default stream: launches and the synchronous host copies are enqueued there in program order, and
// rs_allgather_rows -- the one user of OTHER streams (the per-context copy streams) -- makes every copy INTO a block wait for
// an event recorded on the destination's default stream at the start of the call (so kernels still reading the block's
// previous contents finish first) and makes every default stream wait for all copies before it continues.
struct BufPool {
  std::mutex mu;
  std::multimap<size_t, void*> idle;     // size -> block
  std::map<void*, size_t> size_of;       // every block this pool owns
  size_t idle_bytes = 0;
};
std::mutex g_pool_lock;
std::map<rs_ctx*, std::unique_ptr<BufPool>> g_pools;
constexpr size_t kPoolIdleCap = size_t(3) << 30;   // idle bytes kept per context before blocks go back to the runtime
BufPool* pool_of(rs_ctx* c) {
  std::lock_guard<std::mutex> g(g_pool_lock);
  auto& p = g_pools[c];
  if (!p) p.reset(new BufPool);
  return p.get();
}
void* pool_alloc(rs_ctx* c, size_t bytes) {
  if (bytes == 0) bytes = 4;
  BufPool* P = pool_of(c);
  {
    std::lock_guard<std::mutex> g(P->mu);
    auto it = P->idle.lower_bound(bytes);
    if (it != P->idle.end() && it->first <= bytes + bytes / 2 + 4096) {   // close fit: stage sizes recur exactly
      void* p = it->second;
      P->idle_bytes -= it->first;
      P->idle.erase(it);
      return p;
    }
  }
  void* p = nullptr;
  if (rs_dev_alloc(c, &p, bytes) != 0) {
    // out of device memory with blocks idle: give them back and try once more
    std::vector<void*> drop;
    { std::lock_guard<std::mutex> g(P->mu); for (auto& kv : P->idle) { drop.push_back(kv.second); P->size_of.erase(kv.second); } P->idle.clear(); P->idle_bytes = 0; }
    for (void* q : drop) (void)rs_dev_free(c, q);
    if (rs_dev_alloc(c, &p, bytes) != 0) { fprintf(stderr, "redsec layers: rs_dev_alloc(%zu bytes): %s\n", bytes, rs_last_error()); abort(); }
  }
  std::lock_guard<std::mutex> g(P->mu);
  P->size_of[p] = bytes;
  return p;
}
void pool_free(rs_ctx* c, void* p) {
  if (!p) return;
  BufPool* P = pool_of(c);
  std::vector<void*> drop;
  {
    std::lock_guard<std::mutex> g(P->mu);
    auto it = P->size_of.find(p);
    if (it == P->size_of.end()) { drop.push_back(p); }          // not ours (should not happen): plain free
    else {
      P->idle.emplace(it->second, p);
      P->idle_bytes += it->second;
      while (P->idle_bytes > kPoolIdleCap && !P->idle.empty()) { // keep the cache bounded: largest blocks go first
        auto big = std::prev(P->idle.end());
        P->idle_bytes -= big->first;
        P->size_of.erase(big->second);
        drop.push_back(big->second);
        P->idle.erase(big);
      }
    }
  }
  for (void* q : drop) (void)rs_dev_free(c, q);
}

struct DevSlab {
  std::vector<int32_t*> ptr;     // [device]
  std::vector<rs_ctx*> ctx;      // [device]
  size_t rows = 0;
  uint64_t tag = 0;   // fingerprint of the host copy handed out with it
  uint64_t seq = 0;   // publication order
  bool lazy = false;  // REDSEC_LAZY_HOST: the host arrays handed out with it were never filled (see publish)
  bool bits = true;   // the host array is tBit[rows] (else tMultiBit[rows] with one sample each)
  int row_words = 0;  // W of a lazy slab (redsec_materialize)
  void release() { for (size_t d = 0; d < ptr.size(); ++d) pool_free(ctx[d], ptr[d]); ptr.clear(); ctx.clear(); }
};

// Device copies of the ciphertext arrays handed back to the caller, keyed by host pointer. A slab is
// reused only if the host array still carries the ciphertexts it was published with -- EVERY word of every row, by a
// 64-bit content hash: an address recycled for other data, or any sample edited between two stages (legal through the
// BinFunc::* / IntFunc::* stage API and through plain field writes), is uploaded afresh. (Rounds 1-3 hashed 64 sampled rows;
// an edit to any other row went unnoticed.) Arrays the caller never feeds to another layer (the logits) would stay here
// for good, so the table keeps the kMaxResident most recent slabs and releases the rest.
std::mutex g_lock;
std::map<const void*, DevSlab> g_resident;
uint64_t g_seq = 0;
constexpr size_t kMaxResident = 16;

// Hash of one ciphertext = its W = n + 1 words (a[0..n), b), two words per multiply. Rows are combined by a sum of
// position-mixed row hashes, so the rows can be hashed in any order and in parallel (131,072 x 351 words per CIFAR stage:
// a few milliseconds on the host threads, against ~100 ms for the array-of-structs copy the same stage performs anyway).
inline uint64_t mix64(uint64_t x) { x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32; return x; }
inline uint64_t row_hash(const int32_t* a, int n, int32_t b, size_t row) {
  uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t)row;
  int w = 0;
  for (; w + 1 < n; w += 2) { h = (h ^ ((uint64_t)(uint32_t)a[w] | ((uint64_t)(uint32_t)a[w + 1] << 32))) * 0x100000001b3ull; h ^= h >> 29; }
  if (w < n) { h = (h ^ (uint64_t)(uint32_t)a[w]) * 0x100000001b3ull; h ^= h >> 29; }
  h = (h ^ ((uint64_t)(uint32_t)b << 1 | 1ull)) * 0x100000001b3ull;
  return mix64(h);
}
template <class RowFn>   // RowFn(row) -> row hash; summed over [0, rows) on up to 8 threads
uint64_t hash_rows(size_t rows, RowFn fn) {
  const size_t hw = std::thread::hardware_concurrency();
  const size_t T = rows < 4096 ? 1 : std::min<size_t>(8, hw ? hw : 1);
  std::vector<uint64_t> part(T, 0);
  auto work = [&](size_t t) { uint64_t s = 0; for (size_t r = rows * t / T; r < rows * (t + 1) / T; ++r) s += fn(r); part[t] = s; };
  std::vector<std::thread> th;
  for (size_t t = 1; t < T; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  uint64_t h = 0x243f6a8885a308d3ull ^ (uint64_t)rows;
  for (uint64_t v : part) h += v;
  return h;
}
// ... of a packed [rows][W] host copy (what publish() downloads) and of the caller's LweSample array (what a stage receives)
uint64_t fingerprint(const int32_t* words, size_t rows, int W) {
  return hash_rows(rows, [&](size_t r) { const int32_t* row = words + r * (size_t)W; return row_hash(row, W - 1, row[W - 1], r); });
}
uint64_t fingerprint(const std::vector<const LweSample*>& samples, int n) {
  return hash_rows(samples.size(), [&](size_t r) { return row_hash(samples[r]->a, n, samples[r]->b, r); });
}

void remember(const void* host, DevSlab s) {
  std::vector<DevSlab> evicted;
  {
    std::lock_guard<std::mutex> g(g_lock);
    s.seq = ++g_seq;
    auto old = g_resident.find(host);
    if (old != g_resident.end()) { evicted.push_back(old->second); g_resident.erase(old); }
    g_resident[host] = s;
    while (g_resident.size() > kMaxResident) {
      auto oldest = g_resident.begin();
      for (auto it = g_resident.begin(); it != g_resident.end(); ++it) if (it->second.seq < oldest->second.seq) oldest = it;
      evicted.push_back(oldest->second);
      g_resident.erase(oldest);
    }
  }
  for (DevSlab& e : evicted) e.release();
}
bool is_resident(const void* host) {
  std::lock_guard<std::mutex> g(g_lock);
  return g_resident.count(host) != 0;
}
bool take(const void* host, DevSlab* s) {
  std::lock_guard<std::mutex> g(g_lock);
  auto it = g_resident.find(host);
  if (it == g_resident.end()) return false;
  *s = it->second;
  g_resident.erase(it);
  return true;
}

struct Geometry { int H, Wd, C, Ho, Wo, win_h, win_w, st_h, st_w, off_h, off_w; };

constexpr int32_t kUnit4096 = 1 << 20;   // modSwitchToTorus32(1, 4096)
constexpr int32_t kUnitRelu = 1 << 18;   // ReLU outputs: 1/16384, so that 1,024 of them sum inside a quarter turn
constexpr int32_t kQuarter = 1 << 30;
constexpr int kSlopeBitsInt = 8;         // lib/IntFunc.cpp:45

// The unit in which values travel (1/4096 for client pixels and sign bits, 1/16384 behind a ReLU) is a property of the data,
// so it rides in the object that describes the data: the tDimensions the driver threads through every prep() (nets/*/*/net.cpp:
// p_dim = layerK->prep(f, p_dim)). Its layout is the reference's (the drivers are compiled against the reference's own
// lib/Layer.h), so the unit is carried by a MARK on the `scale` field: Quantize::prep's ReLU branch stores the reference's
// value (2^s - 1 or 2^s, lib/IntFunc.cpp:835-844, lib/BinFunc.cpp:1019-1030) lowered by one part in 2^20. Every other
// value the chain ever holds there is an integer or a multiple of 1/2 (drivers start at 1, a sign stores 1/2 or 1, pools
// multiply by their window area), so "just below an integer" can only mean "behind a ReLU"; the mark survives the pools'
// multiplications, as a unit must, a sign overwrites it, and no consumer of `scale` changes its result (the only one is the
// ReLU's own ceil(log2(scale)), which a 2^-20 decrease cannot move). No table beside the chain, nothing keyed by an address:
// any number of networks in a process, stage-wise (BinFunc::* / IntFunc::*) or layer-wise.
float mark_relu_scale(float v) { return v * (1.0f - 9.5367431640625e-7f); }
int32_t unit_of(const tDimensions* d) {
  const float r = nearbyintf(d->scale), gap = r - d->scale;
  return (gap > 0.0f && gap <= d->scale * 3.8e-6f) ? kUnitRelu : kUnit4096;
}

struct LayerImpl {
  bool is_int;
  eConvType e_conv;
  ePoolType e_pool;
  eQuantType e_act;
  uint32_t depth;
  tNetParams np;
  TFheGateBootstrappingCloudKeySet* bk;
  bool prepared = false;
  // one-stage use through the BinFunc::* / IntFunc::* classes: a convolution, sum-pool or max-pool on its own reads no
  // bias record and adds none
  bool no_bias = false;
  // a max-pool on its own (BinFunc::MaxPooling) receives bare +-1/4096 bits: they are multiplied by this factor ahead
  // of the re-encoding bootstrap so that the bit stands clear of the mod-switch rounding noise (see the class below)
  int32_t prescale = 1;
  // geometry fixed by prep()
  Geometry conv{}, pool{};
  int in_count = 0, quant_count = 0, quant_depth = 0, out_count = 0;
  std::vector<uint8_t> sign, zero;
  std::vector<int32_t> bias;          // torus words (b of the trivial bias samples)
  std::vector<int32_t> pool_index;    // max-pool taps: [tap][out] row indices, -1 = outside
  int pool_taps = 0;
  // device copies, one set per device of the fleet
  struct DevWeights { uint8_t *sign = nullptr, *zero = nullptr; int32_t *bias = nullptr, *pool_index = nullptr, *pool_bias = nullptr, *lut = nullptr; };
  std::vector<DevWeights> dw;
  // fused max-pool (every window inside the image): the window's w sign bits are emitted as +-pool_mu =
  // +-1/(4w), summed, and ONE bootstrap of  sum + (w-1)/(4w)  is their OR (d_pool_bias holds that constant)
  bool pool_fused = false;
  int32_t pool_mu = 0;
  // Encoding: torus32 value of ONE integer step of this layer's input / output (1/4096 = 2^20 for client
  // pixels and sign bits, 1/16384 = 2^18 for ReLU outputs; DESIGN.md "ReLU semantics")
  int32_t unit_in = kUnit4096, unit_out = kUnit4096;
  int32_t final_rescale = 1;   // see prep_impl
  // IntFunc::Convolution's plaintext branch negates by one's complement (-x - 1): per output channel the
  // number of negative taps, folded into the bias of integer layers (see int_conv_plain)
  std::vector<int32_t> neg_taps;
  bool int_conv_plain = true;
  // ReLU (Quantize::relu_shift): slope per channel, shift amount, and the test polynomials [depth][N]
  std::vector<int32_t> slope, raw_bias;
  int shift_bits = 0, relu_shift = 0;

  rs_ctx** fleet(int* count) const { return redsec_fleet_of(bk, count); }
  int W() const { return bk->params->in_out_params->n + 1; }
};

void set_version(uint8_t v, tNetParams* net) {   // BinLayer.cpp:251-261: pre-v1 files had no strides
  if (v < 1) { net->conv.stride.h = 1; net->conv.stride.w = 1; net->pool.stride.h = 0; net->pool.stride.w = 0; }
}

LayerImpl* make_impl(bool is_int, eConvType ec, uint32_t dep, ePoolType ep, eQuantType eq, tNetParams* np, TFheGateBootstrappingCloudKeySet* bk) {
  assert(np != NULL && ec < NUM_CONVS && ep < NUM_POOLS && np->e_bias < NUM_BIASES);
  LayerImpl* L = new LayerImpl;
  L->is_int = is_int; L->e_conv = ec; L->e_pool = ep; L->e_act = eq; L->depth = dep; L->np = *np; L->bk = bk;
  set_version((uint8_t)L->np.version, &L->np);
  if (ec == E_FC || ec == E_FC_FINAL) { L->np.conv.window.h = 1; L->np.conv.window.w = 1; L->np.conv.same_pad = true; }
  if (L->np.pool.stride.h == 0) L->np.pool.stride.h = L->np.pool.window.h;   // SumPooling/MaxPooling ctor
  if (L->np.pool.stride.w == 0) L->np.pool.stride.w = L->np.pool.window.w;
  if (ep == E_MAXPOOL) assert(eq == E_ACTIVATION_SIGN);
  if (eq == E_ACTIVATION_RELU) {
    L->shift_bits = L->np.quant.shift_bits;
    if (L->shift_bits < 2 || L->shift_bits > 8) { fprintf(stderr, "redsec layers: ReLU shift_bits %d outside [2, 8]\n", L->shift_bits); abort(); }
  }
  const char* ic = getenv("REDSEC_INTCONV");   // "enc": the ENCRYPTED branch's constants (-1/4096 per zero/padding tap)
  L->int_conv_plain = !(ic && strcmp(ic, "enc") == 0);
  return L;
}

// -1 per negative tap of an integer layer's convolution on the plaintext branch's constants (a sum-pool
// behind the convolution adds one such term per window tap; windows inside the image, as in every shipped net)
int32_t neg_fold(const LayerImpl* L, size_t channel) {
  if (!(L->is_int && L->int_conv_plain) || L->neg_taps.empty()) return 0;
  const int mul = (L->e_pool == E_SUMPOOL && L->e_conv != E_NO_CONV) ? L->pool.win_h * L->pool.win_w : 1;
  return L->neg_taps[channel % L->neg_taps.size()] * mul;
}

uint8_t bits_for(uint32_t up_bound, uint8_t from) {
  uint8_t b = from;
  while ((up_bound >> b) > 0) ++b;
  return b;
}

// prep(): dimension propagation and weight loading in the reference's order
// conv (Convolution::prep) -> sumpool (SumPooling::prep) -> quantize (Quantize::prep) -> maxpool.
tDimensions* prep_impl(LayerImpl* L, FILE* fd, tDimensions* dim, tDimensions* in_dim, tDimensions* out_dim) {
  assert(!L->prepared && dim != NULL);
  *in_dim = *dim;
  L->in_count = dim->hw.h * dim->hw.w * (int)dim->in_dep;
  L->unit_in = unit_of(dim);
  if (L->e_conv != E_NO_CONV) {
    if (L->e_conv == E_FC || L->e_conv == E_FC_FINAL) { dim->in_dep *= dim->hw.h * dim->hw.w; dim->hw.h = 1; dim->hw.w = 1; }   // flatten
    const tConvParams& c = L->np.conv;
    assert(fd != NULL && c.stride.h != 0 && c.stride.w != 0);
    Geometry g{};
    g.H = dim->hw.h; g.Wd = dim->hw.w; g.C = (int)dim->in_dep; g.win_h = c.window.h; g.win_w = c.window.w; g.st_h = c.stride.h; g.st_w = c.stride.w;
    if (c.same_pad) {   // BinFunc.cpp:84-94 / IntFunc.cpp:86-95
      g.Ho = (g.H - 1) / g.st_h + 1; g.Wo = (g.Wd - 1) / g.st_w + 1;
      g.off_h = g.st_h == 1 ? (g.win_h - 1) / 2 : (g.Ho * g.st_h - g.H) / 2;
      g.off_w = g.st_w == 1 ? (g.win_w - 1) / 2 : (g.Wo * g.st_w - g.Wd) / 2;
    } else {
      g.off_h = g.off_w = 0;
      g.Ho = (g.H - 2 * ((g.win_h - 1) / 2)) / g.st_h; g.Wo = (g.Wd - 2 * ((g.win_w - 1) / 2)) / g.st_w;
    }
    L->conv = g;
    const size_t flen = (size_t)g.win_h * g.win_w * g.C * L->depth;
    L->sign.resize(flen); L->zero.resize(flen);
    BinOps::get_ternfilters(fd, L->sign.data(), L->zero.data(), (uint32_t)flen, c.tern_thresh, L->bk);
    L->neg_taps.assign(L->depth, 0);
    for (size_t i = 0; i < flen; ++i)
      if (!L->zero[i] && !L->sign[i]) ++L->neg_taps[i % L->depth];   // filter index ((fh*fw_+fw)*Cin+di)*Cout+od
    dim->up_bound *= (uint32_t)dim->filter_bits * g.win_w * g.win_h * g.C;
    dim->in_bits = bits_for(dim->up_bound, dim->in_bits);
    dim->hw.h = (int16_t)g.Ho; dim->hw.w = (int16_t)g.Wo; dim->in_dep = L->depth; dim->out_bits = SINGLE_BIT;
  }
  if (L->e_pool == E_SUMPOOL) {
    const tPoolParams& p = L->np.pool;
    Geometry g{};
    g.H = dim->hw.h; g.Wd = dim->hw.w; g.C = (int)dim->in_dep; g.win_h = p.window.h; g.win_w = p.window.w; g.st_h = p.stride.h; g.st_w = p.stride.w;
    if (p.same_pad) {   // BinFunc.cpp:631-639
      g.Ho = (g.H - 1) / g.st_h + 1; g.Wo = (g.Wd - 1) / g.st_w + 1;
      g.off_h = g.st_h == 1 ? (g.win_h - 1) / 2 : (g.Ho * g.st_h - g.H) / 2;
      g.off_w = g.st_w == 1 ? (g.win_w - 1) / 2 : (g.Wo * g.st_w - g.Wd) / 2;
    } else {
      g.off_h = g.off_w = 0;
      g.Ho = (g.H - g.win_h / 2 - 1) / g.st_h + 1; g.Wo = (g.Wd - g.win_w / 2 - 1) / g.st_w + 1;
    }
    L->pool = g;
    dim->up_bound *= (uint32_t)(g.win_w * g.win_h);
    dim->in_bits = bits_for(dim->up_bound, dim->in_bits);
    dim->scale *= (float)(g.win_w * g.win_h);
    dim->hw.h = (int16_t)g.Ho; dim->hw.w = (int16_t)g.Wo; dim->out_bits = SINGLE_BIT;
  }
  // Quantize::prep: bias = int32[in_dep] -> trivial samples of bias/4096
  L->quant_depth = (int)dim->in_dep;
  L->quant_count = dim->hw.h * dim->hw.w * (int)dim->in_dep;
  if (L->no_bias) {
    L->raw_bias.assign((size_t)L->quant_depth, 0);
    L->bias.assign((size_t)L->quant_depth, 0);
  } else {
    assert(fd != NULL);
    uint8_t tag = 0;
    size_t got = fread(&tag, 1, 1, fd); (void)got;
    assert(tag == 3 || tag == 4);
    std::vector<int32_t> v((size_t)L->quant_depth);
    got = fread(v.data(), sizeof(int32_t), v.size(), fd);
    L->raw_bias = v;
    L->bias.resize(v.size());
    // trivial samples of bias * unit (get_intfilters: modSwitchToTorus32(b, 4096), i.e. unit 2^20, for 1/4096 inputs);
    // integer layers running on the plaintext branch's constants also fold its -1 per negative tap
    for (size_t i = 0; i < v.size(); ++i) L->bias[i] = (int32_t)((uint32_t)(v[i] - neg_fold(L, i)) * (uint32_t)L->unit_in);
    // slope: read when the layer is IntLayer(RELU, E_BNORM) (IntLayer::prep allocates p_slope only then,
    // IntLayer.cpp:96-100) or BinLayer(RELU) (BinFunc::Quantize::prep reads it whenever p_slope != NULL)
    if (L->e_act == E_ACTIVATION_RELU && (!L->is_int || L->np.e_bias == E_BNORM)) {
      got = fread(&tag, 1, 1, fd);
      assert(tag == 3 || tag == 4);
      L->slope.resize((size_t)L->quant_depth);
      got = fread(L->slope.data(), sizeof(int32_t), L->slope.size(), fd);
    }
  }
  L->unit_out = L->unit_in;
  if (L->e_act == E_ACTIVATION_SIGN) { dim->in_bits = 1; dim->up_bound = 1; dim->scale = L->is_int ? 1.0f : 0.5f; L->unit_out = kUnit4096; }
  if (L->e_act == E_ACTIVATION_RELU) {
    if (L->slope.empty()) { fprintf(stderr, "redsec layers: ReLU layer without a slope record (needs E_BNORM)\n"); abort(); }
    if (L->is_int) {
      int sc_b = 0;                                       // IntFunc::Quantize::prep, lib/IntFunc.cpp:812-815
      while ((float)(1 << sc_b) < dim->scale) ++sc_b;
      L->relu_shift = kSlopeBitsInt + sc_b - L->shift_bits;
      dim->in_bits = (uint8_t)L->shift_bits;             // :835-844
      dim->scale = mark_relu_scale((float)((1 << L->shift_bits) - 1));
      dim->up_bound = 1u << (L->shift_bits - 1);
    } else {
      L->relu_shift = L->shift_bits + 1;                  // BinFunc::Quantize::relu_shift shifts by shift_bits + 1 (lib/BinFunc.cpp:1154)
      dim->in_bits = (uint8_t)(L->shift_bits + 1);        // lib/BinFunc.cpp:1019-1030
      dim->up_bound = 1u << L->shift_bits;
      dim->scale = mark_relu_scale((float)dim->up_bound);
    }
    if (L->relu_shift < 0 || L->relu_shift > 40) { fprintf(stderr, "redsec layers: ReLU shift %d out of range\n", L->relu_shift); abort(); }
    L->unit_out = kUnitRelu;
  }
  dim->out_bits = SINGLE_BIT;
  L->out_count = L->quant_count;
  if (L->e_pool == E_MAXPOOL && L->e_act == E_ACTIVATION_SIGN && !(L->is_int && L->e_conv == E_FC_FINAL)) {
    const tPoolParams& p = L->np.pool;
    Geometry g{};
    g.H = dim->hw.h; g.Wd = dim->hw.w; g.C = (int)dim->in_dep; g.win_h = p.window.h; g.win_w = p.window.w; g.st_h = p.stride.h; g.st_w = p.stride.w;
    if (p.same_pad) { g.Ho = (g.H - 1) / g.st_h + 1; g.Wo = (g.Wd - 1) / g.st_w + 1; }   // BinFunc.cpp:852-861
    else { g.Ho = g.H / g.win_h; g.Wo = g.Wd / g.win_w; }
    g.off_h = g.off_w = 0;   // MaxPooling::prep never sets offset_window (BinFunc.cpp:836-872); valid pooling uses 0
    L->pool = g;
    L->pool_taps = g.win_h * g.win_w;
    L->out_count = g.Ho * g.Wo * g.C;
    L->pool_index.assign((size_t)L->pool_taps * L->out_count, -1);
    for (int oh = 0; oh < g.Ho; ++oh)
      for (int ow = 0; ow < g.Wo; ++ow)
        for (int c = 0; c < g.C; ++c) {
          const int o = (oh * g.Wo + ow) * g.C + c;
          for (int fh = 0; fh < g.win_h; ++fh)
            for (int fw = 0; fw < g.win_w; ++fw) {
              const int ih = oh * g.st_h + fh, iw = ow * g.st_w + fw;
              if (ih < g.H && iw < g.Wd) L->pool_index[(size_t)(fh * g.win_w + fw) * L->out_count + o] = (ih * g.Wd + iw) * g.C + c;
            }
        }
    dim->hw.h = (int16_t)g.Ho; dim->hw.w = (int16_t)g.Wo;
  }
  // A final layer (no activation) behind a ReLU hands its logits back in the unit they were summed in, 1/16384. The reference's
  // client decodes with message space 4096 (client/decrypt_image.cpp:52-58) and therefore reads round(logit / 4): two bits of
  // resolution less, the same class except for near ties. REDSEC_RESCALE_LOGITS=1 multiplies the logits by 4 instead (exact,
  // word-wise), which gives that client the integers themselves -- but wraps around its +-2048 range for logits near +-2000,
  // which relu1024x3 reaches on the bundled images (measured: tests/test_gpu_relu.py history), so it is NOT the default.
  const char* rescale = getenv("REDSEC_RESCALE_LOGITS");
  L->final_rescale = (L->e_act == E_ACTIVATION_NONE && L->unit_out != kUnit4096 && rescale && *rescale && strcmp(rescale, "0") != 0) ? kUnit4096 / L->unit_out : 1;
  *out_dim = *dim;
  L->prepared = true;
  return dim;
}

// Test polynomials of a ReLU layer (same table as redsec_amd/nets.py::relu_luts): index t stands for
// pre = (t - N/2) * (2^32 / 2N / unit_in), the centre of the phase bucket the mod-switch rounds to after the
// quarter-turn shift; entry = clamp((slope * pre + bias) >> relu_shift, 0, 2^shift_bits - 1) * unit_out
// (IntOps::shift + IntOps::relu, lib/IntOps.cpp).
std::vector<int32_t> relu_luts(const LayerImpl* L) {
  const int kRingN = L->bk->params->tgsw_params->tlwe_params->N;
  const int64_t kLutStep = (1ll << 31) / kRingN;   // one mod-switched phase step, 2^32 / 2N (2^21 for N = 1024)
  const int64_t top = (1 << L->shift_bits) - 1;
  std::vector<int32_t> lut((size_t)L->quant_depth * kRingN);
  for (int m = 0; m < L->quant_depth; ++m)
    for (int t = 0; t < kRingN; ++t) {
      // bucket centre in input steps; rings above N = 1024 resolve finer than one step: nearest integer
      const int64_t num = (int64_t)(t - kRingN / 2) * kLutStep;
      const int64_t pre = (num + L->unit_in / 2) >> (L->unit_in == kUnit4096 ? 20 : 18);
      const int64_t x = (int64_t)L->slope[m] * pre + (int64_t)L->raw_bias[m];
      const int64_t y = x < 0 ? 0 : ((x >> L->relu_shift) > top ? top : (x >> L->relu_shift));
      lut[(size_t)m * kRingN + t] = (int32_t)((uint32_t)y * (uint32_t)L->unit_out);
    }
  return lut;
}

template <class T>
T* to_device(rs_ctx* c, const std::vector<T>& host) {
  T* p = (T*)pool_alloc(c, host.size() * sizeof(T));
  RS_CHECK(rs_copy_to_dev(c, p, host.data(), host.size() * sizeof(T)));
  return p;
}

void upload_weights(LayerImpl* L) {
  if (!L->dw.empty()) return;
  int D = 0;
  rs_ctx** fleet = L->fleet(&D);
  std::vector<int32_t> lut, pool_bias;
  if (L->e_act == E_ACTIVATION_RELU) {
    // the bias lives inside the test polynomial; what joins the linear stage is the quarter turn that moves
    // pre in [-N/2, N/2) steps onto [0, 1/2), and the -1 per negative tap of the plaintext branch
    for (size_t i = 0; i < L->bias.size(); ++i) L->bias[i] = (int32_t)((uint32_t)kQuarter - (uint32_t)neg_fold(L, i) * (uint32_t)L->unit_in);
    lut = relu_luts(L);
  }
  if (!L->pool_index.empty()) {
    bool full = L->pool_taps >= 2;
    for (int32_t v : L->pool_index) full = full && v >= 0;
    const char* mode = getenv("REDSEC_MAXPOOL");          // "chain" selects the OR chain of BinOps::max calls
    L->pool_fused = full && !(mode && strcmp(mode, "chain") == 0);
    if (L->pool_fused) {
      L->pool_mu = (int32_t)((1ull << 32) / (4ull * (unsigned)L->pool_taps));
      pool_bias.push_back((int32_t)((uint32_t)(L->pool_taps - 1) * (uint32_t)L->pool_mu));
    }
  }
  L->dw.resize((size_t)D);
  for (int d = 0; d < D; ++d) {
    LayerImpl::DevWeights& w = L->dw[d];
    if (!L->sign.empty()) { w.sign = to_device(fleet[d], L->sign); w.zero = to_device(fleet[d], L->zero); }
    w.bias = to_device(fleet[d], L->bias);
    if (!lut.empty()) w.lut = to_device(fleet[d], lut);
    if (!L->pool_index.empty()) w.pool_index = to_device(fleet[d], L->pool_index);
    if (!pool_bias.empty()) w.pool_bias = to_device(fleet[d], pool_bias);
  }
}

int32_t* dev_rows(rs_ctx* c, size_t rows, int W) { return (int32_t*)pool_alloc(c, rows * (size_t)W * 4); }

// contiguous, balanced slice of `total` rows for device d of D (sizes differ by at most one; the same split as
// redsec_amd/sharding.py::shard_range)
void shard_range(size_t total, int d, int D, size_t* lo, size_t* hi) {
  const size_t base = total / (size_t)D, rem = total % (size_t)D;
  *lo = (size_t)d * base + ((size_t)d < rem ? (size_t)d : rem);
  *hi = *lo + base + ((size_t)d < rem ? 1 : 0);
}

// One host thread per device beyond the first, as the reference drives its GPUs (lib/GPU/BinFunc_gpu.cu:118-137: one
// thread per enc_segs[g]): the per-device work of a stage (allocation, per-call host preparation, launches) is issued
// concurrently, device 0's by the calling thread. Threads are started once per process and parked between stages.
class DeviceWorkers {
  struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool busy = false, quit = false;
  };
  std::vector<std::unique_ptr<Worker>> w_;
  std::mutex grow_;
  static void loop(Worker* w) {
    std::unique_lock<std::mutex> lk(w->mu);
    for (;;) {
      w->cv.wait(lk, [&] { return w->busy || w->quit; });
      if (w->quit) return;
      std::function<void()> job = std::move(w->job);
      lk.unlock();
      job();
      lk.lock();
      w->busy = false;
      w->cv.notify_all();
    }
  }
 public:
  ~DeviceWorkers() {
    for (auto& w : w_) { { std::lock_guard<std::mutex> g(w->mu); w->quit = true; } w->cv.notify_all(); w->th.join(); }
  }
  // f(d) for d in [0, D): d >= 1 on worker d - 1, d = 0 here; returns when all are done (a host-side join, no device wait)
  template <class F>
  void run(int D, F f) {
    if (D <= 1) { if (D == 1) f(0); return; }
    std::lock_guard<std::mutex> serial(grow_);   // one stage at a time (layers of one process run one after the other anyway)
    while ((int)w_.size() < D - 1) { w_.emplace_back(new Worker); Worker* w = w_.back().get(); w->th = std::thread(loop, w); }
    for (int d = 1; d < D; ++d) {
      Worker* w = w_[(size_t)d - 1].get();
      { std::lock_guard<std::mutex> g(w->mu); w->job = [&f, d] { f(d); }; w->busy = true; }
      w->cv.notify_all();
    }
    f(0);
    for (int d = 1; d < D; ++d) {
      Worker* w = w_[(size_t)d - 1].get();
      std::unique_lock<std::mutex> lk(w->mu);
      w->cv.wait(lk, [&] { return !w->busy; });
    }
  }
};
DeviceWorkers g_workers;

// The devices a layer runs on, and the two ways a stage uses them. Nothing here waits for a device: stages follow one
// another in each device's default-stream order, the slice exchange is ordered by events (rs_allgather_rows), and the only
// synchronisation of a layer chain is the download of what the caller asked for.
struct Fleet {
  rs_ctx** c;
  int D;
  int W;
  // the same launch on every device, each on its own full replica (linear stages: every device needs the whole
  // input of the next bootstrap stage, and recomputing < 2 % of a layer beats exchanging it)
  template <class F>
  std::vector<int32_t*> replicated(size_t rows, F launch) const {
    std::vector<int32_t*> y((size_t)D);
    g_workers.run(D, [&](int d) { y[d] = dev_rows(c[d], rows, W); launch(d, c[d], y[d]); });
    return y;
  }
  // a stage of independent ciphertexts (every bootstrap: lib/BinFunc.cpp:1056-1071 has no cross-iteration dependence):
  // device d computes its contiguous slice into its own replica, then every device pulls the other slices as they become
  // ready, so that the next linear stage finds the whole vector everywhere. One device: no exchange at all.
  template <class F>
  std::vector<int32_t*> sharded(size_t rows, F launch) const {
    std::vector<int32_t*> y((size_t)D);
    g_workers.run(D, [&](int d) {
      y[d] = dev_rows(c[d], rows, W);
      size_t lo, hi;
      shard_range(rows, d, D, &lo, &hi);
      if (hi > lo) launch(d, c[d], y[d] + lo * (size_t)W, lo, hi - lo);
    });
    if (D > 1) RS_CHECK(rs_allgather_rows(c, D, y.data(), rows, (size_t)W));
    return y;
  }
  void release(std::vector<int32_t*>& v) const {
    for (int d = 0; d < D; ++d) pool_free(c[d], v[d]);
    v.clear();
  }
};

// One layer on the device(s): x is the input slab (consumed), returns the output slab.
DevSlab run_layer(LayerImpl* L, DevSlab x) {
  Fleet f{nullptr, 0, L->W()};
  f.c = L->fleet(&f.D);
  const int32_t mu4096 = modSwitchToTorus32(1, 4096), mu8 = modSwitchToTorus32(1, 8);
  upload_weights(L);
  const std::vector<LayerImpl::DevWeights>& dw = L->dw;
  const bool pool_sum = L->e_pool == E_SUMPOOL;
  auto replace = [&](std::vector<int32_t*> y, size_t rows) { f.release(x.ptr); x.ptr = std::move(y); x.rows = rows; };
  // IntFunc constants: ternary-zero and padding taps contribute the trivial -1/4096 (IntFunc.cpp:268,277)
  // (REDSEC_INTCONV=enc; the default follows the plaintext branch, whose constant is folded into the bias:
  // the two disagree and only the plaintext one matches the trained biases, DESIGN.md "ReLU semantics")
  const int32_t tap_const = (L->is_int && !L->int_conv_plain) ? -mu4096 : 0;
  if (L->e_conv != E_NO_CONV) {
    const Geometry& g = L->conv;
    const size_t rows = (size_t)g.Ho * g.Wo * L->depth;
    replace(f.replicated(rows, [&](int d, rs_ctx* c, int32_t* y) {
      const int32_t* bias = (pool_sum || L->no_bias) ? nullptr : dw[d].bias;   // bias joins at the last linear op before the activation
      if (g.win_h == 1 && g.win_w == 1 && g.H == 1 && g.Wd == 1) {
        RS_CHECK(rs_linear_fc_dev(c, y, x.ptr[d], dw[d].sign, dw[d].zero, g.C, (int32_t)L->depth, tap_const, bias, L->quant_depth, nullptr));
      } else {
        rs_conv_shape s{g.H, g.Wd, g.C, (int32_t)L->depth, g.win_h, g.win_w, g.st_h, g.st_w, g.off_h, g.off_w, g.Ho, g.Wo};
        RS_CHECK(rs_conv_ternary_dev(c, y, x.ptr[d], dw[d].sign, dw[d].zero, &s, tap_const, tap_const, bias, L->quant_depth, nullptr));
      }
    }), rows);
  }
  if (pool_sum || (L->e_conv == E_NO_CONV && !L->no_bias)) {
    // SumPooling::execute; a layer with neither conv nor pooling still needs its bias: 1x1 window
    Geometry g = L->pool;
    if (!pool_sum) { g = Geometry{1, (int)x.rows / L->quant_depth, L->quant_depth, 1, (int)x.rows / L->quant_depth, 1, 1, 1, 1, 0, 0}; }
    const size_t rows = (size_t)g.Ho * g.Wo * g.C;
    replace(f.replicated(rows, [&](int d, rs_ctx* c, int32_t* y) {
      rs_pool_shape s{g.H, g.Wd, g.C, g.win_h, g.win_w, g.st_h, g.st_w, g.off_h, g.off_w, g.Ho, g.Wo};
      RS_CHECK(rs_sumpool_dev(c, y, x.ptr[d], &s, L->no_bias ? nullptr : dw[d].bias, L->quant_depth, nullptr));
    }), rows);
  }
  assert((int)x.rows == L->quant_count);
  if (L->e_act == E_ACTIVATION_RELU) {
    // Quantize::relu_shift, corrected: ONE programmable bootstrap per neuron evaluates the plaintext
    // branch's staircase (lib/IntFunc.cpp:964-967) on the mod-switched phase of the pre-activation
    replace(f.sharded(x.rows, [&](int d, rs_ctx* c, int32_t* y, size_t lo, size_t count) {
      RS_CHECK(rs_bootstrap_lut_dev(c, y, x.ptr[d] + lo * (size_t)f.W, dw[d].lut, (size_t)L->quant_depth, lo, count, nullptr));
    }), x.rows);
  }
  if (L->prescale != 1) {
    replace(f.replicated(x.rows, [&](int d, rs_ctx* c, int32_t* y) {
      RS_CHECK(rs_lincomb_dev(c, y, x.ptr[d], L->prescale, nullptr, 0, 0, x.rows, nullptr));
    }), x.rows);
  }
  if (L->e_act == E_ACTIVATION_SIGN) {
    const bool maxpool = !L->pool_index.empty();
    // Quantize::execute: one sign bootstrap per neuron (BinOps_enc.cpp:182-186). Ahead of a max-pool
    // the bits are emitted as +-1/8 so that the OR gates see the encoding they assume.
    const int32_t mu = !maxpool ? mu4096 : (L->pool_fused ? L->pool_mu : mu8);
    replace(f.sharded(x.rows, [&](int d, rs_ctx* c, int32_t* y, size_t lo, size_t count) {
      RS_CHECK(rs_bootstrap_dev(c, y, x.ptr[d] + lo * (size_t)f.W, mu, count, nullptr));
    }), x.rows);
    if (maxpool && L->pool_fused) {
      // OR over a full window of w bits in ONE bootstrap: with the bits at +-1/(4w) the windowed LWE sum
      // plus (w-1)/(4w) is >= +1/(4w) unless every bit is false (then -1/(4w)) and stays below 1/2 - 1/(4w):
      // its sign bootstrap IS the OR, emitted at +-1/4096 for the next linear stage (SURVEY.md hard part 6).
      const Geometry& g = L->pool;
      const size_t out = (size_t)L->out_count;
      std::vector<int32_t*> z = f.replicated(out, [&](int d, rs_ctx* c, int32_t* y) {
        rs_pool_shape s{g.H, g.Wd, g.C, g.win_h, g.win_w, g.st_h, g.st_w, 0, 0, g.Ho, g.Wo};
        RS_CHECK(rs_sumpool_dev(c, y, x.ptr[d], &s, dw[d].pool_bias, 1, nullptr));
      });
      std::vector<int32_t*> o = f.sharded(out, [&](int d, rs_ctx* c, int32_t* y, size_t lo, size_t count) {
        RS_CHECK(rs_bootstrap_dev(c, y, z[d] + lo * (size_t)f.W, mu4096, count, nullptr));
      });
      f.release(z);
      replace(std::move(o), out);
    } else if (maxpool) {
      // MaxPooling::execute: OR over the window in (fh, fw) order; the first tap is copied (the
      // reference ORs into an uninitialised accumulator, BinFunc.cpp:891,917), the last OR re-encodes
      // to +-1/4096 for the next linear stage.
      const size_t out = (size_t)L->out_count;
      auto gather = [&](int t) {
        return f.replicated(out, [&](int d, rs_ctx* c, int32_t* y) {
          RS_CHECK(rs_gather_rows_dev(c, y, x.ptr[d], dw[d].pool_index + (size_t)t * out, out, nullptr));
        });
      };
      std::vector<int32_t*> accv = gather(0);
      for (int t = 1; t < L->pool_taps; ++t) {
        std::vector<int32_t*> tap = gather(t);
        const bool last = t == L->pool_taps - 1;
        // a tap outside the image gathers the zero sample: OR(acc, 0-phase) keeps acc's sign only if
        // windows are full, which holds for every shipped net (even feature maps, 2x2 windows)
        std::vector<int32_t*> next = f.sharded(out, [&](int d, rs_ctx* c, int32_t* y, size_t lo, size_t count) {
          RS_CHECK(rs_gate_mu_dev(c, RS_OR, y, accv[d] + lo * (size_t)f.W, tap[d] + lo * (size_t)f.W, last ? mu4096 : mu8, count, nullptr));
        });
        f.release(tap);
        f.release(accv);
        accv = std::move(next);
      }
      if (L->pool_taps == 1) {
        std::vector<int32_t*> next = f.sharded(out, [&](int d, rs_ctx* c, int32_t* y, size_t lo, size_t count) {
          RS_CHECK(rs_bootstrap_dev(c, y, accv[d] + lo * (size_t)f.W, mu4096, count, nullptr));
        });
        f.release(accv);
        accv = std::move(next);
      }
      replace(std::move(accv), out);
    }
  }
  if (L->final_rescale != 1) {
    // REDSEC_RESCALE_LOGITS: logits summed in the ReLU unit (1/16384) leave in the client's (1/4096): an exact word-wise multiple
    replace(f.replicated(x.rows, [&](int d, rs_ctx* c, int32_t* y) {
      RS_CHECK(rs_lincomb_dev(c, y, x.ptr[d], L->final_rescale, nullptr, 0, 0, x.rows, nullptr));
    }), x.rows);
  }
  x.ctx.assign(f.c, f.c + f.D);
  return x;
}

// host array of LweSample -> device slab (or the resident copy a previous layer left)
DevSlab stage_input(LayerImpl* L, const void* key, const std::vector<const LweSample*>& samples) {
  DevSlab s;
  int D = 0;
  rs_ctx** fleet = L->fleet(&D);
  const int W = L->W(), n = W - 1;
  const bool had = take(key, &s);
  if (had && s.lazy && s.rows == samples.size() && (int)s.ctx.size() == D && std::equal(s.ctx.begin(), s.ctx.end(), fleet)) {
    s.lazy = false;
    return s;   // the host arrays are placeholders (REDSEC_LAZY_HOST): the device copy is the only one there is
  }
  if (had && s.rows == samples.size() && (int)s.ctx.size() == D && std::equal(s.ctx.begin(), s.ctx.end(), fleet)) {
    // the same content hash as at publication, over every word the caller now holds
    if (fingerprint(samples, n) == s.tag) return s;
  }
  if (had) s.release();   // stale or foreign: release it and upload what the host holds
  s = DevSlab{};
  std::vector<int32_t> host(samples.size() * (size_t)W);
  for (size_t i = 0; i < samples.size(); ++i) redsec_pack(&host[i * W], samples[i], n);
  s.rows = samples.size();
  for (int d = 0; d < D; ++d) {
    s.ctx.push_back(fleet[d]);
    s.ptr.push_back(dev_rows(fleet[d], samples.size(), W));
    RS_CHECK(rs_copy_to_dev(fleet[d], s.ptr[d], host.data(), host.size() * 4));
  }
  return s;
}

std::vector<int32_t> download(const DevSlab& s, int W) {
  std::vector<int32_t> host(s.rows * (size_t)W);
  // nothing to certify here: in FFT mode every bootstrapped call is followed on the device by its gated exact
  // recomputation (include/redsec_hip.h), so what comes down is exact by construction
  RS_CHECK(rs_copy_to_host(s.ctx[0], host.data(), s.ptr[0], host.size() * 4));
  return host;
}

// Output as the reference returns it: tBit* for sign layers, tMultiBit* (ctxt[0] used) otherwise.
// REDSEC_LAZY_HOST=1 (opt-in): a layer WITH an activation -- every layer but a network's last -- hands back host arrays of the
// right shape that are allocated but never filled; the resident device slab is the data. Downloading 131,072 x 351 words and
// rebuilding the array-of-structs costs ~100 ms per CIFAR layer, and an unmodified driver only ever passes those arrays to
// the next layer. A layer without activation (the logits) is always downloaded. Code that wants to READ an intermediate
// array calls redsec_materialize(array) first.
static bool lazy_host() { static const bool on = [] { const char* v = getenv("REDSEC_LAZY_HOST"); return v && *v && strcmp(v, "0") != 0; }(); return on; }
void* publish(LayerImpl* L, DevSlab out) {
  const int W = L->W(), n = W - 1;
  if (lazy_host() && L->e_act != E_ACTIVATION_NONE) {
    out.lazy = true;
    out.bits = L->e_act == E_ACTIVATION_SIGN;
    out.row_words = W;
    void* ret = L->e_act == E_ACTIVATION_SIGN ? (void*)bit_calloc((uint32_t)out.rows, L->bk) : (void*)mbit_calloc((uint32_t)out.rows, 1, L->bk);
    remember(ret, out);
    return ret;
  }
  std::vector<int32_t> host = download(out, W);
  out.tag = fingerprint(host.data(), out.rows, W);
  void* ret = nullptr;
  if (L->e_act == E_ACTIVATION_SIGN) {
    tBit* bits = bit_calloc((uint32_t)out.rows, L->bk);
    for (size_t i = 0; i < out.rows; ++i) redsec_unpack(&bits[i], &host[i * W], n);
    ret = bits;
  } else {
    tMultiBit* mb = mbit_calloc((uint32_t)out.rows, 1, L->bk);
    for (size_t i = 0; i < out.rows; ++i) redsec_unpack(&mb[i].ctxt[0], &host[i * W], n);
    ret = mb;
  }
  remember(ret, out);
  return ret;
}

}  // namespace redsec_host

// Fills a host array that REDSEC_LAZY_HOST left empty (no-op for any other pointer): the slab comes down and is unpacked
// into the caller's array; the device copy stays resident for the next layer.
void redsec_materialize(void* host_array) {
  using namespace redsec_host;
  DevSlab s;
  {
    std::lock_guard<std::mutex> g(g_lock);
    auto it = g_resident.find(host_array);
    if (it == g_resident.end() || !it->second.lazy) return;
    s = it->second;
  }
  const size_t rows = s.rows;
  const int W = s.row_words, n = W - 1;
  std::vector<int32_t> host(rows * (size_t)W);
  RS_CHECK(rs_copy_to_host(s.ctx[0], host.data(), s.ptr[0], host.size() * 4));
  if (s.bits) { tBit* b = (tBit*)host_array; for (size_t i = 0; i < rows; ++i) redsec_unpack(&b[i], &host[i * W], n); }
  else { tMultiBit* m = (tMultiBit*)host_array; for (size_t i = 0; i < rows; ++i) redsec_unpack(&m[i].ctxt[0], &host[i * W], n); }
  std::lock_guard<std::mutex> g(g_lock);
  auto it = g_resident.find(host_array);
  if (it != g_resident.end() && it->second.seq == s.seq) { it->second.lazy = false; it->second.tag = fingerprint(host.data(), rows, W); }
}
// Before a context is destroyed (tfhe_shim.cpp, delete_gate_bootstrapping_*_keyset): resident slabs and cached blocks of it go.
void redsec_pool_release(rs_ctx* c) {
  using namespace redsec_host;
  std::vector<DevSlab> gone;
  {
    std::lock_guard<std::mutex> g(g_lock);
    for (auto it = g_resident.begin(); it != g_resident.end();) {
      if (std::find(it->second.ctx.begin(), it->second.ctx.end(), c) != it->second.ctx.end()) { gone.push_back(it->second); it = g_resident.erase(it); }
      else ++it;
    }
  }
  for (DevSlab& sl : gone) sl.release();
  std::unique_ptr<BufPool> P;
  {
    std::lock_guard<std::mutex> g(g_pool_lock);
    auto it = g_pools.find(c);
    if (it == g_pools.end()) return;
    P = std::move(it->second);
    g_pools.erase(it);
  }
  for (auto& kv : P->idle) (void)rs_dev_free(c, kv.second);
}

using redsec_host::LayerImpl;

namespace redsec_host {

// execute() of a layer or of a single stage: stage the input (or find it resident), free it as the reference's callee
// does (BinFunc.cpp:327, IntFunc.cpp:698), run, hand back fresh host arrays with the device copy remembered
// REDSEC_TRACE=1: per-layer host timing on stderr (input staging incl. freeing the caller's arrays, device stages, download +
// host arrays), for finding out where a driver's wall time goes beside the kernels
static bool trace_on() { static const bool on = getenv("REDSEC_TRACE") != nullptr; return on; }
static double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
// device-context creation and key upload happen inside the first call that touches the GPU (the first layer's staging): reported
// on a line of their own and taken out of that layer's "stage"
static double trace_setup() {
  const double s = redsec_take_setup_seconds();
  if (s > 0.0 && trace_on()) fprintf(stderr, "redsec trace: device context + key upload and transform (once per process): %.1f ms\n", 1e3 * s);
  return s;
}
static void trace(const LayerImpl* L, double t0, double t1, double t2, double t3, size_t rows_out) {
  fprintf(stderr, "redsec trace: layer in %d -> out %zu ciphertexts: stage %.1f ms, device %.1f ms, publish %.1f ms\n", L->in_count, rows_out,
          1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2));
}
void* execute_bits(LayerImpl* impl, tBit* p_in) {
  assert(impl->prepared);
  const double t0 = now_s();
  std::vector<const LweSample*> in((size_t)impl->in_count);
  for (size_t i = 0; i < in.size(); ++i) in[i] = &p_in[i];
  DevSlab x = stage_input(impl, p_in, in);
  bit_free((uint32_t)impl->in_count, p_in);
  const double setup = trace_setup();
  const double t1 = now_s();
  DevSlab y = run_layer(impl, x);
  const double t2 = now_s();
  const size_t rows = y.rows;
  void* ret = publish(impl, y);
  if (trace_on()) trace(impl, t0 + setup, t1, t2, now_s(), rows);
  return ret;
}
void* execute_mbits(LayerImpl* impl, tMultiBit* p_in) {
  assert(impl->prepared);
  std::vector<const LweSample*> in((size_t)impl->in_count);
  for (size_t i = 0; i < in.size(); ++i) in[i] = &p_in[i].ctxt[0];
  const bool ours = is_resident(p_in);     // produced by one of these layers (calloc) or by the driver (new[])
  const double t0 = now_s();
  DevSlab x = stage_input(impl, p_in, in);
  for (int i = 0; i < impl->in_count; ++i) delete_gate_bootstrapping_ciphertext_array((int32_t)p_in[i].size, p_in[i].ctxt);
  if (ours) free(p_in); else delete[] p_in;
  const double setup = trace_setup();
  const double t1 = now_s();
  DevSlab y = run_layer(impl, x);
  const double t2 = now_s();
  const size_t rows = y.rows;
  void* ret = publish(impl, y);
  if (trace_on()) trace(impl, t0 + setup, t1, t2, now_s(), rows);
  return ret;
}

// a LayerImpl that is ONE stage (the BinFunc::* / IntFunc::* classes)
LayerImpl* make_stage(bool is_int, eConvType ec, uint32_t dep, ePoolType ep, eQuantType eq, bool no_bias) {
  LayerImpl* L = new LayerImpl;
  L->is_int = is_int; L->e_conv = ec; L->e_pool = ep; L->e_act = eq; L->depth = dep; L->bk = nullptr; L->no_bias = no_bias;
  memset(&L->np, 0, sizeof L->np);
  L->np.e_bias = E_BNORM;                                // a ReLU stage reads its slope record
  L->np.conv.stride.h = L->np.conv.stride.w = 1;
  const char* ic = getenv("REDSEC_INTCONV");
  L->int_conv_plain = !(ic && strcmp(ic, "enc") == 0);
  return L;
}
tDimensions* prep_stage(LayerImpl* L, FILE* fd, tDimensions* dim, TFheGateBootstrappingCloudKeySet* bk) {
  tDimensions in_dim, out_dim;
  L->bk = bk;
  if (L->np.pool.stride.h == 0) L->np.pool.stride.h = L->np.pool.window.h;   // SumPooling/MaxPooling ctor
  if (L->np.pool.stride.w == 0) L->np.pool.stride.w = L->np.pool.window.w;
  return prep_impl(L, fd, dim, &in_dim, &out_dim);
}
// Quantize::prep hands the records it read to the caller's arrays, as the reference does (get_intfilters into p_bias:
// trivial samples of bias/4096; get_intfilters_ptxt into p_slope)
void export_bias(const LayerImpl* L, tMultiBit* p_bias, uint32_t* p_slope) {
  const LweParams* p = L->bk->params->in_out_params;
  for (int i = 0; p_bias && i < L->quant_depth; ++i) {
    p_bias[i].size = 1;
    p_bias[i].ctxt = new_LweSample(p);
    lweNoiselessTrivial(&p_bias[i].ctxt[0], modSwitchToTorus32(L->raw_bias[(size_t)i], 4096), p);
  }
  for (size_t i = 0; p_slope && i < L->slope.size(); ++i) p_slope[i] = (uint32_t)L->slope[i];
}

}  // namespace redsec_host

// ---- BinFunc / IntFunc: the reference's per-stage classes (lib/BinFunc.h:37-173, lib/IntFunc.h:27-140), each ONE batched
// stage of the same engine the layers use. Chained through their host arrays they stay device-resident like the layers. ----
#include "lib/BinFunc.h"
#include "lib/IntFunc.h"
using redsec_host::make_stage;
using redsec_host::prep_stage;

BinFunc::Convolution::Convolution(uint32_t out_depth, tConvParams* in_params) {
  impl = make_stage(false, E_CONV, out_depth, E_NO_POOL, E_ACTIVATION_NONE, true);
  impl->np.conv = *in_params;
}
tDimensions* BinFunc::Convolution::prep(FILE* fd_filt, tDimensions* ret_dim, TFheGateBootstrappingCloudKeySet* in_bk) { return prep_stage(impl, fd_filt, ret_dim, in_bk); }
tMultiBit* BinFunc::Convolution::execute(tBit* p_inputs) { return (tMultiBit*)redsec_host::execute_bits(impl, p_inputs); }
void BinFunc::Convolution::get_outhw(tRectangle* r) { r->h = (int16_t)impl->conv.Ho; r->w = (int16_t)impl->conv.Wo; }
void BinFunc::Convolution::get_outdep(uint32_t* d) { *d = impl->depth; }

BinFunc::SumPooling::SumPooling(tPoolParams* in_params) {
  impl = make_stage(false, E_NO_CONV, 0, E_SUMPOOL, E_ACTIVATION_NONE, true);
  impl->np.pool = *in_params;
}
tDimensions* BinFunc::SumPooling::prep(tDimensions* ret_dim, TFheGateBootstrappingCloudKeySet* in_bk) { return prep_stage(impl, NULL, ret_dim, in_bk); }
tMultiBit* BinFunc::SumPooling::execute(tMultiBit* p_inputs) { return (tMultiBit*)redsec_host::execute_mbits(impl, p_inputs); }
void BinFunc::SumPooling::get_outhw(tRectangle* r) { r->h = (int16_t)impl->pool.Ho; r->w = (int16_t)impl->pool.Wo; }
void BinFunc::SumPooling::get_outdep(uint32_t* d) { *d = (uint32_t)impl->quant_depth; }

// On its own a max-pool receives +-1/4096 sign bits (Quantize::execute's output). That is half a step of the 2N = 2048
// mod-switch, whose rounding noise alone has sigma ~ 2^-9 (n = 350), so neither an OR gate nor a sign bootstrap can read
// such a bit directly. The stage therefore multiplies every input by 128 (bits at +-1/32; a fresh or bootstrapped
// sample's noise of <= 2^-15 becomes <= 2^-8: 7 sigma of margin), re-encodes each bit to +-1/(4w) with one sign
// bootstrap, then ORs each window in one bootstrap (DESIGN.md "Max-pool semantics"). Inside a BinLayer none of this is
// needed: the re-encoding is the layer's own sign bootstrap of the pre-activation.
BinFunc::MaxPooling::MaxPooling(tPoolParams* in_params) {
  impl = make_stage(false, E_NO_CONV, 0, E_MAXPOOL, E_ACTIVATION_SIGN, true);
  impl->np.pool = *in_params;
  impl->prescale = 128;
}
tDimensions* BinFunc::MaxPooling::prep(tDimensions* ret_dim, TFheGateBootstrappingCloudKeySet* in_bk) { return prep_stage(impl, NULL, ret_dim, in_bk); }
tBit* BinFunc::MaxPooling::execute(tBit* p_inputs) { return (tBit*)redsec_host::execute_bits(impl, p_inputs); }

BinFunc::Quantize::Quantize(tQParams* qparam) {
  assert(qparam != NULL && qparam->shift_bits > 0);
  impl = make_stage(false, E_NO_CONV, 0, E_NO_POOL, qparam->shift_bits > 1 ? E_ACTIVATION_RELU : E_ACTIVATION_SIGN, false);
  impl->np.quant = *qparam;
  impl->shift_bits = qparam->shift_bits;
}
tDimensions* BinFunc::Quantize::prep(FILE* fd_bias, tDimensions* ret_dim, tMultiBit* p_bias, uint32_t* p_slope, TFheGateBootstrappingCloudKeySet* in_bk) {
  tDimensions* d = prep_stage(impl, fd_bias, ret_dim, in_bk);
  redsec_host::export_bias(impl, p_bias, p_slope);
  return d;
}
// the bias the stage applies is the record prep() read (the one it also wrote into p_bias)
tBit* BinFunc::Quantize::execute(tMultiBit* p_inputs, tMultiBit*) { assert(impl->e_act == E_ACTIVATION_SIGN); return (tBit*)redsec_host::execute_mbits(impl, p_inputs); }
tMultiBit* BinFunc::Quantize::add_bias(tMultiBit* p_inputs, tMultiBit*) {
  const eQuantType keep = impl->e_act;
  impl->e_act = E_ACTIVATION_NONE;
  tMultiBit* r = (tMultiBit*)redsec_host::execute_mbits(impl, p_inputs);
  impl->e_act = keep;
  return r;
}
tFixedPoint* BinFunc::Quantize::relu_shift(tMultiBit* p_inputs, tMultiBit*, uint32_t*) { assert(impl->e_act == E_ACTIVATION_RELU); return (tFixedPoint*)redsec_host::execute_mbits(impl, p_inputs); }

IntFunc::Convolution::Convolution(uint16_t out_depth, tConvParams* in_params) {
  impl = make_stage(true, E_CONV, out_depth, E_NO_POOL, E_ACTIVATION_NONE, true);
  impl->np.conv = *in_params;
}
tDimensions* IntFunc::Convolution::prep(FILE* fd_filt, tDimensions* ret_dim, TFheGateBootstrappingCloudKeySet* in_bk) { return prep_stage(impl, fd_filt, ret_dim, in_bk); }
tFixedPoint* IntFunc::Convolution::execute(tFixedPoint* p_inputs) { return (tFixedPoint*)redsec_host::execute_mbits(impl, p_inputs); }

IntFunc::SumPooling::SumPooling(tPoolParams* in_params) {
  impl = make_stage(true, E_NO_CONV, 0, E_SUMPOOL, E_ACTIVATION_NONE, true);
  impl->np.pool = *in_params;
}
tDimensions* IntFunc::SumPooling::prep(tDimensions* ret_dim, TFheGateBootstrappingCloudKeySet* in_bk) { return prep_stage(impl, NULL, ret_dim, in_bk); }
tFixedPoint* IntFunc::SumPooling::execute(tFixedPoint* p_inputs) { return (tFixedPoint*)redsec_host::execute_mbits(impl, p_inputs); }

IntFunc::Quantize::Quantize(tQParams* qparam) {   // shift_bits 0: no activation, 1: sign, > 1: ReLU (lib/IntFunc.cpp:819-838)
  assert(qparam != NULL);
  const eQuantType eq = qparam->shift_bits == 0 ? E_ACTIVATION_NONE : (qparam->shift_bits == 1 ? E_ACTIVATION_SIGN : E_ACTIVATION_RELU);
  impl = make_stage(true, E_NO_CONV, 0, E_NO_POOL, eq, false);
  impl->np.quant = *qparam;
  impl->shift_bits = qparam->shift_bits;
}
tDimensions* IntFunc::Quantize::prep(FILE* fd_bias, tDimensions* ret_dim, tMultiBit* p_bias, uint32_t* p_slope, TFheGateBootstrappingCloudKeySet* in_bk) {
  if (p_slope == NULL && impl->e_act == E_ACTIVATION_RELU) impl->np.e_bias = E_NO_BIAS;   // no slope record to read (IntFunc.cpp:802)
  tDimensions* d = prep_stage(impl, fd_bias, ret_dim, in_bk);
  redsec_host::export_bias(impl, p_bias, p_slope);
  return d;
}
tBit* IntFunc::Quantize::execute(tFixedPoint* p_inputs, tMultiBit*) { assert(impl->e_act == E_ACTIVATION_SIGN); return (tBit*)redsec_host::execute_mbits(impl, p_inputs); }
tFixedPoint* IntFunc::Quantize::add_bias(tFixedPoint* p_inputs, tMultiBit*) {
  const eQuantType keep = impl->e_act;
  impl->e_act = E_ACTIVATION_NONE;
  tFixedPoint* r = (tFixedPoint*)redsec_host::execute_mbits(impl, p_inputs);
  impl->e_act = keep;
  return r;
}
tFixedPoint* IntFunc::Quantize::relu_shift(tFixedPoint* p_inputs, tMultiBit*, uint32_t*) { assert(impl->e_act == E_ACTIVATION_RELU); return (tFixedPoint*)redsec_host::execute_mbits(impl, p_inputs); }

// ---- BinLayer ----
BinLayer::BinLayer(eConvType ec, uint16_t dep, ePoolType ep, eQuantType eq, tNetParams* np, TFheGateBootstrappingCloudKeySet* in_bk) {
  assert(ec != E_NO_CONV);
  impl = redsec_host::make_impl(false, ec, dep, ep, eq, np, in_bk);
}
tDimensions* BinLayer::prep(FILE* fd, tDimensions* dim) { return redsec_host::prep_impl(impl, fd, dim, &in_dim, &out_dim); }
void* BinLayer::execute(tBit* p_in) { return redsec_host::execute_bits(impl, p_in); }
void BinLayer::export_weights(FILE*) { printf("Weight convert not defined\r\n"); }

// ---- IntLayer ----
IntLayer::IntLayer(eConvType ec, uint16_t dep, ePoolType ep, eQuantType eq, tNetParams* np, TFheGateBootstrappingCloudKeySet* in_bk) {
  impl = redsec_host::make_impl(true, ec, dep, ep, eq, np, in_bk);
}
tDimensions* IntLayer::prep(FILE* fd, tDimensions* dim) { return redsec_host::prep_impl(impl, fd, dim, &in_dim, &out_dim); }
void* IntLayer::execute(tMultiBit* p_in) { return redsec_host::execute_mbits(impl, p_in); }
void IntLayer::export_weights(FILE*) { printf("Weight convert not defined\r\n"); }
