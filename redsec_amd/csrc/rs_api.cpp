// rs_api.cpp -- C ABI of libredsec_hip.so (declared in include/redsec_hip.h).
//
// Host-side control only: context, key upload + transform, workspace, launches. No arithmetic on
// ciphertexts happens on the CPU here, and there is no fallback path: without a HIP device every
// compute entry point returns RS_ERR_NO_DEVICE.
#include "redsec_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "rs_diag.h"
#include "rs_host.h"
#include "rs_general.h"
#include "rs_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define RS_HIP(call)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) return fail(RS_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

}  // namespace

// Everything a launch writes besides its outputs lives in a LANE = the per-stream slice of a context:
// the extracted-sample workspace between blind rotation and keyswitch, the persistent-wave work counter,
// the certificate slots, the convolution scratch and the timing events. Calls on DIFFERENT streams of one
// context therefore never share mutable state (calls on one stream are ordered by the stream itself).
constexpr unsigned kCertSlots = 1024;   // ring of per-call certificate slots; slot kCertSlots = running max, +1 = recomputed calls
struct Lane {
  hipStream_t stream = nullptr;
  int32_t* d_u0 = nullptr;
  int32_t* d_u1 = nullptr;
  size_t ws_batch = 0;
  size_t sample_words = rs::kN + 1;       // words of one extracted sample (N + 1)
  unsigned int* d_counter = nullptr;      // work counter of the persistent blind-rotate launches
  unsigned long long* d_cert = nullptr;   // [kCertSlots + 2]
  unsigned slot_next = 0;
  uint32_t* d_conv_scratch = nullptr;     // expanded weights of the tiled convolution (grown on demand)
  size_t conv_scratch_words = 0;
  uint32_t* d_ks_scratch = nullptr;       // partial sums of the sliced (small-batch) keyswitch (grown on demand; <= ~50 MB)
  size_t ks_scratch_words = 0;
  int* d_progress = nullptr;              // XCD cohort table of the split lock-step kernel: [8][kCohortSlots] step counts
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  bool ev_valid = false;
  rs::LaunchInfo last;
  // enforced split-mode certificate: the lane's running maximum is copied here (pinned host memory) behind every general-kernel
  // call, and looked at by the next call, rs_certify and rs_sync
  unsigned long long* h_split_max = nullptr;
};

struct rs_ctx {
  rs_params p{};
  int device = 0;
  int cfg = 0;  // 0 default128-shaped gadget, 1 redsec_v2-shaped gadget
  rs::Tables tables;
  int mode = 1;  // RS_MODE_FFT by default; RS_MODE_EXACT_NTT = 0
  double cert_limit = RS_CERTIFICATE_LIMIT;
  rs::LaunchOpts opts;
  double* d_tw = nullptr;          // exact-NTT tables (kTwTotal doubles)
  double* d_tw_fft = nullptr;      // FFT tables (kFftTwDoubles doubles)
  double* d_bk_ntt = nullptr;      // key in the NTT domain
  double* d_bk_fft = nullptr;      // key in the FFT domain
  // general ring path (rs_general.h): every parameter set has it; sets outside the specialised N = 1024 kernels
  // (`general`) have nothing else
  bool general = false;
  int wgs_cfg = -1;                // gadget id of the lock-step split-key kernel, or -1 (general kernels only)
  int logn = 10;
  double split_bound = 0.0;        // a-priori error bound of the split-key product (rs_general.h; the mode is offered below 1/2)
  double split_cert_limit = 0.25;    // enforced on the general kernels' rounding distances (REDSEC_SPLIT_CERT_LIMIT lowers it: tests)
  std::atomic<bool> inexact{false};  // a split-mode call rounded a value 1/4 or more away from an integer: sticky, RS_ERR_INEXACT
  double* d_tw_gen = nullptr;      // gen_make_twiddles(logn)
  double* d_bk_gen = nullptr;      // split key in the FFT domain
  int32_t* d_ksk = nullptr;
  size_t bk_bytes = 0, ksk_bytes = 0;
  bool keys = false;
  int num_cus = 256;
  bool timing = false;
  std::mutex lanes_mu;                              // guards the map (a lane itself belongs to its stream's caller)
  std::map<hipStream_t, std::unique_ptr<Lane>> lanes;
  std::mutex host_mu;                               // serialises the synchronous host-pointer calls (shared staging buffers)
  int32_t* d_io[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t io_batch = 0;
  // rs_allgather_rows: private copy stream, "my slice is computed" / "my copies are done" events, peers already enabled
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_slice = nullptr, ev_copied = nullptr, ev_staged = nullptr;
  bool copied_recorded = false;
  std::vector<int> peers_enabled, peers_denied;
  int32_t* h_stage = nullptr;                       // pinned staging buffer of this context's slice (host-staged exchange only)
  size_t h_stage_bytes = 0;
};

namespace {

struct GateCoef { int32_t bconst, sa, sb; };

// TFHE boolean-gates.cpp constants; mirrored at /root/reference/lib/GPU/gates.cu:246-286.
bool gate_coef(rs_gate_op op, GateCoef* g) {
  const int32_t e8 = 1 << 29, e4 = 1 << 30;  // modSwitchToTorus32(1,8), (1,4)
  switch (op) {
    case RS_NAND:  *g = {e8, -1, -1}; return true;
    case RS_OR:    *g = {e8, 1, 1}; return true;
    case RS_AND:   *g = {-e8, 1, 1}; return true;
    case RS_NOR:   *g = {-e8, -1, -1}; return true;
    case RS_XOR:   *g = {e4, 2, 2}; return true;
    case RS_XNOR:  *g = {-e4, -2, -2}; return true;
    case RS_ANDNY: *g = {-e8, -1, 1}; return true;
    case RS_ANDYN: *g = {-e8, 1, -1}; return true;
    case RS_ORNY:  *g = {e8, -1, 1}; return true;
    case RS_ORYN:  *g = {e8, 1, -1}; return true;
  }
  return false;
}

int use_device(rs_ctx* c) {
  if (!c) return fail(RS_ERR_INVALID, "null context");
  RS_HIP(hipSetDevice(c->device));
  return RS_OK;
}

void free_lane(Lane* ln) {
  if (ln->h_split_max) (void)hipHostFree(ln->h_split_max);
  (void)hipFree(ln->d_u0); (void)hipFree(ln->d_u1); (void)hipFree(ln->d_counter); (void)hipFree(ln->d_cert);
  (void)hipFree(ln->d_conv_scratch); (void)hipFree(ln->d_ks_scratch); (void)hipFree(ln->d_progress);
  for (auto& e : ln->ev) if (e) (void)hipEventDestroy(e);
}

// the calling stream's lane, created on first use
int lane_of(rs_ctx* c, hipStream_t st, Lane** out) {
  std::lock_guard<std::mutex> g(c->lanes_mu);
  auto it = c->lanes.find(st);
  if (it != c->lanes.end()) { *out = it->second.get(); return RS_OK; }
  std::unique_ptr<Lane> ln(new Lane);
  ln->stream = st;
  ln->sample_words = (size_t)c->p.N + 1;
  const size_t cert_bytes = sizeof(unsigned long long) * (kCertSlots + 2);
  if (hipMalloc(&ln->d_cert, cert_bytes) != hipSuccess || hipMemset(ln->d_cert, 0, cert_bytes) != hipSuccess ||
      hipHostMalloc((void**)&ln->h_split_max, sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess ||
      (!c->opts.no_persist && hipMalloc(&ln->d_counter, 256) != hipSuccess)) {
    free_lane(ln.get());
    return fail(RS_ERR_HIP, "per-stream state allocation failed");
  }
  *ln->h_split_max = 0;
  for (auto& e : ln->ev) (void)hipEventCreate(&e);
  *out = ln.get();
  c->lanes[st] = std::move(ln);
  return RS_OK;
}

int ensure_ws(Lane* ln, size_t B) {
  if (B <= ln->ws_batch) return RS_OK;
  // hipFree waits for the device: no launch can still be using the old workspace
  if (ln->d_u0) { (void)hipFree(ln->d_u0); ln->d_u0 = nullptr; }
  if (ln->d_u1) { (void)hipFree(ln->d_u1); ln->d_u1 = nullptr; }
  ln->ws_batch = 0;
  const size_t bytes = B * ln->sample_words * sizeof(int32_t);
  RS_HIP(hipMalloc(&ln->d_u0, bytes));
  RS_HIP(hipMalloc(&ln->d_u1, bytes));
  ln->ws_batch = B;
  return RS_OK;
}

// Scratch of the sliced keyswitch for this launch (rs_kernels.h); a failed allocation is not an error: the launch then takes
// the atomics form.
void attach_ks_scratch(const rs_ctx* c, Lane* ln, rs::KeyswitchArgs& k) {
  const size_t need = rs::keyswitch_scratch_words(k);
  if (need == 0 || c->opts.ks_atomics) return;
  if (need > ln->ks_scratch_words) {
    if (ln->d_ks_scratch) { (void)hipFree(ln->d_ks_scratch); ln->d_ks_scratch = nullptr; ln->ks_scratch_words = 0; }   // hipFree waits for the device
    if (hipMalloc(&ln->d_ks_scratch, need * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); ln->d_ks_scratch = nullptr; return; }
    ln->ks_scratch_words = need;
  }
  k.scratch = ln->d_ks_scratch;
  k.scratch_words = ln->ks_scratch_words;
}

// the lane's XCD cohort table (rs_kernels.h), allocated on first use; nullptr if that fails (the launch then runs free)
int* lane_progress(Lane* ln) {
  if (!ln->d_progress) {
    // 0x7f7f7f7f = "no workgroup owns this entry": also what rs_debug_cohort_table shows before the first cohort launch
    if (hipMalloc(&ln->d_progress, 8 * rs::kCohortSlots * sizeof(int)) != hipSuccess ||
        hipMemset(ln->d_progress, 0x7f, 8 * rs::kCohortSlots * sizeof(int)) != hipSuccess) {
      (void)hipGetLastError();
      if (ln->d_progress) (void)hipFree(ln->d_progress);
      ln->d_progress = nullptr;
    }
  }
  return ln->d_progress;
}

int ensure_io(rs_ctx* c, size_t B) {
  if (B <= c->io_batch) return RS_OK;
  for (auto& p : c->d_io) { if (p) { (void)hipFree(p); p = nullptr; } }
  c->io_batch = 0;
  const size_t bytes = B * (size_t)(c->p.n + 1) * sizeof(int32_t);
  for (auto& p : c->d_io) RS_HIP(hipMalloc(&p, bytes));
  c->io_batch = B;
  return RS_OK;
}

int pick_wpb(const rs_ctx* c, size_t B) {
  const size_t cus = (size_t)c->num_cus;
  if (B >= 8 * cus) return 8;
  if (B >= 4 * cus) return 4;
  if (B >= 2 * cus) return 2;
  return 1;
}

int ready(rs_ctx* c) {
  int rc = use_device(c);
  if (rc) return rc;
  if (!c->keys) return fail(RS_ERR_STATE, "rs_load_keys has not been called");
  return RS_OK;
}

// Enforced certificate of the split-key mode (general kernels). `bits` is a rounding distance as the bit pattern of a
// non-negative double (ordered like the value). A distance of 1/4 or more cannot be told from an error of the opposite sign
// around the next integer once the a-priori bound exceeds 1/4, so it poisons the context instead of passing silently.
bool split_distance_ok(const rs_ctx* c, unsigned long long bits) {
  if (rs::diag::kWrongOnPurpose) return true;   // diagnostic builds whose kernels compute wrong values on purpose (rs_diag.h)
  double d;
  memcpy(&d, &bits, sizeof d);
  return d < c->split_cert_limit;
}
int inexact_error(rs_ctx* c) {
  return fail(RS_ERR_INEXACT, "split-key mode: a rounding distance of 1/4 or more was observed (a-priori bound %.3g): results of this context are not certified; "
                              "rs_certify(reset = 1) clears the flag", c->split_bound);
}

// after a synchronisation that covers `ln`'s stream: what its last split-mode call reported is in the pinned word
int split_check_lane(rs_ctx* c, Lane* ln) {
  if (c->mode != RS_MODE_FFT_SPLIT) return RS_OK;
  if (!split_distance_ok(c, *(volatile unsigned long long*)ln->h_split_max)) c->inexact.store(true);
  return c->inexact.load() ? inexact_error(c) : RS_OK;
}

struct Combo { const int32_t* in0; const int32_t* in1; int32_t c0, c1, bconst; int32_t* u; };

struct Lut { const int32_t* table = nullptr; size_t count = 0, first = 0; };

rs::BlindRotateArgs br_args(rs_ctx* c, Lane* ln, int mode, const Combo& x, int32_t mu, const Lut& lut, size_t B) {
  rs::BlindRotateArgs a;
  a.in0 = x.in0; a.in1 = x.in1; a.c0 = x.c0; a.c1 = x.c1; a.bconst = x.bconst; a.mu = mu;
  a.bk_x = mode == 1 ? c->d_bk_fft : c->d_bk_ntt;
  a.tw = mode == 1 ? c->d_tw_fft : c->d_tw;
  a.f = c->tables.f;
  a.n = c->p.n; a.W = c->p.n + 1; a.B = (long)B; a.u_out = x.u;
  a.counter = ln->d_counter;
  a.dev_flag = nullptr;
  a.lut = lut.table; a.lut_count = (int32_t)lut.count; a.lut_first = (int32_t)lut.first;
  return a;
}

// The one path every bootstrapped entry point takes: `count` blind rotations (1, or 2 for bootsMUX) then the
// keyswitch of their (summed) extracted samples.
//   exact mode: the NTT kernels.
//   FFT mode:   the FFT kernels write the call's largest rounding distance into a fresh certificate slot, then
//               the SAME rotations are enqueued as exact-NTT launches gated on that slot: they return at once
//               unless the distance reached the limit, in which case they overwrite the extracted samples with
//               the guaranteed-exact result before the keyswitch reads them. No host round trip, stream-ordered,
//               so every *_dev result is exact (= RS_MODE_EXACT_NTT = the CPU oracle) by construction.
int run_bootstrap(rs_ctx* c, hipStream_t st, int32_t* out, const Combo* combos, int count, int32_t ks_bconst, int32_t mu,
                  const Lut& lut, size_t B) {
  Lane* ln = nullptr;
  int rc = lane_of(c, st, &ln);
  if (rc) return rc;
  rc = ensure_ws(ln, B);
  if (rc) return rc;
  const int wpb = pick_wpb(c, B);
  const int mode = c->mode;
  Combo cs[2];
  for (int k = 0; k < count; ++k) { cs[k] = combos[k]; cs[k].u = k == 0 ? ln->d_u0 : ln->d_u1; }
  if (c->timing) RS_HIP(hipEventRecord(ln->ev[0], st));
  if (mode == RS_MODE_FFT_SPLIT) {
    // exact by the a-priori bound of rs_general.h (below 1/2 for every set offered): nothing to recompute. The general kernels
    // also publish the largest rounding distance; it is ENFORCED (see split_distance_ok): what the previous call on this
    // stream reported is looked at here, without waiting for anything.
    if (!split_distance_ok(c, *(volatile unsigned long long*)ln->h_split_max)) c->inexact.store(true);
    if (c->inexact.load()) return inexact_error(c);
    bool split_wg = false;
    for (int k = 0; k < count; ++k) {
      if (c->wgs_cfg >= 0 && !c->opts.no_wg) {
        // N = 1024 with one of the reference's gadgets at throughput batch sizes: lock-step workgroups on the split key
        rs::BlindRotateArgs w = br_args(c, ln, 1, cs[k], mu, lut, B);
        w.bk_x = c->d_bk_gen; w.tw = c->d_tw_fft;
        w.progress = lane_progress(ln);   // optional: without the table the workgroups run free
        const hipError_t e = rs::launch_blind_rotate_split_wg(c->wgs_cfg, w, c->num_cus, c->opts, st, &ln->last);
        if (e == hipSuccess) { split_wg = true; continue; }
        if (e != hipErrorNotSupported) return fail(RS_ERR_HIP, "split workgroup launch failed: %s", hipGetErrorString(e));
      }
      rs::GenArgs a;
      a.in0 = cs[k].in0; a.in1 = cs[k].in1; a.c0 = cs[k].c0; a.c1 = cs[k].c1; a.bconst = cs[k].bconst; a.mu = mu;
      a.bk_x = c->d_bk_gen; a.tw = c->d_tw_gen;
      a.n = c->p.n; a.W = c->p.n + 1; a.l = c->p.bk_l; a.bgbit = c->p.bk_Bgbit; a.B = (long)B; a.u_out = cs[k].u;
      a.lut = lut.table; a.lut_count = (int32_t)lut.count; a.lut_first = (int32_t)lut.first;
      a.dev_flag = ln->d_cert + kCertSlots;   // the stream's running maximum
      RS_HIP(rs::launch_gen_blind_rotate(c->logn, a, c->num_cus, st));
    }
    if (!split_wg) RS_HIP(hipMemcpyAsync(ln->h_split_max, ln->d_cert + kCertSlots, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    if (!split_wg) {
      ln->last.form = rs::kFormGeneral; ln->last.waves_per_block = (c->p.N / 16) / 64;
      ln->last.resident = std::min<long>((long)B, rs::gen_resident_ciphertexts(c->logn, c->num_cus));
    }
  } else if (mode == RS_MODE_FFT) {
    unsigned long long* slot = ln->d_cert + (ln->slot_next++ % kCertSlots);
    RS_HIP(hipMemsetAsync(slot, 0, sizeof *slot, st));
    for (int k = 0; k < count; ++k) {
      rs::BlindRotateArgs a = br_args(c, ln, 1, cs[k], mu, lut, B);
      a.dev_flag = slot;
      a.progress = lane_progress(ln);   // used by the lock-step form only, and only when its workgroups sweep the key more than once
      RS_HIP(rs::launch_blind_rotate(c->cfg, 1, a, wpb, c->num_cus, c->opts, st, &ln->last));
    }
    unsigned long long limit_bits;
    const double lim = rs::diag::kWrongOnPurpose ? 1e300 : c->cert_limit;   // (timing probes of diagnostic builds: never recompute)
    memcpy(&limit_bits, &lim, sizeof limit_bits);
    for (int k = 0; k < count; ++k) {
      rs::BlindRotateArgs a = br_args(c, ln, 0, cs[k], mu, lut, B);
      a.gate_flag = slot; a.gate_limit_bits = limit_bits;
      a.running_flag = k == 0 ? ln->d_cert + kCertSlots : nullptr;
      a.fallback_count = k == 0 ? ln->d_cert + kCertSlots + 1 : nullptr;
      RS_HIP(rs::launch_blind_rotate(c->cfg, 0, a, wpb, c->num_cus, c->opts, st, nullptr));
    }
  } else {
    for (int k = 0; k < count; ++k)
      RS_HIP(rs::launch_blind_rotate(c->cfg, 0, br_args(c, ln, 0, cs[k], mu, lut, B), wpb, c->num_cus, c->opts, st, &ln->last));
  }
  if (c->timing) RS_HIP(hipEventRecord(ln->ev[1], st));
  if (out) {
    rs::KeyswitchArgs k;
    k.u0 = ln->d_u0; k.u1 = count == 2 ? ln->d_u1 : nullptr; k.bconst = ks_bconst; k.ksk = c->d_ksk;
    k.W = c->p.n + 1; k.t = c->p.ks_t; k.basebit = c->p.ks_basebit; k.B = (long)B; k.out = out; k.N = c->p.N;
    attach_ks_scratch(c, ln, k);
    RS_HIP(rs::launch_keyswitch(k, st));
  }
  if (c->timing) { RS_HIP(hipEventRecord(ln->ev[2], st)); ln->ev_valid = true; }
  return RS_OK;
}

void destroy_ctx(rs_ctx* c) {
  (void)hipFree(c->d_tw); (void)hipFree(c->d_tw_fft); (void)hipFree(c->d_bk_ntt); (void)hipFree(c->d_bk_fft); (void)hipFree(c->d_ksk);
  (void)hipFree(c->d_tw_gen); (void)hipFree(c->d_bk_gen);
  for (auto& kv : c->lanes) free_lane(kv.second.get());
  for (auto& p : c->d_io) (void)hipFree(p);
  if (c->ev_slice) (void)hipEventDestroy(c->ev_slice);
  if (c->ev_copied) (void)hipEventDestroy(c->ev_copied);
  if (c->ev_staged) (void)hipEventDestroy(c->ev_staged);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  delete c;
}

bool env_on(const char* name) { const char* v = getenv(name); return v && *v && strcmp(v, "0") != 0; }

}  // namespace

extern "C" {

const char* rs_last_error(void) { return g_err.c_str(); }
const char* rs_version(void) { return "redsec_hip 0.4 (gfx950; fp64 fft with on-device exact recomputation + exact fp64-carried ntt + split-key fft, N up to 8192)"; }

int rs_params_default128(rs_params* p) {
  if (!p) return fail(RS_ERR_INVALID, "null params");
  *p = {630, 1024, 1, 3, 7, 8, 2};
  return RS_OK;
}
int rs_params_redsec_small_v2(rs_params* p) {
  if (!p) return fail(RS_ERR_INVALID, "null params");
  *p = {350, 1024, 1, 10, 3, 9, 3};
  return RS_OK;
}

// client/gen_secure_keyset.cpp:9-68: the sets the reference defines beside the one it ships
int rs_params_redsec_small(rs_params* p) {
  if (!p) return fail(RS_ERR_INVALID, "null params");
  *p = {500, 1024, 1, 3, 10, 18, 1};
  return RS_OK;
}
int rs_params_redsec_medium(rs_params* p) {
  if (!p) return fail(RS_ERR_INVALID, "null params");
  *p = {3072, 4096, 1, 3, 10, 18, 1};
  return RS_OK;
}
int rs_params_redsec_large(rs_params* p) {
  if (!p) return fail(RS_ERR_INVALID, "null params");
  *p = {6144, 8192, 1, 3, 10, 18, 1};
  return RS_OK;
}

int rs_create(rs_ctx** out, const rs_params* p, int device) {
  if (!out || !p) return fail(RS_ERR_INVALID, "null argument");
  *out = nullptr;
  int logn = 0;
  while ((1 << logn) < p->N) ++logn;
  if (p->k != 1 || (1 << logn) != p->N || logn < rs::kGenMinLogN || logn > rs::kGenMaxLogN)
    return fail(RS_ERR_INVALID, "unsupported ring: N=%d k=%d (need k=1 and N in {1024, 2048, 4096, 8192})", p->N, p->k);
  if (p->n < 1 || p->n > 16384) return fail(RS_ERR_INVALID, "unsupported LWE dimension n=%d", p->n);
  if (p->ks_t < 1 || p->ks_basebit < 1 || p->ks_t * p->ks_basebit > 31) return fail(RS_ERR_INVALID, "bad keyswitch parameters");
  if (p->bk_l < 1 || p->bk_Bgbit < 1 || p->bk_l * p->bk_Bgbit > 32) return fail(RS_ERR_INVALID, "bad gadget l=%d Bgbit=%d", p->bk_l, p->bk_Bgbit);
  rs::PrimeSpec ps{};
  // the specialised kernels (exact NTT and unsplit FFT) exist for N = 1024 with the two shipped gadgets; every other
  // set runs on the general split-key path alone
  const bool special = p->N == rs::kN && rs::prime_for(p->bk_l, p->bk_Bgbit, &ps);
  const double split_bound = rs::gen_error_bound(logn, p->bk_l, p->bk_Bgbit);
  if (!special && !(split_bound < rs::kSplitBoundOffer))
    return fail(RS_ERR_INVALID, "gadget l=%d Bgbit=%d on N=%d: split-key product bound %.3g is not below 1/2", p->bk_l, p->bk_Bgbit, p->N, split_bound);
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(RS_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= count) return fail(RS_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, count);
  rs_ctx* c = new (std::nothrow) rs_ctx();
  if (!c) return fail(RS_ERR_INVALID, "out of host memory");
  c->p = *p;
  c->device = device;
  c->general = !special;
  c->logn = logn;
  c->split_bound = split_bound;
  c->cfg = special ? ((p->bk_l == 3) ? 0 : 1) : -1;
  if (special) {
    const unsigned fwd_mask = c->cfg == 0 ? rs::CfgDefault128::FWD_MASK : rs::CfgRedsecV2::FWD_MASK;
    const unsigned inv_mask = c->cfg == 0 ? rs::CfgDefault128::INV_MASK : rs::CfgRedsecV2::INV_MASK;
    const int fuse = c->cfg == 0 ? rs::CfgDefault128::FUSE : rs::CfgRedsecV2::FUSE;
    const bool mid = c->cfg == 0 ? rs::CfgDefault128::MID_REDUCE : rs::CfgRedsecV2::MID_REDUCE;
    c->tables = rs::make_tables(ps, fuse);
    const std::string why = rs::validate_schedule(c->tables.f.p, p->bk_l, p->bk_Bgbit, fwd_mask, inv_mask, fuse, mid);
    if (!why.empty()) { destroy_ctx(c); return fail(RS_ERR_INVALID, "transform schedule not exact: %s", why.c_str()); }
  } else {
    c->mode = RS_MODE_FFT_SPLIT;
  }
  if (hipSetDevice(device) != hipSuccess) { destroy_ctx(c); return fail(RS_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device); }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
    c->num_cus = prop.multiProcessorCount;
    if (!strstr(prop.gcnArchName, "gfx950")) {
      destroy_ctx(c);
      return fail(RS_ERR_NO_DEVICE, "device %d is %s; this library ships gfx950 code only", device, prop.gcnArchName);
    }
  }
  if (special) {
    if (hipMalloc(&c->d_tw, sizeof(double) * rs::kTwTotal) != hipSuccess ||
        hipMemcpy(c->d_tw, c->tables.tw.data(), sizeof(double) * rs::kTwTotal, hipMemcpyHostToDevice) != hipSuccess) {
      destroy_ctx(c);   // releases whatever was allocated
      return fail(RS_ERR_HIP, "twiddle table upload failed");
    }
  }
  // lock-step split-key kernel (N = 1024): the two shipped gadgets and redsec_params_small's; it runs on rs_fft.h's tables
  c->wgs_cfg = special ? c->cfg : ((p->N == rs::kN && p->bk_l == 3 && p->bk_Bgbit == 10) ? 2 : -1);
  if (special || c->wgs_cfg >= 0) {
    const std::vector<double> fft_tw = rs::make_fft_tables();
    if (hipMalloc(&c->d_tw_fft, sizeof(double) * rs::kFftTwDoubles) != hipSuccess ||
        hipMemcpy(c->d_tw_fft, fft_tw.data(), sizeof(double) * rs::kFftTwDoubles, hipMemcpyHostToDevice) != hipSuccess) {
      destroy_ctx(c);
      return fail(RS_ERR_HIP, "twiddle table upload failed");
    }
  }
  if (split_bound < rs::kSplitBoundOffer) {
    std::vector<double> gtw((size_t)p->N);   // M complex entries
    rs::gen_make_twiddles(logn, gtw.data());
    if (hipMalloc(&c->d_tw_gen, sizeof(double) * gtw.size()) != hipSuccess ||
        hipMemcpy(c->d_tw_gen, gtw.data(), sizeof(double) * gtw.size(), hipMemcpyHostToDevice) != hipSuccess) {
      destroy_ctx(c);
      return fail(RS_ERR_HIP, "twiddle table upload failed");
    }
  }
  // the environment is read here, once: launches never call getenv
  if (const char* m = getenv("REDSEC_MODE")) {
    if (special) c->mode = (strcmp(m, "exact") == 0 || strcmp(m, "ntt") == 0) ? RS_MODE_EXACT_NTT : (strcmp(m, "split") == 0 && c->d_tw_gen ? RS_MODE_FFT_SPLIT : RS_MODE_FFT);
  }
  if (const char* v = getenv("REDSEC_SPLIT_CERT_LIMIT")) { const double x = atof(v); if (x > 0.0 && x < 0.25) c->split_cert_limit = x; }   // test hook: can only tighten
  c->opts.no_coop = env_on("RS_NO_COOP"); c->opts.no_wg = env_on("RS_NO_WG"); c->opts.no_duo = env_on("RS_NO_DUO");
  c->opts.no_persist = env_on("RS_NO_PERSIST"); c->opts.no_conv_tiled = env_on("RS_NO_CONV_TILED");
  c->opts.no_wg4 = env_on("RS_NO_WG4"); c->opts.no_tail = env_on("RS_NO_TAIL"); c->opts.no_coop8 = env_on("RS_NO_COOP8"); c->opts.no_coop8_listed = env_on("RS_NO_COOP8_LISTED"); c->opts.ks_atomics = env_on("RS_KS_ATOMICS"); c->opts.force_host_staged = env_on("RS_FORCE_HOST_STAGED"); c->opts.no_cohort = env_on("RS_NO_COHORT");
  Lane* ln = nullptr;
  if (lane_of(c, nullptr, &ln) != RS_OK) { destroy_ctx(c); return RS_ERR_HIP; }   // the default stream's lane
  *out = c;
  return RS_OK;
}

int rs_destroy(rs_ctx* c) {
  if (!c) return RS_OK;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  // the enforced split-mode certificate of every stream's LAST call: nothing looks at it after this point, so a device-pointer
  // caller who never calls rs_sync / rs_certify still learns about a failed check here (the context is destroyed either way)
  int rc = RS_OK;
  {
    std::lock_guard<std::mutex> g(c->lanes_mu);
    for (auto& kv : c->lanes) { const int r = split_check_lane(c, kv.second.get()); if (r) rc = r; }
  }
  if (rc == RS_OK && c->inexact.load()) rc = inexact_error(c);
  destroy_ctx(c);
  return rc;
}

// bk / ksk: host arrays, or both null for the synthetic key of `seed` generated on the device (rs_load_synthetic_keys)
static int load_keys_impl(rs_ctx* c, const int32_t* bk, const int32_t* ksk, uint64_t seed) {
  int rc = use_device(c);
  if (rc) return rc;
  const bool synthetic = !bk && !ksk;
  if (!synthetic && (!bk || !ksk)) return fail(RS_ERR_INVALID, "null key pointer");
  const rs_params& p = c->p;
  const size_t n_polys = (size_t)p.n * (size_t)(2 * p.bk_l) * 2;
  const size_t bk_words = n_polys * (size_t)p.N;
  const size_t ksk_words = (size_t)p.N * p.ks_t * ((size_t)1 << p.ks_basebit) * (size_t)(p.n + 1);
  if (c->d_bk_ntt) { (void)hipFree(c->d_bk_ntt); c->d_bk_ntt = nullptr; }
  if (c->d_bk_fft) { (void)hipFree(c->d_bk_fft); c->d_bk_fft = nullptr; }
  if (c->d_bk_gen) { (void)hipFree(c->d_bk_gen); c->d_bk_gen = nullptr; }
  if (c->d_ksk) { (void)hipFree(c->d_ksk); c->d_ksk = nullptr; }
  c->keys = false;
  int32_t* d_bk = nullptr;
  RS_HIP(hipMalloc(&d_bk, bk_words * sizeof(int32_t)));
  if (synthetic) RS_HIP(rs::launch_synthetic_words(d_bk, seed, bk_words, nullptr));
  else RS_HIP(hipMemcpy(d_bk, bk, bk_words * sizeof(int32_t), hipMemcpyHostToDevice));
  c->bk_bytes = 0;
  if (!c->general) {
    // both transform domains are kept resident (62 + 62 MB default-128, 115 + 115 MB REDsec): the FFT mode's
    // gated exact recomputation needs the NTT-domain key, and the mode can be switched per call
    RS_HIP(hipMalloc(&c->d_bk_ntt, bk_words * sizeof(double)));
    RS_HIP(hipMalloc(&c->d_bk_fft, bk_words * sizeof(double)));
    RS_HIP(rs::launch_bk_transform(c->cfg, 0, d_bk, c->d_bk_ntt, c->d_tw, c->tables.f, c->tables.ninv, (long)n_polys, nullptr));
    RS_HIP(rs::launch_bk_transform(c->cfg, 1, d_bk, c->d_bk_fft, c->d_tw_fft, c->tables.f, 0.0, (long)n_polys, nullptr));
    c->bk_bytes = bk_words * sizeof(double);
  }
  if (c->d_tw_gen) {
    // the split key: two transformed halves per polynomial (twice the bytes of one domain)
    RS_HIP(hipMalloc(&c->d_bk_gen, 2 * bk_words * sizeof(double)));
    RS_HIP(rs::launch_gen_bk_transform(c->logn, d_bk, c->d_bk_gen, c->d_tw_gen, (long)n_polys, c->num_cus, nullptr));
    if (c->general) c->bk_bytes = 2 * bk_words * sizeof(double);
  }
  RS_HIP(hipDeviceSynchronize());
  RS_HIP(hipFree(d_bk));
  RS_HIP(hipMalloc(&c->d_ksk, ksk_words * sizeof(int32_t)));
  if (synthetic) { RS_HIP(rs::launch_synthetic_words(c->d_ksk, seed ^ 0x6b73ull, ksk_words, nullptr)); RS_HIP(hipDeviceSynchronize()); }
  else RS_HIP(hipMemcpy(c->d_ksk, ksk, ksk_words * sizeof(int32_t), hipMemcpyHostToDevice));
  c->ksk_bytes = ksk_words * sizeof(int32_t);
  c->keys = true;
  return RS_OK;
}

int rs_load_keys(rs_ctx* c, const int32_t* bk, const int32_t* ksk) {
  if (!bk || !ksk) return fail(RS_ERR_INVALID, "null key pointer");
  return load_keys_impl(c, bk, ksk, 0);
}
int rs_load_synthetic_keys(rs_ctx* c, uint64_t seed) { return load_keys_impl(c, nullptr, nullptr, seed); }

int rs_reserve(rs_ctx* c, size_t max_batch) { return rs_reserve_stream(c, max_batch, nullptr); }

int rs_reserve_stream(rs_ctx* c, size_t max_batch, void* stream) {
  int rc = use_device(c);
  if (rc) return rc;
  Lane* ln = nullptr;
  rc = lane_of(c, (hipStream_t)stream, &ln);
  if (rc) return rc;
  return ensure_ws(ln, max_batch);
}

int rs_bootstrap_dev(rs_ctx* c, int32_t* out, const int32_t* in, int32_t mu, size_t B, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  if (B == 0) return RS_OK;
  if (!out || !in) return fail(RS_ERR_INVALID, "null ciphertext pointer");
  const Combo x{in, nullptr, 1, 0, 0, nullptr};
  return run_bootstrap(c, (hipStream_t)stream, out, &x, 1, 0, mu, Lut{}, B);
}

int rs_bootstrap_lut_dev(rs_ctx* c, int32_t* out, const int32_t* in, const int32_t* lut, size_t lut_count, size_t lut_first, size_t B,
                         void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  if (B == 0) return RS_OK;
  if (!out || !in || !lut) return fail(RS_ERR_INVALID, "null pointer");
  if (lut_count < 1 || lut_count > 0x7fffffffu) return fail(RS_ERR_INVALID, "lut_count must be >= 1");
  const Combo x{in, nullptr, 1, 0, 0, nullptr};
  Lut l;
  l.table = lut; l.count = lut_count; l.first = lut_first % lut_count;
  return run_bootstrap(c, (hipStream_t)stream, out, &x, 1, 0, 0, l, B);
}

int rs_gate_mu_dev(rs_ctx* c, rs_gate_op op, int32_t* out, const int32_t* a, const int32_t* b, int32_t mu, size_t B, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  GateCoef g;
  if (!gate_coef(op, &g)) return fail(RS_ERR_INVALID, "unknown gate %d", (int)op);
  if (B == 0) return RS_OK;
  if (!out || !a || !b) return fail(RS_ERR_INVALID, "null ciphertext pointer");
  const Combo x{a, b, g.sa, g.sb, g.bconst, nullptr};
  return run_bootstrap(c, (hipStream_t)stream, out, &x, 1, 0, mu, Lut{}, B);
}

int rs_gate_dev(rs_ctx* c, rs_gate_op op, int32_t* out, const int32_t* a, const int32_t* b, size_t B, void* stream) {
  return rs_gate_mu_dev(c, op, out, a, b, 1 << 29, B, stream);
}

int rs_gather_rows_dev(rs_ctx* c, int32_t* out, const int32_t* in, const int32_t* row_index, size_t B, void* stream) {
  int rc = use_device(c);
  if (rc) return rc;
  if (B == 0) return RS_OK;
  if (!out || !in || !row_index) return fail(RS_ERR_INVALID, "null pointer");
  RS_HIP(rs::launch_gather_rows(out, in, row_index, c->p.n + 1, (long)B, (hipStream_t)stream));
  return RS_OK;
}

int rs_mux_dev(rs_ctx* c, int32_t* out, const int32_t* a, const int32_t* b, const int32_t* cc, size_t B, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  if (B == 0) return RS_OK;
  if (!out || !a || !b || !cc) return fail(RS_ERR_INVALID, "null ciphertext pointer");
  const int32_t e8 = 1 << 29;
  // u1 = woKS(AND(a,b)), u2 = woKS(ANDNY(a,c)); out = KS((0,1/8) + u1 + u2)
  const Combo xs[2] = {{a, b, 1, 1, -e8, nullptr}, {a, cc, -1, 1, -e8, nullptr}};
  return run_bootstrap(c, (hipStream_t)stream, out, xs, 2, e8, e8, Lut{}, B);
}

int rs_bootstrap_wo_ks_dev(rs_ctx* c, int32_t* u, const int32_t* in, int32_t mu, size_t B, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  if (B == 0) return RS_OK;
  if (!u || !in) return fail(RS_ERR_INVALID, "null ciphertext pointer");
  // rotate into the lane workspace (so that the gated exact recomputation applies), then hand the samples over
  const Combo x{in, nullptr, 1, 0, 0, nullptr};
  hipStream_t st = (hipStream_t)stream;
  rc = run_bootstrap(c, st, nullptr, &x, 1, 0, mu, Lut{}, B);
  if (rc) return rc;
  Lane* ln = nullptr;
  rc = lane_of(c, st, &ln);
  if (rc) return rc;
  RS_HIP(hipMemcpyAsync(u, ln->d_u0, B * ln->sample_words * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  return RS_OK;
}

int rs_keyswitch_dev(rs_ctx* c, int32_t* out, const int32_t* u, size_t B, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  if (B == 0) return RS_OK;
  if (!out || !u) return fail(RS_ERR_INVALID, "null ciphertext pointer");
  rs::KeyswitchArgs k;
  k.u0 = u; k.u1 = nullptr; k.bconst = 0; k.ksk = c->d_ksk;
  k.W = c->p.n + 1; k.t = c->p.ks_t; k.basebit = c->p.ks_basebit; k.B = (long)B; k.out = out; k.N = c->p.N;
  Lane* ln = nullptr;
  rc = lane_of(c, (hipStream_t)stream, &ln);
  if (rc) return rc;
  attach_ks_scratch(c, ln, k);
  RS_HIP(rs::launch_keyswitch(k, (hipStream_t)stream));
  return RS_OK;
}

// ---- host-pointer conveniences: H2D, the *_dev call on the default stream, D2H. Serialised per context
// (the per-ciphertext TFHE-style wrappers above this ABI are called from OpenMP regions). ----
static int host_roundtrip(rs_ctx* c, int32_t* out, const int32_t* const* ins, int n_in, size_t B,
                          int (*run)(rs_ctx*, int32_t*, int32_t* const*, size_t, void*), void* extra) {
  int rc = ready(c);
  if (rc) return rc;
  if (B == 0) return RS_OK;
  std::lock_guard<std::mutex> g(c->host_mu);
  rc = ensure_io(c, B);
  if (rc) return rc;
  const size_t bytes = B * (size_t)(c->p.n + 1) * sizeof(int32_t);
  for (int i = 0; i < n_in; ++i)
    if (!ins[i]) return fail(RS_ERR_INVALID, "null ciphertext pointer");
  for (int i = 0; i < n_in; ++i) RS_HIP(hipMemcpy(c->d_io[i], ins[i], bytes, hipMemcpyHostToDevice));
  rc = run(c, c->d_io[3], c->d_io, B, extra);
  if (rc) return rc;
  RS_HIP(hipStreamSynchronize(nullptr));
  {
    Lane* ln = nullptr;
    rc = lane_of(c, nullptr, &ln);
    if (rc) return rc;
    rc = split_check_lane(c, ln);   // a host call never hands back an uncertified split-mode result
    if (rc) return rc;
  }
  RS_HIP(hipMemcpy(out, c->d_io[3], bytes, hipMemcpyDeviceToHost));
  return RS_OK;
}

int rs_bootstrap(rs_ctx* c, int32_t* out, const int32_t* in, int32_t mu, size_t B) {
  if (!out) return fail(RS_ERR_INVALID, "null output");
  const int32_t* ins[1] = {in};
  return host_roundtrip(c, out, ins, 1, B,
                        [](rs_ctx* cc, int32_t* o, int32_t* const* d, size_t b, void* ex) {
                          return rs_bootstrap_dev(cc, o, d[0], *(int32_t*)ex, b, nullptr);
                        }, &mu);
}

int rs_gate(rs_ctx* c, rs_gate_op op, int32_t* out, const int32_t* a, const int32_t* b, size_t B) {
  if (!out) return fail(RS_ERR_INVALID, "null output");
  const int32_t* ins[2] = {a, b};
  return host_roundtrip(c, out, ins, 2, B,
                        [](rs_ctx* cc, int32_t* o, int32_t* const* d, size_t bb, void* ex) {
                          return rs_gate_dev(cc, *(rs_gate_op*)ex, o, d[0], d[1], bb, nullptr);
                        }, &op);
}

int rs_mux(rs_ctx* c, int32_t* out, const int32_t* a, const int32_t* b, const int32_t* cc, size_t B) {
  if (!out) return fail(RS_ERR_INVALID, "null output");
  const int32_t* ins[3] = {a, b, cc};
  return host_roundtrip(c, out, ins, 3, B,
                        [](rs_ctx* c2, int32_t* o, int32_t* const* d, size_t bb, void*) {
                          return rs_mux_dev(c2, o, d[0], d[1], d[2], bb, nullptr);
                        }, nullptr);
}

int rs_debug_polymul(rs_ctx* c, int32_t* out, const int32_t* a_small, const int32_t* b_torus, size_t count) {
  int rc = use_device(c);
  if (rc) return rc;
  if (count == 0) return RS_OK;
  if (!out || !a_small || !b_torus) return fail(RS_ERR_INVALID, "null pointer");
  Lane* ln = nullptr;
  rc = lane_of(c, nullptr, &ln);
  if (rc) return rc;
  const size_t bytes = count * (size_t)c->p.N * sizeof(int32_t);
  int32_t *da = nullptr, *db = nullptr, *dout = nullptr;
  double* scratch = nullptr;
  RS_HIP(hipMalloc(&da, bytes)); RS_HIP(hipMalloc(&db, bytes)); RS_HIP(hipMalloc(&dout, bytes));
  RS_HIP(hipMalloc(&scratch, 2 * count * (size_t)c->p.N * sizeof(double)));
  RS_HIP(hipMemcpy(da, a_small, bytes, hipMemcpyHostToDevice));
  RS_HIP(hipMemcpy(db, b_torus, bytes, hipMemcpyHostToDevice));
  if (c->mode == RS_MODE_FFT_SPLIT) {
    RS_HIP(rs::launch_gen_polymul(c->logn, da, db, dout, scratch, c->d_tw_gen, (long)count, ln->d_cert + kCertSlots, c->num_cus, nullptr));
  } else {
    RS_HIP(rs::launch_polymul(c->cfg, c->mode, da, db, dout, scratch, c->mode == 1 ? c->d_tw_fft : c->d_tw, c->tables.f, c->tables.ninv,
                              (long)count, c->mode == 1 ? ln->d_cert + kCertSlots : nullptr, nullptr));
  }
  RS_HIP(hipDeviceSynchronize());
  RS_HIP(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout); (void)hipFree(scratch);
  return RS_OK;
}

int rs_set_mode(rs_ctx* c, int mode) {
  if (!c) return fail(RS_ERR_INVALID, "null context");
  if (mode != RS_MODE_EXACT_NTT && mode != RS_MODE_FFT && mode != RS_MODE_FFT_SPLIT) return fail(RS_ERR_INVALID, "unknown mode %d", mode);
  if (c->general && mode != RS_MODE_FFT_SPLIT)
    return fail(RS_ERR_INVALID, "this parameter set (N=%d l=%d Bgbit=%d) runs on the split-key path only", c->p.N, c->p.bk_l, c->p.bk_Bgbit);
  if (mode == RS_MODE_FFT_SPLIT && !c->d_tw_gen)
    return fail(RS_ERR_INVALID, "split-key mode not offered for this set: a-priori bound %.3g", c->split_bound);
  c->mode = mode;
  return RS_OK;
}

int rs_split_bound(rs_ctx* c, double* bound) {
  if (!c || !bound) return fail(RS_ERR_INVALID, "null argument");
  *bound = c->split_bound;
  return RS_OK;
}

int rs_get_mode(rs_ctx* c, int* mode) {
  if (!c || !mode) return fail(RS_ERR_INVALID, "null argument");
  *mode = c->mode;
  return RS_OK;
}

int rs_set_certificate_limit(rs_ctx* c, double limit) {
  if (!c) return fail(RS_ERR_INVALID, "null context");
  if (!(limit >= 0.0 && limit <= 0.5)) return fail(RS_ERR_INVALID, "certificate limit must lie in [0, 0.5]");
  c->cert_limit = limit;
  return RS_OK;
}

// running maximum / recomputed-call count of one lane (device must be idle on that lane)
static int read_lane(Lane* ln, double* dist, int64_t* fallbacks, bool reset) {
  unsigned long long v[2] = {0, 0};
  RS_HIP(hipMemcpy(v, ln->d_cert + kCertSlots, sizeof v, hipMemcpyDeviceToHost));
  memcpy(dist, &v[0], sizeof *dist);
  *fallbacks = (int64_t)v[1];
  if (reset) { RS_HIP(hipMemset(ln->d_cert + kCertSlots, 0, sizeof(unsigned long long))); *ln->h_split_max = 0; }
  return RS_OK;
}
// split mode: the synchronised running maximum against the enforced limit (see split_distance_ok)
static int split_verdict(rs_ctx* c, double dist, bool reset) {
  if (c->mode != RS_MODE_FFT_SPLIT) return RS_OK;
  if (!(dist < c->split_cert_limit)) c->inexact.store(true);
  const bool bad = c->inexact.load();
  if (reset) c->inexact.store(false);
  return bad ? inexact_error(c) : RS_OK;
}

int rs_certify(rs_ctx* c, void* stream, double* max_distance, int64_t* recomputed_calls, int reset) {
  int rc = use_device(c);
  if (rc) return rc;
  Lane* ln = nullptr;
  rc = lane_of(c, (hipStream_t)stream, &ln);
  if (rc) return rc;
  RS_HIP(hipStreamSynchronize((hipStream_t)stream));
  double d = 0.0;
  int64_t n = 0;
  rc = read_lane(ln, &d, &n, reset != 0);
  if (rc) return rc;
  if (max_distance) *max_distance = d;
  if (recomputed_calls) *recomputed_calls = n;
  return split_verdict(c, d, reset != 0);
}

int rs_rounding_certificate(rs_ctx* c, double* max_distance, int reset) {
  int rc = use_device(c);
  if (rc) return rc;
  if (!max_distance) return fail(RS_ERR_INVALID, "null pointer");
  RS_HIP(hipDeviceSynchronize());
  std::lock_guard<std::mutex> g(c->lanes_mu);
  double best = 0.0;
  for (auto& kv : c->lanes) {
    double d = 0.0;
    int64_t n = 0;
    rc = read_lane(kv.second.get(), &d, &n, reset != 0);
    if (rc) return rc;
    if (d > best) best = d;
  }
  *max_distance = best;
  return split_verdict(c, best, reset != 0);
}

int rs_fft_fallbacks(rs_ctx* c, int64_t* count) {
  int rc = use_device(c);
  if (rc) return rc;
  if (!count) return fail(RS_ERR_INVALID, "null argument");
  RS_HIP(hipDeviceSynchronize());
  std::lock_guard<std::mutex> g(c->lanes_mu);
  int64_t total = 0;
  for (auto& kv : c->lanes) {
    double d = 0.0;
    int64_t n = 0;
    rc = read_lane(kv.second.get(), &d, &n, false);
    if (rc) return rc;
    total += n;
  }
  *count = total;
  return RS_OK;
}

// ---- linear stage ----
int rs_lincomb_dev(rs_ctx* c, int32_t* out, const int32_t* a, int32_t ca, const int32_t* b, int32_t cb, int32_t bconst, size_t B,
                   void* stream) {
  int rc = use_device(c);
  if (rc) return rc;
  if (B == 0) return RS_OK;
  if (!out || !a) return fail(RS_ERR_INVALID, "null ciphertext pointer");
  RS_HIP(rs::launch_lincomb(out, a, ca, b, cb, bconst, c->p.n + 1, (long)B, (hipStream_t)stream));
  return RS_OK;
}

int rs_linear_fc_dev(rs_ctx* c, int32_t* out, const int32_t* in, const uint8_t* sign, const uint8_t* zero, int32_t K, int32_t M,
                     int32_t zero_tap_b, const int32_t* bias_b, int32_t bias_depth, void* stream) {
  int rc = use_device(c);
  if (rc) return rc;
  if (!out || !in || !sign) return fail(RS_ERR_INVALID, "null pointer");
  if (K < 1 || M < 1 || M > 65535) return fail(RS_ERR_INVALID, "bad fully-connected shape K=%d M=%d", K, M);
  if (bias_b && bias_depth < 1) return fail(RS_ERR_INVALID, "bias_depth must be >= 1");
  RS_HIP(rs::launch_linear_fc(out, in, sign, zero, K, M, c->p.n + 1, zero_tap_b, bias_b, bias_depth, (hipStream_t)stream));
  return RS_OK;
}

int rs_conv_ternary_dev(rs_ctx* c, int32_t* out, const int32_t* in, const uint8_t* sign, const uint8_t* zero,
                        const rs_conv_shape* s, int32_t zero_tap_b, int32_t pad_tap_b, const int32_t* bias_b, int32_t bias_depth,
                        void* stream) {
  int rc = use_device(c);
  if (rc) return rc;
  if (!out || !in || !sign || !s) return fail(RS_ERR_INVALID, "null pointer");
  if (s->Cout < 1 || s->Cout > 65535 || s->Ho * s->Wo < 1 || s->Ho * s->Wo > 65535 || s->stride_h < 1 || s->stride_w < 1)
    return fail(RS_ERR_INVALID, "bad convolution shape");
  if (bias_b && bias_depth < 1) return fail(RS_ERR_INVALID, "bias_depth must be >= 1");
  rs::ConvShape cs{s->H, s->Wd, s->Cin, s->Cout, s->fh, s->fw, s->stride_h, s->stride_w, s->off_h, s->off_w, s->Ho, s->Wo};
  if (zero_tap_b == 0 && pad_tap_b == 0 && !c->opts.no_conv_tiled) {
    // BinFunc-style convolution (ternary-zero and padding taps contribute nothing): register-tiled kernel.
    // The expanded weights live in the stream's own scratch buffer, rebuilt on every call (microseconds).
    Lane* ln = nullptr;
    rc = lane_of(c, (hipStream_t)stream, &ln);
    if (rc) return rc;
    const size_t words = rs::conv_tiled_scratch_words(cs);
    if (words > ln->conv_scratch_words) {
      RS_HIP(hipStreamSynchronize((hipStream_t)stream));   // a previous call may still be reading the old buffer
      if (ln->d_conv_scratch) RS_HIP(hipFree(ln->d_conv_scratch));
      ln->d_conv_scratch = nullptr; ln->conv_scratch_words = 0;
      RS_HIP(hipMalloc(&ln->d_conv_scratch, words * sizeof(uint32_t)));
      ln->conv_scratch_words = words;
    }
    RS_HIP(rs::launch_conv_ternary_tiled(out, in, sign, zero, cs, c->p.n + 1, bias_b, bias_depth, ln->d_conv_scratch, (hipStream_t)stream));
    return RS_OK;
  }
  RS_HIP(rs::launch_conv_ternary(out, in, sign, zero, cs, c->p.n + 1, zero_tap_b, pad_tap_b, bias_b, bias_depth, (hipStream_t)stream));
  return RS_OK;
}

int rs_sumpool_dev(rs_ctx* c, int32_t* out, const int32_t* in, const rs_pool_shape* s, const int32_t* bias_b, int32_t bias_depth,
                   void* stream) {
  int rc = use_device(c);
  if (rc) return rc;
  if (!out || !in || !s) return fail(RS_ERR_INVALID, "null pointer");
  if (s->C < 1 || s->C > 65535 || s->Ho * s->Wo < 1 || s->Ho * s->Wo > 65535 || s->stride_h < 1 || s->stride_w < 1)
    return fail(RS_ERR_INVALID, "bad pooling shape");
  if (bias_b && bias_depth < 1) return fail(RS_ERR_INVALID, "bias_depth must be >= 1");
  rs::PoolShape ps{s->H, s->Wd, s->C, s->win_h, s->win_w, s->stride_h, s->stride_w, s->off_h, s->off_w, s->Ho, s->Wo};
  RS_HIP(rs::launch_sumpool(out, in, ps, c->p.n + 1, bias_b, bias_depth, (hipStream_t)stream));
  return RS_OK;
}

// ---- memory helpers ----
int rs_dev_alloc(rs_ctx* c, void** ptr, size_t bytes) {
  int rc = use_device(c);
  if (rc) return rc;
  if (!ptr) return fail(RS_ERR_INVALID, "null pointer");
  RS_HIP(hipMalloc(ptr, bytes ? bytes : 1));
  return RS_OK;
}
int rs_dev_free(rs_ctx* c, void* ptr) {
  int rc = use_device(c);
  if (rc) return rc;
  RS_HIP(hipFree(ptr));
  return RS_OK;
}
int rs_copy_to_dev(rs_ctx* c, void* dst, const void* src, size_t bytes) {
  int rc = use_device(c);
  if (rc) return rc;
  RS_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return RS_OK;
}
int rs_copy_to_host(rs_ctx* c, void* dst, const void* src, size_t bytes) {
  int rc = use_device(c);
  if (rc) return rc;
  RS_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return RS_OK;
}
int rs_copy_dev_to_dev(rs_ctx* dc, void* dst, rs_ctx* sc, const void* src, size_t bytes) {
  if (!dc || !sc) return fail(RS_ERR_INVALID, "null context");
  if (bytes == 0) return RS_OK;
  if (!dst || !src) return fail(RS_ERR_INVALID, "null pointer");
  RS_HIP(hipSetDevice(dc->device));
  if (dc->device == sc->device) RS_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
  else RS_HIP(hipMemcpyPeer(dst, dc->device, src, sc->device, bytes));   // staged through the host where peer access is off
  return RS_OK;
}
static int exchange_state(rs_ctx* c) {
  RS_HIP(hipSetDevice(c->device));
  if (!c->copy_stream) RS_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  if (!c->ev_slice) RS_HIP(hipEventCreateWithFlags(&c->ev_slice, hipEventDisableTiming));
  if (!c->ev_copied) RS_HIP(hipEventCreateWithFlags(&c->ev_copied, hipEventDisableTiming));
  if (!c->ev_staged) RS_HIP(hipEventCreateWithFlags(&c->ev_staged, hipEventDisableTiming));
  return RS_OK;
}
// May context dc's device read context sc's device directly? Asked once per device pair; a refusal (no link, or enabling fails)
// is remembered and the pair exchanges through pinned host memory from then on.
static bool peer_access(rs_ctx* dc, rs_ctx* sc) {
  if (dc->device == sc->device) return true;
  if (std::find(dc->peers_enabled.begin(), dc->peers_enabled.end(), sc->device) != dc->peers_enabled.end()) return true;
  if (std::find(dc->peers_denied.begin(), dc->peers_denied.end(), sc->device) != dc->peers_denied.end()) return false;
  int can = 0;
  bool ok = hipSetDevice(dc->device) == hipSuccess && hipDeviceCanAccessPeer(&can, dc->device, sc->device) == hipSuccess && can;
  if (ok) {
    const hipError_t pe = hipDeviceEnablePeerAccess(sc->device, 0);
    ok = pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled;
  }
  (void)hipGetLastError();
  (ok ? dc->peers_enabled : dc->peers_denied).push_back(sc->device);
  return ok;
}
int rs_allgather_rows(rs_ctx* const* ctxs, int n, int32_t* const* bufs, size_t rows, size_t row_words) {
  if (!ctxs || !bufs || n < 1) return fail(RS_ERR_INVALID, "null argument");
  if (n == 1 || rows == 0) return RS_OK;
  std::vector<int> devices(n);
  for (int d = 0; d < n; ++d) {
    if (!ctxs[d] || !bufs[d]) return fail(RS_ERR_INVALID, "null context or buffer %d", d);
    const int rc = exchange_state(ctxs[d]);
    if (rc) return rc;
    // behind everything queued on device d's default stream so far: the kernels that wrote slice d, and the kernels that still
    // read the block bufs[d] was recycled from (see rs_host.h, exchange_plan)
    RS_HIP(hipEventRecord(ctxs[d]->ev_slice, nullptr));
    devices[d] = ctxs[d]->device;
  }
  std::vector<unsigned char> peer((size_t)n * n, 1);
  bool force_staged = false;
  for (int d = 0; d < n; ++d) {
    force_staged = force_staged || ctxs[d]->opts.force_host_staged;
    for (int e = 0; e < n; ++e) if (e != d) peer[(size_t)d * n + e] = peer_access(ctxs[d], ctxs[e]) ? 1 : 0;
  }
  const std::vector<rs::ExchangeOp> ops = rs::exchange_plan(rows, n, devices.data(), peer.data(), force_staged);
  for (const rs::ExchangeOp& op : ops) {
    rs_ctx* c = ctxs[op.ctx];
    rs_ctx* o = ctxs[op.other];
    RS_HIP(hipSetDevice(c->device));
    const size_t off = op.lo * row_words, bytes = (op.hi - op.lo) * row_words * sizeof(int32_t);
    switch (op.kind) {
      case rs::kOpWaitSlice: RS_HIP(hipStreamWaitEvent(c->copy_stream, o->ev_slice, 0)); break;
      case rs::kOpWaitCopied: if (o->copied_recorded) RS_HIP(hipStreamWaitEvent(c->copy_stream, o->ev_copied, 0)); break;
      case rs::kOpWaitStaged: RS_HIP(hipStreamWaitEvent(c->copy_stream, o->ev_staged, 0)); break;
      case rs::kOpStageOut:
        if (bytes > c->h_stage_bytes) {
          // growing the pinned buffer: nothing of an earlier exchange may still read the old one
          for (int e = 0; e < n; ++e) { RS_HIP(hipSetDevice(ctxs[e]->device)); RS_HIP(hipStreamSynchronize(ctxs[e]->copy_stream)); }
          RS_HIP(hipSetDevice(c->device));
          if (c->h_stage) { (void)hipHostFree(c->h_stage); c->h_stage = nullptr; c->h_stage_bytes = 0; }
          RS_HIP(hipHostMalloc((void**)&c->h_stage, bytes, hipHostMallocPortable));
          c->h_stage_bytes = bytes;
        }
        RS_HIP(hipMemcpyAsync(c->h_stage, bufs[op.ctx] + off, bytes, hipMemcpyDeviceToHost, c->copy_stream));
        RS_HIP(hipEventRecord(c->ev_staged, c->copy_stream));
        break;
      case rs::kOpCopy:
        if (op.path == rs::kPathSameDevice) RS_HIP(hipMemcpyAsync(bufs[op.ctx] + off, bufs[op.other] + off, bytes, hipMemcpyDeviceToDevice, c->copy_stream));
        else if (op.path == rs::kPathPeer) RS_HIP(hipMemcpyPeerAsync(bufs[op.ctx] + off, c->device, bufs[op.other] + off, o->device, bytes, c->copy_stream));
        else RS_HIP(hipMemcpyAsync(bufs[op.ctx] + off, o->h_stage, bytes, hipMemcpyHostToDevice, c->copy_stream));
        break;
      case rs::kOpRecordCopied: RS_HIP(hipEventRecord(c->ev_copied, c->copy_stream)); c->copied_recorded = true; break;
    }
  }
  // every default stream continues behind ALL copies: its own (it reads the gathered slices next) and the others' (they read
  // its slice; the caller may reuse or free the buffer in stream order afterwards)
  for (int d = 0; d < n; ++d) {
    RS_HIP(hipSetDevice(ctxs[d]->device));
    for (int e = 0; e < n; ++e) RS_HIP(hipStreamWaitEvent(nullptr, ctxs[e]->ev_copied, 0));
  }
  return RS_OK;
}
int rs_release_stream(rs_ctx* c, void* stream) {
  int rc = use_device(c);
  if (rc) return rc;
  if (!stream) return fail(RS_ERR_INVALID, "the default stream's state lives as long as the context");
  RS_HIP(hipStreamSynchronize((hipStream_t)stream));
  std::unique_ptr<Lane> ln;
  {
    std::lock_guard<std::mutex> g(c->lanes_mu);
    auto it = c->lanes.find((hipStream_t)stream);
    if (it == c->lanes.end()) return RS_OK;
    ln = std::move(it->second);
    c->lanes.erase(it);
  }
  rc = split_check_lane(c, ln.get());   // what its last split-mode call reported is not lost with the lane
  free_lane(ln.get());
  return rc;
}
int rs_sync(rs_ctx* c) {
  int rc = use_device(c);
  if (rc) return rc;
  RS_HIP(hipDeviceSynchronize());
  std::lock_guard<std::mutex> g(c->lanes_mu);
  for (auto& kv : c->lanes) {
    rc = split_check_lane(c, kv.second.get());
    if (rc) return rc;
  }
  return RS_OK;
}

int rs_set_timing(rs_ctx* c, int enable) {
  if (!c) return fail(RS_ERR_INVALID, "null context");
  c->timing = enable != 0;
  std::lock_guard<std::mutex> g(c->lanes_mu);
  for (auto& kv : c->lanes) kv.second->ev_valid = false;
  return RS_OK;
}

int rs_last_kernel_ms_stream(rs_ctx* c, void* stream, float* br_ms, float* ks_ms) {
  int rc = use_device(c);
  if (rc) return rc;
  Lane* ln = nullptr;
  rc = lane_of(c, (hipStream_t)stream, &ln);
  if (rc) return rc;
  if (!ln->ev_valid) return fail(RS_ERR_STATE, "no timed launch recorded on this stream");
  RS_HIP(hipEventSynchronize(ln->ev[2]));
  float a = -1.f, b = -1.f;
  RS_HIP(hipEventElapsedTime(&a, ln->ev[0], ln->ev[1]));
  RS_HIP(hipEventElapsedTime(&b, ln->ev[1], ln->ev[2]));
  if (br_ms) *br_ms = a;
  if (ks_ms) *ks_ms = b;
  return RS_OK;
}
int rs_last_kernel_ms(rs_ctx* c, float* br_ms, float* ks_ms) { return rs_last_kernel_ms_stream(c, nullptr, br_ms, ks_ms); }

int rs_last_launch(rs_ctx* c, void* stream, int32_t* form, int32_t* waves_per_block, int64_t* resident) {
  if (!c) return fail(RS_ERR_INVALID, "null context");
  Lane* ln = nullptr;
  int rc = lane_of(c, (hipStream_t)stream, &ln);
  if (rc) return rc;
  if (ln->last.form < 0) return fail(RS_ERR_STATE, "no blind rotation launched on this stream yet");
  if (form) *form = ln->last.form;
  if (waves_per_block) *waves_per_block = ln->last.waves_per_block;
  if (resident) *resident = ln->last.resident;
  return RS_OK;
}

int rs_debug_fp64_rate(rs_ctx* c, double* lane_ops_per_s) {
  if (!c || !lane_ops_per_s) return fail(RS_ERR_INVALID, "null argument");
  RS_HIP(hipSetDevice(c->device));
  double* d_out = nullptr;
  RS_HIP(hipMalloc(&d_out, (size_t)c->num_cus * 512 * sizeof(double)));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t err = hipEventCreate(&e0);
  if (err == hipSuccess) err = hipEventCreate(&e1);
  double ops = 0.0, best_ms = 1e30;
  for (int rep = 0; rep < 4 && err == hipSuccess; ++rep) {   // the first launch warms up; the best of three counts
    err = hipEventRecord(e0, nullptr);
    if (err == hipSuccess) err = rs::launch_fp64_rate(d_out, c->num_cus, 1 << 17, &ops, nullptr);
    if (err == hipSuccess) err = hipEventRecord(e1, nullptr);
    if (err == hipSuccess) err = hipEventSynchronize(e1);
    float ms = 0.f;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    if (err == hipSuccess && rep > 0 && ms < best_ms) best_ms = ms;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(d_out);
  if (err != hipSuccess) return fail(RS_ERR_HIP, "rs_debug_fp64_rate: %s", hipGetErrorString(err));
  *lane_ops_per_s = ops / (best_ms * 1e-3);
  return RS_OK;
}

int rs_debug_cohort_table(rs_ctx* c, void* stream, int32_t* out) {
  if (!c || !out) return fail(RS_ERR_INVALID, "null argument");
  Lane* ln = nullptr;
  int rc = lane_of(c, (hipStream_t)stream, &ln);
  if (rc) return rc;
  if (!ln->d_progress) return fail(RS_ERR_STATE, "no lock-step launch with XCD cohorts on this stream yet");
  RS_HIP(hipStreamSynchronize(ln->stream));
  RS_HIP(hipMemcpy(out, ln->d_progress, 8 * rs::kCohortSlots * sizeof(int), hipMemcpyDeviceToHost));
  return RS_OK;
}

int rs_info(rs_ctx* c, int64_t* bk_bytes, int64_t* ksk_bytes, int32_t* wpb, int32_t* cus) {
  if (!c) return fail(RS_ERR_INVALID, "null context");
  if (bk_bytes) *bk_bytes = (int64_t)c->bk_bytes;
  if (ksk_bytes) *ksk_bytes = (int64_t)c->ksk_bytes;
  if (wpb) {   // of the last blind rotation on the default stream (8 = what a full-chip batch would use, before any launch)
    Lane* ln = nullptr;
    *wpb = (lane_of(c, nullptr, &ln) == RS_OK && ln->last.form >= 0) ? ln->last.waves_per_block : 8;
  }
  if (cus) *cus = c->num_cus;
  return RS_OK;
}

}  // extern "C"
