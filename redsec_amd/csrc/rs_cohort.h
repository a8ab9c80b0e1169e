// rs_cohort.h -- XCD cohorts: keeping the workgroups that share an L2 within reach of each other (device code; used by the
// two lock-step kernels of rs_bootstrap.hip. The general ring kernels of rs_general.hip sit at the 8-XCD traffic floor without it).
#pragma once

#include <hip/hip_runtime.h>

#include "rs_kernels.h"

namespace rs {

// XCD cohorts. The lock-step split kernel streams 2 x the key bytes of the unsplit one; the workgroups of an XCD share them
// through that XCD's 4 MB L2 only while they are within a few CMUX steps of each other (one step = 2 l x 2 half-rows of 16 KB:
// 192 KB default-128, 640 KB REDsec set), and nothing kept them there: counter traffic of a 65,536-gate launch was 58 GB in
// round 2 and 106 GB in round 3 against 32 GB if every XCD fetched every half-row once per round. So every `every` steps wave 0
// of a workgroup publishes its step count and looks at its XCD's table (workgroups are dealt to the XCDs round-robin:
// xcd = blockIdx.x & 7); more than `lag` steps ahead of the slowest one it waits -- bounded: at most kCohortPolls polls, so a
// workgroup that is not resident (a shared GPU) or a stale table can delay a launch but never hang it. The other waves of the
// workgroup notice nothing: they wait for wave 0 at the next publish barrier, as they do anyway.
constexpr int kCohortPolls = 48;
// Two halves, one CMUX step apart, so that the table's L2 round trip is never waited for: at step g (this workgroup's running
// count of CMUX steps) wave 0 publishes g and REQUESTS its XCD's row (lane L: entry L; kCohortSlots = 64 = one wavefront) every
// `every` (1 or 2) steps; one step later it looks at what came back and only if some entry lags polls synchronously (bounded).
//
// The whole step is ONE asm block that owns nothing between calls: the lock-step kernels sit at 256 VGPRs with their SGPRs full
// too, and written in C++ the protocol's state (table pointer, lag, period, the row in flight, lane offsets the compiler hoisted
// out of the loop) put spills -- and in front of each an `s_waitcnt vmcnt(0)` that drained the key prefetch -- into every CMUX
// step: 6-10 GB of scratch writes per 65,536-gate launch of the split kernel (profiles/r04/pmc/), and it is what made cohorts a
// loss in the general ring kernels. Here the parameters come from the kernel-argument segment by scalar loads at each use, the
// row in flight lands in LDS (`mail`: kCohortSlots ints of the workgroup, touched by wave 0 only) by an LDS-DMA load, "armed" and
// the count published last follow from g (armed: g > 0 and (g - 1) % every == 0; published: g - 1), and every register is a
// temporary of the block. `sc1`: agent scope (the table is read from and written to the L2, as relaxed agent-scope atomics are).
// The kernels' own vmcnt waits (one per key row, many per step) have long covered the row when the check reads the mailbox; a
// stale mailbox could only send the wave into the poll loop, which loads afresh and waits.
// Must be called by all 64 lanes of wave 0 (EXEC full). Args: the kernel's ONE by-value argument (kernarg offset 0).
template <class Args>
__device__ __forceinline__ void cohort_step(int g, int* mail) {
  const unsigned long long ka = (unsigned long long)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  const unsigned mail_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)mail;
  unsigned long long row, ex;
  unsigned ev, lag, m, tmp, xoff, t0, t1, t2;
  asm volatile(
      "s_load_dwordx2 %[row], %[ka], %[o_prog]\n\t"
      "s_load_dword %[ev], %[ka], %[o_every]\n\t"
      "s_load_dword %[lag], %[ka], %[o_lag]\n\t"
      "s_and_b32 %[xoff], %[bid], 7\n\t"
      "s_lshl_b32 %[xoff], %[xoff], 8\n\t"                 // byte offset of this XCD's row: 64 ints
      "v_mbcnt_lo_u32_b32 %[t2], -1, 0\n\t"
      "v_mbcnt_hi_u32_b32 %[t2], -1, %[t2]\n\t"
      "v_lshlrev_b32 %[t2], 2, %[t2]\n\t"                  // t2 = lane * 4
      "v_add_u32 %[t1], %[xoff], %[t2]\n\t"                // t1 = byte offset of table entry (xcd, lane)
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_cmp_eq_u64 %[row], 0\n\t"
      "s_cbranch_scc1 .Lcoh_done_%=\n\t"                   // no table: free-running
      "s_sub_u32 %[ev], %[ev], 1\n\t"                      // every - 1: the period's mask
      "s_sub_u32 %[m], %[g], 1\n\t"                        // what this workgroup published last, if it did
      "s_cmp_lt_i32 %[m], 0\n\t"
      "s_cbranch_scc1 .Lcoh_post_%=\n\t"
      "s_and_b32 %[tmp], %[m], %[ev]\n\t"
      "s_cmp_lg_u32 %[tmp], 0\n\t"
      "s_cbranch_scc1 .Lcoh_post_%=\n\t"
      "v_add_u32 %[t0], %[mail], %[t2]\n\t"
      "ds_read_b32 %[t0], %[t0]\n\t"                       // the row requested a step ago
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_add_u32 %[t0], %[lag], %[t0]\n\t"
      "v_cmp_gt_i32 vcc, %[m], %[t0]\n\t"                  // some workgroup of the XCD more than lag behind?
      "s_cbranch_vccz .Lcoh_post_%=\n\t"
      "s_movk_i32 %[tmp], %[polls]\n"
      ".Lcoh_poll_%=:\n\t"
      "s_sleep 64\n\t"
      "global_load_dword %[t0], %[t1], %[row] sc1\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "v_add_u32 %[t0], %[lag], %[t0]\n\t"
      "v_cmp_gt_i32 vcc, %[m], %[t0]\n\t"
      "s_cbranch_vccz .Lcoh_post_%=\n\t"
      "s_sub_u32 %[tmp], %[tmp], 1\n\t"
      "s_cmp_lg_u32 %[tmp], 0\n\t"
      "s_cbranch_scc1 .Lcoh_poll_%=\n"
      ".Lcoh_post_%=:\n\t"
      "s_and_b32 %[tmp], %[g], %[ev]\n\t"
      "s_cmp_lg_u32 %[tmp], 0\n\t"
      "s_cbranch_scc1 .Lcoh_done_%=\n\t"
      "s_lshr_b32 %[tmp], %[bid], 3\n\t"
      "s_lshl_b32 %[tmp], %[tmp], 2\n\t"
      "s_add_u32 %[tmp], %[tmp], %[xoff]\n\t"              // byte offset of this workgroup's own entry
      "v_mov_b32 %[t0], %[tmp]\n\t"
      "v_mov_b32 %[t2], %[g]\n\t"
      "s_mov_b64 %[ex], exec\n\t"
      "s_mov_b64 exec, 1\n\t"
      "global_store_dword %[t0], %[t2], %[row] sc1\n\t"
      "s_mov_b64 exec, %[ex]\n\t"
      "s_mov_b32 %[tmp], m0\n\t"
      "s_mov_b32 m0, %[mail]\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %[t1], %[row] sc1\n\t"        // lane L: entry (xcd, L) -> mail[L]
      "s_mov_b32 m0, %[tmp]\n"
      ".Lcoh_done_%=:"
      : [row] "=&s"(row), [ex] "=&s"(ex), [ev] "=&s"(ev), [lag] "=&s"(lag), [m] "=&s"(m), [tmp] "=&s"(tmp), [xoff] "=&s"(xoff),
        [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)
      : [ka] "s"(ka), [g] "s"(g), [mail] "s"(mail_lds), [bid] "s"((unsigned)blockIdx.x),
        [o_prog] "n"(__builtin_offsetof(Args, progress)), [o_every] "n"(__builtin_offsetof(Args, cohort_every)),
        [o_lag] "n"(__builtin_offsetof(Args, cohort_lag)), [polls] "n"(kCohortPolls)
      : "memory", "vcc", "scc");
}
// a finished workgroup holds nobody back: its entry becomes kCohortGone + the CMUX steps it walked (beyond every step count a
// launch can reach; the sum is also what rs_debug_cohort_table lets a test read back: which entries a launch wrote, and with
// what). Wave 0, EXEC full.
constexpr int kCohortGone = 0x40000000;
template <class Args>
__device__ __forceinline__ void cohort_leave(int steps) {
  const unsigned long long ka = (unsigned long long)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned long long row, ex;
  unsigned tmp, xoff, t0, t2;
  asm volatile(
      "s_load_dwordx2 %[row], %[ka], %[o_prog]\n\t"
      "s_and_b32 %[xoff], %[bid], 7\n\t"
      "s_lshl_b32 %[xoff], %[xoff], 8\n\t"
      "s_lshr_b32 %[tmp], %[bid], 3\n\t"
      "s_lshl_b32 %[tmp], %[tmp], 2\n\t"
      "s_add_u32 %[tmp], %[tmp], %[xoff]\n\t"
      "v_mov_b32 %[t0], %[tmp]\n\t"
      "v_mov_b32 %[t2], %[gone]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_cmp_eq_u64 %[row], 0\n\t"
      "s_cbranch_scc1 .Lcoh_left_%=\n\t"
      "s_mov_b64 %[ex], exec\n\t"
      "s_mov_b64 exec, 1\n\t"
      "global_store_dword %[t0], %[t2], %[row] sc1\n\t"
      "s_mov_b64 exec, %[ex]\n"
      ".Lcoh_left_%=:"
      : [row] "=&s"(row), [ex] "=&s"(ex), [tmp] "=&s"(tmp), [xoff] "=&s"(xoff), [t0] "=&v"(t0), [t2] "=&v"(t2)
      : [ka] "s"(ka), [bid] "s"((unsigned)blockIdx.x), [gone] "s"(kCohortGone | steps), [o_prog] "n"(__builtin_offsetof(Args, progress))
      : "memory", "scc");
}

}  // namespace rs
