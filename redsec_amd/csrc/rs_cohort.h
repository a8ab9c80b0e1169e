// rs_cohort.h -- XCD cohorts: keeping the workgroups that share an L2 within reach of each other (device code; used by the
// lock-step kernels of rs_bootstrap.hip and by the general ring kernels of rs_general.hip).
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
// Two halves, one CMUX step apart, so that the table's L2 round trip is never waited for: cohort_post publishes this workgroup's
// step count and REQUESTS its XCD's row (lane L: entry L; kCohortSlots = 64 = one wavefront); cohort_check, a step later, looks
// at what came back and only if some entry lags polls synchronously (bounded).
__device__ __forceinline__ int cohort_post(int* progress, int mine, int lane) {
  int* row = progress + (blockIdx.x & 7) * kCohortSlots;
  if (lane == 0) __hip_atomic_store(row + (blockIdx.x >> 3), mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __hip_atomic_load(row + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void cohort_check(int* progress, int seen, int mine, int lag, int lane) {
  if (__builtin_amdgcn_ballot_w64(seen + lag < mine) == 0) return;
  const int* row = progress + (blockIdx.x & 7) * kCohortSlots;
  for (int poll = 0; poll < kCohortPolls; ++poll) {
    __builtin_amdgcn_s_sleep(64);
    const int v = __hip_atomic_load(row + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__builtin_amdgcn_ballot_w64(v + lag < mine) == 0) break;
  }
}
// one call per CMUX step and workgroup (wave 0): check what the previous post brought back, post again every `every` steps
struct CohortState { int seen = 0x7f7f7f7f, mine = 0; bool armed = false; };
__device__ __forceinline__ void cohort_step(int* progress, int every, int lag, CohortState& st, long step, int i, int lane) {
  if (st.armed) { cohort_check(progress, st.seen, st.mine, lag, lane); st.armed = false; }
  if (i % every == 0) { st.mine = (int)step; st.seen = cohort_post(progress, st.mine, lane); st.armed = true; }
}
template <class Args>
__device__ __forceinline__ void cohort_step(const Args& a, CohortState& st, long step, int i, int lane) {
  cohort_step(a.progress, a.cohort_every, a.cohort_lag, st, step, i, lane);
}
__device__ __forceinline__ void cohort_leave(int* progress, int lane) {
  if (lane == 0) __hip_atomic_store(progress + (blockIdx.x & 7) * kCohortSlots + (blockIdx.x >> 3), 0x7f7f7f7f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


}  // namespace rs
