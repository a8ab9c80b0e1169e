// rs_lds_plan.h -- WHERE the N = 1024 blind-rotation kernels put the data that changes hands between wavefronts in LDS.
//
// The kernels of rs_bootstrap.hip exchange data between the waves of a workgroup in three ways: partial column sums
// (cooperative forms), partials swapped between the two waves of a ciphertext through the idle key buffer (duo forms), and
// key (half-)rows that arrive by direct global->LDS loads into ring slots while other waves still read the previous slots.
// Every placement decision those protocols rest on is a function HERE, used by the kernels and -- on the host -- by the
// protocol model of rs_emulate.cpp (rs_emu_lds_protocol_conflicts), which replays each form's barrier epochs and counts
// pairs of accesses by DIFFERENT waves to overlapping bytes in one epoch with at least one write. A placement that breaks a
// protocol is therefore a failing CPU test (tests/test_emulator.py), not a one-in-ten GPU failure.
#pragma once

#include "rs_ntt.h"

namespace rs {

// ---- blind_rotate_coop8_kernel: 8 waves, one ciphertext ------------------------------------------------------------------
// Wave w runs on SIMD w & 3, and of the two waves of a SIMD (s and s + 4) the OLDER one (s) wins the issue arbitration: in the
// phase stamps of round 4 (profiles/r04/am_coop8_phase_stamps.json) the older wave ran its rows at 2,900 cycles each whatever
// its partner did, the younger one at 3,600-4,600 while they overlapped. So the extra rows go to the older waves: every SIMD
// then finishes its five rows (l = 10) together, instead of two SIMDs waiting for a younger wave that had three.
// Components alternate (wave & 1), j = wave >> 1 is the wave's place among the four of its component: j = 0, 1 are the older
// waves of SIMDs 0/1 and 2/3, j = 2, 3 the younger ones.
constexpr int kCoop8Waves = 8;
// The LISTED step (round 6; the deals with at most one row per wave, whose inverse waves carry no row: l < 4): the kernel lists
// the CMUX steps that are not the identity in LDS once, in its prologue ((i << 16) | bara) -- LWE dimensions up to
// kCoop8MaxSteps, the launcher's condition for the form (larger ones run the four- / two-wave forms) -- and builds the prepared
// rotated difference of a step with all 512 threads: thread t takes coefficients (t & 255) + 256 m, m < 4, of component t >> 8
// (waves 0-3: component 0, waves 4-7: component 1), whichever row the wave then transforms.
constexpr int kCoop8MaxSteps = 2048;
RS_HD constexpr int coop8_diff_comp(int wave) { return wave >> 2; }
RS_HD constexpr int coop8_diff_coeff(int thread, int m) { return (thread & 255) + 256 * m; }   // m < kCoop8DiffPerThread
constexpr int kCoop8DiffPerThread = 4;
// l >= 4 (the REDsec set: l = 10): the deal by wave age described above. l < 4 (default-128: l = 3, six rows for eight waves)
// keeps the deal of the kernel's first form -- waves 0-3 component 0, waves 4-7 component 1, the extra rows to the first waves of
// component 0 and the last of component 1, inverse transforms on the two waves without rows (3 and 4): measured 2.66 ms against
// 2.73 ms for the deal by age at 196 default-128 ciphertexts (profiles/r04/aw_*).
RS_HD constexpr bool coop8_by_age(int L) { return L >= 4; }
RS_HD constexpr bool coop8_listed(int L) { return !coop8_by_age(L); }
RS_HD constexpr int coop8_inv_a(int L) { return coop8_by_age(L) ? 6 : 3; }   // inverts column 0: a wave with the fewest rows
RS_HD constexpr int coop8_inv_b(int L) { return coop8_by_age(L) ? 7 : 4; }   // inverts column 1: likewise, on another SIMD
RS_HD constexpr int coop8_comp(int L, int wave) { return coop8_by_age(L) ? (wave & 1) : (wave >> 2); }
RS_HD constexpr int coop8_row_count(int L, int wave) {
  if (coop8_by_age(L)) return L / 4 + ((wave >> 1) < L % 4 ? 1 : 0);
  return L / 4 + ((wave >> 2) == 0 ? ((wave & 3) < L % 4 ? 1 : 0) : ((wave & 3) >= 4 - L % 4 ? 1 : 0));
}
RS_HD constexpr int coop8_row_first(int L, int wave) {
  const int j = coop8_by_age(L) ? (wave >> 1) : (wave & 3), rem = L % 4;
  if (coop8_by_age(L) || (wave >> 2) == 0) return j * (L / 4) + (j < rem ? j : rem);
  return j * (L / 4) + (j > 4 - rem ? j - (4 - rem) : 0);
}

// ---- blind_rotate_coops_kernel<G>: four sums (2 key halves x 2 columns), one owner wave each -----------------------------
template <int G> RS_HD constexpr int coops_owner(int sum) { return G == 4 ? sum : (sum & 1); }
// index of `sum` among the sums wave g does NOT own (its slots of s_part)
template <int G> RS_HD constexpr int coops_slot(int sum, int g) { return G == 4 ? (sum < g ? sum : sum - 1) : (sum >> 1); }

// ---- duo forms: 4 ciphertexts x 2 waves, partials swapped through the key buffer (8 x kN doubles = 64 KB) ----------------
RS_HD constexpr int duo_xchg_doubles(int wave) { return wave * kN; }   // offset of the wave's 8 KB window in the key buffer
RS_HD constexpr int duo_partner(int wave) { return wave ^ 1; }
// blind_rotate_duo_kernel: a "quad" = rows (comp, 2p + k) in slot 2 comp + k; wave fetches chunks [first, first + 8) of its slot
RS_HD constexpr int duo_quad_slot(int wave) { return wave >> 1; }
RS_HD constexpr int duo_quad_first_chunk(int wave) { return (wave & 1) * 8; }
// blind_rotate_duos_kernel: pair p = the half-rows of both components in slots 2 (p & 1) + comp; wave fetches 4 chunks
RS_HD constexpr int duos_pair_slot(int p, int comp) { return 2 * (p & 1) + comp; }
RS_HD constexpr int duos_fetch_comp(int wave) { return wave >> 2; }
RS_HD constexpr int duos_first_chunk(int wave) { return (wave & 3) * 4; }

// ---- lock-step rings ---------------------------------------------------------------------------------------------------
RS_HD constexpr int wg_ring_slot(long row) { return (int)(row & 1); }       // blind_rotate_wg_kernel: rows in pairs, two slots
RS_HD constexpr int wgs_ring_slot(long half_row) { return (int)(half_row % 3); }   // blind_rotate_wgs_kernel: three slots, two ahead

}  // namespace rs
