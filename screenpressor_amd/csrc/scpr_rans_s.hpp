// scpr_rans_s.hpp - byte-wise rANS with the coder state on the SCALAR unit: one wave per block of <= 131072 entries
// (RansMTCoder::writeBlock, ransmt.h:116-134; RansEncPut, rans_byte.h:59-87 in the reciprocal form of :199-240).
//
// k_rans (scpr_kernels.hpp) keeps 64 blocks in the lanes of one wave; its step is ~137 cycles per entry whatever the number of
// blocks, because a lone wave issues a dependent VECTOR instruction every ~8 cycles.  The same arithmetic on the scalar unit
// issues every ~4.5 (tools/rans_sload_bench.hip: 56 cycles per entry), but a scalar register can only be filled by a lane read
// (16-30 cycles each: tools/rans_scalar_bench.hip, 137 cycles per entry again) or by a SCALAR LOAD from memory.  So the entry's
// constants go through memory: the wave's own lanes turn 64 entries into 16-byte records { x_max, reciprocal, 4096 - freq,
// bias << 5 | shift }, store them (they stay in the L2 of the wave's own XCD), and two trips later the scalar unit reads them
// back four at a time (s_load_dwordx16), eight entries ahead of the step that uses them.  The state before every step is
// dropped into a lane of a vector register (v_writelane), and after the 64 steps the lanes work out what each step emitted
// (0, 1 or 2 bytes, or a raw byte), scan the counts and store the bytes: the serial chain is 9 scalar instructions per entry
// (12 for the sets of eight entries that hold a freq below 16, the only ones whose step can emit two bytes).
//
// Two properties of the memory system shape the hand-over (tools/rans_sload2_bench.hip, rans_sload3_bench.hip):
//  * a store to a line the L2 does not hold does not bring it in: records laid into fresh memory come back from HBM (116 cycles
//    per entry instead of 57).  The records therefore live in a small RING per block whose lines stay in the L2 after the first
//    lap: 4 trips, 4 KB (trip t reads slot t while it lays slot t + 2).  Round 4 had 16 trips: at 1020 blocks the rings (16.7 MB)
//    no longer stayed in the L2s and the stage took 4.8 ms instead of 3.7; with 4 trips it is 3.65 (tools/r5/rans_matrix.sh);
//  * the scalar data cache is not coherent with those stores: a lap later it may still hold the old records (a ring without an
//    invalidate ends in a wrong state, the tool shows it).  Every trip starts with s_dcache_inv (2 cycles per entry; each line is
//    read once per lap anyway, there is nothing to lose).  The loads' own GLC bit does the same and costs 60 % more time (5.98
//    against 3.65 ms: every load then waits for the L2).
// Neither property is specified, so the kernel does not rely on them for CORRECTNESS: every step is taken a second time by the
// lanes, from the entry itself (see the end of the trip), and a call in which one differs is coded again by k_rans.
// The scalar data cache is filled at ~0.3 TB/s for the whole card from HBM (tools/rans_sload_bench.hip); out of the L2-resident
// rings 1800 waves draw ~0.6 TB/s without slowing down.  What bounds the stage is the scalar unit's issue rate: one instruction
// per SIMD every four cycles, so a wave alone on its SIMD takes ~59 cycles per entry, two take ~97 each, three ~135 each
// (3.2 / 5.3 / 7.4 ms for a full block); the host picks this form up to kRansScalarMax blocks per call (scpr_amd.hip).
#pragma once
#include "scpr_kernels.hpp"
#include "scpr_wave.hpp"

namespace scpr {

constexpr int RANS_S_TRIP = 64;  // entries per trip: one lane of the hand-over register each
#ifndef SCPR_RANS_RING
#define SCPR_RANS_RING 4
#endif
constexpr int RANS_S_RING = SCPR_RANS_RING;  // trips of records per block (a power of two, >= 4: trip t lays the records of trip t + 2)

// the entry's record for the scalar step; a raw byte (freq 0) and the padding of the last trip (0xFFFFFFFF) leave the state as it is.
// (The reciprocal is fetched a trip before the record is laid: rans_rcp_of; a lone wave pays every load it waits for.)
__device__ __forceinline__ uint2 rans_rcp_of(u32 v, const RansRcp* __restrict__ rcp_g) {  // { reciprocal, shift | pad << 16 }
  const u32 fr = v & 0xFFFFu;
  return ((const uint2*)rcp_g)[fr - 1u < (u32)kProbScale ? fr : 1u];
}
typedef uint4 rans_rec_t;
__device__ __forceinline__ uint4 rans_record_s(u32 v, const uint2 r) {
  const u32 fr = v & 0xFFFFu, cf = v >> 16;
  const bool live = fr - 1u < (u32)kProbScale;
  uint4 o;
  o.x = live ? fr << 19 : 0xFFFFFFFFu;  // x_max = ((L >> 12) << 8) * freq; no state reaches 2^32 - 1
  o.y = live ? r.x : 0u;
  o.z = live ? (u32)kProbScale - fr : 0u;
  o.w = live ? ((cf + (r.y >> 16)) << 5) | (r.y & 0xFFFFu) : 0u;  // (a scalar shift reads the low five bits of its operand: the shift rides below the bias)
  return o;
}

// one entry: state in %[x], its record in four scalar registers, the state before the step to lane K of %[vo].
// SCPR_RS_ENT is the whole step (12 scalar instructions); SCPR_RS_ENTF leaves the second renormalisation test out (9): a state
// below 2^31 passes freq << 27 only when freq < 16, and the lanes know which sets of eight entries hold such a freq (1.1 % of a
// key frame's entries, 7 % of the sets) - the trip branches per set.
#define SCPR_RS_ENT(XM, RC, ML, BS, K)                                                                                          \
  "s_cmp_ge_u32 %[x], s" #XM "\n\ts_cselect_b32 s20, 8, 0\n\tv_writelane_b32 %[vo], %[x], " #K "\n\ts_lshr_b32 s21, %[x], s20\n\t"     \
  "s_cmp_ge_u32 s21, s" #XM "\n\ts_cselect_b32 s20, 8, 0\n\ts_lshr_b32 s23, s" #BS ", 5\n\ts_lshr_b32 s21, s21, s20\n\t"           \
  "s_mul_hi_u32 s22, s21, s" #RC "\n\ts_lshr_b32 s22, s22, s" #BS "\n\ts_mul_i32 s22, s22, s" #ML "\n\ts_add_u32 s21, s21, s23\n\t"  \
  "s_add_u32 %[x], s22, s21\n\t"
#define SCPR_RS_ENTF(XM, RC, ML, BS, K)                                                                                         \
  "s_cmp_ge_u32 %[x], s" #XM "\n\ts_cselect_b32 s20, 8, 0\n\tv_writelane_b32 %[vo], %[x], " #K "\n\ts_lshr_b32 s23, s" #BS ", 5\n\t"   \
  "s_lshr_b32 s21, %[x], s20\n\ts_mul_hi_u32 s22, s21, s" #RC "\n\ts_lshr_b32 s22, s22, s" #BS "\n\ts_mul_i32 s22, s22, s" #ML "\n\t" \
  "s_add_u32 s21, s21, s23\n\ts_add_u32 %[x], s22, s21\n\t"
#define SCPR_RS_SET_A(E, K0, K1, K2, K3, K4, K5, K6, K7) E(36, 37, 38, 39, K0) E(40, 41, 42, 43, K1) E(44, 45, 46, 47, K2) E(48, 49, 50, 51, K3) E(52, 53, 54, 55, K4) E(56, 57, 58, 59, K5) E(60, 61, 62, 63, K6) E(64, 65, 66, 67, K7)
#define SCPR_RS_SET_B(E, K0, K1, K2, K3, K4, K5, K6, K7) E(68, 69, 70, 71, K0) E(72, 73, 74, 75, K1) E(76, 77, 78, 79, K2) E(80, 81, 82, 83, K3) E(84, 85, 86, 87, K4) E(88, 89, 90, 91, K5) E(92, 93, 94, 95, K6) E(96, 97, 98, 99, K7)
#define SCPR_RS_LOAD_A(I) "s_waitcnt lgkmcnt(0)\n\ts_load_dwordx16 s[36:51], %[p], " SCPR_RS_OFF16A(I) SCPR_RS_GLC "\n\ts_load_dwordx16 s[52:67], %[p], " SCPR_RS_OFF16B(I) SCPR_RS_GLC "\n\t"
#define SCPR_RS_LOAD_B_NOWAIT(I) "s_load_dwordx16 s[68:83], %[p], " SCPR_RS_OFF16A(I) SCPR_RS_GLC "\n\ts_load_dwordx16 s[84:99], %[p], " SCPR_RS_OFF16B(I) SCPR_RS_GLC "\n\t"
#define SCPR_RS_LOAD_B(I) "s_waitcnt lgkmcnt(0)\n\t" SCPR_RS_LOAD_B_NOWAIT(I)
#define SCPR_RS_OFF16A(I) SCPR_RS_OFF16A_##I
#define SCPR_RS_OFF16B(I) SCPR_RS_OFF16B_##I
#define SCPR_RS_OFF16A_0 "0x0"
#define SCPR_RS_OFF16B_0 "0x40"
#define SCPR_RS_OFF16A_1 "0x80"
#define SCPR_RS_OFF16B_1 "0xc0"
#define SCPR_RS_OFF16A_2 "0x100"
#define SCPR_RS_OFF16B_2 "0x140"
#define SCPR_RS_OFF16A_3 "0x180"
#define SCPR_RS_OFF16B_3 "0x1c0"
#define SCPR_RS_OFF16A_4 "0x200"
#define SCPR_RS_OFF16B_4 "0x240"
#define SCPR_RS_OFF16A_5 "0x280"
#define SCPR_RS_OFF16B_5 "0x2c0"
#define SCPR_RS_OFF16A_6 "0x300"
#define SCPR_RS_OFF16B_6 "0x340"
#define SCPR_RS_OFF16A_7 "0x380"
#define SCPR_RS_OFF16B_7 "0x3c0"
// (s[36:67], the A set, is not in this list: it is an operand pinned to those registers - it carries the first eight records of
// the NEXT trip over the loop's back edge, see the kernel)
#define SCPR_RS_CLOBBERS "s20", "s21", "s22", "s23", "scc", "memory", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99"
// how the scalar loads get past a lap-old line of the scalar data cache: s_dcache_inv in front of the trip's loads, or
// (SCPR_RANS_GLC, experiment) the loads' own GLC bit
#ifdef SCPR_RANS_GLC
#define SCPR_RS_GLC " glc"
#define SCPR_RS_INV ""
#else
#define SCPR_RS_GLC ""
#define SCPR_RS_INV "s_dcache_inv\n\t"
#endif

// the first eight records of the next trip (slot at %[pn]), asked for while this trip's last eight steps are taken: a trip then
// starts on records that have arrived instead of waiting a scalar-load round trip (~10 % of a trip) for them
#define SCPR_RS_LOAD_NEXT SCPR_RS_INV "s_load_dwordx16 s[36:51], %[pn], 0x0" SCPR_RS_GLC "\n\ts_load_dwordx16 s[52:67], %[pn], 0x40" SCPR_RS_GLC "\n\t"
typedef u32 rans_sgpr16 __attribute__((ext_vector_type(16)));

// a set of eight entries: the short step unless the trip's mask has a bit of the set (the long form sits behind the trip)
#ifdef SCPR_RANS_NOFAST  // (A/B timing: every set takes the whole step)
#define SCPR_RS_TRY(M, BITS, S)
#else
#define SCPR_RS_TRY(M, BITS, S) "s_and_b32 s20, %[" #M "], " #BITS "\n\ts_cbranch_scc1 .Lrs_slow" #S "_%=\n\t"
#endif
#define SCPR_RS_JOIN(S) ".Lrs_join" #S "_%=:\n\t"
#define SCPR_RS_SLOW(S, SET) ".Lrs_slow" #S "_%=:\n\t" SET "s_branch .Lrs_join" #S "_%=\n\t"

// one wave per block, four per workgroup.  rec: the records' rings (RANS_S_RING * RANS_S_TRIP records per block); everything else as k_rans.
// Four blocks per workgroup, TWO waves per block.  Waves 0-3 are the chains (one per SIMD of the CU): entries in, records out,
// the 64 scalar steps of a trip, and the states before the steps dropped into LDS.  Waves 4-7 are their writers, a trip behind:
// they turn the states into bytes (0, 1 or 2 per step, or the raw byte), store them, and take every step again from the entry
// itself (the check below).  Round 5: with everything in one wave the ~100 vector instructions of a trip's tail sat in the
// chain's own in-order instruction stream (3.44 ms for a full block alone on its SIMD, 5.42 with two blocks per SIMD); in a
// second wave they issue beside the chain's scalar instructions.  The two meet at ONE workgroup barrier per trip (a wave that
// waits at a barrier issues nothing, unlike one that polls), so the four chains of a workgroup run their trips in step and all
// eight waves run the trips of the workgroup's longest block - a shorter block's surplus trips are padding, which leaves the
// state alone and emits nothing.
template <bool POISON>  // POISON: the tests' instance (scpr_debug_inject 3), which leaves the records of trip `poison_trip` unlaid
__global__ __launch_bounds__(512) void k_rans_s(const u32* __restrict__ entries, const RansBlock* __restrict__ blocks, const RansRcp* __restrict__ rcp_g, rans_rec_t* __restrict__ rec,
                                               int nblocks, u8* __restrict__ scratch, u32* __restrict__ blksize, u32* __restrict__ err, int poison_trip) {
  __shared__ u32 s_state[4][2][RANS_S_TRIP];  // chain -> writer: the state before each step of a trip, two trips
  __shared__ u32 s_after[4][2];               //                  and after its last step
  __shared__ int s_trips[4];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), cw = wv & 3, lane = threadIdx.x & 63;
  const bool writer = wv >= 4;
  const int b = (int)blockIdx.x * 4 + cw;
  const bool none = b >= nblocks || (*err & 32u) != 0;  // (bit 5: see k_rans - nothing to code, the block is given the size 0)
  RansBlock blk{0, 0};
  if (!none) blk = blocks[b];
  const int len = (int)blk.len;
  if (!writer && lane == 0) s_trips[cw] = (len + RANS_S_TRIP - 1) / RANS_S_TRIP;
  __syncthreads();
  const int trips = max(max(s_trips[0], s_trips[1]), max(s_trips[2], s_trips[3]));
  const u32* e = entries + blk.begin;
  // entry of trip t in this lane, in coding order (last entry first); past the block's first entry: padding
  auto entry_of = [&](int t) -> u32 {
    const int i = len - 1 - (t * RANS_S_TRIP + lane);
    return i >= 0 ? e[i] : 0xFFFFFFFFu;
  };
  if (!writer) {
    __builtin_amdgcn_s_setprio(3);  // the chains in front of the writers at their SIMD's issue: 5.30 -> 5.21 ms with two blocks to a SIMD (tools/exp_rans.py), nothing with one
    rans_rec_t* const r0 = rec + (size_t)(none ? 0 : b) * (RANS_S_RING * RANS_S_TRIP);
    auto slot = [&](int t) { return r0 + (size_t)(t & (RANS_S_RING - 1)) * RANS_S_TRIP; };
    // the records of trips 0 and 1 before anything is coded; from then on trip t lays the records of trip t + 2 (their
    // reciprocals were asked for during trip t - 1, their entries during trip t - 2) while it runs
    u32 ev0 = entry_of(0), ev1 = entry_of(1), ev2 = entry_of(2), ev3 = entry_of(3);
    if (!none) {
      slot(0)[lane] = rans_record_s(ev0, rans_rcp_of(ev0, rcp_g));
      slot(1)[lane] = rans_record_s(ev1, rans_rcp_of(ev1, rcp_g));
    }
    uint2 rc2 = rans_rcp_of(ev2, rcp_g);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ev2), "+v"(ev3), "+v"(rc2.x), "+v"(rc2.y) : : "memory");  // (nothing is on its way when the loop is entered: see the wait inside it)
    u32 x = kRansL;
    rans_sgpr16 alo, ahi;  // s[36:51], s[52:67]: the A set of the trip's scalar code, alive from one trip to the next
    {
      const rans_rec_t* pn = slot(0);
      if (!none) asm volatile(SCPR_RS_LOAD_NEXT : "={s[36:51]}"(alo), "={s[52:67]}"(ahi) : [pn] "s"(pn) : "memory");
    }
    for (int t = 0; t < trips; t++) {
      // (read last by trip t + 2 - RANS_S_RING.  poison_trip, tests only: the records of one trip are NOT laid - the scalar unit
      // then reads what the ring held a lap before, which is what a stale line of the scalar data cache would hand it)
      if (!none && (!POISON || t + 2 != poison_trip)) slot(t + 2)[lane] = rans_record_s(ev2, rc2);
      uint2 rc3 = rans_rcp_of(ev3, rcp_g);
      u32 ev4 = entry_of(t + 4);
      u32 vout = 0;
      const rans_rec_t* p = slot(t);
      const rans_rec_t* pn = slot(t + 1);
      // which sets of eight hold an entry whose freq is below 16 (the only ones a second byte can come from)
      const u64 two = __ballot((ev0 & 0xFFFFu) - 1u < 15u);
      const u32 mlo = (u32)two, mhi = (u32)(two >> 32);
#ifdef SCPR_RANS_NOFAST
#define SCPR_RS_E SCPR_RS_ENT
#else
#define SCPR_RS_E SCPR_RS_ENTF
#endif
      if (none) __syncthreads();  // (a wave without a block keeps the barriers' company and nothing else: its ring was never laid)
      else
      // The trip begins with the wait every trip needs anyway - for its first records, asked for a trip ago - and that wait also
      // covers the LDS stores that handed the previous trip to the writer: the workgroup's barrier sits right behind it, inside
      // the scalar code.  (__syncthreads() after those stores would wait for the scalar loads just issued as well: LDS and
      // scalar memory share one counter.)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\t" SCPR_RS_LOAD_B_NOWAIT(1) SCPR_RS_TRY(mlo, 0xff, 0) SCPR_RS_SET_A(SCPR_RS_E, 0, 1, 2, 3, 4, 5, 6, 7) SCPR_RS_JOIN(0)
                   SCPR_RS_LOAD_A(2) SCPR_RS_TRY(mlo, 0xff00, 1) SCPR_RS_SET_B(SCPR_RS_E, 8, 9, 10, 11, 12, 13, 14, 15) SCPR_RS_JOIN(1)
                   SCPR_RS_LOAD_B(3) SCPR_RS_TRY(mlo, 0xff0000, 2) SCPR_RS_SET_A(SCPR_RS_E, 16, 17, 18, 19, 20, 21, 22, 23) SCPR_RS_JOIN(2)
                   SCPR_RS_LOAD_A(4) SCPR_RS_TRY(mlo, 0xff000000, 3) SCPR_RS_SET_B(SCPR_RS_E, 24, 25, 26, 27, 28, 29, 30, 31) SCPR_RS_JOIN(3)
                   SCPR_RS_LOAD_B(5) SCPR_RS_TRY(mhi, 0xff, 4) SCPR_RS_SET_A(SCPR_RS_E, 32, 33, 34, 35, 36, 37, 38, 39) SCPR_RS_JOIN(4)
                   SCPR_RS_LOAD_A(6) SCPR_RS_TRY(mhi, 0xff00, 5) SCPR_RS_SET_B(SCPR_RS_E, 40, 41, 42, 43, 44, 45, 46, 47) SCPR_RS_JOIN(5)
                   SCPR_RS_LOAD_B(7) SCPR_RS_TRY(mhi, 0xff0000, 6) SCPR_RS_SET_A(SCPR_RS_E, 48, 49, 50, 51, 52, 53, 54, 55) SCPR_RS_JOIN(6)
                   "s_waitcnt lgkmcnt(0)\n\t" SCPR_RS_LOAD_NEXT SCPR_RS_TRY(mhi, 0xff000000, 7) SCPR_RS_SET_B(SCPR_RS_E, 56, 57, 58, 59, 60, 61, 62, 63) SCPR_RS_JOIN(7)
#ifndef SCPR_RANS_NOFAST
                   "s_branch .Lrs_end_%=\n\t"
                   SCPR_RS_SLOW(0, SCPR_RS_SET_A(SCPR_RS_ENT, 0, 1, 2, 3, 4, 5, 6, 7)) SCPR_RS_SLOW(1, SCPR_RS_SET_B(SCPR_RS_ENT, 8, 9, 10, 11, 12, 13, 14, 15))
                   SCPR_RS_SLOW(2, SCPR_RS_SET_A(SCPR_RS_ENT, 16, 17, 18, 19, 20, 21, 22, 23)) SCPR_RS_SLOW(3, SCPR_RS_SET_B(SCPR_RS_ENT, 24, 25, 26, 27, 28, 29, 30, 31))
                   SCPR_RS_SLOW(4, SCPR_RS_SET_A(SCPR_RS_ENT, 32, 33, 34, 35, 36, 37, 38, 39)) SCPR_RS_SLOW(5, SCPR_RS_SET_B(SCPR_RS_ENT, 40, 41, 42, 43, 44, 45, 46, 47))
                   SCPR_RS_SLOW(6, SCPR_RS_SET_A(SCPR_RS_ENT, 48, 49, 50, 51, 52, 53, 54, 55)) SCPR_RS_SLOW(7, SCPR_RS_SET_B(SCPR_RS_ENT, 56, 57, 58, 59, 60, 61, 62, 63))
                   ".Lrs_end_%=:\n\t"
#endif
                   : [x] "+s"(x), [vo] "+v"(vout), "+{s[36:51]}"(alo), "+{s[52:67]}"(ahi) : [p] "s"(p), [pn] "s"(pn), [mlo] "s"(mlo), [mhi] "s"(mhi) : SCPR_RS_CLOBBERS);
      // this trip's stores and loads were issued 64 steps ago: the records are in the L2 before a scalar load asks for them, and the
      // compiler's own wait for the loaded values lands HERE, where everything has long arrived
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(ev4), "+v"(rc3.x), "+v"(rc3.y) : : "memory");
      // trip t is the writer's at the NEXT barrier (the start of trip t + 1, or the one behind the loop); this half of the buffers is
      // written again at the end of trip t + 2, behind the barrier after that, which the writer reaches when it is done with trip t
      s_state[cw][t & 1][lane] = vout;
      if (lane == 0) s_after[cw][t & 1] = x;
      ev0 = ev1, ev1 = ev2, ev2 = ev3, ev3 = ev4, rc2 = rc3;
    }
    __syncthreads();
    return;
  }
  // ---- the writer ----
  u8* const base = scratch + (size_t)(none ? 0 : b) * RANS_SCRATCH;
  u32 ev0 = entry_of(0);
  uint2 rc0 = rans_rcp_of(ev0, rcp_g);
  u32 off = RANS_SCRATCH, x = kRansL;
  u64 wrong = 0;
  __syncthreads();  // (the barrier at the start of the chains' trip 0: nothing to take over yet)
  for (int t = 0; t < trips; t++) {
    const u32 ev1 = entry_of(t + 1);  // (on their way while this trip is waited for and written)
    const uint2 rc1 = rans_rcp_of(ev1, rcp_g);
    __syncthreads();
    const u32 vout = s_state[cw][t & 1][lane];
    x = s_after[cw][t & 1];
    // what the 64 steps emitted: the renormalisation's bytes (the low bytes of the state before the step, first byte at the
    // higher address: the block is written from its end), or the raw byte itself
    const u32 fr = ev0 & 0xFFFFu, xm = fr << 19;
    const bool live = fr - 1u < (u32)kProbScale, raw = fr == 0u;
    const int n = live ? (int)(vout >= xm) + (int)((vout >> 8) >= xm) : (raw ? 1 : 0);
    const u32 b0 = raw ? (ev0 >> 16) & 255u : vout & 255u, b1 = (vout >> 8) & 255u;
    const int incl = wave_incl_scan(n);
    const u32 o = off - (u32)(incl - n);
    if (n >= 1) base[o - 1] = (u8)b0;
    if (n == 2) base[o - 2] = (u8)b1;
    off -= rdl((u32)incl, 63);
    // The hand-over is CHECKED, not trusted: every lane takes its entry's step again, from the entry itself and the state the
    // scalar unit started that step from, and must arrive at the state the next step started from (lane 63: the state the
    // trip ended with).  A record that did not reach the scalar unit as it was laid (the two measured, unspecified properties
    // of the memory system at the top of this file) shows here unless it changes nothing; the host then codes the call's blocks
    // again with k_rans (scpr_amd.hip).
#ifndef SCPR_RANS_NOCHECK  // (A/B timing only)
    {
      const u32 xr = live ? vout >> (8 * n) : vout;
      const u32 q = __umulhi(xr, rc0.x) >> (rc0.y & 31u);
      const u32 want = live ? xr + (ev0 >> 16) + (rc0.y >> 16) + q * ((u32)kProbScale - fr) : vout;
      const u32 next = (u32)__builtin_amdgcn_update_dpp((int)x, (int)vout, 0x130, 0xf, 0xf, false);  // wave_shl:1 - lane i reads lane i + 1, lane 63 keeps x
      wrong |= __ballot(want != next);
    }
#endif
    ev0 = ev1, rc0 = rc1;
  }
  if (none) {
    if (b < nblocks && lane == 0) blksize[b] = 0;
    return;
  }
  if (wrong && lane == 0) atomicOr(err, 64u);
  if (lane == 0) {  // RansEncFlush, rans_byte.h:90-102
    u8* q = base + off - 4;
    q[0] = (u8)x, q[1] = (u8)(x >> 8), q[2] = (u8)(x >> 16), q[3] = (u8)(x >> 24);
    blksize[b] = (u32)RANS_SCRATCH - (off - 4);
  }
}

}  // namespace scpr
