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
// (0, 1 or 2 bytes, or a raw byte), scan the counts and store the bytes: the serial chain is 12 scalar instructions per entry.
//
// Two properties of the memory system shape the hand-over (tools/rans_sload2_bench.hip, rans_sload3_bench.hip):
//  * a store to a line the L2 does not hold does not bring it in: records laid into fresh memory come back from HBM (116 cycles
//    per entry instead of 57).  The records therefore live in a small RING per block (16 trips, 16 KB) whose lines stay in the
//    L2 after the first lap;
//  * the scalar data cache is not coherent with those stores: a lap later it may still hold the old records (a 4 KB or 16 KB
//    ring without an invalidate ends in a wrong state, the tool shows it).  Every trip starts with s_dcache_inv (2 cycles per
//    entry; each line is read once per lap anyway, there is nothing to lose).
// The scalar data cache is filled at ~0.3 TB/s for the whole card (tools/rans_sload_bench.hip: 1024 waves streaming 32 bytes per
// entry fall to 215-363 cycles per entry): this form is for calls with FEW blocks - the per-frame calls, I+P batches (a P-frame
// is one short block) - and the host picks it by block count (scpr_amd.hip); a batch of key frames (1800 blocks) stays with k_rans.
#pragma once
#include "scpr_kernels.hpp"
#include "scpr_wave.hpp"

namespace scpr {

constexpr int RANS_S_TRIP = 64;  // entries per trip: one lane of the hand-over register each
constexpr int RANS_S_RING = 16;  // trips of records per block (a power of two)

// the entry's record for the scalar step; a raw byte (freq 0) and the padding of the last trip (0xFFFFFFFF) leave the state as it is.
// (The reciprocal is fetched a trip before the record is laid: rans_rcp_of; a lone wave pays every load it waits for.)
__device__ __forceinline__ uint2 rans_rcp_of(u32 v, const RansRcp* __restrict__ rcp_g) {  // { reciprocal, shift | pad << 16 }
  const u32 fr = v & 0xFFFFu;
  return ((const uint2*)rcp_g)[fr - 1u < (u32)kProbScale ? fr : 1u];
}
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

// one entry: state in %[x], its record in four scalar registers, the state before the step to lane K of %[vo]
#define SCPR_RS_ENT(XM, RC, ML, BS, K)                                                                                          \
  "s_cmp_ge_u32 %[x], s" #XM "\n\ts_cselect_b32 s20, 8, 0\n\tv_writelane_b32 %[vo], %[x], " #K "\n\ts_lshr_b32 s21, %[x], s20\n\t"     \
  "s_cmp_ge_u32 s21, s" #XM "\n\ts_cselect_b32 s20, 8, 0\n\ts_lshr_b32 s23, s" #BS ", 5\n\ts_lshr_b32 s21, s21, s20\n\t"           \
  "s_mul_hi_u32 s22, s21, s" #RC "\n\ts_lshr_b32 s22, s22, s" #BS "\n\ts_mul_i32 s22, s22, s" #ML "\n\ts_add_u32 s21, s21, s23\n\t"  \
  "s_add_u32 %[x], s22, s21\n\t"
#define SCPR_RS_SET_A(K0, K1, K2, K3, K4, K5, K6, K7) SCPR_RS_ENT(36, 37, 38, 39, K0) SCPR_RS_ENT(40, 41, 42, 43, K1) SCPR_RS_ENT(44, 45, 46, 47, K2) SCPR_RS_ENT(48, 49, 50, 51, K3) SCPR_RS_ENT(52, 53, 54, 55, K4) SCPR_RS_ENT(56, 57, 58, 59, K5) SCPR_RS_ENT(60, 61, 62, 63, K6) SCPR_RS_ENT(64, 65, 66, 67, K7)
#define SCPR_RS_SET_B(K0, K1, K2, K3, K4, K5, K6, K7) SCPR_RS_ENT(68, 69, 70, 71, K0) SCPR_RS_ENT(72, 73, 74, 75, K1) SCPR_RS_ENT(76, 77, 78, 79, K2) SCPR_RS_ENT(80, 81, 82, 83, K3) SCPR_RS_ENT(84, 85, 86, 87, K4) SCPR_RS_ENT(88, 89, 90, 91, K5) SCPR_RS_ENT(92, 93, 94, 95, K6) SCPR_RS_ENT(96, 97, 98, 99, K7)
#define SCPR_RS_LOAD_A(OFF0, OFF1) "s_waitcnt lgkmcnt(0)\n\ts_load_dwordx16 s[36:51], %[p], " #OFF0 "\n\ts_load_dwordx16 s[52:67], %[p], " #OFF1 "\n\t"
#define SCPR_RS_LOAD_B(OFF0, OFF1) "s_waitcnt lgkmcnt(0)\n\ts_load_dwordx16 s[68:83], %[p], " #OFF0 "\n\ts_load_dwordx16 s[84:99], %[p], " #OFF1 "\n\t"
#define SCPR_RS_CLOBBERS "s20", "s21", "s22", "s23", "scc", "memory", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99"

// one wave per block, four per workgroup.  rec: the records' rings (RANS_S_RING * RANS_S_TRIP records per block); everything else as k_rans.
__global__ __launch_bounds__(256) void k_rans_s(const u32* __restrict__ entries, const RansBlock* __restrict__ blocks, const RansRcp* __restrict__ rcp_g, uint4* __restrict__ rec,
                                               int nblocks, u8* __restrict__ scratch, u32* __restrict__ blksize, const u32* __restrict__ err) {
  // four blocks per workgroup: its waves go to the four SIMDs of a CU, and two such waves on ONE SIMD would take turns at the
  // scalar unit (tools/exp_rans.py: 3.7 ms with one wave per SIMD, 6.3 with two, 9.0 with three - whatever the number of blocks)
  const int b = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6))), lane = threadIdx.x & 63;
  if (b >= nblocks) return;
  if (*err & 32u) {  // (see k_rans)
    if (lane == 0) blksize[b] = 0;
    return;
  }
  const RansBlock blk = blocks[b];
  const int len = (int)blk.len, trips = (len + RANS_S_TRIP - 1) / RANS_S_TRIP;
  const u32* e = entries + blk.begin;
  uint4* const r0 = rec + (size_t)b * (RANS_S_RING * RANS_S_TRIP);
  auto slot = [&](int t) { return r0 + (size_t)(t & (RANS_S_RING - 1)) * RANS_S_TRIP; };
  u8* const base = scratch + (size_t)b * RANS_SCRATCH;
  // entry of trip t in this lane, in coding order (last entry first); past the block's first entry: padding
  auto entry_of = [&](int t) -> u32 {
    const int i = len - 1 - (t * RANS_S_TRIP + lane);
    return i >= 0 ? e[i] : 0xFFFFFFFFu;
  };
  // the records of trips 0 and 1 before anything is coded; from then on trip t lays the records of trip t + 2 (their
  // reciprocals were asked for during trip t - 1, their entries during trip t - 2) while it runs
  u32 ev0 = entry_of(0), ev1 = entry_of(1), ev2 = entry_of(2), ev3 = entry_of(3);
  slot(0)[lane] = rans_record_s(ev0, rans_rcp_of(ev0, rcp_g));
  slot(1)[lane] = rans_record_s(ev1, rans_rcp_of(ev1, rcp_g));
  uint2 rc2 = rans_rcp_of(ev2, rcp_g);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(ev2), "+v"(ev3), "+v"(rc2.x), "+v"(rc2.y) : : "memory");  // (nothing is on its way when the loop is entered: see the wait inside it)
  u32 x = kRansL, off = RANS_SCRATCH;
  for (int t = 0; t < trips; t++) {
    slot(t + 2)[lane] = rans_record_s(ev2, rc2);  // (read last by trip t - 14)
    uint2 rc3 = rans_rcp_of(ev3, rcp_g);
    u32 ev4 = entry_of(t + 4);
    u32 vout = 0;
    const uint4* p = slot(t);
  asm volatile("s_dcache_inv\n\t" SCPR_RS_LOAD_A(0x0, 0x40)
               SCPR_RS_LOAD_B(0x80, 0xc0) SCPR_RS_SET_A(0, 1, 2, 3, 4, 5, 6, 7)
               SCPR_RS_LOAD_A(0x100, 0x140) SCPR_RS_SET_B(8, 9, 10, 11, 12, 13, 14, 15)
               SCPR_RS_LOAD_B(0x180, 0x1c0) SCPR_RS_SET_A(16, 17, 18, 19, 20, 21, 22, 23)
               SCPR_RS_LOAD_A(0x200, 0x240) SCPR_RS_SET_B(24, 25, 26, 27, 28, 29, 30, 31)
               SCPR_RS_LOAD_B(0x280, 0x2c0) SCPR_RS_SET_A(32, 33, 34, 35, 36, 37, 38, 39)
               SCPR_RS_LOAD_A(0x300, 0x340) SCPR_RS_SET_B(40, 41, 42, 43, 44, 45, 46, 47)
               SCPR_RS_LOAD_B(0x380, 0x3c0) SCPR_RS_SET_A(48, 49, 50, 51, 52, 53, 54, 55)
               "s_waitcnt lgkmcnt(0)\n\t" SCPR_RS_SET_B(56, 57, 58, 59, 60, 61, 62, 63)
               : [x] "+s"(x), [vo] "+v"(vout) : [p] "s"(p) : SCPR_RS_CLOBBERS);
    // this trip's stores and loads were issued 64 steps ago: the records are in the L2 before a scalar load asks for them, and the
    // compiler's own wait for the loaded values lands HERE - not behind the byte stores below, whose way to memory it would pay
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ev4), "+v"(rc3.x), "+v"(rc3.y) : : "memory");
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
    ev0 = ev1, ev1 = ev2, ev2 = ev3, ev3 = ev4, rc2 = rc3;
  }
  if (lane == 0) {  // RansEncFlush, rans_byte.h:90-102
    u8* q = base + off - 4;
    q[0] = (u8)x, q[1] = (u8)(x >> 8), q[2] = (u8)(x >> 16), q[3] = (u8)(x >> 24);
    blksize[b] = (u32)RANS_SCRATCH - (off - 4);
  }
}

}  // namespace scpr
