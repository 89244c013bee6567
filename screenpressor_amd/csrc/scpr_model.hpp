// Adaptive context models of the ScreenPressor v3/v4 bitstream, in the form the
// MI355X kernels use them.
//
// The reference keeps seven polymorphic per-context structures (inline lists,
// sorted small tables, a robin-hood hash table, a dense table:
// ans_contexts.h:98-998, ans_contexts.cpp:3-84 in the reference tree).  Only
// the intervals they hand to the coder are part of the format, so this file
// keeps the *semantic* state instead:
//
//   kinds 0-3  "no symbol seen twice": a 256-bit set + count        (all raw)
//   kinds 4-5  sorted table of <=4 / <=16 symbols with live counts
//   kind  6    dense 256-entry partition; absent symbols are 1<<fshift wide
//   kind  7    dense 256-entry partition, every symbol counted
//
// Kind 6 note: the reference stores intervals only for symbols it has met and
// derives an unmet symbol's interval from its nearest lower neighbour
// (ans_contexts.h:596-619).  Its tables always form a complete partition in
// which every unmet symbol is exactly 1<<fshift wide (creation :454-531, insert
// :621-638, rebuild :742-796), so holding cum[] for all 256 symbols is
// equivalent and turns that search into one load.
//
// This header holds the state layouts, the format constants and the coder's exact
// reciprocal; the wave-cooperative model code the kernels run is in scpr_wave.hpp.
// (A serial restatement over the same state, used only by the CPU unit tests,
// lives in tests/host_model_serial.hpp.)
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define SCPR_HD __host__ __device__ __forceinline__
#else
#define SCPR_HD inline
#endif

namespace scpr {

enum : int {
  kProbBits = 12,
  kProbScale = 4096,     // ans_contexts.h:66-67
  kStepSmall = 50,       // STEP_CX5 (also kind 4), :56
  kStepHash = 25,        // STEP_CX6, :57
  kStepDense = 16,       // STEP_CX7 / STEP_FX, :58-59
  kHashMaxSyms = 40,     // Cx6::MaxD6, :385
  kBlockEntries = 131072 // RansMTCoder::B, ransmt.h:38
};

struct Ivl {  // coder entry: freq == 0 -> raw byte in cum (ans_contexts.h:62-64)
  uint16_t freq, cum;
};

// ----------------------------------------------------------------- colour ---
struct DenseTab {  // kinds 6 and 7
  uint16_t freq[256], cum[256], cnt[256];
};

struct ColState {  // 64 bytes per (plane, context)
  uint8_t kind, maxpos, fshift, pad0;
  uint16_t d;      // distinct symbols (kinds 1-6)
  uint16_t total;  // kind 5: cached total (drifts by design, :334-338/:174-184); kinds 6/7: running count total
  uint32_t dense;  // arena index of the DenseTab (kinds 6/7)
  uint32_t pad1;
  union {
    uint32_t seen[8];  // kinds 1-3: symbols seen once; kind 6: symbols met so far
    struct {
      uint8_t sym[16];
      uint16_t fr[16];
    } s;  // kinds 4-5
  } u;
};
static_assert(sizeof(ColState) == 64, "ColState is one 64-byte line");

// ------------------------------------------------------------------- rANS ---
// ryg byte-wise rANS, 32-bit state, L = 2^23, 12-bit scale (rans_byte.h:47-146)
enum : uint32_t { kRansL = 1u << 23 };

// exact x / freq for x < 2^31 via a precomputed reciprocal: the algebra of
// RansEncSymbolInit (rans_byte.h:171-240), which the reference ships but does
// not call.  freq in [1, 4096].
struct RansRcp {
  uint32_t rcp;    // ceil(2^(31+shift) / freq) low 32 bits
  uint16_t shift;  // ceil(log2(freq)) - 1 (0 for freq <= 2)
  uint16_t pad;    // 4095 for freq 1 (its quotient comes out as x - 1), else 0: added to the entry's start
};
SCPR_HD RansRcp rans_rcp(uint32_t freq) {
  RansRcp r;
  r.pad = freq < 2 ? 4095 : 0;
  if (freq < 2) {
    r.rcp = ~0u;
    r.shift = 0;
  } else {
    uint32_t sh = 0;
    while (freq > (1u << sh)) sh++;
    r.rcp = (uint32_t)(((1ull << (sh + 31)) + freq - 1) / freq);
    r.shift = (uint16_t)(sh - 1);
  }
  return r;
}

}  // namespace scpr
