// TEST HARNESS (CPU): compiles the product's host/device model header
// (screenpressor_amd/csrc/scpr_model.hpp) with the host compiler so the
// semantic model code the kernels run can be checked against the oracle
// without a GPU.  Not linked into the product library.
#include <stdint.h>
#include <string.h>
#include "host_model_serial.hpp"

using namespace scpr;

extern "C" {

void hm_chain_colour(const uint8_t* syms, int n, int f0, uint16_t* out) {
  ColState st;
  DenseTab tab;
  col_reset(st);
  auto alloc = [&](ColState& s) { s.dense = 0; return &tab; };
  auto get = [&](ColState&) { return &tab; };
  for (int i = 0; i < n; i++) {
    Ivl e = col_encode(st, syms[i], f0, alloc, get);
    out[2 * i] = e.freq;
    out[2 * i + 1] = e.cum;
  }
}

int hm_chain_colour_dec(const uint16_t* ivl, const uint8_t* expect, int n, int f0, int probe) {
  ColState st;
  DenseTab tab;
  col_reset(st);
  auto alloc = [&](ColState& s) { s.dense = 0; return &tab; };
  auto get = [&](ColState&) { return &tab; };
  int bad = 0;
  for (int i = 0; i < n; i++) {
    int fr = ivl[2 * i], cf = ivl[2 * i + 1];
    int v = fr ? cf + (probe ? fr - 1 : 0) : 0;
    uint8_t c = 0;
    Ivl e;
    if (col_decode(st, v, c, e, alloc, get)) {
      if (!fr || c != expect[i] || e.freq != fr || e.cum != cf) bad++;
    } else {
      if (fr) bad++;
      col_note_raw(st, expect[i], f0, alloc);
    }
  }
  return bad;
}

void hm_chain_fixed(int nsym, const uint16_t* syms, int n, uint16_t* out) {
  FixedTab<512> t;
  fixed_reset(t, nsym);
  for (int i = 0; i < n; i++) {
    Ivl e = fixed_encode(t, syms[i]);
    out[2 * i] = e.freq;
    out[2 * i + 1] = e.cum;
  }
}

int hm_chain_fixed_dec(int nsym, const uint16_t* ivl, const uint16_t* expect, int n, int probe) {
  FixedTab<512> t;
  fixed_reset(t, nsym);
  int bad = 0;
  for (int i = 0; i < n; i++) {
    int fr = ivl[2 * i], cf = ivl[2 * i + 1];
    Ivl e;
    int c = fixed_decode(t, cf + (probe ? fr - 1 : 0), e);
    if (c != expect[i] || e.freq != fr || e.cum != cf) bad++;
  }
  return bad;
}

// the table entry the rANS kernels read for `freq` (rans_rcp, scpr_model.hpp): { rcp, shift, pad }
void hm_rans_params(uint32_t freq, uint32_t* out) {
  const RansRcp r = rans_rcp(freq);
  out[0] = r.rcp, out[1] = r.shift, out[2] = r.pad;
}
// one encoder step in the kernels' algebra (k_rans / k_rans_s: x_max = freq << 19, at most two bytes leave the state, then
// x + start + pad + (x / freq by the reciprocal) * (4096 - freq)); bytes in emission order
uint32_t hm_rans_step(uint32_t x, uint32_t start, uint32_t freq, uint8_t* bytes, int* n) {
  const RansRcp r = rans_rcp(freq);
  const uint32_t xm = freq << 19;
  int k = 0;
  while (x >= xm && k < 2) bytes[k++] = (uint8_t)x, x >>= 8;
  *n = k;
  const uint32_t q = (uint32_t)(((uint64_t)x * r.rcp) >> 32) >> r.shift;
  return x + start + r.pad + q * ((uint32_t)kProbScale - freq);
}

// exact-division check of the reciprocal used by the rANS kernel
uint64_t hm_rcp_mismatches(uint32_t freq, const uint32_t* xs, int n) {
  RansRcp r = rans_rcp(freq);
  uint64_t bad = 0;
  for (int i = 0; i < n; i++) {
    uint32_t x = xs[i];
    uint32_t q = (uint32_t)(((uint64_t)x * r.rcp) >> 32) >> r.shift;
    if (freq == 1) q = x;
    if (q != x / freq) bad++;
  }
  return bad;
}
}
