// TEST INFRASTRUCTURE: the serial (one symbol at a time) form of the context models over the product's
// semantic state (screenpressor_amd/csrc/scpr_model.hpp: ColState, DenseTab).  The kernels do not use any of
// this - they run the wave-cooperative form in scpr_wave.hpp; tests/host_model_harness.cpp compiles it for the
// host and tests/test_host_model.py checks it against the oracle's literal structures, which pins the semantic
// state (dense partitions instead of the reference's hash table) without a GPU.
#pragma once
#include "scpr_model.hpp"

namespace scpr {
// ------------------------------------------------------------------ fixed ---
// FixedSizeRansCtx<NSym> (ans_contexts.h:1054-1132), serial form.
template <int CAP>
struct FixedTab {
  uint16_t freq[CAP], cum[CAP], cnt[CAP];
  int32_t total;
  int32_t nsym;
};

template <int CAP>
SCPR_HD void fixed_reset(FixedTab<CAP>& t, int n) {  // renew(): :1114-1131
  int fr = kProbScale / n, c0 = fr - (fr >> 1), cf = 0;
  t.nsym = n;
  t.total = c0 * n;
  for (int i = 0; i < n; i++) {
    t.freq[i] = (uint16_t)fr;
    t.cum[i] = (uint16_t)cf;
    t.cnt[i] = (uint16_t)c0;
    cf += fr;
  }
}
template <int CAP>
SCPR_HD void fixed_bump(FixedTab<CAP>& t, int c) {  // incrCnt(): :1070-1091
  t.cnt[c] = (uint16_t)(t.cnt[c] + kStepDense);
  t.total += kStepDense;
  if (t.total + kStepDense > kProbScale) {
    int cf = 0, tot = 0;
    for (int j = 0; j < t.nsym; j++) {
      int fr = t.cnt[j];
      t.cum[j] = (uint16_t)cf;
      t.freq[j] = (uint16_t)fr;
      cf += fr;
      fr -= fr >> 1;
      t.cnt[j] = (uint16_t)fr;
      tot += fr;
    }
    t.total = tot;
  }
}
template <int CAP>
SCPR_HD Ivl fixed_encode(FixedTab<CAP>& t, int c) {  // :1063-1068
  Ivl e = {t.freq[c], t.cum[c]};
  fixed_bump(t, c);
  return e;
}
// decode(): :1093-1112.  The reference walks forward from a 128-wide bucket;
// cum[] is strictly increasing (every freq >= 1), so a binary search for the
// last j with cum[j] <= v names the same symbol.
template <int CAP>
SCPR_HD int fixed_decode(FixedTab<CAP>& t, int v, Ivl& e) {
  int lo = 0, hi = t.nsym - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (t.cum[mid] <= v) lo = mid; else hi = mid - 1;
  }
  e.freq = t.freq[lo];
  e.cum = t.cum[lo];
  fixed_bump(t, lo);
  return lo;
}

SCPR_HD void col_reset(ColState& st) {  // Context::renew, ans_contexts.h:1050
  st.kind = 0;
  st.d = 0;
  st.dense = 0xFFFFFFFFu;
}

SCPR_HD int scale_shift(int tot) {  // while (tot <= 2048) tot <<= 1  (:196-199)
  int sh = 0;
  while (tot <= kProbScale / 2) {
    tot <<= 1;
    sh++;
  }
  return sh;
}

// -- small sorted tables (SmallContext<S>, ans_contexts.h:155-290) -----------
SCPR_HD int small_cap(const ColState& st) { return st.kind == 4 ? 4 : 16; }
SCPR_HD int small_exact_total(const ColState& st) {  // :303, :334-338
  int t = 256 - st.d;
  for (int i = 0; i < st.d; i++) t += st.u.s.fr[i];
  return t;
}
SCPR_HD void small_halve(ColState& st, int& tot) {  // rescale(): :186-193
  int s = 256 - st.d;
  for (int i = 0; i < st.d; i++) {
    uint16_t f = st.u.s.fr[i];
    f = (uint16_t)(f - (f >> 1));
    st.u.s.fr[i] = f;
    s += f;
  }
  tot = s;
}
// addSymb(): :174-184.  false when the table is full (nothing modified).
SCPR_HD bool small_insert(ColState& st, int pos, uint8_t c, int& tot) {
  if (st.d == small_cap(st)) return false;
  for (int i = st.d - 1; i >= pos; i--) {
    st.u.s.sym[i + 1] = st.u.s.sym[i];
    st.u.s.fr[i + 1] = st.u.s.fr[i];
  }
  st.u.s.sym[pos] = c;
  st.u.s.fr[pos] = kStepSmall;
  st.d++;
  if (st.maxpos >= pos) st.maxpos++;
  tot += kStepSmall;
  if (tot + kStepSmall > kProbScale) small_halve(st, tot);
  return true;
}
SCPR_HD void small_hit(ColState& st, int pos, int& tot) {  // :210-215
  st.u.s.fr[pos] = (uint16_t)(st.u.s.fr[pos] + kStepSmall);
  tot += kStepSmall;
  if (pos != st.maxpos && st.u.s.fr[pos] > st.u.s.fr[st.maxpos]) st.maxpos = (uint8_t)pos;
  if (tot + kStepSmall > kProbScale) small_halve(st, tot);
}
// encode(): :195-236.  The spare code space (4096 - scaled total) is lent to
// the most frequent symbol for the duration of the call.
SCPR_HD bool small_encode(ColState& st, uint8_t c, Ivl& e, int& tot) {
  const int sh = scale_shift(tot);
  const int bonus = (kProbScale - (tot << sh)) >> sh;
  int acc = 0, next_unmet = 0, pos = 0;
  for (; pos < st.d; pos++) {
    int s = st.u.s.sym[pos];
    int f = st.u.s.fr[pos] + (pos == st.maxpos ? bonus : 0);
    if (s == c) {
      acc += c - next_unmet;
      e.cum = (uint16_t)(acc << sh);
      e.freq = (uint16_t)(f << sh);
      small_hit(st, pos, tot);
      return true;
    }
    if (c < s) break;
    acc += s - next_unmet + (f & 0xFFFF);
    next_unmet = s + 1;
  }
  acc += c - next_unmet;
  e.cum = (uint16_t)(acc << sh);
  e.freq = (uint16_t)(1 << sh);
  return small_insert(st, pos, c, tot);
}
// decode(): :238-283
SCPR_HD bool small_decode(ColState& st, int v, uint8_t& c, Ivl& e, int& tot) {
  const int sh = scale_shift(tot);
  const int bonus = (kProbScale - (tot << sh)) >> sh;
  v >>= sh;
  int acc = 0, next_unmet = 0, pos = 0;
  for (; pos < st.d; pos++) {
    int s = st.u.s.sym[pos];
    int f = (st.u.s.fr[pos] + (pos == st.maxpos ? bonus : 0)) & 0xFFFF;
    int start = acc + s - next_unmet;
    if (v < start) {
      c = (uint8_t)(v - acc + next_unmet);
      e.cum = (uint16_t)(v << sh);
      e.freq = (uint16_t)(1 << sh);
      return small_insert(st, pos, c, tot);
    }
    if (start + f > v) {
      c = (uint8_t)s;
      e.cum = (uint16_t)(start << sh);
      e.freq = (uint16_t)(f << sh);
      small_hit(st, pos, tot);
      return true;
    }
    acc = start + f;
    next_unmet = s + 1;
  }
  c = (uint8_t)(next_unmet + v - acc);
  e.cum = (uint16_t)(v << sh);
  e.freq = (uint16_t)(1 << sh);
  return small_insert(st, pos, c, tot);
}

// -- set helpers --------------------------------------------------------------
SCPR_HD bool set_has(const uint32_t* w, int c) { return (w[c >> 5] >> (c & 31)) & 1u; }
SCPR_HD void set_add(uint32_t* w, int c) { w[c >> 5] |= 1u << (c & 31); }

// -- kind 6 / 7 dense tables --------------------------------------------------
// Cx6::rescale (ans_contexts.h:742-796): met symbols take their counts as
// widths, unmet ones 1<<(fshift-1) (min 1); fshift steps down; counts halve.
SCPR_HD void hash_rebuild(ColState& st, DenseTab& t) {
  const int w = 1 << (st.fshift > 0 ? st.fshift - 1 : 0);
  if (st.fshift > 0) st.fshift--;
  const int base = (st.fshift > 0) ? st.fshift - 1 : 0;
  int cf = 0, tot = (256 - st.d) << base;
  for (int j = 0; j < 256; j++) {
    t.cum[j] = (uint16_t)cf;
    if (set_has(st.u.seen, j)) {
      int fr = t.cnt[j];
      t.freq[j] = (uint16_t)fr;
      cf += fr;
      fr -= fr >> 1;
      t.cnt[j] = (uint16_t)fr;
      tot += fr;
    } else {
      t.freq[j] = (uint16_t)w;
      cf += w;
    }
  }
  st.total = (uint16_t)tot;
}
SCPR_HD void hash_bump(ColState& st, DenseTab& t, int c) {  // incrCnt(): :686-691
  const int step = kStepHash << st.fshift;
  t.cnt[c] = (uint16_t)(t.cnt[c] + step);
  st.total = (uint16_t)(st.total + step);
  if (st.total + step > kProbScale) hash_rebuild(st, t);
}
// Cx7::create(const Cx6&): :868-915.  Unmet symbols keep their width and get a
// half-width count; the running total carries over unchanged.
SCPR_HD void hash_to_dense(ColState& st, DenseTab& t) {
  const int w = 1 << st.fshift, base = w - (w >> 1);
  for (int j = 0; j < 256; j++)
    if (!set_has(st.u.seen, j)) t.cnt[j] = (uint16_t)base;
  st.kind = 7;
}
SCPR_HD void dense_bump(ColState& st, DenseTab& t, int c) {  // Cx7::incrCnt: :959-981
  t.cnt[c] = (uint16_t)(t.cnt[c] + kStepDense);
  int tot = st.total + kStepDense;
  if (tot + kStepDense > kProbScale) {
    int cf = 0;
    tot = 0;
    for (int j = 0; j < 256; j++) {
      int fr = t.cnt[j];
      t.cum[j] = (uint16_t)cf;
      t.freq[j] = (uint16_t)fr;
      cf += fr;
      fr -= fr >> 1;
      t.cnt[j] = (uint16_t)fr;
      tot += fr;
    }
  }
  st.total = (uint16_t)tot;
}
// last j with cum[j] <= v; cum[] is non-decreasing and every width >= 1
SCPR_HD int dense_find(const DenseTab& t, int v) {
  int lo = 0, hi = 255;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (t.cum[mid] <= v) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// -- promotions ---------------------------------------------------------------
// kind 1 -> 4/5 on the first repeated symbol c (ans_contexts.cpp:5-8,
// ans_contexts.h:161-172): every symbol 50, the repeated one 100.
SCPR_HD void promote_unique_to_small(ColState& st, uint8_t c) {
  uint32_t seen[8];
  for (int i = 0; i < 8; i++) seen[i] = st.u.seen[i];
  int n = 0;
  for (int s = 0; s < 256; s++)
    if (set_has(seen, s)) {
      st.u.s.sym[n] = (uint8_t)s;
      st.u.s.fr[n] = (s == c) ? 2 * kStepSmall : kStepSmall;
      if (s == c) st.maxpos = (uint8_t)n;
      n++;
    }
  for (int i = n; i < 16; i++) st.u.s.fr[i] = 0;
  st.kind = (n <= 4) ? 4 : 5;
  if (st.kind == 5) st.total = (uint16_t)small_exact_total(st);
}
// kind 4 -> 5 when a 5th symbol arrives (Cx5::create(Cx4&, c), :350-369).
// The new table starts with maxpos = 0 whatever the counts say (the
// reference value-initialises it and never sets it).
SCPR_HD void promote_small4_to_16(ColState& st, uint8_t c) {
  int pos = 0;
  while (pos < 4 && st.u.s.sym[pos] < c) pos++;
  for (int i = 3; i >= pos; i--) {
    st.u.s.sym[i + 1] = st.u.s.sym[i];
    st.u.s.fr[i + 1] = st.u.s.fr[i];
  }
  st.u.s.sym[pos] = c;
  st.u.s.fr[pos] = kStepSmall;
  for (int i = 5; i < 16; i++) st.u.s.fr[i] = 0;
  st.d = 5;
  st.maxpos = 0;
  st.kind = 5;
  st.total = (uint16_t)small_exact_total(st);
}
// kind 5 -> 6 when a 17th symbol arrives (Cx6::create(Cx5&, c), :454-489):
// the live table is frozen into intervals (no bonus), then c is added and
// counted once.
SCPR_HD void promote_small_to_hash(ColState& st, DenseTab& t, uint8_t c) {
  const int tot = small_exact_total(st), sh = scale_shift(tot), n = st.d;
  const int w = 1 << sh, base = w - (w >> 1);
  uint8_t osym[16];
  uint16_t ofr[16];
  for (int i = 0; i < 16; i++) {  // the small table shares storage with the set
    osym[i] = st.u.s.sym[i];
    ofr[i] = st.u.s.fr[i];
  }
  for (int i = 0; i < 8; i++) st.u.seen[i] = 0;
  int k = 0, cf = 0, sum = 0;
  for (int j = 0; j < 256; j++) {
    t.cum[j] = (uint16_t)cf;
    if (k < n && osym[k] == j) {
      int fr = ofr[k] << sh;
      t.freq[j] = (uint16_t)fr;
      t.cnt[j] = (uint16_t)(fr - (fr >> 1));
      set_add(st.u.seen, j);
      k++;
    } else {
      t.freq[j] = (uint16_t)w;
      t.cnt[j] = 0;
    }
    cf += t.freq[j];
  }
  set_add(st.u.seen, c);
  t.cnt[c] = (uint16_t)(base + (kStepHash << sh));
  st.fshift = (uint8_t)sh;
  st.d = (uint16_t)(n + 1);
  for (int j = 0; j < 256; j++)
    if (set_has(st.u.seen, j)) sum += t.cnt[j];
  st.total = (uint16_t)(((256 - st.d) << (sh > 0 ? sh - 1 : 0)) + sum);
  st.kind = 6;
}
// kind 2 -> 6 on the first repeat (Cx6::create23, :491-531): every symbol f0,
// the repeated one 2*f0; f0 = 32 (v4) / 64 (v3), screencap.cpp:1613-1614.
SCPR_HD void promote_unique_to_hash(ColState& st, DenseTab& t, uint8_t c, int f0) {
  const int n = st.d, tot = 256 - n + n * f0 + f0, sh = scale_shift(tot);
  const int w = 1 << sh;
  int cf = 0, sum = 0;
  for (int j = 0; j < 256; j++) {
    t.cum[j] = (uint16_t)cf;
    if (set_has(st.u.seen, j)) {
      int fr = ((j == c) ? 2 * f0 : f0) << sh;
      t.freq[j] = (uint16_t)fr;
      t.cnt[j] = (uint16_t)((fr & 0xFFFF) - ((fr & 0xFFFF) >> 1));
      sum += t.cnt[j];
    } else {
      t.freq[j] = (uint16_t)w;
      t.cnt[j] = 0;
    }
    cf += t.freq[j];
  }
  st.fshift = (uint8_t)sh;
  st.total = (uint16_t)(((256 - n) << (sh > 0 ? sh - 1 : 0)) + sum);
  st.kind = 6;
}
// kind 3 -> 7 on the first repeat (Cx7::create(Cx3&, c), :917-951)
SCPR_HD void promote_unique_to_dense(ColState& st, DenseTab& t, uint8_t c) {
  const int d = st.d, f0 = (kProbScale - (256 - d)) / (d + 1), c0 = f0 - (f0 >> 1);
  int cf = 0, tot = 0;
  for (int j = 0; j < 256; j++) {
    int fr = 1, cn = 1;
    if (set_has(st.u.seen, j)) {
      fr = f0;
      cn = c0;
    }
    if (j == c) {
      fr += f0;
      cn += kStepDense;
    }
    t.freq[j] = (uint16_t)fr;
    t.cnt[j] = (uint16_t)cn;
    t.cum[j] = (uint16_t)cf;
    cf += fr;
    tot += cn;
  }
  st.total = (uint16_t)tot;
  st.kind = 7;
}

// A symbol passes through a context that has not seen any symbol twice
// (encoder: Context::encode kinds 0-3; decoder: Context::update), both in
// ans_contexts.cpp:3-59.  `alloc` hands out a DenseTab when one is needed.
template <class Alloc>
SCPR_HD void col_note_raw(ColState& st, uint8_t c, int f0, Alloc&& alloc) {
  if (st.kind == 0) {
    for (int i = 0; i < 8; i++) st.u.seen[i] = 0;
    set_add(st.u.seen, c);
    st.d = 1;
    st.kind = 1;
    return;
  }
  if (!set_has(st.u.seen, c)) {  // capacities 14 / 64 / 256 (ans_contexts.h:100, :118, :140)
    set_add(st.u.seen, c);
    st.d++;
    if (st.kind == 1 && st.d == 15) st.kind = 2;
    else if (st.kind == 2 && st.d == 65) st.kind = 3;
    return;
  }
  if (st.kind == 1) {
    promote_unique_to_small(st, c);
  } else if (st.kind == 2) {
    DenseTab* t = alloc(st);
    promote_unique_to_hash(st, *t, c, f0);
  } else {
    DenseTab* t = alloc(st);
    promote_unique_to_dense(st, *t, c);
  }
}

// Encoder side of one symbol: Context::encode, ans_contexts.cpp:34-50.
// `tab` maps st.dense to its DenseTab.  Returns the coder entry.
template <class Alloc, class Tab>
SCPR_HD Ivl col_encode(ColState& st, uint8_t c, int f0, Alloc&& alloc, Tab&& tab) {
  Ivl e;
  switch (st.kind) {
    case 0: case 1: case 2: case 3:
      col_note_raw(st, c, f0, alloc);
      e.freq = 0;
      e.cum = c;
      return e;
    case 4: {
      int tot = small_exact_total(st);
      if (!small_encode(st, c, e, tot)) promote_small4_to_16(st, c);
      return e;
    }
    case 5: {
      int tot = st.total;
      bool ok = small_encode(st, c, e, tot);
      st.total = (uint16_t)tot;
      if (!ok) {
        DenseTab* t = alloc(st);
        promote_small_to_hash(st, *t, c);
      }
      return e;
    }
    case 6: {
      DenseTab& t = *tab(st);
      e.freq = t.freq[c];
      e.cum = t.cum[c];
      if (set_has(st.u.seen, c)) {
        hash_bump(st, t, c);
      } else if (st.d >= kHashMaxSyms) {  // :631 / :670: the 41st symbol goes uncounted
        hash_to_dense(st, t);
      } else {  // placeSymbol, :621-638
        set_add(st.u.seen, c);
        t.cnt[c] = (uint16_t)(e.freq - (e.freq >> 1));
        st.d++;
        hash_bump(st, t, c);
      }
      return e;
    }
    default: {
      DenseTab& t = *tab(st);
      e.freq = t.freq[c];
      e.cum = t.cum[c];
      dense_bump(st, t, c);
      return e;
    }
  }
}

// Decoder side: Context::decode, ans_contexts.cpp:61-74.  v = state & 4095.
// Returns false for kinds 0-3: the caller reads a raw byte and calls
// col_note_raw.
template <class Alloc, class Tab>
SCPR_HD bool col_decode(ColState& st, int v, uint8_t& c, Ivl& e, Alloc&& alloc, Tab&& tab) {
  switch (st.kind) {
    case 0: case 1: case 2: case 3:
      return false;
    case 4: {
      int tot = small_exact_total(st);
      if (!small_decode(st, v, c, e, tot)) promote_small4_to_16(st, c);
      return true;
    }
    case 5: {
      int tot = st.total;
      bool ok = small_decode(st, v, c, e, tot);
      st.total = (uint16_t)tot;
      if (!ok) {
        DenseTab* t = alloc(st);
        promote_small_to_hash(st, *t, c);
      }
      return true;
    }
    case 6: {  // Cx6::decode, ans_contexts.h:705-740
      DenseTab& t = *tab(st);
      int j = dense_find(t, v);
      c = (uint8_t)j;
      e.freq = t.freq[j];
      e.cum = t.cum[j];
      if (set_has(st.u.seen, j)) {
        hash_bump(st, t, j);
      } else if (st.d >= kHashMaxSyms) {
        hash_to_dense(st, t);
      } else {
        set_add(st.u.seen, j);
        t.cnt[j] = (uint16_t)(e.freq - (e.freq >> 1));
        st.d++;
        hash_bump(st, t, j);
      }
      return true;
    }
    default: {  // Cx7::decode, :983-997
      DenseTab& t = *tab(st);
      int j = dense_find(t, v);
      c = (uint8_t)j;
      e.freq = t.freq[j];
      e.cum = t.cum[j];
      dense_bump(st, t, j);
      return true;
    }
  }
}

}  // namespace scpr
