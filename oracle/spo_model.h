// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of the ScreenPressor v3/v4 adaptive context models, written
// from a reading of the reference sources (cited per function as
// ans_contexts.h:LINE / ans_contexts.cpp:LINE, relative to /root/reference).
//
// PARITY STATUS: "parity unpinned" for the models.  The reference ships no
// tests or golden vectors, and its model code cannot be compiled in this image
// (every translation unit pulls <windows.h> through defines.h:4).  Only
// rans_byte.h is self-contained; it is compiled in place into oracle/_ref and
// pins the rANS primitives (see oracle/Makefile, tests/test_oracle_rans.py).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// anything in this directory.
#pragma once
#include <stdint.h>
#include <string.h>
#include <algorithm>

namespace spo {

enum { kProbBits = 12, kProbScale = 1 << kProbBits };   // ans_contexts.h:66-67
enum { kStepSmall = 50, kStepHash = 25, kStepDense = 16 }; // ans_contexts.h:56-59

// One coder entry: an interval [cum, cum+freq) on the 4096 scale, or a raw
// byte when freq == 0 (byte value in cum).  ans_contexts.h:62-64
struct Ivl {
  uint16_t freq, cum;
};

// ---------------------------------------------------------------------------
// Fixed-alphabet model (run lengths, pixel types, block types, ...).
// ans_contexts.h:1054-1132
// ---------------------------------------------------------------------------
struct FixedModel {
  int nsym = 0;
  int total = 0;  // running sum of counts ("cntsum")
  uint16_t freq[512], cum[512], cnt[512];
  uint8_t bucket[kProbScale >> 7];  // decode accelerator, 128-wide buckets

  void fill_buckets(int j, int cf, int fr) {  // :1081-1084
    int k0 = (cf + 127) >> 7, k1 = ((cf + fr - 1) >> 7) + 1;
    for (int k = k0; k < k1; k++) bucket[k] = (uint8_t)j;
  }
  // renew(): equal probabilities, counts at half.  :1114-1131
  void reset(int n) {
    nsym = n;
    int fr = kProbScale / n, c0 = fr - (fr >> 1), cf = 0;
    total = c0 * n;
    memset(bucket, 0, sizeof bucket);
    for (int i = 0; i < n; i++) {
      freq[i] = (uint16_t)fr;
      cum[i] = (uint16_t)cf;
      cnt[i] = (uint16_t)c0;
      fill_buckets(i, cf, fr);
      cf += fr;
    }
  }
  // incrCnt(): count +16; when the next step would overflow the scale the
  // counts become the new frequencies and are halved.  :1070-1091
  void bump(int c) {
    cnt[c] += kStepDense;
    total += kStepDense;
    if (total + kStepDense > kProbScale) {
      total = 0;
      int cf = 0;
      for (int j = 0; j < nsym; j++) {
        int fr = cnt[j];
        cum[j] = (uint16_t)cf;
        freq[j] = (uint16_t)fr;
        fill_buckets(j, cf, fr);
        cf += fr;
        cnt[j] -= fr >> 1;
        total += cnt[j];
      }
    }
  }
  Ivl encode(int c) {  // :1063-1068  (interval is the one BEFORE the update)
    Ivl iv = {freq[c], cum[c]};
    bump(c);
    return iv;
  }
  int decode(int v, Ivl& iv) {  // :1093-1112
    int j = bucket[v >> 7];
    for (; j < nsym - 1; j++)
      if (cum[j + 1] > v) break;
    iv.freq = freq[j];
    iv.cum = cum[j];
    bump(j);
    return j;
  }
};

// ---------------------------------------------------------------------------
// Small sorted table, capacity 4 (kind 4, inline) or 16 (kind 5).
// ans_contexts.h:155-290
// ---------------------------------------------------------------------------
struct SmallTab {
  uint8_t cap, d, maxpos;
  uint8_t sym[16];
  uint16_t fr[16];

  void zero(int capacity) {
    memset(this, 0, sizeof *this);
    cap = (uint8_t)capacity;
  }
  // create(Cx1&, c): all symbols were seen once, c is the one seen twice. :161-172
  // NB: does not touch maxpos unless c is found (it always is).
  void from_unique(const uint8_t* u, int n, uint8_t c) {
    d = (uint8_t)n;
    for (int i = 0; i < n; i++) sym[i] = u[i];
    std::sort(sym, sym + n);
    for (int i = 0; i < n; i++) {
      if (sym[i] == c) {
        fr[i] = 2 * kStepSmall;
        maxpos = (uint8_t)i;
      } else
        fr[i] = kStepSmall;
    }
    for (int i = n; i < cap; i++) fr[i] = 0;
  }
  void halve(uint16_t& tot) {  // rescale(): :186-193
    int s = 256 - d;
    for (int i = 0; i < d; i++) {
      fr[i] -= fr[i] >> 1;
      s += fr[i];
    }
    tot = (uint16_t)s;
  }
  bool insert_at(int pos, uint8_t c, uint16_t& tot) {  // addSymb(): :174-184
    if (d == cap) return false;
    for (int i = d - 1; i >= pos; i--) {
      sym[i + 1] = sym[i];
      fr[i + 1] = fr[i];
    }
    sym[pos] = c;
    fr[pos] = kStepSmall;
    d++;
    if (maxpos >= pos) maxpos++;
    tot += kStepSmall;
    if (tot + kStepSmall > kProbScale) halve(tot);
    return true;
  }
  static int scale_shift(int tot) {  // :196-199
    int sh = 0;
    while (tot <= kProbScale / 2) {
      tot <<= 1;
      sh++;
    }
    return sh;
  }
  // encode(): :195-236.  Returns false when the table is full and c is new
  // (the interval for c has been produced all the same).
  bool encode(uint8_t c, Ivl& iv, uint16_t& tot) {
    int sh = scale_shift(tot);
    int bonus = (kProbScale - (tot << sh)) >> sh;  // spare code space -> top symbol
    uint16_t keep = fr[maxpos];
    fr[maxpos] = (uint16_t)(fr[maxpos] + bonus);
    int acc = 0, next_unmet = 0, pos = 0;
    for (; pos < d; pos++) {
      int s = sym[pos];
      if (s == c) {
        acc += c - next_unmet;
        iv.cum = (uint16_t)(acc << sh);
        iv.freq = (uint16_t)(fr[pos] << sh);
        fr[maxpos] = keep;
        fr[pos] += kStepSmall;
        tot += kStepSmall;
        if (pos != maxpos && fr[pos] > fr[maxpos]) maxpos = (uint8_t)pos;
        if (tot + kStepSmall > kProbScale) halve(tot);
        return true;
      }
      if (c < s) break;
      acc += s - next_unmet + fr[pos];
      next_unmet = s + 1;
    }
    // new symbol, to be inserted at pos (before a larger one, or at the end)
    acc += c - next_unmet;
    iv.cum = (uint16_t)(acc << sh);
    iv.freq = (uint16_t)(1 << sh);
    fr[maxpos] = keep;
    return insert_at(pos, c, tot);
  }
  // decode(): :238-283
  bool decode(int v, uint8_t& c, Ivl& iv, uint16_t& tot) {
    int sh = scale_shift(tot);
    v >>= sh;
    int bonus = (kProbScale - (tot << sh)) >> sh;
    uint16_t keep = fr[maxpos];
    fr[maxpos] = (uint16_t)(fr[maxpos] + bonus);
    int acc = 0, next_unmet = 0, pos = 0;
    for (; pos < d; pos++) {
      int s = sym[pos];
      int start = acc + s - next_unmet;
      if (v < start) {  // an unmet symbol below s
        c = (uint8_t)(v - acc + next_unmet);
        iv.cum = (uint16_t)(v << sh);
        iv.freq = (uint16_t)(1 << sh);
        fr[maxpos] = keep;
        return insert_at(pos, c, tot);
      }
      int f = fr[pos];
      if (start + f > v) {
        c = (uint8_t)s;
        iv.cum = (uint16_t)(start << sh);
        iv.freq = (uint16_t)(f << sh);
        fr[maxpos] = keep;
        fr[pos] += kStepSmall;
        tot += kStepSmall;
        if (pos != maxpos && fr[pos] > fr[maxpos]) maxpos = (uint8_t)pos;
        if (tot + kStepSmall > kProbScale) halve(tot);
        return true;
      }
      acc = start + f;
      next_unmet = s + 1;
    }
    fr[maxpos] = keep;
    c = (uint8_t)(next_unmet + v - acc);
    iv.cum = (uint16_t)(v << sh);
    iv.freq = (uint16_t)(1 << sh);
    return insert_at(pos, c, tot);
  }
  int exact_total() const {  // Cx4: :303 / Cx5::calcSum: :334-338
    int t = 256 - d;
    for (int i = 0; i < d; i++) t += fr[i];
    return t;
  }
};

// ---------------------------------------------------------------------------
// Kind 6: 17..40 symbols.  Encoder keeps a linear-probing table with
// robin-hood style eviction; decoder keeps the first d slots ordered by
// count.  ans_contexts.h:377-829
// ---------------------------------------------------------------------------
struct HashTab {
  enum { kEmptySym = 1, kMaxSyms = 40 };
  uint8_t d, S, fshift;
  uint8_t sym[64];
  Ivl iv[64];
  uint16_t cnt[65];  // cnt[S] is the running total

  void init(int S0) {  // :438-446
    S = (uint8_t)S0;
    d = 0;
    fshift = 0;
    for (int i = 0; i < 64; i++) sym[i] = kEmptySym;
    memset(iv, 0, sizeof iv);
    memset(cnt, 0, sizeof cnt);
  }
  // add(): returns the slot the ORIGINAL symbol landed in.  :387-415
  int place(uint8_t c, Ivl f) {
    int mask = S - 1, p0 = c & mask, landed = -1;
    uint16_t weight = (uint16_t)(f.freq - (f.freq >> 1));
    for (int j = 0;; j++) {
      int pos = (p0 + j) & mask;
      if (cnt[pos] == 0) {
        if (landed < 0) landed = pos;
        sym[pos] = c;
        iv[pos] = f;
        cnt[pos] = weight;
        d++;
        return landed;
      }
      if (cnt[pos] < weight) {  // evict the weaker occupant, carry it onward
        if (landed < 0) landed = pos;
        std::swap(sym[pos], c);
        std::swap(iv[pos], f);
        std::swap(cnt[pos], weight);
      }
    }
  }
  int append_dec(uint8_t c, Ivl f) {  // addDec(): :417-426
    if (d >= kMaxSyms || d >= S) return -1;
    int pos = d;
    sym[pos] = c;
    iv[pos] = f;
    cnt[pos] = (uint16_t)(f.freq - (f.freq >> 1));
    d++;
    return pos;
  }
  void order_by_freq() {  // sortByFreqs(): :428-436
    for (int i = 0; i < S - 1; i++)
      for (int j = i + 1; j < S; j++)
        if (iv[j].freq > iv[i].freq) {
          std::swap(iv[i], iv[j]);
          std::swap(cnt[i], cnt[j]);
          std::swap(sym[i], sym[j]);
        }
  }
  void recount() {  // calcSum(): :549-555
    int sh = fshift > 0 ? fshift - 1 : 0;
    int sum = (256 - d) << sh;
    for (int i = 0; i < S; i++) sum += cnt[i];
    cnt[S] = (uint16_t)sum;
  }
  // Interval of a symbol that is not in the table: it sits after its nearest
  // lower neighbour, each unmet symbol being 1<<fshift wide.  :596-619
  Ivl unmet(uint8_t c) const {
    Ivl f;
    f.freq = (uint16_t)(1 << fshift);
    f.cum = 0;
    if (c > 0) {
      int low = -1;
      Ivl lf = {0, 0};
      for (int i = 0; i < S; i++)
        if (cnt[i] > 0) {
          int s = sym[i];
          if (s > low && s < c) {
            low = s;
            lf = iv[i];
          }
        }
      if (lf.freq > 0)
        f.cum = (uint16_t)(lf.cum + lf.freq + ((c - low - 1) << fshift));
      else
        f.cum = (uint16_t)(c << fshift);
    }
    return f;
  }
  void widen() {  // grow(): S doubles, everything re-placed, counts kept. :557-575
    uint8_t os[64];
    Ivl oi[64];
    uint16_t oc[65];
    int oldS = S;
    memcpy(os, sym, sizeof os);
    memcpy(oi, iv, sizeof oi);
    memcpy(oc, cnt, sizeof oc);
    uint8_t keep_shift = fshift;
    init(oldS * 2);
    fshift = keep_shift;
    for (int i = 0; i < oldS; i++)
      if (oc[i] > 0) {
        int pos = place(os[i], oi[i]);
        cnt[pos] = oc[i];
      }
    recount();
  }
  void widen_dec() {  // growDec(): :577-594
    int oldS = S;
    uint16_t tot = cnt[oldS];
    S = (uint8_t)(oldS * 2);
    for (int i = d; i < S; i++) {
      sym[i] = kEmptySym;
      iv[i].freq = iv[i].cum = 0;
      cnt[i] = 0;
    }
    cnt[S] = tot;
  }
  // rescale(): all 256 intervals rebuilt from the counts; unmet symbols get
  // 1<<(fshift-1) (min 1); fshift decrements; counts halve.  :742-796
  void rebuild() {
    uint16_t nc[256];
    Ivl nf[256];
    int mask = S - 1;
    int sh = fshift > 0 ? fshift - 1 : 0;
    for (int i = 0; i < 256; i++) nc[i] = (uint16_t)(1 << sh);
    bool reorder = false;
    for (int i = 0; i < S; i++)
      if (cnt[i] > 0) {
        int s = sym[i];
        nc[s] = cnt[i];
        int p0 = s & mask;
        if (i != p0) reorder = reorder || (cnt[p0] < cnt[i]);
      }
    int acc = 0;
    for (int i = 0; i < 256; i++) {
      nf[i].freq = nc[i];
      nf[i].cum = (uint16_t)acc;
      acc += nc[i];
    }
    if (fshift > 0) fshift--;
    if (!reorder) {
      int sh2 = fshift > 0 ? fshift - 1 : 0;
      int total = (256 - d) << sh2;
      for (int i = 0; i < S; i++)
        if (cnt[i]) {
          cnt[i] -= cnt[i] >> 1;
          total += cnt[i];
          iv[i] = nf[sym[i]];
        }
      cnt[S] = (uint16_t)total;
    } else {
      uint8_t ls[64];
      int n = 0;
      for (int i = 0; i < S; i++)
        if (cnt[i]) {
          ls[n++] = sym[i];
          cnt[i] = 0;
        }
      d = 0;
      for (int i = 0; i < n; i++) place(ls[i], nf[ls[i]]);
      recount();
    }
  }
  void rebuild_dec() {  // rescaleDec(): :800-828
    uint16_t nc[256];
    Ivl nf[256];
    int sh = fshift > 0 ? fshift - 1 : 0;
    for (int i = 0; i < 256; i++) nc[i] = (uint16_t)(1 << sh);
    for (int i = 0; i < d; i++) nc[sym[i]] = cnt[i];
    int acc = 0;
    for (int i = 0; i < 256; i++) {
      nf[i].freq = nc[i];
      nf[i].cum = (uint16_t)acc;
      acc += nc[i];
    }
    if (fshift > 0) fshift--;
    int sh2 = fshift > 0 ? fshift - 1 : 0;
    int total = (256 - d) << sh2;
    for (int i = 0; i < d; i++) {
      cnt[i] -= cnt[i] >> 1;
      total += cnt[i];
      iv[i] = nf[sym[i]];
    }
    cnt[S] = (uint16_t)total;
  }
  void bump(int pos) {  // incrCnt(): :686-691
    int step = kStepHash << fshift;
    cnt[pos] = (uint16_t)(cnt[pos] + step);
    cnt[S] = (uint16_t)(cnt[S] + step);
    if (cnt[S] + step > kProbScale) rebuild();
  }
  void bump_dec(int pos) {  // incrCntDec(): :693-703
    int step = kStepHash << fshift;
    cnt[pos] = (uint16_t)(cnt[pos] + step);
    cnt[S] = (uint16_t)(cnt[S] + step);
    if (pos > 0 && cnt[pos] > cnt[pos - 1]) {
      std::swap(cnt[pos], cnt[pos - 1]);
      std::swap(iv[pos], iv[pos - 1]);
      std::swap(sym[pos], sym[pos - 1]);
    }
    if (cnt[S] + step > kProbScale) rebuild_dec();
  }
  // placeSymbol(): :621-638
  bool admit(uint8_t c, int pos, Ivl& out) {
    out = unmet(c);
    if (S == 32 && d >= 24) {
      widen();
      bump(place(c, out));
      return true;
    }
    if (d >= kMaxSyms) return false;
    iv[pos] = out;
    sym[pos] = c;
    cnt[pos] = (uint16_t)(out.freq - (out.freq >> 1));
    d++;
    bump(pos);
    return true;
  }
  bool hit(int pos, Ivl& out) {  // found(): :680-684
    out = iv[pos];
    bump(pos);
    return true;
  }
  // encode(): :640-678.  false => caller upgrades to the dense kind.
  bool encode(uint8_t c, Ivl& out) {
    int mask = S - 1, p0 = c & mask;
    if (sym[p0] == c) return cnt[p0] == 0 ? admit(c, p0, out) : hit(p0, out);
    if (c != kEmptySym) {
      if (cnt[p0] == 0) return admit(c, p0, out);
      for (int j = 1; j < S; j++) {
        int pos = (p0 + j) & mask;
        uint8_t s = sym[pos];
        if (s == c) return hit(pos, out);
        if (s == kEmptySym && cnt[pos] == 0) return admit(c, pos, out);
      }
    } else {
      for (int j = 0; j < S; j++) {
        int pos = (p0 + j) & mask;
        if (sym[pos] == kEmptySym) return cnt[pos] > 0 ? hit(pos, out) : admit(c, pos, out);
      }
    }
    out = unmet(c);  // table full
    if (S >= 64) return false;
    Ivl f = out;
    widen();
    bump(place(c, f));
    return true;
  }
  // decode(): :705-740
  bool decode(int v, uint8_t& c, Ivl& out) {
    Ivl lf = {0, 0};
    int low = 0;
    for (int i = 0; i < d; i++) {
      int cf = iv[i].cum;
      if (cf <= v) {
        if (cf + iv[i].freq > v) {
          c = sym[i];
          out = iv[i];
          bump_dec(i);
          return true;
        }
        if (cf >= lf.cum) {
          lf = iv[i];
          low = sym[i];
        }
      }
    }
    Ivl f;
    f.freq = (uint16_t)(1 << fshift);
    if (lf.freq) {
      int base = lf.cum + lf.freq;
      int x = (v - base) >> fshift;
      c = (uint8_t)(x + low + 1);
      f.cum = (uint16_t)(base + (x << fshift));
    } else {
      c = (uint8_t)(v >> fshift);
      f.cum = (uint16_t)(c << fshift);
    }
    out = f;
    int p = append_dec(c, f);
    if (p < 0) {
      if (S == 64) return false;
      widen_dec();
      p = append_dec(c, f);
    }
    bump_dec(p);
    return true;
  }
};

// ---------------------------------------------------------------------------
// Kind 7: dense 256-entry model.  ans_contexts.h:847-998
// ---------------------------------------------------------------------------
struct DenseTab {
  int total;
  uint16_t freq[256], cum[256], cnt[256];
  uint8_t bucket[kProbScale >> 7];

  void fill_buckets(int j, int cf, int fr) {
    int k0 = (cf + 127) >> 7, k1 = ((cf + fr - 1) >> 7) + 1;
    for (int k = k0; k < k1; k++) bucket[k] = (uint8_t)j;
  }
  void bump(int c) {  // incrCnt(): :959-981
    cnt[c] += kStepDense;
    total += kStepDense;
    if (total + kStepDense > kProbScale) {
      total = 0;
      int cf = 0;
      for (int j = 0; j < 256; j++) {
        int fr = cnt[j];
        cum[j] = (uint16_t)cf;
        freq[j] = (uint16_t)fr;
        fill_buckets(j, cf, fr);
        cf += fr;
        cnt[j] -= fr >> 1;
        total += cnt[j];
      }
    }
  }
  // create(Cx3&, c): d unique symbols, c seen for the second time.  :917-951
  void from_unique(const uint8_t* u, int d, uint8_t c) {
    memset(this, 0, sizeof *this);
    for (int i = 0; i < 256; i++) freq[i] = cnt[i] = 1;
    int f0 = (kProbScale - (256 - d)) / (d + 1), c0 = f0 - (f0 >> 1);
    for (int i = 0; i < d; i++) {
      freq[u[i]] = (uint16_t)f0;
      cnt[u[i]] = (uint16_t)c0;
    }
    freq[c] = (uint16_t)(freq[c] + f0);
    cnt[c] = (uint16_t)(cnt[c] + kStepDense);
    int cf = 0;
    total = 0;
    for (int i = 0; i < 256; i++) {
      total += cnt[i];
      cum[i] = (uint16_t)cf;
      fill_buckets(i, cf, freq[i]);
      cf += freq[i];
    }
  }
  // create(const Cx6&, c): the triggering symbol c gets no count.  :868-915
  void from_hash(const HashTab& h) {
    memset(this, 0, sizeof *this);
    total = h.cnt[h.S];
    for (int i = 0; i < h.S; i++)
      if (h.cnt[i]) {
        int s = h.sym[i];
        freq[s] = h.iv[i].freq;
        cum[s] = h.iv[i].cum;
        cnt[s] = h.cnt[i];
      }
    int wide = 1 << h.fshift, base = wide - (wide >> 1), cf = 0;
    for (int i = 0; i < 256; i++) {
      if (!freq[i]) {
        freq[i] = (uint16_t)wide;
        cum[i] = (uint16_t)cf;
        cnt[i] = (uint16_t)base;
      }
      fill_buckets(i, cf, freq[i]);
      cf += freq[i];
    }
  }
  Ivl encode(uint8_t c) {  // :953-957
    Ivl out = {freq[c], cum[c]};
    bump(c);
    return out;
  }
  uint8_t decode(int v, Ivl& out) {  // :983-997
    int j = bucket[v >> 7];
    for (; j < 255; j++)
      if (cum[j + 1] > v) break;
    out.freq = freq[j];
    out.cum = cum[j];
    bump(j);
    return (uint8_t)j;
  }
};

// ---------------------------------------------------------------------------
// diagnostics: how many colour symbols met each kind on the encoder side (tests / design notes only)
inline uint64_t* kind_hist() {
  static uint64_t h[8];
  return h;
}
// One colour context: the 7-kind state machine.
// ans_contexts.h:98-150 (kinds 1-3), :1018-1051, ans_contexts.cpp:3-84
// ---------------------------------------------------------------------------
struct ColourCtx {
  uint8_t kind = 0;
  uint16_t d = 0;          // distinct symbols while kind <= 3
  uint16_t small_total = 0;  // kind 5's cached total ("cntsum", drifts by design)
  uint8_t* uniq = nullptr;  // kinds 1..3: symbols in arrival order (256 B)
  SmallTab* small = nullptr;
  HashTab* hash = nullptr;
  DenseTab* dense = nullptr;

  void reset() {  // renew(): :1050
    delete[] uniq;
    delete small;
    delete hash;
    delete dense;
    uniq = nullptr;
    small = nullptr;
    hash = nullptr;
    dense = nullptr;
    kind = 0;
    d = 0;
  }
  ~ColourCtx() { reset(); }

  bool seen(uint8_t c) const {
    for (int i = 0; i < d; i++)
      if (uniq[i] == c) return true;
    return false;
  }
  // kinds 0..3: every symbol is sent raw; remember the set.  A repeat
  // promotes to a counting kind.  ans_contexts.cpp:3-31
  void note_raw(uint8_t c, bool decoding, int f0) {
    if (kind == 0) {
      uniq = new uint8_t[256];
      uniq[0] = c;
      d = 1;
      kind = 1;
      return;
    }
    if (!seen(c)) {
      // capacity 14 / 64 / 256: overflow moves to the next kind, which
      // stores the newcomer too.  ans_contexts.h:83-90, :120-124, :142-146
      uniq[d++] = c;
      if (kind == 1 && d == 15) kind = 2;
      else if (kind == 2 && d == 65) kind = 3;
      return;
    }
    if (kind == 1) {  // ans_contexts.cpp:5-8
      small = new SmallTab;
      if (d <= 4) {
        small->zero(4);  // inline union storage in the reference; every field is set below
        small->from_unique(uniq, d, c);
        kind = 4;
      } else {
        small->zero(16);
        small->from_unique(uniq, d, c);
        small_total = (uint16_t)small->exact_total();
        kind = 5;
      }
    } else if (kind == 2) {  // -> kind 6 via create23.  ans_contexts.h:491-533
      hash = new HashTab;
      hash->init(d <= 32 ? 32 : 64);
      int tot = 256 - d + d * f0 + f0;
      int sh = SmallTab::scale_shift(tot);
      std::sort(uniq, uniq + d);
      int acc = 0, next_unmet = 0;
      for (int i = 0; i < d; i++) {
        int s = uniq[i];
        acc += s - next_unmet;
        int cf = (s == c) ? 2 * f0 : f0;
        Ivl f = {(uint16_t)(cf << sh), (uint16_t)(acc << sh)};
        hash->place((uint8_t)s, f);
        acc += cf;
        next_unmet = s + 1;
      }
      hash->fshift = (uint8_t)sh;
      hash->recount();
      if (decoding) hash->order_by_freq();  // ans_contexts.cpp:18
      kind = 6;
    } else {  // kind 3 -> 7
      dense = new DenseTab;
      dense->from_unique(uniq, d, c);
      kind = 7;
    }
    delete[] uniq;
    uniq = nullptr;
  }

  void small_to_hash(uint8_t c, bool decoding) {  // Cx6::create(Cx5&, c): :454-489
    hash = new HashTab;
    hash->init(32);
    int n = small->d;
    int tot = small->exact_total();
    int sh = SmallTab::scale_shift(tot);
    int acc = 0, next_unmet = 0;
    for (int i = 0; i < n; i++) {
      int s = small->sym[i];
      acc += s - next_unmet;
      int cf = small->fr[i];
      Ivl f = {(uint16_t)(cf << sh), (uint16_t)(acc << sh)};
      hash->place((uint8_t)s, f);
      acc += cf;
      next_unmet = s + 1;
    }
    hash->fshift = (uint8_t)sh;
    Ivl f = hash->unmet(c);
    int p = hash->place(c, f);
    hash->bump(p);
    hash->recount();
    if (decoding) hash->order_by_freq();  // ans_contexts.cpp:65
    delete small;
    small = nullptr;
    kind = 6;
  }
  void small4_to_16(uint8_t c) {  // Cx5::create(Cx4&, c): :350-369 (maxpos stays 0)
    SmallTab* t = new SmallTab;
    t->zero(16);
    int i = 0, j = 0, n = small->d, sum = 0;
    while (i < n && small->sym[i] < c) {
      t->sym[j] = small->sym[i];
      sum += t->fr[j] = small->fr[i];
      i++, j++;
    }
    t->sym[j] = c;
    sum += t->fr[j] = kStepSmall;
    j++;
    while (i < n) {
      t->sym[j] = small->sym[i];
      sum += t->fr[j] = small->fr[i];
      i++, j++;
    }
    t->d = (uint8_t)(n + 1);
    if (sum > kProbScale) t->halve(small_total);
    small_total = (uint16_t)t->exact_total();
    delete small;
    small = t;
    kind = 5;
  }
  void hash_to_dense() {
    dense = new DenseTab;
    dense->from_hash(*hash);
    delete hash;
    hash = nullptr;
    kind = 7;
  }

  // Context::encode: false => symbol goes out raw.  ans_contexts.cpp:34-50
  bool encode(uint8_t c, Ivl& out, int f0) {
    kind_hist()[kind]++;
    switch (kind) {
      case 0: case 1: case 2: case 3:
        note_raw(c, false, f0);
        return false;
      case 4: {
        uint16_t tot = (uint16_t)(small->fr[0] + small->fr[1] + small->fr[2] + small->fr[3] + 256 - small->d);
        if (!small->encode(c, out, tot)) small4_to_16(c);
        return true;
      }
      case 5:
        if (!small->encode(c, out, small_total)) small_to_hash(c, false);
        return true;
      case 6:
        if (!hash->encode(c, out)) hash_to_dense();
        return true;
      default:
        out = dense->encode(c);
        return true;
    }
  }
  // Context::decode: false => caller reads a raw byte and calls note_raw.
  // ans_contexts.cpp:61-74
  bool decode(int v, uint8_t& c, Ivl& out) {
    switch (kind) {
      case 0: case 1: case 2: case 3:
        return false;
      case 4: {
        uint16_t tot = (uint16_t)(small->fr[0] + small->fr[1] + small->fr[2] + small->fr[3] + 256 - small->d);
        if (!small->decode(v, c, out, tot)) small4_to_16(c);
        return true;
      }
      case 5:
        if (!small->decode(v, c, out, small_total)) small_to_hash(c, true);
        return true;
      case 6:
        if (!hash->decode(v, c, out)) hash_to_dense();
        return true;
      default:
        c = dense->decode(v, out);
        return true;
    }
  }
};

}  // namespace spo
