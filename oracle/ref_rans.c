/* ORACLE — TEST INFRASTRUCTURE ONLY.
 * Thin exports over the reference's own rans_byte.h, which is self-contained C
 * and is compiled IN PLACE from /root/reference (never copied).  Built into
 * oracle/_ref/librefrans.so by oracle/Makefile when /root/reference exists; it
 * pins the oracle's restated rANS arithmetic (tests/test_oracle_rans.py). */
#include <stdint.h>
#include <string.h>
#include REF_RANS_HEADER

/* Encodes n entries the way ransmt.h:116-134 drives the primitives
 * (reverse order, raw bytes interleaved, 4-byte flush). */
int ref_rans_block(const uint16_t* entries, int n, uint8_t* out, uint8_t* scratch, int scratch_len)
{
    RansState r;
    uint8_t* end = scratch + scratch_len;
    uint8_t* p = end;
    int i;
    RansEncInit(&r);
    for (i = n - 1; i >= 0; i--) {
        if (entries[2*i]) RansEncPut(&r, &p, entries[2*i+1], entries[2*i], 12);
        else *--p = (uint8_t)entries[2*i+1];
    }
    RansEncFlush(&r, &p);
    memcpy(out, p, (size_t)(end - p));
    return (int)(end - p);
}

/* Decoder walk over the same bytes: returns the 12-bit values seen before each
 * advance, so a test can check them against the intervals. */
int ref_rans_decode_values(const uint8_t* in, const uint16_t* entries, int n, uint16_t* values)
{
    RansState r;
    uint8_t* p = (uint8_t*)in;
    int i;
    RansDecInit(&r, &p);
    for (i = 0; i < n; i++) {
        if (entries[2*i]) {
            values[i] = (uint16_t)RansDecGet(&r, 12);
            RansDecAdvance(&r, &p, entries[2*i+1], entries[2*i], 12);
        } else {
            values[i] = *p++;
        }
    }
    return (int)(p - in);
}

/* The reference's OTHER form of the encoder step (rans_byte.h:171-241, :255-278: RansEncSymbolInit / RansEncPutSymbol, which the
 * reference ships and never calls): the algebra the product's rANS kernels implement (csrc/scpr_model.hpp rans_rcp, k_rans, k_rans_s).
 * out = { x_max, rcp_freq, bias, cmpl_freq, rcp_shift } */
void ref_enc_symbol_init(uint32_t start, uint32_t freq, uint32_t* out)
{
    RansEncSymbol s;
    RansEncSymbolInit(&s, start, freq, 12);
    out[0] = s.x_max; out[1] = s.rcp_freq; out[2] = s.bias; out[3] = s.cmpl_freq; out[4] = s.rcp_shift;
}

/* one RansEncPutSymbol from state x: returns the new state, the bytes it emitted to bytes[0..*n) in emission order */
uint32_t ref_enc_put_symbol(uint32_t x, uint32_t start, uint32_t freq, uint8_t* bytes, int* n)
{
    RansEncSymbol s;
    RansState r = x;
    uint8_t buf[8];
    uint8_t* p = buf + 8;
    int i, k;
    RansEncSymbolInit(&s, start, freq, 12);
    RansEncPutSymbol(&r, &p, &s);
    k = (int)(buf + 8 - p);
    for (i = 0; i < k; i++) bytes[i] = buf[7 - i];   /* the byte written first sits at the highest address */
    *n = k;
    return r;
}
