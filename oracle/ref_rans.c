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
