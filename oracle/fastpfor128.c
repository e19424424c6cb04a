/* fastpfor128.c — TEST INFRASTRUCTURE (the oracle): lemire/FastPFor's `FastPFor<4>` for the one call Trinity makes of it.
 *
 * The reference's lucene_codec stores every non-constant group of 128 integers as `u8 L` + the L words that
 * FastPForLib::FastPFor<4>::encodeArray(values, 128, out, L) produced (lucene_codec.cpp:57-64) and reads them back with decodeArray
 * (:91-95).  The library is a git submodule (Switch/ext/FastPFor, .gitmodules:1-3) that is NOT in /root/reference and has no pinned
 * version visible there, so this file restates the library's PUBLISHED algorithm — Lemire & Boytsov, "Decoding billions of integers per
 * second through vectorization", Softw. Pract. Exper. 45(1), 2015, the FastPFOR scheme; the scalar class FastPFor of
 * headers/fastpfor.h — for a single block of BlockSize = 4 * 32 = 128 values in a single page.
 *
 * PARITY UNPINNED: there is no golden vector, no reference test and no library build to check these bytes against.  What
 * the tests do check: this restatement and the product's (trinity_amd/csrc/fastpfor128.hpp, written separately: whole-word packers
 * there, a bit-at-a-time stream here) produce the same words and decode each other's; a segment written with these words loads on
 * the GPU and answers every query like the same corpus in the other encodings.
 *
 * Layout of one call's output (32-bit little-endian words):
 *   [0]            128: encodeArray stores the value count
 *   [1]            offset, in words from [1], of the metadata behind the packed block = 1 + 4 b
 *   [2 .. 2+4b)    packed block: the low b bits of every value, 32 values per group of b words, LSB first
 *   then           byte-container size s; ceil(s / 4) words holding: b, #exceptions, and if any: maxb, then each exception's position
 *   then           bitmap: bit (k - 1) set when exception values of width k = maxb - b >= 2 follow
 *   then           for that k: count, and the values (v >> b) packed at k bits in groups of 32 (zero-padded)
 * A width-1 exception (maxb - b == 1) stores no value: its high part can only be 1.
 * b: from the top width down, the first strictly cheaper of  128 b + #exc * (8 + maxb - b) + 8  (minus #exc when maxb - b == 1).
 */
#include <stdint.h>
#include <string.h>

static unsigned width_of(uint32_t v) {
        unsigned n = 0;
        while (v) {
                ++n;
                v >>= 1;
        }
        return n;
}

/* append `nbits` of value to a word stream, LSB first */
static void put_bits(uint32_t *words, uint64_t *bitpos, uint32_t value, unsigned nbits) {
        for (unsigned i = 0; i < nbits; ++i, ++*bitpos)
                if ((value >> i) & 1u)
                        words[*bitpos >> 5] |= 1u << (*bitpos & 31);
}
static uint32_t get_bits(const uint32_t *words, uint64_t *bitpos, unsigned nbits) {
        uint32_t v = 0;
        for (unsigned i = 0; i < nbits; ++i, ++*bitpos)
                v |= ((words[*bitpos >> 5] >> (*bitpos & 31)) & 1u) << i;
        return v;
}

/* out: room for 160 words.  Returns the word count L. */
uint32_t to_fastpfor_encode128(const uint32_t *v, uint32_t *out) {
        unsigned hist[33] = {0};
        for (int i = 0; i < 128; ++i)
                ++hist[width_of(v[i])];
        unsigned maxb = 32;
        while (maxb && !hist[maxb])
                --maxb;
        unsigned b = maxb, nexc = 0;
        {
                unsigned best = maxb * 128, c = 0;
                for (int cand = (int)maxb - 1; cand >= 0; --cand) {
                        c += hist[cand + 1];
                        unsigned cost = (unsigned)cand * 128 + c * (8 + maxb - (unsigned)cand) + 8;
                        if (maxb - (unsigned)cand == 1)
                                cost -= c;
                        if (cost < best) {
                                best = cost;
                                b = (unsigned)cand;
                                nexc = c;
                        }
                }
        }
        memset(out, 0, 160 * sizeof(uint32_t));
        out[0] = 128;
        out[1] = 1 + 4 * b;
        uint64_t bit = 0;
        for (int i = 0; i < 128; ++i) /* (32 values of b bits fill b words exactly: the groups are word-aligned by themselves) */
                put_bits(out + 2, &bit, b == 32 ? v[i] : (v[i] & ((1u << b) - 1u)), b);
        uint32_t *p = out + 2 + 4 * b;
        uint8_t bytes[3 + 128];
        unsigned nb = 0;
        bytes[nb++] = (uint8_t)b;
        bytes[nb++] = (uint8_t)nexc;
        if (nexc) {
                bytes[nb++] = (uint8_t)maxb;
                for (int i = 0; i < 128; ++i)
                        if (b < 32 && (v[i] >> b))
                                bytes[nb++] = (uint8_t)i;
        }
        *p++ = nb;
        memcpy(p, bytes, nb);
        p += (nb + 3) / 4;
        const unsigned k = nexc ? maxb - b : 0;
        *p++ = k >= 2 ? 1u << (k - 1) : 0u;
        if (k >= 2) {
                *p++ = nexc;
                bit = 0;
                for (int i = 0; i < 128; ++i)
                        if (v[i] >> b)
                                put_bits(p, &bit, v[i] >> b, k);
                p += ((nexc + 31) / 32) * k; /* whole groups of 32 values */
        }
        return (uint32_t)(p - out);
}

/* 1: decoded; 0: not a FastPFor<4> stream of one 128-value block */
int to_fastpfor_decode128(const uint32_t *w, uint32_t L, uint32_t *v) {
        if (L < 5 || w[0] != 128 || w[1] < 1 || w[1] > L || (w[1] - 1) % 4 || (uint64_t)w[1] + 3 > L) /* (64-bit: no wrap) */
                return 0;
        const unsigned b = (w[1] - 1) / 4;
        if (b > 32)
                return 0;
        uint64_t bit = 0;
        for (int i = 0; i < 128; ++i)
                v[i] = get_bits(w + 2, &bit, b);
        const uint32_t *p = w + 1 + w[1];
        const uint32_t nb = *p++;
        if (nb < 2 || nb > 4u * L || (uint64_t)(p - w) + (nb + 3) / 4 + 1 > L)
                return 0;
        const uint8_t *bytes = (const uint8_t *)p;
        p += (nb + 3) / 4;
        const unsigned nexc = bytes[1];
        if (bytes[0] != b || nb != (nexc ? 3 + nexc : 2))
                return 0;
        const uint32_t bitmap = *p++;
        if (!nexc)
                return bitmap == 0 && (uint32_t)(p - w) == L;
        const unsigned maxb = bytes[2];
        if (maxb <= b || maxb > 32)
                return 0;
        const unsigned k = maxb - b;
        bit = 0;
        if (k >= 2) {
                if (bitmap != 1u << (k - 1) || (uint32_t)(p - w) >= L || *p != nexc || (uint32_t)(p + 1 - w) + ((nexc + 31) / 32) * k != L)
                        return 0;
                ++p;
        } else if (bitmap || (uint32_t)(p - w) != L)
                return 0;
        for (unsigned e = 0; e < nexc; ++e) {
                const unsigned pos = bytes[3 + e];
                if (pos >= 128 || (e && pos <= bytes[2 + e]))
                        return 0;
                v[pos] |= (k >= 2 ? get_bits(p, &bit, k) : 1u) << b;
        }
        return 1;
}
