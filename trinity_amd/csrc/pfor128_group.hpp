// pfor128_group.hpp — ONE ints() group of the Lucene-shaped codec with this repo's PFOR128 payload (include/pfor128.md), as a pair of functions the
// device encoder's lanes (k_lencode.hpp) and the CPU tests (csrc/host/plan_host.cpp) share: plan (the framing's all-equal short form, lucene_codec.cpp:31-39,
// or the packed width / exceptions that cost the fewest words) and emit (the bytes).  The values come through a getter, so a lane needs no 128-entry array:
// deltas are recomputed from the documents, the second pass reads what the first one read.  Byte-identical to csrc/host/lucene_encoder.hpp::ints_encode
// (tests/test_fastpfor.py holds them against each other).  New code, no reference source.
#pragma once
#include "dev_structs.hpp"

struct Pfor128Plan {
        uint32_t b, nexc, eb; // packed width, exceptions, width of an exception's high part
        uint32_t bytes;       // of the whole group, the L byte included
        uint32_t v0;          // equal: the value
        bool equal;           // every value the same: `u8 0, varbyte v0`
};
TRI_HD inline uint32_t pf_bit_length(const uint32_t v) { return v ? 32u - (uint32_t)__builtin_clz(v) : 0u; }
TRI_HD inline uint32_t pf_vlen(const uint32_t v) { return v < (1u << 7) ? 1u : v < (1u << 14) ? 2u : v < (1u << 21) ? 3u : v < (1u << 28) ? 4u : 5u; }
// prefix varint (Switch/switch_compiler_aux.h:23-51)
TRI_HD inline uint8_t *pf_put_varbyte(uint8_t *o, const uint32_t v) {
        if (v < (1u << 7))
                *o++ = (uint8_t)v;
        else if (v < (1u << 14)) {
                *o++ = (uint8_t)(0x80u | (v >> 8));
                *o++ = (uint8_t)v;
        } else if (v < (1u << 21)) {
                *o++ = (uint8_t)(0xc0u | (v >> 16));
                *o++ = (uint8_t)v;
                *o++ = (uint8_t)(v >> 8);
        } else if (v < (1u << 28)) {
                *o++ = (uint8_t)(0xe0u | (v >> 24));
                *o++ = (uint8_t)(v >> 16);
                *o++ = (uint8_t)(v >> 8);
                *o++ = (uint8_t)v;
        } else {
                *o++ = 0xf0u; // (the five-byte form carries the value little-endian)
                *o++ = (uint8_t)v;
                *o++ = (uint8_t)(v >> 8);
                *o++ = (uint8_t)(v >> 16);
                *o++ = (uint8_t)(v >> 24);
        }
        return o;
}

template <class GET>
TRI_HD inline Pfor128Plan pfor128_plan(GET get) {
        uint32_t hist[33];
        for (uint32_t i = 0; i < 33; ++i)
                hist[i] = 0;
        const uint32_t v0 = get(0);
        bool eq = true;
        for (uint32_t i = 0; i < 128; ++i) {
                const uint32_t v = get(i);
                eq &= v == v0;
                ++hist[pf_bit_length(v)];
        }
        Pfor128Plan p{32, 0, 0, 0, v0, eq};
        if (eq) {
                p.bytes = 1 + pf_vlen(v0);
                return p;
        }
        uint32_t maxbl = 32;
        while (maxbl && !hist[maxbl])
                --maxbl;
        uint32_t best_cost = 4 * 32, above = 0;
        for (uint32_t l = 32; l > 0; --l)
                above += hist[l]; // values of bit length > 0
        // (b ascending, a strictly smaller cost wins: ties go to the smaller width, as in lucene_encoder.hpp)
        uint32_t nexc = above;
        for (uint32_t b = 0; b < 32; ++b) {
                if (b)
                        nexc -= hist[b]; // values of bit length > b
                const uint32_t eb = nexc ? maxbl - b : 0u;
                const uint32_t cost = 4 * b + (nexc + 3) / 4 + (nexc * eb + 31) / 32;
                if (cost < best_cost) {
                        best_cost = cost;
                        p.b = b, p.nexc = nexc, p.eb = eb;
                }
        }
        p.bytes = 1 + 4 * (1 + best_cost);
        return p;
}

// a little-endian bit stream into bytes: put(value, width), LSB first; flush() pads the last word
struct PfBits {
        uint8_t *o;
        uint64_t acc = 0;
        uint32_t n = 0;
        TRI_HD void word() {
                o[0] = (uint8_t)acc, o[1] = (uint8_t)(acc >> 8), o[2] = (uint8_t)(acc >> 16), o[3] = (uint8_t)(acc >> 24);
                o += 4;
                acc >>= 32;
                n -= 32;
        }
        TRI_HD void put(const uint64_t value, const uint32_t width) {
                acc |= value << n;
                n += width;
                if (n >= 32)
                        word();
        }
        TRI_HD void flush() { // (every section is a whole number of words)
                if (n)
                        n = 32, word();
                acc = 0;
                n = 0;
        }
};

template <class GET>
TRI_HD inline uint8_t *pfor128_emit(GET get, const Pfor128Plan &p, uint8_t *out) {
        if (p.equal) {
                *out++ = 0;
                return pf_put_varbyte(out, p.v0);
        }
        const uint32_t words = (p.bytes - 1) / 4;
        *out++ = (uint8_t)words;
        PfBits w{out};
        w.put(p.b | p.nexc << 8 | p.eb << 16, 32);
        const uint64_t mask = p.b == 32 ? 0xffffffffull : ((1ull << p.b) - 1);
        if (p.b)
                for (uint32_t i = 0; i < 128; ++i)
                        w.put(get(i) & mask, p.b);
        w.flush();
        if (p.nexc) {
                for (uint32_t i = 0; i < 128; ++i)
                        if (get(i) >> p.b)
                                w.put(i, 8);
                w.flush();
                for (uint32_t i = 0; i < 128; ++i)
                        if (get(i) >> p.b)
                                w.put(get(i) >> p.b, p.eb);
                w.flush();
        }
        return out + 4 * words;
}
