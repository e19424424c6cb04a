// codec_streams.hpp — per-lane value streams over one 32-document (sub-)block, per codec
// Part of libtrinity_hip.so (MI355X / gfx950); included by trinity_hip.hip.  New code, no reference source.
#pragma once
#include "dev_stream.hpp"

constexpr int CODEC_GOOGLE = TRI_CODEC_GOOGLE;
constexpr int CODEC_LUCENE = TRI_CODEC_LUCENE;

// The engine's unit of decode is a run of <= 32 documents owned by one lane, whatever the codec:
//   GOOGLE  one codec block (google_codec.h:18): n-1 prefix-varint deltas, then n freqs, then the hits
//   LUCENE  a quarter of a 128-document block (lucene_codec.h:52-55): the directory built at upload carries one row per 32
//           documents, so a lane starts from its own base docID; in this repo's PFOR128 payload (include/pfor128.md) the 32
//           values of quarter s are exactly `width` words starting at word 1 + s * width; the <128-document tail of a
//           term is prefix-varint (delta, freq) pairs (lucene_codec.cpp:321-337)
// DeltaStream<CODEC>::next() yields the next document delta; FreqStream<CODEC>::next() the next frequency.

__device__ __forceinline__ uint32_t ld32u(const uint8_t *p) { // unaligned little-endian dword
        const uint32_t *q = (const uint32_t *)((uintptr_t)p & ~(uintptr_t)3);
        return __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)((uintptr_t)p & 3u));
}

// One quarter (32 values) of an ints() group, or one side of the varbyte tail.
struct LValStream {
        const uint8_t *wp;    // first packed word of this quarter
        const uint8_t *epos;  // exception index bytes
        const uint8_t *ehigh; // exception high-part stream
        uint64_t win;
        uint32_t width, mask, bitpos, k;
        uint32_t nexc, eb, ecur, next_j, j, sub;
        uint32_t equal_v;
        int mode; // 0 packed, 1 all-equal, 2 tail: deltas, 3 tail: freqs
        VbStream vb;

        // group_off: byte offset of the ints() group's header byte; returns nothing (see group_bytes for its length)
        __device__ __forceinline__ void init_group(const uint8_t *__restrict__ index, const uint32_t group_off, const uint32_t quarter) {
                const uint8_t *g = index + group_off;
                const uint32_t L = g[0];
                j = 0;
                sub = quarter;
                if (!L) { // lucene_codec.cpp:31-39 / 74-84: every value equal
                        mode = 1;
                        uint32_t len;
                        const uint64_t w = (uint64_t)ld32u(g + 1) | ((uint64_t)ld32u(g + 5) << 32);
                        equal_v = vb_decode(w, len);
                        return;
                }
                mode = 0;
                const uint32_t w0 = ld32u(g + 1);
                width = w0 & 0xffu;
                nexc = (w0 >> 8) & 0xffu;
                eb = (w0 >> 16) & 0xffu;
                mask = width >= 32 ? 0xffffffffu : ((1u << width) - 1u);
                wp = g + 1 + 4 * (1 + quarter * width);
                epos = g + 1 + 4 * (1 + 4 * width);
                ehigh = epos + 4 * ((nexc + 3) / 4);
                bitpos = 0;
                k = 2;
                win = width ? ((uint64_t)ld32u(wp) | ((uint64_t)ld32u(wp + 4) << 32)) : 0;
                ecur = 0;
                while (ecur < nexc && (uint32_t)(epos[ecur] >> 5) < quarter)
                        ++ecur;
                next_j = (ecur < nexc && (uint32_t)(epos[ecur] >> 5) == quarter) ? (epos[ecur] & 31u) : 255u;
        }
        __device__ __forceinline__ void init_tail(const uint8_t *__restrict__ index, const uint32_t off, const bool freqs) {
                mode = freqs ? 3 : 2;
                vb.init(index + off);
        }
        __device__ __forceinline__ uint32_t next() {
                if (mode == 1)
                        return equal_v;
                if (mode == 2) {
                        const uint32_t d = vb.next();
                        (void)vb.next();
                        return d;
                }
                if (mode == 3) {
                        (void)vb.next();
                        return vb.next();
                }
                uint32_t v = 0;
                if (width) {
                        const uint32_t sh = bitpos & 31u;
                        v = (uint32_t)(win >> sh) & mask;
                        const uint32_t nb = bitpos + width;
                        if ((nb >> 5) != (bitpos >> 5)) { // moved into the next word: slide the 64-bit window
                                win = (win >> 32) | ((uint64_t)ld32u(wp + 4 * k) << 32);
                                ++k;
                        }
                        bitpos = nb;
                }
                if (j == next_j) { // patch the exception's high part
                        const uint32_t eo = ecur * eb;
                        const uint64_t hw = (uint64_t)ld32u(ehigh + 4 * (eo >> 5)) | ((uint64_t)ld32u(ehigh + 4 * (eo >> 5) + 4) << 32);
                        const uint32_t high = (uint32_t)(hw >> (eo & 31u)) & (eb >= 32 ? 0xffffffffu : ((1u << eb) - 1u));
                        v |= width >= 32 ? 0u : (high << width);
                        ++ecur;
                        next_j = (ecur < nexc && (uint32_t)(epos[ecur] >> 5) == sub) ? (epos[ecur] & 31u) : 255u;
                }
                ++j;
                return v;
        }
};

// bytes of the ints() group starting at g (header byte included)
__device__ __forceinline__ uint32_t lucene_group_bytes(const uint8_t *g) {
        const uint32_t L = g[0];
        if (L)
                return 1 + 4 * L;
        const uint32_t b0 = g[1];
        return 1 + (b0 < 0x80 ? 1 : b0 < 0xc0 ? 2 : b0 < 0xe0 ? 3 : b0 < 0xf0 ? 4 : 5);
}

template <int CODEC>
struct DeltaStream;
template <int CODEC>
struct FreqStream;

template <>
struct DeltaStream<CODEC_GOOGLE> {
        VbStream s;
        __device__ __forceinline__ void init(const uint8_t *__restrict__ index, const DevTerm &, const uint32_t, const uint32_t off) { s.init(index + off); }
        __device__ __forceinline__ uint32_t next() { return s.next(); }
};
template <>
struct FreqStream<CODEC_GOOGLE> { // the freqs follow the n-1 deltas in the same byte stream
        VbStream s;
        __device__ __forceinline__ void init(const uint8_t *__restrict__, const DevTerm &, const uint32_t, const uint32_t, const DeltaStream<CODEC_GOOGLE> &after_deltas) {
                s = after_deltas.s;
        }
        __device__ __forceinline__ uint32_t next() { return s.next(); }
};

template <>
struct DeltaStream<CODEC_LUCENE> {
        LValStream s;
        __device__ __forceinline__ void init(const uint8_t *__restrict__ index, const DevTerm &t, const uint32_t b, const uint32_t off) {
                if (b < t.npfor)
                        s.init_group(index, off, b & 3u);
                else
                        s.init_tail(index, off, false);
        }
        __device__ __forceinline__ uint32_t next() { return s.next(); }
};
template <>
struct FreqStream<CODEC_LUCENE> {
        LValStream s;
        __device__ __forceinline__ void init(const uint8_t *__restrict__ index, const DevTerm &t, const uint32_t b, const uint32_t off, const DeltaStream<CODEC_LUCENE> &) {
                if (b < t.npfor)
                        s.init_group(index, off + lucene_group_bytes(index + off), b & 3u); // the freqs group follows the deltas group
                else
                        s.init_tail(index, off, true);
        }
        __device__ __forceinline__ uint32_t next() { return s.next(); }
};
