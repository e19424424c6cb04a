// fastpfor128.hpp — the two ints() payloads of the Lucene-shaped codec, host side.
//
// (1) FastPFor<4>, as the reference's own build writes it: lucene_codec.cpp:57-64 hands 128 values to
//     FastPForLib::FastPFor<4>::encodeArray and stores `u8 L` + the L 32-bit words it produced; :91-95 reads them back with decodeArray.
//     lemire/FastPFor is an un-vendored, unpinned submodule (Switch/ext/FastPFor, .gitmodules:1-3) that is ABSENT from the reference tree:
//     what follows restates the library's PUBLISHED algorithm (D. Lemire, L. Boytsov, "Decoding billions of integers per second through
//     vectorization", SPE 45(1), 2015, §"FastPFOR"; the scalar class `FastPFor` of headers/fastpfor.h) for exactly the call Trinity
//     makes — one block of 128 values per encodeArray / decodeArray.  PARITY UNPINNED: no byte written by the genuine library is
//     available here to check against; DESIGN.md §2 says so.  The words of such a call:
//
//         W[0]            128                       the value count encodeArray stores first
//         W[1]            n                         words from W[1] to the metadata (= 1 + 4b: the packed block follows W[1])
//         W[2 .. 2+4b)    the low b bits of v[0..127], four groups of 32 values, each group b words, value i of a group in stream bits
//                         [i*b, (i+1)*b), LSB first ("fastpackwithoutmask", horizontal)
//         W[1+n]          s                         bytes of the byte container that follows, padded to whole words:
//                            u8 b, u8 nexc, and when nexc > 0: u8 maxb, nexc x u8 position (ascending)
//         next word       bitmap                    bit k-1 set <=> exception values of width k (k = maxb - b, 2..32) follow
//         per set bit k   u32 count, then ceil(count / 32) groups of 32 values packed at k bits (the last group zero-padded)
//                         — the exceptions' high parts v >> b in position order.  Width 1 (maxb - b == 1) stores nothing: the high part is 1.
//
//     b is chosen by the library's cost model: exceptions cost 8 bits of position + (maxb - b) bits each, 8 bits for maxb, minus one bit
//     per exception when maxb - b == 1.  Any b gives a stream decodeArray reads; the model matters for byte identity only.
//
// (2) PFOR128, this repo's own payload (include/pfor128.md): the layout the KERNELS read (k_fused.hpp PfRegs, codec_streams.hpp).  A
//     segment whose groups carry FastPFor words is TRANSCODED group by group at upload (index_host.hpp): the device image — and with it
//     every kernel — only ever sees PFOR128.
//
// Host-only C++17, no HIP.  New code, no reference source (and no FastPFor source: there is none here).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace trif {
        constexpr uint32_t N = 128;
        inline uint32_t bit_length(uint32_t v) { return v ? 32u - (uint32_t)__builtin_clz(v) : 0u; }

        // 32 values at `b` bits, LSB first, into b words (fastpackwithoutmask / fastunpack)
        inline void pack32(const uint32_t *v, uint32_t *out, const uint32_t b) {
                for (uint32_t i = 0; i < b; ++i)
                        out[i] = 0;
                if (!b)
                        return;
                uint64_t bit = 0;
                for (uint32_t i = 0; i < 32; ++i, bit += b) {
                        const uint64_t x = b == 32 ? v[i] : (v[i] & ((1ull << b) - 1));
                        const uint32_t w = (uint32_t)(bit >> 5), sh = (uint32_t)(bit & 31);
                        out[w] |= (uint32_t)(x << sh);
                        if (sh + b > 32)
                                out[w + 1] |= (uint32_t)(x >> (32 - sh));
                }
        }
        inline void unpack32(const uint32_t *in, uint32_t *v, const uint32_t b) {
                if (!b) {
                        for (uint32_t i = 0; i < 32; ++i)
                                v[i] = 0;
                        return;
                }
                uint64_t bit = 0;
                for (uint32_t i = 0; i < 32; ++i, bit += b) {
                        const uint32_t w = (uint32_t)(bit >> 5), sh = (uint32_t)(bit & 31);
                        uint64_t x = in[w] >> sh;
                        if (sh + b > 32)
                                x |= (uint64_t)in[w + 1] << (32 - sh);
                        v[i] = (uint32_t)(b == 32 ? x : (x & ((1ull << b) - 1)));
                }
        }

        // ---- FastPFor<4>::encodeArray(v, 128, out, nwords): the words of ONE 128-value call
        inline void fastpfor_encode(const uint32_t *v, std::vector<uint32_t> &w) {
                // getBestBFromData: widths' histogram, then the cheapest b from the top down
                uint32_t freqs[33] = {};
                for (uint32_t i = 0; i < N; ++i)
                        ++freqs[bit_length(v[i])];
                uint32_t bestb = 32;
                while (bestb && !freqs[bestb])
                        --bestb;
                const uint32_t maxb = bestb;
                uint32_t bestcost = bestb * N, cexcept = 0, bestcexcept = 0;
                for (uint32_t b = bestb; b-- > 0;) {
                        cexcept += freqs[b + 1];
                        uint32_t cost = cexcept * 8 + cexcept * (maxb - b) + b * N + 8; // 8 bits per position, maxb - b per high part, 8 for maxb
                        if (maxb - b == 1)
                                cost -= cexcept; // (a one-bit high part is implied)
                        if (cost < bestcost) {
                                bestcost = cost;
                                bestb = b;
                                bestcexcept = cexcept;
                        }
                }
                w.clear();
                w.push_back(N);
                w.push_back(1 + 4 * bestb);
                w.resize(2 + 4 * bestb);
                for (uint32_t g = 0; g < 4; ++g)
                        pack32(v + 32 * g, w.data() + 2 + g * bestb, bestb);
                std::vector<uint8_t> bc{(uint8_t)bestb, (uint8_t)bestcexcept};
                std::vector<uint32_t> high;
                if (bestcexcept) {
                        bc.push_back((uint8_t)maxb);
                        for (uint32_t i = 0; i < N; ++i)
                                if (bestb < 32 && (v[i] >> bestb)) {
                                        bc.push_back((uint8_t)i);
                                        high.push_back(v[i] >> bestb);
                                }
                }
                w.push_back((uint32_t)bc.size());
                const size_t at = w.size();
                w.resize(at + (bc.size() + 3) / 4, 0u);
                memcpy(w.data() + at, bc.data(), bc.size());
                const uint32_t k = bestcexcept ? maxb - bestb : 0u;
                w.push_back(k >= 2 ? 1u << (k - 1) : 0u);
                if (k >= 2) {
                        w.push_back((uint32_t)high.size());
                        const size_t n = high.size();
                        high.resize((n + 31) / 32 * 32, 0u);
                        for (size_t j = 0; j < high.size(); j += 32) {
                                const size_t o = w.size();
                                w.resize(o + k);
                                pack32(high.data() + j, w.data() + o, k);
                        }
                }
        }

        // ---- FastPFor<4>::decodeArray over the L words of one ints() group -> 128 values.  false: not such a stream
        inline bool fastpfor_decode(const uint32_t *w, const uint32_t L, uint32_t *out) {
                if (L < 5 || w[0] != N)
                        return false;
                const uint32_t n = w[1];
                if (n < 1 || n > L || (n - 1) % 4 || (uint64_t)n + 3 > L) // (64-bit: n near 2^32 must not wrap past the check)
                        return false;
                const uint32_t bpacked = (n - 1) / 4;
                const uint32_t *meta = w + 1 + n;
                const uint32_t bcsize = meta[0];
                const uint32_t bcwords = (bcsize + 3) / 4;
                if (bcsize < 2 || bcsize > 4u * L || (uint64_t)n + 3 + bcwords > L)
                        return false;
                const uint8_t *bc = reinterpret_cast<const uint8_t *>(meta + 1);
                const uint32_t b = bc[0], nexc = bc[1];
                if (b != bpacked || b > 32 || nexc > N || (nexc ? 3 + nexc : 2u) != bcsize)
                        return false;
                for (uint32_t g = 0; g < 4; ++g)
                        unpack32(w + 2 + g * b, out + 32 * g, b);
                const uint32_t *p = meta + 1 + bcwords;
                const uint32_t bitmap = *p++;
                if (!nexc)
                        return bitmap == 0 && (uint32_t)(p - w) == L;
                const uint32_t maxb = bc[2];
                if (maxb <= b || maxb > 32)
                        return false;
                const uint32_t k = maxb - b;
                std::vector<uint32_t> high;
                if (k >= 2) {
                        if (bitmap != 1u << (k - 1) || (uint32_t)(p - w) >= L)
                                return false;
                        const uint32_t cnt = *p++;
                        const uint32_t groups = (cnt + 31) / 32;
                        if (cnt != nexc || (uint64_t)(p - w) + (uint64_t)groups * k != L)
                                return false;
                        high.resize((size_t)groups * 32);
                        for (uint32_t g = 0; g < groups; ++g)
                                unpack32(p + g * k, high.data() + 32 * g, k);
                } else if (bitmap != 0 || (uint32_t)(p - w) != L)
                        return false;
                uint32_t prev = 0;
                for (uint32_t e = 0; e < nexc; ++e) {
                        const uint32_t pos = bc[3 + e];
                        if (pos >= N || (e && pos <= prev))
                                return false;
                        prev = pos;
                        out[pos] |= (k >= 2 ? high[e] : 1u) << b;
                }
                return true;
        }

        // ---- PFOR128 (include/pfor128.md): the whole ints() group — L byte and words — appended to `out`.  The caller has ruled out the
        //      all-equal form (lucene_codec.cpp:31-39)
        inline void pfor128_encode(const uint32_t *v, std::vector<uint8_t> &out) {
                struct Choice {
                        uint32_t b, nexc, eb, cost;
                } best{32, 0, 0, 4 * 32};
                for (uint32_t b = 0; b < 32; ++b) {
                        uint32_t nexc = 0, mx = 0;
                        for (uint32_t i = 0; i < N; ++i) {
                                const uint32_t h = v[i] >> b;
                                nexc += h != 0;
                                mx = std::max(mx, h);
                        }
                        const uint32_t eb = bit_length(mx);
                        const uint32_t cost = 4 * b + (nexc + 3) / 4 + (nexc * eb + 31) / 32;
                        if (cost < best.cost)
                                best = {b, nexc, eb, cost};
                }
                std::vector<uint32_t> words(1 + 4 * best.b + (best.nexc + 3) / 4 + (best.nexc * best.eb + 31) / 32, 0u);
                words[0] = best.b | best.nexc << 8 | best.eb << 16;
                for (uint32_t g = 0; g < 4; ++g)
                        pack32(v + 32 * g, words.data() + 1 + g * best.b, best.b);
                if (best.nexc) {
                        uint8_t *pos = reinterpret_cast<uint8_t *>(words.data() + 1 + 4 * best.b);
                        uint32_t *high = words.data() + 1 + 4 * best.b + (best.nexc + 3) / 4;
                        uint32_t e = 0;
                        uint64_t bit = 0;
                        for (uint32_t i = 0; i < N; ++i)
                                if (best.b < 32 && (v[i] >> best.b)) {
                                        pos[e++] = (uint8_t)i;
                                        const uint64_t x = v[i] >> best.b;
                                        const uint32_t wd = (uint32_t)(bit >> 5), sh = (uint32_t)(bit & 31);
                                        high[wd] |= (uint32_t)(x << sh);
                                        if (sh + best.eb > 32)
                                                high[wd + 1] |= (uint32_t)(x >> (32 - sh));
                                        bit += best.eb;
                                }
                }
                out.push_back((uint8_t)words.size());
                const auto *p = reinterpret_cast<const uint8_t *>(words.data());
                out.insert(out.end(), p, p + words.size() * 4);
        }

        // which payload an ints() group (p: its L byte, L != 0) carries: FastPFor's first word is the value count 128; PFOR128's first word has
        // the packed width (<= 32) in its low byte
        inline bool group_is_fastpfor(const uint8_t *p) {
                uint32_t w0;
                memcpy(&w0, p + 1, 4);
                return w0 == N;
        }
} // namespace trif
