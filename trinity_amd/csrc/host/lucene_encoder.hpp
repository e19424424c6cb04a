// lucene_encoder.hpp — host-side writer for the "LUCENE"-shaped segment format.
//
// Write side of the codec seam, same call protocol as Trinity::Codecs::Encoder (codecs.h:176-200).  The container is the
// reference's (lucene_codec.cpp:163-388): per term a 14-byte header {hitsDataOffset, sumHits,
// positionsChunkSize, skiplistSize}, 128-document blocks as two ints() groups (doc deltas, freqs), a varbyte tail, one
// 22-byte skiplist entry per block; hit positions go to a separate hits.data stream in blocks of 128 hits.  The ints()
// payload is this repo's PFOR128 (include/pfor128.md) because the reference's (lemire/FastPFor) is absent: PARITY
// UNPINNED for those bytes.  New code; independent of oracle/.
#pragma once
#include "../fastpfor128.hpp"
#include "google_encoder.hpp"
#include <algorithm>

namespace trinity_amd {
        namespace Codecs {
                namespace Lucene {
                        constexpr uint32_t BLOCK_SIZE = 128; // lucene_codec.h:52-55

                        inline uint32_t bit_length(uint32_t v) { return v ? 32u - uint32_t(__builtin_clz(v)) : 0u; }

                        struct BitWriter {
                                std::vector<uint32_t> &w;
                                size_t base;
                                uint64_t bit{0};
                                BitWriter(std::vector<uint32_t> &words, size_t nwords)
                                    : w{words}, base{words.size()} { w.resize(base + nwords, 0u); }
                                void put(uint64_t value, uint32_t width) {
                                        if (!width)
                                                return;
                                        const size_t i = base + size_t(bit >> 5);
                                        const uint32_t sh = uint32_t(bit & 31);
                                        w[i] |= uint32_t(value << sh);
                                        if (sh + width > 32)
                                                w[i + 1] |= uint32_t(value >> (32 - sh));
                                        bit += width;
                                }
                        };

                        // which words an ints() group carries: this repo's PFOR128 (what the kernels read), or FastPFor<4>'s as the reference's own
                        // build writes them (csrc/fastpfor128.hpp: a segment a genuine Trinity opens)
                        enum class Payload { PFOR128, FastPFor };

                        // ints() group of 128 values (lucene_codec.cpp:26-66 framing)
                        inline void ints_encode(const uint32_t *v, std::vector<uint8_t> &out, const Payload payload = Payload::PFOR128) {
                                if (std::all_of(v + 1, v + BLOCK_SIZE, [&](uint32_t x) { return x == v[0]; })) {
                                        out.push_back(0);
                                        put_varbyte32(out, v[0]);
                                        return;
                                }
                                if (payload == Payload::FastPFor) {
                                        std::vector<uint32_t> w;
                                        trif::fastpfor_encode(v, w);
                                        out.push_back(uint8_t(w.size())); // lucene_codec.cpp:63: the word count (<= 135 for 128 values)
                                        const auto *p = reinterpret_cast<const uint8_t *>(w.data());
                                        out.insert(out.end(), p, p + w.size() * 4);
                                        return;
                                }
                                struct Choice {
                                        uint32_t b, nexc, eb, cost;
                                } best{32, 0, 0, 4 * 32};
                                for (uint32_t b = 0; b < 32; ++b) {
                                        uint32_t nexc = 0, mx = 0;
                                        for (uint32_t i = 0; i < BLOCK_SIZE; ++i) {
                                                const uint32_t h = v[i] >> b;
                                                nexc += h != 0;
                                                mx = std::max(mx, h);
                                        }
                                        const uint32_t eb = bit_length(mx);
                                        const uint32_t cost = 4 * b + (nexc + 3) / 4 + (nexc * eb + 31) / 32;
                                        if (cost < best.cost)
                                                best = {b, nexc, eb, cost};
                                }
                                std::vector<uint32_t> words;
                                words.push_back(best.b | best.nexc << 8 | best.eb << 16);
                                {
                                        BitWriter packed(words, 4 * best.b);
                                        const uint64_t mask = best.b == 32 ? 0xffffffffull : ((1ull << best.b) - 1);
                                        for (uint32_t i = 0; i < BLOCK_SIZE; ++i)
                                                packed.put(v[i] & mask, best.b);
                                }
                                if (best.nexc) {
                                        BitWriter pos(words, (best.nexc + 3) / 4);
                                        for (uint32_t i = 0; i < BLOCK_SIZE; ++i)
                                                if (best.b < 32 && (v[i] >> best.b))
                                                        pos.put(i, 8);
                                        BitWriter high(words, (best.nexc * best.eb + 31) / 32);
                                        for (uint32_t i = 0; i < BLOCK_SIZE; ++i)
                                                if (best.b < 32 && (v[i] >> best.b))
                                                        high.put(v[i] >> best.b, best.eb);
                                }
                                out.push_back(uint8_t(words.size()));
                                const auto *p = reinterpret_cast<const uint8_t *>(words.data());
                                out.insert(out.end(), p, p + words.size() * 4);
                        }

                        struct IndexSession {
                                std::vector<uint8_t> indexOut;     // codecs.h:75
                                std::vector<uint8_t> positionsOut; // lucene_codec.h:88 -> hits.data
                        };

                        class Encoder {
                                struct SkipEntry { // lucene_codec.h:128-135
                                        uint32_t indexOffset, lastDocID, lastHitsBlockOffset, totalDocumentsSoFar, lastHitsBlockTotalHits;
                                        uint16_t curHitsBlockHits;
                                };
                                IndexSession *const sess;
                                const Payload payload;
                                std::vector<SkipEntry> skiplist;
                                SkipEntry cur{};
                                uint32_t deltas[BLOCK_SIZE], freqs[BLOCK_SIZE], hitPos[BLOCK_SIZE], hitLen[BLOCK_SIZE];
                                uint32_t lastDoc{0}, buffered{0}, hitsInBlock{0}, sumHits{0}, termDocs{0}, termStart{0}, posStart{0};
                                uint32_t lastHitsBlockOffset{0}, lastHitsBlockTotalHits{0}, lastPos{0};

                                template <class T>
                                static void put(std::vector<uint8_t> &o, T v) {
                                        const auto *p = reinterpret_cast<const uint8_t *>(&v);
                                        o.insert(o.end(), p, p + sizeof(T));
                                }
                                void flush_docs_block() {
                                        if (skiplist.size() < UINT16_MAX) // SKIPLIST_STEP == 1: every block (lucene_codec.h:57)
                                                skiplist.push_back(cur);
                                        ints_encode(deltas, sess->indexOut, payload);
                                        ints_encode(freqs, sess->indexOut, payload);
                                        buffered = 0;
                                }

                              public:
                                explicit Encoder(IndexSession *s, const Payload p = Payload::PFOR128)
                                    : sess{s}, payload{p} {}
                                void begin_term() {
                                        lastDoc = hitsInBlock = sumHits = buffered = termDocs = 0;
                                        termStart = uint32_t(sess->indexOut.size());
                                        posStart = uint32_t(sess->positionsOut.size());
                                        lastHitsBlockOffset = lastHitsBlockTotalHits = 0;
                                        skiplist.clear();
                                        put<uint32_t>(sess->indexOut, posStart);
                                        put<uint32_t>(sess->indexOut, 0); // sumHits, patched by end_term
                                        put<uint32_t>(sess->indexOut, 0); // positions chunk size
                                        put<uint16_t>(sess->indexOut, 0); // skiplist entries
                                }
                                void begin_document(uint32_t id) {
                                        if (id <= lastDoc)
                                                throw std::invalid_argument("document IDs must be > 0 and strictly ascending per term");
                                        if (buffered == BLOCK_SIZE)
                                                flush_docs_block();
                                        if (!buffered)
                                                cur = {uint32_t(sess->indexOut.size()) - termStart, lastDoc, lastHitsBlockOffset, termDocs, lastHitsBlockTotalHits, uint16_t(hitsInBlock)};
                                        deltas[buffered] = id - lastDoc;
                                        freqs[buffered] = 0;
                                        ++termDocs;
                                        lastDoc = id;
                                        lastPos = 0;
                                }
                                void new_hit(uint32_t pos) { // payload-less hits only (the synthetic corpus carries none)
                                        if (!pos)
                                                return;
                                        ++freqs[buffered];
                                        hitPos[hitsInBlock] = pos - lastPos;
                                        hitLen[hitsInBlock] = 0;
                                        lastPos = pos;
                                        if (++hitsInBlock == BLOCK_SIZE) {
                                                sumHits += hitsInBlock;
                                                ints_encode(hitPos, sess->positionsOut, payload);
                                                ints_encode(hitLen, sess->positionsOut, payload);
                                                put_varbyte32(sess->positionsOut, 0); // payload bytes of this block
                                                lastHitsBlockTotalHits = sumHits;
                                                lastHitsBlockOffset = uint32_t(sess->positionsOut.size()) - posStart;
                                                hitsInBlock = 0;
                                        }
                                }
                                void end_document() { ++buffered; }
                                void end_term(term_index_ctx *tctx) {
                                        auto &out = sess->indexOut;
                                        sumHits += hitsInBlock;
                                        if (buffered == BLOCK_SIZE)
                                                flush_docs_block();
                                        else
                                                for (uint32_t i = 0; i < buffered; ++i) {
                                                        put_varbyte32(out, deltas[i]);
                                                        put_varbyte32(out, freqs[i]);
                                                }
                                        for (uint32_t i = 0; i < hitsInBlock; ++i) // tail hits: (posDelta << 1 | newLen) [len]; no payloads here
                                                put_varbyte32(sess->positionsOut, hitPos[i] << 1);
                                        const uint32_t posSize = uint32_t(sess->positionsOut.size()) - posStart;
                                        const uint16_t nskip = uint16_t(skiplist.size());
                                        std::memcpy(out.data() + termStart + 4, &sumHits, 4);
                                        std::memcpy(out.data() + termStart + 8, &posSize, 4);
                                        std::memcpy(out.data() + termStart + 12, &nskip, 2);
                                        for (const auto &e : skiplist) {
                                                put<uint32_t>(out, e.indexOffset);
                                                put<uint32_t>(out, e.lastDocID);
                                                put<uint32_t>(out, e.lastHitsBlockOffset);
                                                put<uint32_t>(out, e.totalDocumentsSoFar);
                                                put<uint32_t>(out, e.lastHitsBlockTotalHits);
                                                put<uint16_t>(out, e.curHitsBlockHits);
                                        }
                                        skiplist.clear();
                                        tctx->documents = termDocs;
                                        tctx->offset = termStart;
                                        tctx->size = uint32_t(out.size()) - termStart;
                                }
                        };
                } // namespace Lucene
        }         // namespace Codecs
} // namespace trinity_amd
