// google_encoder.hpp — host-side writer for the "GOOGLE" codec segment format.
//
// Write side of the codec seam: mirrors the call protocol of Trinity::Codecs::Encoder
// (codecs.h:176-200: begin_term / begin_document / new_hit / end_document / end_term) and produces chunks that
// are byte-identical to Trinity::Codecs::Google::Encoder's (google_codec.cpp:9-176; format in SURVEY.md A.2).
// The write side is out of scope as a GPU feature; it exists because the engine, its tests
// and bench.py need segments to execute against.  New code; independent of oracle/.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace trinity_amd {
        struct term_index_ctx { // codecs.h:17-55
                uint32_t documents{0};
                uint32_t offset{0}, size{0}; // indexChunk
        };

        namespace Codecs {
                // Prefix varint: the length sits in the leading bits of the first byte (Switch/switch_compiler_aux.h:23-51)
                inline void put_varbyte32(std::vector<uint8_t> &o, uint32_t v) {
                        if (v < (1u << 7))
                                o.push_back(uint8_t(v));
                        else if (v < (1u << 14)) {
                                o.push_back(uint8_t(0x80u | (v >> 8)));
                                o.push_back(uint8_t(v));
                        } else if (v < (1u << 21)) {
                                o.push_back(uint8_t(0xc0u | (v >> 16)));
                                o.push_back(uint8_t(v));
                                o.push_back(uint8_t(v >> 8));
                        } else if (v < (1u << 28)) {
                                o.push_back(uint8_t(0xe0u | (v >> 24)));
                                o.push_back(uint8_t(v >> 16));
                                o.push_back(uint8_t(v >> 8));
                                o.push_back(uint8_t(v));
                        } else {
                                o.push_back(0xf0u);
                                for (int s = 0; s < 32; s += 8)
                                        o.push_back(uint8_t(v >> s));
                        }
                }

                namespace Google {
                        constexpr uint32_t N = 32;                  // google_codec.h:18
                        constexpr uint32_t SKIPLIST_STEP = 256 / N; // google_codec.h:19

                        struct IndexSession {
                                std::vector<uint8_t> indexOut; // codecs.h:75
                        };

                        class Encoder {
                                IndexSession *const sess;
                                std::vector<uint8_t> skiplist, body, hits;
                                uint32_t deltas[N], freqs[N];
                                uint32_t inBlock{0}, prevBlockLast{0}, curDoc{0}, lastDoc{0}, lastPos{0};
                                uint8_t curPayloadSize{0};
                                uint32_t countdown{SKIPLIST_STEP}; // survives across terms, as in the reference (google_codec.h:57)
                                uint32_t termStart{0}, termDocs{0};

                                void flush_block() {
                                        auto &out = sess->indexOut;
                                        body.clear();
                                        for (uint32_t i = 0; i + 1 < inBlock; ++i)
                                                put_varbyte32(body, deltas[i]);
                                        for (uint32_t i = 0; i < inBlock; ++i)
                                                put_varbyte32(body, freqs[i]);
                                        if (--countdown == 0) {
                                                if (skiplist.size() / 8 < UINT16_MAX) {
                                                        const uint32_t rec[2] = {prevBlockLast, uint32_t(out.size() - termStart)};
                                                        const auto *b = reinterpret_cast<const uint8_t *>(rec);
                                                        skiplist.insert(skiplist.end(), b, b + 8);
                                                }
                                                countdown = SKIPLIST_STEP;
                                        }
                                        put_varbyte32(out, curDoc - prevBlockLast);
                                        put_varbyte32(out, uint32_t(body.size() + hits.size()));
                                        out.push_back(uint8_t(inBlock));
                                        out.insert(out.end(), body.begin(), body.end());
                                        out.insert(out.end(), hits.begin(), hits.end());
                                        hits.clear();
                                        prevBlockLast = curDoc;
                                        inBlock = 0;
                                }

                              public:
                                explicit Encoder(IndexSession *s)
                                    : sess{s} {}

                                void begin_term() {
                                        inBlock = 0;
                                        lastDoc = prevBlockLast = 0;
                                        hits.clear();
                                        termDocs = 0;
                                        termStart = uint32_t(sess->indexOut.size());
                                        sess->indexOut.push_back(0); // u16 skiplist entry count, patched by end_term
                                        sess->indexOut.push_back(0);
                                }
                                void begin_document(uint32_t id) {
                                        if (!id || id <= lastDoc)
                                                throw std::invalid_argument("document IDs must be > 0 and strictly ascending per term");
                                        curDoc = id;
                                        lastPos = 0;
                                        curPayloadSize = 0;
                                        freqs[inBlock] = 0;
                                }
                                void new_hit(uint32_t pos, const uint8_t *payload = nullptr, uint8_t payloadLen = 0) {
                                        if (!pos && !payloadLen)
                                                return;
                                        const uint32_t d = pos - lastPos;
                                        ++freqs[inBlock];
                                        if (payloadLen != curPayloadSize) {
                                                put_varbyte32(hits, (d << 1) | 1u);
                                                hits.push_back(payloadLen);
                                                curPayloadSize = payloadLen;
                                        } else
                                                put_varbyte32(hits, d << 1);
                                        if (payloadLen)
                                                hits.insert(hits.end(), payload, payload + payloadLen);
                                        lastPos = pos;
                                }
                                void end_document() {
                                        deltas[inBlock++] = curDoc - lastDoc;
                                        if (inBlock == N)
                                                flush_block();
                                        lastDoc = curDoc;
                                        ++termDocs;
                                }
                                void end_term(term_index_ctx *tctx) {
                                        if (inBlock)
                                                flush_block();
                                        auto &out = sess->indexOut;
                                        const uint16_t entries = uint16_t(skiplist.size() / 8);
                                        out.insert(out.end(), skiplist.begin(), skiplist.end());
                                        std::memcpy(out.data() + termStart, &entries, 2);
                                        tctx->offset = termStart;
                                        tctx->size = uint32_t(out.size() - termStart);
                                        tctx->documents = termDocs;
                                        skiplist.clear();
                                }
                        };
                } // namespace Google
        }         // namespace Codecs
} // namespace trinity_amd
