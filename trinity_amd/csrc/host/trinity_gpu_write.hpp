// trinity_gpu_write.hpp — the write side of the host-side C++ operator surface: Trinity's SegmentIndexSession and the codec's merge, keeping
// the reference's names and argument meaning (indexer.h:19-230, codecs.h:120-200, merge.h), with tri_commit_google / tri_commit_lucene / tri_merge_google
// (include/trinity_hip.h) underneath.  Nothing here sorts, encodes or walks postings on the CPU: a session buffers what the application inserts, in insertion
// order, and commit() is one call into the engine.  Files, the terms dictionary's on-disk form and fsync stay with the application (out of scope: storage).
// New code — no reference source.
#pragma once
#include "trinity_gpu.hpp"
#include <string_view>
#include <unordered_set>
#include <utility>

namespace trinity_amd {
        // What persist_segment is handed (indexer.cpp:241-300): the encoded `index` (and `hits.data` for the Lucene-shaped codec), the dictionary entries
        // commit built (indexer.cpp:521-533: term string -> term_index_ctx) and the field statistics it added up
        struct committed_segment {
                std::vector<uint8_t> index, hits;
                std::vector<std::pair<std::string, term_index_ctx>> terms; // in the order commit encoded them (bucket termID & 31, then termID)
                tri_commit_stats stats{};
        };

        // Trinity::SegmentIndexSession (indexer.h:19-230)
        class SegmentIndexSession final {
              public:
                enum class Codec { Google, Lucene };

              private:
                tri_dev *dev;
                Codec codec;
                // the session's postings in insertion order: one entry per (document, term) — what commit_document_impl serialises (indexer.cpp:33-111)
                std::vector<uint32_t> termIDs, docIDs, freqs;
                std::vector<uint16_t> positions;
                std::vector<uint8_t> payloadLens;
                std::vector<uint64_t> payloads;
                bool anyPayload{false};
                uint64_t droppedHits{0}; // hits at position 0 without a payload: counted, never stored (google_codec.cpp:42-45)
                std::unordered_map<std::string, uint32_t> dictionary; // transient ids, first seen first: 1, 2, ... (indexer.cpp:161-185)
                std::vector<std::string> invDict;
                std::unordered_set<isrc_docid_t> tracked; // SegmentIndexSession::track (indexer.cpp:187-217)
                struct pending_hit {
                        uint32_t termID;
                        tokenpos_t position;
                        uint8_t payloadLen;
                        uint64_t payload;
                };

              public:
                // indexer.h:96-160.  Buffers the document's hits; SegmentIndexSession::insert() commits them to the session
                struct document_proxy final {
                        SegmentIndexSession &sess;
                        const isrc_docid_t did;
                        std::vector<pending_hit> hits;

                        uint32_t term_id(const std::string_view term) { return sess.term_id(term); }
                        void insert(const uint32_t termID, const tokenpos_t position, const uint8_t *payload = nullptr, const uint8_t payloadSize = 0) {
                                if (!termID)
                                        throw invalid_argument("document_proxy::insert: term id 0 (indexer.cpp:15)");
                                if (payloadSize > sizeof(uint64_t))
                                        throw invalid_argument("document_proxy::insert: payloads hold at most 8 bytes (indexer.cpp:26)");
                                uint64_t v = 0;
                                if (payloadSize)
                                        std::memcpy(&v, payload, payloadSize);
                                hits.push_back({termID, position, payloadSize, v});
                        }
                        void insert(const std::string_view term, const tokenpos_t position, const uint8_t *payload = nullptr, const uint8_t payloadSize = 0) {
                                insert(term_id(term), position, payload, payloadSize);
                        }
                        template <typename T>
                        void insert(const uint32_t termID, const tokenpos_t position, const T &v) { // indexer.h:128-131
                                static_assert(sizeof(T) <= sizeof(uint64_t));
                                insert(termID, position, reinterpret_cast<const uint8_t *>(&v), uint8_t(sizeof(T)));
                        }
                };

                explicit SegmentIndexSession(tri_dev *d, const Codec c = Codec::Google) : dev{d}, codec{c} {}

                uint32_t term_id(const std::string_view term) { // indexer.cpp:161-185
                        if (term.empty())
                                throw invalid_argument("SegmentIndexSession::term_id: empty term");
                        const auto it = dictionary.emplace(std::string(term), 0u);
                        if (it.second) {
                                it.first->second = uint32_t(dictionary.size());
                                invDict.emplace_back(term);
                        }
                        return it.first->second;
                }
                std::string_view term(const uint32_t id) const { return id && id <= invDict.size() ? std::string_view(invDict[id - 1]) : std::string_view(); } // indexer.cpp:152-156

                document_proxy begin(const isrc_docid_t documentID) { return {*this, documentID, {}}; } // indexer.cpp:229-234

                // indexer.h:192-197 -> commit_document_impl (indexer.cpp:33-111): the document's hits grouped by term, positions ascending; a document goes in once
                void insert(document_proxy &proxy) {
                        if (!proxy.did)
                                throw invalid_argument("SegmentIndexSession::insert: document 0");
                        if (!tracked.insert(proxy.did).second)
                                throw data_error("Already committed document " + std::to_string(proxy.did)); // indexer.cpp:219-222
                        auto &h = proxy.hits;
                        std::stable_sort(h.begin(), h.end(), [](const pending_hit &a, const pending_hit &b) { return a.termID < b.termID || (a.termID == b.termID && a.position < b.position); });
                        for (size_t i = 0; i < h.size();) {
                                size_t e = i;
                                uint32_t counted = 0;
                                for (; e < h.size() && h[e].termID == h[i].termID; ++e) {
                                        // a hit at position 0 without a payload is not stored (google_codec.cpp:42-45): the posting's frequency counts the others
                                        if (!h[e].position && !h[e].payloadLen) {
                                                ++droppedHits; // (commit still counts it: defaultFieldStats.sumTermHits += hitsCnt, indexer.cpp:447)
                                                continue;
                                        }
                                        positions.push_back(h[e].position);
                                        payloadLens.push_back(h[e].payloadLen);
                                        payloads.push_back(h[e].payload);
                                        anyPayload |= h[e].payloadLen != 0;
                                        ++counted;
                                }
                                termIDs.push_back(h[i].termID);
                                docIDs.push_back(proxy.did);
                                freqs.push_back(counted);
                                i = e;
                        }
                        h.clear();
                }

                // SegmentIndexSession::commit (indexer.cpp:311-560) up to persist_segment: one call into the engine (its sizing form first)
                committed_segment commit() {
                        committed_segment out;
                        size_t indexLen = 0, hitsLen = 0, nterms = 0;
                        const uint8_t *pl = anyPayload ? payloadLens.data() : nullptr;
                        const uint64_t *pv = anyPayload ? payloads.data() : nullptr;
                        if (codec == Codec::Lucene && anyPayload)
                                throw invalid_argument("SegmentIndexSession::commit: the Lucene-shaped encoder of this engine carries no payloads");
                        auto call = [&](uint8_t *index, size_t icap, uint8_t *hits, size_t hcap, uint32_t *ids, tri_term *tctx, size_t tcap) {
                                if (codec == Codec::Google)
                                        check(tri_commit_google(dev, termIDs.data(), docIDs.data(), freqs.data(), positions.data(), pl, pv, termIDs.size(), positions.size(), index, icap,
                                                                &indexLen, ids, tctx, tcap, &nterms, &out.stats));
                                else
                                        check(tri_commit_lucene(dev, termIDs.data(), docIDs.data(), freqs.data(), positions.data(), termIDs.size(), positions.size(), index, icap, &indexLen,
                                                                hits, hcap, &hitsLen, ids, tctx, tcap, &nterms, &out.stats));
                        };
                        call(nullptr, 0, nullptr, 0, nullptr, nullptr, 0);
                        out.index.resize(indexLen);
                        out.hits.resize(hitsLen);
                        std::vector<uint32_t> ids(nterms);
                        std::vector<tri_term> tctx(nterms);
                        call(out.index.data(), out.index.size(), out.hits.data(), out.hits.size(), ids.data(), tctx.data(), nterms);
                        out.stats.sum_term_hits += droppedHits; // (the hits insert() did not hand over: the reference's statistic counts them, indexer.cpp:447)
                        out.terms.reserve(nterms);
                        for (size_t i = 0; i < nterms; ++i)
                                out.terms.emplace_back(std::string(term(ids[i])), term_index_ctx{tctx[i].documents, tctx[i].offset, tctx[i].size}); // indexer.cpp:525-533
                        return out;
                }
        };

        // Codecs::IndexSession::merge for a whole dictionary (google_codec.cpp:186-438 per term, driven by MergeCandidatesCollection::merge, merge.cpp:40-400):
        // `participants` are the candidates' uploaded indexes, MOST RECENT FIRST, each with its masked documents installed (tri_index_set_masked: what
        // scanner_registry_for(idx) tests, merge.cpp:27-38); `termOf[t][p]` = output term t's index in participant p's term table, or no_term.  Terms that
        // keep no document come back with documents == 0 and are left out of the new dictionary by the caller (merge.cpp:241, 279).
        static constexpr uint32_t no_term = 0xffffffffu;
        inline committed_segment merge_google(tri_dev *dev, const std::vector<tri_index *> &participants, const std::vector<std::vector<uint32_t>> &termOf) {
                committed_segment out;
                const size_t np = participants.size(), nt = termOf.size();
                std::vector<uint32_t> flat(nt * np, no_term);
                for (size_t t = 0; t < nt; ++t) {
                        if (termOf[t].size() != np)
                                throw invalid_argument("merge_google: one entry per participant and output term");
                        std::copy(termOf[t].begin(), termOf[t].end(), flat.begin() + t * np);
                }
                size_t indexLen = 0;
                std::vector<tri_term> tctx(nt);
                check(tri_merge_google(dev, participants.data(), np, flat.data(), nt, nullptr, 0, &indexLen, tctx.data(), &out.stats));
                out.index.resize(indexLen);
                check(tri_merge_google(dev, participants.data(), np, flat.data(), nt, out.index.data(), out.index.size(), &indexLen, tctx.data(), &out.stats));
                for (size_t t = 0; t < nt; ++t)
                        out.terms.emplace_back(std::string(), term_index_ctx{tctx[t].documents, tctx[t].offset, tctx[t].size}); // (the caller names them: it walked the dictionaries)
                return out;
        }
        // ... and the Lucene-shaped codec's (lucene_codec.cpp:963-1396): the participants were uploaded with their hits.data; `index` and `hits` (hits.data) come back
        inline committed_segment merge_lucene(tri_dev *dev, const std::vector<tri_index *> &participants, const std::vector<std::vector<uint32_t>> &termOf) {
                committed_segment out;
                const size_t np = participants.size(), nt = termOf.size();
                std::vector<uint32_t> flat(nt * np, no_term);
                for (size_t t = 0; t < nt; ++t) {
                        if (termOf[t].size() != np)
                                throw invalid_argument("merge_lucene: one entry per participant and output term");
                        std::copy(termOf[t].begin(), termOf[t].end(), flat.begin() + t * np);
                }
                size_t indexLen = 0, hitsLen = 0;
                std::vector<tri_term> tctx(nt);
                check(tri_merge_lucene(dev, participants.data(), np, flat.data(), nt, nullptr, 0, &indexLen, nullptr, 0, &hitsLen, tctx.data(), &out.stats));
                out.index.resize(indexLen);
                out.hits.resize(hitsLen);
                check(tri_merge_lucene(dev, participants.data(), np, flat.data(), nt, out.index.data(), out.index.size(), &indexLen, out.hits.data(), out.hits.size(), &hitsLen, tctx.data(),
                                       &out.stats));
                for (size_t t = 0; t < nt; ++t)
                        out.terms.emplace_back(std::string(), term_index_ctx{tctx[t].documents, tctx[t].offset, tctx[t].size});
                return out;
        }
} // namespace trinity_amd
