// trinity_gpu.hpp — host-side C++ operator surface of the MI355X engine.
//
// Keeps the shape of Trinity's classes on the hot path so application code written against the reference reads
// the same here:  Codecs::{AccessProxy, Decoder, PostingsListIterator}  (codecs.h:211-317),
// DocsSetIterators::{Iterator, Conjuction, Disjunction, …}  (docset_iterators_base.h:45-96, docset_iterators.h),
// relevant_document_provider / IteratorScorer  (relevant_documents.h:22-81),  MatchesProxy + DocsSetSpan
// (docset_spans.h:14-90),  Similarity::{ScorerWeight, IndexSourceTermsScorer, IndexSourcesCollectionBM25Scorer}
// (similarity.h:22-255),  MatchedIndexDocumentsFilter (matches.h:139-185),  IndexSource (index_source.h:19-156),
// ExecFlags + exec_query (exec.h:11-52).
//
// What is different underneath: nothing here walks postings on the CPU.  An Iterator is a *plan node*; the first
// next()/advance() (or a span's process()) lowers the tree to a postfix program, runs it through the C-ABI
// (include/trinity_hip.h) on the GPU and then walks the materialised result.  Errors from the C-ABI surface as the
// exception type the reference would have thrown at that point.  New code — no reference source.
#pragma once
#include "../../../include/trinity_hip.h"
#include "google_encoder.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace trinity_amd {
        using isrc_docid_t = uint32_t; // common.h:36
        using docid_t = uint32_t;      // common.h:39
        using tokenpos_t = uint16_t;   // common.h:46
        using exec_term_id_t = uint16_t; // common.h: per-query term id
        static constexpr isrc_docid_t DocIDsEND{std::numeric_limits<isrc_docid_t>::max()}; // common.h:43

        enum class ExecFlags : uint32_t { DocumentsOnly = 1, AccumulatedScoreScheme = 2 }; // exec.h:11-43

        // Switch::{invalid_argument, data_error, system_error} stand-ins (Switch/switch_exceptions.h)
        struct invalid_argument : std::invalid_argument { using std::invalid_argument::invalid_argument; };
        struct data_error : std::runtime_error { using std::runtime_error::runtime_error; };
        struct system_error : std::runtime_error { using std::runtime_error::runtime_error; };
        struct aborted_search_exception final : std::exception { // matches.h:132-137
                const char *what() const noexcept override { return "Search Aborted"; }
        };

        inline void check(int rc) {
                if (rc == TRI_OK)
                        return;
                const std::string msg = tri_last_error();
                switch (rc) {
                        case TRI_ERR_INVALID:
                        case TRI_ERR_UNSUPPORTED:
                                throw invalid_argument(msg);
                        case TRI_ERR_FORMAT:
                                throw data_error(msg);
                        default:
                                throw system_error(msg);
                }
        }

        // ------------------------------------------------------------------ relevant documents / scorers
        struct relevant_document_provider { // relevant_documents.h:22-41
                virtual isrc_docid_t document() const noexcept = 0;
                inline double score();
                virtual ~relevant_document_provider() = default;
        };

        namespace DocsSetIterators {
                struct Iterator;
        }

        struct IteratorScorer : public relevant_document_provider { // relevant_documents.h:48-72
                DocsSetIterators::Iterator *const it;
                explicit IteratorScorer(DocsSetIterators::Iterator *i)
                    : it{i} {}
                inline isrc_docid_t document() const noexcept override;
                virtual double iterator_score() = 0;
        };

        inline double relevant_document_provider::score() { return static_cast<IteratorScorer *>(this)->iterator_score(); }

        class IndexSource;

        namespace Similarity { // similarity.h
                struct ScorerWeight {
                        virtual ~ScorerWeight() = default;
                };
                struct IndexSourcesCollectionTermsScorer;
                struct IndexSourceTermsScorer {
                        IndexSource *const src;
                        IndexSourcesCollectionTermsScorer *const collectionScorer;
                        IndexSourceTermsScorer(IndexSourcesCollectionTermsScorer *r, IndexSource *s)
                            : src{s}, collectionScorer{r} {}
                        virtual ~IndexSourceTermsScorer() = default;
                        virtual ScorerWeight *new_scorer_weight(const std::string *terms, uint16_t cnt) = 0;
                        virtual float score(isrc_docid_t id, uint16_t freq, const ScorerWeight *) = 0;
                        // the engine evaluates score() on the device: the scorer says which device formula it is and
                        // what per-term weight to feed it
                        virtual int device_similarity() const = 0;
                        virtual double device_weight(const ScorerWeight *) const = 0;
                };
                struct IndexSourcesCollectionTermsScorer {
                        virtual ~IndexSourcesCollectionTermsScorer() = default;
                        virtual IndexSourceTermsScorer *new_source_scorer(IndexSource *) = 0;
                };
        } // namespace Similarity

        // ------------------------------------------------------------------ iterators (plan nodes + cursors)
        namespace DocsSetIterators {
                enum class Type : uint8_t { PostingsListIterator = 0, DisjunctionSome = 1, Filter = 2, Optional = 3, Disjunction = 4, DisjunctionAllPLI, Phrase, Conjuction, ConjuctionAllPLI }; // docset_iterators_base.h:10-23

                struct Iterator : public relevant_document_provider { // docset_iterators_base.h:45-96
                        struct {
                                isrc_docid_t id{0};
                        } curDocument;
                        const Type type;
                        IndexSource *const isrc;

                        Iterator(Type t, IndexSource *s)
                            : type{t}, isrc{s} {}
                        inline isrc_docid_t current() const noexcept { return curDocument.id; }
                        isrc_docid_t document() const noexcept override { return curDocument.id; }

                        // first document > current (DocIDsEND when exhausted)
                        isrc_docid_t next() {
                                materialize();
                                return curDocument.id = cursor < docs.size() ? docs[cursor++] : DocIDsEND;
                        }
                        // first document >= target (may stay on the current one, docset_iterators_base.h:73-80)
                        isrc_docid_t advance(const isrc_docid_t target) {
                                materialize();
                                if (curDocument.id != 0 && curDocument.id >= target)
                                        return curDocument.id;
                                size_t lo = cursor, hi = docs.size();
                                while (lo < hi) {
                                        const size_t mid = (lo + hi) / 2;
                                        if (docs[mid] < target)
                                                lo = mid + 1;
                                        else
                                                hi = mid;
                                }
                                cursor = lo;
                                return next();
                        }
                        virtual uint64_t cost() const = 0; // docset_iterators.cpp:10-64
                        // emit this subtree as postfix tokens; `w` (parallel to prog) receives each TERM token's ScorerWeight
                        virtual void lower(std::vector<uint32_t> &prog, std::vector<double> &w, Similarity::IndexSourceTermsScorer *scorer) const = 0;

                      protected:
                        std::vector<isrc_docid_t> docs;
                        size_t cursor{0};
                        bool materialized{false};
                        virtual void materialize(); // default: run this subtree as one DocumentsOnly query on the GPU
                        friend class ::trinity_amd::IndexSource;
                };
        } // namespace DocsSetIterators

        inline isrc_docid_t IteratorScorer::document() const noexcept { return it->current(); }

        // ------------------------------------------------------------------ codecs
        namespace Codecs { // codecs.h:211-317
                struct Decoder;
                struct PostingsListIterator : public DocsSetIterators::Iterator {
                        Decoder *const dec;
                        tokenpos_t freq{0};
                        inline PostingsListIterator(Decoder *d);
                        inline isrc_docid_t next() {
                                const auto id = Iterator::next();
                                freq = id == DocIDsEND ? 0 : tokenpos_t(freqs[cursor - 1]);
                                return id;
                        }
                        inline isrc_docid_t advance(isrc_docid_t target) {
                                const auto id = Iterator::advance(target);
                                freq = id == DocIDsEND ? 0 : tokenpos_t(freqs[cursor - 1]);
                                return id;
                        }
                        Decoder *decoder() noexcept { return dec; }
                        uint64_t cost() const override;
                        void lower(std::vector<uint32_t> &prog, std::vector<double> &w, Similarity::IndexSourceTermsScorer *scorer) const override;

                      protected:
                        std::vector<uint32_t> freqs;
                        void materialize() override; // tri_decode_terms: the whole list, docIDs + freqs
                };

                struct AccessProxy;
                struct Decoder {
                        term_index_ctx indexTermCtx;
                        uint32_t termId{0}; // row of the uploaded term table
                        std::string term;
                        IndexSource *isrc{nullptr};
                        virtual ~Decoder() = default;
                        virtual void init(const term_index_ctx &, AccessProxy *) = 0;
                        virtual PostingsListIterator *new_iterator() { return new PostingsListIterator(this); }
                };

                struct AccessProxy {
                        const uint8_t *const indexPtr;
                        explicit AccessProxy(const uint8_t *p)
                            : indexPtr{p} {}
                        virtual ~AccessProxy() = default;
                        virtual Decoder *new_decoder(const term_index_ctx &) = 0;
                        virtual const char *codec_identifier() = 0;
                };

                namespace Google {
                        struct Decoder final : public Codecs::Decoder {
                                void init(const term_index_ctx &t, Codecs::AccessProxy *) override { indexTermCtx = t; }
                        };
                        struct AccessProxy final : public Codecs::AccessProxy {
                                using Codecs::AccessProxy::AccessProxy;
                                const char *codec_identifier() override { return "GOOGLE"; }
                                Codecs::Decoder *new_decoder(const term_index_ctx &t) override {
                                        auto d = new Decoder();
                                        d->init(t, this);
                                        return d;
                                }
                        };
                } // namespace Google

                // lucene_codec.h:246-283 — the codec SegmentIndexSource picks for a segment whose `id` file names "LUCENE"
                // (segment_index_source.cpp:172-179).  Besides the index it maps the segment's hits.data (lucene_codec.h:206: positions and payloads
                // live in their own file).  The engine reads this repo's PFOR128 ints() payload (include/pfor128.md), see INTEGRATION.md §1
                namespace Lucene {
                        struct Decoder final : public Codecs::Decoder {
                                void init(const term_index_ctx &t, Codecs::AccessProxy *) override { indexTermCtx = t; }
                        };
                        struct AccessProxy final : public Codecs::AccessProxy {
                                const uint8_t *const hitsDataPtr;
                                AccessProxy(const uint8_t *index, const uint8_t *hits)
                                    : Codecs::AccessProxy(index), hitsDataPtr{hits} {}
                                const char *codec_identifier() override { return "LUCENE"; }
                                Codecs::Decoder *new_decoder(const term_index_ctx &t) override {
                                        auto d = new Decoder();
                                        d->init(t, this);
                                        return d;
                                }
                        };
                } // namespace Lucene
        }         // namespace Codecs

        inline Codecs::PostingsListIterator::PostingsListIterator(Decoder *d)
            : Iterator{DocsSetIterators::Type::PostingsListIterator, d->isrc}, dec{d} {}

        // ------------------------------------------------------------------ composite iterators
        namespace DocsSetIterators {
                struct Conjuction final : public Iterator { // docset_iterators.h:333-362
                        std::vector<Iterator *> its;
                        Conjuction(Iterator **iterators, uint16_t cnt)
                            : Iterator{Type::Conjuction, iterators[0]->isrc}, its(iterators, iterators + cnt) {}
                        uint64_t cost() const override { // the cheapest member leads (exec.cpp:154-170, 44-55)
                                uint64_t c = UINT64_MAX;
                                for (auto it : its)
                                        c = std::min(c, it->cost());
                                return c;
                        }
                        void lower(std::vector<uint32_t> &prog, std::vector<double> &w, Similarity::IndexSourceTermsScorer *scorer) const override {
                                for (auto it : its)
                                        it->lower(prog, w, scorer);
                                prog.push_back(TRI_TOK(TRI_OP_AND, its.size()));
                                w.push_back(0.0);
                        }
                };
                struct Disjunction final : public Iterator { // docset_iterators.h:261-303
                        std::vector<Iterator *> its;
                        Disjunction(Iterator **iterators, uint16_t cnt)
                            : Iterator{Type::Disjunction, iterators[0]->isrc}, its(iterators, iterators + cnt) {}
                        uint64_t cost() const override {
                                uint64_t c = 0;
                                for (auto it : its)
                                        c += it->cost();
                                return c;
                        }
                        void lower(std::vector<uint32_t> &prog, std::vector<double> &w, Similarity::IndexSourceTermsScorer *scorer) const override {
                                for (auto it : its)
                                        it->lower(prog, w, scorer);
                                prog.push_back(TRI_TOK(TRI_OP_OR, its.size()));
                                w.push_back(0.0);
                        }
                };
                struct DisjunctionSome final : public Iterator { // docset_iterators.h:61-137 (matchsome, exec.cpp:276-283): documents at least minMatch members hold
                        std::vector<Iterator *> its;
                        const uint16_t matchThreshold;
                        DisjunctionSome(Iterator **iterators, uint16_t cnt, uint16_t minMatch)
                            : Iterator{Type::DisjunctionSome, iterators[0]->isrc}, its(iterators, iterators + cnt), matchThreshold{minMatch} {}
                        uint64_t cost() const override { // docset_iterators.cpp:733-742: the (cnt - minMatch + 1) cheapest members
                                std::vector<uint64_t> cs;
                                for (auto it : its)
                                        cs.push_back(it->cost());
                                std::sort(cs.begin(), cs.end());
                                uint64_t c = 0;
                                for (size_t i = 0; i + matchThreshold <= cs.size(); ++i)
                                        c += cs[i];
                                return c;
                        }
                        void lower(std::vector<uint32_t> &prog, std::vector<double> &w, Similarity::IndexSourceTermsScorer *scorer) const override {
                                for (auto it : its)
                                        it->lower(prog, w, scorer);
                                prog.push_back(TRI_TOK(TRI_OP_SOME, (uint32_t(matchThreshold) << 16) | uint32_t(its.size())));
                                w.push_back(0.0);
                        }
                };
                struct Filter final : public Iterator { // docset_iterators.h:147-172: documents of req that filter does not hold
                        Iterator *const req, *const filter;
                        Filter(Iterator *const r, Iterator *const f)
                            : Iterator{Type::Filter, r->isrc}, req{r}, filter{f} {}
                        uint64_t cost() const override { return req->cost(); } // docset_iterators.cpp:10-64
                        void lower(std::vector<uint32_t> &prog, std::vector<double> &w, Similarity::IndexSourceTermsScorer *scorer) const override {
                                req->lower(prog, w, scorer);
                                filter->lower(prog, w, nullptr); // the excluded side is never scored (docset_iterators_scorers.cpp:59-73)
                                prog.push_back(TRI_TOK(TRI_OP_NOT, 2));
                                w.push_back(0.0);
                        }
                };
                struct Optional final : public Iterator { // docset_iterators.h:174-206: main's documents; opt adds score / matched terms where it matches
                        Iterator *const main, *const opt;
                        Optional(Iterator *const m, Iterator *const o)
                            : Iterator{Type::Optional, m->isrc}, main{m}, opt{o} {}
                        uint64_t cost() const override { return main->cost(); }
                        void lower(std::vector<uint32_t> &prog, std::vector<double> &w, Similarity::IndexSourceTermsScorer *scorer) const override {
                                main->lower(prog, w, scorer);
                                opt->lower(prog, w, scorer);
                                prog.push_back(TRI_TOK(TRI_OP_OPT, 2));
                                w.push_back(0.0);
                        }
                };
                struct Phrase final : public Iterator { // docset_iterators.h:364-402
                        std::vector<Codecs::PostingsListIterator *> its;
                        Phrase(Codecs::PostingsListIterator **iterators, uint16_t cnt)
                            : Iterator{Type::Phrase, iterators[0]->isrc}, its(iterators, iterators + cnt) {}
                        uint64_t cost() const override { return its[0]->cost() + UINT32_MAX + uint64_t(UINT16_MAX) * its.size(); } // exec.cpp:28-34
                        void lower(std::vector<uint32_t> &prog, std::vector<double> &w, Similarity::IndexSourceTermsScorer *scorer) const override;
                };

                // docset_iterators.h:456-497: the provider a span hands to MatchesProxy::process for a materialised match.
                // score() on a provider static_casts to IteratorScorer (relevant_documents.h:76-81), so it is one.
                struct relevant_document final : public IteratorScorer {
                        struct DummyIterator final : public Iterator {
                                DummyIterator()
                                    : Iterator{Type::PostingsListIterator, nullptr} {}
                                uint64_t cost() const override { return 0; }
                                void lower(std::vector<uint32_t> &, std::vector<double> &, Similarity::IndexSourceTermsScorer *) const override { throw invalid_argument("not a plan node"); }
                        } dummy;
                        double score_{0};
                        relevant_document()
                            : IteratorScorer{&dummy} {}
                        void set_document(const isrc_docid_t id) noexcept { dummy.curDocument.id = id; }
                        double iterator_score() override { return score_; }
                };
        } // namespace DocsSetIterators
        using DocsSetIterators::relevant_document;

        // ------------------------------------------------------------------ spans, proxies, filters
        class MatchesProxy { // docset_spans.h:14-21
              public:
                virtual void process(relevant_document_provider *) {}
                virtual ~MatchesProxy() = default;
        };

        // ---- what the default execution mode hands to the application (matches.h:34-130), positions only (no payloads)
        struct term_hit {
                tokenpos_t pos;
        };
        struct term_hits {
                tokenpos_t freq{0};
                term_hit *all{nullptr};
        };
        struct query_term_ctx {
                struct {
                        exec_term_id_t id;
                        std::string token;
                } term;
        };
        struct matched_query_term {
                const query_term_ctx *queryCtx;
                term_hits *hits;
        };
        struct matched_document {
                docid_t id{0};
                uint16_t matchedTermsCnt{0};
                matched_query_term *matchedTerms{nullptr};
        };

        struct MatchedIndexDocumentsFilter { // matches.h:139-185
                virtual void consider(const matched_document &) {} // default execution mode
                virtual void consider(const docid_t) {}
                virtual void consider(const docid_t *ids, const size_t cnt) {
                        for (size_t i = 0; i != cnt; ++i)
                                consider(ids[i]);
                }
                virtual void consider(const docid_t, const double) {}
                virtual ~MatchedIndexDocumentsFilter() = default;
        };

        struct IndexDocumentsFilter { // matches.h:198-201: return true to disregard the document
                virtual bool filter(const docid_t) = 0;
                virtual ~IndexDocumentsFilter() = default;
        };

        // ------------------------------------------------------------------ index source (GPU resident)
        struct field_statistics { // index_source.h:44-53
                uint64_t sumTermHits{0};
                uint32_t totalTerms{0};
                uint64_t sumTermsDocs{0};
                uint32_t docsCnt{0};
        };

        class IndexSource { // index_source.h:19-156, backed by an uploaded segment
                tri_dev *dev{nullptr};
                tri_index *ix{nullptr};
                std::vector<std::string> names;                 // row -> term
                std::unordered_map<std::string, uint32_t> dict; // term -> row; the reference's SegmentTerms (terms.h) is host-only and out of scope
                std::vector<term_index_ctx> table;
                std::unique_ptr<Codecs::AccessProxy> access; // Google or Lucene, by the codec name of the segment (segment_index_source.cpp:172-179)
                std::vector<std::unique_ptr<Codecs::Decoder>> decoders;
                std::vector<std::unique_ptr<DocsSetIterators::Iterator>> registry; // queryexec_ctx::reg_pli / reg_docset_it
                field_statistics fs;

                std::vector<docid_t> masked; // the set installed with set_masked_documents (kept: exec_query's per-call registry restores it)

              public:
                // `index`/`len`: the segment's raw `index` bytes; terms[i] names table[i].  `codec`: the name in the segment's `id` file — "GOOGLE", or
                // "LUCENE" with the segment's hits.data in `hits` (segment_index_source.cpp:147-179 reads the name and picks the AccessProxy)
                IndexSource(int device, const uint8_t *index, size_t len, const std::vector<std::string> &terms, const std::vector<term_index_ctx> &tctx,
                            const field_statistics &stats, const std::string &codec = "GOOGLE", const uint8_t *hits = nullptr, size_t hits_len = 0)
                    : table(tctx), fs(stats) {
                        if (terms.size() != tctx.size())
                                throw invalid_argument("terms/tctx size mismatch");
                        if (codec != "GOOGLE" && codec != "LUCENE")
                                throw invalid_argument("unknown codec"); // segment_index_source.cpp:178: "Unknown codec"
                        check(tri_dev_open(device, &dev));
                        static_assert(sizeof(term_index_ctx) == sizeof(tri_term), "term_index_ctx layout");
                        const bool lucene = codec == "LUCENE";
                        const int rc = tri_index_upload(dev, index, len, lucene ? hits : nullptr, lucene ? hits_len : 0, lucene ? TRI_CODEC_LUCENE : TRI_CODEC_GOOGLE,
                                                        reinterpret_cast<const tri_term *>(tctx.data()), tctx.size(), stats.docsCnt, &ix);
                        if (rc != TRI_OK) {
                                const std::string why = tri_last_error();
                                tri_dev_close(dev);
                                dev = nullptr;
                                throw data_error(why);
                        }
                        names = terms;
                        for (uint32_t i = 0; i < terms.size(); ++i)
                                dict.emplace(terms[i], i);
                        if (lucene)
                                access.reset(new Codecs::Lucene::AccessProxy(index, hits));
                        else
                                access.reset(new Codecs::Google::AccessProxy(index));
                }
                ~IndexSource() {
                        registry.clear();
                        tri_index_destroy(ix);
                        tri_dev_close(dev);
                }
                IndexSource(const IndexSource &) = delete;

                tri_index *handle() const noexcept { return ix; }
                const std::string &term_name(const uint32_t row) const { return names.at(row); }
                // The documents of this source that newer sources of the collection have updated or deleted — what
                // IndexSourcesCollection::commit() derives per source (index_source.cpp:3-30) and exec_query tests through
                // masked_documents_registry::test before every consider() (exec.cpp:914-975).  Uploaded once per refresh of the
                // collection; the matching kernels drop these documents themselves.
                void set_masked_documents(const std::vector<docid_t> &ids) {
                        check(tri_index_set_masked(ix, ids.data(), ids.size()));
                        masked = ids;
                }
                const std::vector<docid_t> &masked_documents() const noexcept { return masked; }
                field_statistics default_field_stats() const { return fs; }

                // index_source.h:103: unknown term => no documents
                term_index_ctx resolve_term_ctx(const std::string &term) const {
                        const auto it = dict.find(term);
                        return it == dict.end() ? term_index_ctx{} : table[it->second];
                }
                uint32_t term_id(const std::string &term) const {
                        const auto it = dict.find(term);
                        return it == dict.end() ? 0x0fffffffu : it->second; // an id past the table == "no documents" for the engine
                }
                // index_source.h:119
                Codecs::Decoder *new_postings_decoder(const std::string &term, const term_index_ctx ctx) {
                        auto d = access->new_decoder(ctx);
                        d->termId = term_id(term);
                        d->term = term;
                        d->isrc = this;
                        decoders.emplace_back(d);
                        return d;
                }

                // ---- iterator factories == queryexec_ctx::build_iterator's leaves and combinators (exec.cpp:253-449)
                Codecs::PostingsListIterator *term(const std::string &t) {
                        auto dec = new_postings_decoder(t, resolve_term_ctx(t));
                        auto it = dec->new_iterator();
                        registry.emplace_back(it);
                        return it;
                }
                template <class T, class... A>
                T *reg(A &&... a) {
                        auto p = new T(std::forward<A>(a)...);
                        registry.emplace_back(p);
                        return p;
                }
                DocsSetIterators::Iterator *conjunction(std::vector<DocsSetIterators::Iterator *> its) { return reg<DocsSetIterators::Conjuction>(its.data(), uint16_t(its.size())); }
                DocsSetIterators::Iterator *disjunction(std::vector<DocsSetIterators::Iterator *> its) { return reg<DocsSetIterators::Disjunction>(its.data(), uint16_t(its.size())); }
                DocsSetIterators::Iterator *some(std::vector<DocsSetIterators::Iterator *> its, uint16_t minMatch) { return reg<DocsSetIterators::DisjunctionSome>(its.data(), uint16_t(its.size()), minMatch); } // exec.cpp:276-283
                DocsSetIterators::Iterator *optional(DocsSetIterators::Iterator *main, DocsSetIterators::Iterator *opt) { return reg<DocsSetIterators::Optional>(main, opt); } // exec.cpp:366-377
                DocsSetIterators::Iterator *filter(DocsSetIterators::Iterator *req, DocsSetIterators::Iterator *excl) { return reg<DocsSetIterators::Filter>(req, excl); } // exec.cpp:424-427
                DocsSetIterators::Iterator *phrase(const std::vector<std::string> &terms) {
                        std::vector<Codecs::PostingsListIterator *> its;
                        for (const auto &t : terms)
                                its.push_back(term(t));
                        return reg<DocsSetIterators::Phrase>(its.data(), uint16_t(its.size()));
                }
        };

        // ------------------------------------------------------------------ BM25 (similarity.h:165-255)
        namespace Similarity {
                struct IndexSourcesCollectionBM25Scorer : public IndexSourcesCollectionTermsScorer {
                        static constexpr float k1{1.2f};
                        struct Scorer final : public IndexSourceTermsScorer {
                                struct Weight final : public ScorerWeight {
                                        const double idf;
                                        explicit Weight(double i)
                                            : idf{i} {}
                                };
                                using IndexSourceTermsScorer::IndexSourceTermsScorer;
                                // similarity.h:179-181: evaluated in float precision, kept as double
                                static double idf(const uint32_t docFreq, const uint64_t docsCnt) { return std::log(1 + (docsCnt - docFreq + 0.5f) / (docFreq + 0.5f)); }
                                ScorerWeight *new_scorer_weight(const std::string *terms, uint16_t cnt) override {
                                        double w = 0;
                                        for (uint16_t i = 0; i != cnt; ++i)
                                                w += idf(src->resolve_term_ctx(terms[i]).documents, src->default_field_stats().docsCnt);
                                        return new Weight(w);
                                }
                                // similarity.h:228-235 (host copy of what the device evaluates)
                                float score(isrc_docid_t, uint16_t freq, const ScorerWeight *w) override {
                                        return float(static_cast<const Weight *>(w)->idf * float(freq) / double(freq + k1));
                                }
                                int device_similarity() const override { return TRI_SIM_BM25; }
                                double device_weight(const ScorerWeight *w) const override { return static_cast<const Weight *>(w)->idf; }
                        };
                        IndexSourceTermsScorer *new_source_scorer(IndexSource *s) override { return new Scorer(this, s); }
                };

                // similarity.h:75-163: idf = log((docsCnt + 1) / double(df + 1)) + 1, tf = sqrt(float freq), score = tf * weight
                struct IndexSourcesCollectionTFIDFScorer : public IndexSourcesCollectionTermsScorer {
                        struct Scorer final : public IndexSourceTermsScorer {
                                struct Weight final : public ScorerWeight {
                                        const double v;
                                        explicit Weight(double value)
                                            : v{value} {}
                                };
                                using IndexSourceTermsScorer::IndexSourceTermsScorer;
                                static double idf(const uint32_t docFreq, const uint64_t docsCnt) { return std::log((docsCnt + 1) / double(docFreq + 1)) + 1.0; }
                                ScorerWeight *new_scorer_weight(const std::string *terms, uint16_t cnt) override {
                                        double w = 0;
                                        for (uint16_t i = 0; i != cnt; ++i)
                                                w += idf(src->resolve_term_ctx(terms[i]).documents, src->default_field_stats().docsCnt);
                                        return new Weight(w);
                                }
                                float score(isrc_docid_t, uint16_t freq, const ScorerWeight *w) override { return std::sqrt(float(freq)) * static_cast<const Weight *>(w)->v; }
                                int device_similarity() const override { return TRI_SIM_TFIDF; }
                                double device_weight(const ScorerWeight *w) const override { return static_cast<const Weight *>(w)->v; }
                        };
                        IndexSourceTermsScorer *new_source_scorer(IndexSource *s) override { return new Scorer(this, s); }
                };

                // similarity.h:56-72: score = freq, no weight
                struct IndexSourcesCollectionTrivialScorer : public IndexSourcesCollectionTermsScorer {
                        struct Scorer final : public IndexSourceTermsScorer {
                                using IndexSourceTermsScorer::IndexSourceTermsScorer;
                                ScorerWeight *new_scorer_weight(const std::string *, uint16_t) override { return nullptr; }
                                float score(isrc_docid_t, uint16_t freq, const ScorerWeight *) override { return freq; }
                                int device_similarity() const override { return TRI_SIM_TRIVIAL; }
                                double device_weight(const ScorerWeight *) const override { return 0.0; }
                        };
                        IndexSourceTermsScorer *new_source_scorer(IndexSource *s) override { return new Scorer(this, s); }
                };
        } // namespace Similarity

        // ------------------------------------------------------------------ execution
        struct BatchDeleter {
                void operator()(tri_batch *b) const { tri_batch_destroy(b); }
        };
        using BatchPtr = std::unique_ptr<tri_batch, BatchDeleter>;

        inline void validate_flags(const uint32_t f) { // exec.h:45-48
                const auto mask = f & (unsigned(ExecFlags::DocumentsOnly) | unsigned(ExecFlags::AccumulatedScoreScheme));
                if (mask && (mask & (mask - 1)))
                        throw invalid_argument("DocumentsOnly and AccumulatedScoreScheme are mutually exclusive modes");
        }

        // Lower iterator trees (the mirror of build_iterator's output), attach one ScorerWeight per TERM token
        // (docset_iterators_scorers.cpp:16-22), create + run one batch.
        inline BatchPtr run_batch(IndexSource *src, const std::vector<DocsSetIterators::Iterator *> &roots, uint32_t flags, uint32_t topk,
                                  Similarity::IndexSourceTermsScorer *scorer) {
                validate_flags(flags);
                const bool scored = flags & unsigned(ExecFlags::AccumulatedScoreScheme);
                if (!(flags & (unsigned(ExecFlags::DocumentsOnly) | unsigned(ExecFlags::AccumulatedScoreScheme))))
                        flags = TRI_FLAG_MATCHED_TERMS; // no ExecFlags: exec_query's default mode (exec.cpp:1350-1501)
                if (scored && !scorer)
                        throw invalid_argument("IndexSourceTermsScorer not set"); // exec.h:105-108
                std::vector<uint32_t> prog;
                std::vector<double> weights;
                std::vector<tri_query> qs;
                for (auto r : roots) {
                        const uint32_t off = uint32_t(prog.size());
                        r->lower(prog, weights, scored ? scorer : nullptr);
                        qs.push_back({off, uint32_t(prog.size()) - off});
                }
                tri_batch *b = nullptr;
                check(tri_batch_create(src->handle(), prog.data(), prog.size(), qs.data(), qs.size(), scored ? weights.data() : nullptr, flags, topk,
                                       scored ? scorer->device_similarity() : TRI_SIM_BM25, &b));
                BatchPtr bp(b);
                // (a shape the planner does not lower is left out of the batch with a status, not refused as a whole: these entry points
                //  run one caller query at a time — exec_queries: one caller batch —, so such a query surfaces as the exception it always did)
                std::vector<int32_t> st(qs.size(), TRI_OK);
                check(tri_batch_query_status(b, st.data()));
                for (const int32_t x : st)
                        if (x != TRI_OK)
                                check(x);
                check(tri_batch_run(b));
                check(tri_batch_sync(b));
                return bp;
        }

        class DocsSetSpan { // docset_spans.h:36-90
              public:
                virtual isrc_docid_t process(MatchesProxy *, const isrc_docid_t min, const isrc_docid_t max) = 0;
                virtual uint64_t cost() = 0;
                virtual ~DocsSetSpan() = default;
        };

        // The batch-granular seam: the FIRST process() runs the whole iterator tree on the GPU — one tri_batch_create / run / read-back —, and the span
        // keeps the ascending match list (+ scores); every process(mp, min, max) then replays the matches of [min, max) from it, found by a
        // binary search, through the caller's MatchesProxy in ascending docID order, as GenericDocsSetSpan::process (docset_spans.cpp:269-290)
        // and the window-union spans (98-173, 681-790) do, and returns the first match >= max.  A composite caller that walks the docID space
        // window by window (docset_spans.h:292-296: min == max with no proxy means "just advance") therefore costs ONE device batch for the
        // whole walk, not one per window (round 4 compiled and ran the query again on every call); batches_run() says how many there were.
        class GpuDocsSetSpan final : public DocsSetSpan {
                DocsSetIterators::Iterator *const root;
                const uint32_t flags;
                Similarity::IndexSourceTermsScorer *const scorer;
                std::vector<uint32_t> ids; // the query's matches, ascending (filled by the first process())
                std::vector<double> sc;    // ... and their scores (AccumulatedScoreScheme)
                bool ran{false};
                unsigned batches{0};

                void run_once() {
                        if (ran)
                                return;
                        const bool scored = flags & unsigned(ExecFlags::AccumulatedScoreScheme);
                        auto b = run_batch(root->isrc, {root}, flags, 0, scorer);
                        ++batches;
                        size_t n = 0;
                        check(tri_batch_docset(b.get(), 0, nullptr, 0, &n));
                        ids.resize(n);
                        sc.resize(scored ? n : 0);
                        if (n) {
                                check(tri_batch_docset(b.get(), 0, ids.data(), n, &n));
                                if (scored)
                                        check(tri_batch_scores(b.get(), 0, sc.data(), n, &n));
                        }
                        ran = true;
                }

              public:
                GpuDocsSetSpan(DocsSetIterators::Iterator *r, uint32_t f, Similarity::IndexSourceTermsScorer *s)
                    : root{r}, flags{f}, scorer{s} {}
                uint64_t cost() override { return root->cost(); }
                unsigned batches_run() const { return batches; } // device batches this span has compiled and run (1 after any number of windows)
                isrc_docid_t process(MatchesProxy *mp, const isrc_docid_t min, const isrc_docid_t max) override {
                        run_once();
                        const bool scored = !sc.empty();
                        size_t i = size_t(std::lower_bound(ids.begin(), ids.end(), uint32_t(min)) - ids.begin());
                        relevant_document rel;
                        for (; i < ids.size() && ids[i] < max; ++i)
                                if (mp) {
                                        rel.set_document(ids[i]);
                                        rel.score_ = scored ? sc[i] : 0.0;
                                        mp->process(&rel);
                                }
                        return i < ids.size() ? isrc_docid_t(ids[i]) : DocIDsEND;
                }
        };

        // exec.cpp:509-1517 for the two lowered modes: build the span over the iterator tree, process(1, DocIDsEND),
        // deliver through the no-mask handlers (exec.cpp:1213-1229 docs-only, 1322-1341 accumulated score), honouring an
        // IndexDocumentsFilter (matches.h:198-201) and cooperative cancellation (exec.cpp:1505-1510).
        // The default mode: every match is delivered as a matched_document — the query terms that matched it and their hits
        // (prepare_match, queryexec_ctx.cpp:522-648) — rebuilt here from the engine's packed arrays.
        inline void exec_query_default_mode(DocsSetIterators::Iterator *root, IndexSource *src, MatchedIndexDocumentsFilter *mf, IndexDocumentsFilter *df) {
                auto b = run_batch(src, {root}, 0, 0, nullptr);
                size_t n = 0, npos = 0;
                check(tri_batch_docset(b.get(), 0, nullptr, 0, &n));
                std::vector<docid_t> ids(n);
                if (n)
                        check(tri_batch_docset(b.get(), 0, ids.data(), n, &n));
                uint32_t terms[16], nt = 0;
                check(tri_batch_query_terms(b.get(), 0, terms, &nt));
                check(tri_batch_matched_terms(b.get(), 0, nullptr, nullptr, nullptr, 0, &npos));
                std::vector<uint32_t> present(n);
                std::vector<uint16_t> freq(n * std::max<uint32_t>(nt, 1)), pos(npos);
                check(tri_batch_matched_terms(b.get(), 0, present.data(), freq.data(), pos.data(), pos.size(), &npos));
                std::vector<query_term_ctx> qctx(nt);
                for (uint32_t k = 0; k < nt; ++k) {
                        qctx[k].term.id = exec_term_id_t(k + 1);
                        qctx[k].term.token = src->term_name(terms[k]);
                }
                std::vector<term_hits> th(nt);
                std::vector<matched_query_term> mts(nt);
                std::vector<std::vector<term_hit>> store(nt);
                size_t at = 0;
                try {
                        for (size_t i = 0; i < n; ++i) {
                                matched_document md;
                                md.id = ids[i];
                                md.matchedTerms = mts.data();
                                for (uint32_t k = 0; k < nt; ++k) {
                                        const uint32_t f = freq[i * nt + k];
                                        if ((present[i] >> k) & 1u) {
                                                store[k].resize(f);
                                                for (uint32_t h = 0; h < f; ++h)
                                                        store[k][h].pos = pos[at + h];
                                                th[k].freq = tokenpos_t(f);
                                                th[k].all = store[k].data();
                                                mts[md.matchedTermsCnt++] = {&qctx[k], &th[k]};
                                        }
                                        at += f;
                                }
                                if (!(df && df->filter(md.id)))
                                        mf->consider(md);
                        }
                } catch (const aborted_search_exception &) {
                }
        }

        // docidupdates.h:15-119.  The documents of a source that newer sources of the collection have updated or deleted.  The reference packs
        // each source's list into banks with a skiplist (updated_documents, pack_updates / unpack_updates) and hands exec_query a
        // masked_documents_registry over the lists of all the newer sources — a bloom filter in front of one scanner per list, tested
        // document by document right before consider() (exec.cpp:914-975).  Its observable behaviour is set membership, and that is what
        // is mirrored: the registry is the union of its lists, it goes to the device as a docID bitmap (tri_index_set_masked) and the
        // matching kernels test it where exec_query does.
        struct updated_documents final {
                std::vector<docid_t> ids; // ascending (what pack_updates sorts them into)
        };
        struct masked_documents_registry final {
                std::vector<docid_t> ids; // ascending, distinct
                // docidupdates.h:121-142 masked_documents_registry::make(const updated_documents *, n)
                static std::unique_ptr<masked_documents_registry> make(const updated_documents *ud, const std::size_t n) {
                        auto r = std::make_unique<masked_documents_registry>();
                        for (std::size_t i = 0; i < n; ++i)
                                r->ids.insert(r->ids.end(), ud[i].ids.begin(), ud[i].ids.end());
                        std::sort(r->ids.begin(), r->ids.end());
                        r->ids.erase(std::unique(r->ids.begin(), r->ids.end()), r->ids.end());
                        return r;
                }
                bool test(const docid_t id) const { return std::binary_search(ids.begin(), ids.end(), id); }
                bool empty() const noexcept { return ids.empty(); }
        };

        inline void exec_query(DocsSetIterators::Iterator *root, IndexSource *src, MatchedIndexDocumentsFilter *matchesFilter, IndexDocumentsFilter *f = nullptr,
                               const uint32_t flags = 0, Similarity::IndexSourceTermsScorer *scorer = nullptr);

        // exec.h:50: exec_query(query, IndexSource *, masked_documents_registry *, MatchedIndexDocumentsFilter *, IndexDocumentsFilter *, flags,
        // scorer) — the reference's own argument order (the query is the lowered iterator tree here).  The registry becomes the source's
        // masked set for this call (nullptr / empty: none), then the query runs as above.
        inline void exec_query(DocsSetIterators::Iterator *root, IndexSource *src, masked_documents_registry *const maskedDocumentsRegistry,
                               MatchedIndexDocumentsFilter *matchesFilter, IndexDocumentsFilter *const f = nullptr, const uint32_t flags = 0,
                               Similarity::IndexSourceTermsScorer *scorer = nullptr) {
                // (the reference's registry is strictly per call, exec.h:50: whatever set the source carried before is back when the call returns,
                //  also when the application's filter throws)
                struct Restore {
                        IndexSource *src;
                        std::vector<docid_t> before;
                        ~Restore() {
                                try {
                                        src->set_masked_documents(before);
                                } catch (...) {
                                }
                        }
                } restore{src, src->masked_documents()};
                src->set_masked_documents(maskedDocumentsRegistry ? maskedDocumentsRegistry->ids : std::vector<docid_t>{});
                exec_query(root, src, matchesFilter, f, flags, scorer);
        }

        inline void exec_query(DocsSetIterators::Iterator *root, IndexSource *src, MatchedIndexDocumentsFilter *matchesFilter, IndexDocumentsFilter *f,
                               const uint32_t flags, Similarity::IndexSourceTermsScorer *scorer) {
                validate_flags(flags);
                if (!(flags & (unsigned(ExecFlags::DocumentsOnly) | unsigned(ExecFlags::AccumulatedScoreScheme))))
                        return exec_query_default_mode(root, src, matchesFilter, f);
                struct Handler final : public MatchesProxy {
                        MatchedIndexDocumentsFilter *mf;
                        IndexDocumentsFilter *df;
                        bool scored;
                        void process(relevant_document_provider *rdp) override {
                                const auto id = rdp->document();
                                if (df && df->filter(id))
                                        return;
                                if (scored)
                                        mf->consider(id, rdp->score());
                                else
                                        mf->consider(id);
                        }
                } handler;
                handler.mf = matchesFilter;
                handler.df = f;
                handler.scored = flags & unsigned(ExecFlags::AccumulatedScoreScheme);
                GpuDocsSetSpan span(root, flags, scorer);
                try {
                        span.process(&handler, 1, DocIDsEND);
                } catch (const aborted_search_exception &) {
                        // search was aborted by the application's filter
                }
        }

        // The batched sibling of exec_query_par (exec.h:87-177): all queries in ONE engine batch; DocumentsOnly results
        // arrive through consider(ids, cnt) (matches.h:161-165), scored ones through consider(id, score).
        inline void exec_queries(const std::vector<DocsSetIterators::Iterator *> &roots, IndexSource *src, const std::vector<MatchedIndexDocumentsFilter *> &filters,
                                 const uint32_t flags, Similarity::IndexSourceTermsScorer *scorer = nullptr) {
                if (roots.size() != filters.size())
                        throw invalid_argument("one filter per query");
                const bool scored = flags & unsigned(ExecFlags::AccumulatedScoreScheme);
                auto b = run_batch(src, roots, flags, 0, scorer);
                std::vector<uint32_t> ids;
                std::vector<double> sc;
                if (!scored) {
                        // every set in ONE delivery, each in the form the engine holds it (tri_batch_docsets_mixed): a dense set arrives as the words of a bitmap over
                        // its docID range — a bit per document across PCIe instead of four bytes per match — and is expanded here into consider()'s ids
                        std::vector<uint64_t> offs(roots.size() + 1);
                        std::vector<uint32_t> forms(roots.size() + 1), flat;
                        check(tri_batch_docsets_mixed(b.get(), nullptr, 0, offs.data(), forms.data()));
                        flat.resize(offs.back() + 1);
                        check(tri_batch_docsets_mixed(b.get(), flat.data(), flat.size(), offs.data(), forms.data()));
                        for (size_t q = 0; q < roots.size(); ++q) {
                                const uint32_t *part = flat.data() + offs[q];
                                size_t n = offs[q + 1] - offs[q];
                                if (forms[q]) { // bit j of word i: document 32 i + j
                                        ids.clear();
                                        for (size_t i = 0; i < n; ++i)
                                                for (uint32_t m = part[i]; m; m &= m - 1)
                                                        ids.push_back((uint32_t)(32 * i) + (uint32_t)__builtin_ctz(m));
                                        part = ids.data();
                                        n = ids.size();
                                }
                                try {
                                        filters[q]->consider(part, n);
                                } catch (const aborted_search_exception &) {
                                }
                        }
                        return;
                }
                for (size_t q = 0; q < roots.size(); ++q) {
                        size_t n = 0;
                        check(tri_batch_docset(b.get(), q, nullptr, 0, &n));
                        ids.resize(n);
                        if (n)
                                check(tri_batch_docset(b.get(), q, ids.data(), n, &n));
                        try {
                                if (!scored)
                                        filters[q]->consider(ids.data(), n);
                                else {
                                        sc.resize(n);
                                        if (n)
                                                check(tri_batch_scores(b.get(), q, sc.data(), n, &n));
                                        for (size_t i = 0; i < n; ++i)
                                                filters[q]->consider(ids[i], sc[i]);
                                }
                        } catch (const aborted_search_exception &) {
                        }
                }
        }

        // ------------------------------------------------------------------ out-of-line pieces
        inline uint64_t Codecs::PostingsListIterator::cost() const { return dec->indexTermCtx.documents; }
        inline void Codecs::PostingsListIterator::lower(std::vector<uint32_t> &prog, std::vector<double> &w, Similarity::IndexSourceTermsScorer *scorer) const {
                prog.push_back(TRI_TOK(TRI_OP_TERM, dec->termId));
                double weight = 0;
                if (scorer) {
                        std::unique_ptr<Similarity::ScorerWeight> sw(scorer->new_scorer_weight(&dec->term, 1));
                        weight = scorer->device_weight(sw.get());
                }
                w.push_back(weight);
        }
        inline void DocsSetIterators::Phrase::lower(std::vector<uint32_t> &prog, std::vector<double> &w, Similarity::IndexSourceTermsScorer *scorer) const {
                std::vector<std::string> terms;
                for (auto it : its) {
                        it->lower(prog, w, nullptr);
                        terms.push_back(it->dec->term);
                }
                prog.push_back(TRI_TOK(TRI_OP_PHRASE, its.size()));
                double weight = 0;
                if (scorer) { // similarity.h:209-217: a phrase's weight sums its terms' idf
                        std::unique_ptr<Similarity::ScorerWeight> sw(scorer->new_scorer_weight(terms.data(), uint16_t(terms.size())));
                        weight = scorer->device_weight(sw.get());
                }
                w.push_back(weight);
        }
        // a composite iterator driven by hand materialises its docID set with one DocumentsOnly engine run
        inline void DocsSetIterators::Iterator::materialize() {
                if (materialized)
                        return;
                materialized = true;
                auto b = run_batch(isrc, {this}, unsigned(ExecFlags::DocumentsOnly), 0, nullptr);
                size_t n = 0;
                check(tri_batch_docset(b.get(), 0, nullptr, 0, &n));
                docs.resize(n);
                if (n)
                        check(tri_batch_docset(b.get(), 0, docs.data(), n, &n));
        }
        // a postings list materialises docIDs AND freqs through the codec seam (tri_decode_terms)
        inline void Codecs::PostingsListIterator::materialize() {
                if (materialized)
                        return;
                materialized = true;
                const uint32_t n = dec->indexTermCtx.documents;
                docs.resize(n);
                freqs.resize(n);
                if (!n)
                        return;
                uint64_t offs[2];
                check(tri_decode_terms(isrc->handle(), &dec->termId, 1, docs.data(), freqs.data(), offs));
        }
} // namespace trinity_amd
