// host_mirror_test.cpp — exercises the C++ operator surface (trinity_gpu.hpp) the way an application written
// against Trinity would: build an index source, build iterator trees, exec_query with a
// MatchedIndexDocumentsFilter in DocumentsOnly and AccumulatedScoreScheme+BM25 modes, drive a
// PostingsListIterator by hand.  Prints one line per check for the Python test to compare with the oracle.
//   usage: host_mirror_test <index file> <terms file (u32 triples)> <docsCnt> [LUCENE <hits.data file>]
// (with the two extra arguments the segment is opened through Codecs::Lucene::AccessProxy, the way SegmentIndexSource picks the codec by the
//  name in the segment's `id` file, segment_index_source.cpp:172-179; the checks are the same)
#include "../../trinity_amd/csrc/host/trinity_gpu.hpp"
#include <cinttypes>
#include <cstdio>
#include <fstream>

using namespace trinity_amd;

struct Collect final : public MatchedIndexDocumentsFilter {
        std::vector<docid_t> ids;
        std::vector<double> scores;
        void consider(const docid_t id) override { ids.push_back(id); }
        void consider(const docid_t id, const double s) override {
                ids.push_back(id);
                scores.push_back(s);
        }
};

struct EvenOnly final : public IndexDocumentsFilter {
        bool filter(const docid_t id) override { return id & 1; } // disregard odd documents
};

static uint64_t fnv(const std::vector<docid_t> &v) {
        uint64_t h = 1469598103934665603ull;
        for (auto d : v)
                for (int b = 0; b < 4; ++b, d >>= 8)
                        h = (h ^ (d & 0xff)) * 1099511628211ull;
        return h;
}

int main(int argc, char **argv) {
        if (argc < 4)
                return 2;
        std::ifstream fi(argv[1], std::ios::binary);
        std::vector<uint8_t> index((std::istreambuf_iterator<char>(fi)), std::istreambuf_iterator<char>());
        std::ifstream ft(argv[2], std::ios::binary);
        std::vector<char> tb((std::istreambuf_iterator<char>(ft)), std::istreambuf_iterator<char>());
        const size_t nterms = tb.size() / 12;
        std::vector<term_index_ctx> tctx(nterms);
        memcpy(tctx.data(), tb.data(), nterms * 12);
        std::vector<std::string> names(nterms);
        field_statistics fs;
        for (size_t i = 0; i < nterms; ++i) {
                names[i] = "t" + std::to_string(i);
                fs.sumTermsDocs += tctx[i].documents;
                fs.totalTerms += tctx[i].documents != 0;
        }
        fs.docsCnt = uint32_t(strtoul(argv[3], nullptr, 10));
        const std::string codec = argc >= 6 ? argv[4] : "GOOGLE";
        std::vector<uint8_t> hits;
        if (argc >= 6) {
                std::ifstream fh(argv[5], std::ios::binary);
                hits.assign((std::istreambuf_iterator<char>(fh)), std::istreambuf_iterator<char>());
        }
        try {
                IndexSource src(0, index.data(), index.size(), names, tctx, fs, codec, hits.data(), hits.size());
                printf("codec %s\n", src.new_postings_decoder("t0", src.resolve_term_ctx("t0")) ? (codec == "LUCENE" ? "LUCENE" : "GOOGLE") : "?");
                Similarity::IndexSourcesCollectionBM25Scorer bm25;
                std::unique_ptr<Similarity::IndexSourceTermsScorer> scorer(bm25.new_source_scorer(&src));

                auto show = [&](const char *name, const Collect &c) {
                        double sum = 0;
                        for (auto s : c.scores)
                                sum += s;
                        printf("%s n=%zu fnv=%" PRIu64 " score_sum=%.17g\n", name, c.ids.size(), fnv(c.ids), sum);
                };
                { // t0 t1, DocumentsOnly
                        Collect c;
                        exec_query(src.conjunction({src.term("t0"), src.term("t1")}), &src, &c, nullptr, unsigned(ExecFlags::DocumentsOnly));
                        show("and_docs", c);
                }
                { // t0 t1, AccumulatedScoreScheme + BM25
                        Collect c;
                        exec_query(src.conjunction({src.term("t0"), src.term("t1")}), &src, &c, nullptr, unsigned(ExecFlags::AccumulatedScoreScheme), scorer.get());
                        show("and_scored", c);
                }
                { // t0 t1 (t2 OR t3 OR t4), scored
                        Collect c;
                        auto q = src.conjunction({src.term("t0"), src.term("t1"), src.disjunction({src.term("t2"), src.term("t3"), src.term("t4")})});
                        exec_query(q, &src, &c, nullptr, unsigned(ExecFlags::AccumulatedScoreScheme), scorer.get());
                        show("mixed_scored", c);
                }
                { // the span seam window by window, as a composite span drives a child (docset_spans.h:292-296): 8192-document windows over the whole docID
                  // space, some of them "just advance" calls (no proxy, min == max) — ONE device batch for the whole walk
                        struct Feed final : public MatchesProxy {
                                Collect c;
                                void process(relevant_document_provider *p) override { c.consider(p->document(), p->score()); }
                        } feed;
                        auto q = src.conjunction({src.term("t0"), src.term("t1"), src.disjunction({src.term("t2"), src.term("t3"), src.term("t4")})});
                        GpuDocsSetSpan span(q, unsigned(ExecFlags::AccumulatedScoreScheme), scorer.get());
                        isrc_docid_t next = 1;
                        unsigned windows = 0;
                        bool ordered = true;
                        for (isrc_docid_t lo = 1; lo < fs.docsCnt + 8192u; lo += 8192, ++windows) {
                                if (windows % 5 == 4) { // an "advance only" call first: nothing delivered, the estimate of the next match comes back
                                        const isrc_docid_t est = span.process(nullptr, lo, lo);
                                        ordered = ordered && (est == DocIDsEND || est >= lo);
                                }
                                next = span.process(&feed, lo, lo + 8192);
                                ordered = ordered && (next == DocIDsEND || next >= lo + 8192);
                        }
                        for (size_t i = 1; i < feed.c.ids.size(); ++i)
                                ordered = ordered && feed.c.ids[i - 1] < feed.c.ids[i];
                        double sum = 0;
                        for (auto s : feed.c.scores)
                                sum += s;
                        printf("span_windows n=%zu fnv=%" PRIu64 " score_sum=%.17g batches=%u windows=%u ordered=%d end=%d\n", feed.c.ids.size(), fnv(feed.c.ids), sum, span.batches_run(), windows,
                               int(ordered), int(next == DocIDsEND));
                }
                { // t3 OR t7 with an IndexDocumentsFilter
                        Collect c;
                        EvenOnly even;
                        exec_query(src.disjunction({src.term("t3"), src.term("t7")}), &src, &c, &even, unsigned(ExecFlags::DocumentsOnly));
                        show("or_even", c);
                }
                { // "t0 t1" t2 : Phrase iterator inside a conjunction, scored (phrase weight = sum of its terms' idf)
                        Collect c;
                        exec_query(src.conjunction({src.phrase({"t0", "t1"}), src.term("t2")}), &src, &c, nullptr, unsigned(ExecFlags::AccumulatedScoreScheme), scorer.get());
                        show("phrase_scored", c);
                }
                { // "t0 t1" t0 : the standalone t0 scores with its OWN ScorerWeight, not with the (unused) weight of the phrase member
                        Collect c;
                        exec_query(src.conjunction({src.phrase({"t0", "t1"}), src.term("t0")}), &src, &c, nullptr, unsigned(ExecFlags::AccumulatedScoreScheme), scorer.get());
                        show("phrase_and_member_scored", c);
                }
                { // "t0 t1" "t0 t2" : two phrases that start with the same term keep their own weights
                        Collect c;
                        exec_query(src.conjunction({src.phrase({"t0", "t1"}), src.phrase({"t0", "t2"})}), &src, &c, nullptr, unsigned(ExecFlags::AccumulatedScoreScheme), scorer.get());
                        show("two_phrases_scored", c);
                }
                { // t3 t5 NOT (t1 OR t2): DocsSetIterators::Filter over a conjunction, scored (the excluded side does not score)
                        Collect c;
                        auto q = src.filter(src.conjunction({src.term("t3"), src.term("t5")}), src.disjunction({src.term("t1"), src.term("t2")}));
                        exec_query(q, &src, &c, nullptr, unsigned(ExecFlags::AccumulatedScoreScheme), scorer.get());
                        show("not_scored", c);
                }
                { // t3 t1 <t5 OR t2>: DocsSetIterators::Optional — main's documents, the optional side only scores where it matches
                        Collect c;
                        auto q = src.optional(src.conjunction({src.term("t3"), src.term("t1")}), src.disjunction({src.term("t5"), src.term("t2")}));
                        exec_query(q, &src, &c, nullptr, unsigned(ExecFlags::AccumulatedScoreScheme), scorer.get());
                        show("optional_scored", c);
                }
                { // [t0, t1 t2, t3 OR t4] with threshold 2: DocsSetIterators::DisjunctionSome over a term, a conjunction and a disjunction
                        Collect c;
                        auto q = src.some({src.term("t0"), src.conjunction({src.term("t1"), src.term("t2")}), src.disjunction({src.term("t3"), src.term("t4")})}, 2);
                        exec_query(q, &src, &c, nullptr, unsigned(ExecFlags::AccumulatedScoreScheme), scorer.get());
                        show("some_scored", c);
                        Collect d;
                        exec_query(q, &src, &d, nullptr, unsigned(ExecFlags::DocumentsOnly), nullptr);
                        show("some_docs", d);
                }
                { // t5 OR (t1 NOT (t2 t3)): a Filter whose excluded side is a conjunction, under a disjunction
                        Collect c;
                        auto q = src.disjunction({src.term("t5"), src.filter(src.term("t1"), src.conjunction({src.term("t2"), src.term("t3")}))});
                        exec_query(q, &src, &c, nullptr, unsigned(ExecFlags::AccumulatedScoreScheme), scorer.get());
                        show("tree_scored", c);
                }
                { // the other scorers of similarity.h through the same seam
                        Similarity::IndexSourcesCollectionTFIDFScorer tfidf;
                        std::unique_ptr<Similarity::IndexSourceTermsScorer> s2(tfidf.new_source_scorer(&src));
                        Collect c;
                        exec_query(src.conjunction({src.term("t0"), src.term("t1"), src.disjunction({src.term("t2"), src.term("t3")})}), &src, &c, nullptr,
                                   unsigned(ExecFlags::AccumulatedScoreScheme), s2.get());
                        show("tfidf_scored", c);
                        Similarity::IndexSourcesCollectionTrivialScorer trivial;
                        std::unique_ptr<Similarity::IndexSourceTermsScorer> s3(trivial.new_source_scorer(&src));
                        Collect c3;
                        exec_query(src.conjunction({src.term("t0"), src.term("t1")}), &src, &c3, nullptr, unsigned(ExecFlags::AccumulatedScoreScheme), s3.get());
                        show("trivial_scored", c3);
                }
                { // the default execution mode: consider(const matched_document &) with the matched terms and their hits
                        struct Rich final : public MatchedIndexDocumentsFilter {
                                size_t n{0}, terms{0}, hits{0};
                                uint64_t h{1469598103934665603ull};
                                void consider(const matched_document &m) override {
                                        ++n;
                                        terms += m.matchedTermsCnt;
                                        // order-independent digest over (doc, term token, freq, positions)
                                        for (uint16_t i = 0; i < m.matchedTermsCnt; ++i) {
                                                const auto &mt = m.matchedTerms[i];
                                                uint64_t x = 1469598103934665603ull;
                                                auto mix = [&](uint64_t v) { x = (x ^ v) * 1099511628211ull; };
                                                mix(m.id);
                                                mix(strtoul(mt.queryCtx->term.token.c_str() + 1, nullptr, 10));
                                                mix(mt.hits->freq);
                                                for (uint16_t k = 0; k < mt.hits->freq; ++k)
                                                        mix(mt.hits->all[k].pos);
                                                h += x;
                                                hits += mt.hits->freq;
                                        }
                                }
                        } rich;
                        exec_query(src.conjunction({src.term("t0"), src.term("t1"), src.disjunction({src.term("t2"), src.term("t3"), src.term("t4")})}), &src, &rich);
                        printf("rich n=%zu terms=%zu hits=%zu digest=%" PRIu64 "\n", rich.n, rich.terms, rich.hits, rich.h);
                }
                { // masked documents: every 3rd document of the segment was superseded by a newer one
                        std::vector<docid_t> masked;
                        for (docid_t d = 3; d <= fs.docsCnt; d += 3)
                                masked.push_back(d);
                        src.set_masked_documents(masked);
                        Collect c;
                        exec_query(src.conjunction({src.term("t0"), src.term("t1")}), &src, &c, nullptr, unsigned(ExecFlags::AccumulatedScoreScheme), scorer.get());
                        show("masked_scored", c);
                        src.set_masked_documents({});
                }
                { // the same through exec_query's own signature (exec.h:50): a registry over the lists of two newer sources
                        updated_documents newer[2];
                        for (docid_t d = 3; d <= fs.docsCnt; d += 3)
                                newer[d & 1].ids.push_back(d);
                        auto reg = masked_documents_registry::make(newer, 2);
                        Collect c;
                        exec_query(src.conjunction({src.term("t0"), src.term("t1")}), &src, reg.get(), &c, nullptr, unsigned(ExecFlags::AccumulatedScoreScheme), scorer.get());
                        show("masked_registry", c);
                        Collect c2;
                        exec_query(src.conjunction({src.term("t0"), src.term("t1")}), &src, static_cast<masked_documents_registry *>(nullptr), &c2, nullptr,
                                   unsigned(ExecFlags::DocumentsOnly));
                        show("no_registry", c2);
                        // the registry is per call (exec.h:50): the registry-less exec_query that follows sees the source's own set again — none here,
                        // then a set of the application's own that a call WITH a registry must leave in place
                        Collect c3;
                        exec_query(src.conjunction({src.term("t0"), src.term("t1")}), &src, reg.get(), &c3, nullptr, unsigned(ExecFlags::DocumentsOnly));
                        Collect c4;
                        exec_query(src.conjunction({src.term("t0"), src.term("t1")}), &src, &c4, nullptr, unsigned(ExecFlags::DocumentsOnly));
                        show("after_registry", c4);
                        std::vector<docid_t> own;
                        for (docid_t d = 5; d <= fs.docsCnt; d += 5)
                                own.push_back(d);
                        src.set_masked_documents(own);
                        Collect c5;
                        exec_query(src.conjunction({src.term("t0"), src.term("t1")}), &src, reg.get(), &c5, nullptr, unsigned(ExecFlags::DocumentsOnly));
                        Collect c6;
                        exec_query(src.conjunction({src.term("t0"), src.term("t1")}), &src, &c6, nullptr, unsigned(ExecFlags::DocumentsOnly));
                        show("own_set_restored", c6);
                        src.set_masked_documents({});
                }
                { // unknown term => no documents
                        Collect c;
                        exec_query(src.conjunction({src.term("t0"), src.term("nosuchterm")}), &src, &c, nullptr, unsigned(ExecFlags::DocumentsOnly));
                        show("unknown", c);
                }
                { // codec seam by hand: next()/advance()/freq
                        auto it = src.term("t5");
                        uint64_t h = 1469598103934665603ull;
                        uint32_t n = 0;
                        for (auto id = it->next(); id != DocIDsEND; id = it->next(), ++n)
                                h = (h ^ (uint64_t(id) * 31 + it->freq)) * 1099511628211ull;
                        auto it2 = src.term("t5");
                        const auto a = it2->advance(1000);
                        const auto f = it2->freq;
                        const auto b2 = it2->advance(1000);
                        printf("pli n=%u h=%" PRIu64 " adv1000=%u freq=%u again=%u\n", n, h, a, unsigned(f), b2);
                }
                { // batched
                        Collect c0, c1;
                        exec_queries({src.conjunction({src.term("t1"), src.term("t2")}), src.disjunction({src.term("t8"), src.term("t9")})}, &src, {&c0, &c1},
                                     unsigned(ExecFlags::DocumentsOnly));
                        show("batch0", c0);
                        show("batch1", c1);
                }
                try { // error behaviour: mutually exclusive flags (exec.h:45-48)
                        Collect c;
                        exec_query(src.term("t0"), &src, &c, nullptr, 3);
                        printf("flags no-throw\n");
                } catch (const invalid_argument &e) {
                        printf("flags invalid_argument\n");
                }
        } catch (const std::exception &e) {
                printf("EXCEPTION %s\n", e.what());
                return 1;
        }
        return 0;
}
