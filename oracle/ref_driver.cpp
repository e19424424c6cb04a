// ref_driver.cpp — drives the GENUINE reference (compiled from /root/reference where it lies; see
// oracle/Makefile) over this repo's deterministic corpus, and prints one JSON line per command.
//
// TEST INFRASTRUCTURE ONLY.  This file is new code written against the reference's public API
// (Codecs::Google::IndexSession/Encoder, IndexSource, exec_query, Similarity::…BM25Scorer); it contains
// no reference source.  It exists to (1) pin oracle/trinity_oracle.c against the real implementation and
// (2) produce the fixtures under tests/golden/ (see tests/golden/make_golden.py).  It cannot run on the
// GPU box (no /root/reference there) — only its prebuilt binary in oracle/_ref/ travels.
//
// Commands (stdin, one per line):
//   index                          -> {len, fnv, terms_fnv, postings, totalTerms}
//   dumpindex <path>               -> writes raw index bytes + term table (u32 triples) to <path>{.index,.terms}
//   decode <term>                  -> {n, docs_fnv, freqs_fnv, first[], last[]}  via PostingsListIterator::next()
//   advance <term> <seed> <steps>  -> {trace_fnv, n} seeded advance()/next() mix, hash of (doc,freq) after each op
//   positions <term> <everyNth>    -> {fnv} materialize_hits positions of every Nth document
//   query <flags> <k> <text…>      -> {n, fnv, first[], last[], score_sum, top[[doc,score]…]}  exec_query()
//   queryfull <flags> <text…>      -> as query plus full docs[] (+scores[])
//   hits <term>                    -> {docs, fnv} every document's (doc, freq, {pos, payloadLen, payload}…) via materialize_hits
//   timed <flags> <budget s> <count> [threads] + <count> query lines -> {queries, matches, seconds, counts[]}: exec_query timed (bench.py's cpu_baseline of kind "reference")
//   commit <seed> <documents> <vocab> <maxdoc> -> a SegmentIndexSession's input in insertion order and the index / term chunks its commit() wrote
//   merge <seed> <parts> <terms> <maxdoc> -> the input postings of <parts> small segments and the chunks IndexSession::merge writes for them
//
// `ref_driver edge` (instead of D V slots seed) indexes the EDGE corpus below — a few thousand hand-shaped documents over 8 terms
// that put the codec's corner cases into reference-produced bytes: hits with payloads of changing and of constant length, a
// position-0 hit that carries a payload (counted) and one that does not (ignored: a document of frequency 0), a document with
// more than 65535 hits (freq is tokenpos_t: it wraps), positions up to MaxPosition - 1, repeated positions.
#include "exec.h"
#include "compilation_ctx.h"
#include "google_codec.h"
#include "indexer.h"
#include "terms.h"
#include "trinity_oracle.h" // corpus generator + hashing only (this repo's code)
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <algorithm>
#include <atomic>
#include <thread>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <sys/wait.h>
#include <unistd.h>
#include <unordered_map>
#include <vector>

using namespace Trinity;

namespace {
        uint64_t fnv_bytes(const uint8_t *p, size_t n, uint64_t h = 1469598103934665603ull);
        struct Collector final : public MatchedIndexDocumentsFilter {
                std::vector<docid_t> ids;
                std::vector<double> scores;
                // default ("rich match") mode: what consider(const matched_document &) is handed, in a canonical form — the
                // matched terms sorted by term rank (collect_doc_matching_terms visits them in tree / heap order,
                // queryexec_ctx.cpp:382-520), each with its freq and positions
                uint64_t richFnv{1469598103934665603ull}, termsTotal{0}, hitsTotal{0};
                std::vector<std::vector<uint32_t>> richDocs; // first few documents in full: doc, nterms, {rank, freq, pos...}...
                void consider(const matched_document &m) override {
                        ids.push_back(m.id);
                        std::vector<std::vector<uint32_t>> ts;
                        for (uint16_t i = 0; i < m.matchedTermsCnt; ++i) {
                                const auto &mt = m.matchedTerms[i];
                                const auto tok = mt.queryCtx->term.token;
                                uint32_t rank = 0;
                                for (uint32_t k = 1; k < tok.size(); ++k)
                                        rank = rank * 10 + uint32_t(tok.data()[k] - '0');
                                std::vector<uint32_t> one{rank, uint32_t(mt.hits->freq)};
                                for (uint32_t h = 0; h < mt.hits->freq; ++h)
                                        one.push_back(mt.hits->all[h].pos);
                                ts.push_back(std::move(one));
                        }
                        std::sort(ts.begin(), ts.end());
                        std::vector<uint32_t> flat{uint32_t(m.id), uint32_t(ts.size())};
                        for (const auto &one : ts) {
                                flat.insert(flat.end(), one.begin(), one.end());
                                hitsTotal += one[1];
                        }
                        termsTotal += ts.size();
                        richFnv = fnv_bytes(reinterpret_cast<const uint8_t *>(flat.data()), flat.size() * 4, richFnv);
                        static const size_t richLimit = getenv("REF_RICH_DOCS") ? size_t(atol(getenv("REF_RICH_DOCS"))) : 24; // (debugging aid: dump more documents in full)
                        if (richDocs.size() < richLimit)
                                richDocs.push_back(std::move(flat));
                }
                void consider(const docid_t id) override { ids.push_back(id); }
                void consider(const docid_t id, const double score) override {
                        ids.push_back(id);
                        scores.push_back(score);
                }
        };

        // `filter <seed> <permille>`: the documents an application-side IndexDocumentsFilter (matches.h:198-201) rules out — document d when
        // splitmix64(seed + d) % 1000 < permille (a rule instead of a list: the fixture stays small, the tests rebuild the set).  exec_query
        // tests it for every match right before consider(), where it tests masked_documents_registry::test (exec.cpp:1095-1150 and the
        // sibling handlers of the other two modes): with the registry's own bank / bloom structures unbuildable here (docidupdates.cpp needs
        // boost), this pins WHERE and HOW a dropped document disappears — before consider, in all three modes — with reference code
        struct HashFilter final : public IndexDocumentsFilter {
                uint64_t seed{0};
                uint32_t permille{0};
                static uint64_t mix(uint64_t x) {
                        x += 0x9e3779b97f4a7c15ull;
                        x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
                        x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
                        return x ^ (x >> 31);
                }
                bool filter(const docid_t id) override { return mix(seed + id) % 1000u < permille; }
        };

        // A custom IndexSource over an in-memory Google-codec index (index_source.h:19-24 invites exactly this)
        struct MemIndexSource final : public IndexSource {
                std::vector<term_index_ctx> terms;
                Codecs::Google::AccessProxy *access{nullptr};
                field_statistics fs;

                term_index_ctx resolve_term_ctx(const str8_t term) override {
                        if (term.size() < 2 || term.data()[0] != 't')
                                return {};
                        uint64_t r = 0;
                        for (uint32_t i = 1; i < term.size(); ++i) {
                                const char c = term.data()[i];
                                if (c < '0' || c > '9')
                                        return {};
                                r = r * 10 + uint32_t(c - '0');
                        }
                        if (r >= terms.size())
                                return {};
                        return terms[r];
                }
                Codecs::Decoder *new_postings_decoder(const str8_t, const term_index_ctx ctx) override { return access->new_decoder(ctx); }
                field_statistics default_field_stats() override { return fs; }
                bool index_empty() const override { return false; }
                tokenpos_t maxPos{8192}; // index_source.h:134 default; the edge corpus indexes up to MaxPosition - 1 and says so
                tokenpos_t max_indexed_position() const override { return maxPos; }
        };

        uint64_t fnv_bytes(const uint8_t *p, size_t n, uint64_t h) {
                for (size_t i = 0; i < n; ++i)
                        h = (h ^ p[i]) * 1099511628211ull;
                return h;
        }
        uint64_t fnv_u32(uint32_t v, uint64_t h) { return fnv_bytes(reinterpret_cast<const uint8_t *>(&v), 4, h); }

        // ---- `tree`: what the reference's own compile_query (compilation_ctx.cpp:1738-1758) turns a query into — the exec_node tree
        // exec.cpp:253-449 builds its iterators from — printed as nested JSON.  Terms print as their corpus indices ("t17" -> 17).
        struct DumpCtx final : public compilation_ctx {
                std::vector<std::string> names; // exec term id - 1 -> token
                uint16_t resolve_query_term(const str8_t term) override {
                        const std::string t(term.data(), term.size());
                        for (size_t i = 0; i < names.size(); ++i)
                                if (names[i] == t)
                                        return uint16_t(i + 1);
                        names.push_back(t);
                        return uint16_t(names.size());
                }
                long term_index(const exec_term_id_t id) const { return id && id <= names.size() && names[id - 1].size() > 1 ? atol(names[id - 1].c_str() + 1) : -1; }
        };
        void dump_tree(const DumpCtx &c, const exec_node n, std::string &out) {
                auto terms = [&](const char *op, const exec_term_id_t *ids, size_t cnt) {
                        out += std::string("{\"op\":\"") + op + "\",\"t\":[";
                        for (size_t i = 0; i < cnt; ++i)
                                out += (i ? "," : "") + std::to_string(c.term_index(ids[i]));
                        out += "]}";
                };
                auto kids = [&](const char *op, std::initializer_list<exec_node> ks, const std::string &extra = "") {
                        out += std::string("{\"op\":\"") + op + "\"" + extra + ",\"k\":[";
                        bool first = true;
                        for (const auto &k : ks) {
                                if (!first)
                                        out += ",";
                                first = false;
                                dump_tree(c, k, out);
                        }
                        out += "]}";
                };
                switch (n.fp) {
                        case ENT::matchterm: {
                                const exec_term_id_t id = n.u16;
                                terms("term", &id, 1);
                        } break;
                        case ENT::matchallterms:
                        case ENT::matchanyterms: {
                                const auto run = static_cast<const compilation_ctx::termsrun *>(n.ptr);
                                terms(n.fp == ENT::matchallterms ? "allterms" : "anyterms", run->terms, run->size);
                        } break;
                        case ENT::matchphrase: {
                                const auto p = static_cast<const compilation_ctx::phrase *>(n.ptr);
                                terms("phrase", p->termIDs, p->size);
                        } break;
                        case ENT::matchallphrases:
                        case ENT::matchanyphrases: {
                                const auto run = static_cast<const compilation_ctx::phrasesrun *>(n.ptr);
                                out += std::string("{\"op\":\"") + (n.fp == ENT::matchallphrases ? "allphrases" : "anyphrases") + "\",\"k\":[";
                                for (uint16_t i = 0; i < run->size; ++i) {
                                        if (i)
                                                out += ",";
                                        terms("phrase", run->phrases[i]->termIDs, run->phrases[i]->size);
                                }
                                out += "]}";
                        } break;
                        case ENT::logicaland:
                        case ENT::logicalor:
                        case ENT::logicalnot: {
                                const auto b = static_cast<const compilation_ctx::binop_ctx *>(n.ptr);
                                kids(n.fp == ENT::logicaland ? "and" : n.fp == ENT::logicalor ? "or" : "not", {b->lhs, b->rhs});
                        } break;
                        case ENT::unaryand:
                        case ENT::unarynot:
                        case ENT::consttrueexpr: {
                                const auto u = static_cast<const compilation_ctx::unaryop_ctx *>(n.ptr);
                                kids(n.fp == ENT::unaryand ? "unaryand" : n.fp == ENT::unarynot ? "unarynot" : "consttrueexpr", {u->expr});
                        } break;
                        case ENT::matchsome: {
                                const auto g = static_cast<const compilation_ctx::partial_match_ctx *>(n.ptr);
                                out += "{\"op\":\"some\",\"min\":" + std::to_string(g->min) + ",\"k\":[";
                                for (uint16_t i = 0; i < g->size; ++i) {
                                        if (i)
                                                out += ",";
                                        dump_tree(c, g->nodes[i], out);
                                }
                                out += "]}";
                        } break;
                        case ENT::constfalse:
                                out += "{\"op\":\"false\"}";
                                break;
                        case ENT::consttrue:
                                out += "{\"op\":\"true\"}";
                                break;
                        default:
                                out += "{\"op\":\"other\",\"fp\":" + std::to_string(unsigned(n.fp)) + "}";
                                break;
                }
        }

        // `[a, b, c]` parses as MatchSome with min 1 (queries.cpp:424-448); an application sets the threshold on the node
        void set_matchsome_min(ast_node *n, const uint16_t min) {
                if (!n)
                        return;
                switch (n->type) {
                        case ast_node::Type::BinOp:
                                set_matchsome_min(n->binop.lhs, min);
                                set_matchsome_min(n->binop.rhs, min);
                                break;
                        case ast_node::Type::UnaryOp:
                                set_matchsome_min(n->unaryop.expr, min);
                                break;
                        case ast_node::Type::ConstTrueExpr:
                                set_matchsome_min(n->expr, min);
                                break;
                        case ast_node::Type::MatchSome:
                                n->match_some.min = std::min<uint16_t>(min, n->match_some.size);
                                for (uint16_t i = 0; i < n->match_some.size; ++i)
                                        set_matchsome_min(n->match_some.nodes[i], min);
                                break;
                        default:
                                break;
                }
        }

        void print_u32s(const char *name, const uint32_t *v, size_t n) {
                printf("\"%s\":[", name);
                for (size_t i = 0; i < n; ++i)
                        printf("%s%u", i ? "," : "", v[i]);
                printf("]");
        }
} // namespace

// ---- the edge corpus: per term, per document, the hits (position, payload bytes) handed to Encoder::new_hit
struct EdgeHit {
        uint32_t pos;
        std::vector<uint8_t> payload;
};
static constexpr uint32_t EDGE_D = 3000, EDGE_V = 8;
static std::vector<EdgeHit> edge_hits(const uint32_t t, const uint32_t d, bool &present) {
        std::vector<EdgeHit> h;
        present = false;
        auto bytes = [&](uint32_t n, uint32_t salt) {
                std::vector<uint8_t> b(n);
                for (uint32_t k = 0; k < n; ++k)
                        b[k] = uint8_t(d * 7 + salt * 13 + k);
                return b;
        };
        switch (t) {
                case 0: // every document: 1..4 hits, payload lengths that change from hit to hit, or stay put (d % 5 == 0)
                        present = true;
                        for (uint32_t j = 0; j < 1 + d % 4; ++j)
                                h.push_back({1 + 3 * j + (d & 1), bytes(d % 5 == 0 ? 3 : (d + j) % 9, j)});
                        break;
                case 1:
                        if (d % 3 == 0) { // a position-0 hit WITH a payload: counted (google_codec.cpp:38-74)
                                present = true;
                                h.push_back({0, {uint8_t(d), uint8_t(d >> 8)}});
                        } else if (d % 7 == 0) { // a position-0 hit without payload: ignored, the document is committed with frequency 0
                                present = true;
                                h.push_back({0, {}});
                        }
                        break;
                case 2:
                        if (d == 100) { // 70000 hits: freq wraps in tokenpos_t (codecs.h:217)
                                present = true;
                                for (uint32_t j = 0; j < 70000; ++j)
                                        h.push_back({j / 5 + 1, {}});
                        } else if (d == 200) {
                                present = true;
                                for (uint32_t j = 0; j < 300; ++j)
                                        h.push_back({2 * j + 1, bytes(1, j)});
                        } else if (d >= 50 && d <= 150 && d % 10 == 0) {
                                present = true;
                                h.push_back({3, {}});
                                h.push_back({9, {}});
                        }
                        break;
                case 3:
                        if (d % 11 == 0) { // the last legal positions (trinity_limits.h:15)
                                present = true;
                                h.push_back({16380, {}});
                                h.push_back({16381, bytes(8, 1)});
                                h.push_back({16383, {}});
                        }
                        break;
                case 4:
                        if (d % 4 == 0) {
                                present = true;
                                h.push_back({5, bytes(d % 3, 4)});
                                h.push_back({16382, {}});
                        } else if (d % 4 == 2) {
                                present = true;
                                h.push_back({5, {}});
                        }
                        break;
                case 5:
                        if (d % 4 == 0) {
                                present = true;
                                h.push_back({6, {}});
                                h.push_back({16383, bytes(2, 5)});
                        } else if (d % 4 == 2) {
                                present = true;
                                h.push_back({7, bytes(4, 5)});
                        }
                        break;
                case 6:
                        if (d == 7 || d == 1999 || d == 3000) {
                                present = true;
                                h.push_back({d % 100 + 1, {}});
                        }
                        break;
                default:
                        if (d % 13 == 0) { // repeated positions, payload length 0 -> 8 -> 0
                                present = true;
                                h.push_back({4, {}});
                                h.push_back({4, bytes(8, 7)});
                                h.push_back({4, {}});
                        }
                        break;
        }
        return h;
}

int main(int argc, char **argv) {
        const bool edge = argc >= 2 && std::string(argv[1]) == "edge";
        if (argc < 5 && !edge) {
                fprintf(stderr, "usage: %s D V slots seed < commands   |   %s edge < commands\n", argv[0], argv[0]);
                return 2;
        }
        const uint32_t D = edge ? EDGE_D : strtoul(argv[1], nullptr, 10), V = edge ? EDGE_V : strtoul(argv[2], nullptr, 10),
                       slots = edge ? 0 : strtoul(argv[3], nullptr, 10);
        const uint64_t seed = edge ? 0 : strtoull(argv[4], nullptr, 10);
        to_corpus *corpus = edge ? nullptr : to_corpus_generate(D, V, slots, seed);

        // ---- index with the reference's own encoder (google_codec.cpp:9-176), terms in rank order
        Codecs::Google::IndexSession sess("/tmp");
        sess.begin();
        std::unique_ptr<Codecs::Encoder> enc(sess.new_encoder());
        MemIndexSource *src = new MemIndexSource();
        src->terms.resize(V);
        if (edge)
                src->maxPos = Limits::MaxPosition;
        uint64_t postings = 0;
        uint32_t totalTerms = 0;
        uint64_t edgeHits = 0;
        for (uint32_t t = 0; edge && t < V; ++t) {
                term_index_ctx tctx;
                enc->begin_term();
                for (uint32_t d = 1; d <= D; ++d) {
                        bool present;
                        const auto hs = edge_hits(t, d, present);
                        if (!present)
                                continue;
                        enc->begin_document(d);
                        for (const auto &h : hs) {
                                enc->new_hit(h.pos, {h.payload.data(), uint8_t(h.payload.size())});
                                edgeHits += h.pos || !h.payload.empty();
                        }
                        enc->end_document();
                }
                enc->end_term(&tctx);
                src->terms[t] = tctx;
                postings += tctx.documents;
                ++totalTerms;
        }
        for (uint32_t t = 0; !edge && t < V; ++t) {
                const uint64_t b = corpus->term_off[t], e = corpus->term_off[t + 1];
                if (b == e) {
                        src->terms[t] = term_index_ctx{0, range32_t{0, 0}};
                        continue;
                }
                term_index_ctx tctx;
                enc->begin_term();
                for (uint64_t i = b; i < e;) {
                        const uint32_t d = corpus->tok_doc[i];
                        enc->begin_document(d);
                        for (; i < e && corpus->tok_doc[i] == d; ++i)
                                enc->new_hit(corpus->tok_pos[i], {});
                        enc->end_document();
                }
                enc->end_term(&tctx);
                src->terms[t] = tctx;
                postings += tctx.documents;
                ++totalTerms;
        }
        sess.end();
        // keep the bytes alive + 16 bytes of slack
        std::vector<uint8_t> index(sess.indexOut.size() + 16, 0);
        memcpy(index.data(), sess.indexOut.data(), sess.indexOut.size());
        const size_t indexLen = sess.indexOut.size();
        Codecs::Google::AccessProxy access("/tmp", index.data());
        src->access = &access;
        src->fs.sumTermHits = edge ? edgeHits : corpus->ntokens;
        src->fs.totalTerms = totalTerms;
        src->fs.sumTermsDocs = postings;
        src->fs.docsCnt = D;

        IndexSourcesCollection collection;
        collection.insert(src);
        // NB: collection.commit() only gathers masked documents (index_source.cpp:3-30); there are none.
        Similarity::IndexSourcesCollectionBM25Scorer bm25;
        Similarity::IndexSourcesCollectionTFIDFScorer tfidf;
        Similarity::IndexSourcesCollectionTrivialScorer trivial;
        Similarity::IndexSourcesCollectionTermsScorer *collScorer = &bm25; // `sim bm25|tfidf|trivial` switches it
        std::string simName = "bm25";
        auto noMasked = masked_documents_registry::make(nullptr, 0);
        HashFilter docFilter; // (permille 0: off — exec_query is handed no filter at all)

        std::string line;
        while (std::getline(std::cin, line)) {
                std::istringstream is(line);
                std::string cmd;
                if (!(is >> cmd) || cmd[0] == '#')
                        continue;
                if (cmd == "index") {
                        uint64_t th = 1469598103934665603ull;
                        for (uint32_t t = 0; t < V; ++t) {
                                th = fnv_u32(src->terms[t].documents, th);
                                th = fnv_u32(src->terms[t].indexChunk.offset, th);
                                th = fnv_u32(src->terms[t].indexChunk.size(), th);
                        }
                        printf("{\"cmd\":\"index\",\"len\":%zu,\"fnv\":\"%" PRIu64 "\",\"terms_fnv\":\"%" PRIu64 "\",\"postings\":%" PRIu64 ",\"totalTerms\":%u,\"sumTermHits\":%" PRIu64 ",\"docsCnt\":%u}\n", indexLen,
                               fnv_bytes(index.data(), indexLen), th, postings, totalTerms, uint64_t(src->fs.sumTermHits), D);
                } else if (cmd == "dumpindex") {
                        std::string path;
                        is >> path;
                        FILE *f = fopen((path + ".index").c_str(), "wb");
                        fwrite(index.data(), 1, indexLen, f);
                        fclose(f);
                        f = fopen((path + ".terms").c_str(), "wb");
                        for (uint32_t t = 0; t < V; ++t) {
                                const uint32_t rec[3] = {src->terms[t].documents, src->terms[t].indexChunk.offset, src->terms[t].indexChunk.size()};
                                fwrite(rec, 4, 3, f);
                        }
                        fclose(f);
                        printf("{\"cmd\":\"dumpindex\",\"len\":%zu}\n", indexLen);
                } else if (cmd == "decode") {
                        uint32_t t;
                        is >> t;
                        std::unique_ptr<Codecs::Decoder> dec(access.new_decoder(src->terms[t]));
                        std::unique_ptr<Codecs::PostingsListIterator> it(dec->new_iterator());
                        std::vector<uint32_t> docs;
                        uint64_t fh = 1469598103934665603ull;
                        for (auto id = it->next(); id != DocIDsEND; id = it->next()) {
                                docs.push_back(id);
                                fh = fnv_u32(it->freq, fh);
                        }
                        const size_t n = docs.size(), k = n < 8 ? n : 8;
                        printf("{\"cmd\":\"decode\",\"term\":%u,\"n\":%zu,\"docs_fnv\":\"%" PRIu64 "\",\"freqs_fnv\":\"%" PRIu64 "\",", t, n, to_fnv1a_docs(docs.data(), n), fh);
                        print_u32s("first", docs.data(), k);
                        printf(",");
                        print_u32s("last", docs.data() + (n - k), k);
                        printf("}\n");
                } else if (cmd == "advance") {
                        uint32_t t, steps;
                        uint64_t s;
                        is >> t >> s >> steps;
                        std::unique_ptr<Codecs::Decoder> dec(access.new_decoder(src->terms[t]));
                        std::unique_ptr<Codecs::PostingsListIterator> it(dec->new_iterator());
                        uint64_t h = 1469598103934665603ull, st = s;
                        uint32_t done = 0;
                        // seeded mix: next(), small forward jumps, block-sized jumps, big jumps, targets <= current
                        for (; done < steps && it->current() != DocIDsEND; ++done) {
                                const uint64_t r = to_splitmix64(&st);
                                const uint32_t cur = it->current();
                                uint32_t id;
                                switch (r & 7) {
                                        case 0:
                                        case 1:
                                                id = it->next();
                                                break;
                                        case 2:
                                                id = cur ? it->advance(cur) : it->next(); // target == current
                                                break;
                                        case 3:
                                                id = it->advance(cur + 1 + uint32_t((r >> 8) % 3));
                                                break;
                                        case 4:
                                                id = it->advance(cur + 1 + uint32_t((r >> 8) % 64));
                                                break;
                                        case 5:
                                                id = it->advance(cur + 1 + uint32_t((r >> 8) % 2048));
                                                break;
                                        case 6:
                                                id = it->advance(cur + 1 + uint32_t((r >> 8) % (D / 16 + 1)));
                                                break;
                                        default:
                                                id = cur > 3 ? it->advance(cur - uint32_t((r >> 8) % 3)) : it->next(); // target <= current
                                                break;
                                }
                                h = fnv_u32(id, h);
                                h = fnv_u32(id == DocIDsEND ? 0 : it->freq, h);
                        }
                        printf("{\"cmd\":\"advance\",\"term\":%u,\"seed\":\"%" PRIu64 "\",\"steps\":%u,\"done\":%u,\"trace_fnv\":\"%" PRIu64 "\"}\n", t, s, steps, done, h);
                } else if (cmd == "positions") {
                        uint32_t t, nth;
                        is >> t >> nth;
                        std::unique_ptr<Codecs::Decoder> dec(access.new_decoder(src->terms[t]));
                        std::unique_ptr<Codecs::PostingsListIterator> it(dec->new_iterator());
                        DocWordsSpace dws(8192);
                        std::vector<term_hit> hits(65536);
                        uint64_t h = 1469598103934665603ull;
                        uint32_t i = 0, cnt = 0;
                        for (auto id = it->next(); id != DocIDsEND; id = it->next(), ++i) {
                                if (i % nth)
                                        continue;
                                const auto f = it->freq;
                                dws.reset();
                                it->materialize_hits(&dws, hits.data());
                                h = fnv_u32(id, h);
                                for (uint32_t k = 0; k < f; ++k)
                                        h = fnv_u32(hits[k].pos, h);
                                ++cnt;
                        }
                        printf("{\"cmd\":\"positions\",\"term\":%u,\"nth\":%u,\"docs\":%u,\"fnv\":\"%" PRIu64 "\"}\n", t, nth, cnt, h);
                } else if (cmd == "hits") {
                        uint32_t t;
                        is >> t;
                        std::unique_ptr<Codecs::Decoder> dec(access.new_decoder(src->terms[t]));
                        std::unique_ptr<Codecs::PostingsListIterator> it(dec->new_iterator());
                        DocWordsSpace dws(Limits::MaxPosition);
                        std::vector<term_hit> hits(1 << 17);
                        uint64_t h = 1469598103934665603ull, hpos = 1469598103934665603ull;
                        uint32_t cnt = 0;
                        for (auto id = it->next(); id != DocIDsEND; id = it->next(), ++cnt) {
                                const auto f = it->freq; // tokenpos_t: what the iterator exposes
                                dws.reset();
                                it->materialize_hits(&dws, hits.data());
                                h = fnv_u32(id, fnv_u32(f, h));
                                hpos = fnv_u32(id, fnv_u32(f, hpos));
                                for (uint32_t k = 0; k < f; ++k) {
                                        h = fnv_u32(hits[k].pos, h);
                                        h = fnv_u32(hits[k].payloadLen, h);
                                        h = fnv_bytes(reinterpret_cast<const uint8_t *>(&hits[k].payload), 8, h);
                                        hpos = fnv_u32(hits[k].pos, hpos);
                                }
                        }
                        printf("{\"cmd\":\"hits\",\"term\":%u,\"docs\":%u,\"fnv\":\"%" PRIu64 "\",\"pos_fnv\":\"%" PRIu64 "\"}\n", t, cnt, h, hpos);
                } else if (cmd == "tree") {
                        uint32_t someMin = 0;
                        is >> someMin;
                        std::string text;
                        std::getline(is, text);
                        while (!text.empty() && text[0] == ' ')
                                text.erase(0, 1);
                        query q{str32_t(text.data(), uint32_t(text.size())), default_token_parser_impl,
                                unsigned(ast_parser::Flags::ParseConstTrueExpr) | unsigned(ast_parser::Flags::ParseMatchSomeExpr)};
                        if (someMin)
                                set_matchsome_min(q.root, uint16_t(someMin));
                        std::string tree = "null";
                        if (q.root) {
                                DumpCtx ctx;
                                const auto root = compile_query(q.root, ctx);
                                tree.clear();
                                dump_tree(ctx, root, tree);
                        }
                        printf("{\"cmd\":\"tree\",\"min\":%u,\"q\":\"", someMin);
                        for (char c : text) {
                                if (c == '"')
                                        printf("\\\"");
                                else
                                        putchar(c);
                        }
                        printf("\",\"tree\":%s}\n", tree.c_str());
                } else if (cmd == "merge") {
                        // `merge <seed> <participants> <terms> <maxdoc>`: Codecs::Google::IndexSession::merge (google_codec.cpp:186-438) as
                        // MergeCandidatesCollection::merge drives it (merge.cpp:254-288): small segments written by the reference's encoder, the most
                        // recent first, then per output term begin_term / merge(participants holding documents) / end_term into a fresh session.
                        // The line carries the INPUT postings too, so a test needs no generator of its own.  The masked registries are empty
                        // (their scanner lives in docidupdates.cpp, which needs boost: unbuildable here) — what a non-empty one does is the
                        // single `maskedDocsReg->test(lowestDID)` of google_codec.cpp:398, on the registry of the participant that supplies the
                        // document
                        uint64_t mseed;
                        uint32_t nparts, G, maxdoc;
                        is >> mseed >> nparts >> G >> maxdoc;
                        uint64_t st = mseed;
                        auto rnd = [&]() { return st = HashFilter::mix(st + 0x632be59bd9b4e019ull); };
                        struct Part {
                                std::unique_ptr<Codecs::Google::IndexSession> sess;
                                std::vector<uint8_t> bytes;
                                std::unique_ptr<Codecs::Google::AccessProxy> ap;
                                std::vector<term_index_ctx> tctx; // per global term (documents 0: not held)
                        };
                        std::vector<Part> parts(nparts);
                        std::string out = "{\"cmd\":\"merge\",\"seed\":" + std::to_string(mseed) + ",\"maxdoc\":" + std::to_string(maxdoc) + ",\"parts\":[";
                        for (uint32_t pi = 0; pi < nparts; ++pi) {
                                Part &P = parts[pi];
                                P.sess.reset(new Codecs::Google::IndexSession("/tmp"));
                                P.sess->begin();
                                std::unique_ptr<Codecs::Encoder> penc(P.sess->new_encoder());
                                P.tctx.assign(G, term_index_ctx{0, range32_t{0, 0}});
                                out += pi ? ",[" : "[";
                                bool firstTerm = true;
                                for (uint32_t g = 0; g < G; ++g) {
                                        if (rnd() % 10 >= 7)
                                                continue; // the segment does not hold the term
                                        static const uint32_t sizes[] = {1, 2, 31, 32, 33, 64, 70, 129, 300};
                                        const uint32_t want = sizes[rnd() % 9];
                                        std::vector<uint32_t> docs;
                                        for (uint32_t k = 0; k < want; ++k)
                                                docs.push_back(1 + uint32_t(rnd() % (maxdoc - 1)));
                                        std::sort(docs.begin(), docs.end());
                                        docs.erase(std::unique(docs.begin(), docs.end()), docs.end());
                                        const bool payloads = g % 3 == 0;
                                        term_index_ctx tctx;
                                        penc->begin_term();
                                        out += std::string(firstTerm ? "" : ",") + "{\"g\":" + std::to_string(g) + ",\"docs\":[";
                                        firstTerm = false;
                                        std::string hs;
                                        for (size_t di = 0; di < docs.size(); ++di) {
                                                out += (di ? "," : "") + std::to_string(docs[di]);
                                                penc->begin_document(docs[di]);
                                                const uint32_t freq = uint32_t(rnd() % 4);
                                                std::vector<uint32_t> pos;
                                                for (uint32_t k = 0; k < freq; ++k)
                                                        pos.push_back(1 + uint32_t(rnd() % 2000));
                                                std::sort(pos.begin(), pos.end());
                                                hs += std::string(di ? "," : "") + "[";
                                                for (uint32_t k = 0; k < freq; ++k) {
                                                        const uint8_t plen = payloads ? uint8_t(rnd() % 9) : 0;
                                                        uint64_t pv = plen ? rnd() : 0;
                                                        if (plen && plen < 8)
                                                                pv &= (1ull << (8 * plen)) - 1;
                                                        penc->new_hit(tokenpos_t(pos[k]), {reinterpret_cast<const uint8_t *>(&pv), plen});
                                                        hs += std::string(k ? "," : "") + "[" + std::to_string(pos[k]) + "," + std::to_string(plen) + ",\"" + std::to_string(pv) + "\"]";
                                                }
                                                hs += "]";
                                                penc->end_document();
                                        }
                                        penc->end_term(&tctx);
                                        P.tctx[g] = tctx;
                                        out += "],\"hits\":[" + hs + "]}";
                                }
                                out += "]";
                                P.sess->end();
                                P.bytes.assign(P.sess->indexOut.size() + 16, 0);
                                memcpy(P.bytes.data(), P.sess->indexOut.data(), P.sess->indexOut.size());
                                P.ap.reset(new Codecs::Google::AccessProxy("/tmp", P.bytes.data()));
                        }
                        Codecs::Google::IndexSession osess("/tmp");
                        osess.begin();
                        std::unique_ptr<Codecs::Encoder> oenc(osess.new_encoder());
                        out += "],\"out\":[";
                        bool firstOut = true;
                        for (uint32_t g = 0; g < G; ++g) {
                                std::vector<Codecs::IndexSession::merge_participant> mp;
                                for (uint32_t pi = 0; pi < nparts; ++pi) // participant 0 is the most recent (merge.cpp: candidates ordered by generation, latest first)
                                        if (parts[pi].tctx[g].documents)
                                                mp.push_back({parts[pi].ap.get(), parts[pi].tctx[g], masked_documents_registry::make(nullptr, 0).release()});
                                if (mp.empty())
                                        continue;
                                term_index_ctx tctx;
                                const size_t before = osess.indexOut.size();
                                oenc->begin_term();
                                osess.merge(mp.data(), uint16_t(mp.size()), oenc.get());
                                oenc->end_term(&tctx);
                                for (auto &m : mp)
                                        delete m.maskedDocsReg;
                                out += std::string(firstOut ? "" : ",") + "{\"g\":" + std::to_string(g) + ",\"participants\":" + std::to_string(mp.size()) + ",\"documents\":" + std::to_string(tctx.documents) +
                                       ",\"offset\":" + std::to_string(tctx.indexChunk.offset) + ",\"size\":" + std::to_string(tctx.indexChunk.size()) + ",\"written_from\":" + std::to_string(before) + ",\"chunk\":\"";
                                firstOut = false;
                                static const char hexd[] = "0123456789abcdef";
                                for (uint32_t k = 0; k < tctx.indexChunk.size(); ++k) {
                                        const uint8_t b = reinterpret_cast<const uint8_t *>(osess.indexOut.data())[tctx.indexChunk.offset + k];
                                        out += hexd[b >> 4];
                                        out += hexd[b & 15];
                                }
                                out += "\"}";
                        }
                        osess.end();
                        out += "],\"out_len\":" + std::to_string(osess.indexOut.size()) + "}";
                        puts(out.c_str());
                } else if (cmd == "commit") {
                        // `commit <seed> <documents> <vocab> <maxdoc>`: SegmentIndexSession (indexer.cpp:14-230) fed documents in a shuffled order — every
                        // document a few terms with their hits, some with payloads — and SegmentIndexSession::commit (indexer.cpp:311-560) into a Google
                        // IndexSession.  The line carries the input in INSERTION order with the session's transient term ids, and what commit wrote:
                        // the `index` bytes and, per term, its chunk.  commit ends in persist_segment -> pack_updates (docidupdates.cpp: needs boost,
                        // unbuildable here, left unresolved at link time); by then the encoded index and the dictionary are on disk (indexer.cpp:243,
                        // codecs.cpp:17-27), so the commit runs in a CHILD process that is allowed to die at that call and the files are read here
                        uint64_t cseed;
                        uint32_t ndocs, vocab, maxdoc;
                        is >> cseed >> ndocs >> vocab >> maxdoc;
                        uint64_t st = cseed;
                        auto rnd = [&]() { return st = HashFilter::mix(st + 0x2545f4914f6cdd1dull); };
                        struct Hit {
                                uint32_t pos;
                                uint8_t plen;
                                uint64_t pval;
                        };
                        struct DocIn {
                                uint32_t doc;
                                std::vector<std::pair<uint32_t, std::vector<Hit>>> terms; // (vocabulary index, hits in position order)
                        };
                        std::vector<DocIn> docsIn;
                        {
                                std::vector<uint32_t> ids;
                                while (ids.size() < ndocs) {
                                        const uint32_t d = 1 + uint32_t(rnd() % (maxdoc - 1));
                                        if (std::find(ids.begin(), ids.end(), d) == ids.end())
                                                ids.push_back(d);
                                }
                                // insertion order: 4096-document blocks descending, ascending inside a block — not the (term, document) order commit has to
                                // produce, and a pattern the session's document tracker takes (SparseFixedBitSet::try_set reported a fresh document as
                                // "already committed" for a fully shuffled order)
                                std::sort(ids.begin(), ids.end(), [](uint32_t a, uint32_t b) { return (a >> 12) != (b >> 12) ? (a >> 12) > (b >> 12) : a < b; });
                                for (const uint32_t d : ids) {
                                        DocIn di{d, {}};
                                        const uint32_t nt = 1 + uint32_t(rnd() % 6);
                                        while (di.terms.size() < nt) {
                                                const uint64_t r = rnd();
                                                const uint32_t w = uint32_t((r % vocab) * ((r >> 32) % vocab) / vocab); // (skewed towards the low indices)
                                                bool dup = false;
                                                for (const auto &t : di.terms)
                                                        dup |= t.first == w;
                                                if (dup)
                                                        continue;
                                                std::vector<Hit> hs(1 + rnd() % 3);
                                                std::vector<uint32_t> ps;
                                                for (size_t k = 0; k < hs.size(); ++k)
                                                        ps.push_back(1 + uint32_t(rnd() % 3000));
                                                std::sort(ps.begin(), ps.end());
                                                for (size_t k = 0; k < hs.size(); ++k) {
                                                        hs[k].pos = ps[k];
                                                        hs[k].plen = w % 3 == 0 ? uint8_t(rnd() % 9) : 0;
                                                        hs[k].pval = hs[k].plen ? rnd() : 0;
                                                        if (hs[k].plen && hs[k].plen < 8)
                                                                hs[k].pval &= (1ull << (8 * hs[k].plen)) - 1;
                                                }
                                                di.terms.push_back({w, std::move(hs)});
                                        }
                                        docsIn.push_back(std::move(di));
                                }
                        }
                        char dir[] = "/tmp/trinity_ref_commit_XXXXXX";
                        if (!mkdtemp(dir)) {
                                printf("{\"cmd\":\"commit\",\"error\":\"mkdtemp\"}\n");
                                continue;
                        }
                        fflush(stdout);
                        int pfd[2];
                        if (pipe(pfd)) {
                                printf("{\"cmd\":\"commit\",\"error\":\"pipe\"}\n");
                                continue;
                        }
                        const pid_t pid = fork();
                        if (pid == 0) {
                                close(pfd[0]);
                                Codecs::Google::IndexSession cis(dir);
                                SegmentIndexSession sis;
                                std::string ids; // the session's transient id of every vocabulary index it saw, sent to the parent before commit
                                for (const auto &di : docsIn) {
                                        auto proxy = sis.begin(di.doc);
                                        for (const auto &t : di.terms) {
                                                const std::string name = "w" + std::to_string(t.first);
                                                const uint32_t tid = sis.term_id(str8_t(name.data(), uint8_t(name.size())));
                                                ids += std::to_string(t.first) + ":" + std::to_string(tid) + ",";
                                                for (const auto &h : t.second)
                                                        proxy.insert(tid, tokenpos_t(h.pos), {reinterpret_cast<const uint8_t *>(&h.pval), h.plen});
                                        }
                                        sis.insert(proxy);
                                }
                                if (write(pfd[1], ids.data(), ids.size()) != ssize_t(ids.size()))
                                        _exit(3);
                                close(pfd[1]);
                                sis.commit(&cis); // (does not return: see above)
                                _exit(0);
                        }
                        close(pfd[1]);
                        std::string ids;
                        {
                                char buf[4096];
                                ssize_t got;
                                while ((got = read(pfd[0], buf, sizeof buf)) > 0)
                                        ids.append(buf, size_t(got));
                                close(pfd[0]);
                        }
                        int status = 0;
                        waitpid(pid, &status, 0);
                        auto slurp = [&](const char *name) {
                                std::vector<uint8_t> v;
                                if (FILE *f = fopen((std::string(dir) + "/" + name).c_str(), "rb")) {
                                        uint8_t buf[65536];
                                        size_t got;
                                        while ((got = fread(buf, 1, sizeof buf, f)) > 0)
                                                v.insert(v.end(), buf, buf + got);
                                        fclose(f);
                                }
                                return v;
                        };
                        std::vector<uint8_t> cidx = slurp("index.t");
                        if (cidx.empty())
                                cidx = slurp("index");
                        const std::vector<uint8_t> tdata = slurp("terms.data");
                        std::unordered_map<uint32_t, uint32_t> idOf;
                        for (size_t a = 0; a < ids.size();) {
                                const size_t c = ids.find(':', a), e = ids.find(',', c);
                                idOf[uint32_t(strtoul(ids.c_str() + a, nullptr, 10))] = uint32_t(strtoul(ids.c_str() + c + 1, nullptr, 10));
                                a = e + 1;
                        }
                        static const char hexd[] = "0123456789abcdef";
                        std::string out = "{\"cmd\":\"commit\",\"seed\":" + std::to_string(cseed) + ",\"child\":\"" +
                                          (WIFSIGNALED(status) ? "signal " + std::to_string(WTERMSIG(status)) : "exit " + std::to_string(WEXITSTATUS(status))) + "\",\"docs\":[";
                        for (size_t i = 0; i < docsIn.size(); ++i) {
                                out += std::string(i ? "," : "") + "{\"d\":" + std::to_string(docsIn[i].doc) + ",\"terms\":[";
                                for (size_t k = 0; k < docsIn[i].terms.size(); ++k) {
                                        const auto &t = docsIn[i].terms[k];
                                        out += std::string(k ? "," : "") + "{\"w\":" + std::to_string(t.first) + ",\"id\":" + std::to_string(idOf[t.first]) + ",\"hits\":[";
                                        for (size_t h = 0; h < t.second.size(); ++h)
                                                out += std::string(h ? "," : "") + "[" + std::to_string(t.second[h].pos) + "," + std::to_string(t.second[h].plen) + ",\"" + std::to_string(t.second[h].pval) + "\"]";
                                        out += "]}";
                                }
                                out += "]}";
                        }
                        out += "],\"terms\":[";
                        {
                                terms_data_view view({tdata.data(), uint32_t(tdata.size())});
                                bool first = true;
                                for (auto it = view.begin(); it != view.end(); ++it) {
                                        const auto cur = *it;
                                        out += std::string(first ? "" : ",") + "{\"w\":" + std::string(cur.first.data() + 1, cur.first.size() - 1) + ",\"documents\":" + std::to_string(cur.second.documents) +
                                               ",\"offset\":" + std::to_string(cur.second.indexChunk.offset) + ",\"size\":" + std::to_string(cur.second.indexChunk.size()) + "}";
                                        first = false;
                                }
                        }
                        out += "],\"index\":\"";
                        for (const uint8_t b : cidx) {
                                out += hexd[b >> 4];
                                out += hexd[b & 15];
                        }
                        out += "\"}";
                        puts(out.c_str());
                        for (const char *f : {"index.t", "index", "terms.data", "terms.idx", "updated_documents.ids", "codec"})
                                unlink((std::string(dir) + "/" + f).c_str());
                        rmdir(dir);
                } else if (cmd == "timed") {
                        // `timed <flags> <budget seconds> <count>` + <count> lines of query text: bench.py's CPU baseline of kind "reference" — exec_query
                        // (compile_query + the iterator / span loops, exec.cpp) over the queries in order until the budget is spent; the clock runs
                        // around exec_query only (parsing the text stands outside, as the engine's side is handed compiled programs).  Matches go
                        // into a vector (DocumentsOnly) or a (docID, score) pair of vectors, as an application's filter would keep them
                        uint32_t flags, count;
                        double budget;
                        is >> flags >> budget >> count;
                        std::vector<std::string> texts(count);
                        for (auto &t : texts)
                                std::getline(std::cin, t);
                        struct Keep final : public MatchedIndexDocumentsFilter {
                                std::vector<docid_t> ids;
                                std::vector<double> scores;
                                void consider(const docid_t id) override { ids.push_back(id); }
                                void consider(const docid_t id, const double score) override {
                                        ids.push_back(id);
                                        scores.push_back(score);
                                }
                                void consider(const matched_document &m) override { ids.push_back(m.id); }
                        } keep;
                        std::vector<uint64_t> counts;
                        uint64_t matches = 0, ns = 0;
                        for (uint32_t i = 0; i < count; ++i) {
                                query q{str32_t(texts[i].data(), uint32_t(texts[i].size())), default_token_parser_impl,
                                        unsigned(ast_parser::Flags::ParseConstTrueExpr) | unsigned(ast_parser::Flags::ParseMatchSomeExpr)};
                                std::unique_ptr<Similarity::IndexSourceTermsScorer> scorer;
                                if (flags & unsigned(ExecFlags::AccumulatedScoreScheme)) {
                                        collScorer->reset(&collection);
                                        scorer.reset(collScorer->new_source_scorer(src));
                                }
                                keep.ids.clear();
                                keep.scores.clear();
                                struct timespec a, b;
                                clock_gettime(CLOCK_MONOTONIC, &a);
                                exec_query(q, src, noMasked.get(), &keep, nullptr, flags, scorer.get());
                                clock_gettime(CLOCK_MONOTONIC, &b);
                                ns += uint64_t(b.tv_sec - a.tv_sec) * 1000000000ull + uint64_t(b.tv_nsec) - uint64_t(a.tv_nsec);
                                counts.push_back(keep.ids.size());
                                matches += keep.ids.size();
                                if (double(ns) * 1e-9 > budget && i + 1 >= 16)
                                        break;
                        }
                        printf("{\"cmd\":\"timed\",\"flags\":%u,\"queries\":%zu,\"matches\":%" PRIu64 ",\"seconds\":%.6f,\"counts\":[", flags, counts.size(), matches, double(ns) * 1e-9);
                        for (size_t i = 0; i < counts.size(); ++i)
                                printf("%s%" PRIu64, i ? "," : "", counts[i]);
                        printf("]}\n");
                        fflush(stdout); // (the single-thread figure stands whatever happens below)
                        // ... and the same queries, one per thread at a time on <threads> threads (exec_query is re-entrant: one queryexec_ctx per call,
                        // exec.cpp:12; the index source is shared and read-only), drawn from a shared cursor; wall clock around the whole pool
                        unsigned threads = 0;
                        is >> threads;
                        if (threads > 1 && !(flags & unsigned(ExecFlags::AccumulatedScoreScheme))) {
                                const size_t nrun = counts.size();
                                std::vector<uint64_t> mtCounts(nrun, 0);
                                std::atomic<size_t> cursor{0};
                                struct timespec a, b;
                                clock_gettime(CLOCK_MONOTONIC, &a);
                                std::vector<std::thread> pool;
                                for (unsigned t = 0; t < threads; ++t)
                                        pool.emplace_back([&]() {
                                                Keep mine;
                                                for (size_t i; (i = cursor.fetch_add(1)) < nrun;) {
                                                        query q{str32_t(texts[i].data(), uint32_t(texts[i].size())), default_token_parser_impl,
                                                                unsigned(ast_parser::Flags::ParseConstTrueExpr) | unsigned(ast_parser::Flags::ParseMatchSomeExpr)};
                                                        mine.ids.clear();
                                                        exec_query(q, src, noMasked.get(), &mine, nullptr, flags, nullptr);
                                                        mtCounts[i] = mine.ids.size();
                                                }
                                        });
                                for (auto &t : pool)
                                        t.join();
                                clock_gettime(CLOCK_MONOTONIC, &b);
                                const double wall = double(b.tv_sec - a.tv_sec) + double(b.tv_nsec - a.tv_nsec) * 1e-9;
                                printf("{\"cmd\":\"timedmt\",\"queries\":%zu,\"threads\":%u,\"mt_seconds\":%.6f,\"mt_counts_equal\":%s}\n", nrun, threads, wall, mtCounts == counts ? "true" : "false");
                        }
                } else if (cmd == "filter") {
                        is >> docFilter.seed >> docFilter.permille;
                        printf("{\"cmd\":\"filter\",\"seed\":%" PRIu64 ",\"permille\":%u}\n", docFilter.seed, docFilter.permille);
                } else if (cmd == "sim") {
                        is >> simName;
                        collScorer = simName == "tfidf" ? static_cast<Similarity::IndexSourcesCollectionTermsScorer *>(&tfidf)
                                     : simName == "trivial" ? static_cast<Similarity::IndexSourcesCollectionTermsScorer *>(&trivial)
                                                            : static_cast<Similarity::IndexSourcesCollectionTermsScorer *>(&bm25);
                        printf("{\"cmd\":\"sim\",\"name\":\"%s\"}\n", simName.c_str());
                } else if (cmd == "query" || cmd == "queryfull" || cmd == "querysome") {
                        uint32_t flags, k = 0, someMin = 0;
                        is >> flags;
                        if (cmd != "queryfull")
                                is >> k;
                        if (cmd == "querysome")
                                is >> someMin; // threshold for every [a, b, ...] of the query
                        std::string text;
                        std::getline(is, text);
                        while (!text.empty() && text[0] == ' ')
                                text.erase(0, 1);
                        Collector coll;
                        // `a <b>`: ConstTrueExpr needs its parser flag (queries.h:238); default tokens parser
                        query q{str32_t(text.data(), uint32_t(text.size())), default_token_parser_impl,
                                unsigned(ast_parser::Flags::ParseConstTrueExpr) | unsigned(ast_parser::Flags::ParseMatchSomeExpr)};
                        if (someMin)
                                set_matchsome_min(q.root, uint16_t(someMin));
                        std::unique_ptr<Similarity::IndexSourceTermsScorer> scorer;
                        if (flags & unsigned(ExecFlags::AccumulatedScoreScheme)) {
                                collScorer->reset(&collection);
                                scorer.reset(collScorer->new_source_scorer(src));
                        }
                        exec_query(q, src, noMasked.get(), &coll, docFilter.permille ? &docFilter : nullptr, flags, scorer.get());
                        const size_t n = coll.ids.size(), kk = n < 16 ? n : 16;
                        double ssum = 0;
                        for (auto s : coll.scores)
                                ssum += s;
                        printf("{\"cmd\":\"%s\",\"flags\":%u,\"sim\":\"%s\",\"min\":%u,", cmd.c_str(), flags, simName.c_str(), someMin);
                        if (docFilter.permille)
                                printf("\"filter\":[%" PRIu64 ",%u],", docFilter.seed, docFilter.permille);
                        printf("\"q\":\"");
                        for (char c : text) {
                                if (c == '"')
                                        printf("\\\"");
                                else
                                        putchar(c);
                        }
                        printf("\",\"n\":%zu,\"fnv\":\"%" PRIu64 "\",", n, to_fnv1a_docs(coll.ids.data(), n));
                        print_u32s("first", coll.ids.data(), kk);
                        printf(",");
                        print_u32s("last", coll.ids.data() + (n - kk), kk);
                        printf(",\"score_sum\":%.17g", ssum);
                        if (!(flags & 3u)) {
                                printf(",\"rich_fnv\":\"%" PRIu64 "\",\"terms_total\":%" PRIu64 ",\"hits_total\":%" PRIu64 ",\"rich_docs\":[", coll.richFnv, coll.termsTotal,
                                       coll.hitsTotal);
                                for (size_t i = 0; i < coll.richDocs.size(); ++i) {
                                        printf("%s[", i ? "," : "");
                                        for (size_t j = 0; j < coll.richDocs[i].size(); ++j)
                                                printf("%s%u", j ? "," : "", coll.richDocs[i][j]);
                                        printf("]");
                                }
                                printf("]");
                        }
                        if (k && !coll.scores.empty()) {
                                to_result r;
                                r.docs = coll.ids.data();
                                r.scores = coll.scores.data();
                                r.n = r.cap = n;
                                std::vector<uint32_t> td(k);
                                std::vector<float> ts(k);
                                const uint32_t m = to_topk(&r, k, td.data(), ts.data()); // this repo's tie rule
                                printf(",\"top\":[");
                                for (uint32_t i = 0; i < m; ++i)
                                        printf("%s[%u,%.9g]", i ? "," : "", td[i], double(ts[i]));
                                printf("]");
                        }
                        if (cmd == "queryfull") {
                                printf(",");
                                print_u32s("docs", coll.ids.data(), n);
                                if (!coll.scores.empty()) {
                                        printf(",\"scores\":[");
                                        for (size_t i = 0; i < n; ++i)
                                                printf("%s%.17g", i ? "," : "", coll.scores[i]);
                                        printf("]");
                                }
                        }
                        printf("}\n");
                } else {
                        printf("{\"cmd\":\"%s\",\"error\":\"unknown\"}\n", cmd.c_str());
                }
                fflush(stdout);
        }
        if (corpus)
                to_corpus_free(corpus);
        return 0;
}
