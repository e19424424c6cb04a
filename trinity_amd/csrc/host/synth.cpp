// synth.cpp — deterministic synthetic segment + query generator of SURVEY.md §8(d) / BASELINE.md §3, exported
// with a C ABI for bench.py and the tests.  Host-only C++ (no HIP).  New code; independent of oracle/.
//
//   corpus : splitmix64(seed); documents 1..D; `slots` token slots per document at positions 1..slots; every
//            slot draws a term rank r in [0,V) from Zipf(s=1) by inverse CDF (first i with cdf[i] >= x,
//            x = (u >> 11) * 2^-53, cdf built by sequential double sums); term r is the r-th term of the table
//   queries: splitmix64(seed); terms drawn from the same Zipf, distinct within a query
#include "google_encoder.hpp"
#include "lucene_encoder.hpp"
#include <algorithm>
#include <cstdlib>
#include <memory>

using namespace trinity_amd;

namespace {
        inline uint64_t splitmix64(uint64_t &s) {
                uint64_t z = (s += 0x9e3779b97f4a7c15ull);
                z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
                z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
                return z ^ (z >> 31);
        }

        struct Zipf {
                std::vector<double> cdf;
                std::vector<uint32_t> guide; // guide[k] = lower_bound(cdf, k / G): narrows the search, never changes it
                static constexpr uint32_t G = 1u << 16;
                explicit Zipf(uint32_t V)
                    : cdf(V), guide(G + 1) {
                        double s = 0;
                        for (uint32_t i = 0; i < V; ++i) {
                                s += 1.0 / double(i + 1);
                                cdf[i] = s;
                        }
                        for (auto &c : cdf)
                                c /= s;
                        for (uint32_t k = 0; k <= G; ++k) {
                                const auto it = std::lower_bound(cdf.begin(), cdf.end(), double(k) / double(G));
                                guide[k] = std::min<uint32_t>(uint32_t(it - cdf.begin()), V - 1);
                        }
                }
                uint32_t rank(uint64_t u) const {
                        const double x = double(u >> 11) * (1.0 / 9007199254740992.0);
                        const uint32_t k = uint32_t(x * double(G));
                        auto b = cdf.begin() + guide[k];
                        auto e = cdf.begin() + std::min<size_t>(size_t(guide[k + 1]) + 1, cdf.size());
                        while (b != cdf.begin() && *(b - 1) >= x)
                                --b;
                        const auto it = std::lower_bound(b, e, x);
                        return std::min<uint32_t>(uint32_t(it - cdf.begin()), uint32_t(cdf.size() - 1));
                }
        };

        struct Segment {
                std::vector<uint8_t> index, hits; // hits: the LUCENE codec's hits.data
                std::vector<term_index_ctx> terms;
                uint64_t sumTermsDocs{0}, sumTermHits{0};
                uint32_t totalTerms{0}, docsCnt{0};
        };
} // namespace

extern "C" {
static void *segment_build(uint32_t D, uint32_t V, uint32_t slots, uint64_t seed, int codec);
// Build the synthetic GOOGLE-codec segment.  Returns an opaque handle (NULL on failure).
void *tri_synth_segment_build(uint32_t D, uint32_t V, uint32_t slots, uint64_t seed) { return segment_build(D, V, slots, seed, 1); }
// codec: 1 = GOOGLE, 2 = LUCENE-shaped (PFOR128 payload), 3 = LUCENE-shaped with FastPFor<4> payload words (csrc/fastpfor128.hpp)
void *tri_synth_segment_build_codec(uint32_t D, uint32_t V, uint32_t slots, uint64_t seed, int codec) { return segment_build(D, V, slots, seed, codec); }
const uint8_t *tri_synth_segment_hits(void *h, uint64_t *len);

static void *segment_build(uint32_t D, uint32_t V, uint32_t slots, uint64_t seed, int codec) {
        try {
                auto seg = std::make_unique<Segment>();
                const uint64_t ntok = uint64_t(D) * slots;
                Zipf z(V);
                std::vector<uint32_t> ranks(ntok);
                std::vector<uint64_t> off(size_t(V) + 1, 0);
                uint64_t st = seed;
                for (uint64_t i = 0; i < ntok; ++i) {
                        const uint32_t r = z.rank(splitmix64(st));
                        ranks[i] = r;
                        ++off[r + 1];
                }
                for (uint32_t t = 0; t < V; ++t)
                        off[t + 1] += off[t];
                // counting sort of the (term, doc, pos) tokens by term; generation order keeps (doc, pos) ascending
                std::vector<uint32_t> tdoc(ntok);
                std::vector<uint16_t> tpos(ntok);
                {
                        std::vector<uint64_t> cur(off.begin(), off.end() - 1);
                        uint64_t i = 0;
                        for (uint32_t d = 1; d <= D; ++d)
                                for (uint32_t p = 1; p <= slots; ++p, ++i) {
                                        const uint64_t o = cur[ranks[i]]++;
                                        tdoc[o] = d;
                                        tpos[o] = uint16_t(p);
                                }
                }
                std::vector<uint32_t>().swap(ranks);
                seg->terms.resize(V);
                auto feed = [&](auto &enc) {
                        for (uint32_t t = 0; t < V; ++t) {
                                const uint64_t b = off[t], e = off[t + 1];
                                if (b == e)
                                        continue;
                                enc.begin_term();
                                for (uint64_t i = b; i < e;) {
                                        const uint32_t d = tdoc[i];
                                        enc.begin_document(d);
                                        for (; i < e && tdoc[i] == d; ++i)
                                                enc.new_hit(tpos[i]);
                                        enc.end_document();
                                }
                                enc.end_term(&seg->terms[t]);
                                seg->sumTermsDocs += seg->terms[t].documents;
                                ++seg->totalTerms;
                        }
                };
                if (codec == 2 || codec == 3) { // (3: the ints() groups carry FastPFor<4> words, as the reference's own build writes them)
                        Codecs::Lucene::IndexSession sess;
                        Codecs::Lucene::Encoder enc(&sess, codec == 3 ? Codecs::Lucene::Payload::FastPFor : Codecs::Lucene::Payload::PFOR128);
                        feed(enc);
                        seg->index.swap(sess.indexOut);
                        seg->hits.swap(sess.positionsOut);
                } else {
                        Codecs::Google::IndexSession sess;
                        sess.indexOut.reserve(size_t(ntok) * 4);
                        Codecs::Google::Encoder enc(&sess);
                        feed(enc);
                        seg->index.swap(sess.indexOut);
                }
                if (seg->index.size() > 0xffffffffull || seg->hits.size() > 0xffffffffull)
                        return nullptr; // 32-bit chunk offsets (codecs.h:26)
                seg->sumTermHits = ntok;
                seg->docsCnt = D;
                return seg.release();
        } catch (...) {
                return nullptr;
        }
}

void tri_synth_segment_free(void *h) { delete static_cast<Segment *>(h); }
const uint8_t *tri_synth_segment_hits(void *h, uint64_t *len) {
        auto *s = static_cast<Segment *>(h);
        *len = s->hits.size();
        return s->hits.data();
}
const uint8_t *tri_synth_segment_index(void *h, uint64_t *len) {
        auto *s = static_cast<Segment *>(h);
        *len = s->index.size();
        return s->index.data();
}
// term table as {documents, offset, size} u32 triples == term_index_ctx
const uint32_t *tri_synth_segment_terms(void *h, uint32_t *nterms) {
        auto *s = static_cast<Segment *>(h);
        static_assert(sizeof(term_index_ctx) == 12, "term_index_ctx layout");
        *nterms = uint32_t(s->terms.size());
        return reinterpret_cast<const uint32_t *>(s->terms.data());
}
void tri_synth_segment_stats(void *h, uint64_t *sumTermsDocs, uint64_t *sumTermHits, uint32_t *totalTerms, uint32_t *docsCnt) {
        auto *s = static_cast<Segment *>(h);
        *sumTermsDocs = s->sumTermsDocs;
        *sumTermHits = s->sumTermHits;
        *totalTerms = s->totalTerms;
        *docsCnt = s->docsCnt;
}
// nq phrases of nterms terms each: even queries take the terms of nterms consecutive token slots of a random
// document of the corpus (D, slots, corpus_seed) — such a phrase has at least one match —, odd queries take random Zipf terms.
// Token (doc d, slot p) is draw number (d-1)*slots + (p-1) of the corpus stream, and splitmix64 is a counter-based generator,
// so no corpus needs to be materialised.
void tri_synth_phrase_queries(uint32_t D, uint32_t V, uint32_t slots, uint64_t corpus_seed, uint64_t seed, uint32_t nq, uint32_t nterms, uint32_t *out) {
        Zipf z(V);
        uint64_t st = seed;
        if (nterms > slots)
                nterms = slots;
        for (uint32_t q = 0; q < nq; ++q) {
                uint32_t *t = out + size_t(q) * nterms;
                if ((q & 1u) == 0) {
                        const uint32_t d = uint32_t(splitmix64(st) % D);                  // document d + 1
                        const uint32_t p = uint32_t(splitmix64(st) % (slots - nterms + 1)); // first slot
                        for (uint32_t i = 0; i < nterms; ++i) {
                                uint64_t s = corpus_seed + (uint64_t(d) * slots + p + i) * 0x9e3779b97f4a7c15ull; // state before the draw
                                t[i] = z.rank(splitmix64(s));
                        }
                } else
                        for (uint32_t i = 0; i < nterms; ++i)
                                t[i] = z.rank(splitmix64(st));
        }
}
// nq queries x nterms distinct Zipf-sampled term ranks
void tri_synth_queries(uint32_t V, uint64_t seed, uint32_t nq, uint32_t nterms, uint32_t *out) {
        Zipf z(V);
        uint64_t st = seed;
        for (uint32_t q = 0; q < nq; ++q) {
                uint32_t *t = out + size_t(q) * nterms;
                for (uint32_t i = 0; i < nterms;) {
                        const uint32_t r = z.rank(splitmix64(st));
                        if (std::find(t, t + i, r) == t + i)
                                t[i++] = r;
                }
        }
}
// The host encoder (google_encoder.hpp, byte-identical to the reference's) over caller-supplied postings: the checker of the device
// encoder (tri_encode_google, include/trinity_hip.h) in tests.  Same arguments; terms_out rows are {documents, offset, size}.
// Returns the index length, or -1 when `cap` is too small / the input is malformed.
long long tri_host_encode_google_payloads(const uint32_t *docs, const uint32_t *freqs, const uint16_t *positions, const uint8_t *payload_lens,
                                          const uint64_t *payloads, const uint64_t *term_first, uint64_t nterms, uint8_t *out, uint64_t cap, uint32_t *terms_out);
long long tri_host_encode_google(const uint32_t *docs, const uint32_t *freqs, const uint16_t *positions, const uint64_t *term_first, uint64_t nterms,
                                 uint8_t *out, uint64_t cap, uint32_t *terms_out) {
        return tri_host_encode_google_payloads(docs, freqs, positions, nullptr, nullptr, term_first, nterms, out, cap, terms_out);
}
// ... with hit payloads: payload_lens[h] bytes of payloads[h] (first byte in the low 8 bits) per hit; payload_lens == nullptr: none
long long tri_host_encode_google_payloads(const uint32_t *docs, const uint32_t *freqs, const uint16_t *positions, const uint8_t *payload_lens,
                                          const uint64_t *payloads, const uint64_t *term_first, uint64_t nterms, uint8_t *out, uint64_t cap, uint32_t *terms_out) {
        try {
                Codecs::Google::IndexSession sess;
                Codecs::Google::Encoder enc(&sess);
                uint64_t h = 0;
                for (uint64_t t = 0; t < nterms; ++t) {
                        term_index_ctx tctx;
                        enc.begin_term();
                        for (uint64_t p = term_first[t]; p < term_first[t + 1]; ++p) {
                                enc.begin_document(docs[p]);
                                for (uint32_t k = 0; k < freqs[p]; ++k, ++h) {
                                        uint8_t bytes[8];
                                        const uint8_t plen = payload_lens ? payload_lens[h] : 0;
                                        for (uint8_t z = 0; z < plen && z < 8; ++z)
                                                bytes[z] = uint8_t(payloads[h] >> (8 * z));
                                        enc.new_hit(positions[h], bytes, plen);
                                }
                                enc.end_document();
                        }
                        enc.end_term(&tctx);
                        terms_out[3 * t] = tctx.documents;
                        terms_out[3 * t + 1] = tctx.offset;
                        terms_out[3 * t + 2] = tctx.size;
                }
                if (sess.indexOut.size() > cap)
                        return -1;
                std::memcpy(out, sess.indexOut.data(), sess.indexOut.size());
                return (long long)sess.indexOut.size();
        } catch (...) {
                return -1;
        }
}
}
