/*
 * trinity_oracle.h — CPU ORACLE for the Trinity query-execution hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (trinity_amd/) never
 * links, imports or calls anything in this directory.
 *
 * Plain-C restatement of the reference (phaistos-networks/Trinity) algorithms on the
 * hot path: prefix-varint, Google block codec (writer + reader), document-at-a-time
 * iterators (PostingsListIterator / Conjuction / DisjunctionAllPLI / Phrase), the
 * DocsSetSpan drivers, the IteratorScorer wrappers and the BM25 similarity.
 * Every function cites the reference file:line it follows (paths relative to the
 * reference tree).
 *
 * Pinning: the restatement is checked against outputs of the real reference compiled
 * here from its own sources (oracle/_ref, recipe in oracle/Makefile) and against the
 * fixtures those runs produced (tests/golden/, generator tests/golden/make_golden.py).
 * The Lucene/PFOR payload is NOT covered by that pin (FastPFor is an absent, un-vendored
 * submodule of the reference): "parity unpinned" for PFOR payload bytes.
 */
#ifndef TRINITY_ORACLE_H
#define TRINITY_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TO_DOCIDS_END 0xffffffffu /* common.h:43 DocIDsEND */

/* exec.h:11-43 ExecFlags */
#define TO_FLAG_DOCUMENTS_ONLY 1u
#define TO_FLAG_ACCUM_SCORE 2u

/* ---- postfix query program (shared shape with include/trinity_hip.h tri_query) ---- */
#define TO_OP_TERM 0u   /* operand = term id (rank)            */
#define TO_OP_AND 1u    /* operand = number of children (>= 2) */
#define TO_OP_OR 2u     /* operand = number of children (>= 2) */
#define TO_OP_PHRASE 3u /* operand = number of terms; the n preceding tokens must be TERMs */
#define TO_OP_NOT 4u    /* operand = 2: the two preceding sub-programs are (required, excluded); exec.cpp:424-427 logicalnot */
#define TO_OP_OPT 5u    /* operand = 2: (main, optional) — `a <b>`: consttrueexpr under an AND -> DocsSetIterators::Optional (exec.cpp:366-377):
                           the documents of main; optional only adds its score / its matched terms where it matches */
#define TO_OP_SOME 6u   /* operand = (min << 16) | n: at least `min` of the n preceding sub-programs match — matchsome -> DocsSetIterators::DisjunctionSome
                           (exec.cpp:276-283; docset_iterators.cpp:679-811).  Oracle only so far: the GPU planner does not lower it yet */
#define TO_TOK(op, arg) (((uint32_t)(op) << 28) | ((uint32_t)(arg)&0x0fffffffu))
#define TO_TOK_OP(t) ((t) >> 28)
#define TO_TOK_ARG(t) ((t)&0x0fffffffu)

/* ------------------------------------------------------------------ a1: prefix varint */
/* Switch/switch_compiler_aux.h:23-51 (put) and :53-80 (get). Return bytes written/consumed. */
size_t to_varbyte_put32(uint8_t *out, uint32_t v);
size_t to_varbyte_get32(const uint8_t *in, uint32_t *v);

/* ------------------------------------------------------------------ synthetic corpus  */
/* Deterministic corpus of SURVEY.md §8(d): splitmix64(seed); docs 1..D, `slots` token slots at
 * positions 1..slots, each a Zipf(s=1.0) rank in [0,V) by inverse CDF (first i with cdf[i] >= x). */
typedef struct to_corpus {
        uint32_t D, V, slots;
        uint64_t ntokens;   /* D*slots */
        uint64_t *term_off; /* [V+1] token range of each term (tokens sorted by term, doc, pos) */
        uint32_t *tok_doc;  /* [ntokens] */
        uint16_t *tok_pos;  /* [ntokens] */
} to_corpus;

to_corpus *to_corpus_generate(uint32_t D, uint32_t V, uint32_t slots, uint64_t seed);
void to_corpus_free(to_corpus *);
/* Zipf sampler shared with the query generator */
typedef struct to_zipf to_zipf;
to_zipf *to_zipf_new(uint32_t V);
void to_zipf_free(to_zipf *);
uint32_t to_zipf_rank(const to_zipf *, uint64_t u64);
uint64_t to_splitmix64(uint64_t *state);
/* nq queries of nterms distinct Zipf ranks each, seed as given (1337) */
void to_gen_queries(uint32_t V, uint64_t seed, uint32_t nq, uint32_t nterms, uint32_t *out_terms);

/* ------------------------------------------------------------------ index (Google codec) */
typedef struct to_term { /* codecs.h:17-55 term_index_ctx {documents, indexChunk{offset,len}} */
        uint32_t documents, offset, size;
} to_term;

typedef struct to_index {
        uint8_t *bytes; /* concatenated term chunks == the segment's `index` file */
        size_t len;
        to_term *terms;
        uint32_t nterms;
        /* index_source.h:44-53 field_statistics */
        uint64_t sumTermHits;
        uint32_t totalTerms;
        uint64_t sumTermsDocs;
        uint32_t docsCnt;
        int owns;
        /* masked documents (docidupdates.h:90-119 masked_documents_registry::test): documents updated or deleted by a newer
         * segment; exec_query drops them right before consider() (exec.cpp:914-975).  Sorted ascending; NULL/0 = none.
         * The reference's registry (banks + bloom filter) cannot be built here (docidupdates.cpp needs boost spreadsort):
         * only its observable behaviour — set membership — is restated; "parity unpinned" for the registry itself. */
        uint32_t *masked;
        size_t nmasked;
        int codec;     /* TO_CODEC_GOOGLE (default, 0 also means Google) or TO_CODEC_LUCENE */
        uint8_t *hits; /* Lucene: the segment's hits.data (lucene_codec.h:206) */
        size_t hits_len;
        int similarity; /* TO_SIM_*: which Similarity::IndexSourcesCollection*Scorer AccumulatedScoreScheme queries use */
} to_index;

/* similarity.h: BM25 :165-255 (default), TF-IDF :75-163, Trivial :56-72 */
#define TO_SIM_BM25 0
#define TO_SIM_TFIDF 1
#define TO_SIM_TRIVIAL 2

#define TO_CODEC_GOOGLE 1
#define TO_CODEC_LUCENE 2

/* ------------------------------------------------------------------ index (Lucene-shaped codec)
 * Container exactly as lucene_codec.cpp:163-388 writes it (term header, 128-document blocks as two ints() groups, varbyte
 * tail, 22-byte skiplist entries, hits.data framing).  The ints() payload is THIS REPO'S PFOR128 (include/pfor128.md):
 * the reference delegates it to lemire/FastPFor, an absent un-vendored submodule — "parity unpinned" for those bytes. */
to_index *to_lucene_encode(const to_corpus *);
/* ints() group of 128 values (lucene_codec.cpp:26-66 / 69-100): returns bytes written / consumed */
size_t to_ints_encode(const uint32_t *values, uint8_t *out);
size_t to_ints_decode(const uint8_t *in, uint32_t *values);

/* google_codec.cpp:9-176 writer, terms encoded in rank order, hits = positions, no payloads */
to_index *to_google_encode(const to_corpus *);
/* wrap externally produced bytes (e.g. the product-side builder's or the reference's) */
to_index *to_index_wrap(const uint8_t *bytes, size_t len, const to_term *terms, uint32_t nterms, uint32_t docsCnt,
                        uint64_t sumTermsDocs, uint64_t sumTermHits);
void to_index_free(to_index *);
/* replace the index's masked-document set (copied; any order, duplicates allowed) */
void to_index_set_masked(to_index *, const uint32_t *docids, size_t n);

/* Walk one chunk's block headers (format check, algorithmic-byte accounting):
 * returns number of blocks; fills header / delta+freq / hit / skiplist byte counts and #postings. */
uint32_t to_google_chunk_stats(const to_index *, uint32_t term, uint64_t *hdr, uint64_t *docfreq, uint64_t *hits,
                               uint64_t *skip, uint64_t *postings);

/* ------------------------------------------------------------------ postings iterator   */
typedef struct to_pli to_pli;
to_pli *to_pli_new(const to_index *, uint32_t term); /* google_codec.cpp:936-990 + 442-462 */
void to_pli_free(to_pli *);
uint32_t to_pli_next(to_pli *);                    /* google_codec.cpp:777-819 */
uint32_t to_pli_advance(to_pli *, uint32_t target); /* google_codec.cpp:821-934 */
uint32_t to_pli_current(const to_pli *);
uint32_t to_pli_freq(const to_pli *); /* codecs.h:217: exposed as tokenpos_t (u16) */
/* google_codec.cpp:533-594; returns number of hits written; out_pos needs room for freq entries */
uint32_t to_pli_materialize_positions(to_pli *, uint16_t *out_pos);
/* ... with term_hit::payloadLen / ::payload as the reference leaves them (google_codec.cpp:533-594) */
uint32_t to_pli_materialize_hits(to_pli *it, uint16_t *pos_out, uint8_t *plen_out, uint64_t *payload_out);

/* decode a whole term: docs/freqs arrays (capacity documents); returns count */
uint32_t to_decode_term(const to_index *, uint32_t term, uint32_t *docs, uint32_t *freqs);

/* ------------------------------------------------------------------ similarity          */
/* similarity.h:179-181 (float-precision log, returned as double) */
double to_bm25_idf(uint32_t docFreq, uint64_t docsCnt);
/* similarity.h:228-235 */
float to_bm25_score(double idf, uint16_t freq);
/* the ScorerWeight contribution of one term (new_scorer_weight sums it over a phrase's terms) and score(id, freq, weight)
 * of the three scorers: BM25 as above; TF-IDF similarity.h:85-87 idf = log((docsCnt+1)/(double)(df+1)) + 1, :92-94
 * tf = sqrt(float freq), :133-138 score = tf * weight; Trivial :64-66 score = freq, no weight */
double to_sim_weight(int sim, uint32_t docFreq, uint64_t docsCnt);
float to_sim_score(int sim, double weight, uint16_t freq);

/* ------------------------------------------------------------------ query execution     */
typedef struct to_result {
        uint32_t *docs;
        double *scores; /* NULL in DocumentsOnly mode */
        size_t n, cap;
} to_result;

/* exec.cpp:509-1517 for one index source, no masked documents, no filter.
 * flags: TO_FLAG_DOCUMENTS_ONLY or TO_FLAG_ACCUM_SCORE (BM25).  Returns 0 on success. */
int to_exec_query(const to_index *, const uint32_t *prog, uint32_t proglen, uint32_t flags, to_result *out);
void to_result_free(to_result *);

/* ---- the default ("rich match") execution mode: exec_query without DocumentsOnly / AccumulatedScoreScheme (exec.cpp:1350-1501).
 * Every match is handed to consider(const matched_document &) with the query terms that matched it and their hits
 * (queryexec_ctx.cpp:382-648 collect_doc_matching_terms + prepare_match; matches.h:109-130).  Restated in a canonical form:
 * per match, in docID order, the u32 words  doc, nterms, then per matched term — ascending term rank — rank, freq, pos[freq].
 * (The reference collects the terms in tree / heap order; an application sees a set.) */
typedef struct to_rich {
        uint32_t *docs;
        size_t n;
        uint32_t *flat; /* the canonical stream */
        size_t nflat, capflat;
        uint64_t terms_total, hits_total;
} to_rich;
int to_exec_query_rich(const to_index *, const uint32_t *prog, uint32_t proglen, to_rich *out);
void to_rich_free(to_rich *);

/* Application-side top-K over a result (Trinity ships none: matches.h:139-185; the build defines the
 * tie rule: score descending, docID ascending).  Returns min(n,k). */
uint32_t to_topk(const to_result *, uint32_t k, uint32_t *docs, float *scores);

/* bench.py cpu_baseline, all-cores leg: one query per thread from a shared cursor until the batch or the budget runs out */
uint64_t to_exec_batch_mt(const to_index *, const uint32_t *progs, uint32_t proglen, uint32_t nq, uint32_t flags, uint32_t nthreads,
                          double budget_seconds, uint64_t *out_matches, double *out_seconds);

/* FNV-1a (64) over the little-endian docID stream — the fixture hash of SURVEY §8(c). */
uint64_t to_fnv1a_docs(const uint32_t *docs, size_t n);

#ifdef __cplusplus
}
#endif
#endif
