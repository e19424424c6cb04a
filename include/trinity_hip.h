/*
 * trinity_hip.h — C-ABI of libtrinity_hip.so, the MI355X (gfx950) execution engine for Trinity's
 * query hot path: postings decode -> docset intersection/union -> per-document BM25 -> top-K.
 *
 * Plain C: opaque handles, plain pointers and sizes, int status codes.  Nothing throws across this
 * boundary.  Every entry point names the reference interface (file:line under the reference tree)
 * whose work it takes over; INTEGRATION.md shows the binding a Trinity maintainer would add.
 *
 * Threading: one tri_dev per (host thread, device) — with ONE exception, made for a caller that keeps the engine stream fed: while one
 * thread runs, awaits, reads and destroys batches of a device (tri_batch_run / _sync / the result calls / _destroy), other threads may
 * compile the next ones (tri_batch_create) on the same device; the handle's pools, the index's plane cache and everything that enqueues on
 * its streams are locked for that.  The host planner — most of a create — runs outside that lock, in one of the handle's TWO planner
 * contexts (a pool of host threads each): two creates plan side by side, a third waits for a context.  Everything else (uploads, options,
 * the write side) stays one thread at a time.  tri_last_error() is per thread.
 */
#ifndef TRINITY_HIP_H
#define TRINITY_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRI_ABI_VERSION 9 /* 2: tri_batch_info grew (fused_*), tri_dev_set_option / tri_dev_get_option; 3: TRI_OP_SOME, tri_batch_info.cand_needed_bytes / phrase_*;
                             4: tri_batch_info grew (term planes, k_planes), tri_batch_query_status, tri_comm_create_custom; 5: tri_encode_google_payloads;
                             6: tri_batch_info.create_ms / create_plan_ms / *_bound_bytes, options plane_max_bytes / plan_threads, tri_cbatch_query_status;
                             7: TASK_TREE (any query tree), tri_batch_info.tree_ms / tree_queries / tree_scratch_bytes, tri_commit_* / tri_merge_google;
                             8: tri_batch_docsets (every query's docID set in one call), tri_merge_lucene, option planes_rebuild;
                             9: tri_dev_memory (HBM in use), tri_batch_docsets_mixed (dense sets delivered as bitmap words), two planner contexts per handle (two threads may compile at once), plane rows built by need */

/* status codes */
#define TRI_OK 0
#define TRI_ERR_INVALID (-1)     /* bad argument / malformed program   (reference: Switch::invalid_argument) */
#define TRI_ERR_DEVICE (-2)      /* HIP runtime error                  */
#define TRI_ERR_UNSUPPORTED (-3) /* query shape not lowered yet        */
#define TRI_ERR_FORMAT (-4)      /* index bytes fail validation        (reference: Switch::data_error)      */
#define TRI_ERR_NOMEM (-5)
#define TRI_ERR_INTERNAL (-6)    /* the engine contradicts itself (a result block that does not add up): a bug, never a caller's error */

/* codecs (segment `id` file codec name: indexer.cpp:266-270, segment_index_source.cpp:172-179) */
#define TRI_CODEC_GOOGLE 1
#define TRI_CODEC_LUCENE 2

/* exec.h:11-43 ExecFlags (same values) */
#define TRI_FLAG_DOCUMENTS_ONLY 1u
#define TRI_FLAG_ACCUMULATED_SCORE 2u
#define TRI_FLAG_HIT_PAYLOADS 8u  /* with TRI_FLAG_MATCHED_TERMS: every hit also comes with term_hit::payloadLen and ::payload (runtime.h:8-20) as
                                     Google::Decoder::materialize_hits leaves them (google_codec.cpp:533-594): tri_batch_matched_payloads */
#define TRI_FLAG_MATCHED_TERMS 4u /* exec_query's DEFAULT mode (no ExecFlags, exec.cpp:1350-1501): every match comes with the query
                                     terms that matched it and their hits — what consider(const matched_document &) receives
                                     (matches.h:109-130; queryexec_ctx.cpp:382-648).  Results: tri_batch_matched_terms */

/* similarity.h scorers: Trivial :56-72, TF-IDF :75-163, BM25 :165-255 */
#define TRI_SIM_BM25 0
#define TRI_SIM_TFIDF 1
#define TRI_SIM_TRIVIAL 2

/* postfix query program tokens: (op << 28) | arg.  The host-side planner (the mirror of
 * queryexec_ctx::build_iterator, exec.cpp:253-449) emits these from its iterator tree. */
#define TRI_OP_TERM 0u   /* arg = index into the uploaded term table                              */
#define TRI_OP_AND 1u    /* arg = #children   (DocsSetIterators::Conjuction[AllPLI])              */
#define TRI_OP_OR 2u     /* arg = #children   (DocsSetIterators::Disjunction[AllPLI] / union span) */
#define TRI_OP_PHRASE 3u /* arg = #terms, the preceding arg tokens are TERMs (DocsSetIterators::Phrase) */
#define TRI_OP_NOT 4u    /* operand = 2: the two preceding sub-programs are (required, excluded) — exec.cpp:424-427 logicalnot -> DocsSetIterators::Filter.
                            Lowered: NOT at the root or under AND, excluded side = a term or an OR of terms */
#define TRI_OP_OPT 5u    /* operand = 2: (main, optional) — consttrueexpr under an AND (`a <b>`) -> DocsSetIterators::Optional, exec.cpp:366-377: the
                            documents of main; the optional side (a term or an OR of terms) only adds its score / its matched terms */
#define TRI_OP_SOME 6u   /* operand = (min << 16) | #children — matchsome `[a, b, ...]` with a threshold -> DocsSetIterators::DisjunctionSome,
                            exec.cpp:276-283, docset_iterators.cpp:679-811: the documents at least `min` of the children match */
#define TRI_TOK(op, arg) (((uint32_t)(op) << 28) | ((uint32_t)(arg)&0x0fffffffu))

typedef struct tri_dev tri_dev;
typedef struct tri_index tri_index;
typedef struct tri_batch tri_batch;

/* == Trinity::term_index_ctx {documents, indexChunk{offset,len}} (codecs.h:17-55) */
typedef struct tri_term {
        uint32_t documents;
        uint32_t offset;
        uint32_t size;
} tri_term;

/* one query = a slice of the shared postfix program */
typedef struct tri_query {
        uint32_t prog_off;
        uint32_t prog_len;
} tri_query;

typedef struct tri_index_info {
        uint64_t index_bytes;    /* raw `index` bytes resident in HBM                                  */
        uint64_t directory_bytes; /* per-block directory built at upload                                */
        uint64_t blocks;
        uint64_t postings;       /* sum of documents over terms (field_statistics::sumTermsDocs)       */
        uint64_t doc_bytes;      /* header + delta + freq bytes of every chunk            */
        uint64_t hit_bytes;
        uint32_t nterms;
        uint32_t docs_cnt;
} tri_index_info;

typedef struct tri_batch_info {
        uint64_t nqueries;
        uint64_t algorithmic_bytes; /* sum_q [ sum_t docbytes(t) + out(q) ] of the LAST run */
        uint64_t matches;           /* total matches of the last run                                    */
        uint64_t out_capacity;      /* docID slots reserved for docsets                                 */
        float last_run_ms;          /* HIP-event time of the last tri_batch_run (kernels only)          */
        uint32_t launches;          /* kernel launches per run                                          */
        /* per matching kernel (HIP events on the engine stream around each launch; algorithmic bytes of the queries
         * each one processes): k_and_dense = bitmap windows, k_and = candidate tiles */
        float dense_ms, cand_ms;
        uint64_t dense_algorithmic_bytes, cand_algorithmic_bytes;
        uint64_t dense_queries, cand_queries;
        /* k_fused: AccumulatedScore top-K of dense queries in one pass (decode -> match -> score -> select per docID window) */
        float fused_ms, rest_ms; /* rest_ms: everything after the matching kernels and k_phrase (k_rich, k_score, k_topk_merge) */
        uint64_t fused_algorithmic_bytes;
        uint64_t fused_queries;
        /* k_and skips: its algorithmic bytes are no bound.  With the option account_needed_bytes = 1 at batch creation: the bytes a perfect
         * gallop must read for the candidate-tile queries — the lead lists, of every other list the blocks that can hold a lead candidate
         * (per lead block: the blocks its docID range meets, at most one per candidate; docbytes / nblocks each), + 4 B per match.  0: not asked */
        uint64_t cand_needed_bytes;
        /* k_phrase (positional constraints over the match segments): its time, the hit bytes of the phrases' terms (their SURVEY §8(d) share of
         * algorithmic_bytes — no longer counted under cand_algorithmic_bytes), the queries that hold a phrase; rest_ms no longer includes it */
        float phrase_ms, tree_ms; /* tree_ms (ABI 7): the TASK_TREE kernels (k_tree.hpp) — until then part of rest_ms */
        uint64_t phrase_algorithmic_bytes;
        uint64_t phrase_queries;
        /* term planes: the head terms the batch's queries share are decoded ONCE per launch (k_term_planes, first kernel of the run) into
         * two bitmaps over the docID space each (A: holds the term; B: frequency != 1), which k_and probes, k_and_dense ORs into its windows
         * and k_planes — AccumulatedScore top-K of CNF queries, predicates evaluated 32 documents per word — reads instead of decoding the
         * list again for every query that names it.  plane_terms: how many; plane_bytes: their scratch; term_planes_decoded_bytes: the
         * docbytes of those lists (read once per launch).  dense_ms no longer includes the launch's first memset: term_planes_ms does */
        float term_planes_ms, planes_ms;
        uint64_t planes_algorithmic_bytes; /* SURVEY §8(d) bytes of the queries k_planes runs (per query, as if every list were read) */
        uint64_t planes_queries;
        uint64_t plane_terms, plane_bytes, term_planes_decoded_bytes;
        uint64_t unsupported_queries; /* queries the planner left out of the batch (tri_batch_query_status) */
        /* what tri_batch_create itself took: all of it (host planning on the handle's host threads + arena / pool bookkeeping + enqueueing the plan's
         * one copy on the upload stream), and the host planner's share */
        float create_ms, create_plan_ms;
        /* With the option account_needed_bytes = 1 at creation (else 0): the BATCH-LEVEL bound — a batch that shares decodes must read every
         * DISTINCT list its queries name once (doc bytes; hit bytes of the distinct phrase / reported terms) and write every output once
         * (4 B per match, or 8 B x min(matches, K) when scored): bound_bytes for the whole batch, *_bound_bytes for the queries each kernel
         * runs (distinct lists of THOSE queries + their output; k_phrase: the distinct phrase terms' hit bytes).  SURVEY §8(d)'s per-query
         * count (algorithmic_bytes) charges a list once per query that names it, which no kernel that shares decodes reads */
        uint64_t bound_bytes, dense_bound_bytes, cand_bound_bytes, fused_bound_bytes, planes_bound_bytes, phrase_bound_bytes;
        /* k_psets: the bitmap-window queries ALL of whose terms have a term plane (word-wise algebra over the planes, then the expansion into
         * docIDs) — until ABI 6 part of k_and_dense's figures: its time, queries, SURVEY §8(d) bytes and batch-level bound */
        float pset_ms;
        /* k_probe: one short lead list against lists that all have a term plane (a wave decodes the lead's blocks into registers and tests every
         * document with one bit probe per list) — until ABI 6 part of k_and's figures */
        float probe_ms;
        uint64_t pset_queries, pset_algorithmic_bytes, pset_bound_bytes;
        uint64_t probe_queries, probe_algorithmic_bytes, probe_bound_bytes;
        /* TASK_TREE (ABI 7): the queries evaluated as set algebra over one bitmap per leaf (k_tree.hpp) — a multi-word phrase under OR / NOT /
         * matchsome / <optional>, trees over more distinct terms than the truth-table kernel holds, CNFs of more than 16 terms; the bitmap
         * scratch their leaves and match sets take (option tree_max_bytes bounds it: TRI_ERR_NOMEM, split the batch) */
        uint64_t tree_queries, tree_scratch_bytes;
        /* queries whose docID set the engine holds as a bitmap (ABI 7; DocumentsOnly unions / conjunctions of head terms expected to match one
         * document in 32 or more — option result_bitmaps, default 1): tri_batch_docset expands them on read-back, tri_batch_docset_bitmap hands
         * the words over; match counts, hashes and everything else read the same */
        uint64_t bitmap_queries;
} tri_batch_info;

const char *tri_last_error(void);
int tri_abi_version(void);

/* ---- device --------------------------------------------------------------------------------------- */
int tri_dev_open(int device, tri_dev **out);
void tri_dev_close(tri_dev *);
int tri_dev_sync(tri_dev *);
/* the engine's HIP stream (hipStream_t as void*), for callers that order their own work against it */
void *tri_dev_stream(tri_dev *);
/* Planner / launch options of this device handle; they apply to batches created afterwards.  The reference has no counterpart (its
 * planner's constants are compiled in, exec.cpp:35-110); these are the thresholds of the GPU planner:
 *   "dense_min_postings"  a multi-term query runs as bitmap / score windows when its lists hold at least this many postings (default 524288;
 *                         0 = every query whose lists allow it)
 *   "dense_task_cost"     postings per bitmap-window task (default 196608)
 *   "fused"               1 (default): AccumulatedScore top-K batches run their dense queries through the one-pass kernel; 0: match, then score
 *   "fused_task_cost"     postings per one-pass task (default 0: sized from the batch — 256 K .. 8 M, about two tasks per resident workgroup)
 *   "fused_freq_cap"      0 (default): a window field saturates at the largest freq its width holds; else at this freq (documents above it are
 *                         rescored from the postings — same results, slower)
 *   "fused_halfwords"     1 (default): one-pass queries of <= 5 distinct terms keep 16 bits per document (windows twice as long); 0: 32
 *   "overlap_dense_wgs" / "overlap_cand_wgs"  both non-zero: the two matching kernels run side by side with that many workgroups per CU
 *   "planes"              term planes, a bit set (default 7): 1 the candidate-tile kernel probes them, 2 the bitmap-window kernel ORs them into its
 *                         windows, 4 AccumulatedScore top-K CNF queries run over bit planes (k_planes); 0: every query decodes every list it names
 *   "planes_split"        a query that runs as bit planes (k_planes) is cut into this many docID ranges, one task each (default 0: two, or three in a batch that brings few tasks per workgroup; 65536 and up: cut by postings like k_fused's); the
 *                         ranges share the query's threshold, results do not depend on the cut
 *   "plane_div"           a term gets a plane when it holds at least docs_cnt / plane_div documents (default 1024) and the batch's uses repay one
 *                         decode of its list
 *   "account_needed_bytes" 1: tri_batch_create also works out tri_batch_info.cand_needed_bytes (a directory walk per candidate-tile query; default 0)
 *   "plane_max_bytes"     scratch budget of a batch's term planes (default 8 GiB): the terms eligible for a plane are the longest lists that fit
 *   "planes_rebuild"      1: every tri_batch_run decodes the plane rows its batch names again — a COLD plane cache, what a query stream whose head
 *                         terms were all just evicted pays per batch (default 0: a row is built once per index); a measurement switch, read by tri_batch_run
 *   "cand_xcd"            1 (default): the candidate-tile kernel's tasks are queued per XCD by the plane row they probe — a row's probes land in ONE 4 MB L2
 *                         (csrc/planner.hpp, "k_and's queues"); 0: the heaviest-first order dealt round the queues.  Results do not depend on it
 *   "probe_max_blocks"    > 0: a conjunction of ONE lead list of at most this many blocks with lists that all have planes runs in k_probe (a wave per
 *                         task, csrc/k_probe.hpp) instead of candidate tiles (default 0: off — measured slower at cfg2, planner.hpp)
 *   "overlap"             1: the candidate-tile kernel runs on a second stream beside the window kernels (default 0; measured: no gain, the persistent
 *                         grids do not interleave)
 *   "plan_threads"        host threads tri_batch_create plans large batches with (default 0: up to 16, by the host's cores; 1: the calling thread
 *                         only).  The threads belong to the handle, are pinned to distinct CPUs next to the creating thread's, and keep polling for
 *                         about 3 ms after a batch before they sleep (csrc/host_pool.hpp says why); read when the first large batch is created
 * ("fused" also takes 2: only pure unions run in one pass.)  The options are read when a batch is CREATED, except the two overlap_* ones and planes_rebuild,
 * which tri_batch_run reads (they change how existing batches are launched).  Unknown names fail with TRI_ERR_INVALID. */
int tri_dev_set_option(tri_dev *, const char *name, uint64_t value);
int tri_dev_get_option(tri_dev *, const char *name, uint64_t *value);

/* What the handle holds of the device's memory (no reference counterpart: Trinity mmaps its segments and mallocs per query — queryexec_ctx.cpp:187-249).
 * pool_*: the handle's buffer pool — batches' arenas, output regions, score streams, the indexes' plane caches; in use / idle (idle buffers go back to
 * the device beyond 64 GiB, or when an allocation fails).  device_*: hipMemGetInfo — everything on the device, other processes' memory included. */
typedef struct {
        uint64_t pool_in_use_bytes, pool_idle_bytes, pinned_idle_bytes;
        uint64_t device_free_bytes, device_total_bytes;
} tri_dev_memory_info;
int tri_dev_memory(tri_dev *, tri_dev_memory_info *out);

/* ---- index upload ---------------------------------------------------------------------------------
 * Replaces SegmentIndexSource's mmap of `index` (segment_index_source.cpp:84-93) + per-query
 * Codecs::Google::Decoder::init (google_codec.cpp:936-983): copies the segment's raw codec bytes to HBM
 * once and materialises a dense per-block directory {payload offset, last docID} (the analogue of the
 * decoder materialising its skiplist) plus a device term table.  `terms[i]` is what
 * IndexSource::resolve_term_ctx (index_source.h:103) returns for term i.
 * `hits`/`hits_len` are the Lucene codec's hits.data (lucene_codec.h:206); NULL/0 for GOOGLE. */
int tri_index_upload(tri_dev *, const uint8_t *index, size_t len, const uint8_t *hits, size_t hits_len, int codec,
                     const tri_term *terms, size_t nterms, uint32_t docs_cnt, tri_index **out);
void tri_index_destroy(tri_index *);
int tri_index_get_info(const tri_index *, tri_index_info *);
/* algorithmic doc bytes (SURVEY §8d docbytes(t)) of the given terms */
int tri_index_term_docbytes(const tri_index *, const uint32_t *terms, size_t n, uint64_t *out);

/* Masked documents of this segment: documents updated or deleted by newer segments, which exec_query drops right before
 * consider() — masked_documents_registry::test (docidupdates.h:90-119; exec.cpp:914-975, 1000-1030).  Replaces the set
 * (any order, duplicates allowed; n == 0 clears it).  Kept on the device as a bitmap over docIDs and applied inside the
 * matching kernels, so match counts, docsets, scores and top-K of batches run afterwards never contain a masked document.
 * Batches created before the call keep working; the set they see is the one in place when they RUN. */
int tri_index_set_masked(tri_index *, const uint32_t *docids, size_t n);

/* ---- postings decode (codec seam) -----------------------------------------------------------------
 * Replaces Codecs::PostingsListIterator::next() driven to exhaustion (google_codec.cpp:777-819,
 * unpack_block :596-639): decodes whole postings lists on the GPU.  out_offsets[n+1] receives the prefix
 * offsets into docs/freqs (host buffers with room for sum(documents)). */
int tri_decode_terms(tri_index *, const uint32_t *terms, size_t n, uint32_t *docs, uint32_t *freqs, uint64_t *out_offsets);

/* ---- batched query execution (span seam) -----------------------------------------------------------
 * Replaces, for a whole batch of queries at once: queryexec_ctx::build_iterator + build_span
 * (exec.cpp:253-505), DocsSetSpan::process(mp, 1, DocIDsEND) (docset_spans.cpp:98-173, 269-290, 681-790)
 * over Conjuction/Disjunction/Phrase iterators (docset_iterators.cpp:66-405), the IteratorScorer wrappers
 * (docset_iterators_scorers.cpp) with Similarity BM25 (similarity.h:165-255), and the application's
 * top-K MatchedIndexDocumentsFilter::consider(id, score) heap (matches.h:155-171).
 *
 * prog/queries: postfix programs.  flags — exactly one of: TRI_FLAG_DOCUMENTS_ONLY (results = ascending docID sets),
 * TRI_FLAG_ACCUMULATED_SCORE (topk >= 1: results = top-K by score desc, docID asc + total match counts;
 * topk == 0: every match's score is kept, see tri_batch_scores), TRI_FLAG_MATCHED_TERMS (exec_query's default mode:
 * ascending docID sets + per match the matched query terms with their hits, see tri_batch_matched_terms).
 * similarity: TRI_SIM_BM25 / TRI_SIM_TFIDF / TRI_SIM_TRIVIAL — which IndexSourceTermsScorer::score() the device evaluates.
 * weights: optional, one double per program token (TERM tokens: the term's ScorerWeight, PHRASE tokens:
 * the phrase's); NULL => BM25 idf computed from the index's own statistics exactly as
 * IndexSourcesCollectionBM25Scorer does for a single source (similarity.h:179-181, 202-226). */
int tri_batch_create(tri_index *, const uint32_t *prog, size_t prog_len, const tri_query *queries, size_t nq,
                     const double *weights, uint32_t flags, uint32_t topk, int similarity, tri_batch **out);
/* A query whose shape the planner does not lower (today: a multi-word phrase under an OR or inside a general tree, a general tree over
 * more than 8 distinct terms or 16 scored leaves, more than 16 term slots in a CNF, more than 16 reportable terms in the default mode)
 * does NOT fail tri_batch_create: the query is left out of the batch — it reports no matches — and its status says so, so that one such
 * query among thousands costs the caller one CPU span (exec.cpp:509-1517 for that query alone), not the batch.  status[q]: TRI_OK or
 * TRI_ERR_UNSUPPORTED; tri_batch_info.unsupported_queries counts them; tri_last_error() after tri_batch_create describes the last one.
 * (A malformed program is the caller's bug and still fails the call with TRI_ERR_INVALID.) */
int tri_batch_query_status(const tri_batch *, int32_t *status /* [nq] */);
void tri_batch_destroy(tri_batch *);
/* enqueue the batch on the engine stream (asynchronous) */
int tri_batch_run(tri_batch *);
/* wait for THIS batch's completion (its own last event: batches launched behind it on the engine stream are not waited for, so a caller
 * may keep the next batch queued while it collects this one's results); also refreshes tri_batch_info */
int tri_batch_sync(tri_batch *);
int tri_batch_get_info(const tri_batch *, tri_batch_info *);

/* results (call after tri_batch_sync) */
int tri_batch_match_counts(tri_batch *, uint64_t *counts /* [nq] */);
/* Copy query q's ascending docID set; what MatchedIndexDocumentsFilter::consider(ids, cnt) (matches.h:161-165) receives.
 * DocumentsOnly, MatchedTerms and AccumulatedScore batches with topk == 0 materialise every query's set.  An AccumulatedScore batch
 * with topk >= 1 delivers top-K lists and match counts; queries it ran through the one-pass kernel have no materialised set and
 * the call fails with TRI_ERR_INVALID for them (tri_batch_docset_hashes likewise when the batch holds such a query). */
int tri_batch_docset(tri_batch *, size_t q, uint32_t *out, size_t cap, size_t *n);
/* The docID set of query q in the form the engine holds it.  *form = 0: ascending docIDs (tri_batch_docset copies them), nothing else is set.
 * *form = 1: a BITMAP — a DocumentsOnly union / conjunction of head terms expected to match one document in 32 or more keeps one bit per
 * document instead of four bytes per match (option result_bitmaps, default 1; tri_batch_info.bitmap_queries): bit j of words[i] set <=> document
 * *first_doc + 32 i + j matches; *nwords words (words == NULL: sizes only).  tri_batch_docset expands such a set on read-back; a caller that
 * replays consider(ids, cnt) (matches.h:161-165) or intersects further can take the words as they are. */
int tri_batch_docset_bitmap(tri_batch *, size_t q, int *form, uint32_t *words, size_t cap, uint32_t *first_doc, size_t *nwords);
/* EVERY query's docID set in one call — the batch form of the delivery above: out[offsets[q] .. offsets[q + 1]) = query q's ascending docIDs
 * (queries in the caller's order, offsets[nq] = the total; out == NULL: offsets only).  The sets are gathered on the device into one contiguous
 * buffer (task segments in order, bitmap-form results expanded) and cross to the host in ONE copy: with `out` in pinned memory that is PCIe's rate,
 * where tri_batch_docset pays a copy and a synchronisation per task segment.  What a caller that replays
 * MatchedIndexDocumentsFilter::consider(const docid_t *, size_t) (matches.h:161-165; exec.cpp:1213-1229 hands every match to consider()) loops
 * over.  Fails like tri_batch_docset for a query of an AccumulatedScore top-K batch whose set was never materialised. */
int tri_batch_docsets(tri_batch *, uint32_t *out, size_t cap, uint64_t *offsets /* [nq + 1] */);
/* ... each set in the form the engine holds it (the batch form of tri_batch_docset / tri_batch_docset_bitmap): forms[q] = 0: out[offsets[q] .. offsets[q + 1]) = query q's
 * ascending docIDs; 1 (a DocumentsOnly union / conjunction of head terms expected to match one document in 32 or more — option result_bitmaps): the words of a bitmap
 * over the query's docID range, bit j of word i = document 32 i + j matches (tri_batch_match_counts gives its matches).  A dense set crosses PCIe as one bit per document
 * instead of four bytes per match; the caller expands it into the ids of consider(const docid_t *, size_t) (matches.h:161-165), or keeps the bitmap.  Same calling
 * convention as tri_batch_docsets (out == NULL: offsets and forms only).  Both calls wait outside the handle's lock: another thread may compile or run meanwhile. */
int tri_batch_docsets_mixed(tri_batch *, uint32_t *out, size_t cap /* words */, uint64_t *offsets /* [nq + 1] */, uint32_t *forms /* [nq] */);
/* AccumulatedScore with topk == 0: the score of every match of query q, parallel to tri_batch_docset(q) — the
 * (id, score) stream MatchedIndexDocumentsFilter::consider(id, score) receives (matches.h:169; exec.cpp:1322-1341) */
int tri_batch_scores(tri_batch *, size_t q, double *out, size_t cap, size_t *n);
/* TRI_FLAG_MATCHED_TERMS batches.  tri_batch_query_terms: the query's reportable terms (every TERM / PHRASE-member of the
 * program outside the excluded side of a NOT, distinct, in order of first appearance; at most 16), i.e. the meaning of bit k
 * and column k below.  tri_batch_matched_terms, for the n matches of query q in ascending docID order (n and the docIDs as
 * returned by tri_batch_docset): present[i] bit k = term k matched document i; freq[i * nterms + k] = its frequency there
 * (term_hits::freq, 0 when absent); positions = the hits' positions, match-major then term-minor (the run of (i, k) starts at
 * the sum of all freq before it), *npos of them in all; pass positions == NULL to learn *npos. */
int tri_batch_query_terms(tri_batch *, size_t q, uint32_t *terms /* [16] */, uint32_t *nterms);
int tri_batch_matched_terms(tri_batch *, size_t q, uint32_t *present /* [n] */, uint16_t *freq /* [n * nterms] */, uint16_t *positions,
                            size_t pos_cap, size_t *npos);
/* TRI_FLAG_MATCHED_TERMS | TRI_FLAG_HIT_PAYLOADS batches: the payloads of query q's hits, parallel to `positions` above (same order,
 * *n == *npos): lens[i] = term_hit::payloadLen, payloads[i] = term_hit::payload — the word as materialize_hits leaves it, i.e. only its
 * first lens[i] bytes belong to this hit (google_codec.cpp:533-594).  Pass lens == payloads == NULL to learn *n. */
int tri_batch_matched_payloads(tri_batch *, size_t q, uint8_t *lens, uint64_t *payloads, size_t cap, size_t *n);
/* AccumulatedScore with topk >= 1: docids/scores are [nq][topk] row-major, counts[nq] = min(matches, topk) */
int tri_batch_topk(tri_batch *, uint32_t *docids, float *scores, uint32_t *counts);
/* device-resident result blocks for the multi-GPU gather (per rank: [nq][topk] u32 + f32, [nq] u32); valid once the run has
 * completed on the engine stream (tri_dev_stream) */
int tri_batch_topk_device(tri_batch *, void **docids, void **scores, void **counts);
/* device-resident per-query match counts of the last run, u64[nq] (every mode; rich mode: after the COUNT pass) */
int tri_batch_counts_device(tri_batch *, void **counts);
/* FNV-1a(64) of every query's docID set computed from the device results (tests at full size) */
int tri_batch_docset_hashes(tri_batch *, uint64_t *hashes /* [nq] */);

/* ---- collections of segments ------------------------------------------------------------------------
 * IndexSourcesCollection (index_source.cpp:3-30; exec.h:57-62: exec_query per source, every source masked by the documents the
 * newer ones update): one tri_batch per source — the SAME queries in the same order (term indices resolved against each source's
 * term table; per-token weights when the scores of the sources must share their statistics), same flags and topk, all on one device,
 * oldest source first, each index carrying its masked set (tri_index_set_masked).  The collection batch borrows them: run = the parts
 * back to back on the engine stream + a device merge — match counts add up, top-K lists merge K-way from the parts' partial lists
 * (score descending, docID ascending, as one application heap would hold them); docID sets concatenate source after source. */
typedef struct tri_cbatch tri_cbatch;
int tri_cbatch_create(tri_batch *const *parts, size_t n, tri_cbatch **out);
void tri_cbatch_destroy(tri_cbatch *);
/* status[q]: TRI_OK, or TRI_ERR_UNSUPPORTED when the planner left query q out of any part (tri_batch_query_status): its answer over the
 * collection would be incomplete, the caller keeps its CPU path for it */
int tri_cbatch_query_status(const tri_cbatch *, int32_t *status /* [nq] */);
int tri_cbatch_run(tri_cbatch *);
int tri_cbatch_sync(tri_cbatch *);
int tri_cbatch_match_counts(tri_cbatch *, uint64_t *counts /* [nq] */);
int tri_cbatch_topk(tri_cbatch *, uint32_t *docids, float *scores, uint32_t *counts);
int tri_cbatch_docset(tri_cbatch *, size_t q, uint32_t *out, size_t cap, size_t *n);

/* ---- multi-GPU result gather (RCCL over xGMI) -------------------------------------------------------
 * exec_query_par gives every source / shard its own result object and the caller combines them (exec.h:132-176).  One process per GPU,
 * queries sharded, index replicated: after a step every rank's fixed-shape result blocks are exchanged with one group of
 * ncclAllGather calls on the engine stream, straight from the device buffers (no host bounce).  RCCL is bound at run time (dlopen;
 * TRI_ERR_UNSUPPORTED when it cannot be loaded).  tri_comm_unique_id on rank 0, its 128 bytes handed to every rank by the launcher's
 * own means (an env var, a file, torch.distributed's store), tri_comm_create on every rank (collective).
 * tri_gather_results: counts_all u64[nranks][nq]; for AccumulatedScore top-K batches also docids_all u32[nranks][nq][k],
 * scores_all f32[nranks][nq][k], topk_counts_all u32[nranks][nq] — device buffers of the caller; enqueued behind the batch's run,
 * complete after tri_dev_sync.  (Every rank's batch has the same nq and topk.) */
typedef struct tri_comm tri_comm;
int tri_comm_unique_id(uint8_t id[128]);
int tri_comm_create(tri_dev *, const uint8_t id[128], int rank, int nranks, tri_comm **out);
/* The same gather over the caller's own transport (MPI, UCX, a test's gloo group): allgather(user, send, recv, bytes_per_rank, stream)
 * must leave, on every rank, rank r's bytes_per_rank bytes of `send` at recv + r * bytes_per_rank (both device memory), either ordered on
 * `stream` (the engine's hipStream_t) or complete when it returns; non-zero = failure.  tri_gather_results calls it once per block. */
typedef int (*tri_allgather_fn)(void *user, const void *send, void *recv, size_t bytes_per_rank, void *stream);
int tri_comm_create_custom(tri_dev *, int rank, int nranks, tri_allgather_fn allgather, void *user, tri_comm **out);
void tri_comm_destroy(tri_comm *);
int tri_gather_results(tri_batch *, tri_comm *, void *counts_all, void *docids_all, void *scores_all, void *topk_counts_all);

/* ---- write side ----------------------------------------------------------------------
 * Codecs::Google::Encoder (google_codec.cpp:9-176: begin_term / begin_document / new_hit / end_document / end_term, commit_block
 * :118-176) on the device: the postings of `nterms` terms, term after term — docs[] ascending and > 0 within a term, freqs[] the counted
 * hits of each posting, positions[npositions] those hits' positions in posting order (payload-less hits; > 0 — a position-0 hit without
 * payload is not a hit, google_codec.cpp:42-45 — and non-descending within a document, :49; sum(freqs) <= npositions), term_first[t] =
 * postings before term t ([nterms + 1]) — are encoded into the `index` bytes the reference's encoder writes for them, byte for byte
 * (skiplist cadence across terms included), and the term table (term_index_ctx: documents, chunk offset, chunk size).  Input that the
 * reference's encoder would not take (unsorted or zero positions, freqs[] reaching past positions[]) is refused with TRI_ERR_INVALID.
 * index_out == NULL: sizing call (*index_len and terms_out are filled). */
int tri_encode_google(tri_dev *, const uint32_t *docs, const uint32_t *freqs, const uint16_t *positions, size_t npositions, const uint64_t *term_first,
                      size_t nterms, uint8_t *index_out, size_t cap, size_t *index_len, tri_term *terms_out);

/* Codecs::Lucene::Encoder (lucene_codec.cpp:163-388) on the device (ABI 7): the same postings as tri_encode_google (payload-less hits) into the Lucene-shaped
 * segment — `index` (per term: the 14-byte header, 128-document blocks as two ints() groups, the prefix-varint tail, one 22-byte skiplist entry per block) and
 * `hits.data` (128-hit blocks of position deltas, varint tail) — with this repo's PFOR128 ints() payload (include/pfor128.md; the reference's FastPFor words
 * are absent from its tree): byte-identical to trinity_amd/csrc/host/lucene_encoder.hpp, the writer of every Lucene-shaped segment this engine reads.
 * index_out == NULL: sizing call (*index_len, *hits_len, terms_out). */
int tri_encode_lucene(tri_dev *, const uint32_t *docs, const uint32_t *freqs, const uint16_t *positions, size_t npositions, const uint64_t *term_first, size_t nterms,
                      uint8_t *index_out, size_t index_cap, size_t *index_len, uint8_t *hits_out, size_t hits_cap, size_t *hits_len, tri_term *terms_out);

/* SegmentIndexSession::commit (indexer.cpp:311-478) on the device (ABI 7).  The session's postings in INSERTION order — one entry per (document, term), as
 * document_proxy::insert serialised them (indexer.cpp:33-111): term_ids[i], doc_ids[i], freqs[i] = its counted hits, whose positions (and payloads) follow
 * each other in positions[] (payload_lens[] / payloads[]; NULL: none) in the same order — are sorted by (termID & 31, termID, documentID): the order the
 * reference's commit feeds its encoder in (it buckets by termID & 31 and sorts every bucket by (termID, documentID), :399-416; the Google encoder's skiplist
 * cadence runs across terms, so the order is part of the bytes), gathered, and encoded by the device encoder: index_out / *index_len = the `index` bytes,
 * term_ids_out[t] / terms_out[t] = the t-th term committed and its term_index_ctx, *nterms how many, *stats what commit adds to the field statistics
 * (:360 docsCnt, :457 sumTermHits, :470 sumTermsDocs, :476 totalTerms).  The bytes equal tri_encode_google_payloads over the same postings handed over
 * term after term in that order.  index_out == NULL: sizing call (*index_len, *nterms, *stats).  What commit or the encoder would refuse — document 0, the
 * same (term, document) twice, positions out of order — is TRI_ERR_INVALID naming the posting.  A hit at position 0 WITHOUT a payload is not a storable hit:
 * the reference's Encoder::new_hit skips it (google_codec.cpp:42-45) while commit still counts it in sumTermHits (indexer.cpp:447).  The caller drops such hits
 * before the call (freqs[i] = the hits it hands over) and adds their number to stats->sum_term_hits on its side — csrc/host/trinity_gpu_write.hpp's
 * SegmentIndexSession::insert / commit do exactly that; handed over as they are they are refused (TRI_ERR_INVALID) rather than silently re-counted. */
typedef struct tri_commit_stats {
        uint64_t docs_cnt, sum_terms_docs, sum_term_hits, total_terms;
} tri_commit_stats;
int tri_commit_google(tri_dev *, const uint32_t *term_ids, const uint32_t *doc_ids, const uint32_t *freqs, const uint16_t *positions, const uint8_t *payload_lens,
                      const uint64_t *payloads, size_t npostings, size_t npositions, uint8_t *index_out, size_t cap, size_t *index_len, uint32_t *term_ids_out,
                      tri_term *terms_out, size_t terms_cap, size_t *nterms, tri_commit_stats *stats);

/* Codecs::Google::IndexSession::merge (google_codec.cpp:186-438), for a whole dictionary at once (ABI 7): parts[0 .. nparts) are the participants' uploaded
 * google_codec indexes, MOST RECENT FIRST (merge.cpp:100-157 orders them so), each with the documents its newer segments mask installed by
 * tri_index_set_masked (its masked_documents_registry); part_terms[t * nparts + p] = output term t's index in participant p's term table (0xffffffff: the
 * participant does not hold it) — the caller walks its dictionaries as MergeCandidatesCollection::merge does (merge.cpp:100-157) and hands the output terms
 * over in the order it wants them encoded.  Per output term: the union of the participants' documents; a document several participants hold comes from the
 * most recent one; it is dropped when THAT participant's masked set holds it (google_codec.cpp:377-401) — hits and payloads copied, re-encoded by the device
 * encoder.  index_out / terms_out[t] as tri_encode_google_payloads (a term that keeps no document: documents == 0, its two chunk bytes are still in the
 * index — begin_term / end_term ran, merge.cpp:275-282 — and the caller leaves it out of the dictionary); *stats: what the merge adds to the field
 * statistics (docs_cnt is the caller's).  The bytes equal the device / host encoder's over the merged postings; the reference's raw-chunk fast path for a
 * term only one unmasked participant holds (append_index_chunk, merge.cpp:167-178) copies bytes this call re-encodes.  index_out == NULL: sizing call. */
int tri_merge_google(tri_dev *, tri_index *const *parts, size_t nparts, const uint32_t *part_terms, size_t nterms, uint8_t *index_out, size_t cap, size_t *index_len,
                     tri_term *terms_out, tri_commit_stats *stats);
/* Codecs::Lucene::IndexSession::merge (lucene_codec.cpp:963-1396), likewise for a whole dictionary (ABI 8): the same walk — participants most recent first, the most recent
 * holder's posting wins, dropped when that participant masks the document — over lucene_codec indexes uploaded WITH their hits.data; what is kept is re-encoded by the device's
 * Lucene-shaped encoder (tri_encode_lucene: PFOR128 ints() payload, positions only — a participant's payloads are not carried, as in tri_encode_lucene): index_out / hits_out =
 * the merged `index` and `hits.data`, terms_out[t] as tri_encode_lucene.  The bytes equal the host encoder's over the merged postings; the walk is the one
 * tests/golden/ref_merge.json pins for the Google codec (the reference's Lucene side needs FastPFor: unbuildable here, so this pair stays unpinned by reference bytes).
 * index_out == NULL: sizing call (*index_len, *hits_len). */
int tri_merge_lucene(tri_dev *, tri_index *const *parts, size_t nparts, const uint32_t *part_terms, size_t nterms, uint8_t *index_out, size_t cap, size_t *index_len,
                     uint8_t *hits_out, size_t hits_cap, size_t *hits_len, tri_term *terms_out, tri_commit_stats *stats);

/* tri_commit_google with the session's encoder being the Lucene-shaped codec's (commit is codec-agnostic: sess->new_encoder(), indexer.cpp:323): the same sort
 * and gather, then tri_encode_lucene's device encoder — `index` + `hits.data`; payload-less hits. */
int tri_commit_lucene(tri_dev *, const uint32_t *term_ids, const uint32_t *doc_ids, const uint32_t *freqs, const uint16_t *positions, size_t npostings, size_t npositions,
                      uint8_t *index_out, size_t cap, size_t *index_len, uint8_t *hits_out, size_t hits_cap, size_t *hits_len, uint32_t *term_ids_out, tri_term *terms_out,
                      size_t terms_cap, size_t *nterms, tri_commit_stats *stats);

/* The same with hit payloads (Encoder::new_hit(pos, payload), google_codec.cpp:38-74): payload_lens[h] (0 .. 8) and payloads[h] (the
 * payload's first byte in the low 8 bits) per hit, parallel to positions[].  A hit is written as varint(position delta << 1 | the
 * length differs from the previous hit's of the document) [u8 new length] payload bytes, the length state restarting with every
 * document (:34).  A position-0 hit WITH a payload is a counted hit (:42-45); one without is refused, as by tri_encode_google.
 * payload_lens == NULL: no hit has a payload (tri_encode_google). */
int tri_encode_google_payloads(tri_dev *, const uint32_t *docs, const uint32_t *freqs, const uint16_t *positions, const uint8_t *payload_lens,
                               const uint64_t *payloads, size_t npositions, const uint64_t *term_first, size_t nterms, uint8_t *index_out, size_t cap,
                               size_t *index_len, tri_term *terms_out);

#ifdef __cplusplus
}
#endif
#endif
