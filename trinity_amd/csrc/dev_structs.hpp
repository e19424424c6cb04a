// dev_structs.hpp — device-visible plan structures shared by the host planner and the kernels
// Part of libtrinity_hip.so (MI355X / gfx950); included by trinity_hip.hip.  New code, no reference source.
#pragma once
#include <cstdint>
#ifdef __HIPCC__
#define TRI_HD __host__ __device__
#else
#define TRI_HD // (the host planner and its CPU tests compile these headers with g++)
#endif

struct DevTerm {
        uint32_t documents;
        uint32_t first_block;
        uint32_t nblocks;
        uint32_t last_n;  // docs in the final block (1..32)
        uint32_t win_off; // lists of >= WIN_MIN_BLOCKS blocks: row in win[] (first block with last >= w * SPAN_BITS, per window w); else ~0
        uint32_t flags;   // TERM_FULL_BLOCKS: every block but the last holds 32 documents (always true for chunks written by the
                          // reference encoder, google_codec.cpp:76-88; verified at upload) => n needs no load
        uint32_t npfor;   // LUCENE: directory rows that are quarters of full 128-document PFOR blocks (the rest is the varbyte tail)
        uint32_t pad;     // LUCENE with hits.data: the term's row in hdir[] ({nfull, off[nfull], tail_off}); otherwise 0
};
constexpr uint32_t TERM_FULL_BLOCKS = 1u;
constexpr uint32_t TERM_SPARSE = 2u; // GOOGLE: fewer than 1 document in 28 docIDs on average — most blocks hold a multi-byte delta, so the
                                     // bitmap kernel parses them from registers in place instead of trying the static one-byte path

// documents in block b of term t
#define TRI_BLOCK_N(t, b, index, off) (((t).flags & TERM_FULL_BLOCKS) ? ((b) + 1 == (t).nblocks ? (t).last_n : 32u) : (uint32_t)(index)[(off)-1])
constexpr uint32_t WIN_MIN_BLOCKS = 128;

// A query in conjunctive normal form: AND of groups, a group = one term or an OR of terms.  qterms[] lists the terms
// group by group, cheapest group first (exec.cpp:35-110 cost model); bit 31 marks the first term of a group.
// Root OR of terms == a single group.  (Conjuction / DisjunctionAllPLI semantics, docset_iterators.cpp:226-405.)
constexpr uint32_t QT_GROUP = 0x80000000u;
// logicalnot (DocsSetIterators::Filter, docset_iterators.cpp:652-677): the documents of the excluded terms are removed from the
// conjunction.  The excluded terms form ONE group, always the last of the query; QT_NOT marks its first term.
constexpr uint32_t QT_NOT = 0x40000000u;
constexpr uint32_t QT_TERM = 0x3fffffffu; // the term id inside a qterms[] word
constexpr uint32_t MAX_QTERMS = 16;
struct DevQuery {
        uint32_t nterms;    // total terms over all groups (<= MAX_QTERMS)
        uint32_t term_base; // into qterms[]
        uint64_t out_off;   // docID slots
        uint32_t out_cap;
        uint32_t qid; // caller's query index
        uint32_t first_task, ntasks;
        uint32_t score_base, nscore; // AccumulatedScoreScheme: sterms[]/sweights[] slice, reference summation order
        uint32_t phrase_base, nphrases; // positional constraints applied to the match list (DocsSetIterators::Phrase)
        uint32_t fused_idx, form;       // fused_idx: TASK_FUSED queries: row in the batch's DevFused table (k_fused.hpp); TASK_TREE: the query's record in tree[].
                                        // form: RESULT_DOCIDS / RESULT_BITMAP — how the query's docID set lies in out[]
};

// How a materialised docID set lies in its region of out[]: ascending docIDs, task segment after task segment (the in-order concatenation is the
// set), or — DocumentsOnly bitmap-window queries whose expected matches outnumber the bitmap's words (a union of head terms, a conjunction of two
// of them: one document in 32 or denser) — ONE BIT PER DOCUMENT: word i of the region covers docIDs [32 i, 32 i + 32), a task writes the words of
// its docID windows (every one of them, matches or not), counts[] still holds the task's matches.  1.25 MB per query at 10 M documents whatever it
// matches, instead of 4 bytes per match and the expansion's instructions: what a consumer that replays consider(ids, cnt) expands on its side
// (tri_batch_docset does it on read-back; tri_batch_docset_bitmap hands the words over as they are)
constexpr uint32_t RESULT_DOCIDS = 0, RESULT_BITMAP = 1;
// A phrase constraint: its terms (phrase order) live in pterms[term_base .. +nterms); weight = sum of the terms' idf
// (similarity.h:209-217).
constexpr uint32_t MAX_PHRASE_TERMS = 16; // trinity_limits.h:12 MaxPhraseSize
struct DevPhrase {
        uint32_t term_base;
        uint32_t nterms;
        double weight;
};

// Unit of scheduling: a run of lead-list tiles of one query.  Heavy queries are cut into many tasks so that no
// single workgroup carries a multi-millisecond tail; task `i` of a query writes its (ascending) matches at
// out_off + tile_begin * TILE_CANDS, a region no other task can reach because matches are a subset of the
// lead tile's documents.  A query's docID set is the in-order concatenation of its tasks' segments.
struct DevTask {
        uint32_t slot;       // plan slot of the query
        uint32_t tile_begin; // TASK_CAND: lead tiles [tile_begin, tile_end); TASK_DENSE: docID windows [begin, end)
        uint32_t tile_end;
        uint32_t kind;
        uint64_t out_off; // absolute docID slot in out[] where this task's segment starts
};
constexpr uint32_t CAND_QUEUES = 8;      // k_and's task queues: one per XCD (BatchPlan::cand_q), a ticket word each, 64 bytes apart (CAND_TICKET_STRIDE words)
constexpr uint32_t CAND_TICKET_STRIDE = 16;
constexpr uint32_t CAND_HEAVY_TILES = 3; // a TASK_CAND task of more lead tiles runs at its queue's start, in cost order
constexpr uint32_t TASK_CAND = 0;  // candidate tiles of the lead list, filtered by galloping / block-driven merge
constexpr uint32_t TASK_DENSE = 1; // bitmap algebra over fixed docID windows (every list dense enough)
constexpr uint32_t TASK_FUSED = 2; // AccumulatedScoreScheme + top-K of a dense query: decode, match, score and select in one pass over docID
                                   // windows (k_fused.hpp); tile_begin / tile_end count windows of FUS_W documents; nothing is written to out[]
                                   // (general trees in DocumentsOnly mode — FUS_MODE_EMIT — write their matches there)
constexpr uint32_t TASK_FUSED16 = 3; // ... with 16-bit window words (<= 5 distinct terms): windows of 2 * FUS_W documents

constexpr uint32_t TASK_FUSED_GEN = 4; // ... a general tree (truth-table predicate; DocumentsOnly: matches written to out[]): 32-bit window words
constexpr uint32_t TASK_PLANES = 5;    // AccumulatedScoreScheme + top-K of a CNF query over BIT PLANES (k_planes.hpp): per slot a presence bit and an
                                       // "frequency is not 1" bit per document; windows of PL_W documents; tile_begin / tile_end count those windows
constexpr uint32_t TASK_PLANES8 = 6;   // ... of a query with more than five slots (its own instantiation: more words held in registers)
constexpr uint32_t TASK_PSET = 7;      // docID windows of a query ALL of whose terms have a term plane (k_psets.hpp): word-wise algebra over the planes, then
                                       // the expansion; same windows, same private output regions as TASK_DENSE (a docset-materialising kind, not a one-pass one)
constexpr uint32_t TASK_PROBE = 8;     // candidate tiles of ONE lead list tested against lists that all have a term plane (k_probe.hpp): a wave decodes 64 lead
                                       // blocks into registers and probes the planes — same tiles, same private output regions as TASK_CAND
constexpr uint32_t TASK_TREE = 9;      // ANY tree — multi-word phrases under OR / NOT / matchsome, more distinct terms than a truth table holds, CNFs wider than
                                       // MAX_QTERMS — evaluated over whole-corpus bitmaps, one per leaf (k_tree.hpp); one task per query; tile_end = its chunks
constexpr uint32_t TASK_KINDS = 10;
// A TASK_PSET task as k_psets reads it: ONE 64-byte record instead of the sched -> task -> query -> qterms / qplane chain of dependent loads
// (four memory round trips before a two-window task's first plane word: measured, they were most of the kernel's fixed cost).  Written by
// the planner next to the DevTask (which the host keeps reading for the result read-back); units[] is indexed like the schedule's TASK_PSET
// section: sched[n_dense + i] names a task, pset_of_task gives its unit.
constexpr uint32_t PSET_INLINE_TERMS = 4;
#ifndef TRI_PSET_TASK_WINDOWS
#define TRI_PSET_TASK_WINDOWS 4
#endif
constexpr uint32_t PSET_TASK_WINDOWS = TRI_PSET_TASK_WINDOWS; // docID windows per TASK_PSET task: the same for every query, so that the tasks of a window range line up —
                                          // the schedule runs them window range by window range, and a range's plane words (88 head terms x 32 KB at cfg2)
                                          // stay in the XCDs' L2 while every query that reads them is in flight
struct DevPsetUnit { // (also the record of a TASK_PROBE task: w_begin / w_end are its lead tiles, tt[0] its lead term)
        uint64_t out_off;   // the task's private output region
        uint32_t w_begin, w_end;
        uint32_t tix;       // index into counts[]
        uint32_t nterms;
        uint32_t term_base; // qterms[] / qplane[] slice (read by the kernel only when nterms > PSET_INLINE_TERMS)
        uint32_t first;     // bit 0: the first task of its query (planner bookkeeping); bit 1 (PSET_UNIT_BITMAP): the query's result is a bitmap (RESULT_BITMAP); bit 2: PSET_UNIT_SCATTER
        uint32_t tt[PSET_INLINE_TERMS];  // qterms[] words (term | QT_GROUP | QT_NOT) ...
        uint32_t row[PSET_INLINE_TERMS]; // ... and the terms' plane rows
};
static_assert(sizeof(DevPsetUnit) == 64, "one cache-line half per unit");
constexpr uint32_t PSET_UNIT_FIRST = 1u, PSET_UNIT_BITMAP = 2u;
constexpr uint32_t PSET_UNIT_ROUND_SHIFT = 4, PSET_UNIT_ROUND_MASK = 15u; // DevPsetUnit::first bits 4 .. 7: docID windows per ROUND of the task (k_psets.hpp: the waves' counts cross once a round;
                                                                          // planner.hpp sizes it so that a wave's survivors of a round fit its staging buffer); 0 reads as 1
constexpr uint32_t PSET_ROUND_DOCS = 16384, PSET_STAGE_DOCS = 1024; // documents of a window a wave of k_psets owns (eight waves), docIDs its staging buffer holds
constexpr uint32_t PSET_UNIT_SCATTER = 4u; // a UNION some of whose terms have no plane (qplane[] = PL_NONE for them), result a bitmap: the plane terms' words are OR-ed and stored,
                                           // then the other terms' documents of the task's range are set in the stored words one by one (k_psets.hpp: psets_scatter)
// ---- TASK_TREE: the query tree as the kernels read it (k_tree.hpp).  A record in the plan's tree[] words (DevQuery::fused_idx = its first word):
//      TREE_HDR_WORDS header words { nnodes, 0... }, then nnodes DevTreeNode in POSTFIX order (children before parents, the root last)
constexpr uint32_t TREE_MAX_NODES = 64;     // node values and "an iterator sits on the document" flags are bit sets in a 64-bit word
constexpr uint32_t TREE_HDR_WORDS = 8;
constexpr uint32_t TREE_CHUNK_WORDS = 2048; // bitmap words (x 32 documents) a workgroup of the tree kernels takes
constexpr uint32_t TREE_ROW_PHRASE = 0x80000000u; // DevTreeNode::row: a one-plane row of the batch's phrase rows (else: a PL_PLANES-plane row of its tree rows)
struct DevTreeNode {
        uint8_t op;     // TRI_OP_TERM (a term leaf), TRI_OP_PHRASE (a multi-word phrase leaf), TRI_OP_AND / OR / NOT / OPT / SOME
        uint8_t parent; // node index; the root: 0xff
        uint8_t ord;    // position among the parent's children (NOT / OPT: 0 = the required / main side)
        uint8_t thr;    // SOME: the threshold
        uint32_t arg;   // term leaf: the term; phrase leaf: the plan slot of the hidden query that evaluates the phrase
        uint32_t row;   // leaf: its bitmap row
        uint32_t score; // leaf: its scorer (index into the query's sterms / sweights slice), 0xffffffff: none (an excluded side scores nothing)
        uint32_t rmask; // leaf, default mode: the reportable terms (bits into the query's sterms slice) reported where the leaf sits on the document
        uint8_t kid0, kid1, pad0, pad1; // NOT / OPT: the two sides
        uint64_t kids;  // inner node: bit k = node k is a child
};
static_assert(sizeof(DevTreeNode) == 32, "eight words per node");
// the one-pass kinds (decode -> match -> score -> top-K inside one kernel: k_fused / k_planes); the others materialise docID sets
TRI_HD constexpr bool task_onepass(const uint32_t kind) { return kind >= TASK_FUSED && kind <= TASK_PLANES8; }

// ---- term planes: every head term the queries of an index's batches share is decoded ONCE per index (k_term_planes, the first time a batch's run names
//      it; the rows live in tri_index's plane cache) into bitmaps over the docID space, which the matching kernels then read instead of decoding the
//      term's list again for every query that names it (under Zipf a handful of terms carry most of a batch's postings).  A row has TWO parts, kept in
//      two regions and built BY NEED: plane 0 (the document holds the term: all a DocumentsOnly batch reads — k_and's probes, k_psets' / k_and_dense's
//      words, k_phrase's rank records) and, only once a scored batch names the row, the HIGH part: nested planes 1 .. PL_STORED - 1 and the level words
constexpr uint32_t PL_W = 32768;          // documents per plane window (k_term_planes, k_planes)
constexpr uint32_t PL_WORDS = PL_W / 32;  // words of one plane per window
constexpr uint32_t PL_NONE = 0xffffffffu; // "this term has no plane in this batch"
constexpr uint32_t PL_NESTED = 6;         // nested planes per term — plane k (0-based): the document holds the term and its frequency f there is >= k + 1, or is one the
                                          // planes do not tell (f = 0, f >= PL_NESTED: every plane set — the exact value is then read from the postings).  Plane 0 ("A")
                                          // is presence; planes 1 / 2 ("B" / "C") read as before: f is not 1, nor 2.  A document's LEVEL is the number of nested planes
                                          // it is in: levels 1 .. PL_NESTED - 1 ARE the frequency.  The nested planes are what a SWEEP streams (one plane, front to back)
constexpr uint32_t PL_LEVEL_WORDS = 3;    // ... and after them the same levels bit-sliced and INTERLEAVED — words 3 w, 3 w + 1, 3 w + 2: bit 0, 1, 2 of the level of the
                                          // 32 documents of word w — what a PROBE reads: one 12-byte access tells a document's frequency where the nested planes take six
constexpr uint32_t PL_STORED = 4;         // the nested planes a row actually holds: 0 .. PL_STORED - 1 (f >= 1 .. f >= 4).  A sweep that wants "f > c" for a higher c streams
                                          // plane PL_STORED - 1 instead (a superset: the level words sort it out) — the planes above it are one bit in a thousand and cost a
                                          // full plane each to build, to keep and to stream
constexpr uint32_t PL_PLANES = PL_STORED + PL_LEVEL_WORDS; // words of a term's row per word of the docID space, both parts together (a TASK_TREE batch's own rows keep them side by side: stride PL_PLANES * plw)
constexpr uint32_t PL_HI = PL_PLANES - 1;                 // ... of its HIGH part: nested planes 1 .. PL_STORED - 1 (plane k at (k - 1) * plw), then the interleaved level words (at
                                                          // PL_HI_LEVELS * plw + 3 w).  Plane cache: plane 0 of row r = planes0 + r * plw; its high part = planes_hi + r * PL_HI * plw
constexpr uint32_t PL_HI_LEVELS = PL_STORED - 1;
constexpr uint32_t PL_RANK_WORDS = 16;     // ... a 64-byte record: eight pairs, one per plane-0 word of the group: { the posting index of the word's first document, the word } — a
                                           // document's rank is one eight-byte load and a popcount (round 5: the group's first posting + the eight words, three 16-byte loads and eight popcounts)
constexpr uint32_t PL_RANK_DOCS = 256;     // a row's rank directory (tri_index::d_prank) has an entry per this many documents: the posting index of the group's first document
constexpr uint32_t BLK_HITS_PLAIN = 0x80000000u; // GOOGLE blk_hits[]: every hit of the block is a single byte (no payload, position delta < 64).  The entry's
                                                 // low 31 bits: where the block's hits start, in bytes PAST blk_off[] (the block's deltas and frequencies lie
                                                 // between: a few hundred bytes) — index offsets themselves keep all their 32 bits (codecs.h:26: chunks up to 4 GiB)
// ---- geometry the host planner (planner.hpp) and the kernels share
constexpr int TILE_BLOCKS = 256;               // k_and: lead blocks per candidate tile (one 32-candidate row per lane)
constexpr int TILE_CANDS = TILE_BLOCKS * 32;
constexpr uint32_t SPAN_BITS = 1u << 17;       // docIDs per bitmap window (k_and_dense)
constexpr uint32_t SPAN_WORDS = SPAN_BITS / 32;
constexpr uint32_t CELL_LOG2 = 10;             // docID cells of the per-term block index (DevTerm::win_off)
constexpr uint32_t CELL_DOCS = 1u << CELL_LOG2;
constexpr uint32_t CELLS_PER_SPAN = SPAN_BITS / CELL_DOCS;
#ifndef TRI_FUS_CELLS
#define TRI_FUS_CELLS 14
#endif
constexpr uint32_t FUS_CELLS = TRI_FUS_CELLS;  // k_fused: docID cells (of CELL_DOCS) per window (14: 56 KB of words, two 512-thread workgroups per CU)
constexpr uint32_t FUS_W = FUS_CELLS * CELL_DOCS;
constexpr uint32_t PLK_MAX_SPARSE = 6;         // k_planes: slots whose lists are decoded per task (LDS planes); the planner sends wider queries to k_fused
constexpr uint32_t PLK_NS_SMALL = 5;           // k_planes: the instantiation for queries of up to this many slots keeps six words per slot in registers
constexpr uint32_t TOPK_MAX = 256;             // AccumulatedScore: largest K of a top-K batch
constexpr uint32_t FUS_MAX_SLOTS = 8;
constexpr uint32_t FUS_MAX_LEAVES = 16; // scorer leaves of a general tree
// planner -> kernel: how a fused query's terms map onto the window words
struct DevFused {
        uint32_t nslots;                // distinct terms: CNF terms (incl. the excluded group) and scoring-only (optional) terms
        uint32_t fbits;                 // field width — 32-bit words: 8 (<= 4 slots) or 4; 16-bit words (hw): 8 / 5 / 4 / 3 for <= 2 / 3 / 4 / 5 slots
        uint32_t cap;                   // largest freq a field holds exactly; cap + 2 <= 1 << fbits (code cap + 1 = "cap or more")
        uint32_t nreq;                  // required groups
        uint32_t term[FUS_MAX_SLOTS];   // slot -> term
        uint32_t gmask[FUS_MAX_SLOTS];  // per required group: the fields of its slots ((word & gmask) != 0 <=> group satisfied)
        uint32_t gslots[FUS_MAX_SLOTS]; // per required group: bit s = slot s belongs to it (window skipping)
        uint32_t nmask;                 // fields of the excluded group (logicalnot), 0 = none
        uint32_t hw;                    // 1: 16-bit window words (two documents per LDS word, windows twice as long)
        // ---- general trees (matchsome, NOT / Optional of any subtree, nested AND / OR): the match predicate is a truth table over the
        //      slots' PRESENCE bits, and so is every scorer leaf's "contributes to the score of this document" (the reference sums the
        //      sub-iterators that sit on the document: docset_iterators_scorers.cpp:38-57, 77-104, 107-193)
        uint32_t mode;                          // FUS_MODE_* bits
        uint32_t nleaf;                         // scorer leaves (== DevQuery::nscore)
        uint32_t tt[8];                         // bit p: a document that holds exactly the slots of pattern p matches
        uint8_t leaf_slot[FUS_MAX_LEAVES];      // scorer leaf -> slot of its term
        uint32_t ctt[FUS_MAX_LEAVES][8];        // per scorer leaf: the patterns in which it adds its score
        // ---- TASK_PLANES (k_planes.hpp): the slot's row in the batch's term planes (PL_NONE: its list is decoded per window), the slots of
        //      the excluded group as a bit set
        uint32_t plane[FUS_MAX_SLOTS];
        uint32_t negslots;
};
constexpr uint32_t FUS_MODE_TT = 1;   // predicate = tt, scores through ctt
constexpr uint32_t FUS_MODE_EMIT = 2; // DocumentsOnly: the window's matches are written to out[] (ascending), nothing is scored

