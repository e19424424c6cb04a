// k_planes.hpp — term planes (a head term decoded once per launch for every query of the batch that names it) and the
// AccumulatedScoreScheme top-K kernel that runs over them
// Part of libtrinity_hip.so (MI355X / gfx950); included by trinity_hip.hip.  New code, no reference source.
#pragma once
#include <type_traits>
#include "k_fused.hpp"

// Under Zipf a handful of terms carry most of a batch's postings (at the 10M-document configuration the 40 most frequent terms
// hold 98 % of the postings the 5-term queries of SURVEY §8(d) cfg3 touch), and the reference decodes such a list again for every
// query that names it (Decoder::init + next() per query, google_codec.cpp:777-819 / lucene_codec.cpp:568-594).  Here each of those lists is
// decoded ONCE PER INDEX — k_term_planes, from the segment's own codec bytes, the first time a batch's run names the term; the rows stay in the
// index's plane cache (tri_index, trinity_hip.hip) — into PL_NESTED nested bitmaps over the docID space (+ the same levels bit-sliced, for probes):
//     plane 0 ("A")  bit d set  <=>  document d holds the term            (PostingsListIterator::current() would stop on d)
//     plane k        bit d set  <=>  ... and its frequency there is >= k + 1 — or one the planes do not tell (0, >= PL_NESTED: every plane set)
// (planes 1 / 2, "B" / "C": the frequency is not 1, nor 2 — what k_score and k_tree_leaves read).  The number of planes a document is in is its
// LEVEL: levels 1 .. PL_NESTED - 1 are the frequency itself, the top level says "read it from the postings" (round 5: three planes told 1 / 2 /
// anything else, and a union with the most frequent term walked into the postings for every document that holds it three times or more)
// and the matching kernels read the planes: k_and tests a candidate with one bit probe instead of bracketing and decoding a block
// (Conjuction::next_impl's advance(), docset_iterators.cpp:308-348), k_and_dense ORs a plane's words into its window bitmap instead
// of walking the term's rows (docset_spans.cpp:98-173), k_planes (below) evaluates union / CNF predicates 32 documents per word.
// A row is built BY NEED (round 6): plane 0 — 1.25 MB per term at 10 M documents, all a DocumentsOnly batch reads — when any batch names the term, the
// HIGH part (planes 1 .. PL_STORED - 1 + the three level words: 7.5 MB more) the first time a SCORED batch does (dev_structs.hpp: PL_HI).

constexpr uint32_t PL_CELLS = PL_W / CELL_DOCS; // cell-index entries per plane window
constexpr uint32_t PL_STRIDE = PL_WORDS + 32;   // LDS words between a slot's planes (word PL_WORDS of each: the sink)

// A decoded posting into the LDS planes (a[k * PL_STRIDE ...]: plane k).  Documents outside the window land in the sink word.
struct PlanePost {
        uint32_t *a;
        uint32_t *rd;  // the window's rank directory (LDS): per group of PL_RANK_DOCS documents the lowest posting index seen
        uint32_t pidx; // the posting index of the row's next document (rows of a list of full blocks: 32 b + slot)
        bool levels;   // false: plane 0 only (the row's high part is not being built)
        __device__ __forceinline__ void rank(const uint32_t rel) {
                if (rel < PL_W)
                        atomicMin(&rd[rel / PL_RANK_DOCS], pidx);
                ++pidx;
        }
        __device__ __forceinline__ void doc(const uint32_t rel) {
                const uint32_t r = min(rel, PL_W);
                rank(rel);
                atomicOr(&a[r >> 5], 1u << (r & 31u));
        }
        __device__ __forceinline__ void operator()(const uint32_t rel, const uint32_t f) {
                const uint32_t r = min(rel, PL_W);
                rank(rel);
                const uint32_t bit = 1u << (r & 31u), f16 = f & 0xffffu; // (the frequency a scorer sees is tokenpos_t, 16 bits: codecs.h:217)
                const uint32_t level = f16 == 0u || f16 > PL_NESTED ? PL_NESTED : f16; // (a frequency the planes do not tell: every plane)
                atomicOr(&a[r >> 5], bit);
                if (levels) {
#pragma unroll
                        for (uint32_t k = 1; k < PL_NESTED; ++k)
                                if (k < level)
                                        atomicOr(&a[(r >> 5) + k * PL_STRIDE], bit);
                }
        }
};

// Word k of a group's rank record (dev_structs.hpp: PL_RANK_WORDS): eight pairs { the posting index of the first document of the group's word i, that word of
// plane 0 } — a document's rank is ONE eight-byte load and a popcount
__device__ __forceinline__ uint32_t rank_rec_word(const uint32_t first /* the group's first posting */, const uint32_t *w8 /* its eight plane-0 words (LDS) */, const uint32_t k) {
        if (k & 1u)
                return w8[k >> 1];
        uint32_t before = first;
#pragma unroll
        for (uint32_t i = 0; i < 7; ++i)
                before += i < (k >> 1) ? (uint32_t)__popc(w8[i]) : 0u;
        return before;
}

// One workgroup per (plane row, window): the rows (<= 32 documents each) of the term that reach the window are decoded, one lane
// per row, into LDS planes, which are then written out whole — every word of every plane is written by exactly one workgroup,
// so the scratch region needs no clearing between launches.
template <int CODEC>
__global__ __launch_bounds__(AND_WG) void k_term_planes(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last,
                                                        const uint32_t *__restrict__ blk_off, const uint4 *__restrict__ blk_rec,
                                                        const uint32_t *__restrict__ blk_doff, const uint32_t *__restrict__ win,
                                                        const DevTerm *__restrict__ terms, const uint32_t *__restrict__ build /* (term, row) pairs */,
                                                        uint32_t *__restrict__ planes0, const size_t stride0 /* words between two rows' plane 0 */,
                                                        uint32_t *__restrict__ planes_hi /* the rows' high parts, or null: plane 0 only */, const size_t stride_hi,
                                                        const uint32_t plw, uint32_t *__restrict__ prank /* rank directories, or null */) {
        __shared__ uint32_t pl[PL_NESTED * PL_STRIDE];
        __shared__ uint32_t rdir[PL_W / PL_RANK_DOCS];
        const uint32_t tid = threadIdx.x, w = blockIdx.x, row = build[2 * blockIdx.y + 1];
        const bool hi = planes_hi != nullptr; // (uniform)
        for (uint32_t i = tid; i < (hi ? PL_NESTED : 1u) * PL_STRIDE; i += AND_WG)
                pl[i] = 0;
        for (uint32_t i = tid; i < PL_W / PL_RANK_DOCS; i += AND_WG)
                rdir[i] = 0xffffffffu;
        const DevTerm t = terms[build[2 * blockIdx.y]];
        const uint32_t *bl = blk_last + t.first_block;
        const uint32_t w0 = w * PL_W;
        // rows that can hold documents of [w0, w0 + PL_W): first row whose last docID >= w0 ... first row whose last docID >= the next
        // window's first docID (it may still begin inside this one)
        uint32_t b_lo, b_hi;
        if (t.win_off != 0xffffffffu) {
                b_lo = win[t.win_off + w * PL_CELLS];
                b_hi = win[t.win_off + (w + 1) * PL_CELLS];
        } else { // (a short list: planes are made for long ones, but the planner may be told to give every term one)
                uint32_t lo = 0, hi = t.nblocks;
                while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (bl[mid] < w0)
                                lo = mid + 1;
                        else
                                hi = mid;
                }
                b_lo = lo;
                hi = t.nblocks;
                while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (bl[mid] < w0 + PL_W)
                                lo = mid + 1;
                        else
                                hi = mid;
                }
                b_hi = lo;
        }
        b_lo = uni(b_lo);
        b_hi = uni(min(b_hi, t.nblocks - 1));
        __syncthreads();
        if (b_lo < t.nblocks)
                for (uint32_t b = b_lo + tid; b <= b_hi; b += AND_WG) {
                        const uint32_t prev = b ? bl[b - 1] : 0, last = bl[b];
                        PlanePost post{pl, rdir, 32u * b, hi};
#ifdef TRI_PROF
                        ProfClock prof_;
#endif
                        if (CODEC == CODEC_LUCENE) {
                                const uint4 rec = blk_rec[t.first_block + b];
                                row_decode<CODEC, true, PlanePost>(index, t, b, rec.x, rec.y, rec.z, rec.w, TRI_BLOCK_N(t, b, index, 0), prev, last, w0, post PROF_PASS);
                        } else {
                                const uint32_t off = blk_off[t.first_block + b];
                                const uint32_t dlen = blk_doff[t.first_block + b + 1] - blk_doff[t.first_block + b] - 1u;
                                row_decode<CODEC, true, PlanePost>(index, t, b, off, dlen, 0, 0, TRI_BLOCK_N(t, b, index, off), prev, last, w0, post PROF_PASS);
                        }
                }
        __syncthreads();
        uint32_t *pa = planes0 + (size_t)row * stride0 + (size_t)w * PL_WORDS;
        if (prank) { // (rank of a document = its group's entry + the plane-0 bits of the group before it, both in ONE 64-byte record: k_phrase.hpp)
                static_assert(PL_RANK_DOCS == 256 && PL_RANK_WORDS == 16, "a record: eight (rank, plane-0 word) pairs, one cache line");
                uint32_t *rec = prank + ((size_t)row * (plw / (PL_RANK_DOCS / 32u)) + (size_t)w * (PL_W / PL_RANK_DOCS)) * PL_RANK_WORDS;
                for (uint32_t i = tid; i < (PL_W / PL_RANK_DOCS) * PL_RANK_WORDS; i += AND_WG) {
                        const uint32_t g = i / PL_RANK_WORDS, k = i % PL_RANK_WORDS;
                        rec[i] = rank_rec_word(rdir[g], pl + 8u * g, k);
                }
        }
        if (!hi) { // plane 0 alone: all a DocumentsOnly batch reads
                for (uint32_t i = tid; i < PL_WORDS; i += AND_WG)
                        pa[i] = pl[i];
                return;
        }
        static_assert(PL_NESTED == 6 && PL_LEVEL_WORDS == 3, "the level's bits below are written for six nested planes");
        uint32_t *ph = planes_hi + (size_t)row * stride_hi + (size_t)w * PL_WORDS;                           // the window's words of nested plane 1; plane k: + (k - 1) * plw
        uint32_t *lv = planes_hi + (size_t)row * stride_hi + (size_t)PL_HI_LEVELS * plw + 3u * (size_t)w * PL_WORDS; // the window's interleaved level words
        for (uint32_t i = tid; i < PL_WORDS; i += AND_WG) {
                uint32_t x[PL_NESTED];
#pragma unroll
                for (uint32_t k = 0; k < PL_NESTED; ++k) {
                        x[k] = pl[k * PL_STRIDE + i];
                        if (k >= 1 && k < PL_STORED)
                                ph[(size_t)(k - 1) * plw + i] = x[k];
                }
                pa[i] = x[0];
                // the level (the number of nested planes a document is in) bit-sliced: odd; 2, 3 or 6; 4 or more
                lv[3u * i] = x[0] ^ x[1] ^ x[2] ^ x[3] ^ x[4] ^ x[5];
                lv[3u * i + 1u] = (x[1] & ~x[3]) | x[5];
                lv[3u * i + 2u] = x[3];
        }
}

// PLANE 0 ALONE (+ the rank records, when a phrase batch asks for them): what a DocumentsOnly batch builds of a row — and what a stream whose head terms
// change from batch to batch pays per batch (bench.py's rotating.cold_planes leg).  One workgroup per (row, P0_GROUP windows): k_term_planes' workgroup per
// window spends its time on fixed costs for all but the longest lists — a term of the plane threshold's length brings ONE block per window, and the
// workgroup still walks build[] -> terms[] -> win[] -> blk_last[] -> blk_off[] -> the block's bytes (six dependent round trips), clears and writes an LDS
// plane; at cfg2 178 K such workgroups (583 rows x 306 windows) took 1.64 ms to decode 103 MB of lists.  Here the chain is paid once per P0_GROUP windows,
// and the lanes of a round hold that many times more blocks.
constexpr uint32_t P0_GROUP = 4;                     // windows a workgroup takes (16 KB of LDS plane: eight workgroups per CU)
constexpr uint32_t P0_WORDS = P0_GROUP * PL_WORDS;   // ... their words of plane 0 (word P0_WORDS: the sink)
struct Plane0Post {
        uint32_t *a, *rd; // the group's plane words; its rank directory (null: none wanted)
        uint32_t pidx;
        __device__ __forceinline__ void doc(const uint32_t rel) {
                const uint32_t r = min(rel, P0_GROUP * PL_W);
                if (rd) {
                        if (rel < P0_GROUP * PL_W)
                                atomicMin(&rd[rel / PL_RANK_DOCS], pidx);
                        ++pidx;
                }
                atomicOr(&a[r >> 5], 1u << (r & 31u));
        }
        __device__ __forceinline__ void operator()(const uint32_t rel, const uint32_t) { doc(rel); }
};
template <int CODEC>
__global__ __launch_bounds__(AND_WG) void k_term_plane0(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last, const uint32_t *__restrict__ blk_off,
                                                        const uint4 *__restrict__ blk_rec, const uint32_t *__restrict__ blk_doff, const uint32_t *__restrict__ win,
                                                        const DevTerm *__restrict__ terms, const uint32_t *__restrict__ build /* (term, row) pairs */,
                                                        uint32_t *__restrict__ planes0, const uint32_t plw, uint32_t *__restrict__ prank /* rank directories, or null */) {
        __shared__ uint32_t pl[P0_WORDS + 32];
        __shared__ uint32_t rdir[P0_GROUP * PL_W / PL_RANK_DOCS];
        const uint32_t tid = threadIdx.x, g = blockIdx.x, row = build[2 * blockIdx.y + 1];
        const uint32_t nwin = plw / PL_WORDS, wfirst = g * P0_GROUP, wn = min(P0_GROUP, nwin - wfirst); // (the last group may be short)
        for (uint32_t i = tid; i < P0_WORDS + 32; i += AND_WG)
                pl[i] = 0;
        if (prank)
                for (uint32_t i = tid; i < P0_GROUP * PL_W / PL_RANK_DOCS; i += AND_WG)
                        rdir[i] = 0xffffffffu;
        const DevTerm t = terms[build[2 * blockIdx.y]];
        const uint32_t *bl = blk_last + t.first_block;
        const uint32_t w0 = wfirst * PL_W, w1 = w0 + wn * PL_W; // (the planner keeps max docID below 2^31: no wrap)
        // rows that can hold documents of [w0, w1): first row whose last docID >= w0 ... first row whose last docID >= w1 (it may still begin inside)
        uint32_t b_lo, b_hi;
        if (t.win_off != 0xffffffffu) {
                b_lo = win[t.win_off + wfirst * PL_CELLS];
                b_hi = win[t.win_off + (wfirst + wn) * PL_CELLS];
        } else {
                uint32_t lo = 0, hi = t.nblocks;
                while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (bl[mid] < w0)
                                lo = mid + 1;
                        else
                                hi = mid;
                }
                b_lo = lo;
                hi = t.nblocks;
                while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (bl[mid] < w1)
                                lo = mid + 1;
                        else
                                hi = mid;
                }
                b_hi = lo;
        }
        b_lo = uni(b_lo);
        b_hi = uni(min(b_hi, t.nblocks - 1));
        __syncthreads();
        if (b_lo < t.nblocks)
                for (uint32_t b = b_lo + tid; b <= b_hi; b += AND_WG) {
                        const uint32_t prev = b ? bl[b - 1] : 0, last = bl[b];
                        Plane0Post post{pl, prank ? rdir : nullptr, 32u * b};
#ifdef TRI_PROF
                        ProfClock prof_;
#endif
                        // (documents beyond the group's last window land in the sink: a short last group's w1 is the docID space's end anyway)
                        if (CODEC == CODEC_LUCENE) {
                                const uint4 rec = blk_rec[t.first_block + b];
                                row_decode<CODEC, false, Plane0Post>(index, t, b, rec.x, rec.y, rec.z, rec.w, TRI_BLOCK_N(t, b, index, 0), prev, last, w0, post PROF_PASS);
                        } else {
                                const uint32_t off = blk_off[t.first_block + b];
                                const uint32_t dlen = blk_doff[t.first_block + b + 1] - blk_doff[t.first_block + b] - 1u;
                                row_decode<CODEC, false, Plane0Post>(index, t, b, off, dlen, 0, 0, TRI_BLOCK_N(t, b, index, off), prev, last, w0, post PROF_PASS);
                        }
                }
        __syncthreads();
        uint32_t *pa = planes0 + (size_t)row * plw + (size_t)wfirst * PL_WORDS;
        for (uint32_t i = tid; i < wn * PL_WORDS; i += AND_WG)
                pa[i] = pl[i];
        if (prank) { // (a 64-byte record per PL_RANK_DOCS documents: the group's first posting, its eight plane-0 words — k_term_planes)
                uint32_t *rec = prank + ((size_t)row * (plw / (PL_RANK_DOCS / 32u)) + (size_t)wfirst * (PL_W / PL_RANK_DOCS)) * PL_RANK_WORDS;
                for (uint32_t i = tid; i < wn * (PL_W / PL_RANK_DOCS) * PL_RANK_WORDS; i += AND_WG) {
                        const uint32_t gg = i / PL_RANK_WORDS, k = i % PL_RANK_WORDS;
                        rec[i] = rank_rec_word(rdir[gg], pl + 8u * gg, k);
                }
        }
}

// ------------------------------------------------------------------------------------------ k_planes
// AccumulatedScoreScheme + top-K of a CNF query (a union, a conjunction of terms / OR-groups, an excluded group, optional scoring
// terms: everything k_fused's CNF instantiations take) in one pass, on BIT PLANES instead of a word per document.  Round 5's shape:
// the query's slots (distinct terms) fall into two kinds, and each kind is worked the way its size asks for.
//   * DENSE slots — head terms with rows in the index's term planes (A: the document holds the term, B: its frequency is not 1, C: nor 2;
//     k_term_planes decoded the list once for every query of every launch).  They are SWEPT: every wave streams its share of the task's
//     docID range through plane A of the dense slots only — the predicate word-wise (a required group = the OR of its slots' A words, the
//     conjunction their AND, the excluded group an AND-NOT, masked documents another: 32 documents per instruction; what
//     docset_spans.cpp:98-173 / 681-790 do per document and docset_iterators.cpp:226-405 per posting), the match count a popcount, and a
//     PRESENCE FILTER (below) that says, word-wise, which documents could reach the current k-th best score at all.  The A words are
//     fetched several sub-windows AHEAD into a register ring; planes B and C are read only by the lanes whose words hold a candidate, and
//     those loads travel while the wave sweeps the next sub-window (a three-stage software pipeline: no round trip is waited for).
//     Round 4 loaded all three planes of all five slots for every word and waited for them: 31 GB per cfg3 launch against 8.9 GB of
//     algorithmic bytes, latency-bound at 4 waves per SIMD.
//   * SPARSE slots — every other term (at most PLK_MAX_SPARSE).  Their rows that can reach the task's range are decoded ONCE per task, one
//     lane per row of <= 32 documents (the same register row readers as k_fused), into sorted lists of docID << 1 | (frequency is not 1)
//     in a scratch region of the workgroup; then EVERY document of those lists is evaluated on its own (phase A): its level in the dense
//     slots from three plane probes each, in the other sparse slots from a bisection of their lists (a hashed byte filter in LDS says
//     which lists can hold it at all), the predicate, the exact score (frequencies beyond the planes' levels from the postings), an
//     offer to the top-K.  The top-K is mostly made of documents that hold the rare terms, so the threshold is near final before the
//     sweep starts — and the sweep never sees a sparse slot: no LDS planes to set and clear per sub-window, no list cursor in its way.
//   * the two phases meet in the match COUNT and in the candidates.  The sweep counts the documents that match through their dense slots
//     alone; a sparse document adds (matches with all its slots) - (matches with its dense slots alone) — +1 where a rare term completes
//     a conjunction, -1 where an excluded rare term removes a match, 0 where it only adds to the score.  A sweep candidate that the
//     hashed filter and a bisection find in a sparse list is dropped there: phase A has scored it, with all its terms.  A query none of
//     whose required groups can be satisfied by dense slots alone (a conjunction with a rare term: cfg3's `A B (C|D|E)`) has no sweep.
//   * the candidate filter.  A dense slot is at one of four LEVELS in a document (absent, frequency 1, frequency 2 — its scorers then add
//     exactly what they add at that frequency —, any other frequency: at most a bound).  Whenever the threshold (the k-th best score so
//     far) moves, planes_filter rebuilds (a) the PRESENCE table — one bit per set of dense slots: do their bounds reach the threshold? —
//     which the sweep evaluates word-wise as a monotone Boolean function of the A words (planes_presence: a tree of v_and_or under
//     scalar masks), (b) the LEVEL table with one bit per level vector and (c) the essential planes (MaxScore's essential slots refined
//     by level), which stage 3 applies to the few documents that passed (a).
//   * candidates are scored one per lane from the level words in registers; a candidate with a slot of unknown frequency that the bound
//     does not rule out waits on the wave's queue, worked off 64 at a time (directory cell -> block -> register row reader).  The waves
//     share the candidate buffer, the threshold and the tables; they meet at a barrier only when the buffer wants pruning, and at the
//     end.  The docID ranges (tasks) of a query share one threshold (planes_prune).
//   * per task: min(matches, k) ranked (docID, score) pairs and the match count; k_topk_merge folds a query's tasks.
constexpr int PLK_WG = 512;
// (PLK_MAX_SPARSE, PLK_NS_SMALL: dev_structs.hpp)
constexpr uint32_t PLK_CAP = 512;       // candidate buffer (one entry per thread when it is pruned)
constexpr uint32_t PLK_PRUNE_AT = 384;  // the waves stop taking candidates once it holds this many: it is pruned to the best k, then they resume
constexpr uint32_t PLK_LV = PL_NESTED;   // a dense slot's top level (the planes' "read it from the postings"); levels 1 .. PLK_LV - 1 are the frequency
constexpr uint32_t PLK_TAB_ND = 5;       // the level table (three bits per dense position) and the presence table cover queries of up to this many dense slots
constexpr uint32_t PLK_FTAB_WORDS = (1u << (3 * PLK_TAB_ND)) / 32; // the level table: one bit per level vector
#ifndef TRI_PLK_WGS
#define TRI_PLK_WGS 2
#endif
constexpr uint32_t PLK_WGS_PER_CU = TRI_PLK_WGS;
constexpr uint32_t PLK_WQ = 128;        // per-wave queue of candidates waiting for a frequency lookup: worked off 64 at a time, every lane busy
constexpr uint32_t PLK_PAD = 0xffffffffu; // list padding (sorts last)
constexpr uint32_t PLK_SW_WORDS = 128;    // a wave's sub-window: two consecutive words (64 documents) per lane ...
constexpr uint32_t PLK_SW = PLK_SW_WORDS * 32; // ... 4096 documents
constexpr uint32_t PLK_CQ = 64 + 2 * 64;   // per-wave queue of candidate words: a batch of 64 plus what one sub-window can add
constexpr uint32_t PLK_HF = 32768;        // hashed filter of the task's sparse documents: one byte per docID & (PLK_HF - 1), bit j = sparse list j may hold it
static_assert(PL_W % PLK_SW == 0, "a task's windows split into whole sub-windows");
static_assert(PLK_CAP == PLK_WG && TOPK_MAX < PLK_PRUNE_AT && PLK_PRUNE_AT < PLK_CAP, "pruning leaves room; a pruned buffer is below the stop mark");
static_assert(PLK_MAX_SPARSE <= 8, "one filter bit per sparse list");

struct PlanesShared {
        double tk_s[PLK_CAP];
        uint32_t tk_d[PLK_CAP];
        DevTerm term[FUS_MAX_SLOTS];
        double wl[FUS_MAX_SLOTS][8]; // per slot and level (= frequency, below PLK_LV): what its scorers add
        double wf[FUS_MAX_SLOTS][8]; // ... rounded up a hair for the filter (it must never lose a tie to rounding), non-decreasing in the level; level PLK_LV: a bound
        double dwf[FUS_MAX_SLOTS][8]; // wf[] of the DENSE slots, by dense position (the sweep's tables are over dense positions)
        double thr_s;
        uint32_t thr_d;
        uint32_t tk_n, tk_full, matches;
        uint32_t top[FUS_MAX_SLOTS];  // per slot: PLK_LV where it has a scorer, else 0
        uint32_t dtop[FUS_MAX_SLOTS], ddocs[FUS_MAX_SLOTS]; // by dense position: top level, the term's documents
        uint32_t esel;                // the essential planes (three bits per dense position — its nested plane 0 .. PL_NESTED - 1, 7: none): every candidate is in one of them
        uint32_t fall;                // 1: no threshold yet (or one that rules nothing out): every match is a candidate
        uint32_t atab;                // the presence table (up to five dense slots): bit `set` <=> the bounds of the slots of `set` reach the threshold
        uint32_t ftab[PLK_FTAB_WORDS]; // the level table: bit `code` (three bits per dense position: its level) set <=> the levels' weights reach the threshold
        uint32_t flag[PLK_WG / 64];
        uint32_t wq[PLK_WG / 64][PLK_WQ][2]; // per wave: candidates waiting for exact frequencies {docID, the slots' levels (three bits per SLOT)}
        uint32_t bcast[4];
        uint32_t sp_row0[FUS_MAX_SLOTS], sp_n[FUS_MAX_SLOTS], sp_base[FUS_MAX_SLOTS]; // sparse slots: first row, rows, first entry of the list in the scratch region
        uint32_t hf[PLK_HF / 4];      // the hashed filter (bytes)
        // the sweep's arguments (planes_sweep_segment: a function of its own, so that its registers are allocated apart from the rest of the kernel's)
        struct SweepArgs {
                uint32_t nd, nreq, negd, leafd, nsp, zrow;
                uint32_t dsl[FUS_MAX_SLOTS], prow[FUS_MAX_SLOTS], gd[FUS_MAX_SLOTS]; // dense position -> slot, -> plane row; required groups as sets of dense positions
                uint32_t sp_off[PLK_MAX_SPARSE], sp_n32[PLK_MAX_SPARSE];            // the sparse lists (a sweep candidate found in one of them is phase A's)
        } sa;
        // ... and each wave's state between two segments: its next sub-window, its share's end, its two queues' fill
        uint32_t w_sw[PLK_WG / 64], w_end[PLK_WG / 64], w_qn[PLK_WG / 64], w_cqn[PLK_WG / 64];
        uint32_t cq[PLK_WG / 64][PLK_CQ][2]; // per wave: words that hold candidates {word of the plane row, candidate bits}, worked off 64 at a time
        DevFused fq;
};
// ONE object for the kernel and the functions it calls: a static __shared__ behind an accessor keeps every access a plain LDS instruction
// (a reference handed to a non-inlined function would arrive as a flat pointer)
__device__ __forceinline__ PlanesShared &plk_shared() {
        __shared__ PlanesShared sh;
        return sh;
}
static_assert(sizeof(PlanesShared) * PLK_WGS_PER_CU <= 160u * 1024u, "the workgroups of a CU share its LDS");

// A row of a sparse slot into its list: 32 entries per row (docID << 1 | frequency-is-not-1), the unused ones of a short last row padded; the
// frequencies themselves go into a parallel array (fout[i] belongs to out[i])
struct ListPost {
        uint32_t *out, *fout;
        uint32_t i = 0;
        __device__ __forceinline__ void doc(const uint32_t rel) {
                fout[i] = 0u;
                out[i++] = rel << 1 | 1u;
        }
        __device__ __forceinline__ void operator()(const uint32_t rel, const uint32_t f) {
                fout[i] = f & 0xffffu;
                out[i++] = rel << 1 | ((f & 0xffffu) != 1u ? 1u : 0u);
        }
};

// A score as an unsigned key of the same order (0: none), for the per-query threshold the tasks of a query share
__device__ __forceinline__ unsigned long long score_key(const double sc) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(sc);
        return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_score(const unsigned long long key) {
        return __longlong_as_double((long long)((key >> 63) ? (key & 0x7fffffffffffffffull) : ~key));
}

// Keep the best k of the n (<= PLK_CAP = PLK_WG) buffered candidates, best first, and move the threshold.  The buffer is SORTED — a bitonic
// network over one element per thread (the order is strict: documents are distinct; empty places rank last): the exchanges at distances below 64
// are lane shuffles inside the waves, only the six at distances 64 / 128 / 256 go through LDS behind a barrier.  (Round 4 ranked by counting:
// every thread compared its element with all n — 5 K instructions per thread and prune, 18 us; a task prunes four to six times.)
// A query cut into several docID ranges (tasks) shares one threshold through *gthr: a task's k-th best score says that k
// documents reach it, so no task needs documents below it — the later and the slower ranges filter with the best k-th score any range
// has seen, not with their own, and cutting a query into ranges costs next to no extra candidates.  (The threshold only ever prunes:
// results do not depend on when a task sees another's.)
__device__ void planes_prune(PlanesShared &sh, const uint32_t n, const uint32_t k, unsigned long long *__restrict__ gthr) {
        const uint32_t tid = threadIdx.x;
        static_assert(PLK_CAP == PLK_WG && PLK_WG == 512, "one element per thread, a network of 512");
        double es = -__builtin_huge_val(); // (an empty place: every candidate is better — scores are finite)
        uint32_t ed = 0xffffffffu;
        if (tid < n) {
                es = sh.tk_s[tid];
                ed = sh.tk_d[tid];
        }
#pragma unroll
        for (uint32_t k2 = 2; k2 <= PLK_WG; k2 <<= 1) {
#pragma unroll
                for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
                        double ps;
                        uint32_t pd;
                        if (j >= 64) { // the partner sits in another wave
                                sh.tk_s[tid] = es;
                                sh.tk_d[tid] = ed;
                                __syncthreads();
                                ps = sh.tk_s[tid ^ j];
                                pd = sh.tk_d[tid ^ j];
                                __syncthreads();
                        } else {
                                ps = __shfl_xor(es, (int)j, 64);
                                pd = __shfl_xor(ed, (int)j, 64);
                        }
                        // an ascending run (best first) keeps the better element at the lower place, a descending one at the higher
                        const bool take_better = ((tid & j) == 0) == ((tid & k2) == 0);
                        const bool swap = take_better ? better(ps, pd, es, ed) : better(es, ed, ps, pd);
                        es = swap ? ps : es;
                        ed = swap ? pd : ed;
                }
        }
        if (tid < k) { // (place tid holds the element of rank tid)
                sh.tk_s[tid] = es;
                sh.tk_d[tid] = ed;
        }
        __syncthreads();
        const uint32_t m = n < k ? n : k;
        double ts = 0.0;
        uint32_t td = 0;
        if (m == k) {
                ts = sh.tk_s[k - 1];
                td = sh.tk_d[k - 1];
        }
        // ONE reading of the shared cell decides the (score, docID) pair every lane filters with: lane 0 publishes this range's k-th best and
        // takes what the cell then holds, wave 0 derives the pair from that single value and stores it, the barrier hands it to the other
        // waves.  (Each lane reading the cell for itself — other ranges keep raising it — let two waves disagree on whose threshold holds and
        // leave another range's score next to this range's docID: documents tied with that score and above the docID were then dropped.)
        if (tid < 64) { // (wave-uniform)
                unsigned long long g = 0ull;
                if (tid == 0) {
                        const unsigned long long mine = m == k ? score_key(ts) : 0ull;
                        const unsigned long long old = mine ? atomicMax(gthr, mine) : __atomic_load_n(gthr, __ATOMIC_RELAXED);
                        g = old > mine ? old : mine;
                }
                g = ((unsigned long long)uni((uint32_t)(g >> 32)) << 32) | uni((uint32_t)g);
                const bool other = g != 0ull && (m < k || ts < key_score(g)); // another range's k-th best is the higher one: ties pass (no docID to break them with)
                // uniform stores by the lanes of wave 0
                sh.tk_n = m;
                if (m == k || other) {
                        sh.tk_full = 1;
                        sh.thr_s = other ? key_score(g) : ts;
                        sh.thr_d = other ? 0xffffffffu : td;
                }
        }
        __syncthreads();
}

// The candidate filter of the SWEEP, recomputed whenever the threshold moves (every thread calls it; it ends with a barrier).  It speaks of
// the nd DENSE slots by their dense position (a document the sweep may offer holds no sparse term: phase A owns those).  A slot is at a level
// 0 .. dtop[i] in a document and adds at most dwf[i][level] there (exactly, below the top level), so a document's score is at most the sum of
// its slots' level weights:
//   * the LEVEL table (nd <= PLK_TAB_ND) holds, for every level vector (three bits per position), whether that sum reaches the current k-th
//     best score — a document whose frequencies the planes tell (almost all) is thereby tested against its EXACT score without being touched;
//   * the PRESENCE table (nd <= PLK_TAB_ND) holds the same for the sets of slots a document may hold, every slot at its top level's bound: what
//     the sweep evaluates word-wise on plane A alone;
//   * the essential planes: with the slots ordered by their bound, the longest prefix whose bounds sum to less than the threshold cannot lift a
//     document over it, so a candidate holds one of the OTHER slots; what is left of the threshold then buys those slots' LOW levels (a slot
//     capped at level c is essential only through its plane c: frequency > c).  The sweep fetches that plane beside plane A.
// No threshold yet, or one that rules nothing out: every match is a candidate.
__device__ void planes_filter(PlanesShared &sh, const uint32_t nd) {
        const uint32_t tid = threadIdx.x;
        const double thr = sh.thr_s;
        const bool full = uni(sh.tk_full) != 0 && 0.0 < thr;
        sh.fall = full ? 0u : 1u; // (uniform stores)
        sh.esel = 0; // (plane A of every slot)
        if (!full) {
                sh.atab = 0xffffffffu;
                __syncthreads();
                return;
        }
        if (tid < 64) { // (wave 0; lanes 32 .. 63 mirror): the presence table — same summation order as the level table, every addend at least the level's
                const uint32_t set = tid & 31u;
                double sum = 0.0;
                for (uint32_t i = 0; i < nd && i < PLK_TAB_ND; ++i)
                        sum += ((set >> i) & 1u) && sh.dtop[i] ? sh.dwf[i][sh.dtop[i]] : 0.0;
                const uint64_t bm = __builtin_amdgcn_ballot_w64(!(sum < thr));
                sh.atab = (uint32_t)bm; // (same value from every lane)
        }
        if (nd <= PLK_TAB_ND) {
                const uint32_t codes = 1u << (3 * nd), words = codes / 32u > 0 ? codes / 32u : 1u;
                for (uint32_t wd = tid; wd < words; wd += PLK_WG) {
                        uint32_t bits = 0;
                        for (uint32_t j = 0; j < 32; ++j) {
                                const uint32_t code = wd * 32u + j;
                                double sum = 0.0;
                                bool valid = code < codes;
                                for (uint32_t i = 0; i < nd; ++i) {
                                        const uint32_t l = (code >> (3 * i)) & 7u;
                                        valid &= l <= sh.dtop[i];
                                        sum += l ? sh.dwf[i][l & 7u] : 0.0;
                                }
                                bits |= (valid && !(sum < thr) ? 1u : 0u) << j;
                        }
                        sh.ftab[wd] = bits;
                }
        }
        // the essential planes (same values in every lane): whole slots first, by ascending bound ...
        const double thr_lo = thr * (1.0 - 1e-9); // (the table adds the weights in its own order: a hair of room for the rounding)
        uint32_t done = 0, ess = 0;
        double p = 0.0, spent = 0.0;
        for (uint32_t r = 0; r < nd; ++r) { // selection by ascending bound (<= 8 slots)
                uint32_t best = 0;
                double bv = 1e300;
                for (uint32_t i = 0; i < nd; ++i) {
                        const double b = sh.dtop[i] ? sh.dwf[i][sh.dtop[i]] : 0.0;
                        if (!((done >> i) & 1u) && b < bv) {
                                bv = b;
                                best = i;
                        }
                }
                done |= 1u << best;
                p += bv;
                if (!(p < thr_lo))
                        ess |= 1u << best;
                else
                        spent = p;
        }
        // ... then what is left of the threshold buys the essential slots' LOW levels: a slot capped at level c is essential only through its plane
        // c (frequency > c: a fraction of the plane below).  Each round takes the raise that drops the most documents (estimated from the terms'
        // document counts and a frequency's usual share) among those that still fit.
        uint32_t cap[FUS_MAX_SLOTS];
        for (uint32_t i = 0; i < FUS_MAX_SLOTS; ++i)
                cap[i] = i < nd && ((ess >> i) & 1u) ? 0u : 7u;
        for (uint32_t r = 0; r < PLK_LV * FUS_MAX_SLOTS; ++r) {
                uint32_t best = 0xffffffffu;
                double gain = 0.0, cost = 0.0;
                for (uint32_t i = 0; i < nd; ++i) {
                        const uint32_t c = cap[i], tp = sh.dtop[i];
                        if (c >= tp) // (its top plane already, or not essential at all)
                                continue;
                        const double dw = sh.dwf[i][(c + 1) & 7u] - (c ? sh.dwf[i][c & 7u] : 0.0);
                        const double share = c == 0 ? 0.65 : c == 1 ? 0.2 : c == 2 ? 0.1 : c == 3 ? 0.03 : c == 4 ? 0.015 : 0.005; // (the documents at exactly level c + 1, roughly)
                        const double docs = (double)sh.ddocs[i] * share;
                        if (spent + dw < thr_lo && gain < docs) {
                                gain = docs;
                                cost = dw;
                                best = i;
                        }
                }
                if (best == 0xffffffffu)
                        break;
                cap[best] += 1;
                spent += cost;
        }
        uint32_t esel = 0;
        for (uint32_t i = 0; i < FUS_MAX_SLOTS; ++i) {
                const uint32_t c = cap[i], tp = i < nd ? sh.dtop[i] : 0u;
                esel |= (c + 1 > tp ? 7u : c) << (3 * i);
        }
        sh.esel = uni(esel);
        __syncthreads();
}

// The frequency of `doc` in term t (the document is known to be one of the term's): the directory cell brackets the block, one round
// of independent loads (after a bisection down to 16 blocks, if need be) finds it, and the row is read by the same register readers as
// everywhere else (row_decode) with a probe for the one document — four memory round trips in all.
struct FreqProbe {
        uint32_t target, f = 0;
        __device__ __forceinline__ void doc(const uint32_t) {}
        __device__ __forceinline__ void operator()(const uint32_t rel, const uint32_t fr) { f = rel == target ? fr : f; }
};
template <int CODEC>
__device__ __noinline__ uint32_t planes_lookup_freq(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last, const uint32_t *__restrict__ blk_off,
                                                    const uint4 *__restrict__ blk_rec, const uint32_t *__restrict__ blk_doff, const uint32_t *__restrict__ win,
                                                    const DevTerm &t, const uint32_t doc) {
        const uint32_t *bl = blk_last + t.first_block;
        uint32_t lo = 0, hi = t.nblocks - 1; // the first block whose last docID >= doc lies in [lo, hi]
        if (t.win_off != 0xffffffffu) {
                lo = win[t.win_off + (doc >> CELL_LOG2)];
                hi = min(win[t.win_off + (doc >> CELL_LOG2) + 1], t.nblocks - 1);
        }
        while (hi - lo > 16) {
                const uint32_t mid = (lo + hi) >> 1;
                if (bl[mid] < doc)
                        lo = mid + 1;
                else
                        hi = mid;
        }
        uint32_t below = 0;
#pragma unroll
        for (uint32_t i = 0; i < 16; ++i) // (independent loads: one round trip)
                below += (lo + i < hi && bl[lo + i] < doc) ? 1u : 0u;
        const uint32_t b = lo + below;
        const uint32_t prev = b ? bl[b - 1] : 0, last = bl[b];
        FreqProbe probe{doc};
#ifdef TRI_PROF
        ProfClock prof_;
#endif
        if (CODEC == CODEC_LUCENE) {
                const uint4 rec = blk_rec[t.first_block + b];
                row_decode<CODEC, true, FreqProbe>(index, t, b, rec.x, rec.y, rec.z, rec.w, TRI_BLOCK_N(t, b, index, 0), prev, last, 0u, probe PROF_PASS);
        } else {
                const uint32_t off = blk_off[t.first_block + b];
                const uint32_t dlen = blk_doff[t.first_block + b + 1] - blk_doff[t.first_block + b] - 1u;
                row_decode<CODEC, true, FreqProbe>(index, t, b, off, dlen, 0, 0, TRI_BLOCK_N(t, b, index, off), prev, last, 0u, probe PROF_PASS);
        }
        return probe.f & 0xffffu;
}

// One row of a decoded slot into its list (out of line: the row readers' registers must not weigh on the window loop)
template <int CODEC>
__device__ __noinline__ void planes_list_row(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last, const uint32_t *__restrict__ blk_off,
                                             const uint4 *__restrict__ blk_rec, const uint32_t *__restrict__ blk_doff, const DevTerm &t, const uint32_t b,
                                             uint32_t *__restrict__ out, uint32_t *__restrict__ fout) {
        const uint32_t *bl = blk_last + t.first_block;
        const uint32_t prev = b ? bl[b - 1] : 0, last = bl[b];
        ListPost post{out, fout};
#ifdef TRI_PROF
        ProfClock prof_;
#endif
        if (CODEC == CODEC_LUCENE) {
                const uint4 rec = blk_rec[t.first_block + b];
                row_decode<CODEC, true, ListPost>(index, t, b, rec.x, rec.y, rec.z, rec.w, TRI_BLOCK_N(t, b, index, 0), prev, last, 0u, post PROF_PASS);
        } else {
                const uint32_t off = blk_off[t.first_block + b];
                const uint32_t dlen = blk_doff[t.first_block + b + 1] - blk_doff[t.first_block + b] - 1u;
                row_decode<CODEC, true, ListPost>(index, t, b, off, dlen, 0, 0, TRI_BLOCK_N(t, b, index, off), prev, last, 0u, post PROF_PASS);
        }
        for (uint32_t i = post.i; i < 32; ++i)
                out[i] = PLK_PAD;
}

// A uniform bit as an all-ones / all-zeros scalar mask, and (a & mask) | x in one vector instruction (the compiler turns the plain
// expression into a scalar select plus two vector instructions).
template <uint32_t POS> __device__ __forceinline__ uint32_t umask_at(const uint32_t bits) {
        // (volatile: made where it is used — hoisted out of the window loop, the dozens of masks of a query spill to vector lanes and come back
        //  through v_readlane, a vector instruction each)
        uint32_t r;
        const uint32_t sbits = (uint32_t)__builtin_amdgcn_readfirstlane((int)bits); // (folded away where the compiler knows the value to be uniform)
        asm volatile("s_bfe_i32 %0, %1, %2" : "=s"(r) : "s"(sbits), "n"(POS | 0x10000u) : "scc");
        return r;
}
template <uint32_t N, typename F> __device__ __forceinline__ void static_for(F &&f) { // f(integral_constant 0) ... f(integral_constant N - 1)
        if constexpr (N > 0) {
                static_for<N - 1>(f);
                f(std::integral_constant<uint32_t, N - 1>{});
        }
}
__device__ __forceinline__ uint32_t and_or(const uint32_t a, const uint32_t smask, const uint32_t x) {
        uint32_t r;
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(smask), "v"(x));
        return r;
}

// The presence filter on TWO words at once (p, q: the dense slots' plane-A words): bit d of the result <=> the bounds of the slots document d
// holds reach the threshold, i.e. the table T (PlanesShared::atab) looked up with the document's NV presence bits — for 32 documents per
// word.  T is monotone (a superset of slots never scores less), so f = OR over the sets T passes of the AND of their slots' words; factored
// as a tree over the slots with T's bits as uniform scalar masks (s_bfe_i32, made where they are used).  planes_presence_tree<NV, BASE>
// leaves out the constant part T[BASE] (the set without any of the NV slots): the caller ORs it in one level up, where it costs one and_or
// under the parent's slot word — 4 vector instructions per word for two slots, 10 for three, 46 for five, whatever the table holds.
template <uint32_t NV, uint32_t BASE, uint32_t NP>
__device__ __forceinline__ void planes_presence_tree(const uint32_t T, const uint32_t (&p)[NP], const uint32_t (&q)[NP], uint32_t &fp, uint32_t &fq) {
        if constexpr (NV == 0) {
                fp = 0;
                fq = 0;
        } else {
                uint32_t lp, lq, hp, hq;
                planes_presence_tree<NV - 1, BASE, NP>(T, p, q, lp, lq);
                planes_presence_tree<NV - 1, BASE + (1u << (NV - 1)), NP>(T, p, q, hp, hq);
                const uint32_t c = umask_at<BASE + (1u << (NV - 1))>(T); // (the slot alone completes the set)
                fp = and_or(p[NV - 1], c, (p[NV - 1] & hp) | lp);
                fq = and_or(q[NV - 1], c, (q[NV - 1] & hq) | lq);
        }
}

// pointers into global memory, said so: a function that is not a kernel sees its pointer arguments as generic, and every access through them
// becomes a flat instruction (both memory counters, a vector address)
typedef const __attribute__((address_space(1))) uint32_t *PlkG1;
typedef uint32_t PlkU2 __attribute__((ext_vector_type(2))); // (a built-in vector: loads through an address-space pointer need no operator=)
typedef const __attribute__((address_space(1))) PlkU2 *PlkG2;
// where `doc` stands in a sorted list of n entries (docID << 1 | flag; padding sorts last), PLK_NOT_FOUND when the list does not hold it
constexpr uint32_t PLK_NOT_FOUND = 0xffffffffu;
template <typename P> __device__ __forceinline__ uint32_t planes_list_find(const P ls, const uint32_t n, const uint32_t doc) {
        uint32_t lo = 0, hi = n;
        const uint32_t key = doc << 1;
        while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (ls[mid] < key)
                        lo = mid + 1;
                else
                        hi = mid;
        }
        const uint32_t f = lo < n ? ls[lo] : PLK_PAD;
        return (f >> 1) == doc ? lo : PLK_NOT_FOUND;
}

// how far ahead the sweep fetches plane A: ND slots x two words per lane and sub-window in flight per step of the ring (about a kilobyte
// per wave and sub-window and slot; the CU needs some tens of kilobytes in flight to cover HBM's latency at its share of the bandwidth)
template <int ND> struct PlkRing { static constexpr uint32_t PF = ND <= 1 ? 4 : ND == 2 ? 3 : ND == 3 ? 2 : 1; };

// ---- PHASE B of k_planes, one SEGMENT of one wave: the sweep over the dense slots.  The wave walks its share of the task's range a sub-window of
//      PLK_SW documents (two consecutive words per lane) at a time, wave-synchronously — no workgroup barrier inside:
//      * plane A of the next PF sub-windows is on its way into a register ring while the sub-window whose words have arrived is swept:
//        predicate, count, presence filter.  These are the ONLY loads of the loop, consumed in the order they were issued: the wave never
//        waits for the load it has just sent.
//      * the words that hold a candidate (a few lanes of a sub-window, most sub-windows) go on the wave's queue of candidate WORDS in LDS —
//        {word of the plane row, candidate bits} —; when 64 wait, every lane takes one: the A / B / C words of all the dense slots for that
//        word in one round trip, the essential planes, the level table per document, the survivors scored — 64 lanes busy on what one or two
//        lanes per sub-window would otherwise do behind a round trip of their own.
//      A function of its own (not inlined): its registers — the ring, a dozen scalar masks — are allocated apart from the rest of the
//      kernel's (inlined, the compiler spilled the ring to scratch behind every load: a wait per load).  Its arguments and the wave's state
//      between two segments travel through LDS (PlanesShared::sa, w_*).  Returns 0 when the wave's share is done, 1 when it stopped for the
//      frequency queue (64 entries wait) or for the candidate buffer (it wants pruning: the waves meet).
template <typename T> __device__ __forceinline__ T *uni_ptr(T *p) { // a pointer every lane holds alike, as a scalar (a function's arguments arrive in vector registers)
        const uint64_t v = (uint64_t)p;
        return (T *)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
}
// ---- a batch of candidate WORDS of one wave (see planes_sweep_segment): the last (up to) 64 of its queue, one per lane — the A / B / C words
//      of all the dense slots for that word in one round trip, the essential planes, the level table per document, the survivors scored.
//      `state` in and out: the frequency queue's fill (bits 0..15), the word queue's (16..30); bit 31 out: it stopped short (the frequency
//      queue or the candidate buffer is full) — what is left of the words' candidates is back on the word queue.  Not inlined: it runs once
//      per some forty sub-windows, and its registers must not weigh on the sweep's loop.
template <int ND_>
__device__ __noinline__ uint32_t planes_work_words(const uint32_t *__restrict__ planes_hi_, const uint32_t plw_, const uint32_t *__restrict__ lists_, const uint32_t state) {
        constexpr uint32_t ND = (uint32_t)ND_;
        PlanesShared &sh = plk_shared();
        const uint32_t lane = threadIdx.x & 63u, wave = uni(threadIdx.x >> 6);
        const uint32_t *const planes_hi = uni_ptr(planes_hi_), *const lists = uni_ptr(lists_);
        const uint32_t plw = uni(plw_);
        uint32_t qn = uni(state) & 0xffffu, cqn = (uni(state) >> 16) & 0x7fffu;
        const uint32_t nd = uni(sh.sa.nd), leafd = uni(sh.sa.leafd), nsp = uni(sh.sa.nsp);
        uint32_t dsl[ND];
        PlkG1 ph1[ND]; // the dense positions' level words
#pragma unroll
        for (uint32_t i = 0; i < ND; ++i) {
                dsl[i] = uni(sh.sa.dsl[i]);
                ph1[i] = (PlkG1)(planes_hi + (size_t)uni(sh.sa.prow[i]) * PL_HI * plw + (size_t)PL_HI_LEVELS * plw);
        }
        const PlkG1 lists1 = (PlkG1)lists;
        const bool full = uni(sh.tk_full) != 0;
        const double thr_s = sh.thr_s;
        const uint32_t thr_d = sh.thr_d;
        const uint32_t esel = uni(sh.esel);
        const bool fall = uni(sh.fall) != 0;
        auto offer = [&](const double sc, const uint32_t doc) __attribute__((always_inline)) { // false: no room (the buffer wants pruning)
                const uint32_t slot = atomicAdd(&sh.tk_n, 1u);
                if (slot >= PLK_CAP)
                        return false;
                sh.tk_s[slot] = sc;
                sh.tk_d[slot] = doc;
                return true;
        };
        const uint32_t take_n = min(cqn, 64u), base = cqn - take_n;
        cqn = base;
        const bool mine = lane < take_n;
        const uint32_t wi = mine ? sh.cq[wave][base + lane][0] : lane; // (a lane without a word reads the row's first words and has no candidates)
        uint32_t cw = mine ? sh.cq[wave][base + lane][1] : 0u;
        // the word's level in every dense slot, bit-sliced: the planes are nested, the level is the number of them a document is in — its three
        // bits word-wise (l0: the level is odd; l1: it is 2, 3 or 6; l2: it is 4 or more), and the slot's essential plane on the way
        uint32_t l0[ND], l1[ND], l2[ND], ew = 0;
        static_assert(PL_NESTED == 6 && PL_LEVEL_WORDS == 3, "the planes' words below are derived from three level bits of six levels");
#pragma unroll
        for (uint32_t i = 0; i < ND; ++i) { // (all the loads first: one round trip — three adjacent words per slot)
                const PlkG1 lv = ph1[i] + 3u * (i < nd ? wi : lane);
                l0[i] = lv[0];
                l1[i] = lv[1];
                l2[i] = lv[2];
        }
#pragma unroll
        for (uint32_t i = 0; i < ND; ++i) {
                // the slot's essential plane (nested plane e: the level is above e) from the level's bits
                const uint32_t e = uni((esel >> (3 * i)) & 7u);
                const uint32_t x0 = l0[i], x1 = l1[i], x2 = l2[i];
                ew |= e == 0u ? x0 | x1 | x2 : e == 1u ? x1 | x2 : e == 2u ? x2 | (x1 & x0) : e == 3u ? x2 : e == 4u ? x2 & (x0 | x1) : e == 5u ? x2 & x1 : 0u; // (uniform selects)
                if (!((leafd >> i) & 1u)) // (a slot without a scorer: level 0)
                        l0[i] = l1[i] = l2[i] = 0u;
        }
        if (!fall) {
                // the documents in an essential plane; each of them then with its level vector in the table (queries of up to PLK_TAB_ND dense slots)
                ew &= cw;
                if constexpr (ND <= PLK_TAB_ND) {
                        cw = 0;
                        while (__builtin_amdgcn_ballot_w64(ew != 0) != 0ull) {
                                const uint32_t bit = ew ? (uint32_t)__builtin_ctz(ew) : 0u;
                                uint32_t code = 0;
#pragma unroll
                                for (uint32_t i = 0; i < ND; ++i)
                                        code |= (((l0[i] >> bit) & 1u) << (3 * i)) | (((l1[i] >> bit) & 1u) << (3 * i + 1)) | (((l2[i] >> bit) & 1u) << (3 * i + 2));
                                const uint32_t hit = (sh.ftab[code >> 5] >> (code & 31u)) & 1u;
                                cw |= ew ? hit << bit : 0u;
                                ew &= ew - 1u;
                                PROF_COUNT(20, lane == 0 ? 1 : 0);
                        }
                } else
                        cw = ew;
        }
        // One step over the words' candidates: every lane that has one takes its lowest, scores it from its levels — exactly where the planes tell
        // the frequencies, with a bound where they do not — and offers it, queues it (a frequency to be read from the postings that the bound
        // does not rule out) or drops it.  A document of a sparse list is dropped: phase A has scored it.  offer false: no room (the candidate stays).
        bool short_ = false;
        while (__builtin_amdgcn_ballot_w64(cw != 0) != 0ull) {
                if (qn >= 64 || uni(__atomic_load_n(&sh.tk_n, __ATOMIC_RELAXED)) >= PLK_PRUNE_AT) {
                        short_ = true;
                        PROF_COUNT(23, lane == 0 ? 1 : 0);
                        break;
                }
                bool enq = false;
                uint32_t edoc = 0, elev = 0;
                if (cw) {
                        const uint32_t bit = (uint32_t)__builtin_ctz(cw);
                        const uint32_t doc = 32u * wi + bit;
                        const uint32_t hb = (sh.hf[(doc & (PLK_HF - 1u)) >> 2] >> (8u * (doc & 3u))) & 0xffu;
                        bool sparse_doc = false;
                        if (hb) {
                                for (uint32_t j = 0; j < nsp; ++j)
                                        if ((hb >> j) & 1u)
                                                sparse_doc = sparse_doc || planes_list_find(lists1 + sh.sa.sp_off[j], sh.sa.sp_n32[j], doc) != PLK_NOT_FOUND;
                        }
                        bool done = true;
                        if (!sparse_doc) {
                                double sk = 0.0, sb = 0.0; // the known part of the score; bounds of the slots whose frequency the planes do not tell
                                uint32_t levels = 0;
                                bool unk = false;
#pragma unroll
                                for (uint32_t i = 0; i < ND; ++i) {
                                        const uint32_t l = ((l0[i] >> bit) & 1u) | (((l1[i] >> bit) & 1u) << 1) | (((l2[i] >> bit) & 1u) << 2);
                                        if (!l)
                                                continue;
                                        levels |= l << (3u * dsl[i]);
                                        if (l < PLK_LV)
                                                sk += sh.wl[dsl[i]][l];
                                        else {
                                                unk = true;
                                                sb += sh.dwf[i][PLK_LV];
                                        }
                                }
                                if (!full || better(sk + sb, doc, thr_s, thr_d)) {
                                        if (unk) {
                                                enq = true;
                                                edoc = doc;
                                                elev = levels;
                                        } else
                                                done = offer(sk, doc); // (no room: the candidate stays for after the prune)
                                }
                        }
                        if (done)
                                cw &= cw - 1u;
                }
                PROF_COUNT(16, lane == 0 ? 1 : 0);
                const uint64_t em = __builtin_amdgcn_ballot_w64(enq);
                if (enq) {
                        const uint32_t at = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
                        sh.wq[wave][at][0] = edoc;
                        sh.wq[wave][at][1] = elev;
                }
                qn += (uint32_t)__popcll(em);
        }
        if (short_) { // the words that still hold candidates go back (the level table has been through them: only survivors are left)
                const uint64_t bm = __builtin_amdgcn_ballot_w64(cw != 0);
                if (cw) {
                        const uint32_t at = cqn + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                        sh.cq[wave][at][0] = wi;
                        sh.cq[wave][at][1] = cw;
                }
                cqn += (uint32_t)__popcll(bm);
        }
        PROF_COUNT(21, lane == 0 ? 1 : 0);
        return qn | cqn << 16 | (short_ ? 0x80000000u : 0u);
}

template <int ND_>
__device__ __noinline__ uint32_t planes_sweep_segment(const uint32_t *__restrict__ planes0_, const uint32_t *__restrict__ planes_hi_, const uint32_t plw_,
                                                      const uint32_t *__restrict__ masked_, const uint32_t *__restrict__ lists_) {
        constexpr uint32_t ND = (uint32_t)ND_;
        constexpr uint32_t PF = PlkRing<ND_>::PF;
        PlanesShared &sh = plk_shared();
        const uint32_t lane = threadIdx.x & 63u, wave = uni(threadIdx.x >> 6);
        const uint32_t *const planes0 = uni_ptr(planes0_), *const planes_hi = uni_ptr(planes_hi_), *const masked = uni_ptr(masked_), *const lists = uni_ptr(lists_);
        const uint32_t plw = uni(plw_);
        const uint32_t nd = uni(sh.sa.nd), nreq = uni(sh.sa.nreq), negd = uni(sh.sa.negd);
        uint32_t sw = uni(sh.w_sw[wave]), qn = uni(sh.w_qn[wave]), cqn = uni(sh.w_cqn[wave]);
        const uint32_t sw_end = uni(sh.w_end[wave]);
        constexpr uint32_t SWS = PLK_WG / 64; // the waves' sub-windows interleave: wave w sweeps first + w, first + w + 8, ... — whatever the candidates' density
                                              // does along the range, every wave gets its share of it, and the eight of them read each plane row front to back together
        const uint32_t sw_last = sw_end - 1u - (sw_end - 1u - sw) % SWS; // this wave's last sub-window (sw < sw_end here, or nothing is fetched)
        PlkG1 pa1[ND]; // plane A of the dense positions (a position beyond nd: the all-zero row)
#pragma unroll
        for (uint32_t i = 0; i < ND; ++i)
                pa1[i] = (PlkG1)(planes0 + (size_t)uni(sh.sa.prow[i]) * plw);
        // the slots' ESSENTIAL planes (planes_filter: every candidate is in one of them).  Plane A is in the ring anyway; a slot that is essential through
        // a higher plane (a frequency above some level) has that plane's words fetched beside it (any other position fetches the all-zero row's first words: one cache line)
        PlkG1 pe1[ND];
        uint32_t es_fetch = 0; // the positions whose essential plane is one above plane A
#pragma unroll
        for (uint32_t i = 0; i < ND; ++i) {
                const uint32_t e = (uni(sh.esel) >> (3 * i)) & 7u;
#ifdef TRI_PLK_NO_EFETCH
                const bool bc = false;
#else
                const bool bc = i < nd && e >= 1u && e < PL_NESTED;
                const uint32_t es = e < PL_STORED ? e : PL_STORED - 1u; // (a plane the rows do not hold: the highest one they do — a superset)
#endif
                es_fetch |= (bc ? 1u : 0u) << i;
                pe1[i] = bc ? (PlkG1)(planes_hi + (size_t)uni(sh.sa.prow[i]) * PL_HI * plw + (size_t)(es - 1u) * plw) : (PlkG1)(planes0 + (size_t)uni(sh.sa.zrow) * plw); // (bc: es >= 1)
        }
        es_fetch = uni(es_fetch);
        // the first three required groups in scalar registers (a group beyond the query's: every position — it changes nothing), further ones from LDS
        constexpr uint32_t GREG = 3;
        uint32_t gd[GREG];
#pragma unroll
        for (uint32_t g = 0; g < GREG; ++g)
                gd[g] = g < nreq ? uni(sh.sa.gd[g]) : 0xffffffffu;
        const PlkG2 mk2 = (PlkG2)(masked ? masked : planes0 + (size_t)uni(sh.sa.zrow) * plw); // (rows and sub-windows are multiples of 128 words: 8-byte aligned)
        uint32_t my_matches = 0;
        // (threshold and tables move only at a prune, i.e. between two segments)
        const uint32_t esel = uni(sh.esel), atab = uni(sh.atab);
        uint32_t es_a = 0, es_any = 0; // the positions that are essential through plane A; through any plane
#pragma unroll
        for (uint32_t i = 0; i < ND; ++i) {
                const uint32_t e = (esel >> (3 * i)) & 7u;
#ifdef TRI_PLK_NO_EFETCH
                es_a |= (e != 7u ? 1u : 0u) << i;
#else
                es_a |= (e == 0 ? 1u : 0u) << i;
#endif
                es_any |= (e != 7u ? 1u : 0u) << i;
        }
        es_a = uni(es_a), es_any = uni(es_any);
        (void)es_any, (void)atab;
        const bool fall = uni(sh.fall) != 0;
        PlkU2 ring[PF][ND], ringe[PF][ND], ringm[PF];
        // (every fetch issues the same ND + 1 loads — a share's last sub-windows fetch its last one again, a segment without masked documents reads
        //  the all-zero row for them — so that the compiler can COUNT the loads in flight: it then waits for the ring's oldest place only, not for
        //  the places it has just sent for)
        auto fetch_a = [&](const uint32_t sw_, PlkU2 (&g)[ND], PlkU2 (&ge)[ND], PlkU2 &gm) __attribute__((always_inline)) {
                const uint32_t wb = min(sw_, sw_last) * (PLK_SW_WORDS / 2) + lane;
#pragma unroll
                for (uint32_t i = 0; i < ND; ++i) {
                        g[i] = ((PlkG2)pa1[i])[i < nd ? wb : lane];
                        ge[i] = ((PlkG2)pe1[i])[((es_fetch >> i) & 1u) ? wb : lane];
                }
                gm = mk2[masked ? wb : lane];
        };
        bool stop = false; // (uniform, like sw / qn / cqn: the segment's control flow is the wave's)
        // words left over from the previous segment first (they were cut short by a full queue or buffer)
        auto work_words = [&]() __attribute__((always_inline)) { // true: the batch went through
                const uint32_t st = uni(planes_work_words<ND_>(planes_hi, plw, lists, qn | cqn << 16));
                qn = st & 0xffffu, cqn = (st >> 16) & 0x7fffu;
                return (st >> 31) == 0u;
        };
        while (!stop && cqn >= 64)
                stop = !work_words();
        if (!stop && sw < sw_end) {
#pragma unroll
                for (uint32_t r = 0; r < PF; ++r) // fill the ring: the next PF sub-windows' A words
                        fetch_a(sw + r * SWS, ring[r], ringe[r], ringm[r]);
        }
        while (!stop && sw < sw_end) {
                static_for<PF>([&](auto R) __attribute__((always_inline)) {
                        constexpr uint32_t r = decltype(R)::value;
                        sw = uni(sw), cqn = uni(cqn);
                        if (stop || !(sw < sw_end)) // (uniform)
                                return;
                        if (uni(__atomic_load_n(&sh.tk_n, __ATOMIC_RELAXED)) >= PLK_PRUNE_AT) {
                                stop = true; // the buffer wants pruning first (the sub-window stays as it is)
                                return;
                        }
                        uint32_t p[ND], qq[ND];
#pragma unroll
                        for (uint32_t i = 0; i < ND; ++i) {
                                p[i] = ring[r][i].x;
                                qq[i] = ring[r][i].y;
                        }
                        uint32_t e0 = 0, e1 = 0; // the essential planes' words: the higher planes as fetched ...
#pragma unroll
                        for (uint32_t i = 0; i < ND; ++i) {
                                e0 |= ringe[r][i].x;
                                e1 |= ringe[r][i].y;
                        }
                        const PlkU2 mk = ringm[r];
                        // ... and the ring's place goes to the sub-window PF ahead
                        fetch_a(sw + PF * SWS, ring[r], ringe[r], ringm[r]);
                        // the predicate word-wise (the slots' roles are uniform bit sets: a role's word is and_or'ed together under scalar masks — one vector
                        // instruction per slot, word and role), the match count
                        uint32_t m0 = 0xffffffffu, m1 = 0xffffffffu;
                        static_for<GREG>([&](auto G) __attribute__((always_inline)) {
                                constexpr uint32_t g = decltype(G)::value;
                                if (g >= nreq) // (uniform)
                                        return;
                                const uint32_t gmask = gd[g];
                                uint32_t x0 = 0, x1 = 0;
                                static_for<ND>([&](auto I) __attribute__((always_inline)) {
                                        const uint32_t mk_ = umask_at<I>(gmask);
                                        x0 = and_or(p[I], mk_, x0);
                                        x1 = and_or(qq[I], mk_, x1);
                                });
                                m0 &= x0;
                                m1 &= x1;
                        });
                        for (uint32_t g = GREG; g < nreq; ++g) { // (rare: a conjunction of more than three groups that dense slots can satisfy)
                                const uint32_t gmask = uni(sh.sa.gd[g]);
                                uint32_t x0 = 0, x1 = 0;
                                static_for<ND>([&](auto I) __attribute__((always_inline)) {
                                        const uint32_t mk_ = umask_at<I>(gmask);
                                        x0 = and_or(p[I], mk_, x0);
                                        x1 = and_or(qq[I], mk_, x1);
                                });
                                m0 &= x0;
                                m1 &= x1;
                        }
                        if (negd) {
                                uint32_t n0 = 0, n1 = 0;
                                static_for<ND>([&](auto I) __attribute__((always_inline)) {
                                        const uint32_t mk_ = umask_at<I>(negd);
                                        n0 = and_or(p[I], mk_, n0);
                                        n1 = and_or(qq[I], mk_, n1);
                                });
                                m0 &= ~n0;
                                m1 &= ~n1;
                        }
                        m0 &= ~mk.x; // masked_documents_registry::test (docidupdates.h:90-119)
                        m1 &= ~mk.y;
                        my_matches += (uint32_t)__popc(m0) + (uint32_t)__popc(m1);
                        uint32_t c0 = m0, c1 = m1;
                        if (!fall) {
                                // the presence filter: can the slots a document HOLDS reach the threshold at all?  Up to five dense slots: exactly (the
                                // table looked up word-wise); more: MaxScore's essential slots — a candidate holds one of them
                                uint32_t f0 = 0, f1 = 0;
                                if constexpr (ND <= PLK_TAB_ND) {
                                        planes_presence_tree<ND, 0, ND>(atab, p, qq, f0, f1);
                                        const uint32_t cz = umask_at<0>(atab); // (a threshold nothing is needed for)
                                        f0 |= cz;
                                        f1 |= cz;
                                } else {
                                        static_for<ND>([&](auto I) __attribute__((always_inline)) {
                                                const uint32_t mk_ = umask_at<I>(es_any);
                                                f0 = and_or(p[I], mk_, f0);
                                                f1 = and_or(qq[I], mk_, f1);
                                        });
                                }
                                static_for<ND>([&](auto I) __attribute__((always_inline)) { // ... A from the ring
                                        const uint32_t mk_ = umask_at<I>(es_a);
                                        e0 = and_or(p[I], mk_, e0);
                                        e1 = and_or(qq[I], mk_, e1);
                                });
                                c0 &= f0 & e0;
                                c1 &= f1 & e1;
                        }
#if defined(TRI_PLK_EXP) && TRI_PLK_EXP == 1 // (perf probe: the sweep counts and never finds a candidate — what the candidates cost; results wrong)
                        c0 = c1 = 0;
#endif
                        // the words that hold a candidate go on the wave's queue
                        if (__builtin_amdgcn_ballot_w64((c0 | c1) != 0) != 0ull) {
                                const uint32_t w2 = 2u * (sw * (PLK_SW_WORDS / 2) + lane);
                                const uint64_t b0 = __builtin_amdgcn_ballot_w64(c0 != 0), b1 = __builtin_amdgcn_ballot_w64(c1 != 0);
                                if (c0) {
                                        const uint32_t at = cqn + __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u));
                                        sh.cq[wave][at][0] = w2;
                                        sh.cq[wave][at][1] = c0;
                                }
                                cqn += (uint32_t)__popcll(b0);
                                if (c1) {
                                        const uint32_t at = cqn + __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u));
                                        sh.cq[wave][at][0] = w2 + 1u;
                                        sh.cq[wave][at][1] = c1;
                                }
                                cqn += (uint32_t)__popcll(b1);
                        }
                        sw += SWS;
                        PROF_COUNT(19, lane == 0 ? 1 : 0);
                        while (!stop && cqn >= 64)
                                stop = !work_words();
                });
        }
        // the share is swept: the words still on the queue
        while (!stop && sw >= sw_end && cqn)
                stop = !work_words();
        // the wave's state for its next segment, its matches so far
        sh.w_sw[wave] = sw, sh.w_qn[wave] = qn, sh.w_cqn[wave] = cqn; // (wave-uniform values, every lane stores them)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
                my_matches += __shfl_xor(my_matches, d, 64);
        atomicAdd(&sh.matches, lane == 0 ? my_matches : 0u);
        return sw < sw_end || cqn ? 1u : 0u;
}

// NS: the slots the instantiation can hold.  scratch: sparse_cap u32 per workgroup — the lists of the task's sparse slots.
template <int CODEC, int NS>
__global__ __launch_bounds__(PLK_WG, (PLK_WGS_PER_CU * PLK_WG + 255) / 256) void k_planes(
        const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last, const uint32_t *__restrict__ blk_off, const uint4 *__restrict__ blk_rec,
        const uint32_t *__restrict__ blk_doff, const uint32_t *__restrict__ win, const DevTerm *__restrict__ terms, const DevQuery *__restrict__ plan,
        const DevFused *__restrict__ fused, const DevTask *__restrict__ tasks, const uint32_t *__restrict__ sched, const uint32_t *__restrict__ sterms,
        const double *__restrict__ sweights, const uint32_t ntasks, uint32_t *__restrict__ ticket, uint32_t *__restrict__ counts, const uint32_t k,
        uint32_t *__restrict__ part_docs, double *__restrict__ part_scores, uint32_t *__restrict__ part_counts, const uint32_t *__restrict__ masked,
        const int sim, const uint32_t *__restrict__ planes0, const uint32_t *__restrict__ planes_hi, const uint32_t plw, const uint32_t zrow, uint32_t *__restrict__ scratch, const uint32_t sparse_cap,
        unsigned long long *__restrict__ qthr) {
        PlanesShared &sh = plk_shared();
        const uint32_t tid = threadIdx.x, lane = tid & 63u;
        const uint32_t wave = uni(tid >> 6);
        uint32_t *const lists = scratch + (size_t)blockIdx.x * 2u * sparse_cap; // the sparse slots' lists: entries ...
        uint32_t *const lfreq = lists + sparse_cap;                            // ... and their frequencies
        PROF_DECL;
        PROF_START();
#ifdef TRI_PROF
        const unsigned long long wg_t0 = wall_clock64(); // (100 MHz)
#endif
        for (;;) {
                if (wave == 0) { // uniform draw (see k_and)
                        const uint32_t old = atomicAdd(ticket, 1u);
                        sh.bcast[0] = uni(old) >> 6;
                }
                __syncthreads();
                const uint32_t ticket_no = uni(sh.bcast[0]);
                __syncthreads();
                if (ticket_no >= ntasks)
                        break;
                const uint32_t tix = sched[ticket_no];
                TASKTIME_PLANES(8 * ticket_no);
                const DevTask task = tasks[tix];
                const DevQuery q = plan[task.slot];
                unsigned long long *const gthr = qthr + task.slot; // the threshold the query's tasks share
                {
                        const uint32_t wi = min(tid, (uint32_t)(sizeof(DevFused) / 4 - 1)); // (every lane stores: no divergent branch around the barriers)
                        ((uint32_t *)&sh.fq)[wi] = ((const uint32_t *)(fused + q.fused_idx))[wi];
                }
                for (uint32_t i = tid; i < PLK_HF / 4; i += PLK_WG)
                        sh.hf[i] = 0;
                __syncthreads();
                const DevFused &fq = sh.fq;
                const uint32_t nslots = min(uni(fq.nslots), (uint32_t)NS), nreq = uni(fq.nreq), negs = uni(fq.negslots);
                const uint32_t kk = min(lane, nslots - 1); // lane s (< nslots) of every wave looks after slot s, the lanes above mirror the last slot
                const uint32_t wfirst = task.tile_begin, wend = task.tile_end;
                const uint32_t d_lo = wfirst * PL_W, d_hi = wend * PL_W; // (the planner keeps max docID below 2^31: no wrap)
                {
                        const DevTerm myt = terms[fq.term[kk]];
                        sh.term[kk] = myt;
                        const bool dense = fq.plane[kk] != PL_NONE;
                        // what the slot's scorers add at the frequencies the planes tell (1 .. PLK_LV - 1), and a bound of what they add at any frequency:
                        // BM25 float(w f / (f + 1.2)) < w; TF-IDF sqrt(f) w with f <= 65535; Trivial f
                        double tl[PLK_LV], ubs = 0.0;
#pragma unroll
                        for (uint32_t l = 0; l < PLK_LV; ++l)
                                tl[l] = 0.0;
                        bool leaf = false;
                        for (uint32_t si = 0; si < q.nscore; ++si)
                                if (sterms[q.score_base + si] == fq.term[kk]) {
                                        const double wgt = sweights[q.score_base + si];
#pragma unroll
                                        for (uint32_t l = 1; l < PLK_LV; ++l)
                                                tl[l] += (double)sim_score(sim, wgt, l);
                                        ubs += sim == TRI_SIM_TRIVIAL ? 65535.0 : sim == TRI_SIM_TFIDF ? (wgt > 0 ? 256.0 * wgt : 0.0) : (wgt > 0 ? wgt : 0.0);
                                        leaf = true;
                                }
                        auto up = [](const double x) { return x > 0 ? x * (1.0 + 1e-9) : x * (1.0 - 1e-9); };
                        // the filter's weights: non-decreasing in the level (a negative contribution is bounded by the level below), the top one a bound
                        double fl = 0.0;
                        sh.wl[kk][0] = 0.0, sh.wf[kk][0] = 0.0;
#pragma unroll
                        for (uint32_t l = 1; l < PLK_LV; ++l) {
                                fl = fmax(up(tl[l]), fl);
                                sh.wl[kk][l] = tl[l];
                                sh.wf[kk][l] = fl;
                        }
                        sh.wl[kk][PLK_LV] = 0.0;
                        sh.wf[kk][PLK_LV] = fmax(ubs * (1.0 + 1e-6), fl);
                        sh.top[kk] = leaf ? PLK_LV : 0u;
                        sh.tk_n = 0;
                        {
                                // (another range of the query may have a threshold already: this one filters with it from its first document on)
                                const unsigned long long g = __atomic_load_n(gthr, __ATOMIC_RELAXED);
                                sh.tk_full = g != 0ull ? 1u : 0u;
                                sh.thr_s = g != 0ull ? key_score(g) : 0.0;
                                sh.thr_d = 0xffffffffu;
                        }
                        sh.matches = 0;
                        sh.fall = 1; // no threshold yet: every match is a candidate
                        sh.esel = 0;
                        sh.atab = 0xffffffffu;
                        // a sparse slot's rows that can hold documents of the task's range [first window's first docID, last window's end)
                        uint32_t row0 = 0, nrows = 0;
                        if (!dense && myt.win_off != 0xffffffffu) { // (the cell index: the first block whose last docID reaches a cell's start — two loads)
                                row0 = win[myt.win_off + (d_lo >> CELL_LOG2)];
                                const uint32_t r1 = win[myt.win_off + (d_hi >> CELL_LOG2)];
                                nrows = row0 < myt.nblocks ? min(r1, myt.nblocks - 1) - row0 + 1 : 0;
                        } else if (!dense) {
                                const uint32_t *bl = blk_last + myt.first_block;
                                uint32_t lo = 0, hi = myt.nblocks;
                                while (lo < hi) {
                                        const uint32_t mid = (lo + hi) >> 1;
                                        if (bl[mid] < d_lo)
                                                lo = mid + 1;
                                        else
                                                hi = mid;
                                }
                                row0 = lo;
                                hi = myt.nblocks;
                                while (lo < hi) {
                                        const uint32_t mid = (lo + hi) >> 1;
                                        if (bl[mid] < d_hi)
                                                lo = mid + 1;
                                        else
                                                hi = mid;
                                }
                                nrows = row0 < myt.nblocks ? min(lo, myt.nblocks - 1) - row0 + 1 : 0;
                        }
                        // the lists' places in the scratch region: an exclusive scan over the slots (lanes 0 .. nslots-1 hold distinct slots)
                        uint32_t base = 0;
                        for (uint32_t s2 = 0; s2 < nslots; ++s2) {
                                const uint32_t n2 = (uint32_t)__builtin_amdgcn_readlane((int)nrows, (int)s2);
                                base += s2 < kk ? n2 * 32u : 0u;
                        }
                        sh.sp_row0[kk] = row0;
                        sh.sp_n[kk] = nrows;
                        sh.sp_base[kk] = base;
                }
                __syncthreads();
                // ---- per task, uniform: the dense slots by position, the sparse slots by ordinal, the slots' roles over both index spaces
                uint32_t dense_mask = 0, leafm = 0, list_rows = 0;
#pragma unroll
                for (uint32_t s = 0; s < NS; ++s)
                        if (s < nslots) {
                                if (uni(fq.plane[s]) != PL_NONE)
                                        dense_mask |= 1u << s;
                                else
                                        list_rows += uni(sh.sp_n[s]);
                                leafm |= (uni(sh.top[s]) ? 1u : 0u) << s;
                        }
                const uint32_t sparse_mask = ((1u << nslots) - 1u) & ~dense_mask;
                dense_mask = uni(dense_mask), leafm = uni(leafm), list_rows = uni(list_rows);
                const uint32_t nd = (uint32_t)__builtin_popcount(dense_mask), nsp = (uint32_t)__builtin_popcount(sparse_mask);
                uint32_t dsl[NS]; // dense position -> slot
                const uint32_t *pLV[NS]; // ... -> its row's level words
                {
                        uint32_t dm = dense_mask;
#pragma unroll
                        for (uint32_t i = 0; i < NS; ++i) {
                                const uint32_t s = dm ? (uint32_t)__builtin_ctz(dm) : 0u;
                                dsl[i] = uni(s);
                                pLV[i] = planes_hi + (size_t)(dm ? uni(fq.plane[s]) : zrow) * PL_HI * plw + (size_t)PL_HI_LEVELS * plw; // (zrow: the all-zero row — what a position beyond nd reads)
                                dm &= dm - 1u;
                        }
                }
                auto to_dense = [&](const uint32_t slots) { // a set of slots as a set of dense positions
                        uint32_t out = 0;
#pragma unroll
                        for (uint32_t i = 0; i < NS; ++i)
                                out |= (i < nd ? (slots >> dsl[i]) & 1u : 0u) << i;
                        return out;
                };
                uint32_t gsl[FUS_MAX_SLOTS], gd[FUS_MAX_SLOTS];
                bool sweepable = true; // every required group holds a dense slot: documents can match through their dense slots alone
#pragma unroll
                for (uint32_t g = 0; g < FUS_MAX_SLOTS; ++g) {
                        gsl[g] = g < nreq ? uni(fq.gslots[g]) : 0u;
                        gd[g] = uni(to_dense(gsl[g]));
                        sweepable = sweepable && (g >= nreq || gd[g] != 0);
                }
                const uint32_t negd = uni(to_dense(negs)), leafd = uni(to_dense(leafm));
#pragma unroll
                for (uint32_t i = 0; i < NS; ++i)
                        if (tid == i && i < nd) {
#pragma unroll
                                for (uint32_t l = 0; l <= PLK_LV; ++l)
                                        sh.dwf[i][l] = sh.wf[dsl[i]][l];
                                sh.dtop[i] = sh.top[dsl[i]];
                                sh.ddocs[i] = sh.term[dsl[i]].documents;
                        }
                PROF_LAP(0);
                // ---- the sparse slots' lists: every row that can reach the task's range, one lane per row, 32 entries each
                for (uint32_t v0 = 0; v0 < list_rows; v0 += PLK_WG) {
                        const uint32_t v = v0 + tid;
                        if (v < list_rows) {
                                uint32_t s = 0, r = v;
                                for (uint32_t s2 = 0; s2 < nslots; ++s2) { // which slot's rows v falls into
                                        const uint32_t n2 = ((sparse_mask >> s2) & 1u) ? uni(sh.sp_n[s2]) : 0u;
                                        if (s == s2 && r >= n2) {
                                                r -= n2;
                                                s = s2 + 1;
                                        }
                                }
                                planes_list_row<CODEC>(index, blk_last, blk_off, blk_rec, blk_doff, sh.term[s], sh.sp_row0[s] + r, lists + sh.sp_base[s] + r * 32u, lfreq + sh.sp_base[s] + r * 32u);
                        }
                }
                __syncthreads(); // (the lists are written: a workgroup barrier orders the global stores for the workgroup's own later loads)
                PROF_LAP(7);
                // the sparse slots by ordinal: slot, list extent, list base (uniform registers)
                uint32_t ssl[PLK_MAX_SPARSE], sp_n32[PLK_MAX_SPARSE], sp_off[PLK_MAX_SPARSE];
                {
                        uint32_t sm = sparse_mask;
#pragma unroll
                        for (uint32_t j = 0; j < PLK_MAX_SPARSE; ++j) {
                                const uint32_t s = sm ? (uint32_t)__builtin_ctz(sm) : 0u;
                                ssl[j] = uni(s);
                                sp_n32[j] = sm ? uni(sh.sp_n[s]) * 32u : 0u;
                                sp_off[j] = sm ? uni(sh.sp_base[s]) : 0u;
                                sm &= sm - 1u;
                        }
                }
                const uint32_t total_items = list_rows * 32u;
                // an item (index into the concatenation of the lists) -> its list's ordinal and the entry
                auto item_entry = [&](const uint32_t v, uint32_t &j_out, uint32_t &at_out) {
                        uint32_t j = 0, ei = v, off = 0;
#pragma unroll
                        for (uint32_t j2 = 0; j2 < PLK_MAX_SPARSE; ++j2) {
                                if (j == j2 && j2 < nsp && ei >= sp_n32[j2]) {
                                        ei -= sp_n32[j2];
                                        j = j2 + 1;
                                }
                                off = j == j2 ? sp_off[j2] : off;
                        }
                        j_out = j;
                        at_out = off + ei;
                        return v < total_items && j < nsp ? lists[off + ei] : PLK_PAD;
                };
                // ---- the hashed filter: every sparse document of the range marks its byte with its list's bit
                for (uint32_t v = tid; v < total_items; v += PLK_WG) {
                        uint32_t j, at;
                        const uint32_t e = item_entry(v, j, at), doc = e >> 1;
                        if (e != PLK_PAD && doc >= d_lo && doc < d_hi)
                                atomicOr(&sh.hf[(doc & (PLK_HF - 1u)) >> 2], (1u << j) << (8u * (doc & 3u)));
                }
                __syncthreads();
                PROF_LAP(1);
                TASKTIME_PLANES(8 * ticket_no + 2);
                uint32_t my_matches = 0; // (wraps: phase A adds signed corrections)
                uint32_t qn = 0;         // entries on this wave's queue of candidates that wait for an exact frequency (wave-uniform)
                auto offer = [&](const double sc, const uint32_t doc) { // false: no room (the buffer wants pruning)
                        const uint32_t slot = atomicAdd(&sh.tk_n, 1u);
                        if (slot >= PLK_CAP)
                                return false;
                        sh.tk_s[slot] = sc;
                        sh.tk_d[slot] = doc;
                        return true;
                };
                // the exact score of a document from its DENSE slots' levels (three bits per slot): the frequencies the planes tell from the level weights,
                // the others from the postings
                auto exact_score = [&](const uint32_t doc, const uint32_t levels) {
                        double sk = 0.0;
                        for (uint32_t s = 0; s < nslots; ++s) {
                                const uint32_t l = (levels >> (3 * s)) & 7u;
                                if (!l)
                                        continue;
                                if (l < PLK_LV) {
                                        sk += sh.wl[s][l];
                                        continue;
                                }
                                const uint32_t f = planes_lookup_freq<CODEC>(index, blk_last, blk_off, blk_rec, blk_doff, win, sh.term[s], doc);
                                const uint32_t term = fq.term[s];
                                for (uint32_t si = 0; si < q.nscore; ++si)
                                        if (sterms[q.score_base + si] == term)
                                                sk += (double)sim_score(sim, sweights[q.score_base + si], f);
                        }
                        return sk;
                };
                auto work_queue = [&]() { // the last (up to) 64 entries of the queue; entries that found no room go back
                        const uint32_t take_n = min(qn, 64u), base = qn - take_n;
                        qn = base;
                        bool back = false;
                        uint32_t doc = 0, levels = 0;
                        if (lane < take_n) {
                                doc = sh.wq[wave][base + lane][0];
                                levels = sh.wq[wave][base + lane][1];
                                const double sk = exact_score(doc, levels);
                                if (!uni(sh.tk_full) || better(sk, doc, sh.thr_s, sh.thr_d))
                                        back = !offer(sk, doc);
                        }
                        PROF_COUNT(17, lane == 0 ? take_n : 0);
                        const uint64_t bm = __builtin_amdgcn_ballot_w64(back);
                        if (back) {
                                const uint32_t at = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                                sh.wq[wave][at][0] = doc;
                                sh.wq[wave][at][1] = levels;
                        }
                        qn += (uint32_t)__popcll(bm);
                        PROF_LAP(11);
                };
                // the waves meet: prune if the buffer wants it; false: no wave has work left
                auto meet = [&](const bool more) {
                        sh.flag[wave] = more ? 1u : 0u; // (wave-uniform value, every lane stores it)
                        __syncthreads();
                        uint32_t anyp = 0;
#pragma unroll
                        for (uint32_t wv = 0; wv < PLK_WG / 64; ++wv)
                                anyp |= sh.flag[wv];
                        anyp = uni(anyp);
                        const uint32_t n = min(uni(sh.tk_n), PLK_CAP);
                        __syncthreads(); // (every lane has read the flags and tk_n)
                        bool pruned = false;
                        if (n >= PLK_PRUNE_AT) {
                                planes_prune(sh, n, k, gthr);
                                pruned = true;
                                PROF_COUNT(18, tid == 0 ? 1 : 0);
                        }
                        PROF_LAP(5);
                        return ((uint32_t)(anyp != 0)) | (pruned ? 2u : 0u);
                };
                // ---- PHASE A: every document of the sparse lists, each on its own — 64 per wave and step, the waves interleaved
                {
                        const uint32_t nchunks = (total_items + 63u) / 64u;
                        uint32_t chunk = wave;
                        bool have = false, pend = false;
                        double pscore = 0.0;
                        uint32_t pdoc = 0;
                        for (;;) {
                                while (chunk < nchunks) {
                                        if (uni(__atomic_load_n(&sh.tk_n, __ATOMIC_RELAXED)) >= PLK_PRUNE_AT)
                                                break; // the buffer wants pruning first: to the barrier (the chunk stays as it is)
                                        const bool full = uni(sh.tk_full) != 0;
                                        if (!have) {
                                                uint32_t j, at;
                                                const uint32_t e = item_entry(chunk * 64u + lane, j, at), doc = e >> 1;
                                                bool valid = e != PLK_PAD && doc >= d_lo && doc < d_hi && doc != 0;
                                                uint32_t present = 0, levels = 0;
                                                double sk = 0.0; // the sparse slots' part of the score: their frequencies stand in the lists
                                                if (valid) {
                                                        // the other sparse lists that may hold the document (the filter's byte), each bisected: a document an
                                                        // EARLIER list holds is that list's item
                                                        const uint32_t hb = (sh.hf[(doc & (PLK_HF - 1u)) >> 2] >> (8u * (doc & 3u))) & 0xffu;
#pragma unroll
                                                        for (uint32_t j2 = 0; j2 < PLK_MAX_SPARSE; ++j2) {
                                                                if (j2 >= nsp)
                                                                        break;
                                                                uint32_t fat = PLK_NOT_FOUND;
                                                                if (j2 == j)
                                                                        fat = at;
                                                                else if ((hb >> j2) & 1u) {
                                                                        fat = planes_list_find(lists + sp_off[j2], sp_n32[j2], doc);
                                                                        fat = fat != PLK_NOT_FOUND ? fat + sp_off[j2] : fat;
                                                                }
                                                                if (fat != PLK_NOT_FOUND) {
                                                                        valid = valid && j2 >= j;
                                                                        present |= 1u << ssl[j2];
                                                                        if ((leafm >> ssl[j2]) & 1u) {
                                                                                const uint32_t f = lfreq[fat], term = fq.term[ssl[j2]];
                                                                                for (uint32_t si = 0; si < q.nscore; ++si)
                                                                                        if (sterms[q.score_base + si] == term)
                                                                                                sk += (double)sim_score(sim, sweights[q.score_base + si], f);
                                                                        }
                                                                }
                                                        }
                                                }
                                                pend = false;
                                                if (valid) {
                                                        // its level in the dense slots: one probe each
                                                        const uint32_t wi = doc >> 5, bit = doc & 31u;
#pragma unroll
                                                        for (uint32_t i = 0; i < NS; ++i) {
                                                                if (i >= nd)
                                                                        break;
                                                                const uint32_t *lv = pLV[i] + 3u * wi; // (the level's three bits: adjacent words)
                                                                const uint32_t l = ((lv[0] >> bit) & 1u) | (((lv[1] >> bit) & 1u) << 1) | (((lv[2] >> bit) & 1u) << 2);
                                                                present |= (l ? 1u : 0u) << dsl[i];
                                                                levels |= (((leafd >> i) & 1u) ? l : 0u) << (3u * dsl[i]);
                                                        }
                                                        // the predicate with all the slots, and with the dense slots alone (what the sweep counts)
                                                        const uint32_t pd = present & dense_mask;
                                                        bool okf = !(present & negs), okd = sweepable && !(pd & negs);
#pragma unroll
                                                        for (uint32_t g = 0; g < FUS_MAX_SLOTS; ++g) {
                                                                okf = okf && (g >= nreq || (present & gsl[g]) != 0);
                                                                okd = okd && (g >= nreq || (pd & gsl[g]) != 0);
                                                        }
                                                        if (masked && ((masked[wi] >> bit) & 1u)) // masked_documents_registry::test (docidupdates.h:90-119)
                                                                okf = okd = false;
                                                        my_matches += (okf ? 1u : 0u) - (okd ? 1u : 0u);
                                                        if (okf) {
                                                                // a bound first: a frequency the planes do not tell costs a walk into the postings
                                                                double sb = sk;
                                                                for (uint32_t s = 0; s < nslots; ++s) {
                                                                        const uint32_t l = (levels >> (3 * s)) & 7u;
                                                                        sb += l ? sh.wf[s][l] : 0.0;
                                                                }
                                                                if (!full || better(sb * (1.0 + 1e-9), doc, sh.thr_s, sh.thr_d)) {
                                                                        pscore = sk + exact_score(doc, levels);
                                                                        pdoc = doc;
                                                                        pend = true;
                                                                }
                                                        }
                                                }
                                                have = true;
                                                PROF_COUNT(22, lane == 0 ? 1 : 0);
                                        }
                                        if (pend && (!full || better(pscore, pdoc, sh.thr_s, sh.thr_d)))
                                                pend = !offer(pscore, pdoc);
                                        else
                                                pend = false;
                                        if (__builtin_amdgcn_ballot_w64(pend) != 0ull)
                                                break; // no room: to the barrier, the offers are made again behind it
                                        have = false;
                                        chunk += PLK_WG / 64;
                                }
                                PROF_LAP(2);
                                if (!(meet(chunk < nchunks) & 1u))
                                        break;
                        }
                }
                TASKTIME_PLANES(8 * ticket_no + 3);
                // ---- the threshold after phase A (or another range's), the sweep's tables
                planes_prune(sh, min(uni(sh.tk_n), PLK_CAP), k, gthr);
                const bool sweep_on = sweepable && nd != 0;
                if (sweep_on)
                        planes_filter(sh, nd);
                PROF_LAP(3);
                TASKTIME_PLANES(8 * ticket_no + 4);
                // ---- PHASE B: the sweep over the dense slots (planes_sweep_segment, above), segment by segment: a segment ends where the wave's queue
                //      wants working off, where the candidate buffer wants pruning, or with the wave's share of the range
                if (sweep_on) {
                        if (tid == 0) {
                                sh.sa.nd = nd, sh.sa.nreq = nreq, sh.sa.negd = negd, sh.sa.leafd = leafd, sh.sa.nsp = nsp, sh.sa.zrow = zrow;
#pragma unroll
                                for (uint32_t i = 0; i < NS; ++i) {
                                        sh.sa.dsl[i] = dsl[i];
                                        sh.sa.prow[i] = i < nd ? fq.plane[dsl[i]] : zrow;
                                }
#pragma unroll
                                for (uint32_t g = 0; g < FUS_MAX_SLOTS; ++g)
                                        sh.sa.gd[g] = gd[g];
#pragma unroll
                                for (uint32_t j = 0; j < PLK_MAX_SPARSE; ++j) {
                                        sh.sa.sp_off[j] = sp_off[j];
                                        sh.sa.sp_n32[j] = sp_n32[j];
                                }
                        }
                        {
                                sh.w_sw[wave] = wfirst * (PL_W / PLK_SW) + wave; // the waves' sub-windows interleave (wave-uniform values, every lane stores them)
                                sh.w_end[wave] = wend * (PL_W / PLK_SW);
                                sh.w_qn[wave] = 0, sh.w_cqn[wave] = 0;
                        }
                        __syncthreads();
                        for (;;) {
                                bool done = false;
                                for (;;) {
                                        sh.w_qn[wave] = qn;
                                        uint32_t r;
                                        if constexpr (NS == PLK_NS_SMALL) {
                                                if (nd <= 1)
                                                        r = planes_sweep_segment<1>(planes0, planes_hi, plw, masked, lists);
                                                else if (nd == 2)
                                                        r = planes_sweep_segment<2>(planes0, planes_hi, plw, masked, lists);
                                                else if (nd == 3)
                                                        r = planes_sweep_segment<3>(planes0, planes_hi, plw, masked, lists);
                                                else
                                                        r = planes_sweep_segment<5>(planes0, planes_hi, plw, masked, lists);
                                        } else
                                                r = planes_sweep_segment<NS>(planes0, planes_hi, plw, masked, lists);
                                        qn = uni(sh.w_qn[wave]);
                                        // the queue is worked off out here: a call among the sweep's live registers would have the compiler spill them on the hot path
                                        while (qn >= 64 && uni(__atomic_load_n(&sh.tk_n, __ATOMIC_RELAXED)) < PLK_PRUNE_AT)
                                                work_queue();
                                        done = uni(r) == 0;
                                        if (done || uni(__atomic_load_n(&sh.tk_n, __ATOMIC_RELAXED)) >= PLK_PRUNE_AT)
                                                break;
                                }
                                PROF_LAP(4);
                                // ---- the waves meet: prune if the buffer wants it, go on while any of them has work left
                                const uint32_t mres = meet(!done);
                                if (mres & 2u)
                                        planes_filter(sh, nd);
                                if (!(mres & 1u))
                                        break;
                        }
                }
                TASKTIME_PLANES(8 * ticket_no + 5);
                // ---- the candidates still waiting for their frequencies
                for (;;) {
                        while (qn && uni(__atomic_load_n(&sh.tk_n, __ATOMIC_RELAXED)) < PLK_PRUNE_AT)
                                work_queue();
                        if (!(meet(qn != 0) & 1u))
                                break;
                }
                TASKTIME_PLANES(8 * ticket_no + 6);
                // ---- the task's result: its best k (ranked) and its match count
                __syncthreads();
                planes_prune(sh, min(uni(sh.tk_n), PLK_CAP), k, gthr);
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1)
                        my_matches += __shfl_xor(my_matches, d, 64);
                atomicAdd(&sh.matches, lane == 0 ? my_matches : 0u); // (every lane issues it: no single-lane branch; the corrections wrap, the sum does not)
                __syncthreads();
                const uint32_t n = uni(sh.tk_n);
                for (uint32_t i = tid; i < n; i += PLK_WG) {
                        part_docs[(uint64_t)tix * k + i] = sh.tk_d[i];
                        part_scores[(uint64_t)tix * k + i] = sh.tk_s[i];
                }
                if (wave == 0) {
                        part_counts[tix] = n;
                        counts[tix] = uni(sh.matches);
                }
                TASKTIME_PLANES(8 * ticket_no + 1);
                __syncthreads();
                PROF_LAP(6);
        }
        PROF_FLUSH();
#ifdef TRI_PROF // when the workgroups ended: g_prof[28] how many, [29] ~(earliest start), [30] sum of the end times, [31] the latest end
        if (tid == 0) {
                const unsigned long long t1 = wall_clock64();
                atomicAdd(&g_prof[28], 1ull);
                atomicMax(&g_prof[29], ~wg_t0);
                atomicAdd(&g_prof[30], t1);
                atomicMax(&g_prof[31], t1);
        }
#endif
}
