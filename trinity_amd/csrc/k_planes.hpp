// k_planes.hpp — term planes (a head term decoded once per launch for every query of the batch that names it) and the
// AccumulatedScoreScheme top-K kernel that runs over them
// Part of libtrinity_hip.so (MI355X / gfx950); included by trinity_hip.hip.  New code, no reference source.
#pragma once
#include <type_traits>
#include "k_fused.hpp"

// Under Zipf a handful of terms carry most of a batch's postings (at the 10M-document configuration the 40 most frequent terms
// hold 98 % of the postings the 5-term queries of SURVEY §8(d) cfg3 touch), and the reference decodes such a list again for every
// query that names it (Decoder::init + next() per query, google_codec.cpp:777-819 / lucene_codec.cpp:568-594).  Here every LAUNCH
// decodes each of those lists ONCE — k_term_planes, inside the timed region, from the segment's own codec bytes — into three
// bitmaps over the docID space:
//     plane A   bit d set  <=>  document d holds the term            (PostingsListIterator::current() would stop on d)
//     plane B   bit d set  <=>  ... and its frequency there is not 1
//     plane C   bit d set  <=>  ... and not 2 either (the exact frequency is then read from the postings on demand)
// and the matching kernels read the planes: k_and tests a candidate with one bit probe instead of bracketing and decoding a block
// (Conjuction::next_impl's advance(), docset_iterators.cpp:308-348), k_and_dense ORs a plane's words into its window bitmap instead
// of walking the term's rows (docset_spans.cpp:98-173), k_planes (below) evaluates union / CNF predicates 32 documents per word.
// The planes live in a scratch region owned by the batch (3 x (max docID / 8) bytes per term); nothing survives the launch.

constexpr uint32_t PL_CELLS = PL_W / CELL_DOCS; // cell-index entries per plane window
constexpr uint32_t PL_STRIDE = PL_WORDS + 32;   // LDS words between a slot's planes (word PL_WORDS of each: the sink)

// A decoded posting into LDS planes A, B, C (a[], a[PL_STRIDE], a[2 * PL_STRIDE]).  Documents outside the window land in the sink word.
struct PlanePost {
        uint32_t *a;
        __device__ __forceinline__ void doc(const uint32_t rel) {
                const uint32_t r = min(rel, PL_W);
                atomicOr(&a[r >> 5], 1u << (r & 31u));
        }
        __device__ __forceinline__ void operator()(const uint32_t rel, const uint32_t f) {
                const uint32_t r = min(rel, PL_W);
                const uint32_t bit = 1u << (r & 31u), f16 = f & 0xffffu; // (the frequency a scorer sees is tokenpos_t, 16 bits: codecs.h:217)
                atomicOr(&a[r >> 5], bit);
                if (f16 != 1u) {
                        atomicOr(&a[(r >> 5) + PL_STRIDE], bit);
                        if (f16 != 2u)
                                atomicOr(&a[(r >> 5) + 2 * PL_STRIDE], bit);
                }
        }
};

// One workgroup per (plane row, window): the rows (<= 32 documents each) of the term that reach the window are decoded, one lane
// per row, into LDS planes, which are then written out whole — every word of every plane is written by exactly one workgroup,
// so the scratch region needs no clearing between launches.
template <int CODEC>
__global__ __launch_bounds__(AND_WG) void k_term_planes(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last,
                                                        const uint32_t *__restrict__ blk_off, const uint4 *__restrict__ blk_rec,
                                                        const uint32_t *__restrict__ blk_doff, const uint32_t *__restrict__ win,
                                                        const DevTerm *__restrict__ terms, const uint32_t *__restrict__ build /* (term, row) pairs */,
                                                        uint32_t *__restrict__ planes, const uint32_t plw) {
        __shared__ uint32_t pl[PL_PLANES * PL_STRIDE];
        const uint32_t tid = threadIdx.x, w = blockIdx.x, row = build[2 * blockIdx.y + 1];
        for (uint32_t i = tid; i < PL_PLANES * PL_STRIDE; i += AND_WG)
                pl[i] = 0;
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
                        PlanePost post{pl};
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
        uint32_t *pa = planes + (size_t)row * PL_PLANES * plw + (size_t)w * PL_WORDS;
        for (uint32_t i = tid; i < PL_WORDS; i += AND_WG) {
                pa[i] = pl[i];
                pa[plw + i] = pl[PL_STRIDE + i];
                pa[2 * plw + i] = pl[2 * PL_STRIDE + i];
        }
}

// ------------------------------------------------------------------------------------------ k_planes
// AccumulatedScoreScheme + top-K of a CNF query (a union, a conjunction of terms / OR-groups, an excluded group, optional scoring
// terms: everything k_fused's CNF instantiations take) in one pass over windows of PL_W documents, on BIT PLANES instead of a word
// per document:
//   * every slot (distinct term) of the query presents, per window, LEVEL planes: A (the document holds the term), B (its frequency
//     is not 1) and, for a head term, C (nor 2).  A head term's planes come straight from the batch's term planes (global memory,
//     L2 / Infinity-Cache resident: k_term_planes decoded the list once for every query of the launch).  Any other term's rows that
//     fall into the task's docID range are decoded ONCE, at the start of the task — one lane per row of <= 32 documents, every lane
//     of the workgroup busy, the same row readers as k_fused — into a sorted (docID, frequency-is-not-1) list in a scratch region
//     of the workgroup; per window the list's next entries are picked up with one coalesced load and OR-ed into LDS planes.
//   * the predicate is word-wise: a required group = the OR of its slots' A words, the conjunction their AND, the excluded group an
//     AND-NOT, masked documents (docidupdates.h:90-119) another — 32 documents per instruction; the match count is a popcount.
//     (What docset_spans.cpp:98-173 / 681-790 do per document and docset_iterators.cpp:226-405 per posting.)
//   * the candidate filter.  A slot is at one of up to four LEVELS in a document: absent, frequency 1, frequency 2 (head terms; its
//     scorers then add exactly what they add at that frequency), any other frequency (they add at most a bound).  Whenever the
//     threshold (the k-th best score so far) moves, planes_filter rebuilds a TABLE with one bit per level vector (do the levels'
//     weights reach the threshold?) and the ESSENTIAL PLANES — MaxScore's essential slots refined by level: a slot capped at level c
//     is essential only through its plane c + 1.  The sweep ORs the essential planes word-wise; only the documents in that word are
//     looked up in the table (level code from word-wise level bits), and only table hits become candidates — a match whose
//     frequencies are all known (almost all of them) is thereby tested against its EXACT score without being touched.
//   * a SEED pass scores the documents of the query's rarest decoded lists before the sweep (plane probes + bisections), so the
//     threshold is near final from the first window on; the docID ranges (tasks) of a query share one threshold (planes_prune).
//   * candidates are scored one per lane by the wave that owns their words, no workgroup barrier: levels from registers; a
//     candidate with a slot of unknown frequency that the bound does not rule out waits on the wave's queue, and the queue is
//     worked off 64 at a time — the exact frequencies come from the postings (directory cell -> block -> register row reader).
//   * per task: min(matches, k) ranked (docID, score) pairs and the match count; k_topk_merge folds a query's tasks.
constexpr int PLK_WG = 512;
// (PLK_MAX_SPARSE, PLK_NS_SMALL: dev_structs.hpp)
constexpr uint32_t PLK_CAP = 512;       // candidate buffer (one entry per thread when it is pruned)
constexpr uint32_t PLK_PRUNE_AT = 384;  // the waves stop taking candidates once it holds this many: it is pruned to the best k, then they resume
constexpr uint32_t PLK_FTAB_WORDS = (1u << (2 * FUS_MAX_SLOTS)) / 32; // the filter table: one bit per level vector (two bits per slot)
#ifndef TRI_PLK_WGS
#define TRI_PLK_WGS 2
#endif
constexpr uint32_t PLK_WGS_PER_CU = TRI_PLK_WGS; // (LDS: two fit)
constexpr uint32_t PLK_WQ = 128;        // per-wave queue of candidates waiting for a frequency lookup: worked off 64 at a time, every lane busy
constexpr uint32_t PLK_PAD = 0xffffffffu; // list padding (sorts last)
constexpr uint32_t PLK_SEED_FIRST = 4096;  // the seed pass takes the shortest decoded list if it has at most this many entries in the task's range ...
constexpr uint32_t PLK_SEED_MORE = 2048;   // ... and further ones while the total stays below this
constexpr uint32_t PLK_SW_WORDS = 128;    // a wave's sub-window: two words (64 documents) per lane ...
constexpr uint32_t PLK_SW = PLK_SW_WORDS * 32; // ... 4096 documents
constexpr uint32_t PLK_SW_STRIDE = PLK_SW_WORDS + 4; // LDS words between a decoded slot's A and B plane of a sub-window
static_assert(PL_W % PLK_SW == 0, "a task's windows split into whole sub-windows");
static_assert(PLK_CAP == PLK_WG && TOPK_MAX < PLK_PRUNE_AT && PLK_PRUNE_AT < PLK_CAP, "pruning leaves room; a pruned buffer is below the stop mark");

struct PlanesShared {
        uint32_t pl[PLK_WG / 64][PLK_MAX_SPARSE][2 * PLK_SW_STRIDE]; // per wave and decoded slot: planes A and B of the wave's current sub-window
        double tk_s[PLK_CAP];
        uint32_t tk_d[PLK_CAP];
        DevTerm term[FUS_MAX_SLOTS];
        double wl[FUS_MAX_SLOTS][4]; // per slot and level: what its scorers add (exact below the slot's top level)
        double wf[FUS_MAX_SLOTS][4]; // ... rounded up a hair for the filter (it must never lose a tie to rounding), non-decreasing in the level; top level: a bound
        double thr_s;
        uint32_t thr_d;
        uint32_t tk_n, tk_full, matches;
        uint32_t leaf;                // the slots that have a scorer
        uint32_t top[FUS_MAX_SLOTS];  // per slot: its top level (3: a term plane; 2: a decoded list; 0: no scorer)
        uint32_t esel;                // the essential planes (two bits per slot — 0: A, 1: B, 2: C, 3: none): every candidate is in one of them
        uint32_t fall;                // 1: no threshold yet (or one that rules nothing out): every match is a candidate
        uint32_t ftab[PLK_FTAB_WORDS]; // the candidate filter: bit `code` (two bits per slot: its level) set <=> the levels' weights reach the threshold
        uint32_t flag[PLK_WG / 64];
        uint32_t wq[PLK_WG / 64][PLK_WQ][2]; // per wave: candidates waiting for exact frequencies {docID, the slots' levels (two bits each)}
        uint32_t bcast[4];
        uint32_t zero[2 * PLK_SW_STRIDE]; // what a slot WITHOUT decoded list reads as its LDS planes (so that the sweep needs no per-slot selects)
        uint32_t sp_row0[FUS_MAX_SLOTS], sp_n[FUS_MAX_SLOTS], sp_base[FUS_MAX_SLOTS]; // decoded slots: first row, rows, first entry of the list in the scratch region
        DevFused fq;
};
static_assert(sizeof(PlanesShared) * PLK_WGS_PER_CU <= 160u * 1024u, "two workgroups per CU");

// A row of a decoded slot into its list: 32 entries per row (docID << 1 | frequency-is-not-1), the unused ones of a short last row padded.
struct ListPost {
        uint32_t *out;
        uint32_t i = 0;
        __device__ __forceinline__ void doc(const uint32_t rel) { out[i++] = rel << 1 | 1u; }
        __device__ __forceinline__ void operator()(const uint32_t rel, const uint32_t f) { out[i++] = rel << 1 | ((f & 0xffffu) != 1u ? 1u : 0u); }
};

// A score as an unsigned key of the same order (0: none), for the per-query threshold the tasks of a query share
__device__ __forceinline__ unsigned long long score_key(const double sc) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(sc);
        return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_score(const unsigned long long key) {
        return __longlong_as_double((long long)((key >> 63) ? (key & 0x7fffffffffffffffull) : ~key));
}

// Keep the best k of the n (<= PLK_CAP = PLK_WG) buffered candidates, best first (rank by counting: the order is strict), and move the
// threshold.  A query cut into several docID ranges (tasks) shares one threshold through *gthr: a task's k-th best score says that k
// documents reach it, so no task needs documents below it — the later and the slower ranges filter with the best k-th score any range
// has seen, not with their own, and cutting a query into ranges costs next to no extra candidates.  (The threshold only ever prunes:
// results do not depend on when a task sees another's.)
__device__ void planes_prune(PlanesShared &sh, const uint32_t n, const uint32_t k, unsigned long long *__restrict__ gthr) {
        const uint32_t tid = threadIdx.x;
        double es = 0;
        uint32_t ed = 0, rk = 0xffffffffu;
        if (tid < n) {
                es = sh.tk_s[tid];
                ed = sh.tk_d[tid];
                uint32_t c = 0;
                for (uint32_t j = 0; j < n; ++j)
                        c += better(sh.tk_s[j], sh.tk_d[j], es, ed) ? 1u : 0u;
                rk = c;
        }
        __syncthreads();
        if (rk < k) {
                sh.tk_s[rk] = es;
                sh.tk_d[rk] = ed;
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

// The candidate filter, recomputed whenever the threshold moves (every thread calls it; it ends with a barrier).  Every scoring slot
// is at a level 0 .. top[slot] in a document and adds at most wf[slot][level] there (exactly, below the top level), so a document's
// score is at most the sum of its slots' level weights: the TABLE holds, for every level vector (two bits per slot), whether that sum
// reaches the current k-th best score.  A match is looked up with its own level vector — a few shifts on the words the sweep already
// holds — so documents whose frequencies are all known (almost all) are tested against their exact score.  To keep that off most
// documents, MaxScore's essential slots come first, word-wise: with the slots ordered by their bound, the longest prefix whose bounds
// sum to less than the threshold cannot lift a document over it, so a candidate holds one of the OTHER slots — the OR of their A words
// picks the few documents that are looked up at all.  No threshold yet, or one that rules nothing out: every match is a candidate.
__device__ void planes_filter(PlanesShared &sh, const uint32_t nslots) {
        const uint32_t tid = threadIdx.x;
        const double thr = sh.thr_s;
        const bool full = uni(sh.tk_full) != 0 && 0.0 < thr;
        sh.fall = full ? 0u : 1u; // (uniform stores)
        sh.esel = 0; // (plane A of every slot)
        if (!full) {
                __syncthreads();
                return;
        }
        const uint32_t words = (1u << (2 * nslots)) / 32u > 0 ? (1u << (2 * nslots)) / 32u : 1u;
        for (uint32_t wd = tid; wd < words; wd += PLK_WG) {
                uint32_t bits = 0;
                for (uint32_t j = 0; j < 32; ++j) {
                        const uint32_t code = wd * 32u + j;
                        double sum = 0.0;
                        bool valid = code < (1u << (2 * nslots));
                        for (uint32_t sl = 0; sl < nslots; ++sl) {
                                const uint32_t l = (code >> (2 * sl)) & 3u;
                                valid &= l <= sh.top[sl];
                                sum += l ? sh.wf[sl][l] : 0.0;
                        }
                        bits |= (valid && !(sum < thr) ? 1u : 0u) << j;
                }
                sh.ftab[wd] = bits;
        }
        // the essential planes (same values in every lane): whole slots first, by ascending bound ...
        const double thr_lo = thr * (1.0 - 1e-9); // (the table adds the weights in its own order: a hair of room for the rounding)
        uint32_t done = 0, ess = 0;
        double p = 0.0, spent = 0.0;
        for (uint32_t r = 0; r < nslots; ++r) { // selection by ascending bound (<= 8 slots)
                uint32_t best = 0;
                double bv = 1e300;
                for (uint32_t sl = 0; sl < nslots; ++sl) {
                        const double b = sh.top[sl] ? sh.wf[sl][sh.top[sl]] : 0.0;
                        if (!((done >> sl) & 1u) && b < bv) {
                                bv = b;
                                best = sl;
                        }
                }
                done |= 1u << best;
                p += bv;
                if (!(p < thr_lo))
                        ess |= 1u << best;
                else
                        spent = p;
        }
        // ... then what is left of the threshold buys the essential slots' LOW levels: a slot capped at level c is essential only through
        // its plane c + 1 (B: frequency not 1 — a fraction of A; C: nor 2).  Each round takes the raise that drops the most documents
        // (estimated from the terms' document counts) among those that still fit.
        uint32_t cap[FUS_MAX_SLOTS];
        for (uint32_t sl = 0; sl < FUS_MAX_SLOTS; ++sl)
                cap[sl] = sl < nslots && ((ess >> sl) & 1u) ? 0u : 3u;
        for (uint32_t r = 0; r < 2 * FUS_MAX_SLOTS; ++r) {
                uint32_t best = 0xffffffffu;
                double gain = 0.0, cost = 0.0;
                for (uint32_t sl = 0; sl < nslots; ++sl) {
                        const uint32_t c = cap[sl], tp = sh.top[sl];
                        if (c >= tp) // (its top plane already, or not essential at all)
                                continue;
                        const double dw = sh.wf[sl][c + 1] - (c ? sh.wf[sl][c] : 0.0);
                        const double docs = (double)sh.term[sl].documents * (c == 0 ? 0.65 : c == 1 ? 0.2 : 0.15); // (the documents at exactly level c + 1, roughly)
                        if (spent + dw < thr_lo && gain < docs) {
                                gain = docs;
                                cost = dw;
                                best = sl;
                        }
                }
                if (best == 0xffffffffu)
                        break;
                cap[best] += 1;
                spent += cost;
        }
        uint32_t esel = 0;
        for (uint32_t sl = 0; sl < FUS_MAX_SLOTS; ++sl) {
                const uint32_t c = cap[sl], tp = sl < nslots ? sh.top[sl] : 0u;
                esel |= (c + 1 > tp ? 3u : c) << (2 * sl);
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
                                             uint32_t *__restrict__ out) {
        const uint32_t *bl = blk_last + t.first_block;
        const uint32_t prev = b ? bl[b - 1] : 0, last = bl[b];
        ListPost post{out};
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
        asm volatile("s_bfe_i32 %0, %1, %2" : "=s"(r) : "s"(bits), "n"(POS | 0x10000u) : "scc");
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

// NS: the slots the instantiation keeps in registers (six words each: the A, B and C words of the thread's two window words).
// scratch: sparse_cap u32 per workgroup — the lists of the task's decoded slots.
template <int CODEC, int NS>
__global__ __launch_bounds__(PLK_WG, (PLK_WGS_PER_CU * PLK_WG + 255) / 256) void k_planes(
        const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last, const uint32_t *__restrict__ blk_off, const uint4 *__restrict__ blk_rec,
        const uint32_t *__restrict__ blk_doff, const uint32_t *__restrict__ win, const DevTerm *__restrict__ terms, const DevQuery *__restrict__ plan,
        const DevFused *__restrict__ fused, const DevTask *__restrict__ tasks, const uint32_t *__restrict__ sched, const uint32_t *__restrict__ sterms,
        const double *__restrict__ sweights, const uint32_t ntasks, uint32_t *__restrict__ ticket, uint32_t *__restrict__ counts, const uint32_t k,
        uint32_t *__restrict__ part_docs, double *__restrict__ part_scores, uint32_t *__restrict__ part_counts, const uint32_t *__restrict__ masked,
        const int sim, const uint32_t *__restrict__ planes, const uint32_t plw, const uint32_t zrow, uint32_t *__restrict__ scratch, const uint32_t sparse_cap,
        unsigned long long *__restrict__ qthr) {
        __shared__ PlanesShared sh;
        const uint32_t tid = threadIdx.x, lane = tid & 63u;
        const uint32_t wave = uni(tid >> 6);
        uint32_t *const lists = scratch + (size_t)blockIdx.x * sparse_cap;
        for (uint32_t i = tid; i < (PLK_WG / 64) * PLK_MAX_SPARSE * 2 * PLK_SW_STRIDE; i += PLK_WG)
                (&sh.pl[0][0][0])[i] = 0;
        for (uint32_t i = tid; i < 2 * PLK_SW_STRIDE; i += PLK_WG)
                sh.zero[i] = 0;
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
                __syncthreads();
                const DevFused &fq = sh.fq;
                const uint32_t nslots = min(uni(fq.nslots), (uint32_t)NS), nreq = uni(fq.nreq), negs = uni(fq.negslots);
                const uint32_t kk = min(lane, nslots - 1); // lane s (< nslots) of every wave looks after slot s, the lanes above mirror the last slot
                const uint32_t wfirst = task.tile_begin, wend = task.tile_end;
                {
                        const DevTerm myt = terms[fq.term[kk]];
                        sh.term[kk] = myt;
                        const bool dense = fq.plane[kk] != PL_NONE;
                        // what the slot's scorers add at frequency 1 and 2, and a bound of what they add at any frequency: BM25 float(w f / (f + 1.2)) < w;
                        // TF-IDF sqrt(f) w with f <= 65535; Trivial f
                        double t1 = 0.0, t2 = 0.0, ubs = 0.0;
                        bool leaf = false;
                        for (uint32_t si = 0; si < q.nscore; ++si)
                                if (sterms[q.score_base + si] == fq.term[kk]) {
                                        const double wgt = sweights[q.score_base + si];
                                        t1 += (double)sim_score(sim, wgt, 1u);
                                        t2 += (double)sim_score(sim, wgt, 2u);
                                        ubs += sim == TRI_SIM_TRIVIAL ? 65535.0 : sim == TRI_SIM_TFIDF ? (wgt > 0 ? 256.0 * wgt : 0.0) : (wgt > 0 ? wgt : 0.0);
                                        leaf = true;
                                }
                        const uint32_t top = !leaf ? 0u : dense ? 3u : 2u;
                        auto up = [](const double x) { return x > 0 ? x * (1.0 + 1e-9) : x * (1.0 - 1e-9); };
                        const double bound = fmax(ubs * (1.0 + 1e-6), fmax(up(t1), up(t2)));
                        sh.wl[kk][0] = 0.0, sh.wl[kk][1] = t1, sh.wl[kk][2] = t2, sh.wl[kk][3] = 0.0;
                        // the filter's weights: non-decreasing in the level (a negative contribution is bounded by the level below), the top one a bound
                        const double f1 = fmax(up(t1), 0.0), f2 = top == 3 ? fmax(up(t2), f1) : bound;
                        sh.wf[kk][0] = 0.0, sh.wf[kk][1] = f1, sh.wf[kk][2] = f2, sh.wf[kk][3] = bound;
                        sh.top[kk] = top;
                        const uint64_t lm = __builtin_amdgcn_ballot_w64(leaf && lane < nslots);
                        sh.leaf = (uint32_t)lm; // (same value from every lane)
                        sh.tk_n = 0;
                        {
                                // (another range of the query may have a threshold already: this one filters with it from its first window on)
                                const unsigned long long g = __atomic_load_n(gthr, __ATOMIC_RELAXED);
                                sh.tk_full = g != 0ull ? 1u : 0u;
                                sh.thr_s = g != 0ull ? key_score(g) : 0.0;
                                sh.thr_d = 0xffffffffu;
                        }
                        sh.matches = 0;
                        sh.fall = 1; // no threshold yet: every match is a candidate
                        sh.esel = 0;
                        // a decoded slot's rows that can hold documents of the task's range [first window's first docID, last window's end)
                        uint32_t row0 = 0, nrows = 0;
                        if (!dense) {
                                const uint32_t *bl = blk_last + myt.first_block;
                                const uint32_t d0 = wfirst * PL_W, d1 = wend * PL_W; // (the planner keeps max docID below 2^31: no wrap)
                                uint32_t lo = 0, hi = myt.nblocks;
                                while (lo < hi) {
                                        const uint32_t mid = (lo + hi) >> 1;
                                        if (bl[mid] < d0)
                                                lo = mid + 1;
                                        else
                                                hi = mid;
                                }
                                row0 = lo;
                                hi = myt.nblocks;
                                while (lo < hi) {
                                        const uint32_t mid = (lo + hi) >> 1;
                                        if (bl[mid] < d1)
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
                // ---- per task, uniform: which slots read term planes, where the others' LDS planes are, which slots score
                uint32_t dense_mask = 0;
                uint32_t lidx[NS], top[NS], prows[NS];
                uint32_t list_rows = 0;
                {
                        uint32_t nl = 0;
#pragma unroll
                        for (uint32_t s = 0; s < NS; ++s) {
                                const uint32_t prow = s < nslots ? uni(fq.plane[s]) : PL_NONE;
                                prows[s] = prow;
                                lidx[s] = 0;
                                top[s] = s < nslots ? uni(sh.top[s]) : 0u;
                                if (s < nslots) {
                                        if (prow != PL_NONE)
                                                dense_mask |= 1u << s;
                                        else {
                                                lidx[s] = nl++;
                                                list_rows += uni(sh.sp_n[s]);
                                        }
                                }
                        }
                }
                const uint32_t sparse_mask = ((1u << nslots) - 1u) & ~dense_mask;
                uint32_t leafm = 0; // the slots that have a scorer
#pragma unroll
                for (uint32_t s = 0; s < NS; ++s)
                        leafm |= (top[s] ? 1u : 0u) << s;
                PROF_LAP(0);
                // ---- the decoded slots' lists: every row that can reach the task's range, one lane per row, 32 entries each
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
                                planes_list_row<CODEC>(index, blk_last, blk_off, blk_rec, blk_doff, sh.term[s], sh.sp_row0[s] + r, lists + sh.sp_base[s] + r * 32u);
                        }
                }
                __syncthreads(); // (the lists are written: a workgroup barrier orders the global stores for the workgroup's own later loads)
                PROF_LAP(7);
                uint32_t my_matches = 0;
                // this wave's queue of candidates that wait for an exact frequency: it lives across the windows and is worked off 64 entries at
                // a time — every lane fetching from the postings at once, not one lane while 63 wait — and emptied at the end of the task
                uint32_t qn = 0; // entries on the queue (wave-uniform)
                auto offer = [&](const double sc, const uint32_t doc) { // false: no room (the buffer wants pruning)
                        const uint32_t slot = atomicAdd(&sh.tk_n, 1u);
                        if (slot >= PLK_CAP)
                                return false;
                        sh.tk_s[slot] = sc;
                        sh.tk_d[slot] = doc;
                        return true;
                };
                auto work_queue = [&]() { // the last (up to) 64 entries of the queue; entries that found no room go back
                        PROF_LAP(11);
                        const uint32_t take_n = min(qn, 64u), base = qn - take_n;
                        qn = base;
                        bool back = false;
                        uint32_t doc = 0, levels = 0;
                        if (lane < take_n) {
                                doc = sh.wq[wave][base + lane][0];
                                levels = sh.wq[wave][base + lane][1];
                                double sk = 0.0;
                                for (uint32_t s = 0; s < nslots; ++s) {
                                        const uint32_t l = (levels >> (2 * s)) & 3u;
                                        if (!l)
                                                continue;
                                        if (l < sh.top[s]) {
                                                sk += sh.wl[s][l];
                                                continue;
                                        }
                                        const uint32_t f = planes_lookup_freq<CODEC>(index, blk_last, blk_off, blk_rec, blk_doff, win, sh.term[s], doc);
                                        const uint32_t term = fq.term[s];
                                        for (uint32_t si = 0; si < q.nscore; ++si)
                                                if (sterms[q.score_base + si] == term)
                                                        sk += (double)sim_score(sim, sweights[q.score_base + si], f);
                                }
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
                        PROF_LAP(13);
                };
                // ---- Every WAVE walks its own contiguous share of the task's range, a sub-window of PLK_SW documents (two words per lane) at a
                //      time, wave-synchronously: no workgroup barrier inside — sixteen independent chains of loads per CU instead of two, which is
                //      what this kernel is bound by (a window is a few hundred instructions behind a memory round trip).  The waves share the
                //      candidate buffer, the threshold and the filter; they meet at a barrier only when the buffer wants pruning, and at the end.
                const uint32_t nsw = (wend - wfirst) * (PL_W / PLK_SW);
                const uint32_t sw_first = wfirst * (PL_W / PLK_SW) + (uint32_t)((uint64_t)nsw * wave / (PLK_WG / 64));
                const uint32_t sw_end = wfirst * (PL_W / PLK_SW) + (uint32_t)((uint64_t)nsw * (wave + 1) / (PLK_WG / 64));
                uint32_t sw = sw_first;
                // the wave's cursors into the decoded slots' lists: the first entry at or beyond its share's first document
                uint32_t curv[NS];
#pragma unroll
                for (uint32_t s = 0; s < NS; ++s) {
                        curv[s] = 0;
                        if ((sparse_mask >> s) & 1u)
                                curv[s] = wave_lower_bound(lists + uni(sh.sp_base[s]), 0u, uni(sh.sp_n[s]) * 32u, (sw_first * PLK_SW) << 1);
                }
                // per-task scalars the sub-window loop reads again and again, lifted out of LDS once: the lists' extents, the groups' slot sets, the
                // term planes' bases (the loads take a scalar base plus the lane's offset)
                uint32_t sp_n32[NS], sp_off[NS], gsl[FUS_MAX_SLOTS];
                const uint32_t *pA[NS];
#pragma unroll
                for (uint32_t s = 0; s < NS; ++s) {
                        sp_n32[s] = ((sparse_mask >> s) & 1u) ? uni(sh.sp_n[s]) * 32u : 0u;
                        sp_off[s] = ((sparse_mask >> s) & 1u) ? uni(sh.sp_base[s]) : 0u;
                        pA[s] = planes + (size_t)(prows[s] != PL_NONE ? prows[s] : zrow) * PL_PLANES * plw; // (zrow: the batch's all-zero row — what a slot without term planes reads)
                }
#pragma unroll
                for (uint32_t g = 0; g < FUS_MAX_SLOTS; ++g)
                        gsl[g] = g < nreq ? uni(fq.gslots[g]) : 0u;
                // ---- SEED: the documents of the query's rarest decoded list(s) — a few thousand at most — are scored first, each on its own:
                //      its levels in the other slots come from plane probes (head terms) and bisections of the other lists.  The top-K is mostly
                //      made of documents that hold the rare terms, so the threshold — and with it the candidate filter — is close to final before
                //      the sweep starts, instead of converging over the whole range.  The sweep then leaves those documents out of its candidates
                //      (they are counted as matches there like every other document).
                uint32_t seedmask = 0;
                {
                        uint32_t seed_items = 0;
                        for (uint32_t r = 0; r < nslots; ++r) { // ascending list length (uniform)
                                uint32_t best = 0xffffffffu, bn = 0xffffffffu;
#pragma unroll
                                for (uint32_t s = 0; s < NS; ++s)
                                        if (((sparse_mask >> s) & 1u) && !((seedmask >> s) & 1u) && !((negs >> s) & 1u) && top[s] && sp_n32[s] && sp_n32[s] < bn) {
                                                bn = sp_n32[s];
                                                best = s;
                                        }
                                if (best == 0xffffffffu || (seedmask ? seed_items + bn > PLK_SEED_MORE : bn > PLK_SEED_FIRST))
                                        break;
                                seedmask |= 1u << best;
                                seed_items += bn;
                        }
                        const uint32_t d_lo = wfirst * PL_W, d_hi = wend * PL_W;
                        const bool full0 = false;
                        (void)full0;
                        for (uint32_t v0 = 0; v0 < seed_items; v0 += PLK_WG) { // (uniform trip count)
                                // this thread's item: an entry of a seed slot's list
                                const uint32_t v = v0 + tid;
                                uint32_t ss = 0, ei = v;
                                bool pend = v < seed_items;
#pragma unroll
                                for (uint32_t s = 0; s < NS; ++s)
                                        if ((seedmask >> s) & 1u) {
                                                if (ss == s && ei >= sp_n32[s]) {
                                                        ei -= sp_n32[s];
                                                        ss = s + 1;
                                                }
                                        } else if (ss == s)
                                                ss = s + 1;
                                // (ss: the first seed slot whose entries are not all before item v)
                                uint32_t e = PLK_PAD;
                                if (pend && ss < NS) {
                                        uint32_t off = 0;
#pragma unroll
                                        for (uint32_t s = 0; s < NS; ++s)
                                                off = ss == s ? sp_off[s] : off;
                                        e = lists[off + ei];
                                }
                                const uint32_t doc = e >> 1;
                                pend = pend && e != PLK_PAD && doc >= d_lo && doc < d_hi && doc != 0;
                                double score = 0.0;
                                if (pend) {
                                        // the document's level in every slot
                                        uint32_t present = 0, levels = 0;
#pragma unroll
                                        for (uint32_t s = 0; s < NS; ++s) {
                                                if (s >= nslots)
                                                        continue;
                                                uint32_t l = 0;
                                                if (prows[s] != PL_NONE) {
                                                        const uint32_t *pa = pA[s], *pb = pa + plw, *pc = pb + plw;
                                                        const uint32_t wi = doc >> 5, bit = doc & 31u;
                                                        l = ((pa[wi] >> bit) & 1u) + ((pb[wi] >> bit) & 1u) + ((pc[wi] >> bit) & 1u);
                                                } else if (s == ss)
                                                        l = 1u + (e & 1u);
                                                else if (sp_n32[s]) { // another decoded list: bisect it
                                                        const uint32_t *ls = lists + sp_off[s];
                                                        uint32_t lo = 0, hi = sp_n32[s];
                                                        const uint32_t key = doc << 1;
                                                        while (lo < hi) {
                                                                const uint32_t mid = (lo + hi) >> 1;
                                                                if (ls[mid] < key)
                                                                        lo = mid + 1;
                                                                else
                                                                        hi = mid;
                                                        }
                                                        const uint32_t f = lo < sp_n32[s] ? ls[lo] : PLK_PAD;
                                                        l = (f >> 1) == doc ? 1u + (f & 1u) : 0u;
                                                }
                                                present |= (l ? 1u : 0u) << s;
                                                levels |= (top[s] ? l : 0u) << (2 * s);
                                        }
                                        // a document that an earlier seed slot holds is that slot's item; the predicate: every required group, no excluded slot, not masked
                                        bool ok = !(present & seedmask & ((1u << ss) - 1u)) && !(present & negs);
#pragma unroll
                                        for (uint32_t g = 0; g < FUS_MAX_SLOTS; ++g)
                                                ok = ok && (g >= nreq || (present & gsl[g]) != 0);
                                        if (ok && masked)
                                                ok = !((masked[doc >> 5] >> (doc & 31u)) & 1u);
                                        pend = ok;
                                        if (ok)
                                                for (uint32_t s = 0; s < nslots; ++s) {
                                                        const uint32_t l = (levels >> (2 * s)) & 3u;
                                                        if (!l)
                                                                continue;
                                                        if (l < sh.top[s]) {
                                                                score += sh.wl[s][l];
                                                                continue;
                                                        }
                                                        const uint32_t f = planes_lookup_freq<CODEC>(index, blk_last, blk_off, blk_rec, blk_doff, win, sh.term[s], doc);
                                                        const uint32_t term = fq.term[s];
                                                        for (uint32_t si = 0; si < q.nscore; ++si)
                                                                if (sterms[q.score_base + si] == term)
                                                                        score += (double)sim_score(sim, sweights[q.score_base + si], f);
                                                }
                                }
                                // offered in rounds: a full buffer is pruned in between
                                for (;;) {
                                        if (pend && (!uni(sh.tk_full) || better(score, doc, sh.thr_s, sh.thr_d)))
                                                pend = !offer(score, doc);
                                        else
                                                pend = false;
                                        const uint32_t anyp = (uint32_t)__syncthreads_or(pend ? 1 : 0);
                                        const uint32_t n = min(uni(sh.tk_n), PLK_CAP);
                                        __syncthreads(); // (every lane has read tk_n)
                                        if (n >= PLK_PRUNE_AT)
                                                planes_prune(sh, n, k, gthr);
                                        if (!uni(anyp))
                                                break;
                                }
                        }
                        planes_prune(sh, min(uni(sh.tk_n), PLK_CAP), k, gthr); // (no seeds: still the query's shared threshold, if there is one)
                        planes_filter(sh, nslots);
                }
                uint32_t c0 = 0, c1 = 0, rows_mask = 0; // the current sub-window's candidates (per lane) and the decoded slots that put something into its LDS planes
                bool open = false;                      // the current sub-window has been swept (its candidates are being worked off)
                for (;;) {
                        // (threshold and filter move only at a prune, i.e. behind the barrier below: read once per stretch)
                        const bool full = uni(sh.tk_full) != 0;
                        const double thr_s = sh.thr_s;
                        const uint32_t thr_d = sh.thr_d;
                        const uint32_t esel = uni(sh.esel);
                        uint32_t es_a = 0, es_b = 0, es_c = 0; // the essential planes as bit sets over the slots
#pragma unroll
                        for (uint32_t s = 0; s < NS; ++s) {
                                const uint32_t e = (esel >> (2 * s)) & 3u;
                                es_a |= (e == 0 ? 1u : 0u) << s;
                                es_b |= (e == 1 ? 1u : 0u) << s;
                                es_c |= (e == 2 ? 1u : 0u) << s;
                        }
                        const bool fall = uni(sh.fall) != 0;
                        while (sw < sw_end) {
                                if (uni(__atomic_load_n(&sh.tk_n, __ATOMIC_RELAXED)) >= PLK_PRUNE_AT)
                                        break; // the buffer wants pruning first: to the barrier (the sub-window stays as it is)
                                const uint32_t w0 = sw * PLK_SW, wE = w0 + PLK_SW;
                                // The level words of one of this lane's two words of the sub-window (which: 0 / 1): a = plane A, b = plane B (frequency not 1),
                                // c = plane C (nor 2) — a head term's from the batch's term planes, a decoded slot's from the wave's LDS planes.  Fetched when
                                // needed (the sweep; a candidate's scoring) and not kept.  Straight-line on purpose: every load is issued — a slot without a
                                // term plane reads row 0's words, one without LDS planes reads slot 0's, and the uniform selects drop them — so that one wait
                                // covers them all; every address is a scalar base plus the lane's offset.  (Round 4 measured the other way — words b / c loaded
                                // only where word a has a bit: cfg3 10.5 -> 14.4 ms; the branch and the second wait cost more than the loads they save.)
                                auto plane_words = [&](const uint32_t which, uint32_t (&ga)[NS], uint32_t (&gb)[NS], uint32_t (&gc)[NS]) { // the term planes' part: global loads
                                        const uint32_t wi = lane + which * 64u;
#pragma unroll
                                        for (uint32_t s = 0; s < NS; ++s) {
                                                const uint32_t gi = prows[s] != PL_NONE ? (w0 >> 5) + wi : wi; // (a plane is far below 2^32 bytes: a 32-bit offset)
                                                const uint32_t *pa = pA[s], *pb = pa + plw, *pc = pb + plw;
                                                ga[s] = pa[gi];
                                                gb[s] = pb[gi];
                                                gc[s] = pc[gi];
                                        }
                                };
                                auto finish_words = [&](const uint32_t which, const uint32_t (&ga)[NS], const uint32_t (&gb)[NS], const uint32_t (&gc)[NS], uint32_t (&a)[NS],
                                                        uint32_t (&b)[NS], uint32_t (&c)[NS]) { // ... the decoded slots' part (LDS)
                                        const uint32_t wi = lane + which * 64u;
                                        // (no selects: a slot with term planes reads the zero LDS plane here, a decoded slot read the batch's zero row there)
#pragma unroll
                                        for (uint32_t s = 0; s < NS; ++s) {
                                                a[s] = ga[s];
                                                b[s] = gb[s];
                                                c[s] = gc[s];
                                        }
                                        if (rows_mask) { // (uniform: a sub-window no decoded list reaches reads no LDS)
#pragma unroll
                                                for (uint32_t s = 0; s < NS; ++s) {
                                                        const uint32_t *lp = ((sparse_mask >> s) & 1u) ? &sh.pl[wave][lidx[s]][0] : &sh.zero[0];
                                                        a[s] |= lp[wi];
                                                        b[s] |= lp[PLK_SW_STRIDE + wi];
                                                }
                                        }
                                };
                                auto level_words = [&](const uint32_t which, uint32_t (&a)[NS], uint32_t (&b)[NS], uint32_t (&c)[NS]) {
                                        uint32_t ga[NS], gb[NS], gc[NS];
                                        plane_words(which, ga, gb, gc);
                                        finish_words(which, ga, gb, gc, a, b, c);
                                };
                                // One step over the candidates `cw` of one word: every lane that has one takes its lowest, scores it from the level words
                                // a / b / c — the known part of the score and a bound for the rest — and offers it, queues it (a slot of unknown
                                // frequency that the bound does not rule out) or drops it.  false: the buffer had no room (the candidate stays).
                                auto candidate_step = [&](const uint32_t which, const uint32_t (&a)[NS], const uint32_t (&b)[NS], const uint32_t (&c)[NS], uint32_t &cw) {
                                        bool enq = false;
                                        uint32_t edoc = 0, elev = 0;
                                        if (cw) {
                                                const uint32_t bit = (uint32_t)__builtin_ctz(cw);
                                                const uint32_t doc = w0 + 32u * (lane + which * 64u) + bit;
                                                double sk = 0.0, sb = 0.0; // the known part of the score; bounds of the slots whose frequency is not known
                                                uint32_t levels = 0;
                                                bool unk = false;
#pragma unroll
                                                for (uint32_t s = 0; s < NS; ++s) {
                                                        if (!top[s] || !((a[s] >> bit) & 1u))
                                                                continue;
                                                        const uint32_t bb = (b[s] >> bit) & 1u, cc = (c[s] >> bit) & 1u;
                                                        const uint32_t l = 1u + bb + (bb & cc);
                                                        levels |= l << (2 * s);
                                                        if (l < top[s])
                                                                sk += sh.wl[s][l];
                                                        else {
                                                                unk = true;
                                                                sb += sh.wf[s][l];
                                                        }
                                                }
                                                bool done = true;
                                                if (!full || better(sk + sb, doc, thr_s, thr_d)) {
                                                        if (unk) {
                                                                enq = true;
                                                                edoc = doc;
                                                                elev = levels;
                                                        } else
                                                                done = offer(sk, doc); // (no room: the candidate stays for after the prune)
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
                                };
                                bool stuck = false; // (the buffer filled up under a candidate)
                                if (!open) {
                                        // (the term planes' words of both of this lane's words travel together with the lists' entries: one round trip)
                                        uint32_t g0a[NS], g0b[NS], g0c[NS], g1a[NS], g1b[NS], g1c[NS];
                                        plane_words(0, g0a, g0b, g0c);
                                        plane_words(1, g1a, g1b, g1c);
                                        // ---- set pass: the decoded slots' entries of this sub-window, 64 per slot and round, into the wave's LDS planes
                                        rows_mask = 0;
                                        for (;;) {
                                                uint32_t ent[NS];
                                                bool more = false;
#pragma unroll
                                                for (uint32_t s = 0; s < NS; ++s) { // (all the slots' loads first: one round trip)
                                                        ent[s] = PLK_PAD;
                                                        if (!((sparse_mask >> s) & 1u))
                                                                continue;
                                                        if (curv[s] >= sp_n32[s])
                                                                continue;
                                                        rows_mask |= 1u << s;
                                                        if (curv[s] + lane < sp_n32[s])
                                                                ent[s] = lists[sp_off[s] + curv[s] + lane];
                                                }
#pragma unroll
                                                for (uint32_t s = 0; s < NS; ++s) {
                                                        if (!((rows_mask >> s) & 1u))
                                                                continue;
                                                        const uint32_t e = ent[s], d = e >> 1;
                                                        const bool before = d < wE; // (ascending: the entries below the sub-window's end are a prefix of the chunk)
                                                        if (before && d >= w0) {
                                                                uint32_t *p = &sh.pl[wave][lidx[s]][0];
                                                                const uint32_t r = d - w0, bit = 1u << (r & 31u);
                                                                atomicOr(&p[r >> 5], bit);
                                                                if (e & 1u)
                                                                        atomicOr(&p[PLK_SW_STRIDE + (r >> 5)], bit);
                                                        }
                                                        const uint32_t cnt = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(before));
                                                        curv[s] += cnt;
                                                        more |= cnt == 64; // (the whole chunk lies below the sub-window's end: there may be more)
                                                }
                                                if (!more)
                                                        break;
                                        }
                                        PROF_LAP(8);
                                        // ---- sweep, one word at a time: the predicate, then the candidate filter (planes_filter) on the level words
                                        uint32_t cand[2];
                                        bool later = false; // (the queue or the buffer is full: what is left of the candidates goes through the resume path)
#pragma unroll
                                        for (uint32_t which = 0; which < 2; ++which) {
                                                uint32_t a[NS], b[NS], c[NS];
                                                if (which)
                                                        finish_words(1, g1a, g1b, g1c, a, b, c);
                                                else
                                                        finish_words(0, g0a, g0b, g0c, a, b, c);
                                                // (the slots' roles are uniform bit sets: a role's word is and_or'ed together under scalar masks — one vector
                                                //  instruction per slot and role, where a select costs two and a scalar one)
                                                uint32_t m = 0xffffffffu;
#pragma unroll
                                                for (uint32_t g = 0; g < FUS_MAX_SLOTS; ++g) {
                                                        if (g >= nreq)
                                                                break;
                                                        uint32_t x = 0;
                                                        static_for<NS>([&](auto S) { x = and_or(a[S], umask_at<S>(gsl[g]), x); });
                                                        m &= x;
                                                }
                                                if (negs) {
                                                        uint32_t nx = 0;
                                                        static_for<NS>([&](auto S) { nx = and_or(a[S], umask_at<S>(negs), nx); });
                                                        m &= ~nx;
                                                }
                                                if (masked) // masked_documents_registry::test (docidupdates.h:90-119)
                                                        m &= ~masked[(w0 >> 5) + lane + which * 64u];
                                                my_matches += (uint32_t)__popc(m);
                                                // the candidate filter: the documents in an essential plane, word-wise; each of them then with its level
                                                // vector in the table (one per lane and step)
                                                uint32_t seeded = 0; // (documents the seed pass has scored: never candidates here)
                                                if (seedmask) {
                                                        static_for<NS>([&](auto S) { seeded = and_or(a[S], umask_at<S>(seedmask), seeded); });
                                                }
                                                uint32_t cw = m & ~seeded;
                                                if (!fall) {
                                                        uint32_t ew = 0;
                                                        static_for<NS>([&](auto S) { ew = and_or(a[S], umask_at<S>(es_a), ew); });
                                                        if (es_b) {
                                                                static_for<NS>([&](auto S) { ew = and_or(b[S], umask_at<S>(es_b), ew); });
                                                        }
                                                        if (es_c) {
                                                                static_for<NS>([&](auto S) { ew = and_or(c[S], umask_at<S>(es_c), ew); });
                                                        }
                                                        ew &= m & ~seeded;
                                                        cw = 0;
                                                        if (__builtin_amdgcn_ballot_w64(ew != 0) != 0ull) {
                                                        // the planes are nested, the level is the number of them a document is in: its two bits, word-wise
                                                        uint32_t lo[NS], hi[NS];
#pragma unroll
                                                        for (uint32_t s = 0; s < NS; ++s) {
                                                                lo[s] = ((leafm >> s) & 1u) ? a[s] ^ b[s] ^ c[s] : 0u; // (a slot without a scorer: level 0)
                                                                hi[s] = ((leafm >> s) & 1u) ? b[s] : 0u;
                                                        }
                                                        do {
                                                                const uint32_t bit = ew ? (uint32_t)__builtin_ctz(ew) : 0u;
                                                                uint32_t code = 0;
#pragma unroll
                                                                for (uint32_t s = 0; s < NS; ++s)
                                                                        code |= (((lo[s] >> bit) & 1u) << (2 * s)) | (((hi[s] >> bit) & 1u) << (2 * s + 1));
                                                                const uint32_t hit = (sh.ftab[code >> 5] >> (code & 31u)) & 1u;
                                                                cw |= ew ? hit << bit : 0u;
                                                                ew &= ew - 1u;
                                                                PROF_COUNT(20, lane == 0 ? 1 : 0);
                                                        } while (__builtin_amdgcn_ballot_w64(ew != 0) != 0ull);
                                                        }
                                                }
                                                // the word's candidates are worked off right here, while its level words are in registers (a sub-window
                                                // of a union has one or two: a step of their own, with the words fetched again, cost more than the sweep)
                                                // (no call in here — the queue is worked off, and a full buffer waited out, in the resume path below: a call
                                                //  among the sweep's live registers made the compiler spill them on the hot path)
                                                PROF_LAP(9);
                                                while (!later && __builtin_amdgcn_ballot_w64(cw != 0) != 0ull) {
                                                        if (qn >= 64 || uni(__atomic_load_n(&sh.tk_n, __ATOMIC_RELAXED)) >= PLK_PRUNE_AT) {
                                                                later = true;
                                                                PROF_COUNT(23, lane == 0 ? 1 : 0);
                                                        }
                                                        else
                                                                candidate_step(which, a, b, c, cw);
                                                }
                                                cand[which] = cw;
                                                PROF_LAP(10);
                                        }
                                        c0 = cand[0], c1 = cand[1];
                                        open = true;
                                        PROF_COUNT(19, lane == 0 ? 1 : 0);
                                }
                                // ---- candidates left over from before a prune: the same steps, with the level words fetched again
                                while (!stuck && __builtin_amdgcn_ballot_w64((c0 | c1) != 0) != 0ull) {
                                        if (uni(__atomic_load_n(&sh.tk_n, __ATOMIC_RELAXED)) >= PLK_PRUNE_AT) {
                                                stuck = true;
                                                break;
                                        }
                                        if (qn >= 64) { // (a step may add 64 entries: the queue is kept below 64 before it)
                                                work_queue();
                                                continue;
                                        }
                                        const uint32_t which = c0 ? 0u : 1u; // (lanes without a candidate fetch word 1 and drop it)
                                        uint32_t a[NS], b[NS], c[NS];
                                        level_words(which, a, b, c);
                                        uint32_t cw = which ? c1 : c0;
                                        candidate_step(which, a, b, c, cw);
                                        if (which)
                                                c1 = cw;
                                        else
                                                c0 = cw;
                                }
                                PROF_LAP(11);
                                if (stuck)
                                        break;
                                // the sub-window is done: its LDS planes are cleared for the next one
#pragma unroll
                                for (uint32_t s = 0; s < NS; ++s)
                                        if ((rows_mask >> s) & 1u) {
                                                uint32_t *p = &sh.pl[wave][lidx[s]][0];
                                                p[lane] = 0;
                                                p[lane + 64] = 0;
                                                p[PLK_SW_STRIDE + lane] = 0;
                                                p[PLK_SW_STRIDE + lane + 64] = 0;
                                        }
                                open = false;
                                ++sw;
                                PROF_LAP(12);
                        }
                        PROF_LAP(4);
                        // ---- the waves meet: prune if the buffer wants it, go on while any of them has sub-windows left
                        sh.flag[wave] = sw < sw_end ? 1u : 0u; // (wave-uniform value, every lane stores it)
                        __syncthreads();
                        uint32_t anyp = 0;
#pragma unroll
                        for (uint32_t wv = 0; wv < PLK_WG / 64; ++wv)
                                anyp |= sh.flag[wv];
                        anyp = uni(anyp);
                        const uint32_t n = min(uni(sh.tk_n), PLK_CAP);
                        __syncthreads(); // (every lane has read the flags and tk_n)
                        if (n >= PLK_PRUNE_AT) {
                                planes_prune(sh, n, k, gthr);
                                planes_filter(sh, nslots);
                                PROF_COUNT(18, tid == 0 ? 1 : 0);
                        }
                        PROF_LAP(5);
                        if (!anyp)
                                break;
                }
                // ---- the candidates still waiting for their frequencies
                for (;;) {
                        while (qn && uni(__atomic_load_n(&sh.tk_n, __ATOMIC_RELAXED)) < PLK_PRUNE_AT)
                                work_queue();
                        sh.flag[wave] = qn ? 1u : 0u; // (wave-uniform value, every lane stores it)
                        __syncthreads();
                        uint32_t anyp = 0;
#pragma unroll
                        for (uint32_t wv = 0; wv < PLK_WG / 64; ++wv)
                                anyp |= sh.flag[wv];
                        anyp = uni(anyp);
                        const uint32_t n = min(uni(sh.tk_n), PLK_CAP);
                        __syncthreads(); // (every lane has read the flags and tk_n)
                        if (n >= PLK_PRUNE_AT)
                                planes_prune(sh, n, k, gthr);
                        if (!anyp)
                                break;
                }
                // ---- the task's result: its best k (ranked) and its match count
                __syncthreads();
                planes_prune(sh, min(uni(sh.tk_n), PLK_CAP), k, gthr);
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1)
                        my_matches += __shfl_xor(my_matches, d, 64);
                atomicAdd(&sh.matches, lane == 0 ? my_matches : 0u); // (every lane issues it: no single-lane branch)
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
