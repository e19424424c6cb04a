// k_planes.hpp — term planes (a head term decoded once per launch for every query of the batch that names it) and the
// AccumulatedScoreScheme top-K kernel that runs over them
// Part of libtrinity_hip.so (MI355X / gfx950); included by trinity_hip.hip.  New code, no reference source.
#pragma once
#include "k_fused.hpp"

// Under Zipf a handful of terms carry most of a batch's postings (at the 10M-document configuration the 40 most frequent terms
// hold 98 % of the postings the 5-term queries of SURVEY §8(d) cfg3 touch), and the reference decodes such a list again for every
// query that names it (Decoder::init + next() per query, google_codec.cpp:777-819 / lucene_codec.cpp:568-594).  Here every LAUNCH
// decodes each of those lists ONCE — k_term_planes, inside the timed region, from the segment's own codec bytes — into two
// bitmaps over the docID space:
//     plane A   bit d set  <=>  document d holds the term            (PostingsListIterator::current() would stop on d)
//     plane B   bit d set  <=>  ... and its frequency there is not 1 (the exact frequency is then read from the postings on demand)
// and the matching kernels read the planes: k_and tests a candidate with one bit probe instead of bracketing and decoding a block
// (Conjuction::next_impl's advance(), docset_iterators.cpp:308-348), k_and_dense ORs a plane's words into its window bitmap instead
// of walking the term's rows (docset_spans.cpp:98-173), k_planes (below) evaluates union / CNF predicates 32 documents per word.
// The planes live in a scratch region owned by the batch (2 x (max docID / 8) bytes per term); nothing survives the launch.

constexpr uint32_t PL_CELLS = PL_W / CELL_DOCS; // cell-index entries per plane window
constexpr uint32_t PL_STRIDE = PL_WORDS + 32;   // LDS words between a slot's A and B plane (word PL_WORDS of each: the sink)

// A decoded posting into a pair of LDS planes (A at a[], B at a[PL_STRIDE]).  Documents outside the window land in the sink word.
struct PlanePost {
        uint32_t *a;
        __device__ __forceinline__ void doc(const uint32_t rel) {
                const uint32_t r = min(rel, PL_W);
                atomicOr(&a[r >> 5], 1u << (r & 31u));
        }
        __device__ __forceinline__ void operator()(const uint32_t rel, const uint32_t f) {
                const uint32_t r = min(rel, PL_W);
                const uint32_t bit = 1u << (r & 31u);
                atomicOr(&a[r >> 5], bit);
                if ((f & 0xffffu) != 1u) // (the frequency a scorer sees is tokenpos_t, 16 bits: codecs.h:217)
                        atomicOr(&a[(r >> 5) + PL_STRIDE], bit);
        }
};

// One workgroup per (plane row, window): the rows (<= 32 documents each) of the term that reach the window are decoded, one lane
// per row, into LDS planes, which are then written out whole — every word of both planes is written by exactly one workgroup,
// so the scratch region needs no clearing between launches.
template <int CODEC>
__global__ __launch_bounds__(AND_WG) void k_term_planes(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last,
                                                        const uint32_t *__restrict__ blk_off, const uint4 *__restrict__ blk_rec,
                                                        const uint32_t *__restrict__ blk_doff, const uint32_t *__restrict__ win,
                                                        const DevTerm *__restrict__ terms, const uint32_t *__restrict__ plane_terms,
                                                        uint32_t *__restrict__ planes, const uint32_t plw) {
        __shared__ uint32_t pl[2 * PL_STRIDE];
        const uint32_t tid = threadIdx.x, w = blockIdx.x, row = blockIdx.y;
        for (uint32_t i = tid; i < 2 * PL_STRIDE; i += AND_WG)
                pl[i] = 0;
        const DevTerm t = terms[plane_terms[row]];
        const uint32_t *bl = blk_last + t.first_block;
        const uint32_t w0 = w * PL_W;
        // rows that can hold documents of [w0, w0 + PL_W): first row whose last docID >= w0 ... first row whose last docID >= the next
        // window's first docID (it may still begin inside this one)
        uint32_t b_lo, b_hi;
        if (t.win_off != 0xffffffffu) {
                b_lo = win[t.win_off + w * PL_CELLS];
                b_hi = win[t.win_off + (w + 1) * PL_CELLS];
        } else { // (planes are made for long lists, which are indexed; kept for completeness)
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
        uint32_t *pa = planes + (size_t)row * 2 * plw + (size_t)w * PL_WORDS, *pb = pa + plw;
        for (uint32_t i = tid; i < PL_WORDS; i += AND_WG) {
                pa[i] = pl[i];
                pb[i] = pl[PL_STRIDE + i];
        }
}

// ------------------------------------------------------------------------------------------ k_planes
// AccumulatedScoreScheme + top-K of a CNF query (a union, a conjunction of terms / OR-groups, an excluded group, optional scoring
// terms: everything k_fused's CNF instantiations take) in one pass over windows of PL_W documents, on BIT PLANES instead of a word
// per document:
//   * every slot (distinct term) of the query presents, per window, plane A (the document holds the term) and plane B (its
//     frequency is not 1).  A head term's planes come straight from the batch's term planes (global memory, L2 / Infinity-Cache
//     resident: k_term_planes decoded the list once for every query of the launch); any other term's rows that reach the window
//     are decoded into LDS planes (one lane per row of <= 32 documents, the same row readers as k_fused).
//   * the predicate is word-wise: a required group = the OR of its slots' A words, the conjunction their AND, the excluded group an
//     AND-NOT, masked documents (docidupdates.h:90-119) another — 32 documents per instruction; the match count is a popcount.
//     (What docset_spans.cpp:98-173 / 681-790 do per document and docset_iterators.cpp:226-405 per posting.)
//   * the candidate filter is word-wise too.  A slot is at one of three LEVELS in a document: absent, frequency 1 (its scorers add
//     exactly tab1), any other frequency (they add at most ub).  Whenever the threshold (the k-th best score so far) moves, the
//     minimal level assignments whose bounds reach it are listed (planes_filter); a match is a candidate iff it meets one of them
//     — an OR of ANDs over the slots' A / B words.  Matches whose frequencies are all 1 (most of them) are thereby tested against
//     their EXACT score without ever being touched individually.  (This replaces MaxScore's "holds an essential slot".)
//   * candidates are scored one per lane by the wave that owns their words, no workgroup barrier: levels from registers, the
//     exact frequency of a level-2 slot from the postings (directory cell -> block -> walk) only when the bound does not rule the
//     document out.  A slot no assignment needs is decoded for presence only: the freqs group of its PFOR blocks is not addressed.
//   * per task: min(matches, k) ranked (docID, score) pairs and the match count; k_topk_merge folds a query's tasks.
constexpr int PLK_WG = 512;
constexpr uint32_t PLK_MAX_SPARSE = 6;  // slots whose lists are decoded per window (LDS planes); the planner sends wider queries to k_fused
constexpr uint32_t PLK_CAP = 512;       // candidate buffer (one entry per thread when it is pruned)
constexpr uint32_t PLK_PRUNE_AT = 384;  // the waves stop taking candidates once it holds this many: it is pruned to the best k, then they resume
constexpr uint32_t PLK_MAXPAT = 32;     // level assignments of the candidate filter kept as such (more: one per essential slot)
constexpr uint32_t PLK_WGS_PER_CU = 2;
constexpr uint32_t PLK_WQ = 128;        // per-wave queue of candidates waiting for a frequency lookup: worked off 64 at a time, every lane busy
constexpr uint32_t PLK_NS_SMALL = 5;    // the instantiation for queries of up to this many slots keeps four words per slot in registers
static_assert(PL_WORDS == 2 * PLK_WG, "the sweep gives every thread two words of the window");
static_assert(PLK_CAP == PLK_WG && TOPK_MAX < PLK_PRUNE_AT && PLK_PRUNE_AT < PLK_CAP, "pruning leaves room; a pruned buffer is below the stop mark");

struct PlanesShared {
        uint32_t pl[PLK_MAX_SPARSE][2 * PL_STRIDE]; // per decoded slot: plane A, plane B (word PL_WORDS of each: sink)
        double tk_s[PLK_CAP];
        uint32_t tk_d[PLK_CAP];
        DevTerm term[FUS_MAX_SLOTS];
        double tab1[FUS_MAX_SLOTS]; // per slot: what its scorers add at frequency 1
        double ub[FUS_MAX_SLOTS];   // per slot: an upper bound of what they add at any frequency
        double w1[FUS_MAX_SLOTS];   // per slot: tab1 rounded up a hair (the filter must never lose a tie to rounding)
        double thr_s;
        uint32_t thr_d;
        uint32_t tk_n, tk_full, matches, ess; // ess: the slots whose frequencies are worth decoding (some assignment names them)
        uint32_t leaf;                        // the slots that have a scorer
        uint32_t npat;                        // the candidate filter: 0xffffffff = every match (no threshold yet), else that many assignments
        uint32_t pat[PLK_MAXPAT];             // ... bits 0-7: slots at level >= 1, bits 8-15: slots at level 2
        uint32_t flag[PLK_WG / 64];
        uint32_t wq[PLK_WG / 64][PLK_WQ]; // per wave: candidates waiting for exact frequencies (rel docID << 16 | level-2 slots << 8 | slots held)
        uint32_t bcast[4];
        uint32_t rng_lo[2][FUS_MAX_SLOTS], rng_cnt[2][FUS_MAX_SLOTS]; // per window parity: the decoded slots' row ranges
        uint32_t alive[2];                                             // ... and the slots whose lists are not exhausted
        uint32_t hint_row[FUS_MAX_SLOTS], hint_doc[FUS_MAX_SLOTS];     // a row that reached beyond its window: its first document past it
        DevFused fq;
};
static_assert(sizeof(PlanesShared) * PLK_WGS_PER_CU <= 160u * 1024u, "two workgroups per CU");

// PlanePost that also keeps the row's first document past the window (k_fused's hint: a sparse list's row is decoded once, not once
// per window it spans)
struct PlanePostH {
        uint32_t *a;
        uint32_t past = 0xffffffffu;
        __device__ __forceinline__ void doc(const uint32_t rel) {
                past = min(past, rel - PL_W);
                const uint32_t r = min(rel, PL_W);
                atomicOr(&a[r >> 5], 1u << (r & 31u));
        }
        __device__ __forceinline__ void operator()(const uint32_t rel, const uint32_t f) {
                past = min(past, rel - PL_W);
                const uint32_t r = min(rel, PL_W);
                const uint32_t bit = 1u << (r & 31u);
                atomicOr(&a[r >> 5], bit);
                if ((f & 0xffffu) != 1u)
                        atomicOr(&a[(r >> 5) + PL_STRIDE], bit);
        }
};

// Keep the best k of the n (<= PLK_CAP = PLK_WG) buffered candidates, best first (rank by counting: the order is strict).
__device__ void planes_prune(PlanesShared &sh, const uint32_t n, const uint32_t k) {
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
        // uniform stores by every lane
        sh.tk_n = m;
        if (m == k) {
                sh.tk_full = 1;
                sh.thr_s = sh.tk_s[k - 1];
                sh.thr_d = sh.tk_d[k - 1];
        }
        __syncthreads();
}

// MaxScore's essential slots — the fallback filter: with the slots ordered by their score bound, the longest prefix whose bounds sum to
// less than the k-th best cannot lift a document over it; a bit set of the OTHER slots, same value in every lane.
__device__ __forceinline__ uint32_t planes_essential(const PlanesShared &sh, const uint32_t nslots) {
        const double thr = sh.thr_s;
        uint32_t done = 0, ess = 0;
        double p = 0.0;
        for (uint32_t r = 0; r < nslots; ++r) { // selection by ascending bound (<= 8 slots)
                uint32_t best = 0;
                double bv = 1e300;
                for (uint32_t sl = 0; sl < nslots; ++sl)
                        if (!((done >> sl) & 1u) && sh.ub[sl] < bv) {
                                bv = sh.ub[sl];
                                best = sl;
                        }
                done |= 1u << best;
                p += bv;
                if (!(p < thr)) // this slot (and every later one) can carry a document over the threshold
                        ess |= 1u << best;
        }
        return ess;
}

// The candidate filter, recomputed whenever the threshold moves (every thread calls it; it ends with a barrier).  Every scoring slot is
// at level 0 (absent), 1 (frequency 1: adds exactly tab1) or 2 (another frequency: adds at most ub >= tab1) in a document, so a
// document's score is at most the sum of its slots' level weights.  The MINIMAL level assignments whose weights reach the current k-th
// best score are listed (lowering any slot by one level drops below it); a match is a candidate iff it is at least at those levels for
// one of them.  3^nslots assignments, a few per thread.  No threshold yet, or one that rules nothing out: every match is a candidate.
__device__ void planes_filter(PlanesShared &sh, const uint32_t nslots) {
        const uint32_t tid = threadIdx.x;
        const double thr = sh.thr_s;
        const bool full = uni(sh.tk_full) != 0 && 0.0 < thr;
        const uint32_t leaf = uni(sh.leaf);
        sh.npat = full ? 0u : 0xffffffffu; // (uniform stores)
        sh.ess = (1u << nslots) - 1u;
        __syncthreads();
        if (!full)
                return;
        uint32_t total = 1;
        for (uint32_t sl = 0; sl < nslots; ++sl)
                total *= 3u;
        for (uint32_t a = tid; a < total; a += PLK_WG) {
                uint32_t lv[FUS_MAX_SLOTS], x = a;
                bool valid = a != 0;
                for (uint32_t sl = 0; sl < nslots; ++sl) {
                        lv[sl] = x % 3u;
                        x /= 3u;
                        valid &= lv[sl] == 0 || ((leaf >> sl) & 1u); // (a slot without a scorer adds nothing at any level)
                }
                if (!valid)
                        continue;
                auto reaches = [&](const uint32_t lowered) { // (lowered: the slot taken down one level; nslots: none)
                        double sum = 0.0;
                        for (uint32_t sl = 0; sl < nslots; ++sl) {
                                const uint32_t l = lv[sl] - (sl == lowered ? 1u : 0u);
                                sum += l == 2 ? sh.ub[sl] : l == 1 ? sh.w1[sl] : 0.0;
                        }
                        return !(sum < thr);
                };
                bool minimal = reaches(nslots);
                for (uint32_t sl = 0; sl < nslots && minimal; ++sl)
                        if (lv[sl] && reaches(sl))
                                minimal = false;
                if (minimal) {
                        uint32_t m1 = 0, m2 = 0;
                        for (uint32_t sl = 0; sl < nslots; ++sl) {
                                m1 |= (lv[sl] ? 1u : 0u) << sl;
                                m2 |= (lv[sl] == 2 ? 1u : 0u) << sl;
                        }
                        const uint32_t at = atomicAdd(&sh.npat, 1u);
                        if (at < PLK_MAXPAT)
                                sh.pat[at] = m1 | m2 << 8;
                }
        }
        __syncthreads();
        uint32_t np = uni(sh.npat);
        if (np > PLK_MAXPAT) {
                // too many assignments: one level only — the minimal SETS of slots whose bounds reach the threshold (a weaker filter,
                // never a wrong one; at most C(8, 4) = 70 of them, 10 for five slots)
                __syncthreads(); // (every lane has read npat)
                sh.npat = 0;
                __syncthreads();
                if (tid && tid < (1u << nslots) && !(tid & ~leaf)) {
                        auto reaches = [&](const uint32_t pset) {
                                double sum = 0.0;
                                for (uint32_t sl = 0; sl < nslots; ++sl)
                                        if ((pset >> sl) & 1u)
                                                sum += sh.ub[sl];
                                return !(sum < thr);
                        };
                        bool minimal = reaches(tid);
                        for (uint32_t sl = 0; sl < nslots && minimal; ++sl)
                                if (((tid >> sl) & 1u) && reaches(tid & ~(1u << sl)))
                                        minimal = false;
                        if (minimal) {
                                const uint32_t at = atomicAdd(&sh.npat, 1u);
                                if (at < PLK_MAXPAT)
                                        sh.pat[at] = tid;
                        }
                }
                __syncthreads();
                np = uni(sh.npat);
                PROF_COUNT(21, tid == 0 ? 1 : 0);
        }
        if (np > PLK_MAXPAT) { // still too many: the essential slots, one set each
                const uint32_t e = planes_essential(sh, nslots);
                __syncthreads(); // (every lane has read npat)
                np = 0;
                for (uint32_t sl = 0; sl < nslots; ++sl)
                        if ((e >> sl) & 1u)
                                sh.pat[np++] = 1u << sl; // (uniform stores)
                sh.npat = np;
                __syncthreads();
                PROF_COUNT(22, tid == 0 ? 1 : 0);
        }
        uint32_t need = 0;
        for (uint32_t i = 0; i < np; ++i)
                need |= sh.pat[i] & 0xffu;
        sh.ess = uni(need);
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

// NS: the slots the instantiation keeps in registers (four words each: A and level-2 words of the thread's two window words)
template <int CODEC, int NS>
__global__ __launch_bounds__(PLK_WG, (PLK_WGS_PER_CU * PLK_WG + 255) / 256) void k_planes(
        const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last, const uint32_t *__restrict__ blk_off, const uint4 *__restrict__ blk_rec,
        const uint32_t *__restrict__ blk_doff, const uint32_t *__restrict__ win, const DevTerm *__restrict__ terms, const DevQuery *__restrict__ plan,
        const DevFused *__restrict__ fused, const DevTask *__restrict__ tasks, const uint32_t *__restrict__ sched, const uint32_t *__restrict__ sterms,
        const double *__restrict__ sweights, const uint32_t ntasks, uint32_t *__restrict__ ticket, uint32_t *__restrict__ counts, const uint32_t k,
        uint32_t *__restrict__ part_docs, double *__restrict__ part_scores, uint32_t *__restrict__ part_counts, const uint32_t *__restrict__ masked,
        const int sim, const uint32_t *__restrict__ planes, const uint32_t plw) {
        __shared__ PlanesShared sh;
        const uint32_t tid = threadIdx.x, lane = tid & 63u;
        const uint32_t wave = uni(tid >> 6);
        for (uint32_t i = tid; i < PLK_MAX_SPARSE * 2 * PL_STRIDE; i += PLK_WG)
                (&sh.pl[0][0])[i] = 0;
        PROF_DECL;
        PROF_START();
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
                const DevTask task = tasks[tix];
                const DevQuery q = plan[task.slot];
                {
                        const uint32_t wi = min(tid, (uint32_t)(sizeof(DevFused) / 4 - 1)); // (every lane stores: no divergent branch around the barriers)
                        ((uint32_t *)&sh.fq)[wi] = ((const uint32_t *)(fused + q.fused_idx))[wi];
                }
                __syncthreads();
                const DevFused &fq = sh.fq;
                const uint32_t nslots = min(uni(fq.nslots), (uint32_t)NS), nreq = uni(fq.nreq), negs = uni(fq.negslots);
                const uint32_t kk = min(lane, nslots - 1); // lane s (< nslots) of every wave looks after slot s, the lanes above mirror the last slot
                {
                        sh.term[kk] = terms[fq.term[kk]];
                        sh.hint_row[kk] = 0xffffffffu;
                        // what the slot's scorers add at frequency 1, and a bound of what they add at any frequency: BM25 float(w f / (f + 1.2)) < w;
                        // TF-IDF sqrt(f) w with f <= 65535; Trivial f
                        double t1 = 0.0, ubs = 0.0;
                        bool leaf = false;
                        for (uint32_t si = 0; si < q.nscore; ++si)
                                if (sterms[q.score_base + si] == fq.term[kk]) {
                                        const double wgt = sweights[q.score_base + si];
                                        t1 += (double)sim_score(sim, wgt, 1u);
                                        ubs += sim == TRI_SIM_TRIVIAL ? 65535.0 : sim == TRI_SIM_TFIDF ? (wgt > 0 ? 256.0 * wgt : 0.0) : (wgt > 0 ? wgt : 0.0);
                                        leaf = true;
                                }
                        sh.tab1[kk] = t1;
                        sh.w1[kk] = t1 > 0 ? t1 * (1.0 + 1e-9) : t1 * (1.0 - 1e-9);
                        sh.ub[kk] = fmax(ubs * (1.0 + 1e-6), t1 > 0 ? t1 * (1.0 + 1e-9) : 0.0);
                        const uint64_t lm = __builtin_amdgcn_ballot_w64(leaf && lane < nslots);
                        sh.leaf = (uint32_t)lm; // (same value from every lane)
                        sh.tk_n = 0;
                        sh.tk_full = 0;
                        sh.matches = 0;
                        sh.ess = (1u << nslots) - 1u; // no threshold yet: every slot's frequencies are wanted ...
                        sh.npat = 0xffffffffu;        // ... and every match is a candidate
                }
                __syncthreads();
                // ---- per task, uniform: which slots read term planes, where the others' LDS planes are, which slots score
                uint32_t dense_mask = 0;
                const uint32_t leaf_mask = uni(sh.leaf);
                uint32_t lidx[NS];
                size_t pbase[NS];
                {
                        uint32_t nl = 0;
#pragma unroll
                        for (uint32_t s = 0; s < NS; ++s) {
                                const uint32_t prow = s < nslots ? uni(fq.plane[s]) : PL_NONE;
                                pbase[s] = prow != PL_NONE ? (size_t)prow * 2 * plw : 0;
                                lidx[s] = 0;
                                if (s < nslots) {
                                        if (prow != PL_NONE)
                                                dense_mask |= 1u << s;
                                        else
                                                lidx[s] = nl++;
                                }
                        }
                }
                const uint32_t sparse_mask = ((1u << nslots) - 1u) & ~dense_mask;
                const uint32_t wfirst = task.tile_begin, wend = task.tile_end;
                // ---- wave 0, lane s: the directory position of decoded slot s (indexed lists: two cell-index entries per window; short lists:
                //      a cursor with its block's last docID)
                uint32_t cur = 0, cur_last = 0xffffffffu;
                const DevTerm myt = sh.term[kk];
                const uint32_t *mybl = blk_last + myt.first_block;
                const bool my_sparse = (sparse_mask >> kk) & 1u, my_indexed = myt.win_off != 0xffffffffu;
                if (wave == 0 && my_sparse && !my_indexed) {
                        uint32_t lo = 0, hi = myt.nblocks; // first block whose last document >= the task's first docID
                        const uint32_t key = wfirst * PL_W;
                        while (lo < hi) {
                                const uint32_t mid = (lo + hi) >> 1;
                                if (mybl[mid] < key)
                                        lo = mid + 1;
                                else
                                        hi = mid;
                        }
                        cur = lo;
                        cur_last = cur < myt.nblocks ? mybl[cur] : 0xffffffffu;
                }
                // rows of my slot that can hold documents of window w: [lo, hi] (first row whose last docID >= w0 ... first whose last >= the
                // next window's first docID); nothing when lo is beyond the list
                auto range_of = [&](const uint32_t w, uint32_t &lo, uint32_t &cnt) {
                        const uint32_t w0 = w * PL_W;
                        uint32_t hi;
                        if (!my_sparse) {
                                lo = 0;
                                cnt = 0;
                                return;
                        }
                        if (my_indexed) {
                                lo = win[myt.win_off + w * PL_CELLS];
                                hi = win[myt.win_off + (w + 1) * PL_CELLS];
                        } else {
                                while (cur < myt.nblocks && cur_last < w0) {
                                        ++cur;
                                        cur_last = cur < myt.nblocks ? mybl[cur] : 0xffffffffu;
                                }
                                lo = hi = cur;
                                if (cur_last < w0 + (PL_W - 1)) // (rare for a short list: further blocks end inside the window)
                                        while (hi + 1 < myt.nblocks && mybl[hi] < w0 + (PL_W - 1))
                                                ++hi;
                        }
                        hi = min(hi, myt.nblocks - 1);
                        cnt = lo < myt.nblocks ? hi - lo + 1 : 0xffffffffu; // (0xffffffff: the list is exhausted)
                };
                // wave 0 only, behind a barrier that follows the last set pass: the ranges of window w for everybody.  A row known (from the
                // hint it left when it was decoded) to continue past this window without a document in it adds nothing here — decided once,
                // by one wave, so that every wave works from the same ranges
                auto publish = [&](const uint32_t w, const uint32_t lo, uint32_t cnt) {
                        const bool dead = my_sparse && cnt == 0xffffffffu;
                        const uint64_t dm = __builtin_amdgcn_ballot_w64(dead);
                        if (dead)
                                cnt = 0;
                        if (cnt && sh.hint_row[kk] == lo && sh.hint_doc[kk] > w * PL_W + (PL_W - 1))
                                cnt = 0; // (then lo is the slot's only row here: a row that reaches past the window is the last one that touches it)
                        if (lane < nslots) {
                                sh.rng_lo[w & 1u][kk] = lo;
                                sh.rng_cnt[w & 1u][kk] = cnt;
                        }
                        sh.alive[w & 1u] = ~(uint32_t)dm; // (same value from every lane)
                };
                if (wave == 0) {
                        uint32_t lo, cnt;
                        range_of(wfirst, lo, cnt);
                        publish(wfirst, lo, cnt);
                }
                __syncthreads();
                uint32_t my_matches = 0;
                for (uint32_t w = wfirst; w < wend; ++w) {
                        const uint32_t par = w & 1u, w0 = w * PL_W;
                        // ---- this window's row ranges (left by wave 0 a window ago)
                        uint32_t s_lo[NS], s_cnt[NS], total = 0, rows_mask = 0;
                        const uint32_t alive = uni(sh.alive[par]);
#pragma unroll
                        for (uint32_t s = 0; s < NS; ++s) {
                                s_lo[s] = 0;
                                s_cnt[s] = 0;
                                if (s < nslots && ((sparse_mask >> s) & 1u)) {
                                        s_lo[s] = uni(sh.rng_lo[par][s]);
                                        s_cnt[s] = uni(sh.rng_cnt[par][s]);
                                        total += s_cnt[s];
                                        if (s_cnt[s])
                                                rows_mask |= 1u << s;
                                }
                        }
                        // a required group all of whose lists are exhausted ends the task; one that neither reads a term plane nor has a row in
                        // this window rules the window out
                        bool dead = false, possible = true;
                        for (uint32_t g = 0; g < nreq; ++g) {
                                const uint32_t gs = uni(fq.gslots[g]);
                                dead |= !(gs & (dense_mask | alive));
                                possible &= (gs & (dense_mask | rows_mask)) != 0;
                        }
                        if (dead)
                                break;
                        // wave 0 fetches the next window's directory entries now; they are published behind the set pass
                        uint32_t n_lo = 0, n_cnt = 0;
                        const bool more = w + 1 < wend;
                        if (wave == 0 && more)
                                range_of(w + 1, n_lo, n_cnt);
                        if (!possible) {
                                if (wave == 0 && more)
                                        publish(w + 1, n_lo, n_cnt);
                                __syncthreads();
                                continue;
                        }
                        // ---- the term planes' words of this thread's two window words travel while the other lists are decoded: a = plane A,
                        //      h = the level-2 plane (B: frequency not 1)
                        uint32_t a0[NS], a1[NS], h0[NS], h1[NS];
#pragma unroll
                        for (uint32_t s = 0; s < NS; ++s) {
                                a0[s] = a1[s] = h0[s] = h1[s] = 0;
                                if ((dense_mask >> s) & 1u) {
                                        const uint32_t *pa = planes + pbase[s] + (size_t)w * PL_WORDS;
                                        a0[s] = pa[tid];
                                        a1[s] = pa[tid + PLK_WG];
                                        if ((leaf_mask >> s) & 1u) {
                                                h0[s] = pa[plw + tid];
                                                h1[s] = pa[plw + tid + PLK_WG];
                                        }
                                }
                        }
                        const uint32_t ess = uni(sh.ess);
                        PROF_LAP(1);
                        // ---- set pass: the rows of the decoded slots form one work list, one lane per row; a slot no assignment of the filter
                        //      names is decoded for presence only
                        for (uint32_t v0 = 0; v0 < total; v0 += PLK_WG) {
                                const uint32_t v = v0 + tid;
                                uint32_t s = 0, r = v, lo = s_lo[0];
#pragma unroll
                                for (uint32_t k2 = 0; k2 + 1 < NS; ++k2) { // which slot's rows v falls into (ranges are wave-uniform)
                                        const bool nextslot = s == k2 && r >= s_cnt[k2];
                                        r = nextslot ? r - s_cnt[k2] : r;
                                        lo = nextslot ? s_lo[k2 + 1] : lo;
                                        s = nextslot ? k2 + 1 : s;
                                }
                                const bool act = v < total;
                                const bool needf = act && ((ess >> s) & 1u);
                                const bool anyf = __builtin_amdgcn_ballot_w64(needf) != 0ull; // (wave-uniform: one code path per wave)
                                if (act) {
                                        const DevTerm t = sh.term[s];
                                        const uint32_t b = lo + r;
                                        const uint32_t *bl = blk_last + t.first_block;
                                        const uint32_t prev = b ? bl[b - 1] : 0;
                                        const uint32_t last = bl[b];
                                        uint32_t ls = 0;
#pragma unroll
                                        for (uint32_t k2 = 0; k2 < NS; ++k2)
                                                ls = s == k2 ? lidx[k2] : ls;
                                        PlanePostH post{&sh.pl[ls][0]};
                                        if (CODEC == CODEC_LUCENE) {
                                                const uint4 rec = blk_rec[t.first_block + b];
                                                if (anyf)
                                                        row_decode<CODEC, true, PlanePostH>(index, t, b, rec.x, rec.y, rec.z, rec.w, TRI_BLOCK_N(t, b, index, 0), prev, last, w0, post PROF_PASS);
                                                else
                                                        row_decode<CODEC, false, PlanePostH>(index, t, b, rec.x, rec.y, rec.z, rec.w, TRI_BLOCK_N(t, b, index, 0), prev, last, w0, post PROF_PASS);
                                        } else {
                                                const uint32_t off = blk_off[t.first_block + b];
                                                const uint32_t dlen = blk_doff[t.first_block + b + 1] - blk_doff[t.first_block + b] - 1u;
                                                if (anyf)
                                                        row_decode<CODEC, true, PlanePostH>(index, t, b, off, dlen, 0, 0, TRI_BLOCK_N(t, b, index, off), prev, last, w0, post PROF_PASS);
                                                else
                                                        row_decode<CODEC, false, PlanePostH>(index, t, b, off, dlen, 0, 0, TRI_BLOCK_N(t, b, index, off), prev, last, w0, post PROF_PASS);
                                        }
                                        if (post.past < 0x80000000u) { // the row reaches past the window (it is the slot's last row here): leave the hint
                                                sh.hint_row[s] = b;
                                                sh.hint_doc[s] = w0 + PL_W + post.past;
                                        }
                                }
                        }
                        const uint32_t nof = sparse_mask & ~ess; // decoded slots whose B plane says nothing in this window: level 2 wherever present
                        PROF_LAP(2);
                        __syncthreads();
                        PROF_LAP(3);
                        if (wave == 0 && more) // (this window's hints are in; the next barrier — the sweep's — makes the ranges visible)
                                publish(w + 1, n_lo, n_cnt);
                        // ---- sweep: the decoded slots' words, the predicate, the candidates
#pragma unroll
                        for (uint32_t s = 0; s < NS; ++s)
                                if ((rows_mask >> s) & 1u) {
                                        const uint32_t *p = &sh.pl[lidx[s]][0];
                                        a0[s] = p[tid];
                                        a1[s] = p[tid + PLK_WG];
                                        if ((leaf_mask >> s) & 1u) {
                                                h0[s] = ((nof >> s) & 1u) ? a0[s] : p[PL_STRIDE + tid];
                                                h1[s] = ((nof >> s) & 1u) ? a1[s] : p[PL_STRIDE + tid + PLK_WG];
                                        }
                                }
                        uint32_t m0 = 0xffffffffu, m1 = 0xffffffffu;
                        for (uint32_t g = 0; g < nreq; ++g) {
                                const uint32_t gs = uni(fq.gslots[g]);
                                uint32_t x0 = 0, x1 = 0;
#pragma unroll
                                for (uint32_t s = 0; s < NS; ++s)
                                        if ((gs >> s) & 1u) {
                                                x0 |= a0[s];
                                                x1 |= a1[s];
                                        }
                                m0 &= x0;
                                m1 &= x1;
                        }
                        {
                                uint32_t n0 = 0, n1 = 0;
#pragma unroll
                                for (uint32_t s = 0; s < NS; ++s)
                                        if ((negs >> s) & 1u) {
                                                n0 |= a0[s];
                                                n1 |= a1[s];
                                        }
                                m0 &= ~n0;
                                m1 &= ~n1;
                        }
                        if (masked) { // masked_documents_registry::test (docidupdates.h:90-119)
                                m0 &= ~masked[(w0 >> 5) + tid];
                                m1 &= ~masked[(w0 >> 5) + tid + PLK_WG];
                        }
                        my_matches += (uint32_t)(__popc(m0) + __popc(m1));
                        // ---- the matches that can still enter the top-K: the candidate filter, word-wise (planes_filter)
                        auto candidates = [&](uint32_t &c0, uint32_t &c1) { // (ANDed into c0 / c1)
                                const uint32_t np = uni(sh.npat);
                                if (np == 0xffffffffu)
                                        return;
                                uint32_t y0 = 0, y1 = 0;
                                for (uint32_t i = 0; i < np; ++i) {
                                        const uint32_t ps = uni(sh.pat[i]);
                                        uint32_t x0 = 0xffffffffu, x1 = 0xffffffffu;
#pragma unroll
                                        for (uint32_t s = 0; s < NS; ++s) {
                                                if ((ps >> s) & 1u) {
                                                        x0 &= a0[s];
                                                        x1 &= a1[s];
                                                }
                                                if ((ps >> (8 + s)) & 1u) {
                                                        x0 &= h0[s];
                                                        x1 &= h1[s];
                                                }
                                        }
                                        y0 |= x0;
                                        y1 |= x1;
                                }
                                c0 &= y0;
                                c1 &= y1;
                        };
                        uint32_t c0 = m0, c1 = m1;
                        candidates(c0, c1);
                        // the decoded slots' words of this window (A and B) are cleared by their owner as soon as it has no candidate left in them
                        auto clear_mine = [&]() {
#pragma unroll
                                for (uint32_t s = 0; s < NS; ++s)
                                        if ((rows_mask >> s) & 1u) {
                                                uint32_t *p = &sh.pl[lidx[s]][0];
                                                p[tid] = 0;
                                                p[tid + PLK_WG] = 0;
                                                p[PL_STRIDE + tid] = 0;
                                                p[PL_STRIDE + tid + PLK_WG] = 0;
                                        }
                        };
                        bool cleared = false;
                        uint32_t qn = 0; // entries on this wave's lookup queue (wave-uniform)
                        PROF_LAP(4);
                        for (;;) {
                                // ---- every wave works its own candidates off, one per lane and step, no workgroup barrier: the levels from
                                //      registers give the known part of the score and a bound for the rest.  A candidate the bound does not rule
                                //      out and whose score is not fully known goes onto the wave's queue; the queue is worked off 64 at a time —
                                //      every lane fetching exact frequencies from the postings at once, not one lane while 63 wait
                                const bool full = uni(sh.tk_full) != 0;
                                const double thr_s = sh.thr_s;
                                const uint32_t thr_d = sh.thr_d;
                                auto offer = [&](const double sc, const uint32_t doc) { // false: no room (the buffer wants pruning)
                                        const uint32_t slot = atomicAdd(&sh.tk_n, 1u);
                                        if (slot >= PLK_CAP)
                                                return false;
                                        sh.tk_s[slot] = sc;
                                        sh.tk_d[slot] = doc;
                                        return true;
                                };
                                auto work_queue = [&]() { // the last (up to) 64 entries of the queue; entries that found no room go back
                                        const uint32_t take_n = min(qn, 64u), base = qn - take_n;
                                        qn = base;
                                        bool back = false;
                                        uint32_t ent = 0;
                                        if (lane < take_n) {
                                                ent = sh.wq[wave][base + lane];
                                                const uint32_t doc = w0 + (ent >> 16), held = ent & 0xffu, unk = (ent >> 8) & 0xffu;
                                                double sk = 0.0;
                                                for (uint32_t s = 0; s < nslots; ++s) {
                                                        if (!((held >> s) & 1u))
                                                                continue;
                                                        if (!((unk >> s) & 1u)) {
                                                                sk += sh.tab1[s];
                                                                continue;
                                                        }
                                                        const uint32_t f = planes_lookup_freq<CODEC>(index, blk_last, blk_off, blk_rec, blk_doff, win, sh.term[s], doc);
                                                        const uint32_t term = fq.term[s];
                                                        for (uint32_t si = 0; si < q.nscore; ++si)
                                                                if (sterms[q.score_base + si] == term)
                                                                        sk += (double)sim_score(sim, sweights[q.score_base + si], f);
                                                }
                                                if (!full || better(sk, doc, thr_s, thr_d))
                                                        back = !offer(sk, doc);
                                        }
                                        PROF_COUNT(17, lane == 0 ? take_n : 0);
                                        const uint64_t bm = __builtin_amdgcn_ballot_w64(back);
                                        if (back)
                                                sh.wq[wave][qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u))] = ent;
                                        qn += (uint32_t)__popcll(bm);
                                };
                                for (;;) {
                                        if (uni(__atomic_load_n(&sh.tk_n, __ATOMIC_RELAXED)) >= PLK_PRUNE_AT)
                                                break; // the buffer wants pruning first (candidates and queue stay where they are)
                                        if (qn >= 64) { // (a step below may add 64 entries: the queue is kept below 64 before it)
                                                work_queue();
                                                continue;
                                        }
                                        const bool has = (c0 | c1) != 0;
                                        if (__builtin_amdgcn_ballot_w64(has) == 0ull) {
                                                if (!qn)
                                                        break;
                                                work_queue();
                                                continue;
                                        }
                                        bool enq = false;
                                        uint32_t ent = 0;
                                        if (has) {
                                                const uint32_t which = c0 ? 0u : 1u;
                                                const uint32_t bit = (uint32_t)__builtin_ctz(which ? c1 : c0);
                                                const uint32_t rel = 32u * (tid + which * PLK_WG) + bit, doc = w0 + rel;
                                                double sk = 0.0, sb = 0.0; // the known part of the score; bounds of the slots whose frequency is not known yet
                                                uint32_t unk = 0, held = 0;
#pragma unroll
                                                for (uint32_t s = 0; s < NS; ++s) {
                                                        const uint32_t av = which ? a1[s] : a0[s], hv = which ? h1[s] : h0[s];
                                                        if (!((leaf_mask >> s) & 1u) || !((av >> bit) & 1u))
                                                                continue;
                                                        held |= 1u << s;
                                                        if (!((hv >> bit) & 1u))
                                                                sk += sh.tab1[s];
                                                        else {
                                                                unk |= 1u << s;
                                                                sb += sh.ub[s];
                                                        }
                                                }
                                                bool done = true;
                                                if (!full || better(sk + sb, doc, thr_s, thr_d)) {
                                                        if (unk) {
                                                                enq = true;
                                                                ent = rel << 16 | unk << 8 | held;
                                                        } else
                                                                done = offer(sk, doc); // (no room: the candidate stays for after the prune)
                                                }
                                                if (done) {
                                                        if (which)
                                                                c1 &= c1 - 1u;
                                                        else
                                                                c0 &= c0 - 1u;
                                                }
                                        }
                                        PROF_COUNT(16, lane == 0 ? __popcll(__builtin_amdgcn_ballot_w64(has)) : 0);
                                        const uint64_t em = __builtin_amdgcn_ballot_w64(enq);
                                        if (enq)
                                                sh.wq[wave][qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u))] = ent;
                                        qn += (uint32_t)__popcll(em);
                                }
                                const bool pending = (c0 | c1) != 0;
                                if (!pending && !cleared) {
                                        clear_mine();
                                        cleared = true;
                                }
                                sh.flag[wave] = (__builtin_amdgcn_ballot_w64(pending) != 0ull || qn) ? 1u : 0u; // (wave-uniform value, every lane stores it)
                                __syncthreads();
                                uint32_t anyp = 0;
#pragma unroll
                                for (uint32_t wv = 0; wv < PLK_WG / 64; ++wv)
                                        anyp |= sh.flag[wv];
                                anyp = uni(anyp);
                                const uint32_t n = min(uni(sh.tk_n), PLK_CAP);
                                if (n >= PLK_PRUNE_AT) {
                                        __syncthreads(); // (every lane has read the flags and tk_n)
                                        planes_prune(sh, n, k);
                                        planes_filter(sh, nslots);
                                        PROF_COUNT(18, tid == 0 ? 1 : 0);
                                        if (anyp)
                                                candidates(c0, c1); // (the threshold moved: what is left is filtered again)
                                } else if (anyp)
                                        __syncthreads(); // (cannot happen — a wave only stops early at a full buffer —; kept so that a flag is never rewritten while read)
                                if (!anyp)
                                        break;
                        }
                        PROF_COUNT(19, tid == 0 ? 1 : 0);
                        PROF_LAP(5);
                }
                // ---- the task's result: its best k (ranked) and its match count
                __syncthreads();
                planes_prune(sh, min(uni(sh.tk_n), PLK_CAP), k);
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
                __syncthreads();
                PROF_LAP(6);
        }
        PROF_FLUSH();
}
