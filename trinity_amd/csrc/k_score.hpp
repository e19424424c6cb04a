// k_score.hpp — BM25 scoring over match segments, per-task and per-query top-K, docset hashing
// Part of libtrinity_hip.so (MI355X / gfx950); included by trinity_hip.hip.  New code, no reference source.
#pragma once
#include "k_match.hpp"

// ------------------------------------------------------------------------------------------ k_score / k_topk_merge
// AccumulatedScoreScheme (exec.h:36-41).  k_and has produced every query's ascending match list; k_score walks each
// task's segment in tiles of 4096 matches, and for every scoring term looks the matches up again through the
// directory (each match binary-searches its block, the first match of a block decodes it once: deltas to locate the
// matching positions, then the freqs), adding  float(idf * float(f) / double(f + 1.2f))  to a per-match double in LDS —
// IndexSourcesCollectionBM25Scorer::Scorer::score (similarity.h:228-235) summed in iterator order by the Conjuction
// wrapper (docset_iterators_scorers.cpp:173-193).  The tile is then offered to the task's top-K (score descending,
// docID ascending: the application-side MatchedIndexDocumentsFilter heap, matches.h:155-171).  k_topk_merge folds
// the tasks' partial lists into one list per query.
#ifndef TRI_SCORE_TILE
#define TRI_SCORE_TILE 4096 // (cfg1, ms: 4096 -> 0.86, 2048 -> 1.00, 1024 -> 1.45: every tile repeats the per-term searches)
#endif
constexpr uint32_t SCORE_TILE = TRI_SCORE_TILE;
// (TOPK_MAX: dev_structs.hpp)
constexpr uint32_t TOPK_CAP = TOPK_MAX + AND_WG; // survivors + one wave of newcomers

struct TopK {
        double s[TOPK_CAP];
        uint32_t d[TOPK_CAP];
        uint32_t n;       // entries held (uniform)
        uint32_t full;    // n has reached k at least once => thr_* valid
        double thr_s;     // the k-th best entry
        uint32_t thr_d;
};

__device__ __forceinline__ bool better(const double s1, const uint32_t d1, const double s2, const uint32_t d2) {
        return s1 > s2 || (s1 == s2 && d1 < d2);
}

// Keep the best k of the n entries, sorted best-first (rank by counting: the order is strict, ranks are unique).
__device__ void topk_prune(TopK &tk, const uint32_t k, uint32_t *scan) {
        const uint32_t tid = threadIdx.x;
        const uint32_t n = uni(tk.n);
        double es[2];
        uint32_t ed[2], rk[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
                const uint32_t i = tid + r * AND_WG;
                rk[r] = 0xffffffffu;
                if (i < n) {
                        es[r] = tk.s[i];
                        ed[r] = tk.d[i];
                        uint32_t c = 0;
                        for (uint32_t j = 0; j < n; ++j)
                                c += better(tk.s[j], tk.d[j], es[r], ed[r]) ? 1u : 0u;
                        rk[r] = c;
                }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; ++r)
                if (rk[r] < k) {
                        tk.s[rk[r]] = es[r];
                        tk.d[rk[r]] = ed[r];
                }
        __syncthreads();
        const uint32_t m = n < k ? n : k;
        // uniform stores by every lane (no single-lane branch around the barrier loop that calls us)
        tk.n = m;
        if (m == k) {
                tk.full = 1;
                tk.thr_s = tk.s[k - 1];
                tk.thr_d = tk.d[k - 1];
        }
        (void)scan;
        __syncthreads();
}

// Every lane offers at most one (score, doc); survivors of the threshold are appended, pruning when the buffer fills.
__device__ void topk_offer(TopK &tk, const uint32_t k, const bool valid, const double sc, const uint32_t doc, uint32_t *scan) {
        const uint32_t tid = threadIdx.x;
        const uint32_t n0 = uni(tk.n); // stable: the previous call ended with a barrier
        const bool take = valid && (!tk.full || better(sc, doc, tk.thr_s, tk.thr_d));
        const uint64_t m = __ballot(take);
        const uint32_t lane = tid & 63;
        const uint32_t before = __popcll(m & ((1ull << lane) - 1ull));
        scan[tid >> 6] = __popcll(m);
        __syncthreads();
        uint32_t base = n0, tot = 0;
        for (uint32_t w = 0; w < AND_WG / 64; ++w) {
                if (w < (tid >> 6))
                        base += scan[w];
                tot += scan[w];
        }
        tot = uni(tot);
        if (take) {
                tk.s[base + before] = sc;
                tk.d[base + before] = doc;
        }
        __syncthreads();
        tk.n = n0 + tot; // same value from every lane
        __syncthreads();
        if (n0 + tot > TOPK_MAX)
                topk_prune(tk, k, scan);
}

// The frequency of `doc` in term t (a document of the term): the block through the docID-cell index (or a bisection of the directory), the
// deltas to the document's slot, the freqs up to it.  What a scorer whose term has a PLANE needs only for the documents planes B and C mark
// "neither 1 nor 2" (k_planes.hpp).
template <int CODEC>
__device__ __noinline__ uint32_t score_lookup_freq(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last, const uint32_t *__restrict__ blk_off,
                                                   const uint32_t *__restrict__ win, const DevTerm &t, const uint32_t doc) {
        const uint32_t *bl = blk_last + t.first_block;
        uint32_t lo = 0, hi = t.nblocks;
        if (t.win_off != 0xffffffffu) {
                lo = win[t.win_off + (doc >> CELL_LOG2)];
                hi = min(win[t.win_off + (doc >> CELL_LOG2) + 1] + 1u, t.nblocks);
        }
        while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (bl[mid] < doc)
                        lo = mid + 1;
                else
                        hi = mid;
        }
        const uint32_t b = lo;
        const uint32_t off = blk_off[t.first_block + b];
        const uint32_t n = TRI_BLOCK_N(t, b, index, off);
        uint32_t d = b ? bl[b - 1] : 0, pos = n - 1;
        DeltaStream<CODEC> ds;
        ds.init(index, t, b, off);
        for (uint32_t i = 0; i + 1 < n; ++i) { // (GOOGLE: the freqs start where the deltas end, so all of them are walked)
                d += ds.next();
                if (d == doc && pos == n - 1)
                        pos = i;
        }
        FreqStream<CODEC> fs;
        fs.init(index, t, b, off, ds);
        uint32_t f = 0;
        for (uint32_t i = 0; i <= pos; ++i)
                f = fs.next();
        return f;
}

struct ScoreShared {
        uint32_t cand[SCORE_TILE];
        double score[SCORE_TILE];
        uint16_t mptr[32][AND_WG]; // per lane (column): the tile indices of the matches its block coincides with
        uint32_t blkof[AND_WG + 1];
        uint32_t scan[8];
        uint32_t bcast[4];
        TopK tk;
};


// ---- the order k_score runs its tasks in: heaviest first, by their MATCH counts — which only the device knows, once the matching kernels are through.  In the schedule's
//      order (k_and's: by the plane row a task probes) cfg3's 6 209 tasks — half of them without a match, a fifth of them 77 % of the work — kept 68 % of the
//      workgroups busy: 200 us tasks started 250 us into a 463 us span, the last tenth of it ran on 14 workgroups of 512 (per-task stamps, -DTRI_TASKTIMES=2).  One
//      workgroup, a counting sort by the count's octave, descending (any order within an octave).
constexpr int SORD_WG = 1024;
__global__ __launch_bounds__(SORD_WG) void k_score_order(const uint32_t *__restrict__ sched, const uint32_t *__restrict__ counts, const uint32_t n, uint32_t *__restrict__ order) {
        __shared__ uint32_t hist[33], cur[33];
        const uint32_t tid = threadIdx.x;
        if (tid < 33)
                hist[tid] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += SORD_WG) {
                const uint32_t m = counts[sched[i]];
                atomicAdd(&hist[m ? 32u - (uint32_t)__clz(m) : 0u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
                uint32_t at = 0;
                for (uint32_t b = 33; b-- > 0;) {
                        cur[b] = at;
                        at += hist[b];
                }
        }
        __syncthreads();
        for (uint32_t i = tid; i < n; i += SORD_WG) {
                const uint32_t tix = sched[i], m = counts[tix];
                order[atomicAdd(&cur[m ? 32u - (uint32_t)__clz(m) : 0u], 1u)] = tix;
        }
}

constexpr uint32_t SCORE_WGS_PER_CU = (160u * 1024u / sizeof(ScoreShared)) < 6u ? (160u * 1024u / sizeof(ScoreShared)) : 6u; // LDS; ~80-110 registers

template <int CODEC>
__global__ __launch_bounds__(AND_WG) void k_score(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last,
                                                  const uint32_t *__restrict__ blk_off, const DevTerm *__restrict__ terms,
                                                  const DevQuery *__restrict__ plan, const DevTask *__restrict__ tasks,
                                                  const uint32_t *__restrict__ sched, const uint32_t *__restrict__ sterms,
                                                  const double *__restrict__ sweights, const uint32_t ntasks, uint32_t *__restrict__ ticket,
                                                  const uint32_t *__restrict__ out, const uint32_t *__restrict__ counts, const uint32_t k,
                                                  uint32_t *__restrict__ part_docs, double *__restrict__ part_scores,
                                                  uint32_t *__restrict__ part_counts, double *__restrict__ all_scores,
                                                  const double *__restrict__ pscore, const int sim, const uint32_t *__restrict__ win,
                                                  const uint32_t *__restrict__ splane, const uint32_t *__restrict__ planes, const uint32_t plw) {
        __shared__ ScoreShared sh;
        const uint32_t tid = threadIdx.x;
        const uint32_t wave = uni(tid >> 6);
        PROF_DECL;
        PROF_START();
        for (;;) {
                if (wave == 0) {
                        const uint32_t old = atomicAdd(ticket, 1u);
                        sh.bcast[0] = uni(old) >> 6;
                }
                __syncthreads();
                const uint32_t ticket_no = uni(sh.bcast[0]);
                __syncthreads();
                if (ticket_no >= ntasks)
                        break;
                const uint32_t tix = sched[ticket_no];
                TASKTIME_SCORE(8 * ticket_no);
                const DevTask task = tasks[tix];
                const DevQuery q = plan[task.slot];
                const uint32_t M = counts[tix];
                const uint32_t *seg = out + task.out_off;
                sh.tk.n = 0;
                sh.tk.full = 0;
                __syncthreads();
                for (uint32_t tb = 0; tb < M; tb += SCORE_TILE) {
                        const uint32_t C = min(SCORE_TILE, M - tb);
                        for (uint32_t j = tid; j < C; j += AND_WG) {
                                sh.cand[j] = seg[tb + j];
                                // a Phrase iterator scores as one unit; k_phrase left its contribution beside the match
                                sh.score[j] = (q.nphrases && pscore) ? pscore[task.out_off + tb + j] : 0.0;
                        }
                        __syncthreads();
                        PROF_LAP(0);
                        for (uint32_t ti = 0; ti < q.nscore; ++ti) {
                                const DevTerm t = terms[sterms[q.score_base + ti]];
                                const double w = sweights[q.score_base + ti];
                                const uint32_t *bl = blk_last + t.first_block;
                                const uint32_t *bo = blk_off + t.first_block;
                                const uint32_t prow = splane ? uni(splane[q.score_base + ti]) : PL_NONE;
                                if (prow != PL_NONE) {
                                        // the term has a plane (k_term_planes decoded its list once, for every batch of the index): whether a match holds
                                        // the term is bit A, its frequency there 1 unless bit B says otherwise, 2 unless bit C does — only a match both
                                        // mark goes to the postings.  (A head term's blocks in a tile's docID range outnumber the tile's matches: without
                                        // the planes every match decoded a block of every scorer — cfg3: 900 us for a tile of 8192 matches and four head terms)
                                        // (round 5: the row's interleaved LEVEL words tell the frequency itself up to PL_NESTED - 1 — one 12-byte probe —; the postings are
                                        //  walked only for the top level: a frequency of 0 or of PL_NESTED and more)
                                        const uint32_t *lvw = planes + (size_t)prow * PL_HI * plw + (size_t)PL_HI_LEVELS * plw; // (`planes`: the rows' HIGH parts)
                                        for (uint32_t j = tid; j < C; j += AND_WG) {
                                                const uint32_t doc = sh.cand[j], wi = doc >> 5, bit = doc & 31u;
                                                const uint32_t *lv = lvw + 3u * wi;
                                                uint32_t f = ((lv[0] >> bit) & 1u) | (((lv[1] >> bit) & 1u) << 1) | (((lv[2] >> bit) & 1u) << 2);
                                                if (!f)
                                                        continue;
                                                if (f == PL_NESTED)
                                                        f = score_lookup_freq<CODEC>(index, blk_last, blk_off, win, t, doc);
                                                sh.score[j] += (double)sim_score(sim, w, f);
                                        }
                                        __syncthreads();
                                        continue;
                                }
                                // one block of the term against the matches from index j on (cv = match j, known to be <= the
                                // block's last document).  Deltas: every document gallops forward through the tile's matches
                                // (exponential probe + bisection in LDS — under an OR a sparse term's neighbours are thousands of
                                // matches apart, a linear walk would visit them all) and remembers the index of the match it
                                // coincides with; freqs: the i-th marked document scores the i-th remembered match.  A match is a
                                // document of at most one block, so no two lanes touch the same score.
                                auto score_block = [&](const uint32_t bj, const uint32_t j, uint32_t cv) {
                                        const uint32_t prev = bj ? bl[bj - 1] : 0;
                                        const uint32_t last = bl[bj];
                                        const uint32_t off = bo[bj];
                                        const uint32_t n = TRI_BLOCK_N(t, bj, index, off);
                                        DeltaStream<CODEC> s;
                                        s.init(index, t, bj, off);
                                        uint32_t doc = prev, ptr = j, mask = 0, nm = 0;
                                        for (uint32_t i = 0; i < n; ++i) {
                                                doc = (i + 1 < n) ? doc + s.next() : last;
                                                if (cv < doc) {
                                                        uint32_t step = 1, lo = ptr + 1; // first index in (ptr, C] whose match >= doc
                                                        while (lo + step <= C && sh.cand[lo + step - 1] < doc) {
                                                                lo += step;
                                                                step <<= 1;
                                                        }
                                                        uint32_t hi = min(lo + step - 1, C);
                                                        while (lo < hi) {
                                                                const uint32_t mid = (lo + hi) >> 1;
                                                                if (sh.cand[mid] < doc)
                                                                        lo = mid + 1;
                                                                else
                                                                        hi = mid;
                                                        }
                                                        ptr = lo;
                                                        cv = ptr < C ? sh.cand[ptr] : 0xffffffffu;
                                                }
                                                if (cv == doc) {
                                                        mask |= 1u << i;
                                                        sh.mptr[nm++][tid] = (uint16_t)ptr;
                                                }
                                        }
                                        PROF_LAP(9);
                                        // freqs follow the n-1 deltas
                                        FreqStream<CODEC> fs;
                                        fs.init(index, t, bj, off, s);
                                        nm = 0;
                                        for (uint32_t i = 0; i < n; ++i) {
                                                const uint32_t f = fs.next();
                                                if ((mask >> i) & 1u)
                                                        sh.score[sh.mptr[nm++][tid]] += (double)sim_score(sim, w, f);
                                        }
                                        PROF_LAP(10);
                                };
                                // the term's blocks that can hold a match of this tile
                                const uint32_t cmin = sh.cand[0], cmax = sh.cand[C - 1];
                                const uint32_t b0 = wg_lower_bound<AND_WG>(sh.scan, bl, t.nblocks, cmin);
                                uint32_t b1 = b0;
                                if (b0 < t.nblocks) {
                                        b1 = b0 + wg_lower_bound<AND_WG>(sh.scan, bl + b0, t.nblocks - b0, cmax);
                                        if (b1 >= t.nblocks)
                                                b1 = t.nblocks - 1;
                                }
                                __syncthreads(); // searches done
                                PROF_LAP(1);
                                if (b0 < t.nblocks && b1 - b0 + 1 <= C) {
                                        // block-driven (dense results — unions of head terms, window-sized conjunctions): one lane per
                                        // block of the term in range, a binary search of the tile's matches in LDS for the first one
                                        // the block can hold.  (Match-driven, 256 consecutive matches of a dense result map to a
                                        // handful of blocks and leave most lanes idle.)
                                        for (uint32_t b = b0 + tid; b <= b1; b += AND_WG) {
                                                const uint32_t prev = b ? bl[b - 1] : 0;
                                                uint32_t lo = 0, hi = C;
                                                while (lo < hi) {
                                                        const uint32_t mid = (lo + hi) >> 1;
                                                        if (sh.cand[mid] <= prev)
                                                                lo = mid + 1;
                                                        else
                                                                hi = mid;
                                                }
                                                PROF_LAP(8);
                                                if (lo < C && sh.cand[lo] <= bl[b])
                                                        score_block(b, lo, sh.cand[lo]);
                                        }
                                        PROF_LAP(2);
                                        __syncthreads();
                                        PROF_LAP(3);
                                } else if (b0 < t.nblocks) {
                                        // match-driven: every match finds its block inside [b0, b1]; the first match of a block run
                                        // decodes the block and merges forward
                                        sh.blkof[0] = 0xffffffffu;
                                        __syncthreads();
                                        for (uint32_t base = 0; base < C; base += AND_WG) {
                                                const uint32_t j = base + tid;
                                                uint32_t bj = 0xffffffffu, cv = 0;
                                                if (j < C) {
                                                        cv = sh.cand[j];
                                                        uint32_t lo = b0, hi = b1 + 1;
                                                        while (lo < hi) {
                                                                const uint32_t mid = (lo + hi) >> 1;
                                                                if (bl[mid] < cv)
                                                                        lo = mid + 1;
                                                                else
                                                                        hi = mid;
                                                        }
                                                        bj = lo;
                                                }
                                                sh.blkof[tid + 1] = bj;
                                                __syncthreads();
                                                const uint32_t prevb = sh.blkof[tid];
                                                __syncthreads();
                                                {
                                                        const uint32_t lastb = __shfl(bj, 63, 64);
                                                        const bool lastwave = (tid >> 6) == (AND_WG / 64 - 1);
                                                        sh.blkof[lastwave ? 0 : tid + 1] = lastwave ? lastb : bj;
                                                }
                                                if (j < C && bj < t.nblocks && bj != prevb)
                                                        score_block(bj, j, cv);
                                                __syncthreads();
                                        }
                                        PROF_LAP(4);
                                }
                        }
                        PROF_LAP(5);
                        if (all_scores) // full score stream: what consider(id, score) receives for every match
                                for (uint32_t j = tid; j < C; j += AND_WG)
                                        all_scores[task.out_off + tb + j] = sh.score[j];
                        if (k) {
                                // offer the tile to the task's top-K
                                for (uint32_t base = 0; base < C; base += AND_WG) {
                                        const uint32_t j = base + tid;
                                        topk_offer(sh.tk, k, j < C, j < C ? sh.score[j] : 0.0, j < C ? sh.cand[j] : 0u, sh.scan);
                                }
                        }
                        __syncthreads();
                        PROF_LAP(6);
                }
                if (k) {
                        topk_prune(sh.tk, k, sh.scan);
                        const uint32_t n = uni(sh.tk.n);
                        for (uint32_t i = tid; i < n; i += AND_WG) {
                                part_docs[(uint64_t)tix * k + i] = sh.tk.d[i];
                                part_scores[(uint64_t)tix * k + i] = sh.tk.s[i];
                        }
                        if (wave == 0)
                                part_counts[tix] = n;
                }
                TASKTIME_SCORE(8 * ticket_no + 1);
                __syncthreads();
                PROF_LAP(7);
        }
        PROF_FLUSH();
}

// one workgroup per query: stream the tasks' partial lists through the same top-K structure
__global__ __launch_bounds__(AND_WG) void k_topk_merge(const DevQuery *__restrict__ plan, const uint32_t nq, const uint32_t k,
                                                       const uint32_t *__restrict__ part_docs, const double *__restrict__ part_scores,
                                                       const uint32_t *__restrict__ part_counts, uint32_t *__restrict__ top_docs,
                                                       float *__restrict__ top_scores, uint32_t *__restrict__ top_counts) {
        __shared__ TopK tk;
        __shared__ uint32_t scan[8];
        const uint32_t tid = threadIdx.x;
        for (uint32_t slot = blockIdx.x; slot < nq; slot += gridDim.x) {
                const DevQuery q = plan[slot];
                if (q.qid == 0xffffffffu) // (a hidden phrase query of a TASK_TREE query: no caller query, no row)
                        continue;
                tk.n = 0;
                tk.full = 0;
                __syncthreads();
                for (uint32_t t = 0; t < q.ntasks; ++t) {
                        const uint32_t tix = q.first_task + t;
                        const uint32_t c = part_counts[tix];
                        for (uint32_t base = 0; base < c; base += AND_WG) {
                                const uint32_t i = base + tid;
                                const bool v = i < c;
                                topk_offer(tk, k, v, v ? part_scores[(uint64_t)tix * k + i] : 0.0, v ? part_docs[(uint64_t)tix * k + i] : 0u, scan);
                        }
                }
                topk_prune(tk, k, scan);
                const uint32_t n = uni(tk.n);
                for (uint32_t i = tid; i < k; i += AND_WG) {
                        top_docs[(uint64_t)q.qid * k + i] = i < n ? tk.d[i] : 0u;
                        top_scores[(uint64_t)q.qid * k + i] = i < n ? (float)tk.s[i] : 0.0f;
                }
                if (uni(tid >> 6) == 0)
                        top_counts[q.qid] = n;
                __syncthreads();
        }
}

// per-query match counts on the device (the multi-GPU result gather reads them without a host bounce): one lane per plan slot
__global__ void k_query_counts(const DevQuery *__restrict__ plan, const uint32_t *__restrict__ counts_by_task, const uint32_t nq,
                               uint64_t *__restrict__ qcounts) {
        const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
        if (s >= nq)
                return;
        const DevQuery q = plan[s];
        if (q.qid == 0xffffffffu) // (a hidden phrase query of a TASK_TREE query)
                return;
        uint64_t c = 0;
        for (uint32_t t = 0; t < q.ntasks; ++t)
                c += counts_by_task[q.first_task + t];
        qcounts[q.qid] = c;
}

// ---- collections of segments (IndexSourcesCollection, index_source.cpp:3-30): the same queries ran over every source; one
// workgroup per query streams the sources' partial top-K lists (doubles, task by task) through the same top-K structure — what
// the application's heap sees when exec_query hands it consider(id, score) source after source — and adds up the match counts.
struct DevSource {
        const DevQuery *plan;
        const uint32_t *slot_of_query; // caller query -> plan slot of this source's batch (0xffffffff: matches nothing there)
        const uint32_t *part_docs;
        const double *part_scores;
        const uint32_t *part_counts;
        const uint64_t *qcounts;
};
__global__ __launch_bounds__(AND_WG) void k_topk_merge_sources(const DevSource *__restrict__ src, const uint32_t nsrc, const uint32_t nq, const uint32_t k,
                                                               uint32_t *__restrict__ top_docs, float *__restrict__ top_scores,
                                                               uint32_t *__restrict__ top_counts, uint64_t *__restrict__ counts) {
        __shared__ TopK tk;
        __shared__ uint32_t scan[8];
        const uint32_t tid = threadIdx.x;
        for (uint32_t qi = blockIdx.x; qi < nq; qi += gridDim.x) {
                tk.n = 0;
                tk.full = 0;
                __syncthreads();
                uint64_t total = 0;
                for (uint32_t s = 0; s < nsrc; ++s) {
                        const DevSource S = src[s];
                        total += S.qcounts[qi];
                        const uint32_t slot = uni(S.slot_of_query[qi]);
                        if (slot == 0xffffffffu || !k) // (uniform)
                                continue;
                        const DevQuery q = S.plan[slot];
                        const uint32_t ntasks = uni(q.ntasks);
                        for (uint32_t t = 0; t < ntasks; ++t) {
                                const uint32_t tix = uni(q.first_task) + t;
                                const uint32_t c = uni(S.part_counts[tix]);
                                for (uint32_t base = 0; base < c; base += AND_WG) {
                                        const uint32_t i = base + tid;
                                        const bool v = i < c;
                                        topk_offer(tk, k, v, v ? S.part_scores[(uint64_t)tix * k + i] : 0.0, v ? S.part_docs[(uint64_t)tix * k + i] : 0u, scan);
                                }
                        }
                }
                if (k) {
                        topk_prune(tk, k, scan);
                        const uint32_t n = uni(tk.n);
                        for (uint32_t i = tid; i < k; i += AND_WG) {
                                top_docs[(uint64_t)qi * k + i] = i < n ? tk.d[i] : 0u;
                                top_scores[(uint64_t)qi * k + i] = i < n ? (float)tk.s[i] : 0.0f;
                        }
                        if (uni(tid >> 6) == 0) // (a wave-uniform branch: every lane of wave 0 stores the same word)
                                top_counts[qi] = n;
                }
                if (uni(tid >> 6) == 0)
                        counts[qi] = total;
                __syncthreads();
        }
}

// FNV-1a(64) of each query's docID set (little-endian bytes), one lane per query — verification helper
__global__ void k_hash_docsets(const DevQuery *__restrict__ plan, const DevTask *__restrict__ tasks_by_query,
                               const uint32_t *__restrict__ counts_by_query, const uint32_t nq, const uint32_t *__restrict__ out,
                               uint64_t *__restrict__ hashes) {
        const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
        if (s >= nq)
                return;
        const DevQuery q = plan[s];
        uint64_t h = 1469598103934665603ull;
        if (q.form == RESULT_BITMAP) { // one bit per document: the tasks' windows in order, a word's bits ascending
                for (uint32_t t = 0; t < q.ntasks; ++t) {
                        const DevTask tk = tasks_by_query[q.first_task + t];
                        const uint32_t *p = out + tk.out_off;
                        const uint32_t nw = (tk.tile_end - tk.tile_begin) * SPAN_WORDS;
                        for (uint32_t i = 0; i < nw; ++i)
                                for (uint32_t m = p[i]; m; m &= m - 1u) {
                                        uint32_t d = (tk.tile_begin * SPAN_WORDS + i) * 32u + (uint32_t)__builtin_ctz(m);
                                        for (int b = 0; b < 4; ++b) {
                                                h = (h ^ (d & 0xffu)) * 1099511628211ull;
                                                d >>= 8;
                                        }
                                }
                }
                hashes[s] = h;
                return;
        }
        for (uint32_t t = 0; t < q.ntasks; ++t) {
                const uint32_t *p = out + tasks_by_query[q.first_task + t].out_off;
                const uint32_t n = counts_by_query[q.first_task + t];
                for (uint32_t i = 0; i < n; ++i) {
                        uint32_t d = p[i];
                        for (int b = 0; b < 4; ++b) {
                                h = (h ^ (d & 0xffu)) * 1099511628211ull;
                                d >>= 8;
                        }
                }
        }
        hashes[s] = h;
}

// Every query's docID set into ONE contiguous buffer (tri_batch_docsets): a workgroup per task copies the task's segment — or, for a query
// whose result is a bitmap (RESULT_BITMAP), writes out the documents of its windows' words — to flat[slot_off[query] + the matches of the
// query's earlier tasks ...).  The in-order concatenation of a query's task segments IS its ascending docID set.
// `mixed` (tri_batch_docsets_mixed): a RESULT_BITMAP query's words go out AS THEY ARE — flat[slot_off[query] + word of the query's region] — instead of as docIDs.
__global__ __launch_bounds__(256) void k_deliver_docsets(const DevQuery *__restrict__ plan, const DevTask *__restrict__ tasks, const uint32_t *__restrict__ counts,
                                                         const uint32_t *__restrict__ out, const uint64_t *__restrict__ slot_off, uint32_t *__restrict__ flat, const uint32_t mixed) {
        __shared__ uint32_t scan[256];
        __shared__ uint64_t base_sh;
        const uint32_t tix = blockIdx.x, tid = threadIdx.x;
        const DevTask tk = tasks[tix];
        if (task_onepass(tk.kind) && tk.kind != TASK_FUSED_GEN)
                return; // (a one-pass scored task keeps no docID set; uniform)
        const DevQuery q = plan[tk.slot];
        if (q.qid == 0xffffffffu)
                return; // (a hidden phrase query — a leaf of a TASK_TREE query: no caller query, no place in flat[]; uniform)
        const uint32_t c = counts[tix];
        if (mixed && q.form == RESULT_BITMAP) { // the task's windows, word for word (16 bytes a lane and step: regions and windows are multiples of SPAN_WORDS)
                const uint4 *src = (const uint4 *)(out + tk.out_off);
                uint4 *dst4 = (uint4 *)(flat + slot_off[tk.slot] + (tk.out_off - q.out_off));
                const uint32_t n4 = (tk.tile_end - tk.tile_begin) * (SPAN_WORDS / 4);
                for (uint32_t i = threadIdx.x; i < n4; i += 256)
                        dst4[i] = src[i];
                return;
        }
        // the matches of the query's earlier tasks
        uint64_t before = 0;
        for (uint32_t t = q.first_task + tid; t < tix; t += 256)
                before += counts[t];
        scan[tid] = (uint32_t)before; // (a query's matches fit 32 bits)
        __syncthreads();
        for (uint32_t d = 128; d; d >>= 1) {
                if (tid < d)
                        scan[tid] += scan[tid + d];
                __syncthreads();
        }
        if (tid == 0)
                base_sh = slot_off[tk.slot] + scan[0];
        __syncthreads();
        uint32_t *dst = flat + base_sh;
        if (q.form != RESULT_BITMAP) {
                const uint32_t *src = out + tk.out_off;
                for (uint32_t i = tid; i < c; i += 256)
                        dst[i] = src[i];
                return;
        }
        const uint32_t *p = out + tk.out_off;
        const uint32_t nw = (tk.tile_end - tk.tile_begin) * SPAN_WORDS;
        uint32_t done = 0; // documents written so far (uniform)
        for (uint32_t w0 = 0; w0 < nw; w0 += 256) {
                __syncthreads();
                uint32_t m = w0 + tid < nw ? p[w0 + tid] : 0u;
                const uint32_t pc = (uint32_t)__popc(m);
                scan[tid] = pc;
                __syncthreads();
                for (uint32_t d = 1; d < 256; d <<= 1) { // inclusive scan
                        const uint32_t v = tid >= d ? scan[tid - d] : 0u;
                        __syncthreads();
                        scan[tid] += v;
                        __syncthreads();
                }
                uint32_t at = done + scan[tid] - pc;
                const uint32_t doc0 = (tk.tile_begin * SPAN_WORDS + w0 + tid) * 32u;
                for (; m; m &= m - 1u)
                        dst[at++] = doc0 + (uint32_t)__builtin_ctz(m);
                done += scan[255];
        }
}
