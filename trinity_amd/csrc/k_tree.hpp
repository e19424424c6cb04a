// k_tree.hpp — ANY query tree over whole-corpus bitmaps (TASK_TREE): what the CNF kernels and the truth-table kernel do not take — a
// multi-word phrase under an OR / NOT / matchsome / <optional>, more distinct terms than a truth table holds, conjunctive normal forms wider
// than MAX_QTERMS.  Part of libtrinity_hip.so (MI355X / gfx950); included by trinity_hip.hip.  New code, no reference source.
//
// The reference builds one iterator per node (queryexec_ctx::build_iterator, exec.cpp:253-449: Phrase :297-324 like any other) and pulls
// documents through the tree one advance() at a time (docset_iterators.cpp:226-677).  Here a tree is evaluated as SET ALGEBRA, 32 documents
// per machine word, over one bitmap per leaf:
//   * a TERM leaf reads plane A of its term's row in the batch's tree rows — k_term_planes decodes every distinct term the batch's tree
//     queries name once per run (planes B / C: "frequency is not 1 / nor 2", for the scorers);
//   * a multi-word PHRASE leaf reads a bitmap of the phrase's matches.  The planner added the phrase to the batch as a HIDDEN query (the
//     conjunction of its terms + the positional check: k_and / k_and_dense / k_psets, then k_phrase — the kernels every phrase query runs
//     through, match counts capped as exec.cpp:296 says); k_tree_gather joins its tasks' segments into one ascending list (+ the phrase's
//     score per match) and scatters the list into the bitmap;
//   * k_tree_eval walks the nodes in postfix order per word: Conjuction = AND of the children's words, Disjunction = OR, DisjunctionSome =
//     a bit-sliced count of the children compared with the threshold, Filter = required AND NOT excluded, Optional = its main side; the
//     root's word, less the segment's masked documents, is the query's match bitmap, counted per chunk of the docID space;
//   * k_tree_expand turns the bitmap into the ascending docID list every consumer of a docset reads (a chunk's place in the query's region
//     is the sum of the chunks before it: the matches are contiguous, the region is bound by the tree's upper bound — planner.hpp);
//   * k_tree_leaves (AccumulatedScore / default mode) does per MATCH what the reference's recursion over the iterators that sit on the
//     document does (score wrappers docset_iterators_scorers.cpp:38-228; collect_doc_matching_terms queryexec_ctx.cpp:382-520): node
//     values from the leaves' bits, then top-down which nodes are reached — every child of a Conjuction, the matching children of a
//     Disjunction / DisjunctionSome, the required side of a Filter, Optional's main side and its optional side where that matches — and
//     the reached leaves add their scores (a term's frequency from planes B / C, from the postings beyond 2; a phrase's score from its
//     hidden query's list) or name their reportable terms (k_rich then fetches exactly those terms' hits);
//   * k_tree_topk keeps a query's best k of the scored stream (the application-side heap, matches.h:155-171).
// The bitmaps cost memory, not time (a plane of 10 M documents is 1.25 MB: thousands fit); the planner bounds them (tree_max_bytes).
#pragma once

constexpr int TREE_WG = 256;
static_assert(TREE_CHUNK_WORDS % TREE_WG == 0, "a chunk is a whole number of passes of the workgroup");
constexpr uint32_t TREE_PASSES = TREE_CHUNK_WORDS / TREE_WG;

struct TreeNodes {
        DevTreeNode node[TREE_MAX_NODES];
        uint32_t red[TREE_WG / 64];
        uint32_t nn;
};

// the query's record into LDS (every thread of the workgroup calls it); returns the node count
__device__ __forceinline__ uint32_t tree_load(TreeNodes &sh, const uint32_t *__restrict__ rec) {
        const uint32_t nn = min(uni(rec[0]), TREE_MAX_NODES);
        for (uint32_t i = threadIdx.x; i < nn * (uint32_t)(sizeof(DevTreeNode) / 4); i += TREE_WG)
                ((uint32_t *)sh.node)[i] = rec[TREE_HDR_WORDS + i];
        __syncthreads();
        return nn;
}
// plane A of a leaf's row
__device__ __forceinline__ const uint32_t *tree_row(const DevTreeNode &nd, const uint32_t *__restrict__ trows, const uint32_t *__restrict__ prows, const uint32_t plw) {
        return (nd.row & TREE_ROW_PHRASE) ? prows + (size_t)(nd.row & ~TREE_ROW_PHRASE) * plw : trows + (size_t)nd.row * PL_PLANES * plw;
}
// sum over the workgroup (every thread calls it, every thread gets it)
__device__ __forceinline__ uint32_t tree_block_sum(TreeNodes &sh, uint32_t v) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
                v += __shfl_xor(v, d, 64);
        __syncthreads();
        if ((threadIdx.x & 63u) == 0)
                sh.red[threadIdx.x >> 6] = v;
        __syncthreads();
        uint32_t s = 0;
#pragma unroll
        for (int w = 0; w < TREE_WG / 64; ++w)
                s += sh.red[w];
        return uni(s);
}
// exclusive prefix over the workgroup's threads; total: the sum
__device__ __forceinline__ uint32_t tree_block_scan(TreeNodes &sh, const uint32_t v, uint32_t &total) {
        uint32_t wtot;
        const uint32_t ex = wave_excl_scan(v, wtot);
        __syncthreads();
        if ((threadIdx.x & 63u) == 0)
                sh.red[threadIdx.x >> 6] = wtot;
        __syncthreads();
        uint32_t base = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < TREE_WG / 64; ++w) {
                base += w < (int)(threadIdx.x >> 6) ? sh.red[w] : 0u;
                tot += sh.red[w];
        }
        total = uni(tot);
        return base + ex;
}

// ---- the hidden phrase queries: one workgroup each.  The tasks' segments (k_phrase compacted each in place) move down into ONE ascending
//      list at the head of the query's region — scores alongside —, the list is scattered into the phrase's bitmap row (cleared by the
//      launcher), and the first task's count becomes the list's length (the other tasks': 0), so that whatever reads the hidden query's
//      docset afterwards still reads a consistent one.
__global__ __launch_bounds__(TREE_WG) void k_tree_gather(const DevQuery *__restrict__ plan, const DevTask *__restrict__ tasks, const uint32_t *__restrict__ hidden,
                                                         uint32_t *__restrict__ out, uint32_t *__restrict__ counts, double *__restrict__ pscore,
                                                         uint32_t *__restrict__ prows, const uint32_t plw) {
        const uint32_t tid = threadIdx.x;
        const DevQuery q = plan[hidden[blockIdx.x]];
        uint32_t *const dst = out + q.out_off;
        double *const pdst = pscore ? pscore + q.out_off : nullptr;
        uint32_t pos = 0;
        for (uint32_t t = 0; t < q.ntasks; ++t) {
                const uint32_t n = uni(counts[q.first_task + t]);
                const uint64_t so = tasks[q.first_task + t].out_off;
                if (so != q.out_off + pos)
                        for (uint32_t i0 = 0; i0 < n; i0 += TREE_WG) { // (the list moves DOWN: a round's stores never reach what a later round still has to read)
                                const uint32_t j = i0 + tid;
                                const uint32_t v = j < n ? out[so + j] : 0u;
                                const double pv = (pscore && j < n) ? pscore[so + j] : 0.0;
                                __syncthreads();
                                if (j < n) {
                                        dst[pos + j] = v;
                                        if (pdst)
                                                pdst[pos + j] = pv;
                                }
                                __syncthreads();
                        }
                pos += n;
        }
        __syncthreads();
        uint32_t *const row = prows + (size_t)blockIdx.x * plw;
        for (uint32_t j = tid; j < pos; j += TREE_WG) {
                const uint32_t d = dst[j];
                atomicOr(&row[d >> 5], 1u << (d & 31u));
        }
        if (tid == 0)
                for (uint32_t t = 0; t < q.ntasks; ++t)
                        counts[q.first_task + t] = t ? 0u : pos;
}

// ---- the tree per bitmap word.  grid: (chunks of the docID space, tree queries); sched[y]: the query's task
struct TreeEvalShared {
        TreeNodes t;
        uint32_t val[TREE_MAX_NODES][TREE_WG]; // per node, per thread: the node's word (a thread reads and writes its own column only)
};
__global__ __launch_bounds__(TREE_WG) void k_tree_eval(const DevQuery *__restrict__ plan, const DevTask *__restrict__ tasks, const uint32_t *__restrict__ sched,
                                                       const uint32_t *__restrict__ tree, const uint32_t *__restrict__ trows, const uint32_t *__restrict__ prows,
                                                       const uint32_t *__restrict__ masked, uint32_t *__restrict__ qbits, uint32_t *__restrict__ chunk_counts,
                                                       const uint32_t plw) {
        __shared__ TreeEvalShared sh;
        const uint32_t tid = threadIdx.x, chunk = blockIdx.x, nchunks = gridDim.x, qi = blockIdx.y;
        const DevQuery q = plan[tasks[sched[qi]].slot];
        const uint32_t nn = tree_load(sh.t, tree + q.fused_idx);
        uint32_t count = 0;
        for (uint32_t pass = 0; pass < TREE_PASSES; ++pass) {
                const uint32_t w = chunk * TREE_CHUNK_WORDS + pass * TREE_WG + tid;
                if (w >= plw)
                        continue;
                for (uint32_t n = 0; n < nn; ++n) { // (uniform: the record is the same for every thread)
                        const DevTreeNode &nd = sh.t.node[n];
                        const uint32_t op = uni((uint32_t)nd.op);
                        uint32_t v;
                        if (op == TRI_OP_TERM || op == TRI_OP_PHRASE)
                                v = tree_row(nd, trows, prows, plw)[w];
                        else if (op == TRI_OP_NOT) // Filter (docset_iterators.cpp:652-677)
                                v = sh.val[nd.kid0][tid] & ~sh.val[nd.kid1][tid];
                        else if (op == TRI_OP_OPT) // Optional (docset_iterators.h:174-206): the documents of its main side
                                v = sh.val[nd.kid0][tid];
                        else {
                                uint64_t km = nd.kids;
                                if (op == TRI_OP_AND) {
                                        v = 0xffffffffu;
                                        for (; km; km &= km - 1ull)
                                                v &= sh.val[__builtin_ctzll(km)][tid];
                                } else if (op == TRI_OP_OR) {
                                        v = 0;
                                        for (; km; km &= km - 1ull)
                                                v |= sh.val[__builtin_ctzll(km)][tid];
                                } else { // DisjunctionSome (docset_iterators.cpp:733-860): at least thr of the children — a bit-sliced counter per document
                                        uint32_t c[7] = {0, 0, 0, 0, 0, 0, 0};
                                        for (; km; km &= km - 1ull) {
                                                uint32_t carry = sh.val[__builtin_ctzll(km)][tid];
#pragma unroll
                                                for (int p = 0; p < 7; ++p) {
                                                        const uint32_t t = c[p] & carry;
                                                        c[p] ^= carry;
                                                        carry = t;
                                                }
                                        }
                                        // count >= thr, from the top bit down: greater so far, or equal so far and this bit decides
                                        const uint32_t thr = nd.thr;
                                        uint32_t gt = 0, eq = 0xffffffffu;
#pragma unroll
                                        for (int p = 6; p >= 0; --p) {
                                                const uint32_t tb = ((thr >> p) & 1u) ? 0xffffffffu : 0u;
                                                gt |= eq & c[p] & ~tb;
                                                eq &= ~(c[p] ^ tb);
                                        }
                                        v = gt | eq;
                                }
                        }
                        sh.val[n][tid] = v;
                }
                uint32_t m = sh.val[nn - 1][tid];
                if (masked) // masked_documents_registry::test (docidupdates.h:90-119): documents updated / deleted elsewhere never match
                        m &= ~masked[w];
                qbits[(size_t)qi * plw + w] = m;
                count += __popc(m);
        }
        const uint32_t total = tree_block_sum(sh.t, count);
        if (tid == 0)
                chunk_counts[(size_t)qi * nchunks + chunk] = total;
}

// matches of the chunks before `chunk` (every thread calls it)
__device__ __forceinline__ uint32_t tree_chunk_base(TreeNodes &sh, const uint32_t *__restrict__ cc, const uint32_t chunk) {
        uint32_t s = 0;
        for (uint32_t c = threadIdx.x; c < chunk; c += TREE_WG)
                s += cc[c];
        return tree_block_sum(sh, s);
}

// ---- the match bitmap as the ascending docID list.  Same grid
__global__ __launch_bounds__(TREE_WG) void k_tree_expand(const DevQuery *__restrict__ plan, const DevTask *__restrict__ tasks, const uint32_t *__restrict__ sched,
                                                         const uint32_t *__restrict__ qbits, const uint32_t *__restrict__ chunk_counts, uint32_t *__restrict__ out,
                                                         uint32_t *__restrict__ counts, const uint32_t plw) {
        __shared__ TreeNodes sh;
        const uint32_t tid = threadIdx.x, chunk = blockIdx.x, nchunks = gridDim.x, qi = blockIdx.y;
        const uint32_t tix = sched[qi];
        const DevQuery q = plan[tasks[tix].slot];
        uint32_t run = tree_chunk_base(sh, chunk_counts + (size_t)qi * nchunks, chunk);
        for (uint32_t pass = 0; pass < TREE_PASSES; ++pass) {
                const uint32_t w = chunk * TREE_CHUNK_WORDS + pass * TREE_WG + tid;
                uint32_t m = w < plw ? qbits[(size_t)qi * plw + w] : 0u;
                uint32_t total;
                uint32_t at = run + tree_block_scan(sh, (uint32_t)__popc(m), total);
                for (; m; m &= m - 1u, ++at)
                        if (at < q.out_cap) // (the planner's bound holds: belt and braces against a store outside the region)
                                out[q.out_off + at] = w * 32u + (uint32_t)__builtin_ctz(m);
                run += total;
        }
        if (chunk + 1 == nchunks && tid == 0)
                counts[tix] = run;
}

// ---- per match: which leaves' iterators sit on it, what they score / report.  Same grid (a workgroup takes the matches of its chunk)
template <int CODEC>
__global__ __launch_bounds__(TREE_WG) void k_tree_leaves(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last, const uint32_t *__restrict__ blk_off,
                                                         const DevTerm *__restrict__ terms, const DevQuery *__restrict__ plan, const DevTask *__restrict__ tasks,
                                                         const uint32_t *__restrict__ sched, const uint32_t *__restrict__ tree, const uint32_t *__restrict__ trows,
                                                         const uint32_t *__restrict__ prows, const uint32_t *__restrict__ chunk_counts, const uint32_t *__restrict__ out,
                                                         const uint32_t *__restrict__ counts, const double *__restrict__ sweights, const double *__restrict__ pscore,
                                                         double *__restrict__ all_scores, uint32_t *__restrict__ allow, const uint32_t plw, const int sim) {
        __shared__ TreeNodes sh;
        const uint32_t tid = threadIdx.x, chunk = blockIdx.x, nchunks = gridDim.x, qi = blockIdx.y;
        const DevQuery q = plan[tasks[sched[qi]].slot];
        const uint32_t nn = tree_load(sh, tree + q.fused_idx);
        const uint32_t *cc = chunk_counts + (size_t)qi * nchunks;
        const uint32_t base = tree_chunk_base(sh, cc, chunk), cnt = uni(cc[chunk]);
        for (uint32_t j = tid; j < cnt; j += TREE_WG) {
                if (base + j >= q.out_cap)
                        break;
                const uint64_t o = q.out_off + base + j;
                const uint32_t doc = out[o], wi = doc >> 5, bit = doc & 31u;
                // node values for this document, leaves up
                uint64_t val = 0;
                for (uint32_t n = 0; n < nn; ++n) {
                        const DevTreeNode &nd = sh.node[n];
                        bool v;
                        switch (nd.op) {
                                case TRI_OP_TERM:
                                case TRI_OP_PHRASE:
                                        v = (tree_row(nd, trows, prows, plw)[wi] >> bit) & 1u;
                                        break;
                                case TRI_OP_AND:
                                        v = (val & nd.kids) == nd.kids;
                                        break;
                                case TRI_OP_OR:
                                        v = (val & nd.kids) != 0ull;
                                        break;
                                case TRI_OP_SOME:
                                        v = (uint32_t)__popcll(val & nd.kids) >= nd.thr;
                                        break;
                                case TRI_OP_NOT:
                                        v = ((val >> nd.kid0) & 1ull) && !((val >> nd.kid1) & 1ull);
                                        break;
                                default: // TRI_OP_OPT
                                        v = (val >> nd.kid0) & 1ull;
                                        break;
                        }
                        val |= (uint64_t)v << n;
                }
                // reached nodes, root down: the iterators the reference's recursion visits on this document
                uint64_t reach = 1ull << (nn - 1);
                for (uint32_t n = nn - 1; n-- > 0;) {
                        const DevTreeNode &nd = sh.node[n];
                        const uint32_t pop = sh.node[nd.parent].op;
                        const bool mine = (val >> n) & 1ull;
                        const bool via = pop == TRI_OP_AND ? true : (pop == TRI_OP_OR || pop == TRI_OP_SOME) ? mine : pop == TRI_OP_NOT ? nd.ord == 0 : (nd.ord == 0 || mine);
                        reach |= (uint64_t)(((reach >> nd.parent) & 1ull) && via) << n;
                }
                double s = 0.0;
                uint32_t rep = 0;
                for (uint32_t n = 0; n < nn; ++n) {
                        const DevTreeNode &nd = sh.node[n];
                        if (!((reach >> n) & 1ull) || (nd.op != TRI_OP_TERM && nd.op != TRI_OP_PHRASE))
                                continue;
                        rep |= nd.rmask;
                        if (!all_scores || nd.score == 0xffffffffu)
                                continue;
                        if (nd.op == TRI_OP_TERM) {
                                const uint32_t *pa = tree_row(nd, trows, prows, plw);
                                // (the row's interleaved level words: the frequency itself up to PL_NESTED - 1; the top level: read it from the postings)
                                const uint32_t *lv = pa + (size_t)PL_STORED * plw + 3u * wi;
                                uint32_t f = ((lv[0] >> bit) & 1u) | (((lv[1] >> bit) & 1u) << 1) | (((lv[2] >> bit) & 1u) << 2);
                                if (f == PL_NESTED || !f)
                                        f = fused_lookup_freq<CODEC>(index, blk_last, blk_off, terms[nd.arg], doc);
                                s += (double)sim_score(sim, sweights[q.score_base + nd.score], f);
                        } else { // the phrase's score for this document: its hidden query's list holds it (k_phrase: scorer->score(id, matchCnt, weight))
                                const DevQuery hq = plan[nd.arg];
                                uint32_t lo = 0, hi = counts[hq.first_task];
                                while (lo < hi) {
                                        const uint32_t mid = (lo + hi) >> 1;
                                        if (out[hq.out_off + mid] < doc)
                                                lo = mid + 1;
                                        else
                                                hi = mid;
                                }
                                s += pscore[hq.out_off + lo];
                        }
                }
                if (all_scores)
                        all_scores[o] = s;
                if (allow)
                        allow[o] = rep;
        }
}

// ---- a query's best k of its scored stream: one workgroup per tree query
struct TreeTopShared {
        TopK tk;
        uint32_t scan[8];
};
__global__ __launch_bounds__(AND_WG) void k_tree_topk(const uint32_t *__restrict__ sched, const DevTask *__restrict__ tasks, const uint32_t *__restrict__ out,
                                                      const uint32_t *__restrict__ counts, const double *__restrict__ all_scores, const uint32_t k,
                                                      uint32_t *__restrict__ part_docs, double *__restrict__ part_scores, uint32_t *__restrict__ part_counts) {
        __shared__ TreeTopShared sh;
        const uint32_t tid = threadIdx.x;
        const uint32_t tix = sched[blockIdx.x];
        const DevTask task = tasks[tix];
        const uint32_t M = uni(counts[tix]);
        sh.tk.n = 0; // (uniform stores)
        sh.tk.full = 0;
        __syncthreads();
        for (uint32_t b0 = 0; b0 < M; b0 += AND_WG) {
                const uint32_t j = b0 + tid;
                topk_offer(sh.tk, k, j < M, j < M ? all_scores[task.out_off + j] : 0.0, j < M ? out[task.out_off + j] : 0u, sh.scan);
        }
        topk_prune(sh.tk, k, sh.scan);
        const uint32_t n = uni(sh.tk.n);
        for (uint32_t i = tid; i < n; i += AND_WG) {
                part_docs[(uint64_t)tix * k + i] = sh.tk.d[i];
                part_scores[(uint64_t)tix * k + i] = sh.tk.s[i];
        }
        if (tid == 0)
                part_counts[tix] = n;
}
