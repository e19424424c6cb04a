// planner.hpp — the host planner behind tri_batch_create: postfix query programs -> the plan the kernels execute.
//
// Takes over, for a whole batch of queries at once, what the reference does per query before the first posting is touched:
// queryexec_ctx::build_iterator (exec.cpp:253-449: flattening of nested AND / OR :339-358, 382-393, operand ordering by cost
// :35-110, 133-240), build_span's choice of execution strategy (:452-505) and the scorer-weight set-up (similarity.h:179-226) —
// restated as data: CNF groups or a truth table per query, an execution class (candidate tiles / bitmap windows / one-pass scored
// windows / bit planes), tasks cut to even cost with private output regions, the head terms the batch shares (term planes), and one
// contiguous host block that is copied to the device in a single transfer.
//
// Host-only C++17, no HIP: trinity_hip.hip materialises the plan on the device; tools/plan_probe.cpp and tests/test_planner.py drive it
// without one.  The queries of a batch are independent, so every pass over them runs on a few host threads (HostPool) over contiguous
// fragments of the batch; what is global (the one-pass task size, the chosen planes, offsets into the shared arrays) is settled between
// the passes from per-fragment sums.  New code, no reference source.
#pragma once
#include "host_pool.hpp"
#include "index_host.hpp"

#include <array>
#include <chrono>
#include <cmath>
#include <memory>
#include <numeric>

// planner / launch options of a device handle (tri_dev_set_option); the defaults are what bench.py measures
struct tri_options {
        uint64_t dense_min_postings = 512 * 1024; // TASK_DENSE needs at least this many postings over the query's lists (0: every multi-term query)
        uint64_t dense_task_cost = 192 * 1024;    // postings per bitmap-window task
        uint64_t dense_window_cost = 16 * 1024;   // ... a docID window counts this many postings whatever it holds (k_and_dense: directory look-ups, barriers, the sweep:
                                                  // measured 28 us a window on unions of rare terms — a task of 77 near-empty windows ran 2.2 ms, cfg5's k_and_dense 58 % busy)
        uint64_t cand_task_cost = 32 * 1024;      // cost units (postings decoded + 32 per partner block that can hold a candidate) per candidate-tile task
                                                  // (cfg3's k_and, ms at 96 K / 32 K / 8 K: 0.61 / 0.60 / 0.60 — before the galloping merge 2.93 / 1.75 / 1.77)
        uint64_t fused = 1;                       // AccumulatedScore top-K of dense queries in one pass (k_fused); 0: k_and_dense + k_score
        uint64_t fused_task_cost = 0;             // postings per one-pass task; 0: sized from the batch (256 K .. 8 M, about two tasks per resident workgroup)
        uint64_t fused_freq_cap = 0;              // 0: the field width decides; else a smaller saturation point (exercises the rescoring path)
        uint64_t account_needed_bytes = 0;        // 1: tri_batch_create also works out tri_batch_info.cand_needed_bytes (a directory walk per candidate-tile query)
        uint64_t fused_halfwords = 1;             // 16-bit window words for queries of <= 5 distinct terms (windows twice as long); 0: always 32-bit
        uint64_t overlap_dense_wgs = 0, overlap_cand_wgs = 0; // both non-zero: the two matching kernels side by side on two streams
        uint64_t overlap = 0;                                 // 1: the candidate-tile kernel (k_and) on a second stream beside the window kernels (k_and_dense, k_psets, k_probe), full grids
        uint64_t planes = 7;     // term planes (k_planes.hpp), a bit set: 1 k_and probes them, 2 k_and_dense ORs them in, 4 top-K CNF queries run in k_planes; 0: off
        uint64_t planes_split = 0; // a k_planes query is cut into this many docID ranges (tasks) that share its threshold; 0: 2 or 3 by the batch's size; >= 65536: by postings like the other one-pass tasks.  cfg3's unions: 0 10.9 ms, 2 8.0, 3 8.5, 4 9.2 (a task has fixed costs)
        uint64_t plane_div = 1024; // a term gets a plane when it holds at least docs_cnt / plane_div documents (and the batch's uses repay one decode of its list).
                                 // Round 5 (k_planes' level words, k_and's row queues), step ms at 512 / 1024 / 2048: cfg3 5.95 / 5.63 / 5.57, cfg5 7.42 / 7.27 / 7.28, cfg2 1.30 / 1.31 / 1.31
                                 // — 698 rows at cfg3 (8.75 MB each at 10 M documents).  Earlier rounds:
                                 // While a batch built its own planes: step ms at 32 / 64 / 128 / 256: cfg3 16.9 / 15.9 / 15.3 / 15.4, cfg2 - / 2.90 / 2.71 / 2.76 (the
                                 // build grew with it).  The planes live with the index now (built once): 128 / 512 / 4096: cfg2 1.48 / 1.44 / 1.40, cfg3 12.64 / 12.40 / 12.4,
                                 // cfg4 17.4 / 16.9 / 16.9 — 355 rows (1.3 GB at 10 M documents) at 512
        uint64_t plane_amortize = 1;           // a term is given a plane when plane_amortize x (the postings the batch's uses of it save) repay one decode of its list: the rows live with the
                                               // index, so a stream of batches repays a row over several of them (1: every batch repays its own rows)
        uint64_t plane_max_bytes = 8ull << 30; // scratch budget of a batch's term planes: the eligible terms are the longest lists that fit (each costs PL_PLANES bitmaps over the docID space and one decode per index)
        uint64_t planes_rebuild = 0;           // 1: every tri_batch_run decodes the plane rows its batch names AGAIN (a cold plane cache: what a query stream pays whose head
                                               // terms have all just been evicted) — a measurement switch (bench.py's rotating leg), never a speed-up
        uint64_t phrase_task_div = 0;          // the tasks of a query with phrases are cut this many times finer (they are k_phrase's tasks too, and a phrase candidate costs far more than the
                                               // planner's unit: k_phrase's span is its longest task); 0: by the batch's phrase queries per compute unit (4 / 2 / 1: plan_batch)
        uint64_t planes_order = 1;             // k_planes' tasks: 1 docID range by range, within a range by the heaviest plane row they sweep (the workgroups in flight stream the same
                                               // head rows from about the same place: those words come from L2); 2: row by row, a query's ranges side by side; 0: heaviest task first.
                                               // Round 6, k_planes ms at 0 / 1 / 2: cfg3 4.40 / 4.12 / 4.33, cfg5's shard 2.17 / 2.10 / 2.20 (with four ranges a query: 5.44 / 4.73 / -)
        uint64_t scatter_bitmap_slack = 4;     // a DocumentsOnly union of head terms with terms that have no plane runs in k_psets (PSET_UNIT_SCATTER: plane words OR-ed, the other terms'
                                               // documents listed by k_psets_prep) — its result a bitmap — when its head terms hold one document in 32 x slack or more; 1: only where a
                                               // bitmap is no larger than the docID list (the rule of every other query).  The rest of those unions decode every list into LDS window
                                               // bitmaps (k_and_dense): 1.9 us a query against 0.5 (cfg5's 100 K batch).  A result at one document in 128 costs 4 x the bytes as a bitmap
        uint64_t pset_order = 1;               // k_psets' tasks: 1 docID range by range and, within a range, by the query's HEAVIEST term (PSET_SUBS places by its df rank): the workgroups in
                                               // flight read that term's words of the range one after the other — the second and later readers from L2, not over the fabric; 0: batch order
                                               // within a range
        uint64_t cand_xcd = 1;                 // k_and's tasks queued per XCD by the plane row they probe (planner.hpp "k_and's queues"); 0: the cost order dealt round the queues
        uint64_t plan_threads = 0;             // host threads a planner context plans with; 0: up to 16, one per 512 queries, within the process's CPU budget (affinity mask, cgroup quota) shared by the handle's contexts
        uint64_t plan_hot_us = 300;            // ... keep polling for a job this long after their last one before they sleep (a polling thread uses a CPU of the process's quota;
                                               // the gaps between the passes of one create are below 0.1 ms).  Read when a pool starts
        uint64_t plan_pin = 2;                 // the planner's workers: 1 each pinned to ONE CPU; 2 to its pool's STRETCH of CPUs (host_pool.hpp: spread); 0 not pinned at all.  Read when a pool starts
        uint64_t result_bitmaps = 1;           // DocumentsOnly: a bitmap-window query whose expected matches outnumber the words of a bitmap over its docID range
                                               // delivers its docID set AS that bitmap (RESULT_BITMAP, dev_structs.hpp); 0: always ascending docIDs
        uint64_t tree_max_bytes = 16ull << 30; // scratch budget of a batch's TASK_TREE queries (a PL_PLANES-plane row per distinct term leaf, a plane per phrase leaf and per query)
        uint64_t probe_max_blocks = 0;         // > 0: a lead list of at most this many blocks against lists that all have planes runs in k_probe (a wave per task) instead of
                                               // k_and's candidate tiles.  Off by default — measured at cfg2 (step ms / k_probe / k_and): 0: 2.14 / - / 0.78; 64: 2.25 / 0.15 / 0.75;
                                               // 256: 2.26 / 0.22 / 0.70; 1024: 2.35 / 0.41 / 0.59; all: 2.55 / 0.82 / 0.41 — k_and's time is its tail, not its task count
};

struct PlanEnv {
        tri_options opt;
        uint32_t cus = 256;          // compute units of the device (task sizes aim at a couple of tasks per resident workgroup)
        uint32_t fus_wgs_per_cu = 2; // k_fused workgroups a CU holds (LDS)
        uint32_t plk_wgs_per_cu = 2; // k_planes workgroups a CU holds
};

template <class T>
struct Span { // a typed window into the plan's host block
        T *p = nullptr;
        size_t n = 0;
        size_t size() const { return n; }
        bool empty() const { return !n; }
        T *data() { return p; }
        const T *data() const { return p; }
        T &operator[](size_t i) { return p[i]; }
        const T &operator[](size_t i) const { return p[i]; }
        T *begin() { return p; }
        T *end() { return p + n; }
        const T *begin() const { return p; }
        const T *end() const { return p + n; }
};

struct PlanInput {
        const uint32_t *prog = nullptr;
        size_t prog_len = 0;
        const tri_query *queries = nullptr;
        size_t nq = 0;
        const double *weights = nullptr; // optional: one ScorerWeight per program token
        uint32_t flags = 0, topk = 0;
        int similarity = TRI_SIM_BM25;
};

// What tri_batch_create hands to the device and keeps on the host to read results back.  Every array lives in ONE block (64-byte aligned
// sections, `off_*` = a section's byte offset): the device copy is one transfer of block[0, block_bytes) and a section's device address is
// arena + off_*.
struct BatchPlan {
        uint8_t *block = nullptr;
        size_t block_bytes = 0;
        Span<DevQuery> plan;        // one per lowered query, in query order
        Span<uint32_t> qterms;      // CNF term lists (QT_GROUP / QT_NOT marks)
        Span<DevTask> tasks;        // a query's tasks are consecutive
        Span<uint32_t> sched;       // task indices by kernel, heaviest first: [0, n_dense) TASK_DENSE, then TASK_PSET, TASK_PROBE, TASK_CAND, TASK_FUSED, TASK_FUSED16, TASK_FUSED_GEN, TASK_PLANES, TASK_PLANES8
        Span<DevFused> fused;       // slot maps of the one-pass queries (DevQuery::fused_idx)
        Span<uint32_t> qplane;      // parallel to qterms: the term's row in the batch's term planes, or PL_NONE (empty: no planes)
        Span<uint32_t> plane_terms; // row -> term
        Span<uint32_t> splane;      // parallel to sterms (scored batches with planes; else empty): the scorer's term's plane row, or PL_NONE — k_score reads a
                                    // match's frequency off planes B / C instead of decoding a block of the term
        size_t off_splane = 0;
        Span<uint32_t> sterms;      // scored: scorer terms in the reference's summation order; default mode: reportable terms
        Span<double> sweights;      // scored: their ScorerWeights
        Span<DevPhrase> phrases;
        Span<uint32_t> pterms, ptasks;
        Span<DevPsetUnit> units;    // the TASK_PSET and TASK_PROBE tasks as k_psets / k_probe read them (task order) ...
        Span<uint32_t> pset_sched;  // ... and the order they are run in, as unit indices: [0, n_pset) TASK_PSET, docID window range by window range; then
                                    // the n_probe TASK_PROBE ones, heaviest first
        size_t off_units = 0, off_pset_sched = 0;
        Span<uint32_t> cand_q;      // k_and's task queues, one per XCD: queue x = the TASK_CAND section of sched at [cand_q[x], cand_q[x + 1]) (CAND_QUEUES + 1 bounds)
        size_t off_cand_q = 0;
        Span<uint32_t> tree;        // TASK_TREE records: TREE_HDR_WORDS header words + DevTreeNode per node (DevQuery::fused_idx: the record's first word)
        Span<uint32_t> tree_terms;  // the distinct term leaves of the batch's TASK_TREE queries, ascending: term -> row of the batch's tree rows
        Span<uint32_t> tree_hidden; // hidden phrase queries: their plan slots (position: the row of the batch's phrase rows)
        size_t off_tree = 0, off_tree_terms = 0, off_tree_hidden = 0;
        uint32_t n_tree = 0;        // TASK_TREE tasks (the last section of sched)
        uint64_t tree_queries = 0, tree_scratch_bytes = 0;
        uint64_t bitmap_queries = 0; // queries whose docID set is delivered as a bitmap (RESULT_BITMAP)
        uint64_t pscatter_queries = 0, pscatter_docs = 0; // ... of them the unions k_psets runs although some of their terms have no plane (PSET_UNIT_SCATTER), and those terms'
                                                          // documents over all such queries: the slots k_psets_prep lists them in, task by task
        size_t off_plan = 0, off_qterms = 0, off_tasks = 0, off_sched = 0, off_fused = 0, off_qplane = 0, off_plane_terms = 0, off_sterms = 0, off_sweights = 0,
               off_phrases = 0, off_pterms = 0, off_ptasks = 0;
        std::vector<uint32_t> slot_of_query; // caller query -> plan slot (UINT32_MAX: can never match)
        std::vector<int32_t> qstatus;        // per caller query: TRI_OK, or why the planner left it out of the batch (it then reports no matches)
        uint32_t n_dense = 0, n_pset = 0, n_probe = 0, n_cand = 0, n_fused = 0, n_fused16 = 0, n_fusedgen = 0, n_planes = 0, n_planes8 = 0;
        uint32_t plw = 0;        // words of one term plane
        uint32_t plane_rows = 0; // rows the batch may address: the terms eligible for a plane under the options it was planned with (row = df rank)
        uint32_t sparse_cap = 0; // k_planes: list entries a task's decoded slots can need
        uint32_t rich_R = 0;     // default mode: reportable terms of the widest query
        bool rich_allow = false; // default mode: the batch holds general trees (per match: which reportable terms the tree sits on)
        uint64_t out_capacity = 0;
        uint64_t term_bytes = 0, term_bytes_dense = 0, term_bytes_fused = 0, term_bytes_planes = 0, term_bytes_phrase_hits = 0, plane_decoded_bytes = 0,
                 cand_needed_term_bytes = 0;
        uint64_t dense_queries = 0, pset_queries = 0, probe_queries = 0, cand_queries = 0, fused_queries = 0, planes_queries = 0, unsupported_queries = 0;
        uint64_t term_bytes_pset = 0, term_bytes_probe = 0;
        // option account_needed_bytes (a diagnostic of bench.py, untimed): the bytes of the DISTINCT lists the batch's queries name — each
        // list once, however many queries share it: what a batch that shares decodes has to read at least — over the whole batch (doc bytes,
        // plus the hit bytes of the distinct phrase / reported terms) and per execution class (by task kind; [TASK_KINDS]: the phrases' hit bytes)
        uint64_t distinct_bytes = 0, distinct_bytes_kind[TASK_KINDS + 1] = {};
        std::string last_unsupported; // describes the last query that was left out
        double plan_ms[4] = {0, 0, 0, 0}; // lowering + classes, tasks, layout + fill, schedule + planes
};

namespace trip {
        constexpr size_t SECTION_ALIGN = 64;
        constexpr uint32_t SCHED_NB = 64 * 4; // schedule buckets per kernel: cost octave + 2 bits
        constexpr uint32_t CAND_SUBS = 128, CAND_COST_SUBS = 16; // ... k_and's in row order have buckets of their own (behind the kernels': CAND_KEY0): per queue, 16 for the long and the
                                                               // row-less tasks by cost, 111 places for rows, one for the stragglers
        constexpr uint32_t PSET_RANGE_BKS = 64, PSET_SUBS = 64; // ... k_psets' by (docID window range, heaviest term of the query): the ranges of a long docID space share the 64 range buckets
        constexpr uint32_t CAND_KEY0 = TASK_KINDS * SCHED_NB, PSET_KEY0 = CAND_KEY0 + CAND_QUEUES * CAND_SUBS, SCHED_KEYS = PSET_KEY0 + PSET_RANGE_BKS * PSET_SUBS;
        // a TASK_PSET task's tcost word: its first window in the low half, the df rank of its query's heaviest term in the high half
        inline uint32_t pset_sub(const uint32_t rank) { return rank < PSET_SUBS / 2 ? rank : std::min(PSET_SUBS / 2 + (rank - PSET_SUBS / 2) / 8, PSET_SUBS - 1); }
        constexpr uint64_t CAND_ROWS_MIN_LEAD = 1024;
        // launch order of the task kinds: TASK_DENSE, TASK_PSET, TASK_PROBE, TASK_CAND, then the one-pass kinds as numbered
        constexpr uint32_t SCHED_RANK[TASK_KINDS] = {3, 0, 4, 5, 6, 7, 8, 1, 2, 9};
        inline uint32_t sched_key(const uint32_t kind, const uint64_t cost) {
                if (kind == TASK_PSET) // by docID window range, ascending (`cost` holds the first window)
                        return SCHED_RANK[TASK_PSET] * SCHED_NB + (uint32_t)std::min<uint64_t>((cost & 0xffffffffull) / PSET_TASK_WINDOWS, SCHED_NB - 1);
                const uint64_t c = std::max<uint64_t>(1, cost);
                const uint32_t lg = 63u - (uint32_t)__builtin_clzll(c);
                const uint32_t frac = lg >= 2 ? (uint32_t)((c >> (lg - 2)) & 3u) : (uint32_t)((c << (2 - lg)) & 3u);
                return SCHED_RANK[kind] * SCHED_NB + (SCHED_NB - 1 - (lg * 4 + frac));
        }

        struct PNode {
                uint32_t op = 0, term = 0;
                uint32_t tok = 0; // index of the program token this node came from (caller-supplied ScorerWeights are per token)
                uint32_t kid_off = 0, kid_n = 0; // children: kidpool[kid_off, +kid_n)
                uint64_t cost = 0;
                bool empty = false;
        };

        // one fragment's scratch for parsing and lowering a query: reused from query to query (clear() keeps the capacity — a query costs
        // no allocation once the vectors have grown to the batch's widest query)
        struct Scratch {
                std::vector<PNode> nodes;
                std::vector<int> kidpool, st, tmpk;
                std::vector<uint64_t> cs;
                std::vector<uint32_t> gt, gs; // CNF groups, flat: group g = gt[gs[g], gs[g + 1])
                std::vector<uint32_t> gorder, leaves, leaf_tok, negs, opts, opt_tok, ts, ts_tok, u, uniq, rt, seen, phterms, slots;
                struct PhraseTmp {
                        uint32_t t0, n;
                        double weight;
                };
                std::vector<PhraseTmp> qphrases;
                std::vector<std::pair<uint32_t, double>> sc;
                const int *kids(const PNode &x) const { return kidpool.data() + x.kid_off; }
        };

        // Parse one postfix program into a tree with the reference's flattening (exec.cpp:339-358, 382-393), emptiness propagation and cost
        // model (exec.cpp:35-110).  Returns root index or -1.
        inline int parse_program(const HostIndex &ix, const uint32_t *prog, uint32_t len, Scratch &S) {
                auto &nodes = S.nodes;
                auto &st = S.st;
                auto &pool = S.kidpool;
                nodes.clear();
                st.clear();
                pool.clear();
                for (uint32_t i = 0; i < len; ++i) {
                        const uint32_t op = prog[i] >> 28, arg = prog[i] & 0x0fffffffu;
                        PNode n;
                        n.op = op;
                        n.tok = i;
                        if (op == TRI_OP_TERM) {
                                n.term = arg;
                                n.cost = arg < ix.terms.size() ? ix.terms[arg].documents : 0;
                                n.empty = n.cost == 0; // unknown term == no documents (index_source.h:60-72)
                        } else {
                                const uint32_t nk = op == TRI_OP_SOME ? (arg & 0xffffu) : arg; // operands taken off the stack
                                if (nk < 1 || nk > st.size())
                                        return -1;
                                S.tmpk.assign(st.end() - nk, st.end());
                                st.resize(st.size() - nk);
                                const std::vector<int> &kids = S.tmpk;
                                n.kid_off = (uint32_t)pool.size();
                                if (op == TRI_OP_SOME) {
                                        // matchsome (exec.cpp:276-283): operands that can never match are dropped; fewer live operands than
                                        // the threshold: never matches.  cost: docset_iterators.cpp:733-742, the (cnt - min + 1) cheapest
                                        const uint32_t mn = arg >> 16;
                                        if (!mn || mn > nk)
                                                return -1;
                                        for (int k : kids)
                                                if (!nodes[k].empty)
                                                        pool.push_back(k);
                                        n.kid_n = (uint32_t)pool.size() - n.kid_off;
                                        n.term = mn; // (the threshold rides in the otherwise unused field)
                                        n.empty = n.kid_n < mn;
                                        S.cs.clear();
                                        for (uint32_t k = 0; k < n.kid_n; ++k)
                                                S.cs.push_back(nodes[pool[n.kid_off + k]].cost);
                                        std::sort(S.cs.begin(), S.cs.end());
                                        for (size_t k = 0; k + mn <= S.cs.size(); ++k)
                                                n.cost += S.cs[k];
                                } else if (op == TRI_OP_PHRASE) {
                                        if (arg > MAX_PHRASE_TERMS) // trinity_limits.h:12 MaxPhraseSize
                                                return -1;
                                        for (int k : kids) {
                                                if (nodes[k].op != TRI_OP_TERM)
                                                        return -1;
                                                n.empty |= nodes[k].empty;
                                                pool.push_back(k);
                                        }
                                        n.kid_n = nk;
                                        n.cost = nodes[kids[0]].cost + UINT32_MAX + (uint64_t)UINT16_MAX * arg;
                                } else if (op == TRI_OP_AND) {
                                        for (int k : kids) {
                                                n.empty |= nodes[k].empty;
                                                if (nodes[k].op == TRI_OP_AND)
                                                        for (uint32_t j = 0; j < nodes[k].kid_n; ++j)
                                                                pool.push_back(pool[nodes[k].kid_off + j]);
                                                else
                                                        pool.push_back(k);
                                        }
                                        n.kid_n = (uint32_t)pool.size() - n.kid_off;
                                        if (n.kid_n <= 16) { // stable insertion sort (std::stable_sort takes a heap buffer per call: a malloc per AND of two terms)
                                                int *kb = pool.data() + n.kid_off;
                                                for (uint32_t a = 1; a < n.kid_n; ++a) {
                                                        const int v = kb[a];
                                                        uint32_t b = a;
                                                        for (; b && nodes[kb[b - 1]].cost > nodes[v].cost; --b)
                                                                kb[b] = kb[b - 1];
                                                        kb[b] = v;
                                                }
                                        } else
                                                std::stable_sort(pool.begin() + n.kid_off, pool.end(), [&](int a, int b) { return nodes[a].cost < nodes[b].cost; });
                                        n.cost = nodes[pool[n.kid_off]].cost;
                                } else if (op == TRI_OP_OR) {
                                        for (int k : kids) {
                                                if (nodes[k].empty)
                                                        continue;
                                                if (nodes[k].op == TRI_OP_OR)
                                                        for (uint32_t j = 0; j < nodes[k].kid_n; ++j)
                                                                pool.push_back(pool[nodes[k].kid_off + j]);
                                                else
                                                        pool.push_back(k);
                                        }
                                        n.kid_n = (uint32_t)pool.size() - n.kid_off;
                                        n.empty = n.kid_n == 0;
                                        for (uint32_t k = 0; k < n.kid_n; ++k)
                                                n.cost += nodes[pool[n.kid_off + k]].cost;
                                } else if (op == TRI_OP_OPT) {
                                        if (arg != 2)
                                                return -1;
                                        if (nodes[kids[1]].empty) { // an optional side that can never match adds nothing
                                                st.push_back(kids[0]);
                                                continue;
                                        }
                                        pool.push_back(kids[0]); // {main, optional}
                                        pool.push_back(kids[1]);
                                        n.kid_n = 2;
                                        n.empty = nodes[kids[0]].empty;
                                        n.cost = nodes[kids[0]].cost;
                                } else if (op == TRI_OP_NOT) {
                                        if (arg != 2)
                                                return -1;
                                        if (nodes[kids[1]].empty) { // [a NOT <never matches>] => a
                                                st.push_back(kids[0]);
                                                continue;
                                        }
                                        pool.push_back(kids[0]); // {required, excluded}
                                        pool.push_back(kids[1]);
                                        n.kid_n = 2;
                                        n.empty = nodes[kids[0]].empty;
                                        n.cost = nodes[kids[0]].cost; // exec.cpp:55-60
                                } else
                                        return -1;
                        }
                        nodes.push_back(n);
                        st.push_back((int)nodes.size() - 1);
                }
                return st.size() == 1 ? st[0] : -1;
        }

        // ---- general trees: what the CNF lowering does not take (matchsome, NOT / Optional of any subtree, AND under OR ...) runs as
        // TASK_FUSED with a truth table over the presence of the query's distinct terms (<= FUS_MAX_SLOTS, no multi-word phrase).
        struct TruthPlan {
                std::vector<uint32_t> slots;            // distinct terms, order of first appearance
                std::vector<uint32_t> leaves, leaf_tok; // scorer leaves (positive TERM nodes) in tree order, and their program tokens
                std::vector<uint32_t> leaf_slot;
                uint32_t tt[8] = {};
                std::vector<std::array<uint32_t, 8>> ctt;
        };
        struct TruthBuilder {
                const Scratch &S;
                TruthPlan &tp;
                std::vector<int> leaf_of_node; // node -> scorer leaf index (-1: none)
                bool ok = true;
                uint32_t slot_of(uint32_t term) {
                        for (size_t i = 0; i < tp.slots.size(); ++i)
                                if (tp.slots[i] == term)
                                        return (uint32_t)i;
                        tp.slots.push_back(term);
                        return (uint32_t)tp.slots.size() - 1;
                }
                // first walk: slots for every term, scorer leaves for the terms an iterator of the tree can report
                void scan(int ni, bool positive) {
                        const PNode &x = S.nodes[ni];
                        if (x.op == TRI_OP_TERM || (x.op == TRI_OP_PHRASE && x.kid_n == 1)) {
                                const PNode &t = x.op == TRI_OP_TERM ? x : S.nodes[S.kids(x)[0]];
                                const uint32_t sl = slot_of(t.term);
                                if (positive) {
                                        leaf_of_node[ni] = (int)tp.leaves.size();
                                        tp.leaves.push_back(t.term);
                                        tp.leaf_tok.push_back(t.tok);
                                        tp.leaf_slot.push_back(sl);
                                }
                                return;
                        }
                        if (x.op == TRI_OP_PHRASE) {
                                ok = false; // a positional constraint is not a function of presence
                                return;
                        }
                        for (uint32_t k = 0; k < x.kid_n; ++k)
                                scan(S.kids(x)[k], positive && !(x.op == TRI_OP_NOT && k == 1));
                }
                uint32_t slot_const(uint32_t term) const {
                        for (size_t i = 0; i < tp.slots.size(); ++i)
                                if (tp.slots[i] == term)
                                        return (uint32_t)i;
                        return 0;
                }
                bool eval(int ni, uint32_t p) const {
                        const PNode &x = S.nodes[ni];
                        const int *kd = S.kids(x);
                        switch (x.op) {
                                case TRI_OP_TERM:
                                        return (p >> slot_const(x.term)) & 1u;
                                case TRI_OP_PHRASE:
                                        return (p >> slot_const(S.nodes[kd[0]].term)) & 1u;
                                case TRI_OP_AND:
                                        for (uint32_t k = 0; k < x.kid_n; ++k)
                                                if (!eval(kd[k], p))
                                                        return false;
                                        return true;
                                case TRI_OP_OR:
                                        for (uint32_t k = 0; k < x.kid_n; ++k)
                                                if (eval(kd[k], p))
                                                        return true;
                                        return false;
                                case TRI_OP_SOME: {
                                        uint32_t c = 0;
                                        for (uint32_t k = 0; k < x.kid_n; ++k)
                                                c += eval(kd[k], p) ? 1u : 0u;
                                        return c >= x.term;
                                }
                                case TRI_OP_NOT: // Filter (docset_iterators.cpp:652-677)
                                        return eval(kd[0], p) && !eval(kd[1], p);
                                case TRI_OP_OPT: // Optional (docset_iterators.h:174-206): the documents of main
                                        return eval(kd[0], p);
                        }
                        return false;
                }
                // the scorer leaves that sit on a document of pattern p, through the tree (node ni matches p): what the reference's score() /
                // collect_doc_matching_terms recursion reaches (docset_iterators_scorers.cpp:38-57, 77-104, 107-193; queryexec_ctx.cpp:382-520)
                void collect(int ni, uint32_t p, uint32_t &mask) const {
                        const PNode &x = S.nodes[ni];
                        const int *kd = S.kids(x);
                        switch (x.op) {
                                case TRI_OP_TERM:
                                case TRI_OP_PHRASE:
                                        if (leaf_of_node[ni] >= 0)
                                                mask |= 1u << leaf_of_node[ni];
                                        break;
                                case TRI_OP_AND:
                                        for (uint32_t k = 0; k < x.kid_n; ++k)
                                                collect(kd[k], p, mask);
                                        break;
                                case TRI_OP_OR:
                                case TRI_OP_SOME:
                                        for (uint32_t k = 0; k < x.kid_n; ++k)
                                                if (eval(kd[k], p))
                                                        collect(kd[k], p, mask);
                                        break;
                                case TRI_OP_NOT:
                                        collect(kd[0], p, mask);
                                        break;
                                case TRI_OP_OPT:
                                        collect(kd[0], p, mask);
                                        if (eval(kd[1], p))
                                                collect(kd[1], p, mask);
                                        break;
                        }
                }
        };
        inline bool build_truth(const Scratch &S, int root, TruthPlan &tp) {
                TruthBuilder tb{S, tp, std::vector<int>(S.nodes.size(), -1)};
                tb.scan(root, true);
                if (!tb.ok || tp.slots.size() > FUS_MAX_SLOTS || tp.leaves.size() > FUS_MAX_LEAVES || tp.leaves.empty())
                        return false;
                tp.ctt.assign(tp.leaves.size(), std::array<uint32_t, 8>{});
                for (uint32_t p = 0; p < (1u << tp.slots.size()); ++p) {
                        if (!tb.eval(root, p))
                                continue;
                        tp.tt[p >> 5] |= 1u << (p & 31u);
                        uint32_t mask = 0;
                        tb.collect(root, p, mask);
                        for (size_t j = 0; j < tp.leaves.size(); ++j)
                                if ((mask >> j) & 1u)
                                        tp.ctt[j][p >> 5] |= 1u << (p & 31u);
                }
                return !(tp.tt[0] & 1u); // (a tree that matches documents holding none of its terms cannot be enumerated from postings)
        }

        // a lowered query before it has its place in the batch
        struct Tmp {
                DevQuery q;
                uint64_t cost;
                uint32_t nlead;
                int32_t fz;   // index into the fragment's slot maps (-1: none): may run in one pass (k_fused / k_planes)
                bool truth;   // a general tree: runs as TASK_FUSED whatever its density (there is no other path for it)
                bool tree;    // ... one the truth table does not hold: TASK_TREE (q.fused_idx: its record in the fragment's treepool; tree_ub: its matches at most)
                bool hidden;  // a phrase evaluated for a TASK_TREE query of the batch (no caller query of its own); hidden_ord: which of the fragment's
                uint64_t tree_ub;
                uint32_t hidden_ord;
                // execution class (second half of the first pass)
                uint64_t sumdf, lead_docs;
                uint32_t last_doc; // no match beyond the (required) group whose lists end first
                bool dense, fuse;
        };

        struct QUse { // a CNF term position that could read a plane
                uint32_t qpos, term;
        };
        struct FUse { // a one-pass slot that reads a plane
                uint32_t fidx, slot, term;
        };

        // everything a fragment (a contiguous range of the batch's queries) produces; offsets are relative to the fragment
        struct Frag {
                size_t q_lo = 0, q_hi = 0;
                Scratch S;
                std::vector<Tmp> tmp;
                std::vector<uint32_t> qterms, pterms, sterms;
                std::vector<double> sweights;
                std::vector<DevPhrase> phrases;
                std::vector<DevFused> fz; // slot maps of the queries that may run in one pass (Tmp::fz)
                uint64_t term_bytes = 0, term_bytes_phrase_hits = 0;
                uint32_t rich_R = 0;
                bool rich_allow = false;
                std::vector<size_t> left_out; // queries the planner does not lower (status TRI_ERR_UNSUPPORTED)
                uint64_t onepass_queries = 0, fused_postings = 0, phrase_queries = 0;
                // second pass
                std::vector<DevTask> tasks; // slot: index into tmp; out_off: relative to the fragment's first output slot
                std::vector<uint64_t> tcost;
                std::vector<DevFused> fused;
                std::vector<uint32_t> ptasks;
                std::vector<DevPsetUnit> units; // tix: index into the fragment's tasks; row[]: filled once the planes are chosen
                std::vector<QUse> quses, suses; // (suses: scorer positions — qpos indexes the fragment's sterms)
                std::vector<FUse> fuses;
                std::vector<uint64_t> benefit; // per eligible term (by df rank): postings of decoding the batch's uses save
                std::vector<uint64_t> cand_row; // per eligible term: tiles (+ 1 a task) of the candidate-tile tasks whose first probed term it is
                std::vector<uint32_t> keys, hist; // (fill pass) per task its schedule bucket; tasks per bucket
                std::vector<uint32_t> treepool;   // TASK_TREE records (DevQuery::fused_idx: a record's first word)
                std::vector<uint32_t> tree_terms; // the term leaves of the fragment's TASK_TREE queries
                uint32_t n_hidden = 0;            // hidden phrase queries (Tmp::hidden_ord)
                uint64_t tree_queries = 0, bitmap_queries = 0, pscatter_queries = 0, pscatter_docs = 0;
                uint64_t off = 0;
                uint32_t sparse_cap = 0;
                uint64_t term_bytes_dense = 0, term_bytes_fused = 0, term_bytes_planes = 0, cand_needed = 0;
                uint64_t dense_queries = 0, pset_queries = 0, probe_queries = 0, cand_queries = 0, fused_queries = 0, planes_queries = 0, term_bytes_pset = 0, term_bytes_probe = 0;
                uint64_t cand_lead_docs = 0, cand_terms = 0; // (candidate-tile queries: their leads' documents, their terms)
                uint64_t probe_demoted = 0, probe_demoted_bytes = 0; // (fill pass) queries whose probes found no plane: candidate tiles after all
                // bases in the batch's arrays (settled between the passes)
                size_t b_plan = 0, b_qterms = 0, b_sterms = 0, b_phrases = 0, b_pterms = 0, b_tasks = 0, b_fused = 0, b_ptasks = 0, b_units = 0, b_tree = 0, b_hidden = 0;
                uint64_t b_off = 0;
                int rc = TRI_OK;
                std::string err;
                // A fragment of an earlier plan as a fresh one that keeps its buffers: every field takes its default, the vectors named below come back
                // EMPTY with their capacity (a vector not named here is simply allocated anew — never stale).  A caller that compiles a batch per step
                // otherwise mallocs, grows by doubling and page-faults about 10 MB of fragment arrays per plan (cfg2, one thread: 8.0 -> 6.4 ms of
                // planning with the memory recycled)
                void recycle() {
                        Frag fresh;
                        auto keep = [](auto &dst, auto &src) {
                                src.clear();
                                dst = std::move(src);
                        };
                        keep(fresh.S.nodes, S.nodes), keep(fresh.S.kidpool, S.kidpool), keep(fresh.S.st, S.st), keep(fresh.S.tmpk, S.tmpk), keep(fresh.S.cs, S.cs);
                        keep(fresh.S.gt, S.gt), keep(fresh.S.gs, S.gs), keep(fresh.S.gorder, S.gorder), keep(fresh.S.leaves, S.leaves), keep(fresh.S.leaf_tok, S.leaf_tok);
                        keep(fresh.S.negs, S.negs), keep(fresh.S.opts, S.opts), keep(fresh.S.opt_tok, S.opt_tok), keep(fresh.S.ts, S.ts), keep(fresh.S.ts_tok, S.ts_tok);
                        keep(fresh.S.u, S.u), keep(fresh.S.uniq, S.uniq), keep(fresh.S.rt, S.rt), keep(fresh.S.seen, S.seen), keep(fresh.S.phterms, S.phterms);
                        keep(fresh.S.slots, S.slots), keep(fresh.S.qphrases, S.qphrases), keep(fresh.S.sc, S.sc);
                        keep(fresh.tmp, tmp), keep(fresh.qterms, qterms), keep(fresh.pterms, pterms), keep(fresh.sterms, sterms), keep(fresh.sweights, sweights);
                        keep(fresh.phrases, phrases), keep(fresh.fz, fz), keep(fresh.left_out, left_out), keep(fresh.tasks, tasks), keep(fresh.tcost, tcost);
                        keep(fresh.fused, fused), keep(fresh.ptasks, ptasks), keep(fresh.units, units), keep(fresh.quses, quses), keep(fresh.suses, suses);
                        keep(fresh.fuses, fuses), keep(fresh.benefit, benefit), keep(fresh.cand_row, cand_row), keep(fresh.keys, keys), keep(fresh.hist, hist), keep(fresh.treepool, treepool);
                        keep(fresh.tree_terms, tree_terms);
                        *this = std::move(fresh);
                }
        };
        // the fragments of a caller's earlier plans (tri_dev keeps one; plan_batch takes what it needs out of it and puts it back)
        struct FragCache {
                std::vector<Frag> frags;
        };

        struct Ctx {
                const HostIndex &ix;
                const PlanEnv &env;
                const PlanInput &in;
                bool scored, rich;
                uint32_t mode;
                // term planes: a term is eligible when its df rank is below n_ok
                uint32_t n_ok = 0;
                bool plane_ok(uint32_t term) const { return ix.df_rank[term] < n_ok; }
                // settled after the first pass
                uint64_t planes_split = 2, fused_task_cost = 0, phrase_task_div = 1;
                uint32_t plw = 0; // words of a bitmap over the docID space (BatchPlan::plw)
                // the ScorerWeight contribution of one term (IndexSourceTermsScorer::new_scorer_weight sums it over a phrase's terms):
                // BM25 similarity.h:179-181 (float math), TF-IDF :85-87 (double), Trivial has none
                double term_weight(const uint32_t df) const {
                        if (in.similarity == TRI_SIM_TFIDF)
                                return std::log((double)((uint64_t)ix.info.docs_cnt + 1) / (double)(df + 1)) + 1.0;
                        if (in.similarity == TRI_SIM_TRIVIAL)
                                return 0.0;
                        const float num = (float)((uint64_t)ix.info.docs_cnt - (uint64_t)df) + 0.5f;
                        const float den = (float)df + 0.5f;
                        return (double)std::log(1 + num / den);
                }
                // first block of `t` whose last docID >= key: the docID-cell index when the list has one and key is a cell boundary (every
                // window boundary is), else a search of the directory column
                uint32_t first_block_ge(const DevTerm &t, const uint64_t key) const {
                        if (t.win_off != 0xffffffffu && !ix.win.empty() && !(key & (CELL_DOCS - 1)) && (key >> CELL_LOG2) < ix.nwin)
                                return ix.win[t.win_off + (key >> CELL_LOG2)];
                        const uint32_t *lb = &ix.blk_last[t.first_block];
                        return (uint32_t)(std::lower_bound(lb, lb + t.nblocks, (uint32_t)std::min<uint64_t>(key, 0xffffffffull)) - lb);
                }
        };

        inline int lower_tree(const Ctx &C, Frag &f, size_t qi, const uint32_t *prog, uint32_t plen, const double *wq, int root);

        // ---- first pass, one query: the program prog[0, plen) of caller query qi lowered into `f` and classed.  wq: the ScorerWeights of the
        //      program's tokens (or null); hidden: a phrase that a TASK_TREE query of the batch reads as a leaf (lower_tree) — it has a plan
        //      slot and tasks like any phrase query, and no caller query of its own
        inline int lower_query(const Ctx &C, Frag &f, const size_t qi, const uint32_t *prog, const uint32_t plen, const double *wq, const bool hidden) {
                const HostIndex &ix = C.ix;
                const PlanInput &in = C.in;
                const tri_options &opt = C.env.opt;
                const bool scored = C.scored, rich = C.rich && !hidden;
                const uint32_t mode = C.mode, topk = in.topk;
                Scratch &S = f.S;
                {
                        const int root = parse_program(ix, prog, plen, S);
                        if (root < 0)
                                return herr(f.err, TRI_ERR_INVALID, "query %zu: malformed postfix program", qi);
                        const std::vector<PNode> &nodes = S.nodes;
                        if (nodes[root].empty)
                                return TRI_OK; // matches nothing (compiles to constfalse in the reference)
                        // ---- conjunctive normal form over terms: AND of (term | OR of terms); a root OR is one group
                        auto &gt = S.gt;
                        auto &gs = S.gs;
                        gt.clear();
                        gs.assign(1, 0u);
                        S.leaves.clear();   // every TERM leaf in evaluation order: one scorer each
                        S.leaf_tok.clear(); // ... and the program token it came from
                        S.qphrases.clear();
                        S.phterms.clear();
                        S.negs.clear();
                        S.opts.clear();
                        S.opt_tok.clear();
                        auto ngroups = [&]() { return (uint32_t)gs.size() - 1; };
                        auto single_seen = [&](uint32_t x) {
                                for (uint32_t g = 0; g < ngroups(); ++g)
                                        if (gs[g + 1] - gs[g] == 1 && gt[gs[g]] == x)
                                                return true;
                                return false;
                        };
                        auto add_group = [&](const PNode &g) -> bool {
                                const int *kd = S.kids(g);
                                if (g.op == TRI_OP_PHRASE && g.kid_n > 1) {
                                        // Phrase = conjunction of its terms + a positional constraint on the matches (k_phrase);
                                        // it scores as ONE iterator with the summed idf (docset_iterators_scorers.cpp:195-228)
                                        Scratch::PhraseTmp ph{(uint32_t)S.phterms.size(), g.kid_n, 0.0};
                                        for (uint32_t k = 0; k < g.kid_n; ++k) {
                                                const uint32_t x = nodes[kd[k]].term;
                                                S.phterms.push_back(x);
                                                ph.weight += C.term_weight(ix.terms[x].documents);
                                                if (!single_seen(x)) {
                                                        gt.push_back(x);
                                                        gs.push_back((uint32_t)gt.size());
                                                }
                                        }
                                        if (wq) // the PHRASE token's own ScorerWeight, when the caller supplies weights (by token position: two phrases
                                                // that start with the same term keep their own weights)
                                                ph.weight = wq[g.tok];
                                        S.qphrases.push_back(ph);
                                        return true;
                                }
                                auto &ts = S.ts;
                                auto &ts_tok = S.ts_tok;
                                ts.clear();
                                ts_tok.clear();
                                if (g.op == TRI_OP_PHRASE) {
                                        ts.push_back(nodes[kd[0]].term); // a one-word phrase is a term (exec.cpp: phrase of size 1)
                                        ts_tok.push_back(nodes[kd[0]].tok);
                                } else if (g.op == TRI_OP_TERM) {
                                        ts.push_back(g.term);
                                        ts_tok.push_back(g.tok);
                                } else if (g.op == TRI_OP_OR) {
                                        for (uint32_t k = 0; k < g.kid_n; ++k) {
                                                if (nodes[kd[k]].op != TRI_OP_TERM)
                                                        return false;
                                                ts.push_back(nodes[kd[k]].term);
                                                ts_tok.push_back(nodes[kd[k]].tok);
                                        }
                                } else
                                        return false;
                                S.leaves.insert(S.leaves.end(), ts.begin(), ts.end());
                                S.leaf_tok.insert(S.leaf_tok.end(), ts_tok.begin(), ts_tok.end());
                                // a term repeated inside a group, or a single-term group seen before, adds nothing to the docID set
                                auto &u = S.u;
                                u.clear();
                                for (uint32_t x : ts)
                                        if (std::find(u.begin(), u.end(), x) == u.end())
                                                u.push_back(x);
                                if (u.size() == 1 && single_seen(u[0]))
                                        return true;
                                gt.insert(gt.end(), u.begin(), u.end());
                                gs.push_back((uint32_t)gt.size());
                                return true;
                        };
                        // logicalnot at the root or under an AND: its required side joins the conjunction, its excluded side (a term or an
                        // OR of terms) joins the query's excluded set: A B -C == A ∧ B ∧ ¬C (Filter semantics, docset_iterators.cpp:652-677)
                        bool ok = true;
                        struct Rec {
                                const std::vector<PNode> &nodes;
                                Scratch &S;
                                bool &ok;
                                decltype(add_group) &add;
                                void side(const PNode &e, std::vector<uint32_t> &terms, std::vector<uint32_t> *toks) {
                                        const int *kd = S.kids(e);
                                        if (e.op == TRI_OP_TERM) {
                                                terms.push_back(e.term);
                                                if (toks)
                                                        toks->push_back(e.tok);
                                        } else if (e.op == TRI_OP_PHRASE && e.kid_n == 1) {
                                                terms.push_back(nodes[kd[0]].term);
                                                if (toks)
                                                        toks->push_back(nodes[kd[0]].tok);
                                        } else if (e.op == TRI_OP_OR) {
                                                for (uint32_t k = 0; k < e.kid_n; ++k) {
                                                        if (nodes[kd[k]].op != TRI_OP_TERM)
                                                                ok = false;
                                                        else {
                                                                terms.push_back(nodes[kd[k]].term);
                                                                if (toks)
                                                                        toks->push_back(nodes[kd[k]].tok);
                                                        }
                                                }
                                        } else
                                                ok = false;
                                }
                                void lower(int ni) {
                                        const PNode &x = nodes[ni];
                                        const int *kd = S.kids(x);
                                        if (x.op == TRI_OP_OPT) {
                                                // Optional(main, opt): the documents of main; opt's terms score (and are reported) where they match —
                                                // exactly how k_score / k_rich treat a term a match does not hold
                                                lower(kd[0]);
                                                side(nodes[kd[1]], S.opts, &S.opt_tok);
                                        } else if (x.op == TRI_OP_NOT) {
                                                lower(kd[0]);
                                                side(nodes[kd[1]], S.negs, nullptr);
                                        } else if (x.op == TRI_OP_AND) {
                                                for (uint32_t k = 0; k < x.kid_n; ++k)
                                                        lower(kd[k]);
                                        } else
                                                ok &= add(x);
                                }
                        } rec{nodes, S, ok, add_group};
                        rec.lower(root);
                        if (ok && ngroups()) // (a general tree — below — counts every term once through its slot list)
                                for (size_t oi = 0; oi < S.opts.size(); ++oi)
                                        if (const uint32_t x = S.opts[oi]; ix.terms[x].documents) {
                                                S.leaves.push_back(x); // one more scorer / reportable term each; never part of the docID set
                                                S.leaf_tok.push_back(S.opt_tok[oi]);
                                                if (mode != TRI_FLAG_DOCUMENTS_ONLY)
                                                        f.term_bytes += ix.docbytes[x]; // its postings are read by k_score / k_rich
                                        }
                        TruthPlan tp;
                        bool truth = false;
                        if (!ok || !ngroups()) {
                                // not a CNF of terms: a general tree over <= FUS_MAX_SLOTS distinct terms runs off a truth table (k_fused.hpp);
                                // anything else — a multi-word phrase below the root conjunction, more terms or leaves — over leaf bitmaps (k_tree.hpp)
                                if (!build_truth(S, root, tp))
                                        return hidden ? herr(f.err, TRI_ERR_INVALID, "query %zu: a phrase leaf that is not a phrase", qi) : lower_tree(C, f, qi, prog, plen, wq, root);
                                truth = true;
                                gt = tp.slots; // (one group of every slot: the bookkeeping below — term list, cost, output bound — sees a union)
                                gs.assign({0u, (uint32_t)gt.size()});
                                S.negs.clear();
                                S.leaves = tp.leaves;
                                S.leaf_tok = tp.leaf_tok;
                                S.qphrases.clear();
                                S.phterms.clear();
                        }
                        auto gcost = [&](uint32_t g) {
                                uint64_t c = 0;
                                for (uint32_t i = gs[g]; i < gs[g + 1]; ++i)
                                        c += ix.terms[gt[i]].documents;
                                return c;
                        };
                        auto &gorder = S.gorder;
                        gorder.resize(ngroups());
                        std::iota(gorder.begin(), gorder.end(), 0u);
                        if (ngroups() == 2) { // (the common case: a stable two-element sort)
                                if (gcost(1) < gcost(0))
                                        std::swap(gorder[0], gorder[1]);
                        } else if (ngroups() > 2)
                                std::stable_sort(gorder.begin(), gorder.end(), [&](uint32_t x, uint32_t y) { return gcost(x) < gcost(y); });
                        auto &uniq = S.uniq; // terms group by group, QT_GROUP on the first of each group
                        uniq.clear();
                        for (uint32_t g : gorder)
                                for (uint32_t i = gs[g]; i < gs[g + 1]; ++i)
                                        uniq.push_back(gt[i] | (i == gs[g] ? QT_GROUP : 0u));
                        {
                                // the excluded terms: one more group, the last, marked QT_NOT
                                auto &u = S.u;
                                u.clear();
                                for (uint32_t x : S.negs)
                                        if (ix.terms[x].documents && std::find(u.begin(), u.end(), x) == u.end())
                                                u.push_back(x);
                                for (size_t i = 0; i < u.size(); ++i)
                                        uniq.push_back(u[i] | (i == 0 ? (QT_GROUP | QT_NOT) : 0u));
                        }
                        if (uniq.size() > MAX_QTERMS) // a conjunctive normal form wider than the CNF kernels' term lists: the tree path
                                return hidden ? herr(f.err, TRI_ERR_INVALID, "query %zu: a phrase of more than %u terms", qi, MAX_QTERMS) : lower_tree(C, f, qi, prog, plen, wq, root);
                        // (default mode: the reportable terms — every postings iterator collect_doc_matching_terms can reach (queryexec_ctx.cpp:382-520):
                        //  group members and phrase terms, not the excluded side of a NOT —, distinct, in order of first appearance; counted before
                        //  anything of the query is recorded, so that a query with too many of them can still be left out cleanly)
                        auto &rt = S.rt;
                        rt.clear();
                        if (rich) {
                                for (uint32_t pi = 0; pi < plen; ++pi) {
                                        const uint32_t tok = prog[pi];
                                        if ((tok >> 28) != TRI_OP_TERM)
                                                continue;
                                        const uint32_t x = tok & 0x0fffffffu;
                                        const bool positive = std::find(S.leaves.begin(), S.leaves.end(), x) != S.leaves.end() ||
                                                              std::find(S.phterms.begin(), S.phterms.end(), x) != S.phterms.end();
                                        if (positive && std::find(rt.begin(), rt.end(), x) == rt.end())
                                                rt.push_back(x);
                                }
                                if (rt.size() > 16) {
                                        f.left_out.push_back(qi);
                                        herr(f.err, TRI_ERR_UNSUPPORTED, "query %zu: more than 16 reportable terms", qi);
                                        return TRI_OK;
                                }
                        }
                        const uint32_t g0 = gorder[0];
                        const uint32_t nlead = gs[g0 + 1] - gs[g0];
                        const uint64_t lead_docs = gcost(g0);
                        Tmp t{};
                        if (!S.qphrases.empty() && ix.codec == TRI_CODEC_LUCENE && !ix.has_hdir)
                                return herr(f.err, TRI_ERR_INVALID, "query %zu: phrase over a LUCENE segment that was uploaded without hits.data", qi);
                        t.q.phrase_base = (uint32_t)f.phrases.size();
                        t.q.nphrases = (uint32_t)S.qphrases.size();
                        f.phrase_queries += t.q.nphrases ? 1 : 0;
                        for (const auto &ph : S.qphrases) {
                                f.phrases.push_back({(uint32_t)f.pterms.size(), ph.n, ph.weight});
                                for (uint32_t k = 0; k < ph.n; ++k) {
                                        const uint32_t x = S.phterms[ph.t0 + k];
                                        f.pterms.push_back(x);
                                        f.term_bytes += ix.hitbytes[x]; // phrase queries also stream the hit bytes
                                        f.term_bytes_phrase_hits += ix.hitbytes[x];
                                }
                        }
                        t.q.score_base = (uint32_t)f.sterms.size();
                        t.q.nscore = 0;
                        if (rich) {
                                for (uint32_t x : rt) {
                                        f.sterms.push_back(x);
                                        f.term_bytes += ix.hitbytes[x]; // the hits of every reported term are read
                                }
                                t.q.nscore = (uint32_t)rt.size();
                                f.rich_R = std::max<uint32_t>(f.rich_R, t.q.nscore);
                        }
                        if (scored) {
                                // one scorer per PostingsListIterator of the conjunction, summed in iterator order
                                // (docset_iterators_scorers.cpp:173-193); weight = BM25 idf (similarity.h:179-181, float math)
                                // unless the caller supplied ScorerWeights per TERM token — the leaf's OWN token (a term that also sits
                                // inside a phrase or on an excluded side has another token with another weight)
                                for (size_t li = 0; li < S.leaves.size(); ++li) {
                                        f.sterms.push_back(S.leaves[li]);
                                        f.sweights.push_back(wq ? wq[S.leaf_tok[li]] : C.term_weight(ix.terms[S.leaves[li]].documents));
                                }
                                t.q.nscore = (uint32_t)S.leaves.size();
                        }
                        // ---- slot map for the one-pass scored path (k_fused.hpp): the query's distinct terms, CNF terms first
                        t.fz = -1;
                        t.truth = truth;
                        if (truth) {
                                DevFused z{};
                                z.nslots = (uint32_t)tp.slots.size();
                                z.hw = 0; // (general trees run in their own instantiation, 32-bit window words)
                                z.fbits = z.nslots <= 4 ? 8u : 4u;
                                z.cap = (1u << z.fbits) - 2u;
                                if (opt.fused_freq_cap && opt.fused_freq_cap < z.cap)
                                        z.cap = (uint32_t)opt.fused_freq_cap;
                                const uint32_t fm = (1u << z.fbits) - 1u;
                                for (size_t i = 0; i < tp.slots.size(); ++i)
                                        z.term[i] = tp.slots[i];
                                // DocumentsOnly, the default mode and the full score stream (topk == 0) need the docID set; top-K batches do not
                                z.mode = FUS_MODE_TT | ((scored && topk) ? 0u : FUS_MODE_EMIT);
                                memcpy(z.tt, tp.tt, sizeof z.tt);
                                if (rich) {
                                        // per REPORTABLE term (distinct, f.sterms order): reported where any of its leaves sits on the document
                                        z.nleaf = t.q.nscore;
                                        for (uint32_t j = 0; j < t.q.nscore; ++j) {
                                                const uint32_t term = f.sterms[t.q.score_base + j];
                                                for (size_t l = 0; l < tp.leaves.size(); ++l)
                                                        if (tp.leaves[l] == term) {
                                                                z.leaf_slot[j] = (uint8_t)tp.leaf_slot[l];
                                                                for (int wd = 0; wd < 8; ++wd)
                                                                        z.ctt[j][wd] |= tp.ctt[l][wd];
                                                        }
                                        }
                                        f.rich_allow = true;
                                } else {
                                        z.nleaf = (uint32_t)tp.leaves.size();
                                        for (size_t j = 0; j < tp.leaves.size(); ++j) {
                                                z.leaf_slot[j] = (uint8_t)tp.leaf_slot[j];
                                                memcpy(z.ctt[j], tp.ctt[j].data(), sizeof z.ctt[j]);
                                        }
                                }
                                // window skipping needs groups of slots one of which every match holds: the slots of the scorer leaves if no
                                // matching pattern lacks them all (else every slot: pattern 0 never matches), then every slot all matches hold
                                const uint32_t npat = 1u << z.nslots;
                                auto matches = [&](uint32_t p) { return (tp.tt[p >> 5] >> (p & 31u)) & 1u; };
                                uint32_t gl = 0;
                                for (uint32_t sl : tp.leaf_slot)
                                        gl |= 1u << sl;
                                for (uint32_t p = 0; p < npat; ++p)
                                        if (matches(p) && !(p & gl))
                                                gl = npat - 1;
                                auto add_req = [&](uint32_t gsl) {
                                        z.gslots[z.nreq] = gsl;
                                        for (uint32_t sl = 0; sl < z.nslots; ++sl)
                                                if ((gsl >> sl) & 1u)
                                                        z.gmask[z.nreq] |= fm << (sl * z.fbits);
                                        ++z.nreq;
                                };
                                add_req(gl);
                                for (uint32_t sl = 0; sl < z.nslots && z.nreq < FUS_MAX_SLOTS; ++sl) {
                                        bool all = gl != (1u << sl);
                                        for (uint32_t p = 0; p < npat && all; ++p)
                                                all = !matches(p) || ((p >> sl) & 1u);
                                        if (all)
                                                add_req(1u << sl);
                                }
                                t.fz = (int32_t)f.fz.size();
                                f.fz.push_back(z);
                        } else if (scored && topk && S.qphrases.empty() && opt.fused) {
                                auto &slots = S.slots;
                                slots.clear();
                                auto slot_of = [&](uint32_t term) {
                                        for (size_t i = 0; i < slots.size(); ++i)
                                                if (slots[i] == term)
                                                        return (uint32_t)i;
                                        slots.push_back(term);
                                        return (uint32_t)slots.size() - 1;
                                };
                                for (uint32_t tt : uniq)
                                        slot_of(tt & QT_TERM);
                                for (uint32_t x : S.leaves)
                                        slot_of(x);
                                if (slots.size() <= FUS_MAX_SLOTS) {
                                        DevFused z{};
                                        z.nslots = (uint32_t)slots.size();
                                        z.hw = (opt.fused_halfwords && z.nslots <= 5) ? 1u : 0u;
                                        z.fbits = z.hw ? std::min(8u, 16u / z.nslots) : (z.nslots <= 4 ? 8u : 4u);
                                        z.cap = (1u << z.fbits) - 2u;
                                        if (opt.fused_freq_cap && opt.fused_freq_cap < z.cap)
                                                z.cap = (uint32_t)opt.fused_freq_cap;
                                        const uint32_t fm = (1u << z.fbits) - 1u;
                                        for (size_t i = 0; i < slots.size(); ++i)
                                                z.term[i] = slots[i];
                                        int g = -1;
                                        bool in_not = false;
                                        uint32_t nreq_groups = 0;
                                        for (uint32_t tt : uniq)
                                                nreq_groups += (tt & QT_GROUP) && !(tt & QT_NOT);
                                        for (uint32_t tt : uniq) {
                                                if (nreq_groups > FUS_MAX_SLOTS)
                                                        break; // (a CNF that repeats its terms over more groups than the slot map holds)
                                                if (tt & QT_GROUP) {
                                                        in_not = tt & QT_NOT;
                                                        if (!in_not)
                                                                ++g;
                                                }
                                                const uint32_t sidx = slot_of(tt & QT_TERM);
                                                if (in_not)
                                                        z.nmask |= fm << (sidx * z.fbits);
                                                else {
                                                        z.gmask[g] |= fm << (sidx * z.fbits);
                                                        z.gslots[g] |= 1u << sidx;
                                                }
                                        }
                                        z.nreq = (uint32_t)(g + 1);
                                        if (z.nreq >= 1 && nreq_groups <= FUS_MAX_SLOTS) {
                                                t.fz = (int32_t)f.fz.size();
                                                f.fz.push_back(z);
                                        }
                                }
                        }
                        t.q.fused_idx = 0;
                        t.q.form = RESULT_DOCIDS;
                        t.q.nterms = (uint32_t)uniq.size();
                        t.q.term_base = (uint32_t)f.qterms.size();
                        t.q.out_cap = 0;
                        t.q.out_off = 0;
                        t.q.qid = (uint32_t)qi;
                        t.cost = 0;
                        t.nlead = nlead;
                        {
                                auto &seen = S.seen;
                                seen.clear();
                                for (uint32_t tt : uniq) {
                                        const uint32_t term = tt & QT_TERM;
                                        f.qterms.push_back(tt);
                                        if (std::find(seen.begin(), seen.end(), term) == seen.end()) {
                                                seen.push_back(term);
                                                f.term_bytes += ix.docbytes[term];
                                        }
                                }
                                // cost estimate: the lead group is decoded fully; every other list costs min(its blocks x 32, lead docs x 32)
                                for (size_t i = 0; i < uniq.size(); ++i) {
                                        const DevTerm &tk = ix.terms[uniq[i] & QT_TERM];
                                        t.cost += i < nlead ? tk.documents : 32ull * std::min<uint64_t>(tk.nblocks, lead_docs);
                                }
                        }
                        // ---- execution class.  TASK_DENSE (bitmap windows) when the lead group is an OR (it has to be materialised as a set
                        //      anyway), or when every other list is within a factor 32 of the lead (no block could be skipped) and there is
                        //      enough work per docID window to keep 256 lanes busy; one pass (TASK_FUSED / TASK_PLANES) when such a query
                        //      asks for a top-K, or is a general tree
                        {
                                t.sumdf = 0;
                                t.lead_docs = lead_docs;
                                t.last_doc = 0xffffffffu;
                                t.dense = uniq.size() >= 2;
                                uint32_t glast = 0;
                                bool in_neg = false;
                                for (size_t k = 0; k < uniq.size(); ++k) {
                                        const DevTerm &tk = ix.terms[uniq[k] & QT_TERM];
                                        t.sumdf += tk.documents;
                                        t.dense &= tk.nblocks <= lead_docs;
                                        if (k && (uniq[k] & QT_GROUP)) {
                                                t.last_doc = std::min(t.last_doc, glast);
                                                glast = 0;
                                                in_neg = uniq[k] & QT_NOT;
                                        }
                                        if (!in_neg)
                                                glast = std::max(glast, ix.blk_last[tk.first_block + tk.nblocks - 1]);
                                }
                                if (!in_neg)
                                        t.last_doc = std::min(t.last_doc, glast);
                                t.dense &= t.sumdf >= opt.dense_min_postings;
                                t.dense |= nlead > 1;
                                const bool fusable = t.fz >= 0;
                                t.fuse = truth || (t.dense && fusable && (opt.fused != 2 || f.fz[t.fz].nreq == 1)); // (fused == 2: only pure unions)
                                if (t.fuse) {
                                        ++f.onepass_queries;
                                        const DevFused &z = f.fz[t.fz];
                                        for (uint32_t sidx = 0; sidx < z.nslots; ++sidx)
                                                f.fused_postings += ix.terms[z.term[sidx]].documents;
                                }
                        }
                        t.hidden = hidden;
                        f.tmp.push_back(t);
                }
                return TRI_OK;
        }

        // ---- first pass: lower the queries [q_lo, q_hi) of the batch into `f` and class them
        inline int lower_range(const Ctx &C, Frag &f) {
                const PlanInput &in = C.in;
                // a query's lowering reads a handful of per-term records (directory entry, list bytes, rank by document count) at term ids drawn from
                // a vocabulary of millions: the pass is a chain of cache misses (cfg2: 280 ns per 2-term query on one thread).  The term ids stand in
                // the program's TERM tokens, so the records of the query AHEAD queries on are requested while this one is lowered
                constexpr size_t AHEAD = 6;
                const size_t nterms = C.ix.terms.size();
                auto prefetch_query = [&](const size_t q) {
                        const tri_query &t = in.queries[q];
                        if ((uint64_t)t.prog_off + t.prog_len > in.prog_len)
                                return;
                        for (uint32_t i = 0; i < t.prog_len && i < 16; ++i) {
                                const uint32_t tok = in.prog[t.prog_off + i];
                                const uint32_t x = tok & 0x0fffffffu;
                                if ((tok >> 28) == TRI_OP_TERM && x < nterms) {
                                        __builtin_prefetch(&C.ix.terms[x]);
                                        __builtin_prefetch(&C.ix.docbytes[x]);
                                        __builtin_prefetch(&C.ix.df_rank[x]);
                                }
                        }
                };
                // ... and, half as far ahead, what hangs off those records: the directory entry of the list's last block (the query's docID range)
                auto prefetch_tails = [&](const size_t q) {
                        const tri_query &t = in.queries[q];
                        if ((uint64_t)t.prog_off + t.prog_len > in.prog_len)
                                return;
                        for (uint32_t i = 0; i < t.prog_len && i < 16; ++i) {
                                const uint32_t tok = in.prog[t.prog_off + i];
                                const uint32_t x = tok & 0x0fffffffu;
                                if ((tok >> 28) == TRI_OP_TERM && x < nterms) {
                                        const DevTerm &tk = C.ix.terms[x];
                                        if (tk.nblocks)
                                                __builtin_prefetch(&C.ix.blk_last[tk.first_block + tk.nblocks - 1]);
                                }
                        }
                };
                for (size_t q = f.q_lo; q < std::min(f.q_hi, f.q_lo + AHEAD); ++q)
                        prefetch_query(q);
                for (size_t qi = f.q_lo; qi < f.q_hi; ++qi) {
                        if (qi + AHEAD < f.q_hi)
                                prefetch_query(qi + AHEAD);
                        if (qi + AHEAD / 2 < f.q_hi)
                                prefetch_tails(qi + AHEAD / 2);
                        const tri_query &tq = in.queries[qi];
                        if ((uint64_t)tq.prog_off + tq.prog_len > in.prog_len || !tq.prog_len)
                                return herr(f.err, TRI_ERR_INVALID, "query %zu: program slice out of range", qi);
                        if (const int rc = lower_query(C, f, qi, in.prog + tq.prog_off, tq.prog_len, in.weights ? in.weights + tq.prog_off : nullptr, false))
                                return rc;
                }
                return TRI_OK;
        }


        // ---- a query the CNF lowering and the truth table leave: the tree itself goes to the device (TASK_TREE, k_tree.hpp).  Every leaf
        //      becomes a bitmap over the docID space — a term's plane row (k_term_planes, once per run for every tree query of the batch that
        //      names it), a multi-word phrase's matches (a HIDDEN query of the same batch: the conjunction of its terms + the positional check,
        //      through the kernels every phrase query takes; its match list is scattered into the row) —, the inner nodes are word-wise algebra
        //      (DocsSetIterators::Conjuction / Disjunction / DisjunctionSome / Filter / Optional, docset_iterators.cpp:226-677, as set operations),
        //      and scores / reported terms follow the reference's recursion over the iterators that sit on a match
        //      (docset_iterators_scorers.cpp:38-228, queryexec_ctx.cpp:382-520) document by document.
        inline int lower_tree(const Ctx &C, Frag &f, const size_t qi, const uint32_t *prog, const uint32_t plen, const double *wq, const int root) {
                const HostIndex &ix = C.ix;
                Scratch &S = f.S;
                auto leave_out = [&](const char *why) {
                        f.left_out.push_back(qi);
                        herr(f.err, TRI_ERR_UNSUPPORTED, "query %zu: %s", qi, why);
                        return TRI_OK;
                };
                struct PhraseLeaf {
                        uint32_t node, t0, n, tok;
                };
                std::vector<DevTreeNode> tn;
                std::vector<PhraseLeaf> phl;
                std::vector<uint32_t> phterms, leaf_tok; // phrase leaves' terms; per node, the program token of a leaf
                std::vector<uint8_t> positive;           // per node: a leaf an iterator of the tree can report (not under an excluded side)
                bool ok = true;
                // postfix emission (children first); returns the node's index
                std::function<int(int, bool)> emit = [&](const int ni, const bool pos) -> int {
                        const PNode x = S.nodes[ni];
                        const int *kd = S.kids(x);
                        DevTreeNode d{};
                        d.parent = 0xff;
                        d.score = 0xffffffffu;
                        uint32_t tok = x.tok;
                        if (x.op == TRI_OP_TERM || (x.op == TRI_OP_PHRASE && x.kid_n == 1)) {
                                const PNode &t = x.op == TRI_OP_TERM ? x : S.nodes[kd[0]];
                                d.op = TRI_OP_TERM;
                                d.arg = t.term;
                                tok = t.tok;
                                if (t.term >= ix.terms.size() || !ix.terms[t.term].documents)
                                        ok = false; // (parse_program drops what can never match: not reached)
                        } else if (x.op == TRI_OP_PHRASE) {
                                d.op = TRI_OP_PHRASE;
                                phl.push_back({(uint32_t)tn.size(), (uint32_t)phterms.size(), x.kid_n, x.tok});
                                for (uint32_t k = 0; k < x.kid_n; ++k)
                                        phterms.push_back(S.nodes[kd[k]].term);
                        } else {
                                d.op = (uint8_t)x.op;
                                d.thr = x.op == TRI_OP_SOME ? (uint8_t)std::min<uint32_t>(x.term, 255) : 0;
                                std::vector<int> kids;
                                for (uint32_t k = 0; k < x.kid_n && ok; ++k)
                                        kids.push_back(emit(kd[k], pos && !(x.op == TRI_OP_NOT && k == 1)));
                                if (!ok || tn.size() + 1 > TREE_MAX_NODES)
                                        return ok = false, -1;
                                for (size_t k = 0; k < kids.size(); ++k) {
                                        d.kids |= 1ull << kids[k];
                                        tn[kids[k]].parent = (uint8_t)tn.size();
                                        tn[kids[k]].ord = (uint8_t)k;
                                }
                                if (x.op == TRI_OP_NOT || x.op == TRI_OP_OPT)
                                        d.kid0 = (uint8_t)kids[0], d.kid1 = (uint8_t)kids[1];
                        }
                        if (tn.size() + 1 > TREE_MAX_NODES)
                                return ok = false, -1;
                        tn.push_back(d);
                        leaf_tok.push_back(tok);
                        positive.push_back(pos && (d.op == TRI_OP_TERM || d.op == TRI_OP_PHRASE));
                        return (int)tn.size() - 1;
                };
                emit(root, true);
                if (!ok)
                        return leave_out("a tree of more than 64 nodes");
                const uint32_t nn = (uint32_t)tn.size();
                if (C.scored && std::find(positive.begin(), positive.end(), 1) == positive.end())
                        return leave_out("a tree without a scoring leaf");
                // the value of every node for a document that holds none of the leaves, and an upper bound of a node's matches
                uint64_t ub_root = 0;
                {
                        uint64_t val = 0;
                        std::vector<uint64_t> ub(nn, 0);
                        for (uint32_t i = 0; i < nn; ++i) {
                                const DevTreeNode &d = tn[i];
                                bool v = false;
                                uint64_t u = 0, sum = 0, mn = UINT64_MAX;
                                for (uint32_t k = 0; k < i; ++k)
                                        if ((d.kids >> k) & 1ull)
                                                sum += ub[k], mn = std::min(mn, ub[k]);
                                switch (d.op) {
                                        case TRI_OP_TERM:
                                                u = ix.terms[d.arg].documents;
                                                break;
                                        case TRI_OP_PHRASE:
                                                u = UINT64_MAX;
                                                for (const PhraseLeaf &p : phl)
                                                        if (p.node == i)
                                                                for (uint32_t k = 0; k < p.n; ++k)
                                                                        u = std::min<uint64_t>(u, ix.terms[phterms[p.t0 + k]].documents);
                                                break;
                                        case TRI_OP_AND:
                                                v = (val & d.kids) == d.kids;
                                                u = mn;
                                                break;
                                        case TRI_OP_OR:
                                                v = (val & d.kids) != 0;
                                                u = sum;
                                                break;
                                        case TRI_OP_SOME:
                                                v = (uint32_t)__builtin_popcountll(val & d.kids) >= d.thr;
                                                u = sum;
                                                break;
                                        case TRI_OP_NOT:
                                                v = ((val >> d.kid0) & 1ull) && !((val >> d.kid1) & 1ull);
                                                u = ub[d.kid0];
                                                break;
                                        case TRI_OP_OPT:
                                                v = (val >> d.kid0) & 1ull;
                                                u = ub[d.kid0];
                                                break;
                                }
                                val |= (uint64_t)v << i;
                                ub[i] = std::min<uint64_t>(u, ix.max_doc);
                        }
                        if ((val >> (nn - 1)) & 1ull)
                                return leave_out("a tree that matches documents holding none of its terms cannot be enumerated from postings");
                        if (ub[nn - 1] > 0xffffffffull)
                                return leave_out("a tree of more than 2^32 possible matches");
                        ub_root = ub[nn - 1];
                }
                // the reportable terms (default mode): what the positive leaves' iterators are, distinct, in order of first appearance in the program
                std::vector<uint32_t> rt;
                if (C.rich) {
                        auto is_pos = [&](uint32_t term) {
                                for (uint32_t i = 0; i < nn; ++i)
                                        if (positive[i] && tn[i].op == TRI_OP_TERM && tn[i].arg == term)
                                                return true;
                                for (const PhraseLeaf &p : phl)
                                        if (positive[p.node])
                                                for (uint32_t k = 0; k < p.n; ++k)
                                                        if (phterms[p.t0 + k] == term)
                                                                return true;
                                return false;
                        };
                        for (uint32_t pi = 0; pi < plen; ++pi) {
                                if ((prog[pi] >> 28) != TRI_OP_TERM)
                                        continue;
                                const uint32_t x = prog[pi] & 0x0fffffffu;
                                if (std::find(rt.begin(), rt.end(), x) == rt.end() && is_pos(x))
                                        rt.push_back(x);
                        }
                        if (rt.size() > 16)
                                return leave_out("more than 16 reportable terms");
                        auto bit_of = [&](uint32_t term) { return 1u << (uint32_t)(std::find(rt.begin(), rt.end(), term) - rt.begin()); };
                        for (uint32_t i = 0; i < nn; ++i)
                                if (positive[i] && tn[i].op == TRI_OP_TERM)
                                        tn[i].rmask = bit_of(tn[i].arg);
                        for (const PhraseLeaf &p : phl)
                                if (positive[p.node])
                                        for (uint32_t k = 0; k < p.n; ++k)
                                                tn[p.node].rmask |= bit_of(phterms[p.t0 + k]);
                }
                if (!phl.empty() && ix.codec == TRI_CODEC_LUCENE && !ix.has_hdir)
                        return herr(f.err, TRI_ERR_INVALID, "query %zu: phrase over a LUCENE segment that was uploaded without hits.data", qi);
                // ---- the phrase leaves: one hidden query each (S is reused by their lowering: nothing of this query's parse is read below)
                std::vector<double> pweight(phl.size(), 0.0);
                for (size_t pi = 0; pi < phl.size(); ++pi) {
                        const PhraseLeaf &p = phl[pi];
                        std::vector<uint32_t> hp;
                        std::vector<double> hw;
                        for (uint32_t k = 0; k < p.n; ++k)
                                hp.push_back((TRI_OP_TERM << 28) | phterms[p.t0 + k]);
                        hp.push_back((TRI_OP_PHRASE << 28) | p.n);
                        if (wq) {
                                hw.assign(p.n + 1, 0.0);
                                hw[p.n] = wq[p.tok];
                        }
                        const size_t before = f.tmp.size(), lo_before = f.left_out.size();
                        if (const int rc = lower_query(C, f, qi, hp.data(), (uint32_t)hp.size(), wq ? hw.data() : nullptr, true))
                                return rc;
                        if (f.tmp.size() != before + 1 || f.left_out.size() != lo_before) {
                                f.left_out.resize(lo_before);
                                return leave_out("a phrase leaf the planner does not lower");
                        }
                        Tmp &h = f.tmp.back();
                        h.hidden_ord = f.n_hidden++;
                        tn[p.node].arg = (uint32_t)before;                 // (fragment-relative plan slot: rebased in the fill pass)
                        tn[p.node].row = TREE_ROW_PHRASE | h.hidden_ord;   // (likewise)
                        pweight[pi] = f.phrases[h.q.phrase_base].weight;
                }
                // ---- the query itself
                Tmp t{};
                t.tree = true;
                t.tree_ub = ub_root;
                t.fz = -1;
                t.q.qid = (uint32_t)qi;
                t.q.term_base = (uint32_t)f.qterms.size();
                t.q.phrase_base = (uint32_t)f.phrases.size();
                t.q.score_base = (uint32_t)f.sterms.size();
                t.cost = ub_root;
                std::vector<uint32_t> seen;
                auto once = [&](uint32_t term) {
                        if (std::find(seen.begin(), seen.end(), term) != seen.end())
                                return false;
                        seen.push_back(term);
                        return true;
                };
                for (uint32_t i = 0; i < nn; ++i)
                        if (tn[i].op == TRI_OP_TERM) {
                                f.tree_terms.push_back(tn[i].arg);
                                if (once(tn[i].arg))
                                        f.term_bytes += ix.docbytes[tn[i].arg];
                        }
                if (C.rich) {
                        for (uint32_t x : rt) {
                                f.sterms.push_back(x);
                                f.term_bytes += ix.hitbytes[x];
                        }
                        t.q.nscore = (uint32_t)rt.size();
                        f.rich_R = std::max<uint32_t>(f.rich_R, t.q.nscore);
                        f.rich_allow = true;
                } else if (C.scored) {
                        // one scorer per positive leaf, summed in tree order (docset_iterators_scorers.cpp:38-228)
                        for (uint32_t i = 0; i < nn; ++i) {
                                if (!positive[i])
                                        continue;
                                tn[i].score = t.q.nscore++;
                                if (tn[i].op == TRI_OP_TERM) {
                                        f.sterms.push_back(tn[i].arg);
                                        f.sweights.push_back(wq ? wq[leaf_tok[i]] : C.term_weight(ix.terms[tn[i].arg].documents));
                                } else { // (a phrase leaf's score comes with its hidden query's matches — k_phrase; the slot keeps the arrays parallel)
                                        size_t pi = 0;
                                        while (phl[pi].node != i)
                                                ++pi;
                                        f.sterms.push_back(phterms[phl[pi].t0]);
                                        f.sweights.push_back(pweight[pi]);
                                }
                        }
                }
                t.q.fused_idx = (uint32_t)f.treepool.size();
                f.treepool.resize(f.treepool.size() + TREE_HDR_WORDS + nn * (sizeof(DevTreeNode) / 4), 0u);
                f.treepool[t.q.fused_idx] = nn;
                memcpy(&f.treepool[t.q.fused_idx + TREE_HDR_WORDS], tn.data(), nn * sizeof(DevTreeNode));
                f.tmp.push_back(t);
                return TRI_OK;
        }

        // ---- second pass: cut the fragment's queries into tasks (offsets relative to the fragment)
        inline int task_range(const Ctx &C, Frag &f) {
                const HostIndex &ix = C.ix;
                const tri_options &opt = C.env.opt;
                const uint64_t planes_opt = opt.planes;
                const uint64_t DENSE_TASK_COST = std::max<uint64_t>(1, opt.dense_task_cost); // bitmap-window tasks stage their terms once: two windows of a head pair per task
                const uint64_t PLANES_SPLIT = C.planes_split, FUSED_TASK_COST = C.fused_task_cost;
                f.benefit.assign(C.n_ok, 0);
                f.cand_row.assign(C.n_ok, 0);
                uint64_t off = 0;
                for (size_t ti = 0; ti < f.tmp.size(); ++ti) {
                        if (ti + 6 < f.tmp.size()) { // (the per-term records of the query six queries on: see lower_range)
                                const Tmp &a = f.tmp[ti + 6];
                                for (uint32_t k = 0; k < a.q.nterms && k < 8 && !a.tree; ++k) {
                                        const uint32_t x = f.qterms[a.q.term_base + k] & QT_TERM;
                                        __builtin_prefetch(&ix.terms[x]);
                                        __builtin_prefetch(&ix.docbytes[x]);
                                        __builtin_prefetch(&ix.df_rank[x]);
                                }
                        }
                        Tmp &t = f.tmp[ti];
                        const uint32_t slot = (uint32_t)ti;
                        if (t.tree) { // one task: its chunks of the docID space are the kernels' grid, its region the bound of the tree's matches
                                t.q.out_off = off;
                                t.q.out_cap = (uint32_t)t.tree_ub;
                                t.q.first_task = (uint32_t)f.tasks.size();
                                t.q.ntasks = 1;
                                f.tcost.push_back(std::max<uint64_t>(1, t.tree_ub));
                                f.tasks.push_back({slot, 0, (C.plw + TREE_CHUNK_WORDS - 1) / TREE_CHUNK_WORDS, TASK_TREE, off});
                                off += t.q.out_cap;
                                ++f.tree_queries;
                                continue;
                        }
                        const uint32_t *qt = &f.qterms[t.q.term_base];
                        const DevTerm &lead = ix.terms[qt[0] & QT_TERM];
                        const uint32_t nlead = t.nlead;
                        const uint32_t last_doc = t.last_doc;
                        if (t.fuse) {
                                // a CNF query whose top-K runs over bit planes (k_planes): its head terms read from the batch's term planes, the
                                // others (at most PLK_MAX_SPARSE) decoded per window into LDS planes
                                DevFused z = f.fz[t.fz];
                                uint32_t nsparse = 0;
                                for (uint32_t sidx = 0; sidx < z.nslots; ++sidx) {
                                        z.plane[sidx] = PL_NONE;
                                        nsparse += C.plane_ok(z.term[sidx]) ? 0u : 1u;
                                }
                                const bool pk = !t.truth && (planes_opt & 4u) && nsparse <= PLK_MAX_SPARSE && ix.max_doc < 0x7fff0000u; // (list entries are docID << 1 | flag)
                                {
                                        const uint32_t fm = (1u << z.fbits) - 1u;
                                        z.negslots = 0;
                                        for (uint32_t sidx = 0; sidx < z.nslots; ++sidx)
                                                if ((z.nmask >> (sidx * z.fbits)) & fm)
                                                        z.negslots |= 1u << sidx;
                                }
                                // every list of the slot map is read once (the optional terms too)
                                uint64_t slotdf = 0;
                                for (uint32_t sidx = 0; sidx < z.nslots; ++sidx) {
                                        slotdf += ix.terms[z.term[sidx]].documents;
                                        (pk ? f.term_bytes_planes : f.term_bytes_fused) += ix.docbytes[z.term[sidx]];
                                        if (pk && C.plane_ok(z.term[sidx])) {
                                                f.benefit[ix.df_rank[z.term[sidx]]] += ix.terms[z.term[sidx]].documents;
                                                f.fuses.push_back({(uint32_t)f.fused.size(), sidx, z.term[sidx]});
                                        }
                                }
                                ++(pk ? f.planes_queries : f.fused_queries);
                                t.q.fused_idx = (uint32_t)f.fused.size();
                                t.q.out_off = off;
                                t.q.out_cap = 0; // the docID set is never materialised ...
                                t.q.first_task = (uint32_t)f.tasks.size();
                                const uint32_t fw = pk ? PL_W : FUS_W << z.hw; // documents per window: plane windows, or this query's word width
                                const uint32_t nwin = last_doc / fw + 1;
                                const uint64_t per_win = std::max<uint64_t>(1, slotdf / (ix.info.docs_cnt / fw + 1));
                                // (k_planes' cost is the sweep of the range plus its candidates, not the postings: equal ranges, a few per query)
                                const uint32_t win_per_task = pk && PLANES_SPLIT < 65536 ? (uint32_t)((nwin + PLANES_SPLIT - 1) / PLANES_SPLIT)
                                                                                         : (uint32_t)std::max<uint64_t>(1, FUSED_TASK_COST / per_win);
                                const bool emit = z.mode & FUS_MODE_EMIT; // ... except by a general tree in DocumentsOnly mode: a private region per task,
                                                                          // bounded like TASK_DENSE's by the slots' blocks that reach the task's windows
                                uint32_t ord = 0;
                                for (uint32_t wb = 0; wb < nwin; wb += win_per_task, ++ord) {
                                        const uint32_t we = std::min(nwin, wb + win_per_task);
                                        uint64_t b1 = 0;
                                        if (emit)
                                                for (uint32_t sidx = 0; sidx < z.nslots; ++sidx)
                                                        b1 += C.first_block_ge(ix.terms[z.term[sidx]], (uint64_t)wb * fw);
                                        uint64_t entries = 0;
                                        if (pk) { // the rows of the decoded slots that can reach the task's docID range: 32 list entries each (k_planes)
                                                for (uint32_t sidx = 0; sidx < z.nslots; ++sidx) {
                                                        if (C.plane_ok(z.term[sidx]))
                                                                continue;
                                                        const DevTerm &tk = ix.terms[z.term[sidx]];
                                                        const uint32_t r0 = C.first_block_ge(tk, (uint64_t)wb * fw);
                                                        const uint32_t r1 = C.first_block_ge(tk, (uint64_t)we * fw);
                                                        if (r0 < tk.nblocks)
                                                                entries += 32ull * (std::min(r1, tk.nblocks - 1) - r0 + 1);
                                                }
                                                if (entries > 0x7fffffffull)
                                                        return herr(f.err, TRI_ERR_UNSUPPORTED, "query %u: a task's decoded lists exceed 2^31 entries", t.q.qid);
                                                f.sparse_cap = std::max(f.sparse_cap, (uint32_t)entries);
                                        }
                                        // (largest first, by postings: for k_planes a poor estimate — its cost is the sweep plus the candidates — but ordering by the
                                        //  decoded entries instead measured worse: cfg3's unions 10.6 ms against 9.2)
                                        f.tcost.push_back(per_win * (we - wb));
                                        f.tasks.push_back({slot, wb, we, pk ? (z.nslots <= PLK_NS_SMALL ? TASK_PLANES : TASK_PLANES8) : z.mode ? TASK_FUSED_GEN : z.hw ? TASK_FUSED16 : TASK_FUSED,
                                                           off + (emit ? b1 * 32 + 32ull * ord * z.nslots : 0)});
                                }
                                if (emit) {
                                        uint64_t blocks = 0;
                                        for (uint32_t sidx = 0; sidx < z.nslots; ++sidx)
                                                blocks += ix.terms[z.term[sidx]].nblocks;
                                        t.q.out_cap = (uint32_t)std::min<uint64_t>(0xffffffffull, blocks * 32 + 32ull * (ord + 1) * z.nslots);
                                        off += t.q.out_cap;
                                }
                                t.q.ntasks = (uint32_t)f.tasks.size() - t.q.first_task;
                                f.fused.push_back(z);
                                continue;
                        }
                        // every term of a bitmap-window query has a plane (a use as a window operand repays the decode by itself: such a term is
                        // always chosen): the query's windows are word-wise algebra over the planes — its own kernel (k_psets.hpp)
                        bool probe = false;
                        bool pset = t.dense && (planes_opt & 2u);
                        for (uint32_t k = 0; pset && k < t.q.nterms; ++k)
                                pset = C.plane_ok(qt[k] & QT_TERM);
                        // ... and a DocumentsOnly UNION (one group, nothing excluded) of head terms AND others whose result is a bitmap anyway (the head terms alone match
                        // one document in 32 or more): the head terms' plane words are OR-ed and stored like any other k_psets window, the other terms' few
                        // documents are then set in the stored words one by one (PSET_UNIT_SCATTER) — where k_and_dense decodes every list into an LDS window
                        // bitmap behind half a dozen barriers per window (cfg5's 5-way unions: 2.5 of the shard's 8.9 ms)
                        bool pscatter = false;
                        if (t.dense && !pset && (planes_opt & 2u) && C.mode == TRI_FLAG_DOCUMENTS_ONLY && opt.result_bitmaps && !t.q.nphrases && t.q.nterms >= 2) {
                                uint64_t plane_df = 0;
                                bool one_group = true;
                                for (uint32_t k = 0; k < t.q.nterms; ++k) {
                                        one_group = one_group && !(qt[k] & QT_NOT) && (k == 0) == ((qt[k] & QT_GROUP) != 0);
                                        if (C.plane_ok(qt[k] & QT_TERM))
                                                plane_df += ix.terms[qt[k] & QT_TERM].documents;
                                }
                                const uint32_t nwin = last_doc / SPAN_BITS + 1;
                                // (the result's form is decided below the same way on ALL the terms — min(N, sum of their documents) against the bitmap's words —: what
                                //  holds for the head terms alone holds for all of them, so a scatter union's result IS a bitmap)
                                const double N = std::max<double>(1.0, (double)ix.info.docs_cnt);
                                // (option scatter_bitmap_slack: such a union is run this way — and its result kept as a bitmap — from 1 / (32 x slack) of the documents on)
                                pscatter = one_group && plane_df && std::min<double>(N, (double)plane_df) * (double)std::max<uint64_t>(1, opt.scatter_bitmap_slack) >= (double)nwin * SPAN_WORDS;
                                pset = pscatter;
                        }
                        // a single lead list too short for a plane against lists that all have one: candidate tiles, every candidate tested with one
                        // bit probe per list (k_and) — the bitmap kernel would decode the lead into an LDS window bitmap and expand the window
                        // workgroup-wide for a handful of matches per window (cfg2: 253 such queries took 0.57 ms there, a third of the dense class's time)
                        if (t.dense && !pset && nlead == 1 && (planes_opt & 1u) && !C.plane_ok(qt[0] & QT_TERM) && t.q.nterms >= 2) {
                                bool probes = true;
                                for (uint32_t k = 1; probes && k < t.q.nterms; ++k)
                                        probes = C.plane_ok(qt[k] & QT_TERM);
                                if (probes)
                                        t.dense = false;
                        }
                        if (t.dense) {
                                auto &seen = f.S.seen;
                                seen.clear();
                                for (uint32_t k = 0; k < t.q.nterms; ++k) {
                                        const uint32_t term = qt[k] & QT_TERM;
                                        if (std::find(seen.begin(), seen.end(), term) == seen.end()) {
                                                seen.push_back(term);
                                                (pset ? f.term_bytes_pset : f.term_bytes_dense) += ix.docbytes[term];
                                        }
                                        if ((planes_opt & 2u) && C.plane_ok(term)) {
                                                f.benefit[ix.df_rank[term]] += ix.terms[term].documents;
                                                f.quses.push_back({t.q.term_base + k, term});
                                        }
                                }
                                ++(pset ? f.pset_queries : f.dense_queries);
                        } else {
                                // one lead list against lists that are all long enough for a plane: should the batch's uses repay every one of those
                                // planes (settled once the whole batch is known: the fill pass), the task runs in k_probe, else as candidate tiles
                                probe = nlead == 1 && t.q.nterms >= 2 && t.q.nphrases == 0 && (planes_opt & 1u) && lead.nblocks <= opt.probe_max_blocks;
                                for (uint32_t k = 1; probe && k < t.q.nterms; ++k)
                                        probe = C.plane_ok(qt[k] & QT_TERM);
                                ++(probe ? f.probe_queries : f.cand_queries);
                                f.cand_lead_docs += lead.documents, f.cand_terms += t.q.nterms;
                                if (probe) {
                                        auto &seen = f.S.seen;
                                        seen.clear();
                                        for (uint32_t k = 0; k < t.q.nterms; ++k)
                                                if (std::find(seen.begin(), seen.end(), qt[k] & QT_TERM) == seen.end()) {
                                                        seen.push_back(qt[k] & QT_TERM);
                                                        f.term_bytes_probe += ix.docbytes[qt[k] & QT_TERM];
                                                }
                                }
                                bool first_row = !probe;
                                for (uint32_t k = 1; k < t.q.nterms; ++k) { // (the lead list is decoded into the candidate tiles; the others are probed)
                                        const uint32_t term = qt[k] & QT_TERM;
                                        if ((planes_opt & 1u) && C.plane_ok(term)) {
                                                f.benefit[ix.df_rank[term]] += std::min<uint64_t>(ix.terms[term].documents, 32ull * lead.documents);
                                                f.quses.push_back({t.q.term_base + k, term});
                                                if (first_row) { // (k_and's queues: what the row's tasks will weigh)
                                                        const uint32_t ntiles = (lead.nblocks + TILE_BLOCKS - 1) / TILE_BLOCKS;
                                                        f.cand_row[ix.df_rank[term]] += ntiles + (ntiles + CAND_HEAVY_TILES - 1) / CAND_HEAVY_TILES;
                                                        first_row = false;
                                                }
                                        }
                                }
                        }
                        if (C.scored && (planes_opt & 1u)) // k_score: a scorer whose term has a plane reads the match's frequency off the planes
                                for (uint32_t k = 0; k < t.q.nscore; ++k) {
                                        const uint32_t term = f.sterms[t.q.score_base + k];
                                        if (C.plane_ok(term)) {
                                                f.benefit[ix.df_rank[term]] += std::min<uint64_t>(ix.terms[term].documents, 32ull * t.lead_docs);
                                                f.suses.push_back({t.q.score_base + k, term});
                                        }
                                }
                        t.q.out_off = off;
                        t.q.first_task = (uint32_t)f.tasks.size();
                        if (t.dense) {
                                const uint32_t nwin = last_doc / SPAN_BITS + 1;
                                // the result's form: a bitmap over the query's docID range when the matches to expect — the lead group's documents, thinned
                                // by every further required group as if the lists were independent — outnumber the bitmap's words
                                bool bitmap = false;
                                double est = 1.0; // the share of the documents expected to match
                                if (pset || (C.mode == TRI_FLAG_DOCUMENTS_ONLY && opt.result_bitmaps && !t.q.nphrases)) {
                                        const double N = std::max<double>(1.0, (double)ix.info.docs_cnt);
                                        double g = 0.0;
                                        bool negg = false;
                                        for (uint32_t k = 0; k <= t.q.nterms; ++k) {
                                                if (k == t.q.nterms || (k && (qt[k] & QT_GROUP))) {
                                                        if (!negg)
                                                                est *= std::min(1.0, g / N);
                                                        g = 0.0;
                                                }
                                                if (k == t.q.nterms)
                                                        break;
                                                if (qt[k] & QT_GROUP)
                                                        negg = qt[k] & QT_NOT;
                                                g += ix.terms[qt[k] & QT_TERM].documents;
                                        }
                                        bitmap = C.mode == TRI_FLAG_DOCUMENTS_ONLY && opt.result_bitmaps && !t.q.nphrases && (pscatter || est * N >= (double)nwin * SPAN_WORDS);
                                }
                                // (TASK_PSET) windows per ROUND of k_psets: a wave stages the survivors of its share of a round — a sub-window of PSET_ROUND_DOCS documents per
                                // window — in PSET_STAGE_DOCS LDS slots before the round's counts cross; as many windows as are expected to fill three quarters of them
                                uint32_t round_win = 1;
                                while (round_win < PSET_TASK_WINDOWS && est * (double)PSET_ROUND_DOCS * (double)(round_win * 2u) <= 0.75 * (double)PSET_STAGE_DOCS)
                                        round_win *= 2u;
                                if (pscatter && !bitmap)
                                        return herr(f.err, TRI_ERR_INTERNAL, "query %u: a scatter union whose result is not a bitmap", t.q.qid);
                                t.q.form = bitmap ? RESULT_BITMAP : RESULT_DOCIDS;
                                const uint64_t per_win = std::max<uint64_t>(1, t.sumdf / (ix.info.docs_cnt / SPAN_BITS + 1)) + (pset ? 0 : opt.dense_window_cost);
                                // (a query with phrases: its tasks are k_phrase's too, where a candidate costs a walk into two or three lists' hits — tens of times a bitmap
                                //  word; a task of four windows of two head terms was 76 K candidates, 1 - 2 ms, and k_phrase's span is its longest task: option phrase_task_div)
                                const uint32_t pdiv = t.q.nphrases ? (uint32_t)std::max<uint64_t>(1, C.phrase_task_div) : 1u;
                                const uint32_t win_per_task = pset ? std::max(1u, PSET_TASK_WINDOWS / pdiv) : (uint32_t)std::max<uint64_t>(1, DENSE_TASK_COST / pdiv / per_win);
                                uint32_t ord = 0;
                                uint64_t lead_blocks = 0;
                                for (uint32_t k = 0; k < nlead; ++k)
                                        lead_blocks += ix.terms[qt[k] & QT_TERM].nblocks;
                                uint64_t heaviest = 0xffffffffull; // (TASK_PSET) the df rank of the query's heaviest term: the schedule's place within a window range
                                for (uint32_t k = 0; pset && k < t.q.nterms; ++k)
                                        heaviest = std::min<uint64_t>(heaviest, ix.df_rank[qt[k] & QT_TERM]);
                                for (uint32_t wb = 0; wb < nwin; wb += win_per_task, ++ord) {
                                        const uint32_t we = std::min(nwin, wb + win_per_task);
                                        // matches of windows [wb, we) are lead-group documents of blocks b1 .. (next task's b1) of every
                                        // lead list: a private region (+32 slots of slack per lead list and task for the straddling block)
                                        uint64_t b1 = 0;
                                        for (uint32_t k = 0; k < nlead; ++k)
                                                b1 += C.first_block_ge(ix.terms[qt[k] & QT_TERM], (uint64_t)wb * SPAN_BITS);
                                        f.tcost.push_back(pset ? wb | heaviest << 32 : per_win * (we - wb)); // (TASK_PSET: the schedule goes by window range, not by cost)
                                        const uint64_t task_off = bitmap ? off + (uint64_t)wb * SPAN_WORDS : off + b1 * 32 + 32ull * ord * nlead;
                                        if (pset) {
                                                DevPsetUnit u{};
                                                u.out_off = task_off;
                                                u.first = (bitmap ? PSET_UNIT_BITMAP : 0u) | (pscatter ? PSET_UNIT_SCATTER : 0u) | round_win << PSET_UNIT_ROUND_SHIFT;
                                                u.w_begin = wb, u.w_end = we;
                                                u.tix = (uint32_t)f.tasks.size();
                                                u.nterms = t.q.nterms;
                                                u.term_base = t.q.term_base;
                                                for (uint32_t k = 0; k < t.q.nterms && k < PSET_INLINE_TERMS; ++k)
                                                        u.tt[k] = qt[k];
                                                f.units.push_back(u);
                                        }
                                        f.tasks.push_back({slot, wb, we, pset ? TASK_PSET : TASK_DENSE, task_off});
                                }
                                t.q.out_cap = bitmap ? nwin * SPAN_WORDS : (uint32_t)std::min<uint64_t>(0xffffffffull, lead_blocks * 32 + 32ull * (ord + 1) * nlead);
                                f.bitmap_queries += bitmap;
                                f.pscatter_queries += pscatter;
                        } else {
                                const uint32_t ntiles = (lead.nblocks + TILE_BLOCKS - 1) / TILE_BLOCKS;
                                const uint64_t per_tile = std::max<uint64_t>(1, t.cost / ntiles);
                                const uint32_t tiles_per_task = (uint32_t)std::max<uint64_t>(1, std::max<uint64_t>(1, opt.cand_task_cost / (t.q.nphrases ? std::max<uint64_t>(1, C.phrase_task_div) : 1)) / per_tile);
                                for (uint32_t tb = 0; tb < ntiles; tb += tiles_per_task) {
                                        const uint32_t te = std::min(ntiles, tb + tiles_per_task);
                                        f.tcost.push_back(per_tile * (te - tb));
                                        if (probe) {
                                                DevPsetUnit u{};
                                                u.out_off = off + (uint64_t)tb * TILE_CANDS;
                                                u.w_begin = tb, u.w_end = te;
                                                u.tix = (uint32_t)f.tasks.size();
                                                u.nterms = t.q.nterms;
                                                u.term_base = t.q.term_base;
                                                u.first = tb == 0 ? PSET_UNIT_FIRST : 0u;
                                                for (uint32_t k = 0; k < t.q.nterms && k < PSET_INLINE_TERMS; ++k)
                                                        u.tt[k] = qt[k];
                                                f.units.push_back(u);
                                        }
                                        f.tasks.push_back({slot, tb, te, probe ? TASK_PROBE : TASK_CAND, off + (uint64_t)tb * TILE_CANDS});
                                }
                                t.q.out_cap = lead.documents; // |A ∩ …| <= df of the lead
                                if (opt.account_needed_bytes) {
                                        // what a perfect gallop must read: the lead list, and of every other list the blocks that can hold a lead
                                        // candidate — per lead block the other list's blocks its docID range meets, at most one per candidate
                                        // (directories only; a block counts docbytes / nblocks)
                                        uint64_t need = ix.docbytes[qt[0] & QT_TERM];
                                        const uint32_t *ll = &ix.blk_last[lead.first_block];
                                        for (uint32_t k = 1; k < t.q.nterms; ++k) {
                                                const DevTerm &tk = ix.terms[qt[k] & QT_TERM];
                                                const uint32_t *ol = &ix.blk_last[tk.first_block];
                                                uint64_t blocks = 0;
                                                uint32_t at = 0; // (both directories ascend: the searches move forward)
                                                for (uint32_t lb = 0; lb < lead.nblocks && at < tk.nblocks; ++lb) {
                                                        const uint32_t lo_doc = lb ? ll[lb - 1] + 1 : 1u, hi_doc = ll[lb];
                                                        at = (uint32_t)(std::lower_bound(ol + at, ol + tk.nblocks, lo_doc) - ol);
                                                        if (at >= tk.nblocks)
                                                                break;
                                                        const uint32_t last = (uint32_t)(std::lower_bound(ol + at, ol + tk.nblocks, hi_doc) - ol);
                                                        const uint32_t span = std::min(last, tk.nblocks - 1) - at + 1;
                                                        const uint32_t ndocs = lb + 1 == lead.nblocks ? lead.last_n : 32u;
                                                        blocks += std::min(span, ndocs);
                                                }
                                                need += (uint64_t)((double)ix.docbytes[qt[k] & QT_TERM] * std::min(1.0, (double)blocks / std::max(1u, tk.nblocks)));
                                        }
                                        f.cand_needed += need;
                                }
                        }
                        off += t.q.out_cap;
                        t.q.ntasks = (uint32_t)f.tasks.size() - t.q.first_task;
                        if (t.q.nphrases)
                                for (uint32_t k = t.q.first_task; k < t.q.first_task + t.q.ntasks; ++k)
                                        f.ptasks.push_back(k);
                }
                f.off = off;
                return TRI_OK;
        }

        inline double ms_since(std::chrono::steady_clock::time_point &t0) {
                const auto now = std::chrono::steady_clock::now();
                const double ms = std::chrono::duration<double, std::milli>(now - t0).count();
                t0 = now;
                return ms;
        }
} // namespace trip

// Plan a batch.  `alloc_block(bytes)` provides the host block the plan's arrays are laid out in (pinned memory when a device will copy
// it; it must stay valid as long as the plan is used, and is 64-byte aligned); `pool` may be null (everything on the calling thread).
// Returns TRI_OK, or an error code with its text in `err` (a query shape the planner does not lower is NOT an error: BatchPlan::qstatus).
inline int plan_batch(const HostIndex &ix, const PlanEnv &env, const PlanInput &in, HostPool *pool, const std::function<uint8_t *(size_t)> &alloc_block,
                      BatchPlan &P, std::string &err, trip::FragCache *frag_cache = nullptr) {
        using namespace trip;
        auto t0 = std::chrono::steady_clock::now();
        const uint32_t mode = in.flags & (TRI_FLAG_DOCUMENTS_ONLY | TRI_FLAG_ACCUMULATED_SCORE | TRI_FLAG_MATCHED_TERMS);
        Ctx C{ix, env, in, mode == TRI_FLAG_ACCUMULATED_SCORE, mode == TRI_FLAG_MATCHED_TERMS, mode};
        const tri_options &opt = env.opt;
        const size_t nq = in.nq;
        P.slot_of_query.assign(nq, UINT32_MAX);
        P.qstatus.assign(nq, TRI_OK);
        // ---- which terms may get a plane: an indexed list of at least docs_cnt / plane_div documents, the longest lists first up to the
        //      scratch budget (a plane row is PL_PLANES bitmaps over the docID space)
        P.plw = ((ix.max_doc >> 17) + 2u) * (SPAN_BITS / 32u); // whole bitmap windows (k_and_dense reads SPAN_WORDS at a time) + a spare one
        C.plw = P.plw;
        if (opt.planes && opt.plane_div && !ix.df_sorted.empty()) {
                const uint64_t min_df = std::max<uint64_t>(1, ix.info.docs_cnt / opt.plane_div);
                // df_sorted descends: the first rank whose list is too short
                const size_t n = (size_t)(std::partition_point(ix.df_sorted.begin(), ix.df_sorted.end(), [&](uint32_t d) { return d && d >= min_df; }) - ix.df_sorted.begin());
                const uint64_t row_bytes = (uint64_t)PL_PLANES * P.plw * 4;
                C.n_ok = (uint32_t)std::min<uint64_t>(n, std::max<uint64_t>(1, opt.plane_max_bytes / std::max<uint64_t>(1, row_bytes)));
        }
        // ---- fragments: contiguous ranges of the batch, a couple per thread (dealt out dynamically)
        const unsigned nthreads = pool ? std::min<unsigned>(pool->size(), (unsigned)std::max<size_t>(1, nq / 512)) : 1u;
        const size_t nfrag = nthreads <= 1 ? 1 : std::min<size_t>(2 * nthreads, std::max<size_t>(1, nq / 256));
        std::vector<Frag> frags(nfrag);
        struct GiveBack { // (every way out of this function hands the fragments' buffers back to the caller's cache)
                std::vector<Frag> &frags;
                trip::FragCache *cache;
                ~GiveBack() {
                        if (!cache)
                                return;
                        if (cache->frags.size() < frags.size())
                                cache->frags.resize(frags.size());
                        for (size_t k = 0; k < frags.size(); ++k)
                                cache->frags[k] = std::move(frags[k]);
                }
        } give_back{frags, frag_cache};
        for (size_t k = 0; k < nfrag; ++k) {
                if (frag_cache && k < frag_cache->frags.size()) {
                        frags[k] = std::move(frag_cache->frags[k]);
                        frags[k].recycle();
                }
                frags[k].q_lo = nq * k / nfrag;
                frags[k].q_hi = nq * (k + 1) / nfrag;
        }
        auto run = [&](const std::function<void(unsigned)> &fn) {
                if (pool && nfrag > 1)
                        pool->run((unsigned)nfrag, fn);
                else
                        for (unsigned k = 0; k < nfrag; ++k)
                                fn(k);
        };
        static const bool dbg_plan = getenv("TRINITY_DEBUG_PLAN") != nullptr; // (stderr: the passes and the serial stretches between them)
        auto dbg_t0 = std::chrono::steady_clock::now();
        std::string dbg_line;
        auto dbg = [&](const char *what) {
                if (!dbg_plan)
                        return;
                const auto now = std::chrono::steady_clock::now();
                char buf[64];
                snprintf(buf, sizeof buf, " %s %.3f", what, std::chrono::duration<double, std::milli>(now - dbg_t0).count());
                dbg_line += buf;
                dbg_t0 = now;
        };
        auto first_error = [&]() -> int {
                for (Frag &f : frags)
                        if (f.rc != TRI_OK) {
                                err = f.err;
                                return f.rc;
                        }
                return TRI_OK;
        };
        dbg("setup");
        run([&](unsigned k) {
                Frag &f = frags[k];
                try {
                        f.rc = lower_range(C, f);
                } catch (const std::bad_alloc &) {
                        f.rc = herr(f.err, TRI_ERR_NOMEM, "tri_batch_create: out of host memory while lowering the batch");
                } catch (...) {
                        f.rc = herr(f.err, TRI_ERR_INVALID, "tri_batch_create: unexpected exception while lowering the batch");
                }
        });
        dbg("LOWER");
        if (int rc = first_error())
                return rc;
        // ---- between the passes: what depends on the whole batch
        uint64_t onepass_queries = 0, fused_postings = 0, phrase_queries = 0;
        for (const Frag &f : frags) {
                onepass_queries += f.onepass_queries;
                fused_postings += f.fused_postings;
                phrase_queries += f.phrase_queries;
        }
        // k_phrase's span is its longest task (a phrase candidate costs a walk into two or three lists' hits: a task of four windows of two head terms is 76 K
        // candidates, 1 - 2 ms) — a batch with few phrase queries per resident workgroup cuts their tasks finer; one with many has tasks enough to fill the tail and
        // keeps the cheaper large ones.  Measured (k_phrase ms at 1 / 2 / 4 / 8): cfg5's shard, 1 250 phrase queries: 2.05 / 1.13 / 0.64 / 0.64; cfg4, 16 384: 8.05 / 8.26 / 8.42 / 8.41
        C.phrase_task_div = opt.phrase_task_div ? opt.phrase_task_div : !phrase_queries ? 1 : phrase_queries <= 8ull * env.cus ? 4 : phrase_queries <= 16ull * env.cus ? 2 : 1;
        // k_planes: docID ranges per query.  A task has fixed costs (seed pass, end-of-task imbalance: about 140 us), the kernel's tail is its
        // longest tasks: two ranges when the batch brings ten or more tasks per resident workgroup anyway, three when it does not (measured,
        // cfg3's mix: 8192 queries 2 > 3 > 4; 3750 queries 6.5 / 5.9 / 6.2 ms for 2 / 3 / 4; 1024 queries 2.11 / 1.97 / 1.96)
        C.planes_split = opt.planes_split ? opt.planes_split : (2 * onepass_queries >= 4ull * (uint64_t)env.cus * env.plk_wgs_per_cu ? 2 : 3) /* (round 5: a task's tail is short now — two ranges from four tasks per resident workgroup on; cfg5's shard, ms: 2 -> 2.25, 3 -> 2.35, 4 -> 2.50) */;
        // one-pass tasks stage the query (slot map, score tables) once per task: the longer the task the better, as long as the batch still
        // cuts into a couple of tasks per workgroup the device holds (measured at cfg3: 512 K postings per task 55.4 ms, 1 M 51.1, 2 M 49.2,
        // 4 M 48.0, 8 M and more 47.1).  fused_task_cost = 0 (the default): sized from the batch; otherwise as given
        C.fused_task_cost = opt.fused_task_cost;
        if (!C.fused_task_cost) {
                const uint64_t want_tasks = 2ull * (uint64_t)env.cus * env.fus_wgs_per_cu;
                C.fused_task_cost = std::min<uint64_t>(8u << 20, std::max<uint64_t>(256u << 10, fused_postings / std::max<uint64_t>(1, want_tasks)));
        }
        dbg("glue0");
        P.plan_ms[0] = ms_since(t0);
        run([&](unsigned k) {
                Frag &f = frags[k];
                try {
                        f.rc = task_range(C, f);
                } catch (const std::bad_alloc &) {
                        f.rc = herr(f.err, TRI_ERR_NOMEM, "tri_batch_create: out of host memory while cutting the batch into tasks");
                } catch (...) {
                        f.rc = herr(f.err, TRI_ERR_INVALID, "tri_batch_create: unexpected exception while cutting the batch into tasks");
                }
        });
        dbg("TASKS");
        if (int rc = first_error())
                return rc;
        P.plan_ms[1] = ms_since(t0);
        // ---- the fragments' places in the batch's arrays; sums
        size_t n_plan = 0, n_qterms = 0, n_sterms = 0, n_phrases = 0, n_pterms = 0, n_tasks = 0, n_fused = 0, n_ptasks = 0, n_units = 0, n_treewords = 0, n_hidden = 0;
        std::vector<uint32_t> tree_terms;
        uint64_t off = 0, cand_lead_docs = 0, cand_terms = 0, cand_queries_all = 0;
        std::vector<uint64_t> benefit(C.n_ok, 0), cand_row(C.n_ok, 0);
        for (Frag &f : frags) {
                f.b_plan = n_plan, f.b_qterms = n_qterms, f.b_sterms = n_sterms, f.b_phrases = n_phrases, f.b_pterms = n_pterms, f.b_tasks = n_tasks, f.b_fused = n_fused,
                f.b_ptasks = n_ptasks, f.b_off = off, f.b_units = n_units;
                n_units += f.units.size();
                f.b_tree = n_treewords, f.b_hidden = n_hidden;
                n_treewords += f.treepool.size(), n_hidden += f.n_hidden;
                tree_terms.insert(tree_terms.end(), f.tree_terms.begin(), f.tree_terms.end());
                P.tree_queries += f.tree_queries;
                P.bitmap_queries += f.bitmap_queries;
                P.pscatter_queries += f.pscatter_queries;
                n_plan += f.tmp.size(), n_qterms += f.qterms.size(), n_sterms += f.sterms.size(), n_phrases += f.phrases.size(), n_pterms += f.pterms.size(),
                        n_tasks += f.tasks.size(), n_fused += f.fused.size(), n_ptasks += f.ptasks.size(), off += f.off;
                P.term_bytes += f.term_bytes, P.term_bytes_phrase_hits += f.term_bytes_phrase_hits, P.term_bytes_dense += f.term_bytes_dense,
                        P.term_bytes_fused += f.term_bytes_fused, P.term_bytes_planes += f.term_bytes_planes, P.cand_needed_term_bytes += f.cand_needed;
                cand_lead_docs += f.cand_lead_docs, cand_terms += f.cand_terms, cand_queries_all += f.cand_queries + f.probe_queries;
                P.dense_queries += f.dense_queries, P.cand_queries += f.cand_queries, P.fused_queries += f.fused_queries, P.planes_queries += f.planes_queries;
                P.pset_queries += f.pset_queries, P.term_bytes_pset += f.term_bytes_pset;
                P.probe_queries += f.probe_queries, P.term_bytes_probe += f.term_bytes_probe;
                P.rich_R = std::max(P.rich_R, f.rich_R);
                P.rich_allow |= f.rich_allow;
                P.sparse_cap = std::max(P.sparse_cap, f.sparse_cap);
                for (uint32_t r = 0; r < C.n_ok; ++r)
                        benefit[r] += f.benefit[r], cand_row[r] += f.cand_row[r];
                for (const size_t qi : f.left_out) { // a query shape the planner does not lower does not fail the batch: the query is left out (status
                                                     // TRI_ERR_UNSUPPORTED, no matches) and the caller keeps its CPU span for it
                        P.qstatus[qi] = TRI_ERR_UNSUPPORTED;
                        ++P.unsupported_queries;
                }
                if (!f.left_out.empty())
                        P.last_unsupported = f.err;
        }
        if (n_qterms > 0xfffffff0ull || n_sterms > 0xfffffff0ull || n_tasks > 0xfffffff0ull || n_pterms > 0xfffffff0ull)
                return herr(err, TRI_ERR_UNSUPPORTED, "tri_batch_create: the batch exceeds 2^32 terms or tasks: split it");
        P.out_capacity = off;
        std::sort(tree_terms.begin(), tree_terms.end());
        tree_terms.erase(std::unique(tree_terms.begin(), tree_terms.end()), tree_terms.end());
        P.tree_scratch_bytes = ((uint64_t)tree_terms.size() * PL_PLANES + n_hidden + P.tree_queries) * P.plw * 4;
        if (P.tree_scratch_bytes > opt.tree_max_bytes)
                return herr(err, TRI_ERR_NOMEM, "tri_batch_create: the batch's %llu tree queries need %llu bytes of bitmap scratch (%zu distinct term leaves, %zu phrase leaves; option tree_max_bytes = %llu): split the batch",
                            (unsigned long long)P.tree_queries, (unsigned long long)P.tree_scratch_bytes, tree_terms.size(), n_hidden, (unsigned long long)opt.tree_max_bytes);
        // ---- the planes that pay: rows in term order (deterministic), the uses pointed at them.  A term is chosen when the batch's uses repay
        //      one decode of its list (a one-pass slot counts a whole decode: always chosen)
        std::vector<uint32_t> chosen; // terms
        {
                std::vector<uint32_t> rank_term; // df rank -> term, for the eligible ranks only (built lazily from the uses)
                rank_term.assign(C.n_ok, UINT32_MAX);
                for (const Frag &f : frags) {
                        for (const QUse &u : f.quses)
                                rank_term[ix.df_rank[u.term]] = u.term;
                        for (const QUse &u : f.suses)
                                rank_term[ix.df_rank[u.term]] = u.term;
                        for (const FUse &u : f.fuses)
                                rank_term[ix.df_rank[u.term]] = u.term;
                }
                std::vector<uint8_t> forced(C.n_ok, 0);
                for (const Frag &f : frags)
                        for (const FUse &u : f.fuses)
                                forced[ix.df_rank[u.term]] = 1;
                for (uint32_t r = 0; r < C.n_ok; ++r)
                        if (rank_term[r] != UINT32_MAX && (forced[r] || benefit[r] * std::max<uint64_t>(1, opt.plane_amortize) >= ix.terms[rank_term[r]].documents))
                                chosen.push_back(rank_term[r]);
                std::sort(chosen.begin(), chosen.end());
        }
        // a term's plane row is its DF RANK: the rows live with the INDEX (tri_index's plane cache: a head term is decoded into its planes the
        // first time any batch wants them and stays — the index does not change), so every batch addresses the same row for the same term
        std::vector<uint32_t> row_of_rank(C.n_ok, PL_NONE);
        for (size_t i = 0; i < chosen.size(); ++i) {
                row_of_rank[ix.df_rank[chosen[i]]] = ix.df_rank[chosen[i]];
                P.plane_decoded_bytes += ix.docbytes[chosen[i]];
        }
        P.plane_rows = C.n_ok;
        // k_and's tasks ordered by the plane row they probe ("k_and's queues" below): where the probes are the kernel's traffic — conjunctions of two or
        // three terms whose leads average a thousand documents or more (cfg2: 3.4 K).  Rare leads against four lists (cfg3 / cfg5: a few hundred
        // candidates a task, several rows each) gain nothing from the order and lose the heaviest-first start: measured 0.49 -> 0.54 ms, 0.68 -> 0.72 ms
        const uint32_t pset_ranges = ((ix.max_doc >> 17) + 1u + PSET_TASK_WINDOWS - 1) / PSET_TASK_WINDOWS; // window ranges of the docID space (a TASK_PSET task's range: its first window / PSET_TASK_WINDOWS)
        const bool cand_rows = opt.cand_xcd && !chosen.empty() && cand_queries_all && cand_lead_docs >= CAND_ROWS_MIN_LEAD * cand_queries_all && cand_terms <= 3 * cand_queries_all;
        uint32_t cand_first = 0, cand_qat[CAND_QUEUES] = {};
        // row -> queue(s) and the row's place in the queue: heaviest row first to the least loaded queue (the rows are a few hundred); a row that outweighs
        // a 16th of the section is cut into pieces of that size, each placed on its own (a Zipf batch's first term is probed by a sixth of the tasks)
        struct RowQ {
                uint8_t n = 0, q[CAND_QUEUES] = {}, sub[CAND_QUEUES] = {};
        };
        std::vector<RowQ> rowq(cand_rows ? C.n_ok : 0);
        if (cand_rows) {
                uint64_t total = 0, load[CAND_QUEUES] = {};
                uint32_t placed[CAND_QUEUES] = {};
                std::vector<uint32_t> rows;
                for (uint32_t r = 0; r < C.n_ok; ++r)
                        if (cand_row[r] && row_of_rank[r] != PL_NONE)
                                rows.push_back(r), total += cand_row[r];
                std::sort(rows.begin(), rows.end(), [&](uint32_t a, uint32_t b) { return cand_row[a] != cand_row[b] ? cand_row[a] > cand_row[b] : a < b; });
                const uint64_t cap = std::max<uint64_t>(1, total / (2 * CAND_QUEUES));
                for (const uint32_t r : rows) {
                        RowQ &z = rowq[r];
                        z.n = (uint8_t)std::min<uint64_t>(CAND_QUEUES, (cand_row[r] + cap - 1) / cap);
                        for (uint32_t k = 0; k < z.n; ++k) {
                                const uint32_t x = (uint32_t)(std::min_element(load, load + CAND_QUEUES) - load);
                                load[x] += cand_row[r] / z.n;
                                z.q[k] = (uint8_t)x;
                                z.sub[k] = (uint8_t)(CAND_COST_SUBS + placed[x]++ % (CAND_SUBS - CAND_COST_SUBS - 1));
                        }
                }
        }
        if (dbg_plan) {
                char buf[96];
                snprintf(buf, sizeof buf, " [cand queries %llu lead docs %llu terms %llu rows %d]", (unsigned long long)cand_queries_all, (unsigned long long)cand_lead_docs, (unsigned long long)cand_terms, (int)cand_rows);
                dbg_line += buf;
        }
        // ---- layout of the host block
        size_t bytes = 0;
        auto section = [&](size_t &off_out, size_t n, size_t elem) {
                off_out = bytes;
                bytes += (n * elem + SECTION_ALIGN + SECTION_ALIGN - 1) & ~(SECTION_ALIGN - 1); // (a spare 64 bytes behind every array: wide loads at an array's end stay inside the block)
        };
        const size_t n_qplane = chosen.empty() ? 0 : n_qterms;
        section(P.off_plan, n_plan, sizeof(DevQuery));
        section(P.off_qterms, n_qterms, 4);
        section(P.off_tasks, n_tasks, sizeof(DevTask));
        section(P.off_sched, n_tasks, 4);
        section(P.off_fused, n_fused, sizeof(DevFused));
        section(P.off_qplane, n_qplane, 4);
        section(P.off_plane_terms, chosen.size(), 4);
        const size_t n_splane = (chosen.empty() || !C.scored) ? 0 : n_sterms;
        section(P.off_splane, n_splane, 4);
        section(P.off_sterms, n_sterms, 4);
        section(P.off_sweights, C.scored ? n_sterms : 0, 8);
        section(P.off_phrases, n_phrases, sizeof(DevPhrase));
        section(P.off_pterms, n_pterms, 4);
        section(P.off_ptasks, n_ptasks, 4);
        section(P.off_units, n_units, sizeof(DevPsetUnit));
        section(P.off_pset_sched, n_units, 4);
        section(P.off_cand_q, CAND_QUEUES + 1, 4);
        section(P.off_tree, n_treewords, 4);
        section(P.off_tree_terms, tree_terms.size(), 4);
        section(P.off_tree_hidden, n_hidden, 4);
        P.block_bytes = bytes;
        dbg("sums+planes+layout");
        P.block = alloc_block(bytes);
        dbg("alloc_block");
        if (!P.block)
                return herr(err, TRI_ERR_NOMEM, "tri_batch_create: no host memory for the plan (%zu bytes)", bytes);
        auto span = [&](auto &s, size_t off_, size_t n) {
                using T = std::remove_reference_t<decltype(*s.p)>;
                s.p = reinterpret_cast<T *>(P.block + off_);
                s.n = n;
        };
        span(P.plan, P.off_plan, n_plan);
        span(P.qterms, P.off_qterms, n_qterms);
        span(P.tasks, P.off_tasks, n_tasks);
        span(P.sched, P.off_sched, n_tasks);
        span(P.fused, P.off_fused, n_fused);
        span(P.qplane, P.off_qplane, n_qplane);
        span(P.plane_terms, P.off_plane_terms, chosen.size());
        span(P.splane, P.off_splane, n_splane);
        span(P.sterms, P.off_sterms, n_sterms);
        span(P.sweights, P.off_sweights, C.scored ? n_sterms : 0);
        span(P.phrases, P.off_phrases, n_phrases);
        span(P.pterms, P.off_pterms, n_pterms);
        span(P.ptasks, P.off_ptasks, n_ptasks);
        span(P.units, P.off_units, n_units);
        span(P.pset_sched, P.off_pset_sched, n_units);
        span(P.cand_q, P.off_cand_q, CAND_QUEUES + 1);
        span(P.tree, P.off_tree, n_treewords);
        span(P.tree_terms, P.off_tree_terms, tree_terms.size());
        span(P.tree_hidden, P.off_tree_hidden, n_hidden);
        std::copy(tree_terms.begin(), tree_terms.end(), P.tree_terms.p);
        std::vector<uint32_t> unit_of_task(n_units ? n_tasks : 0);
        std::copy(chosen.begin(), chosen.end(), P.plane_terms.p);
        dbg("spans");
        // ---- every fragment writes its part of the arrays, rebased
        run([&](unsigned k) {
                Frag &f = frags[k];
                for (size_t i = 0; i < f.tmp.size(); ++i) {
                        DevQuery q = f.tmp[i].q;
                        q.term_base += (uint32_t)f.b_qterms;
                        q.score_base += (uint32_t)f.b_sterms;
                        q.phrase_base += (uint32_t)f.b_phrases;
                        q.first_task += (uint32_t)f.b_tasks;
                        q.out_off += f.b_off;
                        if (f.tmp[i].fuse)
                                q.fused_idx += (uint32_t)f.b_fused;
                        if (f.tmp[i].tree)
                                q.fused_idx += (uint32_t)f.b_tree;
                        if (f.tmp[i].hidden) { // (no caller query of its own: its matches are a leaf of a TASK_TREE query)
                                q.qid = 0xffffffffu;
                                P.tree_hidden[f.b_hidden + f.tmp[i].hidden_ord] = (uint32_t)(f.b_plan + i);
                        } else
                                P.slot_of_query[q.qid] = (uint32_t)(f.b_plan + i);
                        P.plan[f.b_plan + i] = q;
                }
                if (!f.treepool.empty()) {
                        memcpy(&P.tree[f.b_tree], f.treepool.data(), f.treepool.size() * 4);
                        for (size_t i = 0; i < f.tmp.size(); ++i) {
                                if (!f.tmp[i].tree)
                                        continue;
                                uint32_t *rec = &P.tree[f.b_tree + f.tmp[i].q.fused_idx];
                                DevTreeNode *tn = reinterpret_cast<DevTreeNode *>(rec + TREE_HDR_WORDS);
                                for (uint32_t k = 0; k < rec[0]; ++k)
                                        if (tn[k].op == TRI_OP_TERM)
                                                tn[k].row = (uint32_t)(std::lower_bound(P.tree_terms.begin(), P.tree_terms.end(), tn[k].arg) - P.tree_terms.begin());
                                        else if (tn[k].op == TRI_OP_PHRASE) {
                                                tn[k].arg += (uint32_t)f.b_plan;
                                                tn[k].row += (uint32_t)f.b_hidden;
                                        }
                        }
                }
                for (size_t i = 0; i < f.tasks.size(); ++i) {
                        DevTask t = f.tasks[i];
                        t.slot += (uint32_t)f.b_plan;
                        t.out_off += f.b_off;
                        P.tasks[f.b_tasks + i] = t;
                }
                if (!f.qterms.empty())
                        memcpy(&P.qterms[f.b_qterms], f.qterms.data(), f.qterms.size() * 4);
                if (!f.sterms.empty())
                        memcpy(&P.sterms[f.b_sterms], f.sterms.data(), f.sterms.size() * 4);
                if (C.scored && !f.sweights.empty())
                        memcpy(&P.sweights[f.b_sterms], f.sweights.data(), f.sweights.size() * 8);
                for (size_t i = 0; i < f.phrases.size(); ++i) {
                        DevPhrase ph = f.phrases[i];
                        ph.term_base += (uint32_t)f.b_pterms;
                        P.phrases[f.b_phrases + i] = ph;
                }
                if (!f.pterms.empty())
                        memcpy(&P.pterms[f.b_pterms], f.pterms.data(), f.pterms.size() * 4);
                for (size_t i = 0; i < f.ptasks.size(); ++i)
                        P.ptasks[f.b_ptasks + i] = f.ptasks[i] + (uint32_t)f.b_tasks;
                for (size_t i = 0; i < f.fused.size(); ++i)
                        P.fused[f.b_fused + i] = f.fused[i];
                for (const FUse &u : f.fuses)
                        P.fused[f.b_fused + u.fidx].plane[u.slot] = row_of_rank[ix.df_rank[u.term]];
                for (size_t i = 0; i < f.units.size(); ++i) {
                        DevPsetUnit u = f.units[i];
                        u.out_off += f.b_off;
                        u.tix += (uint32_t)f.b_tasks;
                        u.term_base += (uint32_t)f.b_qterms;
                        const bool is_probe = f.tasks[f.units[i].tix].kind == TASK_PROBE;
                        bool rows = true;
                        for (uint32_t k = is_probe ? 1u : 0u; k < u.nterms; ++k) { // (a TASK_PROBE unit's term 0 is the lead: decoded, never probed)
                                const uint32_t term = (k < PSET_INLINE_TERMS ? u.tt[k] : f.qterms[f.units[i].term_base + k]) & QT_TERM;
                                const uint32_t rank = ix.df_rank[term];
                                const uint32_t row = rank < row_of_rank.size() ? row_of_rank[rank] : PL_NONE; // (a PSET_UNIT_SCATTER union names terms without a plane)
                                rows &= row != PL_NONE;
                                if (k < PSET_INLINE_TERMS)
                                        u.row[k] = row;
                                if (row == PL_NONE && (u.first & PSET_UNIT_SCATTER) && u.w_begin == 0) // (a scatter union's first task: the documents k_psets_prep lists for the query)
                                        f.pscatter_docs += ix.terms[term].documents;
                        }
                        if (is_probe && !rows) { // a probed list did not get its plane (the batch's uses do not repay its decode): candidate tiles after all
                                P.tasks[u.tix].kind = TASK_CAND;
                                if (u.first & PSET_UNIT_FIRST) {
                                        ++f.probe_demoted;
                                        const uint32_t *qt = &f.qterms[f.units[i].term_base];
                                        auto &seen = f.S.seen;
                                        seen.clear();
                                        for (uint32_t k = 0; k < u.nterms; ++k)
                                                if (std::find(seen.begin(), seen.end(), qt[k] & QT_TERM) == seen.end()) {
                                                        seen.push_back(qt[k] & QT_TERM);
                                                        f.probe_demoted_bytes += ix.docbytes[qt[k] & QT_TERM];
                                                }
                                }
                        }
                        P.units[f.b_units + i] = u;
                        unit_of_task[u.tix] = (uint32_t)(f.b_units + i);
                }
                if (n_splane) {
                        std::fill(&P.splane.p[f.b_sterms], &P.splane.p[f.b_sterms] + f.sterms.size(), PL_NONE);
                        for (const QUse &u : f.suses)
                                P.splane[f.b_sterms + u.qpos] = row_of_rank[ix.df_rank[u.term]];
                }
                if (n_qplane) {
                        std::fill(&P.qplane.p[f.b_qterms], &P.qplane.p[f.b_qterms] + f.qterms.size(), PL_NONE);
                        for (const QUse &u : f.quses)
                                P.qplane[f.b_qterms + u.qpos] = row_of_rank[ix.df_rank[u.term]];
                }
                f.keys.resize(f.tasks.size());
                f.hist.assign(SCHED_KEYS, 0u);
                for (size_t i = 0; i < f.tasks.size(); ++i) { // (the kinds are final: a probe task whose planes were not chosen is a candidate-tile task by now)
                        const DevTask &tk = P.tasks[f.b_tasks + i];
                        if ((tk.kind == TASK_PLANES || tk.kind == TASK_PLANES8) && opt.planes_order) {
                                // (option planes_order: k_planes' tasks range by range, within a range by the heaviest plane row they sweep — the workgroups
                                //  in flight then stream the same head rows from the same place: L2 instead of the fabric)
                                const DevQuery &q = P.plan[tk.slot];
                                const DevFused &z = P.fused[q.fused_idx];
                                uint32_t minrow = PL_NONE;
                                for (uint32_t sidx = 0; sidx < z.nslots; ++sidx)
                                        minrow = std::min(minrow, z.plane[sidx]);
                                const uint32_t ord = (uint32_t)(f.b_tasks + i) - q.first_task, per = SCHED_NB / 4;
                                ++f.hist[f.keys[i] = SCHED_RANK[tk.kind] * SCHED_NB + (opt.planes_order == 2 ? std::min(minrow, per - 1) * 4 + std::min(ord, 3u) : std::min(ord, 3u) * per + std::min(minrow, per - 1))];
                                continue;
                        }
                        if (tk.kind == TASK_PSET && opt.pset_order) {
                                // (option pset_order: k_psets' tasks range by range, within a range by the query's heaviest term — the tasks in flight read ITS words of
                                //  the range one after the other, the second reader on from L2)
                                const uint32_t range = (uint32_t)(f.tcost[i] & 0xffffffffull) / PSET_TASK_WINDOWS;
                                const uint32_t rb = pset_ranges <= PSET_RANGE_BKS ? std::min(range, PSET_RANGE_BKS - 1) : (uint32_t)std::min<uint64_t>((uint64_t)range * PSET_RANGE_BKS / pset_ranges, PSET_RANGE_BKS - 1);
                                ++f.hist[f.keys[i] = PSET_KEY0 + rb * PSET_SUBS + pset_sub((uint32_t)(f.tcost[i] >> 32))];
                                continue;
                        }
                        if (tk.kind != TASK_CAND || !cand_rows) {
                                ++f.hist[f.keys[i] = sched_key(tk.kind, f.tcost[i])];
                                continue;
                        }
                        // k_and's queues (below): the XCD's queue and the place in it by the plane row the task probes first
                        const DevQuery &q = P.plan[tk.slot];
                        uint32_t row = PL_NONE;
                        for (uint32_t k = 1; k < q.nterms && row == PL_NONE; ++k)
                                row = P.qplane[q.term_base + k];
                        uint32_t queue, sub;
                        if (tk.tile_end - tk.tile_begin > CAND_HEAVY_TILES || row == PL_NONE)
                                // the long tasks, and the ones that gallop through every list (100 us and more where a probing task takes 20): first, dealt
                                // round the queues, heaviest first — left to the end they were the kernel's tail (a tenth of its span on a tenth of the workgroups)
                                queue = (uint32_t)(f.b_tasks + i) % CAND_QUEUES, sub = (sched_key(TASK_CAND, f.tcost[i]) % SCHED_NB) / (SCHED_NB / CAND_COST_SUBS);
                        else if (row < rowq.size() && rowq[row].n) {
                                const RowQ &z = rowq[row];
                                const uint32_t piece = (uint32_t)(f.b_tasks + i) % z.n;
                                queue = z.q[piece], sub = z.sub[piece];
                        } else // (a row the tally above did not see: a demoted probe task's)
                                queue = row % CAND_QUEUES, sub = CAND_SUBS - 1;
                        ++f.hist[f.keys[i] = CAND_KEY0 + queue * CAND_SUBS + sub];
                }
        });
        for (const Frag &f : frags) { // (what the fill pass sent back to the candidate tiles)
                P.pscatter_docs += f.pscatter_docs;
                P.probe_queries -= f.probe_demoted, P.cand_queries += f.probe_demoted;
                P.term_bytes_probe -= f.probe_demoted_bytes;
        }
        dbg("FILL");
        P.plan_ms[2] = ms_since(t0);
        // ---- the schedule: per kernel, heaviest tasks first.  A counting sort by (kernel, cost octave + 2 bits) — tasks within a fifth of each
        //      other keep their order in the batch: all a longest-first dispatch needs; TASK_PSET goes by docID window range instead.  The
        //      fragments counted their tasks per bucket in the fill pass; their places are settled here, the scatter runs on the pool again
        {
                uint32_t *const per_kernel[TASK_KINDS] = {&P.n_dense, &P.n_pset, &P.n_probe, &P.n_cand, &P.n_fused, &P.n_fused16, &P.n_fusedgen, &P.n_planes, &P.n_planes8, &P.n_tree};
                uint32_t at = 0;
                for (uint32_t r = 0; r < TASK_KINDS; ++r) {
                        const uint32_t before = at;
                        if (r == SCHED_RANK[TASK_CAND])
                                cand_first = at;
                        auto place = [&](const uint32_t bk) {
                                for (Frag &f : frags) {
                                        const uint32_t c = f.hist[bk];
                                        f.hist[bk] = at; // (count -> the fragment's first place in the bucket)
                                        at += c;
                                }
                        };
                        for (uint32_t bk = r * SCHED_NB; bk < (r + 1) * SCHED_NB; ++bk)
                                place(bk);
                        if (r == SCHED_RANK[TASK_PSET] && opt.pset_order) // (k_psets' tasks by range and heaviest term)
                                for (uint32_t bk = PSET_KEY0; bk < PSET_KEY0 + std::min(pset_ranges, PSET_RANGE_BKS) * PSET_SUBS; ++bk)
                                        place(bk);
                        if (r == SCHED_RANK[TASK_CAND]) // (a `cand_rows` batch: k_and's tasks are all in the row buckets)
                                for (uint32_t bk = CAND_KEY0; bk < PSET_KEY0; ++bk) {
                                        if ((bk - CAND_KEY0) % CAND_SUBS == 0)
                                                cand_qat[(bk - CAND_KEY0) / CAND_SUBS] = at;
                                        place(bk);
                                }
                        *per_kernel[r] = at - before;
                }
                const uint32_t n_dense = P.n_dense, n_units_run = P.n_pset + P.n_probe;
                run([&](unsigned k) {
                        Frag &f = frags[k];
                        for (size_t i = 0; i < f.tasks.size(); ++i) {
                                const uint32_t pos = f.hist[f.keys[i]]++, ti = (uint32_t)(f.b_tasks + i);
                                P.sched[pos] = ti;
                                if (pos >= n_dense && pos - n_dense < n_units_run) // (a TASK_PSET / TASK_PROBE task: its unit record runs at the same place)
                                        P.pset_sched[pos - n_dense] = unit_of_task[ti];
                        }
                });
        }
        // ---- k_and's queues.  A candidate tile probes the planes of the query's other terms: ONE bit per candidate, a 64-byte sector of a 1.25 MB row
        //      each — in cost order the tasks in flight probe a hundred rows at once and every sector comes from HBM (cfg2: 2.9 GB per step of them).
        //      The section is cut into one queue per XCD (workgroups draw from the queue of the XCD they run on, and from the next ones when theirs is
        //      empty).  A batch of few-term conjunctions with long leads (`cand_rows`) orders a queue BY THE ROW ITS TASKS PROBE, every row in ONE
        //      queue (the keys above; the counting sort placed them): an XCD works through a couple of rows at a time, its 4 MB L2 keeps their sectors
        //      for the row's next tasks — k_and reads 0.76 GB (PMC, profiles/README.md).  Any other batch: the cost order, dealt round the queues
        {
                const uint32_t nc = P.n_cand;
                uint32_t *const sc = P.sched.p + cand_first;
                if (cand_rows) {
                        for (uint32_t x = 0; x <= CAND_QUEUES; ++x)
                                P.cand_q[x] = x < CAND_QUEUES ? cand_qat[x] - cand_first : nc;
                } else {
                        const std::vector<uint32_t> was(sc, sc + nc);
                        uint32_t pos = 0;
                        for (uint32_t x = 0; x < CAND_QUEUES; ++x) {
                                P.cand_q[x] = pos;
                                for (uint32_t i = x; i < nc; i += CAND_QUEUES)
                                        sc[pos++] = was[i];
                        }
                        P.cand_q[CAND_QUEUES] = pos;
                }
        }
        // ---- k_phrase's tasks, heaviest first: a task's candidates are bounded by its output region (the lead's documents in its range); in query
        //      order the 4 ms tasks of a head x head phrase started anywhere in the kernel's span and its last fifth ran on a tenth of the
        //      workgroups (cfg4: 70 % busy).  A counting sort by the region's size (octave + 2 bits), descending, stable
        if (P.ptasks.size() > 1) {
                const size_t np = P.ptasks.size();
                std::vector<uint32_t> key(np), sorted(np);
                uint32_t hist[SCHED_NB + 1] = {};
                for (size_t i = 0; i < np; ++i) {
                        const uint32_t ti = P.ptasks[i];
                        const DevTask &tk = P.tasks[ti];
                        const DevQuery &q = P.plan[tk.slot];
                        const uint64_t end = ti + 1 < q.first_task + q.ntasks ? P.tasks[ti + 1].out_off : q.out_off + q.out_cap;
                        const uint64_t c = std::max<uint64_t>(1, end - tk.out_off);
                        const uint32_t lg = 63u - (uint32_t)__builtin_clzll(c);
                        const uint32_t frac = lg >= 2 ? (uint32_t)((c >> (lg - 2)) & 3u) : (uint32_t)((c << (2 - lg)) & 3u);
                        key[i] = SCHED_NB - 1 - (lg * 4 + frac);
                        ++hist[key[i] + 1];
                }
                for (uint32_t bk = 0; bk < SCHED_NB; ++bk)
                        hist[bk + 1] += hist[bk];
                for (size_t i = 0; i < np; ++i)
                        sorted[hist[key[i]]++] = P.ptasks[i];
                std::copy(sorted.begin(), sorted.end(), P.ptasks.p);
        }
        P.sparse_cap = (P.sparse_cap + 63u) & ~63u;
        dbg("sched+rest");
        if (dbg_plan)
                fprintf(stderr, "[tri plan] nq %zu frags %zu:%s\n", nq, nfrag, dbg_line.c_str());
        P.plan_ms[3] = ms_since(t0);
        if (opt.account_needed_bytes && !ix.terms.empty()) {
                // (diagnostic: one pass over the plan with a mark per term and class)
                std::vector<uint8_t> seen(ix.terms.size(), 0); // bit k: counted for kind k; bit 7: counted for the batch
                std::vector<uint8_t> seen_hits(ix.terms.size(), 0);
                for (size_t sidx = 0; sidx < n_plan; ++sidx) {
                        const DevQuery &q = P.plan[sidx];
                        const uint32_t kind = P.tasks[q.first_task].kind;
                        auto touch = [&](uint32_t term) {
                                if (!(seen[term] & 0x80u))
                                        P.distinct_bytes += ix.docbytes[term];
                                if (!(seen[term] & (1u << kind)))
                                        P.distinct_bytes_kind[kind] += ix.docbytes[term];
                                seen[term] |= (uint8_t)(0x80u | (1u << kind));
                        };
                        if (kind == TASK_TREE) {
                                const uint32_t *rec = &P.tree[q.fused_idx];
                                const DevTreeNode *tn = reinterpret_cast<const DevTreeNode *>(rec + TREE_HDR_WORDS);
                                for (uint32_t k = 0; k < rec[0]; ++k)
                                        if (tn[k].op == TRI_OP_TERM)
                                                touch(tn[k].arg);
                        } else if (task_onepass(kind)) {
                                const DevFused &z = P.fused[q.fused_idx];
                                for (uint32_t k = 0; k < z.nslots; ++k)
                                        touch(z.term[k]);
                        } else {
                                for (uint32_t k = 0; k < q.nterms; ++k)
                                        touch(P.qterms[q.term_base + k] & QT_TERM);
                                if (mode != TRI_FLAG_DOCUMENTS_ONLY) // (k_score / k_rich read the scorer / reported terms' lists)
                                        for (uint32_t k = 0; k < q.nscore; ++k)
                                                touch(P.sterms[q.score_base + k]);
                        }
                        auto touch_hits = [&](uint32_t term, bool phrase) {
                                if (!(seen_hits[term] & 1u))
                                        P.distinct_bytes += ix.hitbytes[term];
                                if (phrase && !(seen_hits[term] & 2u))
                                        P.distinct_bytes_kind[TASK_KINDS] += ix.hitbytes[term];
                                seen_hits[term] |= (uint8_t)(1u | (phrase ? 2u : 0u));
                        };
                        for (uint32_t ph = 0; ph < q.nphrases; ++ph)
                                for (uint32_t k = 0; k < P.phrases[q.phrase_base + ph].nterms; ++k)
                                        touch_hits(P.pterms[P.phrases[q.phrase_base + ph].term_base + k], true);
                        if (C.rich)
                                for (uint32_t k = 0; k < q.nscore; ++k)
                                        touch_hits(P.sterms[q.score_base + k], false);
                }
        }
        return TRI_OK;
}
