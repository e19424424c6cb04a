// trinity_hip.hip — libtrinity_hip.so: MI355X (gfx950 / CDNA4) execution engine for Trinity's query hot
// path.  Hand-written HIP; wave64; no MFMA (integer/byte work bounded by HBM + LDS + VALU issue).
//
// Data layout in HBM (built once at tri_index_upload):
//   index[]        raw reference-format segment bytes (google_codec.cpp:9-176 layout), +64 B slack
//   blk_last[]     u32 last docID of every block, all terms concatenated      (SoA: searched, 4 B/blk)
//   blk_off[]      u32 byte offset of every block's payload (first delta byte) (SoA: touched on decode)
//   terms[]        {documents, first_block, nblocks, last_n} per term
// The reference discovers block boundaries by hopping headers serially (google_codec.cpp:641-697) and
// keeps a sparse skiplist; a dense directory is the GPU analogue of Decoder::init (936-983).
//
// Kernels
//   k_decode_terms   one lane per block: prefix-varint stream decode of deltas+freqs (unpack_block 596-639)
//   k_and            persistent workgroups pull queries; per query the lead (lowest-df) list is decoded in
//                    tiles of 256 blocks into an LDS candidate array; every other term filters the tile:
//                    block-driven (dense) or candidate-driven galloping (sparse) over the block directory,
//                    one lane per needed block, merging the decoded docs against the candidates in LDS
//                    (Conjuction::next_impl leapfrog, docset_iterators.cpp:308-348, as a set operation)
#include "../../include/trinity_hip.h"
#include <chrono>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <unistd.h>
#include <dlfcn.h>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <unordered_map>
#include <functional>
#include <vector>

// ------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(g_err, sizeof g_err, fmt, ap);
        va_end(ap);
        return code;
}
#define HIP_TRY(expr)                                                                                       \
        do {                                                                                                \
                hipError_t e_ = (expr);                                                                     \
                if (e_ != hipSuccess)                                                                       \
                        return fail(e_ == hipErrorOutOfMemory ? TRI_ERR_NOMEM : TRI_ERR_DEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
        } while (0)

extern "C" const char *tri_last_error(void) { return g_err; }
extern "C" int tri_abi_version(void) { return TRI_ABI_VERSION; }

// the host planner (planner.hpp: options, PlanEnv, BatchPlan, plan_batch), the upload-time walk (index_host.hpp) and the structures
// the planner shares with the kernels (dev_structs.hpp)
#include "planner.hpp"

struct tri_dev {
        int device;
        hipStream_t stream, stream2; // stream2: the candidate-tile kernel when the two matching kernels run side by side
        hipStream_t stream_up;       // plans travel to the device on their own stream: a batch is compiled and uploaded while the previous one runs
        hipStream_t stream_rb;       // read-backs of a SYNCED batch's results (docID sets, scores, hashes): they wait for nothing queued behind that batch on the engine stream
        hipEvent_t ev_fork, ev_join;
        int cus;
        tri_options opt;
        int refs = 0;         // indexes and batches alive on this handle ...
        bool closing = false; // ... tri_dev_close with some left: the handle goes with the last of them
        // The large buffers of a batch (output regions, score streams, term planes, decoded lists) and its plan arena are recycled from
        // batch to batch: a caller that compiles a batch per step would otherwise hipMalloc and hipFree gigabytes per step — hipFree
        // synchronises the device (the next batch cannot be compiled while the current one runs), and a cold 15 GB hipMalloc was measured
        // anywhere between 10 ms and 1 s (bench.py's end_to_end.batch_create_cold_ms).  One tri_dev per host thread: no lock.
        struct Pool {
                std::vector<std::pair<size_t, void *>> idle;    // (bytes, buffer) not in use
                std::unordered_map<void *, size_t> size_of;     // every pooled buffer, in use or idle
                size_t idle_bytes = 0;
        } pool;
        // ... likewise the pinned host blocks the plans are laid out in (hipHostMalloc costs about a millisecond per megabyte) ...
        std::vector<std::pair<size_t, void *>> pinned_idle;
        // ... and the HIP events of a batch (nine per batch)
        std::vector<hipEvent_t> events_idle;
        // The planner's host threads: PLAN_CTXS independent planner contexts — a pool of host threads and the per-fragment arrays it recycles, under a
        // lock of their own — so that TWO threads may compile batches of this device at the same time (a create is 0.6 - 1.1 ms of host planning for
        // 16 K queries on 8 - 16 threads and scales no further: a caller whose steps are shorter than that compiles two batches side by side).  A
        // context's pool is started by the first large batch that finds the context free; pool k pins its workers to its own stretch of the CPUs.
        static constexpr unsigned PLAN_CTXS = 2;
        struct PlanCtx {
                std::mutex mu;                  // one batch at a time per context
                std::unique_ptr<HostPool> pool; // (created under mu)
                trip::FragCache frag_cache;     // (under mu)
        } planners[PLAN_CTXS];
        int plan_anchor = -1; // where the handle's pools count their CPUs from among the pin candidates (HostPool::anchor of the first pool started): context k's stretch starts k pools further on
        // The batch calls of one handle may come from TWO host threads: one compiling the next batch (tri_batch_create) while the other runs,
        // awaits and releases earlier ones (tri_batch_run / _sync / _destroy) — bench.py's loop; the planner's share of a create (most of it)
        // runs outside the lock, the pools above, the index's plane cache and everything that enqueues on the streams inside it.  Every other
        // entry point (uploads, options, encoders): one thread at a time, as before.
        std::recursive_mutex mu;
};
using DevLock = std::lock_guard<std::recursive_mutex>;
constexpr size_t TICKET_SCAT_WORD = 44;        // ... k_psets_prep's cursor into the batch's scatter list
constexpr size_t TICKET_CAND_WORD = 64;        // a batch's ticket words: [0, 64) one per kernel; then k_and's CAND_QUEUES, 64 bytes apart
constexpr size_t TICKET_BYTES = (TICKET_CAND_WORD + CAND_QUEUES * CAND_TICKET_STRIDE) * 4;
constexpr size_t POOL_MIN_BYTES = 64u << 10;  // smaller buffers are not worth pooling
constexpr size_t POOL_IDLE_CAP = 64ull << 30; // idle buffers beyond this are given back to the device (largest first)
constexpr size_t PINNED_IDLE_MAX = 16;        // idle pinned blocks kept (the longest idle one is dropped for a newly released one)

static void dev_destroy(tri_dev *d) {
        hipSetDevice(d->device);
        for (auto &pc : d->planners)
                pc.pool.reset();
        hipEventDestroy(d->ev_fork);
        hipEventDestroy(d->ev_join);
        for (hipEvent_t e : d->events_idle)
                hipEventDestroy(e);
        hipStreamDestroy(d->stream_up);
        hipStreamDestroy(d->stream_rb);
        hipStreamDestroy(d->stream2);
        hipStreamDestroy(d->stream);
        for (auto &b : d->pool.idle)
                hipFree(b.second);
        for (auto &b : d->pinned_idle)
                hipHostFree(b.second);
        delete d;
}
static void dev_retain(tri_dev *d) {
        DevLock g(d->mu);
        ++d->refs;
}
static void dev_release(tri_dev *d) {
        if (!d)
                return;
        bool last;
        {
                DevLock g(d->mu);
                last = --d->refs == 0 && d->closing;
        }
        if (last)
                dev_destroy(d);
}

// a buffer of at least `bytes`: an idle one of the pool that is not more than twice as large, else a fresh allocation
static hipError_t pool_alloc(tri_dev *dev, void **out, const size_t bytes) {
        if (bytes < POOL_MIN_BYTES)
                return hipMalloc(out, bytes);
        DevLock g(dev->mu);
        auto &P = dev->pool;
        size_t best = SIZE_MAX;
        for (size_t i = 0; i < P.idle.size(); ++i)
                if (P.idle[i].first >= bytes && P.idle[i].first <= 2 * bytes && (best == SIZE_MAX || P.idle[i].first < P.idle[best].first))
                        best = i;
        if (best != SIZE_MAX) {
                *out = P.idle[best].second;
                P.idle_bytes -= P.idle[best].first;
                P.idle.erase(P.idle.begin() + (ptrdiff_t)best);
                return hipSuccess;
        }
        const size_t gran = bytes >= (8u << 20) ? (2u << 20) : (64u << 10);
        const size_t rounded = (bytes + gran - 1) & ~(gran - 1);
        static const bool dbg_pool = getenv("TRINITY_DEBUG_CREATE") != nullptr;
        if (dbg_pool) {
                std::string have;
                for (const auto &b : P.idle)
                        have += " " + std::to_string(b.first >> 20);
                fprintf(stderr, "[tri pool_alloc] COLD hipMalloc of %zu MB (idle buffers, MB:%s)\n", rounded >> 20, have.c_str());
        }
        hipError_t e = hipMalloc(out, rounded);
        if (e != hipSuccess && !P.idle.empty()) { // out of memory with idle buffers around: give them back and try again
                (void)hipGetLastError();
                for (auto &b : P.idle) {
                        P.size_of.erase(b.second);
                        hipFree(b.second);
                }
                P.idle.clear();
                P.idle_bytes = 0;
                e = hipMalloc(out, rounded);
        }
        if (e == hipSuccess)
                P.size_of[*out] = rounded;
        return e;
}
static void pool_free(tri_dev *dev, void *p) {
        if (!p)
                return;
        if (!dev) {
                hipFree(p);
                return;
        }
        DevLock g(dev->mu);
        auto &P = dev->pool;
        const auto it = P.size_of.find(p);
        if (it == P.size_of.end()) { // (below POOL_MIN_BYTES: never pooled)
                hipFree(p);
                return;
        }
        P.idle.emplace_back(it->second, p);
        P.idle_bytes += it->second;
        while (P.idle_bytes > POOL_IDLE_CAP) {
                size_t big = 0;
                for (size_t i = 1; i < P.idle.size(); ++i)
                        if (P.idle[i].first > P.idle[big].first)
                                big = i;
                P.idle_bytes -= P.idle[big].first;
                P.size_of.erase(P.idle[big].second);
                hipFree(P.idle[big].second);
                P.idle.erase(P.idle.begin() + (ptrdiff_t)big);
        }
}
// pinned host block of at least `bytes` (64-byte aligned: hipHostMalloc is page-aligned); *cap = its size
static uint8_t *pinned_alloc(tri_dev *dev, const size_t bytes, size_t *cap) {
        DevLock g(dev->mu);
        auto &I = dev->pinned_idle;
        size_t best = SIZE_MAX;
        for (size_t i = 0; i < I.size(); ++i)
                if (I[i].first >= bytes && I[i].first <= 4 * bytes + (1u << 20) && (best == SIZE_MAX || I[i].first < I[best].first))
                        best = i;
        if (best != SIZE_MAX) {
                void *p = I[best].second;
                *cap = I[best].first;
                I.erase(I.begin() + (ptrdiff_t)best);
                return static_cast<uint8_t *>(p);
        }
        const size_t rounded = (bytes + bytes / 4 + (256u << 10)) & ~(size_t)((64u << 10) - 1); // (a quarter of slack: the next batch of the same caller is about as large)
        void *p = nullptr;
        if (hipHostMalloc(&p, rounded, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                return nullptr;
        }
        *cap = rounded;
        return static_cast<uint8_t *>(p);
}
static void pinned_free(tri_dev *dev, void *p, const size_t cap) {
        if (!p)
                return;
        if (!dev) {
                hipHostFree(p);
                return;
        }
        DevLock g(dev->mu);
        if (dev->pinned_idle.size() >= PINNED_IDLE_MAX) { // full: the block idle for longest goes, the one just released stays — the list follows the
                hipHostFree(dev->pinned_idle.front().second); // caller's CURRENT batches (a list full of another workload's block sizes made every create
                dev->pinned_idle.erase(dev->pinned_idle.begin()); // of the next one a hipHostMalloc: bench.py's cfg2 loop after its cfg5 leg, 1.2 -> 4 ms a create)
        }
        dev->pinned_idle.emplace_back(cap, p);
}
static void event_put(tri_dev *dev, hipEvent_t e) {
        if (!e)
                return;
        DevLock g(dev->mu);
        dev->events_idle.push_back(e);
}
static hipError_t event_get(tri_dev *dev, hipEvent_t *e) {
        DevLock g(dev->mu);
        if (!dev->events_idle.empty()) {
                *e = dev->events_idle.back();
                dev->events_idle.pop_back();
                return hipSuccess;
        }
        return hipEventCreate(e);
}

// the uploaded segment: what the walk derived (HostIndex; the planner reads terms / blk_last / docbytes / hitbytes / win / the df order)
// plus its device copies
struct tri_index : HostIndex {
        tri_dev *dev = nullptr;
        uint8_t *d_index = nullptr, *d_hits = nullptr;
        uint32_t *d_blk_last = nullptr, *d_blk_off = nullptr, *d_win = nullptr;
        uint32_t *d_blk_hits = nullptr, *d_hdir = nullptr; // where a directory row's hits start — LUCENE + hits.data: hit ordinal within the term
                                                           // (+ hdir: the 128-hit blocks of hits.data); GOOGLE: byte offset into index[]
        // GOOGLE: the document deltas of every block re-laid out as one contiguous stream per term ([n][n-1 prefix varints] per
        // block, bytes exactly as in the chunk) with its own offset column.  In the chunk a block's deltas are followed by its
        // freqs and hits, so a DocumentsOnly scan of a head term drags ~3x the bytes it decodes through HBM; the matching kernels
        // (k_and_dense, k_and) read this stream instead.  Scoring and phrases keep reading the chunk itself.
        uint8_t *d_dstream = nullptr;
        uint32_t *d_blk_doff = nullptr;
        // LUCENE: one 16-byte record per directory row (a quarter of a 128-document block, or a run of the varbyte tail) with all a lane
        // needs to address the row's payload in ONE load: {offset of the deltas group (tail: of the pairs), exception index, header word of
        // the deltas group, header word of the freqs group}.  Exception index = where THIS quarter's exceptions sit in the two groups'
        // lists: e0_deltas | cnt_deltas << 8 | e0_freqs << 16 | cnt_freqs << 24 (a lane patches its quarter without scanning the other
        // three's).  Header word = the group's first payload word (width | nexc << 8 | excwidth << 16), or bit 31 | value for an
        // all-equal group (k_fused.hpp PfRegs)
        uint4 *d_blk_rec = nullptr;
        uint32_t *d_masked = nullptr; // bitmap over docIDs of the masked documents (nullptr: none); max_doc / 32 + 2 words
        DevTerm *d_terms = nullptr;
        // ---- the term planes live with the index (they are a function of its lists alone): row r = the planes of the term of df rank r, built
        //      by k_term_planes the first time a batch's run wants them and kept — a caller that compiles a batch per step does not decode the
        //      same head terms step after step.  Two regions, built BY NEED (dev_structs.hpp: PL_HI): d_pcache holds every row's PLANE 0 (plw
        //      words a row: all a DocumentsOnly batch reads), d_pcache_hi the rows' HIGH parts (nested planes 1 .. 3 + level words, PL_HI * plw
        //      words a row) — allocated by the first scored batch that reads planes, a row's part built when such a batch names the row.  pc_cap
        //      rows + one all-zero row (index pc_cap) in each region; pc_built[r]: bit 0 plane 0, bit 1 the high part, bit 2 the rank records — the build has been
        //      enqueued on the engine stream (every later kernel of the stream sees it).  Grown (rows moved on the upload stream) when a batch is
        //      planned with more eligible terms than it holds.
        uint32_t *d_pcache = nullptr, *d_pcache_hi = nullptr;
        uint32_t pc_cap = 0, pc_plw = 0;
        std::vector<uint8_t> pc_built;
        // ... and what k_phrase needs to find a head term's hits WITHOUT walking its blocks (round 5): per row a RANK DIRECTORY over plane 0 — a 64-byte record per docID group g of 256 documents at d_prank[(row * (plw / 8) + g) * 16]:
        // the posting index of the group's first document and the group's eight plane-0 words, filled by k_term_planes —, and per posting the hits locator and frequency entry
        // phrase_locate_block would compute (d_phs[hs_off[row] + posting]; k_term_hits; GOOGLE, lists of full blocks).  d_term_row[term] = its row once both are there
        uint32_t *d_prank = nullptr, *d_term_row = nullptr;
        unsigned long long *d_phs = nullptr;
        uint64_t *d_hs_off = nullptr;
        uint32_t *d_ph_pairs = nullptr;   // (term, row) pairs of the rows whose hits entries were built, appended run after run (a row is built once: 2 * pc_cap words)
        size_t ph_pairs_n = 0;
        std::vector<uint64_t> hs_off;     // [pc_cap + 1]: prefix sums of the rows' document counts (row = df rank)
        std::vector<uint8_t> ph_built;    // [pc_cap]: the row's hits entries have been enqueued
        std::vector<uint32_t> rank_term;  // [pc_cap]: the term of df rank r
        hipEvent_t ev_pc_ready = nullptr;                       // the last growth's move of the rows (upload stream): every run waits for it
        std::vector<std::pair<void *, hipEvent_t>> pc_retired; // outgrown row buffers and the engine-stream point their last readers precede
        ~tri_index() { // also runs when tri_index_upload fails half-way
                if (dev)
                        hipSetDevice(dev->device);
                hipFree(d_index);
                hipFree(d_hits);
                hipFree(d_blk_hits);
                hipFree(d_hdir);
                hipFree(d_dstream);
                hipFree(d_blk_doff);
                hipFree(d_blk_rec);
                hipFree(d_masked);
                hipFree(d_blk_last);
                hipFree(d_blk_off);
                hipFree(d_win);
                hipFree(d_terms);
                pool_free(dev, d_pcache); // (pooled: tri_batch_create grows it without a device-wide synchronisation)
                pool_free(dev, d_pcache_hi);
                pool_free(dev, d_prank);
                pool_free(dev, d_phs);
                pool_free(dev, d_hs_off);
                pool_free(dev, d_ph_pairs);
                hipFree(d_term_row);
                for (auto &r : pc_retired) {
                        pool_free(dev, r.first);
                        hipEventDestroy(r.second);
                }
                if (ev_pc_ready)
                        hipEventDestroy(ev_pc_ready);
                dev_release(dev);
        }
};

// a compiled batch: the plan (BatchPlan: the host block with every array the kernels read, laid out by the planner) and its device side —
// ONE arena that holds the block's copy followed by the batch's small device-only arrays, plus the large pooled buffers
struct tri_batch : BatchPlan {
        tri_index *ix = nullptr;
        tri_dev *dev = nullptr;
        uint32_t flags, topk;
        int similarity = TRI_SIM_BM25;
        size_t nq;
        size_t block_cap = 0;       // size of the pinned host block (BatchPlan::block) as the pool knows it
        uint8_t *d_arena = nullptr; // [copy of block][counts][ticket][qthr][part_counts][task_hits][task_pos_base] [zeroed at creation: qcounts, top_counts, top_docs, top_scores]
        DevQuery *d_plan = nullptr;
        DevTask *d_tasks = nullptr;
        uint32_t *d_sched = nullptr; // task indices, heaviest first: [0, n_dense) TASK_DENSE, then the TASK_CAND ones, then the one-pass kinds
        uint32_t *d_plane_terms = nullptr, *d_qplane = nullptr, *d_build = nullptr; // d_build: (term, row) pairs of the plane rows a run has to build first // d_qplane: parallel to d_qterms, the term's row or PL_NONE (nullptr: the batch has no planes)
        unsigned long long *d_qthr = nullptr; // k_planes: per query, the best k-th score any of its tasks has seen (cleared at every run)
        uint32_t *d_sparse = nullptr;      // k_planes: per resident workgroup, the lists of a task's decoded (non-plane) slots
        DevFused *d_fused = nullptr;
        // HIP events on the engine stream: start, after k_term_planes, k_and_dense, k_and, k_fused, k_planes, k_phrase, end (owned by the
        // batch: two batches in flight on one device keep their own timings); ev_up: the plan has arrived (upload stream)
        hipEvent_t ev0 = nullptr, ev_a = nullptr, ev_s = nullptr, ev_r = nullptr, ev_b = nullptr, ev_c = nullptr, ev_p = nullptr, ev1 = nullptr, ev_pl = nullptr, ev_k = nullptr, ev_up = nullptr, ev_t = nullptr; // (ev_s: after k_psets; ev_r: after k_probe; ev_t: after the tree kernels)
        // TASK_TREE (k_tree.hpp): one scratch block — [tree rows: a PL_PLANES-plane row per distinct term leaf][phrase rows: a plane per hidden phrase query]
        // [a match bitmap per tree query][per query and chunk: matches][(term, row) pairs for k_term_planes]
        uint32_t *d_tree_scratch = nullptr, *d_tree_rows = nullptr, *d_tree_prows = nullptr, *d_tree_qbits = nullptr, *d_tree_cc = nullptr, *d_tree_build = nullptr;
        double *d_tree_scores = nullptr; // scored top-K batches: the tree queries' score stream (topk == 0: d_all_scores holds it)
        uint32_t *d_score_order = nullptr; // AccumulatedScore: the tasks k_score runs, heaviest first by their match counts (k_score_order)
        uint32_t *d_scat_list = nullptr; // ... the scatter queries' first units (k_psets_prep_list)
        uint32_t *d_scat_off = nullptr, *d_scat_cnt = nullptr, *d_scat_docs = nullptr; // PSET_UNIT_SCATTER unions: per task its slice of the list k_psets_prep makes (k_psets.hpp)
        uint32_t scat_cap = 0;
        bool ran = false;
        bool planes_hi = false; // the batch reads the HIGH parts of its plane rows (k_planes, k_score's level words): its run builds them where they are missing
        uint32_t *d_qterms = nullptr;
        uint32_t *d_out = nullptr;
        uint32_t *d_counts = nullptr; // per task, indexed first_task + i in query order
        uint32_t *d_ticket = nullptr;
        uint32_t *d_rich_allow = nullptr; // default mode, batches that hold general trees: per match the reportable terms the tree sits on
        uint64_t *d_hashes = nullptr;
        uint64_t *d_qcounts = nullptr; // per caller query: matches of the last run (device copy for the result gather)
        // AccumulatedScoreScheme
        uint32_t *d_sterms = nullptr;
        double *d_sweights = nullptr;
        uint32_t *d_part_docs = nullptr, *d_part_counts = nullptr, *d_top_docs = nullptr, *d_top_counts = nullptr;
        double *d_part_scores = nullptr;
        float *d_top_scores = nullptr;
        double *d_all_scores = nullptr; // topk == 0: one double per out[] slot
        // TRI_FLAG_MATCHED_TERMS (k_rich.hpp): sterms[] holds every query's reportable terms; R = the widest query's count
        uint32_t *d_rich_present = nullptr, *d_task_hits = nullptr;
        uint16_t *d_rich_freq = nullptr, *d_rich_pool = nullptr;
        uint8_t *d_rich_plen = nullptr;     // TRI_FLAG_HIT_PAYLOADS: per hit of the pool, term_hit::payloadLen ...
        uint64_t *d_rich_payload = nullptr; // ... and term_hit::payload
        uint64_t *d_task_pos_base = nullptr;
        std::vector<uint64_t> h_task_pos_base; // per task; [ntasks] = the pool's size
        size_t rich_pool_cap = 0;
        // phrases
        DevPhrase *d_phrases = nullptr;
        uint32_t *d_pterms = nullptr, *d_ptasks = nullptr;
        double *d_pscore = nullptr; // per out[] slot: sum of the phrase scores of the match (scored mode)
        std::vector<uint32_t> h_counts;       // per task
        std::vector<uint64_t> h_query_counts; // per plan slot
        bool synced = false;
        tri_batch_info info{};
        ~tri_batch() { // also runs when tri_batch_create fails half-way: nothing allocated so far is leaked
                if (dev) {
                        hipSetDevice(dev->device);
                        if (ran && !synced) // (its large buffers go back to the device's pool: nothing of this batch may still be running on them)
                                hipStreamSynchronize(dev->stream);
                        if (ev_up)
                                hipEventSynchronize(ev_up); // (the pinned block goes back to the pool: its copy must have left)
                }
                for (hipEvent_t e : {ev0, ev_a, ev_s, ev_r, ev_b, ev_c, ev_p, ev1, ev_pl, ev_k, ev_up, ev_t})
                        if (e) {
                                if (dev) {
                                        DevLock g(dev->mu);
                                        dev->events_idle.push_back(e);
                                } else
                                        hipEventDestroy(e);
                        }
                pinned_free(dev, block, block_cap);
                pool_free(dev, d_arena);
                pool_free(dev, d_sparse);
                pool_free(dev, d_out);
                pool_free(dev, d_rich_allow);
                hipFree(d_hashes);
                pool_free(dev, d_part_docs);
                pool_free(dev, d_part_scores);
                pool_free(dev, d_all_scores);
                pool_free(dev, d_rich_present);
                pool_free(dev, d_rich_freq);
                hipFree(d_rich_pool);
                hipFree(d_rich_plen);
                hipFree(d_rich_payload);
                pool_free(dev, d_pscore);
                pool_free(dev, d_tree_scratch);
                pool_free(dev, d_tree_scores);
                pool_free(dev, d_scat_docs);
                dev_release(dev);
        }
};


#include "dev_stream.hpp"
#include "k_decode.hpp"
#include "k_match.hpp"
#include "k_score.hpp"
#include "k_fused.hpp"
#include "k_planes.hpp"
#include "k_psets.hpp"
#include "k_probe.hpp"
#include "k_encode.hpp"
#include "k_phrase.hpp"
#include "k_rich.hpp"
#include "k_tree.hpp"
#include "k_commit.hpp"
#include "k_lencode.hpp"

// launch the instantiation of a codec-templated kernel that matches the uploaded segment
#define TRI_LAUNCH(K, codec, grid, block, stream, ...)                                              \
        do {                                                                                        \
                if ((codec) == TRI_CODEC_LUCENE)                                                    \
                        hipLaunchKernelGGL(K<CODEC_LUCENE>, grid, block, 0, stream, __VA_ARGS__);   \
                else                                                                                \
                        hipLaunchKernelGGL(K<CODEC_GOOGLE>, grid, block, 0, stream, __VA_ARGS__);   \
        } while (0)

// ------------------------------------------------------------------------------------------ host: device
extern "C" int tri_dev_open(int device, tri_dev **out) {
        if (!out)
                return fail(TRI_ERR_INVALID, "tri_dev_open: null out");
        int n = 0;
        HIP_TRY(hipGetDeviceCount(&n));
        if (device < 0 || device >= n)
                return fail(TRI_ERR_INVALID, "tri_dev_open: device %d out of range (%d devices)", device, n);
        HIP_TRY(hipSetDevice(device));
        auto d = std::make_unique<tri_dev>();
        d->device = device;
        HIP_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&d->stream2, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&d->stream_up, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&d->stream_rb, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&d->ev_join, hipEventDisableTiming));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        d->cus = prop.multiProcessorCount;
        *out = d.release();
        return TRI_OK;
}

extern "C" void tri_dev_close(tri_dev *d) {
        if (!d)
                return;
        // indexes or batches of this handle still alive (a caller that closes first and destroys later): they keep using the handle's
        // streams and pools, and the last of them to go takes the handle along
        if (d->refs > 0) {
                d->closing = true;
                return;
        }
        dev_destroy(d);
}

namespace {
        uint64_t *option_slot(tri_options &o, const char *name) {
                static const struct {
                        const char *name;
                        uint64_t tri_options::*field;
                } table[] = {{"dense_min_postings", &tri_options::dense_min_postings}, {"dense_task_cost", &tri_options::dense_task_cost},
                             {"fused", &tri_options::fused},
                             {"fused_task_cost", &tri_options::fused_task_cost},
                             {"fused_freq_cap", &tri_options::fused_freq_cap},
                             {"fused_halfwords", &tri_options::fused_halfwords},
                             {"account_needed_bytes", &tri_options::account_needed_bytes},
                             {"overlap_dense_wgs", &tri_options::overlap_dense_wgs},
                             {"overlap_cand_wgs", &tri_options::overlap_cand_wgs},
                             {"overlap", &tri_options::overlap},
                             {"planes", &tri_options::planes},
                             {"plane_div", &tri_options::plane_div},
                             {"planes_split", &tri_options::planes_split},
                             {"plane_max_bytes", &tri_options::plane_max_bytes},
                             {"plane_amortize", &tri_options::plane_amortize},
                             {"planes_rebuild", &tri_options::planes_rebuild},
                             {"cand_xcd", &tri_options::cand_xcd},
                             {"plan_threads", &tri_options::plan_threads},
                             {"probe_max_blocks", &tri_options::probe_max_blocks}, {"phrase_task_div", &tri_options::phrase_task_div}, {"plan_hot_us", &tri_options::plan_hot_us}, {"plan_pin", &tri_options::plan_pin}, {"planes_order", &tri_options::planes_order}, {"pset_order", &tri_options::pset_order}, {"scatter_bitmap_slack", &tri_options::scatter_bitmap_slack}, {"tree_max_bytes", &tri_options::tree_max_bytes}, {"result_bitmaps", &tri_options::result_bitmaps}, {"cand_task_cost", &tri_options::cand_task_cost}, {"dense_window_cost", &tri_options::dense_window_cost}};
                for (const auto &e : table)
                        if (!strcmp(e.name, name))
                                return &(o.*(e.field));
                return nullptr;
        }
} // namespace

extern "C" int tri_dev_set_option(tri_dev *d, const char *name, uint64_t value) {
        if (!d || !name)
                return fail(TRI_ERR_INVALID, "tri_dev_set_option: null argument");
        uint64_t *slot = option_slot(d->opt, name);
        if (!slot)
                return fail(TRI_ERR_INVALID, "tri_dev_set_option: unknown option '%s'", name);
        *slot = value;
        return TRI_OK;
}

extern "C" int tri_dev_get_option(tri_dev *d, const char *name, uint64_t *value) {
        if (!d || !name || !value)
                return fail(TRI_ERR_INVALID, "tri_dev_get_option: null argument");
        const uint64_t *slot = option_slot(d->opt, name);
        if (!slot)
                return fail(TRI_ERR_INVALID, "tri_dev_get_option: unknown option '%s'", name);
        *value = *slot;
        return TRI_OK;
}

extern "C" int tri_dev_memory(tri_dev *d, tri_dev_memory_info *out) {
        if (!d || !out)
                return fail(TRI_ERR_INVALID, "tri_dev_memory: null argument");
        HIP_TRY(hipSetDevice(d->device));
        size_t fr = 0, total = 0;
        HIP_TRY(hipMemGetInfo(&fr, &total));
        DevLock g(d->mu);
        uint64_t all = 0, pinned = 0;
        for (const auto &kv : d->pool.size_of)
                all += kv.second;
        for (const auto &pb : d->pinned_idle)
                pinned += pb.first;
        out->pool_idle_bytes = d->pool.idle_bytes;
        out->pool_in_use_bytes = all - d->pool.idle_bytes;
        out->pinned_idle_bytes = pinned;
        out->device_free_bytes = fr;
        out->device_total_bytes = total;
        return TRI_OK;
}

extern "C" int tri_dev_sync(tri_dev *d) {
        if (!d)
                return fail(TRI_ERR_INVALID, "null dev");
        HIP_TRY(hipStreamSynchronize(d->stream));
        return TRI_OK;
}

extern "C" void *tri_dev_stream(tri_dev *d) { return d ? (void *)d->stream : nullptr; }

// ------------------------------------------------------------------------------------------ host: upload
namespace {
        template <class T>
        int dev_upload(T **dst, const std::vector<T> &src, size_t extra = 0) {
                HIP_TRY(hipMalloc((void **)dst, (src.size() + extra) * sizeof(T) + 16));
                if (!src.empty())
                        HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
                return TRI_OK;
        }
} // namespace

extern "C" int tri_index_upload(tri_dev *dev, const uint8_t *index, size_t len, const uint8_t *hits, size_t hits_len, int codec,
                                const tri_term *terms, size_t nterms, uint32_t docs_cnt, tri_index **out) {
        if (!dev || !out || (!index && len) || (!terms && nterms) || (!hits && hits_len))
                return fail(TRI_ERR_INVALID, "tri_index_upload: null argument");
        HIP_TRY(hipSetDevice(dev->device));
        auto ix = std::make_unique<tri_index>();
        // One pass over every chunk on the host (index_host.hpp): hop block headers (google_codec.cpp:641-697), validate, record the
        // directory, the delta streams / row records, the docID-cell index and the algorithmic byte split of SURVEY §8(d)
        {
                std::string err;
                if (const int rc = build_host_index(index, len, hits, hits_len, codec, terms, nterms, docs_cnt, *ix, err))
                        return fail(rc, "%s", err.c_str());
        }
        ix->dev = dev;
        dev_retain(dev);
        if (!ix->dev_index.empty()) { // a LUCENE segment with FastPFor<4> payload words: the device gets its PFOR128 transcription (index_host.hpp)
                index = ix->dev_index.data();
                len = ix->dev_index.size();
                hits = ix->dev_hits.empty() ? nullptr : ix->dev_hits.data();
                hits_len = ix->dev_hits.size();
        }
        // device copies
        int rc;
        if ((rc = dev_upload(&ix->d_win, ix->win, 3 * CELLS_PER_SPAN + 8))) // (lanes of terms without a row read entries 0, CELLS_PER_SPAN, 2 * CELLS_PER_SPAN and drop them)
                return rc;
        HIP_TRY(hipMemset(ix->d_win + ix->win.size(), 0, (3 * CELLS_PER_SPAN + 8) * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void **)&ix->d_index, len + 256)); // over-read slack: the byte streams keep several qwords in flight past the cursor
        HIP_TRY(hipMemset(ix->d_index, 0, len + 256));
        if (len)
                HIP_TRY(hipMemcpy(ix->d_index, index, len, hipMemcpyHostToDevice));
        if ((rc = dev_upload(&ix->d_blk_last, ix->blk_last)) || (rc = dev_upload(&ix->d_blk_off, ix->blk_off)) || (rc = dev_upload(&ix->d_terms, ix->terms)))
                return rc;
        if (codec == TRI_CODEC_GOOGLE) {
                ix->blk_doff.push_back((uint32_t)ix->dstream.size() + 1); // (sentinel: block b's delta bytes = blk_doff[b + 1] - blk_doff[b] - 1 for the last block too)
                ix->dstream.resize(ix->dstream.size() + 256, 0); // over-read slack, like index[]
                if ((rc = dev_upload(&ix->d_dstream, ix->dstream)) || (rc = dev_upload(&ix->d_blk_doff, ix->blk_doff)))
                        return rc;
        }
        if (hits_len) { // LUCENE: hits.data (positions) resident next to the index
                HIP_TRY(hipMalloc((void **)&ix->d_hits, hits_len + 256));
                HIP_TRY(hipMemset(ix->d_hits, 0, hits_len + 256));
                HIP_TRY(hipMemcpy(ix->d_hits, hits, hits_len, hipMemcpyHostToDevice));
                if (ix->has_hdir && ((rc = dev_upload(&ix->d_blk_hits, ix->blk_hits)) || (rc = dev_upload(&ix->d_hdir, ix->hdir))))
                        return rc;
        }
        if (codec == TRI_CODEC_GOOGLE && (rc = dev_upload(&ix->d_blk_hits, ix->blk_hits)))
                return rc;
        if (codec == TRI_CODEC_LUCENE) {
                static_assert(sizeof(RowRec) == sizeof(uint4), "row records are read as uint4");
                HIP_TRY(hipMalloc((void **)&ix->d_blk_rec, ix->blk_rec.size() * sizeof(uint4) + 16));
                if (!ix->blk_rec.empty())
                        HIP_TRY(hipMemcpy(ix->d_blk_rec, ix->blk_rec.data(), ix->blk_rec.size() * sizeof(uint4), hipMemcpyHostToDevice));
        }
        ix->release_device_columns(); // (the host keeps what the planner reads: terms, blk_last, docbytes / hitbytes, win, the df order)
        *out = ix.release();
        return TRI_OK;
}


extern "C" int tri_index_set_masked(tri_index *ix, const uint32_t *docids, size_t n) {
        if (!ix || (!docids && n))
                return fail(TRI_ERR_INVALID, "tri_index_set_masked: null argument");
        HIP_TRY(hipSetDevice(ix->dev->device));
        HIP_TRY(hipStreamSynchronize(ix->dev->stream)); // no batch of this device is reading the old bitmap
        if (!n) {
                hipFree(ix->d_masked);
                ix->d_masked = nullptr;
                return TRI_OK;
        }
        // whole docID windows (the bitmap kernel reads a window's worth of words at a time), one spare window
        const size_t words = ((size_t)ix->max_doc / SPAN_BITS + 2) * (SPAN_BITS / 32);
        std::vector<uint32_t> bm(words, 0);
        for (size_t i = 0; i < n; ++i)
                if (docids[i] <= ix->max_doc) // a document this segment does not hold cannot match anyway
                        bm[docids[i] >> 5] |= 1u << (docids[i] & 31);
        if (!ix->d_masked)
                HIP_TRY(hipMalloc((void **)&ix->d_masked, words * 4));
        HIP_TRY(hipMemcpy(ix->d_masked, bm.data(), words * 4, hipMemcpyHostToDevice));
        return TRI_OK;
}

extern "C" void tri_index_destroy(tri_index *ix) {
        delete ix; // ~tri_index releases the device buffers
}

extern "C" int tri_index_get_info(const tri_index *ix, tri_index_info *info) {
        if (!ix || !info)
                return fail(TRI_ERR_INVALID, "null argument");
        *info = ix->info;
        return TRI_OK;
}

extern "C" int tri_index_term_docbytes(const tri_index *ix, const uint32_t *terms, size_t n, uint64_t *out) {
        if (!ix || (!terms && n) || (!out && n))
                return fail(TRI_ERR_INVALID, "null argument");
        for (size_t i = 0; i < n; ++i)
                out[i] = terms[i] < ix->docbytes.size() ? ix->docbytes[terms[i]] : 0;
        return TRI_OK;
}

// ------------------------------------------------------------------------------------------ host: decode
extern "C" int tri_decode_terms(tri_index *ix, const uint32_t *terms, size_t n, uint32_t *docs, uint32_t *freqs, uint64_t *out_offsets) {
        if (!ix || (!terms && n) || !out_offsets)
                return fail(TRI_ERR_INVALID, "null argument");
        tri_dev *dev = ix->dev;
        HIP_TRY(hipSetDevice(dev->device));
        std::vector<DecodeJob> jobs(n);
        uint64_t tot = 0, padded = 0;
        uint32_t maxblocks = 0;
        for (size_t i = 0; i < n; ++i) {
                if (terms[i] >= ix->terms.size())
                        return fail(TRI_ERR_INVALID, "term %u out of range", terms[i]);
                const DevTerm &t = ix->terms[terms[i]];
                jobs[i] = {terms[i], 0, padded};
                out_offsets[i] = tot;
                tot += t.documents;
                padded += (uint64_t)t.nblocks * 32;
                maxblocks = std::max(maxblocks, t.nblocks);
        }
        out_offsets[n] = tot;
        if (!tot || !docs)
                return TRI_OK;
        DecodeJob *d_jobs = nullptr;
        uint32_t *d_docs = nullptr, *d_freqs = nullptr;
        HIP_TRY(hipMalloc((void **)&d_jobs, n * sizeof(DecodeJob)));
        HIP_TRY(hipMemcpyAsync(d_jobs, jobs.data(), n * sizeof(DecodeJob), hipMemcpyHostToDevice, dev->stream));
        HIP_TRY(hipMalloc((void **)&d_docs, padded * 4));
        if (freqs)
                HIP_TRY(hipMalloc((void **)&d_freqs, padded * 4));
        dim3 grid(std::min<uint32_t>((maxblocks + 255) / 256, 4096), (uint32_t)n);
        TRI_LAUNCH(k_decode_terms, ix->codec, grid, dim3(256), dev->stream, ix->d_index, ix->d_blk_last, ix->d_blk_off, ix->d_terms, d_jobs, d_docs,
                           d_freqs);
        HIP_TRY(hipGetLastError());
        // blocks are full (32) except the last one of each term: the padded layout is dense per term
        for (size_t i = 0; i < n; ++i) {
                const DevTerm &t = ix->terms[terms[i]];
                if (!t.documents)
                        continue;
                HIP_TRY(hipMemcpyAsync(docs + out_offsets[i], d_docs + jobs[i].out_off, (size_t)t.documents * 4, hipMemcpyDeviceToHost, dev->stream));
                if (freqs)
                        HIP_TRY(hipMemcpyAsync(freqs + out_offsets[i], d_freqs + jobs[i].out_off, (size_t)t.documents * 4, hipMemcpyDeviceToHost, dev->stream));
        }
        HIP_TRY(hipStreamSynchronize(dev->stream));
        hipFree(d_jobs);
        hipFree(d_docs);
        hipFree(d_freqs);
        return TRI_OK;
}

// ------------------------------------------------------------------------------------------ host: batches
// tri_batch_create = the host planner (planner.hpp: lowering, execution classes, tasks, term planes, schedule — on the device handle's
// host threads) + the plan's way to the device: ONE pinned block, ONE arena, ONE copy on the upload stream.  Everything a steady caller
// needs per batch comes from the handle's pools (arena, pinned block, output regions, events): no hipMalloc, no hipFree, no device
// synchronisation — the next batch is compiled and uploaded while the current one runs.
extern "C" int tri_batch_create(tri_index *ix, const uint32_t *prog, size_t prog_len, const tri_query *queries, size_t nq, const double *weights,
                                uint32_t flags, uint32_t topk, int similarity, tri_batch **out) {
        if (!ix || !out || (!prog && prog_len) || (!queries && nq))
                return fail(TRI_ERR_INVALID, "tri_batch_create: null argument");
        const uint32_t mode = flags & (TRI_FLAG_DOCUMENTS_ONLY | TRI_FLAG_ACCUMULATED_SCORE | TRI_FLAG_MATCHED_TERMS);
        if (mode != TRI_FLAG_DOCUMENTS_ONLY && mode != TRI_FLAG_ACCUMULATED_SCORE && mode != TRI_FLAG_MATCHED_TERMS)
                return fail(TRI_ERR_INVALID, "exactly one of DocumentsOnly, AccumulatedScoreScheme, MatchedTerms (exec_query's default mode): the modes are mutually exclusive (exec.h:45-48)");
        const bool scored = mode == TRI_FLAG_ACCUMULATED_SCORE;
        const bool rich = mode == TRI_FLAG_MATCHED_TERMS;
        if ((flags & TRI_FLAG_HIT_PAYLOADS) && !rich)
                return fail(TRI_ERR_INVALID, "TRI_FLAG_HIT_PAYLOADS goes with TRI_FLAG_MATCHED_TERMS (the mode that delivers hits)");
        if (scored && topk > TOPK_MAX)
                return fail(TRI_ERR_INVALID, "AccumulatedScoreScheme: topk <= %u (0 = keep every match's score instead of a top-K)", TOPK_MAX);
        if (similarity != TRI_SIM_BM25 && similarity != TRI_SIM_TFIDF && similarity != TRI_SIM_TRIVIAL)
                return fail(TRI_ERR_INVALID, "unknown similarity %d", similarity);
        tri_dev *dev = ix->dev;
        HIP_TRY(hipSetDevice(dev->device));
        const auto t_create = std::chrono::steady_clock::now();
        auto b = std::make_unique<tri_batch>();
        b->ix = ix;
        b->dev = dev;
        dev_retain(dev);
        b->flags = flags;
        b->topk = topk;
        b->similarity = similarity;
        b->nq = nq;
        // ---- plan (host)
        PlanEnv env;
        env.opt = dev->opt;
        env.cus = (uint32_t)dev->cus;
        env.fus_wgs_per_cu = FUS_WGS_PER_CU;
        env.plk_wgs_per_cu = PLK_WGS_PER_CU;
        PlanInput in;
        in.prog = prog;
        in.prog_len = prog_len;
        in.queries = queries;
        in.nq = nq;
        in.weights = weights;
        in.flags = flags;
        in.topk = topk;
        in.similarity = similarity;
        {
                std::string err;
                int rc;
                try {
                        // a free planner context (the first one when both are busy: creates from more threads than contexts plan one after the other)
                        unsigned which = 0;
                        std::unique_lock<std::mutex> plan_lock(dev->planners[0].mu, std::try_to_lock);
                        for (unsigned k = 1; k < tri_dev::PLAN_CTXS && !plan_lock.owns_lock(); ++k) {
                                plan_lock = std::unique_lock<std::mutex>(dev->planners[k].mu, std::try_to_lock);
                                which = k;
                        }
                        if (!plan_lock.owns_lock()) {
                                which = 0;
                                plan_lock = std::unique_lock<std::mutex>(dev->planners[0].mu);
                        }
                        tri_dev::PlanCtx &pc = dev->planners[which];
                        if (nq >= 1024 && !pc.pool) { // (batches below a thousand queries are planned on the calling thread)
                                // (default: the handle's contexts share what the process may use — the affinity mask, capped by the cgroup's CPU quota — less six CPUs: the two
                                //  compiling callers, the thread that runs and awaits batches, the runtime's own threads, and slack — a process that uses its whole quota is
                                //  throttled at the first neighbour's burst: host_pool.hpp.  Under the GPU box's 16-CPU quota: 6 threads a context; measured there, 600 steps of cfg2
                                //  with the kernels at 1.18 ms, 5 / 7 threads x 2 compilers: 1.185 ms per step either way, 8 x 1: 1.39 - 1.49 (the planner is the bound), 12 x 1: 1.18;
                                //  at the round's end, kernels 1.11 ms, 100 steps, threads a context -> ms per step, a create's median / longest: 4 -> 1.29, 2.65 / 41;
                                //  5 -> 1.121, 2.0 - 2.2 / 2.6 - 39; 6 -> 1.114, 1.58 / 1.97; 7 -> 1.117, 1.59 / 1.88; 8 -> 1.127, 1.56 / 44 — six: four CPUs of the quota stay free)
                                const unsigned budget = host_cpu_budget();
                                const unsigned fair = budget >= 10 ? std::min(16u, (budget - 4) / tri_dev::PLAN_CTXS) : budget >= 4 ? 2u : 1u;
                                const unsigned want = dev->opt.plan_threads ? (unsigned)std::min<uint64_t>(dev->opt.plan_threads, 64) : fair;
                                if (want > 1) {
                                        try {
                                                // (every pool of the handle counts its stretch of CPUs from the FIRST pool's anchor, not from its own creator's CPU;
                                                //  under the handle's lock: two contexts starting their pools at once agree on it)
                                                DevLock g(dev->mu);
                                                pc.pool = std::make_unique<HostPool>(want, dev->opt.plan_pin != 0, (unsigned)dev->opt.plan_hot_us, which, tri_dev::PLAN_CTXS, dev->plan_anchor, dev->opt.plan_pin == 2);
                                                if (dev->plan_anchor < 0)
                                                        dev->plan_anchor = pc.pool->anchor();
                                        } catch (...) { // (no threads to be had: the calling thread plans alone)
                                        }
                                }
                        }
                        rc = plan_batch(*ix, env, in, pc.pool.get(), [&](size_t bytes) { return pinned_alloc(dev, bytes, &b->block_cap); }, *b, err, &pc.frag_cache);
                } catch (const std::bad_alloc &) {
                        return fail(TRI_ERR_NOMEM, "tri_batch_create: out of host memory");
                }
                if (rc != TRI_OK)
                        return fail(rc, "%s", err.c_str());
                if (!b->last_unsupported.empty())
                        fail(TRI_ERR_UNSUPPORTED, "%s", b->last_unsupported.c_str()); // (tri_last_error() describes the last query that was left out; the call succeeds)
        }
        const size_t nt = b->tasks.size(), np = b->plan.size();
        const uint64_t off = b->out_capacity;
        static const bool dbg_create = getenv("TRINITY_DEBUG_CREATE") != nullptr; // (stderr: where a create's time goes past the planner)
        double dbg_t[6] = {0, 0, 0, 0, 0, 0};
        auto dbg_lap = [&](int i) {
                if (dbg_create)
                        dbg_t[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create).count();
        };
        dbg_lap(0);
        DevLock dev_lock(dev->mu); // (from here on: the device's pools, the index's plane cache, the streams — see tri_dev::mu)
        // ---- the arena: the block's copy, then the batch's small device-only arrays; the part that must start out zero comes last
        size_t a = b->block_bytes;
        auto carve = [&](size_t bytes) {
                const size_t at = a;
                a += (bytes + 255) & ~(size_t)255;
                return at;
        };
        const size_t a_counts = carve((nt + 1) * 4), a_ticket = carve(TICKET_BYTES), a_build = carve((b->plane_terms.size() + 1) * 8);
        const bool planes_tasks = b->n_planes + b->n_planes8;
        const size_t a_qthr = planes_tasks ? carve((np + 1) * 8) : 0;
        const size_t a_part_counts = scored ? carve((nt + 1) * 4) : 0, a_score_order = scored ? carve((nt + 1) * 4) : 0;
        const size_t a_task_hits = rich ? carve((nt + 1) * 4) : 0, a_task_pos = rich ? carve((nt + 1) * 8) : 0;
        const size_t a_scat_off = b->pscatter_queries ? carve((nt + 1) * 4) : 0, a_scat_cnt = b->pscatter_queries ? carve((nt + 1) * 4) : 0;
        const size_t a_scat_list = b->pscatter_queries ? carve((b->pscatter_queries + 1) * 4) : 0;
        const size_t a_zero = a;
        const size_t a_qcounts = carve((nq + 1) * 8); // queries that can never match keep count 0 ...
        const size_t a_top_counts = scored ? carve((nq + 1) * 4) : 0;
        const size_t a_top_docs = scored ? carve((nq * topk + 1) * 4) : 0; // ... and zeroed rows (k_topk_merge only writes the rows of queries that have a plan slot;
        const size_t a_top_scores = scored ? carve((nq * topk + 1) * 4) : 0; // the blocks travel whole to the host and to the other ranks)
        HIP_TRY(pool_alloc(dev, (void **)&b->d_arena, a + 256));
        uint8_t *const A = b->d_arena;
        b->d_plan = (DevQuery *)(A + b->off_plan);
        b->d_qterms = (uint32_t *)(A + b->off_qterms);
        b->d_tasks = (DevTask *)(A + b->off_tasks);
        b->d_sched = (uint32_t *)(A + b->off_sched);
        b->d_fused = (DevFused *)(A + b->off_fused);
        b->d_qplane = b->qplane.empty() ? nullptr : (uint32_t *)(A + b->off_qplane);
        b->d_plane_terms = (uint32_t *)(A + b->off_plane_terms);
        b->d_sterms = (uint32_t *)(A + b->off_sterms);
        b->d_sweights = (double *)(A + b->off_sweights);
        b->d_phrases = (DevPhrase *)(A + b->off_phrases);
        b->d_pterms = (uint32_t *)(A + b->off_pterms);
        b->d_ptasks = (uint32_t *)(A + b->off_ptasks);
        b->d_counts = (uint32_t *)(A + a_counts);
        b->d_ticket = (uint32_t *)(A + a_ticket);
        b->d_build = (uint32_t *)(A + a_build);
        b->d_qthr = planes_tasks ? (unsigned long long *)(A + a_qthr) : nullptr;
        b->d_part_counts = scored ? (uint32_t *)(A + a_part_counts) : nullptr;
        b->d_score_order = scored ? (uint32_t *)(A + a_score_order) : nullptr;
        b->d_scat_off = b->pscatter_queries ? (uint32_t *)(A + a_scat_off) : nullptr;
        b->d_scat_cnt = b->pscatter_queries ? (uint32_t *)(A + a_scat_cnt) : nullptr;
        b->d_scat_list = b->pscatter_queries ? (uint32_t *)(A + a_scat_list) : nullptr;
        if (b->pscatter_queries) {
                if (b->pscatter_docs + 64 > 0xffffffffull)
                        return fail(TRI_ERR_UNSUPPORTED, "tri_batch_create: the unions' terms without a plane hold more than 2^32 documents");
                b->scat_cap = (uint32_t)b->pscatter_docs;
                HIP_TRY(pool_alloc(dev, (void **)&b->d_scat_docs, ((size_t)b->scat_cap + 64) * 4));
        }
        b->d_task_hits = rich ? (uint32_t *)(A + a_task_hits) : nullptr;
        b->d_task_pos_base = rich ? (uint64_t *)(A + a_task_pos) : nullptr;
        b->d_qcounts = (uint64_t *)(A + a_qcounts);
        b->d_top_counts = scored ? (uint32_t *)(A + a_top_counts) : nullptr;
        b->d_top_docs = scored ? (uint32_t *)(A + a_top_docs) : nullptr;
        b->d_top_scores = scored ? (float *)(A + a_top_scores) : nullptr;
        for (hipEvent_t *e : {&b->ev0, &b->ev_a, &b->ev_s, &b->ev_r, &b->ev_b, &b->ev_c, &b->ev_p, &b->ev1, &b->ev_pl, &b->ev_k, &b->ev_up, &b->ev_t})
                HIP_TRY(event_get(dev, e));
        if (b->block_bytes)
                HIP_TRY(hipMemcpyAsync(A, b->block, b->block_bytes, hipMemcpyHostToDevice, dev->stream_up));
        HIP_TRY(hipMemsetAsync(A + a_zero, 0, a - a_zero, dev->stream_up));
        dbg_lap(1);
        // ---- the large buffers (the device handle's pool)
        if (!b->plane_terms.empty() || planes_tasks) {
                // the index's plane cache holds a row for every term this batch could name (row = df rank < plane_rows), plus an all-zero row:
                // what a k_planes slot WITHOUT term planes reads (so its sweep needs no select)
                const uint32_t want = std::max<uint32_t>(1, b->plane_rows);
                // (rows whose readers are through go back to the pool: no call here waits for the device)
                for (size_t i = 0; i < ix->pc_retired.size();)
                        if (hipEventQuery(ix->pc_retired[i].second) == hipSuccess) {
                                pool_free(dev, ix->pc_retired[i].first);
                                event_put(dev, ix->pc_retired[i].second);
                                ix->pc_retired.erase(ix->pc_retired.begin() + (long)i);
                        } else
                                ++i;
                // (does this batch read the rows' HIGH parts — a scored batch whose one-pass kernel or scorers read planes?)
                b->planes_hi = planes_tasks || !b->splane.empty();
                if (want > ix->pc_cap || b->plw != ix->pc_plw || (b->planes_hi && !ix->d_pcache_hi)) {
                        // Grown WITHOUT draining the engine stream (round 4 synchronised it here: a stall in the serving loop whenever a batch was planned with more
                        // eligible terms): the rows move on the upload stream behind everything the engine stream holds so far (runs that read or build the old
                        // rows), later runs wait for the move's event (tri_batch_run), and the old buffer is retired — pooled again once that point has passed
                        const size_t row = (size_t)b->plw * 4, row_hi = (size_t)PL_HI * b->plw * 4;
                        const bool resize = want > ix->pc_cap || b->plw != ix->pc_plw; // (else: only the high region is new)
                        const uint32_t cap = std::max(want, b->plw == ix->pc_plw ? ix->pc_cap : 0u);
                        const bool with_hi = b->planes_hi || ix->d_pcache_hi;
                        uint32_t *fresh = nullptr, *fresh_hi = nullptr;
                        if (resize)
                                HIP_TRY(pool_alloc(dev, (void **)&fresh, ((size_t)cap + 1) * row + 64));
                        if (with_hi && (resize || !ix->d_pcache_hi))
                                HIP_TRY(pool_alloc(dev, (void **)&fresh_hi, ((size_t)cap + 1) * row_hi + 64));
                        hipEvent_t drained = nullptr;
                        std::vector<void *> outgrown; // the buffers this growth replaces
                        HIP_TRY(event_get(dev, &drained));
                        if (!ix->ev_pc_ready)
                                HIP_TRY(event_get(dev, &ix->ev_pc_ready));
                        HIP_TRY(hipEventRecord(drained, dev->stream));
                        HIP_TRY(hipStreamWaitEvent(dev->stream_up, drained, 0));
                        const bool same_plw = ix->d_pcache && b->plw == ix->pc_plw;
                        if (fresh) {
                                if (same_plw)
                                        HIP_TRY(hipMemcpyAsync(fresh, ix->d_pcache, (size_t)ix->pc_cap * row, hipMemcpyDeviceToDevice, dev->stream_up));
                                else
                                        ix->pc_built.clear();
                                HIP_TRY(hipMemsetAsync((uint8_t *)fresh + (size_t)cap * row, 0, row + 64, dev->stream_up));
                        }
                        if (fresh_hi) {
                                if (same_plw && ix->d_pcache_hi)
                                        HIP_TRY(hipMemcpyAsync(fresh_hi, ix->d_pcache_hi, (size_t)ix->pc_cap * row_hi, hipMemcpyDeviceToDevice, dev->stream_up));
                                else
                                        for (auto &bb : ix->pc_built)
                                                bb &= (uint8_t)~2u; // (no row has its high part yet)
                                HIP_TRY(hipMemsetAsync((uint8_t *)fresh_hi + (size_t)cap * row_hi, 0, row_hi + 64, dev->stream_up));
                        }
                        if (resize) {
                                // the rank directories and hits entries of the rows (k_phrase's rank path): sized for every row the cache can hold, moved like the rows
                                const bool same = same_plw;
                                std::vector<uint64_t> hs(cap + 1, 0);
                                std::vector<uint32_t> rt(cap, 0xffffffffu);
                                for (size_t t = 0; t < ix->terms.size(); ++t)
                                        if (ix->df_rank[t] < cap)
                                                rt[ix->df_rank[t]] = (uint32_t)t;
                                for (uint32_t r = 0; r < cap; ++r)
                                        hs[r + 1] = hs[r] + (rt[r] != 0xffffffffu ? ix->terms[rt[r]].documents : 0u);
                                uint32_t *prank = nullptr;
                                unsigned long long *phs = nullptr;
                                uint64_t *hso = nullptr;
                                uint32_t *pairs = nullptr;
                                const size_t groups = b->plw / 8;
                                HIP_TRY(pool_alloc(dev, (void **)&pairs, (size_t)cap * 8 + POOL_MIN_BYTES));
                                HIP_TRY(pool_alloc(dev, (void **)&prank, (size_t)cap * groups * PL_RANK_WORDS * 4 + 64));
                                HIP_TRY(pool_alloc(dev, (void **)&phs, (hs[cap] + 8) * 8));
                                HIP_TRY(pool_alloc(dev, (void **)&hso, ((size_t)cap + 1) * 8 + POOL_MIN_BYTES));
                                if (same && ix->d_prank) {
                                        HIP_TRY(hipMemcpyAsync(pairs, ix->d_ph_pairs, ix->ph_pairs_n * 8, hipMemcpyDeviceToDevice, dev->stream_up));
                                        HIP_TRY(hipMemcpyAsync(prank, ix->d_prank, (size_t)ix->pc_cap * groups * PL_RANK_WORDS * 4, hipMemcpyDeviceToDevice, dev->stream_up));
                                        HIP_TRY(hipMemcpyAsync(phs, ix->d_phs, ix->hs_off[ix->pc_cap] * 8, hipMemcpyDeviceToDevice, dev->stream_up));
                                } else {
                                        ix->ph_built.clear();
                                        ix->ph_pairs_n = 0;
                                }
                                ix->hs_off = hs; // (a row's offset depends on the rows before it alone: what was built stays where it was)
                                HIP_TRY(hipMemcpyAsync(hso, ix->hs_off.data(), ((size_t)cap + 1) * 8, hipMemcpyHostToDevice, dev->stream_up)); // (hs_off outlives the copy: a member)
                                if (!ix->d_term_row) {
                                        HIP_TRY(hipMalloc((void **)&ix->d_term_row, (ix->terms.size() + 1) * 4));
                                        HIP_TRY(hipMemsetAsync(ix->d_term_row, 0xff, (ix->terms.size() + 1) * 4, dev->stream_up));
                                } else if (!same)
                                        HIP_TRY(hipMemsetAsync(ix->d_term_row, 0xff, (ix->terms.size() + 1) * 4, dev->stream_up));
                                if (ix->d_prank) // (retired with the rows: same readers, and the copies above read them)
                                        for (void *old : {(void *)ix->d_prank, (void *)ix->d_phs, (void *)ix->d_hs_off, (void *)ix->d_ph_pairs})
                                                outgrown.push_back(old);
                                ix->d_prank = prank, ix->d_phs = phs, ix->d_hs_off = hso, ix->d_ph_pairs = pairs;
                                ix->rank_term = rt;
                                ix->ph_built.resize(cap, 0);
                        }
                        HIP_TRY(hipEventRecord(ix->ev_pc_ready, dev->stream_up));
                        // The outgrown buffers are retired on events recorded on the UPLOAD stream, behind the copies that read them: that point is past
                        // the engine stream's earlier readers too (stream_up waited for `drained`).  (Round 5 retired them on events of the engine
                        // stream recorded BEFORE the copies were enqueued: the next tri_batch_create could pool a buffer the copy had not read yet.)
                        if (fresh && ix->d_pcache)
                                outgrown.push_back(ix->d_pcache);
                        if (fresh_hi && ix->d_pcache_hi)
                                outgrown.push_back(ix->d_pcache_hi);
                        for (void *old : outgrown) {
                                hipEvent_t e2 = nullptr;
                                HIP_TRY(event_get(dev, &e2));
                                HIP_TRY(hipEventRecord(e2, dev->stream_up));
                                ix->pc_retired.emplace_back(old, e2);
                        }
                        event_put(dev, drained);
                        if (fresh)
                                ix->d_pcache = fresh;
                        if (fresh_hi)
                                ix->d_pcache_hi = fresh_hi;
                        ix->pc_cap = cap;
                        ix->pc_plw = b->plw;
                        ix->pc_built.resize(cap, 0);
                }
        }
        if (planes_tasks) {
                const uint64_t wgs = std::min<uint64_t>(std::max(b->n_planes, b->n_planes8), (uint64_t)dev->cus * PLK_WGS_PER_CU);
                HIP_TRY(pool_alloc(dev, (void **)&b->d_sparse, (2 * wgs * b->sparse_cap + 64) * 4)); // (per workgroup: the lists' entries, then their frequencies)
        }
        dbg_lap(2);
        HIP_TRY(pool_alloc(dev, (void **)&b->d_out, (off + 64) * 4));
        dbg_lap(3);
        if (!b->phrases.empty() && scored) {
                HIP_TRY(pool_alloc(dev, (void **)&b->d_pscore, (off + 64) * 8)); // (no clearing: k_phrase writes the entry of every match it keeps, and only
                                                                                  //  the matches of queries that hold a phrase are ever read — k_score, k_tree_leaves)
        }
        if (rich) {
                b->rich_R = std::max<uint32_t>(b->rich_R, 1);
                HIP_TRY(pool_alloc(dev, (void **)&b->d_rich_present, (off + 64) * 4));
                HIP_TRY(pool_alloc(dev, (void **)&b->d_rich_freq, (off + 64) * 2 * b->rich_R));
                if (b->rich_allow) {
                        HIP_TRY(pool_alloc(dev, (void **)&b->d_rich_allow, (off + 64) * 4));
                        HIP_TRY(hipMemsetAsync(b->d_rich_allow, 0xff, (off + 64) * 4, dev->stream_up)); // (every other query's matches: all terms allowed)
                }
        }
        if (scored) {
                if (!topk)
                        HIP_TRY(pool_alloc(dev, (void **)&b->d_all_scores, (off + 64) * 8));
                HIP_TRY(pool_alloc(dev, (void **)&b->d_part_docs, (nt * topk + 1) * 4));
                HIP_TRY(pool_alloc(dev, (void **)&b->d_part_scores, (nt * topk + 1) * 8));
        }
        if (b->n_tree) {
                const size_t plw = b->plw, nterms = b->tree_terms.size(), nhid = b->tree_hidden.size(), nchunks = (plw + TREE_CHUNK_WORDS - 1) / TREE_CHUNK_WORDS;
                const size_t w_rows = nterms * PL_PLANES * plw, w_prows = nhid * plw, w_qbits = (size_t)b->n_tree * plw, w_cc = (size_t)b->n_tree * nchunks + 64;
                HIP_TRY(pool_alloc(dev, (void **)&b->d_tree_scratch, (w_rows + w_prows + w_qbits + w_cc + 2 * nterms + 64) * 4));
                b->d_tree_rows = b->d_tree_scratch;
                b->d_tree_prows = b->d_tree_rows + w_rows;
                b->d_tree_qbits = b->d_tree_prows + w_prows;
                b->d_tree_cc = b->d_tree_qbits + w_qbits;
                b->d_tree_build = b->d_tree_cc + w_cc;
                std::vector<uint32_t> build(2 * nterms);
                for (size_t i = 0; i < nterms; ++i)
                        build[2 * i] = b->tree_terms[i], build[2 * i + 1] = (uint32_t)i;
                if (nterms)
                        HIP_TRY(hipMemcpyAsync(b->d_tree_build, build.data(), build.size() * 4, hipMemcpyHostToDevice, dev->stream_up)); // (pageable source: staged before the call returns)
                if (scored && topk)
                        HIP_TRY(pool_alloc(dev, (void **)&b->d_tree_scores, (off + 64) * 8));
        }
        HIP_TRY(hipEventRecord(b->ev_up, dev->stream_up));
        b->info.nqueries = nq;
        b->info.tree_queries = b->tree_queries;
        b->info.bitmap_queries = b->bitmap_queries;
        b->info.tree_scratch_bytes = b->tree_scratch_bytes;
        b->info.out_capacity = off;
        b->info.dense_queries = b->dense_queries;
        b->info.cand_queries = b->cand_queries;
        b->info.fused_queries = b->fused_queries;
        b->info.planes_queries = b->planes_queries;
        b->info.unsupported_queries = b->unsupported_queries;
        b->info.plane_terms = b->plane_terms.size();
        b->info.plane_bytes = (uint64_t)b->plane_terms.size() * (b->planes_hi ? PL_PLANES : 1u) * b->plw * 4; // (what this batch reads of its rows of the index's plane cache: plane 0, a scored batch the high parts too)
        b->info.launches = (b->n_dense != 0) + (b->n_pset != 0) + (b->n_probe != 0) + (b->n_cand != 0) + (b->n_fused != 0) + (b->n_fused16 != 0) + (b->n_fusedgen != 0) + (b->n_planes != 0) + (b->n_planes8 != 0) + (!b->plane_terms.empty()) +
                           (!b->ptasks.empty()) + (rich ? 2 : 0) +
                           ((scored && b->n_dense + b->n_pset + b->n_probe + b->n_cand) ? 1 : 0) + ((scored && topk) ? 1 : 0);
        b->info.create_plan_ms = (float)(b->plan_ms[0] + b->plan_ms[1] + b->plan_ms[2] + b->plan_ms[3]);
        b->info.create_ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create).count();
        if (dbg_create)
                fprintf(stderr, "[tri create] nq %zu: plan %.3f  arena+copy %.3f  planes/sparse %.3f  out(%.1f MB) %.3f  rest %.3f  total %.3f ms\n", nq, dbg_t[0], dbg_t[1] - dbg_t[0],
                        dbg_t[2] - dbg_t[1], (double)off * 4 / 1e6, dbg_t[3] - dbg_t[2], b->info.create_ms - dbg_t[3], b->info.create_ms);
        *out = b.release();
        return TRI_OK;
}


extern "C" void tri_batch_destroy(tri_batch *b) {
        delete b; // ~tri_batch releases the device buffers
}

extern "C" int tri_batch_run(tri_batch *b) {
        if (!b)
                return fail(TRI_ERR_INVALID, "null batch");
        tri_dev *dev = b->dev;
        DevLock dev_lock(dev->mu);
        HIP_TRY(hipSetDevice(dev->device));
        b->synced = false;
        const uint32_t n = (uint32_t)b->tasks.size();
#ifdef TRI_TRACE
        if (!g_trace_host) {
                HIP_TRY(hipHostMalloc((void **)&g_trace_host, 64 * 16, hipHostMallocMapped | hipHostMallocCoherent));
                uint32_t *dptr = nullptr;
                HIP_TRY(hipHostGetDevicePointer((void **)&dptr, g_trace_host, 0));
                HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &dptr, sizeof dptr));
        }
        memset(g_trace_host, 0, 64 * 16);
#endif
        b->ran = true;
#ifdef TRI_TASKTIMES
        if (b->tasks.size()) {
                if (g_tt_cap < 8 * (size_t)b->tasks.size()) {
                        if (g_tt_host)
                                hipHostFree(g_tt_host);
                        g_tt_cap = 8 * (size_t)b->tasks.size() + 1024;
                        HIP_TRY(hipHostMalloc((void **)&g_tt_host, g_tt_cap * 8, hipHostMallocMapped | hipHostMallocCoherent));
                        unsigned long long *dptr = nullptr;
                        HIP_TRY(hipHostGetDevicePointer((void **)&dptr, g_tt_host, 0));
                        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_tt), &dptr, sizeof dptr));
                }
                memset(g_tt_host, 0, g_tt_cap * 8);
        }
#endif
        HIP_TRY(hipStreamWaitEvent(dev->stream, b->ev_up, 0)); // the plan's copy (upload stream) has arrived
        if (b->ix->ev_pc_ready)
                HIP_TRY(hipStreamWaitEvent(dev->stream, b->ix->ev_pc_ready, 0)); // ... and so have the plane cache's rows, should it have been grown since
        HIP_TRY(hipEventRecord(b->ev0, dev->stream));
        if (n) {
                HIP_TRY(hipMemsetAsync(b->d_ticket, 0, TICKET_BYTES, dev->stream));
                // two persistent kernels back to back on the engine stream: bitmap windows (512 threads), then candidate tiles.
                // GOOGLE: matching reads the contiguous delta streams, not the chunks (see tri_index::d_dstream)
                const uint8_t *match_bytes = b->ix->codec == TRI_CODEC_GOOGLE ? b->ix->d_dstream : b->ix->d_index;
                const uint32_t *match_off = b->ix->codec == TRI_CODEC_GOOGLE ? b->ix->d_blk_doff : b->ix->d_blk_off;
                uint32_t dense_wgs = TRI_DENSE_WAVES * 256 / DENSE_WG, cand_wgs = 4; // workgroups per CU
                bool overlap = false;
                if (dev->opt.overlap_dense_wgs && dev->opt.overlap_cand_wgs && (b->n_dense || b->n_pset) && b->n_cand) { // the window kernels and the candidate-tile kernel side by side
                        overlap = true;
                        dense_wgs = (uint32_t)dev->opt.overlap_dense_wgs;
                        cand_wgs = (uint32_t)dev->opt.overlap_cand_wgs;
                } else if (dev->opt.overlap && b->n_cand && b->n_dense + b->n_pset + b->n_probe)
                        overlap = true; // (full grids: the second kernel's workgroups take the slots the first one's tail leaves)
                b->info.term_planes_decoded_bytes = 0;
                if (!b->plane_terms.empty()) {
                        // the head terms the batch's queries share: the rows of the index's plane cache that no earlier run has built are decoded now
                        // — once for the index, not once per batch (every word of a row is written: no memset)
                        tri_index *ix = b->ix;
                        std::vector<uint32_t> build;
                        const uint8_t needs = b->planes_hi ? 3u : 1u; // (bit 0: plane 0; bit 1: the row's high part — scored batches only)
                        for (const uint32_t term : b->plane_terms) {
                                const uint32_t row = ix->df_rank[term];
                                if (row < ix->pc_cap && ((ix->pc_built[row] & needs) != needs || dev->opt.planes_rebuild)) {
                                        build.push_back(term);
                                        build.push_back(row);
                                        b->info.term_planes_decoded_bytes += ix->docbytes[term];
                                }
                        }
                        if (!build.empty()) {
                                HIP_TRY(hipMemcpyAsync(b->d_build, build.data(), build.size() * 4, hipMemcpyHostToDevice, dev->stream)); // (pageable source: staged before the call returns)
                                const uint32_t nrows = (uint32_t)(build.size() / 2), nwin = b->plw / PL_WORDS;
                                for (uint32_t y0 = 0; y0 < nrows; y0 += 65535u) { // (gridDim.y <= 65535)
                                        const uint32_t ny = std::min(65535u, nrows - y0);
                                        if (b->planes_hi) // (a row that gains its high part is decoded whole again: plane 0 is rewritten with the words it holds)
                                                TRI_LAUNCH(k_term_planes, ix->codec, dim3(nwin, ny), dim3(AND_WG), dev->stream, ix->d_index, ix->d_blk_last, ix->d_blk_off, ix->d_blk_rec,
                                                           ix->d_blk_doff, ix->d_win, ix->d_terms, (const uint32_t *)b->d_build + 2 * (size_t)y0, ix->d_pcache, (size_t)b->plw, ix->d_pcache_hi,
                                                           (size_t)PL_HI * b->plw, b->plw, (uint32_t *)nullptr);
                                        else // plane 0 alone, P0_GROUP windows to a workgroup (the rank records: built when a phrase batch asks for them, below)
                                                TRI_LAUNCH(k_term_plane0, ix->codec, dim3((nwin + P0_GROUP - 1) / P0_GROUP, ny), dim3(AND_WG), dev->stream, ix->d_index, ix->d_blk_last, ix->d_blk_off,
                                                           ix->d_blk_rec, ix->d_blk_doff, ix->d_win, ix->d_terms, (const uint32_t *)b->d_build + 2 * (size_t)y0, ix->d_pcache, b->plw, (uint32_t *)nullptr);
                                        HIP_TRY(hipGetLastError());
                                }
                                for (size_t i = 1; i < build.size(); i += 2)
                                        ix->pc_built[build[i]] |= needs;
                        }
                }
                HIP_TRY(hipEventRecord(b->ev_pl, dev->stream));
                hipStream_t cand_stream = dev->stream;
                if (overlap) {
                        HIP_TRY(hipEventRecord(dev->ev_fork, dev->stream));
                        HIP_TRY(hipStreamWaitEvent(dev->stream2, dev->ev_fork, 0));
                        cand_stream = dev->stream2;
                }
                // unions with terms that have no plane (PSET_UNIT_SCATTER): those terms' documents listed task by task, a workgroup per query (units[] holds the TASK_PROBE units
                // too) — on the second stream, beside k_and_dense, where that stream is not k_and's (option overlap)
                const bool prep = b->n_pset && b->pscatter_queries, prep_forked = prep && !overlap && b->n_dense;
                if (prep) {
                        if (prep_forked) {
                                HIP_TRY(hipEventRecord(dev->ev_fork, dev->stream));
                                HIP_TRY(hipStreamWaitEvent(dev->stream2, dev->ev_fork, 0));
                        }
                        hipStream_t prep_stream = prep_forked ? dev->stream2 : dev->stream;
                        const uint32_t nunits = b->n_pset + b->n_probe, nscat = (uint32_t)b->pscatter_queries;
                        hipLaunchKernelGGL(k_psets_prep_list, dim3((nunits + 255) / 256), dim3(256), 0, prep_stream, (const DevPsetUnit *)(b->d_arena + b->off_units), nunits,
                                           b->d_ticket + TICKET_SCAT_WORD + 1, b->d_scat_list, nscat);
                        HIP_TRY(hipGetLastError());
                        TRI_LAUNCH(k_psets_prep, b->ix->codec, dim3(nscat), dim3(PSCAT_WG), prep_stream, (const DevPsetUnit *)(b->d_arena + b->off_units), (const uint32_t *)b->d_scat_list,
                                   (const uint32_t *)(b->d_ticket + TICKET_SCAT_WORD + 1), b->d_plan, b->d_tasks, (const uint32_t *)b->d_qterms, (const uint32_t *)b->d_qplane, b->ix->d_masked, b->ix->d_index, b->ix->d_blk_last, b->ix->d_blk_off,
                                   b->ix->d_blk_rec, b->ix->d_blk_doff, b->ix->d_terms, b->d_ticket + TICKET_SCAT_WORD, b->d_scat_off, b->d_scat_cnt, b->d_scat_docs, b->scat_cap);
                        HIP_TRY(hipGetLastError());
                        if (prep_forked)
                                HIP_TRY(hipEventRecord(dev->ev_join, dev->stream2));
                }
                if (b->n_dense) {
                        TRI_LAUNCH(k_and_dense, b->ix->codec, dim3(std::min<uint32_t>(b->n_dense, (uint32_t)dev->cus * dense_wgs)), dim3(DENSE_WG), dev->stream, match_bytes,
                                           b->ix->d_blk_last, match_off, b->ix->d_win, b->ix->d_terms, b->d_plan, b->d_tasks, b->d_sched, b->d_qterms, b->n_dense,
                                           b->d_ticket + 16, b->d_out, b->d_counts, b->ix->d_masked, (const uint32_t *)b->d_qplane, (const uint32_t *)b->ix->d_pcache, b->plw);
                        HIP_TRY(hipGetLastError());
                }
                HIP_TRY(hipEventRecord(b->ev_a, dev->stream));
                if (prep_forked) // (k_psets reads the lists k_psets_prep made beside k_and_dense)
                        HIP_TRY(hipStreamWaitEvent(dev->stream, dev->ev_join, 0));
                if (b->n_pset) {
                        // the queries all of whose terms have planes: word-wise algebra over the planes + expansion (k_psets.hpp)
                        TRI_LAUNCH(k_psets, b->ix->codec, dim3(std::min<uint32_t>(b->n_pset, (uint32_t)dev->cus * (overlap && dev->opt.overlap_dense_wgs ? std::min<uint32_t>(dense_wgs, TRI_PSET_WAVES * 256 / PSET_WG) : TRI_PSET_WAVES * 256 / PSET_WG))), dim3(PSET_WG), dev->stream,
                                   (const DevPsetUnit *)(b->d_arena + b->off_units), (const uint32_t *)(b->d_arena + b->off_pset_sched), b->n_pset, b->d_ticket + 20,
                                   (const uint32_t *)b->d_qterms, (const uint32_t *)b->d_qplane, b->d_out, b->d_counts, b->ix->d_masked, (const uint32_t *)b->ix->d_pcache, b->plw,
                                   (const uint32_t *)b->d_scat_off, (const uint32_t *)b->d_scat_cnt, (const uint32_t *)b->d_scat_docs);
                        HIP_TRY(hipGetLastError());
                }
                HIP_TRY(hipEventRecord(b->ev_s, dev->stream));
                if (b->n_probe) {
                        // one short lead list against lists that all have planes: a wave per task, the lead's documents probed from registers (k_probe.hpp)
                        TRI_LAUNCH(k_probe, b->ix->codec, dim3(std::min<uint32_t>((b->n_probe + PROBE_WG / 64 - 1) / (PROBE_WG / 64), (uint32_t)dev->cus * (TRI_PROBE_WAVES * 256 / PROBE_WG))),
                                   dim3(PROBE_WG), dev->stream, match_bytes, b->ix->d_blk_last, match_off, b->ix->d_terms, (const DevPsetUnit *)(b->d_arena + b->off_units),
                                   (const uint32_t *)(b->d_arena + b->off_pset_sched) + b->n_pset, b->n_probe, b->d_ticket + 22, (const uint32_t *)b->d_qterms,
                                   (const uint32_t *)b->d_qplane, b->d_out, b->d_counts, b->ix->d_masked, (const uint32_t *)b->ix->d_pcache, b->plw);
                        HIP_TRY(hipGetLastError());
                }
                HIP_TRY(hipEventRecord(b->ev_r, dev->stream));
                if (b->n_cand)
                        TRI_LAUNCH(k_and, b->ix->codec, dim3(std::min<uint32_t>(b->n_cand, (uint32_t)dev->cus * cand_wgs)), dim3(AND_WG), cand_stream, match_bytes,
                                           b->ix->d_blk_last, match_off, b->ix->d_win, b->ix->d_terms, b->d_plan, b->d_tasks, b->d_sched + b->n_dense + b->n_pset + b->n_probe, b->d_qterms,
                                           (const uint32_t *)(b->d_arena + b->off_cand_q), b->d_ticket + TICKET_CAND_WORD, b->d_out, b->d_counts, b->ix->d_masked, (const uint32_t *)b->d_qplane, (const uint32_t *)b->ix->d_pcache, b->plw);
                HIP_TRY(hipGetLastError());
                if (overlap) {
                        HIP_TRY(hipEventRecord(dev->ev_join, dev->stream2));
                        HIP_TRY(hipStreamWaitEvent(dev->stream, dev->ev_join, 0));
                }
                HIP_TRY(hipEventRecord(b->ev_b, dev->stream));
                // AccumulatedScore top-K of the dense queries: decode -> match -> score -> select in one pass; as many workgroups per CU as its
                // LDS holds.  Two instantiations: 32-bit window words, and 16-bit ones (queries of <= 5 distinct terms: windows twice as long)
                for (int variant = 0; variant < 3; ++variant) { // 0: 32-bit words, 1: 16-bit words, 2: general trees (32-bit words)
                        const uint32_t nf = variant == 0 ? b->n_fused : variant == 1 ? b->n_fused16 : b->n_fusedgen;
                        if (!nf)
                                continue;
                        const uint32_t *fsched = b->d_sched + b->n_dense + b->n_pset + b->n_probe + b->n_cand + (variant >= 1 ? b->n_fused : 0) + (variant == 2 ? b->n_fused16 : 0);
                        const dim3 grid(std::min<uint32_t>(nf, (uint32_t)dev->cus * FUS_WGS_PER_CU));
#define TRI_FUSED_ARGS                                                                                                                                 \
        b->ix->d_index, b->ix->d_blk_last, b->ix->d_blk_off, b->ix->d_blk_rec, b->ix->d_blk_doff, b->ix->d_win, b->ix->d_terms, b->d_plan, b->d_fused, b->d_tasks, fsched, \
                b->d_sterms, b->d_sweights, nf, b->d_ticket + 56 + 2 * variant, b->d_counts, b->topk, b->d_part_docs, b->d_part_scores,                \
                b->d_part_counts, b->ix->d_masked, b->similarity, b->d_out, b->d_all_scores, b->d_rich_allow
                        if (b->ix->codec == TRI_CODEC_LUCENE) {
                                if (variant == 0)
                                        hipLaunchKernelGGL((k_fused<CODEC_LUCENE, 0, 0>), grid, dim3(FUS_WG), 0, dev->stream, TRI_FUSED_ARGS);
                                else if (variant == 1)
                                        hipLaunchKernelGGL((k_fused<CODEC_LUCENE, 1, 0>), grid, dim3(FUS_WG), 0, dev->stream, TRI_FUSED_ARGS);
                                else
                                        hipLaunchKernelGGL((k_fused<CODEC_LUCENE, 0, 1>), grid, dim3(FUS_WG), 0, dev->stream, TRI_FUSED_ARGS);
                        } else {
                                if (variant == 0)
                                        hipLaunchKernelGGL((k_fused<CODEC_GOOGLE, 0, 0>), grid, dim3(FUS_WG), 0, dev->stream, TRI_FUSED_ARGS);
                                else if (variant == 1)
                                        hipLaunchKernelGGL((k_fused<CODEC_GOOGLE, 1, 0>), grid, dim3(FUS_WG), 0, dev->stream, TRI_FUSED_ARGS);
                                else
                                        hipLaunchKernelGGL((k_fused<CODEC_GOOGLE, 0, 1>), grid, dim3(FUS_WG), 0, dev->stream, TRI_FUSED_ARGS);
                        }
#undef TRI_FUSED_ARGS
                        HIP_TRY(hipGetLastError());
                }
                HIP_TRY(hipEventRecord(b->ev_c, dev->stream));
                if (b->d_qthr)
                        HIP_TRY(hipMemsetAsync(b->d_qthr, 0, (b->plan.size() + 1) * 8, dev->stream));
                for (int wide = 0; wide < 2; ++wide) {
                        // AccumulatedScore top-K of the CNF queries over bit planes: the head terms' planes from k_term_planes, the other lists
                        // decoded per window into LDS planes; union / conjunction predicates and the candidate filter 32 documents per word
                        // (two instantiations: queries of up to five slots, wider ones)
                        const uint32_t np = wide ? b->n_planes8 : b->n_planes;
                        if (!np)
                                continue;
                        const uint32_t *psched = b->d_sched + b->n_dense + b->n_pset + b->n_probe + b->n_cand + b->n_fused + b->n_fused16 + b->n_fusedgen + (wide ? b->n_planes : 0);
                        const dim3 grid(std::min<uint32_t>(np, (uint32_t)dev->cus * PLK_WGS_PER_CU));
#define TRI_PLANES_ARGS                                                                                                                                      \
        b->ix->d_index, b->ix->d_blk_last, b->ix->d_blk_off, b->ix->d_blk_rec, b->ix->d_blk_doff, b->ix->d_win, b->ix->d_terms, b->d_plan, b->d_fused, b->d_tasks, psched, \
                b->d_sterms, b->d_sweights, np, b->d_ticket + 24 + 2 * wide, b->d_counts, b->topk, b->d_part_docs, b->d_part_scores, b->d_part_counts, b->ix->d_masked,     \
                b->similarity, (const uint32_t *)b->ix->d_pcache, (const uint32_t *)b->ix->d_pcache_hi, b->plw, b->ix->pc_cap, b->d_sparse, b->sparse_cap, b->d_qthr
                        if (b->ix->codec == TRI_CODEC_LUCENE) {
                                if (wide)
                                        hipLaunchKernelGGL((k_planes<CODEC_LUCENE, FUS_MAX_SLOTS>), grid, dim3(PLK_WG), 0, dev->stream, TRI_PLANES_ARGS);
                                else
                                        hipLaunchKernelGGL((k_planes<CODEC_LUCENE, PLK_NS_SMALL>), grid, dim3(PLK_WG), 0, dev->stream, TRI_PLANES_ARGS);
                        } else {
                                if (wide)
                                        hipLaunchKernelGGL((k_planes<CODEC_GOOGLE, FUS_MAX_SLOTS>), grid, dim3(PLK_WG), 0, dev->stream, TRI_PLANES_ARGS);
                                else
                                        hipLaunchKernelGGL((k_planes<CODEC_GOOGLE, PLK_NS_SMALL>), grid, dim3(PLK_WG), 0, dev->stream, TRI_PLANES_ARGS);
                        }
#undef TRI_PLANES_ARGS
                        HIP_TRY(hipGetLastError());
                }
                HIP_TRY(hipEventRecord(b->ev_k, dev->stream));
                if (!b->ptasks.empty() && b->ix->codec == TRI_CODEC_GOOGLE && b->ix->d_prank && b->ix->pc_cap) {
                        // the phrases' head terms are located by RANK in plane 0 (k_phrase.hpp): rows (plane 0 + rank directory) and per-posting hits entries of the
                        // phrase terms that have none yet are built now — once for the index
                        tri_index *ix = b->ix;
                        std::vector<uint32_t> hits_build, hits_only; // (term, row) pairs: the rows that also need their rank records FIRST, then the ones that have them
                        uint32_t max_blocks = 0;
                        for (size_t i = 0; i < b->pterms.size(); ++i) {
                                const uint32_t term = b->pterms[i], r = ix->df_rank[term];
                                if (r >= ix->pc_cap || ix->ph_built[r])
                                        continue;
                                const DevTerm &t = ix->terms[term];
                                if (!(t.flags & TERM_FULL_BLOCKS) || !t.documents)
                                        continue;
                                ix->ph_built[r] = 1;
                                max_blocks = std::max(max_blocks, t.nblocks);
                                std::vector<uint32_t> &dst = (ix->pc_built[r] & 4u) ? hits_only : hits_build; // (bit 2: the row's rank records — with them plane 0, if it is not there yet)
                                ix->pc_built[r] |= 5u;
                                dst.push_back(term);
                                dst.push_back(r);
                        }
                        const uint32_t nrows_build = (uint32_t)(hits_build.size() / 2);
                        hits_build.insert(hits_build.end(), hits_only.begin(), hits_only.end());
                        if (!hits_build.empty()) {
                                uint32_t *d_pairs = ix->d_ph_pairs + 2 * ix->ph_pairs_n; // (every row is built once: the pairs of all runs fit 2 * pc_cap words)
                                HIP_TRY(hipMemcpyAsync(d_pairs, hits_build.data(), hits_build.size() * 4, hipMemcpyHostToDevice, dev->stream)); // (pageable source: staged before the call returns)
                                ix->ph_pairs_n += hits_build.size() / 2;
                                // the rows without rank records: plane 0 + records in ONE launch over the list's head (round 5 launched a grid per row: 711 launches, 11 ms, the
                                // first time cfg4's phrases met an index)
                                for (uint32_t y0 = 0; y0 < nrows_build; y0 += 65535u) {
                                        const dim3 grid((b->plw / PL_WORDS + P0_GROUP - 1) / P0_GROUP, std::min(65535u, nrows_build - y0));
                                        TRI_LAUNCH(k_term_plane0, ix->codec, grid, dim3(AND_WG), dev->stream, ix->d_index, ix->d_blk_last, ix->d_blk_off, ix->d_blk_rec, ix->d_blk_doff, ix->d_win,
                                                   ix->d_terms, (const uint32_t *)d_pairs + 2 * (size_t)y0, ix->d_pcache, b->plw, ix->d_prank);
                                        HIP_TRY(hipGetLastError());
                                }
                                const uint32_t npairs = (uint32_t)(hits_build.size() / 2);
                                for (uint32_t y0 = 0; y0 < npairs; y0 += 65535u) { // (gridDim.y <= 65535: a small index at a high plane_div makes almost every term eligible)
                                        hipLaunchKernelGGL(k_term_hits, dim3((max_blocks + 255) / 256, std::min(65535u, npairs - y0)), dim3(256), 0, dev->stream, ix->d_index, ix->d_blk_off,
                                                           ix->d_blk_hits, ix->d_terms, (const uint32_t *)d_pairs + 2 * (size_t)y0, (const uint64_t *)ix->d_hs_off, ix->d_phs, ix->d_term_row);
                                        HIP_TRY(hipGetLastError());
                                }
                        }
                }
                if (!b->ptasks.empty()) {
                        // positional constraints: filter + compact the match segments of the queries that hold phrases
                        const uint32_t np = (uint32_t)b->ptasks.size();
                        TRI_LAUNCH(k_phrase, b->ix->codec, dim3(std::min<uint32_t>(np, (uint32_t)dev->cus * PHRASE_WGS_PER_CU)), dim3(AND_WG), dev->stream, b->ix->d_index,
                                           b->ix->d_hits, b->ix->d_blk_hits, b->ix->d_hdir, b->ix->d_blk_last, b->ix->d_blk_off, b->ix->d_win, b->ix->d_terms, b->d_plan, b->d_tasks, b->d_ptasks, np, b->d_phrases, b->d_pterms,
                                           b->d_ticket + 48, b->d_out, b->d_counts, b->d_pscore,
                                           (b->flags & TRI_FLAG_ACCUMULATED_SCORE) ? 65535u : 1u, // exec.cpp:296 trackCnt
                                           b->similarity, (const uint32_t *)b->ix->d_pcache, b->ix->pc_plw, (const uint32_t *)b->ix->d_prank, (const unsigned long long *)b->ix->d_phs,
                                           (const uint64_t *)b->ix->d_hs_off, (const uint32_t *)(b->ix->codec == TRI_CODEC_GOOGLE ? b->ix->d_term_row : nullptr));
                        HIP_TRY(hipGetLastError());
                }
                HIP_TRY(hipEventRecord(b->ev_p, dev->stream));
                if (b->n_tree) {
                        // the queries no other kernel takes (k_tree.hpp): leaf bitmaps — the distinct term leaves decoded once, the phrase leaves from their
                        // hidden queries' match lists (k_phrase has just filtered them) —, the trees word by word, the match bitmaps expanded
                        const uint32_t plw = b->plw, nterms = (uint32_t)b->tree_terms.size(), nhid = (uint32_t)b->tree_hidden.size();
                        const uint32_t nchunks = (plw + TREE_CHUNK_WORDS - 1) / TREE_CHUNK_WORDS;
                        const uint32_t *tsched = b->d_sched + (n - b->n_tree);
                        const uint32_t *d_tree = (const uint32_t *)(b->d_arena + b->off_tree);
                        tri_index *ix = b->ix;
                        for (uint32_t y0 = 0; y0 < nterms; y0 += 65535u) { // (gridDim.y <= 65535)
                                const dim3 grid(plw / PL_WORDS, std::min(65535u, nterms - y0));
                                TRI_LAUNCH(k_term_planes, ix->codec, grid, dim3(AND_WG), dev->stream, ix->d_index, ix->d_blk_last, ix->d_blk_off, ix->d_blk_rec, ix->d_blk_doff, ix->d_win,
                                           ix->d_terms, (const uint32_t *)b->d_tree_build + 2 * (size_t)y0, b->d_tree_rows, (size_t)PL_PLANES * plw, b->d_tree_rows + plw,
                                           (size_t)PL_PLANES * plw, plw, (uint32_t *)nullptr); // (a tree row keeps its two parts side by side: plane k at k * plw)
                                HIP_TRY(hipGetLastError());
                        }
                        if (nhid) {
                                HIP_TRY(hipMemsetAsync(b->d_tree_prows, 0, (size_t)nhid * plw * 4, dev->stream));
                                hipLaunchKernelGGL(k_tree_gather, dim3(nhid), dim3(TREE_WG), 0, dev->stream, b->d_plan, b->d_tasks, (const uint32_t *)(b->d_arena + b->off_tree_hidden), b->d_out,
                                                   b->d_counts, b->d_pscore, b->d_tree_prows, plw);
                                HIP_TRY(hipGetLastError());
                        }
                        const bool scored_run = b->flags & TRI_FLAG_ACCUMULATED_SCORE, rich_run = b->flags & TRI_FLAG_MATCHED_TERMS;
                        double *tscores = scored_run ? (b->topk ? b->d_tree_scores : b->d_all_scores) : nullptr;
                        for (uint32_t y0 = 0; y0 < b->n_tree; y0 += 65535u) {
                                const dim3 grid(nchunks, std::min(65535u, b->n_tree - y0));
                                uint32_t *qbits = b->d_tree_qbits + (size_t)y0 * plw, *cc = b->d_tree_cc + (size_t)y0 * nchunks;
                                hipLaunchKernelGGL(k_tree_eval, grid, dim3(TREE_WG), 0, dev->stream, b->d_plan, b->d_tasks, tsched + y0, d_tree, (const uint32_t *)b->d_tree_rows,
                                                   (const uint32_t *)b->d_tree_prows, (const uint32_t *)ix->d_masked, qbits, cc, plw);
                                hipLaunchKernelGGL(k_tree_expand, grid, dim3(TREE_WG), 0, dev->stream, b->d_plan, b->d_tasks, tsched + y0, (const uint32_t *)qbits, (const uint32_t *)cc,
                                                   b->d_out, b->d_counts, plw);
                                if (scored_run || rich_run)
                                        TRI_LAUNCH(k_tree_leaves, ix->codec, grid, dim3(TREE_WG), dev->stream, ix->d_index, ix->d_blk_last, ix->d_blk_off, ix->d_terms, b->d_plan, b->d_tasks,
                                                   tsched + y0, d_tree, (const uint32_t *)b->d_tree_rows, (const uint32_t *)b->d_tree_prows, (const uint32_t *)cc, (const uint32_t *)b->d_out,
                                                   (const uint32_t *)b->d_counts, (const double *)b->d_sweights, (const double *)b->d_pscore, tscores, rich_run ? b->d_rich_allow : nullptr,
                                                   plw, b->similarity);
                                HIP_TRY(hipGetLastError());
                        }
                        if (scored_run && b->topk) {
                                hipLaunchKernelGGL(k_tree_topk, dim3(b->n_tree), dim3(AND_WG), 0, dev->stream, tsched, b->d_tasks, (const uint32_t *)b->d_out, (const uint32_t *)b->d_counts,
                                                   (const double *)tscores, b->topk, b->d_part_docs, b->d_part_scores, b->d_part_counts);
                                HIP_TRY(hipGetLastError());
                        }
                }
                HIP_TRY(hipEventRecord(b->ev_t, dev->stream));
                if (b->flags & TRI_FLAG_MATCHED_TERMS) {
                        // COUNT pass: which reportable terms hold each match, with what frequency; hit totals per task
                        HIP_TRY(hipMemsetAsync(b->d_rich_present, 0, (b->out_capacity + 64) * 4, dev->stream));
                        HIP_TRY(hipMemsetAsync(b->d_rich_freq, 0, (b->out_capacity + 64) * 2 * b->rich_R, dev->stream));
                        HIP_TRY(hipMemsetAsync(b->d_task_hits, 0, (b->tasks.size() + 1) * 4, dev->stream));
                        if (b->ix->codec == TRI_CODEC_LUCENE)
                                hipLaunchKernelGGL((k_rich<CODEC_LUCENE, false>), dim3(std::min<uint32_t>(n, (uint32_t)dev->cus * 3)), dim3(AND_WG), 0, dev->stream, b->ix->d_index,
                                                   b->ix->d_hits, b->ix->d_blk_hits, b->ix->d_hdir, b->ix->d_blk_last, b->ix->d_blk_off, b->ix->d_terms, b->d_plan, b->d_tasks, b->d_sched,
                                                   b->d_sterms, n, b->d_ticket + 32, b->d_out, b->d_counts, b->rich_R, b->d_rich_present, b->d_rich_freq, b->d_task_hits,
                                                   (const uint64_t *)nullptr, (uint16_t *)nullptr, (const uint32_t *)b->d_rich_allow, (uint8_t *)nullptr, (uint64_t *)nullptr);
                        else
                                hipLaunchKernelGGL((k_rich<CODEC_GOOGLE, false>), dim3(std::min<uint32_t>(n, (uint32_t)dev->cus * 3)), dim3(AND_WG), 0, dev->stream, b->ix->d_index,
                                                   b->ix->d_hits, b->ix->d_blk_hits, b->ix->d_hdir, b->ix->d_blk_last, b->ix->d_blk_off, b->ix->d_terms, b->d_plan, b->d_tasks, b->d_sched,
                                                   b->d_sterms, n, b->d_ticket + 32, b->d_out, b->d_counts, b->rich_R, b->d_rich_present, b->d_rich_freq, b->d_task_hits,
                                                   (const uint64_t *)nullptr, (uint16_t *)nullptr, (const uint32_t *)b->d_rich_allow, (uint8_t *)nullptr, (uint64_t *)nullptr);
                        HIP_TRY(hipGetLastError());
                }
                if (b->flags & TRI_FLAG_ACCUMULATED_SCORE) {
                        const uint32_t nlegacy = b->n_dense + b->n_pset + b->n_probe + b->n_cand; // (the sets k_and_dense / k_psets / k_probe / k_and materialised; the one-pass tasks have scored themselves)
                        if (nlegacy) {
                        hipLaunchKernelGGL(k_score_order, dim3(1), dim3(SORD_WG), 0, dev->stream, (const uint32_t *)b->d_sched, (const uint32_t *)b->d_counts, nlegacy, b->d_score_order);
                        HIP_TRY(hipGetLastError());
                        TRI_LAUNCH(k_score, b->ix->codec, dim3(std::min<uint32_t>(nlegacy, (uint32_t)dev->cus * SCORE_WGS_PER_CU)), dim3(AND_WG), dev->stream, b->ix->d_index,
                                           b->ix->d_blk_last, b->ix->d_blk_off, b->ix->d_terms, b->d_plan, b->d_tasks, (const uint32_t *)b->d_score_order, b->d_sterms, b->d_sweights, nlegacy,
                                           b->d_ticket + 32, b->d_out, b->d_counts, b->topk, b->d_part_docs, b->d_part_scores, b->d_part_counts,
                                           b->d_all_scores, b->d_pscore, b->similarity, b->ix->d_win, b->splane.empty() ? (const uint32_t *)nullptr : (const uint32_t *)(b->d_arena + b->off_splane),
                                           (const uint32_t *)b->ix->d_pcache_hi, b->plw); // (the scorers read the level words: the rows' high parts)
                        }
                        HIP_TRY(hipGetLastError());
                        const uint32_t nqs = (uint32_t)b->plan.size();
                        if (b->topk)
                                hipLaunchKernelGGL(k_topk_merge, dim3(std::min<uint32_t>(nqs, (uint32_t)dev->cus * 8)), dim3(AND_WG), 0, dev->stream, b->d_plan, nqs,
                                           b->topk, b->d_part_docs, b->d_part_scores, b->d_part_counts, b->d_top_docs, b->d_top_scores, b->d_top_counts);
                        HIP_TRY(hipGetLastError());
                }
        }
        else {
                HIP_TRY(hipEventRecord(b->ev_pl, dev->stream));
                HIP_TRY(hipEventRecord(b->ev_a, dev->stream));
                HIP_TRY(hipEventRecord(b->ev_s, dev->stream));
                HIP_TRY(hipEventRecord(b->ev_r, dev->stream));
                HIP_TRY(hipEventRecord(b->ev_b, dev->stream));
                HIP_TRY(hipEventRecord(b->ev_c, dev->stream));
                HIP_TRY(hipEventRecord(b->ev_k, dev->stream));
                HIP_TRY(hipEventRecord(b->ev_p, dev->stream));
                HIP_TRY(hipEventRecord(b->ev_t, dev->stream));
        }
        if (!b->plan.empty()) {
                const uint32_t nqs = (uint32_t)b->plan.size();
                hipLaunchKernelGGL(k_query_counts, dim3((nqs + 255) / 256), dim3(256), 0, dev->stream, b->d_plan, b->d_counts, nqs, b->d_qcounts);
                HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(b->ev1, dev->stream));
        return TRI_OK;
}

extern "C" int tri_batch_sync(tri_batch *b) {
        if (!b)
                return fail(TRI_ERR_INVALID, "null batch");
        tri_dev *dev = b->ix->dev;
        HIP_TRY(hipSetDevice(dev->device));
        if (!b->ran)
                return fail(TRI_ERR_INVALID, "tri_batch_sync: the batch has not been run");
#if (defined(TRI_TRACE) && !defined(TRI_TRACE_NOPOLL)) || defined(TRI_POLL)
        {
                const char *w = getenv("TRINITY_WATCHDOG_S");
                const double limit = w ? atof(w) : 10.0;
                double waited = 0;
                while (hipEventQuery(b->ev1) == hipErrorNotReady) {
                        struct timespec ts = {0, 50 * 1000 * 1000};
                        nanosleep(&ts, nullptr);
                        waited += 0.05;
                        if (waited > limit) {
                                fprintf(stderr, "[tri watchdog] kernel still running after %.1fs; per-workgroup markers {stage,a,b,count}:\n", waited);
#ifdef TRI_TRACE
                                for (int i = 0; i < 64; ++i)
                                        if (g_trace_host[i * 4 + 3])
                                                fprintf(stderr, "  wg%%64=%d stage=%u a=%u b=%u n=%u\n", i, g_trace_host[i * 4], g_trace_host[i * 4 + 1],
                                                        g_trace_host[i * 4 + 2], g_trace_host[i * 4 + 3]);
#endif
                                fflush(stderr);
                                _exit(3);
                        }
                }
        }
#endif
        // THIS batch's last event, not the engine stream: a caller that keeps the stream fed (the next batch launched before this one is
        // awaited — bench.py's loop) gets this batch's results when THEY are ready, not when everything queued behind them is
        HIP_TRY(hipEventSynchronize(b->ev1));
        DevLock dev_lock(dev->mu); // (the wait above stands outside the lock; what follows may enqueue on the engine stream)
#ifdef TRI_TASKTIMES
        if (b->tasks.size() && getenv("TRINITY_TASKTIMES")) {
#if TRI_TASKTIMES == 2 // (k_score: tickets run over the docset-materialising tasks, sched[0 ..))
                const uint32_t nc = b->n_dense + b->n_pset + b->n_probe + b->n_cand, first = 0;
#elif TRI_TASKTIMES == 5 // (k_phrase: tickets run over ptasks[])
                const uint32_t nc = (uint32_t)b->ptasks.size(), first = 0;
#elif TRI_TASKTIMES == 4 // (k_and_dense)
                const uint32_t nc = b->n_dense, first = 0;
#elif TRI_TASKTIMES == 3 // (k_planes, the narrow instantiation)
                const uint32_t nc = b->n_planes, first = b->n_dense + b->n_pset + b->n_probe + b->n_cand + b->n_fused + b->n_fused16 + b->n_fusedgen;
#else
                const uint32_t nc = b->n_cand, first = b->n_dense + b->n_pset + b->n_probe;
#endif
                unsigned long long t0 = ~0ull, t1 = 0, busy = 0;
                std::vector<std::pair<unsigned long long, uint32_t>> by;
                for (uint32_t i = 0; i < nc; ++i) {
                        const unsigned long long s = g_tt_host[8 * i], e = g_tt_host[8 * i + 1];
                        if (!s || !e)
                                continue;
                        t0 = std::min(t0, s), t1 = std::max(t1, e);
                        busy += e - s;
                        by.emplace_back(e - s, i);
                }
                std::sort(by.rbegin(), by.rend());
                const double span_us = (double)(t1 - t0) / 100.0;
                const unsigned wgs = std::min<uint32_t>(nc, (uint32_t)dev->cus * (TRI_TASKTIMES == 3 ? PLK_WGS_PER_CU : TRI_TASKTIMES == 2 ? SCORE_WGS_PER_CU : TRI_TASKTIMES == 5 ? PHRASE_WGS_PER_CU : 4));
                fprintf(stderr, "[tri tasktimes] k_and: %u tasks, span %.1f us, busy %.1f %% of %u workgroups; mean task %.2f us\n", nc, span_us,
                        100.0 * (double)busy / ((double)(t1 - t0) * wgs), wgs, (double)busy / 100.0 / std::max<size_t>(1, by.size()));
                if (!by.empty()) { // (by: descending) the distribution, and what share of the workgroups' time the longest tasks take
                        auto at = [&](double f) { return (double)by[std::min(by.size() - 1, (size_t)(f * by.size()))].first / 100.0; };
                        unsigned long long top1 = 0, top5 = 0, top20 = 0;
                        for (size_t i = 0; i < by.size(); ++i) {
                                if (i < by.size() / 100)
                                        top1 += by[i].first;
                                if (i < by.size() / 20)
                                        top5 += by[i].first;
                                if (i < by.size() / 5)
                                        top20 += by[i].first;
                        }
                        fprintf(stderr, "   task us: max %.1f  p99 %.1f  p90 %.1f  p50 %.1f  p10 %.1f; the longest 1 %% / 5 %% / 20 %% of the tasks take %.1f / %.1f / %.1f %% of the time\n", at(0), at(0.01), at(0.1),
                                at(0.5), at(0.9), 100.0 * top1 / busy, 100.0 * top5 / busy, 100.0 * top20 / busy);
                }
                // the finish-time profile: tasks still running at 25 / 50 / 75 / 90 % of the span
                for (const double f : {0.25, 0.5, 0.75, 0.9}) {
                        const unsigned long long at = t0 + (unsigned long long)((double)(t1 - t0) * f);
                        unsigned running = 0;
                        for (uint32_t i = 0; i < nc; ++i)
                                running += g_tt_host[8 * i] <= at && g_tt_host[8 * i + 1] > at;
                        fprintf(stderr, "   at %2.0f %% of the span: %u tasks running\n", f * 100, running);
                }
#if TRI_TASKTIMES == 3 // (k_planes: stamps 2 .. 6 = lists decoded + filter filled, phase A done, tables ready, sweep done, queue drained; 1 = end)
                {
                        const char *nm[8] = {"", "", "setup+lists", "phase A", "prune+tables", "sweep", "drain", "end"};
                        double sum[8] = {0};
                        std::vector<double> span[8];
                        for (uint32_t i = 0; i < nc; ++i) {
                                unsigned long long prev = g_tt_host[8 * i];
                                for (int j = 2; j <= 7; ++j) {
                                        const unsigned long long at = j == 7 ? g_tt_host[8 * i + 1] : g_tt_host[8 * i + j];
                                        if (!at || !prev)
                                                continue;
                                        const double d = (double)(at - prev) / 100.0;
                                        sum[j] += d;
                                        span[j].push_back(d);
                                        prev = at;
                                }
                        }
                        fprintf(stderr, "   phases (us per task: mean / median / p90):");
                        for (int j = 2; j <= 7; ++j)
                                if (!span[j].empty()) {
                                        std::sort(span[j].begin(), span[j].end());
                                        fprintf(stderr, "  %s %.1f / %.1f / %.1f", nm[j], sum[j] / span[j].size(), span[j][span[j].size() / 2], span[j][span[j].size() * 9 / 10]);
                                }
                        fprintf(stderr, "\n");
                        // ... and by the query's DENSE slots (the sweep's instantiation) and whether it holds sparse ones: where the kernel's time goes
                        double tsum[9][2], ssum[9][2], asum[9][2];
                        unsigned tcnt[9][2];
                        for (int x = 0; x < 9; ++x)
                                for (int y = 0; y < 2; ++y)
                                        tsum[x][y] = ssum[x][y] = asum[x][y] = 0.0, tcnt[x][y] = 0;
                        for (uint32_t i = 0; i < nc; ++i) {
                                const unsigned long long s0 = g_tt_host[8 * i], e0 = g_tt_host[8 * i + 1];
                                if (!s0 || !e0)
                                        continue;
                                const DevTask &tk = b->tasks[b->sched[first + i]];
                                const DevFused &z = b->fused[b->plan[tk.slot].fused_idx];
                                uint32_t nd = 0;
                                for (uint32_t k = 0; k < z.nslots; ++k)
                                        nd += z.plane[k] != PL_NONE;
                                const int sp = nd < z.nslots;
                                nd = std::min(nd, 8u);
                                tsum[nd][sp] += (double)(e0 - s0) / 100.0;
                                if (g_tt_host[8 * i + 5] > g_tt_host[8 * i + 4] && g_tt_host[8 * i + 4])
                                        ssum[nd][sp] += (double)(g_tt_host[8 * i + 5] - g_tt_host[8 * i + 4]) / 100.0;
                                if (g_tt_host[8 * i + 3] > g_tt_host[8 * i + 2] && g_tt_host[8 * i + 2])
                                        asum[nd][sp] += (double)(g_tt_host[8 * i + 3] - g_tt_host[8 * i + 2]) / 100.0;
                                ++tcnt[nd][sp];
                        }
                        for (uint32_t nd = 0; nd <= 8; ++nd)
                                for (int sp = 0; sp < 2; ++sp)
                                        if (tcnt[nd][sp])
                                                fprintf(stderr, "   %u dense slots%s: %u tasks, %.1f %% of the time; per task %.1f us (sweep %.1f, phase A %.1f)\n", nd, sp ? " + sparse" : "          ",
                                                        tcnt[nd][sp], 100.0 * tsum[nd][sp] * 100.0 / (double)busy, tsum[nd][sp] / tcnt[nd][sp], ssum[nd][sp] / tcnt[nd][sp], asum[nd][sp] / tcnt[nd][sp]);
                }
#endif
#if TRI_TASKTIMES == 1 // (k_and: the mean of every stamp over all tasks, relative to the task's start)
                {
                        double sum[8] = {0}, cnt[8] = {0};
                        for (uint32_t i = 0; i < nc; ++i)
                                for (int j = 1; j < 8; ++j)
                                        if (g_tt_host[8 * i + j] && g_tt_host[8 * i])
                                                sum[j] += (double)(g_tt_host[8 * i + j] - g_tt_host[8 * i]) / 100.0, cnt[j] += 1;
                        fprintf(stderr, "   means over the tasks (us after the task's start): end %.1f  last tile's lead decoded %.1f", sum[1] / std::max(1.0, cnt[1]), sum[2] / std::max(1.0, cnt[2]));
                        for (int j = 3; j < 8; ++j)
                                if (cnt[j])
                                        fprintf(stderr, "  term %d %.1f (%.0f tasks)", j - 2, sum[j] / cnt[j], cnt[j]);
                        fprintf(stderr, "\n");
                }
#endif
                for (size_t k = 0; k < std::min<size_t>(12, by.size()); ++k) {
#if TRI_TASKTIMES == 5
                        const uint32_t ti = b->ptasks[by[k].second];
#else
                        const uint32_t ti = b->sched[first + by[k].second];
#endif
                        const DevTask &tk = b->tasks[ti];
                        const DevQuery &q = b->plan[tk.slot];
                        std::string tt;
                        if (task_onepass(tk.kind)) {
                                const DevFused &z = b->fused[q.fused_idx];
                                char buf[96];
                                for (uint32_t j = 0; j < z.nslots; ++j) {
                                        snprintf(buf, sizeof buf, " %u(df %u%s)", z.term[j], b->ix->terms[z.term[j]].documents, z.plane[j] != PL_NONE ? " plane" : "");
                                        tt += buf;
                                }
                                snprintf(buf, sizeof buf, " nreq %u", z.nreq);
                                tt += buf;
                        } else
                        for (uint32_t j = 0; j < q.nterms; ++j) {
                                const uint32_t term = b->qterms[q.term_base + j] & QT_TERM;
                                char buf[96];
                                snprintf(buf, sizeof buf, " %s%u(df %u%s)", (b->qterms[q.term_base + j] & QT_GROUP) ? "|" : "", term, b->ix->terms[term].documents,
                                         (!b->qplane.empty() && b->qplane[q.term_base + j] != PL_NONE) ? " plane" : "");
                                tt += buf;
                        }
                        {
                                char buf[64];
                                snprintf(buf, sizeof buf, "  matches %u phrases %u", b->h_counts.empty() ? 0u : b->h_counts[ti], q.nphrases);
                                tt += buf;
                        }
                        fprintf(stderr, "   %.1f us (started %.1f us in)  ticket %u  tiles [%u, %u)  query %u:%s\n", (double)by[k].first / 100.0,
                                (double)(g_tt_host[8 * by[k].second] - t0) / 100.0, by[k].second, tk.tile_begin, tk.tile_end, q.qid, tt.c_str());
                        fprintf(stderr, "        last tile: lead decoded +%.1f us", ((double)g_tt_host[8 * by[k].second + 2] - (double)g_tt_host[8 * by[k].second]) / 100.0);
                        for (int j = 3; j < 8; ++j)
                                if (g_tt_host[8 * by[k].second + j])
                                        fprintf(stderr, "  term %d +%.1f", j - 2, ((double)g_tt_host[8 * by[k].second + j] - (double)g_tt_host[8 * by[k].second]) / 100.0);
                        fprintf(stderr, "\n");
                }
        }
#endif
        float ms = 0;
        if (hipEventElapsedTime(&ms, b->ev0, b->ev1) == hipSuccess)
                b->info.last_run_ms = ms;
        b->info.dense_ms = b->info.pset_ms = b->info.probe_ms = b->info.cand_ms = b->info.fused_ms = b->info.phrase_ms = b->info.tree_ms = b->info.rest_ms = b->info.term_planes_ms = b->info.planes_ms = 0;
        if (!b->tasks.empty()) {
                if (hipEventElapsedTime(&ms, b->ev0, b->ev_pl) == hipSuccess)
                        b->info.term_planes_ms = ms; // includes the ticket memset that precedes it
                if (hipEventElapsedTime(&ms, b->ev_pl, b->ev_a) == hipSuccess)
                        b->info.dense_ms = ms;
                if (hipEventElapsedTime(&ms, b->ev_a, b->ev_s) == hipSuccess)
                        b->info.pset_ms = ms;
                if (hipEventElapsedTime(&ms, b->ev_s, b->ev_r) == hipSuccess)
                        b->info.probe_ms = ms;
                if (hipEventElapsedTime(&ms, b->ev_r, b->ev_b) == hipSuccess)
                        b->info.cand_ms = ms;
                if (hipEventElapsedTime(&ms, b->ev_b, b->ev_c) == hipSuccess)
                        b->info.fused_ms = ms;
                if (hipEventElapsedTime(&ms, b->ev_c, b->ev_k) == hipSuccess)
                        b->info.planes_ms = ms;
                if (hipEventElapsedTime(&ms, b->ev_k, b->ev_p) == hipSuccess)
                        b->info.phrase_ms = ms;
                if (hipEventElapsedTime(&ms, b->ev_p, b->ev_t) == hipSuccess)
                        b->info.tree_ms = ms;
                if (hipEventElapsedTime(&ms, b->ev_t, b->ev1) == hipSuccess)
                        b->info.rest_ms = ms;
        }
        b->h_counts.resize(b->tasks.size());
        if (!b->tasks.empty())
                HIP_TRY(hipMemcpy(b->h_counts.data(), b->d_counts, b->tasks.size() * 4, hipMemcpyDeviceToHost));
        uint64_t m = 0;
        b->h_query_counts.assign(b->plan.size(), 0);
        uint64_t m_dense = 0, m_pset = 0, m_probe = 0, m_fused = 0, out_fused = 0, out_planes = 0, m_tree = 0;
        uint64_t outb_dense = 0, outb_pset = 0; // bytes the bitmap-window queries' results take in the form they are delivered in (a bitmap: its words)
        for (size_t sidx = 0; sidx < b->plan.size(); ++sidx) {
                const DevQuery &q = b->plan[sidx];
                for (uint32_t t = 0; t < q.ntasks; ++t)
                        b->h_query_counts[sidx] += b->h_counts[q.first_task + t];
                if (q.qid == 0xffffffffu) // (a hidden phrase query: its matches are a leaf of a TASK_TREE query, not a result)
                        continue;
                m += b->h_query_counts[sidx];
                if (q.ntasks && b->tasks[q.first_task].kind == TASK_TREE)
                        m_tree += b->h_query_counts[sidx];
                if (q.ntasks && b->tasks[q.first_task].kind == TASK_DENSE) {
                        m_dense += b->h_query_counts[sidx];
                        outb_dense += q.form == RESULT_BITMAP ? 4ull * q.out_cap : 4 * b->h_query_counts[sidx];
                }
                if (q.ntasks && b->tasks[q.first_task].kind == TASK_PSET) {
                        m_pset += b->h_query_counts[sidx];
                        outb_pset += q.form == RESULT_BITMAP ? 4ull * q.out_cap : 4 * b->h_query_counts[sidx];
                }
                if (q.ntasks && b->tasks[q.first_task].kind == TASK_PROBE)
                        m_probe += b->h_query_counts[sidx];
                if (q.ntasks && task_onepass(b->tasks[q.first_task].kind)) {
                        m_fused += b->h_query_counts[sidx]; // (every one-pass kind, k_planes' included)
                        (b->tasks[q.first_task].kind >= TASK_PLANES ? out_planes : out_fused) += // (one-pass kinds only: TASK_PLANES / TASK_PLANES8 are the last two of them)
                                q.out_cap ? 4 * b->h_query_counts[sidx] : 8 * std::min<uint64_t>(b->h_query_counts[sidx], b->topk); // (docIDs of a DocumentsOnly general tree)
                }
        }
        b->info.dense_algorithmic_bytes = b->term_bytes_dense + 4 * m_dense;
        b->info.pset_algorithmic_bytes = b->term_bytes_pset + 4 * m_pset;
        b->info.pset_queries = b->pset_queries;
        b->info.probe_algorithmic_bytes = b->term_bytes_probe + 4 * m_probe;
        b->info.probe_queries = b->probe_queries;
        b->info.cand_queries = b->cand_queries;
        b->info.cand_algorithmic_bytes = (b->term_bytes - b->term_bytes_dense - b->term_bytes_pset - b->term_bytes_probe - b->term_bytes_fused - b->term_bytes_planes - b->term_bytes_phrase_hits) + 4 * (m - m_dense - m_pset - m_probe - m_fused - m_tree);
        b->info.planes_algorithmic_bytes = b->term_bytes_planes + out_planes; // docbytes + 8 B x min(matches, K), per query — the lists
                                                                              // the batch's queries share are nevertheless decoded once per launch
        // (term_planes_decoded_bytes: set by tri_batch_run — the list bytes of the plane rows THAT run had to build; 0 once the index's cache holds them)
        b->info.phrase_algorithmic_bytes = b->term_bytes_phrase_hits; // what k_phrase streams by the SURVEY §8(d) count: the hit bytes of the phrases' terms
        b->info.phrase_queries = 0;
        for (const DevQuery &q : b->plan)
                b->info.phrase_queries += q.nphrases != 0;
        b->info.cand_needed_bytes = b->cand_needed_term_bytes ? b->cand_needed_term_bytes + 4 * (m - m_dense - m_pset - m_fused - m_tree) : 0; // (the candidate-tile AND the probe queries: what a perfect gallop reads)
        b->info.fused_algorithmic_bytes = b->term_bytes_fused + out_fused; // docbytes + 8 B x min(matches, K)
        b->info.matches = m;
        if (b->distinct_bytes) { // (option account_needed_bytes: the batch-level bound — every distinct list once + every output once)
                const uint64_t m_cand = m - m_dense - m_pset - m_probe - m_fused - m_tree;
                const bool sc = b->flags & TRI_FLAG_ACCUMULATED_SCORE;
                uint64_t out_legacy_dense = outb_dense, out_legacy_pset = outb_pset, out_legacy_probe = 4 * m_probe, out_legacy_cand = 4 * m_cand; // (the bound counts a result in the form it is delivered in)
                if (sc && b->topk) { // (queries matched by k_and_dense / k_psets / k_and of a top-K batch deliver 8 B x min(matches, K))
                        out_legacy_dense = out_legacy_pset = out_legacy_probe = out_legacy_cand = 0;
                        for (size_t sidx = 0; sidx < b->plan.size(); ++sidx) {
                                const DevQuery &q = b->plan[sidx];
                                if (!q.ntasks || task_onepass(b->tasks[q.first_task].kind) || b->tasks[q.first_task].kind == TASK_TREE || q.qid == 0xffffffffu)
                                        continue;
                                const uint32_t kd = b->tasks[q.first_task].kind;
                                (kd == TASK_DENSE ? out_legacy_dense : kd == TASK_PSET ? out_legacy_pset : kd == TASK_PROBE ? out_legacy_probe : out_legacy_cand) += 8 * std::min<uint64_t>(b->h_query_counts[sidx], b->topk);
                        }
                }
                b->info.pset_bound_bytes = b->distinct_bytes_kind[TASK_PSET] + out_legacy_pset;
                b->info.probe_bound_bytes = b->distinct_bytes_kind[TASK_PROBE] + out_legacy_probe;
                b->info.dense_bound_bytes = b->distinct_bytes_kind[TASK_DENSE] + out_legacy_dense;
                b->info.cand_bound_bytes = b->distinct_bytes_kind[TASK_CAND] + out_legacy_cand;
                b->info.fused_bound_bytes = b->distinct_bytes_kind[TASK_FUSED] + b->distinct_bytes_kind[TASK_FUSED16] + b->distinct_bytes_kind[TASK_FUSED_GEN] + out_fused;
                b->info.planes_bound_bytes = b->distinct_bytes_kind[TASK_PLANES] + b->distinct_bytes_kind[TASK_PLANES8] + out_planes;
                b->info.phrase_bound_bytes = b->distinct_bytes_kind[TASK_KINDS];
                b->info.bound_bytes = b->distinct_bytes + out_legacy_dense + out_legacy_pset + out_legacy_probe + out_legacy_cand + out_fused + out_planes;
        }
        if (b->flags & TRI_FLAG_ACCUMULATED_SCORE) {
                uint64_t outb = 0; // 8 B x min(matches, K) per query
                for (uint64_t c : b->h_query_counts)
                        outb += 8 * std::min<uint64_t>(c, b->topk);
                b->info.algorithmic_bytes = b->term_bytes + outb;
        } else
                b->info.algorithmic_bytes = b->term_bytes + 4 * m; // docbytes + 4 B per match (docs-only)
        if ((b->flags & TRI_FLAG_MATCHED_TERMS) && !b->tasks.empty()) {
                // the COUNT pass left every task's hit total: turn them into pool offsets (the pool is packed: task after task in
                // query order, inside a task match-major then term-minor), then the WRITE pass fills in the positions
                const size_t nt = b->tasks.size();
                std::vector<uint32_t> th(nt);
                HIP_TRY(hipMemcpy(th.data(), b->d_task_hits, nt * 4, hipMemcpyDeviceToHost));
                b->h_task_pos_base.assign(nt + 1, 0);
                for (size_t i = 0; i < nt; ++i)
                        b->h_task_pos_base[i + 1] = b->h_task_pos_base[i] + th[i];
                const size_t total = b->h_task_pos_base[nt];
                if (total + 64 > b->rich_pool_cap) {
                        hipFree(b->d_rich_pool);
                        b->d_rich_pool = nullptr;
                        b->rich_pool_cap = total + total / 8 + 64;
                        HIP_TRY(hipMalloc((void **)&b->d_rich_pool, b->rich_pool_cap * 2));
                        if (b->flags & TRI_FLAG_HIT_PAYLOADS) {
                                hipFree(b->d_rich_plen);
                                hipFree(b->d_rich_payload);
                                b->d_rich_plen = nullptr;
                                b->d_rich_payload = nullptr;
                                HIP_TRY(hipMalloc((void **)&b->d_rich_plen, b->rich_pool_cap));
                                HIP_TRY(hipMalloc((void **)&b->d_rich_payload, b->rich_pool_cap * 8));
                        }
                }
                HIP_TRY(hipMemcpy(b->d_task_pos_base, b->h_task_pos_base.data(), (nt + 1) * 8, hipMemcpyHostToDevice));
                HIP_TRY(hipMemsetAsync(b->d_ticket + 40, 0, 4, dev->stream));
                const uint32_t n = (uint32_t)nt;
                if (b->ix->codec == TRI_CODEC_LUCENE)
                        hipLaunchKernelGGL((k_rich<CODEC_LUCENE, true>), dim3(std::min<uint32_t>(n, (uint32_t)dev->cus * 3)), dim3(AND_WG), 0, dev->stream, b->ix->d_index, b->ix->d_hits,
                                           b->ix->d_blk_hits, b->ix->d_hdir, b->ix->d_blk_last, b->ix->d_blk_off, b->ix->d_terms, b->d_plan, b->d_tasks, b->d_sched, b->d_sterms, n,
                                           b->d_ticket + 40, b->d_out, b->d_counts, b->rich_R, b->d_rich_present, b->d_rich_freq, b->d_task_hits,
                                           (const uint64_t *)b->d_task_pos_base, b->d_rich_pool, (const uint32_t *)b->d_rich_allow, b->d_rich_plen, b->d_rich_payload);
                else
                        hipLaunchKernelGGL((k_rich<CODEC_GOOGLE, true>), dim3(std::min<uint32_t>(n, (uint32_t)dev->cus * 3)), dim3(AND_WG), 0, dev->stream, b->ix->d_index, b->ix->d_hits,
                                           b->ix->d_blk_hits, b->ix->d_hdir, b->ix->d_blk_last, b->ix->d_blk_off, b->ix->d_terms, b->d_plan, b->d_tasks, b->d_sched, b->d_sterms, n,
                                           b->d_ticket + 40, b->d_out, b->d_counts, b->rich_R, b->d_rich_present, b->d_rich_freq, b->d_task_hits,
                                           (const uint64_t *)b->d_task_pos_base, b->d_rich_pool, (const uint32_t *)b->d_rich_allow, b->d_rich_plen, b->d_rich_payload);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipStreamSynchronize(dev->stream));
                b->info.algorithmic_bytes += 2 * total + 4 * m; // + the positions handed over and a present mask per match
        }
        b->synced = true;
        return TRI_OK;
}

extern "C" int tri_batch_query_terms(tri_batch *b, size_t q, uint32_t *terms, uint32_t *nterms) {
        if (!b || !terms || !nterms || q >= b->nq)
                return fail(TRI_ERR_INVALID, "bad argument");
        if (!(b->flags & TRI_FLAG_MATCHED_TERMS))
                return fail(TRI_ERR_INVALID, "not a TRI_FLAG_MATCHED_TERMS batch");
        const uint32_t slot = b->slot_of_query[q];
        *nterms = 0;
        if (slot == UINT32_MAX)
                return TRI_OK;
        const DevQuery &dq = b->plan[slot];
        for (uint32_t k = 0; k < dq.nscore; ++k)
                terms[k] = b->sterms[dq.score_base + k];
        *nterms = dq.nscore;
        return TRI_OK;
}

extern "C" int tri_batch_matched_terms(tri_batch *b, size_t q, uint32_t *present, uint16_t *freq, uint16_t *positions, size_t pos_cap, size_t *npos) {
        if (!b || !npos || q >= b->nq)
                return fail(TRI_ERR_INVALID, "bad argument");
        if (!(b->flags & TRI_FLAG_MATCHED_TERMS))
                return fail(TRI_ERR_INVALID, "not a TRI_FLAG_MATCHED_TERMS batch");
        if (!b->synced)
                return fail(TRI_ERR_INVALID, "tri_batch_sync first");
        *npos = 0;
        const uint32_t slot = b->slot_of_query[q];
        if (slot == UINT32_MAX)
                return TRI_OK;
        const DevQuery &dq = b->plan[slot];
        tri_dev *dev = b->ix->dev;
        HIP_TRY(hipSetDevice(dev->device));
        // the query's tasks are consecutive, so its hits are one contiguous run of the pool
        const uint64_t p0 = b->h_task_pos_base[dq.first_task], p1 = b->h_task_pos_base[dq.first_task + dq.ntasks];
        *npos = (size_t)(p1 - p0);
        if (positions) {
                if (pos_cap < *npos)
                        return fail(TRI_ERR_INVALID, "positions need %zu slots, %zu given", *npos, pos_cap);
                if (*npos)
                        HIP_TRY(hipMemcpyAsync(positions, b->d_rich_pool + p0, *npos * 2, hipMemcpyDeviceToHost, dev->stream));
        }
        // per-match rows live at the tasks' out[] slots; freq rows are R wide on the device, nscore wide for the caller
        size_t w = 0;
        std::vector<uint16_t> rows;
        for (uint32_t t = 0; t < dq.ntasks; ++t) {
                const uint32_t c = b->h_counts[dq.first_task + t];
                if (!c)
                        continue;
                const uint64_t off = b->tasks[dq.first_task + t].out_off;
                if (present)
                        HIP_TRY(hipMemcpyAsync(present + w, b->d_rich_present + off, (size_t)c * 4, hipMemcpyDeviceToHost, dev->stream));
                if (freq) {
                        if (b->rich_R == dq.nscore)
                                HIP_TRY(hipMemcpyAsync(freq + w * dq.nscore, b->d_rich_freq + off * b->rich_R, (size_t)c * 2 * b->rich_R, hipMemcpyDeviceToHost, dev->stream));
                        else {
                                rows.resize((size_t)c * b->rich_R);
                                HIP_TRY(hipMemcpy(rows.data(), b->d_rich_freq + off * b->rich_R, (size_t)c * 2 * b->rich_R, hipMemcpyDeviceToHost));
                                for (size_t i = 0; i < c; ++i)
                                        for (uint32_t k = 0; k < dq.nscore; ++k)
                                                freq[(w + i) * dq.nscore + k] = rows[i * b->rich_R + k];
                        }
                }
                w += c;
        }
        HIP_TRY(hipStreamSynchronize(dev->stream));
        return TRI_OK;
}

// the payloads of query q's hits, parallel to the positions tri_batch_matched_terms returns (same order, same count)
extern "C" int tri_batch_matched_payloads(tri_batch *b, size_t q, uint8_t *lens, uint64_t *payloads, size_t cap, size_t *n) {
        if (!b || !n || q >= b->nq)
                return fail(TRI_ERR_INVALID, "bad argument");
        if (!(b->flags & TRI_FLAG_MATCHED_TERMS) || !(b->flags & TRI_FLAG_HIT_PAYLOADS))
                return fail(TRI_ERR_INVALID, "not a TRI_FLAG_MATCHED_TERMS | TRI_FLAG_HIT_PAYLOADS batch");
        if (!b->synced)
                return fail(TRI_ERR_INVALID, "tri_batch_sync first");
        *n = 0;
        const uint32_t slot = b->slot_of_query[q];
        if (slot == UINT32_MAX)
                return TRI_OK;
        const DevQuery &dq = b->plan[slot];
        const uint64_t p0 = b->h_task_pos_base[dq.first_task], p1 = b->h_task_pos_base[dq.first_task + dq.ntasks];
        *n = (size_t)(p1 - p0);
        if (!lens && !payloads)
                return TRI_OK;
        if (cap < *n)
                return fail(TRI_ERR_INVALID, "payloads need %zu slots, %zu given", *n, cap);
        HIP_TRY(hipSetDevice(b->ix->dev->device));
        if (*n && lens)
                HIP_TRY(hipMemcpy(lens, b->d_rich_plen + p0, *n, hipMemcpyDeviceToHost));
        if (*n && payloads)
                HIP_TRY(hipMemcpy(payloads, b->d_rich_payload + p0, *n * 8, hipMemcpyDeviceToHost));
        return TRI_OK;
}

extern "C" int tri_batch_get_info(const tri_batch *b, tri_batch_info *info) {
        if (!b || !info)
                return fail(TRI_ERR_INVALID, "null argument");
        *info = b->info;
        return TRI_OK;
}

extern "C" int tri_batch_query_status(const tri_batch *b, int32_t *status) {
        if (!b || !status)
                return fail(TRI_ERR_INVALID, "null argument");
        for (size_t q = 0; q < b->nq; ++q)
                status[q] = b->qstatus[q];
        return TRI_OK;
}

extern "C" int tri_batch_match_counts(tri_batch *b, uint64_t *counts) {
        if (!b || !counts)
                return fail(TRI_ERR_INVALID, "null argument");
        if (!b->synced)
                return fail(TRI_ERR_INVALID, "tri_batch_sync first");
        for (size_t q = 0; q < b->nq; ++q)
                counts[q] = b->slot_of_query[q] == UINT32_MAX ? 0 : b->h_query_counts[b->slot_of_query[q]];
        return TRI_OK;
}

extern "C" int tri_batch_docset(tri_batch *b, size_t q, uint32_t *out, size_t cap, size_t *n) {
        if (!b || !n || q >= b->nq)
                return fail(TRI_ERR_INVALID, "bad argument");
        if (!b->synced)
                return fail(TRI_ERR_INVALID, "tri_batch_sync first");
        const uint32_t slot = b->slot_of_query[q];
        *n = slot == UINT32_MAX ? 0 : b->h_query_counts[slot];
        if (!*n || !out)
                return TRI_OK;
        if (b->plan[slot].ntasks && !b->plan[slot].out_cap && task_onepass(b->tasks[b->plan[slot].first_task].kind))
                return fail(TRI_ERR_INVALID, "query %zu ran through the one-pass scored kernel: an AccumulatedScore top-K batch keeps top-K lists and match counts, not docID sets (use topk == 0 or DocumentsOnly)", q);
        if (cap < *n)
                return fail(TRI_ERR_INVALID, "docset needs %zu slots, %zu given", *n, cap);
        HIP_TRY(hipSetDevice(b->ix->dev->device));
        // the docID set is the in-order concatenation of the query's task segments
        const DevQuery &dq = b->plan[slot];
        if (dq.form == RESULT_BITMAP) { // one bit per document (dev_structs.hpp): the region's words come over as they are, the docIDs are written out here
                const uint32_t w_lo = b->tasks[dq.first_task].tile_begin, w_hi = b->tasks[dq.first_task + dq.ntasks - 1].tile_end;
                std::vector<uint32_t> words((size_t)(w_hi - w_lo) * SPAN_WORDS);
                HIP_TRY(hipMemcpy(words.data(), b->d_out + dq.out_off, words.size() * 4, hipMemcpyDeviceToHost));
                size_t w = 0;
                for (size_t i = 0; i < words.size(); ++i)
                        for (uint32_t m = words[i]; m; m &= m - 1u) {
                                if (w == *n)
                                        return fail(TRI_ERR_INTERNAL, "query %zu: its bitmap holds more documents than its tasks counted", q);
                                out[w++] = (uint32_t)(((size_t)w_lo * SPAN_WORDS + i) * 32u + (uint32_t)__builtin_ctz(m));
                        }
                if (w != *n)
                        return fail(TRI_ERR_INTERNAL, "query %zu: its bitmap holds %zu documents, its tasks counted %zu", q, w, *n);
                return TRI_OK;
        }
        // (the batch is synced: its results are complete — the copies go on the read-back stream and wait for nothing queued behind the batch)
        DevLock g(b->ix->dev->mu);
        size_t w = 0;
        for (uint32_t t = 0; t < dq.ntasks; ++t) {
                const uint32_t c = b->h_counts[dq.first_task + t];
                if (!c)
                        continue;
                const uint32_t *src = b->d_out + b->tasks[dq.first_task + t].out_off;
                HIP_TRY(hipMemcpyAsync(out + w, src, (size_t)c * 4, hipMemcpyDeviceToHost, b->ix->dev->stream_rb));
                w += c;
        }
        HIP_TRY(hipStreamSynchronize(b->ix->dev->stream_rb));
        return TRI_OK;
}

// The docID set of query q as the engine holds it when the planner chose the bitmap form for it (dev_structs.hpp RESULT_BITMAP: DocumentsOnly
// unions / conjunctions of head terms): *first_doc = the docID of bit 0 of words[0] (a multiple of 32), *nwords = the words that follow — bit j of
// word i: document *first_doc + 32 i + j matches.  words == NULL: only *form (0 docIDs: use tri_batch_docset; 1 bitmap), *first_doc, *nwords.
extern "C" int tri_batch_docset_bitmap(tri_batch *b, size_t q, int *form, uint32_t *words, size_t cap, uint32_t *first_doc, size_t *nwords) {
        if (!b || !form || !first_doc || !nwords || q >= b->nq)
                return fail(TRI_ERR_INVALID, "bad argument");
        if (!b->synced)
                return fail(TRI_ERR_INVALID, "tri_batch_sync first");
        const uint32_t slot = b->slot_of_query[q];
        *form = 0, *first_doc = 0, *nwords = 0;
        if (slot == UINT32_MAX || b->plan[slot].form != RESULT_BITMAP)
                return TRI_OK;
        const DevQuery &dq = b->plan[slot];
        const uint32_t w_lo = b->tasks[dq.first_task].tile_begin, w_hi = b->tasks[dq.first_task + dq.ntasks - 1].tile_end;
        *form = 1;
        *first_doc = w_lo * SPAN_BITS;
        *nwords = (size_t)(w_hi - w_lo) * SPAN_WORDS;
        if (!words)
                return TRI_OK;
        if (cap < *nwords)
                return fail(TRI_ERR_INVALID, "bitmap needs %zu words, %zu given", *nwords, cap);
        HIP_TRY(hipSetDevice(b->ix->dev->device));
        HIP_TRY(hipMemcpy(words, b->d_out + dq.out_off, *nwords * 4, hipMemcpyDeviceToHost));
        return TRI_OK;
}

extern "C" int tri_batch_docset_hashes(tri_batch *b, uint64_t *hashes) {
        if (!b || !hashes)
                return fail(TRI_ERR_INVALID, "null argument");
        if (!b->synced)
                return fail(TRI_ERR_INVALID, "tri_batch_sync first");
        tri_dev *dev = b->ix->dev;
        HIP_TRY(hipSetDevice(dev->device));
        const uint32_t n = (uint32_t)b->plan.size();
        if ((b->n_fused + b->n_fused16 + b->n_fusedgen + b->n_planes + b->n_planes8) && (b->flags & TRI_FLAG_ACCUMULATED_SCORE)) // (DocumentsOnly: the one-pass kernel's tasks wrote their matches)
                return fail(TRI_ERR_INVALID, "the batch holds queries that ran through the one-pass scored kernel: their docID sets are not materialised");
        std::vector<uint64_t> h(n);
        if (n) {
                if (!b->d_hashes)
                        HIP_TRY(hipMalloc((void **)&b->d_hashes, (size_t)n * 8));
                DevLock g(dev->mu);
                hipLaunchKernelGGL(k_hash_docsets, dim3((n + 63) / 64), dim3(64), 0, dev->stream_rb, b->d_plan, b->d_tasks, b->d_counts, n, b->d_out,
                                   b->d_hashes);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipStreamSynchronize(dev->stream_rb));
                HIP_TRY(hipMemcpy(h.data(), b->d_hashes, (size_t)n * 8, hipMemcpyDeviceToHost));
        }
        for (size_t q = 0; q < b->nq; ++q)
                hashes[q] = b->slot_of_query[q] == UINT32_MAX ? 1469598103934665603ull : h[b->slot_of_query[q]];
        return TRI_OK;
}

extern "C" int tri_batch_topk(tri_batch *b, uint32_t *docids, float *scores, uint32_t *counts) {
        if (!b || !docids || !scores || !counts)
                return fail(TRI_ERR_INVALID, "null argument");
        if (!(b->flags & TRI_FLAG_ACCUMULATED_SCORE))
                return fail(TRI_ERR_INVALID, "batch was not created with TRI_FLAG_ACCUMULATED_SCORE");
        if (!b->synced)
                return fail(TRI_ERR_INVALID, "tri_batch_sync first");
        HIP_TRY(hipSetDevice(b->ix->dev->device));
        HIP_TRY(hipMemcpy(docids, b->d_top_docs, b->nq * b->topk * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(scores, b->d_top_scores, b->nq * b->topk * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(counts, b->d_top_counts, b->nq * 4, hipMemcpyDeviceToHost));
        return TRI_OK;
}

extern "C" int tri_batch_scores(tri_batch *b, size_t q, double *out, size_t cap, size_t *n) {
        if (!b || !n || q >= b->nq)
                return fail(TRI_ERR_INVALID, "bad argument");
        if (!(b->flags & TRI_FLAG_ACCUMULATED_SCORE) || b->topk)
                return fail(TRI_ERR_INVALID, "per-match scores are kept only for AccumulatedScoreScheme batches created with topk == 0");
        if (!b->synced)
                return fail(TRI_ERR_INVALID, "tri_batch_sync first");
        const uint32_t slot = b->slot_of_query[q];
        *n = slot == UINT32_MAX ? 0 : b->h_query_counts[slot];
        if (!*n || !out)
                return TRI_OK;
        if (cap < *n)
                return fail(TRI_ERR_INVALID, "scores need %zu slots, %zu given", *n, cap);
        HIP_TRY(hipSetDevice(b->ix->dev->device));
        const DevQuery &dq = b->plan[slot];
        DevLock g(b->ix->dev->mu);
        size_t w = 0;
        for (uint32_t t = 0; t < dq.ntasks; ++t) {
                const uint32_t c = b->h_counts[dq.first_task + t];
                if (!c)
                        continue;
                HIP_TRY(hipMemcpyAsync(out + w, b->d_all_scores + b->tasks[dq.first_task + t].out_off, (size_t)c * 8, hipMemcpyDeviceToHost, b->ix->dev->stream_rb));
                w += c;
        }
        HIP_TRY(hipStreamSynchronize(b->ix->dev->stream_rb));
        return TRI_OK;
}

// (forms == nullptr: every set as ascending docIDs — tri_batch_docsets; else tri_batch_docsets_mixed: a RESULT_BITMAP query's region goes out as its words)
static int docsets_deliver(tri_batch *b, uint32_t *out, size_t cap, uint64_t *offsets, uint32_t *forms, const char *fn) {
        if (!b || !offsets)
                return fail(TRI_ERR_INVALID, "%s: null argument", fn);
        if (!b->synced)
                return fail(TRI_ERR_INVALID, "tri_batch_sync first");
        tri_dev *dev = b->ix->dev;
        const size_t nslots = b->plan.size();
        std::vector<uint64_t> slot_off(nslots + 1, 0);
        uint64_t total = 0;
        for (size_t q = 0; q < b->nq; ++q) {
                const uint32_t slot = b->slot_of_query[q];
                offsets[q] = total;
                if (forms)
                        forms[q] = RESULT_DOCIDS;
                if (slot == UINT32_MAX)
                        continue;
                const DevQuery &dq = b->plan[slot];
                if (dq.ntasks && !dq.out_cap && task_onepass(b->tasks[dq.first_task].kind) && b->h_query_counts[slot])
                        return fail(TRI_ERR_INVALID, "query %zu ran through the one-pass scored kernel: an AccumulatedScore top-K batch keeps top-K lists and match counts, not docID sets", q);
                slot_off[slot] = total;
                if (forms && dq.form == RESULT_BITMAP) {
                        forms[q] = RESULT_BITMAP;
                        total += dq.out_cap; // (the region's words: one bit per document of the query's docID range, from document 0)
                } else
                        total += b->h_query_counts[slot];
        }
        offsets[b->nq] = total;
        if (!out || !total)
                return TRI_OK;
        if (cap < total)
                return fail(TRI_ERR_INVALID, "the docID sets need %llu slots, %zu given", (unsigned long long)total, cap);
        HIP_TRY(hipSetDevice(dev->device));
        uint32_t *d_flat = nullptr;
        uint64_t *d_slot_off = nullptr;
        hipError_t e;
        {
                // (the device lock covers the pool and the enqueues; the WAIT for the copy stands outside it: a thread that compiles the next batch, or runs
                //  one, is not held up by the seconds a large delivery spends on PCIe — the read-back stream's work overlaps the engine stream's)
                DevLock g(dev->mu);
                HIP_TRY(pool_alloc(dev, (void **)&d_flat, (total + 64) * 4));
                e = pool_alloc(dev, (void **)&d_slot_off, (nslots + 1) * 8 + POOL_MIN_BYTES);
                if (e == hipSuccess)
                        e = hipMemcpyAsync(d_slot_off, slot_off.data(), (nslots + 1) * 8, hipMemcpyHostToDevice, dev->stream_rb); // (pageable source: staged before the call returns)
                if (e == hipSuccess) {
                        hipLaunchKernelGGL(k_deliver_docsets, dim3((uint32_t)b->tasks.size()), dim3(256), 0, dev->stream_rb, (const DevQuery *)b->d_plan, (const DevTask *)b->d_tasks,
                                           (const uint32_t *)b->d_counts, (const uint32_t *)b->d_out, (const uint64_t *)d_slot_off, d_flat, forms ? 1u : 0u);
                        e = hipGetLastError();
                }
                if (e == hipSuccess)
                        e = hipMemcpyAsync(out, d_flat, total * 4, hipMemcpyDeviceToHost, dev->stream_rb);
        }
        if (e == hipSuccess)
                e = hipStreamSynchronize(dev->stream_rb);
        pool_free(dev, d_flat);
        pool_free(dev, d_slot_off);
        HIP_TRY(e);
        return TRI_OK;
}

// ---- every query's docID set in ONE call: what a caller that replays MatchedIndexDocumentsFilter::consider(const docid_t *, size_t) (matches.h:161-165)
//      per query needs on the host.  out[offsets[q] .. offsets[q + 1]) = query q's ascending docIDs, queries in the caller's order; the sets are
//      gathered on the device into one contiguous buffer (k_deliver_docsets: the tasks' segments in order, bitmap-form results expanded) and come
//      over in a single copy — at a pinned `out` that is PCIe's rate, not a copy and a synchronisation per task segment
extern "C" int tri_batch_docsets(tri_batch *b, uint32_t *out, size_t cap, uint64_t *offsets) { return docsets_deliver(b, out, cap, offsets, nullptr, "tri_batch_docsets"); }

// ... and each set in the FORM THE ENGINE HOLDS IT: forms[q] = RESULT_DOCIDS: ascending docIDs as above; RESULT_BITMAP (a union / conjunction of head terms that matches one
// document in 32 or more): out[offsets[q] .. offsets[q + 1]) = the words of a bitmap over the query's docID range — bit j of word i = document 32 i + j matches.  A dense set
// crosses PCIe as a bit per document instead of four bytes per match (and is not expanded on the device first); the consumer expands it, or hands the bitmap on
extern "C" int tri_batch_docsets_mixed(tri_batch *b, uint32_t *out, size_t cap, uint64_t *offsets, uint32_t *forms) {
        if (!forms)
                return fail(TRI_ERR_INVALID, "tri_batch_docsets_mixed: null forms");
        return docsets_deliver(b, out, cap, offsets, forms, "tri_batch_docsets_mixed");
}

extern "C" int tri_batch_counts_device(tri_batch *b, void **counts) {
        if (!b || !counts)
                return fail(TRI_ERR_INVALID, "null argument");
        *counts = b->d_qcounts;
        return TRI_OK;
}

extern "C" int tri_batch_topk_device(tri_batch *b, void **docids, void **scores, void **counts) {
        if (!b || !docids || !scores || !counts)
                return fail(TRI_ERR_INVALID, "null argument");
        if (!(b->flags & TRI_FLAG_ACCUMULATED_SCORE))
                return fail(TRI_ERR_INVALID, "batch was not created with TRI_FLAG_ACCUMULATED_SCORE");
        *docids = b->d_top_docs;
        *scores = b->d_top_scores;
        *counts = b->d_top_counts;
        return TRI_OK;
}

// ------------------------------------------------------------------------------------------ collections of segments
// IndexSourcesCollection (index_source.cpp:3-30): a query runs over every source of the collection, each source masked by what the
// newer ones update (tri_index_set_masked), and the application's filter sees the matches of all of them (exec_query per source,
// exec.h:57-62).  A tri_cbatch borrows one tri_batch per source — the same queries, term indices resolved per source — runs them back
// to back on the engine stream and merges on the device: match counts add up, top-K lists merge K-way from the parts' partial lists.
struct tri_cbatch {
        std::vector<tri_batch *> parts;
        std::vector<uint32_t *> d_slots; // per part: caller query -> plan slot
        DevSource *d_src = nullptr;
        uint32_t *d_top_docs = nullptr, *d_top_counts = nullptr;
        float *d_top_scores = nullptr;
        uint64_t *d_counts = nullptr;
        bool ran = false, synced = false;
        ~tri_cbatch() {
                if (!parts.empty())
                        hipSetDevice(parts[0]->ix->dev->device);
                for (auto p : d_slots)
                        hipFree(p);
                hipFree(d_src);
                hipFree(d_top_docs);
                hipFree(d_top_counts);
                hipFree(d_top_scores);
                hipFree(d_counts);
        }
};

extern "C" int tri_cbatch_create(tri_batch *const *parts, size_t n, tri_cbatch **out) {
        if (!parts || !n || !out)
                return fail(TRI_ERR_INVALID, "tri_cbatch_create: null argument");
        for (size_t i = 0; i < n; ++i) {
                if (!parts[i])
                        return fail(TRI_ERR_INVALID, "tri_cbatch_create: null part %zu", i);
                if (parts[i]->ix->dev != parts[0]->ix->dev || parts[i]->nq != parts[0]->nq || parts[i]->flags != parts[0]->flags ||
                    parts[i]->topk != parts[0]->topk)
                        return fail(TRI_ERR_INVALID, "tri_cbatch_create: part %zu differs from part 0 in device, query count, flags or topk", i);
        }
        tri_dev *dev = parts[0]->ix->dev;
        HIP_TRY(hipSetDevice(dev->device));
        auto c = std::make_unique<tri_cbatch>();
        c->parts.assign(parts, parts + n);
        const size_t nq = parts[0]->nq, k = parts[0]->topk;
        std::vector<DevSource> src(n);
        for (size_t i = 0; i < n; ++i) {
                uint32_t *d = nullptr;
                int rc;
                if ((rc = dev_upload(&d, parts[i]->slot_of_query)))
                        return rc;
                c->d_slots.push_back(d);
                src[i] = {parts[i]->d_plan, d, parts[i]->d_part_docs, parts[i]->d_part_scores, parts[i]->d_part_counts, parts[i]->d_qcounts};
        }
        int rc;
        if ((rc = dev_upload(&c->d_src, src)))
                return rc;
        HIP_TRY(hipMalloc((void **)&c->d_counts, (nq + 1) * 8));
        if ((parts[0]->flags & TRI_FLAG_ACCUMULATED_SCORE) && k) {
                HIP_TRY(hipMalloc((void **)&c->d_top_docs, (nq * k + 1) * 4));
                HIP_TRY(hipMalloc((void **)&c->d_top_scores, (nq * k + 1) * 4));
                HIP_TRY(hipMalloc((void **)&c->d_top_counts, (nq + 1) * 4));
        }
        *out = c.release();
        return TRI_OK;
}

extern "C" void tri_cbatch_destroy(tri_cbatch *c) { delete c; }

// per query: TRI_OK, or TRI_ERR_UNSUPPORTED when the planner left the query out of ANY part (its answer over the collection is then
// incomplete: the caller keeps its CPU path for that query, as with tri_batch_query_status)
extern "C" int tri_cbatch_query_status(const tri_cbatch *c, int32_t *status) {
        if (!c || !status)
                return fail(TRI_ERR_INVALID, "null argument");
        const size_t nq = c->parts[0]->nq;
        for (size_t q = 0; q < nq; ++q) {
                status[q] = TRI_OK;
                for (const tri_batch *p : c->parts)
                        if (p->qstatus[q] != TRI_OK)
                                status[q] = p->qstatus[q];
        }
        return TRI_OK;
}

extern "C" int tri_cbatch_run(tri_cbatch *c) {
        if (!c)
                return fail(TRI_ERR_INVALID, "null collection batch");
        for (tri_batch *p : c->parts)
                if (int rc = tri_batch_run(p))
                        return rc;
        tri_dev *dev = c->parts[0]->ix->dev;
        const uint32_t nq = (uint32_t)c->parts[0]->nq;
        const uint32_t k = c->d_top_docs ? c->parts[0]->topk : 0u;
        if (nq) {
                hipLaunchKernelGGL(k_topk_merge_sources, dim3(std::min<uint32_t>(nq, (uint32_t)dev->cus * 8)), dim3(AND_WG), 0, dev->stream, c->d_src,
                                   (uint32_t)c->parts.size(), nq, k, c->d_top_docs, c->d_top_scores, c->d_top_counts, c->d_counts);
                HIP_TRY(hipGetLastError());
        }
        c->ran = true;
        c->synced = false;
        return TRI_OK;
}

extern "C" int tri_cbatch_sync(tri_cbatch *c) {
        if (!c || !c->ran)
                return fail(TRI_ERR_INVALID, "tri_cbatch_sync: the collection batch has not been run");
        for (tri_batch *p : c->parts)
                if (int rc = tri_batch_sync(p))
                        return rc;
        HIP_TRY(hipStreamSynchronize(c->parts[0]->ix->dev->stream));
        c->synced = true;
        return TRI_OK;
}

extern "C" int tri_cbatch_match_counts(tri_cbatch *c, uint64_t *counts) {
        if (!c || !counts)
                return fail(TRI_ERR_INVALID, "null argument");
        if (!c->synced)
                return fail(TRI_ERR_INVALID, "tri_cbatch_sync first");
        HIP_TRY(hipSetDevice(c->parts[0]->ix->dev->device));
        HIP_TRY(hipMemcpy(counts, c->d_counts, c->parts[0]->nq * 8, hipMemcpyDeviceToHost));
        return TRI_OK;
}

extern "C" int tri_cbatch_topk(tri_cbatch *c, uint32_t *docids, float *scores, uint32_t *counts) {
        if (!c || !docids || !scores || !counts)
                return fail(TRI_ERR_INVALID, "null argument");
        if (!c->d_top_docs)
                return fail(TRI_ERR_INVALID, "the parts were not created with TRI_FLAG_ACCUMULATED_SCORE and topk >= 1");
        if (!c->synced)
                return fail(TRI_ERR_INVALID, "tri_cbatch_sync first");
        const size_t nq = c->parts[0]->nq, k = c->parts[0]->topk;
        HIP_TRY(hipSetDevice(c->parts[0]->ix->dev->device));
        HIP_TRY(hipMemcpy(docids, c->d_top_docs, nq * k * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(scores, c->d_top_scores, nq * k * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(counts, c->d_top_counts, nq * 4, hipMemcpyDeviceToHost));
        return TRI_OK;
}

// the docID set of query q over the collection: the sources' sets one after the other (each ascending; the sources are disjoint where
// the newer ones mask the older) — the order exec_query delivers them in when it is called source after source
extern "C" int tri_cbatch_docset(tri_cbatch *c, size_t q, uint32_t *out, size_t cap, size_t *n) {
        if (!c || !n)
                return fail(TRI_ERR_INVALID, "null argument");
        if (!c->synced)
                return fail(TRI_ERR_INVALID, "tri_cbatch_sync first");
        size_t total = 0;
        for (tri_batch *p : c->parts) {
                size_t m = 0;
                if (int rc = tri_batch_docset(p, q, nullptr, 0, &m))
                        return rc;
                total += m;
        }
        *n = total;
        if (!out)
                return TRI_OK;
        if (cap < total)
                return fail(TRI_ERR_INVALID, "docset needs %zu slots, %zu given", total, cap);
        size_t w = 0;
        for (tri_batch *p : c->parts) {
                size_t m = 0;
                if (int rc = tri_batch_docset(p, q, out + w, cap - w, &m))
                        return rc;
                w += m;
        }
        return TRI_OK;
}

// ------------------------------------------------------------------------------------------ multi-GPU result gather (RCCL)
// exec_query_par hands every source / shard its own result object and the caller combines them (exec.h:132-176).  With the queries
// sharded over one process per GPU, the fixed-shape result blocks of a batch — per-query match counts and, for top-K batches, the
// [nq][k] docID / score blocks and list lengths — are exchanged with ONE group of ncclAllGather calls on the engine stream, straight
// from the device buffers.  RCCL is bound at run time (dlopen): the library has no link-time dependency on it, and inside a process
// that already holds an RCCL (PyTorch's) the same one is used.
namespace {
        struct RcclApi {
                struct UniqueId {
                        char internal[128];
                };
                int (*GetUniqueId)(UniqueId *) = nullptr;
                int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
                int (*CommDestroy)(void *) = nullptr;
                int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
                int (*GroupStart)() = nullptr;
                int (*GroupEnd)() = nullptr;
                const char *(*GetErrorString)(int) = nullptr;
                bool ok = false;
        };
        RcclApi &rccl() {
                static RcclApi api;
                static bool tried = false;
                if (tried)
                        return api;
                tried = true;
                void *h = nullptr;
                for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                        if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)))
                                break;
                if (!h)
                        return api;
                api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
                api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
                api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
                api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
                api.GroupStart = (decltype(api.GroupStart))dlsym(h, "ncclGroupStart");
                api.GroupEnd = (decltype(api.GroupEnd))dlsym(h, "ncclGroupEnd");
                api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
                api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.GroupStart && api.GroupEnd;
                return api;
        }
        int nccl_fail(const char *what, int rc) { return fail(TRI_ERR_DEVICE, "%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error"); }
} // namespace

struct tri_comm {
        tri_dev *dev = nullptr;
        void *comm = nullptr;
        int rank = 0, nranks = 1;
        tri_allgather_fn custom = nullptr; // the caller's own transport instead of RCCL (tri_comm_create_custom)
        void *custom_user = nullptr;
};

extern "C" int tri_comm_unique_id(uint8_t id[128]) {
        if (!id)
                return fail(TRI_ERR_INVALID, "null argument");
        if (!rccl().ok)
                return fail(TRI_ERR_UNSUPPORTED, "librccl could not be loaded");
        RcclApi::UniqueId u;
        if (int rc = rccl().GetUniqueId(&u))
                return nccl_fail("ncclGetUniqueId", rc);
        memcpy(id, u.internal, 128);
        return TRI_OK;
}

extern "C" int tri_comm_create(tri_dev *dev, const uint8_t id[128], int rank, int nranks, tri_comm **out) {
        if (!dev || !id || !out || nranks < 1 || rank < 0 || rank >= nranks)
                return fail(TRI_ERR_INVALID, "tri_comm_create: bad argument");
        if (!rccl().ok)
                return fail(TRI_ERR_UNSUPPORTED, "librccl could not be loaded");
        HIP_TRY(hipSetDevice(dev->device));
        auto c = std::make_unique<tri_comm>();
        c->dev = dev;
        c->rank = rank;
        c->nranks = nranks;
        RcclApi::UniqueId u;
        memcpy(u.internal, id, 128);
        if (int rc = rccl().CommInitRank(&c->comm, nranks, u, rank))
                return nccl_fail("ncclCommInitRank", rc);
        *out = c.release();
        return TRI_OK;
}

extern "C" int tri_comm_create_custom(tri_dev *dev, int rank, int nranks, tri_allgather_fn allgather, void *user, tri_comm **out) {
        if (!dev || !out || !allgather || nranks < 1 || rank < 0 || rank >= nranks)
                return fail(TRI_ERR_INVALID, "tri_comm_create_custom: bad argument");
        auto c = std::make_unique<tri_comm>();
        c->dev = dev;
        c->rank = rank;
        c->nranks = nranks;
        c->custom = allgather;
        c->custom_user = user;
        *out = c.release();
        return TRI_OK;
}

extern "C" void tri_comm_destroy(tri_comm *c) {
        if (!c)
                return;
        if (c->comm && rccl().ok)
                rccl().CommDestroy(c->comm);
        delete c;
}

// every rank's blocks of batch b (same nq and topk on every rank) into [nranks][...] device buffers: counts_all u64[nranks][nq]; and for
// AccumulatedScore top-K batches docids_all u32[nranks][nq][k], scores_all f32[nranks][nq][k], topk_counts_all u32[nranks][nq] (NULL
// for the other modes).  Enqueued on the engine stream behind the batch's run; complete after tri_dev_sync / a stream wait.
extern "C" int tri_gather_results(tri_batch *b, tri_comm *c, void *counts_all, void *docids_all, void *scores_all, void *topk_counts_all) {
        if (!b || !c || !counts_all)
                return fail(TRI_ERR_INVALID, "null argument");
        if (b->ix->dev != c->dev)
                return fail(TRI_ERR_INVALID, "tri_gather_results: the batch and the communicator live on different device handles");
        const bool topk = (b->flags & TRI_FLAG_ACCUMULATED_SCORE) && b->topk;
        if (topk && (!docids_all || !scores_all || !topk_counts_all))
                return fail(TRI_ERR_INVALID, "tri_gather_results: a top-K batch needs all four receive buffers");
        tri_dev *dev = c->dev;
        HIP_TRY(hipSetDevice(dev->device));
        if (c->custom) { // the same blocks, the same [nranks][...] layout, over the caller's transport
                struct {
                        const void *send;
                        void *recv;
                        size_t bytes;
                } blocks[4] = {{b->d_qcounts, counts_all, b->nq * 8},
                               {topk ? b->d_top_docs : nullptr, docids_all, b->nq * b->topk * 4},
                               {topk ? b->d_top_scores : nullptr, scores_all, b->nq * b->topk * 4},
                               {topk ? b->d_top_counts : nullptr, topk_counts_all, b->nq * 4}};
                for (const auto &x : blocks)
                        if (x.send)
                                if (int rc = c->custom(c->custom_user, x.send, x.recv, x.bytes, (void *)dev->stream))
                                        return fail(TRI_ERR_DEVICE, "tri_gather_results: the caller's allgather returned %d", rc);
                return TRI_OK;
        }
        const RcclApi &R = rccl();
        const int U8 = 1; // ncclUint8: the blocks travel as bytes
        int rc = R.GroupStart();
        if (!rc)
                rc = R.AllGather(b->d_qcounts, counts_all, b->nq * 8, U8, c->comm, dev->stream);
        if (!rc && topk) {
                rc = R.AllGather(b->d_top_docs, docids_all, b->nq * b->topk * 4, U8, c->comm, dev->stream);
                if (!rc)
                        rc = R.AllGather(b->d_top_scores, scores_all, b->nq * b->topk * 4, U8, c->comm, dev->stream);
                if (!rc)
                        rc = R.AllGather(b->d_top_counts, topk_counts_all, b->nq * 4, U8, c->comm, dev->stream);
        }
        const int rc2 = R.GroupEnd();
        if (rc || rc2)
                return nccl_fail("ncclAllGather", rc ? rc : rc2);
        return TRI_OK;
}

// ------------------------------------------------------------------------------------------ write side
// Codecs::Google::Encoder (google_codec.cpp:9-176) on the device: postings in, the segment's `index` bytes and term table out —
// byte for byte what the reference's encoder writes for the same begin_term / begin_document / new_hit / end_document / end_term
// calls (payload-less hits).  See k_encode.hpp.
extern "C" int tri_encode_google(tri_dev *dev, const uint32_t *docs, const uint32_t *freqs, const uint16_t *positions, size_t npositions, const uint64_t *term_first,
                                 size_t nterms, uint8_t *index_out, size_t cap, size_t *index_len, tri_term *terms_out) {
        return tri_encode_google_payloads(dev, docs, freqs, positions, nullptr, nullptr, npositions, term_first, nterms, index_out, cap, index_len, terms_out);
}

// ---- the device side of the Google encoder, shared by tri_encode_google[_payloads] (postings uploaded from the host) and tri_commit_google
//      (postings sorted and gathered on the device): d.docs / d.freqs / d.pos (/ d.plens, d.payloads) hold np postings and nhits hits grouped by term as
//      term_first (host) says, validated by the caller
namespace {
struct EncBufs {
        uint32_t *docs = nullptr, *freqs = nullptr, *blk_first = nullptr, *blk_term = nullptr, *sizes = nullptr, *tails = nullptr;
        uint16_t *pos = nullptr;
        uint64_t *hit_off = nullptr, *term_first = nullptr, *blk_off = nullptr, *term_off = nullptr, *payloads = nullptr, *scan_sums = nullptr;
        uint64_t scan_cap = 0;
        uint8_t *out = nullptr, *plens = nullptr;
        ~EncBufs() {
                for (void *p : {(void *)docs, (void *)freqs, (void *)blk_first, (void *)blk_term, (void *)sizes, (void *)tails, (void *)pos, (void *)hit_off,
                                (void *)term_first, (void *)blk_off, (void *)term_off, (void *)out, (void *)payloads, (void *)plens, (void *)scan_sums})
                        hipFree(p);
        }
};
// exclusive scan of n u32 into u64 over the whole device: chunk sums, chunk bases, chunks (k_encode.hpp)
int enc_scan(tri_dev *dev, EncBufs &d, const uint32_t *in, uint64_t *outp, const uint64_t n) {
        const uint64_t nchunks = (n + ENC_SCAN_CHUNK - 1) / ENC_SCAN_CHUNK;
        if (nchunks <= 1) {
                hipLaunchKernelGGL(k_enc_scan, dim3(1), dim3(1024), 0, dev->stream, in, outp, n);
                return TRI_OK;
        }
        if (nchunks + 1 > d.scan_cap) {
                hipFree(d.scan_sums);
                d.scan_sums = nullptr;
                d.scan_cap = nchunks + 1;
                HIP_TRY(hipMalloc((void **)&d.scan_sums, d.scan_cap * 8));
        }
        hipLaunchKernelGGL(k_enc_scan_sums, dim3((uint32_t)nchunks), dim3(1024), 0, dev->stream, in, d.scan_sums, n);
        hipLaunchKernelGGL(k_enc_scan_bases, dim3(1), dim3(1024), 0, dev->stream, d.scan_sums, nchunks);
        hipLaunchKernelGGL(k_enc_scan_chunks, dim3((uint32_t)nchunks), dim3(1024), 0, dev->stream, in, (const uint64_t *)d.scan_sums, outp, n);
        return TRI_OK;
}
int encode_google_device(tri_dev *dev, EncBufs &d, const uint64_t *term_first, const size_t nterms, const uint64_t np, const uint64_t nhits, uint8_t *index_out,
                         const size_t cap, size_t *index_len, tri_term *terms_out) {
        (void)np;
        (void)nhits;
        // ---- host: the block structure (which block belongs to which term)
        std::vector<uint32_t> blk_first(nterms + 1, 0), blk_term;
        for (size_t t = 0; t < nterms; ++t) {
                const uint64_t nb = (term_first[t + 1] - term_first[t] + 31) / 32;
                if ((uint64_t)blk_first[t] + nb > 0xfffffff0ull)
                        return fail(TRI_ERR_UNSUPPORTED, "more than 2^32 blocks");
                blk_first[t + 1] = blk_first[t] + (uint32_t)nb;
                blk_term.insert(blk_term.end(), (size_t)nb, (uint32_t)t);
        }
        const uint32_t nblocks = blk_first[nterms];
        std::vector<uint64_t> term_off(nterms + 1, 0);
        std::vector<uint64_t> blk_off(nblocks + 1, 0);
        if (nblocks) {
                HIP_TRY(hipMalloc((void **)&d.hit_off, (np + 1) * 8));
                HIP_TRY(hipMalloc((void **)&d.term_first, (nterms + 1) * 8));
                HIP_TRY(hipMalloc((void **)&d.blk_first, (nterms + 1) * 4));
                HIP_TRY(hipMalloc((void **)&d.blk_term, (size_t)nblocks * 4));
                HIP_TRY(hipMalloc((void **)&d.sizes, (size_t)nblocks * 4));
                HIP_TRY(hipMalloc((void **)&d.tails, (size_t)nblocks * 4));
                HIP_TRY(hipMalloc((void **)&d.blk_off, ((size_t)nblocks + 1) * 8));
                HIP_TRY(hipMemcpyAsync(d.term_first, term_first, (nterms + 1) * 8, hipMemcpyHostToDevice, dev->stream));
                HIP_TRY(hipMemcpyAsync(d.blk_first, blk_first.data(), (nterms + 1) * 4, hipMemcpyHostToDevice, dev->stream));
                HIP_TRY(hipMemcpyAsync(d.blk_term, blk_term.data(), (size_t)nblocks * 4, hipMemcpyHostToDevice, dev->stream));
                // hits before every posting, then the blocks' sizes and their running sum
                // (exclusive scans over the whole device: chunk sums, chunk bases, chunks — k_encode.hpp)
                int rcs;
                if ((rcs = enc_scan(dev, d, d.freqs, d.hit_off, np)))
                        return rcs;
                const EncArgs a{d.docs, d.freqs, d.pos, d.plens, d.payloads, d.hit_off, d.term_first, d.blk_first, d.blk_term, nblocks};
                hipLaunchKernelGGL(k_enc_size, dim3((nblocks + 255) / 256), dim3(256), 0, dev->stream, a, d.sizes, d.tails);
                if ((rcs = enc_scan(dev, d, d.sizes, d.blk_off, (uint64_t)nblocks)))
                        return rcs;
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(blk_off.data(), d.blk_off, ((size_t)nblocks + 1) * 8, hipMemcpyDeviceToHost, dev->stream));
                HIP_TRY(hipStreamSynchronize(dev->stream));
        }
        // ---- host: where every term's chunk starts (2 bytes + its blocks + its skiplist entries)
        for (size_t t = 0; t < nterms; ++t) {
                const uint32_t g0 = blk_first[t], g1 = blk_first[t + 1];
                uint64_t entries = 0;
                if (g1 > g0) {
                        const uint32_t first_marked = (g0 + 8) / 8 * 8 - 1;
                        if (g1 - 1 >= first_marked)
                                entries = std::min<uint64_t>(65535, (g1 - 1 - first_marked) / 8 + 1);
                }
                const uint64_t size = 2 + (blk_off[g1] - blk_off[g0]) + 8 * entries;
                if (term_off[t] + size > 0xffffffffull)
                        return fail(TRI_ERR_UNSUPPORTED, "the index would exceed 4 GiB (term_index_ctx offsets are 32 bits)");
                terms_out[t] = {(uint32_t)(term_first[t + 1] - term_first[t]), (uint32_t)term_off[t], (uint32_t)size};
                term_off[t + 1] = term_off[t] + size;
        }
        *index_len = (size_t)term_off[nterms];
        if (!index_out)
                return TRI_OK; // (sizing call)
        if (cap < *index_len)
                return fail(TRI_ERR_INVALID, "tri_encode_google: the index needs %zu bytes, %zu given", *index_len, cap);
        if (!*index_len)
                return TRI_OK;
        HIP_TRY(hipMalloc((void **)&d.out, *index_len));
        HIP_TRY(hipMemsetAsync(d.out, 0, *index_len, dev->stream)); // (a term without documents is two zero bytes)
        if (nblocks) {
                HIP_TRY(hipMalloc((void **)&d.term_off, (nterms + 1) * 8));
                HIP_TRY(hipMemcpyAsync(d.term_off, term_off.data(), (nterms + 1) * 8, hipMemcpyHostToDevice, dev->stream));
                const EncArgs a{d.docs, d.freqs, d.pos, d.plens, d.payloads, d.hit_off, d.term_first, d.blk_first, d.blk_term, nblocks};
                hipLaunchKernelGGL(k_enc_write, dim3((nblocks + 255) / 256), dim3(256), 0, dev->stream, a, d.blk_off, d.tails, d.term_off, d.out);
                HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipMemcpyAsync(index_out, d.out, *index_len, hipMemcpyDeviceToHost, dev->stream));
        HIP_TRY(hipStreamSynchronize(dev->stream));
        return TRI_OK;
}

} // namespace

extern "C" int tri_encode_google_payloads(tri_dev *dev, const uint32_t *docs, const uint32_t *freqs, const uint16_t *positions, const uint8_t *payload_lens,
                                          const uint64_t *payloads, size_t npositions, const uint64_t *term_first, size_t nterms, uint8_t *index_out, size_t cap,
                                          size_t *index_len, tri_term *terms_out) {
        if (!dev || !term_first || !index_len || (nterms && !terms_out) || (payload_lens && !payloads))
                return fail(TRI_ERR_INVALID, "tri_encode_google: null argument");
        HIP_TRY(hipSetDevice(dev->device));
        const uint64_t np = nterms ? term_first[nterms] : 0;
        if (np && (!docs || !freqs))
                return fail(TRI_ERR_INVALID, "tri_encode_google: null postings");
        if (npositions && !positions)
                return fail(TRI_ERR_INVALID, "tri_encode_google: null positions");
        // ---- host: input validation (what the reference's encoder would refuse)
        uint64_t nhits = 0;
        for (size_t t = 0; t < nterms; ++t) {
                if (term_first[t + 1] < term_first[t])
                        return fail(TRI_ERR_INVALID, "tri_encode_google: term_first must ascend");
                const uint64_t n = term_first[t + 1] - term_first[t];
                if (n > 0xffffffffull)
                        return fail(TRI_ERR_UNSUPPORTED, "term %zu: more than 2^32 documents", t);
                uint32_t prev = 0;
                for (uint64_t p = term_first[t]; p < term_first[t + 1]; ++p) {
                        if (!docs[p] || docs[p] <= prev)
                                return fail(TRI_ERR_INVALID, "term %zu: document IDs must be > 0 and strictly ascending (codecs.h:188-190)", t);
                        prev = docs[p];
                        // the posting's hits: positions[nhits .. nhits + freqs[p]) — counted hits only (new_hit drops a payload-less hit at
                        // position 0, google_codec.cpp:42-45), non-descending within the document (:49: the encoder writes pos - lastPos)
                        if ((uint64_t)freqs[p] > npositions - std::min<uint64_t>(npositions, nhits))
                                return fail(TRI_ERR_INVALID, "term %zu, document %u: freqs[] asks for more positions than the %zu given", t, docs[p], npositions);
                        uint32_t last_pos = 0;
                        for (uint64_t h = nhits; h < nhits + freqs[p]; ++h) {
                                const uint32_t plen = payload_lens ? payload_lens[h] : 0u;
                                if (plen > 8)
                                        return fail(TRI_ERR_INVALID, "term %zu, document %u: a payload of %u bytes (at most 8: google_codec.cpp:46)", t, docs[p], plen);
                                if ((!positions[h] && !plen) || positions[h] < last_pos) // (a position-0 hit WITH a payload is a counted hit: :42-45)
                                        return fail(TRI_ERR_INVALID, "term %zu, document %u: positions must be non-descending within a document, and > 0 for a hit without payload (google_codec.cpp:42-49)", t, docs[p]);
                                last_pos = positions[h];
                        }
                        nhits += freqs[p];
                }
        }
        EncBufs d;
        if (np) {
                HIP_TRY(hipMalloc((void **)&d.docs, np * 4));
                HIP_TRY(hipMalloc((void **)&d.freqs, np * 4));
                HIP_TRY(hipMalloc((void **)&d.pos, (nhits + 1) * 2));
                HIP_TRY(hipMemcpyAsync(d.docs, docs, np * 4, hipMemcpyHostToDevice, dev->stream));
                HIP_TRY(hipMemcpyAsync(d.freqs, freqs, np * 4, hipMemcpyHostToDevice, dev->stream));
                if (nhits)
                        HIP_TRY(hipMemcpyAsync(d.pos, positions, nhits * 2, hipMemcpyHostToDevice, dev->stream));
                if (nhits && payload_lens) {
                        HIP_TRY(hipMalloc((void **)&d.plens, nhits));
                        HIP_TRY(hipMalloc((void **)&d.payloads, nhits * 8));
                        HIP_TRY(hipMemcpyAsync(d.plens, payload_lens, nhits, hipMemcpyHostToDevice, dev->stream));
                        HIP_TRY(hipMemcpyAsync(d.payloads, payloads, nhits * 8, hipMemcpyHostToDevice, dev->stream));
                }
        }
        return encode_google_device(dev, d, term_first, nterms, np, nhits, index_out, cap, index_len, terms_out);
}

// ---- Codecs::Lucene::Encoder (lucene_codec.cpp:163-388) on the device, PFOR128 payload (k_lencode.hpp, lucene_enc_units.hpp)
// temporaries of the write-side calls: from the device handle's buffer pool, back to it when the call returns (the stream has been synchronised by then)
namespace {
struct PoolTmp {
        tri_dev *dev;
        std::vector<void *> p;
        ~PoolTmp() {
                if (!p.empty())
                        hipStreamSynchronize(dev->stream); // (an early error return: nothing may still be running on what goes back to the pool)
                for (void *q : p)
                        pool_free(dev, q);
        }
        hipError_t get(void **out, size_t bytes) {
                const hipError_t e = pool_alloc(dev, out, bytes ? bytes : 8);
                if (e == hipSuccess)
                        p.push_back(*out);
                return e;
        }
};
} // namespace

// the device side of the Lucene-shaped encoder: d_docs / d_freqs / d_pos hold np postings and nhits hits, term after term as term_first (host) says
static int encode_lucene_device(tri_dev *dev, const uint32_t *d_docs, const uint32_t *d_freqs, const uint16_t *d_pos, const uint64_t np, const uint64_t nhits, const uint64_t *term_first,
                                const size_t nterms, uint8_t *index_out, const size_t index_cap, size_t *index_len, uint8_t *hits_out, const size_t hits_cap, size_t *hits_len,
                                tri_term *terms_out) {
        PoolTmp tmp{dev}; // (the large temporaries come from the device handle's pool: a sizing call and the call that follows it use the same ones)
        EncBufs scratch; // (enc_scan's chunk sums)
        uint32_t *d_hdelta, *d_dcnt, *d_hcnt, *d_dsize, *d_hsize, *d_tail_d, *d_tail_h, *d_isize, *d_hsz;
        uint64_t *d_hit_off, *d_term_first, *d_dblk_first, *d_hblk_first, *d_doff, *d_hoff, *d_term_off, *d_hterm_off;
        HIP_TRY(tmp.get((void **)&d_hdelta, (nhits + 1) * 4));
        HIP_TRY(tmp.get((void **)&d_hit_off, (np + 2) * 8));
        HIP_TRY(tmp.get((void **)&d_term_first, (nterms + 1) * 8));
        HIP_TRY(tmp.get((void **)&d_dcnt, (nterms + 1) * 4));
        HIP_TRY(tmp.get((void **)&d_hcnt, (nterms + 1) * 4));
        HIP_TRY(tmp.get((void **)&d_dblk_first, (nterms + 2) * 8));
        HIP_TRY(tmp.get((void **)&d_hblk_first, (nterms + 2) * 8));
        HIP_TRY(tmp.get((void **)&d_tail_d, (nterms + 1) * 4));
        HIP_TRY(tmp.get((void **)&d_tail_h, (nterms + 1) * 4));
        HIP_TRY(tmp.get((void **)&d_isize, (nterms + 1) * 4));
        HIP_TRY(tmp.get((void **)&d_hsz, (nterms + 1) * 4));
        HIP_TRY(tmp.get((void **)&d_term_off, (nterms + 2) * 8));
        HIP_TRY(tmp.get((void **)&d_hterm_off, (nterms + 2) * 8));
        HIP_TRY(hipMemcpyAsync(d_term_first, term_first, (nterms + 1) * 8, hipMemcpyHostToDevice, dev->stream));
        int rcs;
        if ((rcs = enc_scan(dev, scratch, d_freqs, d_hit_off, np)))
                return rcs;
        const dim3 block(256);
        auto grid = [](uint64_t n) { return dim3((uint32_t)std::max<uint64_t>(1, (n + 255) / 256)); };
        LencArgs a{d_docs, d_freqs, d_pos, d_hit_off, d_term_first, d_hdelta, d_dblk_first, d_hblk_first, (uint64_t)nterms};
        hipLaunchKernelGGL(k_lenc_hdelta, grid(np), block, 0, dev->stream, a, d_hdelta, np);
        hipLaunchKernelGGL(k_lenc_term_counts, grid(nterms), block, 0, dev->stream, (const uint64_t *)d_term_first, (const uint64_t *)d_hit_off, (uint64_t)nterms, d_dcnt, d_hcnt);
        if ((rcs = enc_scan(dev, scratch, d_dcnt, d_dblk_first, nterms)) || (rcs = enc_scan(dev, scratch, d_hcnt, d_hblk_first, nterms)))
                return rcs;
        uint64_t nd = 0, nh = 0;
        HIP_TRY(hipMemcpyAsync(&nd, d_dblk_first + nterms, 8, hipMemcpyDeviceToHost, dev->stream));
        HIP_TRY(hipMemcpyAsync(&nh, d_hblk_first + nterms, 8, hipMemcpyDeviceToHost, dev->stream));
        HIP_TRY(hipStreamSynchronize(dev->stream));
        HIP_TRY(tmp.get((void **)&d_dsize, (nd + 1) * 4));
        HIP_TRY(tmp.get((void **)&d_hsize, (nh + 1) * 4));
        HIP_TRY(tmp.get((void **)&d_doff, (nd + 2) * 8));
        HIP_TRY(tmp.get((void **)&d_hoff, (nh + 2) * 8));
        hipLaunchKernelGGL(k_lenc_dblk_size, grid(nd), block, 0, dev->stream, a, nd, d_dsize);
        hipLaunchKernelGGL(k_lenc_hblk_size, grid(nh), block, 0, dev->stream, a, nh, d_hsize);
        hipLaunchKernelGGL(k_lenc_tail_size, grid(nterms), block, 0, dev->stream, a, d_tail_d, d_tail_h);
        if ((rcs = enc_scan(dev, scratch, d_dsize, d_doff, nd)) || (rcs = enc_scan(dev, scratch, d_hsize, d_hoff, nh)))
                return rcs;
        LencPlace pl{d_doff, d_hoff, d_term_off, d_hterm_off, d_tail_d, d_tail_h};
        hipLaunchKernelGGL(k_lenc_term_sizes, grid(nterms), block, 0, dev->stream, a, pl, d_isize, d_hsz);
        if ((rcs = enc_scan(dev, scratch, d_isize, d_term_off, nterms)) || (rcs = enc_scan(dev, scratch, d_hsz, d_hterm_off, nterms)))
                return rcs;
        HIP_TRY(hipGetLastError());
        std::vector<uint64_t> term_off(nterms + 1), hterm_off(nterms + 1);
        HIP_TRY(hipMemcpyAsync(term_off.data(), d_term_off, (nterms + 1) * 8, hipMemcpyDeviceToHost, dev->stream));
        HIP_TRY(hipMemcpyAsync(hterm_off.data(), d_hterm_off, (nterms + 1) * 8, hipMemcpyDeviceToHost, dev->stream));
        HIP_TRY(hipStreamSynchronize(dev->stream));
        if (term_off[nterms] > 0xffffffffull || hterm_off[nterms] > 0xffffffffull)
                return fail(TRI_ERR_UNSUPPORTED, "the index or hits.data would exceed 4 GiB (term_index_ctx offsets and the term header's hits offset are 32 bits)");
        for (size_t t = 0; t < nterms; ++t)
                terms_out[t] = {(uint32_t)(term_first[t + 1] - term_first[t]), (uint32_t)term_off[t], (uint32_t)(term_off[t + 1] - term_off[t])};
        *index_len = (size_t)term_off[nterms];
        *hits_len = (size_t)hterm_off[nterms];
        if (!index_out)
                return TRI_OK; // (sizing call)
        if (index_cap < *index_len || hits_cap < *hits_len || (*hits_len && !hits_out))
                return fail(TRI_ERR_INVALID, "tri_encode_lucene: the index needs %zu bytes (%zu given), hits.data %zu (%zu given)", *index_len, index_cap, *hits_len, hits_cap);
        uint8_t *d_index, *d_hits;
        HIP_TRY(tmp.get((void **)&d_index, *index_len + 8));
        HIP_TRY(tmp.get((void **)&d_hits, *hits_len + 8));
        hipLaunchKernelGGL(k_lenc_dblk_write, grid(nd), block, 0, dev->stream, a, pl, nd, d_index);
        hipLaunchKernelGGL(k_lenc_hblk_write, grid(nh), block, 0, dev->stream, a, pl, nh, d_hits);
        hipLaunchKernelGGL(k_lenc_term_write, grid(nterms), block, 0, dev->stream, a, pl, d_index, d_hits);
        HIP_TRY(hipGetLastError());
        if (*index_len)
                HIP_TRY(hipMemcpyAsync(index_out, d_index, *index_len, hipMemcpyDeviceToHost, dev->stream));
        if (*hits_len)
                HIP_TRY(hipMemcpyAsync(hits_out, d_hits, *hits_len, hipMemcpyDeviceToHost, dev->stream));
        HIP_TRY(hipStreamSynchronize(dev->stream));
        return TRI_OK;
}

extern "C" int tri_encode_lucene(tri_dev *dev, const uint32_t *docs, const uint32_t *freqs, const uint16_t *positions, size_t npositions, const uint64_t *term_first, size_t nterms,
                                 uint8_t *index_out, size_t index_cap, size_t *index_len, uint8_t *hits_out, size_t hits_cap, size_t *hits_len, tri_term *terms_out) {
        if (!dev || !term_first || !index_len || !hits_len || (nterms && !terms_out))
                return fail(TRI_ERR_INVALID, "tri_encode_lucene: null argument");
        HIP_TRY(hipSetDevice(dev->device));
        const uint64_t np = nterms ? term_first[nterms] : 0;
        if (np && (!docs || !freqs))
                return fail(TRI_ERR_INVALID, "tri_encode_lucene: null postings");
        if (npositions && !positions)
                return fail(TRI_ERR_INVALID, "tri_encode_lucene: null positions");
        // ---- host: what the encoder would refuse (lucene_encoder.hpp: documents > 0 and ascending within a term; a hit at position 0 is not a hit — refused
        //      here, as by tri_encode_google, rather than dropped silently; positions non-descending within a document)
        uint64_t nhits = 0;
        for (size_t t = 0; t < nterms; ++t) {
                if (term_first[t + 1] < term_first[t])
                        return fail(TRI_ERR_INVALID, "tri_encode_lucene: term_first must ascend");
                uint32_t prev = 0;
                for (uint64_t p = term_first[t]; p < term_first[t + 1]; ++p) {
                        if (!docs[p] || docs[p] <= prev)
                                return fail(TRI_ERR_INVALID, "term %zu: document IDs must be > 0 and strictly ascending (codecs.h:188-190)", t);
                        prev = docs[p];
                        if ((uint64_t)freqs[p] > npositions - std::min<uint64_t>(npositions, nhits))
                                return fail(TRI_ERR_INVALID, "term %zu, document %u: freqs[] asks for more positions than the %zu given", t, docs[p], npositions);
                        uint32_t last_pos = 0;
                        for (uint64_t h = nhits; h < nhits + freqs[p]; ++h) {
                                if (!positions[h] || positions[h] < last_pos)
                                        return fail(TRI_ERR_INVALID, "term %zu, document %u: positions must be > 0 and non-descending within a document", t, docs[p]);
                                last_pos = positions[h];
                        }
                        nhits += freqs[p];
                }
        }
        struct Up {
                uint32_t *docs = nullptr, *freqs = nullptr;
                uint16_t *pos = nullptr;
                ~Up() {
                        hipFree(docs), hipFree(freqs), hipFree(pos);
                }
        } u;
        HIP_TRY(hipMalloc((void **)&u.docs, (np + 1) * 4));
        HIP_TRY(hipMalloc((void **)&u.freqs, (np + 1) * 4));
        HIP_TRY(hipMalloc((void **)&u.pos, (nhits + 1) * 2));
        if (np) {
                HIP_TRY(hipMemcpyAsync(u.docs, docs, np * 4, hipMemcpyHostToDevice, dev->stream));
                HIP_TRY(hipMemcpyAsync(u.freqs, freqs, np * 4, hipMemcpyHostToDevice, dev->stream));
        }
        if (nhits)
                HIP_TRY(hipMemcpyAsync(u.pos, positions, nhits * 2, hipMemcpyHostToDevice, dev->stream));
        return encode_lucene_device(dev, u.docs, u.freqs, u.pos, np, nhits, term_first, nterms, index_out, index_cap, index_len, hits_out, hits_cap, hits_len, terms_out);
}

// ---- SegmentIndexSession::commit (indexer.cpp:311-478) on the device: sort, gather, encode (k_commit.hpp, commit_sort.hip, k_encode.hpp)
extern "C" int tri_sort_pairs_u64_u32(const unsigned long long *keys_in, unsigned long long *keys_out, const unsigned *vals_in, unsigned *vals_out, size_t n, void *tmp,
                                      size_t *tmp_bytes, hipStream_t stream); // (commit_sort.hip)

// (codec: TRI_CODEC_GOOGLE — index_out only —, or TRI_CODEC_LUCENE — index_out + hits_out, payload-less hits)
static int commit_device(tri_dev *dev, const int codec, const uint32_t *term_ids, const uint32_t *doc_ids, const uint32_t *freqs, const uint16_t *positions, const uint8_t *payload_lens,
                         const uint64_t *payloads, size_t npostings, size_t npositions, uint8_t *index_out, size_t cap, size_t *index_len, uint8_t *hits_out, size_t hits_cap,
                         size_t *hits_len, uint32_t *term_ids_out, tri_term *terms_out, size_t terms_cap, size_t *nterms, tri_commit_stats *stats) {
        if (!dev || !index_len || !nterms || (npostings && (!term_ids || !doc_ids || !freqs)) || (payload_lens && !payloads) || (npositions && !positions))
                return fail(TRI_ERR_INVALID, "tri_commit_google: null argument");
        if (npostings > 0xfffffff0ull)
                return fail(TRI_ERR_UNSUPPORTED, "tri_commit_google: more than 2^32 postings in one session: commit in parts");
        HIP_TRY(hipSetDevice(dev->device));
        const uint64_t np = npostings;
        uint64_t nhits = 0, docs_cnt = 0;
        for (uint64_t i = 0; i < np; ++i) { // (the session's own bookkeeping: hits in all, documents = runs of one documentID in insertion order)
                nhits += freqs[i];
                docs_cnt += i == 0 || doc_ids[i] != doc_ids[i - 1];
        }
        if (nhits > npositions)
                return fail(TRI_ERR_INVALID, "tri_commit_google: freqs[] asks for %llu positions, %zu given", (unsigned long long)nhits, npositions);
        *nterms = 0;
        *index_len = 0;
        if (stats)
                *stats = tri_commit_stats{docs_cnt, np, nhits, 0};
        if (!np)
                return TRI_OK;
        PoolTmp tmp{dev}; // (the large temporaries come from the device handle's pool: a sizing call and the call that follows it use the same ones)
        uint32_t *d_terms, *d_docs_in, *d_freqs_in, *d_vals, *d_perm, *d_marks, *d_term_ids;
        unsigned long long *d_keys, *d_keys_sorted, *d_err;
        uint16_t *d_pos_in = nullptr;
        uint8_t *d_plens_in = nullptr;
        uint64_t *d_payloads_in = nullptr, *d_hit_off_in, *d_hit_off_out, *d_mark_rank, *d_term_first;
        HIP_TRY(tmp.get((void **)&d_terms, np * 4));
        HIP_TRY(tmp.get((void **)&d_docs_in, np * 4));
        HIP_TRY(tmp.get((void **)&d_freqs_in, np * 4));
        HIP_TRY(tmp.get((void **)&d_keys, np * 8));
        HIP_TRY(tmp.get((void **)&d_keys_sorted, np * 8));
        HIP_TRY(tmp.get((void **)&d_vals, np * 4));
        HIP_TRY(tmp.get((void **)&d_perm, np * 4));
        HIP_TRY(tmp.get((void **)&d_marks, np * 4));
        HIP_TRY(tmp.get((void **)&d_hit_off_in, (np + 1) * 8));
        HIP_TRY(tmp.get((void **)&d_hit_off_out, (np + 1) * 8));
        HIP_TRY(tmp.get((void **)&d_mark_rank, (np + 1) * 8));
        HIP_TRY(tmp.get((void **)&d_err, 8));
        HIP_TRY(hipMemcpyAsync(d_terms, term_ids, np * 4, hipMemcpyHostToDevice, dev->stream));
        HIP_TRY(hipMemcpyAsync(d_docs_in, doc_ids, np * 4, hipMemcpyHostToDevice, dev->stream));
        HIP_TRY(hipMemcpyAsync(d_freqs_in, freqs, np * 4, hipMemcpyHostToDevice, dev->stream));
        if (nhits) {
                HIP_TRY(tmp.get((void **)&d_pos_in, nhits * 2));
                HIP_TRY(hipMemcpyAsync(d_pos_in, positions, nhits * 2, hipMemcpyHostToDevice, dev->stream));
                if (payload_lens) {
                        HIP_TRY(tmp.get((void **)&d_plens_in, nhits));
                        HIP_TRY(tmp.get((void **)&d_payloads_in, nhits * 8));
                        HIP_TRY(hipMemcpyAsync(d_plens_in, payload_lens, nhits, hipMemcpyHostToDevice, dev->stream));
                        HIP_TRY(hipMemcpyAsync(d_payloads_in, payloads, nhits * 8, hipMemcpyHostToDevice, dev->stream));
                }
        }
        EncBufs d; // (the sorted postings: what the encoder reads)
        HIP_TRY(hipMalloc((void **)&d.docs, np * 4));
        HIP_TRY(hipMalloc((void **)&d.freqs, np * 4));
        HIP_TRY(hipMalloc((void **)&d.pos, (nhits + 1) * 2));
        if (nhits && payload_lens) {
                HIP_TRY(hipMalloc((void **)&d.plens, nhits));
                HIP_TRY(hipMalloc((void **)&d.payloads, nhits * 8));
        }
        const dim3 grid((uint32_t)((np + 255) / 256)), block(256);
        // ---- keys in the order the reference's commit walks (bucket = termID & 31, then termID, then documentID), sorted with the postings' indices
        hipLaunchKernelGGL(k_commit_keys, grid, block, 0, dev->stream, (const uint32_t *)d_terms, (const uint32_t *)d_docs_in, d_keys, d_vals, np);
        size_t sort_bytes = 0;
        HIP_TRY((hipError_t)tri_sort_pairs_u64_u32(d_keys, d_keys_sorted, d_vals, d_perm, np, nullptr, &sort_bytes, dev->stream));
        void *d_sort_tmp = nullptr;
        HIP_TRY(tmp.get(&d_sort_tmp, sort_bytes));
        HIP_TRY((hipError_t)tri_sort_pairs_u64_u32(d_keys, d_keys_sorted, d_vals, d_perm, np, d_sort_tmp, &sort_bytes, dev->stream));
        // ---- documents and frequencies in sorted order; the hits follow their postings
        hipLaunchKernelGGL(k_commit_gather, grid, block, 0, dev->stream, (const unsigned long long *)d_keys_sorted, (const uint32_t *)d_perm, (const uint32_t *)d_freqs_in, d.docs,
                           d.freqs, d_marks, np);
        int rcs;
        EncBufs scan_scratch; // (enc_scan's chunk sums)
        if ((rcs = enc_scan(dev, scan_scratch, d_freqs_in, d_hit_off_in, np)) || (rcs = enc_scan(dev, scan_scratch, d.freqs, d_hit_off_out, np)) ||
            (rcs = enc_scan(dev, scan_scratch, d_marks, d_mark_rank, np)))
                return rcs;
        hipLaunchKernelGGL(k_commit_hits, grid, block, 0, dev->stream, (const uint32_t *)d_perm, (const uint64_t *)d_hit_off_in, (const uint64_t *)d_hit_off_out,
                           (const uint32_t *)d.freqs, (const uint16_t *)d_pos_in, d.pos, (const uint8_t *)d_plens_in, d.plens, (const uint64_t *)d_payloads_in, d.payloads, np);
        HIP_TRY(hipMemsetAsync(d_err, 0xff, 8, dev->stream));
        hipLaunchKernelGGL(k_commit_validate, grid, block, 0, dev->stream, (const unsigned long long *)d_keys_sorted, (const uint32_t *)d.freqs, (const uint64_t *)d_hit_off_out,
                           (const uint16_t *)d.pos, (const uint8_t *)d.plens, np, d_err);
        HIP_TRY(hipGetLastError());
        unsigned long long err = 0;
        uint64_t nt = 0;
        HIP_TRY(hipMemcpyAsync(&err, d_err, 8, hipMemcpyDeviceToHost, dev->stream));
        HIP_TRY(hipMemcpyAsync(&nt, d_mark_rank + np, 8, hipMemcpyDeviceToHost, dev->stream));
        HIP_TRY(hipStreamSynchronize(dev->stream));
        if (err != ~0ull) {
                static const char *const why[] = {"", "document 0", "the same (term, document) twice (indexer.cpp:446: documentID > prevDID)",
                                                  "positions must be non-descending within a document, and > 0 for a hit without payload (google_codec.cpp:42-49)",
                                                  "a payload of more than 8 bytes (google_codec.cpp:46)"};
                return fail(TRI_ERR_INVALID, "tri_commit_google: sorted posting %llu: %s", (unsigned long long)(err >> 8) - 1, why[std::min<unsigned long long>(err & 0xff, 4)]);
        }
        *nterms = (size_t)nt;
        if (stats)
                stats->total_terms = nt;
        // ---- the distinct terms: first postings and termIDs, commit order
        HIP_TRY(tmp.get((void **)&d_term_first, (nt + 1) * 8));
        HIP_TRY(tmp.get((void **)&d_term_ids, nt * 4));
        hipLaunchKernelGGL(k_commit_terms, grid, block, 0, dev->stream, (const unsigned long long *)d_keys_sorted, (const uint32_t *)d_marks, (const uint64_t *)d_mark_rank, d_term_first,
                           d_term_ids, np);
        HIP_TRY(hipGetLastError());
        std::vector<uint64_t> term_first(nt + 1);
        std::vector<uint32_t> tids(nt);
        HIP_TRY(hipMemcpyAsync(term_first.data(), d_term_first, nt * 8, hipMemcpyDeviceToHost, dev->stream));
        HIP_TRY(hipMemcpyAsync(tids.data(), d_term_ids, nt * 4, hipMemcpyDeviceToHost, dev->stream));
        HIP_TRY(hipStreamSynchronize(dev->stream));
        term_first[nt] = np;
        std::vector<tri_term> tt(nt);
        uint8_t *const io = index_out && terms_cap >= nt ? index_out : nullptr;
        if (codec == TRI_CODEC_LUCENE) {
                if (int rc = encode_lucene_device(dev, d.docs, d.freqs, d.pos, np, nhits, term_first.data(), nt, io, cap, index_len, hits_out, hits_cap, hits_len, tt.data()))
                        return rc;
        } else if (int rc = encode_google_device(dev, d, term_first.data(), nt, np, nhits, io, cap, index_len, tt.data()))
                return rc;
        if (!index_out)
                return TRI_OK; // (sizing call: *index_len and *nterms)
        if (terms_cap < nt || !terms_out || !term_ids_out)
                return fail(TRI_ERR_INVALID, "tri_commit_google: the session holds %llu distinct terms, room for %zu given", (unsigned long long)nt, terms_cap);
        memcpy(terms_out, tt.data(), nt * sizeof(tri_term));
        memcpy(term_ids_out, tids.data(), nt * 4);
        return TRI_OK;
}

extern "C" int tri_commit_google(tri_dev *dev, const uint32_t *term_ids, const uint32_t *doc_ids, const uint32_t *freqs, const uint16_t *positions, const uint8_t *payload_lens,
                                 const uint64_t *payloads, size_t npostings, size_t npositions, uint8_t *index_out, size_t cap, size_t *index_len, uint32_t *term_ids_out,
                                 tri_term *terms_out, size_t terms_cap, size_t *nterms, tri_commit_stats *stats) {
        return commit_device(dev, TRI_CODEC_GOOGLE, term_ids, doc_ids, freqs, positions, payload_lens, payloads, npostings, npositions, index_out, cap, index_len, nullptr, 0, nullptr,
                             term_ids_out, terms_out, terms_cap, nterms, stats);
}
extern "C" int tri_commit_lucene(tri_dev *dev, const uint32_t *term_ids, const uint32_t *doc_ids, const uint32_t *freqs, const uint16_t *positions, size_t npostings, size_t npositions,
                                 uint8_t *index_out, size_t cap, size_t *index_len, uint8_t *hits_out, size_t hits_cap, size_t *hits_len, uint32_t *term_ids_out, tri_term *terms_out,
                                 size_t terms_cap, size_t *nterms, tri_commit_stats *stats) {
        if (!hits_len)
                return fail(TRI_ERR_INVALID, "tri_commit_lucene: null argument");
        *hits_len = 0;
        return commit_device(dev, TRI_CODEC_LUCENE, term_ids, doc_ids, freqs, positions, nullptr, nullptr, npostings, npositions, index_out, cap, index_len, hits_out, hits_cap, hits_len,
                             term_ids_out, terms_out, terms_cap, nterms, stats);
}

// ---- Codecs::Google::IndexSession::merge (google_codec.cpp:186-438) for a whole dictionary, on the device (k_commit.hpp)
// The codecs' merge for a whole dictionary (Codecs::Google::IndexSession::merge, google_codec.cpp:186-438; Codecs::Lucene::IndexSession::merge, lucene_codec.cpp:963-1396 — the
// same k-way walk over the participants' postings, most recent first, the winner kept unless its participant masks it; the codecs differ in how postings and hits are stored,
// i.e. in the decode and the encode at the two ends of the sort below)
static int merge_device(tri_dev *dev, const int codec, tri_index *const *parts, size_t nparts, const uint32_t *part_terms, size_t nterms, uint8_t *index_out, size_t cap, size_t *index_len,
                        uint8_t *hits_out, size_t hits_cap, size_t *hits_len, tri_term *terms_out, tri_commit_stats *stats) {
        if (!dev || !parts || !nparts || (nterms && (!part_terms || !terms_out)) || !index_len)
                return fail(TRI_ERR_INVALID, "tri_merge_%s: null argument", codec == TRI_CODEC_GOOGLE ? "google" : "lucene");
        if (nparts > 65535)
                return fail(TRI_ERR_INVALID, "tri_merge_%s: at most 65535 participants (google_codec.cpp:186 / lucene_codec.cpp:963: uint16_t participantsCnt)", codec == TRI_CODEC_GOOGLE ? "google" : "lucene");
        HIP_TRY(hipSetDevice(dev->device));
        for (size_t p = 0; p < nparts; ++p) {
                if (!parts[p] || parts[p]->dev != dev || parts[p]->codec != codec)
                        return fail(TRI_ERR_INVALID, "tri_merge_%s: participant %zu is not a %s index of this device", codec == TRI_CODEC_GOOGLE ? "google" : "lucene", p, codec == TRI_CODEC_GOOGLE ? "google_codec" : "lucene_codec");
                if (codec == TRI_CODEC_LUCENE && !parts[p]->d_hits && parts[p]->info.postings)
                        return fail(TRI_ERR_INVALID, "tri_merge_lucene: participant %zu was uploaded without its hits.data (the merged segment needs every hit)", p);
        }
        // ---- the jobs: every (participant, output term) that holds postings, participant-major — the most recent participant's postings first, so that
        //      a stable sort leaves them first among equal (term, document) keys
        std::vector<std::vector<MergeJob>> jobs(nparts);
        std::vector<uint64_t> part_first(nparts + 1, 0);
        uint64_t np = 0;
        for (size_t p = 0; p < nparts; ++p) {
                part_first[p] = np;
                for (size_t t = 0; t < nterms; ++t) {
                        const uint32_t idx = part_terms[t * nparts + p];
                        if (idx == 0xffffffffu)
                                continue;
                        if (idx >= parts[p]->terms.size())
                                return fail(TRI_ERR_INVALID, "tri_merge_google: output term %zu: term %u out of range in participant %zu", t, idx, p);
                        const DevTerm &dt = parts[p]->terms[idx];
                        if (!dt.documents)
                                continue; // (merge.cpp:263-270: a participant without documents for the term takes no part)
                        if (!(dt.flags & TERM_FULL_BLOCKS))
                                return fail(TRI_ERR_UNSUPPORTED, "tri_merge_google: term %u of participant %zu has short blocks inside its list (not written by the reference's encoder)", idx, p);
                        jobs[p].push_back({idx, (uint32_t)t, np});
                        np += dt.documents;
                }
        }
        part_first[nparts] = np;
        if (np > 0xfffffff0ull)
                return fail(TRI_ERR_UNSUPPORTED, "tri_merge_google: more than 2^32 postings: merge in parts");
        *index_len = 0;
        if (stats)
                *stats = tri_commit_stats{0, 0, 0, 0};
        PoolTmp tmp{dev}; // (the large temporaries come from the device handle's pool: a sizing call and the call that follows it use the same ones)
        EncBufs d;       // the merged postings: what the encoder reads
        EncBufs scratch; // (enc_scan's chunk sums)
        std::vector<uint64_t> term_first(nterms + 1, 0);
        uint64_t kept = 0, nh_out = 0;
        if (np) {
                unsigned long long *d_keys, *d_keys_sorted;
                uint32_t *d_vals, *d_perm, *d_freqs_all, *d_keep, *d_src_of, *d_term_cnt;
                uint64_t *d_hit_off_all, *d_rank, *d_part_first, *d_hit_off_out, *d_term_first;
                const uint32_t **d_masked;
                HIP_TRY(tmp.get((void **)&d_keys, np * 8));
                HIP_TRY(tmp.get((void **)&d_keys_sorted, np * 8));
                HIP_TRY(tmp.get((void **)&d_vals, np * 4));
                HIP_TRY(tmp.get((void **)&d_perm, np * 4));
                HIP_TRY(tmp.get((void **)&d_freqs_all, np * 4));
                HIP_TRY(tmp.get((void **)&d_keep, np * 4));
                HIP_TRY(tmp.get((void **)&d_hit_off_all, (np + 1) * 8));
                HIP_TRY(tmp.get((void **)&d_rank, (np + 1) * 8));
                HIP_TRY(tmp.get((void **)&d_part_first, (nparts + 1) * 8));
                HIP_TRY(tmp.get((void **)&d_masked, nparts * sizeof(void *)));
                HIP_TRY(tmp.get((void **)&d_term_cnt, (nterms + 1) * 4));
                HIP_TRY(tmp.get((void **)&d_term_first, (nterms + 2) * 8));
                std::vector<const uint32_t *> masked(nparts);
                for (size_t p = 0; p < nparts; ++p)
                        masked[p] = parts[p]->d_masked;
                HIP_TRY(hipMemcpyAsync(d_part_first, part_first.data(), (nparts + 1) * 8, hipMemcpyHostToDevice, dev->stream));
                HIP_TRY(hipMemcpyAsync(d_masked, masked.data(), nparts * sizeof(void *), hipMemcpyHostToDevice, dev->stream));
                std::vector<MergeJob *> d_jobs(nparts, nullptr);
                for (size_t p = 0; p < nparts; ++p) {
                        if (jobs[p].empty())
                                continue;
                        HIP_TRY(tmp.get((void **)&d_jobs[p], jobs[p].size() * sizeof(MergeJob)));
                        HIP_TRY(hipMemcpyAsync(d_jobs[p], jobs[p].data(), jobs[p].size() * sizeof(MergeJob), hipMemcpyHostToDevice, dev->stream));
                        const tri_index *ix = parts[p];
                        TRI_LAUNCH(k_merge_decode, codec, dim3((uint32_t)std::min<size_t>(jobs[p].size(), (size_t)dev->cus * 16)), dim3(256), dev->stream, ix->d_index,
                                   ix->d_blk_last, ix->d_blk_off, ix->d_terms, (const MergeJob *)d_jobs[p], (uint32_t)jobs[p].size(), d_freqs_all, d_keys, d_vals);
                }
                HIP_TRY(hipGetLastError());
                int rcs;
                if ((rcs = enc_scan(dev, scratch, d_freqs_all, d_hit_off_all, np)))
                        return rcs;
                uint64_t nh_all = 0;
                HIP_TRY(hipMemcpyAsync(&nh_all, d_hit_off_all + np, 8, hipMemcpyDeviceToHost, dev->stream));
                HIP_TRY(hipStreamSynchronize(dev->stream));
                uint16_t *d_pos_all;
                uint8_t *d_plens_all;
                uint64_t *d_payloads_all;
                HIP_TRY(tmp.get((void **)&d_pos_all, (nh_all + 1) * 2));
                HIP_TRY(tmp.get((void **)&d_plens_all, nh_all + 1));
                HIP_TRY(tmp.get((void **)&d_payloads_all, (nh_all + 1) * 8));
                for (size_t p = 0; p < nparts; ++p) {
                        if (jobs[p].empty())
                                continue;
                        const tri_index *ix = parts[p];
                        if (codec == TRI_CODEC_LUCENE)
                                hipLaunchKernelGGL(k_merge_hits_lucene, dim3((uint32_t)std::min<size_t>(jobs[p].size(), (size_t)dev->cus * 16)), dim3(256), 0, dev->stream, ix->d_hits,
                                                   ix->d_blk_hits, ix->d_hdir, ix->d_terms, (const MergeJob *)d_jobs[p], (uint32_t)jobs[p].size(), (const uint32_t *)d_freqs_all,
                                                   (const uint64_t *)d_hit_off_all, d_pos_all, d_plens_all, d_payloads_all);
                        else
                                hipLaunchKernelGGL(k_merge_hits, dim3((uint32_t)std::min<size_t>(jobs[p].size(), (size_t)dev->cus * 16)), dim3(256), 0, dev->stream, ix->d_index, ix->d_blk_off,
                                                   ix->d_blk_hits, ix->d_terms, (const MergeJob *)d_jobs[p], (uint32_t)jobs[p].size(), (const uint32_t *)d_freqs_all, (const uint64_t *)d_hit_off_all,
                                                   d_pos_all, d_plens_all, d_payloads_all);
                }
                HIP_TRY(hipGetLastError());
                // ---- sort by (output term, document); the first of equal keys is the most recent participant's
                size_t sort_bytes = 0;
                HIP_TRY((hipError_t)tri_sort_pairs_u64_u32(d_keys, d_keys_sorted, d_vals, d_perm, np, nullptr, &sort_bytes, dev->stream));
                void *d_sort_tmp = nullptr;
                HIP_TRY(tmp.get(&d_sort_tmp, sort_bytes));
                HIP_TRY((hipError_t)tri_sort_pairs_u64_u32(d_keys, d_keys_sorted, d_vals, d_perm, np, d_sort_tmp, &sort_bytes, dev->stream));
                const dim3 grid((uint32_t)((np + 255) / 256)), block(256);
                hipLaunchKernelGGL(k_merge_select, grid, block, 0, dev->stream, (const unsigned long long *)d_keys_sorted, (const uint32_t *)d_perm, (const uint64_t *)d_part_first,
                                   (uint32_t)nparts, (const uint32_t *const *)d_masked, d_keep, np);
                if ((rcs = enc_scan(dev, scratch, d_keep, d_rank, np)))
                        return rcs;
                HIP_TRY(hipMemcpyAsync(&kept, d_rank + np, 8, hipMemcpyDeviceToHost, dev->stream));
                HIP_TRY(hipStreamSynchronize(dev->stream));
                HIP_TRY(hipMalloc((void **)&d.docs, (kept + 1) * 4));
                HIP_TRY(hipMalloc((void **)&d.freqs, (kept + 1) * 4));
                HIP_TRY(tmp.get((void **)&d_src_of, (kept + 1) * 4));
                HIP_TRY(tmp.get((void **)&d_hit_off_out, (kept + 2) * 8));
                HIP_TRY(hipMemsetAsync(d_term_cnt, 0, (nterms + 1) * 4, dev->stream));
                hipLaunchKernelGGL(k_merge_compact, grid, block, 0, dev->stream, (const unsigned long long *)d_keys_sorted, (const uint32_t *)d_perm, (const uint32_t *)d_keep,
                                   (const uint64_t *)d_rank, (const uint32_t *)d_freqs_all, d.docs, d.freqs, d_src_of, d_term_cnt, np);
                if ((rcs = enc_scan(dev, scratch, d.freqs, d_hit_off_out, kept)) || (rcs = enc_scan(dev, scratch, d_term_cnt, d_term_first, nterms)))
                        return rcs;
                HIP_TRY(hipMemcpyAsync(&nh_out, d_hit_off_out + kept, 8, hipMemcpyDeviceToHost, dev->stream));
                HIP_TRY(hipMemcpyAsync(term_first.data(), d_term_first, (nterms + 1) * 8, hipMemcpyDeviceToHost, dev->stream));
                HIP_TRY(hipStreamSynchronize(dev->stream));
                HIP_TRY(hipMalloc((void **)&d.pos, (nh_out + 1) * 2));
                HIP_TRY(hipMalloc((void **)&d.plens, nh_out + 1));
                HIP_TRY(hipMalloc((void **)&d.payloads, (nh_out + 1) * 8));
                if (kept)
                        hipLaunchKernelGGL(k_commit_hits, dim3((uint32_t)((kept + 255) / 256)), block, 0, dev->stream, (const uint32_t *)d_src_of, (const uint64_t *)d_hit_off_all,
                                           (const uint64_t *)d_hit_off_out, (const uint32_t *)d.freqs, (const uint16_t *)d_pos_all, d.pos, (const uint8_t *)d_plens_all, d.plens,
                                           (const uint64_t *)d_payloads_all, d.payloads, kept);
                HIP_TRY(hipGetLastError());
        }
        if (codec == TRI_CODEC_LUCENE) {
                if (int rc = encode_lucene_device(dev, d.docs, d.freqs, d.pos, kept, nh_out, term_first.data(), nterms, index_out, cap, index_len, hits_out, hits_cap, hits_len, terms_out))
                        return rc;
        } else if (int rc = encode_google_device(dev, d, term_first.data(), nterms, kept, nh_out, index_out, cap, index_len, terms_out))
                return rc;
        if (stats) {
                stats->sum_terms_docs = kept;
                stats->sum_term_hits = nh_out;
                for (size_t t = 0; t < nterms; ++t)
                        stats->total_terms += term_first[t + 1] > term_first[t]; // (merge.cpp:241: a term that keeps no document is dropped from the dictionary)
        }
        return TRI_OK;
}
extern "C" int tri_merge_google(tri_dev *dev, tri_index *const *parts, size_t nparts, const uint32_t *part_terms, size_t nterms, uint8_t *index_out, size_t cap, size_t *index_len,
                                tri_term *terms_out, tri_commit_stats *stats) {
        return merge_device(dev, TRI_CODEC_GOOGLE, parts, nparts, part_terms, nterms, index_out, cap, index_len, nullptr, 0, nullptr, terms_out, stats);
}
extern "C" int tri_merge_lucene(tri_dev *dev, tri_index *const *parts, size_t nparts, const uint32_t *part_terms, size_t nterms, uint8_t *index_out, size_t cap, size_t *index_len,
                                uint8_t *hits_out, size_t hits_cap, size_t *hits_len, tri_term *terms_out, tri_commit_stats *stats) {
        if (!hits_len)
                return fail(TRI_ERR_INVALID, "tri_merge_lucene: null argument");
        *hits_len = 0;
        return merge_device(dev, TRI_CODEC_LUCENE, parts, nparts, part_terms, nterms, index_out, cap, index_len, hits_out, hits_cap, hits_len, terms_out, stats);
}

#ifdef TRI_PROF
// perf-probe builds: read back and reset the per-phase cycle totals (dev_stream.hpp)
extern "C" int tri_debug_prof(uint64_t *out32) {
        unsigned long long h[32];
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prof), sizeof h));
        for (int i = 0; i < 32; ++i)
                out32[i] = h[i];
        memset(h, 0, sizeof h);
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_prof), h, sizeof h));
        return TRI_OK;
}
#endif
