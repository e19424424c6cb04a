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

// device-visible structures and kernels
#include "dev_structs.hpp"

// planner / launch options of a device handle (tri_dev_set_option); the defaults are what bench.py measures
struct tri_options {
        uint64_t dense_min_postings = 512 * 1024; // TASK_DENSE needs at least this many postings over the query's lists (0: every multi-term query)
        uint64_t dense_task_cost = 192 * 1024;    // postings per bitmap-window task
        uint64_t fused = 1;                       // AccumulatedScore top-K of dense queries in one pass (k_fused); 0: k_and_dense + k_score
        uint64_t fused_task_cost = 0;             // postings per one-pass task; 0: sized from the batch (256 K .. 8 M, about two tasks per resident workgroup)
        uint64_t fused_freq_cap = 0;              // 0: the field width decides; else a smaller saturation point (exercises the rescoring path)
        uint64_t account_needed_bytes = 0;        // 1: tri_batch_create also works out tri_batch_info.cand_needed_bytes (a directory walk per candidate-tile query)
        uint64_t fused_halfwords = 1;             // 16-bit window words for queries of <= 5 distinct terms (windows twice as long); 0: always 32-bit
        uint64_t overlap_dense_wgs = 0, overlap_cand_wgs = 0; // both non-zero: the two matching kernels side by side on two streams
        uint64_t planes = 7;     // term planes (k_planes.hpp), a bit set: 1 k_and probes them, 2 k_and_dense ORs them in, 4 top-K CNF queries run in k_planes; 0: off
        uint64_t planes_split = 0; // a k_planes query is cut into this many docID ranges (tasks) that share its threshold; 0: 2 or 3 by the batch's size; >= 65536: by postings like the other one-pass tasks.  cfg3's unions: 0 10.9 ms, 2 8.0, 3 8.5, 4 9.2 (a task has fixed costs)
        uint64_t plane_div = 128; // a term gets a plane when it holds at least docs_cnt / plane_div documents (and the batch's uses repay one decode of its list);
                                 // measured, step ms at 32 / 64 / 128 / 256: cfg3 16.9 / 15.9 / 15.3 / 15.4, cfg2 - / 2.90 / 2.71 / 2.76 (the planes' build grows with it)
};

struct tri_dev {
        int device;
        hipStream_t stream, stream2; // stream2: the candidate-tile kernel when the two matching kernels run side by side
        hipEvent_t ev_fork, ev_join;
        int cus;
        tri_options opt;
        // The large buffers of a batch (output regions, score streams, term planes, decoded lists) are recycled from batch to batch: a
        // caller that compiles a batch per step would otherwise hipMalloc and hipFree gigabytes per step — hipFree synchronises the device
        // (the next batch cannot be compiled while the current one runs), and a cold 15 GB hipMalloc was measured anywhere between 10 ms
        // and 1 s (bench.py's end_to_end.batch_create_cold_ms).  One tri_dev per host thread: no lock.
        struct Pool {
                std::vector<std::pair<size_t, void *>> idle;    // (bytes, buffer) not in use
                std::unordered_map<void *, size_t> size_of;     // every pooled buffer, in use or idle
                size_t idle_bytes = 0;
        } pool;
};
constexpr size_t POOL_MIN_BYTES = 1u << 20;   // smaller buffers are not worth pooling
constexpr size_t POOL_IDLE_CAP = 64ull << 30; // idle buffers beyond this are given back to the device (largest first)

// a buffer of at least `bytes`: an idle one of the pool that is not more than twice as large, else a fresh allocation
static hipError_t pool_alloc(tri_dev *dev, void **out, const size_t bytes) {
        if (bytes < POOL_MIN_BYTES)
                return hipMalloc(out, bytes);
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
        const size_t rounded = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
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

struct tri_index {
        tri_dev *dev = nullptr;
        int codec = TRI_CODEC_GOOGLE;
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
        uint32_t max_doc = 0;
        uint32_t nwin = 0; // cells per win[] row
        DevTerm *d_terms = nullptr;
        std::vector<DevTerm> terms;
        std::vector<uint32_t> h_blk_last; // host copy of the directory's last-docID column (planner: task output offsets)
        std::vector<tri_term> tctx;
        std::vector<uint64_t> docbytes, hitbytes;
        tri_index_info info{};
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
        }
};

struct tri_batch {
        tri_index *ix = nullptr;
        uint32_t flags, topk;
        int similarity = TRI_SIM_BM25;
        size_t nq;
        std::vector<DevQuery> plan; // execution order (cost descending)
        std::vector<uint32_t> qterms;
        std::vector<uint32_t> slot_of_query; // caller query -> plan slot (UINT32_MAX: trivially empty)
        std::vector<int32_t> qstatus;        // per caller query: TRI_OK, or why the planner left it out of the batch (it then reports no matches)
        DevQuery *d_plan = nullptr;
        std::vector<DevTask> tasks; // scheduling order (cost descending)
        DevTask *d_tasks = nullptr;
        uint32_t *d_sched = nullptr; // task indices, heaviest first: [0, n_dense) TASK_DENSE, then the TASK_CAND ones, then the TASK_FUSED ones
        uint32_t n_dense = 0, n_cand = 0, n_fused = 0, n_fused16 = 0, n_fusedgen = 0; // (n_fused: 32-bit window words; n_fused16: 16-bit; n_fusedgen: general trees)
        uint32_t n_planes = 0, n_planes8 = 0; // TASK_PLANES / TASK_PLANES8 tasks (k_planes), scheduled after the general trees
        // term planes (k_planes.hpp): the head terms the batch's queries share, decoded once per launch into d_planes
        std::vector<uint32_t> plane_terms; // row -> term
        uint32_t *d_plane_terms = nullptr, *d_planes = nullptr, *d_qplane = nullptr; // d_qplane: parallel to d_qterms, the term's row or PL_NONE
        uint32_t plw = 0;                  // words of one plane
        unsigned long long *d_qthr = nullptr; // k_planes: per query, the best k-th score any of its tasks has seen (cleared at every run)
        uint32_t *d_sparse = nullptr;      // k_planes: per resident workgroup, the lists of a task's decoded (non-plane) slots
        uint32_t sparse_cap = 0;           // ... entries per workgroup
        uint64_t term_bytes_planes = 0, plane_decoded_bytes = 0;
        hipEvent_t ev_pl = nullptr, ev_k = nullptr; // after k_term_planes; after k_planes
        std::vector<DevFused> fused; // slot maps of the TASK_FUSED queries (DevQuery::fused_idx)
        DevFused *d_fused = nullptr;
        uint64_t term_bytes_fused = 0;
        // HIP events on the engine stream: start, after k_and_dense, after k_and, after k_fused, end (owned by the batch: two batches
        // in flight on one device keep their own timings)
        hipEvent_t ev0 = nullptr, ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_p = nullptr, ev1 = nullptr;
        bool ran = false;
        uint32_t *d_qterms = nullptr;
        uint32_t *d_out = nullptr;
        uint32_t *d_counts = nullptr; // per task, indexed first_task + i in query order
        uint32_t *d_ticket = nullptr;
        uint64_t term_bytes_phrase_hits = 0; // hit bytes of the phrase terms (part of term_bytes): k_phrase's share
        uint64_t cand_needed_term_bytes = 0; // option account_needed_bytes: see tri_batch_info.cand_needed_bytes
        uint32_t *d_rich_allow = nullptr; // default mode, batches that hold general trees: per match the reportable terms the tree sits on
        bool rich_allow = false;
        uint64_t *d_hashes = nullptr;
        uint64_t *d_qcounts = nullptr; // per caller query: matches of the last run (device copy for the result gather)
        // AccumulatedScoreScheme
        std::vector<uint32_t> sterms;
        std::vector<double> sweights;
        uint32_t *d_sterms = nullptr;
        double *d_sweights = nullptr;
        uint32_t *d_part_docs = nullptr, *d_part_counts = nullptr, *d_top_docs = nullptr, *d_top_counts = nullptr;
        double *d_part_scores = nullptr;
        float *d_top_scores = nullptr;
        double *d_all_scores = nullptr; // topk == 0: one double per out[] slot
        // TRI_FLAG_MATCHED_TERMS (k_rich.hpp): sterms[] holds every query's reportable terms; R = the widest query's count
        uint32_t rich_R = 0;
        uint32_t *d_rich_present = nullptr, *d_task_hits = nullptr;
        uint16_t *d_rich_freq = nullptr, *d_rich_pool = nullptr;
        uint8_t *d_rich_plen = nullptr;     // TRI_FLAG_HIT_PAYLOADS: per hit of the pool, term_hit::payloadLen ...
        uint64_t *d_rich_payload = nullptr; // ... and term_hit::payload
        uint64_t *d_task_pos_base = nullptr;
        std::vector<uint64_t> h_task_pos_base; // per task; [ntasks] = the pool's size
        size_t rich_pool_cap = 0;
        // phrases
        std::vector<DevPhrase> phrases;
        std::vector<uint32_t> pterms, ptasks;
        DevPhrase *d_phrases = nullptr;
        uint32_t *d_pterms = nullptr, *d_ptasks = nullptr;
        double *d_pscore = nullptr; // per out[] slot: sum of the phrase scores of the match (scored mode)
        uint64_t out_capacity = 0;
        uint64_t term_bytes = 0; // sum of docbytes over all query terms
        uint64_t term_bytes_dense = 0; // … of the queries that run as TASK_DENSE
        std::vector<uint32_t> h_counts;       // per task
        std::vector<uint64_t> h_query_counts; // per plan slot
        bool synced = false;
        tri_batch_info info{};
        ~tri_batch() { // also runs when tri_batch_create fails half-way: nothing allocated so far is leaked
                if (ix) {
                        hipSetDevice(ix->dev->device);
                        if (ran && !synced) // (its large buffers go back to the device's pool: nothing of this batch may still be running on them)
                                hipStreamSynchronize(ix->dev->stream);
                }
                for (hipEvent_t e : {ev0, ev_a, ev_b, ev_c, ev_p, ev1, ev_pl, ev_k})
                        if (e)
                                hipEventDestroy(e);
                pool_free(ix ? ix->dev : nullptr, d_sparse);
                hipFree(d_qthr);
                hipFree(d_plane_terms);
                pool_free(ix ? ix->dev : nullptr, d_planes);
                hipFree(d_qplane);
                hipFree(d_fused);
                hipFree(d_plan);
                hipFree(d_tasks);
                hipFree(d_sched);
                hipFree(d_qterms);
                pool_free(ix ? ix->dev : nullptr, d_out);
                hipFree(d_counts);
                hipFree(d_ticket);
                pool_free(ix ? ix->dev : nullptr, d_rich_allow);
                hipFree(d_hashes);
                hipFree(d_qcounts);
                hipFree(d_sterms);
                hipFree(d_sweights);
                pool_free(ix ? ix->dev : nullptr, d_part_docs);
                pool_free(ix ? ix->dev : nullptr, d_part_scores);
                hipFree(d_part_counts);
                hipFree(d_top_docs);
                hipFree(d_top_scores);
                hipFree(d_top_counts);
                pool_free(ix ? ix->dev : nullptr, d_all_scores);
                pool_free(ix ? ix->dev : nullptr, d_rich_present);
                pool_free(ix ? ix->dev : nullptr, d_rich_freq);
                hipFree(d_task_hits);
                hipFree(d_task_pos_base);
                hipFree(d_rich_pool);
                hipFree(d_rich_plen);
                hipFree(d_rich_payload);
                hipFree(d_phrases);
                hipFree(d_pterms);
                hipFree(d_ptasks);
                pool_free(ix ? ix->dev : nullptr, d_pscore);
        }
};

#include "dev_stream.hpp"
#include "k_decode.hpp"
#include "k_match.hpp"
#include "k_score.hpp"
#include "k_fused.hpp"
#include "k_planes.hpp"
#include "k_encode.hpp"
#include "k_phrase.hpp"
#include "k_rich.hpp"

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
        hipSetDevice(d->device);
        hipEventDestroy(d->ev_fork);
        hipEventDestroy(d->ev_join);
        hipStreamDestroy(d->stream2);
        hipStreamDestroy(d->stream);
        for (auto &b : d->pool.idle) // (buffers still in use belong to batches the caller has not destroyed: theirs to release)
                hipFree(b.second);
        delete d;
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
                             {"planes", &tri_options::planes},
                             {"plane_div", &tri_options::plane_div},
                             {"planes_split", &tri_options::planes_split}};
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

extern "C" int tri_dev_sync(tri_dev *d) {
        if (!d)
                return fail(TRI_ERR_INVALID, "null dev");
        HIP_TRY(hipStreamSynchronize(d->stream));
        return TRI_OK;
}

extern "C" void *tri_dev_stream(tri_dev *d) { return d ? (void *)d->stream : nullptr; }

// ------------------------------------------------------------------------------------------ host: upload
namespace {
        // host-side prefix varint (Switch/switch_compiler_aux.h:53-80) — used only by the upload-time walk
        inline size_t h_vb_get(const uint8_t *ip, uint32_t &v) {
                const uint32_t x = ip[0];
                if (!(x & 0x80u)) {
                        v = x;
                        return 1;
                } else if (!(x & 0x40u)) {
                        v = ((x & 0x3fu) << 8) | ip[1];
                        return 2;
                } else if (!(x & 0x20u)) {
                        v = ((x & 0x1fu) << 16) | ip[1] | ((uint32_t)ip[2] << 8);
                        return 3;
                } else if (!(x & 0x10u)) {
                        v = ((x & 0x0fu) << 24) | ((uint32_t)ip[1] << 16) | ((uint32_t)ip[2] << 8) | ip[3];
                        return 4;
                }
                v = ip[1] | ((uint32_t)ip[2] << 8) | ((uint32_t)ip[3] << 16) | ((uint32_t)ip[4] << 24);
                return 5;
        }
        inline size_t h_vb_len(uint8_t b0) { return b0 < 0x80 ? 1 : b0 < 0xc0 ? 2 : b0 < 0xe0 ? 3 : b0 < 0xf0 ? 4 : 5; }

        // ints() group of 128 values (lucene_codec.cpp:69-100 framing; PFOR128 payload, include/pfor128.md).  Returns the
        // bytes consumed, 0 when malformed.  Upload-time only: it yields the per-32-document directory rows.
        inline size_t h_ints_decode(const uint8_t *p, const uint8_t *end, uint32_t *v) {
                if (p >= end)
                        return 0;
                const uint32_t L = p[0];
                if (!L) {
                        if (p + 1 >= end || p + 1 + h_vb_len(p[1]) > end)
                                return 0;
                        uint32_t x;
                        const size_t n = h_vb_get(p + 1, x);
                        for (int i = 0; i < 128; ++i)
                                v[i] = x;
                        return 1 + n;
                }
                if (p + 1 + 4 * (size_t)L > end)
                        return 0;
                std::vector<uint32_t> w(L + 2, 0);
                memcpy(w.data(), p + 1, (size_t)L * 4);
                const uint32_t b = w[0] & 0xff, nexc = (w[0] >> 8) & 0xff, eb = (w[0] >> 16) & 0xff;
                if (b > 32 || eb > 32 || 1 + 4 * b + (nexc + 3) / 4 + (nexc * eb + 31) / 32 != L)
                        return 0;
                const uint32_t *packed = w.data() + 1, *epos = packed + 4 * b, *ehigh = epos + (nexc + 3) / 4;
                for (uint32_t i = 0; i < 128; ++i) {
                        uint32_t x = 0;
                        if (b) {
                                const uint32_t bit = i * b;
                                uint64_t win = packed[bit >> 5];
                                if ((bit & 31) + b > 32)
                                        win |= (uint64_t)packed[(bit >> 5) + 1] << 32;
                                x = (uint32_t)((win >> (bit & 31)) & (b == 32 ? 0xffffffffull : ((1ull << b) - 1)));
                        }
                        v[i] = x;
                }
                for (uint32_t e = 0; e < nexc; ++e) {
                        const uint32_t pos = (epos[e >> 2] >> ((e & 3) * 8)) & 0xff;
                        const uint32_t bit = e * eb;
                        uint64_t win = ehigh[bit >> 5];
                        if ((bit & 31) + eb > 32)
                                win |= (uint64_t)ehigh[(bit >> 5) + 1] << 32;
                        if (pos >= 128 || b >= 32)
                                return 0;
                        v[pos] |= (uint32_t)((win >> (bit & 31)) & (eb == 32 ? 0xffffffffull : ((1ull << eb) - 1))) << b;
                }
                return 1 + (size_t)L * 4;
        }
        inline size_t h_ints_skip(const uint8_t *p, const uint8_t *end) {
                if (p >= end)
                        return 0;
                if (!p[0] && p + 1 >= end)
                        return 0;
                const size_t n = p[0] ? 1 + 4 * (size_t)p[0] : 1 + h_vb_len(p[1]);
                return p + n <= end ? n : 0;
        }
        // Per quarter (32 values) of a VALIDATED ints() group: where its exceptions start in the group's list and how many it has,
        // packed e0 | cnt << 8.  (The positions are ascending, so a quarter's exceptions are one run of the list.)
        inline bool h_ints_exc(const uint8_t *p, uint32_t out[4]) {
                out[0] = out[1] = out[2] = out[3] = 0;
                const uint32_t L = p[0];
                if (!L)
                        return true;
                uint32_t w0;
                memcpy(&w0, p + 1, 4);
                const uint32_t b = w0 & 0xff, nexc = (w0 >> 8) & 0xff;
                const uint8_t *epos = p + 5 + 16 * (size_t)b;
                uint32_t cnt[4] = {0, 0, 0, 0}, e0[4] = {0, 0, 0, 0};
                for (uint32_t e = 0; e < nexc; ++e) {
                        const uint32_t q = epos[e] >> 5;
                        if (q > 3 || (e && epos[e] <= epos[e - 1]))
                                return false;
                        if (!cnt[q])
                                e0[q] = e;
                        ++cnt[q];
                }
                for (int q = 0; q < 4; ++q)
                        out[q] = e0[q] | cnt[q] << 8;
                return true;
        }

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
        if (codec != TRI_CODEC_GOOGLE && codec != TRI_CODEC_LUCENE)
                return fail(TRI_ERR_INVALID, "tri_index_upload: unknown codec %d", codec);
        if (len > 0xffffffffull)
                return fail(TRI_ERR_FORMAT, "index exceeds 32-bit chunk offsets (codecs.h:26)");
        if (codec == TRI_CODEC_GOOGLE && len >= 0x80000000ull) // bit 31 of a block's hits offset carries BLK_HITS_PLAIN (k_phrase / k_rich mask it off)
                return fail(TRI_ERR_UNSUPPORTED, "a google_codec index of 2 GiB or more (%zu bytes): split the segment", len);
        HIP_TRY(hipSetDevice(dev->device));
        auto ix = std::make_unique<tri_index>();
        ix->dev = dev;
        ix->codec = codec;
        ix->terms.resize(nterms);
        ix->tctx.assign(terms, terms + nterms);
        ix->docbytes.assign(nterms, 0);
        ix->hitbytes.assign(nterms, 0);
        std::vector<uint32_t> blk_last, blk_off;
        std::vector<uint32_t> blk_hits, hdir; // LUCENE + hits.data only
        std::vector<uint4> blk_rec;           // LUCENE only
        std::vector<uint8_t> dstream;         // GOOGLE only
        std::vector<uint32_t> blk_doff;
        if (codec == TRI_CODEC_GOOGLE) {
                dstream.reserve(len / 3 + 64);
                blk_doff.reserve(len / 96 + nterms);
        }
        const bool want_hits = codec == TRI_CODEC_LUCENE && hits_len;
        blk_last.reserve(len / 96 + nterms);
        blk_off.reserve(len / 96 + nterms);
        uint64_t postings = 0, docb = 0, hitb = 0;
        // One pass over every chunk: hop block headers (google_codec.cpp:641-697), validate, record the directory
        // and the algorithmic byte split of SURVEY §8(d).
        for (size_t ti = 0; ti < nterms; ++ti) {
                const tri_term &t = terms[ti];
                DevTerm &dt = ix->terms[ti];
                dt.documents = t.documents;
                dt.first_block = (uint32_t)blk_last.size();
                dt.nblocks = 0;
                dt.last_n = 0;
                dt.flags = 0;
                dt.npfor = 0;
                dt.pad = 0;
                if (!t.size || !t.documents) {
                        dt.documents = 0;
                        continue;
                }
                if (codec == TRI_CODEC_LUCENE) {
                        // Lucene-shaped chunk (lucene_codec.cpp:163-388): 14-byte header, full 128-document blocks as two ints()
                        // groups, varbyte (delta, freq) tail, 22-byte skiplist entries.  One directory row per 32 documents.
                        if ((uint64_t)t.offset + t.size > len || t.size < 14)
                                return fail(TRI_ERR_FORMAT, "term %zu: chunk [%u,+%u) outside index (%zu)", ti, t.offset, t.size, len);
                        const uint8_t *base = index + t.offset, *p = base + 14;
                        uint32_t posChunk, hitsOff, sumHits;
                        uint16_t sk;
                        memcpy(&hitsOff, base, 4);
                        memcpy(&sumHits, base + 4, 4);
                        memcpy(&posChunk, base + 8, 4);
                        memcpy(&sk, base + 12, 2);
                        if (14 + (size_t)sk * 22 > t.size)
                                return fail(TRI_ERR_FORMAT, "term %zu: skiplist larger than chunk", ti);
                        const uint8_t *end = base + t.size - (size_t)sk * 22;
                        uint32_t left = t.documents, doc = 0;
                        uint32_t vals[128], fvals[128];
                        uint64_t hits_seen = 0;
                        while (left >= 128) {
                                if (p >= end)
                                        return fail(TRI_ERR_FORMAT, "term %zu: truncated block", ti);
                                const uint32_t goff = (uint32_t)(p - index);
                                const size_t used = h_ints_decode(p, end, vals);
                                if (!used) // (the group's header word does not describe a PFOR128 payload of the declared length)
                                        return fail(TRI_ERR_FORMAT, "term %zu: an ints() group that is not PFOR128 (include/pfor128.md) — a lucene_codec segment written by the reference's "
                                                                    "own build carries lemire/FastPFor<4> payloads (lucene_codec.cpp:57-64), which this engine does not read: re-encode "
                                                                    "the segment with csrc/host/lucene_encoder.hpp, or use google_codec", ti);
                                p += used;
                                uint32_t xd[4], xf[4];
                                const size_t usedf = h_ints_decode(p, end, fvals);
                                if (!usedf || !h_ints_exc(index + goff, xd) || !h_ints_exc(p, xf))
                                        return fail(TRI_ERR_FORMAT, "term %zu: a freqs group / exception list that is not PFOR128 (include/pfor128.md; FastPFor<4> payloads of the reference's own "
                                                                    "build are not readable)", ti);
                                p += usedf;
                                uint32_t hdr[2];
                                for (int gi = 0; gi < 2; ++gi) { // the two groups' header words as the row records cache them
                                        const uint8_t *gp = gi ? p - usedf : index + goff;
                                        if (gp[0])
                                                memcpy(&hdr[gi], gp + 1, 4);
                                        else {
                                                const uint32_t v = gi ? fvals[0] : vals[0];
                                                if (v >> 31)
                                                        return fail(TRI_ERR_UNSUPPORTED, "term %zu: an all-equal group of value %u", ti, v);
                                                hdr[gi] = 0x80000000u | v;
                                        }
                                }
                                for (uint32_t q4 = 0; q4 < 4; ++q4) {
                                        blk_rec.push_back(make_uint4(goff, xd[q4] | xf[q4] << 16, hdr[0], hdr[1]));
                                        if (want_hits)
                                                blk_hits.push_back((uint32_t)hits_seen);
                                        for (uint32_t i = 0; i < 32; ++i) {
                                                if (!vals[q4 * 32 + i])
                                                        return fail(TRI_ERR_FORMAT, "term %zu: zero document delta", ti);
                                                doc += vals[q4 * 32 + i];
                                                if (want_hits)
                                                        hits_seen += fvals[q4 * 32 + i];
                                        }
                                        blk_last.push_back(doc);
                                        blk_off.push_back(goff);
                                        dt.nblocks++;
                                }
                                left -= 128;
                        }
                        dt.npfor = dt.nblocks;
                        dt.last_n = 32;
                        while (left) {
                                const uint32_t n = std::min(left, 32u);
                                blk_off.push_back((uint32_t)(p - index));
                                blk_rec.push_back(make_uint4((uint32_t)(p - index), 0, 0, 0));
                                if (want_hits)
                                        blk_hits.push_back((uint32_t)hits_seen);
                                for (uint32_t i = 0; i < n; ++i) {
                                        uint32_t d, f;
                                        if (p >= end || p + h_vb_len(*p) >= end || p + h_vb_len(*p) + h_vb_len(p[h_vb_len(*p)]) > end)
                                                return fail(TRI_ERR_FORMAT, "term %zu: truncated tail", ti);
                                        p += h_vb_get(p, d);
                                        p += h_vb_get(p, f);
                                        if (!d)
                                                return fail(TRI_ERR_FORMAT, "term %zu: zero document delta", ti);
                                        doc += d;
                                        hits_seen += f;
                                }
                                blk_last.push_back(doc);
                                dt.nblocks++;
                                dt.last_n = n;
                                left -= n;
                        }
                        if (p != end)
                                return fail(TRI_ERR_FORMAT, "term %zu: %zd stray bytes before the skiplist", ti, (ssize_t)(end - p));
                        dt.flags = TERM_FULL_BLOCKS;
                        if (want_hits) {
                                // hits.data of this term (lucene_codec.cpp:245-307, 339-352): sumHits / 128 full blocks
                                // { ints(posDeltas) ints(payloadLens) varbyte(payloadBytes) payload }, then the varbyte tail
                                if (hits_seen != sumHits)
                                        return fail(TRI_ERR_FORMAT, "term %zu: %llu hits by frequency, %u declared", ti, (unsigned long long)hits_seen, sumHits);
                                if ((uint64_t)hitsOff + posChunk > hits_len)
                                        return fail(TRI_ERR_FORMAT, "term %zu: positions chunk [%u,+%u) outside hits.data (%zu)", ti, hitsOff, posChunk, hits_len);
                                const uint8_t *hp = hits + hitsOff, *hend = hp + posChunk;
                                const uint32_t nfull = sumHits / 128;
                                dt.pad = (uint32_t)hdir.size();
                                hdir.push_back(nfull);
                                for (uint32_t hb = 0; hb < nfull; ++hb) {
                                        hdir.push_back((uint32_t)(hp - hits));
                                        for (int g = 0; g < 2; ++g) {
                                                const size_t used = h_ints_skip(hp, hend);
                                                if (!used || hp + used > hend)
                                                        return fail(TRI_ERR_FORMAT, "term %zu: bad hits block %u", ti, hb);
                                                hp += used;
                                        }
                                        uint32_t payloadBytes;
                                        hp += h_vb_get(hp, payloadBytes);
                                        if (hp + payloadBytes > hend)
                                                return fail(TRI_ERR_FORMAT, "term %zu: hits block %u payload overruns the chunk", ti, hb);
                                        hp += payloadBytes;
                                }
                                hdir.push_back((uint32_t)(hp - hits));
                        }
                        const uint64_t db = (uint64_t)(end - base); // 14-byte header + block bytes, no skiplist, no hits.data
                        ix->docbytes[ti] = db;
                        ix->hitbytes[ti] = posChunk;
                        postings += t.documents;
                        docb += db;
                        hitb += posChunk;
                        continue;
                }
                if ((uint64_t)t.offset + t.size > len || t.size < 2)
                        return fail(TRI_ERR_FORMAT, "term %zu: chunk [%u,+%u) outside index (%zu)", ti, t.offset, t.size, len);
                const uint8_t *base = index + t.offset, *p = base + 2, *end = base + t.size;
                uint16_t sk;
                memcpy(&sk, base, 2);
                if ((size_t)sk * 8 + 2 > t.size)
                        return fail(TRI_ERR_FORMAT, "term %zu: skiplist larger than chunk", ti);
                end -= (size_t)sk * 8;
                uint64_t db = 2, hb = 0;
                uint32_t lastDoc = 0, docs = 0;
                bool full_blocks = true;
                while (p != end) {
                        if (p + 3 > end)
                                return fail(TRI_ERR_FORMAT, "term %zu: truncated block header", ti);
                        const uint8_t *h = p;
                        uint32_t delta, blockLength;
                        // (every varint is bounded before it is read: a malformed or truncated chunk must end in TRI_ERR_FORMAT, not in a read past the buffer)
                        if (p + h_vb_len(*p) >= end)
                                return fail(TRI_ERR_FORMAT, "term %zu: truncated block header", ti);
                        p += h_vb_get(p, delta);
                        if (p + h_vb_len(*p) >= end)
                                return fail(TRI_ERR_FORMAT, "term %zu: truncated block header", ti);
                        p += h_vb_get(p, blockLength);
                        const uint32_t n = *p++;
                        if (n < 1 || n > 32 || !delta || (uint64_t)(end - p) < blockLength)
                                return fail(TRI_ERR_FORMAT, "term %zu: bad block header (n=%u, delta=%u, len=%u)", ti, n, delta, blockLength);
                        lastDoc += delta;
                        const uint8_t *s = p, *const bend = p + blockLength;
                        for (uint32_t i = 0; i + 1 < n; ++i) {
                                if (s >= bend)
                                        return fail(TRI_ERR_FORMAT, "term %zu: deltas overrun the block", ti);
                                s += h_vb_len(*s);
                        }
                        if (s > bend)
                                return fail(TRI_ERR_FORMAT, "term %zu: deltas overrun the block", ti);
                        if (dstream.size() + 256 > 0xffffffffull)
                                return fail(TRI_ERR_UNSUPPORTED, "delta stream exceeds 4 GiB");
                        dstream.push_back((uint8_t)n);
                        blk_doff.push_back((uint32_t)dstream.size());
                        dstream.insert(dstream.end(), p, s);
                        uint64_t nhits = 0;
                        for (uint32_t i = 0; i < n; ++i) {
                                if (s >= bend || s + h_vb_len(*s) > bend)
                                        return fail(TRI_ERR_FORMAT, "term %zu: deltas+freqs overrun the block", ti);
                                uint32_t f;
                                s += h_vb_get(s, f);
                                nhits += f;
                        }
                        // GOOGLE: byte offset of the block's first hit (k_phrase / k_rich start there).  Bit 31 (BLK_HITS_PLAIN): every hit of the
                        // block is ONE byte — a position delta < 64 without the new-payload-length flag (google_codec.cpp:38-74) —, so a document's
                        // hits start at the block's first hit + the frequencies before it and no hit has to be parsed to find them
                        uint32_t hits_at = (uint32_t)(s - index);
                        if ((uint64_t)(bend - s) == nhits && !(hits_at >> 31)) {
                                bool plain = true;
                                for (const uint8_t *q = s; q < bend && plain; ++q)
                                        plain = !(*q & 0x81u);
                                if (plain)
                                        hits_at |= BLK_HITS_PLAIN;
                        }
                        blk_hits.push_back(hits_at);
                        db += (uint64_t)(s - h);
                        hb += blockLength - (uint64_t)(s - p);
                        blk_last.push_back(lastDoc);
                        blk_off.push_back((uint32_t)(p - index));
                        if (dt.nblocks && dt.last_n != 32)
                                full_blocks = false; // a short block that is not the last one
                        dt.nblocks++;
                        dt.last_n = n;
                        docs += n;
                        p += blockLength;
                }
                if (docs != t.documents)
                        return fail(TRI_ERR_FORMAT, "term %zu: %u documents in blocks, %u declared", ti, docs, t.documents);
                if (!full_blocks) // the reference encoder only ever leaves the LAST block short (google_codec.cpp:76-88); the kernels' tile and
                                  // output layouts (32 slots per non-final block) rely on it, so a foreign chunk that does not is refused here
                        return fail(TRI_ERR_UNSUPPORTED, "term %zu: a block other than the last holds fewer than 32 documents", ti);
                dt.flags = TERM_FULL_BLOCKS;
                if ((uint64_t)docs * 28 < lastDoc)
                        dt.flags |= TERM_SPARSE;
                ix->docbytes[ti] = db;
                ix->hitbytes[ti] = hb;
                postings += docs;
                docb += db;
                hitb += hb;
        }
        // docID-cell index of the longer lists: win[row + c] = first block whose last docID >= c * CELL_DOCS.  With 288 GB of HBM
        // a 4-byte entry per 1024 docIDs per indexed term is cheap (66 MB at the 10M-document config) and turns directory
        // searches into one load pair: TASK_DENSE reads the entries of its window's ends (every SPAN_BITS / CELL_DOCS-th), a
        // galloping candidate brackets its block to the handful of blocks that end inside its cell.
        const uint32_t max_doc = blk_last.empty() ? 0 : *std::max_element(blk_last.begin(), blk_last.end());
        ix->nwin = (max_doc / SPAN_BITS + 2) * (SPAN_BITS / CELL_DOCS) + 1;
        ix->max_doc = max_doc;
        std::vector<uint32_t> win;
        for (size_t ti = 0; ti < nterms; ++ti) {
                DevTerm &dt = ix->terms[ti];
                dt.win_off = 0xffffffffu;
                if (dt.nblocks < WIN_MIN_BLOCKS)
                        continue;
                dt.win_off = (uint32_t)win.size();
                const uint32_t *bl = &blk_last[dt.first_block];
                uint32_t b = 0;
                if ((uint64_t)win.size() + ix->nwin > 0xfffffff0ull)
                        return fail(TRI_ERR_UNSUPPORTED, "cell index exceeds 2^32 entries");
                for (uint32_t w = 0; w < ix->nwin; ++w) {
                        const uint64_t key = (uint64_t)w * CELL_DOCS;
                        while (b < dt.nblocks && bl[b] < key)
                                ++b;
                        win.push_back(b);
                }
        }
        // device copies
        int rcw;
        if ((rcw = dev_upload(&ix->d_win, win, 3 * CELLS_PER_SPAN + 8))) // (lanes of terms without a row read entries 0, CELLS_PER_SPAN, 2 * CELLS_PER_SPAN and drop them)
                return rcw;
        HIP_TRY(hipMemset(ix->d_win + win.size(), 0, (3 * CELLS_PER_SPAN + 8) * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void **)&ix->d_index, len + 256)); // over-read slack: the byte streams keep several qwords in flight past the cursor
        HIP_TRY(hipMemset(ix->d_index, 0, len + 256));
        if (len)
                HIP_TRY(hipMemcpy(ix->d_index, index, len, hipMemcpyHostToDevice));
        int rc;
        if ((rc = dev_upload(&ix->d_blk_last, blk_last)) || (rc = dev_upload(&ix->d_blk_off, blk_off)) || (rc = dev_upload(&ix->d_terms, ix->terms)))
                return rc;
        if (codec == TRI_CODEC_GOOGLE) {
                blk_doff.push_back((uint32_t)dstream.size() + 1); // (sentinel: block b's delta bytes = blk_doff[b + 1] - blk_doff[b] - 1 for the last block too)
                dstream.resize(dstream.size() + 256, 0); // over-read slack, like index[]
                if ((rc = dev_upload(&ix->d_dstream, dstream)) || (rc = dev_upload(&ix->d_blk_doff, blk_doff)))
                        return rc;
        }
        if (hits_len) { // LUCENE: hits.data (positions) resident next to the index
                HIP_TRY(hipMalloc((void **)&ix->d_hits, hits_len + 256));
                HIP_TRY(hipMemset(ix->d_hits, 0, hits_len + 256));
                HIP_TRY(hipMemcpy(ix->d_hits, hits, hits_len, hipMemcpyHostToDevice));
                if (want_hits && ((rc = dev_upload(&ix->d_blk_hits, blk_hits)) || (rc = dev_upload(&ix->d_hdir, hdir))))
                        return rc;
        }
        if (codec == TRI_CODEC_GOOGLE && (rc = dev_upload(&ix->d_blk_hits, blk_hits)))
                return rc;
        if (codec == TRI_CODEC_LUCENE && (rc = dev_upload(&ix->d_blk_rec, blk_rec)))
                return rc;
        ix->h_blk_last = std::move(blk_last);
        ix->info.index_bytes = len;
        ix->info.directory_bytes = ix->h_blk_last.size() * 8 + nterms * sizeof(DevTerm) + win.size() * 4;
        ix->info.blocks = ix->h_blk_last.size();
        ix->info.postings = postings;
        ix->info.doc_bytes = docb;
        ix->info.hit_bytes = hitb;
        ix->info.nterms = (uint32_t)nterms;
        ix->info.docs_cnt = docs_cnt;
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

// ------------------------------------------------------------------------------------------ host: planner
namespace {
        struct PNode {
                uint32_t op, term;
                uint32_t tok = 0; // index of the program token this node came from (caller-supplied ScorerWeights are per token)
                std::vector<int> kids;
                uint64_t cost = 0;
                bool empty = false;
        };

        // Parse one postfix program into a tree with the reference's flattening (exec.cpp:339-358, 382-393),
        // emptiness propagation and cost model (exec.cpp:35-110).  Returns root index or -1.
        int parse_program(const tri_index *ix, const uint32_t *prog, uint32_t len, std::vector<PNode> &nodes) {
                std::vector<int> st;
                for (uint32_t i = 0; i < len; ++i) {
                        const uint32_t op = prog[i] >> 28, arg = prog[i] & 0x0fffffffu;
                        PNode n;
                        n.op = op;
                        n.tok = i;
                        if (op == TRI_OP_TERM) {
                                n.term = arg;
                                n.cost = arg < ix->terms.size() ? ix->terms[arg].documents : 0;
                                n.empty = n.cost == 0; // unknown term == no documents (index_source.h:60-72)
                        } else {
                                const uint32_t nk = op == TRI_OP_SOME ? (arg & 0xffffu) : arg; // operands taken off the stack
                                if (nk < 1 || nk > st.size())
                                        return -1;
                                std::vector<int> kids(st.end() - nk, st.end());
                                st.resize(st.size() - nk);
                                if (op == TRI_OP_SOME) {
                                        // matchsome (exec.cpp:276-283): operands that can never match are dropped; fewer live operands than
                                        // the threshold: never matches.  cost: docset_iterators.cpp:733-742, the (cnt - min + 1) cheapest
                                        const uint32_t mn = arg >> 16;
                                        if (!mn || mn > nk)
                                                return -1;
                                        for (int k : kids)
                                                if (!nodes[k].empty)
                                                        n.kids.push_back(k);
                                        n.term = mn; // (the threshold rides in the otherwise unused field)
                                        n.empty = n.kids.size() < mn;
                                        std::vector<uint64_t> cs;
                                        for (int k : n.kids)
                                                cs.push_back(nodes[k].cost);
                                        std::sort(cs.begin(), cs.end());
                                        for (size_t i = 0; i + mn <= cs.size(); ++i)
                                                n.cost += cs[i];
                                } else if (op == TRI_OP_PHRASE) {
                                        if (arg > 16) // trinity_limits.h:12 MaxPhraseSize
                                                return -1;
                                        for (int k : kids) {
                                                if (nodes[k].op != TRI_OP_TERM)
                                                        return -1;
                                                n.empty |= nodes[k].empty;
                                        }
                                        n.kids = kids;
                                        n.cost = nodes[kids[0]].cost + UINT32_MAX + (uint64_t)UINT16_MAX * arg;
                                } else if (op == TRI_OP_AND) {
                                        for (int k : kids) {
                                                n.empty |= nodes[k].empty;
                                                if (nodes[k].op == TRI_OP_AND)
                                                        n.kids.insert(n.kids.end(), nodes[k].kids.begin(), nodes[k].kids.end());
                                                else
                                                        n.kids.push_back(k);
                                        }
                                        std::stable_sort(n.kids.begin(), n.kids.end(), [&](int a, int b) { return nodes[a].cost < nodes[b].cost; });
                                        n.cost = nodes[n.kids[0]].cost;
                                } else if (op == TRI_OP_OR) {
                                        for (int k : kids) {
                                                if (nodes[k].empty)
                                                        continue;
                                                if (nodes[k].op == TRI_OP_OR)
                                                        n.kids.insert(n.kids.end(), nodes[k].kids.begin(), nodes[k].kids.end());
                                                else
                                                        n.kids.push_back(k);
                                        }
                                        n.empty = n.kids.empty();
                                        for (int k : n.kids)
                                                n.cost += nodes[k].cost;
                                } else if (op == TRI_OP_OPT) {
                                        if (arg != 2)
                                                return -1;
                                        if (nodes[kids[1]].empty) { // an optional side that can never match adds nothing
                                                st.push_back(kids[0]);
                                                continue;
                                        }
                                        n.kids = kids; // {main, optional}
                                        n.empty = nodes[kids[0]].empty;
                                        n.cost = nodes[kids[0]].cost;
                                } else if (op == TRI_OP_NOT) {
                                        if (arg != 2)
                                                return -1;
                                        if (nodes[kids[1]].empty) { // [a NOT <never matches>] => a
                                                st.push_back(kids[0]);
                                                continue;
                                        }
                                        n.kids = kids; // {required, excluded}
                                        n.empty = nodes[kids[0]].empty;
                                        n.cost = nodes[kids[0]].cost; // exec.cpp:55-60
                                } else
                                        return -1;
                        }
                        nodes.push_back(std::move(n));
                        st.push_back((int)nodes.size() - 1);
                }
                return st.size() == 1 ? st[0] : -1;
        }

        // ---- general trees: what the CNF lowering does not take (matchsome, NOT / Optional of any subtree, AND under OR ...) runs as
        // TASK_FUSED with a truth table over the presence of the query's distinct terms (<= FUS_MAX_SLOTS, no multi-word phrase).
        struct TruthPlan {
                std::vector<uint32_t> slots;               // distinct terms, order of first appearance
                std::vector<uint32_t> leaves, leaf_tok;    // scorer leaves (positive TERM nodes) in tree order, and their program tokens
                std::vector<uint32_t> leaf_slot;
                uint32_t tt[8] = {};
                std::vector<std::array<uint32_t, 8>> ctt;
        };
        struct TruthBuilder {
                const std::vector<PNode> &nodes;
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
                        const PNode &x = nodes[ni];
                        if (x.op == TRI_OP_TERM || (x.op == TRI_OP_PHRASE && x.kids.size() == 1)) {
                                const PNode &t = x.op == TRI_OP_TERM ? x : nodes[x.kids[0]];
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
                        for (size_t k = 0; k < x.kids.size(); ++k)
                                scan(x.kids[k], positive && !(x.op == TRI_OP_NOT && k == 1));
                }
                bool eval(int ni, uint32_t p) const {
                        const PNode &x = nodes[ni];
                        switch (x.op) {
                                case TRI_OP_TERM:
                                        return (p >> slot_const(x.term)) & 1u;
                                case TRI_OP_PHRASE:
                                        return (p >> slot_const(nodes[x.kids[0]].term)) & 1u;
                                case TRI_OP_AND:
                                        for (int k : x.kids)
                                                if (!eval(k, p))
                                                        return false;
                                        return true;
                                case TRI_OP_OR:
                                        for (int k : x.kids)
                                                if (eval(k, p))
                                                        return true;
                                        return false;
                                case TRI_OP_SOME: {
                                        uint32_t c = 0;
                                        for (int k : x.kids)
                                                c += eval(k, p) ? 1u : 0u;
                                        return c >= x.term;
                                }
                                case TRI_OP_NOT: // Filter (docset_iterators.cpp:652-677)
                                        return eval(x.kids[0], p) && !eval(x.kids[1], p);
                                case TRI_OP_OPT: // Optional (docset_iterators.h:174-206): the documents of main
                                        return eval(x.kids[0], p);
                        }
                        return false;
                }
                uint32_t slot_const(uint32_t term) const {
                        for (size_t i = 0; i < tp.slots.size(); ++i)
                                if (tp.slots[i] == term)
                                        return (uint32_t)i;
                        return 0;
                }
                // the scorer leaves that sit on a document of pattern p, through the tree (node ni matches p): what the reference's score() /
                // collect_doc_matching_terms recursion reaches (docset_iterators_scorers.cpp:38-57, 77-104, 107-193; queryexec_ctx.cpp:382-520)
                void collect(int ni, uint32_t p, uint32_t &mask) const {
                        const PNode &x = nodes[ni];
                        switch (x.op) {
                                case TRI_OP_TERM:
                                case TRI_OP_PHRASE:
                                        if (leaf_of_node[ni] >= 0)
                                                mask |= 1u << leaf_of_node[ni];
                                        break;
                                case TRI_OP_AND:
                                        for (int k : x.kids)
                                                collect(k, p, mask);
                                        break;
                                case TRI_OP_OR:
                                case TRI_OP_SOME:
                                        for (int k : x.kids)
                                                if (eval(k, p))
                                                        collect(k, p, mask);
                                        break;
                                case TRI_OP_NOT:
                                        collect(x.kids[0], p, mask);
                                        break;
                                case TRI_OP_OPT:
                                        collect(x.kids[0], p, mask);
                                        if (eval(x.kids[1], p))
                                                collect(x.kids[1], p, mask);
                                        break;
                        }
                }
        };
        bool build_truth(const std::vector<PNode> &nodes, int root, TruthPlan &tp) {
                TruthBuilder tb{nodes, tp, std::vector<int>(nodes.size(), -1)};
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
} // namespace

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
        // the ScorerWeight contribution of one term (IndexSourceTermsScorer::new_scorer_weight sums it over a phrase's terms):
        // BM25 similarity.h:179-181 (float math), TF-IDF :85-87 (double), Trivial has none
        auto term_weight = [&](const uint32_t df) -> double {
                if (similarity == TRI_SIM_TFIDF)
                        return std::log((double)((uint64_t)ix->info.docs_cnt + 1) / (double)(df + 1)) + 1.0;
                if (similarity == TRI_SIM_TRIVIAL)
                        return 0.0;
                const float num = (float)((uint64_t)ix->info.docs_cnt - (uint64_t)df) + 0.5f;
                const float den = (float)df + 0.5f;
                return (double)std::log(1 + num / den);
        };
        tri_dev *dev = ix->dev;
        HIP_TRY(hipSetDevice(dev->device));
        auto b = std::make_unique<tri_batch>();
        b->ix = ix;
        b->flags = flags;
        b->topk = topk;
        b->similarity = similarity;
        b->nq = nq;
        b->slot_of_query.assign(nq, UINT32_MAX);
        b->qstatus.assign(nq, TRI_OK);
        // a query shape the planner does not lower does not fail the batch: the query is left out (status TRI_ERR_UNSUPPORTED, no matches,
        // tri_batch_query_status) and the caller keeps its CPU span for it; tri_last_error() describes the last such query
        auto leave_out = [&](const size_t qi) {
                b->qstatus[qi] = TRI_ERR_UNSUPPORTED;
                ++b->info.unsupported_queries;
        };
        struct Tmp {
                DevQuery q;
                uint64_t cost;
                uint32_t nlead;
                bool fusable; // AccumulatedScore + top-K, no phrase, <= FUS_MAX_SLOTS distinct terms: may run as TASK_FUSED
                bool truth;   // a general tree: runs as TASK_FUSED whatever its density (there is no other path for it)
                DevFused fz;
        };
#ifdef TRI_CREATE_TIMES // (debug builds: where tri_batch_create's time goes, to stderr)
        auto ct_last = std::chrono::steady_clock::now();
        auto ct_mark = [&](const char *what) {
                const auto now = std::chrono::steady_clock::now();
                fprintf(stderr, "  create: %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - ct_last).count());
                ct_last = now;
        };
#define CT_MARK(x) ct_mark(x)
#else
#define CT_MARK(x)
#endif
        // ---- lowering, query by query.  Queries are independent, and a query costs about 0.6 us of host time (tree, groups, slot map: small
        //      allocations) — 9 ms for 16 384 queries, three times the GPU step they compile to: contiguous ranges of the batch are lowered by
        //      a few host threads, each into a fragment of its own (offsets relative to the fragment), and the fragments are joined in order.
        struct Frag {
                std::vector<Tmp> tmp;
                std::vector<uint32_t> qterms, pterms, sterms;
                std::vector<double> sweights;
                std::vector<DevPhrase> phrases;
                uint64_t term_bytes = 0, term_bytes_phrase_hits = 0;
                uint32_t rich_R = 0;
                bool rich_allow = false;
                std::vector<size_t> left_out; // queries the planner does not lower (status TRI_ERR_UNSUPPORTED)
                int rc = TRI_OK;
                std::string err; // the fragment's last error text (fail() keeps it per thread)
        };
        auto lower_range = [&](const size_t q_lo, const size_t q_hi, Frag &f) -> int {
                std::vector<PNode> nodes;
                for (size_t qi = q_lo; qi < q_hi; ++qi) {
                        const tri_query &tq = queries[qi];
                        if ((uint64_t)tq.prog_off + tq.prog_len > prog_len || !tq.prog_len)
                                return fail(TRI_ERR_INVALID, "query %zu: program slice out of range", qi);
                        nodes.clear();
                        const int root = parse_program(ix, prog + tq.prog_off, tq.prog_len, nodes);
                        if (root < 0)
                                return fail(TRI_ERR_INVALID, "query %zu: malformed postfix program", qi);
                        const PNode &r = nodes[root];
                        if (r.empty)
                                continue; // matches nothing (compiles to constfalse in the reference)
                        // ---- conjunctive normal form over terms: AND of (term | OR of terms); a root OR is one group
                        std::vector<std::vector<uint32_t>> groups;
                        std::vector<uint32_t> leaves;     // every TERM leaf in evaluation order: one scorer each
                        std::vector<uint32_t> leaf_tok;   // ... and the program token it came from
                        struct PhraseTmp {
                                std::vector<uint32_t> terms;
                                double weight;
                        };
                        std::vector<uint32_t> ts_tok; // (scratch of add_group: token indices parallel to ts)
                        std::vector<PhraseTmp> qphrases;
                        auto add_group = [&](const PNode &g) -> bool {
                                std::vector<uint32_t> ts;
                                if (g.op == TRI_OP_PHRASE && g.kids.size() > 1) {
                                        // Phrase = conjunction of its terms + a positional constraint on the matches (k_phrase);
                                        // it scores as ONE iterator with the summed idf (docset_iterators_scorers.cpp:195-228)
                                        PhraseTmp ph;
                                        ph.weight = 0;
                                        for (int k : g.kids) {
                                                const uint32_t x = nodes[k].term;
                                                ph.terms.push_back(x);
                                                const uint32_t df = ix->terms[x].documents;
                                                ph.weight += term_weight(df);
                                                bool dup = false;
                                                for (const auto &og : groups)
                                                        dup |= og.size() == 1 && og[0] == x;
                                                if (!dup)
                                                        groups.push_back({x});
                                        }
                                        if (weights) // the PHRASE token's own ScorerWeight, when the caller supplies weights (by token position: two phrases
                                                     // that start with the same term keep their own weights)
                                                ph.weight = weights[tq.prog_off + g.tok];
                                        qphrases.push_back(std::move(ph));
                                        return true;
                                }
                                ts_tok.clear();
                                if (g.op == TRI_OP_PHRASE) {
                                        ts.push_back(nodes[g.kids[0]].term); // a one-word phrase is a term (exec.cpp: phrase of size 1)
                                        ts_tok.push_back(nodes[g.kids[0]].tok);
                                } else if (g.op == TRI_OP_TERM) {
                                        ts.push_back(g.term);
                                        ts_tok.push_back(g.tok);
                                } else if (g.op == TRI_OP_OR) {
                                        for (int k : g.kids) {
                                                if (nodes[k].op != TRI_OP_TERM)
                                                        return false;
                                                ts.push_back(nodes[k].term);
                                                ts_tok.push_back(nodes[k].tok);
                                        }
                                } else
                                        return false;
                                leaves.insert(leaves.end(), ts.begin(), ts.end());
                                leaf_tok.insert(leaf_tok.end(), ts_tok.begin(), ts_tok.end());
                                // a term repeated inside a group, or a single-term group seen before, adds nothing to the docID set
                                std::vector<uint32_t> u;
                                for (uint32_t x : ts)
                                        if (std::find(u.begin(), u.end(), x) == u.end())
                                                u.push_back(x);
                                if (u.size() == 1)
                                        for (const auto &og : groups)
                                                if (og.size() == 1 && og[0] == u[0])
                                                        return true;
                                groups.push_back(std::move(u));
                                return true;
                        };
                        // logicalnot at the root or under an AND: its required side joins the conjunction, its excluded side (a term or an
                        // OR of terms) joins the query's excluded set: A B -C == A ∧ B ∧ ¬C (Filter semantics, docset_iterators.cpp:652-677)
                        bool ok = true;
                        std::vector<uint32_t> negs, opts, opt_tok;
                        std::function<void(int)> lower = [&](int ni) {
                                const PNode &x = nodes[ni];
                                if (x.op == TRI_OP_OPT) {
                                        // Optional(main, opt): the documents of main; opt's terms score (and are reported) where they match —
                                        // exactly how k_score / k_rich treat a term a match does not hold
                                        lower(x.kids[0]);
                                        const PNode &e = nodes[x.kids[1]];
                                        if (e.op == TRI_OP_TERM) {
                                                opts.push_back(e.term);
                                                opt_tok.push_back(e.tok);
                                        } else if (e.op == TRI_OP_PHRASE && e.kids.size() == 1) {
                                                opts.push_back(nodes[e.kids[0]].term);
                                                opt_tok.push_back(nodes[e.kids[0]].tok);
                                        } else if (e.op == TRI_OP_OR) {
                                                for (int k : e.kids) {
                                                        if (nodes[k].op != TRI_OP_TERM)
                                                                ok = false;
                                                        else {
                                                                opts.push_back(nodes[k].term);
                                                                opt_tok.push_back(nodes[k].tok);
                                                        }
                                                }
                                        } else
                                                ok = false;
                                } else if (x.op == TRI_OP_NOT) {
                                        lower(x.kids[0]);
                                        const PNode &e = nodes[x.kids[1]];
                                        if (e.op == TRI_OP_TERM)
                                                negs.push_back(e.term);
                                        else if (e.op == TRI_OP_PHRASE && e.kids.size() == 1)
                                                negs.push_back(nodes[e.kids[0]].term);
                                        else if (e.op == TRI_OP_OR) {
                                                for (int k : e.kids) {
                                                        if (nodes[k].op != TRI_OP_TERM)
                                                                ok = false;
                                                        else
                                                                negs.push_back(nodes[k].term);
                                                }
                                        } else
                                                ok = false;
                                } else if (x.op == TRI_OP_AND) {
                                        for (int k : x.kids)
                                                lower(k);
                                } else
                                        ok &= add_group(x);
                        };
                        lower(root);
                        if (ok && !groups.empty()) // (a general tree — below — counts every term once through its slot list)
                                for (size_t oi = 0; oi < opts.size(); ++oi)
                                        if (const uint32_t x = opts[oi]; ix->terms[x].documents) {
                                                leaves.push_back(x); // one more scorer / reportable term each; never part of the docID set
                                                leaf_tok.push_back(opt_tok[oi]);
                                                if (mode != TRI_FLAG_DOCUMENTS_ONLY)
                                                        f.term_bytes += ix->docbytes[x]; // its postings are read by k_score / k_rich
                                        }
                        TruthPlan tp;
                        bool truth = false;
                        if (!ok || groups.empty()) {
                                // not a CNF of terms: a general tree over <= FUS_MAX_SLOTS distinct terms runs off a truth table (k_fused.hpp)
                                if (!build_truth(nodes, root, tp)) {
                                        f.left_out.push_back(qi);
                                        fail(TRI_ERR_UNSUPPORTED, "query %zu: lowered so far: AND of terms / phrases / OR-of-terms groups, a root OR of terms, NOT (at the root or under AND) of a term or an OR of terms, <optional> terms under AND; and — no multi-word phrase, <= %u distinct terms, <= %u scored leaves — any tree of AND / OR / NOT / <optional> / matchsome", qi, FUS_MAX_SLOTS, FUS_MAX_LEAVES);
                                        continue;
                                }
                                truth = true;
                                groups.assign(1, tp.slots); // (one group of every slot: the bookkeeping below — term list, cost, output bound — sees a union)
                                negs.clear();
                                leaves = tp.leaves;
                                leaf_tok = tp.leaf_tok;
                                qphrases.clear();
                        }
                        auto gcost = [&](const std::vector<uint32_t> &g) {
                                uint64_t c = 0;
                                for (uint32_t x : g)
                                        c += ix->terms[x].documents;
                                return c;
                        };
                        std::stable_sort(groups.begin(), groups.end(), [&](const auto &x, const auto &y) { return gcost(x) < gcost(y); });
                        std::vector<uint32_t> uniq; // terms group by group, QT_GROUP on the first of each group
                        for (const auto &g : groups)
                                for (size_t i = 0; i < g.size(); ++i)
                                        uniq.push_back(g[i] | (i == 0 ? QT_GROUP : 0u));
                        {
                                // the excluded terms: one more group, the last, marked QT_NOT
                                std::vector<uint32_t> u;
                                for (uint32_t x : negs)
                                        if (ix->terms[x].documents && std::find(u.begin(), u.end(), x) == u.end())
                                                u.push_back(x);
                                for (size_t i = 0; i < u.size(); ++i)
                                        uniq.push_back(u[i] | (i == 0 ? (QT_GROUP | QT_NOT) : 0u));
                        }
                        if (uniq.size() > MAX_QTERMS) {
                                f.left_out.push_back(qi);
                                fail(TRI_ERR_UNSUPPORTED, "query %zu: more than %u terms", qi, MAX_QTERMS);
                                continue;
                        }
                        // (default mode: the reportable terms — every postings iterator collect_doc_matching_terms can reach (queryexec_ctx.cpp:382-520):
                        //  group members and phrase terms, not the excluded side of a NOT —, distinct, in order of first appearance; counted before
                        //  anything of the query is recorded, so that a query with too many of them can still be left out cleanly)
                        std::vector<uint32_t> rt;
                        if (rich) {
                                auto add = [&](uint32_t x) {
                                        if (std::find(rt.begin(), rt.end(), x) == rt.end())
                                                rt.push_back(x);
                                };
                                for (uint32_t pi = 0; pi < tq.prog_len; ++pi) {
                                        const uint32_t tok = prog[tq.prog_off + pi];
                                        if ((tok >> 28) != TRI_OP_TERM)
                                                continue;
                                        const uint32_t x = tok & 0x0fffffffu;
                                        bool positive = std::find(leaves.begin(), leaves.end(), x) != leaves.end();
                                        for (const auto &ph : qphrases)
                                                positive |= std::find(ph.terms.begin(), ph.terms.end(), x) != ph.terms.end();
                                        if (positive)
                                                add(x);
                                }
                                if (rt.size() > 16) {
                                        f.left_out.push_back(qi);
                                        fail(TRI_ERR_UNSUPPORTED, "query %zu: more than 16 reportable terms", qi);
                                        continue;
                                }
                        }
                        const uint32_t nlead = (uint32_t)groups[0].size();
                        const uint64_t lead_docs = gcost(groups[0]);
                        Tmp t;
                        if (!qphrases.empty() && ix->codec == TRI_CODEC_LUCENE && !ix->d_hdir)
                                return fail(TRI_ERR_INVALID, "query %zu: phrase over a LUCENE segment that was uploaded without hits.data", qi);
                        t.q.phrase_base = (uint32_t)f.phrases.size();
                        t.q.nphrases = (uint32_t)qphrases.size();
                        for (const auto &ph : qphrases) {
                                f.phrases.push_back({(uint32_t)f.pterms.size(), (uint32_t)ph.terms.size(), ph.weight});
                                for (uint32_t x : ph.terms) {
                                        f.pterms.push_back(x);
                                        f.term_bytes += ix->hitbytes[x]; // phrase queries also stream the hit bytes
                                        f.term_bytes_phrase_hits += ix->hitbytes[x];
                                }
                        }
                        t.q.score_base = (uint32_t)f.sterms.size();
                        t.q.nscore = 0;
                        if (rich) {
                                for (uint32_t x : rt) {
                                        f.sterms.push_back(x);
                                        f.term_bytes += ix->hitbytes[x]; // the hits of every reported term are read
                                }
                                t.q.nscore = (uint32_t)rt.size();
                                f.rich_R = std::max<uint32_t>(f.rich_R, t.q.nscore);
                        }
                        if (scored) {
                                // one scorer per PostingsListIterator of the conjunction, summed in iterator order
                                // (docset_iterators_scorers.cpp:173-193); weight = BM25 idf (similarity.h:179-181, float math)
                                // unless the caller supplied ScorerWeights per TERM token
                                std::vector<std::pair<uint32_t, double>> sc;
                                for (size_t li = 0; li < leaves.size(); ++li) // caller-provided weights: the leaf's OWN TERM token (a term that also sits inside a
                                                                              // phrase or on an excluded side has another token with another weight)
                                        sc.emplace_back(leaves[li], weights ? weights[tq.prog_off + leaf_tok[li]] : term_weight(ix->terms[leaves[li]].documents));
                                for (auto &e : sc) {
                                        f.sterms.push_back(e.first);
                                        f.sweights.push_back(e.second);
                                }
                                t.q.nscore = (uint32_t)sc.size();
                        }
                        // ---- slot map for the one-pass scored path (k_fused.hpp): the query's distinct terms, CNF terms first
                        t.fusable = false;
                        t.truth = truth;
                        t.fz = DevFused{};
                        if (truth) {
                                DevFused &z = t.fz;
                                z.nslots = (uint32_t)tp.slots.size();
                                z.hw = 0; // (general trees run in their own instantiation, 32-bit window words)
                                z.fbits = z.nslots <= 4 ? 8u : 4u;
                                z.cap = (1u << z.fbits) - 2u;
                                if (dev->opt.fused_freq_cap && dev->opt.fused_freq_cap < z.cap)
                                        z.cap = (uint32_t)dev->opt.fused_freq_cap;
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
                                uint32_t g0 = 0;
                                for (uint32_t sl : tp.leaf_slot)
                                        g0 |= 1u << sl;
                                for (uint32_t p = 0; p < npat; ++p)
                                        if (matches(p) && !(p & g0))
                                                g0 = npat - 1;
                                auto add_req = [&](uint32_t gs) {
                                        z.gslots[z.nreq] = gs;
                                        for (uint32_t sl = 0; sl < z.nslots; ++sl)
                                                if ((gs >> sl) & 1u)
                                                        z.gmask[z.nreq] |= fm << (sl * z.fbits);
                                        ++z.nreq;
                                };
                                add_req(g0);
                                for (uint32_t sl = 0; sl < z.nslots && z.nreq < FUS_MAX_SLOTS; ++sl) {
                                        bool all = g0 != (1u << sl);
                                        for (uint32_t p = 0; p < npat && all; ++p)
                                                all = !matches(p) || ((p >> sl) & 1u);
                                        if (all)
                                                add_req(1u << sl);
                                }
                                t.fusable = true;
                        } else if (scored && topk && qphrases.empty() && dev->opt.fused) {
                                std::vector<uint32_t> slots;
                                auto slot_of = [&](uint32_t term) {
                                        for (size_t i = 0; i < slots.size(); ++i)
                                                if (slots[i] == term)
                                                        return (uint32_t)i;
                                        slots.push_back(term);
                                        return (uint32_t)slots.size() - 1;
                                };
                                for (uint32_t tt : uniq)
                                        slot_of(tt & QT_TERM);
                                for (uint32_t x : leaves)
                                        slot_of(x);
                                if (slots.size() <= FUS_MAX_SLOTS) {
                                        DevFused &z = t.fz;
                                        z.nslots = (uint32_t)slots.size();
                                        z.hw = (dev->opt.fused_halfwords && z.nslots <= 5) ? 1u : 0u;
                                        z.fbits = z.hw ? std::min(8u, 16u / z.nslots) : (z.nslots <= 4 ? 8u : 4u);
                                        z.cap = (1u << z.fbits) - 2u;
                                        if (dev->opt.fused_freq_cap && dev->opt.fused_freq_cap < z.cap)
                                                z.cap = (uint32_t)dev->opt.fused_freq_cap;
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
                                        t.fusable = z.nreq >= 1 && nreq_groups <= FUS_MAX_SLOTS;
                                }
                        }
                        t.q.fused_idx = 0;
                        t.q.pad0 = 0;
                        t.q.nterms = (uint32_t)uniq.size();
                        t.q.term_base = (uint32_t)f.qterms.size();
                        t.q.out_cap = 0;
                        t.q.out_off = 0;
                        t.q.qid = (uint32_t)qi;
                        t.cost = 0;
                        t.nlead = nlead;
                        {
                                std::vector<uint32_t> seen;
                                for (uint32_t tt : uniq) {
                                        const uint32_t term = tt & QT_TERM;
                                        f.qterms.push_back(tt);
                                        if (std::find(seen.begin(), seen.end(), term) == seen.end()) {
                                                seen.push_back(term);
                                                f.term_bytes += ix->docbytes[term];
                                        }
                                }
                                // cost estimate: the lead group is decoded fully; every other list costs min(its blocks x 32, lead docs x 32)
                                for (size_t i = 0; i < uniq.size(); ++i) {
                                        const DevTerm &tk = ix->terms[uniq[i] & QT_TERM];
                                        t.cost += i < nlead ? tk.documents : 32ull * std::min<uint64_t>(tk.nblocks, lead_docs);
                                }
                        }
                        f.tmp.push_back(t);
                }
                return TRI_OK;
        };
        const size_t nthreads = std::max<size_t>(1, std::min<size_t>({(size_t)8, nq / 1024, (size_t)std::max(1u, std::thread::hardware_concurrency())}));
        std::vector<Frag> frags(nthreads);
        {
                auto work = [&](const size_t k) {
                        Frag &f = frags[k];
                        f.rc = lower_range(nq * k / nthreads, nq * (k + 1) / nthreads, f);
                        f.err = tri_last_error();
                };
                std::vector<std::thread> pool;
                for (size_t k = 1; k < nthreads; ++k)
                        pool.emplace_back(work, k);
                work(0);
                for (auto &th : pool)
                        th.join();
        }
        std::vector<Tmp *> tmp; // every lowered query, in query order (the records stay in their fragments)
        for (Frag &f : frags) {
                if (f.rc != TRI_OK)
                        return fail(f.rc, "%s", f.err.c_str());
                const uint32_t qb = (uint32_t)b->qterms.size(), sb = (uint32_t)b->sterms.size(), pb = (uint32_t)b->phrases.size(), ptb = (uint32_t)b->pterms.size();
                for (Tmp &t : f.tmp) {
                        t.q.term_base += qb;
                        t.q.score_base += sb;
                        t.q.phrase_base += pb;
                        tmp.push_back(&t);
                }
                for (DevPhrase ph : f.phrases) {
                        ph.term_base += ptb;
                        b->phrases.push_back(ph);
                }
                b->qterms.insert(b->qterms.end(), f.qterms.begin(), f.qterms.end());
                b->pterms.insert(b->pterms.end(), f.pterms.begin(), f.pterms.end());
                b->sterms.insert(b->sterms.end(), f.sterms.begin(), f.sterms.end());
                b->sweights.insert(b->sweights.end(), f.sweights.begin(), f.sweights.end());
                b->term_bytes += f.term_bytes;
                b->term_bytes_phrase_hits += f.term_bytes_phrase_hits;
                b->rich_R = std::max(b->rich_R, f.rich_R);
                b->rich_allow |= f.rich_allow;
                for (const size_t qi : f.left_out)
                        leave_out(qi);
                if (!f.left_out.empty())
                        fail(TRI_ERR_UNSUPPORTED, "%s", f.err.c_str()); // (tri_last_error() describes the last query that was left out)
        }
        CT_MARK("lowering");
        // (heaviest first — by index: a Tmp is 800 bytes, sorting the records themselves was a third of tri_batch_create)
        std::vector<uint32_t> qorder(tmp.size());
        std::iota(qorder.begin(), qorder.end(), 0u);
        std::stable_sort(qorder.begin(), qorder.end(), [&](const uint32_t a, const uint32_t c) { return tmp[a]->cost > tmp[c]->cost; });
        uint64_t off = 0;
        b->plan.reserve(tmp.size());
        // cut every query into tasks of roughly TASK_COST postings, then schedule heaviest first
        constexpr uint64_t TASK_COST = 96 * 1024;
        // planner thresholds: tri_dev_set_option (defaults: tri_options)
        const uint64_t DENSE_MIN_POSTINGS = dev->opt.dense_min_postings;
        const uint64_t DENSE_TASK_COST = std::max<uint64_t>(1, dev->opt.dense_task_cost); // bitmap-window tasks stage their terms once: two windows of a head pair per task
        // TASK_DENSE (bitmap windows) when the lead group is an OR (it has to be materialised as a set anyway), or when every other list is
        // within a factor 32 of the lead (no block could be skipped) and there is enough work per docID window to keep 256 lanes busy;
        // TASK_FUSED when such a query asks for a top-K (or is a general tree)
        struct Class {
                uint64_t sumdf, lead_docs;
                uint32_t last_doc; // no match beyond the (required) group whose lists end first
                bool dense, fuse;
        };
        auto classify = [&](const Tmp &t) {
                const uint32_t *qt = &b->qterms[t.q.term_base];
                Class c{0, 0, 0xffffffffu, t.q.nterms >= 2, false};
                for (uint32_t k = 0; k < t.nlead; ++k)
                        c.lead_docs += ix->terms[qt[k] & QT_TERM].documents;
                uint32_t glast = 0;
                bool in_neg = false;
                for (uint32_t k = 0; k < t.q.nterms; ++k) {
                        const DevTerm &tk = ix->terms[qt[k] & QT_TERM];
                        c.sumdf += tk.documents;
                        c.dense &= tk.nblocks <= c.lead_docs;
                        if (k && (qt[k] & QT_GROUP)) {
                                c.last_doc = std::min(c.last_doc, glast);
                                glast = 0;
                                in_neg = qt[k] & QT_NOT;
                        }
                        if (!in_neg)
                                glast = std::max(glast, ix->h_blk_last[tk.first_block + tk.nblocks - 1]);
                }
                if (!in_neg)
                        c.last_doc = std::min(c.last_doc, glast);
                c.dense &= c.sumdf >= DENSE_MIN_POSTINGS;
                c.dense |= t.nlead > 1;
                c.fuse = t.truth || (c.dense && t.fusable && (dev->opt.fused != 2 || t.fz.nreq == 1)); // (fused == 2: only pure unions)
                return c;
        };
        // one-pass tasks stage the query (slot map, score tables) once per task: the longer the task the better, as long as the batch still
        // cuts into a couple of tasks per workgroup the device holds (measured at cfg3: 512 K postings per task 55.4 ms, 1 M 51.1, 2 M 49.2,
        // 4 M 48.0, 8 M and more 47.1).  fused_task_cost = 0 (the default): sized from the batch; otherwise as given
        uint64_t FUSED_TASK_COST = dev->opt.fused_task_cost;
        std::vector<Class> classes(tmp.size()); // (once per query: the passes below and the task loop all ask)
        for (size_t i = 0; i < tmp.size(); ++i)
                classes[i] = classify(*tmp[i]);
        uint64_t onepass_queries = 0;
        for (const Class &c : classes)
                onepass_queries += c.fuse ? 1 : 0;
        // k_planes: docID ranges per query.  A task has fixed costs (seed pass, end-of-task imbalance: about 140 us), the kernel's tail is its
        // longest tasks: two ranges when the batch brings ten or more tasks per resident workgroup anyway, three when it does not (measured,
        // cfg3's mix: 8192 queries 2 > 3 > 4; 3750 queries 6.5 / 5.9 / 6.2 ms for 2 / 3 / 4; 1024 queries 2.11 / 1.97 / 1.96)
        const uint64_t PLANES_SPLIT = dev->opt.planes_split ? dev->opt.planes_split : (2 * onepass_queries >= 10ull * (uint64_t)dev->cus * PLK_WGS_PER_CU ? 2 : 3);
        if (!FUSED_TASK_COST) {
                uint64_t fused_postings = 0;
                for (size_t i = 0; i < tmp.size(); ++i)
                        if (classes[i].fuse)
                                for (uint32_t sidx = 0; sidx < tmp[i]->fz.nslots; ++sidx)
                                        fused_postings += ix->terms[tmp[i]->fz.term[sidx]].documents;
                const uint64_t want_tasks = 2ull * (uint64_t)dev->cus * FUS_WGS_PER_CU;
                FUSED_TASK_COST = std::min<uint64_t>(8u << 20, std::max<uint64_t>(256u << 10, fused_postings / want_tasks));
        }
        // ---- term planes (k_planes.hpp): a head term is decoded once per launch for all the queries that name it.  Eligible: an indexed list
        //      of at least docs_cnt / plane_div documents; built when the batch's uses repay one decode of the list (a use as a bitmap-window
        //      or one-pass slot saves a whole walk, a use as the probed side of a candidate tile saves at most 32 postings per lead document)
        const uint64_t planes_opt = dev->opt.planes;
        const uint64_t plane_min_df = dev->opt.plane_div ? std::max<uint64_t>(1, ix->info.docs_cnt / dev->opt.plane_div) : UINT64_MAX;
        auto plane_ok = [&](uint32_t term) {
                const DevTerm &tk = ix->terms[term];
                return planes_opt && tk.documents && tk.documents >= plane_min_df;
        };
        std::unordered_map<uint32_t, uint64_t> plane_benefit;
        struct QUse {
                uint32_t qpos, term;
        };
        struct FUse {
                uint32_t fidx, slot, term;
        };
        std::vector<QUse> quses;
        std::vector<FUse> fuses;
        std::vector<std::pair<uint64_t, uint32_t>> order; // (task cost, task index)
        CT_MARK("query order + classes");
        for (const uint32_t qo : qorder) {
                Tmp &t = *tmp[qo];
                const uint32_t slot = (uint32_t)b->plan.size();
                b->slot_of_query[t.q.qid] = slot;
                const uint32_t *qt = &b->qterms[t.q.term_base];
                const DevTerm &lead = ix->terms[qt[0] & QT_TERM];
                const uint32_t nlead = t.nlead;
                const Class cls = classes[qo];
                const uint64_t sumdf = cls.sumdf;
                const uint32_t last_doc = cls.last_doc;
                const bool dense = cls.dense, fuse = cls.fuse;
                if (fuse) {
                        // a CNF query whose top-K runs over bit planes (k_planes): its head terms read from the batch's term planes, the
                        // others (at most PLK_MAX_SPARSE) decoded per window into LDS planes
                        uint32_t nsparse = 0;
                        for (uint32_t sidx = 0; sidx < t.fz.nslots; ++sidx) {
                                t.fz.plane[sidx] = PL_NONE;
                                nsparse += plane_ok(t.fz.term[sidx]) ? 0u : 1u;
                        }
                        const bool pk = !t.truth && (planes_opt & 4u) && nsparse <= PLK_MAX_SPARSE && ix->max_doc < 0x7fff0000u; // (list entries are docID << 1 | flag)
                        {
                                const uint32_t fm = (1u << t.fz.fbits) - 1u;
                                t.fz.negslots = 0;
                                for (uint32_t sidx = 0; sidx < t.fz.nslots; ++sidx)
                                        if ((t.fz.nmask >> (sidx * t.fz.fbits)) & fm)
                                                t.fz.negslots |= 1u << sidx;
                        }
                        // every list of the slot map is read once (the optional terms too)
                        uint64_t slotdf = 0;
                        for (uint32_t sidx = 0; sidx < t.fz.nslots; ++sidx) {
                                slotdf += ix->terms[t.fz.term[sidx]].documents;
                                (pk ? b->term_bytes_planes : b->term_bytes_fused) += ix->docbytes[t.fz.term[sidx]];
                                if (pk && plane_ok(t.fz.term[sidx])) {
                                        plane_benefit[t.fz.term[sidx]] += ix->terms[t.fz.term[sidx]].documents;
                                        fuses.push_back({(uint32_t)b->fused.size(), sidx, t.fz.term[sidx]});
                                }
                        }
                        ++(pk ? b->info.planes_queries : b->info.fused_queries);
                        t.q.fused_idx = (uint32_t)b->fused.size();
                        b->fused.push_back(t.fz);
                        t.q.out_off = off;
                        t.q.out_cap = 0; // the docID set is never materialised ...
                        t.q.first_task = (uint32_t)b->tasks.size();
                        const uint32_t fw = pk ? PL_W : FUS_W << t.fz.hw; // documents per window: plane windows, or this query's word width
                        const uint32_t nwin = last_doc / fw + 1;
                        const uint64_t per_win = std::max<uint64_t>(1, slotdf / (ix->info.docs_cnt / fw + 1));
                        // (k_planes' cost is the sweep of the range plus its candidates, not the postings: equal ranges, a few per query)
                        const uint32_t win_per_task = pk && PLANES_SPLIT < 65536 ? (uint32_t)((nwin + PLANES_SPLIT - 1) / PLANES_SPLIT)
                                                                                 : (uint32_t)std::max<uint64_t>(1, FUSED_TASK_COST / per_win);
                        const bool emit = t.fz.mode & FUS_MODE_EMIT; // ... except by a general tree in DocumentsOnly mode: a private region per task,
                                                                     // bounded like TASK_DENSE's by the slots' blocks that reach the task's windows
                        uint32_t ord = 0;
                        for (uint32_t wb = 0; wb < nwin; wb += win_per_task, ++ord) {
                                const uint32_t we = std::min(nwin, wb + win_per_task);
                                uint64_t b1 = 0;
                                if (emit)
                                        for (uint32_t sidx = 0; sidx < t.fz.nslots; ++sidx) {
                                                const DevTerm &tk = ix->terms[t.fz.term[sidx]];
                                                const uint32_t *lb = &ix->h_blk_last[tk.first_block];
                                                b1 += (uint64_t)(std::lower_bound(lb, lb + tk.nblocks, wb * fw) - lb);
                                        }
                                uint64_t entries = 0;
                                if (pk) { // the rows of the decoded slots that can reach the task's docID range: 32 list entries each (k_planes)
                                        for (uint32_t sidx = 0; sidx < t.fz.nslots; ++sidx) {
                                                if (plane_ok(t.fz.term[sidx]))
                                                        continue;
                                                const DevTerm &tk = ix->terms[t.fz.term[sidx]];
                                                const uint32_t *lb = &ix->h_blk_last[tk.first_block];
                                                const uint32_t r0 = (uint32_t)(std::lower_bound(lb, lb + tk.nblocks, wb * fw) - lb);
                                                const uint32_t r1 = (uint32_t)(std::lower_bound(lb + r0, lb + tk.nblocks, we * fw) - lb);
                                                if (r0 < tk.nblocks)
                                                        entries += 32ull * (std::min(r1, tk.nblocks - 1) - r0 + 1);
                                        }
                                        if (entries > 0x7fffffffull)
                                                return fail(TRI_ERR_UNSUPPORTED, "query %u: a task's decoded lists exceed 2^31 entries", t.q.qid);
                                        b->sparse_cap = std::max(b->sparse_cap, (uint32_t)entries);
                                }
                                // (largest first, by postings: for k_planes a poor estimate — its cost is the sweep plus the candidates — but ordering by the
                                //  decoded entries instead measured worse: cfg3's unions 10.6 ms against 9.2)
                                order.emplace_back(per_win * (we - wb), (uint32_t)b->tasks.size());
                                b->tasks.push_back({slot, wb, we, pk ? (t.fz.nslots <= PLK_NS_SMALL ? TASK_PLANES : TASK_PLANES8) : t.fz.mode ? TASK_FUSED_GEN : t.fz.hw ? TASK_FUSED16 : TASK_FUSED, off + (emit ? b1 * 32 + 32ull * ord * t.fz.nslots : 0)});
                        }
                        if (emit) {
                                uint64_t blocks = 0;
                                for (uint32_t sidx = 0; sidx < t.fz.nslots; ++sidx)
                                        blocks += ix->terms[t.fz.term[sidx]].nblocks;
                                t.q.out_cap = (uint32_t)std::min<uint64_t>(0xffffffffull, blocks * 32 + 32ull * (ord + 1) * t.fz.nslots);
                                off += t.q.out_cap;
                        }
                        t.q.ntasks = (uint32_t)b->tasks.size() - t.q.first_task;
                        b->plan.push_back(t.q);
                        continue;
                }
                if (dense) {
                        std::vector<uint32_t> seen;
                        for (uint32_t k = 0; k < t.q.nterms; ++k) {
                                const uint32_t term = qt[k] & QT_TERM;
                                if (std::find(seen.begin(), seen.end(), term) == seen.end()) {
                                        seen.push_back(term);
                                        b->term_bytes_dense += ix->docbytes[term];
                                }
                                if ((planes_opt & 2u) && plane_ok(term)) {
                                        plane_benefit[term] += ix->terms[term].documents;
                                        quses.push_back({t.q.term_base + k, term});
                                }
                        }
                        ++b->info.dense_queries;
                } else {
                        ++b->info.cand_queries;
                        for (uint32_t k = 1; k < t.q.nterms; ++k) { // (the lead list is decoded into the candidate tiles; the others are probed)
                                const uint32_t term = qt[k] & QT_TERM;
                                if ((planes_opt & 1u) && plane_ok(term)) {
                                        plane_benefit[term] += std::min<uint64_t>(ix->terms[term].documents, 32ull * lead.documents);
                                        quses.push_back({t.q.term_base + k, term});
                                }
                        }
                }
                t.q.out_off = off;
                t.q.first_task = (uint32_t)b->tasks.size();
                if (dense) {
                        const uint32_t nwin = last_doc / SPAN_BITS + 1;
                        const uint64_t per_win = std::max<uint64_t>(1, sumdf / (ix->info.docs_cnt / SPAN_BITS + 1));
                        const uint32_t win_per_task = (uint32_t)std::max<uint64_t>(1, DENSE_TASK_COST / per_win);
                        uint32_t ord = 0;
                        uint64_t lead_blocks = 0;
                        for (uint32_t k = 0; k < nlead; ++k)
                                lead_blocks += ix->terms[qt[k] & QT_TERM].nblocks;
                        for (uint32_t wb = 0; wb < nwin; wb += win_per_task, ++ord) {
                                const uint32_t we = std::min(nwin, wb + win_per_task);
                                // matches of windows [wb, we) are lead-group documents of blocks b1 .. (next task's b1) of every
                                // lead list: a private region (+32 slots of slack per lead list and task for the straddling block)
                                uint64_t b1 = 0;
                                for (uint32_t k = 0; k < nlead; ++k) {
                                        const DevTerm &tk = ix->terms[qt[k] & QT_TERM];
                                        const uint32_t *lb = &ix->h_blk_last[tk.first_block];
                                        b1 += (uint64_t)(std::lower_bound(lb, lb + tk.nblocks, wb * SPAN_BITS) - lb);
                                }
                                order.emplace_back(per_win * (we - wb), (uint32_t)b->tasks.size());
                                b->tasks.push_back({slot, wb, we, TASK_DENSE, off + b1 * 32 + 32ull * ord * nlead});
                        }
                        t.q.out_cap = (uint32_t)std::min<uint64_t>(0xffffffffull, lead_blocks * 32 + 32ull * (ord + 1) * nlead);
                } else {
                        const uint32_t ntiles = (lead.nblocks + TILE_BLOCKS - 1) / TILE_BLOCKS;
                        const uint64_t per_tile = std::max<uint64_t>(1, t.cost / ntiles);
                        const uint32_t tiles_per_task = (uint32_t)std::max<uint64_t>(1, TASK_COST / per_tile);
                        for (uint32_t tb = 0; tb < ntiles; tb += tiles_per_task) {
                                const uint32_t te = std::min(ntiles, tb + tiles_per_task);
                                order.emplace_back(per_tile * (te - tb), (uint32_t)b->tasks.size());
                                b->tasks.push_back({slot, tb, te, TASK_CAND, off + (uint64_t)tb * TILE_CANDS});
                        }
                        t.q.out_cap = lead.documents; // |A ∩ …| <= df of the lead
                        if (dev->opt.account_needed_bytes) {
                                // what a perfect gallop must read: the lead list, and of every other list the blocks that can hold a lead
                                // candidate — per lead block the other list's blocks its docID range meets, at most one per candidate
                                // (directories only; a block counts docbytes / nblocks)
                                uint64_t need = ix->docbytes[qt[0] & QT_TERM];
                                const uint32_t *ll = &ix->h_blk_last[lead.first_block];
                                for (uint32_t k = 1; k < t.q.nterms; ++k) {
                                        const DevTerm &tk = ix->terms[qt[k] & QT_TERM];
                                        const uint32_t *ol = &ix->h_blk_last[tk.first_block];
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
                                        need += (uint64_t)((double)ix->docbytes[qt[k] & QT_TERM] * std::min(1.0, (double)blocks / std::max(1u, tk.nblocks)));
                                }
                                b->cand_needed_term_bytes += need;
                        }
                }
                off += t.q.out_cap;
                t.q.ntasks = (uint32_t)b->tasks.size() - t.q.first_task;
                if (t.q.nphrases)
                        for (uint32_t ti = t.q.first_task; ti < t.q.first_task + t.q.ntasks; ++ti)
                                b->ptasks.push_back(ti);
                b->plan.push_back(t.q);
        }
        CT_MARK("tasks");
        std::stable_sort(order.begin(), order.end(), [](const auto &a, const auto &c) { return a.first > c.first; });
        std::vector<uint32_t> sched;
        sched.reserve(order.size());
        for (const auto &o : order)
                if (b->tasks[o.second].kind == TASK_DENSE)
                        sched.push_back(o.second);
        b->n_dense = (uint32_t)sched.size();
        for (const auto &o : order)
                if (b->tasks[o.second].kind == TASK_CAND)
                        sched.push_back(o.second);
        b->n_cand = (uint32_t)sched.size() - b->n_dense;
        for (const auto &o : order)
                if (b->tasks[o.second].kind == TASK_FUSED)
                        sched.push_back(o.second);
        b->n_fused = (uint32_t)sched.size() - b->n_dense - b->n_cand;
        for (const auto &o : order)
                if (b->tasks[o.second].kind == TASK_FUSED16)
                        sched.push_back(o.second);
        b->n_fused16 = (uint32_t)sched.size() - b->n_dense - b->n_cand - b->n_fused;
        for (const auto &o : order)
                if (b->tasks[o.second].kind == TASK_FUSED_GEN)
                        sched.push_back(o.second);
        b->n_fusedgen = (uint32_t)sched.size() - b->n_dense - b->n_cand - b->n_fused - b->n_fused16;
        for (const auto &o : order)
                if (b->tasks[o.second].kind == TASK_PLANES)
                        sched.push_back(o.second);
        b->n_planes = (uint32_t)sched.size() - b->n_dense - b->n_cand - b->n_fused - b->n_fused16 - b->n_fusedgen;
        for (const auto &o : order)
                if (b->tasks[o.second].kind == TASK_PLANES8)
                        sched.push_back(o.second);
        b->n_planes8 = (uint32_t)sched.size() - b->n_dense - b->n_cand - b->n_fused - b->n_fused16 - b->n_fusedgen - b->n_planes;
        // the planes that pay: rows in term order (deterministic), the uses pointed at them
        {
                std::vector<uint32_t> chosen;
                for (const auto &e : plane_benefit)
                        if (e.second >= ix->terms[e.first].documents)
                                chosen.push_back(e.first);
                for (const auto &u : fuses) // (a one-pass slot counts a whole decode: always chosen; kept explicit)
                        if (std::find(chosen.begin(), chosen.end(), u.term) == chosen.end())
                                chosen.push_back(u.term);
                std::sort(chosen.begin(), chosen.end());
                std::unordered_map<uint32_t, uint32_t> row_of;
                for (uint32_t x : chosen) {
                        row_of[x] = (uint32_t)b->plane_terms.size();
                        b->plane_terms.push_back(x);
                        b->plane_decoded_bytes += ix->docbytes[x];
                }
                if (!chosen.empty()) {
                        std::vector<uint32_t> qplane(b->qterms.size(), PL_NONE);
                        for (const auto &u : quses)
                                if (auto it = row_of.find(u.term); it != row_of.end())
                                        qplane[u.qpos] = it->second;
                        for (const auto &u : fuses)
                                b->fused[u.fidx].plane[u.slot] = row_of[u.term];
                        int rcp;
                        if ((rcp = dev_upload(&b->d_qplane, qplane)) || (rcp = dev_upload(&b->d_plane_terms, b->plane_terms)))
                                return rcp;
                }
                if (!chosen.empty() || b->n_planes + b->n_planes8) {
                        // the rows k_term_planes fills, plus an all-zero row: what a k_planes slot WITHOUT term planes reads (so its sweep needs no select)
                        b->plw = ((ix->max_doc >> 17) + 2u) * (SPAN_BITS / 32u); // whole bitmap windows (k_and_dense reads SPAN_WORDS at a time) + a spare one
                        const size_t row = (size_t)PL_PLANES * b->plw * 4;
                        HIP_TRY(pool_alloc(dev, (void **)&b->d_planes, (b->plane_terms.size() + 1) * row + 64));
                        HIP_TRY(hipMemset((uint8_t *)b->d_planes + b->plane_terms.size() * row, 0, row + 64));
                }
        }
        if (b->n_planes + b->n_planes8) {
                HIP_TRY(hipMalloc((void **)&b->d_qthr, (b->plan.size() + 1) * 8));
                b->sparse_cap = (b->sparse_cap + 63u) & ~63u;
                const uint64_t wgs = std::min<uint64_t>(std::max(b->n_planes, b->n_planes8), (uint64_t)dev->cus * PLK_WGS_PER_CU);
                HIP_TRY(pool_alloc(dev, (void **)&b->d_sparse, (wgs * b->sparse_cap + 64) * 4));
        }
        CT_MARK("order + planes + scratch");
        b->out_capacity = off;
        int rc;
        if ((rc = dev_upload(&b->d_plan, b->plan)) || (rc = dev_upload(&b->d_qterms, b->qterms)) || (rc = dev_upload(&b->d_tasks, b->tasks)) ||
            (rc = dev_upload(&b->d_sched, sched)) || (rc = dev_upload(&b->d_fused, b->fused)))
                return rc;
        for (hipEvent_t *e : {&b->ev0, &b->ev_a, &b->ev_b, &b->ev_c, &b->ev_p, &b->ev1, &b->ev_pl, &b->ev_k})
                HIP_TRY(hipEventCreate(e));
        HIP_TRY(pool_alloc(dev, (void **)&b->d_out, (off + 64) * 4));
        HIP_TRY(hipMalloc((void **)&b->d_counts, (b->tasks.size() + 1) * 4));
        HIP_TRY(hipMalloc((void **)&b->d_ticket, 256));
        HIP_TRY(hipMalloc((void **)&b->d_qcounts, (nq + 1) * 8));
        HIP_TRY(hipMemset(b->d_qcounts, 0, (nq + 1) * 8)); // queries that can never match keep count 0
        if (!b->phrases.empty()) {
                if ((rc = dev_upload(&b->d_phrases, b->phrases)) || (rc = dev_upload(&b->d_pterms, b->pterms)) || (rc = dev_upload(&b->d_ptasks, b->ptasks)))
                        return rc;
                if (scored) {
                        HIP_TRY(pool_alloc(dev, (void **)&b->d_pscore, (off + 64) * 8));
                        HIP_TRY(hipMemset(b->d_pscore, 0, (off + 64) * 8));
                }
        }
        if (rich) {
                if ((rc = dev_upload(&b->d_sterms, b->sterms)))
                        return rc;
                b->rich_R = std::max<uint32_t>(b->rich_R, 1);
                HIP_TRY(pool_alloc(dev, (void **)&b->d_rich_present, (off + 64) * 4));
                HIP_TRY(pool_alloc(dev, (void **)&b->d_rich_freq, (off + 64) * 2 * b->rich_R));
                HIP_TRY(hipMalloc((void **)&b->d_task_hits, (b->tasks.size() + 1) * 4));
                HIP_TRY(hipMalloc((void **)&b->d_task_pos_base, (b->tasks.size() + 1) * 8));
                if (b->rich_allow) {
                        HIP_TRY(pool_alloc(dev, (void **)&b->d_rich_allow, (off + 64) * 4));
                        HIP_TRY(hipMemset(b->d_rich_allow, 0xff, (off + 64) * 4)); // (every other query's matches: all terms allowed)
                }
        }
        if (scored) {
                if ((rc = dev_upload(&b->d_sterms, b->sterms)) || (rc = dev_upload(&b->d_sweights, b->sweights)))
                        return rc;
                const size_t nt = b->tasks.size();
                if (!topk)
                        HIP_TRY(pool_alloc(dev, (void **)&b->d_all_scores, (off + 64) * 8));
                HIP_TRY(pool_alloc(dev, (void **)&b->d_part_docs, (nt * topk + 1) * 4));
                HIP_TRY(pool_alloc(dev, (void **)&b->d_part_scores, (nt * topk + 1) * 8));
                HIP_TRY(hipMalloc((void **)&b->d_part_counts, (nt + 1) * 4));
                HIP_TRY(hipMalloc((void **)&b->d_top_docs, (nq * topk + 1) * 4));
                HIP_TRY(hipMalloc((void **)&b->d_top_scores, (nq * topk + 1) * 4));
                HIP_TRY(hipMalloc((void **)&b->d_top_counts, (nq + 1) * 4));
                HIP_TRY(hipMemset(b->d_top_counts, 0, (nq + 1) * 4)); // queries that can never match keep count 0 ...
                HIP_TRY(hipMemset(b->d_top_docs, 0, (nq * topk + 1) * 4)); // ... and zeroed rows (k_topk_merge only writes the rows of queries that have a plan slot;
                HIP_TRY(hipMemset(b->d_top_scores, 0, (nq * topk + 1) * 4)); // the blocks travel whole to the host and to the other ranks)
        }
        CT_MARK("uploads + allocations");
        b->info.nqueries = nq;
        b->info.out_capacity = off;
        b->info.plane_terms = b->plane_terms.size();
        b->info.plane_bytes = (uint64_t)b->plane_terms.size() * PL_PLANES * b->plw * 4;
        b->info.launches = (b->n_dense != 0) + (b->n_cand != 0) + (b->n_fused != 0) + (b->n_fused16 != 0) + (b->n_fusedgen != 0) + (b->n_planes != 0) + (b->n_planes8 != 0) + (!b->plane_terms.empty()) +
                           (!b->ptasks.empty()) + (rich ? 2 : 0) +
                           ((scored && b->n_dense + b->n_cand) ? 1 : 0) + ((scored && topk) ? 1 : 0);
        *out = b.release();
        return TRI_OK;
}

extern "C" void tri_batch_destroy(tri_batch *b) {
        delete b; // ~tri_batch releases the device buffers
}

extern "C" int tri_batch_run(tri_batch *b) {
        if (!b)
                return fail(TRI_ERR_INVALID, "null batch");
        tri_dev *dev = b->ix->dev;
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
        HIP_TRY(hipEventRecord(b->ev0, dev->stream));
        if (n) {
                HIP_TRY(hipMemsetAsync(b->d_ticket, 0, 256, dev->stream));
                // two persistent kernels back to back on the engine stream: bitmap windows (512 threads), then candidate tiles.
                // GOOGLE: matching reads the contiguous delta streams, not the chunks (see tri_index::d_dstream)
                const uint8_t *match_bytes = b->ix->codec == TRI_CODEC_GOOGLE ? b->ix->d_dstream : b->ix->d_index;
                const uint32_t *match_off = b->ix->codec == TRI_CODEC_GOOGLE ? b->ix->d_blk_doff : b->ix->d_blk_off;
                uint32_t dense_wgs = TRI_DENSE_WAVES * 256 / DENSE_WG, cand_wgs = 4; // workgroups per CU
                bool overlap = false;
                if (dev->opt.overlap_dense_wgs && dev->opt.overlap_cand_wgs && b->n_dense && b->n_cand) { // both kernels side by side
                        overlap = true;
                        dense_wgs = (uint32_t)dev->opt.overlap_dense_wgs;
                        cand_wgs = (uint32_t)dev->opt.overlap_cand_wgs;
                }
                if (!b->plane_terms.empty()) {
                        // the head terms the batch shares, decoded once for this launch (every word of every plane is rewritten)
                        const dim3 grid(b->plw / PL_WORDS, (uint32_t)b->plane_terms.size());
                        TRI_LAUNCH(k_term_planes, b->ix->codec, grid, dim3(AND_WG), dev->stream, b->ix->d_index, b->ix->d_blk_last, b->ix->d_blk_off, b->ix->d_blk_rec,
                                   b->ix->d_blk_doff, b->ix->d_win, b->ix->d_terms, b->d_plane_terms, b->d_planes, b->plw);
                        HIP_TRY(hipGetLastError());
                }
                HIP_TRY(hipEventRecord(b->ev_pl, dev->stream));
                hipStream_t cand_stream = dev->stream;
                if (overlap) {
                        HIP_TRY(hipEventRecord(dev->ev_fork, dev->stream));
                        HIP_TRY(hipStreamWaitEvent(dev->stream2, dev->ev_fork, 0));
                        cand_stream = dev->stream2;
                }
                if (b->n_dense) {
                        TRI_LAUNCH(k_and_dense, b->ix->codec, dim3(std::min<uint32_t>(b->n_dense, (uint32_t)dev->cus * dense_wgs)), dim3(DENSE_WG), dev->stream, match_bytes,
                                           b->ix->d_blk_last, match_off, b->ix->d_win, b->ix->d_terms, b->d_plan, b->d_tasks, b->d_sched, b->d_qterms, b->n_dense,
                                           b->d_ticket + 16, b->d_out, b->d_counts, b->ix->d_masked, (const uint32_t *)b->d_qplane, (const uint32_t *)b->d_planes, b->plw);
                        HIP_TRY(hipGetLastError());
                }
                HIP_TRY(hipEventRecord(b->ev_a, dev->stream));
                if (b->n_cand)
                        TRI_LAUNCH(k_and, b->ix->codec, dim3(std::min<uint32_t>(b->n_cand, (uint32_t)dev->cus * cand_wgs)), dim3(AND_WG), cand_stream, match_bytes,
                                           b->ix->d_blk_last, match_off, b->ix->d_win, b->ix->d_terms, b->d_plan, b->d_tasks, b->d_sched + b->n_dense, b->d_qterms,
                                           b->n_cand, b->d_ticket, b->d_out, b->d_counts, b->ix->d_masked, (const uint32_t *)b->d_qplane, (const uint32_t *)b->d_planes, b->plw);
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
                        const uint32_t *fsched = b->d_sched + b->n_dense + b->n_cand + (variant >= 1 ? b->n_fused : 0) + (variant == 2 ? b->n_fused16 : 0);
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
                        const uint32_t *psched = b->d_sched + b->n_dense + b->n_cand + b->n_fused + b->n_fused16 + b->n_fusedgen + (wide ? b->n_planes : 0);
                        const dim3 grid(std::min<uint32_t>(np, (uint32_t)dev->cus * PLK_WGS_PER_CU));
#define TRI_PLANES_ARGS                                                                                                                                      \
        b->ix->d_index, b->ix->d_blk_last, b->ix->d_blk_off, b->ix->d_blk_rec, b->ix->d_blk_doff, b->ix->d_win, b->ix->d_terms, b->d_plan, b->d_fused, b->d_tasks, psched, \
                b->d_sterms, b->d_sweights, np, b->d_ticket + 24 + 2 * wide, b->d_counts, b->topk, b->d_part_docs, b->d_part_scores, b->d_part_counts, b->ix->d_masked,     \
                b->similarity, (const uint32_t *)b->d_planes, b->plw, (uint32_t)b->plane_terms.size(), b->d_sparse, b->sparse_cap, b->d_qthr
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
                if (!b->ptasks.empty()) {
                        // positional constraints: filter + compact the match segments of the queries that hold phrases
                        const uint32_t np = (uint32_t)b->ptasks.size();
                        TRI_LAUNCH(k_phrase, b->ix->codec, dim3(std::min<uint32_t>(np, (uint32_t)dev->cus * PHRASE_WGS_PER_CU)), dim3(AND_WG), dev->stream, b->ix->d_index,
                                           b->ix->d_hits, b->ix->d_blk_hits, b->ix->d_hdir, b->ix->d_blk_last, b->ix->d_blk_off, b->ix->d_win, b->ix->d_terms, b->d_plan, b->d_tasks, b->d_ptasks, np, b->d_phrases, b->d_pterms,
                                           b->d_ticket + 48, b->d_out, b->d_counts, b->d_pscore,
                                           (b->flags & TRI_FLAG_ACCUMULATED_SCORE) ? 65535u : 1u, // exec.cpp:296 trackCnt
                                           b->similarity);
                        HIP_TRY(hipGetLastError());
                }
                HIP_TRY(hipEventRecord(b->ev_p, dev->stream));
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
                        const uint32_t nlegacy = b->n_dense + b->n_cand; // the TASK_FUSED tasks have scored themselves
                        if (nlegacy)
                        TRI_LAUNCH(k_score, b->ix->codec, dim3(std::min<uint32_t>(nlegacy, (uint32_t)dev->cus * SCORE_WGS_PER_CU)), dim3(AND_WG), dev->stream, b->ix->d_index,
                                           b->ix->d_blk_last, b->ix->d_blk_off, b->ix->d_terms, b->d_plan, b->d_tasks, b->d_sched, b->d_sterms, b->d_sweights, nlegacy,
                                           b->d_ticket + 32, b->d_out, b->d_counts, b->topk, b->d_part_docs, b->d_part_scores, b->d_part_counts,
                                           b->d_all_scores, b->d_pscore, b->similarity);
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
                HIP_TRY(hipEventRecord(b->ev_b, dev->stream));
                HIP_TRY(hipEventRecord(b->ev_c, dev->stream));
                HIP_TRY(hipEventRecord(b->ev_k, dev->stream));
                HIP_TRY(hipEventRecord(b->ev_p, dev->stream));
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
        HIP_TRY(hipStreamSynchronize(dev->stream));
        float ms = 0;
        if (hipEventElapsedTime(&ms, b->ev0, b->ev1) == hipSuccess)
                b->info.last_run_ms = ms;
        b->info.dense_ms = b->info.cand_ms = b->info.fused_ms = b->info.phrase_ms = b->info.rest_ms = b->info.term_planes_ms = b->info.planes_ms = 0;
        if (!b->tasks.empty()) {
                if (hipEventElapsedTime(&ms, b->ev0, b->ev_pl) == hipSuccess)
                        b->info.term_planes_ms = ms; // includes the 256-byte ticket memset that precedes it
                if (hipEventElapsedTime(&ms, b->ev_pl, b->ev_a) == hipSuccess)
                        b->info.dense_ms = ms;
                if (hipEventElapsedTime(&ms, b->ev_a, b->ev_b) == hipSuccess)
                        b->info.cand_ms = ms;
                if (hipEventElapsedTime(&ms, b->ev_b, b->ev_c) == hipSuccess)
                        b->info.fused_ms = ms;
                if (hipEventElapsedTime(&ms, b->ev_c, b->ev_k) == hipSuccess)
                        b->info.planes_ms = ms;
                if (hipEventElapsedTime(&ms, b->ev_k, b->ev_p) == hipSuccess)
                        b->info.phrase_ms = ms;
                if (hipEventElapsedTime(&ms, b->ev_p, b->ev1) == hipSuccess)
                        b->info.rest_ms = ms;
        }
        b->h_counts.resize(b->tasks.size());
        if (!b->tasks.empty())
                HIP_TRY(hipMemcpy(b->h_counts.data(), b->d_counts, b->tasks.size() * 4, hipMemcpyDeviceToHost));
        uint64_t m = 0;
        b->h_query_counts.assign(b->plan.size(), 0);
        uint64_t m_dense = 0, m_fused = 0, out_fused = 0, out_planes = 0;
        for (size_t sidx = 0; sidx < b->plan.size(); ++sidx) {
                const DevQuery &q = b->plan[sidx];
                for (uint32_t t = 0; t < q.ntasks; ++t)
                        b->h_query_counts[sidx] += b->h_counts[q.first_task + t];
                m += b->h_query_counts[sidx];
                if (q.ntasks && b->tasks[q.first_task].kind == TASK_DENSE)
                        m_dense += b->h_query_counts[sidx];
                if (q.ntasks && b->tasks[q.first_task].kind >= TASK_FUSED) {
                        m_fused += b->h_query_counts[sidx]; // (every one-pass kind, k_planes' included)
                        (b->tasks[q.first_task].kind >= TASK_PLANES ? out_planes : out_fused) +=
                                q.out_cap ? 4 * b->h_query_counts[sidx] : 8 * std::min<uint64_t>(b->h_query_counts[sidx], b->topk); // (docIDs of a DocumentsOnly general tree)
                }
        }
        b->info.dense_algorithmic_bytes = b->term_bytes_dense + 4 * m_dense;
        b->info.cand_algorithmic_bytes = (b->term_bytes - b->term_bytes_dense - b->term_bytes_fused - b->term_bytes_planes - b->term_bytes_phrase_hits) + 4 * (m - m_dense - m_fused);
        b->info.planes_algorithmic_bytes = b->term_bytes_planes + out_planes; // docbytes + 8 B x min(matches, K), per query — the lists
                                                                              // the batch's queries share are nevertheless decoded once per launch
        b->info.term_planes_decoded_bytes = b->plane_decoded_bytes;
        b->info.phrase_algorithmic_bytes = b->term_bytes_phrase_hits; // what k_phrase streams by the SURVEY §8(d) count: the hit bytes of the phrases' terms
        b->info.phrase_queries = 0;
        for (const DevQuery &q : b->plan)
                b->info.phrase_queries += q.nphrases != 0;
        b->info.cand_needed_bytes = b->cand_needed_term_bytes ? b->cand_needed_term_bytes + 4 * (m - m_dense - m_fused) : 0;
        b->info.fused_algorithmic_bytes = b->term_bytes_fused + out_fused; // docbytes + 8 B x min(matches, K)
        b->info.matches = m;
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
        if (b->plan[slot].ntasks && !b->plan[slot].out_cap && b->tasks[b->plan[slot].first_task].kind >= TASK_FUSED)
                return fail(TRI_ERR_INVALID, "query %zu ran through the one-pass scored kernel: an AccumulatedScore top-K batch keeps top-K lists and match counts, not docID sets (use topk == 0 or DocumentsOnly)", q);
        if (cap < *n)
                return fail(TRI_ERR_INVALID, "docset needs %zu slots, %zu given", *n, cap);
        HIP_TRY(hipSetDevice(b->ix->dev->device));
        // the docID set is the in-order concatenation of the query's task segments
        const DevQuery &dq = b->plan[slot];
        size_t w = 0;
        for (uint32_t t = 0; t < dq.ntasks; ++t) {
                const uint32_t c = b->h_counts[dq.first_task + t];
                if (!c)
                        continue;
                const uint32_t *src = b->d_out + b->tasks[dq.first_task + t].out_off;
                HIP_TRY(hipMemcpyAsync(out + w, src, (size_t)c * 4, hipMemcpyDeviceToHost, b->ix->dev->stream));
                w += c;
        }
        HIP_TRY(hipStreamSynchronize(b->ix->dev->stream));
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
                hipLaunchKernelGGL(k_hash_docsets, dim3((n + 63) / 64), dim3(64), 0, dev->stream, b->d_plan, b->d_tasks, b->d_counts, n, b->d_out,
                                   b->d_hashes);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipStreamSynchronize(dev->stream));
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
        size_t w = 0;
        for (uint32_t t = 0; t < dq.ntasks; ++t) {
                const uint32_t c = b->h_counts[dq.first_task + t];
                if (!c)
                        continue;
                HIP_TRY(hipMemcpyAsync(out + w, b->d_all_scores + b->tasks[dq.first_task + t].out_off, (size_t)c * 8, hipMemcpyDeviceToHost, b->ix->dev->stream));
                w += c;
        }
        HIP_TRY(hipStreamSynchronize(b->ix->dev->stream));
        return TRI_OK;
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
        // ---- host: the block structure (which block belongs to which term) and input validation
        std::vector<uint32_t> blk_first(nterms + 1, 0), blk_term;
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
                const uint64_t nb = (n + 31) / 32;
                if ((uint64_t)blk_first[t] + nb > 0xfffffff0ull)
                        return fail(TRI_ERR_UNSUPPORTED, "more than 2^32 blocks");
                blk_first[t + 1] = blk_first[t] + (uint32_t)nb;
                blk_term.insert(blk_term.end(), (size_t)nb, (uint32_t)t);
        }
        const uint32_t nblocks = blk_first[nterms];
        struct Bufs {
                uint32_t *docs = nullptr, *freqs = nullptr, *blk_first = nullptr, *blk_term = nullptr, *sizes = nullptr, *tails = nullptr;
                uint16_t *pos = nullptr;
                uint64_t *hit_off = nullptr, *term_first = nullptr, *blk_off = nullptr, *term_off = nullptr, *payloads = nullptr, *scan_sums = nullptr;
                uint64_t scan_cap = 0;
                uint8_t *out = nullptr, *plens = nullptr;
                ~Bufs() {
                        for (void *p : {(void *)docs, (void *)freqs, (void *)blk_first, (void *)blk_term, (void *)sizes, (void *)tails, (void *)pos, (void *)hit_off,
                                        (void *)term_first, (void *)blk_off, (void *)term_off, (void *)out, (void *)payloads, (void *)plens, (void *)scan_sums})
                                hipFree(p);
                }
        } d;
        std::vector<uint64_t> term_off(nterms + 1, 0);
        std::vector<uint64_t> blk_off(nblocks + 1, 0);
        if (nblocks) {
                HIP_TRY(hipMalloc((void **)&d.docs, np * 4));
                HIP_TRY(hipMalloc((void **)&d.freqs, np * 4));
                HIP_TRY(hipMalloc((void **)&d.pos, (nhits + 1) * 2));
                HIP_TRY(hipMalloc((void **)&d.hit_off, (np + 1) * 8));
                HIP_TRY(hipMalloc((void **)&d.term_first, (nterms + 1) * 8));
                HIP_TRY(hipMalloc((void **)&d.blk_first, (nterms + 1) * 4));
                HIP_TRY(hipMalloc((void **)&d.blk_term, (size_t)nblocks * 4));
                HIP_TRY(hipMalloc((void **)&d.sizes, (size_t)nblocks * 4));
                HIP_TRY(hipMalloc((void **)&d.tails, (size_t)nblocks * 4));
                HIP_TRY(hipMalloc((void **)&d.blk_off, ((size_t)nblocks + 1) * 8));
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
                HIP_TRY(hipMemcpyAsync(d.term_first, term_first, (nterms + 1) * 8, hipMemcpyHostToDevice, dev->stream));
                HIP_TRY(hipMemcpyAsync(d.blk_first, blk_first.data(), (nterms + 1) * 4, hipMemcpyHostToDevice, dev->stream));
                HIP_TRY(hipMemcpyAsync(d.blk_term, blk_term.data(), (size_t)nblocks * 4, hipMemcpyHostToDevice, dev->stream));
                // hits before every posting, then the blocks' sizes and their running sum
                // (exclusive scans over the whole device: chunk sums, chunk bases, chunks — k_encode.hpp)
                auto scan = [&](const uint32_t *in, uint64_t *outp, const uint64_t n) -> int {
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
                };
                int rcs;
                if ((rcs = scan(d.freqs, d.hit_off, np)))
                        return rcs;
                const EncArgs a{d.docs, d.freqs, d.pos, d.plens, d.payloads, d.hit_off, d.term_first, d.blk_first, d.blk_term, nblocks};
                hipLaunchKernelGGL(k_enc_size, dim3((nblocks + 255) / 256), dim3(256), 0, dev->stream, a, d.sizes, d.tails);
                if ((rcs = scan(d.sizes, d.blk_off, (uint64_t)nblocks)))
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
