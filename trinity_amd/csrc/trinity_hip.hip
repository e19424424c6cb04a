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
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <unistd.h>
#include <memory>
#include <numeric>
#include <string>
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
                        return fail(TRI_ERR_DEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
        } while (0)

extern "C" const char *tri_last_error(void) { return g_err; }
extern "C" int tri_abi_version(void) { return TRI_ABI_VERSION; }

// ------------------------------------------------------------------------------------------ structs
struct DevTerm {
        uint32_t documents;
        uint32_t first_block;
        uint32_t nblocks;
        uint32_t last_n;  // docs in the final block (1..32)
        uint32_t win_off; // lists of >= WIN_MIN_BLOCKS blocks: row in win[] (first block with last >= w * SPAN_BITS, per window w); else ~0
        uint32_t pad[3];
};
constexpr uint32_t WIN_MIN_BLOCKS = 128;

// A query in conjunctive normal form: AND of groups, a group = one term or an OR of terms.  qterms[] lists the terms
// group by group, cheapest group first (exec.cpp:35-110 cost model); bit 31 marks the first term of a group.
// Root OR of terms == a single group.  (Conjuction / DisjunctionAllPLI semantics, docset_iterators.cpp:226-405.)
constexpr uint32_t QT_GROUP = 0x80000000u;
constexpr uint32_t MAX_QTERMS = 16;
struct DevQuery {
        uint32_t nterms;    // total terms over all groups (<= MAX_QTERMS)
        uint32_t term_base; // into qterms[]
        uint64_t out_off;   // docID slots
        uint32_t out_cap;
        uint32_t qid; // caller's query index
        uint32_t first_task, ntasks;
        uint32_t score_base, nscore; // AccumulatedScoreScheme: sterms[]/sweights[] slice, reference summation order
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
constexpr uint32_t TASK_CAND = 0;  // candidate tiles of the lead list, filtered by galloping / block-driven merge
constexpr uint32_t TASK_DENSE = 1; // bitmap algebra over fixed docID windows (every list dense enough)

struct tri_dev {
        int device;
        hipStream_t stream;
        hipEvent_t ev0, ev1;
        int cus;
};

struct tri_index {
        tri_dev *dev;
        uint8_t *d_index = nullptr;
        uint32_t *d_blk_last = nullptr, *d_blk_off = nullptr, *d_win = nullptr;
        uint32_t nwin = 0; // windows per win[] row (+1 sentinel column)
        DevTerm *d_terms = nullptr;
        std::vector<DevTerm> terms;
        std::vector<uint32_t> h_blk_last; // host copy of the directory's last-docID column (planner: task output offsets)
        std::vector<tri_term> tctx;
        std::vector<uint64_t> docbytes, hitbytes;
        tri_index_info info{};
};

struct tri_batch {
        tri_index *ix;
        uint32_t flags, topk;
        size_t nq;
        std::vector<DevQuery> plan; // execution order (cost descending)
        std::vector<uint32_t> qterms;
        std::vector<uint32_t> slot_of_query; // caller query -> plan slot (UINT32_MAX: trivially empty)
        DevQuery *d_plan = nullptr;
        std::vector<DevTask> tasks; // scheduling order (cost descending)
        DevTask *d_tasks = nullptr;
        uint32_t *d_sched = nullptr; // task indices, heaviest first
        uint32_t *d_qterms = nullptr;
        uint32_t *d_out = nullptr;
        uint32_t *d_counts = nullptr; // per task, indexed first_task + i in query order
        uint32_t *d_ticket = nullptr;
        uint64_t *d_hashes = nullptr;
        // AccumulatedScoreScheme
        std::vector<uint32_t> sterms;
        std::vector<double> sweights;
        uint32_t *d_sterms = nullptr;
        double *d_sweights = nullptr;
        uint32_t *d_part_docs = nullptr, *d_part_counts = nullptr, *d_top_docs = nullptr, *d_top_counts = nullptr;
        double *d_part_scores = nullptr;
        float *d_top_scores = nullptr;
        double *d_all_scores = nullptr; // topk == 0: one double per out[] slot
        uint64_t out_capacity = 0;
        uint64_t term_bytes = 0; // sum of docbytes over all query terms
        std::vector<uint32_t> h_counts;       // per task
        std::vector<uint64_t> h_query_counts; // per plan slot
        bool synced = false;
        tri_batch_info info{};
};

// ------------------------------------------------------------------------------------------ debug trace
// -DTRI_TRACE builds write per-workgroup progress markers into host-pinned memory; tri_batch_sync then polls
// with a watchdog (env TRINITY_WATCHDOG_S) and dumps the markers instead of hanging.  Not in product builds.
#ifdef TRI_TRACE
static uint32_t *g_trace_host = nullptr;
__device__ volatile uint32_t *g_trace = nullptr;
#ifndef TRI_TRACE_MASK
#define TRI_TRACE_MASK 0xffffffffu
#endif
#define TRACE(stage, a, b)                                                      \
        do {                                                                    \
                if (((TRI_TRACE_MASK >> (stage)) & 1u) && threadIdx.x == 0 && g_trace) {                              \
                        volatile uint32_t *t_ = g_trace + (blockIdx.x & 63) * 4; \
                        t_[0] = (stage);                                        \
                        t_[1] = (a);                                            \
                        t_[2] = (b);                                            \
                        t_[3] = t_[3] + 1;                                      \
                        __threadfence_system();                                 \
                }                                                               \
        } while (0)
#else
#define TRACE(stage, a, b) \
        do {               \
        } while (0)
#endif

// ------------------------------------------------------------------------------------------ device: varint
// Prefix varint of Switch/switch_compiler_aux.h:53-80, branch-free.  `w` holds the next >= 5 stream bytes,
// least-significant byte first.
__device__ __forceinline__ uint32_t vb_decode(uint64_t w, uint32_t &len) {
        const uint32_t lo32 = (uint32_t)w;
        const uint32_t b0 = lo32 & 0xffu;
        const uint32_t ones = __clz(~(lo32 << 24)); // leading 1-bits of b0 (0..8)
        const uint32_t n = ones < 4u ? ones : 4u;
        const uint32_t be = __builtin_bswap32(lo32); // b0 b1 b2 b3
        const uint32_t v1 = b0;
        const uint32_t v2 = (be >> 16) & 0x3fffu;
        const uint32_t v3 = ((b0 & 0x1fu) << 16) | ((lo32 >> 8) & 0xffffu);
        const uint32_t v4 = be & 0x0fffffffu;
        const uint32_t v5 = (uint32_t)(w >> 8);
        len = n + 1;
        uint32_t v = v1;
        v = n == 1 ? v2 : v;
        v = n == 2 ? v3 : v;
        v = n == 3 ? v4 : v;
        v = n == 4 ? v5 : v;
        return v;
}

// Per-lane byte stream over global memory: a 16-byte register window (lo = next 8 bytes, hi = the following
// ones) refilled from aligned 8-byte loads, with three further qwords always in flight so that the load
// latency sits behind ~24 bytes of decoding (index[] carries >= 64 bytes of slack past the last chunk).
struct VbStream {
        const uint64_t *q;
        uint64_t lo, hi, n1, n2, n3;
        int valid;

        __device__ __forceinline__ void init(const uint8_t *p) {
                const uintptr_t a = (uintptr_t)p;
                const uint32_t sk = (uint32_t)(a & 7u);
                q = (const uint64_t *)(a & ~(uintptr_t)7);
                const uint64_t w0 = q[0], w1 = q[1];
                n1 = q[2];
                n2 = q[3];
                n3 = q[4];
                q += 5;
                const uint32_t sh = sk * 8;
                lo = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
                hi = sh ? (w1 >> sh) : w1;
                valid = 16 - (int)sk;
        }
        __device__ __forceinline__ void refill() {
                if (valid <= 8) {
                        const uint64_t w = n1;
                        n1 = n2;
                        n2 = n3;
                        n3 = *q++;
                        const uint32_t sh = (uint32_t)valid * 8; // 0..64
                        if (valid == 8)
                                hi = w;
                        else if (valid == 0) {
                                lo = w;
                                hi = 0;
                        } else {
                                lo |= w << sh;
                                hi = w >> (64 - sh);
                        }
                        valid += 8;
                }
        }
        __device__ __forceinline__ uint32_t next() {
                refill();
                uint32_t len;
                const uint32_t v = vb_decode(lo, len);
                const uint32_t s = len * 8;
                lo = (lo >> s) | (hi << (64 - s));
                hi >>= s;
                valid -= (int)len;
                return v;
        }
        // after refill(): true when the next k (1..8) bytes are k one-byte varints (values < 128)
        __device__ __forceinline__ bool small_run(const uint32_t k) const { return (lo & (0x8080808080808080ull >> (8u * (8u - k)))) == 0; }
        // consume k (1..8) bytes, returning the window they were in (byte j = j-th value)
        __device__ __forceinline__ uint64_t take(const uint32_t k) {
                const uint64_t w = lo;
                if (k == 8) {
                        lo = hi;
                        hi = 0;
                } else {
                        const uint32_t s = k * 8;
                        lo = (lo >> s) | (hi << (64 - s));
                        hi >>= s;
                }
                valid -= (int)k;
                return w;
        }
};

// ------------------------------------------------------------------------------------------ k_decode_terms
struct DecodeJob {
        uint32_t term;
        uint32_t pad;
        uint64_t out_off;
};

// grid.x covers blocks of job blockIdx.y in chunks of 256; one lane per block (google_codec.cpp:596-639)
__global__ __launch_bounds__(256) void k_decode_terms(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last,
                                                      const uint32_t *__restrict__ blk_off, const DevTerm *__restrict__ terms,
                                                      const DecodeJob *__restrict__ jobs, uint32_t *__restrict__ docs,
                                                      uint32_t *__restrict__ freqs) {
        const DecodeJob job = jobs[blockIdx.y];
        const DevTerm t = terms[job.term];
        for (uint32_t b = blockIdx.x * 256 + threadIdx.x; b < t.nblocks; b += gridDim.x * 256) {
                const uint32_t gb = t.first_block + b;
                const uint32_t off = blk_off[gb];
                const uint32_t n = index[off - 1];
                const uint32_t last = blk_last[gb];
                uint32_t doc = b ? blk_last[gb - 1] : 0;
                VbStream s;
                s.init(index + off);
                uint32_t *od = docs + job.out_off + (uint64_t)b * 32;
                for (uint32_t i = 0; i + 1 < n; ++i) {
                        doc += s.next();
                        od[i] = doc;
                }
                od[n - 1] = last;
                if (freqs) {
                        uint32_t *of = freqs + job.out_off + (uint64_t)b * 32;
                        for (uint32_t i = 0; i < n; ++i)
                                of[i] = s.next();
                }
        }
}

// ------------------------------------------------------------------------------------------ k_and
constexpr int AND_WG = 256;
constexpr int TILE_BLOCKS = 256;
constexpr int TILE_CANDS = TILE_BLOCKS * 32;

// Values that are workgroup-uniform by construction but read back from LDS look divergent to the compiler; a
// loop whose exit depends on one gets exec-masked structurisation, which is fatal around s_barrier (lanes
// "leave" the loop at different times).  uni() pins such values into an SGPR so the branch is scalar.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// LDS candidate layout: logical slot j lives at phys(j); rotating each 32-slot row by its row number keeps
// the one-lane-per-row writes of the lead decode (lane t writes row t, column i) off a single bank.
__device__ __forceinline__ uint32_t phys(uint32_t j) { return (j & ~31u) | ((j + (j >> 5)) & 31u); }

__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total) {
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(x, d, 64);
                if ((int)(threadIdx.x & 63) >= d)
                        x += y;
        }
        total = __shfl(x, 63, 64);
        return x - v;
}

#ifndef TRI_DENSE_V
#define TRI_DENSE_V 1
#endif
constexpr uint32_t SPAN_BITS = 1u << 17; // docIDs per dense window
constexpr uint32_t SPAN_WORDS = SPAN_BITS / 32;

struct AndShared {
        union {
                struct {
                        uint32_t cand[TILE_CANDS];
                        uint32_t hit[TILE_BLOCKS]; // bit k of hit[r] <=> logical candidate r*32+k matched
                        uint32_t blkof[AND_WG + 1];
                };
                uint32_t bits[2][SPAN_WORDS + 1]; // TASK_DENSE: two docID-window bitmaps (candidates / survivors), +1 sink word
        };
        uint32_t tbase[AND_WG];
        uint32_t scan[8];
        uint32_t bcast[4];
        uint32_t lcur[16]; // per term: directory cursor, uniform across the workgroup
};

// Workgroup-cooperative lower bound over a sorted global array: first i in [0, n) with a[i] >= key, else n.
// 256-ary search: every lane probes the end of its segment, one ballot per wave finds the first segment whose
// last element is >= key; ~log256(n) rounds of one (L2-resident) load each instead of log2(n) dependent loads.
__device__ uint32_t wg_lower_bound(AndShared &sh, const uint32_t *__restrict__ a, const uint32_t n, const uint32_t key) {
        const uint32_t tid = threadIdx.x;
        uint32_t lo = 0, hi = n; // answer in [lo, hi]
        while (hi > lo) {
                const uint32_t len = hi - lo;
                const uint32_t step = (len + AND_WG - 1) / AND_WG;
                const uint32_t pos = lo + (tid + 1) * step - 1;
                const bool ge = pos >= hi ? true : a[pos] >= key;
                const uint64_t m = __ballot(ge);
                sh.scan[tid >> 6] = m ? (tid & ~63u) + (uint32_t)__builtin_ctzll(m) : 0xffffffffu;
                __syncthreads();
                const uint32_t first = uni(min(min(sh.scan[0], sh.scan[1]), min(sh.scan[2], sh.scan[3])));
                __syncthreads();
                const uint32_t nlo = lo + first * step;
                const uint32_t nhi = min(hi, lo + (first + 1) * step - 1);
                lo = nlo;
                hi = step == 1 ? nlo : nhi;
        }
        return lo;
}

// Filter the C candidates in sh.cand (logical order ascending) against term `t`: sets sh.hit bits.
// Caller syncs before and after.
__device__ void and_filter_tile(AndShared &sh, const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last,
                                const uint32_t *__restrict__ blk_off, const DevTerm t, const uint32_t C, const uint32_t lcur_slot,
                                const bool block_driven) {
        const uint32_t tid = threadIdx.x;
        const uint32_t *bl = blk_last + t.first_block;
        const uint32_t *bo = blk_off + t.first_block;
        const uint32_t cmin = sh.cand[phys(0)], cmax = sh.cand[phys(C - 1)];

        if (block_driven) {
                // advance lcur to the first block whose last docID >= cmin (tiles arrive in ascending docID order)
                uint32_t lcur = uni(sh.lcur[lcur_slot]);
                if (lcur == 0xffffffffu) // first tile of this task: position by cooperative search, then gallop forward
                        lcur = wg_lower_bound(sh, bl, t.nblocks, cmin);
                for (;;) {
                        const uint32_t b = lcur + tid;
                        const bool below = b < t.nblocks && bl[b] < cmin;
                        const uint64_t m = __ballot(below);
                        // number of leading lanes (from lane 0) with below == true, per wave
                        const uint32_t lead = m == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~m);
                        sh.scan[tid >> 6] = lead; // wave-uniform value, every lane stores it: no divergent branch
                        __syncthreads();
                        uint32_t adv = 0;
                        for (int w = 0; w < AND_WG / 64; ++w) {
                                adv += sh.scan[w];
                                if (sh.scan[w] != 64)
                                        break;
                        }
                        adv = uni(adv);
                        __syncthreads();
                        lcur += adv;
                        if (adv != AND_WG || lcur >= t.nblocks)
                                break;
                }
                sh.lcur[lcur_slot] = lcur; // uniform value, branch-free store
                TRACE(10, lcur, t.nblocks);
                for (uint32_t cb = lcur; cb < t.nblocks; cb += AND_WG) {
                        TRACE(11, cb, t.nblocks);
                        const uint32_t b = cb + tid;
                        bool beyond = true;
                        if (b < t.nblocks) {
                                const uint32_t prev = b ? bl[b - 1] : 0; // docs of block b lie in (prev, last]
                                const uint32_t last = bl[b];
                                beyond = last >= cmax;
                                if (prev < cmax) {
                                        // first candidate > prev
                                        uint32_t lo = 0, hi = C;
                                        while (lo < hi) {
                                                const uint32_t mid = (lo + hi) >> 1;
                                                if (sh.cand[phys(mid)] <= prev)
                                                        lo = mid + 1;
                                                else
                                                        hi = mid;
                                        }
                                        uint32_t ptr = lo;
                                        uint32_t cv = ptr < C ? sh.cand[phys(ptr)] : 0xffffffffu;
                                        if (cv <= last) {
                                                const uint32_t off = bo[b];
                                                const uint32_t n = index[off - 1];
                                                VbStream s;
                                                s.init(index + off);
                                                uint32_t doc = prev;
                                                for (uint32_t i = 0; i < n; ++i) {
                                                        doc = (i + 1 < n) ? doc + s.next() : last;
                                                        while (cv < doc) {
                                                                ++ptr;
                                                                cv = ptr < C ? sh.cand[phys(ptr)] : 0xffffffffu;
                                                        }
                                                        if (cv == doc)
                                                                atomicOr(&sh.hit[ptr >> 5], 1u << (ptr & 31));
                                                        if (cv > last)
                                                                break;
                                                }
                                        }
                                }
                        }
                        // workgroup-wide OR of `beyond`, branch-free: one ballot per wave, four LDS words
                        sh.scan[4 + (tid >> 6)] = __ballot(beyond) != 0ull;
                        __syncthreads();
                        const uint32_t any_beyond = uni(sh.scan[4] | sh.scan[5] | sh.scan[6] | sh.scan[7]);
                        __syncthreads();
                        if (any_beyond)
                                break;
                }
        } else {
                // candidate-driven galloping: each candidate finds its block in the directory; the first
                // candidate of each run that maps to the same block decodes it and merges forward
                sh.blkof[0] = 0xffffffffu;
                __syncthreads();
                for (uint32_t base = 0; base < C; base += AND_WG) {
                        TRACE(20, base, C);
                        const uint32_t j = base + tid;
                        uint32_t bj = 0xffffffffu;
                        uint32_t cv = 0;
                        if (j < C) {
                                cv = sh.cand[phys(j)];
                                uint32_t lo = 0, hi = t.nblocks; // first block with last >= cv
                                while (lo < hi) {
                                        const uint32_t mid = (lo + hi) >> 1;
                                        if (bl[mid] < cv)
                                                lo = mid + 1;
                                        else
                                                hi = mid;
                                }
                                bj = lo; // == nblocks: beyond the list
                        }
                        sh.blkof[tid + 1] = bj;
                        __syncthreads();
                        const uint32_t prevb = sh.blkof[tid];
                        __syncthreads();
                        {
                                // carry the last lane's block into the next round (slot 0), branch-free:
                                // lanes of the last wave all store lane 63's value, other waves rewrite their own slot
                                const uint32_t lastb = __shfl(bj, 63, 64);
                                const bool lastwave = (tid >> 6) == (AND_WG / 64 - 1);
                                sh.blkof[lastwave ? 0 : tid + 1] = lastwave ? lastb : bj;
                        }
                        if (j < C && bj < t.nblocks && bj != prevb) {
                                const uint32_t prev = bj ? bl[bj - 1] : 0;
                                const uint32_t last = bl[bj];
                                const uint32_t off = bo[bj];
                                const uint32_t n = index[off - 1];
                                VbStream s;
                                s.init(index + off);
                                uint32_t doc = prev;
                                uint32_t ptr = j;
                                for (uint32_t i = 0; i < n; ++i) {
                                        doc = (i + 1 < n) ? doc + s.next() : last;
                                        while (cv < doc) {
                                                ++ptr;
                                                cv = ptr < C ? sh.cand[phys(ptr)] : 0xffffffffu;
                                        }
                                        if (cv == doc)
                                                atomicOr(&sh.hit[ptr >> 5], 1u << (ptr & 31));
                                        if (cv > last)
                                                break;
                                }
                        }
                        __syncthreads();
                }
        }
}

// ---- TASK_DENSE: bitmap algebra over docID windows -------------------------------------------------------
// One lane decodes one block (unpack_block, google_codec.cpp:596-639) and ORs its documents into / tests them
// against a window bitmap in LDS.  Consecutive documents of a dense list fall into the same 32-bit word, so the
// lane keeps the current word in registers and touches LDS once per word, not once per posting.
// Bitmap word -> LDS slot.  Neighbouring lanes decode neighbouring blocks, i.e. words a small constant stride apart,
// which lands lanes l and l+16 on one bank; XOR-ing in the next five index bits spreads every 32-word row differently.
// A bijection inside each 1024-word group; the sink word (index SPAN_WORDS) maps to itself.
#if TRI_DENSE_V == 1
__device__ __forceinline__ uint32_t bswz(const uint32_t w) { return w ^ ((w >> 5) & 31u); }
#else
__device__ __forceinline__ uint32_t bswz(const uint32_t w) { return w; }
#endif

template <bool FIRST>
__device__ __forceinline__ void dense_block(const uint8_t *__restrict__ index, const uint32_t off, const uint32_t n, const uint32_t prev,
                                            const uint32_t last, const uint32_t w0, const uint32_t *src, uint32_t *dst) {
        VbStream s;
        s.init(index + off);
        uint32_t doc = prev;
#if TRI_DENSE_V == 0
        uint32_t curword = 0xffffffffu, cw = 0, acc = 0;
        auto visit = [&](const uint32_t d) {
                const uint32_t rel = d - w0; // documents outside the window land on word >= SPAN_WORDS
                const uint32_t word = rel >> 5;
                if (word != curword) {
                        if (acc)
                                atomicOr(&dst[curword], acc);
                        acc = 0;
                        curword = word;
                        cw = word < SPAN_WORDS ? (FIRST ? 0xffffffffu : src[word]) : 0u;
                }
                acc |= cw & (1u << (rel & 31u));
        };
#else
        // branch-free: one LDS OR (plus one LDS read when testing) per posting; documents outside the window go to the
        // sink word.  No lane-divergent control flow inside the 8-posting fast path, so the reads pipeline.
        auto visit = [&](const uint32_t d) {
                const uint32_t rel = d - w0;
                const uint32_t word = bswz(min(rel >> 5, SPAN_WORDS));
                const uint32_t bit = 1u << (rel & 31u);
                if (FIRST)
                        atomicOr(&dst[word], bit);
                else
                        atomicOr(&dst[word], src[word] & bit);
        };
#endif
        const uint32_t nd = n - 1;
        uint32_t i = 0;
        while (i < nd) {
                s.refill();
                const uint32_t k = min(8u, nd - i);
                if (s.small_run(k)) { // k one-byte deltas (the rule for head terms): no per-value length decode
                        uint64_t w = s.take(k);
#pragma unroll
                        for (uint32_t j = 0; j < 8; ++j) {
                                if (j < k) {
                                        doc += (uint32_t)(w & 0xffu);
                                        w >>= 8;
                                        visit(doc);
                                }
                        }
                        i += k;
                } else {
                        doc += s.next();
                        visit(doc);
                        ++i;
                }
        }
        visit(last);
#if TRI_DENSE_V == 0
        if (acc)
                atomicOr(&dst[curword], acc);
#endif
}

__device__ void dense_task(AndShared &sh, const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last,
                           const uint32_t *__restrict__ blk_off, const uint32_t *__restrict__ win, const DevTerm *__restrict__ terms,
                           const uint32_t *__restrict__ qterms, const DevQuery q, const DevTask task, uint32_t *__restrict__ out,
                           uint32_t *__restrict__ count_out) {
        const uint32_t tid = threadIdx.x;
        uint32_t *qout = out + task.out_off;
        uint32_t produced = 0;
        sh.lcur[tid & 15] = 0; // per term: a block index at or before the first block that can matter
        __syncthreads();
        // number of terms in the lead group (it creates the candidates; the other groups test them)
        uint32_t nlead = 1;
        while (nlead < q.nterms && !(qterms[q.term_base + nlead] & QT_GROUP))
                ++nlead;
        bool done = false; // uniform
        for (uint32_t w = task.tile_begin; w < task.tile_end && !done;) {
                // ---- skip windows no lead-group list reaches: position the lead cursors at w, look at the first
                //      document each may hold at or after it
                uint32_t wnext = 0xffffffffu;
                for (uint32_t k = 0; k < nlead; ++k) {
                        const DevTerm t = terms[qterms[q.term_base + k] & ~QT_GROUP];
                        const uint32_t *bl = blk_last + t.first_block;
                        uint32_t cur;
                        if (t.win_off != 0xffffffffu)
                                cur = win[t.win_off + w]; // indexed list: first block with last >= w * SPAN_BITS
                        else {
                                cur = uni(sh.lcur[k]);
                                if (cur < t.nblocks && bl[cur] < w * SPAN_BITS)
                                        cur += wg_lower_bound(sh, bl + cur, t.nblocks - cur, w * SPAN_BITS);
                                __syncthreads();
                                sh.lcur[k] = cur;
                        }
                        if (cur < t.nblocks) {
                                const uint32_t first_possible = cur ? bl[cur - 1] + 1 : 1;
                                wnext = min(wnext, max(w, first_possible / SPAN_BITS));
                        }
                }
                __syncthreads();
                if (wnext >= task.tile_end)
                        break; // the lead group holds nothing more in this task's range
                w = wnext;
                const uint32_t w0 = w * SPAN_BITS;
                const uint32_t wlast = w0 + (SPAN_BITS - 1);
                uint32_t gi = 0;       // group index
                bool galive = false;   // some term of the current group reaches this window or beyond
                for (uint32_t k = 0; k < q.nterms; ++k) {
                        const uint32_t tt = qterms[q.term_base + k];
                        const DevTerm t = terms[tt & ~QT_GROUP];
                        const uint32_t *bl = blk_last + t.first_block;
                        const uint32_t *bo = blk_off + t.first_block;
                        if (k && (tt & QT_GROUP)) {
                                if (!galive) { // an exhausted conjunct: no further match anywhere
                                        done = true;
                                        break;
                                }
                                ++gi;
                                galive = false;
                        }
                        uint32_t *dst = sh.bits[gi & 1];
                        const uint32_t *src = sh.bits[(gi & 1) ^ 1];
                        if (tt & QT_GROUP)
                                for (uint32_t i = tid; i < SPAN_WORDS; i += AND_WG)
                                        dst[i] = 0;
                        // blocks that can hold documents of [w0, wlast]: first block with last >= w0 ... first with last >= wlast
                        uint32_t b_lo, b_hi;
                        if (t.win_off != 0xffffffffu) {
                                // indexed list: two scalar loads replace both directory searches (win[w + 1] is the first block
                                // with last >= the next window's first docID; it may still hold documents of this window)
                                b_lo = win[t.win_off + w];
                                b_hi = min(win[t.win_off + w + 1], t.nblocks - 1);
                                if (b_lo < t.nblocks)
                                        galive = true;
                                __syncthreads(); // dst cleared, earlier passes complete
                        } else {
                                b_lo = uni(sh.lcur[k]);
                                if (b_lo < t.nblocks && bl[b_lo] < w0)
                                        b_lo += wg_lower_bound(sh, bl + b_lo, t.nblocks - b_lo, w0);
                                b_hi = b_lo;
                                if (b_lo < t.nblocks) {
                                        galive = true;
                                        b_hi = b_lo + wg_lower_bound(sh, bl + b_lo, t.nblocks - b_lo, wlast);
                                        if (b_hi >= t.nblocks)
                                                b_hi = t.nblocks - 1;
                                }
                                __syncthreads(); // dst cleared, earlier passes complete, cursor reads done
                                sh.lcur[k] = b_lo < t.nblocks ? b_hi : b_lo;
                        }
                        if (b_lo < t.nblocks) {
                                for (uint32_t cb = b_lo; cb <= b_hi; cb += AND_WG) {
                                        const uint32_t b = cb + tid;
                                        if (b <= b_hi) {
                                                const uint32_t prev = b ? bl[b - 1] : 0;
                                                const uint32_t last = bl[b];
                                                const uint32_t off = bo[b];
                                                const uint32_t n = index[off - 1];
                                                if (gi == 0)
                                                        dense_block<true>(index, off, n, prev, last, w0, src, dst);
                                                else
                                                        dense_block<false>(index, off, n, prev, last, w0, src, dst);
                                        }
                                }
                        }
                        __syncthreads();
                }
                if (done || !galive) {
                        done = true;
                        break;
                }
                // ---- expand the survivors bitmap into ascending docIDs
                const uint32_t *fin = sh.bits[gi & 1];
                uint32_t *pre = sh.bits[(gi & 1) ^ 1]; // the other bitmap is dead: per-word exclusive prefix
                {
                        uint32_t run = 0;
                        for (uint32_t j = 0; j < SPAN_WORDS / AND_WG; ++j) {
                                const uint32_t wi = tid * (SPAN_WORDS / AND_WG) + j;
                                pre[wi] = run;
                                run += __popc(fin[bswz(wi)]);
                        }
                        uint32_t wtot;
                        const uint32_t ex = wave_excl_scan(run, wtot);
                        sh.scan[tid >> 6] = wtot;
                        __syncthreads();
                        uint32_t wbase = 0, total = 0;
                        for (int wv = 0; wv < AND_WG / 64; ++wv) {
                                if (wv < (int)(tid >> 6))
                                        wbase += sh.scan[wv];
                                total += sh.scan[wv];
                        }
                        sh.tbase[tid] = ex + wbase;
                        __syncthreads();
                        // word-strided sweep: neighbouring lanes own neighbouring words, so a wave's stores stay together
                        for (uint32_t wi = tid; wi < SPAN_WORDS; wi += AND_WG) {
                                uint32_t m = fin[bswz(wi)];
                                uint32_t o = produced + sh.tbase[wi / (SPAN_WORDS / AND_WG)] + pre[wi];
                                const uint32_t base = w0 + wi * 32;
                                while (m) {
                                        qout[o++] = base + (uint32_t)__builtin_ctz(m);
                                        m &= m - 1;
                                }
                        }
                        produced += uni(total);
                        __syncthreads();
                }
                ++w;
        }
        __syncthreads();
        if (uni(tid >> 6) == 0)
                *count_out = produced;
}

__global__ __launch_bounds__(AND_WG) void k_and(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last,
                                                const uint32_t *__restrict__ blk_off, const uint32_t *__restrict__ win,
                                                const DevTerm *__restrict__ terms,
                                                const DevQuery *__restrict__ plan, const DevTask *__restrict__ tasks,
                                                const uint32_t *__restrict__ sched, const uint32_t *__restrict__ qterms,
                                                const uint32_t ntasks, uint32_t *__restrict__ ticket,
                                                uint32_t *__restrict__ out, uint32_t *__restrict__ counts) {
        __shared__ AndShared sh;
        const uint32_t tid = threadIdx.x;
        const uint32_t wave = uni(tid >> 6);
        for (;;) {
                // next query: wave 0 draws the ticket.  All 64 lanes add 1 (the compiler folds that into ONE
                // global atomic of +64 with a uniform operand — no lane-divergent branch at the loop head), so the
                // counter advances in units of 64 per draw.
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
                const DevTask task = tasks[tix];
                const uint32_t slot = task.slot;
                const DevQuery q = plan[slot];
                if (task.kind == TASK_DENSE) {
                        dense_task(sh, index, blk_last, blk_off, win, terms, qterms, q, task, out, counts + tix);
                        continue;
                }
                const DevTerm lead = terms[qterms[q.term_base] & ~QT_GROUP];
                TRACE(1, slot, q.nterms);
                uint32_t *qout = out + task.out_off;
                uint32_t produced = 0;
                sh.lcur[tid & 15] = 0xffffffffu; // "not positioned yet"
                const uint32_t tb_end = min(lead.nblocks, task.tile_end * TILE_BLOCKS);

                for (uint32_t tb = task.tile_begin * TILE_BLOCKS; tb < tb_end; tb += TILE_BLOCKS) {
                        const uint32_t nb = min((uint32_t)TILE_BLOCKS, lead.nblocks - tb);
                        uint32_t C = (tb + nb == lead.nblocks) ? (nb - 1) * 32 + lead.last_n : nb * 32;
                        // ---- decode the lead tile: one lane per block (unpack_block, google_codec.cpp:596-639)
                        if (tid < nb) {
                                const uint32_t b = tb + tid, gb = lead.first_block + b;
                                const uint32_t off = blk_off[gb];
                                const uint32_t n = index[off - 1];
                                const uint32_t last = blk_last[gb];
                                uint32_t doc = b ? blk_last[gb - 1] : 0;
                                VbStream s;
                                s.init(index + off);
                                const uint32_t row = tid * 32;
                                for (uint32_t i = 0; i + 1 < n; ++i) {
                                        doc += s.next();
                                        sh.cand[row | ((i + tid) & 31u)] = doc;
                                }
                                sh.cand[row | ((n - 1 + tid) & 31u)] = last;
                        }
                        __syncthreads();
                        TRACE(2, slot, tb);

                        // ---- every other group filters the surviving candidates: a candidate survives a group when any
                        //      of the group's terms holds it (hit bits are OR-ed across the group's terms)
                        for (uint32_t k = 1; k < q.nterms && C; ++k) {
                                const uint32_t tt = qterms[q.term_base + k];
                                const DevTerm t = terms[tt & ~QT_GROUP];
                                if (tt & QT_GROUP)
                                        sh.hit[tid] = 0;
                                __syncthreads();
#if defined(TRI_FORCE_CAND)
                                const bool bd = false;
#elif defined(TRI_FORCE_BLOCK)
                                const bool bd = true;
#else
                                const bool bd = t.nblocks <= lead.documents;
#endif
                                TRACE(3, slot, (k << 16) | (bd ? 1 : 0));
                                and_filter_tile(sh, index, blk_last, blk_off, t, C, k - 1, bd);
                                TRACE(4, slot, C);
                                __syncthreads();
                                const bool lastterm = k + 1 == q.nterms;
                                if (!lastterm && !(qterms[q.term_base + k + 1] & QT_GROUP))
                                        continue; // more terms of this OR group to come
                                // compact survivors (stable => still ascending)
                                const uint32_t bits = sh.hit[tid];
                                const uint32_t cnt = __popc(bits);
                                uint32_t wtot;
                                uint32_t ex = wave_excl_scan(cnt, wtot);
                                sh.scan[tid >> 6] = wtot; // wave-uniform
                                __syncthreads();
                                uint32_t wbase = 0, total = 0;
                                for (int w = 0; w < AND_WG / 64; ++w) {
                                        if (w < (int)(tid >> 6))
                                                wbase += sh.scan[w];
                                        total += sh.scan[w];
                                }
                                ex += wbase;
                                if (lastterm) {
                                        // last group: survivors go straight to the result, ascending
                                        uint32_t m = bits, o = produced + ex;
                                        while (m) {
                                                const uint32_t kbit = __builtin_ctz(m);
                                                m &= m - 1;
                                                qout[o++] = sh.cand[phys(tid * 32 + kbit)];
                                        }
                                } else {
                                        // in-place compaction: every lane lifts its row into registers first
                                        uint32_t vals[32];
#pragma unroll
                                        for (int kk = 0; kk < 32; ++kk)
                                                vals[kk] = sh.cand[(tid * 32) | ((kk + tid) & 31u)];
                                        __syncthreads();
                                        uint32_t o = ex;
#pragma unroll
                                        for (int kk = 0; kk < 32; ++kk)
                                                if ((bits >> kk) & 1u) {
                                                        sh.cand[phys(o)] = vals[kk];
                                                        ++o;
                                                }
                                }
                                C = uni(total);
                                __syncthreads();
                        }
                        if (q.nterms == 1) {
                                for (uint32_t j = tid; j < C; j += AND_WG)
                                        qout[produced + j] = sh.cand[phys(j)];
                        }
                        produced += C;
                        __syncthreads();
                }
                if (wave == 0)
                        counts[tix] = produced; // scalar branch; the wave's lanes store one identical dword
                TRACE(5, slot, produced);
        }
        TRACE(6, 0, 0);
}

// ------------------------------------------------------------------------------------------ k_score / k_topk_merge
// AccumulatedScoreScheme (exec.h:36-41).  k_and has produced every query's ascending match list; k_score walks each
// task's segment in tiles of 4096 matches, and for every scoring term looks the matches up again through the
// directory (each match binary-searches its block, the first match of a block decodes it once: deltas to locate the
// matching positions, then the freqs), adding  float(idf * float(f) / double(f + 1.2f))  to a per-match double in LDS —
// IndexSourcesCollectionBM25Scorer::Scorer::score (similarity.h:228-235) summed in iterator order by the Conjuction
// wrapper (docset_iterators_scorers.cpp:173-193).  The tile is then offered to the task's top-K (score descending,
// docID ascending: the application-side MatchedIndexDocumentsFilter heap, matches.h:155-171).  k_topk_merge folds
// the tasks' partial lists into one list per query.
constexpr uint32_t SCORE_TILE = 4096;
constexpr uint32_t TOPK_MAX = 256;
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

struct ScoreShared {
        uint32_t cand[SCORE_TILE];
        double score[SCORE_TILE];
        uint32_t hit[SCORE_TILE / 32]; // per scoring term: which matches this term holds
        uint32_t blkof[AND_WG + 1];
        uint32_t scan[8];
        uint32_t bcast[4];
        TopK tk;
};

__device__ __forceinline__ float bm25_term(const double idf, const uint32_t freq32) {
        const uint16_t freq = (uint16_t)freq32; // PostingsListIterator::freq is tokenpos_t (codecs.h:217)
        return (float)(idf * (double)(float)freq / (double)((float)freq + 1.2f));
}

__global__ __launch_bounds__(AND_WG) void k_score(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last,
                                                  const uint32_t *__restrict__ blk_off, const DevTerm *__restrict__ terms,
                                                  const DevQuery *__restrict__ plan, const DevTask *__restrict__ tasks,
                                                  const uint32_t *__restrict__ sched, const uint32_t *__restrict__ sterms,
                                                  const double *__restrict__ sweights, const uint32_t ntasks, uint32_t *__restrict__ ticket,
                                                  const uint32_t *__restrict__ out, const uint32_t *__restrict__ counts, const uint32_t k,
                                                  uint32_t *__restrict__ part_docs, double *__restrict__ part_scores,
                                                  uint32_t *__restrict__ part_counts, double *__restrict__ all_scores) {
        __shared__ ScoreShared sh;
        const uint32_t tid = threadIdx.x;
        const uint32_t wave = uni(tid >> 6);
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
                                sh.score[j] = 0.0;
                        }
                        __syncthreads();
                        for (uint32_t ti = 0; ti < q.nscore; ++ti) {
                                const DevTerm t = terms[sterms[q.score_base + ti]];
                                const double w = sweights[q.score_base + ti];
                                const uint32_t *bl = blk_last + t.first_block;
                                const uint32_t *bo = blk_off + t.first_block;
                                sh.blkof[0] = 0xffffffffu;
                                if (tid < SCORE_TILE / 32)
                                        sh.hit[tid] = 0;
                                __syncthreads();
                                for (uint32_t base = 0; base < C; base += AND_WG) {
                                        const uint32_t j = base + tid;
                                        uint32_t bj = 0xffffffffu, cv = 0;
                                        if (j < C) {
                                                cv = sh.cand[j];
                                                uint32_t lo = 0, hi = t.nblocks;
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
                                        if (j < C && bj < t.nblocks && bj != prevb) {
                                                const uint32_t prev = bj ? bl[bj - 1] : 0;
                                                const uint32_t last = bl[bj];
                                                const uint32_t off = bo[bj];
                                                const uint32_t n = index[off - 1];
                                                VbStream s;
                                                s.init(index + off);
                                                // deltas: merge the block's documents against the matches from j on; remember
                                                // the block positions (mask) and the matches (hit bits) that coincide.  Under an
                                                // OR a match need not be a document of this list.
                                                uint32_t doc = prev, ptr = j, mask = 0;
                                                for (uint32_t i = 0; i < n; ++i) {
                                                        doc = (i + 1 < n) ? doc + s.next() : last;
                                                        while (cv < doc) {
                                                                ++ptr;
                                                                cv = ptr < C ? sh.cand[ptr] : 0xffffffffu;
                                                        }
                                                        if (cv == doc) {
                                                                mask |= 1u << i;
                                                                atomicOr(&sh.hit[ptr >> 5], 1u << (ptr & 31));
                                                        }
                                                }
                                                // freqs follow the n-1 deltas; the i-th marked position belongs to the i-th
                                                // marked match (both ascending; only this lane marks matches in its block's range)
                                                ptr = j;
                                                for (uint32_t i = 0; i < n; ++i) {
                                                        const uint32_t f = s.next();
                                                        if ((mask >> i) & 1u) {
                                                                while (!((sh.hit[ptr >> 5] >> (ptr & 31)) & 1u))
                                                                        ++ptr;
                                                                sh.score[ptr] += (double)bm25_term(w, f);
                                                                ++ptr;
                                                        }
                                                }
                                        }
                                        __syncthreads();
                                }
                        }
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
                __syncthreads();
        }
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

// FNV-1a(64) of each query's docID set (little-endian bytes), one lane per query — verification helper
__global__ void k_hash_docsets(const DevQuery *__restrict__ plan, const DevTask *__restrict__ tasks_by_query,
                               const uint32_t *__restrict__ counts_by_query, const uint32_t nq, const uint32_t *__restrict__ out,
                               uint64_t *__restrict__ hashes) {
        const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
        if (s >= nq)
                return;
        const DevQuery q = plan[s];
        uint64_t h = 1469598103934665603ull;
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
        HIP_TRY(hipEventCreate(&d->ev0));
        HIP_TRY(hipEventCreate(&d->ev1));
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
        hipEventDestroy(d->ev0);
        hipEventDestroy(d->ev1);
        hipStreamDestroy(d->stream);
        delete d;
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
        (void)hits;
        (void)hits_len;
        if (!dev || !out || (!index && len) || (!terms && nterms))
                return fail(TRI_ERR_INVALID, "tri_index_upload: null argument");
        if (codec != TRI_CODEC_GOOGLE)
                return fail(TRI_ERR_UNSUPPORTED, "tri_index_upload: codec %d not supported yet", codec);
        if (len > 0xffffffffull)
                return fail(TRI_ERR_FORMAT, "index exceeds 32-bit chunk offsets (codecs.h:26)");
        HIP_TRY(hipSetDevice(dev->device));
        auto ix = std::make_unique<tri_index>();
        ix->dev = dev;
        ix->terms.resize(nterms);
        ix->tctx.assign(terms, terms + nterms);
        ix->docbytes.assign(nterms, 0);
        ix->hitbytes.assign(nterms, 0);
        std::vector<uint32_t> blk_last, blk_off;
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
                if (!t.size || !t.documents) {
                        dt.documents = 0;
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
                while (p != end) {
                        if (p + 3 > end)
                                return fail(TRI_ERR_FORMAT, "term %zu: truncated block header", ti);
                        const uint8_t *h = p;
                        uint32_t delta, blockLength;
                        p += h_vb_get(p, delta);
                        p += h_vb_get(p, blockLength);
                        const uint32_t n = *p++;
                        if (n < 1 || n > 32 || !delta || (uint64_t)(end - p) < blockLength)
                                return fail(TRI_ERR_FORMAT, "term %zu: bad block header (n=%u, delta=%u, len=%u)", ti, n, delta, blockLength);
                        lastDoc += delta;
                        const uint8_t *s = p;
                        for (uint32_t i = 0; i < 2 * n - 1; ++i)
                                s += h_vb_len(*s);
                        if ((uint64_t)(s - p) > blockLength)
                                return fail(TRI_ERR_FORMAT, "term %zu: deltas+freqs overrun the block", ti);
                        db += (uint64_t)(s - h);
                        hb += blockLength - (uint64_t)(s - p);
                        blk_last.push_back(lastDoc);
                        blk_off.push_back((uint32_t)(p - index));
                        dt.nblocks++;
                        dt.last_n = n;
                        docs += n;
                        p += blockLength;
                }
                if (docs != t.documents)
                        return fail(TRI_ERR_FORMAT, "term %zu: %u documents in blocks, %u declared", ti, docs, t.documents);
                ix->docbytes[ti] = db;
                ix->hitbytes[ti] = hb;
                postings += docs;
                docb += db;
                hitb += hb;
        }
        // per-window block index of the longer lists (TASK_DENSE reads two entries instead of searching the directory)
        const uint32_t max_doc = blk_last.empty() ? 0 : *std::max_element(blk_last.begin(), blk_last.end());
        ix->nwin = max_doc / SPAN_BITS + 2;
        std::vector<uint32_t> win;
        for (size_t ti = 0; ti < nterms; ++ti) {
                DevTerm &dt = ix->terms[ti];
                dt.win_off = 0xffffffffu;
                dt.pad[0] = dt.pad[1] = dt.pad[2] = 0;
                if (dt.nblocks < WIN_MIN_BLOCKS)
                        continue;
                dt.win_off = (uint32_t)win.size();
                const uint32_t *bl = &blk_last[dt.first_block];
                uint32_t b = 0;
                for (uint32_t w = 0; w < ix->nwin; ++w) {
                        const uint64_t key = (uint64_t)w * SPAN_BITS;
                        while (b < dt.nblocks && bl[b] < key)
                                ++b;
                        win.push_back(b);
                }
        }
        // device copies
        int rcw;
        if ((rcw = dev_upload(&ix->d_win, win)))
                return rcw;
        HIP_TRY(hipMalloc((void **)&ix->d_index, len + 64));
        HIP_TRY(hipMemset(ix->d_index, 0, len + 64));
        if (len)
                HIP_TRY(hipMemcpy(ix->d_index, index, len, hipMemcpyHostToDevice));
        int rc;
        if ((rc = dev_upload(&ix->d_blk_last, blk_last)) || (rc = dev_upload(&ix->d_blk_off, blk_off)) || (rc = dev_upload(&ix->d_terms, ix->terms)))
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

extern "C" void tri_index_destroy(tri_index *ix) {
        if (!ix)
                return;
        hipSetDevice(ix->dev->device);
        hipFree(ix->d_index);
        hipFree(ix->d_blk_last);
        hipFree(ix->d_blk_off);
        hipFree(ix->d_win);
        hipFree(ix->d_terms);
        delete ix;
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
        hipLaunchKernelGGL(k_decode_terms, grid, dim3(256), 0, dev->stream, ix->d_index, ix->d_blk_last, ix->d_blk_off, ix->d_terms, d_jobs, d_docs,
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
                        if (op == TRI_OP_TERM) {
                                n.term = arg;
                                n.cost = arg < ix->terms.size() ? ix->terms[arg].documents : 0;
                                n.empty = n.cost == 0; // unknown term == no documents (index_source.h:60-72)
                        } else {
                                if (arg < 1 || arg > st.size())
                                        return -1;
                                std::vector<int> kids(st.end() - arg, st.end());
                                st.resize(st.size() - arg);
                                if (op == TRI_OP_PHRASE) {
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
                                } else
                                        return -1;
                        }
                        nodes.push_back(std::move(n));
                        st.push_back((int)nodes.size() - 1);
                }
                return st.size() == 1 ? st[0] : -1;
        }
} // namespace

extern "C" int tri_batch_create(tri_index *ix, const uint32_t *prog, size_t prog_len, const tri_query *queries, size_t nq, const double *weights,
                                uint32_t flags, uint32_t topk, int similarity, tri_batch **out) {
        if (!ix || !out || (!prog && prog_len) || (!queries && nq))
                return fail(TRI_ERR_INVALID, "tri_batch_create: null argument");
        const uint32_t mode = flags & (TRI_FLAG_DOCUMENTS_ONLY | TRI_FLAG_ACCUMULATED_SCORE);
        if (mode == 0 || mode == (TRI_FLAG_DOCUMENTS_ONLY | TRI_FLAG_ACCUMULATED_SCORE))
                return fail(TRI_ERR_INVALID, "DocumentsOnly and AccumulatedScoreScheme are mutually exclusive; the default rich mode is not lowered (exec.h:45-48)");
        const bool scored = mode == TRI_FLAG_ACCUMULATED_SCORE;
        if (scored && topk > TOPK_MAX)
                return fail(TRI_ERR_INVALID, "AccumulatedScoreScheme: topk <= %u (0 = keep every match's score instead of a top-K)", TOPK_MAX);
        if (scored && similarity != TRI_SIM_BM25)
                return fail(TRI_ERR_UNSUPPORTED, "only TRI_SIM_BM25 is lowered");
        tri_dev *dev = ix->dev;
        HIP_TRY(hipSetDevice(dev->device));
        auto b = std::make_unique<tri_batch>();
        b->ix = ix;
        b->flags = flags;
        b->topk = topk;
        b->nq = nq;
        b->slot_of_query.assign(nq, UINT32_MAX);
        struct Tmp {
                DevQuery q;
                uint64_t cost;
                uint32_t nlead;
        };
        std::vector<Tmp> tmp;
        std::vector<PNode> nodes;
        for (size_t qi = 0; qi < nq; ++qi) {
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
                std::vector<uint32_t> leaves; // every TERM leaf in evaluation order: one scorer each
                auto add_group = [&](const PNode &g) -> bool {
                        std::vector<uint32_t> ts;
                        if (g.op == TRI_OP_TERM)
                                ts.push_back(g.term);
                        else if (g.op == TRI_OP_OR) {
                                for (int k : g.kids) {
                                        if (nodes[k].op != TRI_OP_TERM)
                                                return false;
                                        ts.push_back(nodes[k].term);
                                }
                        } else
                                return false;
                        leaves.insert(leaves.end(), ts.begin(), ts.end());
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
                bool ok = true;
                if (r.op == TRI_OP_AND)
                        for (int k : r.kids)
                                ok &= add_group(nodes[k]);
                else
                        ok = add_group(r);
                if (!ok)
                        return fail(TRI_ERR_UNSUPPORTED, "query %zu: only AND of terms / OR-of-terms groups (and a root OR of terms) are lowered so far", qi);
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
                if (uniq.size() > MAX_QTERMS)
                        return fail(TRI_ERR_UNSUPPORTED, "query %zu: more than %u terms", qi, MAX_QTERMS);
                const uint32_t nlead = (uint32_t)groups[0].size();
                const uint64_t lead_docs = gcost(groups[0]);
                Tmp t;
                t.q.score_base = (uint32_t)b->sterms.size();
                t.q.nscore = 0;
                if (scored) {
                        // one scorer per PostingsListIterator of the conjunction, summed in iterator order
                        // (docset_iterators_scorers.cpp:173-193); weight = BM25 idf (similarity.h:179-181, float math)
                        // unless the caller supplied ScorerWeights per TERM token
                        std::vector<std::pair<uint32_t, double>> sc;
                        for (uint32_t x : leaves)
                                sc.emplace_back(x, 0.0);
                        for (auto &e : sc) {
                                const uint32_t df = ix->terms[e.first].documents;
                                const float num = (float)((uint64_t)ix->info.docs_cnt - (uint64_t)df) + 0.5f;
                                const float den = (float)df + 0.5f;
                                e.second = (double)std::log(1 + num / den);
                        }
                        if (weights) {
                                // caller-provided weights follow the program's TERM tokens; map by first occurrence
                                for (auto &e : sc)
                                        for (uint32_t pi = 0; pi < tq.prog_len; ++pi)
                                                if (prog[tq.prog_off + pi] == TRI_TOK(TRI_OP_TERM, e.first)) {
                                                        e.second = weights[tq.prog_off + pi];
                                                        break;
                                                }
                        }
                        for (auto &e : sc) {
                                b->sterms.push_back(e.first);
                                b->sweights.push_back(e.second);
                        }
                        t.q.nscore = (uint32_t)sc.size();
                }
                t.q.nterms = (uint32_t)uniq.size();
                t.q.term_base = (uint32_t)b->qterms.size();
                t.q.out_cap = 0;
                t.q.out_off = 0;
                t.q.qid = (uint32_t)qi;
                t.cost = 0;
                t.nlead = nlead;
                {
                        std::vector<uint32_t> seen;
                        for (uint32_t tt : uniq) {
                                const uint32_t term = tt & ~QT_GROUP;
                                b->qterms.push_back(tt);
                                if (std::find(seen.begin(), seen.end(), term) == seen.end()) {
                                        seen.push_back(term);
                                        b->term_bytes += ix->docbytes[term];
                                }
                        }
                        // cost estimate: the lead group is decoded fully; every other list costs min(its blocks x 32, lead docs x 32)
                        for (size_t i = 0; i < uniq.size(); ++i) {
                                const DevTerm &tk = ix->terms[uniq[i] & ~QT_GROUP];
                                t.cost += i < nlead ? tk.documents : 32ull * std::min<uint64_t>(tk.nblocks, lead_docs);
                        }
                }
                tmp.push_back(t);
        }
        std::stable_sort(tmp.begin(), tmp.end(), [](const Tmp &a, const Tmp &c) { return a.cost > c.cost; });
        uint64_t off = 0;
        b->plan.reserve(tmp.size());
        // cut every query into tasks of roughly TASK_COST postings, then schedule heaviest first
        constexpr uint64_t TASK_COST = 96 * 1024;
        // tunables (environment overrides exist for tests and perf probes)
        uint64_t DENSE_MIN_POSTINGS = 512 * 1024;
        if (const char *e = getenv("TRINITY_DENSE_MIN"))
                DENSE_MIN_POSTINGS = strtoull(e, nullptr, 10);
        std::vector<std::pair<uint64_t, uint32_t>> order; // (task cost, task index)
        for (auto &t : tmp) {
                const uint32_t slot = (uint32_t)b->plan.size();
                b->slot_of_query[t.q.qid] = slot;
                const uint32_t *qt = &b->qterms[t.q.term_base];
                const DevTerm &lead = ix->terms[qt[0] & ~QT_GROUP];
                const uint32_t nlead = t.nlead;
                uint64_t lead_docs = 0;
                for (uint32_t k = 0; k < nlead; ++k)
                        lead_docs += ix->terms[qt[k] & ~QT_GROUP].documents;
                // TASK_DENSE (bitmap windows) when the lead group is an OR (it has to be materialised as a set anyway), or
                // when every other list is within a factor 32 of the lead (no block could be skipped) and there is enough
                // work per docID window to keep 256 lanes busy
                uint64_t sumdf = 0;
                bool dense = t.q.nterms >= 2;
                uint32_t last_doc = 0xffffffffu, glast = 0; // no match beyond the group whose lists end first
                for (uint32_t k = 0; k < t.q.nterms; ++k) {
                        const DevTerm &tk = ix->terms[qt[k] & ~QT_GROUP];
                        sumdf += tk.documents;
                        dense &= tk.nblocks <= lead_docs;
                        if (k && (qt[k] & QT_GROUP)) {
                                last_doc = std::min(last_doc, glast);
                                glast = 0;
                        }
                        glast = std::max(glast, ix->h_blk_last[tk.first_block + tk.nblocks - 1]);
                }
                last_doc = std::min(last_doc, glast);
                dense &= sumdf >= DENSE_MIN_POSTINGS;
                dense |= nlead > 1;
                t.q.out_off = off;
                t.q.first_task = (uint32_t)b->tasks.size();
                if (dense) {
                        const uint32_t nwin = last_doc / SPAN_BITS + 1;
                        const uint64_t per_win = std::max<uint64_t>(1, sumdf / (ix->info.docs_cnt / SPAN_BITS + 1));
                        const uint32_t win_per_task = (uint32_t)std::max<uint64_t>(1, TASK_COST / per_win);
                        uint32_t ord = 0;
                        uint64_t lead_blocks = 0;
                        for (uint32_t k = 0; k < nlead; ++k)
                                lead_blocks += ix->terms[qt[k] & ~QT_GROUP].nblocks;
                        for (uint32_t wb = 0; wb < nwin; wb += win_per_task, ++ord) {
                                const uint32_t we = std::min(nwin, wb + win_per_task);
                                // matches of windows [wb, we) are lead-group documents of blocks b1 .. (next task's b1) of every
                                // lead list: a private region (+32 slots of slack per lead list and task for the straddling block)
                                uint64_t b1 = 0;
                                for (uint32_t k = 0; k < nlead; ++k) {
                                        const DevTerm &tk = ix->terms[qt[k] & ~QT_GROUP];
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
                }
                off += t.q.out_cap;
                t.q.ntasks = (uint32_t)b->tasks.size() - t.q.first_task;
                b->plan.push_back(t.q);
        }
        std::stable_sort(order.begin(), order.end(), [](const auto &a, const auto &c) { return a.first > c.first; });
        std::vector<uint32_t> sched(order.size());
        for (size_t i = 0; i < order.size(); ++i)
                sched[i] = order[i].second;
        b->out_capacity = off;
        int rc;
        if ((rc = dev_upload(&b->d_plan, b->plan)) || (rc = dev_upload(&b->d_qterms, b->qterms)) || (rc = dev_upload(&b->d_tasks, b->tasks)) ||
            (rc = dev_upload(&b->d_sched, sched)))
                return rc;
        HIP_TRY(hipMalloc((void **)&b->d_out, (off + 64) * 4));
        HIP_TRY(hipMalloc((void **)&b->d_counts, (b->tasks.size() + 1) * 4));
        HIP_TRY(hipMalloc((void **)&b->d_ticket, 256));
        if (scored) {
                if ((rc = dev_upload(&b->d_sterms, b->sterms)) || (rc = dev_upload(&b->d_sweights, b->sweights)))
                        return rc;
                const size_t nt = b->tasks.size();
                if (!topk)
                        HIP_TRY(hipMalloc((void **)&b->d_all_scores, (off + 64) * 8));
                HIP_TRY(hipMalloc((void **)&b->d_part_docs, (nt * topk + 1) * 4));
                HIP_TRY(hipMalloc((void **)&b->d_part_scores, (nt * topk + 1) * 8));
                HIP_TRY(hipMalloc((void **)&b->d_part_counts, (nt + 1) * 4));
                HIP_TRY(hipMalloc((void **)&b->d_top_docs, (nq * topk + 1) * 4));
                HIP_TRY(hipMalloc((void **)&b->d_top_scores, (nq * topk + 1) * 4));
                HIP_TRY(hipMalloc((void **)&b->d_top_counts, (nq + 1) * 4));
                HIP_TRY(hipMemset(b->d_top_counts, 0, (nq + 1) * 4)); // queries that can never match keep count 0
        }
        b->info.nqueries = nq;
        b->info.out_capacity = off;
        b->info.launches = scored ? 3 : 1;
        *out = b.release();
        return TRI_OK;
}

extern "C" void tri_batch_destroy(tri_batch *b) {
        if (!b)
                return;
        hipSetDevice(b->ix->dev->device);
        hipFree(b->d_plan);
        hipFree(b->d_tasks);
        hipFree(b->d_sched);
        hipFree(b->d_qterms);
        hipFree(b->d_out);
        hipFree(b->d_counts);
        hipFree(b->d_ticket);
        hipFree(b->d_hashes);
        hipFree(b->d_sterms);
        hipFree(b->d_sweights);
        hipFree(b->d_part_docs);
        hipFree(b->d_part_scores);
        hipFree(b->d_part_counts);
        hipFree(b->d_top_docs);
        hipFree(b->d_top_scores);
        hipFree(b->d_top_counts);
        hipFree(b->d_all_scores);
        delete b;
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
        HIP_TRY(hipEventRecord(dev->ev0, dev->stream));
        if (n) {
                HIP_TRY(hipMemsetAsync(b->d_ticket, 0, 256, dev->stream));
                const uint32_t grid = std::min<uint32_t>(n, (uint32_t)dev->cus * 4);
                hipLaunchKernelGGL(k_and, dim3(grid), dim3(AND_WG), 0, dev->stream, b->ix->d_index, b->ix->d_blk_last, b->ix->d_blk_off, b->ix->d_win,
                                   b->ix->d_terms,
                                   b->d_plan, b->d_tasks, b->d_sched, b->d_qterms, n, b->d_ticket, b->d_out, b->d_counts);
                HIP_TRY(hipGetLastError());
                if (b->flags & TRI_FLAG_ACCUMULATED_SCORE) {
                        hipLaunchKernelGGL(k_score, dim3(std::min<uint32_t>(n, (uint32_t)dev->cus * 2)), dim3(AND_WG), 0, dev->stream, b->ix->d_index,
                                           b->ix->d_blk_last, b->ix->d_blk_off, b->ix->d_terms, b->d_plan, b->d_tasks, b->d_sched, b->d_sterms, b->d_sweights, n,
                                           b->d_ticket + 32, b->d_out, b->d_counts, b->topk, b->d_part_docs, b->d_part_scores, b->d_part_counts,
                                           b->d_all_scores);
                        HIP_TRY(hipGetLastError());
                        const uint32_t nqs = (uint32_t)b->plan.size();
                        if (b->topk)
                                hipLaunchKernelGGL(k_topk_merge, dim3(std::min<uint32_t>(nqs, (uint32_t)dev->cus * 8)), dim3(AND_WG), 0, dev->stream, b->d_plan, nqs,
                                           b->topk, b->d_part_docs, b->d_part_scores, b->d_part_counts, b->d_top_docs, b->d_top_scores, b->d_top_counts);
                        HIP_TRY(hipGetLastError());
                }
        }
        HIP_TRY(hipEventRecord(dev->ev1, dev->stream));
        return TRI_OK;
}

extern "C" int tri_batch_sync(tri_batch *b) {
        if (!b)
                return fail(TRI_ERR_INVALID, "null batch");
        tri_dev *dev = b->ix->dev;
        HIP_TRY(hipSetDevice(dev->device));
#if (defined(TRI_TRACE) && !defined(TRI_TRACE_NOPOLL)) || defined(TRI_POLL)
        {
                const char *w = getenv("TRINITY_WATCHDOG_S");
                const double limit = w ? atof(w) : 10.0;
                double waited = 0;
                while (hipEventQuery(dev->ev1) == hipErrorNotReady) {
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
        if (hipEventElapsedTime(&ms, dev->ev0, dev->ev1) == hipSuccess)
                b->info.last_run_ms = ms;
        b->h_counts.resize(b->tasks.size());
        if (!b->tasks.empty())
                HIP_TRY(hipMemcpy(b->h_counts.data(), b->d_counts, b->tasks.size() * 4, hipMemcpyDeviceToHost));
        uint64_t m = 0;
        b->h_query_counts.assign(b->plan.size(), 0);
        for (size_t sidx = 0; sidx < b->plan.size(); ++sidx) {
                const DevQuery &q = b->plan[sidx];
                for (uint32_t t = 0; t < q.ntasks; ++t)
                        b->h_query_counts[sidx] += b->h_counts[q.first_task + t];
                m += b->h_query_counts[sidx];
        }
        b->info.matches = m;
        if (b->flags & TRI_FLAG_ACCUMULATED_SCORE) {
                uint64_t outb = 0; // 8 B x min(matches, K) per query
                for (uint64_t c : b->h_query_counts)
                        outb += 8 * std::min<uint64_t>(c, b->topk);
                b->info.algorithmic_bytes = b->term_bytes + outb;
        } else
                b->info.algorithmic_bytes = b->term_bytes + 4 * m; // docbytes + 4 B per match (docs-only)
        b->synced = true;
        return TRI_OK;
}

extern "C" int tri_batch_get_info(const tri_batch *b, tri_batch_info *info) {
        if (!b || !info)
                return fail(TRI_ERR_INVALID, "null argument");
        *info = b->info;
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
