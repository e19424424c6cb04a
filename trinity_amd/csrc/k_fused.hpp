// k_fused.hpp — AccumulatedScoreScheme in ONE pass: decode -> match -> BM25 -> top-K per docID window, nothing spilled to HBM
// Part of libtrinity_hip.so (MI355X / gfx950); included by trinity_hip.hip.  New code, no reference source.
#pragma once
#include "k_score.hpp"

// What the reference does per matching document (docset_spans.cpp:681-790 window accumulate; docset_iterators_scorers.cpp:10-36,
// 107-193 score wrappers; similarity.h:228-235 BM25; matches.h:155-171 the application's top-K heap) happens here per docID
// WINDOW, for every query whose lists are dense enough that no block could be skipped (the planner's TASK_DENSE class) and
// that asks for a top-K:
//
//   * LDS holds one word per document of the window: 32 bits, or — queries of <= 5 distinct terms — 16 bits, which doubles the
//     window (HW = 1: twice the documents per barrier, per directory lookup, per sweep).  The query's distinct terms (<= 8) are
//     SLOTS; a slot owns a field of the word (32-bit words: 8 bits with <= 4 slots, else 4; 16-bit words: 8 / 5 / 4 / 3 bits for
//     <= 2 / 3 / 4 / 5 slots) holding 0 = "the document does not have the term", else min(freq, cap) + 1.  A posting is ONE fire-and-forget ds_or_b32: the docID picks the word, the
//     freq the code.  All terms of the query are decoded in one pass (no per-group passes, no bitmap folds), each list exactly
//     once, freqs in the same block visit as the deltas.
//   * the epilogue sweeps the window's words: the CNF predicate is a handful of mask tests on the word (a required group =
//     "any of these fields non-zero", the excluded group = "all zero"), the score is the sum of one table lookup per BYTE of the
//     word (tables built per task in LDS from the scorers' weights: entry = sum over the byte's fields of
//     IndexSourceTermsScorer::score(freq), doubles — the sum of a handful of floats is exact in double, so the order of
//     summation cannot matter), compared against the task's current k-th best (score desc, docID asc); the few survivors are
//     appended to a candidate buffer that is pruned by rank when it fills.  The sweep re-zeroes the words.
//   * a field that saturates (freq > cap) makes its table entry NaN; such a (rare) document is rescored exactly: the term's
//     block is located through the directory and its freq decoded (fused_lookup_freq).
//   * a row (<= 32 documents) that reaches beyond the window leaves a HINT — the first of its documents past the window — so
//     the windows it merely spans neither decode it again nor count as windows its list can match in: sparse lists cost one
//     decode per row, and a conjunction jumps to the next window every required group can reach.
// Per task the kernel leaves min(matches, k) (docID, score) pairs and the match count; k_topk_merge folds the tasks of a query.
#ifndef TRI_FUS_WG
#define TRI_FUS_WG 512
#endif
#ifndef TRI_FUS_UNROLL
#define TRI_FUS_UNROLL 2 // postings per trip of the PFOR row loop (1: 78.9 ms, 2: 74.0 ms at the time it was measured)
#endif
constexpr int FUS_WG = TRI_FUS_WG;
// (FUS_CELLS, FUS_W: dev_structs.hpp)
// window geometry by word width (HW = 1: two documents per 32-bit LDS word)
template <int HW>
struct FusGeom {
        static constexpr uint32_t W = FUS_W << HW;         // documents per window
        static constexpr uint32_t CELLS = FUS_CELLS << HW; // docID cells per window
};
constexpr uint32_t FUS_CAP = 512; // candidate buffer; k <= TOPK_MAX = 256
constexpr uint32_t FUS_CHUNKS = FUS_W / (4 * FUS_WG); // the sweep takes the window in 16-byte chunks: this many per thread
constexpr uint32_t FUS_WLIST = 512;                   // per wave: documents waiting to be scored (a word position adds up to 64, 128 with 16-bit words: flushed when the next might not fit)
static_assert(FUS_W % (4 * FUS_WG) == 0, "the sweep deals whole 16-byte chunks of words to every thread");
static_assert(TOPK_MAX * 2 <= FUS_CAP, "a pruned buffer must leave room for a round of newcomers");

struct FusedShared {
        alignas(16) uint32_t acc[FUS_W + 64]; // [FUS_W] = sink for documents outside the window
        double tab[4][256];       // per byte of the word: score contribution of the byte's fields (NaN: a saturated field)
        double tk_s[FUS_CAP];
        uint32_t tk_d[FUS_CAP];
        DevTerm term[FUS_MAX_SLOTS];
        uint32_t hint_row[FUS_MAX_SLOTS]; // the slot's row that reached beyond the last window it was decoded in ...
        uint32_t hint_doc[FUS_MAX_SLOTS]; // ... and the first of its documents past that window
        uint16_t wlist[FUS_WG / 64][FUS_WLIST]; // per wave: window-relative docIDs waiting to be scored
        double ub[FUS_MAX_SLOTS];               // per slot: an upper bound of its score contribution
        double thr_s;
        uint32_t thr_d;
        uint32_t emask; // fields of the ESSENTIAL slots: a document none of whose essential fields is set cannot beat the threshold
        uint32_t tk_n, tk_full, overflow, matches;
        uint32_t bcast[4];
        alignas(16) uint32_t rng_lo[FUS_MAX_SLOTS]; // the next window (next_w; 0xffffffff: none) and the slots' row ranges in it, left by wave 0
        alignas(16) uint32_t rng_cnt[FUS_MAX_SLOTS];
        uint32_t next_w;
        uint32_t dirpos[4][FUS_MAX_SLOTS]; // wave 0's per-slot directory positions between two look-aheads (k_fused: find_window)
        DevFused fq; // the query's slot map, staged once per task (dynamic indexing stays in LDS, not in scratch)
};

// workgroups a CU holds: LDS (160 KB) and the 2048-thread limit; the register budget follows (launch bounds)
constexpr uint32_t FUS_WGS_PER_CU = (160u * 1024u / sizeof(FusedShared)) < (2048u / FUS_WG) ? (160u * 1024u / sizeof(FusedShared)) : (2048u / FUS_WG);
static_assert(FUS_WGS_PER_CU >= 1, "the window state must fit the CU's LDS");

typedef uint32_t u32_a1 __attribute__((aligned(1)));
typedef uint64_t u64_a1 __attribute__((aligned(1)));
typedef uint32_t u32x2_a1 __attribute__((ext_vector_type(2), aligned(1)));
typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
__device__ __forceinline__ uint32_t ldu32(const uint8_t *p) { return *(const u32_a1 *)p; } // gfx950 global loads take any alignment
__device__ __forceinline__ uint64_t ldu64(const uint8_t *p) { return *(const u64_a1 *)p; }

// bytes of an ints() group from its header word as the row record caches it (bit 31: every value equal to the low 31 bits)
__device__ __forceinline__ uint32_t pfor_group_bytes(const uint32_t hdr) {
        if (hdr >> 31) {
                const uint32_t v = hdr & 0x7fffffffu;
                return 1u + (v < (1u << 7) ? 1u : v < (1u << 14) ? 2u : v < (1u << 21) ? 3u : v < (1u << 28) ? 4u : 5u);
        }
        const uint32_t b = hdr & 0xffu, nexc = (hdr >> 8) & 0xffu, eb = (hdr >> 16) & 0xffu;
        return 1u + 4u * (1u + 4u * b + (nexc + 3u) / 4u + (nexc * eb + 31u) / 32u);
}

// ---- LUCENE: one quarter (32 values) of an ints() group, read by one lane without the general stream's state machine
// (codec_streams.hpp LValStream).  The quarter's packed words sit in a register queue of NW words fetched with wide loads (a
// quarter of width b occupies exactly b words; widths > NW re-load the upper half of the queue every NW / 2 words, far ahead of
// their use); the low parts come out of a two-word funnel (v_alignbit), a used-up word is a shift of the queue.  The quarter's
// exceptions — found through the per-row exception index built at upload instead of a scan of the group's list — sit in a
// 32-bit position mask and a packed queue of high parts and are patched branch-free.  Everything the lane needs to address its
// loads (offset, header words, exception index) comes in ONE 16-byte row record, so a row costs two memory round trips.
template <int NW>
struct PfRegs {
        uint32_t w[NW];
        const uint8_t *wp; // wide quarters: the NW / 2 words that follow the queue
        uint32_t sh, b, mask, excmask, eb, emask, used;
        uint64_t hq;
        // g: the group's first byte; hdr: its header word (row record); e0 / cnt: the quarter's run of the exception list.
        // false: not representable here (> 16 exceptions or > 64 bits of high parts in the quarter): general streams.
        __device__ __forceinline__ bool init(const uint8_t *g, const uint32_t hdr, const uint32_t q, const uint32_t e0, const uint32_t cnt) {
                excmask = 0;
                eb = 0;
                emask = 0;
                hq = 0;
                sh = 0;
                used = 0;
                wp = g;
                if (hdr >> 31) { // lucene_codec.cpp:31-39: every value equal
                        w[0] = hdr & 0x7fffffffu;
#pragma unroll
                        for (int k = 1; k < NW; ++k)
                                w[k] = 0;
                        b = 0;
                        mask = 0xffffffffu;
                        return true;
                }
                b = hdr & 0xffu;
                const uint32_t nexc = (hdr >> 8) & 0xffu;
                eb = (hdr >> 16) & 0xffu;
                if (cnt > 16 || cnt * eb > 64)
                        return false;
                mask = b >= 32 ? 0xffffffffu : ((1u << b) - 1u);
                const uint8_t *q0 = g + 5 + 4 * (q * b);
#pragma unroll
                for (int k = 0; k < NW; k += 4) {
                        const u32x4_a1 v = *(const u32x4_a1 *)(q0 + 4 * k);
                        w[k] = v.x, w[k + 1] = v.y, w[k + 2] = v.z, w[k + 3] = v.w;
                }
                wp = q0 + 4 * NW;
                if (cnt) {
                        emask = eb >= 32 ? 0xffffffffu : ((1u << eb) - 1u);
                        const uint8_t *epos = g + 5 + 16 * b;
                        const uint8_t *ehigh = epos + 4 * ((nexc + 3) / 4);
                        uint64_t p0 = ldu64(epos + e0);
                        const uint64_t p1 = ldu64(epos + e0 + 8);
                        const uint32_t bit0 = e0 * eb;
                        const uint8_t *hp = ehigh + (bit0 >> 3);
                        const uint32_t hs = bit0 & 7u;
                        const uint64_t h0 = ldu64(hp), h1 = ldu64(hp + 8);
                        hq = hs ? ((h0 >> hs) | (h1 << (64 - hs))) : h0; // 64 bits from any bit offset
                        for (uint32_t e = 0; e < cnt; ++e) {
                                if (e == 8)
                                        p0 = p1;
                                excmask |= 1u << ((uint32_t)p0 & 31u);
                                p0 >>= 8;
                        }
                }
                return true;
        }
        __device__ __forceinline__ uint32_t next(const uint32_t i) {
                uint32_t x = __builtin_amdgcn_alignbit(w[1], w[0], sh) & mask;
                sh += b;
                // a word is used up: shift the queue.  The branches are WAVE-uniform (lanes of one list share a width and cross word
                // boundaries together), the work inside is per lane by select — a lane-divergent branch here made the compiler copy
                // the whole queue on every value
                if (__builtin_amdgcn_ballot_w64(sh >= 32) != 0ull) {
                        const bool r = sh >= 32;
#pragma unroll
                        for (int k = 0; k + 1 < NW; ++k)
                                w[k] = r ? w[k + 1] : w[k];
                        sh = r ? sh - 32 : sh;
                        used += r ? 1u : 0u;
                        const bool f = r && used == NW / 2 && b > NW; // a wide quarter has moved half a queue: fetch the next half
                        if (__builtin_amdgcn_ballot_w64(f) != 0ull) {
                                if (NW == 8) {
                                        const u32x4_a1 v = *(const u32x4_a1 *)wp; // (lanes that do not need it read a valid address and drop it)
                                        w[4] = f ? v.x : w[4], w[5] = f ? v.y : w[5], w[6] = f ? v.z : w[6], w[7] = f ? v.w : w[7];
                                } else {
                                        const u32x2_a1 v = *(const u32x2_a1 *)wp;
                                        w[NW / 2] = f ? v.x : w[NW / 2], w[NW / 2 + 1] = f ? v.y : w[NW / 2 + 1];
                                }
                                wp += f ? 2 * NW : 0;
                                used = f ? 0u : used;
                        }
                }
                const uint32_t m = (uint32_t)(-(int32_t)((excmask >> i) & 1u)); // all ones at an exception
                x |= ((uint32_t)hq & emask & m) << (b & 31u);
                hq >>= (eb & m);
                return x;
        }
};

// The freq of `doc` in term t (the document is known to be one of the term's): rescoring of a saturated field.
template <int CODEC>
__device__ __noinline__ uint32_t fused_lookup_freq(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last, const uint32_t *__restrict__ blk_off,
                                                   const DevTerm &t, const uint32_t doc) {
        const uint32_t *bl = blk_last + t.first_block;
        uint32_t lo = 0, hi = t.nblocks;
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
        return f & 0xffffu;
}

// One posting into the window words.  rel = docID - w0 (documents below the window wrap to huge values: they and the documents
// past the window land in the sink word); `past` keeps the smallest rel - W >= 0, i.e. the row's first document past the window.
template <int HW>
__device__ __forceinline__ void fused_post(uint32_t *acc, const uint32_t rel, const uint32_t f, const uint32_t cap, const uint32_t shift, uint32_t &past) {
        constexpr uint32_t W = FusGeom<HW>::W;
        past = min(past, rel - W); // (in-window and below-window documents give values >= 2^31: never the minimum of a real one)
        const uint32_t r = min(rel, W), code = min(f & 0xffffu, cap) + 1u;
        if (HW)
                atomicOr(&acc[r >> 1], code << (shift + ((r & 1u) << 4))); // (the sink r == W is the low half of word FUS_W)
        else
                atomicOr(&acc[r], code << shift);
}

// What a decoded posting is handed to.  The window-word kernels (k_fused) use FusedPost: one ds_or of the slot's freq code into the
// document's word; the bit-plane kernels (k_planes.hpp) bring their own.  post(rel, f): a document at window-relative position rel
// (documents below the window wrap to huge values) with frequency f; post.doc(rel): the same without a frequency (rows decoded for
// presence only — NEEDF = false).
template <int HW>
struct FusedPost {
        uint32_t *acc;
        uint32_t shift, cap;
        uint32_t past = 0xffffffffu; // the row's first document past the window, as rel - W (>= 2^31: none)
        __device__ __forceinline__ void operator()(const uint32_t rel, const uint32_t f) { fused_post<HW>(acc, rel, f, cap, shift, past); }
        __device__ __forceinline__ void doc(const uint32_t rel) { fused_post<HW>(acc, rel, cap + 1u, cap, shift, past); }
};

// One directory row (<= 32 documents) of a slot's term through the codec's general value streams: GOOGLE rows that miss the
// 63-byte path, and the PFOR quarters the register reader (PfRegs) does not take.
template <int CODEC, bool NEEDF, class POST>
__device__ __forceinline__ void row_streams(const uint8_t *__restrict__ index, const DevTerm &t, const uint32_t b, const uint32_t off, const uint32_t n,
                                            const uint32_t prev, const uint32_t last, const uint32_t w0, POST &post, const uint32_t google_delta_bytes = 0) {
        uint32_t rel = prev - w0;
        DeltaStream<CODEC> ds;
        ds.init(index, t, b, off);
        if (!NEEDF) {
                for (uint32_t i = 0; i < n; ++i) {
                        rel = (i + 1 < n) ? rel + ds.next() : last - w0;
                        post.doc(rel);
                }
                return;
        }
        FreqStream<CODEC> fs;
        if (CODEC == CODEC_GOOGLE) { // the freqs follow the n - 1 deltas (their byte length is known from the directory): both walked in step
                DeltaStream<CODEC> sk;
                sk.init(index, t, b, off + google_delta_bytes);
                fs.init(index, t, b, off, sk);
        } else
                fs.init(index, t, b, off, ds);
        for (uint32_t i = 0; i < n; ++i) {
                rel = (i + 1 < n) ? rel + ds.next() : last - w0;
                post(rel, fs.next());
        }
}
// (LUCENE: out of line — the general streams' state must not weigh on the registers of the PFOR fast path; only quarters with
// more than 16 exceptions come here.)
template <bool NEEDF, class POST>
__device__ __noinline__ void row_streams_lucene(const uint8_t *__restrict__ index, const DevTerm &t, const uint32_t b, const uint32_t off, const uint32_t n,
                                                const uint32_t prev, const uint32_t last, const uint32_t w0, POST &post) {
        row_streams<CODEC_LUCENE, NEEDF, POST>(index, t, b, off, n, prev, last, w0, post);
}

// One directory row into `post`.  LUCENE: rec = the row record {group offset, exception index, deltas header, freqs header};
// GOOGLE: rec_x = payload offset, rec_y = byte length of the block's deltas.
template <int CODEC, bool NEEDF, class POST>
__device__ __forceinline__ void row_decode(const uint8_t *__restrict__ index, const DevTerm &t, const uint32_t b, const uint32_t rec_x, const uint32_t rec_y,
                                           const uint32_t rec_z, const uint32_t rec_w, const uint32_t n, const uint32_t prev, const uint32_t last,
                                           const uint32_t w0, POST &post PROF_ARG) {
        if (CODEC == CODEC_LUCENE) {
                uint32_t rel = prev - w0;
                if (b >= t.npfor) { // the list's varbyte tail (lucene_codec.cpp:321-337): (delta, freq) pairs, fewer than 128 documents
                        VbStream vb;
                        vb.init(index + rec_x);
                        for (uint32_t i = 0; i < n; ++i) {
                                rel += vb.next();
                                const uint32_t f = vb.next();
                                if (NEEDF)
                                        post(rel, f);
                                else
                                        post.doc(rel);
                        }
                        return;
                }
                const uint8_t *g = index + rec_x;
                PfRegs<8> rd;
                PfRegs<4> rf;
                const bool okd = rd.init(g, rec_z, b & 3u, rec_y & 0xffu, (rec_y >> 8) & 0xffu);
                const bool okf = !NEEDF || rf.init(g + pfor_group_bytes(rec_z), rec_w, b & 3u, (rec_y >> 16) & 0xffu, rec_y >> 24);
                PROF_LAP(9);
                if (okd && okf) {
#pragma unroll TRI_FUS_UNROLL
                        for (uint32_t i = 0; i < 32; ++i) {
                                rel += rd.next(i);
                                if (NEEDF)
                                        post(rel, rf.next(i));
                                else
                                        post.doc(rel);
                        }
                        PROF_LAP(10);
                        return;
                }
                row_streams_lucene<NEEDF, POST>(index, t, b, rec_x, n, prev, last, w0, post);
                return;
        }
        if (CODEC == CODEC_GOOGLE && n == 32) {
                // a full block whose 31 deltas and 32 freqs are single bytes (every block of a head term): 63 bytes in four wide loads,
                // then byte adds from registers — no varint stream, and no walk over the deltas to find where the freqs start
                const uint8_t *p = index + rec_x;
                const u32x4_a1 A = *(const u32x4_a1 *)p, B = *(const u32x4_a1 *)(p + 16), Cq = *(const u32x4_a1 *)(p + 32), Dq = *(const u32x4_a1 *)(p + 48);
                const uint32_t w[16] = {A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w, Cq.x, Cq.y, Cq.z, Cq.w, Dq.x, Dq.y, Dq.z, Dq.w};
                uint32_t any = w[15] & 0x00ffffffu; // (byte 63 is the block's first hit)
#pragma unroll
                for (int k = 0; k < 15; ++k)
                        any |= w[k];
                if (!(any & 0x80808080u)) {
                        uint32_t rel = prev - w0;
#pragma unroll
                        for (int i = 0; i < 32; ++i) {
                                rel = i < 31 ? rel + ((w[i >> 2] >> ((i & 3) * 8)) & 0xffu) : last - w0;
                                if (NEEDF)
                                        post(rel, (w[(31 + i) >> 2] >> (((31 + i) & 3) * 8)) & 0xffu);
                                else
                                        post.doc(rel);
                        }
                        return;
                }
        }
        row_streams<CODEC, NEEDF, POST>(index, t, b, rec_x, n, prev, last, w0, post, rec_y);
}

// k_fused's form: the row into the window words; returns the row's first document past the window as rel - W (>= 2^31: none).
template <int CODEC, int HW>
__device__ __forceinline__ uint32_t fused_row(const uint8_t *__restrict__ index, const DevTerm &t, const uint32_t b, const uint32_t rec_x,
                                              const uint32_t rec_y, const uint32_t rec_z, const uint32_t rec_w, const uint32_t n, const uint32_t prev,
                                              const uint32_t last, const uint32_t w0, uint32_t *acc, const uint32_t shift, const uint32_t cap PROF_ARG) {
        FusedPost<HW> post{acc, shift, cap};
        row_decode<CODEC, true, FusedPost<HW>>(index, t, b, rec_x, rec_y, rec_z, rec_w, n, prev, last, w0, post PROF_PASS);
        return post.past;
}

// Keep the best k of the n (<= FUS_CAP) buffered candidates, best first (rank by counting: the order is strict).
__device__ void fused_prune(FusedShared &sh, const uint32_t n, const uint32_t k) {
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
        sh.overflow = 0;
        if (m == k) {
                sh.tk_full = 1;
                sh.thr_s = sh.tk_s[k - 1];
                sh.thr_d = sh.tk_d[k - 1];
        }
        __syncthreads();
}

// MaxScore pruning (exact): with the slots ordered by their score bound, the longest prefix whose bounds sum to less than the current
// k-th best cannot lift a document over the threshold on its own — only documents holding at least one of the OTHER (essential)
// slots need a score.  Recomputed by every lane (uniform) whenever the threshold moves; no threshold yet: every slot is essential.
__device__ __forceinline__ uint32_t fused_essential(const FusedShared &sh, const uint32_t nslots, const uint32_t fbits) {
        if (!uni(sh.tk_full))
                return 0xffffffffu;
        const double thr = sh.thr_s;
        const uint32_t fm = (1u << fbits) - 1u;
        uint32_t done = 0, emask = 0;
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
                        emask |= fm << (best * fbits);
        }
        return emask;
}

// Score the documents a wave has queued (window-relative docIDs in its wlist), one per lane: table lookups (one per chunk of
// the word: cb bits, a whole number of fields), the rare exact rescoring, threshold, append to the workgroup's candidate buffer.
// A full buffer puts the word back for the resumed sweep.
template <int CODEC, int HW, int GEN>
__device__ __noinline__ uint32_t fused_flush(FusedShared &sh, const uint32_t wn, const uint32_t w0, const uint32_t nch, const uint32_t cb, const bool full,
                                             const double thr_s, const uint32_t thr_d, const uint8_t *__restrict__ index,
                                             const uint32_t *__restrict__ blk_last, const uint32_t *__restrict__ blk_off, const DevQuery &q,
                                             const uint32_t *__restrict__ sterms, const double *__restrict__ sweights, const int sim) {
        const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
        const DevFused &fq = sh.fq;
        const uint32_t cm = (1u << cb) - 1u;
        const bool ttm = GEN && (uni(fq.mode) & FUS_MODE_TT);
        uint32_t nback = 0; // documents put back (wave-uniform: counted where the whole wave has reconverged)
        for (uint32_t c0 = 0; c0 < wn; c0 += 64) {
                const uint32_t c = c0 + lane;
                bool back = false;
                if (c < wn) {
                        const uint32_t idx = sh.wlist[wv][c];
                        uint32_t x;
                        if (HW) { // two documents per word: the other half may be on another lane's plate
                                const uint32_t wsh = (idx & 1u) << 4;
                                x = (sh.acc[idx >> 1] >> wsh) & 0xffffu;
                                atomicAnd(&sh.acc[idx >> 1], ~(0xffffu << wsh));
                        } else {
                                x = sh.acc[idx];
                                sh.acc[idx] = 0;
                        }
                        double s = sh.tab[0][x & cm];
                        if (nch > 1)
                                s += sh.tab[1][(x >> cb) & cm];
                        if (nch > 2)
                                s += sh.tab[2][(x >> (2 * cb)) & cm];
                        if (nch > 3)
                                s += sh.tab[3][(x >> (3 * cb)) & cm];
                        const uint32_t doc = w0 + idx;
                        if (ttm) { // a general tree: the scorer leaves that sit on a document of this presence pattern (DevFused::ctt)
                                s = 0.0;
                                const uint32_t fbits = fq.fbits, cap = fq.cap, fmask = (1u << fbits) - 1u;
                                uint32_t p = 0;
                                for (uint32_t sl = 0; sl < fq.nslots; ++sl)
                                        p |= (((x >> (sl * fbits)) & fmask) ? 1u : 0u) << sl;
                                for (uint32_t si = 0; si < q.nscore; ++si) {
                                        if (!((fq.ctt[si][p >> 5] >> (p & 31u)) & 1u))
                                                continue;
                                        const uint32_t sl = fq.leaf_slot[si], code = (x >> (sl * fbits)) & fmask;
                                        const uint32_t f = code > cap ? fused_lookup_freq<CODEC>(index, blk_last, blk_off, sh.term[sl], doc) : code - 1u;
                                        s += (double)sim_score(sim, sweights[q.score_base + si], f);
                                }
                        } else if (!(s == s)) { // a saturated field: rescore from the postings, scorer by scorer
                                s = 0.0;
                                const uint32_t fbits = fq.fbits, cap = fq.cap, fmask = (1u << fbits) - 1u;
                                for (uint32_t si = 0; si < q.nscore; ++si) {
                                        const uint32_t term = sterms[q.score_base + si];
                                        uint32_t sl = 0;
                                        while (sl + 1 < fq.nslots && fq.term[sl] != term)
                                                ++sl;
                                        const uint32_t code = (x >> (sl * fbits)) & fmask;
                                        if (!code)
                                                continue;
                                        const uint32_t f = code > cap ? fused_lookup_freq<CODEC>(index, blk_last, blk_off, sh.term[sl], doc) : code - 1u;
                                        s += (double)sim_score(sim, sweights[q.score_base + si], f);
                                }
                        }
                        if (!full || better(s, doc, thr_s, thr_d)) {
                                const uint32_t slot = atomicAdd(&sh.tk_n, 1u);
                                if (slot >= FUS_CAP) { // no room: the word goes back, the resumed sweep takes (and counts) it again
                                        back = true;
                                        sh.overflow = 1;
                                        if (HW)
                                                atomicOr(&sh.acc[idx >> 1], x << ((idx & 1u) << 4));
                                        else
                                                sh.acc[idx] = x;
                                } else {
                                        sh.tk_s[slot] = s;
                                        sh.tk_d[slot] = doc;
                                }
                        }
                }
                nback += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(back));
        }
        return nback; // (returned, not subtracted through a reference: the caller's counter stays in a scalar register)
}

// Sweep stage 1 over the window's words, 16 bytes per lane at a time: the CNF predicate (PK: 0 = a union — any field of the one
// required group; 1 = up to four groups from registers, excluded fields; 2 = any number of groups, masked documents) counts the
// matches; a match that holds an ESSENTIAL slot is queued on the wave's list, everything else is re-zeroed at once; the list is
// scored by the wave itself when it fills and at the end (fused_flush).
template <int CODEC, int PK, int HW, int GEN>
__device__ __forceinline__ void fused_sweep(FusedShared &sh, const uint32_t w0, const uint32_t nch, const uint32_t cb, const bool full, const double thr_s,
                                            const uint32_t thr_d, const uint32_t emask, const uint32_t nmask, const uint32_t nreq, const uint32_t gm0,
                                            const uint32_t gm1, const uint32_t gm2, const uint32_t gm3, const uint32_t *__restrict__ masked,
                                            uint32_t &wave_matches, const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last,
                                            const uint32_t *__restrict__ blk_off, const DevQuery &q, const uint32_t *__restrict__ sterms,
                                            const double *__restrict__ sweights, const int sim PROF_ARG) {
        const uint32_t wave = uni(threadIdx.x >> 6);
        constexpr uint32_t DPW = HW ? 2 : 1; // documents per word
        // (the lane's word address is recomputed per chunk from the hardware lane count: kept in a register across the flush calls it
        //  was spilled, and every chunk began with a scratch round trip)
        auto lane_id = []() {
                uint32_t l;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
                return l;
        };
        uint32_t wn = 0;                     // entries on this wave's list (wave-uniform)
        const uint32_t fmode = (GEN && PK == 2) ? uni(sh.fq.mode) : 0u, tt_nslots = (GEN && PK == 2) ? uni(sh.fq.nslots) : 0u,
                       tt_fbits = (GEN && PK == 2) ? uni(sh.fq.fbits) : 0u, tt_fmask = (1u << tt_fbits) - 1u;
        if (HW && PK != 2) {
                // 16-bit words, predicates of up to four groups: BOTH documents of a word at once.  nz16(y) = v_pk_min_u16(y, 1|1<<16)
                // has bit 0 / 16 set where the low / high half of y is non-zero; a required group is nz16(x & its fields), the
                // excluded group its complement; the matches are counted with one v_bcnt per word, and only a word that holds a
                // document with an ESSENTIAL field (rare once the threshold stands) takes the per-document path below.
                auto nz16 = [](const uint32_t y) {
                        uint32_t r;
                        asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(y), "s"(0x00010001u));
                        return r;
                };
                auto rep = [](const uint32_t m16) { return m16 | (m16 << 16); };
                const uint32_t G0 = rep(gm0), G1 = rep(gm1), G2 = rep(gm2), G3 = rep(gm3), N = rep(nmask), E = rep(emask);
                uint32_t lane_matches = 0;
#pragma unroll 1
                for (uint32_t j = 0; j < FUS_CHUNKS; ++j) {
                        const uint32_t i0 = j * (4 * FUS_WG) + 256 * wave + 4 * lane_id();
                        const uint4 v = *(const uint4 *)&sh.acc[i0];
                        if (__builtin_amdgcn_ballot_w64((v.x | v.y | v.z | v.w) != 0) == 0ull)
                                continue; // nothing in this wave's slice
                        const uint32_t xs[4] = {v.x, v.y, v.z, v.w};
                        uint32_t es[4], anye = 0;
#pragma unroll
                        for (uint32_t c = 0; c < 4; ++c) {
                                const uint32_t x = xs[c];
                                uint32_t m = nz16(x & G0);
                                if (PK == 1) {
                                        m &= nz16(x & G1);
                                        if (nreq > 2) // (uniform)
                                                m &= nz16(x & G2) & nz16(x & G3);
                                        if (nmask)
                                                m &= ~nz16(x & N);
                                }
                                lane_matches += __popc(m);
                                es[c] = m & nz16(x & E);
                                anye |= es[c];
                        }
                        if (__builtin_amdgcn_ballot_w64(anye != 0) == 0ull) { // no document of these 512 words can beat the threshold
                                uint32_t z; // (materialised here: hoisted out of the loop the zero vector was spilled and re-loaded from scratch per chunk)
                                asm volatile("v_mov_b32 %0, 0" : "=v"(z));
                                *(uint4 *)&sh.acc[i0] = make_uint4(z, z, z, z);
                                continue;
                        }
                        // the queued documents keep their codes until they are scored, everything else is re-zeroed now (a flush below
                        // reads — and clears — the words of the documents on the list)
                        *(uint4 *)&sh.acc[i0] = make_uint4(xs[0] & (es[0] * 0xffffu), xs[1] & (es[1] * 0xffffu), xs[2] & (es[2] * 0xffffu), xs[3] & (es[3] * 0xffffu));
                        // the lane's (<= 8) essential documents as a bit set, queued one per round: bit 2c + h = half h of word c
                        uint32_t e8 = ((es[0] | (es[0] >> 15)) & 3u) | (((es[1] | (es[1] >> 15)) & 3u) << 2) | (((es[2] | (es[2] >> 15)) & 3u) << 4) |
                                      (((es[3] | (es[3] >> 15)) & 3u) << 6);
                        for (;;) {
                                const uint64_t bal = __builtin_amdgcn_ballot_w64(e8 != 0);
                                if (bal == 0ull)
                                        break;
                                if (wn > FUS_WLIST - 64) { // this round might not fit: score what is queued
                                        __builtin_amdgcn_wave_barrier();
                                        PROF_LAP(6);
                                        wave_matches -= fused_flush<CODEC, HW, GEN>(sh, wn, w0, nch, cb, full, thr_s, thr_d, index, blk_last, blk_off, q, sterms, sweights, sim);
                                        PROF_LAP(11);
                                        wn = 0;
                                }
                                if (e8)
                                        sh.wlist[wave][wn + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = (uint16_t)(i0 * 2 + (uint32_t)__builtin_ctz(e8));
                                wn += (uint32_t)__popcll(bal);
                                e8 &= e8 - 1u;
                        }
                }
                PROF_LAP(6);
                // the lanes' counts into the wave's (scalar) counter
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1)
                        lane_matches += __shfl_xor(lane_matches, d, 64);
                wave_matches += uni(lane_matches);
                if (wn) {
                        __builtin_amdgcn_wave_barrier();
                        wave_matches -= fused_flush<CODEC, HW, GEN>(sh, wn, w0, nch, cb, full, thr_s, thr_d, index, blk_last, blk_off, q, sterms, sweights, sim);
                }
                PROF_LAP(11);
                return;
        }
#pragma unroll 1
        for (uint32_t j = 0; j < FUS_CHUNKS; ++j) { // (kept a loop: unrolled, the chunk bodies' invariants crowd the register file)
                const uint32_t i0 = j * (4 * FUS_WG) + 256 * wave + 4 * lane_id();
                const uint4 v = *(const uint4 *)&sh.acc[i0];
                if (__builtin_amdgcn_ballot_w64((v.x | v.y | v.z | v.w) != 0) == 0ull)
                        continue; // nothing in this wave's slice
                uint32_t xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (uint32_t c = 0; c < 4; ++c) {
                        uint32_t keep = 0;
#pragma unroll
                        for (uint32_t h = 0; h < DPW; ++h) {
                                const uint32_t x = HW ? (xs[c] >> (16 * h)) & 0xffffu : xs[c];
                                const uint32_t idx = (i0 + c) * DPW + h; // window-relative docID
                                bool m;
                                if (PK == 0)
                                        m = (x & gm0) != 0;
                                else if (PK == 1)
                                        m = (x & nmask) == 0 && (x & gm0) && (x & gm1) && (x & gm2) && (x & gm3);
                                else {
                                        if (fmode & FUS_MODE_TT) { // a general tree: the truth table over the slots' presence bits
                                                uint32_t p = 0;
                                                for (uint32_t sl = 0; sl < tt_nslots; ++sl)
                                                        p |= (((x >> (sl * tt_fbits)) & tt_fmask) ? 1u : 0u) << sl;
                                                m = x != 0 && ((sh.fq.tt[p >> 5] >> (p & 31u)) & 1u);
                                        } else {
                                                m = x != 0 && (x & nmask) == 0 && (x & gm0) && (x & gm1) && (x & gm2) && (x & gm3);
                                                for (uint32_t g = 4; g < nreq; ++g)
                                                        m &= (x & sh.fq.gmask[g]) != 0;
                                        }
                                        if (masked && m) { // masked_documents_registry::test (docidupdates.h:90-119)
                                                const uint32_t doc = w0 + idx;
                                                m = !((masked[doc >> 5] >> (doc & 31u)) & 1u);
                                        }
                                        if ((fmode & FUS_MODE_EMIT) && m) // DocumentsOnly: the window's matches as a bitmap, expanded by the caller
                                                atomicOr(&((uint32_t *)sh.tk_s)[idx >> 5], 1u << (idx & 31u));
                                }
                                wave_matches += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(m)); // (wave-uniform counter: scalar registers)
                                const bool e = m && (x & emask) != 0 && !(PK == 2 && (fmode & FUS_MODE_EMIT));
                                const uint64_t bal = __builtin_amdgcn_ballot_w64(e);
                                if (bal != 0ull) { // (wave-uniform)
                                        if (e)
                                                sh.wlist[wave][wn + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = (uint16_t)idx;
                                        wn += (uint32_t)__popcll(bal);
                                }
                                keep |= ((e || (GEN && (fmode & FUS_MODE_EMIT) && m)) ? x : 0u) << (16 * h); // queued (emitted) documents keep their code until they are scored (written)
                        }
                        sh.acc[i0 + c] = keep; // (written back word by word: a flush may run before the chunk is through)
                        if (wn > FUS_WLIST - 64 * DPW) { // the next word position might not fit: score what is queued
                                __builtin_amdgcn_wave_barrier();
                                wave_matches -= fused_flush<CODEC, HW, GEN>(sh, wn, w0, nch, cb, full, thr_s, thr_d, index, blk_last, blk_off, q, sterms, sweights, sim);
                                wn = 0;
                        }
                }
        }
        if (wn) {
                __builtin_amdgcn_wave_barrier();
                wave_matches -= fused_flush<CODEC, HW, GEN>(sh, wn, w0, nch, cb, full, thr_s, thr_d, index, blk_last, blk_off, q, sterms, sweights, sim);
        }
}

// GEN = 1: the instantiation that also knows general trees (truth-table predicate, DocumentsOnly emission) — kept out of the
// CNF kernels, whose registers it would weigh on.
template <int CODEC, int HW, int GEN>
__global__ __launch_bounds__(FUS_WG, (FUS_WGS_PER_CU * FUS_WG + 255) / 256) void k_fused(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last,
                                                     const uint32_t *__restrict__ blk_off, const uint4 *__restrict__ blk_rec,
                                                     const uint32_t *__restrict__ blk_doff, const uint32_t *__restrict__ win, const DevTerm *__restrict__ terms,
                                                     const DevQuery *__restrict__ plan, const DevFused *__restrict__ fused,
                                                     const DevTask *__restrict__ tasks, const uint32_t *__restrict__ sched,
                                                     const uint32_t *__restrict__ sterms, const double *__restrict__ sweights, const uint32_t ntasks,
                                                     uint32_t *__restrict__ ticket, uint32_t *__restrict__ counts, const uint32_t k,
                                                     uint32_t *__restrict__ part_docs, double *__restrict__ part_scores,
                                                     uint32_t *__restrict__ part_counts, const uint32_t *__restrict__ masked, const int sim,
                                                     uint32_t *__restrict__ out, double *__restrict__ all_scores, uint32_t *__restrict__ allow) {
        __shared__ FusedShared sh;
        constexpr uint32_t W = FusGeom<HW>::W, CELLS = FusGeom<HW>::CELLS;
        const uint32_t tid = threadIdx.x;
        const uint32_t wave = uni(tid >> 6);
        for (uint32_t i = tid; i < FUS_W + 64; i += FUS_WG)
                sh.acc[i] = 0;
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
                // ---- per task: the query's slot map, the slots' terms, the score tables, an empty candidate buffer
                {
                        const uint32_t wi = min(tid, (uint32_t)(sizeof(DevFused) / 4 - 1)); // (every lane stores: no divergent branch around the barriers)
                        ((uint32_t *)&sh.fq)[wi] = ((const uint32_t *)(fused + q.fused_idx))[wi];
                }
                __syncthreads();
                const DevFused &fq = sh.fq;
                const uint32_t nslots = uni(fq.nslots), fbits = uni(fq.fbits), cap = uni(fq.cap), nreq = uni(fq.nreq), nmask = uni(fq.nmask);
                const uint32_t fmode = GEN ? uni(fq.mode) : 0u; // general tree (truth-table predicate) / DocumentsOnly (matches written to out[])
                uint32_t produced = 0;               // FUS_MODE_EMIT: docIDs this task has written
                if (fmode & FUS_MODE_EMIT)           // (the window's match bitmap lives in the candidate buffer nobody else needs)
                        for (uint32_t i = tid; i < 2 * FUS_CAP; i += FUS_WG)
                                ((uint32_t *)sh.tk_s)[i] = 0;
                const uint32_t per = 8 / fbits, cb = per * fbits, nch = (nslots + per - 1) / per; // the word is scored in nch chunks of cb bits (per fields each)
                const uint32_t kk = min(tid & 63u, nslots - 1); // in EVERY wave lane s (< nslots) tracks slot s, the lanes above mirror the last slot:
                                                                // a wave reads the slots' block ranges out of its own lanes (readlane), no LDS, no barrier
                sh.term[kk] = terms[fq.term[kk]];
                sh.hint_row[kk] = 0xffffffffu;
                sh.tk_n = 0;
                sh.tk_full = 0;
                sh.overflow = 0;
                sh.matches = 0;
                sh.emask = 0xffffffffu; // no threshold yet: every slot is essential
                if (!(fmode & FUS_MODE_EMIT)) { // (emitting tasks keep no top-K: no tables, no bounds — and in the default mode there are no weights)
                        // tab[c][v]: the fields inside chunk c of the word (v < 1 << cb).  code 0 = absent; code - 1 = freq; code cap + 1 = saturated
                        const uint32_t fmask = (1u << fbits) - 1u;
                        for (uint32_t e = tid; e < nch * 256; e += FUS_WG) {
                                const uint32_t c = e >> 8, v = e & 255u;
                                double s = 0.0;
                                bool sat = false;
                                for (uint32_t j = 0; j < per; ++j) {
                                        const uint32_t slot = c * per + j, code = (v >> (j * fbits)) & fmask;
                                        if (slot >= nslots || !code)
                                                continue;
                                        if (code > cap) {
                                                sat = true;
                                                continue;
                                        }
                                        const uint32_t term = fq.term[slot];
                                        for (uint32_t si = 0; si < q.nscore; ++si) // every scorer of this term, reference order
                                                if (sterms[q.score_base + si] == term)
                                                        s += (double)sim_score(sim, sweights[q.score_base + si], code - 1u);
                                }
                                sh.tab[c][v] = sat ? __builtin_nan("") : s;
                        }
                        // per slot an upper bound of what it can add: BM25 float(w f / (f + 1.2)) < w; TF-IDF sqrt(f) w with f <= 65535; Trivial f
                        double ubs = 0.0;
                        for (uint32_t si = 0; si < q.nscore; ++si)
                                if (sterms[q.score_base + si] == fq.term[kk]) {
                                        const double wgt = sweights[q.score_base + si];
                                        ubs += sim == TRI_SIM_TRIVIAL ? 65535.0 : sim == TRI_SIM_TFIDF ? (wgt > 0 ? 256.0 * wgt : 0.0) : (wgt > 0 ? wgt : 0.0);
                                }
                        sh.ub[kk] = ubs * (1.0 + 1e-6);
                }
                __syncthreads();
                const uint32_t wfirst = task.tile_begin;
                // wave 0, lane s (< nslots): the directory position of slot s's list — indexed lists keep the two cell-index entries of the
                // window's ends (the far one is fetched a window ahead), short lists a cursor with its block's bounds.  The state lives in LDS
                // (sh.dirpos) between the look-aheads: held in registers it weighed on — and was spilled by — every wave of the workgroup.
                if (wave == 0) {
                        const DevTerm myt = sh.term[kk];
                        const uint32_t *mybl = blk_last + myt.first_block;
                        uint32_t e_lo = 0, e_hi = 0, pf = 0, cur = 0;
                        if (myt.win_off != 0xffffffffu) {
                                e_lo = win[myt.win_off + wfirst * CELLS];
                                e_hi = win[myt.win_off + (wfirst + 1) * CELLS];
                                pf = win[myt.win_off + (wfirst + 2) * CELLS];
                        } else {
                                uint32_t lo = 0, hi = myt.nblocks; // first block whose last document >= the task's first docID
                                const uint32_t key = wfirst * W;
                                while (lo < hi) {
                                        const uint32_t mid = (lo + hi) >> 1;
                                        if (mybl[mid] < key)
                                                lo = mid + 1;
                                        else
                                                hi = mid;
                                }
                                cur = lo;
                                e_lo = cur < myt.nblocks ? mybl[cur] : 0xffffffffu; // (short lists: e_lo / e_hi hold the cursor block's last / previous docID)
                                e_hi = cur ? mybl[cur - 1] : 0u;
                        }
                        sh.dirpos[0][kk] = e_lo;
                        sh.dirpos[1][kk] = e_hi;
                        sh.dirpos[2][kk] = pf;
                        sh.dirpos[3][kk] = cur;
                }
                // the first four required groups' masks live in registers (a missing group tests true on any non-zero word)
                const uint32_t gm0 = uni(fq.gmask[0]), gm1 = nreq > 1 ? uni(fq.gmask[1]) : 0xffffffffu, gm2 = nreq > 2 ? uni(fq.gmask[2]) : 0xffffffffu,
                               gm3 = nreq > 3 ? uni(fq.gmask[3]) : 0xffffffffu;
                const uint32_t gsl0 = uni(fq.gslots[0]), gsl1 = nreq > 1 ? uni(fq.gslots[1]) : 0u, gsl2 = nreq > 2 ? uni(fq.gslots[2]) : 0u,
                               gsl3 = nreq > 3 ? uni(fq.gslots[3]) : 0u; // (an absent group has no slots: its "next possible" would be "never" — skipped below)
                uint32_t wave_matches = 0; // matches this wave has counted (wave-uniform)
                // ---- WAVE 0 finds the next window that can hold a match and the slots' row ranges in it (lane s tracks slot s), while
                //      the other waves sweep the current one; everybody picks the result up behind the sweep's barrier
                auto find_window = [&](uint32_t w) {
                        const DevTerm myt = sh.term[kk];
                        const uint32_t *mybl = blk_last + myt.first_block;
                        const bool indexed = myt.win_off != 0xffffffffu;
                        uint32_t e_lo = sh.dirpos[0][kk], e_hi = sh.dirpos[1][kk], pf = sh.dirpos[2][kk], cur = sh.dirpos[3][kk];
                        uint32_t cur_last = e_lo, cur_prev = e_hi; // (short lists)
                        auto save = [&]() {
                                sh.dirpos[0][kk] = indexed ? e_lo : cur_last;
                                sh.dirpos[1][kk] = indexed ? e_hi : cur_prev;
                                sh.dirpos[2][kk] = pf;
                                sh.dirpos[3][kk] = cur;
                        };
                        for (;;) {
                                if (w >= task.tile_end) {
                                        sh.next_w = 0xffffffffu;
                                        return;
                                }
                                const uint32_t w0 = w * W, wlast = w0 + (W - 1);
                                // my slot's blocks that can hold documents of [w0, wlast], and the first docID >= w0 it may still hold
                                uint32_t b_lo, b_hi, np = 0xffffffffu;
                                bool here = false;
                                if (indexed) {
                                        b_lo = e_lo;
                                        b_hi = min(e_hi, myt.nblocks - 1);
                                        here = e_lo != e_hi; // a block ends inside the window
                                } else {
                                        while (cur < myt.nblocks && cur_last < w0) {
                                                ++cur;
                                                cur_prev = cur_last;
                                                cur_last = cur < myt.nblocks ? mybl[cur] : 0xffffffffu;
                                        }
                                        b_lo = b_hi = cur;
                                        if (cur_last < wlast) // (rare for a short list: further blocks end inside the window)
                                                while (b_hi + 1 < myt.nblocks && mybl[b_hi] < wlast)
                                                        ++b_hi;
                                }
                                uint32_t cnt = 0;
                                if (b_lo < myt.nblocks) {
                                        cnt = b_hi - b_lo + 1;
                                        const uint32_t hd = sh.hint_doc[kk];
                                        if (sh.hint_row[kk] == b_lo) { // the row was decoded before: its next document is known
                                                np = max(w0, hd);
                                                if (hd > wlast)
                                                        cnt = 0; // ... and lies past this window: the row (and with it the list) has nothing here
                                        } else {
                                                const uint32_t first_possible = here ? w0 : !indexed ? cur_prev + 1 : (b_lo ? mybl[b_lo - 1] + 1 : 1u);
                                                np = max(w0, first_possible);
                                        }
                                }
                                // no match before the latest "first possible document" over the required groups (a group: its earliest slot)
                                uint32_t need = 0;
                                for (uint32_t g = 0; g < nreq; ++g) {
                                        const uint32_t gs = g == 0 ? gsl0 : g == 1 ? gsl1 : g == 2 ? gsl2 : g == 3 ? gsl3 : uni(fq.gslots[g]);
                                        uint32_t gnp = 0xffffffffu;
#pragma unroll
                                        for (uint32_t s = 0; s < FUS_MAX_SLOTS; ++s) {
                                                const uint32_t snp = (uint32_t)__builtin_amdgcn_readlane((int)np, (int)s);
                                                gnp = (s < nslots && ((gs >> s) & 1u)) ? min(gnp, snp) : gnp;
                                        }
                                        need = max(need, gnp);
                                }
                                if (need == 0xffffffffu) { // a required group is exhausted: no further match anywhere
                                        sh.next_w = 0xffffffffu;
                                        return;
                                }
                                const uint32_t wnext = need / W;
                                if (wnext > w) { // nothing can match before window wnext: jump
                                        w = wnext;
                                        if (indexed && w < task.tile_end) {
                                                e_lo = win[myt.win_off + w * CELLS];
                                                e_hi = win[myt.win_off + (w + 1) * CELLS];
                                                pf = win[myt.win_off + (w + 2) * CELLS];
                                        }
                                        continue;
                                }
                                if ((tid & 63u) < nslots) {
                                        sh.rng_lo[kk] = b_lo;
                                        sh.rng_cnt[kk] = cnt;
                                }
                                sh.next_w = w;
                                // the far cell-index entry of the window after this one travels while this one is worked on
                                if (indexed) {
                                        e_lo = e_hi;
                                        e_hi = pf;
                                        pf = win[myt.win_off + (w + 3) * CELLS];
                                }
                                save();
                                return;
                        }
                };
                PROF_LAP(0);
                // software pipeline: wave 0 looks for window n + 1 while window n (set in the previous round) is swept
                uint32_t look_from = wfirst, w0 = 0;
                bool have = false;
                for (;;) {
                        if (wave == 0) // (window n's hints are in — the set pass ended with a barrier)
                                find_window(look_from);
                        PROF_LAP(1);
                        if (have) {
                                // ---- sweep, stage 1 per 16-byte chunk: the CNF predicate counts the matches; a match that holds an ESSENTIAL slot is
                                //      queued on the wave's list (everything else is re-zeroed at once); the list is scored by the wave itself when
                                //      it fills and at the end (fused_flush), so only documents that can beat the threshold ever pay the table
                                //      lookups.  A full candidate buffer is pruned and the sweep resumed over the words that were put back.
                                for (;;) {
                                        const bool full = uni(sh.tk_full) != 0;
                                        const double thr_s = sh.thr_s;
                                        const uint32_t thr_d = sh.thr_d;
                                        const uint32_t emask = uni(sh.emask);
                                        // the predicate in its cheapest form for the query at hand (uniform): one required group and nothing excluded (a
                                        // union: any of its fields), up to four groups from registers, or the general walk with masked documents on top
                                        if (!fmode && !masked && nreq == 1 && nmask == 0)
                                                fused_sweep<CODEC, 0, HW, GEN>(sh, w0, nch, cb, full, thr_s, thr_d, emask, nmask, nreq, gm0, gm1, gm2, gm3, masked, wave_matches, index, blk_last, blk_off, q, sterms, sweights, sim PROF_PASS);
                                        else if (!fmode && !masked && nreq <= 4)
                                                fused_sweep<CODEC, 1, HW, GEN>(sh, w0, nch, cb, full, thr_s, thr_d, emask, nmask, nreq, gm0, gm1, gm2, gm3, masked, wave_matches, index, blk_last, blk_off, q, sterms, sweights, sim PROF_PASS);
                                        else
                                                fused_sweep<CODEC, 2, HW, GEN>(sh, w0, nch, cb, full, thr_s, thr_d, emask, nmask, nreq, gm0, gm1, gm2, gm3, masked, wave_matches, index, blk_last, blk_off, q, sterms, sweights, sim PROF_PASS);
                                        PROF_LAP(6);
                                        __syncthreads();
                                        PROF_LAP(7);
                                        if (fmode & FUS_MODE_EMIT) { // the window's match bitmap -> ascending docIDs (two words per thread)
                                                uint32_t *bm = (uint32_t *)sh.tk_s;
                                                constexpr uint32_t BW = W / 32;
                                                static_assert(BW <= 2 * FUS_WG && BW <= 2 * FUS_CAP, "two bitmap words per thread, inside the candidate buffer");
                                                uint32_t m0 = 2 * tid < BW ? bm[2 * tid] : 0u, m1 = 2 * tid + 1 < BW ? bm[2 * tid + 1] : 0u;
                                                uint32_t wtot;
                                                const uint32_t ex = wave_excl_scan((uint32_t)(__popc(m0) + __popc(m1)), wtot);
                                                sh.tk_d[tid >> 6] = wtot;
                                                __syncthreads();
                                                uint32_t wbase = 0, total = 0;
                                                for (uint32_t wv = 0; wv < FUS_WG / 64; ++wv) {
                                                        wbase += wv < (tid >> 6) ? sh.tk_d[wv] : 0u;
                                                        total += sh.tk_d[wv];
                                                }
                                                uint64_t o = task.out_off + produced + wbase + ex;
                                                const uint32_t fmask = (1u << fbits) - 1u;
                                                for (uint32_t half = 0; half < 2; ++half)
                                                        for (uint32_t mm = half ? m1 : m0; mm; mm &= mm - 1u, ++o) {
                                                                const uint32_t idx = 64 * tid + 32 * half + (uint32_t)__builtin_ctz(mm), doc = w0 + idx;
                                                                out[o] = doc;
                                                                if (!all_scores && !allow)
                                                                        continue;
                                                                // the document's word is still there: which leaves sit on it (DevFused::ctt), and what they add
                                                                const uint32_t x = HW ? (sh.acc[idx >> 1] >> ((idx & 1u) << 4)) & 0xffffu : sh.acc[idx];
                                                                uint32_t p = 0, on = 0;
                                                                for (uint32_t sl = 0; sl < nslots; ++sl)
                                                                        p |= (((x >> (sl * fbits)) & fmask) ? 1u : 0u) << sl;
                                                                double sc = 0.0;
                                                                for (uint32_t si = 0; si < uni(fq.nleaf); ++si) {
                                                                        if (!((fq.ctt[si][p >> 5] >> (p & 31u)) & 1u))
                                                                                continue;
                                                                        on |= 1u << si;
                                                                        if (all_scores) {
                                                                                const uint32_t sl = fq.leaf_slot[si], code = (x >> (sl * fbits)) & fmask;
                                                                                const uint32_t f = code > cap ? fused_lookup_freq<CODEC>(index, blk_last, blk_off, sh.term[sl], doc) : code - 1u;
                                                                                sc += (double)sim_score(sim, sweights[q.score_base + si], f);
                                                                        }
                                                                }
                                                                if (all_scores)
                                                                        all_scores[o] = sc;
                                                                if (allow)
                                                                        allow[o] = on;
                                                        }
                                                if (2 * tid < BW)
                                                        bm[2 * tid] = 0;
                                                if (2 * tid + 1 < BW)
                                                        bm[2 * tid + 1] = 0;
                                                produced += uni(total);
                                                __syncthreads();
                                                for (uint32_t i = tid; i < FUS_W; i += FUS_WG) // (the sweep left the matches' words in place)
                                                        sh.acc[i] = 0;
                                                __syncthreads();
                                        }
                                        const uint32_t ov = uni(sh.overflow);
                                        const uint32_t n = min(uni(sh.tk_n), FUS_CAP);
                                        if (ov || n > (FUS_CAP + k) / 2) {
                                                __syncthreads(); // (every lane has read overflow / tk_n)
                                                fused_prune(sh, n, k);
                                                sh.emask = fused_essential(sh, nslots, fbits); // same value from every lane
                                                __syncthreads();
                                        }
                                        if (!ov)
                                                break;
                                }
                                PROF_LAP(4);
                        } else
                                __syncthreads();
                        const uint32_t w = uni(sh.next_w);
                        if (w == 0xffffffffu)
                                break;
                        w0 = w * W;
                        have = true;
                        look_from = w + 1;
                        uint32_t s_lo[FUS_MAX_SLOTS], s_cnt[FUS_MAX_SLOTS], total = 0;
                        {
                                const uint4 l0 = *(const uint4 *)&sh.rng_lo[0], l1 = *(const uint4 *)&sh.rng_lo[4], c0 = *(const uint4 *)&sh.rng_cnt[0],
                                            c1 = *(const uint4 *)&sh.rng_cnt[4];
                                const uint32_t lo8[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w}, cn8[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
                                for (uint32_t s = 0; s < FUS_MAX_SLOTS; ++s) {
                                        s_lo[s] = uni(lo8[s]);
                                        s_cnt[s] = s < nslots ? uni(cn8[s]) : 0u;
                                        total += s_cnt[s];
                                }
                        }
                        PROF_LAP(1);
                        // ---- set pass: the rows of all slots form one work list, dealt out round by round
                        for (uint32_t v0 = 0; v0 < total; v0 += FUS_WG) {
                                const uint32_t v = v0 + tid;
                                if (v < total) {
                                        uint32_t s = 0, r = v, lo = s_lo[0];
#pragma unroll
                                        for (uint32_t k2 = 0; k2 + 1 < FUS_MAX_SLOTS; ++k2) { // which slot's rows v falls into (ranges are wave-uniform)
                                                const bool nextslot = s == k2 && r >= s_cnt[k2];
                                                r = nextslot ? r - s_cnt[k2] : r;
                                                lo = nextslot ? s_lo[k2 + 1] : lo;
                                                s = nextslot ? k2 + 1 : s;
                                        }
                                        const DevTerm t = sh.term[s];
                                        const uint32_t b = lo + r;
                                        const uint32_t *bl = blk_last + t.first_block;
                                        const uint32_t prev = b ? bl[b - 1] : 0;
                                        const uint32_t last = bl[b];
                                        uint32_t past;
                                        if (CODEC == CODEC_LUCENE) {
                                                const uint4 rec = blk_rec[t.first_block + b];
#ifdef TRI_PROF
                                                if (rec.x + prev + last == 0xfffffff0u) // (probe builds: the record has arrived when the lap is taken)
                                                        sh.bcast[3] = 1;
                                                PROF_LAP(8);
#endif
                                                past = fused_row<CODEC, HW>(index, t, b, rec.x, rec.y, rec.z, rec.w, TRI_BLOCK_N(t, b, index, 0), prev, last, w0, sh.acc,
                                                                        s * fbits, cap PROF_PASS);
                                        } else {
                                                const uint32_t off = blk_off[t.first_block + b];
                                                // (GOOGLE: the freqs start where the deltas end — their length comes from the delta stream's offset column)
                                                const uint32_t dlen = blk_doff[t.first_block + b + 1] - blk_doff[t.first_block + b] - 1u;
                                                past = fused_row<CODEC, HW>(index, t, b, off, dlen, 0, 0, TRI_BLOCK_N(t, b, index, off), prev, last, w0, sh.acc, s * fbits, cap PROF_PASS);
                                        }
                                        if (past < 0x80000000u) { // the row reaches past the window (it is the slot's last row here): leave the hint
                                                sh.hint_row[s] = b;
                                                sh.hint_doc[s] = w0 + W + past;
                                        }
                                }
                        }
                        PROF_LAP(2);
                        __syncthreads();
                        PROF_LAP(3);
                }
                // ---- the task's result: its best k (ranked) and its match count
                __syncthreads();
                fused_prune(sh, min(uni(sh.tk_n), FUS_CAP), k);
                atomicAdd(&sh.matches, (tid & 63u) == 0 ? wave_matches : 0u); // (every lane issues it: no single-lane branch)
                __syncthreads();
                const uint32_t n = (fmode & FUS_MODE_EMIT) ? 0u : uni(sh.tk_n);
                for (uint32_t i = tid; i < n; i += FUS_WG) {
                        part_docs[(uint64_t)tix * k + i] = sh.tk_d[i];
                        part_scores[(uint64_t)tix * k + i] = sh.tk_s[i];
                }
                if (wave == 0) {
                        if (!(fmode & FUS_MODE_EMIT))
                                part_counts[tix] = n;
                        counts[tix] = uni(sh.matches);
                }
                __syncthreads();
                PROF_LAP(5);
        }
        PROF_FLUSH();
}
