// k_match.hpp — docset matching kernels: candidate tiles (galloping / block-driven) and dense bitmap windows
// Part of libtrinity_hip.so (MI355X / gfx950); included by trinity_hip.hip.  New code, no reference source.
#pragma once
#include "codec_streams.hpp"

// ------------------------------------------------------------------------------------------ k_and
constexpr int AND_WG = 256;   // candidate-tile kernel (k_and) and the scoring kernels
constexpr int DENSE_WG = 512; // bitmap-window kernel (k_and_dense): 8 waves share one 38 KB window state
static_assert(TILE_BLOCKS == AND_WG, "the candidate kernel maps one 32-candidate row to each of its 256 lanes"); // (TILE_BLOCKS, TILE_CANDS: dev_structs.hpp)

// Values that are workgroup-uniform by construction but read back from LDS look divergent to the compiler; a
// loop whose exit depends on one gets exec-masked structurisation, which is fatal around s_barrier (lanes
// "leave" the loop at different times).  uni() pins such values into an SGPR so the branch is scalar.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// LDS candidate layout: logical slot j lives at phys(j); rotating each 32-slot row by its row number keeps
// the one-lane-per-row writes of the lead decode (lane t writes row t, column i) off a single bank.
__device__ __forceinline__ uint32_t phys(uint32_t j) { return (j & ~31u) | ((j + (j >> 5)) & 31u); }

// Exclusive prefix sum across the wave's 64 lanes (every lane active) and the wave's total.  Six DPP adds — within each row of 16 lanes by row_shr:1 / 2 / 4 / 8, then
// row_bcast:15 carries rows 0 / 2's sums into rows 1 / 3 and row_bcast:31 the lower half's into the upper — and a v_readlane: no LDS crossbar trips (until round 6 this was
// six __shfl_up steps: a ds_bpermute_b32 and an s_waitcnt lgkmcnt(0) each, a chain of seven LDS round trips per scan in k_psets' / k_and's inner loops).
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total) {
        uint32_t x = v;
        // (written out: left to the compiler, an update_dpp + add pair became v_mov 0 / v_mov_dpp / v_add — eighteen instructions — where registers were tight.  A VALU
        //  result read by a DPP instruction wants two wait states: the s_nop 1 between them; the ones at the ends stand for what the compiler cannot see into)
        asm volatile("s_nop 1\n\t"
                     "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "s_nop 1\n\t"
                     "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "s_nop 1\n\t"
                     "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "s_nop 1\n\t"
                     "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "s_nop 1\n\t"
                     "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                     "s_nop 1"
                     : "+v"(x));
        total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
        return x - v;
}

// (SPAN_BITS, SPAN_WORDS, CELL_LOG2, CELL_DOCS, CELLS_PER_SPAN: dev_structs.hpp — the host planner cuts tasks by them)
// Window bitmap layout in LDS: logical word w lives at bm[w + (w >> 5)] — every row of 32 words is followed by one pad
// word, so lanes whose blocks lie a small constant number of words apart do not pile onto one bank.  Bitmap B is bitmap A
// shifted by BM_B_WORDS logical words, i.e. a lane selects it by adding BM_B_WORDS * 32 to its window-relative docID once
// per block; the per-posting address is then two shifts and an add.  Logical word SPAN_WORDS (first of the spare row) is
// the sink for documents outside the window.
constexpr uint32_t BM_B_WORDS = SPAN_WORDS + 32;
constexpr uint32_t BM_STRIDE = BM_B_WORDS + BM_B_WORDS / 32; // physical words per bitmap == pad(BM_B_WORDS)
__device__ __forceinline__ uint32_t bm_pad(const uint32_t w) { return w + (w >> 5); }

// IndexSourceTermsScorer::score(id, freq, weight) of the three scorers of similarity.h — BM25 :228-235
// float(idf * float(f) / double(f + 1.2f)); TF-IDF :92-94, :133-138 float(sqrt(float f) * weight); Trivial :64-66 f.
// freq is what PostingsListIterator::freq / Phrase::matchCnt expose: tokenpos_t, 16 bits (codecs.h:217).
__device__ __forceinline__ float sim_score(const int sim, const double weight, const uint32_t freq32) {
        const float f = (float)(uint16_t)freq32;
        if (sim == TRI_SIM_TFIDF)
                return (float)((double)sqrtf(f) * weight);
        if (sim == TRI_SIM_TRIVIAL)
                return f;
        return (float)(weight * (double)f / (double)(f + 1.2f));
}

// LDS state of the candidate-tile kernel
struct AndShared {
        uint32_t cand[TILE_CANDS];
        uint32_t hit[TILE_BLOCKS]; // bit k of hit[r] <=> logical candidate r*32+k matched
        uint32_t scan[8];
        uint32_t bcast[4];
        uint32_t lcur[16]; // per term: directory cursor, uniform across the workgroup
};

// LDS state of the bitmap-window kernel
struct DenseShared {
        uint32_t bm[2 * BM_STRIDE]; // two docID-window bitmaps A, B (rows padded, one spare row holding the sink word)
        uint32_t tbase[DENSE_WG]; // expansion: per-thread output base; decode passes: the deferred (slow) block list
        uint32_t scan[8];
        uint32_t bcast[4];
        uint32_t lcur[16];
        uint32_t seg_lo[MAX_QTERMS];      // per query term: first block reaching the current window ...
        uint32_t seg_cnt[MAX_QTERMS + 1]; // ... and how many do (+ sentinel)
        DevTerm seg_term[MAX_QTERMS];     // the query's terms (staged once per task)
        uint32_t seg_tt[MAX_QTERMS];      // ... and their qterms[] words (term id | QT_GROUP)
        uint32_t seg_wlo[MAX_QTERMS];     // win[w], win[w + 1] of every indexed term for the current window
        uint32_t seg_whi[MAX_QTERMS];
        uint32_t seg_plane[MAX_QTERMS];   // the term's row in the batch's term planes (PL_NONE: its rows are decoded)
        uint32_t nslow;
};
constexpr uint32_t DENSE_SLOW_CAP = DENSE_WG / 4; // 16-byte entries in tbase[]

// Workgroup-cooperative lower bound over a sorted global array: first i in [0, n) with a[i] >= key, else n.
// 256-ary search: every lane probes the end of its segment, one ballot per wave finds the first segment whose
// last element is >= key; ~log256(n) rounds of one (L2-resident) load each instead of log2(n) dependent loads.
template <int WG = AND_WG>
__device__ uint32_t wg_lower_bound(uint32_t *scan, const uint32_t *__restrict__ a, const uint32_t n, const uint32_t key) {
        const uint32_t tid = threadIdx.x;
        uint32_t lo = 0, hi = n; // answer in [lo, hi]
        while (hi > lo) {
                const uint32_t len = hi - lo;
                const uint32_t step = (len + WG - 1) / WG;
                const uint32_t pos = lo + (tid + 1) * step - 1;
                const bool ge = pos >= hi ? true : a[pos] >= key;
                const uint64_t m = __ballot(ge);
                scan[tid >> 6] = m ? (tid & ~63u) + (uint32_t)__builtin_ctzll(m) : 0xffffffffu;
                __syncthreads();
                uint32_t first = 0xffffffffu;
#pragma unroll
                for (int wv = 0; wv < WG / 64; ++wv)
                        first = min(first, scan[wv]);
                first = uni(first);
                __syncthreads(); // (before the early return too: the caller's next store into scan[] must not pass a slower wave's reads above)
                if (first == 0xffffffffu) // every probe < key (the last probe sat exactly on a[hi - 1]): nothing >= key
                        return hi;
                const uint32_t nlo = lo + first * step;
                const uint32_t nhi = min(hi, lo + (first + 1) * step - 1);
                lo = nlo;
                hi = step == 1 ? nlo : nhi;
        }
        return lo;
}

// The first 32 payload bytes of a block from three wide loads at the enclosing dword boundary, realigned in registers
// (v_alignbyte).  Returns true when bytes 0..30 — the 31 doc deltas of a full block — are all one-byte varints (every
// block of a head term), in which case delta j is byte j of v[].  One wave-load touches 64 cache lines whatever its width,
// so 3 wide loads per block instead of ~13 eight-byte stream loads takes the pressure off the L1/TA path; the freqs and
// hits behind the deltas are never touched in DocumentsOnly mode.  The pointer is derived by arithmetic (not through an
// integer) so the loads stay global_load (a flat load would also tie up the LDS counter the bitmap atomics use).
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
// (issue / finish are separate so that a caller can put other work — an LDS search — between the loads and their first use)
__device__ __forceinline__ void block_bytes32_issue(const uint8_t *__restrict__ p, uint32_t (&r)[9]) {
        const uint8_t *q = p - ((uintptr_t)p & 3u);
        const u32x4_a4 A = *(const u32x4_a4 *)q, B = *(const u32x4_a4 *)(q + 16);
        r[0] = A.x, r[1] = A.y, r[2] = A.z, r[3] = A.w, r[4] = B.x, r[5] = B.y, r[6] = B.z, r[7] = B.w;
        r[8] = *(const uint32_t *)(q + 32);
}
__device__ __forceinline__ bool block_bytes32_finish(const uint8_t *__restrict__ p, const uint32_t (&r)[9], uint32_t (&v)[8]) {
        const uint32_t sk = (uint32_t)((uintptr_t)p & 3u);
#pragma unroll
        for (int i = 0; i < 8; ++i)
                v[i] = __builtin_amdgcn_alignbyte(r[i + 1], r[i], sk);
        return ((v[0] | v[1] | v[2] | v[3] | v[4] | v[5] | v[6] | (v[7] & 0x00ffffffu)) & 0x80808080u) == 0;
}
__device__ __forceinline__ bool load_block_bytes32(const uint8_t *__restrict__ p, uint32_t (&v)[8]) {
        uint32_t r[9];
        block_bytes32_issue(p, r);
        return block_bytes32_finish(p, r, v);
}

// The first candidate at or behind `ptr` that is >= doc (C: none), found by GALLOPING from ptr: a block of a rare term spans the docID range of
// thousands of candidates, and the lane that merges it must not walk them one by one (measured: a 21-document list against a tile of 8192
// candidates took 680 us of single-lane LDS reads — cfg3's k_and spent half its span in a handful of such tiles)
__device__ __forceinline__ uint32_t cand_gallop(const AndShared &sh, uint32_t ptr, const uint32_t C, const uint32_t doc) {
        uint32_t step = 1, lo = ptr;
        while (lo + step <= C && sh.cand[phys(lo + step - 1)] < doc) {
                lo += step;
                step <<= 1;
        }
        uint32_t hi = min(lo + step - 1, C);
        while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (sh.cand[phys(mid)] < doc)
                        lo = mid + 1;
                else
                        hi = mid;
        }
        return lo;
}

// Merge one block of term t against the candidates from `ptr` on (cv = candidate at ptr): set the hit bit of every
// candidate that is a document of the block.  Full blocks of one-byte deltas take the register path above.
// (PRE: the caller has already issued the payload loads into raw[] — by reference, so the array stays in registers)
template <int CODEC, bool PRE>
__device__ __forceinline__ void merge_block(AndShared &sh, const uint8_t *__restrict__ index, const DevTerm &t, const uint32_t b, const uint32_t off,
                                            const uint32_t n, const uint32_t prev, const uint32_t last, uint32_t ptr, uint32_t cv, const uint32_t C,
                                            const uint32_t (&raw)[9]) {
        uint32_t doc = prev;
        uint32_t v[8];
        if (CODEC == CODEC_GOOGLE && n == 32 && (PRE ? block_bytes32_finish(index + off, raw, v) : load_block_bytes32(index + off, v))) {
                // the (at most 8) candidates that can fall into this block, fetched from LDS in one go and kept in registers as a
                // shift queue: the per-document step is then an add and a compare, with no LDS round trip in the lane's chain
                uint32_t c[8];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                        c[i] = ptr + i < C ? sh.cand[phys(ptr + i)] : 0xffffffffu;
                if (c[7] > last) {
                        uint32_t at = ptr;
#pragma unroll
                        for (int j = 0; j < 32; ++j) {
                                doc = j < 31 ? doc + ((v[j >> 2] >> ((j & 3) * 8)) & 0xffu) : last;
                                while (c[0] <= doc) {
                                        if (c[0] == doc)
                                                atomicOr(&sh.hit[at >> 5], 1u << (at & 31));
                                        ++at;
#pragma unroll
                                        for (int i = 0; i < 7; ++i)
                                                c[i] = c[i + 1];
                                        c[7] = 0xffffffffu;
                                }
                                if (c[0] > last)
                                        return;
                        }
                        return;
                }
                // nine or more candidates in one block's range: walk them through LDS
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                        doc = j < 31 ? doc + ((v[j >> 2] >> ((j & 3) * 8)) & 0xffu) : last;
                        if (cv <= doc) {
                                if (cv == doc)
                                        atomicOr(&sh.hit[ptr >> 5], 1u << (ptr & 31));
                                ptr = cand_gallop(sh, ptr + 1, C, doc); // (usually the next candidate or the one after)
                                cv = ptr < C ? sh.cand[phys(ptr)] : 0xffffffffu;
                                if (cv == doc) // (only after skipping candidates that were below doc)
                                        atomicOr(&sh.hit[ptr >> 5], 1u << (ptr & 31));
                                if (cv > last)
                                        return;
                        }
                }
                return;
        }
        DeltaStream<CODEC> s;
        s.init(index, t, b, off);
        for (uint32_t i = 0; i < n; ++i) {
                doc = (i + 1 < n) ? doc + s.next() : last;
                if (cv < doc) {
                        ptr = cand_gallop(sh, ptr + 1, C, doc);
                        cv = ptr < C ? sh.cand[phys(ptr)] : 0xffffffffu;
                }
                if (cv == doc)
                        atomicOr(&sh.hit[ptr >> 5], 1u << (ptr & 31));
                if (cv > last)
                        break;
        }
}

// Wave-cooperative lower bound over a[lo, hi): first i with a[i] >= key (key wave-uniform), hi when none.  64-ary search
// with one ballot per round — no LDS, no barrier; every active lane gets the same answer.
__device__ __forceinline__ uint32_t wave_lower_bound(const uint32_t *__restrict__ a, uint32_t lo, uint32_t hi, const uint32_t key) {
        const uint32_t lane = threadIdx.x & 63u;
        while (hi > lo) {
                const uint32_t len = hi - lo;
                const uint32_t step = (len + 63u) / 64u;
                const uint32_t pos = lo + (lane + 1) * step - 1;
                const bool ge = pos >= hi ? true : a[pos] >= key;
                const uint64_t m = __ballot(ge);
                if (m == 0)
                        return hi;
                const uint32_t first = (uint32_t)__builtin_ctzll(m);
                const uint32_t nlo = lo + first * step;
                const uint32_t nhi = min(hi, lo + (first + 1) * step - 1);
                lo = nlo;
                hi = step == 1 ? nlo : nhi;
        }
        return lo;
}

// Filter the C candidates in sh.cand (logical order ascending) against term `t`: sets sh.hit bits.
// Caller syncs before and after.
template <int CODEC>
__device__ void and_filter_tile(AndShared &sh, const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last,
                                const uint32_t *__restrict__ blk_off, const uint32_t *__restrict__ win, const DevTerm t, const uint32_t C,
                                const uint32_t lcur_slot, const bool block_driven PROF_ARG) {
        const uint32_t tid = threadIdx.x;
        const uint32_t *bl = blk_last + t.first_block;
        const uint32_t *bo = blk_off + t.first_block;
        const uint32_t cmin = sh.cand[phys(0)], cmax = sh.cand[phys(C - 1)];

        if (block_driven) {
                // advance lcur to the first block whose last docID >= cmin (tiles arrive in ascending docID order)
                uint32_t lcur = uni(sh.lcur[lcur_slot]);
                if (lcur == 0xffffffffu) // first tile of this task: position by cooperative search, then gallop forward
                        lcur = wg_lower_bound(sh.scan, bl, t.nblocks, cmin);
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
                PROF_LAP(5);
                TRACE(10, lcur, t.nblocks);
                // directory entries are fetched one round ahead and the block's payload loads are issued before the LDS search, so
                // a round costs about one memory round trip instead of three dependent ones
                uint32_t nprev = 0, nlast = 0, noff = 0;
                {
                        const uint32_t b = lcur + tid;
                        if (b < t.nblocks) {
                                nprev = b ? bl[b - 1] : 0;
                                nlast = bl[b];
                                noff = bo[b];
                        }
                }
                for (uint32_t cb = lcur; cb < t.nblocks; cb += AND_WG) {
                        TRACE(11, cb, t.nblocks);
                        const uint32_t b = cb + tid;
                        bool beyond = true;
                        const uint32_t prev = nprev, last = nlast, off = noff; // docs of block b lie in (prev, last]
                        uint32_t raw[9];
                        const bool full = CODEC == CODEC_GOOGLE && b < t.nblocks && prev < cmax && TRI_BLOCK_N(t, b, index, off) == 32;
                        if (full)
                                block_bytes32_issue(index + off, raw);
                        {
                                const uint32_t b2 = b + AND_WG;
                                if (b2 < t.nblocks) {
                                        nprev = bl[b2 - 1];
                                        nlast = bl[b2];
                                        noff = bo[b2];
                                }
                        }
                        if (b < t.nblocks) {
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
                                        PROF_LAP(6);
                                        if (cv <= last) {
                                                if (full)
                                                        merge_block<CODEC, true>(sh, index, t, b, off, 32, prev, last, ptr, cv, C, raw);
                                                else
                                                        merge_block<CODEC, false>(sh, index, t, b, off, TRI_BLOCK_N(t, b, index, off), prev, last, ptr, cv, C, raw);
                                        }
                                        PROF_LAP(7);
                                }
                        }
                        // workgroup-wide OR of `beyond`, branch-free: one ballot per wave, four LDS words
                        sh.scan[4 + (tid >> 6)] = __ballot(beyond) != 0ull;
                        __syncthreads();
                        const uint32_t any_beyond = uni(sh.scan[4] | sh.scan[5] | sh.scan[6] | sh.scan[7]);
                        __syncthreads();
                        PROF_LAP(8);
                        if (any_beyond)
                                break;
                }
        } else {
                // candidate-driven galloping: each candidate finds its block in the directory; the first
                // candidate of each run that maps to the same block decodes it and merges forward.  Waves run
                // independently here (no workgroup barrier inside the loop): a run is recognised inside the wave by a
                // shuffle, and a block that two waves both start on is simply merged twice (hit bits are idempotent).
                uint32_t carry = 0xffffffffu; // block of the last candidate this wave looked at
                uint32_t wcur = 0; // this wave's directory cursor: its candidates only move forward from round to round
                for (uint32_t base = 0; base < C; base += AND_WG) {
                        TRACE(20, base, C);
                        const uint32_t j = base + tid;
                        uint32_t bj = 0xffffffffu;
                        uint32_t cv = 0;
                        // hierarchical narrowing: the wave's 64 ascending candidates bracket a directory range with two
                        // cooperative 64-ary searches; each lane then bisects only inside that (cache-resident) range
                        const uint32_t wbase = base + (tid & ~63u);
                        uint32_t rlo = 0, rhi = 0;
                        if (t.win_off != 0xffffffffu) {
                                // indexed list: the candidate's docID cell brackets its block with one independent load pair
                                if (j < C) {
                                        const uint32_t c = sh.cand[phys(j)] >> CELL_LOG2;
                                        rlo = win[t.win_off + c];
                                        rhi = win[t.win_off + c + 1];
                                }
                        } else if (wbase < C) { // wave-uniform
                                const uint32_t nval = min(64u, C - wbase);
                                const uint32_t klo = sh.cand[phys(wbase)], khi = sh.cand[phys(wbase + nval - 1)];
                                rlo = wave_lower_bound(bl, wcur, t.nblocks, klo);
                                rhi = wave_lower_bound(bl, rlo, t.nblocks, khi);
                                wcur = rlo;
                        }
                        if (j < C) {
                                cv = sh.cand[phys(j)];
                                uint32_t lo = rlo, hi = rhi; // first block with last >= cv lies in [rlo, rhi]
                                while (lo < hi) {
                                        const uint32_t mid = (lo + hi) >> 1;
                                        if (bl[mid] < cv)
                                                lo = mid + 1;
                                        else
                                                hi = mid;
                                }
                                bj = lo; // == nblocks: beyond the list
                        }
                        uint32_t prevb = __shfl_up(bj, 1, 64);
                        if ((tid & 63u) == 0)
                                prevb = carry;
                        carry = __shfl(bj, 63, 64);
                        if (j < C && bj < t.nblocks && bj != prevb) {
                                const uint32_t prev = bj ? bl[bj - 1] : 0;
                                const uint32_t none[9] = {};
                                merge_block<CODEC, false>(sh, index, t, bj, bo[bj], TRI_BLOCK_N(t, bj, index, bo[bj]), prev, bl[bj], j, cv, C, none);
                        }
                }
        }
}

// ---- TASK_DENSE: bitmap algebra over docID windows -------------------------------------------------------
// One lane decodes one block (unpack_block, google_codec.cpp:596-639) and sets its documents' bits in a window bitmap in
// LDS; groups are combined by word-wise AND / AND-NOT of whole bitmaps; the survivors are expanded to ascending docIDs.
// One posting into a window bitmap.  `rel` = docID - window start (+ BM_B_WORDS * 32 for bitmap B).  Fire-and-forget:
// no value comes back from LDS, so nothing in the lane's chain waits on it.  Every term only SETS bits; a conjunct is
// folded in by AND-ing whole bitmaps afterwards (dense_task), 8 words per thread instead of a dependent LDS read per
// posting.  dense_visit: the block lies wholly inside the window.  dense_visit_clamped: it may not — documents outside go
// to the sink word (documents below the window wrap to huge values).
__device__ __forceinline__ void dense_visit(uint32_t *bm, const uint32_t rel) {
        // byte address of word pad(rel >> 5) in three instructions (the compiler's own form — two shifts, two masks, an add —
        // takes five, and this is the innermost statement of the engine)
        uint32_t w, r, a;
        asm("v_lshrrev_b32 %0, 5, %1" : "=v"(w) : "v"(rel));
        asm("v_lshrrev_b32 %0, 10, %1" : "=v"(r) : "v"(rel));
        asm("v_add_lshl_u32 %0, %1, %2, 2" : "=v"(a) : "v"(w), "v"(r));
        atomicOr((uint32_t *)((uint8_t *)bm + a), 1u << (rel & 31u));
}
__device__ __forceinline__ void dense_visit_clamped(uint32_t *bm, const uint32_t rel, const uint32_t wbase) {
        atomicOr(&bm[bm_pad(min(rel >> 5, SPAN_WORDS) + wbase)], 1u << (rel & 31u));
}

// Generic block walk over the per-lane byte stream (any varint lengths, any n, any position relative to the window).
template <int CODEC>
__device__ __forceinline__ void dense_block_stream(const uint8_t *__restrict__ index, const DevTerm &t, const uint32_t b, const uint32_t off,
                                                   const uint32_t n, const uint32_t prev, const uint32_t last, const uint32_t w0, uint32_t *bm,
                                                   const uint32_t wbase) {
        uint32_t rel = prev - w0;
        const uint32_t nd = n - 1;
        if (CODEC != CODEC_GOOGLE) {
                DeltaStream<CODEC> ls;
                ls.init(index, t, b, off);
                for (uint32_t i = 0; i < nd; ++i) {
                        rel += ls.next();
                        dense_visit_clamped(bm, rel, wbase);
                }
                dense_visit_clamped(bm, last - w0, wbase);
                return;
        }
        VbStream s;
        s.init(index + off);
        uint32_t i = 0;
        while (i < nd) {
                s.refill();
                const uint32_t k = min(8u, nd - i);
                if (s.small_run(k)) { // k one-byte deltas: no per-value length decode
                        uint64_t w = s.take(k);
#pragma unroll
                        for (uint32_t j = 0; j < 8; ++j) {
                                if (j < k) {
                                        rel += (uint32_t)(w & 0xffu);
                                        w >>= 8;
                                        dense_visit_clamped(bm, rel, wbase);
                                }
                        }
                        i += k;
                } else {
                        rel += s.next();
                        dense_visit_clamped(bm, rel, wbase);
                        ++i;
                }
        }
        dense_visit_clamped(bm, last - w0, wbase);
}

// GOOGLE blocks that miss the static path (a multi-byte delta somewhere, a short last block, a block straddling the
// window's end): the first 60 payload bytes come from four wide loads issued together — one memory round trip, like the
// static path — and the varints are parsed out of registers: the outer loop over the loaded dwords is static, the inner
// loop takes every varint that is complete in the 64-bit window.  (The byte stream's refills are dependent loads; walking
// a block through it costs several round trips.)  More than 60 bytes of deltas: the stream finishes the block.
__device__ __forceinline__ void dense_block_regs(const uint8_t *__restrict__ index, const uint32_t off, const uint32_t n, const uint32_t prev,
                                                 const uint32_t last, const uint32_t w0, uint32_t *bm, const uint32_t wbase) {
        const uint8_t *p = index + off;
        const uint32_t sk = (uint32_t)((uintptr_t)p & 3u);
        const uint8_t *q = p - sk;
        const u32x4_a4 A = *(const u32x4_a4 *)q, B = *(const u32x4_a4 *)(q + 16), C = *(const u32x4_a4 *)(q + 32), D = *(const u32x4_a4 *)(q + 48);
        const uint32_t raw[16] = {A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w, C.x, C.y, C.z, C.w, D.x, D.y, D.z, D.w};
        uint64_t win = 0;
        uint32_t have = 0, done = 0, rel = prev - w0;
        const uint32_t nd = n - 1;
#pragma unroll
        for (int d = 0; d < 15; ++d) {
                win |= (uint64_t)__builtin_amdgcn_alignbyte(raw[d + 1], raw[d], sk) << (have * 8); // have <= 4 here
                have += 4;
                while (done < nd) {
                        uint32_t len;
                        const uint32_t v = vb_decode(win, len);
                        if (len > have)
                                break;
                        win >>= 8 * len;
                        have -= len;
                        rel += v;
                        dense_visit_clamped(bm, rel, wbase);
                        ++done;
                }
        }
        if (done < nd) {
                VbStream st;
                st.init(p + (60 - have));
                for (; done < nd; ++done) {
                        rel += st.next();
                        dense_visit_clamped(bm, rel, wbase);
                }
        }
        dense_visit_clamped(bm, last - w0, wbase);
}

// One deferred GOOGLE block decoded by a whole wave (arguments wave-uniform): lane i looks at payload byte i as if a varint
// started there; the true starts are found by walking the lengths (a scalar loop of readlanes, <= 31 steps), the deltas of
// the start lanes are prefix-summed across the wave, and every start lane sets its bit.  ~200 instructions of latency
// instead of the ~800 of a lane parsing the block alone — this is what runs while the rest of the workgroup waits at
// the barrier, so latency is what counts.
__device__ __forceinline__ void dense_block_coop(const uint8_t *__restrict__ index, const uint32_t off, const uint32_t n, const uint32_t prev,
                                                 const uint32_t last, const uint32_t w0, uint32_t *bm, const uint32_t wbase) {
        const uint32_t lane = threadIdx.x & 63u;
        const uint8_t *p = index + off + lane;
        const uint32_t sk = (uint32_t)((uintptr_t)p & 3u);
        const uint32_t *q = (const uint32_t *)(p - sk);
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
        const uint64_t w = (uint64_t)__builtin_amdgcn_alignbyte(d1, d0, sk) | ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, sk) << 32);
        uint32_t len;
        const uint32_t v = vb_decode(w, len);
        const uint32_t nd = n - 1;
        uint64_t starts = 0;
        uint32_t s = 0, found = 0;
        for (; found < nd && s < 64; ++found) {
                starts |= 1ull << s;
                s += (uint32_t)__builtin_amdgcn_readlane((int)len, (int)s);
        }
        if (found < nd) { // > 64 bytes of deltas (rare): one lane parses the block alone; setting a bit twice is harmless
                if (lane == 0)
                        dense_block_regs(index, off, n, prev, last, w0, bm, wbase);
                return;
        }
        const bool is_start = (starts >> lane) & 1ull;
        uint32_t x = is_start ? v : 0u, wtot;
        x += wave_excl_scan(x, wtot); // (inclusive: six DPP adds)
        if (is_start)
                dense_visit_clamped(bm, prev - w0 + x, wbase);
        if (lane == 0)
                dense_visit_clamped(bm, last - w0, wbase);
}

// One pass over a window: the blocks of terms [kbeg, kend) that reach the window form one virtual work list (term after
// term), dealt out to the lanes round by round — so a short list does not leave most of the workgroup idle and the terms
// of a pass share one barrier.  Terms below ksplit set bits in A, the others in B.  GOOGLE blocks take one of:
//   static   a full block of one-byte deltas (every block of a head term): 31 unrolled byte adds from registers;
//            the block straddling a window end runs the same code with the clamped visit
//   regs     terms flagged TERM_SPARSE (most blocks hold a multi-byte delta): varints parsed from registers, in place
//   deferred the odd block of a dense term (a multi-byte delta, a short last block): decoding it in place would make its
//            whole wave run the slow path, so it is appended to an LDS list that the waves then work off cooperatively
template <int WG, int CODEC>
__device__ __forceinline__ void dense_pass(DenseShared &sh, const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last,
                                           const uint32_t *__restrict__ blk_off, const uint32_t kbeg, const uint32_t kend, const uint32_t ksplit,
                                           const uint32_t w0, const uint32_t *__restrict__ planes, const uint32_t plw PROF_ARG) {
        const uint32_t tid = threadIdx.x;
        uint32_t total = 0;
        for (uint32_t k = kbeg; k < kend; ++k) {
                total += uni(sh.seg_cnt[k]);
                // a term with a plane (decoded once per launch by k_term_planes): its documents of this window are SPAN_WORDS words
                // of plane A, OR-ed into the group's bitmap — 8 coalesced loads per thread instead of a walk over the term's rows
                const uint32_t prow = uni(sh.seg_plane[k]);
                if (prow != PL_NONE) {
                        const uint32_t *pa = planes + (size_t)prow * plw + (w0 >> 5);
                        const uint32_t wbase = k < ksplit ? 0u : BM_B_WORDS;
#pragma unroll
                        for (uint32_t j = 0; j < SPAN_WORDS / WG; ++j) {
                                const uint32_t i = j * WG + tid;
                                const uint32_t v = pa[i];
                                if (v)
                                        atomicOr(&sh.bm[bm_pad(i + wbase)], v);
                        }
                }
        }
        for (uint32_t v0 = 0; v0 < total; v0 += WG) {
                const uint32_t v = v0 + tid;
                if (v < total) {
                        uint32_t k = kbeg, r = v;
                        for (uint32_t c = sh.seg_cnt[k]; r >= c; c = sh.seg_cnt[k]) {
                                r -= c;
                                ++k;
                        }
                        const DevTerm t = sh.seg_term[k];
                        const uint32_t b = sh.seg_lo[k] + r;
                        const uint32_t *bl = blk_last + t.first_block;
                        const uint32_t prev = b ? bl[b - 1] : 0;
                        const uint32_t last = bl[b];
                        const uint32_t off = blk_off[t.first_block + b];
                        const uint32_t n = TRI_BLOCK_N(t, b, index, off);
                        const uint32_t wbase = k < ksplit ? 0u : BM_B_WORDS;
                        if (CODEC != CODEC_GOOGLE)
                                dense_block_stream<CODEC>(index, t, b, off, n, prev, last, w0, sh.bm, wbase);
                        else if (t.flags & TERM_SPARSE)
                                dense_block_regs(index, off, n, prev, last, w0, sh.bm, wbase);
                        else {
                                uint32_t dv[8];
                                const bool fits = n == 32 && load_block_bytes32(index + off, dv);
                                const bool inside = prev + 1 >= w0 && last - w0 < SPAN_BITS;
                                if (fits && inside) {
                                        uint32_t rel = prev - w0 + wbase * 32u;
#pragma unroll
                                        for (int j = 0; j < 31; ++j) {
                                                rel += (dv[j >> 2] >> ((j & 3) * 8)) & 0xffu;
                                                dense_visit(sh.bm, rel);
                                        }
                                        dense_visit(sh.bm, last - w0 + wbase * 32u);
                                } else if (fits) {
                                        uint32_t rel = prev - w0;
#pragma unroll
                                        for (int j = 0; j < 31; ++j) {
                                                rel += (dv[j >> 2] >> ((j & 3) * 8)) & 0xffu;
                                                dense_visit_clamped(sh.bm, rel, wbase);
                                        }
                                        dense_visit_clamped(sh.bm, last - w0, wbase);
                                } else {
                                        const uint32_t slot = atomicAdd(&sh.nslow, 1u);
                                        if (slot < DENSE_SLOW_CAP) {
                                                sh.tbase[4 * slot + 0] = off;
                                                sh.tbase[4 * slot + 1] = prev;
                                                sh.tbase[4 * slot + 2] = last;
                                                sh.tbase[4 * slot + 3] = n | (wbase ? 0x100u : 0u);
                                        } else
                                                dense_block_regs(index, off, n, prev, last, w0, sh.bm, wbase);
                                }
                        }
                }
        }
        PROF_LAP(3);
        if (CODEC == CODEC_GOOGLE) {
                __syncthreads();
                PROF_LAP(4);
                const uint32_t ns = min(uni(sh.nslow), DENSE_SLOW_CAP);
                for (uint32_t i = uni(tid >> 6); i < ns; i += WG / 64) {
                        const uint32_t off = uni(sh.tbase[4 * i + 0]), prev = uni(sh.tbase[4 * i + 1]), last = uni(sh.tbase[4 * i + 2]);
                        const uint32_t nw = uni(sh.tbase[4 * i + 3]);
                        dense_block_coop(index, off, nw & 0xffu, prev, last, w0, sh.bm, (nw & 0x100u) ? BM_B_WORDS : 0u);
                }
                if (ns) { // uniform
                        __syncthreads();
                        sh.nslow = 0; // uniform store; the next pass's appends come after at least one more barrier
                }
        }
        __syncthreads();
        PROF_LAP(5);
}

// The window's survivors bitmap (A, with the last group still in B when `fold`: ANDed in, or removed when `neg`) expanded into
// ascending docIDs at qout[produced ..]; both bitmaps are left zeroed.  Returns the window's match count (uniform).
template <int WG>
__device__ __forceinline__ uint32_t dense_expand(DenseShared &sh, const uint32_t w0, const bool fold, const bool neg, const uint32_t *__restrict__ masked,
                                                 uint32_t *__restrict__ qout, const uint32_t produced PROF_ARG) {
        const uint32_t tid = threadIdx.x;
        uint32_t window_total = 0;
        // ---- expand the survivors bitmap into ascending docIDs
        uint32_t *fin = sh.bm;
        uint32_t *pre = sh.bm + BM_STRIDE; // B dies word by word as it is folded in: per-word exclusive prefix takes its place
        {
                uint32_t run = 0;
                for (uint32_t j = 0; j < SPAN_WORDS / WG; ++j) {
                        const uint32_t wi = bm_pad(tid * (SPAN_WORDS / WG) + j);
                        uint32_t m = fin[wi];
                        if (fold)
                                m &= neg ? ~pre[wi] : pre[wi]; // the last group is the one still in B; an excluded group removes
                        if (masked) // masked_documents_registry::test (docidupdates.h:90-119): updated / deleted elsewhere
                                m &= ~masked[w0 / 32 + tid * (SPAN_WORDS / WG) + j];
                        fin[wi] = m;
                        pre[wi] = run;
                        run += __popc(m);
                }
                uint32_t wtot;
                const uint32_t ex = wave_excl_scan(run, wtot);
                sh.scan[tid >> 6] = wtot;
                __syncthreads();
                uint32_t wbase = 0, total = 0;
                for (int wv = 0; wv < WG / 64; ++wv) {
                        if (wv < (int)(tid >> 6))
                                wbase += sh.scan[wv];
                        total += sh.scan[wv];
                }
                sh.tbase[tid] = ex + wbase;
                __syncthreads();
                PROF_LAP(6);
                // word-strided sweep: neighbouring lanes own neighbouring words, so a wave's stores stay together
                // (both bitmaps are left zeroed for the next window as they are read)
                if (uni(total) >= SPAN_BITS / 8) {
                        // dense result (a union of head terms): one lane per BIT.  Each wave walks its own 512 words, 64 bits at
                        // a time: ballot, rank by mbcnt, one coalesced store per step.  (Lane-per-word stores of a dense window
                        // hit 64 different cache lines per instruction.)
                        const uint32_t lane = tid & 63u, wv = tid >> 6;
                        uint32_t o = produced + sh.tbase[wv * 64]; // matches before this wave's first word
                        for (uint32_t c = 0; c < SPAN_WORDS / (WG / 64) / 2; ++c) {
                                const uint32_t wi = wv * (SPAN_WORDS / (WG / 64)) + 2 * c + (lane >> 5);
                                const bool bit = (fin[bm_pad(wi)] >> (lane & 31u)) & 1u;
                                const uint64_t m = __ballot(bit);
                                if (bit)
                                        qout[o + __popcll(m & ((1ull << lane) - 1ull))] = w0 + wi * 32 + (lane & 31u);
                                o += (uint32_t)__popcll(m);
                        }
                        __syncthreads();
                        for (uint32_t i = tid; i < 2 * BM_STRIDE; i += WG)
                                sh.bm[i] = 0;
                } else
                for (uint32_t wi = tid; wi < SPAN_WORDS; wi += WG) {
                        const uint32_t pw = bm_pad(wi);
                        uint32_t m = fin[pw];
                        uint32_t ob = (produced + sh.tbase[wi / (SPAN_WORDS / WG)] + pre[pw]) * 4u; // byte offset: uniform base + 32-bit lane offset
                        fin[pw] = 0;
                        pre[pw] = 0;
                        const uint32_t base = w0 + wi * 32;
                        while (m) {
                                *(uint32_t *)((uint8_t *)qout + ob) = base + (uint32_t)__builtin_ctz(m);
                                ob += 4;
                                m &= m - 1;
                        }
                }
                window_total = uni(total);
                __syncthreads();
                PROF_LAP(7);
        }
        return window_total;
}

// RESULT_BITMAP (dev_structs.hpp): the window's survivors bitmap written out AS IT IS — SPAN_WORDS words at wout, eight per thread in two 16-byte
// stores — instead of expanded; both LDS bitmaps are left zeroed.  Returns the window's match count (uniform).
template <int WG>
__device__ __forceinline__ uint32_t dense_store(DenseShared &sh, const uint32_t w0, const bool fold, const bool neg, const uint32_t *__restrict__ masked,
                                                uint32_t *__restrict__ wout) {
        constexpr uint32_t PER = SPAN_WORDS / WG;
        static_assert(PER == 8, "two 16-byte stores per thread");
        const uint32_t tid = threadIdx.x;
        uint32_t *fin = sh.bm, *pre = sh.bm + BM_STRIDE;
        uint32_t m[PER], run = 0;
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
                const uint32_t wi = bm_pad(tid * PER + j);
                m[j] = fin[wi];
                if (fold)
                        m[j] &= neg ? ~pre[wi] : pre[wi];
                if (masked) // masked_documents_registry::test (docidupdates.h:90-119)
                        m[j] &= ~masked[w0 / 32 + tid * PER + j];
                fin[wi] = 0;
                pre[wi] = 0;
                run += __popc(m[j]);
        }
        uint4 *o = (uint4 *)(wout + tid * PER);
        o[0] = make_uint4(m[0], m[1], m[2], m[3]);
        o[1] = make_uint4(m[4], m[5], m[6], m[7]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
                run += __shfl_xor(run, d, 64);
        sh.scan[tid >> 6] = run;
        __syncthreads();
        uint32_t total = 0;
        for (int wv = 0; wv < WG / 64; ++wv)
                total += sh.scan[wv];
        total = uni(total);
        __syncthreads();
        return total;
}
// ... a window that cannot hold a match (the lead group skipped it, a conjunct is exhausted): all zero
template <int WG>
__device__ __forceinline__ void dense_store_zero(uint32_t *__restrict__ wout) {
        uint4 *o = (uint4 *)(wout + threadIdx.x * (SPAN_WORDS / WG));
        o[0] = make_uint4(0, 0, 0, 0);
        o[1] = make_uint4(0, 0, 0, 0);
}

template <int WG, int CODEC>
__device__ __forceinline__ void dense_task(DenseShared &sh, const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last,
                           const uint32_t *__restrict__ blk_off, const uint32_t *__restrict__ win, const DevTerm *__restrict__ terms,
                           const uint32_t *__restrict__ qterms, const DevQuery q, const DevTask task, uint32_t *__restrict__ out,
                           uint32_t *__restrict__ count_out, const uint32_t *__restrict__ masked, const uint32_t *__restrict__ qplane,
                           const uint32_t *__restrict__ planes, const uint32_t plw PROF_ARG) {
        const uint32_t tid = threadIdx.x;
        uint32_t *qout = out + task.out_off;
        uint32_t produced = 0;
        const bool as_bitmap = uni(q.form) == RESULT_BITMAP; // the windows' words go out as they are: window w at qout + (w - tile_begin) * SPAN_WORDS
        uint32_t wdone = task.tile_begin;                    // ... the first window of the task not written yet
        sh.lcur[tid & 15] = 0; // per term: a block index at or before the first block that can matter
        sh.nslow = 0;
        for (uint32_t i = tid; i < 2 * BM_STRIDE; i += WG)
                sh.bm[i] = 0;
        __syncthreads();
        // ---- the query's terms, staged in LDS once per task (every lane runs the same code: lanes >= nterms repeat the last term)
        {
                const uint32_t kk = min(tid, q.nterms - 1);
                const uint32_t tt = qterms[q.term_base + kk];
                sh.seg_tt[kk] = tt;
                sh.seg_term[kk] = terms[tt & QT_TERM];
                sh.seg_plane[kk] = qplane ? qplane[q.term_base + kk] : PL_NONE;
        }
        __syncthreads();
        // ---- every term of the query has a plane (k_term_planes decoded the lists once for the whole batch): a window's matches are
        //      word-wise algebra over plane words held in registers — OR inside a group, AND across groups, AND-NOT for the excluded
        //      group — eight words per thread, no rows, no directory, no set pass; the expansion is the usual one
        bool allp = planes != nullptr;
        for (uint32_t k = 0; k < q.nterms; ++k)
                allp &= uni(sh.seg_plane[k]) != PL_NONE;
        if (allp) {
                constexpr uint32_t PER = SPAN_WORDS / WG;
                static_assert(PER == 8, "two 16-byte loads per thread and term");
                for (uint32_t w = task.tile_begin; w < task.tile_end; ++w) {
                        const uint32_t w0 = w * SPAN_BITS;
                        uint32_t acc[PER], grp[PER];
                        bool have_acc = false, cur_neg = false;
#pragma unroll
                        for (uint32_t j = 0; j < PER; ++j)
                                acc[j] = grp[j] = 0;
                        for (uint32_t k = 0; k <= q.nterms; ++k) {
                                const uint32_t tt = k < q.nterms ? uni(sh.seg_tt[k]) : QT_GROUP; // (k == nterms: the last group is folded in)
                                if (k && (tt & QT_GROUP)) {
#pragma unroll
                                        for (uint32_t j = 0; j < PER; ++j) {
                                                acc[j] = !have_acc ? grp[j] : cur_neg ? acc[j] & ~grp[j] : acc[j] & grp[j];
                                                grp[j] = 0;
                                        }
                                        have_acc = true;
                                }
                                if (k == q.nterms)
                                        break;
                                if (tt & QT_GROUP)
                                        cur_neg = tt & QT_NOT;
                                const uint4 *pa = (const uint4 *)(planes + (size_t)uni(sh.seg_plane[k]) * plw + (w0 >> 5) + tid * PER);
                                const uint4 v0 = pa[0], v1 = pa[1];
                                grp[0] |= v0.x, grp[1] |= v0.y, grp[2] |= v0.z, grp[3] |= v0.w;
                                grp[4] |= v1.x, grp[5] |= v1.y, grp[6] |= v1.z, grp[7] |= v1.w;
                        }
#pragma unroll
                        for (uint32_t j = 0; j < PER; ++j)
                                sh.bm[bm_pad(tid * PER + j)] = acc[j]; // (this thread's own words: dense_expand reads them back first)
                        if (as_bitmap)
                                produced += dense_store<WG>(sh, w0, false, false, masked, qout + (size_t)(w - task.tile_begin) * SPAN_WORDS);
                        else
                                produced += dense_expand<WG>(sh, w0, false, false, masked, qout, produced PROF_PASS);
                }
                __syncthreads();
                if (uni(tid >> 6) == 0)
                        *count_out = produced;
                PROF_LAP(8);
                return;
        }
        // number of terms in the lead group (it creates the candidates; the other groups test them)
        uint32_t nlead = 1;
        while (nlead < q.nterms && !(uni(sh.seg_tt[nlead]) & QT_GROUP))
                ++nlead;
        bool done = false; // uniform
        for (uint32_t w = task.tile_begin; w < task.tile_end && !done;) {
                // ---- window index entries of every (indexed) term for window w, fetched side by side: one memory round trip
                //      per window instead of one per term.  win[w] = first block with last >= w * SPAN_BITS.
                for (;;) {
                        {
                                const uint32_t kk = min(tid, q.nterms - 1);
                                const uint32_t wo = sh.seg_term[kk].win_off;
                                const uint32_t at = wo != 0xffffffffu ? wo + w * CELLS_PER_SPAN : 0u; // (a win[] row covers one window more than the corpus)
                                sh.seg_wlo[kk] = win[at];
                                sh.seg_whi[kk] = win[at + CELLS_PER_SPAN];
                        }
                        __syncthreads();
                        // skip windows no lead-group list reaches: look at the first document each may hold at or after w
                        uint32_t wnext = 0xffffffffu;
                        for (uint32_t k = 0; k < nlead; ++k) {
                                const DevTerm t = sh.seg_term[k];
                                const uint32_t *bl = blk_last + t.first_block;
                                uint32_t cur;
                                bool here = false; // the list surely holds a document of window w
                                if (uni(sh.seg_plane[k]) != PL_NONE) { // (a plane term is a head term: treated as present in every window)
                                        cur = 0;
                                        here = true;
                                } else if (uni(t.win_off) != 0xffffffffu) {
                                        cur = uni(sh.seg_wlo[k]);
                                        here = cur != uni(sh.seg_whi[k]); // a block ends inside the window
                                } else {
                                        // a list too short for a cell index (fewer than WIN_MIN_BLOCKS blocks): every wave finds the block by itself — two
                                        // rounds of 64 probes, no LDS, no barrier (a workgroup-wide search cost two barriers per list and window: most of the
                                        // 28 us a near-empty window of a union of rare terms took).  The cursor only moves forward and every wave stores the
                                        // same value: a wave that reads it late starts from the answer
                                        cur = uni(sh.lcur[k]);
                                        if (cur < uni(t.nblocks) && bl[cur] < w * SPAN_BITS)
                                                cur = uni(wave_lower_bound(bl, cur, t.nblocks, w * SPAN_BITS));
                                        sh.lcur[k] = cur;
                                }
                                if (here)
                                        wnext = w;
                                else if (cur < uni(t.nblocks)) {
                                        const uint32_t first_possible = cur ? bl[cur - 1] + 1 : 1;
                                        wnext = min(wnext, max(w, first_possible / SPAN_BITS));
                                }
                        }
                        wnext = uni(wnext);
                        __syncthreads(); // seg_wlo / seg_whi may be rewritten
                        if (wnext == w || wnext >= task.tile_end) {
                                w = wnext;
                                break;
                        }
                        w = wnext; // the lead jumped ahead: fetch that window's entries
                }
                if (w >= task.tile_end)
                        break; // the lead group holds nothing more in this task's range
                const uint32_t w0 = w * SPAN_BITS;
                const uint32_t wlast = w0 + (SPAN_BITS - 1);
                // ---- block range of every term in this window; a group none of whose lists reaches the window or beyond
                //      ends the task (an exhausted conjunct: no further match anywhere)
                uint32_t ngroups = 0, g1 = q.nterms, g2 = q.nterms; // first term of group 1 / group 2
                bool galive = false, neg = false;
                for (uint32_t k = 0; k < q.nterms; ++k) {
                        const uint32_t tt = uni(sh.seg_tt[k]);
                        const DevTerm t = sh.seg_term[k];
                        const uint32_t nblocks = uni(t.nblocks);
                        const uint32_t *bl = blk_last + t.first_block;
                        if (tt & QT_GROUP) {
                                if (k && !galive)
                                        done = true;
                                if (ngroups == 1)
                                        g1 = k;
                                else if (ngroups == 2)
                                        g2 = k;
                                ++ngroups;
                                galive = false;
                                neg = tt & QT_NOT; // the excluded group (always last): exhausted or not, it ends nothing
                        }
                        // blocks that can hold documents of [w0, wlast]: first block with last >= w0 ... first with last >= wlast
                        uint32_t b_lo, b_hi;
                        if (uni(sh.seg_plane[k]) != PL_NONE) { // no rows to walk: dense_pass reads the plane (seg_cnt 0 below); the planner bounds
                                                               // the task's windows by the lists' last documents, so "alive" costs nothing
                                galive = true;
                                sh.seg_lo[k] = 0;
                                sh.seg_cnt[k] = 0;
                                continue;
                        }
                        if (uni(t.win_off) != 0xffffffffu) {
                                // indexed list: the two staged entries replace both directory searches (win[w + 1] is the first
                                // block with last >= the next window's first docID; it may still hold documents of this window)
                                b_lo = uni(sh.seg_wlo[k]);
                                b_hi = min(uni(sh.seg_whi[k]), nblocks - 1);
                        } else {
                                b_lo = uni(sh.lcur[k]); // (the lead pass above left a lead list's cursor on its first block of this window; the cursor stays there:
                                                        //  the next window's search starts from it)
                                if (b_lo < nblocks && bl[b_lo] < w0)
                                        b_lo = uni(wave_lower_bound(bl, b_lo, nblocks, w0));
                                b_hi = b_lo;
                                if (b_lo < nblocks) {
                                        b_hi = uni(wave_lower_bound(bl, b_lo, nblocks, wlast));
                                        if (b_hi >= nblocks)
                                                b_hi = nblocks - 1;
                                }
                        }
                        if (b_lo < nblocks)
                                galive = true;
                        // uniform stores by every lane (see the control-flow note in k_and)
                        sh.seg_lo[k] = b_lo;
                        sh.seg_cnt[k] = b_lo < nblocks ? b_hi - b_lo + 1 : 0;
                }
                sh.seg_cnt[q.nterms] = 0xffffffffu; // sentinel: the lane-to-term walk stops here
                if (!galive && !neg)
                        done = true;
                if (done)
                        break;
                // bits[0] = A: the lead group's union, then the running conjunction; bits[1] = B: the union of the group being
                // read.  Every term only sets bits; a finished group is folded in word-wise (A &= B) before B is reused, and
                // the last group's fold is fused into the expansion below.
                PROF_LAP(1);
                __syncthreads(); // (bitmaps: zeroed at task start, re-zeroed by every expansion sweep)
                PROF_LAP(2);
                dense_pass<WG, CODEC>(sh, index, blk_last, blk_off, 0, g2, g1, w0, planes, plw PROF_PASS); // groups 0 and 1 together
                for (uint32_t kb = g2; kb < q.nterms;) {
                        uint32_t ke = kb + 1;
                        while (ke < q.nterms && !(qterms[q.term_base + ke] & QT_GROUP))
                                ++ke;
                        for (uint32_t i = tid; i < BM_STRIDE; i += WG) {
                                sh.bm[i] &= sh.bm[BM_STRIDE + i];
                                sh.bm[BM_STRIDE + i] = 0;
                        }
                        __syncthreads();
                        dense_pass<WG, CODEC>(sh, index, blk_last, blk_off, kb, ke, kb, w0, planes, plw PROF_PASS);
                        kb = ke;
                }
                if (as_bitmap) {
                        for (; wdone < w; ++wdone) // (windows the lead group skipped)
                                dense_store_zero<WG>(qout + (size_t)(wdone - task.tile_begin) * SPAN_WORDS);
                        produced += dense_store<WG>(sh, w0, ngroups > 1, neg, masked, qout + (size_t)(w - task.tile_begin) * SPAN_WORDS);
                        wdone = w + 1;
                } else
                        produced += dense_expand<WG>(sh, w0, ngroups > 1, neg, masked, qout, produced PROF_PASS);
                ++w;
        }
        if (as_bitmap)
                for (; wdone < task.tile_end; ++wdone) // (... and the ones behind the last window that could hold a match)
                        dense_store_zero<WG>(qout + (size_t)(wdone - task.tile_begin) * SPAN_WORDS);
        __syncthreads();
        if (uni(tid >> 6) == 0)
                *count_out = produced;
        PROF_LAP(8);
}

// bitmap-window tasks: persistent 512-thread workgroups draw TASK_DENSE tasks, heaviest first
#ifndef TRI_DENSE_WAVES
#define TRI_DENSE_WAVES 8 // waves per SIMD the register budget is cut for (8: 64 VGPRs, four 512-thread workgroups per CU)
#endif
template <int CODEC>
__global__ __launch_bounds__(DENSE_WG, TRI_DENSE_WAVES) void k_and_dense(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last,
                                                        const uint32_t *__restrict__ blk_off, const uint32_t *__restrict__ win,
                                                        const DevTerm *__restrict__ terms, const DevQuery *__restrict__ plan,
                                                        const DevTask *__restrict__ tasks, const uint32_t *__restrict__ sched,
                                                        const uint32_t *__restrict__ qterms, const uint32_t ntasks, uint32_t *__restrict__ ticket,
                                                        uint32_t *__restrict__ out, uint32_t *__restrict__ counts,
                                                        const uint32_t *__restrict__ masked, const uint32_t *__restrict__ qplane,
                                                        const uint32_t *__restrict__ planes, const uint32_t plw) {
        __shared__ DenseShared sh;
        const uint32_t wave = uni(threadIdx.x >> 6);
        PROF_DECL;
        PROF_START();
        for (;;) {
                if (wave == 0) { // uniform draw: 64 lanes add 1 each (one +64 atomic), see k_and
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
                PROF_LAP(0);
                TASKTIME_DENSE(8 * ticket_no);
                dense_task<DENSE_WG, CODEC>(sh, index, blk_last, blk_off, win, terms, qterms, q, task, out, counts + tix, masked, qplane, planes, plw PROF_PASS);
                TASKTIME_DENSE(8 * ticket_no + 1);
        }
        PROF_LAP(9);
        PROF_FLUSH();
}

// candidate-tile tasks (TASK_CAND)
#ifndef TRI_AND_PROBES
#define TRI_AND_PROBES 8
#endif
constexpr int AND_PROBES = TRI_AND_PROBES; // plane probes of a lane in flight together
template <int CODEC>
#ifndef TRI_AND_WAVES
#define TRI_AND_WAVES 4
#endif
__global__ __launch_bounds__(AND_WG, TRI_AND_WAVES) void k_and(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last,
                                                const uint32_t *__restrict__ blk_off, const uint32_t *__restrict__ win,
                                                const DevTerm *__restrict__ terms,
                                                const DevQuery *__restrict__ plan, const DevTask *__restrict__ tasks,
                                                const uint32_t *__restrict__ sched, const uint32_t *__restrict__ qterms,
                                                const uint32_t *__restrict__ cand_q, uint32_t *__restrict__ ticket,
                                                uint32_t *__restrict__ out, uint32_t *__restrict__ counts,
                                                const uint32_t *__restrict__ masked, const uint32_t *__restrict__ qplane,
                                                const uint32_t *__restrict__ planes, const uint32_t plw) {
        __shared__ AndShared sh;
        const uint32_t tid = threadIdx.x;
        const uint32_t wave = uni(tid >> 6);
        // the queue of the XCD this workgroup runs on (HW_REG_XCC_ID: register 20, bits 0 .. 3) — placement changes speed only: every queue is
        // drained by whoever gets there, the own one first
        const uint32_t home = uni(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11))) % CAND_QUEUES;
        uint32_t qdone = 0; // (wave 0's: the queues found empty)
        PROF_DECL;
        PROF_START();
        for (;;) {
                // (Tried in round 4 and dropped: ONE 128-byte record per task — geometry, the first four terms with their plane rows, the lead's and the
                //  second term's DevTerm — stored in run order and fetched a task ahead by one wave, instead of the sched -> task -> query ->
                //  qterms -> terms chain below.  k_and 0.71 -> 0.72 ms at cfg2 (the chain's loads are scalar and L2-resident: the task's time is the
                //  lead's decode and the probes), 0.77 ms with the ticket drawn two tasks ahead (every task held in reserve is one no idle
                //  workgroup can take at the kernel's end), and 0.3 ms more host planning per batch for the 3 MB of records.)
                // next query: wave 0 draws the ticket.  All 64 lanes add 1 (the compiler folds that into ONE
                // global atomic of +64 with a uniform operand — no lane-divergent branch at the loop head), so the
                // counter advances in units of 64 per draw.
                //
                // One ticket word per XCD (a single word drawn by 1024 workgroups costs ~3 us a draw under load; a word per XCD a tenth), and the
                // queue behind it holds the tasks that probe the SAME plane rows (planner.hpp, "k_and's queues"): the XCD's L2 keeps a row's
                // sectors from one task's probes to the next's.
                if (wave == 0) {
                        uint32_t got = 0xffffffffu;
                        for (uint32_t t = 0; t < CAND_QUEUES && got == 0xffffffffu; ++t) {
                                const uint32_t x = (home + t) % CAND_QUEUES;
                                if ((qdone >> x) & 1u)
                                        continue;
                                const uint32_t q0 = cand_q[x], qn = cand_q[x + 1] - q0;
                                const uint32_t old = uni(atomicAdd(ticket + x * CAND_TICKET_STRIDE, 1u)) >> 6;
                                if (old < qn)
                                        got = q0 + old;
                                else
                                        qdone |= 1u << x;
                        }
                        sh.bcast[0] = got;
                }
                __syncthreads();
                const uint32_t ticket_no = uni(sh.bcast[0]);
                __syncthreads();
                if (ticket_no == 0xffffffffu)
                        break;
                const uint32_t tix = sched[ticket_no];
                TASKTIME(8 * ticket_no);
                const DevTask task = tasks[tix];
                const uint32_t slot = task.slot;
                const DevQuery q = plan[slot];
                const DevTerm lead = terms[qterms[q.term_base] & QT_TERM];
                TRACE(1, slot, q.nterms);
                uint32_t *qout = out + task.out_off;
                uint32_t produced = 0;
                sh.lcur[tid & 15] = 0xffffffffu; // "not positioned yet"
                const uint32_t tb_end = min(lead.nblocks, task.tile_end * TILE_BLOCKS);
                PROF_LAP(10);

                for (uint32_t tb = task.tile_begin * TILE_BLOCKS; tb < tb_end; tb += TILE_BLOCKS) {
                        const uint32_t nb = min((uint32_t)TILE_BLOCKS, lead.nblocks - tb);
                        uint32_t C = (tb + nb == lead.nblocks) ? (nb - 1) * 32 + lead.last_n : nb * 32;
                        // ---- decode the lead tile: one lane per block (unpack_block, google_codec.cpp:596-639)
                        if (tid < nb) {
                                const uint32_t b = tb + tid, gb = lead.first_block + b;
                                const uint32_t off = blk_off[gb];
                                const uint32_t n = TRI_BLOCK_N(lead, b, index, off);
                                const uint32_t last = blk_last[gb];
                                uint32_t doc = b ? blk_last[gb - 1] : 0;
#if defined(TRI_AND_EXP) && TRI_AND_EXP == 2 // (perf probe: no delta stream — made-up ascending documents; what the lead's decode costs)
                                const uint32_t row = tid * 32, step = (last - doc) / 32u + 1u;
                                for (uint32_t i = 0; i + 1 < n; ++i) {
                                        doc += step;
                                        sh.cand[row | ((i + tid) & 31u)] = min(doc, last);
                                }
                                sh.cand[row | ((n - 1 + tid) & 31u)] = last;
                                (void)off;
#else
                                DeltaStream<CODEC> s;
                                s.init(index, lead, b, off);
                                const uint32_t row = tid * 32;
                                for (uint32_t i = 0; i + 1 < n; ++i) {
                                        doc += s.next();
                                        sh.cand[row | ((i + tid) & 31u)] = doc;
                                }
                                sh.cand[row | ((n - 1 + tid) & 31u)] = last;
#endif
                        }
                        __syncthreads();
                        PROF_LAP(11);
                        TASKTIME(8 * ticket_no + 2); // (probe builds: the last tile's stamps stay)
                        TRACE(2, slot, tb);

                        // ---- every other group filters the surviving candidates: a candidate survives a group when any
                        //      of the group's terms holds it (hit bits are OR-ed across the group's terms)
                        bool gneg = false; // the group being filtered is the excluded one (logicalnot): its hits remove
                        for (uint32_t k = 1; k < q.nterms && C; ++k) {
                                const uint32_t tt = qterms[q.term_base + k];
                                const DevTerm t = terms[tt & QT_TERM];
                                if (tt & QT_GROUP) {
                                        sh.hit[tid] = 0;
                                        gneg = tt & QT_NOT;
                                }
                                __syncthreads();
#if defined(TRI_FORCE_CAND)
                                const bool bd = false;
#elif defined(TRI_FORCE_BLOCK)
                                const bool bd = true;
#else
                                const bool bd = t.nblocks <= lead.documents;
#endif
                                TRACE(3, slot, (k << 16) | (bd ? 1 : 0));
                                const uint32_t prow = qplane ? qplane[q.term_base + k] : PL_NONE;
                                if (prow != PL_NONE) {
                                        // the term has a plane (k_term_planes decoded it once for the whole batch): advance(candidate) is a bit probe
                                        // (eight probes of a lane in flight together: one at a time, a tile of 8192 candidates was 32 dependent round trips —
                                        //  12 us of a 24 us task at cfg2)
                                        const uint32_t *pa = planes + (size_t)prow * plw; // (plane 0 of the row: the plane cache's first region)
                                        for (uint32_t j0 = tid; j0 < C; j0 += AND_WG * AND_PROBES) {
                                                uint32_t doc[AND_PROBES], w[AND_PROBES];
#pragma unroll
                                                for (int u = 0; u < AND_PROBES; ++u) {
                                                        const uint32_t j = j0 + u * AND_WG;
                                                        doc[u] = j < C ? sh.cand[phys(j)] : 0u;
                                                }
#pragma unroll
                                                for (int u = 0; u < AND_PROBES; ++u)
#if defined(TRI_AND_EXP) && TRI_AND_EXP == 1 // (perf probe: every gather inside one 4 KB stretch of the row — what the probes cost when they do not diverge)
                                                        w[u] = pa[(doc[u] >> 5) & 0x3ffu];
#else
                                                        w[u] = pa[doc[u] >> 5];
#endif
#pragma unroll
                                                for (int u = 0; u < AND_PROBES; ++u) {
                                                        const uint32_t j = j0 + u * AND_WG;
                                                        if (j < C && ((w[u] >> (doc[u] & 31u)) & 1u))
                                                                atomicOr(&sh.hit[j >> 5], 1u << (j & 31u));
                                                }
                                        }
                                } else
                                and_filter_tile<CODEC>(sh, index, blk_last, blk_off, win, t, C, k - 1, bd PROF_PASS);
                                TRACE(4, slot, C);
                                __syncthreads();
                                TASKTIME(8 * ticket_no + 2 + min(k, 5u));
                                PROF_LAP(bd ? 12 : 13);
                                const bool lastterm = k + 1 == q.nterms;
                                if (!lastterm && !(qterms[q.term_base + k + 1] & QT_GROUP))
                                        continue; // more terms of this OR group to come
                                // compact survivors (stable => still ascending)
                                uint32_t bits = sh.hit[tid];
                                if (gneg) { // row tid holds candidates tid * 32 ..: keep the ones NOT hit
                                        const uint32_t left = C > tid * 32 ? C - tid * 32 : 0u;
                                        bits = ~bits & (left >= 32 ? 0xffffffffu : ((1u << left) - 1u));
                                }
                                if (masked && lastterm) // drop the survivors a newer segment has masked (docidupdates.h:90-119)
                                        for (uint32_t m = bits; m;) {
                                                const uint32_t kbit = __builtin_ctz(m);
                                                m &= m - 1;
                                                const uint32_t doc = sh.cand[phys(tid * 32 + kbit)];
                                                if ((masked[doc >> 5] >> (doc & 31u)) & 1u)
                                                        bits &= ~(1u << kbit);
                                        }
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
                                if (!masked) {
                                        for (uint32_t j = tid; j < C; j += AND_WG)
                                                qout[produced + j] = sh.cand[phys(j)];
                                } else {
                                        // single-term query over a segment with masked documents: row-wise keep mask, scan, write
                                        const uint32_t left = C > tid * 32 ? C - tid * 32 : 0u;
                                        uint32_t bits = left >= 32 ? 0xffffffffu : ((1u << left) - 1u);
                                        for (uint32_t m = bits; m;) {
                                                const uint32_t kbit = __builtin_ctz(m);
                                                m &= m - 1;
                                                const uint32_t doc = sh.cand[phys(tid * 32 + kbit)];
                                                if ((masked[doc >> 5] >> (doc & 31u)) & 1u)
                                                        bits &= ~(1u << kbit);
                                        }
                                        uint32_t wtot;
                                        uint32_t ex = wave_excl_scan(__popc(bits), wtot);
                                        sh.scan[tid >> 6] = wtot;
                                        __syncthreads();
                                        uint32_t wbase = 0, total = 0;
                                        for (int w = 0; w < AND_WG / 64; ++w) {
                                                if (w < (int)(tid >> 6))
                                                        wbase += sh.scan[w];
                                                total += sh.scan[w];
                                        }
                                        uint32_t o = produced + ex + wbase;
                                        for (uint32_t m = bits; m;) {
                                                const uint32_t kbit = __builtin_ctz(m);
                                                m &= m - 1;
                                                qout[o++] = sh.cand[phys(tid * 32 + kbit)];
                                        }
                                        C = uni(total);
                                }
                        }
                        produced += C;
                        __syncthreads();
                        PROF_LAP(14);
                }
                if (wave == 0)
                        counts[tix] = produced; // scalar branch; the wave's lanes store one identical dword
                TASKTIME(8 * ticket_no + 1);
                TRACE(5, slot, produced);
        }
        PROF_LAP(15);
        PROF_FLUSH();
        TRACE(6, 0, 0);
}

