// k_phrase.hpp — positional (phrase) constraints over match segments
// Part of libtrinity_hip.so (MI355X / gfx950); included by trinity_hip.hip.  New code, no reference source.
#pragma once
#include "k_match.hpp"

// DocsSetIterators::Phrase (docset_iterators.cpp:66-224) is a conjunction on docIDs followed by a positional check
// through the DocWordsSpace (docwordspace.h:16-92): the hits of every phrase term are materialised for the candidate
// document in phrase order (candidate_document::materialize_term_hits, queryexec_ctx.cpp:317-351; a term repeated in
// the phrase only once), each hit doing dws->set(termID, pos) — ONE term per position, last writer wins — and the
// phrase matches at a start position p0 of term 0 when dws->test(term_k, p0 + k) holds for k = 1..n-1.  matchCnt
// counts the matching start positions (capped at 1 unless scores are accumulated, exec.cpp:296).
//
// Here k_and has already intersected the phrase's terms (they are conjuncts of the query), so every candidate of a
// task's match segment holds every phrase term.  One lane takes one candidate: for each term it finds the block through
// the directory, walks the deltas to the document's slot, the freqs to its frequency, then the hits of the preceding
// slots (Google::Decoder::skip_block_doc, google_codec.cpp:497-531) to the document's own hits; positions are streamed
// from there (materialize_hits, :533-594) for the membership / ownership tests.  Survivors are compacted in place.
//
// LUCENE keeps the positions of a term as ONE stream in hits.data, in blocks of 128 hits that are independent of the document
// blocks (lucene_codec.cpp:245-307 writer, 401-462 refill_hits): a document's hits are found by COUNT — the hits of all preceding
// documents of the term.  The upload pass therefore records, per directory row, the hits before the row (blk_hits[]) and, per
// term, the byte offset of every 128-hit block and of the varbyte tail (hdir[]); a lane adds the freqs of the preceding slots of
// its row and lands on the absolute hit index, which HitStream<CODEC_LUCENE> turns into (block, quarter, slot).
struct HitCtx {
        const uint8_t *base;      // GOOGLE: index[] (hits are inline), LUCENE: hits.data
        const uint32_t *blk_hits; // per directory row, where its hits start — LUCENE: hit ordinal within the term; GOOGLE: byte offset into index[]
        const uint32_t *hdir;     // LUCENE: per term { nfull, off[0..nfull-1], tail_off } into hits.data
};

template <int CODEC>
struct HitStream;

template <>
struct HitStream<CODEC_GOOGLE> { // google_codec.cpp:533-594: varbyte (delta << 1 | newPayloadLen) [u8 len] payload
        VbStream s;
        uint32_t plen;
        uint64_t pw; // plain: the document's (<= 8) single-byte hits
        bool plain;
        static constexpr uint32_t FREQ_MASK = 0xffffu; // th->freq is tokenpos_t
        static constexpr uint32_t FREQ_PLAIN = 0x40000000u; // k_phrase's freq entries: the document's hits are single bytes (BLK_HITS_PLAIN)
        // ... and the entry HOLDS them (round 6, k_term_hits: a document of a head term with at most seven single-byte hits — almost all): the 64-bit entry is
        // { hit bytes 0 .. 6 in bits 0 .. 55, the frequency in bits 56 .. 62, bit 63 } — a candidate's hits are in the lane's registers with the entry's load, where
        // a locator costs one more dependent gather (a 64-byte sector of index[] for eight bytes) per candidate and phrase term
        static constexpr uint32_t FREQ_INLINE = 0x80000000u;
        static constexpr uint32_t INLINE_MAX = 7;
        static __device__ __forceinline__ uint32_t count(const uint32_t fentry) { return (fentry & FREQ_INLINE) ? (fentry >> 24) & 0x7fu : fentry & FREQ_MASK; }
        static __device__ __forceinline__ uint64_t inline_bytes(const uint32_t loc, const uint32_t fentry) { return (uint64_t)loc | ((uint64_t)(fentry & 0x00ffffffu) << 32); }
        __device__ __forceinline__ void init(const HitCtx &c, const uint32_t, const uint32_t loc) {
                plain = false;
                s.init(c.base + loc);
                plen = 0; // payload length state restarts with every document
        }
        // fentry: the candidate's frequency as phrase_locate_block left it.  At most eight single-byte hits: one unaligned load holds
        // them all, and the walk is shifts of a register instead of a byte stream with loads in flight
        __device__ __forceinline__ void init_entry(const HitCtx &c, const uint32_t pad, const uint32_t loc, const uint32_t fentry) {
                if (fentry & FREQ_INLINE) {
                        plain = true;
                        pw = inline_bytes(loc, fentry);
                } else if ((fentry & FREQ_PLAIN) && (fentry & FREQ_MASK) <= 8u) {
                        typedef uint64_t ph_u64_a1 __attribute__((aligned(1)));
                        plain = true;
                        pw = *(const ph_u64_a1 *)(c.base + loc);
                } else
                        init(c, pad, loc);
        }
        __device__ __forceinline__ uint32_t next() {
                if (plain) {
                        const uint32_t v = (uint32_t)pw & 0xffu;
                        pw >>= 8;
                        return v >> 1;
                }
                const uint32_t v = s.next();
                if (v & 1u)
                        plen = s.byte();
                s.skip(plen);
                return v >> 1;
        }
};

template <>
struct HitStream<CODEC_LUCENE> {
        const uint8_t *hits;
        const uint32_t *hd;
        uint32_t nfull, h;
        LValStream lv;
        VbStream vb;
        bool tail;
        static constexpr uint32_t FREQ_MASK = 0xffffffffu;
        static __device__ __forceinline__ uint32_t count(const uint32_t fentry) { return fentry; }
        __device__ __forceinline__ void seek() {
                const uint32_t hb = h >> 7;
                if (hb < nfull) {
                        tail = false;
                        lv.init_group(hits, hd[1 + hb], (h & 127u) >> 5);
                        for (uint32_t k = 0; k < (h & 31u); ++k)
                                (void)lv.next();
                } else { // lucene_codec.cpp:339-352: varbyte (posDelta << 1 | newLen) [u8 len]; payload bytes follow the whole tail
                        tail = true;
                        vb.init(hits + hd[1 + nfull]);
                        for (uint32_t k = nfull * 128u; k < h; ++k)
                                if (vb.next() & 1u)
                                        (void)vb.byte();
                }
        }
        __device__ __forceinline__ void init(const HitCtx &c, const uint32_t hdir_off, const uint32_t loc) {
                hits = c.base;
                hd = c.hdir + hdir_off;
                nfull = hd[0];
                h = loc;
                seek();
        }
        __device__ __forceinline__ void init_entry(const HitCtx &c, const uint32_t hdir_off, const uint32_t loc, const uint32_t) { init(c, hdir_off, loc); }
        __device__ __forceinline__ uint32_t next() {
                if (tail) {
                        const uint32_t v = vb.next();
                        if (v & 1u)
                                (void)vb.byte();
                        return v >> 1;
                }
                const uint32_t v = lv.next();
                ++h;
                if (!(h & 31u))
                        seek();
                return v;
        }
};

// Candidates are handled in tiles held in LDS: up to PHRASE_TILE of them, fewer when the phrase has many distinct terms
// (one row of hit locators per distinct term: PHRASE_SLOTS entries in all).
#ifndef TRI_PHRASE_TILE
#define TRI_PHRASE_TILE 512 // (cfg4, ms: 1024 -> 25.8, 512 -> 23.0, 256 -> 31.0: more workgroups per CU against more tiles per task)
#endif
constexpr uint32_t PHRASE_TILE = TRI_PHRASE_TILE;
constexpr uint32_t PHRASE_SLOTS = 4 * PHRASE_TILE;
struct PhraseShared {
        uint32_t cdoc[PHRASE_TILE];     // the tile's candidates (ascending)
        uint32_t hits_off[PHRASE_SLOTS]; // [row * tile + j]: where candidate j's hits of the row's term start (GOOGLE: byte offset into
                                         // index[]; LUCENE: hit ordinal within the term)
        uint32_t freq[PHRASE_SLOTS];
        double ps[PHRASE_TILE];         // scored mode: sum of the phrase scores of the candidate
        uint8_t alive[PHRASE_TILE];
        uint32_t row[MAX_PHRASE_TERMS]; // phrase position -> row (a term repeated in the phrase shares the row of its first occurrence)
        uint32_t rpad[MAX_PHRASE_TERMS]; // row -> its term's DevTerm::pad (LUCENE: the term's row in hdir[])
        DevTerm rterm[MAX_PHRASE_TERMS]; // row -> its term
        uint32_t rrow[MAX_PHRASE_TERMS]; // row -> its term's plane row when the term is located by rank (PL_NONE: by walking its blocks)
        uint64_t rhs[MAX_PHRASE_TERMS];  // ... and where that row's hits entries start (hs_off[row])
        uint32_t rtk[MAX_PHRASE_TERMS], rbx[2 * MAX_PHRASE_TERMS]; // row -> its term id; the two ends of the tile's block range as the waves found them
        uint32_t rb0[MAX_PHRASE_TERMS], rnb[MAX_PHRASE_TERMS], rstart[MAX_PHRASE_TERMS]; // row -> first block of the tile's docID range, blocks walked (0: the scattered path), where its blocks start in the pass
        uint32_t scan[8];
        uint32_t bcast[4];
};
// ONE object for the kernel and the functions it calls: a static __shared__ behind an accessor keeps every access a plain LDS instruction (k_planes.hpp)
__device__ __forceinline__ PhraseShared &phrase_shared() {
        __shared__ PhraseShared sh;
        return sh;
}
#ifndef TRI_PHRASE_WAVES
#define TRI_PHRASE_WAVES 6
#endif
constexpr uint32_t PHRASE_WGS_PER_CU = (160u * 1024u / sizeof(PhraseShared)) < (uint32_t)TRI_PHRASE_WAVES ? (160u * 1024u / sizeof(PhraseShared)) : (uint32_t)TRI_PHRASE_WAVES; // LDS; the register budget (k_phrase's launch bounds)

// Is position `q` among the `freq` hits starting at index[hits_off]?  (positions ascend within a document)
template <int CODEC>
__device__ __forceinline__ bool phrase_has_pos(const HitCtx &ctx, const uint32_t hdir_off, const uint32_t hits_off, const uint32_t freq, const uint32_t q) {
        HitStream<CODEC> s;
        s.init_entry(ctx, hdir_off, hits_off, freq);
        uint32_t pos = 0;
        for (uint32_t h = 0, n = HitStream<CODEC>::count(freq); h < n; ++h) {
                pos = (pos + s.next()) & 0xffffu;
                if (pos == q)
                        return true;
                if (pos > q)
                        return false;
        }
        return false;
}

// Block-driven location: one lane walks one block of term t ONCE for all the tile's candidates it holds (phrase terms are
// conjuncts of the query, so every candidate in the block's docID range is a document of the block): deltas mark the slots,
// then freqs and hits are walked in lockstep up to the last marked slot, leaving (hits locator, freq) for each candidate.
// Few candidates scattered over a long list: every candidate finds its block through the cell index and the first
// candidate of a block calls this for all of them (k_phrase).
template <int CODEC>
__device__ __forceinline__ void phrase_locate_block(PhraseShared &sh, const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last,
                                                    const uint32_t *__restrict__ blk_off, const HitCtx &ctx, const DevTerm t, const uint32_t b,
                                                    const uint32_t C, const uint32_t slot0) {
        const uint32_t *bl = blk_last + t.first_block;
        const uint32_t prev = b ? bl[b - 1] : 0, last = bl[b];
        uint32_t lo = 0, hi = C; // first candidate > prev
        while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (sh.cdoc[mid] <= prev)
                        lo = mid + 1;
                else
                        hi = mid;
        }
        const uint32_t ci = lo;
        if (ci >= C || sh.cdoc[ci] > last)
                return;
        const uint32_t off = blk_off[t.first_block + b];
        const uint32_t n = TRI_BLOCK_N(t, b, index, off);
        uint32_t mask = 0, cj = ci, doc = prev;
        uint32_t cv = sh.cdoc[cj];
        if (CODEC == CODEC_GOOGLE) {
                const uint32_t hits_plain = ctx.blk_hits[t.first_block + b];
                if (n == 32 && (hits_plain & BLK_HITS_PLAIN)) {
                        // a full block: 31 deltas and 32 freqs, 63 bytes if every one of them is a single byte (every block of a head term) —
                        // four wide loads, then byte adds from registers; with one byte per hit the locators are a running sum of the freqs
                        typedef uint32_t ph_u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
                        const uint8_t *p = index + off;
                        const ph_u32x4_a1 A = *(const ph_u32x4_a1 *)p, B = *(const ph_u32x4_a1 *)(p + 16), Cq = *(const ph_u32x4_a1 *)(p + 32),
                                          Dq = *(const ph_u32x4_a1 *)(p + 48);
                        const uint32_t w[16] = {A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w, Cq.x, Cq.y, Cq.z, Cq.w, Dq.x, Dq.y, Dq.z, Dq.w};
                        uint32_t any = w[15] & 0x00ffffffu; // (byte 63 is the block's first hit)
#pragma unroll
                        for (int k = 0; k < 15; ++k)
                                any |= w[k];
                        if (!(any & 0x80808080u)) {
#pragma unroll
                                for (int j = 0; j < 32; ++j) {
                                        doc = j < 31 ? doc + ((w[j >> 2] >> ((j & 3) * 8)) & 0xffu) : last;
                                        if (cv == doc) {
                                                mask |= 1u << j;
                                                ++cj;
                                                cv = cj < C ? sh.cdoc[cj] : 0xffffffffu;
                                        }
                                }
                                uint32_t h = off + (hits_plain & ~BLK_HITS_PLAIN);
                                cj = ci;
#pragma unroll
                                for (int j = 0; j < 32; ++j) {
                                        const uint32_t f = (w[(31 + j) >> 2] >> (((31 + j) & 3) * 8)) & 0xffu;
                                        if ((mask >> j) & 1u) {
                                                sh.hits_off[slot0 + cj] = h;
                                                sh.freq[slot0 + cj] = f | HitStream<CODEC_GOOGLE>::FREQ_PLAIN;
                                                ++cj;
                                        }
                                        h += f;
                                }
                                return;
                        }
                }
                VbStream s;
                s.init(index + off);
                for (uint32_t i = 0; i < n; ++i) {
                        doc = i + 1 < n ? doc + s.next() : last;
                        if (cv == doc) {
                                mask |= 1u << i;
                                ++cj;
                                cv = cj < C ? sh.cdoc[cj] : 0xffffffffu;
                        }
                }
                VbStream sf = s; // freqs start here
                const uint32_t hits_at = ctx.blk_hits[t.first_block + b];
                cj = ci;
                if (hits_at & BLK_HITS_PLAIN) { // one byte per hit: locators follow from the frequencies alone
                        uint32_t h = off + (hits_at & ~BLK_HITS_PLAIN);
                        for (uint32_t i = 0; i < n && (mask >> i); ++i) {
                                const uint32_t f = sf.next();
                                if ((mask >> i) & 1u) {
                                        sh.hits_off[slot0 + cj] = h;
                                        sh.freq[slot0 + cj] = (f & HitStream<CODEC_GOOGLE>::FREQ_MASK) == f ? (f | HitStream<CODEC_GOOGLE>::FREQ_PLAIN) : f;
                                        ++cj;
                                }
                                h += f;
                        }
                        return;
                }
                s.init(index + off + hits_at); // hits start (from the directory)
                for (uint32_t i = 0; i < n && (mask >> i); ++i) {
                        const uint32_t f = sf.next();
                        if ((mask >> i) & 1u) {
                                sh.hits_off[slot0 + cj] = (uint32_t)(s.tell() - index);
                                sh.freq[slot0 + cj] = f;
                                ++cj;
                                if (!((mask >> i) >> 1))
                                        break;
                        }
                        uint32_t plen = 0; // payload length state restarts with every document
                        for (uint32_t h = 0; h < f; ++h) {
                                const uint32_t v = s.next();
                                if (v & 1u)
                                        plen = s.byte();
                                s.skip(plen);
                        }
                }
        } else {
                DeltaStream<CODEC> ds;
                ds.init(index, t, b, off);
                for (uint32_t i = 0; i < n; ++i) {
                        doc = i + 1 < n ? doc + ds.next() : last;
                        if (cv == doc) {
                                mask |= 1u << i;
                                ++cj;
                                cv = cj < C ? sh.cdoc[cj] : 0xffffffffu;
                        }
                }
                FreqStream<CODEC> fs;
                fs.init(index, t, b, off, ds);
                uint32_t h = ctx.blk_hits[t.first_block + b];
                cj = ci;
                for (uint32_t i = 0; i < n && (mask >> i); ++i) {
                        const uint32_t f = fs.next();
                        if ((mask >> i) & 1u) {
                                sh.hits_off[slot0 + cj] = h;
                                sh.freq[slot0 + cj] = f;
                                ++cj;
                        }
                        h += f;
                }
        }
}

// The first block of term t whose last docID is >= key (t.nblocks: none), found by the whole wave: the cell index brackets it to at
// most a cell's worth of blocks, then 64 probes a round.
__device__ __forceinline__ uint32_t phrase_first_block(const uint32_t *__restrict__ bl, const uint32_t *__restrict__ win, const DevTerm &t, const uint32_t key) {
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t lo = 0, hi = t.nblocks;
        if (t.win_off != 0xffffffffu) {
                lo = uni(win[t.win_off + (key >> CELL_LOG2)]);
                hi = min(uni(win[t.win_off + (key >> CELL_LOG2) + 1]) + 1u, t.nblocks);
        }
        for (;;) { // (uniform)
                const bool below = lo + lane < hi && bl[lo + lane] < key;
                const uint32_t cnt = (uint32_t)__popcll(__ballot(below));
                lo += cnt;
                if (cnt != 64u)
                        return lo;
        }
}

// ---- the hits entries of a head term's postings (GOOGLE; lists of full blocks), once per index: phs[hs_off[row] + 32 b + slot] = what phrase_locate_block leaves
//      for that document — the hits locator (byte offset into index[]) in the low word, the frequency entry (with FREQ_PLAIN where the document's hits are single bytes)
//      in the high word.  With the row's rank directory (k_term_planes) a candidate's entry is TWO dependent loads away — its rank in plane 0, then the entry — where
//      the block walk decodes a 63-byte block per 32 documents of the tile's range, a lane per block, most lanes idle.  One lane per block here, all blocks of the term.
__global__ __launch_bounds__(256) void k_term_hits(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_off, const uint32_t *__restrict__ blk_hits,
                                                   const DevTerm *__restrict__ terms, const uint32_t *__restrict__ build /* (term, row) pairs */, const uint64_t *__restrict__ hs_off,
                                                   unsigned long long *__restrict__ phs, uint32_t *__restrict__ term_row) {
        const uint32_t term = build[2 * blockIdx.y], row = build[2 * blockIdx.y + 1];
        const DevTerm t = terms[term];
        const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
        if (b == 0)
                term_row[term] = row; // (every later kernel of the stream sees the row: k_phrase takes the rank path for this term from now on)
        if (b >= t.nblocks)
                return;
        unsigned long long *out = phs + hs_off[row] + 32ull * b;
        const uint32_t gb = t.first_block + b;
        const uint32_t off = blk_off[gb];
        const uint32_t n = TRI_BLOCK_N(t, b, index, off);
        VbStream s;
        s.init(index + off);
        for (uint32_t i = 0; i + 1 < n; ++i)
                (void)s.next(); // (the deltas: the frequencies follow them)
        const uint32_t hits_at = blk_hits[gb];
        if (hits_at & BLK_HITS_PLAIN) { // one byte per hit: locators follow from the frequencies alone
                uint32_t h = off + (hits_at & ~BLK_HITS_PLAIN);
                typedef uint64_t ph_u64_a1 __attribute__((aligned(1)));
                for (uint32_t i = 0; i < n; ++i) {
                        const uint32_t f = s.next();
                        const uint32_t fe = (f & HitStream<CODEC_GOOGLE>::FREQ_MASK) == f ? (f | HitStream<CODEC_GOOGLE>::FREQ_PLAIN) : f;
                        if (f <= HitStream<CODEC_GOOGLE>::INLINE_MAX) // the hits themselves (see HitStream<CODEC_GOOGLE>::FREQ_INLINE)
                                out[i] = (*(const ph_u64_a1 *)(index + h) & 0x00ffffffffffffffull) | ((unsigned long long)f << 56) | 0x8000000000000000ull;
                        else
                                out[i] = (unsigned long long)h | ((unsigned long long)fe << 32);
                        h += f;
                }
                return;
        }
        VbStream sh_;
        sh_.init(index + off + hits_at); // hits start (from the directory)
        for (uint32_t i = 0; i < n; ++i) {
                const uint32_t f = s.next();
                out[i] = (unsigned long long)(uint32_t)(sh_.tell() - index) | ((unsigned long long)f << 32);
                uint32_t plen = 0; // payload length state restarts with every document
                for (uint32_t h = 0; h < f; ++h) {
                        const uint32_t v = sh_.next();
                        if (v & 1u)
                                plen = sh_.byte();
                        sh_.skip(plen);
                }
        }
}

// ---- every row of the phrase a head term: the rows' hits entries by RANK, for ALL of a lane's candidates at once (see k_phrase's check).  A function of its
//      own (not inlined): its registers — eight record pairs, eight entries in flight per lane — are allocated apart from the kernel's (inlined, the compiler
//      spilled four dozen registers around the check).  The entries go to the lane's own places of hits_off[] / freq[].
__device__ __noinline__ void phrase_rank_entries(const uint32_t *__restrict__ prank_, const unsigned long long *__restrict__ phs_, const uint32_t plw_, const uint32_t rows_,
                                                 const uint32_t tile_, const uint32_t C_, const uint32_t cmin_) {
        PhraseShared &sh = phrase_shared();
        constexpr uint32_t PER = PHRASE_TILE / AND_WG;
        const uint32_t tid = threadIdx.x;
        const uint32_t *const prank = uni_ptr(prank_);
        const unsigned long long *const phs = uni_ptr(phs_);
        const uint32_t plw = uni(plw_), rows = uni(rows_), tile = uni(tile_), C = uni(C_), cmin = uni(cmin_);
        {
                uint2 rp[PER][4]; // (rank before the document's word, the word)
                uint32_t below[PER];
                bool act[PER];
#pragma unroll
                for (uint32_t i = 0; i < PER; ++i) {
                        const uint32_t j = tid + i * AND_WG;
                        act[i] = j < C && sh.alive[j];
                        const uint32_t doc = act[i] ? sh.cdoc[j] : cmin;
                        below[i] = (1u << (doc & 31u)) - 1u;
#pragma unroll
                        for (uint32_t r = 0; r < 4; ++r) {
                                if (r >= rows)
                                        break;
                                rp[i][r] = ((const uint2 *)(prank + (size_t)uni(sh.rrow[r]) * (plw / (PL_RANK_DOCS / 32u)) * PL_RANK_WORDS))[doc >> 5];
                        }
                }
                unsigned long long ent[PER][4];
#pragma unroll
                for (uint32_t i = 0; i < PER; ++i)
#pragma unroll
                        for (uint32_t r = 0; r < 4; ++r) {
                                if (r >= rows)
                                        break;
                                const uint32_t rank = rp[i][r].x + (uint32_t)__popc(rp[i][r].y & below[i]);
                                ent[i][r] = phs[sh.rhs[r] + rank];
                        }
#pragma unroll
                for (uint32_t i = 0; i < PER; ++i) {
                        const uint32_t j = tid + i * AND_WG;
                        if (!act[i])
                                continue;
#pragma unroll
                        for (uint32_t r = 0; r < 4; ++r) {
                                if (r >= rows)
                                        break;
                                sh.hits_off[r * tile + j] = (uint32_t)ent[i][r]; // (this lane's own places: read back below without a barrier)
                                sh.freq[r * tile + j] = (uint32_t)(ent[i][r] >> 32);
                        }
                }
        }
}

#ifndef TRI_PHRASE_WAVES
#define TRI_PHRASE_WAVES 6
#endif
template <int CODEC>
__global__ __launch_bounds__(AND_WG, CODEC == CODEC_GOOGLE ? TRI_PHRASE_WAVES : 3) void k_phrase(const uint8_t *__restrict__ index, const uint8_t *__restrict__ hits, const uint32_t *__restrict__ blk_hits,
                                                   const uint32_t *__restrict__ hdir, const uint32_t *__restrict__ blk_last,
                                                   const uint32_t *__restrict__ blk_off, const uint32_t *__restrict__ win, const DevTerm *__restrict__ terms,
                                                   const DevQuery *__restrict__ plan, const DevTask *__restrict__ tasks,
                                                   const uint32_t *__restrict__ ptasks, const uint32_t nptasks, const DevPhrase *__restrict__ phrases,
                                                   const uint32_t *__restrict__ pterms, uint32_t *__restrict__ ticket, uint32_t *__restrict__ out,
                                                   uint32_t *__restrict__ counts, double *__restrict__ pscore, const uint32_t max_match_cnt,
                                                   const int sim, const uint32_t *__restrict__ planes, const uint32_t plw, const uint32_t *__restrict__ prank,
                                                   const unsigned long long *__restrict__ phs, const uint64_t *__restrict__ hs_off, const uint32_t *__restrict__ term_row) {
        PhraseShared &sh = phrase_shared();
        const uint32_t tid = threadIdx.x;
        const uint32_t wave = uni(tid >> 6);
        const HitCtx ctx{CODEC == CODEC_GOOGLE ? index : hits, blk_hits, hdir};
#ifdef TRI_PHRASE_NOLOCATE
        for (uint32_t i = tid; i < PHRASE_SLOTS; i += AND_WG)
                sh.hits_off[i] = 0, sh.freq[i] = 1u | 0x80000000u;
        __syncthreads();
#endif
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
                if (ticket_no >= nptasks)
                        break;
                const uint32_t tix = ptasks[ticket_no];
                TASKTIME_PHRASE(8 * ticket_no);
                const DevTask task = tasks[tix];
                const DevQuery q = plan[task.slot];
                const uint32_t M = counts[tix];
                uint32_t *seg = out + task.out_off;
                // tile size: PHRASE_SLOTS locator entries shared by the rows (distinct terms) of the query's widest phrase
                uint32_t maxrows = 1;
                for (uint32_t pi = 0; pi < q.nphrases; ++pi) {
                        const DevPhrase ph = phrases[q.phrase_base + pi];
                        uint32_t rows = 0;
                        for (uint32_t k = 0; k < ph.nterms; ++k) {
                                bool seen = false;
                                for (uint32_t m = 0; m < k; ++m)
                                        seen |= pterms[ph.term_base + m] == pterms[ph.term_base + k];
                                rows += !seen;
                        }
                        maxrows = max(maxrows, rows);
                }
                const uint32_t tile = uni(min(PHRASE_TILE, max(64u, (PHRASE_SLOTS / maxrows) & ~63u)));
                uint32_t wpos = 0, task_rows = 0; // task_rows: the rows of the query's one phrase, once they are set up (0: set them up)
                // the NEXT tile's candidates travel while this one is worked on (the compaction below writes behind the read cursor: it never reaches them)
                constexpr uint32_t PER = PHRASE_TILE / AND_WG;
                uint32_t nx[PER];
#pragma unroll
                for (uint32_t i = 0; i < PER; ++i)
                        nx[i] = tid + i * AND_WG < min(tile, M) ? seg[tid + i * AND_WG] : 0u;
                for (uint32_t tb = 0; tb < M; tb += tile) {
                        const uint32_t C = min(tile, M - tb);
#pragma unroll
                        for (uint32_t i = 0; i < PER; ++i) {
                                const uint32_t j = tid + i * AND_WG;
                                if (j < C) {
                                        sh.cdoc[j] = nx[i];
                                        sh.alive[j] = 1;
                                        sh.ps[j] = 0;
                                }
                                nx[i] = tb + tile + j < M && j < tile ? seg[tb + tile + j] : 0u;
                        }
                        __syncthreads();
                        const uint32_t cmin = sh.cdoc[0], cmax = sh.cdoc[C - 1];
                        PROF_LAP(0);
                        for (uint32_t pi = 0; pi < q.nphrases; ++pi) {
                                const DevPhrase ph = phrases[q.phrase_base + pi];
                                // ---- locate: (hits locator, freq) of every candidate for every distinct phrase term.  The rows (distinct terms) are set up
                                //      first — term, the blocks of the tile's docID range —, then ONE pass walks the blocks of ALL the rows, a lane per block
                                //      (round 4 walked row after row with a barrier in between: a two-word phrase of head terms kept 50 and then 30 of the 256
                                //      lanes busy for one block's walk each, twice)
                                uint32_t rows = task_rows, total_blocks = 0;
                                if (!rows) { // (a query with ONE phrase sets its rows up once per task, not once per tile: a round trip and a barrier less per tile)
                                for (uint32_t k = 0; k < ph.nterms; ++k) { // the rows: a term repeated in the phrase shares the row of its first occurrence (uniform stores)
                                        const uint32_t tk = pterms[ph.term_base + k];
                                        uint32_t first = k;
                                        for (uint32_t m = 0; m < k; ++m)
                                                if (pterms[ph.term_base + m] == tk) {
                                                        first = m;
                                                        break;
                                                }
                                        if (first != k) {
                                                sh.row[k] = uni(sh.row[first]);
                                                continue;
                                        }
                                        sh.rtk[rows] = tk;
                                        sh.row[k] = rows++;
                                }
                                if (tid < rows) { // the rows' terms: one load each, together
                                        const DevTerm t = terms[sh.rtk[tid]];
                                        sh.rterm[tid] = t;
                                        sh.rpad[tid] = t.pad;
                                        const uint32_t pr = CODEC == CODEC_GOOGLE && term_row ? term_row[sh.rtk[tid]] : PL_NONE; // the term's plane row, once its rank directory and hits entries are there
                                        sh.rrow[tid] = pr;
                                        sh.rhs[tid] = pr != PL_NONE ? hs_off[pr] : 0ull;
                                }
                                __syncthreads();
                                if (q.nphrases == 1)
                                        task_rows = rows;
                                }
                                // the blocks of the tile's docID range: a WAVE per (row, end of the range) brackets it through the cell index (one round of 64
                                // probes; a list too short for a cell index: two rounds) — the four waves search side by side (round 4: every wave searched
                                // every row's two ends itself, one after the other)
                                for (uint32_t it = wave; it < 2 * rows; it += AND_WG / 64) {
                                        if (uni(sh.rrow[it >> 1]) != PL_NONE)
                                                continue; // (located by rank below: no block range)
                                        const DevTerm t = sh.rterm[it >> 1];
                                        const uint32_t res = phrase_first_block(blk_last + t.first_block, win, t, (it & 1u) ? cmax : cmin);
                                        sh.rbx[it] = res; // (wave-uniform value, every lane stores it)
                                }
                                __syncthreads();
                                for (uint32_t r = 0; r < rows; ++r) { // (uniform stores)
                                        if (uni(sh.rrow[r]) != PL_NONE) {
                                                sh.rb0[r] = 0xffffffffu, sh.rnb[r] = 0, sh.rstart[r] = total_blocks;
                                                continue;
                                        }
                                        const uint32_t nb = uni(sh.rterm[r].nblocks), b0 = uni(sh.rbx[2 * r]);
                                        const uint32_t b1 = b0 < nb ? min(uni(sh.rbx[2 * r + 1]), nb - 1) : b0;
                                        const bool walk = b0 < nb && b1 - b0 + 1 <= 2 * C; // (else: few candidates scattered over a long list, below)
                                        sh.rb0[r] = b0;
                                        sh.rnb[r] = walk ? b1 - b0 + 1 : 0u;
                                        sh.rstart[r] = total_blocks;
                                        total_blocks += walk ? b1 - b0 + 1 : 0u;
                                }
                                PROF_LAP(8);
#ifdef TRI_PHRASE_NOLOCATE // (perf probe: what the kernel costs WITHOUT the block walks — wrong results)
                                if (total_blocks > 0x7fffffffu)
#endif
                                for (uint32_t v = tid; v < total_blocks; v += AND_WG) {
                                        uint32_t r = 0;
                                        for (uint32_t r2 = 1; r2 < rows; ++r2)
                                                r = v >= sh.rstart[r2] ? r2 : r; // (rstart ascends; a row without a walk has an empty range: the next row takes its start)
                                        while (!sh.rnb[r])
                                                ++r;
                                        phrase_locate_block<CODEC>(sh, index, blk_last, blk_off, ctx, sh.rterm[r], sh.rb0[r] + (v - sh.rstart[r]), C, r * tile);
                                }
                                PROF_LAP(9);
                                // every row a head term located by rank (the usual shape of an expensive phrase): the lookups move into the check below — a candidate's
                                // lane fetches the rows' directory entries and plane words TOGETHER, then the rows' entries together, with no barrier in between
                                bool all_rank = CODEC == CODEC_GOOGLE && rows <= 4;
                                for (uint32_t r = 0; r < rows; ++r)
                                        all_rank = all_rank && uni(sh.rrow[r]) != PL_NONE;
                                for (uint32_t r = 0; r < rows; ++r) {
                                        const uint32_t prow = uni(sh.rrow[r]);
                                        if (prow != PL_NONE && all_rank)
                                                continue;
                                        if (prow != PL_NONE) {
                                                // a head term: every candidate's entry by RANK — its posting index is its group's directory entry plus the plane-0 bits of the
                                                // group before it (one 32-byte piece of plane 0), and the posting's entry (k_term_hits) is what a block walk would have left
                                                const uint32_t *rd = prank + (size_t)prow * (plw / (PL_RANK_DOCS / 32u)) * PL_RANK_WORDS;
                                                const unsigned long long *hs = phs + hs_off[prow];
                                                for (uint32_t j = tid; j < C; j += AND_WG) {
                                                        const uint32_t doc = sh.cdoc[j];
                                                        const uint2 rp = ((const uint2 *)rd)[doc >> 5]; // (one pair of the group's record: the rank before the document's word, the word)
                                                        const uint32_t rank = rp.x + (uint32_t)__popc(rp.y & ((1u << (doc & 31u)) - 1u));
                                                        const unsigned long long e = hs[rank];
                                                        sh.hits_off[r * tile + j] = (uint32_t)e;
                                                        sh.freq[r * tile + j] = (uint32_t)(e >> 32);
                                                }
                                                PROF_LAP(10);
                                                continue;
                                        }
                                        if (sh.rnb[r] || sh.rb0[r] >= sh.rterm[r].nblocks) // (uniform)
                                                continue;
                                        // few candidates scattered over a long list: every candidate brackets its block with the two cell-index
                                        // entries of its docID cell (short lists: a bisection of the directory), and the first candidate of a
                                        // block walks it for all the block's candidates
                                        const DevTerm t = sh.rterm[r];
                                        const uint32_t *bl = blk_last + t.first_block;
                                        for (uint32_t j = tid; j < C; j += AND_WG) {
                                                const uint32_t doc = sh.cdoc[j];
                                                uint32_t lo = 0, hi = t.nblocks;
                                                if (t.win_off != 0xffffffffu) {
                                                        lo = win[t.win_off + (doc >> CELL_LOG2)];
                                                        hi = min(win[t.win_off + (doc >> CELL_LOG2) + 1] + 1, t.nblocks);
                                                }
                                                while (lo < hi) {
                                                        const uint32_t mid = (lo + hi) >> 1;
                                                        if (bl[mid] < doc)
                                                                lo = mid + 1;
                                                        else
                                                                hi = mid;
                                                }
                                                const uint32_t prevdoc = lo ? bl[lo - 1] : 0;
                                                if (j == 0 || sh.cdoc[j - 1] <= prevdoc)
                                                        phrase_locate_block<CODEC>(sh, index, blk_last, blk_off, ctx, t, lo, C, r * tile);
                                        }
                                        PROF_LAP(10);
                                }
                                __syncthreads();
                                PROF_LAP(1);
                                // ---- every row a head term: the rows' entries by RANK, for ALL of a lane's candidates at once — first every candidate's and row's rank
                                //      record pair (ONE eight-byte load: the posting index of the first document of its plane-0 word, and the word), then every entry:
                                //      two round trips per tile where a lane that took its candidates one after the other paid
                                //      two per candidate (the kernel waits for these gathers: round 6, 57 % of its cycles in this check at 7 waves per SIMD)
                                if (all_rank) // (uniform)
                                        phrase_rank_entries(prank, phs, plw, rows, tile, C, cmin);
                                // ---- check: one lane per candidate
                                for (uint32_t j = tid; j < C; j += AND_WG) {
                                        if (!sh.alive[j])
                                                continue;
                                        uint32_t cnt = 0;
                                        // GOOGLE, the usual case — up to four distinct terms, every one of the candidate's hit runs at most eight single-byte hits —:
                                        // the rows' hit bytes are fetched TOGETHER, one unaligned 8-byte load each (one round trip for the whole check, where the
                                        // streams below send for a row's bytes when the walk reaches it), and the walks are shifts of registers
                                        bool fast = CODEC == CODEC_GOOGLE && rows <= 4;
                                        uint64_t pw[4] = {0, 0, 0, 0};
                                        uint32_t pf[4] = {0, 0, 0, 0};
                                        if (CODEC == CODEC_GOOGLE && rows <= 4) {
                                                typedef uint64_t ph_u64_a1 __attribute__((aligned(1)));
                                                uint32_t inl = 0; // the rows whose entry holds the hits themselves (FREQ_INLINE): nothing more to fetch
#pragma unroll
                                                for (uint32_t r = 0; r < 4; ++r) {
                                                        if (r >= rows)
                                                                break;
                                                        const uint32_t fe = sh.freq[r * tile + j];
                                                        pf[r] = HitStream<CODEC_GOOGLE>::count(fe);
                                                        if (fe & HitStream<CODEC_GOOGLE>::FREQ_INLINE) {
                                                                pw[r] = HitStream<CODEC_GOOGLE>::inline_bytes(sh.hits_off[r * tile + j], fe);
                                                                inl |= 1u << r;
                                                        } else
                                                                fast = fast && (fe & HitStream<CODEC_GOOGLE>::FREQ_PLAIN) && pf[r] <= 8u;
                                                }
                                                if (fast) {
#pragma unroll
                                                        for (uint32_t r = 0; r < 4; ++r)
                                                                if (r < rows && !((inl >> r) & 1u))
                                                                        pw[r] = *(const ph_u64_a1 *)(index + sh.hits_off[r * tile + j]);
                                                }
                                        }
                                        // ... and where every one of those hits sits below position 64 (short documents and fields; bench.py's ten-slot documents), the
                                        // DocWordsSpace IS a 64-bit word per row: bit p <=> the row's term has a hit at position p.  Ownership (last writer wins: a row
                                        // materialised later takes the slot) is an AND-NOT with the later rows' words, the phrase a shifted AND, matchCnt a popcount —
                                        // a few dozen instructions without a divergent loop, where the walk below runs (hits of term 0) x (terms) x (hits of the term)
                                        // steps and every lane of the wave waits for its slowest (round 6: the kernel's VALU instructions were 4.4 of its 8 ms)
                                        bool masks = false;
                                        if (fast) {
                                                uint64_t hm[4] = {0, 0, 0, 0};
                                                bool ovf = false;
#pragma unroll
                                                for (uint32_t r = 0; r < 4; ++r) {
                                                        if (r >= rows)
                                                                break;
                                                        uint32_t pos = 0;
                                                        uint64_t m = 0;
#pragma unroll
                                                        for (uint32_t h = 0; h < 8; ++h) {
                                                                if (__builtin_amdgcn_ballot_w64(h < pf[r]) == 0ull) // (uniform)
                                                                        break;
                                                                pos += ((uint32_t)(pw[r] >> (8u * h)) & 0xffu) >> 1;
                                                                m |= h < pf[r] ? 1ull << (pos & 63u) : 0ull;
                                                                ovf = ovf || (h < pf[r] && pos >= 64u);
                                                        }
                                                        hm[r] = m;
                                                }
                                                if (!ovf) {
                                                        masks = true;
                                                        uint64_t own[4], later = 0;
#pragma unroll
                                                        for (int r = 3; r >= 0; --r) {
                                                                own[r] = hm[r] & ~later;
                                                                later |= hm[r];
                                                        }
                                                        auto sel = [&](const uint64_t (&a)[4], const uint32_t r) { return r == 0 ? a[0] : r == 1 ? a[1] : r == 2 ? a[2] : a[3]; }; // (r: uniform)
                                                        uint64_t acc = sel(hm, uni(sh.row[0])) & ~1ull; // (a start position 0 is no start: docset_iterators.cpp:101-143)
                                                        for (uint32_t k = 1; k < ph.nterms; ++k)
                                                                acc &= k < 64u ? sel(own, uni(sh.row[k])) >> k : 0ull;
                                                        cnt = min((uint32_t)__popcll(acc), max_match_cnt);
                                                }
                                        }
                                        if (masks) {
                                        } else if (fast) {
                                                auto has_pos = [&](const uint32_t r, const uint32_t q) { // is position q among row r's hits? (positions ascend)
                                                        uint64_t w = r == 0 ? pw[0] : r == 1 ? pw[1] : r == 2 ? pw[2] : pw[3];
                                                        const uint32_t f = r == 0 ? pf[0] : r == 1 ? pf[1] : r == 2 ? pf[2] : pf[3];
                                                        uint32_t pos = 0;
                                                        for (uint32_t h = 0; h < f; ++h) {
                                                                pos += ((uint32_t)w & 0xffu) >> 1;
                                                                w >>= 8;
                                                                if (pos >= q)
                                                                        return pos == q;
                                                        }
                                                        return false;
                                                };
                                                const uint32_t r0 = sh.row[0];
                                                uint64_t w0 = r0 == 0 ? pw[0] : r0 == 1 ? pw[1] : r0 == 2 ? pw[2] : pw[3];
                                                const uint32_t f0 = r0 == 0 ? pf[0] : r0 == 1 ? pf[1] : r0 == 2 ? pf[2] : pf[3];
                                                uint32_t p0 = 0;
                                                for (uint32_t h = 0; h < f0 && cnt < max_match_cnt; ++h) {
                                                        p0 += ((uint32_t)w0 & 0xffu) >> 1;
                                                        w0 >>= 8;
                                                        if (!p0)
                                                                continue;
                                                        bool all = true;
                                                        for (uint32_t k = 1; k < ph.nterms && all; ++k) {
                                                                const uint32_t qpos = p0 + k, rk = sh.row[k];
                                                                all = has_pos(rk, qpos);
                                                                for (uint32_t rm = rk + 1; rm < rows && all; ++rm) // (last writer wins: a row materialised later owns the slot)
                                                                        if (has_pos(rm, qpos))
                                                                                all = false;
                                                        }
                                                        if (all)
                                                                ++cnt;
                                                }
                                        } else {
                                        // walk the start positions of term 0 (docset_iterators.cpp:101-143)
                                        HitStream<CODEC> s0;
                                        s0.init_entry(ctx, terms[pterms[ph.term_base]].pad, sh.hits_off[sh.row[0] * tile + j], sh.freq[sh.row[0] * tile + j]);
                                        uint32_t p0 = 0;
                                        const uint32_t f0 = HitStream<CODEC>::count(sh.freq[sh.row[0] * tile + j]);
                                        for (uint32_t h = 0; h < f0 && cnt < max_match_cnt; ++h) {
                                                p0 = (p0 + s0.next()) & 0xffffu;
                                                if (!p0)
                                                        continue;
                                                // dws->test(term_k, p0 + k): term k has a hit there and no term materialised AFTER it overwrote the slot
                                                // (last writer wins; only first occurrences materialise, in phrase order — and rows are numbered in that
                                                // order, so the later writers are exactly the rows above term k's)
                                                bool all = true;
                                                for (uint32_t k = 1; k < ph.nterms && all; ++k) {
                                                        const uint32_t qpos = p0 + k;
                                                        const uint32_t rk = sh.row[k];
                                                        all = phrase_has_pos<CODEC>(ctx, sh.rpad[rk], sh.hits_off[rk * tile + j], sh.freq[rk * tile + j], qpos);
                                                        for (uint32_t rm = rk + 1; rm < rows && all; ++rm)
                                                                if (phrase_has_pos<CODEC>(ctx, sh.rpad[rm], sh.hits_off[rm * tile + j], sh.freq[rm * tile + j], qpos))
                                                                        all = false;
                                                }
                                                if (all)
                                                        ++cnt;
                                        }
                                        }
                                        if (!cnt)
                                                sh.alive[j] = 0;
                                        // docset_iterators_scorers.cpp:220-224: scorer->score(id, matchCnt, weight)
                                        sh.ps[j] += cnt ? (double)sim_score(sim, ph.weight, cnt) : 0.0;
                                }
                                PROF_LAP(2);
                                __syncthreads();
                                PROF_LAP(3);
                        }
                        // ---- stable in-place compaction of the survivors (the write cursor never passes the read cursor)
                        for (uint32_t base = 0; base < C; base += AND_WG) {
                                const uint32_t j = base + tid;
                                const bool ok = j < C && sh.alive[j];
                                const uint64_t m = __ballot(ok);
                                const uint32_t before = __popcll(m & ((1ull << (tid & 63)) - 1ull));
                                sh.scan[tid >> 6] = __popcll(m);
                                __syncthreads();
                                uint32_t wbase = 0, tot = 0;
                                for (uint32_t w = 0; w < AND_WG / 64; ++w) {
                                        if (w < (tid >> 6))
                                                wbase += sh.scan[w];
                                        tot += sh.scan[w];
                                }
                                tot = uni(tot);
                                if (ok) {
                                        seg[wpos + wbase + before] = sh.cdoc[j];
                                        if (pscore)
                                                pscore[task.out_off + wpos + wbase + before] = sh.ps[j];
                                }
                                wpos += tot;
                                __syncthreads();
                        }
                }
                if (wave == 0)
                        counts[tix] = wpos;
                TASKTIME_PHRASE(8 * ticket_no + 1);
                __syncthreads();
                PROF_LAP(4);
        }
        PROF_FLUSH();
}
