// lucene_enc_units.hpp — the Lucene-shaped codec's encoder (lucene_codec.cpp:163-388: begin_term / begin_document / new_hit / end_document / end_term) cut
// into UNITS that do not depend on each other: what one lane of the device encoder (k_lencode.hpp) does, and — the same functions, compiled by g++ — what
// the CPU tests run in plain loops (csrc/host/plan_host.cpp: tri_host_lucene_encode_units) against the sequential host encoder (csrc/host/lucene_encoder.hpp).
//
// The reference's encoder is a state machine over one term at a time: 128 documents are buffered and flushed as two ints() groups (deltas, frequencies), a
// skiplist entry is remembered at every block's first document, hits are buffered 128 at a time into hits.data, the tail goes out as prefix varints, the
// 14-byte term header is patched at end_term.  Every quantity it carries from call to call is a function of the postings' INDICES:
//   block j of a term     = its postings [128 j, 128 j + 128)                         -> two groups (pfor128_group.hpp), independent of every other block
//   hit block k of a term = its hits [128 k, 128 k + 128) in posting order            -> one group of position deltas (a document's first hit: from 0)
//   skiplist entry j      = {bytes of the blocks before j (+ 14), document before block j, where hit block floor(h / 128) starts, 128 j,
//                            128 floor(h / 128), h mod 128} with h = the term's hits before block j's first document
// so sizes are computed per unit, prefix sums place the units, and every unit writes its own bytes.  Payload-less hits (what the host encoder writes).
// New code, no reference source.
#pragma once
#include "pfor128_group.hpp"

constexpr uint32_t LENC_BLOCK = 128;       // lucene_codec.h:52-55
constexpr uint32_t LENC_TERM_HEADER = 14;  // {u32 hits.data offset, u32 hits, u32 hits.data bytes, u16 skiplist entries}
constexpr uint32_t LENC_SKIP_BYTES = 22;   // lucene_codec.h:128-135
constexpr uint32_t LENC_HIT_TRAILER = 3;   // a full hit block ends with ints(128 zero payload lengths) = {0, 0} and varbyte(0 payload bytes)

struct LencArgs {
        const uint32_t *docs, *freqs;  // postings, term after term
        const uint16_t *pos;           // their hits' positions, posting after posting
        const uint64_t *hit_off;       // [np + 1]: hits before every posting
        const uint64_t *term_first;    // [nterms + 1]: postings before every term
        const uint32_t *hdelta;        // [nhits]: a hit's position less the previous hit's of the same document (unit A)
        const uint64_t *dblk_first;    // [nterms + 1]: full document blocks before every term
        const uint64_t *hblk_first;    // [nterms + 1]: full hit blocks before every term
        uint64_t nterms;
};

// the term that owns global block g (first[] ascends; first[t] <= g < first[t + 1])
TRI_HD inline uint64_t lenc_term_of(const uint64_t *first, const uint64_t nterms, const uint64_t g) {
        uint64_t lo = 0, hi = nterms; // (first[0] = 0 <= g < first[nterms])
        while (hi - lo > 1) {
                const uint64_t mid = (lo + hi) >> 1;
                if (first[mid] <= g)
                        lo = mid;
                else
                        hi = mid;
        }
        return lo; // (first[lo] <= g < first[lo + 1]: a term without blocks — first[t] == first[t + 1] — is never the answer)
}

// ---- unit A, per posting: the position deltas of its hits
TRI_HD inline void lenc_unit_hdelta(const LencArgs &a, const uint64_t p, uint32_t *hdelta) {
        const uint64_t h0 = a.hit_off[p];
        uint32_t last = 0;
        for (uint32_t i = 0; i < a.freqs[p]; ++i) {
                hdelta[h0 + i] = (uint32_t)a.pos[h0 + i] - last;
                last = a.pos[h0 + i];
        }
}

// a document block's two getters
struct LencDelta {
        const uint32_t *docs;
        uint64_t p0, term_p0;
        TRI_HD uint32_t operator()(const uint32_t i) const { return docs[p0 + i] - ((p0 + i) == term_p0 ? 0u : docs[p0 + i - 1]); }
};
struct LencAt {
        const uint32_t *v;
        uint64_t at;
        TRI_HD uint32_t operator()(const uint32_t i) const { return v[at + i]; }
};

// ---- unit B, per full document block g: its bytes
TRI_HD inline uint32_t lenc_unit_dblk_size(const LencArgs &a, const uint64_t g) {
        const uint64_t t = lenc_term_of(a.dblk_first, a.nterms, g), p0 = a.term_first[t] + LENC_BLOCK * (g - a.dblk_first[t]);
        return pfor128_plan(LencDelta{a.docs, p0, a.term_first[t]}).bytes + pfor128_plan(LencAt{a.freqs, p0}).bytes;
}
// ---- unit C, per full hit block h: its bytes
TRI_HD inline uint32_t lenc_unit_hblk_size(const LencArgs &a, const uint64_t h) {
        const uint64_t t = lenc_term_of(a.hblk_first, a.nterms, h), h0 = a.hit_off[a.term_first[t]] + LENC_BLOCK * (h - a.hblk_first[t]);
        return pfor128_plan(LencAt{a.hdelta, h0}).bytes + LENC_HIT_TRAILER;
}
// ---- unit D, per term: the bytes of its varbyte tails (documents that fill no block: (delta, frequency) pairs; hits that fill no block: delta << 1)
TRI_HD inline void lenc_unit_tail_size(const LencArgs &a, const uint64_t t, uint32_t *tail_docs, uint32_t *tail_hits) {
        const uint64_t p_lo = a.term_first[t], p_hi = a.term_first[t + 1], p_tail = p_lo + (p_hi - p_lo) / LENC_BLOCK * LENC_BLOCK;
        uint32_t bd = 0, bh = 0;
        for (uint64_t p = p_tail; p < p_hi; ++p)
                bd += pf_vlen(a.docs[p] - (p == p_lo ? 0u : a.docs[p - 1])) + pf_vlen(a.freqs[p]);
        const uint64_t h_lo = a.hit_off[p_lo], h_hi = a.hit_off[p_hi], h_tail = h_lo + (h_hi - h_lo) / LENC_BLOCK * LENC_BLOCK;
        for (uint64_t h = h_tail; h < h_hi; ++h)
                bh += pf_vlen(a.hdelta[h] << 1);
        *tail_docs = bd;
        *tail_hits = bh;
}

// where the units' bytes go: per-block running sums (doff / hoff: [blocks + 1]) and per-term chunk offsets and sizes
struct LencPlace {
        const uint64_t *doff, *hoff;           // bytes of the document / hit blocks before every block (over all terms)
        const uint64_t *term_off, *hterm_off;  // [nterms + 1]: where a term's index chunk / hits.data chunk starts
        const uint32_t *tail_docs, *tail_hits; // (unit D)
};
TRI_HD inline uint32_t lenc_nskip(const uint64_t nfull) { return (uint32_t)(nfull < 65535 ? nfull : 65535); } // (lucene_encoder.hpp: skiplist.size() < UINT16_MAX)
TRI_HD inline uint32_t lenc_term_index_size(const LencArgs &a, const LencPlace &pl, const uint64_t t) {
        const uint64_t nfull = a.dblk_first[t + 1] - a.dblk_first[t];
        return (uint32_t)(LENC_TERM_HEADER + (pl.doff[a.dblk_first[t + 1]] - pl.doff[a.dblk_first[t]]) + pl.tail_docs[t] + (uint64_t)LENC_SKIP_BYTES * lenc_nskip(nfull));
}
TRI_HD inline uint32_t lenc_term_hits_size(const LencArgs &a, const LencPlace &pl, const uint64_t t) {
        return (uint32_t)((pl.hoff[a.hblk_first[t + 1]] - pl.hoff[a.hblk_first[t]]) + pl.tail_hits[t]);
}
TRI_HD inline uint8_t *lenc_put32(uint8_t *o, const uint32_t v) {
        o[0] = (uint8_t)v, o[1] = (uint8_t)(v >> 8), o[2] = (uint8_t)(v >> 16), o[3] = (uint8_t)(v >> 24);
        return o + 4;
}

// ---- unit E, per full document block g: its two groups, and its skiplist entry
TRI_HD inline void lenc_unit_dblk_write(const LencArgs &a, const LencPlace &pl, const uint64_t g, uint8_t *index_out) {
        const uint64_t t = lenc_term_of(a.dblk_first, a.nterms, g), j = g - a.dblk_first[t], p0 = a.term_first[t] + LENC_BLOCK * j;
        const uint32_t in_chunk = (uint32_t)(LENC_TERM_HEADER + (pl.doff[g] - pl.doff[a.dblk_first[t]]));
        uint8_t *o = index_out + pl.term_off[t] + in_chunk;
        const LencDelta gd{a.docs, p0, a.term_first[t]};
        o = pfor128_emit(gd, pfor128_plan(gd), o);
        const LencAt gf{a.freqs, p0};
        pfor128_emit(gf, pfor128_plan(gf), o);
        const uint64_t nfull = a.dblk_first[t + 1] - a.dblk_first[t];
        if (j < lenc_nskip(nfull)) {
                // what the encoder remembered at the block's first document (lucene_encoder.hpp begin_document): where the block starts in the chunk, the document
                // before it, the hit block the term's hits had reached — where it starts in the term's hits.data chunk, the hits before it, the hits inside it
                const uint64_t hb = a.hit_off[p0] - a.hit_off[a.term_first[t]], hblk = hb / LENC_BLOCK;
                uint8_t *s = index_out + pl.term_off[t] + lenc_term_index_size(a, pl, t) - (uint64_t)LENC_SKIP_BYTES * lenc_nskip(nfull) + LENC_SKIP_BYTES * j;
                s = lenc_put32(s, in_chunk);
                s = lenc_put32(s, j ? a.docs[p0 - 1] : 0u);
                s = lenc_put32(s, (uint32_t)(pl.hoff[a.hblk_first[t] + hblk] - pl.hoff[a.hblk_first[t]]));
                s = lenc_put32(s, (uint32_t)(LENC_BLOCK * j));
                s = lenc_put32(s, (uint32_t)(hblk * LENC_BLOCK));
                s[0] = (uint8_t)(hb % LENC_BLOCK), s[1] = 0;
        }
}
// ---- unit F, per full hit block h
TRI_HD inline void lenc_unit_hblk_write(const LencArgs &a, const LencPlace &pl, const uint64_t h, uint8_t *hits_out) {
        const uint64_t t = lenc_term_of(a.hblk_first, a.nterms, h), h0 = a.hit_off[a.term_first[t]] + LENC_BLOCK * (h - a.hblk_first[t]);
        uint8_t *o = hits_out + pl.hterm_off[t] + (pl.hoff[h] - pl.hoff[a.hblk_first[t]]);
        const LencAt gh{a.hdelta, h0};
        o = pfor128_emit(gh, pfor128_plan(gh), o);
        o[0] = 0, o[1] = 0, o[2] = 0; // ints(128 payload lengths, all zero) = {0, varbyte 0}; varbyte(0 payload bytes)
}
// ---- unit G, per term: the header and the two tails
TRI_HD inline void lenc_unit_term_write(const LencArgs &a, const LencPlace &pl, const uint64_t t, uint8_t *index_out, uint8_t *hits_out) {
        const uint64_t p_lo = a.term_first[t], p_hi = a.term_first[t + 1], nfull = a.dblk_first[t + 1] - a.dblk_first[t];
        const uint64_t h_lo = a.hit_off[p_lo], h_hi = a.hit_off[p_hi];
        uint8_t *o = index_out + pl.term_off[t];
        o = lenc_put32(o, (uint32_t)pl.hterm_off[t]);
        o = lenc_put32(o, (uint32_t)(h_hi - h_lo));
        o = lenc_put32(o, lenc_term_hits_size(a, pl, t));
        const uint32_t nskip = lenc_nskip(nfull);
        o[0] = (uint8_t)nskip, o[1] = (uint8_t)(nskip >> 8);
        o = index_out + pl.term_off[t] + LENC_TERM_HEADER + (pl.doff[a.dblk_first[t + 1]] - pl.doff[a.dblk_first[t]]);
        for (uint64_t p = p_lo + nfull * LENC_BLOCK; p < p_hi; ++p) {
                o = pf_put_varbyte(o, a.docs[p] - (p == p_lo ? 0u : a.docs[p - 1]));
                o = pf_put_varbyte(o, a.freqs[p]);
        }
        uint8_t *ho = hits_out + pl.hterm_off[t] + (pl.hoff[a.hblk_first[t + 1]] - pl.hoff[a.hblk_first[t]]);
        for (uint64_t h = h_lo + (h_hi - h_lo) / LENC_BLOCK * LENC_BLOCK; h < h_hi; ++h)
                ho = pf_put_varbyte(ho, a.hdelta[h] << 1);
}
