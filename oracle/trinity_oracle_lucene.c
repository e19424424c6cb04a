/*
 * trinity_oracle_lucene.c — CPU ORACLE, Lucene-shaped codec (test infrastructure only; see trinity_oracle.h).
 *
 * Restates the CONTAINER logic of the reference's lucene_codec.cpp: term header, 128-document blocks as two ints()
 * groups, varbyte tail, per-block 22-byte skiplist entries, hits.data framing, and the iterator's next()/advance()
 * with its skiplist seek.  The ints() payload itself is delegated by the reference to lemire/FastPFor
 * (lucene_codec.cpp:57-64, 91-95), an absent un-vendored submodule with no pinned version: this repo defines its own
 * PFOR128 payload (include/pfor128.md) instead, so the payload bytes are PARITY UNPINNED.  What IS pinned: results on a
 * Lucene-coded segment must equal the results on the Google-coded segment of the same corpus (tests/test_oracle.py),
 * and the Google side is pinned to the genuine reference.
 */
#include "oracle_internal.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LBLOCK 128 /* lucene_codec.h:52-55 BLOCK_SIZE (FastPFor / StreamVByte builds) */
#define SKIP_ENTRY 22 /* lucene_codec.cpp:877-894: 5 x u32 + u16 */

/* ================================================================== PFOR128 (this repo's payload format) */
static uint32_t bitlen(uint32_t v) { return v ? 32u - (uint32_t)__builtin_clz(v) : 0u; }

/* lucene_codec.cpp:26-66 ints_encode: all-equal => u8 0 + varbyte; else u8 L + L payload words */
size_t to_ints_encode(const uint32_t *v, uint8_t *out) {
        int eq = 1;
        for (int i = 1; i < LBLOCK; ++i)
                eq &= v[i] == v[0];
        if (eq) {
                out[0] = 0;
                return 1 + to_varbyte_put32(out + 1, v[0]);
        }
        /* choose the packed width b that minimises the word count; ties -> smaller b */
        uint32_t best_b = 32, best_cost = 4 * 32, best_nexc = 0, best_eb = 0;
        for (uint32_t b = 0; b < 32; ++b) {
                uint32_t nexc = 0, mx = 0;
                for (int i = 0; i < LBLOCK; ++i)
                        if ((v[i] >> b) != 0) {
                                ++nexc;
                                if ((v[i] >> b) > mx)
                                        mx = v[i] >> b;
                        }
                const uint32_t eb = bitlen(mx);
                const uint32_t cost = 4 * b + (nexc + 3) / 4 + (nexc * eb + 31) / 32;
                if (cost < best_cost) {
                        best_cost = cost;
                        best_b = b;
                        best_nexc = nexc;
                        best_eb = eb;
                }
        }
        const uint32_t b = best_b, nexc = best_nexc, eb = best_eb, L = 1 + best_cost;
        uint32_t w[1 + 128 + 32 + 128];
        memset(w, 0, sizeof w);
        w[0] = b | (nexc << 8) | (eb << 16);
        uint32_t *packed = w + 1, *epos = w + 1 + 4 * b, *ehigh = epos + (nexc + 3) / 4;
        uint32_t e = 0;
        for (uint32_t i = 0; i < LBLOCK; ++i) {
                if (b) {
                        const uint64_t low = b == 32 ? v[i] : (v[i] & ((1u << b) - 1u));
                        const uint32_t bit = i * b;
                        packed[bit >> 5] |= (uint32_t)(low << (bit & 31));
                        if ((bit & 31) + b > 32)
                                packed[(bit >> 5) + 1] |= (uint32_t)(low >> (32 - (bit & 31)));
                }
                if (b < 32 && (v[i] >> b) != 0) {
                        epos[e >> 2] |= i << ((e & 3) * 8);
                        const uint64_t high = v[i] >> b;
                        const uint32_t bit = e * eb;
                        ehigh[bit >> 5] |= (uint32_t)(high << (bit & 31));
                        if ((bit & 31) + eb > 32)
                                ehigh[(bit >> 5) + 1] |= (uint32_t)(high >> (32 - (bit & 31)));
                        ++e;
                }
        }
        out[0] = (uint8_t)L;
        memcpy(out + 1, w, (size_t)L * 4);
        return 1 + (size_t)L * 4;
}

/* lucene_codec.cpp:69-100 ints_decode */
size_t to_ints_decode(const uint8_t *in, uint32_t *v) {
        const uint32_t L = in[0];
        if (!L) {
                uint32_t x;
                const size_t n = to_varbyte_get32(in + 1, &x);
                for (int i = 0; i < LBLOCK; ++i)
                        v[i] = x;
                return 1 + n;
        }
        uint32_t w[256];
        memcpy(w, in + 1, (size_t)L * 4);
        w[L] = 0;
        const uint32_t b = w[0] & 0xff, nexc = (w[0] >> 8) & 0xff, eb = (w[0] >> 16) & 0xff;
        const uint32_t *packed = w + 1, *epos = w + 1 + 4 * b, *ehigh = epos + (nexc + 3) / 4;
        for (uint32_t i = 0; i < LBLOCK; ++i) {
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
                const uint32_t high = (uint32_t)((win >> (bit & 31)) & (eb == 32 ? 0xffffffffull : ((1ull << eb) - 1)));
                v[pos & 127] |= high << b;
        }
        return 1 + (size_t)L * 4;
}

static void buf_ints(buf_t *b, const uint32_t *v) {
        buf_room(b, 1 + 4 * 300);
        b->n += to_ints_encode(v, b->d + b->n);
}

/* ================================================================== writer (lucene_codec.cpp:163-388) */
typedef struct {
        uint32_t indexOffset, lastDocID, lastHitsBlockOffset, totalDocumentsSoFar, lastHitsBlockTotalHits;
        uint16_t curHitsBlockHits;
} lskip_t; /* lucene_codec.h:128-135 */

typedef struct {
        buf_t *indexOut, *positionsOut;
        lskip_t *skiplist;
        size_t nskip, capskip;
        uint32_t lastDocID;
        uint32_t docDeltas[LBLOCK], docFreqs[LBLOCK], hitPayloadSizes[LBLOCK], hitPosDeltas[LBLOCK];
        uint32_t buffered, totalHits, sumHits, termDocuments;
        uint16_t lastPosition;
        uint32_t termIndexOffset, termPositionsOffset;
        uint32_t skiplistCountdown, lastHitsBlockOffset, lastHitsBlockTotalHits;
        lskip_t cur_block;
} lenc_t;

static void put_u32_at(buf_t *b, size_t at, uint32_t v) { memcpy(b->d + at, &v, 4); }
static void put_u16_at(buf_t *b, size_t at, uint16_t v) { memcpy(b->d + at, &v, 2); }
static void buf_u16(buf_t *b, uint16_t v) {
        buf_room(b, 2);
        memcpy(b->d + b->n, &v, 2);
        b->n += 2;
}

/* lucene_codec.cpp:163-181 */
static void lenc_begin_term(lenc_t *e) {
        e->lastDocID = 0;
        e->totalHits = 0;
        e->sumHits = 0;
        e->buffered = 0;
        e->termDocuments = 0;
        e->termIndexOffset = (uint32_t)e->indexOut->n;
        e->termPositionsOffset = (uint32_t)e->positionsOut->n;
        e->lastHitsBlockOffset = 0;
        e->lastHitsBlockTotalHits = 0;
        e->skiplistCountdown = 1; /* SKIPLIST_STEP, lucene_codec.h:57 */
        e->nskip = 0;
        buf_u32(e->indexOut, e->termPositionsOffset);
        buf_u32(e->indexOut, 0);
        buf_u32(e->indexOut, 0);
        buf_u16(e->indexOut, 0);
}

/* lucene_codec.cpp:183-210 */
static void lenc_output_block(lenc_t *e) {
        if (--e->skiplistCountdown == 0) {
                if (e->nskip < UINT16_MAX) {
                        if (e->nskip == e->capskip) {
                                e->capskip = e->capskip ? e->capskip * 2 : 64;
                                e->skiplist = (lskip_t *)xrealloc(e->skiplist, e->capskip * sizeof(lskip_t));
                        }
                        e->skiplist[e->nskip++] = e->cur_block;
                }
                e->skiplistCountdown = 1;
        }
        buf_ints(e->indexOut, e->docDeltas);
        buf_ints(e->indexOut, e->docFreqs);
        e->buffered = 0;
}

/* lucene_codec.cpp:212-243 */
static void lenc_begin_document(lenc_t *e, uint32_t documentID) {
        if (documentID <= e->lastDocID) {
                fprintf(stderr, "trinity_oracle: documentID %u <= %u\n", documentID, e->lastDocID);
                abort();
        }
        if (e->buffered == LBLOCK)
                lenc_output_block(e);
        if (!e->buffered) {
                e->cur_block.indexOffset = (uint32_t)e->indexOut->n - e->termIndexOffset;
                e->cur_block.lastDocID = e->lastDocID;
                e->cur_block.totalDocumentsSoFar = e->termDocuments;
                e->cur_block.lastHitsBlockOffset = e->lastHitsBlockOffset;
                e->cur_block.lastHitsBlockTotalHits = e->lastHitsBlockTotalHits;
                e->cur_block.curHitsBlockHits = (uint16_t)e->totalHits;
        }
        e->docDeltas[e->buffered] = documentID - e->lastDocID;
        e->docFreqs[e->buffered] = 0;
        ++e->termDocuments;
        e->lastDocID = documentID;
        e->lastPosition = 0;
}

/* lucene_codec.cpp:245-307, payload-less hits */
static void lenc_new_hit(lenc_t *e, uint32_t pos) {
        if (!pos)
                return;
        const uint32_t delta = pos - e->lastPosition;
        ++e->docFreqs[e->buffered];
        e->hitPosDeltas[e->totalHits] = delta;
        e->hitPayloadSizes[e->totalHits] = 0;
        e->lastPosition = (uint16_t)pos;
        ++e->totalHits;
        if (e->totalHits == LBLOCK) {
                e->sumHits += e->totalHits;
                buf_ints(e->positionsOut, e->hitPosDeltas);
                buf_ints(e->positionsOut, e->hitPayloadSizes);
                buf_varbyte(e->positionsOut, 0); /* payloadsBuf.size() */
                e->lastHitsBlockTotalHits = e->sumHits;
                e->lastHitsBlockOffset = (uint32_t)e->positionsOut->n - e->termPositionsOffset;
                e->totalHits = 0;
        }
}

/* lucene_codec.cpp:309-311 */
static void lenc_end_document(lenc_t *e) { ++e->buffered; }

/* lucene_codec.cpp:313-388 */
static void lenc_end_term(lenc_t *e, to_term *out) {
        e->sumHits += e->totalHits;
        if (e->buffered == LBLOCK)
                lenc_output_block(e);
        else
                for (uint32_t i = 0; i != e->buffered; ++i) {
                        buf_varbyte(e->indexOut, e->docDeltas[i]);
                        buf_varbyte(e->indexOut, e->docFreqs[i]);
                }
        put_u32_at(e->indexOut, e->termIndexOffset + 4, e->sumHits);
        if (e->totalHits) {
                uint8_t lastPayloadLen = 0;
                for (uint32_t i = 0; i != e->totalHits; ++i) {
                        const uint32_t posDelta = e->hitPosDeltas[i];
                        const uint8_t payloadLen = (uint8_t)e->hitPayloadSizes[i];
                        if (payloadLen != lastPayloadLen) {
                                lastPayloadLen = payloadLen;
                                buf_varbyte(e->positionsOut, (posDelta << 1) | 1);
                                buf_u8(e->positionsOut, payloadLen);
                        } else
                                buf_varbyte(e->positionsOut, posDelta << 1);
                }
        }
        const uint16_t skiplistSize = (uint16_t)e->nskip;
        put_u32_at(e->indexOut, e->termIndexOffset + 8, (uint32_t)e->positionsOut->n - e->termPositionsOffset);
        put_u16_at(e->indexOut, e->termIndexOffset + 12, skiplistSize);
        for (size_t i = 0; i < e->nskip; ++i) {
                const lskip_t *s = &e->skiplist[i];
                buf_u32(e->indexOut, s->indexOffset);
                buf_u32(e->indexOut, s->lastDocID);
                buf_u32(e->indexOut, s->lastHitsBlockOffset);
                buf_u32(e->indexOut, s->totalDocumentsSoFar);
                buf_u32(e->indexOut, s->lastHitsBlockTotalHits);
                buf_u16(e->indexOut, s->curHitsBlockHits);
        }
        e->nskip = 0;
        out->documents = e->termDocuments;
        out->offset = e->termIndexOffset;
        out->size = (uint32_t)e->indexOut->n - e->termIndexOffset;
}

to_index *to_lucene_encode(const to_corpus *c) {
        to_index *ix = (to_index *)xcalloc(1, sizeof *ix);
        buf_t out = {0, 0, 0}, pos = {0, 0, 0};
        lenc_t e;
        memset(&e, 0, sizeof e);
        e.indexOut = &out;
        e.positionsOut = &pos;
        ix->terms = (to_term *)xcalloc(c->V, sizeof(to_term));
        ix->nterms = c->V;
        for (uint32_t t = 0; t < c->V; ++t) {
                const uint64_t b = c->term_off[t], end = c->term_off[t + 1];
                if (b == end)
                        continue;
                lenc_begin_term(&e);
                for (uint64_t i = b; i < end;) {
                        const uint32_t d = c->tok_doc[i];
                        lenc_begin_document(&e, d);
                        for (; i < end && c->tok_doc[i] == d; ++i)
                                lenc_new_hit(&e, c->tok_pos[i]);
                        lenc_end_document(&e);
                }
                lenc_end_term(&e, &ix->terms[t]);
                ix->totalTerms++;
                ix->sumTermsDocs += ix->terms[t].documents;
        }
        buf_room(&out, 64);
        memset(out.d + out.n, 0, 64);
        buf_room(&pos, 64);
        memset(pos.d + pos.n, 0, 64);
        ix->bytes = out.d;
        ix->len = out.n;
        ix->hits = pos.d;
        ix->hits_len = pos.n;
        ix->sumTermHits = c->ntokens;
        ix->docsCnt = c->D;
        ix->owns = 1;
        ix->codec = TO_CODEC_LUCENE;
        free(e.skiplist);
        return ix;
}

/* ================================================================== reader */
typedef struct { /* lucene_codec.h:254-261 Decoder::skiplist_entry */
        uint32_t indexOffset, lastDocID, lastHitsBlockOffset, totalDocumentsSoFar, totalHitsSoFar;
        uint16_t curHitsBlockHits;
} lsk_t;

typedef struct to_lpli { /* lucene_codec.h:214-250 PostingsListIterator + 252-340 Decoder */
        TO_PLI_HEAD
        /* decoder */
        const uint8_t *postingListBase, *chunkEnd, *hitsBase;
        lsk_t *skiplist;
        uint32_t skiplistSize;
        uint32_t totalDocuments, totalHits;
        /* iterator */
        const uint8_t *p;
        uint32_t lastDocID, docsLeft;
        uint16_t docsIndex, bufferedDocs;
        uint32_t docDeltas[LBLOCK + 1], docFreqs[LBLOCK + 1];
        uint32_t skipListIdx, curSkipListLastDocID;
        /* hits: the reference threads hdp/hitsIndex/skippedHits through next()/advance() (lucene_codec.cpp:401-513); the
         * oracle keeps the equivalent absolute index of the current document's first hit and decodes hits.data from the
         * term's start on demand — same bytes, simpler bookkeeping */
        uint64_t hitAbs;
        uint32_t *allPos; /* lazily: every hit position delta of the term, in order */
} to_lpli;

/* lucene_codec.h:313-319 */
static void l_update_curdoc(to_lpli *it) {
        it->it.cur = it->lastDocID + it->docDeltas[it->docsIndex];
        it->freq = (uint16_t)it->docFreqs[it->docsIndex];
}

/* lucene_codec.cpp:515-558 */
static void l_refill_documents(to_lpli *it) {
        if (it->docsLeft >= LBLOCK) {
                it->p += to_ints_decode(it->p, it->docDeltas);
                it->p += to_ints_decode(it->p, it->docFreqs);
                it->bufferedDocs = LBLOCK;
                it->docsLeft -= LBLOCK;
        } else {
                const uint8_t *p = it->p;
                const uint32_t docsLeft = it->docsLeft;
                for (uint32_t i = 0; i != docsLeft; ++i) {
                        uint32_t v;
                        p += to_varbyte_get32(p, &v);
                        it->docDeltas[i] = v;
                        p += to_varbyte_get32(p, &v);
                        it->docFreqs[i] = v;
                }
                it->p = p;
                it->bufferedDocs = (uint16_t)docsLeft;
                it->docsLeft = 0;
        }
        it->docsIndex = 0;
        l_update_curdoc(it);
}

/* lucene_codec.cpp:568-594 */
static uint32_t l_next(to_iter *self) {
        to_lpli *it = (to_lpli *)self;
        uint16_t idx = it->docsIndex;
        it->hitAbs += it->docFreqs[idx]; /* it->skippedHits += docFreqs[idx] */
        it->lastDocID += it->docDeltas[idx++];
        if (idx >= it->bufferedDocs) {
                if (it->p != it->chunkEnd) {
                        it->docsIndex = idx;
                        l_refill_documents(it); /* decode_next_block */
                        idx = it->docsIndex;
                } else {
                        it->it.cur = TO_DOCIDS_END; /* finalize */
                        it->docsIndex = idx;
                        return it->it.cur;
                }
        }
        it->it.cur = it->lastDocID + it->docDeltas[idx];
        it->freq = (uint16_t)it->docFreqs[idx];
        it->docsIndex = idx;
        return it->it.cur;
}

/* lucene_codec.cpp:596-656 (the branch-free variant that is compiled in) */
static uint32_t l_skiplist_search(const to_lpli *it, uint32_t target) {
        const uint32_t idx = it->skipListIdx;
        const lsk_t *data = it->skiplist + idx;
        uint32_t n = it->skiplistSize - idx;
        for (uint32_t h; (h = n / 2) != 0;) {
                const lsk_t *m = data + h;
                data = (m->lastDocID < target) ? m : data;
                n -= h;
        }
        return target > data->lastDocID ? (uint32_t)(data - it->skiplist) : UINT32_MAX;
}

/* lucene_codec.cpp:658-765 */
static uint32_t l_advance(to_iter *self, uint32_t target) {
        to_lpli *it = (to_lpli *)self;
        uint16_t localBufferedDocs = it->bufferedDocs;
        uint16_t docsIndex = it->docsIndex;
        int seek_early = target > it->curSkipListLastDocID; /* LUCENE_SKIPLIST_SEEK_EARLY */
        for (;;) {
                if (seek_early || docsIndex == localBufferedDocs) {
                        if (!seek_early && it->p == it->chunkEnd) {
                                it->it.cur = TO_DOCIDS_END;
                                it->docsIndex = docsIndex;
                                return it->it.cur;
                        }
                        if (seek_early || it->skipListIdx != it->skiplistSize) {
                                seek_early = 0;
                                const uint32_t index = l_skiplist_search(it, target);
                                if (index != UINT32_MAX) {
                                        it->skipListIdx = index + 1;
                                        it->curSkipListLastDocID = it->skipListIdx == it->skiplistSize ? TO_DOCIDS_END : it->skiplist[it->skipListIdx].lastDocID;
                                        const lsk_t *r = &it->skiplist[index];
                                        it->p = it->postingListBase + r->indexOffset;
                                        it->lastDocID = r->lastDocID;
                                        it->docsLeft = it->totalDocuments - r->totalDocumentsSoFar;
                                        l_refill_documents(it);
                                        it->hitAbs = (uint64_t)r->totalHitsSoFar + r->curHitsBlockHits;
                                        localBufferedDocs = it->bufferedDocs;
                                        docsIndex = it->docsIndex;
                                        goto l10;
                                }
                                /* not found: when we came here early (the block is not exhausted) keep scanning it */
                                if (docsIndex != localBufferedDocs)
                                        goto l10;
                                if (it->p == it->chunkEnd) {
                                        it->it.cur = TO_DOCIDS_END;
                                        it->docsIndex = docsIndex;
                                        return it->it.cur;
                                }
                        }
                        l_refill_documents(it); /* decode_next_block */
                        localBufferedDocs = it->bufferedDocs;
                        docsIndex = it->docsIndex;
                } else {
                l10:
                        if (it->it.cur >= target) {
                                it->docsIndex = docsIndex;
                                return it->it.cur;
                        }
                        it->hitAbs += it->docFreqs[docsIndex];
                        it->lastDocID += it->docDeltas[docsIndex];
                        ++docsIndex;
                        it->it.cur = it->lastDocID + it->docDeltas[docsIndex];
                        it->freq = (uint16_t)it->docFreqs[docsIndex];
                }
        }
}

/* Every position delta of the term, decoded once from hits.data (lucene_codec.cpp:401-462 refill_hits framing):
 * full blocks of 128 hits = ints(posDeltas) ints(payloadLens) varbyte(payloadBytes) payload; tail = varbyte
 * (posDelta << 1 | newLen) [u8 len] ... then the payload bytes. */
static void l_load_positions(to_lpli *it) {
        it->allPos = (uint32_t *)xmalloc(sizeof(uint32_t) * ((size_t)it->totalHits + LBLOCK));
        const uint8_t *p = it->hitsBase;
        uint32_t left = it->totalHits, n = 0, lens[LBLOCK];
        while (left >= LBLOCK) {
                p += to_ints_decode(p, it->allPos + n);
                p += to_ints_decode(p, lens);
                uint32_t payloadBytes;
                p += to_varbyte_get32(p, &payloadBytes);
                p += payloadBytes;
                n += LBLOCK;
                left -= LBLOCK;
        }
        uint8_t payloadLen = 0;
        for (uint32_t i = 0; i < left; ++i) {
                uint32_t v;
                p += to_varbyte_get32(p, &v);
                if (v & 1)
                        payloadLen = *p++;
                (void)payloadLen;
                it->allPos[n++] = v >> 1;
        }
}

/* lucene_codec.cpp:767-856 (positions only) */
static uint32_t l_materialize(to_pli *self, uint16_t *out) {
        to_lpli *it = (to_lpli *)self;
        if (!it->allPos)
                l_load_positions(it);
        const uint32_t freq = it->docFreqs[it->docsIndex];
        uint16_t pos = 0;
        for (uint32_t i = 0; i < freq; ++i) {
                pos = (uint16_t)(pos + it->allPos[it->hitAbs + i]);
                out[i] = pos;
        }
        it->hitAbs += freq;                /* the reference has consumed these hits … */
        it->docFreqs[it->docsIndex] = 0; /* … and zeroes the freq so next() does not skip them again (:855) */
        return freq;
}

static void l_destroy(to_pli *self) {
        to_lpli *it = (to_lpli *)self;
        free(it->skiplist);
        free(it->allPos);
}

/* lucene_codec.cpp:896-932 (init) + 877-894 (init_skiplist) + 858-875 (new_iterator) */
to_pli *to_lucene_pli_new(const to_index *ix, uint32_t term) {
        to_lpli *it = (to_lpli *)xcalloc(1, sizeof *it);
        const to_term *t = &ix->terms[term];
        it->it.type = IT_PLI;
        it->it.next = l_next;
        it->it.advance = l_advance;
        it->it.score = to_pli_score_bm25;
        it->it.cost = t->documents;
        it->term = term;
        it->documents = t->documents;
        it->materialize = l_materialize;
        it->destroy = l_destroy;
        it->sim = ix->similarity;
        it->idf = to_sim_weight(ix->similarity, t->documents, ix->docsCnt);
        it->curSkipListLastDocID = TO_DOCIDS_END;
        if (!t->size) {
                it->it.cur = TO_DOCIDS_END;
                it->postingListBase = it->chunkEnd = it->p = ix->bytes;
                return (to_pli *)it;
        }
        const uint8_t *ptr = ix->bytes + t->offset;
        uint32_t hitsDataOffset;
        uint16_t skiplistSize;
        memcpy(&hitsDataOffset, ptr, 4);
        memcpy(&it->totalHits, ptr + 4, 4);
        memcpy(&skiplistSize, ptr + 12, 2);
        it->postingListBase = ptr;
        it->chunkEnd = ptr + t->size - (size_t)skiplistSize * SKIP_ENTRY;
        it->totalDocuments = t->documents;
        it->hitsBase = ix->hits + hitsDataOffset;
        it->skiplistSize = skiplistSize;
        if (skiplistSize) {
                it->skiplist = (lsk_t *)xmalloc(sizeof(lsk_t) * skiplistSize);
                const uint8_t *s = it->chunkEnd;
                for (uint32_t i = 0; i < skiplistSize; ++i, s += SKIP_ENTRY) {
                        lsk_t *e = &it->skiplist[i];
                        memcpy(&e->indexOffset, s, 4);
                        memcpy(&e->lastDocID, s + 4, 4);
                        memcpy(&e->lastHitsBlockOffset, s + 8, 4);
                        memcpy(&e->totalDocumentsSoFar, s + 12, 4);
                        memcpy(&e->totalHitsSoFar, s + 16, 4);
                        memcpy(&e->curHitsBlockHits, s + 20, 2);
                }
        }
        it->lastDocID = 0;
        it->docsLeft = it->totalDocuments;
        it->p = ptr + 14;
        return (to_pli *)it;
}
