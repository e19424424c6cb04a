/*
 * trinity_oracle.c — CPU ORACLE (test infrastructure only; see trinity_oracle.h).
 *
 * Plain-C restatement of the reference's document-at-a-time execution path.  Nothing here is
 * reachable from the product path.  Citations are file:line in the reference tree.
 */
#include "trinity_oracle.h"
#include "oracle_internal.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define GOOGLE_N 32            /* google_codec.h:18 */
#define GOOGLE_SKIPLIST_STEP 8 /* google_codec.h:19  (256 / N) */
#define SPAN_SHIFT 13          /* docset_spans.h:74 */
#define SPAN_SIZE (1u << SPAN_SHIFT)
#define SPAN_MASK (SPAN_SIZE - 1)
#define MAX_POSITION (1u << 14) /* trinity_limits.h:15 */
#define MAX_PHRASE 16           /* trinity_limits.h:12 */

void *xmalloc(size_t n) {
        void *p = malloc(n ? n : 1);
        if (!p) {
                fprintf(stderr, "trinity_oracle: out of memory (%zu)\n", n);
                abort();
        }
        return p;
}
void *xcalloc(size_t n, size_t s) {
        void *p = calloc(n ? n : 1, s ? s : 1);
        if (!p) {
                fprintf(stderr, "trinity_oracle: out of memory\n");
                abort();
        }
        return p;
}
void *xrealloc(void *q, size_t n) {
        void *p = realloc(q, n ? n : 1);
        if (!p) {
                fprintf(stderr, "trinity_oracle: out of memory\n");
                abort();
        }
        return p;
}

/* ================================================================== a1: prefix varint */
/* Switch/switch_compiler_aux.h:23-51 */
size_t to_varbyte_put32(uint8_t *op, uint32_t x) {
        if (x < (1u << 7)) {
                op[0] = (uint8_t)x;
                return 1;
        } else if (x < (1u << 14)) { /* bswap16(x | 0x8000): big-endian 14 bit */
                op[0] = (uint8_t)((x >> 8) | 0x80u);
                op[1] = (uint8_t)x;
                return 2;
        } else if (x < (1u << 21)) { /* high 5 bits, then low 16 bits little-endian */
                op[0] = (uint8_t)((x >> 16) | 0xc0u);
                op[1] = (uint8_t)x;
                op[2] = (uint8_t)(x >> 8);
                return 3;
        } else if (x < (1u << 28)) { /* bswap32(x | 0xe0000000): big-endian 28 bit */
                op[0] = (uint8_t)((x >> 24) | 0xe0u);
                op[1] = (uint8_t)(x >> 16);
                op[2] = (uint8_t)(x >> 8);
                op[3] = (uint8_t)x;
                return 4;
        } else { /* (u64)x >> 32 | 0xf0 == 0xf0, then little-endian u32 */
                op[0] = 0xf0u;
                op[1] = (uint8_t)x;
                op[2] = (uint8_t)(x >> 8);
                op[3] = (uint8_t)(x >> 16);
                op[4] = (uint8_t)(x >> 24);
                return 5;
        }
}

/* Switch/switch_compiler_aux.h:53-80 */
size_t to_varbyte_get32(const uint8_t *ip, uint32_t *v) {
        uint32_t x = ip[0];
        if (!(x & 0x80u)) {
                *v = x;
                return 1;
        } else if (!(x & 0x40u)) {
                *v = ((x & 0x3fu) << 8) | ip[1];
                return 2;
        } else if (!(x & 0x20u)) {
                *v = ((x & 0x1fu) << 16) | ip[1] | ((uint32_t)ip[2] << 8);
                return 3;
        } else if (!(x & 0x10u)) {
                *v = ((x & 0x0fu) << 24) | ((uint32_t)ip[1] << 16) | ((uint32_t)ip[2] << 8) | ip[3];
                return 4;
        } else { /* ((x & 7) << 32) truncated to 32 bit | ctou32(ip+1) */
                *v = ip[1] | ((uint32_t)ip[2] << 8) | ((uint32_t)ip[3] << 16) | ((uint32_t)ip[4] << 24);
                return 5;
        }
}

/* ================================================================== corpus generator */
uint64_t to_splitmix64(uint64_t *s) {
        uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
}

struct to_zipf {
        uint32_t V;
        double *cdf;     /* cdf[i] = (sum_{j<=i} 1/(j+1)) / H, sequential double sums */
        uint32_t *guide; /* guide[k] = first i with cdf[i] >= k/G  (search accelerator only) */
        uint32_t G;
};

static uint32_t lower_bound_d(const double *a, uint32_t lo, uint32_t hi, double x) {
        while (lo < hi) {
                uint32_t mid = lo + (hi - lo) / 2;
                if (a[mid] < x)
                        lo = mid + 1;
                else
                        hi = mid;
        }
        return lo;
}

to_zipf *to_zipf_new(uint32_t V) {
        to_zipf *z = (to_zipf *)xmalloc(sizeof *z);
        z->V = V;
        z->cdf = (double *)xmalloc(sizeof(double) * V);
        double s = 0;
        for (uint32_t i = 0; i < V; ++i) {
                s += 1.0 / (double)(i + 1);
                z->cdf[i] = s;
        }
        for (uint32_t i = 0; i < V; ++i)
                z->cdf[i] /= s;
        z->G = 1u << 16;
        z->guide = (uint32_t *)xmalloc(sizeof(uint32_t) * (z->G + 1));
        for (uint32_t k = 0; k <= z->G; ++k) {
                uint32_t r = lower_bound_d(z->cdf, 0, V, (double)k / (double)z->G);
                z->guide[k] = r < V ? r : V - 1;
        }
        return z;
}

void to_zipf_free(to_zipf *z) {
        if (!z)
                return;
        free(z->cdf);
        free(z->guide);
        free(z);
}

uint32_t to_zipf_rank(const to_zipf *z, uint64_t u) {
        const double x = (double)(u >> 11) * (1.0 / 9007199254740992.0);
        const uint32_t k = (uint32_t)(x * (double)z->G);
        uint32_t lo = z->guide[k], hi = z->guide[k + 1] + 1;
        if (hi > z->V)
                hi = z->V;
        /* guide[] only narrows the window; result == lower_bound over the whole table */
        while (lo > 0 && z->cdf[lo - 1] >= x)
                --lo;
        uint32_t r = lower_bound_d(z->cdf, lo, hi, x);
        if (r >= z->V)
                r = z->V - 1;
        return r;
}

to_corpus *to_corpus_generate(uint32_t D, uint32_t V, uint32_t slots, uint64_t seed) {
        to_corpus *c = (to_corpus *)xcalloc(1, sizeof *c);
        c->D = D;
        c->V = V;
        c->slots = slots;
        c->ntokens = (uint64_t)D * slots;
        to_zipf *z = to_zipf_new(V);
        uint32_t *ranks = (uint32_t *)xmalloc(sizeof(uint32_t) * c->ntokens);
        c->term_off = (uint64_t *)xcalloc((size_t)V + 1, sizeof(uint64_t));
        uint64_t st = seed;
        for (uint64_t i = 0; i < c->ntokens; ++i) {
                const uint32_t r = to_zipf_rank(z, to_splitmix64(&st));
                ranks[i] = r;
                c->term_off[r + 1]++;
        }
        for (uint32_t t = 0; t < V; ++t)
                c->term_off[t + 1] += c->term_off[t];
        c->tok_doc = (uint32_t *)xmalloc(sizeof(uint32_t) * c->ntokens);
        c->tok_pos = (uint16_t *)xmalloc(sizeof(uint16_t) * c->ntokens);
        uint64_t *cur = (uint64_t *)xmalloc(sizeof(uint64_t) * V);
        memcpy(cur, c->term_off, sizeof(uint64_t) * V);
        uint64_t i = 0;
        for (uint32_t d = 1; d <= D; ++d)
                for (uint32_t p = 1; p <= slots; ++p, ++i) {
                        const uint64_t o = cur[ranks[i]]++;
                        c->tok_doc[o] = d;
                        c->tok_pos[o] = (uint16_t)p;
                }
        free(cur);
        free(ranks);
        to_zipf_free(z);
        return c;
}

void to_corpus_free(to_corpus *c) {
        if (!c)
                return;
        free(c->term_off);
        free(c->tok_doc);
        free(c->tok_pos);
        free(c);
}

void to_gen_queries(uint32_t V, uint64_t seed, uint32_t nq, uint32_t nterms, uint32_t *out) {
        to_zipf *z = to_zipf_new(V);
        uint64_t st = seed;
        for (uint32_t q = 0; q < nq; ++q) {
                uint32_t *t = out + (size_t)q * nterms;
                for (uint32_t i = 0; i < nterms;) {
                        const uint32_t r = to_zipf_rank(z, to_splitmix64(&st));
                        int dup = 0;
                        for (uint32_t j = 0; j < i; ++j)
                                dup |= (t[j] == r);
                        if (!dup)
                                t[i++] = r;
                }
        }
        to_zipf_free(z);
}

/* ================================================================== Google codec: writer */
void buf_room(buf_t *b, size_t extra) {
        if (b->n + extra > b->cap) {
                size_t nc = b->cap ? b->cap * 2 : 4096;
                while (nc < b->n + extra)
                        nc *= 2;
                b->d = (uint8_t *)xrealloc(b->d, nc);
                b->cap = nc;
        }
}
void buf_varbyte(buf_t *b, uint32_t v) {
        buf_room(b, 5);
        b->n += to_varbyte_put32(b->d + b->n, v);
}
void buf_u8(buf_t *b, uint8_t v) {
        buf_room(b, 1);
        b->d[b->n++] = v;
}
void buf_u32(buf_t *b, uint32_t v) {
        buf_room(b, 4);
        memcpy(b->d + b->n, &v, 4);
        b->n += 4;
}
void buf_bytes(buf_t *b, const void *p, size_t n) {
        buf_room(b, n);
        memcpy(b->d + b->n, p, n);
        b->n += n;
}

typedef struct { /* google_codec.h:46-60 Encoder state */
        buf_t *out;
        buf_t skipListData, block, hitsData;
        uint32_t prevBlockLastDocumentID, curDocID, lastCommitedDocID;
        uint8_t curBlockSize, curPayloadSize;
        uint32_t lastPos;
        uint32_t docDeltas[GOOGLE_N];
        uint32_t blockFreqs[GOOGLE_N];
        uint32_t skiplistEntryCountdown; /* NOT reset per term: google_codec.h:57 vs google_codec.cpp:9-23 */
        uint32_t curTermOffset;
        uint32_t termDocuments;
} genc_t;

/* google_codec.cpp:9-23 */
static void genc_begin_term(genc_t *e) {
        e->curBlockSize = 0;
        e->lastCommitedDocID = 0;
        e->prevBlockLastDocumentID = 0;
        e->hitsData.n = 0;
        e->termDocuments = 0;
        e->curTermOffset = (uint32_t)e->out->n;
        buf_room(e->out, 2); /* u16 skiplist entry count, patched in end_term */
        e->out->d[e->out->n] = 0;
        e->out->d[e->out->n + 1] = 0;
        e->out->n += 2;
}

/* google_codec.cpp:25-36 */
static void genc_begin_document(genc_t *e, uint32_t documentID) {
        if (!documentID || documentID <= e->lastCommitedDocID) {
                fprintf(stderr, "trinity_oracle: unexpected documentID %u <= %u\n", documentID, e->lastCommitedDocID);
                abort();
        }
        e->curDocID = documentID;
        e->lastPos = 0;
        e->curPayloadSize = 0;
        e->blockFreqs[e->curBlockSize] = 0;
}

/* google_codec.cpp:38-74, payload-less hits (payloadSize == 0) */
static void genc_new_hit(genc_t *e, uint32_t pos) {
        const uint8_t payloadSize = 0;
        if (!pos && !payloadSize)
                return;
        const uint32_t delta = pos - e->lastPos;
        ++e->blockFreqs[e->curBlockSize];
        if (payloadSize != e->curPayloadSize) {
                buf_varbyte(&e->hitsData, (delta << 1) | 1);
                buf_u8(&e->hitsData, payloadSize);
                e->curPayloadSize = payloadSize;
        } else
                buf_varbyte(&e->hitsData, delta << 1);
        e->lastPos = pos;
}

/* google_codec.cpp:118-176 */
static void genc_commit_block(genc_t *e) {
        const uint32_t delta = e->curDocID - e->prevBlockLastDocumentID;
        const uint32_t n = e->curBlockSize - 1u;
        e->block.n = 0;
        for (uint32_t i = 0; i != n; ++i)
                buf_varbyte(&e->block, e->docDeltas[i]);
        for (uint32_t i = 0; i != e->curBlockSize; ++i)
                buf_varbyte(&e->block, e->blockFreqs[i]);
        const uint32_t blockLength = (uint32_t)(e->block.n + e->hitsData.n);
        if (--e->skiplistEntryCountdown == 0) {
                if (e->skipListData.n / 8 < UINT16_MAX) {
                        buf_u32(&e->skipListData, e->prevBlockLastDocumentID);
                        buf_u32(&e->skipListData, (uint32_t)(e->out->n - e->curTermOffset));
                }
                e->skiplistEntryCountdown = GOOGLE_SKIPLIST_STEP;
        }
        buf_varbyte(e->out, delta);
        buf_varbyte(e->out, blockLength);
        buf_u8(e->out, e->curBlockSize);
        buf_bytes(e->out, e->block.d, e->block.n);
        buf_bytes(e->out, e->hitsData.d, e->hitsData.n);
        e->hitsData.n = 0;
        e->prevBlockLastDocumentID = e->curDocID;
        e->curBlockSize = 0;
}

/* google_codec.cpp:76-88 */
static void genc_end_document(genc_t *e) {
        e->docDeltas[e->curBlockSize++] = e->curDocID - e->lastCommitedDocID;
        if (e->curBlockSize == GOOGLE_N)
                genc_commit_block(e);
        e->lastCommitedDocID = e->curDocID;
        ++e->termDocuments;
}

/* google_codec.cpp:90-116 */
static void genc_end_term(genc_t *e, to_term *tctx) {
        if (e->curBlockSize)
                genc_commit_block(e);
        const uint16_t skipListEntries = (uint16_t)(e->skipListData.n / 8);
        buf_bytes(e->out, e->skipListData.d, e->skipListData.n);
        memcpy(e->out->d + e->curTermOffset, &skipListEntries, 2);
        tctx->offset = e->curTermOffset;
        tctx->size = (uint32_t)(e->out->n - e->curTermOffset);
        tctx->documents = e->termDocuments;
        e->skipListData.n = 0;
}

to_index *to_google_encode(const to_corpus *c) {
        to_index *ix = (to_index *)xcalloc(1, sizeof *ix);
        buf_t out = {0, 0, 0};
        genc_t e;
        memset(&e, 0, sizeof e);
        e.out = &out;
        e.skiplistEntryCountdown = GOOGLE_SKIPLIST_STEP;
        ix->terms = (to_term *)xcalloc(c->V, sizeof(to_term));
        ix->nterms = c->V;
        for (uint32_t t = 0; t < c->V; ++t) {
                const uint64_t b = c->term_off[t], end = c->term_off[t + 1];
                if (b == end)
                        continue; /* term never occurs: no chunk, documents = 0 */
                genc_begin_term(&e);
                for (uint64_t i = b; i < end;) {
                        const uint32_t d = c->tok_doc[i];
                        genc_begin_document(&e, d);
                        for (; i < end && c->tok_doc[i] == d; ++i)
                                genc_new_hit(&e, c->tok_pos[i]);
                        genc_end_document(&e);
                }
                genc_end_term(&e, &ix->terms[t]);
                ix->totalTerms++;
                ix->sumTermsDocs += ix->terms[t].documents;
        }
        if (out.n > 0xffffffffull) {
                fprintf(stderr, "trinity_oracle: index exceeds 32-bit offsets (codecs.h:26)\n");
                abort();
        }
        buf_room(&out, 16); /* slack so that wide reads near the end stay in bounds */
        memset(out.d + out.n, 0, 16);
        ix->bytes = out.d;
        ix->len = out.n;
        ix->sumTermHits = c->ntokens;
        ix->docsCnt = c->D;
        ix->owns = 1;
        free(e.skipListData.d);
        free(e.block.d);
        free(e.hitsData.d);
        return ix;
}

to_index *to_index_wrap(const uint8_t *bytes, size_t len, const to_term *terms, uint32_t nterms, uint32_t docsCnt,
                        uint64_t sumTermsDocs, uint64_t sumTermHits) {
        to_index *ix = (to_index *)xcalloc(1, sizeof *ix);
        ix->bytes = (uint8_t *)xmalloc(len + 16);
        memcpy(ix->bytes, bytes, len);
        memset(ix->bytes + len, 0, 16);
        ix->len = len;
        ix->terms = (to_term *)xmalloc(sizeof(to_term) * nterms);
        memcpy(ix->terms, terms, sizeof(to_term) * nterms);
        ix->nterms = nterms;
        ix->docsCnt = docsCnt;
        ix->sumTermsDocs = sumTermsDocs;
        ix->sumTermHits = sumTermHits;
        for (uint32_t i = 0; i < nterms; ++i)
                ix->totalTerms += terms[i].documents != 0;
        ix->owns = 1;
        return ix;
}

static int cmp_u32(const void *a, const void *b) {
        const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
        return x < y ? -1 : x > y;
}

void to_index_set_masked(to_index *ix, const uint32_t *docids, size_t n) {
        free(ix->masked);
        ix->masked = NULL;
        ix->nmasked = 0;
        if (!n)
                return;
        ix->masked = (uint32_t *)xmalloc(sizeof(uint32_t) * n);
        memcpy(ix->masked, docids, sizeof(uint32_t) * n);
        qsort(ix->masked, n, sizeof(uint32_t), cmp_u32);
        ix->nmasked = n;
}

void to_index_free(to_index *ix) {
        if (!ix)
                return;
        free(ix->masked);
        if (ix->owns) {
                free(ix->bytes);
                free(ix->terms);
                free(ix->hits);
        }
        free(ix);
}

/* Appendix A.2 of SURVEY.md / google_codec.cpp:118-176, 936-983 */
uint32_t to_google_chunk_stats(const to_index *ix, uint32_t term, uint64_t *hdr, uint64_t *docfreq, uint64_t *hits,
                               uint64_t *skip, uint64_t *postings) {
        const to_term *t = &ix->terms[term];
        *hdr = *docfreq = *hits = *skip = *postings = 0;
        if (!t->size)
                return 0;
        const uint8_t *base = ix->bytes + t->offset, *p = base, *end = base + t->size;
        uint16_t sk;
        memcpy(&sk, p, 2);
        p += 2;
        *hdr += 2;
        end -= (size_t)sk * 8;
        *skip = (uint64_t)sk * 8;
        uint32_t blocks = 0;
        while (p != end) {
                uint32_t v, blockLength;
                const uint8_t *h = p;
                p += to_varbyte_get32(p, &v);
                p += to_varbyte_get32(p, &blockLength);
                const uint8_t n = *p++;
                *hdr += (uint64_t)(p - h);
                const uint8_t *q = p;
                for (uint32_t i = 0; i + 1 < n; ++i)
                        q += to_varbyte_get32(q, &v);
                for (uint32_t i = 0; i < n; ++i)
                        q += to_varbyte_get32(q, &v);
                *docfreq += (uint64_t)(q - p);
                *hits += blockLength - (uint64_t)(q - p);
                *postings += n;
                p += blockLength;
                ++blocks;
        }
        return blocks;
}

/* ================================================================== Google codec: reader */
struct to_pli { /* google_codec.h:104-133 PostingsListIterator + :143-187 Decoder */
        TO_PLI_HEAD
        /* decoder */
        const uint8_t *base, *chunkEnd;
        const uint8_t *skiplist; /* entries {u32 prevBlockLastDocID, u32 offset} */
        uint32_t skiplistSize;
        /* iterator */
        uint8_t blockDocIdx;
        uint32_t documentsArr[GOOGLE_N];
        uint32_t blockLastDocID;
        uint32_t freqs[GOOGLE_N];
        uint32_t skipListIdx;
        const uint8_t *p;
};

static uint32_t sk_first(const to_pli *d, uint32_t i) {
        uint32_t v;
        memcpy(&v, d->skiplist + (size_t)i * 8, 4);
        return v;
}
static uint32_t sk_second(const to_pli *d, uint32_t i) {
        uint32_t v;
        memcpy(&v, d->skiplist + (size_t)i * 8 + 4, 4);
        return v;
}

/* google_codec.h:165-172 */
static void pli_finalize(to_pli *it) {
        it->blockDocIdx = 0;
        it->blockLastDocID = TO_DOCIDS_END;
        it->documentsArr[0] = TO_DOCIDS_END;
        it->p = it->chunkEnd;
        it->it.cur = TO_DOCIDS_END;
}

/* google_codec.cpp:464-495 */
static uint32_t pli_skiplist_search(const to_pli *it, uint32_t target) {
        uint32_t idx = UINT32_MAX;
        const uint32_t skipListIdx = it->skipListIdx;
        for (int32_t top = (int32_t)it->skiplistSize - 1, btm = (int32_t)skipListIdx; btm <= top;) {
                const int32_t mid = (btm + top) / 2;
                const uint32_t v = sk_first(it, (uint32_t)mid);
                if (target < v)
                        top = mid - 1;
                else {
                        if (v != target)
                                idx = (uint32_t)mid;
                        else if ((uint32_t)mid != skipListIdx)
                                idx = (uint32_t)mid - 1;
                        btm = mid + 1;
                }
        }
        return idx;
}

/* google_codec.cpp:497-531 */
static void pli_skip_block_doc(to_pli *it) {
        const uint32_t freq = it->freqs[it->blockDocIdx];
        uint8_t curPayloadSize = 0;
        uint32_t dummy;
        const uint8_t *p = it->p;
        for (uint32_t i = 0; i != freq; ++i) {
                p += to_varbyte_get32(p, &dummy);
                if (dummy & 1)
                        curPayloadSize = *p++;
                p += curPayloadSize;
        }
        it->p = p;
}

/* google_codec.cpp:596-639 */
static void pli_unpack_block(to_pli *it, uint32_t thisBlockLastDocID, uint8_t n) {
        const uint32_t k = n - 1u;
        uint32_t id = it->blockLastDocID;
        const uint8_t *p = it->p;
        for (uint32_t i = 0; i != k; ++i) {
                uint32_t delta;
                p += to_varbyte_get32(p, &delta);
                id += delta;
                it->documentsArr[i] = id;
        }
        for (uint32_t i = 0; i != n; ++i) {
                uint32_t v;
                p += to_varbyte_get32(p, &v);
                it->freqs[i] = v;
        }
        it->p = p;
        it->blockLastDocID = thisBlockLastDocID;
        it->documentsArr[k] = thisBlockLastDocID;
        it->blockDocIdx = 0;
}

/* google_codec.cpp:641-697 */
static void pli_seek_block(to_pli *it, uint32_t target) {
        const uint8_t *p = it->p;
        uint32_t blockLastDocID = it->blockLastDocID;
        for (;;) {
                uint32_t v, blockSize;
                p += to_varbyte_get32(p, &v);
                const uint32_t thisBlockLastDocID = blockLastDocID + v;
                p += to_varbyte_get32(p, &blockSize);
                const uint8_t blockDocsCnt = *p++;
                if (target > thisBlockLastDocID) {
                        p += blockSize;
                        if (p == it->chunkEnd) {
                                pli_finalize(it);
                                return;
                        }
                        blockLastDocID = thisBlockLastDocID;
                } else {
                        it->p = p;
                        it->blockLastDocID = blockLastDocID;
                        pli_unpack_block(it, thisBlockLastDocID, blockDocsCnt);
                        return;
                }
        }
}

/* google_codec.cpp:699-724 */
static void pli_unpack_next_block(to_pli *it) {
        uint32_t v, blockSize;
        const uint8_t *p = it->p;
        p += to_varbyte_get32(p, &v);
        const uint32_t thisBlockLastDocID = it->blockLastDocID + v;
        p += to_varbyte_get32(p, &blockSize);
        const uint8_t blockDocsCnt = *p++;
        it->p = p;
        pli_unpack_block(it, thisBlockLastDocID, blockDocsCnt);
}

/* google_codec.cpp:726-775 */
static void pli_skip_remaining_block_documents(to_pli *it) {
        uint8_t blockDocIdx = it->blockDocIdx;
        const uint32_t blockLastDocID = it->blockLastDocID;
        const uint8_t *p = it->p;
        for (;;) {
                uint32_t freq = it->freqs[blockDocIdx];
                uint32_t dummy;
                uint8_t payloadSize = 0;
                while (freq) {
                        --freq;
                        p += to_varbyte_get32(p, &dummy);
                        if (dummy & 1)
                                payloadSize = *p++;
                        p += payloadSize;
                }
                if (it->documentsArr[blockDocIdx] == blockLastDocID)
                        break;
                else
                        ++blockDocIdx;
        }
        it->p = p;
        it->blockDocIdx = blockDocIdx;
}

/* google_codec.cpp:777-819 */
static uint32_t pli_next(to_iter *self) {
        to_pli *it = (to_pli *)self;
        if (it->documentsArr[it->blockDocIdx] == it->blockLastDocID) {
                pli_skip_block_doc(it);
                if (it->p != it->chunkEnd)
                        pli_unpack_next_block(it);
                else {
                        pli_finalize(it);
                        return it->it.cur;
                }
        } else {
                pli_skip_block_doc(it);
                ++it->blockDocIdx;
        }
        it->it.cur = it->documentsArr[it->blockDocIdx];
        it->freq = (uint16_t)it->freqs[it->blockDocIdx];
        return it->it.cur;
}

/* google_codec.cpp:821-934 */
static uint32_t pli_advance(to_iter *self, uint32_t target) {
        to_pli *it = (to_pli *)self;
        if (target > it->blockLastDocID) {
                pli_skip_remaining_block_documents(it);
                if (it->p == it->chunkEnd) {
                        pli_finalize(it);
                        return it->it.cur;
                }
                if (it->skipListIdx != it->skiplistSize) {
                        const uint32_t idx = pli_skiplist_search(it, target);
                        if (idx != UINT32_MAX) {
                                const uint32_t savedBlockLastDocID = it->blockLastDocID;
                                it->blockLastDocID = sk_first(it, idx);
                                it->p = it->base + sk_second(it, idx);
                                if (target > savedBlockLastDocID)
                                        it->skipListIdx = idx + 1;
                        }
                }
                pli_seek_block(it, target);
                /* NB: after finalize() documents[0] == END > target, the scan below settles on END */
        }
        uint8_t blockDocIdx = it->blockDocIdx;
        for (;;) {
                const uint32_t docID = it->documentsArr[blockDocIdx];
                if (docID > target)
                        break;
                else if (docID == target)
                        break;
                else if (docID == it->blockLastDocID)
                        break;
                else {
                        it->blockDocIdx = blockDocIdx;
                        pli_skip_block_doc(it);
                        ++blockDocIdx;
                }
        }
        it->it.cur = it->documentsArr[blockDocIdx];
        it->freq = (uint16_t)it->freqs[blockDocIdx];
        it->blockDocIdx = blockDocIdx;
        return it->it.cur;
}

/* google_codec.cpp:533-594 (positions only; payload bytes are skipped exactly as the reference reads them) */
uint32_t to_pli_materialize_positions(to_pli *it, uint16_t *out) { return it->materialize(it, out); }

static uint32_t google_materialize_positions(to_pli *it, uint16_t *out);
/* google_codec.cpp:533-594 in full: out[i] = {payload, pos, curPayloadSize}.  The payload word is a local that lives across the
 * document's hits: a new payload overwrites its first curPayloadSize bytes only (memcpy), size 0 clears it. */
uint32_t to_pli_materialize_hits(to_pli *it, uint16_t *pos_out, uint8_t *plen_out, uint64_t *payload_out) {
        if (it->materialize != google_materialize_positions) { /* the Lucene-shaped segments of this repo carry no payloads */
                const uint32_t n = it->materialize(it, pos_out);
                for (uint32_t i = 0; i < n; ++i) {
                        plen_out[i] = 0;
                        payload_out[i] = 0;
                }
                return n;
        }
        const uint32_t freq = it->freqs[it->blockDocIdx];
        uint16_t pos = 0;
        uint8_t curPayloadSize = 0;
        uint64_t payload = 0;
        uint32_t step;
        const uint8_t *p = it->p;
        for (uint32_t i = 0; i != (uint16_t)freq; ++i) {
                p += to_varbyte_get32(p, &step);
                if (step & 1)
                        curPayloadSize = *p++;
                pos = (uint16_t)(pos + (step >> 1));
                if (curPayloadSize) {
                        memcpy(&payload, p, curPayloadSize <= 8 ? curPayloadSize : 8);
                        p += curPayloadSize;
                } else
                        payload = 0;
                pos_out[i] = pos;
                plen_out[i] = curPayloadSize;
                payload_out[i] = payload;
        }
        it->p = p;
        it->freqs[it->blockDocIdx] = 0; /* google_codec.cpp:593 */
        return (uint16_t)freq;
}

static uint32_t google_materialize_positions(to_pli *it, uint16_t *out) {
        const uint32_t freq = it->freqs[it->blockDocIdx];
        uint16_t pos = 0;
        uint8_t curPayloadSize = 0;
        uint32_t step;
        const uint8_t *p = it->p;
        for (uint32_t i = 0; i != (uint16_t)freq; ++i) { /* loop index is tokenpos_t in the reference */
                p += to_varbyte_get32(p, &step);
                if (step & 1)
                        curPayloadSize = *p++;
                pos = (uint16_t)(pos + (step >> 1));
                p += curPayloadSize;
                out[i] = pos;
        }
        it->p = p;
        it->freqs[it->blockDocIdx] = 0; /* google_codec.cpp:593 */
        return (uint16_t)freq;
}


/* google_codec.cpp:936-990 (Decoder::init) + 442-462 (new_iterator) */
to_pli *to_pli_new(const to_index *ix, uint32_t term) {
        if (ix->codec == TO_CODEC_LUCENE)
                return to_lucene_pli_new(ix, term);
        to_pli *it = (to_pli *)xcalloc(1, sizeof *it);
        it->materialize = google_materialize_positions;
        const to_term *t = &ix->terms[term];
        const uint8_t *ptr = ix->bytes + t->offset;
        it->it.type = IT_PLI;
        it->it.next = pli_next;
        it->it.advance = pli_advance;
        it->it.score = to_pli_score_bm25;
        it->it.cost = t->documents; /* docset_iterators.cpp:57-58 */
        it->term = term;
        it->documents = t->documents;
        it->base = ptr;
        it->chunkEnd = ptr + t->size;
        if (t->size) {
                uint16_t cnt;
                memcpy(&cnt, ptr, 2);
                if (cnt) {
                        it->skiplist = (ptr + t->size) - (size_t)cnt * 8;
                        it->skiplistSize = cnt;
                        it->chunkEnd = it->skiplist;
                }
                it->blockDocIdx = 0;
                it->documentsArr[0] = 0;
                it->blockLastDocID = 0;
                it->freqs[0] = 0;
                it->skipListIdx = 0;
                it->p = ptr + 2;
        } else
                pli_finalize(it);
        it->sim = ix->similarity;
        it->idf = to_sim_weight(ix->similarity, t->documents, ix->docsCnt);
        return it;
}

void to_pli_free(to_pli *it) {
        if (it && it->destroy)
                it->destroy(it);
        free(it);
}
uint32_t to_pli_next(to_pli *it) { return it->it.next(&it->it); }
uint32_t to_pli_advance(to_pli *it, uint32_t t) { return it->it.advance(&it->it, t); }
uint32_t to_pli_current(const to_pli *it) { return it->it.cur; }
uint32_t to_pli_freq(const to_pli *it) { return it->freq; }

uint32_t to_decode_term(const to_index *ix, uint32_t term, uint32_t *docs, uint32_t *freqs) {
        to_pli *it = to_pli_new(ix, term);
        uint32_t n = 0;
        for (uint32_t id = it->it.next(&it->it); id != TO_DOCIDS_END; id = it->it.next(&it->it)) {
                docs[n] = id;
                if (freqs)
                        freqs[n] = it->freq;
                ++n;
        }
        to_pli_free(it);
        return n;
}

/* ================================================================== similarity (BM25) */
/* similarity.h:179-181: std::log(1 + (docsCnt - docFreq + 0.5f) / (docFreq + 0.5f)) — the whole
 * expression is float (u64 -> float, u32 -> float, int 1 -> float, std::log(float) == logf). */
double to_bm25_idf(uint32_t docFreq, uint64_t docsCnt) {
        const float num = (float)(docsCnt - (uint64_t)docFreq) + 0.5f;
        const float den = (float)docFreq + 0.5f;
        return (double)logf(1 + num / den);
}

/* similarity.h:228-235: return idf * float(freq) / double(freq + k1), k1 = 1.2f, rounded to float */
float to_bm25_score(double idf, uint16_t freq) {
        const float norm = 1.2f;
        return (float)(idf * (float)freq / (double)((float)freq + norm));
}

double to_sim_weight(int sim, uint32_t docFreq, uint64_t docsCnt) {
        if (sim == TO_SIM_TFIDF)
                return log((double)(docsCnt + 1) / (double)(docFreq + 1)) + 1.0; /* u64 / double -> double division */
        if (sim == TO_SIM_TRIVIAL)
                return 0.0;
        return to_bm25_idf(docFreq, docsCnt);
}

float to_sim_score(int sim, double weight, uint16_t freq) {
        if (sim == TO_SIM_TFIDF) {
                const float tf = sqrtf((float)freq);
                return (float)(tf * weight);
        }
        if (sim == TO_SIM_TRIVIAL)
                return (float)freq;
        return to_bm25_score(weight, freq);
}

/* docset_iterators_scorers.cpp:28-32 */
double to_pli_score_bm25(to_iter *self) {
        to_pli *it = (to_pli *)self;
        return to_sim_score(it->sim, it->idf, it->freq);
}

/* ================================================================== Conjuction */
typedef struct { /* docset_iterators.h:333-362 */
        to_iter it;
        to_iter **its;
        uint16_t size;
} to_conj;

/* docset_iterators.cpp:308-348 */
static uint32_t conj_next_impl(to_conj *c, uint32_t id) {
        const uint16_t localSize = c->size;
restart:
        for (uint16_t i = 1; i != localSize; ++i) {
                to_iter *it = c->its[i];
                if (it->cur != id) {
                        const uint32_t next = it->advance(it, id);
                        if (next > id) {
                                if (next == TO_DOCIDS_END) {
                                        c->size = 0;
                                        return c->it.cur = TO_DOCIDS_END;
                                }
                                id = c->its[0]->advance(c->its[0], next);
                                if (id == TO_DOCIDS_END) {
                                        c->size = 0;
                                        return c->it.cur = TO_DOCIDS_END;
                                }
                                goto restart;
                        }
                }
        }
        return c->it.cur = id;
}

/* docset_iterators.cpp:295-306 */
static uint32_t conj_next(to_iter *self) {
        to_conj *c = (to_conj *)self;
        if (c->size) {
                const uint32_t id = c->its[0]->next(c->its[0]);
                if (id == TO_DOCIDS_END) {
                        c->size = 0;
                        return c->it.cur = TO_DOCIDS_END;
                }
                return conj_next_impl(c, id);
        }
        return TO_DOCIDS_END;
}

/* docset_iterators.cpp:282-293 */
static uint32_t conj_advance(to_iter *self, uint32_t target) {
        to_conj *c = (to_conj *)self;
        if (c->size) {
                const uint32_t id = c->its[0]->advance(c->its[0], target);
                if (id == TO_DOCIDS_END) {
                        c->size = 0;
                        return c->it.cur = TO_DOCIDS_END;
                }
                return conj_next_impl(c, id);
        }
        return TO_DOCIDS_END;
}

/* docset_iterators_scorers.cpp:173-193; `size` is read at scoring time in the reference, but a
 * drained conjunction is never scored, so the construction-time count is equivalent */
typedef struct {
        to_conj c;
        uint16_t nscore;
} to_conj_s;
static double conj_score(to_iter *self) {
        to_conj_s *c = (to_conj_s *)self;
        double res = 0;
        for (uint16_t i = 0; i != c->nscore; ++i)
                res += c->c.its[i]->score(c->c.its[i]);
        return res;
}

/* ================================================================== Disjunction */
typedef struct { /* docset_iterators.h:221-303; Switch/prioqueue.h min-heap on current() */
        to_iter it;
        to_iter **heap; /* 1-based */
        uint32_t n;
} to_disj;

static void heap_down(to_iter **h, uint32_t n, uint32_t i) {
        to_iter *x = h[i];
        for (;;) {
                uint32_t c = i * 2;
                if (c > n)
                        break;
                if (c + 1 <= n && h[c + 1]->cur < h[c]->cur)
                        ++c;
                if (h[c]->cur < x->cur) {
                        h[i] = h[c];
                        i = c;
                } else
                        break;
        }
        h[i] = x;
}
static void heap_up(to_iter **h, uint32_t i) {
        to_iter *x = h[i];
        while (i > 1 && x->cur < h[i / 2]->cur) {
                h[i] = h[i / 2];
                i /= 2;
        }
        h[i] = x;
}
static void heap_push(to_iter **h, uint32_t *n, to_iter *x) {
        h[++*n] = x;
        heap_up(h, *n);
}
static to_iter *heap_pop(to_iter **h, uint32_t *n) {
        to_iter *top = h[1];
        h[1] = h[*n];
        --*n;
        if (*n)
                heap_down(h, *n, 1);
        return top;
}

/* docset_iterators.cpp:350-372 (DisjunctionAllPLI::next) == 592-614 (Disjunction::next) */
static uint32_t disj_next(to_iter *self) {
        to_disj *d = (to_disj *)self;
        if (!d->n)
                return TO_DOCIDS_END;
        to_iter *top = d->heap[1];
        const uint32_t doc = top->cur;
        do {
                if (top->next(top) != TO_DOCIDS_END) {
                        heap_down(d->heap, d->n, 1); /* pq.update_top() */
                        top = d->heap[1];
                } else {
                        heap_pop(d->heap, &d->n); /* pq.erase(top) */
                        if (!d->n)
                                return d->it.cur = TO_DOCIDS_END;
                        top = d->heap[1];
                }
        } while ((d->it.cur = top->cur) == doc);
        return d->it.cur;
}

/* docset_iterators.cpp:374-405 */
static uint32_t disj_advance(to_iter *self, uint32_t target) {
        to_disj *d = (to_disj *)self;
        if (!d->n)
                return TO_DOCIDS_END;
        to_iter *top = d->heap[1];
        do {
                const uint32_t res = top->advance(top, target);
                if (res != TO_DOCIDS_END) {
                        heap_down(d->heap, d->n, 1);
                        top = d->heap[1];
                } else {
                        heap_pop(d->heap, &d->n);
                        if (!d->n)
                                return d->it.cur = TO_DOCIDS_END;
                        top = d->heap[1];
                }
        } while ((d->it.cur = top->cur) < target);
        return d->it.cur;
}

/* docset_iterators_scorers.cpp:107-148: sum over heap entries positioned on the current doc
 * (for_each_top tree walk, Switch/prioqueue.h) */
static void disj_score_walk(to_disj *d, uint32_t i, uint32_t doc, double *sum) {
        if (i > d->n || d->heap[i]->cur != doc)
                return;
        *sum += d->heap[i]->score(d->heap[i]);
        disj_score_walk(d, i * 2, doc, sum);
        disj_score_walk(d, i * 2 + 1, doc, sum);
}
static double disj_score(to_iter *self) {
        to_disj *d = (to_disj *)self;
        double sum = 0;
        if (d->n)
                disj_score_walk(d, 1, d->heap[1]->cur, &sum);
        return sum;
}

/* ================================================================== Phrase */
typedef struct { /* docwordspace.h:16-92: one term per position, last writer wins; docSeq marks validity */
        uint16_t termID[65536 + MAX_PHRASE + 1];
        uint32_t docSeq[65536 + MAX_PHRASE + 1];
        uint32_t curSeq;
} to_dws;

typedef struct { /* docset_iterators.h:364-402 */
        to_iter it;
        to_pli **its;
        uint16_t size;
        uint16_t nterms;
        uint16_t maxMatchCnt, matchCnt;
        uint16_t execTermID[MAX_PHRASE]; /* distinct per distinct term (queryexec_ctx::resolve_term) */
        double weight;                   /* similarity.h:202-226: sum of the terms' idf */
        int sim;                         /* TO_SIM_* */
        to_dws *dws;
        uint16_t *hits0;
        uint16_t *scratch;
} to_phrase;

/* docset_iterators.cpp:66-158 + queryexec_ctx.cpp:317-351 (materialize_term_hits) */
static int phrase_consider(to_phrase *ph) {
        to_dws *dws = ph->dws;
        const uint16_t n = ph->nterms;
        /* fresh DocWordsSpace for this candidate document (docwordspace.h:40-55 reset()) */
        if (++dws->curSeq == 0) {
                memset(dws->docSeq, 0, sizeof dws->docSeq);
                dws->curSeq = 1;
        }
        uint32_t firstTermFreq = 0;
        ph->matchCnt = 0;
        for (uint16_t i = 0; i != n; ++i) {
                /* term hits are materialized once per (document, exec term id): a term repeated inside
                 * the phrase is NOT re-materialized (queryexec_ctx.cpp:330 th->doc_id != did) */
                int seen = 0;
                for (uint16_t j = 0; j < i; ++j)
                        seen |= (ph->execTermID[j] == ph->execTermID[i]);
                if (seen)
                        continue;
                to_pli *it = ph->its[i];
                if (i == 0)
                        firstTermFreq = it->freq; /* th->set_freq(it->freq): u16 */
                uint16_t *dst = i == 0 ? ph->hits0 : ph->scratch;
                const uint32_t cnt = to_pli_materialize_positions(it, dst);
                for (uint32_t k = 0; k < cnt; ++k) {
                        const uint16_t pos = dst[k];
                        if (pos) { /* google_codec.cpp:578-584: dws->set(termID, pos) */
                                dws->termID[pos] = ph->execTermID[i];
                                dws->docSeq[pos] = dws->curSeq;
                        }
                }
        }
        for (uint32_t i = 0; i != firstTermFreq; ++i) {
                const uint16_t pos = ph->hits0[i];
                if (!pos)
                        continue;
                for (uint16_t k = 1;; ++k) {
                        if (k == n) {
                                if (++ph->matchCnt == ph->maxMatchCnt)
                                        return 1;
                                break;
                        }
                        const uint32_t q = (uint32_t)pos + k;
                        if (!(dws->docSeq[q] == dws->curSeq && dws->termID[q] == ph->execTermID[k]))
                                break;
                }
        }
        return ph->matchCnt != 0;
}

/* docset_iterators.cpp:160-183 */
static uint32_t phrase_next_impl(to_phrase *ph, uint32_t id) {
restart:
        for (uint16_t i = 1; i != ph->size; ++i) {
                to_iter *it = &ph->its[i]->it;
                if (it->cur != id) {
                        const uint32_t next = it->advance(it, id);
                        if (next > id) {
                                if (next == TO_DOCIDS_END)
                                        return TO_DOCIDS_END;
                                id = ph->its[0]->it.advance(&ph->its[0]->it, next);
                                if (id == TO_DOCIDS_END)
                                        return TO_DOCIDS_END;
                                goto restart;
                        }
                }
        }
        return ph->it.cur = id;
}

/* docset_iterators.cpp:205-224 */
static uint32_t phrase_next(to_iter *self) {
        to_phrase *ph = (to_phrase *)self;
        if (ph->size) {
                to_iter *lead = &ph->its[0]->it;
                uint32_t id = lead->next(lead);
                if (id == TO_DOCIDS_END) {
                        ph->size = 0;
                        return ph->it.cur = TO_DOCIDS_END;
                }
                for (id = phrase_next_impl(ph, id);; id = phrase_next_impl(ph, lead->next(lead))) {
                        if (id == TO_DOCIDS_END) {
                                ph->size = 0;
                                return ph->it.cur = TO_DOCIDS_END;
                        } else if (phrase_consider(ph))
                                return ph->it.cur = id;
                }
        }
        return TO_DOCIDS_END;
}

/* docset_iterators.cpp:185-203.  NB: when lead->next() returns END inside the for-increment the
 * reference hands END to next_impl(), whose members then advance(END) and report END. */
static uint32_t phrase_advance(to_iter *self, uint32_t target) {
        to_phrase *ph = (to_phrase *)self;
        if (ph->size) {
                to_iter *lead = &ph->its[0]->it;
                uint32_t id = lead->advance(lead, target);
                if (id == TO_DOCIDS_END) {
                        ph->size = 0;
                        return ph->it.cur = TO_DOCIDS_END;
                }
                for (id = phrase_next_impl(ph, id);; id = phrase_next_impl(ph, lead->next(lead))) {
                        if (id == TO_DOCIDS_END) {
                                ph->size = 0;
                                return ph->it.cur = TO_DOCIDS_END;
                        } else if (phrase_consider(ph))
                                return id;
                }
        }
        return TO_DOCIDS_END;
}

/* docset_iterators_scorers.cpp:195-228: scorer->score(id, matchCnt, weight) */
static double phrase_score(to_iter *self) {
        to_phrase *ph = (to_phrase *)self;
        return to_sim_score(ph->sim, ph->weight, ph->matchCnt);
}

/* ================================================================== plan -> iterator tree */
typedef struct {
        const to_index *ix;
        uint32_t flags;
        void **owned;
        size_t nowned, capowned;
        /* exec term ids: one per distinct term of the query (queryexec_ctx.cpp:279-296) */
        uint32_t termOf[64];
        uint16_t nTermIDs;
} to_ctx;

static void *ctx_own(to_ctx *c, void *p) {
        if (c->nowned == c->capowned) {
                c->capowned = c->capowned ? c->capowned * 2 : 32;
                c->owned = (void **)xrealloc(c->owned, sizeof(void *) * c->capowned);
        }
        c->owned[c->nowned++] = p;
        return p;
}

static uint16_t ctx_term_id(to_ctx *c, uint32_t term) {
        for (uint16_t i = 0; i < c->nTermIDs; ++i)
                if (c->termOf[i] == term)
                        return (uint16_t)(i + 1);
        if (c->nTermIDs == 64)
                abort();
        c->termOf[c->nTermIDs++] = term;
        return c->nTermIDs;
}

/* ================================================================== Filter (logicalnot) */
typedef struct { /* docset_iterators.h:147-172 */
        to_iter it;
        to_iter *req, *filter;
} to_filter;

/* docset_iterators.cpp:652-659 */
static int filter_matches(to_filter *f, uint32_t id) {
        uint32_t excl = f->filter->cur;
        if (excl < id)
                excl = f->filter->advance(f->filter, id);
        return excl != id;
}

/* docset_iterators.cpp:661-668 */
static uint32_t filter_next(to_iter *self) {
        to_filter *f = (to_filter *)self;
        for (uint32_t id = f->req->next(f->req);; id = f->req->next(f->req)) {
                if (id == TO_DOCIDS_END)
                        return f->it.cur = TO_DOCIDS_END;
                else if (filter_matches(f, id))
                        return f->it.cur = id;
        }
}

/* docset_iterators.cpp:670-677 */
static uint32_t filter_advance(to_iter *self, uint32_t target) {
        to_filter *f = (to_filter *)self;
        for (uint32_t id = f->req->advance(f->req, target);; id = f->req->next(f->req)) {
                if (id == TO_DOCIDS_END)
                        return f->it.cur = TO_DOCIDS_END;
                else if (filter_matches(f, id))
                        return f->it.cur = id;
        }
}

/* docset_iterators_scorers.cpp:59-73: the score of a Filter is the score of what it requires */
static double filter_score(to_iter *self) {
        to_filter *f = (to_filter *)self;
        return f->req->score(f->req);
}

/* ================================================================== Optional (consttrueexpr under an AND) */
typedef struct { /* docset_iterators.h:174-206 */
        to_iter it;
        to_iter *main, *opt;
} to_optional;

static uint32_t optional_next(to_iter *self) {
        to_optional *o = (to_optional *)self;
        return o->it.cur = o->main->next(o->main);
}

static uint32_t optional_advance(to_iter *self, uint32_t target) {
        to_optional *o = (to_optional *)self;
        return o->it.cur = o->main->advance(o->main, target);
}

/* does the optional side hold the document the main side sits on?  (docset_iterators_scorers.cpp:87-101, queryexec_ctx.cpp:418-432) */
static int optional_opt_matches(to_optional *o) {
        const uint32_t id = o->main->cur;
        uint32_t optId = o->opt->cur;
        if (optId < id)
                optId = o->opt->advance(o->opt, id);
        return optId == id;
}

/* docset_iterators_scorers.cpp:77-104 */
static double optional_score(to_iter *self) {
        to_optional *o = (to_optional *)self;
        double score = o->main->score(o->main);
        if (optional_opt_matches(o))
                score += o->opt->score(o->opt);
        return score;
}

/* ================================================================== DisjunctionSome (matchsome) */
typedef struct sm_tracker { /* docset_iterators.h:66-72 it_tracker */
        to_iter *it;
        uint64_t cost;
        uint32_t id;
        struct sm_tracker *next;
} sm_tracker;

typedef struct { /* docset_iterators.h:61-137; Switch/prioqueue.h binary heaps (1-based), top = least by the comparator */
        to_iter it;
        sm_tracker *lead, *store;
        uint16_t threshold, curCnt;
        sm_tracker **head; /* least id on top */
        uint32_t nhead;
        sm_tracker **tail; /* least cost on top, at most threshold - 1 entries */
        uint32_t ntail, captail;
} to_some;

static int sm_less_id(const sm_tracker *a, const sm_tracker *b) { return a->id < b->id; }
static int sm_less_cost(const sm_tracker *a, const sm_tracker *b) { return a->cost < b->cost; }

static void sm_up(sm_tracker **h, uint32_t i, int (*less)(const sm_tracker *, const sm_tracker *)) {
        sm_tracker *x = h[i];
        while (i > 1 && less(x, h[i / 2])) {
                h[i] = h[i / 2];
                i /= 2;
        }
        h[i] = x;
}

static void sm_down(sm_tracker **h, uint32_t n, uint32_t i, int (*less)(const sm_tracker *, const sm_tracker *)) {
        sm_tracker *x = h[i];
        for (;;) {
                uint32_t c = i * 2;
                if (c > n)
                        break;
                if (c + 1 <= n && less(h[c + 1], h[c]))
                        ++c;
                if (!less(h[c], x))
                        break;
                h[i] = h[c];
                i = c;
        }
        h[i] = x;
}

static void sm_push(sm_tracker **h, uint32_t *n, sm_tracker *t, int (*less)(const sm_tracker *, const sm_tracker *)) {
        h[++*n] = t;
        sm_up(h, *n, less);
}

static sm_tracker *sm_pop(sm_tracker **h, uint32_t *n, int (*less)(const sm_tracker *, const sm_tracker *)) {
        sm_tracker *top = h[1];
        h[1] = h[*n];
        --*n;
        if (*n)
                sm_down(h, *n, 1, less);
        return top;
}

/* Switch/prioqueue.h:178-204 try_push on the tail (capacity threshold - 1): 1 when pushed, else 0 with *evicted = the entry that
 * has to go (the former top when it compares less than v, else v itself) */
static int sm_tail_try_push(to_some *s, sm_tracker *v, sm_tracker **evicted) {
        if (s->ntail < s->captail) {
                sm_push(s->tail, &s->ntail, v, sm_less_cost);
                return 1;
        }
        if (s->ntail && sm_less_cost(s->tail[1], v)) {
                *evicted = s->tail[1];
                s->tail[1] = v;
                sm_down(s->tail, s->ntail, 1, sm_less_cost);
                return 0;
        }
        *evicted = v;
        return 0;
}

static void sm_add_lead(to_some *s, sm_tracker *t) { /* docset_iterators.h:111-115 */
        t->next = s->lead;
        s->lead = t;
        ++s->curCnt;
}

static void sm_update_current(to_some *s) { /* docset_iterators.cpp:679-691 */
        s->lead = sm_pop(s->head, &s->nhead, sm_less_id);
        s->lead->next = NULL;
        s->curCnt = 1;
        s->it.cur = s->lead->id;
        while (s->nhead && s->head[1]->id == s->it.cur)
                sm_add_lead(s, sm_pop(s->head, &s->nhead, sm_less_id));
}

static void sm_advance_tail(to_some *s, sm_tracker *top) { /* :787-794 */
        top->id = top->it->advance(top->it, s->it.cur);
        if (top->id == s->it.cur)
                sm_add_lead(s, top);
        else
                sm_push(s->head, &s->nhead, top, sm_less_id);
}

static uint32_t sm_next_impl(to_some *s) { /* :693-709 */
        while (s->curCnt < s->threshold) {
                if (s->curCnt + s->ntail >= s->threshold)
                        sm_advance_tail(s, sm_pop(s->tail, &s->ntail, sm_less_cost));
                else {
                        for (sm_tracker *t = s->lead; t; t = t->next)
                                sm_push(s->tail, &s->ntail, t, sm_less_cost);
                        sm_update_current(s);
                }
        }
        return s->it.cur;
}

static uint32_t some_next(to_iter *self) { /* :745-761 */
        to_some *s = (to_some *)self;
        const uint32_t doc = s->it.cur;
        sm_tracker *evicted = NULL;
        for (sm_tracker *t = s->lead, *nx; t; t = nx) {
                nx = t->next; /* (the list is rebuilt by update_current; a pushed tracker's link is dead) */
                if (!sm_tail_try_push(s, t, &evicted)) {
                        evicted->id = evicted->id == doc ? evicted->it->next(evicted->it) : evicted->it->advance(evicted->it, doc + 1);
                        sm_push(s->head, &s->nhead, evicted, sm_less_id);
                }
        }
        sm_update_current(s);
        return sm_next_impl(s);
}

static uint32_t some_advance(to_iter *self, uint32_t target) { /* :763-785 */
        to_some *s = (to_some *)self;
        sm_tracker *evicted = NULL;
        for (sm_tracker *t = s->lead, *nx; t; t = nx) {
                nx = t->next;
                if (!sm_tail_try_push(s, t, &evicted)) {
                        evicted->id = evicted->it->advance(evicted->it, target);
                        sm_push(s->head, &s->nhead, evicted, sm_less_id);
                }
        }
        for (sm_tracker *top = s->head[1]; top->id < target; top = s->head[1]) {
                /* the tail is full here (it holds threshold - 1 entries and at least threshold were offered) */
                sm_tail_try_push(s, top, &evicted);
                evicted->id = evicted->it->advance(evicted->it, target);
                s->head[1] = evicted;
                sm_down(s->head, s->nhead, 1, sm_less_id);
        }
        sm_update_current(s);
        return sm_next_impl(s);
}

static void some_update_matched_cnt(to_some *s) { /* :796-811: the tail may hold more matches of the current document */
        for (uint32_t i = s->ntail; i;)
                sm_advance_tail(s, s->tail[i--]);
        s->ntail = 0;
}

static double some_score(to_iter *self) { /* docset_iterators_scorers.cpp:38-57 */
        to_some *s = (to_some *)self;
        double sum = 0;
        some_update_matched_cnt(s);
        for (sm_tracker *t = s->lead; t; t = t->next)
                sum += t->it->score(t->it);
        return sum;
}

typedef struct pnode {
        uint32_t op, term;
        struct pnode **kids;
        uint32_t nkids;
        uint64_t cost;
        int empty; /* can never match (a term with documents == 0 under AND / PHRASE) */
} pnode;

static pnode *pn_new(to_ctx *c, uint32_t op) {
        pnode *n = (pnode *)ctx_own(c, xcalloc(1, sizeof *n));
        n->op = op;
        return n;
}

/* Parse the postfix program, flatten nested AND/AND and OR/OR (exec.cpp:339-358, 382-393),
 * drop never-matching operands of OR, propagate emptiness through AND/PHRASE (what the reference's
 * compiler does with unknown terms: compilation_ctx.cpp constfalse propagation), and compute costs
 * (exec.cpp:35-110 reorder_execnode_impl / docset_iterators.cpp:10-64 cost()). */
static pnode *parse_prog(to_ctx *c, const uint32_t *prog, uint32_t len) {
        pnode **stack = (pnode **)ctx_own(c, xmalloc(sizeof(pnode *) * (len + 1)));
        uint32_t sp = 0;
        for (uint32_t i = 0; i < len; ++i) {
                const uint32_t op = TO_TOK_OP(prog[i]), arg = TO_TOK_ARG(prog[i]);
                if (op == TO_OP_TERM) {
                        pnode *n = pn_new(c, op);
                        n->term = arg;
                        /* unknown term == no documents (index_source.h:60-72 term_ctx -> {0,{}}) */
                        n->cost = arg < c->ix->nterms ? c->ix->terms[arg].documents : 0;
                        n->empty = n->cost == 0;
                        stack[sp++] = n;
                        continue;
                }
                const uint32_t nk = op == TO_OP_SOME ? (arg & 0xffffu) : arg; /* operands taken off the stack */
                if (nk < 1 || nk > sp)
                        return NULL;
                pnode *n = pn_new(c, op);
                pnode **kids = stack + (sp - nk);
                n->kids = (pnode **)ctx_own(c, xmalloc(sizeof(pnode *) * (len + 1)));
                if (op == TO_OP_PHRASE) {
                        if (arg > MAX_PHRASE)
                                return NULL;
                        for (uint32_t k = 0; k < arg; ++k) {
                                if (kids[k]->op != TO_OP_TERM)
                                        return NULL;
                                n->kids[n->nkids++] = kids[k];
                                n->empty |= kids[k]->empty;
                        }
                        /* exec.cpp:28-34 phrase_cost */
                        n->cost = (uint64_t)kids[0]->cost + UINT32_MAX + (uint64_t)UINT16_MAX * arg;
                } else if (op == TO_OP_AND) {
                        for (uint32_t k = 0; k < arg; ++k) {
                                n->empty |= kids[k]->empty;
                                if (kids[k]->op == TO_OP_AND)
                                        for (uint32_t j = 0; j < kids[k]->nkids; ++j)
                                                n->kids[n->nkids++] = kids[k]->kids[j];
                                else
                                        n->kids[n->nkids++] = kids[k];
                        }
                        /* lowest cost first (exec.cpp:154-170 sort by df; 44-55 lhs/rhs swap); stable */
                        for (uint32_t a = 1; a < n->nkids; ++a) {
                                pnode *x = n->kids[a];
                                uint32_t b = a;
                                while (b > 0 && n->kids[b - 1]->cost > x->cost) {
                                        n->kids[b] = n->kids[b - 1];
                                        --b;
                                }
                                n->kids[b] = x;
                        }
                        n->cost = n->kids[0]->cost;
                } else if (op == TO_OP_OR) {
                        for (uint32_t k = 0; k < arg; ++k) {
                                if (kids[k]->empty)
                                        continue;
                                if (kids[k]->op == TO_OP_OR)
                                        for (uint32_t j = 0; j < kids[k]->nkids; ++j)
                                                n->kids[n->nkids++] = kids[k]->kids[j];
                                else
                                        n->kids[n->nkids++] = kids[k];
                        }
                        n->empty = n->nkids == 0;
                        for (uint32_t j = 0; j < n->nkids; ++j)
                                n->cost += n->kids[j]->cost;
                } else if (op == TO_OP_SOME) {
                        const uint32_t cnt = arg & 0xffffu, min = arg >> 16;
                        if (!min || min > cnt)
                                return NULL;
                        uint32_t alive = 0;
                        for (uint32_t k = 0; k < cnt; ++k)
                                if (!kids[k]->empty) {
                                        n->kids[n->nkids++] = kids[k];
                                        ++alive;
                                }
                        n->term = min; /* the threshold rides in the otherwise unused field */
                        n->empty = alive < min;
                        { /* docset_iterators.cpp:733-742: the sum of the (cnt - min + 1) smallest costs */
                                uint64_t cs[64];
                                uint32_t m = 0;
                                for (uint32_t k = 0; k < n->nkids && m < 64; ++k)
                                        cs[m++] = n->kids[k]->cost;
                                for (uint32_t a = 1; a < m; ++a) {
                                        const uint64_t x = cs[a];
                                        uint32_t b = a;
                                        while (b > 0 && cs[b - 1] > x) {
                                                cs[b] = cs[b - 1];
                                                --b;
                                        }
                                        cs[b] = x;
                                }
                                const uint32_t take = m >= min ? m - min + 1 : 0;
                                for (uint32_t k = 0; k < take; ++k)
                                        n->cost += cs[k];
                        }
                        sp -= cnt;
                        stack[sp++] = n;
                        continue;
                } else if (op == TO_OP_OPT) {
                        if (arg != 2)
                                return NULL;
                        if (kids[1]->empty) { /* an optional side that can never match adds nothing */
                                sp -= arg;
                                stack[sp++] = kids[0];
                                continue;
                        }
                        n->kids[n->nkids++] = kids[0];
                        n->kids[n->nkids++] = kids[1];
                        n->empty = kids[0]->empty;
                        n->cost = kids[0]->cost; /* exec.cpp:71-76: the cost of the other side of the AND */
                } else if (op == TO_OP_NOT) {
                        if (arg != 2)
                                return NULL;
                        /* exec.cpp:55-60: cost of a logicalnot is its lhs'; an excluded operand that can never match leaves
                         * the required one alone (compilation_ctx.cpp: [a NOT <constfalse>] => a) */
                        if (kids[1]->empty) {
                                sp -= arg;
                                stack[sp++] = kids[0];
                                continue;
                        }
                        n->kids[n->nkids++] = kids[0];
                        n->kids[n->nkids++] = kids[1];
                        n->empty = kids[0]->empty;
                        n->cost = kids[0]->cost;
                } else
                        return NULL;
                sp -= arg;
                stack[sp++] = n;
        }
        return sp == 1 ? stack[0] : NULL;
}

/* exec.cpp:253-449 build_iterator (+ docset_iterators_scorers.cpp wrap_iterator) */
static to_iter *build_iter(to_ctx *c, const pnode *n) {
        switch (n->op) {
                case TO_OP_TERM: {
                        to_pli *p = (to_pli *)ctx_own(c, to_pli_new(c->ix, n->term));
                        ctx_term_id(c, n->term);
                        return &p->it;
                }
                case TO_OP_PHRASE: {
                        if (n->nkids == 1)
                                return build_iter(c, n->kids[0]);
                        to_phrase *ph = (to_phrase *)ctx_own(c, xcalloc(1, sizeof *ph));
                        ph->it.type = IT_PHRASE;
                        ph->it.next = phrase_next;
                        ph->it.advance = phrase_advance;
                        ph->it.score = phrase_score;
                        ph->sim = c->ix->similarity;
                        ph->it.cost = n->cost;
                        ph->size = ph->nterms = (uint16_t)n->nkids;
                        ph->its = (to_pli **)ctx_own(c, xmalloc(sizeof(to_pli *) * n->nkids));
                        /* exec.cpp:296: trackCnt = AccumulatedScoreScheme */
                        ph->maxMatchCnt = (c->flags & TO_FLAG_ACCUM_SCORE) ? UINT16_MAX : 1;
                        for (uint32_t i = 0; i < n->nkids; ++i) {
                                ph->its[i] = (to_pli *)ctx_own(c, to_pli_new(c->ix, n->kids[i]->term));
                                ph->execTermID[i] = ctx_term_id(c, n->kids[i]->term);
                                ph->weight += to_sim_weight(c->ix->similarity, c->ix->terms[n->kids[i]->term].documents, c->ix->docsCnt);
                        }
                        ph->dws = (to_dws *)ctx_own(c, xcalloc(1, sizeof(to_dws)));
                        ph->hits0 = (uint16_t *)ctx_own(c, xmalloc(sizeof(uint16_t) * 65536));
                        ph->scratch = (uint16_t *)ctx_own(c, xmalloc(sizeof(uint16_t) * 65536));
                        return &ph->it;
                }
                case TO_OP_AND: {
                        to_conj_s *cj = (to_conj_s *)ctx_own(c, xcalloc(1, sizeof *cj));
                        cj->c.it.type = IT_CONJ;
                        cj->c.it.next = conj_next;
                        cj->c.it.advance = conj_advance;
                        cj->c.it.score = conj_score;
                        cj->c.it.cost = n->cost;
                        cj->c.size = cj->nscore = (uint16_t)n->nkids;
                        cj->c.its = (to_iter **)ctx_own(c, xmalloc(sizeof(to_iter *) * n->nkids));
                        for (uint32_t i = 0; i < n->nkids; ++i)
                                cj->c.its[i] = build_iter(c, n->kids[i]);
                        return &cj->c.it;
                }
                case TO_OP_SOME: { /* exec.cpp:276-283 */
                        const uint32_t cnt = n->nkids, min = n->term;
                        to_some *s = (to_some *)ctx_own(c, xcalloc(1, sizeof *s));
                        s->it.type = IT_SOME;
                        s->it.next = some_next;
                        s->it.advance = some_advance;
                        s->it.score = some_score;
                        s->it.cost = n->cost;
                        s->threshold = (uint16_t)min;
                        s->store = (sm_tracker *)ctx_own(c, xcalloc(cnt + 1, sizeof(sm_tracker)));
                        s->head = (sm_tracker **)ctx_own(c, xcalloc(cnt + 2, sizeof(sm_tracker *)));
                        s->tail = (sm_tracker **)ctx_own(c, xcalloc(cnt + 2, sizeof(sm_tracker *)));
                        s->captail = min - 1;
                        for (uint32_t i = 0; i < cnt; ++i) { /* docset_iterators.cpp:711-731: everything starts on the lead list */
                                sm_tracker *t = &s->store[i];
                                t->it = build_iter(c, n->kids[i]);
                                t->cost = n->kids[i]->cost;
                                t->id = 0;
                                sm_add_lead(s, t);
                        }
                        return &s->it;
                }
                case TO_OP_OPT: { /* exec.cpp:366-377 */
                        to_optional *o = (to_optional *)ctx_own(c, xcalloc(1, sizeof *o));
                        o->it.type = IT_OPTIONAL;
                        o->it.next = optional_next;
                        o->it.advance = optional_advance;
                        o->it.score = optional_score;
                        o->it.cost = n->cost;
                        o->main = build_iter(c, n->kids[0]);
                        o->opt = build_iter(c, n->kids[1]);
                        return &o->it;
                }
                case TO_OP_NOT: { /* exec.cpp:424-427 */
                        to_filter *f = (to_filter *)ctx_own(c, xcalloc(1, sizeof *f));
                        f->it.type = IT_FILTER;
                        f->it.next = filter_next;
                        f->it.advance = filter_advance;
                        f->it.score = filter_score;
                        f->it.cost = n->cost;
                        f->req = build_iter(c, n->kids[0]);
                        f->filter = build_iter(c, n->kids[1]);
                        return &f->it;
                }
                case TO_OP_OR: {
                        if (n->nkids == 1)
                                return build_iter(c, n->kids[0]);
                        to_disj *d = (to_disj *)ctx_own(c, xcalloc(1, sizeof *d));
                        d->it.type = IT_DISJ;
                        d->it.next = disj_next;
                        d->it.advance = disj_advance;
                        d->it.score = disj_score;
                        d->it.cost = n->cost;
                        d->heap = (to_iter **)ctx_own(c, xmalloc(sizeof(to_iter *) * (n->nkids + 2)));
                        for (uint32_t i = 0; i < n->nkids; ++i)
                                heap_push(d->heap, &d->n, build_iter(c, n->kids[i])); /* all at cur == 0 */
                        return &d->it;
                }
        }
        return NULL;
}

/* ================================================================== spans + handlers */
static void res_push(to_result *r, uint32_t id, double score, int scored) {
        if (r->n == r->cap) {
                r->cap = r->cap ? r->cap * 2 : 1024;
                r->docs = (uint32_t *)xrealloc(r->docs, sizeof(uint32_t) * r->cap);
                if (scored)
                        r->scores = (double *)xrealloc(r->scores, sizeof(double) * r->cap);
        }
        r->docs[r->n] = id;
        if (scored)
                r->scores[r->n] = score;
        ++r->n;
}

/* docset_spans.cpp:269-290 GenericDocsSetSpan::process(mp, 1, DocIDsEND) fast path, with the
 * exec.cpp:1213-1229 (docs-only) / 1322-1341 (accumulated score) no-filter handlers inlined */
static void span_generic(to_iter *it, uint32_t flags, to_result *out) {
        const int scored = (flags & TO_FLAG_ACCUM_SCORE) != 0;
        for (uint32_t id = it->next(it); id != TO_DOCIDS_END; id = it->next(it))
                res_push(out, id, scored ? it->score(it) : 0.0, scored);
}

/* docset_spans.cpp:98-173 (DocsSetSpanForDisjunctions) and 681-790 (…WithThreshold(1, its, true)) */
static void span_disjunction(to_ctx *c, to_iter **its, uint32_t cnt, uint32_t flags, to_result *out) {
        const int scored = (flags & TO_FLAG_ACCUM_SCORE) != 0;
        uint64_t *matching = (uint64_t *)ctx_own(c, xcalloc(SPAN_SIZE / 64, sizeof(uint64_t)));
        double *trkScore = (double *)ctx_own(c, xcalloc(SPAN_SIZE, sizeof(double)));
        to_iter **heap = (to_iter **)ctx_own(c, xmalloc(sizeof(to_iter *) * (cnt + 2)));
        to_iter **collected = (to_iter **)ctx_own(c, xmalloc(sizeof(to_iter *) * (cnt + 1)));
        uint32_t hn = 0;
        for (uint32_t i = 0; i < cnt; ++i) { /* docset_spans.h:157-163, 208-214: ctor primes children */
                its[i]->next(its[i]);
                heap_push(heap, &hn, its[i]);
        }
        const uint32_t max = TO_DOCIDS_END;
        for (;;) {
                to_iter *it = heap[1];
                const uint32_t id = it->cur;
                if (id >= max)
                        break;
                const uint32_t windowBase = id & ~SPAN_MASK;
                const uint64_t wm = (uint64_t)windowBase + SPAN_SIZE;
                const uint32_t windowMax = wm < max ? (uint32_t)wm : max;
                uint32_t collectedCnt = 1;
                collected[0] = it;
                for (heap_pop(heap, &hn); hn && (it = heap[1])->cur < windowMax; heap_pop(heap, &hn))
                        collected[collectedCnt++] = it;
                if (collectedCnt == 1) { /* docset_spans.cpp:120-129 / 712-719 */
                        to_iter *one = collected[0];
                        for (uint32_t d = one->cur; d < windowMax; d = one->next(one))
                                res_push(out, d, scored ? one->score(one) : 0.0, scored);
                        heap_push(heap, &hn, one);
                } else {
                        uint32_t m = 0;
                        for (uint32_t i_ = 0; i_ != collectedCnt; ++i_) {
                                to_iter *ci = collected[i_];
                                for (uint32_t d = ci->cur; d < windowMax; d = ci->next(ci)) {
                                        const uint32_t i = d - windowBase, mi = i >> 6;
                                        if (mi > m)
                                                m = mi;
                                        matching[mi] |= (uint64_t)1 << (i & 63);
                                        if (scored)
                                                trkScore[i] += ci->score(ci);
                                }
                                heap_push(heap, &hn, ci);
                        }
                        for (uint32_t idx = 0; idx <= m; ++idx) {
                                for (uint64_t b = matching[idx]; b;) {
                                        const uint32_t bidx = (uint32_t)__builtin_ctzll(b);
                                        const uint32_t translated = (idx << 6) + bidx;
                                        b ^= (uint64_t)1 << bidx;
                                        res_push(out, windowBase + translated, trkScore[translated], scored);
                                        trkScore[translated] = 0;
                                }
                        }
                        memset(matching, 0, (m + 1) * sizeof(uint64_t));
                }
        }
}

/* the result buffers of `out` are reused when it already owns some (n is reset): the multi-threaded baseline keeps one result
 * per thread — the application-side collector a MatchedIndexDocumentsFilter would be — instead of growing a fresh one per
 * query (large reallocations are mmap/munmap calls, which serialise the threads in the kernel) */
static int exec_query_into(const to_index *ix, const uint32_t *prog, uint32_t proglen, uint32_t flags, to_result *out);

int to_exec_query(const to_index *ix, const uint32_t *prog, uint32_t proglen, uint32_t flags, to_result *out) {
        memset(out, 0, sizeof *out);
        return exec_query_into(ix, prog, proglen, flags, out);
}

static int exec_query_into(const to_index *ix, const uint32_t *prog, uint32_t proglen, uint32_t flags, to_result *out) {
        out->n = 0;
        if ((flags & (TO_FLAG_DOCUMENTS_ONLY | TO_FLAG_ACCUM_SCORE)) == 0 ||
            (flags & (TO_FLAG_DOCUMENTS_ONLY | TO_FLAG_ACCUM_SCORE)) == (TO_FLAG_DOCUMENTS_ONLY | TO_FLAG_ACCUM_SCORE))
                return -1; /* exec.h:45-48: mutually exclusive; the default rich mode is out of scope */
        to_ctx c;
        memset(&c, 0, sizeof c);
        c.ix = ix;
        c.flags = flags;
        int rc = 0;
        pnode *root = parse_prog(&c, prog, proglen);
        if (!root)
                rc = -2;
        else if (!root->empty) {
                to_iter *sit = build_iter(&c, root);
                /* exec.cpp:452-505 build_span: a root disjunction hands its children to the window span */
                if (sit->type == IT_DISJ) {
                        to_disj *d = (to_disj *)sit;
                        to_iter **kids = (to_iter **)ctx_own(&c, xmalloc(sizeof(to_iter *) * d->n));
                        const uint32_t cnt = d->n;
                        for (uint32_t i = 0; i < cnt; ++i)
                                kids[i] = d->heap[i + 1];
                        span_disjunction(&c, kids, cnt, flags, out);
                } else
                        span_generic(sit, flags, out);
        }
        /* exec.cpp:914-975 / 1000-1030: a match is handed to consider() only if !maskedDocumentsRegistry->test(id).  Both lists
         * ascend, so one merge pass drops the masked matches (and their scores). */
        if (ix->nmasked && out->n) {
                size_t w = 0, mi = 0;
                for (size_t i = 0; i < out->n; ++i) {
                        const uint32_t d = out->docs[i];
                        while (mi < ix->nmasked && ix->masked[mi] < d)
                                ++mi;
                        if (mi < ix->nmasked && ix->masked[mi] == d)
                                continue;
                        out->docs[w] = d;
                        if (out->scores)
                                out->scores[w] = out->scores[i];
                        ++w;
                }
                out->n = w;
        }
        for (size_t i = 0; i < c.nowned; ++i)
                free(c.owned[i]);
        free(c.owned);
        return rc;
}

/* ================================================================== default ("rich match") mode */
typedef struct {
        uint32_t terms[64];
        uint32_t n;
} termset;

static void termset_add(termset *ts, uint32_t t) {
        for (uint32_t i = 0; i < ts->n; ++i)
                if (ts->terms[i] == t)
                        return;
        if (ts->n < 64)
                ts->terms[ts->n++] = t;
}

/* queryexec_ctx.cpp:382-520 collect_doc_matching_terms: the postings iterators that sit on docID, through the tree */
static void collect_terms(to_iter *it, uint32_t doc, termset *out) {
        switch (it->type) {
                case IT_PLI:
                        termset_add(out, ((to_pli *)it)->term);
                        break;
                case IT_PHRASE: {
                        to_phrase *ph = (to_phrase *)it;
                        for (uint16_t i = 0; i < ph->size; ++i)
                                termset_add(out, ph->its[i]->term);
                } break;
                case IT_CONJ: {
                        to_conj_s *c = (to_conj_s *)it;
                        for (uint16_t i = 0; i < c->nscore; ++i)
                                collect_terms(c->c.its[i], doc, out);
                } break;
                case IT_DISJ: {
                        to_disj *d = (to_disj *)it;
                        for (uint32_t i = 1; i <= d->n; ++i)
                                if (d->heap[i]->cur == doc)
                                        collect_terms(d->heap[i], doc, out);
                } break;
                case IT_FILTER:
                        collect_terms(((to_filter *)it)->req, doc, out);
                        break;
                case IT_SOME: { /* queryexec_ctx.cpp:396-409 */
                        to_some *s = (to_some *)it;
                        some_update_matched_cnt(s);
                        for (sm_tracker *t = s->lead; t; t = t->next)
                                collect_terms(t->it, doc, out);
                } break;
                case IT_OPTIONAL: { /* queryexec_ctx.cpp:418-432 */
                        to_optional *o = (to_optional *)it;
                        collect_terms(o->main, doc, out);
                        if (optional_opt_matches(o))
                                collect_terms(o->opt, doc, out);
                } break;
        }
}

static void rich_push(to_rich *r, uint32_t v) {
        if (r->nflat == r->capflat) {
                r->capflat = r->capflat ? r->capflat * 2 : 4096;
                r->flat = (uint32_t *)xrealloc(r->flat, sizeof(uint32_t) * r->capflat);
        }
        r->flat[r->nflat++] = v;
}

int to_exec_query_rich(const to_index *ix, const uint32_t *prog, uint32_t proglen, to_rich *out) {
        memset(out, 0, sizeof *out);
        to_ctx c;
        memset(&c, 0, sizeof c);
        c.ix = ix;
        c.flags = 0; /* neither DocumentsOnly nor AccumulatedScoreScheme: phrases stop at the first match (exec.cpp:296) */
        int rc = 0;
        pnode *root = parse_prog(&c, prog, proglen);
        if (!root)
                rc = -2;
        else if (!root->empty) {
                /* exec.cpp:452-505: outside the two fast modes the root iterator itself drives the execution (GenericDocsSetSpan) */
                to_iter *sit = build_iter(&c, root);
                /* hits come from separate postings iterators (one per distinct term): the tree's own have been consumed by the
                 * phrase checks, which is what the reference's term_hits cache per (term, document) is for (prepare_match) */
                to_pli *hp[64];
                uint32_t hterm[64], nh = 0;
                uint16_t *pos = (uint16_t *)xmalloc(sizeof(uint16_t) * 65536);
                size_t capdocs = 0, mi = 0;
                for (uint32_t doc = sit->next(sit); doc != TO_DOCIDS_END; doc = sit->next(sit)) {
                        while (mi < ix->nmasked && ix->masked[mi] < doc)
                                ++mi;
                        if (mi < ix->nmasked && ix->masked[mi] == doc)
                                continue; /* exec.cpp:1350-1380: masked documents never reach prepare_match */
                        termset ts;
                        ts.n = 0;
                        collect_terms(sit, doc, &ts);
                        /* canonical order: ascending term rank */
                        for (uint32_t a = 1; a < ts.n; ++a) {
                                const uint32_t x = ts.terms[a];
                                uint32_t b = a;
                                while (b > 0 && ts.terms[b - 1] > x) {
                                        ts.terms[b] = ts.terms[b - 1];
                                        --b;
                                }
                                ts.terms[b] = x;
                        }
                        if (out->n == capdocs) {
                                capdocs = capdocs ? capdocs * 2 : 1024;
                                out->docs = (uint32_t *)xrealloc(out->docs, sizeof(uint32_t) * capdocs);
                        }
                        out->docs[out->n++] = doc;
                        rich_push(out, doc);
                        rich_push(out, ts.n);
                        for (uint32_t i = 0; i < ts.n; ++i) {
                                uint32_t k = 0;
                                while (k < nh && hterm[k] != ts.terms[i])
                                        ++k;
                                if (k == nh) {
                                        if (nh == 64)
                                                abort();
                                        hterm[nh] = ts.terms[i];
                                        hp[nh++] = (to_pli *)ctx_own(&c, to_pli_new(ix, ts.terms[i]));
                                }
                                to_pli *p = hp[k];
                                if (p->it.cur < doc)
                                        p->it.advance(&p->it, doc);
                                if (p->it.cur != doc)
                                        abort(); /* a collected term holds the document by construction */
                                const uint32_t f = p->materialize(p, pos);
                                rich_push(out, ts.terms[i]);
                                rich_push(out, f);
                                for (uint32_t h = 0; h < f; ++h)
                                        rich_push(out, pos[h]);
                                out->hits_total += f;
                        }
                        out->terms_total += ts.n;
                }
                free(pos);
        }
        for (size_t i = 0; i < c.nowned; ++i)
                free(c.owned[i]);
        free(c.owned);
        return rc;
}

void to_rich_free(to_rich *r) {
        free(r->docs);
        free(r->flat);
        memset(r, 0, sizeof *r);
}

void to_result_free(to_result *r) {
        free(r->docs);
        free(r->scores);
        memset(r, 0, sizeof *r);
}

/* Application-side top-K (matches.h:139-185 leaves ranking to the application).
 * Order: score descending, docID ascending. */
uint32_t to_topk(const to_result *r, uint32_t k, uint32_t *docs, float *scores) {
        if (!k)
                return 0;
        uint32_t *hd = (uint32_t *)xmalloc(sizeof(uint32_t) * k);
        double *hs = (double *)xmalloc(sizeof(double) * k);
        uint32_t n = 0;
        /* "worse" = lower score, or equal score and larger docID; keep a min-heap on "better" */
#define WORSE(s1, d1, s2, d2) ((s1) < (s2) || ((s1) == (s2) && (d1) > (d2)))
        for (size_t i = 0; i < r->n; ++i) {
                const double s = r->scores ? r->scores[i] : 0.0;
                const uint32_t d = r->docs[i];
                if (n < k) {
                        uint32_t j = n++;
                        while (j > 0) {
                                uint32_t pa = (j - 1) / 2;
                                if (WORSE(s, d, hs[pa], hd[pa])) {
                                        hs[j] = hs[pa];
                                        hd[j] = hd[pa];
                                        j = pa;
                                } else
                                        break;
                        }
                        hs[j] = s;
                        hd[j] = d;
                } else if (WORSE(hs[0], hd[0], s, d)) {
                        uint32_t j = 0;
                        for (;;) {
                                uint32_t l = 2 * j + 1, rr = l + 1, m = l;
                                if (l >= n)
                                        break;
                                if (rr < n && WORSE(hs[rr], hd[rr], hs[l], hd[l]))
                                        m = rr;
                                if (WORSE(hs[m], hd[m], s, d)) {
                                        hs[j] = hs[m];
                                        hd[j] = hd[m];
                                        j = m;
                                } else
                                        break;
                        }
                        hs[j] = s;
                        hd[j] = d;
                }
        }
        /* selection sort out of the heap: best first */
        for (uint32_t i = 0; i < n; ++i) {
                uint32_t best = i;
                for (uint32_t j = i + 1; j < n; ++j)
                        if (WORSE(hs[best], hd[best], hs[j], hd[j]))
                                best = j;
                double ts = hs[i];
                hs[i] = hs[best];
                hs[best] = ts;
                uint32_t td = hd[i];
                hd[i] = hd[best];
                hd[best] = td;
                docs[i] = hd[i];
                scores[i] = (float)hs[i];
        }
#undef WORSE
        free(hd);
        free(hs);
        return n;
}

uint64_t to_fnv1a_docs(const uint32_t *docs, size_t n) {
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < n; ++i) {
                uint32_t d = docs[i];
                for (int b = 0; b < 4; ++b) {
                        h = (h ^ (d & 0xff)) * 1099511628211ull;
                        d >>= 8;
                }
        }
        return h;
}

/* ------------------------------------------------------------------ timed multi-threaded batch (bench.py cpu_baseline) */
/* One query per thread (exec_query is re-entrant per thread in the reference: exec.cpp:12 thread_local curRCTX; exec.h:132-154
 * exec_query_par runs one std::async per source).  Threads draw programs from a shared cursor until the batch or the time
 * budget is exhausted.  Results are counted and discarded. */
#include <pthread.h>
#include <time.h>

typedef struct {
        const to_index *ix;
        const uint32_t *progs;
        const uint32_t *order; /* query indices, heaviest first (longest-processing-time-first keeps the tail short) */
        uint32_t proglen, nq, flags;
        double deadline;
        uint64_t cursor, done, matches;
        pthread_mutex_t mu;
} mt_state;

static double mt_now(void) {
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *mt_worker(void *arg) {
        mt_state *st = (mt_state *)arg;
        uint64_t done = 0, matches = 0;
        to_result r; /* this thread's collector, reused from query to query */
        memset(&r, 0, sizeof r);
        for (;;) {
                if (mt_now() >= st->deadline)
                        break;
                const uint64_t at = __atomic_fetch_add(&st->cursor, 1, __ATOMIC_RELAXED);
                if (at >= st->nq)
                        break;
                const uint64_t i = st->order[at];
                if (exec_query_into(st->ix, st->progs + i * st->proglen, st->proglen, st->flags, &r) == 0) {
                        ++done;
                        matches += r.n;
                }
        }
        to_result_free(&r);
        pthread_mutex_lock(&st->mu);
        st->done += done;
        st->matches += matches;
        pthread_mutex_unlock(&st->mu);
        return NULL;
}

/* nq programs of `proglen` tokens each, back to back in `progs`.  Returns the queries completed; *out_matches and
 * *out_seconds (wall) are filled. */
uint64_t to_exec_batch_mt(const to_index *ix, const uint32_t *progs, uint32_t proglen, uint32_t nq, uint32_t flags, uint32_t nthreads,
                          double budget_seconds, uint64_t *out_matches, double *out_seconds) {
        mt_state st;
        memset(&st, 0, sizeof st);
        st.ix = ix;
        st.progs = progs;
        st.proglen = proglen;
        st.nq = nq;
        st.flags = flags;
        pthread_mutex_init(&st.mu, NULL);
        if (!nthreads)
                nthreads = 1;
        /* heaviest queries first: cost = sum of the document frequencies of the program's terms */
        uint64_t *cost = (uint64_t *)xmalloc(sizeof(uint64_t) * (nq ? nq : 1));
        uint32_t *order = (uint32_t *)xmalloc(sizeof(uint32_t) * (nq ? nq : 1));
        for (uint32_t q = 0; q < nq; ++q) {
                uint64_t c = 0;
                for (uint32_t k = 0; k < proglen; ++k) {
                        const uint32_t tok = progs[(size_t)q * proglen + k];
                        if (TO_TOK_OP(tok) == TO_OP_TERM && TO_TOK_ARG(tok) < ix->nterms)
                                c += ix->terms[TO_TOK_ARG(tok)].documents;
                }
                cost[q] = c;
                order[q] = q;
        }
        /* a stable counting-free sort is enough here: qsort on (cost desc, index asc) */
        {
                /* sort indices by cost, descending */
                uint32_t *tmp = order;
                /* simple shell sort: no extra context pointer needed */
                for (uint32_t gap = nq / 2; gap > 0; gap /= 2)
                        for (uint32_t a = gap; a < nq; ++a) {
                                const uint32_t x = tmp[a];
                                uint32_t b = a;
                                while (b >= gap && (cost[tmp[b - gap]] < cost[x] || (cost[tmp[b - gap]] == cost[x] && tmp[b - gap] > x))) {
                                        tmp[b] = tmp[b - gap];
                                        b -= gap;
                                }
                                tmp[b] = x;
                        }
        }
        st.order = order;
        pthread_t *th = (pthread_t *)xmalloc(sizeof(pthread_t) * nthreads);
        const double t0 = mt_now();
        st.deadline = t0 + budget_seconds;
        uint32_t started = 0;
        for (; started < nthreads; ++started)
                if (pthread_create(&th[started], NULL, mt_worker, &st))
                        break;
        for (uint32_t i = 0; i < started; ++i)
                pthread_join(th[i], NULL);
        if (out_seconds)
                *out_seconds = mt_now() - t0;
        if (out_matches)
                *out_matches = st.matches;
        free(th);
        free(cost);
        free(order);
        pthread_mutex_destroy(&st.mu);
        return st.done;
}
