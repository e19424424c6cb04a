/* oracle_internal.h — shared internals of the CPU oracle (test infrastructure only; see trinity_oracle.h). */
#ifndef ORACLE_INTERNAL_H
#define ORACLE_INTERNAL_H
#include "trinity_oracle.h"

void *xmalloc(size_t n);
void *xcalloc(size_t n, size_t s);
void *xrealloc(void *q, size_t n);

typedef struct {
        uint8_t *d;
        size_t n, cap;
} buf_t;
void buf_room(buf_t *b, size_t extra);
void buf_varbyte(buf_t *b, uint32_t v);
void buf_u8(buf_t *b, uint8_t v);
void buf_u32(buf_t *b, uint32_t v);
void buf_bytes(buf_t *b, const void *p, size_t n);

enum { IT_PLI = 0, IT_CONJ, IT_DISJ, IT_PHRASE, IT_FILTER, IT_OPTIONAL, IT_SOME };

typedef struct to_iter to_iter;
struct to_iter { /* docset_iterators_base.h:45-96 Iterator + relevant_documents.h:43-67 IteratorScorer */
        uint8_t type;
        uint32_t cur; /* curDocument.id; 0 before the first next() */
        uint32_t (*next)(to_iter *);
        uint32_t (*advance)(to_iter *, uint32_t);
        double (*score)(to_iter *);
        uint64_t cost;
};

/* what every postings-list iterator starts with, whatever the codec (codecs.h:211-246) */
#define TO_PLI_HEAD                                                  \
        to_iter it;                                                  \
        uint16_t freq; /* codecs.h:217 tokenpos_t */                 \
        double idf;    /* the term's ScorerWeight (docset_iterators_scorers.cpp:10-36): BM25 idf, TF-IDF idf, unused for Trivial */ \
        int sim;       /* TO_SIM_* of the index at creation time */  \
        uint32_t term, documents;                                    \
        uint32_t (*materialize)(struct to_pli *, uint16_t *out_pos); \
        void (*destroy)(struct to_pli *);

/* the Lucene-shaped codec lives in trinity_oracle_lucene.c */
to_pli *to_lucene_pli_new(const to_index *, uint32_t term);
double to_pli_score_bm25(to_iter *self);

#endif
