// k_commit.hpp — SegmentIndexSession::commit on the device: the session's postings, in insertion order, sorted by (term, document) and gathered into the
// term-after-term arrays the device encoder takes.  Part of libtrinity_hip.so (MI355X / gfx950); included by trinity_hip.hip.  New code, no reference source.
//
// The reference (indexer.cpp:311-478) scans the session's buffer into one record per (document, term) — {termID, documentID, hits offset, hits count} —,
// buckets the records by termID & 31, sorts every bucket by (termID, documentID) on its own thread (:399-416) and then walks bucket after bucket through
// the encoder, strictly sequentially (:423-477; its own comment :302-309: encoding dominates).  Here the records are sorted by ONE key whose order is the
// reference's walk — bucket, then term, then document: the termID rotated right by five bits above the documentID — with a device radix sort
// (commit_sort.hip), the postings' frequencies and hits are gathered into that order, the distinct terms and their first postings fall out of the sorted
// keys, and the encoder (k_encode.hpp) runs on what is already in HBM.
#pragma once

__device__ __forceinline__ uint32_t commit_ror5(const uint32_t t) { return (t >> 5) | (t << 27); }
__device__ __forceinline__ uint32_t commit_rol5(const uint32_t t) { return (t << 5) | (t >> 27); }

__global__ void k_commit_keys(const uint32_t *__restrict__ term_ids, const uint32_t *__restrict__ doc_ids, unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals,
                              const uint64_t n) {
        const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= n)
                return;
        keys[i] = ((unsigned long long)commit_ror5(term_ids[i]) << 32) | doc_ids[i];
        vals[i] = (uint32_t)i;
}
// sorted posting j: its document, its frequency, whether it opens a term
__global__ void k_commit_gather(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ freqs_in, uint32_t *__restrict__ docs,
                                uint32_t *__restrict__ freqs, uint32_t *__restrict__ marks, const uint64_t n) {
        const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (j >= n)
                return;
        const unsigned long long k = keys[j];
        docs[j] = (uint32_t)k;
        freqs[j] = freqs_in[perm[j]];
        marks[j] = (j == 0 || (uint32_t)(keys[j - 1] >> 32) != (uint32_t)(k >> 32)) ? 1u : 0u;
}
// the hits of sorted posting j: copied from where the session left them
__global__ void k_commit_hits(const uint32_t *__restrict__ perm, const uint64_t *__restrict__ hit_off_in, const uint64_t *__restrict__ hit_off_out, const uint32_t *__restrict__ freqs,
                              const uint16_t *__restrict__ pos_in, uint16_t *__restrict__ pos_out, const uint8_t *__restrict__ plens_in, uint8_t *__restrict__ plens_out,
                              const uint64_t *__restrict__ payloads_in, uint64_t *__restrict__ payloads_out, const uint64_t n) {
        const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (j >= n)
                return;
        const uint64_t src = hit_off_in[perm[j]], dst = hit_off_out[j];
        const uint32_t f = freqs[j];
        for (uint32_t h = 0; h < f; ++h) {
                pos_out[dst + h] = pos_in[src + h];
                if (plens_in) {
                        plens_out[dst + h] = plens_in[src + h];
                        payloads_out[dst + h] = payloads_in[src + h];
                }
        }
}
// the distinct terms in commit order: where each one's postings start, and its termID
__global__ void k_commit_terms(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ marks, const uint64_t *__restrict__ mark_rank, uint64_t *__restrict__ term_first,
                               uint32_t *__restrict__ term_ids, const uint64_t n) {
        const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (j >= n || !marks[j])
                return;
        const uint64_t t = mark_rank[j];
        term_first[t] = j;
        term_ids[t] = commit_rol5((uint32_t)(keys[j] >> 32));
}
// what the reference's encoder (or commit itself) would refuse: *err (initialised to all ones) = min over the offending sorted postings of
// (posting + 1) << 8 | why (1: document 0; 2: the same (term, document) twice — require(documentID > prevDID), indexer.cpp:446; 3: a position out of order / a position-0 hit without
// payload — google_codec.cpp:42-49; 4: a payload of more than 8 bytes — :46)
__global__ void k_commit_validate(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ freqs, const uint64_t *__restrict__ hit_off, const uint16_t *__restrict__ pos,
                                  const uint8_t *__restrict__ plens, const uint64_t n, unsigned long long *__restrict__ err) {
        const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (j >= n)
                return;
        uint32_t why = 0;
        if (!(uint32_t)keys[j])
                why = 1;
        else if (j && keys[j - 1] == keys[j])
                why = 2;
        else {
                uint32_t last = 0;
                const uint64_t h0 = hit_off[j];
                for (uint32_t h = 0; h < freqs[j] && !why; ++h) {
                        const uint32_t p = pos[h0 + h], pl = plens ? plens[h0 + h] : 0u;
                        if (pl > 8)
                                why = 4;
                        else if ((!p && !pl) || p < last)
                                why = 3;
                        last = p;
                }
        }
        if (why) {
                const unsigned long long mine = ((unsigned long long)(j + 1) << 8) | why; // (the smallest posting wins)
                atomicMin(err, mine);
        }
}
