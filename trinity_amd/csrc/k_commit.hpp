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

// ------------------------------------------------------------------------------------------ merge (Codecs::Google::IndexSession::merge)
// The reference merges ONE term at a time (google_codec.cpp:186-438; driven per dictionary term by MergeCandidatesCollection::merge, merge.cpp:40-400): a
// k-way walk over the participants' chunks, most recent first — the lowest documentID wins, among equals the most recent participant, and the winner's
// document is appended (hits and payloads copied: append_from :323-369) unless that participant's masked_documents_registry masks it (:393-401); every
// participant that holds the document steps past it.  Here the whole dictionary is merged at once as a SORT: every participant's postings are decoded
// (k_merge_decode: documents, frequencies; k_merge_hits: positions and payloads), keyed (output term << 32 | documentID) in participant-major order — most
// recent first —, sorted stably (commit_sort.hip), and the first posting of every run of equal keys is the winner; it is kept unless its participant's
// masked bitmap (tri_index_set_masked) holds the document.  What is kept is gathered term after term and goes through the device encoder (k_encode.hpp).
struct MergeJob {
        uint32_t term;    // the term's index in the participant
        uint32_t out;     // the output term
        uint64_t out_off; // where the term's postings go in the concatenated arrays (dense: every block but a list's last holds 32 documents)
};
// one workgroup per job (strided): documents, frequencies, keys — one lane per block
template <int CODEC>
__global__ __launch_bounds__(256) void k_merge_decode(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last, const uint32_t *__restrict__ blk_off,
                                                      const DevTerm *__restrict__ terms, const MergeJob *__restrict__ jobs, const uint32_t njobs, uint32_t *__restrict__ freqs,
                                                      unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals) {
        for (uint32_t ji = blockIdx.x; ji < njobs; ji += gridDim.x) {
                const MergeJob job = jobs[ji];
                const DevTerm t = terms[job.term];
                for (uint32_t b = threadIdx.x; b < t.nblocks; b += blockDim.x) {
                        const uint32_t gb = t.first_block + b;
                        const uint32_t off = blk_off[gb];
                        const uint32_t n = TRI_BLOCK_N(t, b, index, off);
                        const uint32_t last = blk_last[gb];
                        uint32_t doc = b ? blk_last[gb - 1] : 0;
                        DeltaStream<CODEC> s;
                        s.init(index, t, b, off);
                        const uint64_t at = job.out_off + (uint64_t)b * 32;
                        for (uint32_t i = 0; i < n; ++i) {
                                doc = i + 1 < n ? doc + s.next() : last;
                                keys[at + i] = ((unsigned long long)job.out << 32) | doc;
                                vals[at + i] = (uint32_t)(at + i);
                        }
                        FreqStream<CODEC> fs;
                        fs.init(index, t, b, off, s);
                        for (uint32_t i = 0; i < n; ++i)
                                freqs[at + i] = fs.next();
                }
        }
}
// ... the hits of every posting (GOOGLE: they follow the block's frequencies; blk_hits[] says where, relative to the block's payload): position, payload
// length, payload bytes (first byte low) — google_codec.cpp:323-369
__global__ __launch_bounds__(256) void k_merge_hits(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_off, const uint32_t *__restrict__ blk_hits,
                                                    const DevTerm *__restrict__ terms, const MergeJob *__restrict__ jobs, const uint32_t njobs, const uint32_t *__restrict__ freqs,
                                                    const uint64_t *__restrict__ hit_off, uint16_t *__restrict__ pos_out, uint8_t *__restrict__ plens_out,
                                                    uint64_t *__restrict__ payloads_out) {
        for (uint32_t ji = blockIdx.x; ji < njobs; ji += gridDim.x) {
                const MergeJob job = jobs[ji];
                const DevTerm t = terms[job.term];
                for (uint32_t b = threadIdx.x; b < t.nblocks; b += blockDim.x) {
                        const uint32_t gb = t.first_block + b;
                        const uint32_t off = blk_off[gb];
                        const uint32_t n = TRI_BLOCK_N(t, b, index, off);
                        VbStream s;
                        s.init(index + off + (blk_hits[gb] & ~BLK_HITS_PLAIN));
                        const uint64_t at = job.out_off + (uint64_t)b * 32;
                        for (uint32_t i = 0; i < n; ++i) {
                                const uint32_t f = freqs[at + i];
                                uint64_t dst = hit_off[at + i];
                                uint32_t pos = 0, plen = 0; // (position and payload-length state restart with every document)
                                for (uint32_t h = 0; h < f; ++h, ++dst) {
                                        const uint32_t v = s.next();
                                        if (v & 1u)
                                                plen = s.byte();
                                        uint64_t payload = 0;
                                        for (uint32_t k = 0; k < plen; ++k)
                                                payload |= (uint64_t)s.byte() << (8 * k);
                                        pos += v >> 1;
                                        pos_out[dst] = (uint16_t)pos;
                                        plens_out[dst] = (uint8_t)plen;
                                        payloads_out[dst] = payload;
                                }
                        }
                }
        }
}
// ... LUCENE: a term's positions are ONE stream in hits.data, found by count (blk_hits[]: the hits before the row; hdir[]: the 128-hit blocks' offsets — k_phrase.hpp);
// this engine's Lucene-shaped segments carry no payloads (lucene_encoder.hpp), so length and word are zero
__global__ __launch_bounds__(256) void k_merge_hits_lucene(const uint8_t *__restrict__ hits, const uint32_t *__restrict__ blk_hits, const uint32_t *__restrict__ hdir,
                                                           const DevTerm *__restrict__ terms, const MergeJob *__restrict__ jobs, const uint32_t njobs, const uint32_t *__restrict__ freqs,
                                                           const uint64_t *__restrict__ hit_off, uint16_t *__restrict__ pos_out, uint8_t *__restrict__ plens_out,
                                                           uint64_t *__restrict__ payloads_out) {
        const HitCtx ctx{hits, blk_hits, hdir};
        for (uint32_t ji = blockIdx.x; ji < njobs; ji += gridDim.x) {
                const MergeJob job = jobs[ji];
                const DevTerm t = terms[job.term];
                for (uint32_t b = threadIdx.x; b < t.nblocks; b += blockDim.x) {
                        const uint32_t gb = t.first_block + b;
                        const uint32_t n = (t.flags & TERM_FULL_BLOCKS) ? (b + 1 == t.nblocks ? t.last_n : 32u) : 0u; // (merge_device takes full-block lists only)
                        const uint64_t at = job.out_off + (uint64_t)b * 32;
                        uint32_t total = 0;
                        for (uint32_t i = 0; i < n; ++i)
                                total += freqs[at + i];
                        if (!total)
                                continue;
                        HitStream<CODEC_LUCENE> s;
                        s.init(ctx, t.pad, blk_hits[gb]);
                        for (uint32_t i = 0; i < n; ++i) {
                                const uint32_t f = freqs[at + i];
                                uint64_t dst = hit_off[at + i];
                                uint32_t pos = 0; // (the position restarts with every document)
                                for (uint32_t h = 0; h < f; ++h, ++dst) {
                                        pos += s.next();
                                        pos_out[dst] = (uint16_t)pos;
                                        plens_out[dst] = 0;
                                        payloads_out[dst] = 0;
                                }
                        }
                }
        }
}
// sorted posting j: the first of a run of equal (term, document) keys — the most recent participant's — wins; kept unless its participant masks the document
__global__ void k_merge_select(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ perm, const uint64_t *__restrict__ part_first /* [nparts + 1] */,
                               const uint32_t nparts, const uint32_t *const *__restrict__ masked /* [nparts]: bitmap or null */, uint32_t *__restrict__ keep, const uint64_t n) {
        const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (j >= n)
                return;
        const unsigned long long k = keys[j];
        uint32_t kp = (j == 0 || keys[j - 1] != k) ? 1u : 0u;
        if (kp) {
                const uint64_t src = perm[j];
                uint32_t p = 0;
                while (p + 1 < nparts && part_first[p + 1] <= src)
                        ++p;
                const uint32_t doc = (uint32_t)k;
                const uint32_t *m = masked[p];
                if (m && ((m[doc >> 5] >> (doc & 31u)) & 1u))
                        kp = 0; // masked_documents_registry::test of the winner's participant (google_codec.cpp:393-401)
        }
        keep[j] = kp;
}
// the kept postings, in order: documents, frequencies (gathered from the decode order), how many each output term keeps
__global__ void k_merge_compact(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ keep, const uint64_t *__restrict__ rank,
                                const uint32_t *__restrict__ freqs_in, uint32_t *__restrict__ docs, uint32_t *__restrict__ freqs, uint32_t *__restrict__ src_of,
                                uint32_t *__restrict__ term_cnt, const uint64_t n) {
        const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (j >= n || !keep[j])
                return;
        const uint64_t o = rank[j];
        docs[o] = (uint32_t)keys[j];
        freqs[o] = freqs_in[perm[j]];
        src_of[o] = perm[j];
        atomicAdd(&term_cnt[(uint32_t)(keys[j] >> 32)], 1u);
}
