// k_decode.hpp — k_decode_terms: whole postings lists, one lane per block (codec seam)
// Part of libtrinity_hip.so (MI355X / gfx950); included by trinity_hip.hip.  New code, no reference source.
#pragma once
#include "codec_streams.hpp"

// ------------------------------------------------------------------------------------------ k_decode_terms
struct DecodeJob {
        uint32_t term;
        uint32_t pad;
        uint64_t out_off;
};

// grid.x covers blocks of job blockIdx.y in chunks of 256; one lane per block (google_codec.cpp:596-639)
template <int CODEC>
__global__ __launch_bounds__(256) void k_decode_terms(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last,
                                                      const uint32_t *__restrict__ blk_off, const DevTerm *__restrict__ terms,
                                                      const DecodeJob *__restrict__ jobs, uint32_t *__restrict__ docs,
                                                      uint32_t *__restrict__ freqs) {
        const DecodeJob job = jobs[blockIdx.y];
        const DevTerm t = terms[job.term];
        for (uint32_t b = blockIdx.x * 256 + threadIdx.x; b < t.nblocks; b += gridDim.x * 256) {
                const uint32_t gb = t.first_block + b;
                const uint32_t off = blk_off[gb];
                const uint32_t n = TRI_BLOCK_N(t, b, index, off);
                const uint32_t last = blk_last[gb];
                uint32_t doc = b ? blk_last[gb - 1] : 0;
                DeltaStream<CODEC> s;
                s.init(index, t, b, off);
                uint32_t *od = docs + job.out_off + (uint64_t)b * 32;
                for (uint32_t i = 0; i + 1 < n; ++i) {
                        doc += s.next();
                        od[i] = doc;
                }
                od[n - 1] = last;
                if (freqs) {
                        uint32_t *of = freqs + job.out_off + (uint64_t)b * 32;
                        FreqStream<CODEC> fs;
                        fs.init(index, t, b, off, s);
                        for (uint32_t i = 0; i < n; ++i)
                                of[i] = fs.next();
                }
        }
}

