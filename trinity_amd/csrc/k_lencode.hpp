// k_lencode.hpp — the write side of the Lucene-shaped codec on the device: Codecs::Lucene::Encoder (lucene_codec.cpp:163-388) with this
// repo's PFOR128 ints() payload, byte-identical to csrc/host/lucene_encoder.hpp.  Part of libtrinity_hip.so (MI355X / gfx950); included by trinity_hip.hip.
// Every kernel is one UNIT of lucene_enc_units.hpp per lane — the units are the algorithm (and are tested on the CPU in plain loops against the sequential
// encoder, tests/test_fastpfor.py); the device scans of k_encode.hpp place them.  New code, no reference source.
#pragma once
#include "lucene_enc_units.hpp"

__global__ void k_lenc_hdelta(const LencArgs a, uint32_t *__restrict__ hdelta, const uint64_t np) {
        const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (p < np)
                lenc_unit_hdelta(a, p, hdelta);
}
// per term: its full document blocks and full hit blocks
__global__ void k_lenc_term_counts(const uint64_t *__restrict__ term_first, const uint64_t *__restrict__ hit_off, const uint64_t nterms, uint32_t *__restrict__ dcnt,
                                   uint32_t *__restrict__ hcnt) {
        const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (t >= nterms)
                return;
        dcnt[t] = (uint32_t)((term_first[t + 1] - term_first[t]) / LENC_BLOCK);
        hcnt[t] = (uint32_t)((hit_off[term_first[t + 1]] - hit_off[term_first[t]]) / LENC_BLOCK);
}
__global__ void k_lenc_dblk_size(const LencArgs a, const uint64_t nd, uint32_t *__restrict__ dsize) {
        const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (g < nd)
                dsize[g] = lenc_unit_dblk_size(a, g);
}
__global__ void k_lenc_hblk_size(const LencArgs a, const uint64_t nh, uint32_t *__restrict__ hsize) {
        const uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (h < nh)
                hsize[h] = lenc_unit_hblk_size(a, h);
}
__global__ void k_lenc_tail_size(const LencArgs a, uint32_t *__restrict__ tail_docs, uint32_t *__restrict__ tail_hits) {
        const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (t < a.nterms)
                lenc_unit_tail_size(a, t, tail_docs + t, tail_hits + t);
}
// per term: the bytes of its index chunk and of its hits.data chunk
__global__ void k_lenc_term_sizes(const LencArgs a, const LencPlace pl, uint32_t *__restrict__ isize, uint32_t *__restrict__ hsize) {
        const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (t >= a.nterms)
                return;
        isize[t] = lenc_term_index_size(a, pl, t);
        hsize[t] = lenc_term_hits_size(a, pl, t);
}
__global__ void k_lenc_dblk_write(const LencArgs a, const LencPlace pl, const uint64_t nd, uint8_t *__restrict__ index_out) {
        const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (g < nd)
                lenc_unit_dblk_write(a, pl, g, index_out);
}
__global__ void k_lenc_hblk_write(const LencArgs a, const LencPlace pl, const uint64_t nh, uint8_t *__restrict__ hits_out) {
        const uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (h < nh)
                lenc_unit_hblk_write(a, pl, h, hits_out);
}
__global__ void k_lenc_term_write(const LencArgs a, const LencPlace pl, uint8_t *__restrict__ index_out, uint8_t *__restrict__ hits_out) {
        const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (t < a.nterms)
                lenc_unit_term_write(a, pl, t, index_out, hits_out);
}
