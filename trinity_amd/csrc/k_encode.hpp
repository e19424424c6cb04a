// k_encode.hpp — the write side of the GOOGLE codec on the device
// Part of libtrinity_hip.so (MI355X / gfx950); included by trinity_hip.hip.  New code, no reference source.
#pragma once
#include "dev_structs.hpp"

// Codecs::Google::Encoder (google_codec.cpp:9-176) turns a term's postings — ascending docIDs, per document the counted hits'
// positions — into a chunk:  [u16 skiplist entries] blocks... [skiplist entries {u32 previous block's last docID, u32 offset of
// the block in the chunk}].  A block of n <= 32 documents is  varint(last docID - previous block's last) varint(length of what
// follows the n byte) u8(n)  n-1 delta varints  n freq varints  the hits (per document: varint((position delta) << 1 | length
// changes) [u8 payload length] payload bytes each).  Every 8th block COUNTED ACROSS TERMS leaves a skiplist entry in its term (at most 65535 per term).
//
// Here one lane owns one block: a sizing pass, a scan, a writing pass.  The blocks of all terms form one array; term t's blocks
// are [blk_first[t], blk_first[t + 1]).  Nothing is sequential but the scans.

__device__ __forceinline__ uint32_t enc_vlen(const uint32_t v) { return v < (1u << 7) ? 1u : v < (1u << 14) ? 2u : v < (1u << 21) ? 3u : v < (1u << 28) ? 4u : 5u; }

// prefix varint (Switch/switch_compiler_aux.h:23-51): the length sits in the leading ones of the first byte
__device__ __forceinline__ uint8_t *enc_put(uint8_t *o, const uint32_t v) {
        if (v < (1u << 7))
                *o++ = (uint8_t)v;
        else if (v < (1u << 14)) {
                *o++ = (uint8_t)(0x80u | (v >> 8));
                *o++ = (uint8_t)v;
        } else if (v < (1u << 21)) {
                *o++ = (uint8_t)(0xc0u | (v >> 16));
                *o++ = (uint8_t)v;
                *o++ = (uint8_t)(v >> 8);
        } else if (v < (1u << 28)) {
                *o++ = (uint8_t)(0xe0u | (v >> 24));
                *o++ = (uint8_t)(v >> 16);
                *o++ = (uint8_t)(v >> 8);
                *o++ = (uint8_t)v;
        } else {
                *o++ = 0xf0u;
                *o++ = (uint8_t)v;
                *o++ = (uint8_t)(v >> 8);
                *o++ = (uint8_t)(v >> 16);
                *o++ = (uint8_t)(v >> 24);
        }
        return o;
}

// Exclusive prefix sums of n u32 values, out[n] = the total — over the whole device in three launches: every workgroup adds up its chunk
// of ENC_SCAN_CHUNK values (k_enc_scan_sums), ONE workgroup turns the chunk sums into chunk bases (k_enc_scan: the single-workgroup scan,
// also what short arrays use on their own), every workgroup scans its chunk again on top of its base (k_enc_scan_chunks).
constexpr uint32_t ENC_SCAN_CHUNK = 1024 * 32;
__device__ __forceinline__ uint64_t enc_wave_incl(uint64_t x, const uint32_t lane) {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
                const uint64_t y = __shfl_up(x, d, 64);
                if ((int)lane >= d)
                        x += y;
        }
        return x;
}
// the values [at0, at1) scanned by the calling workgroup (1024 threads) on top of `base`; returns (in every thread) base + their sum
// (in and out may be the same array: a round reads its 1024 values before the barrier and writes them after)
template <typename IN>
__device__ __forceinline__ uint64_t enc_scan_range(const IN *in, uint64_t *out, const uint64_t at0, const uint64_t at1, uint64_t base,
                                                   uint64_t (&wsum)[16], uint64_t &base_s) {
        const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
        if (tid == 0)
                base_s = base;
        __syncthreads();
        for (uint64_t at = at0; at < at1; at += 1024) {
                const uint64_t i = at + tid;
                const uint64_t v = i < at1 ? (uint64_t)in[i] : 0;
                const uint64_t x = enc_wave_incl(v, lane);
                if (lane == 63)
                        wsum[wave] = x;
                __syncthreads();
                uint64_t before = base_s;
                for (uint32_t w = 0; w < wave; ++w)
                        before += wsum[w];
                if (i < at1 && out)
                        out[i] = before + x - v;
                __syncthreads();
                if (tid == 1023)
                        base_s = before + x;
                __syncthreads();
        }
        return base_s;
}
__global__ __launch_bounds__(1024) void k_enc_scan(const uint32_t *__restrict__ in, uint64_t *__restrict__ out, const uint64_t n) {
        __shared__ uint64_t wsum[16];
        __shared__ uint64_t base_s;
        const uint64_t total = enc_scan_range(in, out, 0, n, 0, wsum, base_s);
        if (threadIdx.x == 0)
                out[n] = total;
}
// sums[c] = the sum of chunk c
__global__ __launch_bounds__(1024) void k_enc_scan_sums(const uint32_t *__restrict__ in, uint64_t *__restrict__ sums, const uint64_t n) {
        __shared__ uint64_t wsum[16];
        __shared__ uint64_t base_s;
        const uint64_t at0 = (uint64_t)blockIdx.x * ENC_SCAN_CHUNK, at1 = at0 + ENC_SCAN_CHUNK < n ? at0 + ENC_SCAN_CHUNK : n;
        const uint64_t total = enc_scan_range(in, (uint64_t *)nullptr, at0, at1, 0, wsum, base_s);
        if (threadIdx.x == 0)
                sums[blockIdx.x] = total;
}
// sums[] in place: chunk sums -> chunk bases (exclusive), sums[nchunks] = the total
__global__ __launch_bounds__(1024) void k_enc_scan_bases(uint64_t *__restrict__ sums, const uint64_t nchunks) {
        __shared__ uint64_t wsum[16];
        __shared__ uint64_t base_s;
        const uint64_t total = enc_scan_range(sums, sums, 0, nchunks, 0, wsum, base_s);
        if (threadIdx.x == 0)
                sums[nchunks] = total;
}
__global__ __launch_bounds__(1024) void k_enc_scan_chunks(const uint32_t *__restrict__ in, const uint64_t *__restrict__ bases, uint64_t *__restrict__ out, const uint64_t n) {
        __shared__ uint64_t wsum[16];
        __shared__ uint64_t base_s;
        const uint64_t at0 = (uint64_t)blockIdx.x * ENC_SCAN_CHUNK, at1 = at0 + ENC_SCAN_CHUNK < n ? at0 + ENC_SCAN_CHUNK : n;
        const uint64_t total = enc_scan_range(in, out, at0, at1, bases[blockIdx.x], wsum, base_s);
        if (threadIdx.x == 0 && at1 == n)
                out[n] = total;
}

struct EncArgs {
        const uint32_t *docs, *freqs;
        const uint16_t *positions;
        const uint8_t *plens;       // per hit: its payload's length (0 .. 8); nullptr: no hit has a payload
        const uint64_t *payloads;   // per hit: the payload, its first byte in the low 8 bits
        const uint64_t *hit_off;    // [postings + 1]: hits before posting p
        const uint64_t *term_first; // [nterms + 1]: postings before term t
        const uint32_t *blk_first;  // [nterms + 1]: blocks before term t
        const uint32_t *blk_term;   // [nblocks]
        uint32_t nblocks;
};

// the block's place in its term and what the sizing and the writing pass both need of it
struct EncBlock {
        uint64_t p0;
        uint32_t n, prev_last, last;
};
__device__ __forceinline__ EncBlock enc_block(const EncArgs &a, const uint32_t g) {
        const uint32_t t = a.blk_term[g], j = g - a.blk_first[t];
        EncBlock b;
        b.p0 = a.term_first[t] + 32ull * j;
        const uint64_t left = a.term_first[t + 1] - b.p0;
        b.n = left < 32 ? (uint32_t)left : 32u;
        b.prev_last = j ? a.docs[b.p0 - 1] : 0u;
        b.last = a.docs[b.p0 + b.n - 1];
        return b;
}

// sizes[g] = bytes of block g (header included); tails[g] = bytes after the n byte (what the header's second varint says)
__global__ void k_enc_size(const EncArgs a, uint32_t *__restrict__ sizes, uint32_t *__restrict__ tails) {
        const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
        if (g >= a.nblocks)
                return;
        const EncBlock b = enc_block(a, g);
        uint32_t body = 0, prev = b.prev_last;
        for (uint32_t i = 0; i < b.n; ++i) {
                const uint32_t d = a.docs[b.p0 + i];
                if (i + 1 < b.n)
                        body += enc_vlen(d - prev);
                prev = d;
                body += enc_vlen(a.freqs[b.p0 + i]);
                // a hit: varint(position delta << 1 | the payload length changes) [u8 new length] payload bytes; the length state restarts
                // with every document (Encoder::new_hit, google_codec.cpp:38-74; begin_document :34)
                uint32_t last_pos = 0, cur_plen = 0;
                for (uint64_t h = a.hit_off[b.p0 + i]; h < a.hit_off[b.p0 + i + 1]; ++h) {
                        const uint32_t pos = a.positions[h], plen = a.plens ? a.plens[h] : 0u, chg = plen != cur_plen ? 1u : 0u;
                        body += enc_vlen((pos - last_pos) << 1 | chg) + chg + plen;
                        cur_plen = plen;
                        last_pos = pos;
                }
        }
        tails[g] = body;
        sizes[g] = enc_vlen(b.last - b.prev_last) + enc_vlen(body) + 1u + body;
}

// blk_off[g]: bytes of all blocks before g (over all terms); term_off[t]: where term t's chunk starts in out[]
__global__ void k_enc_write(const EncArgs a, const uint64_t *__restrict__ blk_off, const uint32_t *__restrict__ tails, const uint64_t *__restrict__ term_off,
                            uint8_t *__restrict__ out) {
        const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
        if (g >= a.nblocks)
                return;
        const EncBlock b = enc_block(a, g);
        const uint32_t t = a.blk_term[g], g0 = a.blk_first[t], g1 = a.blk_first[t + 1];
        const uint64_t chunk = term_off[t];
        const uint64_t in_chunk = 2 + (blk_off[g] - blk_off[g0]); // where the block starts inside its chunk
        uint8_t *o = out + chunk + in_chunk;
        o = enc_put(o, b.last - b.prev_last);
        o = enc_put(o, tails[g]);
        *o++ = (uint8_t)b.n;
        uint32_t prev = b.prev_last;
        for (uint32_t i = 0; i + 1 < b.n; ++i) {
                const uint32_t d = a.docs[b.p0 + i];
                o = enc_put(o, d - prev);
                prev = d;
        }
        for (uint32_t i = 0; i < b.n; ++i)
                o = enc_put(o, a.freqs[b.p0 + i]);
        for (uint32_t i = 0; i < b.n; ++i) {
                uint32_t last_pos = 0, cur_plen = 0;
                for (uint64_t h = a.hit_off[b.p0 + i]; h < a.hit_off[b.p0 + i + 1]; ++h) {
                        const uint32_t pos = a.positions[h], plen = a.plens ? a.plens[h] : 0u, chg = plen != cur_plen ? 1u : 0u;
                        o = enc_put(o, (pos - last_pos) << 1 | chg);
                        if (chg)
                                *o++ = (uint8_t)plen;
                        if (plen) {
                                const uint64_t pl = a.payloads[h];
                                for (uint32_t k = 0; k < plen; ++k)
                                        *o++ = (uint8_t)(pl >> (8 * k));
                        }
                        cur_plen = plen;
                        last_pos = pos;
                }
        }
        // the skiplist: every 8th block counted ACROSS terms (the encoder's countdown survives end_term, google_codec.h:57) leaves
        // {previous block's last docID, the block's offset in the chunk} in its term's list, the first 65535 of a term only
        const uint32_t first_marked = (g0 + 8) / 8 * 8 - 1; // the term's first block g' (>= g0) with (g' + 1) % 8 == 0
        if (((g + 1) & 7u) == 0) {
                const uint32_t idx = (g - first_marked) / 8;
                if (idx < 65535u) {
                        uint8_t *s = out + chunk + 2 + (blk_off[g1] - blk_off[g0]) + 8ull * idx;
                        const uint32_t rec[2] = {b.prev_last, (uint32_t)in_chunk};
                        for (int k = 0; k < 8; ++k)
                                s[k] = (uint8_t)(rec[k >> 2] >> ((k & 3) * 8));
                }
        }
        if (g == g0) { // the chunk's first two bytes: its skiplist entries
                const uint32_t last_g = g1 - 1;
                uint32_t entries = last_g >= first_marked ? (last_g - first_marked) / 8 + 1 : 0u;
                entries = entries < 65535u ? entries : 65535u;
                out[chunk] = (uint8_t)entries;
                out[chunk + 1] = (uint8_t)(entries >> 8);
        }
}
