// index_host.hpp — the host side of tri_index_upload: one validating walk over every term chunk of a segment (the GPU analogue of
// Codecs::Google::Decoder::init, google_codec.cpp:936-983, and Lucene::Decoder::init, lucene_codec.cpp:896-932, done once for all
// terms) that yields the dense block directory, the delta streams, the row records, the docID-cell index and the SURVEY §8(d) byte
// split.  Host-only C++17 (no HIP): trinity_hip.hip uploads what it builds; tools/plan_probe.cpp and the CPU tests of the planner
// build the same structures without a device.  New code, no reference source.
#pragma once
#include "../../include/trinity_hip.h"
#include "dev_structs.hpp"
#include "fastpfor128.hpp"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <sys/types.h>
#include <vector>

// error text of the host-only layers: formatted into `err`, the code handed back (the C-ABI wrappers pass it to tri_last_error)
inline int herr(std::string &err, int code, const char *fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
}

struct RowRec { // == uint4 on the device (DevTerm rows of a LUCENE segment, tri_index::d_blk_rec)
        uint32_t x, y, z, w;
};
static_assert(sizeof(RowRec) == 16, "row records are read as uint4");

// Everything tri_index_upload derives from the segment bytes.  The planner needs terms / blk_last / docbytes / hitbytes / the df order;
// the rest is copied to the device and dropped (release_device_columns).
struct HostIndex {
        int codec = TRI_CODEC_GOOGLE;
        std::vector<DevTerm> terms;
        std::vector<tri_term> tctx;
        std::vector<uint64_t> docbytes, hitbytes;
        std::vector<uint32_t> blk_last, blk_off, blk_hits, hdir, blk_doff, win;
        std::vector<RowRec> blk_rec;
        std::vector<uint8_t> dstream;
        uint32_t max_doc = 0, nwin = 0;
        bool has_hdir = false; // LUCENE uploaded with hits.data: phrases and the default mode can address positions
        tri_index_info info{};
        // terms by descending document count (ties: term id): df_rank[term] = position in that order, df_sorted[rank] = documents.
        // "A term gets a plane when it holds at least N documents" is then rank < (number of terms with >= N documents) — one binary
        // search per batch instead of a hash map per query term (planner.hpp)
        std::vector<uint32_t> df_rank, df_sorted;
        // LUCENE segments whose ints() groups carry FastPFor<4> words (the reference's own build, fastpfor128.hpp): the image the DEVICE gets is
        // this transcoded copy — every group re-encoded as PFOR128, chunk offsets rewritten — instead of the caller's bytes (empty: the caller's
        // bytes are what the kernels read).  tctx keeps the CALLER's term table; dev_tctx is the transcoded image's
        std::vector<uint8_t> dev_index, dev_hits;
        uint64_t transcoded_groups = 0;
        void release_device_columns() {
                for (auto *v : {&blk_off, &blk_hits, &hdir, &blk_doff, &win})
                        std::vector<uint32_t>().swap(*v);
                std::vector<RowRec>().swap(blk_rec);
                std::vector<uint8_t>().swap(dstream);
                std::vector<uint8_t>().swap(dev_index);
                std::vector<uint8_t>().swap(dev_hits);
        }
};

namespace trih {
        // host-side prefix varint (Switch/switch_compiler_aux.h:53-80) — used only by the upload-time walk
        inline size_t h_vb_get(const uint8_t *ip, uint32_t &v) {
                const uint32_t x = ip[0];
                if (!(x & 0x80u)) {
                        v = x;
                        return 1;
                } else if (!(x & 0x40u)) {
                        v = ((x & 0x3fu) << 8) | ip[1];
                        return 2;
                } else if (!(x & 0x20u)) {
                        v = ((x & 0x1fu) << 16) | ip[1] | ((uint32_t)ip[2] << 8);
                        return 3;
                } else if (!(x & 0x10u)) {
                        v = ((x & 0x0fu) << 24) | ((uint32_t)ip[1] << 16) | ((uint32_t)ip[2] << 8) | ip[3];
                        return 4;
                }
                v = ip[1] | ((uint32_t)ip[2] << 8) | ((uint32_t)ip[3] << 16) | ((uint32_t)ip[4] << 24);
                return 5;
        }
        inline size_t h_vb_len(uint8_t b0) { return b0 < 0x80 ? 1 : b0 < 0xc0 ? 2 : b0 < 0xe0 ? 3 : b0 < 0xf0 ? 4 : 5; }

        // ints() group of 128 values (lucene_codec.cpp:69-100 framing; PFOR128 payload, include/pfor128.md).  Returns the
        // bytes consumed, 0 when malformed.  Upload-time only: it yields the per-32-document directory rows.
        inline size_t h_ints_decode(const uint8_t *p, const uint8_t *end, uint32_t *v) {
                if (p >= end)
                        return 0;
                const uint32_t L = p[0];
                if (!L) {
                        if (p + 1 >= end || p + 1 + h_vb_len(p[1]) > end)
                                return 0;
                        uint32_t x;
                        const size_t n = h_vb_get(p + 1, x);
                        for (int i = 0; i < 128; ++i)
                                v[i] = x;
                        return 1 + n;
                }
                if (p + 1 + 4 * (size_t)L > end)
                        return 0;
                std::vector<uint32_t> w(L + 2, 0);
                memcpy(w.data(), p + 1, (size_t)L * 4);
                const uint32_t b = w[0] & 0xff, nexc = (w[0] >> 8) & 0xff, eb = (w[0] >> 16) & 0xff;
                if (b > 32 || eb > 32 || 1 + 4 * b + (nexc + 3) / 4 + (nexc * eb + 31) / 32 != L)
                        return 0;
                const uint32_t *packed = w.data() + 1, *epos = packed + 4 * b, *ehigh = epos + (nexc + 3) / 4;
                for (uint32_t i = 0; i < 128; ++i) {
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
                        if (pos >= 128 || b >= 32)
                                return 0;
                        v[pos] |= (uint32_t)((win >> (bit & 31)) & (eb == 32 ? 0xffffffffull : ((1ull << eb) - 1))) << b;
                }
                return 1 + (size_t)L * 4;
        }
        inline size_t h_ints_skip(const uint8_t *p, const uint8_t *end) {
                if (p >= end)
                        return 0;
                if (!p[0] && p + 1 >= end)
                        return 0;
                const size_t n = p[0] ? 1 + 4 * (size_t)p[0] : 1 + h_vb_len(p[1]);
                return p + n <= end ? n : 0;
        }
        // Per quarter (32 values) of a VALIDATED ints() group: where its exceptions start in the group's list and how many it has,
        // packed e0 | cnt << 8.  (The positions are ascending, so a quarter's exceptions are one run of the list.)
        inline bool h_ints_exc(const uint8_t *p, uint32_t out[4]) {
                out[0] = out[1] = out[2] = out[3] = 0;
                const uint32_t L = p[0];
                if (!L)
                        return true;
                uint32_t w0;
                memcpy(&w0, p + 1, 4);
                const uint32_t b = w0 & 0xff, nexc = (w0 >> 8) & 0xff;
                const uint8_t *epos = p + 5 + 16 * (size_t)b;
                uint32_t cnt[4] = {0, 0, 0, 0}, e0[4] = {0, 0, 0, 0};
                for (uint32_t e = 0; e < nexc; ++e) {
                        const uint32_t q = epos[e] >> 5;
                        if (q > 3 || (e && epos[e] <= epos[e - 1]))
                                return false;
                        if (!cnt[q])
                                e0[q] = e;
                        ++cnt[q];
                }
                for (int q = 0; q < 4; ++q)
                        out[q] = e0[q] | cnt[q] << 8;
                return true;
        }
} // namespace trih

// A LUCENE segment with FastPFor<4> payload words -> the same segment with PFOR128 ones (fastpfor128.hpp), chunk by chunk: 14-byte term
// header, 128-document blocks as two ints() groups, the varbyte tail (lucene_codec.cpp:163-388); the chunk's trailing skiplist is dropped
// (the engine builds its own directory; skiplistSize 0), the term's positions chunk in hits.data — blocks of 128 hits as two ints() groups +
// payload bytes, then the tail (lucene_codec.cpp:245-307, 339-366) — likewise.  Returns TRI_OK with out_index empty when no group of the
// segment is FastPFor-flavoured (nothing to do).
inline int lucene_transcode(const uint8_t *index, size_t len, const uint8_t *hits, size_t hits_len, const tri_term *terms, size_t nterms,
                            std::vector<uint8_t> &out_index, std::vector<uint8_t> &out_hits, std::vector<tri_term> &out_terms, uint64_t &ngroups, std::string &err) {
        using namespace trih;
        // ---- is there anything to do?  (the first ints() group of the first term that has a full block tells: a segment is written by one build)
        bool any = false;
        for (size_t ti = 0; ti < nterms && !any; ++ti) {
                const tri_term &t = terms[ti];
                if (t.documents < 128 || t.size < 14 + 6 || (uint64_t)t.offset + t.size > len)
                        continue;
                const uint8_t *g = index + t.offset + 14;
                if (g[0] == 0)
                        continue; // (an all-equal group: look further)
                any = trif::group_is_fastpfor(g);
                break;
        }
        out_index.clear();
        out_hits.clear();
        if (!any)
                return TRI_OK;
        out_terms.assign(terms, terms + nterms);
        out_index.reserve(len + len / 8);
        out_hits.reserve(hits_len + hits_len / 8);
        ngroups = 0;
        // one ints() group at p -> re-encoded at the end of `out`; returns the bytes consumed (0: malformed)
        auto group = [&](const uint8_t *p, const uint8_t *end, std::vector<uint8_t> &out) -> size_t {
                if (p >= end)
                        return 0;
                const uint32_t L = p[0];
                if (!L) {
                        if (p + 1 >= end)
                                return 0;
                        const size_t n = 1 + h_vb_len(p[1]);
                        if (p + n > end)
                                return 0;
                        out.insert(out.end(), p, p + n);
                        return n;
                }
                if (p + 1 + 4 * (size_t)L > end)
                        return 0;
                if (!trif::group_is_fastpfor(p)) { // (already PFOR128: kept as it is — h_ints_decode validates it in the walk)
                        out.insert(out.end(), p, p + 1 + 4 * (size_t)L);
                        return 1 + 4 * (size_t)L;
                }
                std::vector<uint32_t> w(L);
                memcpy(w.data(), p + 1, 4 * (size_t)L);
                uint32_t v[128];
                if (!trif::fastpfor_decode(w.data(), L, v))
                        return 0;
                bool eq = true;
                for (uint32_t i = 1; i < 128; ++i)
                        eq &= v[i] == v[0];
                if (eq) { // (FastPFor never sees such a group — lucene_codec.cpp:31-39 short-cuts it — but a foreign writer might)
                        out.push_back(0);
                        uint8_t tmp[5];
                        size_t n = 0;
                        const uint32_t x = v[0];
                        if (x < (1u << 7))
                                tmp[n++] = (uint8_t)x;
                        else if (x < (1u << 14))
                                tmp[n++] = (uint8_t)(0x80u | (x >> 8)), tmp[n++] = (uint8_t)x;
                        else if (x < (1u << 21))
                                tmp[n++] = (uint8_t)(0xc0u | (x >> 16)), tmp[n++] = (uint8_t)x, tmp[n++] = (uint8_t)(x >> 8);
                        else if (x < (1u << 28))
                                tmp[n++] = (uint8_t)(0xe0u | (x >> 24)), tmp[n++] = (uint8_t)(x >> 16), tmp[n++] = (uint8_t)(x >> 8), tmp[n++] = (uint8_t)x;
                        else
                                tmp[n++] = 0xf0u, tmp[n++] = (uint8_t)x, tmp[n++] = (uint8_t)(x >> 8), tmp[n++] = (uint8_t)(x >> 16), tmp[n++] = (uint8_t)(x >> 24);
                        out.insert(out.end(), tmp, tmp + n);
                } else
                        trif::pfor128_encode(v, out);
                ++ngroups;
                return 1 + 4 * (size_t)L;
        };
        for (size_t ti = 0; ti < nterms; ++ti) {
                const tri_term &t = terms[ti];
                tri_term &o = out_terms[ti];
                if (!t.size || !t.documents)
                        continue;
                if ((uint64_t)t.offset + t.size > len || t.size < 14)
                        return herr(err, TRI_ERR_FORMAT, "term %zu: chunk [%u,+%u) outside index (%zu)", ti, t.offset, t.size, len);
                const uint8_t *base = index + t.offset, *p = base + 14;
                uint32_t hitsOff, sumHits, posChunk;
                uint16_t sk;
                memcpy(&hitsOff, base, 4);
                memcpy(&sumHits, base + 4, 4);
                memcpy(&posChunk, base + 8, 4);
                memcpy(&sk, base + 12, 2);
                if (14 + (size_t)sk * 22 > t.size)
                        return herr(err, TRI_ERR_FORMAT, "term %zu: skiplist larger than chunk", ti);
                const uint8_t *end = base + t.size - (size_t)sk * 22;
                if (out_index.size() > 0xfffffff0ull - t.size)
                        return herr(err, TRI_ERR_UNSUPPORTED, "the transcoded index would exceed 4 GiB");
                o.offset = (uint32_t)out_index.size();
                out_index.insert(out_index.end(), base, base + 14); // (header: patched below)
                for (uint32_t left = t.documents; left >= 128; left -= 128)
                        for (int gi = 0; gi < 2; ++gi) {
                                const size_t used = group(p, end, out_index);
                                if (!used)
                                        return herr(err, TRI_ERR_FORMAT, "term %zu: an ints() group that is neither FastPFor<4> (csrc/fastpfor128.hpp) nor PFOR128 (include/pfor128.md)", ti);
                                p += used;
                        }
                out_index.insert(out_index.end(), p, end); // the varbyte (delta, freq) tail; the skiplist stays behind
                // ---- the term's positions chunk
                uint32_t new_hits_off = (uint32_t)out_hits.size(), new_pos_chunk = 0;
                if (hits_len) {
                        if ((uint64_t)hitsOff + posChunk > hits_len)
                                return herr(err, TRI_ERR_FORMAT, "term %zu: positions chunk [%u,+%u) outside hits.data (%zu)", ti, hitsOff, posChunk, hits_len);
                        const uint8_t *hp = hits + hitsOff, *hend = hp + posChunk;
                        for (uint32_t hb = 0; hb < sumHits / 128; ++hb) {
                                for (int gi = 0; gi < 2; ++gi) {
                                        const size_t used = group(hp, hend, out_hits);
                                        if (!used)
                                                return herr(err, TRI_ERR_FORMAT, "term %zu: bad hits block %u", ti, hb);
                                        hp += used;
                                }
                                if (hp >= hend || hp + h_vb_len(*hp) > hend)
                                        return herr(err, TRI_ERR_FORMAT, "term %zu: hits block %u is truncated", ti, hb);
                                uint32_t payloadBytes;
                                const size_t n = h_vb_get(hp, payloadBytes);
                                if (hp + n + payloadBytes > hend)
                                        return herr(err, TRI_ERR_FORMAT, "term %zu: hits block %u payload overruns the chunk", ti, hb);
                                out_hits.insert(out_hits.end(), hp, hp + n + payloadBytes);
                                hp += n + payloadBytes;
                        }
                        out_hits.insert(out_hits.end(), hp, hend); // the tail hits
                        new_pos_chunk = (uint32_t)(out_hits.size() - new_hits_off);
                }
                const uint16_t nosk = 0;
                memcpy(out_index.data() + o.offset, &new_hits_off, 4);
                memcpy(out_index.data() + o.offset + 8, &new_pos_chunk, 4);
                memcpy(out_index.data() + o.offset + 12, &nosk, 2);
                o.size = (uint32_t)(out_index.size() - o.offset);
        }
        return TRI_OK;
}

// The walk.  `index` / `hits`: the segment's raw codec bytes; `terms[i]` what IndexSource::resolve_term_ctx returns for term i.
inline int build_host_index(const uint8_t *index, size_t len, const uint8_t *hits, size_t hits_len, int codec, const tri_term *terms, size_t nterms,
                            uint32_t docs_cnt, HostIndex &H, std::string &err) {
        using namespace trih;
        if (codec != TRI_CODEC_GOOGLE && codec != TRI_CODEC_LUCENE)
                return herr(err, TRI_ERR_INVALID, "tri_index_upload: unknown codec %d", codec);
        if (len > 0xffffffffull)
                return herr(err, TRI_ERR_FORMAT, "index exceeds 32-bit chunk offsets (codecs.h:26)");
        H = HostIndex{};
        H.codec = codec;
        H.terms.resize(nterms);
        H.tctx.assign(terms, terms + nterms);
        H.docbytes.assign(nterms, 0);
        H.hitbytes.assign(nterms, 0);
        // a LUCENE segment as the reference's own build writes it (FastPFor<4> payload words): the walk below — and the device — see its
        // PFOR128 transcription; the SURVEY §8(d) byte counts are taken from the bytes the caller handed over (what the reference would stream)
        std::vector<tri_term> dev_terms;
        std::vector<uint64_t> orig_docbytes, orig_hitbytes;
        if (codec == TRI_CODEC_LUCENE) {
                if (const int rc = lucene_transcode(index, len, hits, hits_len, terms, nterms, H.dev_index, H.dev_hits, dev_terms, H.transcoded_groups, err))
                        return rc;
                if (!H.dev_index.empty()) {
                        orig_docbytes.assign(nterms, 0);
                        orig_hitbytes.assign(nterms, 0);
                        for (size_t ti = 0; ti < nterms; ++ti)
                                if (terms[ti].size >= 14 && terms[ti].documents) {
                                        uint16_t sk;
                                        uint32_t pc;
                                        memcpy(&sk, index + terms[ti].offset + 12, 2);
                                        memcpy(&pc, index + terms[ti].offset + 8, 4);
                                        orig_docbytes[ti] = terms[ti].size - (uint64_t)sk * 22;
                                        orig_hitbytes[ti] = pc;
                                }
                        H.dev_index.resize(H.dev_index.size() + 16, 0); // (slack for the walk's bounded reads; dropped again below)
                        index = H.dev_index.data();
                        len = H.dev_index.size() - 16;
                        hits = H.dev_hits.empty() ? nullptr : H.dev_hits.data();
                        hits_len = H.dev_hits.size();
                        terms = dev_terms.data();
                }
        }
        std::vector<uint32_t> &blk_last = H.blk_last, &blk_off = H.blk_off;
        std::vector<uint32_t> &blk_hits = H.blk_hits, &hdir = H.hdir; // (hdir: LUCENE + hits.data only)
        std::vector<RowRec> &blk_rec = H.blk_rec;                     // LUCENE only
        std::vector<uint8_t> &dstream = H.dstream;                    // GOOGLE only
        std::vector<uint32_t> &blk_doff = H.blk_doff;
        if (codec == TRI_CODEC_GOOGLE) {
                dstream.reserve(len / 3 + 64);
                blk_doff.reserve(len / 96 + nterms);
        }
        const bool want_hits = codec == TRI_CODEC_LUCENE && hits_len;
        blk_last.reserve(len / 96 + nterms);
        blk_off.reserve(len / 96 + nterms);
        uint64_t postings = 0, docb = 0, hitb = 0;
        // One pass over every chunk: hop block headers (google_codec.cpp:641-697), validate, record the directory
        // and the algorithmic byte split of SURVEY §8(d).
        for (size_t ti = 0; ti < nterms; ++ti) {
                const tri_term &t = terms[ti];
                DevTerm &dt = H.terms[ti];
                dt.documents = t.documents;
                dt.first_block = (uint32_t)blk_last.size();
                dt.nblocks = 0;
                dt.last_n = 0;
                dt.flags = 0;
                dt.npfor = 0;
                dt.pad = 0;
                if (!t.size || !t.documents) {
                        dt.documents = 0;
                        continue;
                }
                if (codec == TRI_CODEC_LUCENE) {
                        // Lucene-shaped chunk (lucene_codec.cpp:163-388): 14-byte header, full 128-document blocks as two ints()
                        // groups, varbyte (delta, freq) tail, 22-byte skiplist entries.  One directory row per 32 documents.
                        if ((uint64_t)t.offset + t.size > len || t.size < 14)
                                return herr(err, TRI_ERR_FORMAT, "term %zu: chunk [%u,+%u) outside index (%zu)", ti, t.offset, t.size, len);
                        const uint8_t *base = index + t.offset, *p = base + 14;
                        uint32_t posChunk, hitsOff, sumHits;
                        uint16_t sk;
                        memcpy(&hitsOff, base, 4);
                        memcpy(&sumHits, base + 4, 4);
                        memcpy(&posChunk, base + 8, 4);
                        memcpy(&sk, base + 12, 2);
                        if (14 + (size_t)sk * 22 > t.size)
                                return herr(err, TRI_ERR_FORMAT, "term %zu: skiplist larger than chunk", ti);
                        const uint8_t *end = base + t.size - (size_t)sk * 22;
                        uint32_t left = t.documents, doc = 0;
                        uint32_t vals[128], fvals[128];
                        uint64_t hits_seen = 0;
                        while (left >= 128) {
                                if (p >= end)
                                        return herr(err, TRI_ERR_FORMAT, "term %zu: truncated block", ti);
                                const uint32_t goff = (uint32_t)(p - index);
                                const size_t used = h_ints_decode(p, end, vals);
                                if (!used) // (the group's header word does not describe a PFOR128 payload of the declared length)
                                        return herr(err, TRI_ERR_FORMAT, "term %zu: an ints() group that is neither PFOR128 (include/pfor128.md) nor FastPFor<4> words as the reference's "
                                                                    "lucene_codec writes them (lucene_codec.cpp:57-64; csrc/fastpfor128.hpp transcribes those at upload)", ti);
                                p += used;
                                uint32_t xd[4], xf[4];
                                const size_t usedf = h_ints_decode(p, end, fvals);
                                if (!usedf || !h_ints_exc(index + goff, xd) || !h_ints_exc(p, xf))
                                        return herr(err, TRI_ERR_FORMAT, "term %zu: a freqs group / exception list that is neither PFOR128 (include/pfor128.md) nor FastPFor<4> words "
                                                                    "(csrc/fastpfor128.hpp)", ti);
                                p += usedf;
                                uint32_t hdr[2];
                                for (int gi = 0; gi < 2; ++gi) { // the two groups' header words as the row records cache them
                                        const uint8_t *gp = gi ? p - usedf : index + goff;
                                        if (gp[0])
                                                memcpy(&hdr[gi], gp + 1, 4);
                                        else {
                                                const uint32_t v = gi ? fvals[0] : vals[0];
                                                if (v >> 31)
                                                        return herr(err, TRI_ERR_UNSUPPORTED, "term %zu: an all-equal group of value %u", ti, v);
                                                hdr[gi] = 0x80000000u | v;
                                        }
                                }
                                for (uint32_t q4 = 0; q4 < 4; ++q4) {
                                        blk_rec.push_back(RowRec{goff, xd[q4] | xf[q4] << 16, hdr[0], hdr[1]});
                                        if (want_hits)
                                                blk_hits.push_back((uint32_t)hits_seen);
                                        for (uint32_t i = 0; i < 32; ++i) {
                                                if (!vals[q4 * 32 + i])
                                                        return herr(err, TRI_ERR_FORMAT, "term %zu: zero document delta", ti);
                                                doc += vals[q4 * 32 + i];
                                                if (want_hits)
                                                        hits_seen += fvals[q4 * 32 + i];
                                        }
                                        blk_last.push_back(doc);
                                        blk_off.push_back(goff);
                                        dt.nblocks++;
                                }
                                left -= 128;
                        }
                        dt.npfor = dt.nblocks;
                        dt.last_n = 32;
                        while (left) {
                                const uint32_t n = std::min(left, 32u);
                                blk_off.push_back((uint32_t)(p - index));
                                blk_rec.push_back(RowRec{(uint32_t)(p - index), 0, 0, 0});
                                if (want_hits)
                                        blk_hits.push_back((uint32_t)hits_seen);
                                for (uint32_t i = 0; i < n; ++i) {
                                        uint32_t d, f;
                                        if (p >= end || p + h_vb_len(*p) >= end || p + h_vb_len(*p) + h_vb_len(p[h_vb_len(*p)]) > end)
                                                return herr(err, TRI_ERR_FORMAT, "term %zu: truncated tail", ti);
                                        p += h_vb_get(p, d);
                                        p += h_vb_get(p, f);
                                        if (!d)
                                                return herr(err, TRI_ERR_FORMAT, "term %zu: zero document delta", ti);
                                        doc += d;
                                        hits_seen += f;
                                }
                                blk_last.push_back(doc);
                                dt.nblocks++;
                                dt.last_n = n;
                                left -= n;
                        }
                        if (p != end)
                                return herr(err, TRI_ERR_FORMAT, "term %zu: %zd stray bytes before the skiplist", ti, (ssize_t)(end - p));
                        dt.flags = TERM_FULL_BLOCKS;
                        if (want_hits) {
                                // hits.data of this term (lucene_codec.cpp:245-307, 339-352): sumHits / 128 full blocks
                                // { ints(posDeltas) ints(payloadLens) varbyte(payloadBytes) payload }, then the varbyte tail
                                if (hits_seen != sumHits)
                                        return herr(err, TRI_ERR_FORMAT, "term %zu: %llu hits by frequency, %u declared", ti, (unsigned long long)hits_seen, sumHits);
                                if ((uint64_t)hitsOff + posChunk > hits_len)
                                        return herr(err, TRI_ERR_FORMAT, "term %zu: positions chunk [%u,+%u) outside hits.data (%zu)", ti, hitsOff, posChunk, hits_len);
                                const uint8_t *hp = hits + hitsOff, *hend = hp + posChunk;
                                const uint32_t nfull = sumHits / 128;
                                dt.pad = (uint32_t)hdir.size();
                                hdir.push_back(nfull);
                                for (uint32_t hb = 0; hb < nfull; ++hb) {
                                        hdir.push_back((uint32_t)(hp - hits));
                                        for (int g = 0; g < 2; ++g) {
                                                const size_t used = h_ints_skip(hp, hend);
                                                if (!used || hp + used > hend)
                                                        return herr(err, TRI_ERR_FORMAT, "term %zu: bad hits block %u", ti, hb);
                                                hp += used;
                                        }
                                        uint32_t payloadBytes;
                                        hp += h_vb_get(hp, payloadBytes);
                                        if (hp + payloadBytes > hend)
                                                return herr(err, TRI_ERR_FORMAT, "term %zu: hits block %u payload overruns the chunk", ti, hb);
                                        hp += payloadBytes;
                                }
                                hdir.push_back((uint32_t)(hp - hits));
                        }
                        const uint64_t db = (uint64_t)(end - base); // 14-byte header + block bytes, no skiplist, no hits.data
                        H.docbytes[ti] = db;
                        H.hitbytes[ti] = posChunk;
                        postings += t.documents;
                        docb += db;
                        hitb += posChunk;
                        continue;
                }
                if ((uint64_t)t.offset + t.size > len || t.size < 2)
                        return herr(err, TRI_ERR_FORMAT, "term %zu: chunk [%u,+%u) outside index (%zu)", ti, t.offset, t.size, len);
                const uint8_t *base = index + t.offset, *p = base + 2, *end = base + t.size;
                uint16_t sk;
                memcpy(&sk, base, 2);
                if ((size_t)sk * 8 + 2 > t.size)
                        return herr(err, TRI_ERR_FORMAT, "term %zu: skiplist larger than chunk", ti);
                end -= (size_t)sk * 8;
                uint64_t db = 2, hb = 0;
                uint32_t lastDoc = 0, docs = 0;
                bool full_blocks = true;
                while (p != end) {
                        if (p + 3 > end)
                                return herr(err, TRI_ERR_FORMAT, "term %zu: truncated block header", ti);
                        const uint8_t *h = p;
                        uint32_t delta, blockLength;
                        // (every varint is bounded before it is read: a malformed or truncated chunk must end in TRI_ERR_FORMAT, not in a read past the buffer)
                        if (p + h_vb_len(*p) >= end)
                                return herr(err, TRI_ERR_FORMAT, "term %zu: truncated block header", ti);
                        p += h_vb_get(p, delta);
                        if (p + h_vb_len(*p) >= end)
                                return herr(err, TRI_ERR_FORMAT, "term %zu: truncated block header", ti);
                        p += h_vb_get(p, blockLength);
                        const uint32_t n = *p++;
                        if (n < 1 || n > 32 || !delta || (uint64_t)(end - p) < blockLength)
                                return herr(err, TRI_ERR_FORMAT, "term %zu: bad block header (n=%u, delta=%u, len=%u)", ti, n, delta, blockLength);
                        if ((uint64_t)lastDoc + delta > 0xffffffffull) // (a running sum past 2^32 would wrap: the directory's last documents must ascend)
                                return herr(err, TRI_ERR_FORMAT, "term %zu: the blocks' document deltas run past 2^32", ti);
                        lastDoc += delta;
                        const uint8_t *s = p, *const bend = p + blockLength;
                        // the interior deltas are READ, not only stepped over: every kernel places a block's documents inside (previous block's last, this
                        // block's last] — window bitmaps, plane rows sized by the segment's last documentID — and a chunk whose deltas say otherwise (a
                        // damaged file) must end here, not in a store outside a bitmap
                        uint64_t span = 0;
                        for (uint32_t i = 0; i + 1 < n; ++i) {
                                if (s >= bend || s + h_vb_len(*s) > bend)
                                        return herr(err, TRI_ERR_FORMAT, "term %zu: deltas overrun the block", ti);
                                uint32_t dv;
                                s += h_vb_get(s, dv);
                                if (!dv)
                                        return herr(err, TRI_ERR_FORMAT, "term %zu: a document repeats inside a block (delta 0)", ti);
                                span += dv;
                        }
                        if (span >= delta)
                                return herr(err, TRI_ERR_FORMAT, "term %zu: a block's documents run past its last document (%llu >= %u)", ti, (unsigned long long)span, delta);
                        if (dstream.size() + 256 > 0xffffffffull)
                                return herr(err, TRI_ERR_UNSUPPORTED, "delta stream exceeds 4 GiB");
                        dstream.push_back((uint8_t)n);
                        blk_doff.push_back((uint32_t)dstream.size());
                        dstream.insert(dstream.end(), p, s);
                        uint64_t nhits = 0;
                        for (uint32_t i = 0; i < n; ++i) {
                                if (s >= bend || s + h_vb_len(*s) > bend)
                                        return herr(err, TRI_ERR_FORMAT, "term %zu: deltas+freqs overrun the block", ti);
                                uint32_t f;
                                s += h_vb_get(s, f);
                                nhits += f;
                        }
                        // GOOGLE: byte offset of the block's first hit (k_phrase / k_rich start there).  Bit 31 (BLK_HITS_PLAIN): every hit of the
                        // block is ONE byte — a position delta < 64 without the new-payload-length flag (google_codec.cpp:38-74) —, so a document's
                        // hits start at the block's first hit + the frequencies before it and no hit has to be parsed to find them
                        uint32_t hits_at = (uint32_t)(s - p); // (relative to the block's payload: blk_off[] below)
                        if ((uint64_t)(bend - s) == nhits && !(hits_at >> 31)) {
                                bool plain = true;
                                for (const uint8_t *q = s; q < bend && plain; ++q)
                                        plain = !(*q & 0x81u);
                                if (plain)
                                        hits_at |= BLK_HITS_PLAIN;
                        }
                        blk_hits.push_back(hits_at);
                        db += (uint64_t)(s - h);
                        hb += blockLength - (uint64_t)(s - p);
                        blk_last.push_back(lastDoc);
                        blk_off.push_back((uint32_t)(p - index));
                        if (dt.nblocks && dt.last_n != 32)
                                full_blocks = false; // a short block that is not the last one
                        dt.nblocks++;
                        dt.last_n = n;
                        docs += n;
                        p += blockLength;
                }
                if (docs != t.documents)
                        return herr(err, TRI_ERR_FORMAT, "term %zu: %u documents in blocks, %u declared", ti, docs, t.documents);
                if (!full_blocks) // the reference encoder only ever leaves the LAST block short (google_codec.cpp:76-88); the kernels' tile and
                                  // output layouts (32 slots per non-final block) rely on it, so a foreign chunk that does not is refused here
                        return herr(err, TRI_ERR_UNSUPPORTED, "term %zu: a block other than the last holds fewer than 32 documents", ti);
                dt.flags = TERM_FULL_BLOCKS;
                if ((uint64_t)docs * 28 < lastDoc)
                        dt.flags |= TERM_SPARSE;
                H.docbytes[ti] = db;
                H.hitbytes[ti] = hb;
                postings += docs;
                docb += db;
                hitb += hb;
        }
        // docID-cell index of the longer lists: win[row + c] = first block whose last docID >= c * CELL_DOCS.  With 288 GB of HBM
        // a 4-byte entry per 1024 docIDs per indexed term is cheap (66 MB at the 10M-document config) and turns directory
        // searches into one load pair: TASK_DENSE reads the entries of its window's ends (every SPAN_BITS / CELL_DOCS-th), a
        // galloping candidate brackets its block to the handful of blocks that end inside its cell.
        const uint32_t max_doc = blk_last.empty() ? 0 : *std::max_element(blk_last.begin(), blk_last.end());
        H.nwin = (max_doc / SPAN_BITS + 2) * (SPAN_BITS / CELL_DOCS) + 1;
        H.max_doc = max_doc;
        std::vector<uint32_t> &win = H.win;
        for (size_t ti = 0; ti < nterms; ++ti) {
                DevTerm &dt = H.terms[ti];
                dt.win_off = 0xffffffffu;
                if (dt.nblocks < WIN_MIN_BLOCKS)
                        continue;
                dt.win_off = (uint32_t)win.size();
                const uint32_t *bl = &blk_last[dt.first_block];
                uint32_t b = 0;
                if ((uint64_t)win.size() + H.nwin > 0xfffffff0ull)
                        return herr(err, TRI_ERR_UNSUPPORTED, "cell index exceeds 2^32 entries");
                for (uint32_t w = 0; w < H.nwin; ++w) {
                        const uint64_t key = (uint64_t)w * CELL_DOCS;
                        while (b < dt.nblocks && bl[b] < key)
                                ++b;
                        win.push_back(b);
                }
        }
        H.has_hdir = want_hits;
        if (!orig_docbytes.empty()) { // (a transcoded segment: the roofline's byte counts are those of the caller's encoding)
                H.dev_index.resize(len);
                docb = hitb = 0;
                for (size_t ti = 0; ti < nterms; ++ti) {
                        H.docbytes[ti] = orig_docbytes[ti];
                        H.hitbytes[ti] = orig_hitbytes[ti];
                        docb += orig_docbytes[ti];
                        hitb += orig_hitbytes[ti];
                }
        }
        H.info.index_bytes = len;
        H.info.directory_bytes = blk_last.size() * 8 + nterms * sizeof(DevTerm) + win.size() * 4;
        H.info.blocks = blk_last.size();
        H.info.postings = postings;
        H.info.doc_bytes = docb;
        H.info.hit_bytes = hitb;
        H.info.nterms = (uint32_t)nterms;
        H.info.docs_cnt = docs_cnt;
        // the df order (planner: which terms are long enough for a plane)
        std::vector<uint32_t> by_df(nterms);
        std::iota(by_df.begin(), by_df.end(), 0u);
        std::sort(by_df.begin(), by_df.end(), [&](uint32_t a, uint32_t b) {
                return H.terms[a].documents != H.terms[b].documents ? H.terms[a].documents > H.terms[b].documents : a < b;
        });
        H.df_rank.resize(nterms);
        H.df_sorted.resize(nterms);
        for (size_t r = 0; r < nterms; ++r) {
                H.df_rank[by_df[r]] = (uint32_t)r;
                H.df_sorted[r] = H.terms[by_df[r]].documents;
        }
        return TRI_OK;
}
