// Application code written against the reference's write side — SegmentIndexSession (indexer.h:19-230): begin / insert / commit — over the mirror
// (trinity_amd/csrc/host/trinity_gpu_write.hpp).  Built (and linked against libtrinity_hip.so) by tests/test_host_mirror.py on every machine; it needs a
// device to RUN: it then prints the committed segment's sizes and term table for a handful of documents fed out of order.
#include "../../trinity_amd/csrc/host/trinity_gpu_write.hpp"
#include "../../trinity_amd/csrc/host/google_encoder.hpp"
#include <algorithm>
#include <cstdio>
#include <map>
#include <tuple>

using namespace trinity_amd;

int main() {
        tri_dev *dev = nullptr;
        check(tri_dev_open(0, &dev));
        SegmentIndexSession sess(dev);
        const char *texts[][4] = {{"world", "of", "warcraft", "mists"}, {"hello", "world", "again", "world"}, {"of", "mice", "and", "men"}};
        const isrc_docid_t ids[] = {30, 10, 20}; // (not in document order: commit sorts)
        for (int d = 0; d < 3; ++d) {
                auto doc = sess.begin(ids[d]);
                for (tokenpos_t p = 0; p < 4; ++p) {
                        if (d == 1 && p == 3) {
                                const uint16_t weight = 7; // a hit with a payload (indexer.h:128)
                                doc.insert(doc.term_id(texts[d][p]), tokenpos_t(p + 1), weight);
                        } else
                                doc.insert(texts[d][p], tokenpos_t(p + 1));
                }
                if (d == 2)
                        doc.insert("mice", tokenpos_t(0)); // a hit at position 0 without a payload: counted by commit, never stored (google_codec.cpp:42-45; indexer.cpp:447)
                sess.insert(doc);
        }
        const committed_segment seg = sess.commit();
        printf("index %zu bytes, %zu terms, documents %llu, postings %llu, hits %llu\n", seg.index.size(), seg.terms.size(), (unsigned long long)seg.stats.docs_cnt,
               (unsigned long long)seg.stats.sum_terms_docs, (unsigned long long)seg.stats.sum_term_hits);
        for (const auto &t : seg.terms)
                printf("%s documents=%u chunk=[%u,+%u)\n", t.first.c_str(), t.second.documents, t.second.offset, t.second.size);
        // the same session through the HOST encoder (csrc/host/google_encoder.hpp, byte-identical to the reference's): commit's walk — terms by (id & 31, id),
        // a term's documents ascending (indexer.cpp:399-416) — must give the bytes the device committed
        {
                struct Hit {
                        tokenpos_t pos;
                        uint8_t len;
                        uint64_t payload;
                };
                std::map<std::tuple<uint32_t, uint32_t, uint32_t>, std::vector<Hit>> walk; // (id & 31, id, document) -> its stored hits
                SegmentIndexSession names(dev);                                            // (a second session assigns the same term ids in the same order)
                for (int d = 0; d < 3; ++d) {
                        auto doc = names.begin(ids[d]);
                        for (tokenpos_t p = 0; p < 4; ++p) {
                                const uint32_t id = doc.term_id(texts[d][p]);
                                const uint16_t weight = 7;
                                Hit h{tokenpos_t(p + 1), 0, 0};
                                if (d == 1 && p == 3) {
                                        h.len = sizeof weight;
                                        memcpy(&h.payload, &weight, sizeof weight);
                                }
                                walk[{id & 31u, id, ids[d]}].push_back(h);
                        }
                }
                Codecs::Google::IndexSession host;
                Codecs::Google::Encoder enc(&host);
                size_t same_terms = 0, nt = 0;
                for (auto it = walk.begin(); it != walk.end();) {
                        const uint32_t id = std::get<1>(it->first);
                        enc.begin_term();
                        for (; it != walk.end() && std::get<1>(it->first) == id; ++it) {
                                enc.begin_document(std::get<2>(it->first));
                                auto hits = it->second;
                                std::sort(hits.begin(), hits.end(), [](const Hit &a, const Hit &b) { return a.pos < b.pos; });
                                for (const Hit &h : hits)
                                        enc.new_hit(h.pos, reinterpret_cast<const uint8_t *>(&h.payload), h.len);
                                enc.end_document();
                        }
                        term_index_ctx t;
                        enc.end_term(&t);
                        same_terms += nt < seg.terms.size() && seg.terms[nt].second.documents == t.documents && seg.terms[nt].second.offset == t.offset && seg.terms[nt].second.size == t.size;
                        ++nt;
                }
                printf("host_encoder %zu bytes, same=%d, term_table_same=%d\n", host.indexOut.size(), int(host.indexOut == seg.index), int(nt == seg.terms.size() && same_terms == nt));
                if (host.indexOut != seg.index || same_terms != nt)
                        return 2;
        }
        // the committed bytes are a segment the read side takes: upload it and merge it with itself (one participant, nothing masked: unchanged postings)
        std::vector<tri_term> table;
        for (const auto &t : seg.terms)
                table.push_back({t.second.documents, t.second.offset, t.second.size});
        tri_index *ix = nullptr;
        check(tri_index_upload(dev, seg.index.data(), seg.index.size(), nullptr, 0, TRI_CODEC_GOOGLE, table.data(), table.size(), 30, &ix));
        std::vector<std::vector<uint32_t>> termOf(table.size());
        for (size_t t = 0; t < table.size(); ++t)
                termOf[t] = {uint32_t(t)};
        const committed_segment merged = merge_google(dev, {ix}, termOf);
        printf("merged %zu bytes, same=%d\n", merged.index.size(), int(merged.index == seg.index));
        tri_index_destroy(ix);
        tri_dev_close(dev);
        return merged.index == seg.index ? 0 : 1;
}
