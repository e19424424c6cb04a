// plan_host.cpp — the host planner (csrc/planner.hpp) and the upload-time walk (csrc/index_host.hpp) behind plain C entry points
// of libtrinity_host.so, WITHOUT a device: what tests/test_planner.py checks on a CPU-only machine (plan invariants, the same plan
// whatever the thread count) and tools/plan_probe.py times.  The product library (libtrinity_hip.so) includes the same two headers;
// nothing here is part of the C-ABI of include/trinity_hip.h.  New code, no reference source.
#include "../planner.hpp"
#include "../lucene_enc_units.hpp"
#include "lucene_encoder.hpp"

#include <cstdlib>
#include <memory>
#include <mutex>

namespace {
        struct HostPlan {
                BatchPlan P;
                uint8_t *block = nullptr;
                ~HostPlan() { free(block); }
        };
        void put_err(char *err, size_t cap, const std::string &s) {
                if (err && cap) {
                        const size_t n = std::min(cap - 1, s.size());
                        memcpy(err, s.data(), n);
                        err[n] = 0;
                }
        }
} // namespace

extern "C" {
void *tri_host_index_build(const uint8_t *index, uint64_t len, const uint8_t *hits, uint64_t hits_len, int codec, const uint32_t *terms3, uint64_t nterms,
                           uint32_t docs_cnt, char *err, uint64_t errcap) {
        auto H = std::make_unique<HostIndex>();
        std::string e;
        static_assert(sizeof(tri_term) == 12, "tri_term is three u32");
        if (build_host_index(index, len, hits, hits_len, codec, reinterpret_cast<const tri_term *>(terms3), nterms, docs_cnt, *H, e) != TRI_OK) {
                put_err(err, errcap, e);
                return nullptr;
        }
        return H.release();
}
void tri_host_index_free(void *h) { delete static_cast<HostIndex *>(h); }

// options: `nopt` (name, value) pairs by the names of tri_dev_set_option that the planner reads.  threads: host threads (1 = none).
// Returns a plan handle (NULL + err on failure).
void *tri_host_plan(void *hindex, const uint32_t *prog, uint64_t prog_len, const uint32_t *queries2, uint64_t nq, uint32_t flags, uint32_t topk, int similarity,
                    unsigned threads, const char *const *opt_names, const uint64_t *opt_values, unsigned nopt, uint32_t cus, char *err, uint64_t errcap) {
        const HostIndex &H = *static_cast<HostIndex *>(hindex);
        PlanEnv env;
        env.cus = cus ? cus : 256;
        bool use_frag_cache = false; // option "frag_cache" = 1 (this harness only): the fragments' buffers of earlier calls are reused, as tri_dev does
        for (unsigned i = 0; i < nopt; ++i) {
                const std::string n = opt_names[i];
                if (n == "frag_cache") {
                        use_frag_cache = opt_values[i] != 0;
                        continue;
                }
                tri_options &o = env.opt;
                uint64_t *slot = n == "dense_min_postings" ? &o.dense_min_postings : n == "dense_task_cost" ? &o.dense_task_cost : n == "fused" ? &o.fused
                                 : n == "fused_task_cost" ? &o.fused_task_cost : n == "fused_freq_cap" ? &o.fused_freq_cap : n == "fused_halfwords" ? &o.fused_halfwords
                                 : n == "account_needed_bytes" ? &o.account_needed_bytes : n == "planes" ? &o.planes : n == "planes_split" ? &o.planes_split
                                 : n == "plane_div" ? &o.plane_div : n == "plane_max_bytes" ? &o.plane_max_bytes : n == "probe_max_blocks" ? &o.probe_max_blocks : n == "tree_max_bytes" ? &o.tree_max_bytes : n == "result_bitmaps" ? &o.result_bitmaps : n == "cand_task_cost" ? &o.cand_task_cost : n == "dense_window_cost" ? &o.dense_window_cost
                                 : n == "cand_xcd" ? &o.cand_xcd : n == "planes_order" ? &o.planes_order : n == "pset_order" ? &o.pset_order : n == "scatter_bitmap_slack" ? &o.scatter_bitmap_slack : n == "phrase_task_div" ? &o.phrase_task_div : n == "plane_amortize" ? &o.plane_amortize : nullptr;
                if (!slot) {
                        put_err(err, errcap, "unknown option " + n);
                        return nullptr;
                }
                *slot = opt_values[i];
        }
        PlanInput in;
        in.prog = prog;
        in.prog_len = prog_len;
        static_assert(sizeof(tri_query) == 8, "tri_query is two u32");
        in.queries = reinterpret_cast<const tri_query *>(queries2);
        in.nq = nq;
        in.flags = flags;
        in.topk = topk;
        in.similarity = similarity;
        auto hp = std::make_unique<HostPlan>();
        std::unique_ptr<HostPool> pool;
        if (threads > 1)
                pool = std::make_unique<HostPool>(threads);
        std::string e;
        static trip::FragCache g_frag_cache;
        static std::mutex g_frag_mu;
        std::unique_lock<std::mutex> frag_lock(g_frag_mu, std::defer_lock);
        if (use_frag_cache)
                frag_lock.lock();
        const int rc = plan_batch(
                H, env, in, pool.get(),
                [&](size_t bytes) {
                        hp->block = static_cast<uint8_t *>(aligned_alloc(64, (bytes + 63) & ~(size_t)63));
                        if (hp->block)
                                memset(hp->block, 0, bytes);
                        return hp->block;
                },
                hp->P, e, use_frag_cache ? &g_frag_cache : nullptr);
        if (rc != TRI_OK) {
                put_err(err, errcap, e);
                return nullptr;
        }
        return hp.release();
}
void tri_host_plan_free(void *p) { delete static_cast<HostPlan *>(p); }

// sizes and offsets of the plan's sections, counters: out[0..] in the order below
void tri_host_plan_summary(void *p, uint64_t *out /* [65] */, double *ms /* [4] */) {
        const BatchPlan &P = static_cast<HostPlan *>(p)->P;
        const uint64_t v[] = {P.block_bytes,       P.plan.size(),    P.qterms.size(),      P.tasks.size(),  P.fused.size(),       P.qplane.size(),    P.plane_terms.size(), P.sterms.size(),
                              P.sweights.size(),   P.phrases.size(), P.pterms.size(),      P.ptasks.size(), P.off_plan,           P.off_qterms,       P.off_tasks,          P.off_sched,
                              P.off_fused,         P.off_qplane,     P.off_plane_terms,    P.off_sterms,    P.off_sweights,       P.off_phrases,      P.off_pterms,         P.off_ptasks,
                              P.n_dense,           P.n_cand,         P.n_fused,            P.n_fused16,     P.n_fusedgen,         P.n_planes,         P.n_planes8,          P.plw,
                              P.sparse_cap,        P.out_capacity,   P.term_bytes,         P.term_bytes_dense, P.dense_queries,   P.cand_queries,     P.fused_queries,      P.planes_queries,
                              P.unsupported_queries, P.rich_R,       sizeof(DevQuery),     sizeof(DevTask), sizeof(DevFused),     sizeof(DevPhrase),  P.cand_needed_term_bytes, P.plane_decoded_bytes,
                              P.n_pset,            P.pset_queries,   P.n_probe,            P.probe_queries, P.units.size(),       P.off_units,        P.off_pset_sched,     sizeof(DevPsetUnit),
                              P.n_tree,            P.tree_queries,   P.tree.size(),        P.off_tree,      P.tree_terms.size(),  P.off_tree_terms,   P.tree_hidden.size(), P.off_tree_hidden,
                              P.off_cand_q};
        static_assert(sizeof v / sizeof v[0] == 65, "summary layout");
        memcpy(out, v, sizeof v);
        if (ms)
                memcpy(ms, P.plan_ms, sizeof P.plan_ms);
}
const uint8_t *tri_host_plan_block(void *p) { return static_cast<HostPlan *>(p)->P.block; }
void tri_host_plan_query_maps(void *p, uint32_t *slot_of_query, int32_t *qstatus) {
        const BatchPlan &P = static_cast<HostPlan *>(p)->P;
        memcpy(slot_of_query, P.slot_of_query.data(), P.slot_of_query.size() * 4);
        memcpy(qstatus, P.qstatus.data(), P.qstatus.size() * 4);
}
}

// ---- the Lucene-shaped encoder two ways, for the CPU tests: the sequential host encoder (lucene_encoder.hpp: the state machine the reference's encoder is),
//      and the device encoder's UNITS (lucene_enc_units.hpp) run in plain loops with serial prefix sums in between — exactly what k_lencode.hpp's kernels and
//      the device scans do.  Both write index / hits.data / the term table {documents, offset, size}; return 0, or -1 when a buffer is too small.
extern "C" {
int tri_host_lucene_encode(const uint32_t *docs, const uint32_t *freqs, const uint16_t *pos, const uint64_t *term_first, uint64_t nterms, uint8_t *index_out, uint64_t icap,
                           uint64_t *ilen, uint8_t *hits_out, uint64_t hcap, uint64_t *hlen, uint32_t *terms3) {
        using namespace trinity_amd::Codecs;
        Lucene::IndexSession sess;
        Lucene::Encoder enc(&sess);
        uint64_t h = 0;
        for (uint64_t t = 0; t < nterms; ++t) {
                enc.begin_term();
                for (uint64_t p = term_first[t]; p < term_first[t + 1]; ++p) {
                        enc.begin_document(docs[p]);
                        for (uint32_t i = 0; i < freqs[p]; ++i)
                                enc.new_hit(pos[h++]);
                        enc.end_document();
                }
                trinity_amd::term_index_ctx tctx;
                enc.end_term(&tctx);
                terms3[3 * t] = tctx.documents, terms3[3 * t + 1] = tctx.offset, terms3[3 * t + 2] = tctx.size;
        }
        *ilen = sess.indexOut.size(), *hlen = sess.positionsOut.size();
        if (*ilen > icap || *hlen > hcap)
                return -1;
        if (*ilen)
                memcpy(index_out, sess.indexOut.data(), *ilen);
        if (*hlen) // (a session without hits has no buffer to copy from)
                memcpy(hits_out, sess.positionsOut.data(), *hlen);
        return 0;
}
int tri_host_lucene_encode_units(const uint32_t *docs, const uint32_t *freqs, const uint16_t *pos, const uint64_t *term_first, uint64_t nterms, uint8_t *index_out, uint64_t icap,
                                 uint64_t *ilen, uint8_t *hits_out, uint64_t hcap, uint64_t *hlen, uint32_t *terms3) {
        const uint64_t np = term_first[nterms];
        std::vector<uint64_t> hit_off(np + 1, 0), dblk_first(nterms + 1, 0), hblk_first(nterms + 1, 0);
        for (uint64_t p = 0; p < np; ++p)
                hit_off[p + 1] = hit_off[p] + freqs[p];
        for (uint64_t t = 0; t < nterms; ++t) {
                dblk_first[t + 1] = dblk_first[t] + (term_first[t + 1] - term_first[t]) / LENC_BLOCK;
                hblk_first[t + 1] = hblk_first[t] + (hit_off[term_first[t + 1]] - hit_off[term_first[t]]) / LENC_BLOCK;
        }
        std::vector<uint32_t> hdelta(hit_off[np] + 1);
        LencArgs a{docs, freqs, pos, hit_off.data(), term_first, hdelta.data(), dblk_first.data(), hblk_first.data(), nterms};
        for (uint64_t p = 0; p < np; ++p)
                lenc_unit_hdelta(a, p, hdelta.data());
        const uint64_t nd = dblk_first[nterms], nh = hblk_first[nterms];
        std::vector<uint64_t> doff(nd + 1, 0), hoff(nh + 1, 0), term_off(nterms + 1, 0), hterm_off(nterms + 1, 0);
        std::vector<uint32_t> tail_d(nterms + 1), tail_h(nterms + 1);
        for (uint64_t g = 0; g < nd; ++g)
                doff[g + 1] = doff[g] + lenc_unit_dblk_size(a, g);
        for (uint64_t h = 0; h < nh; ++h)
                hoff[h + 1] = hoff[h] + lenc_unit_hblk_size(a, h);
        for (uint64_t t = 0; t < nterms; ++t)
                lenc_unit_tail_size(a, t, &tail_d[t], &tail_h[t]);
        LencPlace pl{doff.data(), hoff.data(), term_off.data(), hterm_off.data(), tail_d.data(), tail_h.data()};
        for (uint64_t t = 0; t < nterms; ++t) {
                term_off[t + 1] = term_off[t] + lenc_term_index_size(a, pl, t);
                hterm_off[t + 1] = hterm_off[t] + lenc_term_hits_size(a, pl, t);
        }
        *ilen = term_off[nterms], *hlen = hterm_off[nterms];
        if (*ilen > icap || *hlen > hcap)
                return -1;
        for (uint64_t g = 0; g < nd; ++g)
                lenc_unit_dblk_write(a, pl, g, index_out);
        for (uint64_t h = 0; h < nh; ++h)
                lenc_unit_hblk_write(a, pl, h, hits_out);
        for (uint64_t t = 0; t < nterms; ++t) {
                lenc_unit_term_write(a, pl, t, index_out, hits_out);
                terms3[3 * t] = (uint32_t)(term_first[t + 1] - term_first[t]), terms3[3 * t + 1] = (uint32_t)term_off[t], terms3[3 * t + 2] = lenc_term_index_size(a, pl, t);
        }
        return 0;
}
}

// ---- the two ints() payloads (csrc/fastpfor128.hpp) for the CPU tests: encode / decode one 128-value group
extern "C" {
// one ints() group two ways: the device encoder's plan / emit pair (pfor128_group.hpp) into a, the host encoder (lucene_encoder.hpp) into b; returns both lengths
void tri_host_pfor128_group(const uint32_t *v, uint8_t *a, uint32_t *alen, uint8_t *b, uint32_t *blen) {
        auto get = [&](uint32_t i) { return v[i]; };
        const Pfor128Plan p = pfor128_plan(get);
        uint8_t *e = pfor128_emit(get, p, a);
        *alen = (uint32_t)(e - a);
        if ((uint32_t)(e - a) != p.bytes)
                *alen = 0xffffffffu; // (the plan and the emitter disagree)
        std::vector<uint8_t> o;
        trinity_amd::Codecs::Lucene::ints_encode(v, o);
        memcpy(b, o.data(), o.size());
        *blen = (uint32_t)o.size();
}
// FastPFor<4>::encodeArray of 128 values: words into out (cap >= 160), returns the word count
uint32_t tri_host_fastpfor_encode(const uint32_t *v, uint32_t *out) {
        std::vector<uint32_t> w;
        trif::fastpfor_encode(v, w);
        memcpy(out, w.data(), w.size() * 4);
        return (uint32_t)w.size();
}
int tri_host_fastpfor_decode(const uint32_t *w, uint32_t L, uint32_t *v) { return trif::fastpfor_decode(w, L, v) ? 1 : 0; }
// facts of a host index the tests compare between the two payload flavours of one corpus: per term {documents, nblocks, last document}
// the CPUs a device handle's planner pool of `threads` threads pins its workers to in THIS process (LOCAL_RANK / LOCAL_WORLD_SIZE in the environment:
// the rank's slice of the affinity mask, csrc/host_pool.hpp); returns how many (<= cap written)
uint32_t tri_host_pool_cpus(uint32_t threads, int32_t *out, uint32_t cap) {
        HostPool pool(threads);
        const std::vector<int> &c = pool.pinned_cpus();
        for (size_t i = 0; i < c.size() && i < cap; ++i)
                out[i] = c[i];
        return (uint32_t)c.size();
}

uint32_t tri_host_cpu_budget() { return host_cpu_budget(); } // (host_pool.hpp: what the planner's pools are sized to)

void tri_host_index_facts(void *h, uint64_t *info6, uint32_t *per_term3) {
        const HostIndex &H = *static_cast<HostIndex *>(h);
        info6[0] = H.info.postings, info6[1] = H.info.blocks, info6[2] = H.info.doc_bytes, info6[3] = H.info.hit_bytes, info6[4] = H.transcoded_groups, info6[5] = H.dev_index.size();
        for (size_t t = 0; t < H.terms.size(); ++t) {
                per_term3[3 * t] = H.terms[t].documents;
                per_term3[3 * t + 1] = H.terms[t].nblocks;
                per_term3[3 * t + 2] = H.terms[t].nblocks ? H.blk_last[H.terms[t].first_block + H.terms[t].nblocks - 1] : 0;
        }
}
}
