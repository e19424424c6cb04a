// plan_host.cpp — the host planner (csrc/planner.hpp) and the upload-time walk (csrc/index_host.hpp) behind plain C entry points
// of libtrinity_host.so, WITHOUT a device: what tests/test_planner.py checks on a CPU-only machine (plan invariants, the same plan
// whatever the thread count) and tools/plan_probe.py times.  The product library (libtrinity_hip.so) includes the same two headers;
// nothing here is part of the C-ABI of include/trinity_hip.h.  New code, no reference source.
#include "../planner.hpp"

#include <cstdlib>
#include <memory>

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
        for (unsigned i = 0; i < nopt; ++i) {
                const std::string n = opt_names[i];
                tri_options &o = env.opt;
                uint64_t *slot = n == "dense_min_postings" ? &o.dense_min_postings : n == "dense_task_cost" ? &o.dense_task_cost : n == "fused" ? &o.fused
                                 : n == "fused_task_cost" ? &o.fused_task_cost : n == "fused_freq_cap" ? &o.fused_freq_cap : n == "fused_halfwords" ? &o.fused_halfwords
                                 : n == "account_needed_bytes" ? &o.account_needed_bytes : n == "planes" ? &o.planes : n == "planes_split" ? &o.planes_split
                                 : n == "plane_div" ? &o.plane_div : n == "plane_max_bytes" ? &o.plane_max_bytes : n == "probe_max_blocks" ? &o.probe_max_blocks : n == "tree_max_bytes" ? &o.tree_max_bytes : n == "result_bitmaps" ? &o.result_bitmaps : n == "cand_task_cost" ? &o.cand_task_cost : n == "dense_window_cost" ? &o.dense_window_cost : nullptr;
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
        const int rc = plan_batch(
                H, env, in, pool.get(),
                [&](size_t bytes) {
                        hp->block = static_cast<uint8_t *>(aligned_alloc(64, (bytes + 63) & ~(size_t)63));
                        if (hp->block)
                                memset(hp->block, 0, bytes);
                        return hp->block;
                },
                hp->P, e);
        if (rc != TRI_OK) {
                put_err(err, errcap, e);
                return nullptr;
        }
        return hp.release();
}
void tri_host_plan_free(void *p) { delete static_cast<HostPlan *>(p); }

// sizes and offsets of the plan's sections, counters: out[0..] in the order below
void tri_host_plan_summary(void *p, uint64_t *out /* [64] */, double *ms /* [4] */) {
        const BatchPlan &P = static_cast<HostPlan *>(p)->P;
        const uint64_t v[] = {P.block_bytes,       P.plan.size(),    P.qterms.size(),      P.tasks.size(),  P.fused.size(),       P.qplane.size(),    P.plane_terms.size(), P.sterms.size(),
                              P.sweights.size(),   P.phrases.size(), P.pterms.size(),      P.ptasks.size(), P.off_plan,           P.off_qterms,       P.off_tasks,          P.off_sched,
                              P.off_fused,         P.off_qplane,     P.off_plane_terms,    P.off_sterms,    P.off_sweights,       P.off_phrases,      P.off_pterms,         P.off_ptasks,
                              P.n_dense,           P.n_cand,         P.n_fused,            P.n_fused16,     P.n_fusedgen,         P.n_planes,         P.n_planes8,          P.plw,
                              P.sparse_cap,        P.out_capacity,   P.term_bytes,         P.term_bytes_dense, P.dense_queries,   P.cand_queries,     P.fused_queries,      P.planes_queries,
                              P.unsupported_queries, P.rich_R,       sizeof(DevQuery),     sizeof(DevTask), sizeof(DevFused),     sizeof(DevPhrase),  P.cand_needed_term_bytes, P.plane_decoded_bytes,
                              P.n_pset,            P.pset_queries,   P.n_probe,            P.probe_queries, P.units.size(),       P.off_units,        P.off_pset_sched,     sizeof(DevPsetUnit),
                              P.n_tree,            P.tree_queries,   P.tree.size(),        P.off_tree,      P.tree_terms.size(),  P.off_tree_terms,   P.tree_hidden.size(), P.off_tree_hidden};
        static_assert(sizeof v / sizeof v[0] == 64, "summary layout");
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

// ---- the two ints() payloads (csrc/fastpfor128.hpp) for the CPU tests: encode / decode one 128-value group
extern "C" {
// FastPFor<4>::encodeArray of 128 values: words into out (cap >= 160), returns the word count
uint32_t tri_host_fastpfor_encode(const uint32_t *v, uint32_t *out) {
        std::vector<uint32_t> w;
        trif::fastpfor_encode(v, w);
        memcpy(out, w.data(), w.size() * 4);
        return (uint32_t)w.size();
}
int tri_host_fastpfor_decode(const uint32_t *w, uint32_t L, uint32_t *v) { return trif::fastpfor_decode(w, L, v) ? 1 : 0; }
// facts of a host index the tests compare between the two payload flavours of one corpus: per term {documents, nblocks, last document}
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
