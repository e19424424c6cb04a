// k_probe.hpp — conjunctions of ONE short lead list with lists that all have a term plane (TASK_PROBE): the commonest query of a Zipf
// batch — a rare term AND a head term (cfg2: 7980 of 16384 queries).  Part of libtrinity_hip.so (MI355X / gfx950); included by
// trinity_hip.hip.  New code, no reference source.
//
// What it replaces in the reference: Conjuction::next_impl's leapfrog (docset_iterators.cpp:308-348) — the lead iterator's next(), then
// advance(candidate) on every other iterator (google_codec.cpp:821-934: skiplist search, header hops, block unpack, linear scan).  With
// the other lists held as bitmaps over the docID space (k_term_planes, once per launch for the whole batch) advance(candidate) is ONE bit
// test.  Until round 3 these queries ran as k_and's candidate tiles: a 256-thread workgroup per tile — lead blocks decoded one per lane into an
// LDS candidate array, a barrier, the probes with LDS atomics for the hit bits, a barrier, a workgroup-wide scan, the stores: five barriers
// and ten dependent global loads for a lead list that has a median of 23 blocks (the phase clocks: 41 % decoding with a tenth of the lanes
// busy, 42 % in the filter step, 9 % fetching the task; about 30 us per tile).  Here a WAVE owns a task:
//   * lane b decodes lead block b into row b of the wave's private LDS candidate array (rows rotated: conflict-free), 64 blocks per pass — no
//     workgroup barrier anywhere;
//   * the candidates are then tested 64 CONSECUTIVE ones a step: neighbouring lanes probe neighbouring documents, so a step's gathers fall into
//     a few cache lines of a plane (probing from the decoding lane's registers — lane = block, 32 gathers per lane — was measured first: 1.05 ms
//     for the class against k_and's 0.39: every gather instruction touched 64 cache lines) — OR inside a group, AND across groups, AND-NOT for
//     the excluded group, the segment's masked documents likewise;
//   * the survivors of a step go out ranked by its ballot: ascending, up to 64 per store instruction, no scan;
//   * the task is ONE 64-byte record (DevPsetUnit) instead of the sched -> task -> query -> qterms -> qplane chain; the next task's ticket
//     is drawn while the current one runs.
// A task has the private output region TASK_CAND would have had (planner.hpp), so k_score / k_rich / the result read-back see no difference.
#pragma once

constexpr int PROBE_WG = 256;
constexpr uint32_t PROBE_BLOCKS = 64;                 // lead blocks a wave decodes per pass: one per lane
constexpr uint32_t PROBE_CANDS = PROBE_BLOCKS * 32;   // ... = candidates of a pass
constexpr uint32_t PROBE_UNROLL = 8;                  // steps of 64 candidates whose plane gathers are in flight together
static_assert(TILE_BLOCKS % PROBE_BLOCKS == 0, "a lead tile is a whole number of passes");

struct ProbeShared {
        uint32_t cand[PROBE_WG / 64][PROBE_CANDS]; // per wave: the pass's candidates, rows rotated (phys(): lane-per-row writes hit 32 banks)
};

// waves per SIMD the register budget is cut for: the decoder's state (the prefix-varint byte stream: a 16-byte window + three qwords in
// flight; the PFOR128 quarter reader) is all a lane keeps — the documents go to LDS as they are decoded
#ifndef TRI_PROBE_WAVES
#define TRI_PROBE_WAVES 5 // (LDS: 32 KB per 256-thread workgroup: five per CU)
#endif

template <int CODEC>
__global__ __launch_bounds__(PROBE_WG, TRI_PROBE_WAVES) void k_probe(const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last, const uint32_t *__restrict__ blk_off,
                                                                      const DevTerm *__restrict__ terms, const DevPsetUnit *__restrict__ units, const uint32_t *__restrict__ order,
                                                                      const uint32_t ntasks, uint32_t *__restrict__ ticket, const uint32_t *__restrict__ qterms,
                                                                      const uint32_t *__restrict__ qplane, uint32_t *__restrict__ out, uint32_t *__restrict__ counts,
                                                                      const uint32_t *__restrict__ masked, const uint32_t *__restrict__ planes, const uint32_t plw) {
        __shared__ ProbeShared sh;
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t *const cand = sh.cand[uni(threadIdx.x >> 6)];
        // a wave draws its own tickets (one atomic per task and wave; the next ticket is requested before the current task is run)
        uint32_t tk = lane == 0 ? atomicAdd(ticket, 1u) : 0u;
        tk = uni(tk);
        while (tk < ntasks) {
                uint32_t tk_next = lane == 0 ? atomicAdd(ticket, 1u) : 0u; // (used at the loop's end)
                // DevPsetUnit: words 0-1 out_off, 2 first tile, 3 end tile, 4 tix, 5 nterms, 6 term_base, 8.. tt[4], 12.. row[4]: lane l holds word l
                const uint32_t rec = ((const uint32_t *)(units + order[tk]))[lane & 15u];
                auto word = [&](const int i) { return (uint32_t)__builtin_amdgcn_readlane((int)rec, i); };
                const uint32_t nterms = word(5), term_base = word(6), tix = word(4);
                uint32_t *const qout = out + (((uint64_t)word(1) << 32) | word(0));
                const DevTerm lead = terms[word(8) & QT_TERM];
                const uint32_t tb_end = min(lead.nblocks, word(3) * (uint32_t)TILE_BLOCKS);
                uint32_t produced = 0;
                for (uint32_t tb = word(2) * (uint32_t)TILE_BLOCKS; tb < tb_end; tb += PROBE_BLOCKS) {
                        // ---- decode: lane b takes lead block tb + b (unpack_block, google_codec.cpp:596-639): its documents into row b of cand[]
                        const uint32_t b = tb + lane;
                        const uint32_t nb = min(PROBE_BLOCKS, tb_end - tb);
                        // (only a list's last block may hold fewer than 32 documents — verified at upload —, so the pass's candidates are contiguous)
                        const uint32_t C = tb + nb == lead.nblocks ? (nb - 1) * 32 + lead.last_n : nb * 32;
                        if (b < tb_end) {
                                const uint32_t gb = lead.first_block + b;
                                const uint32_t off = blk_off[gb];
                                const uint32_t n = TRI_BLOCK_N(lead, b, index, off);
                                const uint32_t last = blk_last[gb];
                                uint32_t doc = b ? blk_last[gb - 1] : 0;
                                DeltaStream<CODEC> s;
                                s.init(index, lead, b, off);
                                const uint32_t row = lane * 32;
                                for (uint32_t i = 0; i + 1 < n; ++i) {
                                        doc += s.next();
                                        cand[row | ((i + lane) & 31u)] = doc;
                                }
                                cand[row | ((n - 1 + lane) & 31u)] = last;
                        }
                        __builtin_amdgcn_wave_barrier();
                        // ---- test, 64 consecutive candidates a step: neighbouring lanes probe neighbouring documents (a step's gathers fall into a
                        //      few cache lines of the plane) — OR inside a group, AND across groups, AND-NOT for the excluded group — and the
                        //      survivors go out ranked by the step's ballot: ascending, 64 at most per store instruction
                        // (PROBE_UNROLL steps at a time: their gathers are all requested before the first is used — one memory round trip per PROBE_UNROLL
                        //  steps, not per step; with the steps one after the other a pass was 32 dependent round trips: 1.2 ms for the class)
                        for (uint32_t j0 = 0; j0 < C; j0 += 64u * PROBE_UNROLL) {
                                uint32_t doc[PROBE_UNROLL], keep = 0, grp = 0, live = 0; // (bit u: step u's candidate of this lane)
#pragma unroll
                                for (uint32_t u = 0; u < PROBE_UNROLL; ++u) {
                                        const uint32_t j = j0 + 64u * u + lane;
                                        doc[u] = j < C ? cand[phys(j)] : 0u;
                                        live |= (j < C ? 1u : 0u) << u;
                                }
                                keep = live;
                                bool neg = false;
                                for (uint32_t k = 1; k < nterms; ++k) {
                                        uint32_t tt, row;
                                        if (k < PSET_INLINE_TERMS) {
                                                tt = word((int)(8 + k));
                                                row = word((int)(12 + k));
                                        } else {
                                                tt = uni(qterms[term_base + k]);
                                                row = uni(qplane[term_base + k]);
                                        }
                                        if (tt & QT_GROUP) {
                                                if (k > 1)
                                                        keep &= neg ? ~grp : grp;
                                                grp = 0;
                                                neg = tt & QT_NOT;
                                        }
                                        const uint32_t *pa = planes + (size_t)row * plw;
#pragma unroll
                                        for (uint32_t u = 0; u < PROBE_UNROLL; ++u)
                                                grp |= ((pa[doc[u] >> 5] >> (doc[u] & 31u)) & 1u) << u;
                                }
                                if (nterms > 1)
                                        keep &= neg ? ~grp : grp;
                                if (masked) { // masked_documents_registry::test (docidupdates.h:90-119): documents updated / deleted elsewhere never match
                                        uint32_t hit = 0;
#pragma unroll
                                        for (uint32_t u = 0; u < PROBE_UNROLL; ++u)
                                                hit |= ((masked[doc[u] >> 5] >> (doc[u] & 31u)) & 1u) << u;
                                        keep &= ~hit;
                                }
#pragma unroll
                                for (uint32_t u = 0; u < PROBE_UNROLL; ++u) {
                                        const bool kp = (keep >> u) & 1u;
                                        const uint64_t bm = __builtin_amdgcn_ballot_w64(kp);
                                        if (kp)
                                                qout[produced + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u))] = doc[u];
                                        produced += (uint32_t)__popcll(bm);
                                }
                        }
                        __builtin_amdgcn_wave_barrier(); // (the next pass's rows stay behind these reads)
                }
                if (lane == 0)
                        counts[tix] = produced;
                tk = uni(tk_next);
        }
}
