// k_psets.hpp — docset algebra of the queries ALL of whose terms have a term plane (TASK_PSET): intersections / unions / exclusions of head
// terms (and, round 5, unions of head terms with others whose few documents are scattered into the stored result: PSET_UNIT_SCATTER), the queries that materialise the batch's largest docID sets (cfg2: 1551 of 16384 queries write 380 M of the step's 395 M matches).
// Part of libtrinity_hip.so (MI355X / gfx950); included by trinity_hip.hip.  New code, no reference source.
//
// What it replaces in the reference: Conjuction::next_impl / DisjunctionAllPLI::next over Google::Decoder::next (docset_iterators.cpp:
// 226-405; google_codec.cpp:777-819) driven by GenericDocsSetSpan::process / DocsSetSpanForDisjunctions::process (docset_spans.cpp:98-173,
// 269-290) — per matching document two virtual calls and a decode step; here a docID window of 131072 documents is 4096 words of
// AND / OR / AND-NOT over the planes k_term_planes decoded once for the whole batch, and the work is the EXPANSION of the survivors
// into ascending docIDs.
//
// Until round 3 these queries ran in k_and_dense's plane-only branch: plane words -> LDS bitmap -> a workgroup-wide scan (five barriers
// per window) -> every lane storing its own words' docIDs straight to HBM, 4 bytes at a time, a wave's store instruction spread over a
// dozen cache lines (r03: 11.8 us per window and workgroup).  Here (rounds 3 - 6; psets_round below says what round 6 changed and why):
//   * a task (four docID windows of one query) runs in ROUNDS of 1 / 2 / 4 windows (the planner picks: DevPsetUnit::first); in a round a
//     wave owns a contiguous eighth of the words — a sub-window of 512 words (16384 documents) per window: two 16-byte loads per lane
//     and term —, keeps a sub-window's survivors in registers, expands them into a private LDS staging buffer (scattered 4-byte LDS
//     writes are cheap), and only its COUNT crosses to the other waves, ONCE per round; then it copies the buffer out with coalesced
//     stores: a store instruction covers 256 contiguous bytes;
//   * a wave whose share holds more survivors than the staging buffer walks its words again behind the barrier, sub-window by
//     sub-window; a sub-window denser than the buffer (a union of head terms: one document in 16 or denser) one lane per BIT — ballot,
//     rank by mbcnt, one coalesced store per 64 bits;
//   * a task is ONE 64-byte record (DevPsetUnit) fetched a task ahead by wave 0 together with the ticket after it, instead of the
//     sched -> task -> query -> qterms / qplane chain of dependent loads; the schedule runs the tasks window range by window range (and,
//     option pset_order, by the query's heaviest term within a range), so that a range's plane words are in the XCDs' L2 / the
//     Infinity Cache while every query that reads them is in flight.
// A task has a private, bound-allocated output region (planner.hpp: the same layout as TASK_DENSE, so k_score / k_rich / k_phrase / the
// result read-back see no difference).
// Measured at cfg2 (1804 queries, 36 K tasks, 380 M matches of which 279 M stay bitmap bits): round 3 1.05 ms, round 5 0.56 (the pair path, dense
// results as bitmaps), round 6 0.49 - 0.50.  Probe builds (-DTRI_PSET_VARIANT=1 counts only, =2 no copy-out, =3 the general loop without its
// plane loads) and what was tried and not kept (each correct, each measured): DESIGN.md §15.9.
#pragma once

#ifndef TRI_PSET_WG
#define TRI_PSET_WG 512
#endif
constexpr int PSET_WG = TRI_PSET_WG;
constexpr uint32_t PSET_WAVES = PSET_WG / 64;
constexpr uint32_t PSET_WORDS = 512;                     // words of a step a wave owns (16384 documents) ...
constexpr uint32_t PSET_PER = PSET_WORDS / 64;           // ... and a lane: 8 (two 16-byte loads per term)
constexpr uint32_t PSET_STEP_WORDS = PSET_WAVES * PSET_WORDS; // words the workgroup takes per step (between two barriers)
static_assert(SPAN_WORDS % PSET_STEP_WORDS == 0, "a docID window is a whole number of steps");
constexpr uint32_t PSET_STAGE = 1024;                    // docIDs a wave stages per sub-window before it copies them out
static_assert(PSET_PER == 8, "two 16-byte loads per lane and term");
static_assert(PSET_WAVES == 8, "the waves' counts are prefix-summed over lanes 0 .. 7 (psets_task)");
static_assert(PSET_WORDS * 32 == PSET_ROUND_DOCS && PSET_STAGE == PSET_STAGE_DOCS, "the planner sizes a task's rounds by them (dev_structs.hpp)");
static_assert(SPAN_WORDS == PSET_WAVES * PSET_WORDS, "a wave's share of a task: one sub-window of PSET_WORDS words per docID window");
static_assert(PSET_STAGE >= PSET_WORDS, "the dense walk parks the wave's words in its staging buffer");

struct PsetShared {
        uint32_t stage[PSET_WAVES][PSET_STAGE];
        alignas(16) uint32_t cnt[2][PSET_WAVES]; // per window parity: the waves' survivor counts (psets_pair reads a parity's eight as two 16-byte words)
        DevPsetUnit unit[2];         // the task being run and the next one (fetched while the current one runs)
        uint32_t tick[2];            // ... and their tickets (>= ntasks: none)
};

// ---- a task's windows -> its output region.  `words(word0, acc)` gives the lane's eight survivor words at word0 (the pair loop and the general loop of the kernel below).
//      Round 6, in two steps:
//      (1) the kernel's VALUs were four fifths busy (SQ counters) and 150 of a wave's 250 vector instructions per step were NOT the expansion: a wave scan of six __shfl_up
//          (a ds_bpermute_b32, an s_waitcnt and four VALUs each), the eight counts read one by one and added under eight `wv < wave` conditions the compiler hoisted out of
//          the loop into SGPR pairs, spilled, and read back lane by lane with v_readlane.  Now the scan is six DPP adds (wave_excl_scan), the counts are read ONE per lane
//          and prefix-summed by three more, the wave's base and the task's total are two v_readlane (0.56 -> 0.50 ms at cfg2).
//      (2) a wave owned 512 words of every 4096-word STEP, the waves' counts crossed at a barrier per step, and every step ended with its copy-out stores — and gfx9 has ONE
//          counter for loads and stores: the next step's `s_waitcnt vmcnt(0)` for its plane words waited for those stores' acknowledgements too (a probe build without the
//          copy-out ran as fast as one without the whole expansion; requesting the next step's words early gained nothing).  Now a wave owns a CONTIGUOUS eighth of the
//          task's words (its windows x 512 words), stages every survivor of its share in LDS — no barrier, no store until the share is through —, the waves' totals cross
//          ONCE per task, and each wave copies its docIDs out in one piece.  A wave whose share holds more survivors than its staging buffer (a union of head terms: one
//          document in 16 or denser) counts them all the same and then walks its words AGAIN behind the barrier, one lane per BIT — ballot, rank by mbcnt, one coalesced
//          store per 64 bits.
//      Returns the task's survivor count (as_bitmap: this LANE's share of it — the caller reduces).
//      `request(word0, buf)` issues the loads of the lane's words at word0, `words(buf, word0, acc)` folds them into the eight survivor words: with
//      TRI_PSET_PREFETCH a sub-window's words are requested BEFORE the one ahead of it is counted and expanded.
#ifndef TRI_PSET_PREFETCH
#define TRI_PSET_PREFETCH 0 // 1: a sub-window's words are requested before the one ahead of it is counted and expanded.  Measured (cfg2): 0.525 ms against 0.489 at eight waves per
                            // SIMD (sixteen more registers: 8 spilled), 0.503 at six without spills — the kernel does not wait for the latency of ITS loads
#endif
struct PsetNoBuf {};
struct PsetPairBuf {
        uint4 a0, a1, b0, b1;
};
template <class BUF, class R, class F>
__device__ __forceinline__ uint32_t psets_round(PsetShared &sh, R &&request, F &&words, const uint32_t w_origin, const uint32_t w_begin, const uint32_t w_end, uint32_t *__restrict__ qout,
                                                const bool as_bitmap, const uint32_t lane, const uint32_t wave, const uint32_t par) {
        const uint32_t nsub = w_end - w_begin; // sub-windows of PSET_WORDS words of this wave's share (the task's words / PSET_WAVES)
        const uint32_t wave0 = w_begin * SPAN_WORDS + wave * nsub * PSET_WORDS + lane * PSET_PER; // this lane's first word of its wave's share
        uint32_t *const st = sh.stage[wave];
        if (as_bitmap) { // nothing to expand, nothing to rank: two 16-byte stores per lane and sub-window, the counts summed at the task's end
                uint32_t lane_cnt = 0;
                BUF nxt;
                if (TRI_PSET_PREFETCH)
                        request(wave0, nxt);
                for (uint32_t sb = 0; sb < nsub; ++sb) {
                        const uint32_t word0 = wave0 + sb * PSET_WORDS;
                        BUF cur;
                        if (TRI_PSET_PREFETCH) {
                                cur = nxt;
                                if (sb + 1 < nsub)
                                        request(word0 + PSET_WORDS, nxt);
                        } else
                                request(word0, cur);
                        uint32_t acc[PSET_PER];
                        words(cur, word0, acc);
                        uint4 *o = (uint4 *)(qout + (word0 - w_origin * SPAN_WORDS));
                        o[0] = make_uint4(acc[0], acc[1], acc[2], acc[3]);
                        o[1] = make_uint4(acc[4], acc[5], acc[6], acc[7]);
#pragma unroll
                        for (uint32_t j = 0; j < PSET_PER; ++j)
                                lane_cnt += (uint32_t)__popc(acc[j]);
                }
                return lane_cnt;
        }
        uint32_t fill = 0; // (wave-uniform) survivors of the share so far; staged while they fit
        BUF nxt;
        if (TRI_PSET_PREFETCH)
                request(wave0, nxt);
        for (uint32_t sb = 0; sb < nsub; ++sb) {
                const uint32_t word0 = wave0 + sb * PSET_WORDS;
                BUF cur;
                if (TRI_PSET_PREFETCH) {
                        cur = nxt;
                        if (sb + 1 < nsub)
                                request(word0 + PSET_WORDS, nxt);
                } else
                        request(word0, cur);
                uint32_t acc[PSET_PER];
                words(cur, word0, acc);
                uint32_t c = 0;
#pragma unroll
                for (uint32_t j = 0; j < PSET_PER; ++j)
                        c += (uint32_t)__popc(acc[j]);
                uint32_t T;
                const uint32_t ex = wave_excl_scan(c, T);
#if !defined(TRI_PSET_VARIANT) || TRI_PSET_VARIANT != 1 // (perf probe 1: counts only)
                if (T && fill + T <= PSET_STAGE) { // (uniform) every lane writes its words' docIDs into the wave's staging buffer at its rank
                        uint32_t o = fill + ex;
#pragma unroll
                        for (uint32_t j = 0; j < PSET_PER; ++j) {
                                uint32_t m = acc[j];
                                const uint32_t d0 = (word0 + j) << 5;
                                while (m) {
                                        st[o++] = d0 + (uint32_t)__builtin_ctz(m);
                                        m &= m - 1u;
                                }
                        }
                }
#endif
                fill += T;
        }
        // ---- the waves' totals cross: lane k (k < 8; the others repeat them) reads wave k's, an inclusive prefix over lanes 0 .. 7 is three DPP adds
        if (lane == 0)
                sh.cnt[par][wave] = fill;
        __syncthreads();
        uint32_t own = sh.cnt[par][lane & (PSET_WAVES - 1u)], pre = own;
        asm volatile("s_nop 1\n\t"
                     "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "s_nop 1\n\t"
                     "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "s_nop 1\n\t"
                     "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "s_nop 1"
                     : "+v"(pre));
        const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)(pre - own), (int)wave);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)pre, (int)PSET_WAVES - 1);
#if defined(TRI_PSET_VARIANT) && (TRI_PSET_VARIANT == 1 || TRI_PSET_VARIANT == 2) // (perf probes: no copy-out)
        return total;
#endif
        if (fill <= PSET_STAGE) { // (uniform) the wave's docIDs go out in one piece — 64 consecutive docIDs per store instruction
                __builtin_amdgcn_wave_barrier();
                for (uint32_t i = lane; i < fill; i += 64u)
                        qout[base + i] = st[i];
                __builtin_amdgcn_wave_barrier(); // (the next task's staging writes stay behind these reads)
        } else {
                // ---- the share held more than the buffer (the planner's estimate fell short, or a single sub-window is that dense): its words once more (from L2),
                //      sub-window by sub-window, now that the wave knows where its docIDs go — staged and copied out where a sub-window's survivors fit, else the
                //      sub-window's 512 words parked in LDS and one lane per BIT, 64 bits a step: ballot, rank by mbcnt, one coalesced store
                uint32_t o = base;
                for (uint32_t sb = 0; sb < nsub; ++sb) {
                        const uint32_t word0 = wave0 + sb * PSET_WORDS;
                        uint32_t acc[PSET_PER];
                        BUF cur;
                        request(word0, cur);
                        words(cur, word0, acc);
                        uint32_t c = 0;
#pragma unroll
                        for (uint32_t j = 0; j < PSET_PER; ++j)
                                c += (uint32_t)__popc(acc[j]);
                        uint32_t T;
                        const uint32_t ex = wave_excl_scan(c, T);
                        if (!T)
                                continue;
                        if (T <= PSET_STAGE) {
                                uint32_t at = ex;
#pragma unroll
                                for (uint32_t j = 0; j < PSET_PER; ++j) {
                                        uint32_t m = acc[j];
                                        const uint32_t d0 = (word0 + j) << 5;
                                        while (m) {
                                                st[at++] = d0 + (uint32_t)__builtin_ctz(m);
                                                m &= m - 1u;
                                        }
                                }
                                __builtin_amdgcn_wave_barrier();
                                for (uint32_t i = lane; i < T; i += 64u)
                                        qout[o + i] = st[i];
                                __builtin_amdgcn_wave_barrier();
                                o += T;
                                continue;
                        }
#pragma unroll
                        for (uint32_t j = 0; j < PSET_PER; ++j)
                                st[lane * PSET_PER + j] = acc[j];
                        __builtin_amdgcn_wave_barrier();
                        const uint32_t wbase = word0 - lane * PSET_PER;
                        for (uint32_t cidx = 0; cidx < PSET_WORDS / 2; ++cidx) {
                                const uint32_t wi = 2u * cidx + (lane >> 5);
                                const bool bit = (st[wi] >> (lane & 31u)) & 1u;
                                const uint64_t bm = __builtin_amdgcn_ballot_w64(bit);
                                if (bit)
                                        qout[o + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u))] = ((wbase + wi) << 5) + (lane & 31u);
                                o += (uint32_t)__popcll(bm);
                        }
                        __builtin_amdgcn_wave_barrier();
                }
        }
        return total;
}

#ifndef TRI_PSET_WAVES
#define TRI_PSET_WAVES 8 // waves per SIMD the register budget is cut for (512-thread workgroups: four per CU, 33 KB of LDS each)
#endif
// units[]: the TASK_PSET tasks (DevPsetUnit, dev_structs.hpp); order[]: the units in the order they are run (window range by window range);
// ticket: the persistent workgroups' shared cursor into order[].
template <int CODEC>
__global__ __launch_bounds__(PSET_WG, TRI_PSET_WAVES) void k_psets(const DevPsetUnit *__restrict__ units, const uint32_t *__restrict__ order, const uint32_t ntasks,
                                                                   uint32_t *__restrict__ ticket, const uint32_t *__restrict__ qterms, const uint32_t *__restrict__ qplane,
                                                                   uint32_t *__restrict__ out, uint32_t *__restrict__ counts, const uint32_t *__restrict__ masked,
                                                                   const uint32_t *__restrict__ planes, const uint32_t plw, const uint32_t *__restrict__ scat_off,
                                                                   const uint32_t *__restrict__ scat_cnt, const uint32_t *__restrict__ scat_docs) {
        __shared__ PsetShared sh;
        const uint32_t tid = threadIdx.x, lane = tid & 63u;
        const uint32_t wave = uni(tid >> 6);
        // ---- task pipeline: a task's ticket is drawn two tasks ahead and its unit record is fetched one task ahead, by wave 0, while the
        //      workgroup runs the current task — the ticket's atomic and the record's two dependent loads (order[] -> units[]) are off the
        //      critical path.  Lane l of wave 0 carries word l of the 16-word record.
        uint32_t nt = 0xffffffffu; // (wave 0) the ticket after the one in sh.tick[]
        if (wave == 0) {
                const uint32_t t0 = uni(atomicAdd(ticket, 1u)) >> 6; // uniform draw: 64 lanes add 1 each (one +64 atomic), see k_and
                uint32_t wd = 0;
                if (t0 < ntasks)
                        wd = ((const uint32_t *)(units + order[t0]))[lane & 15u];
                ((uint32_t *)&sh.unit[0])[lane & 15u] = wd;
                sh.tick[0] = t0;
                nt = uni(atomicAdd(ticket, 1u)) >> 6;
        }
        __syncthreads();
        uint32_t par = 0; // which of cnt[]'s two halves the next exchange of the waves' counts uses
        for (uint32_t p = 0;; p ^= 1u) {
                if (uni(sh.tick[p]) >= ntasks)
                        break;
                const DevPsetUnit &U = sh.unit[p];
                const uint32_t nterms = uni(U.nterms), w_begin = uni(U.w_begin), w_end = uni(U.w_end), tix = uni(U.tix), term_base = uni(U.term_base);
                const bool as_bitmap = uni(U.first) & PSET_UNIT_BITMAP; // RESULT_BITMAP (dev_structs.hpp): the survivors' words go out as they are
                uint32_t *const qout = out + (((uint64_t)uni((uint32_t)(U.out_off >> 32)) << 32) | uni((uint32_t)U.out_off));
                // (wave 0) the next task's record and the ticket after it: issued now, used when this task is done
                uint32_t nwd = 0, nnt = 0xffffffffu;
                if (wave == 0) {
                        if (nt < ntasks)
                                nwd = ((const uint32_t *)(units + order[nt]))[lane & 15u];
                        nnt = atomicAdd(ticket, 1u);
                }
                uint32_t produced = 0;
                uint32_t scat_at = 0, scat_n = 0; // (a PSET_UNIT_SCATTER task's slice of the scatter list: asked for now, needed when the windows are through)
                if (uni(U.first) & PSET_UNIT_SCATTER)
                        scat_at = uni(scat_off[tix]), scat_n = uni(scat_cnt[tix]);
                const uint32_t round_win = std::max(1u, (uni(U.first) >> PSET_UNIT_ROUND_SHIFT) & PSET_UNIT_ROUND_MASK); // windows per round: the waves' counts cross once a round
                const uint32_t pair_row0 = uni(U.row[0]), pair_row1 = uni(U.row[1]), pair_tt1 = uni(U.tt[1]);
                if (nterms == 2 && pair_row0 != PL_NONE && pair_row1 != PL_NONE) {
                        // ---- two terms with planes — the batch's usual query (cfg2: every k_psets query): the operator is chosen once per task (the second term opens a
                        //      group of its own: AND, or AND-NOT; else OR), both rows' words are requested together (the general loop below waits for a term's two loads
                        //      before it issues the next term's: a round trip per term)
                        const uint32_t *const row_a = planes + (size_t)pair_row0 * plw, *const row_b = planes + (size_t)pair_row1 * plw;
                        const bool pair_or = !(pair_tt1 & QT_GROUP);
                        const uint32_t flip = (pair_tt1 & QT_NOT) ? ~0u : 0u;
                        for (uint32_t wb = w_begin; wb < w_end; wb += round_win, par ^= 1u)
                        produced += psets_round<PsetPairBuf>(
                            sh,
                            [&](const uint32_t word0, PsetPairBuf &buf) {
                                    const uint4 *pa = (const uint4 *)(row_a + word0);
                                    const uint4 *pb = (const uint4 *)(row_b + word0);
                                    buf.a0 = pa[0], buf.a1 = pa[1], buf.b0 = pb[0], buf.b1 = pb[1];
                            },
                            [&](const PsetPairBuf &buf, const uint32_t word0, uint32_t(&acc)[PSET_PER]) {
                                    const uint4 a0 = buf.a0, a1 = buf.a1, b0 = buf.b0, b1 = buf.b1;
                                    if (pair_or) { // (uniform)
                                            acc[0] = a0.x | b0.x, acc[1] = a0.y | b0.y, acc[2] = a0.z | b0.z, acc[3] = a0.w | b0.w;
                                            acc[4] = a1.x | b1.x, acc[5] = a1.y | b1.y, acc[6] = a1.z | b1.z, acc[7] = a1.w | b1.w;
                                    } else {
                                            acc[0] = a0.x & (b0.x ^ flip), acc[1] = a0.y & (b0.y ^ flip), acc[2] = a0.z & (b0.z ^ flip), acc[3] = a0.w & (b0.w ^ flip);
                                            acc[4] = a1.x & (b1.x ^ flip), acc[5] = a1.y & (b1.y ^ flip), acc[6] = a1.z & (b1.z ^ flip), acc[7] = a1.w & (b1.w ^ flip);
                                    }
                                    if (masked) { // masked_documents_registry::test (docidupdates.h:90-119): updated / deleted elsewhere
                                            const uint4 *pm = (const uint4 *)(masked + word0);
                                            const uint4 m0 = pm[0], m1 = pm[1];
                                            acc[0] &= ~m0.x, acc[1] &= ~m0.y, acc[2] &= ~m0.z, acc[3] &= ~m0.w;
                                            acc[4] &= ~m1.x, acc[5] &= ~m1.y, acc[6] &= ~m1.z, acc[7] &= ~m1.w;
                                    }
                            },
                            w_begin, wb, min(w_end, wb + round_win), as_bitmap ? qout : qout + produced, as_bitmap, lane, wave, par);
                } else
                        for (uint32_t wb = w_begin; wb < w_end; wb += round_win, par ^= 1u)
                        produced += psets_round<PsetNoBuf>(
                            sh, [](const uint32_t, PsetNoBuf &) {},
                            [&](const PsetNoBuf &, const uint32_t word0, uint32_t(&acc)[PSET_PER]) {
                                    // ---- the lane's eight words: OR inside a group, AND across groups, AND-NOT for the excluded group
                                    uint32_t grp[PSET_PER];
                                    bool have_acc = false, cur_neg = false;
#pragma unroll
                                    for (uint32_t j = 0; j < PSET_PER; ++j)
                                            acc[j] = grp[j] = 0;
                                    for (uint32_t k = 0; k <= nterms; ++k) {
                                            uint32_t tt = QT_GROUP, row = 0; // (k == nterms: the last group is folded in)
                                            if (k < nterms) {
                                                    if (nterms <= PSET_INLINE_TERMS) {
                                                            tt = uni(U.tt[k]);
                                                            row = uni(U.row[k]);
                                                    } else {
                                                            tt = uni(qterms[term_base + k]);
                                                            row = uni(qplane[term_base + k]);
                                                    }
                                            }
                                            if (k && (tt & QT_GROUP)) {
#pragma unroll
                                                    for (uint32_t j = 0; j < PSET_PER; ++j) {
                                                            acc[j] = !have_acc ? grp[j] : cur_neg ? acc[j] & ~grp[j] : acc[j] & grp[j];
                                                            grp[j] = 0;
                                                    }
                                                    have_acc = true;
                                            }
                                            if (k == nterms)
                                                    break;
                                            if (tt & QT_GROUP)
                                                    cur_neg = tt & QT_NOT;
                                            if (row == PL_NONE) // (a PSET_UNIT_SCATTER union's term without a plane: its documents are set after the windows, below)
                                                    continue;
                                            const uint4 *pa = (const uint4 *)(planes + (size_t)row * plw + word0);
#if defined(TRI_PSET_VARIANT) && TRI_PSET_VARIANT == 3 // (perf probe 3: no plane loads — words made up from the lane's address)
                                            const uint32_t hsh = (word0 * 2654435761u) ^ (k * 40503u);
                                            const uint4 v0 = make_uint4(hsh & (hsh >> 3) & (hsh >> 7), 0, (hsh >> 5) & (hsh << 2) & (hsh >> 11), 0), v1 = make_uint4(0, hsh & 0x10001u, 0, hsh & 0x200u);
#else
                                            const uint4 v0 = pa[0], v1 = pa[1];
#endif
                                            grp[0] |= v0.x, grp[1] |= v0.y, grp[2] |= v0.z, grp[3] |= v0.w;
                                            grp[4] |= v1.x, grp[5] |= v1.y, grp[6] |= v1.z, grp[7] |= v1.w;
                                    }
                                    if (masked) { // masked_documents_registry::test (docidupdates.h:90-119): updated / deleted elsewhere
                                            const uint4 *pm = (const uint4 *)(masked + word0);
                                            const uint4 m0 = pm[0], m1 = pm[1];
                                            acc[0] &= ~m0.x, acc[1] &= ~m0.y, acc[2] &= ~m0.z, acc[3] &= ~m0.w;
                                            acc[4] &= ~m1.x, acc[5] &= ~m1.y, acc[6] &= ~m1.z, acc[7] &= ~m1.w;
                                    }
                            },
                            w_begin, wb, min(w_end, wb + round_win), as_bitmap ? qout : qout + produced, as_bitmap, lane, wave, par);
                if (uni(U.first) & PSET_UNIT_SCATTER) { // (uniform)
                        // a union some of whose terms have NO plane: their documents of this task's windows — k_psets_prep listed them, task by task — go into the words
                        // just stored (still in L2), one atomic OR each; its old value says whether the document is new to the union (the count)
                        __syncthreads(); // (the task's words are stored)
                        const uint32_t off = scat_at, n = scat_n, doc0 = w_begin * SPAN_BITS;
                        for (uint32_t i = tid; i < n; i += PSET_WG) {
                                const uint32_t rel = scat_docs[off + i] - doc0, bit = 1u << (rel & 31u);
                                produced += (atomicOr(&qout[rel >> 5], bit) & bit) ? 0u : 1u; // (per lane: as_bitmap's reduction below adds the lanes up)
                        }
                }
                if (as_bitmap) { // the lanes' counts -> the task's (uniform branch: the record is the workgroup's)
#pragma unroll
                        for (int d = 32; d >= 1; d >>= 1)
                                produced += __shfl_xor(produced, d, 64);
                        sh.cnt[par][wave] = produced; // (cnt[par] was last read two barriers ago)
                        __syncthreads();
                        produced = 0;
#pragma unroll
                        for (uint32_t wv = 0; wv < PSET_WAVES; ++wv)
                                produced += uni(sh.cnt[par][wv]);
                        par ^= 1u;
                }
                if (wave == 0) { // (uniform stores by the lanes of wave 0: no lane-divergent branch next to the loop's barriers)
                        counts[tix] = produced;
                        ((uint32_t *)&sh.unit[p ^ 1u])[lane & 15u] = nwd; // the next task, read by everybody behind the barrier at the loop's head
                        sh.tick[p ^ 1u] = nt;
                        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): the record's words are in LDS before this wave reaches the barrier
                        nt = uni(nnt) >> 6;
                }
                // (the record's barrier stands here, in the block of the LDS stores above, not at the loop's head: reached over the loop's back
                //  edge the compiler put no s_waitcnt lgkmcnt(0) between the stores and the s_barrier — see k_and)
                __syncthreads();
        }
}

// ---- PSET_UNIT_SCATTER, before k_psets runs: the documents of a union's terms WITHOUT a plane, listed task by task (k_psets_prep).  One workgroup per query (k_psets_prep_list
//      names the queries' first units): lane k looks after term k (all the terms' records side by side),
//      then a lane per directory row of <= 32 documents (the register row readers of k_fused) — counted per task in LDS, the tasks' places settled by one scan and one
//      draw from the batch's cursor, then decoded once more into their places.  scat_off[tix] / scat_cnt[tix]: a task's slice of scat_docs[].
//      Until round 6 every TASK of such a query did the lookups for its own docID range inside k_psets, between two barriers: a rare term's two or three rows each span
//      millions of docIDs, so each of the query's twenty tasks went term record -> block directory -> block bytes (five dependent round trips, one lane in 512 working),
//      decoded the straddling row and kept next to nothing of it — 0.39 of k_psets' 0.96 ms for cfg5's five-way unions, for 0.06 % of their matches.
constexpr int PSCAT_WG = 256;
constexpr uint32_t PSCAT_MAX_TASKS = 4096; // tasks of a query: 2^31 documents / (PSET_TASK_WINDOWS windows of 2^17)
static_assert((1ull << 31) / ((uint64_t)PSET_TASK_WINDOWS * SPAN_BITS) <= PSCAT_MAX_TASKS, "a query's tasks fit the workgroup's LDS counters");
struct PscatShared {
        DevTerm term[MAX_QTERMS];   // the union's terms without a plane ...
        uint32_t nrows[MAX_QTERMS]; // ... and their rows (0: the term has a plane)
        uint32_t cnt[PSCAT_MAX_TASKS]; // documents per task, then (second pass) the task's cursor into scat_docs[]
        uint32_t red[PSCAT_WG / 64 + 1];
};
struct PscatPost { // pass 1 (place == nullptr): count per task; pass 2: into the task's place
        uint32_t *cnt;
        const uint32_t *masked;
        uint32_t *place;
        uint32_t nbits, task_docs, cap;
        __device__ __forceinline__ void doc(const uint32_t d) {
                if (d >= nbits || (masked && ((masked[d >> 5] >> (d & 31u)) & 1u))) // masked_documents_registry::test (docidupdates.h:90-119)
                        return;
                const uint32_t at = atomicAdd(&cnt[d / task_docs], 1u);
                if (place && at < cap) // (a task that did not fit the list — the planner's bound holds: belt and braces — has its cursor at cap)
                        place[at] = d;
        }
        __device__ __forceinline__ void operator()(const uint32_t d, const uint32_t) { doc(d); }
};
// the scatter queries' first units, in any order: k_psets_prep's work list (a grid over ALL the units — twenty a query, a million for the 100 K batch — whose workgroups
// left at once unless theirs was such a unit took 5.1 ms there, longer than the k_and_dense it runs beside)
__global__ __launch_bounds__(256) void k_psets_prep_list(const DevPsetUnit *__restrict__ units, const uint32_t nunits, uint32_t *__restrict__ cursor, uint32_t *__restrict__ list,
                                                         const uint32_t cap) {
        const uint32_t u = blockIdx.x * 256u + threadIdx.x;
        if (u >= nunits || !(units[u].first & PSET_UNIT_SCATTER) || units[u].w_begin != 0)
                return;
        const uint32_t at = atomicAdd(cursor, 1u);
        if (at < cap)
                list[at] = u;
}
template <int CODEC>
__global__ __launch_bounds__(PSCAT_WG) void k_psets_prep(const DevPsetUnit *__restrict__ units, const uint32_t *__restrict__ list, const uint32_t *__restrict__ listed, const DevQuery *__restrict__ plan, const DevTask *__restrict__ tasks,
                                                         const uint32_t *__restrict__ qterms, const uint32_t *__restrict__ qplane, const uint32_t *__restrict__ masked,
                                                         const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last, const uint32_t *__restrict__ blk_off,
                                                         const uint4 *__restrict__ blk_rec, const uint32_t *__restrict__ blk_doff, const DevTerm *__restrict__ terms,
                                                         uint32_t *__restrict__ cursor, uint32_t *__restrict__ scat_off, uint32_t *__restrict__ scat_cnt,
                                                         uint32_t *__restrict__ scat_docs, const uint32_t scat_cap) {
        __shared__ PscatShared sh;
        const uint32_t tid = threadIdx.x;
        if (blockIdx.x >= uni(*listed)) // (grid: the planner's count of scatter queries == the units k_psets_prep_list found; belt and braces)
                return;
        const DevPsetUnit &U = units[uni(list[blockIdx.x])];
        const uint32_t nterms = uni(U.nterms), term_base = uni(U.term_base), tix = uni(U.tix);
        if (tid < MAX_QTERMS) {
                uint32_t n = 0;
                if (tid < nterms && qplane[term_base + tid] == PL_NONE) {
                        const DevTerm t = terms[qterms[term_base + tid] & QT_TERM];
                        sh.term[tid] = t;
                        n = t.nblocks;
                }
                sh.nrows[tid] = n;
        }
        const DevQuery &q = plan[uni(tasks[tix].slot)];
        const uint32_t ntasks = min(uni(q.ntasks), PSCAT_MAX_TASKS), nbits = uni(q.out_cap) * 32u, task_docs = (uni(U.w_end) - uni(U.w_begin)) * SPAN_BITS;
        for (uint32_t t = tid; t < ntasks; t += PSCAT_WG)
                sh.cnt[t] = 0;
        __syncthreads();
        uint32_t total = 0;
        for (uint32_t k = 0; k < nterms; ++k)
                total += uni(sh.nrows[k]);
        for (uint32_t pass = 0; pass < 2; ++pass) {
                PscatPost post{sh.cnt, masked, pass ? scat_docs : nullptr, nbits, task_docs, scat_cap};
                for (uint32_t v = tid; v < total; v += PSCAT_WG) { // one lane per row, the terms' rows one after the other
                        uint32_t k = 0, b = v;
                        for (; b >= sh.nrows[k]; ++k)
                                b -= sh.nrows[k];
                        const DevTerm t = sh.term[k];
                        const uint32_t *bl = blk_last + t.first_block;
                        const uint32_t prev = b ? bl[b - 1] : 0, last = bl[b];
#ifdef TRI_PROF
                        ProfClock prof_;
#endif
                        if (CODEC == CODEC_LUCENE) {
                                const uint4 rec = blk_rec[t.first_block + b];
                                row_decode<CODEC, false, PscatPost>(index, t, b, rec.x, rec.y, rec.z, rec.w, TRI_BLOCK_N(t, b, index, 0), prev, last, 0u, post PROF_PASS);
                        } else {
                                const uint32_t off = blk_off[t.first_block + b];
                                const uint32_t dlen = blk_doff[t.first_block + b + 1] - blk_doff[t.first_block + b] - 1u;
                                row_decode<CODEC, false, PscatPost>(index, t, b, off, dlen, 0, 0, TRI_BLOCK_N(t, b, index, off), prev, last, 0u, post PROF_PASS);
                        }
                }
                __syncthreads();
                if (pass)
                        break;
                // ---- the tasks' places: an exclusive prefix over cnt[0 .. ntasks) — every thread a stretch of it —, based at one draw from the batch's cursor
                const uint32_t per = (ntasks + PSCAT_WG - 1) / PSCAT_WG, t0 = tid * per, t1 = min(ntasks, t0 + per);
                uint32_t mine = 0;
                for (uint32_t t = t0; t < t1; ++t)
                        mine += sh.cnt[t];
                uint32_t wtot;
                uint32_t ex = wave_excl_scan(mine, wtot);
                if ((tid & 63u) == 0)
                        sh.red[tid >> 6] = wtot;
                __syncthreads();
                uint32_t before = 0, all = 0;
                for (uint32_t w = 0; w < PSCAT_WG / 64; ++w) {
                        before += w < (tid >> 6) ? sh.red[w] : 0u;
                        all += sh.red[w];
                }
                if (tid == 0)
                        sh.red[PSCAT_WG / 64] = atomicAdd(cursor, all);
                __syncthreads();
                const uint32_t base = uni(sh.red[PSCAT_WG / 64]);
                uint32_t at = base + before + ex;
                for (uint32_t t = t0; t < t1; ++t) {
                        const uint32_t c = sh.cnt[t];
                        const bool fits = at + c <= scat_cap; // (the planner's bound holds: belt and braces against a store outside the list)
                        scat_off[tix + t] = fits ? at : 0u;
                        scat_cnt[tix + t] = fits ? c : 0u;
                        sh.cnt[t] = fits ? at : scat_cap; // the task's cursor (a task that does not fit: the second pass stores nothing for it)
                        at += c;
                }
                __syncthreads();
        }
}
