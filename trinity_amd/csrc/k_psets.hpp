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
// dozen cache lines (r03: 11.8 us per window and workgroup).  Here:
//   * a wave owns 512 consecutive words (16384 documents) of a step — two 16-byte loads per lane and term —, keeps the survivors in
//     registers, and only its COUNT crosses to the other waves: ONE barrier per step (double-buffered counts) instead of six;
//   * the wave expands its survivors into a private LDS staging buffer (scattered 4-byte LDS writes are cheap) and copies the buffer out
//     with coalesced stores: a store instruction covers 256 contiguous bytes;
//   * a wave whose sub-window holds more survivors than the staging buffer (a union of head terms: one document in 16 or denser) walks
//     its words one lane per BIT — ballot, rank by mbcnt, one coalesced store per 64 bits;
//   * a task is ONE 64-byte record (DevPsetUnit) fetched a task ahead by wave 0 together with the ticket after it, instead of the
//     sched -> task -> query -> qterms / qplane chain of dependent loads; tasks are two windows of one query, and the schedule runs them
//     window range by window range, so that a range's plane words are in the XCDs' L2 while every query that reads them is in flight.
// A task has a private, bound-allocated output region (planner.hpp: the same layout as TASK_DENSE, so k_score / k_rich / k_phrase / the
// result read-back see no difference).
// Measured (cfg2, 1551 queries, 119 K windows, 380 M docIDs): 1.05 ms.  Probe builds said where it goes: without the
// expansion 0.59 ms, without the plane loads 0.58 ms (half the matches), without the copy-out 0.92 ms; the phase clocks: 39 % at the first
// use of the plane words, 19 % at the barrier, 16 % expanding, 13 % counting.  About 400 wave instructions per sub-window, 250 of them the
// expansion's count-trailing-zeros loops (eight words per lane, a loop's trips = the wave's densest word: 20 % of the lane-iterations
// extract a bit) — 0.6 ms of issue slots by themselves.  Tried on the way and not kept (each correct, each measured): the next window's
// words requested before the expansion (1.03 - 1.13 ms: 16 more registers, spills at 8 waves per SIMD, nothing gained at 6); 256- and
// 128-thread workgroups (1.09 / 1.30 ms); no workgroup synchronisation at all — every wave an item on its own, the counts published in
// global cells and read back by the later sub-windows of the task (1.37 ms: two more memory round trips on every wave's critical path).
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
static_assert(PSET_STAGE >= PSET_WORDS, "the dense walk parks the wave's words in its staging buffer");

struct PsetScatterShared { // (part of PsetShared)
        DevTerm term[MAX_QTERMS]; // the union's terms without a plane ...
        uint32_t b_lo[MAX_QTERMS], nrows[MAX_QTERMS]; // ... their rows that can reach the task's range
};
struct PsetShared {
        uint32_t stage[PSET_WAVES][PSET_STAGE];
        uint32_t cnt[2][PSET_WAVES]; // per window parity: the waves' survivor counts
        PsetScatterShared scatter;   // PSET_UNIT_SCATTER tasks: the terms without a plane
        DevPsetUnit unit[2];         // the task being run and the next one (fetched while the current one runs)
        uint32_t tick[2];            // ... and their tickets (>= ntasks: none)
};

// ---- PSET_UNIT_SCATTER: a union's terms WITHOUT a plane, after the plane terms' words of the task's windows have been stored: every row of such a
//      term that can reach the task's docID range is decoded, one lane per row of <= 32 documents (the register row readers of k_fused), and its
//      documents are set in the stored words one by one — an atomic OR whose old value says whether the document is new to the union (the count).
//      A rare term brings a handful of documents per task; k_and_dense decoded every list of such a query into an LDS window bitmap, window by
//      window, behind half a dozen barriers each.  Not inlined: the row readers' registers must not weigh on the windows' loop.
struct PsetScatterPost {
        uint32_t *bm;               // the task's words (bit 0 of word 0: the task's first document)
        const uint32_t *masked;     // masked documents (absolute docIDs), or nullptr
        uint32_t doc0, nbits, added = 0;
        __device__ __forceinline__ void doc(const uint32_t rel) {
                if (rel >= nbits) // (a row reaches across the range's ends: the neighbouring tasks take those documents)
                        return;
                const uint32_t d = doc0 + rel, bit = 1u << (rel & 31u);
                if (masked && ((masked[d >> 5] >> (d & 31u)) & 1u))
                        return;
                added += (atomicOr(&bm[rel >> 5], bit) & bit) ? 0u : 1u;
        }
        __device__ __forceinline__ void operator()(const uint32_t rel, const uint32_t) { doc(rel); }
};
template <int CODEC>
__device__ __noinline__ uint32_t psets_scatter(PsetScatterShared &ss, const uint8_t *__restrict__ index, const uint32_t *__restrict__ blk_last, const uint32_t *__restrict__ blk_off,
                                               const uint4 *__restrict__ blk_rec, const uint32_t *__restrict__ blk_doff, const uint32_t *__restrict__ win,
                                               const DevTerm *__restrict__ terms, const uint32_t *__restrict__ qterms, const uint32_t *__restrict__ qplane, const uint32_t nterms,
                                               const uint32_t w_begin, const uint32_t w_end, uint32_t *__restrict__ bm, const uint32_t *__restrict__ masked) {
        const uint32_t tid = threadIdx.x;
        const uint32_t d0 = w_begin * SPAN_BITS, d1 = w_end * SPAN_BITS; // (the planner keeps max docID below 2^31: no wrap)
        // lane k looks after term k: its record and its rows that can hold documents of [d0, d1) — first row whose last docID >= d0 ... first row
        // whose last docID >= d1 (it may still begin inside) — all the terms side by side: one chain of dependent loads for the task, not one per term
        if (tid < MAX_QTERMS) {
                uint32_t lo_b = 0, n = 0;
                if (tid < nterms && qplane[tid] == PL_NONE) {
                        const DevTerm t = terms[qterms[tid] & QT_TERM];
                        ss.term[tid] = t;
                        if (t.nblocks) {
                                const uint32_t *bl = blk_last + t.first_block;
                                uint32_t b_lo, b_hi;
                                if (t.win_off != 0xffffffffu) {
                                        b_lo = win[t.win_off + w_begin * CELLS_PER_SPAN];
                                        b_hi = win[t.win_off + w_end * CELLS_PER_SPAN];
                                } else {
                                        uint32_t lo = 0, hi = t.nblocks;
                                        while (lo < hi) {
                                                const uint32_t mid = (lo + hi) >> 1;
                                                if (bl[mid] < d0)
                                                        lo = mid + 1;
                                                else
                                                        hi = mid;
                                        }
                                        b_lo = lo;
                                        hi = t.nblocks;
                                        while (lo < hi) {
                                                const uint32_t mid = (lo + hi) >> 1;
                                                if (bl[mid] < d1)
                                                        lo = mid + 1;
                                                else
                                                        hi = mid;
                                        }
                                        b_hi = lo;
                                }
                                b_hi = min(b_hi, t.nblocks - 1);
                                lo_b = b_lo;
                                n = b_lo < t.nblocks ? b_hi - b_lo + 1 : 0u;
                        }
                }
                ss.b_lo[tid] = lo_b;
                ss.nrows[tid] = n;
        }
        __syncthreads();
        uint32_t total = 0;
        for (uint32_t k = 0; k < nterms; ++k)
                total += uni(ss.nrows[k]);
        PsetScatterPost post{bm, masked, d0, d1 - d0};
        for (uint32_t v = tid; v < total; v += PSET_WG) { // one lane per row, the terms' rows one after the other
                uint32_t k = 0, r = v;
                for (; r >= ss.nrows[k]; ++k)
                        r -= ss.nrows[k];
                const DevTerm t = ss.term[k];
                const uint32_t b = ss.b_lo[k] + r;
                const uint32_t *bl = blk_last + t.first_block;
                const uint32_t prev = b ? bl[b - 1] : 0, last = bl[b];
#ifdef TRI_PROF
                ProfClock prof_;
#endif
                if (CODEC == CODEC_LUCENE) {
                        const uint4 rec = blk_rec[t.first_block + b];
                        row_decode<CODEC, false, PsetScatterPost>(index, t, b, rec.x, rec.y, rec.z, rec.w, TRI_BLOCK_N(t, b, index, 0), prev, last, d0, post PROF_PASS);
                } else {
                        const uint32_t off = blk_off[t.first_block + b];
                        const uint32_t dlen = blk_doff[t.first_block + b + 1] - blk_doff[t.first_block + b] - 1u;
                        row_decode<CODEC, false, PsetScatterPost>(index, t, b, off, dlen, 0, 0, TRI_BLOCK_N(t, b, index, off), prev, last, d0, post PROF_PASS);
                }
        }
        __syncthreads(); // (ss is the next task's, too)
        return post.added;
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
                                                                   const uint32_t *__restrict__ planes, const uint32_t plw, const uint8_t *__restrict__ index,
                                                                   const uint32_t *__restrict__ blk_last, const uint32_t *__restrict__ blk_off, const uint4 *__restrict__ blk_rec,
                                                                   const uint32_t *__restrict__ blk_doff, const uint32_t *__restrict__ win, const DevTerm *__restrict__ terms) {
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
                uint32_t produced = 0, par = 0;
                const uint32_t pair_row0 = uni(U.row[0]), pair_row1 = uni(U.row[1]), pair_tt1 = uni(U.tt[1]);
                const bool pair = nterms == 2 && pair_row0 != PL_NONE && pair_row1 != PL_NONE;
                const bool pair_and = pair_tt1 & QT_GROUP, pair_not = pair_tt1 & QT_NOT; // (the second term opens a group of its own: AND, or AND-NOT)
                for (uint32_t sw = w_begin * SPAN_WORDS; sw < w_end * SPAN_WORDS; sw += PSET_STEP_WORDS, par ^= 1u) {
                        const uint32_t word0 = sw + tid * PSET_PER; // this lane's first word of the step
                        // ---- the window's survivors, this lane's eight words: OR inside a group, AND across groups, AND-NOT for the excluded group
                        uint32_t acc[PSET_PER], grp[PSET_PER];
                        bool have_acc = false, cur_neg = false;
#pragma unroll
                        for (uint32_t j = 0; j < PSET_PER; ++j)
                                acc[j] = grp[j] = 0;
                        if (pair) {
                                // two terms with planes — the batch's usual query: both terms' words travel together (the general loop below waits for a
                                // term's two loads before it issues the next term's: a round trip per term and step)
                                const uint4 *pa = (const uint4 *)(planes + (size_t)pair_row0 * plw + word0);
                                const uint4 *pb = (const uint4 *)(planes + (size_t)pair_row1 * plw + word0);
                                const uint4 a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
                                const uint32_t av[PSET_PER] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, bv[PSET_PER] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                                for (uint32_t j = 0; j < PSET_PER; ++j)
                                        acc[j] = !pair_and ? av[j] | bv[j] : pair_not ? av[j] & ~bv[j] : av[j] & bv[j];
                        } else
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
                        if (as_bitmap) { // nothing to expand, nothing to rank: two 16-byte stores per lane, the counts summed at the task's end
                                uint4 *o = (uint4 *)(qout + (word0 - w_begin * SPAN_WORDS));
                                o[0] = make_uint4(acc[0], acc[1], acc[2], acc[3]);
                                o[1] = make_uint4(acc[4], acc[5], acc[6], acc[7]);
#pragma unroll
                                for (uint32_t j = 0; j < PSET_PER; ++j)
                                        produced += (uint32_t)__popc(acc[j]); // (per lane here; reduced below)
                                continue;
                        }
                        // ---- counts: lane -> wave (shuffles) -> workgroup (one LDS word per wave, one barrier)
                        uint32_t c = 0;
#pragma unroll
                        for (uint32_t j = 0; j < PSET_PER; ++j)
                                c += (uint32_t)__popc(acc[j]);
                        uint32_t T;
                        const uint32_t ex = wave_excl_scan(c, T);
                        T = uni(T);
                        sh.cnt[par][wave] = T; // (the same value from every lane of the wave)
                        __syncthreads();
                        uint32_t base = produced, tot = 0;
#pragma unroll
                        for (uint32_t wv = 0; wv < PSET_WAVES; ++wv) {
                                const uint32_t x = uni(sh.cnt[par][wv]);
                                base += wv < wave ? x : 0u;
                                tot += x;
                        }
                        produced += tot;
                        if (!T)
                                continue; // (wave-uniform; the barrier above is the window's only one)
#if defined(TRI_PSET_VARIANT) && TRI_PSET_VARIANT == 1 // (perf probe: counts only)
                        continue;
#endif
                        uint32_t *const st = sh.stage[wave];
                        if (T <= PSET_STAGE) {
                                // ---- sparse: every lane writes its words' docIDs into the wave's staging buffer at its rank, then the wave copies the
                                //      buffer out — 64 consecutive docIDs per store instruction
                                uint32_t o = ex;
#pragma unroll
                                for (uint32_t j = 0; j < PSET_PER; ++j) {
                                        uint32_t m = acc[j];
                                        const uint32_t b0 = (word0 + j) << 5;
                                        while (m) {
                                                st[o++] = b0 + (uint32_t)__builtin_ctz(m);
                                                m &= m - 1u;
                                        }
                                }
                                __builtin_amdgcn_wave_barrier();
#if !defined(TRI_PSET_VARIANT) || TRI_PSET_VARIANT != 2 // (perf probe 2: no copy-out)
                                for (uint32_t i = lane; i < T; i += 64u)
                                        qout[base + i] = st[i];
#endif
                                __builtin_amdgcn_wave_barrier(); // (the next window's staging writes stay behind these reads)
                        } else {
                                // ---- dense (a union of head terms): the wave's 512 words parked in LDS, then one lane per BIT, 64 bits a step:
                                //      ballot, rank by mbcnt, one coalesced store
#pragma unroll
                                for (uint32_t j = 0; j < PSET_PER; ++j)
                                        st[lane * PSET_PER + j] = acc[j];
                                __builtin_amdgcn_wave_barrier();
                                uint32_t o = base;
                                const uint32_t wbase = sw + wave * PSET_WORDS;
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
                if (uni(U.first) & PSET_UNIT_SCATTER) { // (uniform)
                        __syncthreads(); // the task's words are stored: the documents of the terms without a plane go in on top of them
                        produced += psets_scatter<CODEC>(sh.scatter, index, blk_last, blk_off, blk_rec, blk_doff, win, terms, qterms + term_base, qplane + term_base, nterms, w_begin, w_end, qout, masked);
                }
                if (as_bitmap) { // the lanes' counts -> the task's (uniform branch: the record is the workgroup's)
#pragma unroll
                        for (int d = 32; d >= 1; d >>= 1)
                                produced += __shfl_xor(produced, d, 64);
                        __syncthreads(); // (cnt[] of the previous task's last step has been read by everybody)
                        sh.cnt[0][wave] = produced;
                        __syncthreads();
                        produced = 0;
#pragma unroll
                        for (uint32_t wv = 0; wv < PSET_WAVES; ++wv)
                                produced += uni(sh.cnt[0][wv]);
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
