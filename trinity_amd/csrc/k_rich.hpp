// k_rich.hpp — the default ("rich match") execution mode: per match, the query terms that matched it and their hits
// Part of libtrinity_hip.so (MI355X / gfx950); included by trinity_hip.hip.  New code, no reference source.
#pragma once
#include "k_phrase.hpp"

// exec_query without DocumentsOnly / AccumulatedScoreScheme hands every match to consider(const matched_document &) with
// matchedTerms[] = the query's postings iterators that sit on the document (collect_doc_matching_terms,
// queryexec_ctx.cpp:382-520: every term of a conjunction or phrase, the members of an OR that hold the document, nothing
// from the excluded side of a NOT) and, per term, term_hits{freq, all[]} materialised by prepare_match (:522-648).
//
// Here the matching kernels have produced every query's ascending match list.  k_rich walks each task's segment in tiles of
// RICH_TILE matches held in LDS and, for every distinct reportable term of the query (DevQuery::score_base/nscore in this
// mode), visits the term's blocks that can hold a match of the tile exactly as k_score does (block-driven for dense results,
// match-driven for sparse ones; every document gallops through the tile's matches).  Two passes, because the hit pool is
// packed:
//   COUNT  present[slot] |= 1 << k, freq[slot * R + k] = the term's frequency in the document; the task's hit total
//   WRITE  (after the host has turned the tasks' totals into pool offsets) positions of (match m, term k) go to
//          pool[task_base + sum of freq over (m' < m, all k') + sum of freq over (m, k' < k)] — match-major, term-minor, so a
//          task's (and therefore a query's) hits are one contiguous run and the offsets follow from the freqs alone
// slot = task.out_off + index in the segment; R = the batch's widest reportable-term count.
constexpr uint32_t RICH_TILE = 2048;

struct RichShared {
        uint32_t cand[RICH_TILE];
        uint32_t rowoff[RICH_TILE]; // WRITE: hits of the task before match m
        uint16_t mptr[32][AND_WG];  // per lane (column): the tile indices of the matches its block coincides with
        uint32_t blkof[AND_WG + 1];
        uint32_t scan[8];
        uint32_t bcast[4];
        uint32_t hits; // COUNT: hits of the task so far
};

template <int CODEC, bool WRITE>
__global__ __launch_bounds__(AND_WG) void k_rich(const uint8_t *__restrict__ index, const uint8_t *__restrict__ hits, const uint32_t *__restrict__ blk_hits,
                                                 const uint32_t *__restrict__ hdir, const uint32_t *__restrict__ blk_last, const uint32_t *__restrict__ blk_off,
                                                 const DevTerm *__restrict__ terms, const DevQuery *__restrict__ plan, const DevTask *__restrict__ tasks,
                                                 const uint32_t *__restrict__ sched, const uint32_t *__restrict__ rterms, const uint32_t ntasks,
                                                 uint32_t *__restrict__ ticket, const uint32_t *__restrict__ out, const uint32_t *__restrict__ counts, const uint32_t R,
                                                 uint32_t *__restrict__ present, uint16_t *__restrict__ freq, uint32_t *__restrict__ task_hits,
                                                 const uint64_t *__restrict__ task_pos_base, uint16_t *__restrict__ pool, const uint32_t *__restrict__ allow,
                                                 uint8_t *__restrict__ pool_plen, uint64_t *__restrict__ pool_payload) {
        // pool_plen / pool_payload (or null; TRI_FLAG_HIT_PAYLOADS): per hit, parallel to `pool`, term_hit::payloadLen and ::payload as
        // Google::Decoder::materialize_hits leaves them (google_codec.cpp:533-594) — the payload word is carried from hit to hit within a
        // document and only its first payloadLen bytes are rewritten, exactly as the reference's local variable is
        // allow (or null): per match, the reportable terms an iterator of the query tree sits on — general trees, where holding a term
        // is not enough (the matching kernel left the mask; every other query's matches carry all ones)
        __shared__ RichShared sh;
        const uint32_t tid = threadIdx.x;
        const uint32_t wave = uni(tid >> 6);
        const HitCtx ctx{CODEC == CODEC_GOOGLE ? index : hits, blk_hits, hdir};
        for (;;) {
                if (wave == 0) {
                        const uint32_t old = atomicAdd(ticket, 1u);
                        sh.bcast[0] = uni(old) >> 6;
                }
                __syncthreads();
                const uint32_t ticket_no = uni(sh.bcast[0]);
                __syncthreads();
                if (ticket_no >= ntasks)
                        break;
                const uint32_t tix = sched[ticket_no];
                const DevTask task = tasks[tix];
                const DevQuery q = plan[task.slot];
                const uint32_t M = counts[tix];
                const uint32_t *seg = out + task.out_off;
                sh.hits = 0;
                uint32_t tile_base = 0; // WRITE: hits of the task before this tile
                __syncthreads();
                for (uint32_t tb = 0; tb < M; tb += RICH_TILE) {
                        const uint32_t C = min(RICH_TILE, M - tb);
                        for (uint32_t j = tid; j < C; j += AND_WG)
                                sh.cand[j] = seg[tb + j];
                        if (WRITE) {
                                // exclusive scan over the tile of each match's hit count (its freq row, written by the COUNT pass)
                                uint32_t run = 0;
                                constexpr uint32_t PER = RICH_TILE / AND_WG;
                                uint32_t rowsum[PER];
#pragma unroll
                                for (uint32_t i = 0; i < PER; ++i) {
                                        const uint32_t j = tid * PER + i;
                                        uint32_t f = 0;
                                        if (j < C) {
                                                const uint64_t row = ((uint64_t)task.out_off + tb + j) * R;
                                                for (uint32_t k = 0; k < q.nscore; ++k)
                                                        f += freq[row + k];
                                        }
                                        rowsum[i] = run;
                                        run += f;
                                }
                                uint32_t wtot;
                                const uint32_t ex = wave_excl_scan(run, wtot);
                                sh.scan[tid >> 6] = wtot;
                                __syncthreads();
                                uint32_t wbase = 0, total = 0;
                                for (int w = 0; w < AND_WG / 64; ++w) {
                                        if (w < (int)(tid >> 6))
                                                wbase += sh.scan[w];
                                        total += sh.scan[w];
                                }
#pragma unroll
                                for (uint32_t i = 0; i < PER; ++i)
                                        sh.rowoff[tid * PER + i] = tile_base + wbase + ex + rowsum[i];
                                tile_base += uni(total);
                        }
                        __syncthreads();
                        for (uint32_t ti = 0; ti < q.nscore; ++ti) {
                                const uint32_t term = rterms[q.score_base + ti];
                                const DevTerm t = terms[term];
                                const uint32_t *bl = blk_last + t.first_block;
                                const uint32_t *bo = blk_off + t.first_block;
                                // one block of the term against the matches from index j on (see k_score): deltas mark the coinciding
                                // slots and remember their matches, then freqs (COUNT) or freqs + hits in lockstep (WRITE)
                                auto walk = [&](const uint32_t bj, const uint32_t j, uint32_t cv) {
                                        const uint32_t prev = bj ? bl[bj - 1] : 0;
                                        const uint32_t last = bl[bj];
                                        const uint32_t off = bo[bj];
                                        const uint32_t n = TRI_BLOCK_N(t, bj, index, off);
                                        DeltaStream<CODEC> s;
                                        s.init(index, t, bj, off);
                                        uint32_t doc = prev, ptr = j, mask = 0, nm = 0;
                                        for (uint32_t i = 0; i < n; ++i) {
                                                doc = (i + 1 < n) ? doc + s.next() : last;
                                                if (cv < doc) {
                                                        uint32_t step = 1, lo = ptr + 1;
                                                        while (lo + step <= C && sh.cand[lo + step - 1] < doc) {
                                                                lo += step;
                                                                step <<= 1;
                                                        }
                                                        uint32_t hi = min(lo + step - 1, C);
                                                        while (lo < hi) {
                                                                const uint32_t mid = (lo + hi) >> 1;
                                                                if (sh.cand[mid] < doc)
                                                                        lo = mid + 1;
                                                                else
                                                                        hi = mid;
                                                        }
                                                        ptr = lo;
                                                        cv = ptr < C ? sh.cand[ptr] : 0xffffffffu;
                                                }
                                                if (cv == doc && (!allow || ((allow[(uint64_t)task.out_off + tb + ptr] >> ti) & 1u))) {
                                                        mask |= 1u << i;
                                                        sh.mptr[nm++][tid] = (uint16_t)ptr;
                                                }
                                        }
                                        if (!mask)
                                                return;
                                        FreqStream<CODEC> fs;
                                        fs.init(index, t, bj, off, s);
                                        nm = 0;
                                        if (!WRITE) {
                                                uint32_t sum = 0;
                                                for (uint32_t i = 0; i < n && (mask >> i); ++i) {
                                                        const uint32_t f = fs.next() & HitStream<CODEC>::FREQ_MASK & 0xffffu; // term_hits::freq is tokenpos_t
                                                        if ((mask >> i) & 1u) {
                                                                const uint64_t slot = (uint64_t)task.out_off + tb + sh.mptr[nm++][tid];
                                                                atomicOr(&present[slot], 1u << ti);
                                                                freq[slot * R + ti] = (uint16_t)f;
                                                                sum += f;
                                                        }
                                                }
                                                atomicAdd(&sh.hits, sum);
                                                return;
                                        }
                                        // WRITE: where this (match, term) run starts = the task's base + the match's row offset + the freqs of
                                        // the row's earlier terms
                                        auto dest = [&](const uint32_t m) -> uint16_t * {
                                                const uint64_t row = ((uint64_t)task.out_off + tb + m) * R;
                                                uint32_t before = 0;
                                                for (uint32_t k = 0; k < ti; ++k)
                                                        before += freq[row + k];
                                                return pool + task_pos_base[tix] + sh.rowoff[m] + before;
                                        };
                                        if constexpr (CODEC == CODEC_GOOGLE) {
                                                // the hits follow the n freqs in the same byte stream (google_codec.cpp:533-594)
                                                VbStream hs;
                                                hs.init(index + off + (ctx.blk_hits[t.first_block + bj] & ~BLK_HITS_PLAIN)); // where the block's hits start (directory: bytes past the block's payload offset)
                                                for (uint32_t i = 0; i < n && (mask >> i); ++i) {
                                                        const uint32_t f = fs.next();
                                                        const uint32_t fw = f & 0xffffu; // what the COUNT pass recorded (term_hits::freq is tokenpos_t)
                                                        uint16_t *d = ((mask >> i) & 1u) ? dest(sh.mptr[nm++][tid]) : nullptr;
                                                        uint32_t pos = 0, plen = 0; // position and payload-length state restart with every document
                                                        uint64_t payload = 0;
                                                        for (uint32_t h = 0; h < f; ++h) {
                                                                const uint32_t v = hs.next();
                                                                if (v & 1u)
                                                                        plen = hs.byte();
                                                                if (pool_payload && d) { // the payload bytes, little end first, over the word's low bytes
                                                                        if (!plen)
                                                                                payload = 0;
                                                                        for (uint32_t k = 0; k < plen; ++k) {
                                                                                const uint64_t by = hs.byte();
                                                                                if (k < 8)
                                                                                        payload = (payload & ~(0xffull << (8 * k))) | (by << (8 * k));
                                                                        }
                                                                } else
                                                                        hs.skip(plen);
                                                                pos = (pos + (v >> 1)) & 0xffffu;
                                                                if (d && h < fw) {
                                                                        d[h] = (uint16_t)pos;
                                                                        if (pool_payload) {
                                                                                const uint64_t at = (uint64_t)(d + h - pool);
                                                                                pool_plen[at] = (uint8_t)plen;
                                                                                pool_payload[at] = payload;
                                                                        }
                                                                }
                                                        }
                                                }
                                        } else {
                                                uint32_t h0 = ctx.blk_hits[t.first_block + bj];
                                                for (uint32_t i = 0; i < n && (mask >> i); ++i) {
                                                        const uint32_t f = fs.next();
                                                        if ((mask >> i) & 1u) {
                                                                uint16_t *d = dest(sh.mptr[nm++][tid]);
                                                                HitStream<CODEC> hs;
                                                                hs.init(ctx, t.pad, h0);
                                                                uint32_t pos = 0;
                                                                for (uint32_t h = 0; h < (f & 0xffffu); ++h) {
                                                                        pos = (pos + hs.next()) & 0xffffu;
                                                                        d[h] = (uint16_t)pos;
                                                                        if (pool_payload) { // (the Lucene-shaped segments of this repo carry no payloads)
                                                                                pool_plen[(uint64_t)(d + h - pool)] = 0;
                                                                                pool_payload[(uint64_t)(d + h - pool)] = 0;
                                                                        }
                                                                }
                                                        }
                                                        h0 += f;
                                                }
                                        }
                                };
                                // the term's blocks that can hold a match of this tile
                                const uint32_t cmin = sh.cand[0], cmax = sh.cand[C - 1];
                                const uint32_t b0 = wg_lower_bound<AND_WG>(sh.scan, bl, t.nblocks, cmin);
                                uint32_t b1 = b0;
                                if (b0 < t.nblocks) {
                                        b1 = b0 + wg_lower_bound<AND_WG>(sh.scan, bl + b0, t.nblocks - b0, cmax);
                                        if (b1 >= t.nblocks)
                                                b1 = t.nblocks - 1;
                                }
                                __syncthreads();
                                if (b0 < t.nblocks && b1 - b0 + 1 <= C) {
                                        for (uint32_t b = b0 + tid; b <= b1; b += AND_WG) {
                                                const uint32_t prev = b ? bl[b - 1] : 0;
                                                uint32_t lo = 0, hi = C;
                                                while (lo < hi) {
                                                        const uint32_t mid = (lo + hi) >> 1;
                                                        if (sh.cand[mid] <= prev)
                                                                lo = mid + 1;
                                                        else
                                                                hi = mid;
                                                }
                                                if (lo < C && sh.cand[lo] <= bl[b])
                                                        walk(b, lo, sh.cand[lo]);
                                        }
                                        __syncthreads();
                                } else if (b0 < t.nblocks) {
                                        sh.blkof[0] = 0xffffffffu;
                                        __syncthreads();
                                        for (uint32_t base = 0; base < C; base += AND_WG) {
                                                const uint32_t j = base + tid;
                                                uint32_t bj = 0xffffffffu, cv = 0;
                                                if (j < C) {
                                                        cv = sh.cand[j];
                                                        uint32_t lo = b0, hi = b1 + 1;
                                                        while (lo < hi) {
                                                                const uint32_t mid = (lo + hi) >> 1;
                                                                if (bl[mid] < cv)
                                                                        lo = mid + 1;
                                                                else
                                                                        hi = mid;
                                                        }
                                                        bj = lo;
                                                }
                                                sh.blkof[tid + 1] = bj;
                                                __syncthreads();
                                                const uint32_t prevb = sh.blkof[tid];
                                                __syncthreads();
                                                {
                                                        const uint32_t lastb = __shfl(bj, 63, 64);
                                                        const bool lastwave = (tid >> 6) == (AND_WG / 64 - 1);
                                                        sh.blkof[lastwave ? 0 : tid + 1] = lastwave ? lastb : bj;
                                                }
                                                if (j < C && bj < t.nblocks && bj != prevb)
                                                        walk(bj, j, cv);
                                                __syncthreads();
                                        }
                                }
                        }
                        __syncthreads();
                }
                if (!WRITE && wave == 0)
                        task_hits[tix] = uni(sh.hits);
                __syncthreads();
        }
}
