// dev_stream.hpp — debug trace hooks, prefix-varint decode and the per-lane byte stream
// Part of libtrinity_hip.so (MI355X / gfx950); included by trinity_hip.hip.  New code, no reference source.
#pragma once
#include "dev_structs.hpp"
#include <hip/hip_runtime.h>

// ------------------------------------------------------------------------------------------ debug trace
// -DTRI_TRACE builds write per-workgroup progress markers into host-pinned memory; tri_batch_sync then polls
// with a watchdog (env TRINITY_WATCHDOG_S) and dumps the markers instead of hanging.  Not in product builds.
// -DTRI_PROF builds accumulate, per kernel phase, the shader-clock cycles wave 0 of every workgroup spent in it
// (g_prof[phase], summed over workgroups; read back and reset with tri_debug_prof).  Perf-probe builds only.
#ifdef TRI_PROF
__device__ unsigned long long g_prof[32];
struct ProfClock {
        long long t;
        unsigned long long acc[16];
        __device__ __forceinline__ void start() {
                for (int i = 0; i < 16; ++i)
                        acc[i] = 0;
                t = clock64();
        }
        __device__ __forceinline__ void lap(const int phase) {
                const long long n = clock64();
                acc[phase] += (unsigned long long)(n - t);
                t = n;
        }
        __device__ __forceinline__ void flush() {
                if (threadIdx.x == 0)
                        for (int i = 0; i < 16; ++i)
                                if (acc[i])
                                        atomicAdd(&g_prof[i], acc[i]);
        }
};
#define PROF_DECL ProfClock prof_
#define PROF_START() prof_.start()
#define PROF_LAP(p) prof_.lap(p)
#define PROF_FLUSH() prof_.flush()
#define PROF_ARG , ProfClock &prof_
#define PROF_PASS , prof_
#ifdef TRI_PROF_COUNTS // (event counters, g_prof[16 ..]: global atomics on a handful of addresses — they distort the phase times, so a build of their own)
#define PROF_COUNT(slot, n) atomicAdd(&g_prof[slot], (unsigned long long)(n))
#else
#define PROF_COUNT(slot, n)
#endif
#else
#define PROF_DECL
#define PROF_START()
#define PROF_LAP(p)
#define PROF_FLUSH()
#define PROF_ARG
#define PROF_PASS
#define PROF_COUNT(slot, n)
#endif

// -DTRI_TASKTIMES builds (perf probes): k_and leaves, per ticket, when the task started and ended (wall_clock64: 100 MHz) in a host-mapped buffer;
// with TRINITY_TASKTIMES set tri_batch_sync prints the kernel's span, how busy its workgroups were and its longest tasks
#ifdef TRI_TASKTIMES
static unsigned long long *g_tt_host = nullptr;
static size_t g_tt_cap = 0;
__device__ unsigned long long *g_tt = nullptr;
#define TASKTIME_(i) do { if (threadIdx.x == 0) g_tt[(i)] = wall_clock64(); } while (0)
#if TRI_TASKTIMES == 2 // (2: k_score's tasks instead of k_and's; 3: k_planes')
#define TASKTIME(i)
#define TASKTIME_SCORE(i) TASKTIME_(i)
#define TASKTIME_PLANES(i)
#define TASKTIME_DENSE(i)
#define TASKTIME_PHRASE(i)
#elif TRI_TASKTIMES == 3
#define TASKTIME(i)
#define TASKTIME_SCORE(i)
#define TASKTIME_PLANES(i) TASKTIME_(i)
#define TASKTIME_DENSE(i)
#define TASKTIME_PHRASE(i)
#elif TRI_TASKTIMES == 4 // (k_and_dense)
#define TASKTIME(i)
#define TASKTIME_SCORE(i)
#define TASKTIME_PLANES(i)
#define TASKTIME_DENSE(i) TASKTIME_(i)
#define TASKTIME_PHRASE(i)
#elif TRI_TASKTIMES == 5 // (k_phrase)
#define TASKTIME(i)
#define TASKTIME_SCORE(i)
#define TASKTIME_PLANES(i)
#define TASKTIME_DENSE(i)
#define TASKTIME_PHRASE(i) TASKTIME_(i)
#else
#define TASKTIME(i) TASKTIME_(i)
#define TASKTIME_SCORE(i)
#define TASKTIME_PLANES(i)
#define TASKTIME_DENSE(i)
#define TASKTIME_PHRASE(i)
#endif
#else
#define TASKTIME(i)
#define TASKTIME_SCORE(i)
#define TASKTIME_PLANES(i)
#define TASKTIME_DENSE(i)
#define TASKTIME_PHRASE(i)
#endif

#ifdef TRI_TRACE
static uint32_t *g_trace_host = nullptr;
__device__ volatile uint32_t *g_trace = nullptr;
#ifndef TRI_TRACE_MASK
#define TRI_TRACE_MASK 0xffffffffu
#endif
#define TRACE(stage, a, b)                                                      \
        do {                                                                    \
                if (((TRI_TRACE_MASK >> (stage)) & 1u) && threadIdx.x == 0 && g_trace) {                              \
                        volatile uint32_t *t_ = g_trace + (blockIdx.x & 63) * 4; \
                        t_[0] = (stage);                                        \
                        t_[1] = (a);                                            \
                        t_[2] = (b);                                            \
                        t_[3] = t_[3] + 1;                                      \
                        __threadfence_system();                                 \
                }                                                               \
        } while (0)
#else
#define TRACE(stage, a, b) \
        do {               \
        } while (0)
#endif

// ------------------------------------------------------------------------------------------ device: varint
// Prefix varint of Switch/switch_compiler_aux.h:53-80, branch-free.  `w` holds the next >= 5 stream bytes,
// least-significant byte first.
__device__ __forceinline__ uint32_t vb_decode(uint64_t w, uint32_t &len) {
        const uint32_t lo32 = (uint32_t)w;
        const uint32_t b0 = lo32 & 0xffu;
        const uint32_t ones = __clz(~(lo32 << 24)); // leading 1-bits of b0 (0..8)
        const uint32_t n = ones < 4u ? ones : 4u;
        const uint32_t be = __builtin_bswap32(lo32); // b0 b1 b2 b3
        const uint32_t v1 = b0;
        const uint32_t v2 = (be >> 16) & 0x3fffu;
        const uint32_t v3 = ((b0 & 0x1fu) << 16) | ((lo32 >> 8) & 0xffffu);
        const uint32_t v4 = be & 0x0fffffffu;
        const uint32_t v5 = (uint32_t)(w >> 8);
        len = n + 1;
        uint32_t v = v1;
        v = n == 1 ? v2 : v;
        v = n == 2 ? v3 : v;
        v = n == 3 ? v4 : v;
        v = n == 4 ? v5 : v;
        return v;
}

// Per-lane byte stream over global memory: a 16-byte register window (lo = next 8 bytes, hi = the following
// ones) refilled from aligned 8-byte loads, with AHEAD further qwords always in flight so that the load
// latency sits behind 8 * AHEAD bytes of decoding (index[] carries >= 64 bytes of slack past the last chunk).
#ifndef TRI_VB_AHEAD
#define TRI_VB_AHEAD 3
#endif
struct VbStream {
        static constexpr int AHEAD = TRI_VB_AHEAD; // qwords in flight beyond the 16-byte window
        const uint64_t *q;
        uint64_t lo, hi, n[AHEAD];
        int valid;

        __device__ __forceinline__ void init(const uint8_t *p) {
                const uint32_t sk = (uint32_t)((uintptr_t)p & 7u);
                q = (const uint64_t *)(p - sk); // by pointer arithmetic, not through an integer: the loads stay global_load
                const uint64_t w0 = q[0], w1 = q[1];
#pragma unroll
                for (int i = 0; i < AHEAD; ++i)
                        n[i] = q[2 + i];
                q += 2 + AHEAD;
                const uint32_t sh = sk * 8;
                lo = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
                hi = sh ? (w1 >> sh) : w1;
                valid = 16 - (int)sk;
        }
        __device__ __forceinline__ void refill() {
                if (valid <= 8) {
                        const uint64_t w = n[0];
#pragma unroll
                        for (int i = 0; i + 1 < AHEAD; ++i)
                                n[i] = n[i + 1];
                        n[AHEAD - 1] = *q++;
                        const uint32_t sh = (uint32_t)valid * 8; // 0..64
                        if (valid == 8)
                                hi = w;
                        else if (valid == 0) {
                                lo = w;
                                hi = 0;
                        } else {
                                lo |= w << sh;
                                hi = w >> (64 - sh);
                        }
                        valid += 8;
                }
        }
        __device__ __forceinline__ uint32_t next() {
                refill();
                uint32_t len;
                const uint32_t v = vb_decode(lo, len);
                const uint32_t s = len * 8;
                lo = (lo >> s) | (hi << (64 - s));
                hi >>= s;
                valid -= (int)len;
                return v;
        }
        // address of the next unread byte (8 * AHEAD bytes sit in n[], `valid` more in the window)
        __device__ __forceinline__ const uint8_t *tell() const { return (const uint8_t *)q - 8 * AHEAD - valid; }
        __device__ __forceinline__ uint32_t byte() {
                refill();
                const uint32_t b = (uint32_t)lo & 0xffu;
                lo = (lo >> 8) | (hi << 56);
                hi >>= 8;
                valid -= 1;
                return b;
        }
        __device__ __forceinline__ void skip(uint32_t n) {
                for (; n; --n)
                        (void)byte();
        }
        // after refill(): true when the next k (1..8) bytes are k one-byte varints (values < 128)
        __device__ __forceinline__ bool small_run(const uint32_t k) const { return (lo & (0x8080808080808080ull >> (8u * (8u - k)))) == 0; }
        // consume k (1..8) bytes, returning the window they were in (byte j = j-th value)
        __device__ __forceinline__ uint64_t take(const uint32_t k) {
                const uint64_t w = lo;
                if (k == 8) {
                        lo = hi;
                        hi = 0;
                } else {
                        const uint32_t s = k * 8;
                        lo = (lo >> s) | (hi << (64 - s));
                        hi >>= s;
                }
                valid -= (int)k;
                return w;
        }
};

