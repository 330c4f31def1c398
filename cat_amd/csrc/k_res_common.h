// cat_amd/csrc/k_res_common.h -- device helpers the two register-resident families (k_res.hip: generic layout over K compute units; k_fac.hip:
// factored layout) share: block reductions on LDS-only barriers, the tagged-granule exchange between the compute units of one recursion,
// the packed gather / accumulate macros of a chunk of four arcs.
#pragma once
#include "crf_kernels_decl.h"

namespace crf {

template <int NW = kResWaves>
__device__ __forceinline__ float res_block_sum(float v, float *red, int tid) {
    v = wave_sum(v);
    sync_lds();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    sync_lds();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += red[i];
    return s;
}
template <int NW = kResWaves, typename P>
__device__ __forceinline__ double res_mx_total(const P &p, int b, int lx, double *red, int tid) {
    double part = 0.0;
    for (int t = tid; t < lx; t += NW * kWave) part += (double)p.mx[(int64_t)b * p.T + t];
    part = wave_sum_d(part);
    sync_lds();
    if ((tid & 63) == 0) red[tid >> 6] = part;
    sync_lds();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += red[i];
    return s;
}
template <int NW = kResWaves>
__device__ __forceinline__ float res_frame_max(const float *wm) {
    float m = wm[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) m = fmaxf(m, wm[i]);
    return m;
}

// Exchange of the state vector between the K CUs of one recursion.  Granule = {tag << 32 | float
// bits}, ONE aligned 8-byte agent-scope store / load each: the data is the flag
// (cdna_hip_programming.md G16 R2).  Slots alternate per frame; a peer can be at most one exchange
// ahead, so two slots suffice.  Values are published from the row epilogues as soon as they are
// final, so most of the hand-off latency overlaps the rest of the frame's gathers.
// `same_l2`: every peer of this recursion was found (at run time, from HW_REG_XCC_ID) on this CU's XCD.
// Then a PLAIN 8-byte store is enough: it is written through to the shared L2 and stays there, and the
// peers' loads bypass their L1 (sc1), so the hand-off is an L2 round trip.  Otherwise the store is
// write-through to memory (sc1): correct under any placement, ~1 us slower per frame (an sc1 store drops
// the line from L2, MI355X_MICROARCH.md "stores of each flavour").
__device__ __forceinline__ void res_publish(gu64 *slot, int i, unsigned tag, float v, bool same_l2) {
    const unsigned long long g = ((unsigned long long)tag << 32) | __float_as_uint(v);
    if (same_l2) {
        asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(slot + i), "v"(g) : "memory");
    } else {
        __hip_atomic_store(slot + i, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// Fetch entries [lo,hi) published by a peer into LDS `v`; returns the largest value seen.
template <int NT = kResThreads>
__device__ __forceinline__ float res_fetch(gu64 *slot, float *v, int lo, int hi, unsigned tag, int *err, int tid) {
    float mx = 0.f;
    for (int base = lo; base < hi; base += kPoll * NT) {
        unsigned pending = 0;
#pragma unroll
        for (int q = 0; q < kPoll; ++q)
            if (base + q * NT + tid < hi) pending |= 1u << q;
        for (unsigned spins = 0; pending; ++spins) {
            unsigned long long gv[kPoll];
#pragma unroll
            for (int q = 0; q < kPoll; ++q)
                if (pending >> q & 1) gv[q] = __hip_atomic_load(slot + base + q * NT + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int q = 0; q < kPoll; ++q)
                if ((pending >> q & 1) && (unsigned)(gv[q] >> 32) == tag) {
                    const float x = __uint_as_float((unsigned)gv[q]);
                    v[base + q * NT + tid] = x;
                    mx = fmaxf(mx, x);
                    pending &= ~(1u << q);
                }
            if (!pending) break;
            // a peer died or was never scheduled: give up loudly (error word -> NaN loss), never hang;
            // once the word is set nobody waits again
            if (spins > (1u << 24)) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }  // ~tens of seconds
            if ((spins & 255u) == 255u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if (spins > 64) __builtin_amdgcn_s_sleep(8);  // back off when the peer is clearly not there yet
            else if (spins > 2) __builtin_amdgcn_s_sleep(1);
        }
    }
    return mx;
}

// ... the same for a LIST of entries (one per thread and round)
template <int NT>
__device__ __forceinline__ float res_fetch_list(gu64 *slot, float *v, const int *__restrict__ list, int l0, int l1, unsigned tag, int *err, int tid) {
    float mx = 0.f;
    for (int j = l0 + tid; j < l1; j += NT) {
        const int e = list[j];
        for (unsigned spins = 0;; ++spins) {
            const unsigned long long g = __hip_atomic_load(slot + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(g >> 32) == tag) { const float x = __uint_as_float((unsigned)g); v[e] = x; mx = fmaxf(mx, x); break; }
            if (spins > (1u << 24)) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            if ((spins & 255u) == 255u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if (spins > 64) __builtin_amdgcn_s_sleep(8);
            else if (spins > 2) __builtin_amdgcn_s_sleep(1);
        }
    }
    return mx;
}

// ... a range AND a list in ONE polling loop (round 4: the two-CU forward frame fetched the peer's U range and then, behind it, its list of
// L / A entries -- two dependent L2 round trips per frame where one will do): every thread polls up to kPoll granules of the range and one
// entry of the list together; a list longer than the workgroup takes res_fetch_list for the rest.
template <int NT>
__device__ __forceinline__ float res_fetch_both(gu64 *slot, float *v, int lo, int hi, const int *__restrict__ list, int l0, int l1, unsigned tag, int *err, int tid) {
    float mx = 0.f;
    const int le = l0 + tid < l1 ? list[l0 + tid] : -1;   // this thread's list entry (first round of the list)
    bool first = true;
    for (int base = lo; base < hi || first; base += kPoll * NT) {
        unsigned pending = 0;
#pragma unroll
        for (int q = 0; q < kPoll; ++q)
            if (base + q * NT + tid < hi) pending |= 1u << q;
        if (first && le >= 0) pending |= 1u << kPoll;
        first = false;
        for (unsigned spins = 0; pending; ++spins) {
            unsigned long long gv[kPoll + 1];
#pragma unroll
            for (int q = 0; q < kPoll; ++q)
                if (pending >> q & 1) gv[q] = __hip_atomic_load(slot + base + q * NT + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (pending >> kPoll & 1) gv[kPoll] = __hip_atomic_load(slot + le, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int q = 0; q < kPoll; ++q)
                if ((pending >> q & 1) && (unsigned)(gv[q] >> 32) == tag) {
                    const float x = __uint_as_float((unsigned)gv[q]);
                    v[base + q * NT + tid] = x;
                    mx = fmaxf(mx, x);
                    pending &= ~(1u << q);
                }
            if ((pending >> kPoll & 1) && (unsigned)(gv[kPoll] >> 32) == tag) {
                const float x = __uint_as_float((unsigned)gv[kPoll]);
                v[le] = x;
                mx = fmaxf(mx, x);
                pending &= ~(1u << kPoll);
            }
            if (!pending) break;
            if (spins > (1u << 24)) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            if ((spins & 255u) == 255u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if (spins > 64) __builtin_amdgcn_s_sleep(8);
            else if (spins > 2) __builtin_amdgcn_s_sleep(1);
        }
    }
    if (l0 + NT < l1) mx = fmaxf(mx, res_fetch_list<NT>(slot, v, list, l0 + NT, l1, tag, err, tid));
    return mx;
}

// Chunk sums in batches of kResBatch chunks: the 4*kResBatch gathers of a batch are one straight-line
// block (no branch between them), so every wave keeps 24 independent ds_read_b32 in flight -- with
// only 2 waves per SIMD that, not occupancy, is what hides the LDS latency.  The (uniform) slice-end
// branches sit between batches.  Unused chunks hold zero weights.
// Packed math: the four products of a chunk are two v_pk_fma_f32 lanes (PMC showed the frame loop is
// as much VALU-issue-bound as LDS-bound: ~530 VALU instructions per wave and frame before packing).
// The 4*kResBatch gathers of a batch; the products are chained FMAs into the row accumulator (measured:
// the frame loop is bound by instruction issue as much as by LDS -- one packed instruction less per
// chunk than "multiply, FMA, add" was worth 3%).
#define CRF_RES_GATHER(g01, g23, A, xb, c0) CRF_RES_GATHER_N(g01, g23, A, xb, c0, kResBatch)
#define CRF_RES_GATHER_N(g01, g23, A, xb, c0, NB_)                                                        \
    _Pragma("unroll") for (int ci = 0; ci < (NB_); ++ci) {                                            \
        const int c = (c0) + ci;                                                                          \
        const unsigned i01 = A[6 * c], i23 = A[6 * c + 1];                                                \
        g01[ci].x = *(const float *)(xb + (i01 & 0xffffu)); g01[ci].y = *(const float *)(xb + (i01 >> 16)); \
        g23[ci].x = *(const float *)(xb + (i23 & 0xffffu)); g23[ci].y = *(const float *)(xb + (i23 >> 16)); \
    }
#define CRF_RES_CHUNK_ACC(accv, g01, g23, A, c, ci)                                                       \
    {                                                                                                     \
        f32x2 w01, w23;                                                                                   \
        w01.x = __uint_as_float(A[6 * (c) + 2]); w01.y = __uint_as_float(A[6 * (c) + 3]);                 \
        w23.x = __uint_as_float(A[6 * (c) + 4]); w23.y = __uint_as_float(A[6 * (c) + 5]);                 \
        accv = __builtin_elementwise_fma(g23[ci], w23, __builtin_elementwise_fma(g01[ci], w01, accv));    \
    }


}  // namespace crf
