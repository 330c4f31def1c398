// cat_amd/csrc/k_batch.hip -- utterance-minor denominator for graphs beyond the registers (one launch per frame, or one persistent launch)
// (one translation unit per kernel family, built in parallel by cat_amd/build.py; the explicit instantiations at the end are the
//  ones the host side in crf_host.hip launches -- a missing one is a link error, -Wl,-z,defs)
#include "crf_device.h"
#include "crf_kernels_decl.h"

namespace crf {

// =============================================================================================
// UTTERANCE-MINOR ("batch") denominator for graphs that do not fit the register-resident layouts (crf_internal.h:
// BatchDev).  The reference runs ANY graph with one launch per frame and one block per utterance, re-reading every arc
// for every utterance (den_calculate.cu:75-103, 189-227, 443-476); the streaming kernels above do the same from one
// persistent workgroup per utterance.  Here the batch is the minor dimension of everything:
//     a_t  [group][state][ul]     z_t [group][pair][ul]     Q_t, BP_t [group][pair][ul]     e'_t [group][label][ul]
// (utterance u = group * UL + ul; UL utterances = one 32..256-byte segment per entry, chosen so that ONE group's state
// vector stays in an XCD's 4 MiB L2: ws_layout).  A wave takes one row (a destination state forward, a source state backward) with the utterances in its lanes (UL
// utterances x 64/UL arcs of the row side by side), so an arc is fetched ONCE per frame for the whole batch and every
// gather of a state-vector entry is one contiguous UL*4-byte segment.  One launch per frame -- the kernel boundary is
// the grid barrier and makes the vectors visible across XCDs -- with the forward step of frame j and the backward step
// of frame T-j in the same launch.  XCD placement (speed only): the per-XCD L2s do not share, and with every XCD
// gathering from every group's vectors of both directions (8.4 MB at S = 16 k, B = 64) nearly every gather missed L2
// (measured 93 us per launch = 3 TB/s of fabric reads).  A (group, direction) "combo" therefore belongs to 8 / #combos
// XCDs (block b runs on XCD b % 8): an XCD gathers from ONE vector that fits its L2 and streams its share of the arcs.  Scaling: per utterance and frame an exact power of two from the maximum of the
// vector (atomic max per utterance, three slots in rotation), integer exponents carried per utterance.
// Backward frames are aligned at the END of the padded batch (iteration i handles frame T-1-i of every utterance); an
// utterance joins when the iteration reaches its last frame.
// =============================================================================================

// ep [B][T][V] -> ept [T][grp][V][UL] through a 64 x UL tile in LDS.  grid (ceil(V / 64), T, Bp / UL)
template <int UL>
__global__ __launch_bounds__(kBatThreads) void crf_batch_transpose_kernel(BatchParams p) {
    __shared__ float tile[UL][65];
    const int v0 = blockIdx.x * 64, t = blockIdx.y, u0 = blockIdx.z * UL, tid = threadIdx.x;
    for (int i = tid; i < UL * 64; i += kBatThreads) {
        const int u = i >> 6, v = i & 63;
        const bool ok = u0 + u < p.B && v0 + v < p.V && t < p.lx[u0 + u];
        tile[u][v] = ok ? p.ep[((int64_t)(u0 + u) * p.T + t) * p.V + v0 + v] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < UL * 64; i += kBatThreads) {
        const int v = i / UL, u = i % UL;
        if (v0 + v < p.V) p.ept[(((int64_t)t * p.ngrp + blockIdx.z) * p.V + v0 + v) * UL + u] = tile[u][v];
    }
}

// a_0, the slots, the exponents.  grid: enough blocks for S * Bp elements
__global__ __launch_bounds__(kBatThreads) void crf_batch_init_kernel(BatchParams p) {
    const int64_t i = (int64_t)blockIdx.x * kBatThreads + threadIdx.x;
    const int UL = p.Bp / p.ngrp;
    if (i < (int64_t)p.SX * p.Bp) p.Af[i] = p.x_start[(i / UL) % p.SX] * pow2f(kScaleExp);
    if (blockIdx.x == 0) {
        float m = 0.f;
        for (int s = threadIdx.x; s < p.SX; s += kBatThreads) m = fmaxf(m, p.x_start[s]);
        __shared__ float red[kBatWaves];
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * pow2f(kScaleExp);
        for (int i = threadIdx.x; i < 640; i += kBatThreads) p.bar[i] = 0u;   // grid barrier words and time-out word of the persistent launch
        for (int u = threadIdx.x; u < p.Bp; u += kBatThreads) {
            p.mxf[u] = __float_as_uint(m); p.mxf[p.Bp + u] = 0u; p.mxf[2 * p.Bp + u] = 0u;
            p.mxb[u] = 0u; p.mxb[p.Bp + u] = 0u; p.mxb[2 * p.Bp + u] = 0u;
            p.Ef[u] = kScaleExp; p.Fb[u] = kScaleExp; p.zs[u] = 0.f; p.zb[u] = 0.f;
        }
    }
}

// sum over the AL = 64/UL arc lanes that share an utterance (lanes u, u + UL, u + 2 UL, ...)
template <int UL>
__device__ __forceinline__ float arc_lane_sum(float v) {
#pragma unroll
    for (int o = 32; o >= UL; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int UL>
__device__ __forceinline__ float arc_lane_max(float v) {
#pragma unroll
    for (int o = 32; o >= UL; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---------------------------------------------------------------------------------------------
// PERSISTENT form (round 6; crf_batch_persist_kernel below): all frames in ONE launch, the kernel boundary replaced by a grid barrier.
// Only the state vectors (and the per-utterance maxima) cross compute units inside the launch; they do so by PER-ACCESS agent-scope
// operations -- sc1 (write-through) stores and sc1 loads, which bypass the reading CU's L1 and are served by its XCD's L2 or the fabric
// (MI355X_MICROARCH.md "inter-workgroup visibility") -- so no wave ever executes a bulk buffer_wbl2 / buffer_inv (round 3's
// persistent experiment fenced in every wave: 80 us per frame; profiles/round6_grid_barrier.txt: this protocol 0 stale values in
// 3 x 1 500 frames x 150 MB of checked gathers, the empty barrier 2.2 us).  Arc records, row descriptors and emissions are written
// before the launch and stay plainly cached; Q / BP rows are read by a later kernel.  PERS = false compiles to the plain accesses of
// the per-frame launches.
// ---------------------------------------------------------------------------------------------
#define CRF_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#ifndef CRF_X_BATLD
#define CRF_X_BATLD 0       // TIMING EXPERIMENTS ONLY (wrong results): the persistent kernel's 16-byte gathers as 1 = plain global loads, 2 = buffer loads without sc1
#endif
template <bool PERS>
struct VecRef {                                                    // a state vector of one utterance group: base + byte size
    const char *base;
    __amdgpu_buffer_rsrc_t rs;
    __device__ __forceinline__ VecRef(const float *p, size_t bytes) : base((const char *)p) {
        if constexpr (PERS) rs = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x27000);
    }
    __device__ __forceinline__ f32x4 ld4(unsigned off) const {      // 16 bytes at byte offset off
        if constexpr (PERS && CRF_X_BATLD != 1) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, CRF_X_BATLD == 2 ? 0 : 16));
        else return *(const f32x4 *)(base + off);
    }
    __device__ __forceinline__ float ld1(size_t idx) const {
        if constexpr (PERS) return __builtin_bit_cast(float, __hip_atomic_load((const unsigned *)base + idx, CRF_RLX_AGENT));
        else return ((const float *)base)[idx];
    }
    __device__ __forceinline__ void st4(size_t idx, const f32x4 &v) const {   // 16 bytes at FLOAT index idx
        if constexpr (PERS) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (unsigned)idx * 4u, 0, 16);
        else *(f32x4 *)(base + idx * 4) = v;
    }
    __device__ __forceinline__ void st1(size_t idx, float v) const {
        if constexpr (PERS) __hip_atomic_store((unsigned *)base + idx, __float_as_uint(v), CRF_RLX_AGENT);
        else ((float *)base)[idx] = v;
    }
};
template <bool PERS>
__device__ __forceinline__ unsigned ld_word(const unsigned *p) {
    if constexpr (PERS) return __hip_atomic_load(p, CRF_RLX_AGENT);
    else return *p;
}
template <bool PERS>
__device__ __forceinline__ void st_word(unsigned *p, unsigned v) {
    if constexpr (PERS) __hip_atomic_store(p, v, CRF_RLX_AGENT);
    else *p = v;
}

// sum over the arcs [a0, a1) of w * X[idx][u].  The wave fetches 64 arcs at a time (one coalesced 512-byte load), the arc
// lanes of an utterance then walk them through lane broadcasts: the gathers of a chunk are independent loads, all in flight
// together -- with the arc fetched inside the loop every gather waited for its own arc first (two dependent trips to L2 per
// arc: 77 us per frame on the S = 16 k graph).
template <int UL, bool PERS>
__device__ __forceinline__ float bat_row_sum(const int2 *__restrict__ arcs, int a0, int a1, const VecRef<PERS> &X, int ul, int lane, int aj) {
    constexpr int AL = 64 / UL;
    float acc0 = 0.f, acc1 = 0.f;
    for (int c = a0; c < a1; c += 64) {
        const int n = min(64, a1 - c);
        int2 arc = int2{0, 0};                                    // (lanes past the end: state 0, weight 0)
        if (lane < n) arc = arcs[c + lane];
        const int steps = (n + AL - 1) / AL;
#pragma unroll 8
        for (int i = 0; i < steps; i += 2) {
            int s0, w0, s1, w1;
            if (AL == 1) {
                s0 = __builtin_amdgcn_readlane(arc.x, i); w0 = __builtin_amdgcn_readlane(arc.y, i);
                s1 = __builtin_amdgcn_readlane(arc.x, min(i + 1, 63)); w1 = i + 1 < steps ? __builtin_amdgcn_readlane(arc.y, min(i + 1, 63)) : 0;
            } else {
                s0 = __shfl(arc.x, i * AL + aj); w0 = __shfl(arc.y, i * AL + aj);
                const int l1 = min((i + 1) * AL + aj, 63);
                s1 = __shfl(arc.x, l1); w1 = i + 1 < steps ? __shfl(arc.y, l1) : 0;
            }
            acc0 = fmaf(X.ld1((size_t)s0 * UL + ul), __int_as_float(w0), acc0);
            acc1 = fmaf(X.ld1((size_t)s1 * UL + ul), __int_as_float(w1), acc1);
        }
    }
    return arc_lane_sum<UL>(acc0 + acc1);
}

// A wave's TASK of an arc stream (crf_internal.h: StreamDirDev).  A lane takes FOUR utterances (uq: which four of the
// group's UL) -- one 16-byte gather per arc and lane, LG = UL / 4 lanes a row -- and lane group aj walks row aj of every
// bundle: measured with one utterance per lane, the frame kernel was bound by the NUMBER of memory instructions a CU can
// take (~9 cycles each; with every gather hitting the same cached lines it was no faster), not by what they fetched.
// X = the gathered vector of this utterance group ([entry][UL]).  Everything but the gathers reaches the wave through its
// slice of LDS: the records (two 4 KB halves, kStreamChunk / AL batches each, staged one chunk ahead: the loads of the chunk
// after the next are the OLDEST entries of the memory queue), the row descriptors of the task's (at most kStreamBundles)
// bundles, and a ring of two emission rows filled from a register set loaded one bundle earlier.  So the memory queue holds
// the gathers -- D batches of kStreamBatch in flight, consumed in order behind partial vmcnt waits -- and a row's stores;
// a wait for something just requested happens nowhere.  epi(acc, m, e) is called at every bundle end with the row's
// descriptor {state, pair, label} and the four utterances' emissions et[label].
template <int UL, int D, int NM, int NR, bool PERS, typename Epi>
__device__ __forceinline__ void bat_stream(const StreamDirDev &sd, const int4 tk, const VecRef<PERS> &X, const float *__restrict__ et,
                                           int uq, int aj, int lane, char *ldsw, int tmi, const bool fill, Epi &&epi) {
    // fill (uniform): the task's records and row descriptors are fetched into the wave's slice of LDS as described above.  false: they ARE
    // there -- the persistent kernel's later frames of a wave with ONE task of at most two chunks of records, which then fetches nothing
    // but state-vector entries and emissions
    constexpr int LG = UL / 4, AL = 64 / LG, CB = kStreamChunk / AL;
    static_assert(kStreamBatch == 4 && CB % D == 0 && CB >= D, "a chunk holds a whole number of pipeline rounds");
    static_assert((NM == 1 && NR == 1) || (NM == 3 && (NR == 3 || NR == 4)), "plain or factored rows");
    f32x4 *elds = (f32x4 *)(ldsw + 2 * kStreamRecB);               // [2][NR][lane]
    int4 *mlds = (int4 *)(ldsw + 2 * kStreamRecB + 2 * 64 * 16 * (NM == 1 ? 1 : 4));   // [bundle][AL][NM]
    const int b0 = __builtin_amdgcn_readfirstlane(tk.x), nb = __builtin_amdgcn_readfirstlane(tk.y);
    const int bund0 = __builtin_amdgcn_readfirstlane(tk.z), nbund = __builtin_amdgcn_readfirstlane(tk.w);
    CRF_TM(tmi >= 0, tmi + 2);
    const int4 *gsrc = (const int4 *)sd.recs + (size_t)b0 * AL * 2 + lane;   // a chunk = 256 int4: four per lane (the stream is padded)
    if (fill) {   // chunk 0 -> LDS half 0 (the only wait for something just requested: once per task)
        const int4 s0 = gsrc[0], s1 = gsrc[64], s2 = gsrc[128], s3 = gsrc[192];
        const int4 *mp = sd.meta + (size_t)bund0 * AL * NM;
        constexpr int MAXB = stream_max_bundles(UL, NM == 3);
        constexpr int NMW = (MAXB * AL * NM + 63) / 64;
        int4 mm[NMW];
#pragma unroll
        for (int q = 0; q < NMW; ++q) mm[q] = (q * 64 + lane < nbund * AL * NM) ? mp[q * 64 + lane] : int4{-1, 0, 0, 0};
        *(int4 *)(ldsw + lane * 16) = s0; *(int4 *)(ldsw + (64 + lane) * 16) = s1;
        *(int4 *)(ldsw + (128 + lane) * 16) = s2; *(int4 *)(ldsw + (192 + lane) * 16) = s3;
#pragma unroll
        for (int q = 0; q < NMW; ++q)
            if ((NMW * 64 == MAXB * AL * NM) || q * 64 + lane < MAXB * AL * NM) mlds[q * 64 + lane] = mm[q];   // (the slice ends there)
    }
    CRF_TM(tmi >= 0, tmi + 3);
    int4 st0 = int4{0, 0, 0, 0}, st1 = st0, st2 = st0, st3 = st0;
    if (fill) { st0 = gsrc[256]; st1 = gsrc[320]; st2 = gsrc[384]; st3 = gsrc[448]; }   // chunk 1 (padding if there is none)
    gsrc += 512;
    const f32x4 *et4 = (const f32x4 *)et + uq;                     // et[label * UL + 4 uq ..]
    // value k of the row (bundle bd, lane group aj): emissions of its label(s), entries its epilogue reads
    auto ringsrc = [&](const int bd, const int k) __attribute__((always_inline)) -> f32x4 {
        const int4 *m = mlds + ((size_t)bd * AL + aj) * NM;
        if (k == 0) return et4[(size_t)m[0].z * LG];
        if (k == 1) return et4[(size_t)m[NM > 1 ? 1 : 0].z * LG];
        // X[entry * UL + 4 uq ..]; a padding row's entry may be -1 (it used to read in front of the vector; the value is never used)
        return X.ld4((unsigned)max(k == 2 ? m[NM > 1 ? 2 : 0].x : m[NM > 1 ? 2 : 0].z, 0) * (unsigned)(UL * 4) + (unsigned)uq * 16u);
    };
    // values of bundles 0 and 1 -> ring (written once the first gathers are out), bundle 2 -> eA
    f32x4 ei0[NR], ei1[NR], eA[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) { ei0[k] = ringsrc(0, k); ei1[k] = ringsrc(nbund > 1 ? 1 : 0, k); eA[k] = ringsrc(nbund > 2 ? 2 : 0, k); }
    const unsigned uq16 = (unsigned)uq * 16u;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 x[D][4];
    float w[D][4];
    unsigned fl[D];
    int bund = 0;
    auto issue = [&](const int j, const int b) __attribute__((always_inline)) {   // gathers of batch b of the task into slot j
        const int4 *lp = (const int4 *)(ldsw + (((b % (2 * CB)) * AL + aj) * 32));
        int4 r0 = lp[0], r1 = lp[1];
        // (a record's index is entry * UL with the flag in bit 31: the shift to bytes drops the flag -- one VALU per gather --
        // and the address is a 32-bit offset to a uniform base)
        fl[j] = (unsigned)__builtin_amdgcn_readfirstlane(r0.x) >> 31;
        x[j][0] = X.ld4(((unsigned)r0.x << 2) + uq16);
        x[j][1] = X.ld4(((unsigned)r0.z << 2) + uq16);
        x[j][2] = X.ld4(((unsigned)r1.x << 2) + uq16);
        x[j][3] = X.ld4(((unsigned)r1.z << 2) + uq16);
        w[j][0] = __int_as_float(r0.y); w[j][1] = __int_as_float(r0.w); w[j][2] = __int_as_float(r1.y); w[j][3] = __int_as_float(r1.w);
    };
    auto consume = [&](const int j) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = __builtin_elementwise_fma(x[j][k], (f32x4){w[j][k], w[j][k], w[j][k], w[j][k]}, acc);
        if (fl[j] & 1u) {                                          // (uniform) the rows of the bundle end with this batch
            f32x4 e[NR];
#pragma unroll
            for (int k = 0; k < NR; ++k) e[k] = elds[((bund & 1) * NR + k) * 64 + lane];
            epi(acc, mlds + ((size_t)bund * AL + aj) * NM, e);
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < NR; ++k) elds[((bund & 1) * NR + k) * 64 + lane] = eA[k];   // values of bundle + 2 (asked for one bundle ago)
            if (bund + 3 < nbund) {
#pragma unroll
                for (int k = 0; k < NR; ++k) eA[k] = ringsrc(bund + 3, k);
            }
            ++bund;
        }
    };
    auto stage = [&](const int c) __attribute__((always_inline)) {   // entering chunk c: chunk c + 1 -> the other half, ask for chunk c + 2
        if (fill && (c + 1) * CB < nb) {
            char *h = ldsw + ((c + 1) & 1) * kStreamRecB;
            *(int4 *)(h + lane * 16) = st0; *(int4 *)(h + (64 + lane) * 16) = st1;
            *(int4 *)(h + (128 + lane) * 16) = st2; *(int4 *)(h + (192 + lane) * 16) = st3;
            if ((c + 2) * CB < nb) { st0 = gsrc[0]; st1 = gsrc[64]; st2 = gsrc[128]; st3 = gsrc[192]; gsrc += 256; }
        }
    };
    stage(0);
#pragma unroll
    for (int j = 0; j < D - 1; ++j)
        if (j < nb) issue(j, j);
#pragma unroll
    for (int k = 0; k < NR; ++k) { elds[k * 64 + lane] = ei0[k]; elds[(NR + k) * 64 + lane] = ei1[k]; }   // (older than the gathers just issued)
    CRF_TM(tmi >= 0, tmi + 4);
#ifdef CRF_TIMING
    if (tmi >= 0 && lane == 0) { g_tm[tmi + 8] = (unsigned long long)nb; g_tm[tmi + 9] = (unsigned long long)nbund; }
#endif
    int b = 0;
    for (; b + 2 * D - 1 <= nb; b += D) {                          // steady state: no conditions around the loads
        if (b > 0 && b % CB == 0) stage(b / CB);
#pragma unroll
        for (int j = 0; j < D; ++j) { issue((j + D - 1) % D, b + j + D - 1); consume(j); }
    }
    for (; b < nb; b += D) {
        if (b > 0 && b % CB == 0) stage(b / CB);
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (b + j + D - 1 < nb) issue((j + D - 1) % D, b + j + D - 1);
            if (b + j < nb) consume(j);
        }
    }
    CRF_TM(tmi >= 0, tmi + 5);
}

// What a workgroup (and each of its lanes) is in EVERY frame: its combo, its utterances and their lengths, its first task -- worked
// out once per launch (the persistent kernel keeps it in registers over all frames).
template <int UL>
struct BatWg {
    int dir, grp, chunk, nchunk, w0, NW;
    int ul, aj, uq, sj;                                            // rest rows: utterance ul, arc lane aj; stream: utterances 4 uq .. 4 uq + 3, row sj of a bundle
    int u, lx, u4, lx4[4];
    bool lead;                                                     // one writer per utterance for the scalars
    int4 tk0;                                                      // the wave's first task descriptor
};
template <int UL>
__device__ __forceinline__ BatWg<UL> bat_setup(const BatchParams &p, int lane, int wave) {
    constexpr int LG = UL / 4;
    BatWg<UL> c;
    int combo;
    bat_decode((int)blockIdx.x, (int)gridDim.x, 2 * p.ngrp, &combo, &c.chunk, &c.nchunk);   // (crf_internal.h)
    c.dir = combo & 1; c.grp = combo >> 1;
    c.ul = lane % UL; c.aj = lane / UL; c.uq = lane % LG; c.sj = lane / LG;
    c.u = c.grp * UL + c.ul;
    c.lx = c.u < p.B ? p.lx[c.u] : 0;
    c.u4 = c.grp * UL + 4 * c.uq;                                  // first of the lane's four utterances (stream)
#pragma unroll
    for (int k = 0; k < 4; ++k) c.lx4[k] = c.u4 + k < p.B ? p.lx[c.u4 + k] : 0;
    c.w0 = c.chunk * kBatWaves + wave; c.NW = c.nchunk * kBatWaves;   // this wave among the waves of its combo
    // the wave's first task descriptor: asked for before anything else (it heads the chain descriptor -> records and row
    // descriptors -> emissions -> first gathers, three dependent trips to a cold L2 at the start of every launch)
    const StreamDirDev &sdd = c.dir == 0 ? p.st.f : p.st.b;
    c.tk0 = int4{0, 0, 0, 0};
    if (c.w0 < sdd.ntasks) c.tk0 = sdd.tasks[c.w0];
    c.lead = c.chunk == 0 && wave == 0 && c.aj == 0;
    return c;
}

// One frame of both recursions: launch number j = forward frame j, backward frame T - j.  The waves of a combo take the tasks of its
// arc stream (rows with one entering pair: all of a T o LM graph) and then the remaining rows one at a time (bat_row_sum: one utterance
// per lane, 64 / UL arcs of the row side by side).  PERS: the state vectors and the per-utterance maxima by agent-scope accesses (above).
template <int UL, int D, bool FAC, bool PERS>
__device__ __forceinline__ void bat_frame(const BatchParams &p, const BatWg<UL> &c, const int j, unsigned *umax, char *ldsw, const int tmi, bool &filled) {
    constexpr int ALR = 64 / UL;                                   // rest rows: arc lanes per utterance
    constexpr int NM = FAC ? 3 : 1;                                // descriptor words per stream row
    const int tid = threadIdx.x, lane = tid & 63;
    const int ul = c.ul, aj = c.aj, uq = c.uq, sj = c.sj, dir = c.dir, grp = c.grp, u = c.u, lx = c.lx, u4 = c.u4, w0 = c.w0, NW = c.NW;
    const int T = p.T, P = p.P;
    const BatchDev &g = p.g;
    const size_t gS = (size_t)grp * p.SX * UL, gP = (size_t)grp * P * UL, gV = (size_t)grp * p.V * UL;
    const size_t Sall = (size_t)p.SX * p.Bp, Pall = (size_t)P * p.Bp, Vall = (size_t)p.V * p.Bp;
    // persistent launch: a wave with ONE task whose records fit the two halves of its LDS slice (kStreamChunk batches per lane group) fills the
    // slice in its first frame only -- records and row descriptors are the same in every frame
    constexpr int kResidentBatches = 2 * (kStreamChunk / (64 / (UL / 4)));
    const StreamDirDev &sdd = dir == 0 ? p.st.f : p.st.b;
    const bool one_task = w0 < sdd.ntasks && w0 + NW >= sdd.ntasks && __builtin_amdgcn_readfirstlane(c.tk0.y) <= kResidentBatches;
    const bool fill = !(PERS && one_task && filled);               // (filled: an earlier frame of this launch has run the wave's task)
    if (tid < UL) umax[tid] = 0u;
    __syncthreads();
    CRF_TM(tmi >= 0 && c.lx4[0] + c.lx4[1] + c.lx4[2] + c.lx4[3] + lx >= 0, tmi + 1);   // (the scalar loads have landed)
    float mymax = 0.f;                                             // rest rows: utterance ul
    f32x4 mymax4 = {0.f, 0.f, 0.f, 0.f};                           // stream: the lane's four utterances
    if (dir == 0) {
        const int t = j;
        if (t >= T) return;
        const bool active = t < lx;
        const int k = rescale_exp(__uint_as_float(ld_word<PERS>(p.mxf + (t % 3) * p.Bp + u)));
        const float sc = pow2f(k);
        f32x4 sc4;
        bool act4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { sc4[q] = pow2f(rescale_exp(__uint_as_float(ld_word<PERS>(p.mxf + (t % 3) * p.Bp + u4 + q)))); act4[q] = t < c.lx4[q]; }
        const bool all4 = act4[0] && act4[1] && act4[2] && act4[3];
        const VecRef<PERS> Ac(p.Af + (size_t)(t & 1) * Sall + gS, (size_t)p.SX * UL * 4);
        const VecRef<PERS> An(p.Af + (size_t)((t + 1) & 1) * Sall + gS, (size_t)p.SX * UL * 4);
        const float *et = p.ept + (size_t)t * Vall + gV;
        float *Qt = p.Q + (size_t)t * Pall + gP;
        filled = true;
        for (int task = w0; task < p.st.f.ntasks; task += NW)
            bat_stream<UL, D, NM, FAC ? 3 : 1, PERS>(p.st.f, task == w0 ? c.tk0 : p.st.f.tasks[task], Ac, et, uq, sj, lane, ldsw, tmi, fill, [&](const f32x4 &acc, const int4 *m, const f32x4 *e) __attribute__((always_inline)) {
                if (m[0].x < 0) return;                            // padding row of the last bundle
                // an utterance that has ended keeps a_lx where it is: nobody writes that buffer for it again
                // (crf_batch_zsum_kernel reads it there)
                const f32x4 q = acc * sc4, an = e[0] * q;
                float *qp = Qt + (size_t)m[0].y * UL + 4 * uq;
                const size_t ai = (size_t)m[0].x * UL + 4 * uq;
                if (all4) { *(f32x4 *)qp = q; An.st4(ai, an); }
                else {
#pragma unroll
                    for (int q_ = 0; q_ < 4; ++q_) if (act4[q_]) { qp[q_] = q[q_]; An.st1(ai + q_, an[q_]); }
                }
                f32x4 top = an;                                    // the largest entry this row writes
                if constexpr (FAC) {
                    if (m[1].x >= 0) {                             // the couple's tail row, folded in: q = w * U_t, and U_{t+1} = both states' a
                        const float tw = __int_as_float(m[2].y);
                        const f32x4 qt = e[2] * (f32x4){tw, tw, tw, tw} * sc4, at = e[1] * qt, un = an + at;
                        float *qp1 = Qt + (size_t)m[1].y * UL + 4 * uq;
                        const size_t ai1 = (size_t)m[1].x * UL + 4 * uq, ui = (size_t)m[2].x * UL + 4 * uq;
                        if (all4) { *(f32x4 *)qp1 = qt; An.st4(ai1, at); An.st4(ui, un); }
                        else {
#pragma unroll
                            for (int q_ = 0; q_ < 4; ++q_) if (act4[q_]) { qp1[q_] = qt[q_]; An.st1(ai1 + q_, at[q_]); An.st1(ui + q_, un[q_]); }
                        }
                        top = un;
                    }
                }
#pragma unroll
                for (int q_ = 0; q_ < 4; ++q_) if (act4[q_]) mymax4[q_] = fmaxf(mymax4[q_], top[q_]);
            });
        for (int i = w0; i < p.st.f.nrest; i += NW) {
            const int r = __builtin_amdgcn_readfirstlane(p.st.f.rest[i]);
            const int d = __builtin_amdgcn_readfirstlane(g.frow_d[r]);   // (the row is the wave's: everything about it is uniform)
            int4 ds = g.frow[r];
            ds.x = __builtin_amdgcn_readfirstlane(ds.x); ds.y = __builtin_amdgcn_readfirstlane(ds.y);
            ds.z = __builtin_amdgcn_readfirstlane(ds.z); ds.w = __builtin_amdgcn_readfirstlane(ds.w);
            float acc = 0.f;
            if (ds.w & 0x40000000) {                               // one pair enters the state
                const float e = et[(size_t)(ds.w & 0xffff) * UL + ul];   // (requested before the arcs)
                const float q = bat_row_sum<UL, PERS>(g.farcs, ds.x, ds.y, Ac, ul, lane, aj) * sc;
                if (aj == 0 && active) Qt[(size_t)ds.z * UL + ul] = q;
                acc = e * q;
            } else {
                for (int kk = ds.z; kk < ds.w; ++kk) {
                    int4 pl = g.stp[kk];
                    pl.z = __builtin_amdgcn_readfirstlane(pl.z); pl.w = __builtin_amdgcn_readfirstlane(pl.w);
                    const float q = bat_row_sum<UL, PERS>(g.farcs, pl.z, pl.w, Ac, ul, lane, aj) * sc;
                    if (aj == 0 && active) Qt[(size_t)pl.x * UL + ul] = q;
                    acc = fmaf(et[(size_t)pl.y * UL + ul], q, acc);
                }
            }
            if (aj == 0 && active) An.st1((size_t)d * UL + ul, acc);
            if (active) mymax = fmaxf(mymax, acc);
        }
        if (c.lead) {
            if (active) st_word<PERS>((unsigned *)p.Ef + u, ld_word<PERS>((const unsigned *)p.Ef + u) + (unsigned)(k + kEpExp));   // exponent of a_{t+1}
            st_word<PERS>(p.mxf + ((t + 2) % 3) * p.Bp + u, 0u);   // the slot the launch after next adds to
        }
    } else {
        const int t = T - j;                                       // t = T (nothing active yet) ... 0
        const bool active = t < lx;                                // b_t of this utterance is computed
        const bool starts = t - 1 == lx - 1 && lx > 0;             // frame t-1 is its last frame: z_{lx-1} is set up
        const int k = rescale_exp(__uint_as_float(ld_word<PERS>(p.mxb + (j % 3) * p.Bp + u)));
        const float sc = pow2f(k);
        f32x4 sc4;
        bool act4[4], st4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc4[q] = pow2f(rescale_exp(__uint_as_float(ld_word<PERS>(p.mxb + (j % 3) * p.Bp + u4 + q))));
            act4[q] = t < c.lx4[q]; st4[q] = t == c.lx4[q] && c.lx4[q] > 0;
        }
        const bool all4 = act4[0] && act4[1] && act4[2] && act4[3];
        const VecRef<PERS> Zc(p.Zb + (size_t)(j & 1) * Pall + gP, (size_t)P * UL * 4);
        const VecRef<PERS> Zn(p.Zb + (size_t)((j + 1) & 1) * Pall + gP, (size_t)P * UL * 4);
        const float *ep1 = p.ept + (size_t)(t >= 1 ? t - 1 : 0) * Vall + gV;   // (t = 0: read, not used)
        float *BPt = t >= 1 ? p.BP + (size_t)(t - 1) * Pall + gP : nullptr;
        const bool any_active = __ballot(active) != 0ull;
        if (any_active || __ballot(starts) != 0ull) {
            filled = true;
            for (int task = w0; task < p.st.b.ntasks; task += NW)
                bat_stream<UL, D, NM, FAC ? 4 : 1, PERS>(p.st.b, task == w0 ? c.tk0 : p.st.b.tasks[task], Zc, ep1, uq, sj, lane, ldsw, tmi, fill, [&](const f32x4 &acc, const int4 *m, const f32x4 *e) __attribute__((always_inline)) {
                    if (m[0].x < 0) return;
                    // one output (plain rows), or the two states of a couple: the common out-arcs' sum + each state's extra arc
                    auto output = [&](const f32x4 &bv, const int st_, const int pr_, const f32x4 &em) __attribute__((always_inline)) {
                        if (t == 0) {
                            const float st = p.start_lin[st_];
                            if (st != 0.f) {
#pragma unroll
                                for (int q_ = 0; q_ < 4; ++q_) if (act4[q_]) atomicAdd(&p.zb[u4 + q_], st * bv[q_]);
                            }
                            return;
                        }
                        float *bp = BPt + (size_t)pr_ * UL + 4 * uq;
                        const size_t zi = (size_t)pr_ * UL + 4 * uq;
                        if (all4) {
                            const f32x4 z = em * bv;
                            *(f32x4 *)bp = bv; Zn.st4(zi, z);
                            mymax4 = __builtin_elementwise_max(mymax4, z);
                        } else {
                            const float eend = p.end_lin[st_] * pow2f(kScaleExp);
#pragma unroll
                            for (int q_ = 0; q_ < 4; ++q_)
                                if (act4[q_] || st4[q_]) {
                                    const float out = act4[q_] ? bv[q_] : eend, z = em[q_] * out;
                                    bp[q_] = out; Zn.st1(zi + q_, z);
                                    mymax4[q_] = fmaxf(mymax4[q_], z);
                                }
                        }
                    };
                    if constexpr (FAC) {
                        const float w0_ = __int_as_float(m[2].y), w1_ = __int_as_float(m[2].w);
                        output(__builtin_elementwise_fma(e[2], (f32x4){w0_, w0_, w0_, w0_}, acc) * sc4, m[0].x, m[0].y, e[0]);
                        if (m[1].x >= 0) output(__builtin_elementwise_fma(e[3], (f32x4){w1_, w1_, w1_, w1_}, acc) * sc4, m[1].x, m[1].y, e[1]);
                    } else {
                        output(acc * sc4, m[0].x, m[0].y, e[0]);
                    }
                });
        }
        for (int i = w0; i < p.st.b.nrest; i += NW) {
            const int r = __builtin_amdgcn_readfirstlane(p.st.b.rest[i]);
            const int s = __builtin_amdgcn_readfirstlane(g.brow_s[r]);
            int4 ds = g.brow[r];
            ds.x = __builtin_amdgcn_readfirstlane(ds.x); ds.y = __builtin_amdgcn_readfirstlane(ds.y);
            ds.z = __builtin_amdgcn_readfirstlane(ds.z); ds.w = __builtin_amdgcn_readfirstlane(ds.w);
            const bool one = (ds.w & 0x40000000) != 0;
            float e1 = 0.f;
            if (one && t >= 1) e1 = ep1[(size_t)(ds.w & 0xffff) * UL + ul];      // (requested before the arcs)
            const float bv = any_active ? bat_row_sum<UL, PERS>(g.barcs, ds.x, ds.y, Zc, ul, lane, aj) * sc : 0.f;
            if (t == 0) {
                const float st = p.start_lin[s];
                if (st != 0.f && active && aj == 0) atomicAdd(&p.zb[u], st * bv);
            } else if (active || starts) {
                const float out = active ? bv : p.end_lin[s] * pow2f(kScaleExp);
                if (one) {
                    const float z = e1 * out;
                    if (aj == 0) { BPt[(size_t)ds.z * UL + ul] = out; Zn.st1((size_t)ds.z * UL + ul, z); }
                    mymax = fmaxf(mymax, z);
                } else {
                    for (int kk = ds.z + aj; kk < ds.w; kk += ALR) {
                        const int4 pl = g.stp[kk];
                        BPt[(size_t)pl.x * UL + ul] = out;
                        const float z = ep1[(size_t)pl.y * UL + ul] * out;
                        Zn.st1((size_t)pl.x * UL + ul, z);
                        mymax = fmaxf(mymax, z);
                    }
                }
            }
        }
        if (c.lead) {
            if (starts) st_word<PERS>((unsigned *)p.Fb + u, (unsigned)kScaleExp);
            else if (active) st_word<PERS>((unsigned *)p.Fb + u, ld_word<PERS>((const unsigned *)p.Fb + u) + (unsigned)(k + kEpExp));
            st_word<PERS>(p.mxb + ((j + 2) % 3) * p.Bp + u, 0u);
        }
    }
    // persistent launch: the emission lines the NEXT frame starts with (its first three bundles' labels) are asked for now, so that they
    // sit in the L2 when that frame requests them behind the grid barrier (a new frame reads a new slab of e': a miss to the fabric
    // otherwise, on the frame's critical path); the values are not used -- they are waited for at the end of this function, where
    // every wave drains its stores anyway
    f32x4 warm[NM == 1 ? 3 : 6];
    bool warmed = false;
    if constexpr (PERS) {
        const int jn = j + 1;
        const int tn = dir == 0 ? jn : T - jn - 1;                 // the frame whose emissions iteration jn reads
        if (one_task && filled && tn >= 0 && tn < T) {
            const int nbund = __builtin_amdgcn_readfirstlane(c.tk0.w);
            constexpr int LGs = UL / 4, ALs = 64 / LGs;
            const int4 *mlds = (const int4 *)(ldsw + 2 * kStreamRecB + 2 * 64 * 16 * (NM == 1 ? 1 : 4));
            const f32x4 *et4 = (const f32x4 *)(p.ept + (size_t)tn * Vall + gV) + uq;
#pragma unroll
            for (int bd = 0; bd < 3; ++bd) {
                const int4 *m = mlds + ((size_t)(bd < nbund ? bd : 0) * ALs + sj) * NM;
                warm[(NM == 1 ? 1 : 2) * bd] = et4[(size_t)m[0].z * LGs];
                if constexpr (NM > 1) warm[2 * bd + 1] = et4[(size_t)m[1].z * LGs];
            }
            warmed = true;
        }
    }
    // maximum of the vector this launch wrote, per utterance: lanes -> LDS (the values are non-negative: their bits order
    // like unsigned integers) -> one atomic per utterance and workgroup
    if (mymax > 0.f) atomicMax(&umax[ul], __float_as_uint(mymax));
#pragma unroll
    for (int q_ = 0; q_ < 4; ++q_) if (mymax4[q_] > 0.f) atomicMax(&umax[4 * uq + q_], __float_as_uint(mymax4[q_]));
    __syncthreads();
    if (tid < UL) {
        const unsigned m = umax[tid];
        unsigned *slot = (dir == 0 ? p.mxf : p.mxb) + ((j + 1) % 3) * p.Bp + grp * UL + tid;
        if (m != 0u) atomicMax(slot, m);
    }
    if constexpr (PERS) {
        if (warmed) {
#pragma unroll
            for (int q_ = 0; q_ < (NM == 1 ? 3 : 6); ++q_) asm volatile("" : : "v"(warm[q_]));
        }
    }
    CRF_TM(tmi >= 0, tmi + 6);
}

// One frame per launch: the kernel boundary is the grid barrier and what makes the vectors visible across XCDs.  1-D grid of 8 * nslot
// workgroups; block b sits on XCD b % 8 (observed; a matter of speed only) and works for ONE combo = (utterance group, direction):
//   #combos <  8: XCD x serves combo x % #combos together with the other XCDs of that residue, the combo's workgroups
//                 ("chunks") dealt round-robin among them;
//   #combos >= 8: XCD x serves the combos x, x + 8, ..., its slots dealt round-robin among them.
template <int UL, int D, bool FAC>
__global__ __launch_bounds__(kBatThreads) void crf_batch_frame_kernel(BatchParams p) {
    __shared__ unsigned umax[UL];                                  // maximum of the vector this workgroup wrote, per utterance (float bits)
    __shared__ __attribute__((aligned(16))) char stage[kBatWaves][stream_lds<UL, FAC>()];   // bat_stream: records, value ring, descriptors of a wave's task
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (timing build: launch 700, four workgroups of the first XCD x their four waves, 16 stamps each from g_tm[14000])
    const int tmi = (p.j == 700 && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) % 24 == 0 && (blockIdx.x >> 3) < 96) ? 14000 + (((blockIdx.x >> 3) / 24) * 4 + wave) * 16 : -1;
    CRF_TM(tmi >= 0, tmi + 0);
    const BatWg<UL> c = bat_setup<UL>(p, lane, wave);
    bool filled = false;
    bat_frame<UL, D, FAC, false>(p, c, p.j, umax, stage[wave], tmi, filled);
}

// ALL frames in one launch (round 6): the same frame body, the launch boundary replaced by a grid barrier -- an XCD-hierarchical counter
// barrier (tools/grid_barrier_probe.hip: 2.2 us per frame empty at 256 workgroups, 3.6 us with a 1 MB sc1 exchange; one flat counter
// 3.8 / 5.9): every wave drains its own stores (they are write-through: performed when acknowledged), the workgroup's lane 0 arrives at
// its XCD's counter, the XCD's last arriver goes on to the top counter and, once all XCDs are there, publishes the XCD's generation word,
// which the others poll (relaxed sc1 loads, s_sleep).  Workgroups per XCD come from a census at the start of the launch (HW_REG_XCC_ID;
// any placement is correct, b % 8 is only the expected one).  What the per-frame launch paid 12 of its 16.8 us for -- cold instruction
// caches, an invalidated L2, three dependent trips to the fabric before the first gather (profiles/round2_r2w_timing_probe_batch.txt) --
// is paid once: the task descriptor stays in registers, arc records and row descriptors stay in the XCD's L2.
// The grid must be co-resident (the host checks it against the occupancy the runtime reports and falls back to one launch per frame
// otherwise); every spin is bounded, a time-out sets p.err (the costs of the call then come out as NaN) instead of hanging.
// bar: [0] census barrier | [16 + 16 x] arrivals of XCD x | [160 + 16 x] generation of XCD x | [300] top | [304 + x] workgroups on XCD x
__device__ __forceinline__ bool bat_poll_ge(unsigned *w, unsigned target, int *err) {
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(w, CRF_RLX_AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(1);
        if (spins > (1u << 22)) { __hip_atomic_store(err, 1, CRF_RLX_AGENT); return false; }   // ~seconds: a workgroup of the grid is not resident
    }
}
template <int UL, int D, bool FAC>
__global__ __launch_bounds__(kBatThreads, 2) void crf_batch_persist_kernel(BatchParams p) {   // (2: two workgroups per CU = two waves per SIMD -- VGPRs + AGPRs <= 256)
    __shared__ unsigned umax[UL];
    __shared__ unsigned bar_s[2];                                  // workgroups on this XCD, XCDs in use
    __shared__ __attribute__((aligned(16))) char stage[kBatWaves][stream_lds<UL, FAC>()];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    const BatWg<UL> c = bat_setup<UL>(p, lane, wave);
    unsigned *bar = p.bar;
    if (tid == 0) {                                                // census, then one flat barrier so that every count is final
        __hip_atomic_fetch_add(bar + 304 + xcc, 1u, CRF_RLX_AGENT);
        __hip_atomic_fetch_add(bar + 0, 1u, CRF_RLX_AGENT);
        bat_poll_ge(bar + 0, gridDim.x, p.err);
        unsigned n = 0;
        for (int x = 0; x < 8; ++x) n += __hip_atomic_load(bar + 304 + x, CRF_RLX_AGENT) != 0u;
        bar_s[0] = __hip_atomic_load(bar + 304 + xcc, CRF_RLX_AGENT); bar_s[1] = n;
    }
    __syncthreads();
    const unsigned nx = bar_s[0], nxcd = bar_s[1];
    bool filled = false;                                           // the wave's slice of LDS holds its task's records and row descriptors
    for (int j = 0; j <= p.T; ++j) {
        const int tmi = (j == 700 && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) % 24 == 0 && (blockIdx.x >> 3) < 96) ? 14000 + (((blockIdx.x >> 3) / 24) * 4 + wave) * 16 : -1;
        CRF_TM(tmi >= 0, tmi + 0);
#ifdef CRF_TIMING
        if (j == 700 && tid == 0 && blockIdx.x < 1000) g_tm[5000 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();   // frame start (100 MHz, device-wide)
#endif
        bat_frame<UL, D, FAC, true>(p, c, j, umax, stage[wave], tmi, filled);
        if (j == p.T) break;                                       // nothing in this launch reads what the last frame wrote
        // ---- the frame boundary ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every wave: its vector stores and maxima are performed
        CRF_TM(tmi >= 0, tmi + 7);
        __syncthreads();
        CRF_TM(tmi >= 0, tmi + 10);
#ifdef CRF_TIMING
        if (j == 700 && tid == 0 && blockIdx.x < 1000) g_tm[2000 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();   // every workgroup's arrival at the grid barrier
#endif
        if (tid == 0) {
            const unsigned f1 = (unsigned)(j + 1);
            const unsigned old = __hip_atomic_fetch_add(bar + 16 + 16 * xcc, 1u, CRF_RLX_AGENT);
            if (old + 1u == nx * f1) {                             // the XCD's last arriver
                __hip_atomic_fetch_add(bar + 300, 1u, CRF_RLX_AGENT);
                bat_poll_ge(bar + 300, nxcd * f1, p.err);
                __hip_atomic_store(bar + 160 + 16 * xcc, f1, CRF_RLX_AGENT);
            } else {
                bat_poll_ge(bar + 160 + 16 * xcc, f1, p.err);
            }
        }
        __syncthreads();
#ifdef CRF_TIMING
        if (j == 700 && tid == 0 && blockIdx.x < 1000) { g_tm[3000 + blockIdx.x] = __builtin_amdgcn_s_memrealtime(); g_tm[4000 + blockIdx.x] = xcc; }
#endif
        CRF_TM(tmi >= 0, tmi + 11);
    }
}


// zs[u] = sum_s a_{lx}[s][u] * end[s].  grid (ceil(S / (4 * 64)), 1, Bp / UL): a wave sums 64 states
template <int UL>
__global__ __launch_bounds__(kBatThreads) void crf_batch_zsum_kernel(BatchParams p) {
    constexpr int AL = 64 / UL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ul = lane % UL, aj = lane / UL, u = blockIdx.z * UL + ul;
    const int lxu = u < p.B ? p.lx[u] : 0;                         // a_lx sits in the buffer frame lx - 1 wrote: parity lx & 1
    const float *Af = p.Af + (size_t)(lxu & 1) * p.SX * p.Bp + (size_t)blockIdx.z * p.SX * UL;
    const int s0 = (blockIdx.x * kBatWaves + wave) * 64;
    float acc = 0.f;
    for (int s = s0 + aj; s < min(s0 + 64, p.S); s += AL) acc = fmaf(Af[(size_t)s * UL + ul], p.end_lin[s], acc);
    acc = arc_lane_sum<UL>(acc);
    if (aj == 0 && acc != 0.f) atomicAdd(&p.zs[u], acc);
}

// per utterance: costs from the scaled sums, the exponents and the per-frame offsets; flags for the robust fallback
__global__ __launch_bounds__(kBatThreads) void crf_batch_cost_kernel(BatchParams p) {
    __shared__ double red[kBatWaves];
    const int b = blockIdx.x, tid = threadIdx.x, lx = p.lx[b];
    double part = 0.0;
    for (int t = tid; t < lx; t += kBatThreads) part += (double)p.moff[(int64_t)b * p.T + t];
    part = wave_sum_d(part);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        const double mxs = red[0] + red[1] + red[2] + red[3];
        float zs = p.zs[b], zb = p.zb[b];
        int ef = p.Ef[b], fb = p.Fb[b];
        if (lx <= 0) {                                             // empty utterance: logZ = LSE(start + end)
            float z0 = 0.f;
            for (int s = 0; s < p.S; ++s) z0 += p.start_lin[s] * p.end_lin[s];
            zs = zb = z0 * pow2f(kScaleExp); ef = fb = kScaleExp;
        }
        p.den_zs[b] = zs; p.den_ez[b] = ef;
        const bool timed_out = *p.err != 0;                        // the persistent launch's grid barrier gave up: no result
        p.cost_alpha[b] = timed_out ? __builtin_nanf("") : to_log(zs, ef, mxs);
        p.cost_beta[b] = timed_out ? __builtin_nanf("") : to_log(zb, fb, mxs);
        if (!(zs > 0.f && zs < INFINITY)) p.redo[b] = 1;
        if (!(zb > 0.f && zb < INFINITY)) p.redo[p.B + b] = 1;
    }
}

// gamma_den[u][t][v] = u_v / sum_v u_v,  u_v = e'_t[v][u] * sum_{p: lab_p = v} Q_t[p][u] * BP_t[p][u]
// (each frame normalises itself).  grid (T, 1, Bp / UL); a wave takes the labels wave, wave + 4, ...; the un-normalised
// values go to the grad row first and are scaled by the same lanes afterwards.
template <int UL>
__global__ __launch_bounds__(kBatThreads) void crf_batch_grad_kernel(BatchParams p) {
    constexpr int AL = 64 / UL;
    __shared__ float wsum[kBatWaves][64], wemx[kBatWaves][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ul = lane % UL, aj = lane / UL, u = blockIdx.z * UL + ul;
    const int t = blockIdx.x, V = p.V, Bp = p.Bp;
    const bool real = u < p.B;
    const bool active = real && t < p.lx[u];
    const float *Qt = p.Q + (size_t)t * p.P * Bp + (size_t)blockIdx.z * p.P * UL, *Bt = p.BP + (size_t)t * p.P * Bp + (size_t)blockIdx.z * p.P * UL;
    const float *et = p.ept + (size_t)t * V * Bp + (size_t)blockIdx.z * V * UL;
    float *row = real ? p.grad + ((size_t)u * p.T + t) * V : nullptr;
    float part = 0.f, emx = 0.f;                                   // emx: the largest emission of a label the graph has (lost-term bound below)
    for (int v = wave; v < V; v += kBatWaves) {
        float acc = 0.f;
        const float ev = et[(size_t)v * UL + ul];
        if (v <= p.max_label && active) {
            const int p0 = p.g.lab_off[v], p1 = p.g.lab_off[v + 1];
#pragma unroll 4
            for (int q = p0 + aj; q < p1; q += AL) acc = fmaf(Qt[(size_t)q * UL + ul], Bt[(size_t)q * UL + ul], acc);
            acc = arc_lane_sum<UL>(acc);                          // (every arc lane of the utterance holds the sum)
            if (p1 > p0) emx = fmaxf(emx, ev);
        }
        const float uv = active ? (ev * pow2f(-kGradDescale)) * acc : 0.f;
        part += uv;
        if (aj == 0 && real) row[v] = uv;                          // un-normalised; 0 past the utterance's length
    }
    wsum[wave][lane] = part; wemx[wave][lane] = emx;
    __syncthreads();
    const float nrm = wsum[0][lane] + wsum[1][lane] + wsum[2][lane] + wsum[3][lane];
    const float em = fmaxf(fmaxf(wemx[0][lane], wemx[1][lane]), fmaxf(wemx[2][lane], wemx[3][lane]));
    const float inv = nrm > 0.f ? p.c_den / nrm : 0.f;
    if (aj != 0 || !active) return;
    // The utterance goes to the log-domain fallback when a frame's mass is not a normal, finite float, or when products the rows can no longer
    // hold could have mattered -- the bound of crf_grad_den_kernel (k_grad.hip, CRF_GD_CHECK): a pair whose q or b lies below 2^-126 is lost from
    // rows scaled to 2^20; such terms weigh at most e'max * 2^-4 * 2^-105 each, so a frame mass below e'max * 2^-82 could be missing more than
    // 1e-4 of itself.  This family had the first half only until round 6: a frame's posteriors 1 / 0 instead of 0.9953 / 0.0041 / 0.0006 with
    // network outputs 40 nats apart, found when the fuzz's seeds met this family (tests/test_gpu_fuzz.py seed 9 after the mode list grew).
    if (wave == 0 && !(nrm >= 0x1p-120f && nrm < INFINITY && nrm >= em * 0x1p-82f) && p.redo) p.redo[u] = 1;
    for (int v = wave; v < V; v += kBatWaves) row[v] *= inv;       // the lane's own stores: program order
}


// ---- explicit instantiations ----
#define CRF_INST_BAT(UL)                                                              \
    template __global__ void crf_batch_transpose_kernel<UL>(BatchParams);             \
    template __global__ void crf_batch_frame_kernel<UL, 4, false>(BatchParams);       \
    template __global__ void crf_batch_frame_kernel<UL, 4, true>(BatchParams);        \
    template __global__ void crf_batch_persist_kernel<UL, 4, false>(BatchParams);     \
    template __global__ void crf_batch_persist_kernel<UL, 4, true>(BatchParams);      \
    template __global__ void crf_batch_zsum_kernel<UL>(BatchParams);                  \
    template __global__ void crf_batch_grad_kernel<UL>(BatchParams);
CRF_INST_BAT(64)
CRF_INST_BAT(32)
CRF_INST_BAT(16)
CRF_INST_BAT(8)
#undef CRF_INST_BAT

}  // namespace crf
