// cat_amd/csrc/k_fac_body.h -- register-resident denominator recursions, factored layout (T o LM graphs): the DEFINITIONS of the dominant
// kernels.  Included by k_fac_1024.hip / k_fac_768.hip / k_fac_pair2.hip, which instantiate their share (three units: this family is what
// a rebuild waits for).
#pragma once
#include "crf_device.h"
#include "crf_kernels_decl.h"
#include "k_res_common.h"

namespace crf {

// =============================================================================================
// FACTORED resident recursions (crf_internal.h: FacDev, res_layout.cpp: build_factored): one compute unit
// per recursion and utterance, no exchange.  Same arithmetic and scaling as crf_res_chain_kernel, same arc
// format (4 gathers of one float + 4 weights per chunk); what differs is the row epilogue, which has TWO outputs:
//   forward : a row belongs to the main state s2 of a pair (s1, s2).  L' = e'[l2] * rowsum is a[s2];
//             A' = e'[l1] * w * U is a[s1] (its own row is exactly w * (a[s1] + a[s2]) = w * U); U' = A' + L'
//             is the entry every other row gathers for the pair.  Q row: [rowsum of each row | w*U of each row].
//   backward: a row is the common out-arc sum of one or two states; the epilogue adds each state's one
//             extra arc (BP positions / z entries 2*rid, 2*rid + 1).
// LDS: V0 | V1 (two state vectors of Gp floats) | row metadata int4[R] | EP[2][Vp] | wm | red
// =============================================================================================

// FLAG: publish stage flags and store rows write-through (one instantiation per use: the frame loop has no
// run-time switch for it)
// NTH threads with NCH chunks of arcs each, gathered in batches of NB chunks: 512 x 30 (2 waves per SIMD) or
// 768 x 21 (3 waves per SIMD at <= 168 VGPRs -- the frame is latency-bound, a third wave fills the gaps).
// ML: some rows are cut into pieces on adjacent lanes (graphs with long rows; a separate instantiation, the check costs the
// row epilogue of the others 2 %)
// K2: TWO CUs per recursion (graphs of 120 k - 240 k arcs).  Each CU holds half of the rows (its share of the arc registers)
// and the WHOLE state vector: what its row epilogues produce -- U', L', A' of a forward row, the two z of a backward row -- is
// also published as {frame tag, value} granules (res_chain_body's protocol: the data is the flag), and after its last chunk the
// CU fetches the peer's entries into its own vector before the frame barrier.  Table geometry (RL) only, one copy of the
// gathered entries, no stages (2 B x 2 workgroups are every CU of the device: nothing runs beside the recursions).
template <int DIR, bool FLAG, int NTH, int NCH, int NB, bool ML, bool RL, bool K2 = false>
__device__ __forceinline__ void fac_chain_body(const FacParams &p, float *lds, const int b, const int k = 0) {
    static_assert(!K2 || (RL && !FLAG), "two CUs per recursion: table geometry, no stage flags");
    constexpr int NW = NTH / kWave;
    // 768-thread geometry: the last chunk slot of a thread holds ROW CONSTANTS instead of arcs -- two words for each of
    // the (at most three) rows the lane finishes per frame -- and the entries of a row sit where its row id says
    // (res_layout.cpp, "implicit"): a row epilogue asks for everything it needs from LDS in ONE round trip.  With a
    // table of row constants in LDS it was a chain of three (constants -> the values they point to -> emissions),
    // ~320 cycles of a wave's time per slice, 70 % of the frame loop (timing build).
    // RL (768 threads): the row constants are a TABLE in LDS after all -- 8 (forward) / 16 (backward) bytes per row, read one
    // slice AHEAD (the next slice's constants are requested in the epilogue of this one and arrive behind its gathers), so an
    // epilogue is still one round trip, a wave may finish any number of slices per frame, and no select between registers
    // is needed (with three slices that select is ~10 VALU instructions per epilogue).  For graphs with many short rows.
    static_assert(!RL || NTH != kResThreads, "the row-constant table goes with the 768- and 1024-thread geometries");
    constexpr bool IMP = NTH != kResThreads;                 // entries of a row lie where its row id says
    constexpr bool RC = IMP && !RL;
    constexpr int NCHA = RC ? kFac3ArcCh : NCH;              // chunk slots that hold arcs (RC: the last slot holds the row constants)
    constexpr int RCW = NCHA * 6;                            // first row-constant word
    static_assert(!RC || NCH == kFac3NCH, "row constants in registers: 20 chunks of arcs + the constants' slot");

    const FacDirDev &L = p.L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int V = p.V, lx = p.lx[b], G = L.G, R = L.R;
    const unsigned dup = (unsigned)L.dup;                   // second copy of the gathered entries, other banks (res_layout.cpp pack_arcs)
    const int Vp = rup64(V + 1), Gp = rup64(G);
    const int XB = Gp * 4;                                   // bytes per state vector
    const int64_t bt0 = (int64_t)b * p.T;
    float *X = lds;                                          // [2][Gp]
    // row table.  K2: only the rows of this CU, [tr0, tr1) -- RMc is then the table's VIRTUAL base, so that "RMc + row bytes * rid"
    // addresses it by the global row id like everywhere else (the state vectors in front are longer than any such shift)
    constexpr int kRowB = RL ? (DIR == 0 ? 8 : 16) : 16;
    const int tr0 = K2 ? L.cu_row[k] : 0, tr1 = K2 ? L.cu_row[k + 1] : R;
    const int trn = K2 ? max(L.cu_row[1] - L.cu_row[0], L.cu_row[2] - L.cu_row[1]) : R;   // rows the table has room for (both CUs the same: fac_lds_bytes)
    char *RMc0 = (char *)(X + 2 * Gp);                       // int4[R]
    char *RMc = RMc0 - (size_t)tr0 * kRowB;
    float *EP = (float *)(RMc0 + (RL ? (size_t)(trn + 64) * kRowB : (size_t)R * 16));   // [2][Vp]  (RL: 64 rows of slack for the read-ahead)
    float *wm = EP + 2 * Vp;                                 // [2][NW]
    double *red = (double *)(wm + 2 * NW);            // [NW]
    if (tid == 0 && p.started && p.i0 == 0) atomicAdd(p.started, 1);   // this workgroup holds its CU: see crf_gate_kernel
    [[maybe_unused]] const bool lead = !K2 || k == 0;        // the CU that writes what exists once per recursion (exponents, logZ)
    [[maybe_unused]] gu64 *xchd = nullptr;                   // K2: this recursion's granules, [2 slots][G]
    [[maybe_unused]] gu64 *hs = nullptr;                     // K2: two handshake words of this recursion
    [[maybe_unused]] bool same_l2 = false;
    if constexpr (K2) {
        xchd = (gu64 *)p.xch + (DIR == 0 ? 0 : (size_t)p.B * 2 * (size_t)p.Gf) + (size_t)b * 2 * (size_t)G;
        hs = (gu64 *)p.xch + (size_t)p.B * 2 * ((size_t)p.Gf + (size_t)p.Gb) + ((size_t)DIR * p.B + b) * kResMaxK;
    }

    unsigned A[(NCH * 6)];
    unsigned rc00 = 0, rc01 = 0, rc10 = 0, rc11 = 0, rc20 = 0, rc21 = 0;   // row constants of the lane's (up to) three rows: scalars,
    {                                                                      // not array elements (a select between array elements
        const unsigned *src = L.arcs + (K2 ? (size_t)k * (NCH * 6) * NTH : (size_t)0) + tid;   // becomes a variable index and the array leaves the registers)
#pragma unroll
        for (int i = 0; i < (RC ? RCW : NCH * 6); ++i) A[i] = src[(size_t)i * NTH];
        if (RC) {
            rc00 = src[(size_t)(RCW + 0) * NTH]; rc01 = src[(size_t)(RCW + 1) * NTH];
            rc10 = src[(size_t)(RCW + 2) * NTH]; rc11 = src[(size_t)(RCW + 3) * NTH];
            rc20 = src[(size_t)(RCW + 4) * NTH]; rc21 = src[(size_t)(RCW + 5) * NTH];
        }
    }
    const uint4 wi = L.wave_info[(K2 ? k * NW : 0) + wave];
    const unsigned ends = __builtin_amdgcn_readfirstlane(wi.x);
    const int nch = __builtin_amdgcn_readfirstlane(wi.y);
    const int row0 = __builtin_amdgcn_readfirstlane(wi.z);
    const unsigned lgbits = __builtin_amdgcn_readfirstlane(wi.w);   // 3 bits per slice: its rows are cut into 2^lg pieces on adjacent lanes
    if (RL) {
        if (DIR == 0) {
            uint2 *RT = (uint2 *)RMc;                        // {emission byte offsets main | tail << 16, tail weight}
            for (int r = tr0 + tid; r < tr1; r += NTH) {
                const int4 m = p.frow_meta[r];
                RT[r] = uint2{(((unsigned)m.x >> 16) * 4u) | (((unsigned)m.w * 4u) << 16), (unsigned)m.z};
            }
        } else {
            uint4 *RT = (uint4 *)RMc;                        // {z byte offsets of the two extra arcs, emission byte offsets, their weights}
            for (int r = tr0 + tid; r < tr1; r += NTH) {
                const int4 m = p.brow_meta[r];
                const unsigned l0 = (unsigned)m.w & 0xffffu, l1 = (unsigned)m.w >> 16;   // 0xffff = no label: emission 0 at EP[V]
                RT[r] = uint4{(unsigned)m.x, ((l0 == 0xffffu ? (unsigned)V : l0) * 4u) | (((l1 == 0xffffu ? (unsigned)V : l1) * 4u) << 16), (unsigned)m.y, (unsigned)m.z};
            }
        }
    } else if (!RC) {
        int4 *RM = (int4 *)RMc;
        for (int r = tid; r < R; r += NTH) {
            int4 m = DIR == 0 ? p.frow_meta[r] : p.brow_meta[r];
            if (DIR == 1) {
                const int l0 = (short)(m.w & 0xffff), l1 = m.w >> 16;       // -1 = no label: emission 0 at EP[V]
                m.w = ((l0 < 0 ? V : l0) & 0xffff) | ((l1 < 0 ? V : l1) << 16);
            }
            RM[r] = m;
        }
    } else if (DIR == 1) {
        f32x2 *RW = (f32x2 *)RMc;                            // the weights of the two extra arcs of a row
        for (int r = tid; r < R; r += NTH) { const int4 m = p.brow_meta[r]; RW[r] = f32x2{__int_as_float(m.y), __int_as_float(m.z)}; }
        auto fix = [&](unsigned w) {                         // emission byte offsets: 0xffff = no label -> emission 0 at EP[V]
            const unsigned l0 = w & 0xffffu, l1 = w >> 16;
            return (l0 == 0xffffu ? (unsigned)V * 4u : l0) | ((l1 == 0xffffu ? (unsigned)V * 4u : l1) << 16);
        };
        rc01 = fix(rc01); rc11 = fix(rc11); rc21 = fix(rc21);
    }
    // One launch runs the iterations [i0, i1) of the recursion ("segment"): the host cuts a long recursion into
    // a few launches so that the grad pass can be released stage by stage with stream events (a kernel that
    // waits for another kernel's progress is not safe, see crf_loss_fwd_bwd).  Between launches the state --
    // the current vector and its exponent -- rests in HBM (p.state: [B][Gp + 64] floats per direction).
    const int i0 = p.i0, i1 = min(p.i1, lx);
    if (i0 > 0 && i0 >= lx) return;                          // this utterance was finished by an earlier segment
    float *state = p.state + (size_t)b * (Gp + 64);
    const int par0 = i0 & 1;
    int E = kScaleExp;
    float zpart = 0.f;
    for (int s = tid; s < 2 * Gp; s += NTH) X[s] = 0.f;
    if (tid < 2) EP[tid * Vp + V] = 0.f;
    // The frame maximum is FOUR LDS words per frame (three sets in rotation: read / accumulated by ds_max_f32 / cleared), one per
    // row of 16 lanes: a wave's frame top reads them with one ds_read_b128 and works the scale out on the scalar unit; with twelve wave maxima per frame every wave spent ~17 VALU instructions there and ~8 more in the tail --
    // the frame is bound by instruction issue (timing build: a wave with NO rows still took 520 cycles per frame).
    if (tid < 12) wm[tid] = 0.f;                             // [3 frames][4 rows of 16 lanes]: distinct addresses per lane (a same-address
                                                             // LDS atomic of several lanes is turned into a scalar loop by the compiler)
    const bool rowlead = (lane & 15) == 0;
    int sr = i0 % 3;                                         // word read by the next frame
    if (lx > 0)
        for (int v = tid; v < V; v += NTH) {
            if (DIR == 0) EP[par0 * Vp + v] = p.ep[(bt0 + i0) * V + v];                 // e'_t of the first frame
            else {
                const int t = lx - 1 - i0;                                             // first frame of this segment
                if (i0 == 0) EP[v] = p.ep[(bt0 + lx - 1) * V + v];                     // for the initial z vector
                if (t >= 1) EP[(1 - par0) * Vp + v] = p.ep[(bt0 + t - 1) * V + v];     // e'_{t-1}
            }
        }
    __syncthreads();
    {
        float m0 = 0.f;
        if (i0 > 0) {                                        // resume
            float *Xc = X + par0 * Gp;
            for (int s = tid; s < G; s += NTH) { const float v = state[s]; Xc[s] = v; m0 = fmaxf(m0, v); }
            E = __builtin_amdgcn_readfirstlane(__float_as_int(state[Gp]));   // (uniform: E lives on the scalar unit)
        } else if (DIR == 0) {
            for (int s = tid; s < G; s += NTH) { const float v = p.x_start[s] * pow2f(kScaleExp); X[s] = v; m0 = fmaxf(m0, v); }
        } else if (lx > 0) {
            for (int z = tid; z < G; z += NTH) {
                const int l = p.z_lab[z];
                const float v = EP[l < 0 ? V : l] * (p.z_end[z] * pow2f(kScaleExp));
                X[z] = v; m0 = fmaxf(m0, v);
            }
            float *BProw = p.Out + (bt0 + lx - 1) * p.Rout;
            if (lead) for (int r = tid; r < 2 * R; r += NTH) BProw[r] = p.brow_end[r] * pow2f(kScaleExp);
            if (tid == 0 && lead) p.Eout[bt0 + lx - 1] = E;
        } else {
            if (lead) for (int r = tid; r < 2 * R; r += NTH) zpart += p.brow_start[r] * p.brow_end[r] * pow2f(kScaleExp);
        }
        m0 = row_max16(m0);
        if (rowlead) lds_fmax(wm + sr * 4 + (lane >> 4), m0);
    }
    __syncthreads();
    if constexpr (K2) {
        // where is my peer?  (res_chain_body: both publish the id of their XCD, write-through; both on one XCD => plain stores
        // into the shared L2 are enough for the hand-off, otherwise write-through stores.  Both take the decision from the same
        // two ids.)
        const unsigned my_xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;  // HW_REG_XCC_ID[3:0]
        if (tid == 0) __hip_atomic_store(hs + k, (1ull << 32) | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int same = 1;
        if (tid == 0) {
            unsigned long long g = 0;
            for (unsigned spins = 0; (g >> 32) != 1ull; ++spins) {
                g = __hip_atomic_load(hs + (1 - k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (spins > (1u << 22)) { __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                if ((spins & 255u) == 255u && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                __builtin_amdgcn_s_sleep(2);
            }
            same = ((unsigned)g & 0xf) == my_xcc && (g >> 32) == 1ull;
        }
        same_l2 = __syncthreads_and(same) != 0;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // arcs and tables have landed (see crf_res_chain_kernel)

    // Stage flags (p.nb > 1): when the recursion reaches iteration bound[k], the rows of all earlier iterations
    // are made visible device-wide (they are stored write-through; every wave drains its stores, then a barrier)
    // and a counter in fine-grained memory is bumped.  The host has queued the grad launch of stage k behind a
    // STREAM-level wait on that counter (hipStreamWaitValue32: the command processor polls, no wave spins, the
    // launch is not even dispatched before) -- so the grad pass follows the recursions without a relaunch.
    constexpr bool flagged = FLAG;
    int next_stage = 1;
    int next_bound = (FLAG && p.nb > 1) ? p.bound[1] : 0x7fffffff;   // iteration at which the next stage is published
    auto publish_stage = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(p.stage_cnt + next_stage, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ++next_stage;
        next_bound = next_stage < p.nb ? p.bound[next_stage] : 0x7fffffff;
    };
    const bool pre_w = wave * kWave < V;                     // this wave holds emissions
    float last_sc = 1.f;                                     // scale of the last frame (rowless states, after the loop)
    // lagged scale (CRF_X_LAG): exponent of the NEXT frame's scale, worked out in the tail of the frame before; the first frame of a
    // launch takes the unlagged rule on the vector it starts from (any integer is exact).  lag_lo: the smallest scaled maximum seen (below kLagLow: fallback).
    constexpr bool LAG = CRF_X_LAG != 0 && !K2;
    // The maximum read in the tail of frame t travels to frame t+1 in a VECTOR register (lag_mx, the same bits in every lane) and is turned
    // into the scale there, BEHIND the frame's first gathers: lag_k is the exponent of the scale of the frame before.  First frame of a
    // launch: lag_mx = the maximum of the vector it starts from and lag_k = kLagTarget - kScaleExp give exactly the unlagged rule.
    [[maybe_unused]] int lag_k = kLagTarget - kScaleExp, lag_lo = 0x7fffffff, lag_mx = 0;
    if constexpr (LAG) {
        typedef int i32x4_t __attribute__((ext_vector_type(4)));
        const i32x4_t m4 = *(const i32x4_t *)(wm + sr * 4);
        lag_mx = max(max(m4.x, m4.y), max(m4.z, m4.w));
    }
    constexpr bool GFIRST = CRF_X_GFIRST != 0 && CRF_X_EARLY != 0 && !K2;
    constexpr int EPR = NTH >= 2 * kResThreads ? 1 : kEpRegsR;   // (V <= 2 * 512 everywhere: use_factored)
    float epn[EPR] = {};                                    // next emission row, in flight across the frame (waves that hold emissions only)
    // this utterance's emissions, rows and exponents (the frame loop adds 32-bit offsets: one s_mul instead of a 64-bit product per address)
    const float *ep_b = p.ep + bt0 * V;
    float *Out_b = p.Out + bt0 * p.Rout;
    int *Eo_b = p.Eout + bt0;
    auto frame = [&](const int par, int i) __attribute__((always_inline)) {
        [[maybe_unused]] const bool tm_on = b == 3 && i >= 100 && i < 228 && wave == 0;
        [[maybe_unused]] const int tm_i = (DIR * 4) * 1024 + (i - 100) * 8;
        CRF_TM(tm_on, tm_i + 0);
#ifdef CRF_TIMING
        if (b == 3 && i >= 150 && i < 158 && wave == 0) CRF_TM(true, 12288 + 1024 + (DIR * 4) * 8 + (i - 150));  // frame start
#endif
        const char *xb = (const char *)lds + par * XB;
        // GFIRST: the frame's first batch of gathers needs the source vector and nothing else -- it is requested before anything else is
        // worked out (every wave: idle waves hold padding arcs, offset 0 and weight 0).  A wave issues one instruction per ~4 cycles whatever
        // it is (tools/ubench_issue.py), and the ~35 scalar instructions a frame used to begin with kept the LDS idle for ~200 cycles right
        // behind every barrier (timing build, round 5: "frame top" 240 cycles with nothing to wait for).
        constexpr int NB0 = NB < NCHA ? NB : NCHA;
        [[maybe_unused]] f32x2 g01_0[NB0], g23_0[NB0];
        if constexpr (GFIRST) {
#if CRF_X_PRIO
            __builtin_amdgcn_s_setprio(3);
#endif
            CRF_RES_GATHER_N(g01_0, g23_0, A, xb, 0, NB0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int t = DIR == 0 ? i : lx - 1 - i;
        if (FLAG && i == next_bound) publish_stage();
        char *xnb = (char *)lds + (1 - par) * XB;
        const float *EPu = EP + (DIR == 0 ? par : 1 - par) * Vp;     // e'_t (fwd) / e'_{t-1} (bwd)
        const int tpre = DIR == 0 ? t + 1 : t - 2;
        // (only the waves that hold emissions take part in the prefetch: the compiler waits for vmcnt(0) around these
        // loads -- i.e. for the acknowledgement of the previous frame's row stores -- and the other waves need not)
        // The frame's side jobs belong to a few waves -- the emission prefetch to the waves that hold emissions, the exponent store
        // and the clearing of the maximum words to wave 0 -- and are tested as UNIFORM, UNLIKELY conditions: the other waves fall
        // through.  (As lane conditions they were exec-masked blocks that every other wave JUMPED over, ~32 cycles of instruction
        // refetch per taken branch and wave, six of them before the frame's first gather: tools/ubench_issue.py.)
        // (NOT the prefetch: as a cold block its register -- live across the whole frame -- is what the allocator spills first, and a
        // reload from scratch waits for vmcnt(0), i.e. for the frame's row stores: the S = 513 graph went from 1.84 to 2.35 ms)
        const bool pre = pre_w && (DIR == 0 ? (t + 1 < lx) : (t >= 2));
        if (pre) {
            const float *er = ep_b + (unsigned)tpre * (unsigned)V;   // (32-bit products: B * T * max(V, Rout) floats per utterance < 2^32, checked by the host)
#pragma unroll
            for (int q = 0; q < EPR; ++q) { const int v = tid + q * NTH; if (v < V) epn[q] = er[v]; }
        }
        const int sw = sr == 2 ? 0 : sr + 1, sz = sw == 2 ? 0 : sw + 1;   // accumulated during this frame / cleared in it
        typedef int i32x4_t __attribute__((ext_vector_type(4)));
        [[maybe_unused]] i32x4_t m4{};
        if constexpr (!LAG) m4 = *(const i32x4_t *)(wm + sr * 4);    // (non-negative floats: their bits order like integers)
        // The frame's scale and exponent are worked out BEHIND the first batch of gathers (EARLY: the batch loop calls `bookkeeping`
        // once its first gathers are requested -- they need nothing but the vector; the scale enters in the row epilogues only):
        // with the scale first, every wave of the workgroup sat out one LDS round trip right after the frame barrier, the LDS idle.
        float sc = 1.f;
        float *Orow = nullptr;
        auto bookkeeping = [&]() __attribute__((always_inline)) {
            // (the maxima's use ends up in FRONT of the first gathers all the same -- the scheduler gives their four registers to the gathers' addresses -- so
            // the frame still starts with one LDS round trip.  Reading them by inline asm BEHIND the gathers instead: metric step 2.833 vs 2.848 ms, but the
            // S = 513 graph 2.012 vs 1.978 -- dropped, profiles/round4_ab_waits_found_in_the_isa.txt)
            int ksc;
            if constexpr (LAG) {
                // lag_mx: maximum of the source vector of the frame BEFORE (read in that frame's tail), lag_k: that frame's scale
                const unsigned bits = (unsigned)__builtin_amdgcn_readfirstlane(lag_mx);
                int u = (int)(bits >> 23) - 127 + lag_k;             // exponent of the scaled maximum of that frame's source
                ksc = kLagTarget - u;
                // (all on the scalar unit; lag_lo: the smallest u of the recursion -- an all-zero vector lands far below kLagLow too, it is flagged at the end anyway)
                const int uu = i == i0 ? 0x7fffffff : u;           // (the first frame's u is the start-up convention above, not a measurement)
                asm("s_min_i32 %0, %0, %1" : "+s"(lag_lo) : "s"(uu) : "scc");
                asm("s_max_i32 %0, %0, %1\n\ts_min_i32 %0, %0, %2" : "+s"(ksc) : "s"(-100), "s"(100) : "scc");
                lag_k = ksc;
            } else ksc = rescale_exp_bits_uniform((unsigned)__builtin_amdgcn_readfirstlane(max(max(m4.x, m4.y), max(m4.z, m4.w))));   // (uniform)
            sc = pow2f(ksc);
            if (DIR == 1) last_sc = sc;
            if (DIR == 0) {
                E += ksc;
                if (__builtin_expect(wave == 0, 0)) {
                    if (lane < 4) wm[sz * 4 + lane] = 0.f;
                    if (tid == 0 && lead) Eo_b[t] = E;
                }
                E += kEpExp;
                Orow = Out_b + (unsigned)t * (unsigned)p.Rout;
            } else {
                E += ksc + kEpExp;
                if (__builtin_expect(wave == 0, 0)) {
                    if (lane < 4) wm[sz * 4 + lane] = 0.f;
                    if (t > 0 && tid == 0 && lead) Eo_b[t - 1] = E;
                }
                Orow = t > 0 ? Out_b + (unsigned)(t - 1) * (unsigned)p.Rout : p.Row0 + (int64_t)b * p.Rout;
            }
        };
        constexpr bool EARLY = CRF_X_EARLY != 0;   // (also with two CUs per recursion: H = 3 072 recursions 4.22 -> 4.04 ms)
        if constexpr (!EARLY) bookkeeping();
        unsigned ends_f = ends;
        int nch_f = nch;
        asm volatile("" : "+s"(ends_f), "+s"(nch_f));
        f32x2 acc = {0.f, 0.f};
        float mymax = 0.f;
        CRF_TM(tm_on, tm_i + 1);
        unsigned r4 = (unsigned)(row0 + lane) * 4u;   // 4 * row id
        constexpr bool EARLY_ = CRF_X_EARLY != 0;
        [[maybe_unused]] gu64 *slot = nullptr;        // K2: the granules of the vector this frame produces
        [[maybe_unused]] const unsigned tag = (unsigned)(i + 1);
        if constexpr (K2) slot = xchd + (size_t)(1 - par) * G;
        typedef std::conditional_t<DIR == 0, uint2, uint4> rct_t;
        [[maybe_unused]] rct_t kc{};                  // RL: constants of the slice that ends next
        // (KCLATE: the first slice's constants are requested BEHIND the first batch of gathers -- sixteen waves' 8- / 16-byte-per-lane reads
        // in front of them keep the LDS busy for 64 / 128 cycles right behind the barrier before the first gather is served)
        constexpr bool KCLATE = CRF_X_KCLATE != 0 && RL && EARLY_ && !K2;
        if constexpr (RL && !KCLATE) kc = *(const rct_t *)(RMc + (DIR == 0 ? 2u : 4u) * r4);
        // the end of a slice (row): everything between the row's sum and its entries of the next vector
        auto row_end = [&](const unsigned ks) __attribute__((always_inline)) {   // ks: slice number (uniform)
            // rows longer than a lane's registers lie on 2^lg adjacent lanes (res_layout.cpp place_rows): a butterfly
            // leaves the row's sum in every lane of the group, the first one owns the outputs
            float tot = acc.x + acc.y;
            if constexpr (ML) {
                const unsigned lg = ks < 10u ? (lgbits >> (3u * ks)) & 7u : 0u;   // (ten 3-bit fields; later slices have whole rows)
                if (lg) {
                    // DPP for groups of up to 16 lanes (pair swap, quad half swap, mirror of 8, mirror of 16: after
                    // each step every lane of the growing group holds the group's sum, so ANY lane of the other half
                    // will do); a __shfl_xor is a ds_bpermute round trip (~100+ cycles each, dependent) and cost the
                    // graphs with long rows -- every den_lm estimated from text -- a quarter of the frame
#define CRF_DPP_ADD(ctrl) tot += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tot), ctrl, 0xf, 0xf, false))
                    CRF_DPP_ADD(0xB1);                            // quad_perm [1,0,3,2]
                    if (lg >= 2) CRF_DPP_ADD(0x4E);               // quad_perm [2,3,0,1]
                    if (lg >= 3) CRF_DPP_ADD(0x141);              // row_half_mirror
                    if (lg >= 4) CRF_DPP_ADD(0x140);              // row_mirror
#undef CRF_DPP_ADD
                    if (lg >= 5) tot += __shfl_xor(tot, 16, 64);
                    if (lg >= 6) tot += __shfl_xor(tot, 32, 64);
                }
            }
            if constexpr (IMP) {
                unsigned k0, k1;
                [[maybe_unused]] f32x2 wrl{};
                if constexpr (RC) {
                    // (masks, not ?: -- the compiler turns a three-way select of registers by a uniform index
                    // into an indexed array, which it then cannot keep in registers)
                    const unsigned s0 = 0u - (unsigned)(ks == 0), s1 = 0u - (unsigned)(ks == 1), s2 = 0u - (unsigned)(ks >= 2);
                    k0 = (rc00 & s0) | (rc10 & s1) | (rc20 & s2);
                    k1 = (rc01 & s0) | (rc11 & s1) | (rc21 & s2);
                } else {
                    k0 = kc.x; k1 = kc.y;
                    if constexpr (DIR == 1) wrl = f32x2{__uint_as_float(kc.z), __uint_as_float(kc.w)};
                    kc = *(const rct_t *)(RMc + (DIR == 0 ? 2u : 4u) * (r4 + kWave * 4u));   // the next slice's (64 rows of slack behind the table)
                }

                if (DIR == 0) {   // k0 = main label | tail label << 16, k1 = tail weight; U, L, A at rid, R + rid, 2R + rid
                    const float uold = *(const float *)(xb + r4);                   // U_t of the row's pair
                    const float em = *(const float *)((const char *)EPu + (k0 & 0xffffu)), et = *(const float *)((const char *)EPu + (k0 >> 16));   // (byte offsets)
                    const float rv = tot * sc;                                      // q_t[pair of the main state]
                    const float qt = __uint_as_float(k1) * uold * sc;               // q_t[pair of the tail state]
                    if (flagged) {
                        __hip_atomic_store((unsigned *)((char *)Orow + r4), __float_as_uint(rv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store((unsigned *)((char *)Orow + r4 + 4u * (unsigned)R), __float_as_uint(qt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        *(float *)((char *)Orow + r4) = rv;
                        *(float *)((char *)Orow + r4 + 4u * (unsigned)R) = qt;
                    }
                    const float Lp = em * rv, Ap = et * qt, Up = Ap + Lp;           // a_{t+1}[main], [tail], their sum
                    if constexpr (CRF_X_ADDTID != 0 && !K2) {
                        // the four entries lie at (uniform base) + 4 * lane: ds_write_addtid_b32 takes the base from M0 and no address VGPR -- half the
                        // VGPR -> LDS traffic of a ds_write_b32 and no v_add per store (M0 reaches the whole LDS: tools/addtid_probe)
                        const unsigned mb = (unsigned)(uintptr_t)xnb + (unsigned)__builtin_amdgcn_readfirstlane((int)r4);
                        lds_st_addtid(Up, mb); lds_st_addtid(Up, mb + dup); lds_st_addtid(Lp, mb + 4u * (unsigned)R); lds_st_addtid(Ap, mb + 8u * (unsigned)R);
                    } else {
                    *(float *)(xnb + r4) = Up;
                    *(float *)(xnb + r4 + dup) = Up;
                    *(float *)(xnb + r4 + 4u * (unsigned)R) = Lp;
                    *(float *)(xnb + r4 + 8u * (unsigned)R) = Ap;
                    }
                    if constexpr (K2) {           // entries rid, R + rid, 2 R + rid of the peer's vector
                        gu64 *gs = (gu64 *)((char *)slot + 2u * r4);
                        res_publish(gs, 0, tag, Up, same_l2); res_publish(gs, R, tag, Lp, same_l2); res_publish(gs, 2 * R, tag, Ap, same_l2);
                    }
                    mymax = __int_as_float(max(__float_as_int(mymax), __float_as_int(Up)));   // (non-negative: bits order like integers; fmaxf canonicalises first)
                } else {          // k0 = z offsets of the two extra arcs, k1 = label 0 | label 1 << 16
                    const float z0 = *(const float *)(xb + (k0 & 0xffffu)), z1 = *(const float *)(xb + (k0 >> 16));
                    const float e0 = *(const float *)((const char *)EPu + (k1 & 0xffffu)), e1 = *(const float *)((const char *)EPu + (k1 >> 16));
                    const f32x2 w01 = RL ? wrl : *(const f32x2 *)(RMc + 2u * r4);
                    const float craw = tot;                                         // common out-arcs of the row's states
                    f32x2 bv;                                                        // b_t of the two states
                    bv.x = fmaf(w01.x, z0, craw) * sc;
                    bv.y = fmaf(w01.y, z1, craw) * sc;
                    if (flagged)
                        __hip_atomic_store((unsigned long long *)((char *)Orow + 2u * r4),
                                           (unsigned long long)__float_as_uint(bv.x) | ((unsigned long long)__float_as_uint(bv.y) << 32),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        *(f32x2 *)((char *)Orow + 2u * r4) = bv;
                    f32x2 zv;                                                        // z_{t-1} of the pairs entering them
                    zv.x = e0 * bv.x;
                    zv.y = e1 * bv.y;
                    *(f32x2 *)(xnb + 2u * r4) = zv;
                    typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
                    *(f32x2u *)(xnb + 2u * r4 + dup) = zv;
                    if constexpr (K2) {           // entries 2 rid, 2 rid + 1
                        gu64 *gs = (gu64 *)((char *)slot + 4u * r4);
                        res_publish(gs, 0, tag, zv.x, same_l2); res_publish(gs, 1, tag, zv.y, same_l2);
                    }
                    mymax = __int_as_float(max(__float_as_int(mymax), max(__float_as_int(zv.x), __float_as_int(zv.y))));
                }
            } else {
            const int4 m = *(const int4 *)(RMc + 4u * r4);
            if (DIR == 0) {
                const float uold = *(const float *)(xb + (m.x & 0xffff));   // U_t of the row's pair
                const float em = EPu[(unsigned)m.x >> 16], et = EPu[m.w];   // (requested together: one LDS round trip)
                const float rv = tot * sc;                                  // q_t[pair of the main state]
                const float qt = __int_as_float(m.z) * uold * sc;           // q_t[pair of the tail state]
                if (flagged) {   // write-through: the grad pass reads the rows from other XCDs while this kernel runs
                    __hip_atomic_store((unsigned *)((char *)Orow + r4), __float_as_uint(rv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store((unsigned *)((char *)Orow + r4 + 4u * (unsigned)R), __float_as_uint(qt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    *(float *)((char *)Orow + r4) = rv;
                    *(float *)((char *)Orow + r4 + 4u * (unsigned)R) = qt;
                }
                const float Lp = em * rv;                                   // a_{t+1}[main]
                const float Ap = et * qt;                                   // a_{t+1}[tail]
                const float Up = Ap + Lp;
                *(float *)(xnb + (m.x & 0xffff)) = Up;
                if (dup) *(float *)(xnb + (m.x & 0xffff) + dup) = Up;
                *(float *)(xnb + (m.y & 0xffff)) = Lp;
                *(float *)(xnb + ((unsigned)m.y >> 16)) = Ap;
                mymax = fmaxf(mymax, Up);
            } else {
                const float craw = tot;                                     // common out-arcs of the row's states
                const float z0 = *(const float *)(xb + (m.x & 0xffff)), z1 = *(const float *)(xb + ((unsigned)m.x >> 16));
                const float e0 = EPu[m.w & 0xffff], e1 = EPu[(unsigned)m.w >> 16];
                f32x2 bv;                                                    // b_t of the two states
                bv.x = fmaf(__int_as_float(m.y), z0, craw) * sc;
                bv.y = fmaf(__int_as_float(m.z), z1, craw) * sc;
                if (flagged)
                    __hip_atomic_store((unsigned long long *)((char *)Orow + 2u * r4),
                                       (unsigned long long)__float_as_uint(bv.x) | ((unsigned long long)__float_as_uint(bv.y) << 32),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else
                    *(f32x2 *)((char *)Orow + 2u * r4) = bv;
                f32x2 zv;                                                    // z_{t-1} of the pairs entering them
                zv.x = e0 * bv.x;
                zv.y = e1 * bv.y;
                *(f32x2 *)(xnb + 2u * r4) = zv;
                typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
                if (dup) *(f32x2u *)(xnb + 2u * r4 + dup) = zv;   // the copy's distance is an odd number of floats: ds_write2_b32
                mymax = fmaxf(mymax, fmaxf(zv.x, zv.y));
            }
            }
            acc = f32x2{0.f, 0.f};
            r4 += kWave * 4u;
        };
#pragma unroll
        for (int c0 = 0; c0 < NCHA; c0 += NB) {
            const int nb = NCHA - c0 < NB ? NCHA - c0 : NB;   // (the last batch may be short: 21 chunks in batches of 4)
#if CRF_X_PRIO
            // least progress first: a wave's issue priority (s_setprio) falls as it advances through its chunks, so that the four
            // waves of a SIMD reach the frame barrier together.  Without it the SIMD issues its OLDEST wave first: in the timing
            // build the oldest waves were through their chunks at 2 560 cycles and the youngest -- alone on their SIMDs at the end,
            // one wave's latency hiding -- at 4 140 of a 4 700-cycle frame; with it 3 050 ... 3 790 of 4 350.  (Round 2 had tried
            // STATIC priorities for the younger waves: slower.)
            {
                constexpr int f1 = CRF_X_PRIO == 2 ? 4 : CRF_X_PRIO == 3 ? 1 : 2, f2 = CRF_X_PRIO == 2 ? 6 : CRF_X_PRIO == 3 ? 2 : 4, f3 = CRF_X_PRIO == 2 ? 7 : CRF_X_PRIO == 3 ? 4 : 6;   // eighths of the chunks
                constexpr int qlast = (NCHA - 1) / NB * NB;   // start of the last batch: a threshold rounded up beyond it would never be reached (768 x 20 in batches of 4: 7/8 -> 20)
                constexpr int q1 = (f1 * NCHA / 8 + NB - 1) / NB * NB, q2 = (f2 * NCHA / 8 + NB - 1) / NB * NB,
                              q3 = (f3 * NCHA / 8 + NB - 1) / NB * NB < qlast ? (f3 * NCHA / 8 + NB - 1) / NB * NB : qlast;
                // (not with two CUs per recursion: a wave that polls for the peer's entries at the lowest priority delays BOTH CUs -- H = 3 072 recursions 4.17 without, 4.24 ms with)
                if constexpr (!K2) {
                    if (c0 == 0) { if constexpr (!GFIRST) __builtin_amdgcn_s_setprio(3); }
                    else if (c0 == q1) __builtin_amdgcn_s_setprio(2);
                    else if (c0 == q2) __builtin_amdgcn_s_setprio(1);
                    else if (c0 == q3) __builtin_amdgcn_s_setprio(0);
                }
            }
#endif
            // (EARLY: the first batch is gathered by every wave -- the slots of a wave without arcs hold padding, offset 0 and weight 0)
            if ((EARLY && c0 == 0) || c0 < nch_f) {
                f32x2 g01[NB], g23[NB];
                if (GFIRST && c0 == 0) {
#pragma unroll
                    for (int ci = 0; ci < NB0; ++ci) { g01[ci] = g01_0[ci]; g23[ci] = g23_0[ci]; }
                } else {
                    CRF_RES_GATHER_N(g01, g23, A, xb, c0, nb);
                }
                if constexpr (KCLATE) {
                    if (c0 == 0) {
                        asm volatile("" ::: "memory");   // (keeps the read behind the gathers in the instruction stream)
                        kc = *(const rct_t *)(RMc + (DIR == 0 ? 2u : 4u) * r4);
                    }
                }
                if constexpr (EARLY) { if (c0 == 0) bookkeeping(); }
#pragma unroll
                for (int ci = 0; ci < nb; ++ci) {
                    CRF_RES_CHUNK_ACC(acc, g01, g23, A, c0 + ci, ci);
                    // (a row ends after 2 - 3 of a wave's 15 chunks: the test is laid out so that the common path falls through --
                    // a taken branch costs a wave ~32 cycles of instruction refetch, tools/ubench_issue.py.  With a probability, not
                    // "never": blocks the allocator believes cold are where it spills, and a reload from scratch waits for
                    // vmcnt(0), i.e. for the acknowledgement of the frame's write-through row stores)
                    if (__builtin_expect_with_probability((ends_f >> (c0 + ci) & 1u) != 0u, 0, 0.8)) row_end((unsigned)__builtin_popcount(ends_f & ((1u << (c0 + ci)) - 1u)));
                }
            }
        }
        CRF_TM(tm_on, tm_i + 2);
#ifdef CRF_TIMING
        if (b == 3 && i >= 150 && i < 158) {
            const int o = 12288 + ((DIR * 4) * 8 + wave) * 16;
            CRF_TM(true, o + (i - 150));
            if (lane == 0 && i == 150) { g_tm[o + 8] = (unsigned long long)nch; g_tm[o + 9] = (unsigned long long)__builtin_popcount(ends); }
        }
#endif
        if constexpr (K2) {   // the peer's entries (published from its row epilogues) into this CU's vector
            const int p0 = L.cu_row[1 - k], p1 = L.cu_row[2 - k];
            float *Xn = (float *)xnb;
            float fm;
            if (DIR == 0) {
                // the U entries, and of the L / A entries those this CU's rows gather (a list) -- all of them after the last frame
                // (logZ; only the CU that computes it)
                if (i == lx - 1) {
                    fm = res_fetch<NTH>(slot, Xn, p0, p1, tag, p.err, tid);
                    if (lead) {
                        fm = fmaxf(fm, res_fetch<NTH>(slot, Xn, R + p0, R + p1, tag, p.err, tid));
                        fm = fmaxf(fm, res_fetch<NTH>(slot, Xn, 2 * R + p0, 2 * R + p1, tag, p.err, tid));
                    }
                } else {
                    fm = res_fetch_both<NTH>(slot, Xn, p0, p1, p.xlist, p.xlist_off[k], p.xlist_off[k + 1], tag, p.err, tid);   // (one polling loop for both)
                }
            } else {
                fm = res_fetch<NTH>(slot, Xn, 2 * p0, 2 * p1, tag, p.err, tid);
            }
            mymax = fmaxf(mymax, fm);
        }
        // lagged scale: the maximum of THIS frame's source vector (deposited during the frame before, complete since the last barrier) gives
        // the scale of the NEXT frame -- requested here, where the gathers' registers are free, ahead of the wave maximum's DPP chain; the
        // scalar arithmetic runs while the deposit below is on its way
        [[maybe_unused]] i32x4_t m4n{};
        if constexpr (LAG) m4n = *(const i32x4_t *)(wm + sr * 4);
        mymax = row_max16(mymax);   // (sending the maximum from every row end instead -- no reduction in the tail -- was measured 3 % slower, round 4)
        if (rowlead) lds_fmax(wm + sw * 4 + (lane >> 4), mymax);
        sr = sw;
        if (pre) {
            float *EPw = EP + (DIR == 0 ? 1 - par : par) * Vp;
#pragma unroll
            for (int q = 0; q < EPR; ++q) { const int v = tid + q * NTH; if (v < V) EPw[v] = epn[q]; }
        }
        if constexpr (LAG) lag_mx = max(max(m4n.x, m4n.y), max(m4n.z, m4n.w));   // (behind the emission staging: ONE wait for the LDS in the tail, the one the barrier needs anyway)
        CRF_TM(tm_on, tm_i + 3);
#ifdef CRF_TIMING
        if (b == 3 && i >= 150 && i < 158) CRF_TM(true, 15360 + (DIR * 16 + wave) * 8 + (i - 150));   // this wave's arrival at the frame barrier
#endif
        sync_lds();
        CRF_TM(tm_on, tm_i + 4);
    };
    // Entries no row produces (states nobody enters: the start state) still hold a_0 in the buffer frame 0 read from, and
    // nothing rewrites them: they are cleared before that buffer becomes the source again (frame 2).  In an ordinary frame
    // the stale start mass is ~2^-60 of the vector; once the vector underflows it would be ALL of it -- a finite, wrong
    // logZ instead of the zero that sends the utterance to the robust kernels.  The clearing sits BETWEEN two runs of the
    // one frame loop (frame 0 alone, then the rest): inside the loop body it cost 2.6 % of the step (measured).
    int i = i0;
#pragma clang loop unroll(disable)
    for (int seg = 0; seg < 2; ++seg) {
        const int iend = (DIR == 0 && seg == 0) ? min(i1, 1) : i1;
#pragma clang loop unroll(disable)
        for (; i < iend; ++i) frame(i & 1, i);
        if (DIR == 0 && seg == 0 && i0 == 0 && i1 > 0) {
            for (int s = tid; s < G; s += NTH) if (p.x_start[s] != 0.f) X[s] = 0.f;
            sync_lds();
        }
    }
    if (i1 < lx) {                                           // not the last segment of this utterance: park the state
        const float *Xc = X + (i1 & 1) * Gp;
        for (int s = tid; s < G; s += NTH) state[s] = Xc[s];
        if (tid == 0) state[Gp] = __int_as_float(E);
        if (tid == 0 && LAG && lag_lo < kLagLow) p.redo[DIR * p.B + b] = 1;  // (the flag does not travel with the parked state)
        return;
    }
    if (flagged)
        while (next_stage < p.nb) publish_stage();           // the rest (at least the last stage: bound = T)
    if (DIR == 0) {
        if (!lead) return;                                   // (the vector is complete on both CUs)
        const float *Xf = X + (lx & 1) * Gp;
        float part = 0.f;
        for (int s = tid; s < G; s += NTH) part += Xf[s] * p.x_end[s];
        const float zs = res_block_sum<NW>(part, (float *)red, tid);
        const double mxs = res_mx_total<NW>(p, b, lx, red, tid);
        if (tid == 0) { p.den_zs[b] = zs; p.den_ez[b] = E; p.cost_alpha[b] = to_log(zs, E, mxs); if (!(zs > 0.f && zs < INFINITY) || (LAG && lag_lo < kLagLow)) p.redo[b] = 1; }
    } else {
        if (lx > 0) {
            __syncthreads();  // drains vmcnt: this workgroup's own stores to the spare row are visible to it
            const float *r0 = p.Row0 + (int64_t)b * p.Rout;
            const int ra = K2 ? 2 * L.cu_row[k] : 0, rb = K2 ? 2 * L.cu_row[k + 1] : 2 * R;   // the rows this CU wrote
            for (int r = ra + tid; r < rb; r += NTH) zpart += p.brow_start[r] * r0[r];
            // states without a row: b_0 = (sum over their arcs of w * z_0) * (scale of the last frame); z_0 is the vector the
            // last frame read
            const float *Xl = X + ((lx - 1) & 1) * Gp;
            if (lead) for (int a = tid; a < p.nbx; a += NTH) zpart += p.bx_w[a] * Xl[p.bx_idx[a]] * last_sc;
        } else if (tid == 0 && lead) zpart += p.bx_se * pow2f(kScaleExp);
        float zb = res_block_sum<NW>(zpart, (float *)red, tid);
        if constexpr (K2) {   // the peer's part of the sum: one more granule through the handshake words (tag 2)
            if (!lead) {
                if (tid == 0) __hip_atomic_store(hs + k, (2ull << 32) | __float_as_uint(zb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            if (tid == 0) {
                unsigned long long g = 0;
                for (unsigned spins = 0; (g >> 32) != 2ull; ++spins) {
                    g = __hip_atomic_load(hs + (1 - k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (spins > (1u << 22)) { __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    if ((spins & 255u) == 255u && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                zb += (g >> 32) == 2ull ? __uint_as_float((unsigned)g) : 0.f;
            }
        }
        const double mxs = res_mx_total<NW>(p, b, lx, red, tid);
        if (tid == 0) { p.cb_part[(size_t)b * kResMaxK] = zb; p.cb_F[b] = E; p.cb_mxs[b] = mxs; if (!(zb > 0.f && zb < INFINITY) || (LAG && lag_lo < kLagLow)) p.redo[p.B + b] = 1; }
    }
}

// =============================================================================================
// TWO UTTERANCES per workgroup (throughput mode: batches above CUs / 4 utterances per GPU).  Same layout tables, same arithmetic,
// same rows in memory as fac_chain_body -- bit for bit -- but the state vector is [entry][2 utterances] (float2): ONE address
// computation and ONE ds_read_b64 (the LDS cycles of a ds_read_b32) gather an entry for both, the weight is shared in its register,
// the product is one v_pk_fma_f32 whose two lanes are the two utterances.  Per arc and utterance that is half an address
// instruction, half an LDS instruction and half a packed FMA, against 1 + 1 + 1/2 with one utterance per workgroup; the price is
// two row epilogues per slice and twice the frame's bookkeeping (two scales, two exponents, two row pointers).  A pair runs
// max(lx0, lx1) frames: the shorter utterance's sums are taken when it ends (its half of the vector then holds garbage nobody
// reads, its row stores go to a dump row).  768-thread geometries only (row constants in registers or in the LDS table).
// LDS: X2[2][Gp] float2 | row table (RL: as fac_chain_body; RC backward: the two extra-arc weights per row) | EP2[2][Vp] float2 |
//      wm[3][2][4] | red
// =============================================================================================
__device__ __forceinline__ f32x2 lds_ld2(unsigned a) { return *(const __attribute__((address_space(3))) f32x2 *)(uintptr_t)a; }
__device__ __forceinline__ float lds_ld1(unsigned a) { return *(const __attribute__((address_space(3))) float *)(uintptr_t)a; }
__device__ __forceinline__ void lds_st2(unsigned a, f32x2 v) { *(__attribute__((address_space(3))) f32x2 *)(uintptr_t)a = v; }
// address of entry (16-bit byte offset of the one-utterance layout, low / high half of `w`) in a float2 vector at LDS address `base`
__device__ __forceinline__ unsigned addr2_lo(unsigned w, unsigned base) { unsigned r; asm("v_mad_u32_u16 %0, %1, 2, %2" : "=v"(r) : "v"(w), "s"(base)); return r; }
__device__ __forceinline__ unsigned addr2_hi(unsigned w, unsigned base) { unsigned r; asm("v_mad_u32_u16 %0, %1, 2, %2 op_sel:[1,0,0,0]" : "=v"(r) : "v"(w), "s"(base)); return r; }

template <int DIR, bool FLAG, int NTH, int NCH, int NB, bool ML, bool RL>
__device__ __forceinline__ void fac_chain_body2(const FacParams &p, float *lds, const int pair) {
    // NTH: 768 (the one-utterance kernel's 768-thread layouts as they are; 168 registers per wave -- this kernel spills there) or
    // 512 x 30 chunks (round 5: a layout of its own, HostGraph::facp; two waves per SIMD with 256 registers each)
    constexpr int NW = NTH / kWave;
    constexpr bool RC = !RL;
    static_assert(NTH == kFac3Threads || (NTH == kResThreads && RL), "two utterances per workgroup: 768 threads, or 512 threads with the row table");
    constexpr int NCHA = RC ? kFac3ArcCh : NCH;
    constexpr int RCW = NCHA * 6;
    static_assert(!RC || NCH == kFac3NCH, "row constants in registers: 20 chunks of arcs + the constants' slot");
    const FacDirDev &L = p.L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int V = p.V, G = L.G, R = L.R;
    const int bu[2] = {2 * pair, 2 * pair + 1};
    const bool real1 = bu[1] < p.B;                          // (an odd batch: the last pair's second utterance does not exist)
    const int lxu[2] = {p.lx[bu[0]], real1 ? p.lx[bu[1]] : 0};
    const unsigned dup2 = 2u * (unsigned)L.dup;
    const int Vp = rup64(V + 1), Gp = rup64(G);
    const unsigned XB2 = (unsigned)Gp * 8u;                  // bytes per state vector
    constexpr int kRowB = RL ? (DIR == 0 ? 8 : 16) : 8;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_char *)lds;   // LDS address of the carve
    f32x2 *X2 = (f32x2 *)lds;                                // [2][Gp]
    char *RMc = (char *)(X2 + 2 * Gp);
    const size_t tbytes = RL ? (size_t)(R + 64) * kRowB : (DIR == 1 ? (size_t)R * 8 : (size_t)0);
    f32x2 *EP2 = (f32x2 *)(RMc + tbytes);                    // [2][Vp]
    float *wm = (float *)(EP2 + 2 * Vp);                     // [3][2][4]
    double *red = (double *)(wm + 24);
    const unsigned rm0 = lds0 + 2u * XB2, ep0 = rm0 + (unsigned)tbytes, EB2 = (unsigned)Vp * 8u;
    if (tid == 0 && p.started) atomicAdd(p.started, 1);

    unsigned A[(NCH * 6)];
    unsigned rc00 = 0, rc01 = 0, rc10 = 0, rc11 = 0, rc20 = 0, rc21 = 0;
    {
        const unsigned *src = L.arcs + tid;
#pragma unroll
        for (int i = 0; i < (RC ? RCW : NCH * 6); ++i) A[i] = src[(size_t)i * NTH];
        if (RC) {
            rc00 = src[(size_t)(RCW + 0) * NTH]; rc01 = src[(size_t)(RCW + 1) * NTH];
            rc10 = src[(size_t)(RCW + 2) * NTH]; rc11 = src[(size_t)(RCW + 3) * NTH];
            rc20 = src[(size_t)(RCW + 4) * NTH]; rc21 = src[(size_t)(RCW + 5) * NTH];
        }
    }
    const uint4 wi = L.wave_info[wave];
    const unsigned ends = __builtin_amdgcn_readfirstlane(wi.x);
    const int nch = __builtin_amdgcn_readfirstlane(wi.y);
    const int row0 = __builtin_amdgcn_readfirstlane(wi.z);
    const unsigned lgbits = __builtin_amdgcn_readfirstlane(wi.w);
    if (RL) {
        if (DIR == 0) {
            uint2 *RT = (uint2 *)RMc;
            for (int r = tid; r < R; r += NTH) {
                const int4 m = p.frow_meta[r];
                RT[r] = uint2{(((unsigned)m.x >> 16) * 4u) | (((unsigned)m.w * 4u) << 16), (unsigned)m.z};
            }
        } else {
            uint4 *RT = (uint4 *)RMc;
            for (int r = tid; r < R; r += NTH) {
                const int4 m = p.brow_meta[r];
                const unsigned l0 = (unsigned)m.w & 0xffffu, l1 = (unsigned)m.w >> 16;
                RT[r] = uint4{(unsigned)m.x, ((l0 == 0xffffu ? (unsigned)V : l0) * 4u) | (((l1 == 0xffffu ? (unsigned)V : l1) * 4u) << 16), (unsigned)m.y, (unsigned)m.z};
            }
        }
    } else if (DIR == 1) {
        f32x2 *RW = (f32x2 *)RMc;
        for (int r = tid; r < R; r += NTH) { const int4 m = p.brow_meta[r]; RW[r] = f32x2{__int_as_float(m.y), __int_as_float(m.z)}; }
        auto fix = [&](unsigned w) {
            const unsigned l0 = w & 0xffffu, l1 = w >> 16;
            return (l0 == 0xffffu ? (unsigned)V * 4u : l0) | ((l1 == 0xffffu ? (unsigned)V * 4u : l1) << 16);
        };
        rc01 = fix(rc01); rc11 = fix(rc11); rc21 = fix(rc21);
    }
    const int64_t bt0u[2] = {(int64_t)bu[0] * p.T, (int64_t)bu[1] * p.T};
    const float *ep_b[2] = {p.ep + bt0u[0] * V, p.ep + (real1 ? bt0u[1] : bt0u[0]) * V};
    float *Out_b[2] = {p.Out + bt0u[0] * p.Rout, p.Out + (real1 ? bt0u[1] : bt0u[0]) * p.Rout};
    int *Eo_b[2] = {p.Eout + bt0u[0], p.Eout + (real1 ? bt0u[1] : bt0u[0])};
    float *dump = p.dump + ((size_t)DIR * p.npair + pair) * (size_t)p.dump_stride;   // rows of an utterance that has ended
    int E[2] = {kScaleExp, kScaleExp};
    float zpart[2] = {0.f, 0.f};
    for (int s = tid; s < 2 * Gp; s += NTH) X2[s] = f32x2{0.f, 0.f};
    if (tid < 2) EP2[tid * Vp + V] = f32x2{0.f, 0.f};
    if (tid < 24) wm[tid] = 0.f;                             // [3 frames][2 utterances][4 rows of 16 lanes]
    const bool rowlead = (lane & 15) == 0;
    int sr = 0;
    for (int v = tid; v < V; v += NTH) {
        f32x2 e0{0.f, 0.f}, e1{0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int lx = lxu[u];
            if (lx <= 0) continue;
            if (DIR == 0) e0[u] = ep_b[u][v];                                             // e'_0
            else { e0[u] = ep_b[u][(unsigned)(lx - 1) * (unsigned)V + v]; if (lx >= 2) e1[u] = ep_b[u][(unsigned)(lx - 2) * (unsigned)V + v]; }
        }
        EP2[v] = e0;
        if (DIR == 1) EP2[Vp + v] = e1;
    }
    __syncthreads();
    {
        f32x2 m0{0.f, 0.f};
        if (DIR == 0) {
            for (int s = tid; s < G; s += NTH) { const float v = p.x_start[s] * pow2f(kScaleExp); X2[s] = f32x2{v, v}; m0.x = fmaxf(m0.x, v); }
            m0.y = m0.x;
        } else {
            for (int z = tid; z < G; z += NTH) {
                const int l = p.z_lab[z];
                const f32x2 e = EP2[l < 0 ? V : l];
                const float ze = p.z_end[z] * pow2f(kScaleExp);
                const f32x2 v{lxu[0] > 0 ? e.x * ze : 0.f, lxu[1] > 0 ? e.y * ze : 0.f};
                X2[z] = v; m0.x = fmaxf(m0.x, v.x); m0.y = fmaxf(m0.y, v.y);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !real1) continue;
                if (lxu[u] > 0) {
                    float *BProw = Out_b[u] + (unsigned)(lxu[u] - 1) * (unsigned)p.Rout;
                    for (int r = tid; r < 2 * R; r += NTH) BProw[r] = p.brow_end[r] * pow2f(kScaleExp);
                    if (tid == 0) Eo_b[u][lxu[u] - 1] = E[u];
                } else {
                    for (int r = tid; r < 2 * R; r += NTH) zpart[u] += p.brow_start[r] * p.brow_end[r] * pow2f(kScaleExp);
                }
            }
        }
        m0.x = row_max16(m0.x); m0.y = row_max16(m0.y);
        if (rowlead) { lds_fmax(wm + (lane >> 4), m0.x); lds_fmax(wm + 4 + (lane >> 4), m0.y); }
    }
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0x0F70);

    int next_stage = 1;
    int next_bound = (FLAG && p.nb > 1) ? p.bound[1] : 0x7fffffff;
    const int nreal = real1 ? 2 : 1;
    auto publish_stage = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(p.stage_cnt + next_stage, nreal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ++next_stage;
        next_bound = next_stage < p.nb ? p.bound[next_stage] : 0x7fffffff;
    };
    const bool pre_w = wave * kWave < V;
    f32x2 last_sc{1.f, 1.f};
    f32x2 epn[kEpRegsR] = {};
    // one frame; `act`: bit u = utterance u has not ended (its rows, exponents and emissions are real)
    auto frame = [&](const int par, const int i, const int act) __attribute__((always_inline)) {
        if (FLAG && i == next_bound) publish_stage();
        const unsigned xb = lds0 + (unsigned)par * XB2, xnb = lds0 + (unsigned)(1 - par) * XB2;
        const unsigned epu = ep0 + (unsigned)(DIR == 0 ? par : 1 - par) * EB2;
        int tu[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) tu[u] = DIR == 0 ? i : lxu[u] - 1 - i;
        bool pre[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            pre[u] = pre_w && (act >> u & 1) && (DIR == 0 ? (tu[u] + 1 < lxu[u]) : (tu[u] >= 2));
            if (pre[u]) {
                const float *er = ep_b[u] + (unsigned)(DIR == 0 ? tu[u] + 1 : tu[u] - 2) * (unsigned)V;
#pragma unroll
                for (int q = 0; q < kEpRegsR; ++q) { const int v = tid + q * NTH; if (v < V) epn[q][u] = er[v]; }
            }
        }
        const int sw = sr == 2 ? 0 : sr + 1, sz = sw == 2 ? 0 : sw + 1;
        const int4 ma = *(const int4 *)(wm + sr * 8), mb = *(const int4 *)(wm + sr * 8 + 4);
        const int ksc0 = rescale_exp_bits((unsigned)__builtin_amdgcn_readfirstlane(max(max(ma.x, ma.y), max(ma.z, ma.w))));
        const int ksc1 = rescale_exp_bits((unsigned)__builtin_amdgcn_readfirstlane(max(max(mb.x, mb.y), max(mb.z, mb.w))));
        if (wave == 0 && lane < 8) wm[sz * 8 + lane] = 0.f;
        const f32x2 sc{pow2f(ksc0), pow2f(ksc1)};
        if (DIR == 1) last_sc = sc;
        float *Orow[2];
        const int ksc[2] = {ksc0, ksc1};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool on = act >> u & 1;
            if (DIR == 0) {
                E[u] += ksc[u];
                if (tid == 0 && on) Eo_b[u][tu[u]] = E[u];
                E[u] += kEpExp;
                Orow[u] = on ? Out_b[u] + (unsigned)tu[u] * (unsigned)p.Rout : dump;
            } else {
                E[u] += ksc[u] + kEpExp;
                if (tid == 0 && on && tu[u] > 0) Eo_b[u][tu[u] - 1] = E[u];
                Orow[u] = !on ? dump : tu[u] > 0 ? Out_b[u] + (unsigned)(tu[u] - 1) * (unsigned)p.Rout : p.Row0 + (int64_t)bu[u] * p.Rout;
            }
        }
        unsigned ends_f = ends;
        int nch_f = nch;
        asm volatile("" : "+s"(ends_f), "+s"(nch_f));
        f32x2 acc{0.f, 0.f}, accb{0.f, 0.f};
        f32x2 mymax{0.f, 0.f};
        unsigned r4 = (unsigned)(row0 + lane) * 4u;   // 4 * row id
        typedef std::conditional_t<DIR == 0, uint2, uint4> rct_t;
        [[maybe_unused]] rct_t kc{};
        if constexpr (RL) kc = *(const rct_t *)(RMc + (DIR == 0 ? 2u : 4u) * r4);
        auto row_end = [&](const unsigned ks) __attribute__((always_inline)) {
            f32x2 tot = acc + accb;
            if constexpr (ML) {
                const unsigned lg = ks < 10u ? (lgbits >> (3u * ks)) & 7u : 0u;
                if (lg) {
                    // (two scalars, not the halves of the float2: with the DPP source a sub-register of a 64-bit tuple the compiler's
                    // DPP combiner took the OTHER half as source -- v_add_f32_dpp v7, v6, v7 -- found by the parity tests)
                    float tx = tot.x, ty = tot.y;
                    asm volatile("" : "+v"(tx), "+v"(ty));
#define CRF_DPP_ADD2(ctrl) { tx += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tx), ctrl, 0xf, 0xf, false)); \
                             ty += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ty), ctrl, 0xf, 0xf, false)); }
                    CRF_DPP_ADD2(0xB1);
                    if (lg >= 2) CRF_DPP_ADD2(0x4E);
                    if (lg >= 3) CRF_DPP_ADD2(0x141);
                    if (lg >= 4) CRF_DPP_ADD2(0x140);
#undef CRF_DPP_ADD2
                    if (lg >= 5) { tx += __shfl_xor(tx, 16, 64); ty += __shfl_xor(ty, 16, 64); }
                    if (lg >= 6) { tx += __shfl_xor(tx, 32, 64); ty += __shfl_xor(ty, 32, 64); }
                    tot = f32x2{tx, ty};
                }
            }
            unsigned k0, k1;
            [[maybe_unused]] f32x2 wrl{};
            if constexpr (RC) {
                const unsigned s0 = 0u - (unsigned)(ks == 0), s1 = 0u - (unsigned)(ks == 1), s2 = 0u - (unsigned)(ks >= 2);
                k0 = (rc00 & s0) | (rc10 & s1) | (rc20 & s2);
                k1 = (rc01 & s0) | (rc11 & s1) | (rc21 & s2);
            } else {
                k0 = kc.x; k1 = kc.y;
                if constexpr (DIR == 1) wrl = f32x2{__uint_as_float(kc.z), __uint_as_float(kc.w)};
                kc = *(const rct_t *)(RMc + (DIR == 0 ? 2u : 4u) * (r4 + kWave * 4u));
            }
            if (DIR == 0) {   // k0 = main label | tail label << 16 (byte offsets of a float row), k1 = tail weight; U, L, A at rid, R + rid, 2R + rid
                const f32x2 uold = lds_ld2(xb + 2u * r4);
                const f32x2 em = lds_ld2(epu + 2u * (k0 & 0xffffu)), et = lds_ld2(epu + 2u * (k0 >> 16));
                const f32x2 rv = tot * sc;
                const f32x2 qt = __uint_as_float(k1) * uold * sc;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (FLAG) {
                        __hip_atomic_store((unsigned *)((char *)Orow[u] + r4), __float_as_uint(rv[u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store((unsigned *)((char *)Orow[u] + r4 + 4u * (unsigned)R), __float_as_uint(qt[u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        *(float *)((char *)Orow[u] + r4) = rv[u];
                        *(float *)((char *)Orow[u] + r4 + 4u * (unsigned)R) = qt[u];
                    }
                }
                const f32x2 Lp = em * rv, Ap = et * qt, Up = Ap + Lp;
                lds_st2(xnb + 2u * r4, Up);
                lds_st2(xnb + 2u * r4 + dup2, Up);
                lds_st2(xnb + 2u * r4 + 8u * (unsigned)R, Lp);
                lds_st2(xnb + 2u * r4 + 16u * (unsigned)R, Ap);
                mymax.x = __int_as_float(max(__float_as_int(mymax.x), __float_as_int(Up.x)));
                mymax.y = __int_as_float(max(__float_as_int(mymax.y), __float_as_int(Up.y)));
            } else {          // k0 = z offsets of the two extra arcs, k1 = label 0 | label 1 << 16 (byte offsets of a float row)
                const f32x2 z0 = lds_ld2(xb + 2u * (k0 & 0xffffu)), z1 = lds_ld2(xb + 2u * (k0 >> 16));
                const f32x2 e0 = lds_ld2(epu + 2u * (k1 & 0xffffu)), e1 = lds_ld2(epu + 2u * (k1 >> 16));
                const f32x2 w01 = RL ? wrl : *(const f32x2 *)(RMc + 2u * r4);
                f32x2 bx, by;                                     // b_t of the row's two states, per utterance
                bx = __builtin_elementwise_fma(f32x2{w01.x, w01.x}, z0, tot) * sc;
                by = __builtin_elementwise_fma(f32x2{w01.y, w01.y}, z1, tot) * sc;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (FLAG)
                        __hip_atomic_store((unsigned long long *)((char *)Orow[u] + 2u * r4),
                                           (unsigned long long)__float_as_uint(bx[u]) | ((unsigned long long)__float_as_uint(by[u]) << 32),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        *(f32x2 *)((char *)Orow[u] + 2u * r4) = f32x2{bx[u], by[u]};
                }
                const f32x2 zx = e0 * bx, zy = e1 * by;           // z_{t-1} of the pairs entering them
                lds_st2(xnb + 4u * r4, zx);
                lds_st2(xnb + 4u * r4 + 8u, zy);
                lds_st2(xnb + 4u * r4 + dup2, zx);
                lds_st2(xnb + 4u * r4 + dup2 + 8u, zy);
                mymax.x = __int_as_float(max(__float_as_int(mymax.x), max(__float_as_int(zx.x), __float_as_int(zy.x))));
                mymax.y = __int_as_float(max(__float_as_int(mymax.y), max(__float_as_int(zx.y), __float_as_int(zy.y))));
            }
            acc = f32x2{0.f, 0.f}; accb = f32x2{0.f, 0.f};
            r4 += kWave * 4u;
        };
#pragma unroll
        for (int c0 = 0; c0 < NCHA; c0 += NB) {
            const int nb = NCHA - c0 < NB ? NCHA - c0 : NB;
            if (c0 < nch_f) {
                f32x2 g[NB][4];
#pragma unroll
                for (int ci = 0; ci < nb; ++ci) {
                    const unsigned i01 = A[6 * (c0 + ci)], i23 = A[6 * (c0 + ci) + 1];
                    g[ci][0] = lds_ld2(addr2_lo(i01, xb)); g[ci][1] = lds_ld2(addr2_hi(i01, xb));
                    g[ci][2] = lds_ld2(addr2_lo(i23, xb)); g[ci][3] = lds_ld2(addr2_hi(i23, xb));
                }
#pragma unroll
                for (int ci = 0; ci < nb; ++ci) {
                    const int c = c0 + ci;
                    // the weights stay the register PAIRS the one-utterance kernel multiplies with; a packed FMA takes one half of a
                    // pair for both of its lanes (op_sel) -- a splat built in C makes the compiler keep a second register per weight
                    // alive across the whole kernel (400 spilled registers).  The one-utterance kernel sums (a0 w0 + a2 w2) and
                    // (a1 w1 + a3 w3) in the two lanes of its packed FMA and adds the two at the row's end: the same two chains here,
                    // per utterance -- the results are bit-identical
                    f32x2 w01, w23;
                    w01.x = __uint_as_float(A[6 * c + 2]); w01.y = __uint_as_float(A[6 * c + 3]);
                    w23.x = __uint_as_float(A[6 * c + 4]); w23.y = __uint_as_float(A[6 * c + 5]);
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(g[ci][0]), "v"(w01));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(accb) : "v"(g[ci][1]), "v"(w01));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(g[ci][2]), "v"(w23));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(accb) : "v"(g[ci][3]), "v"(w23));
                    if (ends_f >> c & 1u) row_end((unsigned)__builtin_popcount(ends_f & ((1u << c) - 1u)));
                }
            }
        }
        mymax.x = row_max16(mymax.x); mymax.y = row_max16(mymax.y);
        if (rowlead) { lds_fmax(wm + sw * 8 + (lane >> 4), mymax.x); lds_fmax(wm + sw * 8 + 4 + (lane >> 4), mymax.y); }
        sr = sw;
        if (pre[0] | pre[1]) {
            f32x2 *EPw = EP2 + (DIR == 0 ? 1 - par : par) * Vp;
#pragma unroll
            for (int q = 0; q < kEpRegsR; ++q) {
                const int v = tid + q * NTH;
                if (v < V) {   // (an utterance that is not prefetched keeps its old emissions: never read again, or read by garbage only)
                    if (pre[0] && pre[1]) EPw[v] = epn[q];
                    else if (pre[0]) ((float *)(EPw + v))[0] = epn[q].x;
                    else ((float *)(EPw + v))[1] = epn[q].y;
                }
            }
        }
        sync_lds();
    };
    // sums of an utterance that has just ended (forward: its logZ; backward: its part of the backward logZ)
    auto finish = [&](const int u, const int nfr) __attribute__((always_inline)) {
        if (u == 1 && !real1) return;
        const int b = bu[u], lx = lxu[u];
        if (DIR == 0) {
            const f32x2 *Xf = X2 + (nfr & 1) * Gp;
            float part = 0.f;
            for (int s = tid; s < G; s += NTH) part += Xf[s][u] * p.x_end[s];
            const float zs = res_block_sum<NW>(part, (float *)red, tid);
            const double mxs = res_mx_total<NW>(p, b, lx, red, tid);
            if (tid == 0) { p.den_zs[b] = zs; p.den_ez[b] = E[u]; p.cost_alpha[b] = to_log(zs, E[u], mxs); if (!(zs > 0.f && zs < INFINITY)) p.redo[b] = 1; }
        } else {
            float zp = zpart[u];
            if (lx > 0) {
                __syncthreads();  // drains vmcnt: this workgroup's own stores to the spare row are visible to it
                const float *r0 = p.Row0 + (int64_t)b * p.Rout;
                for (int r = tid; r < 2 * R; r += NTH) zp += p.brow_start[r] * r0[r];
                const f32x2 *Xl = X2 + ((lx - 1) & 1) * Gp;     // the vector the utterance's last frame read
                for (int a = tid; a < p.nbx; a += NTH) zp += p.bx_w[a] * Xl[p.bx_idx[a]][u] * last_sc[u];
            } else if (tid == 0) zp += p.bx_se * pow2f(kScaleExp);
            const float zb = res_block_sum<NW>(zp, (float *)red, tid);
            const double mxs = res_mx_total<NW>(p, b, lx, red, tid);
            if (tid == 0) { p.cb_part[(size_t)b * kResMaxK] = zb; p.cb_F[b] = E[u]; p.cb_mxs[b] = mxs; if (!(zb > 0.f && zb < INFINITY)) p.redo[p.B + b] = 1; }
        }
    };
    const int us = lxu[1] < lxu[0] ? 1 : 0, ul = 1 - us;      // the utterance that ends first / last
    const int lmin = lxu[us], lmax = lxu[ul];
    if (lmin == 0) finish(us, 0);                             // an empty utterance: its sums come from the untouched start vector
    // ONE call site of the frame (the loop body is ~10 KB of code; the 64 KiB instruction cache is shared by two CUs): three runs of
    // the same loop -- frame 0 of the forward recursion alone (the entries no row produces, the start state, are cleared before their
    // buffer becomes the source again: see fac_chain_body), the frames with both utterances, the frames of the longer one alone
    int i = 0;
#pragma clang loop unroll(disable)
    for (int seg = 0; seg < 3; ++seg) {
        const int iend = seg == 0 ? (DIR == 0 ? min(lmax, 1) : 0) : seg == 1 ? lmin : lmax;
#pragma clang loop unroll(disable)
        for (; i < iend; ++i) frame(i & 1, i, i < lmin ? 3 : (1 << ul));
        if (seg == 0 && DIR == 0 && lmax > 0) {
            for (int s = tid; s < G; s += NTH) if (p.x_start[s] != 0.f) X2[s] = f32x2{0.f, 0.f};
            sync_lds();
        }
        if (seg == 1 && lmin > 0 && lmin < lmax) finish(us, lmin);
    }
    if (FLAG)
        while (next_stage < p.nb) publish_stage();
    if (lmin > 0 && lmin == lmax) finish(us, lmin);
    finish(ul, lmax);
}

// Both recursions of every utterance as ONE grid of 2B workgroups (block x < B: forward recursion of utterance x,
// else the backward recursion of utterance x - B).  One launch on one stream: the two directions used to be two
// kernels on two streams, which ran side by side only while those streams sat on different hardware queues -- not
// guaranteed inside a training process (HIP maps all streams of a process onto GPU_MAX_HW_QUEUES = 4 queues; with
// RCCL's and torch's streams around, the recursions were observed to run one after the other: 5.3 instead of 3.25 ms).
// NBF / NBB: chunks gathered per batch, forward / backward.
template <bool FLAG, int NTH, int NCH, int NBF, int NBB, bool ML, bool RL>
__global__ __launch_bounds__(NTH) void crf_fac_pair_kernel(FacParams pf, FacParams pb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int B = pf.B;
    if ((int)blockIdx.x < B) fac_chain_body<0, FLAG, NTH, NCH, NBF, ML, RL>(pf, lds, (int)blockIdx.x);
    else fac_chain_body<1, FLAG, NTH, NCH, NBB, ML, RL>(pb, lds, (int)blockIdx.x - B);
}
// ... with TWO UTTERANCES per workgroup: 2 * ceil(B / 2) workgroups, forward recursions first
template <bool FLAG, int NTH, int NCH, int NBF, int NBB, bool ML, bool RL>
__global__ __launch_bounds__(NTH) void crf_fac_pair2_kernel(FacParams pf, FacParams pb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int np = pf.npair;
    if ((int)blockIdx.x < np) fac_chain_body2<0, FLAG, NTH, NCH, NBF, ML, RL>(pf, lds, (int)blockIdx.x);
    else fac_chain_body2<1, FLAG, NTH, NCH, NBB, ML, RL>(pb, lds, (int)blockIdx.x - np);
}
// ... with TWO CUs per recursion: 2 * nbu * 2 workgroups for the utterances [b0, b0 + nbu), forward recursions first; the
// two CUs of a recursion 8 block ids apart (block x is observed on XCD x % 8: one L2 for the hand-off; a matter of speed only)
template <int NTH, int NCH, int NBF, int NBB>
__global__ __launch_bounds__(NTH) void crf_fac2_pair_kernel(FacParams pf, FacParams pb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int K = 2;
    const int total = pf.nbu * K;
    const bool fwd = (int)blockIdx.x < total;
    const int x = fwd ? (int)blockIdx.x : (int)blockIdx.x - total;
    const int full = total / (8 * K) * (8 * K);
    int b, k;
    if (x < full) { const int grp = x / (8 * K), within = x % (8 * K); k = within / 8; b = pf.b0 + grp * 8 + within % 8; }
    else { const int y = x - full; k = y % K; b = pf.b0 + full / K + y / K; }
    if (fwd) fac_chain_body<0, false, NTH, NCH, NBF, true, true, true>(pf, lds, b, k);
    else fac_chain_body<1, false, NTH, NCH, NBB, true, true, true>(pb, lds, b, k);
}

}  // namespace crf
