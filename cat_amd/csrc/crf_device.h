// cat_amd/csrc/crf_device.h -- what every kernel family of the CTC-CRF loss shares: the kernel argument block (LossParams), the build-time
// A/B switches, wave / block reductions, the exact power-of-two rescale helpers, LDS-only barriers, the in-kernel timing stamps.
// Included by every translation unit under cat_amd/csrc/k_*.hip and by crf_host.hip (the host side and the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/ctc_crf_hip.h"
#include "crf_internal.h"

namespace crf {

constexpr int kEpRegs = 8;    // ep row prefetch registers per thread  -> V  <= 8 * 1024
constexpr int kCtcThreads = 512, kCtcWaves = kCtcThreads / 64;  // numerator chains: 8 waves (S' = 2L+1 is a few hundred)
constexpr int kCtcRegs = 8;   // ctc states per thread                  -> 2L+1 <= 8 * 512
constexpr int kCtcPF = 4;     // frames per emission prefetch batch (ctc_forward)
constexpr int kGradThreads = 256;
constexpr int kGradFrames = 4;  // frames per crf_grad_kernel workgroup
// The streaming grad kernels normalise every frame by its own sum, so the power of two they take out of e' * sum(q * b) is free -- and must
// leave the product in range at BOTH ends: the label sums sit at ~2^40 (two rows rescaled to 2^20; up to 2^55 for a label with thousands of
// pairs), e' = exp(logp - rowmax) * 2^64 reaches down to 2^-125.  Rounds 1 - 4 took out 2^-64 ("undo the emissions' factor"): a frame whose
// ALLOWED labels all lie 100 - 131 nats below the row maximum -- the recursions carry it, nothing is flagged -- then had e' * 2^-64 * 2^40
// below the smallest normal float and came out with zero or garbage posteriors, silently (found in round 5 by
// tests/test_gpu_parity.py::test_single_frame_shrink_window[110]: gradient 90 % off, loss right).  2^-4: the product stays normal wherever
// e' itself is, and 2^64 * 2^55 * 2^-4 < 2^127 at the other end.
constexpr int kGradDescale = 4;
constexpr int kFlagFallback = 40;   // words [40], [41] behind the error word: utterances of the last call redone by the denominator / numerator fallback (crf_finalize_kernel)

// ---- build-time A/B switches of the frame loops (the defaults are the measured best: DESIGN.md section 2, profiles/round4_ab_*) ----
#ifndef CRF_X_GDEARLY
#define CRF_X_GDEARLY 1     // crf_grad_den_kernel: the rows of frame t+2 are requested right behind the staging of frame t+1 (0: at the top of frame t+1;
                            // 1: in the one-chunk, one-emission-register instantiation; 2: in every one-chunk instantiation not held to 128 VGPRs; 3: two-chunk ones too)
#endif
#ifndef CRF_X_GDMOVE
#define CRF_X_GDMOVE 1      // ... with the cur <- next register moves spelled out in front of the requests (0: left to the compiler, which waited for the rows
                            // of t+2 right behind their requests)
#endif
#ifndef CRF_X_GDW2
#define CRF_X_GDW2 1        // crf_grad_den_kernel, rows of 5121 .. 8192 floats: 1 = four row registers per thread at 128 VGPRs (two workgroups per CU)
#endif
#ifndef CRF_X_PRIO
#define CRF_X_PRIO 2        // fac_chain_body: issue priority by progress through the frame's chunks: 0 off, 1 steps at 1/4, 1/2, 3/4 of the
                            // chunks, 2 at 1/2, 3/4, 7/8 (product), 3 at 1/8, 1/4, 1/2 -- profiles/round4_ab_setprio_by_progress.txt
#endif
#ifndef CRF_X_EARLY
#define CRF_X_EARLY 1       // fac_chain_body: the frame's scale / exponent bookkeeping behind the first batch of gathers (0: in front of it)
#endif
#ifndef CRF_X_CTCSUM
#define CRF_X_CTCSUM 0      // crf_grad_ctc_kernel: mark a frame whose posteriors do not sum to one (built with the round-5 fixes; +16 % on that kernel, and what it
                            // caught is decided in front of the grad pass by crf_ctc_check_kernel and the frame factor's range check: the fuzz is green without it)
#endif
#ifndef CRF_X_CTCWPE
#define CRF_X_CTCWPE 5
#endif
#ifndef CRF_X_GCHK
#define CRF_X_GCHK 1        // crf_grad_den_kernel: the emission-weighted lost-term bound per frame (0: only "the frame's mass is a normal float"; A/B of what the check costs)
#endif
#ifndef CRF_X_GFIRST
#define CRF_X_GFIRST 0      // fac_chain_body (one CU per recursion): a frame BEGINS with its first batch of gathers -- everything else a frame starts with
                            // (stage check, emission prefetch, scale, exponent bookkeeping, row pointers: ~35 scalar instructions, ~4 cycles of a
                            // wave's issue each) follows behind a scheduling barrier, while the gathers are on their way -- profiles/round5_ab_*.txt
#endif
#ifndef CRF_X_KCLATE
#define CRF_X_KCLATE 0      // fac_chain_body, table geometries: the first slice's row constants are requested behind the first batch of gathers (0: at the frame top)
#endif
#ifndef CRF_X_ADDTID
#define CRF_X_ADDTID 0      // fac_chain_body, forward row epilogue: the four next-vector entries by ds_write_addtid_b32 (base in M0) instead of ds_write_b32 -- profiles/round6_ab_addtid.txt
#endif
#ifndef CRF_X_LAG
#define CRF_X_LAG 0         // fac_chain_body (one CU per recursion): the scale of frame t+1 is worked out in the TAIL of frame t from the maximum
                            // deposited in frame t-1 -- known before barrier t, so no frame starts with an LDS round trip for its scale (0: the scale of
                            // frame t from the maximum of its own source vector, read behind the barrier) -- profiles/round5_ab_lagged_scale.txt
#endif
// Lagged scale: the vector of frame t+1 is produced with a scale chosen before its size is known.  With u_t = (exponent of max X_t) + k_t
// the exponent of the SCALED source of frame t, the rule k_{t+1} = kLagTarget - u_t gives u_{t+1} = kLagTarget + kEpExp + g_t, where 2^g_t is
// what frame t's emissions and weights did to the maximum (g_t in [-8, 1] for ordinary network outputs): the scaled maximum of a frame depends
// on the growth of ONE earlier frame, nothing accumulates.  Any integer k is exact (a power of two; the exponent word E carries the sum), so
// only the range is at stake -- and not in the recursions first but in the GRAD pass, which multiplies a row of q ~ 2^u_t with a row of
// b ~ 2^u'_t and e' 2^-kEpExp ~ 2^g: with the unlagged rule both rows sit at 2^kScaleExp whatever the frame did, here they carry the frame's
// growth (found by tests/test_gpu_parity.py::test_lagged_scale_window: NaN gradients at 45 nats with the first thresholds).  Hence: a frame
// whose scaled maximum falls below 2^kLagLow -- it shrank the vector by more than 2^40, 28 nats below the row maximum in ONE frame, against
// 131 nats with the unlagged rule -- marks the utterance for the log-shifted fallback (crf_robust_den_kernel), as total underflow does.
constexpr int kLagTarget = -28;   // u_{t+1} = 36 + g_t (the next vector's maximum: 2^(100 + g_t + g_{t+1}) < 2^127)
constexpr int kLagLow = -4;       // u_t below this: q * b * e' could leave the fp32 range in the grad pass
struct LossParams {
    GraphDev g;
    const float *logp;
    const int *labels, *lab_off, *lx, *ly;
    int B, T, V;
    int res_lds_rows_f, res_lds_rows_b;  // max rows per CU (LDS carve of the resident kernels)
    int Sc;       // row stride of the ctc per-frame stores: 2*max_label_len+1 rounded up to 64
    float c_den, c_ctc;
    // workspace
    float *ep, *mx;               // [B*T*V] exp(logp - mx), [B*T] row max
    // fused log_softmax (crf_loss_fwd_bwd_logits): `logp` points to RAW logits of type in_dtype (0 f32, 1 bf16, 2 f16);
    // log_softmax(x)[v] - rowmax = x[v] - max x, so everything that works on differences to the row maximum is unchanged:
    // only the per-frame OFFSET that enters the log-likelihoods differs (moff = max x - lse x = -log sum exp(x - max x);
    // without fusion moff = mx), and the gradient w.r.t. x gets the softmax term of log_softmax's backward
    // (inv_s = 1 / sum exp(x - max x); applied by the numerator half of the grad pass, which runs once per call).
    int fused, in_dtype;
    float *moff, *inv_s;          // [B*T]
    float *Q, *BP;                // [B*T*Rq] q_t[row], [B*T*Rb] b_{t+1}[row]  (scaled)
    int Rq, Rb;                   // their row strides
    const int *gq, *gb;           // label-sorted pair list -> index into a Q row / a BP row
    const int *gchunk, *glab;     // its chunks (<= kChunk entries of one label) and per-label chunk ranges
    int gNC;
    int res;                      // 1: register-resident den kernels (g.res), 0: streaming kernels
    int gd_stage, gd_nb;          // crf_grad_den_kernel: process only the 16-frame blocks completed by den segment `gd_stage` (0 = all)
    int gd_nf;                    // > 0: the launch holds only 2 * gd_nf candidate blocks per utterance (see the kernel)
    int gd_bound[16];             //   segment k (1-based) runs the recursion iterations [gd_bound[k-1], gd_bound[k])
    int fin_fold;                 // crf_robust_grad_kernel: workgroup (0, 0) also does what crf_finalize_kernel does (which is then not launched)
    int gd_persist;               // crf_grad_den_kernel: 1 = ONE launch for the stages gd_stage .. gd_nb-1 -- 1-D grid, stage-major, the candidates of stage k
    int gd_poff[17];              //   from block gd_poff[k] on -- whose workgroups wait for their stage's counter themselves:
    const int *gd_cnt;            //   the den kernels' stage counters (FacParams::stage_cnt; fine-grained memory) ...
    int gd_target;                //   ... and the value that releases a stage (2 B: every recursion has published it)
    int *gd_timeout_host;         //   pinned host word set when a workgroup's wait times out (the context then goes back to per-stage launches)
    int gd_fpb[16];               //   frames per workgroup in stage k: kGDFrames, or a divisor of it -- the short last stages, where a workgroup's 16 frames one after the
                                  //   other (4 - 5 us each) would BE the tail: a block is still of the stage its 16 frames make it, and split among 16 / fpb workgroups there
    int grad_den_acc;             // crf_grad_den_kernel: 1 = add to the row (the numerator half has written it) instead of writing; 2 = atomic add into a
                                  // row the prep kernel has zeroed (the numerator half adds its part from ANOTHER stream at the same time)
    int zero_grad;                // crf_prep_kernel: zero the gradient rows (the two halves of the grad pass then ADD, in any order)
    int grad_phase;               // crf_grad_kernel: 0 = den and ctc in one pass, 1 = den part only (writes), 2 = ctc part only (subtracts)
                                  // (crf_grad_ctc_kernel: 0 writes, 2 subtracts from the row, 3 = atomic add into a zeroed row)
    int b0;                       // first utterance of this launch (resident kernels with K > 1 run in groups)
    unsigned long long *xch;      // [2][B][2][G] tagged granules for the K-way exchange of the state vector
    float *cb_part;               // [B][kResMaxK] partial backward partition sums
    double *cb_mxs;               // [B]
    int *cb_F;                    // [B]
    int *err;                     // [1] set if an exchange timed out
    int *clear; int nclear;       // words the prep kernel zeroes (error word, start and stage counters of the factored schedule)
    float *Row0;                  // [B][Rb] spare rows (b_0 of the resident backward recursion)
    float *dump; int dump_stride; // two utterances per workgroup (fac_chain_body2): [2][ceil(B / 2)][dump_stride] dump rows
    float *gvec;                  // streaming kernels, graphs too large for LDS: [B][3*Sp + 4*Pr] state vectors in global memory
    int grad_stage;               // crf_grad_kernel: 1 = stage the Q / BP rows in LDS, 0 = gather them from global memory
    int *EQ, *EB;                 // [B*T] their binary exponents
    double *CA, *CB;              // [B*T*Sc] ctc forward (incl. emission) / backward (excl.)  (scaled, fp64)
    int *ECA, *ECB;
    float *den_zs;                // [B] scaled partition sums
    double *ctc_zc;
    int *den_ez, *ctc_ez;         // [B] their exponents
    float *cost_alpha, *cost_beta, *cost_ctc;  // [B]
    int *invalid;                 // [B]
    int *redo;                    // [2][B] utterance whose scaled-fp32 denominator lost all its mass (forward / backward): redone by the robust kernels
    int force_redo;               // switch robust = 1: every utterance takes the robust path
    // Numerator fallback.  The fp64 chains rescale a frame's vector by its maximum; a frame whose posterior mass sits more than
    // ~650 nats below (max alpha) * (max beta) -- long utterances with many labels whose emissions do not follow the labels:
    // T = 3000, L = 500 on random inputs -- has the products that matter at the bottom of the fp64 range (and beyond: 0 * inf).
    // The grad kernels recognise such frames from the frame's exponents alone (ctc_frame_factor), contribute nothing for them
    // and mark them; crf_robust_ctc_kernel redoes the marked utterances' chains in the log domain and
    // crf_robust_ctc_fix_kernel adds the marked frames' posteriors.
    int *redo_ctc;                // [B] 0 = fine, 1 = some frames marked in ctc_bad, 2 = the whole utterance (the forward chain lost its mass, or forced)
    int *ctc_bad;                 // [B*T] marked frames (cleared by the prep kernel)
    int force_redo_ctc;           // switches robust = 1 / robust_ctc = 1: every utterance's numerator takes the log-domain path
    int *ctc_logdom;              // [B] 0, or the pass (1 = right behind the numerator's grad half, beside the denominator recursions of the
                                  // staged schedule; 2 = end of the call) in which the log-domain kernels redid the utterance: its CA / CB
                                  // rows and ctc_zc then hold LOGARITHMS
    int ctc_pass;                 // the pass of this launch of the log-domain kernels
    int ctc_tilt;                 // percent of the numerator chains' tilt (ctc_rho): 100 by default, 0 = none
    int *ctc_seen;                // host-visible word: the number of the last call in which an utterance took the log-domain chains
    int call_id;                  // this call's number (per context)
    int64_t gvec_stride;          // floats per utterance of `gvec`
    // outputs
    float *grad, *loss, *out_den, *out_beta, *out_ctc;
    int *out_invalid;
};

// Wave-wide max without touching the LDS crossbar (a __shfl_xor butterfly is six dependent
// ds_bpermute round trips, ~0.25 us on the per-frame critical path): rotate-and-max inside each row
// of 16 lanes with DPP (row_ror 8/4/2/1), then combine the four rows through readlane + scalar max.
__device__ __forceinline__ float wave_max(float v) {
#define CRF_DPP_MAX(ctrl) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false)))
    CRF_DPP_MAX(0x128);
    CRF_DPP_MAX(0x124);
    CRF_DPP_MAX(0x122);
    CRF_DPP_MAX(0x121);
#undef CRF_DPP_MAX
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// LDS float maximum without a returned value (ds_max_f32; the intrinsic, not atomicrmw: the compiler's atomic optimiser turns a
// same-address atomicrmw of several lanes into a scalar loop over the active lanes)
__device__ __forceinline__ void lds_fmax(float *lds_ptr, float v) {
    (void)__builtin_amdgcn_ds_fmaxf((__attribute__((address_space(3))) float *)lds_ptr, v, 0, 0, false);
}
// maximum over each row of 16 lanes of NON-NEGATIVE values, left in every lane of the row: on the float bits as integers (they
// order alike, and an integer maximum takes the DPP operand directly -- fmaxf costs two canonicalising v_max per step on top)
__device__ __forceinline__ float row_max16(float v) {
    int b = __builtin_bit_cast(int, v);
#define CRF_DPP_IMAXB(ctrl) b = max(b, __builtin_amdgcn_update_dpp(0, b, ctrl, 0xf, 0xf, false))
    CRF_DPP_IMAXB(0x128);
    CRF_DPP_IMAXB(0x124);
    CRF_DPP_IMAXB(0x122);
    CRF_DPP_IMAXB(0x121);
#undef CRF_DPP_IMAXB
    return __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Wave-wide max of NON-NEGATIVE doubles, exact in the top 32 bits (sign, exponent, 20 mantissa bits) --
// all the rescaling needs is the binary exponent.  Positive doubles order like their high words as
// integers, so this is an integer DPP max (no LDS round trips).
__device__ __forceinline__ double wave_max_d(double v) {
    int hi = (int)((unsigned long long)__double_as_longlong(v) >> 32);
#define CRF_DPP_IMAX(ctrl) hi = max(hi, __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xf, 0xf, false))
    CRF_DPP_IMAX(0x128);
    CRF_DPP_IMAX(0x124);
    CRF_DPP_IMAX(0x122);
    CRF_DPP_IMAX(0x121);
#undef CRF_DPP_IMAX
    const int m = max(max(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(hi, 16)),
                      max(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(hi, 48)));
    return __longlong_as_double((long long)(unsigned)m << 32);
}
// ... the same maximum as its high word (an int): what the chains' frame maximum is kept as in LDS (one ds_max_i32 per wave)
__device__ __forceinline__ int wave_max_hi(double v) {
    int hi = (int)((unsigned long long)__double_as_longlong(v) >> 32);
#define CRF_DPP_IMAX(ctrl) hi = max(hi, __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xf, 0xf, false))
    CRF_DPP_IMAX(0x128);
    CRF_DPP_IMAX(0x124);
    CRF_DPP_IMAX(0x122);
    CRF_DPP_IMAX(0x121);
#undef CRF_DPP_IMAX
    return max(max(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(hi, 16)), max(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(hi, 48)));
}
// rescale exponent from the HIGH WORD of a non-negative double maximum (0: nothing to scale by)
__device__ __forceinline__ int rescale_exp_hi(int hi) {
    if (hi <= 0) return 0;
    const int k = kScaleExpD - (((hi >> 20) & 0x7ff) - 1023);
    return k < -900 ? -900 : (k > 900 ? 900 : k);
}
// fp64 twin of rescale_exp / pow2f for the numerator chains
__device__ __forceinline__ int rescale_exp_d(double m) {
    if (!(m > 0.0)) return 0;
    int e = (int)(((unsigned long long)__double_as_longlong(m) >> 52) & 0x7ffull) - 1023;
    int k = kScaleExpD - e;
    return k < -900 ? -900 : (k > 900 ? 900 : k);
}
__device__ __forceinline__ double pow2d(int k) { return __longlong_as_double((long long)(k + 1023) << 52); }
// exp(d) * 2^add for d <= 0 without intermediate underflow: d = k ln2 + r, result = exp(r) * 2^(k+add)
__device__ __forceinline__ float exp_scaled(float d, int add) {
    const float k = rintf(d * 1.4426950408889634f);
    float r = fmaf(-k, 0.693145751953125f, d);
    r = fmaf(-k, 1.428606765330187e-06f, r);
    return ldexpf(expf(r), (int)k + add);
}
__device__ __forceinline__ double exp_scaled_d(float d) {
    const float k = rintf(d * 1.4426950408889634f);
    float r = fmaf(-k, 0.693145751953125f, d);
    r = fmaf(-k, 1.428606765330187e-06f, r);
    return ldexp((double)expf(r), (int)k);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it waits
// until every global store of the frame (the Q / BP rows) has been acknowledged by L2 -- about a
// microsecond per frame on a 1500-frame dependency chain.  Nothing inside the frame loops reads
// global data written by another wave of the same workgroup, so the LDS-only form is sufficient.
__device__ __forceinline__ void sync_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// LDS reads whose ISSUE point is fixed in the source (the compiler sinks an ordinary read to its first use, behind whatever is computed in
// between).  The value may be used only behind lds_landed() of the same variable -- the compiler does not know these are LDS operations and
// inserts no wait of its own; its waits for its own LDS operations stay correct (lgkmcnt counts in order: at worst they wait for these too).
__device__ __forceinline__ double lds_issue_f64(const double *q) {
    double v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"((unsigned)(uintptr_t)q));
    return v;
}
__device__ __forceinline__ int lds_issue_i32(const int *q) {
    int v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"((unsigned)(uintptr_t)q));
    return v;
}
__device__ __forceinline__ void lds_landed(double &a, double &b, double &c, int &w) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(w));
}

// LDS store of one float per lane to (uniform byte address `base`) + 4 * lane: the base goes through M0, the instruction carries no address VGPR.
// (1 wait state between the SALU write of M0 and the add-TID instruction; volatile asms keep their order, so the frame's sync_lds follows it.)
__device__ __forceinline__ void lds_st_addtid(float v, unsigned base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0" : : "v"(v), "s"(base));
}

// In-kernel phase timing for diagnosis (build with CRF_BUILD_DEFS=-DCRF_TIMING; tools/timing_probe.py):
// one chosen workgroup stamps s_memtime (shader cycles) at phase boundaries into g_tm, read back with
// crf_timing_read().  Compiled out of the product build.
#ifdef CRF_TIMING
__device__ unsigned long long g_tm[16384];
#define CRF_TM(on, idx) do { if (on) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) g_tm[(idx)] = t_; } } while (0)
#else
#define CRF_TM(on, idx) do { } while (0)
#endif

// exact power-of-two rescale that brings m into [2^kScaleExp, 2^(kScaleExp+1))
__device__ __forceinline__ int rescale_exp(float m) {
    if (!(m > 0.f)) return 0;
    int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;
    int k = kScaleExp - e;
    return k < -100 ? -100 : (k > 100 ? 100 : k);
}
__device__ __forceinline__ float pow2f(int k) { return __uint_as_float((unsigned)(k + 127) << 23); }
// the same from the float BITS of a non-negative maximum (integer operations only: on a wave-uniform value they run on the
// scalar unit; a float compare would not)
__device__ __forceinline__ int rescale_exp_bits(unsigned bits) {
    if (bits == 0u) return 0;
    const int k = kScaleExp + 127 - (int)((bits >> 23) & 0xffu);
    return k < -100 ? -100 : (k > 100 ? 100 : k);
}
// ... for a value the caller KNOWS to be wave-uniform (an SGPR): the clamp as s_max / s_min.  The compiler selects v_med3_i32 for the
// C form above even on uniform operands, which costs the frame loops a VALU instruction and three VGPRs (the two bounds and the result).
__device__ __forceinline__ int rescale_exp_bits_uniform(unsigned bits) {
    int k = kScaleExp + 127 - (int)((bits >> 23) & 0xffu);
    asm("s_max_i32 %0, %0, %1\n\ts_min_i32 %0, %0, %2" : "+s"(k) : "s"(-100), "s"(100) : "scc");
    return bits == 0u ? 0 : k;
}

// element `i` of the network output: fp32 log-probs (reference interface) or, fused, raw logits in fp32 / bf16 / fp16
__device__ __forceinline__ float ld_x(const LossParams &p, int64_t i) {
    if (p.in_dtype == 0) return p.logp[i];
    const unsigned short u = ((const unsigned short *)p.logp)[i];
    if (p.in_dtype == 1) return __uint_as_float((unsigned)u << 16);
    return (float)__builtin_bit_cast(_Float16, u);
}

// block-wide helpers for the 1024-thread chain workgroups --------------------------------------
__device__ __forceinline__ float block_sum(float v, float *red, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kChainWaves; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ double block_sum_d(double v, double *red, int tid) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < kChainWaves; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ double mx_total(const LossParams &p, int b, int lx, double *red, int tid) {
    double part = 0.0;
    for (int t = tid; t < lx; t += kChainThreads) part += (double)p.moff[(int64_t)b * p.T + t];
    return block_sum_d(part, red, tid);
}
__device__ __forceinline__ float frame_max(const float *wm) {
    float m = wm[0];
#pragma unroll
    for (int i = 1; i < kChainWaves; ++i) m = fmaxf(m, wm[i]);
    return m;
}
__device__ __forceinline__ float to_log(float zs, int e, double mxs) {
    return zs > 0.f ? (float)(log((double)zs) - (double)e * 0.6931471805599453 + mxs) : -INFINITY;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(3))) char lds_char;

// LDS carve (floats) shared by host sizing and the kernels
__host__ __device__ inline int rup64(int x) { return (x + 63) & ~63; }

// ---- numerator: what the chains, the grad pass and the log-domain fallback share ----
// LDS carve of the CTC chains: Abuf[2][Sxp] (double) | wmax[2][8] (double) | red[8] (double) | lab[Sxp] (int)
struct CtcLds {
    double *A, *wm, *red;
    int *lab;
};
__device__ __forceinline__ CtcLds ctc_carve(float *lds, int Sxp) {
    CtcLds c;
    c.A = (double *)lds;
    c.wm = c.A + 2 * Sxp;
    c.red = c.wm + 2 * kCtcWaves;
    c.lab = (int *)(c.red + kCtcWaves);
    return c;
}
constexpr int kCtcSafeExp = 900;
__device__ __forceinline__ float to_log_d(double zs, int e, double mxs) {
    return zs > 0.0 ? (float)(log(zs) - (double)e * 0.6931471805599453 + mxs) : -INFINITY;
}

// The factor that turns the scaled products A~_t[s] * Bx~_t[s] of frame `bt` into posteriors, or 0 with the frame MARKED for the
// log-domain fallback: A~ and Bx~ are rescaled to a maximum in [2^40, 2^41), so a factor beyond 2^900 means the entries that carry
// the frame's mass are ~2^-900 below the maxima -- at the bottom of the fp64 range, where they are rounded away or flushed (and the
// factor itself overflows: 0 * inf).  A frame that passes has every entry that matters to 1e-9 of the posterior as a normal number,
// at this frame and -- mass never grows along a path -- at every frame before it.
__device__ __forceinline__ double ctc_frame_factor(const LossParams &p, int b, int64_t bt, double invc, int ezc) {
    const int e = ezc - p.ECA[bt] - p.ECB[bt];
    // (atomicMax, not a store: blocks of ONE utterance run side by side, and a plain "= 1" here took back the 2 -- redo the whole utterance --
    // that the chains had set, after other blocks had already left their frames to the whole-utterance fix: frames with no numerator mass at
    // all, found by tests/test_gpu_fuzz.py in round 5)
    if (e + ilogb(invc) > kCtcSafeExp) { p.ctc_bad[bt] = 1; atomicMax(&p.redo_ctc[b], 1); return 0.0; }
    return ldexp(invc, e);
}
__device__ __forceinline__ float ctc_block_sum(float v, float *red, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kCtcWaves; ++i) s += red[i];
    return s;
}
// label sequence with blanks into LDS; false = not a valid label sequence for this utterance (L + repeats > T_b, gpu_ctc.h:161-174)
__device__ __forceinline__ bool ctc_setup(const LossParams &p, int b, const CtcLds &c, int L, int lx, int tid) {
    const int *ul = p.labels + p.lab_off[b];
    const int Sx = 2 * L + 1;
    float rep = 0.f;
    for (int s = tid; s < Sx; s += kCtcThreads) c.lab[s] = (s & 1) ? ul[s >> 1] : 0;
    for (int i = tid + 1; i < L; i += kCtcThreads) rep += (ul[i] == ul[i - 1]) ? 1.f : 0.f;
    const int repeats = (int)(ctc_block_sum(rep, (float *)c.red, tid) + 0.5f);  // also orders the lab[] writes
    __syncthreads();
    return lx > 0 && L + repeats <= lx;
}
// scaled partition sum of the numerator as the grad kernels use it: 0 (= "contribute nothing") for an utterance that is redone whole
__device__ __forceinline__ double ctc_zc_for_grad(const LossParams &p, int b) { return p.redo_ctc[b] == 2 ? 0.0 : p.ctc_zc[b]; }

}  // namespace crf
