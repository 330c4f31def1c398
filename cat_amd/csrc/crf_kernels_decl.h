// cat_amd/csrc/crf_kernels_decl.h -- what the host side (crf_host.hip) needs to know of the kernel families: their argument blocks, the constants
// that size grids and LDS, and the DECLARATIONS of the kernel templates.  The definitions live in k_*.hip, one translation unit per family,
// each of which instantiates explicitly what the host launches.
#pragma once
#include "crf_device.h"

namespace crf {

// ---- k_chain.hip ----
template <int G> __global__ void crf_prep_kernel(LossParams p);
__global__ void crf_stage_i32_kernel(int *__restrict__ dst, const int *__restrict__ src, int64_t n);
__global__ void crf_gate_kernel(const int *started, int target);
template <bool GV> __global__ void crf_den_pair_kernel(LossParams p);
template <int NR> __global__ void crf_ctc_pair_kernel(LossParams p);
__global__ void crf_ctc_check_kernel(LossParams p);

// ---- k_res.hip ----
constexpr int kEpRegsR = 2;   // emission-row prefetch registers (V <= 2*512 for the resident kernels)
constexpr int kPoll = 4;      // granules polled concurrently per thread
constexpr int kResBatch = 6;   // measured: 5 -> 6 = -2% (fewer, longer straight-line blocks); 10 spills
static_assert(kResNCH % kResBatch == 0, "kResNCH must be a multiple of kResBatch");
// Kernel arguments of the resident kernels: only what ONE direction needs (the full LossParams is ~90
// SGPRs of pointers, most of which the compiler would keep live or spill around the unrolled frame body).
struct ResParams {
    ResDirDev L;
    int K, B, T, V, b0, rows_cu_max, Rout, Gf, Gb;
    const int *lx;
    const float *ep, *mx;
    float *Out;                 // Q (fwd) or BP (bwd) rows
    float *Row0;                // [B][Rout] spare rows: b_0 of the backward recursion
    int *Eout;                  // EQ (fwd) or EB (bwd)
    unsigned long long *xch;
    int *err;
    // forward side tables / results
    const float *x_start, *x_end;
    float *den_zs, *cost_alpha;
    int *den_ez;
    // backward side tables / results
    const int *z_lab;
    const float *z_end, *brow_start, *brow_end;
    const int2 *bcsr;
    float *cb_part;
    double *cb_mxs;
    int *cb_F;
    int *redo;                  // [2][B], see LossParams
};

// LDS map of the resident kernels: the two state-vector buffers sit at byte offsets 0 and kResXB; a gather
// is `ds_read_b32 v, (buffer base SGPR + 16-bit offset from the packed arc word)` -- one VALU (an SDWA add)
// per arc besides the FMA.
constexpr int kResXB = 65536;                 // bytes per state-vector buffer  -> gather vector <= 16384 entries (16-bit byte offsets);
                                              // 32 KiB until round 2: graphs of 8 k - 16 k states fell to the utterance-minor kernels (27 ms
                                              // instead of ~14 at S = 8193 / A = 208 k)
constexpr int kResGmax = kResXB / 4;
__global__ void crf_res_pair_kernel(ResParams pf, ResParams pb);

// ---- k_fac.hip ----
// chunks gathered per batch (template arguments of the instantiations the host launches and k_fac.hip instantiates)
#ifndef CRF_FAC4_NB
#define CRF_FAC4_NB 2       // chunks gathered per batch by the 1024-thread kernels (3: one weight pair spills INSIDE the frame loop, behind a vmcnt(0))
#endif
#ifndef CRF_FAC4_NB_ML
#define CRF_FAC4_NB_ML 2    // ... with multi-lane rows: the butterfly's registers make batches of 3 spill (V = 217: recursions 3.32 -> 3.02 ms)
#endif
#ifndef CRF_FAC5_NB2
#define CRF_FAC5_NB2 4      // ... by the two-utterance kernel on 512 threads x 30 chunks (256 registers per wave: 16 ds_read_b64 = 32 registers in flight)
#endif
#ifndef CRF_FAC3_NB2
#define CRF_FAC3_NB2 2      // chunks gathered per batch by the two-utterance kernels (8 ds_read_b64 = 16 registers in flight)
#endif
#define CRF_STR_(x) #x
#define CRF_STR(x) CRF_STR_(x)
#ifndef CRF_FAC3_NB_F
#define CRF_FAC3_NB_F 4
#endif
#ifndef CRF_FAC3_NB_B
#define CRF_FAC3_NB_B 4
#endif
struct FacParams {
    FacDirDev L;
    int B, T, V, Rout, NT, Rf;
    const int *lx;
    const float *ep, *mx;
    float *Out;                 // Q (fwd) or BP (bwd) rows
    float *Row0;                // [B][Rout] spare rows: b_0 of the backward recursion
    int *Eout;                  // EQ (fwd) or EB (bwd)
    int *started;               // workgroups of the den kernels that have started (gate for the numerator chains)
    int i0, i1;                 // iterations of this launch (segment)
    int nb;                     // stage bounds (iterations), bound[0] = 0 < ... < bound[nb-1] >= T; nb <= 1: no stage flags
    int bound[16];
    int *stage_cnt;             // [16] fine-grained counters: += 1 per utterance when the rows of all iterations < bound[k] are in memory
    float *state;               // [B][rup64(G) + 64] parked state vector and exponent between segments
    const int4 *frow_meta;
    const float *x_start, *x_end;
    float *den_zs, *cost_alpha;
    int *den_ez;
    const int4 *brow_meta;
    const int *z_lab;
    const float *z_end, *brow_start, *brow_end;
    const int *bx_idx; const float *bx_w; int nbx; float bx_se;   // rowless states of the backward recursion (FacDev)
    float *cb_part;
    double *cb_mxs;
    int *cb_F;
    int *redo;                  // [2][B], see LossParams
    // two CUs per recursion (FacDev::K = 2): the utterances [b0, b0 + nbu) of this launch, the exchange granules, the error word
    int K, b0, nbu, Gf, Gb;
    const int *xlist;           // forward: the L / A entries this CU fetches every frame (FacDev::xlist), [xl0, xl1)
    int xlist_off[3];
    unsigned long long *xch;
    int *err;
    // two utterances per workgroup (fac_chain_body2): pairs of this launch, rows that take the stores of an utterance that has ended
    int npair, dump_stride;
    float *dump;                // [2 directions][npair][dump_stride]
};
template <bool FLAG, int NTH, int NCH, int NBF, int NBB, bool ML, bool RL = false> __global__ void crf_fac_pair_kernel(FacParams pf, FacParams pb);
template <bool FLAG, int NTH, int NCH, int NBF, int NBB, bool ML, bool RL> __global__ void crf_fac_pair2_kernel(FacParams pf, FacParams pb);
template <int NTH, int NCH, int NBF, int NBB> __global__ void crf_fac2_pair_kernel(FacParams pf, FacParams pb);

// ---- k_grad.hip ----
#ifndef CRF_GD_FRAMES
#define CRF_GD_FRAMES 16
#endif
constexpr int kGDThreads = 256, kGDFrames = CRF_GD_FRAMES, kGDRowRegs = 5;   // rows of up to 5*256 float4 = 5120 floats
constexpr int kGDEpRegs = 4;                                      // V <= 4*256
constexpr int kGCThreads = 256, kGCFrames = 16, kGCRegs = 16, kGCVRegs = 4;  // 2L+1 <= 4096, V <= 1024
__global__ void crf_grad_kernel(LossParams p);
template <int NCPT, int EPR, int NT = kGDThreads, int CH = kChunk, int RR = kGDRowRegs, int WPE = 1> __global__ void crf_grad_den_kernel(LossParams p);
template <int REGS> __global__ void crf_grad_ctc_kernel(LossParams p);

// ---- k_batch.hip ----
struct BatchParams {
    BatchDev g;
    StreamDev st;                  // arc streams for AL = 64 / UL lane groups
    const float *start_lin, *end_lin;
    const float *x_start;          // [SX] a_0 of the forward vector's entries: the S states, then the U entries of factored streams
    int S, P, B, Bp, T, V, max_label, ngrp;
    int SX;                        // entries of the forward vector (S + StreamDev::NU)
    const int *lx;
    const float *ep, *moff;        // [B][T][V] e' (prep kernel), [B][T] log-likelihood offset per frame
    float *ept;                    // [T][grp][V][UL] e' transposed
    float *Af, *Zb;                // [2][grp][SX][UL], [2][grp][P][UL]
    float *Q, *BP;                 // [T][grp][P][UL]
    unsigned *mxf, *mxb;           // [3][Bp] maxima of the vectors (float bits; the values are non-negative)
    int *Ef, *Fb;                  // [Bp] running exponents
    float *zs, *zb;                // [Bp] scaled partition sums
    float *den_zs, *cost_alpha, *cost_beta;
    int *den_ez, *redo;
    float *grad;                   // [B][T][V]
    float c_den;
    int j;                         // launch number: forward frame j, backward frame T - j
    unsigned *bar;                 // crf_batch_persist_kernel: [512] words of the grid barrier, zeroed by crf_batch_init_kernel
    int *err;                      // ... its time-out word (crf_batch_cost_kernel turns the costs into NaN)
};
constexpr int kBatThreads = 256, kBatWaves = kBatThreads / kWave;
constexpr int kStreamBatch = 4, kStreamChunk = 128;   // steps per batch; batches * AL per 4 KB chunk (batches in flight: template parameter D)
constexpr int kStreamBundles = 8;                     // bundles per task at most (plain streams; factored: crf_internal.h stream_max_bundles)
constexpr int kStreamRecB = kStreamChunk * 32;        // bytes of a chunk of records
// Factored streams (NM = 3 descriptor words per row, crf_internal.h StreamDev): the ring holds NR values per row instead of the
// one emission row -- forward {e[label 0], e[label 1], U of the row's couple}, backward {e[label 0], e[label 1], z of the two
// extra arcs} -- all known from the descriptor, so all requested a bundle ahead like the emissions; epi gets them as e[NR].
constexpr int kStreamLds = 2 * kStreamRecB + 2 * 64 * 16 + kStreamBundles * 32 * 16;   // per wave: records (2 halves) | emission ring | descriptors
// (factored streams of small utterance groups -- many rows side by side -- have tasks of fewer bundles, so that a workgroup's
// four slices stay below half of the LDS: crf_internal.h stream_max_bundles, shared with the host's task cutter)
template <int UL, bool FAC>
constexpr int stream_lds() { return FAC ? 2 * kStreamRecB + 2 * 64 * 16 * 4 + stream_max_bundles(UL, true) * (256 / UL) * 48 : kStreamLds; }
template <int UL> __global__ void crf_batch_transpose_kernel(BatchParams p);
__global__ void crf_batch_init_kernel(BatchParams p);
template <int UL, int D, bool FAC = false> __global__ void crf_batch_frame_kernel(BatchParams p);
template <int UL, int D, bool FAC = false> __global__ void crf_batch_persist_kernel(BatchParams p);
template <int UL> __global__ void crf_batch_zsum_kernel(BatchParams p);
__global__ void crf_batch_cost_kernel(BatchParams p);
template <int UL> __global__ void crf_batch_grad_kernel(BatchParams p);

// ---- k_robust.hip ----
__global__ void crf_den_check_kernel(LossParams p, int parts);
template <bool GV> __global__ void crf_robust_den_kernel(LossParams p);
template <int NR> __global__ void crf_robust_ctc_kernel(LossParams p);
__global__ void crf_robust_ctc_fix_kernel(LossParams p);
__global__ void crf_finalize_kernel(LossParams p);
__global__ void crf_robust_grad_kernel(LossParams p);

}  // namespace crf
