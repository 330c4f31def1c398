/*
 * include/ctc_crf_hip.h -- C ABI of libctc_crf_hip.so, the MI355X-native (gfx950) CTC-CRF loss.
 *
 * This is the drop-in boundary for the reference's native layer L0:
 * plain pointers and sizes, no torch types.  It deliberately does NOT keep the reference's C
 * symbols (binding.cpp:20-49: Init / Release / compute_alpha / compute_beta_and_grad, and
 * gpu_ctc/ctc.h:76-109: compute_ctc_loss / get_workspace_size) because those bake in the
 * per-frame-launch design (separate alpha and beta entry points, [32]-striped grad_storage,
 * host-resident labels with two stream syncs).  Each entry point below names the reference
 * interface it replaces.
 *
 * Conventions
 *   - every pointer named *_dev is device memory on the graph's device; all work is enqueued on
 *     `stream` (a hipStream_t passed as void*); nothing here synchronises the host.
 *   - return value: 0 = CRF_OK, otherwise a crf_status; crf_last_error() gives the message
 *     (the reference printf()s and exit(1)s, den_calculate.cu:16-25, or drops the status,
 *     binding.cpp:111).
 *   - log_probs: [B][T][V] float32, contiguous (the reference's `logits`, ctc_crf/__init__.py:61);
 *     labels: flattened int32 without padding, blank = 0; lx/ly: int32 [B].
 */
#ifndef CTC_CRF_HIP_H_
#define CTC_CRF_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    CRF_OK = 0,
    CRF_ERR_IO = 1,          /* cannot open / parse the FST file                      */
    CRF_ERR_FORMAT = 2,      /* not an OpenFst vector/standard binary, or epsilon ilabel */
    CRF_ERR_ARG = 3,         /* bad argument (null pointer, V <= max label, ...)      */
    CRF_ERR_HIP = 4,         /* a HIP runtime call failed                             */
    CRF_ERR_WORKSPACE = 5,   /* workspace too small                                   */
    CRF_ERR_UNSUPPORTED = 6  /* graph / label length exceeds what this build handles  */
} crf_status;

typedef struct crf_graph crf_graph; /* opaque: the denominator graph, resident on one GPU */

/* Replaces Init(fst_name, n_gpus, gpus) (den_calculate.cu:288-392) + ReadFst (fst_read.cc:11-62),
 * for ONE device: reads an OpenFst vector/standard binary without OpenFst, applies the same
 * conventions (label = ilabel-1, weight = -cost, end_weight = -Final), builds the device arc
 * tables and uploads them.  Unlike the reference there is no process-global state: any number of
 * graphs may coexist; one is created per (den_lm, device). */
int crf_graph_create(const char *fst_path, int device, crf_graph **out);

/* Same, from arc arrays already in reference conventions (lab = ilabel-1, w = -cost, log domain;
 * start_w/end_w = -inf for non-start / non-final states). */
int crf_graph_create_from_arcs(int64_t num_states, int64_t num_arcs, const int32_t *src,
                               const int32_t *dst, const int32_t *lab, const float *w,
                               const float *start_w, const float *end_w, int device,
                               crf_graph **out);

/* Replaces Release(n_gpus, gpus) (den_calculate.cu:394-425). */
void crf_graph_destroy(crf_graph *g);

/* Replaces the globals DEN_NUM_STATES / DEN_NUM_ARCS (binding.cpp:14-15).  `num_pairs` is the
 * number of distinct (destination state, label) pairs, the unit the kernels store per frame. */
int crf_graph_dims(const crf_graph *g, int64_t *num_states, int64_t *num_arcs, int64_t *num_pairs,
                   int64_t *max_label);

/* Diagnostics of the compiled tables: out[0..9] = states, arcs, pairs, padded pair rows, padded state
 * rows, forward ELL arcs incl. padding, backward ELL arcs incl. padding, forward / backward LDS
 * gathers that share a bank with an earlier lane of their half-wave (extra LDS cycles per frame),
 * max_in_degree*100000 + max_out_degree; out[10..15] = register-resident layout: CUs per recursion K
 * (0 = graph too large, streaming kernels are used), forward / backward arc slots incl. padding,
 * forward / backward extra LDS cycles (busiest bank per half-wave gather), forward_rows*100000 + backward_rows;
 * out[16..23] = factored layout (one CU per recursion, T o LM structure): available (0/1), matched state pairs,
 * weights re-gauged (0/1), single-gather (tail) rows, forward / backward arc slots, fused backward rows, Gf*100000 + Gb;
 * out[24] = geometry of the factored kernels: 0 = 768 threads, row constants in registers (at most 3 slices of rows per wave),
 * 1 = 768 threads, row constants in an LDS table, 2 = 512 threads, 3 = as 1 over TWO compute units per recursion, 4 = 1024
 * threads with the table (four waves per SIMD: the planner's first choice); -1 = no factored layout; out[25] = chunks of four
 * arcs per thread in that geometry (20; 21 = the 768-thread table geometry with all chunk slots holding arcs, taken by graphs
 * that do not fit 20; 15 with 1024 threads; 30 with 512 threads).
 * A graph created with device < 0 is compiled on the host
 * only (no GPU needed) and can be used with crf_graph_dims / crf_graph_stats / crf_graph_destroy. */
int crf_graph_stats(const crf_graph *g, int64_t *out, int n);

/* Replaces get_workspace_size (ctc.h:99-109) and the torch::empty temporaries of gpu_den
 * (binding.cpp:77-79): bytes of device scratch crf_loss_fwd_bwd needs.  `g` may be NULL when
 * c_den == 0 (plain CTC).  `max_label_len` >= max(ly). */
int64_t crf_workspace_bytes(const crf_graph *g, int64_t B, int64_t T, int64_t V, int64_t max_label_len);

/* Which denominator kernels a call of this shape takes (no reference counterpart: the reference has one set of kernels,
 * den_calculate.cu:63-261, for every graph): 0 streaming, 1 generic register-resident (K CUs per recursion),
 * 2 factored register-resident, 3 utterance-minor; < 0 on error.  Diagnostics / bench labels. */
int crf_den_kernels(const crf_graph *g, int64_t B, int64_t T, int64_t V);

/* Test aid (no reference counterpart): builds the arc streams of the utterance-minor kernels for UL utterances per group and
 * about `want` tasks per direction ON THE HOST and checks them against the graph's row tables (every row once, records = arcs,
 * flags, task limits); works on host-only graphs.  out4 = {tasks, rows outside the streams, steps, arc records}, both
 * directions summed.  UL < 0: the FACTORED streams for -UL utterances per group (T o LM graphs; all zero for other graphs):
 * records, flags, the three descriptor words of every row and the bundles-per-task limit against the factored rows. */
int crf_debug_stream_check(const crf_graph *g, int UL, int want, int64_t *out4);

/* Test aid: the block -> (utterance group, direction, chunk) mapping of the utterance-minor frame kernel for a grid of
 * 8 * nslot workgroups and `ncombo` combos: 0 when every (combo, chunk) is taken by exactly one workgroup. */
int crf_debug_decode_check(int nslot, int ncombo);
/* Host check of the factored rows of the utterance-minor kernels (T o LM graphs; no GPU): one step of both recursions through
 * them equals the step through the plain tables on random vectors.  out4: {U entries, forward records, backward records, arcs};
 * all zero when the graph has no such rows. */
int crf_debug_facbatch_check(const crf_graph *g, int64_t *out4);
/* Test aid (no GPU): emulates the data flow of the factored register-resident kernels on the layout tables of a (host-only)
 * graph for T frames of random emissions -- packed arc words, slice ends, multi-lane rows, row constants, entries, second copy,
 * rowless states, and with two compute units per recursion exactly what crosses between them (a gather of anything else yields
 * NaN).  out3 = {sum over end states by the graph's own row tables, factored forward, factored backward}: all three agree. */
int crf_debug_fac_emulate(const crf_graph *g, int T, unsigned seed, double *out3);
/* Test aid (no GPU): the stage plan of the staged grad pass for utterances of up to T frames and a batch of B, as crf_loss_fwd_bwd makes it
 * under the current debug switches (piece, taper, stages, segments, gd_stage_launches, gd_sub).  out (n_out >= 76 ints): [0] number of stages,
 * [1] length of the equal pieces, [2] 1 when the stages 2.. are ONE launch, [3] workgroups of that launch, then four arrays of 18 ints indexed by
 * the stage: its end (bound[0] = 0 ... bound[nstage] = T, in iterations of the recursions), its first workgroup in the one launch, frames per
 * workgroup, candidate blocks per run.  tests/test_stage_plan.py walks the grid the way crf_grad_den_kernel does: every frame block once. */
int crf_debug_stage_plan(int64_t T, int64_t B, int32_t *out, int n_out);
/* The same for the GENERIC register-resident layout over K compute units (any graph that fits: rows = pairs forward, state
 * copies backward, one produced entry per row, every product exchanged): out3 = {sum by the recursion over the graph's arcs,
 * layout forward, layout backward}; the grad pass's pair lists are checked frame by frame as well. */
int crf_debug_res_emulate(const crf_graph *g, int T, unsigned seed, double *out3);

/* The hot path.  Replaces, in one call and with no host synchronisation:
 *   gpu_ctc  (binding.cpp:86-117  -> compute_ctc_loss, ctc_entrypoint.cu:29-60)
 *   gpu_den  (binding.cpp:65-84   -> compute_alpha + compute_beta_and_grad, den_calculate.cu:427-481)
 *   and the combine of _CTC_CRF.forward (ctc_crf/__init__.py:78-87).
 *
 *   grad_dev[b][t][v]  = c_den * gamma_den[b][t][v] - c_ctc * gamma_ctc[b][t][v]   (0 for t >= lx[b])
 *   loss_dev[0]        = sum_b ( c_den * logZ_den[b] - c_ctc * logp_ctc[b] )
 *   costs_den_dev[b]   = logZ_den[b]   (the reference's costs_alpha_den; may be NULL)
 *   costs_beta_dev[b]  = logZ_den[b] computed from the backward recursion (costs_beta_den; may be NULL)
 *   costs_ctc_dev[b]   = logp_ctc[b]   (+loglike, as the modified warp-ctc returns; may be NULL)
 *
 * CTC-CRF loss:  c_den = s, c_ctc = s*(1+lamb), s = 1/B if size_average else 1.
 * gpu_den alone: c_den = 1, c_ctc = 0.     gpu_ctc / WARP_CTC_LOSS: c_den = 0, c_ctc = s (g may be NULL).
 *
 * labels_dev: flattened labels; label_off_dev[b] = start of utterance b in it (int32 [B]).
 * Utterances the reference treats as invalid (L + repeats > T, gpu_ctc.h:166-174, where it returns
 * uninitialised memory) contribute logp_ctc = 0 and gamma_ctc = 0 and set invalid_dev[b] = 1
 * (invalid_dev may be NULL). */
int crf_loss_fwd_bwd(const crf_graph *g, const float *log_probs_dev, const int32_t *labels_dev,
                     const int32_t *label_off_dev, const int32_t *lx_dev, const int32_t *ly_dev,
                     int64_t B, int64_t T, int64_t V, int64_t max_label_len, float c_den, float c_ctc,
                     float *grad_dev, float *loss_dev, float *costs_den_dev, float *costs_beta_dev,
                     float *costs_ctc_dev, int32_t *invalid_dev, void *workspace_dev,
                     int64_t workspace_bytes, void *stream);

/* Replaces the cudaMemcpyAsync calls that bring labels, label lengths and input lengths to the device
 * (gpu_ctc.h:143-229; `input_lengths.cuda()`, ctc_crf/__init__.py:73): copies n int32 from PINNED host
 * memory (hipHostMalloc / torch pin_memory: device-accessible) to device memory with a kernel on `stream` --
 * no DMA engine start-up between two calls.  The host buffer must stay untouched until the stream has passed. */
int crf_stage_i32(int32_t *dst_dev, const int32_t *src_pinned_host, int64_t n, void *stream);

/* crf_loss_fwd_bwd with the log_softmax in front of it fused in.  Replaces, in addition,
 *   logits = torch.log_softmax(netout, dim=-1)   and   criterion(logits.float(), ...)   (cat/ctc/train.py:174-186)
 * and log_softmax's backward: `logits_dev` are the RAW network outputs [B][T][V], dtype 0 = fp32, 1 = bf16, 2 = fp16
 * (upcast in registers; the recursions, costs and the gradient stay fp32/fp64), and
 *   grad_dev[b][t][v] = d loss / d logits[b][t][v]
 *                     = (c_den gamma_den - c_ctc gamma_ctc) - softmax(logits)[v] * sum_v (c_den gamma_den - c_ctc gamma_ctc)
 * in fp32.  Everything else as crf_loss_fwd_bwd; c_ctc must not be 0 (the softmax term rides on the numerator half of
 * the grad pass).  Costs are those of log_softmax(logits). */
int crf_loss_fwd_bwd_logits(const crf_graph *g, const void *logits_dev, int dtype, const int32_t *labels_dev,
                            const int32_t *label_off_dev, const int32_t *lx_dev, const int32_t *ly_dev,
                            int64_t B, int64_t T, int64_t V, int64_t max_label_len, float c_den, float c_ctc,
                            float *grad_dev, float *loss_dev, float *costs_den_dev, float *costs_beta_dev,
                            float *costs_ctc_dev, int32_t *invalid_dev, void *workspace_dev,
                            int64_t workspace_bytes, void *stream);

/* Diagnostics (no reference counterpart; the reference has no profiler hooks).
 * crf_profile_enable(1): every following crf_loss_fwd_bwd on this thread brackets each of its
 * kernel launches with HIP events on the stream the kernel is launched on.
 * crf_profile_read synchronises on those events and returns, for the LAST call, up to `n`
 * durations in milliseconds in the fixed order
 *   [0] prep  [1] den forward chain  [2] den backward chain  [3] ctc forward chain
 *   [4] ctc backward chain  [5] grad  [6] finalize  [7] whole call (first launch .. last launch)
 * (-1 for kernels that were not launched).  Returns the number of slots written. */
void crf_profile_enable(int on);
/* The template instantiation of the kernel that ran the denominator recursions in this thread's last call, e.g.
 * "crf_fac_pair_kernel<true,768,21,4,4,false,false>" (the prefix of the name rocprofv3 reports): bench.py keys its committed PMC
 * traffic numbers by workload AND by this string. */
const char *crf_last_den_kernel(void);
/* Streams this thread's last call put work on: 1 = the caller's only, 2 = + the context's side stream, 3 = + its third stream
 * (the numerator's log-domain fallback chains beside the staged grad pass: taken when a recent call of the context needed them). */
int crf_last_call_streams(void);
/* What the side stream of the (device, caller stream) context of this thread's last call is: "plain (candidate 2)", "priority-low
 * (candidate 9)", "cu-mask (candidate 13) + third stream", "none (candidate 13)" -- the kind of stream that was found to run
 * beside the caller's and how many candidates had been probed by then (crf_kernels.hip find_beside).  The reference runs everything
 * on the caller's stream (binding.cpp:75,102) and has nothing to report. */
const char *crf_last_side_stream(void);
/* How many utterances of this thread's last call were redone by a fallback: out2[0] = denominator (crf_robust_den_kernel: the scaled fp32
 * recursions lost the utterance's mass, or -- lagged scale -- a frame shrank the vector by more than 2^90), out2[1] = numerator
 * (crf_robust_ctc_kernel: frames outside the fp64 range of the rescaled chains).  Synchronises `stream` (the stream of that call) and
 * copies two words: a diagnostic for benchmarks and tests, not for the training loop.  LIFETIME: the words live in that call's workspace
 * (or, with fine-grained flag words, in the buffer of its (device, stream) context): ask right after the call, before the workspace is
 * freed or handed to another call and before another thread calls on the same device and stream -- later the counts are another call's
 * or the pointer dangles.  The reference has no fallback: its log-domain kernels (den_calculate.cu:29-35) pay exp + log1p on every arc
 * instead. */
int crf_last_fallback_counts(int32_t *out2, void *stream);
/* The build-time A/B switches of the frame loops this library was compiled with, e.g. "LAG=1 KCLATE=0 PRIO=2 EARLY=1 ..." (crf_kernels.hip,
 * CRF_X_*: the defaults are the measured best; tools build variants with CRF_BUILD_DEFS=-DCRF_X_...=n and tests ask which one they run). */
const char *crf_build_switches(void);
int crf_profile_read(float *ms_out, int n);

/* Diagnostics, timing builds only (CRF_BUILD_DEFS=-DCRF_TIMING python -m cat_amd.build --force): copies
 * the in-kernel phase stamps (shader cycles, s_memtime) of the last call into `out`; returns the number of
 * values, 0 in a product build.  tools/timing_probe.py decodes them.  No reference counterpart. */
int crf_timing_read(unsigned long long *out, int n);

/* Debug / experiment switches of tests and tools (no reference counterpart).  The library NEVER reads the process
 * environment: every switch is set here, by name, process-wide; crf_debug_unset returns it to its default.  Switches marked
 * G are read when a graph is created, C per loss call, X when a (device, stream) context is first used;
 * crf_debug_list() returns one "name: when  what" line per switch.  CRF_ERR_ARG for an unknown name. */
int crf_debug_set(const char *key, int value);
int crf_debug_unset(const char *key);
const char *crf_debug_list(void);

/* Message for the last non-zero status returned on this thread. */
const char *crf_last_error(void);

/* Library version string, e.g. "ctc_crf_hip 0.1.0 (gfx950)". */
const char *crf_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CTC_CRF_HIP_H_ */
