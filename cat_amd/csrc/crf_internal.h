// cat_amd/csrc/crf_internal.h -- shared between the host-side graph compiler (fst_graph.cpp) and the
// gfx950 kernels (crf_kernels.hip).  Not part of the public ABI (include/ctc_crf_hip.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace crf {

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kChainWaves = 16;    // waves per chain workgroup (1024 threads = one full CU)
constexpr int kChainThreads = kChainWaves * kWave;
constexpr int kChunk = 32;         // pairs per grad-pass chunk
constexpr int kScaleExp = 20;      // den per-frame rescale target: max entry in [2^20, 2^21)
constexpr int kEpExp = 64;         // den emission factors are stored as exp(logp - rowmax) * 2^64, so
                                   // they stay normal fp32 numbers down to e^-131 below the row max
constexpr int kScaleExpD = 40;     // ctc (fp64) per-frame rescale target

// One direction of the recursion as sliced-ELL: rows are grouped in slices of 64 (one wave), rows
// sorted by degree so a slice pads only to its own widest row; two arcs per 16-byte element
// {idx0, w0, idx1, w1}, element (kk, lane) of slice j at arcs[slice_off[j] + kk*64 + lane] so a
// wave's load is one contiguous 1 KiB.  Padding arcs are {0, 0.0f}.
struct EllDev {
    const uint4 *arcs;
    const int *slice_off;   // [nslices] in uint4 units
    const int *slice_w2;    // [nslices] arc-pairs per row in this slice
    const int *wave_off;    // [kChainWaves+1] -> range in wave_slices owned by each wave
    const int *wave_slices; // [nslices] slice ids, longest-processing-time balanced over waves
    int nslices;
};

// ---------------------------------------------------------------------------------------------
// Register-resident layout of one recursion direction, split over K compute units per utterance.
// Each CU owns a set of rows; a thread keeps the arcs of its rows in VGPRs for the whole kernel
// (kResNCH chunks of kResW arcs: per chunk 2 words of packed 16-bit LDS byte offsets + 4 weights).
// Rows are grouped in slices of 64 equal-length rows (one per lane); a wave's slices are laid end to
// end along the chunk axis and a per-wave bit mask marks the chunk after which a slice (row) ends.
// Row ids are numbered CU by CU, wave by wave, slice by slice, lane by lane, so a wave's row results
// are 64 consecutive floats of the per-frame HBM row.
// ---------------------------------------------------------------------------------------------
constexpr int kResThreads = 512;
constexpr int kResWaves = kResThreads / kWave;
constexpr int kResW = 4;
constexpr int kResNCH = 30;
constexpr int kResWords = kResNCH * 6;
constexpr int kResMaxK = 4;
// second geometry of the factored kernels: 12 waves = 3 per SIMD, 21 chunks per thread (<= 168 VGPRs)
constexpr int kFac3Threads = 768;
constexpr int kFac3NCH = 21;       // chunk slots (6 words each) per thread: 20 hold arcs, the last holds row constants
constexpr int kFac3ArcCh = 20;     // chunks of arcs per thread
#ifndef CRF_FAC3L_NCH
#define CRF_FAC3L_NCH 21
#endif
constexpr int kFac3LNCH = CRF_FAC3L_NCH;   // chunks of arcs per thread with the row constants in the LDS table (all 21 slots hold arcs)
#ifndef CRF_FAC4_NCH
#define CRF_FAC4_NCH 15
#endif
constexpr int kFac4Threads = 1024, kFac4NCH = CRF_FAC4_NCH;   // four waves per SIMD at <= 128 VGPRs, chunks of arcs per thread, row constants in the LDS table (the planner's first choice)
constexpr int kFac3MaxSl = 3;      // slices (row epilogues) per wave: two words of row constants each, at word kFac3ArcCh * 6 on

struct ResDirDev {
    const unsigned *arcs;    // [K][kResWords][kResThreads]
    const uint4 *wave_info;  // [K][kResWaves] {slice-end mask, chunks used, first row id, unused}
    const int *row_lab;      // [R] label whose emission scales the row's result (fwd: the pair's label; bwd: the
                             //     label of the pair the row produces z for); V (= "emission 0") for rows that
                             //     produce nothing.  The gather entry a row produces is IMPLICIT:
                             //     own_off[k] + (row id - cu_row_off[k]) -- every row, padding included, owns one.
    const int *cu_row_off;   // [K+1] row-id range of each CU
    const int *own_off;      // [K+1] range of gather-vector indices PRODUCED by each CU
    const int *ex_cnt;       // [2K] per CU: [k] produced (= exchanged) entries = its rows; [K+k] unused
    int R;                   // rows incl. padding = row stride of the per-frame HBM store
    int G;                   // gather-vector length (fwd: states, bwd: pairs)
};
struct ResDev {
    int K;                    // 0 = resident layout not available (graph too large): streaming kernels
    ResDirDev f, b;
    const float *x_start;     // [S] exp(start weight), forward x-index order
    const float *x_end;       // [S] exp(end weight),   forward x-index order
    const int *z_lab;         // [P] label of each z entry (backward z-index order)
    const float *z_end;       // [P] exp(end weight) of the pair's destination state
    const float *brow_start;  // [Rb] exp(start weight) of the row's state
    const float *brow_end;    // [Rb] exp(end weight) of the row's state
    const int2 *bcsr;         // {z index, label} lists for states entered with more than one label
    const int *gq, *gb;       // [NR] grad pass over label-sorted forward (sub-)rows: index into the Q row / BP row
    const int *chunk_off;     // [NC+1] chunks (<= kChunk entries, one label each) of that list
    const int *lab_chunk_off; // [max_label+2]
    int NC;
};

// ---------------------------------------------------------------------------------------------
// FACTORED register-resident layout: ONE compute unit per recursion (no exchange at all), made possible
// by two structural facts of a CTC-topology o LM graph (den_lm = T o LM: every LM state g appears as the
// states (g, blank) and (g, last token)), found by the graph compiler without being told:
//  * forward: the two states of such a pair feed the same rows with the same weights, and the row of
//    (g, blank) is just w * (sum of the pair): the recursion carries the pair SUM as an entry of its own, so
//    one gather and one weight serve two arcs, and that row is folded into the epilogue of its mate's row;
//  * backward: the two states have the same out-arcs but (at most) one each, so ONE row computes the
//    common sum and its epilogue adds each state's extra arc: two outputs per row.
// Both halve the arc registers: the 104 k-arc benchmark graph fits 512 threads x 180 words per direction.
// Graphs without that structure (or with states entered by several labels) keep the generic layout.
// ---------------------------------------------------------------------------------------------
struct FacDirDev {
    const unsigned *arcs;    // [words][threads]
    const uint4 *wave_info;  // [waves] {slice-end mask, chunks used, first row id, unused}
    int R;                   // rows incl. padding (multiple of 64)
    int G;                   // gather-vector entries (floats) incl. the sink pair at [G-2, G-1]
    int dup;                 // byte distance of the SECOND copy of the gathered entries (other LDS banks), see pack_arcs
    int cu_row[3];           // FacDev::K = 2: rows [cu_row[k], cu_row[k + 1]) belong to CU k (arcs and wave_info are [K][...])
};
struct FacDev {
    int ok;                    // 0 = not available for this graph
    FacDirDev f, b;
    // forward: rows = pairs except "tail" rows; a (tail, main) state pair keeps three entries U = sum, L, A.
    const int4 *frow_meta;     // [Rf] {U byte offset | main label << 16, L offset | A offset << 16, tail weight bits, tail label}
                               //      plain rows: U = A = sink, tail weight 0
    int NT;                    // unused (0)
    int threads;               // workgroup size the tables were built for: 1024 (15 chunks per thread), 768 (21) or 512 (30)
    int imp;                   // the entries of a row lie where its row id says (every geometry but the 512-thread FALLBACK; the 512-thread
                               // layout of the two-utterance kernel, HostGraph::facp, is implicit too)
    int K;                     // CUs per recursion: 1, or 2 (graphs of 120 k - 240 k arcs: each CU holds half of the rows and the whole
                               // state vector, the products cross through L2 every frame like the generic layout's; rcl geometry only)
    const int *xlist;          // K = 2, forward: the L / A entries (and plain states) CU k fetches from its peer every frame,
    int xlist_off[3];          //   [xlist_off[k], xlist_off[k + 1]); the U entries cross as a range, everything once more at the end
    int rcl;                   // 768 threads with the row constants in an LDS table read one slice ahead (any number of slices per wave)
    int multilane;             // some rows lie on several adjacent lanes (wave_info.w != 0 somewhere): kernel variant with the butterfly
    const float *x_start, *x_end;   // [Gf]
    // backward: rows = one or two states with common out-arcs; z entry of output o of row r = 2r + o.
    const int4 *brow_meta;     // [Rb] {extra-arc z byte offset 0 | offset 1 << 16, weight 0 bits, weight 1 bits, label 0 | label 1 << 16}
    const int *z_lab;          // [Gb] label of each z entry (V = none)
    const float *z_end;        // [Gb] exp(end weight) of the entry's state
    const float *brow_start, *brow_end;  // [2*Rb] per output
    const int *bx_idx; const float *bx_w; int nbx; float bx_se;   // states without a row (nobody enters them): z entry and start * weight of each
                               // of their arcs, added to the backward logZ after the last frame; bx_se = their start * end (empty utterance)
    // grad pass
    const int *gq, *gb, *chunk_off, *lab_chunk_off;
    int NC;
    int chunk_cap;             // entries per chunk at most: kChunk, or 8 for graphs with few pairs per label (crf_grad_den_kernel<..., 8>)
    int Rq;                    // Q row stride = 2*Rf: main rows, then the tail row of each main row
    int Rbp;                   // BP row stride = 2*Rb
};

// ---------------------------------------------------------------------------------------------
// UTTERANCE-MINOR ("batch") layout for graphs that do not fit the register-resident layouts (S = 16 k ... 65 k states,
// BASELINE config #5): the state vectors live in global memory as [state][utterance], a wave works on one row with the
// utterances in its lanes, and every arc is read ONCE per frame for the whole batch.  One launch per frame (the kernel
// boundary is the grid barrier), forward and backward recursion in the same launch.
// Pairs are numbered in (label, destination) order, so the pairs of a label are a contiguous range (grad pass).
// ---------------------------------------------------------------------------------------------
struct BatchDev {
    int ok;                  // 0: tables not built
    // forward rows = destination states, in processing order (most arcs first); a row's arcs are contiguous
    const int *frow_d;       // [S] state of the k-th row
    const int4 *frow;        // [S] {first arc, arc end, first entry in stp, entry end}; ONE entering pair: {.., .., pair id, label | 1 << 30}
    const int4 *stp;         // [P] {pair id, label, first arc, arc end}, grouped by destination state
    const int2 *farcs;       // [A] {source state, weight bits}: grouped by pair, the pairs of a destination state adjacent
    // backward rows = source states, in processing order (most arcs first)
    const int *brow_s;       // [S]
    const int4 *brow;        // [S] {first arc, arc end, first entry in stp (pairs ENTERING the state), entry end}; one: as frow
    const int2 *barcs;       // [A] {pair id, weight bits}, grouped by source state
    const int *lab_off;      // [max_label+2] pair-id range of each label
};

// Arc STREAMS of the utterance-minor kernels (built on first use for UL utterances per group, ensure_stream_tables).  A
// lane takes FOUR utterances (one 16-byte gather), UL / 4 lanes a row, so a wave walks AL = 256 / UL rows side by side:
// the rows with exactly one entering pair (every state of a T o LM graph), most arcs first, in BUNDLES of AL rows, all AL
// rows padded with null records to the same whole number of BATCHES of 4 steps.  A wave's TASK is a run of whole bundles
// (at most 8); its records are one contiguous stream
//   [batch][AL lane groups][4 records {entry * UL | flag, weight bits}]     (bit 31 of a batch's first record: the bundle ends)
// which the wave copies to LDS 4 KB at a time, one chunk ahead: its memory queue holds the gathers of the state vector --
// D batches deep -- and, at bundle ends, a row's stores and the emissions of the bundle after the next.
// Which (utterance group, direction) combo a workgroup of crf_batch_frame_kernel works for, and which of the combo's
// workgroups ("chunks") it is -- from its block id alone.  Block b sits on XCD b % 8 (observed; a matter of speed only):
//   #combos <  8: XCD x serves combo x % #combos together with the other XCDs of that residue, the combo's chunks dealt
//                 round-robin among them;
//   #combos >= 8: XCD x serves the combos x, x + 8, ..., its slots dealt round-robin among them.
// grid = 8 * nslot blocks.  Every (combo, chunk < nchunk(combo)) is taken by exactly one block (tests/test_abi.py).
__host__ __device__ inline void bat_decode(int block, int grid, int ncombo, int *combo, int *chunk, int *nchunk) {
    const int x = block & 7, slot = block >> 3, nslot = grid >> 3;
    if (ncombo >= 8) {
        const int ncx = (ncombo - x + 7) >> 3, k = slot % ncx;
        *combo = x + 8 * k; *chunk = slot / ncx; *nchunk = (nslot - k + ncx - 1) / ncx;
    } else {
        *combo = x % ncombo;
        const int k = x / ncombo, nk = (8 - *combo + ncombo - 1) / ncombo;
        *chunk = slot * nk + k; *nchunk = nslot * nk;
    }
}

// bundles per task at most: 8; factored streams of groups of 16 / 8 utterances 4 / 2 (three descriptor words for each of the
// 16 / 32 rows of a bundle: the wave's slice of LDS)
constexpr int kBatEpiDefault = 2;   // (switch bat_epi; 0 / 1 / 2 / 4 -> S = 16 385 recursions 25.03 / 24.91 / 24.57 / 25.06 ms, S = 12 289 19.58 / 19.60 / 19.41 / 19.67: one box, round 6)
__host__ __device__ constexpr int stream_max_bundles(int UL, bool fac) { return !fac || UL >= 32 ? 8 : UL == 16 ? 4 : 2; }
struct StreamDirDev {
    const int4 *tasks;       // [ntasks] {first batch, batches, first bundle, bundles}
    const int2 *recs;        // [batches][AL][4] (+ 8 KB of padding)
    const int4 *meta;        // [bundles][AL] {state (-1: padding row), pair id, label, 0}; factored streams: [bundles][AL][3], below
    const int *rest;         // [nrest] rows (indices into BatchDev::frow / brow) that are not in the stream
    int ntasks, nrest;
};
// FACTORED streams (T o LM graphs, StreamDev::fac): the two states (g, blank) = "tail" and (g, token) = "main" of an LM history
// feed the same rows with the same weights, so -- as in the register-resident factored layout (FacDev) -- the forward vector
// gets an entry U[g] = a[tail] + a[main] behind the S states (NU of them) that serves both arcs with ONE record, and the tail
// row (two arcs, from the pair itself) is folded into the epilogue of the main row; backward, the two states have the same
// out-arcs but at most one each, so ONE row sums the common arcs and its epilogue adds each state's extra arc.  Half the
// records, i.e. half the gathers of the state vectors -- what a launch of crf_batch_frame_kernel is bound by.
// A row's descriptor is three int4:
//   {state 0 (-1: padding row), its pair, its label, 0}  {state 1 (-1: none), its pair, its label, 0}
//   forward:  {entry of U (S + k), tail weight bits, 0, 0}            state 0 = main, state 1 = tail
//   backward: {entry (pair id) of state 0's extra arc, its weight bits, the same for state 1}     (weight 0: no extra arc)
struct StreamDev { int ok, AL, want; StreamDirDev f, b; int fac, NU; const float *x_start; };   // want: tasks per direction the tables were cut for;
                                                                                             // x_start [S + NU]: a_0 incl. the U entries (fac)

// The denominator graph as the kernels see it (all pointers device memory).
// A "pair" is a distinct (destination state, label); pairs are numbered in forward-ELL row order.
struct GraphDev {
    int S;          // states
    int A;          // arcs
    int P;          // pairs
    int Pr;         // pair rows incl. padding (multiple of 64) -- row stride of the per-frame stores
    int Sr;         // state rows incl. padding (multiple of 64)
    int max_label;  // largest label on any arc
    EllDev fwd;     // rows = pairs,  arc idx = source state,       w = exp(weight)
    EllDev bwd;     // rows = states, arc idx = pair of that arc,   w = exp(weight)
    const int2 *pair_meta;   // [Pr] {destination state (-1 = padding row), label} of each pair
    const int4 *bwd_row_meta;// [Sr] per backward row {state (-1 = padding), #pairs into it, first pair, its label}
    const int *st_pair_off;  // [S+1] CSR: pairs whose destination is state s
    const int *st_pairs;     // [P]
    const float *start_lin;  // [S] exp(start_weight)
    const float *end_lin;    // [S] exp(end_weight)
    const int *perm;         // [P] pair ids sorted by (label, pair)
    const int *chunk_off;    // [NC+1] ranges of perm, each <= kChunk and within one label
    const int *lab_chunk_off;// [max_label+2] chunk range of each label
    int NC;
    ResDev res;
    FacDev fac;
    BatchDev bat;
};

struct ResBuildStats { int K = 0; int64_t slots_f = 0, slots_b = 0, conflicts_f = 0, conflicts_b = 0; };
struct FacBuildStats { int ok = 0; int64_t matched = 0, solo = 0, tail = 0, slots_f = 0, slots_b = 0, fused = 0, Gf = 0, Gb = 0; };

// host tables of the factored streams (fst_graph.cpp build_batch_factored)
struct FacRowH {
    int st0 = -1, pr0 = 0, lab0 = 0, st1 = -1, pr1 = 0, lab1 = 0;   // outputs: state, pair, label (st1 = -1: one output)
    int x0 = 0, x1 = 0;                                            // forward: x0 = U entry; backward: entries of the extra arcs
    int w0 = 0, w1 = 0;                                            // forward: w0 = tail weight bits; backward: the extra arcs' weight bits
    std::vector<int2> recs;                                        // {entry, weight bits}
};
struct FacBatchH {
    int ok = 0, NU = 0;
    std::vector<FacRowH> frows, brows;  // stream rows, most records first
    std::vector<int> frest, brest;      // rows (indices into BatchDev::frow / brow) left to the row-at-a-time path
    std::vector<float> x_start;         // [S + NU]
    int64_t recs_f = 0, recs_b = 0;     // records (statistics)
};
// host copy of the factored register-resident layout's tables (res_layout.cpp: kept for debug_emulate_factored, the CPU check)
struct FacHostCopy {
    std::vector<unsigned> farcs, barcs;
    std::vector<uint4> fwi, bwi;
    std::vector<int4> frow_meta, brow_meta;
    std::vector<float> x_start, x_end, z_end, brow_start, brow_end, bx_w, start_lin, end_lin;
    std::vector<int> z_lab, bx_idx, xlist;
    std::vector<int> gq, gb, gchunk, glab;   // grad pass lists
    int words = 0;                      // words per thread of the arc tables
};
// host copy of the generic register-resident layout's tables (debug_emulate_resident)
struct ResHostCopy {
    std::vector<unsigned> farcs, barcs;
    std::vector<uint4> fwi, bwi;
    std::vector<int> flab, blab, fcu, bcu, fown, bown, z_lab, gq, gb, gchunk, glab;
    std::vector<float> x_start, x_end, z_end, brow_start, brow_end;
};
struct HostGraph {
    int device = 0;
    int64_t S = 0, A = 0, P = 0;
    GraphDev dev{};
    std::vector<void *> allocs;  // device allocations to free
    // statistics for diagnostics / DESIGN.md
    int64_t fwd_padded_arcs = 0, bwd_padded_arcs = 0, fwd_conflicts = 0, bwd_conflicts = 0;
    int max_in_deg = 0, max_out_deg = 0;
    int regauged = 0;            // the weights were re-gauged (fst_graph.cpp: regauge_pushed)
    ResBuildStats res_stats;
    FacBuildStats fac_stats;
    // host copies of the utterance-minor tables (BatchDev) and the arc streams derived from them per lane-group count
    std::vector<int2> hb_farcs, hb_barcs;
    std::vector<int4> hb_frow, hb_brow;
    std::vector<int> hb_frow_d, hb_brow_s;
    std::vector<StreamDev *> streams;   // one per (AL, tasks wanted) used so far
    FacBatchH fb;                       // factored rows of the utterance-minor kernels (T o LM graphs), see StreamDev
    FacHostCopy fh;                     // host copy of dev.fac's tables
    // A SECOND factored layout for the two-utterance kernel (crf_fac_pair2_kernel on 512 threads x 30 chunks, 256 VGPRs per wave): built next
    // to a 1024-thread main layout, taken by calls whose batch is larger than the one-utterance kernel's staged schedule holds
    // (crf_kernels.hip use_facp).  It has its own rows, entries and grad-pass lists, so a call works with one of the two throughout.
    FacDev facp{};
    FacHostCopy fhp;
    int ncu = 256;                      // compute units of the graph's device (0 / host-only: 256)
    ResHostCopy rh;                     // host copy of dev.res's tables
    std::vector<int> h_src, h_dst, h_lab;   // the graph's arcs as compiled (graphs of up to 2^20 arcs: the emulations' plain reference)
    std::vector<float> h_w, h_start, h_end; // exp(weight), exp(start), exp(end)
    int res_rows_cu_f = 0, res_rows_cu_b = 0;  // max rows of one CU (LDS carve of the resident kernels)
};

void set_error(const std::string &msg);

// ---------------------------------------------------------------------------------------------
// Debug / experiment switches of tests and tools: set with crf_debug_set(key, value) (include/ctc_crf_hip.h), NEVER read from
// the environment -- the library's behaviour does not depend on the process environment.  One table: name, when it is read
// (G = when a graph is created, C = per loss call, X = when a (device, stream) context is created), what it does.
// ---------------------------------------------------------------------------------------------
#define CRF_OPTS(X)                                                                                                           \
    X(no_resident,      "G  no register-resident layout at all: the graph takes the utterance-minor or streaming kernels")   \
    X(no_factored,      "G  no factored layout: T o LM graphs take the generic register-resident layout")                    \
    X(fac_rcl,          "G  factored layout: row constants in the LDS table (fac_geom 1) for every graph")                   \
    X(fac_no_rcl,       "G  factored layout: row constants in registers even for graphs with long rows")                     \
    X(fac_k2,           "G  factored layout over TWO CUs per recursion (fac_geom 3) for every T o LM graph")                 \
    X(fac_no_k2,        "G  never two CUs per recursion: graphs of 120 k - 240 k arcs take the generic layout")              \
    X(fac_threads,      "G  512 / 768 / 1024: that geometry of the factored layout (default: 1024 threads first, then the 768-thread geometries, then 512)")                                            \
    X(fac_no_dup,       "G  factored layout: no second copy of the gathered entries")                                        \
    X(fac_bank_shift,   "G  factored layout: bank distance of the second copy (default 5)")                                  \
    X(res_mink,         "G  generic layout: at least this many CUs per recursion")                                           \
    X(res_epi,          "G  layout cost model: slice end in chunks (default 4)")                                             \
    X(res_piece,        "G  multi-lane rows: piece size in percent of a lane's chunks (default: the cost model's choice)")   \
    X(res_no_simd_order,"G  layout: no SIMD-aware placement of the waves' slice lists")                                      \
    X(res_simd0,        "G  factored layouts: handicap of SIMD 0 in the placement's cost model, in chunks (default 0)")                  \
    X(res_emis,         "G  factored layouts: what the emission staging costs a wave that holds emissions, in chunks (default 8; 0: waves treated alike)") \
    X(res_no_spread,    "G  small graphs: rows that fit a lane are not cut into pieces to give every wave a share")          \
    X(res_owner_first,  "G  multi-lane rows: the first lane of a group owns the outputs (no rotation)")                      \
    X(no_bank_arrange,  "G  no bank-aware arc order (edge colouring) in the register-resident layouts")                      \
    X(no_grad_arrange,  "G  grad pass: (Q, BP) pair lists in plain label order")                                             \
    X(no_regauge,       "G  weight-pushed graphs are not re-gauged")                                                         \
    X(regauge_minhash,  "G  re-gauging: find the two states of a history by min-hashing (what graphs of millions of arcs take)") \
    X(verbose,          "G  print layout statistics to stderr")                                                             \
    X(emu_drop_list,    "-  crf_debug_fac_emulate: drop the two-CU fetch list (negative control of the emulation)")          \
    X(emu_facp,         "-  crf_debug_fac_emulate: the two-utterance kernel's layout (HostGraph::facp) instead of the main one")  \
    X(emu_verbose,      "-  crf_debug_fac_emulate: per-frame masses to stderr")                                              \
    X(bat_no_fac,       "GC utterance-minor kernels: plain arc streams instead of the factored ones")                        \
    X(bat_task,         "C  utterance-minor kernels: steps per task")                                                        \
    X(bat_ul,           "C  utterance-minor kernels: utterances per group (8, 16, 32, 64)")                                  \
    X(bat_fill,         "C  utterance-minor kernels: percent of the device's workgroup slots a launch takes (default 70)")   \
    X(bat_epi,          "GC utterance-minor kernels: what a bundle's row epilogues count for when the arc streams are cut into tasks, in batches of 4 steps") \
    X(bat_persist,      "C  utterance-minor kernels: 1 = all frames in ONE persistent launch with a grid barrier per frame (default when the grid is co-resident), 0 = one launch per frame") \
    X(force_batch,      "C  utterance-minor kernels for every graph")                                                        \
    X(no_batch,         "C  streaming kernels instead of the utterance-minor ones")                                          \
    X(robust,           "C  0: never run the robust fallbacks, 1: every utterance takes them (denominator and numerator)")  \
    X(robust_ctc,       "C  1: every utterance's numerator takes the log-domain fallback")                                  \
    X(ctc_tilt,         "C  numerator chains: strength of the tilt in percent (default 100, 0 = plain chains)")                \
    X(no_fast_grad,     "C  generic grad kernel instead of the streaming grad kernels")                                      \
    X(no_overlap,       "C  no staged grad pass beside the recursions")                                                      \
    X(segments,         "C  staged schedule by relaunching the recursions per stage (events) instead of stream-level waits") \
    X(stage_fill,       "C  staged schedule for den grids of up to this percent of the CUs (default 75)")                     \
    X(stages,           "C  number of grad stages")                                                                          \
    X(piece,            "C  iterations per grad stage")                                                                      \
    X(first_shift,      "C  first grad stage this many 16-frame blocks later (shortens the LAST stage by as much)")           \
    X(gd_full_grid,     "C  stage launches of the grad pass as full grids")                                                  \
    X(gd_stage_launches,"C  one grad launch per stage behind stream-level waits (round 4) instead of ONE launch whose workgroups wait") \
    X(no_fin_fold,      "C  crf_finalize_kernel as a launch of its own (round 4) instead of in the last fallback launch")                \
    X(gd_sub,           "C  frames per grad workgroup in the short last stages of the one-launch grad pass (default 16 = whole blocks; 8 / 4 / 2 split them: measured not better)") \
    X(taper,            "C  the last grad stages get shorter: pieces of piece/2, piece/4, ... down to this many iterations (0 = equal pieces)") \
    X(ctc_after,        "C  0 / 1: numerator chains beside / after the denominator recursions")                              \
    X(serial_chains,    "C  everything on the caller's stream")                                                              \
    X(no_side_stream,   "X  no side stream for this context")                                                                \
    X(trust_side,       "X  take a side stream without probing that it runs beside the caller's (profiler counter passes)")  \
    X(side_kind,        "X  side stream candidates of one kind only: 1 plain, 2 high priority, 3 low priority, 4 CU-masked")  \
    X(grad_par3,        "C  staged schedule: 1 = the den half of the grad pass on the third stream beside the numerator half (atomic adds into zeroed rows; measured slower, default 0)") \
    X(aux_stream,       "C  numerator fallback chains on the third stream: 1 always, 0 never (default: when a recent call needed them)")  \
    X(no_aux_stream,    "X  no third stream for this context (numerator fallback chains in front of the grad stages)")      \
    X(no_facp,          "G  no second (512-thread) factored layout for the two-utterance kernel")                            \
    X(fac_pair2,        "C  factored recursions with TWO utterances per workgroup: 1 = for any batch, 0 = never (default: batches above CUs / 2)")

enum Opt : int {
#define CRF_OPT_ENUM(name, doc) kOpt_##name,
    CRF_OPTS(CRF_OPT_ENUM)
#undef CRF_OPT_ENUM
    kOptCount
};
int opt(Opt k, int dflt = 0);                       // the value set with crf_debug_set, `dflt` while unset
inline bool opt_on(Opt k) { return opt(k, 0) != 0; }
int opt_set(const char *key, int value, bool unset);   // CRF_OK or CRF_ERR_ARG (unknown key)
const char *opt_list();                             // "name: doc\n" of every switch

// Builds pairs, both ELL tables and the grad-pass chunk tables, and uploads them to `device`.
int compile_graph(int64_t S, int64_t A, const int32_t *src, const int32_t *dst, const int32_t *lab,
                  const float *w, const float *start_w, const float *end_w, int device, HostGraph **out);
// Builds the resident layout (smallest K in {1,2,4} that fits the register budget) into h->dev.res and
// uploads it (unless h->device < 0).  Returns CRF_OK with res.K == 0 when the graph does not fit.
int build_resident(HostGraph *h, int S, int P, const std::vector<int> &pair_dst, const std::vector<int> &pair_lab,
                   const std::vector<std::vector<std::pair<int, float>>> &in_arcs_of_pair,
                   const std::vector<std::vector<std::pair<int, float>>> &out_arcs_of_state,
                   const std::vector<float> &start_lin, const std::vector<float> &end_lin,
                   const std::vector<int> &label_sorted_pairs);
// Builds the factored layout into h->dev.fac (fac.ok = 0 when the graph has no such structure or does not fit).
int build_factored(HostGraph *h, int S, int P, const std::vector<int> &pair_dst, const std::vector<int> &pair_lab,
                   const std::vector<std::vector<std::pair<int, float>>> &in_arcs_of_pair,
                   const std::vector<std::vector<std::pair<int, float>>> &out_arcs_of_state,
                   const std::vector<float> &start_lin, const std::vector<float> &end_lin);
// Arc streams for UL (8, 16, 32 or 64) utterances per group cut into about `want` tasks per direction, built and uploaded
// on first use (thread-safe; a graph keeps every variant it has been asked for).
int debug_emulate_factored(const HostGraph *h, int T, unsigned seed, double *out3, int which = 0);   // which = 1: the two-utterance kernel's layout (HostGraph::facp)
int debug_emulate_resident(const HostGraph *h, int T, unsigned seed, double *out3);
int ensure_stream_tables(HostGraph *h, int UL, int want, const StreamDev **out);
bool stream_fac(const HostGraph *h, int UL);   // factored streams for groups of UL utterances?
// Host-side construction + self-check of the arc streams (tests; works on host-only graphs).
int debug_check_streams(const HostGraph *h, int UL, int want, int64_t *out4);
// Host-side check of bat_decode for one (grid, #combos): 0 = every (combo, chunk) exactly once.
int debug_check_decode(int nslot, int ncombo);
int read_fst_file(const char *path, int64_t *S, std::vector<int32_t> *src, std::vector<int32_t> *dst,
                  std::vector<int32_t> *lab, std::vector<float> *w, std::vector<float> *start_w,
                  std::vector<float> *end_w);

}  // namespace crf

struct crf_graph {
    crf::HostGraph *h;
};
