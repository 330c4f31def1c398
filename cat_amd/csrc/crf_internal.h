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
};

struct HostGraph {
    int device = 0;
    int64_t S = 0, A = 0, P = 0;
    GraphDev dev{};
    std::vector<void *> allocs;  // device allocations to free
    // statistics for diagnostics / DESIGN.md
    int64_t fwd_padded_arcs = 0, bwd_padded_arcs = 0, fwd_conflicts = 0, bwd_conflicts = 0;
    int max_in_deg = 0, max_out_deg = 0;
};

void set_error(const std::string &msg);

// Builds pairs, both ELL tables and the grad-pass chunk tables, and uploads them to `device`.
int compile_graph(int64_t S, int64_t A, const int32_t *src, const int32_t *dst, const int32_t *lab,
                  const float *w, const float *start_w, const float *end_w, int device, HostGraph **out);
int read_fst_file(const char *path, int64_t *S, std::vector<int32_t> *src, std::vector<int32_t> *dst,
                  std::vector<int32_t> *lab, std::vector<float> *w, std::vector<float> *start_w,
                  std::vector<float> *end_w);

}  // namespace crf

struct crf_graph {
    crf::HostGraph *h;
};
