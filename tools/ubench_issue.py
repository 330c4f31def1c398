#!/usr/bin/env python
"""tools/ubench_issue.py -- writes tools/ubench_issue.hip: issue-rate micro-benchmarks for the frame loop of the factored
recursions (crf_kernels.hip fac_chain_body), whole loops in inline assembly so that nothing is rescheduled:

  valu_sdwa     32 independent v_add_u32_sdwa (the address unpack of a gather)
  valu_pkfma    16 independent v_pk_fma_f32
  valu_pkchain  16 v_pk_fma_f32 chained on one accumulator (what a row's sum is)
  lds_read      32 conflict-free ds_read_b32, one s_waitcnt
  batch         the loop's batch of 2 chunks: 8 unpack + 8 gathers + partial waits + 4 chained packed FMAs
  batch_pipe    the same, gathers of batch k+1 issued before the FMAs of batch k
  batch4        a batch of 4 chunks (16 gathers in flight)
  barrier       s_waitcnt lgkmcnt(0) + s_barrier only
  branch        8 taken forward branches over 64 instructions each
  frame         8 batches + wave maximum (3 DPP) + ds_max + barrier + ds_read_b128 + readfirstlane: a frame without row ends

    python tools/ubench_issue.py && hipcc --offload-arch=gfx950 -O2 tools/ubench_issue.hip -o tools/ubench_issue.bin
    gpurun -- tools/ubench_issue.bin
Prints shader cycles per loop iteration (s_memtime, slowest wave) for 1, 4, 8 and 16 waves on one CU."""
import os

SDWA = "v_add_u32_sdwa v{d}, s20, v{s} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_{h}"


def batch(nch, base=40, wait=True):
    n = 4 * nch
    a = [SDWA.format(d=base + i, s=10 + (i // 2) % 8, h=i % 2) for i in range(n)]
    r = [f"ds_read_b32 v{base + i}, v{base + i}" for i in range(n)]
    f = []
    for c in range(2 * nch):
        left = n - 2 * (c + 1)
        f.append(f"s_waitcnt lgkmcnt({left})")
        f.append(f"v_pk_fma_f32 v[30:31], v[{base + 2 * c}:{base + 2 * c + 1}], v[{20 + 2 * (c % 4)}:{21 + 2 * (c % 4)}], v[30:31]")
    return a, r, f


def body(mode):
    if mode == "valu_sdwa":
        return [SDWA.format(d=40 + i, s=10 + i % 8, h=i % 2) for i in range(32)], 32
    if mode == "valu_pkfma":
        return [f"v_pk_fma_f32 v[{40 + 2 * i}:{41 + 2 * i}], v[20:21], v[22:23], v[{40 + 2 * i}:{41 + 2 * i}]" for i in range(16)], 16
    if mode == "valu_pkchain":
        return ["v_pk_fma_f32 v[30:31], v[20:21], v[22:23], v[30:31]" for i in range(16)], 16
    if mode == "lds_read":
        return [f"ds_read_b32 v{40 + i}, v{32 + i % 8}" for i in range(32)] + ["s_waitcnt lgkmcnt(0)"], 32
    if mode == "batch":
        a, r, f = batch(2)
        return a + r + f, 1
    if mode == "batch4":
        a, r, f = batch(4)
        return a + r + f, 1
    if mode == "batch_pipe":   # two register sets; the next batch's gathers are in flight while this one's FMAs run
        a0, r0, f0 = batch(2, 40)
        a1, r1, f1 = batch(2, 48)
        fix = lambda f: [x.replace("lgkmcnt(6)", "lgkmcnt(14)").replace("lgkmcnt(4)", "lgkmcnt(12)").replace("lgkmcnt(2)", "lgkmcnt(10)").replace("lgkmcnt(0)", "lgkmcnt(8)") for x in f]
        return a1 + r1 + fix(f0) + a0 + r0 + fix(f1), 2
    if mode == "barrier":
        return ["s_waitcnt lgkmcnt(0)", "s_barrier"], 1
    if mode == "branch":
        out = []
        for k in range(8):
            out += ["s_cmp_lg_u32 s21, 0", f"s_cbranch_scc0 LBL_{mode}_%={k}".replace("%=", "%=_")]
            out += [f"v_add_u32 v{40 + i % 32}, v{40 + i % 32}, v10" for i in range(64)]
            out += [f"LBL_{mode}_%=_{k}:"]
        return out, 8
    if mode in ("frame_b", "frame_bm", "frame_rot"):   # 8 batches + barrier (+ wave maximum into LDS); _rot: the waves start at different batches
        out = []
        for k in range(8):
            a, r, f = batch(2)
            out += a + r + f
        if mode == "frame_bm":
            out += ["v_max_i32_dpp v32, v30, v30 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1", "s_nop 1",
                    "v_max_i32_dpp v32, v32, v32 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1", "s_nop 1",
                    "v_max_i32_dpp v32, v32, v32 row_ror:2 row_mask:0xf bank_mask:0xf bound_ctrl:1", "s_nop 1", "s_mov_b64 s[28:29], exec", "s_mov_b32 exec_lo, 0x00010001", "s_mov_b32 exec_hi, 0x00010001", "ds_max_f32 v33, v32", "s_mov_b64 exec, s[28:29]"]
        return out + ["s_waitcnt lgkmcnt(0)", "s_barrier"], 1
    if mode == "frame8x":   # the same 8 batches as a LOOP of one batch (a taken branch per batch), then the barrier
        a, r, f = batch(2)
        return ["s_mov_b32 s27, 8", "INNER_%=:"] + a + r + f + ["s_sub_u32 s27, s27, 1", "s_cmp_lg_u32 s27, 0", "s_cbranch_scc1 INNER_%=", "s_waitcnt lgkmcnt(0)", "s_barrier"], 1
    if mode == "frame":
        out = []
        for k in range(8):
            a, r, f = batch(2)
            out += a + r + f
        out += ["v_max_i32_dpp v32, v30, v30 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1", "s_nop 1",
                "v_max_i32_dpp v32, v32, v32 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1", "s_nop 1",
                "v_max_i32_dpp v32, v32, v32 row_ror:2 row_mask:0xf bank_mask:0xf bound_ctrl:1", "s_nop 1",
                "s_mov_b64 s[28:29], exec", "s_mov_b32 exec_lo, 0x00010001", "s_mov_b32 exec_hi, 0x00010001", "ds_max_f32 v33, v32", "s_mov_b64 exec, s[28:29]", "s_waitcnt lgkmcnt(0)", "s_barrier",
                "ds_read_b128 v[36:39], v34", "s_waitcnt lgkmcnt(0)", "v_max3_i32 v36, v36, v37, v38", "s_nop 0", "v_readfirstlane_b32 s22, v36"]
        return out, 1
    raise KeyError(mode)


MODES = ["valu_sdwa", "valu_pkfma", "valu_pkchain", "lds_read", "batch", "batch_pipe", "batch4", "barrier", "branch", "frame_b", "frame_bm", "frame8x", "frame"]
CLOB = ", ".join(f'"v{i}"' for i in range(10, 80)) + ', "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "scc", "vcc", "memory"'

src = ["// GENERATED by tools/ubench_issue.py -- do not edit", "#include <hip/hip_runtime.h>", "#include <cstdio>", "#include <vector>", "#include <algorithm>",
       "extern __shared__ float lds[];"]
for m in MODES:
    lines, _ = body(m)
    asm = ["s_mov_b32 s20, 0", "s_mov_b32 s21, 0", "s_mov_b32 s23, %2"]
    # gather offsets: lane * 4 in both halves (+ 256 bytes in the high one): conflict-free; weights 1.0; LDS word for ds_max at 60000 + 16 * (lane / 16)
    asm += [f"v_mov_b32 v{10 + i}, %1" for i in range(8)] + [f"v_mov_b32 v{32 + i}, %3" for i in range(8)]
    asm += [f"v_mov_b32 v{20 + i}, 1.0" for i in range(12)] + ["v_mov_b32 v33, %4", "v_mov_b32 v34, %4"]
    asm += ["s_barrier", "s_memtime s[24:25]", "s_waitcnt lgkmcnt(0)", f"LOOP_{m}_%=:"] + lines
    asm += ["s_sub_u32 s23, s23, 1", "s_cmp_lg_u32 s23, 0", f"s_cbranch_scc1 LOOP_{m}_%=", "s_waitcnt lgkmcnt(0)", "s_memtime s[26:27]", "s_waitcnt lgkmcnt(0)",
            "s_sub_u32 s24, s26, s24", "v_mov_b32 %0, s24"]
    text = "\\n\\t".join(asm)
    src += [f"__global__ void k_{m}(unsigned *out, int iters) {{", "    unsigned cyc;",
            "    const unsigned lane = threadIdx.x & 63, off = (lane * 4u) | ((lane * 4u + 256u) << 16), a2 = lane * 4u, wmax = 60000u + 16u * (lane >> 4);",
            "    if (threadIdx.x < 16) lds[15000 + threadIdx.x] = 0.f;",
            f'    asm volatile("{text}" : "=v"(cyc) : "v"(off), "s"(iters), "v"(a2), "v"(wmax) : {CLOB});',
            "    if (lane == 0) out[threadIdx.x >> 6] = cyc;", "}"]
src += ["int main() {", "    unsigned *d; hipMalloc(&d, 64 * sizeof(unsigned));", "    const int iters = 2000;",
        '    printf("%-14s %10s %10s %10s %10s   (shader cycles per iteration, slowest wave; units per iteration in brackets)\\n", "mode", "1 wave", "4 waves", "8 waves", "16 waves");']
for m in MODES:
    _, units = body(m)
    src += [f"    hipFuncSetAttribute((const void *)k_{m}, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);",
            f'    printf("%-14s", "{m} [{units}]");', "    for (int nt : {64, 256, 512, 1024}) {",
            f"        hipLaunchKernelGGL(k_{m}, dim3(1), dim3(nt), 65536, 0, d, iters);",
            "        std::vector<unsigned> h(16); hipMemcpy(h.data(), d, 16 * sizeof(unsigned), hipMemcpyDeviceToHost);",
            "        unsigned mx = 0; for (int w = 0; w < nt / 64; ++w) mx = std::max(mx, h[w]);",
            '        printf(" %10.1f", (double)mx / iters);', "    }", '    printf("\\n");']
src += ["    return 0;", "}"]
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench_issue.hip"), "w").write("\n".join(src) + "\n")
