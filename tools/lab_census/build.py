"""Lab build of libupamd.so whose GEMM and message-passing kernels take a CENSUS: every workgroup records which kernel it belongs to,
the CU it ran on (HW_REG_HW_ID + HW_REG_XCC_ID), when it started and ended (100 MHz wall clock) and the shader-clock cycles it saw
(s_memtime) -- the evidence for "were the MFMA-bound and the VALU-bound kernels ever RESIDENT ON THE SAME CU at the same time, and
what did that do to each of them" (round-6 review item 1).  A patched COPY of csrc/ under tools/lab_census/csrc (the product
sources are untouched; the copy is not tracked).

    python tools/lab_census/build.py         # here (hipcc cross-compiles); the .so travels with the gpurun snapshot
    UPAMD_LIB_PATH=tools/lab_census/csrc/libupamd.so python tools/lab_census/run.py ...    # on the GPU box
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, 'drl-urban-planning_amd', 'csrc')
# `python tools/lab_census/build.py nt128` -> tools/lab_census/csrc_nt128: the same census build with the LDS-DMA NT GEMM compiled for
# <= 128 VGPRs (__launch_bounds__(256, 4): 126 VGPRs, no spills, instead of 82 + 64 AGPRs = 152 allocated), so that TWO of its
# workgroups fit the registers one resident message-passing workgroup (4 waves x 64 VGPRs per SIMD) leaves free
NT128 = len(sys.argv) > 1 and sys.argv[1] == 'nt128'
DST = os.path.join(ROOT, 'tools', 'lab_census', 'csrc_nt128' if NT128 else 'csrc')
if os.path.isdir(DST):
    shutil.rmtree(DST)
os.makedirs(DST)
for f in os.listdir(SRC):
    if f.endswith(('.hip', '.h', '.cpp', '.c')) or f == 'Makefile':
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, f))


def patch(fname, edits):
    p = os.path.join(DST, fname)
    s = open(p).read()
    for old, new, count in edits:
        assert s.count(old) == count, (fname, old, s.count(old))
        s = s.replace(old, new)
    open(p, 'w').write(s)


CENSUS_H = r'''
// ---- lab census (tools/lab_census/build.py) -------------------------------------------------------------------------------
// buf[0] = next free record (atomic), buf[1] = capacity, records of 8 int64 from buf + 8:
//   [0] kernel id (1 gemm_nt_dma2, 2 gemm_tn_mfma, 3 edge_fwd, 4 edge_bwd)   [1] XCC_ID << 32 | HW_ID
//   [2] start, [4] end: 100 MHz wall clock     [3] start, [5] end: shader clock (s_memtime)     [6] blockIdx.x   [7] gridDim.x
static __device__ long long *upamd_census_buf = nullptr;      // one copy per translation unit (no RDC): set by census_set_tu below
static inline void census_set_tu(void *p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(upamd_census_buf), &p, sizeof(p)); }
__device__ __forceinline__ int census_begin(int kid) {
    long long *buf = upamd_census_buf;
    int slot = -1;
    if (buf && threadIdx.x == 0) {
        slot = (int)atomicAdd(reinterpret_cast<unsigned long long *>(buf), 1ull);
        if (slot < (int)buf[1]) {
            long long *r = buf + 8 + (long long)slot * 8;
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
            const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
            r[0] = kid;
            r[1] = ((long long)xcc << 32) | hw;
            r[2] = wall_clock64();
            r[3] = __builtin_readcyclecounter();
            r[6] = blockIdx.x;
            r[7] = gridDim.x;
        } else {
            slot = -1;
        }
    }
    return slot;
}
__device__ __forceinline__ void census_end(int slot) {
    if (slot >= 0) {
        long long *r = upamd_census_buf + 8 + (long long)slot * 8;
        r[4] = wall_clock64();
        r[5] = __builtin_readcyclecounter();
    }
}
void set_census_gemm(void *buf);
void set_census_edge(void *buf);
'''

patch('kernels.h', [('namespace upamd {\n', 'namespace upamd {\n' + CENSUS_H, 1)])

patch('gemm.hip', [
    # the LDS-DMA NT kernel (slot kept in a register: this kernel has VGPRs to spare)
    ('    if (mt >= MT) return;\n    first_wave_stagger(stagger_mode, stagger_cycles);\n',
     '    if (mt >= MT) return;\n    const int census_slot = census_begin(1);\n    first_wave_stagger(stagger_mode, stagger_cycles);\n', 1),
    ("                *reinterpret_cast<float4 *>(C + o) = make_float4(v[0] * alpha, v[1] * alpha, v[2] * alpha, v[3] * alpha);\n            }\n    }\n}\n\n// which LDS-DMA configuration",
     "                *reinterpret_cast<float4 *>(C + o) = make_float4(v[0] * alpha, v[1] * alpha, v[2] * alpha, v[3] * alpha);\n            }\n    }\n    census_end(census_slot);\n}\n\n// which LDS-DMA configuration", 1),
    # the TN (weight-gradient) kernel
    ('    if (split >= S) return;\n    const int it = tile / JT, jt = tile % JT;\n',
     '    if (split >= S) return;\n    const int census_slot = census_begin(2);\n    const int it = tile / JT, jt = tile % JT;\n', 1),
    ("                slab[(int64_t)gi * J + gj] = acc[i][j][r];\n            }\n        }\n}\n",
     "                slab[(int64_t)gi * J + gj] = acc[i][j][r];\n            }\n        }\n    census_end(census_slot);\n}\n", 1),
    ('void set_gemm_lds_pad(int bytes) { g_lds_pad = bytes; }\n',
     'void set_gemm_lds_pad(int bytes) { g_lds_pad = bytes; }\nvoid set_census_gemm(void *buf) { census_set_tu(buf); }\n', 1),
])

if NT128:
    patch('gemm.hip', [('__global__ __launch_bounds__(64 * WM * WN, (TM * TN > 4 ? 2 : 1)) void gemm_nt_dma2_kernel',
                        '__global__ __launch_bounds__(64 * WM * WN, 4) void gemm_nt_dma2_kernel', 1)])

# message passing: the slot lives in 4 bytes of static LDS (these kernels are built for exactly 64 VGPRs)
patch('edge.hip', [
    ('    const int64_t o = m[14], M = mb.M;\n    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;\n    const int ca = 2 * (lane & 7), g = lane >> 3;      // this lane\'s two columns (ca, ca+1); node slot within the wave\n    const EdgeLds L = carve(smem, n, e, STAGE, false, HLDS);\n',
     '    const int64_t o = m[14], M = mb.M;\n    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;\n    __shared__ int census_slot;\n    if (tid == 0) census_slot = census_begin(3);\n    const int ca = 2 * (lane & 7), g = lane >> 3;      // this lane\'s two columns (ca, ca+1); node slot within the wave\n    const EdgeLds L = carve(smem, n, e, STAGE, false, HLDS);\n', 1),
    ('            else hbarV[(int64_t)b * D + p * 16 + cc] = tot / (float)m[6];\n        }\n    }\n}\n',
     '            else hbarV[(int64_t)b * D + p * 16 + cc] = tot / (float)m[6];\n        }\n    }\n    if (tid == 0) census_end(census_slot);\n}\n', 1),
    ('    const int64_t o = m[14], M = mb.M;\n    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;\n    const int ca = 2 * (lane & 7), g = lane >> 3;      // this lane\'s two columns (ca, ca+1); node slot within the wave\n    const EdgeLds L = carve(smem, n, e, STAGE, true, true, !NBG);\n',
     '    const int64_t o = m[14], M = mb.M;\n    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;\n    __shared__ int census_slot;\n    if (tid == 0) census_slot = census_begin(4);\n    const int ca = 2 * (lane & 7), g = lane >> 3;      // this lane\'s two columns (ca, ca+1); node slot within the wave\n    const EdgeLds L = carve(smem, n, e, STAGE, true, true, !NBG);\n', 1),
    ('        dbias_part[(int64_t)b * (NP * 32) + p * 32 + pq_pos(which, cc)] = tot;\n    }\n}\n',
     '        dbias_part[(int64_t)b * (NP * 32) + p * 32 + pq_pos(which, cc)] = tot;\n    }\n    if (tid == 0) census_end(census_slot);\n}\n', 1),
    ('void set_bwd_nb_global(int on) { g_bwd_nb_global = on ? 1 : 0; }\n',
     'void set_bwd_nb_global(int on) { g_bwd_nb_global = on ? 1 : 0; }\nvoid set_census_edge(void *buf) { census_set_tu(buf); }\n', 1),
    # lab knob UPAMD_LAB_EDGE_LDS=<bytes>: every message-passing launch asks for at least this much dynamic LDS, i.e. the number of
    # walk workgroups a CU can hold is set by hand (96 KB -> ONE per CU, with 64 KB left for two 32 KB GEMM workgroups)
    ('    auto go = [&](bool stage, int64_t lds, int aux_cap, int fit, bool hlds = true) -> int {\n',
     '    auto go = [&](bool stage, int64_t lds, int aux_cap, int fit, bool hlds = true) -> int {\n        { const char *ev_ = getenv("UPAMD_LAB_EDGE_LDS"); if (ev_ && atoll(ev_) > lds) lds = atoll(ev_); }\n', 1),
    ('    auto go = [&](bool stage, int64_t lds, int aux_cap, int fit, bool nbg = false) -> int {\n',
     '    auto go = [&](bool stage, int64_t lds, int aux_cap, int fit, bool nbg = false) -> int {\n        { const char *ev_ = getenv("UPAMD_LAB_EDGE_LDS"); if (ev_ && atoll(ev_) > lds) lds = atoll(ev_); }\n', 1),
    ('#include <type_traits>\n', '#include <cstdlib>\n#include <type_traits>\n', 1),
])

# the census buffer comes in through the (otherwise unused at D = 256) lab hook upamd_tiny_profile
patch('api_rl.hip', [('    set_tiny_prof(buf_dev);\n    return UPAMD_OK;\n', '    set_census_gemm(buf_dev);\n    set_census_edge(buf_dev);\n    return UPAMD_OK;\n', 1)])

p = os.path.join(DST, 'Makefile')
s = open(p).read().replace('-I../../include', '-I../../../include').replace('../../include/upamd.h', '../../../include/upamd.h')
open(p, 'w').write(s)
for f in os.listdir(DST):
    q = os.path.join(DST, f)
    if f.endswith(('.h', '.hip', '.cpp', '.c')):
        t = open(q).read()
        if '"../../include/upamd.h"' in t:
            open(q, 'w').write(t.replace('"../../include/upamd.h"', '"../../../include/upamd.h"'))
res = subprocess.run(['make', '-j8', '-C', DST], capture_output=True, text=True)
print(res.stdout[-400:], res.stderr[-3000:])
print('built', os.path.join(DST, 'libupamd.so'), os.path.exists(os.path.join(DST, 'libupamd.so')))
