"""Lab build of libupamd.so whose fused small-model kernel time-stamps EVERY barrier (line number + 100 MHz clock) of the first graph
of workgroup 0: a patched COPY of csrc/ under tools/lab_trace/csrc (the product sources are untouched).

    python tools/lab_trace/build.py          # here (hipcc cross-compiles); the .so travels with the gpurun snapshot
    python tools/lab_trace/trace_tiny.py     # on the GPU box: prints the phase timeline
"""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, 'drl-urban-planning_amd', 'csrc')
DST = os.path.join(ROOT, 'tools', 'lab_trace', 'csrc')
if os.path.isdir(DST):
    shutil.rmtree(DST)
os.makedirs(DST)
for f in os.listdir(SRC):
    if f.endswith(('.hip', '.h', '.cpp', '.c')) or f == 'Makefile':
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, f))

p = os.path.join(DST, 'tiny_body.h')
s = open(p).read()
old = '#define T_SYNC() __syncthreads()\n'
assert s.count(old) == 1
s = s.replace(old, '''#define T_SYNC()                                                                                   \\
    do {                                                                                           \\
        __syncthreads();                                                                           \\
        if (upamd_trace_buf && blockIdx.x == 0 && threadIdx.x == 0 && upamd_trace_n < 1000) {      \\
            upamd_trace_buf[2 * upamd_trace_n] = wall_clock64();                                   \\
            upamd_trace_buf[2 * upamd_trace_n + 1] = __LINE__;                                     \\
            ++upamd_trace_n;                                                                       \\
        }                                                                                          \\
    } while (0)
''')
assert s.count('#define T_FOR(i, N) for (int i = t_tid();') == 1
s = s.replace('#define T_FOR(i, N) for (int i = t_tid();', '''#define T_TRACE(tag)                                                                               \\
    do {                                                                                           \\
        if (upamd_trace_buf && blockIdx.x == 0 && threadIdx.x == 0 && upamd_trace_n < 1000) {      \\
            upamd_trace_buf[2 * upamd_trace_n] = wall_clock64();                                   \\
            upamd_trace_buf[2 * upamd_trace_n + 1] = 1000000 + (tag);                               \\
            ++upamd_trace_n;                                                                       \\
        }                                                                                          \\
    } while (0)
#define T_FOR(i, N) for (int i = t_tid();''', 1)
anchors = [('    const int t = A.idx[b];\n', '    T_TRACE(1);\n', 'before'),
           ('    const int64_t node_off = m[9];\n', '    T_TRACE(2000 + (n & 1));\n', 'after'),
           ('    T_MARK(0);\n', '    T_TRACE(3);\n', 'before'),
           ('    T_FOR(i, MAXL + 1) bad[i] = 0;\n', '    T_TRACE(4);\n', 'before'),
           ('        if (tid < XPAD) r_cur = A.cur[(int64_t)t * XPAD + tid];\n        __builtin_amdgcn_sched_barrier(0);\n',
            '        if (tid < XPAD) r_cur = A.cur[(int64_t)t * XPAD + tid];\n        T_TRACE(31);\n        __builtin_amdgcn_sched_barrier(0);\n        T_TRACE(32000 + ((r_rp[0] + (int)r_nb[0] + (int)r_x[0].x + (int)r_we[0] + (int)r_u + (int)r_cur) & 1));\n', 'replace'),
           ('        if (tid < XPAD) cur[tid] = r_cur;\n', '        T_TRACE(33);\n', 'after')]
for anchor, ins, how in anchors:
    assert s.count(anchor) == 1, anchor
    if how == 'before':
        s = s.replace(anchor, ins + anchor)
    elif how == 'after':
        s = s.replace(anchor, anchor + ins)
    else:
        s = s.replace(anchor, ins)
s = s.replace('namespace upamd_tiny {\n', '__device__ long long *upamd_trace_buf = nullptr;\n__device__ int upamd_trace_n = 0;\n\nnamespace upamd_tiny {\n', 1)
open(p, 'w').write(s)

p = os.path.join(DST, 'tiny.hip')
s = open(p).read()
old = 'void set_tiny_prof(void *buf) { g_tiny_prof = static_cast<long long *>(buf); }'
assert s.count(old) == 1
s = s.replace(old, '''void set_tiny_prof(void *buf) {
    g_tiny_prof = static_cast<long long *>(buf);
    long long *tb = buf ? static_cast<long long *>(buf) + 64 : nullptr;      // barrier trace behind the 64 section marks
    int zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(upamd_trace_buf), &tb, sizeof(tb));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(upamd_trace_n), &zero, sizeof(zero));
}''')
open(p, 'w').write(s)
# the include path of the copy's Makefile is relative to csrc/: ../../include -> three levels up from tools/lab_trace/csrc
p = os.path.join(DST, 'Makefile')
s = open(p).read().replace('-I../../include', '-I../../../include').replace('../../include/upamd.h', '../../../include/upamd.h')
open(p, 'w').write(s)
for f in os.listdir(DST):      # sources that include the public header by relative path
    q = os.path.join(DST, f)
    if f.endswith(('.h', '.hip', '.cpp', '.c')):
        t = open(q).read()
        if '"../../include/upamd.h"' in t:
            open(q, 'w').write(t.replace('"../../include/upamd.h"', '"../../../include/upamd.h"'))
res = subprocess.run(['make', '-j8', '-C', DST], capture_output=True, text=True)
print(res.stdout[-600:], res.stderr[-2000:])
print('built', os.path.join(DST, 'libupamd.so'), os.path.exists(os.path.join(DST, 'libupamd.so')))
