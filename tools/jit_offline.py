#!/usr/bin/env python
"""
Compile a tree-specialised kernel OFFLINE (no GPU): emit its source with rt_jit_source,
compile it through hiprtc with exactly the options rt_jit_get uses (csrc/jit.hip), write
the code object and its disassembly, and print the resource usage the rejection rule of
rt_jit_get looks at (scratch bytes) next to VGPR / AGPR counts and spill counts.

    python tools/jit_offline.py --states 32 --leaves 32 --tiles 4 --out /tmp/k32
"""
import argparse
import ctypes
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def emit_source(n, nleaves, tiles, prefetch, random_nodes=0, seed=0):
    os.environ['RAOTEH_JIT_TILES'] = str(tiles)
    from raoteh_amd import _lib, synth
    from raoteh_amd._tree import TreeArrays
    if random_nodes:
        T, root, leaves = synth.random_tree(random_nodes, seed=seed)
    else:
        T, root, leaves = synth.balanced_tree(nleaves)
    ta = TreeArrays(T, root)
    obs = np.array([ta.node_to_index[v] for v in leaves], dtype=np.int64)
    buf = ctypes.create_string_buffer(64 << 20)
    p64 = ctypes.POINTER(ctypes.c_int64)
    _lib.check(_lib.lib().rt_jit_source(
        ta.nnodes, ta.indices.ctypes.data_as(p64), ta.indptr.ctypes.data_as(p64), n,
        len(obs), obs.ctypes.data_as(p64), prefetch, buf, len(buf)))
    return buf.value


def hiprtc_compile(src, mfma=True, extra=()):
    rtc = ctypes.CDLL('libhiprtc.so')
    prog = ctypes.c_void_p()
    assert rtc.hiprtcCreateProgram(ctypes.byref(prog), src, b'rt_jit_prune.hip', 0, None,
                                   None) == 0
    opts = [b'--offload-arch=gfx950', b'-O3', b'-std=c++17']
    if mfma:
        opts += [b'-mllvm', b'-amdgpu-mfma-vgpr-form=1']
    opts += [o.encode() for o in extra]
    arr = (ctypes.c_char_p * len(opts))(*opts)
    rc = rtc.hiprtcCompileProgram(prog, len(opts), arr)
    if rc != 0:
        sz = ctypes.c_size_t()
        rtc.hiprtcGetProgramLogSize(prog, ctypes.byref(sz))
        log = ctypes.create_string_buffer(sz.value + 1)
        rtc.hiprtcGetProgramLog(prog, log)
        raise RuntimeError(log.value.decode()[:2000])
    sz = ctypes.c_size_t()
    rtc.hiprtcGetCodeSize(prog, ctypes.byref(sz))
    code = ctypes.create_string_buffer(sz.value)
    rtc.hiprtcGetCode(prog, code)
    return code.raw


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--states', type=int, default=32)
    ap.add_argument('--leaves', type=int, default=32)
    ap.add_argument('--random-nodes', type=int, default=0)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--tiles', type=int, default=4)
    ap.add_argument('--prefetch', type=int, default=2)
    ap.add_argument('--out', default='/tmp/jit_offline')
    ap.add_argument('--opt', action='append', default=[])
    ap.add_argument('--source', default=None, help='compile this file instead')
    args = ap.parse_args()
    src = open(args.source, 'rb').read() if args.source else emit_source(
        args.states, args.leaves, args.tiles, args.prefetch, args.random_nodes, args.seed)
    open(args.out + '.hip', 'wb').write(src)
    code = hiprtc_compile(src, mfma=args.states > 4, extra=args.opt)
    open(args.out + '.co', 'wb').write(code)
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    dis = subprocess.run([objdump, '-d', args.out + '.co'], stdout=subprocess.PIPE).stdout.decode()
    open(args.out + '.s', 'w').write(dis)
    notes = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', args.out + '.co'],
                           stdout=subprocess.PIPE).stdout.decode()
    open(args.out + '.notes', 'w').write(notes)
    keys = ('.vgpr_count', '.agpr_count', '.sgpr_count', '.private_segment_fixed_size',
            '.vgpr_spill_count', '.sgpr_spill_count', '.group_segment_fixed_size')
    info = {}
    for k in keys:
        m = re.search(re.escape(k) + r':\s*(\d+)', notes)
        if m:
            info[k] = int(m.group(1))
    counts = {}
    for pat in ('scratch_store', 'scratch_load', 'v_accvgpr_write', 'v_accvgpr_read',
                'v_mfma_f64_16x16x4', 'v_mfma_f64_4x4x4', 's_waitcnt', 's_nop', 'global_load',
                'ds_read', 'ds_write', 'buffer_', 'v_mov_b32', 'v_pk_mov', 'v_mul_f64',
                'v_fma_f64'):
        counts[pat] = len(re.findall(pat, dis))
    print(info)
    print(counts)


if __name__ == '__main__':
    main()
