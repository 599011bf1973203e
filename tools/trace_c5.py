#!/usr/bin/env python
"""
Where does a step of the one-wave MFMA tree-specialised kernel (config 5, 4x4x4 blocks)
spend its time?   RAOTEH_JIT_TRACE=<workgroup> python tools/trace_c5.py [sites] [tiles]
That wave stamps the shader clock at the start of every step (t0), when the step's
operands are in registers (t1) and after its last MFMA has been issued (t2).
"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('RAOTEH_JIT_TRACE', '0')
os.environ.setdefault('RAOTEH_JIT_NO_VERIFY', '1')
os.environ.setdefault('RAOTEH_JIT_QUAD', '1')

from raoteh_amd import _lib, device, synth             # noqa: E402


def main():
    nsites = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    if len(sys.argv) > 2:
        os.environ['RAOTEH_JIT_TILES'] = sys.argv[2]
    cfg = synth.make_config('c5', nsites=nsites)
    ctx = device.get_context(0)
    model = device.TreeModel(cfg['T'], cfg['root'], cfg['nstates'], ctx=ctx)
    model.set_rates(Q_default=cfg['Q_default'])
    model.set_root_distn(cfg['root_distn'])
    batch = model.upload_sites(cfg['leaves'], synth.leaf_likelihoods(cfg), kind='dense')
    for _ in range(5):
        model.prune(batch)
    ctx.sync()
    nops = ctypes.c_int64(0)
    _lib.check(_lib.lib().rt_model_get_schedule(model._h, None, 0, ctypes.byref(nops)))
    nrec = nops.value
    ops = np.zeros((nrec, 4), dtype=np.int32)
    _lib.check(_lib.lib().rt_model_get_schedule(
        model._h, ops.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), nrec, ctypes.byref(nops)))
    tr = np.zeros((nrec + 1, 3), dtype=np.uint64)
    _lib.check(_lib.lib().rt_debug_jit_global(batch._h, b'rt_trace',
                                              tr.ctypes.data_as(ctypes.c_void_p), tr.nbytes))
    tr = tr.astype(np.int64)
    leaf = ops[:, 2] < 0
    root = ops[:, 3] < 0
    ok = ~root
    t0, t1, t2, nxt = tr[:nrec, 0], tr[:nrec, 1], tr[:nrec, 2], tr[1:, 0]
    out = dict(kernel=batch.kernel_name, workgroup=int(os.environ['RAOTEH_JIT_TRACE']))
    for name, sel in (('leaf', leaf & ok), ('internal', ~leaf & ok)):
        out[name] = dict(steps=int(sel.sum()), operands=float(np.mean((t1 - t0)[sel])),
                         chain=float(np.mean((t2 - t1)[sel])), tail=float(np.mean((nxt - t2)[sel])),
                         step=float(np.mean((nxt - t0)[sel])))
    out['total_cycles'] = int(tr[nrec, 0] - tr[0, 0])
    real = int(tr[nrec, 2] - tr[nrec, 1])              # ticks of the constant 100 MHz clock
    if real > 0:
        out['wave_us'] = real / 100.0
        out['core_clock_ghz'] = out['total_cycles'] / (real / 100.0) / 1e3
    print(json.dumps(out, indent=1))
    print('first 16 steps: (leaf?, operands, chain, tail)')
    for i in range(min(16, nrec - 1)):
        print(i, bool(leaf[i]), int(t1[i] - t0[i]), int(t2[i] - t1[i]), int(nxt[i] - t2[i]))


if __name__ == '__main__':
    main()
