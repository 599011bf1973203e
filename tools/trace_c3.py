#!/usr/bin/env python
"""
Where does a step of the split-M tree-specialised kernel (config 3) spend its time?
Run on a GPU box:   RAOTEH_JIT_TRACE=<workgroup> python tools/trace_c3.py [sites] [tiles]
The waves of that workgroup stamp the shader clock at the start of every step (t0), after
the x-exchange barrier (t1) and after the last MFMA of the step has been issued (t2).
Prints, per wave, the mean prelude (t1 - t0), chain (t2 - t1) and tail (t0' - t2) in
cycles, split into leaf steps and internal steps.
"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('RAOTEH_JIT_TRACE', '0')
os.environ.setdefault('RAOTEH_JIT_NO_VERIFY', '1')     # the trace global changes nothing, skip

from raoteh_amd import _lib, device, synth             # noqa: E402


def main():
    nsites = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    if len(sys.argv) > 2:
        os.environ['RAOTEH_JIT_TILES'] = sys.argv[2]
    cfg = synth.make_config('c3', nsites=nsites)
    ctx = device.get_context(0)
    model = device.TreeModel(cfg['T'], cfg['root'], cfg['nstates'], ctx=ctx)
    model.set_rates(Q_default=cfg['Q_default'])
    model.set_root_distn(cfg['root_distn'])
    batch = model.upload_sites(cfg['leaves'], cfg['leaf_states'].astype(np.uint8), kind='state')
    for _ in range(5):
        model.prune(batch)
    ctx.sync()
    nops = ctypes.c_int64(0)
    _lib.check(_lib.lib().rt_model_get_schedule(model._h, None, 0, ctypes.byref(nops)))
    nrec = nops.value
    ops = np.zeros((nrec, 4), dtype=np.int32)
    _lib.check(_lib.lib().rt_model_get_schedule(
        model._h, ops.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), nrec, ctypes.byref(nops)))
    NT = 4
    tr = np.zeros((NT, nrec + 1, 3), dtype=np.uint64)
    _lib.check(_lib.lib().rt_debug_jit_global(batch._h, b'rt_trace', tr.ctypes.data_as(ctypes.c_void_p),
                                              tr.nbytes))
    tr = tr.astype(np.int64)
    leaf = ops[:, 2] < 0                 # pop < 0
    root = ops[:, 3] < 0
    out = dict(kernel=batch.kernel_name, workgroup=int(os.environ['RAOTEH_JIT_TRACE']), waves=[])
    for w in range(NT):
        t0, t1, t2 = tr[w, :nrec, 0], tr[w, :nrec, 1], tr[w, :nrec, 2]
        nxt = tr[w, 1:, 0]
        ok = ~root
        rows = {}
        for name, sel in (('leaf', leaf & ok), ('internal', ~leaf & ok)):
            rows[name] = dict(steps=int(sel.sum()),
                              prelude=float(np.mean((t1 - t0)[sel])),
                              chain=float(np.mean((t2 - t1)[sel])),
                              tail=float(np.mean((nxt - t2)[sel])),
                              step=float(np.mean((nxt - t0)[sel])))
        # the pipelined generator has no separate prelude: "issue" = step start to last
        # MFMA issued (chain + everything in its shadow), "sync" = from there to the next
        # step's start (barrier + whatever ran serially)
        rows['issue_to_last_mfma'] = float(np.mean((t2 - t0)[ok]))
        rows['sync_after_last_mfma'] = float(np.mean((nxt - t2)[ok]))
        rows['total_cycles'] = int(tr[w, nrec, 0] - tr[w, 0, 0])
        out['waves'].append(rows)
    print(json.dumps(out, indent=1))
    w0 = tr[0]
    print('first 12 steps of wave 0: (leaf?, prelude, chain, tail)')
    for i in range(12):
        print(i, bool(leaf[i]), int(w0[i, 1] - w0[i, 0]), int(w0[i, 2] - w0[i, 1]),
              int(w0[i + 1, 0] - w0[i, 2]))


if __name__ == '__main__':
    main()
