#!/usr/bin/env python
"""Per-step clock stamps of the pipelined leaf-state kernel on the C3 batch uploaded as states
(RAOTEH_JIT_TRACE=<workgroup>): start of step, after the barrier, after the last MFMA issued.
    RAOTEH_JIT_TRACE=0 python tools/trace_leaf_states.py [tiles]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('RAOTEH_JIT_TRACE', '0')
os.environ.setdefault('RAOTEH_JIT_NO_VERIFY', '1')
if len(sys.argv) > 1:
    os.environ['RAOTEH_JIT_TILES'] = sys.argv[1]
from raoteh_amd import _lib, device, synth
_lib.check(_lib.lib().rt_set_option(b'jit_async', 0))
cfg = synth.make_config('c3', nsites=10000)
ctx = device.get_context(0)
model = device.TreeModel(cfg['T'], cfg['root'], cfg['nstates'], ctx=ctx)
model.set_rates(Q_default=cfg['Q_default']); model.set_root_distn(cfg['root_distn'])
batch = model.upload_sites(cfg['leaves'], cfg['leaf_states'].astype(np.uint8), kind='state')
for _ in range(5): model.prune(batch)
ctx.sync()
print(batch.kernel_name)
NT, nmax = 4, 33
for nmax in (33, 32, 34, 64, 65, 128):
    tr = np.zeros((NT, nmax, 3), dtype=np.uint64)
    rc = _lib.lib().rt_debug_jit_global(batch._h, b'rt_trace', tr.ctypes.data_as(ctypes.c_void_p), tr.nbytes)
    if rc == 0: break
print('steps+1 =', nmax)
tr = tr.astype(np.int64)
for w in range(NT):
    t0, t1, t2 = tr[w, :, 0], tr[w, :, 1], tr[w, :, 2]
    n = nmax - 1
    valid = [k for k in range(n - 1) if t2[k] > 0 and t0[k + 1] > 0]
    chain = np.array([t2[k] - t1[k] if t1[k] > 0 else t2[k] - t0[k] for k in valid])
    pre = np.array([t1[k] - t0[k] if t1[k] > 0 else 0 for k in valid])
    tail = np.array([t0[k + 1] - t2[k] for k in valid])
    print('wave %d: %d steps; mean chain %.0f, prelude %.0f, tail %.0f clocks; total %.0f' % (
        w, len(valid), chain.mean(), pre.mean(), tail.mean(), t0[valid[-1] + 1] - t0[valid[0]]))
    if w == 0:
        print('  per step (start->barrier, barrier->last MFMA, ->next start):')
        print('  ' + ' '.join('%d/%d/%d' % (pre[i], chain[i], tail[i]) for i in range(len(valid))))
