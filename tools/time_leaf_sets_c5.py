"""Diagnostics: the C5 batch (allowed pairs of compound states at the leaves) uploaded as masks --
the one-wave kernels read the leaves' columns from the blocks parked in LDS -- next to its dense upload."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raoteh_amd import device, synth, _lib
ctx = device.get_context()
_lib.check(_lib.lib().rt_set_option(b'jit_async', 0))
cfg = synth.make_config('c5')
T, root, n = cfg['T'], cfg['root'], cfg['nstates']
model = device.TreeModel(T, root, n); model.set_root_distn(cfg['root_distn']); model.set_rates()
bd = model.upload_sites(cfg['leaves'], synth.leaf_likelihoods(cfg), kind='dense')
lld = model.log_likelihoods(bd)[0]
table = np.array([sum(1 << s for s in ss) for ss in cfg['leaf_allowed']], dtype=np.uint64)
print('allowed set sizes', sorted(set(len(ss) for ss in cfg['leaf_allowed'])))
masks = table[cfg['leaf_states']]
for tag, up in (('dense', None), ('masks', masks)):
    b = bd if up is None else model.upload_sites(cfg['leaves'], up, kind='mask')
    ll = model.log_likelihoods(b)[0]
    for _ in range(5): model.step(b)
    ctx.sync(); ctx.set_timing(True); ctx.reset_timing()
    import time
    t0 = time.perf_counter()
    for _ in range(50): model.step(b)
    ctx.sync(); dt = (time.perf_counter() - t0) / 50
    ms, cnt, name = ctx.kernel_time(_lib.RT_K_PRUNE)
    ctx.set_timing(False)
    print(tag, b.kernel_name, 'kernel %.1f us' % (ms / cnt * 1e3), 'step (every launch timed) %.1f us' % (dt * 1e6), 'bit-identical to dense:', bool(np.array_equal(ll, lld)))
