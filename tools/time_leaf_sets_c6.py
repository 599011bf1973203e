"""Diagnostics: the c6 batch (allowed sets {c, 61 + c} at the leaves) uploaded as masks -- the
kernels that add two gathered columns of P at a leaf -- next to its dense upload."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raoteh_amd import device, synth, _lib
ctx = device.get_context()
_lib.check(_lib.lib().rt_set_option(b'jit_async', 0))
cfg = synth.make_config('c6')
T, root, n = cfg['T'], cfg['root'], cfg['nstates']
model = device.TreeModel(T, root, n); model.set_root_distn(cfg['root_distn']); model.set_rates(Q_default=cfg['Q_default'])
bd = model.upload_sites(cfg['leaves'], synth.leaf_likelihoods(cfg), kind='dense')
lld = model.log_likelihoods(bd)[0]
words = (n + 63) // 64
table = np.zeros((len(cfg['leaf_allowed']), words), dtype=np.uint64)
for c, ss in enumerate(cfg['leaf_allowed']):
    for k in ss: table[c, k // 64] |= np.uint64(1) << np.uint64(k % 64)
masks = table[cfg['leaf_states']]
for tag, env in (('dense', None), ('masks', {}), ('masks serial', {'RAOTEH_JIT_SPARSE': 'serial'}), ('masks T1 halves', {'RAOTEH_JIT_TILES': '1'})):
    if env is None:
        b = bd
    else:
        for k, v in env.items(): os.environ[k] = v
        b = model.upload_sites(cfg['leaves'], masks, kind='mask')
    ll = model.log_likelihoods(b)[0]
    for _ in range(5): model.prune(b)
    ctx.sync(); ctx.set_timing(True); ctx.reset_timing()
    for _ in range(20): model.prune(b)
    ctx.sync()
    ms, cnt, name = ctx.kernel_time(_lib.RT_K_PRUNE); cms, ccnt, _ = ctx.kernel_time(_lib.RT_K_COMBINE)
    ctx.set_timing(False)
    print(tag, b.kernel_name, '%.1f us' % (ms / cnt * 1e3), '+ combine %.1f' % (cms / max(ccnt, 1) * 1e3), 'bit-identical to dense:', bool(np.array_equal(ll, lld)))
    if env:
        for k in env: del os.environ[k]
