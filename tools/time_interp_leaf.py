"""Diagnostics: the interpreter kernel on the C3 batch uploaded as states (leaf steps gathered)
against the dense upload, and on a 512-leaf tree (no tree-specialised kernel: > 600 steps)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raoteh_amd import device, synth, _lib
ctx = device.get_context()
so = _lib.lib().rt_set_option
_lib.check(so(b'jit', 0))
def run(model, batch, reps=20):
    ll = model.log_likelihoods(batch)[0]
    for _ in range(5): model.prune(batch)
    ctx.sync(); ctx.set_timing(True); ctx.reset_timing()
    for _ in range(reps): model.prune(batch)
    ctx.sync()
    ms, cnt, name = ctx.kernel_time(_lib.RT_K_PRUNE); cms, ccnt, _ = ctx.kernel_time(_lib.RT_K_COMBINE)
    ctx.set_timing(False)
    return ll, '%s %.1f us + combine %.1f' % (batch.kernel_name, ms / cnt * 1e3, cms / max(ccnt, 1) * 1e3)
cfg = synth.make_config('c3')
T, root, n = cfg['T'], cfg['root'], cfg['nstates']
model = device.TreeModel(T, root, n); model.set_root_distn(cfg['root_distn']); model.set_rates(Q_default=cfg['Q_default'])
lld, msg = run(model, model.upload_sites(cfg['leaves'], synth.leaf_likelihoods(cfg), kind='dense')); print('c3 dense :', msg)
lls, msg = run(model, model.upload_sites(cfg['leaves'], cfg['leaf_states'].astype(np.uint8), kind='state')); print('c3 states:', msg, 'bit-identical:', bool(np.array_equal(lls, lld)))
# a big tree
Tb, rootb, leavesb = synth.balanced_tree(512, seed=3)
rng = np.random.RandomState(5)
mb = device.TreeModel(Tb, rootb, n); mb.set_root_distn(cfg['root_distn']); mb.set_rates(Q_default=cfg['Q_default'])
st = rng.randint(0, n, size=(4000, 512)).astype(np.uint8)
lls, msg = run(mb, mb.upload_sites(leavesb, st, kind='state'), 5); print('512 leaves, 4000 sites, states:', msg)
os.environ['RAOTEH_INTERP_NO_SPARSE'] = '1'
lld, msg = run(mb, mb.upload_sites(leavesb, st, kind='state'), 5); print('512 leaves, the products        :', msg, 'bit-identical:', bool(np.array_equal(lls, lld)))
