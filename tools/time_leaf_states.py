import os, sys, time, numpy as np
sys.path.insert(0, '.')
from raoteh_amd import device, synth, _lib
ctx = device.get_context()
_lib.check(_lib.lib().rt_set_option(b'jit_async', 0))
cfg = synth.make_config('c3')
T, root, n = cfg['T'], cfg['root'], cfg['nstates']
model = device.TreeModel(T, root, n); model.set_root_distn(cfg['root_distn']); model.set_rates(Q_default=cfg['Q_default'])
bd = model.upload_sites(cfg['leaves'], synth.leaf_likelihoods(cfg), kind='dense')
lld = model.log_likelihoods(bd)[0]
for tag, env in (('default', {}), ('serial', {'RAOTEH_JIT_SPARSE': 'serial'}), ('T3', {'RAOTEH_JIT_TILES': '3'}), ('T1', {'RAOTEH_JIT_TILES': '1'}), ('nohalves', {'RAOTEH_JIT_HALVES': '0'})):
    for k, v in env.items(): os.environ[k] = v
    bs = model.upload_sites(cfg['leaves'], cfg['leaf_states'].astype(np.uint8), kind='state')
    lls = model.log_likelihoods(bs)[0]
    for _ in range(10): model.prune(bs)
    ctx.sync(); ctx.set_timing(True); ctx.reset_timing()
    for _ in range(30): model.prune(bs)
    ctx.sync()
    ms, cnt, name = ctx.kernel_time(_lib.RT_K_PRUNE)
    cms, ccnt, _ = ctx.kernel_time(_lib.RT_K_COMBINE)
    ctx.set_timing(False)
    print(tag, bs.kernel_name, '%.1f us' % (ms / cnt * 1e3), '+ combine %.1f' % (cms / max(ccnt, 1) * 1e3), 'bit-identical to dense:', bool(np.array_equal(lls, lld)))
    bs.close()
    for k in env: del os.environ[k]
