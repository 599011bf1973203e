import os, sys, numpy as np
sys.path.insert(0, '.')
from raoteh_amd import device, synth
ctx = device.get_context()
cfg = synth.make_config('c6', nsites=2000)
T, root, n = cfg['T'], cfg['root'], cfg['nstates']
for wide, split in ((0, 0), (1, 0), (1, 1)):
    os.environ['RAOTEH_EXPM_WIDE'] = str(wide); os.environ['RAOTEH_EXPM_SPLIT'] = str(split)
    model = device.TreeModel(T, root, n)
    model.set_root_distn(cfg['root_distn'])
    model.set_rates(Q_default=cfg['Q_default'])
    info = model.expm_info()
    for _ in range(5): model.recompute_transitions()
    ctx.sync(); ctx.set_timing(True); ctx.reset_timing()
    for _ in range(20): model.recompute_transitions()
    ctx.sync()
    ms, cnt, name = ctx.kernel_time(0)
    ctx.set_timing(False)
    print(wide, split, name, '%.1f us' % (ms / cnt * 1e3), 'm,s histogram', sorted(set(map(tuple, info.tolist()))))
    model.close()
