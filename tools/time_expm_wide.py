"""Diagnostics: the expm launch of one tree's edges under the kernel variants
(RAOTEH_EXPM_WIDE / RAOTEH_EXPM_SPLIT), with RAOTEH_EXPM_TRACE=1 for the phase stamps.
    python tools/time_expm_wide.py [c6|c3]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raoteh_amd import device, synth
name = sys.argv[1] if len(sys.argv) > 1 else 'c6'
ctx = device.get_context()
cfg = synth.make_config(name, nsites=2000)
T, root, n = cfg['T'], cfg['root'], cfg['nstates']
variants = ((0, 0), (1, 0), (1, 1)) if n > 64 else ((0, 0), (0, 1))
ref = None
for wide, split in variants:
    os.environ['RAOTEH_EXPM_WIDE'] = str(wide); os.environ['RAOTEH_EXPM_SPLIT'] = str(split)
    model = device.TreeModel(T, root, n)
    model.set_root_distn(cfg['root_distn'])
    model.set_rates(Q_default=cfg['Q_default'])
    info = model.expm_info()
    P = model.get_transitions()
    if ref is None: ref = P
    for _ in range(5): model.recompute_transitions()
    ctx.sync(); ctx.set_timing(True); ctx.reset_timing()
    for _ in range(20): model.recompute_transitions()
    ctx.sync()
    ms, cnt, kname = ctx.kernel_time(0)
    ctx.set_timing(False)
    print(wide, split, kname, '%.1f us' % (ms / cnt * 1e3), 'm,s', sorted(set(map(tuple, info.tolist()))),
          'bit-identical to the first variant:', bool(np.array_equal(P, ref)))
    model.close()
