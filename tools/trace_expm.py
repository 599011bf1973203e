#!/usr/bin/env python
"""Phase stamps of the Taylor expm kernel on the C3 model (RAOTEH_EXPM_TRACE=1): workgroup 1
stamps the shader clock at its phase boundaries; the launcher prints the differences."""
import os
import sys
os.environ['RAOTEH_EXPM_TRACE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raoteh_amd import device, synth      # noqa: E402

cfg = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else 'c3', nsites=16)
model = device.TreeModel(cfg['T'], cfg['root'], cfg['nstates'])
for _ in range(4):
    model.set_rates(Q_default=cfg['Q_default'])
print(model.expm_info()[:6].tolist())
