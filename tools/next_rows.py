import json, sys
sys.path.insert(0,'.')
import bench
from raoteh_amd import device
ctx = device.get_context()
out = bench.measure_next_rows(ctx)
for k,v in out.items():
    print(k, json.dumps(v)[:600])
