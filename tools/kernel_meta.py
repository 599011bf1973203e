#!/usr/bin/env python
"""Print per-kernel register / LDS / scratch usage from a hipcc -S listing."""
import re
import sys

text = open(sys.argv[1]).read()
blocks = re.split(r'\n\s+- \.agpr_count:', text)
print('%-58s %5s %5s %5s %7s %7s %6s' % ('kernel', 'vgpr', 'agpr', 'sgpr', 'lds', 'scratch', 'vspill'))
for b in blocks[1:]:
    b = '.agpr_count:' + b
    def g(k):
        m = re.search(r'\.%s:\s+(\S+)' % k, b)
        return m.group(1) if m else '?'
    name = g('name')
    name = re.sub(r'^_Z\d+', '', name)[:58]
    print('%-58s %5s %5s %5s %7s %7s %6s' % (name, g('vgpr_count'), g('agpr_count'), g('sgpr_count'),
                                            g('group_segment_fixed_size'), g('private_segment_fixed_size'),
                                            g('vgpr_spill_count')))
