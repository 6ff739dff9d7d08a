import sys
sys.path.insert(0, '.')
import numpy as np
from gecco_amd import _native as nat
from benchkit import latency
m = nat.Model.from_lcrf(latency.real_blob())
c, g, a = latency.c1_batch(50, m.num_attrs)
p = nat.Session(m, [0]).windowed_marginals(c, g, a, 20)
import torch
print('torch after the library: cuda available', torch.cuda.is_available(), 'tensor sum', float(torch.ones(4, device='cuda').sum()), 'p[0]', float(p[0]))
libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})
print(libs)
