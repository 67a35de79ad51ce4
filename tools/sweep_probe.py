import sys; sys.path.insert(0,'/root/repo')
import numpy as np, slicer_amd
S=slicer_amd.Slicer(0)
for box in (1000.0, 500000.0, 100.0, 250.0, 64.0, 1000.5, 0.1, 3.3e6, 1.0):
    n, ex = S.debug_box_quotient(box)
    print(box, n, [float(e) for e in ex], [hex(int(v)) for v in ex.view(np.uint32)])
