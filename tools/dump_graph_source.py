import sys; sys.path.insert(0,'.')
import madronalib_amd as ml
from madronalib_amd import patches
e=ml.Engine(0)
for vpl in (1,2):
    d,o=patches.synth16()
    g=ml.Graph(e,1024,d,o,voices_per_lane=vpl)
    open(f'gpurun_out/synth16_vpl{vpl}.hip','w').write(g.source)
