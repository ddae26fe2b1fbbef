import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import madronalib_amd as ml
from madronalib_amd import patches
eng = ml.Engine(0)
N, P, T = 16384, 16, 16
ev = ml.Events(eng, N, P, 48000.0)
ev.configure(glide_seconds=0.01, drift=0.5)
ev.set_wanted_rows([0, 1])
ev.reserve_for_graph(T)
desc, outn = patches.synth16(pitch_input=True, event_rows=True)
g = ml.Graph(eng, N * P, desc, outn, output_groups={0: P})
g.bind_events(ev)
open('gpurun_out/synth_rows.hip', 'w').write(g.source)
