"""Developer tool: config 3 with the frequency streamed as a per-sample signal (SURVEY §8d 'signal-freq mode', 8 B/voice-sample)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import madronalib_amd as ml  # noqa: E402
from madronalib_amd.constants import Layout, Proc  # noqa: E402
from madronalib_amd.sharding import cfg3_voice_params  # noqa: E402

V, T = 262144, 30
eng = ml.Engine(0)
n = V * T * 64
bank = eng.bank([Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN], V)
bank.clear()
freq, co = cfg3_voice_params(0, V, V, ml.Bandpass.makeCoeffs)
for i in range(3):
    bank.set_coeff(1, i, co[i])
bank.set_coeff(2, 0, 0.25)
d_f = eng.alloc(4 * n)
d_f.upload(np.tile(np.repeat(freq[None, :, None], 4, 2), (16 * T, 1, 1)).astype(np.float32))   # QUAD [16T][V][4]
d_o = eng.alloc(4 * n)
for _ in range(5):
    bank.process(T, d_o, Layout.QUAD, d_f, Layout.QUAD)
eng.sync()
eng.timer_start()
for _ in range(50):
    bank.process(T, d_o, Layout.QUAD, d_f, Layout.QUAD)
ms = eng.timer_stop_ms() / 50
print(f"signal-freq mode: {ms:.3f} ms per launch, {n / ms / 1e6:.1f} G voice-samples/s, {8 * n / ms / 1e9:.2f} TB/s")
