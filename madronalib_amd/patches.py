"""Graph descriptions ("patches") in the neutral list-of-dicts form `Graph(engine, V, description)` takes.

`synth16` is BASELINE configs[4]: a 16-node synth voice built only from reference objects
(the reference has no executor for it, so the patch is this repository's
definition and the parity oracle is the same patch evaluated node by node on the CPU,
tests/graph_oracle.py).
"""
from .constants import Op, Proc


def synth16():
    """16 processor/op nodes per voice:
         pitch (param, octaves re base) -> exp2Approx -> * baseFreq  = freq (cycles/sample)
         SawGen(freq), PulseGen(freq, width param), LFO SineGen(lfoFreq param), NoiseGen
         osc = saw + pulse * lfo ; pre = osc + noise * noiseLevel
         Lopass -> Hipass -> OnePole -> DCBlocker
         amp ADSR(gate input) ; out = clamp(filtered * env, -1, 1)
       inputs: gate (streamed).  params: pitch, baseFreq, width, lfoFreq, noiseLevel."""
    d = [
        dict(name="gate", type="input"),
        dict(name="pitch", type="param"),
        dict(name="baseFreq", type="param"),
        dict(name="width", type="param"),
        dict(name="lfoFreq", type="param"),
        dict(name="noiseLevel", type="param"),
        dict(name="lo", type="const", value=-1.0),
        dict(name="hi", type="const", value=1.0),
        dict(name="ratio", type="op", kind=Op.EXP2_APPROX, inputs=["pitch"]),                 # 1
        dict(name="freq", type="op", kind=Op.MULTIPLY, inputs=["ratio", "baseFreq"]),         # 2
        dict(name="saw", type="proc", kind=Proc.SAW_GEN, inputs=["freq"]),                    # 3
        dict(name="pulse", type="proc", kind=Proc.PULSE_GEN, inputs=["freq", "width"]),       # 4
        dict(name="lfo", type="proc", kind=Proc.SINE_GEN, inputs=["lfoFreq"]),                # 5
        dict(name="noise", type="proc", kind=Proc.NOISE_GEN, inputs=[]),                      # 6
        dict(name="pulseMod", type="op", kind=Op.MULTIPLY, inputs=["pulse", "lfo"]),          # 7
        dict(name="osc", type="op", kind=Op.ADD, inputs=["saw", "pulseMod"]),                 # 8
        dict(name="noiseScaled", type="op", kind=Op.MULTIPLY, inputs=["noise", "noiseLevel"]),  # 9
        dict(name="pre", type="op", kind=Op.ADD, inputs=["osc", "noiseScaled"]),              # 10
        dict(name="lp", type="proc", kind=Proc.LOPASS, inputs=["pre"]),                       # 11
        dict(name="hp", type="proc", kind=Proc.HIPASS, inputs=["lp"]),                        # 12
        dict(name="smooth", type="proc", kind=Proc.ONE_POLE, inputs=["hp"]),                  # 13
        dict(name="dc", type="proc", kind=Proc.DC_BLOCKER, inputs=["smooth"]),                # 14
        dict(name="env", type="proc", kind=Proc.ADSR, inputs=["gate"]),                       # 15
        dict(name="vca", type="op", kind=Op.MULTIPLY, inputs=["dc", "env"]),                  # 16
        dict(name="out", type="op", kind=Op.CLAMP, inputs=["vca", "lo", "hi"]),               # 17 (output clamp)
    ]
    return d, ["out"]
