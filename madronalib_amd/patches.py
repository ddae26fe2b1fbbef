"""Graph descriptions ("patches") in the neutral list-of-dicts form `Graph(engine, V, description)` takes.

`synth16` is BASELINE configs[4]: a 16-node synth voice built only from reference objects
(the reference has no executor for it, so the patch is this repository's
definition and the parity oracle is the same patch evaluated node by node on the CPU,
tests/graph_oracle.py).
"""
import numpy as np

from .constants import Op, Proc


def synth16(pitch_input=False, full=False, event_rows=False):
    """16 processor/op nodes per voice (pitch_input: `pitch` is a streamed signal, e.g. EventsToSignals' pitch row, instead of
    a per-voice constant; the oscillators then see a frequency per sample; event_rows: gate and pitch are computed in the kernel
    from a bound Events object's records instead of read from memory):
         pitch (param, octaves re base) -> exp2Approx -> * baseFreq  = freq (cycles/sample)
         SawGen(freq), PulseGen(freq, width param), LFO SineGen(lfoFreq param), NoiseGen
         osc = saw + pulse * lfo ; pre = osc + noise * noiseLevel
         Lopass -> Hipass -> OnePole -> DCBlocker
         amp ADSR(gate input) ; out = clamp(filtered * env, -1, 1)
       inputs: gate (streamed).  params: pitch, baseFreq, width, lfoFreq, noiseLevel.
       full=True is the patch exactly as SURVEY §8d lists it (bench.py --workload cfg5full): a second ADSR (the filter
       envelope), the cutoff computed per sample as exp2Approx(cutoffOct + envAmount * filterEnv) * cutoffBase, and the Lopass
       run in its per-sample-coefficient form Lopass(x, omega, k) (MLDSPFilters.h:136-152: two libm sinf per sample, restated
       on the device). Further params: cutoffOct, envAmount, cutoffBase, resonance (k)."""
    d = [
        # event_rows: gate and pitch are rows of an EventsToSignals object computed inside the voice kernel (Graph.bind_events)
        dict(name="gate", type="event_row", kind=1) if event_rows else dict(name="gate", type="input"),
        dict(name="pitch", type="event_row", kind=0) if event_rows else dict(name="pitch", type="input" if pitch_input else "param"),
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
    if full:
        i = next(j for j, n in enumerate(d) if n["name"] == "lp")
        d[i:i + 1] = [
            dict(name="cutoffOct", type="param"), dict(name="envAmount", type="param"), dict(name="cutoffBase", type="param"),
            dict(name="resonance", type="param"),
            dict(name="fenv", type="proc", kind=Proc.ADSR, inputs=["gate"]),                        # filter envelope
            dict(name="envOct", type="op", kind=Op.MULTIPLY, inputs=["fenv", "envAmount"]),
            dict(name="oct", type="op", kind=Op.ADD, inputs=["cutoffOct", "envOct"]),
            dict(name="cutRatio", type="op", kind=Op.EXP2_APPROX, inputs=["oct"]),                  # cutoff exp2Approx
            dict(name="omega", type="op", kind=Op.MULTIPLY, inputs=["cutRatio", "cutoffBase"]),
            dict(name="lp", type="proc", kind=Proc.LOPASS, inputs=["pre", "omega", "resonance"]),   # Lopass(x, omega, k)
        ]
    return d, ["out"]


# ---- the reference's feedback composites written out as graphs (what the C++ shim's classes emit) ------------------

def allpass(prefix, x, delay_kind, max_delay, delay=None):
    """Allpass<DELAY_TYPE> (MLDSPFilters.h:1110-1160): vGain = -mGain; vDelayInput = x - vy1*vGain; y = vDelayInput*vGain
    + vy1; vy1 = mDelay(vDelayInput [, delay - 64]). mGain is the param `<prefix>gain`; the inner delay's memory is
    max_delay - 64 (setMaxDelayInSamples, :1123-1126). `delay`: name of the delay-time signal (PitchbendableDelay)."""
    p = prefix
    d = [dict(name=p + "gain", type="param"), dict(name=p + "m1", type="const", value=-1.0), dict(name=p + "c64", type="const", value=64.0),
         dict(name=p + "vy1", type="feedback", source=p + "delay"),
         dict(name=p + "vgain", type="op", kind=Op.MULTIPLY, inputs=[p + "gain", p + "m1"]),   # DSPVector(-mGain): exact negation
         dict(name=p + "fb", type="op", kind=Op.MULTIPLY, inputs=[p + "vy1", p + "vgain"]),
         dict(name=p + "din", type="op", kind=Op.SUBTRACT, inputs=[x, p + "fb"]),
         dict(name=p + "ff", type="op", kind=Op.MULTIPLY, inputs=[p + "din", p + "vgain"]),
         dict(name=p + "y", type="op", kind=Op.ADD, inputs=[p + "ff", p + "vy1"])]
    if delay is None:
        d.append(dict(name=p + "delay", type="proc", kind=delay_kind, inputs=[p + "din"], max_delay=max_delay - 64.0))
    else:
        d.append(dict(name=p + "dt", type="op", kind=Op.SUBTRACT, inputs=[delay, p + "c64"]))
        d.append(dict(name=p + "delay", type="proc", kind=delay_kind, inputs=[p + "din", p + "dt"], max_delay=max_delay - 64.0))
    return d, p + "y"


def fdn(size, x, max_delay):
    """FDN<SIZE> (MLDSPFilters.h:1162-1239): SIZE IntegerDelays + OnePoles around a Householder feedback matrix, one
    DSPVector of feedback latency. Node names: fdn_delay<n>, fdn_filter<n>, param fdn_gain<n>. Outputs (sumL, sumR)."""
    d = [dict(name="fdn_zero", type="const", value=0.0), dict(name="fdn_k", type="const", value=float(np.float32(2.0) / np.float32(size)))]
    for n in range(size):
        d.append(dict(name=f"fdn_gain{n}", type="param"))
        d.append(dict(name=f"fdn_div{n}", type="feedback", source=f"fdn_next{n}"))
        d.append(dict(name=f"fdn_delay{n}", type="proc", kind=Proc.INTEGER_DELAY, inputs=[f"fdn_div{n}"], max_delay=max_delay))
    # sumR / sumL / sumOfDelays start from a zero vector and accumulate with += (:1201-1223)
    accR, accL, acc = "fdn_zero", "fdn_zero", "fdn_zero"
    for n in range(size & ~1):
        tgt = "L" if (n & 1) else "R"
        prev = accL if (n & 1) else accR
        d.append(dict(name=f"fdn_sum{tgt}{n}", type="op", kind=Op.ADD, inputs=[prev, f"fdn_delay{n}"]))
        if n & 1:
            accL = f"fdn_sum{tgt}{n}"
        else:
            accR = f"fdn_sum{tgt}{n}"
    for n in range(size):
        d.append(dict(name=f"fdn_sum{n}", type="op", kind=Op.ADD, inputs=[acc, f"fdn_delay{n}"]))
        acc = f"fdn_sum{n}"
    d.append(dict(name="fdn_sumk", type="op", kind=Op.MULTIPLY, inputs=[acc, "fdn_k"]))
    for n in range(size):
        d.append(dict(name=f"fdn_sub{n}", type="op", kind=Op.SUBTRACT, inputs=[f"fdn_delay{n}", "fdn_sumk"]))
        d.append(dict(name=f"fdn_filter{n}", type="proc", kind=Proc.ONE_POLE, inputs=[f"fdn_sub{n}"]))
        d.append(dict(name=f"fdn_fb{n}", type="op", kind=Op.MULTIPLY, inputs=[f"fdn_filter{n}", f"fdn_gain{n}"]))
        d.append(dict(name=f"fdn_next{n}", type="op", kind=Op.ADD, inputs=[f"fdn_fb{n}", x]))
    return d, [accL, accR]
