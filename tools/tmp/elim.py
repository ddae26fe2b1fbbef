import os, sys, subprocess, json, re
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
if len(sys.argv) > 1 and sys.argv[1] == "dump":
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
    open('/tmp/synth_base.hip', 'w').write(g.source)
    sys.exit(0)
subprocess.check_call([sys.executable, __file__, "dump"])
base = open('/tmp/synth_base.hip').read()
osc_old = re.search(r"        if \(slocked10 && !oddw11\) (step_locked_stream<true>\([^;]*;)\n        else if \(slocked10\).*?\n        \}\n", base, re.S)
assert osc_old, "oscillator pattern"
osc_only = base.replace(osc_old.group(0), "        " + osc_old.group(1) + "\n")
variants = {
    "base (file = the generated source)": base,
    "oscillator pair: no 3-way test (locked, regular width assumed)": osc_only,
    "ADSR: no segment test": "#define MLGPU_X_ADSR_NO_SEGMENT_TEST 1\n" + base,
    "both": "#define MLGPU_X_ADSR_NO_SEGMENT_TEST 1\n" + osc_only,
    "both + no turns": ("#define MLGPU_X_ADSR_NO_SEGMENT_TEST 1\n" + osc_only).replace("if ((q & 1) == 0) take_turns_by_clock(turn0, 13);", ""),
}
def run(env):
    out = subprocess.run([sys.executable, "bench.py", "--workload", "synth", "--no-cpu-baseline", "--no-extras", "--no-live-counters", "--steps", "20", "--warmup", "5"],
                         env=dict(os.environ, **env), capture_output=True, text=True).stdout.strip().split("\n")[-1]
    d = json.loads(out)
    return d["ms_per_step"], d["roofline"]["kernel_ms"]
print("no file:", run({}))
for name, src in variants.items():
    path = "/tmp/synth_variant.hip"
    open(path, "w").write(src)
    print(name, ":", run({"MLGPU_GRAPH_SOURCE_FILE": path}), flush=True)
print("no file again:", run({}))
