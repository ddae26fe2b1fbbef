cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys; sys.path.insert(0,'.')
import numpy as np
import madronalib_amd as ml
from madronalib_amd.constants import Op, Proc
from madronalib_amd import patches
e=ml.Engine(0)
desc = [dict(name="x", type="input"), dict(name="g", type="const", value=0.995),
        dict(name="fb", type="feedback", source="damp"),
        dict(name="fbg", type="op", kind=Op.MULTIPLY, inputs=["fb", "g"]),
        dict(name="sum", type="op", kind=Op.ADD, inputs=["x", "fbg"]),
        dict(name="line", type="proc", kind=Proc.FRACTIONAL_DELAY, inputs=["sum"], max_delay=1024.0),
        dict(name="damp", type="proc", kind=Proc.ONE_POLE, inputs=["line"])]
for lay in (0,1,2):
    g=ml.Graph(e,262144,desc,["damp"],delay_windows=lay)
    print("strings layout",lay,"workgroups per CU",g.workgroups_per_cu())
for nm,kw in (("cfg5",{}),("cfg5full",dict(full=True)),("synthvoice",dict(pitch_input=True))):
    d,o=patches.synth16(**kw); g=ml.Graph(e,262144,d,o); print(nm,"workgroups per CU",g.workgroups_per_cu())
d,o=patches.synth16(pitch_input=True,event_rows=True); g=ml.Graph(e,262144,d,o,output_groups={0:16}); print("synth","workgroups per CU",g.workgroups_per_cu())
PY
