// madronalib.h (MI355X drop-in): the reference's umbrella header (include/madronalib.h) pulls in mldsp.h plus the
// app layer; of the app layer only the process-function boundary (AudioContext inputs/outputs, SignalProcessFn)
// exists here — see mldsp.h in this directory.
#pragma once
#include "mldsp.h"
