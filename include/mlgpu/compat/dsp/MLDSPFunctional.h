// MLDSPFunctional.h (MI355X drop-in): sources that include madronalib's DSP headers by their own names get the shim.
// Put include/mlgpu/compat/dsp AND include/mlgpu/compat on the include path ahead of madronalib's source/DSP.
#pragma once
#include "../mldsp.h"
