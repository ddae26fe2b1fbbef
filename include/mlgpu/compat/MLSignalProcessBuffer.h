// madronalib's header name, forwarded to the GPU shim (source-level drop-in)
#pragma once
#include "mldsp.h"
