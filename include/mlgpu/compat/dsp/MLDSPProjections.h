// madronalib's header name, forwarded to the shim's own intervals and projections (mlscalar.h)
#pragma once
#include "../mlscalar.h"
