// madronalib's header name, forwarded to the shim's own scalar helpers (mlscalar.h)
#pragma once
#include "../mlscalar.h"
