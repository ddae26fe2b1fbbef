// MLAudioTask.h (MI355X drop-in): the reference's console examples include this header (source/app/MLAudioTask.h) for
// AudioTask, AudioContext and mldsp.h. All three are in mldsp.h of this directory.
#pragma once
#include "madronalib.h"
