// hulc_amd/csrc/engine_f16.hip — the IEEE fp16 build of the engine and of every half-precision kernel (namespace hulc_f16).
// Same sources as the bf16 build in capi.hip; common.h switches the conversions and the MFMA opcode on HULC_HALF_F16.
// The reference trains at `precision: 16` (conf/trainer/play_trainer.yaml:3 = Lightning native AMP: fp16 autocast + GradScaler);
// the loss scaler that goes with this mode lives in Engine<T> (engine.h, "dynamic loss scaling").
#define HULC_HALF_F16 1
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <algorithm>
#include <stdexcept>

#include "engine.h"
