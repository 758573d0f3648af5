// dmx_internal.h -- shared declarations of libdemux_hip.so (not part of the public ABI).
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "demux_hip.h"
#include "demux_hip_debug.h"

namespace dmx {

// records a message for dmx_last_error() and returns `code`
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

}  // namespace dmx
