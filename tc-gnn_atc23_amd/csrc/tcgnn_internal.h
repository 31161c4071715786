// tcgnn_internal.h - declarations shared by the host and device halves of libtcgnn_hip.so.
#ifndef TCGNN_INTERNAL_H
#define TCGNN_INTERNAL_H

#include <cstdarg>
#include <cstdint>

namespace tcgnn {

// Records a thread-local message for tcgnn_last_error() and returns `status`.
int fail(int status, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Geometry of the packed tile stream ("wide blocks": four 16x8 TC blocks side by side).
constexpr int kWinRows = 16;   // BLK_H
constexpr int kTcCols = 8;     // BLK_W
constexpr int kWbCols = 32;    // condensed columns per MFMA operand tile (K of 16x16x32)
constexpr int kMaxChunkDims = 128; // feature columns handled by one workgroup pass

} // namespace tcgnn
#endif
