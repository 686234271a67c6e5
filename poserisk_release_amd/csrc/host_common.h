// Host-side helpers that need no HIP header: the error state and the status macros.  Included by common.h (device
// translation units) and by host_plan.cc, which is also compiled WITHOUT the HIP toolchain for the sanitizer build of
// tests/native (g++ -fsanitize=address,undefined).
#pragma once
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/poserisk_hip.h"

namespace pr {

void set_error(const char* fmt, ...);

#define PR_REQUIRE(cond, ...)        \
  do {                               \
    if (!(cond)) {                   \
      pr::set_error(__VA_ARGS__);    \
      return PR_ERR_INVALID;         \
    }                                \
  } while (0)

#define PR_TRY(expr)             \
  do {                           \
    int s__ = (expr);            \
    if (s__ != PR_OK) return s__; \
  } while (0)

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline long ceil_div(long a, long b) { return (a + b - 1) / b; }

}  // namespace pr
