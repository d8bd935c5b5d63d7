// Shared host/device helpers for the dj_* C-ABI library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#define DJ_OK 0
#define DJ_ERR_ARG (-1)
#define DJ_ERR_HIP (-2)
#define DJ_ERR_UNSUPPORTED (-3)

// Thread-local last-error text, read back through dj_last_error().
void dj_set_error(const char* fmt, ...);

#define DJ_CHECK_ARG(cond, ...)                    \
  do {                                             \
    if (!(cond)) {                                 \
      dj_set_error(__VA_ARGS__);                   \
      return DJ_ERR_ARG;                           \
    }                                              \
  } while (0)

#define DJ_CHECK_LAUNCH(name)                                                  \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      dj_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
      return DJ_ERR_HIP;                                                       \
    }                                                                          \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int dj_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
