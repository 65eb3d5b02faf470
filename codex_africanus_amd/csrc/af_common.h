// Shared host/device helpers for libafhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/afhip.h"

#define AF_EXPORT extern "C" __attribute__((visibility("default")))

// africanus/constants/consts.py:6-9 (bit-for-bit: 2*math.pi/c, left to right)
#define AF_LIGHTSPEED 2.99792458e8
#define AF_TWO_PI_OVER_C (2 * 3.141592653589793 / AF_LIGHTSPEED)
#define AF_MINUS_TWO_PI_OVER_C (-AF_TWO_PI_OVER_C)

void af_set_error(const char *fmt, ...);
int af_hip_fail(hipError_t e, const char *what, const char *file, int line);

#define AF_HIP(call)                                                          \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess) return af_hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

#define AF_REQUIRE(cond, ...)                   \
    do {                                        \
        if (!(cond)) {                          \
            af_set_error(__VA_ARGS__);          \
            return AF_EINVAL;                   \
        }                                       \
    } while (0)

// launch check: kernel launches report configuration errors through hipGetLastError
#define AF_LAUNCH_CHECK() AF_HIP(hipGetLastError())

// bench support: record the thread's profile events (af_profile_events) around the dominant
// kernel(s) of an entry point, on the stream they are launched on
void af_prof_begin(hipStream_t st);
void af_prof_end(hipStream_t st);

// Stage hooks of the fused kernels (tools/profile_fused.sh, tools/profile_gemm_stage.sh: run ONE stage of a kernel, whose
// output is then meaningless) exist only in the profiling build (make HOOKS=1 -> ../lib/prof/libafhip.so, selected with
// AFHIP_LIB): the shipped library compiles them to their defaults and a stray environment variable changes nothing.
#ifdef AFHIP_STAGE_HOOKS
#include <stdlib.h>
#define AF_STAGE_ENV(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define AF_STAGE_ENV(name, dflt) (dflt)
#endif

static inline hipStream_t af_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t af_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t af_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
