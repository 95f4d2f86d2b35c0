// Shared helpers for the libhmx.so kernels (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <set>
#include <utility>

#include "../../include/hmx.h"

#define HMX_WAVE 64

// The HIP "last error" is sticky per thread and may hold a benign error left by other code in the
// process (an event query, a probing call): clear it on entry so that HMX_CHECK_LAUNCH reports only ours.
#define HMX_ENTER() (void)hipGetLastError()

#define HMX_CHECK_LAUNCH()                            \
    do {                                              \
        hipError_t e_ = hipGetLastError();            \
        if (e_ != hipSuccess) return HMX_ELAUNCH;     \
    } while (0)

// Raises a kernel's dynamic-LDS limit, once per (kernel, device): the attribute belongs to the kernel's image on the
// device that is current when it is set, so a process that uses several GPUs sets it on each.  Not capture-safe on
// the first call for a (kernel, device); later calls only look the pair up.
inline void hmx_allow_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(mu);
    if (done.insert({kernel, dev}).second)
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

__host__ __device__ static inline int64_t hmx_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t hmx_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Carves consecutive 256-byte aligned regions out of a caller-provided workspace.
struct HmxCarver {
    char* base;
    size_t off;
    explicit HmxCarver(void* p) : base(static_cast<char*>(p)), off(0) {}
    template <typename T>
    T* take(size_t count) {
        T* r = reinterpret_cast<T*>(base + off);
        off += hmx_align_up(count * sizeof(T), 256);
        return r;
    }
};

// round-to-nearest-even fp32 -> bf16 bits (finite inputs; NaN stays NaN via the quiet bit)
__device__ __forceinline__ uint16_t hmx_f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float hmx_bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// Two fp32 -> two bf16 (round to nearest even) in one v_cvt_pk_bf16_f32; `a` lands in the low half.
typedef __bf16 hmx_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float hmx_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t hmx_pack_bf16x2(float a, float b) {
    const hmx_f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hmx_bf16x2_t));
}
