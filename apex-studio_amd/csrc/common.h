// Shared device/host helpers for libapex_mi355.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/apexmi.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

typedef uint16_t bf16_t;  // storage type at the C-ABI

#define APEXMI_DEVICE __device__ __forceinline__

APEXMI_DEVICE float bf16_lo(uint32_t u) { return __uint_as_float(u << 16); }
APEXMI_DEVICE float bf16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
APEXMI_DEVICE float bf16_to_f32(bf16_t b) { return __uint_as_float((uint32_t)b << 16); }

// round-to-nearest-even pack of two floats into one dword (lowers to v_cvt_pk_bf16_f32)
APEXMI_DEVICE uint32_t pack_bf16(float a, float b) {
    bf16x2 r;
    r[0] = (__bf16)a;
    r[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, r);
}
APEXMI_DEVICE bf16_t f32_to_bf16(float a) {
    __bf16 r = (__bf16)a;
    return __builtin_bit_cast(bf16_t, r);
}

// unpack 8 bf16 held in a u32x4 into 8 floats
APEXMI_DEVICE void unpack8(const u32x4 v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = bf16_lo(v[i]);
        f[2 * i + 1] = bf16_hi(v[i]);
    }
}
APEXMI_DEVICE u32x4 pack8(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf16(f[2 * i], f[2 * i + 1]);
    return v;
}

// ---- activation storage type --------------------------------------------------------------------------------------
// Production stores activations as bf16 (bf16_t).  The f32-storage VERIFICATION mode (entry points with an `_f32`
// suffix) runs the same kernel bodies with T = float: the arithmetic between a load and a store is f32
// in both, so the only difference is the rounding at the store.  8 consecutive elements per lane either way.
template <typename T>
APEXMI_DEVICE void load8(const T* p, float* f);
template <>
APEXMI_DEVICE void load8<bf16_t>(const bf16_t* p, float* f) { unpack8(*(const u32x4*)p, f); }
template <>
APEXMI_DEVICE void load8<float>(const float* p, float* f) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[i] = a[i];
        f[4 + i] = b[i];
    }
}
template <typename T>
APEXMI_DEVICE void store8(T* p, const float* f);
template <>
APEXMI_DEVICE void store8<bf16_t>(bf16_t* p, const float* f) { *(u32x4*)p = pack8(f); }
template <>
APEXMI_DEVICE void store8<float>(float* p, const float* f) {
    *(f32x4*)p = f32x4{f[0], f[1], f[2], f[3]};
    *(f32x4*)(p + 4) = f32x4{f[4], f[5], f[6], f[7]};
}
// the same 8 elements kept RAW (a load issued now, unpacked later; pure data movement such as transposes)
template <typename T>
struct Raw8;
template <>
struct Raw8<bf16_t> {
    u32x4 v;
};
template <>
struct Raw8<float> {
    f32x4 a, b;
};
APEXMI_DEVICE Raw8<bf16_t> ldraw8(const bf16_t* p) { return Raw8<bf16_t>{*(const u32x4*)p}; }
APEXMI_DEVICE Raw8<float> ldraw8(const float* p) { return Raw8<float>{*(const f32x4*)p, *(const f32x4*)(p + 4)}; }
APEXMI_DEVICE void straw8(bf16_t* p, const Raw8<bf16_t>& r) { *(u32x4*)p = r.v; }
APEXMI_DEVICE void straw8(float* p, const Raw8<float>& r) {
    *(f32x4*)p = r.a;
    *(f32x4*)(p + 4) = r.b;
}
APEXMI_DEVICE void unraw8(const Raw8<bf16_t>& r, float* f) { unpack8(r.v, f); }
APEXMI_DEVICE void unraw8(const Raw8<float>& r, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[i] = r.a[i];
        f[4 + i] = r.b[i];
    }
}
template <typename T>
APEXMI_DEVICE Raw8<T> zero_raw8();
template <>
APEXMI_DEVICE Raw8<bf16_t> zero_raw8<bf16_t>() { return Raw8<bf16_t>{u32x4{0u, 0u, 0u, 0u}}; }
template <>
APEXMI_DEVICE Raw8<float> zero_raw8<float>() { return Raw8<float>{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}; }

template <typename T>
APEXMI_DEVICE float load1(const T* p);
template <>
APEXMI_DEVICE float load1<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <>
APEXMI_DEVICE float load1<float>(const float* p) { return *p; }
template <typename T>
APEXMI_DEVICE void store1(T* p, float v);
template <>
APEXMI_DEVICE void store1<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }
template <>
APEXMI_DEVICE void store1<float>(float* p, float v) { *p = v; }
// 4 consecutive elements (the accumulator-layout epilogues)
template <typename T>
APEXMI_DEVICE void load4(const T* p, float* f);
template <>
APEXMI_DEVICE void load4<bf16_t>(const bf16_t* p, float* f) {
    const u32x2 r = *(const u32x2*)p;
    f[0] = bf16_lo(r[0]);
    f[1] = bf16_hi(r[0]);
    f[2] = bf16_lo(r[1]);
    f[3] = bf16_hi(r[1]);
}
template <>
APEXMI_DEVICE void load4<float>(const float* p, float* f) {
    const f32x4 r = *(const f32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = r[i];
}
template <typename T>
APEXMI_DEVICE void store4(T* p, const float* f);
template <>
APEXMI_DEVICE void store4<bf16_t>(bf16_t* p, const float* f) { *(u32x2*)p = u32x2{pack_bf16(f[0], f[1]), pack_bf16(f[2], f[3])}; }
template <>
APEXMI_DEVICE void store4<float>(float* p, const float* f) { *(f32x4*)p = f32x4{f[0], f[1], f[2], f[3]}; }

// exchange with lane ^ 32 via v_permlane32_swap; returns {value of the low-half lane, value of
// the high-half lane} so max(r0,r1) / r0+r1 are the 2-lane reductions without a select.
APEXMI_DEVICE void swap32(float x, float& a, float& b) {
    uint32_t u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
APEXMI_DEVICE float max_xor32(float x) {
    float a, b;
    swap32(x, a, b);
    return fmaxf(a, b);
}
APEXMI_DEVICE float sum_xor32(float x) {
    float a, b;
    swap32(x, a, b);
    return a + b;
}

APEXMI_DEVICE float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
APEXMI_DEVICE float wave_max(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}

APEXMI_DEVICE float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// x * sigmoid(z) with z pre-scaled by log2(e): one v_exp_f32 and one v_rcp_f32 (1 ulp each) instead of expf + an IEEE division
// (v_div_scale / v_div_fmas / v_div_fixup + Newton steps: ~12 instructions).  The activations sit in GEMM epilogues that run with
// every matrix pipe idle — tools/gemm_tile_trace.py priced the old tanh-form GELU at 14 us per 256 x 256 tile (30 VALU instructions
// a value), a fifth of a K = 3072 tile's K-loop.  x -> -inf: exp2 = inf, rcp = 0, result -0; x -> +inf: result x.
APEXMI_DEVICE float x_sigmoid_log2(float x, float z_log2) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-z_log2));
}
APEXMI_DEVICE float silu_f(float x) { return x_sigmoid_log2(x, 1.4426950408889634f * x); }
// gelu(approximate="tanh") = 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3); 0.5 (1 + tanh(u)) = sigmoid(2u) exactly, which
// also has no cancellation for negative x (the 1 - 2 / (e + 1) form loses the result's leading bits there)
APEXMI_DEVICE float gelu_tanh_f(float x) {
    const float a = 2.0f * 0.7978845608028654f * 1.4426950408889634f, b = a * 0.044715f;
    return x_sigmoid_log2(x, x * fmaf(b, x * x, a));
}

// async 16-byte global -> LDS copy: LDS destination is wave-uniform `lds` + lane*16
APEXMI_DEVICE void glds16(const void* gsrc, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// bijective XCD-aware remap of a 1-D block id: workgroup b is observed to run on XCD b % 8;
// give each XCD a contiguous chunk of the logical tile list so neighbours share its L2.
APEXMI_DEVICE int xcd_remap(int bid, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// ---- host side -------------------------------------------------------------------------------
// Function attributes (dynamic LDS size) are per DEVICE: `mask` holds one bit per device ordinal (0..63).  Use:
//     if (apexmi_attr_needed(mask)) { hipFuncSetAttribute(...); apexmi_attr_done(mask); }
// The bit is set only AFTER the attribute calls returned, so a second host thread on the same device either sees the bit
// (attribute applied) or applies the attribute itself again — harmless — and never launches ahead of it.  Devices with an
// ordinal >= 64 simply set the attribute on every call.
static inline bool apexmi_attr_needed(const uint64_t& mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev >= 64 || !(__atomic_load_n(&mask, __ATOMIC_ACQUIRE) & (1ull << dev));
}
static inline void apexmi_attr_done(uint64_t& mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 64) (void)__atomic_fetch_or(&mask, 1ull << dev, __ATOMIC_RELEASE);
}
#define APEXMI_SET_ATTR_ONCE(mask, ...)      \
    do {                                     \
        if (apexmi_attr_needed(mask)) {      \
            __VA_ARGS__;                     \
            apexmi_attr_done(mask);          \
        }                                    \
    } while (0)
void apexmi_set_error(const char* fmt, ...);
int apexmi_check_launch(const char* what);
// device pointer of the live clock probe's two counters on the current device, or nullptr while the probe is off (runtime.hip)
unsigned long long* apexmi_clk_ptr();

struct ApexmiProfScope {
    int cls;
    hipStream_t stream;
    int slot;
    ApexmiProfScope(int cls, hipStream_t s, double flops, double bytes);
    ~ApexmiProfScope();
};

#define APEXMI_REQUIRE(cond, ...)              \
    do {                                       \
        if (!(cond)) {                         \
            apexmi_set_error(__VA_ARGS__);     \
            return 1;                          \
        }                                      \
    } while (0)
