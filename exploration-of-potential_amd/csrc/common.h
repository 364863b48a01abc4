// ep24 - shared device/host helpers for the gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/ep24.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define EP24_WAVE 64

// ---- error plumbing (host) -------------------------------------------------------------------------
void ep24_set_error(const char* fmt, ...);

#define EP24_REQUIRE(cond, code, ...)            \
    do {                                         \
        if (!(cond)) {                           \
            ep24_set_error(__VA_ARGS__);         \
            return (code);                       \
        }                                        \
    } while (0)

// never synchronises the stream: only picks up launch-time errors
#define EP24_LAUNCH_CHECK(name)                                                        \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            ep24_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return EP24_E_LAUNCH;                                                      \
        }                                                                              \
    } while (0)

// ---- device helpers --------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// v_exp_f32 + v_rcp_f32 (1 ulp each).  Written as a division the compiler emits the full IEEE sequence (div_scale,
// rcp, 4 fma, div_fmas, div_fixup: ~10 instructions per element), which made the BN + SiLU kernels VALU-bound.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// activation of the BN+act blocks (get_activation, network_blocks.py:17-26): act = 1 SiLU (the 24p network), 2 ReLU (also the
// ResNet / DenseNet backbones of config 4), 3 LeakyReLU(0.1), 0 identity
__device__ __forceinline__ float act_fwd(float u, int act) {
    if (act == 1) return u * sigmoidf_(u);
    if (act == 3) return u > 0.f ? u : 0.1f * u;
    return act == 2 ? fmaxf(u, 0.f) : u;
}
// d act(u) / du
__device__ __forceinline__ float act_grad(float u, int act) {
    if (!act) return 1.f;
    if (act == 2) return u > 0.f ? 1.f : 0.f;
    if (act == 3) return u > 0.f ? 1.f : 0.1f;
    const float s = sigmoidf_(u);
    return s * fmaf(u, 1.f - s, 1.f);
}

// BatchNorm statistics are accumulated across workgroups as 2^-20 fixed point in int64 atomics: integer
// addition is associative, so the batch statistics (and with them the whole forward pass) are bitwise
// reproducible whatever order the workgroups retire in.  |sum| < 2^43 in real units.
#define EP24_FIX_SCALE 1048576.0f
__device__ __forceinline__ long long to_fix(float v) { return (long long)__float2ll_rn(v * EP24_FIX_SCALE); }
__device__ __forceinline__ float from_fix(long long v) { return (float)((double)v * (1.0 / 1048576.0)); }
// The two sums of BatchNorm BACKWARD (du and du * zhat over the pixels) are sums of GRADIENTS: in the class branch of the head a
// workgroup's partial sum is 1e-5 .. 1e-4, i.e. 10 .. 100 units of 2^-20 - round 3 found channels whose gamma / beta gradient had
// rounded to exactly 0 that way (0.8 % of the channels of the head's first class conv).  They use 2^-36: resolution 1.5e-11,
// valid range |partial sum| < 2^17 (gradient sums over 2 M pixels stay many orders below that).
// A diverged gradient must stay visible: __float2ll_rn turns NaN into 0 and the scaled value wraps beyond 2^27, so
// a partial sum that is NaN or beyond 2^20 in magnitude (sums of O(1) gradients over 2 M pixels stay below it; real ones by orders
// of magnitude) is replaced by the marker 2^57, and a word decodes to NaN when it is at or beyond 2^56 in EITHER direction:
//   - a legitimate word is a sum of partials below 2^56 (2^20 real) in magnitude that itself stays below it;
//   - one marker plus ANY legitimate rest (also a net-negative one: the first form, marker 2^53 tested with v >= 2^53, decoded
//     "marker minus rest" as a finite number) is >= 2^56;
//   - k markers wrap the int64 only at k = 128: for k < 64 the word is >= 2^56, for 64 <= k < 128 it is <= -2^57.  A replica
//     word collects at most grid / replicas partials (<= 64 on the standard path); the fold tests every replica word by itself
//     before adding them (fixg_bad), so replicas cannot cancel each other's markers.
#define EP24_FIXG_MARK (1LL << 57)
#define EP24_FIXG_LIMIT (1LL << 56)
__device__ __forceinline__ long long to_fix_g(float v) {
    const float a = fabsf(v);
    if (!(a < 1048576.0f)) return EP24_FIXG_MARK;                // also NaN (the comparison is false)
    return (long long)__float2ll_rn(v * 68719476736.0f);
}
__device__ __forceinline__ bool fixg_bad(long long v) { return v >= EP24_FIXG_LIMIT || v <= -EP24_FIXG_LIMIT; }
__device__ __forceinline__ float from_fix_g(long long v) {
    if (fixg_bad(v)) return __builtin_nanf("");
    return (float)((double)v * (1.0 / 68719476736.0));
}

// exact x / d for 0 <= x < 2^31 with one mulhi + shift (divisor known at launch time)
struct FastDiv {
    unsigned mul, shr, d;
};
static inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f{0u, 0u, d};
    if (d != 1) {
        unsigned lg = 0;
        while ((1ull << lg) < d) ++lg;
        const unsigned p = 31 + lg;
        f.mul = (unsigned)(((1ull << p) + d - 1) / d);
        f.shr = p - 32;
    }
    return f;
}
__device__ __forceinline__ int fdiv(int x, const FastDiv& f) { return f.d == 1 ? x : (int)(__umulhi((unsigned)x, f.mul) >> f.shr); }

static inline int ep24_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
