// pk_probe.hip -- which packed-fp32 instruction returns wrong values beside another kernel's dense f16 MFMAs?
// Built as a tiny shared library (tools/probes/build_pk_probe.sh) and driven by tools/pk_probe.py: the victim kernels below
// run on one stream while the library's fp16 filter runs in guess mode on another.  The file is compiled WITHOUT packed-fp32
// code generation, so the only packed instructions in a victim are the ones written out in its inline assembly; every
// result is compared in place with the same arithmetic done by the unpacked instructions.
#include <hip/hip_runtime.h>
#include <cstdint>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__device__ __forceinline__ f2 packed(f2 a, f2 b, f2 c) {
    f2 d;
    if constexpr (OP == 0) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (OP == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (OP == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    else if constexpr (OP == 3) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(a));
    // the forms the log-mel kernel had and the sweeps do not: halves swapped / broadcast, negated, a scalar-register operand
    else if constexpr (OP == 4) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (OP == 5) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (OP == 6) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "s"(f2{0.70710678f, 0.92387953f}));
    else if constexpr (OP == 7) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));
    // ... and the two forms the sweeps do have besides the plain ones
    else if constexpr (OP == 8) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    else asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// the same arithmetic by the unpacked instructions, written out too (the compiler must not pack them back)
__device__ __forceinline__ float s_mul(float a, float b) { float d; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float s_add(float a, float b) { float d; asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float s_fma(float a, float b, float c) { float d; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
template <int OP>
__device__ __forceinline__ f2 plain(f2 a, f2 b, f2 c) {
    if constexpr (OP == 0) return f2{s_mul(a.x, b.x), s_mul(a.y, b.y)};
    else if constexpr (OP == 1) return f2{s_add(a.x, b.x), s_add(a.y, b.y)};
    else if constexpr (OP == 2) return f2{s_fma(a.x, b.x, c.x), s_fma(a.y, b.y, c.y)};
    else if constexpr (OP == 3) return a;
    else if constexpr (OP == 4) return f2{s_add(a.x, b.y), s_add(a.y, b.x)};
    else if constexpr (OP == 5) return f2{s_add(a.x, -b.y), s_add(a.x, -b.y)};
    else if constexpr (OP == 6) return f2{s_mul(a.x, 0.70710678f), s_mul(a.y, 0.92387953f)};
    else if constexpr (OP == 7) return f2{a.y, b.x};    // D.lo = S0.hi (op_sel[0] = 1), D.hi = S1.lo (op_sel[1] = 0)
    else if constexpr (OP == 8) return f2{s_add(a.x, -b.x), s_add(a.y, -b.y)};
    else return f2{s_fma(a.x, b.x, 0.0f), s_fma(a.y, b.y, 0.0f)};
}
// operands in [0.5, 1) from integer state (no floating-point instruction the compiler could pack)
__device__ __forceinline__ float unit(unsigned u) { return __uint_as_float(0x3f000000u | (u >> 9)); }

// out[0..3]: mismatches seen by lanes 0-15 / 16-31 / 32-47 / 48-63; out[4]: operations checked (per lane)
template <int OP, int NV, bool VIA_LDS = false>
__global__ void __launch_bounds__(256, 2) victim_kernel(int iters, unsigned long long* __restrict__ out) {
    extern __shared__ float lds[];   // (shapes the occupancy like the log-mel kernel's: two workgroups per CU; VIA_LDS: operand exchange)
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    // VIA_LDS: the packed instruction's operands are written to LDS by the NEIGHBOUR lane (lane ^ 1) and read back with 64- and
    // 128-bit reads right before use, the way the log-mel kernel's transposes hand values between the lanes of a frame
    f2* xch = reinterpret_cast<f2*>(lds) + (threadIdx.x & ~63u) * 4;
    const unsigned ln = threadIdx.x & 63u;
    unsigned u[NV], un[VIA_LDS ? NV : 1];
#pragma unroll
    for (int i = 0; i < NV; i++) u[i] = t * 2654435761u + (unsigned)i * 40503u + 12345u;
    if constexpr (VIA_LDS) {
#pragma unroll
        for (int i = 0; i < NV; i++) un[i] = (t ^ 1u) * 2654435761u + (unsigned)i * 40503u + 12345u;
    }
    unsigned bad = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const unsigned u0 = u[i], u1 = u0 * 1664525u + 1013904223u, u2 = u1 * 1664525u + 1013904223u;
            const f2 a = {unit(u0), unit(u1)}, b = {unit(u2), unit(u0 ^ u2)}, c = {unit((unsigned)it * 2246822519u + (unsigned)i), unit((unsigned)it * 3266489917u + (unsigned)i)};
            f2 pa = a, pb = b;
            if constexpr (VIA_LDS) {
                // my neighbour's operands are what I compute its reference from; it reads mine back from LDS
                const unsigned v0 = u0 ^ 0u;   // (this lane's own state is the neighbour's "other lane" below)
                (void)v0;
                xch[(ln ^ 1u) * 2 + 0] = a;    // slot of lane (ln ^ 1) holds what lane ln wrote
                xch[(ln ^ 1u) * 2 + 1] = b;
                __builtin_amdgcn_wave_barrier();
                typedef float f4 __attribute__((ext_vector_type(4)));
                const f4 both = *reinterpret_cast<const f4*>(&xch[ln * 2]);   // one 128-bit read: (a, b) of lane ln ^ 1
                pa = f2{both.x, both.y};
                pb = f2{both.z, both.w};
                __builtin_amdgcn_wave_barrier();
            }
            const f2 p = packed<OP>(pa, pb, c);
            f2 ra = a, rb = b;
            if constexpr (VIA_LDS) {   // the reference: the neighbour's operands recomputed from its integer state, never through LDS
                const unsigned tn = t ^ 1u;
                unsigned w0 = tn * 2654435761u + (unsigned)i * 40503u + 12345u;
                // the neighbour's state after `it` iterations of three LCG steps each: kept in step in un[] below
                w0 = un[i];
                const unsigned w1 = w0 * 1664525u + 1013904223u, w2 = w1 * 1664525u + 1013904223u;
                ra = f2{unit(w0), unit(w1)};
                rb = f2{unit(w2), unit(w0 ^ w2)};
            }
            const f2 q = plain<OP>(ra, rb, c);
            bad += (__float_as_uint(p.x) != __float_as_uint(q.x)) + (__float_as_uint(p.y) != __float_as_uint(q.y));
            u[i] = u2 * 1664525u + 1013904223u;
            if constexpr (VIA_LDS) {
                const unsigned w1 = un[i] * 1664525u + 1013904223u, w2 = w1 * 1664525u + 1013904223u;
                un[i] = w2 * 1664525u + 1013904223u;
            }
        }
    }
    if (threadIdx.x == 0xffff) lds[0] = unit(u[0]);   // (never: keeps the dynamic LDS allocation referenced)
    if (bad) atomicAdd(&out[(threadIdx.x & 63) >> 4], (unsigned long long)bad);
    if (t == 0) out[4] = (unsigned long long)iters * NV * 2;
}

template <int OP>
static int launch(int iters, int vregs, size_t lds, void* out, hipStream_t s) {
    hipError_t e;
    if (vregs >= 100) {   // operands through LDS (needs >= 16 KB of dynamic LDS per workgroup)
        if (lds < 16384) lds = 16384;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&victim_kernel<OP, 6, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        victim_kernel<OP, 6, true><<<512, 256, lds, s>>>(iters, static_cast<unsigned long long*>(out));
    } else if (vregs >= 24) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&victim_kernel<OP, 24>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        victim_kernel<OP, 24><<<512, 256, lds, s>>>(iters, static_cast<unsigned long long*>(out));
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&victim_kernel<OP, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        victim_kernel<OP, 6><<<512, 256, lds, s>>>(iters, static_cast<unsigned long long*>(out));
    }
    return (int)hipGetLastError();
}

// op: 0 v_pk_mul_f32, 1 v_pk_add_f32, 2 v_pk_fma_f32, 3 v_pk_mov_b32; vregs: 6 or 24 operand triples per lane (36 / 144 VGPRs of
// live data); lds: dynamic LDS bytes per workgroup; out: device buffer of five 64-bit words, zeroed by the caller
extern "C" int pk_probe_launch(int op, int iters, int vregs, long lds, void* out, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (op) {
        case 0: return launch<0>(iters, vregs, (size_t)lds, out, s);
        case 1: return launch<1>(iters, vregs, (size_t)lds, out, s);
        case 2: return launch<2>(iters, vregs, (size_t)lds, out, s);
        case 3: return launch<3>(iters, vregs, (size_t)lds, out, s);
        case 4: return launch<4>(iters, vregs, (size_t)lds, out, s);
        case 5: return launch<5>(iters, vregs, (size_t)lds, out, s);
        case 6: return launch<6>(iters, vregs, (size_t)lds, out, s);
        case 7: return launch<7>(iters, vregs, (size_t)lds, out, s);
        case 8: return launch<8>(iters, vregs, (size_t)lds, out, s);
        default: return launch<9>(iters, vregs, (size_t)lds, out, s);
    }
}
