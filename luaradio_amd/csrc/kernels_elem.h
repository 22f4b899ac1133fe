// kernels_elem.h - the HBM-bound element-wise blocks: frequency translator, FM discriminator, downsampler.
// All are streaming kernels: coalesced 8/16-B-per-lane loads, grid-stride, no LDS.
#pragma once
#include "common.h"
#include "pk_math.h"

namespace lrhip {

// ------------------------------------------------------------------------------------------------
// FrequencyTranslatorBlock  (reference: radio/blocks/signal/frequencytranslator.lua:93-110)
//   y[n] = x[n] * exp(j*omega*n)
// The reference carries a running phase (double accumulator in Lua, f32 phasor in VOLK); here the phase of
// sample n is closed-form from a 64-bit sample counter, so any chunking / any launch split gives the same
// values and there is no drift:   turns(n) = frac(n * omega/2pi)  computed EXACTLY in 0.64 fixed point
// (unsigned 64-bit wrap-around multiply == mod 1 turn).
// Algorithmic traffic: 16 B/sample (8 in + 8 out).
// ------------------------------------------------------------------------------------------------
// exp(j*2*pi*turns) from the exact 0.64 fixed-point turn count.  The quadrant comes straight from the top bits
// (after a 1/8-turn bias), the residual angle phi in [-pi/4, pi/4) keeps 32 fraction bits, and sin/cos of phi are
// the classic single-precision minimax kernels (|err| ~1e-7): ~25 VALU ops instead of libm's sincosf with its
// general argument reduction.  Deterministic, so the fused (tuner) and standalone rotators agree bit for bit.
__device__ __forceinline__ void phasor_from_turns(uint64_t turns_fx, float &c, float &s)
{
    const uint64_t t = turns_fx + (1ull << 61);                  // + 1/8 turn
    const unsigned q = (unsigned)(t >> 62);                       // quadrant 0..3
    const int ri = (int)((unsigned)((t << 2) >> 32) ^ 0x80000000u);   // (rho - 1/2) in 1.31 signed fixed point
    const float phi = (float)ri * 3.6572952e-10f;                 // * (pi/2) * 2^-32
    const float z = phi * phi;
    float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * phi, phi);
    float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
                    fmaf(-0.5f, z, 1.0f));
    // rotate by q * pi/2
    const float cs = (q & 1) ? -sp : cp;
    const float sn = (q & 1) ? cp : sp;
    c = (q & 2) ? -cs : cs;
    s = (q & 2) ? -sn : sn;
}

// The phasor of ABSOLUTE sample index n is defined once, for every kernel that rotates (standalone, fused into the FIR
// staging, edge tiles), so that fused == unfused and any chunking give the same bits:
//     phasor(n) = P(n & ~7) * W[n & 7],   P(m) = polynomial at m * step,   W[k] = polynomial at k * step  (W[0] = 1 exactly)
// One polynomial serves an aligned block of 8 samples, the other 7 cost one packed complex multiply each - the rotator is
// VALU work that the f32 matrix pipe cannot hide.
struct RotTab { cf w[8]; };

__device__ __forceinline__ cf phasor_poly(uint64_t turns_fx)
{
    float c, s;
    phasor_from_turns(turns_fx, c, s);
    return cf{c, s};
}

__device__ __forceinline__ RotTab rot_tab(uint64_t step_fx)
{
    RotTab t;
    t.w[0] = cf{1.f, 0.f};
#pragma unroll
    for (int k = 1; k < 8; k++) t.w[k] = phasor_poly(step_fx * (uint64_t)k);
    return t;
}

// per-sample paths (edge tiles, unaligned vectors): two polynomials, same bits as the table form
__device__ __forceinline__ cf phasor_of(uint64_t step_fx, uint64_t n)
{
    const uint64_t k = n & 7;
    return cmul(phasor_poly(step_fx * (n - k)), k ? phasor_poly(step_fx * k) : cf{1.f, 0.f});
}

__device__ __forceinline__ float2 rotate_sample(float2 v, uint64_t step_fx, uint64_t n)
{
    return cf_to(cmul(cf_from(v), phasor_of(step_fx, n)));
}

// two consecutive samples (one 16-B access) whose first has absolute index cnt
__device__ __forceinline__ float4 rotate_pair(float4 v, uint64_t step_fx, uint64_t cnt, const RotTab &t)
{
    if (cnt & 1) {                    // wave-uniform in every caller (cnt = base + 2 * i); rare: odd chunk offsets
        float2 a = rotate_sample(make_float2(v.x, v.y), step_fx, cnt), b = rotate_sample(make_float2(v.z, v.w), step_fx, cnt + 1);
        return make_float4(a.x, a.y, b.x, b.y);
    }
    const cf p = phasor_poly(step_fx * (cnt & ~7ull));
    cf wa, wb;
    switch ((int)(cnt & 7)) {         // constant register indices in every arm (a dynamic index would put the table in scratch)
        case 0: wa = t.w[0]; wb = t.w[1]; break;
        case 2: wa = t.w[2]; wb = t.w[3]; break;
        case 4: wa = t.w[4]; wb = t.w[5]; break;
        default: wa = t.w[6]; wb = t.w[7]; break;
    }
    cf a = cmul(cf{v.x, v.y}, cmul(p, wa)), b = cmul(cf{v.z, v.w}, cmul(p, wb));
    return make_float4(a.x, a.y, b.x, b.y);
}

// the float4 j (0..3) of an ALIGNED block of 8 samples whose phasor base p = P(block start) is already known
template <int J>
__device__ __forceinline__ float4 rotate_in_block(float4 v, cf p, const RotTab &t)
{
    cf a = cmul(cf{v.x, v.y}, cmul(p, t.w[2 * J])), b = cmul(cf{v.z, v.w}, cmul(p, t.w[2 * J + 1]));
    return make_float4(a.x, a.y, b.x, b.y);
}

// VEC = 2: two samples (one 16-B access) per lane when both pointers are 16-B aligned; VEC = 1 otherwise.
template <int VEC>
__global__ __launch_bounds__(256) void rotator_kernel(const float2 *__restrict__ x, float2 *__restrict__ y,
                                                      unsigned long n, uint64_t step_fx, uint64_t count0)
{
    unsigned long stride = (unsigned long)gridDim.x * blockDim.x;
    if (VEC == 2) {
        const RotTab tab = rot_tab(step_fx);
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        float4 *y4 = reinterpret_cast<float4 *>(y);
        unsigned long n2 = n / 2;
        // two 16-byte items per thread, both loads in flight before the first phasor is evaluated (the grid covers n2 / 2 items)
        for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += 2 * stride) {
            const unsigned long j = i + stride;
            const float4 a = x4[i], b = j < n2 ? x4[j] : a;
            nt_store(y4 + i, rotate_pair(a, step_fx, count0 + 2 * i, tab));
            if (j < n2) nt_store(y4 + j, rotate_pair(b, step_fx, count0 + 2 * j, tab));
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = rotate_sample(x[n - 1], step_fx, count0 + n - 1);
    } else {
        for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
            y[i] = rotate_sample(x[i], step_fx, count0 + i);
    }
}

// ------------------------------------------------------------------------------------------------
// FrequencyDiscriminatorBlock  (reference: radio/blocks/signal/frequencydiscriminator.lua:68-88)
//   tmp[n] = x[n] * conj(x[n-1]);  y[n] = atan2(Im tmp, Re tmp) * (1/gain)
// x[-1] is the carried previous sample (prev_in), zero initially (:34).  The kernel also publishes the last
// sample of the chunk to prev_out (ping-pong, so no block reads what another writes).
// Algorithmic traffic: 12 B/sample (8 in + 4 out); the x[n-1] re-read hits L1/L2.
// ------------------------------------------------------------------------------------------------
// atan2 for finite inputs: the Cephes atanf split at tan(pi/8) with its degree-9 odd minimax polynomial, then the octant/quadrant fix-ups.  The split
// is decided on the operands (mn > tan(pi/8) mx) and both ranges share ONE hardware reciprocal: t = mn / mx or (mn - mx) / (mn + mx).
// |error| <= 3.0e-7 rad (Float32 emulation with a 1-ulp reciprocal against a double arctan2 over 2e6 points, magnitudes 1e-12 .. 1e6); atan2(0, 0) = 0
// like libm.  ~30 VALU ops; hipcc expands __fdividef to the full IEEE division sequence (10 instructions), which made the two-division form of
// round 1 twice as long.  v_rcp_f32 flushes denormal inputs: the caller routes max(|x|, |y|) < 2^-60 to fast_atan2f_tiny.
__device__ __forceinline__ float fast_atan2f(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const bool big = mn > 0.41421356237f * mx;
    const float num = big ? mn - mx : mn, den = big ? mn + mx : mx;
    const float t = mx > 0.f ? num * __builtin_amdgcn_rcpf(den) : 0.f;
    const float z = t * t;
    float p = fmaf(fmaf(fmaf(fmaf(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f) * z, t, t);
    float r = big ? p + 0.785398163397448f : p;
    r = ay > ax ? 1.5707963267948966f - r : r;
    r = x < 0.f ? 3.14159265358979f - r : r;
    return copysignf(r, y);
}
// operands below 2^-60 (down to denormals): rescaled by 2^80 first - the angle does not change
__device__ __noinline__ float fast_atan2f_tiny(float y, float x) { return fast_atan2f(y * 0x1p+80f, x * 0x1p+80f); }

__device__ __forceinline__ float discriminate(float2 a, float2 b, double inv_gain)
{
    // a * conj(b); f32 fused products: the residual (<= 1 ulp of the larger product) moves the angle by < 1e-7 rad
    float tr = fmaf(a.x, b.x, a.y * b.y), ti = fmaf(a.y, b.x, -a.x * b.y);
    if (__builtin_expect(fmaxf(fabsf(tr), fabsf(ti)) < 0x1p-60f, 0)) {
        if (tr == 0.f && ti == 0.f) {
            // zero product (first sample after the zero initial state, or an exactly silent input): the reference's angle is
            // then decided by the SIGNS of the zeros, atan2(+-0, -0) = +-pi (frequencydiscriminator.lua:74 -> complexfloat32.lua:79-81
            // operation order: re = ar*br - ai*(-bi), im = ar*(-bi) + ai*br)
            float nb = -b.y;
            float zr = __fsub_rn(__fmul_rn(a.x, b.x), __fmul_rn(a.y, nb)), zi = __fadd_rn(__fmul_rn(a.x, nb), __fmul_rn(a.y, b.x));
            float r = __builtin_signbit(zr) ? 3.14159265358979f : 0.f;
            return copysignf(r, zi) * (float)inv_gain;
        }
        return fast_atan2f_tiny(ti, tr) * (float)inv_gain;
    }
    return fast_atan2f(ti, tr) * (float)inv_gain;
}

__global__ __launch_bounds__(256) void fmdiscrim_kernel(const float2 *__restrict__ x, float *__restrict__ y,
                                                        unsigned long n, double inv_gain,
                                                        const float2 *__restrict__ prev_in, float2 *__restrict__ prev_out)
{
    unsigned long stride = (unsigned long)gridDim.x * blockDim.x;
    for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float2 a = x[i];
        float2 b = i ? x[i - 1] : *prev_in;
        y[i] = discriminate(a, b, inv_gain);
        if (i == n - 1) *prev_out = a;
    }
}

// 16 B in / 8 B out per lane, one item per thread (one-shot grid, common.h grid_for): samples 2i and 2i+1; the sample before
// them is one more 8-B load from the lines the neighbouring lane fetches anyway.  Needs 16-B aligned x and 8-B aligned y.
__global__ __launch_bounds__(256) void fmdiscrim_vec2_kernel(const float2 *__restrict__ x, float *__restrict__ y,
                                                             unsigned long n, double inv_gain,
                                                             const float2 *__restrict__ prev_in, float2 *__restrict__ prev_out)
{
    const unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x, n2 = n / 2;
    if (i < n2) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        const float2 a = make_float2(v.x, v.y), b = make_float2(v.z, v.w);
        const float2 p = i ? x[2 * i - 1] : *prev_in;
        reinterpret_cast<float2 *>(y)[i] = make_float2(discriminate(a, p, inv_gain), discriminate(b, a, inv_gain));
        if (2 * i + 2 == n) *prev_out = b;
    }
    if ((n & 1) && i == n2) {        // odd tail sample
        const float2 a = x[n - 1], p = n > 1 ? x[n - 2] : *prev_in;
        y[n - 1] = discriminate(a, p, inv_gain);
        *prev_out = a;
    }
}

// 32 B in / 16 B out per lane: samples 4i .. 4i+3 (two 16-byte loads) and one 16-byte store of their four angles - the 8-byte stores of the
// vec2 form are the narrow side of that kernel.  Needs 16-B aligned x and y; the n % 4 tail samples go to the last thread.
__global__ __launch_bounds__(256) void fmdiscrim_vec4_kernel(const float2 *__restrict__ x, float *__restrict__ y,
                                                             unsigned long n, double inv_gain,
                                                             const float2 *__restrict__ prev_in, float2 *__restrict__ prev_out)
{
    const unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x, n4 = n / 4;
    if (i < n4) {
        const float4 v0 = reinterpret_cast<const float4 *>(x)[2 * i], v1 = reinterpret_cast<const float4 *>(x)[2 * i + 1];
        const float2 p = i ? x[4 * i - 1] : *prev_in;
        const float2 a = make_float2(v0.x, v0.y), b = make_float2(v0.z, v0.w), c = make_float2(v1.x, v1.y), d = make_float2(v1.z, v1.w);
        nt_store(reinterpret_cast<float4 *>(y) + i, make_float4(discriminate(a, p, inv_gain), discriminate(b, a, inv_gain), discriminate(c, b, inv_gain),
                                                                discriminate(d, c, inv_gain)));
        if (4 * i + 4 == n) *prev_out = d;
    }
    if (i == n4 && (n & 3)) {        // tail samples
        float2 p = n4 ? x[4 * n4 - 1] : *prev_in;
        for (unsigned long k = 4 * n4; k < n; k++) {
            const float2 a = x[k];
            y[k] = discriminate(a, p, inv_gain);
            p = a;
        }
        *prev_out = p;
    }
}

// ------------------------------------------------------------------------------------------------
// DownsamplerBlock  (reference: radio/blocks/signal/downsampler.lua:45-56):  y[m] = x[index + m*factor].
// Bit-exact copies.  Traffic: E/factor out + between E/factor and E in (sector granularity).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void downsample_kernel(const T *__restrict__ x, T *__restrict__ y,
                                                         unsigned long n_out, unsigned long index, unsigned long factor)
{
    unsigned long stride = (unsigned long)gridDim.x * blockDim.x;
    for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += stride)
        y[i] = x[index + i * factor];
}


// ------------------------------------------------------------------------------------------------
// IQFileSource / RealFileSource sample-format conversion
// (reference: radio/blocks/sources/iqfile.lua:99-113, realfile.lua:99-110, radio/utilities/format_utils.lua:82-97):
//   out = (raw.value - offset) / scale   after an optional byte swap, evaluated in double and rounded once.
// Doing it on the device lets an RTL-SDR style u8 file cross PCIe at 2 B per complex sample instead of 8.
// One thread converts one scalar (the interleaved I/Q stream of a complex file is 2n scalars).
// Traffic: sizeof(T) in + 4 out per scalar.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T byteswap_raw(T v)
{
    if constexpr (sizeof(T) == 2) return (T)__builtin_bswap16((uint16_t)v);
    else if constexpr (sizeof(T) == 4) return (T)__builtin_bswap32((uint32_t)v);
    else if constexpr (sizeof(T) == 8) return (T)__builtin_bswap64((uint64_t)v);
    else return v;
}

// RAW = storage integer type, VAL = the value type it is reinterpreted as (int8_t ... double)
template <typename RAW, typename VAL, bool SWAP>
__global__ __launch_bounds__(256) void format_convert_kernel(const RAW *__restrict__ in, float *__restrict__ out, unsigned long n,
                                                             double offset, double scale)
{
    unsigned long stride = (unsigned long)gridDim.x * blockDim.x;
    for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        RAW r = in[i];
        if (SWAP) r = byteswap_raw(r);
        VAL v;
        __builtin_memcpy(&v, &r, sizeof(v));
        out[i] = (float)(((double)v - offset) / scale);
    }
}


// The same conversion, four raw scalars per thread: one 4- / 8- / 16-byte load (two for the 8-byte formats) and one 16-byte non-temporal store instead of
// four 1- / 2- / 4-byte loads and four 4-byte stores.  The 8-bit formats (RTL-SDR records: the case SURVEY 8(f) ranks first) take their 256 possible
// results from a table every workgroup fills in LDS with the same double-precision expression, so no thread divides in the stream.
// `in` aligned to 4 sizeof(RAW) (8-byte formats: 16), `out` to 16 bytes; `nitems` = n / 4 threads and a spare one for the n % 4 tail.
template <typename RAW, typename VAL, bool SWAP>
__global__ __launch_bounds__(256) void format_convert_vec_kernel(const RAW *__restrict__ in, float *__restrict__ out, unsigned long nitems, unsigned long n,
                                                                 double offset, double scale)
{
    __shared__ float lut[sizeof(RAW) == 1 ? 256 : 1];
    auto conv = [&](RAW r) {
        if (SWAP) r = byteswap_raw(r);
        VAL v;
        __builtin_memcpy(&v, &r, sizeof(v));
        return (float)(((double)v - offset) / scale);
    };
    if (sizeof(RAW) == 1) {
        lut[threadIdx.x] = conv((RAW)threadIdx.x);
        __syncthreads();
    }
    const unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == nitems)
        for (unsigned long k = 4 * nitems; k < n; k++) out[k] = conv(in[k]);
    if (i >= nitems) return;
    RAW r[4];
    if (sizeof(RAW) == 1) { const uint32_t w = reinterpret_cast<const uint32_t *>(in)[i]; __builtin_memcpy(r, &w, 4); }
    else if (sizeof(RAW) == 2) { const uint2 w = reinterpret_cast<const uint2 *>(in)[i]; __builtin_memcpy(r, &w, 8); }
    else if (sizeof(RAW) == 4) { const uint4 w = reinterpret_cast<const uint4 *>(in)[i]; __builtin_memcpy(r, &w, 16); }
    else {
        const uint4 w0 = reinterpret_cast<const uint4 *>(in)[2 * i], w1 = reinterpret_cast<const uint4 *>(in)[2 * i + 1];
        __builtin_memcpy(r, &w0, 16);
        __builtin_memcpy(reinterpret_cast<char *>(r) + 16, &w1, 16);
    }
    float4 o;
    if (sizeof(RAW) == 1) o = make_float4(lut[(uint8_t)r[0]], lut[(uint8_t)r[1]], lut[(uint8_t)r[2]], lut[(uint8_t)r[3]]);
    else o = make_float4(conv(r[0]), conv(r[1]), conv(r[2]), conv(r[3]));
    nt_store(reinterpret_cast<float4 *>(out) + i, o);
}


// IQFileSink / RealFileSink direction (radio/blocks/sinks/iqfile.lua:68-85, realfile.lua): raw.value = x*scale + offset
// evaluated in double and stored into the raw type by the C conversion LuaJIT applies to a cdata assignment
// (truncation toward zero for the integer formats), then the byte swap.
template <typename RAW, typename VAL, bool SWAP>
__global__ __launch_bounds__(256) void format_pack_kernel(const float *__restrict__ in, RAW *__restrict__ out, unsigned long n,
                                                          double offset, double scale)
{
    unsigned long stride = (unsigned long)gridDim.x * blockDim.x;
    for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        VAL v = (VAL)((double)in[i] * scale + offset);
        RAW r;
        __builtin_memcpy(&r, &v, sizeof(r));
        if (SWAP) r = byteswap_raw(r);
        out[i] = r;
    }
}

// ------------------------------------------------------------------------------------------------
// Two-input element-wise blocks: MultiplyBlock (radio/blocks/signal/multiply.lua:43-76), MultiplyConjugateBlock
// (multiplyconjugate.lua:41-59), AddBlock (add.lua), SubtractBlock (subtract.lua).  Complex products follow the
// Lua arithmetic: each component evaluated in double and rounded once (radio/types/complexfloat32.lua:79-81).
// Traffic: 3 x sample size per sample.
// ------------------------------------------------------------------------------------------------
enum { BIN_MULTIPLY = 0, BIN_MULTIPLY_CONJ = 1, BIN_ADD = 2, BIN_SUBTRACT = 3, BIN_F2C = 4 };

template <int OP>
__global__ __launch_bounds__(256) void binary_complex_kernel(const float2 *__restrict__ a, const float2 *__restrict__ b,
                                                             float2 *__restrict__ y, unsigned long n)
{
    unsigned long stride = (unsigned long)gridDim.x * blockDim.x;
    for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float2 p = a[i], q = b[i], o;
        if (OP == BIN_ADD) o = make_float2(p.x + q.x, p.y + q.y);
        else if (OP == BIN_SUBTRACT) o = make_float2(p.x - q.x, p.y - q.y);
        else {
            double ar = p.x, ai = p.y, br = q.x, bi = OP == BIN_MULTIPLY_CONJ ? -(double)q.y : (double)q.y;
            o = make_float2((float)(ar * br - ai * bi), (float)(ar * bi + ai * br));
        }
        y[i] = o;
    }
}

// 16 B per lane (two ComplexFloat32 or four Float32 per thread); `nf` = number of floats, pointers 16-B aligned.
// CPLX_MUL: 0 = element-wise on floats (add / subtract of either type, real multiply), 1 = complex multiply, 2 = multiply by conjugate.
template <int OP, int CPLX_MUL>
__global__ __launch_bounds__(256) void binary_vec4_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ y, unsigned long nf)
{
    const unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x, n4 = nf / 4;
    if (i < n4) {
        const float4 p = reinterpret_cast<const float4 *>(a)[i], q = reinterpret_cast<const float4 *>(b)[i];
        float4 o;
        if (CPLX_MUL) {
            const double s = CPLX_MUL == 2 ? -1.0 : 1.0;
            const double ar0 = p.x, ai0 = p.y, br0 = q.x, bi0 = s * (double)q.y, ar1 = p.z, ai1 = p.w, br1 = q.z, bi1 = s * (double)q.w;
            o = make_float4((float)(ar0 * br0 - ai0 * bi0), (float)(ar0 * bi0 + ai0 * br0), (float)(ar1 * br1 - ai1 * bi1), (float)(ar1 * bi1 + ai1 * br1));
        } else if (OP == BIN_ADD) o = make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
        else if (OP == BIN_SUBTRACT) o = make_float4(p.x - q.x, p.y - q.y, p.z - q.z, p.w - q.w);
        else o = make_float4(p.x * q.x, p.y * q.y, p.z * q.z, p.w * q.w);
        nt_store(reinterpret_cast<float4 *>(y) + i, o);
    }
}

template <int OP>
__global__ __launch_bounds__(256) void binary_real_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                          float *__restrict__ y, unsigned long n)
{
    unsigned long stride = (unsigned long)gridDim.x * blockDim.x;
    for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float p = a[i], q = b[i];
        y[i] = OP == BIN_ADD ? p + q : OP == BIN_SUBTRACT ? p - q : p * q;
    }
}


// ------------------------------------------------------------------------------------------------
// MultiplyConstantBlock (radio/blocks/signal/multiplyconstant.lua:52-71) and UpsamplerBlock
// (radio/blocks/signal/upsampler.lua:45-53, zero-stuffing: y[i*L] = x[i], 0 elsewhere) - the two small blocks the
// Interpolator / RationalResampler composites add around the FIR (radio/composites/interpolator.lua:31-34).
// MODE 0: real x real constant, 1: complex x real constant (scalar_mul), 2: complex x complex constant.
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void multiply_constant_kernel(const float *__restrict__ x, float *__restrict__ y, unsigned long n,
                                                                float cr, float ci)
{
    unsigned long stride = (unsigned long)gridDim.x * blockDim.x;
    for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (MODE == 0) {
            y[i] = x[i] * cr;
        } else {
            float2 v = reinterpret_cast<const float2 *>(x)[i], o;
            if (MODE == 1) o = make_float2(v.x * cr, v.y * cr);
            else {
                double ar = v.x, ai = v.y;
                o = make_float2((float)(ar * (double)cr - ai * (double)ci), (float)(ar * (double)ci + ai * (double)cr));
            }
            reinterpret_cast<float2 *>(y)[i] = o;
        }
    }
}

// 16 B per lane, one item per thread.  MODE 0 / 1 are the same operation on the float stream (x * cr per scalar); MODE 2 =
// two ComplexFloat32 times the complex constant.  `nf` = number of floats (the caller handles nf % 4 with the scalar kernel).
template <int MODE>
__global__ __launch_bounds__(256) void multiply_constant_vec4_kernel(const float *__restrict__ x, float *__restrict__ y, unsigned long nf, float cr, float ci)
{
    const unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nf / 4) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        float4 o;
        if (MODE != 2) o = make_float4(v.x * cr, v.y * cr, v.z * cr, v.w * cr);
        else {
            const double c = cr, d = ci, ar = v.x, ai = v.y, br = v.z, bi = v.w;
            o = make_float4((float)(ar * c - ai * d), (float)(ar * d + ai * c), (float)(br * c - bi * d), (float)(br * d + bi * c));
        }
        nt_store(reinterpret_cast<float4 *>(y) + i, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample_kernel(const T *__restrict__ x, T *__restrict__ y, unsigned long n_out, unsigned long factor)
{
    unsigned long stride = (unsigned long)gridDim.x * blockDim.x;
    for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += stride) {
        unsigned long q = i / factor;
        T v = {};
        if (q * factor == i) v = x[q];
        y[i] = v;
    }
}


// ------------------------------------------------------------------------------------------------
// One-input element-wise blocks (radio/blocks/signal/{complexmagnitude,complexphase,complextoreal,complextoimag,
// complexconjugate,realtocomplex,absolutevalue,addconstant}.lua) - one sample per thread, HBM bound.
// ------------------------------------------------------------------------------------------------
enum { UN_CMAG = 0, UN_CPHASE = 1, UN_CREAL = 2, UN_CIMAG = 3, UN_CCONJ = 4, UN_R2C = 5, UN_ABS = 6,
       UN_ADDC_REAL = 7, UN_ADDC_CPLX_BY_REAL = 8, UN_ADDC_CPLX = 9 };

template <int OP>
__device__ __forceinline__ void unary_sample(const float *__restrict__ x, float *__restrict__ y, unsigned long i, float cr, float ci)
{
    const float2 *xc = reinterpret_cast<const float2 *>(x);
    float2 *yc = reinterpret_cast<float2 *>(y);
    if (OP == UN_CMAG) { float2 v = xc[i]; y[i] = sqrtf((float)((double)v.x * v.x + (double)v.y * v.y)); }   // complexfloat32.lua:163-165
    else if (OP == UN_CPHASE) { float2 v = xc[i]; y[i] = atan2f(v.y, v.x); }                                   // :152-154
    else if (OP == UN_CREAL) y[i] = xc[i].x;
    else if (OP == UN_CIMAG) y[i] = xc[i].y;
    else if (OP == UN_CCONJ) { float2 v = xc[i]; yc[i] = make_float2(v.x, -v.y); }
    else if (OP == UN_R2C) yc[i] = make_float2(x[i], 0.f);
    else if (OP == UN_ABS) y[i] = fabsf(x[i]);
    else if (OP == UN_ADDC_REAL) y[i] = x[i] + cr;
    else if (OP == UN_ADDC_CPLX_BY_REAL) { float2 v = xc[i]; yc[i] = make_float2(v.x + cr, v.y); }             // addconstant.lua:66-72
    else { float2 v = xc[i]; yc[i] = make_float2(v.x + cr, v.y + ci); }
}
template <int OP>
__global__ __launch_bounds__(256) void unary_kernel(const float *__restrict__ x, float *__restrict__ y, unsigned long n, float cr, float ci)
{
    unsigned long stride = (unsigned long)gridDim.x * blockDim.x;
    for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) unary_sample<OP>(x, y, i, cr, ci);
}

// The same operations with 16-byte accesses on the wider side (round 3: the one-sample-per-thread form above ran the reference suite's entries at
// 3.7-5.7 TB/s where the 16-byte kernels stream at 6.1-6.4).  One thread = one float4 of output for the ComplexFloat32 -> Float32 operations (four
// samples, two 16-byte loads), one float4 in and out for the same-type operations, two samples (an 8-byte load, one 16-byte store) for RealToComplex.
// `nitems` = number of such threads; thread `nitems` (the grid has one to spare) takes the up to three samples behind them.  Pointers 16-byte aligned.
// samples one unary_vec_kernel thread covers
__host__ __device__ constexpr int unary_vec_samples(int op)
{
    return (op == UN_CMAG || op == UN_CPHASE || op == UN_CREAL || op == UN_CIMAG || op == UN_ABS || op == UN_ADDC_REAL) ? 4 : 2;
}
template <int OP>
__device__ __forceinline__ float unary_c2r(float re, float im)
{
    if (OP == UN_CMAG) return sqrtf((float)((double)re * re + (double)im * im));
    if (OP == UN_CPHASE) return atan2f(im, re);
    return OP == UN_CREAL ? re : im;
}
template <int OP>
__global__ __launch_bounds__(256) void unary_vec_kernel(const float *__restrict__ x, float *__restrict__ y, unsigned long nitems, unsigned long n, float cr, float ci)
{
    const unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == nitems)
        for (unsigned long k = nitems * unary_vec_samples(OP); k < n; k++) unary_sample<OP>(x, y, k, cr, ci);
    if (i >= nitems) return;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    float4 *y4 = reinterpret_cast<float4 *>(y);
    if (OP == UN_CMAG || OP == UN_CPHASE || OP == UN_CREAL || OP == UN_CIMAG) {
        const float4 a = x4[2 * i], b = x4[2 * i + 1];
        nt_store(y4 + i, make_float4(unary_c2r<OP>(a.x, a.y), unary_c2r<OP>(a.z, a.w), unary_c2r<OP>(b.x, b.y), unary_c2r<OP>(b.z, b.w)));
    } else if (OP == UN_R2C) {
        const float2 v = reinterpret_cast<const float2 *>(x)[i];
        nt_store(y4 + i, make_float4(v.x, 0.f, v.y, 0.f));
    } else {
        const float4 v = x4[i];
        float4 o;
        if (OP == UN_CCONJ) o = make_float4(v.x, -v.y, v.z, -v.w);
        else if (OP == UN_ABS) o = make_float4(fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w));
        else if (OP == UN_ADDC_REAL) o = make_float4(v.x + cr, v.y + cr, v.z + cr, v.w + cr);
        else if (OP == UN_ADDC_CPLX_BY_REAL) o = make_float4(v.x + cr, v.y, v.z + cr, v.w);
        else o = make_float4(v.x + cr, v.y + ci, v.z + cr, v.w + ci);
        nt_store(y4 + i, o);
    }
}

// UpsamplerBlock, 16-byte stores (PER = samples per store: 2 ComplexFloat32 or 4 Float32), UPS_U of them per thread, 256 stores apart (consecutive lanes,
// consecutive 16 bytes in every instruction): ONE 64-bit division per thread, the position inside the zero-stuffing period is carried from store to store
// and from sample to sample; all loads are issued before the first store.  (Rounds 2-3: one store and one division per thread - Upsampler(5) on 2^26
// ComplexFloat32 samples 0.685 ms where the same bytes with no arithmetic take 0.605, tools/mb_rw15.hip.)  y 16-byte aligned; `nitems` whole stores, the
// thread that would own store number `nitems` takes the samples behind them.
constexpr int UPS_U = 4;
template <typename T, int PER>
__global__ __launch_bounds__(256) void upsample_vec_kernel(const T *__restrict__ x, float4 *__restrict__ y, unsigned long nitems, unsigned long factor, unsigned long n_out)
{
    const unsigned long i0 = (unsigned long)blockIdx.x * (256 * UPS_U) + threadIdx.x;
    if (i0 > nitems) return;
    unsigned long q = (i0 * PER) / factor, r = i0 * PER - q * factor;
    const unsigned long dq = (256ul * PER) / factor, dr = 256ul * PER - dq * factor;      // 256 stores further on
    float4 w[UPS_U];
#pragma unroll
    for (int j = 0; j < UPS_U; j++) {
        const unsigned long i = i0 + 256ul * j;
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        if (i < nitems) {
            unsigned long qq = q, rr = r;
#pragma unroll
            for (int k = 0; k < PER; k++) {
                if (rr == 0) {
                    const T v = x[qq];
                    __builtin_memcpy(&e[k * (4 / PER)], &v, sizeof(T));
                }
                if (++rr == factor) { rr = 0; qq++; }
            }
        }
        w[j] = make_float4(e[0], e[1], e[2], e[3]);
        q += dq; r += dr;
        if (r >= factor) { r -= factor; q++; }
    }
#pragma unroll
    for (int j = 0; j < UPS_U; j++) {
        const unsigned long i = i0 + 256ul * j;
        if (i < nitems) nt_store(y + i, w[j]);
        else if (i == nitems)                          // the samples behind the last whole store
            for (unsigned long o = nitems * PER; o < n_out; o++) {
                T v = {};
                if (o % factor == 0) v = x[o / factor];
                reinterpret_cast<T *>(y)[o] = v;
            }
    }
}

// FloatToComplexBlock, two samples per thread: 8-byte loads, one 16-byte store.  Pointers 8- / 16-byte aligned, `nitems` = n / 2 (+ a spare thread for an odd n).
__global__ __launch_bounds__(256) void float_to_complex_vec_kernel(const float2 *__restrict__ a, const float2 *__restrict__ b, float4 *__restrict__ y,
                                                                   unsigned long nitems, unsigned long n)
{
    const unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == nitems && 2 * nitems < n)
        reinterpret_cast<float2 *>(y)[n - 1] = make_float2(reinterpret_cast<const float *>(a)[n - 1], reinterpret_cast<const float *>(b)[n - 1]);
    if (i >= nitems) return;
    const float2 p = a[i], q = b[i];
    nt_store(y + i, make_float4(p.x, q.x, p.y, q.y));
}

// DelayBlock on 16-byte words: D, n and the pointers all whole multiples of 16 bytes (D4, n4 in such words)
__global__ __launch_bounds__(256) void delay_vec_kernel(const float4 *__restrict__ state_in, const float4 *__restrict__ x, float4 *__restrict__ y,
                                                        float4 *__restrict__ state_out, unsigned long n4, unsigned long D4)
{
    const unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4 + D4) return;
    const float4 v = i < D4 ? state_in[i] : x[i - D4];
    if (i < n4) nt_store(y + i, v);
    else state_out[i - n4] = v;
}

// ------------------------------------------------------------------------------------------------
// DelayBlock (radio/blocks/signal/delay.lua:43-72): y[i] = s[i] with s = [state (D samples, zero initially) | x];
// the new state is the last D samples of s.  Bit-exact copies; T = float or float2.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void delay_kernel(const T *__restrict__ state_in, const T *__restrict__ x, T *__restrict__ y,
                                                    T *__restrict__ state_out, unsigned long n, unsigned long D)
{
    unsigned long stride = (unsigned long)gridDim.x * blockDim.x;
    for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < n + D; i += stride) {
        T v = i < D ? state_in[i] : x[i - D];
        if (i < n) y[i] = v;
        if (i >= n) state_out[i - n] = v;
    }
}

// HilbertTransformBlock output assembly (radio/blocks/signal/hilberttransform.lua:111-124):
// out[i] = (s[(M-1)/2 + i], fir[i]) with s = [M-1 history | chunk]
__global__ __launch_bounds__(256) void hilbert_combine_kernel(const float *__restrict__ hist, const float *__restrict__ x,
                                                              const float *__restrict__ fir, float2 *__restrict__ y, unsigned long n, int M)
{
    unsigned long stride = (unsigned long)gridDim.x * blockDim.x;
    const long half = (M - 1) / 2;
    for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long p = half + (long)i;             // stream position of the delayed sample
        float d = p < M - 1 ? hist[p] : x[p - (M - 1)];
        y[i] = make_float2(d, fir[i]);
    }
}

// FloatToComplexBlock (radio/blocks/signal/floattocomplex.lua): y[i] = (a[i], b[i])
__global__ __launch_bounds__(256) void float_to_complex_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                               float2 *__restrict__ y, unsigned long n)
{
    unsigned long stride = (unsigned long)gridDim.x * blockDim.x;
    for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = make_float2(a[i], b[i]);
}

// ------------------------------------------------------------------------------------------------
// Welch / Bartlett averaging of GnuplotSpectrumSink (radio/blocks/sinks/gnuplotspectrum.lua:140-186): the stream
// s = [pending | chunk] is cut into frames of N samples every `hop` = N - overlap samples; every frame's (log) PSD is
// fftshifted and accumulated.  welch_gather lays the overlapping frames out contiguously for the PSD kernel,
// welch_partial / welch_final reduce the per-frame spectra in a fixed order (reproducible sums).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void welch_gather_kernel(const T *__restrict__ pending, unsigned long P, const T *__restrict__ x,
                                                           T *__restrict__ frames, unsigned long nframes, int N, int hop)
{
    unsigned long total = nframes * (unsigned long)N, stride = (unsigned long)gridDim.x * blockDim.x;
    for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        unsigned long f = i / N, j = i % N, p = f * hop + j;
        frames[i] = p < P ? pending[p] : x[p - P];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void welch_pending_kernel(const T *__restrict__ pending, unsigned long P, const T *__restrict__ x,
                                                            unsigned long start, T *__restrict__ out, unsigned long count)
{
    unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        unsigned long p = start + i;
        out[i] = p < P ? pending[p] : x[p - P];
    }
}

constexpr int WELCH_CHUNK = 64;      // frames per partial sum
__global__ __launch_bounds__(256) void welch_partial_kernel(const float *__restrict__ psd, float *__restrict__ partial, unsigned long nframes, int N)
{
    int j = blockIdx.x * 256 + threadIdx.x;
    unsigned long c = blockIdx.y, f0 = c * WELCH_CHUNK, f1 = f0 + WELCH_CHUNK < nframes ? f0 + WELCH_CHUNK : nframes;
    if (j >= N) return;
    float acc = 0.f;
    for (unsigned long f = f0; f < f1; f++) acc += psd[f * N + j];
    partial[c * N + j] = acc;
}

__global__ __launch_bounds__(256) void welch_final_kernel(const float *__restrict__ partial, unsigned long nchunks, float *__restrict__ sum, int N)
{
    int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    float acc = sum[j];
    for (unsigned long c = 0; c < nchunks; c++) acc += partial[c * N + j];
    sum[j] = acc;
}

// ------------------------------------------------------------------------------------------------
// FrequencyModulatorBlock (radio/blocks/signal/frequencymodulator.lua:77-90):
//   phase[n] = (phase[n-1] + 2*pi*k*x[n]) mod 2*pi;   y[n] = (cos phase[n], sin phase[n])
// The running phase is a prefix sum.  It is kept in TURNS as 0.64 fixed point: the per-sample increment frac(k*x[n])
// is converted once (error < 2^-64 turn), and the sum is then an unsigned 64-bit integer scan - exact, associative, the
// mod-1-turn is the integer wrap - so any tiling / chunking gives the same bits and there is no drift.
// Three small passes over tiles of 4096 samples: tile sums, exclusive scan of the tile sums (one workgroup), tile-local
// scan + phasor.  Traffic: 4 B in (read twice) + 8 B out per sample.
// ------------------------------------------------------------------------------------------------
constexpr int FMOD_LC = 16, FMOD_TILE = 256 * FMOD_LC;

__device__ __forceinline__ uint64_t fmod_increment(float x, double k)
{
    double t = k * (double)x;
    t -= floor(t);                                     // [0, 1)
    return (uint64_t)(t * 18446744073709551616.0);     // < 2^64 because t <= 1 - 2^-53
}

__device__ __forceinline__ uint64_t block_sum_u64(uint64_t v, uint64_t *sh)      // sh: 4 slots; returns the total to every thread
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void fmod_tile_sum_kernel(const float *__restrict__ x, unsigned long n, double k, uint64_t *__restrict__ tile_sum)
{
    __shared__ uint64_t sh[4];
    const unsigned long c0 = (unsigned long)blockIdx.x * FMOD_TILE + (unsigned long)threadIdx.x * FMOD_LC;
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < FMOD_LC; i++)
        if (c0 + i < n) acc += fmod_increment(x[c0 + i], k);
    uint64_t tot = block_sum_u64(acc, sh);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}

// exclusive scan of the tile sums in place, seeded with the carried phase; publishes the phase after the chunk
__global__ __launch_bounds__(256) void fmod_tile_scan_kernel(uint64_t *__restrict__ tile_sum, unsigned long ntiles, const uint64_t *__restrict__ phase_in,
                                                             uint64_t *__restrict__ phase_out)
{
    __shared__ uint64_t sh[256];
    const int tid = threadIdx.x;
    const unsigned long seg = (ntiles + 255) / 256, t0 = tid * seg, t1 = t0 + seg < ntiles ? t0 + seg : ntiles;
    uint64_t acc = 0;
    for (unsigned long t = t0; t < t1; t++) acc += tile_sum[t];
    sh[tid] = acc;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {           // Hillis-Steele inclusive scan of the segment sums
        uint64_t v = tid >= off ? sh[tid - off] : 0;
        __syncthreads();
        sh[tid] += v;
        __syncthreads();
    }
    uint64_t run = *phase_in + (tid ? sh[tid - 1] : 0);
    for (unsigned long t = t0; t < t1; t++) {
        uint64_t v = tile_sum[t];
        tile_sum[t] = run;
        run += v;
    }
    if (tid == 255) *phase_out = *phase_in + sh[255];
}

__global__ __launch_bounds__(256) void fmod_emit_kernel(const float *__restrict__ x, float2 *__restrict__ y, unsigned long n, double k,
                                                        const uint64_t *__restrict__ tile_start)
{
    __shared__ uint64_t sh[256];
    const int tid = threadIdx.x;
    const unsigned long c0 = (unsigned long)blockIdx.x * FMOD_TILE + (unsigned long)tid * FMOD_LC;
    uint64_t inc[FMOD_LC], acc = 0;
#pragma unroll
    for (int i = 0; i < FMOD_LC; i++) {
        inc[i] = c0 + i < n ? fmod_increment(x[c0 + i], k) : 0;
        acc += inc[i];
    }
    sh[tid] = acc;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        uint64_t v = tid >= off ? sh[tid - off] : 0;
        __syncthreads();
        sh[tid] += v;
        __syncthreads();
    }
    uint64_t run = tile_start[blockIdx.x] + (tid ? sh[tid - 1] : 0);
#pragma unroll
    for (int i = 0; i < FMOD_LC; i++) {
        run += inc[i];
        float c, s;
        phasor_from_turns(run, c, s);
        if (c0 + i < n) y[c0 + i] = make_float2(c, s);
    }
}

}  // namespace lrhip
