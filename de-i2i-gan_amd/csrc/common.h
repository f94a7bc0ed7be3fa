// Shared device/host helpers for the dei2i HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DEI2I_HD __host__ __device__ __forceinline__
#define DEI2I_D __device__ __forceinline__
#else
#define DEI2I_HD inline
#endif

namespace dei2i {

enum DType : int { DT_BF16 = 0, DT_F32 = 1 };
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2 };
enum PadMode : int { PAD_ZERO = 0, PAD_REFLECT = 1 };

// Division by a launch-invariant 32-bit divisor (numerator < 2^31): q = (mulhi(n, m) + n) >> l.
struct FastDiv {
  uint32_t m;
  uint32_t l;
  uint32_t d;
};

inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  f.l = l;
  f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  return f;
}

DEI2I_HD uint32_t fd_div(uint32_t n, const FastDiv& f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (__umulhi(n, f.m) + n) >> f.l;
#else
  return (uint32_t)((((uint64_t)n * f.m) >> 32) + n) >> f.l;
#endif
}

#if defined(__HIPCC__)
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

typedef uint16_t bf16_t;  // storage type

DEI2I_D float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
DEI2I_D bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int VEC = 4;  // elements per 16-byte vector
  static DEI2I_D float load(const float* p) { return *p; }
  static DEI2I_D void store(float* p, float v) { *p = v; }
  static DEI2I_D void unpack(const u32x4& v, float* f) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
    f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
  }
  static DEI2I_D u32x4 pack(const float* f) {
    u32x4 v;
    v.x = __float_as_uint(f[0]); v.y = __float_as_uint(f[1]);
    v.z = __float_as_uint(f[2]); v.w = __float_as_uint(f[3]);
    return v;
  }
};
template <> struct Elem<bf16_t> {
  static constexpr int VEC = 8;
  static DEI2I_D float load(const bf16_t* p) { return bf16_to_f32(*p); }
  static DEI2I_D void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
  static DEI2I_D void unpack(const u32x4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
  }
  static DEI2I_D u32x4 pack(const float* f) {
    u32x4 v;
    v.x = (uint32_t)f32_to_bf16(f[0]) | ((uint32_t)f32_to_bf16(f[1]) << 16);
    v.y = (uint32_t)f32_to_bf16(f[2]) | ((uint32_t)f32_to_bf16(f[3]) << 16);
    v.z = (uint32_t)f32_to_bf16(f[4]) | ((uint32_t)f32_to_bf16(f[5]) << 16);
    v.w = (uint32_t)f32_to_bf16(f[6]) | ((uint32_t)f32_to_bf16(f[7]) << 16);
    return v;
  }
};

DEI2I_D float apply_act(float v, int act) {
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == ACT_LRELU) return v >= 0.f ? v : 0.2f * v;
  return v;
}
// ---- fp8 (OCP e4m3fn) quantisation of 8 values: clamp to the finite range (+-448), round to nearest even ----
DEI2I_D float clamp_e4m3(float v) { return fminf(fmaxf(v, -448.f), 448.f); }
DEI2I_D u32x2 pack_e4m3_8(const float (&f)[8], float scale) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(f[0] * scale), clamp_e4m3(f[1] * scale), lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(f[2] * scale), clamp_e4m3(f[3] * scale), lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(f[4] * scale), clamp_e4m3(f[5] * scale), hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(f[6] * scale), clamp_e4m3(f[7] * scale), hi, true);
  u32x2 o;
  o.x = (uint32_t)lo; o.y = (uint32_t)hi;
  return o;
}
// the e4m3 copy of a just-packed bf16 vector (quantises the ROUNDED bf16 values, so a fused producer and the standalone
// quantiser give the same bytes)
DEI2I_D void store_e4m3_of_bf16x8(unsigned char* __restrict__ dst, const u32x4& packed_bf16, float scale) {
  float r[8];
  r[0] = __uint_as_float(packed_bf16.x << 16); r[1] = __uint_as_float(packed_bf16.x & 0xffff0000u);
  r[2] = __uint_as_float(packed_bf16.y << 16); r[3] = __uint_as_float(packed_bf16.y & 0xffff0000u);
  r[4] = __uint_as_float(packed_bf16.z << 16); r[5] = __uint_as_float(packed_bf16.z & 0xffff0000u);
  r[6] = __uint_as_float(packed_bf16.w << 16); r[7] = __uint_as_float(packed_bf16.w & 0xffff0000u);
  *reinterpret_cast<u32x2*>(dst) = pack_e4m3_8(r, scale);
}

// Branch-free epilogue helpers of the MFMA kernels (the unrolled per-element `act` switch was most of their code, and
// a dispatch walks its code cold).  act(v) = max(v,0) + slope*min(v,0) with slope 0 / 0.2 / 1 for ReLU / LeakyReLU /
// none -- exact for all three.
DEI2I_D float act_slope(int act) { return act == ACT_RELU ? 0.f : (act == ACT_LRELU ? 0.2f : 1.f); }
// bias of four consecutive output channels n .. n+3 (0 beyond wrows) and the bit masks that clear the packed bf16
// pairs of channels beyond wrows
DEI2I_D void epi_col_consts(const float* __restrict__ bias, int n, int wrows, float (&bq)[4], uint32_t& m01, uint32_t& m23) {
#pragma unroll
  for (int k = 0; k < 4; ++k) bq[k] = (bias != nullptr && n + k < wrows) ? bias[n + k] : 0.f;
  const int live = wrows - n;
  m01 = live >= 2 ? 0xffffffffu : (live == 1 ? 0x0000ffffu : 0u);
  m23 = live >= 4 ? 0xffffffffu : (live == 3 ? 0x0000ffffu : 0u);
}
DEI2I_D u32x2 epi_finish4(float a0, float a1, float a2, float a3, const float (&bq)[4], float slope, uint32_t m01, uint32_t m23) {
  const float a4[4] = {a0, a1, a2, a3};
  float v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float t = a4[k] + bq[k];
    v[k] = fmaf(slope, fminf(t, 0.f), fmaxf(t, 0.f));
  }
  u32x2 pk;
  pk.x = ((uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16)) & m01;
  pk.y = ((uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16)) & m23;
  return pk;
}

// derivative of the activation expressed through its OUTPUT z (valid for relu / lrelu(0.2): sign(z) == sign(pre))
DEI2I_D float act_grad_from_out(float z, int act) {
  if (act == ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == ACT_LRELU) return z >= 0.f ? 1.f : 0.2f;
  return 1.f;
}

// VEC consecutive fp32 coefficients (16-byte aligned) -> registers through 16-byte loads
template <int VEC> DEI2I_D void ldcoef(const float* __restrict__ p, float (&v)[VEC]) {
#pragma unroll
  for (int q = 0; q < VEC / 4; ++q) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
    v[4 * q + 0] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
  }
}

// class of a pixel coordinate in the 5 x 5 gamma / beta table of a SPADE block whose label map is constant per image: the two
// border rows / columns on each side have their own values (zero-padded 3x3 convs over a constant map), the interior is class 2
DEI2I_D int border_class(int i, int extent) { return i < 2 ? i : (i >= extent - 2 ? 4 - (extent - 1 - i) : 2); }

DEI2I_D float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
#endif  // __HIPCC__

#if defined(__HIPCC__)
// LDS-DMA (global -> LDS, 16 bytes per lane, 1 KB per wave instruction) as inline asm: M0 is written in the same statement and
// restored after it (hipcc reserves M0).  Issued through __builtin_amdgcn_global_load_lds, hipcc knows that LDS is being written
// and -- unable to tell that a prefetch into the NEXT buffer does not alias the buffer being read -- puts s_waitcnt vmcnt(0) in
// front of the next LDS read (and into every __syncthreads()): the multi-stage rings of wgrad_halo / wgrad_v2 / the thin convs
// then drain their prefetch before computing.  With the asm form the kernels' own counted vmcnt waits and barriers are what
// order the DMA against the reads -- each of them has to be complete: hipcc adds nothing.
DEI2I_D void glds16_asm(const void* gptr, unsigned char* lds_wave_base) {
  typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_byte_t*)lds_wave_base);   // wave-uniform by contract
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gptr), "s"(dst) : "memory");
}
// the same with a wave-uniform 64-bit base (SGPR pair) + a per-lane unsigned 32-bit BYTE offset: no 64-bit vector add per instruction
// (lds_dst: the LDS byte address of the wave's 1 KB piece, wave-uniform -- lds_addr_of(smem) + offset)
DEI2I_D unsigned lds_addr_of(const void* lds_ptr) {
  typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
  return (unsigned)(uintptr_t)(lds_byte_t*)lds_ptr;
}
DEI2I_D void glds16_asm_s(const void* sbase, unsigned voff_bytes, unsigned lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)sbase);          // (the builtin returns int:
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uintptr_t)sbase >> 32));  //  no sign extension into the high half)
  const unsigned long long sb = ((unsigned long long)hi << 32) | (unsigned long long)lo;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff_bytes), "s"(sb), "s"(dst) : "memory");
}
#endif

}  // namespace dei2i
