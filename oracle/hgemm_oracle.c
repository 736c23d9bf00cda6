/*
 * hgemm_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the HGEMM hot path of
 * deepreinforce-ai/CUDA-L2, used as the checker by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  Nothing in the product path (cuda-l2_amd/) may link or call it.
 *
 * What it restates
 *   - the ground truth of the reference's correctness check:
 *         truth = torch.matmul(a.cpu().float(), b.cpu().float()).half()
 *     (reference zero_one_correctness_check.py:85-90): fp16 operands widened to fp32, fp32
 *     accumulation, ONE rounding to fp16 (round-to-nearest-even) at the end;
 *   - the B^T layout the kernels consume: as_col_major (reference tools/utils.py:110-115);
 *   - the fp16-accumulate variant of the F16F16F16F16 tree (mma.sync ... f16 accumulate,
 *     reference kernels/a100_F16F16F16F16/): products exact, every partial sum rounded to fp16.
 *
 * Parity pinning: the reference ships no golden vectors.  This file is pinned
 * by tests/test_oracle.py against fixtures produced by tests/golden/make_golden.py, which runs the
 * reference's own oracle expression with the torch CPU build of this image and the reference's own
 * tools/utils.py functions.  For {0,1}-valued inputs with |truth| <= 2047 every partial sum is an
 * exact integer, so the summation order is irrelevant and the comparison is bit-exact; for
 * N(0,1) inputs fp32 summation order differs between BLAS libraries, so fixtures are compared
 * within 1 fp16 ulp of the fp32 result (the reference has no test for such inputs: "parity
 * unpinned" there, tolerance defined by BASELINE.json: 1e-3 rel fp32-acc / 1e-2 rel fp16-acc).
 *
 * Plain C99, no dependencies:  gcc -O2 -shared -fPIC hgemm_oracle.c -o libhgemm_oracle.so
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

/* ---- IEEE binary16 <-> binary32, bit-exact, no compiler half support needed ------------------- */
static float half_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1Fu;
  uint32_t man = h & 0x3FFu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal: normalise */
      int e = -1;
      do { man <<= 1; ++e; } while ((man & 0x400u) == 0);
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7F800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

static uint16_t float_to_half(float f) { /* round to nearest even, overflow -> inf */
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
  x &= 0x7FFFFFFFu;
  if (x >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (x > 0x7F800000u ? 0x200u : 0)); /* inf / nan */
  if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                                   /* rounds to inf */
  if (x < 0x33000001u) return sign;                                                           /* underflow to 0 */
  int32_t exp = (int32_t)(x >> 23) - 127;
  uint32_t man = (x & 0x7FFFFFu) | 0x800000u;
  uint32_t shift, half_man;
  if (exp < -14) { /* subnormal half */
    shift = (uint32_t)(13 + (-14 - exp));
    half_man = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_man & 1u))) ++half_man;
    return (uint16_t)(sign | half_man);
  }
  half_man = man >> 13;
  {
    const uint32_t rem = man & 0x1FFFu;
    uint32_t out = ((uint32_t)(exp + 15) << 10) + (half_man & 0x3FFu);
    if (rem > 0x1000u || (rem == 0x1000u && (out & 1u))) ++out; /* carry may bump the exponent: correct */
    return (uint16_t)(sign | out);
  }
}

/* exported for the tests of the conversion itself */
float hgemm_oracle_half_to_float(uint16_t h) { return half_to_float(h); }
uint16_t hgemm_oracle_float_to_half(float f) { return float_to_half(f); }

/* C[M,N] = fp16( sum_k fp32(A[m,k]) * fp32(B[k,n]) ), fp32 accumulation in increasing k.
 * A row-major [M][K], B row-major [K][N], C row-major [M][N]; all IEEE binary16 bit patterns.
 * Restates reference zero_one_correctness_check.py:85-90. */
void hgemm_oracle_f32acc(const uint16_t* A, const uint16_t* B, uint16_t* C, int M, int N, int K) {
  for (int m = 0; m < M; ++m) {
    for (int n = 0; n < N; ++n) {
      float acc = 0.0f;
      for (int k = 0; k < K; ++k)
        acc += half_to_float(A[(size_t)m * K + k]) * half_to_float(B[(size_t)k * N + n]);
      C[(size_t)m * N + n] = float_to_half(acc);
    }
  }
}

/* Same product computed from the transposed operand the kernels actually read:
 * Bt row-major [N][K] (= b_col_major, reference tools/utils.py:110-115). */
void hgemm_oracle_f32acc_tn(const uint16_t* A, const uint16_t* Bt, uint16_t* C, int M, int N, int K) {
  for (int m = 0; m < M; ++m) {
    for (int n = 0; n < N; ++n) {
      float acc = 0.0f;
      for (int k = 0; k < K; ++k)
        acc += half_to_float(A[(size_t)m * K + k]) * half_to_float(Bt[(size_t)n * K + k]);
      C[(size_t)m * N + n] = float_to_half(acc);
    }
  }
}

/* fp16-accumulate model of the F16F16F16F16 tree: exact fp16 x fp16 product (fits fp32), running sum
 * rounded to fp16 after every addition (reference kernels/a100_F16F16F16F16 use the
 * SM80_16x8x16_F16F16F16F16_TN mma; its internal order is unspecified, k-order is the model). */
void hgemm_oracle_f16acc(const uint16_t* A, const uint16_t* B, uint16_t* C, int M, int N, int K) {
  for (int m = 0; m < M; ++m) {
    for (int n = 0; n < N; ++n) {
      uint16_t acc = 0;
      for (int k = 0; k < K; ++k) {
        const float p = half_to_float(A[(size_t)m * K + k]) * half_to_float(B[(size_t)k * N + n]);
        acc = float_to_half(half_to_float(acc) + p);
      }
      C[(size_t)m * N + n] = acc;
    }
  }
}

/* as_col_major: [K][N] row-major -> [N][K] row-major (the storage of b_col_major). */
void hgemm_oracle_as_col_major(const uint16_t* B, uint16_t* Bt, int K, int N) {
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < N; ++n) Bt[(size_t)n * K + k] = B[(size_t)k * N + n];
}

/* max |out - truth| over entries with |truth| <= 2047 (the reference's mask rule,
 * zero_one_correctness_check.py:92,167-172); returns the max as float. */
float hgemm_oracle_masked_max_diff(const uint16_t* out, const uint16_t* truth, size_t count) {
  float worst = 0.0f;
  for (size_t i = 0; i < count; ++i) {
    const float t = half_to_float(truth[i]);
    if (t > 2047.0f || t < -2047.0f) continue;
    float d = half_to_float(out[i]) - t;
    if (d < 0) d = -d;
    if (!(d <= worst)) worst = d; /* NaN propagates as "worst" */
  }
  return worst;
}
