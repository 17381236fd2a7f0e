// fp64 elementary functions specialised for the covariance kernels (host + device).
//
// The stationary leaves of the reference are amp*exp(arg) with arg <= 0 built from
//   SE        arg = -0.5 dx^2 / l^2                      (src/GP.jl:241-245)
//   GammaExp  arg = -(|dx|/l)^gamma, 0 < gamma <= 2      (src/GP.jl:285-289)
//   Periodic  arg = (-2/l^2) sin(pi/p |dx|)^2            (src/GP.jl:331-336)
// The generic libm routines spend most of their instructions on ranges and special cases that
// cannot occur here (negative bases, huge trig arguments, NaN plumbing).  These versions keep
// double-precision accuracy on the domain that does occur:
//   exp_f   |rel err| <~ 1.5 ulp for x in [-745, 700]      (Cody-Waite + degree-13 Horner, 1 ldexp)
//   sin2_f  sin(x)^2, |rel err| <~ 3 ulp for 0 <= x < 2^20 (2-term FMA reduction by pi)
//   pow_f   u^g for u >= 0, 0 < g: exp(g*log u), rel err <~ (2 + |g log u|) ulp
// which keeps every covariance entry within ~1e-15*max(1,|arg|) relative of the reference's
// libm-based value — the reference's own result carries the same |arg|*ulp uncertainty from the
// rounding of its argument.  Validated against mpmath in tests/test_fastmath.py (CPU build of
// this header) and through the covariance parity tests on the GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define AGP_HD __host__ __device__ __forceinline__
#else
#define AGP_HD inline
#endif

namespace agp {
namespace fm {

AGP_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
AGP_HD uint64_t bits_(double x) { return __builtin_bit_cast(uint64_t, x); }
AGP_HD double dbl_(uint64_t b) { return __builtin_bit_cast(double, b); }

// exp(x).  x*log2(e) rounded to the nearest integer k, r = x - k ln2 in two FMA steps
// (ln2_hi has 32 trailing zero bits, so k*ln2_hi is exact), e^r by Horner, scale by 2^k.
AGP_HD double exp_f(double x) {
  const double L2E = 1.44269504088896338700e+00;
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  x = x < -800.0 ? -800.0 : x;               // everything below underflows to 0 anyway
  const double kf = __builtin_rint(x * L2E);
  double r = fma_(-kf, LN2_HI, x);
  r = fma_(-kf, LN2_LO, r);
  double p = 1.6059043836821613e-10;          // 1/13!
  p = fma_(p, r, 2.08767569878681e-09);       // 1/12!
  p = fma_(p, r, 2.505210838544172e-08);      // 1/11!
  p = fma_(p, r, 2.755731922398589e-07);      // 1/10!
  p = fma_(p, r, 2.7557319223985893e-06);     // 1/9!
  p = fma_(p, r, 2.48015873015873e-05);       // 1/8!
  p = fma_(p, r, 1.984126984126984e-04);      // 1/7!
  p = fma_(p, r, 1.388888888888889e-03);      // 1/6!
  p = fma_(p, r, 8.333333333333333e-03);      // 1/5!
  p = fma_(p, r, 4.1666666666666664e-02);     // 1/4!
  p = fma_(p, r, 1.6666666666666666e-01);     // 1/3!
  p = fma_(p, r, 0.5);
  p = fma_(p, r, 1.0);
  p = fma_(p, r, 1.0);
  return __builtin_ldexp(p, (int)kf);
}

// exp(x) with a 128-entry table of 2^(j/128): x = (128 e + j) ln2/128 + r, |r| <= ln2/256, e^r - 1 by a degree-5 polynomial
// (truncation 6e-19 relative), result 2^e * (T[j] + T[j] * p).  11 fp64 operations against the 21 of exp_f — and on gfx950
// vector fp64 work does not hide under fp64 MFMAs (they share the execution units: tools/gpu_dp_share.py), so every fp64
// instruction of the covariance evaluation is time taken from the factorisation.  |rel err| <~ 1.5 ulp on [-745, 700].
// `tab` = EXP_TAB (host) or a copy of it in LDS (device: a per-lane gather).
#define AGP_EXP_TAB_N 128
#define AGP_EXP_TAB_VALUES \
    1.0, 1.0054299011128027, 1.0108892860517005, 1.016378314910953, \
    1.0218971486541166, 1.0274459491187637, 1.0330248790212284, 1.0386341019613787, \
    1.0442737824274138, 1.0499440858006872, 1.0556451783605572, 1.061377227289262, \
    1.0671404006768237, 1.0729348675259756, 1.0787607977571199, 1.0846183622133092, \
    1.0905077326652577, 1.0964290818163769, 1.102382583307841, 1.1083684117236787, \
    1.1143867425958924, 1.1204377524096067, 1.1265216186082418, 1.1326385195987192, \
    1.1387886347566916, 1.1449721444318042, 1.1511892299529827, 1.1574400736337511, \
    1.1637248587775775, 1.1700437696832502, 1.1763969916502812, 1.182784710984341, \
    1.189207115002721, 1.1956643920398273, 1.202156731452703, 1.2086843236265816, \
    1.215247359980469, 1.2218460329727576, 1.22848053610687, 1.2351510639369334, \
    1.241857812073484, 1.2486009771892048, 1.255380757024691, 1.2621973503942507, \
    1.2690509571917332, 1.275941778396392, 1.2828700160787783, 1.2898358734066657, \
    1.2968395546510096, 1.3038812651919358, 1.3109612115247644, 1.318079601266064, \
    1.3252366431597413, 1.3324325470831615, 1.339667524053303, 1.3469417862329458, \
    1.3542555469368927, 1.3616090206382248, 1.3690024229745905, 1.3764359707545302, \
    1.383909881963832, 1.3914243757719262, 1.3989796725383112, 1.4065759938190154, \
    1.4142135623730951, 1.4218926021691656, 1.42961333839197, 1.4373759974489824, \
    1.4451808069770467, 1.4530279958490526, 1.460917794180647, 1.4688504333369818, \
    1.4768261459394993, 1.4848451658727524, 1.4929077282912648, 1.5010140696264256, \
    1.5091644275934228, 1.5173590411982147, 1.5255981507445384, 1.533881997840956, \
    1.5422108254079407, 1.550584877685, 1.559004400237837, 1.567469639965553, \
    1.5759808451078865, 1.5845382652524937, 1.593142151342267, 1.6017927556826934, \
    1.6104903319492543, 1.6192351351948637, 1.6280274218573478, 1.6368674497669644, \
    1.645755478153965, 1.6546917676561943, 1.6636765803267364, 1.6727101796415966, \
    1.681792830507429, 1.6909247992693053, 1.7001063537185235, 1.709337763100463, \
    1.718619298122478, 1.7279512309618377, 1.7373338352737062, 1.746767386199169, \
    1.7562521603732995, 1.7657884359332727, 1.7753764925265212, 1.785016611318935, \
    1.7947090750031072, 1.804454167806624, 1.8142521755003989, 1.8241033854070534, \
    1.8340080864093424, 1.843966568958626, 1.8539791250833855, 1.864046048397789, \
    1.8741676341103, 1.8843441790323345, 1.8945759815869656, 1.9048633418176741, \
    1.9152065613971474, 1.925605943636125, 1.9360617934922943, 1.9465744175792332, \
    1.9571441241754002, 1.9677712232331759, 1.978456026387951, 1.9891988469672663
static const double EXP_TAB[AGP_EXP_TAB_N] = {AGP_EXP_TAB_VALUES};
#if defined(__HIPCC__)
static __device__ __constant__ double c_exp_tab[AGP_EXP_TAB_N] = {AGP_EXP_TAB_VALUES};      // copied into LDS by the kernels that evaluate covariances
#endif
AGP_HD double exp_t(double x, const double* tab) {
  const double INV = 184.6649652337873;                 // 128 / ln 2
  const double LN2N_HI = 0.00541521234663378;           // ln 2 / 128, 32 significant bits (k * LN2N_HI is exact)
  const double LN2N_LO = 1.4907929134926466e-12;
  x = x < -800.0 ? -800.0 : x;
  const double kf = __builtin_rint(x * INV);
  double r = fma_(-kf, LN2N_HI, x);
  r = fma_(-kf, LN2N_LO, r);
  const int ki = (int)kf;
  const double t = tab[ki & (AGP_EXP_TAB_N - 1)];
  double q = fma_(r, 8.333333333333333e-03, 4.1666666666666664e-02);      // 1/120, 1/24
  q = fma_(q, r, 1.6666666666666666e-01);
  q = fma_(q, r, 0.5);
  q = fma_(q, r, 1.0);
  const double p = q * r;
  return __builtin_ldexp(fma_(t, p, t), ki >> 7);
}

// sin(x)^2 for x >= 0: period pi, so reduce to r in [-pi/2, pi/2] and square an odd polynomial.
AGP_HD double sin2_f(double x) {
  const double INV_PI = 3.18309886183790691216e-01;
  const double PI_HI = 3.14159265358979311600e+00;
  const double PI_LO = 1.22464679914735320717e-16;
  const double kf = __builtin_rint(x * INV_PI);
  double r = fma_(-kf, PI_HI, x);
  r = fma_(-kf, PI_LO, r);
  const double z = r * r;
  double p = 3.868170170630684e-23;           //  1/23!
  p = fma_(p, z, -1.9572941063391263e-20);    // -1/21!
  p = fma_(p, z, 8.22063524662433e-18);       //  1/19!
  p = fma_(p, z, -2.8114572543455206e-15);    // -1/17!
  p = fma_(p, z, 7.647163731819816e-13);      //  1/15!
  p = fma_(p, z, -1.6059043836821613e-10);    // -1/13!
  p = fma_(p, z, 2.505210838544172e-08);      //  1/11!
  p = fma_(p, z, -2.7557319223985893e-06);    // -1/9!
  p = fma_(p, z, 1.984126984126984e-04);      //  1/7!
  p = fma_(p, z, -8.333333333333333e-03);     // -1/5!
  p = fma_(p, z, 1.6666666666666666e-01);     //  1/3!  (applied with a minus below)
  // sin r = r - r z (1/6 - z(...))  -> s = r * (1 - z*p) with p built with alternating signs above
  const double s = fma_(-(r * z), p, r);
  return s * s;
}

// (sin r, cos r) with r = x - k pi in [-pi/2, pi/2], x >= 0: sin(x)^2 = s^2 and sin(x) cos(x) = s c (both have
// period pi, so the sign lost in the reduction cancels) -- what the Periodic kernel's derivatives need.
// Absolute error <~ 2e-16 each.
AGP_HD void sincos_pi_f(double x, double* sn, double* cs) {
  const double INV_PI = 3.18309886183790691216e-01;
  const double PI_HI = 3.14159265358979311600e+00;
  const double PI_LO = 1.22464679914735320717e-16;
  const double kf = __builtin_rint(x * INV_PI);
  double r = fma_(-kf, PI_HI, x);
  r = fma_(-kf, PI_LO, r);
  const double z = r * r;
  double p = 3.868170170630684e-23;           //  1/23!
  p = fma_(p, z, -1.9572941063391263e-20);
  p = fma_(p, z, 8.22063524662433e-18);
  p = fma_(p, z, -2.8114572543455206e-15);
  p = fma_(p, z, 7.647163731819816e-13);
  p = fma_(p, z, -1.6059043836821613e-10);
  p = fma_(p, z, 2.505210838544172e-08);
  p = fma_(p, z, -2.7557319223985893e-06);
  p = fma_(p, z, 1.984126984126984e-04);
  p = fma_(p, z, -8.333333333333333e-03);
  p = fma_(p, z, 1.6666666666666666e-01);
  *sn = fma_(-(r * z), p, r);
  double c = 1.6117375710961184e-24;           //  1/24!
  c = fma_(c, z, -8.896791392450574e-22);      // -1/22!
  c = fma_(c, z, 4.110317623312165e-19);       //  1/20!
  c = fma_(c, z, -1.5619206968586225e-16);     // -1/18!
  c = fma_(c, z, 4.779477332387385e-14);       //  1/16!
  c = fma_(c, z, -1.1470745597729725e-11);     // -1/14!
  c = fma_(c, z, 2.08767569878681e-09);        //  1/12!
  c = fma_(c, z, -2.755731922398589e-07);      // -1/10!
  c = fma_(c, z, 2.48015873015873e-05);        //  1/8!
  c = fma_(c, z, -1.388888888888889e-03);      // -1/6!
  c = fma_(c, z, 4.1666666666666664e-02);      //  1/4!
  c = fma_(c, z, -0.5);
  *cs = fma_(c, z, 1.0);
}

// log(u) for finite u > 0 (fdlibm __ieee754_log kernel, ~1 ulp).
AGP_HD double log_f(double u) {
  const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  int e = 0;
  if (u < 2.2250738585072014e-308) { u *= 18014398509481984.0; e = -54; }   // subnormal: scale by 2^54
  const uint64_t b = bits_(u);
  e += (int)(b >> 52) - 1023;
  double m = dbl_((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);       // [1, 2)
  if (m > 1.41421356237309514547) { m *= 0.5; e += 1; }                       // [sqrt(1/2), sqrt(2))
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * fma_(w, fma_(w, Lg6, Lg4), Lg2);
  const double t2 = z * fma_(w, fma_(w, fma_(w, Lg7, Lg5), Lg3), Lg1);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)e;
  return dk * LN2_HI - ((hfsq - (s * (hfsq + R) + dk * LN2_LO)) - f);
}

// u^g for u >= 0 (0^g = 0 for g > 0, the only zero case the kernels produce: dt = 0).
AGP_HD double pow_f(double u, double g) {
  const double y = g * log_f(u > 0.0 ? u : 1.0);
  const double t = exp_f(y);
  return u > 0.0 ? t : 0.0;
}

}  // namespace fm
}  // namespace agp
