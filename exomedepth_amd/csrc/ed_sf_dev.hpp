// ed_sf_dev.hpp -- device-side special functions for the beta-binomial emission kernel (gfx950).
//
// What is computed: log B(x,y) exactly as the reference computes it -- the same sequence of binary64
// operations as gsl_sf_lnbeta (reference src/beta.c:49-114) and the functions below it
// (src/VP_gamma.c:735-756 Lanczos, :928-980 Pade windows, :761-787 small-x series, :986-1007 and
// :1332-1379 Gamma*, src/VP_log.c:196-232 log(1+x), Clenshaw recurrence src/VP_gamma.c:36-67) -- with
// log/exp/sin replaced by the portable definitions of ed_pmath.h.  No a*b+c is contracted (the file
// is built with -ffp-contract=off), so every lane reproduces the CPU checker bit for bit.
//
// How it is written: this is not the reference's control flow.  The result/err structs, status
// plumbing and error printing are gone; tables live in __constant__ memory and are indexed
// wave-uniformly (scalar loads); the positive domain (all the path can reach with 0 < phi < 1,
// 0 < expected < 1) is the fast path and everything else funnels into one cold function.
#pragma once
#include <hip/hip_runtime.h>
#include "ed_pmath.h"
#include "ed_sing_tables.h"

// Cold functions (ranges the path rarely or never reaches) are kept out of line so that the hot routes stay small.
// -DED_COLD_INLINE builds the diagnostic variant of the library in which they are inlined instead
// (tools/soak_emission.py --variant coldinline: same bits expected, different code generation).
#if defined(ED_COLD_INLINE)
#define EDSF_COLD __forceinline__
#else
#define EDSF_COLD __noinline__
#endif

namespace edsf {

// ---- constants, spelled as the reference spells them (src/gsl_math.h, src/gsl_machine.h) ----
#define EDSF_M_E 2.71828182845904523536028747135
#define EDSF_M_PI 3.14159265358979323846264338328
#define EDSF_M_SQRT2 1.41421356237309504880168872421
#define EDSF_M_SQRTPI 1.77245385090551602729816748334
#define EDSF_M_LN2 0.69314718055994530941723212146
#define EDSF_M_LNPI 1.14472988584940017414342735135
#define EDSF_LOGROOT2PI 0.9189385332046727418
#define EDSF_DBL_EPS 2.2204460492503131e-16
#define EDSF_ROOT4_EPS 1.2207031250000000e-04
#define EDSF_ROOT6_EPS 2.4607833005759251e-03

// Chebyshev / Lanczos coefficients of the GSL special-function library (mathematical constants;
// values as at reference src/VP_gamma.c:594-688, src/VP_log.c:73-95).
__constant__ const double k_gstar_a[30] = {
    2.16786447866463034423060819465,     -0.05533249018745584258035832802,    0.01800392431460719960888319748,
    -0.00580919269468937714480019814,    0.00186523689488400339978881560,     -0.00059746524113955531852595159,
    0.00019125169907783353925426722,     -0.00006124996546944685735909697,    0.00001963889633130842586440945,
    -6.3067741254637180272515795142e-06, 2.0288698405861392526872789863e-06,  -6.5384896660838465981983750582e-07,
    2.1108698058908865476480734911e-07,  -6.8260714912274941677892994580e-08, 2.2108560875880560555583978510e-08,
    -7.1710331930255456643627187187e-09, 2.3290892983985406754602564745e-09,  -7.5740371598505586754890405359e-10,
    2.4658267222594334398525312084e-10,  -8.0362243171659883803428749516e-11, 2.6215616826341594653521346229e-11,
    -8.5596155025948750540420068109e-12, 2.7970831499487963614315315444e-12,  -9.1471771211886202805502562414e-13,
    2.9934720198063397094916415927e-13,  -9.8026575909753445931073620469e-14, 3.2116773667767153777571410671e-14,
    -1.0518035333878147029650507254e-14, 3.4144405720185253938994854173e-15,  -1.0115153943081187052322643819e-15};
__constant__ const double k_gstar_b[30] = {
    0.0057502277273114339831606096782,   0.0004496689534965685038254147807,   -0.0001672763153188717308905047405,
    0.0000615137014913154794776670946,   -0.0000223726551711525016380862195,  8.0507405356647954540694800545e-06,
    -2.8671077107583395569766746448e-06, 1.0106727053742747568362254106e-06,  -3.5265558477595061262310873482e-07,
    1.2179216046419401193247254591e-07,  -4.1619640180795366971160162267e-08, 1.4066283500795206892487241294e-08,
    -4.6982570380537099016106141654e-09, 1.5491248664620612686423108936e-09,  -5.0340936319394885789686867772e-10,
    1.6084448673736032249959475006e-10,  -5.0349733196835456497619787559e-11, 1.5357154939762136997591808461e-11,
    -4.5233809655775649997667176224e-12, 1.2664429179254447281068538964e-12,  -3.2648287937449326771785041692e-13,
    7.1528272726086133795579071407e-14,  -9.4831735252566034505739531258e-15, -2.3124001991413207293120906691e-15,
    2.8406613277170391482590129474e-15,  -1.7245370321618816421281770927e-15, 8.6507923128671112154695006592e-16,
    -3.9506563665427555895391869919e-16, 1.6779342132074761078792361165e-16,  -6.0483153034414765129837716260e-17};
__constant__ const double k_lopx[21] = {
    2.16647910664395270521272590407,     -0.28565398551049742084877469679,    0.01517767255690553732382488171,
    -0.00200215904941415466274422081,    0.00019211375164056698287947962,     -0.00002553258886105542567601400,
    2.9004512660400621301999384544e-06,  -3.8873813517057343800270917900e-07, 4.7743678729400456026672697926e-08,
    -6.4501969776090319441714445454e-09, 8.2751976628812389601561347296e-10,  -1.1260499376492049411710290413e-10,
    1.4844576692270934446023686322e-11,  -2.0328515972462118942821556033e-12, 2.7291231220549214896095654769e-13,
    -3.7581977830387938294437434651e-14, 5.1107345870861673561462339876e-15,  -7.0722150011433276578323272272e-16,
    9.7089758328248469219003866867e-17,  -1.3492637457521938883731579510e-17, 1.8657327910677296608121390705e-18};

// n!, psi(n), psi'(n) (tools/gen_sing_tables.py) and B_2j/(2j)! (src/VP_zeta.c:563-579): only log|Gamma| next to a negative
// integer reads them (lngamma_sgn_sing below)
__constant__ const double k_fact[ED_FACT_TABLE_N] = ED_FACT_TABLE;
__constant__ const double k_psi_int[ED_PSI_TABLE_N] = ED_PSI_TABLE;
__constant__ const double k_psi1_int[ED_PSI1_TABLE_N] = ED_PSI1_TABLE;
__constant__ const double k_hzeta_c[15] = ED_HZETA_C;

// ---- correctly rounded division without the scaling / fix-up stages ----------------------------
// a/b as hipcc lowers it is  v_div_scale x2, v_rcp, 2 Newton steps, multiply, residual, v_div_fmas,
// v_div_fixup (11 instructions).  The scale and fix-up stages only matter when an operand or the
// quotient leaves the normal range (or is 0/inf/NaN in the divisor).  Where the caller guarantees
//     b finite, normal, 2^-900 < |b| < 2^900;  a finite (zero allowed), |a/b| normal or zero
// the remaining 8 instructions produce the same correctly rounded quotient (checked bit for bit against
// '/' on the device and against the host in tests/test_gpu_parity.py::test_fast_division_is_exact).
__device__ __forceinline__ double fdiv(double a, double b)
{
  double r = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  const double q = a * r;
  const double rem = __builtin_fma(-b, q, a);
  return __builtin_fma(rem, r, q);
}

// ed_pexp(x) for |x| < 2^-5 only: the short-series branch of ed_pexp, without its range tests.
__device__ __forceinline__ double pexp_small(double x)
{
  double q = 1.0 / 5040.0;
  q = ed_pm_fma_k(q, x, 1.0 / 720.0);
  q = ed_pm_fma_k(q, x, 1.0 / 120.0);
  q = ed_pm_fma_k(q, x, 1.0 / 24.0);
  q = ed_pm_fma_k(q, x, 1.0 / 6.0);
  q = ed_pm_fma_k(q, x, 0.5);
  return 1.0 + ed_pm_fma(x * x, q, x);
}

// ed_plog(x) for normal positive finite x: the core of the portable definition without its special cases.
// LT: the log table staged in LDS by the calling kernel (NULL: the table in constant memory -- 64 lanes gathering
// rows from it keep the texture addresser busy ~48 cycles per load, which cancelled the instruction savings of
// the table-driven log in k_emit_batch until the table moved to LDS).
__device__ __forceinline__ double plog_pos(double x, const double* LT = nullptr)
{
  return LT ? ed_plog_core_t(x, 0, LT) : ed_plog_core(x, 0);
}

// ed_plog with the fast path taken whenever x is a normal positive finite number (the cold generic
// definition handles zero, subnormals, inf, NaN); same bits either way.
__device__ __forceinline__ double plog_fast(double x, const double* LT = nullptr)
{
  if (x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308) return plog_pos(x, LT);
  return ed_plog(x);
}

// Clenshaw recurrence on [-1,1]; the argument map ((2x+1)-1)/2 is kept because it is not an identity
// in binary64 (reference src/VP_gamma.c:45-46).
template <int ORDER>
__device__ __forceinline__ double clenshaw(const double* __restrict__ c, double x)
{
  const double y = ((2.0 * x + 1.0) - 1.0) / 2.0;
  const double y2 = 2.0 * y;
  double d = 0.0, dd = 0.0;
#pragma unroll
  for (int j = ORDER; j >= 1; --j) {
    const double t = d;
    d = (y2 * d - dd) + c[j];
    dd = t;
  }
  return (y * d - dd) + 0.5 * c[0];
}

// log Gamma(x), x >= 0.5 and outside the Pade windows: Lanczos gamma=7 (src/VP_gamma.c:735-756)
__device__ __forceinline__ double lngamma_lanczos(double x, const double* LT = nullptr)
{
  x -= 1.0;
  if (x < 0x1p900) {   // every quotient below is in fdiv's range (x + k >= 0.5)
    double Ag = 0.99999999999980993227684700473478;
    Ag += fdiv(676.520368121885098567009190444019, x + 1.0);
    Ag += fdiv(-1259.13921672240287047156078755283, x + 2.0);
    Ag += fdiv(771.3234287776530788486528258894, x + 3.0);
    Ag += fdiv(-176.61502916214059906584551354, x + 4.0);
    Ag += fdiv(12.507343278686904814458936853, x + 5.0);
    Ag += fdiv(-0.13857109526572011689554707, x + 6.0);
    Ag += fdiv(9.984369578019570859563e-6, x + 7.0);
    Ag += fdiv(1.50563273514931155834e-7, x + 8.0);
    const double term1 = (x + 0.5) * plog_pos(fdiv(x + 7.5, EDSF_M_E), LT);
    const double term2 = EDSF_LOGROOT2PI + plog_pos(Ag, LT);   // 1 < Ag < 1400 for every x >= -0.5 here: a normal number
    return term1 + (term2 - 7.0);
  }
  double Ag = 0.99999999999980993227684700473478;
  Ag += 676.520368121885098567009190444019 / (x + 1.0);
  Ag += -1259.13921672240287047156078755283 / (x + 2.0);
  Ag += 771.3234287776530788486528258894 / (x + 3.0);
  Ag += -176.61502916214059906584551354 / (x + 4.0);
  Ag += 12.507343278686904814458936853 / (x + 5.0);
  Ag += -0.13857109526572011689554707 / (x + 6.0);
  Ag += 9.984369578019570859563e-6 / (x + 7.0);
  Ag += 1.50563273514931155834e-7 / (x + 8.0);
  const double term1 = (x + 0.5) * ed_plog((x + 7.5) / EDSF_M_E);
  const double term2 = EDSF_LOGROOT2PI + ed_plog(Ag);
  return term1 + (term2 - 7.0);
}

// (2,2) Pade + correction for log Gamma(1+eps), log Gamma(2+eps), |eps| < 0.01 (src/VP_gamma.c:928-980)
__device__ EDSF_COLD double lngamma_pade(double eps, int two)
{
  const double n1 = two ? 1.000895834786669227164446568 : -1.0017419282349508699871138440;
  const double n2 = two ? 4.209376735287755081642901277 : 1.7364839209922879823280541733;
  const double d1 = two ? 2.618851904903217274682578255 : 1.2433006018858751556055436011;
  const double d2 = two ? 10.85766559900983515322922936 : 5.0456274100274010152489597514;
  const double pf = two ? 2.85337998765781918463568869 : 2.0816265188662692474880210318;
  const double c0 = two ? 0.0001139406357036744 : 0.004785324257581753;
  const double c1 = two ? -0.0001365435269792533 : -0.01192457083645441;
  const double c2 = two ? 0.0001067287169183665 : 0.01931961413960498;
  const double c3 = two ? -0.0000693271800931282 : -0.02594027398725020;
  const double c4 = two ? 0.0000407220927867950 : 0.03141928755021455;
  const double num = (eps + n1) * (eps + n2);
  const double den = (eps + d1) * (eps + d2);
  const double pade = pf * num / den;
  const double eps5 = eps * eps * eps * eps * eps;
  const double corr = eps5 * (c0 + eps * (c1 + eps * (c2 + eps * (c3 + c4 * eps))));
  return eps * (pade + corr);
}

// log Gamma for 0 < x < 0.5 (cold): small-x series (:761-787) or reflection (:1180-1199 / :1244-1276).
// zform selects sin(pi*(1-x)) (gsl_sf_lngamma_e, used by Gamma*) or sin(pi*x) (gsl_sf_lngamma_sgn_e).
__device__ EDSF_COLD double lngamma_below_half(double x, bool zform)
{
  if (x < 0.02) {
    const double c1 = -0.07721566490153286061, c2 = -0.01094400467202744461, c3 = 0.09252092391911371098,
                 c4 = -0.01827191316559981266, c5 = 0.01800493109685479790, c6 = -0.00685088537872380685,
                 c7 = 0.00399823955756846603, c8 = -0.00189430621687107802, c9 = 0.00097473237804513221,
                 c10 = -0.00048434392722255893;
    const double g6 = c6 + x * (c7 + x * (c8 + x * (c9 + x * c10)));
    const double g = x * (c1 + x * (c2 + x * (c3 + x * (c4 + x * (c5 + x * g6)))));
    const double gee = (g + 1.0 / (1.0 + x)) + 0.5 * x;
    return ed_plog(gee / fabs(x));
  }
  const double z = 1.0 - x;
  const double s = ed_psin_0pi(EDSF_M_PI * (zform ? z : x));
  // for 0.02 <= x < 0.5 : |s| >= sin(0.02 pi) > 0.015 pi, so neither the s==0 nor the near-pole branch applies
  return EDSF_M_LNPI - (ed_plog(fabs(s)) + lngamma_lanczos(z));
}

// log Gamma(x) for x > 0 with the reference's window selection (src/VP_gamma.c:1219-1242)
__device__ __forceinline__ double lngamma_pos(double x, bool zform, const double* LT = nullptr)
{
  if (x > 2.01) return lngamma_lanczos(x, LT);   // beyond both Pade windows and 0.5: the usual case, one comparison
  if (fabs(x - 1.0) < 0.01) return lngamma_pade(x - 1.0, 0);
  if (fabs(x - 2.0) < 0.01) return lngamma_pade(x - 2.0, 1);
  if (x >= 0.5) return lngamma_lanczos(x, LT);
  return lngamma_below_half(x, zform);
}

// Gamma*(x) for x >= 10: exp(Stirling series / x) (src/VP_gamma.c:986-1007), and the 4-term form
// above 1/eps^(1/4) (:1366-1373)
__device__ __forceinline__ double gammastar_large(double x)
{
  if (x < 1.0 / EDSF_ROOT4_EPS) {
    const double y = fdiv(1.0, x * x);   // 10 <= x < 8192
    const double c0 = 1.0 / 12.0, c1 = -1.0 / 360.0, c2 = 1.0 / 1260.0, c3 = -1.0 / 1680.0, c4 = 1.0 / 1188.0,
                 c5 = -691.0 / 360360.0, c6 = 1.0 / 156.0, c7 = -3617.0 / 122400.0;
    const double ser = c0 + y * (c1 + y * (c2 + y * (c3 + y * (c4 + y * (c5 + y * (c6 + y * c7))))));
    return pexp_small(fdiv(ser, x));   // 0 < ser/x < 0.0084
  }
  if (x < 1.0 / EDSF_DBL_EPS) {
    const double xi = fdiv(1.0, x);   // 8192 <= x < 4.6e15
    return 1.0 + fdiv(xi, 12.0) * (1.0 + fdiv(xi, 24.0) * (1.0 - xi * (139.0 / 180.0 + 571.0 / 8640.0 * xi)));
  }
  return 1.0;
}

// Gamma*(x) for 0 < x < 10 (src/VP_gamma.c:1340-1362): less common on the path, kept out of line
__device__ EDSF_COLD double gammastar_small(double x)
{
  if (x < 0.5) {
    const double lg = lngamma_pos(x, true);
    const double lx = ed_plog(x);
    const double c = 0.5 * (EDSF_M_LN2 + EDSF_M_LNPI);
    const double lnr = ((lg - (x - 0.5) * lx) + x) - c;
    return ed_pexp(lnr);
  }
  if (x < 2.0) {
    const double t = 4.0 / 3.0 * (x - 0.5) - 1.0;
    return clenshaw<29>(k_gstar_a, t);
  }
  const double t = 0.25 * (x - 2.0) - 1.0;
  const double c = clenshaw<29>(k_gstar_b, t);
  return (c / (x * x) + 1.0) + 1.0 / (12.0 * x);
}

__device__ __forceinline__ double gammastar_pos(double x) { return x >= 10.0 ? gammastar_large(x) : gammastar_small(x); }

// The same recurrence for a FINITE argument, without the operations whose result is known: with d = dd = 0 the first step
// (y2 * 0 - 0) + c[ORDER] is c[ORDER] itself (y2 * 0 is +-0 for finite y2, +-0 - 0 is +-0, and c[ORDER] != 0), and in the second
// step "- dd" subtracts +0, an identity.  Four operations less, every remaining one unchanged: same bits.
template <int ORDER>
__device__ __forceinline__ double clenshaw_finite(const double* __restrict__ c, double x)
{
  static_assert(ORDER >= 2, "two steps are peeled");
  const double y = ((2.0 * x + 1.0) - 1.0) / 2.0;
  const double y2 = 2.0 * y;
  double dd = c[ORDER];
  double d = y2 * dd + c[ORDER - 1];
#pragma unroll
  for (int j = ORDER - 2; j >= 1; --j) {
    const double t = d;
    d = (y2 * d - dd) + c[j];
    dd = t;
  }
  return (y * d - dd) + 0.5 * c[0];
}

// log(1+x) for 0 < x < 0.2 -- the only range lnbeta's ratio branch produces (src/VP_log.c:204-226)
__device__ __forceinline__ double log1plusx_ratio(double x)
{
  if (x < EDSF_ROOT6_EPS) {
    const double c1 = -0.5, c2 = 1.0 / 3.0, c3 = -1.0 / 4.0, c4 = 1.0 / 5.0, c5 = -1.0 / 6.0, c6 = 1.0 / 7.0,
                 c7 = -1.0 / 8.0, c8 = 1.0 / 9.0, c9 = -1.0 / 10.0;
    const double t = c5 + x * (c6 + x * (c7 + x * (c8 + x * c9)));
    return x * (1.0 + x * (c1 + x * (c2 + x * (c3 + x * (c4 + x * t)))));
  }
  const double t = fdiv(0.5 * (8.0 * x + 1.0), x + 2.0);
  return x * clenshaw_finite<20>(k_lopx, t);   // t is a quotient of finite normal numbers
}

// ---- log|Gamma(x)| with sign for ANY x, as gsl_sf_lngamma_sgn_e computes it (src/VP_gamma.c:1219-1285) -- the cold
// side of the path: a non-positive or NaN shape parameter needs phi >= 1 or expected outside (0, 1).  Operation for
// operation the portable flavour of the checker (oracle/edo_gsl.inc), whose libm flavour is bit-identical to the
// reference build on these arguments (tests/test_oracle_ref.py).

// gsl_sf_lnfact_e (src/VP_gamma.c:1548-1561)
__device__ __forceinline__ double lnfact_u(unsigned n)
{
  if (n <= 170u) return ed_plog(k_fact[n]);
  double x = (double)n + 1.0;          // gsl_sf_lngamma_e(n + 1.0) -> Lanczos; plain divisions as in the checker
  x -= 1.0;
  double Ag = 0.99999999999980993227684700473478;
  Ag += 676.520368121885098567009190444019 / (x + 1.0);
  Ag += -1259.13921672240287047156078755283 / (x + 2.0);
  Ag += 771.3234287776530788486528258894 / (x + 3.0);
  Ag += -176.61502916214059906584551354 / (x + 4.0);
  Ag += 12.507343278686904814458936853 / (x + 5.0);
  Ag += -0.13857109526572011689554707 / (x + 6.0);
  Ag += 9.984369578019570859563e-6 / (x + 7.0);
  Ag += 1.50563273514931155834e-7 / (x + 8.0);
  const double term1 = (x + 0.5) * ed_plog((x + 7.5) / EDSF_M_E);
  const double term2 = EDSF_LOGROOT2PI + ed_plog(Ag);
  return term1 + (term2 - 7.0);
}

// gsl_sf_psi_int_e / gsl_sf_psi_1_int_e (src/VP_psi.c:604-630, :717-741), n >= 1
__device__ __forceinline__ double psi_int(int n)
{
  if (n <= 100) return k_psi_int[n];
  const double c2 = -1.0 / 12.0, c3 = 1.0 / 120.0, c4 = -1.0 / 252.0, c5 = 1.0 / 240.0;
  const double ni2 = (1.0 / n) * (1.0 / n);
  const double ser = ni2 * (c2 + ni2 * (c3 + ni2 * (c4 + ni2 * c5)));
  return (ed_plog((double)n) - 0.5 / n) + ser;
}
__device__ __forceinline__ double psi_1_int(int n)
{
  if (n <= 100) return k_psi1_int[n];
  const double c0 = -1.0 / 30.0, c1 = 1.0 / 42.0, c2 = -1.0 / 30.0;
  const double ni2 = (1.0 / n) * (1.0 / n);
  const double ser = (ni2 * ni2) * (c0 + ni2 * (c1 + c2 * ni2));
  return (((1.0 + 0.5 / n) + 1.0 / ((6.0 * n) * n)) + ser) / n;
}

// gsl_sf_hzeta_e(s, q), Euler-Maclaurin branch (src/VP_zeta.c:775-803), s = 3..7, q >= 3
__device__ __noinline__ double hzeta_int(int si, double q)
{
  const double s = (double)si;
  const int jmax = 12, kmax = 10;
  const double pmax = ed_ppown(kmax + q, si);
  double scp = s;
  double pcp = pmax / (kmax + q);
  double ans = pmax * ((kmax + q) / (s - 1.0) + 0.5);
  for (int k = 0; k < kmax; k++) ans += ed_ppown(k + q, si);
  for (int j = 0; j <= jmax; j++) {
    const double delta = (k_hzeta_c[j + 1] * scp) * pcp;
    ans += delta;
    if (fabs(delta / ans) < 0.5 * EDSF_DBL_EPS) break;
    scp *= ((s + 2 * j) + 1) * ((s + 2 * j) + 2);
    pcp /= (kmax + q) * (kmax + q);
  }
  return ans;
}

// gsl_sf_psi_n_e for n >= 2 (src/VP_psi.c:790-821)
__device__ __forceinline__ double psi_n(int n, double x)
{
  const double v = hzeta_int(n + 1, x) * ed_pexp(lnfact_u((unsigned)n));
  return (n % 2 == 0) ? -v : v;
}

// ---- log|Gamma| next to a pole, x = -N + eps (what gsl reaches at src/VP_gamma.c:795-894) ----------------------------------------------
// Two expansions in eps, written here as tables + the two polynomial walkers below:
//   N = 1    Gamma(-1 + eps) eps = -(1 + eps/2 (1 + 3 eps)/(1 - eps^2)) + eps P(eps), P of degree 9 (k_pole1)
//   N >= 2   log|Gamma| = -[ log N! - eps psi(N+1) + eps^2 psi'(N+1)/2! - ... ] - log( sin(pi eps)/(pi eps) ) - log|eps|:
//            the bracket is an ALTERNATING Horner sum over c_m = psi^(m-1)(N+1) / (m-1)!, m = 1..7, where the polygamma of order n >= 2 enters
//            only once |eps| exceeds k_pole_psi_from[n - 2] (below that its term is under the rounding of the sum); the sine ratio is a
//            degree-5 polynomial in eps^2 (k_pole_sinc).
// The ORDER of the floating-point operations is the checker's (oracle/edo_gsl.inc), hence the reference's: the fixture
// tests/golden/sf_ref_negative.npz and test_lnbeta_outside_the_positive_quadrant_matches_the_checker hold it to the bit.
__constant__ const double k_pole1[10] = {0.07721566490153286061, 0.08815966957356030521, -0.00436125434555340577, 0.01391065882004640689,
                                         -0.00409427227680839100, 0.00275661310191541584, -0.00124162645565305019, 0.00065267976121802783,
                                         -0.00032205261682710437, 0.00016229131039545456};
__constant__ const double k_pole_sinc[6] = {1.0, -1.6449340668482264365, 0.8117424252833536436, -0.1907518241220842137, 0.0261478478176548005,
                                            -0.0023460810354558236};
__constant__ const double k_pole_psi_from[5] = {0.00001, 0.0002, 0.001, 0.005, 0.01};      // |eps| above which psi^(2) .. psi^(6) take part
__constant__ const double k_pole_fact[5] = {6.0, 24.0, 120.0, 720.0, 5040.0};             // 3! .. 7!

// c[0] + x (c[1] + x (... + x c[n-1]))
__device__ __forceinline__ double poly_up(const double* __restrict__ c, int n, double x)
{
  double h = c[n - 1];
  for (int i = n - 2; i >= 0; --i) h = c[i] + x * h;
  return h;
}
// c[0] - x (c[1] - x (... - x c[n-1]))
__device__ __forceinline__ double poly_alt(const double* c, int n, double x)
{
  double h = c[n - 1];
  for (int i = n - 2; i >= 0; --i) h = c[i] - x * h;
  return h;
}

// Returns the value; *site = 4 on eps == 0 (the pole itself: gsl's domain error at :803).
__device__ __noinline__ double lngamma_sgn_sing(int N, double eps, double* sgn, unsigned* site)
{
  if (eps == 0.0) { *sgn = 0.0; *site = 4u; return 0.0; }
  const double mag = fabs(eps);
  if (N == 1) {
    const double tail = eps * poly_up(k_pole1, 10, eps);
    const double gam_e = (tail - 1.0) - ((0.5 * eps) * (1.0 + 3.0 * eps)) / (1.0 - eps * eps);
    *sgn = (eps > 0.0 ? -1.0 : 1.0);
    return ed_plog(fabs(gam_e) / mag);
  }
  const double sinc = poly_up(k_pole_sinc, 6, eps * eps);
  double c[8];
  c[0] = lnfact_u((unsigned)N);
  c[1] = psi_int(N + 1);
  c[2] = psi_1_int(N + 1) / 2.0;
  for (int n = 2; n <= 6; ++n) c[n + 1] = (mag > k_pole_psi_from[n - 2] ? psi_n(n, N + 1.0) : 0.0) / k_pole_fact[n - 2];
  const double g = -poly_alt(c, 8, eps) - ed_plog(sinc);
  *sgn = ((N & 1) ? -1.0 : 1.0) * (eps > 0.0 ? 1.0 : -1.0);
  return g - ed_plog(mag);
}

// gsl_sf_lngamma_sgn_e (src/VP_gamma.c:1219-1285).  *site: the gsl_error() call the reference makes (0 none,
// 1 VP_gamma.c:1283, 2 :1239, 3 :1253, 4 :803, 5 :1261); values on those sites as the reference leaves them.
__device__ __noinline__ double lngamma_sgn_any(double x, double* sgn, unsigned* site)
{
  *site = 0u;
  *sgn = 1.0;
  if (fabs(x - 1.0) < 0.01) return lngamma_pade(x - 1.0, 0);
  if (fabs(x - 2.0) < 0.01) return lngamma_pade(x - 2.0, 1);
  if (x >= 0.5) return lngamma_lanczos(x);
  if (x == 0.0) { *sgn = 0.0; *site = 2u; return ed_pm_nan(); }
  if (fabs(x) < 0.02) {                                     // lngamma_sgn_0 (:761-787), either sign of x
    *sgn = (x >= 0.0) ? 1.0 : -1.0;
    if (x > 0.0) return lngamma_below_half(x, false);
    const double c1 = -0.07721566490153286061, c2 = -0.01094400467202744461, c3 = 0.09252092391911371098,
                 c4 = -0.01827191316559981266, c5 = 0.01800493109685479790, c6 = -0.00685088537872380685,
                 c7 = 0.00399823955756846603, c8 = -0.00189430621687107802, c9 = 0.00097473237804513221,
                 c10 = -0.00048434392722255893;
    const double g6 = c6 + x * (c7 + x * (c8 + x * (c9 + x * c10)));
    const double g = x * (c1 + x * (c2 + x * (c3 + x * (c4 + x * (c5 + x * g6)))));
    const double gee = (g + 1.0 / (1.0 + x)) + 0.5 * x;
    return ed_plog(gee / fabs(x));
  }
  if (x > -0.5 / (EDSF_DBL_EPS * EDSF_M_PI)) {
    if (x > 0.0) return lngamma_below_half(x, false);      // 0.02 <= x < 0.5: |sin(pi x)| > 0.015 pi, sign +
    const double z = 1.0 - x;
    const double s = ed_psin_any(EDSF_M_PI * x);
    const double as = fabs(s);
    if (s == 0.0) { *sgn = 0.0; *site = 3u; return ed_pm_nan(); }
    if (as < EDSF_M_PI * 0.015) {
      if (x < -2147483648.0 + 2.0) { *sgn = 0.0; *site = 5u; return 0.0; }
      const int N = -(int)(x - 0.5);
      const double eps = x + N;
      return lngamma_sgn_sing(N, eps, sgn, site);
    }
    *sgn = (s > 0.0 ? 1.0 : -1.0);
    return EDSF_M_LNPI - (ed_plog(as) + lngamma_lanczos(z));
  }
  *sgn = 0.0; *site = 1u;                                   // |x| too large, or NaN (:1278-1283)
  return 0.0;
}

// Everything lnbeta can be asked outside x > 0, y > 0 (zero, negative, NaN arguments): value semantics of the reference's
// natural-prototype wrapper (src/beta.c:161-164, src/eval.h:3-9) over gsl_sf_lnbeta_e (:38-47) and gsl_sf_lnbeta_sgn_e
// (:49-114, general route :101-112):
//   x == 0 or y == 0, or a negative integer      -> NaN (domain error :56 / :59)
//   otherwise lgamma(x) + lgamma(y) - lgamma(x + y) through gsl_sf_lngamma_sgn_e; a NaN argument contributes 0.0 (its
//   comparisons all fail and the EROUND exit returns 0), so lnbeta(NaN, NaN) = 0.0;  B(x, y) < 0 -> NaN (:43-45)
// *sites: which gsl_error() calls the reference makes -- what it prints through Rprintf (src/error.c:45-48):
//   bits 0-2, 3-5, 6-8  the site inside gsl_sf_lngamma_sgn_e for x, y, x+y (codes of lngamma_sgn_any)
//   9   beta.c:56     10  beta.c:59     11  beta.c:44
// Any bit set => the wrapper adds beta.c:163 and the call counts as one GSL error event.
enum : unsigned { kSiteB56 = 1u << 9, kSiteB59 = 1u << 10, kSiteB44 = 1u << 11 };
__device__ EDSF_COLD double lnbeta_cold_sites(double x, double y, unsigned* sites)
{
  if (x == 0.0 || y == 0.0) { *sites = kSiteB56; return ed_pm_nan(); }
  if ((x < 0.0 && x == __builtin_floor(x)) || (y < 0.0 && y == __builtin_floor(y))) { *sites = kSiteB59; return ed_pm_nan(); }
  double sx, sy, sxy;
  unsigned e1, e2, e3;
  const double lgx = lngamma_sgn_any(x, &sx, &e1);
  const double lgy = lngamma_sgn_any(y, &sy, &e2);
  const double lgxy = lngamma_sgn_any(x + y, &sxy, &e3);
  unsigned code = e1 | (e2 << 3) | (e3 << 6);
  double v = (lgx + lgy) - lgxy;
  if ((sx * sy) * sxy == -1.0) { code |= kSiteB44; v = ed_pm_nan(); }
  *sites = code;
  return v;
}
__device__ __forceinline__ double lnbeta_cold(double x, double y, int* flag)
{
  unsigned sites;
  const double v = lnbeta_cold_sites(x, y, &sites);
  *flag = sites ? 1 : 0;
  return v;
}
// the sites alone, for any (x, y)
__device__ __forceinline__ unsigned lnbeta_sites(double x, double y)
{
  if (x > 0.0 && y > 0.0) return 0u;     // both routes of the positive quadrant are error-free
  unsigned sites;
  (void)lnbeta_cold_sites(x, y, &sites);
  return sites;
}

// The two value-exact evaluation routes of log B for x,y > 0 (src/beta.c:62-113), split so that a
// kernel can bin its tasks by route and run each route in branch-uniform waves:
//   ratio route   min/max < 0.2 : Gamma* form (:69-97).  Only symmetric expressions of (x,y) occur
//                 (gsx*gsy, x+y), so the route takes (mn, mx, rat) and needs no memory of the order.
//   general route otherwise     : lgamma(x) + lgamma(y) - lgamma(x+y) (:101-112), again symmetric.
__device__ __forceinline__ double lnbeta_ratio(double mn, double mx, double rat)
{
  const double gsa = gammastar_pos(mn);
  const double gsb = gammastar_pos(mx);
  const double gsxy = gammastar_pos(mn + mx);
  const double lnopr = log1plusx_ratio(rat);
  // Gamma* of a positive double lies in (0.9, 2e161): product and quotient stay in fdiv's range
  const double lnpre = plog_fast((fdiv(gsa * gsb, gsxy) * EDSF_M_SQRT2) * EDSF_M_SQRTPI);
  const double t1 = mn * plog_fast(rat);
  const double t2 = 0.5 * plog_fast(mn);
  const double t3 = ((mn + mx) - 0.5) * lnopr;
  return lnpre + ((t1 - t2) - t3);
}

// The same with Gamma*(mn) and log(mn) handed in (g > 0: g = Gamma*(mn), l = log(mn)), or with Gamma*(mx) handed in
// (g < 0: -g = Gamma*(mx)), or with nothing known (g NaN).  Gamma* of a positive argument is positive, so the sign
// is free to carry that bit.  The values must come from gammastar_pos / plog_fast themselves: same bits.
__device__ __forceinline__ double lnbeta_ratio_pre(double mn, double mx, double rat, double g, double l,
                                                   const double* LT = nullptr)
{
  // Order of evaluation: everything that does not need the handed-in (g, l) comes first -- in k_emit_batch they are a
  // table gather issued just before this call, and the ~150 instructions of log1p, Gamma*(mn + mx) and log(rat) are what
  // hides its latency.  (The operations and their operands are those of lnbeta_ratio; only their order in time differs.)
  const double lnopr = log1plusx_ratio(rat);
  const double gsxy = gammastar_pos(mn + mx);
  const double t3 = ((mn + mx) - 0.5) * lnopr;
  // mn >= 1e-100 and mx <= 1e100 (one test per task) make rat, mn and the Gamma* quotient normal numbers (Gamma* of such an
  // argument lies in (0.9, 1e51)): the three logarithms skip their own range tests.  Same function, same bits.
  const bool plain = (mn >= 1e-100 && mx <= 1e100);
  const double t1 = mn * (plain ? plog_pos(rat, LT) : plog_fast(rat, LT));
  // ---- from here on (g, l) are needed ----
  const bool have_mn = g > 0.0;
  const double gsa = have_mn ? g : gammastar_pos(mn);
  const double gsb = (g < 0.0) ? -g : gammastar_pos(mx);
  const double pre = (fdiv(gsa * gsb, gsxy) * EDSF_M_SQRT2) * EDSF_M_SQRTPI;
  if (plain) {
    const double lnpre = plog_pos(pre, LT);
    const double t2 = 0.5 * (have_mn ? l : plog_pos(mn, LT));
    return lnpre + ((t1 - t2) - t3);
  }
  const double lnpre = plog_fast(pre, LT);
  const double t2 = 0.5 * (have_mn ? l : plog_fast(mn, LT));
  return lnpre + ((t1 - t2) - t3);
}

// +inf arguments reach this route with rat = NaN; the arithmetic then yields NaN as the reference's does.
// (lgx handed in, from lngamma_pos(x, false) itself; NaN = not known)
__device__ __forceinline__ double lnbeta_general_pre(double x, double y, double lgx_in, const double* LT = nullptr)
{
  // the two terms that do not need the handed-in value first: they hide the latency of the gather it comes from
  const double lgy = lngamma_pos(y, false, LT);
  const double lgxy = lngamma_pos(x + y, false, LT);
  const double lgx = (lgx_in == lgx_in) ? lgx_in : lngamma_pos(x, false, LT);
  return (lgx + lgy) - lgxy;
}

// +inf arguments reach this route with rat = NaN; the arithmetic then yields NaN as the reference's does.
__device__ __forceinline__ double lnbeta_general(double x, double y)
{
  const double lgx = lngamma_pos(x, false);
  const double lgy = lngamma_pos(y, false);
  const double lgxy = lngamma_pos(x + y, false);
  return (lgx + lgy) - lgxy;
}

// log B(x,y) with the reference's branch selection (src/beta.c:62-113).
__device__ __forceinline__ double lnbeta(double x, double y, int* flag)
{
  if (!(x > 0.0 && y > 0.0)) return lnbeta_cold(x, y, flag);
  const double mx = (x > y ? x : y);
  const double mn = (x < y ? x : y);
  const double rat = mn / mx;
  if (rat < 0.2) return lnbeta_ratio(mn, mx, rat);
  return lnbeta_general(x, y);
}

}  // namespace edsf
