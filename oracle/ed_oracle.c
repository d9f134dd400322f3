/* ed_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU checker ("oracle") for the MI355X CNV-calling core.  It is a plain-C restatement of the
 * reference's algorithm for the hot path, written from the reference's source with the file:line of
 * every restated function cited next to it:
 *     emissions      reference src/CNV_estimate.cpp:44-85 -> src/beta.c -> src/VP_gamma.c, src/VP_log.c
 *     Viterbi/calls  reference src/hmm.cpp:18-167, R/tools.R:88-103, R/class_definition.R:343-374
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (exomedepth_amd/, libedcore.so) never does.
 *
 * Parity status.  The special-function layer is PINNED: built in its "libm" flavour it reproduces,
 * bit for bit, the reference's own C sources compiled as they lie into oracle/_ref/libgslsf_ref.so
 * (tests/test_oracle_ref.py; golden vectors generated from that build are committed under
 * tests/golden/).  hmm.cpp and CNV_estimate.cpp themselves cannot be compiled here (they include
 * <Rinternals.h>; R is absent and no stand-in headers are written), so the ~10 lines of arithmetic
 * in myprob/get_loglike_matrix and the Viterbi are restated by reading and pinned by the
 * known-answer example the reference documents (R/tools.R:74-85) and by brute-force optimality
 * checks.  The dispersion fit has no in-tree reference (aod::betabin is third-party): parity unpinned.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <dlfcn.h>
#include <time.h>

#include "../exomedepth_amd/csrc/ed_pmath.h"
#include "../exomedepth_amd/csrc/ed_dtab.h"   /* the table-driven emission mode's entry definition (host/device header) */
#include "../exomedepth_amd/csrc/ed_sing_tables.h"

#define EDO_SUCCESS 0
#define EDO_EDOM 1      /* GSL_EDOM   */
#define EDO_EROUND 18   /* GSL_EROUND */

/* constants as spelled in reference src/gsl_math.h and src/gsl_machine.h */
#define EDO_M_E 2.71828182845904523536028747135
#define EDO_M_PI 3.14159265358979323846264338328
#define EDO_M_SQRT2 1.41421356237309504880168872421
#define EDO_M_SQRTPI 1.77245385090551602729816748334
#define EDO_M_LN2 0.69314718055994530941723212146
#define EDO_M_LNPI 1.14472988584940017414342735135
#define EDO_LOGROOT2PI 0.9189385332046727418 /* src/VP_gamma.c:71 */
#define EDO_DBL_EPSILON 2.2204460492503131e-16
#define EDO_ROOT4_DBL_EPSILON 1.2207031250000000e-04
#define EDO_ROOT6_DBL_EPSILON 2.4607833005759251e-03

/* Chebyshev / Lanczos coefficient tables (mathematical constants of the GSL special-function
 * library, as listed at reference src/VP_gamma.c:594-688 and src/VP_log.c:73-95). */
static const double edo_gstar_a[30] = {
  2.16786447866463034423060819465, -0.05533249018745584258035832802, 0.01800392431460719960888319748,
  -0.00580919269468937714480019814, 0.00186523689488400339978881560, -0.00059746524113955531852595159,
  0.00019125169907783353925426722, -0.00006124996546944685735909697, 0.00001963889633130842586440945,
  -6.3067741254637180272515795142e-06, 2.0288698405861392526872789863e-06, -6.5384896660838465981983750582e-07,
  2.1108698058908865476480734911e-07, -6.8260714912274941677892994580e-08, 2.2108560875880560555583978510e-08,
  -7.1710331930255456643627187187e-09, 2.3290892983985406754602564745e-09, -7.5740371598505586754890405359e-10,
  2.4658267222594334398525312084e-10, -8.0362243171659883803428749516e-11, 2.6215616826341594653521346229e-11,
  -8.5596155025948750540420068109e-12, 2.7970831499487963614315315444e-12, -9.1471771211886202805502562414e-13,
  2.9934720198063397094916415927e-13, -9.8026575909753445931073620469e-14, 3.2116773667767153777571410671e-14,
  -1.0518035333878147029650507254e-14, 3.4144405720185253938994854173e-15, -1.0115153943081187052322643819e-15};
static const double edo_gstar_b[30] = {
  0.0057502277273114339831606096782, 0.0004496689534965685038254147807, -0.0001672763153188717308905047405,
  0.0000615137014913154794776670946, -0.0000223726551711525016380862195, 8.0507405356647954540694800545e-06,
  -2.8671077107583395569766746448e-06, 1.0106727053742747568362254106e-06, -3.5265558477595061262310873482e-07,
  1.2179216046419401193247254591e-07, -4.1619640180795366971160162267e-08, 1.4066283500795206892487241294e-08,
  -4.6982570380537099016106141654e-09, 1.5491248664620612686423108936e-09, -5.0340936319394885789686867772e-10,
  1.6084448673736032249959475006e-10, -5.0349733196835456497619787559e-11, 1.5357154939762136997591808461e-11,
  -4.5233809655775649997667176224e-12, 1.2664429179254447281068538964e-12, -3.2648287937449326771785041692e-13,
  7.1528272726086133795579071407e-14, -9.4831735252566034505739531258e-15, -2.3124001991413207293120906691e-15,
  2.8406613277170391482590129474e-15, -1.7245370321618816421281770927e-15, 8.6507923128671112154695006592e-16,
  -3.9506563665427555895391869919e-16, 1.6779342132074761078792361165e-16, -6.0483153034414765129837716260e-17};
static const double edo_lopx[21] = {
  2.16647910664395270521272590407, -0.28565398551049742084877469679, 0.01517767255690553732382488171,
  -0.00200215904941415466274422081, 0.00019211375164056698287947962, -0.00002553258886105542567601400,
  2.9004512660400621301999384544e-06, -3.8873813517057343800270917900e-07, 4.7743678729400456026672697926e-08,
  -6.4501969776090319441714445454e-09, 8.2751976628812389601561347296e-10, -1.1260499376492049411710290413e-10,
  1.4844576692270934446023686322e-11, -2.0328515972462118942821556033e-12, 2.7291231220549214896095654769e-13,
  -3.7581977830387938294437434651e-14, 5.1107345870861673561462339876e-15, -7.0722150011433276578323272272e-16,
  9.7089758328248469219003866867e-17, -1.3492637457521938883731579510e-17, 1.8657327910677296608121390705e-18};
static const double edo_lanczos_7_c[9] = {
  0.99999999999980993227684700473478, 676.520368121885098567009190444019, -1259.13921672240287047156078755283,
  771.3234287776530788486528258894, -176.61502916214059906584551354, 12.507343278686904814458936853,
  -0.13857109526572011689554707, 9.984369578019570859563e-6, 1.50563273514931155834e-7};

/* n!, psi(n), psi'(n): mathematical constants generated by tools/gen_sing_tables.py (bitwise the reference's tables), and
 * the Euler-Maclaurin coefficients B_2j / (2j)! of src/VP_zeta.c:563-579 */
static const double edo_fact_table[ED_FACT_TABLE_N] = ED_FACT_TABLE;
static const double edo_psi_table[ED_PSI_TABLE_N] = ED_PSI_TABLE;
static const double edo_psi1_table[ED_PSI1_TABLE_N] = ED_PSI1_TABLE;
static const double edo_hzeta_c[15] = ED_HZETA_C;
static double edo_pown_libm(double b, int n) { return pow(b, -(double)n); }
static double edo_pown_port(double b, int n) { return ed_ppown(b, n); }

/* ---- flavour 1: libm transcendental functions (what the reference itself calls) ---- */
#define F(name) edo_libm_##name
#define EDO_LOG log
#define EDO_EXP exp
#define EDO_SIN sin
#define EDO_SIN_ANY sin
#define EDO_POWN edo_pown_libm
#define EDO_PORTABLE 0
#include "edo_gsl.inc"
#undef F
#undef EDO_LOG
#undef EDO_EXP
#undef EDO_SIN
#undef EDO_SIN_ANY
#undef EDO_POWN
#undef EDO_PORTABLE

/* ---- flavour 2: portable transcendental functions (bit-exact target of the HIP kernels) ---- */
#define F(name) edo_port_##name
#define EDO_LOG ed_plog
#define EDO_EXP ed_pexp
#define EDO_SIN ed_psin_0pi
#define EDO_SIN_ANY ed_psin_any
#define EDO_POWN edo_pown_port
#define EDO_PORTABLE 1
#include "edo_gsl.inc"
#undef F
#undef EDO_LOG
#undef EDO_EXP
#undef EDO_SIN
#undef EDO_SIN_ANY
#undef EDO_POWN
#undef EDO_PORTABLE

/* =====================================================================================
 * exported batch evaluators (ctypes-friendly).  flavour: 0 = libm, 1 = portable.
 * ===================================================================================== */
#define EDO_API __attribute__((visibility("default")))

EDO_API void edo_plog_v(long n, const double *x, double *out) { for (long i = 0; i < n; i++) out[i] = ed_plog(x[i]); }
EDO_API void edo_pexp_v(long n, const double *x, double *out) { for (long i = 0; i < n; i++) out[i] = ed_pexp(x[i]); }
EDO_API void edo_psin_v(long n, const double *x, double *out) { for (long i = 0; i < n; i++) out[i] = ed_psin_0pi(x[i]); }
EDO_API void edo_psin_any_v(long n, const double *x, double *out) { for (long i = 0; i < n; i++) out[i] = ed_psin_any(x[i]); }

/* the log-gamma difference tables of the table-driven emission mode (exomedepth_amd/csrc/ed_dtab.h), evaluated on the host:
 * what k_tab_build must hold.  edo_ddlog_v: the double-double logarithm behind them. */
EDO_API void edo_dtab(double x0, long n, double *out)
{
  static const double T[ED_PM_LOGT_N][3] = ED_PM_LOGT_ROWS;
  ed_dtab_fill_seq(x0, n, out, &T[0][0]);
}
EDO_API void edo_ddlog_v(long n, const double *x, double *hi, double *lo)
{
  static const double T[ED_PM_LOGT_N][3] = ED_PM_LOGT_ROWS;
  for (long i = 0; i < n; i++) { const ed_dd r = ed_ddlog_t(x[i], &T[0][0]); hi[i] = r.hi; lo[i] = r.lo; }
}
/* beyond the tables (ed_dtab.h: ed_dtab_lg0, ed_dtab_tail): lgamma(x0) as a double-double, and D(x0, k[i]) for counts k[i] >= 64 */
EDO_API void edo_dtab_lg0(double x0, double *hi, double *lo)
{
  static const double T[ED_PM_LOGT_N][3] = ED_PM_LOGT_ROWS;
  const ed_dd r = ed_dtab_lg0(x0, &T[0][0]);
  *hi = r.hi; *lo = r.lo;
}
EDO_API void edo_dtab_tail_v(double x0, long n, const double *k, double *out)
{
  static const double T[ED_PM_LOGT_N][3] = ED_PM_LOGT_ROWS;
  const ed_dd g = ed_dtab_lg0(x0, &T[0][0]);
  for (long i = 0; i < n; i++) out[i] = ed_dtab_tail(x0, g.hi, g.lo, k[i], &T[0][0]);
}
EDO_API void edo_dtab_combine_v(long n, const double *d1, const double *d2, const double *d3, double *out)
{
  for (long i = 0; i < n; i++) out[i] = ed_dtab_combine(d1[i], d2[i], d3[i]);
}

EDO_API long edo_lnbeta_v(int flavour, long n, const double *x, const double *y, double *out)
{
  long nerr = 0;
  for (long i = 0; i < n; i++)
    out[i] = flavour ? edo_port_lnbeta(x[i], y[i], &nerr) : edo_libm_lnbeta(x[i], y[i], &nerr);
  return nerr;
}

/* which: 0 lngamma (sgn variant), 1 gammastar, 2 log_1plusx, 3 lngamma (gsl_sf_lngamma_e variant) */
EDO_API void edo_sf_v(int flavour, int which, long n, const double *x, double *out, int *status)
{
  for (long i = 0; i < n; i++) {
    double v = NAN, sg;
    int st = 0;
    if (which == 0) st = flavour ? edo_port_lngamma_sgn(x[i], 0, &v, &sg) : edo_libm_lngamma_sgn(x[i], 0, &v, &sg);
    else if (which == 1) st = flavour ? edo_port_gammastar(x[i], &v) : edo_libm_gammastar(x[i], &v);
    else if (which == 2) st = flavour ? edo_port_log_1plusx(x[i], &v) : edo_libm_log_1plusx(x[i], &v);
    else if (which == 3) st = flavour ? edo_port_lngamma_sgn(x[i], 1, &v, &sg) : edo_libm_lngamma_sgn(x[i], 1, &v, &sg);
    out[i] = v;
    if (status) status[i] = st;
  }
}

/* reference src/CNV_estimate.cpp:52-85; out n x 3 column-major (deletion, normal, duplication) */
EDO_API long edo_get_loglike_matrix(int flavour, const double *phi, const double *expected, const int *total,
                                    const int *observed, long n, double mixture, double *out)
{
  return flavour ? edo_port_get_loglike_matrix(phi, expected, total, observed, n, mixture, out)
                 : edo_libm_get_loglike_matrix(phi, expected, total, observed, n, mixture, out);
}

/* =====================================================================================
 * Viterbi + trace-back + run-length call table: reference src/hmm.cpp:18-167.
 *   transitions  3x3 column-major (trans[j*3+k] = P(from k -> into j))            :25, :74-76
 *   proba        nobs x 3 column-major in HMM order (normal, deletion, duplication) :26, :79
 *   positions    int[nobs]; expected_len = expected CNV length                      :29-30, :62-64
 *   path_out     double[nobs] in {0,1,2}                                            :139-141
 *   calls_out    double[max_calls x 4] ROW-major here: (start.p, end.p, type, nexons), 1-based :111-121
 * Returns the number of calls (may exceed max_calls; only max_calls rows are written), or -1 when
 * nstates != 3 (the reference prints and returns a C NULL, :37-40).
 * Defined behaviour where the reference has none: a back-pointer that no candidate sets (all
 * candidates -inf or NaN) is 0 instead of the reference's -1, which the reference would then use as
 * an out-of-bounds index in the trace-back (:46, :99).  Every case in which the reference is defined
 * is unchanged, because a -1 back-pointer can only be followed after such a step.
 * ===================================================================================== */
EDO_API long edo_hmm(int nstates, long nobs, const double *trans_c, const double *proba_c, const int *locations,
                     double Expected, double *path_out, double *calls_out, long max_calls)
{
  if (nstates != 3) return -1;
  if (nobs <= 0) return 0;
  unsigned char *from = (unsigned char *)malloc((size_t)nobs * 3);
  double vit_prev[3] = {0., -HUGE_VAL, -HUGE_VAL}, vit[3], trans[3];
  from[0] = from[1] = from[2] = 0;
  for (long i = 1; i < nobs; i++) {
    double dist = (double)locations[i] - (double)locations[i - 1];
    double dist_effect = exp(-dist / Expected);
    for (int j = 0; j < 3; j++) {
      vit[j] = -HUGE_VAL;
      int fw = 0;
      trans[0] = trans_c[j * 3];
      trans[1] = dist_effect * trans_c[j * 3 + 1] + (1.0 - dist_effect) * trans_c[j * 3];
      trans[2] = dist_effect * trans_c[j * 3 + 2] + (1.0 - dist_effect) * trans_c[j * 3];
      for (int k = 0; k < 3; k++) {
        double newp = proba_c[j * nobs + i] + vit_prev[k] + log(trans[k]);
        if (newp > vit[j]) {
          vit[j] = newp;
          fw = k;
        }
      }
      if (proba_c[j * nobs + i] == -HUGE_VAL) fw = 0;
      from[i * 3 + j] = (unsigned char)fw;
    }
    vit_prev[0] = vit[0]; vit_prev[1] = vit[1]; vit_prev[2] = vit[2];
  }
  /* trace back, :95-100: the last observation is forced into state 0 */
  int *tb = (int *)malloc(sizeof(int) * (size_t)nobs);
  tb[nobs - 1] = 0;
  for (long i = 1; i < nobs; i++) tb[nobs - i - 1] = from[(nobs - i) * 3 + tb[nobs - i]];
  /* run-length summary, :104-126, with its quirks kept: `start` is only set when leaving state 0,
   * `nexons` is only reset when a call is pushed */
  double start = -1., end = -1., nexons = 0;
  int current = 0;
  long ncalls = 0;
  for (long i = 1; i < nobs; i++) {
    if (tb[i - 1] != tb[i]) {
      if (current == 0) start = (double)i;
      if (current != 0) {
        end = (double)(i - 1);
        if (ncalls < max_calls && calls_out) {
          calls_out[ncalls * 4 + 0] = start + 1;
          calls_out[ncalls * 4 + 1] = end + 1;
          calls_out[ncalls * 4 + 2] = current;
          calls_out[ncalls * 4 + 3] = nexons;
        }
        ncalls++;
        nexons = 0;
      }
    }
    if (tb[i] != 0) nexons++;
    current = tb[i];
  }
  if (path_out) for (long i = 0; i < nobs; i++) path_out[i] = tb[i];
  free(tb);
  free(from);
  return ncalls;
}

/* =====================================================================================
 * Per-sample CallCNVs driver: reference R/class_definition.R:343-374 + :408-414.
 * Input exons are already ordered by (chromosome, midpoint) with chrom_off[c]..chrom_off[c+1]
 * delimiting chromosome c (the ordering itself, :323-336, is host logic tested separately).
 *   likelihood  n x 3 column-major (deletion, normal, duplication)  -- the S4 slot layout
 *   path_out    int8[n]: HMM state per exon (0 normal, 1 deletion, 2 duplication), dummies stripped
 *   calls_out   double[max_calls x 4] row-major: start.p, end.p (1-based, global = chromosome-local
 *               minus the dummy, plus shift), type (1 deletion, 2 duplication), nexons
 * ===================================================================================== */
EDO_API long edo_callcnvs(const double *likelihood, long n, const int *chrom_off, int nchrom, const int *start,
                          const int *end, double transition_probability, double expected_cnv_length,
                          signed char *path_out, double *calls_out, long max_calls)
{
  const double t = transition_probability;
  /* matrix(c(1-t, t/2, t/2, .5,.5,0, .5,0,.5), byrow=TRUE) stored column-major as R does (:343-347) */
  double T[9];
  const double rows[3][3] = {{1. - t, t / 2., t / 2.}, {0.5, 0.5, 0.}, {0.5, 0., 0.5}};
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) T[c * 3 + r] = rows[r][c];
  long ncalls = 0;
  for (int c = 0; c < nchrom; c++) {
    long lo = chrom_off[c], hi = chrom_off[c + 1], m = hi - lo;
    if (m <= 0) continue;
    long nobs = m + 2;
    double *ll = (double *)malloc(sizeof(double) * (size_t)nobs * 3);
    int *pos = (int *)malloc(sizeof(int) * (size_t)nobs);
    double *path = (double *)malloc(sizeof(double) * (size_t)nobs);
    /* rbind(c(-Inf,0,-Inf), likelihood[good.pos, c(2,1,3)], c(-100,0,-100))  (:364) */
    ll[0] = -HUGE_VAL; ll[nobs] = 0.; ll[2 * nobs] = -HUGE_VAL;
    for (long i = 0; i < m; i++) {
      ll[1 + i] = likelihood[(lo + i) + n * 1];
      ll[nobs + 1 + i] = likelihood[(lo + i) + n * 0];
      ll[2 * nobs + 1 + i] = likelihood[(lo + i) + n * 2];
    }
    ll[nobs - 1] = -100.; ll[2 * nobs - 1] = 0.; ll[3 * nobs - 1] = -100.;
    /* as.integer(c(positions[1] - 2*L, positions, end.positions[last] + 2*L))  (:368) */
    pos[0] = (int)((double)start[lo] - 2 * expected_cnv_length);
    for (long i = 0; i < m; i++) pos[1 + i] = start[lo + i];
    pos[nobs - 1] = (int)((double)end[hi - 1] + 2 * expected_cnv_length);
    long room = max_calls - ncalls; if (room < 0) room = 0;
    long nc = edo_hmm(3, nobs, T, ll, pos, expected_cnv_length, path,
                      calls_out ? calls_out + 4 * (ncalls < max_calls ? ncalls : max_calls) : NULL, room);
    for (long i = 0; i < m; i++) path_out[lo + i] = (signed char)path[1 + i];
    for (long r = ncalls; r < ncalls + nc && r < max_calls; r++) {
      calls_out[r * 4 + 0] += -1 + (double)lo; /* start.p - 1 + shift (:371, :409) */
      calls_out[r * 4 + 1] += -1 + (double)lo;
    }
    ncalls += nc;
    free(ll); free(pos); free(path);
  }
  return ncalls;
}

/* =====================================================================================
 * Decision margins of the Viterbi path (diagnostic; tools/concordance.py --margins, bench.py's verify).  The forward pass above keeps, per
 * observation and target state, the FIRST strict maximum of three candidates (reference src/hmm.cpp:78-85); an implementation whose emissions
 * differ from the reference's in the last bits -- the table-driven emission mode: <= 2e-13 relative -- decodes the same path as long as no
 * decision ON the decoded path is closer than what those differences can move the candidates.  For every observation i >= 1 and the state the
 * path is in at i: margin = best candidate - runner-up (0 = an exact tie, resolved by the first-maximum rule on both sides as long as it stays
 * exact; decisions with a single finite candidate, or forced by a -inf emission (:87), have none).
 *   likelihood / chrom_off / start / end / transition_probability / expected_cnv_length: as edo_callcnvs
 *   thresholds[n_thr] ascending; below[t] += on-path decisions with margin < thresholds[t]; *ties += margin == 0; *decisions += decisions with
 *   two finite candidates; *min_margin = the smallest non-zero margin seen (start it at +inf); *scale_at_min = |best candidate| at that decision
 * ===================================================================================== */
EDO_API void edo_callcnvs_margins(const double *likelihood, long n, const int *chrom_off, int nchrom, const int *start, const int *end,
                                  double transition_probability, double expected_cnv_length, const double *thresholds, int n_thr, long *below,
                                  long *ties, long *decisions, double *min_margin, double *scale_at_min)
{
  const double t = transition_probability;
  double T[9];
  const double rows[3][3] = {{1. - t, t / 2., t / 2.}, {0.5, 0.5, 0.}, {0.5, 0., 0.5}};
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) T[c * 3 + r] = rows[r][c];
  for (int c = 0; c < nchrom; c++) {
    long lo = chrom_off[c], hi = chrom_off[c + 1], m = hi - lo;
    if (m <= 0) continue;
    long nobs = m + 2;
    unsigned char *from = (unsigned char *)malloc((size_t)nobs * 3);
    double *marg = (double *)malloc(sizeof(double) * (size_t)nobs * 3);
    double *mag = (double *)malloc(sizeof(double) * (size_t)nobs * 3);
    int *tb = (int *)malloc(sizeof(int) * (size_t)nobs);
    double vit_prev[3] = {0., -HUGE_VAL, -HUGE_VAL}, vit[3], trans[3];
    int prev_pos = (int)((double)start[lo] - 2 * expected_cnv_length);
    from[0] = from[1] = from[2] = 0;
    for (long i = 1; i < nobs; i++) {
      /* the padded chain of R/class_definition.R:364-368, row by row */
      int pos = (i <= m) ? start[lo + i - 1] : (int)((double)end[hi - 1] + 2 * expected_cnv_length);
      double e[3];
      if (i <= m) { e[0] = likelihood[(lo + i - 1) + n * 1]; e[1] = likelihood[(lo + i - 1) + n * 0]; e[2] = likelihood[(lo + i - 1) + n * 2]; }
      else { e[0] = 0.; e[1] = -100.; e[2] = -100.; }
      double dist = (double)pos - (double)prev_pos;
      double dist_effect = exp(-dist / expected_cnv_length);
      prev_pos = pos;
      for (int j = 0; j < 3; j++) {
        vit[j] = -HUGE_VAL;
        double second = -HUGE_VAL;
        int fw = 0;
        trans[0] = T[j * 3];
        trans[1] = dist_effect * T[j * 3 + 1] + (1.0 - dist_effect) * T[j * 3];
        trans[2] = dist_effect * T[j * 3 + 2] + (1.0 - dist_effect) * T[j * 3];
        for (int k = 0; k < 3; k++) {
          double newp = e[j] + vit_prev[k] + log(trans[k]);
          if (newp > vit[j]) { second = vit[j]; vit[j] = newp; fw = k; }
          else if (newp > second) second = newp;
        }
        if (e[j] == -HUGE_VAL) fw = 0;
        from[i * 3 + j] = (unsigned char)fw;
        marg[i * 3 + j] = (second > -HUGE_VAL && e[j] > -HUGE_VAL) ? vit[j] - second : HUGE_VAL;
        mag[i * 3 + j] = fabs(vit[j]);
      }
      vit_prev[0] = vit[0]; vit_prev[1] = vit[1]; vit_prev[2] = vit[2];
    }
    tb[nobs - 1] = 0;
    for (long i = 1; i < nobs; i++) tb[nobs - i - 1] = from[(nobs - i) * 3 + tb[nobs - i]];
    for (long i = 1; i < nobs; i++) {
      const double mg = marg[i * 3 + tb[i]];
      if (!(mg < HUGE_VAL)) continue;
      ++*decisions;
      if (mg == 0.0) { ++*ties; continue; }
      for (int q = 0; q < n_thr; q++) if (mg < thresholds[q]) ++below[q];
      if (mg < *min_margin) { *min_margin = mg; *scale_at_min = mag[i * 3 + tb[i]]; }
    }
    free(from); free(marg); free(mag); free(tb);
  }
}

#include "edo_fit.inc"

/* =====================================================================================
 * Loader for oracle/_ref/libgslsf_ref.so -- the reference's own special-function sources compiled
 * as they lie (oracle/Makefile).  That library has exactly one unresolved symbol, gsl_error, whose
 * definition (reference src/error.c) needs R's headers and therefore cannot be built here; no
 * stand-in is written.  The library is opened with RTLD_LAZY so the symbol is only needed if an
 * error path is taken: the _ref functions may be called on their error-free domain only.
 * ===================================================================================== */
static void *edo_ref_handle = NULL;
EDO_API int edo_ref_open(const char *path)
{
  if (edo_ref_handle) return 0;
  edo_ref_handle = dlopen(path, RTLD_LAZY | RTLD_LOCAL);
  return edo_ref_handle ? 0 : -1;
}
/* name: a one-argument function of the reference build, e.g. "gsl_sf_lngamma", "gsl_sf_gammastar",
 * "gsl_sf_log_1plusx", "gsl_sf_psi", "gsl_sf_psi_1" */
EDO_API int edo_ref_call1(const char *name, long n, const double *x, double *out)
{
  if (!edo_ref_handle) return -1;
  double (*fn)(double) = (double (*)(double))dlsym(edo_ref_handle, name);
  if (!fn) return -2;
  for (long i = 0; i < n; i++) out[i] = fn(x[i]);
  return 0;
}
/* two-argument functions, e.g. "gsl_sf_lnbeta" */
EDO_API int edo_ref_call2(const char *name, long n, const double *x, const double *y, double *out)
{
  if (!edo_ref_handle) return -1;
  double (*fn)(double, double) = (double (*)(double, double))dlsym(edo_ref_handle, name);
  if (!fn) return -2;
  for (long i = 0; i < n; i++) out[i] = fn(x[i], y[i]);
  return 0;
}

/* gsl_sf_lngamma_sgn_e(x, &result, &sgn) of the reference build (error-free arguments only: see above) */
EDO_API int edo_ref_lngamma_sgn(long n, const double *x, double *val, double *sgn, int *status)
{
  if (!edo_ref_handle) return -1;
  struct res { double val, err; };
  int (*fn)(double, struct res *, double *) = (int (*)(double, struct res *, double *))dlsym(edo_ref_handle, "gsl_sf_lngamma_sgn_e");
  if (!fn) return -2;
  for (long i = 0; i < n; i++) {
    struct res r = {0, 0};
    double sg = 0;
    status[i] = fn(x[i], &r, &sg);
    val[i] = r.val; sgn[i] = sg;
  }
  return 0;
}

/* the checker's gsl_sf_lngamma_sgn_e: value, sign, status */
EDO_API void edo_lngamma_sgn_v(int flavour, long n, const double *x, double *val, double *sgn, int *status)
{
  for (long i = 0; i < n; i++)
    status[i] = flavour ? edo_port_lngamma_sgn(x[i], 0, &val[i], &sgn[i]) : edo_libm_lngamma_sgn(x[i], 0, &val[i], &sgn[i]);
}

/* gsl_sf_lnbeta with the error sites (codes of edsf::lnbeta_sites) */
EDO_API void edo_lnbeta_sites_v(int flavour, long n, const double *x, const double *y, double *val, int *sites)
{
  for (long i = 0; i < n; i++) {
    unsigned c = 0;
    if (flavour) edo_port_lnbeta_e_sites(x[i], y[i], &val[i], &c); else edo_libm_lnbeta_e_sites(x[i], y[i], &val[i], &c);
    sites[i] = (int)c;
  }
}

/* monotonic seconds, for bench.py's cpu_baseline leg */
EDO_API double edo_now(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
