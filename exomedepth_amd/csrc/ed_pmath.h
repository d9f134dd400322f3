/* ed_pmath.h -- portable, bit-reproducible log / exp / sin for binary64.
 *
 * Why this exists.  The hot path (beta-binomial emissions, reference src/CNV_estimate.cpp:44-85 ->
 * src/beta.c:49-114 -> src/VP_gamma.c) calls libm's log() and exp().  glibc's and ROCm's (ocml)
 * implementations differ in the last bit for a fraction of arguments, and the Viterbi forward pass
 * compares sums of emissions with a strict '>' (reference src/hmm.cpp:79-84), so a 1-ulp difference
 * can in principle flip a back-pointer.  To make "GPU result == CPU checker result" a bit-exact
 * statement rather than a statistical one, the three transcendental functions the path needs are
 * DEFINED here as a fixed sequence of IEEE-754 binary64 operations (+ - * / and fma, all correctly
 * rounded on x86-64 and on gfx950) plus integer bit manipulation.  The same text is compiled by gcc
 * (for the checker in oracle/) and by hipcc (for the kernels), and gives the same bits on both.
 *
 * These are our own algorithms (classic argument reduction + near-minimax polynomials whose
 * coefficients are produced by tools/gen_pmath_coeffs.py); nothing here is taken from glibc, ocml
 * or the reference.  Accuracy (checked in tests/test_pmath.py against mpmath): log < 0.8 ulp, exp < 0.9 ulp,
 * sin (cold path only) < 1.6 ulp.
 *
 * Build requirement: -ffp-contract=off on both compilers (every fused operation below is an explicit
 * ed_pm_fma); on the host also -mfma so that the builtin is a single vfmadd (glibc's software fma()
 * is used otherwise: same bits, slower).
 */
#ifndef ED_PMATH_H
#define ED_PMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define ED_PM_FN __host__ __device__ static __inline__ __attribute__((always_inline))
#else
#define ED_PM_FN static inline
#endif

/* ---- coefficients: output of tools/gen_pmath_coeffs.py (do not edit by hand) ---- */
/* log: G(z)=(2atanh(s)-2s)/(s z), z=s^2 in [0,0.029440195]; 8 coeffs; max rel err of G 5.55e-17 (G contributes <1.5% of the result) */
#define ED_PM_LOG_NC 8
#define ED_PM_LOG_COEFFS { 0x1.5555555555555p-1, 0x1.9999999999a38p-2, 0x1.2492492476c87p-2, 0x1.c71c720160b47p-3, 0x1.745cf901605f7p-3, 0x1.3b1c360763b8ap-3, 0x1.0fbe83c2cbcf3p-3, 0x1.0c04595972ab8p-3 }
/* exp: Q(r)=(e^r-1-r)/r^2 on |r|<=0.3466429; 11 coeffs; max rel err of Q 4.76e-18 */
#define ED_PM_EXP_NC 11
#define ED_PM_EXP_COEFFS { 0x1.0000000000000p-1, 0x1.5555555555557p-3, 0x1.5555555555556p-5, 0x1.11111111100d8p-7, 0x1.6c16c16c162d2p-10, 0x1.a01a01abe9ce8p-13, 0x1.a01a01a6d9931p-16, 0x1.71de022bd5558p-19, 0x1.27e4db6121beep-22, 0x1.af4df5750ec30p-26, 0x1.1f730a202ec17p-29 }
/* sin: P(w)=(sin t - t)/t^3, w=t^2, t in [0,1.5723671]; 8 coeffs; max rel err of P 6.69e-17 */
#define ED_PM_SIN_NC 8
#define ED_PM_SIN_COEFFS { -0x1.5555555555555p-3, 0x1.1111111111107p-7, -0x1.a01a01a018a6bp-13, 0x1.71de3a5453a9cp-19, -0x1.ae6455a012592p-26, 0x1.612400c8c470dp-33, -0x1.ae5109fea6b7ap-41, 0x1.899eb3575b7f2p-49 }
#define ED_PM_LN2_HI 0x1.62e42fee00000p-1
#define ED_PM_LN2_LO 0x1.a39ef35793c76p-33
#define ED_PM_INV_LN2 0x1.71547652b82fep+0
#define ED_PM_SQRT2 0x1.6a09e667f3bcdp+0
/* log table (tools/gen_pmath_coeffs.py): 128 sub-intervals of [45/64, 90/64), boundaries on a 2^-8 grid below 1 and a
 * 2^-7 grid above; row = { invc = double(1/centre), logc_hi, logc_lo } with logc = -log(invc), logc_hi a multiple of
 * 2^-42 (so that k*LN2_HI + logc_hi is exact), LN2_HI likewise */
#define ED_PM_LOGT_LN2_HI 0x1.62e42fefa3800p-1
#define ED_PM_LOGT_LN2_LO 0x1.ef35793c76730p-45
#define ED_PM_LOGT_N 128
#define ED_PM_LOGT_ROWS { \
  { 0x1.6b1490aa31a3dp+0, -0x1.65d558d4ce000p-2, -0x1.558fd2dc5bdc0p-51 }, \
  { 0x1.691473a88d0c0p+0, -0x1.602d08af09000p-2, -0x1.ec69176df3f65p-46 }, \
  { 0x1.6719f3601671ap+0, -0x1.5a8cadbbee000p-2, 0x1.7be9b0af7ecf8p-48 }, \
  { 0x1.6524f853b4aa3p+0, -0x1.54f431b7be000p-2, -0x1.a7ef4c0910952p-46 }, \
  { 0x1.63356b88ac0dep+0, -0x1.4f637ebbaa000p-2, 0x1.fc168cb3124b9p-44 }, \
  { 0x1.614b36831ae94p+0, -0x1.49da7f3bcc000p-2, -0x1.07f134daf4b9ap-44 }, \
  { 0x1.5f66434292dfcp+0, -0x1.44591e053a000p-2, 0x1.6de5892923d88p-47 }, \
  { 0x1.5d867c3ece2a5p+0, -0x1.3edf463c17000p-2, 0x1.f08e4297f2c3fp-44 }, \
  { 0x1.5babcc647fa91p+0, -0x1.396ce359bc000p-2, 0x1.5a15c5663663dp-47 }, \
  { 0x1.59d61f123ccaap+0, -0x1.3401e12aed000p-2, 0x1.17f03556e291dp-44 }, \
  { 0x1.5805601580560p+0, -0x1.2e9e2bce12000p-2, -0x1.42e0c128d1dc2p-45 }, \
  { 0x1.56397ba7c52e2p+0, -0x1.2941afb187000p-2, 0x1.20fd2b730e28bp-44 }, \
  { 0x1.54725e6bb82fep+0, -0x1.23ec5991ec000p-2, 0x1.6dbf448a2e522p-44 }, \
  { 0x1.52aff56a8054bp+0, -0x1.1e9e16788a000p-2, 0x1.82ba6d3c8b65ep-44 }, \
  { 0x1.50f22e111c4c5p+0, -0x1.1956d3b9bc000p-2, -0x1.7c8873ad1aa14p-45 }, \
  { 0x1.4f38f62dd4c9bp+0, -0x1.14167ef367000p-2, -0x1.e11ef824daaf5p-44 }, \
  { 0x1.4d843bedc2c4cp+0, -0x1.0edd060b78000p-2, -0x1.044b52d8435f5p-47 }, \
  { 0x1.4bd3edda68fe1p+0, -0x1.09aa572e6c000p-2, -0x1.b51f9e1734342p-44 }, \
  { 0x1.4a27fad76014ap+0, -0x1.047e60cde8000p-2, -0x1.dba110d397f3cp-45 }, \
  { 0x1.4880522014880p+0, -0x1.feb2233ea0000p-3, -0x1.f2c18de00938bp-45 }, \
  { 0x1.46dce34596066p+0, -0x1.f474b134e0000p-3, 0x1.bb019f1df7b5ep-44 }, \
  { 0x1.453d9e2c776cap+0, -0x1.ea4449f04a000p-3, -0x1.5e90663732a36p-44 }, \
  { 0x1.43a2730abee4dp+0, -0x1.e020cc6236000p-3, 0x1.52df0adb91424p-45 }, \
  { 0x1.420b5265e5951p+0, -0x1.d60a17f904000p-3, 0x1.5d8a86fc20d39p-44 }, \
  { 0x1.40782d10e6566p+0, -0x1.cc000c9db4000p-3, 0x1.d6e985d57aff9p-46 }, \
  { 0x1.3ee8f42a5af07p+0, -0x1.c2028ab180000p-3, 0x1.92a3ee55c7ac6p-45 }, \
  { 0x1.3d5d991aa75c6p+0, -0x1.b811730b82000p-3, -0x1.e9e283b9cd768p-46 }, \
  { 0x1.3bd60d9232955p+0, -0x1.ae2ca6f672000p-3, -0x1.7af2dae54f550p-44 }, \
  { 0x1.3a524387ac822p+0, -0x1.a454082e6a000p-3, -0x1.60587c81f7171p-44 }, \
  { 0x1.38d22d366088ep+0, -0x1.9a8778deba000p-3, -0x1.4744a3efec390p-44 }, \
  { 0x1.3755bd1c945eep+0, -0x1.90c6db9fcc000p-3, 0x1.929357718d7cap-46 }, \
  { 0x1.35dce5f9f2af8p+0, -0x1.871213750e000p-3, -0x1.3272b42f9af75p-44 }, \
  { 0x1.34679ace01346p+0, -0x1.7d6903caf6000p-3, 0x1.4cd0b17c301d7p-45 }, \
  { 0x1.32f5ced6a1dfap+0, -0x1.73cb9074fe000p-3, 0x1.d66b90d0005a6p-44 }, \
  { 0x1.3187758e9ebb6p+0, -0x1.6a399dabbe000p-3, 0x1.8f944e66a15a6p-44 }, \
  { 0x1.301c82ac40260p+0, -0x1.60b3100b0a000p-3, 0x1.71756c988f814p-44 }, \
  { 0x1.2eb4ea1fed14bp+0, -0x1.5737cc9018000p-3, -0x1.9b97fa6b887f6p-44 }, \
  { 0x1.2d50a012d50a0p+0, -0x1.4dc7b897bc000p-3, -0x1.c71b60ae1ff0fp-47 }, \
  { 0x1.2bef98e5a3711p+0, -0x1.4462b9dc9c000p-3, 0x1.84830a711b062p-44 }, \
  { 0x1.2a91c92f3c105p+0, -0x1.3b08b67580000p-3, 0x1.ab150f29320fbp-44 }, \
  { 0x1.293725bb804a5p+0, -0x1.31b994d3a4000p-3, -0x1.f0b76e3a50810p-44 }, \
  { 0x1.27dfa38a1ce4dp+0, -0x1.28753bc11a000p-3, -0x1.74346359302e6p-44 }, \
  { 0x1.268b37cd60127p+0, -0x1.1f3b925f26000p-3, 0x1.5ddee9b083633p-46 }, \
  { 0x1.2539d7e9177b2p+0, -0x1.160c8024b2000p-3, -0x1.ebfb2a9009e3dp-45 }, \
  { 0x1.23eb79717605bp+0, -0x1.0ce7ecdccc000p-3, -0x1.4588dabff5447p-46 }, \
  { 0x1.22a0122a0122ap+0, -0x1.03cdc0a51e000p-3, -0x1.81a8cf169fc5cp-44 }, \
  { 0x1.21579804855e6p+0, -0x1.f57bc7d900000p-4, -0x1.76a2c9ea8b04ep-46 }, \
  { 0x1.2012012012012p+0, -0x1.e3707ee304000p-4, -0x1.0f664e6766abdp-45 }, \
  { 0x1.1ecf43c7fb84cp+0, -0x1.d179788218000p-4, -0x1.36193b5efbeedp-44 }, \
  { 0x1.1d8f5672e4abdp+0, -0x1.bf968769fc000p-4, -0x1.42f7c8d824283p-45 }, \
  { 0x1.1c522fc1ce059p+0, -0x1.adc77ee5b0000p-4, 0x1.5718a09c31904p-44 }, \
  { 0x1.1b17c67f2bae3p+0, -0x1.9c0c32d4d4000p-4, 0x1.ab3589e838668p-44 }, \
  { 0x1.19e0119e0119ep+0, -0x1.8a6477a91c000p-4, -0x1.c28b0af9bd6dfp-44 }, \
  { 0x1.18ab083902bdbp+0, -0x1.78d02263d8000p-4, -0x1.6bb9794b69fb7p-47 }, \
  { 0x1.1778a191bd684p+0, -0x1.674f089364000p-4, -0x1.a78394c9d3302p-44 }, \
  { 0x1.1648d50fc3201p+0, -0x1.55e10050e0000p-4, -0x1.c13340c53c72ep-47 }, \
  { 0x1.151b9a3fdd5c9p+0, -0x1.4485e03dbc000p-4, -0x1.fb04ee8d26ab7p-44 }, \
  { 0x1.13f0e8d344724p+0, -0x1.333d7f8184000p-4, 0x1.6c6b6a81b8848p-49 }, \
  { 0x1.12c8b89edc0acp+0, -0x1.2207b5c784000p-4, -0x1.4a16cfc10c7bfp-44 }, \
  { 0x1.11a3019a74826p+0, -0x1.10e45b3cb0000p-4, 0x1.7d699284a3465p-44 }, \
  { 0x1.107fbbe011080p+0, -0x1.ffa6911ab8000p-5, -0x1.3088c98381a8fp-45 }, \
  { 0x1.0f5edfab325a2p+0, -0x1.dda8adc680000p-5, 0x1.1a74c64d9e42fp-45 }, \
  { 0x1.0e40655826011p+0, -0x1.bbcebfc690000p-5, 0x1.7b8e68c317c2ap-46 }, \
  { 0x1.0d24456359e3ap+0, -0x1.9a187b5740000p-5, 0x1.0bf7e4ec4d90dp-44 }, \
  { 0x1.0c0a7868b4171p+0, -0x1.788595a358000p-5, 0x1.06fed083b3a4cp-46 }, \
  { 0x1.0af2f722eecb5p+0, -0x1.5715c4c040000p-5, 0x1.88f55dfc47628p-44 }, \
  { 0x1.09ddba6af8360p+0, -0x1.35c8bfaa10000p-5, -0x1.8347d5ef9eb35p-44 }, \
  { 0x1.08cabb37565e2p+0, -0x1.149e3e4008000p-5, 0x1.2b99a9a4168fdp-44 }, \
  { 0x1.07b9f29b8eae2p+0, -0x1.e72bf28140000p-6, 0x1.8cb3149774d47p-45 }, \
  { 0x1.06ab59c7912fbp+0, -0x1.a55f548c60000p-6, 0x1.dec609f2d03c9p-45 }, \
  { 0x1.059eea0727586p+0, -0x1.63d6178690000p-6, -0x1.77b7389596542p-47 }, \
  { 0x1.04949cc1664c5p+0, -0x1.228fb1fea0000p-6, -0x1.70513284991fep-45 }, \
  { 0x1.038c6b78247fcp+0, -0x1.c317384c80000p-7, 0x1.41e53fcefb9fep-44 }, \
  { 0x1.02864fc7729e9p+0, -0x1.41929f9680000p-7, -0x1.9862755d01368p-46 }, \
  { 0x1.0182436517a37p+0, -0x1.8121214580000p-8, -0x1.ac06382973f27p-46 }, \
  { 0x1.0080402010080p+0, -0x1.0040155d80000p-9, 0x1.3bf10c7cc7089p-44 }, \
  { 0x1.fe01fe01fe020p-1, 0x1.ff00aa2b00000p-9, 0x1.0ba04a086b56ap-45 }, \
  { 0x1.fa11caa01fa12p-1, 0x1.7dc475f820000p-7, -0x1.eb2d45b5da1f5p-44 }, \
  { 0x1.f6310aca0dbb5p-1, 0x1.3cea443470000p-6, -0x1.69f0c32d6a40bp-44 }, \
  { 0x1.f25f644230ab5p-1, 0x1.b9fc027b00000p-6, -0x1.b99990ae6922ap-44 }, \
  { 0x1.ee9c7f8458e02p-1, 0x1.1b0d989240000p-5, -0x1.340ae9ae889bbp-44 }, \
  { 0x1.eae807aba01ebp-1, 0x1.58a5bafc90000p-5, -0x1.b2d039570ad39p-45 }, \
  { 0x1.e741aa59750e4p-1, 0x1.95c830ec90000p-5, -0x1.c0dc297c5feb8p-45 }, \
  { 0x1.e3a9179dc1a73p-1, 0x1.d276b8adb0000p-5, 0x1.6ac83c78a64b0p-46 }, \
  { 0x1.e01e01e01e01ep-1, 0x1.0759835990000p-4, -0x1.b8ebfe4b59987p-44 }, \
  { 0x1.dca01dca01dcap-1, 0x1.253f62f0a0000p-4, 0x1.41708fb69a701p-44 }, \
  { 0x1.d92f2231e7f8ap-1, 0x1.42edcbea64000p-4, 0x1.bb6aeea7c9acdp-46 }, \
  { 0x1.d5cac807572b2p-1, 0x1.60658a9374000p-4, 0x1.0c3c1dee9c4f8p-44 }, \
  { 0x1.d272ca3fc5b1ap-1, 0x1.7da766d7b0000p-4, 0x1.2d0344480c89bp-44 }, \
  { 0x1.cf26e5c44bfc6p-1, 0x1.9ab4246204000p-4, -0x1.8a46826787061p-45 }, \
  { 0x1.cbe6d9601cbe7p-1, 0x1.b78c82bb10000p-4, -0x1.2604fbc3987e7p-44 }, \
  { 0x1.c8b265afb8a42p-1, 0x1.d4313d66cc000p-4, -0x1.9452379135713p-45 }, \
  { 0x1.c5894d10d4986p-1, 0x1.f0a30c0118000p-4, -0x1.d5bce83368e91p-44 }, \
  { 0x1.c26b5392ea01cp-1, 0x1.0671512ca6000p-3, -0x1.a44979cdc0a3dp-45 }, \
  { 0x1.bf583ee868d8bp-1, 0x1.1478584674000p-3, 0x1.560651027c750p-46 }, \
  { 0x1.bc4fd65883e7bp-1, 0x1.2266f190a6000p-3, -0x1.4cddab840e7f6p-45 }, \
  { 0x1.b951e2b18ff23p-1, 0x1.303d718e48000p-3, -0x1.5b6b5ce3ecb05p-50 }, \
  { 0x1.b65e2e3beee05p-1, 0x1.3dfc2b0ecc000p-3, 0x1.8a9ba62b8c13fp-45 }, \
  { 0x1.b37484ad806cep-1, 0x1.4ba36f39a6000p-3, -0x1.436fbb3f219e5p-44 }, \
  { 0x1.b094b31d922a4p-1, 0x1.59338d9982000p-3, 0x1.0ac68b7555d4ap-48 }, \
  { 0x1.adbe87f94905ep-1, 0x1.66acd4272a000p-3, 0x1.aa1cdbfc6c785p-44 }, \
  { 0x1.aaf1d2f87ebfdp-1, 0x1.740f8f5404000p-3, -0x1.0b9a499018aa1p-44 }, \
  { 0x1.a82e65130e159p-1, 0x1.815c0a1436000p-3, -0x1.02dbaf9201ce8p-44 }, \
  { 0x1.a574107688a4ap-1, 0x1.8e928de886000p-3, 0x1.a8224b13d72d5p-44 }, \
  { 0x1.a2c2a87c51ca0p-1, 0x1.9bb362e7e0000p-3, -0x1.1eca8a1ce0ffcp-45 }, \
  { 0x1.a01a01a01a01ap-1, 0x1.a8becfc882000p-3, 0x1.e3195cf21b9cfp-44 }, \
  { 0x1.9d79f176b682dp-1, 0x1.b5b519e8fc000p-3, -0x1.4b4eaec011f31p-44 }, \
  { 0x1.9ae24ea5510dap-1, 0x1.c2968558c2000p-3, -0x1.cf7d3dee38a40p-45 }, \
  { 0x1.9852f0d8ec0ffp-1, 0x1.cf6354e09c000p-3, 0x1.775339a07d55bp-45 }, \
  { 0x1.95cbb0be377aep-1, 0x1.dc1bca0abe000p-3, 0x1.8f671a628ccc6p-44 }, \
  { 0x1.934c67f9b2ce6p-1, 0x1.e8c0252aa6000p-3, -0x1.6803b80e8e6ffp-45 }, \
  { 0x1.90d4f120190d5p-1, 0x1.f550a564b8000p-3, -0x1.32513a09202fep-45 }, \
  { 0x1.8e6527af1373fp-1, 0x1.00e6c45ad5000p-2, 0x1.cd88d52e01203p-50 }, \
  { 0x1.8bfce8062ff3ap-1, 0x1.071b85fcd6000p-2, -0x1.bcb7ba3e01a11p-44 }, \
  { 0x1.899c0f601899cp-1, 0x1.0d46b579ab000p-2, 0x1.d2d21f640e1e6p-44 }, \
  { 0x1.87427bcc092b9p-1, 0x1.136870293b000p-2, -0x1.d3f3c99d67123p-44 }, \
  { 0x1.84f00c2780614p-1, 0x1.1980d2dd42000p-2, 0x1.b75fa7a361c9ap-45 }, \
  { 0x1.82a4a0182a4a0p-1, 0x1.1f8ff9e48a000p-2, 0x1.7966c040cbe77p-45 }, \
  { 0x1.8060180601806p-1, 0x1.2596010df7000p-2, 0x1.8e7cc224ea3e3p-44 }, \
  { 0x1.7e225515a4f1dp-1, 0x1.2b9303ab8a000p-2, -0x1.6d8c2d6bfb0a5p-45 }, \
  { 0x1.7beb3922e017cp-1, 0x1.31871c9544000p-2, 0x1.84c2b94cecfd9p-46 }, \
  { 0x1.79baa6bb6398bp-1, 0x1.3772662bfe000p-2, -0x1.e8f7eac53b023p-44 }, \
  { 0x1.77908119ac60dp-1, 0x1.3d54fa5c1f000p-2, 0x1.c4054d9a395e3p-44 }, \
  { 0x1.756cac201756dp-1, 0x1.432ef2a04f000p-2, -0x1.fb4c1931715adp-44 }, \
  { 0x1.734f0c541fe8dp-1, 0x1.4900680401000p-2, -0x1.8c037fe1a0f8cp-44 }, \
  { 0x1.713786d9c7c09p-1, 0x1.4ec9732600000p-2, 0x1.345caaf04d104p-45 }, \
  { 0x1.6f26016f26017p-1, 0x1.548a2c3add000p-2, 0x1.3154e63081cf7p-45 }, \
  { 0x1.6d1a62681c861p-1, 0x1.5a42ab0f4d000p-2, -0x1.e71af2df7ba69p-50 } \
}
/* ---- end generated ---- */

ED_PM_FN uint64_t ed_pm_bits(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
ED_PM_FN double ed_pm_from_bits(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
ED_PM_FN double ed_pm_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
/* fma(a, b, k) with a compile-time constant addend k (Horner steps).  Same operation, same bits on host and device.
 * On the device the constant is handed to v_fma_f64 in a scalar register pair: left to itself the compiler picks
 * v_fmac_f64, whose addend must already sit in the destination VGPRs, and spends two extra vector moves per coefficient.
 *
 * The pair is written by two s_mov_b32 INSIDE the asm statement, from immediates.  Round 1 passed the constant as an
 * "s" operand instead; short of SGPRs, the register allocator then parked such constants in VGPR lanes and reloaded them
 * with v_readlane_b32 -- a VALU write of an SGPR -- directly in front of the asm, and gfx940/gfx950 need 2 wait states
 * between a VALU write of an SGPR and a VALU read of it.  LLVM inserts those for its own instructions but does not look
 * inside inline asm, so the fma could pick up the previous coefficient's high half: the one unreproduced log-likelihood
 * mismatch of round 1 (tools/hazard_sgpr_repro.hip shows the hazard in isolation; tools/isa_hazard_scan.py and
 * tests/test_isa_hazards.py keep the compiler's output free of it).  Here nothing but SALU ever writes s[28:29]
 * (caller-saved in the AMDGPU calling convention, so out-of-line callees need not preserve them), and an SALU write
 * followed by a VALU read is interlocked by the hardware. */
#if defined(__HIP_DEVICE_COMPILE__) && defined(ED_PM_FMA_K_SGPR_OPERAND)
/* DIAGNOSTIC VARIANT ONLY (libedcore_sgprasm.so, tools/soak_emission.py --variant sgprasm): round 1's form, kept so that the
 * hazard can be shown on hardware against the fixed library.  Never part of the product build. */
__device__ static __inline__ __attribute__((always_inline)) double ed_pm_fma_k(double a, double b, double k)
{
  double d;
  __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(k));
  return d;
}
#elif defined(__HIP_DEVICE_COMPILE__)
__device__ static __inline__ __attribute__((always_inline)) double ed_pm_fma_k(double a, double b, double k)
{
  double d;
  const unsigned long long u = __builtin_bit_cast(unsigned long long, k);   /* folds to a constant once inlined */
  __asm__("s_mov_b32 s28, %3\n\ts_mov_b32 s29, %4\n\tv_fma_f64 %0, %1, %2, s[28:29]"
          : "=v"(d)
          : "v"(a), "v"(b), "i"((unsigned)(u & 0xffffffffull)), "i"((unsigned)(u >> 32))
          : "s28", "s29");
  return d;
}
#else
ED_PM_FN double ed_pm_fma_k(double a, double b, double k) { return __builtin_fma(a, b, k); }
#endif
ED_PM_FN double ed_pm_inf(void) { return ed_pm_from_bits(0x7ff0000000000000ULL); }
ED_PM_FN double ed_pm_nan(void) { return ed_pm_from_bits(0x7ff8000000000000ULL); }

/* natural logarithm, table-driven.
 *   x = 2^k z, z in [45/64, 90/64); the top 7 mantissa bits past the offset pick sub-interval i with centre c:
 *   log x = k ln2 + log c + log1p(r),  r = z/c - 1 = fma(z, invc, -1), |r| < 2^-8,
 *   log1p(r) = r - r^2/2 + ... + r^7/7 (the next term is below 2^-67).
 * k*LN2_HI + logc_hi is exact (both multiples of 2^-42), r is added with a Fast2Sum (|k ln2 + log c| >= 2^-7.4 > |r|
 * outside the two sub-intervals around 1), everything small is summed first.  For k = 0 and z in [1 - 2^-8, 1 + 2^-7)
 * the result is ~r itself and needs RELATIVE accuracy: r = x - 1 exactly and the series runs to r^9/9.
 * ed_plog_core: x positive, normal, finite. */
ED_PM_FN double ed_plog_core_t(double x, int k0, const double* T)   /* T: the table, row i at T[3 i] */
{
  const uint64_t ix = ed_pm_bits(x);
  if (k0 == 0 && ix - 0x3fefe00000000000ULL < 0x3ff0200000000000ULL - 0x3fefe00000000000ULL) {
    const double r = x - 1.0;                    /* exact */
    double q = 1.0 / 9.0;
    q = ed_pm_fma_k(q, r, -1.0 / 8.0);
    q = ed_pm_fma_k(q, r, 1.0 / 7.0);
    q = ed_pm_fma_k(q, r, -1.0 / 6.0);
    q = ed_pm_fma_k(q, r, 1.0 / 5.0);
    q = ed_pm_fma_k(q, r, -1.0 / 4.0);
    q = ed_pm_fma_k(q, r, 1.0 / 3.0);
    const double r2 = r * r;
    return r + ed_pm_fma(r2 * r, q, -0.5 * r2);
  }
  const uint64_t tmp = ix - 0x3fe6800000000000ULL;                    /* 45/64 */
  const int i = (int)((tmp >> 45) & 127);
  const int k = (int)((int64_t)tmp >> 52) + k0;
  const double z = ed_pm_from_bits(ix - (tmp & 0xfff0000000000000ULL));
  const double r = ed_pm_fma(z, T[3 * i], -1.0);
  const double kd = (double)k;
  const double w = ed_pm_fma(kd, ED_PM_LOGT_LN2_HI, T[3 * i + 1]);    /* exact */
  const double hi = w + r;
  const double lo = (w - hi) + r;
  double q = 1.0 / 7.0;
  q = ed_pm_fma_k(q, r, -1.0 / 6.0);
  q = ed_pm_fma_k(q, r, 1.0 / 5.0);
  q = ed_pm_fma_k(q, r, -1.0 / 4.0);
  q = ed_pm_fma_k(q, r, 1.0 / 3.0);
  q = ed_pm_fma_k(q, r, -0.5);
  const double t = ed_pm_fma(kd, ED_PM_LOGT_LN2_LO, T[3 * i + 2]);
  return hi + ed_pm_fma(r * r, q, lo + t);
}

ED_PM_FN double ed_plog_core(double x, int k0)
{
  const double T[ED_PM_LOGT_N][3] = ED_PM_LOGT_ROWS;
  return ed_plog_core_t(x, k0, &T[0][0]);
}

ED_PM_FN double ed_plog(double x)
{
  const uint64_t u = ed_pm_bits(x);
  if (!(x > 0.0)) {               /* zero, negative, NaN */
    if (x == 0.0) return -ed_pm_inf();
    return ed_pm_nan();
  }
  if (u >= 0x7ff0000000000000ULL) return x;   /* +inf */
  if (u < 0x0010000000000000ULL) return ed_plog_core(x * 0x1p54, -54);   /* subnormal: renormalise */
  return ed_plog_core(x, 0);
}

/* exponential.  x = k ln2 + r, |r| <= ln2/2; e^r = 1 + (r + r^2 Q(r)); result scaled by 2^k in two
 * exact-power-of-two multiplications so that subnormal results round once. */
ED_PM_FN double ed_pexp(double x)
{
  const double c[ED_PM_EXP_NC] = ED_PM_EXP_COEFFS;
  if (x != x) return x;
  if (x > -0x1p-5 && x < 0x1p-5) {
    /* short series (the one case the hot path has: exp of a Stirling tail, |x| < 0.0084): e^x = 1 + (x + x^2 Q),
     * Q = 1/2 + x/6 + ... + x^5/5040; the first dropped term, x^8/40320, is below 2.3e-17 */
    double q = 1.0 / 5040.0;
    q = ed_pm_fma_k(q, x, 1.0 / 720.0);
    q = ed_pm_fma_k(q, x, 1.0 / 120.0);
    q = ed_pm_fma_k(q, x, 1.0 / 24.0);
    q = ed_pm_fma_k(q, x, 1.0 / 6.0);
    q = ed_pm_fma_k(q, x, 0.5);
    return 1.0 + ed_pm_fma(x * x, q, x);
  }
  if (x > 0x1.62e42fefa39efp+9) return ed_pm_inf();     /* > log(DBL_MAX) */
  if (x < -0x1.74910d52d3051p+9) return 0.0;            /* < log(2^-1075) */
  const double t = x * ED_PM_INV_LN2;
  const double kd = (t + 0x1.8p52) - 0x1.8p52;          /* round to nearest integer, ties to even */
  double r = ed_pm_fma(-kd, ED_PM_LN2_HI, x);           /* exact */
  r = ed_pm_fma(-kd, ED_PM_LN2_LO, r);
  double q = c[ED_PM_EXP_NC - 1];
  for (int i = ED_PM_EXP_NC - 2; i >= 0; --i) q = ed_pm_fma_k(q, r, c[i]);
  const double p = ed_pm_fma(r * r, q, r);
  double e = 1.0 + p;
  const int k = (int)kd;
  const int k1 = k >> 1;            /* arithmetic shift: floor(k/2) */
  const int k2 = k - k1;
  e = e * ed_pm_from_bits((uint64_t)(1023 + k1) << 52);
  e = e * ed_pm_from_bits((uint64_t)(1023 + k2) << 52);
  return e;
}

/* sine on [0, pi] only -- the one place the path needs it is the reflection branch of log-gamma
 * for 0.02 <= x < 0.5 (reference src/VP_gamma.c:1180-1183 evaluates sin(pi*(1-x)), :1247-1249
 * sin(pi*x)).  For t > pi/2 the argument is folded with a two-part pi: u = (PI_HI - t) + PI_LO, the
 * first subtraction being exact.  Outside [0, pi] the result is NaN. */
#define ED_PM_PI_HI 0x1.921fb54442d18p+1
#define ED_PM_PI_LO 0x1.1a62633145c07p-53
ED_PM_FN double ed_psin_0pi(double t)
{
  const double c[ED_PM_SIN_NC] = ED_PM_SIN_COEFFS;
  if (!(t >= 0.0 && t <= ED_PM_PI_HI)) return ed_pm_nan();
  if (t > 0.5 * ED_PM_PI_HI) t = (ED_PM_PI_HI - t) + ED_PM_PI_LO;
  const double w = t * t;
  double p = c[ED_PM_SIN_NC - 1];
  for (int i = ED_PM_SIN_NC - 2; i >= 0; --i) p = ed_pm_fma_k(p, w, c[i]);
  return ed_pm_fma(t * w, p, t);
}

/* sine of any finite t with |t| < 2^52: t = k pi + r through a three-part pi (each step one fma; the first two are exact
 * for the k that occur), then the polynomial above on |r| <= pi/2.  Absolute error ~2e-16.  Used by the cold path of
 * log|Gamma(x)| for negative x (sign of sin(pi x), its magnitude where it is at least 0.015 pi). */
#define ED_PM_INV_PI 0x1.45f306dc9c883p-2
#define ED_PM_PI_LO2 -0x1.f1976b7ed8fbcp-109
ED_PM_FN double ed_psin_any(double t)
{
  if (!(t > -0x1p52 && t < 0x1p52)) return ed_pm_nan();
  const double k = (t * ED_PM_INV_PI + 0x1.8p52) - 0x1.8p52;      /* nearest integer */
  double r = ed_pm_fma(-k, ED_PM_PI_HI, t);
  r = ed_pm_fma(-k, ED_PM_PI_LO, r);
  r = ed_pm_fma(-k, ED_PM_PI_LO2, r);
  const double a = ed_psin_0pi(r < 0.0 ? -r : r);
  const int odd = (int)(((int64_t)k) & 1);
  return ((r < 0.0) != (odd != 0)) ? -a : a;
}

/* b^-n for a small positive integer n (3..7 on the path) and b >= 3: repeated multiplication, one division */
ED_PM_FN double ed_ppown(double b, int n)
{
  double p = b;
  for (int i = 1; i < n; ++i) p *= b;
  return 1.0 / p;
}

/* B_2j / (2j)!, j = 0..14 (Euler-Maclaurin coefficients of the Hurwitz zeta function; reference src/VP_zeta.c:563-579) */
#define ED_HZETA_C { 1.00000000000000000000000000000, 0.083333333333333333333333333333, -0.00138888888888888888888888888889, \
  0.000033068783068783068783068783069, -8.2671957671957671957671957672e-07, 2.0876756987868098979210090321e-08, \
  -5.2841901386874931848476822022e-10, 1.3382536530684678832826980975e-11, -3.3896802963225828668301953912e-13, \
  8.5860620562778445641359054504e-15, -2.1748686985580618730415164239e-16, 5.5090028283602295152026526089e-18, \
  -1.3954464685812523340707686264e-19, 3.5347070396294674716932299778e-21, -8.9535174270375468504026113181e-23 }

#endif /* ED_PMATH_H */
