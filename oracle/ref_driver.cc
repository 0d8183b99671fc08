/*
 * oracle/ref_driver.cc -- TEST INFRASTRUCTURE.
 * Thin C wrapper around the two reference headers that compile on their own
 * (psk.hh, polar_tables.hh), included from where they lie under
 * /root/reference (-I$(REF)); built into oracle/_ref/ only.  Used to validate
 * the restatement (oracle/dsp.c, oracle/polar.c) and to generate the golden
 * vectors under tests/golden/ (tests/golden/gen_golden.py).
 */
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <type_traits>
#include "psk.hh"
#include "polar_tables.hh"

typedef std::complex<float> cmplx;
extern "C" {
void ref_psk8_hard(float *b, float re, float im) { PhaseShiftKeying<8, cmplx, float>::hard(b, cmplx(re, im)); }
void ref_psk8_soft(float *b, float re, float im, float p) { PhaseShiftKeying<8, cmplx, float>::soft(b, cmplx(re, im), p); }
void ref_psk8_map(float *out, float *b) { cmplx c = PhaseShiftKeying<8, cmplx, float>::map(b); out[0] = c.real(); out[1] = c.imag(); }
void ref_psk4_hard(float *b, float re, float im) { PhaseShiftKeying<4, cmplx, float>::hard(b, cmplx(re, im)); }
void ref_psk4_soft(float *b, float re, float im, float p) { PhaseShiftKeying<4, cmplx, float>::soft(b, cmplx(re, im), p); }
void ref_psk4_map(float *out, float *b) { cmplx c = PhaseShiftKeying<4, cmplx, float>::map(b); out[0] = c.real(); out[1] = c.imag(); }
const uint32_t *ref_frozen(int table) { return table ? frozen_64512_43072 : frozen_64800_43072; }
}
