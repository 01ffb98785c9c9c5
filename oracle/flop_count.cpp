/*
 * oracle/flop_count.cpp  --  TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * "report algorithmic FLOPs per particle-step as COUNTED by an instrumented cpu_ref build".
 * This file is that build: it compiles oracle/reacher_ref.c unchanged as C++ with `double` replaced by a
 * counting scalar, so every floating-point add / multiply / divide / sqrt / sin / cos / compare the oracle
 * executes is tallied.  Built by `make -C oracle libreacher_flops.so`; used by bench.py (roofline.valu) and
 * tests/test_flop_count_cpu.py through oracle/physics_ref.py::count_flops.  Single-threaded (no OpenMP).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct FlopTally { long add, mul, div, sqrt_, trig, cmp; };
static FlopTally g_tally;

struct cd {
    double v;
    cd() = default;
    cd(double x) : v(x) {}
    cd(int x) : v(x) {}
    cd(long x) : v((double)x) {}
    explicit operator int() const { return (int)v; }
    explicit operator long() const { return (long)v; }
    explicit operator bool() const { return v != 0.0; }
};
static inline cd operator+(cd a, cd b) { g_tally.add++; return cd(a.v + b.v); }
static inline cd operator-(cd a, cd b) { g_tally.add++; return cd(a.v - b.v); }
static inline cd operator*(cd a, cd b) { g_tally.mul++; return cd(a.v * b.v); }
static inline cd operator/(cd a, cd b) { g_tally.div++; return cd(a.v / b.v); }
static inline cd operator-(cd a) { return cd(-a.v); }
static inline cd operator+(cd a) { return a; }
static inline cd& operator+=(cd& a, cd b) { g_tally.add++; a.v += b.v; return a; }
static inline cd& operator-=(cd& a, cd b) { g_tally.add++; a.v -= b.v; return a; }
static inline cd& operator*=(cd& a, cd b) { g_tally.mul++; a.v *= b.v; return a; }
static inline cd& operator/=(cd& a, cd b) { g_tally.div++; a.v /= b.v; return a; }
#define CD_CMP(op) static inline bool operator op(cd a, cd b) { g_tally.cmp++; return a.v op b.v; }
CD_CMP(<) CD_CMP(>) CD_CMP(<=) CD_CMP(>=) CD_CMP(==) CD_CMP(!=)
#define CD_MIX(op, R)                                                    \
    static inline R operator op(cd a, double b) { return a op cd(b); }   \
    static inline R operator op(double a, cd b) { return cd(a) op b; }   \
    static inline R operator op(cd a, int b) { return a op cd(b); }      \
    static inline R operator op(int a, cd b) { return cd(a) op b; }
CD_MIX(+, cd) CD_MIX(-, cd) CD_MIX(*, cd) CD_MIX(/, cd)
CD_MIX(<, bool) CD_MIX(>, bool) CD_MIX(<=, bool) CD_MIX(>=, bool) CD_MIX(==, bool) CD_MIX(!=, bool)
static inline cd sqrt(cd a) { g_tally.sqrt_++; return cd(::sqrt(a.v)); }
static inline cd fabs(cd a) { return cd(::fabs(a.v)); }
static inline cd sin(cd a) { g_tally.trig++; return cd(::sin(a.v)); }
static inline cd cos(cd a) { g_tally.trig++; return cd(::cos(a.v)); }
static inline cd pow(cd a, cd b) { g_tally.trig++; return cd(::pow(a.v, b.v)); }
static inline cd atan2(cd a, cd b) { g_tally.trig++; return cd(::atan2(a.v, b.v)); }
static inline cd fmin(cd a, cd b) { g_tally.cmp++; return cd(::fmin(a.v, b.v)); }
static inline cd fmax(cd a, cd b) { g_tally.cmp++; return cd(::fmax(a.v, b.v)); }
static inline bool isfinite_cd(cd a) { return ::isfinite(a.v); }

#undef _OPENMP
#define double cd
#define OR_NO_POLISH 1
#define OR_DENSE_JACOBIANS 1
extern "C" {
#include "reacher_ref.c"
}
#undef double

extern "C" void or_flops_reset(void) { memset(&g_tally, 0, sizeof g_tally); }
/* out[6] = adds (incl. subtractions), multiplies, divides, square roots, sin/cos/pow calls, compares (incl. min/max) */
extern "C" void or_flops_get(long* out) {
    out[0] = g_tally.add; out[1] = g_tally.mul; out[2] = g_tally.div;
    out[3] = g_tally.sqrt_; out[4] = g_tally.trig; out[5] = g_tally.cmp;
}
