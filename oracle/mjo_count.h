/* mjo_count.h -- operation-counting build of the oracle (TEST / BENCH INFRASTRUCTURE, see mjo.h).
 *
 * SURVEY.md 8d defines the algorithmic flop count of an env-step as "the CPU oracle's instrumented count (fadd / fmul = 1,
 * fma = 2, div / sqrt = 1)".  This header is force-included (g++ -x c++ -include mjo_count.h, oracle/Makefile target
 * libmjo_count.so) ahead of the unchanged oracle sources: every system header they need is pulled in first, then `double`
 * becomes a wrapper whose arithmetic operators bump a thread-local counter.  Counted as 1: + - * / sqrt and every libm call
 * (pow, exp, log, sin, cos, acos, atan2, fmod); not counted: comparisons, negation, fabs, fmin / fmax, copies, integer work.
 * A product feeding a sum counts 2, like an fma.  The layout of the wrapper is that of a double, so the ctypes interface
 * (model description, field pointers) is unchanged. */
#pragma once
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <type_traits>

extern thread_local unsigned long long mjo_nflop;

struct cdbl {
	double v;
	cdbl() = default;
	template <typename T, typename = typename std::enable_if<std::is_arithmetic<T>::value>::type> cdbl(T x) : v((double)x) {}
	explicit operator int() const { return (int)v; }
	explicit operator unsigned int() const { return (unsigned int)v; }
	explicit operator long() const { return (long)v; }
	explicit operator bool() const { return v != 0; }
	explicit operator float() const { return (float)v; }
	cdbl &operator+=(cdbl o) { mjo_nflop++; v += o.v; return *this; }
	cdbl &operator-=(cdbl o) { mjo_nflop++; v -= o.v; return *this; }
	cdbl &operator*=(cdbl o) { mjo_nflop++; v *= o.v; return *this; }
	cdbl &operator/=(cdbl o) { mjo_nflop++; v /= o.v; return *this; }
};
static_assert(sizeof(cdbl) == sizeof(double), "layout");
#define MJO_BIN(op) \
	inline cdbl operator op(cdbl a, cdbl b) { mjo_nflop++; cdbl r; r.v = a.v op b.v; return r; }
MJO_BIN(+) MJO_BIN(-) MJO_BIN(*) MJO_BIN(/)
#undef MJO_BIN
inline cdbl operator-(cdbl a) { cdbl r; r.v = -a.v; return r; }
inline cdbl operator+(cdbl a) { return a; }
#define MJO_CMP(op) inline bool operator op(cdbl a, cdbl b) { return a.v op b.v; }
MJO_CMP(<) MJO_CMP(>) MJO_CMP(<=) MJO_CMP(>=) MJO_CMP(==) MJO_CMP(!=)
#undef MJO_CMP
#define MJO_F1(fn) inline cdbl fn(cdbl a) { mjo_nflop++; cdbl r; r.v = ::fn(a.v); return r; }
MJO_F1(sqrt) MJO_F1(exp) MJO_F1(log) MJO_F1(sin) MJO_F1(cos) MJO_F1(acos) MJO_F1(asin) MJO_F1(tan)
#undef MJO_F1
inline cdbl fabs(cdbl a) { cdbl r; r.v = ::fabs(a.v); return r; }
inline cdbl floor(cdbl a) { cdbl r; r.v = ::floor(a.v); return r; }
inline cdbl fmax(cdbl a, cdbl b) { cdbl r; r.v = ::fmax(a.v, b.v); return r; }
inline cdbl fmin(cdbl a, cdbl b) { cdbl r; r.v = ::fmin(a.v, b.v); return r; }
inline cdbl pow(cdbl a, cdbl b) { mjo_nflop++; cdbl r; r.v = ::pow(a.v, b.v); return r; }
inline cdbl atan2(cdbl a, cdbl b) { mjo_nflop++; cdbl r; r.v = ::atan2(a.v, b.v); return r; }
inline cdbl fmod(cdbl a, cdbl b) { mjo_nflop++; cdbl r; r.v = ::fmod(a.v, b.v); return r; }

#define double cdbl
