// Postfix-program evaluator shared by the host (fh_expr.cpp) and the assembly kernels (source term at the Gauss points).
// Program word = op | (argument << 8).  See fh_expr.cpp for the grammar and the reference interface it replaces.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>

enum { FHX_CONST = 0, FHX_VAR, FHX_NEG, FHX_NOT, FHX_ADD, FHX_SUB, FHX_MUL, FHX_DIV, FHX_MOD, FHX_POW, FHX_EQ, FHX_NE, FHX_LT, FHX_LE, FHX_GT,
       FHX_GE, FHX_AND, FHX_OR, FHX_FUNC };
enum { FHF_ABS = 0, FHF_ACOS, FHF_ACOSH, FHF_ASIN, FHF_ASINH, FHF_ATAN, FHF_ATANH, FHF_CBRT, FHF_CEIL, FHF_COS, FHF_COSH, FHF_COT, FHF_CSC,
       FHF_EXP, FHF_EXP2, FHF_FLOOR, FHF_INT, FHF_LOG, FHF_LOG10, FHF_LOG2, FHF_SEC, FHF_SIN, FHF_SINH, FHF_SQRT, FHF_TAN, FHF_TANH,
       FHF_TRUNC, FHF_ATAN2, FHF_HYPOT, FHF_MAX, FHF_MIN, FHF_POW2, FHF_IF };
constexpr int FHX_STACK = 16;
constexpr double FHX_EPS = 1e-12;   // the parser library's default epsilon for = and != on doubles

__host__ __device__ inline double fhx_truth(double v) { return fabs(v) >= 0.5 ? 1.0 : 0.0; }

__host__ __device__ inline double fh_expr_device_eval(const int* code, int ncode, const double* consts, const double* x) {
  double st[FHX_STACK];
  int sp = 0;
  for (int k = 0; k < ncode; k++) {
    const int op = code[k] & 255, arg = code[k] >> 8;
    switch (op) {
      case FHX_CONST: st[sp++] = consts[arg]; break;
      case FHX_VAR: st[sp++] = x[arg]; break;
      case FHX_NEG: st[sp - 1] = -st[sp - 1]; break;
      case FHX_NOT: st[sp - 1] = 1.0 - fhx_truth(st[sp - 1]); break;
      case FHX_FUNC: {
        if (arg == FHF_IF) {
          sp -= 2;
          st[sp - 1] = fhx_truth(st[sp - 1]) != 0.0 ? st[sp] : st[sp + 1];
          break;
        }
        if (arg >= FHF_ATAN2) {
          sp--;
          const double a = st[sp - 1], b = st[sp];
          st[sp - 1] = arg == FHF_ATAN2 ? atan2(a, b) : arg == FHF_HYPOT ? hypot(a, b) : arg == FHF_MAX ? (a > b ? a : b)
                       : arg == FHF_MIN ? (a < b ? a : b) : pow(a, b);
          break;
        }
        const double a = st[sp - 1];
        double r = 0.0;
        switch (arg) {
          case FHF_ABS: r = fabs(a); break;
          case FHF_ACOS: r = acos(a); break;
          case FHF_ACOSH: r = acosh(a); break;
          case FHF_ASIN: r = asin(a); break;
          case FHF_ASINH: r = asinh(a); break;
          case FHF_ATAN: r = atan(a); break;
          case FHF_ATANH: r = atanh(a); break;
          case FHF_CBRT: r = cbrt(a); break;
          case FHF_CEIL: r = ceil(a); break;
          case FHF_COS: r = cos(a); break;
          case FHF_COSH: r = cosh(a); break;
          case FHF_COT: r = 1.0 / tan(a); break;
          case FHF_CSC: r = 1.0 / sin(a); break;
          case FHF_EXP: r = exp(a); break;
          case FHF_EXP2: r = exp2(a); break;
          case FHF_FLOOR: r = floor(a); break;
          case FHF_INT: r = floor(a + 0.5); break;
          case FHF_LOG: r = log(a); break;
          case FHF_LOG10: r = log10(a); break;
          case FHF_LOG2: r = log2(a); break;
          case FHF_SEC: r = 1.0 / cos(a); break;
          case FHF_SIN: r = sin(a); break;
          case FHF_SINH: r = sinh(a); break;
          case FHF_SQRT: r = sqrt(a); break;
          case FHF_TAN: r = tan(a); break;
          case FHF_TANH: r = tanh(a); break;
          case FHF_TRUNC: r = trunc(a); break;
        }
        st[sp - 1] = r;
        break;
      }
      default: {
        sp--;
        const double a = st[sp - 1], b = st[sp];
        double r = 0.0;
        switch (op) {
          case FHX_ADD: r = a + b; break;
          case FHX_SUB: r = a - b; break;
          case FHX_MUL: r = a * b; break;
          case FHX_DIV: r = a / b; break;
          case FHX_MOD: r = fmod(a, b); break;
          case FHX_POW: r = pow(a, b); break;
          case FHX_EQ: r = fabs(a - b) <= FHX_EPS ? 1.0 : 0.0; break;
          case FHX_NE: r = fabs(a - b) > FHX_EPS ? 1.0 : 0.0; break;
          case FHX_LT: r = a < b ? 1.0 : 0.0; break;
          case FHX_LE: r = a <= b ? 1.0 : 0.0; break;
          case FHX_GT: r = a > b ? 1.0 : 0.0; break;
          case FHX_GE: r = a >= b ? 1.0 : 0.0; break;
          case FHX_AND: r = fhx_truth(a) * fhx_truth(b); break;
          case FHX_OR: r = (fhx_truth(a) + fhx_truth(b)) > 0.0 ? 1.0 : 0.0; break;
        }
        st[sp - 1] = r;
      }
    }
  }
  return st[0];
}
