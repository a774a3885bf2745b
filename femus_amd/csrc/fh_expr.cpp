// Run-time expressions: the evaluator behind femus::ParsedFunction
//   src/02_calculus/function_parser/ParsedFunction.hpp:25-60, ParsedFunction.cpp:28-80
// which wraps the third-party "Function Parser for C++" (fparser.hh; located by cmake-modules/FindFParser.cmake, no version
// pinned, not vendored under /root/reference).  Its published grammar is restated here: a recursive-descent parser that
// compiles the string into a postfix program.  The same program is evaluated on the host (boundary values at nodes,
// ParsedFunction::operator()) and by the assembly kernels at the Gauss points (fh_expr_device_eval in fh_expr_device.h), which
// is what lets applications/001_Poisson run its shipped input files ("func_source", "bdc_func" strings) on the GPU.
//
// Grammar (fparser documentation): precedence, lowest first:  |   &   = != < <= > >=   + -   * / %   unary - !   ^ (right
// associative, binds tighter than unary minus).  Truth is |x| >= 0.5, comparisons return 1 or 0 and use the library's default
// epsilon 1e-12 for = and !=.  Constants "pi" and "e" are the two the reference adds (ParsedFunction.cpp:46-47).
#include <locale>
#include <sstream>
#include "fh_internal.h"
#include "fh_expr_device.h"
#include <cctype>
#include <cmath>
#include <string>

struct fh_expr_s {
  std::vector<int> code;        // op | (argument << 8)
  std::vector<double> consts;
  int nvars = 0, max_stack = 0;
  std::string text;
};

namespace {

struct Parser {
  const std::string& s;
  size_t p = 0;
  std::vector<std::string> vars;
  fh_expr_s* out;
  std::string err;
  int depth = 0, max_depth = 0;
  int nest = 0;                           // recursion depth of the parser itself (parentheses, unary operators, exponents)
  static constexpr int MAX_NEST = 256;    // the expression comes from a user's JSON file: no unbounded host recursion
  struct Nest {
    Parser& q;
    bool ok;
    explicit Nest(Parser& pp) : q(pp), ok(++pp.nest <= MAX_NEST) {
      if (!ok) q.fail("expression nested deeper than " + std::to_string(MAX_NEST) + " levels");
    }
    ~Nest() { q.nest--; }
  };

  Parser(const std::string& str, fh_expr_s* o) : s(str), out(o) {}
  void skip() { while (p < s.size() && isspace((unsigned char)s[p])) p++; }
  bool fail(const std::string& m) {
    if (err.empty()) err = m + " at position " + std::to_string(p);
    return false;
  }
  void emit(int op, int arg, int delta) {
    out->code.push_back(op | (arg << 8));
    depth += delta;
    if (depth > max_depth) max_depth = depth;
  }
  bool accept(const char* tok) {
    skip();
    size_t n = strlen(tok);
    if (s.compare(p, n, tok) == 0) {
      p += n;
      return true;
    }
    return false;
  }
  bool parse_or() {
    Nest guard(*this);
    if (!guard.ok) return false;
    if (!parse_and()) return false;
    while (true) {
      skip();
      if (p < s.size() && s[p] == '|') {
        p++;
        if (!parse_and()) return false;
        emit(FHX_OR, 0, -1);
      } else return true;
    }
  }
  bool parse_and() {
    if (!parse_cmp()) return false;
    while (true) {
      skip();
      if (p < s.size() && s[p] == '&') {
        p++;
        if (!parse_cmp()) return false;
        emit(FHX_AND, 0, -1);
      } else return true;
    }
  }
  bool parse_cmp() {
    if (!parse_add()) return false;
    while (true) {
      int op = -1;
      if (accept("!=")) op = FHX_NE;
      else if (accept("<=")) op = FHX_LE;
      else if (accept(">=")) op = FHX_GE;
      else if (accept("<")) op = FHX_LT;
      else if (accept(">")) op = FHX_GT;
      else if (accept("=")) op = FHX_EQ;
      if (op < 0) return true;
      if (!parse_add()) return false;
      emit(op, 0, -1);
    }
  }
  bool parse_add() {
    if (!parse_mul()) return false;
    while (true) {
      skip();
      if (p < s.size() && (s[p] == '+' || s[p] == '-')) {
        const int op = s[p] == '+' ? FHX_ADD : FHX_SUB;
        p++;
        if (!parse_mul()) return false;
        emit(op, 0, -1);
      } else return true;
    }
  }
  bool parse_mul() {
    if (!parse_unary()) return false;
    while (true) {
      skip();
      if (p < s.size() && (s[p] == '*' || s[p] == '/' || s[p] == '%')) {
        const int op = s[p] == '*' ? FHX_MUL : s[p] == '/' ? FHX_DIV : FHX_MOD;
        p++;
        if (!parse_unary()) return false;
        emit(op, 0, -1);
      } else return true;
    }
  }
  bool parse_unary() {
    Nest guard(*this);
    if (!guard.ok) return false;
    skip();
    if (p < s.size() && s[p] == '-') {
      p++;
      if (!parse_unary()) return false;
      emit(FHX_NEG, 0, 0);
      return true;
    }
    if (p < s.size() && s[p] == '+') {
      p++;
      return parse_unary();
    }
    if (p < s.size() && s[p] == '!' && !(p + 1 < s.size() && s[p + 1] == '=')) {
      p++;
      if (!parse_unary()) return false;
      emit(FHX_NOT, 0, 0);
      return true;
    }
    return parse_pow();
  }
  bool parse_pow() {
    if (!parse_primary()) return false;
    skip();
    if (p < s.size() && s[p] == '^') {
      p++;
      if (!parse_unary_pow()) return false;   // right associative; the exponent may carry its own sign
      emit(FHX_POW, 0, -1);
    }
    return true;
  }
  bool parse_unary_pow() {
    Nest guard(*this);
    if (!guard.ok) return false;
    skip();
    if (p < s.size() && s[p] == '-') {
      p++;
      if (!parse_unary_pow()) return false;
      emit(FHX_NEG, 0, 0);
      return true;
    }
    return parse_pow();
  }
  bool parse_primary() {
    skip();
    if (p >= s.size()) return fail("unexpected end of expression");
    const char c = s[p];
    if (c == '(') {
      p++;
      if (!parse_or()) return false;
      if (!accept(")")) return fail("missing ')'");
      return true;
    }
    if (isdigit((unsigned char)c) || c == '.') {
      // the literal is delimited by hand ([digits][.digits][e[+-]digits]) and converted in the classic "C" locale: strtod would
      // follow LC_NUMERIC of the host application (a decimal comma there must not change what "0.5" means here)
      size_t q = p;
      while (q < s.size() && isdigit((unsigned char)s[q])) q++;
      if (q < s.size() && s[q] == '.') {
        q++;
        while (q < s.size() && isdigit((unsigned char)s[q])) q++;
      }
      if (q < s.size() && (s[q] == 'e' || s[q] == 'E')) {
        size_t r = q + 1;
        if (r < s.size() && (s[r] == '+' || s[r] == '-')) r++;
        if (r < s.size() && isdigit((unsigned char)s[r])) {
          while (r < s.size() && isdigit((unsigned char)s[r])) r++;
          q = r;
        }
      }
      std::istringstream is(s.substr(p, q - p));
      is.imbue(std::locale::classic());
      double v = 0.0;
      is >> v;
      if (is.fail() || q == p || (q - p == 1 && s[p] == '.')) return fail("bad number");
      p = q;
      out->consts.push_back(v);
      emit(FHX_CONST, (int)out->consts.size() - 1, 1);
      return true;
    }
    if (isalpha((unsigned char)c) || c == '_') {
      size_t q = p;
      while (q < s.size() && (isalnum((unsigned char)s[q]) || s[q] == '_')) q++;
      const std::string name = s.substr(p, q - p);
      p = q;
      skip();
      if (p < s.size() && s[p] == '(') {
        p++;
        int f = -1, nargs = 1;
        static const struct { const char* n; int id; int na; } F[] = {
            {"abs", FHF_ABS, 1},   {"acos", FHF_ACOS, 1},   {"acosh", FHF_ACOSH, 1}, {"asin", FHF_ASIN, 1},   {"asinh", FHF_ASINH, 1},
            {"atan", FHF_ATAN, 1}, {"atanh", FHF_ATANH, 1}, {"cbrt", FHF_CBRT, 1},   {"ceil", FHF_CEIL, 1},   {"cos", FHF_COS, 1},
            {"cosh", FHF_COSH, 1}, {"cot", FHF_COT, 1},     {"csc", FHF_CSC, 1},     {"exp", FHF_EXP, 1},     {"exp2", FHF_EXP2, 1},
            {"floor", FHF_FLOOR, 1}, {"int", FHF_INT, 1},   {"log", FHF_LOG, 1},     {"log10", FHF_LOG10, 1}, {"log2", FHF_LOG2, 1},
            {"sec", FHF_SEC, 1},   {"sin", FHF_SIN, 1},     {"sinh", FHF_SINH, 1},   {"sqrt", FHF_SQRT, 1},   {"tan", FHF_TAN, 1},
            {"tanh", FHF_TANH, 1}, {"trunc", FHF_TRUNC, 1}, {"atan2", FHF_ATAN2, 2}, {"hypot", FHF_HYPOT, 2}, {"max", FHF_MAX, 2},
            {"min", FHF_MIN, 2},   {"pow", FHF_POW2, 2},    {"if", FHF_IF, 3}};
        for (auto& e : F)
          if (name == e.n) {
            f = e.id;
            nargs = e.na;
          }
        if (f < 0) return fail("unknown function '" + name + "'");
        for (int a = 0; a < nargs; a++) {
          if (a > 0 && !accept(",")) return fail("function '" + name + "' expects " + std::to_string(nargs) + " arguments");
          if (!parse_or()) return false;
        }
        if (!accept(")")) return fail("missing ')' after the arguments of '" + name + "'");
        emit(FHX_FUNC, f, 1 - nargs);
        return true;
      }
      for (size_t v = 0; v < vars.size(); v++)
        if (vars[v] == name) {
          emit(FHX_VAR, (int)v, 1);
          return true;
        }
      if (name == "pi" || name == "e") {
        out->consts.push_back(name == "pi" ? std::acos(-1.0) : std::exp(1.0));
        emit(FHX_CONST, (int)out->consts.size() - 1, 1);
        return true;
      }
      return fail("unknown identifier '" + name + "'");
    }
    return fail(std::string("unexpected character '") + c + "'");
  }
};

}  // namespace

extern "C" int fh_expr_compile(const char* expression, const char* variables, fh_expr_t* out) {
  FH_REQUIRE(expression && variables && out, "fh_expr_compile: null argument");
  fh_expr_s* e = new fh_expr_s();
  e->text = expression;
  Parser ps(e->text, e);
  std::string v = variables;
  size_t a = 0;
  while (a <= v.size()) {
    size_t b = v.find(',', a);
    if (b == std::string::npos) b = v.size();
    std::string name = v.substr(a, b - a);
    while (!name.empty() && isspace((unsigned char)name.back())) name.pop_back();
    while (!name.empty() && isspace((unsigned char)name.front())) name.erase(name.begin());
    if (!name.empty()) ps.vars.push_back(name);
    a = b + 1;
  }
  e->nvars = (int)ps.vars.size();
  bool ok = ps.parse_or();
  if (ok) {
    ps.skip();
    if (ps.p != e->text.size()) ok = ps.fail("unexpected trailing characters");
  }
  if (!ok || ps.depth != 1) {
    const std::string msg = ps.err.empty() ? std::string("malformed expression") : ps.err;
    delete e;
    fh_set_error("fh_expr_compile: \"%.120s%s\": %s", expression, strlen(expression) > 120 ? "..." : "", msg.c_str());
    return 2;
  }
  e->max_stack = ps.max_depth;
  if (e->max_stack > FHX_STACK) {
    delete e;
    fh_set_error("fh_expr_compile: \"%.120s%s\" needs an evaluation stack of %d (limit %d)", expression, strlen(expression) > 120 ? "..." : "",
                 ps.max_depth, FHX_STACK);
    return 2;
  }
  *out = e;
  return 0;
}

extern "C" int fh_expr_eval(fh_expr_t e, const double* x, double* value) {
  FH_REQUIRE(e && x && value, "fh_expr_eval: null argument");
  *value = fh_expr_device_eval(e->code.data(), (int)e->code.size(), e->consts.data(), x);
  return 0;
}

extern "C" int fh_expr_eval_many(fh_expr_t e, int npts, const double* x, double* values) {
  FH_REQUIRE(e && (npts == 0 || (x && values)), "fh_expr_eval_many: null argument");
  for (int i = 0; i < npts; i++)
    values[i] = fh_expr_device_eval(e->code.data(), (int)e->code.size(), e->consts.data(), x + (size_t)i * e->nvars);
  return 0;
}

extern "C" int fh_expr_program(fh_expr_t e, int* ncode, int* nconst, int* code, double* consts) {
  FH_REQUIRE(e && ncode && nconst, "fh_expr_program: null argument");
  if (code) fh_copy_out(code, e->code);
  if (consts) fh_copy_out(consts, e->consts);
  *ncode = (int)e->code.size();
  *nconst = (int)e->consts.size();
  return 0;
}

extern "C" int fh_expr_nvars(fh_expr_t e, int* nvars) {
  FH_REQUIRE(e && nvars, "fh_expr_nvars: null argument");
  *nvars = e->nvars;
  return 0;
}

extern "C" int fh_expr_destroy(fh_expr_t e) {
  delete e;
  return 0;
}
