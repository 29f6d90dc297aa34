// DECLARATION-ONLY stand-in for <Rcpp.h>, used by ONE test (tests/test_glue_typecheck.py: `g++ -fsyntax-only examples/FM_glue.cpp`).
//
// It exists to catch drift between examples/FM_glue.cpp and include/fmx.h (a renamed entry point, a changed argument type, a dropped
// constant) in an image that has neither R nor Rcpp.  It declares the names and signatures of the small part of the Rcpp / R API the
// glue uses and defines NOTHING: no function here has a body, nothing can be linked or run against it, it pins no numerical behaviour
// and the oracle never sees it.  Reference for the real interface: Rcpp's own headers (List / Vector / Matrix proxies, Named, sugar
// min / max), R's Rinternals.h / Rmath.h; call shapes as in /root/reference/src/RcppExports.cpp:10-52 and src/FM.cpp.
#ifndef FMX_TEST_RCPP_SHIM_H_
#define FMX_TEST_RCPP_SHIM_H_
#include <cstddef>
#include <string>

struct SEXPREC;
typedef SEXPREC* SEXP;
extern SEXP R_NilValue;
bool Rf_isNull(SEXP);
double* REAL(SEXP);
long Rf_xlength(SEXP);   // (R_xlen_t in R)
double Rf_rnorm(double mean, double sd);
double Rf_rgamma(double shape, double scale);
double norm_rand(void);

namespace Rcpp {

[[noreturn]] void stop(const char* message);
[[noreturn]] void stop(const std::string& message);

class List;

// what operator[] / attr() of a list hand back: converts to anything, takes anything
struct Proxy {
  template <typename T> operator T() const;
  template <typename U> Proxy& operator=(const U& value);
};

template <typename T> struct NamedValue {};
struct NamedSlot {
  template <typename T> NamedValue<T> operator=(const T& value) const;
};
struct NamedPlaceholder {
  NamedSlot operator[](const char* name) const;
};
static const NamedPlaceholder _ = NamedPlaceholder();

class NumericVector {
 public:
  typedef double* iterator;
  NumericVector();
  explicit NumericVector(long n);
  template <typename It> NumericVector(It first, It last);
  static NumericVector create(double a, double b);
  long size() const;
  iterator begin();
  iterator end();
  double& operator[](long i);
  const double& operator[](long i) const;
};

class IntegerVector {
 public:
  typedef int* iterator;
  IntegerVector();
  explicit IntegerVector(long n);
  long size() const;
  iterator begin();
  iterator end();
  int& operator[](long i);
  const int& operator[](long i) const;
};

class NumericMatrix {
 public:
  typedef double* iterator;
  NumericMatrix();
  NumericMatrix(int nrow, int ncol);
  int nrow() const;
  int ncol() const;
  iterator begin();
  double& operator()(int i, int j);
};

class String {
 public:
  String();
  const char* get_cstring() const;
};

class List {
 public:
  List();
  explicit List(std::size_t n);
  template <typename... Args> static List create(const Args&... named_values);
  Proxy operator[](const char* name) const;
  Proxy operator[](std::size_t index) const;
  Proxy operator[](long index) const;
  Proxy operator[](int index) const;
  Proxy attr(const char* name) const;
  bool containsElementNamed(const char* name) const;
  long size() const;
};

template <typename T> T as(const Proxy& x);
template <typename T> T as(SEXP x);

double min(const NumericVector& x);
double max(const NumericVector& x);

}  // namespace Rcpp
#endif  // FMX_TEST_RCPP_SHIM_H_
