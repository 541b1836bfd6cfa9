// mini_eigen.hpp -- a minimal, EAGER stand-in for the part of the Eigen 3 API that robotoc's Riccati /
// dynamics / core sources use.  TEST INFRASTRUCTURE ONLY (oracle/_ref: the reference's own .cpp files compiled
// against this header, because Eigen itself is absent from the image; see oracle/ref_shim/README.md).
//
// Design: every dense lvalue (matrix, block, row/column, segment, transpose, diagonal) is a strided view
// {pointer, rows, cols, row stride, column stride} on double storage; every rvalue expression (sum, product,
// scaled matrix, LLT solve) is evaluated immediately into a column-major temporary.  No expression templates, no
// vectorisation, no aliasing hazards (noalias() is the identity).  Products accumulate in k-order like Eigen's
// lazy/coefficient-based product; results agree with real Eigen to rounding (different summation blocking), which
// is what the parity tolerances account for.  Nothing here is copied from Eigen.
#ifndef RTOC_MINI_EIGEN_HPP_
#define RTOC_MINI_EIGEN_HPP_

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <initializer_list>
#include <iostream>
#include <limits>
#include <memory>
#include <type_traits>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_DEFAULT_DENSE_INDEX_TYPE std::ptrdiff_t

namespace Eigen {

typedef std::ptrdiff_t Index;
const int Dynamic = -1;
const int Infinity = -1;
enum StorageOptions { ColMajor = 0, RowMajor = 1, AutoAlign = 0, DontAlign = 2 };
enum ComputationInfo { Success = 0, NumericalIssue = 1, NoConvergence = 2, InvalidInput = 3 };
enum UpLoType { Lower = 1, Upper = 2, StrictlyLower = 9, StrictlyUpper = 10 };

template <class T>
using aligned_allocator = std::allocator<T>;

template <class S, int R, int C, int O = ((R == 1 && C != 1) ? RowMajor : ColMajor), int MR = R, int MC = C>
class Matrix;
typedef Matrix<double, Dynamic, Dynamic> MatrixXd;
typedef Matrix<double, Dynamic, 1> VectorXd;
typedef Matrix<double, 1, Dynamic> RowVectorXd;
// Eigen::Map<const VectorXd>(ptr, n): only ever used by the reference's STO constraints to hand a std::vector to a
// `const VectorXd&` parameter (src/sto/sto_constraints.cpp:23,75) -- an owning copy serves.
template <class T> class Map;

typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<double, 6, 1> Vector6d;
typedef Matrix<double, 2, 2> Matrix2d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<int, Dynamic, 1> VectorXi;

class View;
class ArrayView;
class Diag;
template <class D>
class MatrixBase;

// ---- strided view: the one lvalue type -------------------------------------------------------
struct RawView {
  double* p;
  Index r, c, rs, cs;
  double& at(Index i, Index j) const { return p[i * rs + j * cs]; }
};

inline MatrixXd eval_product(const RawView& a, const RawView& b);

template <class D>
class MatrixBase {
 public:
  typedef double Scalar;
  D& derived() { return *static_cast<D*>(this); }
  const D& derived() const { return *static_cast<const D*>(this); }
  RawView raw() const { return derived().raw_(); }
  Index rows() const { return raw().r; }
  Index cols() const { return raw().c; }
  Index size() const { return raw().r * raw().c; }

  // ---- element access ----
  double& operator()(Index i, Index j) { return raw().at(i, j); }
  double operator()(Index i, Index j) const { return raw().at(i, j); }
  double& coeffRef(Index i, Index j) { return raw().at(i, j); }
  double coeff(Index i, Index j) const { return raw().at(i, j); }
  double& lin_(Index i) const {
    const RawView v = raw();
    assert(v.r == 1 || v.c == 1);
    return v.c == 1 ? v.at(i, 0) : v.at(0, i);
  }
  double& operator()(Index i) { return lin_(i); }
  double operator()(Index i) const { return lin_(i); }
  double& operator[](Index i) { return lin_(i); }
  double operator[](Index i) const { return lin_(i); }
  double& coeffRef(Index i) { return lin_(i); }
  double coeff(Index i) const { return lin_(i); }
  double& x() { return lin_(0); }
  double& y() { return lin_(1); }
  double& z() { return lin_(2); }
  double x() const { return lin_(0); }
  double y() const { return lin_(1); }
  double z() const { return lin_(2); }

  // ---- sub-views (constness is not tracked: a view of a const object is writable storage-wise) ----
  inline View block(Index i, Index j, Index nr, Index nc) const;
  template <int NR, int NC>
  inline View block(Index i, Index j) const;
  template <int NR, int NC>
  inline View block(Index i, Index j, Index, Index) const;
  inline View topRows(Index n) const;
  inline View bottomRows(Index n) const;
  inline View middleRows(Index i, Index n) const;
  inline View leftCols(Index n) const;
  inline View rightCols(Index n) const;
  inline View middleCols(Index j, Index n) const;
  template <int N>
  inline View middleCols(Index j) const;
  bool isIdentity(double prec = 1e-12) const {
    const RawView v = raw();
    for (Index j = 0; j < v.c; ++j)
      for (Index i = 0; i < v.r; ++i)
        if (std::fabs(v.at(i, j) - (i == j ? 1.0 : 0.0)) > prec) return false;
    return true;
  }
  inline View topLeftCorner(Index nr, Index nc) const;
  inline View topRightCorner(Index nr, Index nc) const;
  inline View bottomLeftCorner(Index nr, Index nc) const;
  inline View bottomRightCorner(Index nr, Index nc) const;
  template <int NR, int NC>
  inline View topLeftCorner() const;
  template <int NR, int NC>
  inline View topRightCorner() const;
  template <int NR, int NC>
  inline View bottomLeftCorner() const;
  template <int NR, int NC>
  inline View bottomRightCorner() const;
  template <int N>
  inline View topRows() const;
  template <int N>
  inline View bottomRows() const;
  template <int N>
  inline View leftCols() const;
  template <int N>
  inline View rightCols() const;
  inline View row(Index i) const;
  inline View col(Index j) const;
  inline View head(Index n) const;
  inline View tail(Index n) const;
  inline View segment(Index i, Index n) const;
  template <int N>
  inline View head() const;
  template <int N>
  inline View tail() const;
  template <int N>
  inline View segment(Index i) const;
  inline View transpose() const;
  inline View diagonal() const;
  // Eigen: `m.noalias() = expr` still RESIZES a plain matrix (robotoc relies on it, e.g. DtM in
  // riccati_factorizer.cpp:84), so noalias() hands back the object itself, not a fixed-size view
  D& noalias() const { return const_cast<D&>(derived()); }
  inline ArrayView array() const;
  inline View matrix() const;
  inline Diag asDiagonal() const;
  inline MatrixXd eval() const;

  // ---- in-place fills ----
  D& setZero() { return setConstant(0.0); }
  D& setOnes() { return setConstant(1.0); }
  D& setConstant(double x) {
    const RawView v = raw();
    for (Index j = 0; j < v.c; ++j)
      for (Index i = 0; i < v.r; ++i) v.at(i, j) = x;
    return derived();
  }
  void fill(double x) { setConstant(x); }
  D& setIdentity() {
    const RawView v = raw();
    for (Index j = 0; j < v.c; ++j)
      for (Index i = 0; i < v.r; ++i) v.at(i, j) = (i == j) ? 1.0 : 0.0;
    return derived();
  }
  D& setRandom() {
    const RawView v = raw();
    for (Index j = 0; j < v.c; ++j)
      for (Index i = 0; i < v.r; ++i) v.at(i, j) = 2.0 * (double)std::rand() / (double)RAND_MAX - 1.0;
    return derived();
  }

  // ---- reductions ----
  double squaredNorm() const {
    const RawView v = raw();
    double s = 0.0;
    for (Index j = 0; j < v.c; ++j)
      for (Index i = 0; i < v.r; ++i) s += v.at(i, j) * v.at(i, j);
    return s;
  }
  double norm() const { return std::sqrt(squaredNorm()); }
  double sum() const {
    const RawView v = raw();
    double s = 0.0;
    for (Index j = 0; j < v.c; ++j)
      for (Index i = 0; i < v.r; ++i) s += v.at(i, j);
    return s;
  }
  double trace() const {
    const RawView v = raw();
    double s = 0.0;
    for (Index i = 0; i < std::min(v.r, v.c); ++i) s += v.at(i, i);
    return s;
  }
  template <int P>
  double lpNorm() const {
    const RawView v = raw();
    double s = 0.0;
    for (Index j = 0; j < v.c; ++j)
      for (Index i = 0; i < v.r; ++i) {
        const double a = std::fabs(v.at(i, j));
        if (P == Infinity) s = std::max(s, a);
        else if (P == 1) s += a;
        else if (P == 2) s += a * a;
      }
    return P == 2 ? std::sqrt(s) : s;
  }
  double minCoeff() const {
    const RawView v = raw();
    double s = std::numeric_limits<double>::infinity();
    for (Index j = 0; j < v.c; ++j)
      for (Index i = 0; i < v.r; ++i) s = std::min(s, v.at(i, j));
    return s;
  }
  double maxCoeff() const {
    const RawView v = raw();
    double s = -std::numeric_limits<double>::infinity();
    for (Index j = 0; j < v.c; ++j)
      for (Index i = 0; i < v.r; ++i) s = std::max(s, v.at(i, j));
    return s;
  }
  bool hasNaN() const {
    const RawView v = raw();
    for (Index j = 0; j < v.c; ++j)
      for (Index i = 0; i < v.r; ++i)
        if (v.at(i, j) != v.at(i, j)) return true;
    return false;
  }
  bool allFinite() const {
    const RawView v = raw();
    for (Index j = 0; j < v.c; ++j)
      for (Index i = 0; i < v.r; ++i)
        if (!std::isfinite(v.at(i, j))) return false;
    return true;
  }
  bool isZero(double prec = 1e-12) const {
    const RawView v = raw();
    for (Index j = 0; j < v.c; ++j)
      for (Index i = 0; i < v.r; ++i)
        if (std::fabs(v.at(i, j)) > prec) return false;
    return true;
  }
  template <class E>
  bool isApprox(const MatrixBase<E>& o, double prec = 1e-12) const {
    const RawView a = raw(), b = o.raw();
    if (a.r != b.r || a.c != b.c) return false;
    double d = 0.0, na = 0.0, nb = 0.0;
    for (Index j = 0; j < a.c; ++j)
      for (Index i = 0; i < a.r; ++i) {
        const double x = a.at(i, j), y = b.at(i, j);
        d += (x - y) * (x - y);
        na += x * x;
        nb += y * y;
      }
    return d <= prec * prec * std::min(na, nb);
  }
  template <class E>
  double dot(const MatrixBase<E>& o) const {
    const RawView a = raw(), b = o.raw();
    assert(a.r * a.c == b.r * b.c);
    const Index n = a.r * a.c;
    double s = 0.0;
    for (Index i = 0; i < n; ++i)
      s += (a.c == 1 ? a.at(i, 0) : a.at(0, i)) * (b.c == 1 ? b.at(i, 0) : b.at(0, i));
    return s;
  }

  // ---- assignment through a MatrixBase reference (the reference's `const_cast<MatrixBase<T>&>(x) = ...` idiom) ----
  template <class E>
  D& operator=(const MatrixBase<E>& o) {
    derived() = o;
    return derived();
  }
  MatrixBase& operator=(const MatrixBase& o) {
    if (this != &o) derived() = o.derived();
    return *this;
  }
  MatrixBase() = default;
  MatrixBase(const MatrixBase&) = default;

  // ---- compound assignment (rhs is always materialised or independent storage) ----
  template <class E>
  D& operator+=(const MatrixBase<E>& o) {
    const RawView a = raw(), b = o.raw();
    assert(a.r == b.r && a.c == b.c);
    for (Index j = 0; j < a.c; ++j)
      for (Index i = 0; i < a.r; ++i) a.at(i, j) += b.at(i, j);
    return derived();
  }
  template <class E>
  D& operator-=(const MatrixBase<E>& o) {
    const RawView a = raw(), b = o.raw();
    assert(a.r == b.r && a.c == b.c);
    for (Index j = 0; j < a.c; ++j)
      for (Index i = 0; i < a.r; ++i) a.at(i, j) -= b.at(i, j);
    return derived();
  }
  D& operator*=(double s) {
    const RawView a = raw();
    for (Index j = 0; j < a.c; ++j)
      for (Index i = 0; i < a.r; ++i) a.at(i, j) *= s;
    return derived();
  }
  D& operator/=(double s) {
    const RawView a = raw();
    for (Index j = 0; j < a.c; ++j)
      for (Index i = 0; i < a.r; ++i) a.at(i, j) /= s;
    return derived();
  }

  // ---- element-wise helpers ----
  template <class E>
  inline MatrixXd cwiseProduct(const MatrixBase<E>& o) const;
  template <class E>
  inline MatrixXd cross(const MatrixBase<E>& o) const;
  template <class E>
  inline MatrixXd cwiseQuotient(const MatrixBase<E>& o) const;
  inline MatrixXd cwiseAbs() const;
  inline MatrixXd cwiseInverse() const;
  inline MatrixXd cwiseSqrt() const;
  inline MatrixXd inverse() const;
};

// copy src into dst (same shape), through a temporary if the storages may overlap
inline void assign_view(const RawView& dst, const RawView& src) {
  assert(dst.r == src.r && dst.c == src.c);
  const Index n = dst.r * dst.c;
  if (n == 0) return;
  const double* dlo = dst.p;
  const double* slo = src.p;
  auto span = [](const RawView& v) {
    return (v.r - 1) * (v.rs < 0 ? -v.rs : v.rs) + (v.c - 1) * (v.cs < 0 ? -v.cs : v.cs) + 1;
  };
  const bool overlap = !(dlo + span(dst) <= slo || slo + span(src) <= dlo);
  if (overlap && !(dst.p == src.p && dst.rs == src.rs && dst.cs == src.cs)) {
    std::vector<double> tmp((size_t)n);
    for (Index j = 0; j < src.c; ++j)
      for (Index i = 0; i < src.r; ++i) tmp[(size_t)(i + j * src.r)] = src.at(i, j);
    for (Index j = 0; j < dst.c; ++j)
      for (Index i = 0; i < dst.r; ++i) dst.at(i, j) = tmp[(size_t)(i + j * dst.r)];
    return;
  }
  for (Index j = 0; j < dst.c; ++j)
    for (Index i = 0; i < dst.r; ++i) dst.at(i, j) = src.at(i, j);
}

class View : public MatrixBase<View> {
 public:
  RawView v_;
  View() : v_{nullptr, 0, 0, 1, 1} {}
  View(double* p, Index r, Index c, Index rs, Index cs) : v_{p, r, c, rs, cs} {}
  explicit View(const RawView& v) : v_(v) {}
  template <class D>
  View(const MatrixBase<D>& m) : v_(m.raw()) {}  // Ref<> / Block<> construction from any dense object
  View(const View& o) = default;
  RawView raw_() const { return v_; }
  double* data() const { return v_.p; }
  Index outerStride() const { return v_.cs == 1 ? v_.rs : v_.cs; }
  // assignment copies VALUES (Eigen semantics for blocks), it does not rebind
  View& operator=(const View& o) {
    assign_view(v_, o.v_);
    return *this;
  }
  template <class D>
  View& operator=(const MatrixBase<D>& o) {
    assign_view(v_, o.raw());
    return *this;
  }
  inline View& operator=(const Diag& d);
  // `block << a, b, c, ...;` fills row by row, as Eigen's comma initialiser
  struct Comma {
    RawView v;
    Index k;
    Comma& operator,(double x) {
      v.at(k / v.c, k % v.c) = x;
      ++k;
      return *this;
    }
  };
  Comma operator<<(double x) {
    Comma c{v_, 0};
    c, x;
    return c;
  }
};

// ---- coefficient-wise world: what `.array()` hands out -----------------------------------------
// ArrayXd: an owning temporary of an array expression; ArrayView: the lvalue `m.array()`.  Binary operators are
// ELEMENTWISE (the matrix world's View * View is a matrix product).
class ArrayXd {
 public:
  Index r_, c_;
  std::vector<double> d_;
  ArrayXd(Index r, Index c) : r_(r), c_(c), d_(static_cast<size_t>(r * c), 0.0) {}
  double& at(Index i, Index j) { return d_[static_cast<size_t>(i + j * r_)]; }
  double at(Index i, Index j) const { return d_[static_cast<size_t>(i + j * r_)]; }
  Index rows() const { return r_; }
  Index cols() const { return c_; }
  double sum() const {
    double s = 0.0;
    for (double x : d_) s += x;
    return s;
  }
  double minCoeff() const {
    double m = d_.at(0);
    for (double x : d_) m = std::min(m, x);
    return m;
  }
  double maxCoeff() const {
    double m = d_.at(0);
    for (double x : d_) m = std::max(m, x);
    return m;
  }
  ArrayXd log() const {
    ArrayXd o(r_, c_);
    for (size_t k = 0; k < d_.size(); ++k) o.d_[k] = std::log(d_[k]);
    return o;
  }
  ArrayXd abs() const {
    ArrayXd o(r_, c_);
    for (size_t k = 0; k < d_.size(); ++k) o.d_[k] = std::fabs(d_[k]);
    return o;
  }
  ArrayXd operator-() const {
    ArrayXd o(r_, c_);
    for (size_t k = 0; k < d_.size(); ++k) o.d_[k] = -d_[k];
    return o;
  }
};

class ArrayView {
 public:
  RawView v_;
  explicit ArrayView(const RawView& v) : v_(v) {}
  Index rows() const { return v_.r; }
  Index cols() const { return v_.c; }
  ArrayXd eval() const {
    ArrayXd o(v_.r, v_.c);
    for (Index j = 0; j < v_.c; ++j)
      for (Index i = 0; i < v_.r; ++i) o.at(i, j) = v_.at(i, j);
    return o;
  }
  operator ArrayXd() const { return eval(); }
  void check(Index r, Index c) const {
    if (r != v_.r || c != v_.c) {
      std::fprintf(stderr, "mini_eigen: array shape mismatch %ldx%ld vs %ldx%ld\n", (long)v_.r, (long)v_.c, (long)r, (long)c);
      std::abort();
    }
  }
  ArrayView& operator=(const ArrayXd& o) {
    check(o.r_, o.c_);
    for (Index j = 0; j < v_.c; ++j)
      for (Index i = 0; i < v_.r; ++i) v_.at(i, j) = o.at(i, j);
    return *this;
  }
  ArrayView& operator=(const ArrayView& o) { return *this = o.eval(); }
  ArrayView& operator+=(const ArrayXd& o) {
    check(o.r_, o.c_);
    for (Index j = 0; j < v_.c; ++j)
      for (Index i = 0; i < v_.r; ++i) v_.at(i, j) += o.at(i, j);
    return *this;
  }
  ArrayView& operator-=(const ArrayXd& o) {
    check(o.r_, o.c_);
    for (Index j = 0; j < v_.c; ++j)
      for (Index i = 0; i < v_.r; ++i) v_.at(i, j) -= o.at(i, j);
    return *this;
  }
  ArrayView& operator+=(const ArrayView& o) { return *this += o.eval(); }
  ArrayView& operator*=(double s) {
    for (Index j = 0; j < v_.c; ++j)
      for (Index i = 0; i < v_.r; ++i) v_.at(i, j) *= s;
    return *this;
  }
  ArrayView& operator/=(double s) { return *this *= (1.0 / s); }
  ArrayView& operator+=(double s) {
    for (Index j = 0; j < v_.c; ++j)
      for (Index i = 0; i < v_.r; ++i) v_.at(i, j) += s;
    return *this;
  }
  double sum() const { return eval().sum(); }
  double minCoeff() const { return eval().minCoeff(); }
  double maxCoeff() const { return eval().maxCoeff(); }
  ArrayXd log() const { return eval().log(); }
  ArrayXd abs() const { return eval().abs(); }
  ArrayXd operator-() const { return -eval(); }
};

#define MINI_EIGEN_ARRAY_BINOP(OP)                                                                         \
  inline ArrayXd operator OP(const ArrayXd& a, const ArrayXd& b) {                                          \
    if (a.r_ != b.r_ || a.c_ != b.c_) {                                                                     \
      std::fprintf(stderr, "mini_eigen: array operands differ in shape\n");                                 \
      std::abort();                                                                                         \
    }                                                                                                       \
    ArrayXd o(a.r_, a.c_);                                                                                  \
    for (size_t k = 0; k < o.d_.size(); ++k) o.d_[k] = a.d_[k] OP b.d_[k];                                  \
    return o;                                                                                               \
  }                                                                                                         \
  inline ArrayXd operator OP(const ArrayXd& a, double s) {                                                  \
    ArrayXd o(a.r_, a.c_);                                                                                  \
    for (size_t k = 0; k < o.d_.size(); ++k) o.d_[k] = a.d_[k] OP s;                                        \
    return o;                                                                                               \
  }                                                                                                         \
  inline ArrayXd operator OP(double s, const ArrayXd& a) {                                                  \
    ArrayXd o(a.r_, a.c_);                                                                                  \
    for (size_t k = 0; k < o.d_.size(); ++k) o.d_[k] = s OP a.d_[k];                                        \
    return o;                                                                                               \
  }                                                                                                         \
  inline ArrayXd operator OP(const ArrayView& a, const ArrayView& b) { return a.eval() OP b.eval(); }       \
  inline ArrayXd operator OP(const ArrayView& a, const ArrayXd& b) { return a.eval() OP b; }                \
  inline ArrayXd operator OP(const ArrayXd& a, const ArrayView& b) { return a OP b.eval(); }                \
  inline ArrayXd operator OP(const ArrayView& a, double s) { return a.eval() OP s; }                        \
  inline ArrayXd operator OP(double s, const ArrayView& a) { return s OP a.eval(); }
MINI_EIGEN_ARRAY_BINOP(+)
MINI_EIGEN_ARRAY_BINOP(-)
MINI_EIGEN_ARRAY_BINOP(*)
MINI_EIGEN_ARRAY_BINOP(/)
#undef MINI_EIGEN_ARRAY_BINOP

template <class T, int BR = Dynamic, int BC = Dynamic, bool Inner = false>
using Block = View;
template <class T, int N = Dynamic>
using VectorBlock = View;
template <class T, int O = 0, class S = void>
using Ref = View;

// ---- owning matrix ---------------------------------------------------------------------------
template <class S, int R, int C, int O, int MR, int MC>
class Matrix : public MatrixBase<Matrix<S, R, C, O, MR, MC>> {
  static_assert(std::is_same<S, double>::value || std::is_same<S, int>::value, "mini_eigen: double (and int vectors) only");
  std::vector<double> d_;
  Index r_, c_;
  static constexpr bool kRowMajor = (O & RowMajor) != 0;

 public:
  typedef MatrixBase<Matrix> Base;
  Matrix() : d_((size_t)((R > 0 ? R : 0) * (C > 0 ? C : 0)), 0.0), r_(R > 0 ? R : 0), c_(C > 0 ? C : 0) {
    if (R == Dynamic && C == 1) c_ = 1;
    if (C == Dynamic && R == 1) r_ = 1;
  }
  explicit Matrix(Index n) : r_(C == 1 ? n : (R == 1 ? 1 : n)), c_(C == 1 ? 1 : (R == 1 ? n : n)) {
    if (R > 0 && C > 0) {  // fixed size: Vector1d(x)-style construction is not used by robotoc
      r_ = R;
      c_ = C;
    }
    d_.assign((size_t)(r_ * c_), 0.0);
  }
  Matrix(Index r, Index c) : d_((size_t)(r * c), 0.0), r_(r), c_(c) {}
  Matrix(int r, int c) : d_((size_t)r * (size_t)c, 0.0), r_(r), c_(c) {}
  Matrix(double a, double b) : d_{a, b}, r_(R > 0 ? R : 2), c_(C > 0 ? C : 1) {}
  Matrix(double a, double b, double c) : d_{a, b, c}, r_(R > 0 ? R : 3), c_(C > 0 ? C : 1) {}
  Matrix(double a, double b, double c, double d) : d_{a, b, c, d}, r_(R > 0 ? R : 4), c_(C > 0 ? C : 1) {}
  Matrix(const Matrix& o) = default;
  Matrix(Matrix&& o) = default;
  template <class D>
  Matrix(const MatrixBase<D>& o) : r_(0), c_(0) {
    *this = o;
  }
  Matrix(const Diag& d);
  RawView raw_() const {
    double* p = const_cast<double*>(d_.data());
    return kRowMajor ? RawView{p, r_, c_, c_, 1} : RawView{p, r_, c_, 1, r_};
  }
  double* data() { return d_.data(); }
  const double* data() const { return d_.data(); }
  Index outerStride() const { return kRowMajor ? c_ : r_; }
  void resize(Index r, Index c) {
    if (r != r_ || c != c_) {
      r_ = r;
      c_ = c;
      d_.assign((size_t)(r * c), 0.0);
    }
  }
  void resize(Index n) {
    if (C == 1) resize(n, 1);
    else resize(1, n);
  }
  void conservativeResize(Index r, Index c) {
    Matrix t(r, c);
    for (Index j = 0; j < std::min(c, c_); ++j)
      for (Index i = 0; i < std::min(r, r_); ++i) t(i, j) = (*this)(i, j);
    *this = t;
  }
  void conservativeResize(Index n) {
    if (C == 1) conservativeResize(n, 1);
    else conservativeResize(1, n);
  }
  Matrix& operator=(const Matrix& o) {
    if (this != &o) {
      r_ = o.r_;
      c_ = o.c_;
      d_ = o.d_;
    }
    return *this;
  }
  Matrix& operator=(Matrix&& o) = default;
  template <class D>
  Matrix& operator=(const MatrixBase<D>& o) {
    const RawView s = o.raw();
    if ((const void*)s.p >= (const void*)d_.data() && (const void*)s.p < (const void*)(d_.data() + d_.size()) &&
        !d_.empty()) {  // source aliases our own storage: go through a temporary
      std::vector<double> tmp((size_t)(s.r * s.c));
      for (Index j = 0; j < s.c; ++j)
        for (Index i = 0; i < s.r; ++i) tmp[(size_t)(i + j * s.r)] = s.at(i, j);
      Index rr = s.r, cc = s.c;
      if (C == 1 && cc != 1 && rr == 1) std::swap(rr, cc);
      r_ = rr;
      c_ = cc;
      d_.assign((size_t)(rr * cc), 0.0);
      const RawView m = raw_();
      for (Index j = 0; j < s.c; ++j)
        for (Index i = 0; i < s.r; ++i) (rr == s.r ? m.at(i, j) : m.at(j, i)) = tmp[(size_t)(i + j * s.r)];
      return *this;
    }
    Index rr = s.r, cc = s.c;
    const bool flip = (C == 1 && cc != 1 && rr == 1) || (R == 1 && rr != 1 && cc == 1);
    if (flip) std::swap(rr, cc);
    resize(rr, cc);
    const RawView m = raw_();
    for (Index j = 0; j < s.c; ++j)
      for (Index i = 0; i < s.r; ++i) (flip ? m.at(j, i) : m.at(i, j)) = s.at(i, j);
    return *this;
  }
  Matrix& operator=(const Diag& d);

  // comma initialiser:  m << a, b, c;
  struct Comma {
    Matrix& m;
    Index k;
    Comma& operator,(double x) {
      const Index i = k / m.c_, j = k % m.c_;  // row by row, as Eigen
      m(i, j) = x;
      ++k;
      return *this;
    }
  };
  Comma operator<<(double x) {
    Comma c{*this, 0};
    c, x;
    return c;
  }

  static Matrix Zero() { return Matrix(); }
  static Matrix Zero(Index n) { return Matrix(n); }
  static Matrix Zero(Index r, Index c) { return Matrix(r, c); }
  static Matrix Constant(Index n, double x) {
    Matrix m(n);
    m.setConstant(x);
    return m;
  }
  static Matrix Constant(Index r, Index c, double x) {
    Matrix m(r, c);
    m.setConstant(x);
    return m;
  }
  static Matrix Constant(double x) {
    Matrix m;
    m.setConstant(x);
    return m;
  }
  static Matrix Ones(Index n) { return Constant(n, 1.0); }
  static Matrix Ones(Index r, Index c) { return Constant(r, c, 1.0); }
  static Matrix Ones() { return Constant(1.0); }
  static Matrix Identity() {
    Matrix m;
    m.setIdentity();
    return m;
  }
  static Matrix Identity(Index r, Index c) {
    Matrix m(r, c);
    m.setIdentity();
    return m;
  }
  static Matrix Random() {
    Matrix m;
    m.setRandom();
    return m;
  }
  static Matrix Random(Index n) {
    Matrix m(n);
    m.setRandom();
    return m;
  }
  static Matrix Random(Index r, Index c) {
    Matrix m(r, c);
    m.setRandom();
    return m;
  }
  static Matrix UnitX() {
    Matrix m;
    m(0) = 1.0;
    return m;
  }
  static Matrix UnitY() {
    Matrix m;
    m(1) = 1.0;
    return m;
  }
  static Matrix UnitZ() {
    Matrix m;
    m(2) = 1.0;
    return m;
  }
  Matrix& setZero() {
    Base::setZero();
    return *this;
  }
  Matrix& setZero(Index n) {
    resize(n);
    Base::setZero();
    return *this;
  }
  Matrix& setZero(Index r, Index c) {
    resize(r, c);
    Base::setZero();
    return *this;
  }
};

// ---- diagonal wrapper (rvalue only) ----------------------------------------------------------
class Diag {
 public:
  RawView v;
  explicit Diag(const RawView& x) : v(x) {}
  Index n() const { return v.r * v.c; }
  double d(Index i) const { return v.c == 1 ? v.at(i, 0) : v.at(0, i); }
  inline MatrixXd toDense() const;
};

// ---- member definitions that need View / MatrixXd --------------------------------------------
#define RTOC_ME_SUB(expr_p, nr_, nc_) \
  const RawView v = this->raw();      \
  return View(expr_p, nr_, nc_, v.rs, v.cs)
template <class D>
View MatrixBase<D>::block(Index i, Index j, Index nr, Index nc) const {
  const RawView v = raw();
  assert(i >= 0 && j >= 0 && i + nr <= v.r && j + nc <= v.c);
  return View(v.p + i * v.rs + j * v.cs, nr, nc, v.rs, v.cs);
}
template <class D>
template <int NR, int NC>
View MatrixBase<D>::block(Index i, Index j) const { return block(i, j, NR, NC); }
template <class D>
template <int NR, int NC>
View MatrixBase<D>::block(Index i, Index j, Index nr, Index nc) const { return block(i, j, nr, nc); }
template <class D>
View MatrixBase<D>::topRows(Index n) const { return block(0, 0, n, cols()); }
template <class D>
View MatrixBase<D>::bottomRows(Index n) const { return block(rows() - n, 0, n, cols()); }
template <class D>
View MatrixBase<D>::middleRows(Index i, Index n) const { return block(i, 0, n, cols()); }
template <class D>
View MatrixBase<D>::leftCols(Index n) const { return block(0, 0, rows(), n); }
template <class D>
View MatrixBase<D>::rightCols(Index n) const { return block(0, cols() - n, rows(), n); }
template <class D>
View MatrixBase<D>::middleCols(Index j, Index n) const { return block(0, j, rows(), n); }
template <class D>
template <int N>
View MatrixBase<D>::middleCols(Index j) const { return block(0, j, rows(), N); }
template <class D>
View MatrixBase<D>::topLeftCorner(Index nr, Index nc) const { return block(0, 0, nr, nc); }
template <class D>
View MatrixBase<D>::topRightCorner(Index nr, Index nc) const { return block(0, cols() - nc, nr, nc); }
template <class D>
View MatrixBase<D>::bottomLeftCorner(Index nr, Index nc) const { return block(rows() - nr, 0, nr, nc); }
template <class D>
View MatrixBase<D>::bottomRightCorner(Index nr, Index nc) const { return block(rows() - nr, cols() - nc, nr, nc); }
template <class D>
template <int NR, int NC>
View MatrixBase<D>::topLeftCorner() const { return topLeftCorner(NR, NC); }
template <class D>
template <int NR, int NC>
View MatrixBase<D>::topRightCorner() const { return topRightCorner(NR, NC); }
template <class D>
template <int NR, int NC>
View MatrixBase<D>::bottomLeftCorner() const { return bottomLeftCorner(NR, NC); }
template <class D>
template <int NR, int NC>
View MatrixBase<D>::bottomRightCorner() const { return bottomRightCorner(NR, NC); }
template <class D>
template <int N>
View MatrixBase<D>::topRows() const { return topRows(N); }
template <class D>
template <int N>
View MatrixBase<D>::bottomRows() const { return bottomRows(N); }
template <class D>
template <int N>
View MatrixBase<D>::leftCols() const { return leftCols(N); }
template <class D>
template <int N>
View MatrixBase<D>::rightCols() const { return rightCols(N); }
template <class D>
View MatrixBase<D>::row(Index i) const { return block(i, 0, 1, cols()); }
template <class D>
View MatrixBase<D>::col(Index j) const { return block(0, j, rows(), 1); }
template <class D>
View MatrixBase<D>::segment(Index i, Index n) const {
  const RawView v = raw();
  assert(v.r == 1 || v.c == 1);
  return v.c == 1 ? block(i, 0, n, 1) : block(0, i, 1, n);
}
template <class D>
View MatrixBase<D>::head(Index n) const { return segment(0, n); }
template <class D>
View MatrixBase<D>::tail(Index n) const { return segment(size() - n, n); }
template <class D>
template <int N>
View MatrixBase<D>::head() const { return head(N); }
template <class D>
template <int N>
View MatrixBase<D>::tail() const { return tail(N); }
template <class D>
template <int N>
View MatrixBase<D>::segment(Index i) const { return segment(i, N); }
template <class D>
View MatrixBase<D>::transpose() const {
  const RawView v = raw();
  return View(v.p, v.c, v.r, v.cs, v.rs);
}
template <class D>
View MatrixBase<D>::diagonal() const {
  const RawView v = raw();
  return View(v.p, std::min(v.r, v.c), 1, v.rs + v.cs, 0);
}
template <class D>
ArrayView MatrixBase<D>::array() const { return ArrayView(raw()); }
template <class D>
View MatrixBase<D>::matrix() const { return View(raw()); }
template <class D>
Diag MatrixBase<D>::asDiagonal() const { return Diag(raw()); }
template <class D>
MatrixXd MatrixBase<D>::eval() const { return MatrixXd(*this); }

inline MatrixXd Diag::toDense() const {
  MatrixXd m(n(), n());
  for (Index i = 0; i < n(); ++i) m(i, i) = d(i);
  return m;
}
inline View& View::operator=(const Diag& d) {
  const MatrixXd m = d.toDense();
  assign_view(v_, m.raw());
  return *this;
}
template <class S, int R, int C, int O, int MR, int MC>
Matrix<S, R, C, O, MR, MC>::Matrix(const Diag& d) : r_(0), c_(0) {
  *this = d.toDense();
}
template <class S, int R, int C, int O, int MR, int MC>
Matrix<S, R, C, O, MR, MC>& Matrix<S, R, C, O, MR, MC>::operator=(const Diag& d) {
  *this = d.toDense();
  return *this;
}

// ---- arithmetic (eager) -----------------------------------------------------------------------
inline MatrixXd eval_product(const RawView& a, const RawView& b) {
  assert(a.c == b.r);
  MatrixXd out(a.r, b.c);
  const RawView o = out.raw();
  for (Index j = 0; j < b.c; ++j)
    for (Index i = 0; i < a.r; ++i) {
      double s = 0.0;
      for (Index k = 0; k < a.c; ++k) s += a.at(i, k) * b.at(k, j);
      o.at(i, j) = s;
    }
  return out;
}
template <class A, class B>
inline MatrixXd operator*(const MatrixBase<A>& a, const MatrixBase<B>& b) {
  return eval_product(a.raw(), b.raw());
}
template <class A>
inline MatrixXd operator*(const MatrixBase<A>& a, const Diag& d) {
  const RawView v = a.raw();
  assert(v.c == d.n());
  MatrixXd out(v.r, v.c);
  for (Index j = 0; j < v.c; ++j)
    for (Index i = 0; i < v.r; ++i) out(i, j) = v.at(i, j) * d.d(j);
  return out;
}
template <class B>
inline MatrixXd operator*(const Diag& d, const MatrixBase<B>& b) {
  const RawView v = b.raw();
  assert(v.r == d.n());
  MatrixXd out(v.r, v.c);
  for (Index j = 0; j < v.c; ++j)
    for (Index i = 0; i < v.r; ++i) out(i, j) = d.d(i) * v.at(i, j);
  return out;
}
#define RTOC_ME_BINOP(op)                                                              \
  template <class A, class B>                                                          \
  inline MatrixXd operator op(const MatrixBase<A>& a, const MatrixBase<B>& b) {        \
    const RawView x = a.raw(), y = b.raw();                                            \
    assert(x.r == y.r && x.c == y.c);                                                  \
    MatrixXd out(x.r, x.c);                                                            \
    for (Index j = 0; j < x.c; ++j)                                                    \
      for (Index i = 0; i < x.r; ++i) out(i, j) = x.at(i, j) op y.at(i, j);            \
    return out;                                                                        \
  }
RTOC_ME_BINOP(+)
RTOC_ME_BINOP(-)
#undef RTOC_ME_BINOP
template <class A>
inline MatrixXd operator-(const MatrixBase<A>& a) {
  const RawView x = a.raw();
  MatrixXd out(x.r, x.c);
  for (Index j = 0; j < x.c; ++j)
    for (Index i = 0; i < x.r; ++i) out(i, j) = -x.at(i, j);
  return out;
}
template <class A>
inline MatrixXd scaled_(const MatrixBase<A>& a, double s) {
  const RawView x = a.raw();
  MatrixXd out(x.r, x.c);
  for (Index j = 0; j < x.c; ++j)
    for (Index i = 0; i < x.r; ++i) out(i, j) = s * x.at(i, j);
  return out;
}
template <class A>
inline MatrixXd operator*(const MatrixBase<A>& a, double s) { return scaled_(a, s); }
template <class A>
inline MatrixXd operator*(double s, const MatrixBase<A>& a) { return scaled_(a, s); }
template <class A>
inline MatrixXd operator*(const MatrixBase<A>& a, int s) { return scaled_(a, (double)s); }
template <class A>
inline MatrixXd operator*(int s, const MatrixBase<A>& a) { return scaled_(a, (double)s); }
template <class A>
inline MatrixXd operator/(const MatrixBase<A>& a, double s) {
  const RawView x = a.raw();
  MatrixXd out(x.r, x.c);
  for (Index j = 0; j < x.c; ++j)
    for (Index i = 0; i < x.r; ++i) out(i, j) = x.at(i, j) / s;
  return out;
}
template <class D>
template <class E>
MatrixXd MatrixBase<D>::cross(const MatrixBase<E>& o) const {
  const RawView a = raw(), b = o.raw();
  if (a.r * a.c != 3 || b.r * b.c != 3) {
    std::fprintf(stderr, "mini_eigen: cross needs 3-vectors\n");
    std::abort();
  }
  auto A = [&](Index k) { return a.c == 1 ? a.at(k, 0) : a.at(0, k); };
  auto B = [&](Index k) { return b.c == 1 ? b.at(k, 0) : b.at(0, k); };
  MatrixXd r(3, 1);
  r(0, 0) = A(1) * B(2) - A(2) * B(1);
  r(1, 0) = A(2) * B(0) - A(0) * B(2);
  r(2, 0) = A(0) * B(1) - A(1) * B(0);
  return r;
}
template <class D>
template <class E>
MatrixXd MatrixBase<D>::cwiseProduct(const MatrixBase<E>& o) const {
  const RawView x = raw(), y = o.raw();
  MatrixXd out(x.r, x.c);
  for (Index j = 0; j < x.c; ++j)
    for (Index i = 0; i < x.r; ++i) out(i, j) = x.at(i, j) * y.at(i, j);
  return out;
}
template <class D>
template <class E>
MatrixXd MatrixBase<D>::cwiseQuotient(const MatrixBase<E>& o) const {
  const RawView x = raw(), y = o.raw();
  MatrixXd out(x.r, x.c);
  for (Index j = 0; j < x.c; ++j)
    for (Index i = 0; i < x.r; ++i) out(i, j) = x.at(i, j) / y.at(i, j);
  return out;
}
template <class D>
MatrixXd MatrixBase<D>::cwiseAbs() const {
  const RawView x = raw();
  MatrixXd out(x.r, x.c);
  for (Index j = 0; j < x.c; ++j)
    for (Index i = 0; i < x.r; ++i) out(i, j) = std::fabs(x.at(i, j));
  return out;
}
template <class D>
MatrixXd MatrixBase<D>::cwiseInverse() const {
  const RawView x = raw();
  MatrixXd out(x.r, x.c);
  for (Index j = 0; j < x.c; ++j)
    for (Index i = 0; i < x.r; ++i) out(i, j) = 1.0 / x.at(i, j);
  return out;
}
template <class D>
MatrixXd MatrixBase<D>::cwiseSqrt() const {
  const RawView x = raw();
  MatrixXd out(x.r, x.c);
  for (Index j = 0; j < x.c; ++j)
    for (Index i = 0; i < x.r; ++i) out(i, j) = std::sqrt(x.at(i, j));
  return out;
}
// general inverse by Gauss-Jordan with partial pivoting (used by reference TESTS' closed forms only)
template <class D>
MatrixXd MatrixBase<D>::inverse() const {
  const Index n = rows();
  assert(n == cols());
  MatrixXd a(*this), inv = MatrixXd::Identity(n, n);
  for (Index k = 0; k < n; ++k) {
    Index p = k;
    for (Index i = k + 1; i < n; ++i)
      if (std::fabs(a(i, k)) > std::fabs(a(p, k))) p = i;
    if (p != k)
      for (Index j = 0; j < n; ++j) {
        std::swap(a(k, j), a(p, j));
        std::swap(inv(k, j), inv(p, j));
      }
    const double piv = 1.0 / a(k, k);
    for (Index j = 0; j < n; ++j) {
      a(k, j) *= piv;
      inv(k, j) *= piv;
    }
    for (Index i = 0; i < n; ++i) {
      if (i == k) continue;
      const double f = a(i, k);
      if (f == 0.0) continue;
      for (Index j = 0; j < n; ++j) {
        a(i, j) -= f * a(k, j);
        inv(i, j) -= f * inv(k, j);
      }
    }
  }
  return inv;
}

template <class D>
inline std::ostream& operator<<(std::ostream& os, const MatrixBase<D>& m) {
  const RawView v = m.raw();
  for (Index i = 0; i < v.r; ++i) {
    for (Index j = 0; j < v.c; ++j) os << (j ? " " : "") << v.at(i, j);
    if (i + 1 < v.r) os << "\n";
  }
  return os;
}
inline std::ostream& operator<<(std::ostream& os, const Diag& d) { return os << d.toDense(); }

// ---- Cholesky (Eigen::LLT: A = L L^T, lower, unpivoted) ---------------------------------------
template <class M, int UpLo = Lower>
class LLT {
  MatrixXd l_;
  ComputationInfo info_;

 public:
  LLT() : info_(Success) {}
  explicit LLT(Index n) : l_(n, n), info_(Success) {}
  template <class D>
  explicit LLT(const MatrixBase<D>& a) : info_(Success) {
    compute(a);
  }
  template <class D>
  LLT& compute(const MatrixBase<D>& a) {
    const Index n = a.rows();
    assert(n == a.cols());
    l_ = a;
    info_ = Success;
    for (Index j = 0; j < n; ++j) {
      double d = l_(j, j);
      for (Index k = 0; k < j; ++k) d -= l_(j, k) * l_(j, k);
      if (!(d > 0.0)) {
        info_ = NumericalIssue;
        d = std::fabs(d) > 0 ? std::fabs(d) : 1.0;
      }
      const double ljj = std::sqrt(d);
      l_(j, j) = ljj;
      for (Index i = j + 1; i < n; ++i) {
        double s = l_(i, j);
        for (Index k = 0; k < j; ++k) s -= l_(i, k) * l_(j, k);
        l_(i, j) = s / ljj;
      }
      for (Index i = 0; i < j; ++i) l_(i, j) = 0.0;
    }
    return *this;
  }
  ComputationInfo info() const { return info_; }
  const MatrixXd& matrixLLT() const { return l_; }
  MatrixXd matrixL() const { return l_; }
  MatrixXd matrixU() const { return MatrixXd(l_.transpose()); }
  template <class D>
  MatrixXd solve(const MatrixBase<D>& b) const {
    MatrixXd x(b);
    const Index n = l_.rows(), m = x.cols();
    assert(x.rows() == n);
    for (Index c = 0; c < m; ++c) {
      for (Index i = 0; i < n; ++i) {
        double s = x(i, c);
        for (Index k = 0; k < i; ++k) s -= l_(i, k) * x(k, c);
        x(i, c) = s / l_(i, i);
      }
      for (Index i = n - 1; i >= 0; --i) {
        double s = x(i, c);
        for (Index k = i + 1; k < n; ++k) s -= l_(k, i) * x(k, c);
        x(i, c) = s / l_(i, i);
      }
    }
    return x;
  }
  template <class D>
  void solveInPlace(const MatrixBase<D>& b) const {
    const MatrixXd x = solve(b);
    assign_view(b.raw(), x.raw());
  }
};

template <>
class Map<const VectorXd> : public VectorXd {
 public:
  Map(const double* p, Index n) : VectorXd(n) {
    for (Index i = 0; i < n; ++i) (*this)(i) = p[i];
  }
};

}  // namespace Eigen
#endif  // RTOC_MINI_EIGEN_HPP_
