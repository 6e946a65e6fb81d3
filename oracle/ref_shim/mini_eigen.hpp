// mini_eigen.hpp — a small stand-in for the part of Eigen3 that the reference's solver sources use.
//
// TEST INFRASTRUCTURE ONLY (see oracle/README in the Makefile header).  Eigen is not installed in
// this image and there is no network, so the reference's own translation units
// (src/LaseCamCalCeres.cpp, src/pose_local_parameterization.cpp, src/utilities.cpp) cannot be
// compiled against the real library.  This header implements — eagerly, with value semantics and
// no expression templates — exactly the API surface those three files touch, so that they can be
// compiled FROM WHERE THEY LIE and run, and the oracle's restatement of their arithmetic can be
// checked against the reference's own code (oracle/Makefile target `ref`, tests/test_ref_pin.py).
//
// It is not Eigen: conversions (quaternion <-> matrix), inverse(), JacobiSVD, LDLT and EigenSolver
// use textbook algorithms, so results agree with the real library to rounding, not bitwise.
// Written from the public Eigen API documentation; no Eigen source was available or used.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <complex>
#include <cstddef>
#include <iostream>
#include <type_traits>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

namespace Eigen {

typedef std::ptrdiff_t Index;
const int Dynamic = -1;
enum StorageOptions { ColMajor = 0, RowMajor = 1 };
enum DecompositionOptions { ComputeFullU = 0x04, ComputeThinU = 0x08, ComputeFullV = 0x10, ComputeThinV = 0x20 };

template <typename T, int R, int C, int Opt = ColMajor> class Matrix;
template <typename Derived> class MatrixBase;
template <typename Xpr, int R, int C> class Block;
template <typename Plain> class Map;
template <typename T> class Quaternion;
template <typename D, typename T> class QuaternionBase;
template <typename M> class LDLT;

template <typename D> struct traits;
template <typename T, int R, int C, int O>
struct traits<Matrix<T, R, C, O>> {
  typedef T Scalar;
  enum { Rows = R, Cols = C, Options = O };
};
template <typename X, int R, int C>
struct traits<Block<X, R, C>> {
  typedef typename traits<typename std::remove_const<X>::type>::Scalar Scalar;
  enum { Rows = R, Cols = C, Options = 0 };
};
template <typename P>
struct traits<Map<P>> : traits<typename std::remove_const<P>::type> {};

template <typename S> struct real_of { typedef S type; };
template <typename S> struct real_of<std::complex<S>> { typedef S type; };

namespace internal {
struct Sized {};  // tag of the (rows, cols) constructor
constexpr int pick(int a, int b) { return a != Dynamic ? a : b; }
inline double real_part(double v) { return v; }
inline double real_part(const std::complex<double>& v) { return v.real(); }
}  // namespace internal

// ------------------------------------------------------------------------------------------
template <typename D>
class CommaInitializer {
 public:
  typedef typename traits<D>::Scalar Scalar;
  explicit CommaInitializer(D* t) : t_(t), row_(0), col_(0), blockrows_(1) {}
  template <typename S, typename = typename std::enable_if<std::is_arithmetic<S>::value>::type>
  CommaInitializer& operator,(const S& s) {
    put_scalar(static_cast<Scalar>(s));
    return *this;
  }
  template <typename OD>
  CommaInitializer& operator,(const MatrixBase<OD>& m) {
    put_block(m);
    return *this;
  }
  void put_scalar(Scalar s) {
    wrap(1);
    t_->coeffRef(row_, col_) = s;
    col_ += 1;
  }
  template <typename OD>
  void put_block(const MatrixBase<OD>& m) {
    wrap(m.rows());
    for (Index i = 0; i < m.rows(); ++i)
      for (Index j = 0; j < m.cols(); ++j) t_->coeffRef(row_ + i, col_ + j) = m.coeff(i, j);
    col_ += m.cols();
  }

 private:
  void wrap(Index next_rows) {
    if (col_ == t_->cols()) {
      row_ += blockrows_;
      col_ = 0;
    }
    if (col_ == 0) blockrows_ = next_rows;
    assert(row_ < t_->rows() && col_ < t_->cols());
  }
  D* t_;
  Index row_, col_, blockrows_;
};

// 1x1 -> scalar conversion (inner products).  A plain (non-template) conversion function, present
// only for compile-time 1x1 types: built-in operators (double + M11, d += M11) only look at those.
template <typename D, bool Is11>
struct ScalarConversion {};
template <typename D>
struct ScalarConversion<D, true> {
  operator typename traits<D>::Scalar() const { return static_cast<const D*>(this)->coeff(0, 0); }
};

// ------------------------------------------------------------------------------------------
template <typename D>
class MatrixBase : public ScalarConversion<D, traits<D>::Rows == 1 && traits<D>::Cols == 1> {
 public:
  typedef typename traits<D>::Scalar Scalar;
  typedef typename real_of<Scalar>::type RealScalar;
  enum { Rows = traits<D>::Rows, Cols = traits<D>::Cols };
  typedef Matrix<Scalar, Rows, Cols> PlainObject;

  D& derived() { return *static_cast<D*>(this); }
  const D& derived() const { return *static_cast<const D*>(this); }
  Index rows() const { return derived().rows(); }
  Index cols() const { return derived().cols(); }
  Index size() const { return rows() * cols(); }
  Scalar coeff(Index i, Index j) const { return derived().coeff(i, j); }
  Scalar& coeffRef(Index i, Index j) { return derived().coeffRef(i, j); }

  Scalar operator()(Index i, Index j) const { return coeff(i, j); }
  Scalar& operator()(Index i, Index j) { return coeffRef(i, j); }
  // linear access for vectors
  Scalar operator()(Index i) const { return cols() == 1 ? coeff(i, 0) : coeff(0, i); }
  Scalar& operator()(Index i) { return cols() == 1 ? coeffRef(i, 0) : coeffRef(0, i); }
  Scalar operator[](Index i) const { return (*this)(i); }
  Scalar& operator[](Index i) { return (*this)(i); }
  Scalar x() const { return (*this)(0); }
  Scalar y() const { return (*this)(1); }
  Scalar z() const { return (*this)(2); }
  Scalar w() const { return (*this)(3); }
  Scalar& x() { return (*this)(0); }
  Scalar& y() { return (*this)(1); }
  Scalar& z() { return (*this)(2); }
  Scalar& w() { return (*this)(3); }

  PlainObject eval() const { return PlainObject(*this); }

  Matrix<Scalar, Cols, Rows> transpose() const {
    Matrix<Scalar, Cols, Rows> r(cols(), rows(), internal::Sized());
    for (Index i = 0; i < rows(); ++i)
      for (Index j = 0; j < cols(); ++j) r.coeffRef(j, i) = coeff(i, j);
    return r;
  }

  template <typename OD>
  Matrix<Scalar, Rows, traits<OD>::Cols> operator*(const MatrixBase<OD>& o) const {
    assert(cols() == o.rows());
    Matrix<Scalar, Rows, traits<OD>::Cols> r(rows(), o.cols(), internal::Sized());
    for (Index i = 0; i < rows(); ++i)
      for (Index j = 0; j < o.cols(); ++j) {
        Scalar s = Scalar(0);
        for (Index k = 0; k < cols(); ++k) s += coeff(i, k) * o.coeff(k, j);
        r.coeffRef(i, j) = s;
      }
    return r;
  }
  template <typename OD>
  Matrix<Scalar, internal::pick(Rows, traits<OD>::Rows), internal::pick(Cols, traits<OD>::Cols)> operator+(
      const MatrixBase<OD>& o) const {
    assert(rows() == o.rows() && cols() == o.cols());
    Matrix<Scalar, internal::pick(Rows, traits<OD>::Rows), internal::pick(Cols, traits<OD>::Cols)> r(rows(), cols(), internal::Sized());
    for (Index i = 0; i < rows(); ++i)
      for (Index j = 0; j < cols(); ++j) r.coeffRef(i, j) = coeff(i, j) + o.coeff(i, j);
    return r;
  }
  template <typename OD>
  Matrix<Scalar, internal::pick(Rows, traits<OD>::Rows), internal::pick(Cols, traits<OD>::Cols)> operator-(
      const MatrixBase<OD>& o) const {
    assert(rows() == o.rows() && cols() == o.cols());
    Matrix<Scalar, internal::pick(Rows, traits<OD>::Rows), internal::pick(Cols, traits<OD>::Cols)> r(rows(), cols(), internal::Sized());
    for (Index i = 0; i < rows(); ++i)
      for (Index j = 0; j < cols(); ++j) r.coeffRef(i, j) = coeff(i, j) - o.coeff(i, j);
    return r;
  }
  PlainObject operator-() const {
    PlainObject r(rows(), cols(), internal::Sized());
    for (Index i = 0; i < rows(); ++i)
      for (Index j = 0; j < cols(); ++j) r.coeffRef(i, j) = -coeff(i, j);
    return r;
  }
  PlainObject operator*(const Scalar& s) const {
    PlainObject r(rows(), cols(), internal::Sized());
    for (Index i = 0; i < rows(); ++i)
      for (Index j = 0; j < cols(); ++j) r.coeffRef(i, j) = coeff(i, j) * s;
    return r;
  }
  PlainObject operator/(const Scalar& s) const {
    PlainObject r(rows(), cols(), internal::Sized());
    for (Index i = 0; i < rows(); ++i)
      for (Index j = 0; j < cols(); ++j) r.coeffRef(i, j) = coeff(i, j) / s;
    return r;
  }
  friend PlainObject operator*(const Scalar& s, const MatrixBase& m) {
    PlainObject r(m.rows(), m.cols(), internal::Sized());
    for (Index i = 0; i < m.rows(); ++i)
      for (Index j = 0; j < m.cols(); ++j) r.coeffRef(i, j) = s * m.coeff(i, j);
    return r;
  }
  template <typename OD>
  D& operator+=(const MatrixBase<OD>& o) {
    assert(rows() == o.rows() && cols() == o.cols());
    for (Index i = 0; i < rows(); ++i)
      for (Index j = 0; j < cols(); ++j) coeffRef(i, j) += o.coeff(i, j);
    return derived();
  }
  template <typename OD>
  D& operator-=(const MatrixBase<OD>& o) {
    assert(rows() == o.rows() && cols() == o.cols());
    for (Index i = 0; i < rows(); ++i)
      for (Index j = 0; j < cols(); ++j) coeffRef(i, j) -= o.coeff(i, j);
    return derived();
  }
  D& operator*=(const Scalar& s) {
    for (Index i = 0; i < rows(); ++i)
      for (Index j = 0; j < cols(); ++j) coeffRef(i, j) *= s;
    return derived();
  }
  D& operator/=(const Scalar& s) {
    for (Index i = 0; i < rows(); ++i)
      for (Index j = 0; j < cols(); ++j) coeffRef(i, j) /= s;
    return derived();
  }

  template <typename OD>
  Scalar dot(const MatrixBase<OD>& o) const {
    assert(size() == o.size());
    Scalar s = Scalar(0);
    for (Index i = 0; i < size(); ++i) s += (*this)(i) * o(i);
    return s;
  }
  template <typename OD>
  Matrix<Scalar, 3, 1> cross(const MatrixBase<OD>& o) const {
    assert(size() == 3 && o.size() == 3);
    const Scalar a0 = (*this)(0), a1 = (*this)(1), a2 = (*this)(2), b0 = o(0), b1 = o(1), b2 = o(2);
    return Matrix<Scalar, 3, 1>(a1 * b2 - a2 * b1, a2 * b0 - a0 * b2, a0 * b1 - a1 * b0);
  }
  RealScalar squaredNorm() const {
    RealScalar s = 0;
    for (Index i = 0; i < rows(); ++i)
      for (Index j = 0; j < cols(); ++j) s += std::norm(coeff(i, j));
    return s;
  }
  RealScalar norm() const { return std::sqrt(squaredNorm()); }
  PlainObject normalized() const { return (*this) / norm(); }
  void normalize() { (*this) /= norm(); }

  D& setZero() {
    for (Index i = 0; i < rows(); ++i)
      for (Index j = 0; j < cols(); ++j) coeffRef(i, j) = Scalar(0);
    return derived();
  }
  D& setIdentity() {
    for (Index i = 0; i < rows(); ++i)
      for (Index j = 0; j < cols(); ++j) coeffRef(i, j) = (i == j) ? Scalar(1) : Scalar(0);
    return derived();
  }

  // ---- blocks ----
  Block<D, Dynamic, Dynamic> block(Index i, Index j, Index r, Index c) { return Block<D, Dynamic, Dynamic>(derived(), i, j, r, c); }
  Block<const D, Dynamic, Dynamic> block(Index i, Index j, Index r, Index c) const {
    return Block<const D, Dynamic, Dynamic>(derived(), i, j, r, c);
  }
  template <int R, int C>
  Block<D, R, C> block(Index i, Index j) { return Block<D, R, C>(derived(), i, j, R, C); }
  template <int R, int C>
  Block<const D, R, C> block(Index i, Index j) const { return Block<const D, R, C>(derived(), i, j, R, C); }
  Block<D, Rows, 1> col(Index j) { return Block<D, Rows, 1>(derived(), 0, j, rows(), 1); }
  Block<const D, Rows, 1> col(Index j) const { return Block<const D, Rows, 1>(derived(), 0, j, rows(), 1); }
  Block<D, 1, Cols> row(Index i) { return Block<D, 1, Cols>(derived(), i, 0, 1, cols()); }
  Block<const D, 1, Cols> row(Index i) const { return Block<const D, 1, Cols>(derived(), i, 0, 1, cols()); }
  template <int N>
  Block<D, Rows, N> leftCols() { return Block<D, Rows, N>(derived(), 0, 0, rows(), N); }
  template <int N>
  Block<const D, Rows, N> leftCols() const { return Block<const D, Rows, N>(derived(), 0, 0, rows(), N); }
  template <int N>
  Block<D, Rows, N> rightCols() { return Block<D, Rows, N>(derived(), 0, cols() - N, rows(), N); }
  template <int N>
  Block<const D, Rows, N> rightCols() const { return Block<const D, Rows, N>(derived(), 0, cols() - N, rows(), N); }
  Block<D, Rows, Dynamic> leftCols(Index n) { return Block<D, Rows, Dynamic>(derived(), 0, 0, rows(), n); }
  Block<const D, Rows, Dynamic> leftCols(Index n) const { return Block<const D, Rows, Dynamic>(derived(), 0, 0, rows(), n); }
  Block<D, Rows, Dynamic> rightCols(Index n) { return Block<D, Rows, Dynamic>(derived(), 0, cols() - n, rows(), n); }
  Block<const D, Rows, Dynamic> rightCols(Index n) const {
    return Block<const D, Rows, Dynamic>(derived(), 0, cols() - n, rows(), n);
  }
  template <int N>
  Block<D, N, Cols> topRows() { return Block<D, N, Cols>(derived(), 0, 0, N, cols()); }
  template <int N>
  Block<const D, N, Cols> topRows() const { return Block<const D, N, Cols>(derived(), 0, 0, N, cols()); }
  template <int N>
  Block<D, N, Cols> bottomRows() { return Block<D, N, Cols>(derived(), rows() - N, 0, N, cols()); }
  template <int N>
  Block<const D, N, Cols> bottomRows() const { return Block<const D, N, Cols>(derived(), rows() - N, 0, N, cols()); }
  // sub-vectors (column vectors; row vectors when Rows == 1 at compile time)
  Block<D, (Rows == 1 ? 1 : Dynamic), (Rows == 1 ? Dynamic : 1)> segment(Index i, Index n) {
    typedef Block<D, (Rows == 1 ? 1 : Dynamic), (Rows == 1 ? Dynamic : 1)> B;
    return Rows == 1 ? B(derived(), 0, i, 1, n) : B(derived(), i, 0, n, 1);
  }
  Block<const D, (Rows == 1 ? 1 : Dynamic), (Rows == 1 ? Dynamic : 1)> segment(Index i, Index n) const {
    typedef Block<const D, (Rows == 1 ? 1 : Dynamic), (Rows == 1 ? Dynamic : 1)> B;
    return Rows == 1 ? B(derived(), 0, i, 1, n) : B(derived(), i, 0, n, 1);
  }
  template <int N>
  Block<D, (Rows == 1 ? 1 : N), (Rows == 1 ? N : 1)> segment(Index i) {
    typedef Block<D, (Rows == 1 ? 1 : N), (Rows == 1 ? N : 1)> B;
    return Rows == 1 ? B(derived(), 0, i, 1, N) : B(derived(), i, 0, N, 1);
  }
  template <int N>
  Block<const D, (Rows == 1 ? 1 : N), (Rows == 1 ? N : 1)> segment(Index i) const {
    typedef Block<const D, (Rows == 1 ? 1 : N), (Rows == 1 ? N : 1)> B;
    return Rows == 1 ? B(derived(), 0, i, 1, N) : B(derived(), i, 0, N, 1);
  }
  auto head(Index n) -> decltype(this->segment(0, n)) { return segment(0, n); }
  auto head(Index n) const -> decltype(this->segment(0, n)) { return segment(0, n); }
  template <int N>
  auto head() -> decltype(this->template segment<N>(0)) { return this->template segment<N>(0); }
  template <int N>
  auto head() const -> decltype(this->template segment<N>(0)) { return this->template segment<N>(0); }
  auto tail(Index n) -> decltype(this->segment(0, n)) { return segment(size() - n, n); }
  auto tail(Index n) const -> decltype(this->segment(0, n)) { return segment(size() - n, n); }

  // ---- misc ----
  Matrix<RealScalar, Rows, Cols> real() const {
    Matrix<RealScalar, Rows, Cols> r(rows(), cols(), internal::Sized());
    for (Index i = 0; i < rows(); ++i)
      for (Index j = 0; j < cols(); ++j) r.coeffRef(i, j) = internal::real_part(coeff(i, j));
    return r;
  }
  template <typename I>
  Scalar maxCoeff(I* index) const {
    Index best = 0;
    for (Index i = 1; i < size(); ++i)
      if ((*this)(i) > (*this)(best)) best = i;
    if (index) *index = static_cast<I>(best);
    return (*this)(best);
  }
  Scalar maxCoeff() const { return maxCoeff(static_cast<Index*>(nullptr)); }

  PlainObject inverse() const;         // Gauss-Jordan with partial pivoting
  LDLT<Matrix<Scalar, Dynamic, Dynamic>> ldlt() const;

  CommaInitializer<D> operator<<(const Scalar& s) {
    CommaInitializer<D> ci(&derived());
    ci.put_scalar(s);
    return ci;
  }
  template <typename OD>
  CommaInitializer<D> operator<<(const MatrixBase<OD>& m) {
    CommaInitializer<D> ci(&derived());
    ci.put_block(m);
    return ci;
  }

 protected:
  // generic assignment helper: same shape, or vector <- transposed vector (as Eigen allows)
  template <typename OD>
  void assign_from(const MatrixBase<OD>& o) {
    if (rows() == o.rows() && cols() == o.cols()) {
      for (Index i = 0; i < rows(); ++i)
        for (Index j = 0; j < cols(); ++j) coeffRef(i, j) = o.coeff(i, j);
    } else {
      assert((rows() == 1 || cols() == 1) && (o.rows() == 1 || o.cols() == 1) && size() == o.size());
      for (Index i = 0; i < size(); ++i) (*this)(i) = o(i);
    }
  }
};

template <typename D>
std::ostream& operator<<(std::ostream& os, const MatrixBase<D>& m) {
  for (Index i = 0; i < m.rows(); ++i) {
    for (Index j = 0; j < m.cols(); ++j) os << (j ? " " : "") << m.coeff(i, j);
    if (i + 1 < m.rows()) os << "\n";
  }
  return os;
}

// ------------------------------------------------------------------------------------------
template <typename T, int R, int C, int Opt>
class Matrix : public MatrixBase<Matrix<T, R, C, Opt>> {
  typedef MatrixBase<Matrix<T, R, C, Opt>> Base;

 public:
  typedef T Scalar;
  Matrix() : r_(R == Dynamic ? 0 : R), c_(C == Dynamic ? 0 : C), d_((size_t)(r_ * c_), T(0)) {}
  // internal sized constructor (the tag keeps it apart from the coefficient constructors)
  Matrix(Index r, Index c, internal::Sized) : r_(r), c_(c), d_((size_t)(r * c), T(0)) {
    assert((R == Dynamic || R == r) && (C == Dynamic || C == c));
  }
  // VectorXd(n) / RowVectorXd(n); for a fixed 1x1 matrix the argument is the coefficient
  explicit Matrix(Index n) {
    if (R != Dynamic && C != Dynamic) { r_ = R; c_ = C; }
    else if (R == Dynamic && C != Dynamic) { r_ = n; c_ = C; }
    else if (R != Dynamic && C == Dynamic) { r_ = R; c_ = n; }
    else { r_ = n; c_ = 1; }
    d_.assign((size_t)(r_ * c_), T(0));
    if (R == 1 && C == 1) d_[0] = T(n);
  }
  // (x, y) for fixed 2-vectors, (rows, cols) for dynamic matrices
  template <typename A, typename B, typename = typename std::enable_if<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value>::type>
  Matrix(const A& a, const B& b) {
    if (R == Dynamic || C == Dynamic) {
      r_ = (Index)a; c_ = (Index)b;
      assert((R == Dynamic || R == r_) && (C == Dynamic || C == c_));
      d_.assign((size_t)(r_ * c_), T(0));
    } else {
      assert(R * C == 2);
      r_ = R; c_ = C;
      d_.assign(2, T(0));
      d_[0] = T(a); d_[1] = T(b);
    }
  }
  Matrix(const T& x, const T& y, const T& z) : r_(R), c_(C), d_(3) { assert(R * C == 3); d_[0] = x; d_[1] = y; d_[2] = z; }
  Matrix(const T& x, const T& y, const T& z, const T& w) : r_(R), c_(C), d_(4) { assert(R * C == 4); d_[0] = x; d_[1] = y; d_[2] = z; d_[3] = w; }
  Matrix(const Matrix& o) = default;
  template <typename OD>
  Matrix(const MatrixBase<OD>& o) { init_from(o); }
  Matrix& operator=(const Matrix& o) = default;
  template <typename OD>
  Matrix& operator=(const MatrixBase<OD>& o) {
    if (static_cast<const void*>(&o) == static_cast<const void*>(this)) return *this;
    Matrix tmp;
    tmp.init_from(o);
    r_ = tmp.r_; c_ = tmp.c_; d_.swap(tmp.d_);
    return *this;
  }

  Index rows() const { return r_; }
  Index cols() const { return c_; }
  T coeff(Index i, Index j) const { return d_[idx(i, j)]; }
  T& coeffRef(Index i, Index j) { return d_[idx(i, j)]; }
  T* data() { return d_.data(); }
  const T* data() const { return d_.data(); }
  void resize(Index r, Index c) { r_ = r; c_ = c; d_.assign((size_t)(r * c), T(0)); }
  void resize(Index n) { if (C == 1 || (R == Dynamic && C == Dynamic)) resize(n, 1); else resize(1, n); }

  static Matrix Zero() { return Matrix(); }
  static Matrix Zero(Index n) { Matrix m(n); return m; }
  static Matrix Zero(Index r, Index c) { return Matrix(r, c, internal::Sized()); }
  static Matrix Identity() { Matrix m; m.setIdentity(); return m; }
  static Matrix Identity(Index r, Index c) { Matrix m(r, c, internal::Sized()); m.setIdentity(); return m; }
  static Matrix Ones() { Matrix m; for (auto& v : m.d_) v = T(1); return m; }
  static Matrix Unit(Index k) { Matrix m; m.d_[(size_t)k] = T(1); return m; }
  static Matrix UnitX() { return Unit(0); }
  static Matrix UnitY() { return Unit(1); }
  static Matrix UnitZ() { return Unit(2); }
  // rotation (quaternion / angle-axis product) -> 3x3 matrix, as Eigen's RotationBase assignment
  template <typename QD>
  Matrix(const QuaternionBase<QD, T>& q) { init_from(q.toRotationMatrix()); }
  template <typename QD>
  Matrix& operator=(const QuaternionBase<QD, T>& q) { return *this = q.toRotationMatrix(); }

 private:
  template <typename OD>
  void init_from(const MatrixBase<OD>& o) {
    Index r = o.rows(), c = o.cols();
    // vector <- transposed vector
    if ((R != Dynamic && R != r) || (C != Dynamic && C != c)) {
      assert((r == 1 || c == 1) && (R == 1 || C == 1));
      std::swap(r, c);
      assert((R == Dynamic || R == r) && (C == Dynamic || C == c));
      r_ = r; c_ = c;
      d_.assign((size_t)(r * c), T(0));
      for (Index i = 0; i < r * c; ++i) d_[(size_t)i] = o(i);
      return;
    }
    r_ = r; c_ = c;
    d_.assign((size_t)(r * c), T(0));
    for (Index i = 0; i < r; ++i)
      for (Index j = 0; j < c; ++j) d_[idx(i, j)] = o.coeff(i, j);
  }
  size_t idx(Index i, Index j) const {
    assert(i >= 0 && i < r_ && j >= 0 && j < c_);
    return (size_t)((Opt & RowMajor) ? i * c_ + j : j * r_ + i);
  }
  Index r_, c_;
  std::vector<T> d_;
};

typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<double, Dynamic, 1> VectorXd;
typedef Matrix<double, 2, 2> Matrix2d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<double, Dynamic, Dynamic> MatrixXd;

// ------------------------------------------------------------------------------------------
template <typename Xpr, int R, int C>
class Block : public MatrixBase<Block<Xpr, R, C>> {
  typedef MatrixBase<Block<Xpr, R, C>> Base;

 public:
  typedef typename traits<Block>::Scalar Scalar;
  Block(Xpr& x, Index i, Index j, Index r, Index c) : x_(x), i_(i), j_(j), r_(r), c_(c) {
    assert(i >= 0 && j >= 0 && i + r <= x.rows() && j + c <= x.cols());
  }
  Block(const Block&) = default;
  Index rows() const { return r_; }
  Index cols() const { return c_; }
  Scalar coeff(Index i, Index j) const { return x_.coeff(i_ + i, j_ + j); }
  Scalar& coeffRef(Index i, Index j) { return x_.coeffRef(i_ + i, j_ + j); }
  template <typename OD>
  Block& operator=(const MatrixBase<OD>& o) {
    typename MatrixBase<OD>::PlainObject tmp(o);  // rhs may alias the parent
    this->assign_from(tmp);
    return *this;
  }
  Block& operator=(const Block& o) {
    typename Base::PlainObject tmp(o);
    this->assign_from(tmp);
    return *this;
  }

 private:
  Xpr& x_;
  Index i_, j_, r_, c_;
};

// ------------------------------------------------------------------------------------------
template <typename Plain>
class Map : public MatrixBase<Map<Plain>> {
  typedef typename std::remove_const<Plain>::type P;
  typedef MatrixBase<Map<Plain>> Base;
  typedef typename traits<P>::Scalar S;
  typedef typename std::conditional<std::is_const<Plain>::value, const S, S>::type Elem;

 public:
  typedef S Scalar;
  explicit Map(Elem* p) : p_(p), r_(traits<P>::Rows), c_(traits<P>::Cols) {
    static_assert(traits<P>::Rows != Dynamic && traits<P>::Cols != Dynamic, "dynamic Map needs sizes");
  }
  Map(Elem* p, Index r, Index c) : p_(p), r_(r), c_(c) {}
  Map(Elem* p, Index n) : p_(p), r_(traits<P>::Cols == 1 ? n : 1), c_(traits<P>::Cols == 1 ? 1 : n) {}
  Map(const Map&) = default;
  Index rows() const { return r_; }
  Index cols() const { return c_; }
  S coeff(Index i, Index j) const { return p_[idx(i, j)]; }
  S& coeffRef(Index i, Index j) { return const_cast<S*>(p_)[idx(i, j)]; }
  template <typename OD>
  Map& operator=(const MatrixBase<OD>& o) {
    typename MatrixBase<OD>::PlainObject tmp(o);
    this->assign_from(tmp);
    return *this;
  }
  Map& operator=(const Map& o) {
    typename Base::PlainObject tmp(o);
    this->assign_from(tmp);
    return *this;
  }

 private:
  size_t idx(Index i, Index j) const {
    assert(i >= 0 && i < r_ && j >= 0 && j < c_);
    return (size_t)((traits<P>::Options & RowMajor) ? i * c_ + j : j * r_ + i);
  }
  Elem* p_;
  Index r_, c_;
};

// ------------------------------------------------------------------------------------------
// inverse: Gauss-Jordan elimination with partial pivoting
template <typename D>
typename MatrixBase<D>::PlainObject MatrixBase<D>::inverse() const {
  const Index n = rows();
  assert(n == cols());
  Matrix<Scalar, Dynamic, Dynamic> a(*this), inv = Matrix<Scalar, Dynamic, Dynamic>::Identity(n, n);
  for (Index k = 0; k < n; ++k) {
    Index piv = k;
    for (Index i = k + 1; i < n; ++i)
      if (std::abs(a(i, k)) > std::abs(a(piv, k))) piv = i;
    if (piv != k)
      for (Index j = 0; j < n; ++j) {
        std::swap(a(k, j), a(piv, j));
        std::swap(inv(k, j), inv(piv, j));
      }
    const Scalar p = a(k, k);
    for (Index j = 0; j < n; ++j) {
      a(k, j) /= p;
      inv(k, j) /= p;
    }
    for (Index i = 0; i < n; ++i) {
      if (i == k) continue;
      const Scalar f = a(i, k);
      if (f == Scalar(0)) continue;
      for (Index j = 0; j < n; ++j) {
        a(i, j) -= f * a(k, j);
        inv(i, j) -= f * inv(k, j);
      }
    }
  }
  return PlainObject(inv);
}

// LDL^T with symmetric pivoting on the largest remaining diagonal entry and a pseudo-inverse of D in solve() — what
// Eigen documents for LDLT ("robust Cholesky decomposition of a matrix with pivoting"; positive or negative
// SEMI-definite input allowed): the reference calls it on a normal matrix it has just reported as unobservable
// (LaseCamCalCeres.cpp:173-181), so the semi-definite case must not divide by zero.
// Left-looking like Eigen's unblocked kernel (LDLT.h, ldlt_inplace<Lower>::unblocked): the pivot search runs over
// diagonal entries that have not been updated yet — the original diagonal of the remaining rows — not over the Schur
// complement's diagonal.  Stand-in written from the published source; not checked against an Eigen build.
template <typename M>
class LDLT {
 public:
  template <typename OD>
  explicit LDLT(const MatrixBase<OD>& A) : n_(A.rows()), L_(A.rows(), A.rows(), internal::Sized()), d_(A.rows()), p_((size_t)A.rows()) {
    for (Index i = 0; i < n_; ++i)
      for (Index j = 0; j < n_; ++j) L_(i, j) = A.coeff(i, j);
    for (Index i = 0; i < n_; ++i) p_[(size_t)i] = i;
    for (Index k = 0; k < n_; ++k) {
      Index best = k;
      for (Index i = k + 1; i < n_; ++i)
        if (std::abs(L_(i, i)) > std::abs(L_(best, best))) best = i;
      if (best != k) {
        for (Index j = 0; j < n_; ++j) std::swap(L_(k, j), L_(best, j));
        for (Index i = 0; i < n_; ++i) std::swap(L_(i, k), L_(i, best));
        std::swap(p_[(size_t)k], p_[(size_t)best]);
      }
      double d = L_(k, k);
      for (Index j = 0; j < k; ++j) d -= L_(k, j) * L_(k, j) * d_(j);
      d_(k) = d;
      for (Index i = k + 1; i < n_; ++i) {
        double s = L_(i, k);
        for (Index j = 0; j < k; ++j) s -= L_(i, j) * L_(k, j) * d_(j);
        L_(i, k) = d != 0.0 ? s / d : 0.0;
      }
    }
  }
  template <typename OD>
  Matrix<double, Dynamic, traits<OD>::Cols> solve(const MatrixBase<OD>& b) const {
    Matrix<double, Dynamic, traits<OD>::Cols> x(b);
    for (Index c = 0; c < x.cols(); ++c) {
      std::vector<double> y((size_t)n_);
      for (Index i = 0; i < n_; ++i) y[(size_t)i] = b.coeff(p_[(size_t)i], c);
      for (Index i = 0; i < n_; ++i)
        for (Index k = 0; k < i; ++k) y[(size_t)i] -= L_(i, k) * y[(size_t)k];
      for (Index i = 0; i < n_; ++i) y[(size_t)i] = std::abs(d_(i)) > 2.2250738585072014e-308 ? y[(size_t)i] / d_(i) : 0.0;
      for (Index i = n_ - 1; i >= 0; --i)
        for (Index k = i + 1; k < n_; ++k) y[(size_t)i] -= L_(k, i) * y[(size_t)k];
      for (Index i = 0; i < n_; ++i) x(p_[(size_t)i], c) = y[(size_t)i];
    }
    return x;
  }

 private:
  Index n_;
  MatrixXd L_;
  VectorXd d_;
  std::vector<Index> p_;
};
template <typename D>
LDLT<Matrix<typename MatrixBase<D>::Scalar, Dynamic, Dynamic>> MatrixBase<D>::ldlt() const {
  return LDLT<Matrix<Scalar, Dynamic, Dynamic>>(*this);
}

// One-sided Jacobi SVD (Hestenes): A = U diag(s) V^T, singular values sorted in decreasing order.
template <typename M>
class JacobiSVD {
 public:
  template <typename OD>
  JacobiSVD(const MatrixBase<OD>& A, unsigned int = 0) {
    const Index m = A.rows(), n = A.cols();
    MatrixXd W(A);  // columns are rotated until mutually orthogonal
    MatrixXd V = MatrixXd::Identity(n, n);
    for (int sweep = 0; sweep < 80; ++sweep) {
      double off = 0.0;
      for (Index p = 0; p < n - 1; ++p)
        for (Index q = p + 1; q < n; ++q) {
          double alpha = 0, beta = 0, gamma = 0;
          for (Index i = 0; i < m; ++i) {
            alpha += W(i, p) * W(i, p);
            beta += W(i, q) * W(i, q);
            gamma += W(i, p) * W(i, q);
          }
          if (gamma == 0.0 || std::abs(gamma) <= 1e-300) continue;
          off = std::max(off, std::abs(gamma) / std::sqrt(std::max(alpha * beta, 1e-300)));
          const double zeta = (beta - alpha) / (2.0 * gamma);
          const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::abs(zeta) + std::sqrt(1.0 + zeta * zeta));
          const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
          for (Index i = 0; i < m; ++i) {
            const double wp = W(i, p), wq = W(i, q);
            W(i, p) = c * wp - s * wq;
            W(i, q) = s * wp + c * wq;
          }
          for (Index i = 0; i < n; ++i) {
            const double vp = V(i, p), vq = V(i, q);
            V(i, p) = c * vp - s * vq;
            V(i, q) = s * vp + c * vq;
          }
        }
      if (off < 1e-15) break;
    }
    std::vector<double> sv((size_t)n);
    std::vector<Index> order((size_t)n);
    for (Index j = 0; j < n; ++j) {
      double s = 0;
      for (Index i = 0; i < m; ++i) s += W(i, j) * W(i, j);
      sv[(size_t)j] = std::sqrt(s);
      order[(size_t)j] = j;
    }
    std::sort(order.begin(), order.end(), [&](Index a, Index b) { return sv[(size_t)a] > sv[(size_t)b]; });
    s_.resize(n);
    U_.resize(m, n);
    V_.resize(n, n);
    for (Index k = 0; k < n; ++k) {
      const Index j = order[(size_t)k];
      s_(k) = sv[(size_t)j];
      for (Index i = 0; i < m; ++i) U_(i, k) = sv[(size_t)j] > 0 ? W(i, j) / sv[(size_t)j] : 0.0;
      for (Index i = 0; i < n; ++i) V_(i, k) = V(i, j);
    }
    // Eigen's U is unitary whatever the rank: columns of vanishing singular values are completed to an orthonormal
    // basis (Gram-Schmidt of the canonical vectors against the columns found so far)
    const double cutoff = n > 0 ? 1e-14 * s_(0) : 0.0;
    for (Index k = 0; k < n && k < m; ++k) {
      if (s_(k) > cutoff && s_(k) > 0.0) continue;
      double best_len = -1.0;
      std::vector<double> best((size_t)m, 0.0);
      for (Index e = 0; e < m; ++e) {
        std::vector<double> v((size_t)m, 0.0);
        v[(size_t)e] = 1.0;
        for (Index c = 0; c < n; ++c) {
          if (c == k || (c > k && !(s_(c) > cutoff && s_(c) > 0.0))) continue;
          double dot = 0.0;
          for (Index i = 0; i < m; ++i) dot += U_(i, c) * v[(size_t)i];
          for (Index i = 0; i < m; ++i) v[(size_t)i] -= dot * U_(i, c);
        }
        double len = 0.0;
        for (Index i = 0; i < m; ++i) len += v[(size_t)i] * v[(size_t)i];
        if (len > best_len) { best_len = len; best = v; }
      }
      const double nrm = std::sqrt(best_len);
      for (Index i = 0; i < m; ++i) U_(i, k) = best[(size_t)i] / nrm;
    }
  }
  const VectorXd& singularValues() const { return s_; }
  const MatrixXd& matrixU() const { return U_; }
  const MatrixXd& matrixV() const { return V_; }

 private:
  VectorXd s_;
  MatrixXd U_, V_;
};

// Eigen-decomposition for the (symmetric) 4x4 matrices the reference averages quaternions with;
// complex-typed accessors as in Eigen::EigenSolver, imaginary parts are zero.
template <typename M>
class EigenSolver {
 public:
  typedef std::complex<double> Cx;
  template <typename OD>
  explicit EigenSolver(const MatrixBase<OD>& Ain) {
    const Index n = Ain.rows();
    MatrixXd A(Ain), V = MatrixXd::Identity(n, n);
    for (int sweep = 0; sweep < 100; ++sweep) {
      double off = 0;
      for (Index p = 0; p < n; ++p)
        for (Index q = p + 1; q < n; ++q) off += A(p, q) * A(p, q);
      if (off < 1e-300) break;
      for (Index p = 0; p < n - 1; ++p)
        for (Index q = p + 1; q < n; ++q) {
          if (A(p, q) == 0.0) continue;
          const double theta = (A(q, q) - A(p, p)) / (2.0 * A(p, q));
          const double t = (theta >= 0 ? 1.0 : -1.0) / (std::abs(theta) + std::sqrt(theta * theta + 1.0));
          const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
          for (Index k = 0; k < n; ++k) {
            const double akp = A(k, p), akq = A(k, q);
            A(k, p) = c * akp - s * akq;
            A(k, q) = s * akp + c * akq;
          }
          for (Index k = 0; k < n; ++k) {
            const double apk = A(p, k), aqk = A(q, k);
            A(p, k) = c * apk - s * aqk;
            A(q, k) = s * apk + c * aqk;
          }
          for (Index k = 0; k < n; ++k) {
            const double vkp = V(k, p), vkq = V(k, q);
            V(k, p) = c * vkp - s * vkq;
            V(k, q) = s * vkp + c * vkq;
          }
        }
    }
    w_.resize(n, 1);
    V_.resize(n, n);
    for (Index i = 0; i < n; ++i) {
      w_(i, 0) = Cx(A(i, i), 0.0);
      for (Index j = 0; j < n; ++j) V_(i, j) = Cx(V(i, j), 0.0);
    }
  }
  const Matrix<Cx, Dynamic, 1>& eigenvalues() const { return w_; }
  const Matrix<Cx, Dynamic, Dynamic>& eigenvectors() const { return V_; }

 private:
  Matrix<Cx, Dynamic, 1> w_;
  Matrix<Cx, Dynamic, Dynamic> V_;
};

// ------------------------------------------------------------------------------------------
// Array (coefficient-wise): only what TranScanToPoints needs
template <typename T, int R, int C>
class Array {
 public:
  Array() : r_(0), c_(0) {}
  Array(Index r, Index c) : r_(r), c_(c), d_((size_t)(r * c), T(0)) {}
  Index rows() const { return r_; }
  Index cols() const { return c_; }
  T operator()(Index i, Index j) const { return d_[(size_t)(j * r_ + i)]; }
  T& operator()(Index i, Index j) { return d_[(size_t)(j * r_ + i)]; }
  Array operator*(const Array& o) const {
    assert(r_ == o.r_ && c_ == o.c_);
    Array r(r_, c_);
    for (size_t k = 0; k < d_.size(); ++k) r.d_[k] = d_[k] * o.d_[k];
    return r;
  }

 private:
  Index r_, c_;
  std::vector<T> d_;
};
typedef Array<double, Dynamic, Dynamic> ArrayXXd;

// ------------------------------------------------------------------------------------------
// Quaternions: coefficients stored (x, y, z, w) like Eigen
template <typename D, typename T>
class QuaternionBase {
 public:
  D& derived() { return *static_cast<D*>(this); }
  const D& derived() const { return *static_cast<const D*>(this); }
  T x() const { return derived().c(0); }
  T y() const { return derived().c(1); }
  T z() const { return derived().c(2); }
  T w() const { return derived().c(3); }
  T& x() { return derived().c(0); }
  T& y() { return derived().c(1); }
  T& z() { return derived().c(2); }
  T& w() { return derived().c(3); }
  Matrix<T, 4, 1> coeffs() const { return Matrix<T, 4, 1>(x(), y(), z(), w()); }
  T squaredNorm() const { return x() * x() + y() * y() + z() * z() + w() * w(); }
  T norm() const { return std::sqrt(squaredNorm()); }
  Quaternion<T> normalized() const {
    const T n = norm();
    return Quaternion<T>(w() / n, x() / n, y() / n, z() / n);
  }
  void normalize() {
    const T n = norm();
    x() /= n; y() /= n; z() /= n; w() /= n;
  }
  Quaternion<T> conjugate() const { return Quaternion<T>(w(), -x(), -y(), -z()); }
  Quaternion<T> inverse() const {
    const T n2 = squaredNorm();
    return Quaternion<T>(w() / n2, -x() / n2, -y() / n2, -z() / n2);
  }
  template <typename OD>
  Quaternion<T> operator*(const QuaternionBase<OD, T>& b) const {
    const QuaternionBase& a = *this;
    return Quaternion<T>(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                         a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                         a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                         a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
  }
  // rotation matrix of a (not necessarily unit) quaternion, no normalisation
  Matrix<T, 3, 3> toRotationMatrix() const {
    Matrix<T, 3, 3> R;
    const T tx = T(2) * x(), ty = T(2) * y(), tz = T(2) * z();
    const T twx = tx * w(), twy = ty * w(), twz = tz * w();
    const T txx = tx * x(), txy = ty * x(), txz = tz * x();
    const T tyy = ty * y(), tyz = tz * y(), tzz = tz * z();
    R(0, 0) = T(1) - (tyy + tzz); R(0, 1) = txy - twz;          R(0, 2) = txz + twy;
    R(1, 0) = txy + twz;          R(1, 1) = T(1) - (txx + tzz); R(1, 2) = tyz - twx;
    R(2, 0) = txz - twy;          R(2, 1) = tyz + twx;          R(2, 2) = T(1) - (txx + tyy);
    return R;
  }
  Matrix<T, 3, 1> operator*(const Matrix<T, 3, 1>& v) const { return toRotationMatrix() * v; }
};

template <typename T>
class Quaternion : public QuaternionBase<Quaternion<T>, T> {
 public:
  Quaternion() { q_[0] = q_[1] = q_[2] = T(0); q_[3] = T(1); }
  Quaternion(const T& w, const T& x, const T& y, const T& z) { q_[0] = x; q_[1] = y; q_[2] = z; q_[3] = w; }
  Quaternion(const Quaternion&) = default;
  template <typename OD>
  Quaternion(const QuaternionBase<OD, T>& o) { q_[0] = o.x(); q_[1] = o.y(); q_[2] = o.z(); q_[3] = o.w(); }
  // from a 3x3 rotation matrix (largest-diagonal branch selection)
  template <typename OD>
  explicit Quaternion(const MatrixBase<OD>& m) {
    assert(m.rows() == 3 && m.cols() == 3);
    T t = m.coeff(0, 0) + m.coeff(1, 1) + m.coeff(2, 2);
    if (t > T(0)) {
      t = std::sqrt(t + T(1));
      q_[3] = T(0.5) * t;
      t = T(0.5) / t;
      q_[0] = (m.coeff(2, 1) - m.coeff(1, 2)) * t;
      q_[1] = (m.coeff(0, 2) - m.coeff(2, 0)) * t;
      q_[2] = (m.coeff(1, 0) - m.coeff(0, 1)) * t;
    } else {
      int i = 0;
      if (m.coeff(1, 1) > m.coeff(0, 0)) i = 1;
      if (m.coeff(2, 2) > m.coeff(i, i)) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      t = std::sqrt(m.coeff(i, i) - m.coeff(j, j) - m.coeff(k, k) + T(1));
      q_[i] = T(0.5) * t;
      t = T(0.5) / t;
      q_[3] = (m.coeff(k, j) - m.coeff(j, k)) * t;
      q_[j] = (m.coeff(j, i) + m.coeff(i, j)) * t;
      q_[k] = (m.coeff(k, i) + m.coeff(i, k)) * t;
    }
  }
  Quaternion& operator=(const Quaternion&) = default;
  template <typename OD>
  Quaternion& operator=(const QuaternionBase<OD, T>& o) {
    const T x = o.x(), y = o.y(), z = o.z(), w = o.w();
    q_[0] = x; q_[1] = y; q_[2] = z; q_[3] = w;
    return *this;
  }
  static Quaternion Identity() { return Quaternion(T(1), T(0), T(0), T(0)); }
  T c(int i) const { return q_[i]; }
  T& c(int i) { return q_[i]; }

 private:
  T q_[4];
};
typedef Quaternion<double> Quaterniond;

// Rotation by `angle` about the unit vector `axis`; products compose as quaternions (like Eigen's RotationBase).
template <typename T>
class AngleAxis {
 public:
  template <typename OD>
  AngleAxis(const T& angle, const MatrixBase<OD>& axis) : angle_(angle), axis_(axis) {}
  Quaternion<T> toQuaternion() const {
    const T s = std::sin(angle_ * T(0.5)), c = std::cos(angle_ * T(0.5));
    return Quaternion<T>(c, s * axis_(0), s * axis_(1), s * axis_(2));
  }
  Matrix<T, 3, 3> toRotationMatrix() const { return toQuaternion().toRotationMatrix(); }
  Quaternion<T> operator*(const AngleAxis& o) const { return toQuaternion() * o.toQuaternion(); }
  friend Quaternion<T> operator*(const Quaternion<T>& q, const AngleAxis& a) { return q * a.toQuaternion(); }

 private:
  T angle_;
  Matrix<T, 3, 1> axis_;
};
typedef AngleAxis<double> AngleAxisd;

template <typename T>
class Map<Quaternion<T>> : public QuaternionBase<Map<Quaternion<T>>, T> {
 public:
  explicit Map(T* p) : p_(p) {}
  template <typename OD>
  Map& operator=(const QuaternionBase<OD, T>& o) {
    const T x = o.x(), y = o.y(), z = o.z(), w = o.w();
    p_[0] = x; p_[1] = y; p_[2] = z; p_[3] = w;
    return *this;
  }
  T c(int i) const { return p_[i]; }
  T& c(int i) { return p_[i]; }

 private:
  T* p_;
};
template <typename T>
class Map<const Quaternion<T>> : public QuaternionBase<Map<const Quaternion<T>>, T> {
 public:
  explicit Map(const T* p) : p_(p) {}
  T c(int i) const { return p_[i]; }
  T& c(int i) { return const_cast<T*>(p_)[i]; }

 private:
  const T* p_;
};

}  // namespace Eigen
