// vector.h -- D-component fp64 vector, layout-locked to the reference's Vector<D>
// (nbody-sim-new/vector.h:9-12: one std::array<double, D> member named `components`, no padding).
// Written from scratch for this repository's host code; when methods_hip.h is dropped into the
// reference tree, the reference's own vector.h is used instead (same names, same memory).
#ifndef NBODY_AMD_VECTOR_H
#define NBODY_AMD_VECTOR_H

#include <array>
#include <cmath>
#include <cstddef>

template <int D>
class Vector {
    static_assert(D == 2 || D == 3, "Vector<D>: D must be 2 or 3");

public:
    std::array<double, D> components{};  // zero-initialised, like the reference's default constructor

    constexpr Vector() = default;
    constexpr explicit Vector(const std::array<double, D>& v) : components(v) {}

    constexpr double& operator[](int i) { return components[static_cast<std::size_t>(i)]; }
    constexpr const double& operator[](int i) const { return components[static_cast<std::size_t>(i)]; }

    // element-wise arithmetic; every operation rounds once per component, in index order
    friend Vector operator+(Vector a, const Vector& b) { return a += b; }
    friend Vector operator-(Vector a, const Vector& b) { return a -= b; }
    friend Vector operator*(Vector a, double s) { return a *= s; }
    friend Vector operator*(double s, Vector a) { return a *= s; }
    friend Vector operator/(Vector a, double s) { return a /= s; }

    Vector& operator+=(const Vector& o) { for (int k = 0; k < D; ++k) components[k] += o.components[k]; return *this; }
    Vector& operator-=(const Vector& o) { for (int k = 0; k < D; ++k) components[k] -= o.components[k]; return *this; }
    Vector& operator*=(double s) { for (int k = 0; k < D; ++k) components[k] *= s; return *this; }
    Vector& operator/=(double s) { for (int k = 0; k < D; ++k) components[k] /= s; return *this; }

    // sum of squares accumulated left to right from 0.0 (vector.h:81-85 semantics)
    double magnitude_squared() const {
        double acc = 0.0;
        for (int k = 0; k < D; ++k) acc += components[k] * components[k];
        return acc;
    }
    double magnitude() const { return std::sqrt(magnitude_squared()); }
    double dot(const Vector& o) const {
        double acc = 0.0;
        for (int k = 0; k < D; ++k) acc += components[k] * o.components[k];
        return acc;
    }
    // unit vector by component-wise DIVISION; the zero vector below 1e-10 (vector.h:93-97 semantics)
    Vector normalized() const {
        const double len = magnitude();
        return len < 1e-10 ? Vector() : *this / len;
    }
};

template <int D>
bool operator!=(const Vector<D>& a, const Vector<D>& b) { return a.components != b.components; }

using Vector2D = Vector<2>;
using Vector3D = Vector<3>;

static_assert(sizeof(Vector<2>) == 16 && sizeof(Vector<3>) == 24, "Vector<D> must be D packed doubles");

#endif  // NBODY_AMD_VECTOR_H
