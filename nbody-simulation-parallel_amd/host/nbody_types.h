// nbody_types.h -- the two value types of the hot path, layout-locked to the reference's:
//   Vector<D>: D packed doubles (nbody-sim-new/vector.h:9-12: one std::array<double, D> member `components`)
//   Body<D>  : position, velocity, mass (nbody-sim-new/body.h:8-11): 56 B for D=3, 40 B for D=2, no padding
// Written from scratch for this repository's host code.  When methods_hip.h is dropped into the reference
// tree this header is absent and the reference's own body.h / vector.h are used (same names, same memory).
#ifndef NBODY_AMD_TYPES_H
#define NBODY_AMD_TYPES_H

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


// ---- Body<D>: the memory the C ABI (include/nbody_hip.h) reads and writes -----------------------------


template <int D>
struct Body {
    Vector<D> position;
    Vector<D> velocity;
    double mass = 0.0;

    Body() = default;
    Body(const Vector<D>& p, double m) : position(p), mass(m) {}
    Body(const Vector<D>& p, const Vector<D>& v, double m) : position(p), velocity(v), mass(m) {}
};

using Body2D = Body<2>;
using Body3D = Body<3>;

static_assert(sizeof(Body<3>) == 56 && offsetof(Body<3>, velocity) == 24 && offsetof(Body<3>, mass) == 48, "Body<3> layout");
static_assert(sizeof(Body<2>) == 40 && offsetof(Body<2>, velocity) == 16 && offsetof(Body<2>, mass) == 32, "Body<2> layout");


#endif  // NBODY_AMD_TYPES_H
