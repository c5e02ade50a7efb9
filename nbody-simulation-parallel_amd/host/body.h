// body.h -- point mass, layout-locked to the reference's Body<D> (nbody-sim-new/body.h:8-11):
//   Vector<D> position; Vector<D> velocity; double mass;   56 B for D=3 (40 B for D=2), no padding.
// This is the memory the C ABI (include/nbody_hip.h) reads and writes.
#ifndef NBODY_AMD_BODY_H
#define NBODY_AMD_BODY_H

#include <cstddef>

#include "vector.h"

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

#endif  // NBODY_AMD_BODY_H
