// proxsuite/proxqp/dense/dense.hpp -- umbrella header of the dense ProxQP API, MI355X build
// (reference include/proxsuite/proxqp/dense/dense.hpp).
#ifndef PROXSUITE_AMD_PROXQP_DENSE_DENSE_HPP
#define PROXSUITE_AMD_PROXQP_DENSE_DENSE_HPP

#include "proxsuite/proxqp/dense/compute_ECJ.hpp"
#include "proxsuite/proxqp/dense/wrapper.hpp"
#include "proxsuite/proxqp/timings.hpp"

#endif
