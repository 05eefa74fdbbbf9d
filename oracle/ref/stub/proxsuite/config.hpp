// TEST INFRASTRUCTURE ONLY.
// Stand-in for the header ProxSuite's CMake build generates (proxsuite/config.hpp: version macros
// and export declarations).  oracle/ref/build_ref.sh puts this directory on the include path in
// front of /root/reference/include so that the reference's own headers compile without running its
// build system.  v0.7.2 = the tree under /root/reference (package.xml).
#ifndef PROXSUITE_CONFIG_HPP_STUB
#define PROXSUITE_CONFIG_HPP_STUB
#define PROXSUITE_VERSION "0.7.2"
#define PROXSUITE_MAJOR_VERSION 0
#define PROXSUITE_MINOR_VERSION 7
#define PROXSUITE_PATCH_VERSION 2
#define PROXSUITE_VERSION_AT_LEAST(major, minor, patch)                                             \
  (PROXSUITE_MAJOR_VERSION > major ||                                                               \
   (PROXSUITE_MAJOR_VERSION >= major &&                                                             \
    (PROXSUITE_MINOR_VERSION > minor || (PROXSUITE_MINOR_VERSION >= minor && PROXSUITE_PATCH_VERSION >= patch))))
#define PROXSUITE_VERSION_AT_MOST(major, minor, patch)                                              \
  (PROXSUITE_MAJOR_VERSION < major ||                                                               \
   (PROXSUITE_MAJOR_VERSION <= major &&                                                             \
    (PROXSUITE_MINOR_VERSION < minor || (PROXSUITE_MINOR_VERSION <= minor && PROXSUITE_PATCH_VERSION <= patch))))
#define PROXSUITE_DLLAPI
#endif
