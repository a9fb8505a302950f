# long_term_plannerConfig.cmake — package file of the MI355X drop-in.
#
# The reference installs a CMake package `long_term_planner` (version 1.0.0) that exports the target
# long_term_planner::long_term_planner (/root/reference/CMakeLists.txt:17-57, cmake/Config.cmake.in). A project that does
#     find_package(long_term_planner REQUIRED)
#     target_link_libraries(app long_term_planner::long_term_planner)
# keeps working against this repository: point CMake at this directory (-Dlong_term_planner_DIR=<repo>/cmake, or add <repo>
# to CMAKE_PREFIX_PATH). The target carries the drop-in headers (include/long_term_planner/long_term_planner.h, roots.h,
# ltp_hip.h) and links longtermplanner_amd/libltp_hip.so, the gfx950 library built by `make -C longtermplanner_amd/csrc`.
# Unlike the reference's target it does not pull in Eigen3 (the root finder runs on the device).
get_filename_component(_ltp_root "${CMAKE_CURRENT_LIST_DIR}/.." ABSOLUTE)
set(_ltp_lib "${_ltp_root}/longtermplanner_amd/libltp_hip.so")
if(NOT EXISTS "${_ltp_lib}")
  set(long_term_planner_FOUND FALSE)
  set(long_term_planner_NOT_FOUND_MESSAGE
      "libltp_hip.so has not been built: run `make -C ${_ltp_root}/longtermplanner_amd/csrc` (hipcc, gfx950)")
  return()
endif()
if(NOT TARGET long_term_planner::long_term_planner)
  add_library(long_term_planner::long_term_planner SHARED IMPORTED)
  set_target_properties(long_term_planner::long_term_planner PROPERTIES
    IMPORTED_LOCATION "${_ltp_lib}"
    IMPORTED_NO_SONAME TRUE
    INTERFACE_INCLUDE_DIRECTORIES "${_ltp_root}/include"
    INTERFACE_COMPILE_FEATURES cxx_std_17)
  # the drop-in class guards its lazily created device handle with a mutex
  find_package(Threads QUIET)
  if(TARGET Threads::Threads)
    set_property(TARGET long_term_planner::long_term_planner APPEND PROPERTY INTERFACE_LINK_LIBRARIES Threads::Threads)
  endif()
endif()
set(long_term_planner_INCLUDE_DIRS "${_ltp_root}/include")
set(long_term_planner_LIBRARIES long_term_planner::long_term_planner)
unset(_ltp_root)
unset(_ltp_lib)
